"""Whole-model training step through the HIP autograd path vs the CPU oracle under torch autograd (tiny W8 model,
BatchNorm batch statistics, dropout / drop-path off): the 7 outputs, the two-term ST_OHKW loss of
script/Common.py:122-130 and the gradient of every parameter."""
import pytest
import torch

from oracle import otpose_oracle as O
from otpose_amd import OTPose, tiny_cfg
from otpose_amd import synthetic as S
from otpose_amd import train as TR
from tests.conftest import seeded

pytestmark = pytest.mark.gpu


def _targets(b, j, h, w):
    g = seeded((b, j, h, w), 11).abs() * 0.2
    g[:, ::2, 3, 4] = 1.0                                   # exact-1 peaks for every second joint
    g.clamp_(max=1.0)
    wt = (seeded((b, j, 1), 12) > -1.0).float()
    return g, wt


def _rel_stats(named_grads, leaves):
    stats = []
    for name, gq in named_grads:
        ref = leaves[name].grad
        if ref is None:
            continue
        g = gq.cpu().double().flatten()
        r = ref.double().flatten()
        nr = float(r.norm())
        stats.append((float((g - r).norm()) / max(nr, 1e-12), float(torch.dot(g, r)) / max(float(g.norm()) * nr, 1e-24), name, nr))
    stats.sort(reverse=True)
    num = sum(((gq.cpu().double() - leaves[n].grad) ** 2).sum() for n, gq in named_grads if leaves[n].grad is not None)
    den = sum((leaves[n].grad ** 2).sum() for n, gq in named_grads if leaves[n].grad is not None)
    return stats, float(num / den) ** 0.5


# Tolerances against the float64 oracle.  f32: see the calibration note below.  bf16 (BASELINE configs[2]: bf16 activations
# in the backbone, fp32 accumulation / statistics / master weights): every stored activation and activation gradient of
# the 40-stage HRNet carries 2^-9 relative rounding, so heat-maps are held to 3e-2 of their range, the loss to 3e-2
# relative, and gradients to the L2 / cosine bounds listed (measured on MI355X: see DESIGN.md section 5).
TOL = {
    "f32": dict(out=1e-3, loss=1e-3, med=8e-3, glob=3e-3, gcos=0.9999, gnorm=1e-3, rel=0.15, cos=0.99),
    # measured on MI355X: outputs <= 7.3e-2 of their range, loss 2.3e-3; the gradient figures of this fixture are not bounded for
    # bf16 (see the note in the test: 0.13 .. 2.25 over input seeds for either conv kernel)
    "bf16": dict(out=1e-1, loss=1e-2, med=10.0, glob=0.25, gcos=0.98, gnorm=0.15, rel=100.0, cos=-1.0),
}


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_train_step_matches_oracle_autograd(dtype):
    tol = TOL[dtype]
    cfg = tiny_cfg(8, (64, 96))
    model = OTPose(cfg)
    S.fill_synthetic_(model)
    sd_cpu = {k: v.detach().clone() for k, v in model.state_dict().items()}
    # float64 reference: a float32 CPU graph carries rounding noise of the same size as the path under test
    leaves = {k: v.double().requires_grad_() for k, v in sd_cpu.items()
              if v.is_floating_point() and k in dict(model.named_parameters())}
    sd_ref = {k: (v.double() if v.is_floating_point() else v) for k, v in sd_cpu.items()}
    sd_ref.update(leaves)
    x, margin = S.synthetic_clip(2, cfg.MODEL.IMAGE_SIZE)
    outs_ref = O.otpose_forward(sd_ref, cfg, x.double(), margin, training_bn=True)
    B, J, h, w = outs_ref[0].shape
    g, wt = _targets(B, J, h, w)
    gd, wd = g.double(), wt.double()
    l1 = O.st_ohkw_mse_loss(outs_ref[0], outs_ref[1][:B], gd, wd)["final_loss"]
    l2 = O.st_ohkw_mse_loss(outs_ref[4], outs_ref[4], (gd + outs_ref[2]) / 2, wd)["final_loss"]
    loss_ref = l1 + l2
    loss_ref.backward()

    model = model.cuda().train()
    model.train_dropout = False
    model.train_dtype = dtype
    outs = model(x.cuda(), margin=margin.cuda())
    errs = {}
    for name, o, r in zip(("output", "rough", "intersection", "prev_b", "context", "squeezed", "total_b"), outs, outs_ref):
        errs[name] = (float((o.detach().cpu().double() - r.detach()).abs().max()), float(r.detach().abs().max()))
        print("%s: max abs err %.3e (ref max %.3e)" % ((name,) + errs[name]))
    for name, (err, mx) in errs.items():
        assert err <= tol["out"] * max(1.0, mx), f"{name}: {err}"
    loss = TR.criterion(outs, g.cuda(), wt.cuda())
    print("loss %.6f ref %.6f" % (float(loss), float(loss_ref.detach())))
    assert abs(float(loss) - float(loss_ref.detach())) <= tol["loss"] * max(1.0, abs(float(loss_ref.detach()))), (float(loss), float(loss_ref.detach()))
    loss.backward()
    # fp32 on both sides: a ReLU / max-pool / top-k decision that flips on a last-bit difference moves single gradient
    # entries by O(1) of their size, so tensors are compared in the L2 sense (relative error and cosine)
    named = [(n, p.grad) for n, p in model.named_parameters() if leaves[n].grad is not None]
    for n, p in model.named_parameters():
        if leaves[n].grad is None:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, n
        else:
            assert p.grad is not None, n
    stats, glob = _rel_stats(named, leaves)
    print("params checked", len(stats))
    for rel, cos, name, nr in [s_ for s_ in stats if s_[3] > 1e-6][:8]:
        print("  rel L2 err %.3e  cos %.6f  |ref| %.3e  %s" % (rel, cos, nr, name))
    assert len(stats) > 300
    med = sorted(s_[0] for s_ in stats)[len(stats) // 2]
    gq = torch.cat([g_.cpu().double().flatten() for _, g_ in named])
    gr = torch.cat([leaves[n].grad.flatten() for n, _ in named])
    gcos = float(torch.dot(gq, gr) / (gq.norm() * gr.norm()))
    worst = max(s_[0] for s_ in stats if s_[3] > 1e-6)
    mincos = min(s_[1] for s_ in stats if s_[3] > 1e-6)
    print("%s: median rel L2 %.3e, whole-gradient rel L2 %.3e, whole-gradient cosine %.5f, |g| %.4e vs %.4e, worst tensor %.3e, "
          "min cosine %.4f" % (dtype, med, glob, gcos, float(gq.norm()), float(gr.norm()), worst, mincos))
    # calibration (tools/grad_noise.py, 5 input seeds): against the float64 gradients the oracle in float32 on the CPU,
    # the oracle through PyTorch-ROCm eager in float32 and this HIP path all scatter in the same band - median relative L2
    # 4e-5 .. 6e-3, worst tensor 3e-2 .. 1e-1 - because the gradients reaching the encoders pass the offset branch of the
    # DCN (differences of neighbouring samples) and 40 BatchNorm+ReLU stages at batch 2: the fixture amplifies rounding by
    # ~1e4.  That band is the rounding floor of this fixture in fp32; with bf16 activations in the backbone (2^-9 per stored
    # value) the same amplification saturates the per-tensor errors of the small encoder tensors, so the bf16 step is held to
    # the whole-gradient error / cosine / norm here and per tensor in test_backbone_gradients_match_oracle below.
    if dtype == "bf16":
        # Round 5: NOT bounded for bf16.  The whole-gradient figures of this fixture are chaos, not accuracy: over four input seeds and
        # the two conv kernels of the bf16 backbone (same products, fp32 sums in another order: tools/bf16_e2e_spread.py,
        # profiles/r05_bf16_e2e_spread.txt) the relative L2 error ranges 0.13 .. 2.25 and the cosine -0.48 .. 0.994 for BOTH kernels -
        # BatchNorm over as few as 60 values and ReLU decisions that flip on a last-bit difference.  What bounds the bf16 gradients:
        # every conv + BN layer in context against float64 on its own operands (test_gpu_train_bf16_yardstick.py), the backbone end
        # to end against the rounding-aware oracle (test_backbone_gradients_match_oracle below), the ops against fp64 autograd
        # (test_gpu_bf16_ops.py), and cfg3 at full size against the fp32 step (bench.py: train_step.full_size_checks).
        assert bool(torch.isfinite(gq).all()) and float(gq.norm()) > 0.0
    else:
        assert med <= tol["med"] and glob <= tol["glob"] and gcos >= tol["gcos"]
        assert abs(float(gq.norm()) / float(gr.norm()) - 1.0) <= tol["gnorm"]
        assert worst <= tol["rel"] and mincos >= tol["cos"]
    # BatchNorm running statistics were updated like nn.BatchNorm2d does
    rm = model.state_dict()["rough_pose_estimation_net.bn1.running_mean"].cpu()
    assert float((rm - sd_cpu["rough_pose_estimation_net.bn1.running_mean"]).abs().max()) > 0


def test_train_forward_dropout_and_drop_path():
    """Dropout(0.1) / drop-path(0.1) of the ConvTransformers (blocks.py:251-253, 298-316, 450) are active under
    model.train(): seeded draws repeat, different draws change the DCN output but not the HRNet heat-maps, and the
    gradients still reach every encoder parameter."""
    cfg = tiny_cfg(8, (64, 96))
    model = OTPose(cfg)
    S.fill_synthetic_(model)
    model = model.cuda().train()
    x, margin = S.synthetic_clip(2, cfg.MODEL.IMAGE_SIZE)
    x, margin = x.cuda(), margin.cuda()
    torch.manual_seed(5)
    a = model(x, margin=margin)
    torch.manual_seed(5)
    b = model(x, margin=margin)
    torch.manual_seed(6)
    c = model(x, margin=margin)
    assert torch.equal(a[0], b[0]) and torch.equal(a[4], b[4])
    assert torch.equal(a[1], c[1])                                    # HRNet has no stochastic layer
    assert float((a[0] - c[0]).abs().max()) > 0 and float((a[4] - c[4]).abs().max()) > 0
    model.train_dropout = False
    d = model(x, margin=margin)
    assert float((a[0] - d[0]).abs().max()) > 0
    B, J, h, w = c[0].shape
    g, wt = _targets(B, J, h, w)
    TR.criterion(c, g.cuda(), wt.cuda()).backward()
    for name, p in model.named_parameters():
        if name.startswith(("temporal_encoder", "flow_encoder")):
            assert p.grad is not None and bool(torch.isfinite(p.grad).all()), name


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_backbone_gradients_match_oracle(dtype):
    """The HRNet backbone alone (model/HRNet.py:116-152, BatchNorm batch statistics) under a plain heat-map MSE: heat-maps and
    the gradient of every backbone parameter vs the float64 oracle.  Without the DCN offset branch and the top-k loss behind
    it this fixture is not chaotic, so it bounds what the bf16 path itself loses: 2^-9 relative rounding of every stored
    activation / activation gradient across 40 conv + BatchNorm stages."""
    cfg = tiny_cfg(16, (128, 192))
    model = OTPose(cfg)
    S.fill_synthetic_(model)
    pre = "rough_pose_estimation_net"
    sd_cpu = {k: v.detach().clone() for k, v in model.state_dict().items()}
    names = [k for k, _ in model.named_parameters() if k.startswith(pre)]
    leaves = {k: sd_cpu[k].double().requires_grad_() for k in names}
    sd_ref = {k: (v.double() if v.is_floating_point() else v) for k, v in sd_cpu.items()}
    sd_ref.update(leaves)
    x, _ = S.synthetic_clip(2, cfg.MODEL.IMAGE_SIZE)
    frames = torch.cat(x.split(3, dim=1), 0)
    stages = [cfg["MODEL"]["EXTRA"][f"STAGE{s}"] for s in (2, 3, 4)]
    ref = O.hrnet_forward(sd_ref, pre, frames.double(), stages, training=True)
    tgt = seeded(tuple(ref.shape), 21).abs() * 0.3
    (0.5 * ((ref - tgt.double()) ** 2).mean()).backward()

    model = model.cuda().train()
    graph = (TR.TrainGraphBF16 if dtype == "bf16" else TR.TrainGraph)(model)
    out = graph.hrnet(pre, graph.hrnet_input(x.cuda()))
    (0.5 * ((out - tgt.cuda()) ** 2).mean()).backward()
    err = float((out.detach().cpu().double() - ref.detach()).abs().max()) / float(ref.detach().abs().max())
    P = dict(model.named_parameters())
    stats, glob = _rel_stats([(n, P[n].grad) for n in names], leaves)
    live = [s_ for s_ in stats if s_[3] > 1e-9]
    med = sorted(s_[0] for s_ in live)[len(live) // 2]
    print("\nBACKBONE %s: heat-map max err / range %.3e, whole-gradient rel L2 %.3e, median %.3e, worst %.3e (%s), min cos %.4f"
          % (dtype, err, glob, med, live[0][0], live[0][2], min(s_[1] for s_ in live)))
    # tensors that carry the gradient (norm >= 1 % of the largest): BatchNorm biases in front of another BatchNorm have
    # gradients that cancel to ~0, where a relative error says nothing
    big = max(s_[3] for s_ in live)
    sig = [s_ for s_ in live if s_[3] >= 1e-2 * big]
    print("   significant tensors %d: worst rel L2 %.3e, min cos %.5f" % (len(sig), max(s_[0] for s_ in sig), min(s_[1] for s_ in sig)))
    # measured on MI355X (bf16): heat-maps 3.4e-2 of their range, whole-gradient rel L2 3.3e-3 (fp32: 3.4e-6 / 4.0e-5)
    tol = {"f32": dict(out=1e-4, glob=1e-3, med=5e-3, rel=5e-2, cos=0.995, sig=1e-3),
           "bf16": dict(out=6e-2, glob=1e-2, med=5e-1, rel=2.0, cos=0.0, sig=0.3)}[dtype]
    assert err <= tol["out"] and glob <= tol["glob"] and med <= tol["med"]
    assert live[0][0] <= tol["rel"] and min(s_[1] for s_ in live) >= tol["cos"]
    assert max(s_[0] for s_ in sig) <= tol["sig"]


def test_bf16_step_on_a_cfg2_shaped_clip():
    """BASELINE configs[2] at its real shape (HRNet-W48, one 5-frame 384x288 clip) vs the float64 oracle under torch
    autograd (dropout / drop-path off), the float32 HIP step beside it as the rounding floor:
      * whole step: loss and the backbone-side outputs (rough heat-maps, total_b);
      * gradients: the backbone under a plain heat-map MSE - |g|, direction and relative L2 error.
    Why not the whole-step gradient: on the seeded synthetic weights the flow encoder's channel LayerNorm (C = 17) sees
    background tokens whose variance is ~eps, so d(loss)/d(total_b) is dominated by a few tokens amplified by
    rstd ~ 1/sqrt(eps); the 2.7 % bf16 rounding noise of the backbone moves exactly those variances (measured: |g| 103 in
    fp32 vs 2.2 with bf16 activations, for IDENTICAL gradients arriving at the encoder output) - a property of the
    fixture, not of the kernels, so it cannot carry a tolerance."""
    from otpose_amd import cfg2
    cfg = cfg2()
    model = OTPose(cfg)
    S.fill_synthetic_(model)
    sd_cpu = {k: v.detach().clone() for k, v in model.state_dict().items()}
    pre = "rough_pose_estimation_net"
    names = [k for k, _ in model.named_parameters() if k.startswith(pre)]
    x, margin = S.synthetic_clip(1, cfg.MODEL.IMAGE_SIZE)
    torch.set_num_threads(16)
    # ---- whole step, forward: loss + outputs ------------------------------------------------------------------------
    sd_ref = {k: (v.double() if v.is_floating_point() else v) for k, v in sd_cpu.items()}
    with torch.no_grad():
        outs_ref = O.otpose_forward(sd_ref, cfg, x.double(), margin, training_bn=True)
        B, J, h, w = outs_ref[0].shape
        g, wt = _targets(B, J, h, w)
        gd, wd = g.double(), wt.double()
        loss_ref = float(O.st_ohkw_mse_loss(outs_ref[0], outs_ref[1][:B], gd, wd)["final_loss"]
                         + O.st_ohkw_mse_loss(outs_ref[4], outs_ref[4], (gd + outs_ref[2]) / 2, wd)["final_loss"])
    # ---- backbone gradient reference ---------------------------------------------------------------------------------
    leaves = {k: sd_cpu[k].double().requires_grad_() for k in names}
    sd_g = dict(sd_ref)
    sd_g.update(leaves)
    frames = torch.cat(x.split(3, dim=1), 0)
    stages = [cfg["MODEL"]["EXTRA"][f"STAGE{s_}"] for s_ in (2, 3, 4)]
    rough_ref = O.hrnet_forward(sd_g, pre, frames.double(), stages, training=True)
    tgt = seeded(tuple(rough_ref.shape), 21).abs() * 0.3
    (0.5 * ((rough_ref - tgt.double()) ** 2).mean()).backward()
    gr = torch.cat([leaves[n].grad.flatten() for n in names])
    res = {}
    for dtype in ("f32", "bf16"):
        m = OTPose(cfg)
        m.load_state_dict(sd_cpu)
        m = m.cuda().train()
        m.train_dropout = False
        m.train_dtype = dtype
        with torch.no_grad():
            outs = m(x.cuda(), margin=margin.cuda())
            loss = float(TR.criterion(outs, g.cuda(), wt.cuda()))
        rel = lambda a, r: float((a.detach().cpu().double() - r).abs().max() / r.abs().max())     # noqa: E731
        e_rough, e_total = rel(outs[1], outs_ref[1]), rel(outs[6], outs_ref[6])
        del outs
        graph = (TR.TrainGraphBF16 if dtype == "bf16" else TR.TrainGraph)(m)
        out = graph.hrnet(pre, graph.hrnet_input(x.cuda()))
        (0.5 * ((out - tgt.cuda()) ** 2).mean()).backward()
        P = dict(m.named_parameters())
        gq = torch.cat([P[n].grad.cpu().double().flatten() for n in names])
        res[dtype] = (abs(loss - loss_ref) / abs(loss_ref), e_rough, e_total, float(gq.norm() / gr.norm()),
                      float(torch.dot(gq, gr) / (gq.norm() * gr.norm())), float((gq - gr).norm() / gr.norm()))
        print("cfg2 clip %s: loss rel err %.3e, rough / total_b max err over range %.3e / %.3e; backbone gradient |g|/|g_ref| "
              "%.4f, cosine %.6f, rel L2 %.3e" % ((dtype,) + res[dtype]))
        del m, graph, out, P
        torch.cuda.empty_cache()
    f, b = res["f32"], res["bf16"]
    assert f[0] <= 1e-4 and f[1] <= 1e-3 and abs(f[3] - 1) <= 1e-3 and f[4] >= 0.9999 and f[5] <= 5e-3
    # bf16, measured on MI355X: loss 9.7e-3, rough / total_b 3.4e-2 / 2.3e-2 of their range, backbone gradient |g| ratio 1.0000,
    # cosine 0.999999, relative L2 error 1.7e-3 (fp32: 2.8e-6, 3.7e-6 / 2.0e-6, 1.0000, 1.000000, 1.6e-5)
    assert b[0] <= 3e-2 and b[1] <= 8e-2 and b[2] <= 8e-2 and abs(b[3] - 1) <= 1e-2 and b[4] >= 0.9999 and b[5] <= 1e-2


def test_bf16_step_same_with_and_without_streams_and_batched_packs(monkeypatch):
    """The scheduling of the bf16 step - HRNet branches / temporal encoders on side streams, every weight re-layout in one
    launch per step (bf16_ops.PackCache), BasicBlocks and attention fronts as single autograd nodes (skip gradient added in
    the input-gradient conv, intermediates rebuilt in the backward) - must not change its arithmetic: with all of it
    switched off (one stream, a pack launch per use, per-layer nodes) three forward / backward passes give the same losses
    and gradients.  The weights are rescaled by 10 % between the passes, so an operator left over from the previous pass (a job
    missed by the batched launch, a launch ordered before the rescale) would show as a 10 % error; pass 2 and 3 are the ones
    that read batched re-layouts.  The forward has no atomics: its outputs must agree bit for bit.  The backward has a few
    (DCN input gradient, fp32 weight gradients), and on this fixture a last-bit difference entering the flow encoder's
    LayerNorm comes out of the backbone as up to 3e-3 (see test_bf16_step_on_a_cfg2_shaped_clip), so the gradients are held
    to 1e-2 - ten times under what a stale operator does.
    (No optimizer in the loop: Adam's first steps are lr * sign(g) and turn such last-bit differences into 1e-4 loss
    differences.)"""
    cfg = tiny_cfg()
    ref = OTPose(cfg)
    S.fill_synthetic_(ref)
    sd = {k: v.detach().clone() for k, v in ref.state_dict().items()}
    x, margin = S.synthetic_clip(2, cfg.MODEL.IMAGE_SIZE)
    x, margin = x.cuda(), margin.cuda()
    w, h = cfg.MODEL.HEATMAP_SIZE
    g, wt = _targets(2, cfg.MODEL.NUM_JOINTS, h, w)
    g, wt = g.cuda(), wt.cuda()
    runs = {}
    for name, env in (("plain", {"OTPOSE_TRAIN_STREAMS": "0", "OTPOSE_PACK_BATCH": "0", "OTPOSE_BLOCK_FUSE": "0",
                                 "OTPOSE_TRAIN_RECOMPUTE": "0"}), ("scheduled", {})):
        for k in ("OTPOSE_TRAIN_STREAMS", "OTPOSE_PACK_BATCH", "OTPOSE_BLOCK_FUSE", "OTPOSE_TRAIN_RECOMPUTE"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        m = OTPose(cfg)
        m.load_state_dict(sd)
        m = m.cuda().train()
        m.train_dropout = False
        m.train_dtype = "bf16"
        params = [p for p in m.parameters() if p.requires_grad]
        losses, grads, fwd = [], [], []
        for it in range(3):
            outs = m(x, margin=margin)
            loss = TR.criterion(outs, g, wt)
            m.zero_grad(set_to_none=True)
            loss.backward()
            losses.append(float(loss))
            fwd.append([o.detach().clone() for o in outs])
            grads.append(torch.cat([p.grad.flatten() for p in params if p.grad is not None]).double())
            with torch.no_grad():
                for p in params:
                    p.mul_(1.1)
        torch.cuda.synchronize()
        runs[name] = (losses, grads, fwd, len(m.__dict__.get("_otp_pack_cache") or ()))
    (la, ga, fa, na), (lb, gb, fb, nb) = runs["plain"], runs["scheduled"]
    assert na == 0 and nb > 0                      # the scheduled run really went through the cache
    assert abs(lb[0] - lb[2]) > 1e-2 * abs(lb[0])  # the rescale is visible: a stale operator would be, too
    for oa, ob in zip(fa, fb):
        for a, b in zip(oa, ob):
            assert torch.equal(a, b)
    for a, b in zip(la, lb):
        assert abs(a - b) <= 2e-6 * abs(a), (la, lb)
    for a, b in zip(ga, gb):
        assert float((a - b).norm() / a.norm()) <= 1e-2, float((a - b).norm() / a.norm())
