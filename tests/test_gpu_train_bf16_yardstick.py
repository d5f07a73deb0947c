"""A yardstick for the bf16 training path (BASELINE configs[2]) that is tight enough to notice a wrong kernel
(missing item 6).

The reference has no bf16 mode, and against an fp32 / fp64 graph the bf16 activations move the whole gradient by ~0.1
relative L2 (tests/test_gpu_train_e2e.py).  Round 4 built the oracle that "rounds at the same points"
(``oracle.bf16_points()``: values AND gradients rounded to bfloat16 exactly where otpose_amd/train.py::TrainGraphBF16 stores
bfloat16) - and measured that, end to end, it is NO tighter (first test below: heat-maps 3.2e-2 of their range against 3.4e-2
for the plain fp64 oracle).  The reason is structural: rounding is discontinuous, so the 1e-7 summation-order difference
between two fp32-accumulating implementations flips a few bf16 roundings in the first layer, every flipped element (one
bf16 ulp = 2^-8 of its value) perturbs 9 x Cout outputs of the next layer by a fraction of an ulp and flips a share of THEIR
roundings - an avalanche that saturates after a handful of layers, after which the two runs are two independent draws of the
bf16 rounding noise.  No whole-graph comparison of a 60-layer bf16 chain can beat that noise floor.

What CAN be tight is one layer at a time, in context (second test): the step runs once through the product with a tap on
every conv + BatchNorm (+ residual) (+ ReLU) layer of the backbone (292 layers at W16) recording the layer's own bf16 input,
its result, the gradient arriving at the result and the gradients it hands on; each layer is then recomputed from THOSE
operands in float64 with the product's rounding points.  Now the only difference is summation order inside one layer: a
share ~1e-4 of the elements lands on the other side of a rounding boundary (one ulp each), i.e. relative L2 ~1e-4 .. 1e-3,
two orders of magnitude below the end-to-end figures - a wrong tap offset, a stale packed weight or a dropped residual is an
O(1) error in exactly one layer and cannot hide."""
import pytest
import torch

from oracle import otpose_oracle as O
from otpose_amd import OTPose, tiny_cfg
from otpose_amd import synthetic as S
from otpose_amd import train as TR
from tests.conftest import seeded
from tests.test_gpu_train_e2e import _rel_stats

pytestmark = pytest.mark.gpu


def _backbone(cfg):
    model = OTPose(cfg)
    S.fill_synthetic_(model)
    x, _ = S.synthetic_clip(2, cfg.MODEL.IMAGE_SIZE)
    return model, x


def test_rounding_aware_oracle_is_no_tighter_end_to_end():
    """HRNet alone (model/HRNet.py:116-152, BatchNorm batch statistics) under a plain heat-map MSE, bf16 path, against the
    float64 oracle WITH the product's bf16 rounding points: still at the rounding-noise floor (module docstring), which is the
    measurement that motivates the per-layer test below.  Bounds as for the plain fp64 oracle."""
    cfg = tiny_cfg(16, (128, 192))
    model, x = _backbone(cfg)
    pre = "rough_pose_estimation_net"
    sd_cpu = {k: v.detach().clone() for k, v in model.state_dict().items()}
    names = [k for k, _ in model.named_parameters() if k.startswith(pre)]
    leaves = {k: sd_cpu[k].double().requires_grad_() for k in names}
    sd_ref = {k: (v.double() if v.is_floating_point() else v) for k, v in sd_cpu.items()}
    sd_ref.update(leaves)
    frames = torch.cat(x.split(3, dim=1), 0)
    stages = [cfg["MODEL"]["EXTRA"][f"STAGE{s}"] for s in (2, 3, 4)]
    with O.bf16_points():
        ref = O.hrnet_forward(sd_ref, pre, frames.double(), stages, training=True)
        tgt = seeded(tuple(ref.shape), 21).abs() * 0.3
        (0.5 * ((ref - tgt.double()) ** 2).mean()).backward()
    model = model.cuda().train()
    graph = TR.TrainGraphBF16(model)
    out = graph.hrnet(pre, graph.hrnet_input(x.cuda()))
    (0.5 * ((out - tgt.cuda()) ** 2).mean()).backward()
    err = float((out.detach().cpu().double() - ref.detach()).abs().max()) / float(ref.detach().abs().max())
    P = dict(model.named_parameters())
    stats, glob = _rel_stats([(n, P[n].grad) for n in names], leaves)
    print("\nbf16 backbone vs rounding-aware oracle, end to end: heat-map max err / range %.3e, whole-gradient rel L2 %.3e"
          % (err, glob))
    assert err <= 6e-2 and glob <= 1e-2


def _nchw(t, c):
    """(N, H, W, CS) bf16 -> (N, c, H, W) float64 on the CPU"""
    return t.detach().float().cpu().double().permute(0, 3, 1, 2)[:, :c].contiguous()


def _rel(a, b):
    return float((a - b).norm()) / max(float(b.norm()), 1e-30)


def test_every_bf16_backbone_layer_matches_fp64_on_its_own_operands():
    """Per-layer, in-context check of the bf16 training backbone: forward result, input / residual gradients, weight and
    BatchNorm parameter gradients of EVERY conv + BN (+ residual) (+ ReLU) layer against float64 arithmetic on the layer's own
    bf16 operands (module docstring).  Measured on MI355X (printed): see the bounds at the end."""
    cfg = tiny_cfg(16, (128, 192))
    model, x = _backbone(cfg)
    pre = "rough_pose_estimation_net"
    model = model.cuda().train()
    graph = TR.TrainGraphBF16(model)
    layers = []
    graph.taps = {"layers": layers}
    out = graph.hrnet(pre, graph.hrnet_input(x.cuda()))
    tgt = seeded(tuple(out.shape), 21).abs() * 0.3
    (0.5 * ((out - tgt.cuda()) ** 2).mean()).backward()
    torch.cuda.synchronize()
    P = {n: p for n, p in model.named_parameters()}
    assert len(layers) > 100
    worst = dict(y=0.0, ymax=0.0, gx=0.0, gres=0.0, gw=0.0, gg=0.0, gb=0.0)
    where = dict(worst)
    flips = decisions = 0
    for rec in layers:
        conv, bn = rec["conv"], rec["bn"]
        w = P[conv + ".weight"]
        cout, cin = w.shape[:2]
        sd = {"c.weight": w.detach().cpu().double().requires_grad_(), "b.weight": P[bn + ".weight"].detach().cpu().double().requires_grad_(),
              "b.bias": P[bn + ".bias"].detach().cpu().double().requires_grad_()}
        xin = _nchw(rec["x"], cin).requires_grad_()
        res = _nchw(rec["res"], cout).requires_grad_() if rec["res"] is not None else None
        yh = _nchw(rec["y"], cout)
        with O.bf16_points():
            # the layer without its ReLU, then the ReLU with the DECISIONS the kernel took (yh > 0): a pre-activation within one
            # rounding step of zero may fall on either side - the synthetic frames have flat regions, so one such value repeats over
            # many pixels (round 5: 18 of 122 880 decisions of one layer differed, all in one channel, and carried 1.8 % of the
            # gradient arriving there) - and a gradient taken through the other branch says nothing about the arithmetic.  The
            # number of differing decisions is bounded below instead.
            pre = O._conv_bn(sd, "c", "b", xin, rec["stride"], rec["pad"], False, True, res=res)
            if rec["relu"]:
                keep = (yh > 0).to(pre.dtype)
                y = pre * keep
                flips += int(((pre.detach() > 0) != (yh > 0)).sum())
                decisions += yh.numel()
            else:
                y = pre
            y.backward(_nchw(rec["gy"], cout))
            y = torch.relu(pre.detach()) if rec["relu"] else pre.detach()
        errs = dict(y=_rel(yh, y.detach()),
                    ymax=float((yh - y.detach()).abs().max()) / float(y.detach().abs().max()),
                    gw=_rel(w.grad.cpu().double(), sd["c.weight"].grad),
                    gg=_rel(P[bn + ".weight"].grad.cpu().double(), sd["b.weight"].grad),
                    gb=_rel(P[bn + ".bias"].grad.cpu().double(), sd["b.bias"].grad))
        if "gx" in rec:                                    # (the stem's input needs no gradient)
            errs["gx"] = _rel(_nchw(rec["gx"], cin), xin.grad)
        if res is not None:
            errs["gres"] = _rel(_nchw(rec["gres"], cout), res.grad)
        for k, v in errs.items():
            if v > worst[k]:
                worst[k], where[k] = v, conv
    print("\n%d conv + BN layers of the bf16 training backbone, each against float64 on its own bf16 operands - worst layer:"
          % len(layers))
    for k in ("y", "ymax", "gx", "gres", "gw", "gg", "gb"):
        print("   %-5s %.3e   (%s)" % (k, worst[k], where[k]))
    print("   ReLU decisions that differ from float64's: %d of %d" % (flips, decisions))
    assert flips <= 2e-4 * decisions
    # y / gx / gres: stored bf16 - a share of elements rounds the other way (relative L2); ymax: the largest single difference
    # as a fraction of the layer's range (one bf16 ulp of the largest value is 2^-8 = 3.9e-3); gw / gg / gb: fp32 sums
    assert worst["y"] <= TOL["y"] and worst["ymax"] <= TOL["ymax"]
    assert worst["gx"] <= TOL["gx"] and worst["gres"] <= TOL["gres"]
    assert worst["gw"] <= TOL["gw"] and worst["gg"] <= TOL["gg"] and worst["gb"] <= TOL["gb"]


# ~3-4x the worst of the 292 layers measured on MI355X (profiles/r04_bf16_layer_yardstick.txt): y 2.3e-4, ymax 5.2e-3 (1.3 ulp of
# the range), gx 4.8e-4, gres exact, gw 4.4e-4, dgamma 3.0e-4, dbeta 6.2e-5
TOL = dict(y=1e-3, ymax=1.6e-2, gx=2e-3, gres=1e-6, gw=2e-3, gg=2e-3, gb=5e-4)
