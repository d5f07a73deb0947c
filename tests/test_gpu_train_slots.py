"""The optimisation step of the reference loop (script/Common.py:118-144: forward, two ST_OHKW terms, backward,
clip_grad_norm_, AdamW) with the backward kernels writing parameter gradients straight into FusedAdamW's flat buffer
(``train_ops.grad_slot``), against the SAME HIP graph with ordinary per-parameter gradients fed to torch.optim.AdamW +
clip_grad_norm_.  Covers the accumulate-vs-overwrite and aliasing hazards of the direct-to-slot path: optimizer.zero_grad(),
``model.zero_grad()`` (set_to_none), two forwards summed into one backward, f32 and bf16 graphs.  (The RCCL exchange through
a one-rank process group: tests/test_gpu_zz_rccl_world1.py.)"""
import copy
import os

import pytest
import torch

from otpose_amd import OTPose, tiny_cfg
from otpose_amd import parallel as PAR
from otpose_amd import synthetic as S
from otpose_amd import train as TR
from otpose_amd.optim import FusedAdamW
from tests.conftest import seeded

pytestmark = pytest.mark.gpu

LR, WD, CLIP = 1e-3, 0.01, 1.0


def _targets(b, j, h, w, seed=11):
    g = seeded((b, j, h, w), seed).abs() * 0.2
    g[:, ::2, 3, 4] = 1.0
    g.clamp_(max=1.0)
    wt = (seeded((b, j, 1), seed + 1) > -1.0).float()
    return g.cuda(), wt.cuda()


def _pair(dtype):
    cfg = tiny_cfg(8, (64, 96))
    a = OTPose(cfg)
    S.fill_synthetic_(a)
    b = copy.deepcopy(a)
    for m in (a, b):
        m.cuda().train()
        m.train_dropout = False                           # the two replicas must see the same graph
        m.train_dtype = dtype
    return cfg, a, b


def _loss(model, x, margin, g, wt):
    return TR.criterion(TR.forward_train(model, x, margin), g, wt)


def _grad_err(pa, pb):
    """relative L2 error of the whole gradient and the worst per-tensor one (tensors with a non-negligible norm)."""
    num = den = 0.0
    worst, wname = 0.0, ""
    for n, ga, gb in pa:
        d = float((ga.double() - gb.double()).norm()) ** 2
        r = float(gb.double().norm()) ** 2
        num, den = num + d, den + r
        if r > 1e-16 and (d / r) ** 0.5 > worst:
            worst, wname = (d / r) ** 0.5, n
    return (num / max(den, 1e-300)) ** 0.5, worst, wname


# Run-to-run spread of the HIP backward itself on identical weights and inputs (tools/train_determinism.py,
# tools/train_nondet_taps.py): f32 1e-7 (float atomics in the DCN / glue gradients, fp32 reduction order); bf16 usually 1e-9,
# but about one backward in five deviates by 1e-3 .. 6e-3 in the whole-gradient L2 sense (same loss to the last bit).  Traced
# in round 3 with gradient taps on every HRNet module: dL/d(heat-maps) agrees to 3e-8 (the atomics), the first bf16 activation
# gradient behind it (final layer's input gradient) then differs in a handful of bf16 roundings (1e-7 .. 2e-5 relative), and
# every further bf16-rounded layer of the backward decorrelates a little more: 1e-4 two modules down, 1.5e-2 at stage 2 -
# rounding noise of a 60-layer bf16 chain seeded by summation order, not a wrong kernel (every tap of the fp32 graph behind
# the backbone is bit-identical).  A stale or doubled gradient - what this test is about - is an O(1) error.
GRAD_TOL = {"f32": 2e-4, "bf16": 2e-2}
# Weights after one AdamW step from identical weights: Adam normalises, so a stale / doubled gradient moves EVERY weight by
# ~lr.  f32: max |dw| <= 0.05 lr.  bf16: the rounding noise above flips the sign of a few near-zero gradient entries (those
# weights then differ by up to 2 lr), so the bound is on the mean |dw| (measured 1e-3 .. 1e-2 lr) and on the share of weights
# that moved apart by more than 0.05 lr
W_MEAN_TOL_BF16, W_SHARE_TOL_BF16 = 0.05, 0.02


def _assert_weights_close(a, b, dtype, tag):
    pb = dict(b.named_parameters())
    with torch.no_grad():
        dws = torch.cat([(p.detach() - pb[n].detach()).abs().flatten() for n, p in a.named_parameters() if p.requires_grad])
    diff, mean = float(dws.max()), float(dws.mean())
    share = float((dws > 0.05 * LR).float().mean())
    print("%s: |dw| max %.3e mean %.3e, share > 0.05 lr %.2e (lr %.0e)" % (tag, diff, mean, share, LR))
    if dtype == "f32":
        assert diff <= 0.05 * LR, (tag, diff)
    else:
        assert mean <= W_MEAN_TOL_BF16 * LR and share <= W_SHARE_TOL_BF16, (tag, mean, share)


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("zero_mode", ["optimizer", "model"])
def test_slot_gradients_and_weights_match_per_parameter_path(dtype, zero_mode):
    cfg, a, b = _pair(dtype)
    x, margin = S.synthetic_clip(2, cfg.MODEL.IMAGE_SIZE)
    x, margin = x.cuda(), margin.cuda()
    J, (w, h) = cfg.MODEL.NUM_JOINTS, cfg.MODEL.HEATMAP_SIZE
    opt_a = FusedAdamW([p for p in a.parameters() if p.requires_grad], lr=LR, weight_decay=WD, max_grad_norm=CLIP)
    opt_b = torch.optim.AdamW([p for p in b.parameters() if p.requires_grad], lr=LR, weight_decay=WD)
    assert all(hasattr(p, "_otp_grad_slot") for p in a.parameters() if p.requires_grad)
    assert not any(hasattr(p, "_otp_grad_slot") for p in b.parameters())
    names = [n for n, p in a.named_parameters() if p.requires_grad]
    for it in range(3):
        g, wt = _targets(2, J, h, w, seed=11 + 5 * it)
        if zero_mode == "optimizer":
            opt_a.zero_grad()
        else:
            a.zero_grad()                                   # set_to_none: slots keep the previous step's values
        opt_b.zero_grad()
        la = _loss(a, x, margin, g, wt)
        lb = _loss(b, x, margin, g, wt)
        assert abs(float(la) - float(lb)) <= 1e-5 * max(1.0, abs(float(lb)))
        la.backward()
        lb.backward()
        if zero_mode == "optimizer" and it == 0:
            # the direct path really ran: autograd adopted the slot views
            pa = dict(a.named_parameters())
            adopted = sum(1 for n in names if pa[n].grad is not None and pa[n].grad.data_ptr() == pa[n]._otp_grad_slot.data_ptr())
            assert adopted > 0.9 * len(names), adopted
        opt_a.flat_grads()                                  # re-homes any ordinary gradient into its slot
        pb = dict(b.named_parameters())
        triples = [(n, p.grad, pb[n].grad if pb[n].grad is not None else torch.zeros_like(p))
                   for n, p in a.named_parameters() if p.requires_grad]
        glob, worst, wname = _grad_err(triples, None)
        print("step %d %s/%s: whole-gradient rel L2 %.2e, worst tensor %.2e (%s)" % (it, dtype, zero_mode, glob, worst, wname))
        assert glob <= GRAD_TOL[dtype], (it, glob)
        torch.nn.utils.clip_grad_norm_([p for p in b.parameters() if p.requires_grad], CLIP)
        opt_a.step()
        opt_b.step()
        # one AdamW step from identical weights and (to 1e-6) identical gradients: Adam normalises, so a stale / doubled
        # gradient would move weights by ~lr; rounding of the two implementations by ~1e-3 lr
        _assert_weights_close(a, b, dtype, "step %d slot vs per-parameter" % it)
        with torch.no_grad():
            # re-align the replicas: the graph amplifies 1e-7 weight differences (ReLU / top-k decisions that flip) into
            # 1e-3 gradient differences one step later, which would mask what this test is looking for
            for n, p in b.named_parameters():
                p.copy_(dict(a.named_parameters())[n])
            for (_, ba), (_, bb) in zip(a.named_buffers(), b.named_buffers()):
                bb.copy_(ba)
    sa, sb = a.state_dict(), b.state_dict()
    k = "rough_pose_estimation_net.bn1.running_var"
    assert float((sa[k] - sb[k]).abs().max()) <= 1e-4 * float(sb[k].abs().max())


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_two_forwards_one_backward(dtype):
    """One parameter feeding two autograd nodes of the same backward: the slot goes to one of them only."""
    cfg, a, b = _pair(dtype)
    x, margin = S.synthetic_clip(2, cfg.MODEL.IMAGE_SIZE)
    x, margin = x.cuda(), margin.cuda()
    x2 = torch.roll(x, 1, 0) * 0.9
    J, (w, h) = cfg.MODEL.NUM_JOINTS, cfg.MODEL.HEATMAP_SIZE
    g, wt = _targets(2, J, h, w)
    opt_a = FusedAdamW([p for p in a.parameters() if p.requires_grad], lr=LR, weight_decay=WD, max_grad_norm=CLIP)
    opt_a.zero_grad()
    (_loss(a, x, margin, g, wt) + _loss(a, x2, margin, g, wt)).backward()
    (_loss(b, x, margin, g, wt) + _loss(b, x2, margin, g, wt)).backward()
    opt_a.flat_grads()
    pb = dict(b.named_parameters())
    triples = [(n, p.grad, pb[n].grad if pb[n].grad is not None else torch.zeros_like(p))
               for n, p in a.named_parameters() if p.requires_grad]
    glob, worst, wname = _grad_err(triples, None)
    print("two forwards %s: whole-gradient rel L2 %.2e, worst tensor %.2e (%s)" % (dtype, glob, worst, wname))
    assert glob <= GRAD_TOL[dtype], glob


def test_pack_cache_retires_entries_of_weights_that_moved():
    """bf16_ops.PackCache (one weight re-layout launch per step): building FusedAdamW after a first forward re-homes every
    ``p.data`` into the flat buffer, so the next forward records new jobs - the old ones (which pin a full copy of the old fp32
    weights and would be re-packed every step) must be gone one forward later, and the step must still match."""
    cfg, a, b = _pair("bf16")
    x, margin = S.synthetic_clip(2, cfg.MODEL.IMAGE_SIZE)
    x, margin = x.cuda(), margin.cuda()
    J, (w, h) = cfg.MODEL.NUM_JOINTS, cfg.MODEL.HEATMAP_SIZE
    g, wt = _targets(2, J, h, w)
    _loss(a, x, margin, g, wt).backward()
    cache = a.__dict__["_otp_pack_cache"]
    n0 = len(cache)
    assert n0 > 100
    for p in a.parameters():
        p.grad = None
    opt = FusedAdamW([p for p in a.parameters() if p.requires_grad], lr=LR, weight_decay=WD, max_grad_norm=CLIP)   # moves p.data
    opt.zero_grad()
    la = _loss(a, x, margin, g, wt)
    la.backward()
    assert len(cache) == 2 * n0                       # old + new storage addresses, until the next forward's repack
    lb = _loss(b, x, margin, g, wt)
    assert abs(float(la) - float(lb)) <= 1e-5 * max(1.0, abs(float(lb)))
    opt.zero_grad()
    _loss(a, x, margin, g, wt).backward()
    assert len(cache) == n0
