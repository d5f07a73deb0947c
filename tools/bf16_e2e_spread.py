#!/usr/bin/env python
"""Spread of tests/test_gpu_train_e2e.py::test_train_step_matches_oracle_autograd[bf16]'s whole-gradient figures over input seeds and
over the two conv kernels of the bf16 backbone (OTPOSE_NHWC_HB): the fixture (batch 2, 64 x 96 frames, BatchNorm over as few as 60
values) amplifies rounding by ~1e4, so the figures say how chaotic the fixture is, not how accurate a kernel is.  Development tool."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from oracle import otpose_oracle as O  # noqa: E402
from otpose_amd import OTPose, tiny_cfg  # noqa: E402
from otpose_amd import synthetic as S  # noqa: E402
from otpose_amd import train as TR  # noqa: E402
from tests.test_gpu_train_e2e import _rel_stats, _targets  # noqa: E402


def run(seed, hb):
    os.environ["OTPOSE_NHWC_HB"] = hb
    cfg = tiny_cfg(8, (64, 96))
    model = OTPose(cfg)
    S.fill_synthetic_(model)
    sd_cpu = {k: v.detach().clone() for k, v in model.state_dict().items()}
    leaves = {k: v.double().requires_grad_() for k, v in sd_cpu.items() if v.is_floating_point() and k in dict(model.named_parameters())}
    sd_ref = {k: (v.double() if v.is_floating_point() else v) for k, v in sd_cpu.items()}
    sd_ref.update(leaves)
    x, margin = S.synthetic_clip(2, cfg.MODEL.IMAGE_SIZE, seed=seed)
    outs_ref = O.otpose_forward(sd_ref, cfg, x.double(), margin, training_bn=True)
    B, J, h, w = outs_ref[0].shape
    g, wt = _targets(B, J, h, w)
    gd, wd = g.double(), wt.double()
    l1 = O.st_ohkw_mse_loss(outs_ref[0], outs_ref[1][:B], gd, wd)["final_loss"]
    l2 = O.st_ohkw_mse_loss(outs_ref[4], outs_ref[4], (gd + outs_ref[2]) / 2, wd)["final_loss"]
    (l1 + l2).backward()
    model = model.cuda().train()
    model.train_dropout = False
    model.train_dtype = "bf16"
    outs = model(x.cuda(), margin=margin.cuda())
    TR.criterion(outs, g.cuda(), wt.cuda()).backward()
    named = [(n, p.grad) for n, p in model.named_parameters() if leaves[n].grad is not None]
    stats, glob = _rel_stats(named, leaves)
    gq = torch.cat([g_.cpu().double().flatten() for _, g_ in named])
    gr = torch.cat([leaves[n].grad.flatten() for n, _ in named])
    return glob, float(torch.dot(gq, gr) / (gq.norm() * gr.norm())), float(gq.norm() / gr.norm())


for seed in (S.INPUT_SEED, S.INPUT_SEED + 11, S.INPUT_SEED + 12, S.INPUT_SEED + 13):
    for hb in ("0", "1"):
        print("input seed %d, OTPOSE_NHWC_HB=%s: whole-gradient rel L2 %.3f, cosine %.4f, |g| / |g_ref| %.3f" % ((seed, hb) + run(seed, hb)), flush=True)
