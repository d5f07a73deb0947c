#!/usr/bin/env python
"""Host-side cost of enqueuing one bf16 training forward / backward (cProfile, top functions by own time): the forward is
bound by the host once the HRNet branches overlap on the device.  Development tool."""
import cProfile
import os
import pstats
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from otpose_amd import OTPose, cfg2           # noqa: E402
from otpose_amd import synthetic as S         # noqa: E402
from otpose_amd import train as TR            # noqa: E402
from otpose_amd.optim import FusedAdamW       # noqa: E402

cfg = cfg2()
model = OTPose(cfg)
S.fill_synthetic_(model)
model = model.cuda().train()
model.train_dtype = sys.argv[1] if len(sys.argv) > 1 else "bf16"
x, margin = S.synthetic_clip(16, cfg.MODEL.IMAGE_SIZE)
x, margin = x.cuda(), margin.cuda()
J = cfg.MODEL.NUM_JOINTS
w, h = cfg.MODEL.HEATMAP_SIZE
g = torch.rand(16, J, h, w, device="cuda") * 0.2
wt = (torch.rand(16, J, 1, device="cuda") > 0.15).float()
opt = FusedAdamW([p for p in model.parameters() if p.requires_grad], lr=1e-4, weight_decay=0.01, max_grad_norm=1.0)


def fwd():
    return TR.criterion(model(x, margin=margin), g, wt)


for _ in range(2):
    loss = fwd()
    opt.zero_grad()
    loss.backward()
    opt.step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
loss = fwd()
pr.disable()
torch.cuda.synchronize()
print("=== forward (host) ===")
pstats.Stats(pr).sort_stats("tottime").print_stats(28)
opt.zero_grad()
pr = cProfile.Profile()
pr.enable()
loss.backward()
pr.disable()
torch.cuda.synchronize()
print("=== backward (host, autograd thread not included) ===")
pstats.Stats(pr).sort_stats("tottime").print_stats(8)
