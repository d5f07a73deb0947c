#!/usr/bin/env python
"""The "reference single-GPU PyTorch forward" denominator of BASELINE.json's 5x target: the
restated eager graph (oracle/otpose_oracle.py: stock PyTorch-ROCm ops = MIOpen / rocBLAS, DCN as gather-based torch
ops) at batch 16 x 5 x 384x288, fp32, HIP-event timed.  Measurement tool only - never part of the product path."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import otpose_oracle as O          # noqa: E402
from otpose_amd import OTPose, cfg2            # noqa: E402
from otpose_amd import synthetic as S          # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
dev = torch.device("cuda", 0)
torch.backends.cudnn.benchmark = True
cfg = cfg2()
m = OTPose(cfg)
S.fill_synthetic_(m)
sd = {k: v.detach().to(dev) for k, v in m.state_dict().items()}
x, margin = S.synthetic_clip(B, cfg.MODEL.IMAGE_SIZE)
x, margin = x.to(dev), margin.to(dev)
with torch.no_grad():
    for _ in range(3):
        outs = O.otpose_forward(sd, cfg, x, margin)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    times = []
    for _ in range(10):
        e0.record()
        outs = O.otpose_forward(sd, cfg, x, margin)
        e1.record()
        e1.synchronize()
        times.append(e0.elapsed_time(e1))
times.sort()
med = times[len(times) // 2]
print("eager PyTorch-ROCm forward, batch %d: median %.1f ms -> %.1f frames/s (min %.1f ms)" %
      (B, med, 5 * B / med * 1e3, times[0]))
