#!/usr/bin/env python
"""Probe: one batch-16 graph vs two batch-8 graphs replayed concurrently on two streams (development tool)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from otpose_amd import OTPose, cfg2                    # noqa: E402
from otpose_amd import synthetic as S                  # noqa: E402
from otpose_amd.engine import InferenceEngine          # noqa: E402

dev = torch.device("cuda", 0)
cfg = cfg2()
m = OTPose(cfg)
S.fill_synthetic_(m)
m = m.to(dev).eval()
x, margin = S.synthetic_clip(16, cfg.MODEL.IMAGE_SIZE)
x, margin = x.to(dev), margin.to(dev)
parts = int(sys.argv[1]) if len(sys.argv) > 1 else 2
n = 16 // parts
with torch.no_grad():
    engines = [InferenceEngine(m, n, dev) for _ in range(parts)]
    streams = [torch.cuda.Stream(dev) for _ in range(parts)]
    for i, e in enumerate(engines):                      # capture each graph on its own
        e.run(x[i * n:(i + 1) * n], margin[i * n:(i + 1) * n])
    torch.cuda.synchronize()

    def step():
        cur = torch.cuda.current_stream(dev)
        for i, (e, s) in enumerate(zip(engines, streams)):
            s.wait_stream(cur)
            with torch.cuda.stream(s):
                e.run(x[i * n:(i + 1) * n], margin[i * n:(i + 1) * n])
        for s in streams:
            cur.wait_stream(s)
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10
print("%d x batch %d concurrently: %.2f ms per 80 frames -> %.1f frames/s" % (parts, n, dt * 1e3, 80 / dt))
