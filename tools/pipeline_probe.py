#!/usr/bin/env python
"""Experiment: two inference engines of one model replaying their hipGraphs on two streams, alternately (two whole
forwards in flight), against the single-engine loop.  Run on the GPU box."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from otpose_amd import OTPose, cfg2                 # noqa: E402
from otpose_amd import synthetic as S               # noqa: E402
from otpose_amd.engine import InferenceEngine       # noqa: E402

dev = torch.device("cuda", 0)
cfg = cfg2()
model = OTPose(cfg)
S.fill_synthetic_(model)
model = model.to(dev).eval()
x, margin = S.synthetic_clip(16, cfg.MODEL.IMAGE_SIZE)
x, margin = x.to(dev), margin.to(dev)
steps = 20
with torch.no_grad():
    engs = [InferenceEngine(model, 16, dev), InferenceEngine(model, 16, dev)]
    streams = [torch.cuda.Stream(dev), torch.cuda.Stream(dev)]
    for e, s in zip(engs, streams):                 # capture each engine's graph on its own stream
        with torch.cuda.stream(s):
            for _ in range(3):
                e.run(x, margin)
    torch.cuda.synchronize()
    ref = [o.clone() for o in engs[0].outputs]
    # single engine
    t0 = time.perf_counter()
    with torch.cuda.stream(streams[0]):
        for _ in range(steps):
            engs[0].run(x, margin)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    print("single engine: %.2f ms / forward" % ((t1 - t0) / steps * 1e3))
    t0 = time.perf_counter()
    for i in range(steps):
        with torch.cuda.stream(streams[i & 1]):
            engs[i & 1].run(x, margin)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    print("two engines, two streams: %.2f ms / forward" % ((t1 - t0) / steps * 1e3))
    for a, b in zip(ref, engs[1].outputs):
        assert torch.equal(a, b)
    print("outputs identical")
