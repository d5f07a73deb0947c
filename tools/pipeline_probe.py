#!/usr/bin/env python
"""Experiment: consecutive batch-16 forwards on TWO engines (own buffers, own hipGraph, own stream) launched alternately without
joining in between, so that the low-occupancy tail of forward k runs beside the backbone of forward k + 1 and the graph-boundary
bubble of one stream is covered by the other.  usage (GPU box): python tools/pipeline_probe.py [steps=20]"""
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
B = 16
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
x, margin = S.synthetic_clip(B, cfg.MODEL.IMAGE_SIZE)
x, margin = x.to(dev), margin.to(dev).float()
with torch.no_grad():
    engines = [InferenceEngine(m, B, dev, stream_set=i) for i in range(2)]
    streams = [torch.cuda.Stream(dev) for _ in range(2)]
    for e, s in zip(engines, streams):
        with torch.cuda.stream(s):
            e.run(x, margin, alias_outputs=True)
            e.run(x, margin, alias_outputs=True)
    torch.cuda.synchronize()
    ref = [o.clone() for o in engines[0].outputs]
    for i, o in enumerate(engines[1].outputs):
        assert torch.equal(o, ref[i]), "engine 1 differs from engine 0 in output %d" % i
    xb = [(e.inp, e.margin) for e in engines]           # the engines' own input buffers: no copy inside run()
    for e in engines:
        e.inp.copy_(x), e.margin.copy_(margin)
    torch.cuda.synchronize()

    def timed(fn):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps * 1e3

    def serial():
        with torch.cuda.stream(streams[0]):
            for _ in range(steps):
                engines[0].run(*xb[0], alias_outputs=True)

    def pipelined():
        for k in range(steps):
            with torch.cuda.stream(streams[k & 1]):
                engines[k & 1].run(*xb[k & 1], alias_outputs=True)

    a = timed(serial)
    b = timed(pipelined)
    a2 = timed(serial)
    b2 = timed(pipelined)
    print("one engine, back to back: %.2f / %.2f ms per forward;  two engines alternating on two streams: %.2f / %.2f ms" % (a, a2, b, b2))
    for e in engines:
        for i, o in enumerate(e.outputs):
            assert torch.equal(o, ref[i]), "output %d changed under pipelining" % i
    print("outputs of both engines bit-identical to the serial forward")
