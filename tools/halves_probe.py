#!/usr/bin/env python
"""Experiment: the batch-16 forward as P sub-batches inside ONE hipGraph, software-pipelined - the backbone of sub-batch
i + 1 starts when the backbone of sub-batch i has finished, so that the low-occupancy tail of i (temporal encoders, RSB heads,
warping head: 1-2 busy streams) runs beside it.  usage (GPU box): python tools/halves_probe.py [parts=2] [stagger=1]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from otpose_amd import OTPose, cfg2                    # noqa: E402
from otpose_amd import hip                             # noqa: E402
from otpose_amd import synthetic as S                  # noqa: E402
from otpose_amd.engine import InferenceEngine          # noqa: E402

dev = torch.device("cuda", 0)
cfg = cfg2()
m = OTPose(cfg)
S.fill_synthetic_(m)
m = m.to(dev).eval()
B = 16
x, margin = S.synthetic_clip(B, cfg.MODEL.IMAGE_SIZE)
x, margin = x.to(dev), margin.to(dev).float()
parts = int(sys.argv[1]) if len(sys.argv) > 1 else 2
stagger = (int(sys.argv[2]) if len(sys.argv) > 2 else 1) != 0
n = B // parts
with torch.no_grad():
    ref_eng = InferenceEngine(m, B, dev)
    ref = [o.clone() for o in ref_eng.run(x, margin)]
    t0 = time.perf_counter()
    for _ in range(10):
        ref_eng.run(x, margin, alias_outputs=True)
    torch.cuda.synchronize()
    print("one batch-%d engine: %.2f ms" % (B, (time.perf_counter() - t0) / 10 * 1e3))
    del ref_eng
    engines = [InferenceEngine(m, n, dev, use_graph=False, stream_set=i, inp=x[i * n:(i + 1) * n].contiguous(),
                               margin=margin[i * n:(i + 1) * n].contiguous()) for i in range(parts)]
    mains = [hip.side_streams(dev, 1, 4 * i + 3)[0] for i in range(parts)]

    def launch():
        cur = torch.cuda.current_stream(dev)
        prev = None
        for e, s in zip(engines, mains):
            s.wait_stream(cur)
            with torch.cuda.stream(s):
                if prev is not None and stagger:
                    s.wait_event(prev)
                e._launch_all(0, e.hr_end)
                prev = torch.cuda.Event()
                prev.record(s)
                e._launch_all(e.hr_end, None)
        for s in mains:
            cur.wait_stream(s)

    launch()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        launch()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        g.replay()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10
    print("%d x batch %d in one graph, stagger=%d: %.2f ms per %d frames -> %.1f frames/s" % (parts, n, stagger, dt * 1e3, 5 * B, 5 * B / dt))
    out = torch.cat([e.outputs[0] for e in engines])
    print("max |delta| of `output` vs the batch-%d engine: %.3e" % (B, float((out - ref[0]).abs().max())))
