"""Development aid (GPU box): host time of one OTPose.forward call (no synchronisation) and of its pieces - is the hipGraph
replay asynchronous, and what does the host do between two replays?"""
import os
import sys
import time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from otpose_amd import OTPose, cfg2                 # noqa: E402
from otpose_amd import synthetic as S               # noqa: E402

cfg = cfg2()
x, margin = S.synthetic_clip(16, cfg.MODEL.IMAGE_SIZE)
m = OTPose(cfg)
S.fill_synthetic_(m)
m = m.cuda().eval()
m.alias_outputs = True
x, margin = x.cuda(), margin.cuda()
with torch.no_grad():
    for _ in range(3):
        m(x, margin=margin)
    torch.cuda.synchronize()
    ts = []
    t_all = time.perf_counter()
    for _ in range(10):
        t0 = time.perf_counter()
        m(x, margin=margin)
        ts.append(time.perf_counter() - t0)
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t_all
    print("host time per forward call, no sync: %s ms; wall per forward %.2f ms" % (" ".join("%.2f" % (t * 1e3) for t in ts), t_all / 10 * 1e3))
    e = m._engine
    t0 = time.perf_counter()
    for _ in range(10):
        e.matches(x)
    print("engine.matches(): %.3f ms" % ((time.perf_counter() - t0) / 10 * 1e3))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    e.graph.replay()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("graph.replay() returns after %.2f ms, GPU done after %.2f ms" % ((t1 - t0) * 1e3, (t2 - t0) * 1e3))
    t0 = time.perf_counter()
    e.graph.replay()
    e.graph.replay()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("two replays back to back return after %.2f ms, GPU done after %.2f ms" % ((t1 - t0) * 1e3, (t2 - t0) * 1e3))
