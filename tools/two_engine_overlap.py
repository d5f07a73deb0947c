"""Development tool (GPU box): what does overlapping two forwards buy?  Two engines (two captured hipGraphs of the headline
forward, separate buffers) replayed (a) back to back on one stream, (b) on two streams at once - the upper bound for a
pipelined engine that runs the HRNet of batch t + 1 beside the encoders / heads of batch t.
usage: python tools/two_engine_overlap.py [batch] [replays]"""
import sys
import time

import torch

sys.path.insert(0, '.')
from otpose_amd import OTPose, cfg2                        # noqa: E402
from otpose_amd import synthetic as S                      # noqa: E402

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 16
K = int(sys.argv[2]) if len(sys.argv) > 2 else 20
cfg = cfg2()
x, margin = S.synthetic_clip(batch, cfg.MODEL.IMAGE_SIZE)
x, margin = x.cuda(), margin.cuda()
ms = []
for _ in range(2):
    m = OTPose(cfg)
    S.fill_synthetic_(m)
    m = m.cuda().eval()
    with torch.no_grad():
        for _ in range(3):
            m(x, margin=margin)
    ms.append(m)
torch.cuda.synchronize()
g1, g2 = ms[0]._engine.graph, ms[1]._engine.graph
assert g1 is not None and g2 is not None


def timed(fn):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn()
    torch.cuda.synchronize()
    return time.perf_counter() - t0


def seq():
    for _ in range(K):
        g1.replay()
        g2.replay()


s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def conc():
    for _ in range(K):
        with torch.cuda.stream(s1):
            g1.replay()
        with torch.cuda.stream(s2):
            g2.replay()


def one():
    for _ in range(2 * K):
        g1.replay()


t1 = timed(one)
ts = timed(seq)
tc = timed(conc)
print("batch %d: one engine %.2f ms / forward; two engines back to back %.2f ms / forward; two engines on two streams %.2f ms / forward "
      "(x%.3f)" % (batch, 1e3 * t1 / (2 * K), 1e3 * ts / (2 * K), 1e3 * tc / (2 * K), ts / tc))
