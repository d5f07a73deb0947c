"""Development tool (GPU box): the 7-frame-window forward (BASELINE configs[4], batch 16) a few times, for tools/prof.sh.
usage: python tools/cfg5_forward.py [replays]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from otpose_amd import OTPose                              # noqa: E402
from otpose_amd import synthetic as S                      # noqa: E402
from otpose_amd.config import cfg5                         # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 10
c5 = cfg5()
m = OTPose(c5)
S.fill_synthetic_(m)
m = m.cuda().eval()
m.alias_outputs = True
x, margin = S.synthetic_clip(16, c5.MODEL.IMAGE_SIZE, frames=7)
x, margin = x.cuda(), margin.cuda()
with torch.no_grad():
    for _ in range(3):
        m(x, margin=margin)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(K):
        m(x, margin=margin)
    torch.cuda.synchronize()
print("cfg5 forward: %.2f ms" % (1e3 * (time.perf_counter() - t0) / K))
