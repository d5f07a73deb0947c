#!/usr/bin/env python
"""One modulated-DCN call at the cfg2 shape (16 x 17 x 96 x 72, one dilation): time and algorithmic GB/s."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from otpose_amd import hip                      # noqa: E402

BYTES = 13_630_464
dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(7)
B = 16
xd = torch.randn(B, 17, 96, 72, generator=g).to(dev)
scale = float(sys.argv[1]) if len(sys.argv) > 1 else 3.0
off = (torch.randn(B, 306, 96, 72, generator=g) * scale).to(dev)
msk = torch.randn(B, 153, 96, 72, generator=g).to(dev)
wd = (torch.randn(17, 17, 3, 3, generator=g) * 0.2).to(dev)
bd = torch.zeros(17, device=dev)
od = torch.empty(B, 17, 96, 72, device=dev)
L = hip.lib()


def dcn():
    hip.check(L.otp_mdcn_forward(hip.ptr(xd), hip.ptr(off), hip.ptr(msk), hip.ptr(wd), hip.ptr(bd), hip.ptr(od),
                                 B, 17, 96, 72, 17, 3, 3, 1, 6, 6, 1, 17, 0.2, 0.0, 0, hip.stream_of(xd)), "dcn")


dcn()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50):
    dcn()
e1.record()
e1.synchronize()
ms = e0.elapsed_time(e1) / 50
print("dcn fwd: %.1f us  %.0f GB/s  (%.1f %% of 8 TB/s)" % (ms * 1e3, BYTES * B / ms / 1e6, BYTES * B / ms / 1e6 / 80))
