#!/usr/bin/env python
"""A few back-to-back launches of the fused MLP at the cfg2 shape (for rocprofv3 --pmc passes)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from otpose_amd import ops                      # noqa: E402

B, C, HID, T = 16, 136, 544, 6912
g = torch.Generator().manual_seed(0)
x, res = torch.randn(B, C, T, generator=g).cuda(), torch.randn(B, C, T, generator=g).cuda()
w1, w2 = (torch.randn(HID, C, 1, generator=g) / C ** 0.5).cuda(), (torch.randn(C, HID, 1, generator=g) / HID ** 0.5).cuda()
b1, b2, sc = torch.randn(HID, generator=g).cuda(), torch.randn(C, generator=g).cuda(), torch.randn(C, generator=g).cuda()
packed = ops.pack_mlp_weights(w1, b1, w2)
out = torch.empty_like(x)
one, zero = torch.ones(C, device="cuda"), torch.zeros(C, device="cuda")
for _ in range(4):
    ops.ln_mlp_fused(x, one, zero, 1e-5, packed, sc, (b2 * sc).contiguous(), out=out)
torch.cuda.synchronize()
