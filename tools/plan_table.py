#!/usr/bin/env python
"""Print the tile plan otp_conv2d picks for the cfg2 conv shapes (host arithmetic only: runs without a GPU)."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from otpose_amd import hip                      # noqa: E402
from tools.conv_bench import SHAPES             # noqa: E402

L = hip.lib()
for name, n, cin, cout, k, s, d, h, w, calls in SHAPES:
    pad = d * (k // 2)
    desc = hip.ConvDesc()
    desc.N, desc.Cin, desc.H, desc.W, desc.Cout, desc.kh, desc.kw = n, cin, h, w, cout, k, k
    desc.stride, desc.pad, desc.dil = s, pad, d
    desc.in_ctot, desc.out_ctot = cin, cout
    desc.Ho = (h + 2 * pad - (d * (k - 1) + 1)) // s + 1
    desc.Wo = (w + 2 * pad - (d * (k - 1) + 1)) // s + 1
    desc.act = 1
    out = (ctypes.c_int * 8)()
    L.otp_conv2d_plan(ctypes.byref(desc), out)
    print("%-28s %s" % (name, list(out)))
