#!/usr/bin/env python
"""List the conv launches of the cfg2 engine with their tile plans (run on the GPU box); flags generic-kernel convs."""
import collections
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from otpose_amd import OTPose, cfg2, hip            # noqa: E402
from otpose_amd import synthetic as S              # noqa: E402
from otpose_amd import engine as E                 # noqa: E402

descs = []
orig = E.ops.conv_desc


def spy(*a, **k):
    d = orig(*a, **k)
    descs.append((d, k.get("in2") if "in2" in k else (a[9] if len(a) > 9 else None)))
    return d


E.ops.conv_desc = spy
m = OTPose(cfg2())
S.fill_synthetic_(m)
m = m.cuda().eval()
eng = E.InferenceEngine(m, int(sys.argv[1]) if len(sys.argv) > 1 else 16, torch.device("cuda", 0), use_graph=False)
L = hip.lib()
cnt = collections.Counter()
for d, in2 in descs:
    out = (ctypes.c_int * 8)()
    L.otp_conv2d_plan(ctypes.byref(d), out)
    key = (d.N, d.Cin, d.Cout, d.kh, d.stride, d.dil, d.H, d.W, d.res_up, d.frame_split, in2 is not None, tuple(out)[:5])
    cnt[key] += 1
for k, v in sorted(cnt.items(), key=lambda kv: -kv[1]):
    flag = "GENERIC" if (k[-1][0] == 0 or k[-2]) else ""
    print(v, k, flag)
