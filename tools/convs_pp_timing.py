#!/usr/bin/env python
"""Slot stamps of convs_pp_kernel (development build, -DOTP_CONVS_TIMING), GPU box only:
    OTP_PP=1 bash tools/convs_timing.sh 80 48 48 96 72 [res] [c4|nchw] [s8]
Per half (A = waves 0-3, B = waves 4-7) and per chunk slot: wait for DMA, barrier, MFMA phase, barrier, DMA issue (+ epilogue at
item ends), medians over workgroups in shader cycles."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from otpose_amd import hip, ops                 # noqa: E402

n, cin, cout, h, w = (int(a) for a in sys.argv[1:6])
flags = sys.argv[6:]
raw = ctypes.CDLL(hip.LIB_PATH)
x = torch.randn(n, cin, h, w, device="cuda")
wt = torch.randn(cout, cin, 3, 3, device="cuda") * 0.05
res = None
if "res" in flags:
    res = ops.c4_empty(n, cout, h, w, "cuda")
    ops.s8_pack(torch.randn(n, cout, h, w, device="cuda"), out_c4=res)
xs = ops.s8_pack(x)
f32 = "nchw" if "nchw" in flags else ("c4" if "c4" in flags else None)
run = lambda: ops.conv3x3_s8(xs, (n, cin, h, w), wt, None, None, ops.ACT_RELU, res, f32=f32, want_s8="s8" in flags or f32 is None)   # noqa: E731
for _ in range(3):
    run()
torch.cuda.synchronize()
buf = np.zeros(512 * 128, dtype=np.uint64)
raw.otp_convs_read_pp_stamps.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
assert raw.otp_convs_read_pp_stamps(buf.ctypes.data, buf.nbytes) == 0
t = buf.reshape(256, 2, 128).astype(np.int64)
nch = cin // 16
if os.environ.get("OTPOSE_S8_PERSISTENT", "1") == "1":
    # double-buffered form: per chunk [top -> DMA issued -> MFMA done -> (epilogue -> adopt) -> barrier passed]
    th = t[:, 0]
    th = th[th[:, 0] > 0]
    print(f"{len(th)} workgroups; setup median {np.median(th[:, 1] - th[:, 0]):.0f}")
    ev, rnd = 2, 0
    while ev + 3 * nch + 2 < 128 and th[:, ev].max() > 0:
        for c in range(nch):
            last = c == nch - 1
            nxt = ev + (5 if last else 3)
            iss, mf = th[:, ev + 1] - th[:, ev], th[:, ev + 2] - th[:, ev + 1]
            if last:
                tail = (f"epilogue {np.median(th[:, ev + 3] - th[:, ev + 2]):6.0f}  adopt+decode {np.median(th[:, ev + 4] - th[:, ev + 3]):6.0f}  "
                        f"wait+barrier {np.median(th[:, nxt] - th[:, ev + 4]):6.0f}")
            else:
                tail = f"wait+barrier {np.median(th[:, nxt] - th[:, ev + 2]):6.0f}"
            print(f"  round {rnd} chunk {c}: issue DMA {np.median(iss):6.0f}  MFMA {np.median(mf):6.0f}  {tail}")
            ev = nxt
        rnd += 1
    print(f"  total {np.median(th[:, ev] - th[:, 0]):.0f} cycles")
    sys.exit(0)
for half in (0, 1):
    th = t[:, half]
    th = th[th[:, 0] > 0]
    print(f"half {'AB'[half]}: {len(th)} workgroups; setup (decode + first DMA issue + adopt) median {np.median(th[:, 1] - th[:, 0]):.0f}")
    ev = 2
    rnd = 0
    while ev + 5 * nch + 2 < 128 and th[:, ev].max() > 0:
        for c in range(nch):
            last = c == nch - 1
            prev = th[:, ev - 1]
            wait, b1, mf, b2 = (th[:, ev] - prev, th[:, ev + 1] - th[:, ev], th[:, ev + 2] - th[:, ev + 1], th[:, ev + 3] - th[:, ev + 2])
            if last:
                iss, epi, ado = th[:, ev + 4] - th[:, ev + 3], th[:, ev + 5] - th[:, ev + 4], th[:, ev + 6] - th[:, ev + 5]
                tail = f"decode+issue {np.median(iss):6.0f}  epilogue {np.median(epi):6.0f}  adopt {np.median(ado):6.0f}"
                ev += 7
            else:
                tail = f"issue {np.median(th[:, ev + 4] - th[:, ev + 3]):6.0f}"
                ev += 5
            print(f"  round {rnd} chunk {c}: wait {np.median(wait):6.0f}  barrier {np.median(b1):6.0f}  MFMA {np.median(mf):6.0f}  "
                  f"barrier {np.median(b2):6.0f}  {tail}")
        rnd += 1
    lastev = ev - 1
    print(f"  total {np.median(th[:, lastev] - th[:, 0]):.0f} cycles")
