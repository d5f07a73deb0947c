#!/usr/bin/env python
"""Phase stamps of nhwc_conv_kernel from a development build of the library (-DOTP_NHWC_TIMING), GPU box only:
    cd otpose_amd/csrc && hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DOTP_NHWC_TIMING -c nhwc.hip -o /tmp/nhwc_t.o &&
    hipcc --offload-arch=gfx950 -shared -o /tmp/libotp_t.so /tmp/nhwc_t.o $(ls *.o | grep -v nhwc.o) && cd ../.. &&
    OTPOSE_HIP_LIB=/tmp/libotp_t.so OTP_NHWC_NO_PIPE=1 python tools/nhwc_timing.py 48 48 96 72
prints, per phase, the median / p90 over workgroups in shader cycles, and the workgroup start / end spread."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from otpose_amd import bf16_ops as B        # noqa: E402
from otpose_amd import hip                  # noqa: E402

cin, cout, h, w = (int(a) for a in sys.argv[1:5])
WGRAD = len(sys.argv) > 5 and sys.argv[5] == "wgrad"
n = 80
dev = torch.device("cuda", 0)
L = hip.lib()
raw = ctypes.CDLL(hip.LIB_PATH)
x = torch.randn(n, h, w, B.cs(cin), device=dev).to(B.BF16)
wt = torch.randn(cout, cin, 3, 3, device=dev) * 0.05
gy = torch.randn(n, h, w, B.cs(cout), device=dev).to(B.BF16)
for _ in range(3):
    if WGRAD:
        B.conv_wgrad(x, gy, tuple(wt.shape), 1, 1, 1)
    else:
        out, stats, rows = B.conv_forward(x, wt, None, 1, 1, 1)
torch.cuda.synchronize()
d = B._desc(n, h, w, cin, cout, 3, 3, 1, 1, 1)
plan = (ctypes.c_int * 8)()
L.otp_nhwc_conv_plan(ctypes.byref(d), plan)
grid = min(plan[5], 8192)
if WGRAD:
    nco, nci = (cout + 47) // 48, (cin + 47) // 48
    grid = min(8192, nco * nci * max(1, min(512, 768 // (nco * nci))))
    names = ["prologue", "stage tile 0 (loads + LDS writes + barrier)", "k-steps of tile 0", "remaining tiles", "partial store"]
buf = np.zeros(8192 * 8, dtype=np.uint64)
raw.otp_nhwc_read_stamps.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
assert raw.otp_nhwc_read_stamps(buf.ctypes.data, buf.nbytes) == 0
t = buf.reshape(8192, 8)[:grid].astype(np.int64)
if not WGRAD:
    names = ["prologue", "stage chunk 0 (loads + LDS writes + barrier)", "MFMA loop chunk 0", "remaining chunks", "epilogue"]
print("plan MB%d NB%d CK%d chunks%d grid%d lds%d" % (plan[0], plan[1], plan[2], plan[3], plan[5], plan[6]))
for i, nm in enumerate(names):
    dt = t[:, i + 1] - t[:, i]
    print("%-48s median %7d  p90 %7d cycles" % (nm, np.median(dt), np.percentile(dt, 90)))
life = t[:, 5] - t[:, 0]
print("workgroup lifetime median %d p90 %d; kernel span (first start -> last end) %d cycles; starts spread over %d"
      % (np.median(life), np.percentile(life, 90), t[:, 5].max() - t[:, 0].min(), t[:, 0].max() - t[:, 0].min()))
