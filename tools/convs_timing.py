#!/usr/bin/env python
"""Phase stamps of convs_kernel from a development build of the library (-DOTP_CONVS_TIMING), GPU box only:
    bash tools/convs_timing.sh 80 48 48 96 72 [res] [c4|nchw] [s8]
prints, per phase, the median / p90 over workgroups in shader cycles and the workgroup lifetime."""
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
buf = np.zeros(8192 * 32, dtype=np.uint64)
raw.otp_convs_read_stamps.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
assert raw.otp_convs_read_stamps(buf.ctypes.data, buf.nbytes) == 0
t = buf.reshape(8192, 32).astype(np.int64)
t = t[t[:, 0] > 0]
end = t[:, 19]
rt = (t[:, 31] - t[:, 30]) / 100.0 if t[:, 31].max() > 0 else None
print(f"{len(t)} workgroups; lifetime median {np.median(end - t[:, 0]):.0f} p90 {np.percentile(end - t[:, 0], 90):.0f} cycles; "
      f"kernel span {end.max() - t[:, 0].min()} cycles; starts spread over {t[:, 0].max() - t[:, 0].min()}")
if rt is not None:
    print(f"   real time per workgroup {np.median(rt):.2f} us -> {np.median((end - t[:, 0]) / np.maximum(rt, 1e-3)):.0f} cycles/us")
rows = [(0, 1, "tile index math + issue chunk 0 DMA"), (1, 2, "fragment addresses, accumulators")]
for c in range(min(3, cin // 16)):
    b = 3 + 4 * c
    rows += [((2 if c == 0 else b - 1), b, f"chunk {c}: wait vmcnt(0)"), (b, b + 1, f"chunk {c}: barrier"),
             (b + 1, b + 2, f"chunk {c}: MFMA phase"), (b + 2, b + 3, f"chunk {c}: barrier + issue next DMA")]
last = 3 + 4 * (min(3, cin // 16) - 1) + 3
rows += [(last, 16, "remaining chunks"), (16, 17, "ReLU + fp32 stores"), (17, 19, "swap + split + S8 stores")]
for a, b, nm in rows:
    dt = t[:, b] - t[:, a]
    print("%-46s median %7d  p90 %7d" % (nm, np.median(dt), np.percentile(dt, 90)))
