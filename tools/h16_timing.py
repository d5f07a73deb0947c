#!/usr/bin/env python
"""Phase stamps of h16_conv3x3_kernel from a development build of the library (-DOTP_H16_TIMING), GPU box only:
    bash tools/h16_timing.sh 80 48 48 96 72 [stride] [res]
prints, per phase, the median / p90 over workgroups in shader cycles and the workgroup lifetime."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from otpose_amd import hip, ops                 # noqa: E402

n, cin, cout, h, w = (int(a) for a in sys.argv[1:6])
stride = int(sys.argv[6]) if len(sys.argv) > 6 and sys.argv[6].isdigit() else 1
with_res = "res" in sys.argv
raw = ctypes.CDLL(hip.LIB_PATH)
x = ops.h8_pack(torch.randn(n, cin, h, w, device="cuda"))
wt = torch.randn(cout, cin, 3, 3, device="cuda") * 0.05
wp = ops.pack_h16_conv_weight(wt, None, 0)
sh = torch.zeros(cout, device="cuda")
res = ops.h8_pack(torch.randn(n, cout, h // stride, w // stride, device="cuda")) if with_res else None
out = ops.h8_empty(n, cout, h // stride, w // stride, "cuda")
run = lambda: ops.h16_conv3x3(x, wp, sh, cout, stride, ops.ACT_RELU, res, out=out)   # noqa: E731
for _ in range(3):
    run()
torch.cuda.synchronize()
buf = np.zeros(8192 * 32, dtype=np.uint64)
raw.otp_h16_read_stamps.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
assert raw.otp_h16_read_stamps(buf.ctypes.data, buf.nbytes) == 0
t = buf.reshape(8192, 32).astype(np.int64)
t = t[t[:, 0] > 0]
end = t[:, 19]
rt = (t[:, 31] - t[:, 30]) / 100.0
print(f"{len(t)} workgroups; lifetime median {np.median(end - t[:, 0]):.0f} p90 {np.percentile(end - t[:, 0], 90):.0f} cycles; "
      f"kernel span {end.max() - t[:, 0].min()} cycles; starts spread over {t[:, 0].max() - t[:, 0].min()}")
print(f"   real time per workgroup {np.median(rt):.2f} us -> {np.median((end - t[:, 0]) / np.maximum(rt, 1e-3)):.0f} cycles/us")
rows = [(0, 1, "tile index math + issue window DMA + first weight loads"), (1, 2, "fragment addresses, accumulators")]
for c in range(min(3, cin // 16)):
    b = 3 + 4 * c
    rows += [((2 if c == 0 else b - 2), b, f"chunk {c}: wait (window / previous chunk's readers)"), (b, b + 1, f"chunk {c}: weights -> LDS, barrier"),
             (b + 1, b + 2, f"chunk {c}: MFMA phase")]
last = 3 + 4 * (min(3, cin // 16) - 1) + 2
rows += [(last, 16, "remaining chunks / stages"), (16, 17, "residual loads"), (17, 19, "epilogue arithmetic + stores")]
for a, b, nm in rows:
    dt = t[:, b] - t[:, a]
    print("%-58s median %7d  p90 %7d" % (nm, np.median(dt), np.percentile(dt, 90)))
