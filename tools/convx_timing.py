#!/usr/bin/env python
"""Phase stamps of convx_kernel from a development build of the library (-DOTP_CONVX_TIMING), GPU box only:
    bash tools/convx_timing.sh 80 48 48 96 72
prints, per phase, the median / p90 over workgroups in shader cycles and the workgroup lifetime."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from otpose_amd import hip, ops                 # noqa: E402

n, cin, cout, h, w = (int(a) for a in sys.argv[1:6])
ksz = int(sys.argv[6]) if len(sys.argv) > 6 else 3          # 1: pointwise conv (with a residual input, as in layer1)
raw = ctypes.CDLL(hip.LIB_PATH)
x = torch.randn(n, cin, h, w, device="cuda")
wt = torch.randn(cout, cin, ksz, ksz, device="cuda") * 0.05
res = torch.randn(n, cout, h, w, device="cuda") if ksz == 1 else None
run = lambda: ops.conv2d_x3(x, wt, None, None, ops.ACT_RELU if ksz == 1 else ops.ACT_NONE, res, ksz // 2, 1, 1)   # noqa: E731
for _ in range(3):
    y = run()
torch.cuda.synchronize()
buf = np.zeros(8192 * 16, dtype=np.uint64)
raw.otp_convx_read_stamps.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
assert raw.otp_convx_read_stamps(buf.ctypes.data, buf.nbytes) == 0
t = buf.reshape(8192, 16).astype(np.int64)
live = t[:, 0] > 0
spans = [int(t[live & (np.arange(8192) % 8 == x), 8].max() - t[live & (np.arange(8192) % 8 == x), 0].min()) for x in range(8)]
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ev0.record()
y = run()
ev1.record()
torch.cuda.synchronize()
print("per-XCD kernel span (ticks):", spans, " one call incl. pack: %.1f us" % (ev0.elapsed_time(ev1) * 1e3))
t = t[live]
names = ["prologue (index math, first loads issued, LDS zero, barrier)", "chunk 0: wait loads + split + LDS writes",
         "chunk 0: barrier", "chunk 0: issue next loads + MFMA phase", "chunk 0: barrier", "middle chunks + last store",
         "last MFMA phase (+ residual loads)", "epilogue stores"]
rt = (t[:, 10] - t[:, 9]) / 100.0          # s_memrealtime: 100 MHz
print(f"{len(t)} workgroups; lifetime {np.median(rt):.2f} us median (real time) -> {np.median((t[:, 8] - t[:, 0]) / rt):.0f} ticks/us; "
      f"kernel span {(t[:, 10].max() - t[:, 9].min()) / 100.0:.1f} us")
for i, nm in enumerate(names):
    dt = t[:, i + 1] - t[:, i]
    print("%-62s median %7d  p90 %7d" % (nm, np.median(dt), np.percentile(dt, 90)))
if t[:, 11].max() > 0:                      # finer stamps inside the prologue
    for a, b, nm in ((0, 11, "index math (items, addresses)"), (11, 12, "issue chunk 0 loads"), (12, 13, "zero the window"),
                     (13, 14, "fragment addresses, accumulators"), (14, 1, "barrier")):
        dt = t[:, b] - t[:, a]
        print("   prologue: %-50s median %7d  p90 %7d" % (nm, np.median(dt), np.percentile(dt, 90)))
life = t[:, 8] - t[:, 0]
print("workgroup lifetime median %d p90 %d; kernel span %d; starts spread over %d"
      % (np.median(life), np.percentile(life, 90), t[:, 8].max() - t[:, 0].min(), t[:, 0].max() - t[:, 0].min()))
