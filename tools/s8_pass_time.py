"""Development aid (GPU box): the two HBM-bound S8 conversion passes of the HRNet fuse layers (csrc/convs.hip: otp_s8_pack,
otp_s8_upsample_add) at cfg2 size (80 frames): time and HBM rate over their algorithmic bytes."""
import ctypes
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from otpose_amd import ops, hip  # noqa: E402

n = 80
L = hip.lib()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)


def timeit(f, iters=20):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    a.record()
    for _ in range(iters):
        f()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


for c, h, w in ((48, 96, 72), (96, 48, 36), (192, 24, 18), (384, 12, 9)):
    x = torch.randn(n, c, h, w, device="cuda")
    s8, c4 = ops.s8_empty(n, c, h, w, "cuda"), ops.c4_empty(n, c, h, w, "cuda")
    t = timeit(lambda: ops.s8_pack(x, out=s8, out_c4=c4))
    byts = 3.0 * 4 * x.numel()
    print("s8_pack %3d ch @%dx%d x%d (S8 + C4): %.1f us, %.2f TB/s" % (c, h, w, n, t, byts / t / 1e6), flush=True)
    # fuse row of this resolution: the lower-resolution branches, upsampled and added
    lows, fs = [], []
    hh, ww, f = h, w, 1
    while hh % 2 == 0 and ww % 2 == 0 and len(lows) < 3 and hh // 2 >= 12:
        hh, ww, f = hh // 2, ww // 2, f * 2
        lows.append(torch.randn(n, c, hh, ww, device="cuda"))
        fs.append(f)
    if not lows or w % 4:                 # (otp_s8_upsample_add wants rows of whole float4s)
        continue
    lp = (ctypes.c_void_p * len(lows))(*[hip.ptr(v) for v in lows])
    fp = (ctypes.c_int * len(lows))(*fs)
    # the forms the engine launches since round 4: no C4 image; the row's own term as fp32 NCHW or as its S8 image
    xs8 = ops.s8_pack(x)
    for lay, what in ((0, "NCHW"), (1, "S8")):
        src = xs8 if lay else x
        g2 = lambda: hip.check(L.otp_s8_upsample_add_ex(lp, fp, len(lows), hip.ptr(src), lay, None, hip.ptr(s8), None,   # noqa: E731
                                                       n, c, h, w, 1, c, 0, c, 0, hip.stream_of(x)), "up")
        t = timeit(g2)
        byts = 4.0 * (x.numel() * 2 + sum(v.numel() for v in lows))
        print("s8_upsample_add_ex %3d ch @%dx%d x%d, %d low terms, %s residual -> S8 only: %.1f us, %.2f TB/s"
              % (c, h, w, n, len(lows), what, t, byts / t / 1e6), flush=True)
    for nchw in (False, True):
        o = torch.empty_like(x)
        g = lambda: hip.check(L.otp_s8_upsample_add(lp, fp, len(lows), hip.ptr(x), hip.ptr(o) if nchw else None, hip.ptr(s8),   # noqa: E731
                                                   hip.ptr(c4), n, c, h, w, 1, c, 0, c, 0, hip.stream_of(x)), "up")
        t = timeit(g)
        byts = 4.0 * (x.numel() * (3 + int(nchw)) + sum(v.numel() for v in lows))
        print("s8_upsample_add %3d ch @%dx%d x%d, %d low terms, NCHW out %d: %.1f us, %.2f TB/s"
              % (c, h, w, n, len(lows), int(nchw), t, byts / t / 1e6), flush=True)
