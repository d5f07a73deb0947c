"""Development aid (GPU box): HRNet layer1's 1x1 convs at cfg2 size (80 frames, 96 x 72) on csrc/pointx.hip and on the 1x1 mode
of csrc/convx.hip (otp_conv2d_x3), time and HBM rate of the algorithmic bytes."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from otpose_amd import ops, hip  # noqa: E402

n, h, w = 80, 96, 72
for cin, cout, res in ((256, 64, False), (64, 256, True), (128, 256, False), (64, 64, False)):
    x = torch.randn(n, cin, h, w, device="cuda")
    wt = torch.randn(cout, cin, 1, 1, device="cuda") / cin ** 0.5
    sc, sh = torch.rand(cout, device="cuda") + 0.5, torch.randn(cout, device="cuda")
    r = torch.randn(n, cout, h, w, device="cuda") if res else None
    o1, o2 = torch.empty(n, cout, h, w, device="cuda"), torch.empty(n, cout, h, w, device="cuda")
    pk = ops.pack_pointwise_x3(wt, sc, sh)
    xv, rv = ops.View(x), (ops.View(r) if res else None)
    d = ops.conv_desc(xv, ops.View(o2), cout, 1, 1, 1, 0, 1, ops.ACT_RELU, None, rv, 1, 0, None)
    xp = ops.pack_x3_weight(wt, sc, 1)
    L = hip.lib()
    f1 = lambda: ops.pointwise_x3(xv, pk, ops.View(o1), rv, True)                      # noqa: E731
    f2 = lambda: hip.check(L.otp_conv2d_x3(hip.ptr(x), hip.ptr(xp), hip.ptr(sh), hip.ptr(r) if res else None, hip.ptr(o2), d,  # noqa: E731
                                           hip.stream_of(x)), "x3")
    ts = []
    for f in (f1, f2):
        for _ in range(3):
            f()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20):
            f()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) / 20 * 1e3)
    byts = 4.0 * n * h * w * (cin + cout * (2 if res else 1))
    print("%3d -> %3d%s: pointx %.1f us (%.2f TB/s), convx 1x1 %.1f us; max |diff| %.2e of %.2f" % (
        cin, cout, " + res" if res else "", ts[0], byts / ts[0] / 1e6, ts[1], float((o1 - o2).abs().max()), float(o2.abs().max())))
