"""Development aid (GPU box): the 3x3 stride-2 convs of HRNet's fuse layers (model/HRNet.py:430-470) on csrc/convx.hip, isolated:
time, algorithmic TFLOP/s and HBM rate at cfg2 size (80 frames)."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from otpose_amd import ops, hip  # noqa: E402

n = 80
L = hip.lib()
for cin, cout, h, w in ((48, 96, 96, 72), (48, 48, 96, 72), (48, 192, 48, 36), (96, 192, 48, 36), (48, 384, 24, 18), (96, 384, 24, 18),
                        (192, 384, 24, 18), (256, 96, 96, 72), (64, 64, 192, 144)):
    x = torch.randn(n, cin, h, w, device="cuda")
    wt = torch.randn(cout, cin, 3, 3, device="cuda") * 0.05
    sc, sh = torch.rand(cout, device="cuda") + 0.5, torch.randn(cout, device="cuda")
    o = torch.empty(n, cout, h // 2, w // 2, device="cuda")
    d = ops.conv_desc(ops.View(x), ops.View(o), cout, 3, 3, 2, 1, 1, ops.ACT_RELU)
    if not ops.x3_supported(d):
        print("%d -> %d @%dx%d: not on convx" % (cin, cout, h, w))
        continue
    xp = ops.pack_x3_weight(wt, sc, 2)
    f = lambda: hip.check(L.otp_conv2d_x3(hip.ptr(x), hip.ptr(xp), hip.ptr(sh), None, hip.ptr(o), d, hip.stream_of(x)), "x3")   # noqa: E731
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20):
        f()
    b.record()
    torch.cuda.synchronize()
    t = a.elapsed_time(b) / 20 * 1e3
    flop = 2.0 * cin * cout * 9 * n * (h // 2) * (w // 2)
    byts = 4.0 * n * (cin * h * w + cout * (h // 2) * (w // 2))
    line = "%3d -> %3d s2 @%dx%d x%d: convx %.1f us, %.0f algorithmic TFLOP/s, %.2f TB/s" % (cin, cout, h, w, n, t, flop / t / 1e6, byts / t / 1e6)
    # the same layer on S8 records (csrc/convs2.hip): S8 -> NCHW (+ ReLU) and S8 -> S8
    if ops.s8_s2_conv_supported(ops.s8_s2_conv_desc(n, cin, cout, h, w, ops.ACT_RELU)):
        xs = ops.s8_pack(x)
        wp = ops.pack_s8_weight(wt, sc)
        o8 = ops.s8_empty(n, cout, h // 2, w // 2, "cuda")
        d1 = ops.s8_s2_conv_desc(n, cin, cout, h, w, ops.ACT_RELU, ops.View(o))
        d2 = ops.s8_s2_conv_desc(n, cin, cout, h, w, ops.ACT_RELU)
        for name, g in (("S8->NCHW", lambda: hip.check(L.otp_conv3x3_s2_s8(hip.ptr(xs), hip.ptr(wp), hip.ptr(sh), None, hip.ptr(o), None, d1, hip.stream_of(x)), "s2")),   # noqa: E731
                        ("S8->S8", lambda: hip.check(L.otp_conv3x3_s2_s8(hip.ptr(xs), hip.ptr(wp), hip.ptr(sh), None, None, hip.ptr(o8), d2, hip.stream_of(x)), "s2"))):   # noqa: E731
            for _ in range(3):
                g()
            torch.cuda.synchronize()
            a.record()
            for _ in range(20):
                g()
            b.record()
            torch.cuda.synchronize()
            t2 = a.elapsed_time(b) / 20 * 1e3
            line += " | %s %.1f us (%.0f TFLOP/s)" % (name, t2, flop / t2 / 1e6)
    print(line, flush=True)
