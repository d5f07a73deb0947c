"""Development aid (GPU box): a layer1 Bottleneck's 3x3 conv (64 -> 64 at 96 x 72 x 80 frames, NCHW in / NCHW out) on
csrc/convx.hip against s8_pack + csrc/convs.hip with an NCHW epilogue."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from otpose_amd import ops, hip  # noqa: E402

n, c, h, w = 80, int(sys.argv[1]) if len(sys.argv) > 1 else 64, 96, 72
co = int(sys.argv[2]) if len(sys.argv) > 2 else c
x = torch.randn(n, c, h, w, device="cuda")
wt = torch.randn(co, c, 3, 3, device="cuda") * 0.05
sc, sh = torch.rand(co, device="cuda") + 0.5, torch.randn(co, device="cuda")
o1, o2 = torch.empty(n, co, h, w, device="cuda"), torch.empty(n, co, h, w, device="cuda")
L = hip.lib()
d = ops.conv_desc(ops.View(x), ops.View(o1), co, 3, 3, 1, 1, 1, ops.ACT_RELU)
xp = ops.pack_x3_weight(wt, sc, 1)
f1 = lambda: hip.check(L.otp_conv2d_x3(hip.ptr(x), hip.ptr(xp), hip.ptr(sh), None, hip.ptr(o1), d, hip.stream_of(x)), "x3")   # noqa: E731
ws = ops.pack_s8_weight(wt, sc)
xs = ops.s8_empty(n, c, h, w, "cuda")
ds = ops.s8_conv_desc(n, c, co, h, w, ops.ACT_RELU, ops.View(o2))
assert ops.s8_conv_supported(ds)
fp = lambda: ops.s8_pack(x, out=xs)                                                                                         # noqa: E731
fc = lambda: ops.conv3x3_s8_launch(xs, ws, sh, ds, None, o2, ops.S8_F32_NCHW, None)                                          # noqa: E731


def timed(f):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20):
        f()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / 20 * 1e3


t1, tp, tc = timed(f1), timed(fp), timed(fc)
print("%d -> %d 3x3 @%dx%d x%d: convx %.1f us; s8_pack %.1f us + convs (NCHW out) %.1f us = %.1f us; max |diff| %.2e of %.2f" % (
    c, co, h, w, n, t1, tp, tc, tp + tc, float((o1 - o2).abs().max()), float(o1.abs().max())))
