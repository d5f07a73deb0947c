"""Development aid (GPU box): HRNet's first conv (3 -> 64, 3x3 stride 2, + BN + ReLU on the 80 frames of a batch-16 clip) on csrc/stem.hip
and on the direct kernel (otp_conv2d with frame_split), time and output rate."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from otpose_amd import ops, hip  # noqa: E402

B, F, H, W, C = 16, 5, 288, 384, 64
clip = torch.randn(B, 3 * F, H, W, device="cuda")
wt = torch.randn(C, 3, 3, 3, device="cuda") * 0.3
sc, sh = torch.rand(C, device="cuda") + 0.5, torch.randn(C, device="cuda")
pk = ops.pack_stem_conv_x3(wt, sc, sh)
o1 = torch.empty(F * B, C, H // 2, W // 2, device="cuda")
o2 = torch.empty_like(o1)
L = hip.lib()
d = ops.conv_desc(ops.View(clip), ops.View(o2), C, 3, 3, 2, 1, 1, ops.ACT_RELU, None, None, 1, B, 3)
wp = ops.pack_conv_weight(wt)
f1 = lambda: ops.stem_conv_x3(clip, pk, C, F, out=o1)                                                                        # noqa: E731
f2 = lambda: hip.check(L.otp_conv2d(hip.ptr(clip), None, hip.ptr(wp), hip.ptr(sc), hip.ptr(sh), None, hip.ptr(o2), d,       # noqa: E731
                                    hip.stream_of(clip)), "conv2d")
for name, f in (("stem.hip", f1), ("direct kernel", f2)):
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
    print("%-14s %.1f us, output %.2f TB/s" % (name, t, o1.numel() * 4 / t / 1e6))
print("max |diff| %.2e of %.2f" % (float((o1 - o2).abs().max()), float(o2.abs().max())))
