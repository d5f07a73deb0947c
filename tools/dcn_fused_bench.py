"""Warping head at the bench shape (16 clips, 96x72, five dilations): fused launch vs the 15 launches it replaces."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from otpose_amd import ops  # noqa: E402

B, J, H, W, dils = 16, 17, 96, 72, (3, 6, 9, 12, 15)
g = torch.Generator().manual_seed(1)
trans, x = torch.randn(B, 32, H, W, generator=g).cuda(), torch.randn(B, J, H, W, generator=g).cuda()
w_off = [(torch.randn(18 * J, 32, 3, 3, generator=g) / 17.0).cuda() for _ in dils]
w_msk = [(torch.randn(9 * J, 32, 3, 3, generator=g) / 17.0).cuda() for _ in dils]
w_dcn = [(torch.randn(J, J, 3, 3, generator=g) * 0.2).cuda() for _ in dils]
bias = [torch.randn(J, generator=g).cuda() for _ in dils]
packed = ops.pack_dcn_fused(w_off, w_msk, w_dcn, bias)
out = torch.empty_like(x)
from otpose_amd import hip  # noqa: E402
ws = torch.empty(hip.lib().otp_dcn_fused_workspace(B, H, W) // 4, dtype=torch.int32, device="cuda")


def timed(fn, reps=10):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


def unfused():
    acc = None
    for i, d in enumerate(dils):
        off = ops.conv2d(trans, w_off[i], None, None, 1, d, d)
        msk = ops.conv2d(trans, w_msk[i], None, None, 1, d, d)
        y = ops.modulated_deform_conv(x, off, msk, w_dcn[i], bias[i], 1, d, d, 1, J)
        acc = y if acc is None else acc + y
    return acc / len(dils)


tf = timed(lambda: ops.dcn_fused(trans, x, packed, dils, 0.2, out=out, workspace=ws))
ref = unfused()
tu = timed(unfused, 5)
print("fused %.1f us   unfused (10 convs + 5 DCN + adds) %.1f us   max |diff| %.3e of range %.2f"
      % (tf, tu, float((out - ref).abs().max()), float(ref.abs().max())))
