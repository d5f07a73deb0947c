"""TransformerBlock MLP launch (ln2 + 136->544->gelu->136 + residual, T = 6912): f32-MFMA kernel vs split-bf16 kernel."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from otpose_amd import ops  # noqa: E402

B, C, HID, T = 16, 136, 544, 6912
g = torch.Generator().manual_seed(3)
xm = torch.randn(B, C, T, generator=g).cuda()
w1, w2 = (torch.randn(HID, C, 1, generator=g) / C ** 0.5).cuda(), (torch.randn(C, HID, 1, generator=g) / HID ** 0.5).cuda()
b1, one, zero = torch.randn(HID, generator=g).cuda() * 0.1, torch.ones(C).cuda(), torch.zeros(C).cuda()
pf, px = ops.pack_mlp_weights(w1, b1, w2), ops.pack_mlp_x3_weights(w1, b1, w2)
of, ox = torch.empty_like(xm), torch.empty_like(xm)


def timed(fn, reps=20):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


tf = timed(lambda: ops.ln_mlp_fused(xm, one, zero, 1e-5, pf, one, zero, out=of))
tx = timed(lambda: ops.ln_mlp_x3(xm, one, zero, 1e-5, px, one, zero, out=ox))
print("f32 %.1f us  x3 %.1f us  max |diff| %.3e (range %.2f)" % (tf, tx, float((of - ox).abs().max()), float(of.abs().max())))
