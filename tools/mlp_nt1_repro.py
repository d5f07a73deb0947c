"""Development aid (GPU box): is the ln2 + MLP launch bit-stable over repeated launches on the same inputs?  With
OTP_MLP_NT1=1 (one token tile per wave: two 78 KB workgroups per CU) the batch-16 forward is not replay-deterministic; this runs
the launch alone.  usage: [OTP_MLP_NT1=1] python tools/mlp_nt1_repro.py [launches=200]"""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from otpose_amd import ops  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
B, C, HID, T = 16, 136, 544, 6912
g = torch.Generator().manual_seed(3)
xm = torch.randn(B, C, T, generator=g).cuda()
w1, w2 = (torch.randn(HID, C, 1, generator=g) / C ** 0.5).cuda(), (torch.randn(C, HID, 1, generator=g) / HID ** 0.5).cuda()
b1, one, zero = torch.randn(HID, generator=g).cuda() * 0.1, torch.ones(C).cuda(), torch.zeros(C).cuda()
px = ops.pack_mlp_x3_weights(w1, b1, w2)
ref = ops.ln_mlp_x3(xm, one, zero, 1e-5, px, one, zero).clone()
out = torch.empty_like(xm)
bad = 0
s2 = torch.cuda.Stream()
y = torch.randn(B, C, T, device="cuda")
for i in range(n):
    out.zero_()
    if "--neighbour" in sys.argv:                       # a second copy of the same launch on another stream
        with torch.cuda.stream(s2):
            ops.ln_mlp_x3(y, one, zero, 1e-5, px, one, zero)
    ops.ln_mlp_x3(xm, one, zero, 1e-5, px, one, zero, out=out)
    torch.cuda.synchronize()
    if not torch.equal(out, ref):
        bad += 1
        d = (out != ref).nonzero()
        print("launch %d: %d elements differ, max |d| %.3e; clips %s channels %s tokens %d..%d" % (
            i, len(d), float((out - ref).abs().max()), d[:, 0].unique().tolist(), d[:, 1].unique()[:12].tolist(),
            int(d[:, 2].min()), int(d[:, 2].max())))
print("%d of %d launches differ (OTP_MLP_NT1=%s)" % (bad, n, os.environ.get("OTP_MLP_NT1", "unset")))
