"""A few launches of each temporal-encoder kernel at cfg2 size with split products (for rocprofv3 --pmc passes):
ln2 + MLP (mlpx), q / k / v front end (qkvx_front), projection + residual (densex_cc), channel attention (scores, softmax, PV)."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from otpose_amd import ops                      # noqa: E402

B, C, HID, T = 16, 136, 544, 6912
g = torch.Generator().manual_seed(0)
r = lambda *s: torch.randn(*s, generator=g).cuda()          # noqa: E731
x, res = r(B, C, T), r(B, C, T)
one, zero = torch.ones(C, device="cuda"), torch.zeros(C, device="cuda")
packed = ops.pack_mlp_x3_weights(r(HID, C, 1) / C ** 0.5, r(HID), r(C, HID, 1) / HID ** 0.5)
out = torch.empty_like(x)
ws = [r(C, C, 1) / C ** 0.5 for _ in range(3)]
table = ops.pack_qkv_table(*[r(C, 1, 3) * 0.6 for _ in range(3)], one, zero, one, zero, one, zero)
packs = [ops.pack_dense_cc(w, None, r(C), x3=True) for w in ws]
outs = [torch.empty_like(x) for _ in range(3)]
pk = ops.pack_dense_cc(ws[0], one, zero, x3=True)
for _ in range(4):
    ops.ln_mlp_x3(x, one, zero, 1e-5, packed, one, zero, out=out)
    ops.qkv_front(x, table, packs, 1e-5, outs=outs, x3=True)
    ops.dense_cc([x], [pk], [res], x3=True)
    ops.chan_attn(outs[0], outs[1], outs[2], 2, 68 ** -0.5)
torch.cuda.synchronize()
