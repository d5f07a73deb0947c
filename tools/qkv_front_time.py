"""Time of one q / k / v front-end launch of a temporal encoder (split products), cfg2 size."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from otpose_amd import ops                       # noqa: E402
B, C, T = 16, 136, 6912
g = torch.Generator().manual_seed(1)
x = torch.randn(B, C, T, generator=g).cuda()
ws = [torch.randn(C, C, 1, generator=g).cuda() / C ** 0.5 for _ in range(3)]
bs = [torch.randn(C, generator=g).cuda() for _ in range(3)]
dws = [torch.randn(C, 1, 3, generator=g).cuda() * 0.6 for _ in range(3)]
gs = [torch.ones(C).cuda() for _ in range(3)]
be = [torch.zeros(C).cuda() for _ in range(3)]
table = ops.pack_qkv_table(dws[0], dws[1], dws[2], gs[0], be[0], gs[1], be[1], gs[2], be[2])
packs = [ops.pack_dense_cc(w, None, b, x3=True) for w, b in zip(ws, bs)]
outs = [torch.empty_like(x) for _ in range(3)]
f = lambda: ops.qkv_front(x, table, packs, 1e-5, outs=outs, x3=True)   # noqa: E731
f()
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(20):
    f()
b.record()
torch.cuda.synchronize()
print("qkv front x3: %.1f us (checksum %.6e)" % (a.elapsed_time(b) / 20 * 1e3, float(outs[0].double().abs().sum())))
# the C -> C projection (+ residual) of the same encoder, one problem
res = torch.randn(B, C, T, generator=g).cuda()
pk = ops.pack_dense_cc(ws[0], gs[0], bs[0], x3=True)
po = torch.empty_like(x)
f2 = lambda: ops.dense_cc([x], [pk], [res], [po], x3=True)   # noqa: E731
f2()
torch.cuda.synchronize()
a.record()
for _ in range(20):
    f2()
b.record()
torch.cuda.synchronize()
print("projection x3: %.1f us (checksum %.6e)" % (a.elapsed_time(b) / 20 * 1e3, float(po.double().abs().sum())))
