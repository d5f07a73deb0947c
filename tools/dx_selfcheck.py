"""Development aid (GPU box): repeated launches of the C -> C projection (csrc/densex.hip) alone; prints how the outputs of
launches 2..n differ from launch 1 (pattern of channels / tokens)."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from otpose_amd import ops  # noqa: E402
B, C, T = 16, 136, 6912
g = torch.Generator().manual_seed(3)
rnd = lambda *s: torch.randn(*s, generator=g).cuda()   # noqa: E731
xa, res = rnd(B, C, T), rnd(B, C, T)
w = rnd(C, C, 1) / C ** 0.5
sc, sh = rnd(C), rnd(C)
pk = ops.pack_dense_cc(w, sc, sh, x3=True)
torch.cuda.synchronize()
ref64 = (torch.nn.functional.conv1d(xa.double(), w.double()) * sc.double()[None, :, None] + sh.double()[None, :, None] + res.double())
outs = []
for i in range(6):
    o = torch.empty_like(xa)
    ops.dense_cc([xa], [pk], [res], [o], x3=True)
    torch.cuda.synchronize()
    outs.append(o)
    err = float((o.double() - ref64).abs().max())
    d = (o != outs[0]).nonzero()
    msg = ""
    if len(d):
        ch, tk = d[:, 1].unique(), d[:, 2].unique()
        msg = "; vs launch 0: %d elements differ, channels %s, tokens %d..%d (%d distinct), clips %s" % (
            len(d), ch[:16].tolist(), int(tk.min()), int(tk.max()), len(tk), d[:, 0].unique()[:8].tolist())
    print("launch %d: max |err| vs fp64 %.3e%s" % (i, err, msg))
