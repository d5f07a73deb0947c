"""One shape of the S8 / LDS-DMA conv (csrc/convs.hip), a few launches: the target of PMC passes
(rocprofv3 --pmc ... -- python3 tools/convs_one.py 80 48 48 96 72 [conv2 | conv2s])."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from otpose_amd import ops  # noqa: E402

n, ci, co, h, w = (int(a) for a in sys.argv[1:6])
conv2 = len(sys.argv) > 6 and sys.argv[6] == "conv2"          # + C4 residual, C4 + S8 outputs (a BasicBlock's second conv, round 3)
conv2s = len(sys.argv) > 6 and sys.argv[6] == "conv2s"        # + S8 residual, S8 output (the same conv since round 4)
x = torch.randn(n, ci, h, w, device="cuda")
wt = torch.randn(co, ci, 3, 3, device="cuda") * 0.05
ws = ops.pack_s8_weight(wt)
xs = ops.s8_pack(x)
ys = ops.s8_empty(n, co, h, w, "cuda")
d = ops.s8_conv_desc(n, ci, co, h, w, ops.ACT_RELU)
rc4 = oc4 = None
if conv2:
    rc4, oc4 = ops.c4_empty(n, co, h, w, "cuda"), ops.c4_empty(n, co, h, w, "cuda")
    ops.s8_pack(torch.randn(n, co, h, w, device="cuda"), out_c4=rc4)
if conv2s:
    rc4 = ops.s8_pack(torch.randn(n, co, h, w, device="cuda"))
    d.res_layout = 1
for _ in range(5):
    ops.conv3x3_s8_launch(xs, ws, None, d, rc4, oc4, ops.S8_F32_C4, ys)
torch.cuda.synchronize()
