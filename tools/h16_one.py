"""One shape of the fp16 engine's 3x3 conv (csrc/h16.hip), a few launches: the target of PMC passes
(rocprofv3 --pmc ... -- python3 tools/h16_one.py 80 48 48 96 72 [stride] [res])."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from otpose_amd import ops  # noqa: E402

n, ci, co, h, w = (int(a) for a in sys.argv[1:6])
stride = int(sys.argv[6]) if len(sys.argv) > 6 and sys.argv[6].isdigit() else 1
x = ops.h8_pack(torch.randn(n, ci, h, w, device="cuda"))
wt = torch.randn(co, ci, 3, 3, device="cuda") * 0.05
wp = ops.pack_h16_conv_weight(wt, None, 0)
sh = torch.zeros(co, device="cuda")
res = ops.h8_pack(torch.randn(n, co, h // stride, w // stride, device="cuda")) if "res" in sys.argv else None
out = ops.h8_empty(n, co, h // stride, w // stride, "cuda")
for _ in range(5):
    ops.h16_conv3x3(x, wp, sh, co, stride, ops.ACT_RELU, res, out=out)
torch.cuda.synchronize()
