"""One shape of the split-bf16 conv, a few launches: the target of PMC passes (rocprofv3 --pmc ... -- python3 tools/convx_one.py)."""
import sys
import torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from otpose_amd import ops  # noqa: E402

n, ci, co, h, w = (int(a) for a in sys.argv[1:6])
st = int(sys.argv[6]) if len(sys.argv) > 6 else 1
with_res = len(sys.argv) > 7 and sys.argv[7] == "res"      # + residual input, as the HRNet BasicBlock's second conv
x = torch.randn(n, ci, h, w, device="cuda")
wt = torch.randn(co, ci, 3, 3, device="cuda") * 0.05
wp = ops.pack_x3_weight(wt, None, st)
ho, wo = (h - 1) // st + 1, (w - 1) // st + 1
y = torch.empty(n, co, ho, wo, device="cuda")
iv, ov = ops.View(x), ops.View(y)
rv = ops.View(torch.randn_like(y)) if with_res else None
d = ops.conv_desc(iv, ov, co, 3, 3, st, 1, 1, ops.ACT_RELU, None, rv)
for _ in range(5):
    ops.conv2d_x3_launch(iv, wp, None, ov, d, rv)
torch.cuda.synchronize()
