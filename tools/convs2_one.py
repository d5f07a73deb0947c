"""One shape of the stride-2 S8 conv (csrc/convs2.hip), a few launches: the target of PMC passes
(rocprofv3 --pmc ... -- python3 tools/convs2_one.py 80 48 96 96 72 [s8])."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from otpose_amd import hip, ops  # noqa: E402

n, ci, co, h, w = (int(a) for a in sys.argv[1:6])
s8_out = len(sys.argv) > 6 and sys.argv[6] == "s8"
x = torch.randn(n, ci, h, w, device="cuda")
wt = torch.randn(co, ci, 3, 3, device="cuda") * 0.05
wp = ops.pack_s8_weight(wt)
xs = ops.s8_pack(x)
o = torch.empty(n, co, h // 2, w // 2, device="cuda")
o8 = ops.s8_empty(n, co, h // 2, w // 2, "cuda")
d = ops.s8_s2_conv_desc(n, ci, co, h, w, ops.ACT_RELU, None if s8_out else ops.View(o))
L = hip.lib()
for _ in range(5):
    hip.check(L.otp_conv3x3_s2_s8(hip.ptr(xs), hip.ptr(wp), None, None, None if s8_out else hip.ptr(o), hip.ptr(o8) if s8_out else None,
                                  d, hip.stream_of(x)), "otp_conv3x3_s2_s8")
torch.cuda.synchronize()
