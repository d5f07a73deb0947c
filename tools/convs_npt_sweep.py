"""Development aid (GPU box): the S8 conv (csrc/convs.hip) of the four HRNet branch shapes at 128- and 256-pixel workgroup
tiles (OTPOSE_S8_NPT=2 / 4 / unset, one process each), conv1 form (S8 -> S8) and conv2 form (S8 + C4 residual -> C4 + S8)."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from otpose_amd import ops  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 80
for c, h, w in ((48, 96, 72), (96, 48, 36), (192, 24, 18), (384, 12, 9)):
    x = torch.randn(n, c, h, w, device="cuda")
    ws = ops.pack_s8_weight(torch.randn(c, c, 3, 3, device="cuda") * 0.05)
    xs, ys = ops.s8_pack(x), ops.s8_empty(n, c, h, w, "cuda")
    d = ops.s8_conv_desc(n, c, c, h, w, ops.ACT_RELU)
    rc4, oc4 = ops.c4_empty(n, c, h, w, "cuda"), ops.c4_empty(n, c, h, w, "cuda")
    ops.s8_pack(torch.randn(n, c, h, w, device="cuda"), out_c4=rc4)
    line = "%3d ch @%dx%d x%d:" % (c, h, w, n)
    for npt in (os.environ.get("OTPOSE_S8_NPT", ""),):       # (read once per process by the library: run once per setting)
        for form in (0, 1):
            f = (lambda: ops.conv3x3_s8_launch(xs, ws, None, d, rc4, oc4, ops.S8_F32_C4, ys)) if form else \
                (lambda: ops.conv3x3_s8_launch(xs, ws, None, d, None, None, ops.S8_F32_C4, ys))
            for _ in range(3):
                f()
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(20):
                f()
            b.record()
            torch.cuda.synchronize()
            line += "  %s %s %.1f us" % ("NPT " + npt if npt else "auto ", "conv2" if form else "conv1", a.elapsed_time(b) / 20 * 1e3)
    print(line)
