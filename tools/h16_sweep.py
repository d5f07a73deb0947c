#!/usr/bin/env python
"""Window width (channels per stage) x pixel tile sweep of otp_h16_conv3x3 (development): OTPOSE_H16_CK / OTPOSE_H16_NPT are read
per launch; a width that does not divide Cin or does not fit the LDS falls back to the planner's choice."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch  # noqa: E402

from otpose_amd import ops  # noqa: E402
from h16_bench import ev  # noqa: E402


def case(n, cin, cout, h, w, stride, res):
    g = torch.Generator().manual_seed(1)
    x = ops.h8_pack(torch.randn(n, cin, h, w, generator=g).cuda())
    wt = (torch.randn(cout, cin, 3, 3, generator=g) * 0.05).cuda()
    sh = torch.zeros(cout, device="cuda")
    wp = ops.pack_h16_conv_weight(wt, None, 0)
    ho, wo = h // stride, w // stride
    r = ops.h8_pack(torch.randn(n, cout, ho, wo, generator=g).cuda()) if res else None
    out = ops.h8_empty(n, cout, ho, wo, "cuda")
    d = ops.h16_conv_desc(x, cout, stride, ops.ACT_RELU, out, r, 0)
    row = []
    for npt in (4, 2):
        for depth in (16, 32, 48, 64, 96):
            if cin % depth:
                continue
            os.environ["OTPOSE_H16_NPT"], os.environ["OTPOSE_H16_CK"] = str(npt), str(depth)
            try:
                t = ev(lambda: ops.h16_conv3x3(x, wp, sh, cout, stride, ops.ACT_RELU, r, out=out, desc=d), 10)
            except RuntimeError:
                t = float("nan")
            row.append("%d/%d:%6.1f" % (npt, depth, t))
    print(f"s{stride} {cin:3d}->{cout:3d} @{h}x{w} res={int(res)}  " + "  ".join(row), flush=True)


if __name__ == "__main__":
    print("columns: pixel tiles per wave / channels per window stage : us per launch")
    for c, h, w in ((48, 96, 72), (96, 48, 36), (192, 24, 18), (384, 12, 9), (64, 96, 72)):
        case(80, c, c, h, w, 1, True)
    case(80, 48, 96, 96, 72, 2, False)
    case(80, 64, 64, 192, 144, 2, False)
    case(80, 256, 48, 96, 72, 1, False)
    os.environ.pop("OTPOSE_H16_NPT"), os.environ.pop("OTPOSE_H16_CK")
    print("planner's choice:")
    import h16_bench
    for c, h, w in ((48, 96, 72), (96, 48, 36), (192, 24, 18), (384, 12, 9), (64, 96, 72)):
        h16_bench.conv_case(80, c, c, h, w, 1, False)
        h16_bench.conv_case(80, c, c, h, w, 1, True)
