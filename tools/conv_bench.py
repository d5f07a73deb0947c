#!/usr/bin/env python
"""Per-shape throughput of otp_conv2d on the conv shapes of the cfg2 forward.
Development tool (run on the GPU box): prints TFLOP/s against the f32-MFMA peak for every shape."""
from __future__ import annotations

import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from otpose_amd import hip, ops                       # noqa: E402

PEAK = 157.3e12
# (name, N, Cin, Cout, k, stride, dil, H, W, calls per forward)
SHAPES = [
    ("b0 48->48 3x3 96x72", 80, 48, 48, 3, 1, 1, 96, 72, 64),
    ("b1 96->96 3x3 48x36", 80, 96, 96, 3, 1, 1, 48, 36, 64),
    ("b2 192->192 3x3 24x18", 80, 192, 192, 3, 1, 1, 24, 18, 56),
    ("b3 384->384 3x3 12x9", 80, 384, 384, 3, 1, 1, 12, 9, 24),
    ("l1 64->64 3x3 96x72", 80, 64, 64, 3, 1, 1, 96, 72, 4),
    ("l1 64->256 1x1 96x72", 80, 64, 256, 1, 1, 1, 96, 72, 5),
    ("l1 256->64 1x1 96x72", 80, 256, 64, 1, 1, 1, 96, 72, 3),
    ("t1 256->48 3x3 96x72", 80, 256, 48, 3, 1, 1, 96, 72, 1),
    ("t1 256->96 3x3s2 96x72", 80, 256, 96, 3, 2, 1, 96, 72, 1),
    ("stem 3->64 s2 384x288", 80, 3, 64, 3, 2, 1, 384, 288, 1),
    ("stem 64->64 s2 192x144", 80, 64, 64, 3, 2, 1, 192, 144, 1),
    ("fd 48->96 s2 96x72", 80, 48, 96, 3, 2, 1, 96, 72, 7),
    ("fd 96->192 s2 48x36", 80, 96, 192, 3, 2, 1, 48, 36, 7),
    ("fd 48->48 s2 96x72", 80, 48, 48, 3, 2, 1, 96, 72, 8),
    ("fd 192->384 s2 24x18", 80, 192, 384, 3, 2, 1, 24, 18, 3),
    ("fu 96->48 1x1 48x36", 80, 96, 48, 1, 1, 1, 48, 36, 8),
    ("fu 192->48 1x1 24x18", 80, 192, 48, 1, 1, 1, 24, 18, 7),
    ("fu 384->48 1x1 12x9", 80, 384, 48, 1, 1, 1, 12, 9, 3),
    ("te 136->136 k1 T6912", 16, 136, 136, 1, 1, 1, 1, 6912, 48),
    ("te 136->544 k1 T6912", 16, 136, 544, 1, 1, 1, 1, 6912, 12),
    ("te 544->136 k1 T6912", 16, 544, 136, 1, 1, 1, 1, 6912, 12),
    ("fe 17->17 k1 T6912", 16, 17, 17, 1, 1, 1, 1, 6912, 24),
    ("off 32->306 3x3 d6 96x72", 16, 32, 306, 3, 1, 6, 96, 72, 5),
    ("msk 32->153 3x3 d6 96x72", 16, 32, 153, 3, 1, 6, 96, 72, 5),
    ("rsb 20->20 3x3 96x72", 16, 20, 20, 3, 1, 1, 96, 72, 10),
    ("rsb 6->6 3x3 96x72", 16, 6, 6, 3, 1, 1, 96, 72, 20),
    ("fin 408->17 1x1 96x72", 16, 408, 17, 1, 1, 1, 96, 72, 2),
]


def time_ms(fn, iters):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--only", default="")
    ap.add_argument("--dbg", type=int, default=0, help="ablation bits (needs a -DOTP_CONV_DEBUG build)")
    ap.add_argument("--tile", default="", help="force MB,PB,WM,WP")
    ap.add_argument("--prio", type=int, default=-1, help="static wave priority on (1) / off (0)")
    ap.add_argument("--shape", action="append", default=[], help="extra shape N,Cin,Cout,k,s,d,H,W (repeatable)")
    ap.add_argument("--wino", action="store_true", help="time the Winograd F(2x2,3x3) kernel on the 3x3 stride-1 shapes")
    ap.add_argument("--sweep", action="store_true", help="try every (MB,PB,WM,WP) tile through the tuning hook")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    if a.dbg:
        hip.lib()
        import ctypes
        hip._lib.otp_conv2d_debug.argtypes = [ctypes.c_int]
        hip._lib.otp_conv2d_debug(a.dbg)
    if a.prio >= 0:
        hip.lib().otp_conv2d_set_tile(-1, a.prio, 0, 0)
    if a.tile:
        hip.lib().otp_conv2d_set_tile(*[int(v) for v in a.tile.split(",")])
    g = torch.Generator().manual_seed(3)
    tot_ms = 0.0
    tot_flop = 0.0
    print("%-58s %9s %9s %7s %9s" % ("shape", "ms/call", "TFLOP/s", "frac", "ms/fwd"))
    shapes = SHAPES
    if a.shape:
        shapes = []
        for sp in a.shape:
            v = [int(t) for t in sp.split(",")]
            shapes.append(("custom " + sp, *v, 1))
    for name, n, cin, cout, k, s, d, h, w, calls in shapes:
        if a.only and a.only not in name:
            continue
        pad = d * (k // 2)
        x = torch.randn(n, cin, h, w, generator=g).to(dev)
        wt = (torch.randn(cout, cin, k, k, generator=g) * 0.05).to(dev)
        sc, sh = torch.ones(cout, device=dev), torch.zeros(cout, device=dev)
        ho = (h + 2 * pad - (d * (k - 1) + 1)) // s + 1
        wo = (w + 2 * pad - (d * (k - 1) + 1)) // s + 1
        out = torch.empty(n, cout, ho, wo, device=dev)
        iv, ov = ops.View(x), ops.View(out)
        wp = ops.pack_conv_weight(wt)
        desc = ops.conv_desc(iv, ov, cout, k, k, s, pad, d, act=ops.ACT_RELU)
        if a.wino:
            if not (k == 3 and s == 1 and d == 1 and ops.wino_supported(desc)):
                continue
            up = ops.pack_wino_weight(wt)
            ms = time_ms(lambda: ops.conv2d_wino_launch(iv, up, sc, sh, ov, desc), a.iters)
            flop = 2.0 * cin * cout * k * k * ho * wo * n
            print("%-58s %9.4f %9.2f %7.3f %9.3f" % (name + " [winograd]", ms, flop / (ms * 1e-3) / 1e12,
                                                    flop / (ms * 1e-3) / PEAK, ms * calls))
            tot_ms += ms * calls
            tot_flop += flop * calls
            continue
        ms = time_ms(lambda: ops.conv2d_launch(iv, wp, sc, sh, ov, desc), a.iters)
        import ctypes
        plan = (ctypes.c_int * 8)()
        hip.lib().otp_conv2d_last_plan(plan)
        name = name + " " + str(list(plan)[:6])
        if a.sweep:
            L = hip.lib()
            res = []
            for MB, PB in ((1, 7), (1, 9), (2, 7), (2, 9), (3, 7), (9, 3)):
                if True:
                    for WM in (1, 2, 4):
                        for WP in (1, 2, 3, 4):
                            if WM * WP > 4:
                                continue
                            L.otp_conv2d_set_tile(MB, PB, WM, WP)
                            try:
                                t = time_ms(lambda: ops.conv2d_launch(iv, wp, sc, sh, ov, desc), 5)
                                res.append((t, (MB, PB, WM, WP)))
                            except RuntimeError:
                                pass
            L.otp_conv2d_set_tile(0, 0, 0, 0)
            res.sort()
            print("   sweep best:", ", ".join("%s %.4f" % (t[1], t[0]) for t in res[:5]), " | chosen %.4f" % ms)
        flop = 2.0 * cin * cout * k * k * ho * wo * n
        tf = flop / (ms * 1e-3) / 1e12
        tot_ms += ms * calls
        tot_flop += flop * calls
        print("%-58s %9.4f %9.2f %7.3f %9.3f" % (name, ms, tf, tf * 1e12 / PEAK, ms * calls))
    print("sum over listed shapes: %.2f ms per forward, %.2f TFLOP -> %.1f TFLOP/s" %
          (tot_ms, tot_flop / 1e12, tot_flop / (tot_ms * 1e-3) / 1e12))


if __name__ == "__main__":
    main()
