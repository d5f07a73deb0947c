#!/usr/bin/env python
"""Stand-alone timing of the bf16 NHWC backbone kernels (csrc/nhwc.hip) at the HRNet-W48 cfg2 shapes, HIP events on the
launch stream: forward conv, input-gradient conv, weight gradient, BatchNorm passes.  Development tool (GPU box)."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from otpose_amd import bf16_ops as B        # noqa: E402
from otpose_amd import hip                  # noqa: E402

SHAPES = [  # cin, cout, k, stride, pad, h, w   (x 80 frames)
    (48, 48, 3, 1, 1, 96, 72), (96, 96, 3, 1, 1, 48, 36), (192, 192, 3, 1, 1, 24, 18), (384, 384, 3, 1, 1, 12, 9),
    (64, 64, 3, 1, 1, 96, 72), (256, 64, 1, 1, 0, 96, 72), (64, 256, 1, 1, 0, 96, 72), (3, 64, 3, 2, 1, 384, 288),
    (64, 64, 3, 2, 1, 192, 144), (48, 96, 3, 2, 1, 96, 72), (96, 48, 1, 1, 0, 48, 36), (384, 48, 1, 1, 0, 12, 9),
]


def timeit(fn, iters=10):
    fn()
    torch.cuda.synchronize()
    st = torch.cuda.current_stream()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(iters):
        fn()
    e1.record(st)
    e1.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 80
    dev = torch.device("cuda", 0)
    L = hip.lib()
    print("%-28s %9s %9s %9s   %s" % ("shape", "fwd us", "dgrad us", "wgrad us", "TFLOP/s fwd/dgrad/wgrad   plan"))
    for cin, cout, k, s, pad, h, w in SHAPES:
        x = torch.randn(n, h, w, B.cs(cin), device=dev).to(B.BF16)
        wt = torch.randn(cout, cin, k, k, device=dev) * 0.05
        ho, wo = (h + 2 * pad - k) // s + 1, (w + 2 * pad - k) // s + 1
        gy = torch.randn(n, ho, wo, B.cs(cout), device=dev).to(B.BF16)
        d = B._desc(n, h, w, cin, cout, k, k, s, pad, 1)
        wp = B._pack(wt, d, 0)
        out = torch.empty(n, ho, wo, B.cs(cout), dtype=B.BF16, device=dev)
        rows = L.otp_nhwc_conv_stats_rows(ctypes.byref(d))
        stats = torch.empty(rows, 2, B.cs(cout), device=dev)
        plan = (ctypes.c_int * 8)()
        L.otp_nhwc_conv_plan(ctypes.byref(d), plan)

        def fwd():
            hip.check(L.otp_nhwc_conv_bf16(hip.ptr(x), hip.ptr(wp), None, hip.ptr(out), hip.ptr(stats), ctypes.byref(d),
                                           hip.stream_of(x)), "conv")
        t_f = timeit(fwd)
        t_d = timeit(lambda: B.conv_dgrad(gy, wt, (h, w), s, pad, 1)) if cin > 3 else float("nan")
        t_w = timeit(lambda: B.conv_wgrad(x, gy, tuple(wt.shape), s, pad, 1))
        flop = 2.0 * cin * cout * k * k * ho * wo * n
        print("%3d->%3d k%d s%d %3dx%-3d x%-3d %9.1f %9.1f %9.1f   %6.0f %6.0f %6.0f   MB%d NB%d CK%d ch%d nM%d grid%d lds%d"
              % (cin, cout, k, s, h, w, n, t_f, t_d, t_w, flop / t_f / 1e6, flop / t_d / 1e6, flop / t_w / 1e6,
                 plan[0], plan[1], plan[2], plan[3], plan[4], plan[5], plan[6]), flush=True)
    # the TransformerBlock MLP's projections on (B, 1, T, C) sequences (16 clips x 96 x 72 tokens)
    for cin, cout in ((136, 544), (544, 136)):
        nb, t = n // 5, 6912
        x = torch.randn(nb, 1, t, B.cs(cin), device=dev).to(B.BF16)
        wt = torch.randn(cout, cin, 1, 1, device=dev) * 0.05
        gy = torch.randn(nb, 1, t, B.cs(cout), device=dev).to(B.BF16)
        bias = torch.zeros(cout, device=dev)
        t_f = timeit(lambda: B.conv_forward(x, wt, bias, 1, 0, 1, out_mode=0, want_stats=False))
        t_o = timeit(lambda: B.conv_forward(x, wt, bias, 1, 0, 1, out_mode=1))
        t_d = timeit(lambda: B.conv_dgrad(gy, wt, (1, t), 1, 0, 1))
        t_w = timeit(lambda: B.conv_wgrad(x, gy, tuple(wt.shape), 1, 0, 1))
        print("seq %3d->%3d x%d x%d: fwd bf16 %.1f us, fwd fp32 NCHW %.1f us, dgrad %.1f us, wgrad %.1f us (with per-call packs)"
              % (cin, cout, nb, t, t_f, t_o, t_d, t_w), flush=True)
    yn = torch.randn(n // 5, 136, 1, 6912, device=dev)
    print("to_nhwc 136 x 6912 x%d: %.1f us;" % (n // 5, timeit(lambda: B.to_nhwc(yn))), end=" ")
    yb = B.to_nhwc(yn)
    print("to_nchw: %.1f us" % timeit(lambda: B.to_nchw(yb, 136)))
    # BatchNorm passes at the widest map
    c = 48
    xx = torch.randn(n, 96, 72, c, device=dev).to(B.BF16)
    g = torch.ones(c, device=dev)
    vec = torch.zeros(4, c, device=dev)
    vec[1] = 1
    vec[2] = 1
    print("bn_apply 48ch 96x72 x%d: %.1f us" % (n, timeit(lambda: B.bn_apply(xx, vec[2], vec[3], xx, True))))
    print("bn_backward 48ch 96x72 x%d: %.1f us" % (n, timeit(lambda: B.bn_backward(xx, xx, xx, vec[0], vec[1], g, c, True, True))))
    c = 384
    xx = torch.randn(n, 12, 9, c, device=dev).to(B.BF16)
    g = torch.ones(c, device=dev)
    vec = torch.zeros(4, c, device=dev)
    vec[1] = 1
    print("bn_backward 384ch 12x9 x%d: %.1f us" % (n, timeit(lambda: B.bn_backward(xx, xx, xx, vec[0], vec[1], g, c, True, True))))


if __name__ == "__main__":
    main()
