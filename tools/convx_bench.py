"""Split-bf16 (bf16x3) convolution kernel: time and error next to the Winograd / direct f32-MFMA kernels.
usage: python tools/convx_bench.py [--reps 20]"""
import sys
import torch
import torch.nn.functional as F

sys.path.insert(0, ".")
from otpose_amd import ops  # noqa: E402

SHAPES = [  # (N, Cin, Cout, H, W, pad, dil, stride)
    (80, 48, 48, 96, 72, 1, 1, 1), (80, 96, 96, 48, 36, 1, 1, 1), (80, 192, 192, 24, 18, 1, 1, 1), (80, 384, 384, 12, 9, 1, 1, 1),
    (80, 64, 64, 96, 72, 1, 1, 1), (80, 256, 48, 96, 72, 1, 1, 1), (16, 48, 48, 96, 72, 1, 1, 1), (3, 32, 17, 20, 12, 1, 1, 1),
    (2, 16, 40, 10, 6, 1, 1, 1), (16, 32, 48, 96, 72, 3, 3, 1),
    (80, 48, 48, 96, 72, 1, 1, 2), (80, 48, 96, 96, 72, 1, 1, 2), (80, 96, 192, 48, 36, 1, 1, 2), (80, 192, 384, 24, 18, 1, 1, 2),
    (80, 64, 64, 192, 144, 1, 1, 2), (80, 256, 96, 96, 72, 1, 1, 2), (3, 16, 24, 20, 12, 1, 1, 2),
]
POINTWISE = [(80, 64, 256, 96, 72), (80, 256, 64, 96, 72), (80, 64, 64, 96, 72), (80, 128, 256, 96, 72), (80, 96, 48, 48, 36),
             (80, 192, 48, 24, 18), (80, 192, 96, 24, 18), (80, 384, 48, 12, 9)]


def timed(fn, reps):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


def main():
    reps = int(sys.argv[sys.argv.index("--reps") + 1]) if "--reps" in sys.argv else 20
    torch.manual_seed(0)
    print(f"{'shape':34s} {'x3 us':>8s} {'wino us':>8s} {'f32 us':>8s}   {'x3 err':>9s} {'wino err':>9s} {'f32 err':>9s}  TF(x3, direct flops)")
    for n, ci, co, h, w, pad, dil, st in SHAPES:
        x = torch.randn(n, ci, h, w, device="cuda")
        wt = torch.randn(co, ci, 3, 3, device="cuda") * (2.0 / (ci * 9)) ** 0.5
        sc = torch.rand(co, device="cuda") + 0.5
        sh = torch.randn(co, device="cuda")
        res = torch.randn(n, co, (h + 2 * pad - 2 * dil - 1) // st + 1, (w + 2 * pad - 2 * dil - 1) // st + 1, device="cuda")
        ref = torch.relu(F.conv2d(x.double(), wt.double(), None, st, pad, dil) * sc.double().view(1, -1, 1, 1)
                         + sh.double().view(1, -1, 1, 1) + res.double())
        scale = float(ref.abs().max())
        y = ops.conv2d_x3(x, wt, sc, sh, ops.ACT_RELU, res, pad, dil, st)
        e3 = float((y.double() - ref).abs().max()) / scale
        wp = ops.pack_x3_weight(wt, sc, st)
        iv, ov, rv = ops.View(x), ops.View(y), ops.View(res)
        d = ops.conv_desc(iv, ov, co, 3, 3, st, pad, dil, ops.ACT_RELU, None, rv)
        t3 = timed(lambda: ops.conv2d_x3_launch(iv, wp, sh, ov, d, rv), reps)
        d0 = ops.conv_desc(iv, ov, co, 3, 3, st, pad, dil, ops.ACT_RELU, None, None)
        t30 = timed(lambda: ops.conv2d_x3_launch(iv, wp, sh, ov, d0, None), reps)
        tw = ew = float("nan")
        if pad == 1 and dil == 1 and st == 1 and ops.wino_supported(d):
            yw = ops.conv2d_wino(x, wt, sc, sh, ops.ACT_RELU, res)
            ew = float((yw.double() - ref).abs().max()) / scale
            up = ops.pack_wino_weight(wt)
            tw = timed(lambda: ops.conv2d_wino_launch(iv, up, sc, sh, ov, d, rv), reps)
        yf = ops.conv2d(x, wt, sc, sh, st, pad, dil, ops.ACT_RELU, res)
        ef = float((yf.double() - ref).abs().max()) / scale
        wpf = ops.pack_conv_weight(wt)
        tf = timed(lambda: ops.conv2d_launch(iv, wpf, sc, sh, ov, d, None, rv), reps)
        fl = 2.0 * n * co * ci * 9 * res.shape[2] * res.shape[3]
        print(f"{n:3d}x{ci:3d}->{co:3d} {h:3d}x{w:<3d} p{pad} d{dil:<2d} s{st}   {t3:8.1f} {tw:8.1f} {tf:8.1f}   {e3:9.2e} {ew:9.2e} {ef:9.2e}  {fl / t3 * 1e-6:6.1f}  nores {t30:6.1f}")


def pointwise():
    reps = 20
    print("1x1 convs (+ residual, ReLU):   x3 us   f32 us   x3 err")
    for n, ci, co, h, w in POINTWISE:
        x = torch.randn(n, ci, h, w, device="cuda")
        wt = torch.randn(co, ci, 1, 1, device="cuda") * (2.0 / ci) ** 0.5
        sc, sh = torch.rand(co, device="cuda") + 0.5, torch.randn(co, device="cuda")
        res = torch.randn(n, co, h, w, device="cuda")
        try:
            y = ops.conv2d_x3(x, wt, sc, sh, ops.ACT_RELU, res, 0, 1, 1)
        except (RuntimeError, ValueError):
            print(f"{n:3d}x{ci:3d}->{co:3d} {h:3d}x{w:<3d}            not covered by the split kernel")
            continue
        ref = torch.relu(F.conv2d(x[:2].double(), wt.double()) * sc.double().view(1, -1, 1, 1) + sh.double().view(1, -1, 1, 1) + res[:2].double())
        e3 = float((y[:2].double() - ref).abs().max()) / float(ref.abs().max())
        wp = ops.pack_x3_weight(wt, sc, 1)
        iv, ov, rv = ops.View(x), ops.View(y), ops.View(res)
        d = ops.conv_desc(iv, ov, co, 1, 1, 1, 0, 1, ops.ACT_RELU, None, rv)
        t3 = timed(lambda: ops.conv2d_x3_launch(iv, wp, sh, ov, d, rv), reps)
        wpf = ops.pack_conv_weight(wt)
        tf = timed(lambda: ops.conv2d_launch(iv, wpf, sc, sh, ov, d, None, rv), reps)
        print(f"{n:3d}x{ci:3d}->{co:3d} {h:3d}x{w:<3d}            {t3:8.1f} {tf:8.1f}   {e3:9.2e}")


if __name__ == "__main__":
    if "--pointwise" in sys.argv:
        pointwise()
    else:
        main()
