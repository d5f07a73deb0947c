"""S8 / LDS-DMA 3x3 convolution (csrc/convs.hip) next to the fp32-input split kernel (csrc/convx.hip): time per launch of the
two BasicBlock convs (conv1: S8 -> S8; conv2: S8 + fp32 residual -> fp32 + S8, or fp32 only) and of the S8 converter.
usage: python tools/convs_bench.py [--reps 20]"""
import sys
import torch

sys.path.insert(0, ".")
from otpose_amd import ops  # noqa: E402

SHAPES = [(80, 48, 48, 96, 72), (80, 96, 96, 48, 36), (80, 192, 192, 24, 18), (80, 384, 384, 12, 9), (80, 64, 64, 96, 72),
          (80, 32, 32, 64, 48), (80, 256, 256, 8, 6)]


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


def desc(n, ci, co, h, w):
    return ops.s8_conv_desc(n, ci, co, h, w, ops.ACT_RELU)


def main():
    reps = int(sys.argv[sys.argv.index("--reps") + 1]) if "--reps" in sys.argv else 20
    torch.manual_seed(0)
    print(f"{'shape':26s} {'convx':>7s} {'convx+r':>8s} | {'s8->s8':>7s} {'+r->nchw':>8s} {'+r->c4+s8':>10s} {'pack':>6s}  us;  block: convx / s8 (us)")
    for n, ci, co, h, w in SHAPES:
        x = torch.randn(n, ci, h, w, device="cuda")
        wt = torch.randn(co, ci, 3, 3, device="cuda") * (2.0 / (ci * 9)) ** 0.5
        sh = torch.randn(co, device="cuda")
        res = torch.randn(n, co, h, w, device="cuda")
        y = torch.empty(n, co, h, w, device="cuda")
        wp = ops.pack_x3_weight(wt, None, 1)
        iv, ov, rv = ops.View(x), ops.View(y), ops.View(res)
        d1 = ops.conv_desc(iv, ov, co, 3, 3, 1, 1, 1, ops.ACT_RELU, None, rv)
        d0 = ops.conv_desc(iv, ov, co, 3, 3, 1, 1, 1, ops.ACT_RELU, None, None)
        tx0 = timed(lambda: ops.conv2d_x3_launch(iv, wp, sh, ov, d0, None), reps)
        tx1 = timed(lambda: ops.conv2d_x3_launch(iv, wp, sh, ov, d1, rv), reps)
        if not ops.s8_conv_supported(desc(n, ci, co, h, w)):
            print(f"{n:3d}x{ci:3d}->{co:3d} {h:3d}x{w:<3d}   {tx0:7.1f} {tx1:8.1f} | not covered")
            continue
        xs = ops.s8_pack(x)
        ys = ops.s8_empty(n, co, h, w, "cuda")
        ds = desc(n, ci, co, h, w)
        ws = ops.pack_s8_weight(wt)
        rc4, oc4 = ops.c4_empty(n, co, h, w, "cuda"), ops.c4_empty(n, co, h, w, "cuda")
        ops.s8_pack(res, out_c4=rc4)
        ta = timed(lambda: ops.conv3x3_s8_launch(xs, ws, sh, ds, None, None, ops.S8_F32_C4, ys), reps)
        tb = timed(lambda: ops.conv3x3_s8_launch(xs, ws, sh, ds, rc4, y, ops.S8_F32_NCHW, None), reps)
        tc = timed(lambda: ops.conv3x3_s8_launch(xs, ws, sh, ds, rc4, oc4, ops.S8_F32_C4, ys), reps)
        tp = timed(lambda: ops.s8_pack(x, xs, rc4), reps)
        fl = 2.0 * n * co * ci * 9 * h * w
        print(f"{n:3d}x{ci:3d}->{co:3d} {h:3d}x{w:<3d}   {tx0:7.1f} {tx1:8.1f} | {ta:7.1f} {tb:8.1f} {tc:10.1f} {tp:6.1f}"
              f"      {tx0 + tx1:6.1f} / {ta + tc:6.1f}   ({fl / tc * 1e-6:5.0f} TF algorithmic on s8+r->f+s)")


if __name__ == "__main__":
    main()
