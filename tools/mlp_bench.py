#!/usr/bin/env python
"""Time the temporal-encoder kernels (csrc/mlp.hip, csrc/dense.hip) against the launches they replace, cfg2 shapes
(run on the GPU box)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from otpose_amd import ops                      # noqa: E402
from otpose_amd.ops import View                 # noqa: E402

ACT_GELU = 2


def timeit(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def main():
    B, C, HID = 16, 136, 544
    g = torch.Generator().manual_seed(0)
    for T in (6912, 3456, 1728):
        x, res = torch.randn(B, C, T, generator=g).cuda(), torch.randn(B, C, T, generator=g).cuda()
        w1, w2 = (torch.randn(HID, C, 1, generator=g) / C ** 0.5).cuda(), (torch.randn(C, HID, 1, generator=g) / HID ** 0.5).cuda()
        b1, b2, sc = torch.randn(HID, generator=g).cuda(), torch.randn(C, generator=g).cuda(), torch.randn(C, generator=g).cuda()
        packed = ops.pack_mlp_weights(w1, b1, w2)
        out = torch.empty_like(x)
        sh = (b2 * sc).contiguous()
        t_f = timeit(lambda: ops.mlp_fused(x, packed, sc, sh, res, out=out))
        hid = torch.empty(B, HID, T, device="cuda")
        o2 = torch.empty_like(x)
        v = lambda t: View(t.view(t.shape[0], t.shape[1], 1, t.shape[2]))        # noqa: E731

        xv, hv, ov, rv = v(x), v(hid), v(o2), v(res)
        p1, p2 = ops.pack_conv_weight(w1.view(HID, C, 1, 1)), ops.pack_conv_weight(w2.view(C, HID, 1, 1))
        one = torch.ones(HID, device="cuda")
        d1 = ops.conv_desc(xv, hv, HID, 1, 1, 1, 0, 1, ACT_GELU, None, None, 1)
        d2 = ops.conv_desc(hv, ov, C, 1, 1, 1, 0, 1, 0, None, rv, 1)

        def two():
            ops.conv2d_launch(xv, p1, one, b1, hv, d1)
            ops.conv2d_launch(hv, p2, sc, sh, ov, d2, None, rv)
        try:
            t_2 = timeit(two)
            err = float((o2 - out).abs().max())
        except Exception as e:                   # noqa: BLE001
            t_2, err = float("nan"), str(e)
        fl = 4.0 * C * HID * B * T
        print(f"T={T}: fused {t_f:.1f} us ({fl / t_f / 1e6:.1f} TFLOP/s)  two launches {t_2:.1f} us  max|diff| {err}")




def dense_main():
    B, C = 16, 136
    g = torch.Generator().manual_seed(1)
    for T in (6912, 3456, 1728):
        xs = [torch.randn(B, C, T, generator=g).cuda() for _ in range(3)]
        ws = [(torch.randn(C, C, 1, generator=g) / C ** 0.5).cuda() for _ in range(3)]
        bs = [torch.randn(C, generator=g).cuda() for _ in range(3)]
        packs = [ops.pack_dense_cc(w, None, b) for w, b in zip(ws, bs)]
        outs = [torch.empty_like(x) for x in xs]
        t3 = timeit(lambda: ops.dense_cc(xs, packs, None, outs))
        t1 = timeit(lambda: ops.dense_cc(xs[:1], packs[:1], None, outs[:1]))
        v = lambda t: View(t.view(t.shape[0], t.shape[1], 1, t.shape[2]))        # noqa: E731
        o2 = torch.empty_like(xs[0])
        xv, ov = v(xs[0]), v(o2)
        pw = ops.pack_conv_weight(ws[0].view(C, C, 1, 1))
        one = torch.ones(C, device="cuda")
        d = ops.conv_desc(xv, ov, C, 1, 1, 1, 0, 1, 0, None, None, 1)
        tc = timeit(lambda: ops.conv2d_launch(xv, pw, one, bs[0], ov, d))
        err = float((o2 - outs[0]).abs().max())
        gb = 2 * 4.0 * B * C * T / 1e3
        print(f"T={T}: dense x3 {t3:.1f} us ({3 * gb / t3:.0f} GB/s)  x1 {t1:.1f} us ({gb / t1:.0f} GB/s)  otp_conv2d x1 {tc:.1f} us  "
              f"max|diff| {err}")




def qkv_main():
    B, C = 16, 136
    g = torch.Generator().manual_seed(2)
    r = lambda *s: torch.randn(*s, generator=g).cuda()       # noqa: E731
    for T in (6912,):
        x = r(B, C, T)
        dws, gs, bs = [r(C, 1, 3) for _ in range(3)], [r(C) for _ in range(3)], [r(C) for _ in range(3)]
        ws, cb = [r(C, C, 1) / C ** 0.5 for _ in range(3)], [r(C) for _ in range(3)]
        table = ops.pack_qkv_table(*dws, gs[0], bs[0], gs[1], bs[1], gs[2], bs[2])
        packs = [ops.pack_dense_cc(w, None, b) for w, b in zip(ws, cb)]
        outs = [torch.empty_like(x) for _ in range(3)]
        tf = timeit(lambda: ops.qkv_front(x, table, packs, 1e-5, outs))
        mids = [torch.empty_like(x) for _ in range(3)]

        def two():
            m = ops.dwconv_ln3(x, dws, gs, bs, 1, 1e-5)
            ops.dense_cc(m, packs, None, mids)
        t2 = timeit(two)
        print(f"T={T}: qkv_front {tf:.1f} us   dwconv_ln3 + dense_cc x3 {t2:.1f} us   max|diff| {float((outs[0] - mids[0]).abs().max())}")


if __name__ == "__main__":
    main()
    dense_main()
    qkv_main()
