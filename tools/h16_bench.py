#!/usr/bin/env python
"""Timing of the fp16 engine (csrc/h16.hip, otpose_amd/engine_h16.py): per-launch times of the backbone's shapes at batch 16 x 5
frames and the whole forward of cfg2 / cfg5 in fp16 next to the fp32 engine.  HIP events on the launch stream."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from otpose_amd import OTPose, cfg2, ops  # noqa: E402
from otpose_amd import synthetic as S  # noqa: E402
from otpose_amd.config import cfg5  # noqa: E402


def ev(fn, iters=20):
    fn()
    torch.cuda.synchronize()
    st = torch.cuda.current_stream()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(iters):
        fn()
    e1.record(st)
    e1.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3          # us


def conv_case(n, cin, cout, h, w, stride, res):
    g = torch.Generator().manual_seed(1)
    x = ops.h8_pack(torch.randn(n, cin, h, w, generator=g).cuda())
    wt = (torch.randn(cout, cin, 3, 3, generator=g) * 0.05).cuda()
    sh = torch.zeros(cout, device="cuda")
    k = ops.h16_weight_exponent(wt)
    wp = ops.pack_h16_conv_weight(wt, None, k)
    ho, wo = h // stride, w // stride
    r = ops.h8_pack(torch.randn(n, cout, ho, wo, generator=g).cuda()) if res else None
    out = ops.h8_empty(n, cout, ho, wo, "cuda")
    d = ops.h16_conv_desc(x, cout, stride, ops.ACT_RELU, out, r, k)
    t = ev(lambda: ops.h16_conv3x3(x, wp, sh, cout, stride, ops.ACT_RELU, r, out=out, desc=d))
    flop = 2.0 * cin * cout * 9 * ho * wo * n
    byt = 2.0 * n * (cin * h * w + cout * ho * wo * (2 if res else 1))
    print(f"conv3x3 s{stride} {cin:3d}->{cout:3d} @{h}x{w} x{n} res={int(res)}: {t:7.1f} us  {flop / t / 1e6:7.1f} TFLOP/s  {byt / t / 1e6:5.2f} TB/s", flush=True)


def pw_case(n, cin, cout, h, w, res):
    g = torch.Generator().manual_seed(2)
    x = ops.h8_pack(torch.randn(n, cin, h, w, generator=g).cuda())
    wt = (torch.randn(cout, cin, generator=g) * 0.05).cuda()
    pk = ops.pack_h16_pointwise(wt, None, torch.zeros(cout, device="cuda"), 0)
    r = ops.h8_pack(torch.randn(n, cout, h, w, generator=g).cuda()) if res else None
    out = ops.h8_empty(n, cout, h, w, "cuda")
    t = ev(lambda: ops.h16_pointwise(x, pk, cout, True, r, out=out))
    byt = 2.0 * n * h * w * (cin + cout * (2 if res else 1))
    print(f"pointwise {cin:3d}->{cout:3d} @{h}x{w} x{n} res={int(res)}: {t:7.1f} us  {byt / t / 1e6:5.2f} TB/s", flush=True)


def forward(cfg, frames, label, steps=20):
    m = OTPose(cfg)
    S.fill_synthetic_(m)
    m = m.cuda().eval()
    m.alias_outputs = True
    x, g = S.synthetic_clip(16, cfg.MODEL.IMAGE_SIZE, frames=frames)
    x, g = x.cuda(), g.cuda()
    with torch.no_grad():
        for _ in range(3):
            o = m(x, margin=g)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            o = m(x, margin=g)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
    print(f"{label}: {dt * 1e3:.2f} ms / forward = {16 * frames / dt:.0f} frames/s (finite {bool(torch.isfinite(o[0]).all())})", flush=True)
    return m


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    n = 80
    if what in ("all", "kernels"):
        for c, h, w in ((48, 96, 72), (96, 48, 36), (192, 24, 18), (384, 12, 9), (64, 96, 72)):
            conv_case(n, c, c, h, w, 1, False)
            conv_case(n, c, c, h, w, 1, True)
        conv_case(n, 256, 48, 96, 72, 1, False)
        for cin, cout, h, w in ((64, 64, 192, 144), (48, 96, 96, 72), (48, 48, 96, 72), (96, 192, 48, 36), (192, 384, 24, 18), (256, 96, 96, 72)):
            conv_case(n, cin, cout, h, w, 2, False)
        for cin, cout, h, w, r in ((256, 64, 96, 72, False), (64, 256, 96, 72, True), (64, 64, 96, 72, False), (96, 48, 48, 36, False),
                                   (384, 48, 12, 9, False), (384, 192, 12, 9, False)):
            pw_case(n, cin, cout, h, w, r)
        g = torch.Generator().manual_seed(3)
        clip = torch.randn(16, 15, 384, 288, generator=g).cuda()
        pk = ops.pack_h16_stem((torch.randn(64, 3, 3, 3, generator=g) * 0.2).cuda(), None, torch.zeros(64, device="cuda"))
        out = ops.h8_empty(80, 64, 192, 144, "cuda")
        t = ev(lambda: ops.h16_stem(clip, pk, 64, 5, out=out))
        print(f"stem 3->64 s2 @384x288 x80: {t:7.1f} us  {(clip.numel() * 4 + 80 * 64 * 192 * 144 * 2) / t / 1e6:5.2f} TB/s", flush=True)
    if what == "enc":                                # the encoders' matrix kernels: split products against half operands
        for C, T in ((136, 6912), (204, 6912)):
            B, HID = 16, 4 * C
            g = torch.Generator().manual_seed(5)
            x = torch.randn(B, C, T, generator=g).cuda()
            w1, w2 = (torch.randn(HID, C, 1, generator=g) / C ** 0.5).cuda(), (torch.randn(C, HID, 1, generator=g) / HID ** 0.5).cuda()
            b1, one, zero = torch.zeros(HID, device="cuda"), torch.ones(C, device="cuda"), torch.zeros(C, device="cuda")
            out = torch.empty_like(x)
            for half in (False, True):
                pk = ops.pack_mlp_x3_weights(w1, b1, w2, half=half)
                t = ev(lambda: ops.ln_mlp_x3(x, one, zero, 1e-5, pk, one, zero, out=out, half=half))
                print(f"ln_mlp C={C} half={int(half)}: {t:7.1f} us", flush=True)
            wq = (torch.randn(C, C, generator=g) / C ** 0.5).cuda()
            pkd = ops.pack_dense_cc(wq, one, zero, x3=True)
            for half in (False, True):
                t = ev(lambda: ops.dense_cc([x], [pkd], [x], outs=[out], x3=True, half=half))
                print(f"dense  C={C} half={int(half)}: {t:7.1f} us", flush=True)
            dws = [(torch.randn(C, 1, 3, generator=g) * 0.5).cuda() for _ in range(3)]
            table = ops.pack_qkv_table(dws[0], dws[1], dws[2], one, zero, one, zero, one, zero)
            outs = [torch.empty_like(x) for _ in range(3)]
            for half in (False, True):
                t = ev(lambda: ops.qkv_front(x, table, [pkd, pkd, pkd], outs=outs, x3=True, half=half))
                print(f"qkv    C={C} half={int(half)}: {t:7.1f} us", flush=True)
    if what == "fwd16":                              # one model only (profiling): cfg2 or cfg5 in fp16
        which = sys.argv[2] if len(sys.argv) > 2 else "cfg2"
        if which == "cfg5":
            forward(cfg5("fp16"), 7, "cfg5 fp16", steps=10)
        else:
            c = cfg2()
            c.MODEL.DTYPE = "fp16"
            forward(c, 5, "cfg2 fp16", steps=10)
    if what in ("all", "forward"):
        c = cfg2()
        c.MODEL.DTYPE = "fp16"
        forward(c, 5, "cfg2 fp16")
        forward(cfg2(), 5, "cfg2 fp32 (f16x3)")
        forward(cfg5("fp16"), 7, "cfg5 fp16")
        forward(cfg5(), 7, "cfg5 fp32 (f16x3)")
