"""Split-record (S8) activations and the LDS-DMA fed 3x3 convolution (csrc/convs.hip, otp_conv3x3_s8) - the HRNet BasicBlock
convs of model/HRNet.py:500-530 - against a float64 ``F.conv2d`` of the same operands, and the S8 converters against their
definition.  Tolerance as for otp_conv2d_x3 (tests/test_gpu_convx.py): 2e-5 of the output range for the fp32 result; the
S8 result additionally carries the split's own remainder (2^-22 relative per element with the IEEE-half pieces of round 4)."""
import pytest
import torch
import torch.nn.functional as F

from otpose_amd import ops

pytestmark = pytest.mark.gpu


def _h16_rne(x):
    """the 16-bit operand type of the split products: IEEE half since round 4 (bfloat16 before; csrc/common.h)"""
    return x.to(torch.float16).to(torch.float32)


@pytest.mark.parametrize("shape,coff,ctot", [((3, 48, 12, 8), 0, 48), ((2, 16, 6, 6), 8, 40), ((5, 96, 24, 18), 0, 96)])
def test_s8_pack_is_hi_lo_of_the_definition(shape, coff, ctot):
    n, c, h, w = shape
    g = torch.Generator().manual_seed(c + h)
    full = (torch.randn(n, ctot, h, w, generator=g) * 3).cuda()
    full[0, coff, 0, 0] = 0.0
    full[0, coff + 1, 0, 1] = 1e-30                       # below the smallest half subnormal: both pieces 0
    full[0, coff + 2, 0, 2] = 3e-6                        # a subnormal half: hi carries 6 bits of it, lo the rest
    x = full[:, coff:coff + c]
    s8 = ops.s8_pack(ops.View(full, coff, c))
    rec = s8.view(torch.int16).view(n, c // 8, 2, h * w, 8)       # [n][g][part][p][e] 16-bit patterns
    hi = _h16_rne(x)
    lo = _h16_rne(x - hi)
    want = torch.stack([hi, lo], 0).view(2, n, c // 8, 8, h * w).permute(1, 2, 0, 4, 3)   # -> [n][g][part][p][e]
    got = rec.view(torch.float16).to(torch.float32)
    assert torch.equal(got, want.contiguous())
    back = ops.s8_unpack(s8, n, c, h, w)
    assert torch.equal(back, hi + lo)
    # two half pieces carry 22 significand bits, or everything down to the half subnormal spacing 2^-24
    assert float((back - x).abs().max()) <= max(2.0 ** -21 * float(x.abs().max()), 2.0 ** -24)


# (N, Cin, Cout, H, W, residual, relu)
CASES = [
    (5, 48, 48, 96, 72, True, True),         # HRNet-W48 branch 0: tiles end inside rows, windows of 7 rows (512 records)
    (3, 96, 96, 48, 36, False, True),        # branch 1: tiles straddle images, two cout blocks share a window
    (3, 192, 192, 24, 18, True, False),      # branch 2
    (7, 384, 96, 12, 9, True, True),         # branch 3 maps: up to four images per tile
    (2, 64, 64, 96, 72, False, False),       # NTW = 2 (layer1 width)
    (2, 32, 80, 10, 6, True, True),          # Cout = 5 tiles of 16: a partial last cout block (lone tile in the S8 store), tail tile
    (1, 16, 16, 8, 4, False, False),         # one partial tile, one chunk
    (80, 384, 384, 12, 9, True, True),       # launches of a few hundred workgroups (XCD-interleaved ranges with ragged ends)
    (40, 192, 192, 24, 18, True, False),
    (16, 48, 48, 96, 72, True, True),
    (3, 32, 32, 64, 48, True, True),         # HRNet-W32 branch 0 (cfg1)
    (3, 128, 128, 16, 12, False, True),
]


@pytest.mark.parametrize("case", CASES, ids=lambda c: "x".join(str(v) for v in c[:5]))
def test_conv3x3_s8_matches_float64(case):
    n, ci, co, h, w, with_res, relu = case
    g = torch.Generator(device="cpu").manual_seed(sum(case[:5]))
    x = torch.randn(n, ci, h, w, generator=g).cuda()
    wt = (torch.randn(co, ci, 3, 3, generator=g) * (2.0 / (ci * 9)) ** 0.5).cuda()
    sc = (torch.rand(co, generator=g) + 0.5).cuda()
    sh = torch.randn(co, generator=g).cuda()
    res = torch.randn(n, co, h, w, generator=g).cuda() if with_res else None
    xs = ops.s8_pack(x)
    res_c4 = None
    if with_res:                                          # the residual travels in the C4 layout
        res_c4 = ops.c4_empty(n, co, h, w, "cuda")
        ops.s8_pack(res, out_c4=res_c4)
        assert torch.equal(ops.c4_unpack(res_c4, n, co, h, w), res)
    # the kernel's input IS hi + lo (the S8 image); the float64 reference convolves exactly that
    xin = ops.s8_unpack(xs, n, ci, h, w).double()
    ref = F.conv2d(xin, wt.double(), None, 1, 1, 1) * sc.double().view(1, -1, 1, 1) + sh.double().view(1, -1, 1, 1)
    if with_res:
        ref = ref + res.double()
    if relu:
        ref = torch.relu(ref)
    act = ops.ACT_RELU if relu else ops.ACT_NONE
    y, ys = ops.conv3x3_s8(xs, (n, ci, h, w), wt, sc, sh, act, res_c4)
    scale = float(ref.abs().max())
    err = float((y.double() - ref).abs().max()) / scale
    assert err <= 2e-5, err
    # against the fp32 input itself (what the engine compares with): the split drops 2^-18 of every input element
    ref0 = F.conv2d(x.double(), wt.double(), None, 1, 1, 1) * sc.double().view(1, -1, 1, 1) + sh.double().view(1, -1, 1, 1)
    ref0 = ref0 + res.double() if with_res else ref0
    ref0 = torch.relu(ref0) if relu else ref0
    assert float((y.double() - ref0).abs().max()) / scale <= 2e-5
    # the S8 output is the split of the fp32 output, bit for bit
    assert torch.equal(ys, ops.s8_pack(y))
    # the C4 output holds the same values; the single-output variants write the same values
    yc, none_s8 = ops.conv3x3_s8(xs, (n, ci, h, w), wt, sc, sh, act, res_c4, f32="c4", want_s8=False)
    assert none_s8 is None and torch.equal(ops.c4_unpack(yc, n, co, h, w), y)
    none_f, ys2 = ops.conv3x3_s8(xs, (n, ci, h, w), wt, sc, sh, act, res_c4, f32=None)
    assert none_f is None and torch.equal(ys2, ys)
    if with_res:
        # the residual as S8 records (otp_conv_desc.res_layout = 1: a BasicBlock's input image serves as its residual): the same
        # bits as the C4 form fed with hi + lo of those records, and within 2^-22 of the residual's magnitude of the fp32 form
        rs8 = ops.s8_pack(res)
        r_hl = ops.s8_unpack(rs8, n, co, h, w)
        rc4 = ops.c4_empty(n, co, h, w, "cuda")
        ops.s8_pack(r_hl, out_c4=rc4)
        y_s, ys_s = ops.conv3x3_s8(xs, (n, ci, h, w), wt, sc, sh, act, res_s8=rs8)
        y_c, ys_c = ops.conv3x3_s8(xs, (n, ci, h, w), wt, sc, sh, act, rc4)
        assert torch.equal(y_s, y_c) and torch.equal(ys_s, ys_c)
        assert float((y_s - y).abs().max()) <= 2.0 ** -21 * float(res.abs().max())


def test_conv3x3_s8_writes_a_channel_slice_of_an_nchw_tensor():
    g = torch.Generator(device="cpu").manual_seed(9)
    x = torch.randn(3, 32, 20, 12, generator=g).cuda()
    wt = (torch.randn(16, 32, 3, 3, generator=g) * 0.1).cuda()
    big = torch.full((3, 40, 20, 12), 7.0, device="cuda")
    d = ops.s8_conv_desc(3, 32, 16, 20, 12, ops.ACT_NONE, ops.View(big, 8, 16))
    e = ops.x3_weight_exponent(wt)                       # the wrapper below stores the weights times 2^e: same here
    d.out_scale = 2.0 ** -e
    ops.conv3x3_s8_launch(ops.s8_pack(x), ops.pack_s8_weight(wt, None, e), None, d, None, big, ops.S8_F32_NCHW, None)
    y, _ = ops.conv3x3_s8(ops.s8_pack(x), tuple(x.shape), wt, want_s8=False)
    assert torch.equal(big[:, 8:24], y) and float((big[:, :8] - 7).abs().max()) == 0 and float((big[:, 24:] - 7).abs().max()) == 0


def test_conv3x3_s8_agrees_with_the_fp32_input_kernel():
    """Same arithmetic as otp_conv2d_x3 on the same operands (the split of the input commutes with where it happens): the
    two kernels differ only in summation order."""
    g = torch.Generator(device="cpu").manual_seed(5)
    x = torch.randn(4, 96, 48, 36, generator=g).cuda()
    wt = (torch.randn(96, 96, 3, 3, generator=g) * 0.05).cuda()
    sh = torch.randn(96, generator=g).cuda()
    a = ops.conv2d_x3(x, wt, None, sh, ops.ACT_RELU)
    b, _ = ops.conv3x3_s8(ops.s8_pack(x), tuple(x.shape), wt, None, sh, ops.ACT_RELU)
    assert float((a - b).abs().max()) <= 3e-6 * float(a.abs().max())


def test_conv3x3_s8_rejects_what_it_does_not_cover():
    d = ops.s8_conv_desc(2, 48, 48, 96, 72)
    assert ops.s8_conv_supported(d)
    for field, val in (("stride", 2), ("dil", 2), ("Cin", 24), ("kh", 1), ("W", 500), ("Cout", 40)):
        e = ops.hip.ConvDesc.from_buffer_copy(bytes(d))
        setattr(e, field, val)
        if field == "W":
            e.Wo = val
        assert not ops.s8_conv_supported(e), field


def test_conv3x3_s8_is_bit_stable_next_to_other_kernels():
    """The engine runs HRNet branches on parallel streams: an S8 block (conv1 -> conv2 with C4 residual, both outputs) must
    give the same bits alone and with fp32-input split convolutions and other S8 convolutions sharing the CUs (a hand-placed
    v_permlane32_swap once made the S8 records depend on who else was resident)."""
    def setup(n, c, h, w, seed):
        g = torch.Generator().manual_seed(seed)
        x = torch.randn(n, c, h, w, generator=g).cuda()
        wt = (torch.randn(c, c, 3, 3, generator=g) * (1.0 / (c * 9)) ** 0.5).cuda()
        xs, xc4 = ops.s8_empty(n, c, h, w, "cuda"), ops.c4_empty(n, c, h, w, "cuda")
        ops.s8_pack(x, xs, xc4)
        return dict(xs=xs, xc4=xc4, wp=ops.pack_s8_weight(wt), sh=(torch.randn(c, generator=g) * 0.1).cuda(),
                    d=ops.s8_conv_desc(n, c, c, h, w, ops.ACT_RELU), y8=ops.s8_empty(n, c, h, w, "cuda"),
                    o4=ops.c4_empty(n, c, h, w, "cuda"), o8=ops.s8_empty(n, c, h, w, "cuda"))

    def run(b, st):
        ops.conv3x3_s8_launch(b["xs"], b["wp"], b["sh"], b["d"], None, None, ops.S8_F32_C4, b["y8"], stream=st.cuda_stream)
        ops.conv3x3_s8_launch(b["y8"], b["wp"], b["sh"], b["d"], b["xc4"], b["o4"], ops.S8_F32_C4, b["o8"], stream=st.cuda_stream)

    blocks = [setup(5, 48, 96, 72, 1), setup(5, 96, 48, 36, 2), setup(5, 192, 24, 18, 3)]
    streams = [torch.cuda.Stream() for _ in range(4)]
    refs = []
    for b in blocks:
        run(b, torch.cuda.current_stream())
        torch.cuda.synchronize()
        refs.append((b["o4"].clone(), b["o8"].clone()))
    xx = torch.randn(5, 96, 48, 36, device="cuda")
    wp = ops.pack_x3_weight(torch.randn(96, 96, 3, 3, device="cuda") * 0.03, None, 1)
    yy = torch.empty_like(xx)
    iv, ov = ops.View(xx), ops.View(yy)
    dd = ops.conv_desc(iv, ov, 96, 3, 3, 1, 1, 1, ops.ACT_RELU)
    for it in range(10):
        for b in blocks:
            b["o4"].zero_(), b["o8"].zero_(), b["y8"].zero_()
        torch.cuda.synchronize()
        for k in range(3):
            ops.conv2d_x3_launch(iv, wp, None, ov, dd, None, stream=streams[3].cuda_stream)
        for b, st in zip(blocks, streams):
            run(b, st)
        for k in range(3):
            ops.conv2d_x3_launch(iv, wp, None, ov, dd, None, stream=streams[3].cuda_stream)
        torch.cuda.synchronize()
        for b, (r4, r8) in zip(blocks, refs):
            assert torch.equal(b["o4"], r4) and torch.equal(b["o8"], r8), it


@pytest.mark.parametrize("factors", [(2,), (2, 4), (2, 4, 8)])
def test_s8_upsample_add_equals_upsample_add_multi_then_pack(factors):
    """otp_s8_upsample_add = otp_upsample_add_multi followed by otp_s8_pack (S8 + C4), bit for bit, with and without the
    NCHW output (HRNet fuse rows, model/HRNet.py:487-494)."""
    import ctypes
    from otpose_amd import hip
    L = hip.lib()
    n, c, hh, wh = 3, 48, 32, 24
    g = torch.Generator().manual_seed(len(factors))
    res = torch.randn(n, c, hh, wh, generator=g).cuda()
    lows = [torch.randn(n, c, hh // f, wh // f, generator=g).cuda() for f in factors]
    lp = (ctypes.c_void_p * len(lows))(*[hip.ptr(t) for t in lows])
    fp = (ctypes.c_int * len(lows))(*factors)
    want = torch.empty_like(res)
    hip.check(L.otp_upsample_add_multi(lp, fp, len(lows), hip.ptr(res), hip.ptr(want), n, c, hh, wh, 1, c, 0, c, 0,
                                       hip.stream_of(res)), "multi")
    ref = torch.relu(res + sum(F.interpolate(t, scale_factor=f, mode="nearest") for t, f in zip(lows, factors)))
    assert float((want - ref).abs().max()) <= 1e-6
    want_c4 = ops.c4_empty(n, c, hh, wh, "cuda")
    want_s8 = ops.s8_pack(want, out_c4=want_c4)
    for with_nchw in (True, False):
        out = torch.full_like(res, 7.0)
        s8, c4 = ops.s8_empty(n, c, hh, wh, "cuda"), ops.c4_empty(n, c, hh, wh, "cuda")
        hip.check(L.otp_s8_upsample_add(lp, fp, len(lows), hip.ptr(res), hip.ptr(out) if with_nchw else None, hip.ptr(s8),
                                        hip.ptr(c4), n, c, hh, wh, 1, c, 0, c, 0, hip.stream_of(res)), "s8 up")
        assert torch.equal(s8, want_s8) and torch.equal(c4, want_c4)
        assert torch.equal(out, want) if with_nchw else bool((out == 7.0).all())


# ---- stride 2 (csrc/convs2.hip): the down-sampling chains of the fuse layers, transitions, the stem's conv2 ---------------------
# (N, Cin, Cout, H, W, residual, relu)
S2_CASES = [
    (5, 48, 96, 96, 72, True, False),        # HRNet-W48 fuse 0 -> 1: two-image... 128-pixel tiles, 11 window rows of 73 records
    (3, 48, 48, 96, 72, False, True),        # a chain's intermediate (keeps the width, ReLU)
    (5, 96, 192, 48, 36, True, True),        # fuse 1 -> 2: tiles straddle images
    (7, 192, 384, 24, 18, True, True),       # fuse 2 -> 3: 12 x 9 outputs, more than one image per tile
    (3, 48, 192, 48, 36, True, False),       # last conv of a two-step chain
    (2, 64, 64, 192, 144, False, True),      # the stem's conv2: 64-pixel tiles (145-record window rows), NTW = 2
    (3, 256, 96, 96, 72, False, True),       # transition1's new branch
    (2, 32, 80, 20, 12, True, True),         # Cout = 5 tiles of 16: a lone tile, tail tile
    (1, 16, 16, 8, 8, False, False),         # one partial tile, one chunk
    (80, 192, 384, 24, 18, True, True),      # full batch: XCD-interleaved tile ranges with ragged ends
    (16, 48, 96, 96, 72, True, True),
    (3, 32, 64, 64, 48, True, True),         # HRNet-W32 (cfg1)
]


@pytest.mark.parametrize("case", S2_CASES, ids=lambda c: "x".join(str(v) for v in c[:5]))
def test_conv3x3_stride2_s8_matches_float64(case):
    n, ci, co, h, w, with_res, relu = case
    g = torch.Generator(device="cpu").manual_seed(sum(case[:5]) + 1)
    x = torch.randn(n, ci, h, w, generator=g).cuda()
    wt = (torch.randn(co, ci, 3, 3, generator=g) * (2.0 / (ci * 9)) ** 0.5).cuda()
    sc = (torch.rand(co, generator=g) + 0.5).cuda()
    sh = torch.randn(co, generator=g).cuda()
    res = torch.randn(n, co, h // 2, w // 2, generator=g).cuda() if with_res else None
    xs = ops.s8_pack(x)
    xin = ops.s8_unpack(xs, n, ci, h, w).double()                       # the kernel's input IS hi + lo
    ref = F.conv2d(xin, wt.double(), None, 2, 1, 1) * sc.double().view(1, -1, 1, 1) + sh.double().view(1, -1, 1, 1)
    ref_nores = torch.relu(ref) if relu else ref
    if with_res:
        ref = ref + res.double()
    if relu:
        ref = torch.relu(ref)
    act = ops.ACT_RELU if relu else ops.ACT_NONE
    assert ops.s8_s2_conv_supported(ops.s8_s2_conv_desc(n, ci, co, h, w, act))
    y = ops.conv3x3_s2_s8(xs, (n, ci, h, w), wt, sc, sh, act, res)
    scale = float(ref.abs().max())
    err = float((y.double() - ref).abs().max()) / scale
    assert err <= 2e-5, err
    # S8 output form (no residual): the split of the fp32 result of the same call, bit for bit
    y0 = ops.conv3x3_s2_s8(xs, (n, ci, h, w), wt, sc, sh, act, None)
    assert float((y0.double() - ref_nores).abs().max()) / float(ref_nores.abs().max()) <= 2e-5
    y8 = ops.conv3x3_s2_s8(xs, (n, ci, h, w), wt, sc, sh, act, None, out="s8")
    assert torch.equal(y8, ops.s8_pack(y0))


def test_conv3x3_stride2_s8_accumulates_in_place_into_a_channel_slice():
    """A fuse row adds its terms in place (model/HRNet.py:488-494): residual = output = a channel slice of a wider tensor."""
    g = torch.Generator(device="cpu").manual_seed(19)
    n, ci, co, h, w = 3, 32, 16, 24, 16
    x = torch.randn(n, ci, h, w, generator=g).cuda()
    wt = (torch.randn(co, ci, 3, 3, generator=g) * 0.1).cuda()
    big = torch.randn(n, 40, h // 2, w // 2, generator=g).cuda()
    before = big.clone()
    v = ops.View(big, 8, co)
    d = ops.s8_s2_conv_desc(n, ci, co, h, w, ops.ACT_RELU, v, v)
    from otpose_amd import hip
    e = ops.x3_weight_exponent(wt)
    d.out_scale = 2.0 ** -e
    xs, wp = ops.s8_pack(x), ops.pack_s8_weight(wt, None, e)           # (kept alive across the launch)
    hip.check(hip.lib().otp_conv3x3_s2_s8(hip.ptr(xs), hip.ptr(wp), None, hip.ptr(big), hip.ptr(big), None, d,
                                          hip.stream_of(big)), "otp_conv3x3_s2_s8")
    want = ops.conv3x3_s2_s8(xs, (n, ci, h, w), wt, None, None, ops.ACT_RELU, before[:, 8:24].contiguous())
    assert torch.equal(big[:, 8:24], want)
    assert torch.equal(big[:, :8], before[:, :8]) and torch.equal(big[:, 24:], before[:, 24:])


@pytest.mark.parametrize("kind", ["s8", "s2", "x3", "x3_1x1", "pointwise"])
def test_weight_exponent_keeps_small_weights_to_fp32_accuracy(kind, monkeypatch):
    """BatchNorm-folded weights of magnitude 1e-2: unscaled, the `lo` half piece of a weight is subnormal and the pair holds it to
    2^-25 absolute (~17 bits) - an error common to all pixels; stored times 2^k (ops.x3_weight_exponent) and the sum multiplied
    by 2^-k (otp_conv_desc.out_scale) the result is within fp32 rounding of float64 arithmetic on the same operands."""
    g = torch.Generator(device="cpu").manual_seed(23)
    n, ci, co, h, w = 2, 64, 64, 32, 24
    k = 1 if kind in ("x3_1x1", "pointwise") else 3
    x = torch.randn(n, ci, h, w, generator=g).cuda()
    wt = (torch.randn(co, ci, k, k, generator=g) * 0.01).cuda()
    sc = (torch.rand(co, generator=g) * 0.5 + 0.5).cuda()
    sh = (torch.randn(co, generator=g) * 0.1).cuda()
    xs = ops.s8_pack(x)
    xin = ops.s8_unpack(xs, n, ci, h, w).double()
    stride = 2 if kind == "s2" else 1
    ref = torch.relu(F.conv2d(xin, wt.double(), None, stride, k // 2, 1) * sc.double().view(1, -1, 1, 1) + sh.double().view(1, -1, 1, 1))

    def run():
        if kind == "s8":
            return ops.conv3x3_s8(xs, (n, ci, h, w), wt, sc, sh, ops.ACT_RELU, want_s8=False)[0]
        if kind == "s2":
            return ops.conv3x3_s2_s8(xs, (n, ci, h, w), wt, sc, sh, ops.ACT_RELU)
        if kind == "pointwise":
            out = torch.empty(n, co, h, w, device="cuda")
            ops.pointwise_x3(ops.View(x), ops.pack_pointwise_x3(wt, sc, sh), ops.View(out), None, True)
            return out
        return ops.conv2d_x3(x, wt, sc, sh, ops.ACT_RELU, None, k // 2, 1, 1)

    # the part of the result the weights contribute (shift is exact in both): errors as a fraction of ITS range
    span = float((ref - torch.relu(sh.double()).view(1, -1, 1, 1)).abs().max())
    scaled = float((run().double() - ref).abs().max()) / span
    monkeypatch.setenv("OTPOSE_X3_WSCALE", "0")
    plain = float((run().double() - ref).abs().max()) / span
    print("\n%s: max err / span  scaled %.2e  unscaled %.2e" % (kind, scaled, plain))
    assert scaled <= 2e-6 and plain >= 3 * scaled
