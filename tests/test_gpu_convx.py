"""Split-bf16 (bf16x3) convolution kernel (csrc/convx.hip, otp_conv2d_x3) against a float64 ``F.conv2d`` of the same
operands: the HRNet 3x3 convs of model/HRNet.py:500-571 (stride 1) and :442-470 (stride-2 transitions / fuse downsamples),
its 1x1 convs (Bottleneck conv1 / conv3 / shortcut :533-571, fuse-layer 1x1s :455-462), plus the dilated offset / mask convs
of model/OTPose.py:168-177.  Tolerance: 2e-5 of the output range (measured 4e-6; the
f32-MFMA kernels sit at 3e-7 .. 2e-6): two bf16 pieces carry 16 mantissa bits + rounding, see the kernel header."""
import pytest
import torch
import torch.nn.functional as F

from otpose_amd import ops

pytestmark = pytest.mark.gpu

# (N, Cin, Cout, H, W, pad, dil, stride, residual, relu)
CASES = [
    (5, 48, 48, 96, 72, 1, 1, 1, True, True),        # tiles end inside rows, 27 tiles per image
    (3, 96, 96, 48, 36, 1, 1, 1, False, True),       # tiles straddle images
    (3, 192, 192, 24, 18, 1, 1, 1, True, False),     # W % 4 != 0: float4 items wrap rows
    (7, 384, 96, 12, 9, 1, 1, 1, True, True),        # several images per tile, odd width
    (2, 64, 64, 96, 72, 1, 1, 1, False, False),      # 4 n-tiles per workgroup
    (2, 32, 40, 10, 6, 1, 1, 1, True, True),         # tiny maps, Cout not a multiple of 16, tail tile
    (1, 16, 17, 8, 4, 1, 1, 1, False, False),        # one partial tile
    (2, 32, 459, 24, 20, 3, 3, 1, False, False),     # dilated offset conv shape (pad = dilation)
    (2, 32, 153, 40, 28, 6, 6, 1, False, False),     # dilation 6: half of the window is padding
    (2, 16, 16, 12, 12, 0, 1, 1, False, True),       # valid convolution (no padding)
    (5, 48, 48, 96, 72, 1, 1, 2, False, True),       # stride 2: de-interleaved columns, 8-channel chunks
    (3, 48, 96, 96, 72, 1, 1, 2, True, True),
    (3, 96, 192, 48, 36, 1, 1, 2, True, False),
    (5, 192, 384, 24, 18, 1, 1, 2, True, True),
    (2, 64, 64, 64, 48, 1, 1, 2, False, True),
    (2, 24, 40, 20, 12, 1, 1, 2, True, False),       # Cin % 8 == 0 only (stride-2 chunking)
    # launches of a few hundred workgroups (XCD-interleaved tile ranges with ragged ends)
    (80, 384, 384, 12, 9, 1, 1, 1, True, True),
    (40, 192, 192, 24, 18, 1, 1, 1, True, False),
    (20, 48, 96, 48, 36, 1, 1, 1, False, True),
    (16, 48, 48, 96, 72, 1, 1, 1, True, True),
]


@pytest.mark.parametrize("case", CASES, ids=lambda c: "x".join(str(v) for v in c[:8]))
def test_conv2d_x3_matches_float64(case):
    n, ci, co, h, w, pad, dil, st, with_res, relu = case
    g = torch.Generator(device="cpu").manual_seed(sum(case[:8]))
    x = torch.randn(n, ci, h, w, generator=g).cuda()
    wt = (torch.randn(co, ci, 3, 3, generator=g) * (2.0 / (ci * 9)) ** 0.5).cuda()
    sc = (torch.rand(co, generator=g) + 0.5).cuda()
    sh = torch.randn(co, generator=g).cuda()
    ho, wo = (h + 2 * pad - 2 * dil - 1) // st + 1, (w + 2 * pad - 2 * dil - 1) // st + 1
    res = torch.randn(n, co, ho, wo, generator=g).cuda() if with_res else None
    ref = F.conv2d(x.double(), wt.double(), None, st, pad, dil) * sc.double().view(1, -1, 1, 1) + sh.double().view(1, -1, 1, 1)
    if with_res:
        ref = ref + res.double()
    if relu:
        ref = torch.relu(ref)
    y = ops.conv2d_x3(x, wt, sc, sh, ops.ACT_RELU if relu else ops.ACT_NONE, res, pad, dil, st)
    assert y.shape == ref.shape
    err = float((y.double() - ref).abs().max()) / float(ref.abs().max())
    assert err <= 2e-5, err


POINTWISE = [  # (N, Cin, Cout, H, W, residual, relu)
    (5, 64, 256, 96, 72, True, True),       # layer1 conv3 + shortcut sum
    (5, 256, 64, 96, 72, False, True),
    (3, 96, 48, 48, 36, False, False),      # fuse-layer 1x1 on a low-resolution branch
    (7, 384, 96, 12, 9, False, False),      # several images per tile, odd width
    (2, 32, 17, 10, 6, True, False),        # Cout not a multiple of 16, one partial tile
    (40, 192, 192, 24, 18, True, True),
]


@pytest.mark.parametrize("case", POINTWISE, ids=lambda c: "x".join(str(v) for v in c[:5]))
def test_conv2d_x3_pointwise_matches_float64(case):
    n, ci, co, h, w, with_res, relu = case
    g = torch.Generator(device="cpu").manual_seed(sum(case[:5]))
    x = torch.randn(n, ci, h, w, generator=g).cuda()
    wt = (torch.randn(co, ci, 1, 1, generator=g) * (2.0 / ci) ** 0.5).cuda()
    sc = (torch.rand(co, generator=g) + 0.5).cuda()
    sh = torch.randn(co, generator=g).cuda()
    res = torch.randn(n, co, h, w, generator=g).cuda() if with_res else None
    ref = F.conv2d(x.double(), wt.double()) * sc.double().view(1, -1, 1, 1) + sh.double().view(1, -1, 1, 1)
    if with_res:
        ref = ref + res.double()
    if relu:
        ref = torch.relu(ref)
    y = ops.conv2d_x3(x, wt, sc, sh, ops.ACT_RELU if relu else ops.ACT_NONE, res, 0, 1, 1)
    err = float((y.double() - ref).abs().max()) / float(ref.abs().max())
    assert err <= 2e-5, err


def test_conv2d_x3_channel_sliced_views():
    """Input, output and residual as channel windows of wider tensors (the engine's concat-free layout)."""
    g = torch.Generator(device="cpu").manual_seed(5)
    big_in = torch.randn(3, 80, 24, 20, generator=g).cuda()
    big_out = torch.full((3, 70, 24, 20), 7.0).cuda()
    big_res = torch.randn(3, 50, 24, 20, generator=g).cuda()
    wt = (torch.randn(24, 32, 3, 3, generator=g) * 0.1).cuda()
    sh = torch.randn(24, generator=g).cuda()
    iv, ov, rv = ops.View(big_in, 16, 32), ops.View(big_out, 40, 24), ops.View(big_res, 8, 24)
    d = ops.conv_desc(iv, ov, 24, 3, 3, 1, 1, 1, ops.ACT_RELU, None, rv)
    assert ops.x3_supported(d)
    ops.conv2d_x3_launch(iv, ops.pack_x3_weight(wt), sh, ov, d, rv)
    ref = torch.relu(F.conv2d(big_in[:, 16:48].double(), wt.double(), None, 1, 1) + sh.double().view(1, -1, 1, 1)
                     + big_res[:, 8:32].double())
    assert float((big_out[:, 40:64].double() - ref).abs().max()) <= 2e-5 * float(ref.abs().max())
    assert bool((big_out[:, :40] == 7.0).all()) and bool((big_out[:, 64:] == 7.0).all())


def test_conv2d_x3_rejects_what_it_does_not_cover():
    x = torch.zeros(1, 20, 8, 8).cuda()
    out = torch.zeros(1, 16, 8, 8).cuda()
    d = ops.conv_desc(ops.View(x), ops.View(out), 16, 3, 3, 1, 1, 1)
    assert not ops.x3_supported(d)                    # Cin % 16 != 0
    with pytest.raises(ValueError):
        ops.pack_x3_weight(torch.zeros(16, 20, 3, 3).cuda())
    x = torch.zeros(1, 32, 96, 72).cuda()             # dilation 15 at 96x72: the window does not fit the LDS
    out = torch.zeros(1, 16, 96, 72).cuda()
    assert not ops.x3_supported(ops.conv_desc(ops.View(x), ops.View(out), 16, 3, 3, 1, 15, 15))
    x = torch.zeros(1, 16, 9, 7).cuda()               # H*W % 4 != 0
    out = torch.zeros(1, 16, 9, 7).cuda()
    assert not ops.x3_supported(ops.conv_desc(ops.View(x), ops.View(out), 16, 3, 3, 1, 1, 1))


@pytest.mark.parametrize("case", [(16, 6, 6, 96, 72, True), (3, 13, 13, 24, 20, True), (2, 20, 20, 17, 9, False),
                                  (1, 24, 5, 8, 8, True), (2, 3, 24, 33, 70, False)],
                         ids=lambda c: "x".join(str(v) for v in c[:5]))
def test_conv3x3_small_matches_float64(case):
    """csrc/conv_small.hip (RSB staircase convs, model/RSB.py:80-92): exact fp32, pre-added second input, ragged tiles."""
    n, ci, co, h, w, with_in2 = case
    g = torch.Generator(device="cpu").manual_seed(sum(case[:5]))
    x = torch.randn(n, ci, h, w, generator=g).cuda()
    x2 = torch.randn(n, ci, h, w, generator=g).cuda() if with_in2 else None
    wt = (torch.randn(co, ci, 3, 3, generator=g) * (2.0 / (ci * 9)) ** 0.5).cuda()
    sc, sh = (torch.rand(co, generator=g) + 0.5).cuda(), torch.randn(co, generator=g).cuda()
    xin = x.double() + (x2.double() if with_in2 else 0.0)
    ref = torch.relu(F.conv2d(xin, wt.double(), None, 1, 1) * sc.double().view(1, -1, 1, 1) + sh.double().view(1, -1, 1, 1))
    y = ops.conv3x3_small(x, wt, sc, sh, ops.ACT_RELU, x2)
    assert float((y.double() - ref).abs().max()) <= 2e-6 * float(ref.abs().max())


def test_conv2d_x3_full_size_properties():
    """BASELINE configs[1] size (80 frames, 48 ch, 96x72): size-independent checks instead of a float64 reference.
    (a) a delta kernel reproduces its input channel up to the split's remainder (2^-18 relative); (b) linearity in the
    weights; (c) batch slices are independent (frame 17 alone == frame 17 inside the batch, bit for bit)."""
    g = torch.Generator(device="cpu").manual_seed(123)
    x = torch.randn(80, 48, 96, 72, generator=g).cuda()
    delta = torch.zeros(48, 48, 3, 3).cuda()
    for c in range(48):
        delta[c, (c * 7) % 48, 1, 1] = 1.0
    y = ops.conv2d_x3(x, delta)
    perm = [(c * 7) % 48 for c in range(48)]
    assert float((y - x[:, perm]).abs().max()) <= 2.0 ** -17 * float(x.abs().max())
    w1 = (torch.randn(48, 48, 3, 3, generator=g) * 0.05).cuda()
    w2 = (torch.randn(48, 48, 3, 3, generator=g) * 0.05).cuda()
    y1, y2, y12 = ops.conv2d_x3(x, w1), ops.conv2d_x3(x, w2), ops.conv2d_x3(x, w1 + w2)
    assert float((y12 - (y1 + y2)).abs().max()) <= 3e-5 * float(y12.abs().max())
    solo = ops.conv2d_x3(x[17:18].contiguous(), w1)
    assert torch.equal(solo, y1[17:18])
