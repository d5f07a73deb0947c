"""fp16-storage kernels of the backbone (csrc/h16.hip, BASELINE.json configs[4] "fp16") against float64 arithmetic on the SAME
half-rounded operands: what is left is fp32 accumulation order and the one rounding of the result to half, so the bound is one
half ulp of the result plus 2e-6 of the sum's magnitude.  Layers: model/HRNet.py:500-530 (BasicBlock), :551-571 (Bottleneck),
:442-470 (stride-2 chains), :426-439 (fuse 1x1), :118-120 (stem), :487-494 (fuse rows)."""
import pytest
import torch
import torch.nn.functional as F

from otpose_amd import ops
from otpose_amd.ops import ACT_NONE, ACT_RELU, H8, View

pytestmark = pytest.mark.gpu


def _rand(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(shape, generator=g) * scale


def _h(x):
    """round to half, back in float64"""
    return x.half().double()


def _check_half(out, ref, what=""):
    """|out - ref| within one rounding to half of ref (+ fp32 accumulation noise)"""
    out, ref = out.double().cpu(), ref.double().cpu()
    tol = ref.abs() * 2.0 ** -10 + 2.0 ** -24 + 4e-6 * float(ref.abs().max())
    bad = (out - ref).abs() > tol
    assert not bool(bad.any()), f"{what}: {int(bad.sum())} of {bad.numel()} beyond a half ulp, worst {float(((out - ref).abs() / tol).max()):.2f} x tol"


def test_h8_pack_unpack_round_trip_is_the_rounding_to_half():
    x = _rand((3, 24, 7, 12), 1, 3.0).cuda()
    img = ops.h8_pack(x)
    assert torch.equal(ops.h8_unpack(img), x.half().float())
    # a channel slice in, a group slice out
    wide = ops.h8_empty(3, 40, 7, 12, x.device)
    wide.t.zero_()
    ops.h8_pack(View(x, 8, 16), out=wide.slice(16, 16))
    back = ops.h8_unpack(wide)
    assert torch.equal(back[:, 16:32], x[:, 8:24].half().float()) and float(back[:, :16].abs().max()) == 0.0


CONV_SHAPES = [
    # n, cin, cout, h, w, stride, res, act
    (2, 16, 16, 12, 16, 1, False, ACT_RELU),
    (5, 48, 48, 24, 18, 1, True, ACT_RELU),        # tiles straddle images
    (3, 32, 96, 12, 9, 1, False, ACT_NONE),        # two cout blocks, tiny maps (2.4 images per tile)
    (2, 48, 40, 16, 12, 1, True, ACT_RELU),        # a lone cout tile: half records
    (2, 64, 64, 48, 36, 1, True, ACT_RELU),        # NTW = 2
    (4, 48, 48, 96, 72, 1, True, ACT_RELU),        # the dominant layer's map
    (2, 192, 192, 24, 18, 1, True, ACT_RELU),      # 12 chunks through the ring
    (2, 16, 32, 16, 24, 2, False, ACT_RELU),
    (3, 48, 96, 24, 36, 2, True, ACT_RELU),        # stride 2 + residual (a fuse chain's last conv)
    (2, 96, 192, 48, 36, 2, False, ACT_NONE),
    (2, 64, 64, 96, 144, 2, False, ACT_RELU),      # wide rows (the stem's conv2 at reduced height)
    (5, 256, 48, 24, 18, 1, False, ACT_RELU),      # transition1
]


@pytest.mark.parametrize("n,cin,cout,h,w,stride,with_res,act", CONV_SHAPES)
def test_h16_conv3x3_matches_float64_on_the_same_halves(n, cin, cout, h, w, stride, with_res, act):
    x = _rand((n, cin, h, w), 2, 1.5)
    wt = _rand((cout, cin, 3, 3), 3, 0.6 / (cin * 9) ** 0.5)
    sc = torch.rand(cout, generator=torch.Generator().manual_seed(4)) + 0.5
    sh = _rand((cout,), 5, 0.3)
    ho, wo = h // stride, w // stride
    res = _rand((n, cout, ho, wo), 6, 1.0) if with_res else None
    k = ops.h16_weight_exponent(wt, sc)
    xi = ops.h8_pack(x.cuda())
    ri = ops.h8_pack(res.cuda()) if with_res else None
    wp = ops.pack_h16_conv_weight(wt.cuda(), sc.cuda(), k)
    d = ops.h16_conv_desc(xi, cout, stride, act, None, ri, k)
    assert ops.h16_conv_supported(d)
    out = ops.h16_conv3x3(xi, wp, sh.cuda(), cout, stride, act, ri, k=k)
    wq = _h(wt * sc[:, None, None, None] * 2.0 ** k) * 2.0 ** -k
    ref = F.conv2d(_h(x), wq, None, stride, 1) + sh.double()[None, :, None, None]
    if with_res:
        ref = ref + _h(res)
    if act == ACT_RELU:
        ref = ref.clamp_min(0)
    _check_half(ops.h8_unpack(out), ref, f"conv {cin}->{cout} s{stride}")


def test_h16_conv3x3_window_width_and_group_slices_change_nothing(monkeypatch):
    """The window staged 48 channels at a time against 16 at a time (OTPOSE_H16_CK) and smaller pixel tiles (OTPOSE_H16_NPT), and
    input / output / residual as channel-group slices of wider tensors: the same bits."""
    n, cin, cout, h, w = 3, 48, 48, 24, 18
    x, res = _rand((n, cin, h, w), 7).cuda(), _rand((n, cout, h, w), 8).cuda()
    wt, sh = _rand((cout, cin, 3, 3), 9, 0.05).cuda(), _rand((cout,), 10, 0.2).cuda()
    wp = ops.pack_h16_conv_weight(wt, None, 0)
    xi, ri = ops.h8_pack(x), ops.h8_pack(res)
    a = ops.h8_unpack(ops.h16_conv3x3(xi, wp, sh, cout, 1, ACT_RELU, ri))
    for ck, npt in (("16", "4"), ("48", "2"), ("16", "1")):
        monkeypatch.setenv("OTPOSE_H16_CK", ck)
        monkeypatch.setenv("OTPOSE_H16_NPT", npt)
        b = ops.h8_unpack(ops.h16_conv3x3(xi, wp, sh, cout, 1, ACT_RELU, ri))
        assert torch.equal(a, b), (ck, npt)
    monkeypatch.delenv("OTPOSE_H16_CK")
    monkeypatch.delenv("OTPOSE_H16_NPT")
    wide_in, wide_out, wide_res = (ops.h8_empty(n, c, h, w, x.device) for c in (64, 96, 80))
    for t in (wide_in, wide_out, wide_res):
        t.t.zero_()
    ops.h8_pack(x, out=wide_in.slice(16, 48))
    ops.h8_pack(res, out=wide_res.slice(32, 48))
    ops.h16_conv3x3(wide_in.slice(16, 48), wp, sh, cout, 1, ACT_RELU, wide_res.slice(32, 48), out=wide_out.slice(48, 48))
    full = ops.h8_unpack(wide_out)
    assert torch.equal(full[:, 48:], a) and float(full[:, :48].abs().max()) == 0.0


PW_SHAPES = [
    # n, cin, cout, hw (h, w), res, relu, f32out
    (2, 64, 64, (12, 16), False, True, False),
    (3, 256, 64, (24, 18), False, True, False),
    (3, 64, 256, (24, 18), True, True, False),
    (5, 96, 48, (12, 9), False, False, False),      # 540 pixels: a partial last workgroup, tiles straddling images
    (2, 384, 192, (12, 9), False, False, False),
    (2, 192, 96, (24, 18), False, False, False),
    (2, 48, 17, (24, 18), False, False, True),      # final_layer: fp32 NCHW heat-maps, 17 of 32 rows live
    (2, 32, 17, (16, 12), False, False, True),
]


@pytest.mark.parametrize("n,cin,cout,hw,with_res,relu,f32out", PW_SHAPES)
def test_h16_pointwise_matches_float64_on_the_same_halves(n, cin, cout, hw, with_res, relu, f32out):
    h, w = hw
    x = _rand((n, cin, h, w), 11, 1.2)
    wt = _rand((cout, cin), 12, 0.8 / cin ** 0.5)
    sc = torch.rand(cout, generator=torch.Generator().manual_seed(13)) + 0.5
    sh = _rand((cout,), 14, 0.3)
    res = _rand((n, cout, h, w), 15) if with_res else None
    k = ops.h16_weight_exponent(wt, sc)
    pk = ops.pack_h16_pointwise(wt.cuda(), sc.cuda(), sh.cuda(), k)
    xi = ops.h8_pack(x.cuda())
    ri = ops.h8_pack(res.cuda()) if with_res else None
    wq = _h(wt * sc[:, None] * 2.0 ** k) * 2.0 ** -k
    ref = torch.einsum("oc,nchw->nohw", wq, _h(x)) + sh.double()[None, :, None, None]
    if with_res:
        ref = ref + _h(res)
    if relu:
        ref = ref.clamp_min(0)
    if f32out:
        buf = torch.full((n, cout + 3, h, w), 7.0, device="cuda")
        ops.h16_pointwise(xi, pk, cout, relu, ri, out=View(buf, 2, cout), k=k)
        got = buf[:, 2:2 + cout].cpu().double()
        assert float((got - ref).abs().max()) <= 4e-6 * float(ref.abs().max()) + 1e-6
        assert bool((buf[:, :2] == 7.0).all()) and bool((buf[:, 2 + cout:] == 7.0).all())
    else:
        out = ops.h16_pointwise(xi, pk, cout, relu, ri, k=k)
        _check_half(ops.h8_unpack(out), ref, f"pointwise {cin}->{cout}")


@pytest.mark.parametrize("b,frames,h,w", [(2, 5, 32, 48), (1, 7, 64, 32), (3, 5, 36, 20)])
def test_h16_stem_matches_float64_on_the_same_halves(b, frames, h, w):
    clip = _rand((b, 3 * frames, h, w), 16, 1.0)
    wt = _rand((64, 3, 3, 3), 17, 0.3)
    sc = torch.rand(64, generator=torch.Generator().manual_seed(18)) + 0.5
    sh = _rand((64,), 19, 0.2)
    pk = ops.pack_h16_stem(wt.cuda(), sc.cuda(), sh.cuda())
    out = ops.h16_stem(clip.cuda().contiguous(), pk, 64, frames)
    # frames of the clip stacked on the batch axis in the reference's order (model/OTPose.py:317): n = f B + b
    x = torch.cat(clip.split(3, dim=1), 0)
    ws = wt * sc[:, None, None, None]
    kx = ops.h16_weight_exponent(ws)
    ref = F.conv2d(_h(x), _h(ws * 2.0 ** kx) * 2.0 ** -kx, None, 2, 1) + sh.double()[None, :, None, None]
    _check_half(ops.h8_unpack(out), ref.clamp_min(0), "stem")


@pytest.mark.parametrize("nlow", [1, 2, 3])
def test_h16_upsample_add_is_the_fuse_row_tail(nlow):
    n, c, h, w = 2, 48, 16, 24
    res = _rand((n, c, h, w), 20)
    lows = [_rand((n, c, h // f, w // f), 21 + i) for i, f in enumerate((2, 4, 8)[:nlow])]
    out = ops.h16_upsample_add([ops.h8_pack(l.cuda()) for l in lows], (2, 4, 8)[:nlow], ops.h8_pack(res.cuda()), relu=True)
    ref = _h(res)
    for l, f in zip(lows, (2, 4, 8)):
        ref = ref + F.interpolate(_h(l), scale_factor=f, mode="nearest")
    _check_half(ops.h8_unpack(out), ref.clamp_min(0), "upsample_add")


# ---- the temporal encoders' matrix kernels with half operands (otp_*_h1; model/blocks.py:248-254, 400-419) ---------------------------
@pytest.mark.parametrize("C,T", [(136, 256), (136, 6912), (204, 512)])
def test_encoder_kernels_with_half_operands_stay_within_half_rounding_of_the_split_products(C, T):
    """otp_ln_mlp_h1 / otp_dense_h1 / otp_qkv_front_h1 against their split-product (fp32-grade) twins on the same packed weights:
    what separates them is one rounding to half per operand (2^-11 relative, averaged over K = 136 .. 816 products) and the 6e-5
    GELU - a few 1e-3 of the output range at most."""
    B, HID = 2, 4 * C
    g = torch.Generator().manual_seed(31)
    x = torch.randn(B, C, T, generator=g).cuda()
    w1, w2 = (torch.randn(HID, C, 1, generator=g) / C ** 0.5).cuda(), (torch.randn(C, HID, 1, generator=g) / HID ** 0.5).cuda()
    b1 = (torch.randn(HID, generator=g) * 0.1).cuda()
    gam, bet = (torch.rand(C, generator=g) + 0.5).cuda(), (torch.randn(C, generator=g) * 0.1).cuda()
    sc, sh = (torch.rand(C, generator=g) + 0.5).cuda(), (torch.randn(C, generator=g) * 0.1).cuda()
    packed = ops.pack_mlp_x3_weights(w1, b1, w2)
    ref = ops.ln_mlp_x3(x, gam, bet, 1e-5, packed, sc, sh)
    got = ops.ln_mlp_x3(x, gam, bet, 1e-5, ops.pack_mlp_x3_weights(w1, b1, w2, half=True), sc, sh, half=True)
    e_mlp = float((got - ref).abs().max()) / float(ref.abs().max())
    # projections (+ residual)
    wq = (torch.randn(C, C, generator=g) / C ** 0.5).cuda()
    pk = ops.pack_dense_cc(wq, sc, sh, x3=True)
    r = torch.randn(B, C, T, generator=g).cuda()
    d_ref = ops.dense_cc([x], [pk], [r], x3=True)[0]
    d_got = ops.dense_cc([x], [pk], [r], x3=True, half=True)[0]
    # float64 on the same half-rounded operands: exact up to fp32 accumulation
    d64 = (torch.einsum("oc,bct->bot", wq.half().double().cpu(), x.half().double().cpu()) * sc.double().cpu()[None, :, None]
           + sh.double().cpu()[None, :, None] + r.double().cpu())
    assert float((d_got.double().cpu() - d64).abs().max()) <= 3e-6 * float(d64.abs().max())
    e_dense = float((d_got - d_ref).abs().max()) / float(d_ref.abs().max())
    # q / k / v front end
    dws = [(torch.randn(C, 1, 3, generator=g) * 0.5).cuda() for _ in range(3)]
    lns = [((torch.rand(C, generator=g) + 0.5).cuda(), (torch.randn(C, generator=g) * 0.1).cuda()) for _ in range(3)]
    table = ops.pack_qkv_table(dws[0], dws[1], dws[2], lns[0][0], lns[0][1], lns[1][0], lns[1][1], lns[2][0], lns[2][1])
    packs = [ops.pack_dense_cc((torch.randn(C, C, generator=g) / C ** 0.5).cuda(), None, (torch.randn(C, generator=g) * 0.1).cuda(), x3=True)
             for _ in range(3)]
    q_ref = ops.qkv_front(x, table, packs, x3=True)
    q_got = ops.qkv_front(x, table, packs, x3=True, half=True)
    e_qkv = max(float((a - b).abs().max()) / float(b.abs().max()) for a, b in zip(q_got, q_ref))
    print(f"C={C} T={T}: half operands vs split products, fraction of range: mlp {e_mlp:.2e}  dense {e_dense:.2e}  qkv {e_qkv:.2e}")
    assert e_mlp <= 4e-3 and e_dense <= 2e-3 and e_qkv <= 2e-3
