"""Range guard of the half-piece ("f16x3") and fp16 arithmetic (csrc/range.hip, csrc/common.h): the reference computes these
layers in fp32 (model/HRNet.py:500-530, model/blocks.py:248-254) and has no 65504 limit, so crossing it must be LOUD - a Python
exception and NaN heat-maps - and everything inside the limit must stay within the engine's usual bounds of the oracle.
Never a silent inf, never a NaN swallowed by a ReLU."""
import pytest
import torch

from otpose_amd import OTPose, cfg1, hip, ops, tiny_cfg
from otpose_amd import synthetic as S

pytestmark = [pytest.mark.gpu, pytest.mark.range_overflow_expected]
NAMES = ("output", "rough", "intersection", "prev_b", "context", "squeezed", "total_b")


def _flag(reset=True):
    torch.cuda.synchronize()
    return hip.lib().otp_range_flag_read(1 if reset else 0)


def _model(cfg):
    m = OTPose(cfg)
    S.fill_synthetic_(m)
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    return m.cuda().eval(), sd


def _oracle(sd, cfg, x, margin):
    from oracle import otpose_oracle as O
    with torch.no_grad():
        return O.otpose_forward(sd, cfg, x, margin)


def _within_usual_bounds(outs, ref):
    for n, o, r in zip(NAMES, outs, ref):
        o = o.cpu()
        assert bool(torch.isfinite(o).all()), n
        err = float((o - r).abs().max())
        assert err <= 1e-3 * max(1.0, float(r.abs().max())), f"{n}: {err}"


@pytest.mark.parametrize("scale", [1e-4, 1.0, 1e4])
def test_scaled_input_is_right_or_raises_never_silently_wrong(scale):
    """cfg1 input x 1e-4 / x 1e4: either the 7 outputs are finite and within the engine's usual 1e-3 of the
    oracle, or the range guard raises AND the heat-maps are NaN - the third outcome (finite but wrong, or inf) must not exist."""
    cfg = cfg1()
    m, sd = _model(cfg)
    x, margin = S.synthetic_clip(1, cfg.MODEL.IMAGE_SIZE)
    x = x * scale
    _flag()
    m.alias_outputs = True
    with torch.no_grad():
        outs = m(x.cuda(), margin=margin.cuda())
    try:
        m.check_range()
    except FloatingPointError as e:
        assert "65504" in str(e)
        assert bool(torch.isnan(outs[0]).all()), "a flagged forward must leave NaN heat-maps, not numbers"
        assert scale > 1.0, "only the up-scaled input may overflow"
        return
    _within_usual_bounds(outs, _oracle(sd, cfg, x, margin))


def test_hot_batchnorm_channel_is_right_or_raises():
    """A BatchNorm channel with running_var = 1e-8 (scale 316 x gamma): a real checkpoint's hot channel."""
    cfg = cfg1()
    m, _ = _model(cfg)
    bn = m.rough_pose_estimation_net.stage2[0].branches[0][0].bn1
    with torch.no_grad():
        bn.running_var[3] = 1e-8
    sd = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    x, margin = S.synthetic_clip(1, cfg.MODEL.IMAGE_SIZE)
    _flag()
    m.alias_outputs = True
    with torch.no_grad():
        outs = m(x.cuda(), margin=margin.cuda())
    try:
        m.check_range()
    except FloatingPointError:
        assert bool(torch.isnan(outs[0]).all())
        return
    _within_usual_bounds(outs, _oracle(sd, cfg, x, margin))


def test_overflowing_input_raises_in_sync_mode_and_the_next_forward_is_clean(monkeypatch):
    """An image value beyond a half's range: OTPOSE_RANGE_CHECK=sync raises inside the forward that overflowed; after the
    exception the flag is clear and a sane forward on the same engine gives the usual numbers."""
    monkeypatch.setenv("OTPOSE_RANGE_CHECK", "sync")
    cfg = tiny_cfg(8, (64, 96))
    m, sd = _model(cfg)
    x, margin = S.synthetic_clip(2, cfg.MODEL.IMAGE_SIZE)
    bad = x.clone()
    bad[1, 4, 10, 10] = 1.0e5
    _flag()
    with torch.no_grad():
        good = [o.clone() for o in m(x.cuda(), margin=margin.cuda())]
        with pytest.raises(FloatingPointError):
            m(bad.cuda(), margin=margin.cuda())
        assert _flag(reset=False) == 0
        again = m(x.cuda(), margin=margin.cuda())
    for a, b in zip(good, again):
        assert torch.equal(a, b)
    _within_usual_bounds(good, _oracle(sd, cfg, x, margin))


def test_deferred_mode_raises_at_the_next_forward_and_poisons_the_heat_maps():
    cfg = tiny_cfg(8, (64, 96))
    m, _ = _model(cfg)
    m.alias_outputs = True
    x, margin = S.synthetic_clip(2, cfg.MODEL.IMAGE_SIZE)
    bad = x.clone()
    bad[0, 0, 5, 5] = -3.0e5
    _flag()
    with torch.no_grad():
        outs = m(bad.cuda(), margin=margin.cuda())             # no exception yet: the host has not synchronised
        torch.cuda.synchronize()
        assert bool(torch.isnan(outs[0]).all())
        with pytest.raises(FloatingPointError):
            m(x.cuda(), margin=margin.cuda())                  # ... the next call does
        outs = m(x.cuda(), margin=margin.cuda())
        torch.cuda.synchronize()
        assert bool(torch.isfinite(outs[0]).all())


def test_s8_conv_flags_an_activation_beyond_65504_and_a_relu_cannot_hide_it():
    """csrc/convs.hip alone: a 7e4 activation splits to hi = inf, lo = -inf; the sums are NaN, the ReLU turns them into 0 -
    the guard word is what remains of the overflow."""
    n, c, h, w = 2, 32, 12, 16
    g = torch.Generator().manual_seed(5)
    x = torch.randn(n, c, h, w, generator=g).cuda()
    wt = (torch.randn(c, c, 3, 3, generator=g) * 0.05).cuda()
    sh = torch.zeros(c, device="cuda")
    d = ops.s8_conv_desc(n, c, c, h, w, ops.ACT_RELU)
    wp = ops.pack_s8_weight(wt, None, 0)
    y = ops.s8_empty(n, c, h, w, x.device)
    _flag()
    ops.conv3x3_s8_launch(ops.s8_pack(x), wp, sh, d, None, None, ops.S8_F32_C4, y)
    assert _flag() == 0
    x[1, 7, 3, 3] = 7.0e4
    xs = ops.s8_pack(x)
    assert _flag() == 6                                          # the pack pass saw it (OTP_RANGE_S8PASS)
    ops.conv3x3_s8_launch(xs, wp, sh, d, None, None, ops.S8_F32_C4, y)
    assert _flag() == 2                                          # ... and the conv's NaN sums (OTP_RANGE_CONVS)
    print("values of the result that are still non-finite after the ReLU:", int((~torch.isfinite(ops.s8_unpack(y, n, c, h, w))).sum()))


def test_mlp_and_projection_kernels_flag_an_operand_beyond_65504():
    B, C, HID, T = 1, 136, 544, 256
    g = torch.Generator().manual_seed(6)
    x = torch.randn(B, C, T, generator=g).cuda()
    w1, w2 = (torch.randn(HID, C, 1, generator=g) / C ** 0.5).cuda(), (torch.randn(C, HID, 1, generator=g) / HID ** 0.5).cuda()
    b1, one, zero = torch.zeros(HID, device="cuda"), torch.ones(C, device="cuda"), torch.zeros(C, device="cuda")
    packed = ops.pack_mlp_x3_weights(w1, b1, w2)
    _flag()
    ops.mlp_x3(x, packed, one, zero, x)
    assert _flag() == 0
    x[0, 5, 17] = 1.0e5
    out = ops.mlp_x3(x, packed, one, zero, x)
    assert _flag() == 7
    assert not bool(torch.isfinite(out).all())
