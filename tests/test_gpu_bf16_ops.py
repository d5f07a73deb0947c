"""bf16 NHWC training operators (csrc/nhwc.hip) against plain PyTorch fp32 references of the same ops on the SAME
bf16-rounded operands: the conv products are exact in fp32, so forward outputs differ from the reference only by the
final rounding to bf16 (2^-9 relative) and fp32 summation order; weight gradients are fp32 end to end.
Reference ops: nn.Conv2d / nn.BatchNorm2d(training) / nearest upsample + add as composed in model/HRNet.py:416-571."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

BF = torch.bfloat16


def _dev():
    return torch.device("cuda", 0)


def _nhwc(t):
    """(N, C, H, W) float (values already bf16-representable) -> (N, H, W, CS) bf16 on the GPU, zero padded."""
    from otpose_amd.bf16_ops import cs
    n, c, h, w = t.shape
    out = torch.zeros(n, h, w, cs(c), dtype=BF)
    out[..., :c] = t.permute(0, 2, 3, 1).to(BF)
    return out.to(_dev())


def _nchw(t, c):
    return t[..., :c].float().permute(0, 3, 1, 2).cpu()


def _rb(t):
    return t.to(BF).float()


CONV_CASES = [
    # cin, cout, k, stride, pad, dil, n, h, w
    (48, 48, 3, 1, 1, 1, 3, 24, 18),
    (3, 64, 3, 2, 1, 1, 2, 32, 24),
    (64, 256, 1, 1, 0, 1, 2, 16, 12),
    (96, 48, 1, 1, 0, 1, 2, 12, 9),
    (48, 96, 3, 2, 1, 1, 2, 24, 18),
    (384, 384, 3, 1, 1, 1, 2, 12, 9),
    (192, 192, 3, 1, 1, 1, 2, 24, 18),
    (48, 17, 1, 1, 0, 1, 2, 24, 18),
    (32, 306, 3, 1, 6, 6, 1, 24, 18),
    (40, 24, 3, 1, 1, 1, 1, 7, 5),
    (64, 64, 3, 1, 1, 1, 1, 96, 72),
    (48, 48, 3, 1, 1, 1, 1, 96, 72),
    (256, 64, 1, 1, 0, 1, 1, 96, 72),
    (32, 153, 3, 1, 15, 15, 1, 96, 72),      # widely dilated offset conv on the full map: tap-mode wgrad, 8-channel chunks
    (64, 64, 3, 2, 1, 1, 1, 192, 144),       # second stem conv: strided wide map
]


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_forward_stats_dgrad_wgrad(case):
    from otpose_amd import bf16_ops as B
    cin, cout, k, stride, pad, dil, n, h, w = case
    g = torch.Generator().manual_seed(hash(case) % 1000)
    x = _rb(torch.randn(n, cin, h, w, generator=g))
    wt = _rb(torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5)
    ref = F.conv2d(x.double(), wt.double(), None, stride, pad, dil).float()
    xd, wd = _nhwc(x), wt.to(_dev())
    out, stats, rows = B.conv_forward(xd, wd, None, stride, pad, dil)
    torch.cuda.synchronize()
    got = _nchw(out, cout)
    tol = 2.0 ** -8 * ref.abs().max()
    assert float((got - ref).abs().max()) <= tol, float((got - ref).abs().max())
    # padding channels are zero, statistics are those of the rounded outputs
    assert float(out[..., cout:].float().abs().max() if out.shape[-1] > cout else 0.0) == 0.0
    s = stats.sum(0).cpu()
    o = out.float().cpu().reshape(-1, out.shape[-1])
    assert torch.allclose(s[0], o.sum(0), rtol=1e-4, atol=1e-2)
    assert torch.allclose(s[1], (o * o).sum(0), rtol=1e-4, atol=1e-2)
    assert rows == stats.shape[0]
    # fp32 NCHW output mode with bias
    bias = torch.randn(cout, generator=g)
    o32, _, _ = B.conv_forward(xd, wd, bias.to(_dev()), stride, pad, dil, out_mode=1)
    ref32 = ref + bias.view(1, -1, 1, 1)
    assert float((o32.cpu() - ref32).abs().max()) <= 1e-4 * max(1.0, float(ref32.abs().max()))
    # gradients
    gy = _rb(torch.randn(ref.shape, generator=g))
    xr, wr = x.double().requires_grad_(), wt.double().requires_grad_()
    F.conv2d(xr, wr, None, stride, pad, dil).backward(gy.double())
    gyd = _nhwc(gy)
    gx = _nchw(B.conv_dgrad(gyd, wd, (h, w), stride, pad, dil), cin)
    gxr = xr.grad.float()
    assert float((gx - gxr).abs().max()) <= 2.0 ** -8 * gxr.abs().max(), float((gx - gxr).abs().max())
    gw = B.conv_wgrad(xd, gyd, tuple(wt.shape), stride, pad, dil).cpu()
    gwr = wr.grad.float()
    assert float((gw - gwr).abs().max()) <= 2e-5 * max(1.0, float(gwr.abs().max())), float((gw - gwr).abs().max())


@pytest.mark.parametrize("case", [(48, 48, 1, 5, 96, 72), (96, 96, 1, 3, 48, 36), (192, 192, 1, 3, 24, 18), (384, 384, 1, 5, 12, 9),
                                  (64, 64, 2, 2, 192, 144), (48, 96, 2, 3, 96, 72), (96, 40, 1, 2, 20, 14), (32, 56, 2, 3, 16, 12)])
def test_window_kernel_matches_the_chunked_kernel(case, monkeypatch):
    """3x3 / pad 1 convolutions with Cin % 16 == 0 run on csrc/hb.hip (the fp16 engine's window + weight-stream kernel compiled for
    bfloat16 NHWC tensors); OTPOSE_NHWC_HB=0 keeps nhwc_conv_kernel.  Same operands, same products, fp32 sums in another order:
    the bf16 results may differ by one rounding step, the per-tile statistics must add up to the same channel sums, and both sit
    within the rounding of a float64 convolution."""
    from otpose_amd import bf16_ops as B
    from otpose_amd import hip
    cin, cout, stride, n, h, w = case
    g = torch.Generator().manual_seed(sum(case))
    x = _rb(torch.randn(n, cin, h, w, generator=g))
    wt = _rb(torch.randn(cout, cin, 3, 3, generator=g) / (cin * 9) ** 0.5)
    bias = torch.randn(cout, generator=g).to(_dev())
    ref = (F.conv2d(x.double(), wt.double(), None, stride, 1, 1) + bias.cpu().double().view(1, -1, 1, 1)).float()
    xd, wd = _nhwc(x), wt.to(_dev())
    d = B._desc(n, h, w, cin, cout, 3, 3, stride, 1, 1, 0)
    import ctypes
    res = {}
    for hb in ("1", "0"):
        monkeypatch.setenv("OTPOSE_NHWC_HB", "2" if hb == "1" else "0")       # 2: also the shapes where it does not pay
        rows = hip.lib().otp_nhwc_conv_stats_rows(ctypes.byref(d))
        out, stats, r2 = B.conv_forward(xd, wd, bias, stride, 1, 1)
        torch.cuda.synchronize()
        assert rows == r2 == stats.shape[0]
        res[hb] = (out.float().cpu(), stats.sum(0).cpu())
    monkeypatch.delenv("OTPOSE_NHWC_HB")
    a, b = res["1"][0], res["0"][0]
    assert res["1"][0].shape == res["0"][0].shape
    got = a.permute(0, 3, 1, 2)
    tol = 2.0 ** -8 * float(ref.abs().max())
    assert float((got - ref).abs().max()) <= tol
    assert float((a - b).abs().max()) <= 2.0 ** -7 * float(ref.abs().max())
    assert float((a != b).float().mean()) < 0.05                 # (rounding flips only)
    o = a.reshape(-1, a.shape[-1])
    assert torch.allclose(res["1"][1][0], o.sum(0), rtol=1e-4, atol=1e-2)
    assert torch.allclose(res["1"][1][1], (o * o).sum(0), rtol=1e-4, atol=1e-2)


@pytest.mark.parametrize("case", [(136, 544, 2, 6912), (544, 136, 2, 6912), (136, 136, 3, 200), (40, 24, 1, 77), (204, 816, 1, 1000),
                                  (816, 204, 1, 1000)])
def test_sequence_pointwise_convs_forward_and_input_gradient(case, monkeypatch):
    """The two projections of a TransformerBlock MLP (model/blocks.py:248-254) on the (B, 1, T, C) view of a sequence: csrc/hb.hip's
    pointwise kernel (register-resident input, streamed weights) behind otp_nhwc_conv_* - bf16 NHWC and fp32 NCHW results, bias,
    the input gradient with and without a skip gradient - against float64 and against nhwc_conv_kernel (OTPOSE_NHWC_HB=0)."""
    from otpose_amd import bf16_ops as B
    cin, cout, n, t = case
    g = torch.Generator().manual_seed(sum(case))
    x = _rb(torch.randn(n, cin, 1, t, generator=g))
    wt = _rb(torch.randn(cout, cin, 1, 1, generator=g) / cin ** 0.5)
    bias = torch.randn(cout, generator=g)
    ref = F.conv2d(x.double(), wt.double(), bias.double()).float()
    gy = _rb(torch.randn(n, cout, 1, t, generator=g))
    res = _rb(torch.randn(n, cin, 1, t, generator=g))
    gref = torch.nn.grad.conv2d_input(x.shape, wt.double(), gy.double()).float()
    xd, wd, bd, gyd, resd = _nhwc(x), wt.to(_dev()), bias.to(_dev()), _nhwc(gy), _nhwc(res)
    got = {}
    for hb in ("1", "0"):
        monkeypatch.setenv("OTPOSE_NHWC_HB", hb)
        o16, _, _ = B.conv_forward(xd, wd, bd, 1, 0, 1, out_mode=0, want_stats=False)
        o32, _, _ = B.conv_forward(xd, wd, bd, 1, 0, 1, out_mode=1)
        gx = B.conv_dgrad(gyd, wd, (1, t), 1, 0, 1)
        gxr = B.conv_dgrad(gyd, wd, (1, t), 1, 0, 1, res=resd)
        monkeypatch.setenv("OTPOSE_WGRAD1X1", hb)                # (the co-group 1x1 weight-gradient kernel against the general one)
        gw = B.conv_wgrad(xd, gyd, tuple(wt.shape), 1, 0, 1).cpu()
        torch.cuda.synchronize()
        assert torch.equal(gxr, gx + resd)
        gwr = torch.nn.grad.conv2d_weight(x.double(), wt.shape, gy.double()).float()
        assert float((gw - gwr).abs().max()) <= 2e-5 * max(1.0, float(gwr.abs().max())), hb
        got[hb] = (_nchw(o16, cout), o32.cpu(), _nchw(gx, cin))
    monkeypatch.delenv("OTPOSE_NHWC_HB")
    monkeypatch.delenv("OTPOSE_WGRAD1X1")
    for hb in ("1", "0"):
        o16, o32, gx = got[hb]
        assert float((o16 - ref).abs().max()) <= 2.0 ** -8 * float(ref.abs().max()), hb
        assert float((o32 - ref).abs().max()) <= 1e-4 * max(1.0, float(ref.abs().max())), hb
        assert float((gx - gref).abs().max()) <= 2.0 ** -8 * float(gref.abs().max()), hb
    assert float((got["1"][1] - got["0"][1]).abs().max()) <= 2e-4 * max(1.0, float(ref.abs().max()))


@pytest.mark.parametrize("c,relu,with_res", [(48, True, True), (96, True, False), (17, False, False), (256, False, True)])
def test_conv_bn_function_matches_torch_autograd(c, relu, with_res):
    """ConvBnFunction forward / backward vs conv2d + batch_norm(training) + residual + relu on fp64 autograd."""
    from otpose_amd import bf16_ops as B
    g = torch.Generator().manual_seed(c)
    n, cin, h, w = 3, 48, 12, 9
    x = _rb(torch.randn(n, cin, h, w, generator=g))
    wt = _rb(torch.randn(c, cin, 3, 3, generator=g) / (cin * 9) ** 0.5)
    gamma = torch.rand(c, generator=g) + 0.5
    beta = torch.randn(c, generator=g) * 0.2
    res = _rb(torch.randn(n, c, h, w, generator=g)) if with_res else None
    gy = _rb(torch.randn(n, c, h, w, generator=g))
    rm, rv = torch.zeros(c), torch.ones(c)
    # reference in fp64, with the conv output rounded to bf16 where the kernel rounds it (BatchNorm normalises the stored tensor)
    xr, wr = x.double().requires_grad_(), wt.double().requires_grad_()
    gr, br = gamma.double().requires_grad_(), beta.double().requires_grad_()
    rr = res.double().requires_grad_() if with_res else None
    conv = F.conv2d(xr, wr, None, 1, 1)
    conv_q = conv + (conv.detach().to(BF).double() - conv.detach())          # straight-through rounding
    rmr, rvr = rm.double().clone(), rv.double().clone()
    y = F.batch_norm(conv_q, rmr, rvr, gr, br, True, 0.1, 1e-5)
    if with_res:
        y = y + rr
    if relu:
        y = torch.relu(y)
    y.backward(gy.double())
    dev = _dev()
    xd = _nhwc(x).requires_grad_()
    wd = wt.to(dev).requires_grad_()
    gd, bd = gamma.to(dev).requires_grad_(), beta.to(dev).requires_grad_()
    rd = _nhwc(res).requires_grad_() if with_res else None
    rmd, rvd = rm.to(dev), rv.to(dev)
    out = B.conv_bn(xd, wd, gd, bd, rd, rmd, rvd, 1, 1, relu)
    out.backward(_nhwc(gy))
    torch.cuda.synchronize()
    scale = float(y.detach().abs().max())
    assert float((_nchw(out.detach(), c) - y.detach().float()).abs().max()) <= 2.0 ** -7 * scale
    assert torch.allclose(rmd.cpu(), rmr.float(), atol=1e-4) and torch.allclose(rvd.cpu(), rvr.float(), rtol=1e-3, atol=1e-4)

    def close(a, b, rel):
        return float((a - b).abs().max()) <= rel * max(float(b.abs().max()), 1e-6)

    # input / residual gradients pass through bf16 tensors (gc is rounded before the dgrad conv)
    assert close(_nchw(xd.grad, cin), xr.grad.float(), 3e-2)
    if with_res:
        assert close(_nchw(rd.grad, c), rr.grad.float(), 1e-2)
    assert close(wd.grad.cpu(), wr.grad.float(), 3e-2)
    assert close(gd.grad.cpu(), gr.grad.float(), 2e-2)
    assert close(bd.grad.cpu(), br.grad.float(), 2e-2)


@pytest.mark.parametrize("f,relu", [(2, True), (4, False), (8, True)])
def test_upsample_add_forward_backward(f, relu):
    from otpose_amd import bf16_ops as B
    g = torch.Generator().manual_seed(f)
    n, c, hl, wl = 2, 48, 3, 2
    low = _rb(torch.randn(n, c, hl, wl, generator=g))
    res = _rb(torch.randn(n, c, hl * f, wl * f, generator=g))
    gy = _rb(torch.randn(n, c, hl * f, wl * f, generator=g))
    lr, rr = low.double().requires_grad_(), res.double().requires_grad_()
    y = rr + F.interpolate(lr, scale_factor=f, mode="nearest")
    if relu:
        y = torch.relu(y)
    y.backward(gy.double())
    ld, rd = _nhwc(low).requires_grad_(), _nhwc(res).requires_grad_()
    out = B.upsample_add(ld, rd, f, relu)
    out.backward(_nhwc(gy))
    torch.cuda.synchronize()
    assert float((_nchw(out.detach(), c) - y.detach().float()).abs().max()) <= 2.0 ** -8 * float(y.detach().abs().max())
    assert float((_nchw(rd.grad, c) - rr.grad.float()).abs().max()) == 0.0
    assert float((_nchw(ld.grad, c) - lr.grad.float()).abs().max()) <= 2.0 ** -8 * float(lr.grad.abs().max())


def test_layout_converters_round_trip_and_frame_split():
    from otpose_amd import bf16_ops as B
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 15, 8, 6, generator=g)
    d = B.to_nhwc(x.to(_dev()), frame_split=2)                      # (10, 8, 6, 8): frames on the batch axis, C 3 -> 8
    ref = torch.cat(x.split(3, dim=1), 0)                           # model/OTPose.py:317
    assert d.shape == (10, 8, 6, 8)
    assert torch.equal(_nchw(d, 3), _rb(ref)) and float(d[..., 3:].float().abs().max()) == 0.0
    y = torch.randn(3, 17, 5, 4, generator=g)
    n = B.to_nhwc(y.to(_dev()))
    assert n.shape == (3, 5, 4, 24)
    assert torch.equal(B.to_nchw(n, 17).cpu(), _rb(y))


def test_mlp_interior_matches_fp32_reference():
    """Conv1d(C, 4C, 1) -> GELU -> Conv1d(4C, C, 1) (model/blocks.py:248-254) through the bf16 NHWC kernels on a (B, C, T)
    sequence vs fp64 autograd: output and every gradient."""
    from otpose_amd import bf16_ops as B
    g = torch.Generator().manual_seed(9)
    b, c, t = 2, 136, 864
    x = _rb(torch.randn(b, c, t, generator=g))
    w1 = _rb(torch.randn(4 * c, c, 1, generator=g) / c ** 0.5)
    b1 = torch.randn(4 * c, generator=g) * 0.1
    w2 = _rb(torch.randn(c, 4 * c, 1, generator=g) / (4 * c) ** 0.5)
    b2 = torch.randn(c, generator=g) * 0.1
    gy = torch.randn(b, c, t, generator=g)
    xr, w1r, b1r, w2r, b2r = (v.double().requires_grad_() for v in (x, w1, b1, w2, b2))
    ref = F.conv1d(F.gelu(F.conv1d(xr, w1r, b1r)), w2r, b2r)
    ref.backward(gy.double())
    dev = _dev()
    xd, w1d, b1d, w2d, b2d = (v.to(dev).requires_grad_() for v in (x, w1, b1, w2, b2))
    h = B.gelu(B.conv_bias(B.to_nhwc_grad(xd.unsqueeze(2)), w1d.unsqueeze(-1), b1d))
    out = B.conv_out(h, w2d.unsqueeze(-1), b2d).squeeze(2)
    out.backward(gy.to(dev))
    torch.cuda.synchronize()

    def close(a, r, rel):
        err = float((a.detach().cpu().double() - r).abs().max())
        assert err <= rel * float(r.abs().max()), (err, float(r.abs().max()))

    close(out, ref.detach(), 2e-2)          # two bf16 roundings of the hidden activation
    close(xd.grad, xr.grad, 3e-2)
    close(w1d.grad, w1r.grad, 3e-2)
    close(w2d.grad, w2r.grad, 2e-2)
    close(b1d.grad, b1r.grad, 2e-2)
    close(b2d.grad, b2r.grad, 1e-4)


@pytest.mark.parametrize("case", [(3, 48, 48, 24, 18), (2, 96, 96, 12, 9), (2, 20, 44, 10, 7)])
def test_conv_dgrad_with_skip_gradient_is_the_separate_add(case):
    """otp_nhwc_conv_bf16_res: the input-gradient conv with the skip connection's gradient added in its epilogue equals
    the plain launch followed by a bf16 tensor add, bit for bit (the conv result is rounded to bf16 before the add in both)."""
    from otpose_amd import bf16_ops as B
    from tests.conftest import seeded
    n, cin, cout, h, w = case
    wt = (seeded((cout, cin, 3, 3), 1) * (2.0 / (cin * 9)) ** 0.5).to(_dev())
    gy = _nhwc(_rb(seeded((n, cout, h, w), 2)))
    res = _nhwc(_rb(seeded((n, cin, h, w), 3)))
    plain = B.conv_dgrad(gy, wt, (h, w), 1, 1, 1)
    fused = B.conv_dgrad(gy, wt, (h, w), 1, 1, 1, res=res)
    assert torch.equal(fused, plain + res)


def test_gelu_dropout_is_gelu_then_dropout_with_repeatable_draws():
    """otp_gelu_dropout_bf16_*: nn.GELU -> nn.Dropout(p) of the TransformerBlock MLP (model/blocks.py:250-251) in one pass: kept
    elements carry gelu(x) / (1 - p) (one bf16 rounding), the rest are zero, the keep rate is 1 - p, the gradient passes exactly where
    the forward kept, a seed repeats its draws and PyTorch's generator seeds them."""
    from otpose_amd import bf16_ops as B
    g = torch.Generator().manual_seed(3)
    x = (torch.randn(4, 1, 1000, 544, generator=g) * 1.5).to(BF).to(_dev())
    p = 0.1
    xf = x.float().requires_grad_()
    ref = F.gelu(xf) / (1 - 6554 / 65536)
    xs = x.clone().requires_grad_()
    y = B.gelu_dropout(xs, p, seed=1234)
    live = x != 0                                    # (gelu(0) = 0: whether such an element was kept cannot be read off y)
    kept = (y != 0) | ~live
    rate = float(kept.float().mean())
    n = x.numel()
    assert abs(rate - (1 - 6554 / 65536)) < 5 * (p * (1 - p) / n) ** 0.5 + 2e-3, rate        # (gelu(x) == 0 only at x == 0)
    err = (y.float() - ref.detach())[kept].abs() / ref.detach()[kept].abs().clamp_min(1e-3)
    assert float(err.max()) <= 2.0 ** -8
    # per-channel and per-position rates: no structure along either axis
    assert float((kept.float().mean((0, 1, 2)) - rate).abs().max()) < 0.03 and float((kept.float().mean((0, 1, 3)) - rate).abs().max()) < 0.08
    gy = torch.randn(x.shape, generator=g).to(BF).to(_dev())
    y.backward(gy)
    ref.backward(gy.float() * kept)
    gerr = ((xs.grad.float() - xf.grad) * live).abs().max() / xf.grad.abs().max()
    assert float(gerr) <= 2.0 ** -7
    # (dropped elements pass nothing; a kept x < -5.9 also has y = -0 in fp32 arithmetic - erff = -1 - and a gradient of ~1e-7)
    assert float(xs.grad[~kept].float().abs().max()) <= 1e-5
    assert torch.equal(B.gelu_dropout(x, p, seed=1234), y.detach()) and not torch.equal(B.gelu_dropout(x, p, seed=1235), y.detach())
    torch.manual_seed(7)
    a1, a2 = B.gelu_dropout(x, p), B.gelu_dropout(x, p)
    torch.manual_seed(7)
    b1 = B.gelu_dropout(x, p)
    assert torch.equal(a1, b1) and not torch.equal(a1, a2)
    assert torch.equal(B.gelu_dropout(x, 0.0), B.gelu(x))


@pytest.mark.parametrize("case", [(136, 544, 2, 6912, 0.1), (136, 544, 3, 200, 0.0), (40, 160, 1, 77, 0.25)])
def test_mlp_interior_node_matches_the_three_node_chain(case):
    """MlpInteriorFunction (GELU / dropout and their derivatives inside the projections' launches, otp_nhwc_mlp_*) against
    conv_bias -> gelu_dropout -> conv_out with the SAME dropout seed: the same keep decisions, results and gradients within the bf16
    rounding of the hidden tensor (the fused epilogues use a 6e-5 fit of the normal CDF and round once where the chain rounds twice)."""
    from otpose_amd import bf16_ops as B
    c, hid, n, t, p = case
    g = torch.Generator().manual_seed(sum(int(v) for v in case[:4]))
    x = _nhwc(_rb(torch.randn(n, c, 1, t, generator=g)))
    w1 = (torch.randn(hid, c, 1, 1, generator=g) / c ** 0.5).to(_dev())
    b1 = (torch.randn(hid, generator=g) * 0.1).to(_dev())
    w2 = (torch.randn(c, hid, 1, 1, generator=g) / hid ** 0.5).to(_dev())
    b2 = (torch.randn(c, generator=g) * 0.1).to(_dev())
    go = torch.randn(n, c, 1, t, generator=g).to(_dev())
    assert B.mlp_interior_supported(x, w1, w2)
    res = []
    for fused in (True, False):
        leaves = [v.clone().requires_grad_() for v in (w1, b1, w2, b2)]
        xi = x.clone().requires_grad_()
        if fused:
            o = B.mlp_interior(xi, *leaves, p, seed=99)
        else:
            o = B.conv_out(B.gelu_dropout(B.conv_bias(xi, leaves[0], leaves[1]), p, seed=99), leaves[2], leaves[3])
        o.backward(go)
        torch.cuda.synchronize()
        res.append([o.detach().float()] + [xi.grad.float()] + [v.grad.float() for v in leaves])
    names = ("out", "grad x", "grad w1", "grad b1", "grad w2", "grad b2")
    for nm, a, b in zip(names, *res):
        err = float((a - b).norm() / b.norm())
        assert err <= 6e-3, (nm, err)              # (2^-9 rounding of the 4C-wide hidden tensor on both sides)


def test_basic_block_node_matches_two_conv_bn_nodes():
    """BasicBlockFunction against the two ConvBnFunction nodes it replaces (model/HRNet.py:500-531): same launches, so
    the output, the running statistics and every gradient agree exactly - dL/dx included, where the fused node adds the skip
    gradient inside the input-gradient conv instead of leaving the sum to autograd."""
    from otpose_amd import bf16_ops as B
    from tests.conftest import seeded
    n, c, h, w = 3, 48, 24, 18
    x0 = _nhwc(_rb(seeded((n, c, h, w), 1)))
    go = _nhwc(_rb(seeded((n, c, h, w), 9)))
    mk = lambda s_, shape, k=1.0: (seeded(shape, s_) * k).to(_dev())                       # noqa: E731
    base = dict(w1=mk(2, (c, c, 3, 3), 0.05), g1=1 + mk(3, (c,), 0.1), b1=mk(4, (c,), 0.1),
                w2=mk(5, (c, c, 3, 3), 0.05), g2=1 + mk(6, (c,), 0.1), b2=mk(7, (c,), 0.1))
    res = {}
    for mode in ("nodes", "block"):
        x = x0.clone().requires_grad_()
        P = {k: v.clone().requires_grad_() for k, v in base.items()}
        st = [torch.zeros(c, device=_dev()), torch.ones(c, device=_dev()), torch.zeros(c, device=_dev()),
              torch.ones(c, device=_dev())]
        if mode == "nodes":
            y1 = B.conv_bn(x, P["w1"], P["g1"], P["b1"], None, st[0], st[1], 1, 1, True)
            y = B.conv_bn(y1, P["w2"], P["g2"], P["b2"], x, st[2], st[3], 1, 1, True)
        else:
            y = B.basic_block(x, P["w1"], P["g1"], P["b1"], st[0], st[1], P["w2"], P["g2"], P["b2"], st[2], st[3])
        y.backward(go)
        res[mode] = [y.detach(), x.grad] + [P[k].grad for k in sorted(P)] + st
    for a, b in zip(res["nodes"], res["block"]):
        assert torch.equal(a, b)
