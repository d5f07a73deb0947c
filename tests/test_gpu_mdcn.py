"""HIP modulated DCN (through the C ABI) vs the CPU oracle on the same seeded inputs."""
import pytest
import torch

from oracle import mdcn_scalar as S
from oracle import otpose_oracle as O
from otpose_amd import ops
from tests.conftest import seeded

pytestmark = pytest.mark.gpu
TOL = 2e-5   # fp32, values O(1-10): absolute + relative slack for summation order / fma contraction

CASES = [  # N, C, H, W, Co, k, stride, pad, dil, groups, dg
    (2, 17, 12, 9, 17, 3, 1, 3, 3, 1, 17),      # OTPose shape family (fast path: 3x3, Cout 17)
    (2, 17, 24, 18, 17, 3, 1, 15, 15, 1, 17),   # dilation 15: most taps start outside the image
    (1, 17, 64, 48, 17, 3, 1, 6, 6, 1, 17),     # cfg1 heat-map size, several pixel tiles
    (3, 6, 7, 5, 4, 3, 1, 2, 2, 1, 3),          # generic Cout chunk, W % 4 != 0, dg < C
    (1, 4, 9, 8, 6, 3, 2, 1, 1, 2, 2),          # stride 2, conv groups 2
    (1, 2, 5, 5, 3, 1, 1, 0, 1, 1, 1),          # 1x1 kernel (runtime-tap path)
    (1, 20, 10, 12, 40, 3, 1, 1, 1, 1, 4),      # Cout > 16: several output-channel passes
]


def _inputs(case, off_scale=3.0):
    N, C, H, W, Co, k, stride, pad, dil, groups, dg = case
    Ho = (H + 2 * pad - (dil * (k - 1) + 1)) // stride + 1
    Wo = (W + 2 * pad - (dil * (k - 1) + 1)) // stride + 1
    x = seeded((N, C, H, W), 1)
    off = seeded((N, dg * 2 * k * k, Ho, Wo), 2, off_scale)
    m = seeded((N, dg * k * k, Ho, Wo), 3)
    w = seeded((Co, C // groups, k, k), 4, 0.3)
    b = seeded((Co,), 5)
    return x, off, m, w, b


def _close(a, b, tol=TOL):
    a = a.detach().cpu()
    err = float((a - b).abs().max())
    assert err <= tol * max(1.0, float(b.abs().max())), f"max abs err {err}"


@pytest.mark.parametrize("case", CASES)
def test_forward_matches_oracle(case):
    x, off, m, w, b = _inputs(case)
    a = case[6:]
    ref = O.mdcn_forward(x, off, m, w, b, *a)
    ref_c = S.forward(x, off, m, w, b, *a)
    dev = "cuda"
    out = ops.modulated_deform_conv(x.to(dev), off.to(dev), m.to(dev), w.to(dev), b.to(dev), *a)
    _close(out, ref)
    _close(out, ref_c)
    out_nb = ops.modulated_deform_conv(x.to(dev), off.to(dev), m.to(dev), w.to(dev), None, *a)
    _close(out_nb, O.mdcn_forward(x, off, m, w, None, *a))


def test_forward_special_offsets():
    """integer offsets (exact corner hits), offsets landing exactly on -1 / H (outside, open interval),
    huge and NaN offsets (sample dropped) - kernel.cu:556 and :403-432."""
    case = (1, 17, 12, 9, 17, 3, 1, 3, 3, 1, 17)
    x, off, m, w, b = _inputs(case)
    off = torch.round(off)
    off[0, 0::7] = 1e9
    off[0, 3, 2, :] = -1e9
    a = case[6:]
    ref = O.mdcn_forward(x, off, m, w, b, *a)
    out = ops.modulated_deform_conv(x.cuda(), off.cuda(), m.cuda(), w.cuda(), b.cuda(), *a)
    _close(out, ref)
    offn = off.clone()
    offn[0, 5, 1, 1] = float("nan")
    out = ops.modulated_deform_conv(x.cuda(), offn.cuda(), m.cuda(), w.cuda(), b.cuda(), *a)
    assert torch.isfinite(out).all()


def test_alpha_beta_weighted_sum():
    """out = beta*out + alpha*dcn, the fused form of the reference's 0.2 * sum over dilations."""
    from otpose_amd import hip
    case = (2, 17, 12, 9, 17, 3, 1, 3, 3, 1, 17)
    x, off, m, w, b = _inputs(case)
    ref = O.mdcn_forward(x, off, m, w, b, *case[6:])
    acc0 = seeded(ref.shape, 9)
    xs, offs, ms, ws_, bs = (t.cuda() for t in (x, off, m, w, b))
    out = acc0.cuda().clone()
    st = hip.lib().otp_mdcn_forward(hip.ptr(xs), hip.ptr(offs), hip.ptr(ms), hip.ptr(ws_), hip.ptr(bs), hip.ptr(out),
                                    2, 17, 12, 9, 17, 3, 3, 1, 3, 3, 1, 17, 0.2, 1.0, 0, hip.stream_of(xs))
    assert st == 0
    _close(out, acc0 + 0.2 * ref)


@pytest.mark.parametrize("case", CASES[:3] + [(2, 8, 10, 12, 6, 3, 1, 2, 2, 1, 2), (1, 4, 9, 8, 6, 3, 2, 1, 1, 2, 2)])
def test_backward_matches_oracle(case):
    x, off, m, w, b = _inputs(case)
    a = case[6:]
    ref_out = O.mdcn_forward(x, off, m, w, b, *a)
    go = seeded(ref_out.shape, 6)
    ref = O.mdcn_backward(x, off, m, w, go, *a)
    ts = [t.cuda().requires_grad_() for t in (x, off, m, w, b)]
    out = ops.modulated_deform_conv(ts[0], ts[1], ts[2], ts[3], ts[4], *a)
    out.backward(go.cuda())
    names = ("grad_x", "grad_offset", "grad_mask", "grad_weight", "grad_bias")
    for name, t, r in zip(names, ts, ref):
        err = float((t.grad.cpu() - r).abs().max())
        assert err <= 1e-4 * max(1.0, float(r.abs().max())), f"{name}: {err}"


def test_backward_full_size_chunks_and_atomics():
    """96x72 planes: several pixel chunks per plane (grad_x merged with float atomics)."""
    case = (2, 17, 96, 72, 17, 3, 1, 9, 9, 1, 17)
    x, off, m, w, b = _inputs(case)
    a = case[6:]
    go = seeded((2, 17, 96, 72), 6)
    ref = O.mdcn_backward(x, off, m, w, go, *a)
    ts = [t.cuda().requires_grad_() for t in (x, off, m, w, b)]
    ops.modulated_deform_conv(*ts, *a).backward(go.cuda())
    for t, r in zip(ts, ref):
        err = float((t.grad.cpu() - r).abs().max())
        assert err <= 2e-4 * max(1.0, float(r.abs().max())), err


def test_cpu_tensors_raise_like_reference():
    x, off, m, w, b = _inputs(CASES[0])
    with pytest.raises(NotImplementedError):
        ops.modulated_deform_conv(x, off, m, w, b, 1, 3, 3, 1, 17)


def test_full_size_properties():
    """BASELINE size (16 x 17 x 96 x 72): zero offsets + unit mask == dilated conv2d (known answer),
    linearity in the mask, and batch-slice consistency with the small-size oracle run."""
    torch.manual_seed(0)
    dev = "cuda"
    x = torch.randn(16, 17, 96, 72, device=dev)
    w = torch.randn(17, 17, 3, 3, device=dev) * 0.2
    b = torch.randn(17, device=dev)
    off0 = torch.zeros(16, 306, 96, 72, device=dev)
    m1 = torch.ones(16, 153, 96, 72, device=dev)
    for d in (3, 15):
        out = ops.modulated_deform_conv(x, off0, m1, w, b, 1, d, d, 1, 17)
        ref = torch.nn.functional.conv2d(x.cpu(), w.cpu(), b.cpu(), 1, d, d)
        _close(out, ref, 5e-5)
    off = torch.randn(16, 306, 96, 72, device=dev) * 3
    m = torch.randn(16, 153, 96, 72, device=dev)
    o1 = ops.modulated_deform_conv(x, off, m, w, None, 1, 6, 6, 1, 17)
    o2 = ops.modulated_deform_conv(x, off, 2.5 * m, w, None, 1, 6, 6, 1, 17)
    assert float((o2 - 2.5 * o1).abs().max()) <= 1e-4 * float(o1.abs().max())
    ref = O.mdcn_forward(x[5:6].cpu(), off[5:6].cpu(), m[5:6].cpu(), w.cpu(), None, 1, 6, 6, 1, 17)
    _close(o1[5:6], ref, 5e-5)


@pytest.mark.parametrize("case", [CASES[0], CASES[3], CASES[4]])
def test_dcn_v1_matches_oracle(case):
    """DCN v1 (deform_conv_forward_cuda / backward_input / backward_parameters, deform_conv_cuda.cpp:148-472): forward
    and all three gradients vs the oracle's v1 restatement under autograd; the in-place pybind-style entry points
    including the ``scale`` accumulation of backward_parameters."""
    x, off, _, w, _ = _inputs(case)
    N, C, H, W, Co, k, stride, pad, dil, groups, dg = case
    xr, offr, wr = (t.clone().requires_grad_() for t in (x, off, w))
    ref = O.dcn_v1_forward(xr, offr, wr, stride, pad, dil, groups, dg)
    go = seeded(ref.shape, 9)
    ref.backward(go)
    xs, offs, ws = (t.cuda().requires_grad_() for t in (x, off, w))
    out = ops.deform_conv(xs, offs, ws, stride, pad, dil, groups, dg)
    _close(out, ref.detach())
    out.backward(go.cuda())
    _close(xs.grad, xr.grad, 1e-4)
    _close(offs.grad, offr.grad, 1e-4)
    _close(ws.grad, wr.grad, 1e-4)
    # pybind-style calls
    o2 = torch.empty_like(out)
    assert ops.deform_conv_forward_cuda(xs.detach(), ws.detach(), offs.detach(), o2, None, None, k, k, stride, stride,
                                        pad, pad, dil, dil, groups, dg, 1) == 1
    assert torch.equal(o2, out.detach())
    gi, goff = torch.zeros_like(xs), torch.zeros_like(offs)
    ops.deform_conv_backward_input_cuda(xs.detach(), offs.detach(), go.cuda(), gi, goff, ws.detach(), None, k, k, stride,
                                        stride, pad, pad, dil, dil, groups, dg, 1)
    _close(gi, xr.grad, 1e-4)
    _close(goff, offr.grad, 1e-4)
    gw = torch.ones_like(ws)
    ops.deform_conv_backward_parameters_cuda(xs.detach(), offs.detach(), go.cuda(), gw, None, None, k, k, stride, stride,
                                             pad, pad, dil, dil, groups, dg, 0.5, 1)
    _close(gw, 1.0 + 0.5 * wr.grad, 1e-4)


# ---- the general operator (csrc/mdcn_generic.hip): every case the reference's entry points accept -------------------
GENERAL = [  # N, C, H, W, Co, kh, kw, (sh, sw), (ph, pw), (dh, dw), groups, dg
    (1, 20, 10, 12, 40, 3, 3, (1, 1), (1, 1), (1, 1), 1, 4),        # backward with Cout 40 (three 16-channel passes)
    (2, 8, 9, 7, 64, 3, 3, (1, 1), (2, 2), (2, 2), 2, 2),           # Cout 64, conv groups 2
    (2, 6, 8, 9, 5, 1, 1, (1, 1), (0, 0), (1, 1), 1, 3),            # 1x1 kernel backward
    (1, 4, 11, 10, 6, 5, 5, (1, 1), (2, 2), (1, 1), 1, 2),          # 5x5 kernel backward
    (1, 6, 12, 10, 4, 3, 3, (2, 1), (1, 2), (1, 2), 1, 3),          # independent stride / padding / dilation per axis
    (1, 4, 9, 9, 4, 3, 2, (1, 1), (1, 0), (1, 1), 1, 1),            # non-square kernel
]


def _general_inputs(case, dtype=torch.float64):
    N, C, H, W, Co, kh, kw, (sh, sw), (ph, pw), (dh, dw), groups, dg = case
    Ho = (H + 2 * ph - (dh * (kh - 1) + 1)) // sh + 1
    Wo = (W + 2 * pw - (dw * (kw - 1) + 1)) // sw + 1
    x = seeded((N, C, H, W), 1).to(dtype)
    off = seeded((N, dg * 2 * kh * kw, Ho, Wo), 2, 2.0).to(dtype)
    m = seeded((N, dg * kh * kw, Ho, Wo), 3).to(dtype)
    w = seeded((Co, C // groups, kh, kw), 4, 0.3).to(dtype)
    b = seeded((Co,), 5).to(dtype)
    go = seeded((N, Co, Ho, Wo), 6).to(dtype)
    return x, off, m, w, b, go


def _run_general(case, tensors, with_mask=True):
    """Forward + backward through the pybind-named entry points (full per-axis argument lists, cpp:474-480, 551-558)."""
    N, C, H, W, Co, kh, kw, (sh, sw), (ph, pw), (dh, dw), groups, dg = case
    x, off, m, w, b, go = (t.cuda() for t in tensors)
    out = torch.empty_like(go)
    mm = m if with_mask else None
    ops.modulated_deform_conv_cuda_forward(x, w, b if with_mask else None, None, off, mm, out, None, kh, kw, sh, sw, ph, pw,
                                           dh, dw, groups, dg, with_mask)
    gx, goff, gm = torch.empty_like(x), torch.empty_like(off), (torch.empty_like(m) if with_mask else None)
    gw, gb = torch.zeros_like(w), torch.zeros_like(b)
    ops.modulated_deform_conv_cuda_backward(x, w, b if with_mask else None, None, off, mm, None, gx, gw,
                                            gb if with_mask else None, goff, gm, go, kh, kw, sh, sw, ph, pw, dh, dw, groups, dg,
                                            with_mask)
    return out, gx, goff, gm, gw, gb


def _reference_general(case, tensors, with_mask=True):
    N, C, H, W, Co, kh, kw, s_, p_, d_, groups, dg = case
    x, off, m, w, b, go = (t.double().clone() for t in tensors)
    for t in (x, off, m, w, b):
        t.requires_grad_()
    mm = m if with_mask else torch.ones_like(m)
    out = O.mdcn_forward(x, off, mm, w, b if with_mask else None, s_, p_, d_, groups, dg)
    out.backward(go)
    return out.detach(), x.grad, off.grad, m.grad if with_mask else None, w.grad, b.grad if with_mask else None


@pytest.mark.parametrize("case", GENERAL)
def test_general_operator_fp32_matches_oracle_autograd(case):
    t = _general_inputs(case, torch.float32)
    got = _run_general(case, t)
    ref = _reference_general(case, t)
    for g_, r_ in zip(got, ref):
        _close(g_.double().cpu(), r_, 1e-4)


@pytest.mark.parametrize("dtype,tol", [(torch.float64, 1e-10), (torch.float16, 2e-2), (torch.bfloat16, 8e-2)])
def test_general_operator_dtypes(dtype, tol):
    """The storage types of the reference's dispatch (f64 / f32 / f16, .cu:719,751,784) and bf16: same inputs rounded to the
    storage type on both sides, fp64 oracle; the tolerance is the storage type's rounding of the OUTPUT tensors."""
    case = GENERAL[0]
    t = tuple(v.to(dtype) for v in _general_inputs(case, torch.float64))
    got = _run_general(case, t)
    ref = _reference_general(case, t)
    for g_, r_ in zip(got, ref):
        assert g_.dtype == dtype
        _close(g_.double().cpu(), r_, tol)


def test_dcn_v1_runs_without_a_mask_stream():
    """mask = None: no modulation tensor is read or written (deform_conv_cuda.cpp:148-472), any kernel / Cout."""
    case = GENERAL[3]
    t = _general_inputs(case, torch.float32)
    got = _run_general(case, t, with_mask=False)
    ref = _reference_general(case, t, with_mask=False)
    for g_, r_ in zip(got, ref):
        if r_ is not None:
            _close(g_.double().cpu(), r_, 1e-4)
