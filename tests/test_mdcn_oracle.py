"""The two independent CPU restatements of the modulated DCN agree with each other, with autograd
and with the known answer "zero offsets + unit mask == dilated conv2d" (DCN arithmetic has no
executable reference here - see oracle/otpose_oracle.py header)."""
import pytest
import torch
import torch.nn.functional as F

from oracle import mdcn_scalar as S
from oracle import otpose_oracle as O
from tests.conftest import seeded

CASES = [  # N, C, H, W, Co, k, stride, pad, dil, groups, dg
    (2, 6, 7, 5, 4, 3, 1, 2, 2, 1, 3),
    (1, 4, 9, 8, 6, 3, 2, 1, 1, 2, 2),
    (2, 17, 12, 9, 17, 3, 1, 3, 3, 1, 17),
    (1, 17, 16, 12, 17, 3, 1, 15, 15, 1, 17),   # dilation 15 > image size: most taps start outside
    (1, 2, 5, 5, 3, 1, 1, 0, 1, 1, 1),          # 1x1 kernel
]


def _inputs(case, dtype, off_scale=3.0):
    N, C, H, W, Co, k, stride, pad, dil, groups, dg = case
    Ho = (H + 2 * pad - (dil * (k - 1) + 1)) // stride + 1
    Wo = (W + 2 * pad - (dil * (k - 1) + 1)) // stride + 1
    x = seeded((N, C, H, W), 1, dtype=dtype)
    off = seeded((N, dg * 2 * k * k, Ho, Wo), 2, off_scale, dtype)
    m = seeded((N, dg * k * k, Ho, Wo), 3, dtype=dtype)
    w = seeded((Co, C // groups, k, k), 4, dtype=dtype)
    b = seeded((Co,), 5, dtype=dtype)
    return x, off, m, w, b


@pytest.mark.parametrize("case", CASES)
def test_scalar_c_matches_vectorised_fp64(case):
    x, off, m, w, b = _inputs(case, torch.float64)
    a = case[6:]
    o1 = O.mdcn_forward(x, off, m, w, b, *a)
    o2 = S.forward(x, off, m, w, b, *a)
    assert float((o1 - o2).abs().max()) < 1e-12
    go = seeded(o1.shape, 6, dtype=torch.float64)
    g1 = O.mdcn_backward(x, off, m, w, go, *a)
    g2 = S.backward(x, off, m, w, go, *a)
    for u, v in zip(g1, g2):
        assert float((u - v).abs().max()) < 1e-11


@pytest.mark.parametrize("case", CASES[:3])
def test_analytic_backward_matches_autograd_fp64(case):
    ts = [t.clone().requires_grad_() for t in _inputs(case, torch.float64)]
    a = case[6:]
    out = O.mdcn_forward(*ts, *a)
    go = seeded(out.shape, 6, dtype=torch.float64)
    out.backward(go)
    g = O.mdcn_backward(*[t.detach() for t in ts[:4]], go, *a)
    for u, t in zip(g, ts):
        assert float((u - t.grad).abs().max()) < 1e-11


def test_fp32_scalar_vs_vectorised():
    case = CASES[2]
    x, off, m, w, b = _inputs(case, torch.float32)
    a = case[6:]
    assert float((O.mdcn_forward(x, off, m, w, b, *a) - S.forward(x, off, m, w, b, *a)).abs().max()) < 2e-5


@pytest.mark.parametrize("dil", [1, 2, 3])
def test_zero_offset_unit_mask_is_conv2d(dil):
    x = seeded((2, 6, 9, 7), 1, dtype=torch.float64)
    w = seeded((4, 6, 3, 3), 2, dtype=torch.float64)
    b = seeded((4,), 3, dtype=torch.float64)
    off = torch.zeros(2, 3 * 18, 9, 7, dtype=torch.float64)
    m = torch.ones(2, 3 * 9, 9, 7, dtype=torch.float64)
    ref = F.conv2d(x, w, b, 1, dil, dil)
    assert float((O.mdcn_forward(x, off, m, w, b, 1, dil, dil, 1, 3) - ref).abs().max()) < 1e-12
    assert float((S.forward(x, off, m, w, b, 1, dil, dil, 1, 3) - ref).abs().max()) < 1e-12


def test_integer_offsets_shift_and_border_rule():
    """offset (+1, -2) on every tap == conv2d of the shifted, zero-padded image; samples at exactly
    -1 or H are outside ((-1,H) is open, kernel.cu:556) and contribute 0."""
    x = seeded((1, 2, 6, 6), 1, dtype=torch.float64)
    w = seeded((2, 2, 3, 3), 2, dtype=torch.float64)
    off = torch.zeros(1, 18, 6, 6, dtype=torch.float64)
    off[:, 0::2] = 1.0
    off[:, 1::2] = -2.0
    m = torch.ones(1, 9, 6, 6, dtype=torch.float64)
    xp = F.pad(x, (3, 3, 3, 3))                   # tap (i, j) of pixel (y, x) reads x[y + i, x - 3 + j]
    ref = F.conv2d(xp, w)[:, :, 3:9, 0:6]
    got = O.mdcn_forward(x, off, m, w, None, 1, 1, 1, 1, 1)
    assert float((got - ref).abs().max()) < 1e-12
