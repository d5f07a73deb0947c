"""Seeded random-shape sweep of the convolution family through the C-ABI: forward (direct and Winograd kernels, fused
epilogue), input gradient and weight gradient vs torch on the CPU.  Shapes are drawn around the kernels' tiling edges (ragged
channel chunks / M tiles, odd sizes, widths that are and are not multiples of 4, maps smaller than a tile block)."""
import math
import random

import pytest
import torch
import torch.nn.functional as F

from otpose_amd import ops
from tests.conftest import seeded

pytestmark = pytest.mark.gpu


def _close(a, b, tol):
    a = a.detach().cpu()
    assert a.shape == b.shape
    err = float((a - b).abs().max())
    assert err <= tol * max(1.0, float(b.abs().max())), f"max abs err {err} (ref max {float(b.abs().max())})"


def _cases(n, seed):
    rng = random.Random(seed)
    out = []
    for _ in range(n):
        k = rng.choice((1, 3, 3))
        stride = rng.choice((1, 1, 1, 2))
        dil = 1 if k == 1 else rng.choice((1, 1, 1, 2, 3))
        pad = 0 if k == 1 else rng.choice((dil, dil, 0, 1))
        h, w = rng.choice((3, 5, 8, 9, 12, 16, 17, 24, 26, 32, 48)), rng.choice((4, 6, 7, 8, 9, 12, 13, 16, 18, 20, 24, 36, 40, 72))
        if (h + 2 * pad - (dil * (k - 1) + 1)) < 0 or (w + 2 * pad - (dil * (k - 1) + 1)) < 0:
            continue
        out.append((rng.randint(1, 3), rng.choice((3, 8, 17, 20, 32, 40, 48, 64, 70)), h, w,
                    rng.choice((4, 16, 17, 33, 48, 50, 96, 100, 150)), k, stride, pad, dil))
    return out


@pytest.mark.parametrize("case", _cases(40, 2024) + _cases(40, 7))
def test_conv_forward_and_gradients_random_shapes(case):
    from otpose_amd import train_ops as T
    n, cin, h, w, cout, k, stride, pad, dil = case
    x = seeded((n, cin, h, w), 1).requires_grad_()
    wt = seeded((cout, cin, k, k), 2, 1.0 / math.sqrt(cin * k * k)).requires_grad_()
    b = seeded((cout,), 3).requires_grad_()
    ref = F.conv2d(x, wt, b, stride, pad, dil)
    go = seeded(ref.shape, 4)
    ref.backward(go)
    # inference-path conv with the fused epilogue
    sc = 1 + 0.1 * seeded((cout,), 5)
    res = seeded(ref.shape, 6)
    out = ops.conv2d(x.detach().cuda(), wt.detach().cuda(), sc.cuda(), b.detach().cuda(), stride, pad, dil, act=ops.ACT_RELU,
                     res=res.cuda())
    _close(out, F.relu(F.conv2d(x.detach(), wt.detach(), None, stride, pad, dil) * sc.view(1, -1, 1, 1)
                       + b.detach().view(1, -1, 1, 1) + res), 3e-5)
    if k == 3 and stride == 1 and pad == 1 and dil == 1 and (h * w) % 4 == 0:
        _close(ops.conv2d_wino(x.detach().cuda(), wt.detach().cuda(), sc.cuda(), b.detach().cuda(), act=ops.ACT_RELU,
                               res=res.cuda()),
               F.relu(F.conv2d(x.detach(), wt.detach(), None, 1, 1) * sc.view(1, -1, 1, 1) + b.detach().view(1, -1, 1, 1) + res),
               3e-5)
    # training path: forward + all three gradients
    xs, ws, bs = (t.detach().cuda().requires_grad_() for t in (x, wt, b))
    o2 = T.conv2d(xs, ws, bs, stride, pad, dil)
    _close(o2, ref.detach(), 3e-5)
    o2.backward(go.cuda())
    _close(xs.grad, x.grad, 5e-5)
    _close(ws.grad, wt.grad, 1e-4)
    _close(bs.grad, b.grad, 1e-4)
