"""Training building blocks through the C-ABI vs torch autograd on the CPU (fp32): conv dgrad / wgrad / bias grad for
every conv flavour of the path, BatchNorm2d in training mode fused with residual + ReLU, and a BasicBlock composed of
them (reference model/HRNet.py:514-530 under model.train())."""
import math

import pytest
import torch
import torch.nn.functional as F

from tests.conftest import seeded

pytestmark = pytest.mark.gpu


def _close(a, b, tol=3e-5):
    a = a.detach().cpu()
    assert a.shape == b.shape
    err = float((a - b).abs().max())
    assert err <= tol * max(1.0, float(b.abs().max())), f"max abs err {err} (ref max {float(b.abs().max())})"


CASES = [  # N, Cin, H, W, Cout, k, stride, pad, dil
    (2, 48, 24, 18, 48, 3, 1, 1, 1),
    (3, 96, 12, 9, 96, 3, 1, 1, 1),
    (2, 64, 16, 12, 256, 1, 1, 0, 1),
    (2, 20, 12, 9, 100, 3, 1, 1, 1),      # several co / ci groups with ragged tails
    (2, 48, 24, 18, 96, 3, 2, 1, 1),      # fuse down-sampling
    (2, 3, 32, 24, 64, 3, 2, 1, 1),       # stem
    (1, 32, 24, 18, 40, 3, 1, 6, 6),      # dilated offset conv
    (2, 13, 7, 5, 13, 3, 1, 1, 1),        # odd everything
    (2, 136, 1, 700, 136, 1, 1, 0, 1),    # Conv1d(k=1) of the ConvTransformers as (B, C, 1, T): column-tiled rows
    (1, 24, 5, 300, 40, 3, 1, 1, 1),      # wide image: column tiles with halo
    (2, 32, 16, 24, 34, 3, 1, 15, 15),    # dilation 15 on a small map (sparse row staging)
    (1, 3, 24, 288, 20, 3, 2, 1, 1),      # full-width 384x288 stem rows: strided rows split into column tiles
    (1, 20, 12, 144, 24, 3, 2, 1, 1),     # second stem conv width
    (2, 136, 1, 1240, 544, 1, 1, 0, 1),   # MLP up-projection: four 144-row blocks of dW, ragged last pixel chunk
    (2, 544, 1, 500, 136, 1, 1, 0, 1),    # MLP down-projection: four 144-column blocks
    (3, 96, 12, 9, 48, 1, 1, 0, 1),       # fuse-layer 1x1 on a small map (one chunk per image)
    (2, 40, 7, 5, 24, 1, 1, 0, 1),        # H * W % 4 != 0: stays on the generic wgrad kernel
]


@pytest.mark.parametrize("case", CASES)
def test_conv2d_backward_matches_autograd(case):
    from otpose_amd import train_ops as T
    n, cin, h, w, cout, k, stride, pad, dil = case
    x = seeded((n, cin, h, w), 1).requires_grad_()
    wt = seeded((cout, cin, k, k), 2, 1.0 / math.sqrt(cin * k * k)).requires_grad_()
    b = seeded((cout,), 3).requires_grad_()
    ref = F.conv2d(x, wt, b, stride, pad, dil)
    go = seeded(ref.shape, 4)
    ref.backward(go)
    xs, ws, bs = (t.detach().cuda().requires_grad_() for t in (x, wt, b))
    out = T.conv2d(xs, ws, bs, stride, pad, dil)
    _close(out, ref.detach())
    out.backward(go.cuda())
    _close(xs.grad, x.grad)
    _close(ws.grad, wt.grad, 1e-4)
    _close(bs.grad, b.grad, 1e-4)


@pytest.mark.parametrize("relu,with_res", [(True, True), (True, False), (False, False)])
def test_batch_norm_train_matches_autograd(relu, with_res):
    from otpose_amd import train_ops as T
    n, c, h, w = 4, 37, 12, 9
    x = (seeded((n, c, h, w), 1) * 2 + 0.5).requires_grad_()
    g = (1 + 0.1 * seeded((c,), 2)).requires_grad_()
    b = seeded((c,), 3).requires_grad_()
    res = seeded((n, c, h, w), 4).requires_grad_() if with_res else None
    rm, rv = seeded((c,), 5, 0.1), seeded((c,), 6).abs() + 0.5
    rm_ref, rv_ref = rm.clone(), rv.clone()
    y = F.batch_norm(x, rm_ref, rv_ref, g, b, True, 0.1, 1e-5)
    if with_res:
        y = y + res
    if relu:
        y = F.relu(y)
    go = seeded(y.shape, 7)
    y.backward(go)
    xs, gs, bs = (t.detach().cuda().requires_grad_() for t in (x, g, b))
    rs = res.detach().cuda().requires_grad_() if with_res else None
    rmg, rvg = rm.cuda(), rv.cuda()
    out = T.batch_norm_relu(xs, gs, bs, rs, rmg, rvg, 0.1, 1e-5, relu)
    _close(out, y.detach())
    _close(rmg, rm_ref)
    _close(rvg, rv_ref)
    out.backward(go.cuda())
    _close(xs.grad, x.grad, 1e-4)
    _close(gs.grad, g.grad, 1e-4)
    _close(bs.grad, b.grad, 1e-4)
    if with_res:
        _close(rs.grad, res.grad)


def test_basic_block_train_step_matches_autograd():
    """conv-bn-relu-conv-bn-(+x)-relu with batch statistics: output and every gradient (model/HRNet.py:514-530)."""
    from otpose_amd import train_ops as T
    n, c, h, w = 3, 48, 24, 18
    x = seeded((n, c, h, w), 1).requires_grad_()
    w1 = seeded((c, c, 3, 3), 2, 0.05).requires_grad_()
    w2 = seeded((c, c, 3, 3), 3, 0.05).requires_grad_()
    g1, b1 = (1 + 0.1 * seeded((c,), 4)).requires_grad_(), seeded((c,), 5, 0.1).requires_grad_()
    g2, b2 = (1 + 0.1 * seeded((c,), 6)).requires_grad_(), seeded((c,), 7, 0.1).requires_grad_()
    leaves = [x, w1, w2, g1, b1, g2, b2]

    def block(conv, bn, t):
        xx, a, bb, gg1, bb1, gg2, bb2 = t
        y = bn(conv(xx, a), gg1, bb1, None, True)
        return bn(conv(y, bb), gg2, bb2, xx, True)

    ref = block(lambda t, ww: F.conv2d(t, ww, None, 1, 1),
                lambda t, gg, bb, r, relu: F.relu(F.batch_norm(t, None, None, gg, bb, True, 0.1, 1e-5) + (r if r is not None else 0)),
                leaves)
    go = seeded(ref.shape, 8)
    ref.backward(go)
    dev = [t.detach().cuda().requires_grad_() for t in leaves]
    out = block(lambda t, ww: T.conv2d(t, ww, None, 1, 1, 1),
                lambda t, gg, bb, r, relu: T.batch_norm_relu(t, gg, bb, r, None, None, 0.1, 1e-5, relu), dev)
    _close(out, ref.detach(), 1e-4)
    out.backward(go.cuda())
    for a, b in zip(dev, leaves):
        _close(a.grad, b.grad, 2e-4)


def test_full_size_wgrad_properties():
    """BASELINE-size layer (80 x 48 x 96 x 72, 3x3): the weight gradient is linear in grad_out and matches torch on a
    batch slice (the sum over the batch is checked by splitting it in two halves)."""
    from otpose_amd import train_ops as T
    torch.manual_seed(0)
    x = torch.randn(16, 48, 96, 72, device="cuda")
    go = torch.randn(16, 48, 96, 72, device="cuda")
    gw = T.conv2d_grad_weight(x, go, (48, 48, 3, 3), 1, 1, 1)
    halves = T.conv2d_grad_weight(x[:8], go[:8], (48, 48, 3, 3), 1, 1, 1) + \
        T.conv2d_grad_weight(x[8:], go[8:], (48, 48, 3, 3), 1, 1, 1)
    assert float((gw - halves).abs().max()) <= 1e-3 * float(gw.abs().max())
    xc = x[:1].cpu().requires_grad_(False)
    wc = torch.zeros(48, 48, 3, 3, requires_grad=True)
    F.conv2d(xc, wc, None, 1, 1).backward(go[:1].cpu())
    _close(T.conv2d_grad_weight(x[:1], go[:1], (48, 48, 3, 3), 1, 1, 1), wc.grad, 2e-4)


# ---- ConvTransformer pieces vs torch autograd -----------------------------------------------------------------
def _grads(fn, inputs, go):
    leaves = [t.clone().requires_grad_() for t in inputs]
    out = fn(*leaves)
    out.backward(go)
    return out.detach(), [t.grad for t in leaves]


def _check_op(ref_fn, hip_fn, inputs, tol=1e-4):
    out_ref = ref_fn(*inputs)
    go = seeded(out_ref.shape, 99)
    ref_out, ref_g = _grads(ref_fn, inputs, go)
    out, g = _grads(hip_fn, [t.cuda() for t in inputs], go.cuda())
    _close(out, ref_out, tol)
    for a, b in zip(g, ref_g):
        _close(a, b, tol)


@pytest.mark.parametrize("c,t", [(136, 300), (17, 77), (150, 70), (136, 1728), (136, 1100), (40, 1024)])
def test_layer_norm_backward(c, t):
    from otpose_amd import train_ops as T
    x, g, b = seeded((3, c, t), 1) * 2 + 0.3, 1 + 0.1 * seeded((1, c, 1), 2), seeded((1, c, 1), 3)

    def ref(x, g, b):
        mu = x.mean(1, keepdim=True)
        r = x - mu
        return r / torch.sqrt((r ** 2).mean(1, keepdim=True) + 1e-5) * g + b
    _check_op(ref, lambda x, g, b: T.layer_norm(x, g, b, 1e-5), [x, g, b])


@pytest.mark.parametrize("stride", [1, 2])
def test_dwconv3_backward(stride):
    from otpose_amd import train_ops as T
    x, w = seeded((2, 40, 101), 1), seeded((40, 1, 3), 2, 0.5)
    _check_op(lambda x, w: F.conv1d(x, w, None, stride, 1, 1, 40), lambda x, w: T.dwconv3(x, w, stride), [x, w])


@pytest.mark.parametrize("c,t,stride,bias", [(136, 700, 1, True), (17, 300, 1, False), (40, 101, 2, True)])
def test_attn_front_recompute_matches_autograd(c, t, stride, bias):
    """AttnFrontFunction (ln1 -> per q / k / v: dwconv3 -> LayerNorm -> Conv1d(k=1), intermediates rebuilt in the
    backward) against the same composition under torch autograd on the CPU (model/blocks.py:400-440)."""
    from otpose_amd import train_ops as T
    x = seeded((2, c, t), 1) * 2 + 0.3
    ln1 = [1 + 0.1 * seeded((1, c, 1), 2), seeded((1, c, 1), 3)]
    br = []
    for i in range(3):
        br += [seeded((c, 1, 3), 10 + i, 0.5), 1 + 0.1 * seeded((1, c, 1), 20 + i), seeded((1, c, 1), 30 + i),
               seeded((c, c, 1, 1), 40 + i, c ** -0.5)]
        if bias:
            br.append(seeded((c,), 50 + i))
    nb = 5 if bias else 4

    def ln(x, g, b):
        mu = x.mean(1, keepdim=True)
        r = x - mu
        return r / torch.sqrt((r ** 2).mean(1, keepdim=True) + 1e-5) * g + b

    def ref(x, g1, b1, *bp):
        xn = ln(x, g1, b1)
        outs = []
        for i in range(3):
            q = bp[nb * i: nb * i + nb]
            z = ln(F.conv1d(xn, q[0], None, stride, 1, 1, c), q[1], q[2])
            outs.append(F.conv1d(z, q[3].squeeze(-1), q[4] if bias else None))
        return torch.stack(outs)

    def ours(x, g1, b1, *bp):
        branches = [tuple(bp[nb * i: nb * i + nb]) + (() if bias else (None,)) for i in range(3)]
        return torch.stack(T.attn_front(x, stride, 1e-5, (g1, b1), branches))

    _check_op(ref, ours, [x] + ln1 + br, 2e-4)


def test_gelu_maxpool_upsample_backward():
    from otpose_amd import train_ops as T
    x = seeded((2, 17, 96), 1) * 2
    _check_op(F.gelu, T.gelu, [x])
    _check_op(lambda t: F.max_pool1d(t, 3, 2, 1), T.maxpool3s2, [x])
    _check_op(lambda t: F.max_pool1d(t, 3, 2, 1), T.maxpool3s2, [seeded((2, 5, 77), 2)])
    for f in (2, 4):
        _check_op(lambda t: F.interpolate(t, scale_factor=f, mode="linear", align_corners=False),
                  lambda t: T.upsample_linear(t, f), [seeded((2, 9, 60), 3)])


@pytest.mark.parametrize("c,nh,t", [(136, 2, 520), (17, 1, 300), (34, 2, 70)])
def test_chan_attn_backward(c, nh, t):
    from otpose_amd import train_ops as T
    hs = c // nh
    scale = 1.0 / math.sqrt(hs)
    q, k, v = seeded((2, c, t), 1) * 0.3, seeded((2, c, t), 2) * 0.3, seeded((2, c, t), 3)

    def ref(q, k, v):
        b = q.shape[0]
        qq, kk, vv = (z.view(b, nh, hs, -1) for z in (q, k, v))
        att = F.softmax((qq * scale) @ kk.transpose(-2, -1), dim=-1)
        return (att @ vv).transpose(2, 3).contiguous().view(b, c, -1)
    _check_op(ref, lambda q, k, v: T.chan_attn(q, k, v, nh, scale), [q, k, v], 2e-4)


@pytest.mark.parametrize("b,c,t,masked", [(3, 136, 432, True), (2, 17, 55, False), (4, 136, 6912, True)])
def test_scale_residual_matches_autograd(b, c, t, masked):
    """x + drop_path(scale * a) (blocks.py:277-279, 283-316) as one fused op vs the same expression in torch."""
    from otpose_amd import train_ops as T
    x, a = seeded((b, c, t), 1), seeded((b, c, t), 2)
    scale = seeded((1, c, 1), 3) * 0.5
    g = seeded((b, c, t), 4)
    mask = torch.tensor([0.0, 1 / 0.9, 1 / 0.9, 0.0][:b]) if masked else None
    xr, ar, sr = x.clone().requires_grad_(), a.clone().requires_grad_(), scale.clone().requires_grad_()
    y = sr * ar
    if masked:
        y = y * mask.view(b, 1, 1)
    (xr + y).backward(g)
    xd, ad, sd = x.cuda().requires_grad_(), a.cuda().requires_grad_(), scale.cuda().requires_grad_()
    out = T.scale_residual(xd, ad, sd, mask.cuda() if masked else None)
    out.backward(g.cuda())
    _close(out, (xr + y).detach())
    _close(xd.grad, xr.grad)
    _close(ad.grad, ar.grad)
    _close(sd.grad, sr.grad, 1e-4)


@pytest.mark.parametrize("mag", [1.0, 1e-5, 1e-8])
def test_gradient_operands_of_any_magnitude_keep_their_precision(mag):
    """The backward sends GRADIENTS through the projection and the attention products.  Their magnitude is whatever the loss
    makes it; the split products of those calls therefore use bfloat16 pieces (csrc/densex_grad.hip, transformer_grad.hip:
    16-17 bits at any magnitude), not the IEEE-half pieces of the forward, which flush 1e-8 to zero and hold 1e-5 to 8 bits."""
    from otpose_amd import ops
    from otpose_amd import train_ops as T
    c, t = 136, 520
    w = seeded((c, c, 1, 1), 5) / math.sqrt(c)
    gy = (seeded((2, c, 1, t), 6) * mag).cuda()
    gx = T.conv2d_grad_input(gy, w.cuda(), (2, c, 1, t), 1, 0, 1)
    ref = torch.einsum("oi,nohw->nihw", w.view(c, c).double(), gy.cpu().double())
    err = float((gx.cpu().double() - ref).abs().max()) / float(ref.abs().max())
    assert err <= 3e-5, err
    # what the forward's half pieces would make of the same operand (why the second copy of the kernels exists)
    pk = ops.pack_dense_cc(w.view(c, c).t().contiguous().cuda(), x3=True)
    half = ops.dense_cc([gy.view(2, c, t)], [pk], x3=True)[0]
    err_half = float((half.cpu().double().view_as(ref) - ref).abs().max()) / float(ref.abs().max())
    assert (err_half > 10 * err) == (mag < 1e-3), (err, err_half)
    # attention backward: dS = dO v^T and the three applies all carry a gradient operand
    hs, nh = 68, 2
    q, k, v = (seeded((2, c, t), s_) * 0.3 for s_ in (1, 2, 3))
    go = seeded((2, c, t), 4) * mag
    leaves = [z.clone().cuda().requires_grad_() for z in (q, k, v)]
    T.chan_attn(*leaves, nh, 1.0 / math.sqrt(hs)).backward(go.cuda())
    refs = [z.clone().double().requires_grad_() for z in (q, k, v)]
    qq, kk, vv = (z.view(2, nh, hs, -1) for z in refs)
    att = F.softmax((qq / math.sqrt(hs)) @ kk.transpose(-2, -1), dim=-1)
    (att @ vv).transpose(2, 3).contiguous().view(2, c, -1).backward(go.double())
    for a, b in zip(leaves, refs):
        e = float((a.grad.cpu().double() - b.grad).abs().max()) / float(b.grad.abs().max())
        assert e <= 2e-4, e
