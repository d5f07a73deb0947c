"""Training-mode operators over the C-ABI (building blocks of the OTPose training step, reference
script/Common.py:91,136-144 runs the model under ``model.train()``).

``conv2d`` / ``batch_norm_relu`` are ``torch.autograd.Function``s whose forward AND backward are HIP launches
(conv forward / dgrad: ``otp_conv2d``; wgrad: ``otp_conv2d_wgrad``; bias grad: ``otp_channel_sum``; BatchNorm with
batch statistics fused with the residual add + ReLU that follow it in HRNet / RSB blocks: ``otp_bn_train_*``).
PyTorch supplies memory, streams and the autograd tape only; CPU tensors raise like every operator here.
"""
from __future__ import annotations

import os

import torch
from torch.autograd import Function

from . import hip
from .ops import (ACT_NONE, View, _check_f32, _require_gpu, conv2d_launch, conv2d_wino_launch, conv_desc,
                  pack_conv_weight, pack_wino_weight, wino_supported)


def grad_slot(param):
    """Where the gradient of ``param`` should be written: its slot inside the optimizer's flat gradient buffer
    (:class:`otpose_amd.optim.FusedAdamW`) when that slot is known to be zero-filled AND this is the first request for it in
    the current step, else None (the caller allocates; autograd accumulates as usual).

    Some backward kernels accumulate into their destination (conv2d_grad_weight, dwconv3) and others overwrite, so a slot
    is handed out only while the optimizer vouches for it: ``FusedAdamW.zero_grad()`` opens a new epoch (slots memset),
    ``step()`` closes it.  ``model.zero_grad()`` / ``p.grad = None`` do not open one - the slot then still holds the last
    step's gradient and ordinary tensors are used instead (``FusedAdamW._rehome_grads`` copies them in).  A parameter that
    feeds two autograd nodes of one backward (two forwards summed, shared weights) gets the slot for the first node only:
    autograd sums the second gradient as an ordinary tensor instead of seeing the same memory twice."""
    slot = getattr(param, "_otp_grad_slot", None)
    if slot is None or slot.shape != param.shape:
        return None
    owner = getattr(param, "_otp_grad_owner", param)        # a reshaped view of a parameter names its owner (TrainGraph.conv1d)
    epoch = getattr(owner, "_otp_slot_epoch", None)          # shared [epoch, clean] cell of the owning optimizer
    if epoch is None or not epoch[1] or owner.grad is not None or getattr(owner, "_otp_slot_taken", -1) == epoch[0]:
        return None
    owner._otp_slot_taken = epoch[0]
    # a fresh tensor object over the same memory: autograd adopts an incoming gradient without cloning it only when nobody
    # else holds a reference to that tensor object
    return slot.view(slot.shape)


def _out_hw(h, w, k, stride, pad, dil):
    return (h + 2 * pad - (dil * (k - 1) + 1)) // stride + 1, (w + 2 * pad - (dil * (k - 1) + 1)) // stride + 1


def conv2d_forward(x, weight, bias, stride, pad, dil):
    cout, cin, kh, kw = weight.shape
    ho, wo = _out_hw(x.shape[2], x.shape[3], kh, stride, pad, dil)
    out = torch.empty((x.shape[0], cout, ho, wo), dtype=torch.float32, device=x.device)
    iv, ov = View(x), View(out)
    d = conv_desc(iv, ov, cout, kh, kw, stride, pad, dil, ACT_NONE)
    if _dense_cc(cin, cout, kh, stride, pad, x.shape[2] * x.shape[3]):
        _dense_cc_launch(x, weight.reshape(cout, cin), bias, out)
    elif _winograd(cin, cout, d):
        conv2d_wino_launch(iv, pack_wino_weight(weight), None, bias, ov, d)
    else:
        conv2d_launch(iv, pack_conv_weight(weight), None, bias, ov, d)
    return out


def _dense_cc(cin, cout, k, stride, pad, hw):
    """Pointwise C -> C layers of the temporal encoders (query / key / value / proj): the register-resident-input kernel of
    csrc/dense.hip, forward and input gradient alike (same rule as the inference engine)."""
    from . import ops
    return (os.environ.get("OTPOSE_DENSE_CC", "1") != "0" and k == 1 and stride == 1 and pad == 0 and cin == cout
            and ops.dense_cc_supported(cin, hw))


def _dense_cc_launch(x4, w2, bias, out4, grad=False):
    from . import ops
    n, c = x4.shape[:2]
    # split products (csrc/densex.hip) unless the exact-fp32 kernels are asked for, as in the inference engine; ``grad``: x4 is a
    # gradient - bfloat16 pieces (csrc/densex_grad.hip), an IEEE-half piece would flush the small ones to zero
    x3 = os.environ.get("OTPOSE_CONV_MATH", "x3") != "f32" and ops.dense_x3_supported(c, x4.shape[2] * x4.shape[3])
    pk = ops.pack_dense_cc(w2, None, bias, x3=x3, grad=grad)
    ops.dense_cc([x4.view(n, c, -1)], [pk], None, [out4.view(n, c, -1)], x3=x3, grad=grad)


def _winograd(cin, cout, d):
    """3x3 / stride 1 / pad 1 layers with enough channels run on the Winograd kernel, forward and input gradient alike
    (same rule as the inference engine)."""
    from .engine import InferenceEngine
    return (os.environ.get("OTPOSE_WINOGRAD", "1") != "0" and d.kh == 3 and d.stride == 1 and d.pad == 1 and d.dil == 1
            and InferenceEngine.winograd_pays(cin, cout) and wino_supported(d))


def conv2d_grad_input(grad_out, weight, in_shape, stride, pad, dil):
    """dL/dx: a stride-1 convolution of (zero-inserted) grad_out with the flipped, channel-transposed weights."""
    cout, cin, kh, kw = weight.shape
    n, _, h, w = in_shape
    L = hip.lib()
    g = grad_out.contiguous()
    if stride > 1:
        # zero-insert to the resolution a stride-1 "full" correlation needs: H + 2*pad - dil*(k-1)
        hd, wd = h + 2 * pad - dil * (kh - 1), w + 2 * pad - dil * (kw - 1)
        gd = torch.empty((n, cout, hd, wd), dtype=torch.float32, device=g.device)
        hip.check(L.otp_dilate(hip.ptr(g), hip.ptr(gd), n * cout, g.shape[2], g.shape[3], stride, hd, wd,
                               hip.stream_of(g)), "otp_dilate")
        g = gd
    gx = torch.empty(in_shape, dtype=torch.float32, device=g.device)
    if _dense_cc(cin, cout, kh, stride, pad, h * w):
        _dense_cc_launch(g, weight.reshape(cout, cin).t(), None, gx, grad=True)         # dx = W^T . dy
        return gx
    iv, ov = View(g), View(gx)
    d = conv_desc(iv, ov, cin, kh, kw, 1, dil * (kh - 1) - pad, dil, ACT_NONE)
    if stride == 1 and _winograd(cout, cin, d):
        # the transposed convolution as a Winograd conv: weights flipped and channel-transposed, then G g G^T
        wt = weight.flip(2, 3).transpose(0, 1).contiguous()
        conv2d_wino_launch(iv, pack_wino_weight(wt), None, None, ov, d)
        return gx
    cin16 = (cin + 15) // 16 * 16
    wp = torch.empty(kh * kw * cout * cin16, dtype=torch.float32, device=g.device)
    hip.check(L.otp_conv2d_pack_weight_dgrad(hip.ptr(weight.contiguous()), hip.ptr(wp), cout, cin, kh, kw,
                                             hip.stream_of(g)), "otp_conv2d_pack_weight_dgrad")
    conv2d_launch(iv, wp, None, None, ov, d)
    return gx


_WS = {}


def _workspace(device, nbytes):
    """One grow-only scratch buffer per (device, stream) for the wgrad partial sums: every user launches on the current
    stream, so reuse is stream-ordered; the independent paths of a training step run on several streams and each has its own."""
    key = (device, torch.cuda.current_stream(device).cuda_stream)
    buf = _WS.get(key)
    if buf is None or buf.numel() * 4 < nbytes:
        buf = torch.empty((nbytes + 3) // 4, dtype=torch.float32, device=device)
        _WS[key] = buf
    return buf


def conv2d_grad_weight(x, grad_out, weight_shape, stride, pad, dil, out=None):
    """``out``: a zero-filled (Cout, Cin, kh, kw) destination (the kernel accumulates), e.g. a :func:`grad_slot`."""
    cout, cin, kh, kw = weight_shape
    gw = out if out is not None else torch.zeros(weight_shape, dtype=torch.float32, device=x.device)
    x, g = x.contiguous(), grad_out.contiguous()
    L = hip.lib()
    nbytes = L.otp_conv2d_wgrad_workspace(cin, cout)
    ws = _workspace(x.device, nbytes)
    hip.check(L.otp_conv2d_wgrad(hip.ptr(x), hip.ptr(g), hip.ptr(gw), x.shape[0], cin, x.shape[2], x.shape[3],
                                 cout, kh, kw, stride, pad, dil, cin, 0, cout, 0, hip.ptr(ws), nbytes,
                                 hip.stream_of(x)), "otp_conv2d_wgrad")
    return gw


def channel_sum(t, out=None):
    """Per-channel sum over (N, H, W) of a contiguous (N, C, H, W) tensor (bias gradients)."""
    n, c, h, w = t.shape
    L = hip.lib()
    nbytes = L.otp_bn_workspace(n, c, h * w) + 4 * c
    ws = torch.empty((nbytes + 7) // 8, dtype=torch.float64, device=t.device)
    if out is None:
        out = torch.empty(c, dtype=torch.float32, device=t.device)
    hip.check(L.otp_channel_sum(hip.ptr(t), hip.ptr(out), hip.ptr(ws), nbytes, n, c, h * w, c, 0, hip.stream_of(t)),
              "otp_channel_sum")
    return out


class Conv2dFunction(Function):
    """``F.conv2d(x, weight, bias, stride, padding, dilation)`` (groups = 1, square 1x1 / 3x3 kernels)."""

    @staticmethod
    def forward(ctx, x, weight, bias, stride, pad, dil):
        _require_gpu(x, weight)
        _check_f32(x, weight)
        x, weight = x.contiguous(), weight.contiguous()
        ctx.save_for_backward(x, weight)
        ctx.cfg = (stride, pad, dil, bias is not None)
        ctx.params = (weight, bias)                    # gradient slots are looked up at backward time
        return conv2d_forward(x, weight, bias, stride, pad, dil)

    @staticmethod
    def backward(ctx, grad_out):
        x, weight = ctx.saved_tensors
        stride, pad, dil, has_bias = ctx.cfg
        pw, pb = ctx.params
        grad_out = grad_out.contiguous()
        gx = conv2d_grad_input(grad_out, weight, x.shape, stride, pad, dil) if ctx.needs_input_grad[0] else None
        gw = (conv2d_grad_weight(x, grad_out, weight.shape, stride, pad, dil, grad_slot(pw))
              if ctx.needs_input_grad[1] else None)
        gb = channel_sum(grad_out, grad_slot(pb)) if has_bias and ctx.needs_input_grad[2] else None
        return gx, gw, gb, None, None, None


def conv2d(x, weight, bias=None, stride=1, padding=0, dilation=1):
    return Conv2dFunction.apply(x, weight, bias, stride, padding, dilation)


class BatchNormReluFunction(Function):
    """``relu?(F.batch_norm(x, running_mean, running_var, gamma, beta, training=True, momentum, eps) (+ res))``;
    the running statistics are updated in place like ``nn.BatchNorm2d`` does."""

    @staticmethod
    def forward(ctx, x, gamma, beta, res, running_mean, running_var, momentum, eps, relu):
        _require_gpu(x, gamma, beta)
        _check_f32(x, gamma, beta)
        x = x.contiguous()
        n, c, h, w = x.shape
        L = hip.lib()
        nbytes = L.otp_bn_workspace(n, c, h * w)
        ws = torch.empty((nbytes + 7) // 8, dtype=torch.float64, device=x.device)
        y = torch.empty_like(x)
        mean = torch.empty(c, dtype=torch.float32, device=x.device)
        rstd = torch.empty(c, dtype=torch.float32, device=x.device)
        r = res.contiguous() if res is not None else None
        hip.check(L.otp_bn_train_forward(hip.ptr(x), hip.ptr(gamma), hip.ptr(beta), hip.ptr(r), hip.ptr(y), hip.ptr(mean),
                                         hip.ptr(rstd), hip.ptr(running_mean), hip.ptr(running_var), hip.ptr(ws), nbytes,
                                         n, c, h * w, eps, momentum, int(relu), c, 0, c, 0, c, 0, hip.stream_of(x)),
                  "otp_bn_train_forward")
        ctx.save_for_backward(x, gamma, mean, rstd, y if relu else None)
        ctx.has_res = res is not None
        ctx.params = (gamma, beta)
        return y

    @staticmethod
    def backward(ctx, grad_y):
        x, gamma, mean, rstd, y = ctx.saved_tensors
        grad_y = grad_y.contiguous()
        n, c, h, w = x.shape
        L = hip.lib()
        nbytes = L.otp_bn_workspace(n, c, h * w)
        ws = torch.empty((nbytes + 7) // 8, dtype=torch.float64, device=x.device)
        gx = torch.empty_like(x)
        gres = torch.empty_like(x) if ctx.has_res else None
        sg, sb = grad_slot(ctx.params[0]), grad_slot(ctx.params[1])
        gg = sg if sg is not None else torch.empty(c, dtype=torch.float32, device=x.device)
        gb = sb if sb is not None else torch.empty(c, dtype=torch.float32, device=x.device)
        hip.check(L.otp_bn_train_backward(hip.ptr(grad_y), hip.ptr(x), hip.ptr(y), hip.ptr(mean), hip.ptr(rstd),
                                          hip.ptr(gamma), hip.ptr(gx), hip.ptr(gres), hip.ptr(gg), hip.ptr(gb),
                                          hip.ptr(ws), nbytes, n, c, h * w, c, 0, c, 0, c, 0, hip.stream_of(x)),
                  "otp_bn_train_backward")
        return gx, gg, gb, gres, None, None, None, None, None


def batch_norm_relu(x, gamma, beta, res=None, running_mean=None, running_var=None, momentum=0.1, eps=1e-5, relu=True):
    return BatchNormReluFunction.apply(x, gamma, beta, res, running_mean, running_var, momentum, eps, relu)


# ------------------------------------------------------------------------------------------------
# ConvTransformer pieces (reference model/blocks.py), tensors (B, C, T)
# ------------------------------------------------------------------------------------------------
def _chan_sum3(t):
    b, c, n = t.shape
    return channel_sum(t.reshape(b, c, 1, n))


def _ln_fwd(x, gamma, beta, eps):
    """y = channel LayerNorm of a contiguous (B, C, T) tensor; gamma / beta flat (C)."""
    b, c, t = x.shape
    y = torch.empty_like(x)
    hip.check(hip.lib().otp_ln_channel(hip.ptr(x), hip.ptr(gamma), hip.ptr(beta), hip.ptr(y), None, b, c, t, eps,
                                       hip.stream_of(x)), "otp_ln_channel")
    return y


def _ln_bwd(x, g, gy, eps, pgamma, pbeta):
    """(dx, dgamma (C), dbeta (C)) of :func:`_ln_fwd`; the parameter gradients land in the optimizer's slots when free."""
    gy = gy.contiguous()
    b, c, t = x.shape
    L = hip.lib()
    nbytes = L.otp_ln_channel_backward_workspace(b, c, t)
    if nbytes:
        # dx, dgamma and dbeta from one pass over x and dy
        gx = torch.empty_like(x)
        ws = torch.empty(nbytes // 4, dtype=torch.float32, device=x.device)
        sg, sb = grad_slot(pgamma), grad_slot(pbeta)
        gg = sg if sg is not None else torch.empty(c, dtype=torch.float32, device=x.device)
        gb = sb if sb is not None else torch.empty(c, dtype=torch.float32, device=x.device)
        hip.check(L.otp_ln_channel_backward_params(hip.ptr(x), hip.ptr(gy), hip.ptr(g), hip.ptr(gx), hip.ptr(gg), hip.ptr(gb),
                                                   hip.ptr(ws), nbytes, b, c, t, eps, hip.stream_of(x)),
                  "otp_ln_channel_backward_params")
        return gx, gg, gb
    gx, dyxh = torch.empty_like(x), torch.empty_like(x)
    hip.check(L.otp_ln_channel_backward(hip.ptr(x), hip.ptr(gy), hip.ptr(g), hip.ptr(gx), hip.ptr(dyxh), b, c, t,
                                        eps, hip.stream_of(x)), "otp_ln_channel_backward")
    return gx, _chan_sum3(dyxh), _chan_sum3(gy)


class LayerNormFunction(Function):
    """Channel LayerNorm of (B, C, T) (model/blocks.py:95-110): biased variance over C, weight / bias (C)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps):
        _require_gpu(x)
        _check_f32(x)
        x = x.contiguous()
        g, be = gamma.reshape(-1).contiguous(), beta.reshape(-1).contiguous()
        y = _ln_fwd(x, g, be, eps)
        ctx.save_for_backward(x, g)
        ctx.eps, ctx.pshape = eps, gamma.shape
        ctx.params = (gamma, beta)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, g = ctx.saved_tensors
        gx, gg, gb = _ln_bwd(x, g, gy, ctx.eps, ctx.params[0], ctx.params[1])
        return gx, gg.reshape(ctx.pshape), gb.reshape(ctx.pshape), None


def layer_norm(x, gamma, beta, eps=1e-5):
    return LayerNormFunction.apply(x, gamma, beta, eps)


def _dw_fwd(x, w, stride):
    b, c, t = x.shape
    to = (t + 2 - 3) // stride + 1
    y = torch.empty((b, c, to), dtype=torch.float32, device=x.device)
    hip.check(hip.lib().otp_dwconv3_forward(hip.ptr(x), hip.ptr(w), hip.ptr(y), b, c, t, stride, hip.stream_of(x)),
              "otp_dwconv3_forward")
    return y


def _dw_bwd(x, w, gy, stride, pw):
    b, c, t = x.shape
    sw = grad_slot(pw)                                     # zero-filled by the optimizer's zero_grad; the kernel accumulates
    gx, gw = torch.empty_like(x), (sw if sw is not None else torch.zeros_like(w))
    hip.check(hip.lib().otp_dwconv3_backward(hip.ptr(x), hip.ptr(w), hip.ptr(gy.contiguous()), hip.ptr(gx), hip.ptr(gw),
                                             b, c, t, stride, hip.stream_of(x)), "otp_dwconv3_backward")
    return gx, gw


class DwConv3Function(Function):
    """``F.conv1d(x, w (C,1,3), None, stride, 1, 1, groups=C)`` (model/blocks.py:359-381)."""

    @staticmethod
    def forward(ctx, x, w, stride):
        _require_gpu(x, w)
        x, w = x.contiguous(), w.contiguous()
        y = _dw_fwd(x, w, stride)
        ctx.save_for_backward(x, w)
        ctx.stride = stride
        ctx.params = (w,)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, w = ctx.saved_tensors
        gx, gw = _dw_bwd(x, w, gy, ctx.stride, ctx.params[0])
        return gx, gw, None


def dwconv3(x, w, stride=1):
    return DwConv3Function.apply(x, w, stride)


class AttnFrontFunction(Function):
    """The front of MultiHeadConvAttention (model/blocks.py:400-440) as one node: ``xn = ln1(x)`` and, for query / key /
    value, ``conv1d_1x1(LayerNorm(dwconv3(xn)))``.  Only ``x`` is kept for the backward; ``xn``, the three depthwise-conv
    outputs and the three LayerNorm outputs (seven (B, C, T) tensors, 420 MB per block at cfg2 - 5 GB over the two temporal
    encoders) are rebuilt there by the same seven launches (~160 us per block), branch by branch, so at most three of them are
    alive at a time.  Arguments after ``eps``: ln1 weight / bias, then per branch (dwconv weight, LayerNorm weight, bias,
    projection weight as (C, C, 1, 1), projection bias or None)."""

    @staticmethod
    def forward(ctx, x, stride, eps, g1, b1, *bp):
        _require_gpu(x)
        _check_f32(x)
        x = x.contiguous()
        ctx.cfg = (stride, eps)
        ctx.params = (g1, b1) + tuple(bp)
        xn = _ln_fwd(x, g1.reshape(-1).contiguous(), b1.reshape(-1).contiguous(), eps)
        outs = []
        for i in range(3):
            dw, lg, lb, w4, bias = bp[5 * i: 5 * i + 5]
            z = _ln_fwd(_dw_fwd(xn, dw.contiguous(), stride), lg.reshape(-1).contiguous(), lb.reshape(-1).contiguous(), eps)
            outs.append(conv2d_forward(z.unsqueeze(2), w4.contiguous(), bias, 1, 0, 1).squeeze(2))
        ctx.save_for_backward(x)
        return tuple(outs)

    @staticmethod
    def backward(ctx, *grads):
        (x,) = ctx.saved_tensors
        stride, eps = ctx.cfg
        g1, b1 = ctx.params[:2]
        bp = ctx.params[2:]
        g1f = g1.reshape(-1).contiguous()
        xn = _ln_fwd(x, g1f, b1.reshape(-1).contiguous(), eps)
        gxn, out = None, [None] * 15
        for i in (2, 1, 0):                   # value, key, query: the order autograd would run (and sum) the per-layer nodes in
            dw, lg, lb, w4, bias = bp[5 * i: 5 * i + 5]
            if grads[i] is None:
                continue
            g4 = grads[i].contiguous().unsqueeze(2)
            lgf = lg.reshape(-1).contiguous()
            y = _dw_fwd(xn, dw.contiguous(), stride)
            z4 = _ln_fwd(y, lgf, lb.reshape(-1).contiguous(), eps).unsqueeze(2)
            gz = conv2d_grad_input(g4, w4.contiguous(), z4.shape, 1, 0, 1).squeeze(2)
            gw = conv2d_grad_weight(z4, g4, w4.shape, 1, 0, 1, grad_slot(w4))
            gb = channel_sum(g4, grad_slot(bias)) if bias is not None else None
            del z4
            gy, glg, glb = _ln_bwd(y, lgf, gz, eps, lg, lb)
            del y, gz
            gxi, gdw = _dw_bwd(xn, dw.contiguous(), gy, stride, dw)
            gxn = gxi if gxn is None else gxn.add_(gxi)
            out[5 * i: 5 * i + 5] = [gdw, glg.reshape(lg.shape), glb.reshape(lb.shape), gw, gb]
        if gxn is None:
            return (None,) * (5 + 15)
        gx, gg1, gb1 = _ln_bwd(x, g1f, gxn, eps, g1, b1)
        return (gx, None, None, gg1.reshape(g1.shape), gb1.reshape(b1.shape)) + tuple(out)


def attn_front(x, stride, eps, ln1, branches):
    """``branches``: three (dwconv weight, LayerNorm weight, LayerNorm bias, projection weight (C, C, 1, 1), bias) tuples
    (query, key, value); ``ln1`` = (weight, bias).  Returns (q, k, v)."""
    flat = [t for br in branches for t in br]
    return AttnFrontFunction.apply(x, stride, eps, ln1[0], ln1[1], *flat)


class GeluFunction(Function):
    @staticmethod
    def forward(ctx, x):
        _require_gpu(x)
        x = x.contiguous()
        y = torch.empty_like(x)
        hip.check(hip.lib().otp_gelu_forward(hip.ptr(x), hip.ptr(y), x.numel(), hip.stream_of(x)), "otp_gelu_forward")
        ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, gy):
        (x,) = ctx.saved_tensors
        gx = torch.empty_like(x)
        hip.check(hip.lib().otp_gelu_backward(hip.ptr(x), hip.ptr(gy.contiguous()), hip.ptr(gx), x.numel(),
                                              hip.stream_of(x)), "otp_gelu_backward")
        return gx


gelu = GeluFunction.apply


class MaxPool3s2Function(Function):
    """``nn.MaxPool1d(3, 2, 1)`` on (B, C, T) (model/blocks.py:234-238)."""

    @staticmethod
    def forward(ctx, x):
        _require_gpu(x)
        x = x.contiguous()
        b, c, t = x.shape
        y = torch.empty((b, c, (t + 2 - 3) // 2 + 1), dtype=torch.float32, device=x.device)
        hip.check(hip.lib().otp_maxpool3s2_forward(hip.ptr(x), hip.ptr(y), b * c, t, hip.stream_of(x)),
                  "otp_maxpool3s2_forward")
        ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, gy):
        (x,) = ctx.saved_tensors
        b, c, t = x.shape
        gx = torch.empty_like(x)
        hip.check(hip.lib().otp_maxpool3s2_backward(hip.ptr(x), hip.ptr(gy.contiguous()), hip.ptr(gx), b * c, t,
                                                    hip.stream_of(x)), "otp_maxpool3s2_backward")
        return gx


maxpool3s2 = MaxPool3s2Function.apply


class UpsampleLinearFunction(Function):
    """``nn.Upsample(scale_factor=f, mode='linear', align_corners=False)`` on (B, C, T)."""

    @staticmethod
    def forward(ctx, x, f):
        _require_gpu(x)
        x = x.contiguous()
        b, c, t = x.shape
        y = torch.empty((b, c, t * f), dtype=torch.float32, device=x.device)
        hip.check(hip.lib().otp_upsample_linear(hip.ptr(x), hip.ptr(y), b, c, t, f, c, 0, hip.stream_of(x)),
                  "otp_upsample_linear")
        ctx.f, ctx.shape = f, (b, c, t)
        return y

    @staticmethod
    def backward(ctx, gy):
        b, c, t = ctx.shape
        gy = gy.contiguous()
        gx = torch.empty((b, c, t), dtype=torch.float32, device=gy.device)
        hip.check(hip.lib().otp_upsample_linear_backward(hip.ptr(gy), hip.ptr(gx), b, c, t, ctx.f, c, 0,
                                                         hip.stream_of(gy)), "otp_upsample_linear_backward")
        return gx, None


def upsample_linear(x, f):
    return UpsampleLinearFunction.apply(x, f)


class ChanAttnFunction(Function):
    """Channel attention of model/blocks.py:427-447: per (b, head) S = (q*scale) k^T (hs x hs, contraction over T),
    P = softmax(S), O = P v, returned in the ``transpose(2,3).contiguous().view(B, C, T)`` memory image."""

    @staticmethod
    def forward(ctx, q, k, v, n_head, scale):
        _require_gpu(q, k, v)
        q, k, v = q.contiguous(), k.contiguous(), v.contiguous()
        b, c, t = q.shape
        L = hip.lib()
        nbytes = L.otp_chan_attn_workspace(b, c, t, n_head)
        ws = torch.empty(max(nbytes // 4, 1), dtype=torch.float32, device=q.device)
        out = torch.empty_like(q)
        hip.check(L.otp_chan_attn(hip.ptr(q), hip.ptr(k), hip.ptr(v), hip.ptr(out), hip.ptr(ws), nbytes, b, c, t, n_head,
                                  scale, hip.stream_of(q)), "otp_chan_attn")
        bh, hs = b * n_head, c // n_head
        hsp = (hs + 15) // 16 * 16
        ns = L.otp_chan_attn_splits(bh, t)
        p = ws[bh * ns * hsp * hsp: bh * ns * hsp * hsp + bh * hsp * hsp].clone()     # softmax matrix, zero padded
        ctx.save_for_backward(q, k, v, p)
        ctx.cfg = (n_head, scale)
        return out

    @staticmethod
    def backward(ctx, gout):
        q, k, v, p = ctx.saved_tensors
        n_head, scale = ctx.cfg
        b, c, t = q.shape
        bh, hs = b * n_head, c // n_head
        hsp = (hs + 15) // 16 * 16
        L = hip.lib()
        st = hip.stream_of(q)
        dev = q.device
        new = lambda *s: torch.empty(s, dtype=torch.float32, device=dev)       # noqa: E731
        gout = gout.contiguous()
        d_o = new(bh, hs, t)                                   # dO[bh][i][t] from the [bh][t][i] image
        hip.check(L.otp_transpose_scale(hip.ptr(gout), hip.ptr(d_o), bh, t, hs, 1.0, st), "otp_transpose_scale")
        ns = L.otp_chan_attn_splits(bh, t)
        slabs = new(bh * ns * hsp * hsp)
        # (every product of the backward has a gradient operand: bfloat16 pieces - csrc/transformer_grad.hip)
        hip.check(L.otp_chan_attn_scores_bf16p(hip.ptr(d_o), hip.ptr(v), hip.ptr(slabs), bh, hs, t, st), "otp_chan_attn_scores")
        d_s, d_st, p_t = new(bh, hsp, hsp), new(bh, hsp, hsp), new(bh, hsp, hsp)
        hip.check(L.otp_softmax_backward(hip.ptr(slabs), hip.ptr(p), hip.ptr(d_s), hip.ptr(d_st), hip.ptr(p_t), bh, hs, ns,
                                         st), "otp_softmax_backward")
        tmp = new(bh, t, hs)
        grads = []
        for src, mat, sc in ((k, d_s, scale), (q, d_st, scale), (d_o, p_t, 1.0)):
            hip.check(L.otp_chan_attn_apply_bf16p(hip.ptr(src), hip.ptr(mat), hip.ptr(tmp), bh, hs, t, st), "otp_chan_attn_apply")
            g = new(b, c, t)
            hip.check(L.otp_transpose_scale(hip.ptr(tmp), hip.ptr(g), bh, t, hs, sc, st), "otp_transpose_scale")
            grads.append(g)
        return grads[0], grads[1], grads[2], None, None


def chan_attn(q, k, v, n_head, scale):
    return ChanAttnFunction.apply(q, k, v, n_head, scale)


class ScaleResidualFunction(Function):
    """``x + drop_path(scale * a)`` of a TransformerBlock (model/blocks.py:277-279, AffineDropPath :283-316): ``scale`` is the
    (1, C, 1) AffineDropPath parameter, ``mask`` (B,) the per-sample Bernoulli(keep) / keep factors (None: no drop-path)."""

    @staticmethod
    def forward(ctx, x, a, scale, mask):
        _require_gpu(x, a, scale)
        _check_f32(x, a)
        x, a = x.contiguous(), a.contiguous()
        b, c, t = x.shape
        sc = scale.reshape(-1).contiguous()
        out = torch.empty_like(x)
        hip.check(hip.lib().otp_scale_residual(hip.ptr(x), hip.ptr(a), hip.ptr(sc), hip.ptr(mask), hip.ptr(out), b, c, t,
                                               hip.stream_of(x)), "otp_scale_residual")
        ctx.save_for_backward(a, sc, mask)
        ctx.pshape = scale.shape
        ctx.params = (scale,)
        return out

    @staticmethod
    def backward(ctx, g):
        a, sc, mask = ctx.saved_tensors
        g = g.contiguous()
        b, c, t = a.shape
        L = hip.lib()
        nbytes = L.otp_scale_residual_backward_workspace(b, c, t)
        ws = torch.empty(nbytes // 4, dtype=torch.float32, device=a.device)
        ga = torch.empty_like(a)
        ss = grad_slot(ctx.params[0])
        gs = ss if ss is not None else torch.empty(c, dtype=torch.float32, device=a.device)
        hip.check(L.otp_scale_residual_backward(hip.ptr(g), hip.ptr(a), hip.ptr(sc), hip.ptr(mask), hip.ptr(ga), hip.ptr(gs),
                                                hip.ptr(ws), nbytes, b, c, t, hip.stream_of(a)), "otp_scale_residual_backward")
        return g, ga, gs.reshape(ctx.pshape), None


def scale_residual(x, a, scale, mask=None):
    return ScaleResidualFunction.apply(x, a, scale, mask)
