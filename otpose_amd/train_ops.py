"""Training-mode operators over the C-ABI (building blocks of the OTPose training step, reference
script/Common.py:91,136-144 runs the model under ``model.train()``).

``conv2d`` / ``batch_norm_relu`` are ``torch.autograd.Function``s whose forward AND backward are HIP launches
(conv forward / dgrad: ``otp_conv2d``; wgrad: ``otp_conv2d_wgrad``; bias grad: ``otp_channel_sum``; BatchNorm with
batch statistics fused with the residual add + ReLU that follow it in HRNet / RSB blocks: ``otp_bn_train_*``).
PyTorch supplies memory, streams and the autograd tape only; CPU tensors raise like every operator here.
"""
from __future__ import annotations

import torch
from torch.autograd import Function

from . import hip
from .ops import ACT_NONE, View, _check_f32, _require_gpu, conv2d_launch, conv_desc, pack_conv_weight


def _out_hw(h, w, k, stride, pad, dil):
    return (h + 2 * pad - (dil * (k - 1) + 1)) // stride + 1, (w + 2 * pad - (dil * (k - 1) + 1)) // stride + 1


def conv2d_forward(x, weight, bias, stride, pad, dil):
    cout, cin, kh, kw = weight.shape
    ho, wo = _out_hw(x.shape[2], x.shape[3], kh, stride, pad, dil)
    out = torch.empty((x.shape[0], cout, ho, wo), dtype=torch.float32, device=x.device)
    iv, ov = View(x), View(out)
    d = conv_desc(iv, ov, cout, kh, kw, stride, pad, dil, ACT_NONE)
    conv2d_launch(iv, pack_conv_weight(weight), None, bias, ov, d)
    return out


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
    cin16 = (cin + 15) // 16 * 16
    wp = torch.empty(kh * kw * cout * cin16, dtype=torch.float32, device=g.device)
    hip.check(L.otp_conv2d_pack_weight_dgrad(hip.ptr(weight.contiguous()), hip.ptr(wp), cout, cin, kh, kw,
                                             hip.stream_of(g)), "otp_conv2d_pack_weight_dgrad")
    gx = torch.empty(in_shape, dtype=torch.float32, device=g.device)
    iv, ov = View(g), View(gx)
    d = conv_desc(iv, ov, cin, kh, kw, 1, dil * (kh - 1) - pad, dil, ACT_NONE)
    conv2d_launch(iv, wp, None, None, ov, d)
    return gx


def conv2d_grad_weight(x, grad_out, weight_shape, stride, pad, dil):
    cout, cin, kh, kw = weight_shape
    gw = torch.zeros(weight_shape, dtype=torch.float32, device=x.device)
    x, g = x.contiguous(), grad_out.contiguous()
    hip.check(hip.lib().otp_conv2d_wgrad(hip.ptr(x), hip.ptr(g), hip.ptr(gw), x.shape[0], cin, x.shape[2], x.shape[3],
                                         cout, kh, kw, stride, pad, dil, cin, 0, cout, 0, hip.stream_of(x)),
              "otp_conv2d_wgrad")
    return gw


def channel_sum(t):
    """Per-channel sum over (N, H, W) of a contiguous (N, C, H, W) tensor (bias gradients)."""
    n, c, h, w = t.shape
    L = hip.lib()
    nbytes = L.otp_bn_workspace(n, c, h * w) + 4 * c
    ws = torch.empty((nbytes + 7) // 8, dtype=torch.float64, device=t.device)
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
        return conv2d_forward(x, weight, bias, stride, pad, dil)

    @staticmethod
    def backward(ctx, grad_out):
        x, weight = ctx.saved_tensors
        stride, pad, dil, has_bias = ctx.cfg
        grad_out = grad_out.contiguous()
        gx = conv2d_grad_input(grad_out, weight, x.shape, stride, pad, dil) if ctx.needs_input_grad[0] else None
        gw = conv2d_grad_weight(x, grad_out, weight.shape, stride, pad, dil) if ctx.needs_input_grad[1] else None
        gb = channel_sum(grad_out) if has_bias and ctx.needs_input_grad[2] else None
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
        gg = torch.empty(c, dtype=torch.float32, device=x.device)
        gb = torch.empty(c, dtype=torch.float32, device=x.device)
        hip.check(L.otp_bn_train_backward(hip.ptr(grad_y), hip.ptr(x), hip.ptr(y), hip.ptr(mean), hip.ptr(rstd),
                                          hip.ptr(gamma), hip.ptr(gx), hip.ptr(gres), hip.ptr(gg), hip.ptr(gb),
                                          hip.ptr(ws), nbytes, n, c, h * w, c, 0, c, 0, c, 0, hip.stream_of(x)),
                  "otp_bn_train_backward")
        return gx, gg, gb, gres, None, None, None, None, None


def batch_norm_relu(x, gamma, beta, res=None, running_mean=None, running_var=None, momentum=0.1, eps=1e-5, relu=True):
    return BatchNormReluFunction.apply(x, gamma, beta, res, running_mean, running_var, momentum, eps, relu)
