"""Operator-level Python API over the C-ABI (mirror of the reference's native operator boundary).

``modulated_deform_conv`` replaces ``thirdparty.deform_conv.modulated_deform_conv``
(reference thirdparty/deform_conv/functions/deform_conv.py:109-179): same argument order, same
``NotImplementedError`` for CPU tensors (functions/deform_conv.py:131,149), same gradient tuple.
``modulated_deform_conv_cuda_forward`` / ``_backward`` keep the pybind entry points' names and
argument lists (reference thirdparty/deform_conv/src/deform_conv_cuda.cpp:474-480, 551-558) for
callers that bind the native module directly.
"""
from __future__ import annotations

import ctypes
import math
import os

import torch
from torch.autograd import Function

from . import hip

_DTYPE_F32 = 0


def _require_gpu(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise NotImplementedError("otpose_amd operators run on the GPU only (no CPU path)")


def _check_f32(*tensors):
    for t in tensors:
        if t is not None and t.dtype != torch.float32:
            raise RuntimeError(f"otpose_amd HIP operators are built for float32, got {t.dtype}")


def _out_hw(h, w, kh, kw, stride, pad, dil):
    return ((h + 2 * pad - (dil * (kh - 1) + 1)) // stride + 1,
            (w + 2 * pad - (dil * (kw - 1) + 1)) // stride + 1)


_DTYPES = {torch.float32: 0, torch.float16: 1, torch.bfloat16: 2, torch.float64: 3}


def _dcn_dtype(*tensors):
    """Storage type code of a DCN call: every tensor must share one of the types the operator is instantiated for
    (the reference dispatches double / float / half, deform_conv_cuda_kernel.cu:719,751,784; bf16 is an addition)."""
    ts = [t for t in tensors if t is not None]
    dt = ts[0].dtype
    if dt not in _DTYPES:
        raise RuntimeError(f"deformable convolution is not implemented for {dt}")
    for t in ts:
        if t.dtype != dt:
            raise RuntimeError(f"deformable convolution: mixed dtypes {dt} / {t.dtype}")
    return _DTYPES[dt]


def modulated_deform_conv_cuda_forward(input, weight, bias, ones, offset, mask, output, columns,
                                       kernel_h, kernel_w, stride_h, stride_w, pad_h, pad_w,
                                       dilation_h, dilation_w, group, deformable_group, with_bias):
    """In-place forward with the reference pybind signature (deform_conv_cuda.cpp:474-480).
    ``ones`` and ``columns`` are accepted and ignored: the fused kernel needs no im2col scratch.
    ``mask = None`` runs DCN v1 (no modulation) on the same kernels."""
    _require_gpu(input, weight, offset, mask, output)
    b = bias if with_bias else None
    dtype = _dcn_dtype(input, weight, offset, mask, output, b)
    if not input.is_contiguous():
        raise RuntimeError("input tensor has to be contiguous")
    if not weight.is_contiguous():
        raise RuntimeError("weight tensor has to be contiguous")
    n, c, h, w = input.shape
    cout, cpg, kh_, kw_ = weight.shape
    if (kh_, kw_) != (kernel_h, kernel_w):
        raise RuntimeError(f"Input shape and kernel shape wont match: ({kernel_h} x {kernel_w} vs {kh_} x {kw_}).")
    if c != cpg * group:
        raise RuntimeError(f"Input shape and kernel channels wont match: ({c} vs {cpg * group}).")
    offset = offset.contiguous()
    mask = mask.contiguous() if mask is not None else None
    st = hip.lib().otp_mdcn_forward_ex(
        hip.ptr(input), hip.ptr(offset), hip.ptr(mask), hip.ptr(weight), hip.ptr(b), hip.ptr(output),
        n, c, h, w, cout, kernel_h, kernel_w, stride_h, stride_w, pad_h, pad_w, dilation_h, dilation_w, group,
        deformable_group, 1.0, 0.0, dtype, hip.stream_of(input))
    hip.check(st, "otp_mdcn_forward")


def modulated_deform_conv_cuda_backward(input, weight, bias, ones, offset, mask, columns,
                                        grad_input, grad_weight, grad_bias, grad_offset, grad_mask,
                                        grad_output, kernel_h, kernel_w, stride_h, stride_w, pad_h,
                                        pad_w, dilation_h, dilation_w, group, deformable_group, with_bias):
    """In-place backward with the reference pybind signature (deform_conv_cuda.cpp:551-558).
    grad_input/grad_offset/grad_mask are overwritten; grad_weight/grad_bias are accumulated into
    (the reference accumulates them over the batch with addmm_, cpp:638-650).  ``mask = grad_mask = None``: DCN v1."""
    _require_gpu(input, weight, offset, mask, grad_output)
    gb = grad_bias if with_bias else None
    dtype = _dcn_dtype(input, weight, offset, mask, grad_output, grad_input, grad_weight, grad_offset, grad_mask, gb)
    if not input.is_contiguous():
        raise RuntimeError("input tensor has to be contiguous")
    if not weight.is_contiguous():
        raise RuntimeError("weight tensor has to be contiguous")
    n, c, h, w = input.shape
    cout = weight.shape[0]
    offset = offset.contiguous()
    mask = mask.contiguous() if mask is not None else None
    grad_output = grad_output.contiguous()
    L = hip.lib()
    ws_bytes = L.otp_mdcn_backward_workspace_ex(n, c, h, w, cout, kernel_h, kernel_w, stride_h, stride_w, pad_h, pad_w,
                                                dilation_h, dilation_w, group, deformable_group, dtype, int(mask is not None))
    ws = torch.empty(max(int(ws_bytes), 8) // 8, dtype=torch.float64, device=input.device)
    st = L.otp_mdcn_backward_ex(
        hip.ptr(input), hip.ptr(offset), hip.ptr(mask), hip.ptr(weight), hip.ptr(grad_output),
        hip.ptr(grad_input), hip.ptr(grad_offset), hip.ptr(grad_mask), hip.ptr(grad_weight),
        hip.ptr(gb), hip.ptr(ws), ws_bytes,
        n, c, h, w, cout, kernel_h, kernel_w, stride_h, stride_w, pad_h, pad_w, dilation_h, dilation_w, group,
        deformable_group, dtype, hip.stream_of(input))
    hip.check(st, "otp_mdcn_backward")


class ModulatedDeformConvFunction(Function):
    """autograd wrapper (reference thirdparty/deform_conv/functions/deform_conv.py:109-179)."""

    @staticmethod
    def forward(ctx, input, offset, mask, weight, bias=None, stride=1, padding=0, dilation=1,
                groups=1, deformable_groups=1):
        ctx.stride, ctx.padding, ctx.dilation = stride, padding, dilation
        ctx.groups, ctx.deformable_groups = groups, deformable_groups
        ctx.with_bias = bias is not None
        if not input.is_cuda:
            raise NotImplementedError
        if not ctx.with_bias:
            bias = input.new_empty(1)
        if weight.requires_grad or mask.requires_grad or offset.requires_grad or input.requires_grad:
            ctx.save_for_backward(input, offset, mask, weight, bias)
        kh, kw = weight.shape[2:4]
        ho, wo = _out_hw(input.shape[2], input.shape[3], kh, kw, stride, padding, dilation)
        output = input.new_empty((input.shape[0], weight.shape[0], ho, wo))
        modulated_deform_conv_cuda_forward(
            input.contiguous(), weight.contiguous(), bias, None, offset, mask, output, None, kh, kw,
            stride, stride, padding, padding, dilation, dilation, groups, deformable_groups, ctx.with_bias)
        return output

    @staticmethod
    def backward(ctx, grad_output):
        if not grad_output.is_cuda:
            raise NotImplementedError
        input, offset, mask, weight, bias = ctx.saved_tensors
        grad_input = torch.empty_like(input)
        grad_offset = torch.empty_like(offset)
        grad_mask = torch.empty_like(mask)
        grad_weight = torch.zeros_like(weight)
        grad_bias = torch.zeros_like(bias)
        kh, kw = weight.shape[2:4]
        modulated_deform_conv_cuda_backward(
            input.contiguous(), weight.contiguous(), bias, None, offset, mask, None, grad_input,
            grad_weight, grad_bias, grad_offset, grad_mask, grad_output, kh, kw, ctx.stride, ctx.stride,
            ctx.padding, ctx.padding, ctx.dilation, ctx.dilation, ctx.groups, ctx.deformable_groups,
            ctx.with_bias)
        if not ctx.with_bias:
            grad_bias = None
        return grad_input, grad_offset, grad_mask, grad_weight, grad_bias, None, None, None, None, None


modulated_deform_conv = ModulatedDeformConvFunction.apply


# ---- DCN v1 (no modulation mask, no bias): the other three entry points of the reference's pybind module ----------
# The v1 kernels sample exactly like the modulated ones (same (-1, H) x (-1, W) window and corner tests:
# deform_conv_cuda_kernel.cu:22-51,166 vs :403-432,549), so they are the same HIP operator with mask == NULL: no mask
# stream is read and no mask gradient is written.
def deform_conv_forward_cuda(input, weight, offset, output, columns, ones, kW, kH, dW, dH, padW, padH, dilationW,
                             dilationH, group, deformable_group, im2col_step):
    """Reference pybind signature deform_conv_cuda.cpp:148-153 (returns 1).  ``columns`` / ``ones`` / ``im2col_step``
    are accepted and ignored: there is no im2col scratch and every image is one launch."""
    modulated_deform_conv_cuda_forward(input, weight, None, None, offset, None, output, None, kH, kW, dH, dW, padH, padW,
                                       dilationH, dilationW, group, deformable_group, False)
    return 1


def _v1_backward(input, offset, grad_output, weight, kW, kH, dW, dH, padW, padH, dilationW, dilationH, group,
                 deformable_group):
    gi, go = torch.empty_like(input), torch.empty_like(offset)
    gw = torch.zeros_like(weight)
    modulated_deform_conv_cuda_backward(input, weight, None, None, offset, None, None, gi, gw, None, go, None, grad_output,
                                        kH, kW, dH, dW, padH, padW, dilationH, dilationW, group, deformable_group, False)
    return gi, go, gw


def deform_conv_backward_input_cuda(input, offset, gradOutput, gradInput, gradOffset, weight, columns, kW, kH, dW, dH,
                                    padW, padH, dilationW, dilationH, group, deformable_group, im2col_step):
    """Reference pybind signature deform_conv_cuda.cpp:251-257: overwrites gradInput / gradOffset (returns 1)."""
    gi, go, _ = _v1_backward(input, offset, gradOutput, weight, kW, kH, dW, dH, padW, padH, dilationW, dilationH, group,
                             deformable_group)
    gradInput.copy_(gi)
    gradOffset.copy_(go)
    return 1


def deform_conv_backward_parameters_cuda(input, offset, gradOutput, gradWeight, columns, ones, kW, kH, dW, dH, padW, padH,
                                         dilationW, dilationH, group, deformable_group, scale, im2col_step):
    """Reference pybind signature deform_conv_cuda.cpp:364-370: gradWeight += scale * dL/dW (returns 1)."""
    # dL/dW does not depend on W; the input / offset gradients computed alongside are discarded
    _, _, gw = _v1_backward(input, offset, gradOutput, torch.zeros_like(gradWeight), kW, kH, dW, dH, padW, padH, dilationW,
                            dilationH, group, deformable_group)
    gradWeight.add_(gw, alpha=float(scale))
    return 1


class DeformConvFunction(Function):
    """autograd wrapper of DCN v1 (reference thirdparty/deform_conv/functions/deform_conv.py:10-107)."""

    @staticmethod
    def forward(ctx, input, offset, weight, stride=1, padding=0, dilation=1, groups=1, deformable_groups=1, im2col_step=80):
        if input is not None and input.dim() != 4:
            raise ValueError("Expected 4D tensor as input, got {}D tensor instead.".format(input.dim()))
        if not input.is_cuda:
            raise NotImplementedError
        ctx.stride, ctx.padding, ctx.dilation = stride, padding, dilation
        ctx.groups, ctx.deformable_groups = groups, deformable_groups
        ctx.save_for_backward(input, offset, weight)
        kh, kw = weight.shape[2:4]
        ho, wo = _out_hw(input.shape[2], input.shape[3], kh, kw, stride, padding, dilation)
        if ho <= 0 or wo <= 0:
            raise ValueError("convolution input is too small (output would be {}x{})".format(ho, wo))
        output = input.new_empty((input.shape[0], weight.shape[0], ho, wo))
        deform_conv_forward_cuda(input.contiguous(), weight.contiguous(), offset, output, None, None, kw, kh, stride, stride,
                                 padding, padding, dilation, dilation, groups, deformable_groups, im2col_step)
        return output

    @staticmethod
    def backward(ctx, grad_output):
        if not grad_output.is_cuda:
            raise NotImplementedError
        input, offset, weight = ctx.saved_tensors
        kh, kw = weight.shape[2:4]
        gi, go, gw = _v1_backward(input.contiguous(), offset, grad_output, weight.contiguous(), kw, kh, ctx.stride, ctx.stride,
                                  ctx.padding, ctx.padding, ctx.dilation, ctx.dilation, ctx.groups, ctx.deformable_groups)
        return gi, go, gw, None, None, None, None, None, None


deform_conv = DeformConvFunction.apply


# ------------------------------------------------------------------------------------------------
# thin functional wrappers over the remaining C entry points (used by the engine and the tests)
# ------------------------------------------------------------------------------------------------
ACT_NONE, ACT_RELU, ACT_GELU = 0, 1, 2


class View:
    """A channel slice [coff, coff + C) of a contiguous (N, ctot, H, W) float32 tensor."""

    __slots__ = ("t", "coff", "C")

    def __init__(self, t, coff=0, C=None):
        assert t.is_contiguous() and t.dtype == torch.float32 and t.dim() == 4
        self.t, self.coff, self.C = t, coff, (t.shape[1] - coff if C is None else C)
        assert 0 <= coff and coff + self.C <= t.shape[1]

    @property
    def ctot(self):
        return self.t.shape[1]


def pack_conv_weight(weight):
    """(Cout, Cin, kh, kw) -> packed [kh*kw][Cin][Cout16] device tensor for :func:`conv2d`."""
    _require_gpu(weight)
    _check_f32(weight)
    w = weight.detach().contiguous()
    if w.dim() == 3:                      # Conv1d weight (Cout, Cin, k) with k == 1
        w = w.unsqueeze(-1)
    cout, cin, kh, kw = w.shape
    cout16 = (cout + 15) // 16 * 16
    wp = torch.empty(kh * kw * cin * cout16, dtype=torch.float32, device=w.device)
    hip.check(hip.lib().otp_conv2d_pack_weight(hip.ptr(w), hip.ptr(wp), cout, cin, kh, kw, hip.stream_of(w)),
              "otp_conv2d_pack_weight")
    return wp


def conv_desc(inp: View, out: View, cout, kh, kw, stride, pad, dil, act=ACT_NONE, in2: View = None,
              res: View = None, res_up=1, frame_split=0, cin=None):
    d = hip.ConvDesc()
    n_in, _, h, w = inp.t.shape
    d.N = out.t.shape[0]
    d.Cin = inp.C if cin is None else cin
    d.H, d.W, d.Cout, d.kh, d.kw, d.stride, d.pad, d.dil = h, w, cout, kh, kw, stride, pad, dil
    d.in_ctot, d.in_coff = inp.ctot, inp.coff
    d.in2_ctot, d.in2_coff = (in2.ctot, in2.coff) if in2 is not None else (0, 0)
    d.out_ctot, d.out_coff = out.ctot, out.coff
    d.res_ctot, d.res_coff = (res.ctot, res.coff) if res is not None else (0, 0)
    d.res_up, d.act, d.frame_split = res_up, act, frame_split
    d.Ho = (h + 2 * pad - (dil * (kh - 1) + 1)) // stride + 1
    d.Wo = (w + 2 * pad - (dil * (kw - 1) + 1)) // stride + 1
    f = max(res_up, 1)
    assert out.t.shape[2] == d.Ho * f and out.t.shape[3] == d.Wo * f, (tuple(out.t.shape), d.Ho, d.Wo, f)
    assert out.C == cout
    if res is not None:
        assert res.t.shape[2:] == out.t.shape[2:] and res.C == cout
    if in2 is not None:
        assert in2.t.shape[2:] == inp.t.shape[2:] and in2.C == d.Cin
    if frame_split:
        assert d.N == n_in * (inp.ctot // d.Cin)
    else:
        assert d.N == n_in
    return d


def conv2d_launch(inp: View, wpacked, scale, shift, out: View, desc, in2: View = None, res: View = None, stream=None):
    st = hip.lib().otp_conv2d(hip.ptr(inp.t), hip.ptr(in2.t if in2 is not None else None), hip.ptr(wpacked),
                              hip.ptr(scale), hip.ptr(shift), hip.ptr(res.t if res is not None else None),
                              hip.ptr(out.t), desc, stream if stream is not None else hip.stream_of(out.t))
    hip.check(st, "otp_conv2d")


def pack_wino_weight(weight):
    """(Cout, Cin, 3, 3) -> U[16][Cin][Cout16] = G g G^T device tensor for :func:`conv2d_wino_launch`."""
    _require_gpu(weight)
    _check_f32(weight)
    w = weight.detach().contiguous()
    cout, cin, kh, kw = w.shape
    assert (kh, kw) == (3, 3)
    L = hip.lib()
    u = torch.empty(L.otp_conv2d_wino_weight_bytes(cout, cin) // 4, dtype=torch.float32, device=w.device)
    hip.check(L.otp_conv2d_wino_pack_weight(hip.ptr(w), hip.ptr(u), cout, cin, hip.stream_of(w)), "otp_conv2d_wino_pack_weight")
    return u


def small_conv_supported(desc) -> bool:
    return bool(hip.lib().otp_conv3x3_small_supported(desc))


def pack_small_conv_weight(weight):
    """(Cout, Cin, 3, 3) -> the [ci][tap][co] image :func:`conv3x3_small` reads through the scalar cache."""
    _require_gpu(weight)
    w = weight.detach().contiguous().float()
    cout, cin = w.shape[:2]
    L = hip.lib()
    nbytes = L.otp_conv3x3_small_weight_bytes(cout, cin)
    if not nbytes:
        raise ValueError(f"otp_conv3x3_small: unsupported channel counts ({cout}, {cin})")
    wt = torch.empty(nbytes // 4, dtype=torch.float32, device=w.device)
    hip.check(L.otp_conv3x3_small_pack(hip.ptr(w), hip.ptr(wt), cout, cin, hip.stream_of(w)), "otp_conv3x3_small_pack")
    return wt


def conv3x3_small(x, weight, scale=None, shift=None, act=ACT_NONE, in2=None):
    """act(scale * conv2d(x (+ in2), weight, 3x3, stride 1, pad 1) + shift) for <= 24 channels (csrc/conv_small.hip)."""
    _require_gpu(x, weight)
    _check_f32(x)
    n, cin, h, w = x.shape
    cout = weight.shape[0]
    out = torch.empty(n, cout, h, w, dtype=torch.float32, device=x.device)
    iv, ov = View(x.contiguous()), View(out)
    i2 = View(in2.contiguous()) if in2 is not None else None
    d = conv_desc(iv, ov, cout, 3, 3, 1, 1, 1, act, i2)
    wt = pack_small_conv_weight(weight)
    hip.check(hip.lib().otp_conv3x3_small(hip.ptr(iv.t), hip.ptr(i2.t if i2 is not None else None), hip.ptr(wt), hip.ptr(scale),
                                          hip.ptr(shift), hip.ptr(out), d, hip.stream_of(x)), "otp_conv3x3_small")
    return out


def x3_weight_exponent(weight, scale=None) -> int:
    """The power of two k a layer's split-product weights are stored with: max |weight * scale| * 2^k lands in [2^13, 2^14).

    A weight is carried as two IEEE-half pieces hi + lo (csrc/common.h); `lo` is at most 2^-11 |w|, so for BatchNorm-folded
    weights of magnitude 1e-2 it is a SUBNORMAL half (spacing 2^-24) and the pair holds the weight to 2^-25 absolute - 17
    significand bits - and that error is the same for every pixel of every frame, so it does not average out over a layer's sum
    the way activation rounding does: it was the whole heat-map error of the split-product forward.  Scaled by
    2^k both pieces are normal (22 bits); the kernels multiply the accumulated sum by 2^-k (``otp_conv_desc.out_scale``; the
    pointwise kernels through their per-channel epilogue scale).  ``OTPOSE_X3_WSCALE=0`` stores the weights unscaled."""
    if os.environ.get("OTPOSE_X3_WSCALE", "1") == "0":
        return 0
    w = weight.detach()
    m = w.abs().reshape(w.shape[0], -1).amax(dim=1)
    if scale is not None:
        m = m * scale.detach().abs().to(m.device, m.dtype)
    m = float(m.max())
    if not (m > 0.0 and math.isfinite(m)):
        return 0
    return max(-40, min(40, 14 - math.frexp(m)[1]))


def _scaled_vec(scale, k, cout, device):
    """scale[cout] * 2^k as a contiguous fp32 vector (None when there is nothing to multiply by)"""
    sc = scale.detach().to(device, torch.float32).contiguous() if scale is not None else None
    if k == 0:
        return sc
    f = float(2.0 ** k)
    return sc * f if sc is not None else torch.full((cout,), f, dtype=torch.float32, device=device)


def pack_x3_weight(weight, scale=None, stride=1, k=0):
    """(Cout, Cin, k, k) fp32, k = 3 or 1 (times scale[cout] * 2^k) -> half hi / lo MFMA fragments for :func:`conv2d_x3_launch`
    (the chunking depends on the kernel size and the stride); the descriptor's ``out_scale`` must then be 2^-k
    (:func:`x3_weight_exponent`)."""
    _require_gpu(weight)
    _check_f32(weight)
    w = weight.detach().contiguous()
    if w.dim() == 3:
        w = w.unsqueeze(-1)
    cout, cin, kh, kw = w.shape
    assert kh == kw and kh in (1, 3)
    L = hip.lib()
    nbytes = L.otp_conv2d_x3_weight_bytes(cout, cin, kh, stride)
    if not nbytes:
        raise ValueError(f"otp_conv2d_x3: unsupported channel counts ({cout}, {cin})")
    u = torch.empty(nbytes // 4, dtype=torch.int32, device=w.device)
    sc = _scaled_vec(scale, k, cout, w.device)
    hip.check(L.otp_conv2d_x3_pack_weight(hip.ptr(w), hip.ptr(sc), hip.ptr(u), cout, cin, kh, stride, hip.stream_of(w)),
              "otp_conv2d_x3_pack_weight")
    return u


def x3_supported(desc) -> bool:
    return bool(hip.lib().otp_conv2d_x3_supported(desc))


def conv2d_x3_launch(inp: View, wpacked, shift, out: View, desc, res: View = None, stream=None):
    st = hip.lib().otp_conv2d_x3(hip.ptr(inp.t), hip.ptr(wpacked), hip.ptr(shift), hip.ptr(res.t if res is not None else None),
                                 hip.ptr(out.t), desc, stream if stream is not None else hip.stream_of(out.t))
    hip.check(st, "otp_conv2d_x3")


def conv2d_x3(x, weight, scale=None, shift=None, act=ACT_NONE, res=None, pad=1, dil=1, stride=1):
    """act(conv2d(x, weight, stride, pad, dil) * scale + shift + res) with split-half products (csrc/convx.hip)."""
    _require_gpu(x, weight)
    n, cin, h, w = x.shape
    cout, k = weight.shape[0], weight.shape[2]
    ke = dil * (k - 1)
    ho, wo = (h + 2 * pad - ke - 1) // stride + 1, (w + 2 * pad - ke - 1) // stride + 1
    out = torch.empty(n, cout, ho, wo, dtype=torch.float32, device=x.device)
    iv, ov = View(x.contiguous()), View(out)
    rv = View(res.contiguous()) if res is not None else None
    d = conv_desc(iv, ov, cout, k, k, stride, pad, dil, act, None, rv)
    e = x3_weight_exponent(weight, scale)
    d.out_scale = 2.0 ** -e
    conv2d_x3_launch(iv, pack_x3_weight(weight, scale, stride, e), shift, ov, d, rv)
    return out


# ---- split-record (S8) activations and the LDS-DMA fed 3x3 convolution (csrc/convs.hip) --------------------------------
S8_F32_C4, S8_F32_NCHW = 1, 2


def s8_empty(n, c, h, w, device):
    """Storage of the S8 image of a logical (n, c, h, w) fp32 tensor: [n][c/8][2][h*w] records of 8 bf16 (hi | lo)."""
    nbytes = hip.lib().otp_s8_bytes(n, c, h, w)
    if not nbytes:
        raise ValueError(f"S8 images need a channel count that is a multiple of 8, got {c}")
    return torch.empty(nbytes // 4, dtype=torch.int32, device=device)


def c4_empty(n, c, h, w, device):
    """Storage of the C4 image [n][c/4][h*w][4] fp32 of a logical (n, c, h, w) tensor."""
    assert c % 4 == 0
    return torch.empty(n * c * h * w, dtype=torch.float32, device=device)


def s8_pack(inp, out=None, out_c4=None, stream=None):
    """fp32 NCHW tensor or channel-slice :class:`View` -> S8 image (hi = rne_bf16(x), lo = rne_bf16(x - hi)) and, when
    ``out_c4`` is given, the C4 image of the same values."""
    iv = inp if isinstance(inp, View) else View(inp.contiguous())
    _require_gpu(iv.t)
    n, _, h, w = iv.t.shape
    out = s8_empty(n, iv.C, h, w, iv.t.device) if out is None else out
    hip.check(hip.lib().otp_s8_pack(hip.ptr(iv.t), hip.ptr(out), hip.ptr(out_c4), n, iv.C, h, w, iv.ctot, iv.coff,
                                    stream if stream is not None else hip.stream_of(iv.t)), "otp_s8_pack")
    return out


def s8_unpack(s8, n, c, h, w):
    """S8 image -> fp32 (n, c, h, w) = hi + lo (tests / debugging)."""
    _require_gpu(s8)
    out = torch.empty(n, c, h, w, dtype=torch.float32, device=s8.device)
    hip.check(hip.lib().otp_s8_unpack(hip.ptr(s8), hip.ptr(out), n, c, h, w, hip.stream_of(s8)), "otp_s8_unpack")
    return out


def c4_unpack(c4, n, c, h, w):
    """C4 image -> fp32 NCHW (tests / debugging)."""
    _require_gpu(c4)
    out = torch.empty(n, c, h, w, dtype=torch.float32, device=c4.device)
    hip.check(hip.lib().otp_c4_unpack(hip.ptr(c4), hip.ptr(out), n, c, h, w, hip.stream_of(c4)), "otp_c4_unpack")
    return out


def s8_conv_supported(desc) -> bool:
    return bool(hip.lib().otp_conv3x3_s8_supported(desc))


def s8_conv_desc(n, cin, cout, h, w, act=ACT_NONE, out: View = None):
    d = hip.ConvDesc()
    d.N, d.Cin, d.H, d.W, d.Cout = n, cin, h, w, cout
    d.kh = d.kw = 3
    d.stride, d.pad, d.dil = 1, 1, 1
    d.in_ctot, d.in_coff, d.in2_ctot, d.in2_coff = cin, 0, 0, 0
    d.out_ctot, d.out_coff = (out.ctot, out.coff) if out is not None else (cout, 0)
    d.res_ctot, d.res_coff, d.res_up = 0, 0, 1
    d.act, d.Ho, d.Wo, d.frame_split = act, h, w, 0
    return d


def s8_s2_conv_desc(n, cin, cout, h, w, act=ACT_NONE, out: View = None, res: View = None):
    """Descriptor of the stride-2 S8 convolution (csrc/convs2.hip): (n, cin, h, w) -> (n, cout, h / 2, w / 2)."""
    d = hip.ConvDesc()
    d.N, d.Cin, d.H, d.W, d.Cout = n, cin, h, w, cout
    d.kh = d.kw = 3
    d.stride, d.pad, d.dil = 2, 1, 1
    d.in_ctot, d.in_coff, d.in2_ctot, d.in2_coff = cin, 0, 0, 0
    d.out_ctot, d.out_coff = (out.ctot, out.coff) if out is not None else (cout, 0)
    d.res_ctot, d.res_coff = (res.ctot, res.coff) if res is not None else (0, 0)
    d.res_up = 1
    d.act, d.Ho, d.Wo, d.frame_split = act, h // 2, w // 2, 0
    return d


def s8_s2_conv_supported(desc, nchw_out=True):
    return bool(hip.lib().otp_conv3x3_s2_s8_supported(desc, int(nchw_out)))


def conv3x3_s2_s8(x_s8, shape, weight, scale=None, shift=None, act=ACT_NONE, res=None, out="nchw"):
    """act(conv2d(x, weight, 3x3, stride 2, pad 1) * scale + shift (+ res)) from the S8 image ``x_s8`` of logical ``shape``
    (n, cin, h, w).  ``out = "nchw"``: fp32 (n, cout, h / 2, w / 2) tensor (``res`` an fp32 tensor of that shape or None);
    ``out = "s8"``: the S8 image of the result (no residual)."""
    _require_gpu(x_s8, weight)
    n, cin, h, w = shape
    cout = weight.shape[0]
    e = x3_weight_exponent(weight, scale)
    wp = pack_s8_weight(weight, scale, e)
    sh = shift.detach().contiguous().float() if shift is not None else None
    L = hip.lib()
    if out == "s8":
        assert res is None
        d = s8_s2_conv_desc(n, cin, cout, h, w, act)
        d.out_scale = 2.0 ** -e
        o8 = s8_empty(n, cout, h // 2, w // 2, x_s8.device)
        hip.check(L.otp_conv3x3_s2_s8(hip.ptr(x_s8), hip.ptr(wp), hip.ptr(sh), None, None, hip.ptr(o8), d, hip.stream_of(x_s8)),
                  "otp_conv3x3_s2_s8")
        return o8
    o = torch.empty(n, cout, h // 2, w // 2, dtype=torch.float32, device=x_s8.device)
    rv = View(res.contiguous()) if res is not None else None
    d = s8_s2_conv_desc(n, cin, cout, h, w, act, View(o), rv)
    d.out_scale = 2.0 ** -e
    hip.check(L.otp_conv3x3_s2_s8(hip.ptr(x_s8), hip.ptr(wp), hip.ptr(sh), hip.ptr(rv.t) if rv is not None else None, hip.ptr(o),
                                  None, d, hip.stream_of(x_s8)), "otp_conv3x3_s2_s8")
    return o


def pack_s8_weight(weight, scale=None, k=0):
    """(Cout, Cin, 3, 3) fp32 (times scale[cout] * 2^k) -> half hi / lo MFMA fragments for :func:`conv3x3_s8_launch` and
    :func:`conv3x3_s2_s8`; the descriptor's ``out_scale`` must then be 2^-k (:func:`x3_weight_exponent`)."""
    _require_gpu(weight)
    _check_f32(weight)
    w = weight.detach().contiguous()
    cout, cin, kh, kw = w.shape
    assert kh == kw == 3
    L = hip.lib()
    nbytes = L.otp_conv3x3_s8_weight_bytes(cout, cin)
    if not nbytes:
        raise ValueError(f"otp_conv3x3_s8: unsupported channel counts ({cout}, {cin})")
    u = torch.empty(nbytes // 4, dtype=torch.int32, device=w.device)
    sc = _scaled_vec(scale, k, cout, w.device)
    hip.check(L.otp_conv3x3_s8_pack_weight(hip.ptr(w), hip.ptr(sc), hip.ptr(u), cout, cin, hip.stream_of(w)),
              "otp_conv3x3_s8_pack_weight")
    return u


def conv3x3_s8_launch(in_s8, wpacked, shift, desc, res_c4=None, out_f32=None, f32_layout=S8_F32_C4, out_s8=None, stream=None):
    st = hip.lib().otp_conv3x3_s8(hip.ptr(in_s8), hip.ptr(wpacked), hip.ptr(shift), hip.ptr(res_c4), hip.ptr(out_f32),
                                  f32_layout, hip.ptr(out_s8), desc, stream if stream is not None else hip.stream_of(in_s8))
    hip.check(st, "otp_conv3x3_s8")


def conv3x3_s8(x_s8, shape, weight, scale=None, shift=None, act=ACT_NONE, res_c4=None, f32="nchw", want_s8=True, res_s8=None):
    """act(conv2d(x, weight, 3x3, stride 1, pad 1) * scale + shift + res) from an S8 image ``x_s8`` of logical ``shape``
    (n, cin, h, w); ``res_c4`` a C4 image of the residual.  Returns (fp32 result - an NCHW tensor for f32 = "nchw", a C4 image
    for "c4", None for None -, S8 image of the result or None)."""
    _require_gpu(x_s8, weight)
    n, cin, h, w = shape
    cout = weight.shape[0]
    out = out_s8 = None
    layout = S8_F32_C4
    if f32 == "nchw":
        out, layout = torch.empty(n, cout, h, w, dtype=torch.float32, device=x_s8.device), S8_F32_NCHW
    elif f32 == "c4":
        out = c4_empty(n, cout, h, w, x_s8.device)
    if want_s8:
        out_s8 = s8_empty(n, cout, h, w, x_s8.device)
    d = s8_conv_desc(n, cin, cout, h, w, act)
    e = x3_weight_exponent(weight, scale)
    d.out_scale = 2.0 ** -e
    if res_s8 is not None:                           # the residual as S8 records (its hi + lo) instead of a C4 fp32 image
        assert res_c4 is None
        d.res_layout = 1
    conv3x3_s8_launch(x_s8, pack_s8_weight(weight, scale, e), shift, d, res_s8 if res_s8 is not None else res_c4, out, layout,
                      out_s8)
    return out, out_s8


def ln_mlp_fused(y, gamma, beta, eps, packed, scale, shift, out=None, hid=None, stream=None):
    """out = y + scale * (W2 . gelu(W1 . LN(y) + b1)) + shift: ln2 + MLP + residual of TransformerBlock.forward
    (model/blocks.py:277-279) in one launch."""
    _require_gpu(y, packed)
    _check_f32(y)
    b, c, t = y.shape
    hid = 4 * c if hid is None else hid
    out = torch.empty_like(y) if out is None else out
    hip.check(hip.lib().otp_ln_mlp_fused(hip.ptr(y), hip.ptr(gamma), hip.ptr(beta), eps, hip.ptr(packed), hip.ptr(scale),
                                         hip.ptr(shift), hip.ptr(out), b, c, hid, t,
                                         stream if stream is not None else hip.stream_of(y)), "otp_ln_mlp_fused")
    return out


def dense_cc_supported(c, t) -> bool:
    return bool(hip.lib().otp_dense_cc_supported(int(c), int(t)))


def dense_x3_supported(c, t) -> bool:
    return bool(hip.lib().otp_dense_x3_supported(int(c), int(t)))


def pack_dense_cc(weight, scale=None, shift=None, x3=False, grad=False):
    """(C, C[, 1]) pointwise weight (+ per-output-channel scale / shift) -> the per-16-row fragment image of
    :func:`dense_cc` (MaskedMHCA query / key / value / proj, model/blocks.py:383-386).  ``x3``: the split-half image of
    csrc/densex.hip (pass the same flag to :func:`dense_cc` / :func:`qkv_front`); ``grad`` (with ``x3``): the image with
    bfloat16 pieces for :func:`dense_cc` on operands of unknown magnitude - gradients (csrc/densex_grad.hip)."""
    _require_gpu(weight)
    c = weight.shape[0]
    L = hip.lib()
    nbytes = (L.otp_dense_x3_weight_bytes if x3 else L.otp_dense_cc_weight_bytes)(c)
    if not nbytes or weight.shape[1] != c:
        raise RuntimeError(f"otp_dense_cc: unsupported weight shape {tuple(weight.shape)}")
    f = lambda t: None if t is None else t.detach().to(weight.device, torch.float32).contiguous()   # noqa: E731
    w, sc, sh = f(weight), f(scale), f(shift)
    packed = torch.empty(nbytes // 4, dtype=torch.float32, device=weight.device)
    pack = (L.otp_dense_x3_pack_bf16p if grad else L.otp_dense_x3_pack) if x3 else L.otp_dense_cc_pack
    hip.check(pack(hip.ptr(w), hip.ptr(sc), hip.ptr(sh), hip.ptr(packed), c, hip.stream_of(w)), "otp_dense_cc_pack")
    return packed


def dense_cc_args(xs, packs, ress, outs):
    """ctypes pointer arrays for :func:`dense_cc_launch` (kept by the caller while launches may still be issued)."""
    n = len(xs)
    arr = lambda ts: (ctypes.c_void_p * n)(*[hip.ptr(t) for t in ts])     # noqa: E731
    return arr(xs), arr(packs), arr(ress if ress is not None else [None] * n), arr(outs)


def dense_cc(xs, packs, ress=None, outs=None, stream=None, x3=False, grad=False, half=False):
    """out[p] = scale[p] * (W[p] . x[p]) + shift[p] (+ res[p]) for up to three (B, C, T) problems in one launch (``grad``: see
    :func:`pack_dense_cc`; ``half`` with ``x3``: the fp16 engine's arithmetic - operands rounded to half once, otp_dense_h1)."""
    _require_gpu(*xs)
    b, c, t = xs[0].shape
    outs = [torch.empty_like(x) for x in xs] if outs is None else outs
    ax, ap, ar, ao = dense_cc_args(xs, packs, ress, outs)
    fn = (hip.lib().otp_dense_x3_bf16p if grad else (hip.lib().otp_dense_h1 if half else hip.lib().otp_dense_x3)) if x3 \
        else hip.lib().otp_dense_cc
    hip.check(fn(ax, ap, ar, ao, len(xs), b, c, t, stream if stream is not None else hip.stream_of(xs[0])), "otp_dense_cc")
    return outs


def stem_conv_x3_supported(b, f, h, w, cout) -> bool:
    return bool(hip.lib().otp_stem_conv_x3_supported(int(b), int(f), int(h), int(w), int(cout)))


def pack_stem_conv_x3(weight, scale=None, shift=None):
    """(Cout, 3, 3, 3) weight of HRNet's first conv (+ folded BatchNorm) -> the register image of :func:`stem_conv_x3`."""
    _require_gpu(weight)
    cout = weight.shape[0]
    L = hip.lib()
    nbytes = L.otp_stem_conv_x3_weight_bytes(cout)
    if not nbytes or tuple(weight.shape[1:]) != (3, 3, 3):
        raise RuntimeError(f"otp_stem_conv_x3: unsupported weight shape {tuple(weight.shape)}")
    f = lambda t: None if t is None else t.detach().to(weight.device, torch.float32).contiguous()   # noqa: E731
    w, sc, sh = f(weight), f(scale), f(shift)
    packed = torch.empty(nbytes // 4, dtype=torch.float32, device=weight.device)
    hip.check(L.otp_stem_conv_x3_pack(hip.ptr(w), hip.ptr(sc), hip.ptr(sh), hip.ptr(packed), cout, hip.stream_of(w)),
              "otp_stem_conv_x3_pack")
    return packed


def stem_conv_x3(clip, packed, cout, frames=5, out=None, stream=None):
    """relu(bn(conv3x3 stride 2 pad 1)) of the 3-channel frames of ``clip`` (B, 3 * frames, H, W) -> (frames * B, cout, Ho, Wo),
    frame-major like model/OTPose.py:317."""
    _require_gpu(clip)
    b, c, h, w = clip.shape
    assert c == 3 * frames and clip.is_contiguous() and clip.dtype == torch.float32
    ho, wo = (h - 1) // 2 + 1, (w - 1) // 2 + 1
    out = torch.empty(frames * b, cout, ho, wo, dtype=torch.float32, device=clip.device) if out is None else out
    hip.check(hip.lib().otp_stem_conv_x3(hip.ptr(clip), hip.ptr(packed), hip.ptr(out), b, frames, h, w, cout,
                                         stream if stream is not None else hip.stream_of(clip)), "otp_stem_conv_x3")
    return out


def pointwise_x3_supported(cin, cout, t) -> bool:
    return bool(hip.lib().otp_pointwise_x3_supported(int(cin), int(cout), int(t)))


def pack_pointwise_x3(weight, scale=None, shift=None):
    """(Cout, Cin[, 1, 1]) weight (+ per-output-channel scale / shift, e.g. a folded BatchNorm) -> the block image of
    :func:`pointwise_x3` (HRNet layer1's 1x1 convs, model/HRNet.py:551-571)."""
    _require_gpu(weight)
    cout, cin = weight.shape[:2]
    L = hip.lib()
    nbytes = L.otp_pointwise_x3_weight_bytes(cin, cout)
    if not nbytes:
        raise RuntimeError(f"otp_pointwise_x3: unsupported weight shape {tuple(weight.shape)}")
    f = lambda t: None if t is None else t.detach().to(weight.device, torch.float32).contiguous()   # noqa: E731
    w, sc, sh = f(weight).reshape(cout, cin), f(scale), f(shift)
    e = x3_weight_exponent(w)                     # weights stored times 2^e, undone by the kernel's per-channel epilogue scale
    if e:
        w, sc = w * float(2.0 ** e), _scaled_vec(sc, -e, cout, w.device)
    packed = torch.empty(nbytes // 4, dtype=torch.float32, device=weight.device)
    hip.check(L.otp_pointwise_x3_pack(hip.ptr(w), hip.ptr(sc), hip.ptr(sh), hip.ptr(packed), cin, cout, hip.stream_of(w)),
              "otp_pointwise_x3_pack")
    return packed


def pointwise_x3(x: View, packed, out: View, res: View = None, relu=False, stream=None):
    """out = act(scale * (W . x) + shift (+ res)) over channel-slice views of (B, ctot, H, W) fp32 tensors."""
    _require_gpu(x.t, out.t)
    b = x.t.shape[0]
    t = x.t.shape[2] * x.t.shape[3]
    hip.check(hip.lib().otp_pointwise_x3(hip.ptr(x.t), hip.ptr(packed), hip.ptr(res.t if res is not None else None), hip.ptr(out.t),
                                         b, x.C, out.C, t, x.ctot, x.coff, res.ctot if res is not None else 0,
                                         res.coff if res is not None else 0, out.ctot, out.coff, int(bool(relu)),
                                         stream if stream is not None else hip.stream_of(x.t)), "otp_pointwise_x3")
    return out


def pointwise_x3_s8_supported(cin, cout, t) -> bool:
    return bool(hip.lib().otp_pointwise_x3_s8_supported(int(cin), int(cout), int(t)))


def pack_pointwise_x3_s8(weight, scale=None, shift=None):
    """Weight image of :func:`pointwise_x3_s8` (its own row order: pairs of 16-row tiles interleaved by groups of 4)."""
    _require_gpu(weight)
    cout, cin = weight.shape[:2]
    L = hip.lib()
    nbytes = L.otp_pointwise_x3_s8_weight_bytes(cin, cout)
    if not nbytes:
        raise RuntimeError(f"otp_pointwise_x3_s8: unsupported weight shape {tuple(weight.shape)}")
    f = lambda t: None if t is None else t.detach().to(weight.device, torch.float32).contiguous()   # noqa: E731
    w, sc, sh = f(weight).reshape(cout, cin), f(scale), f(shift)
    e = x3_weight_exponent(w)                     # weights stored times 2^e, undone by the kernel's per-channel epilogue scale
    if e:
        w, sc = w * float(2.0 ** e), _scaled_vec(sc, -e, cout, w.device)
    packed = torch.empty(nbytes // 4, dtype=torch.float32, device=weight.device)
    hip.check(L.otp_pointwise_x3_s8_pack(hip.ptr(w), hip.ptr(sc), hip.ptr(sh), hip.ptr(packed), cin, cout, hip.stream_of(w)),
              "otp_pointwise_x3_s8_pack")
    return packed


def pointwise_x3_s8(x: View, packed, cout, out_s8=None, relu=False, stream=None, res: View = None):
    """S8 image of act(scale * (W . x) + shift (+ res)): a 1x1 conv feeding :func:`conv3x3_s8_launch` without an fp32 round trip
    (``res``: an fp32 NCHW channel-slice view, a Bottleneck's conv3)."""
    _require_gpu(x.t)
    b, _, h, w = x.t.shape
    out_s8 = s8_empty(b, cout, h, w, x.t.device) if out_s8 is None else out_s8
    hip.check(hip.lib().otp_pointwise_x3_s8_res(hip.ptr(x.t), hip.ptr(packed), hip.ptr(res.t if res is not None else None),
                                                hip.ptr(out_s8), b, x.C, cout, h * w, x.ctot, x.coff,
                                                res.ctot if res is not None else 0, res.coff if res is not None else 0,
                                                int(bool(relu)), stream if stream is not None else hip.stream_of(x.t)),
              "otp_pointwise_x3_s8")
    return out_s8


def pack_qkv_table(dwq, dwk, dwv, gq, bq, gk, bk, gv, bv):
    """Depthwise (C, 1, 3) weights and LayerNorm (C) gamma / beta of MaskedMHCA's query / key / value paths
    (model/blocks.py:359-381) -> the per-channel table of :func:`qkv_front`."""
    ts = [t.detach().float().contiguous() for t in (dwq, dwk, dwv, gq, bq, gk, bk, gv, bv)]
    _require_gpu(*ts)
    c = ts[3].numel()
    L = hip.lib()
    table = torch.empty(L.otp_qkv_front_table_bytes(c) // 4, dtype=torch.float32, device=ts[0].device)
    hip.check(L.otp_qkv_front_pack_table(*[hip.ptr(t) for t in ts], hip.ptr(table), c, hip.stream_of(ts[0])),
              "otp_qkv_front_pack_table")
    return table


def qkv_front(x, table, packs, eps=1e-5, outs=None, stream=None, x3=False, half=False):
    """q, k, v = W_p . LN_p(dwconv3_p(x)) + b_p (stride 1) in one launch; ``packs`` = three :func:`pack_dense_cc` images
    (``half`` with ``x3``: operands rounded to half once, otp_qkv_front_h1)."""
    _require_gpu(x, table)
    b, c, t = x.shape
    outs = [torch.empty_like(x) for _ in range(3)] if outs is None else outs
    fn = (hip.lib().otp_qkv_front_h1 if half else hip.lib().otp_qkv_front_x3) if x3 else hip.lib().otp_qkv_front
    hip.check(fn(hip.ptr(x), hip.ptr(table), *[hip.ptr(p) for p in packs], *[hip.ptr(o) for o in outs],
                 b, c, t, eps, stream if stream is not None else hip.stream_of(x)), "otp_qkv_front")
    return outs


def mlp_fused_supported(c, hid, t) -> bool:
    return bool(hip.lib().otp_mlp_fused_supported(int(c), int(hid), int(t)))


def pack_mlp_weights(w1, b1, w2):
    """Conv1d(C,4C,1) / Conv1d(4C,C,1) weights of a TransformerBlock MLP (model/blocks.py:248-254) -> the per-hidden-block
    fragment image :func:`mlp_fused` streams through LDS."""
    _require_gpu(w1, b1, w2)
    hid, c = w1.shape[:2]
    L = hip.lib()
    nbytes = L.otp_mlp_fused_weight_bytes(c, hid)
    if not nbytes:
        raise RuntimeError(f"otp_mlp_fused: unsupported widths C={c}, HID={hid}")
    packed = torch.empty(nbytes // 4, dtype=torch.float32, device=w1.device)
    w1c, w2c, b1c = (t.detach().contiguous().float() for t in (w1, w2, b1))
    hip.check(L.otp_mlp_fused_pack(hip.ptr(w1c), hip.ptr(b1c), hip.ptr(w2c), hip.ptr(packed), c, hid, hip.stream_of(w1c)),
              "otp_mlp_fused_pack")
    return packed


def mlp_fused(x, packed, scale, shift, res, out=None, hid=None, stream=None):
    """out = res + scale * (W2 . gelu(W1 . x + b1)) + shift on (B, C, T) tensors, one launch (eval-mode MLP half of
    TransformerBlock.forward, model/blocks.py:277-279)."""
    _require_gpu(x, packed, res)
    _check_f32(x)
    b, c, t = x.shape
    hid = 4 * c if hid is None else hid
    out = torch.empty_like(x) if out is None else out
    hip.check(hip.lib().otp_mlp_fused(hip.ptr(x), hip.ptr(packed), hip.ptr(scale), hip.ptr(shift), hip.ptr(res),
                                      hip.ptr(out), b, c, hid, t, stream if stream is not None else hip.stream_of(x)),
              "otp_mlp_fused")
    return out


def dcn_fused_supported(cin, j, h, w, nd) -> bool:
    return bool(hip.lib().otp_dcn_fused_supported(int(cin), int(j), int(h), int(w), int(nd)))


def pack_dcn_fused(w_offs, w_masks, w_dcns, biases):
    """Per-dilation offset / mask conv weights ((18 J, 32, 3, 3) / (9 J, 32, 3, 3)), DCN weights (J, J, 3, 3) and biases (J or
    None) -> the packed image of :func:`dcn_fused` (csrc/dcn_fused.hip)."""
    nd, j = len(w_offs), w_dcns[0].shape[0]
    dev = w_offs[0].device
    _require_gpu(*w_offs, *w_masks, *w_dcns)
    keep = [[t.detach().contiguous().float() for t in ts] for ts in (w_offs, w_masks, w_dcns)]
    keep.append([None if b is None else b.detach().contiguous().float() for b in biases])
    ptrs = [torch.tensor([0 if t is None else t.data_ptr() for t in ts], dtype=torch.int64, device=dev) for ts in keep]
    L = hip.lib()
    nbytes = L.otp_dcn_fused_weight_bytes(nd, j)
    if not nbytes:
        raise RuntimeError(f"otp_dcn_fused: unsupported ND={nd}, J={j}")
    packed = torch.empty(nbytes // 4, dtype=torch.int32, device=dev)
    hip.check(L.otp_dcn_fused_pack(*[hip.ptr(t) for t in ptrs], hip.ptr(packed), nd, j, hip.stream_of(packed)),
              "otp_dcn_fused_pack")
    torch.cuda.current_stream(dev).synchronize()          # the pointer arrays and fp32 copies die with this frame
    return packed


def dcn_fused(trans, x, packed, dilations, alpha, out=None, workspace=None, stream=None):
    """out = alpha * sum over dilations of ModulatedDeformConv(x, Conv_off(trans), Conv_mask(trans)) + bias (SURVEY.md
    section 8 row f-2; model/OTPose.py:381-392) in one launch; the offsets and masks never reach HBM."""
    _require_gpu(trans, x, packed)
    _check_f32(trans)
    b, cin, h, w = trans.shape
    j = x.shape[1]
    L = hip.lib()
    out = torch.empty_like(x) if out is None else out
    nws = L.otp_dcn_fused_workspace(b, h, w)
    workspace = torch.empty(nws // 4, dtype=torch.int32, device=x.device) if workspace is None else workspace
    dl = (ctypes.c_int * len(dilations))(*[int(d) for d in dilations])
    hip.check(L.otp_dcn_fused_forward(hip.ptr(trans), hip.ptr(x), hip.ptr(packed), hip.ptr(out), hip.ptr(workspace), nws, b, cin, j,
                                      h, w, dl, len(dilations), float(alpha),
                                      stream if stream is not None else hip.stream_of(x)), "otp_dcn_fused_forward")
    return out


def mlp_x3_supported(c, hid, t) -> bool:
    return bool(hip.lib().otp_mlp_x3_supported(int(c), int(hid), int(t)))


def pack_mlp_x3_weights(w1, b1, w2, half=False):
    """The same MLP weights as :func:`pack_mlp_weights`, split into half hi / lo MFMA fragments per 32 hidden channels
    (csrc/mlpx.hip); ``half``: the hi-only image of the fp16 engine's form (``ln_mlp_x3(..., half=True)``)."""
    _require_gpu(w1, b1, w2)
    hid, c = w1.shape[:2]
    L = hip.lib()
    nbytes = (L.otp_mlp_h1_weight_bytes if half else L.otp_mlp_x3_weight_bytes)(c, hid)
    if not nbytes:
        raise RuntimeError(f"otp_mlp_x3: unsupported widths C={c}, HID={hid}")
    packed = torch.empty(nbytes // 4, dtype=torch.int32, device=w1.device)
    w1c, w2c, b1c = (t.detach().contiguous().float() for t in (w1, w2, b1))
    hip.check((L.otp_mlp_h1_pack if half else L.otp_mlp_x3_pack)(hip.ptr(w1c), hip.ptr(b1c), hip.ptr(w2c), hip.ptr(packed), c, hid,
                                                                 hip.stream_of(w1c)), "otp_mlp_x3_pack")
    return packed


def mlp_x3(x, packed, scale, shift, res, out=None, hid=None, stream=None):
    """:func:`mlp_fused` with split-half products on the 16-bit matrix cores (fp32 storage / accumulation)."""
    _require_gpu(x, packed, res)
    _check_f32(x)
    b, c, t = x.shape
    hid = 4 * c if hid is None else hid
    out = torch.empty_like(x) if out is None else out
    hip.check(hip.lib().otp_mlp_x3(hip.ptr(x), hip.ptr(packed), hip.ptr(scale), hip.ptr(shift), hip.ptr(res),
                                   hip.ptr(out), b, c, hid, t, stream if stream is not None else hip.stream_of(x)),
              "otp_mlp_x3")
    return out


def ln_mlp_x3(y, gamma, beta, eps, packed, scale, shift, out=None, hid=None, stream=None, half=False):
    """:func:`ln_mlp_fused` with split-half products (``half``: operands and the hidden layer rounded to half once, otp_ln_mlp_h1)."""
    _require_gpu(y, packed)
    _check_f32(y)
    b, c, t = y.shape
    hid = 4 * c if hid is None else hid
    out = torch.empty_like(y) if out is None else out
    fn = hip.lib().otp_ln_mlp_h1 if half else hip.lib().otp_ln_mlp_x3
    hip.check(fn(hip.ptr(y), hip.ptr(gamma), hip.ptr(beta), eps, hip.ptr(packed), hip.ptr(scale),
                                      hip.ptr(shift), hip.ptr(out), b, c, hid, t,
                                      stream if stream is not None else hip.stream_of(y)), "otp_ln_mlp_x3")
    return out


def wino_supported(desc) -> bool:
    return bool(hip.lib().otp_conv2d_wino_supported(desc))


def conv2d_wino_launch(inp: View, upacked, scale, shift, out: View, desc, res: View = None, stream=None):
    st = hip.lib().otp_conv2d_wino(hip.ptr(inp.t), hip.ptr(upacked), hip.ptr(scale), hip.ptr(shift),
                                   hip.ptr(res.t if res is not None else None), hip.ptr(out.t), desc,
                                   stream if stream is not None else hip.stream_of(out.t))
    hip.check(st, "otp_conv2d_wino")


def conv2d_wino(x, weight, scale=None, shift=None, act=ACT_NONE, res=None):
    """3x3 / stride 1 / pad 1 convolution through the Winograd F(2x2, 3x3) kernel, fresh output."""
    _require_gpu(x, weight)
    cout = weight.shape[0]
    out = torch.empty((x.shape[0], cout, x.shape[2], x.shape[3]), dtype=torch.float32, device=x.device)
    iv, ov = View(x.contiguous()), View(out)
    rv = View(res.contiguous()) if res is not None else None
    d = conv_desc(iv, ov, cout, 3, 3, 1, 1, 1, act, None, rv, 1)
    conv2d_wino_launch(iv, pack_wino_weight(weight), scale, shift, ov, d, rv)
    return out


def conv2d(x, weight, scale=None, shift=None, stride=1, pad=0, dil=1, act=ACT_NONE, res=None, in2=None, res_up=1):
    """Convenience form: out = act(scale * conv(x (+ in2), weight) + shift (+ res)), fresh output."""
    _require_gpu(x, weight)
    w4 = weight if weight.dim() == 4 else weight.unsqueeze(-1)
    cout, cin, kh, kw = w4.shape
    h, w = x.shape[2:]
    ho = (h + 2 * pad - (dil * (kh - 1) + 1)) // stride + 1
    wo = (w + 2 * pad - (dil * (kw - 1) + 1)) // stride + 1
    out = torch.empty((x.shape[0], cout, ho * max(res_up, 1), wo * max(res_up, 1)), dtype=torch.float32, device=x.device)
    iv, ov = View(x.contiguous()), View(out)
    rv = View(res.contiguous()) if res is not None else None
    i2 = View(in2.contiguous()) if in2 is not None else None
    d = conv_desc(iv, ov, cout, kh, kw, stride, pad, dil, act, i2, rv, res_up)
    conv2d_launch(iv, pack_conv_weight(w4), scale, shift, ov, d, i2, rv)
    return out


def upsample_add(low, res, f, relu=False, out=None):
    """out = act(res + nearest_upsample_f(low)) (HRNet fuse accumulate for f >= 4); ``out`` may be ``res``."""
    _require_gpu(low, res)
    n, c, hl, wl = low.shape
    if out is None:
        out = torch.empty_like(res)
    hip.check(hip.lib().otp_upsample_add(hip.ptr(low), hip.ptr(res), hip.ptr(out), n, c, hl, wl, f, int(relu),
                                         c, 0, c, 0, c, 0, hip.stream_of(low)), "otp_upsample_add")
    return out


def upsample_add_multi(lows, res, relu=False, out=None):
    """out = act(res + sum_k nearest_upsample(lows[k])) in one pass (an HRNet fuse row's upsampled terms, summed in list
    order); each ``lows[k]`` is (N, C, H / f_k, W / f_k) with f_k a power of two >= 2."""
    _require_gpu(res, *lows)
    n, c, hh, wh = res.shape
    out = torch.empty_like(res) if out is None else out
    lows = [t.contiguous() for t in lows]
    lp = (ctypes.c_void_p * len(lows))(*[hip.ptr(t) for t in lows])
    fp = (ctypes.c_int * len(lows))(*[hh // t.shape[2] for t in lows])
    hip.check(hip.lib().otp_upsample_add_multi(lp, fp, len(lows), hip.ptr(res), hip.ptr(out), n, c, hh, wh, int(relu),
                                               c, 0, c, 0, hip.stream_of(res)), "otp_upsample_add_multi")
    return out


def ln_channel(x, gamma, beta, eps=1e-5, pool=False):
    _require_gpu(x)
    b, c, t = x.shape
    y = torch.empty_like(x)
    p = torch.empty((b, c, (t + 2 - 3) // 2 + 1), dtype=x.dtype, device=x.device) if pool else None
    hip.check(hip.lib().otp_ln_channel(hip.ptr(x), hip.ptr(gamma), hip.ptr(beta), hip.ptr(y), hip.ptr(p), b, c, t,
                                       eps, hip.stream_of(x)), "otp_ln_channel")
    return (y, p) if pool else y


def dwconv_ln3(x, dw, gammas, betas, stride, eps=1e-5):
    """dw / gammas / betas: triples for (query, key, value)."""
    _require_gpu(x)
    b, c, t = x.shape
    to = (t + 2 - 3) // stride + 1
    outs = [torch.empty((b, c, to), dtype=x.dtype, device=x.device) for _ in range(3)]
    hip.check(hip.lib().otp_dwconv_ln3(
        hip.ptr(x), hip.ptr(dw[0]), hip.ptr(dw[1]), hip.ptr(dw[2]), hip.ptr(gammas[0]), hip.ptr(betas[0]),
        hip.ptr(gammas[1]), hip.ptr(betas[1]), hip.ptr(gammas[2]), hip.ptr(betas[2]), hip.ptr(outs[0]),
        hip.ptr(outs[1]), hip.ptr(outs[2]), b, c, t, stride, eps, hip.stream_of(x)), "otp_dwconv_ln3")
    return outs


def chan_attn(q, k, v, n_head, scale):
    _require_gpu(q, k, v)
    b, c, t = q.shape
    out = torch.empty_like(q)
    L = hip.lib()
    nbytes = L.otp_chan_attn_workspace(b, c, t, n_head)
    ws = torch.empty(max(nbytes // 4, 1), dtype=torch.float32, device=q.device)
    hip.check(L.otp_chan_attn(hip.ptr(q), hip.ptr(k), hip.ptr(v), hip.ptr(out), hip.ptr(ws), nbytes, b, c, t, n_head,
                              scale, hip.stream_of(q)), "otp_chan_attn")
    return out


def upsample_linear(x, f, out=None, out_coff=0):
    _require_gpu(x)
    b, c, t = x.shape
    if out is None:
        out = torch.empty((b, c, t * f), dtype=x.dtype, device=x.device)
    hip.check(hip.lib().otp_upsample_linear(hip.ptr(x), hip.ptr(out), b, c, t, f, out.shape[1], out_coff,
                                            hip.stream_of(x)), "otp_upsample_linear")
    return out


def get_max_preds(batch_heatmaps):
    """Device form of reference utils/heatmap.py:143-171: (N,J,H,W) heat-maps -> (preds (N,J,2), maxvals (N,J,1)),
    no device-to-host copy of the maps."""
    return _decode(batch_heatmaps, None, None, refine=False)


def get_final_preds(batch_heatmaps, center=None, scale=None):
    """Device form of reference utils/heatmap.py:108-132: argmax, +-0.25 px shift towards the higher neighbour, and
    (when ``center`` / ``scale`` (N,2) are given) the rot = 0 inverse crop transform of transform_preds."""
    return _decode(batch_heatmaps, center, scale, refine=True)


def _decode(hm, center, scale, refine):
    _require_gpu(hm)
    _check_f32(hm)
    if hm.dim() != 4:
        raise AssertionError("batch_images should be 4-ndim")
    hm = hm.contiguous()
    n, j, h, w = hm.shape
    preds = torch.empty((n, j, 2), dtype=torch.float32, device=hm.device)
    maxvals = torch.empty((n, j, 1), dtype=torch.float32, device=hm.device)
    c = s_ = None
    if center is not None:
        c = torch.as_tensor(center, dtype=torch.float32, device=hm.device).reshape(n, 2).contiguous()
        s_ = torch.as_tensor(scale, dtype=torch.float32, device=hm.device).reshape(n, 2).contiguous()
    hip.check(hip.lib().otp_heatmap_decode(hip.ptr(hm), hip.ptr(preds), hip.ptr(maxvals), hip.ptr(c), hip.ptr(s_),
                                           n, j, h, w, int(refine), hip.stream_of(hm)), "otp_heatmap_decode")
    return preds, maxvals


def accuracy(output, target, hm_type="gaussian", thr=0.5):
    """Device form of reference utils/evaluate.py:384-415 (called per iteration at script/Common.py:147-150 on heat-maps
    copied to the host): PCK of the argmax of ``output`` against the argmax of ``target``.  Returns
    ``(acc (J+1), avg_acc, cnt, pred)`` like the reference, as device tensors (no synchronisation)."""
    if hm_type != "gaussian":
        raise NotImplementedError("only hm_type='gaussian' (the reference's only caller)")
    pred, _ = get_max_preds(output)
    tgt, _ = get_max_preds(target)
    n, j, h, w = output.shape
    acc = torch.empty(j + 1, dtype=torch.float32, device=output.device)
    cnt = torch.empty(1, dtype=torch.int32, device=output.device)
    hip.check(hip.lib().otp_pck_accuracy(hip.ptr(pred), hip.ptr(tgt), hip.ptr(acc), hip.ptr(cnt), n, j, h, w, float(thr),
                                         hip.stream_of(output)), "otp_pck_accuracy")
    return acc, acc[0], cnt[0], pred


IMAGENET_MEAN = (0.485, 0.456, 0.406)        # utils/transform.py:7-8
IMAGENET_STD = (0.229, 0.224, 0.225)


def frames_to_clip(frames_u8, mean=IMAGENET_MEAN, std=IMAGENET_STD, out=None):
    """uint8 RGB frames (B, F, H, W, 3) (the five warped crops of dataset/PoseTrackDataset.py:390-399) -> the model input
    (B, 3F, H, W) float32: ToTensor + Normalize per frame (utils/transform.py:11-15) and the channel concat of
    script/Common.py:117 in one kernel, bit-identical to the torchvision float32 arithmetic.  Moves 4x fewer bytes over
    PCIe than shipping normalised floats."""
    _require_gpu(frames_u8)
    if frames_u8.dtype != torch.uint8 or frames_u8.dim() != 5 or frames_u8.shape[-1] != 3:
        raise TypeError("frames_u8 must be a (B, F, H, W, 3) uint8 tensor")
    frames_u8 = frames_u8.contiguous()
    b, f, h, w, _ = frames_u8.shape
    if out is None:
        out = torch.empty((b, 3 * f, h, w), dtype=torch.float32, device=frames_u8.device)
    elif out.shape != (b, 3 * f, h, w) or out.dtype != torch.float32 or not out.is_contiguous():
        raise ValueError("out must be a contiguous float32 (B, 3F, H, W) tensor")
    hip.check(hip.lib().otp_frames_u8_to_clip(hip.ptr(frames_u8), hip.ptr(out), b, f, h, w, *[float(v) for v in mean],
                                              *[float(v) for v in std], hip.stream_of(frames_u8)),
              "otp_frames_u8_to_clip")
    return out


def st_ohkw_loss(s, t, g, w, topk=8, flags=None, with_grad=False):
    """ST_OHKW_MSELoss forward (+ analytic gradients) on the GPU; returns dict like the reference
    (model/loss.py:89-91) plus ``flags`` and, when requested, ``grad_s`` / ``grad_t``."""
    _require_gpu(s, t, g, w)
    b, j = s.shape[:2]
    hw = s[0, 0].numel()
    s, t, g = s.contiguous(), t.contiguous(), g.contiguous()
    wv = w.reshape(b, j).contiguous().float()
    L = hip.lib()
    nbytes = L.otp_loss_workspace(b, j)
    ws = torch.empty(nbytes // 4, dtype=torch.float32, device=s.device)
    res = torch.empty(3, dtype=torch.float32, device=s.device)
    given = flags is not None
    fl = flags.to(torch.int32).contiguous() if given else torch.empty(j, dtype=torch.int32, device=s.device)
    gs = torch.empty_like(s) if with_grad else None
    gt = torch.empty_like(t) if with_grad else None
    hip.check(L.otp_loss_st_ohkw(hip.ptr(s), hip.ptr(t), hip.ptr(g), hip.ptr(wv), hip.ptr(fl), hip.ptr(res),
                                 hip.ptr(gs), hip.ptr(gt), hip.ptr(ws), nbytes, b, j, hw, topk, int(given),
                                 hip.stream_of(s)), "otp_loss_st_ohkw")
    out = {"ohkm_loss_s": res[0], "mse_loss_s": res[1], "final_loss": res[2], "flags": fl}
    if with_grad:
        out["grad_s"], out["grad_t"] = gs, gt
    return out


def _joints_mse(o, g, w, topk, ohkm, effective_num_joints, with_grad):
    _require_gpu(o, g)
    b, j = o.shape[:2]
    hw = o[0, 0].numel()
    o, g = o.contiguous(), g.contiguous()
    wv = None if w is None else w.reshape(b, j).contiguous().float()
    L = hip.lib()
    nbytes = L.otp_loss_workspace(b, j)
    ws = torch.empty(nbytes // 4, dtype=torch.float32, device=o.device)
    res = torch.empty(3, dtype=torch.float32, device=o.device)
    go = torch.empty_like(o) if with_grad else None
    hip.check(L.otp_loss_joints_mse(hip.ptr(o), hip.ptr(g), hip.ptr(wv), hip.ptr(res), hip.ptr(go), hip.ptr(ws), nbytes,
                                    b, j, hw, topk, int(ohkm), int(effective_num_joints or 0), hip.stream_of(o)),
              "otp_loss_joints_mse")
    return res, go


def joints_ohkm_mse_loss(output, target, target_weight=None, effective_num_joints=None, topk=8, with_grad=False):
    """JointsMSE_OHKMMSELoss.forward (model/loss.py:115-148) on the GPU; ``target_weight=None`` is the reference's
    ``use_target_weight=False``.  Returns the reference's dict (+ ``grad_output`` of ``final_loss`` when requested)."""
    res, go = _joints_mse(output, target, target_weight, topk, True, effective_num_joints, with_grad)
    out = {"ohkm_loss": res[0], "mse_loss": res[1], "final_loss": res[2]}
    if with_grad:
        out["grad_output"] = go
    return out


def joint_mse_loss(output, target, target_weight=None, effective_num_joints=None, with_grad=False):
    """JointMSELoss.forward (model/loss.py:158-182) on the GPU: a scalar (and its gradient when requested)."""
    res, go = _joints_mse(output, target, target_weight, 1, False, effective_num_joints, with_grad)
    return (res[2], go) if with_grad else res[2]


# ---- flow-encoder TransformerBlock (C = 17) as two launches around the channel attention (csrc/flowenc.hip) ---------------
def flow_block_supported(blk, c, t) -> bool:
    """Stride-1 TransformerBlock with 17 channels, biases everywhere and one epsilon for ln1 and the q / k / v norms."""
    a = blk.attn
    try:
        eps = {float(m.eps) for m in (blk.ln1, a.query_norm, a.key_norm, a.value_norm)}
        has_bias = all(m.bias is not None for m in (a.query, a.key, a.value, a.proj, blk.mlp[0], blk.mlp[3]))
        dw_ok = all(tuple(m.weight.shape) == (c, 1, 3) and m.bias is None for m in (a.query_conv, a.key_conv, a.value_conv))
    except AttributeError:
        return False
    return (len(eps) == 1 and has_bias and dw_ok
            and bool(hip.lib().otp_flow_block_supported(int(c), int(blk.mlp[0].out_channels), int(t))))


def pack_flow_front(blk, device):
    """Parameter block of ``otp_flow_front`` (layout: include/otpose_hip.h) from a TransformerBlock's modules."""
    a = blk.attn
    f = lambda t: t.detach().to(device, torch.float32).reshape(-1)                     # noqa: E731
    parts = [f(blk.ln1.weight), f(blk.ln1.bias)]
    for conv, norm, proj in ((a.query_conv, a.query_norm, a.query), (a.key_conv, a.key_norm, a.key),
                             (a.value_conv, a.value_norm, a.value)):
        parts += [f(conv.weight), f(norm.weight), f(norm.bias), f(proj.weight), f(proj.bias)]
    prm = torch.cat(parts).contiguous()
    c = blk.ln1.weight.numel()
    assert prm.numel() == hip.lib().otp_flow_front_param_floats(c), (prm.numel(), c)
    return prm


def pack_flow_back(blk, device):
    """Parameter block of ``otp_flow_back``: the drop-path scales (model/blocks.py:298-301, eval: plain per-channel factors)
    folded into the projection / down-projection rows and biases; W_2 transposed to (hidden, C)."""
    a = blk.attn
    d = lambda t: t.detach().to(device, torch.float32)                                 # noqa: E731
    c = blk.ln1.weight.numel()
    hid = blk.mlp[0].out_channels
    sa, sm = d(blk.drop_path_attn.scale).reshape(-1), d(blk.drop_path_mlp.scale).reshape(-1)
    wp = d(a.proj.weight).reshape(c, c) * sa[:, None]
    w2 = d(blk.mlp[3].weight).reshape(c, hid) * sm[:, None]
    parts = [wp.reshape(-1), d(a.proj.bias) * sa, d(blk.ln2.weight).reshape(-1), d(blk.ln2.bias).reshape(-1),
             d(blk.mlp[0].weight).reshape(-1), d(blk.mlp[0].bias), w2.t().contiguous().reshape(-1), d(blk.mlp[3].bias) * sm]
    prm = torch.cat(parts).contiguous()
    assert prm.numel() == hip.lib().otp_flow_back_param_floats(c, hid), (prm.numel(), c, hid)
    return prm


def flow_front(x, prm, eps=1e-5):
    """q, k, v of a flow-encoder block from its input (B, 17, T)."""
    _require_gpu(x, prm)
    _check_f32(x)
    x = x.contiguous()
    b, c, t = x.shape
    q, k, v = torch.empty_like(x), torch.empty_like(x), torch.empty_like(x)
    hip.check(hip.lib().otp_flow_front(hip.ptr(x), hip.ptr(prm), hip.ptr(q), hip.ptr(k), hip.ptr(v), b, c, t, float(eps),
                                       hip.stream_of(x)), "otp_flow_front")
    return q, k, v


def flow_back(x, att, prm, hidden, eps=1e-5):
    """Block output from its input ``x`` and the attention output ``att`` (both (B, 17, T))."""
    _require_gpu(x, att, prm)
    _check_f32(x)
    x, att = x.contiguous(), att.contiguous()
    b, c, t = x.shape
    out = torch.empty_like(x)
    hip.check(hip.lib().otp_flow_back(hip.ptr(x), hip.ptr(att), hip.ptr(prm), hip.ptr(out), b, c, int(hidden), t, float(eps),
                                      hip.stream_of(x)), "otp_flow_back")
    return out


# ---- fp16-storage eval kernels of the backbone (csrc/h16.hip; cfg.MODEL.DTYPE = "fp16") ------------------------------------
class H8:
    """An H8 image: ``t`` = int32 storage of [N][gtot][H * W] 16-byte records (8 halves = channels 8 g .. 8 g + 7 of a pixel),
    of which this tensor is the channel groups [goff, goff + C / 8) - the fp16 engine's activation format (include/otpose_hip.h)."""

    __slots__ = ("t", "N", "C", "H", "W", "gtot", "goff")

    def __init__(self, t, n, c, h, w, gtot=None, goff=0):
        assert c % 8 == 0
        self.t, self.N, self.C, self.H, self.W = t, n, c, h, w
        self.gtot, self.goff = (c // 8 if gtot is None else gtot), goff
        assert 0 <= self.goff and self.goff + c // 8 <= self.gtot

    def slice(self, coff, c):
        assert coff % 8 == 0 and c % 8 == 0 and coff + c <= self.C
        return H8(self.t, self.N, c, self.H, self.W, self.gtot, self.goff + coff // 8)


def h8_empty(n, c, h, w, device):
    nbytes = hip.lib().otp_h8_bytes(n, c, h, w)
    if not nbytes:
        raise ValueError(f"H8 images need a channel count that is a multiple of 8, got {c}")
    return H8(torch.empty(nbytes // 4, dtype=torch.int32, device=device), n, c, h, w)


def h8_pack(x, out: H8 = None, stream=None):
    """fp32 NCHW tensor or channel-slice :class:`View` -> H8 (one rounding to half per element)."""
    iv = x if isinstance(x, View) else View(x.contiguous())
    _require_gpu(iv.t)
    n, _, h, w = iv.t.shape
    out = h8_empty(n, iv.C, h, w, iv.t.device) if out is None else out
    hip.check(hip.lib().otp_h8_pack(hip.ptr(iv.t), hip.ptr(out.t), n, iv.C, h, w, iv.ctot, iv.coff, out.gtot, out.goff,
                                    stream if stream is not None else hip.stream_of(iv.t)), "otp_h8_pack")
    return out


def h8_unpack(img: H8, stream=None):
    """H8 image -> fp32 (N, C, H, W), exact."""
    out = torch.empty(img.N, img.C, img.H, img.W, dtype=torch.float32, device=img.t.device)
    hip.check(hip.lib().otp_h8_unpack(hip.ptr(img.t), hip.ptr(out), img.N, img.C, img.H, img.W, img.gtot, img.goff,
                                      stream if stream is not None else hip.stream_of(img.t)), "otp_h8_unpack")
    return out


def h16_weight_exponent(weight, scale=None) -> int:
    """Power of two k the fp16 engine stores a layer's (BatchNorm-folded) weights with: max |w * scale| * 2^k in [2^13, 2^14), so
    weights down to 2^-27 of the largest stay normal halves; the launch multiplies its fp32 sums by 2^-k (exact)."""
    w = weight.detach()
    m = w.abs().reshape(w.shape[0], -1).amax(dim=1)
    if scale is not None:
        m = m * scale.detach().abs().to(m.device, m.dtype)
    m = float(m.max())
    if not (m > 0.0 and math.isfinite(m)):
        return 0
    return max(-40, min(40, 14 - math.frexp(m)[1]))


def h16_conv_desc(x: H8, cout, stride, act=ACT_NONE, out: H8 = None, res: H8 = None, k=0):
    d = hip.H16ConvDesc()
    d.N, d.Cin, d.H, d.W, d.Cout, d.stride, d.act = x.N, x.C, x.H, x.W, cout, stride, act
    d.in_gtot, d.in_goff = x.gtot, x.goff
    d.out_gtot, d.out_goff = (out.gtot, out.goff) if out is not None else (0, 0)
    d.res_gtot, d.res_goff = (res.gtot, res.goff) if res is not None else (0, 0)
    d.out_scale = 2.0 ** -k
    return d


def h16_conv_supported(desc) -> bool:
    return bool(hip.lib().otp_h16_conv3x3_supported(desc))


def pack_h16_conv_weight(weight, scale=None, k=0):
    """(Cout, Cin, 3, 3) fp32 (x scale[cout] x 2^k) -> the half A-fragment image of csrc/h16.hip."""
    _require_gpu(weight)
    w = weight.detach().contiguous().float()
    cout, cin, kh, kw = w.shape
    assert (kh, kw) == (3, 3)
    L = hip.lib()
    nbytes = L.otp_h16_conv3x3_weight_bytes(cout, cin)
    if not nbytes:
        raise RuntimeError(f"otp_h16_conv3x3: unsupported widths {cin} -> {cout}")
    packed = torch.zeros(nbytes // 4, dtype=torch.int32, device=w.device)
    sc = scale.detach().to(w.device, torch.float32).contiguous() if scale is not None else None
    hip.check(L.otp_h16_conv3x3_pack_weight(hip.ptr(w), hip.ptr(sc), hip.ptr(packed), cout, cin, float(2.0 ** k), hip.stream_of(w)),
              "otp_h16_conv3x3_pack_weight")
    return packed


def h16_conv3x3(x: H8, wpacked, shift, cout, stride=1, act=ACT_NONE, res: H8 = None, out: H8 = None, k=0, stream=None, desc=None):
    """out = act(conv3x3 pad 1 (x) + shift (+ res)) on H8 images (csrc/h16.hip)."""
    ho, wo = x.H // stride, x.W // stride
    out = h8_empty(x.N, cout, ho, wo, x.t.device) if out is None else out
    d = desc if desc is not None else h16_conv_desc(x, cout, stride, act, out, res, k)
    hip.check(hip.lib().otp_h16_conv3x3(hip.ptr(x.t), hip.ptr(wpacked), hip.ptr(shift), hip.ptr(res.t if res is not None else None),
                                        hip.ptr(out.t), d, stream if stream is not None else hip.stream_of(x.t)), "otp_h16_conv3x3")
    return out


def h16_pointwise_supported(cin, cout) -> bool:
    return bool(hip.lib().otp_h16_pointwise_supported(int(cin), int(cout)))


def pack_h16_pointwise(weight, scale=None, shift=None, k=0):
    """(Cout, Cin[, 1, 1]) fp32 (x scale x 2^k) + shift -> the packed image of otp_h16_pointwise."""
    _require_gpu(weight)
    w = weight.detach().reshape(weight.shape[0], -1).contiguous().float()
    cout, cin = w.shape
    L = hip.lib()
    nbytes = L.otp_h16_pointwise_weight_bytes(cin, cout)
    if not nbytes:
        raise RuntimeError(f"otp_h16_pointwise: unsupported widths {cin} -> {cout}")
    packed = torch.zeros(nbytes // 4, dtype=torch.int32, device=w.device)
    sc = scale.detach().to(w.device, torch.float32).contiguous() if scale is not None else None
    sh = shift.detach().to(w.device, torch.float32).contiguous() if shift is not None else None
    hip.check(L.otp_h16_pointwise_pack(hip.ptr(w), hip.ptr(sc), hip.ptr(sh), hip.ptr(packed), cin, cout, float(2.0 ** k),
                                       hip.stream_of(w)), "otp_h16_pointwise_pack")
    return packed


def h16_pointwise(x: H8, packed, cout, relu=False, res: H8 = None, out=None, k=0, stream=None):
    """out = act(W x + shift (+ res)): ``out`` an :class:`H8` (default: fresh) or a :class:`View` of an fp32 NCHW tensor."""
    f32 = isinstance(out, View)
    if out is None:
        out = h8_empty(x.N, cout, x.H, x.W, x.t.device)
    otot, ooff = (out.ctot, out.coff) if f32 else (out.gtot, out.goff)
    hip.check(hip.lib().otp_h16_pointwise(hip.ptr(x.t), hip.ptr(packed), hip.ptr(res.t if res is not None else None), hip.ptr(out.t),
                                          int(f32), x.N, x.C, cout, x.H * x.W, x.gtot, x.goff,
                                          res.gtot if res is not None else 0, res.goff if res is not None else 0, otot, ooff,
                                          int(relu), float(2.0 ** -k), stream if stream is not None else hip.stream_of(x.t)),
              "otp_h16_pointwise")
    return out


def pack_h16_stem(weight, scale=None, shift=None):
    _require_gpu(weight)
    w = weight.detach().contiguous().float()
    cout = w.shape[0]
    L = hip.lib()
    nbytes = L.otp_h16_stem_weight_bytes(cout)
    if not nbytes:
        raise RuntimeError(f"otp_h16_stem: unsupported width {cout}")
    packed = torch.zeros(nbytes // 4, dtype=torch.int32, device=w.device)
    sc = scale.detach().to(w.device, torch.float32).contiguous() if scale is not None else None
    sh = shift.detach().to(w.device, torch.float32).contiguous() if shift is not None else None
    hip.check(L.otp_h16_stem_pack(hip.ptr(w), hip.ptr(sc), hip.ptr(sh), hip.ptr(packed), cout, hip.stream_of(w)), "otp_h16_stem_pack")
    return packed


def h16_stem(clip, packed, cout, frames, out: H8 = None, stream=None):
    """relu(conv3x3 s2 p1 of the frames of the fp32 clip (B, 3 F, H, W) + shift) as the H8 image of (F B, cout, Ho, Wo)."""
    _require_gpu(clip)
    b, c, h, w = clip.shape
    assert c == 3 * frames and clip.is_contiguous()
    ho, wo = (h - 1) // 2 + 1, (w - 1) // 2 + 1
    out = h8_empty(frames * b, cout, ho, wo, clip.device) if out is None else out
    hip.check(hip.lib().otp_h16_stem(hip.ptr(clip), hip.ptr(packed), hip.ptr(out.t), b, frames, h, w, cout,
                                     stream if stream is not None else hip.stream_of(clip)), "otp_h16_stem")
    return out


def h16_upsample_add(lows, factors, res: H8, relu=True, out: H8 = None, stream=None):
    """out = act(res + sum_k nearest_up(lows[k], factors[k])) on dense H8 images."""
    assert res.gtot * 8 == res.C and all(l.gtot * 8 == l.C for l in lows)
    out = h8_empty(res.N, res.C, res.H, res.W, res.t.device) if out is None else out
    lp = (ctypes.c_void_p * len(lows))(*[hip.ptr(l.t) for l in lows])
    fp = (ctypes.c_int * len(lows))(*[int(f) for f in factors])
    hip.check(hip.lib().otp_h16_upsample_add(lp, fp, len(lows), hip.ptr(res.t), hip.ptr(out.t), res.N, res.C, res.H, res.W, int(relu),
                                             stream if stream is not None else hip.stream_of(res.t)), "otp_h16_upsample_add")
    return out
