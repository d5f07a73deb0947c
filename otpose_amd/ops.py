"""Operator-level Python API over the C-ABI (mirror of the reference's native operator boundary).

``modulated_deform_conv`` replaces ``thirdparty.deform_conv.modulated_deform_conv``
(reference thirdparty/deform_conv/functions/deform_conv.py:109-179): same argument order, same
``NotImplementedError`` for CPU tensors (functions/deform_conv.py:131,149), same gradient tuple.
``modulated_deform_conv_cuda_forward`` / ``_backward`` keep the pybind entry points' names and
argument lists (reference thirdparty/deform_conv/src/deform_conv_cuda.cpp:474-480, 551-558) for
callers that bind the native module directly.
"""
from __future__ import annotations

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


def modulated_deform_conv_cuda_forward(input, weight, bias, ones, offset, mask, output, columns,
                                       kernel_h, kernel_w, stride_h, stride_w, pad_h, pad_w,
                                       dilation_h, dilation_w, group, deformable_group, with_bias):
    """In-place forward with the reference pybind signature (deform_conv_cuda.cpp:474-480).
    ``ones`` and ``columns`` are accepted and ignored: the fused kernel needs no im2col scratch."""
    _require_gpu(input, weight, offset, mask, output)
    _check_f32(input, weight, offset, mask, output)
    if not input.is_contiguous():
        raise RuntimeError("input tensor has to be contiguous")
    if not weight.is_contiguous():
        raise RuntimeError("weight tensor has to be contiguous")
    if stride_h != stride_w or pad_h != pad_w or dilation_h != dilation_w:
        raise RuntimeError("otp_mdcn_forward: anisotropic stride/pad/dilation is not supported")
    n, c, h, w = input.shape
    cout, cpg, kh_, kw_ = weight.shape
    if (kh_, kw_) != (kernel_h, kernel_w):
        raise RuntimeError(f"Input shape and kernel shape wont match: ({kernel_h} x {kernel_w} vs {kh_} x {kw_}).")
    if c != cpg * group:
        raise RuntimeError(f"Input shape and kernel channels wont match: ({c} vs {cpg * group}).")
    offset = offset.contiguous()
    mask = mask.contiguous()
    b = bias if with_bias else None
    st = hip.lib().otp_mdcn_forward(
        hip.ptr(input), hip.ptr(offset), hip.ptr(mask), hip.ptr(weight), hip.ptr(b), hip.ptr(output),
        n, c, h, w, cout, kernel_h, kernel_w, stride_h, pad_h, dilation_h, group, deformable_group,
        1.0, 0.0, _DTYPE_F32, hip.stream_of(input))
    hip.check(st, "otp_mdcn_forward")


def modulated_deform_conv_cuda_backward(input, weight, bias, ones, offset, mask, columns,
                                        grad_input, grad_weight, grad_bias, grad_offset, grad_mask,
                                        grad_output, kernel_h, kernel_w, stride_h, stride_w, pad_h,
                                        pad_w, dilation_h, dilation_w, group, deformable_group, with_bias):
    """In-place backward with the reference pybind signature (deform_conv_cuda.cpp:551-558).
    grad_input/grad_offset/grad_mask are overwritten; grad_weight/grad_bias are accumulated into
    (the reference accumulates them over the batch with addmm_, cpp:638-650)."""
    _require_gpu(input, weight, offset, mask, grad_output)
    _check_f32(input, weight, offset, mask, grad_output)
    if not input.is_contiguous():
        raise RuntimeError("input tensor has to be contiguous")
    if not weight.is_contiguous():
        raise RuntimeError("weight tensor has to be contiguous")
    if stride_h != stride_w or pad_h != pad_w or dilation_h != dilation_w:
        raise RuntimeError("otp_mdcn_backward: anisotropic stride/pad/dilation is not supported")
    n, c, h, w = input.shape
    cout = weight.shape[0]
    offset = offset.contiguous()
    mask = mask.contiguous()
    grad_output = grad_output.contiguous()
    L = hip.lib()
    ws_bytes = L.otp_mdcn_backward_workspace(n, c, h, w, cout, kernel_h, kernel_w)
    ws = torch.empty(max(int(ws_bytes), 4) // 4, dtype=torch.float32, device=input.device)
    st = L.otp_mdcn_backward(
        hip.ptr(input), hip.ptr(offset), hip.ptr(mask), hip.ptr(weight), hip.ptr(grad_output),
        hip.ptr(grad_input), hip.ptr(grad_offset), hip.ptr(grad_mask), hip.ptr(grad_weight),
        hip.ptr(grad_bias if with_bias else None), hip.ptr(ws), ws_bytes,
        n, c, h, w, cout, kernel_h, kernel_w, stride_h, pad_h, dilation_h, group, deformable_group,
        _DTYPE_F32, hip.stream_of(input))
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
