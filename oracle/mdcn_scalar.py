"""CPU ORACLE - test infrastructure only.  ctypes loader for oracle/_build/libmdcn_scalar.so
(scalar C restatement of the reference DCN kernels, see mdcn_scalar.c)."""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libmdcn_scalar.so")
_lib = None


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def _load():
    global _lib
    if _lib is None:
        if not os.path.isfile(_SO):
            build()
        _lib = ctypes.CDLL(_SO)
    return _lib


def _p(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _suffix(t):
    return {torch.float32: "f32", torch.float64: "f64"}[t.dtype]


def forward(x, offset, mask, weight, bias, stride, pad, dil, groups, dg):
    x, offset, mask, weight = (t.contiguous() for t in (x, offset, mask, weight))
    N, C, H, W = x.shape
    Co, _, kh, kw = weight.shape
    Ho = (H + 2 * pad - (dil * (kh - 1) + 1)) // stride + 1
    Wo = (W + 2 * pad - (dil * (kw - 1) + 1)) // stride + 1
    out = torch.empty((N, Co, Ho, Wo), dtype=x.dtype)
    fn = getattr(_load(), "mdcn_forward_" + _suffix(x))
    rc = fn(_p(x), _p(offset), _p(mask), _p(weight), _p(bias), _p(out), N, C, H, W, Co, kh, kw,
            stride, pad, dil, groups, dg)
    assert rc == 0
    return out


def backward(x, offset, mask, weight, grad_out, stride, pad, dil, groups, dg, with_bias=True):
    x, offset, mask, weight, grad_out = (t.contiguous() for t in (x, offset, mask, weight, grad_out))
    N, C, H, W = x.shape
    Co, _, kh, kw = weight.shape
    gx, goff, gmask = torch.empty_like(x), torch.empty_like(offset), torch.empty_like(mask)
    gw = torch.zeros_like(weight)
    gb = torch.zeros(Co, dtype=x.dtype) if with_bias else None
    fn = getattr(_load(), "mdcn_backward_" + _suffix(x))
    rc = fn(_p(x), _p(offset), _p(mask), _p(weight), _p(grad_out), _p(gx), _p(goff), _p(gmask),
            _p(gw), _p(gb), N, C, H, W, Co, kh, kw, stride, pad, dil, groups, dg)
    assert rc == 0
    return gx, goff, gmask, gw, gb
