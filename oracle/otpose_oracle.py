"""CPU ORACLE - test infrastructure only.  NOT part of the product path.

A CPU restatement (PyTorch CPU tensor arithmetic, fp32 or fp64) of the reference's OTPose hot path,
written functionally over a plain ``state_dict`` so that it shares no code with ``otpose_amd``.
Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it, and only as the checker.

Pinning (see DESIGN.md "Oracle"):
* every function except the DCN arithmetic is pinned by importing the reference itself in the build
  container (tests/golden/make_golden.py): the reference modules are run on the same seeded
  weights/inputs and the outputs are committed under tests/golden/ - tests/test_oracle_golden.py
  checks this file against those vectors.
* the modulated-DCN arithmetic (``mdcn_forward`` / ``mdcn_backward``) has NO executable reference
  here: its only implementation is CUDA source (thirdparty/deform_conv/src/*.cu) that needs nvcc
  and ATen/THC headers, so it is unbuildable in this image.  **DCN parity is unpinned by execution**;
  it is anchored by (a) this vectorised restatement, (b) an independent scalar C restatement
  (oracle/mdcn_scalar.c) that follows the kernels statement by statement, (c) the known answer
  "zero offsets, unit mask == dilated conv2d", and (d) fp64 autograd-vs-analytic gradient checks.

Each function cites the reference lines it restates (paths relative to /root/reference).
"""
from __future__ import annotations

import math
from typing import Dict, List, Sequence, Tuple

import torch
import torch.nn.functional as F

SD = Dict[str, torch.Tensor]
BN_EPS = 1e-5


# ------------------------------------------------------------------------------------------------
# bf16 storage points: a yardstick for the bf16 TRAINING path of the product (BASELINE configs[2])
# ------------------------------------------------------------------------------------------------
# The reference has no bf16 mode, so a bf16 step cannot be compared with it directly: against an fp32 / fp64
# graph the bf16 activations (2^-9 per stored value) flip ~1 % of the ReLU gates and move the whole gradient by 0.1 relative L2
# - a bound that would not notice a wrong kernel.  Inside ``with bf16_points():`` this restatement rounds to bfloat16 exactly
# where otpose_amd/train.py::TrainGraphBF16 stores bfloat16 - the HRNet input, every HRNet conv result (BatchNorm statistics
# are taken from the rounded values, as csrc/nhwc.hip does), every BatchNorm (+ residual) (+ ReLU) result, every fuse-row
# accumulation, the MLP interior of the temporal encoders, the input of the offset / mask convs, the weights of those
# layers (fp32 masters, straight-through) - and rounds the GRADIENTS at the same tensors in the backward pass (the product's
# input-gradient convs and BatchNorm backward kernels store bf16).  Accumulation stays in the oracle's precision, as the
# product accumulates in fp32; everything outside those layers is untouched.  What is left between the two is summation
# order (1e-7) and the bf16 roundings it flips.
_BF16_POINTS = False


class bf16_points:
    def __enter__(self):
        global _BF16_POINTS
        self.prev, _BF16_POINTS = _BF16_POINTS, True
        return self

    def __exit__(self, *exc):
        global _BF16_POINTS
        _BF16_POINTS = self.prev
        return False


def _bf(x):
    return x.to(torch.bfloat16).to(x.dtype)


class _RoundBoth(torch.autograd.Function):
    """value and gradient both stored as bfloat16"""

    @staticmethod
    def forward(ctx, x):
        return _bf(x)

    @staticmethod
    def backward(ctx, g):
        return _bf(g)


class _RoundGrad(torch.autograd.Function):
    """fp32 value whose gradient is converted to bfloat16 before the layer's backward kernels run"""

    @staticmethod
    def forward(ctx, x):
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return _bf(g)


def _rb(x):
    return _RoundBoth.apply(x) if _BF16_POINTS else x


def _rg(x):
    return _RoundGrad.apply(x) if _BF16_POINTS else x


def _rbw(w):
    """bf16 copy of an fp32 master weight; the weight gradient (fp32 in the product) passes straight through"""
    return w + (_bf(w) - w).detach() if _BF16_POINTS else w


# ------------------------------------------------------------------------------------------------
# building blocks
# ------------------------------------------------------------------------------------------------
def _bn(sd: SD, p: str, x, training=False):
    """nn.BatchNorm2d (eval: running stats; train: batch stats, biased var) - torch semantics."""
    if training:
        return F.batch_norm(x, None, None, sd[p + ".weight"], sd[p + ".bias"], True, 0.0, BN_EPS)
    return F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"], sd[p + ".weight"],
                        sd[p + ".bias"], False, 0.0, BN_EPS)


def _conv(sd: SD, p: str, x, stride=1, pad=0, dil=1):
    return F.conv2d(x, sd[p + ".weight"], sd.get(p + ".bias"), stride, pad, dil)


def _conv_bn(sd, conv, bn, x, stride=1, pad=0, relu=False, training=False, res=None):
    if _BF16_POINTS:
        # csrc/nhwc.hip: bf16 operands, fp32 accumulation, the result stored as bf16 (statistics from the stored values), then
        # one pass y = act(c * scale + shift (+ res)) stored as bf16; ``_rb(x)`` rounds this consumer's input gradient
        c = _rb(F.conv2d(_rb(x), _rbw(sd[conv + ".weight"]), None, stride, pad))
        y = _bn(sd, bn, c, training)
        if res is not None:
            y = y + res
        return _rb(F.relu(y) if relu else y)
    assert res is None
    y = _bn(sd, bn, _conv(sd, conv, x, stride, pad), training)
    return F.relu(y) if relu else y


def basic_block(sd: SD, p: str, x, training=False):
    """model/HRNet.py:514-530"""
    if _BF16_POINTS:
        y = _conv_bn(sd, p + ".conv1", p + ".bn1", x, 1, 1, True, training)
        res = x
        if (p + ".downsample.0.weight") in sd:
            res = _conv_bn(sd, p + ".downsample.0", p + ".downsample.1", x, 1, 0, False, training)
        return _conv_bn(sd, p + ".conv2", p + ".bn2", y, 1, 1, True, training, res=res)
    y = _conv_bn(sd, p + ".conv1", p + ".bn1", x, 1, 1, True, training)
    y = _conv_bn(sd, p + ".conv2", p + ".bn2", y, 1, 1, False, training)
    res = x
    if (p + ".downsample.0.weight") in sd:
        res = _conv_bn(sd, p + ".downsample.0", p + ".downsample.1", x, 1, 0, False, training)
    return F.relu(y + res)


def bottleneck(sd: SD, p: str, x, training=False):
    """model/HRNet.py:551-571"""
    y = _conv_bn(sd, p + ".conv1", p + ".bn1", x, 1, 0, True, training)
    y = _conv_bn(sd, p + ".conv2", p + ".bn2", y, 1, 1, True, training)
    if _BF16_POINTS:
        res = x
        if (p + ".downsample.0.weight") in sd:
            res = _conv_bn(sd, p + ".downsample.0", p + ".downsample.1", x, 1, 0, False, training)
        return _conv_bn(sd, p + ".conv3", p + ".bn3", y, 1, 0, True, training, res=res)
    y = _conv_bn(sd, p + ".conv3", p + ".bn3", y, 1, 0, False, training)
    res = x
    if (p + ".downsample.0.weight") in sd:
        res = _conv_bn(sd, p + ".downsample.0", p + ".downsample.1", x, 1, 0, False, training)
    return F.relu(y + res)


def hr_module(sd: SD, p: str, xs: List[torch.Tensor], n_out: int, training=False):
    """model/HRNet.py:478-496 (branches of 4 BasicBlocks, then fuse rows, summed j = 0..n-1)."""
    n = len(xs)
    xs = list(xs)
    for i in range(n):
        b = 0
        while f"{p}.branches.{i}.{b}.conv1.weight" in sd:
            xs[i] = basic_block(sd, f"{p}.branches.{i}.{b}", xs[i], training)
            b += 1
    if n == 1:
        return xs
    outs = []
    if _BF16_POINTS:
        # the product's order and storage (otpose_amd/train.py::hr_module): the identity term seeds the sum, every other
        # term rides on a fused add whose result is stored as bf16, the last one applies the ReLU
        for i in range(n_out):
            y = xs[i]
            terms = [j for j in range(n) if j != i]
            for idx, j in enumerate(terms):
                last = idx == len(terms) - 1
                q = f"{p}.fuse_layers.{i}.{j}"
                if j > i:
                    low = _conv_bn(sd, q + ".0", q + ".1", xs[j], 1, 0, False, training)
                    y = y + F.interpolate(low, scale_factor=2 ** (j - i), mode="nearest")
                    y = _rb(F.relu(y) if last else y)
                else:
                    t = xs[j]
                    for k in range(i - j - 1):
                        t = _conv_bn(sd, f"{q}.{k}.0", f"{q}.{k}.1", t, 2, 1, True, training)
                    k = i - j - 1
                    y = _conv_bn(sd, f"{q}.{k}.0", f"{q}.{k}.1", t, 2, 1, last, training, res=y)
            outs.append(y)
        return outs
    for i in range(n_out):
        y = None
        for j in range(n):
            q = f"{p}.fuse_layers.{i}.{j}"
            if j == i:
                t = xs[j]
            elif j > i:    # 1x1 conv + BN + nearest upsample (HRNet.py:426-439)
                t = _conv_bn(sd, q + ".0", q + ".1", xs[j], 1, 0, False, training)
                t = F.interpolate(t, scale_factor=2 ** (j - i), mode="nearest")
            else:          # chain of stride-2 3x3 (HRNet.py:442-470)
                t = xs[j]
                for k in range(i - j):
                    t = _conv_bn(sd, f"{q}.{k}.0", f"{q}.{k}.1", t, 2, 1, relu=(k != i - j - 1), training=training)
            y = t if y is None else y + t
        outs.append(F.relu(y))
    return outs


def hrnet_forward(sd: SD, p: str, x, stage_cfgs: Sequence[dict], training=False):
    """model/HRNet.py:116-152.  ``stage_cfgs`` = [STAGE2, STAGE3, STAGE4] dicts."""
    x = _rb(x)                                          # bf16 mode: the frames enter as NHWC bf16
    x = _conv_bn(sd, p + ".conv1", p + ".bn1", x, 2, 1, True, training)
    x = _conv_bn(sd, p + ".conv2", p + ".bn2", x, 2, 1, True, training)
    for b in range(4):
        x = bottleneck(sd, f"{p}.layer1.{b}", x, training)
    ys = [x]
    for si, scfg in enumerate(stage_cfgs):
        s = si + 2
        nb = scfg["NUM_BRANCHES"]
        xs = []
        for i in range(nb):
            t = f"{p}.transition{s - 1}.{i}"
            if (t + ".0.weight") in sd:                 # same-resolution width change (HRNet.py:199-211)
                xs.append(_conv_bn(sd, t + ".0", t + ".1", ys[i], 1, 1, True, training))
            elif (t + ".0.0.weight") in sd:             # new branch from the LAST tensor (HRNet.py:137,145)
                z, k = ys[-1], 0
                while f"{t}.{k}.0.weight" in sd:
                    z = _conv_bn(sd, f"{t}.{k}.0", f"{t}.{k}.1", z, 2, 1, True, training)
                    k += 1
                xs.append(z)
            else:
                xs.append(ys[i])
        ys = xs
        nm = scfg["NUM_MODULES"]
        for m in range(nm):
            n_out = 1 if (s == 4 and m == nm - 1) else nb     # HRNet.py:105-106,172-175
            ys = hr_module(sd, f"{p}.stage{s}.{m}", ys, n_out, training)
    if _BF16_POINTS:                                    # bf16 operands, fp32 NCHW heat-maps out; their gradient enters as bf16
        return _rg(F.conv2d(_rb(ys[0]), _rbw(sd[p + ".final_layer.weight"]), sd.get(p + ".final_layer.bias")))
    return _conv(sd, p + ".final_layer", ys[0])


# ------------------------------------------------------------------------------------------------
# ConvTransformer
# ------------------------------------------------------------------------------------------------
def channel_layernorm(sd: SD, p: str, x, eps=1e-5):
    """model/blocks.py:95-110: normalise over C of (B, C, T), biased variance."""
    mu = x.mean(dim=1, keepdim=True)
    r = x - mu
    sigma = (r * r).mean(dim=1, keepdim=True)
    return r / torch.sqrt(sigma + eps) * sd[p + ".weight"] + sd[p + ".bias"]


def masked_mhca(sd: SD, p: str, x, n_head: int, stride: int):
    """model/blocks.py:400-453 (channel attention: contraction over T, hs x hs scores)."""
    B, C, T = x.shape
    hs = C // n_head

    def branch(name):
        y = F.conv1d(x, sd[f"{p}.{name}_conv.weight"], None, stride, 1, 1, C)
        y = channel_layernorm(sd, f"{p}.{name}_norm", y)
        return F.conv1d(y, sd[f"{p}.{name}.weight"], sd[f"{p}.{name}.bias"])

    q, k, v = branch("query"), branch("key"), branch("value")
    q = q.view(B, n_head, hs, -1)
    k = k.view(B, n_head, hs, -1)
    v = v.view(B, n_head, hs, -1)
    att = (q * (1.0 / math.sqrt(hs))) @ k.transpose(-2, -1)
    att = F.softmax(att, dim=-1)
    out = att @ v                                           # (B, nh, hs, T')
    out = out.transpose(2, 3).contiguous().view(B, C, -1)   # the layout scramble of blocks.py:447
    return F.conv1d(out, sd[p + ".proj.weight"], sd[p + ".proj.bias"])


def transformer_block(sd: SD, p: str, x, n_head: int, stride: int):
    """model/blocks.py:264-280, eval mode (dropout and drop-path are identities)."""
    a = masked_mhca(sd, p + ".attn", channel_layernorm(sd, p + ".ln1", x), n_head, stride)
    skip = x if stride == 1 else F.max_pool1d(x, stride + 1, stride, (stride + 1) // 2)
    y = skip + sd[p + ".drop_path_attn.scale"] * a
    if _BF16_POINTS and y.shape[1] % 8 == 0 and y.shape[2] % 32 == 0:
        # otpose_amd/train.py::TrainGraphBF16.mlp: the MLP interior of the temporal encoders on bf16 (the C = 17 flow encoder
        # and odd lengths stay fp32): LayerNorm output, hidden activation and GELU result stored as bf16, fp32 result
        yn = _rb(channel_layernorm(sd, p + ".ln2", y))
        h = _rb(F.conv1d(yn, _rbw(sd[p + ".mlp.0.weight"]), sd[p + ".mlp.0.bias"]))
        h = _rb(F.gelu(h))
        h = _rg(F.conv1d(_rb(h), _rbw(sd[p + ".mlp.3.weight"]), sd[p + ".mlp.3.bias"]))
        return y + sd[p + ".drop_path_mlp.scale"] * h
    h = F.conv1d(channel_layernorm(sd, p + ".ln2", y), sd[p + ".mlp.0.weight"], sd[p + ".mlp.0.bias"])
    h = F.conv1d(F.gelu(h), sd[p + ".mlp.3.weight"], sd[p + ".mlp.3.bias"])
    return y + sd[p + ".drop_path_mlp.scale"] * h


def conv_transformer(sd: SD, p: str, x4, n_head: int, arch: Tuple[int, int, int]):
    """model/ConvVideoTransformer.py:123-184 with arch[0] == 0; returns arch[2]+1 tensors (B, C, T)."""
    B, C, H, W = x4.shape
    T = H * W
    x = x4.reshape(B, C, T)
    x = x + sd[p + ".pos_embd"][:, :, :T]
    for i in range(arch[1]):
        x = transformer_block(sd, f"{p}.stem.{i}", x, n_head, 1)
    outs = [x]
    for i in range(arch[2]):
        x = transformer_block(sd, f"{p}.branch.{i}", x, n_head, 2)
        outs.append(F.interpolate(x, scale_factor=2 ** (i + 1), mode="linear", align_corners=False))
    return outs


# ------------------------------------------------------------------------------------------------
# RSB heads
# ------------------------------------------------------------------------------------------------
def _cbr(sd: SD, p: str, x, pad, relu=True, training=False):
    y = _bn(sd, p + ".bn", _conv(sd, p + ".conv", x, 1, pad), training)
    return F.relu(y) if relu else y


def rsb_block(sd: SD, p: str, x, training=False):
    """model/RSB.py:77-103"""
    bc = sd[p + ".conv_bn_relu2_1_1.conv.weight"].shape[0]
    s = torch.split(_cbr(sd, p + ".conv_bn_relu1", x, 0, True, training), bc, 1)
    c = lambda name, t: _cbr(sd, f"{p}.conv_bn_relu2_{name}", t, 1, True, training)  # noqa: E731
    o11 = c("1_1", s[0])
    o21 = c("2_1", s[1] + o11)
    o22 = c("2_2", o21)
    o31 = c("3_1", s[2] + o21)
    o32 = c("3_2", o31 + o22)
    o33 = c("3_3", o32)
    o41 = c("4_1", s[3] + o31)
    o42 = c("4_2", o41 + o32)
    o43 = c("4_3", o42 + o33)
    o44 = c("4_4", o43)
    y = _cbr(sd, p + ".conv_bn_relu3", torch.cat((o11, o22, o33, o44), 1), 0, False, training)
    if (p + ".downsample.conv.weight") in sd:
        x = _cbr(sd, p + ".downsample", x, 0, False, training)
    return F.relu(y + x)


def rsb_chain(sd: SD, p: str, x, training=False):
    """model/RSB.py:10-23"""
    i = 0
    while f"{p}.layers.{i}.conv_bn_relu1.conv.weight" in sd:
        x = rsb_block(sd, f"{p}.layers.{i}", x, training)
        i += 1
    return x


# ------------------------------------------------------------------------------------------------
# modulated deformable convolution (vectorised restatement)
# ------------------------------------------------------------------------------------------------
def _mdcn_sample(x, offset, mask, kh, kw, stride, pad, dil, dg):
    """Bilinear-sampled, mask-modulated columns ``col`` (N, C, K, Ho, Wo) and the pieces needed for
    the analytic backward.  Follows thirdparty/deform_conv/src/deform_conv_cuda_kernel.cu:506-571
    (indexing of offset/mask channels, the open-interval (-1,H)x(-1,W) test at :556) and :403-432
    (per-corner bounds of the bilinear kernel)."""
    N, C, H, W = x.shape
    K = kh * kw
    # stride / pad / dil: one int, or (h, w) pairs as the reference's entry points take them (cpp:474-480: stride_h,
    # stride_w, pad_h, pad_w, dilation_h, dilation_w; used as h_in = h_col * stride_h - pad_h + i * dilation_h, .cu:539-553)
    pair = lambda v: (v, v) if isinstance(v, int) else tuple(v)                   # noqa: E731
    (sh, sw), (ph, pw), (dh, dw) = pair(stride), pair(pad), pair(dil)
    Ho = (H + 2 * ph - (dh * (kh - 1) + 1)) // sh + 1
    Wo = (W + 2 * pw - (dw * (kw - 1) + 1)) // sw + 1
    cpg = C // dg
    dt = x.dtype
    off = offset.reshape(N, dg, K, 2, Ho, Wo)
    dev = x.device                       # device-agnostic: also run as the eager-GPU timing baseline (tools/)
    ki = torch.arange(K, device=dev) // kw
    kj = torch.arange(K, device=dev) % kw
    hb = (torch.arange(Ho, device=dev) * sh - ph).to(dt)[None, None, None, :, None]
    wb = (torch.arange(Wo, device=dev) * sw - pw).to(dt)[None, None, None, None, :]
    hs = hb + (ki * dh).to(dt)[None, None, :, None, None] + off[:, :, :, 0]     # (N, dg, K, Ho, Wo)
    ws = wb + (kj * dw).to(dt)[None, None, :, None, None] + off[:, :, :, 1]
    inside = (hs > -1) & (ws > -1) & (hs < H) & (ws < W)
    h0 = torch.floor(hs)
    w0 = torch.floor(ws)
    lh, lw = hs - h0, ws - w0
    hh, hw = 1 - lh, 1 - lw
    h0 = h0.long()
    w0 = w0.long()
    h1, w1 = h0 + 1, w0 + 1
    xg = x.reshape(N, dg, cpg, H * W)

    def corner(hi, wi, ok):
        ok = ok & inside
        idx = (hi.clamp(0, H - 1) * W + wi.clamp(0, W - 1)).reshape(N, dg, 1, -1).expand(N, dg, cpg, -1)
        v = torch.gather(xg, 3, idx).reshape(N, dg, cpg, K, Ho, Wo)
        return v * ok[:, :, None].to(dt), ok

    v1, ok1 = corner(h0, w0, (h0 >= 0) & (w0 >= 0))
    v2, ok2 = corner(h0, w1, (h0 >= 0) & (w1 <= W - 1))
    v3, ok3 = corner(h1, w0, (h1 <= H - 1) & (w0 >= 0))
    v4, ok4 = corner(h1, w1, (h1 <= H - 1) & (w1 <= W - 1))
    u = lambda t: t[:, :, None]                                                   # noqa: E731
    bil = u(hh * hw) * v1 + u(hh * lw) * v2 + u(lh * hw) * v3 + u(lh * lw) * v4   # (N, dg, cpg, K, Ho, Wo)
    m = mask.reshape(N, dg, 1, K, Ho, Wo)
    col = (bil * m).reshape(N, C, K, Ho, Wo)
    ctx = dict(v=(v1, v2, v3, v4), ok=(ok1, ok2, ok3, ok4), h0=h0, w0=w0, h1=h1, w1=w1, lh=lh, lw=lw,
               hh=hh, hw=hw, bil=bil, m=m, inside=inside, Ho=Ho, Wo=Wo)
    return col, ctx


def mdcn_forward(x, offset, mask, weight, bias, stride=1, pad=0, dil=1, groups=1, dg=1):
    """``out[n,o,p] = bias[o] + sum_{c,k} W[o,c,k] col[n,c,k,p]`` (deform_conv_cuda.cpp:519-548).
    Built from differentiable torch ops, so autograd through it is the fp64 gradient oracle."""
    N, C, H, W = x.shape
    Co, cpg_w, kh, kw = weight.shape
    col, ctx = _mdcn_sample(x, offset, mask, kh, kw, stride, pad, dil, dg)
    Ho, Wo = ctx["Ho"], ctx["Wo"]
    colg = col.reshape(N, groups, (C // groups) * kh * kw, Ho * Wo)
    wg = weight.reshape(groups, Co // groups, cpg_w * kh * kw)
    out = torch.einsum("gok,ngkp->ngop", wg, colg).reshape(N, Co, Ho, Wo)
    if bias is not None:
        out = out + bias.view(1, -1, 1, 1)
    return out


def dcn_v1_forward(x, offset, weight, stride=1, pad=0, dil=1, groups=1, dg=1):
    """DCN v1 forward (deform_conv_cuda.cpp:148-249 + deformable_im2col_gpu_kernel, .cu:128-183): the sampling window
    test (.cu:166) and the corner rules of deformable_im2col_bilinear (.cu:22-51) are those of the modulated kernel
    (.cu:549, 403-432), so v1 is the modulated op with mask == 1 and no bias.  Differentiable like mdcn_forward."""
    kh, kw = weight.shape[2:]
    ho = (x.shape[2] + 2 * pad - (dil * (kh - 1) + 1)) // stride + 1
    wo = (x.shape[3] + 2 * pad - (dil * (kw - 1) + 1)) // stride + 1
    mask = x.new_ones((x.shape[0], dg * kh * kw, ho, wo))
    return mdcn_forward(x, offset, mask, weight, None, stride, pad, dil, groups, dg)


def mdcn_backward(x, offset, mask, weight, grad_out, stride=1, pad=0, dil=1, groups=1, dg=1, with_bias=True):
    """Analytic gradients (deform_conv_cuda.cpp:596-660; kernels .cu:574-705, helpers :434-503).
    Returns (grad_x, grad_offset, grad_mask, grad_weight, grad_bias)."""
    N, C, H, W = x.shape
    Co, cpg_w, kh, kw = weight.shape
    K = kh * kw
    dt = x.dtype
    col, c = _mdcn_sample(x, offset, mask, kh, kw, stride, pad, dil, dg)
    Ho, Wo = c["Ho"], c["Wo"]
    cpg = C // dg
    wg = weight.reshape(groups, Co // groups, cpg_w * K)
    gog = grad_out.reshape(N, groups, Co // groups, Ho * Wo)
    gcol = torch.einsum("gok,ngop->ngkp", wg, gog).reshape(N, dg, cpg, K, Ho, Wo)          # cpp:602-605
    colg = col.reshape(N, groups, (C // groups) * K, Ho * Wo)
    grad_weight = torch.einsum("ngop,ngkp->gok", gog, colg).reshape(weight.shape)            # cpp:638-643
    grad_bias = grad_out.sum(dim=(0, 2, 3)) if with_bias else None                           # cpp:644-650
    inside = c["inside"][:, :, None].to(dt)
    # grad_mask = sum_c gcol * bilinear(x)  (.cu:685-692, 701-703); zero for outside samples
    grad_mask = (gcol * c["bil"] * inside).sum(2).reshape(N, dg * K, Ho, Wo)
    # grad_offset: derivative of the bilinear value w.r.t. (h, w) with the same corner bounds (.cu:461-503)
    v1, v2, v3, v4 = c["v"]
    u = lambda t: t[:, :, None]                                                               # noqa: E731
    d_h = -u(c["hw"]) * v1 - u(c["lw"]) * v2 + u(c["hw"]) * v3 + u(c["lw"]) * v4
    d_w = -u(c["hh"]) * v1 + u(c["hh"]) * v2 - u(c["lh"]) * v3 + u(c["lh"]) * v4
    gm = gcol * c["m"] * inside
    grad_offset = torch.stack(((gm * d_h).sum(2), (gm * d_w).sum(2)), dim=3).reshape(N, dg * K * 2, Ho, Wo)
    # grad_x: scatter gcol*mask*corner_weight to the (<=4) in-image neighbours (.cu:434-459, 612-629)
    grad_x = torch.zeros(N, dg, cpg, H * W, dtype=dt)
    wts = (c["hh"] * c["hw"], c["hh"] * c["lw"], c["lh"] * c["hw"], c["lh"] * c["lw"])
    his = (c["h0"], c["h0"], c["h1"], c["h1"])
    wis = (c["w0"], c["w1"], c["w0"], c["w1"])
    for wt, hi, wi, ok in zip(wts, his, wis, c["ok"]):
        idx = (hi.clamp(0, H - 1) * W + wi.clamp(0, W - 1)).reshape(N, dg, 1, -1).expand(N, dg, cpg, -1)
        contrib = (gm * u(wt * ok.to(dt))).reshape(N, dg, cpg, -1)
        grad_x.scatter_add_(3, idx, contrib)
    return grad_x.reshape(N, C, H, W), grad_offset, grad_mask, grad_weight, grad_bias


# ------------------------------------------------------------------------------------------------
# OTPose forward
# ------------------------------------------------------------------------------------------------
def window_maps(frames, mg, squeezed, inter, ctx):
    """The stacked per-joint feature maps of the two temporal encoders (model/OTPose.py:339-359) for a window of
    F = 2R + 1 frames ordered cur, prev_1, next_1, ..., prev_R, next_R with ``mg`` (B, 2R) = their frame distances.
    R = 2 is the reference verbatim (8 maps per joint).  R = 3 is this build's BASELINE configs[4] extension (the reference
    hard-codes 5 frames at :309, 320-321): every ring r adds ``sym_r = cur + (next_r + prev_r)`` (close, far, wide) and
    the outer side sums ``cur + (prev_2 + prev_3)`` / ``cur + (next_2 + next_3)`` join the stack: 12 maps per joint."""
    R = (len(frames) - 1) // 2
    cur = frames[0]
    div = lambda t, k: t / (mg[:, k] + 1)[:, None, None, None]               # noqa: E731  :339-342
    prev = [div(frames[1 + 2 * r], 2 * r) for r in range(R)]
    nxt = [div(frames[2 + 2 * r], 2 * r + 1) for r in range(R)]

    def side(ts):                                                            # cur + ((t1 + t2) + t3 ...)   :345-346
        acc = ts[0]
        for t in ts[1:]:
            acc = acc + t
        return cur + acc

    prev_b, next_b = side(prev), side(nxt)
    sym = [cur + (nxt[r] + prev[r]) for r in range(R)]                       # close_b, far_b (:347-349), wide_b
    b1 = [prev_b] + sym[::-1]                                                # prev_b, far_b, close_b          (:356)
    b2 = [next_b] + sym                                                      # next_b, close_b, far_b          (:358)
    if R == 3:
        b1.append(side(prev[1:]))
        b2.append(side(nxt[1:]))
    elif R != 2:
        raise ValueError("window of 5 or 7 frames expected")
    x1 = torch.stack([inter, ctx] + b1 + [t * squeezed for t in b1], 2).flatten(1, 2)     # :351-356
    x2 = torch.stack([inter, ctx] + b2 + [t * squeezed for t in b2], 2).flatten(1, 2)     # :358
    return x1, x2, prev_b


def otpose_forward(sd: SD, cfg, x, margin, training_bn=False, return_intermediates=False):
    """model/OTPose.py:307-394, eval semantics.  ``x`` (B, 15, H, W), ``margin`` (B, 4) (7-frame extension: (B, 21, H, W),
    (B, 6), see :func:`window_maps`).
    Returns the reference's 7-tuple (output, rough, intersection, prev_b, context, squeezed, total_b)."""
    m = cfg["MODEL"]
    J = m["NUM_JOINTS"]
    pe_w, pe_h = m["HEATMAP_SIZE"]
    stages = [m["EXTRA"][f"STAGE{s}"] for s in (2, 3, 4)]
    dils = list(m["DEFORMABLE_CONV"]["DILATION"])
    F_ = x.shape[1] // 3                                                     # frames of the window (reference: 5)
    x = torch.cat(x.split(3, dim=1), 0)                                      # :317
    B = x.shape[0] // F_
    rough = hrnet_forward(sd, "rough_pose_estimation_net", x, stages, training_bn)   # :319
    frames = rough.split(B, dim=0)                                           # :320  cur, prev1, next1, prev2, next2, ...
    total_b = frames[0]
    for f_ in frames[1:]:
        total_b = total_b + f_                                               # :324 (left-to-right association)
    squeezed = total_b.sum(1, keepdim=True).expand(-1, J, -1, -1).contiguous()   # :325-328
    inter = total_b * squeezed                                               # :330
    ctx = conv_transformer(sd, "flow_encoder", total_b, 1, (0, 6, 0))[0].reshape(B, J, pe_h, pe_w)  # :331-335
    x1, x2, prev_b = window_maps(frames, margin.to(x.dtype), squeezed, inter, ctx)        # :339-359
    t1 = conv_transformer(sd, "temporal_encoder1", x1, 2, (0, 6, 2))          # :360
    t2 = conv_transformer(sd, "temporal_encoder2", x2, 2, (0, 6, 2))
    s1 = torch.stack(t1, 1).contiguous().view(B, -1, pe_h, pe_w)             # :362-369
    s2 = torch.stack(t2, 1).contiguous().view(B, -1, pe_h, pe_w)
    f1 = _conv(sd, "final_layer1", s1)                                        # :372-373
    f2 = _conv(sd, "final_layer2", s2)
    branches = torch.cat([f1, f2], 1)                                         # :375
    def_h = rsb_chain(sd, "def_fuse", total_b, training_bn)                   # :376
    trans = rsb_chain(sd, "offset_mask_combine_conv", torch.cat([branches, def_h], 1), training_bn)  # :378
    out = None
    inter_d = []
    trans_b = _rb(trans)                                                      # bf16 mode: one bf16 copy feeds the ten convs
    for i, d in enumerate(dils):                                              # :381-392
        if _BF16_POINTS:
            off = _rg(F.conv2d(_rb(trans_b), _rbw(sd[f"offsets_list.{i}.0.weight"]), None, 1, d, d))
            msk = _rg(F.conv2d(_rb(trans_b), _rbw(sd[f"masks_list.{i}.0.weight"]), None, 1, d, d))
        else:
            off = F.conv2d(trans, sd[f"offsets_list.{i}.0.weight"], None, 1, d, d)
            msk = F.conv2d(trans, sd[f"masks_list.{i}.0.weight"], None, 1, d, d)
        p = f"modulated_deform_conv_list.{i}.deform_conv"
        wrp = mdcn_forward(def_h, off, msk, sd[p + ".weight"], sd[p + ".bias"], 1, d, d, 1, J)
        out = (1.0 / len(dils)) * wrp if out is None else out + (1.0 / len(dils)) * wrp
        inter_d.append((off, msk, wrp))
    res = (out, rough, inter, prev_b, ctx, squeezed, total_b)
    if return_intermediates:
        return res, dict(x1=x1, x2=x2, t1=t1, t2=t2, f1=f1, f2=f2, def_h=def_h, trans=trans, dcn=inter_d)
    return res


# ------------------------------------------------------------------------------------------------
# losses (model/loss.py)
# ------------------------------------------------------------------------------------------------
def _ohkm(l, topk=8):
    """model/loss.py:13-23: per-sample mean of the top-k joint losses, averaged over the batch."""
    return (torch.topk(l, topk, dim=1).values.sum(1) / topk).mean()


def st_ohkw_mse_loss(s, t, g, w, topk=8, global_flags=None):
    """ST_OHKW_MSELoss.forward (model/loss.py:25-92), use_target_weight=True.
    ``global_flags`` (J,) optionally overrides the per-joint ``max(gt)==1`` test (multi-GPU parity)."""
    B, J = t.shape[:2]
    s, t, g = s.reshape(B, J, -1), t.reshape(B, J, -1), g.reshape(B, J, -1)
    mse = s.new_zeros(())
    per = []
    for j in range(J):
        wj = w[:, j]                                  # (B, 1)
        a, gg, tt = s[:, j] * wj, g[:, j] * wj, t[:, j] * wj
        flag = (g[:, j].max() == 1) if global_flags is None else bool(global_flags[j])
        if flag:
            per.append(0.5 * (a - gg) ** 2)
            mse = mse + ((a - gg) ** 2).mean()
        else:
            per.append(0.5 * ((a - gg) ** 2 + (a - tt) ** 2))
            mse = mse + ((a - gg) ** 2).mean() + ((a - tt) ** 2).mean()
    l = torch.stack([p.mean(dim=1) for p in per], dim=1)
    ohkm = _ohkm(l, topk)
    return {"ohkm_loss_s": ohkm, "mse_loss_s": mse / J, "final_loss": ohkm + mse}


def joints_ohkm_mse_loss(o, g, w, topk=8):
    """JointsMSE_OHKMMSELoss.forward (model/loss.py:115-148)."""
    B, J = o.shape[:2]
    o, g = o.reshape(B, J, -1), g.reshape(B, J, -1)
    a, gg = o * w, g * w
    l = (0.5 * (a - gg) ** 2).mean(2)
    mse = ((a - gg) ** 2).mean(dim=(0, 2)).sum()
    ohkm = _ohkm(l, topk)
    return {"ohkm_loss": ohkm, "mse_loss": mse / J, "final_loss": ohkm + mse}


def joint_mse_loss(o, g, w):
    """JointMSELoss.forward (model/loss.py:158-182), use_target_weight=True."""
    B, J = o.shape[:2]
    o, g = o.reshape(B, J, -1), g.reshape(B, J, -1)
    return (((o * w - g * w) ** 2).mean(dim=(0, 2))).sum() / J


# ------------------------------------------------------------------------------------------------
# heat-map decode (the step after the path)
# ------------------------------------------------------------------------------------------------
def get_max_preds(hm):
    """utils/heatmap.py:143-171 (numpy semantics: first maximum, coordinates zeroed where max <= 0)."""
    import numpy as np
    hm = np.asarray(hm)
    n, j, h, w = hm.shape
    flat = hm.reshape(n, j, -1)
    idx = np.argmax(flat, 2)
    maxvals = np.amax(flat, 2).reshape(n, j, 1)
    preds = np.zeros((n, j, 2), np.float32)
    preds[:, :, 0] = idx % w
    preds[:, :, 1] = np.floor(idx / w)
    preds *= (maxvals > 0.0).astype(np.float32)
    return preds, maxvals


def get_final_preds(hm, center=None, scale=None):
    """utils/heatmap.py:108-132; transform_preds restated for rot = 0 as the similarity the three points of
    utils/transform.py:76-105 define (cv2 is not available in this image, so that last step is unpinned)."""
    import math
    import numpy as np
    hm = np.asarray(hm)
    coords, maxvals = get_max_preds(hm)
    n, j, h, w = hm.shape
    for a in range(n):
        for b in range(j):
            m = hm[a][b]
            px = int(math.floor(coords[a][b][0] + 0.5))
            py = int(math.floor(coords[a][b][1] + 0.5))
            if 1 < px < w - 1 and 1 < py < h - 1:
                diff = np.array([m[py][px + 1] - m[py][px - 1], m[py + 1][px] - m[py - 1][px]])
                coords[a][b] += np.sign(diff) * .25
    preds = coords.copy()
    if center is not None:
        center, scale = np.asarray(center, np.float32), np.asarray(scale, np.float32)
        for a in range(n):
            k = 200.0 * scale[a][0] / w
            preds[a, :, 0] = center[a][0] + (coords[a, :, 0] - 0.5 * w) * k
            preds[a, :, 1] = center[a][1] + (coords[a, :, 1] - 0.5 * h) * k
    return preds, maxvals


def accuracy(output, target, thr=0.5):
    """utils/evaluate.py:384-415 (hm_type='gaussian') with calc_dists (:352-365) and dist_acc (:368-381): PCK on the
    argmax coordinates of predicted and target heat-maps.  Returns (acc (J+1) float64, avg_acc, cnt, pred)."""
    import numpy as np
    pred, _ = get_max_preds(output)
    tgt, _ = get_max_preds(target)
    n, j, h, w = np.asarray(output).shape
    norm = np.ones((n, 2)) * np.array([h, w]) / 10                     # :396 - (x, y) divided by (h, w)/10, as written
    valid = (tgt[:, :, 0] > 1) & (tgt[:, :, 1] > 1)                     # :358
    d = np.linalg.norm(pred.astype(np.float32) / norm[:, None, :] - tgt.astype(np.float32) / norm[:, None, :], axis=2)
    dists = np.where(valid, d, -1.0).T                                  # (J, N), -1 = ignored
    acc = np.zeros(j + 1)
    avg, cnt = 0.0, 0
    for k in range(j):
        use = dists[k] != -1
        acc[k + 1] = (dists[k][use] < thr).sum() * 1.0 / use.sum() if use.sum() > 0 else -1      # :368-381
        if acc[k + 1] >= 0:
            avg += acc[k + 1]
            cnt += 1
    avg = avg / cnt if cnt != 0 else 0
    if cnt != 0:
        acc[0] = avg
    return acc, avg, cnt, pred


def frames_to_clip(frames_u8, mean=(0.485, 0.456, 0.406), std=(0.229, 0.224, 0.225)):
    """The reference's per-frame transform and clip assembly: torchvision ToTensor (uint8 HWC -> float CHW / 255) then
    Normalize ((t - mean) / std) (utils/transform.py:7-15, applied at dataset/PoseTrackDataset.py:397-406), frames
    concatenated on the channel axis (script/Common.py:117).  frames_u8 (B, F, H, W, 3) uint8 -> (B, 3F, H, W) float32."""
    t = frames_u8.permute(0, 1, 4, 2, 3).to(torch.float32).div(255)
    m = torch.tensor(mean, dtype=torch.float32).view(1, 1, 3, 1, 1)
    s = torch.tensor(std, dtype=torch.float32).view(1, 1, 3, 1, 1)
    t = (t - m) / s
    b, f, c, h, w = t.shape
    return t.reshape(b, f * c, h, w)
