"""bf16 training operators of the backbone over the C-ABI (``csrc/nhwc.hip``; BASELINE configs[2], reference step
script/Common.py:118-144 over model/HRNet.py:116-152).

Activations are NHWC bfloat16 tensors ``(N, H, W, CS)`` with ``CS`` = channels rounded up to 8 (padding channels are
zero); parameters, their gradients, BatchNorm statistics and running buffers are fp32 (the fp32 *master* weights are cast
to bf16 while they are packed for the matrix cores, every step).  Every ``torch.autograd.Function`` here launches HIP
kernels in both directions; PyTorch supplies memory, streams and the tape.  CPU tensors raise.
"""
from __future__ import annotations

import ctypes
import os

import torch
from torch.autograd import Function

from . import hip
from .ops import _require_gpu


def grad_slot(param):
    from .train_ops import grad_slot as _gs
    return _gs(param)

BF16 = torch.bfloat16


def cs(c: int) -> int:
    """Channel stride of an NHWC bf16 activation with ``c`` channels."""
    return (c + 7) // 8 * 8


def _desc(n, h, w, cin, cout, kh, kw, stride, pad, dil, out_mode=0):
    return hip.NhwcConvDesc(n, h, w, cin, cout, kh, kw, stride, pad, dil, out_mode)


def _new(shape, dtype, like):
    return torch.empty(shape, dtype=dtype, device=like.device)


def to_nhwc(x: torch.Tensor, frame_split: int = 0) -> torch.Tensor:
    """(N, C, H, W) fp32 -> (N, H, W, CS) bf16.  ``frame_split = B``: ``x`` is the (B, 5*C', H, W) clip tensor read as
    (5B, C', H, W) in the reference's frame order (model/OTPose.py:317) without materialising the re-layout."""
    _require_gpu(x)
    x = x.contiguous()
    if x.dtype != torch.float32:
        raise TypeError("to_nhwc expects float32")
    if frame_split:
        b, cf, h, w = x.shape                   # (B, 3 * frames, H, W): RGB frames stacked on the channel axis
        n, c = (cf // 3) * b, 3
    else:
        n, c, h, w = x.shape
    out = _new((n, h, w, cs(c)), BF16, x)
    hip.check(hip.lib().otp_nchw_f32_to_nhwc_bf16(hip.ptr(x), hip.ptr(out), n, c, h, w, int(frame_split), hip.stream_of(x)),
              "otp_nchw_f32_to_nhwc_bf16")
    return out


def to_nchw(x: torch.Tensor, c: int) -> torch.Tensor:
    """(N, H, W, CS) bf16 -> (N, c, H, W) fp32."""
    _require_gpu(x)
    n, h, w, _ = x.shape
    out = _new((n, c, h, w), torch.float32, x)
    hip.check(hip.lib().otp_nhwc_bf16_to_nchw_f32(hip.ptr(x.contiguous()), hip.ptr(out), n, c, h, w, hip.stream_of(x)),
              "otp_nhwc_bf16_to_nchw_f32")
    return out


def _pack(weight, d, dgrad, packs=None):
    if packs is not None:
        return packs.get(weight, d, dgrad)
    L = hip.lib()
    nbytes = L.otp_nhwc_conv_weight_bytes(ctypes.byref(d))
    if nbytes == 0:
        raise RuntimeError("otp_nhwc_conv: unsupported convolution shape")
    wp = torch.empty(nbytes // 2, dtype=BF16, device=weight.device)
    hip.check(L.otp_nhwc_conv_pack(hip.ptr(weight), hip.ptr(wp), ctypes.byref(d), int(dgrad), hip.stream_of(weight)),
              "otp_nhwc_conv_pack")
    return wp


class PackCache:
    """The packed bf16 operators (forward and input-gradient form) of one model's convolution weights, refreshed by ONE launch
    per training step instead of one per use (~670 launches of a few microseconds at cfg2, and as many allocations on the
    host, which bounds the forward once the branches overlap on the device).

    The first step packs every operator where it is first needed and records a job for it (weight pointer, destination,
    launch plan - ``otp_nhwc_conv_pack_job``); from then on :meth:`repack` at the start of a forward rewrites all
    destinations from the current fp32 master weights (``otp_nhwc_conv_pack_batch``) and the uses only look their tensor
    up.  The input-gradient operators are therefore those of the weights the forward saw.  Entries keep the weight storage
    alive, so a job never reads freed memory; a weight that moved (``model.to(...)``, FusedAdamW re-homing ``p.data`` into
    its flat buffer) gets new entries, and :meth:`repack` RETIRES every entry the previous forward did not look up (their
    old storage is released, they are no longer re-packed each step) and drops the whole table when the device changed."""

    LIMIT = 8192                              # entries; past it the table is rebuilt from scratch

    def __init__(self):
        self.entries = {}                     # (data_ptr, dgrad, desc bytes) -> (packed tensor, storage keep-alive, job record)
        self.jobs = bytearray()
        self.table = None
        self.dirty = False
        self.used = set()                     # keys looked up since the last repack()

    def __len__(self):
        return len(self.entries)

    def __deepcopy__(self, memo):             # jobs hold raw pointers: a copied / unpickled model starts with an empty table
        return PackCache()

    def __reduce__(self):
        return (PackCache, ())

    def clear(self):
        self.__init__()

    def get(self, weight, d, dgrad):
        key = (weight.data_ptr(), int(dgrad), bytes(d))
        e = self.entries.get(key)
        self.used.add(key)
        if e is not None:
            return e[0]
        if len(self.entries) >= self.LIMIT:
            self.clear()
        wp = _pack(weight, d, dgrad)
        L = hip.lib()
        job = ctypes.create_string_buffer(L.otp_nhwc_conv_pack_job_bytes())
        hip.check(L.otp_nhwc_conv_pack_job(hip.ptr(weight), hip.ptr(wp), ctypes.byref(d), int(dgrad), job),
                  "otp_nhwc_conv_pack_job")
        self.jobs += job.raw
        self.entries[key] = (wp, weight.detach(), job.raw)
        self.dirty = True
        return wp

    def repack(self, device):
        """All recorded operators from the current weights, on the current stream of ``device``."""
        if not self.entries:
            return
        if any(e[0].device != device for e in self.entries.values()):
            self.clear()                          # the model moved: every pointer in the table belongs to the old device
            return
        if self.used and len(self.used) < len(self.entries):
            # entries of weights that moved or of launches that no longer happen: retire them (keep-alives released)
            self.entries = {k: e for k, e in self.entries.items() if k in self.used}
            self.jobs = bytearray(b"".join(e[2] for e in self.entries.values()))
            self.dirty = True
        self.used = set()
        if not self.entries:
            return
        if self.dirty or self.table is None or self.table.device != device:
            self.table = torch.frombuffer(bytearray(self.jobs), dtype=torch.uint8).to(device)
            self.dirty = False
        hip.check(hip.lib().otp_nhwc_conv_pack_batch(hip.ptr(self.table), len(self.entries), hip.stream_of(self.table)),
                  "otp_nhwc_conv_pack_batch")


_ACTIVE_PACKS = None          # the PackCache of the training forward being built (train.TrainGraphBF16.forward), else None


def set_active_packs(cache):
    global _ACTIVE_PACKS
    prev, _ACTIVE_PACKS = _ACTIVE_PACKS, cache
    return prev


def conv_forward(x, weight, bias=None, stride=1, pad=0, dil=1, out_mode=0, want_stats=True, packs=None):
    """x (N, H, W, CinS) bf16, weight (Cout, Cin, kh, kw) fp32.  Returns (out, stats, rows): ``out`` NHWC bf16 (out_mode 0)
    or NCHW fp32 (out_mode 1); ``stats`` = per-tile [rows][2][CoutS] fp32 sums / sums of squares (out_mode 0 only)."""
    cout, cin, kh, kw = weight.shape
    n, h, w, cins = x.shape
    assert cins == cs(cin) and x.dtype == BF16 and x.is_contiguous()
    d = _desc(n, h, w, cin, cout, kh, kw, stride, pad, dil, out_mode)
    wp = _pack(weight.contiguous(), d, 0, packs)
    ho = (h + 2 * pad - dil * (kh - 1) - 1) // stride + 1
    wo = (w + 2 * pad - dil * (kw - 1) - 1) // stride + 1
    L = hip.lib()
    stats, rows = None, 0
    if out_mode == 0:
        out = _new((n, ho, wo, cs(cout)), BF16, x)
        if want_stats:
            rows = L.otp_nhwc_conv_stats_rows(ctypes.byref(d))
            stats = _new((rows, 2, cs(cout)), torch.float32, x)
    else:
        out = _new((n, cout, ho, wo), torch.float32, x)
    hip.check(L.otp_nhwc_conv_bf16(hip.ptr(x), hip.ptr(wp), hip.ptr(bias), hip.ptr(out), hip.ptr(stats), ctypes.byref(d),
                                   hip.stream_of(x)), "otp_nhwc_conv_bf16")
    return out, stats, rows


def conv_dgrad(gy, weight, in_hw, stride, pad, dil, packs=None, res=None):
    """dL/dx of ``conv_forward``: gy (N, Ho, Wo, CoutS) bf16 -> (N, H, W, CinS) bf16.  ``res`` (same shape as the result) is
    added in the kernel's epilogue: the gradient arriving over a skip connection (bit-identical to a separate bf16 add)."""
    cout, cin, kh, kw = weight.shape
    n = gy.shape[0]
    h, w = in_hw
    L = hip.lib()
    g = gy
    if stride > 1:
        hd, wd = h + 2 * pad - dil * (kh - 1), w + 2 * pad - dil * (kw - 1)
        gd = _new((n, hd, wd, gy.shape[3]), BF16, gy)
        hip.check(L.otp_nhwc_dilate(hip.ptr(gy), hip.ptr(gd), n, gy.shape[1], gy.shape[2], stride, hd, wd, gy.shape[3],
                                    hip.stream_of(gy)), "otp_nhwc_dilate")
        g = gd
    d = _desc(n, g.shape[1], g.shape[2], cout, cin, kh, kw, 1, dil * (kh - 1) - pad, dil, 0)
    wp = _pack(weight.contiguous(), d, 1, packs)
    gx = _new((n, h, w, cs(cin)), BF16, gy)
    if res is not None:
        assert res.shape == gx.shape and res.dtype == BF16 and res.is_contiguous()
    hip.check(L.otp_nhwc_conv_bf16_res(hip.ptr(g), hip.ptr(wp), None, hip.ptr(res), hip.ptr(gx), None, ctypes.byref(d),
                                       hip.stream_of(gy)), "otp_nhwc_conv_bf16_res(dgrad)")
    return gx


_WS = {}


def _workspace(device, nbytes):
    """Grow-only scratch per (device, stream): every user launches on the current stream, so reuse is stream-ordered; the
    HRNet branches of a training step run on several streams and each gets its own."""
    key = (device, torch.cuda.current_stream(device).cuda_stream)
    buf = _WS.get(key)
    if buf is None or buf.numel() * 4 < nbytes:
        buf = torch.empty((nbytes + 3) // 4, dtype=torch.float32, device=device)
        _WS[key] = buf
    return buf


_WG_STREAMS = {}          # (device index, launching stream) -> its weight-gradient side stream
_WG_PENDING = []          # side streams with work of the running backward pass on them


def _wgrad_side_stream(device):
    cur = torch.cuda.current_stream(device)
    key = (device.index, cur.cuda_stream)
    side = _WG_STREAMS.get(key)
    if side is None:
        side = _WG_STREAMS[key] = torch.cuda.Stream(device)
    return cur, side


def join_wgrad_streams():
    """The current stream waits for every weight-gradient launch of the backward pass that has just run (they go to side
    streams: :func:`conv_wgrad`).  Runs as an autograd engine callback at the end of the pass - on the stream ``backward()``
    was called on - and again at the head of ``FusedAdamW.step`` / ``grad_norm``."""
    while _WG_PENDING:
        side = _WG_PENDING.pop()
        torch.cuda.current_stream(side.device).wait_stream(side)


def conv_wgrad(x, gy, weight_shape, stride, pad, dil, out=None):
    """dL/dW of a convolution (fp32, summed over the batch).  A weight gradient is a LEAF of the backward pass - nothing of
    the pass waits for it - while the input gradient next to it is on the critical chain: when the result goes straight
    into the optimizer's flat gradient buffer (``out`` = :func:`otpose_amd.train_ops.grad_slot`, so autograd launches
    nothing on it) the two kernels of the weight gradient are enqueued on a side stream of the launching stream and overlap
    with the chain (``OTPOSE_WGRAD_STREAM=0``: same stream).  ``x`` / ``gy`` are registered with the caching allocator for
    that stream; :func:`join_wgrad_streams` orders the consumers."""
    cout, cin, kh, kw = weight_shape
    n, h, w, _ = x.shape
    d = _desc(n, h, w, cin, cout, kh, kw, stride, pad, dil, 0)
    L = hip.lib()
    nbytes = L.otp_nhwc_wgrad_workspace(ctypes.byref(d))
    if nbytes == 0:
        raise RuntimeError("otp_nhwc_wgrad: unsupported convolution shape")
    if out is not None and os.environ.get("OTPOSE_WGRAD_STREAM", "1") != "0":
        cur, side = _wgrad_side_stream(x.device)
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            ws = _workspace(x.device, nbytes)
            hip.check(L.otp_nhwc_wgrad_bf16(hip.ptr(x), hip.ptr(gy), hip.ptr(out), hip.ptr(ws), nbytes, ctypes.byref(d),
                                            hip.stream_of(x)), "otp_nhwc_wgrad_bf16")
        x.record_stream(side)
        gy.record_stream(side)
        if not _WG_PENDING:
            try:
                torch.autograd.Variable._execution_engine.queue_callback(join_wgrad_streams)
            except RuntimeError:              # called outside a backward pass: the optimizer (or the caller) joins
                pass
        if side not in _WG_PENDING:
            _WG_PENDING.append(side)
        return out
    ws = _workspace(x.device, nbytes)
    gw = out if out is not None else _new(weight_shape, torch.float32, x)
    hip.check(L.otp_nhwc_wgrad_bf16(hip.ptr(x), hip.ptr(gy), hip.ptr(gw), hip.ptr(ws), nbytes, ctypes.byref(d),
                                    hip.stream_of(x)), "otp_nhwc_wgrad_bf16")
    return gw


def bn_finalize(stats, rows, c, count, gamma, beta, running_mean, running_var, momentum, eps):
    """Per-tile sums -> (mean, rstd, scale, shift) as CS-padded fp32 vectors; updates the running statistics in place."""
    csz = stats.shape[2]
    vec = _new((4, csz), torch.float32, stats)
    hip.check(hip.lib().otp_nhwc_bn_finalize(hip.ptr(stats), rows, c, csz, float(count), hip.ptr(gamma), hip.ptr(beta),
                                             hip.ptr(vec[0]), hip.ptr(vec[1]), hip.ptr(vec[2]), hip.ptr(vec[3]),
                                             hip.ptr(running_mean), hip.ptr(running_var), eps, momentum,
                                             hip.stream_of(stats)), "otp_nhwc_bn_finalize")
    return vec


def bn_apply(x, scale, shift, res, relu, want_mask=False):
    """y = act(x*scale + shift (+ res)); with ``want_mask`` also the ReLU bit mask (1 bit per element) the backward reads
    instead of y (1/16 of its bytes)."""
    y = torch.empty_like(x)
    pixels = x.numel() // x.shape[-1]
    mask = torch.empty(x.numel() // 8, dtype=torch.uint8, device=x.device) if (want_mask and relu) else None
    hip.check(hip.lib().otp_nhwc_bn_apply(hip.ptr(x), hip.ptr(scale), hip.ptr(shift), hip.ptr(res), hip.ptr(y), hip.ptr(mask),
                                          pixels, x.shape[-1], int(relu), hip.stream_of(x)), "otp_nhwc_bn_apply")
    return (y, mask) if want_mask else y


def bn_backward(gy, y, x, mean, rstd, gamma, c, relu, want_res, out_gamma=None, out_beta=None):
    L = hip.lib()
    csz = x.shape[-1]
    pixels = x.numel() // csz
    nbytes = L.otp_nhwc_bn_backward_workspace(pixels, csz)
    ws = _new(((nbytes + 3) // 4,), torch.float32, x)
    gx = torch.empty_like(x)
    gres = torch.empty_like(x) if want_res else None
    dg = out_gamma if out_gamma is not None else _new((c,), torch.float32, x)
    db = out_beta if out_beta is not None else _new((c,), torch.float32, x)
    hip.check(L.otp_nhwc_bn_backward(hip.ptr(gy), hip.ptr(y), hip.ptr(x), hip.ptr(mean), hip.ptr(rstd), hip.ptr(gamma),
                                     hip.ptr(gx), hip.ptr(gres), hip.ptr(dg), hip.ptr(db), hip.ptr(ws), nbytes, pixels, c, csz,
                                     (2 if y.dtype == torch.uint8 else 1) if relu else 0, hip.stream_of(x)),
              "otp_nhwc_bn_backward")
    return gx, gres, dg, db


def conv_bn_forward(x, weight, gamma, beta, res, running_mean, running_var, stride, pad, relu, momentum, eps, packs=None):
    """conv + BatchNorm (batch statistics) (+ res) (+ ReLU) through ONE library call (``otp_nhwc_conv_bn_bf16``: the same three
    launches as conv_forward -> bn_finalize -> bn_apply).  Returns (conv output, mask, vec, y)."""
    cout, cin, kh, kw = weight.shape
    n, h, w, cins = x.shape
    assert cins == cs(cin) and x.dtype == BF16 and x.is_contiguous()
    d = _desc(n, h, w, cin, cout, kh, kw, stride, pad, 1, 0)
    wp = _pack(weight.contiguous(), d, 0, packs)
    ho, wo = (h + 2 * pad - kh) // stride + 1, (w + 2 * pad - kw) // stride + 1
    L = hip.lib()
    csz = cs(cout)
    c = _new((n, ho, wo, csz), BF16, x)
    rows = L.otp_nhwc_conv_stats_rows(ctypes.byref(d))
    stats = _new((rows, 2, csz), torch.float32, x)
    vec = _new((4, csz), torch.float32, x)
    y = torch.empty_like(c)
    mask = torch.empty(c.numel() // 8, dtype=torch.uint8, device=x.device) if relu else None
    hip.check(L.otp_nhwc_conv_bn_bf16(hip.ptr(x), hip.ptr(wp), hip.ptr(res), hip.ptr(c), hip.ptr(stats), hip.ptr(vec), hip.ptr(gamma),
                                      hip.ptr(beta), hip.ptr(running_mean), hip.ptr(running_var), eps, momentum, hip.ptr(y),
                                      hip.ptr(mask), int(relu), ctypes.byref(d), hip.stream_of(x)), "otp_nhwc_conv_bn_bf16")
    return c, mask, vec, y


class ConvBnFunction(Function):
    """``relu?(batch_norm(conv2d(x, weight, None, stride, pad)) (+ res))`` - one HRNet conv + BatchNorm2d (training mode:
    batch statistics, running buffers updated) + residual + ReLU (model/HRNet.py:500-571, 192-231, 416-473)."""

    @staticmethod
    def forward(ctx, x, weight, gamma, beta, res, running_mean, running_var, stride, pad, relu, momentum, eps):
        _require_gpu(x, weight, gamma, beta)
        x = x.contiguous()
        cout = weight.shape[0]
        ctx.packs = _ACTIVE_PACKS
        r = res.contiguous() if res is not None else None
        c, mask, vec, y = conv_bn_forward(x, weight, gamma, beta, r, running_mean, running_var, stride, pad, relu, momentum, eps, ctx.packs)
        ctx.save_for_backward(x, weight, gamma, c, mask, vec)
        ctx.cfg = (stride, pad, relu, res is not None)
        ctx.params = (weight, gamma, beta)             # gradient slots are looked up at backward time
        return y

    @staticmethod
    def backward(ctx, gy):
        x, weight, gamma, c, y, vec = ctx.saved_tensors
        stride, pad, relu, has_res = ctx.cfg
        gy = gy.contiguous()
        pw, pg, pb = ctx.params
        gc, gres, dg, db = bn_backward(gy, y, c, vec[0], vec[1], gamma, weight.shape[0], relu, has_res, grad_slot(pg),
                                       grad_slot(pb))
        gx = conv_dgrad(gc, weight, x.shape[1:3], stride, pad, 1, ctx.packs) if ctx.needs_input_grad[0] else None
        gw = conv_wgrad(x, gc, weight.shape, stride, pad, 1, grad_slot(pw)) if ctx.needs_input_grad[1] else None
        return gx, gw, dg, db, gres, None, None, None, None, None, None, None


def conv_bn(x, weight, gamma, beta, res=None, running_mean=None, running_var=None, stride=1, pad=0, relu=True,
            momentum=0.1, eps=1e-5):
    return ConvBnFunction.apply(x, weight, gamma, beta, res, running_mean, running_var, stride, pad, relu, momentum, eps)


class BasicBlockFunction(Function):
    """``relu(bn2(conv2(relu(bn1(conv1(x))))) + x)`` - an HRNet BasicBlock without down-sample path (model/HRNet.py:500-531;
    every block of stages 2-4) as ONE autograd node.  Same kernels and the same saved tensors as two :class:`ConvBnFunction`
    nodes; what the node adds is the backward's knowledge of the skip connection: dL/dx = dgrad(conv1) + dL/dres leaves the
    input-gradient conv already summed (``otp_nhwc_conv_bf16_res``) instead of autograd adding two tensors afterwards (108
    blocks at cfg2: a 160 MB pass each), and the host builds half as many nodes."""

    @staticmethod
    def forward(ctx, x, w1, g1, b1, rm1, rv1, w2, g2, b2, rm2, rv2, momentum, eps):
        _require_gpu(x, w1, w2)
        x = x.contiguous()
        ctx.packs = _ACTIVE_PACKS
        c1, m1, vec1, y1 = conv_bn_forward(x, w1, g1, b1, None, rm1, rv1, 1, 1, True, momentum, eps, ctx.packs)
        c2, m2, vec2, y2 = conv_bn_forward(y1, w2, g2, b2, x, rm2, rv2, 1, 1, True, momentum, eps, ctx.packs)
        ctx.save_for_backward(x, w1, g1, c1, m1, vec1, y1, w2, g2, c2, m2, vec2)
        ctx.params = (w1, g1, b1, w2, g2, b2)
        return y2

    @staticmethod
    def backward(ctx, gy):
        x, w1, g1, c1, m1, vec1, y1, w2, g2, c2, m2, vec2 = ctx.saved_tensors
        pw1, pg1, pb1, pw2, pg2, pb2 = ctx.params
        gy = gy.contiguous()
        gc2, gres, dg2, db2 = bn_backward(gy, m2, c2, vec2[0], vec2[1], g2, w2.shape[0], True, True, grad_slot(pg2),
                                          grad_slot(pb2))
        gy1 = conv_dgrad(gc2, w2, y1.shape[1:3], 1, 1, 1, ctx.packs)
        gw2 = conv_wgrad(y1, gc2, w2.shape, 1, 1, 1, grad_slot(pw2))
        del gc2
        gc1, _, dg1, db1 = bn_backward(gy1, m1, c1, vec1[0], vec1[1], g1, w1.shape[0], True, False, grad_slot(pg1),
                                       grad_slot(pb1))
        del gy1
        gx = conv_dgrad(gc1, w1, x.shape[1:3], 1, 1, 1, ctx.packs, res=gres)
        gw1 = conv_wgrad(x, gc1, w1.shape, 1, 1, 1, grad_slot(pw1))
        return gx, gw1, dg1, db1, None, None, gw2, dg2, db2, None, None, None, None


def basic_block(x, w1, g1, b1, rm1, rv1, w2, g2, b2, rm2, rv2, momentum=0.1, eps=1e-5):
    return BasicBlockFunction.apply(x, w1, g1, b1, rm1, rv1, w2, g2, b2, rm2, rv2, momentum, eps)


class ConvOutFunction(Function):
    """``F.conv2d(x, weight, bias, stride, pad, dil)`` from NHWC bf16 to the fp32 NCHW tensors of the module boundary (the
    HRNet ``final_layer``, model/HRNet.py:88-94,150)."""

    @staticmethod
    def forward(ctx, x, weight, bias, stride, pad, dil):
        _require_gpu(x, weight)
        x = x.contiguous()
        ctx.packs = _ACTIVE_PACKS
        out, _, _ = conv_forward(x, weight, bias, stride, pad, dil, out_mode=1, packs=ctx.packs)
        ctx.save_for_backward(x, weight)
        ctx.cfg = (stride, pad, dil, bias is not None)
        ctx.params = (weight, bias)
        return out

    @staticmethod
    def backward(ctx, gy):
        from .train_ops import channel_sum
        x, weight = ctx.saved_tensors
        stride, pad, dil, has_bias = ctx.cfg
        gy = gy.contiguous()
        g = to_nhwc(gy)
        gx = conv_dgrad(g, weight, x.shape[1:3], stride, pad, dil, ctx.packs) if ctx.needs_input_grad[0] else None
        gw = conv_wgrad(x, g, weight.shape, stride, pad, dil, grad_slot(ctx.params[0])) if ctx.needs_input_grad[1] else None
        gb = channel_sum(gy, grad_slot(ctx.params[1])) if has_bias and ctx.needs_input_grad[2] else None
        return gx, gw, gb, None, None, None


def conv_out(x, weight, bias=None, stride=1, pad=0, dil=1):
    return ConvOutFunction.apply(x, weight, bias, stride, pad, dil)


class UpsampleAddFunction(Function):
    """``relu?(res + nearest_upsample_f(low))`` on NHWC bf16 (HRNet fuse rows, model/HRNet.py:426-439,488-494)."""

    @staticmethod
    def forward(ctx, low, res, f, relu):
        _require_gpu(low, res)
        low, res = low.contiguous(), res.contiguous()
        n, h, w, c = res.shape
        out = torch.empty_like(res)
        hip.check(hip.lib().otp_nhwc_upsample_add(hip.ptr(low), hip.ptr(res), hip.ptr(out), n, h, w, c, f, int(relu),
                                                  hip.stream_of(low)), "otp_nhwc_upsample_add")
        ctx.save_for_backward(out if relu else None)
        ctx.cfg = (f, relu, low.shape)
        return out

    @staticmethod
    def backward(ctx, gy):
        (out,) = ctx.saved_tensors
        f, relu, (n, hl, wl, c) = ctx.cfg
        gy = gy.contiguous()
        gres = torch.empty_like(gy)
        glow = _new((n, hl, wl, c), BF16, gy)
        hip.check(hip.lib().otp_nhwc_upsample_add_backward(hip.ptr(gy), hip.ptr(out), hip.ptr(gres), hip.ptr(glow), n, hl, wl,
                                                           c, f, int(relu), hip.stream_of(gy)),
                  "otp_nhwc_upsample_add_backward")
        return glow, gres, None, None


def upsample_add(low, res, f, relu):
    return UpsampleAddFunction.apply(low, res, f, relu)


def channel_sum_nhwc(g, c, out=None):
    """Per-channel sums over the pixels of an NHWC bf16 tensor -> (c,) fp32 (bias gradients)."""
    L = hip.lib()
    csz = g.shape[-1]
    pixels = g.numel() // csz
    nbytes = L.otp_nhwc_channel_sum_workspace(pixels, csz)
    ws = _new(((nbytes + 3) // 4,), torch.float32, g)
    if out is None:
        out = _new((c,), torch.float32, g)
    hip.check(L.otp_nhwc_channel_sum(hip.ptr(g), hip.ptr(out), hip.ptr(ws), nbytes, pixels, c, csz, hip.stream_of(g)),
              "otp_nhwc_channel_sum")
    return out


class ConvBiasFunction(Function):
    """``F.conv2d(x, weight, bias, stride, pad, dil)`` on NHWC bf16 in and out (no normalisation behind it): the MLP
    up-projection of a TransformerBlock (model/blocks.py:248-254) on the (B, 1, T, C) view of a (B, C, T) sequence."""

    @staticmethod
    def forward(ctx, x, weight, bias, stride, pad, dil):
        _require_gpu(x, weight)
        x = x.contiguous()
        ctx.packs = _ACTIVE_PACKS
        out, _, _ = conv_forward(x, weight, bias, stride, pad, dil, out_mode=0, want_stats=False, packs=ctx.packs)
        ctx.save_for_backward(x, weight)
        ctx.cfg = (stride, pad, dil, bias is not None)
        ctx.params = (weight, bias)
        return out

    @staticmethod
    def backward(ctx, gy):
        x, weight = ctx.saved_tensors
        stride, pad, dil, has_bias = ctx.cfg
        gy = gy.contiguous()
        gx = conv_dgrad(gy, weight, x.shape[1:3], stride, pad, dil, ctx.packs) if ctx.needs_input_grad[0] else None
        gw = conv_wgrad(x, gy, weight.shape, stride, pad, dil, grad_slot(ctx.params[0])) if ctx.needs_input_grad[1] else None
        gb = (channel_sum_nhwc(gy, weight.shape[0], grad_slot(ctx.params[1]))
              if has_bias and ctx.needs_input_grad[2] else None)
        return gx, gw, gb, None, None, None


def conv_bias(x, weight, bias=None, stride=1, pad=0, dil=1):
    return ConvBiasFunction.apply(x, weight, bias, stride, pad, dil)


class GeluFunction(Function):
    """Exact-erf GELU on bf16 (fp32 arithmetic)."""

    @staticmethod
    def forward(ctx, x):
        _require_gpu(x)
        x = x.contiguous()
        y = torch.empty_like(x)
        hip.check(hip.lib().otp_gelu_bf16_forward(hip.ptr(x), hip.ptr(y), x.numel(), hip.stream_of(x)), "otp_gelu_bf16_forward")
        ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, gy):
        (x,) = ctx.saved_tensors
        gx = torch.empty_like(x)
        hip.check(hip.lib().otp_gelu_bf16_backward(hip.ptr(x), hip.ptr(gy.contiguous()), hip.ptr(gx), x.numel(),
                                                   hip.stream_of(x)), "otp_gelu_bf16_backward")
        return gx


gelu = GeluFunction.apply


def _draw_seed(device):
    """A 64-bit seed for a counter-based draw from PyTorch's CUDA generator state (seed, Philox offset), advancing the offset like a
    random op does: ``torch.manual_seed`` makes the step's draws repeat, consecutive calls differ.  Host side only."""
    gen = torch.cuda.default_generators[device.index if device.index is not None else torch.cuda.current_device()]
    off = gen.get_offset()
    gen.set_offset(off + 4)
    return (gen.initial_seed() * 0x9E3779B97F4A7C15 + (off + 1) * 0xD1B54A32D192ED03) & 0xFFFFFFFFFFFFFFFF


class GeluDropoutFunction(Function):
    """``dropout(gelu(x), p)`` of the TransformerBlock MLP (model/blocks.py:250-251, training mode) on bf16 as ONE pass each way
    (``otp_gelu_dropout_bf16_*``): the separate GELU + ``F.dropout`` launches cost a 240 MB pass forward and a 360 MB one backward more."""

    @staticmethod
    def forward(ctx, x, p, seed):
        _require_gpu(x)
        x = x.contiguous()
        y = torch.empty_like(x)
        keep = torch.empty(x.numel() // 8, dtype=torch.uint8, device=x.device)
        hip.check(hip.lib().otp_gelu_dropout_bf16_forward(hip.ptr(x), hip.ptr(y), hip.ptr(keep), x.numel(), float(p), int(seed),
                                                          hip.stream_of(x)), "otp_gelu_dropout_bf16_forward")
        ctx.save_for_backward(x, keep)
        ctx.p = float(p)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, keep = ctx.saved_tensors
        gx = torch.empty_like(x)
        hip.check(hip.lib().otp_gelu_dropout_bf16_backward(hip.ptr(x), hip.ptr(gy.contiguous()), hip.ptr(keep), hip.ptr(gx), x.numel(),
                                                           ctx.p, hip.stream_of(x)), "otp_gelu_dropout_bf16_backward")
        return gx, None, None


def gelu_dropout(x, p, seed=None):
    """``F.dropout(gelu(x), p, training=True)`` on a bf16 tensor (numel % 8 == 0); ``seed``: None draws one from the CUDA generator."""
    if not p > 0.0:
        return gelu(x)
    return GeluDropoutFunction.apply(x, p, _draw_seed(x.device) if seed is None else seed)


class MlpInteriorFunction(Function):
    """``conv1x1(W2, dropout(gelu(conv1x1(W1, x) + b1), p)) + b2`` - the interior of a TransformerBlock MLP (model/blocks.py:248-254) on
    the (B, 1, T, C) bf16 view of a sequence, fp32 (B, C, 1, T) result - as ONE autograd node whose element-wise passes ride in the
    projections' launches (``otp_nhwc_mlp_up_bf16``: the up-projection stores its result and dropout(gelu(result)) together;
    ``otp_nhwc_mlp_down_dgrad_bf16``: the down-projection's input gradient leaves multiplied by gelu' and the dropout factor).  Against
    the chain conv_bias -> gelu_dropout -> conv_out: one 240 MB pass less forward, one 360 MB pass less backward, same saved tensors."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, p, seed):
        _require_gpu(x, w1, w2)
        x = x.contiguous()
        n, h, w, _ = x.shape
        hid, cin = w1.shape[:2]
        cout = w2.shape[0]
        L = hip.lib()
        ctx.packs = _ACTIVE_PACKS
        d1 = _desc(n, h, w, cin, hid, 1, 1, 1, 0, 1, 0)
        wp1 = _pack(w1.contiguous(), d1, 0, ctx.packs)
        pre = _new((n, h, w, cs(hid)), BF16, x)
        act = torch.empty_like(pre)
        keep = torch.empty(pre.numel() // 8, dtype=torch.uint8, device=x.device)
        hip.check(L.otp_nhwc_mlp_up_bf16(hip.ptr(x), hip.ptr(wp1), hip.ptr(b1), hip.ptr(pre), hip.ptr(act), hip.ptr(keep), float(p),
                                         int(seed), ctypes.byref(d1), hip.stream_of(x)), "otp_nhwc_mlp_up_bf16")
        out, _, _ = conv_forward(act, w2, b2, 1, 0, 1, out_mode=1, packs=ctx.packs)
        ctx.save_for_backward(x, pre, act, keep, w1, w2)
        ctx.p = float(p)
        ctx.params = (w1, b1, w2, b2)
        return out

    @staticmethod
    def backward(ctx, go):
        from .train_ops import channel_sum
        x, pre, act, keep, w1, w2 = ctx.saved_tensors
        pw1, pb1, pw2, pb2 = ctx.params
        n, h, w, _ = x.shape
        hid, cin = w1.shape[:2]
        cout = w2.shape[0]
        L = hip.lib()
        go = go.contiguous()
        g = to_nhwc(go)
        # d(pre) = dropout'/gelu' (.) dgrad of the down-projection: one launch
        dd = _desc(n, h, w, cout, hid, 1, 1, 1, 0, 1, 0)
        wpd = _pack(w2.contiguous(), dd, 1, ctx.packs)
        dpre = torch.empty_like(pre)
        hip.check(L.otp_nhwc_mlp_down_dgrad_bf16(hip.ptr(g), hip.ptr(wpd), hip.ptr(pre), hip.ptr(keep), hip.ptr(dpre), ctx.p,
                                                 ctypes.byref(dd), hip.stream_of(g)), "otp_nhwc_mlp_down_dgrad_bf16")
        gw2 = conv_wgrad(act, g, w2.shape, 1, 0, 1, grad_slot(pw2)) if ctx.needs_input_grad[3] else None
        gb2 = channel_sum(go, grad_slot(pb2)) if pb2 is not None and ctx.needs_input_grad[4] else None
        gx = conv_dgrad(dpre, w1, (h, w), 1, 0, 1, ctx.packs) if ctx.needs_input_grad[0] else None
        gw1 = conv_wgrad(x, dpre, w1.shape, 1, 0, 1, grad_slot(pw1)) if ctx.needs_input_grad[1] else None
        gb1 = channel_sum_nhwc(dpre, hid, grad_slot(pb1)) if pb1 is not None and ctx.needs_input_grad[2] else None
        return gx, gw1, gb1, gw2, gb2, None, None


def mlp_interior_supported(x, w1, w2):
    """Both projections of this MLP run on the kernel that carries the fused epilogues (``otp_nhwc_mlp_fused_supported``)."""
    n, h, w, _ = x.shape
    hid, cin = w1.shape[:2]
    L = hip.lib()
    return bool(L.otp_nhwc_mlp_fused_supported(ctypes.byref(_desc(n, h, w, cin, hid, 1, 1, 1, 0, 1, 0)))
                and L.otp_nhwc_mlp_fused_supported(ctypes.byref(_desc(n, h, w, w2.shape[0], hid, 1, 1, 1, 0, 1, 0))))


def mlp_interior(x, w1, b1, w2, b2, p, seed=None):
    """The MLP interior on a (B, 1, T, CS) bf16 tensor -> (B, C, 1, T) fp32; ``p`` = dropout rate behind the GELU (0: none)."""
    return MlpInteriorFunction.apply(x, w1, b1, w2, b2, p, _draw_seed(x.device) if (seed is None and p > 0.0) else (seed or 0))


class ToNhwcFunction(Function):
    """(N, C, H, W) fp32 -> (N, H, W, CS) bf16 with the matching gradient conversion (the precision / layout hand-over
    in front of a bf16 sub-graph)."""

    @staticmethod
    def forward(ctx, x):
        ctx.c = x.shape[1]
        return to_nhwc(x)

    @staticmethod
    def backward(ctx, g):
        return to_nchw(g.contiguous(), ctx.c)


to_nhwc_grad = ToNhwcFunction.apply
