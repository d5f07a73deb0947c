"""Inference engine: the OTPose forward (reference model/OTPose.py:307-394) as a fixed list of HIP
launches over pre-allocated HBM buffers, replayed as one hipGraph.

Built once per (model, batch size): BatchNorm (eval statistics) and conv biases are folded into the
per-channel scale/shift epilogue of ``otp_conv2d``, weights are re-laid-out for the kernels
(``otp_conv2d_pack_weight``), every activation gets a static buffer (288 GB of HBM: nothing is
recycled inside a forward, so the whole launch sequence is capturable), channel concatenations and
splits become channel-offset views, residual adds / ReLU / GELU / nearest-upsample-accumulate /
residual-scale live in conv epilogues, and the 5-dilation weighted sum is folded into the DCN store.

Data layout: everything NCHW float32, frames of a clip stacked on the batch axis in the reference's
order [cur | prev | next | pprev | nnext] x B (OTPose.py:317) without materialising the re-layout
(``frame_split`` addressing in the stem conv).
"""
from __future__ import annotations

import ctypes
import os
import sys
import warnings
from typing import Callable, List

import torch

from . import hip, ops
from .ops import ACT_GELU, ACT_NONE, ACT_RELU, View

BN_EPS = 1e-5


def _pair(v):
    return (int(v), int(v)) if isinstance(v, int) else (int(v[0]), int(v[1]))


class InferenceEngine:
    def __init__(self, model, batch: int, device, use_graph: bool | None = None, stream_set: int = 0, inp=None, margin=None):
        device = torch.device(device)
        if device.type != "cuda" or not torch.cuda.is_available():
            raise RuntimeError("otpose_amd.OTPose runs on an MI355X through libotpose_hip.so; "
                               f"got device '{device}' (there is no CPU or PyTorch-op fallback)")
        self.lib = hip.lib()
        self.dev = device
        self.B = batch
        self.model = model
        cfg = model.cfg
        self.J = model.num_joints
        self.F = getattr(model, "window_frames", 5)             # frames per clip window
        self.W_img, self.H_img = cfg.MODEL.IMAGE_SIZE
        self.h, self.w = model.pe_h, model.pe_w
        self.ops: List[Callable] = []
        self._aux = {}                        # id(NCHW tensor) -> {"s8", "c4", "nchw_needed"}: S8 / C4 images written by its producer
        self._keep = []                       # parameter-derived tensors that must outlive the ops
        self._bufs = []                       # every activation buffer (see new())
        self._stream = None
        if use_graph is None:
            use_graph = os.environ.get("OTPOSE_HIP_GRAPH", "1") != "0"
        self.use_graph = use_graph
        self.use_winograd = os.environ.get("OTPOSE_WINOGRAD", "1") != "0"     # 3x3 stride-1 convs via csrc/wino.hip
        # conv products: "x3" = fp32 operands split into two bf16 pieces, three bf16 MFMAs per product, fp32 accumulate
        # (csrc/convx.hip); "f32" = the f32 MFMA kernels only (Winograd / direct)
        self.use_x3 = os.environ.get("OTPOSE_CONV_MATH", "x3") != "f32"
        # offset / mask convs + DCN gathers of all dilations in one launch (split-half products for the convs)
        self.use_dcn_fused = self.use_x3 and os.environ.get("OTPOSE_DCN_FUSED", "1") != "0"
        # HRNet branches (chains of BasicBlocks) on split-record activations fed by the LDS-DMA (csrc/convs.hip)
        self.use_s8 = self.use_x3 and os.environ.get("OTPOSE_S8", "1") != "0"
        self.use_flow_fused = os.environ.get("OTPOSE_FLOW_FUSED", "1") != "0"       # flow-encoder blocks via csrc/flowenc.hip
        self.use_small_conv = os.environ.get("OTPOSE_SMALL_CONV", "1") != "0"       # RSB staircase convs via csrc/conv_small.hip
        self.use_fused_mlp = os.environ.get("OTPOSE_FUSED_MLP", "1") != "0"   # transformer MLP via csrc/mlp.hip
        self.fuse_shortcut = os.environ.get("OTPOSE_FUSE_SHORTCUT", "1") != "0"  # layer1 shortcut folded into conv3
        self.fuse_upsample = os.environ.get("OTPOSE_FUSE_UPSAMPLE", "1") != "0"  # a fuse row's upsampled terms in one pass
        self.use_dense_cc = os.environ.get("OTPOSE_DENSE_CC", "1") != "0"     # q / k / v / proj via csrc/dense.hip
        self.use_qkv_front = os.environ.get("OTPOSE_QKV_FRONT", "1") != "0"   # + dwconv / LayerNorm fused in front of them
        self.use_pointx = self.use_x3 and os.environ.get("OTPOSE_POINTX", "1") != "0"   # layer1's 1x1 convs via csrc/pointx.hip
        # independent sub-graphs (the HRNet branches of a stage, the rows of its fuse layer, the two temporal encoders) are
        # emitted on side HIP streams: inside the captured graph they become parallel branches, so the small-map launches
        # (640-960 workgroups on 512 resident slots) fill each other's tails
        self.multi_stream = os.environ.get("OTPOSE_STREAMS", "1") != "0"
        # the last OTPOSE_F32_TAIL HighResolutionModules of stage 4 and the backbone's final 1x1 conv on the exact-fp32 MFMA kernels
        # (Winograd / direct) instead of split products: the layers whose rounding reaches `rough` un-attenuated
        self.f32_tail = int(os.environ.get("OTPOSE_F32_TAIL", "0")) if self.use_x3 else 0
        self._exact = False
        self._sid = 0
        # fp16 engine (engine_h16.py): the encoders' matrix kernels also take their operands as halves rounded once (csrc/mlpx.hip,
        # csrc/densex.hip: the *_h1 entry points); OTPOSE_H16_TAIL=0 keeps their split products
        self.half_products = bool(getattr(self, "h16", False)) and self.use_x3 and os.environ.get("OTPOSE_H16_TAIL", "1") != "0"
        # range guard of the half-piece arithmetic (csrc/range.hip, csrc/common.h:75): "defer" (default) raises at the NEXT
        # forward / at check_range() and NaN-fills this forward's heat-maps on the device; "sync" synchronises and raises in the
        # forward that overflowed; "off" only keeps the device-side NaN fill
        self.range_check = os.environ.get("OTPOSE_RANGE_CHECK", "defer")
        # process-wide pool (see hip.side_streams); stream_set > 0: the streams of another sub-batch of a PipelinedEngine
        self._side = hip.side_streams(device, 3, 4 * stream_set) if self.multi_stream else []
        self._given = (inp, margin)
        self.graph = None
        self.param_version = self._param_version()
        with torch.no_grad():
            self._build()

    @classmethod
    def bare(cls, device, multi_stream=False):
        """An engine with NO model behind it, for tests of one sub-module: the caller emits that module's launches with the
        engine's own emitters (``rsb_chain``, ``tblock``, ...) on buffers from :meth:`new`, sets ``inp`` to any of them (its
        stream is the launch stream) and runs the list with :meth:`_launch_all` - exactly the launches a full engine would
        issue for that module, without building an OTPose around it.  Never captures a graph."""
        self = cls.__new__(cls)
        device = torch.device(device)
        if device.type != "cuda" or not torch.cuda.is_available():
            raise RuntimeError("InferenceEngine.bare needs the GPU (there is no CPU or PyTorch-op fallback)")
        env = os.environ.get
        self.lib, self.dev, self.model, self.B = hip.lib(), device, None, 0
        self.ops, self._aux, self._keep, self._bufs, self._stream = [], {}, [], [], None
        self.use_graph, self.graph = False, None
        self.use_winograd = env("OTPOSE_WINOGRAD", "1") != "0"
        self.use_x3 = env("OTPOSE_CONV_MATH", "x3") != "f32"
        self.use_dcn_fused = self.use_x3 and env("OTPOSE_DCN_FUSED", "1") != "0"
        self.use_s8 = self.use_x3 and env("OTPOSE_S8", "1") != "0"
        self.use_flow_fused = env("OTPOSE_FLOW_FUSED", "1") != "0"
        self.use_small_conv = env("OTPOSE_SMALL_CONV", "1") != "0"
        self.use_fused_mlp = env("OTPOSE_FUSED_MLP", "1") != "0"
        self.fuse_shortcut = env("OTPOSE_FUSE_SHORTCUT", "1") != "0"
        self.fuse_upsample = env("OTPOSE_FUSE_UPSAMPLE", "1") != "0"
        self.use_dense_cc = env("OTPOSE_DENSE_CC", "1") != "0"
        self.use_qkv_front = env("OTPOSE_QKV_FRONT", "1") != "0"
        self.use_pointx = self.use_x3 and env("OTPOSE_POINTX", "1") != "0"
        self.multi_stream = bool(multi_stream)
        self.f32_tail, self._exact = 0, False
        self._sid = 0
        self.half_products = False
        self.range_check = env("OTPOSE_RANGE_CHECK", "defer")
        self._side = hip.side_streams(device, 3, 0) if self.multi_stream else []
        self.inp = None
        return self

    # ---------------------------------------------------------------------------------------------
    def matches(self, x) -> bool:
        return (x.shape[0] == self.B and x.device == self.dev and tuple(x.shape[2:]) == (self.H_img, self.W_img)
                and self._param_version() == self.param_version)

    def matches_shape(self, b, c, h, w, dev) -> bool:
        return (b == self.B and dev == self.dev and (h, w) == (self.H_img, self.W_img) and c == 3 * self.F
                and self._param_version() == self.param_version)

    def _param_version(self):
        """Staleness key of the packed weights: (storage address, autograd version) of every parameter AND buffer
        (BatchNorm running statistics are folded into the conv epilogues), so in-place updates (optimizer steps,
        ``load_state_dict``, ``running_mean.copy_``), re-pointed ``.data`` and module surgery are all seen.  Writes
        through ``p.data`` / raw pointers bump no version: call ``OTPose.invalidate_engine()`` after those."""
        ts = getattr(self, "_tracked", None)
        if ts is None:
            ts = self._tracked = list(self.model.parameters()) + list(self.model.buffers())
        v = a = 0
        for t in ts:
            v += t._version
            a ^= t.data_ptr()
        return (len(ts), v, a)

    def new(self, *shape):
        """A static activation / scratch buffer.  The engine owns it for its whole life: the launch list holds
        raw device pointers, and torch.cuda.graph() empties the allocator cache before capturing."""
        t = torch.empty(shape, dtype=torch.float32, device=self.dev)
        rng = os.environ.get("OTPOSE_POISON")                  # development aid ("lo:hi"): NaN-fill the buffers with these indices to
        if rng:                                                # find a kernel that reads memory nothing wrote (tools/poison_engine.py)
            lo, hi = (int(v) for v in rng.split(":"))
            if lo <= len(self._bufs) < hi:
                t.fill_(float("nan"))
        self._bufs.append(t)
        return t

    def dev_param(self, t):
        t = t.detach().to(self.dev, torch.float32).contiguous()
        self._keep.append(t)
        return t

    @staticmethod
    def winograd_pays(cin, cout):
        """Shapes on which the Winograd kernel measured faster than the direct one (tools/conv_bench.py --wino): whole
        8-channel chunks and output-channel counts that fill its 48-row tiles (64 -> 64 would compute 96 rows)."""
        cout16 = (cout + 15) // 16 * 16
        if cin == 64 and cout16 == 64:
            return True                                   # layer1 Bottleneck conv2: 0.463 -> 0.408 ms despite the ragged 2nd tile
        return cin >= 32 and cin % 8 == 0 and cout16 >= 48 and ((cout16 + 47) // 48) * 48 <= 1.15 * cout16

    # ---- op emitters ----------------------------------------------------------------------------
    def _emit(self, fn):
        self.ops.append((fn, self._sid))

    def fork(self, sids):
        """Side streams ``sids`` (1-based) start after everything emitted so far on the main stream."""
        if not self.multi_stream:
            return
        sids = [s for s in sids if 0 < s <= len(self._side)]

        def run():
            main = torch.cuda.current_stream(self.dev)
            for s_ in sids:
                self._side[s_ - 1].wait_stream(main)
        self.ops.append((run, -1))

    def join(self, sids):
        """The main stream waits for everything emitted on side streams ``sids``."""
        if not self.multi_stream:
            return
        sids = [s for s in sids if 0 < s <= len(self._side)]

        def run():
            main = torch.cuda.current_stream(self.dev)
            for s_ in sids:
                main.wait_stream(self._side[s_ - 1])
        self.ops.append((run, -1))

    def on_stream(self, sid):
        """Ops emitted from now on go to side stream ``sid`` (0 = main; ids past the pool fall back to main)."""
        self._sid = sid if (self.multi_stream and 0 <= sid <= len(self._side)) else 0

    def conv(self, inp: View, weight, out: View, stride=1, pad=0, dil=1, bn=None, bias=None, act=ACT_NONE,
             res: View = None, in2: View = None, res_up=1, frame_split=0, cin=None, scale=None, shift=None):
        """Emit act(scale*conv(inp (+in2)) + shift (+res)) with BN / bias folded into scale / shift."""
        for v_ in (inp, in2, res):
            self._needs_nchw(v_)
        w = self.dev_param(weight)
        if w.dim() == 3:
            w = w.unsqueeze(-1)
        cout, cin_w, kh, kw = w.shape
        if bn is not None:
            g, b = self.dev_param(bn.weight), self.dev_param(bn.bias)
            mu, var = self.dev_param(bn.running_mean), self.dev_param(bn.running_var)
            sc = g / torch.sqrt(var + bn.eps)
            sh = b - mu * sc
            if bias is not None:
                sh = sh + self.dev_param(bias) * sc
        else:
            sc = self.dev_param(scale) if scale is not None else None
            sh = self.dev_param(bias) if bias is not None else None
            if shift is not None:
                sh = self.dev_param(shift)
        if sc is not None:
            sc = sc.contiguous()
            self._keep.append(sc)
        if sh is not None:
            sh = sh.contiguous()
            self._keep.append(sh)
        d = ops.conv_desc(inp, out, cout, kh, kw, stride, pad, dil, act, in2, res, res_up, frame_split, cin)
        self._keep.append(d)
        L = self.lib
        if os.environ.get("OTPOSE_CONV_LOG"):                       # development aid: one line per emitted convolution
            route = ("x3" if self.use_x3 and in2 is None and (kh, kw) in ((3, 3), (1, 1)) and ops.x3_supported(d) else
                     "wino" if self.use_winograd and in2 is None and self.winograd_pays(cin_w, cout) and ops.wino_supported(d)
                     else "direct")
            print(f"conv {d.N}x{d.Cin}->{cout} k{kh} s{stride} p{pad} d{dil} {d.H}x{d.W} in2={in2 is not None} "
                  f"res={res is not None} up={res_up} fs={frame_split} {route}", file=sys.stderr)
        if self.use_small_conv and in2 is not None and res is None and (kh, kw) == (3, 3) and ops.small_conv_supported(d):
            # RSB staircase convs (a few channels, pre-added second input): one thread per pixel, exact fp32 (csrc/conv_small.hip)
            wc = ops.pack_small_conv_weight(w)
            self._keep.append(wc)
            sargs = (hip.ptr(inp.t), hip.ptr(in2.t), hip.ptr(wc), hip.ptr(sc), hip.ptr(sh), hip.ptr(out.t), d)

            def run_small():
                hip.check(L.otp_conv3x3_small(*sargs, self._stream), "otp_conv3x3_small")
            self._emit(run_small)
            return out
        if (self.use_pointx and not self._exact and in2 is None and (kh, kw) == (1, 1) and stride == 1 and pad == 0 and res_up <= 1 and not frame_split
                and act in (ACT_NONE, ACT_RELU) and inp.C == cin_w
                and (cin_w in (64, 128, 256) or (os.environ.get("OTPOSE_POINTX_FUSE", "1") != "0" and cin_w % 16 == 0)
                     or os.environ.get("OTPOSE_POINTX_ANY", "0") == "1")       # (opt-in: the RSB heads' and the final 1x1 convs too -
                                                                              #  26.69 against 26.72 ms, and they leave exact fp32)
                and ops.pointwise_x3_supported(cin_w, cout, inp.t.shape[2] * inp.t.shape[3])):
            # HRNet layer1's and the fuse layers' 1x1 convs (<= 256 channels in and out): register-resident pixels, streamed weights
            # (csrc/pointx.hip) - bound by their HBM streams, which the implicit-GEMM kernel ran at a third of the rate
            pk = ops.pack_pointwise_x3(w, sc, sh)
            self._keep.append(pk)
            pargs = (hip.ptr(inp.t), hip.ptr(pk), hip.ptr(res.t if res is not None else None), hip.ptr(out.t), inp.t.shape[0],
                     cin_w, cout, inp.t.shape[2] * inp.t.shape[3], inp.ctot, inp.coff, res.ctot if res is not None else 0,
                     res.coff if res is not None else 0, out.ctot, out.coff, int(act == ACT_RELU))

            def run_px():
                hip.check(L.otp_pointwise_x3(*pargs, self._stream), "otp_pointwise_x3")
            self._emit(run_px)
            return out
        if self.use_x3 and not self._exact and in2 is None and (kh, kw) in ((3, 3), (1, 1)) and ops.x3_supported(d):
            # 3x3 / stride 1 with Cin % 16 == 0: split-half (f16x3) products on the 16-bit matrix cores, fp32 storage and
            # accumulation (csrc/convx.hip); the per-channel scale is folded into the packed weights
            e = ops.x3_weight_exponent(w, sc)         # weights stored times 2^e, the sum multiplied by 2^-e (otp_conv_desc.out_scale)
            d.out_scale = 2.0 ** -e
            xp = ops.pack_x3_weight(w, sc, stride, e)
            self._keep.append(xp)
            xargs = (hip.ptr(inp.t), hip.ptr(xp), hip.ptr(sh), hip.ptr(res.t if res is not None else None), hip.ptr(out.t), d)

            def run_x3():
                hip.check(L.otp_conv2d_x3(*xargs, self._stream), "otp_conv2d_x3")
            self._emit(run_x3)
            return out
        if self.use_winograd and in2 is None and self.winograd_pays(cin_w, cout) and ops.wino_supported(d):
            # 3x3 / stride 1 / pad 1 with enough channels: Winograd F(2x2,3x3) kernel (csrc/wino.hip), same epilogue
            up = ops.pack_wino_weight(w)
            self._keep.append(up)
            wargs = (hip.ptr(inp.t), hip.ptr(up), hip.ptr(sc), hip.ptr(sh), hip.ptr(res.t if res is not None else None),
                     hip.ptr(out.t), d)

            def run_wino():
                hip.check(L.otp_conv2d_wino(*wargs, self._stream), "otp_conv2d_wino")
            self._emit(run_wino)
            return out
        wp = ops.pack_conv_weight(w)
        self._keep.append(wp)
        args = (hip.ptr(inp.t), hip.ptr(in2.t if in2 is not None else None), hip.ptr(wp), hip.ptr(sc), hip.ptr(sh),
                hip.ptr(res.t if res is not None else None), hip.ptr(out.t), d)

        def run():
            hip.check(L.otp_conv2d(*args, self._stream), "otp_conv2d")
        self._emit(run)
        return out

    def conv_bn(self, inp: View, conv_mod, bn_mod, act=ACT_NONE, res=None, out=None, res_up=1, **kw):
        n, _, h, w = inp.t.shape
        if kw.get("frame_split"):
            n = n * (inp.ctot // kw["cin"])
        k, s, p, dl = conv_mod.kernel_size[0], conv_mod.stride[0], conv_mod.padding[0], conv_mod.dilation[0]
        ho = (h + 2 * p - (dl * (k - 1) + 1)) // s + 1
        wo = (w + 2 * p - (dl * (k - 1) + 1)) // s + 1
        if out is None:
            f = max(res_up, 1)
            out = View(self.new(n, conv_mod.out_channels, ho * f, wo * f))
        return self.conv(inp, conv_mod.weight, out, s, p, dl, bn=bn_mod, bias=conv_mod.bias, act=act, res=res,
                         res_up=res_up, **kw)

    def dense(self, xs, packs, ress, outs, B, C, T):
        """Emit one otp_dense_cc (or, with split-half products, otp_dense_x3) launch over len(xs) problems."""
        ax, ap, ar, ao = ops.dense_cc_args(xs, packs, ress, outs)
        self._keep += [ax, ap, ar, ao, *packs]
        x3 = self.use_x3 and ops.dense_x3_supported(C, T)
        fn = (self.lib.otp_dense_h1 if self.half_products else self.lib.otp_dense_x3) if x3 else self.lib.otp_dense_cc
        self.call(fn, "otp_dense_cc", ax, ap, ar, ao, len(xs), B, C, T)

    def call(self, fn, name, *args):
        def run():
            hip.check(fn(*args, self._stream), name)
        self._emit(run)

    # ---- HRNet (reference model/HRNet.py:116-152) -------------------------------------------------
    def basic_block(self, blk, x: View) -> View:
        y = self.conv_bn(x, blk.conv1, blk.bn1, ACT_RELU)
        res = x
        if blk.downsample is not None:
            res = self.conv_bn(x, blk.downsample[0], blk.downsample[1])
        return self.conv_bn(y, blk.conv2, blk.bn2, ACT_RELU, res=res)

    def _needs_nchw(self, v):
        """A consumer reads the NCHW tensor of ``v``: its producer must write it (see fuse_s8)."""
        a = self._aux.get(id(v.t)) if v is not None else None
        if a is not None:
            a["nchw_needed"] = True

    def fuse_s8(self, lows, factors, res: View, tgt: View):
        """Last op of a fuse row whose tail is upsampled terms (HRNet.py:487-494): relu(res + up(low0) + ...) written as the
        S8 and C4 images the next module's branch reads (csrc/convs.hip, otp_s8_upsample_add); the NCHW tensor ``tgt`` is
        written only if some other consumer asks for it before the first launch.  Returns False when the row is not eligible."""
        n_, c_, hh, wh = tgt.t.shape
        if not self.use_s8 or self._exact or tgt.coff != 0 or tgt.C != c_ or c_ % 16 or wh % 4 or (hh * wh) % 4:
            return False
        if not ops.s8_conv_supported(ops.s8_conv_desc(n_, c_, c_, hh, wh, ACT_RELU)):
            return False
        s8_res = os.environ.get("OTPOSE_S8_RESIDUAL", "1") != "0"
        lazy = s8_res and os.environ.get("OTPOSE_S8_LAZY_NCHW", "1") != "0"
        # the identity term: the S8 image of `res` when its producer wrote one (a branch output: hi + lo of the records, 2^-22) -
        # `res` then needs no NCHW tensor on this account - else the fp32 NCHW tensor
        raux = self._aux.get(id(res.t)) if (lazy and res.coff == 0 and res.C == c_ and res.ctot == c_) else None
        res_img = raux.get("s8") if raux is not None else None
        if res_img is None:
            self._needs_nchw(res)
        # (the C4 fp32 image only when the next module's branch reads its residual from it: OTPOSE_S8_RESIDUAL=0)
        s8 = self.new(n_ * c_ * hh * wh)
        c4 = None if s8_res else self.new(n_ * c_ * hh * wh)
        aux = {"s8": s8, "c4": c4, "nchw_needed": False}
        self._aux[id(tgt.t)] = aux
        lp = (ctypes.c_void_p * len(lows))(*[hip.ptr(v.t) for v in lows])
        fp = (ctypes.c_int * len(lows))(*factors)
        self._keep += [lp, fp]
        L = self.lib
        args = (lp, fp, len(lows), hip.ptr(res_img if res_img is not None else res.t), int(res_img is not None))
        tail = (hip.ptr(s8), hip.ptr(c4), n_, c_, hh, wh, 1, res.ctot, res.coff, tgt.ctot, tgt.coff)

        def run():
            hip.check(L.otp_s8_upsample_add_ex(*args, hip.ptr(tgt.t) if aux["nchw_needed"] else None, *tail, self._stream),
                      "otp_s8_upsample_add")
        self._emit(run)
        return True

    def s8_image(self, v: View, want_c4=False):
        """The S8 image (and, on request, the C4 image) of a full NCHW tensor: what its producer already wrote (``_aux``), or one
        ``otp_s8_pack`` pass emitted here, on the current stream, and cached for later consumers.  None when the tensor is not
        eligible (channel slice, channels % 16, pixels % 4)."""
        n, c, h, w = v.t.shape
        if not self.use_s8 or v.coff != 0 or v.C != c or c % 16 or (h * w) % 4:
            return None
        aux = self._aux.get(id(v.t))
        if aux is not None and "s8" in aux and (not want_c4 or aux.get("c4") is not None):
            return aux
        self._needs_nchw(v)
        s8 = aux["s8"] if aux is not None and "s8" in aux else self.new(n * c * h * w)
        c4 = self.new(n * c * h * w) if want_c4 else None
        self.call(self.lib.otp_s8_pack, "otp_s8_pack", hip.ptr(v.t), hip.ptr(s8), hip.ptr(c4), n, c, h, w, v.ctot, v.coff)
        if aux is None:
            aux = self._aux[id(v.t)] = {"nchw_needed": True}
        aux["s8"] = s8
        if c4 is not None:
            aux["c4"] = c4
        return aux

    def s2_chain_s8(self, layers, x: View, act_last, res: View = None, out: View = None):
        """A chain of 3x3 / stride 2 conv + BN (+ ReLU) layers (a fuse layer's down-sampling path, model/HRNet.py:442-470, or a
        transition layer's new branch, :213-229) on S8 records (csrc/convs2.hip): every intermediate exists only as its S8 image,
        the last layer writes act(conv + shift (+ res)) into the fp32 NCHW tensor ``out``.  ``layers`` = [(conv, bn, relu)].
        Returns the output view, or None (nothing emitted) when a layer is not of that shape."""
        if not (self.use_s8 and not self._exact and os.environ.get("OTPOSE_S8_STRIDE2", "1") != "0"):
            return None
        n, c, h, w = x.t.shape
        if x.coff != 0 or x.C != c:
            return None
        hh, ww, cin = h, w, c
        descs = []
        for li, (cv, bn, relu) in enumerate(layers):
            last = li == len(layers) - 1
            if (cv.kernel_size != (3, 3) or cv.stride != (2, 2) or cv.padding != (1, 1) or cv.dilation != (1, 1) or cv.bias is not None
                    or cv.groups != 1 or cv.in_channels != cin or hh % 2 or ww % 2):
                return None
            co = cv.out_channels
            if last:
                if out is None:
                    out = View(self.new(n, co, hh // 2, ww // 2))
                d = ops.s8_s2_conv_desc(n, cin, co, hh, ww, act_last, out, res)
            else:
                d = ops.s8_s2_conv_desc(n, cin, co, hh, ww, ACT_RELU if relu else ACT_NONE)
            if not ops.s8_s2_conv_supported(d, last):
                return None
            descs.append(d)
            hh, ww, cin = hh // 2, ww // 2, co
        aux = self.s8_image(x)
        if aux is None:
            return None
        self._needs_nchw(res)
        L = self.lib
        cur = aux["s8"]
        hh, ww, cin = h, w, c
        for li, ((cv, bn, relu), d) in enumerate(zip(layers, descs)):
            last = li == len(layers) - 1
            sc, sh = self._bn_fold(bn)
            wp = self._pack_s8(cv, sc, d)
            self._keep += [wp, d]
            if last:
                self.call(L.otp_conv3x3_s2_s8, "otp_conv3x3_s2_s8", hip.ptr(cur), hip.ptr(wp), hip.ptr(sh),
                          hip.ptr(res.t) if res is not None else None, hip.ptr(out.t), None, d)
            else:
                nxt = self.new(n * cv.out_channels * (hh // 2) * (ww // 2))
                self.call(L.otp_conv3x3_s2_s8, "otp_conv3x3_s2_s8", hip.ptr(cur), hip.ptr(wp), hip.ptr(sh), None, None, hip.ptr(nxt), d)
                cur = nxt
            hh, ww, cin = hh // 2, ww // 2, cv.out_channels
        return out

    def _pack_s8(self, conv, sc, desc):
        """Packed split-product weights of a 3x3 conv for csrc/convs.hip / convs2.hip, stored times the layer's power of two
        (ops.x3_weight_exponent) with the inverse in ``desc.out_scale``."""
        w = self.dev_param(conv.weight)
        e = ops.x3_weight_exponent(w, sc)
        desc.out_scale = 2.0 ** -e
        return ops.pack_s8_weight(w, sc, e)

    def _bn_fold(self, bn):
        g, b = self.dev_param(bn.weight), self.dev_param(bn.bias)
        mu, var = self.dev_param(bn.running_mean), self.dev_param(bn.running_var)
        sc = (g / torch.sqrt(var + bn.eps)).contiguous()
        sh = (b - mu * sc).contiguous()
        self._keep += [sc, sh]
        return sc, sh

    def branch_s8(self, blocks, x: View, want_s8=False):
        """A branch of a HighResolutionModule (HRNet.py:478-496: 4 BasicBlocks, :500-530) on split-record activations
        (csrc/convs.hip).  The branch input is converted once to its S8 image (the MFMA operand records): conv input AND
        residual (round 4; before: a C4 fp32 image next to it); inside the branch conv1 goes S8 -> S8, conv2 S8 + S8 residual ->
        S8; the last conv2 writes the NCHW tensor the fuse layer reads.  Returns None when the branch is not of that
        shape (the caller then emits the blocks one convolution at a time)."""
        n, c, h, w = x.t.shape
        if x.coff != 0 or x.C != c or self._exact:
            return None
        for blk in blocks:
            convs = (getattr(blk, "conv1", None), getattr(blk, "conv2", None))
            if (getattr(blk, "downsample", None) is not None or hasattr(blk, "conv3") or any(
                    cv is None or cv.kernel_size != (3, 3) or cv.stride != (1, 1) or cv.padding != (1, 1) or cv.dilation != (1, 1)
                    or cv.bias is not None or cv.in_channels != c or cv.out_channels != c or cv.groups != 1 for cv in convs)):
                return None
        if not ops.s8_conv_supported(ops.s8_conv_desc(n, c, c, h, w, ACT_RELU)):
            return None
        L = self.lib
        new_img = lambda: self.new(n * c * h * w)                          # noqa: E731  (S8 and C4 images are 4 bytes per element)
        # residual of a block: its input's S8 records (hi + lo holds the value to 2^-22: otp_conv_desc.res_layout = 1) - no fp32 (C4)
        # image exists between the blocks of a branch; OTPOSE_S8_RESIDUAL=0: the exact fp32 residual chain of round 3
        res_s8 = os.environ.get("OTPOSE_S8_RESIDUAL", "1") != "0"
        aux = self.s8_image(x, want_c4=not res_s8)                         # written by the producer (a fuse row), or packed here
        if aux is None:
            return None
        xs8, xc4 = aux["s8"], (None if res_s8 else aux["c4"])
        out = None
        for b, blk in enumerate(blocks):
            last = b == len(blocks) - 1
            sc1, sh1 = self._bn_fold(blk.bn1)
            sc2, sh2 = self._bn_fold(blk.bn2)
            y8 = new_img()
            d1 = ops.s8_conv_desc(n, c, c, h, w, ACT_RELU)
            w1 = self._pack_s8(blk.conv1, sc1, d1)
            self._keep += [w1, d1]
            self.call(L.otp_conv3x3_s8, "otp_conv3x3_s8", hip.ptr(xs8), hip.ptr(w1), hip.ptr(sh1), None, None, ops.S8_F32_C4,
                      hip.ptr(y8), d1)
            if last:
                out = View(self.new(n, c, h, w))
                d2 = ops.s8_conv_desc(n, c, c, h, w, ACT_RELU, out)
                d2.res_layout = int(res_s8)
                w2 = self._pack_s8(blk.conv2, sc2, d2)
                self._keep += [w2, d2]
                # a stride-2 consumer in the fuse layer (csrc/convs2.hip) reads the S8 image: written here, next to the NCHW tensor
                o8 = new_img() if want_s8 else None
                lazy = res_s8 and os.environ.get("OTPOSE_S8_LAZY_NCHW", "1") != "0"
                oaux = {"s8": o8, "nchw_needed": not lazy} if o8 is not None else {"nchw_needed": True}
                if o8 is not None:
                    # consumers that can read the S8 image (stride-2 chains, the fuse row's identity term) do; the NCHW tensor is
                    # written only if some consumer asks for it before the first launch (_needs_nchw)
                    self._aux[id(out.t)] = oaux
                largs = (hip.ptr(y8), hip.ptr(w2), hip.ptr(sh2), hip.ptr(xs8 if res_s8 else xc4))
                optr, o8ptr = hip.ptr(out.t), (hip.ptr(o8) if o8 is not None else None)

                def run_last(largs=largs, optr=optr, o8ptr=o8ptr, d2=d2, oaux=oaux):
                    hip.check(L.otp_conv3x3_s8(*largs, optr if oaux["nchw_needed"] else None, ops.S8_F32_NCHW, o8ptr, d2, self._stream),
                              "otp_conv3x3_s8")
                self._emit(run_last)
            else:
                oc4, o8 = (None if res_s8 else new_img()), new_img()
                d2 = ops.s8_conv_desc(n, c, c, h, w, ACT_RELU)
                d2.res_layout = int(res_s8)
                w2 = self._pack_s8(blk.conv2, sc2, d2)
                self._keep += [w2, d2]
                self.call(L.otp_conv3x3_s8, "otp_conv3x3_s8", hip.ptr(y8), hip.ptr(w2), hip.ptr(sh2),
                          hip.ptr(xs8 if res_s8 else xc4), hip.ptr(oc4) if oc4 is not None else None, ops.S8_F32_C4, hip.ptr(o8), d2)
                xs8, xc4 = o8, oc4
        return out

    def conv_bn_s8(self, x: View, conv, bn, act=ACT_RELU):
        """act(bn(conv3x3 stride 1 pad 1 (x))) on csrc/convs.hip from the S8 image of ``x`` (packed here unless its producer wrote
        one), result as S8 records for the branch that follows and as the NCHW tensor only if another consumer asks for it.
        Returns None (nothing emitted) when the layer is not of that shape."""
        n, c, h, w = x.t.shape
        if (not self.use_s8 or self._exact or conv.kernel_size != (3, 3) or conv.stride != (1, 1) or conv.padding != (1, 1)
                or conv.dilation != (1, 1) or conv.bias is not None or conv.groups != 1 or conv.in_channels != c
                or x.coff != 0 or x.C != c or conv.out_channels % 16):
            return None
        co = conv.out_channels
        out = View(self.new(n, co, h, w))
        d = ops.s8_conv_desc(n, c, co, h, w, act, out)
        if not ops.s8_conv_supported(d):
            return None
        aux = self.s8_image(x)
        if aux is None:
            return None
        sc, sh = self._bn_fold(bn)
        wp = self._pack_s8(conv, sc, d)
        o8 = self.new(n * co * h * w)
        lazy = os.environ.get("OTPOSE_S8_RESIDUAL", "1") != "0" and os.environ.get("OTPOSE_S8_LAZY_NCHW", "1") != "0"
        oaux = {"s8": o8, "nchw_needed": not lazy}
        self._aux[id(out.t)] = oaux
        self._keep += [wp, d]
        L = self.lib
        largs = (hip.ptr(aux["s8"]), hip.ptr(wp), hip.ptr(sh), None)
        optr, o8ptr = hip.ptr(out.t), hip.ptr(o8)

        def run():
            hip.check(L.otp_conv3x3_s8(*largs, optr if oaux["nchw_needed"] else None, ops.S8_F32_NCHW, o8ptr, d, self._stream),
                      "otp_conv3x3_s8")
        self._emit(run)
        return out

    def conv1_conv2_s8(self, x: View, conv1, bn1, conv2, bn2, out: View = None):
        """relu(bn2(conv2(relu(bn1(conv1(x)))))) of a Bottleneck (HRNet.py:551-571: 1x1 then 3x3) with the intermediate kept as S8
        records: conv1 on csrc/pointx.hip writes the operand records of csrc/convs.hip, conv2 reads them and writes fp32 NCHW -
        no fp32 round trip of the 64-channel tensor and the S8 conv kernel instead of the implicit-GEMM one (179 -> 115 us at
        cfg2).  Returns None when the pair is not of that shape."""
        if not (self.use_s8 and self.use_pointx and os.environ.get("OTPOSE_L1_S8", "1") != "0"):
            return None
        n, _, h, w = x.t.shape
        c1i, c1o, c2o = conv1.in_channels, conv1.out_channels, conv2.out_channels
        if (conv1.kernel_size != (1, 1) or conv1.stride != (1, 1) or conv1.padding != (0, 0) or conv1.bias is not None
                or conv1.groups != 1 or conv2.kernel_size != (3, 3) or conv2.stride != (1, 1) or conv2.padding != (1, 1)
                or conv2.dilation != (1, 1) or conv2.bias is not None or conv2.groups != 1 or conv2.in_channels != c1o
                or x.C != c1i or not ops.pointwise_x3_s8_supported(c1i, c1o, h * w)):
            return None
        if out is None:
            out = View(self.new(n, c2o, h, w))
        d2 = ops.s8_conv_desc(n, c1o, c2o, h, w, ACT_RELU, out)
        if not ops.s8_conv_supported(d2):
            return None
        L = self.lib
        self._needs_nchw(x)
        sc1, sh1 = self._bn_fold(bn1)
        sc2, sh2 = self._bn_fold(bn2)
        pk = ops.pack_pointwise_x3_s8(self.dev_param(conv1.weight), sc1, sh1)
        w2 = self._pack_s8(conv2, sc2, d2)
        y8 = self.new(n * c1o * h * w)
        self._keep += [pk, w2, d2]
        self.call(L.otp_pointwise_x3_s8, "otp_pointwise_x3_s8", hip.ptr(x.t), hip.ptr(pk), hip.ptr(y8), n, c1i, c1o, h * w, x.ctot,
                  x.coff, 1)
        self.call(L.otp_conv3x3_s8, "otp_conv3x3_s8", hip.ptr(y8), hip.ptr(w2), hip.ptr(sh2), None, hip.ptr(out.t), ops.S8_F32_NCHW,
                  None, d2)
        return out

    def bottleneck(self, blk, x: View, s8_only=False) -> View:
        """One Bottleneck (model/HRNet.py:551-571).  ``s8_only``: the result is wanted as S8 records (transition1's convs read
        them) - conv3 writes them straight from its accumulators (csrc/pointx.hip) and the NCHW tensor of the returned view is
        filled, by a conversion pass, only if some consumer asks for it."""
        res = x
        if blk.downsample is not None:
            # the 1x1 shortcut only needs x: a side stream next to conv1 / conv2
            self.fork((1,))
            self.on_stream(1)
            res = self.conv_bn(x, blk.downsample[0], blk.downsample[1])
            self.on_stream(0)
        y = self.conv1_conv2_s8(x, blk.conv1, blk.bn1, blk.conv2, blk.bn2)
        if y is None:
            y = self.conv_bn(x, blk.conv1, blk.bn1, ACT_RELU)
            y = self.conv_bn(y, blk.conv2, blk.bn2, ACT_RELU)
        if blk.downsample is not None:
            self.join((1,))
        n, _, h, w = y.t.shape
        c3 = blk.conv3
        if (s8_only and self.use_s8 and self.use_pointx and not self._exact and c3.kernel_size == (1, 1) and c3.bias is None
                and c3.stride == (1, 1) and y.C == c3.in_channels and res.C == c3.out_channels
                and ops.pointwise_x3_s8_supported(c3.in_channels, c3.out_channels, h * w)):
            for v_ in (y, res):
                self._needs_nchw(v_)
            co = c3.out_channels
            sc, sh = self._bn_fold(blk.bn3)
            pk = ops.pack_pointwise_x3_s8(self.dev_param(c3.weight), sc, sh)
            out = View(self.new(n, co, h, w))
            o8 = self.new(n * co * h * w)
            oaux = {"s8": o8, "nchw_needed": False}
            self._aux[id(out.t)] = oaux
            self._keep.append(pk)
            L = self.lib
            args = (hip.ptr(y.t), hip.ptr(pk), hip.ptr(res.t), hip.ptr(o8), n, y.C, co, h * w, y.ctot, y.coff, res.ctot, res.coff, 1)

            def run():
                hip.check(L.otp_pointwise_x3_s8_res(*args, self._stream), "otp_pointwise_x3_s8_res")
                if oaux["nchw_needed"]:                         # (a consumer without an S8 path: hi + lo back to fp32 NCHW)
                    hip.check(L.otp_s8_unpack(hip.ptr(o8), hip.ptr(out.t), n, co, h, w, self._stream), "otp_s8_unpack")
            self._emit(run)
            return out
        return self.conv_bn(y, blk.conv3, blk.bn3, ACT_RELU, res=res)

    def hr_module(self, mod, xs: List[View], fork_in=True, join_out=True) -> List[View]:
        """One HighResolutionModule (model/HRNet.py:478-496).  ``fork_in`` / ``join_out`` = False chain consecutive modules of
        a stage stream by stream: branch i of the next module only reads row i of this module's fuse layer, which ran on stream
        i, so neither the join behind the rows nor the fork in front of the next branches is needed - a branch starts as soon
        as ITS row is done instead of when the slowest row is."""
        n = mod.num_branches
        xs = list(xs)
        if fork_in:
            self.fork(range(1, n))
        for i in range(n):
            self.on_stream(i)                                     # branch i is independent of the others until the fuse
            # branch i feeds the stride-2 chains of the fuse rows below it
            down = i < len(mod.fuse_layers) - 1 and not self._exact and os.environ.get("OTPOSE_S8_STRIDE2", "1") != "0"
            y = self.branch_s8(list(mod.branches[i]), xs[i], want_s8=down) if self.use_s8 else None
            if y is not None:
                xs[i] = y
                continue
            for blk in mod.branches[i]:
                xs[i] = self.basic_block(blk, xs[i])
        self.on_stream(0)
        self.join(range(1, n))
        if n == 1:
            return xs
        outs = []
        if self.use_s8 and not self._exact and os.environ.get("OTPOSE_S8_STRIDE2", "1") != "0":
            for j in range(min(n, len(mod.fuse_layers) - 1)):     # S8 images the rows below read, made before the rows fork
                self.s8_image(xs[j])
        self.fork(range(1, len(mod.fuse_layers)))
        for i in range(len(mod.fuse_layers)):
            self.on_stream(i)                                     # fuse row i reads every branch, writes only y_i
            # y_i = sum_j f_ij(x_j), then ReLU (HRNet.py:487-494).  The identity term rides on the first
            # emitted conv as its residual; later terms accumulate in place; the last one applies the ReLU.
            terms = [j for j in range(n) if j != i]
            y = None
            # two or three upsampled terms (the j > i tail of the row): their convs run at low resolution and ONE streaming pass
            # adds them all to the high-resolution tensor (same summation order as chaining them)
            ups = [j for j in terms if j > i]
            hi_w = xs[i].t.shape[3]
            if self.fuse_upsample and len(ups) >= 2 and hi_w % 4 == 0:
                terms = [j for j in terms if j < i]
            else:
                ups = []
            for idx, j in enumerate(terms):
                last = idx == len(terms) - 1 and not ups
                act = ACT_RELU if last else ACT_NONE
                res = xs[i] if y is None else y
                fl = mod.fuse_layers[i][j]
                if j > i:
                    f = 2 ** (j - i)
                    tgt = y if y is not None else View(self.new(*xs[i].t.shape))
                    if f >= 2 and ((xs[j].t.shape[3] * f) % 4 == 0 or os.environ.get("OTPOSE_UP_ANY_WIDTH", "1") != "0"):
                        # (any width: otp_upsample_add has a one-element form - rows of 18 at 384x288 used to send the 384 -> 192
                        #  term of stage 4's row 2 to the generic conv kernel's upsample epilogue, 195 us on the row's critical path)
                        # the upsampled tensor is f*f x the conv result: conv at low resolution, then one streaming
                        # accumulate kernel (measured faster than the conv kernel's element-wise upsample epilogue)
                        low = self.conv_bn(xs[j], fl[0], fl[1], ACT_NONE)
                        n_, c_, hl, wl = low.t.shape
                        if act == ACT_RELU and low.coff == 0 and low.C == c_ and self.fuse_s8([low], [f], res, tgt):
                            y = tgt
                            continue
                        self._needs_nchw(res)
                        self.call(self.lib.otp_upsample_add, "otp_upsample_add", hip.ptr(low.t), hip.ptr(res.t),
                                  hip.ptr(tgt.t), n_, c_, hl, wl, f, int(act == ACT_RELU), low.ctot, low.coff,
                                  res.ctot, res.coff, tgt.ctot, tgt.coff)
                        y = tgt
                    else:
                        y = self.conv_bn(xs[j], fl[0], fl[1], act, res=res, out=tgt, res_up=f)
                else:
                    tgt = y if y is not None else View(self.new(*xs[i].t.shape))
                    chain = [(fl[k][0], fl[k][1], k < len(fl) - 1) for k in range(len(fl))]
                    if self.s2_chain_s8(chain, xs[j], act, res=res, out=tgt) is not None:
                        y = tgt
                        continue
                    t = xs[j]
                    for k in range(len(fl) - 1):
                        t = self.conv_bn(t, fl[k][0], fl[k][1], ACT_RELU)
                    y = self.conv_bn(t, fl[-1][0], fl[-1][1], act, res=res, out=tgt)
            if ups:
                res = xs[i] if y is None else y
                tgt = y if y is not None else View(self.new(*xs[i].t.shape))
                lows = [self.conv_bn(xs[j], mod.fuse_layers[i][j][0], mod.fuse_layers[i][j][1], ACT_NONE) for j in ups]
                n_, c_, hh, wh = tgt.t.shape[0], xs[i].C, xs[i].t.shape[2], xs[i].t.shape[3]
                if self.fuse_s8(lows, [2 ** (j - i) for j in ups], res, tgt):
                    outs.append(tgt)
                    continue
                self._needs_nchw(res)
                lp = (ctypes.c_void_p * len(ups))(*[hip.ptr(v.t) for v in lows])
                fp = (ctypes.c_int * len(ups))(*[2 ** (j - i) for j in ups])
                self._keep += [lp, fp]
                self.call(self.lib.otp_upsample_add_multi, "otp_upsample_add_multi", lp, fp, len(ups), hip.ptr(res.t),
                          hip.ptr(tgt.t), n_, c_, hh, wh, 1, res.ctot, res.coff, tgt.ctot, tgt.coff)
                y = tgt
            outs.append(y)
        self.on_stream(0)
        if join_out:
            self.join(range(1, len(mod.fuse_layers)))
        return outs

    def hrnet(self, net, x_in: View) -> View:
        c1 = net.conv1
        n_in, c_in, h_in, w_in = x_in.t.shape
        if (self.use_x3 and os.environ.get("OTPOSE_STEM_X3", "1") != "0" and x_in.coff == 0 and x_in.C == c_in and c_in == 3 * self.F
                and c1.kernel_size == (3, 3) and c1.stride == (2, 2) and c1.padding == (1, 1) and c1.dilation == (1, 1)
                and c1.in_channels == 3 and c1.bias is None and c1.groups == 1
                and ops.stem_conv_x3_supported(n_in, self.F, h_in, w_in, c1.out_channels)):
            # conv1 + bn1 + relu on the frames of the clip (HRNet.py:118-120 after OTPose.py:317): one k-step of split products per
            # 16 pixels x 16 channels, gathered straight from the image (csrc/stem.hip), stored through an LDS slab as 256-byte runs
            # per channel: 202 against 426 us for the direct kernel (tools/stem_time.py), forward -0.3 ... -0.5 ms
            sc, sh = self._bn_fold(net.bn1)
            pk = ops.pack_stem_conv_x3(self.dev_param(c1.weight), sc, sh)
            self._keep.append(pk)
            x = View(self.new(self.F * n_in, c1.out_channels, (h_in - 1) // 2 + 1, (w_in - 1) // 2 + 1))
            self.call(self.lib.otp_stem_conv_x3, "otp_stem_conv_x3", hip.ptr(x_in.t), hip.ptr(pk), hip.ptr(x.t), n_in, self.F, h_in,
                      w_in, c1.out_channels)
        else:
            x = self.conv_bn(x_in, net.conv1, net.bn1, ACT_RELU, frame_split=self.B, cin=3)
        blocks = list(net.layer1)
        b0 = blocks[0]
        if (self.fuse_shortcut and b0.downsample is not None and b0.conv3.kernel_size == (1, 1)
                and b0.downsample[0].kernel_size == (1, 1) and b0.downsample[0].stride == (1, 1)):
            # first Bottleneck (HRNet.py:551-571 with the 1x1 shortcut of :240-247): relu(bn3(conv3(y)) + bn_d(conv_d(x))) is
            # one 1x1 conv over the channel concatenation [x, y] with both BatchNorm scales folded into the weights - the
            # 256-channel shortcut tensor (566 MB at cfg2) is neither written nor read back.  x (stem conv2) and y
            # (Bottleneck conv2) are written straight into the two channel slices of one buffer.
            n, _, h, w = x.t.shape
            c2 = net.conv2
            ho = (h + 2 * c2.padding[0] - c2.kernel_size[0]) // c2.stride[0] + 1
            wo = (w + 2 * c2.padding[0] - c2.kernel_size[0]) // c2.stride[0] + 1
            cx, cy = c2.out_channels, b0.conv2.out_channels
            cat = self.new(n, cx + cy, ho, wo)
            xin = self.conv_bn(x, c2, net.bn2, ACT_RELU, out=View(cat, 0, cx))
            if self.conv1_conv2_s8(xin, b0.conv1, b0.bn1, b0.conv2, b0.bn2, out=View(cat, cx, cy)) is None:
                y = self.conv_bn(xin, b0.conv1, b0.bn1, ACT_RELU)
                self.conv_bn(y, b0.conv2, b0.bn2, ACT_RELU, out=View(cat, cx, cy))

            def fold(conv, bn):
                g, b = self.dev_param(bn.weight), self.dev_param(bn.bias)
                mu, var = self.dev_param(bn.running_mean), self.dev_param(bn.running_var)
                sc = g / torch.sqrt(var + bn.eps)
                return self.dev_param(conv.weight) * sc[:, None, None, None], b - mu * sc
            wd, shd = fold(b0.downsample[0], b0.downsample[1])
            w3, sh3 = fold(b0.conv3, b0.bn3)
            cout = b0.conv3.out_channels
            x = self.conv(View(cat), torch.cat([wd, w3], 1), View(self.new(n, cout, ho, wo)),
                          scale=torch.ones(cout, device=self.dev), shift=shd + sh3, act=ACT_RELU)
            blocks = blocks[1:]
        else:
            x = self.conv_bn(x, net.conv2, net.bn2, ACT_RELU)
        t1_s8 = self.use_s8 and not self._exact and os.environ.get("OTPOSE_T1_S8", "1") != "0" \
            and os.environ.get("OTPOSE_S8_STRIDE2", "1") != "0"
        for bi, blk in enumerate(blocks):
            x = self.bottleneck(blk, x, s8_only=t1_s8 and bi == len(blocks) - 1)
        ys = [x]
        for s in (2, 3, 4):
            trans = getattr(net, f"transition{s - 1}")
            xs = []
            live = [i for i, tr in enumerate(trans) if tr is not None]
            t1_s8 = s == 2 and self.use_s8 and not self._exact and os.environ.get("OTPOSE_T1_S8", "1") != "0"
            if (s >= 3 or t1_s8) and self.use_s8 and os.environ.get("OTPOSE_S8_STRIDE2", "1") != "0":
                # the lowest-resolution tensor feeds the new branch's stride-2 conv AND its own branch of the next module (s >= 3) /
                # the width-change conv of transition1 (s = 2: 256 -> 48 and 256 -> 96 stride 2 ran at 100-160 TFLOP/s on the
                # fp32-input kernel): one pack pass before the streams fork serves both
                self.s8_image(ys[-1], want_c4=s >= 3 and os.environ.get("OTPOSE_S8_RESIDUAL", "1") == "0")
            self.fork(range(1, len(live)))                         # the transition convs only share their inputs
            for i, tr in enumerate(trans):
                if tr is None:
                    xs.append(ys[i])
                    continue
                self.on_stream(live.index(i))
                if isinstance(tr[0], torch.nn.Conv2d):            # same-resolution width change
                    y_ = self.conv_bn_s8(ys[i], tr[0], tr[1], ACT_RELU) if t1_s8 else None
                    xs.append(y_ if y_ is not None else self.conv_bn(ys[i], tr[0], tr[1], ACT_RELU))
                else:                                              # new branch from the last tensor
                    z = ys[-1]
                    zs = self.s2_chain_s8([(step[0], step[1], True) for step in tr], z, ACT_RELU) if (s >= 3 or t1_s8) else None
                    if zs is None:
                        for step in tr:
                            z = self.conv_bn(z, step[0], step[1], ACT_RELU)
                        zs = z
                    xs.append(zs)
            self.on_stream(0)
            self.join(range(1, len(live)))
            ys = xs
            mods = list(getattr(net, f"stage{s}"))
            chain = os.environ.get("OTPOSE_CHAIN_MODULES", "1") != "0"
            for mi, mod in enumerate(mods):
                nxt = mods[mi + 1] if mi + 1 < len(mods) else None
                # the next module's branch i continues on the stream of this module's row i (same count of rows and branches)
                cont = chain and nxt is not None and len(mod.fuse_layers) == nxt.num_branches and mod.num_branches > 1
                self._exact = s == 4 and mi >= len(mods) - self.f32_tail
                ys = self.hr_module(mod, ys, fork_in=not (mi > 0 and prev_cont), join_out=not cont) if mi > 0 else \
                    self.hr_module(mod, ys, fork_in=True, join_out=not cont)
                prev_cont = cont
        fl = net.final_layer
        rough = View(self.new(self.F * self.B, self.J, self.h, self.w))
        self._exact = self.f32_tail > 0
        out = self.conv(ys[0], fl.weight, rough, 1, fl.padding[0], 1, bias=fl.bias)
        self._exact = False
        return out

    # ---- ConvTransformer (reference model/ConvVideoTransformer.py:123-184, model/blocks.py:264-280,400-453) ----
    def v3(self, t3):
        """(B, C, T) tensor as a 4-D view for the conv kernel."""
        b, c, t = t3.shape
        return View(t3.view(b, c, 1, t))

    def tblock(self, blk, x, stride):
        B, C, T = x.shape
        To = T if stride == 1 else (T + 2 - 3) // 2 + 1
        L = self.lib
        a = blk.attn
        if self.use_flow_fused and stride == 1 and ops.flow_block_supported(blk, C, T):
            # the C = 17 flow encoder: per-token chains in two launches around the attention (csrc/flowenc.hip); the generic
            # kernels below are latency bound at this width (12 launches per block on the serial path)
            fp, bp = ops.pack_flow_front(blk, self.dev), ops.pack_flow_back(blk, self.dev)
            self._keep += [fp, bp]
            q, k, v, att, out = (self.new(B, C, T) for _ in range(5))
            self.call(L.otp_flow_front, "otp_flow_front", hip.ptr(x), hip.ptr(fp), hip.ptr(q), hip.ptr(k), hip.ptr(v),
                      B, C, T, blk.ln1.eps)
            nbytes = L.otp_chan_attn_workspace(B, C, T, a.n_head)
            ws = self.new(max(nbytes // 4, 1))
            self.call(L.otp_chan_attn, "otp_chan_attn", hip.ptr(q), hip.ptr(k), hip.ptr(v), hip.ptr(att), hip.ptr(ws),
                      nbytes, B, C, T, a.n_head, a.scale)
            self.call(L.otp_flow_back, "otp_flow_back", hip.ptr(x), hip.ptr(att), hip.ptr(bp), hip.ptr(out), B, C,
                      blk.mlp[0].out_channels, T, blk.ln2.eps)
            return out
        ln1 = self.new(B, C, T)
        skip = self.new(B, C, To) if stride > 1 else None
        p = self.dev_param
        self.call(L.otp_ln_channel, "otp_ln_channel", hip.ptr(x), hip.ptr(p(blk.ln1.weight)), hip.ptr(p(blk.ln1.bias)),
                  hip.ptr(ln1), hip.ptr(skip), B, C, T, blk.ln1.eps)
        q, k, v = self.new(B, C, To), self.new(B, C, To), self.new(B, C, To)
        # (the split-product kernels of csrc/densex.hip also hold C = 204, the 7-frame window of BASELINE configs[4]; the exact-fp32
        #  ones of csrc/dense.hip only C = 136 - until round 4 this line asked the fp32 predicate first and cfg5 ran its
        #  projections and the q / k / v front end on the generic kernels: 23 ms of kernel time per forward)
        dx3 = self.use_dense_cc and self.use_x3 and ops.dense_x3_supported(C, To)
        dense = dx3 or (self.use_dense_cc and ops.dense_cc_supported(C, To))
        if dense:
            packs = [ops.pack_dense_cc(m.weight.to(self.dev), None, m.bias.to(self.dev), x3=dx3)
                     for m in (a.query, a.key, a.value)]
        if dense and stride == 1 and self.use_qkv_front:
            # depthwise convs + LayerNorms + the three projections in one launch (csrc/dense.hip, qkv_front_kernel)
            table = ops.pack_qkv_table(*[p(t) for t in (a.query_conv.weight, a.key_conv.weight, a.value_conv.weight,
                                                         a.query_norm.weight, a.query_norm.bias, a.key_norm.weight,
                                                         a.key_norm.bias, a.value_norm.weight, a.value_norm.bias)])
            self._keep += [table, *packs]
            self.call((L.otp_qkv_front_h1 if self.half_products else L.otp_qkv_front_x3) if dx3 else L.otp_qkv_front, "otp_qkv_front",
                      hip.ptr(ln1), hip.ptr(table),
                      *[hip.ptr(t) for t in packs],
                      hip.ptr(q), hip.ptr(k), hip.ptr(v), B, C, T, a.query_norm.eps)
        else:
            qn, kn, vn = self.new(B, C, To), self.new(B, C, To), self.new(B, C, To)
            self.call(L.otp_dwconv_ln3, "otp_dwconv_ln3", hip.ptr(ln1), hip.ptr(p(a.query_conv.weight)),
                      hip.ptr(p(a.key_conv.weight)), hip.ptr(p(a.value_conv.weight)),
                      hip.ptr(p(a.query_norm.weight)), hip.ptr(p(a.query_norm.bias)),
                      hip.ptr(p(a.key_norm.weight)), hip.ptr(p(a.key_norm.bias)),
                      hip.ptr(p(a.value_norm.weight)), hip.ptr(p(a.value_norm.bias)),
                      hip.ptr(qn), hip.ptr(kn), hip.ptr(vn), B, C, T, stride, a.query_norm.eps)
            if dense:
                # the three projections as one launch of the register-resident-input kernel (csrc/dense.hip)
                self.dense((qn, kn, vn), packs, None, (q, k, v), B, C, To)
            else:
                self.conv(self.v3(qn), a.query.weight, self.v3(q), bias=a.query.bias)
                self.conv(self.v3(kn), a.key.weight, self.v3(k), bias=a.key.bias)
                self.conv(self.v3(vn), a.value.weight, self.v3(v), bias=a.value.bias)
        att = self.new(B, C, To)
        nbytes = L.otp_chan_attn_workspace(B, C, To, a.n_head)
        ws = self.new(max(nbytes // 4, 1))
        self.call(L.otp_chan_attn, "otp_chan_attn", hip.ptr(q), hip.ptr(k), hip.ptr(v), hip.ptr(att), hip.ptr(ws),
                  nbytes, B, C, To, a.n_head, a.scale)
        # y = pool_skip(x) + scale_attn * (proj(att) + b)   (eval: dropout / drop-path are identities)
        sa = blk.drop_path_attn.scale.detach().reshape(-1)
        y = self.new(B, C, To)
        if dense:
            sad = sa.to(self.dev, torch.float32)
            pk = ops.pack_dense_cc(a.proj.weight.to(self.dev), sad, a.proj.bias.detach().to(self.dev) * sad,
                                   x3=self.use_x3 and ops.dense_x3_supported(C, To))
            self.dense((att,), [pk], (skip if stride > 1 else x,), (y,), B, C, To)
        else:
            self.conv(self.v3(att), a.proj.weight, self.v3(y), scale=sa,
                      shift=a.proj.bias.detach() * sa.to(a.proj.bias.device), res=self.v3(skip if stride > 1 else x))
        sm = blk.drop_path_mlp.scale.detach().reshape(-1)
        out = self.new(B, C, To)
        hid = blk.mlp[0].out_channels
        if self.use_fused_mlp and self.use_x3 and ops.mlp_x3_supported(C, hid, To):
            # the same single launch with split-half products on the 16-bit matrix cores (csrc/mlpx.hip)
            dev = lambda t: t.detach().to(self.dev, torch.float32)     # noqa: E731
            packed = ops.pack_mlp_x3_weights(dev(blk.mlp[0].weight), dev(blk.mlp[0].bias), dev(blk.mlp[3].weight), half=self.half_products)
            scd = dev(sm).contiguous()
            shd = (dev(blk.mlp[3].bias) * scd).contiguous()
            self._keep += [packed, scd, shd]
            self.call(L.otp_ln_mlp_h1 if self.half_products else L.otp_ln_mlp_x3, "otp_ln_mlp_x3", hip.ptr(y), hip.ptr(p(blk.ln2.weight)),
                      hip.ptr(p(blk.ln2.bias)),
                      blk.ln2.eps, hip.ptr(packed), hip.ptr(scd), hip.ptr(shd), hip.ptr(out), B, C, hid, To)
            return out
        if self.use_fused_mlp and ops.mlp_fused_supported(C, hid, To):
            # ln2 -> Conv1d -> GELU -> Conv1d + residual as one launch, hidden activation kept on chip (csrc/mlp.hip)
            dev = lambda t: t.detach().to(self.dev, torch.float32)     # noqa: E731
            packed = ops.pack_mlp_weights(dev(blk.mlp[0].weight), dev(blk.mlp[0].bias), dev(blk.mlp[3].weight))
            scd = dev(sm).contiguous()
            shd = (dev(blk.mlp[3].bias) * scd).contiguous()
            self._keep += [packed, scd, shd]
            self.call(L.otp_ln_mlp_fused, "otp_ln_mlp_fused", hip.ptr(y), hip.ptr(p(blk.ln2.weight)), hip.ptr(p(blk.ln2.bias)),
                      blk.ln2.eps, hip.ptr(packed), hip.ptr(scd), hip.ptr(shd), hip.ptr(out), B, C, hid, To)
            return out
        ln2 = self.new(B, C, To)
        self.call(L.otp_ln_channel, "otp_ln_channel", hip.ptr(y), hip.ptr(p(blk.ln2.weight)), hip.ptr(p(blk.ln2.bias)),
                  hip.ptr(ln2), None, B, C, To, blk.ln2.eps)
        hdn = self.new(B, 4 * C, To)
        self.conv(self.v3(ln2), blk.mlp[0].weight, self.v3(hdn), bias=blk.mlp[0].bias, act=ACT_GELU)
        self.conv(self.v3(hdn), blk.mlp[3].weight, self.v3(out), scale=sm,
                  shift=blk.mlp[3].bias.detach() * sm.to(blk.mlp[3].bias.device), res=self.v3(y))
        return out

    def conv_transformer(self, ct, x, stacked):
        """x (B, C, T) already holds input + positional table; writes levels into stacked (B, L*C, T)."""
        B, C, T = x.shape
        L = self.lib
        for blk in ct.stem:
            x = self.tblock(blk, x, 1)
        self.call(L.otp_upsample_linear, "otp_upsample_linear", hip.ptr(x), hip.ptr(stacked), B, C, T, 1,
                  stacked.shape[1], 0)
        for i, blk in enumerate(ct.branch):
            x = self.tblock(blk, x, 2)
            f = 2 ** (i + 1)
            if x.shape[2] * f != T:
                raise RuntimeError("heat-map size must make every pyramid level divide evenly (T % 4 == 0)")
            self.call(L.otp_upsample_linear, "otp_upsample_linear", hip.ptr(x), hip.ptr(stacked), B, C, x.shape[2], f,
                      stacked.shape[1], (i + 1) * C)

    # ---- RSB heads (reference model/RSB.py:77-103) --------------------------------------------------
    def rsb_block(self, blk, x: View) -> View:
        n, _, h, w = x.t.shape
        bc = blk.branch_ch
        cbr = lambda m, inp, out, act, in2=None, res=None: self.conv(   # noqa: E731
            inp, m.conv.weight, out, 1, m.conv.padding[0], 1, bn=m.bn, bias=m.conv.bias, act=act, in2=in2, res=res)
        t = self.new(n, 4 * bc, h, w)
        cat = self.new(n, 4 * bc, h, w)
        cbr(blk.conv_bn_relu1, x, View(t), ACT_RELU)
        s = [View(t, i * bc, bc) for i in range(4)]
        tmp = lambda: View(self.new(n, bc, h, w))                       # noqa: E731
        # the staircase of RSB.py:80-92 is a DAG of depth 7, not a chain of 10: on the main stream the three convs off the
        # critical path (2_2, 3_2, 3_3) run on a side stream (these launches are tiny and latency-bound)
        par = self.multi_stream and self._sid == 0
        side = (lambda: (self.fork((1,)), self.on_stream(1))) if par else (lambda: None)      # noqa: E731
        main = (lambda: self.on_stream(0)) if par else (lambda: None)                          # noqa: E731
        o11 = cbr(blk.conv_bn_relu2_1_1, s[0], View(cat, 0, bc), ACT_RELU)
        o21 = cbr(blk.conv_bn_relu2_2_1, s[1], tmp(), ACT_RELU, in2=o11)
        side()
        o22 = cbr(blk.conv_bn_relu2_2_2, o21, View(cat, bc, bc), ACT_RELU)
        main()
        o31 = cbr(blk.conv_bn_relu2_3_1, s[2], tmp(), ACT_RELU, in2=o21)
        side()
        o32 = cbr(blk.conv_bn_relu2_3_2, o31, tmp(), ACT_RELU, in2=o22)
        main()
        o41 = cbr(blk.conv_bn_relu2_4_1, s[3], tmp(), ACT_RELU, in2=o31)
        if par:
            self.join((1,))
        o42 = cbr(blk.conv_bn_relu2_4_2, o41, tmp(), ACT_RELU, in2=o32)
        if par:
            self.on_stream(1)
        o33 = cbr(blk.conv_bn_relu2_3_3, o32, View(cat, 2 * bc, bc), ACT_RELU)
        if par:
            self.on_stream(0)
            self.join((1,))
        o43 = cbr(blk.conv_bn_relu2_4_3, o42, tmp(), ACT_RELU, in2=o33)
        cbr(blk.conv_bn_relu2_4_4, o43, View(cat, 3 * bc, bc), ACT_RELU)
        res = x
        if blk.downsample is not None:
            res = cbr(blk.downsample, x, View(self.new(n, blk.conv_bn_relu3.conv.out_channels, h, w)), ACT_NONE)
        out = View(self.new(n, blk.conv_bn_relu3.conv.out_channels, h, w))
        return cbr(blk.conv_bn_relu3, View(cat), out, ACT_RELU, res=res)

    def rsb_chain(self, chain, x: View) -> View:
        for blk in chain.layers:
            x = self.rsb_block(blk, x)
        return x

    # ---- whole forward ------------------------------------------------------------------------------
    def _build(self):
        m, B, J, h, w = self.model, self.B, self.J, self.h, self.w
        L, T = self.lib, self.h * self.w
        self.inp = self._given[0] if self._given[0] is not None else self.new(B, 3 * self.F, self.H_img, self.W_img)
        self.margin = self._given[1] if self._given[1] is not None else self.new(B, self.F - 1)
        assert tuple(self.inp.shape) == (B, 3 * self.F, self.H_img, self.W_img) and self.inp.is_contiguous()
        rough = self.hrnet(m.rough_pose_estimation_net, View(self.inp)).t
        self.hr_end = len(self.ops)               # launches [0, hr_end) = the backbone, the rest = everything behind it

        total, squeezed, inter = self.new(B, J, h, w), self.new(B, J, h, w), self.new(B, J, h, w)
        flow_in = self.new(B, J, T)
        pe_f = self.dev_param(m.flow_encoder.pos_embd[0, :, :T])
        self.call(L.otp_glue_total_n, "otp_glue_total", hip.ptr(rough), hip.ptr(total), hip.ptr(squeezed), hip.ptr(inter),
                  hip.ptr(flow_in), hip.ptr(pe_f), B, J, T, self.F)
        ctx = self.new(B, J, T)
        # def_fuse only needs `total`: it runs on a side stream next to the flow encoder and the temporal encoders
        self.fork((2,))
        self.on_stream(2)
        def_h = self.rsb_chain(m.def_fuse, View(total))
        self.on_stream(0)
        self.conv_transformer(m.flow_encoder, flow_in, ctx)
        D = m.num_frames * J                       # stacked maps per joint: 8 (5-frame window) or 12 (7 frames)
        x1, x2, prev_b = self.new(B, D, T), self.new(B, D, T), self.new(B, J, h, w)
        pe1 = self.dev_param(m.temporal_encoder1.pos_embd[0, :, :T])
        pe2 = self.dev_param(m.temporal_encoder2.pos_embd[0, :, :T])
        self.call(L.otp_glue_stack_n, "otp_glue_stack", hip.ptr(rough), hip.ptr(self.margin), hip.ptr(squeezed),
                  hip.ptr(inter), hip.ptr(ctx), hip.ptr(pe1), hip.ptr(pe2), hip.ptr(x1), hip.ptr(x2), hip.ptr(prev_b),
                  B, J, T, self.F)
        levels = m.scale_arch[-1] + 1
        # the stacked pyramid levels (3 x 136 = 408 channels) feed the final 1x1 layers; padded with zero channels to a multiple
        # of 32 so that those run on the split-product pointwise kernel (432 workgroups, ~45 us) instead of the generic
        # direct kernel (408 -> 17 is one 240-workgroup launch of 13 dependent chunks there: 129 us on the critical path)
        cs_ = levels * D
        cpad = (-cs_) % 32 if (self.use_x3 and m.final_layer1.kernel_size == (1, 1)) else 0
        s1, s2 = self.new(B, cs_ + cpad, T), self.new(B, cs_ + cpad, T)
        if cpad:
            s1[:, cs_:].zero_()
            s2[:, cs_:].zero_()
        # the two temporal encoders are independent: two streams.  Each is followed on its own stream by its final 1x1 layer,
        # which writes straight into the channel-concatenated tensor (OTPose.py:372-378); def_heatmaps feeds the DCN as a
        # dense (B, J, h, w) tensor and the concat as a channel slice: copied once, on def_fuse's stream
        cat3 = self.new(B, 3 * J, h, w)
        def final(fl, s, i):
            wt = fl.weight
            if cpad:
                wt = torch.cat([wt.detach().to(self.dev, torch.float32),
                                torch.zeros(wt.shape[0], cpad, 1, 1, device=self.dev)], 1)
            self.conv(View(s.view(B, cs_ + cpad, h, w)), wt, View(cat3, i * J, J), 1, fl.padding[0], 1, bias=fl.bias)

        self.on_stream(2)
        self.copy_into(def_h, View(cat3, 2 * J, J))
        self.on_stream(0)
        self.fork((1,))
        self.conv_transformer(m.temporal_encoder1, x1, s1)
        final(m.final_layer1, s1, 0)
        self.on_stream(0 if os.environ.get("OTPOSE_TE_SERIAL") == "1" else 1)
        self.conv_transformer(m.temporal_encoder2, x2, s2)
        final(m.final_layer2, s2, 1)
        self.on_stream(0)
        self.join((1, 2))
        trans = self.rsb_chain(m.offset_mask_combine_conv, View(cat3))
        out = self.new(B, J, h, w)
        nd = len(m.deformable_conv_dilations)
        dils = [int(d) for d in m.deformable_conv_dilations]
        cin_t = trans.C
        if (self.use_dcn_fused and trans.coff == 0 and trans.C == trans.ctot and ops.dcn_fused_supported(cin_t, J, h, w, nd)
                and all(c[0].kernel_size == (3, 3) and c[0].bias is None and c[0].stride == (1, 1)
                        for c in list(m.offsets_list) + list(m.masks_list))
                and all(tuple(c[0].padding) == (d, d) == tuple(c[0].dilation)
                        for cs in (m.offsets_list, m.masks_list) for c, d in zip(cs, dils))
                and all(tuple(mm.deform_conv.weight.shape) == (J, J, 3, 3) and mm.deform_conv.deformable_groups == J
                        and mm.deform_conv.groups == 1 and _pair(mm.deform_conv.stride) == (1, 1)
                        and _pair(mm.deform_conv.padding) == (d, d) == _pair(mm.deform_conv.dilation)
                        for mm, d in zip(m.modulated_deform_conv_list, dils))):
            # the ten offset / mask convs and the five DCN gathers as ONE launch; the 459
            # offset / mask channels per pixel and dilation never reach HBM (csrc/dcn_fused.hip)
            dcs = [mm.deform_conv for mm in m.modulated_deform_conv_list]
            packed = ops.pack_dcn_fused([self.dev_param(c[0].weight) for c in m.offsets_list],
                                        [self.dev_param(c[0].weight) for c in m.masks_list],
                                        [self.dev_param(dc.weight) for dc in dcs],
                                        [None if dc.bias is None else self.dev_param(dc.bias) for dc in dcs])
            # the split NHWC copy of trans (128 bytes per pixel) + the zero-bordered copy of the heat-map planes
            ws = self.new(int(L.otp_dcn_fused_workspace(B, h, w)) // 4)
            dl = (ctypes.c_int * nd)(*dils)
            self._keep += [packed, dl]
            self.call(L.otp_dcn_fused_forward, "otp_dcn_fused_forward", hip.ptr(trans.t), hip.ptr(def_h.t), hip.ptr(packed),
                      hip.ptr(out), hip.ptr(ws), ws.numel() * 4, B, cin_t, J, h, w, dl, nd, 1.0 / nd)
            self._finish(out, rough, inter, prev_b, ctx, squeezed, total)
            return
        off_buf, msk_buf = self.new(B, J * 18, h, w), self.new(B, J * 9, h, w)
        for i, d in enumerate(m.deformable_conv_dilations):
            self.conv(trans, m.offsets_list[i][0].weight, View(off_buf), 1, d, d)
            self.conv(trans, m.masks_list[i][0].weight, View(msk_buf), 1, d, d)
            dc = m.modulated_deform_conv_list[i].deform_conv
            wt, bs = self.dev_param(dc.weight), self.dev_param(dc.bias)
            self.call(L.otp_mdcn_forward, "otp_mdcn_forward", hip.ptr(def_h.t), hip.ptr(off_buf), hip.ptr(msk_buf),
                      hip.ptr(wt), hip.ptr(bs), hip.ptr(out), B, J, h, w, J, 3, 3, 1, d, d, 1, J,
                      1.0 / nd, 0.0 if i == 0 else 1.0, 0)
        self._finish(out, rough, inter, prev_b, ctx, squeezed, total)

    def _finish(self, out, rough, inter, prev_b, ctx, squeezed, total):
        """Last launch of the forward: the range guard's device side (csrc/range.hip) - if any 16-bit-operand kernel of this
        process saw a value beyond a half's range, the heat-maps leave as NaN instead of as finite numbers computed from an
        overflowed operand (the host side raises: :meth:`run`, :meth:`check_range`)."""
        B, J, h, w = self.B, self.J, self.h, self.w
        if self.use_x3 or getattr(self, "h16", False):
            self.call(self.lib.otp_range_poison, "otp_range_poison", hip.ptr(out), out.numel())
        self.outputs = (out, rough, inter, prev_b, ctx.view(B, J, h, w), squeezed, total)
        torch.cuda.synchronize(self.dev)

    _RANGE_KERNELS = {1: "convx (3x3 / 1x1 convolutions from fp32 NCHW)", 2: "convs (BasicBlock convolutions on S8 records)",
                      3: "convs2 (stride-2 convolutions on S8 records)", 4: "pointx (1x1 convolutions)", 5: "stem convolution",
                      6: "S8 conversion / fuse passes", 7: "mlpx (TransformerBlock MLP)", 8: "densex (q / k / v / proj projections)",
                      9: "channel attention", 10: "dcn_fused (warping head)", 11: "fp16 engine kernels"}

    def _raise_range(self, code):
        raise FloatingPointError(
            "otpose_amd: a value left the range of the half-precision operand pieces (|x| >= 65504, or a non-finite sum) in the "
            f"{self._RANGE_KERNELS.get(code, 'kernel family %d' % code)} of an eval forward; the heat-maps of that forward were "
            "NaN-filled on the device.  The reference computes these layers in fp32: rescale the input / check the checkpoint's "
            "BatchNorm statistics, or run the exact-fp32 kernels (OTPOSE_CONV_MATH=f32).")

    def check_range(self):
        """Synchronise and raise FloatingPointError if a forward since the last check overflowed the half-piece arithmetic."""
        torch.cuda.synchronize(self.dev)
        code = self.lib.otp_range_flag_read(1)
        if code:
            self._raise_range(code)

    def __del__(self):
        try:
            if getattr(self, "graph", None) is not None:
                torch.cuda.synchronize(self.dev)
        except Exception:                           # interpreter shutdown
            pass

    def copy_into(self, src: View, dst: View):
        """dst channels <- src (per-sample strided copy through the upsample kernel with f = 1)."""
        n, _, h, w = src.t.shape
        assert src.coff == 0 and src.C == src.ctot
        self.call(self.lib.otp_upsample_linear, "otp_upsample_linear", hip.ptr(src.t), hip.ptr(dst.t), n, src.C, h * w, 1,
                  dst.ctot, dst.coff)

    # ---------------------------------------------------------------------------------------------
    def _launch_all(self, first=0, last=None):
        """Enqueue launches [first, last) of the list on the current stream (and the side streams they were emitted on)."""
        main = hip.stream_of(self.inp)
        handles = [main] + [s_.cuda_stream for s_ in self._side]
        for op, sid in self.ops[first:last]:
            self._stream = handles[sid] if sid > 0 else main
            op()
        self._stream = main

    def run(self, x, margin, alias_outputs: bool = False):
        """One forward.  Returns fresh tensors like the reference's ``OTPose.forward`` (7 clones, 85 MB at batch 16:
        ~30 us of device copies next to a 57 ms forward); ``alias_outputs=True`` hands out the engine's static
        buffers instead, which the NEXT ``run`` overwrites (benchmark loops, callers that consume at once)."""
        if not x.is_cuda:
            raise RuntimeError("OTPose.forward expects CUDA (HIP) tensors; there is no CPU path")
        if self.range_check != "off":
            # deferred half of the range guard: a violation recorded by any forward that has completed by now (no synchronisation)
            code = self.lib.otp_range_flag_read(1)
            if code:
                self._raise_range(code)
        if x.dtype == torch.uint8:
            # (B, 5, H, W, 3) uint8 frames: normalise + concatenate straight into the stem conv's input buffer
            ops.frames_to_clip(x, out=self.inp)
        elif x.data_ptr() != self.inp.data_ptr():         # (a caller that fills OTPose.input_buffers() in place skips the copy)
            self.inp.copy_(x)
        if margin.data_ptr() != self.margin.data_ptr():
            self.margin.copy_(margin.to(torch.float32))
        if self.use_graph:
            from . import parallel
            if not parallel.graph_replay_safe():
                # an RCCL communicator of this process was destroyed: replaying a hipGraph after that segfaults inside the
                # runtime (torch 2.10 / ROCm 7.2) - eager launches from here on.  A graph already captured is kept alive,
                # never replayed (tearing it down before eager launches on its streams is the other known crash)
                self.use_graph = False
                self._parked_graph, self.graph = self.graph, None
        if self.use_graph and self.graph is None:
            self._launch_all()                      # warm-up (sets kernel attributes) before capture
            torch.cuda.synchronize(self.dev)
            try:
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    self._launch_all()
                self.graph = g
            except Exception as e:                  # pragma: no cover - depends on the runtime
                warnings.warn(f"hipGraph capture failed ({e}); launching eagerly")
                self.use_graph = False
                torch.cuda.synchronize(self.dev)
        if self.graph is not None:
            self.graph.replay()
        else:
            self._launch_all()
        if self.range_check == "sync":
            self.check_range()
        if alias_outputs:
            return self.outputs
        return tuple(o.clone() for o in self.outputs)
