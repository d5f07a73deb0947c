"""fp16-storage inference engine: ``cfg.MODEL.DTYPE = "fp16"`` (BASELINE.json configs[4]).

The HRNet backbone (reference model/HRNet.py:116-152 - 86 % of the forward's FLOPs and almost all of its activation bytes) runs
on H8 images - every activation between its layers is IEEE half, [N][C / 8][H * W] records of 8 channels (csrc/h16.hip) - with
ONE f16 MFMA product per multiply where the fp32 engine spends three on two-piece operands, fp32 accumulation, BatchNorm (eval)
folded into the half weights (times a per-layer power of two) and an fp32 shift.  ``rough`` leaves the backbone as fp32 NCHW
heat-maps; the temporal encoders, RSB heads and the warping head behind it are the parent engine's launches (fp32 tensors; with
``MODEL.DTYPE = "fp16"`` their matrix products also take half operands once - see ``InferenceEngine.half_products``).

The reference never runs below fp32 (no AMP anywhere: SURVEY.md), although its native DCN op dispatches half
(thirdparty/deform_conv/src/deform_conv_cuda_kernel.cu:719), so this engine is an extension: validated against the fp32 engine
and the oracle with a stated tolerance (tests/test_gpu_h16_engine.py), never the default of cfg2.  There is no fallback inside:
a layer the fp16 kernels do not cover raises at build time.
"""
from __future__ import annotations

import ctypes
import os
from typing import List

import torch

from . import hip, ops
from .engine import InferenceEngine
from .ops import ACT_NONE, ACT_RELU, H8, View


class InferenceEngineH16(InferenceEngine):
    h16 = True

    # ---- emitters -------------------------------------------------------------------------------------------------------------
    def h8(self, n, c, h, w) -> H8:
        """A static H8 activation buffer (2 bytes per element)."""
        return H8(self.new((n * c * h * w + 1) // 2), n, c, h, w)

    @staticmethod
    def _is3x3(conv):
        return (conv.kernel_size == (3, 3) and conv.padding == (1, 1) and conv.dilation == (1, 1) and conv.groups == 1
                and conv.stride in ((1, 1), (2, 2)))

    def h_conv3(self, x: H8, conv, bn, act, res: H8 = None) -> H8:
        """act(bn(conv3x3(x)) (+ res)) on H8 images (csrc/h16.hip: otp_h16_conv3x3)."""
        if not self._is3x3(conv) or conv.bias is not None or conv.in_channels != x.C:
            raise RuntimeError(f"fp16 engine: no kernel for {conv}")
        sc, sh = self._bn_fold(bn)
        w = self.dev_param(conv.weight)
        k = ops.h16_weight_exponent(w, sc)
        wp = ops.pack_h16_conv_weight(w, sc, k)
        s = conv.stride[0]
        out = self.h8(x.N, conv.out_channels, x.H // s, x.W // s)
        d = ops.h16_conv_desc(x, conv.out_channels, s, act, out, res, k)
        if not ops.h16_conv_supported(d):
            raise RuntimeError(f"fp16 engine: otp_h16_conv3x3 does not cover {conv} on a {x.N} x {x.C} x {x.H} x {x.W} input")
        self._keep += [wp, d]
        self.call(self.lib.otp_h16_conv3x3, "otp_h16_conv3x3", hip.ptr(x.t), hip.ptr(wp), hip.ptr(sh),
                  hip.ptr(res.t) if res is not None else None, hip.ptr(out.t), d)
        return out

    def h_pw(self, x: H8, conv, bn, relu, res: H8 = None, out=None):
        """act(bn(conv1x1(x)) (+ res)): H8 -> H8, or -> an fp32 NCHW :class:`View` (``out``)."""
        if conv.kernel_size != (1, 1) or conv.stride != (1, 1) or conv.padding != (0, 0) or conv.groups != 1 or conv.in_channels != x.C:
            raise RuntimeError(f"fp16 engine: no kernel for {conv}")
        cout = conv.out_channels
        if not ops.h16_pointwise_supported(x.C, cout):
            raise RuntimeError(f"fp16 engine: otp_h16_pointwise does not cover {conv}")
        w = self.dev_param(conv.weight)
        if bn is not None:
            sc, sh = self._bn_fold(bn)
            if conv.bias is not None:
                sh = (sh + self.dev_param(conv.bias) * sc).contiguous()
        else:
            sc, sh = None, (self.dev_param(conv.bias) if conv.bias is not None else None)
        k = ops.h16_weight_exponent(w, sc)
        pk = ops.pack_h16_pointwise(w, sc, sh, k)
        self._keep += [pk, sh]
        f32 = isinstance(out, View)
        if out is None:
            out = self.h8(x.N, cout, x.H, x.W)
        otot, ooff = (out.ctot, out.coff) if f32 else (out.gtot, out.goff)
        self.call(self.lib.otp_h16_pointwise, "otp_h16_pointwise", hip.ptr(x.t), hip.ptr(pk), hip.ptr(res.t) if res is not None else None,
                  hip.ptr(out.t), int(f32), x.N, x.C, cout, x.H * x.W, x.gtot, x.goff, res.gtot if res is not None else 0,
                  res.goff if res is not None else 0, otot, ooff, int(relu), float(2.0 ** -k))
        return out

    def h_upsample_add(self, lows: List[H8], factors, res: H8, relu=True) -> H8:
        out = self.h8(res.N, res.C, res.H, res.W)
        lp = (ctypes.c_void_p * len(lows))(*[hip.ptr(v.t) for v in lows])
        fp = (ctypes.c_int * len(lows))(*[int(f) for f in factors])
        self._keep += [lp, fp]
        self.call(self.lib.otp_h16_upsample_add, "otp_h16_upsample_add", lp, fp, len(lows), hip.ptr(res.t), hip.ptr(out.t), res.N, res.C,
                  res.H, res.W, int(relu))
        return out

    # ---- HRNet (reference model/HRNet.py:116-152) ------------------------------------------------------------------------------
    def h_bottleneck(self, blk, x: H8) -> H8:
        """model/HRNet.py:551-571: 1x1 -> 3x3 -> 1x1 (+ shortcut), ReLU after each."""
        res = x
        if blk.downsample is not None:
            self.fork((1,))                                      # the 1x1 shortcut only needs x: beside conv1 / conv2
            self.on_stream(1)
            res = self.h_pw(x, blk.downsample[0], blk.downsample[1], relu=False)
            self.on_stream(0)
        y = self.h_pw(x, blk.conv1, blk.bn1, relu=True)
        y = self.h_conv3(y, blk.conv2, blk.bn2, ACT_RELU)
        if blk.downsample is not None:
            self.join((1,))
        return self.h_pw(y, blk.conv3, blk.bn3, relu=True, res=res)

    def h_module(self, mod, xs: List[H8], fork_in=True, join_out=True) -> List[H8]:
        """One HighResolutionModule (model/HRNet.py:478-496): branches of four BasicBlocks (:500-530), then the fuse rows
        y_i = relu(sum_j f_ij(x_j)) (:487-494).  Branch i and row i run on stream i (see InferenceEngine.hr_module)."""
        n = mod.num_branches
        xs = list(xs)
        if fork_in:
            self.fork(range(1, n))
        for i in range(n):
            self.on_stream(i)
            x = xs[i]
            for blk in mod.branches[i]:
                if getattr(blk, "downsample", None) is not None or hasattr(blk, "conv3"):
                    raise RuntimeError("fp16 engine: HighResolutionModule branches are BasicBlocks")
                y = self.h_conv3(x, blk.conv1, blk.bn1, ACT_RELU)
                x = self.h_conv3(y, blk.conv2, blk.bn2, ACT_RELU, res=x)
            xs[i] = x
        self.on_stream(0)
        self.join(range(1, n))
        if n == 1:
            return xs
        self.fork(range(1, len(mod.fuse_layers)))
        outs = []
        for i in range(len(mod.fuse_layers)):
            self.on_stream(i)
            # the identity term starts the sum; the down-sampling chains (j < i) add to it in their last conv's epilogue, the
            # up-sampled terms (j > i: 1x1 conv + BN at low resolution, :426-439) all at once in the row's tail, which applies the ReLU
            y = xs[i]
            downs, ups = [j for j in range(n) if j < i], [j for j in range(n) if j > i]
            for idx, j in enumerate(downs):
                fl = mod.fuse_layers[i][j]
                t = xs[j]
                for k in range(len(fl) - 1):
                    t = self.h_conv3(t, fl[k][0], fl[k][1], ACT_RELU)
                last = idx == len(downs) - 1 and not ups
                y = self.h_conv3(t, fl[-1][0], fl[-1][1], ACT_RELU if last else ACT_NONE, res=y)
            if ups:
                lows = [self.h_pw(xs[j], mod.fuse_layers[i][j][0], mod.fuse_layers[i][j][1], relu=False) for j in ups]
                y = self.h_upsample_add(lows, [2 ** (j - i) for j in ups], y, relu=True)
            outs.append(y)
        self.on_stream(0)
        if join_out:
            self.join(range(1, len(mod.fuse_layers)))
        return outs

    def hrnet(self, net, x_in: View) -> View:
        L = self.lib
        c1 = net.conv1
        n_in, c_in, h_in, w_in = x_in.t.shape
        if not (x_in.coff == 0 and x_in.C == c_in == 3 * self.F and c1.kernel_size == (3, 3) and c1.stride == (2, 2)
                and c1.padding == (1, 1) and c1.in_channels == 3 and c1.bias is None
                and L.otp_h16_stem_supported(n_in, self.F, h_in, w_in, c1.out_channels)):
            raise RuntimeError("fp16 engine: the stem kernel does not cover this input")
        sc, sh = self._bn_fold(net.bn1)
        pk = ops.pack_h16_stem(self.dev_param(c1.weight), sc, sh)
        self._keep.append(pk)
        x = self.h8(self.F * n_in, c1.out_channels, (h_in - 1) // 2 + 1, (w_in - 1) // 2 + 1)
        self.call(L.otp_h16_stem, "otp_h16_stem", hip.ptr(x_in.t), hip.ptr(pk), hip.ptr(x.t), n_in, self.F, h_in, w_in, c1.out_channels)
        x = self.h_conv3(x, net.conv2, net.bn2, ACT_RELU)
        for blk in net.layer1:
            x = self.h_bottleneck(blk, x)
        ys = [x]
        for s in (2, 3, 4):
            trans = getattr(net, f"transition{s - 1}")
            xs = []
            live = [i for i, tr in enumerate(trans) if tr is not None]
            self.fork(range(1, len(live)))                         # the transition convs only share their inputs
            for i, tr in enumerate(trans):
                if tr is None:
                    xs.append(ys[i])
                    continue
                self.on_stream(live.index(i))
                if isinstance(tr[0], torch.nn.Conv2d):            # same-resolution width change (:213-220)
                    xs.append(self.h_conv3(ys[i], tr[0], tr[1], ACT_RELU))
                else:                                              # new branch from the last tensor (:221-229)
                    z = ys[-1]
                    for step in tr:
                        z = self.h_conv3(z, step[0], step[1], ACT_RELU)
                    xs.append(z)
            self.on_stream(0)
            self.join(range(1, len(live)))
            ys = xs
            mods = list(getattr(net, f"stage{s}"))
            chain = os.environ.get("OTPOSE_CHAIN_MODULES", "1") != "0"
            prev_cont = False
            for mi, mod in enumerate(mods):
                nxt = mods[mi + 1] if mi + 1 < len(mods) else None
                cont = chain and nxt is not None and len(mod.fuse_layers) == nxt.num_branches and mod.num_branches > 1
                ys = self.h_module(mod, ys, fork_in=not (mi > 0 and prev_cont), join_out=not cont)
                prev_cont = cont
        fl = net.final_layer
        rough = View(self.new(self.F * self.B, self.J, self.h, self.w))
        self.h_pw(ys[0], fl, None, relu=False, out=rough)          # final 1x1 conv with bias (:108-114): fp32 heat-maps
        return rough
