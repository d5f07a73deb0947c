"""Training-mode forward of OTPose (reference model/OTPose.py:307-394 under ``model.train()``, called from
script/Common.py:118) as a composition of autograd Functions whose forward AND backward are HIP launches
(:mod:`otpose_amd.train_ops`, :func:`otpose_amd.ops.modulated_deform_conv`), so ``loss.backward()`` walks HIP kernels.

Semantics: BatchNorm2d uses batch statistics over this replica's 5B frames and updates its running statistics (the
reference's DataParallel has no SyncBN either).  Dropout(proj_pdrop) after ``proj`` and inside the MLP (blocks.py:251-253,
450) and the per-sample drop-path of both residual branches (blocks.py:298-316) are applied with masks drawn from
PyTorch's generator, like the reference's; ``model.train_dropout = False`` switches them off for the deterministic
comparison against the oracle (the masks themselves have no counterpart to check against).
What PyTorch itself executes here is plumbing: tensor views / cat / stack, and the handful of element-wise adds,
products and per-sample divisions of the glue (model/OTPose.py:320-359); every convolution, normalisation, attention,
pooling, up-sampling, DCN and loss kernel is in libotpose_hip.so.
"""
from __future__ import annotations

import math
import os

import torch
import torch.utils.checkpoint
from torch.autograd import Function

from . import bf16_ops as B16
from . import hip, ops
from . import train_ops as T


class UpsampleAddFunction(Function):
    """``relu?(res + nearest_upsample_f(low))`` (HRNet fuse rows, model/HRNet.py:426-439,488-494)."""

    @staticmethod
    def forward(ctx, low, res, f, relu):
        low, res = low.contiguous(), res.contiguous()
        n, c, hl, wl = low.shape
        out = torch.empty_like(res)
        hip.check(hip.lib().otp_upsample_add(hip.ptr(low), hip.ptr(res), hip.ptr(out), n, c, hl, wl, f, int(relu),
                                             c, 0, c, 0, c, 0, hip.stream_of(low)), "otp_upsample_add")
        ctx.save_for_backward(out if relu else None)
        ctx.cfg = (f, low.shape)
        return out

    @staticmethod
    def backward(ctx, gy):
        (out,) = ctx.saved_tensors
        f, (n, c, hl, wl) = ctx.cfg
        gy = gy.contiguous()
        gres = torch.empty_like(gy)
        glow = torch.empty((n, c, hl, wl), dtype=torch.float32, device=gy.device)
        hip.check(hip.lib().otp_upsample_add_backward(hip.ptr(gy), hip.ptr(out), hip.ptr(gres), hip.ptr(glow), n * c, hl, wl,
                                                      f, hip.stream_of(gy)), "otp_upsample_add_backward")
        return glow, gres, None, None


class StOhkwLossFunction(Function):
    """``ST_OHKW_MSELoss`` final loss (model/loss.py:25-92) with analytic gradients w.r.t. student, teacher and target
    from one fused launch; ``flags`` (J) optionally carries the globally MAX-reduced "GT has an exact-1 peak" flags."""

    @staticmethod
    def forward(ctx, s, t, g, w, flags, topk=8):
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
        gs, gt, gg = torch.empty_like(s), torch.empty_like(t), torch.empty_like(g)
        hip.check(L.otp_loss_st_ohkw_grads(hip.ptr(s), hip.ptr(t), hip.ptr(g), hip.ptr(wv), hip.ptr(fl), hip.ptr(res),
                                           hip.ptr(gs), hip.ptr(gt), hip.ptr(gg), hip.ptr(ws), nbytes, b, j, hw, int(topk),
                                           int(given), hip.stream_of(s)), "otp_loss_st_ohkw_grads")
        ctx.save_for_backward(gs, gt, gg)
        ctx.mark_non_differentiable(res)
        return res[2].clone(), res                 # res = [ohkm_loss_s, (sum_j mse_j) / J, final_loss]

    @staticmethod
    def backward(ctx, gl, _gres):
        gs, gt, gg = ctx.saved_tensors
        return gs * gl, gt * gl, gg * gl, None, None, None


def st_ohkw_loss(s, t, g, w, flags=None):
    return StOhkwLossFunction.apply(s, t, g, w, flags)[0]


class JointsMseLossFunction(Function):
    """``JointsMSE_OHKMMSELoss`` (model/loss.py:95-148, ``ohkm=True``: returns the three scalars of its dict, gradient
    flows from ``final_loss``) and ``JointMSELoss`` (model/loss.py:151-182, ``ohkm=False``) with the analytic gradient
    from the same launch."""

    @staticmethod
    def forward(ctx, o, g, w, ohkm, topk, eff):
        res, go = ops._joints_mse(o, g, w, topk, ohkm, eff, True)
        ctx.save_for_backward(go)
        ctx.mark_non_differentiable(res)
        return res[2].clone(), res

    @staticmethod
    def backward(ctx, gl, _gres):
        (go,) = ctx.saved_tensors
        return go * gl, None, None, None, None, None


class ST_OHKW_MSELoss(torch.nn.Module):
    """model/loss.py:25-92 — same constructor and call signature; returns the reference's dict."""

    def __init__(self, use_target_weight=True, topk=8):
        super().__init__()
        if not use_target_weight:
            raise NotImplementedError("the reference's use_target_weight=False branch fills only the unused teacher lists and "
                                      "fails at torch.cat of the empty student list (loss.py:73-80)")
        self.use_target_weight, self.topk = use_target_weight, topk

    def forward(self, output_s, output_t, target, target_weight, effective_num_joints: int = None, *, flags=None):
        """Reference signature (model/loss.py:25); ``flags`` (keyword only) carries the globally MAX-reduced per-joint
        ``max(gt) == 1`` flags of a data-parallel run.  One launch yields the three scalars and the gradients."""
        final, res = StOhkwLossFunction.apply(output_s, output_t, target, target_weight, flags, self.topk)
        j = output_t.shape[1]
        mse = res[1] if effective_num_joints in (None, j) else res[1] * (j / float(effective_num_joints))
        return {"ohkm_loss_s": res[0], "mse_loss_s": mse, "final_loss": final}


class JointsMSE_OHKMMSELoss(torch.nn.Module):
    """model/loss.py:95-148."""

    def __init__(self, use_target_weight, topk=8):
        super().__init__()
        self.use_target_weight, self.topk = use_target_weight, topk

    def forward(self, output, target, target_weight, effective_num_joints=None):
        w = target_weight if self.use_target_weight else None
        final, res = JointsMseLossFunction.apply(output, target, w, True, self.topk, effective_num_joints)
        return {"ohkm_loss": res[0], "mse_loss": res[1], "final_loss": final}


class JointMSELoss(torch.nn.Module):
    """model/loss.py:151-182 (``margin`` is accepted and unused, as in the reference)."""

    def __init__(self, use_target_weight):
        super().__init__()
        self.use_target_weight = use_target_weight

    def forward(self, output, target, target_weight, effective_num_joints=None, margin=None):
        w = target_weight if self.use_target_weight else None
        final, _ = JointsMseLossFunction.apply(output, target, w, False, 1, effective_num_joints)
        return final


def build_loss(cfg, **kwargs):
    """model/loss.py:185-189: LOSS.NAME -> criterion; an unknown name returns None there, here it raises."""
    name = cfg["LOSS"]["NAME"] if isinstance(cfg, dict) else cfg.LOSS.NAME
    utw = cfg["LOSS"]["USE_TARGET_WEIGHT"] if isinstance(cfg, dict) else cfg.LOSS.USE_TARGET_WEIGHT
    if name == "ST_OHKW_MSELoss":
        return ST_OHKW_MSELoss(utw)
    if name == "MSELOSS_OHKM":
        return JointsMSE_OHKMMSELoss(utw)
    raise ValueError(f"unknown LOSS.NAME {name!r}")


class TrainGraph:
    """Functional walk over the module's own parameters / buffers (same names as the reference state dict)."""

    def __init__(self, model):
        self.model = model
        # one walk over the module tree (named_parameters / named_buffers / named_modules would be three) - the TREE is kept on the
        # model between forwards (3 of the 4 ms this constructor took at cfg2: ~1400 modules; a step that synchronises every
        # iteration, as the reference's loop does for its loss meter, waits for the host at the start of the forward), the parameter
        # and buffer tables are re-read from the modules every time, so replaced / re-homed tensors are seen.  ``del
        # model._otp_module_walk`` after adding or removing sub-modules by hand.
        walk = model.__dict__.get("_otp_module_walk")
        if walk is None or walk[0] != len(model._modules):
            walk = model.__dict__["_otp_module_walk"] = (len(model._modules), dict(model.named_modules()))
        self.mods = walk[1]
        self.P, self.Bf = {}, {}
        for name, mod in self.mods.items():
            pre = name + "." if name else ""
            for k, v in mod._parameters.items():
                if v is not None:
                    self.P[pre + k] = v
            for k, v in mod._buffers.items():
                if v is not None:
                    self.Bf[pre + k] = v
        self.cfg = model.cfg
        self.taps = None                      # dict -> forward() records named intermediates (tools/grad_noise.py)
        self.stochastic = bool(getattr(model, "train_dropout", True))
        self._nbt = []

    # ---- conv / BN helpers ------------------------------------------------------------------------------------
    def count_batch(self, bn):
        """``num_batches_tracked += 1`` of every BatchNorm2d, applied in one batched launch at the end of forward()."""
        nbt = self.Bf.get(bn + ".num_batches_tracked")
        if nbt is not None:
            self._nbt.append(nbt)

    def has(self, name):
        return name in self.P

    def conv(self, p, x, stride=1, pad=0, dil=1):
        return T.conv2d(x, self.P[p + ".weight"], self.P.get(p + ".bias"), stride, pad, dil)

    def bn(self, p, x, res=None, relu=False):
        self.count_batch(p)
        return T.batch_norm_relu(x, self.P[p + ".weight"], self.P[p + ".bias"], res, self.Bf[p + ".running_mean"],
                                 self.Bf[p + ".running_var"], 0.1, 1e-5, relu)

    def conv_bn(self, conv, bn, x, stride=1, pad=0, relu=False, res=None):
        return self.bn(bn, self.conv(conv, x, stride, pad), res, relu)

    def conv1d(self, p, x):
        """nn.Conv1d(k=1) on (B, C, T)."""
        return T.conv2d(x.unsqueeze(2), self._w4(p + ".weight"), self.P.get(p + ".bias"), 1, 0, 1).squeeze(2)

    def _w4(self, name):
        """(Cout, Cin, 1) Conv1d weight as the (Cout, Cin, 1, 1) view the conv kernels take; the view writes its gradient
        into the parameter's slot."""
        w = self.P[name]
        w4 = w.unsqueeze(-1)
        slot = getattr(w, "_otp_grad_slot", None)
        if slot is not None:
            w4._otp_grad_slot, w4._otp_grad_owner = slot.unsqueeze(-1), w
        return w4

    # ---- HRNet (model/HRNet.py:116-152, 478-496, 514-571) -------------------------------------------------------
    def basic_block(self, p, x):
        y = self.conv_bn(p + ".conv1", p + ".bn1", x, 1, 1, True)
        res = x
        if self.has(p + ".downsample.0.weight"):
            res = self.conv_bn(p + ".downsample.0", p + ".downsample.1", x, 1, 0, False)
        return self.conv_bn(p + ".conv2", p + ".bn2", y, 1, 1, True, res)

    def bottleneck(self, p, x):
        y = self.conv_bn(p + ".conv1", p + ".bn1", x, 1, 0, True)
        y = self.conv_bn(p + ".conv2", p + ".bn2", y, 1, 1, True)
        res = x
        if self.has(p + ".downsample.0.weight"):
            res = self.conv_bn(p + ".downsample.0", p + ".downsample.1", x, 1, 0, False)
        return self.conv_bn(p + ".conv3", p + ".bn3", y, 1, 0, True, res)

    def branch_streams(self, like, n):
        """The launching stream plus n - 1 side streams (one per HRNet branch).  ``OTPOSE_TRAIN_STREAMS=0`` keeps the whole
        step on one stream."""
        main = torch.cuda.current_stream(like.device)
        if os.environ.get("OTPOSE_TRAIN_STREAMS", "1") == "0" or n < 2:
            return [main] * n
        return [main] + hip.side_streams(like.device, n - 1)

    def hr_module(self, p, xs, n_out):
        """One HighResolutionModule (model/HRNet.py:400-497).  The branches are independent until the fuse layers and each
        fuse row is independent of the others: branch i and fuse row i are enqueued on stream i (fork / join around both
        phases), so the low-resolution branches - a few hundred workgroups per launch, half a wave of the chip - run beside
        the high-resolution one instead of after it.  autograd replays every node on the stream its forward ran on, so the
        backward overlaps the same way.  Tensors that cross streams are registered with the caching allocator
        (``record_stream``) so their memory is not handed out again while the other stream still reads it."""
        n = len(xs)
        xs = list(xs)
        st = self.branch_streams(xs[0], n)
        main, par = st[0], n > 1 and st[-1] is not st[0]
        for i in range(n):
            if par and i:
                st[i].wait_stream(main)
                xs[i].record_stream(st[i])
            with torch.cuda.stream(st[i]):
                b = 0
                while self.has(f"{p}.branches.{i}.{b}.conv1.weight"):
                    xs[i] = self.basic_block(f"{p}.branches.{i}.{b}", xs[i])
                    b += 1
        if n == 1:
            return xs
        if par:
            for i in range(1, n):
                main.wait_stream(st[i])
            for i in range(1, n_out):
                st[i].wait_stream(main)
        outs = []
        for i in range(n_out):
            # y = relu(sum_j f_ij(x_j)) (HRNet.py:487-494).  The identity term seeds the sum and every other term rides
            # on a fused add (BatchNorm residual input / up-sample accumulate); the last one applies the ReLU.
            with torch.cuda.stream(st[i]):
                y = xs[i]
                terms = [j for j in range(n) if j != i]
                for idx, j in enumerate(terms):
                    last = idx == len(terms) - 1
                    q = f"{p}.fuse_layers.{i}.{j}"
                    if par:
                        xs[j].record_stream(st[i])
                    if j > i:
                        low = self.conv_bn(q + ".0", q + ".1", xs[j], 1, 0, False)
                        y = self.upsample_add(low, y, 2 ** (j - i), last)
                    else:
                        t = xs[j]
                        for k in range(i - j - 1):
                            t = self.conv_bn(f"{q}.{k}.0", f"{q}.{k}.1", t, 2, 1, True)
                        k = i - j - 1
                        y = self.conv_bn(f"{q}.{k}.0", f"{q}.{k}.1", t, 2, 1, relu=last, res=y)
            outs.append(y)
        if par:
            for i in range(1, n_out):
                main.wait_stream(st[i])
                outs[i].record_stream(main)
        return outs

    def upsample_add(self, low, y, f, relu):
        return UpsampleAddFunction.apply(low, y, f, relu)

    def hrnet_input(self, x):
        """(B, 15, H, W) clip -> the backbone's (5B, 3, H, W) batch (model/OTPose.py:317)."""
        return torch.cat(x.split(3, dim=1), 0).contiguous()

    def hrnet_output(self, p, y):
        return self.conv(p, y)

    def hrnet(self, p, x):
        m = self.cfg["MODEL"]
        stages = [m["EXTRA"][f"STAGE{s}"] for s in (2, 3, 4)]
        x = self.conv_bn(p + ".conv1", p + ".bn1", x, 2, 1, True)
        x = self.conv_bn(p + ".conv2", p + ".bn2", x, 2, 1, True)
        for b in range(4):
            x = self.bottleneck(f"{p}.layer1.{b}", x)
        ys = [x]
        for si, scfg in enumerate(stages):
            s = si + 2
            nb = scfg["NUM_BRANCHES"]
            xs = []
            for i in range(nb):
                t = f"{p}.transition{s - 1}.{i}"
                if self.has(t + ".0.weight"):
                    xs.append(self.conv_bn(t + ".0", t + ".1", ys[i], 1, 1, True))
                elif self.has(t + ".0.0.weight"):
                    z, k = ys[-1], 0
                    while self.has(f"{t}.{k}.0.weight"):
                        z = self.conv_bn(f"{t}.{k}.0", f"{t}.{k}.1", z, 2, 1, True)
                        k += 1
                    xs.append(z)
                else:
                    xs.append(ys[i])
            ys = xs
            nm = scfg["NUM_MODULES"]
            for mi in range(nm):
                n_out = 1 if (s == 4 and mi == nm - 1) else nb
                ys = self.hr_module(f"{p}.stage{s}.{mi}", ys, n_out)
                if self.taps is not None:
                    self.taps.update({f"hr:stage{s}.{mi}.out{i}": y for i, y in enumerate(ys)})
        return self.hrnet_output(p + ".final_layer", ys[0])

    # ---- ConvTransformer (model/blocks.py:264-280, 400-453; ConvVideoTransformer.py:123-184) ------------------------
    def layer_norm(self, p, x):
        return T.layer_norm(x, self.P[p + ".weight"], self.P[p + ".bias"], 1e-5)

    def mhca(self, p, x, ln1, n_head, stride):
        """MultiHeadConvAttention (model/blocks.py:359-453) of ``ln1(x)``.  The front - ln1, then per q / k / v a depthwise
        conv, a channel LayerNorm and the 1x1 projection - is one autograd node that keeps only ``x`` and rebuilds its seven
        (B, C, T) intermediates in the backward (:class:`train_ops.AttnFrontFunction`: 5 GB less at cfg2 for ~2 ms of
        recomputation per step).  ``OTPOSE_TRAIN_RECOMPUTE=0`` composes the per-layer nodes instead, keeping everything."""
        c = x.shape[1]
        hs = c // n_head
        names = ("query", "key", "value")
        if torch.is_grad_enabled() and os.environ.get("OTPOSE_TRAIN_RECOMPUTE", "1") != "0":
            branches = [(self.P[f"{p}.{n}_conv.weight"], self.P[f"{p}.{n}_norm.weight"], self.P[f"{p}.{n}_norm.bias"],
                         self._w4(f"{p}.{n}.weight"), self.P.get(f"{p}.{n}.bias")) for n in names]
            q, k, v = T.attn_front(x, stride, 1e-5, (self.P[ln1 + ".weight"], self.P[ln1 + ".bias"]), branches)
        else:
            xn = self.layer_norm(ln1, x)

            def branch(name):
                y = T.dwconv3(xn, self.P[f"{p}.{name}_conv.weight"], stride)
                y = self.layer_norm(f"{p}.{name}_norm", y)
                return self.conv1d(f"{p}.{name}", y)

            q, k, v = (branch(n) for n in names)
        out = T.chan_attn(q, k, v, n_head, 1.0 / math.sqrt(hs))          # attn_pdrop is 0 in OTPose.py:209-216
        return self.dropout(self.conv1d(p + ".proj", out), self.mods[p].proj_pdrop)

    def dropout(self, x, rate):
        if self.stochastic and rate > 0.0:
            return torch.nn.functional.dropout(x, rate, True)
        return x

    def drop_path_mask(self, x, rate):
        """blocks.py:303-316: one Bernoulli(keep) draw per sample, survivors scaled by 1/keep -> (B,) factors, or None."""
        if not (self.stochastic and rate > 0.0):
            return None
        keep = 1.0 - rate
        return (keep + torch.rand((x.shape[0],), dtype=x.dtype, device=x.device)).floor_().div_(keep)

    def tblock(self, p, x, n_head, stride):
        a = self.mhca(p + ".attn", x, p + ".ln1", n_head, stride)
        skip = x if stride == 1 else T.maxpool3s2(x)
        blk = self.mods[p]
        y = T.scale_residual(skip, a, self.P[p + ".drop_path_attn.scale"], self.drop_path_mask(a, blk.path_pdrop))
        h = self.mlp(p + ".mlp", self.layer_norm(p + ".ln2", y), blk.proj_pdrop)
        return T.scale_residual(y, h, self.P[p + ".drop_path_mlp.scale"], self.drop_path_mask(h, blk.path_pdrop))

    def mlp(self, p, yn, pdrop):
        """Conv1d(C, 4C, 1) -> GELU -> Dropout -> Conv1d(4C, C, 1) -> Dropout (model/blocks.py:248-254)."""
        h = self.conv1d(p + ".0", yn)
        h = self.conv1d(p + ".3", self.dropout(T.gelu(h), pdrop))
        return self.dropout(h, pdrop)

    def conv_transformer(self, p, x4, n_head, arch):
        b, c, h, w = x4.shape
        t = h * w
        x = x4.reshape(b, c, t) + self.Bf[p + ".pos_embd"][:, :, :t]
        for i in range(arch[1]):
            x = self.tblock(f"{p}.stem.{i}", x, n_head, 1)
        outs = [x]
        for i in range(arch[2]):
            x = self.tblock(f"{p}.branch.{i}", x, n_head, 2)
            outs.append(T.upsample_linear(x, 2 ** (i + 1)))
        return outs

    # ---- RSB heads (model/RSB.py:77-103) ---------------------------------------------------------------------------
    def cbr(self, p, x, pad, relu=True, res=None):
        y = self.bn(p + ".bn", self.conv(p + ".conv", x, 1, pad), res, relu)
        if self.taps is not None:
            self.taps["cbr:" + p] = y
        return y

    def rsb_block(self, p, x):
        bc = self.P[p + ".conv_bn_relu2_1_1.conv.weight"].shape[0]
        s = torch.split(self.cbr(p + ".conv_bn_relu1", x, 0), bc, 1)
        c = lambda name, t: self.cbr(f"{p}.conv_bn_relu2_{name}", t, 1)      # noqa: E731
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
        res = x
        if self.has(p + ".downsample.conv.weight"):
            res = self.cbr(p + ".downsample", x, 0, False)
        return self.cbr(p + ".conv_bn_relu3", torch.cat((o11, o22, o33, o44), 1), 0, True, res)

    def rsb_chain(self, p, x):
        i = 0
        while self.has(f"{p}.layers.{i}.conv_bn_relu1.conv.weight"):
            x = self.rsb_block(f"{p}.layers.{i}", x)
            i += 1
        return x

    # ---- dilated offset / mask convs in front of the DCNs (model/OTPose.py:168-177, 381-383) ---------------------------
    def offset_mask_input(self, trans):
        return trans

    def offset_mask_conv(self, trans, weight, d):
        return T.conv2d(trans, weight, None, 1, d, d)

    # ---- whole forward (model/OTPose.py:307-394) ---------------------------------------------------------------
    def forward(self, x, margin):
        m = self.cfg["MODEL"]
        J = m["NUM_JOINTS"]
        pe_w, pe_h = m["HEATMAP_SIZE"]
        dils = list(m["DEFORMABLE_CONV"]["DILATION"])
        B = x.shape[0]
        rough = self.hrnet("rough_pose_estimation_net", self.hrnet_input(x))
        frames = rough.split(B, dim=0)                         # cur, prev_1, next_1, prev_2, next_2 (, prev_3, next_3)
        R = (len(frames) - 1) // 2
        cur = frames[0]
        total_b = cur
        for f_ in frames[1:]:
            total_b = total_b + f_
        squeezed = total_b.sum(1, keepdim=True).expand(-1, J, -1, -1).contiguous()
        inter = total_b * squeezed
        # three independent paths follow (as in the inference engine): def_fuse needs only ``total_b``, and the two temporal
        # encoders (with their final layers) only share their inputs - def_fuse and encoder 1 go to side streams
        st = self.branch_streams(total_b, 3)
        main, par = st[0], st[1] is not st[0]
        if par:
            st[2].wait_stream(main)                    # def_fuse's stream starts from here (its launches are enqueued further down)
            total_b.record_stream(st[2])
        ctx = self.conv_transformer("flow_encoder", total_b, 1, (0, 6, 0))[0].reshape(B, J, pe_h, pe_w)
        mg = margin.to(x.dtype)
        div = lambda t, k: t / (mg[:, k] + 1)[:, None, None, None]           # noqa: E731
        prev = [div(frames[1 + 2 * r], 2 * r) for r in range(R)]
        nxt = [div(frames[2 + 2 * r], 2 * r + 1) for r in range(R)]

        def side(ts):
            acc = ts[0]
            for t in ts[1:]:
                acc = acc + t
            return cur + acc

        prev_b, next_b = side(prev), side(nxt)
        sym = [cur + (nxt[r] + prev[r]) for r in range(R)]                   # close_b, far_b (, wide_b)
        b1, b2 = [prev_b] + sym[::-1], [next_b] + sym
        if R == 3:                                                           # 7-frame extension (oracle window_maps)
            b1.append(side(prev[1:]))
            b2.append(side(nxt[1:]))
        x1 = torch.stack([inter, ctx] + b1 + [t * squeezed for t in b1], 2).flatten(1, 2)
        x2 = torch.stack([inter, ctx] + b2 + [t * squeezed for t in b2], 2).flatten(1, 2)
        fk = self.P["final_layer1.weight"].shape[-1]
        if par:
            st[1].wait_stream(main)
            x1.record_stream(st[1])
        with torch.cuda.stream(st[1]):
            t1 = self.conv_transformer("temporal_encoder1", x1, 2, (0, 6, 2))
            s1 = torch.stack(t1, 1).contiguous().view(B, -1, pe_h, pe_w)
            f1 = self.conv("final_layer1", s1, 1, fk // 2)
        # def_fuse is enqueued BETWEEN the two encoders: autograd runs a backward's ready nodes latest-first, so its backward (a chain
        # of small fp32 launches, ~7 ms alone at a third of the chip) comes between encoder 2's and encoder 1's and overlaps with
        # encoder 1's on the other stream - emitted first it came last, alone, in front of the backbone's backward
        with torch.cuda.stream(st[2]):
            def_h = self.rsb_chain("def_fuse", total_b)
        t2 = self.conv_transformer("temporal_encoder2", x2, 2, (0, 6, 2))
        s2 = torch.stack(t2, 1).contiguous().view(B, -1, pe_h, pe_w)
        f2 = self.conv("final_layer2", s2, 1, fk // 2)
        if par:
            main.wait_stream(st[1])
            main.wait_stream(st[2])
            for t in (f1, def_h) + tuple(t1):
                t.record_stream(main)
        trans = self.rsb_chain("offset_mask_combine_conv", torch.cat([f1, f2, def_h], 1))
        out = None
        if self.taps is not None:
            self.taps.update(x1=x1, x2=x2, t1_0=t1[0], t2_0=t2[0], f1=f1, f2=f2, def_h=def_h, trans=trans)
        trans_in = self.offset_mask_input(trans)
        for i, d in enumerate(dils):
            off = self.offset_mask_conv(trans_in, self.P[f"offsets_list.{i}.0.weight"], d)
            msk = self.offset_mask_conv(trans_in, self.P[f"masks_list.{i}.0.weight"], d)
            p = f"modulated_deform_conv_list.{i}.deform_conv"
            wrp = ops.modulated_deform_conv(def_h, off, msk, self.P[p + ".weight"], self.P[p + ".bias"], 1, d, d, 1, J)
            out = (1.0 / len(dils)) * wrp if out is None else out + (1.0 / len(dils)) * wrp
            if self.taps is not None:
                self.taps.update({f"off{i}": off, f"msk{i}": msk, f"wrp{i}": wrp})
        if self._nbt:
            torch._foreach_add_(self._nbt, 1)
            self._nbt = []
        return out, rough, inter, prev_b, ctx, squeezed, total_b


class _LayerTap(Function):
    """Identity node that records the gradient flowing back through THIS use of a tensor (``taps['layers']`` mode of
    :class:`TrainGraphBF16`: the per-layer, in-context check of tests/test_gpu_train_bf16_yardstick.py)."""

    @staticmethod
    def forward(ctx, x, rec, key):
        ctx.rec, ctx.key = rec, key
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        ctx.rec[ctx.key] = g.detach().clone()
        return g, None, None


class TrainGraphBF16(TrainGraph):
    """The same walk with the HRNet backbone (86 % of the FLOPs, model/HRNet.py:116-152) on the bf16 path of
    :mod:`otpose_amd.bf16_ops`: NHWC bfloat16 activations, bf16 matrix-core convolutions with fp32 accumulation, fp32
    BatchNorm statistics, fp32 master weights and weight gradients (BASELINE configs[2]).  The frames enter through one
    layout / precision conversion that also performs the ``cat(split(3, 1), 0)`` of model/OTPose.py:317, and the final
    1x1 layer hands fp32 NCHW heat-maps to the rest of the graph."""

    def forward(self, x, margin):
        """One launch refreshes every packed bf16 operator from the fp32 master weights (:class:`bf16_ops.PackCache`, kept on
        the model; ``OTPOSE_PACK_BATCH=0`` packs at every use instead), then the walk of :meth:`TrainGraph.forward`."""
        if os.environ.get("OTPOSE_PACK_BATCH", "1") == "0":
            return super().forward(x, margin)
        cache = self.model.__dict__.get("_otp_pack_cache")
        if cache is None:
            cache = self.model.__dict__["_otp_pack_cache"] = B16.PackCache()
        cache.repack(x.device)
        prev = B16.set_active_packs(cache)
        try:
            return super().forward(x, margin)
        finally:
            B16.set_active_packs(prev)

    def conv_bn(self, conv, bn, x, stride=1, pad=0, relu=False, res=None):
        if x.dtype != B16.BF16:
            return super().conv_bn(conv, bn, x, stride, pad, relu, res)
        self.count_batch(bn)
        layers = self.taps.get("layers") if self.taps is not None else None
        if layers is not None:
            # record this layer in context: its bf16 operands, its result, and (after backward) the gradient arriving at the
            # result and the gradients it hands to its input / residual
            rec = dict(conv=conv, bn=bn, stride=stride, pad=pad, relu=relu, x=x.detach(), res=None if res is None else res.detach())
            x = _LayerTap.apply(x, rec, "gx")
            if res is not None:
                res = _LayerTap.apply(res, rec, "gres")
        y = B16.conv_bn(x, self.P[conv + ".weight"], self.P[bn + ".weight"], self.P[bn + ".bias"], res,
                        self.Bf[bn + ".running_mean"], self.Bf[bn + ".running_var"], stride, pad, relu, 0.1, 1e-5)
        if layers is not None:
            rec["y"] = y.detach()
            y = _LayerTap.apply(y, rec, "gy")
            layers.append(rec)
        return y

    def basic_block(self, p, x):
        if (x.dtype != B16.BF16 or self.has(p + ".downsample.0.weight") or os.environ.get("OTPOSE_BLOCK_FUSE", "1") == "0"
                or (self.taps is not None and self.taps.get("layers") is not None)):     # layer taps: two conv + BN nodes
            return super().basic_block(p, x)
        self.count_batch(p + ".bn1")
        self.count_batch(p + ".bn2")
        P, Bf = self.P, self.Bf
        return B16.basic_block(x, P[p + ".conv1.weight"], P[p + ".bn1.weight"], P[p + ".bn1.bias"],
                               Bf[p + ".bn1.running_mean"], Bf[p + ".bn1.running_var"],
                               P[p + ".conv2.weight"], P[p + ".bn2.weight"], P[p + ".bn2.bias"],
                               Bf[p + ".bn2.running_mean"], Bf[p + ".bn2.running_var"], 0.1, 1e-5)

    def upsample_add(self, low, y, f, relu):
        if low.dtype != B16.BF16:
            return super().upsample_add(low, y, f, relu)
        return B16.upsample_add(low, y, f, relu)

    def hrnet_input(self, x):
        return B16.to_nhwc(x, frame_split=x.shape[0])

    def offset_mask_input(self, trans):
        """The 32-channel feature map feeds ten dilated 3x3 convs (32 -> 306 / 153): one conversion to NHWC bf16, every
        conv on the bf16 matrix cores, offsets / masks handed to the DCN as fp32 NCHW."""
        if os.environ.get("OTPOSE_BF16_OFFSETS", "1") == "0":
            return trans
        return B16.to_nhwc_grad(trans)

    def offset_mask_conv(self, trans, weight, d):
        if trans.dtype != B16.BF16:
            return super().offset_mask_conv(trans, weight, d)
        return B16.conv_out(trans, weight, None, 1, d, d)

    def mlp(self, p, yn, pdrop):
        """The MLP interior in bf16: the (B, C, T) fp32 LayerNorm output enters as a (B, 1, T, CS) NHWC bf16 view, the 4C-wide
        hidden activation (the largest tensor of a block: 240 MB at cfg2 in fp32) exists only in bf16, both projections
        and their gradients run on the bf16 matrix cores, and the down-projection hands back fp32 (B, C, T)."""
        if yn.shape[1] % 8 or yn.shape[2] % 32 or os.environ.get("OTPOSE_BF16_MLP", "1") == "0":
            return super().mlp(p, yn, pdrop)                  # the C = 17 flow encoder / odd lengths stay on the fp32 path
        x = B16.to_nhwc_grad(yn.unsqueeze(2))
        w1, w2 = self._w4(p + ".0.weight"), self._w4(p + ".3.weight")
        rate = pdrop if self.stochastic else 0.0
        if os.environ.get("OTPOSE_MLP_FUSED_TRAIN", "0") == "1" and B16.mlp_interior_supported(x, w1, w2):
            # GELU / dropout and their derivatives inside the projections' launches (bf16_ops.MlpInteriorFunction): built, tested and
            # measured NEUTRAL (123.4 against 122.7 ms per step on one box) - the epilogues' ~20 vector instructions per hidden element
            # cost the pointwise kernel what the two element-wise passes cost the HBM - so the three-node chain stays the default
            o = B16.mlp_interior(x, w1, self.P.get(p + ".0.bias"), w2, self.P.get(p + ".3.bias"), rate)
        else:
            h = B16.conv_bias(x, w1, self.P.get(p + ".0.bias"))
            h = B16.gelu_dropout(h, rate)                                    # GELU + Dropout(pdrop) as one pass each way
            o = B16.conv_out(h, w2, self.P.get(p + ".3.bias"))
        return self.dropout(o.squeeze(2), pdrop)

    def hrnet_output(self, p, y):
        return B16.conv_out(y, self.P[p + ".weight"], self.P.get(p + ".bias"))


def forward_train(model, x, margin, taps=None):
    if not x.is_cuda:
        raise RuntimeError("OTPose training forward expects CUDA (HIP) tensors; there is no CPU path")
    dtype = getattr(model, "train_dtype", "f32")
    if dtype not in ("f32", "bf16"):
        raise ValueError(f"OTPose.train_dtype must be 'f32' or 'bf16', got {dtype!r}")
    graph = (TrainGraphBF16 if dtype == "bf16" else TrainGraph)(model)
    graph.taps = taps
    return graph.forward(x, margin)


def criterion(outputs, target, target_weight, flags=None):
    """The two ST_OHKW terms of the reference training loop (script/Common.py:122-130)."""
    B = target.shape[0]
    pred_t = outputs[1].split(B)[0]
    loss = st_ohkw_loss(outputs[0], pred_t, target, target_weight, flags)
    occlusion = (target + outputs[2]) / 2
    return loss + st_ohkw_loss(outputs[4], outputs[4], occlusion, target_weight, flags)
