"""Optimizer step of the reference training loop on flat HBM buffers: ``torch.nn.utils.clip_grad_norm_`` followed by
``torch.optim.AdamW.step()`` (script/Common.py:138-143; groups built by thirdparty/utils/train_utils.py:62-133) as two
HIP passes - a sum of squares over the flat gradients and one fused clip + AdamW update per hyper-parameter group - with no
host synchronisation (the clip coefficient is read from device memory).

``FusedAdamW`` takes the same parameter groups as ``torch.optim.AdamW`` (so ``make_optimizer``'s three groups carry over
unchanged) and re-homes parameters and gradients into one flat fp32 buffer per group: ``p.data`` and ``p.grad`` become views,
autograd keeps accumulating into them, and ``flat_grads()`` hands the few large buffers to the RCCL all-reduce
(:mod:`otpose_amd.parallel`) - large messages are what the point-to-point xGMI links want.
"""
from __future__ import annotations

import torch

from . import hip


def _bump_versions(tensors):
    """The kernels write parameters through raw pointers; tell autograd / the inference engine's staleness check."""
    try:
        torch._C._increment_version(tensors)
    except TypeError:                                   # older signature: one tensor at a time
        for t in tensors:
            torch._C._increment_version(t)


class FusedAdamW(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, max_grad_norm=0.0):
        if lr < 0 or eps < 0 or weight_decay < 0 or not (0 <= betas[0] < 1 and 0 <= betas[1] < 1):
            raise ValueError("invalid AdamW hyper-parameters")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self.max_grad_norm = float(max_grad_norm)
        self._flat = []                                  # per group: dict(p, g, m, v, params, step)
        # [epoch, clean]: zero_grad() opens an epoch in which every slot is known to be zero (train_ops.grad_slot hands a
        # slot out once per epoch); step() closes it.  One cell shared by all parameters of this optimizer.
        self._slot_epoch = [0, False]
        for group in self.param_groups:
            ps = [p for p in group["params"] if p.requires_grad]
            if not ps:
                self._flat.append(None)
                continue
            dev = ps[0].device
            if dev.type != "cuda" or any(p.device != dev or p.dtype != torch.float32 for p in ps):
                raise RuntimeError("FusedAdamW needs float32 parameters on one CUDA (HIP) device; there is no CPU path")
            total = sum(p.numel() for p in ps)
            fp = torch.empty(total, dtype=torch.float32, device=dev)
            fg, fm, fv = torch.zeros_like(fp), torch.zeros_like(fp), torch.zeros_like(fp)
            off = 0
            for p in ps:
                n = p.numel()
                fp[off:off + n].copy_(p.data.reshape(-1))
                p.data = fp[off:off + n].view_as(p)
                if p.grad is not None:
                    fg[off:off + n].copy_(p.grad.reshape(-1))
                p.grad = fg[off:off + n].view_as(p)
                # destination of this parameter's gradient inside the flat buffer: the backward kernels of otpose_amd write
                # there directly (grad_slot() below), so autograd neither allocates nor accumulates per-parameter tensors
                p._otp_grad_slot = p.grad
                p._otp_slot_epoch = self._slot_epoch
                self.state[p] = {"step": 0, "exp_avg": fm[off:off + n].view_as(p), "exp_avg_sq": fv[off:off + n].view_as(p)}
                off += n
            self._flat.append({"p": fp, "g": fg, "m": fm, "v": fv, "params": ps, "step": 0,
                               "gptr": [p._otp_grad_slot.data_ptr() for p in ps]})
        devs = {f["p"].device for f in self._flat if f}
        # [0]: the squared norm; behind it the scratch of otp_grad_sumsq's fixed-order reduction
        scratch = int(hip.lib().otp_grad_sumsq_scratch())
        self._normsq = {d: torch.zeros(1 + scratch, dtype=torch.float64, device=d) for d in devs}

    def _rehome_grads(self):
        """``p.grad`` must stay a view of the flat gradient buffer.  ``model.zero_grad()`` (set_to_none=True by default)
        or any ``p.grad = ...`` detaches it, after which autograd accumulates outside the buffer and step() /
        allreduce_flat_grads() would see zeros: copy such a gradient back and re-point the view.  A parameter whose
        gradient is None keeps a zero slot (its weight decay / moment update then match torch.optim.AdamW only if the
        caller really meant "zero gradient"; use this optimizer's own zero_grad() to keep the views)."""
        from .bf16_ops import join_wgrad_streams
        join_wgrad_streams()                  # weight gradients enqueued on side streams (bf16_ops.conv_wgrad)
        for f in self._flat:
            if not f:
                continue
            # fast path (every step): one pointer comparison per parameter, no tensor objects created
            for p, ptr in zip(f["params"], f["gptr"]):
                g = p.grad
                if g is not None and g.data_ptr() == ptr:
                    continue
                slot = p._otp_grad_slot
                if g is None:
                    slot.zero_()
                else:
                    slot.copy_(g.reshape(slot.shape))
                p.grad = slot.view(slot.shape)

    def flat_grads(self):
        """The flat gradient buffers (one per group) - the units to all-reduce."""
        self._rehome_grads()
        return [f["g"] for f in self._flat if f]

    def zero_grad(self, set_to_none: bool = False):
        """One memset per group.  ``p.grad`` is dropped so that the next backward hands each parameter its gradient
        exactly once: the otpose_amd backward kernels write straight into the parameter's slot of the flat buffer and return
        that view (autograd adopts it - no allocation, no ``grad += new`` launch per parameter, ~2000 tiny kernels per step
        at W48); gradients that arrive as ordinary tensors are copied into their slot by step() / flat_grads()."""
        for f in self._flat:
            if f:
                f["g"].zero_()
                for p in f["params"]:
                    p.grad = None
        self._slot_epoch[0] += 1
        self._slot_epoch[1] = True

    @torch.no_grad()
    def grad_norm(self, _rehomed=False):
        """Global L2 norm of all gradients as a device scalar (what clip_grad_norm_ returns), no host sync."""
        L = hip.lib()
        if not _rehomed:
            self._rehome_grads()
        for acc in self._normsq.values():
            acc[:1].zero_()
        for f in self._flat:
            if f:
                hip.check(L.otp_grad_sumsq(hip.ptr(f["g"]), f["g"].numel(), hip.ptr(self._normsq[f["g"].device]),
                                           hip.stream_of(f["g"])), "otp_grad_sumsq")
        accs = list(self._normsq.values())
        return accs[0][:1].sqrt() if len(accs) == 1 else torch.stack([a[:1].cpu() for a in accs]).sum().sqrt()

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        L = hip.lib()
        self._rehome_grads()
        self._slot_epoch[1] = False                      # the slots now hold this step's gradients: not clean until zero_grad()
        clip = self.max_grad_norm > 0.0
        if clip:
            self.grad_norm(_rehomed=True)
        for group, f in zip(self.param_groups, self._flat):
            if not f:
                continue
            f["step"] += 1
            b1, b2 = group["betas"]
            acc = self._normsq[f["g"].device]
            hip.check(L.otp_adamw_step(hip.ptr(f["p"]), hip.ptr(f["g"]), hip.ptr(f["m"]), hip.ptr(f["v"]), f["p"].numel(),
                                       float(group["lr"]), float(b1), float(b2), float(group["eps"]),
                                       float(group["weight_decay"]), f["step"], hip.ptr(acc) if clip else None,
                                       self.max_grad_norm, hip.stream_of(f["p"])), "otp_adamw_step")
            _bump_versions(f["params"])
        return loss

    def _sync_state_steps(self):
        """``state[p]["step"]`` of every parameter from its group's counter (kept per group on the hot path)."""
        for f in self._flat:
            if f:
                for p in f["params"]:
                    self.state[p]["step"] = f["step"]

    def state_dict(self):
        self._sync_state_steps()
        return super().state_dict()

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        # the base class re-creates the state tensors: copy them back into the flat moment buffers and re-point the views
        for f in self._flat:
            if not f:
                continue
            off = 0
            for p in f["params"]:
                n = p.numel()
                st = self.state[p]
                for key, flat in (("exp_avg", f["m"]), ("exp_avg_sq", f["v"])):
                    flat[off:off + n].copy_(st[key].reshape(-1))
                    st[key] = flat[off:off + n].view_as(p)
                f["step"] = int(st.get("step", f["step"]))
                st["step"] = f["step"]
                off += n
