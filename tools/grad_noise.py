"""Rounding floor of the training-step gradients (tiny W8 model): the oracle in float32 on the CPU, the oracle in float32
through PyTorch-ROCm eager on the GPU and the HIP autograd path, each against the oracle's float64 gradients.
Development tool; calibrates the thresholds of tests/test_gpu_train_e2e.py."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import otpose_oracle as O                   # noqa: E402
from otpose_amd import OTPose, tiny_cfg                 # noqa: E402
from otpose_amd import synthetic as S                   # noqa: E402
from tests.test_gpu_train_e2e import _targets           # noqa: E402

torch.set_num_threads(min(16, len(os.sched_getaffinity(0))))
cfg = tiny_cfg(8, (64, 96))
model = OTPose(cfg)
S.fill_synthetic_(model)
sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
names = set(dict(model.named_parameters()))
SEED = int(sys.argv[1]) if len(sys.argv) > 1 else None
x, margin = S.synthetic_clip(2, cfg.MODEL.IMAGE_SIZE) if SEED is None else S.synthetic_clip(2, cfg.MODEL.IMAGE_SIZE, SEED)


OUTS = ("output", "rough", "intersection", "prev_b", "context", "squeezed", "total_b")


_CBR = O._cbr
_REC = {}


def _cbr_tap(sd_, p, x_, pad, relu=True, training=False):
    y = _CBR(sd_, p, x_, pad, relu, training)
    if relu:                                   # conv_bn_relu3 / downsample are fused with the residual on the HIP side
        _REC["cbr:" + p] = y
    return y


O._cbr = _cbr_tap


def run(dt, dev="cpu"):
    _REC.clear()
    leaves = {k: v.detach().clone().to(dev, dt).requires_grad_() for k, v in sd.items()
              if v.is_floating_point() and k in names}
    s = {k: (v.to(dev, dt) if v.is_floating_point() else v.to(dev)) for k, v in sd.items()}
    s.update(leaves)
    outs, mid = O.otpose_forward(s, cfg, x.to(dev, dt), margin.to(dev), training_bn=True, return_intermediates=True)
    taps = dict(x1=mid["x1"], x2=mid["x2"], t1_0=mid["t1"][0], t2_0=mid["t2"][0], f1=mid["f1"], f2=mid["f2"],
                def_h=mid["def_h"], trans=mid["trans"])
    for i, (off, msk, wrp) in enumerate(mid["dcn"]):
        taps.update({f"off{i}": off, f"msk{i}": msk, f"wrp{i}": wrp})
    taps.update(_REC)
    for t in taps.values():
        t.retain_grad()
    B, J, h, w = outs[0].shape
    g, wt = _targets(B, J, h, w)
    g, wt = g.to(dev, dt), wt.to(dev, dt)
    loss = (O.st_ohkw_mse_loss(outs[0], outs[1][:B], g, wt)["final_loss"]
            + O.st_ohkw_mse_loss(outs[4], outs[4], (g + outs[2]) / 2, wt)["final_loss"])
    for o in outs:
        o.retain_grad()
    loss.backward()
    gr = {k: v.grad.double().cpu() for k, v in leaves.items() if v.grad is not None}
    gr.update({"=" + n: t.detach().double().cpu() for n, t in taps.items()})
    gr.update({"@" + n: o.grad.double().cpu() for n, o in zip(OUTS, outs) if o.grad is not None})
    gr.update({"@" + n: t.grad.double().cpu() for n, t in taps.items() if t.grad is not None})
    return float(loss.detach()), gr


def run_hip():
    from otpose_amd import train as TR
    m = OTPose(cfg)
    m.load_state_dict(sd)
    m = m.cuda().train()
    m.train_dropout = False
    taps = {}
    outs = TR.forward_train(m, x.cuda(), margin.cuda(), taps)
    for t in taps.values():
        t.retain_grad()
    B, J, h, w = outs[0].shape
    g, wt = _targets(B, J, h, w)
    loss = TR.criterion(outs, g.cuda(), wt.cuda())
    for o in outs:
        o.retain_grad()
    loss.backward()
    gr = {k: p.grad.double().cpu() for k, p in m.named_parameters() if p.grad is not None}
    gr.update({"=" + n: t.detach().double().cpu() for n, t in taps.items()})
    gr.update({"@" + n: o.grad.double().cpu() for n, o in zip(OUTS, outs) if o.grad is not None})
    gr.update({"@" + n: t.grad.double().cpu() for n, t in taps.items() if t.grad is not None})
    return float(loss.detach()), gr


def errs(ga, g64):
    d = {}
    for k, r in g64.items():
        nr = float(r.norm())
        if nr > 1e-6 and k in ga:
            d[k] = float((ga[k] - r).norm()) / nr
    return d


l64, g64 = run(torch.float64)
cols = {"cpu32": run(torch.float32)}
if torch.cuda.is_available():
    cols["eager32"] = run(torch.float32, "cuda")
    cols["hip"] = run_hip()
print("loss fp64 %.8f  " % l64 + "  ".join("%s %.8f" % (n, v[0]) for n, v in cols.items()))
E = {n: errs(v[1], g64) for n, v in cols.items()}
for n, d in E.items():
    v = sorted(e for k, e in d.items() if k[0] not in "@=")
    print("%-8s median %.3e  p90 %.3e  worst %.3e  (%d tensors)" % (n, v[len(v) // 2], v[len(v) * 9 // 10], v[-1], len(v)))
last = list(E)[-1]
worst = sorted((kv for kv in E[last].items() if kv[0][0] not in "@="), key=lambda kv: -kv[1])[:12]
print("worst tensors of %s:" % last)
for k, e in worst:
    print("  %-70s " % k + "  ".join("%s %.3e" % (n, E[n].get(k, float("nan"))) for n in E))
print("stage boundaries (gradient w.r.t. the forward outputs):")
for k in sorted(k for k in E[last] if k[0] in "@="):
    print("  %-70s " % k + "  ".join("%s %.3e" % (n, E[n].get(k, float("nan"))) for n in E))
