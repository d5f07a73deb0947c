"""Development tool (GPU box): which parameter gradients differ between two backward passes of the SAME bf16 training graph on
the same weights and inputs?  Steps a replica with FusedAdamW between rounds (the deviation needs fresh weights to show).
usage: python tools/train_nondet_probe.py [rounds=12] [streams=1]"""
import os
import sys
import torch
sys.path.insert(0, '.')
if len(sys.argv) > 2 and sys.argv[2] == "0":
    os.environ["OTPOSE_TRAIN_STREAMS"] = "0"
from otpose_amd import synthetic as S                      # noqa: E402
from otpose_amd.optim import FusedAdamW                    # noqa: E402
from tests.test_gpu_train_slots import _pair, _targets, _loss, LR, WD, CLIP   # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 12
cfg, a, b = _pair("bf16")
x, margin = S.synthetic_clip(2, cfg.MODEL.IMAGE_SIZE)
x, margin = x.cuda(), margin.cuda()
J, (w, h) = cfg.MODEL.NUM_JOINTS, cfg.MODEL.HEATMAP_SIZE
opt = torch.optim.AdamW([p for p in b.parameters() if p.requires_grad], lr=LR, weight_decay=WD)
names = [n for n, p in b.named_parameters() if p.requires_grad]


def grads():
    for p in b.parameters():
        p.grad = None
    loss = _loss(b, x, margin, g, wt)
    loss.backward()
    torch.cuda.synchronize()
    return float(loss), {n: (p.grad.detach().clone() if p.grad is not None else torch.zeros_like(p)) for n, p in b.named_parameters()}


for it in range(rounds):
    g, wt = _targets(2, J, h, w, seed=11 + 5 * it)
    runs = [grads() for _ in range(3)]
    l0, g0 = runs[0]
    for r, (l1, g1) in enumerate(runs[1:], 1):
        num = sum(float(((g0[n].double() - g1[n].double()) ** 2).sum()) for n in names)
        den = sum(float((g0[n].double() ** 2).sum()) for n in names)
        rel = (num / den) ** 0.5
        print("round %d run %d vs 0: loss %.9g / %.9g  whole-gradient rel L2 %.3e" % (it, r, l1, l0, rel))
        if rel > 1e-6:
            rows = []
            for i, n in enumerate(names):
                d = float((g0[n].double() - g1[n].double()).norm())
                rn = float(g0[n].double().norm())
                if d > 1e-7 * max(rn, 1e-12):
                    rows.append((i, n, d / max(rn, 1e-30), d))
            print("   %d of %d tensors differ; by absolute difference:" % (len(rows), len(names)))
            for i, n, rl, d in sorted(rows, key=lambda t: -t[3])[:12]:
                print("     #%3d %-64s rel %.2e abs %.3e" % (i, n, rl, d))
            print("   first (closest to the input) / last (closest to the loss) differing: %s / %s" % (rows[0][1], rows[-1][1]))
    for p in b.parameters():
        p.grad = None
    for n, p in b.named_parameters():
        p.grad = runs[0][1][n]
    torch.nn.utils.clip_grad_norm_([p for p in b.parameters() if p.requires_grad], CLIP)
    opt.step()
