"""Development tool (GPU box): gradients of named intermediates (TrainGraph.taps) in two backward passes of the same bf16 graph -
where does a deviating backward first differ?  usage: python tools/train_nondet_taps.py [rounds=14]"""
import sys
import torch
sys.path.insert(0, '.')
from otpose_amd import synthetic as S                      # noqa: E402
from otpose_amd import train as TR                         # noqa: E402
from tests.test_gpu_train_slots import _pair, _targets, LR, WD, CLIP   # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 14
cfg, a, b = _pair("bf16")
x, margin = S.synthetic_clip(2, cfg.MODEL.IMAGE_SIZE)
x, margin = x.cuda(), margin.cuda()
J, (w, h) = cfg.MODEL.NUM_JOINTS, cfg.MODEL.HEATMAP_SIZE
opt = torch.optim.AdamW([p for p in b.parameters() if p.requires_grad], lr=LR, weight_decay=WD)
NAMES = ("output", "rough", "inter", "prev_b", "context", "squeezed", "total")


def run():
    for p in b.parameters():
        p.grad = None
    taps = {}
    outs = TR.forward_train(b, x, margin, taps)
    keep = {}
    for k, t in list(taps.items()) + [("out:" + n, o) for n, o in zip(NAMES, outs)]:
        if torch.is_tensor(t) and t.requires_grad and t.is_floating_point():
            t.retain_grad()
            keep[k] = t
    loss = TR.criterion(outs, g, wt)
    loss.backward()
    torch.cuda.synchronize()
    tg = {k: (t.grad.detach().float().clone() if t.grad is not None else None) for k, t in keep.items()}
    tv = {k: t.detach().float().clone() for k, t in keep.items()}
    pg = {n: (p.grad.detach().clone() if p.grad is not None else torch.zeros_like(p)) for n, p in b.named_parameters()}
    return float(loss), tg, tv, pg


for it in range(rounds):
    g, wt = _targets(2, J, h, w, seed=11 + 5 * it)
    r = [run() for _ in range(3)]
    for j in (1, 2):
        num = sum(float(((r[0][3][n].double() - r[j][3][n].double()) ** 2).sum()) for n in r[0][3])
        den = sum(float((r[0][3][n].double() ** 2).sum()) for n in r[0][3])
        rel = (num / den) ** 0.5
        print("round %d run %d vs 0: whole-gradient rel L2 %.3e" % (it, j, rel))
        if rel > 1e-6:
            for k in r[0][1]:
                ga, gb = r[0][1][k], r[j][1][k]
                va, vb = r[0][2][k], r[j][2][k]
                dv = float((va - vb).abs().max())
                if ga is None or gb is None:
                    print("   tap %-40s no grad" % k[:40])
                    continue
                d = float((ga - gb).norm()) / max(float(ga.norm()), 1e-30)
                if k.startswith("hr:") or k.startswith("out:") or d > 1e-5:
                    print("   tap %-40s forward max|d| %.2e   grad rel %.3e  (|g| %.3e)" % (k[:40], dv, d, float(ga.norm())))
    for n, p in b.named_parameters():
        p.grad = r[0][3][n]
    torch.nn.utils.clip_grad_norm_([p for p in b.parameters() if p.requires_grad], CLIP)
    opt.step()
