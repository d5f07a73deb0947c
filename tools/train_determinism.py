"""Development tool (GPU box): is the training step bit-reproducible?  A replica is stepped with FusedAdamW (gradient slots,
side streams: the default scheduling); at every step the forward + backward runs TWICE from the same weights and the loss
and every parameter gradient are compared bit for bit; a second replica stepped beside it must hold the same weights at the
end.  Prints per step the number of differing tensors and the whole-gradient relative L2 of the difference, and a summary
line `K / N steps bit-identical`.
usage: python tools/train_determinism.py [bf16|f32] [steps=50] [tiny|cfg2]"""
import sys

import torch

sys.path.insert(0, '.')
from otpose_amd import OTPose, cfg2                        # noqa: E402
from otpose_amd import synthetic as S                      # noqa: E402
from otpose_amd.optim import FusedAdamW                    # noqa: E402
from tests.test_gpu_train_slots import CLIP, LR, WD, _loss, _pair, _targets   # noqa: E402

dtype = sys.argv[1] if len(sys.argv) > 1 else "bf16"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
size = sys.argv[3] if len(sys.argv) > 3 else "tiny"
B = 2
if size == "cfg2":
    import copy
    cfg = cfg2()
    a = OTPose(cfg)
    S.fill_synthetic_(a)
    b = copy.deepcopy(a)
    for m in (a, b):
        m.cuda().train()
        m.train_dropout = False
        m.train_dtype = dtype
else:
    cfg, a, b = _pair(dtype)
x, margin = S.synthetic_clip(B, cfg.MODEL.IMAGE_SIZE)
x, margin = x.cuda(), margin.cuda()
J, (w, h) = cfg.MODEL.NUM_JOINTS, cfg.MODEL.HEATMAP_SIZE
opts = [FusedAdamW([p for p in m.parameters() if p.requires_grad], lr=LR, weight_decay=WD, max_grad_norm=CLIP) for m in (a, b)]


def run(model, opt, g, wt):
    opt.zero_grad()
    loss = _loss(model, x, margin, g, wt)
    loss.backward()
    torch.cuda.synchronize()
    opt.flat_grads()
    return loss.detach().clone(), {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}


def diff(g0, g1):
    bad = [n for n in g0 if not torch.equal(g0[n], g1[n])]
    num = sum(float(((g0[n].double() - g1[n].double()) ** 2).sum()) for n in bad)
    den = sum(float((g0[n].double() ** 2).sum()) for n in g0)
    return bad, (num / max(den, 1e-300)) ** 0.5


same = 0
for it in range(steps):
    g, wt = _targets(B, J, h, w, seed=11 + 5 * it)
    l0, g0 = run(a, opts[0], g, wt)
    l1, g1 = run(a, opts[0], g, wt)
    lb, gb = run(b, opts[1], g, wt)
    bad1, r1 = diff(g0, g1)
    badb, rb = diff(g0, gb)
    ok = not bad1 and not badb and torch.equal(l0, l1) and torch.equal(l0, lb)
    same += int(ok)
    print("step %2d loss %.9g  second pass: %d / %d tensors differ (rel L2 %.2e)  replica: %d differ (rel L2 %.2e)%s"
          % (it, float(l0), len(bad1), len(g0), r1, len(badb), rb, "" if ok else "   <-- " + (bad1 + badb)[0]), flush=True)
    for o in opts:
        o.step()
pb = dict(b.named_parameters())
wbad = [n for n, p in a.named_parameters() if not torch.equal(p.detach(), pb[n].detach())]
print("%s %s: %d / %d steps bit-identical (loss and every gradient, second pass and replica); %d weights differ between the "
      "replicas after %d steps" % (dtype, size, same, steps, len(wbad), steps))
