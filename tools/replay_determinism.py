"""Development aid (GPU box): is the batch-16 cfg2 forward bit-identical across hipGraph replays?  Prints, per replay, which
outputs differ from the first one and by how much.  usage: python tools/replay_determinism.py [replays=8]"""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from otpose_amd import OTPose, cfg2                 # noqa: E402
from otpose_amd import synthetic as S               # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
cfg = cfg2()
x, margin = S.synthetic_clip(16, cfg.MODEL.IMAGE_SIZE)
m = OTPose(cfg)
S.fill_synthetic_(m)
m = m.cuda().eval()
names = ("output", "rough", "intersection", "prev_b", "context", "squeezed", "total_b")
bad = 0
with torch.no_grad():
    first = [o.clone() for o in m(x.cuda(), margin=margin.cuda())]
    for rep in range(n):
        again = m(x.cuda(), margin=margin.cuda())
        diffs = [(nm, float((a - b).abs().max()), int((a != b).sum())) for nm, a, b in zip(names, first, again) if not torch.equal(a, b)]
        if diffs:
            bad += 1
            print("replay %d:" % rep, "  ".join("%s max|d| %.2e (%d elements)" % d for d in diffs))
print("%d of %d replays differ from the first" % (bad, n))
