"""Development tool: the 7 forward outputs of the Winograd-routed engine vs the direct-kernel engine on the same (shifted, scaled)
inputs at cfg2 size - a check that the Winograd transforms stay at fp32 rounding level end to end."""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from otpose_amd import OTPose, cfg2
from otpose_amd import synthetic as S
from otpose_amd.engine import InferenceEngine
cfg = cfg2()
m = OTPose(cfg); S.fill_synthetic_(m); m = m.cuda().eval()
x, margin = S.synthetic_clip(4, cfg.MODEL.IMAGE_SIZE, seed=99)
x = x.cuda() * 2.0 + 0.5; margin = margin.cuda()
outs = {}
for key, env in (("wino", "1"), ("direct", "0")):
    os.environ["OTPOSE_WINOGRAD"] = env
    e = InferenceEngine(m, 4, x.device)
    with torch.no_grad():
        outs[key] = [o.clone().double() for o in e.run(x, margin)]
names = ("output", "rough", "intersection", "prev_b", "context", "squeezed", "total_b")
for n, a, b in zip(names, outs["wino"], outs["direct"]):
    print("%-13s max|.| %.3f  max abs diff wino-direct %.3e" % (n, float(b.abs().max()), float((a - b).abs().max())))
