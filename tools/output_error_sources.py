"""Development tool (GPU box): which family of split-product kernels carries the heat-map error of the default forward?
Clips 0-3 of the headline batch (clip 3 is the worst of the 16) against the fp32 oracle, with one family at a time moved to its
exact-fp32 form through the engine's switches.  usage: python tools/output_error_sources.py"""
import os
import sys

import torch

sys.path.insert(0, '.')
from oracle import otpose_oracle as O                      # noqa: E402
from otpose_amd import OTPose, cfg1, cfg2, hip                   # noqa: E402
from otpose_amd import synthetic as S                      # noqa: E402

NAMES = ("output", "rough", "intersection", "prev_b", "context", "squeezed", "total_b")
import argparse                                           # noqa: E402
ap = argparse.ArgumentParser()
ap.add_argument("--cfg", default="cfg2")
ap.add_argument("--ws", type=int, default=S.WEIGHT_SEED)
args = ap.parse_args()
cfg = cfg2() if args.cfg == "cfg2" else cfg1()
x, margin = S.synthetic_clip(16, cfg.MODEL.IMAGE_SIZE)
x, margin = x[:4], margin[:4]
m = OTPose(cfg)
S.fill_synthetic_(m, args.ws, S.gains_for(cfg))
sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
m = m.cuda().eval()
torch.set_num_threads(min(16, len(os.sched_getaffinity(0))))
with torch.no_grad():
    ref = O.otpose_forward(sd, cfg, x, margin)

CONFIGS = [("default (all split products)", {}),
           ("attention products exact", {"ATTN": "0"}),
           ("encoder MLP / projections exact", {"OTPOSE_FUSED_MLP": "0", "OTPOSE_DENSE_CC": "0", "OTPOSE_QKV_FRONT": "0"}),
           ("encoders exact (attention + MLP + projections)", {"ATTN": "0", "OTPOSE_FUSED_MLP": "0", "OTPOSE_DENSE_CC": "0", "OTPOSE_QKV_FRONT": "0"}),
           ("warping head unfused (exact offset / mask convs)", {"OTPOSE_DCN_FUSED": "0"}),
           ("last HR module + final layer exact", {"OTPOSE_F32_TAIL": "1"}),
           ("everything exact", {"OTPOSE_CONV_MATH": "f32", "ATTN": "0"})]
keys = sorted({k for _, e in CONFIGS for k in e})
for name, env in CONFIGS:
    for k in keys:
        os.environ.pop(k, None)
    for k, v in env.items():
        if k != "ATTN":
            os.environ[k] = v
    hip.lib().otp_chan_attn_set_split(0 if env.get("ATTN") == "0" else 1)
    m.invalidate_engine()
    with torch.no_grad():
        outs = [o.cpu() for o in m(x.cuda(), margin=margin.cuda())]
    errs = {n: float((o - r).abs().max()) for n, o, r in zip(NAMES, outs, ref)}
    per_clip = [float((outs[0][k] - ref[0][k]).abs().max()) for k in range(4)]
    print("%-52s output %.2e (clips %s)  rough %.2e  total_b %.2e  context %.2e"
          % (name, errs["output"], " ".join("%.1e" % v for v in per_clip), errs["rough"], errs["total_b"], errs["context"]), flush=True)
