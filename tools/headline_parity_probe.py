"""Development tool (GPU box): per-clip, per-output max |delta| of the batch-16 cfg2 forward against the CPU oracle, for the
default split-product engine and the exact-fp32 engine, and - the yardstick for ill-conditioned spots of the graph - the
fp32 oracle against the fp64 oracle on the same clips.  usage: python tools/headline_parity_probe.py [clips=16]"""
import os
import sys

import torch

sys.path.insert(0, '.')
from oracle import otpose_oracle as O                      # noqa: E402
from otpose_amd import OTPose, cfg2, hip                   # noqa: E402
from otpose_amd import synthetic as S                      # noqa: E402

NAMES = ("output", "rough", "intersection", "prev_b", "context", "squeezed", "total_b")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
cfg = cfg2()
x, margin = S.synthetic_clip(16, cfg.MODEL.IMAGE_SIZE)
margin[5] = torch.tensor([0.0, 1.0, 0.0, 2.0])
margin[11] = torch.tensor([1.0, 0.0, 2.0, 0.0])
x, margin = x[:B], margin[:B]
m = OTPose(cfg)
S.fill_synthetic_(m)
sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
m = m.cuda().eval()


def run_engine():
    with torch.no_grad():
        return [o.cpu() for o in m(x.cuda(), margin=margin.cuda())]


x3 = run_engine()
os.environ["OTPOSE_CONV_MATH"] = "f32"
hip.lib().otp_chan_attn_set_split(0)
m.invalidate_engine()
f32 = run_engine()


def clip_rows(t, k, n):
    return t[k:k + 1] if t.shape[0] == n else t[k::n]


torch.set_num_threads(min(16, len(os.sched_getaffinity(0))))
print("clip  output: " + "  ".join("%-23s" % n for n in NAMES))
worst = {}
for lo in range(0, B, 4):
    hi = min(B, lo + 4)
    with torch.no_grad():
        r32 = O.otpose_forward(sd, cfg, x[lo:hi], margin[lo:hi])
        r64 = O.otpose_forward(sd64, cfg, x[lo:hi].double(), margin[lo:hi].double())
    for k in range(hi - lo):
        cells = []
        for i, n in enumerate(NAMES):
            ref = clip_rows(r32[i], k, hi - lo)
            e3 = float((clip_rows(x3[i], lo + k, B) - ref).abs().max())
            ef = float((clip_rows(f32[i], lo + k, B) - ref).abs().max())
            eo = float((ref.double() - clip_rows(r64[i], k, hi - lo)).abs().max())
            e3_64 = float((clip_rows(x3[i], lo + k, B).double() - clip_rows(r64[i], k, hi - lo)).abs().max())
            cells.append("%.1e/%.1e/%.1e/%.1e" % (e3, ef, eo, e3_64))
            w = worst.setdefault(n, [0.0, 0.0, 0.0, 0.0, 0.0])
            w[0], w[1], w[2], w[3] = max(w[0], e3), max(w[1], ef), max(w[2], eo), max(w[3], e3_64)
            w[4] = max(w[4], float(ref.abs().max()))
        print("%4d  x3/f32/o32-vs-o64/x3-vs-o64: " % (lo + k) + "  ".join(cells), flush=True)
print("worst over clips (x3 vs o32, f32-engine vs o32, o32 vs o64, x3 vs o64, max |ref|):")
for n in NAMES:
    print("  %-13s %.2e  %.2e  %.2e  %.2e  %.3g" % ((n,) + tuple(worst[n])))
