"""Development aid (GPU box): the convolutions the eval engine emits for cfg2 at batch 16 (OTPOSE_CONV_LOG=1), grouped by shape."""
import collections
import io
import os
import sys
import contextlib
os.environ["OTPOSE_CONV_LOG"] = "1"
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from otpose_amd import OTPose, cfg2                 # noqa: E402
from otpose_amd import synthetic as S               # noqa: E402

cfg = cfg2()
m = OTPose(cfg)
S.fill_synthetic_(m)
m = m.cuda().eval()
x, margin = S.synthetic_clip(16, cfg.MODEL.IMAGE_SIZE)
buf = io.StringIO()
with contextlib.redirect_stderr(buf), torch.no_grad():
    m(x.cuda(), margin=margin.cuda())
c = collections.Counter(l.strip() for l in buf.getvalue().splitlines() if l.startswith("conv "))
for k, v in sorted(c.items(), key=lambda kv: -kv[1]):
    print("%3d x %s" % (v, k))
