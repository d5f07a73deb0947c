"""Development tool (GPU box): one headline forward with OTPOSE_CONV_LOG=1 - one stderr line per convolution the engine emits
(shape and the kernel family the generic emitter would pick).  usage: OTPOSE_CONV_LOG=1 python tools/conv_routes.py 2>&1 | sort | uniq -c"""
import sys, torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from otpose_amd import OTPose, cfg2
from otpose_amd import synthetic as S
cfg = cfg2()
m = OTPose(cfg); S.fill_synthetic_(m); m = m.cuda().eval()
x, margin = S.synthetic_clip(16, cfg.MODEL.IMAGE_SIZE)
with torch.no_grad():
    m(x.cuda(), margin=margin.cuda())
