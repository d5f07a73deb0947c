"""BASELINE configs[4] extension: forward throughput of the 7-frame window (batch 16 x 7 x 384x288, HRNet-W48, fp32 storage,
split-bf16 products) next to the 5-frame headline configuration."""
import os
import sys
import time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from otpose_amd import OTPose, cfg2                # noqa: E402
from otpose_amd.config import cfg5               # noqa: E402
from otpose_amd import synthetic as S            # noqa: E402

for name, cfg, frames in (("cfg2 (5 frames)", cfg2(), 5), ("cfg5 (7 frames)", cfg5(), 7)):
    model = OTPose(cfg)
    S.fill_synthetic_(model)
    model = model.cuda().eval()
    x, margin = S.synthetic_clip(16, cfg.MODEL.IMAGE_SIZE, frames=frames)
    x, margin = x.cuda(), margin.cuda()
    with torch.no_grad():
        for _ in range(3):
            outs = model(x, margin=margin)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            outs = model(x, margin=margin)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 10
    print("%s: %.2f ms per forward, %.0f frames/s, outputs finite: %s, peak memory %.1f GB" %
          (name, dt * 1e3, 16 * frames / dt, all(bool(torch.isfinite(o).all()) for o in outs),
           torch.cuda.max_memory_allocated() / 2 ** 30))
    del model, outs
    torch.cuda.empty_cache()
