#!/usr/bin/env python
"""The "reference single-GPU PyTorch forward" denominator of BASELINE.json's 5x target: the
restated eager graph (oracle/otpose_oracle.py: stock PyTorch-ROCm ops = MIOpen / rocBLAS, DCN as gather-based torch
ops) at batch 16 x 5 x 384x288, fp32, HIP-event timed.  Measurement tool only - never part of the product path."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import otpose_oracle as O          # noqa: E402
from otpose_amd import OTPose, cfg2            # noqa: E402
from otpose_amd import synthetic as S          # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
dev = torch.device("cuda", 0)
TRAIN = "--train" in sys.argv
torch.backends.cudnn.benchmark = not TRAIN      # MIOpen's exhaustive search over every backward conv takes > 20 min
cfg = cfg2()
m = OTPose(cfg)
S.fill_synthetic_(m)
sd = {k: v.detach().to(dev) for k, v in m.state_dict().items()}
x, margin = S.synthetic_clip(B, cfg.MODEL.IMAGE_SIZE)
x, margin = x.to(dev), margin.to(dev)
with torch.no_grad():
    for _ in range(0 if TRAIN else 3):
        outs = O.otpose_forward(sd, cfg, x, margin)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    times = []
    for _ in range(1 if TRAIN else 10):
        e0.record()
        outs = O.otpose_forward(sd, cfg, x, margin)
        e1.record()
        e1.synchronize()
        times.append(e0.elapsed_time(e1))
times.sort()
med = times[len(times) // 2]
print("eager PyTorch-ROCm forward, batch %d: median %.1f ms -> %.1f frames/s (min %.1f ms)" %
      (B, med, 5 * B / med * 1e3, times[0]))
result = {"batch": B, "forward_ms": med, "forward_min_ms": times[0], "frames_per_s": 5 * B / med * 1e3,
          "what": "oracle/otpose_oracle.py graph on stock PyTorch-ROCm ops (MIOpen / rocBLAS, DCN as gather ops), fp32, "
                  "HIP events, median of %d" % len(times), "device": torch.cuda.get_device_name(0)}

if TRAIN:
    # training step of the same eager graph (BatchNorm batch statistics, two ST_OHKW terms, backward, clip, AdamW)
    import time
    names = set(dict(m.named_parameters()))
    leaves = {k: v.clone().requires_grad_() for k, v in sd.items() if k in names}
    sdt = dict(sd)
    sdt.update(leaves)
    J = cfg.MODEL.NUM_JOINTS
    w, h = cfg.MODEL.HEATMAP_SIZE
    g = torch.rand(B, J, h, w, device=dev) * 0.2
    g[:, ::2, 3, 4] = 1.0
    wt = (torch.rand(B, J, 1, device=dev) > 0.15).float()
    opt = torch.optim.AdamW(list(leaves.values()), lr=1e-4, weight_decay=0.01)
    for it in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        outs = O.otpose_forward(sdt, cfg, x, margin, training_bn=True)
        loss = (O.st_ohkw_mse_loss(outs[0], outs[1][:B], g, wt)["final_loss"]
                + O.st_ohkw_mse_loss(outs[4], outs[4], (g + outs[2]) / 2, wt)["final_loss"])
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        opt.zero_grad(set_to_none=True)
        loss.backward()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        torch.nn.utils.clip_grad_norm_(list(leaves.values()), 1.0)
        opt.step()
        torch.cuda.synchronize()
        t3 = time.perf_counter()
        print("eager PyTorch-ROCm train step %d, batch %d: forward+loss %.1f ms  backward %.1f ms  clip+AdamW %.1f ms  "
              "total %.1f ms (%.1f frames/s)  peak mem %.1f GB" %
              (it, B, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t3 - t0) * 1e3, 5 * B / (t3 - t0),
               torch.cuda.max_memory_allocated() / 2**30), flush=True)
        result["train_step_%d" % it] = {"forward_loss_ms": (t1 - t0) * 1e3, "backward_ms": (t2 - t1) * 1e3,
                                         "optimizer_ms": (t3 - t2) * 1e3, "total_ms": (t3 - t0) * 1e3,
                                         "peak_mem_GB": torch.cuda.max_memory_allocated() / 2**30}

for a in sys.argv:
    if a.startswith("--json="):
        import json
        with open(a[7:], "w") as f:
            json.dump(result, f, indent=1)
