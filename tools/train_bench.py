#!/usr/bin/env python
"""Training-step timing of the HIP autograd path (cfg3 of SURVEY.md section 8d, in fp32): forward (BatchNorm batch
statistics), the two ST_OHKW terms, backward through every HIP kernel, global-norm clip and AdamW.  Development tool
(run on the GPU box): `python tools/train_bench.py --batch 16 --steps 3`."""
from __future__ import annotations

import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from otpose_amd import OTPose, cfg2                    # noqa: E402
from otpose_amd import synthetic as S                  # noqa: E402
from otpose_amd import train as TR                     # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--no-dropout", action="store_true")
    ap.add_argument("--free-run", type=int, default=5, help="steps timed back to back without the per-phase synchronisation")
    ap.add_argument("--dtype", default="f32", choices=["f32", "bf16"], help="activation dtype of the backbone")
    ap.add_argument("--torch-optim", action="store_true", help="torch.optim.AdamW + clip_grad_norm_ instead of FusedAdamW")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    cfg = cfg2()
    model = OTPose(cfg)
    S.fill_synthetic_(model)
    model = model.to(dev).train()
    model.train_dropout = not a.no_dropout
    model.train_dtype = a.dtype
    x, margin = S.synthetic_clip(a.batch, cfg.MODEL.IMAGE_SIZE)
    x, margin = x.to(dev), margin.to(dev)
    J = cfg.MODEL.NUM_JOINTS
    w, h = cfg.MODEL.HEATMAP_SIZE
    g = torch.rand(a.batch, J, h, w, device=dev) * 0.2
    g[:, ::2, 3, 4] = 1.0
    wt = (torch.rand(a.batch, J, 1, device=dev) > 0.15).float()
    params = [p for p in model.parameters() if p.requires_grad]
    if a.torch_optim:
        opt = torch.optim.AdamW(params, lr=1e-4, weight_decay=0.01)
    else:
        from otpose_amd.optim import FusedAdamW
        opt = FusedAdamW(params, lr=1e-4, weight_decay=0.01, max_grad_norm=1.0)

    def sync():
        torch.cuda.synchronize()
        return time.perf_counter()


    for it in range(a.steps + 1):
        torch.cuda.reset_peak_memory_stats()
        t0 = sync()
        outs = model(x, margin=margin)
        c1 = time.perf_counter()                # host time to enqueue the forward (before the device catches up)
        t1 = sync()
        loss = TR.criterion(outs, g, wt)
        t2 = sync()
        opt.zero_grad()
        loss.backward()
        c3 = time.perf_counter()
        t3 = sync()
        if a.torch_optim:
            torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)
        opt.step()
        t4 = sync()
        print("step %d%s: forward %.1f ms (host %.1f)  loss %.1f ms  backward %.1f ms (host %.1f)  clip+AdamW %.1f ms  total %.1f ms  "
              "(%.1f frames/s)  loss %.5f  peak mem %.1f GB" %
              (it, " (warm-up)" if it == 0 else "", (t1 - t0) * 1e3, (c1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3,
               (c3 - t2) * 1e3, (t4 - t3) * 1e3,
               (t4 - t0) * 1e3, a.batch * 5 / (t4 - t0), float(loss), torch.cuda.max_memory_allocated() / 2**30),
              flush=True)
    if a.free_run > 0:
        # what a training loop sees: no synchronisation between the phases, the host runs ahead of the device
        t0 = sync()
        for _ in range(a.free_run):
            loss = TR.criterion(model(x, margin=margin), g, wt)
            opt.zero_grad()
            loss.backward()
            if a.torch_optim:
                torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)
            opt.step()
        t1 = sync()
        print("free-running: %.1f ms per step over %d steps (%.1f frames/s), peak mem %.1f GB" %
              ((t1 - t0) * 1e3 / a.free_run, a.free_run, a.batch * 5 * a.free_run / (t1 - t0),
               torch.cuda.max_memory_allocated() / 2**30), flush=True)


if __name__ == "__main__":
    main()
