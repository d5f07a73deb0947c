#!/usr/bin/env python
"""Which host call sites of the bf16 training step launch the small PyTorch kernels (device-to-device copies, elementwise
adds / casts): one profiled step, kernels grouped by (kernel family, innermost otpose_amd frame of the launching op).
Development tool: `python tools/train_torch_profile.py [--dtype bf16] [--match copy,elementwise]`."""
from __future__ import annotations

import argparse
import collections
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from otpose_amd import OTPose, cfg2                    # noqa: E402
from otpose_amd import synthetic as S                  # noqa: E402
from otpose_amd import train as TR                     # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--match", default="Memcpy,copyBuffer,elementwise,CatArray,fill,reduce_kernel")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    cfg = cfg2()
    model = OTPose(cfg)
    S.fill_synthetic_(model)
    model = model.to(dev).train()
    model.train_dtype = a.dtype
    x, margin = S.synthetic_clip(a.batch, cfg.MODEL.IMAGE_SIZE)
    x, margin = x.to(dev), margin.to(dev)
    J = cfg.MODEL.NUM_JOINTS
    w, h = cfg.MODEL.HEATMAP_SIZE
    g = torch.rand(a.batch, J, h, w, device=dev) * 0.2
    wt = (torch.rand(a.batch, J, 1, device=dev) > 0.15).float()
    from otpose_amd.optim import FusedAdamW
    opt = FusedAdamW([p for p in model.parameters() if p.requires_grad], lr=1e-4, weight_decay=0.01, max_grad_norm=1.0)

    def step():
        outs = model(x, margin=margin)
        loss = TR.criterion(outs, g, wt)
        opt.zero_grad()
        loss.backward()
        opt.step()

    for _ in range(2):
        step()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        step()
        torch.cuda.synchronize()
    keys = [k for k in a.match.split(",") if k]
    events = prof.events()
    by_site = collections.defaultdict(lambda: [0, 0.0])
    # kernels hang below the CPU op that launched them: walk the CPU events, attribute their device time to the innermost
    # frame inside this repository
    for ev in events:
        if ev.device_type != torch.autograd.DeviceType.CPU or not ev.kernels:
            continue
        for k in ev.kernels:
            if not any(m in k.name for m in keys):
                continue
            site = "?"
            for fr in ev.stack or []:
                if "otpose_amd" in fr or "tools/" in fr:
                    site = fr.split("/")[-1]
                    break
            fam = k.name.split("<")[0][:60]
            e = by_site[(fam, ev.name, site)]
            e[0] += 1
            e[1] += k.duration
    rows = sorted(by_site.items(), key=lambda kv: -kv[1][1])
    print("%8s %6s  %-50s %-28s %s" % ("us", "calls", "kernel", "aten op", "call site"))
    for (fam, op, site), (n, us) in rows[:70]:
        print("%8.0f %6d  %-50s %-28s %s" % (us, n, fam[:50], op[:28], site))


if __name__ == "__main__":
    main()
