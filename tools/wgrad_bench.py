#!/usr/bin/env python
"""Per-shape throughput of otp_conv2d_wgrad (and the dgrad launch) on the conv shapes of the cfg2/cfg3 training step.
Development tool (run on the GPU box)."""
from __future__ import annotations

import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from otpose_amd import train_ops as T                 # noqa: E402
from tools.conv_bench import SHAPES, PEAK, time_ms     # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--only", default="")
    ap.add_argument("--dgrad", action="store_true")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(3)
    tot = 0.0
    print("%-34s %9s %9s %7s %9s" % ("shape", "ms/call", "TFLOP/s", "frac", "ms/step"))
    for name, n, cin, cout, k, s, d, h, w, calls in SHAPES:
        if a.only and a.only not in name:
            continue
        pad = d * (k // 2)
        ho = (h + 2 * pad - (d * (k - 1) + 1)) // s + 1
        wo = (w + 2 * pad - (d * (k - 1) + 1)) // s + 1
        x = torch.randn(n, cin, h, w, generator=g).to(dev)
        go = torch.randn(n, cout, ho, wo, generator=g).to(dev)
        wt = (torch.randn(cout, cin, k, k, generator=g) * 0.05).to(dev)
        if a.dgrad:
            fn = lambda: T.conv2d_grad_input(go, wt, x.shape, s, pad, d)      # noqa: E731
        else:
            fn = lambda: T.conv2d_grad_weight(x, go, wt.shape, s, pad, d)     # noqa: E731
        ms = time_ms(fn, a.iters)
        flop = 2.0 * cin * cout * k * k * ho * wo * n
        tf = flop / (ms * 1e-3) / 1e12
        tot += ms * calls
        print("%-34s %9.4f %9.2f %7.3f %9.3f" % (name, ms, tf, tf * 1e12 / PEAK, ms * calls))
    print("sum over listed shapes: %.2f ms per step" % tot)


if __name__ == "__main__":
    main()
