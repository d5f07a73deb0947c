#!/usr/bin/env python
"""Fold rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (gpurun_out/traffic_*) into profiles/r01_traffic.json.
FETCH_SIZE / WRITE_SIZE are KiB per dispatch; on gfx950 FETCH_SIZE reads half the bytes of a 16-byte-per-lane stream
(MI355X_MICROARCH.md, HBM section), so the conv kernel's fetch is doubled; the DCN's dword streams are uncalibrated."""
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def mean_counter(path, kernel_substr, counter):
    vals = []
    if not os.path.isfile(path):
        return None
    with open(path) as f:
        for r in csv.DictReader(f):
            if kernel_substr in r["Kernel_Name"] and r["Counter_Name"] == counter:
                vals.append(float(r["Counter_Value"]))
    return sum(vals) / len(vals) if vals else None


def main():
    base = os.path.join(ROOT, "gpurun_out")
    out = {}
    try:
        with open(os.path.join(ROOT, "profiles", "r01_traffic.json")) as fi:
            prev = json.load(fi)
    except (OSError, ValueError):
        prev = {}
    for key, sub, kern, fetch_corr in (("conv_48_48_3x3_96x72_x80", "conv", "conv_win_kernel", 2.0),
                                       ("conv_wino_48_48_3x3_96x72_x80", "wino", "conv_wino_kernel", 1.0),   # 16 B window + 4 B weight streams: raw = lower bound
                                       ("mdcn_fwd_17x96x72_x16", "dcn", "mdcn_fwd_kernel", 1.0),
                                       ("ln_mlp_fused_136_544_T6912_x16", "mlp", "mlp_fused_kernel", 1.0)):   # 8 B streams: raw
        f = mean_counter(os.path.join(base, "traffic_%s_fetch" % sub, "p_counter_collection.csv"), kern, "FETCH_SIZE")
        w = mean_counter(os.path.join(base, "traffic_%s_write" % sub, "p_counter_collection.csv"), kern, "WRITE_SIZE")
        if f is None or w is None:
            if key in prev:
                out[key] = prev[key]                 # passes not re-run this time: keep the committed measurement
            continue
        out[key] = {"FETCH_SIZE_KiB": f, "WRITE_SIZE_KiB": w, "fetch_correction": fetch_corr,
                    "hbm_bytes_per_launch": (f * fetch_corr + w) * 1024.0}
    with open(os.path.join(ROOT, "profiles", "r01_traffic.json"), "w") as fo:
        json.dump(out, fo, indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    sys.exit(main())
