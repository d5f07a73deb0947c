#!/bin/bash
# the round-4 profile set (GPU box, from the repo root): PMC passes + HBM traffic of the dominant conv launch (both forms) and of
# the new stride-2 S8 conv, the kernel trace / timeline / gaps of the bench command, the phase stamps of the dominant kernel
# -> gpurun_out/${TAG}_*  (copied to profiles/ by hand)
set -u
TAG=${TAG:-r04}
mkdir -p gpurun_out
SQ1="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES"
SQ2="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES"
bash tools/pmc.sh ${TAG}_convs "$SQ1" "$SQ2" "FETCH_SIZE" "WRITE_SIZE" -- tools/convs_one.py 80 48 48 96 72
bash tools/pmc.sh ${TAG}_convs_conv2 "FETCH_SIZE" "WRITE_SIZE" -- tools/convs_one.py 80 48 48 96 72 conv2s
bash tools/pmc.sh ${TAG}_convs2 "$SQ1" "$SQ2" "FETCH_SIZE" "WRITE_SIZE" -- tools/convs2_one.py 80 48 96 96 72
python tools/pmc_fold.py convs_kernel gpurun_out/${TAG}_convs_pmc1.csv gpurun_out/${TAG}_convs_pmc2.csv gpurun_out/${TAG}_convs_pmc3.csv gpurun_out/${TAG}_convs_pmc4.csv > gpurun_out/${TAG}_convs_pmc_fold.txt
python tools/pmc_fold.py convs_kernel gpurun_out/${TAG}_convs_conv2_pmc1.csv gpurun_out/${TAG}_convs_conv2_pmc2.csv >> gpurun_out/${TAG}_convs_pmc_fold.txt
python tools/pmc_fold.py convs2_kernel gpurun_out/${TAG}_convs2_pmc1.csv gpurun_out/${TAG}_convs2_pmc2.csv gpurun_out/${TAG}_convs2_pmc3.csv gpurun_out/${TAG}_convs2_pmc4.csv > gpurun_out/${TAG}_convs2_pmc_fold.txt
bash tools/convs_timing.sh 80 48 48 96 72 > gpurun_out/${TAG}_convs_phase_stamps.txt 2>&1
PROF_TIMELINE="glue_total" bash tools/prof.sh ${TAG}_bench bench.py --steps 10 --warmup 3 --no-exact-fp32 --no-train-step --no-cpu-baseline --no-config5 --no-eager-baseline
MIN_GAP=10 bash tools/trace_gaps.sh glue_total bench.py --steps 6 --warmup 3 --no-exact-fp32 --no-train-step --no-cpu-baseline --no-config5 --no-eager-baseline > gpurun_out/${TAG}_bench_gaps.txt 2>&1
TAG=$TAG python3 - <<'PY'
import csv, json, os
tag = os.environ["TAG"]
def mean(path, sub, counter):
    v = [float(r["Counter_Value"]) for r in csv.DictReader(open(path)) if sub in r["Kernel_Name"] and r["Counter_Name"] == counter]
    return sum(v) / len(v) if v else None
out = {}
for key, stem, kern, alg, what in (
        ("convs_48_48_3x3_96x72_x80", "convs", "convs_kernel", 212336640, "convs_kernel<3, false, 4> S8 -> S8, ReLU (a BasicBlock's conv1; tools/convs_one.py 80 48 48 96 72)"),
        ("convs_48_48_3x3_96x72_x80_conv2", "convs_conv2", "convs_kernel", 318504960, "convs_kernel<3, false, 4> S8 + S8 residual -> S8 (a BasicBlock's conv2 since round 4; tools/convs_one.py ... conv2s)"),
        ("convs2_48_96_3x3s2_96x72_x80", "convs2", "convs2_kernel", 4 * 80 * (48 * 96 * 72 + 96 * 48 * 36), "convs2_kernel<3, true, 2> S8 -> fp32 NCHW, ReLU (fuse layer 48 -> 96 stride 2; tools/convs2_one.py 80 48 96 96 72)")):
    fi = 3 if stem != "convs_conv2" else 1
    f = mean("gpurun_out/%s_%s_pmc%d.csv" % (tag, stem, fi), kern, "FETCH_SIZE")
    w = mean("gpurun_out/%s_%s_pmc%d.csv" % (tag, stem, fi + 1), kern, "WRITE_SIZE")
    if f is None or w is None:
        continue
    out[key] = {"kernel": what, "FETCH_SIZE_KiB": f, "WRITE_SIZE_KiB": w, "fetch_correction": 2.0,
                "hbm_bytes_per_launch": (2.0 * f + w) * 1024, "algorithmic_bytes_per_launch": alg,
                "note": "separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (profiles/%s_%s_pmc*.csv); all fetch streams are 16 B per "
                        "lane (LDS-DMA pieces, float4 residual): FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950" % (tag, stem)}
json.dump(out, open("gpurun_out/%s_traffic.json" % tag, "w"), indent=1)
print(json.dumps({k: v["hbm_bytes_per_launch"] for k, v in out.items()}))
PY
