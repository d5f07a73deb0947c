#!/bin/bash
# the round-3 profile set (GPU box, from the repo root): PMC passes + traffic of the dominant conv launch, its phase stamps, the
# kernel trace of the bench command and the bench line itself -> gpurun_out/${TAG}_*  (copied to profiles/ by hand)
set -u
TAG=${TAG:-r03c}
mkdir -p gpurun_out
if [ "${1:-all}" = "all" ]; then
  bash tools/pmc.sh ${TAG}_convs "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES" "FETCH_SIZE" "WRITE_SIZE" -- tools/convs_one.py 80 48 48 96 72
  bash tools/pmc.sh ${TAG}_convs_conv2 "FETCH_SIZE" "WRITE_SIZE" -- tools/convs_one.py 80 48 48 96 72 conv2
  python tools/pmc_fold.py convs_kernel gpurun_out/${TAG}_convs_pmc1.csv gpurun_out/${TAG}_convs_pmc2.csv gpurun_out/${TAG}_convs_pmc3.csv gpurun_out/${TAG}_convs_pmc4.csv > gpurun_out/${TAG}_convs_pmc_fold.txt
  python tools/pmc_fold.py convs_kernel gpurun_out/${TAG}_convs_conv2_pmc1.csv gpurun_out/${TAG}_convs_conv2_pmc2.csv >> gpurun_out/${TAG}_convs_pmc_fold.txt
  bash tools/convs_timing.sh 80 48 48 96 72 > gpurun_out/${TAG}_convs_phase_stamps.txt 2>&1
fi
PROF_TIMELINE="glue_total" bash tools/prof.sh ${TAG}_bench bench.py --steps 10 --warmup 3 --no-exact-fp32 --no-train-step --no-cpu-baseline --no-config5
MIN_GAP=10 bash tools/trace_gaps.sh glue_total bench.py --steps 6 --warmup 3 --no-exact-fp32 --no-train-step --no-cpu-baseline --no-config5 > gpurun_out/${TAG}_bench_gaps.txt 2>&1
python bench.py > gpurun_out/${TAG}_bench_line.json 2> gpurun_out/${TAG}_bench.err
TAG=$TAG python - <<'PY'
import json, os
d = json.loads(open("gpurun_out/%s_bench_line.json" % os.environ["TAG"]).read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["ms_per_launch"], d["roofline"]["frac"], d["vs_eager_rocm"]["speedup"],
      d["train_step"]["ms_per_step"], d["config5"]["ms_per_step"], d["exact_fp32_kernels"]["ms_per_step"], d["roofline_dcn"]["frac"])
PY
