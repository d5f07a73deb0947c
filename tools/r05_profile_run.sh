#!/bin/bash
# the round-5 end-of-round profile set (GPU box, from the repo root) -> gpurun_out/r05_* (copied to profiles/ by hand):
# PMC folds (+ HBM traffic passes) of the encoder kernels and of the fused warping head at the FINAL build,
# kernel trace / timeline of the bench command and of the bf16 training step.
set -u
TAG=${TAG:-r05}
mkdir -p gpurun_out
SQ1="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES"
SQ2="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES"
SQ3="SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"
bash tools/pmc.sh ${TAG}_enc "$SQ1" "$SQ2" "$SQ3" "FETCH_SIZE" "WRITE_SIZE" -- tools/encoder_kernels_probe.py
for k in mlpx_kernel qkvx_front_kernel densex_cc_kernel; do
  python tools/pmc_fold.py $k gpurun_out/${TAG}_enc_pmc1.csv gpurun_out/${TAG}_enc_pmc2.csv gpurun_out/${TAG}_enc_pmc3.csv gpurun_out/${TAG}_enc_pmc4.csv gpurun_out/${TAG}_enc_pmc5.csv
done > gpurun_out/${TAG}_enc_pmc_fold.txt
bash tools/pmc.sh ${TAG}_dcnf "$SQ1" "$SQ2" "$SQ3" "FETCH_SIZE" "WRITE_SIZE" -- tools/dcn_fused_bench.py
python tools/pmc_fold.py dcn_fused_kernel gpurun_out/${TAG}_dcnf_pmc1.csv gpurun_out/${TAG}_dcnf_pmc2.csv gpurun_out/${TAG}_dcnf_pmc3.csv gpurun_out/${TAG}_dcnf_pmc4.csv gpurun_out/${TAG}_dcnf_pmc5.csv > gpurun_out/${TAG}_dcnf_pmc_fold.txt
PROF_TIMELINE="glue_total" bash tools/prof.sh ${TAG}_bench bench.py --steps 10 --warmup 3 --no-exact-fp32 --no-train-step --no-cpu-baseline --no-config5 --no-eager-baseline
PROF_TIMELINE="adamw_kernel" bash tools/prof.sh ${TAG}_train tools/train_bench.py --dtype bf16 --steps 1 --free-run 6
