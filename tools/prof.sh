#!/bin/bash
# rocprofv3 kernel trace of a python tool, folded into gpurun_out/<tag>_by_grid.txt (+ the --stats csv).
# usage (on the GPU box, from the repo root): tools/prof.sh <tag> tools/train_bench.py --dtype bf16 --steps 2
set -u
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/prof_$tag
rm -rf "$out"; mkdir -p "$out"
script=$root/$1; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$out" -o t -- python3 "$script" "$@" > "$root/gpurun_out/${tag}.log" 2>&1
cd "$root"
trace=$(find "$out" -name '*kernel_trace.csv' | tail -1)
stats=$(find "$out" -name '*kernel_stats.csv' | tail -1)
python3 tools/profile_summary.py "$trace" "gpurun_out/${tag}_by_grid.txt" "$(basename "$script") $*"
[ -n "$stats" ] && cp "$stats" "gpurun_out/${tag}_kernel_stats.csv"
[ -n "${PROF_TIMELINE:-}" ] && python3 tools/profile_summary.py --timeline "$trace" "gpurun_out/${tag}_timeline.txt" "$PROF_TIMELINE"
rm -rf "$out"
