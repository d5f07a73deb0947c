#!/bin/bash
# usage (GPU box): tools/trace_neighbors.sh <substring> tools/train_bench.py --dtype bf16 --steps 1
set -u
key=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/prof_nb
rm -rf "$out"; mkdir -p "$out"
script=$root/$1; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d "$out" -o t -- python3 "$script" "$@" > "$root/gpurun_out/nb.log" 2>&1
cd "$root"
trace=$(find "$out" -name '*kernel_trace.csv' | tail -1)
python3 tools/trace_neighbors.py "$trace" "$key" 30
rm -rf "$out"
