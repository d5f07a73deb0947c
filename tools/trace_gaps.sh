#!/bin/bash
# usage (GPU box): tools/trace_gaps.sh <marker> bench.py --steps 5 --no-exact-fp32 --no-train ...
set -u
marker=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/prof_gaps
rm -rf "$out"; mkdir -p "$out"
script=$root/$1; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d "$out" -o t -- python3 "$script" "$@" > "$root/gpurun_out/gaps.log" 2>&1
cd "$root"
trace=$(find "$out" -name '*kernel_trace.csv' | tail -1)
python3 tools/trace_gaps.py "$trace" "$marker" ${MIN_GAP:-15}
rm -rf "$out"
