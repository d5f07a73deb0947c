#!/bin/bash
# PMC passes over one python tool (counters only, no tracing): tools/pmc.sh <tag> "<counters pass 1>" ["<pass 2>" ...] -- script args
# writes gpurun_out/<tag>_pmc<i>.csv (the counter_collection csv of each pass)
set -u
tag=$1; shift
passes=()
while [ "$1" != "--" ]; do passes+=("$1"); shift; done
shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
script=$root/$1; shift
i=0
for p in "${passes[@]}"; do
  i=$((i+1))
  out=$root/gpurun_out/pmc_${tag}_$i
  rm -rf "$out"; mkdir -p "$out"
  (cd /tmp && TMPDIR=/tmp rocprofv3 --pmc $p --output-format csv -d "$out" -o t -- python3 "$script" "$@" > "$root/gpurun_out/${tag}_pmc$i.log" 2>&1)
  f=$(find "$out" -name '*counter_collection.csv' | tail -1)
  [ -n "$f" ] && cp "$f" "$root/gpurun_out/${tag}_pmc$i.csv"
  rm -rf "$out"
done
