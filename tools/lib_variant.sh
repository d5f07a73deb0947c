#!/bin/bash
# development aid (GPU box): build a variant of ONE csrc file with extra -D flags, link it with the other objects of the tree
# into /tmp/libotp_var<i>.so and run a command against it.  usage: tools/lib_variant.sh <file.hip> "<flags 1>" ["<flags 2>" ...] -- cmd...
set -u
cd "$(dirname "$0")/../otpose_amd/csrc"
src=$1; shift
vars=()
while [ "$1" != "--" ]; do vars+=("$1"); shift; done
shift
i=0
for flags in "${vars[@]}"; do
  i=$((i+1))
  hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Xclang -target-feature -Xclang -packed-fp32-ops $flags -I. -I../../include -c $src -o /tmp/var_$i.o 2>/dev/null || { echo "compile failed: $flags"; continue; }
  hipcc --offload-arch=gfx950 -shared -o /tmp/libotp_var$i.so /tmp/var_$i.o $(ls *.o | grep -v "^${src%.hip}.o") || continue
  echo "== $src with [$flags]"
  (cd ../.. && OTPOSE_HIP_LIB=/tmp/libotp_var$i.so "$@" 2>&1 | grep -v amdgpu.ids)
done
