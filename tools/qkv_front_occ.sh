#!/bin/bash
# experiment (GPU box): qkvx_front_kernel<136> under launch bounds of 2 / 3 / 4 waves per SIMD
set -e
cd "$(dirname "$0")/../otpose_amd/csrc"
for occ in 2 3 4; do
  sed "s/__global__ __launch_bounds__(256, 2) void qkvx_front_kernel/__global__ __launch_bounds__(256, $occ) void qkvx_front_kernel/" densex.hip > /tmp/dx_$occ.hip
  hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I. -I../../include -c /tmp/dx_$occ.hip -o /tmp/dx_$occ.o
  hipcc --offload-arch=gfx950 -shared -o /tmp/libotp_$occ.so /tmp/dx_$occ.o $(ls *.o | grep -v '^densex.o')
  echo "== $occ waves per SIMD"
  (cd ../.. && OTPOSE_HIP_LIB=/tmp/libotp_$occ.so python tools/qkv_front_time.py)
done
