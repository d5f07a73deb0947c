#!/bin/bash
# experiments (GPU box) on the corruption of csrc/densex.hip's kernels next to the two-workgroup MLP form
# (tools/encoder_pair_stress.py): build variants of densex.hip with the given -D flags and stress each.  The flags it was used
# with (wait states / sched barriers between the MFMA pairs of dx_project, accumulators in registers of their own, store data
# kept alive past the barrier, scale / shift from global memory, DMA issued after the MFMAs) are recorded in DESIGN.md
# section 3.1d; the #ifdef hooks themselves are not in the tree.
set -e
cd "$(dirname "$0")/../otpose_amd/csrc"
i=0
for flags in "$@"; do
  i=$((i+1))
  hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 $flags -I. -I../../include -c densex.hip -o /tmp/densex_v$i.o
  hipcc --offload-arch=gfx950 -shared -o /tmp/libotp_v$i.so /tmp/densex_v$i.o $(ls *.o | grep -v '^densex.o')
  echo "== densex.hip with [$flags]"
  (cd ../.. && OTP_MLP_NT1=1 OTPOSE_HIP_LIB=/tmp/libotp_v$i.so python ${STRESS:-tools/encoder_pair_stress.py} ${ROUNDS:-80} 2>&1 | grep "next to" | head -${LINES_OUT:-2})
done
