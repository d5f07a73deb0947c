#!/bin/bash
# GPU box: build the -DOTP_NHWC_TIMING library variant and print the phase stamps of nhwc_conv_kernel (tools/nhwc_timing.py)
# usage: tools/nhwc_timing.sh cin cout h w [wgrad]
set -e
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root/otpose_amd/csrc
NOPK="-Xclang -target-feature -Xclang -packed-fp32-ops"
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 $NOPK -DOTP_NHWC_TIMING -c nhwc.hip -o /tmp/nhwc_t.o 2>/dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o /tmp/libotp_t.so /tmp/nhwc_t.o $(ls *.o | grep -v "^nhwc.o")
cd $root
OTPOSE_HIP_LIB=/tmp/libotp_t.so python tools/nhwc_timing.py "$@"
