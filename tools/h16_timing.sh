#!/bin/bash
# development build of the library with the h16 conv phase stamps, then tools/h16_timing.py; GPU box only
set -e
cd "$(dirname "$0")/../otpose_amd/csrc"
hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Xclang -target-feature -Xclang -packed-fp32-ops -DOTP_H16_TIMING ${XFLAGS:-} -c h16.hip -o /tmp/h16_t.o 2>/dev/null
hipcc --offload-arch=gfx950 -shared -o /tmp/libotp_th.so /tmp/h16_t.o $(ls *.o | grep -v '^h16.o')
cd ../..
OTPOSE_HIP_LIB=/tmp/libotp_th.so python tools/h16_timing.py "$@"
