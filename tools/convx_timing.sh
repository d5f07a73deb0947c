#!/bin/bash
# development build of the library with the convx phase stamps, then tools/convx_timing.py (GPU box only)
set -e
cd "$(dirname "$0")/../otpose_amd/csrc"
hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DOTP_CONVX_TIMING ${XFLAGS:-} -c convx.hip -o /tmp/convx_t.o
hipcc --offload-arch=gfx950 -shared -o /tmp/libotp_t.so /tmp/convx_t.o $(ls *.o | grep -v '^convx.o')
cd ../..
OTPOSE_HIP_LIB=/tmp/libotp_t.so python tools/convx_timing.py "$@"
