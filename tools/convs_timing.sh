#!/bin/bash
# development build of the library with the convs phase stamps, then tools/convs_timing.py; GPU box only
set -e
cd "$(dirname "$0")/../otpose_amd/csrc"
hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DOTP_CONVS_TIMING ${XFLAGS:-} -c convs.hip -o /tmp/convs_t.o
hipcc --offload-arch=gfx950 -shared -o /tmp/libotp_ts.so /tmp/convs_t.o $(ls *.o | grep -v '^convs.o')
cd ../..
OTPOSE_HIP_LIB=/tmp/libotp_ts.so python tools/convs_timing.py "$@"
