#!/bin/bash
# register / spill / occupancy table of one .hip source: tools/kernel_resources.sh otpose_amd/csrc/nhwc.hip [filter]
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Rpass-analysis=kernel-resource-usage -c "$1" -o /tmp/kr.o 2>&1 \
 | grep -E "Function Name|VGPRs:|AGPRs:|VGPRs Spill|SGPRs Spill|Occupancy|LDS Size" \
 | sed 's/.*remark: *//; s/\[-Rpass.*//' \
 | awk '/Function Name/{if(n)print n; n=$3} /VGPRs:/{n=n" V="$2} /AGPRs:/{n=n" A="$2} /Occupancy/{n=n" occ="$3} /SGPRs Spill/{n=n" sspill="$3} /VGPRs Spill/{n=n" vspill="$3} END{print n}' \
 | c++filt | sed 's/(anonymous namespace):://g; s/(.*) V=/ V=/' | grep -E "${2:-.}"
