#!/bin/bash
# build guard (see NOPK in the Makefile): disassemble the gfx950 code object of one compiled .o and fail if any packed-fp32
# arithmetic instruction is in it
set -e
L=${LLVM_BIN:-/opt/rocm/lib/llvm/bin}
t=$(mktemp -d)
trap 'rm -rf "$t"' EXIT
$L/llvm-objcopy --dump-section .hip_fatbin=$t/fb.bin "$1" $t/rest.o
$L/clang-offload-bundler --type=o --targets=hipv4-amdgcn-amd-amdhsa--${2:-gfx950} --input=$t/fb.bin --output=$t/dev.co --unbundle
$L/llvm-objdump -d $t/dev.co > $t/dev.s
grep -q "s_endpgm" $t/dev.s || { echo "check_isa: no device code found in $1" >&2; exit 1; }
n=$(grep -c "v_pk_\(fma\|mul\|add\)_f32" $t/dev.s || true)
if [ "$n" != "0" ]; then echo "check_isa: $n packed-fp32 instructions in $1 (see NOPK in the Makefile)" >&2; exit 1; fi
