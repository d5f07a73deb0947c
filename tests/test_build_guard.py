"""The build guard of otpose_amd/csrc (CPU): no packed-fp32 arithmetic instruction in any compiled code object.

On MI355X ``v_pk_fma_f32`` / ``v_pk_mul_f32`` / ``v_pk_add_f32`` whose LOW half takes src0 from the low and src1 from the HIGH
dword of their register pairs (``op_sel:[0,1,..]`` - the SLP vectoriser's form for ``{a0 * s[1] + h, a1 * s[1] + h}``) return a
wrong low half in lanes 48-63 whenever another wave of the same SIMD issues MFMAs in the same cycles
(tools/micro/pkfma_opsel_next_to_mfma.hip).  The library is compiled with the packed-fp32 feature
switched off; ``csrc/check_isa.sh`` fails the build of any object that still holds such an instruction, and this test holds the
objects that are in the tree to it."""
import glob
import os
import shutil
import subprocess

import pytest

CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "otpose_amd", "csrc")
LLVM = os.environ.get("LLVM_BIN", "/opt/rocm/lib/llvm/bin")


def _objects():
    return sorted(glob.glob(os.path.join(CSRC, "*.o")))


@pytest.mark.skipif(not os.path.exists(os.path.join(LLVM, "llvm-objdump")), reason="no ROCm LLVM tools")
def test_no_packed_fp32_instruction_in_any_code_object():
    objs = _objects()
    if not objs:
        pytest.skip("csrc is not built here (python -c 'import __graft_entry__ as g; g.build()')")
    srcs = glob.glob(os.path.join(CSRC, "*.hip"))
    assert len(objs) == len(srcs), "stale build: %d objects for %d sources" % (len(objs), len(srcs))
    for o in objs:
        r = subprocess.run([os.path.join(CSRC, "check_isa.sh"), o], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="no hipcc")
def test_guard_rejects_an_object_built_with_packed_fp32(tmp_path):
    """The guard itself: the same source compiled WITHOUT the feature switch holds hundreds of v_pk_*_f32 and is refused."""
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    obj = str(tmp_path / "densex_pk.o")
    subprocess.check_call([hipcc, "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-I" + CSRC,
                           "-I" + os.path.join(CSRC, "..", "..", "include"), "-c", os.path.join(CSRC, "densex.hip"), "-o", obj],
                          stderr=subprocess.DEVNULL)
    r = subprocess.run([os.path.join(CSRC, "check_isa.sh"), obj], capture_output=True, text=True)
    assert r.returncode != 0 and "packed-fp32" in r.stderr
