"""The C-ABI library loads on a box without a GPU and exports every symbol include/otpose_hip.h declares."""
import os
import re

from otpose_amd import hip

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "otpose_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(otp_[a-z0-9_]+)\s*\(", src)))


def test_every_declared_symbol_is_exported_and_bound():
    names = _declared()
    assert "otp_mdcn_forward" in names and "otp_conv2d" in names and len(names) >= 15
    L = hip.lib()
    for n in names:
        assert hasattr(L, n), f"{n} declared in include/otpose_hip.h but not exported"
        assert n in hip.SIGNATURES, f"{n} has no ctypes signature"
    assert sorted(hip.SIGNATURES) == names
    assert L.otp_version() >= 1


def test_bad_arguments_return_error_codes_without_a_gpu():
    L = hip.lib()
    assert L.otp_mdcn_forward(None, None, None, None, None, None, 1, 1, 1, 1, 1, 3, 3, 1, 1, 1, 1, 1, 1.0, 0.0, 0, None) == -1
    assert L.otp_conv2d(None, None, None, None, None, None, None, None, None) == -1
    assert L.otp_chan_attn_workspace(2, 136, 100, 2) > 0
    assert L.otp_loss_workspace(4, 17) > 0
