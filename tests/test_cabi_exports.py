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


def test_host_side_queries_of_the_temporal_encoder_kernels():
    """Shape predicates, buffer sizes and argument checks of csrc/mlp.hip / dense.hip / the 1x1 wgrad - host code only."""
    L = hip.lib()
    assert L.otp_mlp_fused_supported(136, 544, 6912) == 1
    assert L.otp_mlp_fused_supported(136, 544, 6911) == 0 and L.otp_mlp_fused_supported(17, 68, 6912) == 0
    # 34 hidden blocks of (9 + 9) x 256 fragment floats + 16 biases
    assert L.otp_mlp_fused_weight_bytes(136, 544) == 34 * ((9 + 9) * 256 + 16) * 4
    assert L.otp_mlp_fused_weight_bytes(135, 544) == 0
    assert L.otp_dense_cc_supported(136, 3456) == 1 and L.otp_dense_cc_supported(136, 27) == 0
    assert L.otp_dense_cc_weight_bytes(136) == 9 * 3072 * 4
    assert L.otp_qkv_front_table_bytes(136) == 3 * 136 * 8 * 4
    assert L.otp_ln_channel_backward_workspace(16, 136, 6912) == 2 * 136 * 16 * 108 * 4
    assert L.otp_ln_channel_backward_workspace(16, 150, 6912) == 0            # wider than the register-resident form
    assert L.otp_conv2d_wgrad_workspace(136, 544) >= 96 << 20                 # room for the 1x1 path's per-chunk tiles
    assert L.otp_mlp_fused(None, None, None, None, None, None, 1, 136, 544, 64, None) == -1
    assert L.otp_dense_cc(None, None, None, None, 1, 1, 136, 64, None) == -1
    assert L.otp_qkv_front(None, None, None, None, None, None, None, None, 1, 136, 64, 1e-5, None) == -1
    assert L.otp_loss_joints_mse(None, None, None, None, None, None, 0, 1, 17, 64, 8, 1, 0, None) == -1
