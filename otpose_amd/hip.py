"""ctypes binding of the C-ABI library ``libotpose_hip.so`` (declared in include/otpose_hip.h).

The library is built in-tree by ``__graft_entry__.build()`` (``make -C otpose_amd/csrc``).  There is
no fallback: :func:`lib` raises if the shared object is missing, and every operator raises if its
tensors are not on a GPU.
"""
from __future__ import annotations

import ctypes
import os
import threading
from ctypes import c_float, c_int, c_size_t, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("OTPOSE_HIP_LIB") or os.path.join(_HERE, "csrc", "libotpose_hip.so")   # env: dev builds
_lib = None
_tls = threading.local()          # device of the tensor the latest stream_of() was asked about (per thread)

OTP_OK = 0
_ERRORS = {
    -1: "OTP_ERR_BAD_ARG (null pointer or non-positive dimension)",
    -2: "OTP_ERR_UNSUPPORTED (shape / dtype / stride combination not implemented)",
    -3: "OTP_ERR_LAUNCH (HIP launch error)",
    -4: "OTP_ERR_WORKSPACE (workspace too small)",
}


class ConvDesc(ctypes.Structure):
    """Mirror of ``otp_conv_desc`` (include/otpose_hip.h)."""
    _fields_ = [(n, c_int) for n in (
        "N", "Cin", "H", "W", "Cout", "kh", "kw", "stride", "pad", "dil",
        "in_ctot", "in_coff", "in2_ctot", "in2_coff", "out_ctot", "out_coff",
        "res_ctot", "res_coff", "res_up", "act", "Ho", "Wo", "frame_split")] + [("out_scale", ctypes.c_float), ("res_layout", ctypes.c_int)]


class NhwcConvDesc(ctypes.Structure):
    """Mirror of ``otp_nhwc_conv_desc`` (include/otpose_hip.h)."""
    _fields_ = [(n, c_int) for n in ("N", "H", "W", "Cin", "Cout", "kh", "kw", "stride", "pad", "dil", "out_mode")]


_ND = ctypes.POINTER(NhwcConvDesc)


class H16ConvDesc(ctypes.Structure):
    """Mirror of ``otp_h16_conv_desc`` (include/otpose_hip.h)."""
    _fields_ = [(n, c_int) for n in ("N", "Cin", "H", "W", "Cout", "stride", "act", "in_gtot", "in_goff", "out_gtot", "out_goff",
                                     "res_gtot", "res_goff")] + [("out_scale", ctypes.c_float)]


_HD = ctypes.POINTER(H16ConvDesc)

# name -> (restype, argtypes); kept in one table so tests can check every symbol is exported
SIGNATURES = {
    "otp_version": (c_int, []),
    "otp_range_flag_read": (c_int, [c_int]),
    "otp_range_poison": (c_int, [c_void_p, c_size_t, c_void_p]),
    "otp_h8_bytes": (c_size_t, [c_int] * 4),
    "otp_h8_pack": (c_int, [c_void_p, c_void_p] + [c_int] * 8 + [c_void_p]),
    "otp_h8_unpack": (c_int, [c_void_p, c_void_p] + [c_int] * 6 + [c_void_p]),
    "otp_h16_conv3x3_supported": (c_int, [_HD]),
    "otp_h16_conv3x3_weight_bytes": (c_size_t, [c_int, c_int]),
    "otp_h16_conv3x3_pack_weight": (c_int, [c_void_p] * 3 + [c_int, c_int, c_float, c_void_p]),
    "otp_h16_conv3x3": (c_int, [c_void_p] * 5 + [_HD, c_void_p]),
    "otp_h16_pointwise_supported": (c_int, [c_int, c_int]),
    "otp_h16_pointwise_weight_bytes": (c_size_t, [c_int, c_int]),
    "otp_h16_pointwise_pack": (c_int, [c_void_p] * 4 + [c_int, c_int, c_float, c_void_p]),
    "otp_h16_pointwise": (c_int, [c_void_p] * 4 + [c_int] * 12 + [c_float, c_void_p]),
    "otp_h16_stem_supported": (c_int, [c_int] * 5),
    "otp_h16_stem_weight_bytes": (c_size_t, [c_int]),
    "otp_h16_stem_pack": (c_int, [c_void_p] * 4 + [c_int, c_void_p]),
    "otp_h16_stem": (c_int, [c_void_p] * 3 + [c_int] * 5 + [c_void_p]),
    "otp_mlp_h1_weight_bytes": (c_size_t, [c_int] * 2),
    "otp_mlp_h1_pack": (c_int, [c_void_p] * 4 + [c_int] * 2 + [c_void_p]),
    "otp_ln_mlp_h1": (c_int, [c_void_p] * 3 + [c_float] + [c_void_p] * 4 + [c_int] * 4 + [c_void_p]),
    "otp_dense_h1": (c_int, [ctypes.POINTER(c_void_p)] * 4 + [c_int] * 4 + [c_void_p]),
    "otp_qkv_front_h1": (c_int, [c_void_p] * 8 + [c_int] * 3 + [c_float, c_void_p]),
    "otp_h16_upsample_add": (c_int, [ctypes.POINTER(c_void_p), ctypes.POINTER(c_int), c_int, c_void_p, c_void_p] + [c_int] * 5 + [c_void_p]),
    "otp_mdcn_forward": (c_int, [c_void_p] * 6 + [c_int] * 12 + [c_float, c_float, c_int, c_void_p]),
    "otp_mdcn_forward_ex": (c_int, [c_void_p] * 6 + [c_int] * 15 + [c_float, c_float, c_int, c_void_p]),
    "otp_mdcn_backward_ex": (c_int, [c_void_p] * 10 + [c_void_p, c_size_t] + [c_int] * 15 + [c_int, c_void_p]),
    "otp_mdcn_backward_workspace": (c_size_t, [c_int] * 7),
    "otp_mdcn_backward_workspace_ex": (c_size_t, [c_int] * 17),
    "otp_mdcn_backward": (c_int, [c_void_p] * 10 + [c_void_p, c_size_t] + [c_int] * 12 + [c_int, c_void_p]),
    "otp_conv2d_pack_weight": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "otp_conv2d": (c_int, [c_void_p] * 7 + [ctypes.POINTER(ConvDesc), c_void_p]),
    "otp_conv2d_wino_weight_bytes": (c_size_t, [c_int, c_int]),
    "otp_conv2d_wino_pack_weight": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "otp_conv2d_wino_supported": (c_int, [ctypes.POINTER(ConvDesc)]),
    "otp_conv2d_wino_last_plan": (c_int, [ctypes.POINTER(c_int)]),
    "otp_conv2d_wino": (c_int, [c_void_p] * 6 + [ctypes.POINTER(ConvDesc), c_void_p]),
    "otp_conv3x3_small_supported": (c_int, [ctypes.POINTER(ConvDesc)]),
    "otp_conv3x3_small_weight_bytes": (c_size_t, [c_int, c_int]),
    "otp_conv3x3_small_pack": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "otp_conv3x3_small": (c_int, [c_void_p] * 6 + [ctypes.POINTER(ConvDesc), c_void_p]),
    "otp_conv2d_x3_weight_bytes": (c_size_t, [c_int, c_int, c_int, c_int]),
    "otp_conv2d_x3_pack_weight": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "otp_conv2d_x3_supported": (c_int, [ctypes.POINTER(ConvDesc)]),
    "otp_conv2d_x3": (c_int, [c_void_p] * 5 + [ctypes.POINTER(ConvDesc), c_void_p]),
    "otp_s8_bytes": (c_size_t, [c_int] * 4),
    "otp_s8_pack": (c_int, [c_void_p] * 3 + [c_int] * 6 + [c_void_p]),
    "otp_s8_upsample_add": (c_int, [ctypes.POINTER(c_void_p), ctypes.POINTER(c_int), c_int] + [c_void_p] * 4 + [c_int] * 9 + [c_void_p]),
    "otp_s8_upsample_add_ex": (c_int, [ctypes.POINTER(c_void_p), ctypes.POINTER(c_int), c_int, c_void_p, c_int] + [c_void_p] * 3
                               + [c_int] * 9 + [c_void_p]),
    "otp_s8_unpack": (c_int, [c_void_p, c_void_p] + [c_int] * 4 + [c_void_p]),
    "otp_c4_unpack": (c_int, [c_void_p, c_void_p] + [c_int] * 4 + [c_void_p]),
    "otp_conv3x3_s8_supported": (c_int, [ctypes.POINTER(ConvDesc)]),
    "otp_conv3x3_s8_weight_bytes": (c_size_t, [c_int, c_int]),
    "otp_conv3x3_s8_pack_weight": (c_int, [c_void_p] * 3 + [c_int, c_int, c_void_p]),
    "otp_conv3x3_s8": (c_int, [c_void_p] * 5 + [c_int, c_void_p, ctypes.POINTER(ConvDesc), c_void_p]),
    "otp_conv3x3_s2_s8_supported": (c_int, [ctypes.POINTER(ConvDesc), c_int]),
    "otp_conv3x3_s2_s8": (c_int, [c_void_p] * 6 + [ctypes.POINTER(ConvDesc), c_void_p]),
    "otp_conv2d_set_tile": (c_int, [c_int] * 4),
    "otp_conv2d_last_plan": (c_int, [ctypes.POINTER(c_int)]),
    "otp_conv2d_plan": (c_int, [ctypes.POINTER(ConvDesc), ctypes.POINTER(c_int)]),
    "otp_conv2d_pack_weight_dgrad": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "otp_dilate": (c_int, [c_void_p, c_void_p] + [c_int] * 6 + [c_void_p]),
    "otp_grad_sumsq_scratch": (c_size_t, []),
    "otp_grad_sumsq": (c_int, [c_void_p, c_size_t, c_void_p, c_void_p]),
    "otp_adamw_step": (c_int, [c_void_p] * 4 + [c_size_t] + [c_float] * 5 + [c_int, c_void_p, c_float, c_void_p]),
    "otp_pck_accuracy": (c_int, [c_void_p] * 4 + [c_int] * 4 + [c_float, c_void_p]),
    "otp_frames_u8_to_clip": (c_int, [c_void_p] * 2 + [c_int] * 4 + [c_float] * 6 + [c_void_p]),
    "otp_conv2d_wgrad_workspace": (c_size_t, [c_int] * 2),
    "otp_conv2d_wgrad": (c_int, [c_void_p] * 3 + [c_int] * 14 + [c_void_p, c_size_t, c_void_p]),
    "otp_bn_workspace": (c_size_t, [c_int] * 3),
    "otp_bn_train_forward": (c_int, [c_void_p] * 9 + [c_void_p, c_size_t] + [c_int] * 3 + [c_float, c_float] +
                             [c_int] * 7 + [c_void_p]),
    "otp_bn_train_backward": (c_int, [c_void_p] * 10 + [c_void_p, c_size_t] + [c_int] * 9 + [c_void_p]),
    "otp_channel_sum": (c_int, [c_void_p, c_void_p, c_void_p, c_size_t] + [c_int] * 5 + [c_void_p]),
    "otp_glue_total": (c_int, [c_void_p] * 6 + [c_int] * 3 + [c_void_p]),
    "otp_glue_stack": (c_int, [c_void_p] * 10 + [c_int] * 3 + [c_void_p]),
    "otp_glue_total_n": (c_int, [c_void_p] * 6 + [c_int] * 4 + [c_void_p]),
    "otp_glue_stack_n": (c_int, [c_void_p] * 10 + [c_int] * 4 + [c_void_p]),
    "otp_ln_channel": (c_int, [c_void_p] * 5 + [c_int] * 3 + [c_float, c_void_p]),
    "otp_dwconv_ln3": (c_int, [c_void_p] * 13 + [c_int] * 4 + [c_float, c_void_p]),
    "otp_chan_attn_workspace": (c_size_t, [c_int] * 4),
    "otp_dense_cc_supported": (c_int, [c_int] * 2),
    "otp_dense_cc_weight_bytes": (c_size_t, [c_int]),
    "otp_dense_cc_pack": (c_int, [c_void_p] * 4 + [c_int, c_void_p]),
    "otp_dense_cc": (c_int, [ctypes.POINTER(c_void_p)] * 4 + [c_int] * 4 + [c_void_p]),
    "otp_qkv_front_table_bytes": (c_size_t, [c_int]),
    "otp_qkv_front_pack_table": (c_int, [c_void_p] * 10 + [c_int, c_void_p]),
    "otp_qkv_front": (c_int, [c_void_p] * 8 + [c_int] * 3 + [c_float, c_void_p]),
    "otp_mlp_fused_supported": (c_int, [c_int] * 3),
    "otp_mlp_fused_weight_bytes": (c_size_t, [c_int] * 2),
    "otp_mlp_fused_pack": (c_int, [c_void_p] * 4 + [c_int] * 2 + [c_void_p]),
    "otp_mlp_fused": (c_int, [c_void_p] * 6 + [c_int] * 4 + [c_void_p]),
    "otp_ln_mlp_fused": (c_int, [c_void_p] * 3 + [c_float] + [c_void_p] * 4 + [c_int] * 4 + [c_void_p]),
    "otp_dcn_fused_supported": (c_int, [c_int] * 5),
    "otp_dcn_fused_weight_bytes": (c_size_t, [c_int] * 2),
    "otp_dcn_fused_pack": (c_int, [c_void_p] * 5 + [c_int, c_int, c_void_p]),
    "otp_dcn_fused_workspace": (c_size_t, [c_int] * 3),
    "otp_dcn_fused_forward": (c_int, [c_void_p] * 5 + [c_size_t] + [c_int] * 5 + [ctypes.POINTER(c_int), c_int, c_float, c_void_p]),
    "otp_dense_x3_supported": (c_int, [c_int] * 2),
    "otp_dense_x3_weight_bytes": (c_size_t, [c_int]),
    "otp_dense_x3_pack": (c_int, [c_void_p] * 4 + [c_int, c_void_p]),
    "otp_dense_x3": (c_int, [ctypes.POINTER(c_void_p)] * 4 + [c_int] * 4 + [c_void_p]),
    "otp_dense_x3_pack_bf16p": (c_int, [c_void_p] * 4 + [c_int, c_void_p]),
    "otp_dense_x3_bf16p": (c_int, [ctypes.POINTER(c_void_p)] * 4 + [c_int] * 4 + [c_void_p]),
    "otp_stem_conv_x3_supported": (c_int, [c_int] * 5),
    "otp_stem_conv_x3_weight_bytes": (c_size_t, [c_int]),
    "otp_stem_conv_x3_pack": (c_int, [c_void_p] * 4 + [c_int, c_void_p]),
    "otp_stem_conv_x3": (c_int, [c_void_p] * 3 + [c_int] * 5 + [c_void_p]),
    "otp_pointwise_x3_supported": (c_int, [c_int] * 3),
    "otp_pointwise_x3_weight_bytes": (c_size_t, [c_int] * 2),
    "otp_pointwise_x3_pack": (c_int, [c_void_p] * 4 + [c_int, c_int, c_void_p]),
    "otp_pointwise_x3": (c_int, [c_void_p] * 4 + [c_int] * 11 + [c_void_p]),
    "otp_pointwise_x3_s8_supported": (c_int, [c_int] * 3),
    "otp_pointwise_x3_s8_weight_bytes": (c_size_t, [c_int] * 2),
    "otp_pointwise_x3_s8_pack": (c_int, [c_void_p] * 4 + [c_int, c_int, c_void_p]),
    "otp_pointwise_x3_s8": (c_int, [c_void_p] * 3 + [c_int] * 7 + [c_void_p]),
    "otp_pointwise_x3_s8_res": (c_int, [c_void_p] * 4 + [c_int] * 9 + [c_void_p]),
    "otp_qkv_front_x3": (c_int, [c_void_p] * 8 + [c_int] * 3 + [c_float, c_void_p]),
    "otp_mlp_x3_supported": (c_int, [c_int] * 3),
    "otp_mlp_x3_weight_bytes": (c_size_t, [c_int] * 2),
    "otp_mlp_x3_pack": (c_int, [c_void_p] * 4 + [c_int] * 2 + [c_void_p]),
    "otp_mlp_x3": (c_int, [c_void_p] * 6 + [c_int] * 4 + [c_void_p]),
    "otp_flow_block_supported": (c_int, [c_int] * 3),
    "otp_flow_front_param_floats": (c_size_t, [c_int]),
    "otp_flow_back_param_floats": (c_size_t, [c_int, c_int]),
    "otp_flow_front": (c_int, [c_void_p] * 5 + [c_int] * 3 + [c_float, c_void_p]),
    "otp_flow_back": (c_int, [c_void_p] * 4 + [c_int] * 4 + [c_float, c_void_p]),
    "otp_ln_mlp_x3": (c_int, [c_void_p] * 3 + [c_float] + [c_void_p] * 4 + [c_int] * 4 + [c_void_p]),
    "otp_chan_attn": (c_int, [c_void_p] * 4 + [c_void_p, c_size_t] + [c_int] * 4 + [c_float, c_void_p]),
    "otp_chan_attn_splits": (c_int, [c_int, c_int]),
    "otp_chan_attn_scores": (c_int, [c_void_p] * 3 + [c_int] * 3 + [c_void_p]),
    "otp_chan_attn_set_split": (c_int, [c_int]),
    "otp_chan_attn_apply": (c_int, [c_void_p] * 3 + [c_int] * 3 + [c_void_p]),
    "otp_chan_attn_scores_bf16p": (c_int, [c_void_p] * 3 + [c_int] * 3 + [c_void_p]),
    "otp_chan_attn_apply_bf16p": (c_int, [c_void_p] * 3 + [c_int] * 3 + [c_void_p]),
    "otp_transpose_scale": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_void_p]),
    "otp_softmax_backward": (c_int, [c_void_p] * 5 + [c_int] * 3 + [c_void_p]),
    "otp_ln_channel_backward": (c_int, [c_void_p] * 5 + [c_int] * 3 + [c_float, c_void_p]),
    "otp_ln_channel_backward_workspace": (c_size_t, [c_int] * 3),
    "otp_ln_channel_backward_params": (c_int, [c_void_p] * 6 + [c_void_p, c_size_t] + [c_int] * 3 + [c_float, c_void_p]),
    "otp_dwconv3_forward": (c_int, [c_void_p] * 3 + [c_int] * 4 + [c_void_p]),
    "otp_dwconv3_backward": (c_int, [c_void_p] * 5 + [c_int] * 4 + [c_void_p]),
    "otp_gelu_forward": (c_int, [c_void_p, c_void_p, c_size_t, c_void_p]),
    "otp_gelu_backward": (c_int, [c_void_p] * 3 + [c_size_t, c_void_p]),
    "otp_maxpool3s2_forward": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "otp_maxpool3s2_backward": (c_int, [c_void_p] * 3 + [c_int, c_int, c_void_p]),
    "otp_upsample_linear_backward": (c_int, [c_void_p, c_void_p] + [c_int] * 6 + [c_void_p]),
    "otp_upsample_linear": (c_int, [c_void_p, c_void_p] + [c_int] * 6 + [c_void_p]),
    "otp_upsample_add": (c_int, [c_void_p] * 3 + [c_int] * 12 + [c_void_p]),
    "otp_upsample_add_multi": (c_int, [ctypes.POINTER(c_void_p), ctypes.POINTER(c_int), c_int, c_void_p, c_void_p] + [c_int] * 9 + [c_void_p]),
    "otp_upsample_add_backward": (c_int, [c_void_p] * 4 + [c_int] * 4 + [c_void_p]),
    "otp_axpby": (c_int, [c_void_p, c_void_p, c_float, c_float, c_size_t, c_void_p]),
    "otp_heatmap_decode": (c_int, [c_void_p] * 5 + [c_int] * 5 + [c_void_p]),
    "otp_loss_workspace": (c_size_t, [c_int, c_int]),
    "otp_loss_st_ohkw_grads": (c_int, [c_void_p] * 9 + [c_void_p, c_size_t] + [c_int] * 5 + [c_void_p]),
    "otp_loss_st_ohkw": (c_int, [c_void_p] * 8 + [c_void_p, c_size_t] + [c_int] * 5 + [c_void_p]),
    "otp_nhwc_conv_weight_bytes": (c_size_t, [_ND]),
    "otp_nhwc_conv_stats_rows": (c_int, [_ND]),
    "otp_nhwc_conv_plan": (c_int, [_ND, ctypes.POINTER(c_int)]),
    "otp_nhwc_conv_pack": (c_int, [c_void_p, c_void_p, _ND, c_int, c_void_p]),
    "otp_nhwc_conv_pack_job_bytes": (c_size_t, []),
    "otp_nhwc_conv_pack_job": (c_int, [c_void_p, c_void_p, _ND, c_int, c_void_p]),
    "otp_nhwc_conv_pack_batch": (c_int, [c_void_p, c_int, c_void_p]),
    "otp_nhwc_conv_bf16": (c_int, [c_void_p] * 5 + [_ND, c_void_p]),
    "otp_nhwc_conv_bf16_res": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, _ND, c_void_p]),
    "otp_nhwc_wgrad_workspace": (c_size_t, [_ND]),
    "otp_nhwc_wgrad_bf16": (c_int, [c_void_p] * 4 + [c_size_t, _ND, c_void_p]),
    "otp_nhwc_bn_finalize": (c_int, [c_void_p, c_int, c_int, c_int, c_float] + [c_void_p] * 8 + [c_float, c_float, c_void_p]),
    "otp_nhwc_bn_apply": (c_int, [c_void_p] * 6 + [c_size_t, c_int, c_int, c_void_p]),
    "otp_nhwc_bn_backward_workspace": (c_size_t, [c_size_t, c_int]),
    "otp_nhwc_bn_backward": (c_int, [c_void_p] * 11 + [c_size_t, c_size_t, c_int, c_int, c_int, c_void_p]),
    "otp_nhwc_upsample_add": (c_int, [c_void_p] * 3 + [c_int] * 6 + [c_void_p]),
    "otp_nhwc_upsample_add_backward": (c_int, [c_void_p] * 4 + [c_int] * 6 + [c_void_p]),
    "otp_nchw_f32_to_nhwc_bf16": (c_int, [c_void_p, c_void_p] + [c_int] * 5 + [c_void_p]),
    "otp_nhwc_bf16_to_nchw_f32": (c_int, [c_void_p, c_void_p] + [c_int] * 4 + [c_void_p]),
    "otp_nhwc_dilate": (c_int, [c_void_p, c_void_p] + [c_int] * 7 + [c_void_p]),
    "otp_scale_residual": (c_int, [c_void_p] * 5 + [c_int] * 3 + [c_void_p]),
    "otp_scale_residual_backward_workspace": (c_size_t, [c_int] * 3),
    "otp_scale_residual_backward": (c_int, [c_void_p] * 7 + [c_size_t] + [c_int] * 3 + [c_void_p]),
    "otp_nhwc_channel_sum_workspace": (c_size_t, [c_size_t, c_int]),
    "otp_nhwc_channel_sum": (c_int, [c_void_p] * 3 + [c_size_t, c_size_t, c_int, c_int, c_void_p]),
    "otp_gelu_bf16_forward": (c_int, [c_void_p, c_void_p, c_size_t, c_void_p]),
    "otp_gelu_bf16_backward": (c_int, [c_void_p] * 3 + [c_size_t, c_void_p]),
    "otp_nhwc_conv_bn_bf16": (c_int, [c_void_p] * 10 + [ctypes.c_float, ctypes.c_float, c_void_p, c_void_p, c_int, _ND, c_void_p]),
    "otp_nhwc_mlp_fused_supported": (c_int, [_ND]),
    "otp_nhwc_mlp_up_bf16": (c_int, [c_void_p] * 6 + [ctypes.c_float, ctypes.c_ulonglong, _ND, c_void_p]),
    "otp_nhwc_mlp_down_dgrad_bf16": (c_int, [c_void_p] * 5 + [ctypes.c_float, _ND, c_void_p]),
    "otp_gelu_dropout_bf16_forward": (c_int, [c_void_p] * 3 + [c_size_t, ctypes.c_float, ctypes.c_ulonglong, c_void_p]),
    "otp_gelu_dropout_bf16_backward": (c_int, [c_void_p] * 4 + [c_size_t, ctypes.c_float, c_void_p]),
    "otp_loss_joints_mse": (c_int, [c_void_p] * 5 + [c_void_p, c_size_t] + [c_int] * 6 + [c_void_p]),
}


def lib():
    """Return the loaded library, loading it on first use; raise loudly when it is absent."""
    global _lib
    if _lib is None:
        if not os.path.isfile(LIB_PATH):
            raise RuntimeError(
                f"HIP library {LIB_PATH} is missing - run `python -c 'import __graft_entry__ as g; g.build()'` "
                "(or `make -C otpose_amd/csrc`). otpose_amd has no CPU or PyTorch-op fallback.")
        cdll = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(cdll, name)
            fn.restype = res
            fn.argtypes = args
        # one arithmetic switch for the whole library: OTPOSE_CONV_MATH=f32 keeps every product on the f32 MFMA
        cdll.otp_chan_attn_set_split(0 if os.environ.get("OTPOSE_CONV_MATH", "x3") == "f32" else 1)
        import torch
        if torch.cuda.is_available():
            cdll.otp_range_flag_read(0)       # allocates the range guard's pinned word now, outside any stream capture
        _lib = _DeviceGuarded(cdll)
    return _lib


class _DeviceGuarded:
    """The loaded library with every entry point wrapped so that a launch runs with the device of its tensors current:
    the kernels are enqueued on the stream ``stream_of(t)`` returned, and a HIP launch (and ``hipFuncSetAttribute``)
    applies to the *current* device.  PyTorch's current device is per thread and DataParallel-style callers already set
    it, so the guard only acts when a caller hands tensors of another device (single-process multi-device use)."""

    def __init__(self, cdll):
        self._cdll = cdll
        for name in SIGNATURES:
            setattr(self, name, self._wrap(getattr(cdll, name)))

    @staticmethod
    def _wrap(fn):
        import torch

        cur = getattr(torch._C, "_cuda_getDevice", None) or torch.cuda.current_device      # (the C call: no lazy-init check per launch)

        def call(*args):
            dev = getattr(_tls, "dev", None)
            if dev is not None and dev != cur():
                with torch.cuda.device(dev):
                    return fn(*args)
            return fn(*args)
        return call


def check(status: int, what: str):
    if status != OTP_OK:
        raise RuntimeError(f"{what} failed: {_ERRORS.get(status, status)}")


def ptr(t):
    """Device pointer of a tensor (``None`` -> NULL)."""
    return None if t is None else c_void_p(t.data_ptr())


def stream_of(t):
    """The HIP stream PyTorch is currently enqueuing on for ``t``'s device."""
    import torch
    dev = t.device.index
    _tls.dev = dev
    raw = getattr(torch._C, "_cuda_getCurrentRawStream", None)     # the handle without building a torch.cuda.Stream object
    if raw is not None:                                              # (5 us per launch on the host-bound training forward)
        return c_void_p(raw(dev))
    return c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


_SIDE_STREAMS = {}            # device -> side streams shared by every inference engine and training graph of the process


def side_streams(device, n, first=0):
    """Side streams ``first`` .. ``first + n - 1`` of ``device`` from one process-wide pool (created on first use).
    The runtime multiplexes HIP streams onto a handful of hardware queues (four by default): an engine and a training graph
    that each created their own three would share queues and serialise branches that are meant to overlap (measured: the
    training step after two engines had been built ran 158 instead of 142 ms).  Engines and training steps of one process
    do not run concurrently, so they can share the streams."""
    import torch
    device = torch.device(device)
    if device.index is None:
        device = torch.device("cuda", torch.cuda.current_device())
    pool = _SIDE_STREAMS.setdefault(device, [])
    while len(pool) < first + n:
        pool.append(torch.cuda.Stream(device))
    return pool[first:first + n]

