"""otpose_amd - MI355X-native (gfx950) implementation of the OTPose hot path.

Public surface (mirrors the reference's module / operator API for the path in SURVEY.md section 8):
``OTPose`` (model/OTPose.py), ``ModulatedDeformConv`` / ``DeformableCONV`` /
``modulated_deform_conv`` (thirdparty/deform_conv), the heatmap losses (model/loss.py) and the
1-process-per-GPU data-parallel helpers (replacing nn.DataParallel, train.py:78-79).
"""
from .config import CfgNode, cfg1, cfg2, make_cfg, tiny_cfg, load_yaml  # noqa: F401
from .model import OTPose, ModulatedDeformConv, DeformableCONV  # noqa: F401
from .ops import modulated_deform_conv  # noqa: F401
from . import parallel  # noqa: F401  (installs the process-group hooks: parallel.graph_replay_safe)

__all__ = ["OTPose", "ModulatedDeformConv", "DeformableCONV", "modulated_deform_conv",
           "CfgNode", "make_cfg", "cfg1", "cfg2", "tiny_cfg", "load_yaml"]
