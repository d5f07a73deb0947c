"""``OTPose`` - drop-in for reference model/OTPose.py:180 behind the HIP engine.

Constructor, ``forward(x, margin=...)`` signature, 7-tuple output order and ``state_dict`` key set
follow reference model/OTPose.py:181-257, 307-394.  The forward itself is a sequence of launches
into the C-ABI library declared in include/otpose_hip.h (see :mod:`otpose_amd.engine`); there is
no PyTorch-op or CPU fallback: without a GPU or without the built library the call raises.
"""
from __future__ import annotations

import logging
import math
import os

import torch
from torch import nn

from .modules import (CHAIN_RSB_BLOCKS, ConvTransformer, HRNet, _Container)
from . import ops


class ModulatedDeformConv(nn.Module):
    """Modulated deformable convolution module (reference
    thirdparty/deform_conv/modules/deform_conv.py:85-131): owns ``weight`` (Cout, Cin/groups, kh, kw)
    and ``bias`` (Cout); ``forward(x, offset, mask)`` calls the HIP operator."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1,
                 groups=1, deformable_groups=1, bias=True):
        super().__init__()
        ks = (kernel_size, kernel_size) if isinstance(kernel_size, int) else tuple(kernel_size)
        self.in_channels, self.out_channels, self.kernel_size = in_channels, out_channels, ks
        self.stride, self.padding, self.dilation = stride, padding, dilation
        self.groups, self.deformable_groups, self.with_bias = groups, deformable_groups, bias
        self.weight = nn.Parameter(torch.empty(out_channels, in_channels // groups, *ks))
        if bias:
            self.bias = nn.Parameter(torch.empty(out_channels))
        else:
            self.register_parameter("bias", None)
        self.reset_parameters()

    def reset_parameters(self):
        n = self.in_channels * self.kernel_size[0] * self.kernel_size[1]
        bound = n ** -0.5
        with torch.no_grad():
            self.weight.uniform_(-bound, bound)
            if self.bias is not None:
                self.bias.zero_()

    def forward(self, x, offset, mask):
        return ops.modulated_deform_conv(x, offset, mask, self.weight, self.bias, self.stride,
                                         self.padding, self.dilation, self.groups, self.deformable_groups)


class DeformConv(nn.Module):
    """DCN v1 module (reference thirdparty/deform_conv/modules/deform_conv.py:10-60): no mask, no bias."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1,
                 deformable_groups=1, bias=False):
        super().__init__()
        assert not bias
        assert in_channels % groups == 0 and out_channels % groups == 0
        k = (kernel_size, kernel_size) if isinstance(kernel_size, int) else tuple(kernel_size)
        self.in_channels, self.out_channels, self.kernel_size = in_channels, out_channels, k
        self.stride, self.padding, self.dilation = stride, padding, dilation
        self.groups, self.deformable_groups = groups, deformable_groups
        self.weight = nn.Parameter(torch.empty(out_channels, in_channels // groups, *k))
        stdv = 1.0 / math.sqrt(in_channels * k[0] * k[1])
        with torch.no_grad():
            self.weight.uniform_(-stdv, stdv)

    def forward(self, x, offset):
        from . import ops
        return ops.deform_conv(x, offset, self.weight, self.stride, self.padding, self.dilation, self.groups,
                               self.deformable_groups)


class DeformableCONV(nn.Module):
    """reference model/layers.py:22-29: 3x3 modulated DCN, one deformable group per joint."""

    def __init__(self, num_joints, k, dilation):
        super().__init__()
        self.deform_conv = ModulatedDeformConv(num_joints, num_joints, (k, k), stride=1,
                                               padding=(k // 2) * dilation, dilation=dilation,
                                               deformable_groups=num_joints)

    def forward(self, x, offsets, mask):
        return self.deform_conv(x, offsets, mask)


def _plain_conv(cin, cout, dilation):
    return nn.Conv2d(cin, cout, 3, 1, padding=dilation, dilation=dilation, bias=False)


class OTPose(nn.Module):
    def __init__(self, cfg, **kwargs):
        super().__init__()
        self.logger = logging.getLogger(__name__)
        self.cfg = cfg
        m = cfg.MODEL
        extra = cfg["MODEL"]["EXTRA"]
        # frames per clip window: 5 in the reference (OTPose.py:309,320-321); 7 = the BASELINE configs[4] extension
        self.window_frames = int(m.get("WINDOW_FRAMES", 5)) if hasattr(m, "get") else 5
        if self.window_frames not in (5, 7):
            raise ValueError("MODEL.WINDOW_FRAMES must be 5 (reference) or 7 (extension)")
        self.num_frames = 8 if self.window_frames == 5 else 12    # feature maps stacked per joint (OTPose.py:188)
        self.pe_w, self.pe_h = m.HEATMAP_SIZE
        self.num_joints = m.NUM_JOINTS
        self.num_patches = self.pe_h * self.pe_w
        self.patch_dim = self.num_joints
        self.temporal_encoding_dim = self.patch_dim * self.num_frames
        self.freeze_hrnet_weights = m.FREEZE_HRNET_WEIGHTS
        self.pretrained = m.PRETRAINED
        self.pretrained_layers = extra["PRETRAINED_LAYERS"]

        self.rough_pose_estimation_net = HRNet(cfg)
        self.scale_arch, self.flow_scale_arch = (0, 6, 2), (0, 6, 0)
        tkw = dict(n_embd_ks=3, max_len=self.num_patches, h=self.pe_h, proj_pdrop=0.1, path_pdrop=0.1)
        d = self.temporal_encoding_dim
        self.temporal_encoder1 = ConvTransformer(d, d, n_head=2, arch=self.scale_arch, **tkw)
        self.temporal_encoder2 = ConvTransformer(d, d, n_head=2, arch=self.scale_arch, **tkw)
        self.flow_encoder = ConvTransformer(self.patch_dim, self.patch_dim, n_head=1,
                                            arch=self.flow_scale_arch, **tkw)

        self.deformable_conv_dilations = list(m.DEFORMABLE_CONV.DILATION)
        self.deformable_aggregation_type = m.DEFORMABLE_CONV.AGGREGATION_TYPE
        assert self.deformable_aggregation_type == "weighted_sum"
        fk = extra["FINAL_CONV_KERNEL"]
        levels = self.scale_arch[-1] + 1
        self.final_layer1 = nn.Conv2d(d * levels, self.num_joints, fk, 1, 1 if fk == 3 else 0)
        self.final_layer2 = nn.Conv2d(d * levels, self.num_joints, fk, 1, 1 if fk == 3 else 0)

        def_ch, n_blocks = m.DEFORMABLE_CONV_CH, m.OFFSET_MASK_COMBINE_CONV
        self.offset_mask_combine_conv = CHAIN_RSB_BLOCKS(self.num_joints * 3, def_ch, n_blocks)
        self.def_fuse = CHAIN_RSB_BLOCKS(self.num_joints, self.num_joints, n_blocks)
        k, j = 3, self.num_joints
        self.offsets_list = nn.ModuleList(
            [nn.Sequential(_plain_conv(def_ch, j * 2 * k * k, dd)) for dd in self.deformable_conv_dilations])
        self.masks_list = nn.ModuleList(
            [nn.Sequential(_plain_conv(def_ch, j * k * k, dd)) for dd in self.deformable_conv_dilations])
        self.modulated_deform_conv_list = nn.ModuleList(
            [DeformableCONV(j, k, dd) for dd in self.deformable_conv_dilations])

        self._engine = None
        # eval outputs are fresh tensors (reference semantics); True returns the engine's static buffers, valid until
        # the next forward (bench.py's timed loop)
        self.alias_outputs = False
        self.init_weights()

    # ---- initialisation (reference model/OTPose.py:431-503) ---------------------------------
    def init_weights(self):
        with torch.no_grad():
            for mod in self.modules():
                if isinstance(mod, nn.Conv2d):
                    mod.weight.normal_(0.0, 0.001)
                    if mod.bias is not None:
                        mod.bias.zero_()
                elif isinstance(mod, nn.BatchNorm2d):
                    mod.weight.fill_(1.0)
                    mod.bias.zero_()
                elif isinstance(mod, ModulatedDeformConv):
                    mod.weight.zero_()
                    c = mod.kernel_size[0] // 2
                    for o in range(min(mod.weight.shape[0], mod.weight.shape[1])):
                        mod.weight[o, o, c, c] = 1.0
                    if mod.bias is not None:
                        mod.bias.zero_()
                elif isinstance(mod, nn.Conv1d) and mod.bias is not None:
                    mod.bias.zero_()
        if self.pretrained and os.path.isfile(self.pretrained):
            self._load_pretrained(self.pretrained)
        elif self.pretrained:
            raise ValueError("{} is not exist!".format(self.pretrained))
        if self.freeze_hrnet_weights:
            self.rough_pose_estimation_net.freeze_weight()

    def _load_pretrained(self, path):
        """HRNet checkpoint import with the reference's prefix remap (model/OTPose.py:477-496)."""
        sd = torch.load(path, map_location="cpu")
        sd = sd.get("state_dict", sd)
        tops = {n.split(".")[1] for n, _ in self.rough_pose_estimation_net.named_modules(prefix="r") if "." in n}
        take = {}
        for name, t in sd.items():
            head = name.split(".")[0]
            if not (head in self.pretrained_layers or self.pretrained_layers[0] == "*"):
                continue
            if head == "rough_pose_estimation_net":
                take[name] = t
            elif head in tops:
                take["rough_pose_estimation_net." + name] = t
        self.load_state_dict(take, strict=False)

    # ---- forward -------------------------------------------------------------------------------
    def forward(self, x, **kwargs):
        assert "margin" in kwargs
        margin = kwargs["margin"]
        if self.training:
            # model.train(): BatchNorm batch statistics + an autograd tape over HIP kernels (otpose_amd/train.py);
            # self.train_dropout = False switches Dropout / drop-path off (deterministic comparison with the oracle)
            from .train import forward_train
            self._engine = None
            return forward_train(self, x, margin)
        if self._engine is None or not self._engine.matches(x):
            self._engine = self._engine_class()(self, x.shape[0], x.device)
        return self._engine.run(x, margin, self.alias_outputs)

    def forward_frames(self, frames_u8, margin):
        """Eval forward from raw uint8 RGB crops (B, 5, H, W, 3) in the order cur, prev, next, pprev, nnext: the
        ToTensor + Normalize + concat of the reference data pipeline (dataset/PoseTrackDataset.py:397-406,
        script/Common.py:117) run as one HIP kernel writing the stem's input buffer."""
        if self.training:
            from . import ops
            return self.forward(ops.frames_to_clip(frames_u8), margin=margin)
        b, f, h, w, _ = frames_u8.shape
        if self._engine is None or not self._engine.matches_shape(b, 3 * f, h, w, frames_u8.device):
            self._engine = self._engine_class()(self, b, frames_u8.device)
        return self._engine.run(frames_u8, margin, self.alias_outputs)

    def input_buffers(self, batch, device):
        """The eval engine's own input tensors for ``batch`` clips on ``device``: ``(x (B, 3 F, H, W) fp32, margin (B, F - 1)
        fp32)``.  A data loader that writes the next batch straight into them and passes them to ``forward`` saves the
        per-call device-to-device copy (106 MB at cfg2) - ``forward`` recognises its own buffers by address."""
        device = torch.device(device)
        w_img, h_img = self.cfg.MODEL.IMAGE_SIZE
        f = getattr(self, "window_frames", 5)
        if self._engine is None or not self._engine.matches_shape(batch, 3 * f, h_img, w_img, device):
            self._engine = self._engine_class()(self, batch, device)
        return self._engine.inp, self._engine.margin

    def eval_dtype(self) -> str:
        """``cfg.MODEL.DTYPE`` of the eval forward: "fp32" (default: fp32 tensors, the reference's arithmetic contract) or "fp16"
        (BASELINE configs[4]: the backbone on half activations, csrc/h16.hip - an extension, the reference never runs below fp32)."""
        m = self.cfg.MODEL
        dt = str(m.get("DTYPE", "fp32") if hasattr(m, "get") else getattr(m, "DTYPE", "fp32")).lower()
        if dt in ("fp32", "float32", "f32"):
            return "fp32"
        if dt in ("fp16", "float16", "half", "f16"):
            return "fp16"
        raise ValueError(f"MODEL.DTYPE must be 'fp32' or 'fp16', got {dt!r}")

    def _engine_class(self):
        if self.eval_dtype() == "fp16":
            from .engine_h16 import InferenceEngineH16
            return InferenceEngineH16
        from .engine import InferenceEngine
        return InferenceEngine

    def check_range(self):
        """Synchronise and raise ``FloatingPointError`` if an eval forward since the last check drove a value beyond the range of
        the half-precision operand pieces (|x| >= 65504: csrc/common.h, csrc/range.hip).  The engine also checks, without
        synchronising, at the start of every forward (``OTPOSE_RANGE_CHECK=sync``: at the end of the forward itself)."""
        if self._engine is not None:
            self._engine.check_range()

    def invalidate_engine(self):
        """Drop packed weights (call after changing parameters, e.g. load_state_dict)."""
        self._engine = None

    def load_state_dict(self, *a, **k):
        self._engine = None
        return super().load_state_dict(*a, **k)
