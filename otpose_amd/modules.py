"""Parameter tree of OTPose with the reference's module/parameter names.

This file defines *containers*: every sub-module below owns exactly the parameters and buffers
the corresponding reference module owns, under the same attribute names, so that
``state_dict()`` / ``load_state_dict()`` round-trip with reference checkpoints and so that the
reference's optimizer grouping (thirdparty/utils/train_utils.py:62-113, which keys on the types
``HRNet``, ``LayerNorm``, ``AffineDropPath``, ``DeformableCONV``, ``CHAIN_RSB_BLOCKS``,
``nn.Conv1d`` and on name prefixes) classifies every parameter.  None of the containers computes
anything: the arithmetic lives in the HIP library behind ``include/otpose_hip.h`` and is driven by
:mod:`otpose_amd.engine` (whole forward) and :mod:`otpose_amd.ops` (operator level).

Key-set parity with the reference is checked by tests/test_state_dict_parity.py against
tests/golden/state_dict_w32.json (generated from the reference import).

Reference structure followed (names only): model/HRNet.py:57-114,160-250,341-473,500-571;
model/ConvVideoTransformer.py:21-111; model/blocks.py:67-93,185-262,283-295,336-396;
model/RSB.py:10-75,106-118; model/layers.py:9-26; model/OTPose.py:181-255.
"""
from __future__ import annotations

import math
from typing import List, Sequence

import numpy as np
import torch
from torch import nn

BN_MOMENTUM = 0.1


# ----------------------------------------------------------------------------------------------
# small helpers
# ----------------------------------------------------------------------------------------------
class _Container(nn.Module):
    """A module that only owns parameters; calling it is a programming error."""

    def forward(self, *a, **k):  # pragma: no cover - guard
        raise RuntimeError(
            f"{type(self).__name__} is a parameter container; run the model through "
            "otpose_amd.OTPose.forward (HIP engine)")


class ReLU(_Container):
    """Parameter-free placeholder that keeps Sequential indices aligned with the reference."""


class Interpolate(_Container):
    """Nearest up-sampling marker inside HRNet fuse layers (reference model/HRNet.py:574-583)."""

    def __init__(self, scale_factor: int, mode: str = "nearest"):
        super().__init__()
        self.scale_factor = scale_factor
        self.mode = mode


def _conv(cin, cout, k, stride=1, pad=0, dil=1, bias=False):
    return nn.Conv2d(cin, cout, k, stride, pad, dilation=dil, bias=bias)


def _bn(c):
    return nn.BatchNorm2d(c, momentum=BN_MOMENTUM)


def _seq_conv_bn(cin, cout, k, stride, pad, relu: bool) -> nn.Sequential:
    mods: List[nn.Module] = [_conv(cin, cout, k, stride, pad), _bn(cout)]
    if relu:
        mods.append(ReLU())
    return nn.Sequential(*mods)


# ----------------------------------------------------------------------------------------------
# HRNet (reference model/HRNet.py)
# ----------------------------------------------------------------------------------------------
class BasicBlock(_Container):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = _conv(inplanes, planes, 3, stride, 1)
        self.bn1 = _bn(planes)
        self.conv2 = _conv(planes, planes, 3, 1, 1)
        self.bn2 = _bn(planes)
        self.downsample = downsample
        self.stride = stride


class Bottleneck(_Container):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = _conv(inplanes, planes, 1)
        self.bn1 = _bn(planes)
        self.conv2 = _conv(planes, planes, 3, stride, 1)
        self.bn2 = _bn(planes)
        self.conv3 = _conv(planes, planes * 4, 1)
        self.bn3 = _bn(planes * 4)
        self.downsample = downsample
        self.stride = stride


_BLOCKS = {"BASIC": BasicBlock, "BOTTLENECK": Bottleneck}


def _block_chain(block, inplanes, planes, count, stride=1) -> nn.Sequential:
    down = None
    if stride != 1 or inplanes != planes * block.expansion:
        down = nn.Sequential(_conv(inplanes, planes * block.expansion, 1, stride), _bn(planes * block.expansion))
    chain = [block(inplanes, planes, stride, down)]
    chain += [block(planes * block.expansion, planes) for _ in range(1, count)]
    return nn.Sequential(*chain)


class HighResolutionModule(_Container):
    """Parallel branches + cross-resolution fuse (reference model/HRNet.py:341-473)."""

    def __init__(self, num_branches, block, num_blocks, num_inchannels, num_channels,
                 multi_scale_output=True):
        super().__init__()
        self.num_branches = num_branches
        self.multi_scale_output = multi_scale_output
        self.num_inchannels = list(num_inchannels)
        branches = []
        for i in range(num_branches):
            branches.append(_block_chain(block, self.num_inchannels[i], num_channels[i], num_blocks[i]))
            self.num_inchannels[i] = num_channels[i] * block.expansion
        self.branches = nn.ModuleList(branches)
        self.fuse_layers = self._fuse_layers() if num_branches > 1 else None

    def _fuse_layers(self):
        ch = self.num_inchannels
        rows = []
        for i in range(self.num_branches if self.multi_scale_output else 1):
            row: List[nn.Module | None] = []
            for j in range(self.num_branches):
                if j > i:   # lower resolution -> 1x1 conv, BN, nearest x2^(j-i)
                    row.append(nn.Sequential(_conv(ch[j], ch[i], 1), _bn(ch[i]), Interpolate(2 ** (j - i))))
                elif j == i:
                    row.append(None)
                else:       # higher resolution -> (i-j) stride-2 3x3 convs
                    steps = [_seq_conv_bn(ch[j], ch[j], 3, 2, 1, relu=True) for _ in range(i - j - 1)]
                    steps.append(_seq_conv_bn(ch[j], ch[i], 3, 2, 1, relu=False))
                    row.append(nn.Sequential(*steps))
            rows.append(nn.ModuleList(row))
        return nn.ModuleList(rows)


class HRNet(_Container):
    """HRNet pose backbone container (reference model/HRNet.py:57-114)."""

    def __init__(self, cfg, **kwargs):
        super().__init__()
        extra = cfg["MODEL"]["EXTRA"]
        self.conv1 = _conv(3, 64, 3, 2, 1)
        self.bn1 = _bn(64)
        self.conv2 = _conv(64, 64, 3, 2, 1)
        self.bn2 = _bn(64)
        self.layer1 = _block_chain(Bottleneck, 64, 64, 4)

        pre = [256]
        self.stage_cfgs = []
        for s in (2, 3, 4):
            scfg = extra[f"STAGE{s}"]
            block = _BLOCKS[scfg["BLOCK"]]
            cur = [c * block.expansion for c in scfg["NUM_CHANNELS"]]
            setattr(self, f"transition{s - 1}", self._transition(pre, cur))
            stage, pre = self._stage(scfg, block, cur, multi_scale_output=(s != 4))
            setattr(self, f"stage{s}", stage)
            self.stage_cfgs.append(scfg)
        self.pre_stage_channels = pre
        k = extra["FINAL_CONV_KERNEL"]
        self.final_layer = _conv(pre[0], cfg["MODEL"]["NUM_JOINTS"], k, 1, 1 if k == 3 else 0, bias=True)

    @staticmethod
    def _transition(pre: Sequence[int], cur: Sequence[int]) -> nn.ModuleList:
        layers: List[nn.Module | None] = []
        for i, c in enumerate(cur):
            if i < len(pre):
                layers.append(_seq_conv_bn(pre[i], c, 3, 1, 1, relu=True) if c != pre[i] else None)
            else:
                n_new = i + 1 - len(pre)
                steps = [_seq_conv_bn(pre[-1], c if j == n_new - 1 else pre[-1], 3, 2, 1, relu=True)
                         for j in range(n_new)]
                layers.append(nn.Sequential(*steps))
        return nn.ModuleList(layers)

    @staticmethod
    def _stage(scfg, block, inch, multi_scale_output=True):
        mods = []
        n = scfg["NUM_MODULES"]
        for i in range(n):
            mso = multi_scale_output or i != n - 1
            m = HighResolutionModule(scfg["NUM_BRANCHES"], block, scfg["NUM_BLOCKS"], inch,
                                     scfg["NUM_CHANNELS"], mso)
            inch = m.num_inchannels
            mods.append(m)
        return nn.Sequential(*mods), inch

    def freeze_weight(self):
        """reference model/HRNet.py:154-158"""
        for p in self.parameters():
            p.requires_grad = False


# ----------------------------------------------------------------------------------------------
# ConvTransformer (reference model/ConvVideoTransformer.py, model/blocks.py)
# ----------------------------------------------------------------------------------------------
class LayerNorm(_Container):
    """Channel LayerNorm over (B, C, T); weight/bias are (1, C, 1) (reference model/blocks.py:67-93)."""

    def __init__(self, num_channels, eps=1e-5):
        super().__init__()
        self.num_channels = num_channels
        self.eps = eps
        self.weight = nn.Parameter(torch.ones(1, num_channels, 1))
        self.bias = nn.Parameter(torch.zeros(1, num_channels, 1))


class AffineDropPath(_Container):
    """Per-channel residual scale (+ stochastic depth in training) (reference model/blocks.py:283-298)."""

    def __init__(self, num_dim, drop_prob=0.0, init_scale_value=1e-4):
        super().__init__()
        self.scale = nn.Parameter(init_scale_value * torch.ones(1, num_dim, 1))
        self.drop_prob = drop_prob


class MaskedMHCA(_Container):
    """Depthwise-conv + channel-attention parameters (reference model/blocks.py:336-396)."""

    def __init__(self, n_embd, n_head, n_qx_stride=1, n_kv_stride=1, attn_pdrop=0.0, proj_pdrop=0.0):
        super().__init__()
        assert n_embd % n_head == 0
        self.n_embd, self.n_head = n_embd, n_head
        self.n_channels = n_embd // n_head
        self.scale = 1.0 / math.sqrt(self.n_channels)
        self.n_qx_stride, self.n_kv_stride = n_qx_stride, n_kv_stride
        self.attn_pdrop, self.proj_pdrop = attn_pdrop, proj_pdrop

        def dw(stride_src):
            ks = stride_src + 1 if stride_src > 1 else 3
            return nn.Conv1d(n_embd, n_embd, ks, stride=n_kv_stride, padding=ks // 2, groups=n_embd, bias=False)

        self.query_conv = dw(n_qx_stride)
        self.query_norm = LayerNorm(n_embd)
        self.key_conv = dw(n_kv_stride)
        self.key_norm = LayerNorm(n_embd)
        self.value_conv = dw(n_kv_stride)
        self.value_norm = LayerNorm(n_embd)
        self.key = nn.Conv1d(n_embd, n_embd, 1)
        self.query = nn.Conv1d(n_embd, n_embd, 1)
        self.value = nn.Conv1d(n_embd, n_embd, 1)
        self.proj = nn.Conv1d(n_embd, n_embd, 1)


class _Marker(_Container):
    """Parameter-free slot (GELU / Dropout positions inside the reference's mlp Sequential)."""


class TransformerBlock(_Container):
    """reference model/blocks.py:191-262"""

    def __init__(self, n_embd, n_head, n_ds_strides=(1, 1), attn_pdrop=0.0, proj_pdrop=0.0, path_pdrop=0.0):
        super().__init__()
        self.stride = n_ds_strides[0]
        self.ln1 = LayerNorm(n_embd)
        self.ln2 = LayerNorm(n_embd)
        self.attn = MaskedMHCA(n_embd, n_head, n_ds_strides[0], n_ds_strides[1], attn_pdrop, proj_pdrop)
        self.pool_skip = _Marker()
        self.mlp = nn.Sequential(nn.Conv1d(n_embd, 4 * n_embd, 1), _Marker(), _Marker(),
                                 nn.Conv1d(4 * n_embd, n_embd, 1), _Marker())
        self.proj_pdrop, self.path_pdrop = proj_pdrop, path_pdrop
        if path_pdrop > 0.0:
            self.drop_path_attn = AffineDropPath(n_embd, path_pdrop)
            self.drop_path_mlp = AffineDropPath(n_embd, path_pdrop)
        else:
            self.drop_path_attn = _Marker()
            self.drop_path_mlp = _Marker()


def sinusoid_table(n_position: int, d_hid: int) -> torch.Tensor:
    """(1, d_hid, n_position) float32 table, evaluated in float64 like reference model/blocks.py:114-125."""
    pos = np.arange(n_position, dtype=np.float64)[:, None]
    hid = np.arange(d_hid)[None, :]
    angle = pos / np.power(10000.0, 2.0 * (hid // 2) / d_hid)
    table = np.where(hid % 2 == 0, np.sin(angle), np.cos(angle))
    return torch.from_numpy(table.astype(np.float32)).unsqueeze(0).transpose(1, 2).contiguous()


class ConvTransformer(_Container):
    """reference model/ConvVideoTransformer.py:21-111 (arch[0] == 0: no conv embedding is built)."""

    def __init__(self, n_in, n_embd, n_head, n_embd_ks, max_len, arch, h=72, scale_factor=2,
                 attn_pdrop=0.0, proj_pdrop=0.0, path_pdrop=0.0):
        super().__init__()
        assert len(arch) == 3 and arch[0] == 0, "conv embedding (arch[0] > 0) is unused by OTPose"
        self.arch, self.max_len, self.n_embd, self.n_head = tuple(arch), max_len, n_embd, n_head
        self.scale_factor = scale_factor
        self.register_buffer("pos_embd", sinusoid_table(max_len, n_embd) / (n_embd ** 0.5))
        self.embd = nn.ModuleList()
        self.embd_norm = nn.ModuleList()
        kw = dict(attn_pdrop=attn_pdrop, proj_pdrop=proj_pdrop, path_pdrop=path_pdrop)
        self.stem = nn.ModuleList([TransformerBlock(n_embd, n_head, (1, 1), **kw) for _ in range(arch[1])])
        self.branch = nn.ModuleList(
            [TransformerBlock(n_embd, n_head, (scale_factor, scale_factor), **kw) for _ in range(arch[2])])
        self.upsample = nn.ModuleList([_Marker() for _ in range(arch[2])])


# ----------------------------------------------------------------------------------------------
# RSB heads (reference model/RSB.py)
# ----------------------------------------------------------------------------------------------
class conv_bn_relu(_Container):  # noqa: N801 - reference class name
    def __init__(self, in_planes, out_planes, kernel_size, stride, padding, has_bn=True, has_relu=True):
        super().__init__()
        self.conv = nn.Conv2d(in_planes, out_planes, kernel_size, stride, padding)  # bias=True
        self.bn = nn.BatchNorm2d(out_planes)
        self.has_bn, self.has_relu = has_bn, has_relu


_RSB_STEPS = ("1_1", "2_1", "2_2", "3_1", "3_2", "3_3", "4_1", "4_2", "4_3", "4_4")


class RSB_BLOCK(_Container):  # noqa: N801
    expansion = 1

    def __init__(self, in_planes, planes, stride=1, downsample=None):
        super().__init__()
        self.branch_ch = in_planes * 26 // 64
        bc = self.branch_ch
        self.conv_bn_relu1 = conv_bn_relu(in_planes, 4 * bc, 1, stride, 0)
        for s in _RSB_STEPS:
            setattr(self, f"conv_bn_relu2_{s}", conv_bn_relu(bc, bc, 3, 1, 1))
        self.conv_bn_relu3 = conv_bn_relu(4 * bc, planes, 1, 1, 0, has_relu=False)
        self.downsample = downsample


class CHAIN_RSB_BLOCKS(_Container):  # noqa: N801
    def __init__(self, in_planes, out_planes, num_blocks):
        super().__init__()
        down = conv_bn_relu(in_planes, out_planes, 1, 1, 0, has_relu=False)
        blocks = [RSB_BLOCK(in_planes, out_planes, 1, downsample=down)]
        blocks += [RSB_BLOCK(out_planes, out_planes, 1) for _ in range(1, num_blocks)]
        self.layers = nn.Sequential(*blocks)
