"""Configuration node + built-in OTPose configurations.

The reference reads its configuration through a yacs ``CfgNode`` that is accessed both as
attributes (``cfg.MODEL.NUM_JOINTS``, reference model/OTPose.py:191-199) and as a mapping
(``cfg['MODEL']['EXTRA']``, reference model/OTPose.py:185,223; model/HRNet.py:75).  yacs is not
part of this image, so :class:`CfgNode` below is a minimal stand-in with the same two access
styles plus the ``_BASE_`` yaml inheritance of reference utils/setup.py:54-69.

Only the keys the hot path reads are given defaults ("Config / flags"):
``MODEL.{NUM_JOINTS,HEATMAP_SIZE,IMAGE_SIZE,PRETRAINED,FREEZE_HRNET_WEIGHTS,DEFORMABLE_CONV.*,
DEFORMABLE_CONV_CH,OFFSET_MASK_COMBINE_CONV,EXTRA.*}`` and ``LOSS.{NAME,USE_TARGET_WEIGHT}``.
Values follow reference configs/Base_PoseTrack17.yaml:36-88 and configs/17/model_RSN.yaml:28-40.
"""
from __future__ import annotations

import copy
import os
from typing import Any, Dict, Iterable, Tuple


class CfgNode(dict):
    """dict with attribute access, recursively applied to nested mappings."""

    def __init__(self, init: Dict[str, Any] | None = None):
        super().__init__()
        for k, v in (init or {}).items():
            self[k] = v

    @staticmethod
    def _wrap(v):
        if isinstance(v, dict) and not isinstance(v, CfgNode):
            return CfgNode(v)
        if isinstance(v, tuple):
            return list(v)
        return v

    def __setitem__(self, k, v):
        super().__setitem__(k, self._wrap(v))

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v

    def merge(self, other: Dict[str, Any]) -> "CfgNode":
        """Recursive in-place merge (``other`` wins)."""
        for k, v in other.items():
            if isinstance(v, dict) and isinstance(self.get(k), dict):
                self[k].merge(v)
            else:
                self[k] = copy.deepcopy(v)
        return self

    def clone(self) -> "CfgNode":
        return CfgNode(copy.deepcopy(dict(self)))


def load_yaml(path: str) -> CfgNode:
    """Load a yaml experiment file, following a ``_BASE_`` chain relative to the file."""
    import yaml

    with open(path, "r") as f:
        raw = yaml.safe_load(f) or {}
    base = raw.pop("_BASE_", None)
    if base is not None:
        cfg = load_yaml(os.path.join(os.path.dirname(path), base))
        cfg.merge(raw)
        return cfg
    return CfgNode(raw)


def _stage(modules: int, channels: Iterable[int]) -> Dict[str, Any]:
    channels = list(channels)
    return {
        "NUM_MODULES": modules,
        "NUM_BRANCHES": len(channels),
        "BLOCK": "BASIC",
        "NUM_BLOCKS": [4] * len(channels),
        "NUM_CHANNELS": channels,
        "FUSE_METHOD": "SUM",
    }


def make_cfg(width: int = 48, image_size: Tuple[int, int] = (288, 384),
             dilations: Iterable[int] = (3, 6, 9, 12, 15), frames: int = 5, dtype: str = "fp32") -> CfgNode:
    """Built-in OTPose configuration.

    ``frames`` = frames per clip window: 5 is the reference (hard-coded at model/OTPose.py:309,320-321); 7 is this build's
    extension for BASELINE configs[4] (``MODEL.WINDOW_FRAMES``, 12 stacked maps per joint: oracle ``window_maps``).

    ``dtype`` = ``MODEL.DTYPE`` of the EVAL forward: "fp32" (the reference's arithmetic) or "fp16" (BASELINE configs[4]: backbone
    activations stored as IEEE half, one f16 MFMA per product - otpose_amd/engine_h16.py; an extension, never a default).

    ``width`` is the HRNet branch-0 width (48 = the reference's W48 yaml, 32 = HRNet-W32 used by
    BASELINE.json configs[0]); ``image_size`` is (W, H) as in the reference yaml.
    """
    w, h = image_size
    assert w % 32 == 0 and h % 32 == 0, "image size must be divisible by 32 (four HRNet branches)"
    c = [width, 2 * width, 4 * width, 8 * width]
    return CfgNode({
        "MODEL": {
            "NAME": "OTPose",
            "NUM_JOINTS": 17,
            "WINDOW_FRAMES": int(frames),
            "DTYPE": str(dtype),
            "IMAGE_SIZE": [w, h],
            "HEATMAP_SIZE": [w // 4, h // 4],
            "PRETRAINED": "",
            "FREEZE_HRNET_WEIGHTS": False,
            "DEFORMABLE_CONV_CH": 32,
            "OFFSET_MASK_COMBINE_CONV": 2,
            "DEFORMABLE_CONV": {"DILATION": list(dilations), "AGGREGATION_TYPE": "weighted_sum"},
            "EXTRA": {
                "PRETRAINED_LAYERS": ["*"],
                "FINAL_CONV_KERNEL": 1,
                "STAGE2": _stage(1, c[:2]),
                "STAGE3": _stage(4, c[:3]),
                "STAGE4": _stage(3, c[:4]),
            },
        },
        "LOSS": {"NAME": "ST_OHKW_MSELoss", "USE_TARGET_WEIGHT": True, "TOPK": 8},
        "TRAIN": {"LR": 1e-4, "WD": 0.0, "OPTIMIZER": "AdamW"},
    })


def cfg1() -> CfgNode:
    """BASELINE.json configs[0]: 256x192 clip, HRNet-W32."""
    return make_cfg(32, (192, 256))


def cfg2() -> CfgNode:
    """BASELINE.json configs[1]: 384x288 clip, HRNet-W48 (the metric's configuration)."""
    return make_cfg(48, (288, 384))


def tiny_cfg(width: int = 8, image_size: Tuple[int, int] = (64, 96), frames: int = 5, dtype: str = "fp32") -> CfgNode:
    """A reduced-width, reduced-resolution model with the full OTPose topology (test sizes)."""
    return make_cfg(width, image_size, frames=frames, dtype=dtype)


def cfg5(dtype: str = "fp32") -> CfgNode:
    """BASELINE.json configs[4]: 7-frame window at 384x288 (extension; the reference has no such model); ``dtype="fp16"`` is the
    configuration as BASELINE.json states it."""
    return make_cfg(48, (288, 384), frames=7, dtype=dtype)
