"""Seeded synthetic weights and inputs for parity tests and ``bench.py``.

There is no network for datasets or checkpoints, and the reference's default initialisation
(conv N(0, 1e-3), model/OTPose.py:439-443; residual scale 1e-4, model/blocks.py:289) produces
heatmaps of magnitude 1e-19, which would make a 1e-3 max-abs check
vacuous.  :func:`fill_synthetic_` therefore writes a *calibrated* recipe into any module (or
state dict) that has the reference key set: fan-in scaled convolutions, damped residual branches,
randomised BatchNorm running statistics, residual scales of 0.5, and output layers scaled so
that every compared heatmap is O(0.1 - 1), offsets have a spread of a few pixels (samples cross
the image border) and DCN masks are O(1).

Everything is drawn from one CPU ``torch.Generator`` in a fixed (sorted-by-name) order so that
every process - the reference import that produced tests/golden, the oracle, and the HIP path on
any number of GPUs - sees identical tensors.  Nothing here depends on ``oracle/``.
"""
from __future__ import annotations

import math
import re
from typing import Dict, Tuple

import torch

WEIGHT_SEED = 1234
INPUT_SEED = 4321

# output-layer gains found by measuring activation spreads of the recipe on the oracle
# (tests/golden/make_golden.py --calibrate prints them); keyed by HRNet width and image size.
_GAINS = {
    "default": {"hrnet_final": 0.02, "final12": 0.15, "offset": 0.1, "mask": 0.02},
    "w8_64x96": {"hrnet_final": 0.0412139, "final12": 0.143413, "offset": 0.110823, "mask": 0.017563},
    "w32_192x256": {"hrnet_final": 0.0148956, "final12": 0.0547529, "offset": 0.0583049, "mask": 0.00918305},
    "w48_288x384": {"hrnet_final": 0.0220267, "final12": 0.19774, "offset": 0.131595, "mask": 0.0207387},
}


def gains_for(cfg) -> Dict[str, float]:
    """Calibrated output-layer gains of a built-in configuration (``default`` when unknown)."""
    w, h = cfg["MODEL"]["IMAGE_SIZE"]
    width = cfg["MODEL"]["EXTRA"]["STAGE2"]["NUM_CHANNELS"][0]
    return dict(_GAINS.get(f"w{width}_{w}x{h}", _GAINS["default"]))


def _fan_in(shape) -> int:
    n = 1
    for s in shape[1:]:
        n *= s
    return max(n, 1)


def _normal(gen, shape, std):
    return torch.randn(shape, generator=gen, dtype=torch.float32) * std


def synthetic_state_dict(shapes: Dict[str, Tuple[int, ...]], seed: int = WEIGHT_SEED,
                         gains: Dict[str, float] | None = None) -> Dict[str, torch.Tensor]:
    """Generate a tensor for every (name, shape) of an OTPose state dict; ``pos_embd`` buffers and
    ``num_batches_tracked`` counters are left out (their constructor values are kept)."""
    g = dict(_GAINS["default"])
    g.update(gains or {})
    gen = torch.Generator(device="cpu")
    gen.manual_seed(seed)
    out: Dict[str, torch.Tensor] = {}
    for name in sorted(shapes):
        shape = tuple(shapes[name])
        leaf = name.rsplit(".", 1)[-1]
        if leaf == "num_batches_tracked" or leaf == "pos_embd":
            continue
        is_bn_like = len(shape) == 1 and leaf in ("weight", "bias", "running_mean", "running_var")
        # ---- BatchNorm (eval mode uses the running statistics) --------------------------------
        if leaf == "running_mean":
            out[name] = _normal(gen, shape, 0.1)
        elif leaf == "running_var":
            out[name] = 0.8 + 0.4 * torch.rand(shape, generator=gen)
        elif is_bn_like and leaf == "weight" and _is_bn(name, shapes):
            gamma = 1.0 + _normal(gen, shape, 0.1)
            out[name] = gamma * _bn_branch_gain(name)
        elif is_bn_like and leaf == "bias" and _is_bn(name, shapes):
            out[name] = _normal(gen, shape, 0.1)
        # ---- transformer pieces ---------------------------------------------------------------
        elif leaf == "scale":                       # AffineDropPath
            out[name] = 0.5 + _normal(gen, shape, 0.05)
        elif len(shape) == 3 and shape[0] == 1 and leaf == "weight":   # channel LayerNorm gamma
            out[name] = 1.0 + _normal(gen, shape, 0.1)
        elif len(shape) == 3 and shape[0] == 1 and leaf == "bias":     # channel LayerNorm beta
            out[name] = _normal(gen, shape, 0.1)
        elif len(shape) == 3 and leaf == "weight":  # Conv1d (depthwise k3 or pointwise)
            out[name] = _normal(gen, shape, 1.0 / math.sqrt(_fan_in(shape)))
        # ---- DCN ------------------------------------------------------------------------------
        elif ".deform_conv." in name and leaf == "weight":
            w = _normal(gen, shape, 0.5 / math.sqrt(_fan_in(shape)))
            k = shape[2] // 2
            for o in range(min(shape[0], shape[1])):
                w[o, o, k, k] += 1.0                # identity at the centre tap + noise
            out[name] = w
        elif name.startswith("offsets_list"):
            out[name] = _normal(gen, shape, g["offset"] / math.sqrt(_fan_in(shape)))
        elif name.startswith("masks_list"):
            out[name] = _normal(gen, shape, g["mask"] / math.sqrt(_fan_in(shape)))
        # ---- output 1x1 layers ----------------------------------------------------------------
        elif name == "rough_pose_estimation_net.final_layer.weight":
            out[name] = _normal(gen, shape, g["hrnet_final"] / math.sqrt(_fan_in(shape)))
        elif re.match(r"final_layer[12]\.weight", name):
            out[name] = _normal(gen, shape, g["final12"] / math.sqrt(_fan_in(shape)))
        # ---- generic conv weights / biases ----------------------------------------------------
        elif leaf == "weight" and len(shape) == 4:
            out[name] = _normal(gen, shape, math.sqrt(2.0 / _fan_in(shape)))
        elif leaf == "bias":
            out[name] = _normal(gen, shape, 0.05)
        else:
            raise KeyError(f"synthetic recipe has no rule for {name} {shape}")
    return out


def _is_bn(name: str, shapes) -> bool:
    stem = name.rsplit(".", 1)[0]
    return (stem + ".running_var") in shapes


def _bn_branch_gain(name: str) -> float:
    """Damp the last BN of every residual branch so depth does not blow activations up."""
    if re.search(r"\.bn2\.weight$", name) and re.search(r"(branches|layer1)\.", name) and ".bn3" not in name:
        # BasicBlock.bn2 closes the residual branch; Bottleneck.bn2 does not (bn3 does)
        return 0.25 if "layer1" not in name else 1.0
    if name.endswith(".bn3.weight"):
        return 0.25
    if ".fuse_layers." in name:
        return 0.3
    if ".conv_bn_relu3.bn." in name or ".downsample.bn." in name:
        return 0.5
    return 1.0


def fill_synthetic_(module: torch.nn.Module, seed: int = WEIGHT_SEED, gains=None) -> torch.nn.Module:
    """Write the recipe into ``module`` (any module with the OTPose key set) in place.  When
    ``gains`` is None and the module carries a ``cfg`` (an OTPose), its calibrated gains are used."""
    if gains is None and getattr(module, "cfg", None) is not None:
        gains = gains_for(module.cfg)
    sd = module.state_dict()
    new = synthetic_state_dict({k: tuple(v.shape) for k, v in sd.items()}, seed, gains)
    with torch.no_grad():
        for k, v in new.items():
            sd[k].copy_(v.to(sd[k].dtype))
    return module


def synthetic_clip(batch: int, image_size, seed: int = INPUT_SEED, frames: int = 5):
    """``x`` (B, 15, H, W) ~ N(0, 1) (five ImageNet-normalised frames: cur, prev, next, pprev,
    nnext - reference script/Common.py:112-117) and ``margin`` (B, 4) float frame distances:
    [1, 1, 2, 2] with every fourth row [0, 1, 0, 2] (clip borders, reference
    dataset/PoseTrackDataset.py:263-293).  ``frames = 7`` (the configs[4] extension): (B, 21, H, W) and (B, 6) =
    [1, 1, 2, 2, 3, 3] / [0, 1, 0, 2, 0, 3]."""
    w, h = image_size
    r = (frames - 1) // 2
    gen = torch.Generator(device="cpu")
    gen.manual_seed(seed)
    x = torch.randn((batch, 3 * frames, h, w), generator=gen, dtype=torch.float32)
    margin = torch.tensor([[float(k // 2 + 1) for k in range(2 * r)]]).repeat(batch, 1)
    margin[3::4] = torch.tensor([0.0 if k % 2 == 0 else float(k // 2 + 1) for k in range(2 * r)])
    return x, margin


def synthetic_targets(batch: int, heatmap_size, num_joints: int = 17, sigma: float = 3.0,
                      seed: int = INPUT_SEED + 1):
    """Gaussian target heatmaps with an exact 1.0 at each visible joint centre and a {0,1}
    ``target_weight`` (B, J, 1) with ~15 % zeros (reference utils/heatmap.py:48-105)."""
    w, h = heatmap_size
    gen = torch.Generator(device="cpu")
    gen.manual_seed(seed)
    cx = torch.randint(0, w, (batch, num_joints), generator=gen)
    cy = torch.randint(0, h, (batch, num_joints), generator=gen)
    vis = (torch.rand((batch, num_joints), generator=gen) > 0.15).float()
    ys = torch.arange(h, dtype=torch.float32)[None, None, :, None]
    xs = torch.arange(w, dtype=torch.float32)[None, None, None, :]
    d2 = (xs - cx[..., None, None].float()) ** 2 + (ys - cy[..., None, None].float()) ** 2
    target = torch.exp(-d2 / (2 * sigma * sigma)) * vis[..., None, None]
    return target.contiguous(), vis[..., None].contiguous()
