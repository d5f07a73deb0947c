#!/usr/bin/env python
"""Generate tests/golden/*.npz|json by IMPORTING the reference (build container only).

Run from the repo root:  ``python tests/golden/make_golden.py [--only NAME] [--calibrate]``

The reference at /root/reference is imported with the stubs SURVEY.md section 8c lists: empty
``torchvision`` / ``cv2`` modules, placeholder ``deform_conv_cuda`` / ``deform_pool_cuda`` extension
modules, ``.cuda()`` as identity, and the reference's ``modulated_deform_conv`` call replaced by the
oracle's differentiable CPU restatement (the CUDA operator cannot be built here - its arithmetic is
therefore NOT pinned by these vectors, everything around it is).  The reference never travels to the
GPU box; only the vectors written here do.  Weights are not stored: both sides regenerate them from
``otpose_amd.synthetic`` (seeded CPU generator), inputs likewise.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)

from otpose_amd import config as C            # noqa: E402
from otpose_amd import synthetic as S         # noqa: E402
from oracle import otpose_oracle as O         # noqa: E402


# ------------------------------------------------------------------------------------------------
def import_reference():
    """Make ``model.*`` / ``thirdparty.*`` of the reference importable on CPU."""
    if "model.OTPose" in sys.modules:
        return
    for name in ("torchvision", "torchvision.transforms", "cv2"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["torchvision"].transforms = sys.modules["torchvision.transforms"]
    # reference root must precede the repo root so that `model`, `utils`, ... resolve to it
    sys.path.insert(0, REF)
    for ext in ("deform_conv_cuda", "deform_pool_cuda"):
        name = "thirdparty.deform_conv." + ext
        sys.modules[name] = types.ModuleType(name)
    torch.nn.Module.cuda = lambda self, *a, **k: self
    torch.Tensor.cuda = lambda self, *a, **k: self
    import thirdparty.deform_conv.modules.deform_conv as mod

    def cpu_mdcn(x, offset, mask, weight, bias, stride, padding, dilation, groups, deformable_groups):
        return O.mdcn_forward(x, offset, mask, weight, bias, stride, padding, dilation, groups, deformable_groups)

    mod.modulated_deform_conv = cpu_mdcn


def ref_otpose(cfg):
    import_reference()
    from model.OTPose import OTPose
    return OTPose(cfg)


def save(name, **arrays):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **{k: (v.detach().numpy() if torch.is_tensor(v) else np.asarray(v))
                                 for k, v in arrays.items()})
    print(f"wrote {path} ({os.path.getsize(path) / 1e6:.2f} MB)")


def seeded(shape, seed, scale=1.0, dtype=torch.float32):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(shape, generator=g, dtype=dtype) * scale


def fill(module, seed):
    S.fill_synthetic_(module, seed)
    return module.eval()


# ------------------------------------------------------------------------------------------------
def gen_state_dict_keys():
    """Key set, shapes and the reference's optimizer grouping for W32 (cfg1) and the W48 count."""
    import_reference()
    from thirdparty.utils.train_utils import make_optimizer
    cfg = C.cfg1()
    m = ref_otpose(cfg)
    keys = {k: list(v.shape) for k, v in m.state_dict().items()}
    groups = {"decay": [], "no_decay": [], "pretrained": []}
    ocfg = C.CfgNode({"TRAIN": {"LR": 1e-4, "WD": 0.01, "OPTIMIZER": "AdamW"}})
    opt = make_optimizer(m, ocfg)
    names = {id(p): n for n, p in m.named_parameters()}
    for gname, pg in zip(("decay", "no_decay", "pretrained"), opt.param_groups):
        groups[gname] = sorted(names[id(p)] for p in pg["params"])
    m48 = ref_otpose(C.cfg2())
    out = {"w32": {"keys": keys, "optimizer_groups": groups,
                   "num_params": sum(p.numel() for p in m.parameters())},
           "w48": {"num_keys": len(m48.state_dict()), "num_params": sum(p.numel() for p in m48.parameters())}}
    with open(os.path.join(HERE, "state_dict_w32.json"), "w") as f:
        json.dump(out, f)
    print("state_dict_w32.json:", len(keys), "keys;", {k: len(v) for k, v in groups.items()}, out["w48"])


def gen_blocks():
    """Per-op vectors from the reference's own sub-modules at reduced sizes."""
    import_reference()
    from model.blocks import MaskedMHCA, TransformerBlock, LayerNorm
    from model.ConvVideoTransformer import ConvTransformer
    from model.RSB import CHAIN_RSB_BLOCKS
    out = {}
    with torch.no_grad():
        # channel attention incl. the layout scramble: (C, nh, stride)
        for tag, (c, nh, stride, t) in {"mhca_136_s1": (136, 2, 1, 108), "mhca_136_s2": (136, 2, 2, 108),
                                        "mhca_17_s1": (17, 1, 1, 108), "mhca_136_s2_odd": (136, 2, 2, 27)}.items():
            mod = fill(MaskedMHCA(c, nh, n_qx_stride=stride, n_kv_stride=stride, proj_pdrop=0.1), 11)
            x = seeded((2, c, t), 21)
            out[tag + "_x"], out[tag + "_y"] = x, mod(x)
        for tag, (c, nh, stride) in {"tblock_136_s1": (136, 2, 1), "tblock_136_s2": (136, 2, 2),
                                     "tblock_17_s1": (17, 1, 1)}.items():
            mod = fill(TransformerBlock(c, nh, n_ds_strides=(stride, stride), proj_pdrop=0.1, path_pdrop=0.1), 12)
            x = seeded((2, c, 108), 22)
            out[tag + "_x"], out[tag + "_y"] = x, mod(x)
        ln = fill(LayerNorm(136), 13)
        x = seeded((2, 136, 50), 23, 3.0)
        out["ln_x"], out["ln_y"] = x, ln(x)
        for tag, (c, nh, arch) in {"ct_136": (136, 2, (0, 6, 2)), "ct_17": (17, 1, (0, 6, 0))}.items():
            mod = fill(ConvTransformer(c, c, n_head=nh, n_embd_ks=3, max_len=108, arch=arch,
                                       proj_pdrop=0.1, path_pdrop=0.1, h=12), 14)
            x = seeded((2, c, 12, 9), 24)
            ys = mod(x)
            out[tag + "_x"] = x
            for i, y in enumerate(ys):
                out[f"{tag}_y{i}"] = y
        for tag, (cin, cout) in {"rsb_51_32": (51, 32), "rsb_17_17": (17, 17)}.items():
            mod = fill(CHAIN_RSB_BLOCKS(cin, cout, 2), 15)
            x = seeded((2, cin, 12, 9), 25)
            out[tag + "_x"], out[tag + "_y"] = x, mod(x)
    save("blocks", **out)


def gen_hrnet_tiny():
    """Whole HRNet (all block/transition/fuse kinds) at width 8 on 96x64 images."""
    import_reference()
    from model.HRNet import HRNet
    cfg = C.tiny_cfg(8, (64, 96))
    with torch.no_grad():
        m = fill(HRNet(cfg), 31)
        x = seeded((3, 3, 96, 64), 32)
        y = m(x)
    save("hrnet_tiny", x=x, y=y)


def gen_losses():
    import_reference()
    from model.loss import ST_OHKW_MSELoss, JointsMSE_OHKMMSELoss, JointMSELoss
    b, j, h, w = 4, 17, 16, 12
    s, t = seeded((b, j, h, w), 41, 0.5), seeded((b, j, h, w), 42, 0.5)
    g, tw = S.synthetic_targets(b, (w, h), j, sigma=2.0, seed=43)
    g[:, 3] *= 0.5          # joint 3: no exact-1 peak in the batch -> teacher branch of loss.py:47
    g[:, 7] = 0.0           # joint 7: empty ground truth
    out = dict(s=s, t=t, g=g, w=tw)
    r = ST_OHKW_MSELoss(True)(s, t, g, tw)
    out.update({"st_" + k: v for k, v in r.items()})
    r = JointsMSE_OHKMMSELoss(True)(s, g, tw)
    out.update({"ohkm_" + k: v for k, v in r.items()})
    out["jmse"] = JointMSELoss(True)(s, g, tw)
    # gradients of the default loss w.r.t. student and teacher
    s2, t2 = s.clone().requires_grad_(), t.clone().requires_grad_()
    ST_OHKW_MSELoss(True)(s2, t2, g, tw)["final_loss"].backward()
    out["st_grad_s"], out["st_grad_t"] = s2.grad, t2.grad
    save("losses", **out)


def gen_decode():
    """get_max_preds / get_final_preds of the reference (utils/heatmap.py) on seeded maps; transform_preds needs cv2
    (absent here) and is replaced by the identity, so the vectors pin the argmax and the quarter-pixel shift only."""
    import_reference()
    import utils.heatmap as H
    H.transform_preds = lambda coords, center, scale, output_size: coords
    hm = seeded((3, 17, 24, 18), 41).numpy().copy()
    hm[0, 0] = -1.0                                   # all-negative plane: coordinates masked to 0
    hm[0, 1, 5, 7] = hm[0, 1, 9, 3] = 9.0             # tie: first maximum wins
    hm[1, 2, 0, 0] = 50.0                             # corner peak: no refinement
    hm[1, 3, 2, 2] = 50.0                             # px = 2 passes the strict 1 < px test
    hm[1, 4, 1, 9] = 50.0                             # py = 1 does not
    p0, m0 = H.get_max_preds(hm.copy())
    p1, m1 = H.get_final_preds(hm.copy(), [None] * 3, [None] * 3)
    save("decode", hm=hm, max_preds=p0, maxvals=m0, final_preds=p1)


def gen_accuracy():
    """accuracy() of the reference (utils/evaluate.py:384-415) on seeded heat-map pairs.  The module's unrelated imports
    (motmetrics, shapely, utils.setup -> yacs) are absent here and replaced by empty modules; accuracy / calc_dists /
    dist_acc touch none of them."""
    import_reference()
    for name in ("motmetrics", "shapely", "shapely.geometry", "utils.setup"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["shapely"].geometry = sys.modules["shapely.geometry"]
    sys.modules["utils.setup"].convert_videos = None
    import utils.evaluate as E
    n, j, h, w = 6, 17, 24, 18
    tgt = (seeded((n, j, h, w), 51) * 0.1).numpy().copy()
    out = (seeded((n, j, h, w), 52) * 0.1).numpy().copy()
    rng = np.random.RandomState(5)
    for b in range(n):
        for k in range(j):
            ty, tx = rng.randint(0, h), rng.randint(0, w)
            tgt[b, k, ty, tx] = 1.0
            dy, dx = rng.randint(-2, 3), rng.randint(-2, 3)          # predictions 0..2.8 px away: both sides of thr
            out[b, k, min(max(ty + dy, 0), h - 1), min(max(tx + dx, 0), w - 1)] = 2.0
    tgt[:, 3] = -1.0                                  # joint without any valid target (argmax masked to 0) -> acc -1
    tgt[0, 5] = 0.0
    tgt[0, 5, 1, 7] = 1.0                             # y = 1 fails the strict > 1 test: sample ignored
    acc, avg, cnt, pred = E.accuracy(out.copy(), tgt.copy())
    acc7, avg7, cnt7, _ = E.accuracy(out.copy(), tgt.copy(), thr=0.7)
    save("accuracy", out=out, tgt=tgt, acc=acc, avg=np.array([avg, avg7]), cnt=np.array([cnt, cnt7]), pred=pred,
         acc7=acc7)
    print("  acc", np.round(acc, 3), "avg", avg, "cnt", cnt)


def _e2e(cfg, batch, name, keep_rough=True):
    with torch.no_grad():
        m = ref_otpose(cfg)
        S.fill_synthetic_(m, S.WEIGHT_SEED, S.gains_for(cfg))
        m.eval()
        x, margin = S.synthetic_clip(batch, cfg.MODEL.IMAGE_SIZE)
        outs = m(x, margin=margin)
    names = ("output", "rough", "intersection", "prev_b", "context", "squeezed", "total_b")
    arrays = {n: o for n, o in zip(names, outs) if keep_rough or n != "rough"}
    arrays["stats"] = np.array([[float(o.abs().max()), float(o.std())] for o in outs])
    save(name, **arrays)
    for n, o in zip(names, outs):
        print(f"  {n:13s} max|.|={o.abs().max():.4g} std={o.std():.4g}")


def gen_e2e_tiny():
    _e2e(C.tiny_cfg(8, (64, 96)), 2, "e2e_tiny")


def gen_e2e_cfg1():
    _e2e(C.cfg1(), 1, "e2e_cfg1")


def gen_e2e_cfg2():
    _e2e(C.cfg2(), 1, "e2e_cfg2_b1")


TRAIN_GRAD_KEYS = (
    "rough_pose_estimation_net.conv1.weight",                                   # stem conv
    "rough_pose_estimation_net.bn1.weight",
    "rough_pose_estimation_net.layer1.0.conv2.weight",                          # Bottleneck
    "rough_pose_estimation_net.stage2.0.branches.0.0.conv1.weight",             # BasicBlock
    "rough_pose_estimation_net.stage3.0.fuse_layers.0.1.0.weight",              # fuse 1x1 (+ upsample)
    "rough_pose_estimation_net.stage3.0.fuse_layers.2.0.0.0.weight",            # fuse 3x3 stride 2
    "rough_pose_estimation_net.final_layer.weight",
    "flow_encoder.stem.0.attn.query.weight",
    "temporal_encoder1.stem.0.attn.query.weight", "temporal_encoder1.stem.2.attn.value_conv.weight",
    "temporal_encoder1.stem.3.ln2.weight", "temporal_encoder1.stem.5.mlp.0.weight", "temporal_encoder1.branch.1.mlp.3.weight",
    "temporal_encoder2.stem.1.attn.proj.weight", "temporal_encoder2.branch.0.drop_path_mlp.scale",
    "final_layer1.weight", "def_fuse.layers.0.conv_bn_relu2_3_2.conv.weight",
    "offset_mask_combine_conv.layers.1.conv_bn_relu3.conv.weight",
    "offsets_list.0.0.weight", "masks_list.4.0.weight",
    "modulated_deform_conv_list.2.deform_conv.weight", "modulated_deform_conv_list.2.deform_conv.bias",
)


def gen_train_step():
    """one optimisation step of the reference loop (script/Common.py:118-144) on the imported reference
    model in ``train()`` mode (BatchNorm batch statistics + running-stat update) with every Dropout p = 0 and every drop-path
    probability 0 (the only stochastic pieces): forward, the two ST_OHKW terms of Common.py:122-130 through the reference's
    own ``ST_OHKW_MSELoss``, backward, ``clip_grad_norm_(1.0)``, one AdamW step through the reference's ``make_optimizer``
    groups.  Stored: the 7 outputs' checksums + heat-maps, the loss dict, the global gradient norm, per-parameter gradient
    norms of ALL parameters, a handful of gradient tensors, BatchNorm running statistics after the step and a few updated
    weights.  (The DCN arithmetic inside is the oracle's - see the module docstring.)"""
    import_reference()
    from model.loss import ST_OHKW_MSELoss
    from model.blocks import AffineDropPath
    from thirdparty.utils.train_utils import make_optimizer
    cfg = C.tiny_cfg(8, (64, 96))
    m = ref_otpose(cfg)
    S.fill_synthetic_(m, S.WEIGHT_SEED, S.gains_for(cfg))
    m.train()
    n_do = n_dp = 0
    for mod in m.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p, n_do = 0.0, n_do + 1
        if isinstance(mod, AffineDropPath):
            mod.drop_prob, n_dp = 0.0, n_dp + 1
    print(f"  dropout modules zeroed: {n_do}, drop-path modules zeroed: {n_dp}")
    B = 2
    x, margin = S.synthetic_clip(B, cfg.MODEL.IMAGE_SIZE)
    J = cfg.MODEL.NUM_JOINTS
    g, tw = S.synthetic_targets(B, cfg.MODEL.HEATMAP_SIZE, J, sigma=1.5, seed=77)
    g[:, 3] *= 0.5                                         # joint 3: no exact-1 peak -> teacher branch of loss.py:47
    crit = ST_OHKW_MSELoss(True)
    ocfg = C.CfgNode({"TRAIN": {"LR": 1e-3, "WD": 0.01, "OPTIMIZER": "AdamW"}})
    opt = make_optimizer(m, ocfg)
    outs = m(x, margin=margin)
    pred_t = outs[1].split(B, dim=0)[0]                    # Common.py:124
    loss = crit(outs[0], pred_t, g, tw)                    # :126
    first_final = loss["final_loss"].detach().clone()
    occlusion = (g + outs[2]) / 2                          # :127
    second = crit(outs[4], outs[4], occlusion, tw)         # :129
    loss["final_loss"] = loss["final_loss"] + second["final_loss"]
    opt.zero_grad()
    loss["final_loss"].backward()                          # :137
    grads = {n: p.grad.detach().clone() for n, p in m.named_parameters() if p.grad is not None}
    total_norm = torch.nn.utils.clip_grad_norm_(m.parameters(), 1.0)       # :138-142
    before = {n: p.detach().clone() for n, p in m.named_parameters()}
    opt.step()
    names = ("output", "rough", "intersection", "prev_b", "context", "squeezed", "total_b")
    arrays = {"out_" + n: o.detach() for n, o in zip(names, outs)}
    arrays.update(target=g, target_weight=tw,
                  loss_first_final=first_final, loss_second_final=second["final_loss"].detach(),
                  loss_final=loss["final_loss"].detach(), loss_ohkm_s=loss["ohkm_loss_s"].detach(),
                  loss_mse_s=loss["mse_loss_s"].detach(), grad_total_norm=total_norm.detach())
    pnames = sorted(grads)
    arrays["grad_norm_names"] = np.array(pnames)
    arrays["grad_norms"] = np.array([float(grads[n].double().norm()) for n in pnames])
    missing = [k for k in TRAIN_GRAD_KEYS if k not in grads]
    assert not missing, missing
    for k in TRAIN_GRAD_KEYS:
        arrays["grad/" + k] = grads[k]
    sd = m.state_dict()
    for k in ("rough_pose_estimation_net.bn1.running_mean", "rough_pose_estimation_net.bn1.running_var",
              "rough_pose_estimation_net.stage4.2.branches.3.3.bn2.running_var",
              "def_fuse.layers.0.conv_bn_relu1.bn.running_mean", "rough_pose_estimation_net.bn1.num_batches_tracked"):
        arrays["buf/" + k] = sd[k].detach()
    # one AdamW step (groups of make_optimizer: lr / 100 for the backbone, no decay on norms / biases / scales)
    for k in ("rough_pose_estimation_net.conv1.weight", "temporal_encoder1.stem.0.attn.query.weight", "final_layer1.weight",
              "flow_encoder.stem.0.ln1.weight", "offsets_list.0.0.weight"):
        arrays["step/" + k] = dict(m.named_parameters())[k].detach() - before[k]
    arrays["opt_group_sizes"] = np.array([len(pg["params"]) for pg in opt.param_groups])
    arrays["opt_group_lr"] = np.array([pg["lr"] for pg in opt.param_groups])
    arrays["opt_group_wd"] = np.array([pg["weight_decay"] for pg in opt.param_groups])
    save("train_step_tiny", **arrays)
    print("  loss", {k: float(v) for k, v in loss.items()}, "second", float(second["final_loss"]), "|g|", float(total_norm))
    big = sorted(((float(grads[n].norm()), n) for n in pnames), reverse=True)[:5]
    print("  largest gradient tensors:", big)


def gen_e2e_seeds():
    """the split-product eval path must hold 1e-3 on more than one clip / one weight draw.  cfg1
    (256x192, W32) for 3 input seeds x 2 weight seeds - stored: the output heat-maps and the rough heat-maps of the current
    frame, plus max|.| of all seven outputs - and a 2-clip cfg2 batch (384x288, W48) whose second clip sits at a sequence
    border (margin row [0, 1, 0, 2], reference dataset/PoseTrackDataset.py:263-293)."""
    names = ("output", "rough", "intersection", "prev_b", "context", "squeezed", "total_b")
    arrays = {}
    cfg = C.cfg1()
    with torch.no_grad():
        for ws in (S.WEIGHT_SEED, 777):
            m = ref_otpose(cfg)
            S.fill_synthetic_(m, ws, S.gains_for(cfg))
            m.eval()
            for xs in (S.INPUT_SEED, 11, 12):
                x, margin = S.synthetic_clip(1, cfg.MODEL.IMAGE_SIZE, seed=xs)
                outs = m(x, margin=margin)
                tag = f"cfg1_w{ws}_x{xs}"
                arrays[tag + "_output"] = outs[0]
                arrays[tag + "_rough_cur"] = outs[1][:1]
                arrays[tag + "_context"] = outs[4]
                arrays[tag + "_absmax"] = np.array([float(o.abs().max()) for o in outs])
                print(f"  {tag}: " + " ".join(f"{n}={float(o.abs().max()):.3g}" for n, o in zip(names, outs)))
        cfg = C.cfg2()
        m = ref_otpose(cfg)
        S.fill_synthetic_(m, S.WEIGHT_SEED, S.gains_for(cfg))
        m.eval()
        x, margin = S.synthetic_clip(2, cfg.MODEL.IMAGE_SIZE, seed=21)
        margin[1] = torch.tensor([0.0, 1.0, 0.0, 2.0])
        outs = m(x, margin=margin)
        for n, o in zip(names, outs):
            if n in ("output", "prev_b", "context"):
                arrays["cfg2_b2_" + n] = o
        arrays["cfg2_b2_absmax"] = np.array([float(o.abs().max()) for o in outs])
        print("  cfg2_b2: " + " ".join(f"{n}={float(o.abs().max()):.3g}" for n, o in zip(names, outs)))
    save("e2e_seeds", **arrays)


def gen_e2e_cfg2_seeds():
    """the headline configuration (cfg2: 384x288, HRNet-W48) under two MORE weight draws, one clip each
    with an input seed of its own - the split-product arithmetic's error is data dependent, so one weight draw is not a
    distribution.  Stored per draw: output heat-maps, rough heat-maps of the current frame, context, max|.| of all seven."""
    names = ("output", "rough", "intersection", "prev_b", "context", "squeezed", "total_b")
    arrays = {}
    cfg = C.cfg2()
    with torch.no_grad():
        for ws, xs in ((777, 31), (4242, 32)):
            m = ref_otpose(cfg)
            S.fill_synthetic_(m, ws, S.gains_for(cfg))
            m.eval()
            x, margin = S.synthetic_clip(1, cfg.MODEL.IMAGE_SIZE, seed=xs)
            outs = m(x, margin=margin)
            tag = f"cfg2_w{ws}_x{xs}"
            arrays[tag + "_output"] = outs[0]
            arrays[tag + "_rough_cur"] = outs[1][:1]
            arrays[tag + "_context"] = outs[4]
            arrays[tag + "_absmax"] = np.array([float(o.abs().max()) for o in outs])
            print(f"  {tag}: " + " ".join(f"{n}={float(o.abs().max()):.3g}" for n, o in zip(names, outs)))
    save("e2e_cfg2_seeds", **arrays)


def _oracle_run(cfg, b, gains):
    from otpose_amd import OTPose
    m = OTPose(cfg)
    S.fill_synthetic_(m, gains=gains)
    sd = {k: v.detach() for k, v in m.state_dict().items()}
    x, margin = S.synthetic_clip(b, cfg.MODEL.IMAGE_SIZE)
    with torch.no_grad():
        return O.otpose_forward(sd, cfg, x, margin, return_intermediates=True)


def calibrate():
    """Choose synthetic._GAINS so that rough std = 0.3, final_layer{1,2} std = 0.5, offset std = 3 px,
    mask std = 0.5; prints the table to paste into otpose_amd/synthetic.py."""
    table = {}
    for cfg, b in ((C.tiny_cfg(8, (64, 96)), 2), (C.cfg1(), 1), (C.cfg2(), 1)):
        g = {"hrnet_final": 1.0, "final12": 1.0, "offset": 1.0, "mask": 1.0}
        res, inter = _oracle_run(cfg, b, g)
        g["hrnet_final"] = 0.3 / float(res[1].std())
        res, inter = _oracle_run(cfg, b, g)
        g["final12"] = 0.5 / float(torch.cat([inter["f1"], inter["f2"]]).std())
        res, inter = _oracle_run(cfg, b, g)
        g["offset"] = 3.0 / float(torch.stack([d[0] for d in inter["dcn"]]).std())
        g["mask"] = 0.5 / float(torch.stack([d[1] for d in inter["dcn"]]).std())
        res, inter = _oracle_run(cfg, b, g)
        w, h = cfg.MODEL.IMAGE_SIZE
        key = f"w{cfg.MODEL.EXTRA.STAGE2.NUM_CHANNELS[0]}_{w}x{h}"
        table[key] = {k: float(f"{v:.6g}") for k, v in g.items()}
        print(key, table[key])
        for n, o in zip(("output", "rough", "intersection", "prev_b", "context", "squeezed", "total_b"), res):
            print(f"  {n:13s} max|.|={o.abs().max():.4g} std={o.std():.4g}")
        for k in ("x1", "f1", "f2", "def_h", "trans"):
            print(f"  {k:13s} max|.|={inter[k].abs().max():.4g} std={inter[k].std():.4g}")
        for i, (off, msk, wrp) in enumerate(inter["dcn"]):
            print(f"  dcn{i}: offset std={off.std():.3g} max={off.abs().max():.3g} mask std={msk.std():.3g} "
                  f"warp std={wrp.std():.3g}")
    print(json.dumps(table, indent=4))


GENS = {"keys": gen_state_dict_keys, "blocks": gen_blocks, "hrnet_tiny": gen_hrnet_tiny,
        "losses": gen_losses, "decode": gen_decode, "accuracy": gen_accuracy, "e2e_tiny": gen_e2e_tiny, "e2e_cfg1": gen_e2e_cfg1, "e2e_cfg2": gen_e2e_cfg2, "train_step": gen_train_step, "e2e_seeds": gen_e2e_seeds,
        "e2e_cfg2_seeds": gen_e2e_cfg2_seeds}

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default=None)
    ap.add_argument("--calibrate", action="store_true")
    a = ap.parse_args()
    torch.set_num_threads(8)
    if a.calibrate:
        calibrate()
    else:
        for k, fn in GENS.items():
            if a.only in (None, k):
                print("==", k)
                fn()
