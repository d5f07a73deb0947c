"""Pin the CPU oracle against vectors produced by the reference itself (tests/golden/make_golden.py)."""
import pytest
import torch

from oracle import otpose_oracle as O
from otpose_amd import OTPose, cfg1, cfg2, tiny_cfg
from otpose_amd import modules as M
from otpose_amd import synthetic as S

def golden_names(name, key):
    """String arrays of a fixture (the ``golden`` fixture hands out tensors only)."""
    import os
    import numpy as np
    from tests.conftest import GOLDEN
    return [str(s) for s in np.load(os.path.join(GOLDEN, name + ".npz"))[key]]


TOL = 2e-5   # fp32 re-association noise between the module graph and the functional restatement


def _sd(module, seed):
    S.fill_synthetic_(module, seed)
    return {k: v.detach() for k, v in module.state_dict().items()}


def _prefixed(sd, p):
    return {p + "." + k: v for k, v in sd.items()}


def _close(a, b, tol=TOL):
    scale = max(1.0, float(b.abs().max()))
    err = float((a - b).abs().max())
    assert err <= tol * scale, f"max abs err {err} (scale {scale})"


@pytest.mark.parametrize("tag,c,nh,stride", [("mhca_136_s1", 136, 2, 1), ("mhca_136_s2", 136, 2, 2),
                                             ("mhca_17_s1", 17, 1, 1), ("mhca_136_s2_odd", 136, 2, 2)])
def test_masked_mhca(golden, tag, c, nh, stride):
    g = golden("blocks")
    sd = _prefixed(_sd(M.MaskedMHCA(c, nh, stride, stride), 11), "a")
    _close(O.masked_mhca(sd, "a", g[tag + "_x"], nh, stride), g[tag + "_y"])


@pytest.mark.parametrize("tag,c,nh,stride", [("tblock_136_s1", 136, 2, 1), ("tblock_136_s2", 136, 2, 2),
                                             ("tblock_17_s1", 17, 1, 1)])
def test_transformer_block(golden, tag, c, nh, stride):
    g = golden("blocks")
    sd = _prefixed(_sd(M.TransformerBlock(c, nh, (stride, stride), proj_pdrop=0.1, path_pdrop=0.1), 12), "b")
    _close(O.transformer_block(sd, "b", g[tag + "_x"], nh, stride), g[tag + "_y"])


def test_channel_layernorm(golden):
    g = golden("blocks")
    sd = _prefixed(_sd(M.LayerNorm(136), 13), "ln")
    _close(O.channel_layernorm(sd, "ln", g["ln_x"]), g["ln_y"])


@pytest.mark.parametrize("tag,c,nh,arch", [("ct_136", 136, 2, (0, 6, 2)), ("ct_17", 17, 1, (0, 6, 0))])
def test_conv_transformer(golden, tag, c, nh, arch):
    g = golden("blocks")
    mod = M.ConvTransformer(c, c, nh, 3, 108, arch, h=12, proj_pdrop=0.1, path_pdrop=0.1)
    sd = _prefixed(_sd(mod, 14), "t")
    ys = O.conv_transformer(sd, "t", g[tag + "_x"], nh, arch)
    assert len(ys) == arch[2] + 1
    for i, y in enumerate(ys):
        _close(y, g[f"{tag}_y{i}"], 5e-5)


@pytest.mark.parametrize("tag,cin,cout", [("rsb_51_32", 51, 32), ("rsb_17_17", 17, 17)])
def test_rsb_chain(golden, tag, cin, cout):
    g = golden("blocks")
    sd = _prefixed(_sd(M.CHAIN_RSB_BLOCKS(cin, cout, 2), 15), "r")
    _close(O.rsb_chain(sd, "r", g[tag + "_x"]), g[tag + "_y"])


def test_hrnet_tiny(golden):
    g = golden("hrnet_tiny")
    cfg = tiny_cfg(8, (64, 96))
    sd = _prefixed(_sd(M.HRNet(cfg), 31), "h")
    stages = [cfg.MODEL.EXTRA[f"STAGE{s}"] for s in (2, 3, 4)]
    _close(O.hrnet_forward(sd, "h", g["x"], stages), g["y"])


def test_losses(golden):
    g = golden("losses")
    r = O.st_ohkw_mse_loss(g["s"], g["t"], g["g"], g["w"])
    for k in ("ohkm_loss_s", "mse_loss_s", "final_loss"):
        _close(r[k], g["st_" + k], 1e-6)
    r = O.joints_ohkm_mse_loss(g["s"], g["g"], g["w"])
    for k in ("ohkm_loss", "mse_loss", "final_loss"):
        _close(r[k], g["ohkm_" + k], 1e-6)
    _close(O.joint_mse_loss(g["s"], g["g"], g["w"]), g["jmse"], 1e-6)
    s, t = g["s"].clone().requires_grad_(), g["t"].clone().requires_grad_()
    O.st_ohkw_mse_loss(s, t, g["g"], g["w"])["final_loss"].backward()
    _close(s.grad, g["st_grad_s"], 1e-6)
    _close(t.grad, g["st_grad_t"], 1e-6)
    # both branches of loss.py:47 are exercised by the fixture
    flags = [bool(g["g"][:, j].max() == 1) for j in range(17)]
    assert any(flags) and not all(flags)


NAMES = ("output", "rough", "intersection", "prev_b", "context", "squeezed", "total_b")


def _e2e(golden, name, cfg, batch, tol):
    g = golden(name)
    m = OTPose(cfg)
    S.fill_synthetic_(m)
    sd = {k: v.detach() for k, v in m.state_dict().items()}
    x, margin = S.synthetic_clip(batch, cfg.MODEL.IMAGE_SIZE)
    with torch.no_grad():
        outs = O.otpose_forward(sd, cfg, x, margin)
    for n, o in zip(NAMES, outs):
        assert o.shape == g[n].shape
        _close(o, g[n], tol)
    # the calibrated recipe keeps every compared heatmap O(0.1 - 10): the 1e-3 bar is meaningful
    assert 0.1 < float(g["output"].abs().max()) < 10 and float(g["output"].std()) > 0.1


def test_e2e_tiny(golden):
    _e2e(golden, "e2e_tiny", tiny_cfg(8, (64, 96)), 2, 5e-5)


def test_e2e_cfg1(golden):
    _e2e(golden, "e2e_cfg1", cfg1(), 1, 5e-5)


def test_e2e_cfg2_one_clip(golden):
    _e2e(golden, "e2e_cfg2_b1", cfg2(), 1, 5e-5)


def _oracle_train_step(g, dtype):
    """The training step of script/Common.py:118-144 through the oracle (``training_bn=True``: BatchNorm batch statistics;
    the golden was made with every Dropout / drop-path probability 0) under torch autograd, on the seeded tiny model."""
    cfg = tiny_cfg(8, (64, 96))
    m = OTPose(cfg)
    S.fill_synthetic_(m)
    params = dict(m.named_parameters())
    sd = {k: (v.detach().to(dtype) if v.is_floating_point() else v.detach()) for k, v in m.state_dict().items()}
    leaves = {k: sd[k].clone().requires_grad_() for k in params}
    sd.update(leaves)
    x, margin = S.synthetic_clip(2, cfg.MODEL.IMAGE_SIZE)
    outs = O.otpose_forward(sd, cfg, x.to(dtype), margin, training_bn=True)
    tg, tw = g["target"].to(dtype), g["target_weight"].to(dtype)
    first = O.st_ohkw_mse_loss(outs[0], outs[1][:2], tg, tw)
    second = O.st_ohkw_mse_loss(outs[4], outs[4], (tg + outs[2]) / 2, tw)
    (first["final_loss"] + second["final_loss"]).backward()
    return outs, first, second, leaves


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
def test_train_step_matches_reference_golden(golden, dtype):
    """loss dict, global gradient norm, the norm of EVERY parameter gradient and 22 whole gradient tensors
    of one reference training step (tests/golden/make_golden.py::gen_train_step) pin ``otpose_forward(training_bn=True)`` +
    autograd - the yardstick of the HIP training tests (tests/test_gpu_train_e2e.py)."""
    g = golden("train_step_tiny")
    outs, first, second, leaves = _oracle_train_step(g, dtype)
    for n, o in zip(NAMES, outs):
        _close(o.detach().float(), g["out_" + n], 5e-5)
    _close(first["final_loss"].detach().float(), g["loss_first_final"], 1e-5)
    _close(first["ohkm_loss_s"].detach().float(), g["loss_ohkm_s"], 1e-5)
    _close(first["mse_loss_s"].detach().float(), g["loss_mse_s"], 1e-5)
    _close(second["final_loss"].detach().float(), g["loss_second_final"], 1e-5)
    # Per-tensor relative L2 and cosine, not element-wise equality.  The fp32 oracle runs the reference's own CPU ops in the
    # reference's order and reproduces the golden gradients to the last bit (measured 0.0); the fp64 oracle sees the fp32
    # REFERENCE's rounding as the error: forward differences of ~5e-5 flip a handful of ReLU gates in the RSB chains, and every
    # flip moves one entry of the (small) gradient tensors behind them by its full size - measured up to 1.5e-2 relative L2,
    # 1 - cos up to 1.2e-4
    tol, cos_tol = (2e-3, 1e-5) if dtype == torch.float32 else (3e-2, 5e-4)
    names = golden_names("train_step_tiny", "grad_norm_names")
    norms = g["grad_norms"].double()
    assert len(names) == len(leaves) == norms.numel()
    tot = float(torch.sqrt(sum((p.grad.double() ** 2).sum() for p in leaves.values())))
    assert abs(tot - float(g["grad_total_norm"])) <= 1e-3 * float(g["grad_total_norm"]), (tot, float(g["grad_total_norm"]))
    worst = 0.0
    for n, ref in zip(names, norms.tolist()):
        mine = float(leaves[n].grad.double().norm())
        if ref > 1e-3 * float(g["grad_total_norm"]):
            worst = max(worst, abs(mine - ref) / ref)
    assert worst <= 5e-3, worst
    bad = []
    for k in g:
        if not k.startswith("grad/"):
            continue
        ref, mine = g[k].double(), leaves[k[5:]].grad.double()
        rel = float((mine - ref).norm() / ref.norm().clamp_min(1e-30))
        cos = float((mine * ref).sum() / (mine.norm() * ref.norm()).clamp_min(1e-30))
        if rel > tol or cos < 1 - cos_tol:
            bad.append((k, rel, cos))
    assert not bad, bad


SEED_CASES = [(ws, xs) for ws in (S.WEIGHT_SEED, 777) for xs in (S.INPUT_SEED, 11, 12)]


@pytest.mark.parametrize("ws,xs", SEED_CASES[::3])       # one per weight seed on the CPU (the GPU suite runs all six)
def test_e2e_cfg1_other_seeds(golden, ws, xs):
    g = golden("e2e_seeds")
    cfg = cfg1()
    m = OTPose(cfg)
    S.fill_synthetic_(m, ws, S.gains_for(cfg))
    sd = {k: v.detach() for k, v in m.state_dict().items()}
    x, margin = S.synthetic_clip(1, cfg.MODEL.IMAGE_SIZE, seed=xs)
    with torch.no_grad():
        outs = O.otpose_forward(sd, cfg, x, margin)
    tag = f"cfg1_w{ws}_x{xs}"
    _close(outs[0], g[tag + "_output"], 5e-5)
    _close(outs[1][:1], g[tag + "_rough_cur"], 5e-5)
    _close(outs[4], g[tag + "_context"], 5e-5)
    for o, mx in zip(outs, g[tag + "_absmax"].tolist()):
        assert abs(float(o.abs().max()) - mx) <= 1e-4 * max(1.0, mx)


@pytest.mark.parametrize("ws,xs", [(777, 31), (4242, 32)])
def test_e2e_cfg2_other_weight_seeds(golden, ws, xs):
    """The headline configuration under two more weight draws (tests/golden/make_golden.py::gen_e2e_cfg2_seeds)."""
    g = golden("e2e_cfg2_seeds")
    cfg = cfg2()
    m = OTPose(cfg)
    S.fill_synthetic_(m, ws, S.gains_for(cfg))
    sd = {k: v.detach() for k, v in m.state_dict().items()}
    x, margin = S.synthetic_clip(1, cfg.MODEL.IMAGE_SIZE, seed=xs)
    with torch.no_grad():
        outs = O.otpose_forward(sd, cfg, x, margin)
    tag = f"cfg2_w{ws}_x{xs}"
    _close(outs[0], g[tag + "_output"], 5e-5)
    _close(outs[1][:1], g[tag + "_rough_cur"], 5e-5)
    _close(outs[4], g[tag + "_context"], 5e-5)
    for o, mx in zip(outs, g[tag + "_absmax"].tolist()):
        assert abs(float(o.abs().max()) - mx) <= 1e-4 * max(1.0, mx)
