"""Every other C entry point vs plain PyTorch CPU / the oracle on seeded inputs (float32)."""
import ctypes
import math


import pytest
import torch
import torch.nn.functional as F

from oracle import otpose_oracle as O
from otpose_amd import hip, ops
from otpose_amd import modules as M
from otpose_amd import synthetic as S
from tests.conftest import seeded

pytestmark = pytest.mark.gpu


def _close(a, b, tol=2e-5):
    a = a.detach().cpu()
    assert a.shape == b.shape
    err = float((a - b).abs().max())
    assert err <= tol * max(1.0, float(b.abs().max())), f"max abs err {err} (ref max {float(b.abs().max())})"


CONV_CASES = [  # N, Cin, H, W, Cout, k, stride, pad, dil
    (2, 48, 24, 18, 48, 3, 1, 1, 1),
    (2, 3, 32, 24, 64, 3, 2, 1, 1),       # stem: Cin not a multiple of 4
    (2, 64, 16, 12, 256, 1, 1, 0, 1),     # 1x1 (flat mode)
    (1, 32, 24, 18, 306, 3, 1, 15, 15),   # offset conv, dilation 15
    (1, 32, 24, 18, 153, 3, 1, 6, 6),
    (3, 20, 12, 9, 20, 3, 1, 1, 1),       # RSB channel counts
    (2, 6, 12, 9, 6, 3, 1, 1, 1),
    (2, 96, 12, 9, 192, 3, 2, 1, 1),      # fuse down-sampling
    (1, 408, 12, 9, 17, 1, 1, 0, 1),      # final layers
    (2, 136, 1, 108, 544, 1, 1, 0, 1),    # MLP GEMM on (B, C, T)
    (1, 17, 1, 50, 17, 1, 1, 0, 1),
    (2, 13, 7, 5, 13, 3, 1, 1, 1),        # odd everything
]


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv2d_matches_torch(case):
    n, cin, h, w, cout, k, stride, pad, dil = case
    x = seeded((n, cin, h, w), 1)
    wt = seeded((cout, cin, k, k), 2, 1.0 / math.sqrt(cin * k * k))
    sc, sh = 1.0 + 0.1 * seeded((cout,), 3), seeded((cout,), 4)
    ref = F.conv2d(x, wt, None, stride, pad, dil) * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)
    out = ops.conv2d(x.cuda(), wt.cuda(), sc.cuda(), sh.cuda(), stride, pad, dil)
    _close(out, ref)
    res = seeded(ref.shape, 5)
    out = ops.conv2d(x.cuda(), wt.cuda(), sc.cuda(), sh.cuda(), stride, pad, dil, act=ops.ACT_RELU, res=res.cuda())
    _close(out, F.relu(ref + res))
    out = ops.conv2d(x.cuda(), wt.cuda(), None, sh.cuda(), stride, pad, dil, act=ops.ACT_GELU)
    _close(out, F.gelu(F.conv2d(x, wt, sh, stride, pad, dil)))


@pytest.mark.parametrize("cout", [100, 250, 144])
@pytest.mark.parametrize("tile", [(9, 3, 1, 4)])
def test_conv1x1_tall_tiles(tile, cout):
    """The tall accumulator tiles of the 1x1 GEMMs (forced through the tuning hook), including a ragged last M tile
    (Cout16 not a multiple of the tile height) and the fused epilogue."""
    L = hip.lib()
    x = seeded((3, 136, 1, 460), 1)
    wt = seeded((cout, 136, 1, 1), 2, 0.05)
    sc, sh = 1 + 0.1 * seeded((cout,), 3), seeded((cout,), 4)
    ref = F.conv2d(x, wt) * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)
    res = seeded(ref.shape, 5)
    try:
        assert L.otp_conv2d_set_tile(*tile) == 0
        out = ops.conv2d(x.cuda(), wt.cuda(), sc.cuda(), sh.cuda(), 1, 0, 1, act=ops.ACT_GELU, res=res.cuda())
        plan = (ctypes.c_int * 8)()
        L.otp_conv2d_last_plan(plan)
        assert tuple(plan[:4]) == tile                      # the window kernel ran with the forced tile
        _close(out, F.gelu(ref + res))
    finally:
        L.otp_conv2d_set_tile(0, 0, 0, 0)


@pytest.mark.parametrize("tile", [(1, 7, 1, 1), (2, 7, 2, 2), (3, 7, 1, 4), (2, 9, 4, 1), (1, 9, 1, 3), (2, 9, 2, 1)])
def test_conv2d_every_tile_shape(tile):
    """Results do not depend on the workgroup tiling (forced through the tuning hook)."""
    L = hip.lib()
    x = seeded((2, 40, 20, 14), 1)
    wt = seeded((100, 40, 3, 3), 2, 0.05)
    ref = F.conv2d(x, wt, None, 1, 1, 1)
    try:
        assert L.otp_conv2d_set_tile(*tile) == 0
        _close(ops.conv2d(x.cuda(), wt.cuda(), None, None, 1, 1, 1), ref)
        _close(ops.conv2d(x.cuda(), wt[:, :, 1:2, 1:2].contiguous().cuda(), None, None, 1, 0, 1),
               F.conv2d(x, wt[:, :, 1:2, 1:2]))
    finally:
        L.otp_conv2d_set_tile(0, 0, 0, 0)


def test_conv2d_views_in2_upsample_framesplit():
    # channel-sliced input/output + pre-added second input (RSB staircase)
    t = seeded((2, 24, 12, 9), 1)
    o11 = seeded((2, 6, 12, 9), 2)
    wt = seeded((6, 6, 3, 3), 3, 0.2)
    cat = torch.zeros(2, 24, 12, 9)
    ref = cat.clone()
    ref[:, 12:18] = F.conv2d(t[:, 6:12] + o11, wt, None, 1, 1)
    tg, og, cg = t.cuda(), o11.cuda(), cat.cuda()
    iv, i2, ov = ops.View(tg, 6, 6), ops.View(og), ops.View(cg, 12, 6)
    d = ops.conv_desc(iv, ov, 6, 3, 3, 1, 1, 1, in2=i2)
    ops.conv2d_launch(iv, ops.pack_conv_weight(wt.cuda()), None, None, ov, d, in2=i2)
    _close(cg, ref)
    # nearest-upsample accumulate (HRNet fuse): out = relu(res + up4(conv1x1(x)))
    x = seeded((2, 32, 6, 4), 4)
    w1 = seeded((16, 32, 1, 1), 5, 0.2)
    hi = seeded((2, 16, 24, 16), 6)
    ref = F.relu(hi + F.interpolate(F.conv2d(x, w1), scale_factor=4, mode="nearest"))
    out = ops.conv2d(x.cuda(), w1.cuda(), act=ops.ACT_RELU, res=hi.cuda(), res_up=4)
    _close(out, ref)
    # in-place accumulate (res aliases out)
    acc = hi.cuda().clone()
    av = ops.View(acc)
    xv = ops.View(x.cuda())
    d = ops.conv_desc(xv, av, 16, 1, 1, 1, 0, 1, res=av, res_up=4)
    ops.conv2d_launch(xv, ops.pack_conv_weight(w1.cuda()), None, None, av, d, res=av)
    _close(acc, hi + F.interpolate(F.conv2d(x, w1), scale_factor=4, mode="nearest"))
    # frame split: (B, 15, H, W) read as (5B, 3, H, W), model/OTPose.py:317
    clip = seeded((2, 15, 16, 12), 7)
    w3 = seeded((8, 3, 3, 3), 8, 0.3)
    ref = F.conv2d(torch.cat(clip.split(3, dim=1), 0), w3, None, 2, 1)
    out = torch.empty(10, 8, 8, 6, device="cuda")
    cv, ov = ops.View(clip.cuda()), ops.View(out)
    d = ops.conv_desc(cv, ov, 8, 3, 3, 2, 1, 1, frame_split=2, cin=3)
    ops.conv2d_launch(cv, ops.pack_conv_weight(w3.cuda()), None, None, ov, d)
    _close(out, ref)


@pytest.mark.parametrize("f,relu", [(2, True), (4, False), (8, True)])
def test_upsample_add(f, relu):
    low = seeded((3, 5, 6, 4), 1)
    res = seeded((3, 5, 6 * f, 4 * f), 2)
    ref = res + F.interpolate(low, scale_factor=f, mode="nearest")
    ref = F.relu(ref) if relu else ref
    _close(ops.upsample_add(low.cuda(), res.cuda(), f, relu), ref, 1e-6)
    acc = res.cuda().clone()
    ops.upsample_add(low.cuda(), acc, f, relu, out=acc)                      # in place, like the fuse layers
    _close(acc, ref, 1e-6)


@pytest.mark.parametrize("c,t", [(136, 300), (17, 77), (200, 65)])
def test_ln_channel_and_pool(c, t):
    x = seeded((2, c, t), 1, 2.0) + 0.5
    g, b = 1 + 0.1 * seeded((c,), 2), seeded((c,), 3)
    sd = {"n.weight": g.view(1, c, 1), "n.bias": b.view(1, c, 1)}
    y, p = ops.ln_channel(x.cuda(), g.cuda(), b.cuda(), pool=True)
    _close(y, O.channel_layernorm(sd, "n", x))
    _close(p, F.max_pool1d(x, 3, 2, 1))


@pytest.mark.parametrize("c,nh,stride,t", [(136, 2, 1, 108), (136, 2, 2, 108), (17, 1, 1, 108), (136, 2, 2, 27)])
def test_mhca_pipeline_matches_golden(golden, c, nh, stride, t):
    """dwconv+LN -> q/k/v GEMMs -> channel attention (scores, softmax, PV with the scrambled store) ->
    proj, against the vectors the reference's MaskedMHCA produced (tests/golden/blocks.npz)."""
    tag = f"mhca_{c}_s{stride}" + ("_odd" if t == 27 else "")
    g = golden("blocks")
    mod = M.MaskedMHCA(c, nh, stride, stride)
    S.fill_synthetic_(mod, 11)
    x = g[tag + "_x"].cuda()
    dev = lambda p: p.detach().cuda().contiguous()          # noqa: E731
    qn, kn, vn = ops.dwconv_ln3(
        x, [dev(mod.query_conv.weight), dev(mod.key_conv.weight), dev(mod.value_conv.weight)],
        [dev(mod.query_norm.weight), dev(mod.key_norm.weight), dev(mod.value_norm.weight)],
        [dev(mod.query_norm.bias), dev(mod.key_norm.bias), dev(mod.value_norm.bias)], stride)
    lin = lambda m_, z: ops.conv2d(z.unsqueeze(2), dev(m_.weight), None, dev(m_.bias)).squeeze(2)   # noqa: E731
    q, k, v = lin(mod.query, qn), lin(mod.key, kn), lin(mod.value, vn)
    att = ops.chan_attn(q, k, v, nh, mod.scale)
    y = lin(mod.proj, att)
    _close(y, g[tag + "_y"], 5e-5)


def test_fused_transformer_block_matches_golden(golden):
    """A whole TransformerBlock (model/blocks.py:264-280, eval mode) assembled from the launches the engine uses for the
    temporal encoders - ln_channel, qkv_front, chan_attn, dense_cc (proj + drop-path scale + residual), ln_mlp_fused -
    against the vectors the reference's TransformerBlock produced (tests/golden/blocks.npz, tblock_136_s1)."""
    c, nh = 136, 2
    g = golden("blocks")
    blk = M.TransformerBlock(c, nh, (1, 1), proj_pdrop=0.1, path_pdrop=0.1)
    S.fill_synthetic_(blk, 12)
    a = blk.attn
    x = g["tblock_136_s1_x"].cuda()
    d = lambda p: p.detach().cuda().float().contiguous()          # noqa: E731
    ln1 = ops.ln_channel(x, d(blk.ln1.weight).reshape(-1), d(blk.ln1.bias).reshape(-1), blk.ln1.eps)
    table = ops.pack_qkv_table(d(a.query_conv.weight), d(a.key_conv.weight), d(a.value_conv.weight),
                               d(a.query_norm.weight), d(a.query_norm.bias), d(a.key_norm.weight), d(a.key_norm.bias),
                               d(a.value_norm.weight), d(a.value_norm.bias))
    packs = [ops.pack_dense_cc(d(m.weight), None, d(m.bias)) for m in (a.query, a.key, a.value)]
    q, k, v = ops.qkv_front(ln1, table, packs, a.query_norm.eps)
    att = ops.chan_attn(q, k, v, nh, a.scale)
    sa = d(blk.drop_path_attn.scale).reshape(-1)
    (y,) = ops.dense_cc([att], [ops.pack_dense_cc(d(a.proj.weight), sa, d(a.proj.bias) * sa)], [x])
    sm = d(blk.drop_path_mlp.scale).reshape(-1)
    packed = ops.pack_mlp_weights(d(blk.mlp[0].weight), d(blk.mlp[0].bias), d(blk.mlp[3].weight))
    out = ops.ln_mlp_fused(y, d(blk.ln2.weight).reshape(-1), d(blk.ln2.bias).reshape(-1), blk.ln2.eps, packed, sm,
                           d(blk.mlp[3].bias) * sm)
    _close(out, g["tblock_136_s1_y"], 5e-5)


@pytest.mark.parametrize("streams", [False, True])
@pytest.mark.parametrize("tag,cin,cout", [("rsb_51_32", 51, 32), ("rsb_17_17", 17, 17)])
def test_rsb_chain_launches_match_golden(golden, tag, cin, cout, streams):
    """The two RSB chains of the warping head (model/RSB.py:10-103: `offset_mask_combine_conv` 51 -> 32 and `def_fuse` 17 -> 17)
    as the engine launches them - 1x1 convs on channel-sliced views, the ten staircase convs with their pre-added second input
    on csrc/conv_small.hip, BatchNorm / bias / ReLU / residual in the epilogues, off-critical-path convs on a side stream -
    against the vectors the reference's CHAIN_RSB_BLOCKS produced (tests/golden/blocks.npz).  VERDICT r03: these goldens
    pinned only the CPU oracle."""
    from otpose_amd.engine import InferenceEngine
    from otpose_amd.ops import View
    g = golden("blocks")
    chain = M.CHAIN_RSB_BLOCKS(cin, cout, 2)
    S.fill_synthetic_(chain, 15)
    chain.eval()
    x = g[tag + "_x"]
    eng = InferenceEngine.bare("cuda", multi_stream=streams)
    xin = eng.new(*x.shape)
    xin.copy_(x)
    eng.inp = xin
    with torch.no_grad():
        out = eng.rsb_chain(chain, View(xin))
    torch.cuda.synchronize()
    eng._launch_all()
    torch.cuda.synchronize()
    assert len(eng.ops) >= 24
    _close(out.t, g[tag + "_y"], 5e-5)


def test_flow_encoder_block_matches_golden(golden):
    """The C = 17 TransformerBlock of the flow encoder as the engine runs it - otp_flow_front (ln1 + depthwise convs +
    LayerNorms + q / k / v projections), otp_chan_attn, otp_flow_back (proj + residual, ln2, MLP + residual) - against the
    vectors the reference's TransformerBlock produced (tests/golden/blocks.npz, tblock_17_s1; model/blocks.py:264-280)."""
    c, nh = 17, 1
    g = golden("blocks")
    blk = M.TransformerBlock(c, nh, (1, 1), proj_pdrop=0.1, path_pdrop=0.1)
    S.fill_synthetic_(blk, 12)
    x = g["tblock_17_s1_x"].cuda()
    assert ops.flow_block_supported(blk, c, x.shape[2])
    dev = x.device
    q, k, v = ops.flow_front(x, ops.pack_flow_front(blk, dev), blk.ln1.eps)
    att = ops.chan_attn(q, k, v, nh, blk.attn.scale)
    out = ops.flow_back(x, att, ops.pack_flow_back(blk, dev), blk.mlp[0].out_channels, blk.ln2.eps)
    _close(out, g["tblock_17_s1_y"], 5e-5)
    # ragged length (T not a multiple of the 256-token workgroup) and the sequence ends of the depthwise conv: the three
    # projections against the generic launches they replace
    xr = seeded((3, c, 333), 5).cuda()
    a = blk.attn
    d = lambda p: p.detach().cuda().float().contiguous()          # noqa: E731
    ln1 = ops.ln_channel(xr, d(blk.ln1.weight).reshape(-1), d(blk.ln1.bias).reshape(-1), blk.ln1.eps)
    qn, kn, vn = ops.dwconv_ln3(ln1, [d(a.query_conv.weight), d(a.key_conv.weight), d(a.value_conv.weight)],
                                [d(a.query_norm.weight), d(a.key_norm.weight), d(a.value_norm.weight)],
                                [d(a.query_norm.bias), d(a.key_norm.bias), d(a.value_norm.bias)], 1)
    lin = lambda m_, z: ops.conv2d(z.unsqueeze(2), d(m_.weight), None, d(m_.bias)).squeeze(2)   # noqa: E731
    for got, ref in zip(ops.flow_front(xr, ops.pack_flow_front(blk, dev), blk.ln1.eps),
                        (lin(a.query, qn), lin(a.key, kn), lin(a.value, vn))):
        _close(got, ref.cpu(), 2e-5)


def test_chan_attn_full_size_vs_oracle_slice():
    """cfg2 size (B=2 of 16, C=136, T=6912): compare with the CPU oracle arithmetic."""
    b, c, t, nh = 2, 136, 6912, 2
    q, k, v = seeded((b, c, t), 1, 0.3), seeded((b, c, t), 2, 0.3), seeded((b, c, t), 3)
    hs = c // nh
    att = (q.view(b, nh, hs, t) * (1 / math.sqrt(hs))) @ k.view(b, nh, hs, t).transpose(-2, -1)
    ref = (F.softmax(att, -1) @ v.view(b, nh, hs, t)).transpose(2, 3).contiguous().view(b, c, t)
    out = ops.chan_attn(q.cuda(), k.cuda(), v.cuda(), nh, 1 / math.sqrt(hs))
    _close(out, ref, 1e-4)


@pytest.mark.parametrize("c,nh,t", [(204, 2, 1728), (192, 2, 500), (204, 2, 6912)])
def test_chan_attn_wide_heads_vs_fp64(c, nh, t):
    """Head sizes 96 / 102 (the 12 x 17 stacked maps of the 7-frame window, BASELINE configs[4]): the split-product score and
    P . v kernels with their LDS images sized at run time (round 4; before, heads wider than 80 ran on the f32-MFMA kernels)
    against float64, ragged and full lengths."""
    b = 2
    q, k, v = seeded((b, c, t), 1, 0.3), seeded((b, c, t), 2, 0.3), seeded((b, c, t), 3)
    hs = c // nh
    att = (q.double().view(b, nh, hs, t) * (1 / math.sqrt(hs))) @ k.double().view(b, nh, hs, t).transpose(-2, -1)
    ref = (F.softmax(att, -1) @ v.double().view(b, nh, hs, t)).transpose(2, 3).contiguous().view(b, c, t)
    out = ops.chan_attn(q.cuda(), k.cuda(), v.cuda(), nh, 1 / math.sqrt(hs))
    _close(out, ref.float(), 1e-4)


@pytest.mark.parametrize("f", [1, 2, 4])
def test_upsample_linear(f):
    x = seeded((2, 5, 27), 1)
    ref = x if f == 1 else F.interpolate(x, scale_factor=f, mode="linear", align_corners=False)
    _close(ops.upsample_linear(x.cuda(), f), ref, 1e-6)
    big = torch.zeros(2, 12, 27 * f, device="cuda")
    ops.upsample_linear(x.cuda(), f, big, 4)
    _close(big[:, 4:9], ref, 1e-6)
    assert float(big[:, :4].abs().max()) == 0 and float(big[:, 9:].abs().max()) == 0


def test_loss_matches_golden_and_oracle(golden):
    g = golden("losses")
    r = ops.st_ohkw_loss(g["s"].cuda(), g["t"].cuda(), g["g"].cuda(), g["w"].cuda(), with_grad=True)
    for k in ("ohkm_loss_s", "mse_loss_s", "final_loss"):
        _close(r[k].reshape(()), g["st_" + k].reshape(()), 1e-5)
    _close(r["grad_s"], g["st_grad_s"], 1e-5)
    _close(r["grad_t"], g["st_grad_t"], 1e-5)
    flags = [int(g["g"][:, j].max() == 1) for j in range(17)]
    assert r["flags"].cpu().tolist() == flags
    # externally supplied flags (multi-GPU: MAX-reduced over ranks) change the branch taken
    forced = torch.zeros(17, dtype=torch.int32)
    r2 = ops.st_ohkw_loss(g["s"].cuda(), g["t"].cuda(), g["g"].cuda(), g["w"].cuda(), flags=forced.cuda())
    ref2 = O.st_ohkw_mse_loss(g["s"], g["t"], g["g"], g["w"], global_flags=forced)
    _close(r2["final_loss"].reshape(()), ref2["final_loss"].reshape(()), 1e-5)


@pytest.mark.parametrize("T", [32, 250, 1152, 864])
def test_mlp_fused_matches_fp64(T, monkeypatch):
    """csrc/mlp.hip vs the MLP half of TransformerBlock.forward (model/blocks.py:248-254, 277-279) in fp64; T = 250 leaves
    a ragged last workgroup and half-empty waves."""
    B, C, HID = 2, 136, 544
    if T % 432 == 0:
        # the balanced two-pass form (one 8-wave workgroup per 27 column tiles; picked by itself from B * T / 432 >= 192)
        monkeypatch.setenv("OTP_MLP_BALANCED", "2")
    x, res = seeded((B, C, T), 11), seeded((B, C, T), 12)
    w1, w2 = seeded((HID, C, 1), 13) / C ** 0.5, seeded((C, HID, 1), 14) / HID ** 0.5
    b1, b2, sc = seeded((HID,), 15) * 0.5, seeded((C,), 16), seeded((C,), 17)
    hidden = F.gelu(F.conv1d(x.double(), w1.double(), b1.double()))
    ref = res.double() + sc.double()[None, :, None] * F.conv1d(hidden, w2.double(), b2.double())
    assert ops.mlp_fused_supported(C, HID, T) and not ops.mlp_fused_supported(C, HID, T + 1)
    packed = ops.pack_mlp_weights(w1.cuda(), b1.cuda(), w2.cuda())
    out = ops.mlp_fused(x.cuda(), packed, sc.cuda(), (b2 * sc).cuda(), res.cuda())
    _close(out, ref.float(), 2e-6)
    # ln2 in front (otp_ln_mlp_fused): the residual is the un-normalised input
    g, be = 1.0 + 0.3 * seeded((C,), 18), 0.2 * seeded((C,), 19)
    mu = res.double().mean(1, keepdim=True)
    rc = res.double() - mu
    ln = rc / torch.sqrt((rc * rc).mean(1, keepdim=True) + 1e-5) * g.double()[None, :, None] + be.double()[None, :, None]
    ref_ln = res.double() + sc.double()[None, :, None] * F.conv1d(F.gelu(F.conv1d(ln, w1.double(), b1.double())), w2.double(),
                                                                  b2.double())
    out_ln = ops.ln_mlp_fused(res.cuda(), g.cuda(), be.cuda(), 1e-5, packed, sc.cuda(), (b2 * sc).cuda())
    _close(out_ln, ref_ln.float(), 3e-6)
    # in place on the residual (the engine may alias them)
    r2 = res.cuda().clone()
    ops.mlp_fused(x.cuda(), packed, sc.cuda(), (b2 * sc).cuda(), r2, out=r2)
    assert torch.equal(r2, out)


@pytest.mark.parametrize("C", [136, 204])
@pytest.mark.parametrize("T", [32, 250, -250, 1152, 864])
def test_mlp_x3_matches_fp64(T, C, monkeypatch):
    """csrc/mlpx.hip (split-bf16 products) vs the same fp64 MLP as test_mlp_fused_matches_fp64.  Tolerance 2e-5 of the output
    range: two bf16 pieces per operand carry 16 mantissa bits + rounding (measured 3e-6; the f32-MFMA kernel 4e-7).
    C = 204: the 12 x 17 stacked maps of the 7-frame window (BASELINE configs[4])."""
    B, HID = 2, 4 * C
    if T % 432 == 0:
        monkeypatch.setenv("OTP_MLP_BALANCED", "2")         # the balanced two-pass form (one workgroup per 27 column tiles)
    elif T < 0:
        T = -T
        monkeypatch.setenv("OTP_MLP_NT1", "0")              # two token tiles per wave (the default is one: two workgroups per CU)
    x, res = seeded((B, C, T), 11), seeded((B, C, T), 12)
    w1, w2 = seeded((HID, C, 1), 13) / C ** 0.5, seeded((C, HID, 1), 14) / HID ** 0.5
    b1, b2, sc = seeded((HID,), 15) * 0.5, seeded((C,), 16), seeded((C,), 17)
    hidden = F.gelu(F.conv1d(x.double(), w1.double(), b1.double()))
    ref = res.double() + sc.double()[None, :, None] * F.conv1d(hidden, w2.double(), b2.double())
    assert ops.mlp_x3_supported(C, HID, T) and not ops.mlp_x3_supported(C, HID, T + 1)
    packed = ops.pack_mlp_x3_weights(w1.cuda(), b1.cuda(), w2.cuda())
    out = ops.mlp_x3(x.cuda(), packed, sc.cuda(), (b2 * sc).cuda(), res.cuda())
    _close(out, ref.float(), 2e-5)
    g, be = 1.0 + 0.3 * seeded((C,), 18), 0.2 * seeded((C,), 19)
    mu = res.double().mean(1, keepdim=True)
    rc = res.double() - mu
    ln = rc / torch.sqrt((rc * rc).mean(1, keepdim=True) + 1e-5) * g.double()[None, :, None] + be.double()[None, :, None]
    ref_ln = res.double() + sc.double()[None, :, None] * F.conv1d(F.gelu(F.conv1d(ln, w1.double(), b1.double())), w2.double(),
                                                                  b2.double())
    out_ln = ops.ln_mlp_x3(res.cuda(), g.cuda(), be.cuda(), 1e-5, packed, sc.cuda(), (b2 * sc).cuda())
    _close(out_ln, ref_ln.float(), 2e-5)
    r2 = res.cuda().clone()
    ops.mlp_x3(x.cuda(), packed, sc.cuda(), (b2 * sc).cuda(), r2, out=r2)
    assert torch.equal(r2, out)


@pytest.mark.parametrize("T", [32, 250, 1152])
def test_dense_cc_matches_fp64(T):
    """csrc/dense.hip vs the pointwise projections of MaskedMHCA (model/blocks.py:383-386) in fp64: three problems in one
    launch (bias only, like query / key / value) and one with scale, shift and residual (like proj + drop-path scale)."""
    B, C = 2, 136
    xs = [seeded((B, C, T), 21 + i) for i in range(3)]
    ws = [seeded((C, C, 1), 31 + i) / C ** 0.5 for i in range(3)]
    bs = [seeded((C,), 41 + i) for i in range(3)]
    assert ops.dense_cc_supported(C, T) and not ops.dense_cc_supported(C, T + 1) and not ops.dense_cc_supported(17, T)
    packs = [ops.pack_dense_cc(w.cuda(), None, b.cuda()) for w, b in zip(ws, bs)]
    outs = ops.dense_cc([x.cuda() for x in xs], packs)
    for x, w, b, o in zip(xs, ws, bs, outs):
        _close(o, F.conv1d(x.double(), w.double(), b.double()).float(), 2e-6)
    sc, res = seeded((C,), 51), seeded((B, C, T), 52)
    pk = ops.pack_dense_cc(ws[0].cuda(), sc.cuda(), (bs[0] * sc).cuda())
    (o,) = ops.dense_cc([xs[1].cuda()], [pk], [res.cuda()])
    ref = res.double() + sc.double()[None, :, None] * F.conv1d(xs[1].double(), ws[0].double(), bs[0].double())
    _close(o, ref.float(), 2e-6)


@pytest.mark.parametrize("C", [136, 204])
@pytest.mark.parametrize("T", [32, 250, 1152])
def test_dense_and_qkv_front_x3_match_fp64(T, C):
    """csrc/densex.hip (split-bf16 products) vs the same fp64 references as test_dense_cc_matches_fp64 /
    test_qkv_front_matches_fp64; 2e-5 of the output range (measured 3e-6).  C = 204: the 7-frame window's encoders."""
    B, eps = 2, 1e-5
    xs = [seeded((B, C, T), 21 + i) for i in range(3)]
    ws = [seeded((C, C, 1), 31 + i) / C ** 0.5 for i in range(3)]
    bs = [seeded((C,), 41 + i) for i in range(3)]
    assert ops.dense_x3_supported(C, T) and not ops.dense_x3_supported(C, T + 1) and not ops.dense_x3_supported(17, T)
    packs = [ops.pack_dense_cc(w.cuda(), None, b.cuda(), x3=True) for w, b in zip(ws, bs)]
    outs = ops.dense_cc([x.cuda() for x in xs], packs, x3=True)
    for x, w, b, o in zip(xs, ws, bs, outs):
        _close(o, F.conv1d(x.double(), w.double(), b.double()).float(), 2e-5)
    sc, res = seeded((C,), 51), seeded((B, C, T), 52)
    pk = ops.pack_dense_cc(ws[0].cuda(), sc.cuda(), (bs[0] * sc).cuda(), x3=True)
    (o,) = ops.dense_cc([xs[1].cuda()], [pk], [res.cuda()], x3=True)
    ref = res.double() + sc.double()[None, :, None] * F.conv1d(xs[1].double(), ws[0].double(), bs[0].double())
    _close(o, ref.float(), 2e-5)
    x = seeded((B, C, T), 61)
    dws = [seeded((C, 1, 3), 62 + i) * 0.6 for i in range(3)]
    gs = [1.0 + 0.3 * seeded((C,), 65 + i) for i in range(3)]
    be = [0.2 * seeded((C,), 68 + i) for i in range(3)]
    table = ops.pack_qkv_table(dws[0].cuda(), dws[1].cuda(), dws[2].cuda(), gs[0].cuda(), be[0].cuda(), gs[1].cuda(),
                               be[1].cuda(), gs[2].cuda(), be[2].cuda())
    outs = ops.qkv_front(x.cuda(), table, packs, eps, x3=True)
    for i in range(3):
        d = F.conv1d(x.double(), dws[i].double(), None, 1, 1, 1, C)
        r = d - d.mean(1, keepdim=True)
        ln = r / torch.sqrt((r * r).mean(1, keepdim=True) + eps) * gs[i].double()[None, :, None] + be[i].double()[None, :, None]
        _close(outs[i], F.conv1d(ln, ws[i].double(), bs[i].double()).float(), 2e-5)


@pytest.mark.parametrize("cin,cout,res,relu", [(256, 64, False, True), (64, 256, True, True), (128, 256, False, True),
                                                (64, 64, False, False), (256, 100, True, False), (96, 48, False, False),
                                                (192, 96, True, True), (40, 20, False, True), (48, 17, False, False), (51, 80, True, True), (17, 24, False, False)])
@pytest.mark.parametrize("hw", [(8, 8), (25, 10), (24, 18)])
def test_pointwise_x3_matches_fp64(cin, cout, res, relu, hw):
    """csrc/pointx.hip (HRNet layer1's 1x1 convs, model/HRNet.py:551-571, with the BatchNorm folded into scale / shift) vs
    fp64 on channel-slice views; 2e-5 of the output range like the other split-bf16 kernels.  25 x 10: a ragged last
    workgroup; Cout = 100: a partial last 16-row tile and a partial last weight block."""
    B, (h, w) = 3, hw
    assert ops.pointwise_x3_supported(cin, cout, h * w) and not ops.pointwise_x3_supported(384, cout, h * w)
    assert not ops.pointwise_x3_supported(cin, 260, h * w) and not ops.pointwise_x3_supported(cin, cout, h * w + 1)
    assert not ops.pointwise_x3_supported(8, cout, h * w)
    xt = seeded((B, cin + 24, h, w), 71)                   # the input is channels [16, 16 + cin) of a wider tensor
    wt, sc, sh = seeded((cout, cin), 72) / cin ** 0.5, 1.0 + 0.3 * seeded((cout,), 73), seeded((cout,), 74)
    rt = seeded((B, cout + 8, h, w), 75)                   # residual: channels [8, 8 + cout)
    ot = torch.full((B, cout + 5, h, w), 7.0)              # output: channels [2, 2 + cout); the rest must stay untouched
    x, r = xt[:, 16:16 + cin].double(), rt[:, 8:8 + cout].double()
    ref = torch.einsum("oc,bchw->bohw", wt.double(), x) * sc.double()[None, :, None, None] + sh.double()[None, :, None, None]
    if res:
        ref = ref + r
    if relu:
        ref = ref.clamp_min(0)
    pk = ops.pack_pointwise_x3(wt.cuda(), sc.cuda(), sh.cuda())
    od = ot.cuda()
    ops.pointwise_x3(ops.View(xt.cuda(), 16, cin), pk, ops.View(od, 2, cout), ops.View(rt.cuda(), 8, cout) if res else None, relu)
    _close(od[:, 2:2 + cout], ref.float(), 2e-5)
    assert bool((od[:, :2] == 7).all()) and bool((od[:, 2 + cout:] == 7).all())
    # no scale / shift: identity epilogue
    pk1 = ops.pack_pointwise_x3(wt.cuda())
    o1 = torch.empty(B, cout, h, w, device="cuda")
    ops.pointwise_x3(ops.View(xt.cuda(), 16, cin), pk1, ops.View(o1), None, False)
    _close(o1, torch.einsum("oc,bchw->bohw", wt.double(), x).float(), 2e-5)


@pytest.mark.parametrize("b,frames,h,w,cout", [(2, 5, 32, 24, 64), (3, 5, 30, 40, 64), (1, 7, 17, 15, 40), (2, 1, 16, 16, 64),
                                               (16, 7, 96, 128, 64)])
def test_stem_conv_x3_matches_fp64(b, frames, h, w, cout):
    """csrc/stem.hip (HRNet's conv1 + bn1 + relu, model/HRNet.py:33-36, on the frames of the clip, model/OTPose.py:317) against
    F.conv2d in fp64 on the re-arranged frames; odd sizes: the last row / column of taps reads the padding."""
    clip = seeded((b, 3 * frames, h, w), 91)
    wt, sc, sh = seeded((cout, 3, 3, 3), 92) * 0.3, 1.0 + 0.3 * seeded((cout,), 93), seeded((cout,), 94)
    assert ops.stem_conv_x3_supported(b, frames, h, w, cout)
    fr = clip.view(b, frames, 3, h, w).transpose(0, 1).reshape(frames * b, 3, h, w).double()       # frame n = f * B + b
    ref = F.conv2d(fr, wt.double(), None, 2, 1) * sc.double()[None, :, None, None] + sh.double()[None, :, None, None]
    out = ops.stem_conv_x3(clip.cuda(), ops.pack_stem_conv_x3(wt.cuda(), sc.cuda(), sh.cuda()), cout, frames)
    _close(out, ref.clamp_min(0).float(), 2e-5)
    assert not ops.stem_conv_x3_supported(b, frames, 30, 42, cout)        # Wo = 21: not a multiple of 4


@pytest.mark.parametrize("cin,cout", [(256, 64), (64, 64), (64, 96)])
@pytest.mark.parametrize("hw", [(8, 8), (25, 12), (24, 18)])
def test_pointwise_x3_s8_matches_fp64_and_feeds_the_s8_conv(cin, cout, hw):
    """csrc/pointx.hip writing S8 records (a Bottleneck's conv1 -> conv2, model/HRNet.py:551-571): the unpacked image against
    fp64 (hi + lo carry 16 mantissa bits: 2e-5 of the range), and the records as input of otp_conv3x3_s8 against the same conv
    on the records s8_pack makes from the fp32 result of otp_pointwise_x3 - bit-identical images, so bit-identical outputs."""
    B, (h, w) = 3, hw
    assert ops.pointwise_x3_s8_supported(cin, cout, h * w) and not ops.pointwise_x3_s8_supported(128, cout, h * w)
    assert not ops.pointwise_x3_s8_supported(cin, 48, h * w)
    xt = seeded((B, cin + 8, h, w), 81)
    wt, sc, sh = seeded((cout, cin), 82) / cin ** 0.5, 1.0 + 0.3 * seeded((cout,), 83), seeded((cout,), 84)
    x = xt[:, 8:8 + cin].double()
    ref = (torch.einsum("oc,bchw->bohw", wt.double(), x) * sc.double()[None, :, None, None] + sh.double()[None, :, None, None]).clamp_min(0)
    xv = ops.View(xt.cuda(), 8, cin)
    s8 = ops.pointwise_x3_s8(xv, ops.pack_pointwise_x3_s8(wt.cuda(), sc.cuda(), sh.cuda()), cout, relu=True)
    _close(ops.s8_unpack(s8, B, cout, h, w), ref.float(), 2e-5)
    # the fp32 route: same arithmetic, then s8_pack
    o = torch.empty(B, cout, h, w, device="cuda")
    ops.pointwise_x3(xv, ops.pack_pointwise_x3(wt.cuda(), sc.cuda(), sh.cuda()), ops.View(o), None, True)
    assert torch.equal(s8, ops.s8_pack(o))
    # + an fp32 NCHW residual (a channel slice) before the ReLU: a Bottleneck's conv3 read as S8 only (layer1 -> transition1);
    # the same bits as the fp32 route with the residual, packed
    rt = seeded((B, cout + 4, h, w), 85).cuda()
    rv = ops.View(rt, 4, cout)
    s8r = ops.pointwise_x3_s8(xv, ops.pack_pointwise_x3_s8(wt.cuda(), sc.cuda(), sh.cuda()), cout, relu=True, res=rv)
    o2 = torch.empty(B, cout, h, w, device="cuda")
    ops.pointwise_x3(xv, ops.pack_pointwise_x3(wt.cuda(), sc.cuda(), sh.cuda()), ops.View(o2), rv, True)
    assert torch.equal(s8r, ops.s8_pack(o2))
    refr = (torch.einsum("oc,bchw->bohw", wt.double(), x) * sc.double()[None, :, None, None] + sh.double()[None, :, None, None]
            + rt[:, 4:4 + cout].cpu().double()).clamp_min(0)
    _close(ops.s8_unpack(s8r, B, cout, h, w), refr.float(), 2e-5)


@pytest.mark.parametrize("T", [32, 250, 1152])
def test_qkv_front_matches_fp64(T):
    """csrc/dense.hip qkv_front vs MaskedMHCA's depthwise conv -> channel LayerNorm -> pointwise projection chain
    (model/blocks.py:406-419, LayerNorm of :95-110) in fp64, and vs the two-launch path it replaces."""
    B, C, eps = 2, 136, 1e-5
    x = seeded((B, C, T), 61)
    dws = [seeded((C, 1, 3), 62 + i) * 0.6 for i in range(3)]
    gs = [1.0 + 0.3 * seeded((C,), 65 + i) for i in range(3)]
    bs = [0.2 * seeded((C,), 68 + i) for i in range(3)]
    ws = [seeded((C, C, 1), 71 + i) / C ** 0.5 for i in range(3)]
    cb = [seeded((C,), 74 + i) for i in range(3)]
    table = ops.pack_qkv_table(dws[0].cuda(), dws[1].cuda(), dws[2].cuda(), gs[0].cuda(), bs[0].cuda(), gs[1].cuda(),
                               bs[1].cuda(), gs[2].cuda(), bs[2].cuda())
    packs = [ops.pack_dense_cc(w.cuda(), None, b.cuda()) for w, b in zip(ws, cb)]
    outs = ops.qkv_front(x.cuda(), table, packs, eps)
    for i in range(3):
        d = F.conv1d(x.double(), dws[i].double(), None, 1, 1, 1, C)
        mu = d.mean(1, keepdim=True)
        r = d - mu
        ln = r / torch.sqrt((r * r).mean(1, keepdim=True) + eps) * gs[i].double()[None, :, None] + bs[i].double()[None, :, None]
        ref = F.conv1d(ln, ws[i].double(), cb[i].double())
        _close(outs[i], ref.float(), 3e-6)
    qn, kn, vn = ops.dwconv_ln3(x.cuda(), [t.cuda().contiguous() for t in dws], [t.cuda() for t in gs],
                                [t.cuda() for t in bs], 1, eps)
    two = ops.dense_cc([qn, kn, vn], packs)
    for a, b in zip(outs, two):
        _close(a, b.cpu(), 3e-6)


@pytest.mark.parametrize("nlow", [2, 3])
def test_upsample_add_multi_equals_the_chain(nlow):
    """One pass over the high-resolution tensor == chaining otp_upsample_add term by term (same order: bit-identical), and
    both equal F.interpolate(nearest) sums (HRNet fuse rows, model/HRNet.py:426-439, 488-494)."""
    n, c, h, w = 2, 5, 16, 24
    res = seeded((n, c, h, w), 81)
    lows = [seeded((n, c, h // f, w // f), 82 + i) for i, f in enumerate((2, 4, 8)[:nlow])]
    ref = res.clone()
    chain = res.cuda()
    for i, low in enumerate(lows):
        ref = ref + F.interpolate(low, scale_factor=h // low.shape[2], mode="nearest")
        chain = ops.upsample_add(low.cuda(), chain, h // low.shape[2], relu=(i == nlow - 1))
    out = ops.upsample_add_multi([t.cuda() for t in lows], res.cuda(), relu=True)
    assert torch.equal(out, chain)
    _close(out, F.relu(ref), 1e-6)


def test_joints_losses_match_golden_and_oracle(golden):
    """JointsMSE_OHKMMSELoss / JointMSELoss (model/loss.py:95-182): values vs reference-generated goldens, gradients and
    the use_target_weight=False / effective_num_joints forms vs the oracle."""
    from otpose_amd import train as TR
    g = golden("losses")
    s, gt, w = g["s"], g["g"], g["w"]
    r = ops.joints_ohkm_mse_loss(s.cuda(), gt.cuda(), w.cuda(), with_grad=True)
    for k in ("ohkm_loss", "mse_loss", "final_loss"):
        _close(r[k].reshape(()), g["ohkm_" + k].reshape(()), 1e-5)
    sd = s.double().requires_grad_(True)
    ref = O.joints_ohkm_mse_loss(sd, gt.double(), w.double())
    ref["final_loss"].backward()
    _close(r["grad_output"], sd.grad.float(), 1e-6)
    v, gv = ops.joint_mse_loss(s.cuda(), gt.cuda(), w.cuda(), with_grad=True)
    _close(v.reshape(()), g["jmse"].reshape(()), 1e-5)
    sd = s.double().requires_grad_(True)
    O.joint_mse_loss(sd, gt.double(), w.double()).backward()
    _close(gv, sd.grad.float(), 1e-6)
    # no target weight, effective_num_joints = 13 (mse_loss / plain loss scale only)
    ones = torch.ones_like(w)
    r2 = ops.joints_ohkm_mse_loss(s.cuda(), gt.cuda(), None, effective_num_joints=13, topk=5)
    ref2 = O.joints_ohkm_mse_loss(s, gt, ones, topk=5)
    _close(r2["ohkm_loss"].reshape(()), ref2["ohkm_loss"].reshape(()), 1e-5)
    _close(r2["mse_loss"].reshape(()), (ref2["mse_loss"] * 17 / 13).reshape(()), 1e-5)
    _close(r2["final_loss"].reshape(()), ref2["final_loss"].reshape(()), 1e-5)
    v2 = ops.joint_mse_loss(s.cuda(), gt.cuda(), None, effective_num_joints=13)
    _close(v2.reshape(()), (O.joint_mse_loss(s, gt, ones) * 17 / 13).reshape(()), 1e-5)
    # the nn.Module mirrors + build_loss (loss.py:185-189), autograd through the fused gradient
    crit = TR.build_loss({"LOSS": {"NAME": "MSELOSS_OHKM", "USE_TARGET_WEIGHT": True}})
    assert isinstance(crit, TR.JointsMSE_OHKMMSELoss)
    assert isinstance(TR.build_loss({"LOSS": {"NAME": "ST_OHKW_MSELoss", "USE_TARGET_WEIGHT": True}}), TR.ST_OHKW_MSELoss)
    sc = s.cuda().requires_grad_(True)
    d = crit(sc, gt.cuda(), w.cuda())
    (3.0 * d["final_loss"]).backward()
    _close(sc.grad, 3.0 * r["grad_output"].cpu(), 1e-6)
    sc = s.cuda().requires_grad_(True)
    TR.JointMSELoss(True)(sc, gt.cuda(), w.cuda(), margin=None).backward()
    _close(sc.grad, gv.cpu(), 1e-6)
    st = TR.ST_OHKW_MSELoss(True)(s.cuda(), g["t"].cuda(), gt.cuda(), w.cuda())
    for k in ("ohkm_loss_s", "mse_loss_s", "final_loss"):
        _close(st[k].reshape(()), g["st_" + k].reshape(()), 1e-5)


@pytest.mark.parametrize("case", [  # N, Cin, H, W, Cout
    (2, 48, 24, 18, 48),       # even sizes, one M tile, several tile blocks per image
    (2, 20, 12, 9, 100),       # odd width (last tile column half empty), ragged Cin chunk and ragged last M tile
    (1, 8, 6, 6, 16),          # one partly filled tile block
    (3, 96, 10, 14, 96),       # odd tile-row count inside blocks, two M tiles
    (1, 13, 96, 72, 17),       # the full heat-map size: 4 x 12 tile rectangles (2-D blocks), ragged Cin / Cout
    (2, 40, 16, 12, 48),       # 8 x 6 tile rectangles (the 48x36 branch shape family)
    (1, 16, 8, 96, 32),        # 4 x 12 rectangles on a wide, short map; Cout below one 48-row tile
    (2, 24, 6, 8, 20),         # 3 x 4 tiles: smaller than any rectangle -> row-major 32-tile run
])
def test_conv2d_winograd_matches_conv2d(case):
    """Winograd F(2x2,3x3) kernel vs F.conv2d (3x3, stride 1, pad 1) with the fused epilogue and channel-sliced views."""
    n, cin, h, w, cout = case
    x = seeded((n, cin, h, w), 1)
    wt = seeded((cout, cin, 3, 3), 2, 1.0 / math.sqrt(9 * cin))
    sc, sh = 1 + 0.1 * seeded((cout,), 3), seeded((cout,), 4)
    ref = F.conv2d(x, wt, None, 1, 1) * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)
    res = seeded(ref.shape, 5)
    _close(ops.conv2d_wino(x.cuda(), wt.cuda()), F.conv2d(x, wt, None, 1, 1))
    _close(ops.conv2d_wino(x.cuda(), wt.cuda(), sc.cuda(), sh.cuda(), act=ops.ACT_RELU, res=res.cuda()), F.relu(ref + res))
    # channel-sliced input / output / residual views
    big_in = torch.zeros(n, cin + 5, h, w)
    big_in[:, 3:3 + cin] = x
    big_out = torch.full((n, cout + 4, h, w), 7.0)
    big_res = torch.zeros(n, cout + 2, h, w)
    big_res[:, 2:] = res
    bi, bo, br = big_in.cuda(), big_out.cuda(), big_res.cuda()
    iv, ov, rv = ops.View(bi, 3, cin), ops.View(bo, 1, cout), ops.View(br, 2, cout)
    d = ops.conv_desc(iv, ov, cout, 3, 3, 1, 1, 1, ops.ACT_NONE, None, rv, 1)
    assert ops.wino_supported(d)
    ops.conv2d_wino_launch(iv, ops.pack_wino_weight(wt.cuda()), sc.cuda(), sh.cuda(), ov, d, rv)
    _close(bo[:, 1:1 + cout], ref + res)
    assert float((bo[:, 0] - 7).abs().max()) == 0 and float((bo[:, 1 + cout:] - 7).abs().max()) == 0


@pytest.mark.parametrize("C,stride", [(204, 2), (204, 1), (136, 2), (17, 2)])
def test_dwconv_ln3_every_width_and_stride_matches_float64(C, stride):
    """Depthwise k = 3 convs + channel LayerNorms of MaskedMHCA (model/blocks.py:359-381, 406-416) at the widths the engine launches
    them with: C = 204 (the 7-frame window's stride-2 blocks) runs on the wave-split kernel since round 5, not the generic one."""
    B, T, eps = 2, 250, 1e-5
    x = seeded((B, C, T), 91)
    dws = [seeded((C, 1, 3), 92 + i) * 0.6 for i in range(3)]
    gs = [1.0 + 0.3 * seeded((C,), 95 + i) for i in range(3)]
    bs = [0.2 * seeded((C,), 98 + i) for i in range(3)]
    outs = ops.dwconv_ln3(x.cuda(), [t.cuda().contiguous() for t in dws], [t.cuda() for t in gs], [t.cuda() for t in bs], stride, eps)
    for i in range(3):
        d = F.conv1d(x.double(), dws[i].double(), None, stride, 1, 1, C)
        r = d - d.mean(1, keepdim=True)
        ref = r / torch.sqrt((r * r).mean(1, keepdim=True) + eps) * gs[i].double()[None, :, None] + bs[i].double()[None, :, None]
        _close(outs[i], ref.float(), 3e-6)
