"""Whole OTPose forward through the HIP engine vs the reference-generated goldens (<= 1e-3 max-abs,
BASELINE.json north_star) and vs the CPU oracle."""
import pytest
import torch

from otpose_amd import OTPose, cfg1, cfg2, tiny_cfg
from otpose_amd import synthetic as S

pytestmark = pytest.mark.gpu
NAMES = ("output", "rough", "intersection", "prev_b", "context", "squeezed", "total_b")
TOL = 1e-3     # max-abs on heat-maps, fp32 (BASELINE.json north_star)


def _run(cfg, batch):
    m = OTPose(cfg)
    S.fill_synthetic_(m)
    m = m.cuda().eval()
    x, margin = S.synthetic_clip(batch, cfg.MODEL.IMAGE_SIZE)
    with torch.no_grad():
        outs = m(x.cuda(), margin=margin.cuda())
    torch.cuda.synchronize()
    return m, [o.cpu() for o in outs]


def _check(outs, g):
    worst = {}
    for n, o in zip(NAMES, outs):
        assert o.shape == g[n].shape, n
        err = float((o - g[n]).abs().max())
        scale = max(1.0, float(g[n].abs().max()))
        worst[n] = err
        # heat-maps: absolute 1e-3; the product maps (intersection, ...) reach 1e1-1e2, so scale them
        assert err <= TOL * scale, f"{n}: max abs err {err} (max |ref| {scale})"
    assert worst["output"] <= TOL and worst["rough"] <= TOL
    return worst


def test_e2e_tiny_matches_reference_golden(golden):
    _, outs = _run(tiny_cfg(8, (64, 96)), 2)
    print(_check(outs, golden("e2e_tiny")))


def test_e2e_tiny_exact_fp32_kernels(golden, monkeypatch):
    """OTPOSE_CONV_MATH=f32: the exact-fp32 MFMA kernels of round 1 (Winograd / direct convs, f32 MLP and projections,
    offset / mask convs and DCN as separate launches) stay selectable and green."""
    monkeypatch.setenv("OTPOSE_CONV_MATH", "f32")
    m, outs = _run(tiny_cfg(8, (64, 96)), 2)
    assert not m._engine.use_x3 and not m._engine.use_dcn_fused
    worst = _check(outs, golden("e2e_tiny"))
    assert worst["output"] <= 1e-4


def test_e2e_tiny_unfused_warping_head(golden, monkeypatch):
    """OTPOSE_DCN_FUSED=0: offset / mask convs and DCN gathers as separate launches (the path shapes outside
    otp_dcn_fused_supported take) give the same heat-maps as the fused launch."""
    _, fused = _run(tiny_cfg(8, (64, 96)), 2)
    monkeypatch.setenv("OTPOSE_DCN_FUSED", "0")
    m, outs = _run(tiny_cfg(8, (64, 96)), 2)
    assert m._engine.use_x3 and not m._engine.use_dcn_fused
    _check(outs, golden("e2e_tiny"))
    assert float((outs[0] - fused[0]).abs().max()) <= 2e-4 * float(fused[0].abs().max())


def test_e2e_cfg1_matches_reference_golden(golden):
    _, outs = _run(cfg1(), 1)
    print(_check(outs, golden("e2e_cfg1")))


def test_e2e_cfg2_clip_matches_reference_golden(golden):
    _, outs = _run(cfg2(), 1)
    print(_check(outs, golden("e2e_cfg2_b1")))


@pytest.mark.parametrize("switch", ["OTPOSE_S8_STRIDE2", "OTPOSE_FUSE_UPSAMPLE", "OTPOSE_POINTX", "OTPOSE_S8_RESIDUAL",
                                    "OTPOSE_S8_LAZY_NCHW", "OTPOSE_T1_S8", "OTPOSE_CHAIN_MODULES", "OTPOSE_STEM_X3", "OTPOSE_L1_S8",
                                    "OTPOSE_X3_WSCALE", "OTPOSE_S8", "OTPOSE_UP_ANY_WIDTH", "OTPOSE_F32_TAIL=1", "OTPOSE_F32_TAIL=2"])
def test_e2e_cfg1_with_each_engine_switch_off(golden, monkeypatch, switch):
    """Every kernel-family switch of the engine, one at a time, against the reference-generated golden.  Since round 4 a branch
    output's NCHW tensor is written only when a consumer asks for it (engine._needs_nchw): each switch moves some consumer from
    the S8 records back to the fp32 tensor, and a consumer that forgot to ask would read an unwritten buffer here.
    OTPOSE_F32_TAIL=n switches the S8 / pointx / split-product routes OFF for the last n modules of stage 4 and the
    final layer, in the middle of modules chained stream by stream: the same lazily written tensors, from the other side."""
    name, _, value = switch.partition("=")
    monkeypatch.setenv(name, value or "0")
    m, outs = _run(cfg1(), 1)
    if name == "OTPOSE_F32_TAIL":
        assert m._engine.f32_tail == int(value)
    _check(outs, golden("e2e_cfg1"))


def test_graph_replay_and_eager_agree_and_batch_rows_are_independent(monkeypatch):
    cfg = tiny_cfg(8, (64, 96))
    m, outs = _run(cfg, 2)
    x, margin = S.synthetic_clip(2, cfg.MODEL.IMAGE_SIZE)
    with torch.no_grad():
        again = [o.clone() for o in m(x.cuda(), margin=margin.cuda())]      # graph replay
    # the same launches issued eagerly, by an engine that never captures (a second model: tearing a captured graph down
    # and launching eagerly on its streams afterwards crashed the HIP runtime now and then)
    monkeypatch.setenv("OTPOSE_HIP_GRAPH", "0")
    me, eager = _run(cfg, 2)
    assert me._engine.graph is None and not me._engine.use_graph
    monkeypatch.delenv("OTPOSE_HIP_GRAPH")
    for a, b, c in zip(outs, again, eager):
        assert torch.equal(a, b.cpu()) and torch.equal(a, c)
    # clips shard by batch: sample 1 alone gives the same heat-maps (eval mode, no cross-sample op)
    m2 = OTPose(cfg)
    S.fill_synthetic_(m2)
    m2 = m2.cuda().eval()
    with torch.no_grad():
        solo = m2(x[1:2].cuda(), margin=margin[1:2].cuda())
    assert float((solo[0].cpu() - outs[0][1:2]).abs().max()) <= 1e-5


def test_forward_frames_u8_equals_forward_of_normalised_clip():
    """OTPose.forward_frames(uint8 crops) == OTPose.forward(ToTensor + Normalize + concat of the same crops)."""
    from oracle import otpose_oracle as O
    cfg = tiny_cfg(8, (64, 96))
    model = OTPose(cfg)
    S.fill_synthetic_(model)
    model = model.cuda().eval()
    gen = torch.Generator().manual_seed(3)
    frames = torch.randint(0, 256, (2, 5, 96, 64, 3), generator=gen, dtype=torch.uint8)
    margin = torch.tensor([[1.0, 1.0, 2.0, 2.0], [0.0, 1.0, 0.0, 2.0]])
    with torch.no_grad():
        a = [t.clone() for t in model.forward_frames(frames.cuda(), margin.cuda())]
        b = model(O.frames_to_clip(frames).cuda(), margin=margin.cuda())
    for u, v in zip(a, b):
        assert torch.equal(u, v)


def test_parallel_stream_graph_equals_single_stream(monkeypatch):
    """The engine's parallel graph branches (HRNet branches / fuse rows / temporal encoders on side streams) and the
    Winograd / fused-encoder routing change scheduling and kernels only: single-stream replay is bit-identical, the
    direct-kernel and generic-kernel engines agree to fp32 rounding."""
    cfg = tiny_cfg(8, (64, 96))
    x, margin = S.synthetic_clip(2, cfg.MODEL.IMAGE_SIZE)
    outs = {}
    for key, env in (("default", {}), ("single", {"OTPOSE_STREAMS": "0"}), ("direct", {"OTPOSE_WINOGRAD": "0"}),
                     ("generic", {"OTPOSE_FUSED_MLP": "0", "OTPOSE_DENSE_CC": "0"}),
                     ("unfused", {"OTPOSE_QKV_FRONT": "0", "OTPOSE_FUSE_SHORTCUT": "0", "OTPOSE_FUSE_UPSAMPLE": "0"})):
        for k in ("OTPOSE_STREAMS", "OTPOSE_WINOGRAD", "OTPOSE_FUSED_MLP", "OTPOSE_DENSE_CC", "OTPOSE_QKV_FRONT",
                  "OTPOSE_FUSE_SHORTCUT", "OTPOSE_FUSE_UPSAMPLE"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        m = OTPose(cfg)
        S.fill_synthetic_(m)
        m = m.cuda().eval()
        with torch.no_grad():
            outs[key] = [o.clone() for o in m(x.cuda(), margin=margin.cuda())]
            again = [o.clone() for o in m(x.cuda(), margin=margin.cuda())]          # graph replay
        for a, b in zip(outs[key], again):
            assert torch.equal(a, b)
    for a, b in zip(outs["default"], outs["single"]):
        assert torch.equal(a, b)
    for other in ("direct", "generic", "unfused"):
        # generic: the temporal encoders' dense layers on conv_win_kernel instead of csrc/mlp.hip / csrc/dense.hip;
        # unfused: dwconv_ln3 + otp_dense_cc instead of qkv_front, layer1 shortcut as its own conv + residual
        for a, b in zip(outs["default"], outs[other]):
            assert float((a - b).abs().max()) <= 1e-4 * max(1.0, float(b.abs().max())), other


def test_headline_batch_agrees_with_the_single_clip_the_golden_pins(golden):
    """BASELINE configs[1] at its full size (16 clips x 5 x 384x288, HRNet-W48): the golden vector pins ONE clip of this
    configuration (test_e2e_cfg2_clip_matches_reference_golden); the 16-clip forward - the one bench.py times - takes other
    launch shapes (workgroup rounds of the fused warping head, attention splits, tile boundaries that straddle frames).  Clip 0
    of the batch is made that same clip, so its outputs must agree with the golden within the same tolerance, and clips must
    not see each other: clip 7 alone reproduces its rows of the batch."""
    cfg = cfg2()
    x, margin = S.synthetic_clip(16, cfg.MODEL.IMAGE_SIZE)
    x1, m1_ = S.synthetic_clip(1, cfg.MODEL.IMAGE_SIZE)
    x[0], margin[0] = x1[0], m1_[0]
    m = OTPose(cfg)
    S.fill_synthetic_(m)
    m = m.cuda().eval()
    with torch.no_grad():
        outs = [o.cpu() for o in m(x.cuda(), margin=margin.cuda())]
        solo = [o.cpu() for o in m(x[7:8].cuda(), margin=margin[7:8].cuda())]      # rebuilds the engine for batch 1
    g = golden("e2e_cfg2_b1")
    rows = lambda t, k: t[k:k + 1] if t.shape[0] == 16 else t[k::16]      # `rough` stacks frames: (5 * 16, J, h, w)   # noqa: E731
    print(_check([rows(o, 0) for o in outs], g))          # heat-maps (output, rough): 1e-3 ABSOLUTE, like the 1-clip tests
    for n, a, b in zip(NAMES, solo, outs):
        r = rows(b, 7)
        assert float((a - r).abs().max()) <= 2e-5 * max(1.0, float(r.abs().max())), n


def test_headline_batch_all_16_clips_match_the_oracle():
    """EVERY clip of the batch bench.py times (BASELINE configs[1]: 16 clips x 5 x 384x288, HRNet-W48,
    reference path model/OTPose.py:307-394) against the CPU oracle on the same 16 clips (four oracle forwards of four clips;
    the oracle itself is pinned by the reference-generated goldens, tests/test_oracle_golden.py).  The default eval arithmetic
    is split-half products with fp32 accumulation (csrc/common.h), whose error is data dependent, so the bound is checked per
    clip and the per-clip maxima are printed as a distribution.  The contract of BASELINE.json is 1e-3 ABSOLUTE on the heat-maps;
    what the build delivers since the weights are stored with their power of two (ops.x3_weight_exponent; rounds 2-3 and the
    first half of round 4 carried 5e-4 on `output` and, through the ill-conditioned 17-channel flow encoder of
    model/OTPose.py:331-335, 2.1e-2 on `context` of clip 7) is within 5x of what the fp32 ORACLE itself differs from its fp64 run
    (profiles/r04_headline_parity_probe.txt: output 4.5e-5 against 9.3e-6, rough 5.8e-6 against 1.1e-6, context 8.8e-4 against
    9.9e-5 on clip 7), so the test holds it to ~3x those measurements: output <= 1.5e-4, rough <= 3e-5, every other output <=
    3e-4 of max(1, range)."""
    from oracle import otpose_oracle as O
    cfg = cfg2()
    x, margin = S.synthetic_clip(16, cfg.MODEL.IMAGE_SIZE)
    margin[5] = torch.tensor([0.0, 1.0, 0.0, 2.0])        # sequence borders inside the batch
    margin[11] = torch.tensor([1.0, 0.0, 2.0, 0.0])
    m = OTPose(cfg)
    S.fill_synthetic_(m)
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    m = m.cuda().eval()
    with torch.no_grad():
        outs = [o.cpu() for o in m(x.cuda(), margin=margin.cuda())]
    rows = lambda t, lo, hi: t[lo:hi] if t.shape[0] == 16 else torch.cat([t[f * 16 + lo:f * 16 + hi] for f in range(5)])   # noqa: E731
    per_clip = {n: [] for n in NAMES}
    for lo in range(0, 16, 4):
        with torch.no_grad():
            ref = O.otpose_forward(sd, cfg, x[lo:lo + 4], margin[lo:lo + 4])
        for n, o, r in zip(NAMES, outs, ref):
            ob = rows(o, lo, lo + 4)
            assert ob.shape == r.shape, n
            for k in range(4):
                a = ob[k:k + 1] if o.shape[0] == 16 else ob[k::4]
                b = r[k:k + 1] if o.shape[0] == 16 else r[k::4]
                per_clip[n].append(float((a - b).abs().max()))
                scale = max(1.0, float(b.abs().max()))
                assert per_clip[n][-1] <= 3e-4 * scale, (n, lo + k, per_clip[n][-1], scale)
    for n in NAMES:
        v = per_clip[n]
        print("%-13s per-clip max |delta| vs oracle: min %.2e median %.2e max %.2e" % (n, min(v), sorted(v)[8], max(v)))
    assert max(per_clip["output"]) <= 1.5e-4 and max(per_clip["rough"]) <= 3e-5


@pytest.mark.parametrize("ws,xs", [(777, 31), (4242, 32)])
def test_e2e_cfg2_two_more_weight_seeds(golden, ws, xs):
    """cfg2 under two more weight draws against reference-generated goldens
    (tests/golden/make_golden.py::gen_e2e_cfg2_seeds): heat-maps <= 1e-3 ABSOLUTE."""
    g = golden("e2e_cfg2_seeds")
    cfg = cfg2()
    m = OTPose(cfg)
    S.fill_synthetic_(m, ws, S.gains_for(cfg))
    m = m.cuda().eval()
    x, margin = S.synthetic_clip(1, cfg.MODEL.IMAGE_SIZE, seed=xs)
    with torch.no_grad():
        outs = [o.cpu() for o in m(x.cuda(), margin=margin.cuda())]
    tag = f"cfg2_w{ws}_x{xs}"
    errs = {"output": float((outs[0] - g[tag + "_output"]).abs().max()),
            "rough": float((outs[1][:1] - g[tag + "_rough_cur"]).abs().max()),
            "context": float((outs[4] - g[tag + "_context"]).abs().max())}
    print(tag, errs, "max |output| %.3g" % float(g[tag + "_output"].abs().max()))
    assert errs["output"] <= TOL and errs["rough"] <= TOL, (tag, errs)
    assert errs["context"] <= TOL * max(1.0, float(g[tag + "_context"].abs().max())), (tag, errs)
    for o, mx in zip(outs, g[tag + "_absmax"].tolist()):
        assert abs(float(o.abs().max()) - mx) <= TOL * max(1.0, mx)


@pytest.mark.parametrize("ws", [S.WEIGHT_SEED, 777])
def test_e2e_cfg1_three_input_seeds_two_weight_seeds(golden, ws):
    """The split-product eval path against reference-generated goldens of cfg1 for 3 input seeds x 2 weight seeds
    (tests/golden/make_golden.py::gen_e2e_seeds): heat-maps <= 1e-3 ABSOLUTE everywhere."""
    g = golden("e2e_seeds")
    cfg = cfg1()
    m = OTPose(cfg)
    S.fill_synthetic_(m, ws, S.gains_for(cfg))
    m = m.cuda().eval()
    for xs in (S.INPUT_SEED, 11, 12):
        x, margin = S.synthetic_clip(1, cfg.MODEL.IMAGE_SIZE, seed=xs)
        with torch.no_grad():
            outs = [o.cpu() for o in m(x.cuda(), margin=margin.cuda())]
        tag = f"cfg1_w{ws}_x{xs}"
        errs = {"output": float((outs[0] - g[tag + "_output"]).abs().max()),
                "rough": float((outs[1][:1] - g[tag + "_rough_cur"]).abs().max()),
                "context": float((outs[4] - g[tag + "_context"]).abs().max())}
        print(tag, errs)
        assert errs["output"] <= TOL and errs["rough"] <= TOL, (tag, errs)
        assert errs["context"] <= TOL * max(1.0, float(g[tag + "_context"].abs().max())), (tag, errs)
        for o, mx in zip(outs, g[tag + "_absmax"].tolist()):
            assert abs(float(o.abs().max()) - mx) <= TOL * max(1.0, mx)


def test_e2e_cfg2_two_clips_with_a_sequence_border(golden):
    """cfg2 (384x288, W48), 2 clips, the second at a sequence border (margin row [0, 1, 0, 2]: the prev / pprev frames are
    copies of the current one and their heat-maps are NOT penalised, OTPose.py:339-346)."""
    g = golden("e2e_seeds")
    cfg = cfg2()
    m = OTPose(cfg)
    S.fill_synthetic_(m)
    m = m.cuda().eval()
    x, margin = S.synthetic_clip(2, cfg.MODEL.IMAGE_SIZE, seed=21)
    margin[1] = torch.tensor([0.0, 1.0, 0.0, 2.0])
    with torch.no_grad():
        outs = [o.cpu() for o in m(x.cuda(), margin=margin.cuda())]
    errs = {n: float((outs[NAMES.index(n)] - g["cfg2_b2_" + n]).abs().max()) for n in ("output", "prev_b", "context")}
    print(errs)
    assert errs["output"] <= TOL and errs["prev_b"] <= TOL
    assert errs["context"] <= TOL * max(1.0, float(g["cfg2_b2_context"].abs().max()))
    for o, mx in zip(outs, g["cfg2_b2_absmax"].tolist()):
        assert abs(float(o.abs().max()) - mx) <= TOL * max(1.0, mx)


def test_forward_from_the_engines_own_input_buffers():
    """OTPose.input_buffers(): a caller that writes the batch into the engine's input tensors in place gets the same result as
    one that hands in its own tensors (which the engine copies)."""
    cfg = tiny_cfg(8, (64, 96))
    m = OTPose(cfg)
    S.fill_synthetic_(m)
    m = m.cuda().eval()
    x, margin = S.synthetic_clip(3, cfg.MODEL.IMAGE_SIZE)
    with torch.no_grad():
        ref = [o.clone() for o in m(x.cuda(), margin=margin.cuda())]
        xb, mb = m.input_buffers(3, "cuda")
        assert m._engine is not None and xb.shape == x.shape and mb.shape == margin.shape
        xb.copy_(torch.roll(x, 1, 0).cuda())
        mb.copy_(torch.roll(margin, 1, 0).cuda().float())
        rolled = m(xb, margin=mb)
        for a, b in zip(ref, rolled):
            if a.shape[0] == 3:
                assert torch.equal(torch.roll(a, 1, 0), b)
