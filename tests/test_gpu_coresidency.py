"""Kernels that share CUs must not change each other's results.  The engine runs the two temporal encoders (and the HRNet
branches) on parallel streams; in round 3 the batch-16 forward turned out to differ from replay to replay by up to 2e-3 on the
heat-maps because a workgroup of the attention's P.v kernel on the same CU corrupted an accumulator row of the LDS-DMA fed
projection kernel (csrc/densex.hip).  These tests hold the pair and the whole forward to bit-stability.

The cause, found at the end of round 3: a packed-fp32 instruction (``v_pk_fma_f32 ... op_sel:[0,1,1]``
in the epilogue of csrc/densex.hip's kernels) returns a wrong low half in lanes 48-63 when another wave of the same SIMD issues
MFMAs in the same cycles - tools/micro/pkfma_opsel_next_to_mfma.hip; the library is built without packed-fp32 instructions
(csrc/Makefile, tests/test_build_guard.py)."""
import pytest
import torch

from otpose_amd import OTPose, cfg2, ops
from otpose_amd import synthetic as S

pytestmark = pytest.mark.gpu


def test_projection_kernel_is_bit_stable_next_to_channel_attention():
    B, C, T = 16, 136, 6912
    g = torch.Generator().manual_seed(1)
    x, r = torch.randn(B, C, T, generator=g).cuda(), torch.randn(B, C, T, generator=g).cuda()
    w = (torch.randn(C, C, generator=g) / C ** 0.5).cuda()
    pk = ops.pack_dense_cc(w, (torch.rand(C, generator=g) + 0.5).cuda(), torch.randn(C, generator=g).cuda(), x3=True)
    out = torch.empty_like(x)
    q, k, v = (torch.randn(B, C, T, generator=g).cuda() for _ in range(3))
    s0, s1 = torch.cuda.Stream(), torch.cuda.Stream()

    def proj(st):
        ops.dense_cc([x], [pk], [r], [out], stream=st.cuda_stream, x3=True)

    proj(torch.cuda.current_stream())
    torch.cuda.synchronize()
    ref = out.clone()
    att_ref = ops.chan_attn(q, k, v, 2, 68 ** -0.5).clone()
    for it in range(12):
        out.zero_()
        torch.cuda.synchronize()
        with torch.cuda.stream(s1):
            a1 = ops.chan_attn(q, k, v, 2, 68 ** -0.5)
            a2 = ops.chan_attn(q, k, v, 2, 68 ** -0.5)
        for _ in range(4):
            proj(s0)
        with torch.cuda.stream(s1):
            a3 = ops.chan_attn(q, k, v, 2, 68 ** -0.5)
        torch.cuda.synchronize()
        assert torch.equal(out, ref), it
        assert torch.equal(a1, att_ref) and torch.equal(a2, att_ref) and torch.equal(a3, att_ref), it


def test_qkv_front_and_projection_are_bit_stable_next_to_the_two_workgroup_mlp_launch():
    """The pair that exposed the hazard: the ln2 + MLP launch with two 80 KB workgroups per CU (its default form) on one
    stream, the q / k / v front end or the C -> C projection on another - 7-18 of 80 rounds differed before the fix
    (tools/encoder_pair_stress.py)."""
    B, C, HID, T = 16, 136, 544, 6912
    g = torch.Generator().manual_seed(3)
    rnd = lambda *s: torch.randn(*s, generator=g).cuda()   # noqa: E731
    xm = rnd(B, C, T)
    w1, w2 = rnd(HID, C, 1) / C ** 0.5, rnd(C, HID, 1) / HID ** 0.5
    b1, one, zero = rnd(HID) * 0.1, torch.ones(C).cuda(), torch.zeros(C).cuda()
    px = ops.pack_mlp_x3_weights(w1, b1, w2)
    mlp_out = torch.empty_like(xm)
    xa, res = rnd(B, C, T), rnd(B, C, T)
    pk = ops.pack_dense_cc(rnd(C, C, 1) / C ** 0.5, rnd(C), rnd(C), x3=True)
    proj_out = torch.empty_like(xa)
    packs = [ops.pack_dense_cc(rnd(C, C, 1) / C ** 0.5, None, rnd(C), x3=True) for _ in range(3)]
    dws = [rnd(C, 1, 3) * 0.6 for _ in range(3)]
    table = ops.pack_qkv_table(dws[0], dws[1], dws[2], one, zero, one, zero, one, zero)
    qkv_outs = [torch.empty_like(xa) for _ in range(3)]
    s0, s1 = torch.cuda.Stream(), torch.cuda.Stream()

    def mlp(st):
        ops.ln_mlp_x3(xm, one, zero, 1e-5, px, one, zero, out=mlp_out, stream=st.cuda_stream)

    def front(st):
        ops.qkv_front(xa, table, packs, 1e-5, outs=qkv_outs, stream=st.cuda_stream, x3=True)
        ops.dense_cc([xa], [pk], [res], [proj_out], stream=st.cuda_stream, x3=True)

    torch.cuda.synchronize()                        # operands and packed weights were made on the default stream
    mlp(s0)
    front(s1)
    torch.cuda.synchronize()
    refs = [o.clone() for o in [mlp_out, proj_out] + qkv_outs]
    for it in range(24):
        for o in [mlp_out, proj_out] + qkv_outs:
            o.zero_()
        torch.cuda.synchronize()
        for _ in range(3):
            front(s1)
            mlp(s0)
        torch.cuda.synchronize()
        for i, (a, b) in enumerate(zip([mlp_out, proj_out] + qkv_outs, refs)):
            assert torch.equal(a, b), (it, i)


def test_headline_batch_forward_is_bit_identical_across_replays():
    cfg = cfg2()
    x, margin = S.synthetic_clip(16, cfg.MODEL.IMAGE_SIZE)
    m = OTPose(cfg)
    S.fill_synthetic_(m)
    m = m.cuda().eval()
    with torch.no_grad():
        first = [o.clone() for o in m(x.cuda(), margin=margin.cuda())]
        for rep in range(4):
            again = m(x.cuda(), margin=margin.cuda())
            for name, a, b in zip(("output", "rough", "intersection", "prev_b", "context", "squeezed", "total_b"), first, again):
                assert torch.equal(a, b), (rep, name, float((a - b).abs().max()))
