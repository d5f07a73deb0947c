"""offset / mask convolutions fused into the DCN gather (csrc/dcn_fused.hip, otp_dcn_fused_*)
against the composition it replaces (model/OTPose.py:381-392): float64 ``F.conv2d`` for the offsets / masks, the CPU oracle's
``mdcn_forward`` (restating deform_conv_cuda_kernel.cu:403-432, 506-571) for the gather, the weighted sum over dilations.
Tolerance 2e-4 of the output range: the offsets come out of split-bf16 products (relative error ~4e-6) and move the sample
points by that much."""
import pytest
import torch
import torch.nn.functional as F

from oracle import otpose_oracle as O
from otpose_amd import ops

pytestmark = pytest.mark.gpu
J = 17


def _case(b, h, w, dils, seed, off_scale):
    g = torch.Generator().manual_seed(seed)
    trans = torch.randn(b, 32, h, w, generator=g)
    x = torch.randn(b, J, h, w, generator=g)
    w_off = [torch.randn(18 * J, 32, 3, 3, generator=g) * off_scale / 17.0 for _ in dils]
    w_msk = [torch.randn(9 * J, 32, 3, 3, generator=g) / 17.0 for _ in dils]
    w_dcn = [torch.randn(J, J, 3, 3, generator=g) * 0.2 for _ in dils]
    bias = [torch.randn(J, generator=g) for _ in dils]
    return trans, x, w_off, w_msk, w_dcn, bias


def _reference(trans, x, w_off, w_msk, w_dcn, bias, dils):
    acc = None
    for i, d in enumerate(dils):
        off = F.conv2d(trans.double(), w_off[i].double(), None, 1, d, d)
        msk = F.conv2d(trans.double(), w_msk[i].double(), None, 1, d, d)
        y = O.mdcn_forward(x.double(), off, msk, w_dcn[i].double(), bias[i].double(), 1, d, d, 1, J)
        acc = y if acc is None else acc + y
    return acc / len(dils)


@pytest.mark.parametrize("shape", [(2, 16, 24, (3, 6, 9, 12, 15)), (1, 8, 16, (1, 2)), (3, 32, 12, (4,))],
                         ids=["five_dilations", "tiny", "tall"])
def test_dcn_fused_matches_conv_plus_oracle_dcn(shape):
    b, h, w, dils = shape
    assert ops.dcn_fused_supported(32, J, h, w, len(dils))
    trans, x, w_off, w_msk, w_dcn, bias = _case(b, h, w, dils, 5 + h, 2.0)
    ref = _reference(trans, x, w_off, w_msk, w_dcn, bias, dils)
    packed = ops.pack_dcn_fused([t.cuda() for t in w_off], [t.cuda() for t in w_msk], [t.cuda() for t in w_dcn],
                                [t.cuda() for t in bias])
    out = ops.dcn_fused(trans.cuda(), x.cuda(), packed, dils, 1.0 / len(dils))
    err = float((out.cpu().double() - ref).abs().max()) / float(ref.abs().max())
    assert err <= 2e-4, err


def test_dcn_fused_matches_the_unfused_hip_path():
    """Same inputs through the three launches per dilation the engine used before (otp_conv2d x2 + otp_mdcn_forward)."""
    b, h, w, dils = 2, 16, 24, (3, 6)
    trans, x, w_off, w_msk, w_dcn, bias = _case(b, h, w, dils, 11, 1.0)
    tc, xc = trans.cuda(), x.cuda()
    acc = torch.zeros(b, J, h, w, device="cuda")
    for i, d in enumerate(dils):
        off = ops.conv2d(tc, w_off[i].cuda(), None, None, 1, d, d)
        msk = ops.conv2d(tc, w_msk[i].cuda(), None, None, 1, d, d)
        acc += ops.modulated_deform_conv(xc, off, msk, w_dcn[i].cuda(), bias[i].cuda(), 1, d, d, 1, J) / len(dils)
    packed = ops.pack_dcn_fused([t.cuda() for t in w_off], [t.cuda() for t in w_msk], [t.cuda() for t in w_dcn],
                                [t.cuda() for t in bias])
    out = ops.dcn_fused(tc, xc, packed, dils, 1.0 / len(dils))
    assert float((out - acc).abs().max()) <= 2e-4 * float(acc.abs().max())


def test_dcn_fused_without_bias_and_unsupported_shapes():
    b, h, w, dils = 1, 8, 16, (2,)
    trans, x, w_off, w_msk, w_dcn, bias = _case(b, h, w, dils, 3, 1.0)
    ref = _reference(trans, x, w_off, w_msk, w_dcn, [torch.zeros(J)], dils)
    packed = ops.pack_dcn_fused([w_off[0].cuda()], [w_msk[0].cuda()], [w_dcn[0].cuda()], [None])
    out = ops.dcn_fused(trans.cuda(), x.cuda(), packed, dils, 1.0)
    assert float((out.cpu().double() - ref).abs().max()) <= 2e-4 * float(ref.abs().max())
    assert not ops.dcn_fused_supported(32, J, 9, 7, 1)       # H * W % 128 != 0
    assert not ops.dcn_fused_supported(24, J, 8, 16, 1)      # other channel counts of `trans`


def test_dcn_fused_full_size_batch_slices_are_independent():
    """BASELINE configs[1] size (16 clips, 96x72, five dilations): clip 5 alone gives bit-identical heat-maps to clip 5
    inside the batch, and zero offset / mask weights reduce the head to the DCN biases."""
    b, h, w, dils = 16, 96, 72, (3, 6, 9, 12, 15)
    trans, x, w_off, w_msk, w_dcn, bias = _case(b, h, w, dils, 21, 1.0)
    cu = lambda ts: [t.cuda() for t in ts]      # noqa: E731
    packed = ops.pack_dcn_fused(cu(w_off), cu(w_msk), cu(w_dcn), cu(bias))
    full = ops.dcn_fused(trans.cuda(), x.cuda(), packed, dils, 0.2)
    solo = ops.dcn_fused(trans[5:6].cuda().contiguous(), x[5:6].cuda().contiguous(), packed, dils, 0.2)
    assert torch.equal(solo, full[5:6])
    zeros = [torch.zeros_like(t) for t in w_msk]
    packed0 = ops.pack_dcn_fused(cu(w_off), cu(zeros), cu(w_dcn), cu(bias))
    out0 = ops.dcn_fused(trans.cuda(), x.cuda(), packed0, dils, 0.2)
    expect = (sum(bias) * 0.2).cuda().view(1, J, 1, 1).expand_as(out0)
    assert float((out0 - expect).abs().max()) <= 1e-6
