"""Fused clip + AdamW (otp_grad_sumsq / otp_adamw_step) vs torch.nn.utils.clip_grad_norm_ + torch.optim.AdamW on the CPU
(the two calls of script/Common.py:138-143), three parameter groups like make_optimizer's."""
import copy

import pytest
import torch

pytestmark = pytest.mark.gpu


def _model():
    # no normalisation layer: a bias in front of one has a pure-rounding-noise gradient, which Adam turns into +-lr steps
    torch.manual_seed(3)
    return torch.nn.Sequential(torch.nn.Conv2d(3, 7, 3), torch.nn.Tanh(), torch.nn.Conv2d(7, 5, 1),
                               torch.nn.Flatten(), torch.nn.Linear(5 * 36, 11))


def _groups(m, lr):
    decay = [p for n, p in m.named_parameters() if n.endswith("weight") and p.dim() > 1]
    no_decay = [p for n, p in m.named_parameters() if not (n.endswith("weight") and p.dim() > 1)]
    return [{"params": decay[:1], "weight_decay": 0.05, "lr": lr / 100}, {"params": decay[1:], "weight_decay": 0.05},
            {"params": no_decay, "weight_decay": 0.0}]


@pytest.mark.parametrize("max_norm", [0.0, 0.05])
def test_fused_adamw_matches_torch(max_norm):
    from otpose_amd.optim import FusedAdamW
    ref = _model()
    dut = copy.deepcopy(ref).cuda()
    lr = 3e-3
    o_ref = torch.optim.AdamW(_groups(ref, lr), lr=lr)
    o_dut = FusedAdamW(_groups(dut, lr), lr=lr, max_grad_norm=max_norm)
    versions = [p._version for p in dut.parameters()]
    for it in range(4):
        # identical gradients on both sides (the optimizer arithmetic is under test, not the backward kernels)
        gen = torch.Generator().manual_seed(10 + it)
        o_ref.zero_grad()
        o_dut.zero_grad()
        for a, b in zip(ref.parameters(), dut.parameters()):
            g = torch.randn(a.shape, generator=gen) * (0.1 if a.dim() > 1 else 1e-3)
            a.grad = g.clone()
            b.grad = g.cuda()                                # an ordinary tensor: step() copies it into the flat slot
        if max_norm > 0:
            total = torch.nn.utils.clip_grad_norm_(ref.parameters(), max_norm)
            assert abs(float(o_dut.grad_norm()) - float(total)) <= 1e-5 * float(total)
            assert float(total) > max_norm                  # the clip is active
        o_ref.step()
        o_dut.step()
        for (n, a), b in zip(ref.named_parameters(), dut.parameters()):
            assert float((a - b.cpu()).abs().max()) <= 2e-6 * max(1.0, float(a.abs().max())), (it, n)
    assert all(p._version > v for p, v in zip(dut.parameters(), versions))      # staleness checks see the update
    sd = o_dut.state_dict()
    assert sd["state"][0]["step"] == 4 and sd["state"][0]["exp_avg"].shape == next(dut.parameters()).shape
    o_dut.load_state_dict(sd)
    assert o_dut.state[next(dut.parameters())]["exp_avg"].data_ptr() == o_dut._flat[0]["m"].data_ptr()
