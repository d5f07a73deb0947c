"""Checkpoint tooling with the reference's file layout (model/checkpoints.py:6-74, utils/setup.py:135-165) and the HRNet
pre-training import of model/OTPose.py:477-499."""
import os

import pytest
import torch

from otpose_amd import OTPose, checkpoints as CK, tiny_cfg
from otpose_amd import synthetic as S


def _model():
    m = OTPose(tiny_cfg(8, (64, 96)))
    S.fill_synthetic_(m)
    return m


def _opt(m):
    dec = [p for n, p in m.named_parameters() if p.dim() > 1]
    rest = [p for n, p in m.named_parameters() if p.dim() <= 1]
    return torch.optim.AdamW([{"params": dec, "weight_decay": 0.01}, {"params": rest, "weight_decay": 0.0, "lr": 1e-5}], lr=1e-3)


def _fake_step(m, o, seed):
    g = torch.Generator().manual_seed(seed)
    for p in m.parameters():
        p.grad = torch.randn(p.shape, generator=g) * 1e-3
    o.step()


def test_save_resume_round_trip_and_latest(tmp_path):
    m = _model()
    o = _opt(m)
    _fake_step(m, o, 1)
    paths = [CK.save_checkpoint(e, str(tmp_path), m, o, global_steps=100 * e) for e in (0, 2, 11)]
    assert [os.path.basename(p) for p in paths] == ["epoch_0_state.pth", "epoch_2_state.pth", "epoch_11_state.pth"]
    best = CK.save_best_checkpoint(7, str(tmp_path), m, o, 83.5)
    CK.save_best_checkpoint(9, str(tmp_path), m, o, 84.25)
    assert os.path.basename(best) == "best_mAP_83.5_state.pth"
    assert CK.get_latest_checkpoint(str(tmp_path)) == paths[2]                 # numeric, not lexicographic: 11 > 2
    assert os.path.basename(CK.get_best_checkpoint(str(tmp_path))) == "best_mAP_84.25_state.pth"
    assert CK.get_latest_checkpoint(str(tmp_path / "nothing_here")) is None
    # the file is the reference's dict
    ck = torch.load(paths[2], weights_only=False)
    assert set(ck) == {"begin_epoch", "state_dict", "optimizer", "tensorboard_global_steps"}
    assert ck["begin_epoch"] == 11 and ck["tensorboard_global_steps"] == 1100
    assert set(ck["state_dict"]) == set(m.state_dict())
    m2 = OTPose(tiny_cfg(8, (64, 96)))
    o2 = _opt(m2)
    _, _, begin, ext = CK.resume(m2, o2, paths[2])
    assert begin == 12 and ext == {"tensorboard_global_steps": 1100}
    for (k, a), b in zip(m.state_dict().items(), m2.state_dict().values()):
        assert torch.equal(a, b), k
    # the optimizer continues identically
    _fake_step(m, o, 2)
    _fake_step(m2, o2, 2)
    for a, b in zip(m.parameters(), m2.parameters()):
        assert torch.equal(a, b)


def test_data_parallel_prefix_is_stripped_on_save_and_tolerated_on_load(tmp_path):
    m = _model()
    wrapped = torch.nn.DataParallel(m)
    assert next(iter(wrapped.state_dict())).startswith("module.")
    path = CK.save_checkpoint(3, str(tmp_path), wrapped, _opt(m))
    ck = torch.load(path, weights_only=False)
    assert not any(k.startswith("module.") for k in ck["state_dict"])          # checkpoints.py:35-38
    # a checkpoint that kept the prefix (written from a wrapped model by other tooling) still resumes
    ck["state_dict"] = {"module." + k: v for k, v in ck["state_dict"].items()}
    torch.save(ck, path)
    m2 = OTPose(tiny_cfg(8, (64, 96)))
    CK.resume(m2, _opt(m2), path)
    for (k, a), b in zip(m.state_dict().items(), m2.state_dict().values()):
        assert torch.equal(a, b), k
    # and into a wrapped model
    w3 = torch.nn.DataParallel(OTPose(tiny_cfg(8, (64, 96))))
    CK.resume(w3, None, path)
    assert torch.equal(next(iter(w3.module.state_dict().values())), next(iter(m.state_dict().values())))


def test_pretrained_hrnet_checkpoint_without_prefix(tmp_path):
    """model/OTPose.py:477-496: a stand-alone HRNet checkpoint (keys conv1.weight, layer1.0..., no
    ``rough_pose_estimation_net.`` prefix, optionally wrapped in {"state_dict": ...}) initialises the backbone only."""
    src = _model()
    hr = {k[len("rough_pose_estimation_net."):]: v.clone() + 0.25 for k, v in src.state_dict().items()
          if k.startswith("rough_pose_estimation_net.") and v.is_floating_point()}
    hr["some_other_head.weight"] = torch.zeros(3)                              # a layer the model does not have: ignored
    path = str(tmp_path / "hrnet_w8.pth")
    torch.save({"state_dict": hr}, path)
    cfg = tiny_cfg(8, (64, 96))
    cfg.MODEL.PRETRAINED = path
    m = OTPose(cfg)
    sd = m.state_dict()
    for k, v in hr.items():
        if k.startswith("some_other"):
            continue
        assert torch.equal(sd["rough_pose_estimation_net." + k], v), k
    # nothing outside the backbone was touched by the import: the DCN weights still are the identity initialisation
    w = sd["modulated_deform_conv_list.0.deform_conv.weight"]
    assert float(w[0, 0, 1, 1]) == 1.0 and float(w.abs().sum()) == float(min(w.shape[0], w.shape[1]))
    # a missing file is an error, an empty string is not (OTPose.py:497-499)
    cfg.MODEL.PRETRAINED = str(tmp_path / "missing.pth")
    with pytest.raises(ValueError):
        OTPose(cfg)
    cfg.MODEL.PRETRAINED = ""
    OTPose(cfg)


@pytest.mark.gpu
def test_fused_adamw_state_round_trips_with_torch_adamw(tmp_path):
    """A checkpoint written with torch.optim.AdamW resumes into FusedAdamW and back; both continue with the same update."""
    from otpose_amd.optim import FusedAdamW
    torch.manual_seed(0)
    ref = torch.nn.Sequential(torch.nn.Conv2d(3, 6, 3), torch.nn.Tanh(), torch.nn.Conv2d(6, 4, 1)).cuda()
    import copy
    dut = copy.deepcopy(ref)
    o_ref = torch.optim.AdamW(ref.parameters(), lr=1e-2, weight_decay=0.05)
    g = torch.Generator().manual_seed(1)
    grads = [[torch.randn(p.shape, generator=g).cuda() * 0.1 for p in ref.parameters()] for _ in range(4)]
    for p, gr in zip(ref.parameters(), grads[0]):
        p.grad = gr.clone()
    o_ref.step()
    path = CK.save_checkpoint(0, str(tmp_path), ref, o_ref)
    o_dut = FusedAdamW(dut.parameters(), lr=1e-2, weight_decay=0.05)
    _, _, begin, _ = CK.resume(dut, o_dut, path, map_location="cuda")
    assert begin == 1 and o_dut._flat[0]["step"] == 1
    for step in (1, 2):
        for (p, q), gr in zip(zip(ref.parameters(), dut.parameters()), grads[step]):
            p.grad, q.grad = gr.clone(), gr.clone()
        o_ref.step()
        o_dut.step()
    for p, q in zip(ref.parameters(), dut.parameters()):
        assert float((p - q).abs().max()) <= 2e-6 * max(1.0, float(p.abs().max()))
    # and back: FusedAdamW's checkpoint into a fresh torch AdamW
    path2 = CK.save_checkpoint(1, str(tmp_path), dut, o_dut)
    back = copy.deepcopy(ref)
    o_back = torch.optim.AdamW(back.parameters(), lr=1e-2, weight_decay=0.05)
    CK.resume(back, o_back, path2, map_location="cuda")
    for (p, q), gr in zip(zip(dut.parameters(), back.parameters()), grads[3]):
        p.grad, q.grad = gr.clone(), gr.clone()
    o_dut.step()
    o_back.step()
    for p, q in zip(dut.parameters(), back.parameters()):
        assert float((p - q).abs().max()) <= 2e-6 * max(1.0, float(p.abs().max()))
