"""Checkpoint save / resume with the reference's file layout (model/checkpoints.py:6-74, utils/setup.py:135-165), so that
checkpoints written by either side load in the other:

* file ``<folder>/epoch_<E>_state.pth`` (``best_mAP_<mAP>_state.pth`` for the best one) holding
  ``{"begin_epoch": E, "state_dict": model.state_dict(), "optimizer": optimizer.state_dict(),
  "tensorboard_global_steps": n}``;
* the ``module.`` prefix ``nn.DataParallel`` puts in front of every key is stripped on save (checkpoints.py:35-38) and
  tolerated on load;
* ``resume`` returns ``(model, optimizer, begin_epoch + 1, {"tensorboard_global_steps": n})`` (checkpoints.py:6-25) and
  moves the optimizer state to each parameter's device (the reference calls ``.cuda()`` on it).

``otpose_amd.optim.FusedAdamW`` writes the state layout of ``torch.optim.AdamW`` (per-parameter ``step`` / ``exp_avg`` /
``exp_avg_sq``, the same ``param_groups``), so a reference checkpoint resumes into the fused optimizer and vice versa."""
from __future__ import annotations

import os
import os.path as osp

import torch


def _strip_module(sd):
    """Keys as a bare model has them: nn.DataParallel's ``module.`` prefix removed (checkpoints.py:35-38)."""
    if sd and all(k.startswith("module.") for k in sd):
        return {k[7:]: v for k, v in sd.items()}
    return sd


def _checkpoint_dict(epoch, model, optimizer, global_steps):
    return {"begin_epoch": epoch, "state_dict": _strip_module(dict(model.state_dict())), "optimizer": optimizer.state_dict(),
            "tensorboard_global_steps": global_steps}


def save_checkpoint(epoch, save_folder, model, optimizer, **kwargs):
    """model/checkpoints.py:28-44: ``epoch_<epoch>_state.pth``; returns the path."""
    os.makedirs(save_folder, exist_ok=True)
    path = osp.join(save_folder, "epoch_{}_state.pth".format(epoch))
    torch.save(_checkpoint_dict(epoch, model, optimizer, kwargs.get("global_steps", 0)), path)
    return path


def save_best_checkpoint(epoch, save_folder, model, optimizer, mAP, **kwargs):
    """model/checkpoints.py:47-74: ``best_mAP_<mAP>_state.pth``.  (The reference scans ``save_folder`` - a string -
    character by character for older "best" files, so it never removes one; older best files are kept here too.)"""
    os.makedirs(save_folder, exist_ok=True)
    path = osp.join(save_folder, "best_mAP_{}_state.pth".format(mAP))
    torch.save(_checkpoint_dict(epoch, model, optimizer, kwargs.get("global_steps", 0)), path)
    return path


def _pth_files(folder):
    if not folder or not osp.isdir(folder):
        return []
    return sorted(osp.join(folder, f) for f in os.listdir(folder) if f.endswith(".pth") and osp.isfile(osp.join(folder, f)))


def get_latest_checkpoint(checkpoint_save_folder):
    """utils/setup.py:135-151: the ``epoch_<N>_state.pth`` with the largest N, None when there is none."""
    files = [p for p in _pth_files(checkpoint_save_folder) if "best" not in osp.basename(p)]
    if not files:
        return None
    return max(files, key=lambda p: int(osp.basename(p).split("_")[1]))


def get_best_checkpoint(checkpoint_save_folder):
    """utils/setup.py:154-165: the ``best_mAP_<mAP>_state.pth`` with the largest mAP, None when there is none."""
    files = [p for p in _pth_files(checkpoint_save_folder) if "best" in osp.basename(p)]
    if not files:
        return None
    return max(files, key=lambda p: float(osp.basename(p).split("_")[2]))


def resume(model, optimizer, checkpoint_file, **kwargs):
    """model/checkpoints.py:6-25.  ``map_location`` (default "cpu") is passed to ``torch.load``; tensors of the optimizer
    state end up on the device of the parameter they belong to."""
    checkpoint = torch.load(checkpoint_file, map_location=kwargs.get("map_location", "cpu"), weights_only=False)
    begin_epoch = checkpoint["begin_epoch"] + 1
    state_dict = checkpoint["state_dict"]
    target = model.module if hasattr(model, "module") and isinstance(model, torch.nn.DataParallel) else model
    target.load_state_dict(_strip_module(dict(state_dict)))
    if optimizer is not None and checkpoint.get("optimizer") is not None:
        optimizer.load_state_dict(checkpoint["optimizer"])
        for p, state in optimizer.state.items():
            for k, v in state.items():
                if torch.is_tensor(v) and v.device != p.device and v.dim() > 0:
                    state[k] = v.to(p.device)
    return model, optimizer, begin_epoch, {"tensorboard_global_steps": checkpoint.get("tensorboard_global_steps", 0)}
