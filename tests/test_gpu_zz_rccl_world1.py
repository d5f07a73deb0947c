"""The RCCL code path on ONE GPU: gradient exchange of the data-parallel training step through a one-rank
"nccl" process group.  Kept in a file of its own that sorts LAST: in a process that has initialised (and destroyed) an RCCL
communicator, a later hipGraph replay of an inference engine segfaults inside the runtime (torch 2.10 / ROCm 7.2, seen when
tests/test_gpu_window7.py ran after this test in the full suite) - nothing may replay graphs after it."""
import copy
import os

import pytest
import torch

from otpose_amd import parallel as PAR
from otpose_amd import synthetic as S
from otpose_amd.optim import FusedAdamW
from tests.test_gpu_train_slots import (CLIP, GRAD_TOL, LR, WD, _assert_weights_close, _grad_err, _loss, _pair, _targets)

pytestmark = pytest.mark.gpu


def test_rccl_exchange_on_one_rank_matches_no_exchange():
    """RCCL path on ONE GPU: a process group of a single rank with OTPOSE_FORCE_COLLECTIVES=1 sends the joint flags, the
    flat gradient buffers (FusedAdamW) and GradBuckets' packed buckets (hook mode, launched from inside the backward with
    gradients written on the HRNet side streams) through ``dist.all_reduce`` on device tensors.  A sum over one rank is the
    identity, so the step must reproduce the step without any process group."""
    import socket
    import torch.distributed as dist
    cfg, a, b = _pair("bf16")
    x, margin = S.synthetic_clip(2, cfg.MODEL.IMAGE_SIZE)
    x, margin = x.cuda(), margin.cuda()
    J, (w, h) = cfg.MODEL.NUM_JOINTS, cfg.MODEL.HEATMAP_SIZE
    g, wt = _targets(2, J, h, w)
    c, d = copy.deepcopy(b), copy.deepcopy(b)                         # before any forward: no cached packs travel
    for m in (c, d):
        m.train_dropout, m.train_dtype = False, "bf16"
    opt_a = FusedAdamW([p for p in a.parameters() if p.requires_grad], lr=LR, weight_decay=WD, max_grad_norm=CLIP)
    opt_b = FusedAdamW([p for p in b.parameters() if p.requires_grad], lr=LR, weight_decay=WD, max_grad_norm=CLIP)
    loss_b = PAR.train_step_dp(b, opt_b, x, margin, g, wt)          # no process group: no collective
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ["OTPOSE_FORCE_COLLECTIVES"] = "1"
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1,
                            device_id=torch.device("cuda", 0))
    try:
        assert PAR.collectives_on() and PAR.world_size() == 1
        loss_a = PAR.train_step_dp(a, opt_a, x, margin, g, wt)      # flags MAX, 3 flat all-reduces, loss mean: all RCCL
        torch.cuda.synchronize()
        assert abs(float(loss_a) - float(loss_b)) <= 1e-5 * max(1.0, abs(float(loss_b)))
        _assert_weights_close(a, b, "bf16", "FusedAdamW + forced RCCL vs none")
        # bucketed exchange launched by hooks from inside the backward (per-parameter gradients, torch optimizer)
        params = [p for p in c.parameters() if p.requires_grad]
        bk = PAR.GradBuckets(params, bucket_bytes=1 << 20, hooks=True)
        assert bk.active and len(bk.buckets) > 1
        _loss(c, x, margin, g, wt).backward()
        launched = sum(1 for wk in bk._work if wk is not None)
        bk.finish()
        bk.remove_hooks()
        torch.cuda.synchronize()
        assert launched >= len(bk.buckets) - 1, launched          # buckets went out during the backward
        # reference: the same backward with no exchange at all
        os.environ["OTPOSE_FORCE_COLLECTIVES"] = "0"
        _loss(d, x, margin, g, wt).backward()
        pd = dict(d.named_parameters())
        triples = [(n, p.grad, pd[n].grad) for n, p in c.named_parameters() if p.grad is not None and pd[n].grad is not None]
        glob, worst, wname = _grad_err(triples, None)
        print("GradBuckets (hooks, RCCL world 1) vs plain backward: rel L2 %.2e, worst %.2e (%s)" % (glob, worst, wname))
        assert len(triples) > 300 and glob <= GRAD_TOL["bf16"]
    finally:
        os.environ.pop("OTPOSE_FORCE_COLLECTIVES", None)
        dist.destroy_process_group()
