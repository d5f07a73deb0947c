"""The RCCL code path on ONE GPU: gradient exchange of the data-parallel training step through a one-rank
"nccl" process group.  Kept in a file of its own that sorts LAST: in a process that has initialised (and destroyed) an RCCL
communicator, a later hipGraph replay of an inference engine segfaults inside the runtime (torch 2.10 / ROCm 7.2, seen when
tests/test_gpu_window7.py ran after this test in the full suite) - nothing may replay graphs after it.

Each test runs in a CHILD process of its own (round 4): the communicator's life - init, collectives, teardown - then never
touches the process that holds the rest of the suite.  The teardown of a one-rank group inside the long-lived suite process
(450 tests, dozens of captured graphs behind it) aborted inside ``dist.barrier()`` once in five full runs; a child starts from a
clean runtime, and an abort there fails one test instead of taking the run down."""
import copy
import os
import subprocess
import sys

import pytest
import torch

from otpose_amd import parallel as PAR
from otpose_amd import synthetic as S
from otpose_amd.optim import FusedAdamW
from tests.test_gpu_train_slots import (CLIP, GRAD_TOL, LR, WD, _assert_weights_close, _grad_err, _loss, _pair, _targets)

pytestmark = pytest.mark.gpu
_CHILD = "OTPOSE_RCCL_TEST_CHILD"


def _run_in_child(name):
    """True when this process IS the child that should run the body of ``name``; otherwise runs the child and asserts on it."""
    if os.environ.get(_CHILD) == name:
        return True
    env = dict(os.environ)
    env[_CHILD] = name
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-x", "-m", "gpu", "-k", name,
                        "-p", "no:cacheprovider"], env=env, capture_output=True, text=True, timeout=900,
                       cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    print(r.stdout[-1500:])
    assert r.returncode == 0, "child process failed (rc %s):\n%s\n%s" % (r.returncode, r.stdout[-4000:], r.stderr[-4000:])
    return False


def test_train_then_validate_in_one_process_with_a_live_and_a_destroyed_group():
    """The reference loop trains an epoch and validates in the SAME process (train.py:74-99).  Here: eval forward (hipGraph
    captured), one data-parallel training step through a one-rank RCCL group (flags MAX, flat all-reduce, fused AdamW), eval
    forward again WITH THE GROUP ALIVE (new weights -> the engine re-captures and replays next to a live communicator), then
    the group is destroyed and the eval forward runs once more: graph replays are known to segfault after a communicator
    teardown (module docstring), so the engine must switch itself to eager launches (`parallel.graph_replay_safe`) - same
    kernels, so the same bits as the replay."""
    if not _run_in_child("test_train_then_validate_in_one_process_with_a_live_and_a_destroyed_group"):
        return
    import socket
    import torch.distributed as dist
    from otpose_amd import OTPose, tiny_cfg
    cfg = tiny_cfg(8, (64, 96))
    model = OTPose(cfg)
    S.fill_synthetic_(model)
    model = model.cuda()
    x, margin = S.synthetic_clip(2, cfg.MODEL.IMAGE_SIZE)
    x, margin = x.cuda(), margin.cuda()
    J, (w, h) = cfg.MODEL.NUM_JOINTS, cfg.MODEL.HEATMAP_SIZE
    g, wt = _targets(2, J, h, w)
    assert PAR.graph_replay_safe()
    model.eval()
    with torch.no_grad():
        before = [o.clone() for o in model(x, margin=margin)]
    assert model._engine.graph is not None
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ["OTPOSE_FORCE_COLLECTIVES"] = "1"
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1,
                            device_id=torch.device("cuda", 0))
    try:
        model.train()
        model.train_dropout, model.train_dtype = False, "bf16"
        opt = FusedAdamW([p for p in model.parameters() if p.requires_grad], lr=LR, weight_decay=WD, max_grad_norm=CLIP)
        loss = PAR.train_step_dp(model, opt, x, margin, g, wt)
        assert bool(torch.isfinite(loss))
        model.eval()
        with torch.no_grad():
            live1 = [o.clone() for o in model(x, margin=margin)]      # capture next to the live communicator
            live2 = [o.clone() for o in model(x, margin=margin)]      # replay
        assert model._engine.graph is not None and PAR.graph_replay_safe()
        for a, b in zip(live1, live2):
            assert torch.equal(a, b)
        assert float((live1[0] - before[0]).abs().max()) > 0.0         # the step moved the weights
    finally:
        os.environ.pop("OTPOSE_FORCE_COLLECTIVES", None)
        PAR.shutdown()
    assert not PAR.graph_replay_safe()
    with torch.no_grad():
        after = [o.clone() for o in model(x, margin=margin)]          # would have been a replay: must go eager
    torch.cuda.synchronize()
    assert model._engine.graph is None and not model._engine.use_graph
    for a, b in zip(live1, after):
        assert torch.equal(a, b)
    m2 = OTPose(cfg)                                                   # an engine built after the teardown never captures
    S.fill_synthetic_(m2)
    m2 = m2.cuda().eval()
    with torch.no_grad():
        fresh = [o.clone() for o in m2(x, margin=margin)]
    torch.cuda.synchronize()
    assert m2._engine.graph is None
    for a, b in zip(before, fresh):
        assert torch.equal(a, b)


def test_rccl_exchange_on_one_rank_matches_no_exchange():
    """RCCL path on ONE GPU: a process group of a single rank with OTPOSE_FORCE_COLLECTIVES=1 sends the joint flags, the
    flat gradient buffers (FusedAdamW) and GradBuckets' packed buckets (hook mode, launched from inside the backward with
    gradients written on the HRNet side streams) through ``dist.all_reduce`` on device tensors.  A sum over one rank is the
    identity, so the step must reproduce the step without any process group."""
    if not _run_in_child("test_rccl_exchange_on_one_rank_matches_no_exchange"):
        return
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
