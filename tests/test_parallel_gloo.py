"""N > 1 path on CPU: two gloo ranks exercise clip sharding, bucketed gradient all-reduce (plain and
hook-overlapped), the per-joint flag MAX-reduce and loss-mean composition."""
import os
import socket
import tempfile

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from otpose_amd import parallel as P


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _make_model(seed=0):
    torch.manual_seed(seed)
    return torch.nn.Sequential(torch.nn.Conv2d(3, 8, 3, padding=1), torch.nn.ReLU(), torch.nn.Conv2d(8, 4, 1),
                               torch.nn.Flatten(), torch.nn.Linear(4 * 6 * 5, 7))


def _spawn(target, world, *args):
    """Run ``target(rank, world, port, out_path, *args)`` on ``world`` spawned ranks; rank 0 ``torch.save``s its result to
    ``out_path`` (plain file hand-off: a tensor put on a multiprocessing queue travels through a shared-memory handle
    that dies with the sender, which made this test flaky when rank 0 exited before the parent had read it)."""
    ctx = mp.get_context("spawn")
    port = _free_port()
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "rank0.pt")
        procs = [ctx.Process(target=target, args=(r, world, port, out) + args) for r in range(world)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(180)
            assert p.exitcode == 0, f"rank exited with {p.exitcode}"
        return torch.load(out)


def _worker(rank, world, port, out_path, hooks):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    torch.set_num_threads(1)
    r, w, dev = P.init_from_env("gloo")
    assert (r, w) == (rank, world) and dev.type == "cpu"
    g = torch.Generator().manual_seed(5)
    x = torch.randn(6, 3, 6, 5, generator=g)            # global batch of 6 "clips"
    y = torch.randn(6, 7, generator=g)
    margin = torch.arange(24.).view(6, 4)
    xs, ms = P.shard_clips(x, margin)
    b, e = P.shard_range(6, rank, world)
    assert torch.equal(ms, margin[b:e]) and xs.shape[0] == e - b
    model = _make_model()
    buckets = P.GradBuckets(model.parameters(), bucket_bytes=1024, hooks=hooks)   # several small buckets
    assert len(buckets.buckets) > 1
    # equal shards: mean over the global batch == mean over ranks of the local means
    loss = ((model(xs) - y[b:e]) ** 2).mean()
    loss.backward()
    if hooks:
        buckets.finish()
    else:
        buckets.reduce()
    grads = [p.grad.clone() for p in model.parameters()]
    lm = P.allreduce_mean_(loss.detach().clone())
    flags = torch.tensor([1, 0, 0] if rank == 0 else [0, 0, 1], dtype=torch.int32)
    P.allreduce_joint_flags(flags)
    gathered = P.gather_clips(xs.contiguous())
    # second step reuses the buckets (pending counters reset)
    model.zero_grad()
    ((model(xs) - y[b:e]) ** 2).mean().backward()
    if hooks:
        buckets.finish()
    else:
        buckets.reduce()
    grads2 = [p.grad.clone() for p in model.parameters()]
    if rank == 0:
        torch.save({"grads": grads, "grads2": grads2, "loss": lm, "flags": flags, "gathered": gathered}, out_path)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("hooks", [False, True])
def test_two_rank_gradient_allreduce_matches_global_batch(hooks):
    out = _spawn(_worker, 2, hooks)
    g = torch.Generator().manual_seed(5)
    x = torch.randn(6, 3, 6, 5, generator=g)
    y = torch.randn(6, 7, generator=g)
    model = _make_model()
    loss = ((model(x) - y) ** 2).mean()
    loss.backward()
    for got, got2, p in zip(out["grads"], out["grads2"], model.parameters()):
        assert torch.allclose(got, p.grad, atol=1e-6), float((got - p.grad).abs().max())
        assert torch.allclose(got2, p.grad, atol=1e-6)
    assert abs(float(out["loss"]) - float(loss.detach())) < 1e-6
    assert out["flags"].tolist() == [1, 0, 1]
    assert torch.equal(out["gathered"], x)


def test_shard_range_partitions_ragged_batches():
    for n in (0, 1, 5, 16, 17, 128):
        for world in (1, 2, 3, 8):
            spans = [P.shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [e - b for b, e in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        P.shard_range(4, 2, 2)


def test_single_process_is_a_no_op():
    m = _make_model()
    m(torch.randn(2, 3, 6, 5)).sum().backward()
    before = [p.grad.clone() for p in m.parameters()]
    bk = P.GradBuckets(m.parameters())
    bk.reduce()
    assert all(torch.equal(a, p.grad) for a, p in zip(before, m.parameters()))
    assert P.world_size() == 1 and P.rank() == 0


class _FlatStub:
    """Stands in for FusedAdamW (which needs HIP tensors): anything with flat_grads()."""

    def __init__(self, bufs):
        self.bufs = bufs

    def flat_grads(self):
        return self.bufs


def _flat_worker(rank, world, port, out_path):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    torch.set_num_threads(1)
    P.init_from_env("gloo")
    opt = _FlatStub([torch.full((1000,), float(rank + 1)), torch.arange(7.) * (rank + 1)])
    P.allreduce_flat_grads(opt)
    if rank == 0:
        torch.save([b.clone() for b in opt.bufs], out_path)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_flat_gradient_allreduce():
    """allreduce_flat_grads: one all-reduce per flat group buffer, mean over ranks, in place."""
    a, b = _spawn(_flat_worker, 2)
    assert torch.equal(a, torch.full((1000,), 1.5)) and torch.equal(b, torch.arange(7.) * 1.5)


# ---- hook mode: launch order must not depend on which gradients exist on a rank -----------------------------------
class _TwoHeads(torch.nn.Module):
    def __init__(self):
        super().__init__()
        torch.manual_seed(3)
        self.a = torch.nn.Linear(5, 64)
        self.b = torch.nn.Linear(5, 64)
        self.c = torch.nn.Linear(64, 3)

    def forward(self, x, use_b):
        h = self.a(x) + (self.b(x) if use_b else 0)
        return self.c(h)


def _uneven_worker(rank, world, port, out_path):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    torch.set_num_threads(1)
    P.init_from_env("gloo")
    m = _TwoHeads()
    bk = P.GradBuckets(m.parameters(), bucket_bytes=512, hooks=True)
    assert len(bk.buckets) >= 3
    x = torch.full((4, 5), float(rank + 1))
    # head b gets a gradient on rank 1 only: rank 0's bucket for it is flushed in finish(), after the others were
    # launched in index order on both ranks
    m(x, use_b=(rank == 1)).sum().backward()
    bk.finish()
    res = [p.grad.clone() for p in m.parameters()]
    # a second backward before finish() must be refused in hook mode
    m.zero_grad()
    m(x, True).sum().backward()
    try:
        m(x, True).sum().backward()
        refused = False
    except RuntimeError:
        refused = True
    if rank == 0:
        torch.save({"grads": res, "refused": refused}, out_path)
    # leave without another collective: the second step's buckets are half launched by design
    dist.barrier()
    dist.destroy_process_group()


def test_hook_mode_with_a_gradient_missing_on_one_rank():
    out = _spawn(_uneven_worker, 2)
    assert out["refused"]
    m = _TwoHeads()
    x1, x2 = torch.full((4, 5), 1.0), torch.full((4, 5), 2.0)
    (m(x1, False).sum() + m(x2, True).sum()).backward()
    for got, p in zip(out["grads"], m.parameters()):
        assert torch.allclose(got, p.grad / 2, atol=1e-5), float((got - p.grad / 2).abs().max())


# ---- train_step_dp: MAX-reduced joint flags make the sharded loss the global-batch loss -------------------------------
def _loss_case():
    g = torch.Generator().manual_seed(11)
    B, J, h, w = 4, 17, 6, 5
    s = torch.randn(B, J, h, w, generator=g) * 0.3
    t = torch.randn(B, J, h, w, generator=g) * 0.3
    tgt = torch.rand(B, J, h, w, generator=g) * 0.9
    # exact-1 peaks: joints 0-5 only in the first half, joints 6-9 only in the second half, 10-12 in both, 13-16 nowhere
    for j in range(0, 6):
        tgt[0, j, 2, 2] = 1.0
    for j in range(6, 10):
        tgt[3, j, 1, 4] = 1.0
    for j in range(10, 13):
        tgt[1, j, 0, 0] = 1.0
        tgt[2, j, 5, 4] = 1.0
    wgt = (torch.rand(B, J, 1, generator=g) > 0.15).float()
    return s, t, tgt, wgt


class _Heads(torch.nn.Module):
    """Stand-in model: student / teacher maps are learnable tensors indexed by the clip ids in ``x``."""

    def __init__(self, s, t):
        super().__init__()
        self.s = torch.nn.Parameter(s.clone())
        self.t = torch.nn.Parameter(t.clone())


def _dp_worker(rank, world, port, out_path):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    torch.set_num_threads(1)
    P.init_from_env("gloo")
    from oracle import otpose_oracle as O
    s, t, tgt, wgt = _loss_case()
    b, e = P.shard_range(s.shape[0], rank, world)
    model = _Heads(s, t)
    opt = torch.optim.SGD(model.parameters(), lr=0.5)
    ids = torch.arange(b, e)

    def forward(m, x, margin):
        return m.s[x], m.t[x]

    def criterion(outs, target, target_weight, flags):
        return O.st_ohkw_mse_loss(outs[0], outs[1], target, target_weight, 8, global_flags=flags)["final_loss"]

    local_flags = P.joint_flags(tgt[b:e]).tolist()
    loss = P.train_step_dp(model, opt, ids, None, tgt[b:e], wgt[b:e], forward=forward, criterion=criterion)
    if rank == 0:
        torch.save({"loss": loss, "s": model.s.detach().clone(), "t": model.t.detach().clone(),
                    "local_flags": local_flags}, out_path)
    dist.barrier()
    dist.destroy_process_group()


def test_train_step_dp_matches_the_global_batch_loss():
    """Two half batches with MAX-reduced flags == ST_OHKW on the whole batch (model/loss.py:47,
    script/Common.py:124-130), for the loss value and for the parameter update."""
    from oracle import otpose_oracle as O
    out = _spawn(_dp_worker, 2)
    s, t, tgt, wgt = _loss_case()
    # the local flags of rank 0 differ from the global ones, so the reduce is what makes the branch right
    glob = P.joint_flags(tgt).tolist()
    assert out["local_flags"] != glob and glob == [1] * 13 + [0] * 4
    model = _Heads(s, t)
    ref = O.st_ohkw_mse_loss(model.s, model.t, tgt, wgt, 8)
    # the OHKM mean and every per-joint mean are means over samples: with equal shards the mean over ranks of the
    # local losses is the global loss
    assert abs(float(out["loss"]) - float(ref["final_loss"].detach())) < 1e-6
    ref["final_loss"].backward()
    with torch.no_grad():
        # rank r's gradient is non-zero on its own rows only and carries a 1/(B/world) mean; the all-reduce mean
        # divides by world, so the update equals the global-batch update
        assert torch.allclose(out["s"], model.s - 0.5 * model.s.grad, atol=1e-6)
        assert torch.allclose(out["t"], model.t - 0.5 * model.t.grad, atol=1e-6)


# ---- worlds of 4 and 8 ranks: what the first `bench.py --gpus 8` will exercise --------------------------------
def _loss_case_n(B):
    g = torch.Generator().manual_seed(23)
    J, h, w = 17, 6, 5
    s = torch.randn(B, J, h, w, generator=g) * 0.3
    t = torch.randn(B, J, h, w, generator=g) * 0.3
    tgt = torch.rand(B, J, h, w, generator=g) * 0.9
    for b in range(B):                                   # joint b (mod 13) has its exact-1 peak ONLY in clip b: every rank's local
        tgt[b, b % 13, b % h, b % w] = 1.0               # flags differ from the global ones; joints 13-16 have none anywhere
    wgt = (torch.rand(B, J, 1, generator=g) > 0.15).float()
    return s, t, tgt, wgt


def _dp_world_worker(rank, world, port, out_path, B, steps):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    torch.set_num_threads(1)
    P.init_from_env("gloo")
    from oracle import otpose_oracle as O
    s, t, tgt, wgt = _loss_case_n(B)
    b, e = P.shard_range(B, rank, world)
    model = _Heads(s, t)
    opt = torch.optim.SGD(model.parameters(), lr=0.5)
    ids = torch.arange(b, e)

    def forward(m, x, margin):
        return m.s[x], m.t[x]

    def criterion(outs, target, target_weight, flags):
        return O.st_ohkw_mse_loss(outs[0], outs[1], target, target_weight, 8, global_flags=flags)["final_loss"]

    losses, stats = [], {}
    for it in range(steps):                              # the second step reuses the gradient buckets of the first
        losses.append(float(P.train_step_dp(model, opt, ids, None, tgt[b:e], wgt[b:e], forward=forward, criterion=criterion,
                                            stats=stats if it == steps - 1 else None)))
    assert stats["world"] == world and stats["comm_ms"] >= 0.0 and stats["comm_bytes"] > 0
    # every replica must hold the same parameters after the exchange: compare checksums across ranks
    chk = torch.stack([model.s.detach().double().sum(), model.t.detach().double().sum(), model.s.detach().double().abs().sum()])
    allc = [torch.empty_like(chk) for _ in range(world)]
    dist.all_gather(allc, chk)
    same = all(torch.equal(c, allc[0]) for c in allc)
    flags = P.allreduce_joint_flags(P.joint_flags(tgt[b:e])).tolist()
    if rank == 0:
        torch.save({"losses": losses, "s": model.s.detach().clone(), "t": model.t.detach().clone(), "same": same, "flags": flags,
                    "shard": (b, e)}, out_path)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [4, 8])
def test_train_step_dp_on_four_and_eight_ranks_equals_the_global_batch(world):
    """Equal shards (8 clips): two steps on `world` ranks == two steps of the global-batch loss (flags MAX-reduced, gradients
    averaged, buckets reused by the second step), and every replica ends with identical parameters."""
    from oracle import otpose_oracle as O
    B, steps = 8, 2
    out = _spawn(_dp_world_worker, world, B, steps)
    assert out["same"] and out["flags"] == [1] * 8 + [0] * 9
    s, t, tgt, wgt = _loss_case_n(B)
    model = _Heads(s, t)
    opt = torch.optim.SGD(model.parameters(), lr=0.5)
    ref_losses = []
    for _ in range(steps):
        opt.zero_grad()
        loss = O.st_ohkw_mse_loss(model.s, model.t, tgt, wgt, 8)["final_loss"]
        loss.backward()
        opt.step()
        ref_losses.append(float(loss.detach()))
    for a, b in zip(out["losses"], ref_losses):
        assert abs(a - b) < 1e-6 * max(1.0, abs(b)), (out["losses"], ref_losses)      # (fp32 sums in another order)
    assert torch.allclose(out["s"], model.s.detach(), atol=2e-6) and torch.allclose(out["t"], model.t.detach(), atol=2e-6)


@pytest.mark.parametrize("world,B", [(4, 10), (8, 11)])
def test_train_step_dp_with_ragged_shards_keeps_replicas_identical(world, B):
    """Ragged shards (10 clips on 4 ranks: 3, 3, 2, 2; 11 on 8): the control flow of the step - flag MAX, bucket reuse across
    steps, exchange, optimizer - runs to the end on every rank, the flags are the global ones and the replicas stay identical
    (the mean over ranks then weights clips unequally: the reference's DataLoader drops the last batch for the same reason)."""
    out = _spawn(_dp_world_worker, world, B, 3)
    assert out["same"]
    assert out["flags"] == [1] * min(B, 13) + [0] * (17 - min(B, 13))
    assert all(l == l and abs(l) < 1e6 for l in out["losses"])
    assert out["shard"] == P.shard_range(B, 0, world)


def test_a_group_created_and_destroyed_with_plain_torch_distributed_calls_is_seen():
    """graph_replay_safe() used to depend on a parallel.* helper having run while the group was alive.  The hooks
    installed at import latch creation / destruction whoever calls torch.distributed."""
    import subprocess
    import sys
    code = (
        "import os, torch.distributed as dist\n"
        "import otpose_amd\n"
        "from otpose_amd import parallel as P\n"
        "assert P.graph_replay_safe()\n"
        "os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT='%d')\n"
        "dist.init_process_group('gloo', rank=0, world_size=1)\n"
        "dist.destroy_process_group()\n"
        "assert not P.graph_replay_safe(), 'a torn-down group went unnoticed'\n"
        "print('ok')\n" % _free_port())
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120,
                       cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert r.returncode == 0 and "ok" in r.stdout, r.stderr[-2000:]
