"""N > 1 path on CPU: two gloo ranks exercise clip sharding, bucketed gradient all-reduce (plain and
hook-overlapped), the per-joint flag MAX-reduce and loss-mean composition."""
import os
import socket

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


def _worker(rank, world, port, hooks, q):
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
        q.put({"grads": grads, "grads2": grads2, "loss": lm, "flags": flags, "gathered": gathered})
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("hooks", [False, True])
def test_two_rank_gradient_allreduce_matches_global_batch(hooks):
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, hooks, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = q.get()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
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


def _flat_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    torch.set_num_threads(1)
    P.init_from_env("gloo")
    opt = _FlatStub([torch.full((1000,), float(rank + 1)), torch.arange(7.) * (rank + 1)])
    P.allreduce_flat_grads(opt)
    if rank == 0:
        q.put([b.clone() for b in opt.bufs])
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_flat_gradient_allreduce():
    """allreduce_flat_grads: one all-reduce per flat group buffer, mean over ranks, in place."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_flat_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    a, b = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert torch.equal(a, torch.full((1000,), 1.5)) and torch.equal(b, torch.arange(7.) * 1.5)
