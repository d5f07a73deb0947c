"""One-process-per-GPU data parallelism over RCCL, replacing the reference's single-process
``torch.nn.DataParallel`` (reference train.py:78-79, eval.py:112-113).

The OTPose hot path shards by clip (batch axis 0): nothing in ``forward`` mixes samples except
train-mode BatchNorm statistics, which the reference also keeps per replica (DataParallel has no
SyncBN).  So:

* inference: every rank runs its own clips, **no collective** on the data path
  (:func:`shard_clips`; :func:`gather_clips` only when a caller wants rank 0 to hold all heat-maps,
  as DataParallel's gather did);
* training: identical seeded replicas, gradients summed with **bucketed all-reduce** and scaled by
  1/world (:class:`GradBuckets`) - the reference instead broadcasts 272 MB of parameters and reduces
  272 MB of gradients to GPU 0 every iteration;
* loss parity: ``ST_OHKW_MSELoss`` tests ``max(gt_j) == 1`` over the *global* batch (model/loss.py:47 runs
  on the gathered outputs), so the 17 per-joint flags are MAX-reduced (:func:`allreduce_joint_flags`)
  before the branch is chosen; the mean reductions compose exactly across equal shards
  (:func:`allreduce_mean_`).

Bucket sizing for xGMI: the 8 GPUs are fully connected by point-to-point links (7 x ~153 GB/s per GPU).
RCCL's ring/tree algorithms are per-link bound, so few large buckets beat many small ones: the
default is 64 MiB (5 buckets for the 272 MB of fp32 HRNet-W48 gradients), each launched as soon as its
last gradient is ready so the reduction of bucket k overlaps the backward of bucket k+1.

``backend`` is ``"nccl"`` (= RCCL on ROCm) on GPUs and ``"gloo"`` on CPU (the multi-process tests).
"""
from __future__ import annotations

import os
from typing import Iterable, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist

DEFAULT_BUCKET_BYTES = 64 << 20


def init_from_env(backend: Optional[str] = None) -> Tuple[int, int, torch.device]:
    """Initialise the default process group from the launcher's environment (RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_ADDR / MASTER_PORT, as set by ``python -m torch.distributed.run``).
    Returns (rank, world, device).  With WORLD_SIZE unset or 1 nothing is initialised."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    use_gpu = torch.cuda.is_available()
    device = torch.device("cuda", local) if use_gpu else torch.device("cpu")
    if use_gpu:
        torch.cuda.set_device(device)
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        be = backend or ("nccl" if use_gpu else "gloo")
        kw = {"device_id": device} if be == "nccl" else {}
        dist.init_process_group(be, rank=rank, world_size=world, **kw)
    return rank, world, device


_GROUP_SEEN = False          # this process has used an initialised process group (it may have been destroyed since)
_GROUP_DESTROYED = False     # ... and one was torn down (latched by the hook below, whoever called torch.distributed)


def _install_group_hooks() -> None:
    """Latch process-group creation / destruction at the source: ``torch.distributed.init_process_group`` and
    ``destroy_process_group`` are wrapped once, so a group that the CALLER (or another library) creates and destroys with plain
    torch.distributed calls - no otpose_amd.parallel helper and no engine forward in between - is still seen by
    :func:`graph_replay_safe` (the flag used to be set only when one of this module's helpers ran while the group
    was alive).  Installed at ``import otpose_amd``; a ``from torch.distributed import destroy_process_group`` taken before that
    import bypasses it."""
    if getattr(dist, "_otpose_group_hooks", False) or not dist.is_available():
        return
    orig_init, orig_destroy = dist.init_process_group, dist.destroy_process_group

    def init_process_group(*a, **k):
        global _GROUP_SEEN
        r = orig_init(*a, **k)
        _GROUP_SEEN = True
        return r

    def destroy_process_group(*a, **k):
        global _GROUP_SEEN, _GROUP_DESTROYED
        if dist.is_initialized():
            _GROUP_SEEN = _GROUP_DESTROYED = True
        return orig_destroy(*a, **k)

    init_process_group.__doc__, destroy_process_group.__doc__ = orig_init.__doc__, orig_destroy.__doc__
    dist.init_process_group, dist.destroy_process_group = init_process_group, destroy_process_group
    dist._otpose_group_hooks = True


_install_group_hooks()


def _group_alive() -> bool:
    global _GROUP_SEEN
    if dist.is_available() and dist.is_initialized():
        _GROUP_SEEN = True
        return True
    return False


def graph_replay_safe() -> bool:
    """False once a process group this process worked with has been DESTROYED.  On torch 2.10 / ROCm 7.2 a hipGraph replay
    after an RCCL communicator was torn down segfaults inside the runtime (seen in the GPU suite: an inference engine's
    replay after a test that had destroyed its one-rank group), so :class:`otpose_amd.engine.InferenceEngine` asks here
    before every replay and launches its kernel list eagerly from then on - slower on the host, same kernels, same bits.
    The reference loop never gets there (train.py:74-99 validates after every epoch INSIDE the process, i.e. with the group
    alive: replays stay on); a caller that does tear the group down mid-process (tests, notebooks) stays alive.
    Use :func:`shutdown` to end a job: it leaves the group up until the last GPU work of the process is done."""
    return _group_alive() or not _GROUP_SEEN


def shutdown() -> None:
    """End of a data-parallel job: wait for the device, then destroy the default process group (if any).  Call it LAST -
    after the final validation pass - because graph replays are switched off for the rest of the process afterwards
    (:func:`graph_replay_safe`)."""
    if _group_alive():
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        if dist.get_world_size() > 1:                    # (a one-rank group has nobody to wait for: no collective on the way out)
            dist.barrier()
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        dist.destroy_process_group()


def world_size() -> int:
    return dist.get_world_size() if _group_alive() else 1


def collectives_on() -> bool:
    """True when the gradient / flag collectives must be issued: a world of more than one rank, or an initialised process
    group of ONE rank with ``OTPOSE_FORCE_COLLECTIVES=1`` - the latter sends every tensor through RCCL on a single GPU
    (sum over one rank = identity), so the device-side ordering of the exchange (side-stream gradients -> collective ->
    optimizer) runs on a 1-GPU box exactly as it does on eight."""
    if not _group_alive():
        return False
    return dist.get_world_size() > 1 or os.environ.get("OTPOSE_FORCE_COLLECTIVES") == "1"


def rank() -> int:
    return dist.get_rank() if _group_alive() else 0


def shard_range(n_clips: int, rank_: int, world: int) -> Tuple[int, int]:
    """Contiguous [begin, end) of the clips rank ``rank_`` owns; earlier ranks take the remainder."""
    if world <= 0 or not 0 <= rank_ < world:
        raise ValueError(f"bad rank/world {rank_}/{world}")
    base, rem = divmod(n_clips, world)
    begin = rank_ * base + min(rank_, rem)
    return begin, begin + base + (1 if rank_ < rem else 0)


def shard_clips(x: torch.Tensor, margin: torch.Tensor, rank_: Optional[int] = None, world: Optional[int] = None):
    """This rank's slice of a global batch: ``x`` (B,15,H,W), ``margin`` (B,4) -> views, no copy."""
    r = rank() if rank_ is None else rank_
    w = world_size() if world is None else world
    b, e = shard_range(x.shape[0], r, w)
    return x[b:e], margin[b:e]


def gather_clips(t: torch.Tensor, dst: int = 0) -> Optional[torch.Tensor]:
    """Concatenate equal-sized per-rank heat-map tensors on ``dst`` (what DataParallel's gather to GPU 0
    did, eval.py:113 / Common.py:357).  Returns None on the other ranks.  Not on the timed path."""
    w = world_size()
    if w == 1:
        return t
    parts = [torch.empty_like(t) for _ in range(w)]
    dist.all_gather(parts, t.contiguous())
    return torch.cat(parts, 0) if rank() == dst else None


def allreduce_joint_flags(flags: torch.Tensor) -> torch.Tensor:
    """MAX over ranks of the per-joint "ground truth has an exact-1 peak" flags (model/loss.py:47)."""
    if collectives_on():
        dist.all_reduce(flags, op=dist.ReduceOp.MAX)
    return flags


def allreduce_mean_(t: torch.Tensor) -> torch.Tensor:
    """In-place mean over ranks (loss scalars: the reference computes them on the gathered batch)."""
    if collectives_on():
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        t.div_(world_size())
    return t


def allreduce_flat_grads(optimizer) -> None:
    """Gradient exchange for :class:`otpose_amd.optim.FusedAdamW`: its gradients already live in one flat buffer per
    hyper-parameter group, so the exchange is one all-reduce(SUM) per group (three for ``make_optimizer``'s groups; the
    backbone group is 254 MB at W48 - few, large messages suit the point-to-point xGMI links) and a 1/world scale, with no
    packing or copy-back.  Call between ``loss.backward()`` and ``optimizer.step()``."""
    if not collectives_on():
        return
    w = world_size()
    grads = optimizer.flat_grads()
    if grads and grads[0].is_cuda:
        # the exchange runs after ``loss.backward()`` has returned (no overlap with the backward: at a 140-160 ms step the
        # 272 MB exchange is ~2-3 ms over xGMI).  Gradients of the HRNet branches were written on side streams; autograd syncs
        # them with the stream backward() was called from when it finishes, and waiting for them once more here keeps the
        # collective correct even when a caller drove the backward by hand.
        from .hip import _SIDE_STREAMS
        cur = torch.cuda.current_stream(grads[0].device)
        for s in [torch.cuda.default_stream(grads[0].device)] + _SIDE_STREAMS.get(grads[0].device, []):
            if s != cur:
                cur.wait_stream(s)
    works = [dist.all_reduce(g, op=dist.ReduceOp.SUM, async_op=True) for g in grads]
    for work, g in zip(works, grads):
        work.wait()
        if w > 1:
            g.div_(w)


def broadcast_buffers(module: torch.nn.Module, src: int = 0) -> None:
    """Replica ``src``'s buffers (BatchNorm running statistics) to every rank - the reference keeps
    replica 0's (DataParallel); done at checkpoint time, not per step."""
    if world_size() == 1:
        return
    for b in module.buffers():
        if b.is_floating_point():
            dist.broadcast(b, src)


class GradBuckets:
    """Flat-bucket gradient all-reduce.

    Parameters are packed, in reverse registration order (the order backward produces gradients), into
    flat buffers of about ``bucket_bytes``.  ``reduce()`` copies each gradient into its bucket, launches
    one asynchronous all-reduce(SUM) per bucket, then scales by 1/world and copies back.  With
    ``hooks=True`` every parameter gets a post-accumulate hook and a bucket is launched the moment its
    last gradient lands, overlapping RCCL traffic with the rest of backward; ``finish()`` then waits.
    """

    def __init__(self, params: Iterable[torch.nn.Parameter], bucket_bytes: int = DEFAULT_BUCKET_BYTES,
                 hooks: bool = False):
        self.params: List[torch.nn.Parameter] = [p for p in params if p.requires_grad]
        self.world = world_size()
        self.active = collectives_on()
        self.buckets: List[List[torch.nn.Parameter]] = []
        self.flat: List[torch.Tensor] = []
        cur: List[torch.nn.Parameter] = []
        cur_bytes = 0
        for p in reversed(self.params):
            nbytes = p.numel() * p.element_size()
            if cur and (cur_bytes + nbytes > bucket_bytes or p.dtype != cur[0].dtype or p.device != cur[0].device):
                self.buckets.append(cur)
                cur, cur_bytes = [], 0
            cur.append(p)
            cur_bytes += nbytes
        if cur:
            self.buckets.append(cur)
        for bk in self.buckets:
            self.flat.append(torch.zeros(sum(p.numel() for p in bk), dtype=bk[0].dtype, device=bk[0].device))
        self._where = {id(p): (i, j) for i, bk in enumerate(self.buckets) for j, p in enumerate(bk)}
        self._pending: List[int] = [len(bk) for bk in self.buckets]
        self._work: List[Optional[object]] = [None] * len(self.buckets)
        self._next = 0                      # hook mode: the next bucket allowed to launch (strict index order)
        self._handles = []
        if hooks and self.active:
            for p in self.params:
                self._handles.append(p.register_post_accumulate_grad_hook(self._on_grad))

    # ---- bookkeeping --------------------------------------------------------------------------
    def _offsets(self, i: int) -> Sequence[int]:
        offs, o = [], 0
        for p in self.buckets[i]:
            offs.append(o)
            o += p.numel()
        return offs

    def _pack_and_launch(self, i: int) -> None:
        flat = self.flat[i]
        if flat.is_cuda:
            # hook mode runs this on the stream of the gradient that completed the bucket; the other gradients of the bucket
            # may have been written on the side streams of the HRNet branches (train.TrainGraph.hr_module)
            from .hip import _SIDE_STREAMS
            cur = torch.cuda.current_stream(flat.device)
            for s in [torch.cuda.default_stream(flat.device)] + _SIDE_STREAMS.get(flat.device, []):
                if s != cur:
                    cur.wait_stream(s)
        for p, o in zip(self.buckets[i], self._offsets(i)):
            g = p.grad
            if g is None:
                flat[o:o + p.numel()].zero_()
            else:
                flat[o:o + p.numel()].copy_(g.reshape(-1))
        self._work[i] = dist.all_reduce(flat, op=dist.ReduceOp.SUM, async_op=True)

    def _on_grad(self, p: torch.nn.Parameter) -> None:
        i, _ = self._where[id(p)]
        if self._pending[i] <= 0:
            raise RuntimeError("GradBuckets: a second backward reached a bucket before finish(); gradient accumulation "
                               "needs hooks=False (call reduce() after the last backward)")
        self._pending[i] -= 1
        # collectives must be issued in the same order on every rank: bucket i goes out only after 0..i-1, whatever
        # order the gradients land in on this rank (a parameter without a gradient here must not reorder the launches)
        while self._next < len(self.buckets) and self._pending[self._next] == 0:
            self._pack_and_launch(self._next)
            self._next += 1

    # ---- public -------------------------------------------------------------------------------
    def reduce(self) -> None:
        """All-reduce every bucket now (no-hook mode) and write the averaged gradients back."""
        if not self.active:
            return
        for i in range(len(self.buckets)):
            if self._work[i] is None:
                self._pack_and_launch(i)
        self.finish()

    def finish(self) -> None:
        """Wait for the launched buckets, scale by 1/world, scatter back into ``p.grad``."""
        if not self.active:
            return
        for i, bk in enumerate(self.buckets):
            if self._work[i] is None:          # hook mode: a parameter of this bucket got no gradient
                self._pack_and_launch(i)       # (flushed here in index order, like the launches above)
            self._work[i].wait()
            flat = self.flat[i]
            flat.div_(self.world)
            for p, o in zip(bk, self._offsets(i)):
                if p.grad is None:
                    p.grad = flat[o:o + p.numel()].view_as(p).clone()
                else:
                    p.grad.copy_(flat[o:o + p.numel()].view_as(p))
            self._work[i] = None
            self._pending[i] = len(bk)
        self._next = 0

    def remove_hooks(self) -> None:
        for h in self._handles:
            h.remove()
        self._handles = []


def joint_flags(target: torch.Tensor) -> torch.Tensor:
    """Per-joint "the ground truth of this joint has an exact-1 peak somewhere in the batch" flags (model/loss.py:47:
    ``torch.max(heatmap_gt) == 1``) of this rank's (B, J, h, w) targets, as int32 (J,)."""
    return (target.amax(dim=(0, 2, 3)) == 1).to(torch.int32)


def train_step_dp(model, optimizer, x, margin, target, target_weight, forward=None, criterion=None, stats=None):
    """One data-parallel training step on this rank's clips - the reference's ``nn.DataParallel`` iteration
    (train.py:78-79, script/Common.py:118-144) as one process per GPU:

    forward (training mode) -> ``criterion(outputs, target, target_weight, flags)`` with the per-joint flags of
    model/loss.py:47 MAX-reduced over ranks, so every rank takes the branch the gathered global batch would take ->
    backward -> gradient all-reduce (mean) over RCCL (the optimizer's flat buffers when it has them, bucketed otherwise)
    -> ``optimizer.step()`` (global-norm clip on the already-reduced gradients, identical on every rank).

    ``forward(model, x, margin)`` and ``criterion(outputs, target, target_weight, flags)`` default to the HIP training
    graph (:mod:`otpose_amd.train`); the CPU tests pass their own.  Returns the rank-mean loss (detached).

    ``stats`` (a dict, optional): filled with ``comm_ms`` - the wall time of the gradient exchange, bracketed by device
    synchronisations, i.e. the time between the last backward kernel and the optimizer that a scaling run loses to the collective -
    its payload ``comm_bytes`` and ``overlap`` (what of the exchange runs under the backward pass: nothing in the flat-buffer form).
    Asking for it serialises the step; bench.py measures it in an extra, untimed step."""
    if forward is None or criterion is None:
        from . import train as _train
        forward = forward or _train.forward_train
        criterion = criterion or _train.criterion
    optimizer.zero_grad()
    outputs = forward(model, x, margin)
    flags = allreduce_joint_flags(joint_flags(target))
    loss = criterion(outputs, target, target_weight, flags)
    loss.backward()
    if stats is not None:
        import time
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        t0 = time.perf_counter()
    if hasattr(optimizer, "flat_grads"):
        allreduce_flat_grads(optimizer)
        nbytes = sum(int(g.numel()) * g.element_size() for g in optimizer.flat_grads()) if stats is not None else 0
    elif collectives_on():
        bk = getattr(optimizer, "_otp_buckets", None)
        if bk is None:
            bk = GradBuckets([p for g in optimizer.param_groups for p in g["params"]])
            optimizer._otp_buckets = bk
        bk.reduce()
        nbytes = sum(int(p.numel()) * p.element_size() for g in optimizer.param_groups for p in g["params"]) if stats is not None else 0
    else:
        nbytes = 0
    if stats is not None:
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        stats.update(comm_ms=1e3 * (time.perf_counter() - t0), comm_bytes=nbytes if collectives_on() else 0, world=world_size(),
                     overlap="none: the gradients are reduced after the backward pass (flat fp32 buffers / buckets in one go)")
    optimizer.step()
    return allreduce_mean_(loss.detach().clone())
