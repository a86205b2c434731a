"""Batch-dimension sharding of the forecaster forward across the GPUs of one node.

Batch elements never interact in the reference forward (disconnected graph copies, encoder.py:212-218; per-sample
loops in the efficient path, encoder.py:171-187, processor.py:113-118), so N GPUs run N independent shards: weights,
graph plans and cached batch-independent embeddings are replicated, rank ``k`` owns a contiguous slice of the global
batch, and the forward needs **no data-path collective**.  The only communication is control-plane: a barrier around
the timed region and a MAX-reduction of the elapsed time (RCCL over xGMI on GPUs - torch backend "nccl" - and gloo
in the CPU tests).

One process per GPU, launched by ``python -m torch.distributed.run --nproc-per-node N ...``; RANK / LOCAL_RANK /
WORLD_SIZE / MASTER_* come from the environment.
"""
from __future__ import annotations

import os
import time
from dataclasses import dataclass
from typing import Callable, Optional, Tuple

import torch


@dataclass
class ShardContext:
    rank: int
    local_rank: int
    world: int
    backend: Optional[str]  # None when world == 1 (no process group)

    @property
    def is_root(self) -> bool:
        return self.rank == 0


def shard_range(global_batch: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous slice [lo, hi) of a global batch owned by ``rank``; sizes differ by at most one, earlier ranks
    take the remainder (GPU k of N gets features[k*B/N:(k+1)*B/N] when N divides B)."""
    if world <= 0 or not (0 <= rank < world) or global_batch < 0:
        raise ValueError("bad shard arguments")
    base, rem = divmod(global_batch, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def init_from_env(backend: Optional[str] = None, device: Optional[torch.device] = None) -> ShardContext:
    """Join the process group described by the torchrun environment (no-op for WORLD_SIZE == 1)."""
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world == 1:
        return ShardContext(rank, local_rank, world, None)
    import torch.distributed as dist

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    if not dist.is_initialized():
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=device)
        else:
            dist.init_process_group(backend=backend)
    return ShardContext(rank, local_rank, world, backend)


def barrier(ctx: ShardContext, device: Optional[torch.device] = None) -> None:
    """Process-group barrier followed by a device synchronise (both sides of a timed region)."""
    if ctx.world > 1:
        import torch.distributed as dist

        dist.barrier()
    if device is not None and device.type == "cuda":
        torch.cuda.synchronize(device)


def max_over_ranks(ctx: ShardContext, value: float, device: Optional[torch.device] = None) -> float:
    if ctx.world == 1:
        return float(value)
    import torch.distributed as dist

    t = torch.tensor([value], dtype=torch.float64, device=device if ctx.backend == "nccl" else None)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(ctx: ShardContext, value: float, device: Optional[torch.device] = None) -> float:
    if ctx.world == 1:
        return float(value)
    import torch.distributed as dist

    t = torch.tensor([value], dtype=torch.float64, device=device if ctx.backend == "nccl" else None)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def timed_steps(ctx: ShardContext, step: Callable[[], None], steps: int, warmup: int,
                device: Optional[torch.device] = None, after_warmup: Optional[Callable[[], None]] = None) -> float:
    """``warmup`` untimed calls of ``step``, then exactly ``steps`` calls bracketed by barrier + synchronise on
    both sides; returns the MAX over ranks of the elapsed wall time in seconds."""
    for _ in range(warmup):
        step()
    if after_warmup is not None:
        after_warmup()
    barrier(ctx, device)
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    barrier(ctx, device)
    return max_over_ranks(ctx, time.perf_counter() - t0, device)


def whole_job_rate(ctx: ShardContext, units_this_rank: float, elapsed_max: float,
                   device: Optional[torch.device] = None) -> float:
    """Units processed by ALL ranks divided by the slowest rank's time."""
    return sum_over_ranks(ctx, units_this_rank, device) / elapsed_max


class FlatGradients:
    """Parameters and gradients of a model as views into two persistent flat fp32 buffers (data-parallel training step,
    SURVEY.md 8e / train/run.py:509-521).

    * every ``p.grad`` is a view of ``self.grad`` and stays one: autograd accumulates into it in place, ``zero_()`` is a
      single fill, nothing is concatenated or copied back around the collective;
    * every ``p.data`` is (by default) a view of ``self.param``, so the optimizer updates the whole model with ONE kernel
      launch over the flat buffers (``graph_weather_amd.AdamW(..., flat=...)``) instead of one per tensor;
    * gradients are all-reduced in buckets laid out in REVERSE parameter order - the order the backward produces them -
      and each bucket's collective is launched (asynchronously: RCCL over xGMI on GPUs, gloo in the CPU tests) from a
      post-accumulate hook as soon as its last gradient has been written, i.e. while the backward of the earlier layers is
      still running; ``allreduce()`` after ``backward()`` only waits and averages.
    xGMI rings are per-link bound, so the buckets are few and large (default 8 MiB: 4 collectives for the 30.9 MB model)."""

    def __init__(self, params, bucket_bytes: int = 8 << 20, flatten_params: bool = True):
        self.params = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("FlatGradients: no trainable parameters")
        dev, dt = self.params[0].device, self.params[0].dtype
        if any(p.device != dev or p.dtype != dt for p in self.params):
            raise ValueError("FlatGradients: parameters must share one device and dtype")
        self.offsets, off = [], 0
        for p in self.params:
            self.offsets.append(off)
            off += (p.numel() + 3) // 4 * 4  # 16-byte aligned slices
        self.numel = off
        self.grad = torch.zeros(off, dtype=dt, device=dev)
        self.param = None
        with torch.no_grad():
            if flatten_params:
                self.param = torch.zeros(off, dtype=dt, device=dev)
                for p, o in zip(self.params, self.offsets):
                    view = self.param[o:o + p.numel()].view_as(p)
                    view.copy_(p)
                    p.data = view
            for p, o in zip(self.params, self.offsets):
                p.grad = self.grad[o:o + p.numel()].view_as(p)
        # buckets over the flat buffer, from its tail (last parameters = first gradients of the backward) to its head
        self.buckets = []  # (lo, hi, n_params)
        hi, n, self._bucket_of = off, 0, [0] * len(self.params)
        for i in range(len(self.params) - 1, -1, -1):
            self._bucket_of[i] = len(self.buckets)
            n += 1
            if (hi - self.offsets[i]) * self.grad.element_size() >= bucket_bytes or i == 0:
                self.buckets.append((self.offsets[i], hi, n))
                hi, n = self.offsets[i], 0
        self._pending = [b[2] for b in self.buckets]
        self._handles = [None] * len(self.buckets)
        self._ctx: Optional[ShardContext] = None
        self._sync = True  # False inside no_sync(): gradients accumulate locally, no collective is launched
        self.collectives = 0
        for i, p in enumerate(self.params):
            p.register_post_accumulate_grad_hook(self._make_hook(i))

    def _rearm(self) -> None:
        self._pending = [b[2] for b in self.buckets]
        self._handles = [None] * len(self.buckets)

    def _make_hook(self, i: int):
        def hook(param):
            if not self._sync:  # accumulation step (no_sync): the gradient stays local
                return
            if self._ctx is None or self._ctx.world <= 1:
                # one process (or not attached): nothing is launched from the hooks, so there is no in-flight collective a
                # second backward could corrupt - local accumulation and zero_grad(set_to_none=False) loops just work
                return
            b = self._bucket_of[i]
            self._pending[b] -= 1
            if self._pending[b] < 0:
                # a second backward() into an armed step would add local gradients onto buckets whose all-reduce is already
                # in flight (or done): the ranks would diverge silently
                raise RuntimeError("graph_weather_amd.FlatGradients: a parameter received a second gradient in one step - run "
                                   "every backward() but the last under flat.no_sync() (gradient accumulation), and call "
                                   "flat.allreduce() / flat.zero_() between steps")
            if self._pending[b] == 0 and self._handles[b] is None:
                self._launch(b)
        return hook

    def no_sync(self):
        """Context for gradient accumulation (torch DDP's ``no_sync``): backward passes inside it only accumulate into the flat
        buffer; the first backward outside it launches the bucket all-reduces as usual."""
        flat = self

        class _NoSync:
            def __enter__(self_inner):
                flat._sync = False
                return flat

            def __exit__(self_inner, *exc):
                flat._sync = True
                flat._rearm()  # the next backward sees full buckets again
                return False

        return _NoSync()

    def _launch(self, b: int) -> None:
        import torch.distributed as dist

        lo, hi, _ = self.buckets[b]
        self._handles[b] = dist.all_reduce(self.grad[lo:hi], op=dist.ReduceOp.SUM, async_op=True)
        self.collectives += 1

    def attach(self, ctx: ShardContext) -> "FlatGradients":
        """Arm the overlap: from now on a bucket's all-reduce starts inside ``backward()`` when its last gradient lands."""
        self._ctx = ctx
        return self

    def views_intact(self) -> bool:
        """False when something replaced a ``p.grad`` / ``p.data`` view (``zero_grad(set_to_none=True)``, ``model.to(...)``)."""
        for p, o in zip(self.params, self.offsets):
            if p.grad is None or p.grad.data_ptr() != self.grad.data_ptr() + o * self.grad.element_size():
                return False
            if self.param is not None and p.data_ptr() != self.param.data_ptr() + o * self.param.element_size():
                return False
        return True

    def zero_(self) -> None:
        """One fill for every gradient; re-arms the bucket counters for the next backward."""
        self.grad.zero_()
        self._rearm()

    def allreduce(self, ctx: ShardContext) -> int:
        """Average the gradients over the ranks: launch the buckets the backward has not already launched, wait for all of
        them, scale by 1 / world (one kernel over the flat buffer).  Returns the number of collectives of this step."""
        if ctx.world == 1:
            self._rearm()  # (nothing was counted or launched; keeps the per-step state well defined all the same)
            return 0
        self._ctx = ctx
        n = 0
        for b in range(len(self.buckets)):
            if self._handles[b] is None:
                self._launch(b)
            n += 1
        for h in self._handles:
            h.wait()
        self.grad.div_(ctx.world)
        # the step's collectives are done: stale (completed) handles must not make the next allreduce() skip its launches when
        # the caller zeroes gradients some other way (optimizer.zero_grad(set_to_none=False)) instead of zero_()
        self._rearm()
        return n


def allreduce_gradients(ctx: ShardContext, params, bucket_bytes: int = 64 << 20) -> int:
    """Gradient averaging for parameters that are NOT views of a ``FlatGradients`` buffer (generic fallback; the training
    bench uses ``FlatGradients``): one all-reduce per flat bucket, gradients copied in and out.
    Returns the number of collectives issued (0 for world == 1)."""
    if ctx.world == 1:
        return 0
    import torch.distributed as dist

    grads = [p.grad for p in params if p.grad is not None]
    n_coll, bucket, size = 0, [], 0

    def flush():
        nonlocal n_coll, bucket, size
        if not bucket:
            return
        flat = torch.cat([g.reshape(-1) for g in bucket])
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        flat.div_(ctx.world)
        off = 0
        for g in bucket:
            g.copy_(flat[off:off + g.numel()].view_as(g))
            off += g.numel()
        n_coll += 1
        bucket, size = [], 0

    for g in grads:
        bucket.append(g)
        size += g.numel() * g.element_size()
        if size >= bucket_bytes:
            flush()
    flush()
    return n_coll


def shutdown(ctx: ShardContext) -> None:
    if ctx.world > 1:
        import torch.distributed as dist

        if dist.is_initialized():
            dist.destroy_process_group()
