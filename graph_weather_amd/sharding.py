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


def allreduce_gradients(ctx: ShardContext, params, bucket_bytes: int = 64 << 20) -> int:
    """Data-parallel gradient step: average the gradients of ``params`` over the ranks with one all-reduce
    per flat bucket (RCCL over xGMI on GPUs, gloo in the CPU tests).  The whole model is 7.7 M fp32 gradients = 30.9 MB,
    i.e. a single bucket: xGMI rings are per-link bound, so fewer, larger collectives are the right shape.
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
