"""Batch sharding of clips over the GPUs of one node.

Clips are independent units (no state crosses clips; per-pixel state is private; the device RNG is keyed by the
GLOBAL clip id), so the batch dimension is partitioned with NO data-path collective: rank r simulates clips
[lo, hi) with clip_id0 = lo and gets bit-identical results for any world size.  torch.distributed (backend
"nccl" == RCCL on ROCm, "gloo" on CPU) is used only for the barrier and the max-over-ranks of the timing.

Reference context: the reference shards samples with DistributedSampler under DDP (train.py:54-56); its simulator
runs per sample inside DataLoader workers and exchanges nothing either.
"""
from __future__ import annotations

import os
from dataclasses import dataclass

import torch


@dataclass
class Shard:
    rank: int
    world: int
    lo: int          # first global clip id of this rank
    hi: int          # one past the last

    @property
    def count(self) -> int:
        return self.hi - self.lo


def shard_range(total: int, rank: int, world: int) -> Shard:
    """Contiguous, balanced partition of `total` clips: the first total % world ranks get one extra clip."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside world {world}")
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return Shard(rank, world, lo, lo + base + (1 if rank < rem else 0))


def weak_shard(clips_per_gpu: int, rank: int, world: int) -> Shard:
    """Weak scaling: every rank owns `clips_per_gpu` clips; global ids are rank-major."""
    return shard_range(clips_per_gpu * world, rank, world)


def env_rank_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def init_process_group(backend: str | None = None, device: torch.device | None = None):
    """One process per GPU, launched by torch.distributed.run; returns the module or None for world size 1.

    The data path never communicates: the process group only serves the barrier and the max-over-ranks of the timing.
    The backend is the one asked for ("nccl" == RCCL on a GPU box, "gloo" for CPU dry runs) on EVERY rank or the call
    raises: a rank that quietly fell back to another backend would leave its peers waiting inside an RCCL collective
    until the timeout.  Callers record `dist.get_backend()` next to their results."""
    _, _, world = env_rank_world()
    if world <= 1:
        return None
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    if dist.is_initialized():
        return dist
    if backend == "nccl":
        try:
            dist.init_process_group(backend="nccl", device_id=device)
        except TypeError:                          # older torch: no device_id argument
            dist.init_process_group(backend="nccl")
        probe = torch.zeros(1, device=device if device is not None else "cuda")
        dist.all_reduce(probe)                     # forces communicator creation now, not inside a timed region; raises on failure
        torch.cuda.synchronize()
        return dist
    dist.init_process_group(backend=backend)
    return dist


def barrier(dist, local_rank: int = 0):
    if dist is not None:
        if dist.get_backend() == "nccl":
            try:
                dist.barrier(device_ids=[local_rank])
            except TypeError:
                dist.barrier()
        else:
            dist.barrier()
    if torch.cuda.is_available():
        torch.cuda.synchronize()


def max_over_ranks(dist, value: float, device=None) -> float:
    if dist is None:
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device if dist.get_backend() == "nccl" else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_floats(dist, value: float, device=None) -> list:
    """Every rank's `value`, in rank order, on every rank (timing bookkeeping only; the data path exchanges nothing)."""
    if dist is None:
        return [value]
    dev = device if dist.get_backend() == "nccl" else "cpu"
    mine = torch.tensor([value], dtype=torch.float64, device=dev)
    out = [torch.zeros_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(out, mine)
    return [float(t.item()) for t in out]
