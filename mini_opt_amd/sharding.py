"""Multi-GPU sharding of a batch of independent QPs.

Every QP is independent (the reference solves them in a serial loop re-using one solver, test/qp_test.cc:531-567), so
the batch is split into contiguous shards, one per rank / GPU, with NO data-path collective: the only cross-rank
traffic is the timing barrier and a MAX/SUM of scalars.
"""
from __future__ import annotations

import os
from dataclasses import dataclass


@dataclass(frozen=True)
class RankInfo:
    rank: int
    local_rank: int
    world_size: int

    @staticmethod
    def from_env() -> "RankInfo":
        return RankInfo(int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
                        int(os.environ.get("WORLD_SIZE", "1")))


def shard_range(total: int, rank: int, world_size: int) -> tuple[int, int]:
    """Contiguous [begin, end) of `total` problems owned by `rank` (sizes differ by at most one)."""
    if world_size <= 0 or not (0 <= rank < world_size):
        raise ValueError("bad rank / world_size")
    base, rem = divmod(total, world_size)
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


def init_process_group(info: RankInfo, backend: str):
    """One process per GPU; rendezvous on 127.0.0.1 (MASTER_ADDR / MASTER_PORT from the launcher)."""
    import torch.distributed as dist
    if info.world_size > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        dist.init_process_group(backend=backend, rank=info.rank, world_size=info.world_size)
    return dist


def barrier_max_sum(info: RankInfo, elapsed_s: float, units: int, device=None) -> tuple[float, int]:
    """Max of the per-rank wall time and sum of the per-rank unit counts (host-side scalars only)."""
    if info.world_size == 1:
        return elapsed_s, units
    import torch
    import torch.distributed as dist
    t = torch.tensor([elapsed_s], dtype=torch.float64, device=device)
    u = torch.tensor([units], dtype=torch.int64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.all_reduce(u, op=dist.ReduceOp.SUM)
    return float(t.item()), int(u.item())
