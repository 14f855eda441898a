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


def timing_group(info: RankInfo, want: str, device=None, timeout_s: float = 120.0):
    """The group that carries the timing barrier and the two scalar reductions: (group, label, device for the scalars).

    The default process group (`init_process_group(info, "gloo")`) is the control plane and always works.  With `want == "nccl"` (RCCL on
    ROCm -- the driver's one-rank-per-GPU runs) an RCCL group is created on top and probed with one all-reduce; every rank then reports over
    gloo whether ITS probe worked, and RCCL is used only if it worked everywhere -- otherwise ALL ranks fall back to gloo together (no rank
    is left waiting in a collective the others have given up on).  The path has no data-path collective, so which library carries the
    barrier changes nothing that is measured; the label says which one it was."""
    import torch.distributed as dist
    if info.world_size == 1 or want != "nccl":
        return None, want if info.world_size > 1 else "none", None
    import datetime

    import torch

    def all_agree(ok: int) -> bool:   # over gloo (the default group): did every rank get this far?
        flag = torch.tensor([ok], dtype=torch.int32)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        return int(flag.item()) == 1

    # Step 1: every rank creates the group (new_group is collective over the default group and lazy about the communicator), then all
    # ranks agree over gloo that they have one -- a rank on which creation fails must not leave the others alone in an RCCL collective.
    ok, group, why = 1, None, ""
    try:
        group = dist.new_group(backend="nccl", timeout=datetime.timedelta(seconds=timeout_s))
    except Exception as exc:
        ok, why = 0, repr(exc)[:200]
    if not all_agree(ok):
        return None, "gloo (rccl unavailable" + (": " + why if why else " on another rank") + ")", None
    # Step 2: the probe collective, asynchronous with a bounded wait -- a rank whose communicator cannot form (two ranks on one GPU in a
    # rehearsal, no IPC between the devices) reports over gloo instead of sitting in the collective until the watchdog aborts the process.
    try:
        probe = torch.ones(1, device=device)
        work = dist.all_reduce(probe, group=group, async_op=True)
        if not work.wait(timeout=datetime.timedelta(seconds=min(timeout_s, 60.0))):
            raise RuntimeError("probe all-reduce timed out")
        torch.cuda.synchronize(device)
        if int(probe.item()) != info.world_size:
            raise RuntimeError(f"probe all-reduce returned {probe.item()}")
    except Exception as exc:
        ok, why = 0, repr(exc)[:200]
    if all_agree(ok):
        return group, "rccl", device
    return None, "gloo (rccl unavailable" + (": " + why if why else " on another rank") + ")", None


def barrier_max_sum(info: RankInfo, elapsed_s: float, units: int, device=None, group=None) -> tuple[float, int]:
    """Max of the per-rank wall time and sum of the per-rank unit counts (scalars only)."""
    if info.world_size == 1:
        return elapsed_s, units
    import torch
    import torch.distributed as dist
    t = torch.tensor([elapsed_s], dtype=torch.float64, device=device)
    u = torch.tensor([units], dtype=torch.int64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    dist.all_reduce(u, op=dist.ReduceOp.SUM, group=group)
    return float(t.item()), int(u.item())


def reduce_scalars(info: RankInfo, maxima=(), sums=()) -> dict:
    """MAX of `maxima` and SUM of `sums` over all ranks, over the DEFAULT (control-plane, gloo) group: the correctness figures of a
    multi-rank bench line (parity errors, status words) never depend on RCCL.  Returns {"maxima": [...], "sums": [...]}."""
    if info.world_size == 1:
        return {"maxima": [float(v) for v in maxima], "sums": [int(v) for v in sums]}
    import torch
    import torch.distributed as dist
    mx = torch.tensor([float(v) for v in maxima] or [0.0], dtype=torch.float64)
    sm = torch.tensor([int(v) for v in sums] or [0], dtype=torch.int64)
    dist.all_reduce(mx, op=dist.ReduceOp.MAX)
    dist.all_reduce(sm, op=dist.ReduceOp.SUM)
    return {"maxima": [float(v) for v in mx.tolist()][:len(maxima)], "sums": [int(v) for v in sm.tolist()][:len(sums)]}
