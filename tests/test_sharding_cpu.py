"""Multi-GPU path on CPU: the batch shards over ranks with no data-path collective; only the timing barrier and a
MAX / SUM of scalars cross ranks (gloo, world_size 2).  The per-rank 'step' here is the CPU oracle standing in for the
device kernel (tests may use the oracle), checked against a single-rank run."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from mini_opt_amd import sharding, synth


def test_shard_range_partitions_exactly():
    for total in (0, 1, 7, 64, 65536, 2 ** 20 + 3):
        for world in (1, 2, 3, 8):
            ranges = [sharding.shard_range(total, r, world) for r in range(world)]
            assert ranges[0][0] == 0 and ranges[-1][1] == total
            for (a0, a1), (b0, b1) in zip(ranges, ranges[1:]):
                assert a1 == b0
            sizes = [b - a for a, b in ranges]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        sharding.shard_range(10, 2, 2)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, total, out_dir):
    from oracle import oracle as orc
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    info = sharding.RankInfo.from_env()
    sharding.init_process_group(info, "gloo")
    d = synth.CONFIGS["cfg1"]
    hb = synth.make_batch(d["n"], d["k"], d["m"], d["m_r"], total, stream=99)  # every rank regenerates the same batch
    b0, b1 = sharding.shard_range(total, rank, world)
    sl = slice(b0, b1)
    delta, alpha, status, _ = orc.batched_newton_step(
        hb.n, hb.k, hb.m, J=hb.J[sl], r=hb.r[sl], lam=hb.lam, A_eq=hb.A_eq[sl], b_eq=hb.b_eq[sl], cons_var=hb.cons_var[sl],
        cons_a=hb.cons_a[sl], cons_b=hb.cons_b[sl], vars_=hb.vars[sl], mu=hb.mu[sl], num_threads=1)
    np.save(os.path.join(out_dir, f"delta_{rank}.npy"), delta)
    dist.barrier()
    t_max, units = sharding.barrier_max_sum(info, 0.5 + rank, b1 - b0)
    if rank == 0:
        np.save(os.path.join(out_dir, "agg.npy"), np.array([t_max, units]))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharded_step_matches_single_rank(tmp_path):
    from oracle import oracle as orc
    total, world = 37, 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, total, str(tmp_path)), nprocs=world, join=True)
    got = np.concatenate([np.load(tmp_path / f"delta_{r}.npy") for r in range(world)])
    d = synth.CONFIGS["cfg1"]
    hb = synth.make_batch(d["n"], d["k"], d["m"], d["m_r"], total, stream=99)
    ref, _, status, _ = orc.batched_newton_step(hb.n, hb.k, hb.m, J=hb.J, r=hb.r, lam=hb.lam, A_eq=hb.A_eq, b_eq=hb.b_eq,
                                                cons_var=hb.cons_var, cons_a=hb.cons_a, cons_b=hb.cons_b, vars_=hb.vars, mu=hb.mu,
                                                num_threads=1)
    assert np.all(status == 0)
    np.testing.assert_array_equal(got, ref)  # sharding changes nothing: problems are independent
    agg = np.load(tmp_path / "agg.npy")
    assert agg[0] == 1.5 and agg[1] == total  # MAX of the per-rank times, SUM of the per-rank unit counts


def _timing_worker(rank, world, port, out_dir):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    info = sharding.RankInfo.from_env()
    sharding.init_process_group(info, "gloo")
    group, label, dev = sharding.timing_group(info, "nccl", device=None, timeout_s=30.0)   # no GPU here: RCCL cannot come up on any rank
    dist.barrier(group=group)
    t_max, units = sharding.barrier_max_sum(info, 1.0 + rank, 10 + rank, dev, group)
    if rank == 0:
        with open(os.path.join(out_dir, "timing.txt"), "w") as f:
            f.write(f"{group is None}|{label}|{dev}|{t_max}|{units}")
    dist.barrier()
    dist.destroy_process_group()


def test_timing_group_falls_back_to_gloo_on_every_rank_when_rccl_is_unavailable(tmp_path):
    """bench.py asks for an RCCL group for its timing barrier; where RCCL cannot form one (here: no GPU at all) every rank must agree on gloo --
    the fallback a driver-run `bench.py --gpus N` would take rather than exit non-zero -- and the reductions must still be right."""
    world = 2
    mp.spawn(_timing_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    is_none, label, dev, t_max, units = open(tmp_path / "timing.txt").read().split("|")
    assert is_none == "True" and dev == "None" and label.startswith("gloo (rccl unavailable")
    assert float(t_max) == 2.0 and int(units) == 21
    # one rank, or gloo asked for: no probe at all
    assert sharding.timing_group(sharding.RankInfo(0, 0, 1), "nccl") == (None, "none", None)


def test_synthetic_generator_rejects_shapes_it_cannot_build():
    """synth.make_batch_torch puts a lower and an upper bound on each of m / 2 DISTINCT variables: m must be even and <= 2 n.  The check
    runs on the host before any device op (an out-of-range gather on the device once faulted a GPU box); on a machine without a GPU it is
    reached before the first tensor is created."""
    import pytest
    from mini_opt_amd import synth
    for n, m in ((8, 18), (4, 10), (8, 5), (3, 7)):
        with pytest.raises(ValueError, match="even m <= 2 n"):
            synth.make_batch_torch(n, 2, m, 16, 4, "cpu", None, seed=1)
