"""bench.py's multi-rank path with REAL launches: two ranks started by torch.distributed.run share the one GPU of the test box
(ranks wrap around the visible devices), gloo stands in for RCCL (two ranks on one device cannot form an RCCL communicator).  Everything
else is the code the driver runs at N > 1: config-5 sharding, per-rank plans and launches, barriers, MAX / SUM, the roofline block."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_two_ranks_on_one_gpu_run_config5_shards():
    env = dict(os.environ, MO_BENCH_BACKEND="gloo")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2",
           "--config", "cfg5", "--batch", "16384"]
    res = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout
    out = json.loads(lines[0])
    # the N > 1 line carries its own correctness figure: both ranks checked a strided 256-problem sample of their shard against the oracle,
    # reduced over the gloo control plane; no cpu_baseline at N > 1
    par = out["parity"]
    assert par["passed"] is True and par["ranks"] == 2 and par["sample"] == 512 and par["max_rel_inf"] < 1e-10 and par["status_disagreements"] == 0
    assert out["status_ok"] == out["status_total"] == out["config"]["batch_total"] == 32768
    assert "cpu_baseline" not in out
    assert out["n_gpus"] == 2 and out["steps"] == 4 and out["scaling"] == "strong"
    assert out["config"]["name"] == "cfg5" and out["config"]["batch_per_gpu"] == 16384 and out["config"]["batch_total"] == 32768
    assert out["config"]["kernel"] == "fused_mfma_f64_n64"
    assert out["value"] > 1e6 and out["unit"] == "steps/s"                       # whole-job rate of both ranks
    assert abs(out["value"] * out["ms_per_step"] * 1e-3 - 32768) < 1e-3 * 32768  # value = units of all ranks / max time
    assert out["roofline"]["bound"] == "hbm" and 0 < out["roofline"]["frac"] < 1
    assert out["status_ok"] == out["status_total"]


def test_bare_gpus_flag_runs_two_real_ranks():
    """The driver's scaling command is a bare `python bench.py --gpus N`: the parent (which never touches the GPU) starts the ranks as child
    processes and relays rank 0's line.  Here N = 2 on the box's one GPU with gloo standing in for RCCL."""
    env = dict(os.environ, MO_BENCH_BACKEND="gloo")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2", "--config", "cfg5",
           "--batch", "16384", "--no-cpu-baseline", "--sustain-seconds", "0.3"]
    res = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1 and lines[0].startswith("{"), res.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["batch_total"] == 32768 and out["config"]["kernel"] == "fused_mfma_f64_n64"
    assert out["value"] > 1e6 and out["status_ok"] == out["status_total"]
    assert out["sustained"]["seconds"] >= 0.3 and out["sustained"]["value"] > 1e6


def test_two_ranks_on_one_gpu_fall_back_from_rccl_to_gloo():
    """Without MO_BENCH_BACKEND the ranks ask for RCCL.  Two ranks on ONE device cannot form an RCCL communicator: both must notice, agree over
    the gloo control plane to time over gloo instead, and the run must still produce its line (a driver-run `--gpus N` never exits non-zero
    because of the library that carries its barrier)."""
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "MO_BENCH_BACKEND"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--config", "cfg5",
           "--batch", "8192", "--no-cpu-baseline", "--sustain-seconds", "0"]
    res = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1 and lines[0].startswith("{"), res.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["batch_total"] == 16384 and out["status_ok"] == out["status_total"]
    assert out["timing_sync"] == "rccl" or out["timing_sync"].startswith("gloo (rccl unavailable"), out["timing_sync"]


def test_solve_mode_two_ranks_and_single_rank():
    """`bench.py --mode solve` (one mo_qp_solve launch per step): two ranks on the one GPU with the all-rank parity block -- termination state and
    iteration count of every checked problem equal to the oracle's Solve, optimum within 1e-6 -- and the single-rank line with the flop roofline
    and the oracle's Solve as cpu_baseline."""
    env = dict(os.environ, MO_BENCH_BACKEND="gloo")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--mode", "solve", "--steps", "3", "--warmup", "1", "--config", "cfg3",
           "--batch", "8192", "--sustain-seconds", "0"]
    res = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-3000:]
    out = json.loads([ln for ln in res.stdout.splitlines() if ln.startswith("{")][0])
    assert out["unit"] == "solves/s" and out["n_gpus"] == 2 and out["config"]["kernel"] == "fused_solve_mfma_f64_n64"
    assert out["parity"]["passed"] is True and out["parity"]["sample"] == 512 and out["parity"]["solve_disagreements"] == 0
    assert out["status_ok"] == out["status_total"] == 16384
    assert out["roofline"]["bound"] == "mfma" and out["roofline"]["unit"] == "TFLOP/s" and 0 < out["roofline"]["frac"] < 1
    assert 6.0 < out["roofline"]["mean_iterations"] < 10.0
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--mode", "solve", "--steps", "3", "--warmup", "1", "--batch", "4096", "--sustain-seconds", "0",
           "--cpu-seconds", "2"]
    res = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-3000:]
    out = json.loads([ln for ln in res.stdout.splitlines() if ln.startswith("{")][0])
    assert out["n_gpus"] == 1 and out["parity"]["passed"] is True and out["parity"]["sample"] == 4096      # the whole batch at N = 1
    assert out["cpu_baseline"]["unit"] == "solves/s" and out["cpu_baseline"]["value"] > 0 and out["cpu_baseline"]["one_core"]["cores"] == 1
    assert out["value"] > 20 * out["cpu_baseline"]["value"]
