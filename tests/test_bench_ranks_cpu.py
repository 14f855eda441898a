"""bench.py's rank path on CPU (gloo, world size 2): the very code the driver launches with `python -m torch.distributed.run ...
bench.py --gpus N`, with `--dry` swapping the device launch for a no-op.  Covers the config-5 sharding (2^20 QPs in total,
BASELINE.json configs[4]), the barriers, the MAX / SUM reductions and the JSON line; no GPU, no oracle."""
import json
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run(world, extra):
    env = dict(os.environ, MO_BENCH_BACKEND="gloo")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", "3", "--warmup", "1",
           "--dry"] + extra
    res = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout      # rank 0 prints exactly one JSON line
    return json.loads(lines[0])


def test_two_ranks_default_to_config5_sharded():
    out = _run(2, [])
    assert out["dry"] is True and out["value"] is None
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["warmup"] == 1
    assert out["scaling"] == "strong" and out["config"]["name"] == "cfg5"
    assert "configs[4]" in out["config"]["workload"]
    assert out["config"]["batch_total"] == 2 ** 20 and out["config"]["batch_per_gpu"] == 2 ** 19
    assert out["units_per_step_all_ranks"] == 2 ** 20      # SUM over ranks of the shard sizes: the whole of config 5
    assert (out["config"]["n"], out["config"]["k"], out["config"]["m"], out["config"]["m_r"]) == (64, 8, 32, 128)
    assert out["dtype"] == "f64" and out["unit"] == "steps/s" and out["higher_is_better"] is True
    # the correctness figures of a multi-rank line: every rank contributes a sample of ITS shard, MAX / SUM over the control-plane group
    # (here the reduction runs on placeholders: --dry launches and checks nothing, `passed` says so)
    par = out["parity"]
    assert par["ranks"] == 2 and par["sample_per_rank"] == 256 and par["sample"] == 512 and par["dry"] is True and par["passed"] is None
    assert {"max_rel_inf", "p999_rel_inf", "tolerance", "status_disagreements"} <= set(par)
    assert out["status_ok"] == out["status_total"] == 2 ** 20     # SUM over ranks of the per-rank status counts


def test_two_ranks_solve_mode_schema():
    """`--mode solve`: one mo_qp_solve launch per step, unit solves/s, same line contract and the same all-rank parity block."""
    out = _run(2, ["--mode", "solve", "--config", "cfg3", "--batch", "4096", "--parity-sample", "64"])
    assert out["unit"] == "solves/s" and out["config"]["mode"] == "solve" and "mo_qp_solve" in out["config"]["workload"]
    assert out["config"]["batch_total"] == 8192 and out["units_per_step_all_ranks"] == 8192
    assert out["parity"]["sample"] == 128 and "solve_disagreements" in out["parity"]
    assert out["status_ok"] == out["status_total"] == 8192


def test_two_ranks_weak_scaling_config():
    out = _run(2, ["--config", "cfg3"])
    assert out["scaling"] == "weak" and out["config"]["batch_per_gpu"] == 65536 and out["config"]["batch_total"] == 2 * 65536
    assert "configs[2]" in out["config"]["workload"]


def _clean_env(**extra):
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(extra)
    return env


def test_bare_gpus_flag_starts_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher (what the driver types for the scaling run): the parent spawns the ranks as child
    processes and relays rank 0's one JSON line."""
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--dry"],
                         cwd=ROOT, env=_clean_env(MO_BENCH_BACKEND="gloo"), capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1 and lines[0].startswith("{"), res.stdout     # stdout carries the JSON line and nothing else
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["dry"] is True
    assert out["config"]["name"] == "cfg5" and out["config"]["batch_total"] == 2 ** 20 and out["config"]["batch_per_gpu"] == 2 ** 19
    assert out["units_per_step_all_ranks"] == 2 ** 20


def test_the_launching_parent_never_touches_torch():
    """The parent of a bare `--gpus N` run must not initialise HIP: it does not even import torch (checked through an import hook that
    only the parent process sees -- the ranks are started with a clean interpreter)."""
    code = ("import sys, runpy\n"
            "class Block:\n"
            "    def find_spec(self, name, path=None, target=None):\n"
            "        if name == 'torch' or name.startswith('torch.'):\n"
            "            raise ImportError('the launching parent imported ' + name)\n"
            "sys.meta_path.insert(0, Block())\n"
            f"sys.argv = [{os.path.join(ROOT, 'bench.py')!r}, '--gpus', '2', '--steps', '2', '--warmup', '0', '--dry']\n"
            f"runpy.run_path({os.path.join(ROOT, 'bench.py')!r}, run_name='__main__')\n")
    res = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=_clean_env(MO_BENCH_BACKEND="gloo"), capture_output=True,
                         text=True, timeout=300)
    assert res.returncode == 0, res.stderr[-2000:]
    assert json.loads([ln for ln in res.stdout.splitlines() if ln.startswith("{")][0])["n_gpus"] == 2


def test_a_failing_rank_fails_the_bare_run():
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--dry"],
                         cwd=ROOT, env=_clean_env(MO_BENCH_BACKEND="no-such-backend"), capture_output=True, text=True, timeout=300)
    assert res.returncode != 0 and not [ln for ln in res.stdout.splitlines() if ln.startswith("{")]


def test_world_size_that_disagrees_with_gpus_is_an_error():
    env = _clean_env(WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--dry"], cwd=ROOT, env=env,
                         capture_output=True, text=True, timeout=120)
    assert res.returncode != 0 and "WORLD_SIZE" in res.stderr
