#!/usr/bin/env python3
"""bench.py -- batched dense KKT Newton steps/s on MI355X (BASELINE.json metric).

A "step" = one pass of the hot path (mo_newton_step through the C ABI) over one batch of synthetic QPs that is already
resident in HBM.  N=1 workload: BASELINE.json configs[2] (batch 65536, n=64 / 8 eq / 32 box, fp64, J-level input).
N>1: every rank owns its own batch of the same size on its own GPU (weak scaling, no data-path collective -- the batch
shards embarrassingly); value = problems all ranks processed / max-over-ranks wall time.

Prints ONE JSON line on rank 0 with `roofline` (dominant kernel: algorithmic bytes per launch / average launch duration
measured with events on the launch stream, vs the 8 TB/s HBM peak) and `cpu_baseline` (the oracle's plain-C restatement
of the reference step incl. the reference's explicit inverse, timed on the host cores on a bounded sample; rank 0, N=1).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP32_MFMA_PEAK_TFLOPS = 157.3  # dense fp32 matrix peak (MI355X_MICROARCH.md); v_mfma_f32_16x16x4_f32 = 32 cycles
FP64_PEAK_TFLOPS = 78.6  # datasheet fp64 matrix = vector rate; tools/microbench.hip measures 64 cycles per v_mfma_f64_16x16x4_f64


def measured_traffic_bytes(kernel_name: str, config: str, batch: int):
    """HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/, collected with tools/profile_pmc.sh):
    FETCH_SIZE [KB] x 1024 x 2 (gfx950 reports half of wide 16 B/lane streaming reads, MI355X_MICROARCH.md HBM section)
    + WRITE_SIZE [KB] x 1024.  Only valid for the workload the counters were collected on; otherwise None."""
    if config != "cfg3" or batch != 65536 or not kernel_name.startswith("fused"):
        return None
    path = os.path.join(ROOT, "profiles", "r01_fused_cfg3_pmc_summary.json")
    try:
        with open(path) as f:
            summ = json.load(f)
        row = next(v for k, v in summ.items() if "kkt_fused" in k)
        return float(row["FETCH_SIZE"]) * 1024.0 * 2.0 + float(row["WRITE_SIZE"]) * 1024.0
    except Exception:
        return None


def usable_cores() -> int:
    """Cores this process may actually use: affinity mask and cgroup CPU quota (the GPU box gives a CPU share)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()
            if quota != "max":
                n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // p))
        except Exception:
            pass
    return n


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", default="cfg3", choices=["cfg2", "cfg3", "cfg4"])
    ap.add_argument("--batch", type=int, default=0, help="per-GPU batch (default: the config's)")
    ap.add_argument("--force-generic", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--parity-sample", type=int, default=256)
    args = ap.parse_args()

    import numpy as np
    import torch

    from mini_opt_amd import qp as Q
    from mini_opt_amd import sharding, synth

    info = sharding.RankInfo.from_env()
    if info.world_size != args.gpus and info.world_size > 1:
        raise SystemExit(f"WORLD_SIZE={info.world_size} but --gpus {args.gpus}")
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback)"
    # one process per GPU; on a box with fewer GPUs than ranks (rehearsals only) ranks wrap around the visible devices
    dev_index = info.local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    # RCCL ("nccl" on ROCm) carries only the timing barrier and two scalar reductions; MO_BENCH_BACKEND=gloo rehearses the
    # multi-rank path on a single-GPU box
    backend = os.environ.get("MO_BENCH_BACKEND", "nccl")
    dist = sharding.init_process_group(info, backend)

    cfg = synth.CONFIGS[args.config]
    n, k, m, m_r = cfg["n"], cfg["k"], cfg["m"], cfg["m_r"]
    dtype = torch.float64 if cfg["dtype"] == "f64" else torch.float32
    T = 8 if cfg["dtype"] == "f64" else 4
    batch = args.batch or cfg["batch"]

    prob, vars_, mu = synth.make_batch_torch(n, k, m, m_r, batch, dev, dtype, seed=synth.SEED + 1000 * info.rank)
    solver = Q.QPInteriorPointSolver(prob, force_generic=args.force_generic)
    solver.SetVariables(vars_)
    kernel_name = solver.step_kernel()

    def step():
        return solver.NewtonStep(mu, 0.995)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if info.world_size > 1:
        dist.barrier()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for e0, e1 in evs:  # events sit on torch's current stream == the stream handed to the C ABI
        e0.record()
        delta, alpha, status = step()
        e1.record()
    torch.cuda.synchronize()
    if info.world_size > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    elapsed_max, total_units = sharding.barrier_max_sum(info, elapsed, batch * args.steps, dev if backend == "nccl" else None)
    kernel_ms = float(np.mean([e0.elapsed_time(e1) for e0, e1 in evs]))

    ok = int((status == 0).sum().item())
    out = None
    if info.rank == 0:
        alg_bytes = synth.algorithmic_bytes(n, k, m, m_r, T)
        achieved = alg_bytes * batch / (kernel_ms * 1e-3) / 1e9
        out = {
            "metric": "batched dense KKT Newton steps/sec, n=64 fp64" if args.config == "cfg3" else f"batched dense KKT Newton steps/sec ({args.config})",
            "value": total_units / elapsed_max,
            "unit": "steps/s",
            "n_gpus": info.world_size,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed_max / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": cfg["dtype"],
            "data": "synthetic",
            "config": {"workload": f"BASELINE configs[{ {'cfg2': 1, 'cfg3': 2, 'cfg4': 3}[args.config] }]: batch={batch} per GPU, n={n} / {k} eq / {m} box, m_r={m_r}, J-level input (row-major J), one mo_newton_step launch per step",
                       "kernel": kernel_name, "batch_per_gpu": batch, "n": n, "k": k, "m": m, "m_r": m_r,
                       "parallelism": f"batch-sharded x{info.world_size}, no collectives"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": measured_traffic_bytes(kernel_name, args.config, batch),
                         "algorithmic_bytes_per_step": alg_bytes, "algorithmic_bytes_per_launch": alg_bytes * batch,
                         "kernel_ms": kernel_ms,
                         "fp64_tflops": synth.algorithmic_flops(n, k, m, m_r) * batch / (kernel_ms * 1e-3) / 1e12,
                         "fp64_frac_of_peak": synth.algorithmic_flops(n, k, m, m_r) * batch / (kernel_ms * 1e-3) / 1e12 / FP64_PEAK_TFLOPS},
            "status_ok": ok, "status_total": batch,
        }
        if cfg["dtype"] == "f32":  # cfg 4 is bound by the fp32 matrix cores (AI 37.5 flop/B vs ridge 19.7), not by HBM
            tf = synth.algorithmic_flops(n, k, m, m_r) * batch / (kernel_ms * 1e-3) / 1e12
            out["roofline"] = {"bound": "mfma", "achieved": tf, "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                               "frac": tf / FP32_MFMA_PEAK_TFLOPS, "traffic": None,
                               "algorithmic_flops_per_step": synth.algorithmic_flops(n, k, m, m_r), "kernel_ms": kernel_ms,
                               "hbm_gbs_algorithmic": achieved}

        # The cpu_baseline leg (rank 0 at N = 1 only; --no-cpu-baseline skips it entirely, e.g. under rocprofv3): the ONLY place where
        # bench.py touches oracle/ -- once as the checker of a sample of the launch just timed, once as the timed CPU baseline.
        out["parity"] = None
        try:
            if info.world_size > 1 or args.no_cpu_baseline:
                raise StopIteration
            from oracle import oracle as orc
            ns = min(args.parity_sample, batch)
            sl = slice(0, ns)
            h = lambda t: t[sl].double().cpu().numpy()
            ref, ref_alpha, ref_status, _ = orc.batched_newton_step(
                n, k, m, J=h(prob.J), r=h(prob.r), lam=float(np.float32(prob.lam)) if T == 4 else prob.lam,
                A_eq=h(prob.A_eq), b_eq=h(prob.b_eq), cons_var=prob.cons_var[sl].cpu().numpy(), cons_a=h(prob.cons_a),
                cons_b=h(prob.cons_b), vars_=h(vars_), mu=h(mu))
            got = delta[sl].double().cpu().numpy()
            err = np.max(np.abs(got - ref), axis=1) / np.max(np.abs(ref), axis=1)
            out["parity"] = {"sample": ns, "max_rel_inf": float(err.max()), "tolerance": 1e-10 if T == 8 else 2e-3,
                             "passed": bool(err.max() < (1e-10 if T == 8 else 2e-3))}
            if True:
                cores = usable_cores()
                pilot = min(batch, 4 * cores)
                hs = lambda t, cnt: t[:cnt].double().cpu().numpy()

                def run_cpu(cnt):
                    a = dict(J=hs(prob.J, cnt), r=hs(prob.r, cnt), lam=prob.lam, A_eq=hs(prob.A_eq, cnt), b_eq=hs(prob.b_eq, cnt),
                             cons_var=prob.cons_var[:cnt].cpu().numpy(), cons_a=hs(prob.cons_a, cnt), cons_b=hs(prob.cons_b, cnt),
                             vars_=hs(vars_, cnt), mu=hs(mu, cnt), use_inverse=True, num_threads=cores)
                    t = time.perf_counter()
                    _, _, _, used = orc.batched_newton_step(n, k, m, **a)
                    return time.perf_counter() - t, used

                run_cpu(pilot)
                tp, used = run_cpu(pilot)
                cnt = int(min(batch, max(pilot, pilot * args.cpu_seconds / 3 / max(tp, 1e-6))))
                reps = []
                t_total = 0.0
                while len(reps) < 3 or (t_total < args.cpu_seconds and len(reps) < 50):
                    t, used = run_cpu(cnt)
                    reps.append(t)
                    t_total += t
                out["cpu_baseline"] = {
                    "value": cnt / float(np.median(reps)), "unit": "steps/s", "cores": used, "kind": "port",
                    "sample": f"first {cnt} problems of the same batch, {len(reps)} repeats (median), OpenMP over problems; "
                              "plain-C restatement of the reference step incl. its explicit inverse (qp.cc:310-311); Eigen itself is absent from the image"}
        except StopIteration:
            pass
        except Exception as exc:  # the oracle is only the checker; never let it hide the measurement
            out["parity"] = {"error": repr(exc)}
        print(json.dumps(out), flush=True)
    if info.world_size > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
