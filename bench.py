#!/usr/bin/env python3
"""bench.py -- batched dense KKT Newton steps/s on MI355X (BASELINE.json metric).

A "step" = one pass of the hot path (mo_newton_step through the C ABI) over one batch of synthetic QPs that is already
resident in HBM.

`--mode solve` times the caller-facing entry point instead (what ConstrainedNonlinearLeastSquares::ComputeStepDirection
invokes, nonlinear.cc:221-247): a "step" is then ONE mo_qp_solve launch -- the whole interior-point Solve (qp.cc:100-151) of every QP of the
batch from the NAIVE initial guess, iteration records included -- the unit is solves/s, `roofline` is algorithmic flops of the iterations
the batch actually ran against the fp64 peak, `parity` compares termination state, iteration count and optimum of the problems with the
oracle's Solve, and `cpu_baseline` is the oracle's Solve on the host cores.  `--strategy pc` selects PREDICTOR_CORRECTOR.

Workloads (`--config`; the default follows `--gpus`):
  N = 1  -> cfg3 = BASELINE.json configs[2]: batch 65536, n=64 / 8 eq / 32 box, fp64, J-level input.
  N > 1  -> cfg5 = BASELINE.json configs[4]: 2^20 QPs of the cfg3 shape IN TOTAL, sharded contiguously over the N ranks with
            mini_opt_amd.sharding.shard_range (131072 per GPU at N = 8), no data-path collective ("scaling": "strong").
            `--config cfg5 --gpus 1 --batch 131072` launches one config-5 shard on a single GPU.
  cfg2 / cfg4 are the other single-GPU BASELINE configs (parity-test cases; runnable here for profiling).
value = problems all ranks processed / max-over-ranks wall time.

Prints ONE JSON line on rank 0 with `roofline` (dominant kernel: algorithmic bytes per launch / average launch duration
measured with events on the launch stream, vs the 8 TB/s HBM peak) and `cpu_baseline` (the oracle's plain-C restatement
of the reference step incl. the reference's explicit inverse, timed on the host cores on a bounded sample; rank 0, N=1;
with the 1-core and the direct-solve variants SURVEY.md 8(d) asks for).

At ANY N the line carries a correctness figure: after the timed region every rank checks a sample of ITS OWN shard against the oracle (the
whole shard at N = 1, a 256-problem strided sample per rank at N > 1; `--parity-sample`), and MAX of the errors / SUM of the status words
and disagreements travel over the control-plane group: `parity`, `status_ok` / `status_total` are all-rank figures.

`--dry` rehearses the rank plumbing without a GPU (tests/test_bench_ranks_cpu.py: gloo, world size 2): the launch is replaced
by a no-op and no device is touched; the JSON line then carries "dry": true and no measurement.
"""
import argparse
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP32_MFMA_PEAK_TFLOPS = 157.3  # dense fp32 matrix peak (MI355X_MICROARCH.md); v_mfma_f32_16x16x4_f32 = 32 cycles
FP64_PEAK_TFLOPS = 78.6  # datasheet fp64 matrix = vector rate; tools/microbench.hip measures 64 cycles per v_mfma_f64_16x16x4_f64
CFG5_TOTAL = 1 << 20   # BASELINE.json configs[4]
BASELINE_INDEX = {"cfg2": 1, "cfg3": 2, "cfg4": 3, "cfg5": 4}


def kernel_source_digest() -> str:
    """sha256 over the kernel sources: a committed PMC summary is only quoted while the kernels it measured are unchanged."""
    h = hashlib.sha256()
    for name in ("kkt_fused.hip", "kkt_fused_f32.hip", "kkt_generic.hip", "mo_kernels.h"):  # (the step kernel of every BASELINE config lives in these)
        with open(os.path.join(ROOT, "mini_opt_amd", "csrc", name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def measured_traffic(kernel_name: str, config: str, batch: int, mode: str = "step"):
    """HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/, collected with tools/profile_pmc.sh):
    FETCH_SIZE [KB] x 1024 x 2 (gfx950 reports half of wide 16 B/lane streaming reads, MI355X_MICROARCH.md HBM section)
    + WRITE_SIZE [KB] x 1024.  The counters cannot be read from inside this process, so the figure comes from the committed
    summary -- and only while it is valid: same workload, and the kernel sources still hash to what the summary recorded.
    Returns (bytes or None, provenance dict)."""
    src = {"source": None}
    if not kernel_name.startswith("fused"):
        return None, src
    want = {("step", "cfg3"): "r04_step_cfg3_pmc_summary.json", ("step", "cfg5"): "r04_step_cfg5shard_pmc_summary.json",
            ("step", "cfg2"): "r04_step_cfg2_pmc_summary.json", ("solve", "cfg3"): "r04_solve_cfg3_pmc_summary.json",
            ("solve_pc", "cfg3"): "r04_solve_pc_cfg3_pmc_summary.json", ("solve", "cfg2"): "r04_solve_cfg2_pmc_summary.json"}.get((mode, config))
    if want is None:
        return None, src
    path = os.path.join(ROOT, "profiles", want)
    try:
        with open(path) as f:
            summ = json.load(f)
        meta = summ.get("_meta", {})
        src = {"source": "profiles/" + want, "collected_at_kernel_digest": meta.get("kernel_digest"),
               "collected_at_commit": meta.get("git_head"), "batch": meta.get("batch")}
        if meta.get("kernel_digest") != kernel_source_digest() or int(meta.get("batch", -1)) != batch:
            src["stale"] = True
            return None, src
        row = next(v for k, v in summ.items() if "kkt_fused" in k)
        try:
            # SIMD time per step from the same summary (the fp64 MFMA and the VALU are one datapath, their cycles add up):
            # SQ_VALU_MFMA_BUSY_CYCLES (cycles) + SQ_ACTIVE_INST_VALU (quad-cycles) against duration x clock x 1 024 SIMDs
            simd_cycles = float(row["counter_pass_kernel_ns"]) * float(row["effective_clock_ghz"]) * 1024.0 / batch
            mfma = float(row["SQ_VALU_MFMA_BUSY_CYCLES"]) / batch
            valu = (float(row["SQ_ACTIVE_INST_VALU"]) - float(row["SQ_INSTS_MFMA"])) * 4.0 / batch   # (the counter also holds 4 issue cycles per MFMA)
            src["datapath"] = {"simd_cycles_per_step": simd_cycles, "mfma_busy_cycles_per_step": mfma, "other_valu_active_cycles_per_step": valu,
                               "busy_frac": (mfma + valu) / simd_cycles, "clock_ghz": float(row["effective_clock_ghz"])}
        except Exception:
            pass
        return float(row["FETCH_SIZE"]) * 1024.0 * 2.0 + float(row["WRITE_SIZE"]) * 1024.0, src
    except Exception as exc:  # no summary for this workload (yet): traffic stays null
        src["error"] = repr(exc)
        return None, src


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


def plan_workload(args, info):
    """What this rank runs: (config name, shape dict, per-rank batch, total batch, scaling label, workload text)."""
    from mini_opt_amd import sharding, synth
    config = args.config or ("cfg3" if info.world_size == 1 else "cfg5")
    if config == "cfg5":
        shape = dict(synth.CONFIGS["cfg3"])
        if args.batch:   # one explicit shard per rank (e.g. --gpus 1 --batch 131072 = the 8-GPU shard on one GPU)
            batch, total = args.batch, args.batch * info.world_size
        else:
            b0, b1 = sharding.shard_range(CFG5_TOTAL, info.rank, info.world_size)
            batch, total = b1 - b0, CFG5_TOTAL
        scaling = "strong"
        text = (f"BASELINE configs[4]: {total} QPs in total, n={shape['n']} / {shape['k']} eq / {shape['m']} box, m_r={shape['m_r']}, fp64, "
                f"sharded contiguously over {info.world_size} GPU(s) ({batch} on rank {info.rank}), no collectives, J-level input "
                f"(row-major J), one mo_newton_step launch per step and rank")
    else:
        shape = dict(synth.CONFIGS[config])
        batch = args.batch or shape["batch"]
        total = batch * info.world_size
        scaling = "weak"
        text = (f"BASELINE configs[{BASELINE_INDEX[config]}]: batch={batch} per GPU, n={shape['n']} / {shape['k']} eq / {shape['m']} box, "
                f"m_r={shape['m_r']}, J-level input (row-major J), one mo_newton_step launch per step")
    return config, shape, batch, total, scaling, text


def launch_ranks(n_ranks: int, argv) -> int:
    """`python bench.py --gpus N` without a launcher: start N ranks (one process per GPU) as CHILDREN through
    `python -m torch.distributed.run --standalone --local-addr 127.0.0.1` (the launcher picks the rendezvous port itself: no
    bind-close-reuse race on a port number) and return the launcher's exit code.  Nothing in this process initialises HIP (no torch
    import, no device query): the ranks are fresh processes."""
    import subprocess
    env = dict(os.environ)
    for key in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(key, None)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC only on this pool (RCCL needs it)
    env.setdefault("OMP_NUM_THREADS", "1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1",
           f"--nproc-per-node={n_ranks}", os.path.abspath(__file__)] + list(argv)
    res = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)   # the ranks' stderr goes straight through
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    for ln in res.stdout.splitlines():
        if not ln.startswith("{"):
            print(ln, file=sys.stderr)
    if res.returncode == 0 and len(lines) != 1:
        print(f"bench.py: expected one JSON line from rank 0, got {len(lines)}", file=sys.stderr)
        return 1
    for ln in lines:
        print(ln, flush=True)
    return res.returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", default=None, choices=["cfg2", "cfg3", "cfg4", "cfg5"],
                    help="default: cfg3 at --gpus 1, cfg5 (2^20 QPs in total, sharded) at --gpus > 1")
    ap.add_argument("--batch", type=int, default=0, help="per-GPU batch (default: the config's)")
    ap.add_argument("--mode", default="step", choices=["step", "solve"],
                    help="step: one mo_newton_step launch per step (the BASELINE metric); solve: one mo_qp_solve launch per step (solves/s)")
    ap.add_argument("--strategy", default="complementarity", choices=["complementarity", "pc"], help="--mode solve: barrier strategy")
    ap.add_argument("--no-records", action="store_true", help="--mode solve: without the per-iteration records")
    ap.add_argument("--force-generic", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--parity-sample", type=int, default=0,
                    help="problems of each rank's shard checked against the oracle, strided over the shard (0 = the whole shard at N = 1, 256 per rank at N > 1)")
    ap.add_argument("--sustain-seconds", type=float, default=2.0,
                    help="after the K timed steps: back-to-back launches for at least this long, reported as `sustained` (0 = skip)")
    ap.add_argument("--dry", action="store_true", help="rank plumbing only: no GPU, the launch is a no-op (CPU tests)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # a bare `python bench.py --gpus N`: become the launcher.  This parent never imports torch or touches a device; it starts one
        # rank per GPU as child processes through torch.distributed.run and relays rank 0's JSON line and the exit code.
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))

    from mini_opt_amd import sharding

    info = sharding.RankInfo.from_env()
    if info.world_size != args.gpus:
        # a launcher set WORLD_SIZE and it disagrees with --gpus: refuse rather than measure something else than was asked for
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={info.world_size}; launch with "
                         f"python -m torch.distributed.run --nnodes=1 --nproc-per-node {args.gpus} --master-addr 127.0.0.1 bench.py --gpus {args.gpus} ... "
                         f"(or run `python bench.py --gpus {args.gpus}` without a launcher: it starts its own ranks)")
    config, cfg, batch, total_batch, scaling, workload = plan_workload(args, info)
    n, k, m, m_r = cfg["n"], cfg["k"], cfg["m"], cfg["m_r"]
    T = 8 if cfg["dtype"] == "f64" else 4
    solve_mode = args.mode == "solve"
    # mini_opt's ComputeStepDirection parameters (nonlinear.cc:226-231: initial_mu 1, sigma 0.1) with the iteration cap of qp.hpp:149
    solve_kw = dict(initial_mu=1.0, sigma=0.1, max_iterations=10, termination_kkt_tol=1e-8 if T == 8 else 1e-3,
                    barrier_strategy=2 if args.strategy == "pc" else 0)
    if solve_mode:
        workload = workload.replace("one mo_newton_step launch per step", "one mo_qp_solve launch per step (whole interior-point Solve from the NAIVE guess, "
                                    + ("PREDICTOR_CORRECTOR" if args.strategy == "pc" else "COMPLEMENTARITY") + ", kkt tol %g, <= 10 iterations)" % solve_kw["termination_kkt_tol"])

    if args.dry:
        # RCCL is replaced by gloo and the launch by a no-op; everything else (sharding, barriers, MAX / SUM, the JSON line) is the
        # code the GPU run executes
        dist = sharding.init_process_group(info, os.environ.get("MO_BENCH_BACKEND", "gloo"))
        dev = None
        tgroup, tlabel, tdev = None, os.environ.get("MO_BENCH_BACKEND", "gloo") if info.world_size > 1 else "none", None
        kernel_name = "dry"

        def step():
            return None
        sync = lambda: None
    else:
        import ctypes as C

        import torch

        from mini_opt_amd import _lib as L
        from mini_opt_amd import qp as Q
        from mini_opt_amd import synth
        assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback)"
        # one process per GPU; on a box with fewer GPUs than ranks (rehearsals only) ranks wrap around the visible devices
        dev_index = info.local_rank % torch.cuda.device_count()
        torch.cuda.set_device(dev_index)
        dev = torch.device("cuda", dev_index)
        # RCCL ("nccl" on ROCm) carries only the timing barrier and two scalar reductions (the path has no data-path collective).  The
        # default group is gloo -- a control plane that always works -- and the RCCL group on top of it is probed once: if any rank cannot
        # form it, every rank times over gloo instead and the JSON line says so (`timing_sync`).  MO_BENCH_BACKEND=gloo skips RCCL
        # altogether (rehearsals of the multi-rank path on a single-GPU box).
        dist = sharding.init_process_group(info, "gloo")
        tgroup, tlabel, tdev = sharding.timing_group(info, os.environ.get("MO_BENCH_BACKEND", "nccl"), dev)
        dtype = torch.float64 if cfg["dtype"] == "f64" else torch.float32
        prob, vars_, mu = synth.make_batch_torch(n, k, m, m_r, batch, dev, dtype, seed=synth.SEED + 1000 * info.rank)
        solver = Q.QPInteriorPointSolver(prob, force_generic=args.force_generic)
        solver.SetVariables(vars_.clone())
        if solve_mode:
            # The launch itself, on preallocated outputs (the Python mirror's Solve allocates its result tensors per call): every call starts
            # from the NAIVE guess (qp.cc:439-482 resets x, s, y, z), so repeated launches do identical work.
            kernel_name = solver.solve_kernel()
            sp = Q.Params(**solve_kw).as_struct()
            V = n + 2 * m + k
            term = torch.zeros(batch, dtype=torch.int32, device=dev)
            nit = torch.zeros(batch, dtype=torch.int32, device=dev)
            its = None if args.no_records else torch.zeros(batch, solve_kw["max_iterations"], L.MO_ITER_RECORD, dtype=dtype, device=dev)
            lag = torch.zeros(batch, 2, dtype=dtype, device=dev)
            svars = solver.variables()
            status = torch.zeros(batch, dtype=torch.int32, device=dev)
            lib, plan, pstruct = L.lib(), solver._plan, solver._prob

            def step():
                L.check(lib.mo_qp_solve(plan, C.byref(pstruct), batch, C.byref(sp), Q._ptr(svars), V, Q._ptr(term), Q._ptr(nit), Q._ptr(its),
                                        Q._ptr(lag), Q._ptr(status), Q._stream()))
                return None
        else:
            kernel_name = solver.step_kernel()
            res = {}

            def step():
                res["out"] = solver.NewtonStep(mu, 0.995)
                return None
        sync = torch.cuda.synchronize

    def barrier():
        if info.world_size > 1:
            if tgroup is not None:
                dist.barrier(group=tgroup, device_ids=[dev.index])
            else:
                dist.barrier()

    for _ in range(args.warmup):
        step()
    sync()
    barrier()
    sync()
    if not args.dry:
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    if args.dry:
        for _ in range(args.steps):
            step()
    else:
        for e0, e1 in evs:  # events sit on torch's current stream == the stream handed to the C ABI
            e0.record()
            step()
            e1.record()
    sync()
    barrier()
    sync()
    elapsed = time.perf_counter() - t0
    elapsed_max, total_units = sharding.barrier_max_sum(info, elapsed, batch * args.steps, tdev, tgroup)

    unit = "solves/s" if solve_mode else "steps/s"
    out = None
    if info.rank == 0:
        if solve_mode:
            metric = f"batched interior-point QP solves/sec (mo_qp_solve, {args.strategy}), n={n} {cfg['dtype']}"
        else:
            metric = "batched dense KKT Newton steps/sec, n=64 fp64" if config in ("cfg3", "cfg5") else f"batched dense KKT Newton steps/sec ({config})"
        out = {
            "metric": metric,
            "value": total_units / elapsed_max,
            "unit": unit,
            "n_gpus": info.world_size,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed_max / args.steps,
            "higher_is_better": True,
            "scaling": scaling,
            "vs_baseline": None,
            "dtype": cfg["dtype"],
            "data": "synthetic",
            "config": {"workload": workload, "name": config, "mode": args.mode, "kernel": kernel_name, "batch_per_gpu": batch, "batch_total": total_batch,
                       "n": n, "k": k, "m": m, "m_r": m_r, "parallelism": f"batch-sharded x{info.world_size}, no collectives"},
            "timing_sync": tlabel,   # what carried the barrier + MAX / SUM between the ranks ("none" at N = 1)
        }

    # ---- correctness figure of the timed launch, on every rank (the checker leg: the only place besides cpu_baseline where bench.py touches
    # oracle/).  Sample: the whole shard at N = 1, a strided sample per rank at N > 1; reduced over the control-plane (gloo) group.
    import numpy as np
    ns = args.parity_sample if args.parity_sample > 0 else (batch if info.world_size == 1 else min(batch, 256))
    ns = max(1, min(ns, batch))
    idx = np.unique(np.linspace(0, batch - 1, ns).astype(np.int64))
    tol = 1e-10 if T == 8 else 2e-3
    local = {"max_err": 0.0, "p999": 0.0, "disagree": 0, "ok": batch, "checked": int(len(idx)), "error": None, "extra": {}}
    if not args.dry and not args.no_cpu_baseline:
        try:
            from oracle import oracle as orc
            tix = torch.as_tensor(idx, device=dev)
            h = lambda t: t.index_select(0, tix).double().cpu().numpy()
            pr = dict(J=h(prob.J), r=h(prob.r), lam=float(np.float32(prob.lam)) if T == 4 else prob.lam, A_eq=h(prob.A_eq), b_eq=h(prob.b_eq),
                      cons_var=prob.cons_var.index_select(0, tix).cpu().numpy(), cons_a=h(prob.cons_a), cons_b=h(prob.cons_b))
            if solve_mode:
                rterm, rnit, rvars, _ = orc.batched_solve(n, k, m, **pr, **solve_kw)
                gterm, gnit, gvars = term.index_select(0, tix).cpu().numpy(), nit.index_select(0, tix).cpu().numpy(), h(svars)
                xerr = np.max(np.abs(gvars[:, :n] - rvars[:, :n]), axis=1) / np.maximum(1.0, np.max(np.abs(rvars[:, :n]), axis=1))
                same = (gterm == rterm) & (gnit == rnit)
                # fp64: termination state and iteration count of every checked problem as the oracle's; the optimum within 1e-6 (the reference's own
                # bound on its Solve KATs, qp_test.cc:252-471).  fp32 (no reference counterpart): iteration count within 2, optimum within 2e-2.
                if T == 4:
                    same = (gterm == rterm) & (np.abs(gnit - rnit) <= 2)
                xtol = 1e-6 if T == 8 else 2e-2
                local.update(max_err=float(xerr.max()), p999=float(np.quantile(xerr, 0.999)), disagree=int((~same).sum() + (xerr[same] >= xtol).sum()))
                local["extra"] = {"oracle_mean_iterations": float(rnit.mean()), "oracle_satisfied_frac": float((rterm == 0).mean())}
                tol = xtol
            else:
                delta, alpha, st_t = res["out"]
                ref, ref_alpha, ref_status, _ = orc.batched_newton_step(n, k, m, **pr, vars_=h(vars_), mu=h(mu))
                got = delta.index_select(0, tix).double().cpu().numpy()
                err = np.max(np.abs(got - ref), axis=1) / np.max(np.abs(ref), axis=1)
                st_s = st_t.index_select(0, tix).cpu().numpy()
                local.update(max_err=float(err.max()), p999=float(np.quantile(err, 0.999)), disagree=int((ref_status != st_s).sum()))
        except Exception as exc:  # the oracle is only the checker; never let it hide the measurement
            local["error"] = repr(exc)[:300]
    if not args.dry:
        st_all = status if solve_mode else res["out"][2]
        local["ok"] = int((st_all == 0).sum().item())
    red = sharding.reduce_scalars(info, maxima=[local["max_err"], local["p999"], 1.0 if local["error"] else 0.0],
                                  sums=[local["disagree"], local["ok"], batch, local["checked"]])
    parity = None
    if not args.no_cpu_baseline or args.dry:
        parity = {"sample": int(red["sums"][3]), "sample_per_rank": int(len(idx)), "ranks": info.world_size, "max_rel_inf": red["maxima"][0],
                  "p999_rel_inf": red["maxima"][1], "tolerance": tol,
                  ("solve_disagreements" if solve_mode else "status_disagreements"): int(red["sums"][0]),
                  "passed": bool(red["maxima"][0] < tol and red["sums"][0] == 0 and red["maxima"][2] == 0.0), **local["extra"]}
        if solve_mode:
            parity["what"] = ("termination state and iteration count equal to the oracle's Solve on every checked problem, x within "
                              f"{tol:g} rel-inf (max_rel_inf is the x error)")
        if red["maxima"][2] != 0.0:
            parity["error"] = local["error"] or "the checker failed on another rank"
        if args.dry:
            parity["dry"] = True
            parity["passed"] = None   # nothing was launched, nothing was checked: only the reduction ran
    if info.rank == 0:
        out["parity"] = parity
        out["status_ok"], out["status_total"] = int(red["sums"][1]), int(red["sums"][2])

    if args.dry:
        if info.rank == 0:
            out["dry"] = True
            out["value"] = None  # nothing was measured
            out["units_per_step_all_ranks"] = total_units // args.steps
            print(json.dumps(out), flush=True)
        if info.world_size > 1:
            dist.barrier()
            dist.destroy_process_group()
        return

    kernel_ms = float(np.mean([e0.elapsed_time(e1) for e0, e1 in evs]))
    # The K timed steps above are the contract's figure; K x ~1.4 ms is a short region, so the same launch is also run back to back
    # for >= --sustain-seconds (clock ramp, thermal state and the ticket counter's behaviour over thousands of launches included).
    sustained = None
    if args.sustain_seconds > 0:
        chunk = max(1, int(0.25 / max(kernel_ms * 1e-3, 1e-6)))
        launches, ts = 0, time.perf_counter()
        while True:
            for _ in range(chunk):
                step()
            sync()
            launches += chunk
            dt = time.perf_counter() - ts
            if dt >= args.sustain_seconds:
                break
        sustained = {"value": batch * launches / dt, "unit": unit, "seconds": dt, "launches": launches,
                     "ms_per_step": 1e3 * dt / launches, "scope": "rank 0" if info.world_size > 1 else "the GPU"}
    if info.rank == 0:
        alg_bytes = synth.algorithmic_bytes(n, k, m, m_r, T)
        flops = synth.algorithmic_flops(n, k, m, m_r)
        out["sustained"] = sustained
        if solve_mode:
            # Algorithmic flops of the work the batch actually did (SURVEY.md 8(d) terms): J^T J + J^T r once per problem; per iteration the
            # KKT residual, the LDL^T and one solve (two for the predictor-corrector); one more residual for the final termination test.
            P = n + k
            f_lin = m_r * n * (n + 1) + 2 * m_r * n
            f_res = 2 * n * n + 4 * k * n + 6 * m
            f_fac = P ** 3 / 3.0
            f_sol = 2 * P * P + 8 * m
            its_total = float(nit.double().sum().item())
            total_flops = batch * (f_lin + f_res) + its_total * (f_res + f_fac + f_sol * (2 if args.strategy == "pc" else 1))
            tf = total_flops / (kernel_ms * 1e-3) / 1e12
            peak = FP64_PEAK_TFLOPS if T == 8 else FP32_MFMA_PEAK_TFLOPS
            traffic, traffic_src = measured_traffic(kernel_name, config, batch, mode="solve_pc" if args.strategy == "pc" else "solve")
            # algorithmic HBM bytes of a Solve: the problem read once (J, r, A_eq, b_eq, constraints), the state / records / outputs written once
            rec_bytes = 0 if args.no_records else T * 14 * solve_kw["max_iterations"]
            alg_solve = T * (m_r * n + m_r + k * n + k) + m * (4 + 2 * T) + T * (n + 2 * m + k) + rec_bytes + 2 * T + 12
            out["roofline"] = {"bound": "mfma", "achieved": tf, "peak": peak, "unit": "TFLOP/s", "frac": tf / peak, "traffic": traffic,
                               "traffic_provenance": traffic_src, "algorithmic_flops_per_launch": total_flops,
                               "algorithmic_flops_per_solve_mean": total_flops / batch, "mean_iterations": its_total / batch,
                               "satisfied_kkt_tol_frac": float((term == 0).double().mean().item()),
                               "algorithmic_bytes_per_solve": alg_solve, "algorithmic_bytes_per_launch": alg_solve * batch,
                               "hbm_gbs_algorithmic": alg_solve * batch / (kernel_ms * 1e-3) / 1e9,
                               "kernel_ms": kernel_ms, "scope": "rank 0's launches" if info.world_size > 1 else "the launch"}
        else:
            achieved = alg_bytes * batch / (kernel_ms * 1e-3) / 1e9
            traffic, traffic_src = measured_traffic(kernel_name, config, batch)
            out["roofline"] = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_provenance": traffic_src,
                               "algorithmic_bytes_per_step": alg_bytes, "algorithmic_bytes_per_launch": alg_bytes * batch,
                               "datapath": traffic_src.pop("datapath", None),
                               "kernel_ms": kernel_ms, "scope": "rank 0's launches" if info.world_size > 1 else "the launch",
                               "fp64_tflops": flops * batch / (kernel_ms * 1e-3) / 1e12,
                               "fp64_frac_of_peak": flops * batch / (kernel_ms * 1e-3) / 1e12 / FP64_PEAK_TFLOPS}
            if cfg["dtype"] == "f32":  # cfg 4 is bound by the fp32 matrix cores (AI 37.5 flop/B vs ridge 19.7), not by HBM
                tf = flops * batch / (kernel_ms * 1e-3) / 1e12
                out["roofline"] = {"bound": "mfma", "achieved": tf, "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                                   "frac": tf / FP32_MFMA_PEAK_TFLOPS, "traffic": None,
                                   "algorithmic_flops_per_step": flops, "kernel_ms": kernel_ms,
                                   "hbm_gbs_algorithmic": achieved}

        # The cpu_baseline leg (rank 0 at N = 1 only; --no-cpu-baseline skips it entirely, e.g. under rocprofv3): the oracle timed on the host cores
        try:
            if info.world_size > 1 or args.no_cpu_baseline:
                raise StopIteration
            from oracle import oracle as orc
            cores = usable_cores()
            hs = lambda t, cnt: t[:cnt].double().cpu().numpy()

            def run_cpu(cnt, threads, use_inverse):
                a = dict(J=hs(prob.J, cnt), r=hs(prob.r, cnt), lam=prob.lam, A_eq=hs(prob.A_eq, cnt), b_eq=hs(prob.b_eq, cnt),
                         cons_var=prob.cons_var[:cnt].cpu().numpy(), cons_a=hs(prob.cons_a, cnt), cons_b=hs(prob.cons_b, cnt), num_threads=threads)
                t = time.perf_counter()
                if solve_mode:
                    used = orc.batched_solve(n, k, m, **a, **solve_kw)[3]
                else:
                    used = orc.batched_newton_step(n, k, m, **a, vars_=hs(vars_, cnt), mu=hs(mu, cnt), use_inverse=use_inverse)[3]
                return time.perf_counter() - t, used

            def timed_variant(threads, use_inverse, seconds):
                pilot = min(batch, 4 * max(threads, 1))
                run_cpu(pilot, threads, use_inverse)
                tp, used = run_cpu(pilot, threads, use_inverse)
                cnt = int(min(batch, max(pilot, pilot * seconds / 3 / max(tp, 1e-6))))
                reps, t_total = [], 0.0
                while len(reps) < 3 or (t_total < seconds and len(reps) < 50):
                    t, used = run_cpu(cnt, threads, use_inverse)
                    reps.append(t)
                    t_total += t
                return {"value": cnt / float(np.median(reps)), "unit": unit, "cores": used,
                        "sample": f"first {cnt} problems of the same batch, {len(reps)} repeats (median)"}

            if solve_mode:
                main_v = timed_variant(cores, True, args.cpu_seconds * 0.7)
                out["cpu_baseline"] = {
                    **main_v, "kind": "port",
                    "sample": main_v["sample"] + ", OpenMP over problems; plain-C restatement of QPInteriorPointSolver::Solve (qp.cc:100-151) incl. the "
                              "reference's explicit inverse per iteration; Eigen itself is absent from the image",
                    "one_core": timed_variant(1, True, args.cpu_seconds * 0.3)}
            else:
                # all cores and one core, with the reference's explicit inverse (qp.cc:310-311) and with a direct solve
                main_v = timed_variant(cores, True, args.cpu_seconds * 0.5)
                out["cpu_baseline"] = {
                    **main_v, "kind": "port",
                    "sample": main_v["sample"] + ", OpenMP over problems; plain-C restatement of the reference step incl. its explicit "
                              "inverse (qp.cc:310-311); Eigen itself is absent from the image",
                    "one_core": timed_variant(1, True, args.cpu_seconds * 0.15),
                    "direct": timed_variant(cores, False, args.cpu_seconds * 0.2),
                    "direct_one_core": timed_variant(1, False, args.cpu_seconds * 0.15)}
        except StopIteration:
            pass
        except Exception as exc:  # the oracle is only the checker; never let it hide the measurement
            out["cpu_baseline"] = {"error": repr(exc)}
        print(json.dumps(out), flush=True)
    if info.world_size > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
