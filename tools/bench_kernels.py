#!/usr/bin/env python3
"""One kernel family per invocation, event-timed -- the program tools/profile.py puts behind `rocprofv3 ... --`.
  --mode step        mo_newton_step            (the bench.py workload, any config / batch)
  --mode solve       mo_qp_solve, COMPLEMENTARITY     --mode solve_pc   ... PREDICTOR_CORRECTOR
  --mode linearize   mo_linearize (G = J^T J + lambda I written out)
  --mode generic     mo_newton_step on the generic kernel
Prints one JSON line: ms per launch (events on the launch stream), units/s, and for the Solve modes the iteration statistics."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch

from mini_opt_amd import qp as Q
from mini_opt_amd import synth


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mode", required=True, choices=["step", "generic", "solve", "solve_pc", "linearize"])
    ap.add_argument("--config", default="cfg3")
    ap.add_argument("--batch", type=int, default=0)
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-records", action="store_true", help="Solve without the per-iteration records")
    ap.add_argument("--layout", default="packed", choices=["packed", "col", "rowld"],
                    help="layout of J: packed row-major (16-byte stream), column-major or row-major with ld = n + 2 (gather stream)")
    ap.add_argument("--no-tiny", action="store_true", help="keep n + k <= 15 on the 32-variable tile grid (MO_PLAN_NO_TINY)")
    ap.add_argument("--shape", default="", help="n,k,m,m_r in fp64 instead of a named config (e.g. 64,24,32,128: the two-y-tile kernels)")
    args = ap.parse_args()
    d = synth.CONFIGS[args.config]
    if args.shape:
        n_, k_, m_, mr_ = (int(v) for v in args.shape.split(","))
        d = dict(n=n_, k=k_, m=m_, m_r=mr_, dtype="f64", batch=65536)
    dev = torch.device("cuda:0")
    dt = torch.float64 if d["dtype"] == "f64" else torch.float32
    batch = args.batch or d["batch"]
    prob, vars_, mu = synth.make_batch_torch(d["n"], d["k"], d["m"], d["m_r"], batch, dev, dt)
    if args.layout == "col":
        prob.J = prob.J.transpose(1, 2).contiguous()
        prob.J_layout = "col"
    elif args.layout == "rowld":
        wide = torch.zeros(batch, d["m_r"], d["n"] + 2, dtype=dt, device=dev)
        wide[:, :, :d["n"]] = prob.J
        prob.J = wide
    extra = {}
    if args.mode in ("step", "generic"):
        s = Q.QPInteriorPointSolver(prob, force_generic=args.mode == "generic", no_tiny=args.no_tiny)
        s.SetVariables(vars_)
        kernel = s.step_kernel()
        fn = lambda: s.NewtonStep(mu, 0.995)
    elif args.mode == "linearize":
        kernel = "linearize"
        # plan creation / output allocation are inside Q.linearize: hold them outside the timed call
        import ctypes as C
        from mini_opt_amd import _lib as L
        lib = L.lib()
        n = d["n"]
        desc = L.PlanDesc(n, 0, 0, d["m_r"], Q._DT[dt], 0, 0, 0, batch)
        plan = C.c_void_p()
        L.check(lib.mo_plan_create(C.byref(desc), C.byref(plan)))
        G = torch.empty(batch, n, n, dtype=dt, device=dev); c = torch.empty(batch, n, dtype=dt, device=dev)
        f = torch.empty(batch, dtype=dt, device=dev)
        q = Q.BatchedQP(n=n, J=prob.J, r=prob.r, lam=prob.lam)
        ps = q.as_struct()
        fn = lambda: L.check(lib.mo_linearize(plan, C.byref(ps), batch, Q._ptr(G), n * n, n, Q._ptr(c), n, Q._ptr(f), Q._stream()))
    else:
        s = Q.QPInteriorPointSolver(prob, no_tiny=args.no_tiny)
        kernel = "solve:" + s.step_kernel()
        params = Q.Params(initial_mu=1.0, sigma=0.1, max_iterations=10,
                          barrier_strategy=Q.PREDICTOR_CORRECTOR if args.mode == "solve_pc" else Q.COMPLEMENTARITY,
                          termination_kkt_tol=1e-8 if dt == torch.float64 else 1e-3)
        res = {}

        def fn():
            res["o"] = s.Solve(params, record_iterations=not args.no_records)
    for _ in range(args.warmup):
        fn()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.reps)]
    for e0, e1 in evs:
        e0.record(); fn(); e1.record()
    torch.cuda.synchronize()
    ms = sorted(e0.elapsed_time(e1) for e0, e1 in evs)
    if args.mode.startswith("solve"):
        o = res["o"]
        extra = {"mean_iterations": float(o.num_iterations.double().mean()), "satisfied_frac": float((o.termination_state == 0).double().mean()),
                 "status_ok_frac": float((o.status == 0).double().mean())}
    T = 8 if d["dtype"] == "f64" else 4
    print(json.dumps({"mode": args.mode, "config": args.config, "layout": args.layout, "batch": batch, "kernel": kernel, "reps": args.reps,
                      "ms_mean": sum(ms) / len(ms), "ms_min": ms[0], "units_per_s": batch / (sum(ms) / len(ms) * 1e-3),
                      "algorithmic_bytes_per_unit_step": synth.algorithmic_bytes(d["n"], d["k"], d["m"], d["m_r"], T), **extra}), flush=True)


if __name__ == "__main__":
    main()
