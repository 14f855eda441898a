#!/usr/bin/env python3
"""Secondary measurements (not the bench.py contract line): generic-kernel step rate per config and the on-device
interior-point Solve (row f1) rate.  usage: python tools/bench_modes.py [cfg ...]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch

from mini_opt_amd import qp as Q
from mini_opt_amd import synth


def timeit(fn, reps):
    fn(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / reps


def main():
    dev = torch.device("cuda:0")
    for cfg in (sys.argv[1:] or ["cfg2", "cfg3", "cfg4"]):
        d = synth.CONFIGS[cfg]
        dt = torch.float64 if d["dtype"] == "f64" else torch.float32
        batch = int(os.environ.get("MO_BENCH_BATCH", "8192"))
        prob, vars_, mu = synth.make_batch_torch(d["n"], d["k"], d["m"], d["m_r"], batch, dev, dt)
        out = {"cfg": cfg, "batch": batch}
        for force in (True, False):
            s = Q.QPInteriorPointSolver(prob, force_generic=force)
            s.SetVariables(vars_)
            t = timeit(lambda: s.NewtonStep(mu, 0.995), 5)
            out[("generic" if force else s.step_kernel()) + "_step_per_s"] = batch / t
        for sname, strat in (("", Q.COMPLEMENTARITY), ("_pc", Q.PREDICTOR_CORRECTOR)):
            s = Q.QPInteriorPointSolver(prob)
            params = Q.Params(initial_mu=1.0, sigma=0.1, max_iterations=10, barrier_strategy=strat,
                              termination_kkt_tol=1e-8 if dt == torch.float64 else 1e-3)
            res = {}
            def run():
                res["o"] = s.Solve(params)
            t = timeit(run, 2)
            o = res["o"]
            out["solve%s_per_s" % sname] = batch / t
            t2 = timeit(lambda: s.Solve(params, record_iterations=False), 2)
            out["solve%s_no_records_per_s" % sname] = batch / t2
            out["solve%s_mean_iterations" % sname] = float(o.num_iterations.double().mean())
            out["solve%s_satisfied_frac" % sname] = float((o.termination_state == 0).double().mean())
            out["solve%s_status_ok_frac" % sname] = float((o.status == 0).double().mean())
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
