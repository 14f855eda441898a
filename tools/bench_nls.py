#!/usr/bin/env python3
"""Secondary measurement: the batched SQP loop (mo_nls_solve) on a many-start Himmelblau sweep (nonlinear_test.cc:597-664
scaled up).  usage: python tools/bench_nls.py [starts]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch

from mini_opt_amd import nls as NLS
from tests import nls_problems as P


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
    rng = np.random.default_rng(0)
    guesses = rng.uniform(-4.5, 4.5, (B, 2))
    cons = P.box(-5.0, 5.0)
    kw = dict(max_iterations=20, max_qp_iterations=10, relative_exit_tol=1e-12, absolute_first_derivative_tol=1e-8,
              termination_kkt_tolerance=1e-6)
    g = torch.as_tensor(guesses, device="cuda:0")
    timings = {}
    for name, cost in (("torch_residuals", P.himmelblau_torch), ("device_family", NLS.DeviceFamily(NLS.HIMMELBLAU, 2))):
        nls = NLS.ConstrainedNonlinearLeastSquares(NLS.Problem(2, cost, cost_rows=2, inequality_constraints=cons), batch=B)
        nls.Solve(NLS.Params(**kw), g); torch.cuda.synchronize()
        t = time.perf_counter()
        out = nls.Solve(NLS.Params(**kw), g); torch.cuda.synchronize()
        timings[name] = time.perf_counter() - t
    dt = timings["device_family"]
    x = nls.variables().cpu().numpy()
    sols = np.array(P.HIMMELBLAU_SOLUTIONS)
    dist = np.min(np.linalg.norm(x[:, None, :] - sols[None], axis=2), axis=1)
    res = {"problem": "Himmelblau, box [-5, 5]^2", "starts": B, "seconds": dt, "problems_per_s": B / dt,
           "seconds_with_torch_residuals": timings["torch_residuals"],
           "satisfied_frac": float(NLS.TerminationStateIndicatesSatisfiedTol(out.termination_state).double().mean()),
           "at_an_optimum_frac": float((dist < 5e-5).mean()), "mean_outer_iterations": float(out.num_iterations.double().mean()),
           "mean_qp_iterations": float(out.NumQPIterations().double().mean())}
    print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
