#!/usr/bin/env python3
"""Runs the failing shapes of the 'alpha_dual = 1' finding through one diagnostic build of the library (MO_LIB_PATH) and prints what the
generic kernel's Iterate and its Solve (one iteration) saw: alpha_primal / alpha_dual and the probes z[0], dz[0], the dual step length as
it sat in LDS right after compute_alpha.  One process per build (the library is loaded once): tools/alpha_dual_probe/run_all.sh."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from mini_opt_amd import qp as Q  # noqa: E402


def T(a, dt=torch.float64):
    return torch.as_tensor(np.ascontiguousarray(a), dtype=dt, device="cuda:0")


def main():
    tag = sys.argv[1]
    rows = []
    for (n, k, m) in [(58, 21, 21), (60, 21, 21), (100, 8, 30)]:
        rng = np.random.default_rng(n * 100 + k)
        B, m_r = 3, n + 8
        J = rng.uniform(-1, 1, (B, m_r, n)); r = rng.uniform(-1, 1, (B, m_r))
        A = rng.uniform(-1, 1, (B, n, k)); b = rng.uniform(-1, 1, (B, k))
        cv = rng.integers(0, n, (B, m)).astype(np.int32); ca = rng.choice([-1.0, 1.0, 2.0], (B, m)); cb = rng.uniform(0.5, 2.0, (B, m))
        x = rng.uniform(-0.1, 0.1, (B, n)); sl = rng.uniform(0.2, 1.5, (B, m)); z = rng.uniform(0.1, 2, (B, m)); y = rng.uniform(-1, 1, (B, k))
        vars_ = np.concatenate([x, sl, y, z], axis=1)
        prob = Q.BatchedQP(n=n, k=k, m=m, J=T(J), r=T(r), lam=1e-3, A_eq=T(A), b_eq=T(b), cons_var=T(cv, torch.int32), cons_a=T(ca), cons_b=T(cb))
        s = Q.QPInteriorPointSolver(prob, force_generic=True)
        s.SetVariables(T(vars_))
        ip, st = s.Iterate(T(np.full(B, 1.0)), Q.COMPLEMENTARITY)
        after_it = s.variables().cpu().numpy().copy()
        s.SetVariables(T(vars_))
        out = s.Solve(Q.Params(initial_mu=1.0, sigma=0.1, termination_kkt_tol=1e-12, max_iterations=1, initial_guess_method=Q.USER_PROVIDED))
        after_so = s.variables().cpu().numpy().copy()
        rec = out.iterations.cpu().numpy()[:, 0, :]
        ipn = ip.cpu().numpy()
        rows.append({"build": tag, "shape": [n, k, m], "R": 5 if n + k <= 80 else 9,
                     "iterate": {"alpha_p": ipn[:, 1].tolist(), "alpha_d": ipn[:, 2].tolist(), "z0": ipn[:, 3].tolist(), "dz0": ipn[:, 4].tolist(), "lds_alpha_d": ipn[:, 5].tolist()},
                     "solve": {"alpha_p": rec[:, 9].tolist(), "alpha_d": rec[:, 10].tolist(), "z0": rec[:, 11].tolist(), "dz0": rec[:, 12].tolist(), "lds_alpha_d": rec[:, 13].tolist()},
                     "state_diff": float(np.abs(after_it - after_so).max())})
    for row in rows:
        print(json.dumps(row))


if __name__ == "__main__":
    main()
