#!/usr/bin/env python3
"""Error of the Newton direction on hard inputs (tests/stress_cases.py; run as `python tests/stress_table.py` on the GPU box) against a long-double solve of the unreduced system:
the device kernels (fused step = x+ form, generic step, fused Iterate = residual form) beside the oracle's two variants (reference
arithmetic with Eigen's explicit inverse / direct solve).  Prints one row per case; DESIGN.md section 2 quotes the table."""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch

from mini_opt_amd import qp as Q
from oracle import oracle as orc
from tests import stress_cases as S


def T(a, dt=torch.float64):
    return torch.as_tensor(np.ascontiguousarray(a), dtype=dt, device="cuda:0")


def device_problem(hb):
    return Q.BatchedQP(n=hb.n, k=hb.k, m=hb.m, J=T(hb.J), r=T(hb.r), lam=hb.lam, A_eq=T(hb.A_eq), b_eq=T(hb.b_eq),
                       cons_var=T(hb.cons_var, torch.int32), cons_a=T(hb.cons_a), cons_b=T(hb.cons_b))


def measure(name, hb):
    n, k, m = hb.n, hb.k, hb.m
    out = {}
    for label, force in (("fused_step", False), ("generic_step", True)):
        s = Q.QPInteriorPointSolver(device_problem(hb), force_generic=force)
        s.SetVariables(T(hb.vars))
        delta, alpha, status = s.NewtonStep(T(hb.mu), 0.995)
        out[label] = (delta.cpu().numpy().copy(), status.cpu().numpy().copy())
    s = Q.QPInteriorPointSolver(device_problem(hb))
    s.SetVariables(T(hb.vars))
    ip, status = s.Iterate(T(hb.mu), Q.COMPLEMENTARITY)
    out["fused_iterate"] = (s.delta_.cpu().numpy().copy(), status.cpu().numpy().copy())
    rows = []
    for p in range(hb.batch):
        G, c, A, b, cv, ca, cb = S.dense_problem(hb, p)
        tr = S.truth_direction(G, c, A, b, cv, ca, cb, hb.vars[p], hb.mu[p])
        Gl, cl, _ = orc.linearize_dense(hb.J[p], hb.r[p], hb.lam)
        o = orc.Solver(orc.QP(G=Gl, c=cl, A_eq=hb.A_eq[p].T, b_eq=hb.b_eq[p], cons_var=cv, cons_a=ca, cons_b=cb))
        v = hb.vars[p]
        sv, zv = v[n:n + m], v[n + m + k:]
        Sig = np.zeros(n)
        np.add.at(Sig, cv, ca * zv / sv * ca)
        H = np.zeros((n + k, n + k)); H[:n, :n] = G + np.diag(Sig); H[n:, :n] = A; H[:n, n:] = A.T
        rel = lambda d: float(np.abs(d - tr).max() / np.abs(tr).max())
        row = {"case": name, "p": p, "x_over_dx": float(np.abs(v[:n]).max() / max(np.abs(tr[:n]).max(), 1e-300)),
               "cond_kkt": float(np.linalg.cond(H)), "cond_H11": float(np.linalg.cond(H[:n, :n])),
               "zs_min": float((zv / sv).min()) if m else None, "zs_max": float((zv / sv).max()) if m else None}
        for inv in (True, False):
            st, d, _ = o.newton_step(v, hb.mu[p], 0.995, inv)
            row["oracle_inverse" if inv else "oracle_direct"] = rel(d) if st == 0 else f"status {st}"
        for label, (d, st) in out.items():
            row[label] = rel(d[p]) if st[p] == 0 else f"status {int(st[p])}"
        rows.append(row)
    return rows


def main():
    cases = [("late6 cfg2", S.late_states("cfg2", 4, 6)), ("late9 cfg2", S.late_states("cfg2", 4, 9)), ("late8 cfg3", S.late_states("cfg3", 4, 8)),
             ("late11 cfg3", S.late_states("cfg3", 4, 11)),
             ("cond 1e6 n32", S.ill_conditioned(32, 4, 16, 64, 4, 6)), ("cond 1e8 n64", S.ill_conditioned(64, 8, 32, 128, 4, 8)),
             ("cond 1e10 n32", S.ill_conditioned(32, 4, 16, 64, 4, 10)), ("cond 1e10 n64", S.ill_conditioned(64, 8, 32, 128, 4, 10)),
             ("rank-def, H11 regular", S.rank_deficient(32, 4, 32, 20, 4)), ("rank-def, H11 singular", S.rank_deficient(32, 8, 16, 20, 4)),
             ("rank-def n64, H11 singular", S.rank_deficient(64, 15, 32, 40, 4))]
    for name, hb in cases:
        for row in measure(name, hb):
            print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
