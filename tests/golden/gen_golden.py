#!/usr/bin/env python3
"""Generates the committed golden fixtures in tests/golden/ (run from the repo root: python tests/golden/gen_golden.py).

The reference (C++/Eigen) cannot be built in this image (Eigen is an un-vendored, absent submodule), so the
fixtures restate the DATA of the reference's own tests and compute the expected values INDEPENDENTLY of
oracle/ with numpy:

  * elimination.json  -- the four QPSolverTest.TestElimination* problems (test/qp_test.cc:168-241), the dummy
    state of qp_test.cc:84-97, and delta = flip_yz(solve(H_full, -r_full)) with H_full / r_full built exactly as
    QPInteriorPointSolver::BuildFullSystem (source/qp.cc:595-655) -- i.e. the right-hand side of the reference's
    differential assertion ASSERT_EIGEN_NEAR(update, solver.delta_, 1e-12) (qp_test.cc:137).  Also the
    "no inequalities" variant of qp_test.cc:141-166.
  * alpha.json        -- TestComputeAlpha (qp_test.cc:244-249).
  * solve_kats.json   -- the six full-Solve known-answer tests (qp_test.cc:252-471): problem, params, expected optimum.
  * residual.json     -- residual_test.cc:51-182: DummyFunction J / r at the tested points, index sets, J^T J, J^T r.
  * synthetic.npz     -- a few problems per BASELINE.json config from mini_opt_amd.synth (host code, numpy only)
    with delta from the same full-system numpy solve (fp64).

Only numpy is used.  Nothing from oracle/ or from /root/reference is imported or executed.
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.abspath(os.path.join(HERE, "..", "..")))


def build_quadratic(roots):
    """QPSolverTest::BuildQuadratic, qp_test.cc:58-73"""
    n = len(roots)
    G = np.zeros((n, n))
    c = np.zeros(n)
    for i, (a, b) in enumerate(roots):
        G[i, i] = a * a
        c[i] = -2 * a * b
    return G, c


def var_ge(v, value):  # Var(v) >= value, qp.hpp:86-88
    return (v, 1.0, -value)


def var_le(v, value):  # Var(v) <= value, qp.hpp:81-83
    return (v, -1.0, value)


def dummy_state(x, m, k):
    """PutDummyValuesInSlacksAndMultipliers, qp_test.cc:84-97; order [x|s|y|z]"""
    s = np.array([2.0 / (i + 1) for i in range(m)])
    z = np.array([0.5 * (i + 1) for i in range(m)])
    y = np.array([float((q + 1) * (q + 1)) for q in range(k)])
    return np.concatenate([np.asarray(x, float), s, y, z])


def full_system(G, c, A, b, cons, state):
    """BuildFullSystem, qp.cc:595-655 (G symmetric full here)."""
    n = G.shape[0]
    k = A.shape[0]
    m = len(cons)
    V = n + 2 * m + k
    x, s, y, z = state[:n], state[n:n + m], state[n + m:n + m + k], state[n + m + k:]
    H = np.zeros((V, V))
    H[:n, :n] = G
    H[:n, n + m:n + m + k] = A.T
    H[n + m:n + m + k, :n] = A
    Ai = np.zeros((m, n))
    bi = np.zeros(m)
    for i, (v, a, bb) in enumerate(cons):
        Ai[i, v] = a
        bi[i] = bb
    if m:
        H[:n, n + m + k:] = Ai.T
        H[n + m + k:, :n] = Ai
        H[n + m + k:, n:n + m] = -np.eye(m)
        H[n:n + m, n:n + m] = np.diag(z / s)
        H[n:n + m, n + m + k:] = -np.eye(m)
    r = np.zeros(V)
    r[:n] = G @ x + c - A.T @ y - (Ai.T @ z if m else 0)
    r[n:n + m] = z
    r[n + m:n + m + k] = A @ x + b
    r[n + m + k:] = Ai @ x + bi - s
    return H, r


def kkt_residual(G, c, A, b, cons, state):
    """EvaluateKKTConditions, qp.cc:391-420 -> [r_d | r_comp | r_pe | r_pi] (mu not applied)."""
    n = G.shape[0]
    k = A.shape[0]
    m = len(cons)
    x, s, y, z = state[:n], state[n:n + m], state[n + m:n + m + k], state[n + m + k:]
    r_d = G @ x + c - A.T @ y
    r_pi = np.zeros(m)
    for i, (v, a, bb) in enumerate(cons):
        r_d[v] -= a * z[i]
        r_pi[i] = a * x[v] + bb - s[i]
    return np.concatenate([r_d, s * z, A @ x + b, r_pi])


def reduced_H(G, A, cons, state):
    """ComputeLDLT assembly, qp.cc:289-298 (symmetric full matrix returned; the reference stores the lower part)."""
    n = G.shape[0]
    k = A.shape[0]
    m = len(cons)
    s, z = state[n:n + m], state[n + m + k:]
    H = np.zeros((n + k, n + k))
    H[:n, :n] = G
    H[n:, :n] = A
    H[:n, n:] = A.T
    for i, (v, a, _) in enumerate(cons):
        H[v, v] += a * (z[i] / s[i]) * a
    return H


def full_step(G, c, A, b, cons, state):
    """qp_test.cc:120-129"""
    n = G.shape[0]
    k = A.shape[0]
    m = len(cons)
    H, r = full_system(G, c, A, b, cons, state)
    d = np.linalg.solve(H, -r)
    d[n + m:n + m + k] *= -1
    d[n + m + k:] *= -1
    return d, np.linalg.cond(H)


def tolist(a):
    return np.asarray(a, float).tolist()


def elimination_cases():
    cases = []

    def add(name, roots, A, b, cons, x_guess):
        G, c = build_quadratic(roots)
        n = len(roots)
        A = np.zeros((0, n)) if A is None else np.asarray(A, float)
        b = np.zeros(0) if b is None else np.asarray(b, float)
        state = dummy_state(x_guess, len(cons), A.shape[0])
        delta, cond = full_step(G, c, A, b, cons, state)
        # "no inequalities" variant (qp_test.cc:141-166): the reduced problem solved with its own dummy state
        state_r = dummy_state(x_guess, 0, A.shape[0])
        delta_r, _ = full_step(G, c, A, b, [], state_r)
        cases.append(dict(
            name=name, n=n, k=int(A.shape[0]), m=len(cons), G=tolist(G), c=tolist(c), A_eq=tolist(A), b_eq=tolist(b),
            cons=[[int(v), float(a), float(bb)] for v, a, bb in cons], state=tolist(state),
            expected_r=tolist(kkt_residual(G, c, A, b, cons, state)), expected_H=tolist(reduced_H(G, A, cons, state)),
            expected_delta=tolist(delta), expected_delta_no_ineq_xy=tolist(delta_r), cond_full=float(cond),
            tol_abs=1e-12, cite="test/qp_test.cc:101-138,141-166"))

    add("TestEliminationNoConstraints", [(0.5, 2.0), (5.0, 25.0), (3.0, 9.0)], None, None, [], [0.0, -0.1, -0.3])
    A = np.zeros((1, 3)); A[0, 1] = 1.0; A[0, 2] = -1.0
    add("TestEliminationEqualityConstraints", [(1.0, -0.5), (2.0, -2.0), (-4.0, 5.0)], A, [-0.5], [],
        [0.3, -0.1, -0.3])
    add("TestEliminationInequalityConstraints", [(1.5, 3.0), (-1.0, 4.0)], None, None,
        [var_ge(1, 0), var_le(0, 5), var_ge(0, -5)], [0.0, 2.0])
    A = np.zeros((2, 7)); A[0, 1] = 2.0; A[0, 4] = -1.0; A[1, 0] = 3.0
    add("TestEliminationAllConstraints",
        [(0.5, 2.0), (5.0, 25.0), (3.0, 9.0), (4.0, 1.0), (1.2, 2.4), (-1.0, 2.0), (-0.5, 2.0)], A, [0.5, -2.0],
        [(3, 4.0, -8.0), (5, 2.0, 1.0), (6, 1.0, 0.0)], [0.0, 0.1, 0.2, 0.55, 0.3, 0.7, 1.0])
    return cases


def solve_kats():
    kats = []

    def from_residual(Jdiag, r0):
        J = np.diag(np.asarray(Jdiag, float))
        r0 = np.asarray(r0, float)
        return J.T @ J, J.T @ r0  # UpdateHessian at x = 0 (residual.hpp:186-226), G lower == full (diagonal)

    def add(name, G, c, A, b, cons, params, guesses, expect, cite):
        n = len(c)
        A = np.zeros((0, n)) if A is None else np.atleast_2d(np.asarray(A, float))
        b = np.zeros(0) if b is None else np.asarray(b, float).ravel()
        kats.append(dict(name=name, n=n, k=int(A.shape[0]), m=len(cons), G=tolist(G), c=tolist(c), A_eq=tolist(A),
                         b_eq=tolist(b), cons=[[int(v), float(a), float(bb)] for v, a, bb in cons], params=params,
                         guesses=guesses, expect=expect, cite=cite))

    both = ["NAIVE", "SOLVE_EQUALITY_CONSTRAINED"]
    G, c = from_residual([1.0], [-5.0])
    add("TestWithSingleInequality", G, c, None, None, [var_le(0, 4)],
        dict(termination_kkt_tol=1e-9, sigma=0.1, initial_mu=0.1), ["NAIVE"],
        dict(termination="SATISFIED_KKT_TOL", x=[4.0], x_tol=1e-6, s=[0.0], s_tol=1e-6, z_gt=[[0, 1.0 - 1e-6]]),
        "test/qp_test.cc:252-287")
    G, c = from_residual([1.0, -4.0], [-2.0, -16.0])
    add("TestWithInequalitiesActive", G, c, None, None, [var_le(0, 1.0), var_ge(1, -3.0)],
        dict(termination_kkt_tol=1e-12, sigma=0.1, initial_mu=0.1), both,
        dict(termination="SATISFIED_KKT_TOL", x=[1.0, -3.0], x_tol=1e-6, s=[0.0, 0.0], s_tol=1e-6),
        "test/qp_test.cc:290-332")
    G, c = from_residual([1.0, -1.0, 0.5], [-1.0, -3.0, -5.0])
    add("TestWithInequalitiesPartiallyActive", G, c, None, None, [var_ge(1, -2.0), var_ge(0, -3.5)],
        dict(termination_kkt_tol=1e-12, sigma=0.1, initial_mu=0.1, barrier_strategy="COMPLEMENTARITY"), both,
        dict(termination="SATISFIED_KKT_TOL", x=[1.0, -2.0, 10.0], x_tol=1e-6, s_idx=[[0, 0.0]], z_idx=[[1, 0.0]],
             s_tol=1e-6), "test/qp_test.cc:335-375")
    G, c = build_quadratic([(1.0, 0.5), (3.0, 2.0), (-4.0, 5.0), (0.25, 4)])
    A = np.zeros((2, 4)); A[0, 0] = 1; A[0, 2] = -0.5; A[1, 1] = 0.25; A[1, 3] = 1.0
    add("TestWithEqualitiesOnly", G, c, A, [3.0, -2.0], [], dict(termination_kkt_tol=1e-6, max_iterations=1),
        ["NAIVE"], dict(termination="SATISFIED_KKT_TOL", eq_residual_tol=1e-9), "test/qp_test.cc:379-411")
    G, c = build_quadratic([(1.0, -0.5), (1.0, -0.25), (1.0, 1.0)])
    add("TestWithFullyConstrainedEqualities", G, c, np.eye(3), [-1.0, -2.0, -3.0], [],
        dict(termination_kkt_tol=1e-6, max_iterations=1), ["NAIVE"],
        dict(termination="SATISFIED_KKT_TOL", x=[1.0, 2.0, 3.0], x_tol=1e-9, y_all_gt=1e-2),
        "test/qp_test.cc:414-436")
    G, c = build_quadratic([(1.0, 1.0), (5.0, -10.0), (10.0, 2.0)])
    add("TestWithInequalitiesAndEqualities", G, c, [[0.0, 0.0, 1.0]], [-2.0], [var_le(0, 0.5), var_ge(1, -1.0)],
        dict(termination_kkt_tol=1e-12, sigma=0.1, initial_mu=0.1), both,
        dict(termination="SATISFIED_KKT_TOL", x=[0.5, -1.0, 2.0], x_tol=1e-6, s=[0.0, 0.0], s_tol=1e-6),
        "test/qp_test.cc:439-471")
    return kats


def dummy_function(p):
    """residual_test.cc:14-27"""
    x, y, z = p
    f = np.array([x * x + x * y - z * z * y, x * y * y - z * y * y + z * z * x])
    J = np.array([[x + y, x - z * z, y], [y * y + z * z, x - z, -y * y + x]])
    # NOTE: the reference's J(0,0) is (x + y) although d/dx(x*x + x*y) = 2x + y; the tests only use J as data
    # (J^T J, J^T r), so the fixture restates the reference's J verbatim.
    return f, J


def residual_cases():
    out = []
    for name, index, full, params in [
        ("TestStaticResidualSimple", [0, 1, 2], 3, [-0.5, 1.2, 0.3]),
        ("TestStaticResidualOutOfOrder", [2, 0, 1], 3, [0.23, -0.9, 1.11]),
        ("TestStaticResidualSparseIndex", [5, 1, 3], 7, [0.99, -0.23, 2.2]),
        ("TestDynamicParameterVector", [0, 1, 2], 3, [0.099, -0.5, 0.76]),
    ]:
        f, J = dummy_function(params)
        S = np.zeros((3, full))  # local_D_global, residual_test.cc:33-41
        for row, g in enumerate(index):
            S[row, g] = 1
        Hfull = S.T @ (J.T @ J) @ S
        out.append(dict(name=name, index=index, full_size=full, params_local=tolist(params), J=tolist(J), r=tolist(f),
                        expected_H_lower=tolist(np.tril(Hfull)), expected_b=tolist(S.T @ (J.T @ f)),
                        expected_half_sq=float(0.5 * f @ f), tol_abs=1e-12, cite="test/residual_test.cc:51-182"))
    return out


def synthetic():
    from mini_opt_amd import synth
    arrays = {}
    for cfg, count in [("cfg1", 8), ("cfg2", 6), ("cfg3", 4), ("cfg4", 2)]:
        d = synth.CONFIGS[cfg]
        n, k, m, m_r = d["n"], d["k"], d["m"], d["m_r"]
        B = synth.make_batch(n, k, m, m_r, count)
        J, r = B.J, B.r
        if d["dtype"] == "f32":  # cfg 4: inputs rounded to fp32 first; expected delta still in fp64
            rd = lambda a: a.astype(np.float32).astype(np.float64)
            B.J, B.r, B.A_eq, B.b_eq, B.cons_a, B.cons_b, B.vars, B.mu = map(
                rd, (B.J, B.r, B.A_eq, B.b_eq, B.cons_a, B.cons_b, B.vars, B.mu))
            B.lam = float(np.float32(B.lam))
            J, r = B.J, B.r
        deltas, conds, Gs, cs = [], [], [], []
        for p in range(count):
            G = J[p].T @ J[p] + B.lam * np.eye(n)
            c = J[p].T @ r[p]
            A = B.A_eq[p].T  # [n][k] memory == k x n column-major
            cons = list(zip(B.cons_var[p].tolist(), B.cons_a[p].tolist(), B.cons_b[p].tolist()))
            # mu enters the rhs: replace s^-1 r_comp = z by z - mu/s  (qp.cc:246-247, :341, :362)
            H, rr = full_system(G, c, A, B.b_eq[p], cons, B.vars[p])
            s = B.vars[p][n:n + m]
            rr[n:n + m] -= B.mu[p] / s
            dlt = np.linalg.solve(H, -rr)
            dlt[n + m:n + m + k] *= -1
            dlt[n + m + k:] *= -1
            deltas.append(dlt)
            conds.append(np.linalg.cond(H))
            Gs.append(np.tril(G))
            cs.append(c)
        pre = cfg + "_"
        for key, val in dict(J=B.J, r=B.r, A_eq=B.A_eq, b_eq=B.b_eq, cons_var=B.cons_var, cons_a=B.cons_a,
                             cons_b=B.cons_b, vars=B.vars, mu=B.mu, delta=np.array(deltas), cond=np.array(conds),
                             G_lower=np.array(Gs), c=np.array(cs), lam=np.array(B.lam)).items():
            arrays[pre + key] = val
    return arrays


def main():
    def dump(name, obj):
        with open(os.path.join(HERE, name), "w") as f:
            json.dump(obj, f, indent=1)
        print("wrote", name)

    dump("elimination.json", elimination_cases())
    dump("alpha.json", dict(x=[1.0, 0.8, 1.2, 0.9], dx=[-2.0, 0.6, -1.3, 0.5], head=3,
                            cases=[dict(tau=1.0, alpha=0.5), dict(tau=0.9, alpha=0.45)], tol_abs=1e-12,
                            cite="test/qp_test.cc:244-249"))
    dump("solve_kats.json", solve_kats())
    dump("residual.json", residual_cases())
    np.savez_compressed(os.path.join(HERE, "synthetic.npz"), **synthetic())
    print("wrote synthetic.npz")


if __name__ == "__main__":
    main()
