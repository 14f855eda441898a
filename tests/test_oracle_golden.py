"""Pins oracle/ (the CPU restatement) against the committed golden fixtures, i.e. against the reference's own
differential tests and known-answer tests (see tests/golden/gen_golden.py for what each fixture restates)."""
import json
import os

import numpy as np
import pytest

from oracle import oracle as orc

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def load(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


def qp_from(case):
    cons = case["cons"]
    return orc.QP(G=np.array(case["G"]).reshape(case["n"], case["n"]), c=case["c"],
                  A_eq=np.array(case["A_eq"]).reshape(case["k"], case["n"]), b_eq=case["b_eq"],
                  cons_var=[c[0] for c in cons], cons_a=[c[1] for c in cons], cons_b=[c[2] for c in cons])


@pytest.mark.parametrize("case", load("elimination.json"), ids=lambda c: c["name"])
def test_elimination_matches_full_system(case):
    """qp_test.cc:101-138: EvaluateKKTConditions -> ComputeLDLT -> SolveForUpdate(0) == full-system LU, 1e-12 abs."""
    s = orc.Solver(qp_from(case))
    s.variables[:] = case["state"]
    s.evaluate_kkt()
    np.testing.assert_allclose(s.r, case["expected_r"], rtol=0, atol=1e-12)
    assert s.compute_ldlt() == orc.ORC_OK
    H_expected = np.array(case["expected_H"])
    np.testing.assert_allclose(np.tril(s.H), np.tril(H_expected), rtol=0, atol=1e-13)
    # the upper triangle and the (2,2) block stay exactly zero (qp.cc:47, :289)
    assert np.all(np.triu(s.H, 1) == 0)
    s.solve_for_update(0.0)
    np.testing.assert_allclose(s.delta, case["expected_delta"], rtol=0, atol=case["tol_abs"])
    # direct-solve variant
    s.solve_for_update(0.0, direct=True)
    np.testing.assert_allclose(s.delta, case["expected_delta"], rtol=0, atol=case["tol_abs"])
    # the oracle's own BuildFullSystem + PartialPivLU (qp.cc:595-655, qp_test.cc:120-129)
    rc, d = s.full_system_step()
    assert rc == 0
    np.testing.assert_allclose(d, case["expected_delta"], rtol=0, atol=case["tol_abs"])


@pytest.mark.parametrize("case", load("elimination.json"), ids=lambda c: c["name"])
def test_solve_no_inequalities(case):
    """qp_test.cc:141-166"""
    s = orc.Solver(qp_from(case))
    s.variables[:] = case["state"]
    s.evaluate_kkt(False)
    assert s.compute_ldlt(False) == orc.ORC_OK
    s.solve_no_inequalities()
    x, _, y, _ = s.blocks(s.delta)
    n, k = case["n"], case["k"]
    exp = np.array(case["expected_delta_no_ineq_xy"])
    np.testing.assert_allclose(x, exp[:n], rtol=0, atol=1e-12)
    np.testing.assert_allclose(y, exp[n:n + k], rtol=0, atol=1e-12)


def test_compute_alpha_kat():
    """qp_test.cc:244-249"""
    g = load("alpha.json")
    h = g["head"]
    for c in g["cases"]:
        assert abs(orc.compute_alpha_vec(g["x"][:h], g["dx"][:h], c["tau"]) - c["alpha"]) < g["tol_abs"]


GUESS = {"NAIVE": orc.GUESS_NAIVE, "SOLVE_EQUALITY_CONSTRAINED": orc.GUESS_SOLVE_EQUALITY_CONSTRAINED}
STRAT = {"COMPLEMENTARITY": orc.COMPLEMENTARITY}


def check_kat_expectations(case, x, s, y, z, qp):
    e = case["expect"]
    if "x" in e:
        np.testing.assert_allclose(x, e["x"], rtol=0, atol=e["x_tol"])
    if "s" in e:
        np.testing.assert_allclose(s, e["s"], rtol=0, atol=e["s_tol"])
    for i, v in e.get("s_idx", []):
        assert abs(s[i] - v) < e["s_tol"]
    for i, v in e.get("z_idx", []):
        assert abs(z[i] - v) < e["s_tol"]
    for i, v in e.get("z_gt", []):
        assert z[i] > v
    if "eq_residual_tol" in e:
        np.testing.assert_allclose(qp.A_eq @ x + qp.b_eq, 0, atol=e["eq_residual_tol"])
    if "y_all_gt" in e:
        assert np.all(y > e["y_all_gt"])


@pytest.mark.parametrize("case", load("solve_kats.json"), ids=lambda c: c["name"])
def test_full_solve_kats(case):
    """qp_test.cc:252-471"""
    qp = qp_from(case)
    for guess in case["guesses"]:
        s = orc.Solver(qp)
        params = dict(case["params"])
        if "barrier_strategy" in params:
            params["barrier_strategy"] = STRAT[params["barrier_strategy"]]
        term, its = s.solve(initial_guess_method=GUESS[guess], **params)
        assert term == orc.SATISFIED_KKT_TOL, (case["name"], guess, term, len(its))
        x, sl, y, z = s.blocks(s.variables)
        check_kat_expectations(case, x, sl, y, z, qp)


@pytest.mark.parametrize("case", load("residual.json"), ids=lambda c: c["name"])
def test_update_hessian(case):
    """residual_test.cc:51-182"""
    n = case["full_size"]
    H = np.zeros((n, n), order="F")
    b = np.zeros(n)
    half = orc.update_hessian(case["index"], np.array(case["J"]), case["r"], H, b)
    np.testing.assert_allclose(H, np.array(case["expected_H_lower"]), rtol=0, atol=case["tol_abs"])
    assert np.all(np.triu(H, 1) == 0)  # residual_test.cc:130-134
    np.testing.assert_allclose(b, case["expected_b"], rtol=0, atol=case["tol_abs"])
    assert abs(half - case["expected_half_sq"]) < 1e-14
    # untouched cells stay exactly zero (residual_test.cc:143-147)
    mask = np.zeros((n, n), bool)
    for i in case["index"]:
        for j in case["index"]:
            mask[i, j] = True
    assert np.all(H[~mask] == 0)


def rel_inf(a, b):
    return np.max(np.abs(a - b)) / np.max(np.abs(b))


@pytest.mark.parametrize("cfg", ["cfg1", "cfg2", "cfg3", "cfg4"])
@pytest.mark.parametrize("use_inverse", [True, False])
def test_synthetic_step_vs_numpy_full_system(cfg, use_inverse):
    """Oracle Newton step (with mu) on the synthetic fixtures == numpy full-system solve, <= 1e-11 rel-inf."""
    z = np.load(os.path.join(GOLDEN, "synthetic.npz"))
    g = lambda k: z[f"{cfg}_{k}"]
    J = g("J")
    B, m_r, n = J.shape
    k = g("b_eq").shape[1]
    m = g("cons_var").shape[1]
    delta, alpha, status, _ = orc.batched_newton_step(
        n, k, m, J=J, r=g("r"), lam=float(g("lam")), A_eq=g("A_eq"), b_eq=g("b_eq"), cons_var=g("cons_var"),
        cons_a=g("cons_a"), cons_b=g("cons_b"), vars_=g("vars"), mu=g("mu"), use_inverse=use_inverse)
    assert np.all(status == 0)
    for p in range(B):
        assert rel_inf(delta[p], g("delta")[p]) < 1e-11
    # linearisation itself
    G, c, _ = orc.linearize_dense(J[0], g("r")[0], float(g("lam")))
    np.testing.assert_allclose(np.tril(G), g("G_lower")[0], rtol=1e-13, atol=1e-13)
    np.testing.assert_allclose(c, g("c")[0], rtol=1e-13, atol=1e-13)
    # alpha recomputed in numpy from the state and delta (qp.cc:485-507)
    v = g("vars")
    for p in range(B):
        for blk, a in ((slice(n, n + m), alpha[p, 0]), (slice(n + m + k, n + 2 * m + k), alpha[p, 1])):
            val, d = v[p][blk], delta[p][blk]
            cand = [-0.995 * val[i] / d[i] for i in range(m) if val[i] + d[i] <= 0 and abs(d[i]) > 0]
            assert abs(a - min([1.0] + cand)) < 1e-12


def test_generated_problems():
    """TestGeneratedProblems (qp_test.cc:527-574) with the reference's own assertions: 1000 random N=8 QPs (diagonal PD
    scaling, see tests/helpers.py), COMPLEMENTARITY, termination_kkt_tol 1e-12, max 30 iterations, both initial-guess
    methods: |x - x*|inf <= 5e-5 and |s|inf <= 5e-5 on every problem (qp_test.cc:555-562; every bound is active at x*),
    and 4 * iterations(SOLVE_EQUALITY_CONSTRAINED) < iterations(NAIVE) (qp_test.cc:572-573)."""
    from tests.helpers import generated_qps
    n = 8
    totals = {orc.GUESS_NAIVE: 0, orc.GUESS_SOLVE_EQUALITY_CONSTRAINED: 0}
    for p, (G, c, cons, x_solution) in enumerate(generated_qps(1000, n)):
        for method in totals:
            s = orc.Solver(orc.QP(G=G, c=c, cons_var=[q[0] for q in cons], cons_a=[q[1] for q in cons],
                                  cons_b=[q[2] for q in cons]))
            term, its = s.solve(termination_kkt_tol=1e-12, max_iterations=30, initial_guess_method=method)
            totals[method] += len(its)
            assert term in (orc.SATISFIED_KKT_TOL, orc.MAX_ITERATIONS) and len(its) <= 30
            m = len(cons)
            assert np.abs(s.variables[:n] - x_solution).max() <= 5e-5, (p, method, term)
            if m:
                assert np.abs(s.variables[n:n + m]).max() <= 5e-5, (p, method, term)
    assert totals[orc.GUESS_SOLVE_EQUALITY_CONSTRAINED] * 4 < totals[orc.GUESS_NAIVE], totals


def test_generated_problems_other_streams():
    """The same construction on two other random streams: a run may stop at MAX_ITERATIONS (about 1 NAIVE run in 2000
    does, see tests/helpers.py), but every run that reports SATISFIED_KKT_TOL meets the reference's 5e-5 bounds."""
    from tests.helpers import generated_qps
    n = 8
    at_max = runs = 0
    for seed in (1234, 7):
        for (G, c, cons, x_solution) in generated_qps(300, n, seed=seed):
            for method in (orc.GUESS_NAIVE, orc.GUESS_SOLVE_EQUALITY_CONSTRAINED):
                s = orc.Solver(orc.QP(G=G, c=c, cons_var=[q[0] for q in cons], cons_a=[q[1] for q in cons],
                                      cons_b=[q[2] for q in cons]))
                term, its = s.solve(termination_kkt_tol=1e-12, max_iterations=30, initial_guess_method=method)
                runs += 1
                if term == orc.MAX_ITERATIONS:
                    at_max += 1
                    continue
                assert np.abs(s.variables[:n] - x_solution).max() <= 5e-5
                if cons:
                    assert np.abs(s.variables[n:n + len(cons)]).max() <= 5e-5
    assert at_max <= 0.005 * runs, (at_max, runs)


def test_dense_generated_problems_kkt_certificate():
    """OUR stress test, not a reference test: the dense-PD variant of the generator (cond(G) up to 1e11, coupled bounds).
    Every run that reports SATISFIED_KKT_TOL is certified optimal by an independent numpy check of the KKT conditions of
    the convex QP (stationarity, feasibility, z >= 0, complementarity)."""
    from tests.helpers import dense_generated_qps
    n = 8
    satisfied = 0
    for (G, c, cons, _) in dense_generated_qps(100, n):
        for method in (orc.GUESS_NAIVE, orc.GUESS_SOLVE_EQUALITY_CONSTRAINED):
            s = orc.Solver(orc.QP(G=G, c=c, cons_var=[q[0] for q in cons], cons_a=[q[1] for q in cons],
                                  cons_b=[q[2] for q in cons]))
            term, its = s.solve(termination_kkt_tol=1e-12, max_iterations=30, initial_guess_method=method)
            assert term in (orc.SATISFIED_KKT_TOL, orc.MAX_ITERATIONS)
            if term != orc.SATISFIED_KKT_TOL:
                continue
            satisfied += 1
            m = len(cons)
            x, sl, z = s.variables[:n], s.variables[n:n + m], s.variables[n + m:]
            grad = G @ x + c
            scale = max(1.0, np.abs(G @ x).max(), np.abs(c).max())
            for i, (v, a, b) in enumerate(cons):
                grad[v] -= a * z[i]
                assert a * x[v] + b >= -1e-9 * max(1.0, abs(b))          # primal feasibility
                assert z[i] >= 0                                          # dual feasibility
                assert abs((a * x[v] + b) * z[i]) <= 1e-5 * scale         # complementarity
            assert np.abs(grad).max() <= 1e-9 * scale                     # stationarity
    assert satisfied > 0


def test_predictor_corrector_and_fixed_decrease_converge():
    """The three BarrierStrategy values (structs.hpp:24-31) all reach the optimum of a KAT problem."""
    case = [c for c in load("solve_kats.json") if c["name"] == "TestWithInequalitiesAndEqualities"][0]
    for strat in (orc.COMPLEMENTARITY, orc.FIXED_DECREASE, orc.PREDICTOR_CORRECTOR):
        s = orc.Solver(qp_from(case))
        term, its = s.solve(termination_kkt_tol=1e-10, initial_mu=0.1, sigma=0.1, max_iterations=60, barrier_strategy=strat)
        assert term == orc.SATISFIED_KKT_TOL, strat
        np.testing.assert_allclose(s.variables[:3], [0.5, -1.0, 2.0], atol=1e-6)


def _random_qp(rng, n, k, m):
    """A well-posed QP around a strictly feasible point (the construction of tests/test_gpu_fuzz.py)."""
    m_r = n + 4
    J = rng.uniform(-1, 1, (m_r, n)); r = rng.uniform(-1, 1, m_r)
    A = rng.uniform(-1, 1, (k, n))
    cv = rng.integers(0, n, m).astype(np.int32); ca = rng.choice([-1.0, 1.0, 2.0], m)
    x0 = rng.uniform(-0.5, 0.5, n)
    cb = -ca * x0[cv] + rng.uniform(0.05, 0.5, m)
    G = np.tril(J.T @ J + 1e-2 * np.eye(n))
    return orc.QP(G=G, c=J.T @ r, A_eq=A if k else None, b_eq=-(A @ x0) if k else None, cons_var=cv, cons_a=ca, cons_b=cb)


@pytest.mark.parametrize("strategy", [orc.COMPLEMENTARITY, orc.FIXED_DECREASE, orc.PREDICTOR_CORRECTOR])
@pytest.mark.parametrize("gate", [0, 1])
def test_margin_replay_is_the_oracles_solve(strategy, gate):
    """oracle/margins.py replays orc_solve through the oracle's primitives to log decision margins: it must BE the oracle's Solve
    (termination, iteration count, final state bit for bit) for every strategy, guess and with decrease_mu_only_on_small_error."""
    from oracle import margins as M
    rng = np.random.default_rng(100 + 10 * strategy + gate)
    for trial in range(12):
        n = int(rng.integers(3, 20)); k = int(rng.integers(0, n // 2 + 1)); m = int(rng.integers(1, 2 * n))
        qp = _random_qp(rng, n, k, m)
        guess = orc.GUESS_SOLVE_EQUALITY_CONSTRAINED if (k and trial % 2) else orc.GUESS_NAIVE
        kw = dict(initial_mu=1.0, sigma=0.1, termination_kkt_tol=1e-8, max_iterations=15, barrier_strategy=strategy,
                  initial_guess_method=guess, decrease_mu_only_on_small_error=gate, initialize_mu_with_complementarity=trial % 3 == 0)
        o = orc.Solver(qp)
        term, its = o.solve(**kw)
        t2, n2, v2, marg = M.solve_with_margins(qp, **kw)
        assert (t2, n2) == (term, len(its))
        assert np.array_equal(v2, o.variables)
        assert marg and all(np.isfinite(mm[2]) or mm[1] == "mu_gate" for mm in marg)
        assert any(mm[1] == "mu_gate" for mm in marg) == bool(gate)


def test_decrease_mu_only_on_small_error_changes_the_trajectory():
    """qp.cc:140-146 / qp.hpp:154-157: with the flag set mu is only decreased once kkt_after.Max() <= mu.  On these problems the first
    iterations end with kkt_after.Max() > mu, so the gated run keeps its mu where the ungated one decreases it: different iteration
    records, same optimum.  (The device tests use the same problems, tests/test_gpu_parity.py::test_decrease_mu_only_on_small_error.)"""
    rng = np.random.default_rng(5)
    held = 0
    for trial in range(8):
        qp = _random_qp(rng, 12, 3, 10)
        kw = dict(initial_mu=1e-3, sigma=0.1, termination_kkt_tol=1e-8, max_iterations=40, barrier_strategy=orc.FIXED_DECREASE)
        a, b = orc.Solver(qp), orc.Solver(qp)
        ta, ia = a.solve(decrease_mu_only_on_small_error=0, **kw)
        tb, ib = b.solve(decrease_mu_only_on_small_error=1, **kw)
        assert ta == tb == orc.SATISFIED_KKT_TOL
        np.testing.assert_allclose(a.variables[:12], b.variables[:12], atol=2e-5)   # both within the complementarity tolerance 1e-6 of the optimum
        # the gate held mu at least once: some iteration of the gated run starts with the mu of the one before
        mus = [it.ip.mu for it in ib]
        held += int(any(mus[i + 1] == mus[i] for i in range(len(mus) - 1)))
        kmax0 = max(ib[0].kkt_final.r_dual, ib[0].kkt_final.r_comp, ib[0].kkt_final.r_primal_eq, ib[0].kkt_final.r_primal_ineq)
        if mus[1] == mus[0]:
            assert kmax0 > mus[0]
    assert held >= 6, held
