"""GPU parity tests of the SQP outer loop: mo_fill_qp / mo_nonlinear_errors /
mo_qp_cost_derivative against numpy, and mo_nls_solve against the NLS oracle on the reference's own test problems
(nonlinear_test.cc:390-826), every initial guess of a test being one problem of the batch."""
import numpy as np
import pytest
import torch

from mini_opt_amd import nls as NLS
from mini_opt_amd import qp as Q
from oracle import nls_oracle as N
from tests import nls_problems as P

pytestmark = pytest.mark.gpu


def T(a, dt=torch.float64):
    return torch.as_tensor(np.ascontiguousarray(a), dtype=dt, device="cuda:0").contiguous()


def test_fill_qp_errors_and_derivative_vs_numpy():
    rng = np.random.default_rng(5)
    B, n, k, m, m_r = 37, 12, 3, 7, 20
    J = rng.uniform(-1, 1, (B, m_r, n)); r = rng.uniform(-1, 1, (B, m_r))
    A = rng.uniform(-1, 1, (B, k, n)); b = rng.uniform(-1, 1, (B, k)); b[:, 0] = 0.0    # sign(0) = 0 (nonlinear.cc:440-450)
    cv = rng.integers(0, n, (B, m)).astype(np.int32); ca = rng.choice([-1.0, 1.0, 2.0], (B, m)); cb = rng.uniform(-1, 1, (B, m))
    x = rng.uniform(-2, 2, (B, n)); dx = rng.uniform(-1, 1, (B, n))
    lam = rng.uniform(0, 0.5, B); lam[::3] = 0.0
    prob = Q.BatchedQP(n=n, k=k, m=m, J=T(J), r=T(r), lam_vec=T(lam), A_eq=T(A.transpose(0, 2, 1)), b_eq=T(b),
                       cons_var=T(cv, torch.int32), cons_a=T(ca), cons_b=T(cb))
    G, c, cbs, err, status = NLS.fill_qp(prob, T(x))
    assert torch.all(status == 0)
    Gref = np.einsum("bqi,bqj->bij", J, J) + lam[:, None, None] * np.eye(n)
    Gl = np.tril(G.cpu().numpy().transpose(0, 2, 1))
    np.testing.assert_allclose(Gl, np.tril(Gref), rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(c.cpu().numpy(), np.einsum("bqi,bq->bi", J, r), rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(cbs.cpu().numpy(), ca * np.take_along_axis(x, cv, 1) + cb, rtol=1e-14, atol=1e-14)   # ShiftTo
    np.testing.assert_allclose(err.cpu().numpy(), np.stack([0.5 * np.sum(r * r, 1), np.sum(np.abs(b), 1)], 1), rtol=1e-13)
    # directional derivatives, J-level and QP-level
    dref = np.stack([np.einsum("bi,bi->b", np.einsum("bqi,bq->bi", J, r), dx),
                     np.einsum("bk,bk->b", np.sign(b), np.einsum("bkn,bn->bk", A, dx))], 1)
    qref = np.einsum("bi,bij,bj->b", dx, Gref, dx)
    d1, q1 = NLS.qp_cost_derivative(prob, T(dx))
    np.testing.assert_allclose(d1.cpu().numpy(), dref, rtol=1e-11, atol=1e-12)
    np.testing.assert_allclose(q1.cpu().numpy(), qref, rtol=1e-11)
    prob2 = Q.BatchedQP(n=n, k=k, G=G, c=c, A_eq=T(A.transpose(0, 2, 1)), b_eq=T(b))
    d2, q2 = NLS.qp_cost_derivative(prob2, T(dx))
    np.testing.assert_allclose(d2.cpu().numpy(), dref, rtol=1e-11, atol=1e-12)
    np.testing.assert_allclose(q2.cpu().numpy(), qref, rtol=1e-11)
    # a bad constraint index is reported per problem
    cv2 = cv.copy(); cv2[4, 2] = n
    prob.cons_var = T(cv2, torch.int32)
    _, _, cbs2, _, st2 = NLS.fill_qp(prob, T(x))
    st2 = st2.cpu().numpy()
    assert st2[4] == 4 and np.all(np.delete(st2, 4) == 0) and np.isnan(cbs2.cpu().numpy()[4, 2])


def run_both(oprob, dprob, oparams, dparams, guesses):
    """Solve every guess with the oracle (one at a time) and on the device (one batch)."""
    guesses = np.array(guesses, dtype=float)
    B = len(guesses)
    nls = NLS.ConstrainedNonlinearLeastSquares(dprob, batch=B)
    out = nls.Solve(dparams, T(guesses))
    dev_x = nls.variables().cpu().numpy()
    dev_term = out.termination_state.cpu().numpy(); dev_nit = out.num_iterations.cpu().numpy()
    ref = N.ConstrainedNonlinearLeastSquares(oprob)
    ref_x, ref_term, ref_nit, ref_logs = [], [], [], []
    for g in guesses:
        term, logs = ref.solve(oparams, g)
        ref_x.append(ref.variables.copy()); ref_term.append(term); ref_nit.append(len(logs)); ref_logs.append(logs)
    return out, dev_x, dev_term, dev_nit, np.array(ref_x), np.array(ref_term), np.array(ref_nit), ref_logs


# Outer-loop decisions (Armijo test, exit tests, step-size validity ...) compare quantities the device and the oracle compute in a different
# operation order: they may only disagree where the oracle's deciding quantity sat within NLS_KNIFE_EDGE (relative) of its threshold --
# or where an inner interior-point solve itself sat on a knife edge (oracle/margins.py), after which the two runs continue from QP
# solutions that differ at the level of the QP tolerance.
NLS_KNIFE_EDGE = 1.0e-8


def knife_edge_rule(label, oprob, oparams, guesses, term, nit, rterm, rnit, retraction=None, max_fraction=0.05):
    """The problems whose (termination, iteration count) differ from the oracle's must each sit on a knife edge of the ORACLE's run: the
    oracle is re-run with decision margins logged.  Returns the mask of agreeing problems; the findings go to
    gpurun_out/nls_disagreements.jsonl."""
    import json
    import os
    from oracle import margins as M
    guesses = np.asarray(guesses, dtype=float)
    same = (np.asarray(term) == np.asarray(rterm)) & (np.asarray(nit) == np.asarray(rnit))
    rows = []
    for p in np.flatnonzero(~same):
        o = N.ConstrainedNonlinearLeastSquares(oprob, retraction=retraction, track_margins=True)
        o.solve(oparams, guesses[p])
        qp = min([mm[2] for mm in o.margins if mm[1] == "qp"], default=np.inf)
        outer = min([(mm[2], mm[0], mm[1]) for mm in o.margins if mm[1] != "qp"], default=(np.inf, -1, ""))
        rows.append({"test": label, "problem": int(p), "device": [int(term[p]), int(nit[p])], "oracle": [int(rterm[p]), int(rnit[p])],
                     "min_qp_margin": float(qp), "min_outer_margin": float(outer[0]), "outer_decision": [int(outer[1]), outer[2]]})
        assert qp < 1.0 or outer[0] < NLS_KNIFE_EDGE, (    # ("qp" margins are margin / threshold of the decision's kind, oracle/margins.py)
            f"{label}: problem {p} ends {(term[p], nit[p])} on the device and {(rterm[p], rnit[p])} in the oracle although no decision of the "
            f"oracle's run was near its threshold (inner QP {qp:.2e} x its knife-edge threshold, outer loop {outer[0]:.2e} at {outer[1:]})")
    if rows:
        out_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
        os.makedirs(out_dir, exist_ok=True)
        with open(os.path.join(out_dir, "nls_disagreements.jsonl"), "a") as f:
            for row in rows:
                f.write(json.dumps(row) + "\n")
    assert (~same).mean() <= max_fraction, (label, float((~same).mean()))
    return same


def params_pair(**kw):
    return N.Params(**kw), NLS.Params(**kw)


def check_records(out, ref_logs, p, rtol=1e-6):
    """Per-iteration NLSIteration records of problem p against the oracle's log."""
    rec = out.iterations.cpu().numpy()[p]
    for i, lg in enumerate(ref_logs[p]):
        exp = [lg.state, lg.lam, lg.errors_pre.f, lg.errors_pre.equality, lg.d_f, lg.d_eq, lg.penalty, lg.step_result, len(lg.steps)]
        np.testing.assert_allclose(rec[i][:9], exp, rtol=rtol, atol=1e-9, err_msg=f"problem {p} iteration {i}")
        for j, (alpha, e) in enumerate(lg.steps):
            np.testing.assert_allclose(rec[i][12 + 3 * j:15 + 3 * j], [alpha, e.f, e.equality], rtol=rtol, atol=1e-9)


def test_rosenbrock_and_lm():
    # nonlinear_test.cc:390-461
    oprob = N.Problem(2, P.rosenbrock_np)
    dprob = NLS.Problem(2, P.rosenbrock_torch, cost_rows=2)
    for kw in (dict(max_iterations=5, max_qp_iterations=1),
               dict(max_iterations=10, max_qp_iterations=1, absolute_first_derivative_tol=1e-12, max_line_search_iterations=0)):
        op, dp = params_pair(**kw)
        out, x, term, nit, rx, rterm, rnit, logs = run_both(oprob, dprob, op, dp, P.ROSENBROCK_GUESSES)
        assert np.all(term == NLS.SATISFIED_ABSOLUTE_TOL)
        np.testing.assert_allclose(x, np.ones_like(x), atol=1e-6)
        assert np.array_equal(term, rterm) and np.array_equal(nit, rnit)
        assert np.array_equal(out.NumQPIterations().cpu().numpy(), nit)       # ASSERT_EQ(NumQPIterations(), iterations.size())
        for p in range(len(x)):
            check_records(out, logs, p)


def test_inequality_constrained_rosenbrock():
    # nonlinear_test.cc:463-500
    cons = [(0, 1.0, -1.2), (1, -1.0, 0.5)]
    op, dp = params_pair(max_iterations=10, max_qp_iterations=10)
    out, x, term, nit, rx, rterm, rnit, logs = run_both(N.Problem(2, P.rosenbrock_np, inequality_constraints=cons),
                                                        NLS.Problem(2, P.rosenbrock_torch, cost_rows=2, inequality_constraints=cons),
                                                        op, dp, P.ROSENBROCK_CONSTRAINED_GUESSES)
    assert not np.any((term == NLS.MAX_ITERATIONS) | (term == NLS.MAX_LAMBDA))
    np.testing.assert_allclose(x, np.tile([1.2, 0.5], (len(x), 1)), atol=1e-6)
    assert np.array_equal(term, rterm) and np.array_equal(nit, rnit)
    np.testing.assert_allclose(x, rx, atol=1e-7)


def test_inequality_constrained_rosenbrock_6d():
    # nonlinear_test.cc:524-576
    op, dp = params_pair(max_iterations=30, max_qp_iterations=30, relative_exit_tol=1e-6, absolute_first_derivative_tol=5e-6,
                         termination_kkt_tolerance=1e-6, max_lambda=10.0)
    out, x, term, nit, rx, rterm, rnit, logs = run_both(
        N.Problem(6, P.rosenbrock6_np, inequality_constraints=P.ROSENBROCK6_CONSTRAINTS),
        NLS.Problem(6, P.rosenbrock6_torch, cost_rows=10, inequality_constraints=P.ROSENBROCK6_CONSTRAINTS), op, dp, P.ROSENBROCK6_GUESSES)
    assert np.all(NLS.TerminationStateIndicatesSatisfiedTol(torch.as_tensor(term)).numpy())
    np.testing.assert_allclose(x, np.tile(P.ROSENBROCK6_SOLUTION, (len(x), 1)), atol=1e-5)
    np.testing.assert_allclose(x, rx, atol=1e-5)


@pytest.mark.parametrize("quadrant", [False, True])
def test_himmelblau_grids(quadrant):
    # nonlinear_test.cc:597-720: 961 (resp. 576) starts as ONE batch
    cons = P.box(0.1, 5.0) if quadrant else P.box(-5.0, 5.0)
    guesses = P.himmelblau_quadrant_guesses() if quadrant else P.himmelblau_guesses()
    op, dp = params_pair(max_iterations=20, max_qp_iterations=10, relative_exit_tol=1e-12, absolute_first_derivative_tol=1e-8,
                         termination_kkt_tolerance=1e-6)
    out, x, term, nit, rx, rterm, rnit, logs = run_both(N.Problem(2, P.himmelblau_np, inequality_constraints=cons),
                                                        NLS.Problem(2, P.himmelblau_torch, cost_rows=2, inequality_constraints=cons),
                                                        op, dp, guesses)
    assert np.all(NLS.TerminationStateIndicatesSatisfiedTol(torch.as_tensor(term)).numpy())
    sols = np.array([(3.0, 2.0)] if quadrant else P.HIMMELBLAU_SOLUTIONS)
    dist = np.min(np.linalg.norm(x[:, None, :] - sols[None], axis=2), axis=1)
    assert dist.max() < 5e-5
    # problem-by-problem agreement with the oracle's run of the reference algorithm (knife-edge branch decisions may flip
    # with rounding in a few starts; those still have to reach an optimum, checked above)
    same = knife_edge_rule(f"himmelblau quadrant={quadrant}", N.Problem(2, P.himmelblau_np, inequality_constraints=cons), op, guesses,
                           term, nit, rterm, rnit, max_fraction=0.03)
    np.testing.assert_allclose(x[same], rx[same], atol=1e-6)


def test_sphere_with_nonlinear_equality_constraints():
    # nonlinear_test.cc:745-826 (m = 0, k = 2: the reference's null-space path)
    op, dp = params_pair(max_iterations=100, max_qp_iterations=1, relative_exit_tol=1e-12, absolute_first_derivative_tol=1e-9,
                         termination_kkt_tolerance=1e-6, lambda_initial=0.001)
    out, x, term, nit, rx, rterm, rnit, logs = run_both(N.Problem(6, P.sphere_np, equality=P.sphere_eq_np),
                                                        NLS.Problem(6, P.sphere_torch, cost_rows=6, equality=P.sphere_eq_torch, equality_rows=2),
                                                        op, dp, P.sphere_guesses(100))
    assert np.all(NLS.TerminationStateIndicatesSatisfiedTol(torch.as_tensor(term)).numpy())
    sols = np.array(P.SPHERE_SOLUTIONS)
    dist = np.min(np.linalg.norm(x[:, None, :] - sols[None], axis=2), axis=1)
    assert dist.max() < 5e-5
    assert int(out.NumFailedLineSearches().sum()) == 0
    # (relative_exit_tol = 1e-12 makes the reference's own exit test a comparison at rounding level once the iteration has converged)
    same = knife_edge_rule("sphere", N.Problem(6, P.sphere_np, equality=P.sphere_eq_np), op, P.sphere_guesses(100), term, nit, rterm, rnit)
    np.testing.assert_allclose(x[same], rx[same], atol=1e-6)


def test_nls_argument_errors_and_callback_failure():
    dprob = NLS.Problem(2, P.rosenbrock_torch, cost_rows=2)
    nls = NLS.ConstrainedNonlinearLeastSquares(dprob, batch=3)
    from mini_opt_amd import _lib as L
    with pytest.raises(L.MiniOptError):
        nls.Solve(NLS.Params(max_qp_iterations=0), T(np.zeros((3, 2))))           # CheckParams, nonlinear.cc:48-73
    with pytest.raises(L.MiniOptError):
        nls.Solve(NLS.Params(armijo_search_tau=1.0), T(np.zeros((3, 2))))
    bad = NLS.Problem(2, lambda x, w: (_ for _ in ()).throw(ValueError("boom")), cost_rows=2)
    with pytest.raises(ValueError):
        NLS.ConstrainedNonlinearLeastSquares(bad, batch=3).Solve(NLS.Params(), T(np.zeros((3, 2))))


def test_null_space_solver():
    """mo_nullspace_solve (row f4): the reference's two known-answer problems (qp_test.cc:576-707), a random batch against the
    oracle's QR / Cholesky restatement (qp.cc:679-729), and NOT_POSITIVE_DEFINITE on an indefinite reduced Hessian."""
    for kat in P.nullspace_kats():
        n, k = kat["G"].shape[0], kat["A_eq"].shape[0]
        prob = Q.BatchedQP(n=n, k=k, G=T(np.tril(kat["G"]).T[None]), c=T(kat["c"][None]), A_eq=T(kat["A_eq"].T[None]),
                           b_eq=T(kat["b_eq"][None]))
        s = Q.QPNullSpaceSolver()
        term = s.Solve(prob)
        assert int(term[0]) == Q.QPNullSpaceSolver.SUCCESS
        x = s.variables().cpu().numpy()[0]
        for idx, val, tol in kat["expected"]:
            assert abs(x[idx] - val) <= max(tol, 1e-13), (kat["name"], idx, x[idx], val)
    rng = np.random.default_rng(3)
    B, n, k, m_r = 64, 24, 5, 40
    J = rng.uniform(-1, 1, (B, m_r, n)); r = rng.uniform(-1, 1, (B, m_r))
    A = rng.uniform(-1, 1, (B, k, n)); b = rng.uniform(-1, 1, (B, k))
    G = np.einsum("bqi,bqj->bij", J, J); c = np.einsum("bqi,bq->bi", J, r)
    G[7] = -G[7]                                                   # negative definite -> NOT_POSITIVE_DEFINITE
    G[9, 0, 0] = G[9, 0, 0] - 1e3                                  # indefinite with a large negative direction
    prob = Q.BatchedQP(n=n, k=k, G=T(np.tril(G).transpose(0, 2, 1)), c=T(c), A_eq=T(A.transpose(0, 2, 1)), b_eq=T(b))
    s = Q.QPNullSpaceSolver()
    term = s.Solve(prob).cpu().numpy()
    x = s.variables().cpu().numpy()
    for p in range(B):
        ok, xr = N.null_space_solve(N.QPData(np.tril(G[p]), c[p], A[p], b[p], []))
        assert (term[p] == 0) == ok, p
        if ok:
            np.testing.assert_allclose(x[p], xr, rtol=1e-9, atol=1e-11)
            np.testing.assert_allclose(A[p] @ x[p] + b[p], 0, atol=1e-11)
        else:
            assert np.all(np.isnan(x[p]))
    assert term[7] == 1 and term[9] == 1 and term.sum() == 2
    # J-level input gives the same answers
    probJ = Q.BatchedQP(n=n, k=k, J=T(J), r=T(r), A_eq=T(A.transpose(0, 2, 1)), b_eq=T(b))
    sJ = Q.QPNullSpaceSolver()
    tJ = sJ.Solve(probJ).cpu().numpy()
    good = [p for p in range(B) if p not in (7, 9)]
    assert np.all(tJ == 0)
    np.testing.assert_allclose(sJ.variables().cpu().numpy()[good], x[good], rtol=1e-9, atol=1e-11)


def _nullspace(G, c, A, b, dt=torch.float64):
    """Batch-1 QPNullSpaceSolver call; returns (termination, x)."""
    n, k = G.shape[0], A.shape[0]
    s = Q.QPNullSpaceSolver()
    term = s.Solve(Q.BatchedQP(n=n, k=k, G=T(np.tril(G).T[None], dt), c=T(c[None], dt), A_eq=T(A.T[None], dt), b_eq=T(b[None], dt)))
    return int(term[0]), s.variables().double().cpu().numpy()[0]


def test_null_space_solver_on_singular_and_indefinite_hessians():
    """The inputs on which a KKT LDL^T in natural order cannot stand in for qp.cc:687-727: what decides SUCCESS is the reduced Hessian
    Q2^T G Q2 alone (LLT, qp.cc:711-714), not G."""
    # G = diag(1, 0) is singular, A = [0 1] fixes the direction G does not see: Z^T G Z = 1 > 0
    term, x = _nullspace(np.diag([1.0, 0.0]), np.array([-2.0, 5.0]), np.array([[0.0, 1.0]]), np.array([-3.0]))
    assert term == Q.QPNullSpaceSolver.SUCCESS
    np.testing.assert_allclose(x, [2.0, 3.0], atol=1e-14)
    # the leading entry of G is ZERO (an un-pivoted elimination stops at its first pivot); the null space is span{(1, -1, 0), e3}
    G = np.array([[0.0, 0.0, 0.0], [0.0, 2.0, 0.0], [0.0, 0.0, 1.0]])
    term, x = _nullspace(G, np.array([1.0, 1.0, 1.0]), np.array([[1.0, 1.0, 0.0]]), np.array([-1.0]))
    ok, xr = N.null_space_solve(N.QPData(np.tril(G), np.array([1.0, 1.0, 1.0]), np.array([[1.0, 1.0, 0.0]]), np.array([-1.0]), []))
    assert ok and term == Q.QPNullSpaceSolver.SUCCESS
    np.testing.assert_allclose(x, xr, atol=1e-14)
    # G indefinite (eigenvalue -1 along e1) but positive definite on null(A) = {x1 = 0}
    G = np.diag([-1.0, 2.0, 3.0])
    term, x = _nullspace(G, np.array([1.0, -4.0, 6.0]), np.array([[1.0, 0.0, 0.0]]), np.array([-0.5]))
    assert term == Q.QPNullSpaceSolver.SUCCESS
    np.testing.assert_allclose(x, [0.5, 2.0, -2.0], atol=1e-14)
    # ... and the same G with the equality on x2 instead leaves the negative direction in the null space
    term, x = _nullspace(G, np.array([1.0, -4.0, 6.0]), np.array([[0.0, 1.0, 0.0]]), np.array([-0.5]))
    assert term == Q.QPNullSpaceSolver.NOT_POSITIVE_DEFINITE and np.all(np.isnan(x))
    # a duplicated equality row: rank 1 of 2 (Eigen's QR.rank(), qp.cc:697); the consistent reading (oracle) succeeds
    G = np.diag([1.0, 2.0, 4.0])
    A = np.array([[1.0, 1.0, 1.0], [1.0, 1.0, 1.0]]); b = np.array([-3.0, -3.0])
    ok, xr = N.null_space_solve(N.QPData(np.tril(G), np.array([0.5, -1.0, 2.0]), A, b, []))
    term, x = _nullspace(G, np.array([0.5, -1.0, 2.0]), A, b)
    assert ok and term == Q.QPNullSpaceSolver.SUCCESS
    np.testing.assert_allclose(x, xr, atol=1e-13)
    np.testing.assert_allclose(A @ x + b, 0, atol=1e-13)


@pytest.mark.parametrize("n,k,m_r", [(64, 8, 40), (128, 12, 120), (33, 7, 20), (100, 30, 71), (6, 6, 3), (24, 5, 40)])
def test_null_space_solver_with_rank_deficient_cost(n, k, m_r):
    """J-level input with lambda = 0 and fewer residual rows than variables: G = J^T J is singular, the reduced Hessian is positive
    definite iff m_r + k >= n (general position).  Device vs the oracle's QR / Cholesky restatement, problem by problem."""
    rng = np.random.default_rng(n * 1000 + k)
    B = 9
    J = rng.uniform(-1, 1, (B, m_r, n)); r = rng.uniform(-1, 1, (B, m_r))
    A = rng.uniform(-1, 1, (B, k, n)); b = rng.uniform(-1, 1, (B, k))
    s = Q.QPNullSpaceSolver()
    term = s.Solve(Q.BatchedQP(n=n, k=k, J=T(J), r=T(r), lam=0.0, A_eq=T(A.transpose(0, 2, 1)), b_eq=T(b))).cpu().numpy()
    x = s.variables().cpu().numpy()
    for p in range(B):
        G = J[p].T @ J[p]
        ok, xr = N.null_space_solve(N.QPData(np.tril(G), J[p].T @ r[p], A[p], b[p], []))
        assert ok == (m_r + k >= n) and (term[p] == 0) == ok, (p, ok, term[p])
        if ok:
            np.testing.assert_allclose(x[p], xr, rtol=1e-8, atol=1e-9)
            np.testing.assert_allclose(A[p] @ x[p] + b[p], 0, atol=1e-10)


def test_device_residual_families_match_the_torch_definitions():
    """mo_residual_eval (row f2): values and dense Jacobians of the four families against the torch / numpy definitions of the
    reference's test residuals."""
    rng = np.random.default_rng(11)
    B = 33
    for fam, n, rows, ref in ((NLS.ROSENBROCK, 2, 2, P.rosenbrock_torch), (NLS.ROSENBROCK, 6, 10, P.rosenbrock6_torch),
                              (NLS.HIMMELBLAU, 2, 2, P.himmelblau_torch), (NLS.SPHERE, 6, 6, P.sphere_torch)):
        x = T(rng.uniform(-3, 3, (B, n)))
        r, J = NLS.DeviceFamily(fam, rows)(x, True)
        r0, J0 = ref(x, True)
        np.testing.assert_allclose(r.cpu().numpy(), r0.cpu().numpy(), rtol=1e-14, atol=1e-14)
        np.testing.assert_allclose(J.cpu().numpy(), J0.cpu().numpy(), rtol=1e-14, atol=1e-14)
    x = T(rng.uniform(-3, 3, (B, 6)))
    r, J = NLS.DeviceFamily(NLS.PRODUCT_PAIRS, 2, params=T(np.array([4.0, 9.0])))(x, True)
    r0, J0 = P.sphere_eq_torch(x, True)
    np.testing.assert_allclose(r.cpu().numpy(), r0.cpu().numpy(), rtol=1e-14, atol=1e-14)
    np.testing.assert_allclose(J.cpu().numpy(), J0.cpu().numpy(), rtol=1e-14, atol=1e-14)
    from mini_opt_amd import _lib as L
    with pytest.raises(L.MiniOptError):
        NLS.DeviceFamily(NLS.HIMMELBLAU, 2)(T(np.zeros((3, 5))), False)          # Himmelblau needs n = 2


def test_nls_with_device_residual_families():
    """The whole SQP loop with library residual kernels in the callback (no torch ops, no copies): same outcomes as with the
    torch-evaluated residuals, problem by problem."""
    kw = dict(max_iterations=20, max_qp_iterations=10, relative_exit_tol=1e-12, absolute_first_derivative_tol=1e-8,
              termination_kkt_tolerance=1e-6)
    g = T(np.array(P.himmelblau_guesses()))
    cons = P.box(-5.0, 5.0)
    a = NLS.ConstrainedNonlinearLeastSquares(NLS.Problem(2, P.himmelblau_torch, cost_rows=2, inequality_constraints=cons), batch=len(g))
    b = NLS.ConstrainedNonlinearLeastSquares(NLS.Problem(2, NLS.DeviceFamily(NLS.HIMMELBLAU, 2), cost_rows=2, inequality_constraints=cons), batch=len(g))
    oa, ob = a.Solve(NLS.Params(**kw), g), b.Solve(NLS.Params(**kw), g)
    assert torch.equal(oa.termination_state, ob.termination_state) and torch.equal(oa.num_iterations, ob.num_iterations)
    np.testing.assert_allclose(a.variables().cpu().numpy(), b.variables().cpu().numpy(), atol=1e-9)
    # sphere + product equalities: cost and equality stacks both from device kernels
    kw2 = dict(max_iterations=100, max_qp_iterations=1, relative_exit_tol=1e-12, absolute_first_derivative_tol=1e-9,
               termination_kkt_tolerance=1e-6, lambda_initial=0.001)
    g2 = T(np.array(P.sphere_guesses(64)))
    c = NLS.ConstrainedNonlinearLeastSquares(
        NLS.Problem(6, NLS.DeviceFamily(NLS.SPHERE, 6), cost_rows=6, equality=NLS.DeviceFamily(NLS.PRODUCT_PAIRS, 2, params=T(np.array([4.0, 9.0]))),
                    equality_rows=2), batch=64)
    oc = c.Solve(NLS.Params(**kw2), g2)
    assert bool(NLS.TerminationStateIndicatesSatisfiedTol(oc.termination_state).all())
    x = c.variables().cpu().numpy()
    dist = np.min(np.linalg.norm(x[:, None, :] - np.array(P.SPHERE_SOLUTIONS)[None], axis=2), axis=1)
    assert dist.max() < 5e-5


def test_nls_on_the_fused_solve_kernel_with_per_problem_lambda():
    """n = 32 with a row count the fused kernels accept: inside mo_nls_solve the QPs run on the fused Solve kernel with the
    per-problem lambda vector (mo_problem.lambda_vec).  Sphere cost, four product equalities, a box on eight variables; parity with
    the oracle's SQP loop, problem by problem."""
    n, k = 32, 4
    prods = np.array([4.0, 9.0, 1.0, 2.25])
    cons = P.box(-6.0, 6.0, nvars=8)

    def cost_np(x, want_J):
        return np.array(x, float), (np.eye(n) if want_J else None)

    def eq_np(x, want_J):
        r = np.array([x[2 * q] * x[2 * q + 1] - prods[q] for q in range(k)])
        J = None
        if want_J:
            J = np.zeros((k, n))
            for q in range(k):
                J[q, 2 * q], J[q, 2 * q + 1] = x[2 * q + 1], x[2 * q]
        return r, J

    kw = dict(max_iterations=60, max_qp_iterations=10, relative_exit_tol=1e-12, absolute_first_derivative_tol=1e-9,
              termination_kkt_tolerance=1e-6, lambda_initial=0.001)
    rng = np.random.default_rng(17)
    guesses = rng.uniform(-5.0, 5.0, (48, n))
    dprob = NLS.Problem(n, NLS.DeviceFamily(NLS.SPHERE, n), cost_rows=n, equality=NLS.DeviceFamily(NLS.PRODUCT_PAIRS, k, params=T(prods)),
                        equality_rows=k, inequality_constraints=cons)
    op, dp = params_pair(**kw)
    out, x, term, nit, rx, rterm, rnit, logs = run_both(N.Problem(n, cost_np, equality=eq_np, inequality_constraints=cons), dprob, op, dp, guesses)
    sat = NLS.TerminationStateIndicatesSatisfiedTol(torch.as_tensor(term)).numpy()
    assert sat.mean() > 0.5 and np.array_equal(sat, np.isin(rterm, [1, 2, 3]))        # some starts run out of iterations, as in the oracle
    # at an optimum the pairs satisfy x_{2q} x_{2q+1} = v_q with |x_{2q}| = |x_{2q+1}| and everything else is 0
    for q in range(k):
        np.testing.assert_allclose(x[sat, 2 * q] * x[sat, 2 * q + 1], prods[q], atol=1e-6)
        np.testing.assert_allclose(np.abs(x[sat, 2 * q]), np.sqrt(prods[q]), atol=5e-4)
    np.testing.assert_allclose(x[sat][:, 2 * k:], 0.0, atol=5e-5)
    same = knife_edge_rule("sphere with product pairs and boxes", N.Problem(n, cost_np, equality=eq_np, inequality_constraints=cons), op, guesses,
                           term, nit, rterm, rnit)
    np.testing.assert_allclose(x[same], rx[same], atol=1e-6)
    # lambda really differs from problem to problem along the way (the per-problem vector is exercised)
    lam = out.iterations.cpu().numpy()[:, :, 1]
    assert np.nanmax(np.nanstd(lam, axis=0)) > 0 or np.all(nit == nit[0])


def test_fp32_variants_of_the_support_entry_points():
    """mo_fill_qp / mo_qp_cost_derivative / mo_residual_eval / mo_nullspace_solve on fp32 plans (mo_nls_solve itself is fp64-only
    and says so)."""
    rng = np.random.default_rng(23)
    dt = torch.float32
    f = lambda a_: a_.astype(np.float32).astype(np.float64)
    B, n, k, m, m_r = 21, 10, 2, 5, 16
    J = f(rng.uniform(-1, 1, (B, m_r, n))); r = f(rng.uniform(-1, 1, (B, m_r)))
    A = f(rng.uniform(-1, 1, (B, k, n))); b = f(rng.uniform(-1, 1, (B, k)))
    cv = rng.integers(0, n, (B, m)).astype(np.int32); ca = rng.choice([-1.0, 1.0], (B, m)); cb = f(rng.uniform(-1, 1, (B, m)))
    x = f(rng.uniform(-2, 2, (B, n))); dx = f(rng.uniform(-1, 1, (B, n)))
    prob = Q.BatchedQP(n=n, k=k, m=m, J=T(J, dt), r=T(r, dt), lam=0.25, A_eq=T(A.transpose(0, 2, 1), dt), b_eq=T(b, dt),
                       cons_var=T(cv, torch.int32), cons_a=T(ca, dt), cons_b=T(cb, dt))
    G, c, cbs, err, status = NLS.fill_qp(prob, T(x, dt))
    assert torch.all(status == 0)
    Gref = np.einsum("bqi,bqj->bij", J, J) + 0.25 * np.eye(n)
    np.testing.assert_allclose(np.tril(G.double().cpu().numpy().transpose(0, 2, 1)), np.tril(Gref), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(cbs.double().cpu().numpy(), ca * np.take_along_axis(x, cv, 1) + cb, rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(err.double().cpu().numpy(), np.stack([0.5 * np.sum(r * r, 1), np.sum(np.abs(b), 1)], 1), rtol=1e-5)
    d1, q1 = NLS.qp_cost_derivative(prob, T(dx, dt))
    dref = np.stack([np.einsum("bi,bi->b", np.einsum("bqi,bq->bi", J, r), dx),
                     np.einsum("bk,bk->b", np.sign(b), np.einsum("bkn,bn->bk", A, dx))], 1)
    np.testing.assert_allclose(d1.double().cpu().numpy(), dref, rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(q1.double().cpu().numpy(), np.einsum("bi,bij,bj->b", dx, Gref, dx), rtol=1e-4)
    rr, JJ = NLS.DeviceFamily(NLS.ROSENBROCK, 10)(T(x[:, :6], dt), True)
    r0, J0 = P.rosenbrock6_torch(T(x[:, :6], dt), True)
    np.testing.assert_allclose(rr.cpu().numpy(), r0.cpu().numpy(), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(JJ.cpu().numpy(), J0.cpu().numpy(), rtol=1e-5, atol=1e-5)
    s = Q.QPNullSpaceSolver()
    term = s.Solve(Q.BatchedQP(n=n, k=k, J=T(J, dt), r=T(r, dt), lam=0.25, A_eq=T(A.transpose(0, 2, 1), dt), b_eq=T(b, dt)))
    assert torch.all(term == 0)
    xs = s.variables().double().cpu().numpy()
    for p in range(B):
        ok, xr = N.null_space_solve(N.QPData(np.tril(Gref[p]), np.einsum("qi,q->i", J[p], r[p]), A[p], b[p], []))
        assert ok
        np.testing.assert_allclose(xs[p], xr, rtol=2e-3, atol=2e-4)
    from mini_opt_amd import _lib as L
    with pytest.raises(L.MiniOptError):
        NLS.ConstrainedNonlinearLeastSquares(NLS.Problem(2, P.rosenbrock_torch, cost_rows=2), batch=2, dtype=torch.float32)


def test_nls_isolates_a_failing_problem_and_handles_degenerate_calls():
    """A NaN start makes that problem's QP fail (the reference would throw out of Solve): it ends with QP_FAILURE and its QP status, the
    neighbours are untouched.  max_iterations = 0 returns MAX_ITERATIONS without calling the residuals."""
    guesses = np.array(P.ROSENBROCK_GUESSES, dtype=float)
    guesses[3] = [np.nan, 1.0]
    calls = {"n": 0}

    def cost(x, want_J):
        calls["n"] += 1
        return P.rosenbrock_torch(x, want_J)

    nls = NLS.ConstrainedNonlinearLeastSquares(NLS.Problem(2, cost, cost_rows=2, inequality_constraints=[(0, 1.0, 50.0)]), batch=len(guesses))
    out = nls.Solve(NLS.Params(max_iterations=8, max_qp_iterations=10), T(guesses))
    term = out.termination_state.cpu().numpy(); st = out.status.cpu().numpy()
    assert term[3] == NLS.QP_FAILURE and st[3] != 0
    good = [i for i in range(len(guesses)) if i != 3]
    assert np.all(st[good] == 0) and np.all(np.isin(term[good], [NLS.SATISFIED_ABSOLUTE_TOL, NLS.SATISFIED_RELATIVE_TOL, NLS.SATISFIED_FIRST_ORDER_TOL]))
    np.testing.assert_allclose(nls.variables().cpu().numpy()[good], 1.0, atol=1e-5)
    calls["n"] = 0
    out0 = nls.Solve(NLS.Params(max_iterations=0), T(np.array(P.ROSENBROCK_GUESSES, dtype=float)))
    assert calls["n"] == 0 and torch.all(out0.termination_state == NLS.MAX_ITERATIONS) and torch.all(out0.num_iterations == 0)


# ---- kinematic-chain robots (row f2): nonlinear_test.cc:828-1136 on the device -------------------------------------------------------
def _chain_family(spec, rows):
    return NLS.DeviceFamily(NLS.ACTUATOR_CHAIN, len(rows), NLS.actuator_chain_params(spec["chains"], rows))


def _effector_xy(spec, x):
    return np.stack([P.chain_effector_np(spec, 0, v)[:2] for v in x])


def test_actuator_chain_family_matches_the_chain_restatement():
    """MO_RESIDUAL_ACTUATOR_CHAIN against oracle/chain_oracle.py (ComputeChain / ActuatorChain::Update restated, pinned by numerical
    derivatives): the rows of both robot problems, and a general chain with base rotations and every kind of active parameter."""
    rng = np.random.default_rng(5)
    general = dict(n=7, chains=[[P.chain_link((1.0, 0.5, 2.0), (1, 0, 1, 0, 1, 0), (0, 1, 2), rotation_xyz=(-0.5, 0.5, 0.3)),
                                 P.chain_link((0.5, 0.75, -0.5), (0, 1, 0, 0, 0, 0), (3,), rotation_xyz=(0.8, 0.5, 1.2)),
                                 P.chain_link((1.2, -0.5, 0.1), (0, 0, 0, 1, 0, 1), (4, 5), rotation_xyz=(1.5, -0.2, 0.0)),
                                 P.chain_link((0.1, -0.1, 0.2), (1, 1, 1, 0, 0, 0), (6, 0, 2), rotation_xyz=(0.2, -0.1, 0.3))]])
    general_rows = [P.chain_row(0.3, lin={1: -0.7, 6: 0.2}, terms=[(0, 1.0, -2.0, 0.5)]), P.chain_row(-1.0, terms=[(0, 0.0, 0.0, 1.0)]),
                    P.chain_row(0.0, lin={3: 1.5})]
    for spec, rows in ((P.TWO_ANGLE, P.TWO_ANGLE["cost_rows"] + P.TWO_ANGLE["eq_rows"]), (P.DUAL, P.DUAL["cost_rows"] + P.DUAL["eq_rows"]),
                       (general, general_rows)):
        B, n = 17, spec["n"]
        x = rng.uniform(-1.2, 1.2, (B, n))
        fam = _chain_family(spec, rows)
        r, J = fam(T(x), True)
        fn = P.chain_rows_np(spec, rows)
        for p in range(B):
            r0, J0 = fn(x[p], True)
            np.testing.assert_allclose(r[p].cpu().numpy(), r0, rtol=0, atol=1e-14)
            np.testing.assert_allclose(J[p].cpu().numpy(), J0, rtol=0, atol=1e-14)
        r_only, none = fam(T(x), False)
        assert none is None and torch.equal(r_only, r)


def _run_two_angle(stage, retraction):
    spec = P.TWO_ANGLE
    guesses = np.array(P.two_angle_guesses(stage))
    prm = dict(spec["params"])
    cons = []
    if stage == 2:
        prm["max_qp_iterations"] = 10
        cons = spec["inequalities_stage2"]
    prob = NLS.Problem(2, _chain_family(spec, spec["cost_rows"]), cost_rows=1, equality=_chain_family(spec, spec["eq_rows"]), equality_rows=1,
                       inequality_constraints=cons)
    nls = NLS.ConstrainedNonlinearLeastSquares(prob, batch=len(guesses), retraction=retraction)
    out = nls.Solve(NLS.Params(**prm), T(guesses))
    return spec, guesses, prm, cons, nls, out


@pytest.mark.parametrize("stage", [1, 2])
def test_two_angle_actuator_chain_on_device(stage):
    """TestTwoAngleActuatorChain (nonlinear_test.cc:828-964), every initial guess one problem of the batch, residuals from the device
    chain family, the reference's ModPi retraction built in (WRAP_PI): the reference's assertions (effector within 5e-5 / 1e-3 of the
    target, < 100 line-search steps) and the oracle's termination state and iteration count problem by problem."""
    spec, guesses, prm, cons, nls, out = _run_two_angle(stage, NLS.WRAP_PI)
    x = nls.variables().cpu().numpy()
    assert torch.all(out.status == 0)
    np.testing.assert_allclose(_effector_xy(spec, x), np.tile(spec["target_xy"], (len(x), 1)), atol=5e-5 if stage == 1 else 1e-3)
    assert int(out.NumLineSearchSteps().max()) < 100
    term, nit = out.termination_state.cpu().numpy(), out.num_iterations.cpu().numpy()
    cost, eq = P.chain_rows_np(spec, spec["cost_rows"]), P.chain_rows_np(spec, spec["eq_rows"])
    oprob = N.Problem(2, cost, equality=eq, inequality_constraints=cons)
    rterm, rnit, rx = [], [], []
    for p, g in enumerate(guesses):
        o = N.ConstrainedNonlinearLeastSquares(oprob, retraction=P.mod_pi_retraction_np)
        t, logs = o.solve(N.Params(**prm), g)
        rterm.append(t); rnit.append(len(logs)); rx.append(o.variables.copy())
    same = knife_edge_rule(f"two-angle chain stage {stage}", oprob, N.Params(**prm), guesses, term, nit, rterm, rnit,
                           retraction=P.mod_pi_retraction_np, max_fraction=0.03)
    np.testing.assert_allclose(x[same], np.array(rx)[same], atol=1e-6)


def test_two_angle_with_a_callback_retraction_matches_the_builtin():
    """The caller's own Retraction (nonlinear.hpp:127) through MO_RETRACT_CALLBACK: the same ModPi wrap written with torch ops gives the
    results of the built-in one bit for bit."""
    def wrap(x, dx, alpha):
        v = x + dx * alpha[:, None]
        return v - 2 * np.pi * torch.floor((v + np.pi) / (2 * np.pi))
    _, _, _, _, nls_a, out_a = _run_two_angle(1, NLS.WRAP_PI)
    _, _, _, _, nls_b, out_b = _run_two_angle(1, wrap)
    assert torch.equal(out_a.termination_state, out_b.termination_state) and torch.equal(out_a.num_iterations, out_b.num_iterations)
    np.testing.assert_allclose(nls_a.variables().cpu().numpy(), nls_b.variables().cpu().numpy(), rtol=0, atol=1e-13)


def test_dual_actuator_balancing_on_device():
    """TestDualActuatorBalancing (nonlinear_test.cc:966-1136) on the device: SATISFIED_ABSOLUTE_TOL from all three guesses, every
    residual's quadratic error below 1e-8, fewer than 36 line-search steps; same termination and iteration count as the oracle."""
    spec = P.DUAL
    guesses = np.array(spec["guesses"])
    prob = NLS.Problem(5, _chain_family(spec, spec["cost_rows"]), cost_rows=2, equality=_chain_family(spec, spec["eq_rows"]), equality_rows=2,
                       inequality_constraints=spec["inequalities"])
    nls = NLS.ConstrainedNonlinearLeastSquares(prob, batch=len(guesses), retraction=NLS.WRAP_PI)
    out = nls.Solve(NLS.Params(**spec["params"]), T(guesses))
    assert torch.all(out.termination_state == NLS.SATISFIED_ABSOLUTE_TOL) and torch.all(out.status == 0)
    x = nls.variables().cpu().numpy()
    cost, eq = P.chain_rows_np(spec, spec["cost_rows"]), P.chain_rows_np(spec, spec["eq_rows"])
    for p, g in enumerate(guesses):
        for fn in (cost, eq):
            r, _ = fn(x[p], False)
            assert np.all(0.5 * r * r <= 1e-8)
        o = N.ConstrainedNonlinearLeastSquares(N.Problem(5, cost, equality=eq, inequality_constraints=spec["inequalities"]),
                                               retraction=P.mod_pi_retraction_np)
        t, logs = o.solve(N.Params(**spec["params"]), g)
        assert t == NLS.SATISFIED_ABSOLUTE_TOL and len(logs) == int(out.num_iterations[p])
        # 5 angles, 4 conditions: the optimum is a one-parameter family, only the damped steps pick a point on it -- rounding moves it
        np.testing.assert_allclose(x[p], o.variables, atol=1e-4)
    assert int(out.NumLineSearchSteps().max()) < 36


def test_user_exit_callback_on_device():
    """SetUserExitCallback (nonlinear.hpp:157, nonlinear.cc:142-149): stopping the odd problems after their first iteration ends them
    with USER_CALLBACK after exactly one iteration (Gauss-Newton needs two on Rosenbrock); the even ones run on to their optimum; a
    callback that always proceeds changes nothing."""
    guesses = np.array(P.ROSENBROCK_GUESSES, dtype=float)
    B = len(guesses)
    nls = NLS.ConstrainedNonlinearLeastSquares(NLS.Problem(2, P.rosenbrock_torch, cost_rows=2), batch=B)
    odd = torch.arange(B, device="cuda:0") % 2 == 1
    seen = []

    def cb(iteration, outputs):
        seen.append(iteration)
        assert outputs.iterations.shape[0] == B
        return ~odd
    nls.SetUserExitCallback(cb)
    out = nls.Solve(NLS.Params(max_iterations=10, max_qp_iterations=1), T(guesses))
    term, nit = out.termination_state.cpu().numpy(), out.num_iterations.cpu().numpy()
    assert seen[:2] == [0, 1]
    assert np.all(term[1::2] == NLS.USER_CALLBACK) and np.all(nit[1::2] == 1)
    assert np.all(term[0::2] == NLS.SATISFIED_ABSOLUTE_TOL)
    np.testing.assert_allclose(nls.variables().cpu().numpy()[0::2], 1.0, atol=1e-6)
    nls.SetUserExitCallback(lambda it, o: True)
    out2 = nls.Solve(NLS.Params(max_iterations=10, max_qp_iterations=1), T(guesses))
    nls.SetUserExitCallback(None)
    out3 = nls.Solve(NLS.Params(max_iterations=10, max_qp_iterations=1), T(guesses))
    assert torch.equal(out2.termination_state, out3.termination_state) and torch.equal(out2.num_iterations, out3.num_iterations)


# ---- f3 leftovers: the reference's JSON form of NLSSolverOutputs, Problems made of Residuals -------------------------------------------
def test_json_serialization_of_batched_outputs():
    """mini_opt_amd.serialization writes the reference's schema (source/serialization.cc:32-136): keys, enum strings, one QP record per
    interior-point iteration; the numbers are the oracle's logs."""
    import json
    from mini_opt_amd import serialization as S
    guesses = np.array(P.ROSENBROCK_CONSTRAINED_GUESSES, dtype=float)
    cons = [(0, 1.0, -1.2), (1, -1.0, 0.5)]
    nls = NLS.ConstrainedNonlinearLeastSquares(NLS.Problem(2, P.rosenbrock_torch, cost_rows=2, inequality_constraints=cons), batch=len(guesses))
    prm = dict(max_iterations=10, max_qp_iterations=10)
    out = nls.Solve(NLS.Params(**prm), T(guesses), record_qp_iterations=True)
    for p, g in enumerate(guesses):
        doc = json.loads(S.dumps(out, p))
        assert set(doc) == {"termination_state", "iterations"} and doc["termination_state"] == S.NLS_TERMINATION[int(out.termination_state[p])]
        o = N.ConstrainedNonlinearLeastSquares(N.Problem(2, P.rosenbrock_np, inequality_constraints=cons))
        _, logs = o.solve(N.Params(**prm), g)
        assert len(doc["iterations"]) == len(logs) == int(out.num_iterations[p])
        for i, (it, log) in enumerate(zip(doc["iterations"], logs)):
            assert set(it) == {"iteration", "optimizer_state", "lambda", "errors_initial", "qp_outputs", "qp_eigenvalues",
                               "directional_derivatives", "penalty", "step_result", "line_search_steps"}
            assert it["iteration"] == i and it["qp_eigenvalues"] is None and it["optimizer_state"] in S.OPTIMIZER_STATE
            assert it["step_result"] == S.STEP_RESULT[log.step_result]
            np.testing.assert_allclose(it["errors_initial"]["f"], log.errors_pre.f, rtol=1e-6, atol=1e-12)
            np.testing.assert_allclose(it["directional_derivatives"]["d_f"], log.d_f, rtol=1e-5, atol=1e-9)
            assert len(it["line_search_steps"]) == len(log.steps)
            qp = it["qp_outputs"]
            assert set(qp) == {"termination_state", "iterations", "lagrange_multipliers"} and qp["lagrange_multipliers"] is None
            assert qp["termination_state"] in S.QP_TERMINATION and len(qp["iterations"]) == log.qp_iterations
            for q in qp["iterations"]:
                assert set(q) == {"kkt_initial", "kkt_final", "ip_outputs"} and set(q["kkt_final"]) == {"r_dual", "r_comp", "r_primal_eq", "r_primal_ineq"}
                assert set(q["ip_outputs"]) == {"mu", "alpha", "alpha_probe", "mu_affine"} and q["ip_outputs"]["mu_affine"] is None
    # equality-only problems: qp_outputs is the null-space solver's state (serialization.cc:89-101)
    spec = P.TWO_ANGLE
    prob = NLS.Problem(2, _chain_family(spec, spec["cost_rows"]), cost_rows=1, equality=_chain_family(spec, spec["eq_rows"]), equality_rows=1)
    cn = NLS.ConstrainedNonlinearLeastSquares(prob, batch=2, retraction=NLS.WRAP_PI)
    o2 = cn.Solve(NLS.Params(**spec["params"]), T(np.array([[0.3, 0.2], [1.0, -0.5]])))
    doc = S.nls_outputs_to_json(o2, 1)
    assert doc["iterations"] and all(it["qp_outputs"] == "SUCCESS" for it in doc["iterations"])
    json.dumps(doc)


def test_problem_of_residuals_matches_the_dense_stack():
    """Problem.FromResiduals / MakeResidual (residual.hpp:119-143, nonlinear.hpp:33-52): per-residual functors on their own parameters,
    Jacobians scattered by the index lists -- the sphere with its two product equalities (nonlinear_test.cc:722-826) gives exactly the
    results of the hand-stacked version."""
    def sphere(x, want_J):
        return x.clone(), (torch.eye(6, dtype=x.dtype, device=x.device).expand(x.shape[0], 6, 6).contiguous() if want_J else None)

    def product(target):
        def fn(x, want_J):
            r = (x[:, 0] * x[:, 1] - target).unsqueeze(1)
            J = torch.stack([x[:, 1], x[:, 0]], dim=1).unsqueeze(1) if want_J else None
            return r, J
        return fn
    guesses = np.array(P.sphere_guesses(12))
    prm = NLS.Params(max_iterations=100, max_qp_iterations=1, relative_exit_tol=1e-12, absolute_first_derivative_tol=1e-9,
                     termination_kkt_tolerance=1e-6, lambda_initial=0.001)
    a = NLS.ConstrainedNonlinearLeastSquares(NLS.Problem.FromResiduals(
        6, [NLS.MakeResidual(range(6), sphere, 6)], [NLS.MakeResidual((0, 1), product(4.0), 1), NLS.MakeResidual((2, 3), product(9.0), 1)]), batch=12)
    b = NLS.ConstrainedNonlinearLeastSquares(NLS.Problem(6, P.sphere_torch, cost_rows=6, equality=P.sphere_eq_torch, equality_rows=2), batch=12)
    oa, ob = a.Solve(prm, T(guesses)), b.Solve(prm, T(guesses))
    assert torch.equal(oa.termination_state, ob.termination_state) and torch.equal(oa.num_iterations, ob.num_iterations)
    assert torch.equal(a.variables(), b.variables())
    res = NLS.MakeResidual((2, 3), product(9.0), 1)
    assert res.Dimension() == 1
    np.testing.assert_allclose(res.QuadraticError(T(guesses)).cpu().numpy(), 0.5 * (guesses[:, 2] * guesses[:, 3] - 9.0) ** 2, rtol=1e-14)


@pytest.mark.parametrize("n,k,m_r", [(150, 6, 170), (186, 4, 192), (144, 8, 150)])
def test_null_space_solver_fp32_with_more_than_144_variables(n, k, m_r):
    """fp32 plans reach n = 145 ... 190 (n + k <= 192 and the LDS fits): the J^T J register tiling of the null-space kernel must cover
    every column there (16 x 12 blocks beyond 144; with 16 x 9 the rows and columns >= 144 of J^T J were never accumulated and the
    solver returned a wrong x with SUCCESS).  Referee: the oracle's QR / Cholesky restatement in fp64 on the fp32-rounded inputs."""
    rng = np.random.default_rng(n + k)
    B = 5
    f = lambda a: a.astype(np.float32).astype(np.float64)
    J = f(rng.uniform(-1, 1, (B, m_r, n))); r = f(rng.uniform(-1, 1, (B, m_r)))
    A = f(rng.uniform(-1, 1, (B, k, n))); b = f(rng.uniform(-1, 1, (B, k)))
    lam = float(np.float32(0.25))
    dt = torch.float32
    s = Q.QPNullSpaceSolver()
    term = s.Solve(Q.BatchedQP(n=n, k=k, J=T(J, dt), r=T(r, dt), lam=lam, A_eq=T(A.transpose(0, 2, 1), dt), b_eq=T(b, dt)))
    assert torch.all(term == Q.QPNullSpaceSolver.SUCCESS)
    x = s.variables().double().cpu().numpy()
    for p in range(B):
        G = J[p].T @ J[p] + lam * np.eye(n)
        ok, xr = N.null_space_solve(N.QPData(np.tril(G), J[p].T @ r[p], A[p], b[p], []))
        assert ok
        assert np.abs(x[p] - xr).max() <= 2e-3 * max(1.0, np.abs(xr).max()), (p, np.abs(x[p] - xr).max())
        assert np.abs(A[p] @ x[p] + b[p]).max() <= 1e-3
