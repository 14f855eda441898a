"""The NLS oracle (oracle/nls_oracle.py) replays the reference's own NLS tests (nonlinear_test.cc:390-826): termination class
and optimum of every named problem, plus the polynomial line-search helpers' known answers (nonlinear_test.cc:185-272)."""
import numpy as np
import pytest

from oracle import nls_oracle as N
from tests import nls_problems as P

SATISFIED = (N.SATISFIED_ABSOLUTE_TOL, N.SATISFIED_RELATIVE_TOL, N.SATISFIED_FIRST_ORDER_TOL)


def test_quadratic_and_cubic_helpers():
    # nonlinear_test.cc:185-209
    alpha_0, phi_0, phi_prime_0, phi_alpha_0 = 0.8, 2.0, -1.2, 2.2
    sol = N.quadratic_approx_minimum(phi_0, phi_prime_0, alpha_0, phi_alpha_0)
    a = (phi_alpha_0 - phi_0 - alpha_0 * phi_prime_0) / alpha_0 ** 2
    assert abs(-phi_prime_0 / (2 * a) - sol) < 1e-12
    # cubic through four conditions reproduces its own coefficients (nonlinear_test.cc:211-272)
    ca, cb, c1, c0 = 0.7, -1.3, -0.4, 2.0
    phi = lambda t: ca * t ** 3 + cb * t ** 2 + c1 * t + c0
    ab = N.cubic_approx_coeffs(c0, c1, 0.9, phi(0.9), 0.35, phi(0.35))
    np.testing.assert_allclose(ab, [ca, cb], rtol=1e-10)
    t = N.cubic_approx_minimum(c1, ab)
    assert abs(3 * ca * t * t + 2 * cb * t + c1) < 1e-10 and 6 * ca * t + 2 * cb > 0


def test_cost_derivative_matches_numerical():
    # nonlinear_test.cc:109-183: d/dalpha [0.5 |h|^2 + penalty |c|_1] at alpha = 0 along dx
    rng = np.random.default_rng(0)
    x = np.array([1.1, -0.4]); dx = np.array([0.3, -0.7])
    prob = N.Problem(2, P.rosenbrock_np, equality=lambda v, w: (np.array([v[0] * v[1] - 0.3]), np.array([[v[1], v[0]]]) if w else None))
    qp, _ = N.linearize_and_fill_qp(x, 0.0, prob)
    d_f, d_eq = N.compute_qp_cost_derivative(qp, dx)
    h = 1e-6
    tot = lambda al: N.evaluate_nonlinear_errors(prob, x + al * dx).total(0.334)
    num = (tot(h) - tot(-h)) / (2 * h)
    assert abs(num - (d_f + 0.334 * d_eq)) < 1e-6


@pytest.mark.parametrize("guess", P.ROSENBROCK_GUESSES)
def test_rosenbrock(guess):
    nls = N.ConstrainedNonlinearLeastSquares(N.Problem(2, P.rosenbrock_np))
    term, logs = nls.solve(N.Params(max_iterations=5, max_qp_iterations=1), guess)
    assert term == N.SATISFIED_ABSOLUTE_TOL
    np.testing.assert_allclose(nls.variables, [1, 1], atol=1e-6)


@pytest.mark.parametrize("guess", P.ROSENBROCK_GUESSES)
def test_rosenbrock_lm(guess):
    nls = N.ConstrainedNonlinearLeastSquares(N.Problem(2, P.rosenbrock_np))
    term, logs = nls.solve(N.Params(max_iterations=10, max_qp_iterations=1, absolute_first_derivative_tol=1e-12,
                                    max_line_search_iterations=0), guess)
    assert term == N.SATISFIED_ABSOLUTE_TOL
    np.testing.assert_allclose(nls.variables, [1, 1], atol=1e-6)


@pytest.mark.parametrize("guess", P.ROSENBROCK_CONSTRAINED_GUESSES)
def test_inequality_constrained_rosenbrock(guess):
    prob = N.Problem(2, P.rosenbrock_np, inequality_constraints=[(0, 1.0, -1.2), (1, -1.0, 0.5)])
    nls = N.ConstrainedNonlinearLeastSquares(prob)
    term, logs = nls.solve(N.Params(max_iterations=10, max_qp_iterations=10), guess)
    assert term not in (N.MAX_ITERATIONS, N.MAX_LAMBDA)
    np.testing.assert_allclose(nls.variables, [1.2, 0.5], atol=1e-6)


@pytest.mark.parametrize("guess", P.ROSENBROCK6_GUESSES)
def test_inequality_constrained_rosenbrock_6d(guess):
    prob = N.Problem(6, P.rosenbrock6_np, inequality_constraints=P.ROSENBROCK6_CONSTRAINTS)
    nls = N.ConstrainedNonlinearLeastSquares(prob)
    term, logs = nls.solve(N.Params(max_iterations=30, max_qp_iterations=30, relative_exit_tol=1e-6,
                                    absolute_first_derivative_tol=5e-6, termination_kkt_tolerance=1e-6, max_lambda=10.0), guess)
    assert term in SATISFIED
    np.testing.assert_allclose(nls.variables, P.ROSENBROCK6_SOLUTION, atol=1e-5)


def test_himmelblau_grid():
    prob = N.Problem(2, P.himmelblau_np, inequality_constraints=P.box(-5.0, 5.0))
    for sol in P.HIMMELBLAU_SOLUTIONS:
        assert N.evaluate_nonlinear_errors(prob, np.array(sol)).total(1.0) < 1e-6
    params = N.Params(max_iterations=20, max_qp_iterations=10, relative_exit_tol=1e-12, absolute_first_derivative_tol=1e-8,
                      termination_kkt_tolerance=1e-6)
    nls = N.ConstrainedNonlinearLeastSquares(prob)
    for guess in P.himmelblau_guesses():
        term, logs = nls.solve(params, guess)
        assert term in SATISFIED, (guess, term)
        best = min(P.HIMMELBLAU_SOLUTIONS, key=lambda s: np.linalg.norm(np.array(s) - nls.variables))
        np.testing.assert_allclose(nls.variables, best, atol=5e-5, err_msg=str(guess))


def test_himmelblau_quadrant():
    prob = N.Problem(2, P.himmelblau_np, inequality_constraints=P.box(0.1, 5.0))
    params = N.Params(max_iterations=20, max_qp_iterations=10, relative_exit_tol=1e-12, absolute_first_derivative_tol=1e-8,
                      termination_kkt_tolerance=1e-6)
    nls = N.ConstrainedNonlinearLeastSquares(prob)
    for guess in P.himmelblau_quadrant_guesses():
        term, logs = nls.solve(params, guess)
        assert term in SATISFIED, (guess, term)
        np.testing.assert_allclose(nls.variables, [3.0, 2.0], atol=5e-5, err_msg=str(guess))


def test_sphere_with_nonlinear_equality_constraints():
    prob = N.Problem(6, P.sphere_np, equality=P.sphere_eq_np)
    params = N.Params(max_iterations=100, max_qp_iterations=1, relative_exit_tol=1e-12, absolute_first_derivative_tol=1e-9,
                      termination_kkt_tolerance=1e-6, lambda_initial=0.001)
    nls = N.ConstrainedNonlinearLeastSquares(prob)
    for guess in P.sphere_guesses(40):
        term, logs = nls.solve(params, guess)
        assert term in SATISFIED, (guess, term)
        best = min(P.SPHERE_SOLUTIONS, key=lambda s: np.linalg.norm(np.array(s) - nls.variables))
        np.testing.assert_allclose(nls.variables, best, atol=5e-5, err_msg=str(guess))
        assert all(l.step_result not in (N.STEP_MAX_ITERATIONS, N.STEP_POSITIVE_DERIVATIVE) for l in logs)   # no failed line searches


# ---- QPNullSpaceSolver known answers, qp_test.cc:576-707 (problem data restated in tests/nls_problems.py)
def test_null_space_solver_kats():
    for kat in P.nullspace_kats():
        ok, x = N.null_space_solve(N.QPData(np.tril(kat["G"]), kat["c"], kat["A_eq"], kat["b_eq"], []))
        assert ok
        for idx, val, tol in kat["expected"]:
            assert abs(x[idx] - val) <= tol, (kat["name"], idx, x[idx], val)
    # an indefinite reduced Hessian is reported, qp.cc:709-713
    G = np.diag([1.0, -2.0, 3.0])
    ok, _ = N.null_space_solve(N.QPData(G, np.zeros(3), np.array([[1.0, 0.0, 0.0]]), np.array([0.5]), []))
    assert not ok


# ---- kinematic-chain robots, nonlinear_test.cc:828-1136 (problem data in tests/nls_problems.py, chains in oracle/chain_oracle.py)
def test_chain_residual_derivatives():
    """TestResidualFunctionDerivative on the chain residuals (nonlinear_test.cc:861-862, 1001, 1017, 1029, 1068-1069)."""
    for spec, rows, x in ((P.TWO_ANGLE, P.TWO_ANGLE["cost_rows"], [-0.5, 0.4]), (P.TWO_ANGLE, P.TWO_ANGLE["eq_rows"], [0.3, -0.6]),
                          (P.DUAL, P.DUAL["cost_rows"], [0.22, -0.3, 0.45, 0.6, -0.1]), (P.DUAL, P.DUAL["eq_rows"], [0.4, 0.2221, -0.8, -0.4, 0.5])):
        fn = P.chain_rows_np(spec, rows)
        x = np.array(x)
        _, J = fn(x, True)
        h = 1e-6
        Jn = np.stack([(fn(x + h * e, False)[0] - fn(x - h * e, False)[0]) / (2 * h) for e in np.eye(len(x))], axis=1)
        np.testing.assert_allclose(J, Jn, atol=1e-8)


def _steps(logs):
    return sum(len(l.steps) for l in logs)


def test_two_angle_actuator_chain():
    """TestTwoAngleActuatorChain, nonlinear_test.cc:828-964: every initial guess of both grids reaches the target effector position
    (5e-5 with the equality alone -- the null-space QP path --, 1e-3 and < 100 line-search steps with the box on angle 1)."""
    spec = P.TWO_ANGLE
    prm = dict(spec["params"])
    prob = N.Problem(2, P.chain_rows_np(spec, spec["cost_rows"]), equality=P.chain_rows_np(spec, spec["eq_rows"]))
    for guess in P.two_angle_guesses(1):
        nls = N.ConstrainedNonlinearLeastSquares(prob, retraction=P.mod_pi_retraction_np)
        nls.solve(N.Params(**prm), guess)
        np.testing.assert_allclose(P.chain_effector_np(spec, 0, nls.variables)[:2], spec["target_xy"], atol=5e-5, err_msg=str(guess))
    prob2 = N.Problem(2, prob.cost, equality=prob.equality, inequality_constraints=spec["inequalities_stage2"])
    prm["max_qp_iterations"] = 10                                                   # :928
    for guess in P.two_angle_guesses(2):
        nls = N.ConstrainedNonlinearLeastSquares(prob2, retraction=P.mod_pi_retraction_np)
        _, logs = nls.solve(N.Params(**prm), guess)
        np.testing.assert_allclose(P.chain_effector_np(spec, 0, nls.variables)[:2], spec["target_xy"], atol=1e-3, err_msg=str(guess))
        assert _steps(logs) < 100


def test_dual_actuator_balancing():
    """TestDualActuatorBalancing, nonlinear_test.cc:966-1136: SATISFIED_ABSOLUTE_TOL from all three guesses, every residual's quadratic
    error below 1e-8, fewer than 36 line-search steps."""
    spec = P.DUAL
    cost, eq = P.chain_rows_np(spec, spec["cost_rows"]), P.chain_rows_np(spec, spec["eq_rows"])
    prob = N.Problem(5, cost, equality=eq, inequality_constraints=spec["inequalities"])
    for guess in spec["guesses"]:
        nls = N.ConstrainedNonlinearLeastSquares(prob, retraction=P.mod_pi_retraction_np)
        term, logs = nls.solve(N.Params(**spec["params"]), guess)
        assert term == N.SATISFIED_ABSOLUTE_TOL, (guess, term)
        for fn in (cost, eq):
            r, _ = fn(nls.variables, False)
            assert np.all(0.5 * r * r <= 1e-8), (guess, r)                          # Residual::QuadraticError per residual, :1123-1128
        assert _steps(logs) < 36


def test_user_exit_callback():
    """SetUserExitCallback (nonlinear.hpp:157, nonlinear.cc:142-149): returning false after the second iteration ends the solve with
    USER_CALLBACK; a callback that always proceeds changes nothing."""
    seen = []
    nls = N.ConstrainedNonlinearLeastSquares(N.Problem(2, P.rosenbrock_np), user_exit_callback=lambda log: (seen.append(log), len(seen) < 2)[1])
    term, logs = nls.solve(N.Params(max_iterations=10, max_qp_iterations=1), (-5.0, -3.0))
    assert term == N.USER_CALLBACK and len(logs) == 2
    nls2 = N.ConstrainedNonlinearLeastSquares(N.Problem(2, P.rosenbrock_np), user_exit_callback=lambda log: True)
    term2, _ = nls2.solve(N.Params(max_iterations=5, max_qp_iterations=1), (-5.0, -3.0))
    assert term2 == N.SATISFIED_ABSOLUTE_TOL
