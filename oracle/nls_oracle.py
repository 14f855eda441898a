"""CPU restatement of mini_opt's SQP outer loop (source/nonlinear.cc) -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this; the product (mini_opt_amd) never does.
Plain numpy + Python loops (the problems it is used on have n <= 8), with the interior-point QP solved by the C oracle
(oracle/kkt_oracle.c, orc_solve).  Every function cites the reference lines it follows.

Pin: the reference's own NLS tests (nonlinear_test.cc:390-826) assert termination classes and optima of named problems;
tests/test_oracle_nls.py replays those problems through this restatement.  Eigen is absent from the image, so the
null-space QP path (qp.cc:679-729) is restated with scipy's pivoted QR / Cholesky.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Callable, List, Optional, Sequence, Tuple

import numpy as np
import scipy.linalg

from . import oracle as orc

# NLSTerminationState
(MAX_ITERATIONS, SATISFIED_ABSOLUTE_TOL, SATISFIED_RELATIVE_TOL, SATISFIED_FIRST_ORDER_TOL, MAX_LAMBDA, QP_INDEFINITE,
 USER_CALLBACK) = range(7)                                 # structs.hpp:233-248
# StepSizeSelectionResult, structs.hpp:215-228
(STEP_SUCCESS, STEP_MAX_ITERATIONS, STEP_FIRST_ORDER_SATISFIED, STEP_POSITIVE_DERIVATIVE, STEP_FAILURE_NON_FINITE_COST,
 STEP_FAILURE_INVALID_ALPHA) = range(6)
NOMINAL, ATTEMPTING_RESTORE_LM = 0, 1                      # OptimizerState
ARMIJO_BACKTRACK, POLYNOMIAL_APPROXIMATION = 0, 1          # LineSearchStrategy, structs.hpp:148-153


@dataclass
class Errors:                                              # structs.hpp:169-186
    f: float = 0.0
    equality: float = 0.0

    def total(self, penalty):
        return self.f + penalty * self.equality

    def linf(self):
        return max(self.f, self.equality)

    def invalid(self):
        return not (math.isfinite(self.f) and math.isfinite(self.equality))


@dataclass
class Params:                                              # nonlinear.hpp:64-124 (same names, same defaults)
    max_iterations: int = 10
    max_qp_iterations: int = 10
    termination_kkt_tolerance: float = 1.0e-6
    absolute_exit_tol: float = 1.0e-12
    relative_exit_tol: float = 1.0e-5
    absolute_first_derivative_tol: float = 1.0e-6
    max_line_search_iterations: int = 2
    line_search_strategy: int = POLYNOMIAL_APPROXIMATION
    armijo_search_tau: float = 0.8
    equality_penalty_initial: float = 1.0
    equality_penalty_scale_factor: float = 1.01
    equality_penalty_rho: float = 0.1
    lambda_initial: float = 0.0
    lambda_failure_init: float = 1.0e-2
    lambda_decrease_on_success: float = 0.1
    lambda_decrease_on_restore: float = 0.8
    max_lambda: float = 1.0
    min_lambda: float = 0.0


ResidualFn = Callable[[np.ndarray, bool], Tuple[np.ndarray, Optional[np.ndarray]]]


@dataclass
class Problem:
    """mini_opt::Problem (nonlinear.hpp:33-52) with dense residual stacks: cost(x, want_J) -> (r [m_r], J [m_r, n] | None),
    equality(x, want_J) -> (r_eq [k], J_eq [k, n] | None), inequality_constraints = [(variable, a, b), ...]."""
    dimension: int
    cost: ResidualFn
    equality: Optional[ResidualFn] = None
    inequality_constraints: Sequence[Tuple[int, float, float]] = field(default_factory=list)


@dataclass
class QPData:
    G: np.ndarray
    c: np.ndarray
    A_eq: np.ndarray
    b_eq: np.ndarray
    constraints: List[Tuple[int, float, float]]


def linearize_and_fill_qp(x, lam, problem: Problem) -> Tuple[QPData, Errors]:
    """nonlinear.cc:170-214."""
    n = problem.dimension
    r, J = problem.cost(x, True)
    G = np.tril(J.T @ J)                                   # UpdateHessian, residual.hpp:206-224 (lower triangle only)
    c = J.T @ r
    e = Errors(f=0.5 * float(r @ r))
    if lam > 0:
        G = G + lam * np.eye(n)                            # :187-189
    if problem.equality is not None:
        b_eq, A_eq = problem.equality(x, True)             # UpdateJacobian, :192-206
        e.equality = float(np.sum(np.abs(b_eq)))           # :203
    else:
        A_eq, b_eq = np.zeros((0, n)), np.zeros(0)
    cons = [(v, a, a * x[v] + b) for (v, a, b) in problem.inequality_constraints]   # ShiftTo, :209-212, qp.hpp:57-65
    return QPData(G, c, np.atleast_2d(A_eq), np.asarray(b_eq, float), cons), e


def evaluate_nonlinear_errors(problem: Problem, x) -> Errors:
    """nonlinear.cc:279-293."""
    r, _ = problem.cost(x, False)
    e = Errors(f=0.5 * float(r @ r))
    if problem.equality is not None:
        r_eq, _ = problem.equality(x, False)
        e.equality = float(np.sum(np.abs(r_eq)))
    return e


def compute_qp_cost_derivative(qp: QPData, dx) -> Tuple[float, float]:
    """nonlinear.cc:452-483: (d_f, d_equality)."""
    d_f = float(qp.c @ dx)
    d_eq = 0.0
    for i in range(qp.A_eq.shape[0]):
        d_eq += float(np.sign(qp.b_eq[i])) * float(qp.A_eq[i] @ dx)
    return d_f, d_eq


def select_penalty(qp: QPData, dx, lagrange_linf: Optional[float], rho: float) -> float:
    """nonlinear.cc:485-500."""
    if lagrange_linf is not None:
        return lagrange_linf
    l1_eq = max(float(np.sum(np.abs(qp.b_eq))), np.finfo(float).eps)
    Gs = qp.G + np.tril(qp.G, -1).T
    quad = float(qp.c @ dx) + 0.5 * max(0.0, float(dx @ (Gs @ dx)))
    return quad / ((1 - rho) * l1_eq)


def quadratic_approx_minimum(phi_0, phi_prime_0, alpha_0, phi_alpha_0):
    """nonlinear.cc:524-531."""
    numerator = phi_alpha_0 - phi_prime_0 * alpha_0 - phi_0
    if phi_prime_0 > 0 or numerator <= 0:
        return None
    return -phi_prime_0 * alpha_0 * alpha_0 / (2.0 * numerator)


def cubic_approx_coeffs(phi_0, phi_prime_0, alpha_0, phi_alpha_0, alpha_1, phi_alpha_1):
    """nonlinear.cc:558-574."""
    A = np.array([[alpha_0 ** 3, alpha_0 ** 2], [alpha_1 ** 3, alpha_1 ** 2]])
    rhs = np.array([phi_alpha_0 - phi_0 - phi_prime_0 * alpha_0, phi_alpha_1 - phi_0 - phi_prime_0 * alpha_1])
    det = A[0, 0] * A[1, 1] - A[0, 1] * A[1, 0]           # Matrix2d::inverse() is the adjugate formula
    inv = np.array([[A[1, 1], -A[0, 1]], [-A[1, 0], A[0, 0]]]) / det
    return inv @ rhs


def cubic_approx_minimum(phi_prime_0, ab):
    """nonlinear.cc:592-603."""
    a, b = float(ab[0]), float(ab[1])
    arg_sqrt = b * b - 3 * a * phi_prime_0
    if a == 0.0 or arg_sqrt < -1.0e-12:
        return None
    return (-b + math.sqrt(max(arg_sqrt, 0.0))) / (3 * a)


def null_space_solve(qp: QPData):
    """QPNullSpaceSolver::Solve, qp.cc:679-729.  Returns (ok, x)."""
    A = qp.A_eq
    k, n = A.shape
    Q, R, P = scipy.linalg.qr(A.T, mode="full", pivoting=True)
    diag = np.abs(np.diag(R[:k, :k]))
    # Eigen's default threshold (ColPivHouseholderQR::threshold(): epsilon * diagonalSize(), applied to the largest pivot)
    rank = int(np.sum(diag > diag.max() * min(n, k) * np.finfo(float).eps)) if k else 0
    Q1, Q2 = Q[:, :rank], Q[:, rank:]
    rhs = (-qp.b_eq)[P]
    u = Q1 @ scipy.linalg.solve_triangular(R[:rank, :rank].T, rhs[:rank], lower=True)
    Gs = qp.G + np.tril(qp.G, -1).T
    Gr = Q2.T @ Gs @ Q2
    try:
        L = np.linalg.cholesky(Gr)
    except np.linalg.LinAlgError:
        return False, np.zeros(n)
    y = -(Q2.T @ (qp.c + Gs @ u))
    y = scipy.linalg.cho_solve((L, True), y)
    return True, u + Q2 @ y


@dataclass
class IterationLog:
    lam: float
    errors_pre: Errors
    d_f: float
    d_eq: float
    penalty: float
    step_result: int
    steps: List[Tuple[float, Errors]]
    qp_iterations: int
    state: int


class ConstrainedNonlinearLeastSquares:
    """nonlinear.cc:20-158.  `retraction(x, dx, alpha) -> candidate` is the reference's custom Retraction (nonlinear.hpp:127,
    nonlinear.cc:160-168; default x + alpha dx); `user_exit_callback(log) -> bool` is SetUserExitCallback (nonlinear.hpp:157,
    nonlinear.cc:142-149): returning False ends the solve with USER_CALLBACK unless the iteration terminates anyway."""

    def __init__(self, problem: Problem, retraction=None, user_exit_callback=None, track_margins=False):
        self.p = problem
        self.variables = np.zeros(problem.dimension)
        self.retraction = retraction
        self.user_exit_callback = user_exit_callback
        # Decision margins (oracle/margins.py has the rule they serve): with track_margins every branch of the outer loop logs
        # (outer iteration, name, relative distance of the deciding quantity from its threshold) into self.margins; the inner
        # interior-point solves log the smallest margin of their own run under the name "qp".
        self.track_margins = track_margins
        self.margins: List[Tuple[int, str, float]] = []
        self._iter = 0

    def _note(self, name, margin):
        if self.track_margins:
            self.margins.append((self._iter, name, float(margin)))

    def compute_step_direction(self, qp: QPData, params: Params):
        """nonlinear.cc:216-258.  Returns (dx, lagrange_linf | None, indefinite, n_qp_iterations)."""
        n = self.p.dimension
        m = len(qp.constraints)
        k = qp.A_eq.shape[0]
        if m == 0 and k > 0:                               # nonlinear.cc:83-86: the null-space solver
            ok, x = null_space_solve(qp)
            return (x if ok else np.zeros(n)), None, (not ok), 0
        oq = orc.QP(G=qp.G, c=qp.c, A_eq=qp.A_eq if k else None, b_eq=qp.b_eq if k else None,
                    cons_var=np.array([c[0] for c in qp.constraints], np.int32),
                    cons_a=np.array([c[1] for c in qp.constraints], float),
                    cons_b=np.array([c[2] for c in qp.constraints], float))
        s = orc.Solver(oq)
        guess = orc.GUESS_SOLVE_EQUALITY_CONSTRAINED if k > 0 else orc.GUESS_NAIVE
        qp_kw = dict(max_iterations=params.max_qp_iterations, termination_kkt_tol=params.termination_kkt_tolerance,
                     initial_mu=1.0, sigma=0.1, initialize_mu_with_complementarity=0, initial_guess_method=guess)
        _, its = s.solve(**qp_kw)
        if self.track_margins:
            from . import margins as _margins
            self._note("qp", _margins.closeness(_margins.solve_with_margins(oq, **qp_kw)[3])[0])   # margin / threshold of its kind: < 1 = knife edge
        x, _, y, _ = s.blocks(s.variables)
        linf = float(np.max(np.abs(y))) if k > 0 else None  # qp.cc:539-546
        return x.copy(), linf, False, len(its)

    def select_step_size(self, params: Params, errors_pre: Errors, d_f, d_eq, penalty, dx):
        """nonlinear.cc:346-412 (armijo_c1 = 1e-4, nonlinear.cc:118)."""
        armijo_c1 = 1.0e-4
        steps: List[Tuple[float, Errors]] = []
        directional = d_f + penalty * d_eq
        alpha = 1.0
        candidate = self.variables.copy()
        for it in range(params.max_line_search_iterations + 1):
            if params.line_search_strategy == POLYNOMIAL_APPROXIMATION:
                if it > 0:
                    if it == 1:                            # :414-427
                        a0, e0 = steps[-1]
                        new_alpha = quadratic_approx_minimum(errors_pre.total(penalty), directional, a0, e0.total(penalty))
                    else:
                        (a0, e0), (a1, e1) = steps[-2], steps[-1]
                        ab = cubic_approx_coeffs(errors_pre.total(penalty), directional, a0, e0.total(penalty), a1,
                                                 e1.total(penalty))
                        new_alpha = cubic_approx_minimum(directional, ab)
                    if new_alpha is not None and math.isfinite(new_alpha):
                        self._note("alpha_valid", min(abs(new_alpha), abs(new_alpha - alpha)) / alpha)
                    if new_alpha is None or not math.isfinite(new_alpha) or new_alpha <= 0.0 or new_alpha >= alpha:
                        return STEP_FAILURE_INVALID_ALPHA, steps, candidate
                    alpha = new_alpha
            elif it > 0:
                alpha = alpha * params.armijo_search_tau
            if self.retraction is not None:                # RetractCandidateVars, :160-168
                candidate = np.asarray(self.retraction(self.variables.copy(), dx, alpha), float)
            else:
                candidate = self.variables + dx * alpha
            e = evaluate_nonlinear_errors(self.p, candidate)
            steps.append((alpha, e))
            if e.invalid():
                return STEP_FAILURE_NON_FINITE_COST, steps, candidate
            if it == 0:
                self._note("first_order", abs(max(abs(d_f), abs(d_eq)) / params.absolute_first_derivative_tol - 1.0))
            if max(abs(d_f), abs(d_eq)) < params.absolute_first_derivative_tol:
                return STEP_FIRST_ORDER_SATISFIED, steps, candidate
            if it == 0:
                self._note("derivative_sign", abs(directional) / max(abs(d_f), abs(penalty * d_eq), 1e-300))
            if directional > 0:
                return STEP_POSITIVE_DERIVATIVE, steps, candidate
            armijo_rhs = errors_pre.total(penalty) + directional * alpha * armijo_c1
            self._note("armijo", abs(e.total(penalty) - armijo_rhs) / max(abs(errors_pre.total(penalty)), 1e-300))
            if e.total(penalty) <= armijo_rhs:
                return STEP_SUCCESS, steps, candidate
        return STEP_MAX_ITERATIONS, steps, candidate

    def solve(self, params: Params, x0):
        """nonlinear.cc:75-158.  Returns (termination, [IterationLog])."""
        self.variables = np.array(x0, dtype=float)
        state = NOMINAL
        lam = params.lambda_initial
        penalty = params.equality_penalty_initial
        logs: List[IterationLog] = []
        has_eq = self.p.equality is not None
        self.margins = []
        for outer in range(params.max_iterations):
            self._iter = outer
            qp, errors_pre = linearize_and_fill_qp(self.variables, lam, self.p)
            dx, linf, indefinite, n_qp = self.compute_step_direction(qp, params)
            if indefinite:
                return QP_INDEFINITE, logs
            d_f, d_eq = compute_qp_cost_derivative(qp, dx)
            if has_eq:
                new_penalty = select_penalty(qp, dx, linf, params.equality_penalty_rho)
                self._note("penalty", abs(new_penalty - penalty) / max(abs(penalty), 1e-300))
                if new_penalty > penalty:
                    penalty = new_penalty * params.equality_penalty_scale_factor
            step_result, steps, candidate = self.select_step_size(params, errors_pre, d_f, d_eq, penalty, dx)
            old_lam = lam
            # UpdateLambdaAndCheckExitConditions, nonlinear.cc:296-339
            exit_state = None
            if step_result == STEP_SUCCESS:
                self.variables = candidate
                if state == ATTEMPTING_RESTORE_LM:
                    lam = max(lam * params.lambda_decrease_on_restore, params.min_lambda)
                else:
                    lam = max(lam * params.lambda_decrease_on_success, params.min_lambda)
                state = NOMINAL
                final = steps[-1][1]
                self._note("absolute_exit", abs(final.linf() / params.absolute_exit_tol - 1.0))
                if final.linf() < params.absolute_exit_tol:
                    exit_state = SATISFIED_ABSOLUTE_TOL
                else:
                    rel_rhs = errors_pre.total(penalty) * (1 - params.relative_exit_tol)
                    self._note("relative_exit", abs(final.total(penalty) - rel_rhs) / max(abs(rel_rhs), 1e-300))
                    if final.total(penalty) > rel_rhs:
                        exit_state = SATISFIED_RELATIVE_TOL
            elif step_result == STEP_FIRST_ORDER_SATISFIED:
                exit_state = SATISFIED_FIRST_ORDER_TOL
            elif step_result in (STEP_MAX_ITERATIONS, STEP_POSITIVE_DERIVATIVE):
                if state == NOMINAL:
                    lam = max(params.lambda_failure_init, lam * 10.0)
                    state = ATTEMPTING_RESTORE_LM
                else:
                    lam *= 10.0
                self._note("max_lambda", abs(lam / params.max_lambda - 1.0) if params.max_lambda > 0 else 1.0)
                if lam > params.max_lambda:
                    exit_state = MAX_LAMBDA
            logs.append(IterationLog(old_lam, errors_pre, d_f, d_eq, penalty, step_result, steps, n_qp, state))
            if self.user_exit_callback is not None:        # :142-149
                proceed = self.user_exit_callback(logs[-1])
                if exit_state is None and not proceed:
                    return USER_CALLBACK, logs
            if exit_state is not None:
                return exit_state, logs
        return MAX_ITERATIONS, logs
