"""Host-side (Python) mirror of mini_opt's constrained nonlinear least squares for BATCHES of problems on one MI355X.

    reference (C++)                                              here
    ---------------------------------------------------------    ------------------------------------------------------
    mini_opt::Problem                      nonlinear.hpp:33-52    Problem (dimension, cost, equality, inequality_constraints)
    ConstrainedNonlinearLeastSquares       nonlinear.hpp:127-230  ConstrainedNonlinearLeastSquares (Solve / variables /
      ::Params                             nonlinear.hpp:64-124     EvaluateNonlinearErrors), Params (same names, defaults)
      ::LinearizeAndFillQP (static)        nonlinear.cc:170-214   fill_qp
      ::ComputeQPCostDerivative (static)   nonlinear.cc:452-483   qp_cost_derivative
    NLSSolverOutputs / NLSIteration        structs.hpp:277-347    NLSSolverOutputs (tensors, one row per problem)

The residual functions are the caller's (the reference's Residual objects are host functors, residual.hpp:28-143): here they
are callables on torch batches, cost(x [B, n], want_J) -> (r [B, m_r], J [B, m_r, n] | None) and equality(...) ->
(r_eq [B, k], J_eq [B, k, n] | None), enqueuing device work on the current stream.  Everything else runs in the HIP library
behind mo_nls_solve (include/mini_opt_hip.h); there is no CPU path here.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field
from typing import Callable, Optional, Sequence, Tuple

import torch

from . import _lib as L
from .qp import BatchedQP, _DT, _ptr, _stream

# NLSTerminationState (structs.hpp:233-248) + MO_NLS_QP_FAILURE (the reference throws there)
(MAX_ITERATIONS, SATISFIED_ABSOLUTE_TOL, SATISFIED_RELATIVE_TOL, SATISFIED_FIRST_ORDER_TOL, MAX_LAMBDA, QP_INDEFINITE,
 USER_CALLBACK, QP_FAILURE) = range(8)
# StepSizeSelectionResult (structs.hpp:215-228)
(STEP_SUCCESS, STEP_MAX_ITERATIONS, STEP_FIRST_ORDER_SATISFIED, STEP_POSITIVE_DERIVATIVE, STEP_FAILURE_NON_FINITE_COST,
 STEP_FAILURE_INVALID_ALPHA) = range(6)
ARMIJO_BACKTRACK, POLYNOMIAL_APPROXIMATION = 0, 1  # LineSearchStrategy (structs.hpp:148-153)


def TerminationStateIndicatesSatisfiedTol(state) -> torch.Tensor:
    """structs.hpp:250-262, element-wise."""
    return (state == SATISFIED_ABSOLUTE_TOL) | (state == SATISFIED_RELATIVE_TOL) | (state == SATISFIED_FIRST_ORDER_TOL)


@dataclass
class Params:
    """ConstrainedNonlinearLeastSquares::Params, nonlinear.hpp:64-124 (same names, same defaults)."""
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
    log_qp_eigenvalues: bool = False   # nonlinear.hpp:122-123: record the QPEigenvalues of every iteration's QP Hessian

    def as_struct(self, retraction: int = L.MO_RETRACT_EUCLIDEAN) -> L.NlsParams:
        p = L.NlsParams()
        L.lib().mo_default_nls_params(C.byref(p))
        for name, _ in L.NlsParams._fields_:
            if hasattr(self, name):
                setattr(p, name, int(getattr(self, name)) if name == "log_qp_eigenvalues" else getattr(self, name))
        p.retraction = int(retraction)   # the Retraction is a constructor argument of the solver (nonlinear.hpp:127), not a Param
        return p


ResidualFn = Callable[[torch.Tensor, bool], Tuple[torch.Tensor, Optional[torch.Tensor]]]

ROSENBROCK, HIMMELBLAU, SPHERE, PRODUCT_PAIRS, ACTUATOR_CHAIN = range(5)  # mo_residual_family
WRAP_PI = "wrap_pi"  # built-in retraction: x + alpha dx wrapped into [-pi, pi) (the reference's robot tests, nonlinear_test.cc:874-880)


def actuator_chain_params(chains, rows, dtype=torch.float64, device="cuda:0") -> torch.Tensor:
    """Parameter block of the ACTUATOR_CHAIN device family (layout: include/mini_opt_hip.h).
    chains: list of chains, each a list of links dict(rotation_xyz=(3), translation=(3), mask=(6 flags), params=(x index per active flag));
    rows:   list of dict(const=float, lin={x index: coefficient}, terms=[(chain, wx, wy, wz), ...])."""
    out = [float(len(chains))]
    for links in chains:
        if len(links) > 8:
            raise L.MiniOptError(-3, "ACTUATOR_CHAIN: at most 8 links per chain")
        out.append(float(len(links)))
        for l in links:
            out.extend(float(v) for v in l["rotation_xyz"])
            out.extend(float(v) for v in l["translation"])
            idx = iter(l["params"])
            out.extend(float(next(idx)) if flag else -1.0 for flag in l["mask"])
    for row in rows:
        out.append(float(row["const"]))
        out.append(float(len(row["lin"])))
        for i, cf in row["lin"].items():
            out.extend((float(i), float(cf)))
        out.append(float(len(row["terms"])))
        for c, wx, wy, wz in row["terms"]:
            out.extend((float(c), float(wx), float(wy), float(wz)))
    return torch.tensor(out, dtype=dtype, device=device)


@dataclass
class DeviceFamily:
    """One of the library's device residual families (mo_residual_eval): evaluated by a HIP kernel straight into the solver's
    buffers -- no torch ops, no copies.  `params`: device tensor of family parameters (PRODUCT_PAIRS: the products v_q)."""
    family: int
    rows: int
    params: Optional[torch.Tensor] = None

    def eval_into(self, plan, x: torch.Tensor, r: torch.Tensor, J: Optional[torch.Tensor], n: int, row_major: bool) -> None:
        B = int(x.shape[0])
        L.check(L.lib().mo_residual_eval(plan.h, self.family, self.rows, _ptr(self.params), _ptr(x), int(x.stride(0)), B, _ptr(r),
                                         self.rows, _ptr(J), self.rows * n, n if row_major else self.rows,
                                         L.MO_ROW_MAJOR if row_major else L.MO_COL_MAJOR, _stream()))

    def __call__(self, x: torch.Tensor, want_J: bool):
        """The ResidualFn contract, for use outside mo_nls_solve (returns J as [B, rows, n])."""
        B, n = int(x.shape[0]), int(x.shape[1])
        plan = _Plan(n, 0, 0, max(self.rows, 1), x.dtype, x.device, B)
        r = torch.empty(B, self.rows, dtype=x.dtype, device=x.device)
        J = torch.empty(B, self.rows, n, dtype=x.dtype, device=x.device) if want_J else None
        self.eval_into(plan, x.contiguous(), r, J, n, True)
        return r, J


@dataclass
class Residual:
    """mini_opt::Residual as MakeResidual<R, P>(index, functor) builds it (residual.hpp:28-143): `fn(params [B, P], want_J) ->
    (r [B, R], J [B, R, P] | None)` sees only the residual's own parameters, picked out of the full vector by `index`."""
    index: Sequence[int]
    fn: ResidualFn
    rows: int

    def Dimension(self) -> int:
        return self.rows

    def QuadraticError(self, params: torch.Tensor) -> torch.Tensor:
        r, _ = self.fn(params[:, list(self.index)], False)
        return 0.5 * (r * r).sum(dim=1)


def MakeResidual(index: Sequence[int], fn: ResidualFn, rows: int) -> Residual:
    return Residual(tuple(int(i) for i in index), fn, int(rows))


def stack_residuals(residuals: Sequence[Residual], n: int) -> ResidualFn:
    """The dense stack of a list of Residuals: values concatenated, every residual's Jacobian columns scattered to its index list
    (UpdateJacobian's J_out.col(index[l]) = J.col(l), residual.hpp:230-250; summing J^T J over residuals, residual.hpp:206-224, is the
    J^T J of the stacked rows)."""
    def fn(x, want_J):
        rs, Js = [], []
        for res in residuals:
            idx = torch.as_tensor(list(res.index), dtype=torch.long, device=x.device)
            r, Jl = res.fn(x.index_select(1, idx), want_J)
            rs.append(r)
            if want_J:
                J = torch.zeros(x.shape[0], res.rows, n, dtype=x.dtype, device=x.device)
                J.index_copy_(2, idx, Jl)
                Js.append(J)
        return torch.cat(rs, dim=1), (torch.cat(Js, dim=1) if want_J else None)
    return fn


@dataclass
class Problem:
    """mini_opt::Problem (nonlinear.hpp:33-52) for a batch of independent problems sharing one structure."""
    dimension: int
    cost: ResidualFn
    cost_rows: int                                  # m_r: stacked rows of all cost residuals
    equality: Optional[ResidualFn] = None
    equality_rows: int = 0                          # k
    inequality_constraints: Sequence[Tuple[int, float, float]] = field(default_factory=list)  # (variable, a, b): a x + b >= 0

    @staticmethod
    def FromResiduals(dimension: int, costs: Sequence[Residual], equality_constraints: Sequence[Residual] = (),
                      inequality_constraints: Sequence[Tuple[int, float, float]] = ()) -> "Problem":
        """The reference's own shape: Problem{costs, inequality_constraints, equality_constraints, dimension} of Residuals."""
        k = sum(r.rows for r in equality_constraints)
        return Problem(dimension, stack_residuals(costs, dimension), sum(r.rows for r in costs),
                       stack_residuals(equality_constraints, dimension) if k else None, k, list(inequality_constraints))


@dataclass
class NLSSolverOutputs:
    """NLSSolverOutputs (structs.hpp:332-347), one row per problem."""
    termination_state: torch.Tensor  # [B] int32
    num_iterations: torch.Tensor     # [B] int32
    iterations: torch.Tensor         # [B, max_iterations, 12 + 3 (max_line_search_iterations + 1)] NLSIteration records
    status: torch.Tensor             # [B] int32 QP status of a problem that ended with QP_FAILURE
    qp_iterations: Optional[torch.Tensor] = None   # [max_iterations, B, max_qp_iterations, 14] QPInteriorPointIteration records (on request)
    qp_lagrange: Optional[torch.Tensor] = None     # [max_iterations, B, 2] {min, l_infinity} of each QP's multipliers (k > 0, on request)
    qp_eigenvalues: Optional[torch.Tensor] = None  # [max_iterations, B, 3] QPEigenvalues {min, max, abs_min} (Params.log_qp_eigenvalues)
    null_space_path: bool = False                  # equality-only problems: the QP of every iteration is QPNullSpaceSolver's

    def NumQPIterations(self) -> torch.Tensor:
        it = torch.nan_to_num(self.iterations[:, :, 10], nan=0.0)
        return it.sum(dim=1).to(torch.int64)

    def NumLineSearchSteps(self) -> torch.Tensor:
        return torch.nan_to_num(self.iterations[:, :, 8], nan=0.0).sum(dim=1).to(torch.int64)

    def NumFailedLineSearches(self) -> torch.Tensor:
        r = self.iterations[:, :, 7]
        return ((r == STEP_MAX_ITERATIONS) | (r == STEP_POSITIVE_DERIVATIVE)).sum(dim=1)


class _Plan:
    def __init__(self, n, k, m, m_r, dtype, device, batch):
        desc = L.PlanDesc(n, k, m, m_r, _DT[dtype], device.index or 0, L.EXTRA_PLAN_FLAGS, 0, batch)
        self.h = C.c_void_p()
        L.check(L.lib().mo_plan_create(C.byref(desc), C.byref(self.h)))

    def __del__(self):
        try:
            L.lib().mo_plan_destroy(self.h)
        except Exception:
            pass


class ConstrainedNonlinearLeastSquares:
    """Batched mirror of mini_opt::ConstrainedNonlinearLeastSquares (nonlinear.hpp:127-230)."""

    def __init__(self, problem: Problem, batch: int, device=None, dtype=torch.float64, retraction=None):
        """retraction: None = x + alpha dx; WRAP_PI = the built-in angle wrap; or a callable (x [B, n], dx [B, n], alpha [B]) ->
        candidate [B, n] on torch tensors -- the reference's custom Retraction (nonlinear.hpp:127, nonlinear.cc:160-168)."""
        if problem is None:
            raise L.MiniOptError(-1, "Must have a valid problem")          # F_ASSERT nonlinear.cc:78
        if dtype != torch.float64:
            raise L.MiniOptError(-3, "mo_nls_solve needs fp64")
        self.p_ = problem
        self.batch = int(batch)
        dev = torch.device(device if device is not None else "cuda:0")
        n, k, m, m_r = problem.dimension, problem.equality_rows, len(problem.inequality_constraints), problem.cost_rows
        self.n, self.k, self.m, self.m_r = n, k, m, m_r
        self._plan = _Plan(n, k, m, m_r, dtype, dev, self.batch)
        z = lambda *shape, dt=dtype: torch.zeros(*shape, dtype=dt, device=dev)
        B = self.batch
        self.variables_ = z(B, n)
        self.candidate_vars_ = z(B, n)
        self.J, self.r, self.r_cand = z(B, m_r, n), z(B, m_r), z(B, m_r)
        self.J_eq, self.r_eq, self.r_eq_cand = (z(B, n, k), z(B, k), z(B, k)) if k else (None, None, None)   # J_eq: k x n col-major
        if m:
            self.cons_var = torch.tensor([[c[0] for c in problem.inequality_constraints]], dtype=torch.int32, device=dev)
            self.cons_a = torch.tensor([[c[1] for c in problem.inequality_constraints]], dtype=dtype, device=dev)
            self.cons_b = torch.tensor([[c[2] for c in problem.inequality_constraints]], dtype=dtype, device=dev)
        else:
            self.cons_var = self.cons_a = self.cons_b = None
        self.retraction_ = retraction
        self._retract_code = (L.MO_RETRACT_EUCLIDEAN if retraction is None else L.MO_RETRACT_WRAP_PI if retraction == WRAP_PI
                              else L.MO_RETRACT_CALLBACK)
        self.step_, self.step_alpha_ = (z(B, n), z(B)) if self._retract_code == L.MO_RETRACT_CALLBACK else (None, None)
        self.user_exit_callback_ = None
        self.user_exit_ = None
        self._outputs_view = None

    def SetUserExitCallback(self, callback) -> None:
        """nonlinear.hpp:157: callback(iteration index, NLSSolverOutputs view of the records so far) -> bool, or a bool tensor [B]
        (False = stop that problem).  A stopped problem ends with USER_CALLBACK unless the iteration terminated it anyway."""
        self.user_exit_callback_ = callback
        self.user_exit_ = torch.zeros(self.batch, dtype=torch.int32, device=self.variables_.device) if callback is not None else None

    # ---- callbacks: enqueue the user's residual evaluation on the stream the library works on
    def _eval(self, user, what, stream):
        try:
            if what == L.MO_NLS_EVAL_RETRACT:                  # the caller's Retraction, nonlinear.cc:160-168
                self.candidate_vars_.copy_(self.retraction_(self.variables_, self.step_, self.step_alpha_))
                return 0
            if what == L.MO_NLS_EVAL_ITERATION_DONE:           # SetUserExitCallback, nonlinear.cc:142-149
                self._iter_done += 1
                proceed = self.user_exit_callback_(self._iter_done - 1, self._outputs_view)
                if isinstance(proceed, torch.Tensor):
                    self.user_exit_.copy_((~proceed.to(torch.bool)).to(torch.int32))
                else:
                    self.user_exit_.fill_(0 if proceed else 1)
                return 0
            lin = what == L.MO_NLS_EVAL_LINEARIZE
            x = self.variables_ if lin else self.candidate_vars_
            r_buf = self.r if lin else self.r_cand
            if isinstance(self.p_.cost, DeviceFamily):          # a HIP kernel writes the solver's buffers directly
                self.p_.cost.eval_into(self._plan, x, r_buf, self.J if lin else None, self.n, True)
            else:
                r, J = self.p_.cost(x, lin)
                r_buf.copy_(r)
                if lin:
                    self.J.copy_(J)
            if self.k:
                req_buf = self.r_eq if lin else self.r_eq_cand
                if isinstance(self.p_.equality, DeviceFamily):
                    self.p_.equality.eval_into(self._plan, x, req_buf, self.J_eq if lin else None, self.n, False)  # k x n col-major
                else:
                    r_eq, J_eq = self.p_.equality(x, lin)
                    req_buf.copy_(r_eq)
                    if lin:
                        self.J_eq.copy_(J_eq.transpose(1, 2))
            return 0
        except Exception as e:  # an exception must not unwind through the C frames
            self._callback_error = e
            return 1

    def _problem_struct(self) -> L.NlsProblem:
        n, k, m, m_r = self.n, self.k, self.m, self.m_r
        p = L.NlsProblem()
        p.vars, p.vars_stride = _ptr(self.variables_), n
        p.candidate, p.candidate_stride = _ptr(self.candidate_vars_), n
        p.J, p.J_stride, p.J_ld, p.J_layout = _ptr(self.J), m_r * n, n, L.MO_ROW_MAJOR
        p.r, p.r_stride = _ptr(self.r), m_r
        p.r_cand, p.r_cand_stride = _ptr(self.r_cand), m_r
        if k:
            p.J_eq, p.J_eq_stride, p.J_eq_ld = _ptr(self.J_eq), n * k, k
            p.r_eq, p.r_eq_stride = _ptr(self.r_eq), k
            p.r_eq_cand, p.r_eq_cand_stride = _ptr(self.r_eq_cand), k
        if m:
            p.cons_var, p.cons_a, p.cons_b, p.cons_stride = _ptr(self.cons_var), _ptr(self.cons_a), _ptr(self.cons_b), 0
        if self.step_ is not None:
            p.step, p.step_stride, p.step_alpha = _ptr(self.step_), n, _ptr(self.step_alpha_)
        if self.user_exit_ is not None:
            p.user_exit = _ptr(self.user_exit_)
        return p

    def Solve(self, params: Params, variables: torch.Tensor, record_qp_iterations: bool = False) -> NLSSolverOutputs:
        """nonlinear.cc:75-158 for every problem of the batch.  record_qp_iterations keeps every QP's QPInteriorPointIteration records
        (NLSIteration::qp_outputs) for mini_opt_amd.serialization."""
        if tuple(variables.shape) != (self.batch, self.n):
            raise L.MiniOptError(-2, f"variables must be [{self.batch}, {self.n}]")
        self.variables_.copy_(variables)
        dev = self.variables_.device
        B = self.batch
        rec = L.MO_NLS_ITER_HEADER + 3 * (params.max_line_search_iterations + 1)
        term = torch.zeros(B, dtype=torch.int32, device=dev)
        nit = torch.zeros(B, dtype=torch.int32, device=dev)
        status = torch.zeros(B, dtype=torch.int32, device=dev)
        its = torch.full((B, max(params.max_iterations, 1), rec), float("nan"), dtype=torch.float64, device=dev)
        sp = params.as_struct(self._retract_code)
        prob = self._problem_struct()
        qp_its = qp_lag = None
        if record_qp_iterations:
            qp_its = torch.full((max(params.max_iterations, 1), B, max(params.max_qp_iterations, 1), L.MO_ITER_RECORD), float("nan"),
                                dtype=torch.float64, device=dev)
            prob.qp_iterations = _ptr(qp_its)
            if self.k:
                qp_lag = torch.full((max(params.max_iterations, 1), B, 2), float("nan"), dtype=torch.float64, device=dev)
                prob.qp_lagrange = _ptr(qp_lag)
        qp_eig = None
        if params.log_qp_eigenvalues:   # NLSIteration::qp_eigenvalues (structs.hpp:267-310), NaN where a problem never ran the iteration
            qp_eig = torch.full((max(params.max_iterations, 1), B, 3), float("nan"), dtype=torch.float64, device=dev)
            prob.qp_eigenvalues = _ptr(qp_eig)
        self._callback_error = None
        self._iter_done = 0
        null_path = bool(L.lib().mo_plan_nls_uses_nullspace(self._plan.h))   # the C side's own predicate (shape AND kernel capacity)
        self._outputs_view = NLSSolverOutputs(term, nit, its, status, qp_its, qp_lag, qp_eig, null_path)
        cb = L.NLS_EVAL_FN(self._eval)
        rc = L.lib().mo_nls_solve(self._plan.h, C.byref(prob), B, C.byref(sp), cb, None, _ptr(term), _ptr(nit), _ptr(its),
                                  _ptr(status), _stream())
        if self._callback_error is not None:
            raise self._callback_error
        L.check(rc)
        return self._outputs_view

    def variables(self) -> torch.Tensor:
        return self.variables_

    def EvaluateNonlinearErrors(self, vars_: torch.Tensor) -> torch.Tensor:
        """nonlinear.cc:279-293: [B, 2] = {f, equality}."""
        r, _ = self.p_.cost(vars_, False)
        r_eq = self.p_.equality(vars_, False)[0] if self.k else None
        return nonlinear_errors(self._plan, r.contiguous(), None if r_eq is None else r_eq.contiguous())


# ---- the static pieces, usable on their own ----------------------------------------------------------------------
def nonlinear_errors(plan: _Plan, r: torch.Tensor, r_eq: Optional[torch.Tensor]) -> torch.Tensor:
    B = int(r.shape[0])
    out = torch.empty(B, 2, dtype=r.dtype, device=r.device)
    L.check(L.lib().mo_nonlinear_errors(plan.h, _ptr(r), int(r.shape[1]), _ptr(r_eq), 0 if r_eq is None else int(r_eq.shape[1]),
                                        B, _ptr(out), _stream()))
    return out


def fill_qp(problem: BatchedQP, x: torch.Tensor):
    """LinearizeAndFillQP (nonlinear.cc:170-214) for dense stacks: `problem` carries (J, r, lam), the equality stack as
    (A_eq, b_eq) and the UNSHIFTED constraints.  Returns (G, c, shifted cons_b, errors [B, 2], status [B])."""
    ref = problem.J
    B, n, m = int(ref.shape[0]), problem.n, problem.m
    plan = _Plan(n, problem.k, m, problem.m_r, ref.dtype, ref.device, B)
    G = torch.empty(B, n, n, dtype=ref.dtype, device=ref.device)
    c = torch.empty(B, n, dtype=ref.dtype, device=ref.device)
    cb = torch.empty(B, m, dtype=ref.dtype, device=ref.device)
    err = torch.empty(B, 2, dtype=ref.dtype, device=ref.device)
    status = torch.empty(B, dtype=torch.int32, device=ref.device)
    prob = problem.as_struct()
    L.check(L.lib().mo_fill_qp(plan.h, C.byref(prob), B, _ptr(x), n, _ptr(G), n * n, n, _ptr(c), n, _ptr(cb), m, _ptr(err),
                               _ptr(status), _stream()))
    return G, c, cb, err, status


def qp_cost_derivative(problem: BatchedQP, dx: torch.Tensor):
    """ComputeQPCostDerivative (nonlinear.cc:452-483): ([B, 2] = {d_f, d_equality}, dx^T G dx [B])."""
    ref = problem._any()
    B, n = int(ref.shape[0]), problem.n
    plan = _Plan(n, problem.k, 0, problem.m_r, ref.dtype, ref.device, B)
    out = torch.empty(B, 2, dtype=ref.dtype, device=ref.device)
    quad = torch.empty(B, dtype=ref.dtype, device=ref.device)
    q = BatchedQP(n=n, k=problem.k, J=problem.J, r=problem.r, lam=problem.lam, lam_vec=problem.lam_vec, G=problem.G, c=problem.c,
                  A_eq=problem.A_eq, b_eq=problem.b_eq)
    prob = q.as_struct()
    L.check(L.lib().mo_qp_cost_derivative(plan.h, C.byref(prob), B, _ptr(dx), n, _ptr(out), _ptr(quad), _stream()))
    return out, quad
