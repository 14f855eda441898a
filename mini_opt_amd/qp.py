"""Host-side (Python) mirror of mini_opt's QP interface for BATCHES of problems resident on one MI355X.

    reference (C++)                                   here
    ----------------------------------------------    -------------------------------------------------
    mini_opt::QP                       qp.hpp:104-124  BatchedQP       (G, c | J, r, lambda; A_eq, b_eq; constraints)
    QPInteriorPointSolver              qp.hpp:132-295  QPInteriorPointSolver (Setup / Solve / x_block ... / SetVariables)
      ::Params                         qp.hpp:134-164  Params (same names and defaults)
      private step functions via `friend QPSolverTest` NewtonStep / Iterate / EvaluateKKTConditions (public test hooks)
    ConstrainedNonlinearLeastSquares::LinearizeAndFillQP (cost part, nonlinear.cc:182-189)   BatchedQP.linearize()

All arithmetic happens in the HIP library behind the C ABI (include/mini_opt_hip.h); torch is used only to own device
memory and streams.  There is no CPU path here.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field
from typing import Optional

import torch

from . import _lib as L

COMPLEMENTARITY, FIXED_DECREASE, PREDICTOR_CORRECTOR = 0, 1, 2
NAIVE, SOLVE_EQUALITY_CONSTRAINED, USER_PROVIDED = 0, 1, 2
SATISFIED_KKT_TOL, MAX_ITERATIONS = 0, 1

_DT = {torch.float64: L.MO_F64, torch.float32: L.MO_F32}


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


@dataclass
class BatchedQP:
    """A batch of QPs of identical shape on one GPU (mini_opt::QP, qp.hpp:104-124).

    Tensor layouts (contiguous, problem index first):
      J    [B, m_r, n]  stacked Jacobian rows (row-major)         r  [B, m_r]      lam: LM damping (nonlinear.cc:187-189)
           other layouts the C ABI takes: J_layout = "row" with J [B, m_r, ld >= n] (a leading dimension beyond n), or
           J_layout = "col" with J [B, n, ld >= m_r] (column-major: tensor row i is column i of J; J_rows = m_r if ld > m_r)
      G    [B, n, n]    memory = n x n COLUMN-major, lower triangle read (pass G_colmajor[b] = G.T for a symmetric G)
      c    [B, n]
      A_eq [B, n, k]    memory = k x n COLUMN-major (A_eq[b] = A.T)  b_eq [B, k]
      cons_var [B, m] int32, cons_a [B, m], cons_b [B, m]          a * x[var] + b >= 0  (qp.hpp:28-70)
    Any of the constraint tensors may have B = 1 to be shared by the whole batch (stride 0).
    """
    n: int
    k: int = 0
    m: int = 0
    J: Optional[torch.Tensor] = None
    r: Optional[torch.Tensor] = None
    lam: float = 0.0
    lam_vec: Optional[torch.Tensor] = None   # [B] per-problem damping; overrides lam (the lambda state of nonlinear.cc:92-96)
    G: Optional[torch.Tensor] = None
    c: Optional[torch.Tensor] = None
    A_eq: Optional[torch.Tensor] = None
    b_eq: Optional[torch.Tensor] = None
    cons_var: Optional[torch.Tensor] = None
    cons_a: Optional[torch.Tensor] = None
    cons_b: Optional[torch.Tensor] = None
    J_layout: str = "row"
    J_rows: Optional[int] = None

    @property
    def m_r(self) -> int:
        if self.J is None:
            return 0
        if self.J_layout == "col":
            return int(self.J_rows if self.J_rows is not None else self.J.shape[2])
        return int(self.J.shape[1])

    @property
    def V(self) -> int:
        return self.n + 2 * self.m + self.k

    def ComputeEigenvalueStats(self) -> torch.Tensor:
        """QP::ComputeEigenvalueStats (qp.hpp:122-123): [B, 3] = QPEigenvalues {min, max, abs_min} of the Hessian of every QP of the batch."""
        return eigenvalue_stats(self)

    def _any(self) -> torch.Tensor:
        for t in (self.J, self.G):
            if t is not None:
                return t
        raise ValueError("BatchedQP needs J or G")

    @property
    def dtype(self):
        return self._any().dtype

    @property
    def device(self):
        return self._any().device

    def batch_of(self, t):
        return int(t.shape[0])

    def as_struct(self) -> L.Problem:
        p = L.Problem()
        ref = self._any()

        def chk(t, shape_tail, dtype=None):
            if t is None:
                return
            if not t.is_cuda or t.device != ref.device:
                raise ValueError("all tensors must live on the same GPU")
            if not t.is_contiguous():
                raise ValueError("tensors must be contiguous")
            if tuple(t.shape[1:]) != tuple(shape_tail):
                raise ValueError(f"bad shape {tuple(t.shape)}, expected [B,{shape_tail}]")
            if t.dtype != (dtype or ref.dtype):
                raise ValueError(f"bad dtype {t.dtype}")

        def stride(t, per):
            return 0 if t.shape[0] == 1 else per

        n, k, m = self.n, self.k, self.m
        if self.J is not None:
            chk(self.r, (self.m_r,))
            ld = int(self.J.shape[2])
            if self.J_layout == "col":
                chk(self.J, (n, ld))
                if ld < self.m_r:
                    raise ValueError("column-major J: leading dimension < m_r")
                p.J, p.J_stride, p.J_ld, p.J_layout = _ptr(self.J), stride(self.J, n * ld), ld, L.MO_COL_MAJOR
            else:
                chk(self.J, (self.m_r, ld))
                if ld < n:
                    raise ValueError("row-major J: leading dimension < n")
                p.J, p.J_stride, p.J_ld, p.J_layout = _ptr(self.J), stride(self.J, self.m_r * ld), ld, L.MO_ROW_MAJOR
            p.r, p.r_stride = _ptr(self.r), stride(self.r, self.m_r)
            p.lam = float(self.lam)
            if self.lam_vec is not None:
                chk(self.lam_vec, ())
                p.lambda_vec, p.lambda_stride = _ptr(self.lam_vec), stride(self.lam_vec, 1)
        else:
            chk(self.G, (n, n)); chk(self.c, (n,))
            p.G, p.G_stride, p.G_ld = _ptr(self.G), stride(self.G, n * n), n
            p.c, p.c_stride = _ptr(self.c), stride(self.c, n)
        if k > 0:
            chk(self.A_eq, (n, k)); chk(self.b_eq, (k,))
            p.A_eq, p.A_stride, p.A_ld = _ptr(self.A_eq), stride(self.A_eq, n * k), k
            p.b_eq, p.b_stride = _ptr(self.b_eq), stride(self.b_eq, k)
        if m > 0:
            chk(self.cons_var, (m,), torch.int32); chk(self.cons_a, (m,)); chk(self.cons_b, (m,))
            if not (self.cons_var.shape[0] == self.cons_a.shape[0] == self.cons_b.shape[0]):
                raise ValueError("constraint tensors must share their batch dimension")
            p.cons_var, p.cons_a, p.cons_b = _ptr(self.cons_var), _ptr(self.cons_a), _ptr(self.cons_b)
            p.cons_stride = stride(self.cons_var, m)
        return p


@dataclass
class Params:
    """QPInteriorPointSolver::Params, qp.hpp:134-164 (same names, same defaults)."""
    initial_mu: float = 1.0
    sigma: float = 0.5
    termination_kkt_tol: float = 1.0e-9
    termination_complementarity_tol: float = 1.0e-6
    max_iterations: int = 10
    barrier_strategy: int = COMPLEMENTARITY
    decrease_mu_only_on_small_error: bool = False
    initial_guess_method: int = NAIVE
    initialize_mu_with_complementarity: bool = False

    def as_struct(self) -> L.SolveParams:
        s = L.SolveParams()
        for f in ("initial_mu", "sigma", "termination_kkt_tol", "termination_complementarity_tol"):
            setattr(s, f, float(getattr(self, f)))
        for f in ("max_iterations", "barrier_strategy", "decrease_mu_only_on_small_error", "initial_guess_method",
                  "initialize_mu_with_complementarity"):
            setattr(s, f, int(getattr(self, f)))
        return s


@dataclass
class SolverOutputs:
    """QPInteriorPointSolverOutputs (structs.hpp:116-134), one entry per problem of the batch."""
    termination_state: torch.Tensor          # [B] int32
    num_iterations: torch.Tensor             # [B] int32
    iterations: torch.Tensor                 # [B, max_iterations, 14]  (kkt_initial[4], kkt_final[4], ip[6])
    lagrange_multipliers: Optional[torch.Tensor]  # [B, 2] {min, l_infinity} or None if k == 0
    status: torch.Tensor                     # [B] int32 MO_STATUS_*


class FailedFactorization(RuntimeError):
    """qp.hpp:331-333; raised for batch == 1 (larger batches report per-problem status words)."""


class QPInteriorPointSolver:
    """Batched mirror of mini_opt::QPInteriorPointSolver (qp.hpp:132-295)."""

    def __init__(self, problem: Optional[BatchedQP] = None, batch: Optional[int] = None, force_generic: bool = False, no_tiny: bool = False):
        self._plan = None
        self._force_generic = force_generic
        self._no_tiny = no_tiny
        self.p_: Optional[BatchedQP] = None
        if problem is not None:
            self.Setup(problem, batch)

    # -- Setup, qp.cc:20-73 ------------------------------------------------------------------------------------
    def Setup(self, problem: BatchedQP, batch: Optional[int] = None) -> None:
        if problem is None:
            raise L.MiniOptError(-1, "Must pass a non-null problem")
        lib = L.lib()
        self._destroy()
        self.p_ = problem
        ref = problem._any()
        self.batch = int(batch if batch is not None else ref.shape[0])
        desc = L.PlanDesc(problem.n, problem.k, problem.m, problem.m_r, _DT[problem.dtype],
                          ref.device.index or 0, (L.MO_PLAN_FORCE_GENERIC if self._force_generic else 0) | (L.MO_PLAN_NO_TINY if self._no_tiny else 0) | L.EXTRA_PLAN_FLAGS, 0, self.batch)
        plan = C.c_void_p()
        L.check(lib.mo_plan_create(C.byref(desc), C.byref(plan)))
        self._plan = plan
        self._prob = problem.as_struct()
        self.variables_ = torch.zeros(self.batch, problem.V, dtype=problem.dtype, device=ref.device)
        self.delta_ = torch.zeros_like(self.variables_)
        self.r_ = torch.zeros_like(self.variables_)
        self.status_ = torch.zeros(self.batch, dtype=torch.int32, device=ref.device)

    def _destroy(self):
        if self._plan is not None:
            L.lib().mo_plan_destroy(self._plan)
            self._plan = None

    def __del__(self):
        try:
            self._destroy()
        except Exception:
            pass

    # -- accessors, qp.cc:205-226 ------------------------------------------------------------------------------
    def _blk(self, t, which):
        n, m, k = self.p_.n, self.p_.m, self.p_.k
        lo, hi = {"x": (0, n), "s": (n, n + m), "y": (n + m, n + m + k), "z": (n + m + k, n + 2 * m + k)}[which]
        return t[:, lo:hi]

    def x_block(self): return self._blk(self.variables_, "x")
    def s_block(self): return self._blk(self.variables_, "s")
    def y_block(self): return self._blk(self.variables_, "y")
    def z_block(self): return self._blk(self.variables_, "z")
    def variables(self): return self.variables_

    def SetVariables(self, v: torch.Tensor) -> None:
        self.variables_.copy_(v)

    def problem(self) -> BatchedQP:
        if self.p_ is None:
            raise L.MiniOptError(-1, "Cannot call unless initialized")
        return self.p_

    def step_kernel(self) -> str:
        return L.lib().mo_plan_step_kernel(self._plan, C.byref(self._prob)).decode()

    def solve_kernel(self) -> str:
        return L.lib().mo_plan_solve_kernel(self._plan, C.byref(self._prob)).decode()

    def _mu_arg(self, mu):
        if isinstance(mu, torch.Tensor):
            assert mu.dtype == self.p_.dtype and mu.is_cuda and mu.is_contiguous()
            return mu, (0 if mu.numel() == 1 else 1)
        t = torch.full((1,), float(mu), dtype=self.p_.dtype, device=self.variables_.device)
        return t, 0

    # -- test hooks replacing `friend class QPSolverTest` (qp.hpp:293) -------------------------------------------
    def EvaluateKKTConditions(self, mu=0.0, include_inequalities: bool = True):
        """qp.cc:391-420 + ComputeErrors qp.cc:423-437. Returns (r_ [B,V], kkt [B,4])."""
        mu_t, mu_s = self._mu_arg(mu)
        kkt = torch.zeros(self.batch, 4, dtype=self.p_.dtype, device=self.variables_.device)
        flags = 0 if include_inequalities else L.MO_STEP_NO_INEQUALITIES
        L.check(L.lib().mo_kkt_residual(self._plan, C.byref(self._prob), self.batch, _ptr(self.variables_), self.p_.V,
                                        _ptr(mu_t), mu_s, flags, _ptr(self.r_), self.p_.V, _ptr(kkt), _stream()))
        return self.r_, kkt

    def NewtonStep(self, mu=0.0, tau: float = 0.995, include_inequalities: bool = True):
        """EvaluateKKTConditions -> ComputeLDLT -> SolveForUpdate(mu) -> ComputeAlpha(tau) on the current state
        (qp_test.cc:132-134, qp.cc:192). Returns (delta_ [B,V], alpha [B,2], status [B])."""
        mu_t, mu_s = self._mu_arg(mu)
        alpha = torch.zeros(self.batch, 2, dtype=self.p_.dtype, device=self.variables_.device)
        flags = 0 if include_inequalities else L.MO_STEP_NO_INEQUALITIES
        L.check(L.lib().mo_newton_step(self._plan, C.byref(self._prob), self.batch, _ptr(self.variables_), self.p_.V,
                                       _ptr(mu_t), mu_s, float(tau), flags, _ptr(self.delta_), self.p_.V, _ptr(alpha),
                                       _ptr(self.status_), _stream()))
        return self.delta_, alpha, self.status_

    def Iterate(self, mu, strategy: int = COMPLEMENTARITY):
        """qp.cc:153-201: updates variables_ in place. Returns (ip_outputs [B,6], status [B])."""
        mu_t, mu_s = self._mu_arg(mu)
        ip = torch.zeros(self.batch, 6, dtype=self.p_.dtype, device=self.variables_.device)
        L.check(L.lib().mo_iterate(self._plan, C.byref(self._prob), self.batch, _ptr(self.variables_), self.p_.V,
                                   _ptr(mu_t), mu_s, int(strategy), _ptr(self.delta_), self.p_.V, _ptr(ip),
                                   _ptr(self.status_), _stream()))
        return ip, self.status_

    # -- Solve, qp.cc:100-151 ----------------------------------------------------------------------------------
    def Solve(self, params: Params, record_iterations: bool = True) -> SolverOutputs:
        """qp.cc:100-151.  record_iterations=False skips the per-iteration QPInteriorPointIteration records (outputs.iterations is None):
        the kernel then never takes the square roots of the KKT norms."""
        if self.p_ is None:
            raise L.MiniOptError(-1, "Must have a valid problem")
        dev, dt = self.variables_.device, self.p_.dtype
        term = torch.zeros(self.batch, dtype=torch.int32, device=dev)
        nit = torch.zeros(self.batch, dtype=torch.int32, device=dev)
        its = torch.full((self.batch, max(int(params.max_iterations), 1), L.MO_ITER_RECORD), float("nan"), dtype=dt,
                         device=dev) if record_iterations else None
        lag = torch.zeros(self.batch, 2, dtype=dt, device=dev) if self.p_.k > 0 else None
        sp = params.as_struct()
        L.check(L.lib().mo_qp_solve(self._plan, C.byref(self._prob), self.batch, C.byref(sp), _ptr(self.variables_),
                                    self.p_.V, _ptr(term), _ptr(nit), _ptr(its), _ptr(lag), _ptr(self.status_),
                                    _stream()))
        if self.batch == 1:
            st = int(self.status_[0])
            if st == L.MO_STATUS_FACTORIZATION_FAILED:
                raise FailedFactorization("Failed to solve self-adjoint (lower) system. The hessian may not be semi-definite.")
            if st != L.MO_STATUS_OK:
                raise L.MiniOptError(-1, L.lib().mo_status_string(st).decode())
        return SolverOutputs(term, nit, its, lag, self.status_)


class QPNullSpaceSolver:
    """Batched mirror of mini_opt::QPNullSpaceSolver (qp.hpp:296-315, qp.cc:679-729): equality-constrained QPs, no inequalities.
    Solve returns QPNullSpaceTerminationState per problem (0 SUCCESS, 1 NOT_POSITIVE_DEFINITE); variables() is x [B, n]."""
    SUCCESS, NOT_POSITIVE_DEFINITE = 0, 1

    def __init__(self):
        self.x_ = None

    def Solve(self, p: BatchedQP) -> torch.Tensor:
        lib = L.lib()
        ref = p._any()
        B = max(int(t.shape[0]) for t in (p.J, p.G, p.A_eq) if t is not None)
        desc = L.PlanDesc(p.n, p.k, 0, p.m_r, _DT[p.dtype], ref.device.index or 0, L.EXTRA_PLAN_FLAGS, 0, B)
        plan = C.c_void_p()
        L.check(lib.mo_plan_create(C.byref(desc), C.byref(plan)))
        try:
            q = BatchedQP(n=p.n, k=p.k, J=p.J, r=p.r, lam=p.lam, lam_vec=p.lam_vec, G=p.G, c=p.c, A_eq=p.A_eq, b_eq=p.b_eq)
            prob = q.as_struct()
            self.x_ = torch.empty(B, p.n, dtype=ref.dtype, device=ref.device)
            term = torch.empty(B, dtype=torch.int32, device=ref.device)
            L.check(lib.mo_nullspace_solve(plan, C.byref(prob), B, _ptr(self.x_), p.n, _ptr(term), _stream()))
        finally:
            lib.mo_plan_destroy(plan)
        return term

    def variables(self) -> torch.Tensor:
        return self.x_


def eigenvalue_stats(problem: BatchedQP) -> torch.Tensor:
    """QP::ComputeEigenvalueStats (qp.hpp:122-123, qp.cc:12-16) for every QP of the batch: [B, 3] = QPEigenvalues {min, max, abs_min} of the
    Hessian -- sym(G) from the lower triangle of (G, c) input, or J^T J + lambda I for (J, r, lambda) input."""
    lib = L.lib()
    ref = problem.J if problem.J is not None else problem.G
    B = int(ref.shape[0])
    desc = L.PlanDesc(problem.n, 0, 0, problem.m_r, _DT[problem.dtype], ref.device.index or 0, L.EXTRA_PLAN_FLAGS, 0, B)
    plan = C.c_void_p()
    L.check(lib.mo_plan_create(C.byref(desc), C.byref(plan)))
    try:
        out = torch.empty(B, 3, dtype=problem.dtype, device=ref.device)
        prob = problem.as_struct()   # (only the cost part is read)
        L.check(lib.mo_qp_eigenvalue_stats(plan, C.byref(prob), B, _ptr(out), _stream()))
        return out
    finally:
        lib.mo_plan_destroy(plan)


def linearize(problem: BatchedQP, force_generic: bool = False):
    """Cost part of LinearizeAndFillQP (nonlinear.cc:182-189): returns (G [B,n,n] col-major lower, c [B,n], 0.5|r|^2 [B])."""
    lib = L.lib()
    ref = problem.J
    B, n = int(ref.shape[0]), problem.n
    desc = L.PlanDesc(n, 0, 0, problem.m_r, _DT[problem.dtype], ref.device.index or 0,
                      (L.MO_PLAN_FORCE_GENERIC if force_generic else 0) | L.EXTRA_PLAN_FLAGS, 0, B)
    plan = C.c_void_p()
    L.check(lib.mo_plan_create(C.byref(desc), C.byref(plan)))
    try:
        G = torch.empty(B, n, n, dtype=ref.dtype, device=ref.device)
        c = torch.empty(B, n, dtype=ref.dtype, device=ref.device)
        f = torch.empty(B, dtype=ref.dtype, device=ref.device)
        q = BatchedQP(n=n, J=problem.J, r=problem.r, lam=problem.lam, J_layout=problem.J_layout, J_rows=problem.J_rows)
        prob = q.as_struct()
        L.check(lib.mo_linearize(plan, C.byref(prob), B, _ptr(G), n * n, n, _ptr(c), n, _ptr(f), _stream()))
    finally:
        lib.mo_plan_destroy(plan)
    return G, c, f
