"""ctypes wrapper around oracle/libkkt_oracle.so -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the product
package (mini_opt_amd) never does.  See oracle/kkt_oracle.h for what each entry point restates
(reference file:line citations live there and in kkt_oracle.c).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from dataclasses import dataclass, field

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# MO_ORACLE_LIB: another build of the same source (tests/test_oracle_asan_cpu.py loads libkkt_oracle_asan.so this way)
_LIB_PATH = os.environ.get("MO_ORACLE_LIB") or os.path.join(_HERE, "libkkt_oracle.so")

ORC_OK, ORC_NONPOSITIVE_SLACK, ORC_FACTORIZATION_FAILED = 0, 1, 2
COMPLEMENTARITY, FIXED_DECREASE, PREDICTOR_CORRECTOR = 0, 1, 2
GUESS_NAIVE, GUESS_SOLVE_EQUALITY_CONSTRAINED, GUESS_USER_PROVIDED = 0, 1, 2
SATISFIED_KKT_TOL, MAX_ITERATIONS = 0, 1

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)


def build(force: bool = False) -> str:
    """Compile the oracle with gcc (recipe: oracle/Makefile). Building the checker is not using it."""
    src = os.path.join(_HERE, "kkt_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, os.path.basename(_LIB_PATH)], stdout=subprocess.DEVNULL)
    return _LIB_PATH


class _QP(C.Structure):
    _fields_ = [("n", C.c_int), ("k", C.c_int), ("m", C.c_int), ("G", _dp), ("c", _dp), ("A_eq", _dp),
                ("b_eq", _dp), ("cons_var", _ip), ("cons_a", _dp), ("cons_b", _dp)]


class _Params(C.Structure):
    _fields_ = [("initial_mu", C.c_double), ("sigma", C.c_double), ("termination_kkt_tol", C.c_double),
                ("termination_complementarity_tol", C.c_double), ("max_iterations", C.c_int),
                ("barrier_strategy", C.c_int), ("decrease_mu_only_on_small_error", C.c_int),
                ("initial_guess_method", C.c_int), ("initialize_mu_with_complementarity", C.c_int)]


class _KKT(C.Structure):
    _fields_ = [("r_dual", C.c_double), ("r_comp", C.c_double), ("r_primal_eq", C.c_double),
                ("r_primal_ineq", C.c_double)]


class _IP(C.Structure):
    _fields_ = [("mu", C.c_double), ("alpha_primal", C.c_double), ("alpha_dual", C.c_double),
                ("alpha_probe_primal", C.c_double), ("alpha_probe_dual", C.c_double), ("mu_affine", C.c_double)]


class _Iter(C.Structure):
    _fields_ = [("kkt_initial", _KKT), ("kkt_final", _KKT), ("ip", _IP)]


class _Solver(C.Structure):
    _fields_ = [("qp", _QP), ("N", C.c_int), ("M", C.c_int), ("K", C.c_int), ("P", C.c_int), ("V", C.c_int),
                ("variables", _dp), ("r", _dp), ("r_dual_aug", _dp), ("H", _dp), ("H_inv", _dp), ("delta", _dp),
                ("delta_affine", _dp), ("ldlt_mat", _dp), ("ldlt_transp", _ip), ("ldlt_temp", _dp),
                ("work", _dp)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        L.orc_update_hessian.restype = C.c_double
        L.orc_linearize_dense.restype = C.c_double
        L.orc_compute_alpha_vec.restype = C.c_double
        L.orc_compute_mu.restype = C.c_double
        L.orc_compute_mu_affine.restype = C.c_double
        L.orc_compute_mu_affine.argtypes = [C.c_void_p, C.c_double, C.c_double, C.c_double]
        L.orc_compute_alpha_vec.argtypes = [C.c_int, _dp, _dp, C.c_double]
        L.orc_solve_for_update.argtypes = [C.c_void_p, C.c_double]
        L.orc_solve_for_update_direct.argtypes = [C.c_void_p, C.c_double]
        L.orc_compute_errors.argtypes = [C.c_void_p, C.c_double, C.c_void_p]
        L.orc_iterate.argtypes = [C.c_void_p, C.c_double, C.c_int, C.c_void_p]
        L.orc_newton_step.argtypes = [C.c_void_p, _dp, C.c_double, C.c_double, C.c_int, _dp, _dp]
        L.orc_compute_alpha.argtypes = [C.c_void_p, C.c_double, _dp, _dp]
        L.orc_linearize_dense.argtypes = [C.c_int, C.c_int, _dp, C.c_int, _dp, C.c_double, _dp, _dp]
        L.orc_batched_newton_step.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _dp, C.c_int, _dp,
                                              C.c_double, _dp, _dp, _dp, _dp, _ip, _dp, _dp, _dp, _dp,
                                              C.c_double, C.c_int, C.c_int, _dp, _dp, _ip]
        L.orc_batched_newton_step.restype = C.c_int
        L.orc_batched_solve.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _dp, C.c_int, _dp, C.c_double, _dp, _dp, _dp, _dp,
                                        _ip, _dp, _dp, C.c_void_p, C.c_int, _dp, _ip, _ip]
        L.orc_batched_solve.restype = C.c_int
        _lib = L
    return _lib


def _d(a):
    return a.ctypes.data_as(_dp) if a is not None else None


def _i(a):
    return a.ctypes.data_as(_ip) if a is not None else None


def _f64(a, order="F"):
    return np.array(a, dtype=np.float64, order=order, copy=True)


@dataclass
class QP:
    """Mirror of mini_opt::QP (qp.hpp:104-124); G/A_eq are stored column-major."""
    G: np.ndarray
    c: np.ndarray
    A_eq: np.ndarray | None = None
    b_eq: np.ndarray | None = None
    cons_var: np.ndarray = field(default_factory=lambda: np.zeros(0, np.int32))
    cons_a: np.ndarray = field(default_factory=lambda: np.zeros(0))
    cons_b: np.ndarray = field(default_factory=lambda: np.zeros(0))

    def __post_init__(self):
        self.G = _f64(self.G)
        self.c = _f64(self.c).ravel()
        n = self.G.shape[0]
        if self.A_eq is None or np.size(self.A_eq) == 0:
            self.A_eq = np.zeros((0, n), order="F")
            self.b_eq = np.zeros(0)
        self.A_eq = _f64(np.atleast_2d(self.A_eq))
        self.b_eq = _f64(self.b_eq).ravel()
        self.cons_var = np.ascontiguousarray(self.cons_var, dtype=np.int32).ravel()
        self.cons_a = _f64(self.cons_a).ravel()
        self.cons_b = _f64(self.cons_b).ravel()

    @property
    def dims(self):
        return self.G.shape[0], self.A_eq.shape[0], self.cons_var.shape[0]


class Solver:
    """Mirror of mini_opt::QPInteriorPointSolver incl. its private step functions (qp.hpp:132-295)."""

    def __init__(self, qp: QP):
        self.L = lib()
        self.qp = qp
        n, k, m = qp.dims
        self._cqp = _QP(n, k, m, _d(qp.G), _d(qp.c), _d(qp.A_eq), _d(qp.b_eq), _i(qp.cons_var), _d(qp.cons_a),
                        _d(qp.cons_b))
        self._s = _Solver()
        rc = self.L.orc_solver_setup(C.byref(self._s), C.byref(self._cqp))
        if rc != 0:
            raise ValueError(f"orc_solver_setup failed rc={rc}")
        self.N, self.K, self.M = n, k, m
        self.P, self.V = n + k, n + 2 * m + k

    def __del__(self):
        try:
            self.L.orc_solver_free(C.byref(self._s))
        except Exception:
            pass

    def _vec(self, ptr, n):
        return np.ctypeslib.as_array(ptr, shape=(max(n, 1),))[:n]

    variables = property(lambda self: self._vec(self._s.variables, self.V))
    r = property(lambda self: self._vec(self._s.r, self.V))
    delta = property(lambda self: self._vec(self._s.delta, self.V))
    delta_affine = property(lambda self: self._vec(self._s.delta_affine, self.V))

    @property
    def H(self):
        return np.ctypeslib.as_array(self._s.H, shape=(max(self.P * self.P, 1),))[:self.P * self.P].reshape(
            self.P, self.P, order="F")

    @property
    def H_inv(self):
        return np.ctypeslib.as_array(self._s.H_inv, shape=(max(self.P * self.P, 1),))[:self.P * self.P].reshape(
            self.P, self.P, order="F")

    def blocks(self, v):
        N, M, K = self.N, self.M, self.K
        return v[:N], v[N:N + M], v[N + M:N + M + K], v[N + M + K:]

    def evaluate_kkt(self, include_inequalities=True):
        self.L.orc_evaluate_kkt(C.byref(self._s), int(include_inequalities))

    def compute_ldlt(self, include_inequalities=True):
        return self.L.orc_compute_ldlt(C.byref(self._s), int(include_inequalities))

    def solve_for_update(self, mu, direct=False):
        (self.L.orc_solve_for_update_direct if direct else self.L.orc_solve_for_update)(C.byref(self._s), mu)

    def solve_no_inequalities(self):
        self.L.orc_solve_no_inequalities(C.byref(self._s))

    def compute_alpha(self, tau):
        p, d = C.c_double(), C.c_double()
        self.L.orc_compute_alpha(C.byref(self._s), tau, C.byref(p), C.byref(d))
        return p.value, d.value

    def compute_mu(self):
        return self.L.orc_compute_mu(C.byref(self._s))

    def compute_errors(self, mu):
        e = _KKT()
        self.L.orc_compute_errors(C.byref(self._s), mu, C.byref(e))
        return e

    def iterate(self, mu, strategy=COMPLEMENTARITY):
        out = _IP()
        st = self.L.orc_iterate(C.byref(self._s), mu, strategy, C.byref(out))
        return st, out

    def full_system(self):
        V = self.V
        H = np.zeros((V, V), order="F")
        r = np.zeros(V)
        self.L.orc_build_full_system(C.byref(self._s), _d(H), _d(r))
        return H, r

    def full_system_step(self):
        d = np.zeros(self.V)
        rc = self.L.orc_full_system_step(C.byref(self._s), _d(d))
        return rc, d

    def newton_step(self, vars_, mu, tau=0.995, use_inverse=True):
        vars_ = _f64(vars_).ravel()
        delta = np.zeros(self.V)
        alpha = np.zeros(2)
        st = self.L.orc_newton_step(C.byref(self._s), _d(vars_), mu, tau, int(use_inverse), _d(delta), _d(alpha))
        return st, delta, alpha

    def solve(self, **kw):
        p = _Params()
        self.L.orc_default_params(C.byref(p))
        for k, v in kw.items():
            if not hasattr(p, k):
                raise KeyError(k)
            setattr(p, k, v)
        its = (_Iter * max(p.max_iterations, 1))()
        n_it = C.c_int(0)
        term = self.L.orc_solve(C.byref(self._s), C.byref(p), its, C.byref(n_it))
        return term, [its[i] for i in range(n_it.value)]


def update_hessian(index, J, r, H, b):
    """Residual::Model::UpdateHessian (residual.hpp:186-226). H (n x n, F-order) and b are updated in place."""
    J = _f64(J)
    r = _f64(r).ravel()
    idx = np.ascontiguousarray(index, dtype=np.int32)
    assert H.flags["F_CONTIGUOUS"] and H.dtype == np.float64
    R, Ploc = J.shape
    return lib().orc_update_hessian(R, Ploc, _i(idx), _d(J), _d(r), H.shape[0], _d(H), _d(b))


def linearize_dense(J, r, lam, row_major=True):
    """nonlinear.cc:182-189 for one dense residual: returns (G lower (F-order), c, 0.5|r|^2)."""
    J = np.ascontiguousarray(J, dtype=np.float64) if row_major else _f64(J)
    m_r, n = J.shape
    r = _f64(r).ravel()
    G = np.zeros((n, n), order="F")
    c = np.zeros(n)
    f = lib().orc_linearize_dense(m_r, n, _d(J), int(row_major), _d(r), lam, _d(G), _d(c))
    return G, c, f


def compute_alpha_vec(val, d_val, tau):
    val = _f64(val).ravel()
    d_val = _f64(d_val).ravel()
    return lib().orc_compute_alpha_vec(len(val), _d(val), _d(d_val), tau)


def batched_newton_step(n, k, m, *, J=None, r=None, lam=0.0, G=None, c=None, A_eq=None, b_eq=None, cons_var=None,
                        cons_a=None, cons_b=None, vars_=None, mu=None, tau=0.995, use_inverse=True,
                        num_threads=0, row_major=True):
    """OpenMP batched Newton steps on contiguous [batch][...] slabs (CPU baseline / batch checker)."""
    batch = vars_.shape[0]
    V = n + 2 * m + k
    m_r = 0 if J is None else J.shape[1]
    cc = lambda a, dt=np.float64: None if a is None else np.ascontiguousarray(a, dtype=dt)
    J, r, G, c, A_eq, b_eq, cons_a, cons_b, vars_, mu = map(cc, (J, r, G, c, A_eq, b_eq, cons_a, cons_b, vars_, mu))
    cons_var = cc(cons_var, np.int32)
    delta = np.zeros((batch, V))
    alpha = np.zeros((batch, 2))
    status = np.zeros(batch, dtype=np.int32)
    used = lib().orc_batched_newton_step(batch, n, k, m, m_r, _d(J), int(row_major), _d(r), lam, _d(G), _d(c),
                                         _d(A_eq), _d(b_eq), _i(cons_var), _d(cons_a), _d(cons_b), _d(vars_),
                                         _d(mu), tau, int(use_inverse), num_threads, _d(delta), _d(alpha),
                                         _i(status))
    return delta, alpha, status, used


def batched_solve(n, k, m, *, J=None, r=None, lam=0.0, G=None, c=None, A_eq=None, b_eq=None, cons_var=None, cons_a=None,
                  cons_b=None, vars0=None, batch=None, num_threads=0, row_major=True, **params):
    """OpenMP batched Solve (qp.cc:100-151 per problem) on contiguous [batch][...] slabs: the checker and the CPU baseline of
    `bench.py --mode solve`.  Returns (termination [B] (orc_solve's value: >= 0 termination state, < 0 -status), iterations [B],
    final variables [B, V], threads used)."""
    V = n + 2 * m + k
    if batch is None:
        batch = (J if J is not None else G).shape[0]
    m_r = 0 if J is None else J.shape[1]
    cc = lambda a, dt=np.float64: None if a is None else np.ascontiguousarray(a, dtype=dt)
    J, r, G, c, A_eq, b_eq, cons_a, cons_b = map(cc, (J, r, G, c, A_eq, b_eq, cons_a, cons_b))
    cons_var = cc(cons_var, np.int32)
    p = _Params()
    lib().orc_default_params(C.byref(p))
    for key, val in params.items():
        if not hasattr(p, key):
            raise KeyError(key)
        setattr(p, key, val)
    vars_io = np.zeros((batch, V)) if vars0 is None else np.array(vars0, dtype=np.float64, order="C", copy=True)
    term = np.zeros(batch, dtype=np.int32)
    nit = np.zeros(batch, dtype=np.int32)
    used = lib().orc_batched_solve(batch, n, k, m, m_r, _d(J), int(row_major), _d(r), lam, _d(G), _d(c), _d(A_eq), _d(b_eq),
                                   _i(cons_var), _d(cons_a), _d(cons_b), C.byref(p), num_threads, _d(vars_io), _i(term), _i(nit))
    return term, nit, vars_io, used
