"""Loader + ctypes declarations for the C ABI in include/mini_opt_hip.h (lib/libminiopt_hip.so).

The HIP library is the product: if it is missing, this module raises -- there is no CPU or PyTorch fallback.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MO_LIB_PATH") or os.path.join(_HERE, "lib", "libminiopt_hip.so")  # MO_LIB_PATH: A/B runs against another build
CSRC = os.path.join(_HERE, "csrc")

MO_OK = 0
MO_F64, MO_F32 = 0, 1
MO_COL_MAJOR, MO_ROW_MAJOR = 0, 1
MO_PLAN_FORCE_GENERIC = 1
MO_PLAN_NO_TINY = 2
MO_PLAN_TICKETS_ALWAYS = 4
MO_PLAN_STATIC_ROUNDS_ALWAYS = 8
# Plan flags OR-ed into every plan the Python mirror creates (tests/test_gpu_ticket_path.py re-runs the parity tests under each scheduling
# scheme this way).  A knob of THIS mirror: the library itself reads no environment variable.
EXTRA_PLAN_FLAGS = int(os.environ.get("MO_PLAN_EXTRA_FLAGS", "0"), 0)
MO_STEP_NO_INEQUALITIES = 1
(MO_STATUS_OK, MO_STATUS_NONPOSITIVE_SLACK, MO_STATUS_FACTORIZATION_FAILED, MO_STATUS_NONFINITE, MO_STATUS_BAD_INDEX,
 MO_STATUS_NOT_POSITIVE_DEFINITE) = range(6)
MO_KKT_RECORD, MO_IP_RECORD, MO_ITER_RECORD = 4, 6, 14

# every symbol include/mini_opt_hip.h declares
EXPORTS = ["mo_version_string", "mo_status_string", "mo_last_error", "mo_default_solve_params", "mo_plan_create",
           "mo_plan_destroy", "mo_plan_step_kernel", "mo_plan_solve_kernel", "mo_plan_nls_uses_nullspace", "mo_linearize", "mo_kkt_residual", "mo_newton_step", "mo_iterate",
           "mo_qp_solve", "mo_fill_qp", "mo_nonlinear_errors", "mo_qp_cost_derivative",
           "mo_default_nls_params", "mo_nls_solve", "mo_nullspace_solve", "mo_residual_eval", "mo_qp_eigenvalue_stats"]


class PlanDesc(C.Structure):
    _fields_ = [("n", C.c_int32), ("k", C.c_int32), ("m", C.c_int32), ("m_r", C.c_int32), ("dtype", C.c_int32),
                ("device", C.c_int32), ("flags", C.c_uint32), ("reserved", C.c_int32), ("max_batch", C.c_int64)]


class Problem(C.Structure):
    _fields_ = [("J", C.c_void_p), ("J_stride", C.c_int64), ("J_ld", C.c_int32), ("J_layout", C.c_int32),
                ("r", C.c_void_p), ("r_stride", C.c_int64),
                ("lam", C.c_double),
                ("G", C.c_void_p), ("G_stride", C.c_int64), ("G_ld", C.c_int32), ("reserved0", C.c_int32),
                ("c", C.c_void_p), ("c_stride", C.c_int64),
                ("A_eq", C.c_void_p), ("A_stride", C.c_int64), ("A_ld", C.c_int32), ("reserved1", C.c_int32),
                ("b_eq", C.c_void_p), ("b_stride", C.c_int64),
                ("cons_var", C.c_void_p), ("cons_a", C.c_void_p), ("cons_b", C.c_void_p), ("cons_stride", C.c_int64),
                ("lambda_vec", C.c_void_p), ("lambda_stride", C.c_int64)]


class SolveParams(C.Structure):
    _fields_ = [("initial_mu", C.c_double), ("sigma", C.c_double), ("termination_kkt_tol", C.c_double),
                ("termination_complementarity_tol", C.c_double), ("max_iterations", C.c_int32),
                ("barrier_strategy", C.c_int32), ("decrease_mu_only_on_small_error", C.c_int32),
                ("initial_guess_method", C.c_int32), ("initialize_mu_with_complementarity", C.c_int32),
                ("reserved", C.c_int32)]


class NlsParams(C.Structure):
    _fields_ = [("max_iterations", C.c_int32), ("max_qp_iterations", C.c_int32), ("termination_kkt_tolerance", C.c_double),
                ("absolute_exit_tol", C.c_double), ("relative_exit_tol", C.c_double),
                ("absolute_first_derivative_tol", C.c_double), ("max_line_search_iterations", C.c_int32),
                ("line_search_strategy", C.c_int32), ("armijo_search_tau", C.c_double),
                ("equality_penalty_initial", C.c_double), ("equality_penalty_scale_factor", C.c_double),
                ("equality_penalty_rho", C.c_double), ("lambda_initial", C.c_double), ("lambda_failure_init", C.c_double),
                ("lambda_decrease_on_success", C.c_double), ("lambda_decrease_on_restore", C.c_double),
                ("max_lambda", C.c_double), ("min_lambda", C.c_double), ("retraction", C.c_int32), ("log_qp_eigenvalues", C.c_int32)]


class NlsProblem(C.Structure):
    _fields_ = [("vars", C.c_void_p), ("vars_stride", C.c_int64), ("candidate", C.c_void_p), ("candidate_stride", C.c_int64),
                ("J", C.c_void_p), ("J_stride", C.c_int64), ("J_ld", C.c_int32), ("J_layout", C.c_int32),
                ("r", C.c_void_p), ("r_stride", C.c_int64),
                ("J_eq", C.c_void_p), ("J_eq_stride", C.c_int64), ("J_eq_ld", C.c_int32), ("reserved0", C.c_int32),
                ("r_eq", C.c_void_p), ("r_eq_stride", C.c_int64),
                ("r_cand", C.c_void_p), ("r_cand_stride", C.c_int64), ("r_eq_cand", C.c_void_p), ("r_eq_cand_stride", C.c_int64),
                ("cons_var", C.c_void_p), ("cons_a", C.c_void_p), ("cons_b", C.c_void_p), ("cons_stride", C.c_int64),
                ("step", C.c_void_p), ("step_stride", C.c_int64), ("step_alpha", C.c_void_p), ("user_exit", C.c_void_p),
                ("qp_iterations", C.c_void_p), ("qp_lagrange", C.c_void_p), ("qp_eigenvalues", C.c_void_p)]


NLS_EVAL_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int32, C.c_void_p)
MO_NLS_EVAL_LINEARIZE, MO_NLS_EVAL_ERRORS, MO_NLS_EVAL_RETRACT, MO_NLS_EVAL_ITERATION_DONE = 0, 1, 2, 3
MO_RETRACT_EUCLIDEAN, MO_RETRACT_WRAP_PI, MO_RETRACT_CALLBACK = 0, 1, 2
MO_NLS_ITER_HEADER = 12
MO_RESIDUAL_ROSENBROCK, MO_RESIDUAL_HIMMELBLAU, MO_RESIDUAL_SPHERE, MO_RESIDUAL_PRODUCT_PAIRS, MO_RESIDUAL_ACTUATOR_CHAIN = range(5)


def build(force: bool = False) -> str:
    """Compile the HIP extension in-tree for gfx950 (hipcc cross-compiles without a GPU)."""
    srcs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h"))]
    srcs.append(os.path.join(_HERE, "..", "include", "mini_opt_hip.h"))
    srcs.append(os.path.join(CSRC, "Makefile"))
    srcs.append(os.path.join(_HERE, "..", "tools", "isa_lint.py"))   # the build lints its own listings: a changed lint re-runs it
    stale = (not os.path.exists(LIB_PATH)) or any(os.path.getmtime(s) > os.path.getmtime(LIB_PATH) for s in srcs)
    if force or stale:
        jobs = max(2, min(8, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 4)))   # (eleven units, the longest 5.7 min)
        subprocess.check_call(["make", "-C", CSRC, f"-j{jobs}"], stdout=subprocess.DEVNULL)
    return LIB_PATH


_lib = None


def lib() -> C.CDLL:
    """Load libminiopt_hip.so; raises if the HIP extension has not been built (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: build the HIP extension first (python -c 'import __graft_entry__ as g; g.build()'). "
            "mini_opt_amd has no CPU fallback.")
    L = C.CDLL(LIB_PATH)
    vp, i64, i32, u32, dbl = C.c_void_p, C.c_int64, C.c_int32, C.c_uint32, C.c_double
    L.mo_version_string.restype = C.c_char_p
    L.mo_status_string.restype = C.c_char_p
    L.mo_status_string.argtypes = [i32]
    L.mo_last_error.restype = C.c_char_p
    L.mo_default_solve_params.argtypes = [C.POINTER(SolveParams)]
    L.mo_default_solve_params.restype = None
    L.mo_plan_create.argtypes = [C.POINTER(PlanDesc), C.POINTER(vp)]
    L.mo_plan_destroy.argtypes = [vp]
    L.mo_plan_step_kernel.argtypes = [vp, C.POINTER(Problem)]
    L.mo_plan_step_kernel.restype = C.c_char_p
    L.mo_plan_solve_kernel.argtypes = [vp, C.POINTER(Problem)]
    L.mo_plan_solve_kernel.restype = C.c_char_p
    L.mo_plan_nls_uses_nullspace.argtypes = [vp]
    L.mo_plan_nls_uses_nullspace.restype = C.c_int
    L.mo_linearize.argtypes = [vp, C.POINTER(Problem), i64, vp, i64, i32, vp, i64, vp, vp]
    L.mo_kkt_residual.argtypes = [vp, C.POINTER(Problem), i64, vp, i64, vp, i64, u32, vp, i64, vp, vp]
    L.mo_newton_step.argtypes = [vp, C.POINTER(Problem), i64, vp, i64, vp, i64, dbl, u32, vp, i64, vp, vp, vp]
    L.mo_iterate.argtypes = [vp, C.POINTER(Problem), i64, vp, i64, vp, i64, i32, vp, i64, vp, vp, vp]
    L.mo_qp_solve.argtypes = [vp, C.POINTER(Problem), i64, C.POINTER(SolveParams), vp, i64, vp, vp, vp, vp, vp, vp]
    L.mo_fill_qp.argtypes = [vp, C.POINTER(Problem), i64, vp, i64, vp, i64, i32, vp, i64, vp, i64, vp, vp, vp]
    L.mo_nonlinear_errors.argtypes = [vp, vp, i64, vp, i64, i64, vp, vp]
    L.mo_qp_cost_derivative.argtypes = [vp, C.POINTER(Problem), i64, vp, i64, vp, vp, vp]
    L.mo_nullspace_solve.argtypes = [vp, C.POINTER(Problem), i64, vp, i64, vp, vp]
    L.mo_residual_eval.argtypes = [vp, i32, i32, vp, vp, i64, i64, vp, i64, vp, i64, i32, i32, vp]
    L.mo_qp_eigenvalue_stats.argtypes = [vp, C.POINTER(Problem), i64, vp, vp]
    L.mo_default_nls_params.argtypes = [C.POINTER(NlsParams)]
    L.mo_default_nls_params.restype = None
    L.mo_nls_solve.argtypes = [vp, C.POINTER(NlsProblem), i64, C.POINTER(NlsParams), NLS_EVAL_FN, vp, vp, vp, vp, vp, vp]
    for name in EXPORTS:
        getattr(L, name)  # AttributeError if the library does not export what the header declares
    _lib = L
    return L


class MiniOptError(RuntimeError):
    """Argument / dimension errors of the C ABI (the reference throws assert::default_error, assertions.hpp:49-59)."""

    def __init__(self, code: int, msg: str):
        super().__init__(f"mini_opt_hip error {code}: {msg}")
        self.code = code


def check(rc: int) -> None:
    if rc != MO_OK:
        raise MiniOptError(rc, lib().mo_last_error().decode())
