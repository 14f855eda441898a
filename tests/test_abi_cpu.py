"""CPU-side checks of the C-ABI library: it loads, exports every symbol include/mini_opt_hip.h declares, and refuses
to run without a GPU (no CPU fallback).  No compute calls here."""
import ctypes as C
import os
import re

import pytest

from mini_opt_amd import _lib as L

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


@pytest.fixture(scope="module")
def lib():
    L.build()
    return L.lib()


def test_header_symbols_are_exported(lib):
    header = open(os.path.join(ROOT, "include", "mini_opt_hip.h")).read()
    declared = set(re.findall(r"\b(mo_[a-z_]+)\s*\(", header))
    assert declared == set(L.EXPORTS), declared ^ set(L.EXPORTS)
    for name in declared:
        assert hasattr(lib, name)


def test_version_and_status_strings(lib):
    assert b"gfx950" in lib.mo_version_string()
    assert lib.mo_status_string(0) == b"OK"
    assert lib.mo_status_string(2) == b"FACTORIZATION_FAILED"


def test_default_params_match_reference(lib):
    p = L.SolveParams()
    lib.mo_default_solve_params(C.byref(p))
    # qp.hpp:134-164
    assert (p.initial_mu, p.sigma, p.termination_kkt_tol, p.termination_complementarity_tol) == (1.0, 0.5, 1e-9, 1e-6)
    assert (p.max_iterations, p.barrier_strategy, p.decrease_mu_only_on_small_error, p.initial_guess_method,
            p.initialize_mu_with_complementarity) == (10, 0, 0, 0, 0)


def test_argument_errors_without_gpu(lib):
    plan = C.c_void_p()
    assert lib.mo_plan_create(None, C.byref(plan)) == -1
    bad = L.PlanDesc(0, 0, 0, 0, L.MO_F64, 0, 0, 0, 1)
    assert lib.mo_plan_create(C.byref(bad), C.byref(plan)) == -2
    assert b"bad dimensions" in lib.mo_last_error()
    big = L.PlanDesc(512, 0, 0, 0, L.MO_F64, 0, 0, 0, 1)
    assert lib.mo_plan_create(C.byref(big), C.byref(plan)) == -3
    assert lib.mo_newton_step(None, None, 0, None, 0, None, 0, 0.995, 0, None, 0, None, None, None) == -1


def test_no_cpu_fallback(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    plan = C.c_void_p()
    ok = L.PlanDesc(8, 2, 4, 16, L.MO_F64, 0, 0, 0, 1)
    rc = lib.mo_plan_create(C.byref(ok), C.byref(plan))
    assert rc in (-5, -4), rc  # MO_ERR_NO_DEVICE (or a HIP error): the product path fails loudly
    assert plan.value is None


def test_product_never_imports_oracle():
    """The oracle is test infrastructure: nothing under mini_opt_amd/ or include/ may reference it."""
    for base in ("mini_opt_amd", "include"):
        for dirpath, _, files in os.walk(os.path.join(ROOT, base)):
            for f in files:
                if f.endswith((".py", ".hip", ".h", ".hpp", ".cpp", "Makefile")):
                    txt = open(os.path.join(dirpath, f), errors="ignore").read()
                    assert "oracle" not in txt.lower() or f in (), (dirpath, f)
