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
    big = L.PlanDesc(20000, 0, 0, 0, L.MO_F64, 0, 0, 0, 1)            # the state / residual vectors alone exceed the LDS: refused before any device query
    assert lib.mo_plan_create(C.byref(big), C.byref(plan)) == -3
    assert b"LDS" in lib.mo_last_error()
    sizeable = L.PlanDesc(512, 40, 128, 0, L.MO_F64, 0, 0, 0, 1)       # beyond every LDS-resident kernel, served with H in a global workspace: not a size error
    assert lib.mo_plan_create(C.byref(sizeable), C.byref(plan)) != -3
    if plan.value:
        lib.mo_plan_destroy(plan)
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


def test_only_tests_smoke_and_the_cpu_baseline_leg_touch_the_oracle():
    """tools/ never imports the oracle; bench.py does so only BEHIND its timed region, as the checker of the launch just timed (every rank,
    on a sample of its own shard) and as the timed CPU baseline (rank 0 at N = 1) -- both skipped by --no-cpu-baseline."""
    import re
    for dirpath, _, files in os.walk(os.path.join(ROOT, "tools")):
        for f in files:
            if f.endswith((".py", ".sh", ".hip")):
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                assert not re.search(r"(from|import)\s+oracle", txt), (dirpath, f)
    bench = open(os.path.join(ROOT, "bench.py")).read()
    assert bench.count("from oracle import") == 2   # "from oracle import oracle as orc": the checker leg and the cpu_baseline leg
    timed_region_end = bench.index("elapsed = time.perf_counter() - t0")
    checker = bench.index("from oracle import oracle")
    baseline = bench.index("from oracle import oracle", checker + 1)
    assert timed_region_end < checker < baseline                       # never inside what is measured
    assert bench.index("if not args.dry and not args.no_cpu_baseline:") < checker
    assert checker < bench.index("if info.world_size > 1 or args.no_cpu_baseline:") < baseline


def test_ctypes_mirrors_match_the_header_layout(tmp_path):
    """sizeof / offsetof of every struct of include/mini_opt_hip.h, as gcc sees them, against the ctypes mirrors in
    mini_opt_amd/_lib.py (a silent mismatch would shift every pointer that follows)."""
    import ctypes
    import subprocess
    from mini_opt_amd import _lib as L
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
    src = tmp_path / "abi.c"
    checks = {
        "mo_plan_desc": (L.PlanDesc, ["n", "m_r", "flags", "max_batch"]),
        "mo_problem": (L.Problem, ["J", "r", "lambda", "G", "c", "A_eq", "b_eq", "cons_var", "cons_stride", "lambda_vec", "lambda_stride"]),
        "mo_solve_params": (L.SolveParams, ["initial_mu", "max_iterations", "initial_guess_method", "initialize_mu_with_complementarity"]),
        "mo_nls_params": (L.NlsParams, ["max_iterations", "termination_kkt_tolerance", "max_line_search_iterations", "armijo_search_tau",
                                        "lambda_initial", "min_lambda"]),
        "mo_nls_problem": (L.NlsProblem, ["vars", "candidate", "J", "J_ld", "r", "J_eq", "J_eq_ld", "r_eq", "r_cand", "r_eq_cand", "cons_var",
                                          "cons_stride"]),
    }
    rename = {"lambda": "lam"}  # python keyword
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "mini_opt_hip.h"', "int main(void) {"]
    for cname, (_, fields) in checks.items():
        lines.append(f'  printf("{cname} %zu\\n", sizeof({cname}));')
        for f in fields:
            lines.append(f'  printf("{cname}.{f} %zu\\n", offsetof({cname}, {f}));')
    lines += ["  return 0;", "}"]
    src.write_text("\n".join(lines))
    exe = tmp_path / "abi"
    subprocess.check_call(["gcc", "-I", os.path.join(root, "include"), str(src), "-o", str(exe)])
    out = dict(line.split() for line in subprocess.check_output([str(exe)]).decode().splitlines())
    for cname, (cls, fields) in checks.items():
        assert int(out[cname]) == ctypes.sizeof(cls), cname
        for f in fields:
            assert int(out[f"{cname}.{f}"]) == getattr(cls, rename.get(f, f)).offset, (cname, f)


def test_isa_lint_flags_the_copy_in_front_of_an_exec_restore():
    """tools/isa_lint.py on a fixture of OUR OWN compiler output (tests/golden/isa_defect_block.s: the loop-exit block of the pre-fix generic
    Solve kernel whose `v_accvgpr_write_b32 a85, v166` runs with EXEC = 0) and on the same block with the copy behind
    the restore."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("isa_lint", os.path.join(ROOT, "tools", "isa_lint.py"))
    lint = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(lint)
    bad = open(os.path.join(ROOT, "tests", "golden", "isa_defect_block.s")).read()
    hits = lint.lint_text(bad)
    assert len(hits) == 1 and "v_accvgpr_write_b32 a85, v166" in hits[0][3]
    good = bad.replace("\tv_accvgpr_write_b32 a85, v166\n", "").replace("\ts_or_b64 exec, exec, s[0:1]\n", "\ts_or_b64 exec, exec, s[0:1]\n\tv_accvgpr_write_b32 a85, v166\n")
    assert "v_accvgpr_write_b32" in good and lint.lint_text(good) == []


def test_isa_lint_flags_an_access_to_the_result_of_an_inline_assembly_mfma():
    """tools/isa_lint.py, second check: hipcc cannot see an MFMA inside inline assembly, so a spill of its VGPR result right behind it (what
    the 128-grid kernels with a second y tile got) carries no wait states.  Hand-written listing fragments."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("isa_lint", os.path.join(ROOT, "tools", "isa_lint.py"))
    lint = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(lint)
    asm_mfma = "\t;;#ASMSTART\n\ts_nop 3\n\tv_mfma_f64_16x16x4_f64 v[40:47], v[2:3], v[4:5], v[40:47]\n\t;;#ASMEND\n"
    spill = "\tscratch_store_dwordx4 off, v[44:47], off offset:16 ; 16-byte Folded Spill\n"
    other = "\tv_mfma_f64_16x16x4_f64 a[0:7], v[2:3], v[4:5], a[0:7]\n\tscratch_store_dwordx4 off, v[48:51], off offset:32\n"
    head = "kernel_a:\n"
    hits = lint.lint_asm_mfma(head + asm_mfma + other + spill)
    assert len(hits) == 1 and "v[44:47]" in hits[0][3] and hits[0][2] == "kernel_a"
    assert lint.lint_asm_mfma(head + asm_mfma + other + "\ts_nop 15\n\ts_nop 3\n" + spill) == []          # the wait states are there
    assert lint.lint_asm_mfma(head + asm_mfma.replace(";;#ASMSTART", "").replace(";;#ASMEND", "") + spill) == []  # a builtin MFMA: hipcc's business
    assert lint.lint_asm_mfma(head + asm_mfma + "\tv_mfma_f64_16x16x4_f64 a[8:15], v[40:41], v[4:5], a[8:15]\n") == []  # MFMAs may read it
