"""The CPU sanitizer job SURVEY.md section 5 asks for (GPU AddressSanitizer is not available on the pool): the oracle's plain-C restatement
built with -fsanitize=address,undefined (oracle/Makefile: libkkt_oracle_asan.so) runs the reference's differential elimination cases, its
known-answer tests and the batched OpenMP entry points in a child process; any out-of-bounds access, use-after-free or undefined behaviour in
kkt_oracle.c aborts the child and fails this test."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_oracle_under_address_and_ub_sanitizer():
    asan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    ubsan = subprocess.run(["gcc", "-print-file-name=libubsan.so"], capture_output=True, text=True).stdout.strip()
    assert os.path.isabs(asan) and os.path.exists(asan), asan
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "libkkt_oracle_asan.so"], stdout=subprocess.DEVNULL)
    env = dict(os.environ, MO_ORACLE_LIB=os.path.join(ROOT, "oracle", "libkkt_oracle_asan.so"), LD_PRELOAD=asan + ":" + ubsan,
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=1:halt_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1",
               OMP_NUM_THREADS="2")
    select = "elimination or no_inequalities or compute_alpha or full_solve_kats or update_hessian or (synthetic and cfg1) or (synthetic and cfg2)"
    res = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_oracle_golden.py"), "-x", "-q", "-k", select,
                          "-p", "no:cacheprovider"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, (res.stdout[-3000:], res.stderr[-3000:])
    assert " passed" in res.stdout and "AddressSanitizer" not in res.stderr and "runtime error" not in res.stderr, (res.stdout[-1500:], res.stderr[-3000:])
    # the batched OpenMP entry points (bench.py's checker / cpu_baseline legs) under the sanitizer as well
    code = ("import numpy as np\n"
            "from oracle import oracle as orc\n"
            "from mini_opt_amd import synth\n"
            "d = synth.CONFIGS['cfg2']\n"
            "hb = synth.make_batch(d['n'], d['k'], d['m'], d['m_r'], 12, stream=3)\n"
            "pr = dict(J=hb.J, r=hb.r, lam=hb.lam, A_eq=hb.A_eq, b_eq=hb.b_eq, cons_var=hb.cons_var, cons_a=hb.cons_a, cons_b=hb.cons_b)\n"
            "dl, al, st, _ = orc.batched_newton_step(hb.n, hb.k, hb.m, vars_=hb.vars, mu=hb.mu, **pr)\n"
            "assert np.all(st == 0) and np.all(np.isfinite(dl))\n"
            "t, n_it, v, _ = orc.batched_solve(hb.n, hb.k, hb.m, initial_mu=1.0, sigma=0.1, termination_kkt_tol=1e-8, max_iterations=10, **pr)\n"
            "assert np.all(t == 0) and np.all(n_it > 0), (t, n_it)\n"
            "print('batched ok')\n")
    res = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0 and "batched ok" in res.stdout, (res.stdout[-1000:], res.stderr[-3000:])
