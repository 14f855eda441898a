"""Randomised sweep over (n, k, m, m_r) x the layouts of J the C ABI accepts (packed, a leading dimension beyond n, column-major with
and without padding, rows that are only 8-byte aligned): every case must run on the fused kernels, the step must match the oracle to
1e-9 and the Solve must follow the packed-layout Solve."""
import numpy as np
import pytest
import torch

from mini_opt_amd import qp as Q
from oracle import oracle as orc

import os

pytestmark = pytest.mark.gpu
# MO_FUZZ_EXTRA_SEEDS="100-140" widens both sweeps for a one-off soak run (not part of the default suite)
_extra = os.environ.get("MO_FUZZ_EXTRA_SEEDS", "")
EXTRA_SEEDS = list(range(int(_extra.split("-")[0]), int(_extra.split("-")[1]) + 1)) if _extra else []


def T(a, dt=torch.float64):
    return torch.as_tensor(np.ascontiguousarray(a), dtype=dt, device="cuda:0")


@pytest.mark.parametrize("seed", [1, 2, 3] + EXTRA_SEEDS)
def test_random_shapes_and_layouts(seed):
    rng = np.random.default_rng(seed)
    worst = 0.0
    for trial in range(40):
        n = int(rng.integers(2, 129)); k = int(rng.integers(0, min(32, n + 1))); m = int(rng.integers(0, 65)); m_r = int(rng.integers(1, 300)); B = 5
        J = rng.uniform(-1, 1, (B, m_r, n)); r = rng.uniform(-1, 1, (B, m_r))
        A = rng.uniform(-1, 1, (B, n, k)); b = rng.uniform(-1, 1, (B, k))
        cv = rng.integers(0, n, (B, m)).astype(np.int32); ca = rng.choice([-1.0, 1.0, 2.0], (B, m)); cb = rng.uniform(0.5, 2.0, (B, m))
        x = rng.uniform(-0.1, 0.1, (B, n)); sl = rng.uniform(0.2, 1.5, (B, m)); z = rng.uniform(0.1, 2, (B, m)); y = rng.uniform(-1, 1, (B, k))
        vars_ = np.concatenate([x, sl, y, z], axis=1); mu = np.full(B, 0.05)
        lam = 0.3 if m_r < n else 1e-3
        ref, ref_alpha, ref_status, _ = orc.batched_newton_step(n, k, m, J=J, r=r, lam=lam, A_eq=A if k else None, b_eq=b if k else None,
            cons_var=cv if m else None, cons_a=ca if m else None, cons_b=cb if m else None, vars_=vars_, mu=mu)
        which = int(rng.integers(0, 4))
        kw = {}
        if which == 0: Jt = T(J)
        elif which == 1:
            pad = int(rng.integers(1, 9)); wide = np.full((B, m_r, n + pad), 9.9); wide[:, :, :n] = J; Jt = T(wide)
        elif which == 2:
            pad = int(rng.integers(0, 5)); colw = np.full((B, n, m_r + pad), -9.9); colw[:, :, :m_r] = J.transpose(0, 2, 1); Jt = T(colw); kw = dict(J_layout="col", J_rows=m_r)
        else:
            flat = torch.zeros(B * m_r * n + 1, dtype=torch.float64, device="cuda:0"); flat[1:] = T(J).reshape(-1); Jt = flat[1:].view(B, m_r, n)
        prob = Q.BatchedQP(n=n, k=k, m=m, J=Jt, r=T(r), lam=lam, A_eq=T(A) if k else None, b_eq=T(b) if k else None,
                           cons_var=T(cv, torch.int32) if m else None, cons_a=T(ca) if m else None, cons_b=T(cb) if m else None, **kw)
        s = Q.QPInteriorPointSolver(prob)
        assert s.step_kernel().startswith("fused"), (n, k, m, m_r, which, s.step_kernel())
        s.SetVariables(T(vars_))
        delta, alpha, status = s.NewtonStep(T(mu), 0.995)
        assert torch.all(status == 0) and np.all(ref_status == 0), (n, k, m, m_r, which)
        err = (np.abs(delta.cpu().numpy() - ref).max(axis=1) / np.abs(ref).max(axis=1)).max()
        worst = max(worst, err)
        assert err < 1e-9, (n, k, m, m_r, which, err)
        out = s.Solve(Q.Params(initial_mu=1.0, sigma=0.1, termination_kkt_tol=1e-8, max_iterations=10, initial_guess_method=Q.SOLVE_EQUALITY_CONSTRAINED if k else Q.NAIVE))
        assert torch.all(out.status == 0), (n, k, m, m_r, which)
        sg = Q.QPInteriorPointSolver(Q.BatchedQP(n=n, k=k, m=m, J=T(J), r=T(r), lam=lam, A_eq=T(A) if k else None, b_eq=T(b) if k else None,
                           cons_var=T(cv, torch.int32) if m else None, cons_a=T(ca) if m else None, cons_b=T(cb) if m else None))
        og = sg.Solve(Q.Params(initial_mu=1.0, sigma=0.1, termination_kkt_tol=1e-8, max_iterations=10, initial_guess_method=Q.SOLVE_EQUALITY_CONSTRAINED if k else Q.NAIVE))
        # every stream feeds the matrix cores the same operands in the same order: the layouts are bit-compatible, and so is the whole Solve
        assert torch.equal(out.num_iterations, og.num_iterations) and torch.equal(out.termination_state, og.termination_state), (n, k, m, m_r, which)
        assert torch.equal(s.variables(), sg.variables()), (n, k, m, m_r, which)


@pytest.mark.parametrize("seed", [11, 12] + EXTRA_SEEDS)
def test_random_shapes_solve_and_iterate_against_the_oracle(seed):
    """Random (n, k, m, m_r, input level, barrier strategy, initial guess) through `Solve` and `Iterate` on BOTH kernel families --
    fused and (where the shape fits its LDS) generic -- with the oracle as the referee for each: the Iterate records (mu, step lengths,
    probe step lengths, mu_affine) and the state after it, and for Solve the termination state, the iteration count and the optimum.
    Problems are built around a strictly feasible point with small margins so that some inequalities are active at the optimum."""
    rng = np.random.default_rng(seed)
    from oracle import margins as M
    checked = {"fused": 0, "generic": 0}
    dis = {"fused": M.Disagreements(f"fuzz seed {seed}, fused Solve"), "generic": M.Disagreements(f"fuzz seed {seed}, generic Solve")}
    for trial in range(30):
        n = int(rng.integers(2, 129)); B = 4
        # well-posed problems: at most n / 2 equalities and 2 n inequality entries.  (With k close to n or m >> n the start has slacks on the
        # 1e-9 floor, z / s = 1e18, and the reduced KKT matrix is numerically singular: no double-precision formulation -- the reference's
        # included -- gets the first step right there; tests/test_gpu_stress.py::test_first_step_from_an_overconstrained_naive_start measures
        # every implementation against a long-double truth on exactly those shapes.)
        k = int(rng.integers(0, min(32, n // 2 + 1))); m = int(rng.integers(0, min(65, 2 * n + 1))); m_r = int(rng.integers(n, 2 * n + 8))
        level = rng.choice(["J", "QP"]); strategy = int(rng.integers(0, 3))
        J = rng.uniform(-1, 1, (B, m_r, n)); r = rng.uniform(-1, 1, (B, m_r))
        A = rng.uniform(-1, 1, (B, n, k))
        cv = rng.integers(0, n, (B, m)).astype(np.int32); ca = rng.choice([-1.0, 1.0, 2.0], (B, m))
        x0 = rng.uniform(-0.5, 0.5, (B, n)); b = -np.einsum("bik,bi->bk", A, x0)
        cb = -ca * np.take_along_axis(x0, cv.astype(np.int64), axis=1) + rng.uniform(0.05, 0.5, (B, m))
        lam = 1e-2
        G = np.einsum("bqi,bqj->bij", J, J) + lam * np.eye(n); c = np.einsum("bqi,bq->bi", J, r)
        x = rng.uniform(-0.1, 0.1, (B, n)); sl = rng.uniform(0.2, 1.5, (B, m)); z = rng.uniform(0.1, 2, (B, m)); y = rng.uniform(-1, 1, (B, k))
        vars_ = np.concatenate([x, sl, y, z], axis=1)
        common = dict(A_eq=T(A) if k else None, b_eq=T(b) if k else None, cons_var=T(cv, torch.int32) if m else None,
                      cons_a=T(ca) if m else None, cons_b=T(cb) if m else None)
        prob = (Q.BatchedQP(n=n, k=k, m=m, J=T(J), r=T(r), lam=lam, **common) if level == "J"
                else Q.BatchedQP(n=n, k=k, m=m, G=T(np.tril(G).transpose(0, 2, 1)), c=T(c), **common))
        guess = Q.SOLVE_EQUALITY_CONSTRAINED if (k and rng.integers(0, 2)) else Q.NAIVE
        kw = dict(initial_mu=1.0, sigma=0.1, termination_kkt_tol=1e-6, max_iterations=12, barrier_strategy=strategy, initial_guess_method=guess)
        # the referee
        ref_it, ref_solve = [], []
        for p in range(B):
            qp = orc.QP(G=np.tril(G[p]), c=c[p], A_eq=A[p].T if k else None, b_eq=b[p] if k else None, cons_var=cv[p], cons_a=ca[p], cons_b=cb[p])
            o = orc.Solver(qp)
            o.variables[:] = vars_[p]
            st, ip = o.iterate(0.05, strategy)
            assert st == 0
            ref_it.append(([ip.mu, ip.alpha_primal, ip.alpha_dual, ip.alpha_probe_primal, ip.alpha_probe_dual, ip.mu_affine], o.variables.copy()))
            o2 = orc.Solver(qp)
            term, its = o2.solve(**kw)
            ref_solve.append((term, len(its), o2.variables.copy(), M.solve_with_margins(qp, **kw)[3]))
        tag = (seed, trial, level, n, k, m, m_r, strategy, guess)
        for family, force in (("fused", False), ("generic", True)):
            s = Q.QPInteriorPointSolver(prob, force_generic=force)   # (the generic kernel takes every size: H in a global workspace beyond the LDS)
            s.SetVariables(T(vars_))
            ip, st = s.Iterate(T(np.full(B, 0.05)), strategy)
            if not force:
                assert s.solve_kernel().startswith("fused"), (tag, s.solve_kernel())
            assert torch.all(st == 0), (tag, family)
            after = s.variables().cpu().numpy()
            ipn = ip.cpu().numpy()
            for p in range(B):
                np.testing.assert_allclose(ipn[p], ref_it[p][0], rtol=1e-6, atol=1e-9, equal_nan=True, err_msg=str((tag, family, p)))
                scale = max(1.0, np.abs(ref_it[p][1]).max())
                np.testing.assert_allclose(after[p], ref_it[p][1], rtol=1e-7, atol=1e-8 * scale, err_msg=str((tag, family, p)))
            out = s.Solve(Q.Params(**kw))
            assert torch.all(out.status == 0), (tag, family)
            tm = out.termination_state.cpu().numpy(); nit = out.num_iterations.cpu().numpy(); v = s.variables().cpu().numpy()
            for p in range(B):
                # The rule (oracle/margins.py): the oracle's termination state after the oracle's iteration count, unless a decision of the
                # oracle's own run sat within rounding of its threshold (termination test, a step-length tie, the slack floor) -- a
                # disagreement anywhere else fails.
                same = bool(tm[p] == ref_solve[p][0] and nit[p] == ref_solve[p][1])
                dis[family].check((tag, p, int(tm[p]), int(nit[p]), ref_solve[p][:2]), same, ref_solve[p][3] if not same else None)
                if same and tm[p] == Q.SATISFIED_KKT_TOL:   # (a run that ends in MAX_ITERATIONS has no point to compare)
                    xs = max(1.0, np.abs(ref_solve[p][2][:n]).max())
                    assert np.abs(v[p][:n] - ref_solve[p][2][:n]).max() <= 1e-5 * xs, (tag, family, p)
            checked[family] += 1
    assert checked["fused"] == 30 and checked["generic"] == 30, checked
    os.makedirs(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out"), exist_ok=True)
    for family in ("fused", "generic"):
        print(dis[family].report())
        with open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "fuzz_soak.txt"), "a") as f:
            f.write(dis[family].report() + "\n")    # the soak record (profiles/r04_fuzz_soak.txt): every disagreement with its margin
        assert len(dis[family].items) <= 0.05 * dis[family].total, dis[family].report()


@pytest.mark.parametrize("seed", [21, 22] + EXTRA_SEEDS)
def test_random_shapes_on_the_large_path(seed):
    """Random systems of n + k >= 72 that only the generic kernel takes (more equalities than the fused kernels hold, or more than 128 / 256
    inequality entries, or n > 128) -- since round 4 all of them on the LARGE path (H in the plan's global workspace, kkt_generic.hip):
    ragged super-blocks of J^T J, block columns of every width, half panels with one to four waves owning rows, both input levels, every
    layout of J, both precisions.  Step against the oracle (1e-10 rel-inf in fp64; fp32 on fp32-rounded inputs at its tolerance), the KKT
    residual, and in fp64 the whole Solve: (termination, iterations) under the knife-edge rule of oracle/margins.py."""
    rng = np.random.default_rng(seed)
    from oracle import margins as M
    dis = M.Disagreements(f"fuzz seed {seed}, LARGE path Solve")
    for trial in range(14):
        kind = int(rng.integers(0, 3))
        if kind == 0:      # more equalities than the fused kernels hold (since round 4: k <= 63 up to n = 96, k <= 47 up to n = 128)
            n = int(rng.integers(100, 129)); k = int(rng.integers(48, min(n // 2 + 1, 81))); m = int(rng.integers(0, 65))
        elif kind == 1:    # many inequality entries
            n = int(rng.integers(72, 129)); k = int(rng.integers(0, 17)); m = int(rng.integers(257, 400))
        else:              # beyond the tile grids
            n = int(rng.integers(129, 330)); k = int(rng.integers(0, min(n // 2, 70))); m = int(rng.integers(0, 130))
        m_r = int(rng.integers(n, n + 40)); B = 3
        f32 = bool(rng.integers(0, 4) == 0)
        dt = torch.float32 if f32 else torch.float64
        f = (lambda a_: a_.astype(np.float32).astype(np.float64)) if f32 else (lambda a_: a_)
        level = rng.choice(["J", "QP"]); layout = int(rng.integers(0, 3)) if level == "J" else 0
        J = f(rng.uniform(-1, 1, (B, m_r, n))); r = f(rng.uniform(-1, 1, (B, m_r)))
        A = f(rng.uniform(-1, 1, (B, n, k)))
        cv = rng.integers(0, n, (B, m)).astype(np.int32); ca = rng.choice([-1.0, 1.0, 2.0], (B, m))
        x0 = rng.uniform(-0.5, 0.5, (B, n)); b = f(-np.einsum("bik,bi->bk", A, x0))
        cb = f(-ca * np.take_along_axis(x0, cv.astype(np.int64), axis=1) + rng.uniform(0.05, 0.5, (B, m)))
        lam = float(np.float32(0.05))
        G = f(np.einsum("bqi,bqj->bij", J, J) + lam * np.eye(n)); c = f(np.einsum("bqi,bq->bi", J, r))
        x = f(rng.uniform(-0.1, 0.1, (B, n))); sl = f(rng.uniform(0.2, 1.5, (B, m))); z = f(rng.uniform(0.1, 2, (B, m))); y = f(rng.uniform(-1, 1, (B, k)))
        vars_ = np.concatenate([x, sl, y, z], axis=1); mu = np.full(B, float(np.float32(0.05)))
        common = dict(A_eq=T(A, dt) if k else None, b_eq=T(b, dt) if k else None, cons_var=T(cv, torch.int32) if m else None,
                      cons_a=T(ca, dt) if m else None, cons_b=T(cb, dt) if m else None)
        kwj = {}
        if level == "J":
            if layout == 0: Jt = T(J, dt)
            elif layout == 1:
                wide = np.full((B, m_r, n + 3), 9.9); wide[:, :, :n] = J; Jt = T(wide, dt)
            else:
                colw = np.full((B, n, m_r + 2), -9.9); colw[:, :, :m_r] = J.transpose(0, 2, 1); Jt = T(colw, dt); kwj = dict(J_layout="col", J_rows=m_r)
            prob = Q.BatchedQP(n=n, k=k, m=m, J=Jt, r=T(r, dt), lam=lam, **common, **kwj)
        else:
            prob = Q.BatchedQP(n=n, k=k, m=m, G=T(np.tril(G).transpose(0, 2, 1), dt), c=T(c, dt), **common)
        tag = (seed, trial, n, k, m, m_r, str(level), layout, "f32" if f32 else "f64")
        s = Q.QPInteriorPointSolver(prob)
        assert s.step_kernel() == "generic", (tag, s.step_kernel())
        s.SetVariables(T(vars_, dt))
        delta, alpha, status = s.NewtonStep(T(mu, dt), 0.995)
        assert torch.all(status == 0), tag
        # the oracle works on the matrix the device forms: J^T J + lambda I of the (rounded) J for J-level input, the given G otherwise
        Gl = np.stack([np.tril(J[p].T @ J[p] + lam * np.eye(n)) for p in range(B)]) if level == "J" else np.tril(G)
        cl = np.einsum("bqi,bq->bi", J, r) if level == "J" else c
        qps = []
        for p in range(B):
            qp = orc.QP(G=Gl[p], c=cl[p], A_eq=A[p].T if k else None, b_eq=b[p] if k else None, cons_var=cv[p], cons_a=ca[p], cons_b=cb[p])
            qps.append(qp)
            o = orc.Solver(qp)
            st, d_ref, a_ref = o.newton_step(vars_[p], mu[p] if m else 0.0, 0.995, True)
            assert st == 0, tag
            got = delta[p].double().cpu().numpy()
            err = np.abs(got - d_ref).max() / np.abs(d_ref).max()
            assert err < (1e-10 if not f32 else 2e-3), (tag, p, err)
        if f32:
            continue
        kw = dict(initial_mu=1.0, sigma=0.1, termination_kkt_tol=1e-8, max_iterations=12,
                  initial_guess_method=Q.SOLVE_EQUALITY_CONSTRAINED if k else Q.NAIVE)
        out = s.Solve(Q.Params(**kw))
        assert torch.all(out.status == 0), tag
        for p in range(B):
            term, n_it, _, marg = M.solve_with_margins(qps[p], **kw)
            same = int(out.termination_state[p]) == term and int(out.num_iterations[p]) == n_it
            dis.check((tag, p, int(out.termination_state[p]), int(out.num_iterations[p]), (term, n_it)), same, None if same else marg)
    print(dis.report())
    os.makedirs(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out"), exist_ok=True)
    with open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "fuzz_soak.txt"), "a") as fh:
        fh.write(dis.report() + "\n")
    assert len(dis.items) <= max(1, 0.05 * dis.total), dis.report()
