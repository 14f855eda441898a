"""GPU parity tests (run with -m gpu on an MI355X): the HIP path, called through the C ABI, against
 (a) the committed golden fixtures (the reference's own differential tests / KATs), and
 (b) the CPU oracle on identical seeded inputs.
Tolerances: fp64 direction within 1e-10 rel-inf of the reference arithmetic (BASELINE.json); the reference's own
1e-12 absolute bound is used on its small elimination cases; cfg 4 (fp32, no reference counterpart) 2e-3 rel-inf
against the fp64 oracle on fp32-rounded inputs."""
import json
import os

import numpy as np
import pytest
import torch

from mini_opt_amd import _lib as L
from mini_opt_amd import qp as Q
from mini_opt_amd import synth
from oracle import oracle as orc

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
TOL64 = 1e-10
TOL32 = 2e-3


def dev():
    return torch.device("cuda:0")


def load(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


def T(a, dt=torch.float64):
    return torch.as_tensor(np.ascontiguousarray(a), dtype=dt, device=dev()).contiguous()


def qp_from_case(case, dt=torch.float64):
    """QP-level problem (batch 1) from a JSON fixture; G/A are passed column-major as the ABI expects."""
    n, k, m = case["n"], case["k"], case["m"]
    G = np.array(case["G"]).reshape(n, n)
    A = np.array(case["A_eq"]).reshape(k, n)
    cons = case["cons"]
    return Q.BatchedQP(
        n=n, k=k, m=m, G=T(np.tril(G).T[None], dt), c=T(np.array(case["c"])[None], dt),
        A_eq=T(A.T[None], dt) if k else None, b_eq=T(np.array(case["b_eq"])[None], dt) if k else None,
        cons_var=T(np.array([c[0] for c in cons], dtype=np.int32)[None], torch.int32) if m else None,
        cons_a=T(np.array([c[1] for c in cons])[None], dt) if m else None,
        cons_b=T(np.array([c[2] for c in cons])[None], dt) if m else None)


def batch_to_device(hb, dt=torch.float64):
    return Q.BatchedQP(n=hb.n, k=hb.k, m=hb.m, J=T(hb.J, dt), r=T(hb.r, dt), lam=hb.lam, A_eq=T(hb.A_eq, dt),
                       b_eq=T(hb.b_eq, dt), cons_var=T(hb.cons_var, torch.int32), cons_a=T(hb.cons_a, dt),
                       cons_b=T(hb.cons_b, dt))


# Disagreements with the oracle are counted per TEST (a sweep calls solves_agree_or_knife_edge once per shape with a handful of problems):
# every one of them needs its logged knife edge, a single call may hold at most one, and over a whole test they may not exceed 5 %.
# A disagreement whose closest decision sat on a NUMERICALLY SINGULAR reduced KKT matrix -- its condition-aware threshold 16 eps cond(K) >= 1:
# rounding alone moves the Newton direction by its own size, the oracle's trajectory is one sample of noise -- is tallied apart: it cannot be
# charged to the kernel, but a test may not live on such problems either (at most a third of its problems; each is logged as "singular").
_TALLY = {"off": set(), "singular": set(), "all": set()}   # distinct problems (a test may hold several kernels to the oracle on the same problems)


@pytest.fixture(autouse=True)
def _knife_edge_budget():
    _TALLY["off"], _TALLY["singular"], _TALLY["all"] = set(), set(), set()
    yield
    off, sing, total = len(_TALLY["off"]), len(_TALLY["singular"]), len(_TALLY["all"])
    if total >= 20:
        assert off <= 0.05 * total, f"{off} of {total} problems of this test have a Solve that differs from the oracle (each on a knife edge, but more than 5 %)"
    else:
        assert off <= 1, f"{off} of {total} problems of this test have a Solve that differs from the oracle"
    assert 3 * sing <= total, f"{sing} of {total} problems of this test are excused by a numerically singular KKT matrix: the test does not pin the kernel"


KNIFE_EDGE_LOG = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "qp_disagreements.jsonl")


def solves_agree_or_knife_edge(tag, make_qp, kw, runs):
    """`runs` = {label: (termination [B], iterations [B])} of device Solves of the same problems.  EVERY run is held to the oracle's Solve
    (oracle/margins.py replays it with its decision margins): a run whose (termination, iteration count) differs from the oracle's is only
    accepted on a problem where one decision of the ORACLE's run sat nearer to its threshold than the knife-edge threshold of its kind --
    also when the oracle ends in MAX_ITERATIONS (no blanket exemption: the device must then end there too, or show the knife edge).
    Every accepted disagreement is logged to gpurun_out/qp_disagreements.jsonl; over a whole test at most 5 % of the Solves may be such
    cases (fixture _knife_edge_budget) -- those whose knife edge is a numerically singular KKT matrix (threshold 16 eps cond(K) >= 1) are
    tallied apart, at most a third of a test's problems.  Returns the mask of problems on which all runs agree with the oracle."""
    from oracle import margins as M
    labels = list(runs)
    B = len(runs[labels[0]][0])
    agree = np.ones(B, dtype=bool)
    rows, sing_idx = [], []
    for p in range(B):
        term, n_it, _, marg = M.solve_with_margins(make_qp(p), **kw)
        off = [l for l in labels if int(runs[l][0][p]) != term or int(runs[l][1][p]) != n_it]
        if not off:
            continue
        agree[p] = False
        ratio, where = M.closeness(marg)
        singular = bool(where is not None and len(where) > 3 and where[3] >= 1.0)
        rows.append({"test": str(tag), "problem": int(p), "oracle": [int(term), int(n_it)], "runs": {l: [int(runs[l][0][p]), int(runs[l][1][p])] for l in labels},
                     "closest_decision": list(where) if where else None, "margin_over_threshold": ratio, "singular": singular})
        if singular:
            sing_idx.append(int(p))
        assert ratio < 1.0, (f"{tag}: problem {p}: " + ", ".join(f"{l} ends ({int(runs[l][0][p])}, {int(runs[l][1][p])})" for l in labels)
                             + f", the oracle ({term}, {n_it}) with no decision near its threshold (closest: {where}, {ratio:.3g} x the knife-edge threshold)")
    if rows:
        os.makedirs(os.path.dirname(KNIFE_EDGE_LOG), exist_ok=True)
        with open(KNIFE_EDGE_LOG, "a") as f:
            for row in rows:
                f.write(json.dumps(row) + "\n")
    base = str(tuple(tag)[:5]) if isinstance(tag, tuple) else str(tag)   # (the budget is per test: the _knife_edge_budget fixture)
    _TALLY["all"].update((base, p) for p in range(B))
    _TALLY["off"].update((base, int(p)) for p in np.flatnonzero(~agree) if int(p) not in sing_idx)
    _TALLY["singular"].update((base, p) for p in sing_idx)
    return agree


def rel_inf_rows(got, ref):
    return np.max(np.abs(got - ref), axis=1) / np.max(np.abs(ref), axis=1)


# ------------------------------------------------------------------ reference differential tests (qp_test.cc:101-241)
@pytest.mark.parametrize("case", load("elimination.json"), ids=lambda c: c["name"])
@pytest.mark.parametrize("force_generic", [True, False])
def test_elimination_cases(case, force_generic):
    prob = qp_from_case(case)
    s = Q.QPInteriorPointSolver(prob, force_generic=force_generic)
    s.SetVariables(T(np.array(case["state"])[None]))
    r, _ = s.EvaluateKKTConditions(0.0)
    np.testing.assert_allclose(r.cpu().numpy()[0], case["expected_r"], rtol=0, atol=1e-12)
    delta, alpha, status = s.NewtonStep(0.0, 0.995)
    assert int(status[0]) == 0
    np.testing.assert_allclose(delta.cpu().numpy()[0], case["expected_delta"], rtol=0, atol=case["tol_abs"])
    # qp_test.cc:141-166: inequalities ignored
    delta, _, status = s.NewtonStep(0.0, 0.995, include_inequalities=False)
    assert int(status[0]) == 0
    d = delta.cpu().numpy()[0]
    n, k, m = case["n"], case["k"], case["m"]
    exp = np.array(case["expected_delta_no_ineq_xy"])
    np.testing.assert_allclose(d[:n], exp[:n], rtol=0, atol=1e-12)
    np.testing.assert_allclose(d[n + m:n + m + k], exp[n:n + k], rtol=0, atol=1e-12)
    assert np.all(d[n:n + m] == 0) and np.all(d[n + m + k:] == 0)


# ------------------------------------------------------------------ the reference's KATs that so far only pinned the oracle
@pytest.mark.parametrize("force_generic", [True, False])
def test_compute_alpha_kat_on_device(force_generic):
    """TestComputeAlpha (qp_test.cc:244-249: x = (1, .8, 1.2), dx = (-2, .6, -1.3): alpha = 0.5 at tau = 1, 0.45 at tau = 0.9) through
    mo_newton_step: a batch-1 QP built so that the step's (s, ds) ARE the KAT's (x, dx) -- G = I, c = -2 dx, constraints x_i + s_i >= 0
    at x = 0 with z = s and mu = 0 give (1 + z/s) dx_i = -c_i, ds = dx (qp.cc:337-342, 361) -- so the primal step length the kernel
    returns is ComputeAlpha(s, ds, tau) (qp.cc:485-507)."""
    g = load("alpha.json")
    h = g["head"]
    xk, dxk = np.array(g["x"][:h]), np.array(g["dx"][:h])
    prob = Q.BatchedQP(n=h, k=0, m=h, G=T(np.eye(h)[None]), c=T((-2.0 * dxk)[None]),
                       cons_var=T(np.arange(h, dtype=np.int32)[None], torch.int32), cons_a=T(np.ones((1, h))), cons_b=T(xk[None]))
    s = Q.QPInteriorPointSolver(prob, force_generic=force_generic)
    s.SetVariables(T(np.concatenate([np.zeros(h), xk, xk])[None]))   # [x | s | z]
    for c in g["cases"]:
        delta, alpha, status = s.NewtonStep(0.0, c["tau"])
        assert int(status[0]) == 0
        d = delta.cpu().numpy()[0]
        np.testing.assert_allclose(d[h:2 * h], dxk, rtol=0, atol=1e-15)    # ds is the KAT's dx
        assert abs(float(alpha[0, 0]) - c["alpha"]) < g["tol_abs"], (c, float(alpha[0, 0]))


@pytest.mark.parametrize("case", load("residual.json"), ids=lambda c: c["name"])
@pytest.mark.parametrize("force_generic", [True, False])
def test_update_hessian_kat_on_device(case, force_generic):
    """residual_test.cc:51-182 through mo_linearize: the residual's local Jacobian scattered into the dense stack by its index list
    (what UpdateHessian's gr / gc lookup does, residual.hpp:206-222) gives the reference's J^T J (lower triangle; strict upper exactly 0,
    residual_test.cc:130-134; cells of untouched variables exactly 0, :143-147), J^T r and 0.5 |r|^2."""
    n = case["full_size"]
    Jl = np.array(case["J"])
    Jd = np.zeros((Jl.shape[0], n))
    for l, gidx in enumerate(case["index"]):
        Jd[:, gidx] += Jl[:, l]
    prob = Q.BatchedQP(n=n, k=0, m=0, J=T(Jd[None]), r=T(np.array(case["r"])[None]), lam=0.0)
    G, c, half = Q.linearize(prob, force_generic=force_generic)
    H = G.cpu().numpy()[0].T            # column-major G -> H[row, col]
    np.testing.assert_allclose(H, np.array(case["expected_H_lower"]), rtol=0, atol=case["tol_abs"])
    assert np.all(np.triu(H, 1) == 0)
    np.testing.assert_allclose(c.cpu().numpy()[0], case["expected_b"], rtol=0, atol=case["tol_abs"])
    assert abs(float(half[0]) - case["expected_half_sq"]) < 1e-14
    mask = np.zeros((n, n), bool)
    for i in case["index"]:
        for j in case["index"]:
            mask[i, j] = True
    assert np.all(H[~mask] == 0)


# ------------------------------------------------------------------ synthetic fixtures (numpy full-system LU)
@pytest.mark.parametrize("cfg", ["cfg1", "cfg2", "cfg3", "cfg4"])
@pytest.mark.parametrize("force_generic", [True, False])
def test_synthetic_fixture(cfg, force_generic):
    z = np.load(os.path.join(GOLDEN, "synthetic.npz"))
    g = lambda key: z[f"{cfg}_{key}"]
    f32 = synth.CONFIGS[cfg]["dtype"] == "f32"
    dt = torch.float32 if f32 else torch.float64
    J = g("J")
    B, m_r, n = J.shape
    k, m = g("b_eq").shape[1], g("cons_var").shape[1]
    prob = Q.BatchedQP(n=n, k=k, m=m, J=T(J, dt), r=T(g("r"), dt), lam=float(g("lam")), A_eq=T(g("A_eq"), dt),
                       b_eq=T(g("b_eq"), dt), cons_var=T(g("cons_var"), torch.int32), cons_a=T(g("cons_a"), dt),
                       cons_b=T(g("cons_b"), dt))
    s = Q.QPInteriorPointSolver(prob, force_generic=force_generic)
    s.SetVariables(T(g("vars"), dt))
    delta, alpha, status = s.NewtonStep(T(g("mu"), dt), 0.995)
    assert torch.all(status == 0)
    err = rel_inf_rows(delta.double().cpu().numpy(), g("delta"))
    assert err.max() < (TOL32 if f32 else TOL64), err
    # linearisation (a1/a2)
    G, c, half = Q.linearize(prob)
    Gl = np.tril(G.double().cpu().numpy()[0].T)
    np.testing.assert_allclose(Gl, g("G_lower")[0], rtol=1e-5 if f32 else 1e-12, atol=1e-5 if f32 else 1e-12)
    np.testing.assert_allclose(c.double().cpu().numpy()[0], g("c")[0], rtol=1e-4 if f32 else 1e-12,
                               atol=1e-4 if f32 else 1e-12)
    assert np.all(np.triu(G.cpu().numpy()[0].T, 1) == 0)  # strict upper stays exactly 0 (residual_test.cc:130-134)
    np.testing.assert_allclose(half.double().cpu().numpy(), 0.5 * np.sum(g("r") ** 2, axis=1), rtol=1e-5 if f32 else 1e-13)


# ------------------------------------------------------------------ batches vs the oracle
@pytest.mark.parametrize("cfg,batch", [("cfg1", 257), ("cfg2", 512), ("cfg3", 300), ("cfg4", 64)])
@pytest.mark.parametrize("force_generic", [True, False])
def test_batch_vs_oracle(cfg, batch, force_generic):
    d = synth.CONFIGS[cfg]
    f32 = d["dtype"] == "f32"
    dt = torch.float32 if f32 else torch.float64
    hb = synth.make_batch(d["n"], d["k"], d["m"], d["m_r"], batch, stream=11)
    if f32:
        rd = lambda a: a.astype(np.float32).astype(np.float64)
        for key in ("J", "r", "A_eq", "b_eq", "cons_a", "cons_b", "vars", "mu"):
            setattr(hb, key, rd(getattr(hb, key)))
        hb.lam = float(np.float32(hb.lam))
    s = Q.QPInteriorPointSolver(batch_to_device(hb, dt), force_generic=force_generic)
    s.SetVariables(T(hb.vars, dt))
    delta, alpha, status = s.NewtonStep(T(hb.mu, dt), 0.995)
    ref, ref_alpha, ref_status, _ = orc.batched_newton_step(
        hb.n, hb.k, hb.m, J=hb.J, r=hb.r, lam=hb.lam, A_eq=hb.A_eq, b_eq=hb.b_eq, cons_var=hb.cons_var, cons_a=hb.cons_a,
        cons_b=hb.cons_b, vars_=hb.vars, mu=hb.mu)
    assert np.all(ref_status == 0) and torch.all(status == 0)
    err = rel_inf_rows(delta.double().cpu().numpy(), ref)
    assert err.max() < (TOL32 if f32 else TOL64), (err.max(), s.step_kernel())
    np.testing.assert_allclose(alpha.double().cpu().numpy(), ref_alpha, rtol=0, atol=5e-3 if f32 else 1e-9)


def test_qp_level_input_and_shared_constraints():
    """QP-level (G, c) input with one constraint set shared by the whole batch (stride 0)."""
    d = synth.CONFIGS["cfg2"]
    hb = synth.make_batch(d["n"], d["k"], d["m"], d["m_r"], 64, stream=3)
    G = np.einsum("bqi,bqj->bij", hb.J, hb.J) + hb.lam * np.eye(hb.n)
    c = np.einsum("bqi,bq->bi", hb.J, hb.r)
    cv, ca, cb = hb.cons_var[:1], hb.cons_a[:1], hb.cons_b[:1]
    x = hb.vars[:, :hb.n]
    sl = (ca * x[:, cv[0]] + cb) * 0.9
    vars_ = hb.vars.copy()
    vars_[:, hb.n:hb.n + hb.m] = sl
    prob = Q.BatchedQP(n=hb.n, k=hb.k, m=hb.m, G=T(np.tril(G).transpose(0, 2, 1)), c=T(c), A_eq=T(hb.A_eq), b_eq=T(hb.b_eq),
                       cons_var=T(cv, torch.int32), cons_a=T(ca), cons_b=T(cb))
    s = Q.QPInteriorPointSolver(prob, batch=64)
    s.SetVariables(T(vars_))
    delta, alpha, status = s.NewtonStep(T(hb.mu), 0.995)
    B = 64
    rep = lambda a: np.repeat(a, B, axis=0)
    ref, ref_alpha, ref_status, _ = orc.batched_newton_step(
        hb.n, hb.k, hb.m, G=np.tril(G).transpose(0, 2, 1), c=c, A_eq=hb.A_eq, b_eq=hb.b_eq, cons_var=rep(cv), cons_a=rep(ca),
        cons_b=rep(cb), vars_=vars_, mu=hb.mu)
    assert torch.all(status == 0) and np.all(ref_status == 0)
    assert rel_inf_rows(delta.cpu().numpy(), ref).max() < TOL64


def test_edge_shapes():
    """No constraints at all (m = k = 0), equality only, inequality only, tiny n."""
    rng = np.random.default_rng(5)
    for (n, k, m, m_r) in [(1, 0, 0, 3), (5, 0, 0, 9), (6, 3, 0, 8), (7, 0, 6, 10), (3, 3, 2, 4), (20, 1, 40, 33)]:
        B = 9
        J = rng.uniform(-1, 1, (B, m_r, n)); r = rng.uniform(-1, 1, (B, m_r))
        A = rng.uniform(-1, 1, (B, n, k)); b = rng.uniform(-1, 1, (B, k))
        cv = rng.integers(0, n, (B, m)).astype(np.int32)
        ca = rng.choice([-1.0, 1.0, 2.5], (B, m)); cb = rng.uniform(0.5, 2.0, (B, m))
        x = rng.uniform(-0.1, 0.1, (B, n))
        sl = rng.uniform(0.2, 1.5, (B, m)); z = rng.uniform(0.1, 2, (B, m)); y = rng.uniform(-1, 1, (B, k))
        vars_ = np.concatenate([x, sl, y, z], axis=1)
        mu = np.full(B, 0.05)
        prob = Q.BatchedQP(n=n, k=k, m=m, J=T(J), r=T(r), lam=1e-3, A_eq=T(A) if k else None, b_eq=T(b) if k else None,
                           cons_var=T(cv, torch.int32) if m else None, cons_a=T(ca) if m else None,
                           cons_b=T(cb) if m else None)
        s = Q.QPInteriorPointSolver(prob)
        s.SetVariables(T(vars_))
        delta, alpha, status = s.NewtonStep(T(mu), 0.995)
        ref, ref_alpha, ref_status, _ = orc.batched_newton_step(
            n, k, m, J=J, r=r, lam=1e-3, A_eq=A if k else None, b_eq=b if k else None, cons_var=cv if m else None,
            cons_a=ca if m else None, cons_b=cb if m else None, vars_=vars_, mu=mu)
        assert torch.all(status == 0) and np.all(ref_status == 0), (n, k, m)
        assert rel_inf_rows(delta.cpu().numpy(), ref).max() < 1e-9, (n, k, m)
        np.testing.assert_allclose(alpha.cpu().numpy(), ref_alpha, atol=1e-9)


def test_status_words():
    """Per-problem failures never abort the batch: s <= 0 (qp.cc:285), singular KKT (qp.cc:303-307), bad index (qp.cc:70-72)."""
    d = synth.CONFIGS["cfg1"]
    hb = synth.make_batch(d["n"], d["k"], d["m"], d["m_r"], 8, stream=2)
    hb.vars[1, hb.n] = 0.0            # s = 0
    hb.vars[2, hb.n + 1] = -1.0       # s < 0
    hb.A_eq[3, :, 1] = hb.A_eq[3, :, 0]  # duplicate equality row -> singular Schur complement (zero pivot, non-zero? no: exact dup)
    hb.cons_var[4, 0] = hb.n + 3      # out-of-range index
    hb.J[5, 0, 0] = np.nan
    s = Q.QPInteriorPointSolver(batch_to_device(hb))
    s.SetVariables(T(hb.vars))
    delta, alpha, status = s.NewtonStep(T(hb.mu), 0.995)
    st = status.cpu().numpy()
    assert st[0] == 0 and st[6] == 0 and st[7] == 0
    assert st[1] == L.MO_STATUS_NONPOSITIVE_SLACK and st[2] == L.MO_STATUS_NONPOSITIVE_SLACK
    assert st[4] == L.MO_STATUS_BAD_INDEX
    assert st[5] in (L.MO_STATUS_NONFINITE, L.MO_STATUS_FACTORIZATION_FAILED)
    dn = delta.cpu().numpy()
    assert np.all(np.isnan(dn[1])) and np.all(np.isnan(dn[4])) and np.all(np.isfinite(dn[0]))
    # the oracle agrees on the two failure classes the reference defines
    _, _, ref_status, _ = orc.batched_newton_step(hb.n, hb.k, hb.m, J=hb.J, r=hb.r, lam=hb.lam, A_eq=hb.A_eq, b_eq=hb.b_eq,
                                                  cons_var=np.clip(hb.cons_var, 0, hb.n - 1), cons_a=hb.cons_a,
                                                  cons_b=hb.cons_b, vars_=hb.vars, mu=hb.mu)
    assert ref_status[1] == orc.ORC_NONPOSITIVE_SLACK and ref_status[2] == orc.ORC_NONPOSITIVE_SLACK


def test_argument_errors():
    d = synth.CONFIGS["cfg1"]
    hb = synth.make_batch(d["n"], d["k"], d["m"], d["m_r"], 4)
    prob = batch_to_device(hb)
    s = Q.QPInteriorPointSolver(prob)
    with pytest.raises(L.MiniOptError):
        s.NewtonStep(0.1, tau=1.5)
    with pytest.raises(L.MiniOptError):
        s.Solve(Q.Params(sigma=0.0))
    with pytest.raises(L.MiniOptError):
        s.Solve(Q.Params(max_iterations=0))


# ------------------------------------------------------------------ Iterate (qp.cc:153-201) incl. predictor-corrector
@pytest.mark.parametrize("strategy", [Q.COMPLEMENTARITY, Q.PREDICTOR_CORRECTOR])
def test_iterate_vs_oracle(strategy):
    d = synth.CONFIGS["cfg2"]
    hb = synth.make_batch(d["n"], d["k"], d["m"], d["m_r"], 16, stream=5)
    s = Q.QPInteriorPointSolver(batch_to_device(hb))
    s.SetVariables(T(hb.vars))
    ip, status = s.Iterate(T(hb.mu), strategy)
    assert torch.all(status == 0)
    got_vars = s.variables().cpu().numpy()
    ip = ip.cpu().numpy()
    for p in range(16):
        G, c, _ = orc.linearize_dense(hb.J[p], hb.r[p], hb.lam)
        o = orc.Solver(orc.QP(G=G, c=c, A_eq=hb.A_eq[p].T, b_eq=hb.b_eq[p], cons_var=hb.cons_var[p], cons_a=hb.cons_a[p],
                              cons_b=hb.cons_b[p]))
        o.variables[:] = hb.vars[p]
        st, out = o.iterate(float(hb.mu[p]), strategy)
        assert st == 0
        assert np.max(np.abs(got_vars[p] - o.variables)) / np.max(np.abs(o.variables)) < 1e-10
        exp = [out.mu, out.alpha_primal, out.alpha_dual, out.alpha_probe_primal, out.alpha_probe_dual, out.mu_affine]
        np.testing.assert_allclose(ip[p], exp, rtol=1e-8, atol=1e-12, equal_nan=True)


# ------------------------------------------------------------------ full Solve KATs (qp_test.cc:252-471)
GUESS = {"NAIVE": Q.NAIVE, "SOLVE_EQUALITY_CONSTRAINED": Q.SOLVE_EQUALITY_CONSTRAINED}


@pytest.mark.parametrize("case", load("solve_kats.json"), ids=lambda c: c["name"])
def test_full_solve_kats(case):
    from tests.test_oracle_golden import check_kat_expectations, qp_from, GUESS as OG, STRAT
    prob = qp_from_case(case)
    for guess in case["guesses"]:
        s = Q.QPInteriorPointSolver(prob)
        kw = dict(case["params"])
        kw.pop("barrier_strategy", None)
        out = s.Solve(Q.Params(initial_guess_method=GUESS[guess], **kw))
        assert int(out.status[0]) == 0
        assert int(out.termination_state[0]) == Q.SATISFIED_KKT_TOL, (case["name"], guess)
        n, k, m = case["n"], case["k"], case["m"]
        v = s.variables().cpu().numpy()[0]
        oqp = qp_from(case)
        check_kat_expectations(case, v[:n], v[n:n + m], v[n + m:n + m + k], v[n + m + k:], oqp)
        # iteration-by-iteration agreement with the oracle's Solve
        o = orc.Solver(oqp)
        term, its = o.solve(initial_guess_method=OG[guess], **kw)
        assert term == int(out.termination_state[0]) and len(its) == int(out.num_iterations[0])
        np.testing.assert_allclose(v, o.variables, rtol=1e-7, atol=1e-9)
        rec = out.iterations.cpu().numpy()[0]
        # absolute rounding noise scales with the largest quantity of the whole solve (z starts at 1/s = 1e9 when an
        # initial slack is clamped to 1e-9, qp.cc:475-480)
        scale = max(max(it.kkt_initial.r_dual, it.kkt_initial.r_comp, it.kkt_initial.r_primal_ineq) for it in its)
        for i, it in enumerate(its):
            exp = [it.kkt_initial.r_dual, it.kkt_initial.r_comp, it.kkt_initial.r_primal_eq, it.kkt_initial.r_primal_ineq,
                   it.kkt_final.r_dual, it.kkt_final.r_comp, it.kkt_final.r_primal_eq, it.kkt_final.r_primal_ineq,
                   it.ip.mu, it.ip.alpha_primal, it.ip.alpha_dual]
            # absolute noise scales with the largest residual of the iteration (e.g. |r_dual| ~ 1e9 on the first step)
            np.testing.assert_allclose(rec[i][:11], exp, rtol=1e-6, atol=1e-9 + 1e-14 * scale)
        if k:
            lag = out.lagrange_multipliers.cpu().numpy()[0]
            y = o.blocks(o.variables)[2]
            np.testing.assert_allclose(lag, [y.min(), np.abs(y).max()], rtol=1e-7, atol=1e-9)


def _solve_generated_batch(problems, n, force_generic=False):
    """Group generated problems by constraint count (one plan per m) and run the device Solve with the parameters of
    TestGeneratedProblems (qp_test.cc:541-549); yields (method, group, solver, outputs)."""
    by_m = {}
    for pr in problems:
        by_m.setdefault(len(pr[2]), []).append(pr)
    for m, group in sorted(by_m.items()):
        G = np.stack([np.tril(g[0]).T for g in group])
        c = np.stack([g[1] for g in group])
        kw = {}
        if m:
            kw = dict(cons_var=T(np.array([[q[0] for q in g[2]] for g in group], dtype=np.int32), torch.int32),
                      cons_a=T(np.array([[q[1] for q in g[2]] for g in group])),
                      cons_b=T(np.array([[q[2] for q in g[2]] for g in group])))
        s = Q.QPInteriorPointSolver(Q.BatchedQP(n=n, k=0, m=m, G=T(G), c=T(c), **kw), force_generic=force_generic)
        for method in (Q.NAIVE, Q.SOLVE_EQUALITY_CONSTRAINED):
            out = s.Solve(Q.Params(termination_kkt_tol=1e-12, max_iterations=30, initial_guess_method=method))
            yield method, group, s, out


@pytest.mark.parametrize("force_generic", [False, True])
def test_batched_solve_generated_problems(force_generic):
    """TestGeneratedProblems (qp_test.cc:527-574) on the device Solve with the reference's own assertions -- 1000 random
    N=8 QPs, both initial-guess methods, <= 30 iterations, |x - x*|inf <= 5e-5, |s|inf <= 5e-5 (qp_test.cc:555-562),
    4 * iterations(SOLVE_EQUALITY_CONSTRAINED) < iterations(NAIVE) (:572-573) -- and EVERY problem must end in the
    oracle's termination state after the oracle's number of iterations, at the oracle's optimum."""
    from tests.helpers import generated_qps
    n = 8
    total = {Q.NAIVE: 0, Q.SOLVE_EQUALITY_CONSTRAINED: 0}
    for method, group, s, out in _solve_generated_batch(generated_qps(1000, n), n, force_generic):
        assert torch.all(out.status == 0)
        m = len(group[0][2])
        x = s.x_block().cpu().numpy()
        sl = s.s_block().cpu().numpy() if m else None
        nit = out.num_iterations.cpu().numpy()
        term = out.termination_state.cpu().numpy()
        total[method] += int(nit.sum())
        assert nit.max() <= 30
        for i, (Gi, ci, cons, x_solution) in enumerate(group):
            assert np.abs(x[i] - x_solution).max() <= 5e-5, (m, i, method)
            if m:
                assert np.abs(sl[i]).max() <= 5e-5, (m, i, method)
            o = orc.Solver(orc.QP(G=Gi, c=ci, cons_var=[q[0] for q in cons], cons_a=[q[1] for q in cons],
                                  cons_b=[q[2] for q in cons]))
            t, its = o.solve(termination_kkt_tol=1e-12, max_iterations=30, initial_guess_method=method)
            assert t == term[i] and len(its) == nit[i], (m, i, method, t, term[i], len(its), nit[i])
            np.testing.assert_allclose(x[i], o.variables[:n], rtol=1e-7, atol=1e-8)
    assert total[Q.SOLVE_EQUALITY_CONSTRAINED] * 4 < total[Q.NAIVE], total


def test_batched_solve_dense_generated_problems():
    """OUR stress test, not a reference test (tests/helpers.py::dense_generated_qps: cond(G) up to 1e11, coupled
    bounds): device and oracle need not take identical branches on such problems, but wherever the device Solve reports
    SATISFIED_KKT_TOL its point is certified by an independent KKT check, and wherever both agree on the iteration count
    they agree on the optimum."""
    from tests.helpers import dense_generated_qps
    n = 8
    certified = 0
    for method, group, s, out in _solve_generated_batch(dense_generated_qps(200, n), n):
        assert torch.all(out.status == 0)
        m = len(group[0][2])
        v = s.variables().cpu().numpy()
        nit = out.num_iterations.cpu().numpy()
        term = out.termination_state.cpu().numpy()
        for i, (Gi, ci, cons, _) in enumerate(group):
            o = orc.Solver(orc.QP(G=Gi, c=ci, cons_var=[q[0] for q in cons], cons_a=[q[1] for q in cons],
                                  cons_b=[q[2] for q in cons]))
            t, its = o.solve(termination_kkt_tol=1e-12, max_iterations=30, initial_guess_method=method)
            if t == term[i] and len(its) == nit[i]:  # cond(G) up to 1e11: agreement to cond * eps of the largest component
                assert np.abs(v[i, :n] - o.variables[:n]).max() <= 1e-5 * max(1.0, np.abs(o.variables[:n]).max())
            if term[i] != Q.SATISFIED_KKT_TOL:
                continue
            certified += 1
            x, z = v[i, :n], v[i, n + m:]
            grad = Gi @ x + ci
            scale = max(1.0, np.abs(Gi @ x).max(), np.abs(ci).max())
            for q, (var, a, b) in enumerate(cons):
                grad[var] -= a * z[q]
                assert a * x[var] + b >= -1e-9 * max(1.0, abs(b)) and z[q] >= 0
                assert abs((a * x[var] + b) * z[q]) <= 1e-5 * scale
            assert np.abs(grad).max() <= 1e-9 * scale
    assert certified > 0


def test_fused_kernel_is_selected_for_headline_configs():
    """BASELINE configs[1] and [2] must run on the fused MFMA kernel (not silently on the generic one)."""
    for cfg, name in (("cfg2", "fused_mfma_f64_n32"), ("cfg3", "fused_mfma_f64_n64")):
        d = synth.CONFIGS[cfg]
        hb = synth.make_batch(d["n"], d["k"], d["m"], d["m_r"], 4)
        s = Q.QPInteriorPointSolver(batch_to_device(hb))
        assert s.step_kernel() == name
        assert Q.QPInteriorPointSolver(batch_to_device(hb), force_generic=True).step_kernel() == "generic"


# ------------------------------------------------------------------ fused on-device Solve (row f1) vs the oracle's Solve
@pytest.mark.parametrize("cfg", ["cfg2", "cfg3"])
@pytest.mark.parametrize("guess", [Q.NAIVE, Q.SOLVE_EQUALITY_CONSTRAINED, Q.USER_PROVIDED])
@pytest.mark.parametrize("strategy", [Q.COMPLEMENTARITY, Q.FIXED_DECREASE, Q.PREDICTOR_CORRECTOR])
@pytest.mark.parametrize("mu_from_state", [False, True], ids=["initial_mu", "mu_from_complementarity"])
def test_fused_solve_vs_oracle(cfg, guess, strategy, mu_from_state):
    """mo_qp_solve on J-level input runs the fused Solve kernel: same termination, iteration count, optimum and per-iteration
    KKT records as the oracle's restatement of QPInteriorPointSolver::Solve (qp.cc:100-151), problem by problem; the generic
    kernel must agree as well.  mu_from_state = Params::initialize_mu_with_complementarity (qp.cc:115: mu = s.z / M of the
    initial guess, whichever method produced it -- the caller's own state with USER_PROVIDED)."""
    d = synth.CONFIGS[cfg]
    B = 24
    hb = synth.make_batch(d["n"], d["k"], d["m"], d["m_r"], B, stream=21)
    kw = dict(initial_mu=1.0, sigma=0.1, termination_kkt_tol=1e-9, max_iterations=12, barrier_strategy=strategy,
              initial_guess_method=guess, initialize_mu_with_complementarity=int(mu_from_state))
    results = {}
    for force in (False, True):
        s = Q.QPInteriorPointSolver(batch_to_device(hb), force_generic=force)
        s.SetVariables(T(hb.vars))
        out = s.Solve(Q.Params(**kw))
        assert torch.all(out.status == 0)
        results[force] = (s.variables().cpu().numpy().copy(), out.num_iterations.cpu().numpy(), out.termination_state.cpu().numpy(),
                          out.iterations.cpu().numpy(), out.lagrange_multipliers.cpu().numpy())
    n, k, m = hb.n, hb.k, hb.m
    for p in range(B):
        G, c, _ = orc.linearize_dense(hb.J[p], hb.r[p], hb.lam)
        o = orc.Solver(orc.QP(G=G, c=c, A_eq=hb.A_eq[p].T, b_eq=hb.b_eq[p], cons_var=hb.cons_var[p], cons_a=hb.cons_a[p],
                              cons_b=hb.cons_b[p]))
        o.variables[:] = hb.vars[p]
        term, its = o.solve(**kw)
        for force in (False, True):
            v, nit, tm, rec, lag = results[force]
            assert tm[p] == term and nit[p] == len(its), (p, force, tm[p], term, nit[p], len(its))
            np.testing.assert_allclose(v[p], o.variables, rtol=1e-7, atol=1e-9)
            scale = max(max(i.kkt_initial.r_dual, i.kkt_initial.r_comp, i.kkt_initial.r_primal_ineq, i.kkt_initial.r_primal_eq) for i in its)
            for i, itr in enumerate(its):
                exp = [itr.kkt_initial.r_dual, itr.kkt_initial.r_comp, itr.kkt_initial.r_primal_eq, itr.kkt_initial.r_primal_ineq,
                       itr.kkt_final.r_dual, itr.kkt_final.r_comp, itr.kkt_final.r_primal_eq, itr.kkt_final.r_primal_ineq,
                       itr.ip.mu, itr.ip.alpha_primal, itr.ip.alpha_dual,
                       itr.ip.alpha_probe_primal, itr.ip.alpha_probe_dual, itr.ip.mu_affine]
                np.testing.assert_allclose(rec[p][i], exp, rtol=1e-6, atol=1e-9 + 1e-12 * scale, equal_nan=True)
            y = o.blocks(o.variables)[2]
            np.testing.assert_allclose(lag[p], [y.min(), np.abs(y).max()], rtol=1e-7, atol=1e-9)


# ------------------------------------------------------------------ fused-kernel edge cases (shapes it accepts beyond the BASELINE configs)
@pytest.mark.parametrize("n,k,m,m_r", [(32, 0, 0, 8), (32, 0, 10, 36), (32, 15, 2, 64), (64, 0, 64, 4), (64, 15, 0, 132), (64, 1, 1, 128),
                                       (64, 3, 9, 1), (64, 8, 32, 131), (32, 4, 16, 66), (32, 0, 5, 3)])
def test_fused_edge_shapes(n, k, m, m_r):
    """k = 0, m = 0, k = 15 (the largest the right-hand-side column leaves room for), m = 64, tiny and ragged m_r, duplicated
    constraint variables -- all through the fused kernel, against the oracle."""
    rng = np.random.default_rng(n * 1000 + k * 100 + m + m_r)
    B = 33
    J = rng.uniform(-1, 1, (B, m_r, n)); r = rng.uniform(-1, 1, (B, m_r))
    A = rng.uniform(-1, 1, (B, n, k)); b = rng.uniform(-1, 1, (B, k))
    cv = rng.integers(0, min(n, 5), (B, m)).astype(np.int32)      # few distinct variables -> many duplicates
    ca = rng.choice([-1.0, 1.0, 2.5], (B, m)); cb = rng.uniform(0.5, 2.0, (B, m))
    x = rng.uniform(-0.1, 0.1, (B, n))
    sl = rng.uniform(0.2, 1.5, (B, m)); z = rng.uniform(0.1, 2, (B, m)); y = rng.uniform(-1, 1, (B, k))
    vars_ = np.concatenate([x, sl, y, z], axis=1)
    V = vars_.shape[1]
    mu = np.full(B, 0.05)
    lam = 0.5 if m_r < n else 1e-3                                  # rank-deficient J^T J needs the LM damping
    prob = Q.BatchedQP(n=n, k=k, m=m, J=T(J), r=T(r), lam=lam, A_eq=T(A) if k else None, b_eq=T(b) if k else None,
                       cons_var=T(cv, torch.int32) if m else None, cons_a=T(ca) if m else None, cons_b=T(cb) if m else None)
    s = Q.QPInteriorPointSolver(prob)
    assert s.step_kernel().startswith("fused"), s.step_kernel()
    s.SetVariables(T(vars_))
    delta, alpha, status = s.NewtonStep(T(mu), 0.995)
    ref, ref_alpha, ref_status, _ = orc.batched_newton_step(
        n, k, m, J=J, r=r, lam=lam, A_eq=A if k else None, b_eq=b if k else None, cons_var=cv if m else None,
        cons_a=ca if m else None, cons_b=cb if m else None, vars_=vars_, mu=mu)
    assert torch.all(status == 0) and np.all(ref_status == 0)
    assert rel_inf_rows(delta.cpu().numpy(), ref).max() < 1e-9
    np.testing.assert_allclose(alpha.cpu().numpy(), ref_alpha, atol=1e-9)


def test_fused_status_words_and_shared_constraints():
    """Per-problem failures on the fused path (s <= 0, bad index, NaN input) never disturb the neighbours; one constraint set
    shared by the whole batch (stride 0)."""
    d = synth.CONFIGS["cfg2"]
    hb = synth.make_batch(d["n"], d["k"], d["m"], d["m_r"], 16, stream=8)
    n, m = hb.n, hb.m
    hb.vars[1, n] = 0.0
    hb.vars[2, n + 3] = -0.5
    hb.cons_var[4, 0] = n + 7
    hb.J[5, 3, 2] = np.nan
    hb.cons_var[6, 1] = -1
    s = Q.QPInteriorPointSolver(batch_to_device(hb))
    assert s.step_kernel().startswith("fused")
    s.SetVariables(T(hb.vars))
    delta, alpha, status = s.NewtonStep(T(hb.mu), 0.995)
    st = status.cpu().numpy()
    good = [0, 3, 7, 8, 9, 10, 11, 12, 13, 14, 15]
    assert np.all(st[good] == 0)
    assert st[1] == L.MO_STATUS_NONPOSITIVE_SLACK and st[2] == L.MO_STATUS_NONPOSITIVE_SLACK
    assert st[4] == L.MO_STATUS_BAD_INDEX and st[6] == L.MO_STATUS_BAD_INDEX
    assert st[5] in (L.MO_STATUS_NONFINITE, L.MO_STATUS_FACTORIZATION_FAILED)
    dn = delta.cpu().numpy()
    assert np.all(np.isnan(dn[[1, 2, 4, 5, 6]])) and np.all(np.isfinite(dn[good]))
    ref, _, ref_status, _ = orc.batched_newton_step(hb.n, hb.k, hb.m, J=hb.J[good], r=hb.r[good], lam=hb.lam, A_eq=hb.A_eq[good],
                                                    b_eq=hb.b_eq[good], cons_var=hb.cons_var[good], cons_a=hb.cons_a[good],
                                                    cons_b=hb.cons_b[good], vars_=hb.vars[good], mu=hb.mu[good])
    assert rel_inf_rows(dn[good], ref).max() < 1e-10
    # the same failures inside the fused Solve kernel
    out = s.Solve(Q.Params(initial_guess_method=Q.USER_PROVIDED, max_iterations=5))
    so = out.status.cpu().numpy()
    assert np.all(so[good] == 0) and so[1] == L.MO_STATUS_NONPOSITIVE_SLACK and so[4] == L.MO_STATUS_BAD_INDEX
    # shared constraints (stride 0)
    hb2 = synth.make_batch(d["n"], d["k"], d["m"], d["m_r"], 16, stream=9)
    cv, ca, cb = hb2.cons_var[:1], hb2.cons_a[:1], hb2.cons_b[:1]
    vars2 = hb2.vars.copy()
    vars2[:, n:n + m] = (ca * hb2.vars[:, :n][:, cv[0]] + cb) * 0.8
    prob = Q.BatchedQP(n=hb2.n, k=hb2.k, m=hb2.m, J=T(hb2.J), r=T(hb2.r), lam=hb2.lam, A_eq=T(hb2.A_eq), b_eq=T(hb2.b_eq),
                       cons_var=T(cv, torch.int32), cons_a=T(ca), cons_b=T(cb))
    s2 = Q.QPInteriorPointSolver(prob, batch=16)
    assert s2.step_kernel().startswith("fused")
    s2.SetVariables(T(vars2))
    delta2, _, status2 = s2.NewtonStep(T(hb2.mu), 0.995)
    rep = lambda a_: np.repeat(a_, 16, axis=0)
    ref2, _, rs2, _ = orc.batched_newton_step(hb2.n, hb2.k, hb2.m, J=hb2.J, r=hb2.r, lam=hb2.lam, A_eq=hb2.A_eq, b_eq=hb2.b_eq,
                                              cons_var=rep(cv), cons_a=rep(ca), cons_b=rep(cb), vars_=vars2, mu=hb2.mu)
    assert torch.all(status2 == 0) and np.all(rs2 == 0)
    assert rel_inf_rows(delta2.cpu().numpy(), ref2).max() < 1e-10


@pytest.mark.parametrize("cfg", ["cfg2", "cfg3"])
def test_fused_qp_level_step_and_solve(cfg):
    """QP-level input (G lower-triangular column-major, c: the reference's own mini_opt::QP) on the fused kernels: Newton step,
    Iterate and the full Solve against the oracle, problem by problem.  Uses an odd state stride on purpose (no alignment needs)."""
    d = synth.CONFIGS[cfg]
    B = 17
    hb = synth.make_batch(d["n"], d["k"], d["m"] - 1, d["m_r"], B, stream=31) if False else synth.make_batch(d["n"], d["k"], d["m"], d["m_r"], B, stream=31)
    n, k, m = hb.n, hb.k, hb.m
    G = np.einsum("bqi,bqj->bij", hb.J, hb.J) + hb.lam * np.eye(n)
    c = np.einsum("bqi,bq->bi", hb.J, hb.r)
    Gl = np.tril(G)                                   # only the lower triangle is valid, like Eigen's triangularView<Lower>
    Gl_bad_upper = Gl + np.triu(np.full((n, n), 123.0), 1)   # garbage above the diagonal must be ignored (qp.cc:289, :404)
    prob = Q.BatchedQP(n=n, k=k, m=m, G=T(Gl_bad_upper.transpose(0, 2, 1)), c=T(c), A_eq=T(hb.A_eq), b_eq=T(hb.b_eq),
                       cons_var=T(hb.cons_var, torch.int32), cons_a=T(hb.cons_a), cons_b=T(hb.cons_b))
    s = Q.QPInteriorPointSolver(prob)
    assert s.step_kernel().startswith("fused_qp"), s.step_kernel()
    s.SetVariables(T(hb.vars))
    delta, alpha, status = s.NewtonStep(T(hb.mu), 0.995)
    ref, ref_alpha, ref_status, _ = orc.batched_newton_step(n, k, m, G=Gl.transpose(0, 2, 1), c=c, A_eq=hb.A_eq, b_eq=hb.b_eq,
                                                            cons_var=hb.cons_var, cons_a=hb.cons_a, cons_b=hb.cons_b, vars_=hb.vars, mu=hb.mu)
    assert torch.all(status == 0) and np.all(ref_status == 0)
    assert rel_inf_rows(delta.cpu().numpy(), ref).max() < TOL64
    np.testing.assert_allclose(alpha.cpu().numpy(), ref_alpha, atol=1e-9)
    kw = dict(initial_mu=1.0, sigma=0.1, termination_kkt_tol=1e-10, max_iterations=14, initial_guess_method=Q.SOLVE_EQUALITY_CONSTRAINED)
    out = s.Solve(Q.Params(**kw))
    assert torch.all(out.status == 0)
    v = s.variables().cpu().numpy()
    for p in range(B):
        o = orc.Solver(orc.QP(G=Gl[p], c=c[p], A_eq=hb.A_eq[p].T, b_eq=hb.b_eq[p], cons_var=hb.cons_var[p], cons_a=hb.cons_a[p],
                              cons_b=hb.cons_b[p]))
        term, its = o.solve(**kw)
        assert int(out.termination_state[p]) == term and int(out.num_iterations[p]) == len(its)
        np.testing.assert_allclose(v[p], o.variables, rtol=1e-7, atol=1e-9)


# ------------------------------------------------------------------ fused fp32 kernel (BASELINE configs[3] and neighbours)
@pytest.mark.parametrize("n,k,m,m_r", [(128, 16, 64, 256), (128, 0, 0, 8), (128, 16, 3, 132), (128, 5, 64, 4), (64, 8, 32, 128),
                                       (64, 16, 0, 64), (64, 0, 17, 260),
                                       # any multiple of 4 up to 128, padded inside the kernel to the 64 / 128 grid (round 3)
                                       (100, 10, 40, 200), (36, 4, 20, 64), (8, 2, 4, 16), (124, 16, 64, 256), (68, 7, 33, 72), (4, 0, 3, 8),
                                       (60, 16, 64, 60)])
def test_fused_f32_shapes(n, k, m, m_r):
    """The fp32 MFMA kernel (kkt_fused_f32.hip): cfg 4 itself, k = 16 (a full y tile) and k = 0, m = 64 and m = 0, tiny and ragged
    m_r, duplicated constraint variables, the n = 64 instantiation -- against the fp64 oracle on fp32-rounded inputs (no
    reference counterpart exists for fp32; tolerance 2e-3 rel-inf as for the generic kernel, observed ~1e-6)."""
    rng = np.random.default_rng(n * 1000 + k * 100 + m + m_r)
    B = 19
    f = lambda a: a.astype(np.float32).astype(np.float64)
    J = f(rng.uniform(-1, 1, (B, m_r, n))); r = f(rng.uniform(-1, 1, (B, m_r)))
    A = f(rng.uniform(-1, 1, (B, n, k))); b = f(rng.uniform(-1, 1, (B, k)))
    cv = rng.integers(0, min(n, 7), (B, m)).astype(np.int32)
    ca = rng.choice([-1.0, 1.0, 2.5], (B, m)); cb = f(rng.uniform(0.5, 2.0, (B, m)))
    x = f(rng.uniform(-0.1, 0.1, (B, n)))
    sl = f(rng.uniform(0.2, 1.5, (B, m))); z = f(rng.uniform(0.1, 2, (B, m))); y = f(rng.uniform(-1, 1, (B, k)))
    vars_ = np.concatenate([x, sl, y, z], axis=1)
    V = vars_.shape[1]
    pad = (-V) % 4                                                  # the kernel stores dx with 16-byte stores: stride % 4 == 0
    mu = np.full(B, float(np.float32(0.05)))
    lam = float(np.float32(0.5 if m_r < n else 1e-3))
    dt = torch.float32
    prob = Q.BatchedQP(n=n, k=k, m=m, J=T(J, dt), r=T(r, dt), lam=lam, A_eq=T(A, dt) if k else None, b_eq=T(b, dt) if k else None,
                       cons_var=T(cv, torch.int32) if m else None, cons_a=T(ca, dt) if m else None, cons_b=T(cb, dt) if m else None)
    s = Q.QPInteriorPointSolver(prob)
    # (round 2: V % 4 != 0 fell back to the generic kernel -- dx went out in 16-byte stores only; now scalar stores where the stride asks for them)
    assert s.step_kernel() == ("fused_mfma_f32_n128" if n > 64 else "fused_mfma_f32_n64"), s.step_kernel()
    s.SetVariables(T(vars_, dt))
    delta, alpha, status = s.NewtonStep(T(mu, dt), 0.995)
    ref, ref_alpha, ref_status, _ = orc.batched_newton_step(
        n, k, m, J=J, r=r, lam=lam, A_eq=A if k else None, b_eq=b if k else None, cons_var=cv if m else None,
        cons_a=ca if m else None, cons_b=cb if m else None, vars_=vars_, mu=mu)
    assert torch.all(status == 0) and np.all(ref_status == 0)
    err = rel_inf_rows(delta.double().cpu().numpy(), ref)
    assert err.max() < TOL32, err.max()
    np.testing.assert_allclose(alpha.double().cpu().numpy(), ref_alpha, atol=5e-3)
    # MO_STEP_NO_INEQUALITIES (SolveForUpdateNoInequalities, qp.cc:366-386) runs on the fp32 fused kernel as well: dx, dy of the problem without its
    # inequalities, ds = dz = 0, both step lengths 1 -- against the oracle on the same problem with m = 0
    d0, a0, st0 = s.NewtonStep(T(mu, dt), 0.995, include_inequalities=False)
    assert torch.all(st0 == 0) and torch.all(a0 == 1.0)
    d0 = d0.double().cpu().numpy()
    assert np.all(d0[:, n:n + m] == 0) and np.all(d0[:, n + m + k:] == 0)
    vars0 = np.concatenate([x, y], axis=1)
    ref0, _, rs0, _ = orc.batched_newton_step(n, k, 0, J=J, r=r, lam=lam, A_eq=A if k else None, b_eq=b if k else None, vars_=vars0, mu=mu)
    assert np.all(rs0 == 0)
    got0 = np.concatenate([d0[:, :n], d0[:, n + m:n + m + k]], axis=1)
    assert rel_inf_rows(got0, ref0).max() < TOL32


def test_fused_f32_status_words():
    d = synth.CONFIGS["cfg4"]
    hb = synth.make_batch(d["n"], d["k"], d["m"], d["m_r"], 12, stream=4)
    n = hb.n
    hb.vars[1, n] = 0.0
    hb.cons_var[4, 0] = n + 7
    hb.J[5, 3, 2] = np.nan
    hb.cons_var[6, 1] = -1
    s = Q.QPInteriorPointSolver(batch_to_device(hb, torch.float32))
    assert s.step_kernel() == "fused_mfma_f32_n128"
    s.SetVariables(T(hb.vars, torch.float32))
    delta, alpha, status = s.NewtonStep(T(hb.mu, torch.float32), 0.995)
    st = status.cpu().numpy()
    good = [0, 2, 3, 7, 8, 9, 10, 11]
    assert np.all(st[good] == 0)
    assert st[1] == L.MO_STATUS_NONPOSITIVE_SLACK and st[4] == L.MO_STATUS_BAD_INDEX and st[6] == L.MO_STATUS_BAD_INDEX
    assert st[5] in (L.MO_STATUS_NONFINITE, L.MO_STATUS_FACTORIZATION_FAILED)
    dn = delta.cpu().numpy()
    assert np.all(np.isnan(dn[[1, 4, 5, 6]])) and np.all(np.isfinite(dn[good]))


@pytest.mark.parametrize("n,k,m,m_r,level", [(128, 16, 64, 256, "J"), (128, 0, 0, 128, "J"), (128, 16, 3, 132, "J"), (64, 8, 32, 128, "J"),
                                             (64, 16, 0, 64, "J"), (64, 0, 17, 260, "J"), (128, 16, 64, 0, "QP"), (64, 5, 20, 0, "QP"),
                                             (100, 10, 40, 200, "J"), (36, 4, 20, 64, "J"), (8, 2, 4, 16, "J"), (124, 16, 64, 256, "J"),
                                             (68, 7, 33, 72, "J"), (100, 10, 40, 0, "QP"), (12, 3, 9, 0, "QP")])
def test_fused_f32_solve_iterate_residual(n, k, m, m_r, level):
    """kkt_fused_f32_solve_kernel: EvaluateKKTConditions, Iterate (all three barrier strategies) and the whole Solve in fp32 for the
    n = 64 / 128 tile grids -- BASELINE configs[3] in every mode -- against the fp64 fused kernels on the same (fp32-rounded) inputs
    (those are pinned to the oracle above) and against the generic fp32 kernel.  fp32 has no reference counterpart; tolerances are
    fp32 ones (2e-3 rel-inf for directions and states, as for the step kernel)."""
    rng = np.random.default_rng(n * 1000 + k * 100 + m + m_r)
    B = 13
    f = lambda a: a.astype(np.float32).astype(np.float64)
    mr = m_r if m_r else 2 * n
    J = f(rng.uniform(-1, 1, (B, mr, n))); r = f(rng.uniform(-1, 1, (B, mr)))
    A = f(rng.uniform(-1, 1, (B, n, k))); b = f(0.3 * rng.uniform(-1, 1, (B, k)))
    cv = rng.integers(0, n, (B, m)).astype(np.int32)
    ca = rng.choice([-1.0, 1.0, 2.0], (B, m)); cb = f(rng.uniform(0.5, 2.0, (B, m)))
    x = f(rng.uniform(-0.1, 0.1, (B, n)))
    sl = f(rng.uniform(0.2, 1.5, (B, m))); z = f(rng.uniform(0.1, 2, (B, m))); y = f(rng.uniform(-1, 1, (B, k)))
    vars_ = np.concatenate([x, sl, y, z], axis=1)
    mu = np.full(B, float(np.float32(0.05)))
    lam = float(np.float32(1e-2))
    G = f(np.einsum("bqi,bqj->bij", J, J) + lam * np.eye(n)); c = f(np.einsum("bqi,bq->bi", J, r))

    def problem(dt):
        common = dict(A_eq=T(A, dt) if k else None, b_eq=T(b, dt) if k else None, cons_var=T(cv, torch.int32) if m else None,
                      cons_a=T(ca, dt) if m else None, cons_b=T(cb, dt) if m else None)
        if level == "J":
            return Q.BatchedQP(n=n, k=k, m=m, J=T(J, dt), r=T(r, dt), lam=lam, **common)
        return Q.BatchedQP(n=n, k=k, m=m, G=T(np.tril(G).transpose(0, 2, 1), dt), c=T(c, dt), **common)

    kw = dict(initial_mu=1.0, sigma=0.1, termination_kkt_tol=2e-3, termination_complementarity_tol=1e-3, max_iterations=14,
              initial_guess_method=Q.SOLVE_EQUALITY_CONSTRAINED if k else Q.NAIVE)
    res = {}
    for label, dt, force in (("f32", torch.float32, False), ("f64", torch.float64, False), ("generic32", torch.float32, True)):
        s = Q.QPInteriorPointSolver(problem(dt), force_generic=force)
        out = {}
        s.SetVariables(T(vars_, dt))
        out["res"] = [t.double().cpu().numpy().copy() for t in s.EvaluateKKTConditions(T(mu, dt))]
        out["res_eq"] = [t.double().cpu().numpy().copy() for t in s.EvaluateKKTConditions(T(mu, dt), include_inequalities=False)]
        for strategy in (Q.COMPLEMENTARITY, Q.FIXED_DECREASE, Q.PREDICTOR_CORRECTOR):
            s.SetVariables(T(vars_, dt))
            ip, st = s.Iterate(T(mu, dt), strategy)
            assert torch.all(st == 0), (label, strategy)
            out["it", strategy] = (ip.double().cpu().numpy().copy(), s.variables().double().cpu().numpy().copy(), s.delta_.double().cpu().numpy().copy())
        for strategy in (Q.COMPLEMENTARITY, Q.PREDICTOR_CORRECTOR):
            o = s.Solve(Q.Params(barrier_strategy=strategy, **kw))
            assert torch.all(o.status == 0), (label, strategy)
            out["solve", strategy] = (s.variables().double().cpu().numpy().copy(), o.num_iterations.cpu().numpy(), o.termination_state.cpu().numpy(),
                                      o.iterations.double().cpu().numpy())
        res[label] = out
    f32, f64, g32 = res["f32"], res["f64"], res["generic32"]
    keep = np.r_[0:n, n + m:n + m + k]
    for other, tol in ((f64, 2e-4), (g32, 2e-4)):
        scale = max(1.0, np.abs(other["res"][0]).max())
        np.testing.assert_allclose(f32["res"][0], other["res"][0], rtol=tol, atol=tol * scale)
        np.testing.assert_allclose(f32["res"][1], other["res"][1], rtol=10 * tol, atol=tol * scale)
        np.testing.assert_allclose(f32["res_eq"][0][:, keep], other["res_eq"][0][:, keep], rtol=tol, atol=tol * scale)
        np.testing.assert_allclose(f32["res_eq"][1][:, [0, 2]], other["res_eq"][1][:, [0, 2]], rtol=10 * tol, atol=tol * scale)
        for strategy in (Q.COMPLEMENTARITY, Q.FIXED_DECREASE, Q.PREDICTOR_CORRECTOR):
            a_, b_ = f32["it", strategy], other["it", strategy]
            assert rel_inf_rows(a_[2], b_[2]).max() < TOL32, strategy                    # delta_
            np.testing.assert_allclose(a_[0], b_[0], rtol=5e-3, atol=5e-3, equal_nan=True)  # mu, alphas, probe alphas, mu_affine
            assert rel_inf_rows(a_[1], b_[1]).max() < 5e-3, strategy                     # state after the step
    for strategy in (Q.COMPLEMENTARITY, Q.PREDICTOR_CORRECTOR):
        xs_, nit, tm, rec = f32["solve", strategy]
        x64, nit64, tm64, _ = f64["solve", strategy]
        conv = (tm == Q.SATISFIED_KKT_TOL) & (tm64 == Q.SATISFIED_KKT_TOL)
        assert conv.mean() >= 0.75, (strategy, tm, tm64)
        # iteration counts: within 2 of the fp64 run, or of the other fp32 implementation where fp32 rounding sits on the termination threshold
        # (tiny problems at these loose tolerances: n = 8 stops after 2 - 3 iterations in fp64, one problem needs 6 in fp32 on either fp32 kernel)
        nit_g = g32["solve", strategy][1]
        off = np.minimum(np.abs(nit - nit64), np.abs(nit - nit_g))
        assert off[conv].max() <= 2, (nit, nit64, nit_g)
        # ... and EXACTLY the fp64 oracle's count (and termination state) wherever every decision of the oracle's run on these fp32-rounded
        # inputs was clear of its threshold by more than 1e-3 relative -- fp32 rounding (1e-7 relative, amplified by the conditioning of a
        # late iterate) cannot flip such a decision
        from oracle import margins as M
        clear = 0
        for p in np.flatnonzero(conv):
            qp = orc.QP(G=np.tril(G[p]), c=c[p], A_eq=A[p].T if k else None, b_eq=b[p] if k else None, cons_var=cv[p], cons_a=ca[p], cons_b=cb[p])
            oterm, onit, _, marg = M.solve_with_margins(qp, barrier_strategy=strategy, **kw)
            if M.min_margin(marg)[0] > 1e-3:
                clear += 1
                assert (int(tm[p]), int(nit[p])) == (oterm, onit), (strategy, p, int(tm[p]), int(nit[p]), oterm, onit, M.min_margin(marg))
        print(f"fp32 Solve n={n} strategy {strategy}: {clear} of {int(conv.sum())} converged problems have all oracle margins > 1e-3: iteration-exact")
        assert np.max(np.abs(xs_[conv][:, :n] - x64[conv][:, :n])) <= 5e-3 * max(1.0, np.abs(x64[:, :n]).max())
        # the records of the first iteration follow the fp64 ones
        _, _, _, rec64 = f64["solve", strategy]
        np.testing.assert_allclose(rec[:, 0, 8:11], rec64[:, 0, 8:11], rtol=5e-3, atol=5e-3)
    s = Q.QPInteriorPointSolver(problem(torch.float32))
    assert s.solve_kernel().startswith("fused_solve"), s.solve_kernel()


@pytest.mark.parametrize("n,m_r", [(63, 131), (33, 40), (7, 9), (64, 128), (101, 150), (128, 64), (40, 37)])
@pytest.mark.parametrize("layout", ["packed", "col", "rowld", "unaligned"])
def test_fused_linearize_takes_every_layout_of_J(n, m_r, layout):
    """mo_linearize in fp64 on the fused kernels for every layout of J the C ABI accepts and for odd n (the gather stream): G = J^T J + lambda I
    (lower triangle, strict upper exactly zero), c = J^T r, 0.5 |r|^2 against the dense products, and bit-compatible with the generic kernel's tolerance."""
    rng = np.random.default_rng(n * 3 + m_r)
    B = 7
    J = rng.uniform(-1, 1, (B, m_r, n)); r = rng.uniform(-1, 1, (B, m_r))
    lam = 0.125
    kw = {}
    if layout == "packed":
        Jt = T(J)
    elif layout == "rowld":
        wide = np.full((B, m_r, n + 3), 7.7); wide[:, :, :n] = J; Jt = T(wide)
    elif layout == "col":
        colw = np.full((B, n, m_r + 2), -7.7); colw[:, :, :m_r] = J.transpose(0, 2, 1); Jt = T(colw); kw = dict(J_layout="col", J_rows=m_r)
    else:
        flat = torch.zeros(B * m_r * n + 1, dtype=torch.float64, device="cuda:0"); flat[1:] = T(J).reshape(-1); Jt = flat[1:].view(B, m_r, n)
    prob = Q.BatchedQP(n=n, k=0, m=0, J=Jt, r=T(r), lam=lam, **kw)
    G, c, half = Q.linearize(prob)
    Gn = G.cpu().numpy().transpose(0, 2, 1)
    ref = np.einsum("bqi,bqj->bij", J, J) + lam * np.eye(n)
    assert np.all(np.triu(Gn, 1) == 0.0)
    np.testing.assert_allclose(np.tril(Gn), np.tril(ref), rtol=0, atol=1e-12 * m_r)
    np.testing.assert_allclose(c.cpu().numpy(), np.einsum("bqi,bq->bi", J, r), rtol=0, atol=1e-12 * m_r)
    np.testing.assert_allclose(half.cpu().numpy(), 0.5 * np.einsum("bq,bq->b", r, r), rtol=1e-13)


@pytest.mark.parametrize("n,m_r", [(128, 256), (64, 128), (128, 4), (64, 260)])
def test_fused_f32_linearize(n, m_r):
    """mo_linearize in fp32 on the fused kernel: G = J^T J + lambda I (lower triangle; strict upper exactly zero, residual.hpp:216-220),
    c = J^T r, 0.5 |r|^2 -- against the fp64 products of the fp32-rounded inputs and against the generic fp32 kernel."""
    rng = np.random.default_rng(n + m_r)
    B = 9
    f = lambda a: a.astype(np.float32).astype(np.float64)
    J = f(rng.uniform(-1, 1, (B, m_r, n))); r = f(rng.uniform(-1, 1, (B, m_r)))
    lam = float(np.float32(0.25))
    prob = Q.BatchedQP(n=n, k=0, m=0, J=T(J, torch.float32), r=T(r, torch.float32), lam=lam)
    G, c, half = Q.linearize(prob)
    Gg, cg, halfg = Q.linearize(prob, force_generic=True)
    Gn = G.double().cpu().numpy().transpose(0, 2, 1)                  # [B, row, col] of the column-major output
    ref = np.einsum("bqi,bqj->bij", J, J) + lam * np.eye(n)
    scale = np.abs(ref).max()
    assert np.all(np.triu(Gn, 1) == 0.0)
    np.testing.assert_allclose(np.tril(Gn), np.tril(ref), rtol=0, atol=2e-6 * scale)
    np.testing.assert_allclose(c.double().cpu().numpy(), np.einsum("bqi,bq->bi", J, r), rtol=0, atol=2e-6 * m_r)
    np.testing.assert_allclose(half.double().cpu().numpy(), 0.5 * np.einsum("bq,bq->b", r, r), rtol=2e-6)
    np.testing.assert_allclose(G.cpu().numpy(), Gg.cpu().numpy(), rtol=0, atol=2e-6 * scale)
    np.testing.assert_allclose(c.cpu().numpy(), cg.cpu().numpy(), rtol=0, atol=2e-6 * m_r)
    np.testing.assert_allclose(half.cpu().numpy(), halfg.cpu().numpy(), rtol=2e-6)


@pytest.mark.parametrize("which", ["f32_cfg4", "f64_two_y_tiles", "f64_one_tile", "f64_four_slots"])
@pytest.mark.parametrize("batch", [1, 12])
def test_status_words_of_the_new_solve_kernels(which, batch):
    """The failure channels of the reference (F_ASSERT s > 0 qp.cc:285, constraint index qp.cc:70-72, FailedFactorization / a NaN
    in the data) inside the fp32 Solve kernel and the two-y-tile fp64 kernels: per-problem status words from Solve, Iterate and the
    step, the healthy problems untouched; batch = 1 exercises the ticket loop with fewer problems than waves."""
    if which == "f32_cfg4":
        d = synth.CONFIGS["cfg4"]; n, k, m, m_r, dt = d["n"], d["k"], d["m"], d["m_r"], torch.float32
    elif which == "f64_two_y_tiles":
        n, k, m, m_r, dt = 64, 24, 40, 128, torch.float64
    elif which == "f64_one_tile":
        n, k, m, m_r, dt = 8, 2, 4, 16, torch.float64          # BASELINE configs[0]: kkt_fused_tiny.hip
    else:
        n, k, m, m_r, dt = 96, 6, 160, 192, torch.float64      # kkt_fused_mc4.hip
    hb = synth.make_batch(n, k, m, m_r, 12, stream=5)
    hb.vars[1, n] = 0.0                 # s = 0
    hb.cons_var[4, 0] = n + 7           # index beyond n
    hb.J[5, 3, 2] = np.nan              # NaN in the data
    hb.cons_var[6, 1] = -1              # negative index
    sel = slice(0, 12) if batch == 12 else slice(1, 2)
    cut = lambda a_: a_[sel]
    prob = Q.BatchedQP(n=n, k=k, m=m, J=T(cut(hb.J), dt), r=T(cut(hb.r), dt), lam=hb.lam, A_eq=T(cut(hb.A_eq), dt), b_eq=T(cut(hb.b_eq), dt),
                       cons_var=T(cut(hb.cons_var), torch.int32), cons_a=T(cut(hb.cons_a), dt), cons_b=T(cut(hb.cons_b), dt))
    s = Q.QPInteriorPointSolver(prob)
    assert s.solve_kernel().startswith("fused_solve"), s.solve_kernel()
    assert s.step_kernel().startswith("fused"), s.step_kernel()
    checks = []
    calls = (lambda: s.NewtonStep(T(cut(hb.mu), dt), 0.995)[2],
             lambda: s.Iterate(T(cut(hb.mu), dt), Q.PREDICTOR_CORRECTOR)[1],
             lambda: s.Solve(Q.Params(initial_guess_method=Q.USER_PROVIDED, max_iterations=4, termination_kkt_tol=1e-3)).status)
    for call in calls:
        s.SetVariables(T(cut(hb.vars), dt))
        if batch == 1:                                           # Solve on a single problem re-throws like the reference (F_ASSERT qp.cc:285)
            try:
                call()
            except L.MiniOptError as e:
                assert "NONPOSITIVE_SLACK" in str(e)
            checks.append(s.status_.cpu().numpy().copy())
        else:
            checks.append(call().cpu().numpy().copy())
    for st in checks:
        if batch == 1:
            assert st.tolist() == [L.MO_STATUS_NONPOSITIVE_SLACK]
            continue
        good = [0, 2, 3, 7, 8, 9, 10, 11]
        assert np.all(st[good] == 0), st
        assert st[1] == L.MO_STATUS_NONPOSITIVE_SLACK and st[4] == L.MO_STATUS_BAD_INDEX and st[6] == L.MO_STATUS_BAD_INDEX
        assert st[5] in (L.MO_STATUS_NONFINITE, L.MO_STATUS_FACTORIZATION_FAILED)
    if batch == 12:
        v = s.variables().double().cpu().numpy()
        assert np.all(np.isfinite(v[[0, 2, 3, 7, 8, 9, 10, 11]]))


# ------------------------------------------------------------------ fused fp64 kernels on sizes that are padded to the tile grid
@pytest.mark.parametrize("n,k,m,m_r,level", [(20, 2, 6, 24, "J"), (34, 4, 10, 40, "J"), (46, 0, 8, 48, "J"), (62, 14, 2, 64, "J"),
                                             (2, 0, 4, 0, "QP"), (7, 2, 3, 0, "QP"), (33, 5, 12, 0, "QP"), (50, 8, 0, 0, "QP")])
def test_fused_padded_sizes(n, k, m, m_r, level):
    """Any n <= 64 runs on the fused kernels (even n for J-level input): the system is padded to 32 / 64 variables with a unit
    diagonal inside the kernel.  Newton step and the whole Solve against the oracle."""
    rng = np.random.default_rng(n * 131 + k * 17 + m)
    B = 23
    mr = m_r if m_r else 2 * n
    J = rng.uniform(-1, 1, (B, mr, n)); r = rng.uniform(-1, 1, (B, mr))
    A = rng.uniform(-1, 1, (B, n, k)); b = rng.uniform(-1, 1, (B, k))
    cv = rng.integers(0, n, (B, m)).astype(np.int32)
    ca = rng.choice([-1.0, 1.0, 2.0], (B, m)); cb = rng.uniform(0.5, 2.0, (B, m))
    x = rng.uniform(-0.1, 0.1, (B, n))
    sl = rng.uniform(0.2, 1.5, (B, m)); z = rng.uniform(0.1, 2, (B, m)); y = rng.uniform(-1, 1, (B, k))
    vars_ = np.concatenate([x, sl, y, z], axis=1)
    mu = np.full(B, 0.05)
    lam = 1e-3
    G = np.einsum("bqi,bqj->bij", J, J) + lam * np.eye(n)
    c = np.einsum("bqi,bq->bi", J, r)
    common = dict(A_eq=T(A) if k else None, b_eq=T(b) if k else None, cons_var=T(cv, torch.int32) if m else None,
                  cons_a=T(ca) if m else None, cons_b=T(cb) if m else None)
    if level == "J":
        prob = Q.BatchedQP(n=n, k=k, m=m, J=T(J), r=T(r), lam=lam, **common)
    else:
        prob = Q.BatchedQP(n=n, k=k, m=m, G=T(np.tril(G).transpose(0, 2, 1)), c=T(c), **common)
    s = Q.QPInteriorPointSolver(prob)
    assert s.step_kernel().startswith("fused"), s.step_kernel()
    s.SetVariables(T(vars_))
    delta, alpha, status = s.NewtonStep(T(mu), 0.995)
    ref, ref_alpha, ref_status, _ = orc.batched_newton_step(
        n, k, m, G=np.tril(G).transpose(0, 2, 1).copy(), c=c, A_eq=A if k else None, b_eq=b if k else None, cons_var=cv if m else None,
        cons_a=ca if m else None, cons_b=cb if m else None, vars_=vars_, mu=mu)
    assert torch.all(status == 0) and np.all(ref_status == 0)
    assert rel_inf_rows(delta.cpu().numpy(), ref).max() < 1e-9
    np.testing.assert_allclose(alpha.cpu().numpy(), ref_alpha, atol=1e-9)
    # the whole Solve, fused vs generic (the generic kernel is pinned against the oracle elsewhere)
    kw = dict(initial_mu=1.0, sigma=0.1, termination_kkt_tol=1e-9, max_iterations=12, initial_guess_method=Q.SOLVE_EQUALITY_CONSTRAINED if k else Q.NAIVE)
    res = {}
    for force in (False, True):
        sv = Q.QPInteriorPointSolver(prob, force_generic=force)
        out = sv.Solve(Q.Params(**kw))
        assert torch.all(out.status == 0)
        res[force] = (sv.variables().cpu().numpy().copy(), out.num_iterations.cpu().numpy(), out.termination_state.cpu().numpy())
    make_qp = lambda p: orc.QP(G=np.tril(G[p]), c=c[p], A_eq=A[p].T if k else None, b_eq=b[p] if k else None, cons_var=cv[p], cons_a=ca[p], cons_b=cb[p])
    same = solves_agree_or_knife_edge((n, k, m, m_r, level), make_qp, kw, {"fused": (res[False][2], res[False][1]), "generic": (res[True][2], res[True][1])})
    np.testing.assert_allclose(res[False][0][same], res[True][0][same], rtol=1e-6, atol=1e-8)


# ------------------------------------------------------------------ randomised shape sweep: fused vs generic kernel (the generic one is pinned above)
def test_fused_vs_generic_random_shapes():
    """Forty random (n, k, m, m_r, input level, strategy) combinations through the fused kernels and through the generic kernel
    (MO_PLAN_FORCE_GENERIC): Newton step, Iterate (incl. predictor-corrector) and Solve must agree."""
    rng = np.random.default_rng(2026)
    for _ in range(40):
        level = rng.choice(["J", "QP"])
        n = int(rng.integers(2, 65))                          # any n; odd n with J-level input takes the flat-group stream
        k = int(rng.integers(0, min(16, n)))                  # up to 15 equalities: one y tile
        m = int(rng.integers(0, 65))
        m_r = int(rng.integers(1, 160))                       # any row count: a partial last 4-row group included
        _fused_vs_generic_case(rng, level, n, k, m, m_r)


def test_fused_two_y_tiles_vs_generic_random_shapes():
    """16 <= k <= 31 equality constraints: the fused kernels carry a second y tile (kkt_fused_ny2.hip).  Thirty random shapes on the
    32 / 64 / 96 tile grids (m up to 128 where the one-tile kernels take it, odd n through the gather stream) against the generic kernel."""
    rng = np.random.default_rng(3031)
    for i in range(30):
        level = rng.choice(["J", "QP"])
        n = int(rng.integers(16, 65)) if i % 5 else int(rng.integers(65, 89))   # n + k <= 119: the LDS-resident generic kernel still holds it
        k = int(rng.integers(16, min(32, n + 1)))
        odd_stream = level == "J" and (n & 1)
        m = int(rng.integers(0, 65 if (odd_stream or n > 64) else 129))
        m_r = int(rng.integers(1, 160))
        _fused_vs_generic_case(rng, level, n, k, m, m_r, feasible=True, kkt_tol=1e-6)


def test_fused_three_and_four_y_tiles_vs_generic_random_shapes():
    """32 <= k <= 63 equality constraints on the 32 / 64 tile grids: the fused kernels carry three / four y tiles (kkt_fused_ny34.hip; packed
    even-n J or (G, c), m <= 128).  Twenty random shapes against the generic kernel (pinned to the oracle elsewhere; the knife-edge rule
    with the oracle as referee for the Solves), plus the corner shapes k = 32, 47, 48, 63."""
    rng = np.random.default_rng(4041)
    shapes = [(64, 32), (64, 47), (64, 48), (64, 63), (36, 32), (40, 33)]
    for i in range(14):
        k = int(rng.integers(32, 60))
        n = 2 * int(rng.integers((k + 5) // 2, 33))             # even n in [k + 4, 64]: the equalities leave the inequalities something to do
        shapes.append((n, k))                                    # (k = n fixes x: with inequalities on top there is no trajectory to compare)
    for n, k in shapes:
        level = rng.choice(["J", "QP"])
        m = int(rng.integers(0, 129))
        m_r = int(rng.integers(1, 160))
        _fused_vs_generic_case(rng, level, n, k, m, m_r, feasible=True, kkt_tol=1e-6)


@pytest.mark.parametrize("level,n,k,m,m_r", [("J", 15, 0, 0, 15), ("J", 13, 2, 64, 64), ("J", 2, 1, 1, 1), ("J", 8, 7, 5, 3), ("QP", 14, 1, 64, 0),
                                               ("QP", 2, 0, 0, 0), ("J", 8, 2, 4, 16), ("J", 5, 0, 64, 64), ("QP", 9, 6, 30, 0)])
def test_one_tile_kernel_at_its_boundaries(level, n, k, m, m_r):
    """kkt_fused_tiny.hip at the edges of its range: n + k = 15 (the last free tile column carries the right-hand side), k = 0, m = 0,
    m = 64, m_r = 64 and 1, k = n - 1, and BASELINE configs[0] itself (n = 8, 2 equalities, 4 box entries) -- step, step without
    inequalities, KKT residual, Iterate and Solve against the generic kernel (pinned to the oracle), and the same problems on the
    32-variable tile grid (MO_PLAN_NO_TINY) must give the same Solve."""
    rng = np.random.default_rng(100 * n + 10 * k + m)
    _fused_vs_generic_case(rng, level, n, k, m, m_r if m_r else 2 * n, feasible=True, expect_kernel="tiny")


def _fused_vs_generic_case(rng, level, n, k, m, m_r, feasible=False, kkt_tol=1e-8, expect_kernel=None):
        B = 9
        J = rng.uniform(-1, 1, (B, m_r, n)); r = rng.uniform(-1, 1, (B, m_r))
        A = rng.uniform(-1, 1, (B, n, k)); b = rng.uniform(-1, 1, (B, k))
        cv = rng.integers(0, n, (B, m)).astype(np.int32)
        ca = rng.choice([-1.0, 1.0, 2.0], (B, m)); cb = rng.uniform(0.5, 2.0, (B, m))
        if feasible:  # many equalities and up to 128 inequalities on few variables: make sure a strictly feasible point x0 exists,
            x0 = rng.uniform(-0.5, 0.5, (B, n))                 # with margins small enough that some inequalities are active at the
            b = -np.einsum("bik,bi->bk", A, x0)                 # optimum (none active: the predictor-corrector's full step lands on
            cb = -ca * np.take_along_axis(x0, cv.astype(np.int64), axis=1) + rng.uniform(0.05, 0.5, (B, m))  # z + dz = 0 to rounding)
        x = rng.uniform(-0.1, 0.1, (B, n))
        sl = rng.uniform(0.2, 1.5, (B, m)); z = rng.uniform(0.1, 2, (B, m)); y = rng.uniform(-1, 1, (B, k))
        vars_ = np.concatenate([x, sl, y, z], axis=1)
        mu = np.full(B, 0.05)
        lam = 0.3 if m_r < n else 1e-3
        common = dict(A_eq=T(A) if k else None, b_eq=T(b) if k else None, cons_var=T(cv, torch.int32) if m else None,
                      cons_a=T(ca) if m else None, cons_b=T(cb) if m else None)
        if level == "J":
            prob = Q.BatchedQP(n=n, k=k, m=m, J=T(J), r=T(r), lam=lam, **common)
        else:
            G = np.einsum("bqi,bqj->bij", J, J) + lam * np.eye(n)
            prob = Q.BatchedQP(n=n, k=k, m=m, G=T(np.tril(G).transpose(0, 2, 1)), c=T(np.einsum("bqi,bq->bi", J, r)), **common)
        tag = (level, n, k, m, m_r)
        strategy = int(rng.integers(0, 3))
        got = {}
        for force in (False, True):
            s = Q.QPInteriorPointSolver(prob, force_generic=force)
            if not force:
                assert s.step_kernel().startswith("fused"), (tag, s.step_kernel())
                if expect_kernel:
                    assert expect_kernel in s.step_kernel() and expect_kernel in s.solve_kernel(), (tag, s.step_kernel(), s.solve_kernel())
            s.SetVariables(T(vars_))
            delta, alpha, status = s.NewtonStep(T(mu), 0.995)
            assert torch.all(status == 0), tag
            delta = delta.clone()
            d0, a0, st0 = s.NewtonStep(T(mu), 0.995, include_inequalities=False)      # SolveForUpdateNoInequalities, qp.cc:366-386
            assert torch.all(st0 == 0), tag
            res_all = [t.cpu().numpy().copy() for t in s.EvaluateKKTConditions(T(mu))]              # qp.cc:391-437
            res_eq = [t.cpu().numpy().copy() for t in s.EvaluateKKTConditions(T(mu), include_inequalities=False)]
            extra = (d0.cpu().numpy().copy(), a0.cpu().numpy().copy(), res_all, res_eq)
            ip, st2 = s.Iterate(T(mu), strategy)
            assert torch.all(st2 == 0), tag
            after_iter = s.variables().cpu().numpy().copy()
            out = s.Solve(Q.Params(initial_mu=1.0, sigma=0.1, termination_kkt_tol=kkt_tol, max_iterations=10, barrier_strategy=strategy,
                                   initial_guess_method=Q.SOLVE_EQUALITY_CONSTRAINED if k else Q.NAIVE))
            assert torch.all(out.status == 0), tag
            got[force] = (delta.cpu().numpy(), alpha.cpu().numpy(), ip.cpu().numpy(), after_iter, s.variables().cpu().numpy().copy(),
                          out.num_iterations.cpu().numpy(), out.termination_state.cpu().numpy(), extra)
        f, g_ = got[False], got[True]
        assert rel_inf_rows(f[0], g_[0]).max() < 1e-8, tag
        # ... and both against the ORACLE's step (BASELINE tolerance: 1e-10 rel-inf): the kernels with three / four y tiles and the one-tile
        # kernel's boundary shapes are pinned directly, not through the generic kernel
        Gd = np.einsum("bqi,bqj->bij", J, J) + lam * np.eye(n); cd = np.einsum("bqi,bq->bi", J, r)
        ref, ref_alpha, ref_status, _ = orc.batched_newton_step(n, k, m, G=np.tril(Gd).transpose(0, 2, 1).copy(), c=cd, A_eq=A if k else None, b_eq=b if k else None,
                                                                cons_var=cv if m else None, cons_a=ca if m else None, cons_b=cb if m else None, vars_=vars_, mu=mu)
        assert np.all(ref_status == 0), tag
        assert rel_inf_rows(f[0], ref).max() < TOL64, (tag, rel_inf_rows(f[0], ref).max())
        assert rel_inf_rows(g_[0], ref).max() < TOL64, (tag, rel_inf_rows(g_[0], ref).max())
        np.testing.assert_allclose(f[1], ref_alpha, atol=1e-9, err_msg=str(tag))
        # the step without inequalities (dx, dy only; ds = dz = 0; alpha = 1) and the KKT residual with its four norms, fused vs generic
        (fd0, fa0, fr, fre), (gd0, ga0, gr, gre) = f[7], g_[7]
        np.testing.assert_allclose(fd0, gd0, rtol=1e-8, atol=1e-9 * max(1.0, np.abs(gd0).max()), err_msg=str(tag))
        assert np.all(fd0[:, n:n + m] == 0) and np.all(fd0[:, n + m + k:] == 0) and np.all(fa0 == 1.0) and np.all(ga0 == 1.0), tag
        scale = max(1.0, np.abs(gr[0]).max())
        np.testing.assert_allclose(fr[0], gr[0], rtol=1e-10, atol=1e-12 * scale, err_msg=str(tag))
        np.testing.assert_allclose(fr[1], gr[1], rtol=1e-9, atol=1e-12 * scale, err_msg=str(tag))
        keep = np.r_[0:n, n + m:n + m + k]                                            # r_d and r_pe are what the flag leaves defined
        np.testing.assert_allclose(fre[0][:, keep], gre[0][:, keep], rtol=1e-10, atol=1e-12 * scale, err_msg=str(tag))
        np.testing.assert_allclose(fre[1][:, [0, 2]], gre[1][:, [0, 2]], rtol=1e-9, atol=1e-12 * scale, err_msg=str(tag))
        np.testing.assert_allclose(f[1], g_[1], atol=1e-8, err_msg=str(tag))
        np.testing.assert_allclose(f[2], g_[2], rtol=1e-6, atol=1e-9, equal_nan=True, err_msg=str(tag))
        np.testing.assert_allclose(f[3], g_[3], rtol=1e-7, atol=1e-9, err_msg=str(tag))
        make_qp = lambda p: orc.QP(G=np.tril(Gd[p]), c=cd[p], A_eq=A[p].T if k else None, b_eq=b[p] if k else None, cons_var=cv[p], cons_a=ca[p], cons_b=cb[p])
        solve_kw = dict(initial_mu=1.0, sigma=0.1, termination_kkt_tol=kkt_tol, max_iterations=10, barrier_strategy=strategy,
                        initial_guess_method=orc.GUESS_SOLVE_EQUALITY_CONSTRAINED if k else orc.GUESS_NAIVE)
        same = solves_agree_or_knife_edge(tag, make_qp, solve_kw, {"fused": (f[6], f[5]), "generic": (g_[6], g_[5])})   # (B = 9: at most one problem per shape, 5 % over the sweep)
        # optimum: x of the problems that converged (random constraint sets can be nearly degenerate: multipliers reach 1e12, the
        # interior-point loop runs into MAX_ITERATIONS and the two summation orders drift apart there)
        conv = same & (f[6] == Q.SATISFIED_KKT_TOL)
        if conv.any():
            xa, xb_ = f[4][conv][:, :n], g_[4][conv][:, :n]
            assert np.max(np.abs(xa - xb_)) <= 1e-5 * max(1.0, np.max(np.abs(xb_))), tag
        if expect_kernel == "tiny":   # the same problems on the 32-variable grid
            s32 = Q.QPInteriorPointSolver(prob, no_tiny=True)
            assert "tiny" not in s32.solve_kernel() and s32.solve_kernel().startswith("fused"), s32.solve_kernel()
            o32 = s32.Solve(Q.Params(initial_mu=1.0, sigma=0.1, termination_kkt_tol=kkt_tol, max_iterations=10, barrier_strategy=strategy,
                                     initial_guess_method=Q.SOLVE_EQUALITY_CONSTRAINED if k else Q.NAIVE))
            same32 = solves_agree_or_knife_edge(tag + ("32 grid",), make_qp, solve_kw,
                                                {"one_tile": (f[6], f[5]), "grid32": (o32.termination_state.cpu().numpy(), o32.num_iterations.cpu().numpy())})
            c32 = same32 & (f[6] == Q.SATISFIED_KKT_TOL)
            if c32.any():
                x32 = s32.variables().cpu().numpy()[c32][:, :n]
                assert np.max(np.abs(x32 - f[4][c32][:, :n])) <= 1e-5 * max(1.0, np.max(np.abs(x32))), tag


@pytest.mark.parametrize("n,k,m,m_r,level", [(66, 4, 10, 72, "J"), (96, 8, 32, 192, "J"), (100, 14, 64, 200, "J"), (128, 10, 40, 256, "J"),
                                             (81, 3, 20, 0, "QP"), (128, 14, 64, 0, "QP"),
                                             (96, 20, 32, 192, "J"), (128, 31, 64, 256, "J"), (128, 16, 30, 0, "QP"), (40, 16, 12, 80, "J"),
                                             (64, 31, 64, 128, "J"), (127, 17, 20, 130, "J"),
                                             # round 4: three y tiles on the 96 grid (32 <= k <= 47 up to n = 96)
                                             (96, 40, 32, 192, "J"), (80, 47, 20, 0, "QP"), (66, 32, 64, 140, "J"), (90, 36, 128, 100, "J"),
                                             # ... four on the 96 grid (48 <= k <= 63), three on the 128 grid (32 <= k <= 47)
                                             (96, 63, 32, 200, "J"), (70, 48, 20, 0, "QP"), (128, 47, 64, 256, "J"), (100, 32, 16, 0, "QP"), (128, 40, 128, 140, "J"),
                                             # ... and two y tiles with up to four constraint slots per lane on every grid (a box on each variable beside 16 .. 31 equalities)
                                             (128, 20, 256, 260, "J"), (100, 31, 200, 0, "QP"), (64, 24, 256, 128, "J"), (96, 16, 100, 192, "J"), (32, 16, 200, 64, "J"),
                                             (128, 16, 128, 0, "QP"),
                                             # ... and three / four y tiles with four slots (m up to 256 beside 32 .. 63 equalities)
                                             (100, 40, 200, 210, "J"), (64, 63, 256, 0, "QP"), (96, 48, 150, 192, "J"), (30, 15, 60, 64, "J")])
def test_fused_fp64_up_to_128_variables(n, k, m, m_r, level):
    """The 96- and 128-variable tile grids of the fp64 fused kernels (sizes the LDS-resident generic kernel cannot hold at all once
    n + k > 141): Newton step against the oracle, and the whole Solve against the oracle's Solve."""
    rng = np.random.default_rng(n * 7 + k)
    B = 7
    mr = m_r if m_r else 2 * n
    J = rng.uniform(-1, 1, (B, mr, n)); r = rng.uniform(-1, 1, (B, mr))
    A = rng.uniform(-1, 1, (B, n, k)); b = rng.uniform(-1, 1, (B, k))
    cv = rng.integers(0, n, (B, m)).astype(np.int32)
    ca = rng.choice([-1.0, 1.0, 2.0], (B, m)); cb = rng.uniform(0.5, 2.0, (B, m))
    x = rng.uniform(-0.1, 0.1, (B, n))
    sl = rng.uniform(0.2, 1.5, (B, m)); z = rng.uniform(0.1, 2, (B, m)); y = rng.uniform(-1, 1, (B, k))
    vars_ = np.concatenate([x, sl, y, z], axis=1)
    mu = np.full(B, 0.05)
    lam = 1e-3
    G = np.einsum("bqi,bqj->bij", J, J) + lam * np.eye(n)
    c = np.einsum("bqi,bq->bi", J, r)
    common = dict(A_eq=T(A), b_eq=T(b), cons_var=T(cv, torch.int32), cons_a=T(ca), cons_b=T(cb))
    if level == "J":
        prob = Q.BatchedQP(n=n, k=k, m=m, J=T(J), r=T(r), lam=lam, **common)
    else:
        prob = Q.BatchedQP(n=n, k=k, m=m, G=T(np.tril(G).transpose(0, 2, 1)), c=T(c), **common)
    s = Q.QPInteriorPointSolver(prob)
    assert s.step_kernel().startswith("fused"), s.step_kernel()
    s.SetVariables(T(vars_))
    delta, alpha, status = s.NewtonStep(T(mu), 0.995)
    ref, ref_alpha, ref_status, _ = orc.batched_newton_step(n, k, m, G=np.tril(G).transpose(0, 2, 1).copy(), c=c, A_eq=A, b_eq=b, cons_var=cv,
                                                            cons_a=ca, cons_b=cb, vars_=vars_, mu=mu)
    assert torch.all(status == 0) and np.all(ref_status == 0)
    assert rel_inf_rows(delta.cpu().numpy(), ref).max() < 1e-9
    np.testing.assert_allclose(alpha.cpu().numpy(), ref_alpha, atol=1e-9)
    kw = dict(initial_mu=1.0, sigma=0.1, termination_kkt_tol=1e-8, max_iterations=12, initial_guess_method=Q.SOLVE_EQUALITY_CONSTRAINED)
    out = s.Solve(Q.Params(**kw))
    assert torch.all(out.status == 0)
    v = s.variables().cpu().numpy(); nit = out.num_iterations.cpu().numpy(); tm = out.termination_state.cpu().numpy()
    make_qp = lambda p: orc.QP(G=np.tril(G[p]), c=c[p], A_eq=A[p].T, b_eq=b[p], cons_var=cv[p], cons_a=ca[p], cons_b=cb[p])
    same = solves_agree_or_knife_edge((n, k, m, m_r, level), make_qp, kw, {"fused": (tm, nit)})   # (B = 7: at most one problem, and only on a logged knife edge)
    for p in np.flatnonzero(same):
        o = orc.Solver(make_qp(p))
        term, its = o.solve(**kw)
        if term == Q.SATISFIED_KKT_TOL:
            np.testing.assert_allclose(v[p][:n], o.variables[:n], rtol=1e-6, atol=1e-8)


def test_large_fp64_plan_runs_on_both_kernel_families():
    """n = 128, k = 14 in fp64 does not fit the LDS-resident generic kernel (H alone is 161 KB): the fused kernels serve it, and a call that is
    forced onto the generic kernel now runs there too with H in its global workspace (it used to return MO_ERR_UNSUPPORTED)."""
    rng = np.random.default_rng(1)
    n, k, m, m_r, B = 128, 14, 4, 256, 3
    J = rng.uniform(-1, 1, (B, m_r, n)); r = rng.uniform(-1, 1, (B, m_r))
    A = rng.uniform(-1, 1, (B, n, k)); b = rng.uniform(-1, 1, (B, k))
    cv = rng.integers(0, n, (B, m)).astype(np.int32); ca = np.ones((B, m)); cb = np.ones((B, m))
    vars_ = np.concatenate([np.zeros((B, n)), np.ones((B, m)), np.zeros((B, k)), np.ones((B, m))], axis=1)
    prob = Q.BatchedQP(n=n, k=k, m=m, J=T(J), r=T(r), lam=1e-3, A_eq=T(A), b_eq=T(b), cons_var=T(cv, torch.int32), cons_a=T(ca), cons_b=T(cb))
    s = Q.QPInteriorPointSolver(prob)
    s.SetVariables(T(vars_))
    delta, alpha, status = s.NewtonStep(0.1, 0.995)
    assert torch.all(status == 0) and torch.isfinite(delta).all()
    delta = delta.clone()
    d0, a0, st0 = s.NewtonStep(0.1, 0.995, include_inequalities=False)   # MO_STEP_NO_INEQUALITIES and the KKT residual run fused as well
    assert torch.all(st0 == 0) and torch.isfinite(d0).all() and torch.all(a0 == 1.0)
    r_, kkt = s.EvaluateKKTConditions(0.1)
    assert torch.isfinite(r_).all() and torch.isfinite(kkt).all()
    sg = Q.QPInteriorPointSolver(prob, force_generic=True)
    assert sg.step_kernel() == "generic"
    sg.SetVariables(T(vars_))
    dg, ag, stg = sg.NewtonStep(0.1, 0.995)
    assert torch.all(stg == 0)
    assert rel_inf_rows(dg.cpu().numpy(), delta.cpu().numpy()).max() < 1e-10
    np.testing.assert_allclose(ag.cpu().numpy(), alpha.cpu().numpy(), atol=1e-10)
    G, c, half = Q.linearize(prob)
    np.testing.assert_allclose(c.cpu().numpy(), np.einsum("bqi,bq->bi", J, r), rtol=1e-12, atol=1e-12)


@pytest.mark.parametrize("n,k,zero_rows", [(200, 6, (4, 5)), (230, 40, (38, 39)), (200, 6, (2,)), (230, 40, (17,))],
                         ids=["trailing_short_panel", "trailing_in_a_16_column_panel", "middle_short_panel", "middle_of_a_panel"])
def test_large_generic_path_zero_pivot_rules(n, k, zero_rows):
    """The register panels of the LARGE factorisation (round 4) carry no zero-pivot logic: a zero pivot raises a flag every wave sees alike and
    the half panel is redone by the LDS loop that knows the rules (kkt_generic.hip, left_factor_half).  Zero rows of A_eq make exact zero
    pivots in the y block.  TRAILING zero rows are what LDLT tolerates in natural order too (zero pivot, zero column below, nothing
    behind it): status OK, the multiplier's direction comes out as D^+ leaves it, and the step must match the oracle's.  A zero row with
    non-zero pivots behind it is a failure of the natural-order factorisation: that problem alone reports FACTORIZATION_FAILED, its
    neighbours in the batch are untouched."""
    rng = np.random.default_rng(n + k + len(zero_rows))
    B, m, m_r = 4, 12, n + 10
    J = rng.uniform(-1, 1, (B, m_r, n)); r = rng.uniform(-1, 1, (B, m_r))
    A = rng.uniform(-1, 1, (B, n, k)); b = rng.uniform(-1, 1, (B, k))
    bad = 2                                                    # the problem with the zero rows
    for q in zero_rows:
        A[bad, :, q] = 0.0; b[bad, q] = 0.0
    cv = rng.integers(0, n, (B, m)).astype(np.int32); ca = rng.choice([-1.0, 1.0], (B, m)); cb = rng.uniform(0.5, 2.0, (B, m))
    x = rng.uniform(-0.1, 0.1, (B, n)); sl = rng.uniform(0.2, 1.5, (B, m)); z = rng.uniform(0.1, 2, (B, m)); y = rng.uniform(-1, 1, (B, k))
    vars_ = np.concatenate([x, sl, y, z], axis=1); mu = np.full(B, 0.05)
    prob = Q.BatchedQP(n=n, k=k, m=m, J=T(J), r=T(r), lam=1e-3, A_eq=T(A), b_eq=T(b), cons_var=T(cv, torch.int32), cons_a=T(ca), cons_b=T(cb))
    s = Q.QPInteriorPointSolver(prob)
    assert s.step_kernel() == "generic"
    s.SetVariables(T(vars_))
    delta, alpha, status = s.NewtonStep(T(mu), 0.995)
    st = status.cpu().numpy(); d = delta.cpu().numpy()
    ref, ref_alpha, ref_status, _ = orc.batched_newton_step(n, k, m, J=J, r=r, lam=1e-3, A_eq=A, b_eq=b, cons_var=cv, cons_a=ca, cons_b=cb, vars_=vars_, mu=mu)
    others = [p for p in range(B) if p != bad]
    assert np.all(st[others] == 0) and np.all(ref_status[others] == 0)
    assert rel_inf_rows(d[others], ref[others]).max() < TOL64
    trailing = min(zero_rows) == k - len(zero_rows)
    if trailing:
        assert st[bad] == 0 and ref_status[bad] == 0, (st[bad], ref_status[bad])
        assert rel_inf_rows(d[bad:bad + 1], ref[bad:bad + 1]).max() < TOL64
        np.testing.assert_allclose(alpha.cpu().numpy()[bad], ref_alpha[bad], atol=1e-9)
    else:
        assert st[bad] == L.MO_STATUS_FACTORIZATION_FAILED, st[bad]
        assert np.all(np.isnan(d[bad]))


@pytest.mark.parametrize("n,k,m,m_r,level,dt", [(256, 40, 128, 300, "J", torch.float64), (256, 40, 128, 0, "QP", torch.float64), (200, 0, 30, 210, "J", torch.float64),
                                                (160, 50, 0, 170, "J", torch.float64), (300, 20, 64, 310, "J", torch.float32), (130, 70, 17, 0, "QP", torch.float64),
                                                # shapes at the edges of the matrix-core J^T J / register-panel code (round 3): three rows of J (less than the four one
                                                # MFMA consumes), n + k just beyond 192, three full 128-wide super-blocks with two rows per thread in the panel, a first
                                                # panel of more than 512 rows (LDS panel path) followed by register panels, fp32 with a ragged second super-block
                                                (145, 0, 0, 3, "J", torch.float64), (190, 3, 2, 193, "J", torch.float64), (384, 16, 32, 390, "J", torch.float64),
                                                (520, 8, 16, 64, "J", torch.float64), (260, 0, 10, 270, "J", torch.float32),
                                                # (round 4) a system so large that only 8 panel columns fit the LDS beside its vectors: the left-looking
                                                # update then runs in 8-column blocks whose finished columns are not a multiple of 16 (the tail loop)
                                                (960, 8, 16, 0, "J", torch.float64),
                                                # mid-size systems the LARGE path serves since n + k >= 72 goes there: k beyond the fused kernels, both precisions
                                                (90, 70, 17, 100, "J", torch.float64), (90, 40, 17, 100, "J", torch.float32), (60, 12, 300, 0, "QP", torch.float64)],
                         ids=["n256_J", "n256_QP", "n200_no_eq", "n160_no_ineq", "n300_f32", "n130_k70_QP", "n145_three_rows", "n190_k3", "n384", "n520_short_J", "n260_f32",
                              "n960_panel8", "n90_k70", "n90_k40_f32", "n60_m300_QP"])
def test_sizes_beyond_every_lds_resident_kernel(n, k, m, m_r, level, dt):
    """The reference resizes its solver to any N, K (qp.cc:36-48).  Beyond the fused kernels (n <= 128, k <= 31) and the LDS-resident generic
    kernel (n + k <= 71 since round 4) the generic kernel keeps H in a global workspace of its workgroup: left-looking blocked LDL^T (half
    panels staged in LDS and factorised in registers), 128-wide J^T J super-blocks on the matrix cores.  Step (with and without inequalities), KKT residual,
    Iterate and the whole Solve against the oracle: directions within 1e-10 rel-inf, Solve iteration-exact (fp32: against the oracle on
    fp32-rounded inputs at fp32 tolerances)."""
    rng = np.random.default_rng(n + 7 * k + m)
    B = 3
    f64 = dt == torch.float64
    f = (lambda a_: a_) if f64 else (lambda a_: a_.astype(np.float32).astype(np.float64))
    mr = m_r if m_r else n + 20
    J = f(rng.uniform(-1, 1, (B, mr, n))); r = f(rng.uniform(-1, 1, (B, mr)))
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
    prob = (Q.BatchedQP(n=n, k=k, m=m, J=T(J, dt), r=T(r, dt), lam=lam, **common) if level == "J"
            else Q.BatchedQP(n=n, k=k, m=m, G=T(np.tril(G).transpose(0, 2, 1), dt), c=T(c, dt), **common))
    s = Q.QPInteriorPointSolver(prob)
    assert s.step_kernel() == "generic" and s.solve_kernel() == "generic"
    s.SetVariables(T(vars_, dt))
    delta, alpha, status = s.NewtonStep(T(mu, dt), 0.995)
    assert torch.all(status == 0)
    Gl = np.tril(G) if level == "QP" else np.stack([np.tril(J[p].T @ J[p] + lam * np.eye(n)) for p in range(B)])
    cl = c if level == "QP" else np.einsum("bqi,bq->bi", J, r)
    tol = 1e-10 if f64 else TOL32
    qps = []
    for p in range(B):
        qp = orc.QP(G=Gl[p], c=cl[p], A_eq=A[p].T if k else None, b_eq=b[p] if k else None, cons_var=cv[p], cons_a=ca[p], cons_b=cb[p])
        qps.append(qp)
        o = orc.Solver(qp)
        st, d_ref, a_ref = o.newton_step(vars_[p], mu[p] if m else 0.0, 0.995, True)
        assert st == 0
        got = delta[p].double().cpu().numpy()
        assert np.abs(got - d_ref).max() / np.abs(d_ref).max() < tol, (p, np.abs(got - d_ref).max() / np.abs(d_ref).max())
        np.testing.assert_allclose(alpha[p].double().cpu().numpy(), a_ref, atol=1e-9 if f64 else 1e-3)
        o.variables[:] = vars_[p]
        o.evaluate_kkt(True)
        if p == 0:
            r_dev, kkt_dev = s.EvaluateKKTConditions(T(mu, dt))
            scale = max(1.0, np.abs(o.r).max())
            np.testing.assert_allclose(r_dev[0].double().cpu().numpy(), o.r, rtol=0, atol=(1e-11 if f64 else 2e-4) * scale)
    # Iterate (predictor-corrector) and Solve
    s.SetVariables(T(vars_, dt))
    ip, st = s.Iterate(T(mu, dt), Q.PREDICTOR_CORRECTOR if m else Q.COMPLEMENTARITY)
    assert torch.all(st == 0)
    after = s.variables().double().cpu().numpy()
    # (fp32: the residual of a 300-variable system has a rounding floor of ~ eps32 |K| |x| sqrt(n) ~ 3e-3)
    kw = dict(initial_mu=1.0, sigma=0.1, termination_kkt_tol=1e-8 if f64 else 1e-2, termination_complementarity_tol=1e-6 if f64 else 1e-2,
              max_iterations=15 if f64 else 25, initial_guess_method=Q.SOLVE_EQUALITY_CONSTRAINED if k else Q.NAIVE)
    out = s.Solve(Q.Params(**kw))
    assert torch.all(out.status == 0)
    v = s.variables().double().cpu().numpy()
    for p in range(B):
        o = orc.Solver(qps[p])
        o.variables[:] = vars_[p]
        sti, ipr = o.iterate(mu[p], orc.PREDICTOR_CORRECTOR if m else orc.COMPLEMENTARITY)
        assert sti == 0
        sc = max(1.0, np.abs(o.variables).max())
        np.testing.assert_allclose(after[p], o.variables, rtol=0, atol=(1e-8 if f64 else 5e-3) * sc)
        o2 = orc.Solver(qps[p])
        term, its = o2.solve(**kw)
        if f64:
            assert int(out.termination_state[p]) == term and int(out.num_iterations[p]) == len(its), (p, int(out.num_iterations[p]), len(its))
            np.testing.assert_allclose(v[p][:n], o2.variables[:n], rtol=0, atol=1e-7 * max(1.0, np.abs(o2.variables[:n]).max()))
        else:
            assert int(out.termination_state[p]) == Q.SATISFIED_KKT_TOL and abs(int(out.num_iterations[p]) - len(its)) <= 2
            assert np.abs(v[p][:n] - o2.variables[:n]).max() <= 2e-2 * max(1.0, np.abs(o2.variables[:n]).max())


@pytest.mark.parametrize("n,k,m,dt", [(58, 21, 21, torch.float64), (60, 21, 21, torch.float64), (100, 8, 30, torch.float64), (100, 24, 30, torch.float64),
                                      (128, 16, 40, torch.float32), (150, 20, 40, torch.float32)])
def test_generic_solve_first_iteration_is_iterate(n, k, m, dt):
    """One iteration of Solve from the caller's state (USER_PROVIDED, mu as given) is Iterate on that state: same kernel family, same
    arithmetic, so the states afterwards are IDENTICAL and the iteration record carries Iterate's step lengths.  Covers every
    factorisation variant of the generic kernel (P = n + k up to 80, up to 144, beyond); the P > 80 variants once returned alpha_dual = 1
    from Solve for every problem (a stage hipcc had kept as a real function call), which only this comparison and the oracle caught."""
    rng = np.random.default_rng(n + k)
    B, m_r = 5, n + 30
    J = rng.uniform(-1, 1, (B, m_r, n)); r = rng.uniform(-1, 1, (B, m_r))
    A = rng.uniform(-1, 1, (B, n, k)); b = rng.uniform(-1, 1, (B, k))
    cv = rng.integers(0, n, (B, m)).astype(np.int32); ca = rng.choice([-1.0, 1.0, 2.0], (B, m)); cb = rng.uniform(0.5, 2.0, (B, m))
    x = rng.uniform(-0.1, 0.1, (B, n)); sl = rng.uniform(0.2, 1.5, (B, m)); z = rng.uniform(0.1, 2, (B, m)); y = rng.uniform(-1, 1, (B, k))
    vars_ = np.concatenate([x, sl, y, z], axis=1)
    lam = 1e-2
    prob = Q.BatchedQP(n=n, k=k, m=m, J=T(J, dt), r=T(r, dt), lam=lam, A_eq=T(A, dt), b_eq=T(b, dt), cons_var=T(cv, torch.int32),
                       cons_a=T(ca, dt), cons_b=T(cb, dt))
    for strategy in (Q.COMPLEMENTARITY, Q.PREDICTOR_CORRECTOR):
        s = Q.QPInteriorPointSolver(prob, force_generic=True)
        s.SetVariables(T(vars_, dt))
        out = s.Solve(Q.Params(initial_mu=0.05, sigma=0.1, max_iterations=1, barrier_strategy=strategy, initial_guess_method=Q.USER_PROVIDED,
                               initialize_mu_with_complementarity=False))
        assert torch.all(out.status == 0)
        after_solve = s.variables().clone()
        rec = out.iterations[:, 0].cpu().numpy()
        s.SetVariables(T(vars_, dt))
        ip, st = s.Iterate(T(np.full(B, 0.05), dt), strategy)
        assert torch.all(st == 0)
        assert torch.equal(after_solve, s.variables()), (n, k, strategy)
        np.testing.assert_array_equal(rec[:, 8:14], ip.cpu().numpy())
        assert np.any(rec[:, 10] < 1.0)                                        # the case has a cut dual step to report
        if dt == torch.float64:                                                # and the oracle's Iterate agrees
            G = np.einsum("bqi,bqj->bij", J, J) + lam * np.eye(n); c = np.einsum("bqi,bq->bi", J, r)
            for p in range(2):
                o = orc.Solver(orc.QP(G=np.tril(G[p]), c=c[p], A_eq=A[p].T, b_eq=b[p], cons_var=cv[p], cons_a=ca[p], cons_b=cb[p]))
                o.variables[:] = vars_[p]
                ost, oip = o.iterate(0.05, strategy)
                assert ost == 0
                np.testing.assert_allclose([rec[p, 9], rec[p, 10]], [oip.alpha_primal, oip.alpha_dual], rtol=1e-7)


@pytest.mark.parametrize("n,k,m,m_r,level", [(64, 8, 128, 128, "J"), (32, 4, 65, 64, "J"), (64, 0, 100, 0, "QP"), (20, 3, 128, 0, "QP"),
                                             (96, 8, 128, 192, "J"), (128, 14, 90, 256, "J")])
def test_fused_step_with_up_to_128_constraints(n, k, m, m_r, level):
    """m > 64 (e.g. a two-sided box on every one of 64 variables): the fused step kernel carries two constraint slots per lane.
    Against the oracle; duplicates on one variable, both signs."""
    rng = np.random.default_rng(n + 3 * m)
    B = 11
    mr = m_r if m_r else 2 * n
    J = rng.uniform(-1, 1, (B, mr, n)); r = rng.uniform(-1, 1, (B, mr))
    A = rng.uniform(-1, 1, (B, n, k)); b = rng.uniform(-1, 1, (B, k))
    cv = (np.arange(m)[None, :] // 2 % n + np.zeros((B, 1), int)).astype(np.int32)           # a box per variable, wrapping around
    ca = np.where(np.arange(m) % 2 == 0, 1.0, -1.0)[None, :] * np.ones((B, 1)); cb = rng.uniform(0.5, 2.0, (B, m))
    x = rng.uniform(-0.1, 0.1, (B, n))
    sl = rng.uniform(0.2, 1.5, (B, m)); z = rng.uniform(0.1, 2, (B, m)); y = rng.uniform(-1, 1, (B, k))
    vars_ = np.concatenate([x, sl, y, z], axis=1)
    mu = np.full(B, 0.05)
    lam = 1e-3
    G = np.einsum("bqi,bqj->bij", J, J) + lam * np.eye(n)
    c = np.einsum("bqi,bq->bi", J, r)
    common = dict(A_eq=T(A) if k else None, b_eq=T(b) if k else None, cons_var=T(cv, torch.int32), cons_a=T(ca), cons_b=T(cb))
    prob = (Q.BatchedQP(n=n, k=k, m=m, J=T(J), r=T(r), lam=lam, **common) if level == "J"
            else Q.BatchedQP(n=n, k=k, m=m, G=T(np.tril(G).transpose(0, 2, 1)), c=T(c), **common))
    s = Q.QPInteriorPointSolver(prob)
    assert s.step_kernel().startswith("fused"), s.step_kernel()
    s.SetVariables(T(vars_))
    delta, alpha, status = s.NewtonStep(T(mu), 0.995)
    ref, ref_alpha, ref_status, _ = orc.batched_newton_step(n, k, m, G=np.tril(G).transpose(0, 2, 1).copy(), c=c, A_eq=A if k else None,
                                                            b_eq=b if k else None, cons_var=cv, cons_a=ca, cons_b=cb, vars_=vars_, mu=mu)
    assert torch.all(status == 0) and np.all(ref_status == 0)
    assert rel_inf_rows(delta.cpu().numpy(), ref).max() < 1e-9
    np.testing.assert_allclose(alpha.cpu().numpy(), ref_alpha, atol=1e-9)
    # status words from the second slot: s <= 0 and a bad index beyond lane 63
    vars2 = vars_.copy(); vars2[1, n + m - 1] = 0.0
    cv2 = cv.copy(); cv2[2, m - 2] = n
    prob.cons_var = T(cv2, torch.int32)
    s2 = Q.QPInteriorPointSolver(prob); s2.SetVariables(T(vars2))
    _, _, st2 = s2.NewtonStep(T(mu), 0.995)
    st2 = st2.cpu().numpy()
    assert st2[1] == L.MO_STATUS_NONPOSITIVE_SLACK and st2[2] == L.MO_STATUS_BAD_INDEX and np.all(np.delete(st2, [1, 2]) == 0)


@pytest.mark.parametrize("n,k,m,m_r,level", [(64, 8, 128, 128, "J"), (32, 4, 70, 64, "J"), (50, 0, 100, 0, "QP"), (10, 2, 128, 0, "QP")])
@pytest.mark.parametrize("strategy", [Q.COMPLEMENTARITY, Q.PREDICTOR_CORRECTOR])
def test_fused_solve_with_up_to_128_constraints(n, k, m, m_r, level, strategy):
    """Solve and Iterate with m > 64 on the 32 / 64 tile grids (two constraint slots per lane), fused vs generic kernel and -- for
    the iteration records of a few problems -- vs the oracle."""
    rng = np.random.default_rng(n + 5 * m + strategy)
    B = 9
    mr = m_r if m_r else 2 * n
    J = rng.uniform(-1, 1, (B, mr, n)); r = rng.uniform(-1, 1, (B, mr))
    A = rng.uniform(-1, 1, (B, n, k)); b = 0.1 * rng.uniform(-1, 1, (B, k))
    cv = (np.arange(m)[None, :] // 2 % n + np.zeros((B, 1), int)).astype(np.int32)
    ca = np.where(np.arange(m) % 2 == 0, 1.0, -1.0)[None, :] * np.ones((B, 1)); cb = rng.uniform(1.0, 2.0, (B, m))
    x = rng.uniform(-0.1, 0.1, (B, n))
    sl = rng.uniform(0.2, 1.5, (B, m)); z = rng.uniform(0.1, 2, (B, m)); y = rng.uniform(-1, 1, (B, k))
    vars_ = np.concatenate([x, sl, y, z], axis=1)
    mu = np.full(B, 0.05)
    lam = 1e-3
    G = np.einsum("bqi,bqj->bij", J, J) + lam * np.eye(n)
    c = np.einsum("bqi,bq->bi", J, r)
    common = dict(A_eq=T(A) if k else None, b_eq=T(b) if k else None, cons_var=T(cv, torch.int32), cons_a=T(ca), cons_b=T(cb))
    prob = (Q.BatchedQP(n=n, k=k, m=m, J=T(J), r=T(r), lam=lam, **common) if level == "J"
            else Q.BatchedQP(n=n, k=k, m=m, G=T(np.tril(G).transpose(0, 2, 1)), c=T(c), **common))
    kw = dict(initial_mu=1.0, sigma=0.1, termination_kkt_tol=1e-8, max_iterations=12, barrier_strategy=strategy,
              initial_guess_method=Q.SOLVE_EQUALITY_CONSTRAINED if k else Q.NAIVE)
    res = {}
    for force in (False, True):
        s = Q.QPInteriorPointSolver(prob, force_generic=force)
        s.SetVariables(T(vars_))
        ip, st = s.Iterate(T(mu), strategy)
        assert torch.all(st == 0)
        after = s.variables().cpu().numpy().copy()
        out = s.Solve(Q.Params(**kw))
        assert torch.all(out.status == 0)
        res[force] = (ip.cpu().numpy(), after, s.variables().cpu().numpy().copy(), out.num_iterations.cpu().numpy(),
                      out.termination_state.cpu().numpy(), out.iterations.cpu().numpy())
    f, g_ = res[False], res[True]
    np.testing.assert_allclose(f[0], g_[0], rtol=1e-7, atol=1e-10, equal_nan=True)
    np.testing.assert_allclose(f[1], g_[1], rtol=1e-8, atol=1e-10)
    assert np.array_equal(f[3], g_[3]) and np.array_equal(f[4], g_[4])
    conv = f[4] == Q.SATISFIED_KKT_TOL
    assert conv.mean() > 0.6
    np.testing.assert_allclose(f[2][conv][:, :n], g_[2][conv][:, :n], rtol=1e-6, atol=1e-8)
    for p in range(3):  # iteration records against the oracle
        o = orc.Solver(orc.QP(G=np.tril(G[p]), c=c[p], A_eq=A[p].T if k else None, b_eq=b[p] if k else None, cons_var=cv[p], cons_a=ca[p], cons_b=cb[p]))
        term, its = o.solve(**kw)
        assert f[4][p] == term and f[3][p] == len(its)
        scale = max(max(i.kkt_initial.r_dual, i.kkt_initial.r_comp, i.kkt_initial.r_primal_ineq, i.kkt_initial.r_primal_eq) for i in its)
        for i, itr in enumerate(its):
            exp = [itr.kkt_initial.r_dual, itr.kkt_initial.r_comp, itr.kkt_initial.r_primal_eq, itr.kkt_initial.r_primal_ineq,
                   itr.kkt_final.r_dual, itr.kkt_final.r_comp, itr.kkt_final.r_primal_eq, itr.kkt_final.r_primal_ineq,
                   itr.ip.mu, itr.ip.alpha_primal, itr.ip.alpha_dual, itr.ip.alpha_probe_primal, itr.ip.alpha_probe_dual, itr.ip.mu_affine]
            np.testing.assert_allclose(f[5][p][i], exp, rtol=1e-6, atol=1e-9 + 1e-12 * scale, equal_nan=True)


@pytest.mark.parametrize("n,k,m,m_r,level", [(128, 14, 256, 256, "J"), (64, 8, 200, 128, "J"), (30, 3, 256, 0, "QP"), (96, 8, 160, 192, "J"),
                                             (128, 10, 100, 256, "J"), (96, 6, 128, 0, "QP"), (100, 4, 129, 0, "QP"), (32, 4, 130, 64, "J")])
def test_fused_four_constraint_slots(n, k, m, m_r, level):
    """128 < m <= 256 (a two-sided box on every one of 128 variables is m = 256) and Solve / Iterate with m > 64 on the 96 / 128 tile grids:
    the instantiations of kkt_fused_mc4.hip.  Step, Iterate (predictor-corrector) and the whole Solve against the oracle, problem by problem
    -- the LDS-resident generic kernel cannot hold most of these shapes at all."""
    rng = np.random.default_rng(n + 7 * m)
    B = 6
    mr = m_r if m_r else 2 * n
    J = rng.uniform(-1, 1, (B, mr, n)); r = rng.uniform(-1, 1, (B, mr))
    A = rng.uniform(-1, 1, (B, n, k)); b = 0.1 * rng.uniform(-1, 1, (B, k))
    cv = (np.arange(m)[None, :] // 2 % n + np.zeros((B, 1), int)).astype(np.int32)           # a box per variable, wrapping around
    ca = np.where(np.arange(m) % 2 == 0, 1.0, -1.0)[None, :] * np.ones((B, 1)); cb = rng.uniform(1.0, 2.0, (B, m))
    x = rng.uniform(-0.1, 0.1, (B, n))
    sl = rng.uniform(0.2, 1.5, (B, m)); z = rng.uniform(0.1, 2, (B, m)); y = rng.uniform(-1, 1, (B, k))
    vars_ = np.concatenate([x, sl, y, z], axis=1)
    mu = np.full(B, 0.05)
    lam = 1e-3
    G = np.einsum("bqi,bqj->bij", J, J) + lam * np.eye(n)
    c = np.einsum("bqi,bq->bi", J, r)
    common = dict(A_eq=T(A), b_eq=T(b), cons_var=T(cv, torch.int32), cons_a=T(ca), cons_b=T(cb))
    prob = (Q.BatchedQP(n=n, k=k, m=m, J=T(J), r=T(r), lam=lam, **common) if level == "J"
            else Q.BatchedQP(n=n, k=k, m=m, G=T(np.tril(G).transpose(0, 2, 1)), c=T(c), **common))
    s = Q.QPInteriorPointSolver(prob)
    assert s.step_kernel().startswith("fused") and s.solve_kernel().startswith("fused"), (s.step_kernel(), s.solve_kernel())
    s.SetVariables(T(vars_))
    delta, alpha, status = s.NewtonStep(T(mu), 0.995)
    ref, ref_alpha, ref_status, _ = orc.batched_newton_step(n, k, m, G=np.tril(G).transpose(0, 2, 1).copy(), c=c, A_eq=A, b_eq=b, cons_var=cv,
                                                            cons_a=ca, cons_b=cb, vars_=vars_, mu=mu)
    assert torch.all(status == 0) and np.all(ref_status == 0)
    assert rel_inf_rows(delta.cpu().numpy(), ref).max() < 1e-9
    np.testing.assert_allclose(alpha.cpu().numpy(), ref_alpha, atol=1e-9)
    kw = dict(initial_mu=1.0, sigma=0.1, termination_kkt_tol=1e-8, max_iterations=12, initial_guess_method=Q.SOLVE_EQUALITY_CONSTRAINED)
    for strategy in (Q.COMPLEMENTARITY, Q.PREDICTOR_CORRECTOR):
        s.SetVariables(T(vars_))
        ip, st = s.Iterate(T(mu), strategy)
        assert torch.all(st == 0)
        after = s.variables().cpu().numpy().copy()
        out = s.Solve(Q.Params(barrier_strategy=strategy, **kw))
        assert torch.all(out.status == 0)
        v = s.variables().cpu().numpy(); nit = out.num_iterations.cpu().numpy(); tm = out.termination_state.cpu().numpy()
        make_qp = lambda p: orc.QP(G=np.tril(G[p]), c=c[p], A_eq=A[p].T, b_eq=b[p], cons_var=cv[p], cons_a=ca[p], cons_b=cb[p])
        # the oracle as referee: a Solve that differs from the oracle's needs a logged knife edge (at most one of the six problems)
        same = solves_agree_or_knife_edge((n, k, m, m_r, level, strategy), make_qp, dict(barrier_strategy=strategy, **kw), {"fused": (tm, nit)})
        for p in range(B):
            qp = make_qp(p)
            o = orc.Solver(qp)
            o.variables[:] = vars_[p]
            ost, oip = o.iterate(0.05, strategy)
            assert ost == 0
            np.testing.assert_allclose(ip.cpu().numpy()[p], [oip.mu, oip.alpha_primal, oip.alpha_dual, oip.alpha_probe_primal, oip.alpha_probe_dual, oip.mu_affine],
                                       rtol=1e-6, atol=1e-9, equal_nan=True)
            np.testing.assert_allclose(after[p], o.variables, rtol=1e-7, atol=1e-8 * max(1.0, np.abs(o.variables).max()))
            o2 = orc.Solver(qp)
            term, its = o2.solve(barrier_strategy=strategy, **kw)
            if same[p] and term == Q.SATISFIED_KKT_TOL:
                np.testing.assert_allclose(v[p][:n], o2.variables[:n], rtol=1e-6, atol=1e-8)


@pytest.mark.parametrize("shape", [None, (200, 10, 16, 24)], ids=["cfg2-fused", "n200-generic-large"])
def test_newton_step_is_graph_capturable(shape):
    """mo_newton_step enqueues only a 8-byte memset and one kernel on the caller's stream: it can be captured into a HIP graph
    (torch.cuda.graph) and replayed on new data in the same buffers.  Also beyond the LDS-resident range (n + k > 192): the generic kernel's
    H workspaces belong to the plan and are allocated by mo_plan_create, the launch allocates and synchronises nothing."""
    d = synth.CONFIGS["cfg2"] if shape is None else dict(n=shape[0], k=shape[1], m=shape[2], m_r=shape[3])
    B = 64 if shape is None else 6
    hb = synth.make_batch(d["n"], d["k"], d["m"], d["m_r"], B, stream=31)
    prob = batch_to_device(hb)
    s = Q.QPInteriorPointSolver(prob)
    assert s.step_kernel() == ("generic" if shape else "fused_mfma_f64_n32")
    s.SetVariables(T(hb.vars))
    mu = T(hb.mu)
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        s.NewtonStep(mu, 0.995)                              # warm-up outside the capture (allocations of the Python wrapper)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        delta, alpha, status = s.NewtonStep(mu, 0.995)
    # new state in the captured buffers, then replay
    hb2 = synth.make_batch(d["n"], d["k"], d["m"], d["m_r"], B, stream=32)
    s.variables().copy_(T(hb2.vars)); mu.copy_(T(hb2.mu))
    prob.J.copy_(T(hb2.J)); prob.r.copy_(T(hb2.r)); prob.A_eq.copy_(T(hb2.A_eq)); prob.b_eq.copy_(T(hb2.b_eq))
    prob.cons_var.copy_(T(hb2.cons_var, torch.int32)); prob.cons_a.copy_(T(hb2.cons_a)); prob.cons_b.copy_(T(hb2.cons_b))
    graph.replay()
    torch.cuda.synchronize()
    ref, ref_alpha, ref_status, _ = orc.batched_newton_step(hb2.n, hb2.k, hb2.m, J=hb2.J, r=hb2.r, lam=hb2.lam, A_eq=hb2.A_eq, b_eq=hb2.b_eq,
                                                            cons_var=hb2.cons_var, cons_a=hb2.cons_a, cons_b=hb2.cons_b, vars_=hb2.vars, mu=hb2.mu)
    assert torch.all(status == 0)
    assert rel_inf_rows(delta.cpu().numpy(), ref).max() < TOL64


def test_large_generic_workspace_belongs_to_the_plan():
    """Beyond the LDS-resident range the generic kernel keeps H in a global workspace per workgroup.  It is sized and allocated once, in
    mo_plan_create, for every system the plan launches there -- the full one and the k = m = 0 one of mo_linearize / mo_fill_qp -- so
    interleaving mo_newton_step, mo_linearize and mo_fill_qp on ONE plan neither allocates (device memory stays flat over repeated calls;
    round 3 leaked 160-270 MB per linearisation through a stack copy of the plan) nor frees a workspace another call still owns (the step
    is bit-identical before and after)."""
    import ctypes as C
    n, k, m, m_r, B = 200, 10, 16, 24, 6
    hb = synth.make_batch(n, k, m, m_r, B, stream=91)
    prob = batch_to_device(hb)
    lib = L.lib()
    desc = L.PlanDesc(n, k, m, m_r, L.MO_F64, 0, 0, 0, B)
    plan = C.c_void_p()
    L.check(lib.mo_plan_create(C.byref(desc), C.byref(plan)))
    try:
        ps = prob.as_struct()
        assert lib.mo_plan_step_kernel(plan, C.byref(ps)).decode() == "generic"
        V = n + 2 * m + k
        vars_, mu = T(hb.vars), T(hb.mu)
        delta = torch.empty(B, V, dtype=torch.float64, device=dev()); alpha = torch.empty(B, 2, dtype=torch.float64, device=dev())
        status = torch.empty(B, dtype=torch.int32, device=dev())
        G = torch.empty(B, n, n, dtype=torch.float64, device=dev()); c = torch.empty(B, n, dtype=torch.float64, device=dev())
        f = torch.empty(B, dtype=torch.float64, device=dev()); cb = torch.empty(B, m, dtype=torch.float64, device=dev())
        err = torch.empty(B, 2, dtype=torch.float64, device=dev()); st2 = torch.empty(B, dtype=torch.int32, device=dev())

        def step():
            L.check(lib.mo_newton_step(plan, C.byref(ps), B, Q._ptr(vars_), V, Q._ptr(mu), 1, 0.995, 0, Q._ptr(delta), V, Q._ptr(alpha),
                                       Q._ptr(status), Q._stream()))

        def linearise():
            L.check(lib.mo_linearize(plan, C.byref(ps), B, Q._ptr(G), n * n, n, Q._ptr(c), n, Q._ptr(f), Q._stream()))

        def fill():
            L.check(lib.mo_fill_qp(plan, C.byref(ps), B, Q._ptr(vars_), V, Q._ptr(G), n * n, n, Q._ptr(c), n, Q._ptr(cb), m, Q._ptr(err),
                                   Q._ptr(st2), Q._stream()))

        step(); linearise(); fill()
        torch.cuda.synchronize()
        first = delta.clone()
        ref, _, _, _ = orc.batched_newton_step(n, k, m, J=hb.J, r=hb.r, lam=hb.lam, A_eq=hb.A_eq, b_eq=hb.b_eq, cons_var=hb.cons_var,
                                               cons_a=hb.cons_a, cons_b=hb.cons_b, vars_=hb.vars, mu=hb.mu)
        assert torch.all(status == 0) and rel_inf_rows(first.cpu().numpy(), ref).max() < TOL64
        Gd = np.einsum("bri,brj->bij", hb.J, hb.J) + hb.lam * np.eye(n)
        assert np.abs(np.tril(G.cpu().numpy().transpose(0, 2, 1)) - np.tril(Gd)).max() < 1e-11   # column-major out: G[b] is G^T
        free0 = torch.cuda.mem_get_info()[0]
        for _ in range(4):
            linearise(); step(); fill(); linearise()
        torch.cuda.synchronize()
        assert torch.cuda.mem_get_info()[0] >= free0 - (1 << 20), (free0, torch.cuda.mem_get_info()[0])
        assert torch.equal(delta, first)
    finally:
        lib.mo_plan_destroy(plan)


@pytest.mark.parametrize("cfg", ["cfg3", "n128"])
def test_first_fused_solve_allocates_nothing(cfg):
    """The tile park of the fused fp64 Solve kernels beyond the 32 grid (the G tiles a wave cannot keep in LDS between passes: 69 MB of
    per-wave-slot scratch at n = 64, 239 MB at n = 128) belongs to the plan since round 4: mo_plan_create allocates it, the first
    mo_qp_solve of a plan allocates nothing (it used to hipMalloc it in the launch path).  Device memory is flat over that first call."""
    import ctypes as C
    d = synth.CONFIGS["cfg3"] if cfg == "cfg3" else dict(n=128, k=14, m=64, m_r=256)
    n, k, m, m_r, B = d["n"], d["k"], d["m"], d["m_r"], 96
    hb = synth.make_batch(n, k, m, m_r, B, stream=17)
    prob = batch_to_device(hb)
    s = Q.QPInteriorPointSolver(prob)
    assert s.solve_kernel().startswith("fused_solve"), s.solve_kernel()
    s.SetVariables(T(hb.vars))
    kw = dict(initial_mu=1.0, sigma=0.1, termination_kkt_tol=1e-8, max_iterations=12, initial_guess_method=Q.SOLVE_EQUALITY_CONSTRAINED)
    sp = Q.Params(**kw).as_struct()
    V = n + 2 * m + k
    term = torch.zeros(B, dtype=torch.int32, device=dev()); nit = torch.zeros(B, dtype=torch.int32, device=dev())
    its = torch.zeros(B, kw["max_iterations"], L.MO_ITER_RECORD, dtype=torch.float64, device=dev())
    lag = torch.zeros(B, 2, dtype=torch.float64, device=dev()); status = torch.zeros(B, dtype=torch.int32, device=dev())
    svars = s.variables()
    lib, plan, ps = L.lib(), s._plan, s._prob
    # (what the HIP runtime reserves on its own account happens on ANOTHER plan first: the module load of the process's first launch -- 256 MiB --
    # and the scratch arena of a kernel with a private segment, 160 MB for the 128 grid's Solve)
    warm = Q.QPInteriorPointSolver(batch_to_device(hb))
    warm.SetVariables(T(hb.vars)); warm.Solve(Q.Params(**kw))
    del warm
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    L.check(lib.mo_qp_solve(plan, C.byref(ps), B, C.byref(sp), Q._ptr(svars), V, Q._ptr(term), Q._ptr(nit), Q._ptr(its), Q._ptr(lag), Q._ptr(status),
                            Q._stream()))
    torch.cuda.synchronize()
    assert torch.cuda.mem_get_info()[0] >= free0 - (1 << 20), (free0, torch.cuda.mem_get_info()[0])
    assert torch.all(status == 0) and torch.all(term == Q.SATISFIED_KKT_TOL)
    G = np.einsum("bqi,bqj->bij", hb.J, hb.J) + hb.lam * np.eye(n); c = np.einsum("bqi,bq->bi", hb.J, hb.r)
    for p in range(0, B, 31):
        o = orc.Solver(orc.QP(G=np.tril(G[p]), c=c[p], A_eq=hb.A_eq[p].T, b_eq=hb.b_eq[p], cons_var=hb.cons_var[p], cons_a=hb.cons_a[p], cons_b=hb.cons_b[p]))
        t_ref, its_ref = o.solve(**kw)
        assert int(term[p]) == t_ref and int(nit[p]) == len(its_ref), (p, int(nit[p]), len(its_ref))


@pytest.mark.parametrize("batch", [1, 2, 13])
def test_tiny_batches_on_the_fused_kernels(batch):
    """Fewer problems than waves in one workgroup: the ticket loop must hand out exactly `batch` problems (step, Solve, linearise)."""
    d = synth.CONFIGS["cfg3"]
    hb = synth.make_batch(d["n"], d["k"], d["m"], d["m_r"], batch, stream=77)
    prob = batch_to_device(hb)
    s = Q.QPInteriorPointSolver(prob)
    assert s.step_kernel().startswith("fused")
    s.SetVariables(T(hb.vars))
    delta, alpha, status = s.NewtonStep(T(hb.mu), 0.995)
    ref, ref_alpha, ref_status, _ = orc.batched_newton_step(hb.n, hb.k, hb.m, J=hb.J, r=hb.r, lam=hb.lam, A_eq=hb.A_eq, b_eq=hb.b_eq,
                                                            cons_var=hb.cons_var, cons_a=hb.cons_a, cons_b=hb.cons_b, vars_=hb.vars, mu=hb.mu)
    assert torch.all(status == 0)
    assert rel_inf_rows(delta.cpu().numpy(), ref).max() < TOL64
    out = s.Solve(Q.Params(initial_mu=1.0, sigma=0.1, termination_kkt_tol=1e-8, max_iterations=12))
    assert torch.all(out.status == 0) and torch.all(out.termination_state == Q.SATISFIED_KKT_TOL)
    G, c, half = Q.linearize(prob)
    np.testing.assert_allclose(c.cpu().numpy(), np.einsum("bqi,bq->bi", hb.J, hb.r), rtol=1e-12, atol=1e-12)


@pytest.mark.parametrize("n,k,m,m_r", [(63, 8, 32, 128), (33, 4, 16, 64), (31, 3, 7, 33), (7, 2, 5, 9), (3, 0, 2, 3), (51, 14, 64, 101), (1 + 2 * 16, 0, 0, 4)])
def test_fused_odd_n_with_stacked_jacobian(n, k, m, m_r):
    """Odd n with (J, r, lambda) input: rows of J are only 8-byte aligned, so the 4-row groups go through the ring as flat runs
    (dword DMA) and the operands are picked out with masked 8-byte LDS reads.  Step, Iterate and Solve against the oracle / the
    generic kernel; m_r odd makes the per-problem base of J and r 8-byte aligned only."""
    rng = np.random.default_rng(n * 11 + m_r)
    B = 10
    J = rng.uniform(-1, 1, (B, m_r, n)); r = rng.uniform(-1, 1, (B, m_r))
    A = rng.uniform(-1, 1, (B, n, k)); b = 0.2 * rng.uniform(-1, 1, (B, k))
    cv = rng.integers(0, n, (B, m)).astype(np.int32)
    ca = rng.choice([-1.0, 1.0, 2.0], (B, m)); cb = rng.uniform(0.5, 2.0, (B, m))
    x = rng.uniform(-0.1, 0.1, (B, n))
    sl = rng.uniform(0.2, 1.5, (B, m)); z = rng.uniform(0.1, 2, (B, m)); y = rng.uniform(-1, 1, (B, k))
    vars_ = np.concatenate([x, sl, y, z], axis=1)
    mu = np.full(B, 0.05)
    lam = 0.3 if m_r < n else 1e-3
    prob = Q.BatchedQP(n=n, k=k, m=m, J=T(J), r=T(r), lam=lam, A_eq=T(A) if k else None, b_eq=T(b) if k else None,
                       cons_var=T(cv, torch.int32) if m else None, cons_a=T(ca) if m else None, cons_b=T(cb) if m else None)
    s = Q.QPInteriorPointSolver(prob, no_tiny=True)               # (the smallest shapes would otherwise run on the one-tile kernel)
    assert s.step_kernel().startswith("fused_mfma"), s.step_kernel()
    s.SetVariables(T(vars_))
    delta, alpha, status = s.NewtonStep(T(mu), 0.995)
    ref, ref_alpha, ref_status, _ = orc.batched_newton_step(n, k, m, J=J, r=r, lam=lam, A_eq=A if k else None, b_eq=b if k else None,
                                                            cons_var=cv if m else None, cons_a=ca if m else None, cons_b=cb if m else None,
                                                            vars_=vars_, mu=mu)
    assert torch.all(status == 0) and np.all(ref_status == 0)
    assert rel_inf_rows(delta.cpu().numpy(), ref).max() < 1e-9
    np.testing.assert_allclose(alpha.cpu().numpy(), ref_alpha, atol=1e-9)
    kw = dict(initial_mu=1.0, sigma=0.1, termination_kkt_tol=1e-8, max_iterations=12, barrier_strategy=Q.PREDICTOR_CORRECTOR,
              initial_guess_method=Q.SOLVE_EQUALITY_CONSTRAINED if k else Q.NAIVE)
    res = {}
    for force in (False, True):
        sv = Q.QPInteriorPointSolver(prob, force_generic=force)
        sv.SetVariables(T(vars_))
        ip, st = sv.Iterate(T(mu), Q.COMPLEMENTARITY)
        assert torch.all(st == 0)
        after = sv.variables().cpu().numpy().copy()
        out = sv.Solve(Q.Params(**kw))
        assert torch.all(out.status == 0)
        res[force] = (after, sv.variables().cpu().numpy().copy(), out.num_iterations.cpu().numpy(), out.termination_state.cpu().numpy())
    np.testing.assert_allclose(res[False][0], res[True][0], rtol=1e-8, atol=1e-10)
    Gd = np.einsum("bqi,bqj->bij", J, J) + lam * np.eye(n); cd = np.einsum("bqi,bq->bi", J, r)
    make_qp = lambda p: orc.QP(G=np.tril(Gd[p]), c=cd[p], A_eq=A[p].T if k else None, b_eq=b[p] if k else None, cons_var=cv[p], cons_a=ca[p], cons_b=cb[p])
    same = solves_agree_or_knife_edge((n, k, m, m_r), make_qp, kw, {"fused": (res[False][3], res[False][2]), "generic": (res[True][3], res[True][2])})
    same &= res[False][3] == Q.SATISFIED_KKT_TOL
    np.testing.assert_allclose(res[False][1][same][:, :n], res[True][1][same][:, :n], rtol=1e-6, atol=1e-8)


# ------------------------------------------------------------------ J in the other layouts of the C ABI: the per-lane gather stream
def _layouts_of(J):
    """The same stacked Jacobian [B, m_r, n] as (label, tensor, BatchedQP keyword arguments) in every layout mo_problem accepts."""
    B, m_r, n = J.shape
    out = [("packed", T(J), {})]
    wide = np.full((B, m_r, n + 5), 7.7); wide[:, :, :n] = J                       # row-major, leading dimension n + 5
    out.append(("row ld=n+5", T(wide), {}))
    colm = np.ascontiguousarray(J.transpose(0, 2, 1))                               # column-major, ld = m_r
    out.append(("col", T(colm), dict(J_layout="col")))
    colw = np.full((B, n, m_r + 3), -3.3); colw[:, :, :m_r] = J.transpose(0, 2, 1)   # column-major, ld = m_r + 3
    out.append(("col ld=m_r+3", T(colw), dict(J_layout="col", J_rows=m_r)))
    flat = torch.zeros(B * m_r * n + 1, dtype=torch.float64, device="cuda:0")       # packed but starting 8 bytes off a 16-byte boundary
    flat[1:] = T(J).reshape(-1)
    out.append(("unaligned", flat[1:].view(B, m_r, n), {}))
    return out


@pytest.mark.parametrize("n,k,m,m_r", [(64, 8, 32, 128), (32, 4, 16, 64), (40, 3, 10, 50), (63, 8, 32, 131), (96, 8, 40, 192), (101, 5, 20, 203),
                                       (128, 10, 40, 256), (6, 2, 4, 9), (48, 20, 10, 70), (33, 31, 16, 41), (100, 16, 30, 150),
                                       # round 4: three / four y tiles in every layout
                                       (64, 40, 32, 128), (57, 33, 20, 70), (96, 50, 24, 192), (128, 36, 16, 260), (30, 20, 8, 64)])
def test_fused_kernels_take_every_layout_of_J(n, k, m, m_r):
    """Column-major J, a leading dimension beyond n, rows that are only 8-byte aligned, odd n beyond 64: all of them run on the fused
    kernels (the gather stream, kkt_fused_gather.hip) and give the packed layout's results -- step against the oracle, Iterate, Solve and
    the KKT residual against the packed run."""
    rng = np.random.default_rng(n * 7 + m_r)
    B = 7
    J = rng.uniform(-1, 1, (B, m_r, n)); r = rng.uniform(-1, 1, (B, m_r))
    A = rng.uniform(-1, 1, (B, n, k)); b = rng.uniform(-1, 1, (B, k))
    cv = rng.integers(0, n, (B, m)).astype(np.int32); ca = rng.choice([-1.0, 1.0, 2.0], (B, m)); cb = rng.uniform(0.5, 2.0, (B, m))
    x = rng.uniform(-0.1, 0.1, (B, n)); sl = rng.uniform(0.2, 1.5, (B, m)); z = rng.uniform(0.1, 2, (B, m)); y = rng.uniform(-1, 1, (B, k))
    vars_ = np.concatenate([x, sl, y, z], axis=1); mu = np.full(B, 0.05)
    ref, ref_alpha, ref_status, _ = orc.batched_newton_step(n, k, m, J=J, r=r, lam=1e-3, A_eq=A, b_eq=b, cons_var=cv, cons_a=ca, cons_b=cb,
                                                            vars_=vars_, mu=mu)
    assert np.all(ref_status == 0)
    base = None
    for label, Jt, kw in _layouts_of(J):
        prob = Q.BatchedQP(n=n, k=k, m=m, J=Jt, r=T(r), lam=1e-3, A_eq=T(A), b_eq=T(b), cons_var=T(cv, torch.int32), cons_a=T(ca), cons_b=T(cb), **kw)
        s = Q.QPInteriorPointSolver(prob)
        assert s.step_kernel().startswith("fused"), (label, s.step_kernel())
        s.SetVariables(T(vars_))
        delta, alpha, status = s.NewtonStep(T(mu), 0.995)
        assert torch.all(status == 0), label
        assert rel_inf_rows(delta.cpu().numpy(), ref).max() < 1e-10, label
        np.testing.assert_allclose(alpha.cpu().numpy(), ref_alpha, atol=1e-9, err_msg=label)
        res = [t.cpu().numpy().copy() for t in s.EvaluateKKTConditions(T(mu))]
        ip, st2 = s.Iterate(T(mu), Q.PREDICTOR_CORRECTOR)
        assert torch.all(st2 == 0), label
        after = s.variables().cpu().numpy().copy()
        out = s.Solve(Q.Params(initial_mu=1.0, sigma=0.1, termination_kkt_tol=1e-8, max_iterations=12, initial_guess_method=Q.SOLVE_EQUALITY_CONSTRAINED))
        assert torch.all(out.status == 0), label
        got = (delta.cpu().numpy().copy(), res, ip.cpu().numpy().copy(), after, s.variables().cpu().numpy().copy(), out.num_iterations.cpu().numpy().copy(),
               out.termination_state.cpu().numpy().copy())
        if base is None:
            base = got
            continue
        # the gather stream feeds the matrix cores the same operands in the same order: identical arithmetic
        for a_, b_ in zip((got[0], got[1][0], got[1][1], got[2], got[3], got[4]), (base[0], base[1][0], base[1][1], base[2], base[3], base[4])):
            np.testing.assert_allclose(a_, b_, rtol=1e-12, atol=1e-13, equal_nan=True, err_msg=label)
        assert np.array_equal(got[5], base[5]) and np.array_equal(got[6], base[6]), label


@pytest.mark.parametrize("n,k,m,m_r", [(200, 20, 40, 210), (130, 70, 9, 37), (257, 3, 5, 301)], ids=["n200", "n130_k70_short_J", "n257_odd"])
def test_large_generic_path_takes_every_layout_of_J(n, k, m, m_r):
    """Beyond the LDS-resident range (H in the workgroup's global workspace) J^T J runs on the matrix cores over 128-wide super-blocks whose
    columns are staged through LDS by batched loads -- one code path for row-major J and one for column-major J / a leading dimension.  Every
    layout must give the oracle's step, and the same step as the packed layout (the staged values are the same: identical arithmetic).
    Shapes: a second super-block with a ragged width (n = 200, 257), fewer rows of J than one staging chunk, a row count that is not a
    multiple of the four rows one MFMA consumes, a last column panel narrower than the others."""
    rng = np.random.default_rng(n + 3 * m_r)
    B = 3
    J = rng.uniform(-1, 1, (B, m_r, n)); r = rng.uniform(-1, 1, (B, m_r))
    A = rng.uniform(-1, 1, (B, n, k)); b = rng.uniform(-1, 1, (B, k))
    cv = rng.integers(0, n, (B, m)).astype(np.int32); ca = rng.choice([-1.0, 1.0, 2.0], (B, m)); cb = rng.uniform(0.5, 2.0, (B, m))
    x = rng.uniform(-0.1, 0.1, (B, n)); sl = rng.uniform(0.2, 1.5, (B, m)); z = rng.uniform(0.1, 2, (B, m)); y = rng.uniform(-1, 1, (B, k))
    vars_ = np.concatenate([x, sl, y, z], axis=1); mu = np.full(B, 0.05)
    lam = 1e-3 if m_r >= n else 0.5      # (fewer rows than variables: J^T J alone is singular)
    ref, ref_alpha, ref_status, _ = orc.batched_newton_step(n, k, m, J=J, r=r, lam=lam, A_eq=A, b_eq=b, cons_var=cv, cons_a=ca, cons_b=cb,
                                                            vars_=vars_, mu=mu)
    assert np.all(ref_status == 0)
    base = None
    for label, Jt, kw in _layouts_of(J):
        prob = Q.BatchedQP(n=n, k=k, m=m, J=Jt, r=T(r), lam=lam, A_eq=T(A), b_eq=T(b), cons_var=T(cv, torch.int32), cons_a=T(ca), cons_b=T(cb), **kw)
        s = Q.QPInteriorPointSolver(prob)
        assert s.step_kernel() == "generic", (label, s.step_kernel())
        s.SetVariables(T(vars_))
        delta, alpha, status = s.NewtonStep(T(mu), 0.995)
        assert torch.all(status == 0), label
        assert rel_inf_rows(delta.cpu().numpy(), ref).max() < 1e-10, label
        np.testing.assert_allclose(alpha.cpu().numpy(), ref_alpha, atol=1e-9, err_msg=label)
        G, c, half = Q.linearize(prob, force_generic=True)
        got = (delta.cpu().numpy().copy(), np.tril(G.cpu().numpy().transpose(0, 2, 1)), c.cpu().numpy().copy())
        if base is None:
            Gref = np.einsum("bqi,bqj->bij", J, J) + lam * np.eye(n)
            np.testing.assert_allclose(got[1], np.tril(Gref), rtol=0, atol=1e-11 * m_r)
            np.testing.assert_allclose(got[2], np.einsum("bqi,bq->bi", J, r), rtol=0, atol=1e-11 * m_r)
            base = got
            continue
        for a_, b_ in zip(got, base):
            assert np.array_equal(a_, b_), label


# ------------------------------------------------------------------ Params::decrease_mu_only_on_small_error (qp.hpp:154-157, qp.cc:140-146)
@pytest.mark.parametrize("shape", [(8, 2, 4, 16), (32, 4, 16, 64), (64, 8, 32, 128), (64, 24, 32, 128), (100, 8, 30, 128)],
                         ids=["one_tile", "grid32", "grid64", "two_y_tiles", "grid128"])
@pytest.mark.parametrize("strategy", [Q.COMPLEMENTARITY, Q.FIXED_DECREASE, Q.PREDICTOR_CORRECTOR])
def test_decrease_mu_only_on_small_error(shape, strategy):
    """With the flag set mu is only decreased after an iteration whose kkt_after.Max() <= mu.  Started from initial_mu = 1e-3 the first
    iterations of these problems end ABOVE mu, so the gate holds mu where the ungated Solve decreases it (checked: the records differ from
    the ungated run on every problem): every kernel family -- one-tile, fused 32 / 64 / 128 grids, two y tiles, generic -- must follow the
    oracle's gated run iteration for iteration (termination, iteration count, the mu of every iteration record, the optimum)."""
    from oracle import margins as M
    n, k, m, m_r = shape
    B = 12
    hb = synth.make_batch(n, k, m, m_r, B, stream=33)
    kw = dict(initial_mu=1e-3, sigma=0.1, termination_kkt_tol=1e-9, max_iterations=30, barrier_strategy=strategy)
    ref, ungated = [], []
    for p in range(B):
        G, c, _ = orc.linearize_dense(hb.J[p], hb.r[p], hb.lam)
        qp = orc.QP(G=G, c=c, A_eq=hb.A_eq[p].T, b_eq=hb.b_eq[p], cons_var=hb.cons_var[p], cons_a=hb.cons_a[p], cons_b=hb.cons_b[p])
        o = orc.Solver(qp)
        term, its = o.solve(decrease_mu_only_on_small_error=1, **kw)
        t2, n2, v2, marg = M.solve_with_margins(qp, decrease_mu_only_on_small_error=1, **kw)
        assert (t2, n2) == (term, len(its))
        ref.append((term, [i.ip.mu for i in its], o.variables.copy(), marg))
        o0 = orc.Solver(qp)
        _, its0 = o0.solve(decrease_mu_only_on_small_error=0, **kw)
        ungated.append([i.ip.mu for i in its0])
    assert all(r[1] != u for r, u in zip(ref, ungated))             # the gate matters on every problem
    for force in (False, True):
        if force and n + k > 120:
            continue                                                # the LDS-resident generic kernel does not hold the 128 grid's shapes
        s = Q.QPInteriorPointSolver(batch_to_device(hb), force_generic=force)
        if not force:
            assert s.solve_kernel().startswith("fused"), s.solve_kernel()
        out = s.Solve(Q.Params(decrease_mu_only_on_small_error=True, **kw))
        assert torch.all(out.status == 0)
        tm, nit = out.termination_state.cpu().numpy(), out.num_iterations.cpu().numpy()
        rec, v = out.iterations.cpu().numpy(), s.variables().cpu().numpy()
        dis = M.Disagreements(f"gated Solve {shape} strategy {strategy} {'generic' if force else s.solve_kernel()}")
        for p in range(B):
            term, mus, x_ref, marg = ref[p]
            same = tm[p] == term and nit[p] == len(mus)
            dis.check(p, same, marg)
            if same:
                # PREDICTOR_CORRECTOR records sigma mu of the corrector (qp.cc:183), a cube of a ratio: a little looser
                np.testing.assert_allclose(rec[p, :len(mus), 8], mus, rtol=1e-6 if strategy == Q.PREDICTOR_CORRECTOR else 1e-9, atol=1e-300)
                np.testing.assert_allclose(v[p], x_ref, rtol=1e-7, atol=1e-9)
        assert len(dis.items) <= 1, dis.report()


@pytest.mark.parametrize("n,k,m,m_r", [(64, 8, 32, 128), (128, 16, 64, 256)])
def test_decrease_mu_only_on_small_error_fp32(n, k, m, m_r):
    """The fp32 Solve kernel with the gate: fp32 has no reference counterpart, so the referee is the fp64 fused kernel (pinned to the oracle
    above) on the same fp32-rounded inputs.  The gate must hold mu in fp32 wherever it does in fp64 while the KKT error is well above mu
    (the first iterations; later ones sit within fp32 rounding of the threshold), and the run must reach the same optimum."""
    B = 12
    hb = synth.make_batch(n, k, m, m_r, B, stream=34)
    f = lambda a: np.asarray(a, np.float32).astype(np.float64)
    kw = dict(initial_mu=1e-3, sigma=0.1, termination_kkt_tol=2e-3, termination_complementarity_tol=1e-3, max_iterations=30,
              barrier_strategy=Q.FIXED_DECREASE)
    res = {}
    for dt in (torch.float32, torch.float64):
        prob = Q.BatchedQP(n=n, k=k, m=m, J=T(f(hb.J), dt), r=T(f(hb.r), dt), lam=float(np.float32(hb.lam)), A_eq=T(f(hb.A_eq), dt),
                           b_eq=T(f(hb.b_eq), dt), cons_var=T(hb.cons_var, torch.int32), cons_a=T(f(hb.cons_a), dt), cons_b=T(f(hb.cons_b), dt))
        for gate in (False, True):
            s = Q.QPInteriorPointSolver(prob)
            assert s.solve_kernel().startswith("fused_solve"), s.solve_kernel()
            out = s.Solve(Q.Params(decrease_mu_only_on_small_error=gate, **kw))
            assert torch.all(out.status == 0)
            res[dt, gate] = (out.termination_state.cpu().numpy(), out.num_iterations.cpu().numpy(), out.iterations.double().cpu().numpy(),
                             s.variables().double().cpu().numpy())
    t32, n32, r32, v32 = res[torch.float32, True]
    t64, n64, r64, v64 = res[torch.float64, True]
    assert np.all(t32 == Q.SATISFIED_KKT_TOL) and np.all(t64 == Q.SATISFIED_KKT_TOL)
    assert not np.array_equal(res[torch.float32, False][2][:, :3, 8], r32[:, :3, 8])          # the gate changes the fp32 run ...
    np.testing.assert_allclose(r32[:, :3, 8], r64[:, :3, 8], rtol=1e-5)                        # ... exactly as it changes the fp64 one
    assert np.abs(n32 - n64).max() <= 2
    assert np.abs(v32[:, :n] - v64[:, :n]).max() <= 5e-3 * max(1.0, np.abs(v64[:, :n]).max())
