"""Full-size GPU checks at BASELINE.json's batch sizes through size-independent properties (the oracle only sees a sample):
 * every problem of the batch satisfies the reduced KKT Newton equations (qp.cc:255-268, 359-363), evaluated independently
   with torch fp64 batched products on the device;
 * two launches give bit-identical results; ragged batch sizes work; the sampled problems agree with the oracle."""
import numpy as np
import pytest
import torch

from mini_opt_amd import qp as Q
from mini_opt_amd import synth
from oracle import oracle as orc

pytestmark = pytest.mark.gpu


def newton_equation_residuals(prob, vars_, mu, delta, chunk=8192):
    """max over the batch of the relative residuals of  (G+Sigma) dx - A^T dy = -r_aug,  A dx = -r_pe,
    ds = a dx_v + r_pi,  dz = -(z/s) ds - (r_comp - mu)/s   (all fp64, torch)."""
    n, k, m = prob.n, prob.k, prob.m
    worst = torch.zeros(4, dtype=torch.float64, device=vars_.device)
    B = vars_.shape[0]
    for b0 in range(0, B, chunk):
        sl = slice(b0, min(B, b0 + chunk))
        J = prob.J[sl].double(); r = prob.r[sl].double()
        G = torch.bmm(J.transpose(1, 2), J) + prob.lam * torch.eye(n, dtype=torch.float64, device=J.device)
        c = torch.bmm(J.transpose(1, 2), r.unsqueeze(2)).squeeze(2)
        A = prob.A_eq[sl].double().transpose(1, 2)  # [B, k, n]
        v = vars_[sl].double(); d = delta[sl].double(); mus = mu[sl].double().unsqueeze(1)
        x, s, y, z = v[:, :n], v[:, n:n + m], v[:, n + m:n + m + k], v[:, n + m + k:]
        dx, ds, dy, dz = d[:, :n], d[:, n:n + m], d[:, n + m:n + m + k], d[:, n + m + k:]
        var = prob.cons_var[sl].long(); a = prob.cons_a[sl].double(); bb = prob.cons_b[sl].double()
        r_pi = a * torch.gather(x, 1, var) + bb - s
        r_comp = s * z
        sig = a * (z / s) * a
        rho = a * (z / s) * r_pi + a * (r_comp - mus) / s
        diag = torch.zeros_like(x).scatter_add_(1, var, sig)
        r_aug = torch.bmm(G, x.unsqueeze(2)).squeeze(2) + c - torch.bmm(A.transpose(1, 2), y.unsqueeze(2)).squeeze(2)
        r_aug = r_aug - torch.zeros_like(x).scatter_add_(1, var, a * z) + torch.zeros_like(x).scatter_add_(1, var, rho)
        e1 = torch.bmm(G, dx.unsqueeze(2)).squeeze(2) + diag * dx - torch.bmm(A.transpose(1, 2), dy.unsqueeze(2)).squeeze(2) + r_aug
        r_pe = torch.bmm(A, x.unsqueeze(2)).squeeze(2) + prob.b_eq[sl].double()
        e2 = torch.bmm(A, dx.unsqueeze(2)).squeeze(2) + r_pe
        e3 = ds - (a * torch.gather(dx, 1, var) + r_pi)
        e4 = dz - (-(z / s) * ds - (r_comp - mus) / s)
        scale = lambda t: t.abs().amax(dim=1).clamp_min(1.0)
        errs = torch.stack([(e1.abs().amax(dim=1) / scale(r_aug)).max(), (e2.abs().amax(dim=1) / scale(r_pe)).max(),
                            (e3.abs().amax(dim=1) / scale(ds)).max(), (e4.abs().amax(dim=1) / scale(dz)).max()])
        worst = torch.maximum(worst, errs)
    return worst.cpu().numpy()


@pytest.mark.parametrize("cfg,batch", [("cfg2", 4096), ("cfg3", 65536), ("cfg3", 70001), ("cfg3", 131072)],
                         ids=["cfg2", "cfg3", "cfg3-ragged", "cfg5-shard"])
def test_full_batch_satisfies_newton_equations(cfg, batch):
    """The last case is ONE shard of BASELINE.json configs[4] (2^20 QPs of the cfg3 shape over 8 GPUs = 131072 problems, 9.6 GB of
    J, per GPU -- sharding.shard_range(2**20, r, 8)), launched as a single mo_newton_step like `bench.py --gpus 8` does per rank."""
    if batch == 131072:
        from mini_opt_amd import sharding
        assert [b - a for a, b in (sharding.shard_range(2 ** 20, r, 8) for r in range(8))] == [batch] * 8
    d = synth.CONFIGS[cfg]
    dev = torch.device("cuda:0")
    prob, vars_, mu = synth.make_batch_torch(d["n"], d["k"], d["m"], d["m_r"], batch, dev, torch.float64, seed=1234 + batch)
    solver = Q.QPInteriorPointSolver(prob)
    assert solver.step_kernel().startswith("fused")
    solver.SetVariables(vars_)
    delta, alpha, status = solver.NewtonStep(mu, 0.995)
    d1, a1 = delta.clone(), alpha.clone()
    assert int((status != 0).sum()) == 0
    assert torch.isfinite(delta).all() and torch.isfinite(alpha).all()
    errs = newton_equation_residuals(prob, vars_, mu, d1)
    assert errs.max() < 1e-9, errs
    # alpha in (0, 1], and the step keeps s, z positive (qp.cc:485-507 with tau = 0.995)
    n, k, m = d["n"], d["k"], d["m"]
    assert (alpha > 0).all() and (alpha <= 1).all()
    s_new = vars_[:, n:n + m] + alpha[:, :1] * d1[:, n:n + m]
    z_new = vars_[:, n + m + k:] + alpha[:, 1:] * d1[:, n + m + k:]
    assert (s_new > 0).all() and (z_new > 0).all()
    # determinism: a second launch is bit-identical
    delta2, alpha2, _ = solver.NewtonStep(mu, 0.995)
    assert torch.equal(delta2, d1) and torch.equal(alpha2, a1)
    # oracle on a sample spread over the batch (first, last and strided problems)
    idx = torch.unique(torch.cat([torch.arange(0, 64), torch.arange(batch - 64, batch), torch.arange(0, batch, max(1, batch // 128))])).to(dev)
    h = lambda t: t.index_select(0, idx).double().cpu().numpy()
    ref, ref_alpha, ref_status, _ = orc.batched_newton_step(
        n, k, m, J=h(prob.J), r=h(prob.r), lam=prob.lam, A_eq=h(prob.A_eq), b_eq=h(prob.b_eq),
        cons_var=prob.cons_var.index_select(0, idx).cpu().numpy(), cons_a=h(prob.cons_a), cons_b=h(prob.cons_b), vars_=h(vars_), mu=h(mu))
    got = h(d1)
    err = np.max(np.abs(got - ref), axis=1) / np.max(np.abs(ref), axis=1)
    assert np.all(ref_status == 0) and err.max() < 1e-10, err.max()
    np.testing.assert_allclose(h(a1), ref_alpha, rtol=0, atol=1e-9)


@pytest.mark.parametrize("cfg,batch", [("cfg2", 4096), ("cfg3", 16384)])
def test_full_batch_solve_reaches_kkt_point(cfg, batch):
    """Row f1 at batch scale: the fused on-device Solve terminates with SATISFIED_KKT_TOL for every problem, and the returned
    state is certified independently (torch fp64): stationarity, primal / dual feasibility and complementarity of the convex QP."""
    d = synth.CONFIGS[cfg]
    dev = torch.device("cuda:0")
    n, k, m = d["n"], d["k"], d["m"]
    prob, vars_, mu = synth.make_batch_torch(n, k, m, d["m_r"], batch, dev, torch.float64, seed=4321)
    solver = Q.QPInteriorPointSolver(prob)
    out = solver.Solve(Q.Params(initial_mu=1.0, sigma=0.1, termination_kkt_tol=1e-9, max_iterations=20,
                                initial_guess_method=Q.SOLVE_EQUALITY_CONSTRAINED))
    assert int((out.status != 0).sum()) == 0
    assert int((out.termination_state != Q.SATISFIED_KKT_TOL).sum()) == 0
    assert int(out.num_iterations.max()) <= 20 and int(out.num_iterations.min()) >= 1
    v = solver.variables().double()
    x, s, y, z = v[:, :n], v[:, n:n + m], v[:, n + m:n + m + k], v[:, n + m + k:]
    worst = torch.zeros(4, dtype=torch.float64, device=dev)
    for b0 in range(0, batch, 4096):
        sl = slice(b0, min(batch, b0 + 4096))
        J = prob.J[sl]; r = prob.r[sl]
        G = torch.bmm(J.transpose(1, 2), J) + prob.lam * torch.eye(n, dtype=torch.float64, device=dev)
        c = torch.bmm(J.transpose(1, 2), r.unsqueeze(2)).squeeze(2)
        A = prob.A_eq[sl].transpose(1, 2)
        var = prob.cons_var[sl].long(); a = prob.cons_a[sl]; bb = prob.cons_b[sl]
        grad = torch.bmm(G, x[sl].unsqueeze(2)).squeeze(2) + c - torch.bmm(A.transpose(1, 2), y[sl].unsqueeze(2)).squeeze(2)
        grad = grad - torch.zeros_like(grad).scatter_add_(1, var, a * z[sl])
        feas_eq = torch.bmm(A, x[sl].unsqueeze(2)).squeeze(2) + prob.b_eq[sl]
        ci = a * torch.gather(x[sl], 1, var) + bb
        worst = torch.maximum(worst, torch.stack([grad.abs().max(), feas_eq.abs().max(), (-ci).clamp_min(0).max(), (ci * z[sl]).abs().max()]))
    w = worst.cpu().numpy()
    assert w[0] < 1e-8 and w[1] < 1e-8 and w[2] < 1e-8 and w[3] < 1e-4, w   # complementarity tolerance 1e-6 on the mean
    assert bool((z >= 0).all()) and bool((s > 0).all())


def test_full_batch_fp32_cfg4_satisfies_newton_equations():
    """BASELINE configs[3] at its full batch through the fused fp32 kernel: every problem satisfies the reduced KKT Newton equations
    evaluated in fp64 on the fp32 inputs (relative residual at fp32 rounding level), launches are bit-identical, and a sample agrees
    with the fp64 oracle within the fp32 tolerance."""
    d = synth.CONFIGS["cfg4"]
    dev = torch.device("cuda:0")
    batch = d["batch"]
    prob, vars_, mu = synth.make_batch_torch(d["n"], d["k"], d["m"], d["m_r"], batch, dev, torch.float32, seed=4321)
    solver = Q.QPInteriorPointSolver(prob)
    assert solver.step_kernel() == "fused_mfma_f32_n128"
    solver.SetVariables(vars_)
    delta, alpha, status = solver.NewtonStep(mu, 0.995)
    d1, a1 = delta.clone(), alpha.clone()
    assert int((status != 0).sum()) == 0
    assert torch.isfinite(delta).all() and torch.isfinite(alpha).all()
    errs = newton_equation_residuals(prob, vars_, mu, d1, chunk=4096)
    assert errs.max() < 2e-4, errs                                 # fp32 arithmetic: ~cond * 6e-8
    assert (alpha > 0).all() and (alpha <= 1).all()
    delta2, alpha2, _ = solver.NewtonStep(mu, 0.995)
    assert torch.equal(delta2, d1) and torch.equal(alpha2, a1)
    n, k, m = d["n"], d["k"], d["m"]
    idx = torch.arange(0, batch, batch // 96, device=dev)
    h = lambda t: t.index_select(0, idx).double().cpu().numpy()
    ref, ref_alpha, ref_status, _ = orc.batched_newton_step(
        n, k, m, J=h(prob.J), r=h(prob.r), lam=float(np.float32(prob.lam)), A_eq=h(prob.A_eq), b_eq=h(prob.b_eq),
        cons_var=prob.cons_var.index_select(0, idx).cpu().numpy(), cons_a=h(prob.cons_a), cons_b=h(prob.cons_b), vars_=h(vars_), mu=h(mu))
    got = h(d1)
    err = np.max(np.abs(got - ref), axis=1) / np.max(np.abs(ref), axis=1)
    assert err.max() < 2e-3, err.max()
