import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch, time
from mini_opt_amd import qp as Q, synth
from oracle import oracle as orc
d = synth.CONFIGS["cfg4"]
B = 512
hb = synth.make_batch(d["n"], d["k"], d["m"], d["m_r"], B, stream=11)
rd = lambda a: a.astype(np.float32).astype(np.float64)
for key in ("J", "r", "A_eq", "b_eq", "cons_a", "cons_b", "vars", "mu"):
    setattr(hb, key, rd(getattr(hb, key)))
hb.lam = float(np.float32(hb.lam))
T = lambda a, dt=torch.float32: torch.as_tensor(np.ascontiguousarray(a), dtype=dt, device="cuda:0")
ref, ref_alpha, ref_status, _ = orc.batched_newton_step(hb.n, hb.k, hb.m, J=hb.J, r=hb.r, lam=hb.lam, A_eq=hb.A_eq, b_eq=hb.b_eq,
    cons_var=hb.cons_var, cons_a=hb.cons_a, cons_b=hb.cons_b, vars_=hb.vars, mu=hb.mu)
for force in (True, False):
    prob = Q.BatchedQP(n=hb.n, k=hb.k, m=hb.m, J=T(hb.J), r=T(hb.r), lam=hb.lam, A_eq=T(hb.A_eq), b_eq=T(hb.b_eq),
                       cons_var=T(hb.cons_var, torch.int32), cons_a=T(hb.cons_a), cons_b=T(hb.cons_b))
    s = Q.QPInteriorPointSolver(prob, force_generic=force)
    s.SetVariables(T(hb.vars))
    delta, alpha, status = s.NewtonStep(T(hb.mu), 0.995)
    got = delta.double().cpu().numpy()
    err = np.max(np.abs(got - ref), axis=1) / np.max(np.abs(ref), axis=1)
    print(s.step_kernel(), "status ok", int((status == 0).sum()), "rel-inf max %.3g  p99 %.3g  median %.3g" % (err.max(), np.percentile(err, 99), np.median(err)),
          "alpha err %.3g" % np.abs(alpha.double().cpu().numpy() - ref_alpha).max())
