import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from mini_opt_amd import qp as Q, synth
from oracle import oracle as orc
np.set_printoptions(linewidth=200, precision=6)
d = synth.CONFIGS["cfg2"]; B = 4
hb = synth.make_batch(d["n"], d["k"], d["m"], d["m_r"], B, stream=21)
dev = torch.device("cuda:0")
T = lambda a, dt=torch.float64: torch.as_tensor(np.ascontiguousarray(a), dtype=dt, device=dev)
prob = Q.BatchedQP(n=hb.n, k=hb.k, m=hb.m, J=T(hb.J), r=T(hb.r), lam=hb.lam, A_eq=T(hb.A_eq), b_eq=T(hb.b_eq), cons_var=T(hb.cons_var, torch.int32), cons_a=T(hb.cons_a), cons_b=T(hb.cons_b))
kw = dict(initial_mu=1.0, sigma=0.1, termination_kkt_tol=1e-9, max_iterations=int(sys.argv[1]) if len(sys.argv) > 1 else 12, barrier_strategy=0, initial_guess_method=0)
res = {}
for force in (False, True):
    s = Q.QPInteriorPointSolver(prob, force_generic=force); s.SetVariables(T(hb.vars))
    out = s.Solve(Q.Params(**kw))
    res[force] = (s.variables().cpu().numpy().copy(), out.iterations.cpu().numpy(), out.num_iterations.cpu().numpy())
p = 0
print("iters fused/generic", res[False][2], res[True][2])
for i in range(res[True][2][p]):
    print("it", i, "fused  ", res[False][1][p][i][:11])
    print("it", i, "generic", res[True][1][p][i][:11])
print("max |x_fused - x_generic|", np.abs(res[False][0][p] - res[True][0][p]).max())
n, k, m = hb.n, hb.k, hb.m
dv = res[False][0][p] - res[True][0][p]
print("diff blocks x,s,y,z:", np.abs(dv[:n]).max(), np.abs(dv[n:n+m]).max(), np.abs(dv[n+m:n+m+k]).max(), np.abs(dv[n+m+k:]).max())
