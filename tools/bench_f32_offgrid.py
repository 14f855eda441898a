"""fp32 Newton steps for a shape off the 64 / 128 grid (n a multiple of 4): the fused fp32 kernel (padded inside) against the generic kernel.
usage: python tools/bench_f32_offgrid.py [n k m m_r batch]"""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch

from mini_opt_amd import qp as Q
from mini_opt_amd import synth

n, k, m, m_r, B = (int(v) for v in (sys.argv[1:6] if len(sys.argv) >= 6 else (100, 10, 40, 200, 32768)))
hb = synth.make_batch(n, k, m, m_r, 64, stream=5)
dt = torch.float32
rep = lambda a_, d=dt: torch.as_tensor(np.ascontiguousarray(np.tile(a_, (B // 64,) + (1,) * (a_.ndim - 1))), dtype=d, device="cuda:0")
prob = Q.BatchedQP(n=n, k=k, m=m, J=rep(hb.J), r=rep(hb.r), lam=hb.lam, A_eq=rep(hb.A_eq), b_eq=rep(hb.b_eq),
                   cons_var=rep(hb.cons_var, torch.int32), cons_a=rep(hb.cons_a), cons_b=rep(hb.cons_b))
out = {"shape": [n, k, m, m_r], "batch": B}
for label, force in (("fused", False), ("generic", True)):
    s = Q.QPInteriorPointSolver(prob, force_generic=force)
    s.SetVariables(rep(hb.vars))
    mu = rep(hb.mu)
    for _ in range(2):
        s.NewtonStep(mu, 0.995)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 10 if not force else 3
    e0.record()
    for _ in range(reps):
        s.NewtonStep(mu, 0.995)
    e1.record(); torch.cuda.synchronize()
    out[label] = {"kernel": s.step_kernel(), "steps_per_s": B * reps / (e0.elapsed_time(e1) * 1e-3)}
print(json.dumps(out))
