import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from mini_opt_amd import nls as NLS
from oracle import nls_oracle as N
from tests import nls_problems as P
cons = [(0, 1.0, -1.2), (1, -1.0, 0.5)]
dprob = NLS.Problem(2, P.rosenbrock_torch, cost_rows=2, inequality_constraints=cons)
g = np.array(P.ROSENBROCK_CONSTRAINED_GUESSES, float)
nls = NLS.ConstrainedNonlinearLeastSquares(dprob, batch=len(g))
out = nls.Solve(NLS.Params(max_iterations=10, max_qp_iterations=10), torch.as_tensor(g, device="cuda:0"))
print("term", out.termination_state.cpu().numpy(), "nit", out.num_iterations.cpu().numpy(), "status", out.status.cpu().numpy())
np.set_printoptions(linewidth=200, precision=6)
print(out.iterations.cpu().numpy()[1][:3])
ref = N.ConstrainedNonlinearLeastSquares(N.Problem(2, P.rosenbrock_np, inequality_constraints=cons))
term, logs = ref.solve(N.Params(max_iterations=10, max_qp_iterations=10), g[1])
print(term, len(logs))
for lg in logs[:3]: print(lg)
