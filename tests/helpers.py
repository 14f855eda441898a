"""Shared test helpers (not product code)."""
import numpy as np


def generated_qps(count, n=8, p_inequality=0.5, seed=1234):
    """Portable restatement of QPSolverTest::GenerateRandomQP (test/qp_test.cc:483-524) + GenerateRandomPDMatrix
    (test/test_utils.cc:19-34): roots ~ U(-20,20), G = PD^T diag(1) PD with PD = sum_i u_i u_i^T, c = PD * (-2 roots),
    Bernoulli(p) one-sided bound per variable at scale ~ U(0.1,0.9) of the unconstrained optimum.  std::default_random_engine
    streams are not portable, so numpy's generator is used.  Returns a list of (G, c, cons) with cons = [(var, a, b)]."""
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(count):
        roots = rng.uniform(-20, 20, n)
        U = rng.uniform(-1, 1, (n, n))
        PD = U.T @ U
        G = PD.T @ PD
        c = PD @ (-2.0 * roots)
        shifted = np.linalg.solve(PD, roots) * 2
        cons = []
        for r in range(n):
            if rng.random() < p_inequality:
                scale = rng.uniform(0.1, 0.9)
                if shifted[r] < 0:
                    cons.append((r, 1.0, -shifted[r] * scale))   # Var(r) >= shifted*scale
                else:
                    cons.append((r, -1.0, shifted[r] * scale))   # Var(r) <= shifted*scale
        out.append((G, c, cons))
    return out
