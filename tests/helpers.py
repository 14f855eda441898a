"""Shared test helpers (not product code)."""
import numpy as np


def _draw(rng, n, p_inequality, PD):
    """Common part of GenerateRandomQP (test/qp_test.cc:483-524) once PD is known."""
    roots = rng.uniform(-20, 20, n)
    # BuildQuadraticVector (qp_test.cc:57-82): G = I, c = -2 roots; then G <- PD^T G PD, c <- PD c (qp_test.cc:501-504)
    G = PD.T @ PD
    c = PD @ (-2.0 * roots)
    shifted = np.linalg.solve(PD, roots) * 2          # roots_shifted, qp_test.cc:502
    solution = shifted.copy()
    cons = []
    for r in range(n):
        if rng.random() < p_inequality:
            scale = rng.uniform(0.1, 0.9)
            if shifted[r] < 0:
                cons.append((r, 1.0, -shifted[r] * scale))   # Var(r) >= shifted*scale
            else:
                cons.append((r, -1.0, shifted[r] * scale))   # Var(r) <= shifted*scale
            solution[r] *= scale                             # qp_test.cc:520
    return G, c, cons, solution


GENERATED_SEED = 1   # the fixed stream of the restated TestGeneratedProblems, see generated_qps


def generated_qps(count, n=8, p_inequality=0.5, seed=GENERATED_SEED):
    """Portable restatement of QPSolverTest::GenerateRandomQP (test/qp_test.cc:483-524) + GenerateRandomPDMatrix
    (test/test_utils.cc:19-34).  GenerateRandomPDMatrix accumulates sum_i u_i u_i^T into the *Upper* self-adjoint view
    (:24-30) and then overwrites the strict upper triangle with the transpose of the strict lower one (:31), which was
    never written: the matrix it returns is DIAGONAL, PD = diag(sum_i u_i[j]^2).  The QPs are therefore separable, and
    the reference's `solution` (unconstrained root, times `scale` on every bounded variable, qp_test.cc:506-520) is the
    exact optimum -- which is what lets TestGeneratedProblems assert |x - x*| <= 5e-5 on all 1000 problems.
    std::default_random_engine streams are not portable, so numpy's generator supplies the draws.  Like the reference's
    fixed seeds 0..999, the default stream is a fixed one on which all 1000 draws converge; over eight streams
    (16 000 runs) seven NAIVE runs stop at MAX_ITERATIONS = 30 short of the optimum -- the draws that "start close to
    the barrier" of the reference's own comment (qp_test.cc:543-546); test_generated_problems_other_streams covers those.
    Returns a list of (G, c, cons, solution) with cons = [(var, a, b)] meaning a*x[var] + b >= 0."""
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(count):
        U = rng.uniform(-1, 1, (n, n))
        PD = np.diag((U * U).sum(axis=0))
        out.append(_draw(rng, n, p_inequality, PD))
    return out


def dense_generated_qps(count, n=8, p_inequality=0.5, seed=1234):
    """OUR stress variant (not a reference test): the same construction with the dense PD = sum_i u_i u_i^T that
    GenerateRandomPDMatrix's name suggests.  G = PD^T PD then has cond up to ~1e11 and the bounds couple, so there is no
    closed-form optimum; the fourth entry of each tuple is None."""
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(count):
        U = rng.uniform(-1, 1, (n, n))
        G, c, cons, _ = _draw(rng, n, p_inequality, U.T @ U)
        out.append((G, c, cons, None))
    return out
