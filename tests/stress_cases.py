"""Hard inputs for the un-pivoted block LDL^T and the x+ formulation of the step kernel (test infrastructure, not product code),
and an extended-precision "truth" for the Newton direction they are measured against.

 late_states      states taken from iteration >= 6 of an interior-point Solve: s and z span many decades (z/s from 1e-9 to 1e9) and
                  |delta| << |x|, where the x+ form of mo_newton_step (solve for the new iterate, subtract x) cancels
 ill_conditioned  J-level problems whose G = J^T J + lambda I has cond 1e6 ... 1e10 (column scaling of J)
 rank_deficient   lambda = 0 and m_r < n: G is singular; equalities + barrier terms make the KKT matrix regular, but the leading block
                  G + Sigma that an elimination in natural order (x first, then y) pivots on may be singular or nearly so
"""
import numpy as np

from mini_opt_amd import synth


def truth_direction(G, c, A, b, cons_var, cons_a, cons_b, v, mu):
    """The unreduced Newton system of the interior-point step (qp.cc:595-655 is its mu = 0 form) solved in long double
    (x87 80-bit, eps 1.1e-19) by Gaussian elimination with partial pivoting.  Returns delta = [dx | ds | dy | dz] as float64."""
    ld = np.longdouble
    n, k, m = G.shape[0], A.shape[0], len(cons_var)
    V = n + 2 * m + k
    Gs = np.tril(G) + np.tril(G, -1).T
    x, s, y, z = v[:n], v[n:n + m], v[n + m:n + m + k], v[n + m + k:]
    Ai = np.zeros((m, n))
    for i in range(m):
        Ai[i, cons_var[i]] = cons_a[i]
    K = np.zeros((V, V), dtype=ld)
    rhs = np.zeros(V, dtype=ld)
    ox, os_, oy, oz = 0, n, n + m, n + m + k
    K[ox:ox + n, ox:ox + n] = Gs
    K[ox:ox + n, oy:oy + k] = -A.T
    K[ox:ox + n, oz:oz + m] = -Ai.T
    rhs[ox:ox + n] = -(Gs.astype(ld) @ x.astype(ld) + c - A.T.astype(ld) @ y.astype(ld) - Ai.T.astype(ld) @ z.astype(ld))
    for i in range(m):                       # z_i ds_i + s_i dz_i = -(s_i z_i - mu)
        K[os_ + i, os_ + i] = z[i]
        K[os_ + i, oz + i] = s[i]
        rhs[os_ + i] = -(ld(s[i]) * ld(z[i]) - ld(mu))
    K[oy:oy + k, ox:ox + n] = A
    rhs[oy:oy + k] = -(A.astype(ld) @ x.astype(ld) + b)
    K[oz:oz + m, ox:ox + n] = Ai
    for i in range(m):
        K[oz + i, os_ + i] = -1.0
    rhs[oz:oz + m] = -(Ai.astype(ld) @ x.astype(ld) + cons_b - s)
    for col in range(V):                     # partial-pivot elimination, row operations vectorised
        piv = col + int(np.argmax(np.abs(K[col:, col])))
        if piv != col:
            K[[col, piv]] = K[[piv, col]]
            rhs[[col, piv]] = rhs[[piv, col]]
        f = K[col + 1:, col] / K[col, col]
        K[col + 1:, col:] -= np.outer(f, K[col, col:])
        rhs[col + 1:] -= f * rhs[col]
    sol = np.zeros(V, dtype=ld)
    for row in range(V - 1, -1, -1):
        sol[row] = (rhs[row] - K[row, row + 1:] @ sol[row + 1:]) / K[row, row]
    return sol.astype(np.float64)


def late_states(cfg, count, iterations, seed_stream=31):
    """(batch, list of per-problem mu): the synthetic problems of `cfg` with their state replaced by the oracle's state after
    `iterations` interior-point iterations (COMPLEMENTARITY, sigma 0.1); mu = sigma * s.z / m as Solve would pass next (qp.cc:140-146)."""
    from oracle import oracle as orc
    d = synth.CONFIGS[cfg]
    hb = synth.make_batch(d["n"], d["k"], d["m"], d["m_r"], count, stream=seed_stream)
    mus = np.zeros(count)
    for p in range(count):
        G, c, _ = orc.linearize_dense(hb.J[p], hb.r[p], hb.lam)
        o = orc.Solver(orc.QP(G=G, c=c, A_eq=hb.A_eq[p].T, b_eq=hb.b_eq[p], cons_var=hb.cons_var[p], cons_a=hb.cons_a[p],
                              cons_b=hb.cons_b[p]))
        o.solve(initial_mu=1.0, sigma=0.1, termination_kkt_tol=1e-300, termination_complementarity_tol=1e-300,
                max_iterations=iterations, initial_guess_method=orc.GUESS_SOLVE_EQUALITY_CONSTRAINED)
        hb.vars[p] = o.variables
        mus[p] = 0.1 * o.compute_mu()
    hb.mu = mus
    return hb


def ill_conditioned(n, k, m, m_r, count, log10_cond, seed_stream=41):
    """Synthetic problems whose columns of J are scaled by 10^(-log10_cond/2 * i/(n-1)): cond(J^T J) ~ 10^log10_cond x cond(U^T U).
    lambda is lowered so that it does not mask the conditioning."""
    hb = synth.make_batch(n, k, m, m_r, count, stream=seed_stream + int(log10_cond))
    scale = 10.0 ** (-0.5 * log10_cond * np.arange(n) / (n - 1))
    hb.J = hb.J * scale[None, None, :]
    hb.lam = 0.0
    return hb


def rank_deficient(n, k, m, m_r, count, seed_stream=51):
    """lambda = 0, m_r < n residual rows (G = J^T J of rank m_r), k equalities, m/2 two-sided boxes: m_r + k + m/2 >= n makes the KKT
    matrix regular in general position while G + Sigma (rank <= m_r + m/2) is singular whenever m_r + m/2 < n."""
    assert m_r < n and m_r + k + m // 2 >= n
    hb = synth.make_batch(n, k, m, m_r, count, stream=seed_stream)
    hb.lam = 0.0
    return hb


def dense_problem(hb, p):
    """(G lower+upper dense, c, A (k x n), b, cons...) of problem p of a synth.Batch, in float64."""
    J = hb.J[p]
    G = J.T @ J + hb.lam * np.eye(hb.n)
    c = J.T @ hb.r[p]
    return G, c, hb.A_eq[p].T.copy(), hb.b_eq[p], hb.cons_var[p], hb.cons_a[p], hb.cons_b[p]
