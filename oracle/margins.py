"""Decision margins of the oracle's interior-point `Solve` (test infrastructure, not product code).

The device kernels and the oracle are two correct implementations of the same arithmetic in different operation orders, so they may
take a different branch only where the deciding quantity sits within rounding of its threshold.  `solve_with_margins` replays the
oracle's Solve loop (oracle/kkt_oracle.c::orc_solve, i.e. qp.cc:100-151) step by step through the oracle's own primitives and logs,
for every branch a Solve takes, how far the deciding quantity was from flipping:

  termination      kkt_after.Max() < tol && ComputeMu() < comp_tol             (qp.cc:132-137)   relative distance of the operand(s) that
                                                                                                 would have to cross for the test to flip
  mu_gate          kkt_after.Max() <= mu   (decrease_mu_only_on_small_error)   (qp.cc:140-146)   |kmax - mu| / mu
  alpha_tie        v_i + dv_i <= 0 in ComputeAlpha, for (s, ds) and (z, dz)    (qp.cc:498-503)   min_i |v_i + dv_i| / max(|v_i|, |dv_i|)
                   (a tie moves a step length between 1 and tau)
  slack_floor      s = max(1e-9, a x + b) of the initial guess                 (qp.cc:470-481)   min_i |a x + b - 1e-9| / 1e-9

A test that lets a device result differ from the oracle's must show `min_margin` of that problem below KNIFE_EDGE; a disagreement on a
problem whose every decision was clear of its threshold is a bug.
"""
import ctypes as C

import numpy as np

from . import oracle as orc

# Two implementations agree to ~1e-15 per operation, but the quantities compared here are differences of O(|K| |x|) terms that have
# converged to ~tol: an absolute rounding error of 1e-15 |K||x| (~1e-13 at the tests' scales) against tol = 1e-6..1e-9 is a relative
# margin of up to ~1e-4 at the tightest tolerance; late iterations amplify state differences by |x| / |dx|.
KNIFE_EDGE = 1.0e-4


def _kmax(e):
    return max(e.r_dual, e.r_comp, e.r_primal_eq, e.r_primal_ineq)


def _tie(v, dv):
    if v.size == 0:
        return np.inf
    scale = np.maximum(np.maximum(np.abs(v), np.abs(dv)), 1e-300)
    return float(np.min(np.abs(v + dv) / scale))


def solve_with_margins(qp, vars0=None, **kw):
    """Returns (termination, iterations, variables, margins) with margins = [(iteration, name, relative margin), ...].
    The first three are what orc.Solver(qp).solve(**kw) returns (asserted by tests/test_oracle_golden.py)."""
    s = orc.Solver(qp)
    p = orc._Params()
    s.L.orc_default_params(C.byref(p))
    for key, val in kw.items():
        if not hasattr(p, key):
            raise KeyError(key)
        setattr(p, key, val)
    if vars0 is not None:
        s.variables[:] = vars0
    margins = []
    st = s.L.orc_initial_guess(C.byref(s._s), C.byref(p))
    if st != 0:
        return -st, 0, s.variables.copy(), margins
    N, M, K = s.N, s.M, s.K
    if M and p.initial_guess_method != orc.GUESS_USER_PROVIDED:
        x = s.variables[:N]
        raw = qp.cons_a * x[qp.cons_var] + qp.cons_b
        margins.append((-1, "slack_floor", float(np.min(np.abs(raw - 1e-9)) / 1e-9)))
    s.evaluate_kkt(True)
    mu = s.compute_mu() if p.initialize_mu_with_complementarity else p.initial_mu
    term, n_it = orc.MAX_ITERATIONS, 0
    for it in range(p.max_iterations):
        before = s.variables.copy()
        st, ip = s.iterate(mu, p.barrier_strategy)
        if st != 0:
            return -st, n_it, s.variables.copy(), margins
        if M:
            d = s.delta
            margins.append((it, "alpha_tie_s", _tie(before[N:N + M], d[N:N + M])))
            margins.append((it, "alpha_tie_z", _tie(before[N + M + K:], d[N + M + K:])))
            if p.barrier_strategy == orc.PREDICTOR_CORRECTOR:
                da = s.delta_affine
                margins.append((it, "probe_tie_s", _tie(before[N:N + M], da[N:N + M])))
                margins.append((it, "probe_tie_z", _tie(before[N + M + K:], da[N + M + K:])))
        s.evaluate_kkt(True)
        kmax = _kmax(s.compute_errors(mu))
        cur_mu = s.compute_mu()
        n_it = it + 1
        a = kmax / p.termination_kkt_tol - 1.0
        b = cur_mu / p.termination_complementarity_tol - 1.0
        done = a < 0 and b < 0
        if done:
            flip = min(abs(a), abs(b))                                   # either operand crossing ends the agreement
        else:
            flip = max(abs(v) for v in (a, b) if v >= 0)                 # every operand on the wrong side has to cross
        margins.append((it, "termination", float(flip)))
        if done:
            term = orc.SATISFIED_KKT_TOL
            break
        if p.decrease_mu_only_on_small_error:
            margins.append((it, "mu_gate", float(abs(kmax - mu) / mu) if mu > 0 else np.inf))
        if kmax <= mu or not p.decrease_mu_only_on_small_error:
            mu = mu * p.sigma if p.barrier_strategy == orc.FIXED_DECREASE else p.sigma * cur_mu
    return term, n_it, s.variables.copy(), margins


def min_margin(margins):
    """(smallest relative margin, its (iteration, name)) of one Solve."""
    if not margins:
        return np.inf, None
    i = int(np.argmin([m[2] for m in margins]))
    return margins[i][2], margins[i][:2]


class Disagreements:
    """Collects the problems on which a device result differs from the oracle's and enforces the rule: each of them must sit on a knife
    edge (min_margin < KNIFE_EDGE).  `report()` is what the test prints / asserts at the end."""

    def __init__(self, label):
        self.label, self.total, self.items = label, 0, []

    def check(self, tag, agrees, margins):
        self.total += 1
        if agrees:
            return
        mm, where = min_margin(margins)
        self.items.append((tag, mm, where))
        assert mm < KNIFE_EDGE, (f"{self.label}: {tag} differs from the oracle although no decision of the oracle's run was closer than "
                                 f"{mm:.3e} (relative) to its threshold ({where})")

    def report(self):
        return (f"{self.label}: {len(self.items)} of {self.total} differ from the oracle, all on a knife edge; margins "
                + ", ".join(f"{mm:.1e}@{where}" for _, mm, where in self.items))
