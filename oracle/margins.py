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

Every margin carries the threshold it is held to: the base threshold of its kind, or -- where the ORACLE's own reduced KKT matrix of that
iteration is ill-conditioned -- 16 eps cond(K), the rounding error of the direction that decides it (at eps cond(K) ~ 1 the system is
singular to working precision and there is no double-precision trajectory to follow: random over-constrained problems start from slacks on the
1e-9 floor with z / s = 1e18 and reach cond(K) = 1e38).  This replaces round 3's blanket exemption of every problem the
oracle cannot converge on: the exemption now needs the condition number that justifies it, and it is logged.

A test that lets a device result differ from the oracle's must show one decision of that problem closer to its threshold than the
threshold of ITS KIND (`closeness(margins) < 1`); a disagreement on a problem whose every decision was clear of its threshold is a bug.

Not logged: the tau = 1 probe of PREDICTOR_CORRECTOR (qp.cc:174).  At tau = 1 the step length min(1, -v / dv) is continuous across
v + dv = 0, so a "tie" there is not a branch that can flip; and with mu = 0 the identity s dz + z ds = -s z drives (z + dz) / z = -ds / s
to zero as the iteration converges, so such a margin would fall below any threshold on nearly every converged run and excuse everything.
"""
import ctypes as C

import numpy as np

from . import oracle as orc

# One threshold per KIND of decision.  Termination and the mu gate compare residual norms that have converged to ~tol: an absolute rounding
# error of 1e-15 |K||x| (~1e-13 at the tests' scales) against tol = 1e-6..1e-9 is a relative margin of up to ~1e-4 at the tightest tolerance.
# A step-length tie compares v + dv with 0 relative to max(|v|, |dv|): two summation orders differ there by ~1e-12
# in early iterations (the soaks of round 3 saw ties with margins <= 5.7e-12, profiles/r03_fuzz_soak.txt) -- but the error of dv relative to
# max(|v|, |dv|) is eps times the amplification |x| / |dx| of a late iterate (1e6 - 1e8 from the seventh iteration on),
# and the first suite run under these thresholds met exactly that: n = 35, k = 31, m = 55 (test_fused_two_y_tiles_vs_generic_random_shapes),
# BOTH device kernels end (SATISFIED, 8) where the oracle runs into MAX_ITERATIONS, with z + dz = 0 to 1.7e-9 at iteration 8.  Hence 1e-8 for
# ties -- four orders below the one-size threshold of round 3; the slack floor compares a x + b with 1e-9 in a quantity of O(1).
KNIFE_EDGE = 1.0e-4   # termination / mu_gate
THRESHOLDS = {"termination": KNIFE_EDGE, "mu_gate": KNIFE_EDGE, "alpha_tie_s": 1.0e-8, "alpha_tie_z": 1.0e-8, "slack_floor": 1.0e-9}


def _kmax(e):
    return max(e.r_dual, e.r_comp, e.r_primal_eq, e.r_primal_ineq)


def _tie(v, dv):
    if v.size == 0:
        return np.inf
    scale = np.maximum(np.maximum(np.abs(v), np.abs(dv)), 1e-300)
    return float(np.min(np.abs(v + dv) / scale))


def _cond_threshold(s, base):
    """max(base, 16 eps cond(K)) for the reduced KKT matrix the oracle has just factorised (lower triangle of s.H, qp.cc:281-298)."""
    H = np.array(s.H)
    K = np.tril(H) + np.tril(H, -1).T
    if not np.all(np.isfinite(K)):
        return np.inf
    return max(base, 16.0 * np.finfo(float).eps * float(np.linalg.cond(K)))


def solve_with_margins(qp, vars0=None, **kw):
    """Returns (termination, iterations, variables, margins) with margins = [(iteration, name, relative margin, threshold), ...].
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
        margins.append((-1, "slack_floor", float(np.min(np.abs(raw - 1e-9)) / 1e-9), THRESHOLDS["slack_floor"]))
    s.evaluate_kkt(True)
    mu = s.compute_mu() if p.initialize_mu_with_complementarity else p.initial_mu
    term, n_it = orc.MAX_ITERATIONS, 0
    for it in range(p.max_iterations):
        before = s.variables.copy()
        st, ip = s.iterate(mu, p.barrier_strategy)
        if st != 0:
            return -st, n_it, s.variables.copy(), margins
        thr_tie = _cond_threshold(s, THRESHOLDS["alpha_tie_s"])
        thr_res = max(THRESHOLDS["termination"], thr_tie)
        if M:
            d = s.delta
            margins.append((it, "alpha_tie_s", _tie(before[N:N + M], d[N:N + M]), thr_tie))
            margins.append((it, "alpha_tie_z", _tie(before[N + M + K:], d[N + M + K:]), thr_tie))
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
        margins.append((it, "termination", float(flip), thr_res))
        if done:
            term = orc.SATISFIED_KKT_TOL
            break
        if p.decrease_mu_only_on_small_error:
            margins.append((it, "mu_gate", float(abs(kmax - mu) / mu) if mu > 0 else np.inf, thr_res))
        if kmax <= mu or not p.decrease_mu_only_on_small_error:
            mu = mu * p.sigma if p.barrier_strategy == orc.FIXED_DECREASE else p.sigma * cur_mu
    return term, n_it, s.variables.copy(), margins


def min_margin(margins):
    """(smallest relative margin, its (iteration, name)) of one Solve -- raw margins, for reports."""
    if not margins:
        return np.inf, None
    i = int(np.argmin([m[2] for m in margins]))
    return margins[i][2], margins[i][:2]


def closeness(margins):
    """(min over the decisions of margin / threshold of its kind, (iteration, name, raw margin)): below 1 the run sits on a knife edge."""
    if not margins:
        return np.inf, None
    ratios = [m[2] / (m[3] if len(m) > 3 else THRESHOLDS[m[1]]) for m in margins]
    i = int(np.argmin(ratios))
    return float(ratios[i]), (margins[i][0], margins[i][1], margins[i][2], margins[i][3] if len(margins[i]) > 3 else THRESHOLDS[margins[i][1]])


def on_knife_edge(margins):
    return closeness(margins)[0] < 1.0


class Disagreements:
    """Collects the problems on which a device result differs from the oracle's and enforces the rule: each of them must sit on a knife
    edge (closeness < 1: one decision nearer to its threshold than the threshold of its kind).  `report()` is what the test prints."""

    def __init__(self, label):
        self.label, self.total, self.items = label, 0, []

    def check(self, tag, agrees, margins):
        self.total += 1
        if agrees:
            return
        ratio, where = closeness(margins)
        self.items.append((tag, ratio, where))
        assert ratio < 1.0, (f"{self.label}: {tag} differs from the oracle although no decision of the oracle's run was near its threshold "
                             f"(closest: {where}, {ratio:.3g} x the knife-edge threshold of its kind)")

    def report(self):
        return (f"{self.label}: {len(self.items)} of {self.total} differ from the oracle, all on a knife edge; "
                + ", ".join(f"{tag}: margin {where[2]:.1e} at iteration {where[0]} ({where[1]}, threshold {where[3]:.1e})" for tag, _, where in self.items))
