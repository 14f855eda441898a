"""The reference's nonlinear-least-squares test problems (test/nonlinear_test.cc) as data: residual functions written for
numpy (oracle side) and for torch batches (device side), initial guesses, parameters and the optima the reference asserts.
These are fixtures (problem definitions and expected answers), not reference source."""
import math

import numpy as np

SQRT_B = math.sqrt(100.0)


# ---- Rosenbrock, nonlinear_test.cc:375-386: h = [a - x, sqrt(b) (y - x^2)]
def rosenbrock_np(x, want_J):
    r = np.array([1.0 - x[0], SQRT_B * (x[1] - x[0] * x[0])])
    J = np.array([[-1.0, 0.0], [-2.0 * x[0] * SQRT_B, SQRT_B]]) if want_J else None
    return r, J


def rosenbrock_torch(x, want_J):
    import torch
    r = torch.stack([1.0 - x[:, 0], SQRT_B * (x[:, 1] - x[:, 0] * x[:, 0])], dim=1)
    J = None
    if want_J:
        J = torch.zeros(x.shape[0], 2, 2, dtype=x.dtype, device=x.device)
        J[:, 0, 0] = -1.0
        J[:, 1, 0] = -2.0 * x[:, 0] * SQRT_B
        J[:, 1, 1] = SQRT_B
    return r, J


ROSENBROCK_GUESSES = [(-5, -3), (10, 8), (-20, 3), (0, -5), (4, 0), (100, 50), (-35, 40), (1000, -50), (0.8, -0.3)]
ROSENBROCK_CONSTRAINED_GUESSES = [(12, -5), (100.0, -20.0), (1423.0, -400.0), (-20.0, 10.0), (-120.0, 35.0), (-50.0, 0.5)]


# ---- Rosenbrock 6D, nonlinear_test.cc:502-521
def rosenbrock6_np(x, want_J):
    r = np.zeros(10)
    J = np.zeros((10, 6)) if want_J else None
    for i in range(5):
        r[2 * i] = 1.0 - x[i]
        r[2 * i + 1] = SQRT_B * (x[i + 1] - x[i] * x[i])
        if want_J:
            J[2 * i, i] = -1.0
            J[2 * i + 1, i] = -2.0 * x[i] * SQRT_B
            J[2 * i + 1, i + 1] = SQRT_B
    return r, J


def rosenbrock6_torch(x, want_J):
    import torch
    B = x.shape[0]
    r = torch.zeros(B, 10, dtype=x.dtype, device=x.device)
    J = torch.zeros(B, 10, 6, dtype=x.dtype, device=x.device) if want_J else None
    for i in range(5):
        r[:, 2 * i] = 1.0 - x[:, i]
        r[:, 2 * i + 1] = SQRT_B * (x[:, i + 1] - x[:, i] * x[:, i])
        if want_J:
            J[:, 2 * i, i] = -1.0
            J[:, 2 * i + 1, i] = -2.0 * x[:, i] * SQRT_B
            J[:, 2 * i + 1, i + 1] = SQRT_B
    return r, J


ROSENBROCK6_GUESSES = [(10.5, -8.0, 50.0, -14.0, 4.0, -0.6), (100.0, -50.0, 30.0, -100.0, 150.0, -400.0)]
ROSENBROCK6_SOLUTION = (2.3, -1.2, 3.0, -2.5, 6.19802, 6.19802 ** 2)
ROSENBROCK6_CONSTRAINTS = [(0, 1.0, -2.3), (1, -1.0, -1.2), (2, 1.0, -3.0), (3, -1.0, -2.5)]  # x0>=2.3, x1<=-1.2, x2>=3, x3<=-2.5


# ---- Himmelblau, nonlinear_test.cc:578-593: two residuals x^2 + y - 11, x + y^2 - 7
def himmelblau_np(x, want_J):
    r = np.array([x[0] ** 2 + x[1] - 11.0, x[0] + x[1] ** 2 - 7.0])
    J = np.array([[2.0 * x[0], 1.0], [1.0, 2.0 * x[1]]]) if want_J else None
    return r, J


def himmelblau_torch(x, want_J):
    import torch
    r = torch.stack([x[:, 0] ** 2 + x[:, 1] - 11.0, x[:, 0] + x[:, 1] ** 2 - 7.0], dim=1)
    J = None
    if want_J:
        J = torch.ones(x.shape[0], 2, 2, dtype=x.dtype, device=x.device)
        J[:, 0, 0] = 2.0 * x[:, 0]
        J[:, 1, 1] = 2.0 * x[:, 1]
    return r, J


HIMMELBLAU_SOLUTIONS = [(3.0, 2.0), (-2.805118, 3.131312), (-3.779310, -3.283186), (3.584428, -1.848126)]


def box(lo, hi, nvars=2):
    """Var(i) >= lo, Var(i) <= hi as (variable, a, b) with a x + b >= 0 (qp.hpp:72-95)."""
    out = []
    for i in range(nvars):
        out.append((i, 1.0, -lo))
        out.append((i, -1.0, hi))
    return out


def himmelblau_guesses():
    g = []
    x = -4.5
    while x <= 4.5:
        y = -4.5
        while y <= 4.5:
            g.append((x, y))
            y += 0.3
        x += 0.3
    return g


def himmelblau_quadrant_guesses():
    g = []
    x = 0.2
    while x <= 4.8:
        y = 0.2
        while y <= 4.8:
            g.append((x, y))
            y += 0.2
        x += 0.2
    return g


# ---- sphere with product equality constraints, nonlinear_test.cc:722-826: cost h = x (6 vars), x0 x1 = 4, x2 x3 = 9
def sphere_np(x, want_J):
    return np.array(x, float), (np.eye(6) if want_J else None)


def sphere_eq_np(x, want_J):
    r = np.array([x[0] * x[1] - 4.0, x[2] * x[3] - 9.0])
    J = None
    if want_J:
        J = np.zeros((2, 6))
        J[0, 0], J[0, 1] = x[1], x[0]
        J[1, 2], J[1, 3] = x[3], x[2]
    return r, J


def sphere_torch(x, want_J):
    import torch
    J = torch.eye(6, dtype=x.dtype, device=x.device).expand(x.shape[0], 6, 6).contiguous() if want_J else None
    return x.clone(), J


def sphere_eq_torch(x, want_J):
    import torch
    r = torch.stack([x[:, 0] * x[:, 1] - 4.0, x[:, 2] * x[:, 3] - 9.0], dim=1)
    J = None
    if want_J:
        J = torch.zeros(x.shape[0], 2, 6, dtype=x.dtype, device=x.device)
        J[:, 0, 0], J[:, 0, 1] = x[:, 1], x[:, 0]
        J[:, 1, 2], J[:, 1, 3] = x[:, 3], x[:, 2]
    return r, J


SPHERE_SOLUTIONS = [(a, a, b, b, 0.0, 0.0) for a in (-2.0, 2.0) for b in (-3.0, 3.0)]


def sphere_guesses(count=100, seed=7):
    """The reference draws U(-30, 30) from std::default_random_engine{7} (not portable); any guesses serve the test's purpose."""
    rng = np.random.default_rng(seed)
    return [tuple(rng.uniform(-30.0, 30.0, 6)) for _ in range(count)]


def nullspace_kats():
    """TestNullSpaceSolver1 / 2 (qp_test.cc:576-707): G = sum J^T J, c = sum J^T r at x = 0, one equality block, and the
    values the reference asserts (index, value, tolerance)."""
    A = np.array([[-2.0, 1.4], [2.2, -3.5]]); b = np.array([-0.8, 1.3])
    k1 = dict(name="nullspace1", G=A.T @ A, c=A.T @ (-b), A_eq=np.array([[0.0, 1.0]]), b_eq=np.array([-0.3]),
              expected=[(1, 0.3, 1e-12), (0, 0.860859728506787, 1e-12)])
    G = np.zeros((4, 4)); c = np.zeros(4)
    for idx, M, v in (((0, 1), np.array([[1.7, -0.2], [2.3, 1.2]]), np.array([5.4, -3.4])),
                      ((1, 2), np.array([[-5.0, 3.3], [9.1, 1.9]]), np.array([-3.3, 4.4])),
                      ((0, 3), np.array([[0.2, -0.5], [1.1, -3.1]]), np.array([0.5, 0.0]))):
        J = np.zeros((2, 4)); J[:, list(idx)] = M
        G += J.T @ J; c += J.T @ (-v)
    Aeq = np.zeros((2, 4)); Aeq[:, [1, 3]] = np.array([[1.1, -1.1], [0.3, 0.6]])
    k2 = dict(name="nullspace2", G=G, c=c, A_eq=Aeq, b_eq=-np.array([1.3, -4.0]),
              expected=[(1, -3.656565656565657, 1e-14), (3, -4.838383838383838, 1e-14), (0, -0.707724112814252, 1e-13),
                        (2, 0.0247370254266801, 1e-13)])
    return [k1, k2]


# ---- kinematic-chain robots, nonlinear_test.cc:828-1136 (chains: test/transform_chains.cc) -----------------------------------------
# A problem is data: chains of links (base euler-xyz rotation, translation, 6 activity flags, the index in x of every active parameter)
# and residual rows that are affine in the effector translations of the chains and in x:
#   row = const + sum_i lin[i] x_i + sum_c (wx, wy, wz) . t_effector(chain c)
# Both the numpy evaluator below (oracle side, on oracle/chain_oracle.py) and the device family MO_RESIDUAL_ACTUATOR_CHAIN consume it.
def chain_link(translation, mask=(0, 0, 0, 0, 0, 0), params=(), rotation_xyz=(0.0, 0.0, 0.0)):
    assert sum(1 for v in mask if v) == len(params)
    return dict(rotation_xyz=tuple(rotation_xyz), translation=tuple(translation), mask=tuple(mask), params=tuple(params))


def chain_row(const=0.0, lin=None, terms=()):
    return dict(const=float(const), lin=dict(lin or {}), terms=[(int(c), float(wx), float(wy), float(wz)) for c, wx, wy, wz in terms])


Z = (0, 0, 1, 0, 0, 0)   # only the z angle of the link is optimised (nonlinear_test.cc:833, 967)

# TestTwoAngleActuatorChain, nonlinear_test.cc:828-964: two revolute joints, effector 0.4 beyond the second; cost: y -> 0.6, equality: x = 0.45
TWO_ANGLE = dict(
    n=2,
    chains=[[chain_link((0.0, 0.0, 0.0), Z, (0,)), chain_link((0.4, 0.0, 0.0), Z, (1,)), chain_link((0.4, 0.0, 0.0))]],
    cost_rows=[chain_row(-0.6, terms=[(0, 0, 1, 0)])],
    eq_rows=[chain_row(-0.45, terms=[(0, 1, 0, 0)])],
    target_xy=(0.45, 0.6),
    inequalities_stage2=[(1, 1.0, 0.0), (1, -1.0, math.pi)],     # Var(1) >= 0, Var(1) <= pi, :917-918
    params=dict(max_iterations=50, max_qp_iterations=1, relative_exit_tol=1e-12, absolute_first_derivative_tol=1e-10, absolute_exit_tol=1e-9,
                termination_kkt_tolerance=1e-6, max_line_search_iterations=10, equality_penalty_initial=0.01, line_search_strategy=0,
                lambda_failure_init=0.001, armijo_search_tau=0.5, lambda_initial=0.001, min_lambda=1e-9))   # :881-901 (ARMIJO_BACKTRACK = 0)


def two_angle_guesses(stage):
    g = []
    t0 = 0.1
    while t0 <= math.pi / 2:                                   # :904-909 / :920-925
        t1 = -math.pi / 3 if stage == 1 else 1e-3
        hi = math.pi / 3 if stage == 1 else math.pi / 2 - 1e-3
        while t1 <= hi:
            g.append((t0, t1))
            t1 += 0.1
        t0 += 0.1
    return g


# TestDualActuatorBalancing, nonlinear_test.cc:966-1136: body angle x0 shared by two legs; rear leg x1, x2; front leg x3, x4
_ORIGIN = (0.0, 0.4, 0.0)
DUAL = dict(
    n=5,
    chains=[[chain_link(_ORIGIN, Z, (0,)), chain_link((0.0, 0.0, 0.0), Z, (1,)), chain_link((0.3, 0.0, 0.0), Z, (2,)), chain_link((0.3, 0.0, 0.0))],      # rear, :981-986
            [chain_link(_ORIGIN, Z, (0,)), chain_link((0.25, 0.0, 0.0), Z, (3,)), chain_link((0.3, 0.0, 0.0), Z, (4,)), chain_link((0.3, 0.0, 0.0))]],    # front, :973-978
    # costs: 0.1 * body angle (:992-1001); moments mu1 (yr - yf) + (xr - 0.15) + (xf - 0.15) mu1 / mu2 with mu1 = 1, mu2 = 2 (:1035-1067)
    cost_rows=[chain_row(0.0, lin={0: 0.1}), chain_row(-0.15 - 0.15 * 0.5, terms=[(0, 1.0, 1.0, 0.0), (1, 0.5, -1.0, 0.0)])],
    eq_rows=[chain_row(0.0, terms=[(0, 0, 1, 0)]), chain_row(-0.05, terms=[(1, 0, 1, 0)])],                  # feet on the floor, :1005-1031
    inequalities=[(2, 1.0, 0.0), (2, -1.0, math.pi)],                                                        # knee of the rear leg, :1073-1074
    guesses=[(math.pi / 6, -math.pi / 2, math.pi / 6, -math.pi / 2, math.pi / 4), (-math.pi / 4, -math.pi / 4, math.pi / 6, -math.pi / 3, -math.pi / 4),
             (-math.pi / 3, -math.pi / 2, 0.001, -math.pi / 2, 0.0)],                                        # :1105-1113
    params=dict(max_iterations=100, max_qp_iterations=5, relative_exit_tol=1e-12, absolute_first_derivative_tol=1e-10, absolute_exit_tol=1e-8,
                termination_kkt_tolerance=1e-6, max_line_search_iterations=5, line_search_strategy=0, lambda_failure_init=0.01,
                armijo_search_tau=0.5, lambda_initial=0.001, min_lambda=1e-9))                               # :1087-1102


def _chains_np(spec):
    from oracle import chain_oracle as CH
    out = []
    for links in spec["chains"]:
        chain = CH.ActuatorChain([CH.ActuatorLink(CH.Pose(CH.so3_from_euler_xyz(l["rotation_xyz"])[0], np.array(l["translation"], float)), l["mask"])
                                  for l in links])
        out.append((chain, [i for l in links for i in l["params"]]))
    return out


def chain_rows_np(spec, rows):
    """numpy residual function (x, want_J) -> (r, J) of `rows` of a chain problem, on the oracle's ActuatorChain."""
    chains = _chains_np(spec)
    n = spec["n"]

    def fn(x, want_J):
        x = np.asarray(x, float)
        for chain, idx in chains:
            chain.update(x[idx])
        r = np.zeros(len(rows))
        J = np.zeros((len(rows), n)) if want_J else None
        for q, row in enumerate(rows):
            r[q] = row["const"] + sum(cf * x[i] for i, cf in row["lin"].items())
            if want_J:
                for i, cf in row["lin"].items():
                    J[q, i] += cf
            for c, wx, wy, wz in row["terms"]:
                chain, idx = chains[c]
                w = np.array([wx, wy, wz])
                r[q] += float(w @ chain.translation())
                if want_J:
                    np.add.at(J[q], idx, w @ chain.translation_D_params)
        return r, J
    return fn


def chain_effector_np(spec, chain_id, x):
    chain, idx = _chains_np(spec)[chain_id]
    chain.update(np.asarray(x, float)[idx])
    return chain.translation()


def mod_pi_retraction_np(x, dx, alpha):
    """The reference's custom Retraction on these problems (nonlinear_test.cc:874-880, 1077-1084): angles wrapped to [-pi, pi)."""
    from oracle import chain_oracle as CH
    return np.array([CH.mod_pi(v) for v in (x + dx * alpha)])
