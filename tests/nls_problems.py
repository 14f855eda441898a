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
