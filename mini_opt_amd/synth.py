"""Synthetic batched QP / Newton-step workloads.

Host (numpy) generator: used for parity tests and for the committed fixtures.  Counter-based (Philox keyed by
(seed, config)) so the same problems can be regenerated anywhere.  The device generator used by bench.py
(`make_batch_torch`) follows the same construction with torch's generator on the GPU.

Construction per problem (mirrors the spirit of the reference's GenerateRandomQP, test/qp_test.cc:483-524, but
with a dense least-squares Hessian as produced by LinearizeAndFillQP, source/nonlinear.cc:170-214):
  J (m_r x n) ~ U(-1,1) stacked ROW-MAJOR, r ~ U(-1,1), lambda = 1e-3          -> G = J^T J + lambda I, c = J^T r
  A_eq (k x n) ~ U(-1,1) column-major, b_eq ~ U(-1,1)
  m/2 distinct variables get a lower AND an upper bound: (v,+1,-l), (v,-1,u), l~U(-2,-.5), u~U(.5,2)
  (two entries on one variable exercise the duplicate-index accumulation of qp.cc:296, :340-341)
  state: x ~ U(-.4,.4), s = (a x_v + b) * U(.5,1.5) > 0, z ~ U(.1,2), y ~ U(-1,1); mu = 0.1 s.z / m
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

# (n, k, m, m_r) of BASELINE.json's configs
CONFIGS = {
    "cfg1": dict(n=8, k=2, m=4, m_r=16, dtype="f64", batch=1),
    "cfg2": dict(n=32, k=4, m=16, m_r=64, dtype="f64", batch=4096),
    "cfg3": dict(n=64, k=8, m=32, m_r=128, dtype="f64", batch=65536),
    "cfg4": dict(n=128, k=16, m=64, m_r=256, dtype="f32", batch=65536),
}

LAMBDA = 1.0e-3
SEED = 0x6D696E69


@dataclass
class Batch:
    """Contiguous per-problem slabs. Layouts: J [B][m_r][n] (row-major), A_eq [B][n][k] (= k x n column-major),
    vars [B][V] in [x|s|y|z] order."""
    n: int
    k: int
    m: int
    m_r: int
    J: np.ndarray
    r: np.ndarray
    lam: float
    A_eq: np.ndarray
    b_eq: np.ndarray
    cons_var: np.ndarray
    cons_a: np.ndarray
    cons_b: np.ndarray
    vars: np.ndarray
    mu: np.ndarray

    @property
    def batch(self):
        return self.vars.shape[0]

    @property
    def V(self):
        return self.n + 2 * self.m + self.k


def make_batch(n: int, k: int, m: int, m_r: int, batch: int, seed: int = SEED, stream: int = 0) -> Batch:
    assert m % 2 == 0 and m // 2 <= n
    rng = np.random.Generator(np.random.Philox(key=[seed, (n << 40) ^ (k << 30) ^ (m << 20) ^ (m_r << 8) ^ stream]))
    U = lambda lo, hi, *shape: rng.uniform(lo, hi, size=shape)
    J = U(-1, 1, batch, m_r, n)
    r = U(-1, 1, batch, m_r)
    A_eq = U(-1, 1, batch, n, k)
    b_eq = U(-1, 1, batch, k)
    h = m // 2
    cons_var = np.zeros((batch, m), dtype=np.int32)
    cons_a = np.zeros((batch, m))
    cons_b = np.zeros((batch, m))
    for p in range(batch):
        v = rng.permutation(n)[:h]
        lo = rng.uniform(-2.0, -0.5, size=h)
        hi = rng.uniform(0.5, 2.0, size=h)
        var = np.concatenate([v, v])
        a = np.concatenate([np.ones(h), -np.ones(h)])
        b = np.concatenate([-lo, hi])
        perm = rng.permutation(m)
        cons_var[p], cons_a[p], cons_b[p] = var[perm], a[perm], b[perm]
    x = U(-0.4, 0.4, batch, n)
    xv = np.take_along_axis(x, cons_var.astype(np.int64), axis=1)
    s = (cons_a * xv + cons_b) * U(0.5, 1.5, batch, m)
    z = U(0.1, 2.0, batch, m)
    y = U(-1, 1, batch, k)
    vars_ = np.concatenate([x, s, y, z], axis=1)
    mu = 0.1 * np.sum(s * z, axis=1) / max(m, 1)
    return Batch(n, k, m, m_r, J, r, LAMBDA, A_eq, b_eq, cons_var, cons_a, cons_b, vars_, mu)


def algorithmic_bytes(n: int, k: int, m: int, m_r: int, T: int) -> int:
    """J-level algorithmic bytes per Newton step (in + out)."""
    V = n + 2 * m + k
    return T * (m_r * n + m_r + k * n + k + V + 1) + m * (4 + 2 * T) + T * (V + 2) + 4


def algorithmic_flops(n: int, k: int, m: int, m_r: int) -> float:
    P = n + k
    return m_r * n * (n + 1) + 2 * m_r * n + 2 * n * n + 4 * k * n + 6 * m + P ** 3 / 3 + 2 * P * P + 8 * m


def make_batch_torch(n: int, k: int, m: int, m_r: int, batch: int, device, dtype, seed: int = SEED):
    """Device-side generator with the same construction as make_batch (different random stream).
    Returns (BatchedQP, vars [B,V], mu [B]) resident on `device`."""
    import torch

    from .qp import BatchedQP
    g = torch.Generator(device=device)
    g.manual_seed(seed)

    def U(lo, hi, *shape):
        return torch.rand(*shape, generator=g, device=device, dtype=torch.float64) * (hi - lo) + lo

    h = m // 2
    if h > n or (m & 1):  # a lower and an upper bound on each of m / 2 DISTINCT variables; checked here, on the host, not by an out-of-range gather on the device
        raise ValueError(f"synthetic batches need an even m <= 2 n (got n = {n}, m = {m})")
    J = torch.empty(batch, m_r, n, device=device, dtype=dtype)
    step = max(1, (1 << 27) // max(1, m_r * n))  # generate in slices to bound the fp64 temporary
    for b0 in range(0, batch, step):
        b1 = min(batch, b0 + step)
        J[b0:b1] = U(-1, 1, b1 - b0, m_r, n).to(dtype)
    r = U(-1, 1, batch, m_r).to(dtype)
    A_eq = U(-1, 1, batch, n, k).to(dtype)
    b_eq = U(-1, 1, batch, k).to(dtype)
    v = torch.argsort(torch.rand(batch, n, generator=g, device=device), dim=1)[:, :h]
    lo = U(-2.0, -0.5, batch, h)
    hi = U(0.5, 2.0, batch, h)
    var = torch.cat([v, v], dim=1)
    a = torch.cat([torch.ones(batch, h, device=device, dtype=torch.float64), -torch.ones(batch, h, device=device, dtype=torch.float64)], dim=1)
    b = torch.cat([-lo, hi], dim=1)
    perm = torch.argsort(torch.rand(batch, m, generator=g, device=device), dim=1)
    cons_var = torch.gather(var, 1, perm).to(torch.int32).contiguous()
    cons_a = torch.gather(a, 1, perm).to(dtype).contiguous()
    cons_b = torch.gather(b, 1, perm).to(dtype).contiguous()
    x = U(-0.4, 0.4, batch, n).to(dtype)
    xv = torch.gather(x.double(), 1, cons_var.long())
    s = ((cons_a.double() * xv + cons_b.double()) * U(0.5, 1.5, batch, m)).to(dtype)
    z = U(0.1, 2.0, batch, m).to(dtype)
    y = U(-1, 1, batch, k).to(dtype)
    vars_ = torch.cat([x, s, y, z], dim=1).contiguous()
    mu = (0.1 * torch.sum(s.double() * z.double(), dim=1) / max(m, 1)).to(dtype).contiguous()
    qp = BatchedQP(n=n, k=k, m=m, J=J, r=r.contiguous(), lam=LAMBDA, A_eq=A_eq.contiguous(), b_eq=b_eq.contiguous(),
                   cons_var=cons_var, cons_a=cons_a, cons_b=cons_b)
    return qp, vars_, mu
