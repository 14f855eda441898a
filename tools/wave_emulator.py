#!/usr/bin/env python3
"""Lane-level NumPy emulation of the fused single-wave Newton-step kernel (mini_opt_amd/csrc/kkt_fused.hip).

Development aid only: it mirrors the kernel's data layout (16x16 tiles in the f64 MFMA C/D fragment layout, the
column permutation induced by 16-byte J loads, the symmetric sweep of diagonal tiles, the augmented right-hand-side
column, the backward substitution) so that index maps and masks can be validated on the CPU against a plain numpy
solve before the HIP version is run on a GPU.  Not part of the product, not used by tests.
"""
import numpy as np

LANES = 64
g_of = np.arange(LANES) >> 4
j_of = np.arange(LANES) & 15


def mfma(a, b, c):
    """v_mfma_f64_16x16x4_f64: a, b are [64] lane vectors, c is [4][64] (C/D layout). D = A B + C with
    A[i][k] = a[16k + i], B[k][j] = b[16k + j], D[row = g + 4 t][col = j] at lane (g, j) reg t."""
    A = a.reshape(4, 16).T          # [i][k]
    B = b.reshape(4, 16)            # [k][j]
    D = A @ B                       # [i][j]
    out = c.copy()
    for t in range(4):
        for lane in range(LANES):
            out[t, lane] += D[g_of[lane] + 4 * t, j_of[lane]]
    return out


def tile_to_dense(tile):
    M = np.zeros((16, 16))
    for t in range(4):
        for lane in range(LANES):
            M[g_of[lane] + 4 * t, j_of[lane]] = tile[t, lane]
    return M


def dense_to_tile(M):
    tile = np.zeros((4, LANES))
    for t in range(4):
        tile[t] = M[g_of + 4 * t, j_of]
    return tile


def sigma(c, i, NT):
    """original column of permuted position (block c, index i): 16-byte loads give each lane 2 adjacent columns"""
    return 32 * (c >> 1) + 2 * i + (c & 1)


def sweep_tile(tile, npiv):
    """Symmetric sweep of pivots 0..npiv-1 of a (symmetric) tile in C layout, using only lane-level operations the kernel
    has: readlane (pivot), bpermute (row k to all lane groups, and its R-layout), select masks.  Returns (-inverse on the
    swept block, A11^-1 A12 on swept x unswept), ok flag."""
    tile = tile.copy()
    ok = True
    for k in range(npiv):
        src_g, src_t = k & 3, k >> 2
        d = tile[src_t, 16 * src_g + k]                      # readlane
        if d == 0 or not np.isfinite(d):
            ok = False
            d = 1.0
        inv = 1.0 / d
        rowreg = tile[src_t].copy()                           # all broadcasts are taken BEFORE any register is updated
        rowk_v16 = rowreg[16 * src_g + j_of]                  # bpermute: lane (g,j) <- lane (k&3, j)
        rk = rowk_v16 * inv
        rk = np.where(j_of == k, -inv, rk)
        for t in range(4):
            r_idx = g_of + 4 * t
            f = rowreg[16 * src_g + r_idx]                    # bpermute: row k at column g+4t (symmetry: A(r,k) = A(k,r))
            a = np.where(j_of == k, 0.0, tile[t])             # zero column k
            new = a - f * rk
            if t == src_t:
                new = np.where(g_of == src_g, rk, new)        # row k
            tile[t] = new
    return tile, ok


def run(n, k, m, m_r, J, r, lam, A_eq, b_eq, cons_var, cons_a, cons_b, vars_, mu):
    NT = n // 16
    RC = 15
    assert k <= 14 and n % 32 == 0
    x = vars_[:n]; s = vars_[n:n + m]; y = vars_[n + m:n + m + k]; z = vars_[n + m + k:]
    # P1: J^T J tiles (upper block triangle) + c, in permuted space
    perm = np.array([sigma(c, i, NT) for c in range(NT) for i in range(16)])
    U = {}
    for a in range(NT):
        for b in range(a, NT):
            U[(a, b)] = np.zeros((4, LANES))
    cpart = np.zeros((NT, LANES))
    for q0 in range(0, m_r, 4):
        ops = []
        for c in range(NT):
            ops.append(J[q0 + g_of, perm[16 * c + j_of]])      # lane (g,i) holds J(q0+g, sigma(c,i))
        rq = r[q0 + g_of]
        for a in range(NT):
            cpart[a] += ops[a] * rq
            for b in range(a, NT):
                U[(a, b)] = mfma(ops[a], ops[b], U[(a, b)])
    cvec = np.zeros((NT, LANES))
    for c in range(NT):
        tot = cpart[c].reshape(4, 16).sum(axis=0)               # reduce over g
        cvec[c] = tot[j_of]
    # P3: constraints -> per-variable Sigma and rhs (natural order), then permuted V16
    diagS = np.zeros(n); rhsS = np.zeros(n)
    status = 0
    for c in range(m):
        if not s[c] > 0:
            status = 1
        v = cons_var[c]
        diagS[v] += cons_a[c] * (z[c] / s[c]) * cons_a[c]
        rhsS[v] += cons_a[c] * (z[c] * (s[c] - cons_b[c]) + mu) / s[c]
    if status:
        return None, None, status
    rhs_perm = np.zeros(n)
    for c in range(NT):
        for i in range(16):
            rhs_perm[16 * c + i] = -cvec[c, i] + rhsS[sigma(c, i, NT)]
    # P2: lambda + Sigma on the diagonal
    for a in range(NT):
        dg = np.array([diagS[sigma(a, i, NT)] for i in range(16)])
        for t in range(4):
            mask = (j_of == g_of + 4 * t)
            U[(a, a)][t] += np.where(mask, lam + dg[j_of], 0.0)
    # P4: y tile column [A_eq^T | rhs] and the y diagonal tile
    for a in range(NT):
        M = np.zeros((16, 16))
        for rr in range(16):
            for q in range(k):
                M[rr, q] = A_eq[q, sigma(a, rr, NT)]
            M[rr, RC] = rhs_perm[16 * a + rr]
        U[(a, NT)] = dense_to_tile(M)
    M = np.zeros((16, 16))
    for q in range(k):
        M[q, RC] = -b_eq[q]
        M[RC, q] = -b_eq[q]
    U[(NT, NT)] = dense_to_tile(M)
    # P5: block elimination
    ok = True
    for a in range(NT + 1):
        U[(a, a)], okk = sweep_tile(U[(a, a)], 16 if a < NT else k)
        ok &= okk
        negTinv = U[(a, a)]
        for c in range(a + 1, NT + 1):
            negZ = np.zeros((4, LANES))
            for st in range(4):
                negZ = mfma(negTinv[st], U[(a, c)][st], negZ)
            for b in range(a + 1, c + 1):
                for st in range(4):
                    U[(b, c)] = mfma(U[(a, b)][st], negZ[st], U[(b, c)])
    if not ok:
        return None, None, 2
    # P6: backward
    xb = np.zeros((NT + 1, LANES))  # V16 layout per block
    Ty = tile_to_dense(U[(NT, NT)])
    nu = np.zeros(16)
    nu[:k] = Ty[:k, RC]
    xb[NT] = nu[j_of]
    for a in range(NT - 1, -1, -1):
        p = np.zeros((4, LANES))
        for b in range(a + 1, NT + 1):
            for t in range(4):
                p[t] += U[(a, b)][t] * xb[b]
        v = np.zeros((4, LANES))
        for t in range(4):
            tot = p[t].reshape(4, 16).sum(axis=1)               # reduce over j within each lane group
            ta = U[(a, NT)][t].reshape(4, 16)[:, RC]           # row-broadcast of column RC
            v[t] = (ta - tot)[g_of]
        q = np.zeros(LANES)
        for t in range(4):
            q += U[(a, a)][t] * v[t]                            # (-Tinv) * v
        tot = q.reshape(4, 16).sum(axis=0)                      # reduce over g
        xb[a] = -tot[j_of]
    # P7: epilogue
    xplus = np.zeros(n)
    for c in range(NT):
        for i in range(16):
            xplus[sigma(c, i, NT)] = xb[c, i]
    dx = xplus - x
    dy = -nu[:k] - y
    ds = np.zeros(m); dz = np.zeros(m)
    for c in range(m):
        r_pi = cons_a[c] * x[cons_var[c]] + cons_b[c] - s[c]
        ds[c] = cons_a[c] * dx[cons_var[c]] + r_pi
        dz[c] = -(z[c] / s[c]) * ds[c] - (1 / s[c]) * (s[c] * z[c] - mu)
    return np.concatenate([dx, ds, dy, dz]), None, 0


if __name__ == "__main__":
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
    z = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "synthetic.npz"))
    for cfg in ("cfg2", "cfg3"):
        g = lambda key: z[f"{cfg}_{key}"]
        J = g("J")
        B, m_r, n = J.shape
        k, m = g("b_eq").shape[1], g("cons_var").shape[1]
        for p in range(min(B, 2)):
            d, _, st = run(n, k, m, m_r, J[p], g("r")[p], float(g("lam")), g("A_eq")[p].T, g("b_eq")[p], g("cons_var")[p],
                           g("cons_a")[p], g("cons_b")[p], g("vars")[p], float(g("mu")[p]))
            ref = g("delta")[p]
            print(cfg, p, "status", st, "rel-inf err", np.max(np.abs(d - ref)) / np.max(np.abs(ref)))
