import numpy as np, torch, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mini_opt_amd import qp as Q
dev = torch.device("cuda:0")
T = lambda a, dt=torch.float64: torch.as_tensor(a, dtype=dt, device=dev).contiguous()
for (n, m_r) in ((160, 170), (256, 300), (130, 40)):
    rng = np.random.default_rng(1)
    B = 2
    J = rng.uniform(-1, 1, (B, m_r, n)); r = rng.uniform(-1, 1, (B, m_r))
    prob = Q.BatchedQP(n=n, J=T(J), r=T(r), lam=1e-3)
    G, c, f = Q.linearize(prob, force_generic=True)
    G = G.cpu().numpy()   # [B, n(col), n(row)] col-major: G[b, j, i] = element (i, j)
    ref = np.einsum("bqi,bqj->bij", J, J) + 1e-3 * np.eye(n)
    got = np.transpose(G, (0, 2, 1))
    L = np.tril(np.ones((n, n), bool))
    err = np.abs(got - ref) * L
    print(n, m_r, "max err lower", err.max(), "c err", np.abs(c.cpu().numpy() - np.einsum("bqi,bq->bi", J, r)).max())
    if err.max() > 1e-9:
        bad = np.argwhere(err[0] > 1e-9)
        print(" bad count", len(bad), "first", bad[:10].tolist(), "last", bad[-5:].tolist())
        tiles = sorted(set((int(i) // 16, int(j) // 16) for i, j in bad))
        print(" bad tiles", tiles[:60])
