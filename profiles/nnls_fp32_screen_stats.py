"""CPU experiment (numpy, no GPU): could the NNLS dual be SCREENED in fp32 and only the winner re-checked in fp64?  Runs Lawson-Hanson
on C4-like voxels and, per outer iteration, counts how many zero-set bins lie within the fp32 error bound of the largest dual --
in residual form, w = B32^T r + R^T(...) with the basis rounded to float32.  A band of one decides exactly; anything else needs
the fp64 dual anyway.  usage: python profiles/nnls_fp32_screen_stats.py [n_voxels] [sigma]"""
import numpy as np, sys
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyneapple_amd import synth
bins, B, R = synth.nnls_matrices(32)
A = np.vstack([B, R]); G = A.T @ A
B32 = B.astype(np.float32)
nv = int(sys.argv[1]) if len(sys.argv) > 1 else 200
sigma = float(sys.argv[2]) if len(sys.argv) > 2 else 0.01
_, y, _ = synth.make_numpy("tri_reduced", nv, 32, sigma=sigma, scale=1000.0)
n = 250
bands = {k: [] for k in (21, 20, 19, 18)}
term_amb = {k: 0 for k in bands}; terms = 0; argsame = 0; outer = 0; maxerr = 0
for v in range(nv):
    aty = B.T @ y[v]
    P = []; x = np.zeros(n); inP = np.zeros(n, bool)
    while len(P) < n:
        rt = y[v] - B[:, P] @ x[P] if P else y[v].copy()
        rb = -(R[:, P] @ x[P]) if P else np.zeros(n)
        w = B.T @ rt + R.T @ rb
        # fp32 screen: B32^T r32 accumulated in fp32
        r32 = rt.astype(np.float32)
        w32 = (B32.T.astype(np.float32) @ r32).astype(np.float64) + R.T @ rb
        w[inP] = -np.inf; w32[inP] = -np.inf
        outer += 1
        e = np.abs(w32 - w)[~inP].max(); s1 = np.abs(rt).sum()
        maxerr = max(maxerr, e / (2.0**-23 * s1))
        j = int(np.argmax(w)); j32 = int(np.argmax(w32))
        if w[j] <= 0:
            terms += 1
            for k in bands:
                if w32[j32] > -(2.0**-k) * s1: term_amb[k] += 1
            break
        argsame += (j == j32)
        for k in bands:
            eps = 2.0**-k * s1
            bands[k].append(int(np.sum(w32[~inP] >= w32[j32] - 2 * eps)))
        P.append(j); inP[j] = True
        while True:
            z = np.linalg.solve(G[np.ix_(P, P)], aty[P])
            if (z > 0).all():
                x[:] = 0; x[P] = z; break
            xp = x[P]; mask = z <= 0
            alpha = np.min(xp[mask] / (xp[mask] - z[mask]))
            xp = xp + alpha * (z - xp)
            rem = [P[i] for i in range(len(P)) if not (xp[i] > 1e-14 * np.abs(xp).max())]
            x[:] = 0
            for i, b_ in enumerate(P): x[b_] = xp[i]
            for b_ in rem:
                P.remove(b_); inP[b_] = False; x[b_] = 0
print("outer", outer, "argmax same", argsame / (outer - terms), "max err / (2^-23 sum|r|)", maxerr)
for k in bands:
    bs = np.array(bands[k])
    print("eps=2^-%d*sum|r|: band==1 %.4f <=2 %.4f <=4 %.4f max %d ; ambiguous terminations %d of %d" % (k, (bs == 1).mean(), (bs <= 2).mean(), (bs <= 4).mean(), bs.max(), term_amb[k], terms))
