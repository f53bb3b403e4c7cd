#!/usr/bin/env python3
"""Diagnostic (GPU box): k-means++ seeds of grx_kmeans1d (max_iter = 0) against the oracle's, on dumped inputs
(tools/diag_in/<table>_<roles>_<bits>_<G|F>.npy)."""
import glob, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graphrole_amd import kernels as K
from oracle import kmeans1d

for path in sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'diag_in', '*.npy'))):
    tag = os.path.basename(path)[:-4]
    k = 2 ** int(tag.split('_')[-2])
    v = np.load(path)
    m = len(v)
    x = v - v.mean()
    xsq = x * x
    w = np.ones(m)
    q, c, info = K.kmeans1d(K.to_device(v), k, max_iter=0)
    c = K.to_host(c)
    rs = np.random.RandomState(1)
    trials = 2 + int(np.log(k))
    picked = [rs.choice(m, p=w / w.sum())]
    d = kmeans1d._sq_dist(x[picked[0]], x, xsq)
    pot = d @ w
    ok = True
    for step in range(1, k):
        U = rs.uniform(size=trials)
        rand_vals = U * pot
        C = np.cumsum(w * d)
        cand = np.searchsorted(C, rand_vals)
        np.clip(cand, None, m - 1, out=cand)
        D = np.stack([kmeans1d._sq_dist(x[j], x, xsq) for j in cand])
        np.minimum(d, D, out=D)
        pots = D @ w
        best = int(np.argmin(pots))
        if abs(c[step] - v[cand[best]]) > 1e-9 * np.abs(v).max():
            gi = np.nonzero(np.abs(v - c[step]) < 1e-12 * np.abs(v).max())[0]
            print(tag, 'm', m, 'k', k, 'step', step, 'oracle cand', cand.tolist(), 'best', best, 'gpu picked index', gi.tolist())
            print('   pots', [repr(float(p)) for p in pots])
            print('   rand_vals', rand_vals.tolist(), 'pot', repr(float(pot)), 'C[-1]', repr(float(C[-1])))
            for g in gi:
                print('   C around gpu index', g, [repr(float(z)) for z in C[max(g - 1, 0):g + 1]])
            ok = False
            break
        pot, d = pots[best], D[best]
        picked.append(cand[best])
    if ok:
        print(tag, 'seeds equal')
