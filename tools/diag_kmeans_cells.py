#!/usr/bin/env python3
"""Diagnostic (GPU box): per model-selection cell, grx_kmeans1d against sklearn KMeans on the same factor."""
import json, os, sys, warnings
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
warnings.simplefilter('ignore')
from sklearn.cluster import KMeans
from graphrole_amd import kernels as K
from graphrole_amd.roles import factor
from tests import util

name = sys.argv[1] if len(sys.argv) > 1 else 'dw200_attrs'
g = util.load_refex(name)
V = np.ascontiguousarray(g['final_values'], dtype=np.float64)
Vd = K.to_device(np.ascontiguousarray(V.T))
np.random.seed(0)
n, F = V.shape
for roles in range(2, min(min(V.shape), 8) + 1):
    for bits in range(1, 9):
        k = 2 ** bits
        st_rng = np.random.get_state()
        try:
            state, _ = factor.nmf_state(Vd, V, roles)
        except ValueError:
            continue
        if k > roles * n or k > roles * F:
            continue
        W = K.to_host(state.W)[:, :n].T.copy()
        H = K.to_host(state.H).copy()
        for label, M in (('G', W), ('F', H)):
            flat = M.reshape(-1)
            km = KMeans(n_clusters=k, random_state=1).fit(flat.reshape(-1, 1))
            ref = km.cluster_centers_[km.labels_].ravel()
            q, c, info = K.kmeans1d(K.to_device(flat), k)
            q, info = K.to_host(q), K.to_host(info)
            d = np.abs(q - ref).max() / max(np.abs(flat).max(), 1e-300)
            if d > 1e-9 or int(info[0]) != km.n_iter_:
                os.makedirs('gpurun_out/diag', exist_ok=True)
                np.save(f'gpurun_out/diag/{name}_{roles}_{bits}_{label}.npy', flat)
                np.save(f'gpurun_out/diag/{name}_{roles}_{bits}_{label}_q.npy', q)
                np.save(f'gpurun_out/diag/{name}_{roles}_{bits}_{label}_c.npy', K.to_host(c))
                print(f'roles={roles} bits={bits} {label} m={flat.size} k={k} distinct_values={len(np.unique(flat))} '
                      f'n_iter ours={int(info[0])} sklearn={km.n_iter_} maxdiff={d:.3e} '
                      f'uniq ours={len(np.unique(q))} ref={len(np.unique(ref))}')
print('done')
