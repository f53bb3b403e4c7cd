"""config 5 at full size: the NMF of the 5 M x 115 table against the oracle, with and without the column equilibration
of grx_host_whiten (GRX_WHITEN_NO_SCALE=1).  Usage: PYTHONPATH=. python tools/diag_config5_nmf.py [n] [m]"""
import os
import sys
import time

import numpy as np

from graphrole_amd import RecursiveFeatureExtractor, kernels as K, synth
from graphrole_amd.roles import factor
from oracle import rolx

n = int(sys.argv[1]) if len(sys.argv) > 1 else 5_000_000
m = int(sys.argv[2]) if len(sys.argv) > 2 else 100_000_000
G = synth.directed_weighted_graph(n, m, seed=0)
X = RecursiveFeatureExtractor(G, max_generations=4, attributes=True).extract_features().values.astype(float)
print('table', X.shape, 'column norms 10^', np.round(np.log10(np.sqrt((X * X).sum(0)).min()), 1), '..',
      np.round(np.log10(np.sqrt((X * X).sum(0)).max()), 1))
G1 = X.T @ X
T1, lam, V = K.host_whiten(G1)
print('kept directions', T1.shape[1], 'of', X.shape[1], 'lam range', lam.min(), lam.max(), 'no_scale =', os.environ.get('GRX_WHITEN_NO_SCALE'))
np.random.seed(0)
t0 = time.time()
Gf, Ff, n_iter = factor.nmf_with_info(X, 6)
np.random.seed(0)
We, He, it = rolx.nmf(X, 6)
print('n_iter', n_iter, it, 'W rel', np.abs(Gf - We).max() / np.abs(We).max(), 'H rel', np.abs(Ff - He).max() / np.abs(He).max(),
      f'({time.time() - t0:.0f} s)')
omega = np.random.RandomState(1).normal(size=(X.shape[1], 16))
W0, H0 = rolx.nndsvda_init(X, 6, omega)
Xd = K.to_device(np.ascontiguousarray(X.T))
W0d, H0d, _ = factor.nndsvda_init_device(Xd, X.shape[0], 6, omega)
W0h = K.to_host(W0d)[:, :X.shape[0]].T
H0h = K.to_host(H0d) if hasattr(H0d, 'is_cuda') else np.asarray(H0d)
print('init W0 rel', np.abs(W0h - W0).max() / np.abs(W0).max(), 'H0 rel', np.abs(H0h - H0).max() / np.abs(H0).max())
