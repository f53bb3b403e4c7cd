#!/usr/bin/env python3
"""Wall-clock of the stages of the RolX NMF on the bench workload (run on the GPU box)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from scipy import linalg
from graphrole_amd import RecursiveFeatureExtractor, synth, kernels as K
from graphrole_amd.roles import factor

G = synth.ba_graph(1_000_000, 10, seed=0)
fe = RecursiveFeatureExtractor(G, max_generations=4)
fe.run_on_device()
names, cols = fe.device_features()
n, r = G.n, 6
rng = np.random.RandomState(0)

def sync():
    torch.cuda.synchronize()

for trial in range(3):
    t = {}
    sync(); t0 = time.perf_counter()
    Xd = K.gather_columns(cols, n); sync(); t['gather'] = time.perf_counter() - t0
    omega = rng.normal(size=(len(names), r + 10))
    t0 = time.perf_counter()
    G1, xsum = K.gram(Xd, n); t['gram1 (+D2H)'] = time.perf_counter() - t0
    t0 = time.perf_counter()
    lam, V1 = linalg.eigh(G1); t['eigh1'] = time.perf_counter() - t0
    t0 = time.perf_counter()
    W0, H0, xx = factor.nndsvda_init_device(Xd, n, r, omega); sync(); t['init total'] = time.perf_counter() - t0
    t0 = time.perf_counter()
    st = K.NmfState(Xd, n, W0, H0, x_sq_norm=xx); sync(); t['state'] = time.perf_counter() - t0
    t0 = time.perf_counter()
    state, it = factor.run_mu_loop(st); sync(); t['mu loop'] = time.perf_counter() - t0
    t0 = time.perf_counter()
    st.iterate(10, with_residual=True); sync(); t['iterate(10)+res'] = time.perf_counter() - t0
    print(trial, it, {k: round(v * 1e3, 3) for k, v in t.items()})

import cProfile, pstats
pr = cProfile.Profile()
sync(); pr.enable()
for _ in range(20):
    W0, H0, xx = factor.nndsvda_init_device(Xd, n, r, omega)
sync(); pr.disable()
pstats.Stats(pr).sort_stats('tottime').print_stats(18)
