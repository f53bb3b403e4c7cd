#!/usr/bin/env python3
"""cProfile of the host side of bench.py's step (ReFeX loop + NMF) on a bench workload: where the time between kernels goes."""
import cProfile, pstats, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from graphrole_amd import RecursiveFeatureExtractor, kernels as K
from graphrole_amd.roles import factor
name = sys.argv[1] if len(sys.argv) > 1 else 'ba1m'
G = bench.build_graph(name)
fe = RecursiveFeatureExtractor(G, max_generations=4, attributes=bool(G.attributes))
rng = np.random.RandomState(0)
def step():
    fe.reset()
    fe.run_on_device()
    names, cols = fe.device_features()
    Xd = K.gather_columns(cols, G.n)
    omega = rng.normal(size=(len(names), bench.N_ROLES + 10))
    factor.nmf_device(Xd, G.n, bench.N_ROLES, omega, plan=fe._shard())
    torch.cuda.synchronize()
for _ in range(5): step()
pr = cProfile.Profile(); pr.enable()
for _ in range(20): step()
pr.disable()
pstats.Stats(pr).sort_stats('tottime').print_stats(30)
