#!/usr/bin/env python3
"""cProfile of the two cold API calls on a bench workload (run on the GPU box): tools/profile_api.py [workload]"""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from graphrole_amd import RecursiveFeatureExtractor, RoleExtractor

name = sys.argv[1] if len(sys.argv) > 1 else 'dw5m'
G = bench.build_graph(name)
torch.cuda.synchronize()
# warm the process once on a tiny graph (library load, staging buffers), then the cold path of the real graph
from graphrole_amd import synth
RecursiveFeatureExtractor(synth.ba_graph(2000, 4, seed=1), max_generations=2).extract_features()
pr = cProfile.Profile(); pr.enable()
t0 = time.perf_counter()
fe = RecursiveFeatureExtractor(G, max_generations=4, attributes=bool(G.attributes))
X = fe.extract_features()
t1 = time.perf_counter()
np.random.seed(0)
rx = RoleExtractor(n_roles=6); rx.extract_role_factors(X)
t2 = time.perf_counter()
pr.disable()
print(f'extract_features {t1 - t0:.3f} s   extract_role_factors {t2 - t1:.3f} s   table {X.shape}')
pstats.Stats(pr).sort_stats('cumulative').print_stats(45)
