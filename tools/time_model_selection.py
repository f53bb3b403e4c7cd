#!/usr/bin/env python3
"""Wall-clock of RoleExtractor(n_roles=None) -- the reference's default: MDL model selection over
2..8 roles x 1..8 bits -- on the bench graph's feature table (run on the GPU box)."""
import sys, time, os, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa: F401
from graphrole_amd import RecursiveFeatureExtractor, RoleExtractor, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
G = synth.ba_graph(n, 10, seed=0)
fe = RecursiveFeatureExtractor(G, max_generations=4)
X = fe.extract_features()
for trial in range(2):
    np.random.seed(0)
    t0 = time.perf_counter()
    re_ = RoleExtractor()
    re_.extract_role_factors(X)
    torch.cuda.synchronize()
    print(f'trial {trial}: model selection {1e3 * (time.perf_counter() - t0):.1f} ms -> {re_.node_role_factor.shape[1]} roles')
pr = cProfile.Profile(); pr.enable()
np.random.seed(0); RoleExtractor().extract_role_factors(X); torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(22)
