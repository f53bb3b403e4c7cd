#!/usr/bin/env python3
"""Wall-clock of the drop-in API calls on the bench graph (run on the GPU box)."""
import sys, time, os, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401  (loads the HIP runtime)
from graphrole_amd import RecursiveFeatureExtractor, RoleExtractor, synth

t0 = time.perf_counter(); G = synth.ba_graph(1_000_000, 10, seed=0); t1 = time.perf_counter()
print(f'build CSRGraph (host): {t1 - t0:.2f} s')
for trial in range(2):
    t0 = time.perf_counter()
    fe = RecursiveFeatureExtractor(G, max_generations=4)
    X = fe.extract_features()
    t1 = time.perf_counter()
    re_ = RoleExtractor(n_roles=6)
    re_.extract_role_factors(X)
    t2 = time.perf_counter()
    roles = re_.roles
    t3 = time.perf_counter()
    print(f'trial {trial}: extract_features {1e3 * (t1 - t0):.1f} ms ({X.shape}), extract_role_factors {1e3 * (t2 - t1):.1f} ms, roles dict {1e3 * (t3 - t2):.1f} ms')
pr = cProfile.Profile(); pr.enable()
fe = RecursiveFeatureExtractor(G, max_generations=4); X = fe.extract_features()
re_ = RoleExtractor(n_roles=6); re_.extract_role_factors(X)
pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(30)
