"""Why is the cold extract_features of dw5m slower inside bench.py than alone?  Runs api_wall twice after a bench-like warm-up."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, cProfile, pstats
import bench
from graphrole_amd import RecursiveFeatureExtractor, RoleExtractor
G = bench.build_graph('dw5m')
class A: pass
for rep in range(3):
    fe = RecursiveFeatureExtractor(G, max_generations=4, attributes=True)
    torch.cuda.synchronize()
    pr = cProfile.Profile(); pr.enable()
    t0 = time.perf_counter()
    X = fe.extract_features()
    t1 = time.perf_counter()
    pr.disable()
    print('rep', rep, 'extract_features', round(t1 - t0, 3), X.shape, flush=True)
    if rep == 0 or rep == 2:
        pstats.Stats(pr).sort_stats('cumulative').print_stats(14)
    if rep == 1:
        del X          # rep 2 allocates into freed pages
