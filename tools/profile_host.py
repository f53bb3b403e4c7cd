#!/usr/bin/env python3
"""cProfile of the host side of one ReFeX pass (run on the GPU box): tools/profile_host.py [workload]"""
import cProfile, pstats, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from graphrole_amd import RecursiveFeatureExtractor
name = sys.argv[1] if len(sys.argv) > 1 else 'dw1m'
G = bench.build_graph(name)
fe = RecursiveFeatureExtractor(G, max_generations=4, attributes=bool(G.attributes))
fe.run_on_device(); torch.cuda.synchronize()
fe.reset()
pr = cProfile.Profile(); pr.enable()
fe.run_on_device(); torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(28)
