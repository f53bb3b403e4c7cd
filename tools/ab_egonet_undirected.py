"""Timing of the general ego-net path on an UNDIRECTED weighted graph (BA 1 M / 10 M with weights): every member's row
holds the arc back to v, the case the undirected fast path of the group kernel is for."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from graphrole_amd import synth, kernels as K
from graphrole_amd.graph.csr import CSRGraph
from graphrole_amd.graph.interface import get_interface
src, dst = synth.ba_edges(1_000_000, 10, seed=0)
w = np.random.default_rng(0).uniform(0.1, 5.0, size=len(src))
G = CSRGraph(1_000_000, src, dst, weights=w, validate=False)
adapter = get_interface(G)(G)
host, dev, _ = adapter._device_graph()
rowsum = K.row_sums(dev, False)
times = []
for rep in range(8):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    internal, external = K.egonet_features_general(dev, False, rowsum)
    e1.record()
    torch.cuda.synchronize()
    times.append(e0.elapsed_time(e1))
print(json.dumps({'workload': 'BA 1 M / 10 M undirected, weights U(0.1, 5)', 'best_ms': min(times[1:]), 'median_ms': sorted(times[1:])[3],
                  'internal_sum': float(internal.sum().item()), 'external_sum': float(external.sum().item())}))
