"""Timing of the general ego-net path (grx_egonet_features) on the config-5 graph (during round 5 one process per kernel
variant, selected by an environment switch that is gone with the variants; `variant` below is a free label from
GRX_EGO_VARIANT).  Prints one JSON line: ms per call (HIP events, best / median of 7), checksums."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from graphrole_amd import synth, kernels as K
from graphrole_amd.graph.interface import get_interface

name = sys.argv[1] if len(sys.argv) > 1 else 'dw5m'
d = f'/dev/shm/grx_ab_{name}'
if not os.path.exists(os.path.join(d, 'meta.json')):
    import bench
    G = bench.generate_graph(name)
    os.makedirs(d + '.tmp', exist_ok=True)
    synth.save_graph(G, d + '.tmp')
    os.rename(d + '.tmp', d)
G = synth.load_graph(d)
adapter = get_interface(G)(G)
host, dev, _ = adapter._device_graph()
rowsum = K.row_sums(dev, False) if dev.w is not None else None
times = []
for rep in range(8):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    internal, external = K.egonet_features_general(dev, host.directed, rowsum)
    e1.record()
    torch.cuda.synchronize()
    times.append(e0.elapsed_time(e1))
times = times[1:]
print(json.dumps({'variant': os.environ.get('GRX_EGO_VARIANT', 'default'), 'workload': name, 'best_ms': min(times),
                  'median_ms': sorted(times)[len(times) // 2], 'internal_sum': float(internal.sum().item()),
                  'external_sum': float(external.sum().item())}))
