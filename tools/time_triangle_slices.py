"""Where does generation 0 of an unweighted graph spend its time when it is cut into P shares?  Times, for the bench's
graph, grx_triangle_count over every share of DeviceCSR.triangle_split and egonet_from_triangles over every row range of
the ShardPlan cut.  python tools/time_triangle_slices.py [workload] [P]"""
import json
import sys

import numpy as np
import torch

sys.path.insert(0, __file__.rsplit('/', 2)[0])
sys.path.insert(0, __file__.rsplit('/', 1)[0])
import project_scaling as PS  # noqa: E402
from graphrole_amd import kernels as K  # noqa: E402
from graphrole_amd.features.extract import RecursiveFeatureExtractor  # noqa: E402


def main():
    workload = sys.argv[1] if len(sys.argv) > 1 else 'ba1m'
    world = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    G = PS.build(workload)
    fe = RecursiveFeatureExtractor(G, max_generations=PS.MAX_GENERATIONS)
    fe.run_on_device()
    torch.cuda.synchronize()
    host, dev, _ = fe.graph._device_graph()
    o = dev.oriented()
    dplus = np.diff(o._host[0])
    out = {'workload': workload, 'world': world, 'n': int(host.n)}
    out['triangles_all_ms'] = PS.timed(lambda: K.triangle_counts(dev, 0, host.n))
    T = K.triangle_counts(dev, 0, host.n)
    out['egonet_all_ms'] = PS.timed(lambda: K.egonet_from_triangles(dev, T, 0, host.n))
    bounds = PS.cuts_work(host.row_ptr, world)
    shares = []
    for r in range(world):
        a, b = dev.triangle_split(r, world)
        t_tri = PS.timed(lambda: K.triangle_counts(dev, a, b))
        Ts = K.to_host(K.triangle_counts(dev, a, b))
        rb, re = int(bounds[r]), int(bounds[r + 1])
        t_ego = PS.timed(lambda: K.egonet_from_triangles(dev, T, rb, re))
        shares.append({'rank': r, 'tri_rows': [int(a), int(b)], 'tri_arcs': int(dplus[a:b].sum()),
                       'tri_cost_model': int((dplus[a:b] * (dplus[a:b] + 1) + 1).sum()), 'corner_counts_added': int(Ts.sum()),
                       'max_dplus': int(dplus[a:b].max()) if b > a else 0,
                       'triangles_ms': round(t_tri, 4), 'ego_rows': [rb, re], 'egonet_ms': round(t_ego, 4)})
        print(json.dumps(shares[-1]), flush=True)
    out['shares'] = shares
    print(json.dumps({k: v for k, v in out.items() if k != 'shares'}))


if __name__ == '__main__':
    main()
