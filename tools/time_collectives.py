"""Where the time of the sharded path goes: wraps every ShardPlan exchange with synchronised timers.
Run on a one-GPU box:  GRX_FORCE_COLLECTIVES=1 python tools/time_collectives.py [workload]
(one rank, backend nccl = RCCL; the exchanges are self-exchanges, so this measures launch / staging
overhead, the floor of what N > 1 pays per generation)."""
import os
import sys
import time
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault('GRX_FORCE_COLLECTIVES', '1')
os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
os.environ.setdefault('MASTER_PORT', '29533')

import numpy as np
import torch
import torch.distributed as dist


def main():
    import bench
    from graphrole_amd import RecursiveFeatureExtractor, kernels as K, parallel
    from graphrole_amd.roles import factor
    torch.cuda.set_device(0)
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
    acc, cnt = defaultdict(float), defaultdict(int)

    def wrap(name):
        fn = getattr(parallel.ShardPlan, name)

        def timed(self, *a, **k):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            out = fn(self, *a, **k)
            torch.cuda.synchronize()
            acc[name] += time.perf_counter() - t0
            cnt[name] += 1
            return out
        setattr(parallel.ShardPlan, name, timed)
    for nm in ['all_gather_block', 'columns_to_owners', 'owned_to_rows', 'all_gather_columns', 'all_reduce_max_',
               'all_reduce_sum_', 'all_reduce_sum_host', 'all_gather_host']:
        wrap(nm)
    G = bench.build_graph(sys.argv[1] if len(sys.argv) > 1 else 'ba1m')
    fe = RecursiveFeatureExtractor(G, max_generations=4, distributed=True)
    fe.run_on_device()
    steps = 5
    acc.clear(); cnt.clear()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fe.reset()
        fe.run_on_device()
    torch.cuda.synchronize()
    refex = (time.perf_counter() - t0) / steps
    print(f'refex {refex * 1e3:.2f} ms/step (with the timers\' synchronisations)')
    for k in sorted(acc, key=acc.get, reverse=True):
        print(f'  {k:22s} {acc[k] / steps * 1e3:7.3f} ms/step  {cnt[k] / steps:5.1f} calls/step')
    names, cols = fe.device_features()
    Xd = K.gather_columns(cols, G.n)
    plan = fe._shard()
    omega = np.random.RandomState(0).normal(size=(len(names), 16))
    acc.clear(); cnt.clear()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        factor.nmf_device(Xd, G.n, 6, omega, plan=plan)
    torch.cuda.synchronize()
    print(f'nmf {(time.perf_counter() - t0) / steps * 1e3:.2f} ms/step')
    for k in sorted(acc, key=acc.get, reverse=True):
        print(f'  {k:22s} {acc[k] / steps * 1e3:7.3f} ms/step  {cnt[k] / steps:5.1f} calls/step')
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
