"""Times grx_kmeans1d (the reference's encode, graphrole/roles/factor.py:29-49) on synthetic factor-like values:
python tools/time_kmeans.py [m k]...   (default: the two encode shapes of the bench: 6 M / 64 and 30 M / 512)."""
import json
import sys
import time

import numpy as np
import torch

sys.path.insert(0, __file__.rsplit('/', 2)[0])
from graphrole_amd import kernels as K  # noqa: E402


def main():
    args = [int(a) for a in sys.argv[1:]]
    shapes = list(zip(args[0::2], args[1::2])) or [(6_000_000, 64), (30_000_000, 512)]
    out = []
    for m, k in shapes:
        rng = np.random.default_rng(0)
        v = K.to_device(rng.gamma(0.7, 1.0, size=m))
        K.kmeans1d(v, k)
        torch.cuda.synchronize()
        best = 1e30
        for _ in range(3):
            t0 = time.perf_counter()
            q, c, info = K.kmeans1d(v, k)
            torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t0) * 1e3)
        info = K.to_host(info)
        out.append({'m': m, 'k': k, 'ms': round(best, 3), 'n_iter': int(info[0]), 'levels': int(info[2]), 'faults': int(info[3])})
        print(json.dumps(out[-1]), flush=True)


if __name__ == '__main__':
    main()
