"""Stress of the k-means++ seeding's formulations against each other (no sklearn): random sizes, level counts and
distributions through grx_kmeans1d; prints one line per case with a hash of the quantised values and the centres.
The environment switches of csrc/grx_kmeans.hip are read once per process, so run it once per mode and diff the outputs:
    python tools/stress_kmeans_modes.py 80 3 > a.txt;  GRX_KMEANS_GAIN_PASS=1 GRX_KMEANS_MERGE=0 python ... > b.txt
Equal lines = equal bits (the modes may only differ where two candidate potentials agree to 1e-12)."""
import hashlib
import sys

import numpy as np

sys.path.insert(0, __file__.rsplit('/', 2)[0])
from graphrole_amd import kernels as K  # noqa: E402


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    rng = np.random.default_rng(seed)
    for c in range(cases):
        m = int(10 ** rng.uniform(3.7, 6.6))
        k = int(min(m // 4, 2 + 10 ** rng.uniform(0.3, 2.9)))
        kind = int(rng.integers(0, 5))
        if kind == 0:
            v = rng.gamma(0.6, 1.0, size=m)
        elif kind == 1:
            v = np.abs(rng.standard_normal(m)) * rng.choice([1e-3, 1.0, 40.0], size=m)
        elif kind == 2:
            v = rng.lognormal(0.0, 3.0, size=m)
        elif kind == 3:
            v = np.round(rng.gamma(2.0, 3.0, size=m), 2)          # many ties
        else:
            v = np.concatenate([rng.uniform(0, 1e-6, m // 2), rng.uniform(5, 6, m - m // 2)])   # two far clusters
            rng.shuffle(v)
        q, centers, info = K.kmeans1d(K.to_device(v), k)
        q, centers, info = K.to_host(q), K.to_host(centers), K.to_host(info)
        h = hashlib.sha256(q.tobytes() + centers.tobytes()).hexdigest()[:16]
        print(c, m, k, kind, 'iters', int(info[0]), 'levels', int(info[2]), 'faults', int(info[3]), h, flush=True)


if __name__ == '__main__':
    main()
