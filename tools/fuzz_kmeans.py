"""Differential fuzz of grx_kmeans1d (the reference's encode, graphrole/roles/factor.py:29-49) against
sklearn.cluster.KMeans(n_clusters=k, random_state=1) on the GPU box: random sizes (incl. the one-workgroup path m <= 4096),
level counts and value distributions aimed at the one-dimensional seeding (heavy ties, clusters at tiny spacing next to
far outliers, many orders of magnitude, signed data, factor-like gamma tails).  Per case: the seeding's consistency
faults must be 0; n_iter_, the number of distinct levels and every quantised value (1e-9 of the data's scale) are compared;
cases that differ are REPORTED, not hidden.  Two classes are known and not bugs of the seeding (round 5's O(m k) seeding
gives the same bits there): (i) two candidates that capture only themselves and each other have mathematically EQUAL
potentials; sklearn's argmin is then decided by the last bit of two BLAS sums (observed 1 ulp apart), here the first
wins; (ii) fewer distinct values than levels: sklearn's relocation of empty clusters is arbitrary there.   Usage: PYTHONPATH=. python tools/fuzz_kmeans.py [cases] [seed]"""
import sys
import warnings

import numpy as np

from graphrole_amd import kernels as K


def values(rng, m):
    kind = int(rng.integers(0, 9))
    if kind == 0:
        return rng.gamma(rng.uniform(0.3, 2.0), 1.0, m)
    if kind == 1:
        return rng.lognormal(0, rng.uniform(0.5, 4.0), m)
    if kind == 2:
        return np.round(rng.exponential(1.0, m), int(rng.integers(0, 3)))          # heavy ties
    if kind == 3:
        return rng.standard_normal(m) * 10.0 ** rng.integers(-6, 6)
    if kind == 4:
        x = 1.0 + rng.integers(0, 1000, m) * 1e-9                                    # a cluster at tiny spacing
        x[rng.integers(0, m, max(m // 200, 1))] = rng.uniform(50, 5000, max(m // 200, 1))
        return x
    if kind == 5:
        return np.where(rng.random(m) < 0.7, 0.0, rng.pareto(1.5, m))
    if kind == 6:
        return rng.integers(0, int(rng.choice([3, 20, 500])), m).astype(np.float64)
    if kind == 7:
        return np.abs(rng.standard_normal(m)) * rng.choice([1e-3, 1.0, 40.0], size=m)
    return rng.random(m)


def main():
    from sklearn.cluster import KMeans
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    differ = []
    for case in range(cases):
        m = int(rng.choice([rng.integers(8, 300), rng.integers(300, 4096), rng.integers(4097, 60000), rng.integers(60000, 400000)]))
        k = int(min(m, rng.choice([rng.integers(2, 9), rng.integers(9, 70), rng.integers(70, 400)])))
        v = values(rng, m)
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            km = KMeans(n_clusters=k, random_state=1).fit(v.reshape(-1, 1))
        ref = km.cluster_centers_[km.labels_].ravel()
        q, _, info = K.kmeans1d(K.to_device(v), k)
        q, info = K.to_host(q), K.to_host(info)
        assert int(info[3]) == 0, (case, m, k, info)
        scale = max(np.abs(v).max(), 1e-300)
        distinct = len(np.unique(v)) >= k
        same = np.abs(q - ref).max() <= 1e-9 * scale and (not distinct or int(info[0]) == km.n_iter_)
        if not same:
            differ.append((case, m, k, int(info[0]), int(km.n_iter_), float(np.abs(q - ref).max() / scale)))
        if case % 10 == 9:
            print(case + 1, 'm', m, 'k', k, 'differ so far', len(differ), flush=True)
    print(f'fuzz_kmeans: {cases} cases, faults 0 in all, {cases - len(differ)} equal to sklearn, {len(differ)} differ: {differ}')


if __name__ == '__main__':
    main()
