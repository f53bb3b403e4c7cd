"""
-m gpu: RolX `encode` on the GPU against the reference's quantiser (graphrole/roles/factor.py:29-49 = sklearn
KMeans(n_clusters, random_state=1) on the flattened entries).
  * quantizer='kmeans' (default, grx_kmeans1d): the same procedure -- seeding draws of RandomState(1), Lloyd
    iterations, stopping rule, empty-cluster relocation -- compared NUMERICALLY with sklearn run here: equal
    iteration count, equal number of distinct levels, every quantised value within 1e-9.
  * quantizer='lloyd_max' (grx_lloyd_max): parity by property: <= n_bins distinct values;
    Lloyd-Max fixed-point conditions; quantisation error not above sklearn's; the exact optimum on small inputs.
"""
import warnings

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _sklearn_inertia(data, k):
    from sklearn.cluster import KMeans
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        km = KMeans(n_clusters=k, random_state=1).fit(data.reshape(-1, 1))
    return km.inertia_


def _datasets():
    from oracle import rolx
    rng = np.random.RandomState(0)
    X = np.abs(rng.randn(50000, 10)) * np.linspace(1, 30, 10)
    np.random.seed(0)
    W, H, _ = rolx.nmf(X, 6)
    return [
        ('uniform600', rng.rand(600), 8),
        ('uniform600_k64', rng.rand(600), 64),
        ('gamma60k', rng.gamma(0.5, 2.0, 60000), 64),
        ('spike200k', np.concatenate([rng.exponential(1, 150000), np.full(50000, 1e-3)]), 32),
        ('lognormal1m', rng.lognormal(0, 2, 1000000), 64),
        ('pareto300k', rng.pareto(1.5, 300000), 128),
        ('nmf_W', W.ravel(), 64),
        ('nmf_H', H.ravel(), 32),
        ('few_distinct', rng.randint(0, 20, 5000).astype(float), 8),
        ('signed', rng.randn(20000) * 3, 16),
        ('k256', rng.gamma(2.0, 1.0, 100000), 256),
        # more than 256 levels (n_bits = int(log2(n_roles * min(shape))) = 9..12 on wide feature tables,
        # roles/extract.py:72): companding start + strided Lloyd kernel
        ('k512', rng.gamma(2.0, 1.0, 100000), 512),
        ('k512_of_690', rng.gamma(0.8, 3.0, 690), 512),          # the r x F factor of a 115-feature table, r = 6
        ('k1024_of_1840', rng.gamma(0.8, 3.0, 1840), 1024),      # r = 16, F = 115: merge-across-smallest-gaps start
        ('k4096', rng.lognormal(0, 2, 300000), 4096),
    ]


@pytest.mark.parametrize('case', _datasets(), ids=lambda c: c[0])
def test_lloyd_max_properties_and_error_vs_sklearn(case):
    from graphrole_amd.roles import factor
    name, data, k = case
    q = factor.encode(data.reshape(-1, 1), k, quantizer='lloyd_max').ravel()
    centres = np.unique(q)
    assert len(centres) <= k
    # centroid condition: each output value is the mean of the inputs mapped to it
    for c in centres:
        members = data[q == c]
        assert abs(members.mean() - c) <= 1e-9 * max(1.0, abs(c)), name
    # nearest-neighbour condition: no other centre is closer (ties allowed)
    d_own = np.abs(data - q)
    d_best = np.min(np.abs(data[:, None] - centres[None, :]), axis=1) if len(data) <= 200000 else None
    if d_best is not None:
        assert np.all(d_own <= d_best * (1 + 1e-9) + 1e-12), name
    # monotone: larger inputs never map to smaller centres
    order = np.argsort(data, kind='stable')
    assert np.all(np.diff(q[order]) >= 0)
    # error not above the reference quantiser's
    inertia = float(((data - q) ** 2).sum())
    if k > 1024:
        return                                                     # sklearn needs minutes here; fixed-point checks above
    ref = _sklearn_inertia(data, k)
    # exact-DP start (<= 256 levels, or <= 1024 values): never worse than sklearn; the companding / gap-merge
    # starts of the many-level path are near-optimal
    slack = 1e-6 if (k <= 256 or len(data) <= 1024) else 0.05
    assert inertia <= ref * (1 + slack) + 1e-12, (name, inertia, ref)


def test_lloyd_max_small_input_is_the_exact_optimum():
    """<= 1024 values: the DP stage sees every value, so the result is the global k-means optimum."""
    from graphrole_amd.roles import factor
    rng = np.random.RandomState(3)
    for m, k in [(12, 3), (60, 32), (200, 7), (1000, 16)]:
        data = np.sort(rng.lognormal(0, 1.5, m))
        q = factor.encode(data.reshape(1, -1), k, quantizer='lloyd_max').ravel()
        got = float(((data - q) ** 2).sum())
        # brute-force DP over the sorted values
        P = np.concatenate([[0.0], np.cumsum(data)])
        P2 = np.concatenate([[0.0], np.cumsum(data * data)])
        D = np.full(m + 1, np.inf)
        D[0] = 0.0
        for j in range(k):
            new = np.full(m + 1, np.inf)
            for i in range(j + 1, m + 1):
                mm = np.arange(j, i)
                n = i - mm
                cost = (P2[i] - P2[mm]) - (P[i] - P[mm]) ** 2 / n
                new[i] = np.min(D[mm] + np.maximum(cost, 0))
            D = new
        assert got <= D[m] * (1 + 1e-9) + 1e-12, (m, k, got, D[m])


def test_encode_shape_errors_and_determinism():
    from graphrole_amd.roles import factor
    rng = np.random.RandomState(0)
    X = rng.rand(20, 30)
    for quantizer in ('kmeans', 'lloyd_max'):
        for n_bins in range(1, 8):                               # reference test_factor.py:27-31
            enc = factor.encode(X, n_bins, quantizer=quantizer)
            assert enc.shape == X.shape
            assert len(np.unique(enc)) <= n_bins
        with pytest.raises(ValueError, match='n_clusters'):      # sklearn's error, relied on by _select_model
            factor.encode(rng.rand(3, 2), 8, quantizer=quantizer)
        a = factor.encode(X, 5, quantizer=quantizer)
        b = factor.encode(X, 5, quantizer=quantizer)
        assert np.array_equal(a, b)
        # constant input: a single level
        assert np.unique(factor.encode(np.full((10, 3), 2.5), 4, quantizer=quantizer)).tolist() == [2.5]
    with pytest.raises(ValueError, match='quantizer'):
        factor.encode(X, 4, quantizer='median-cut')


def test_encode_at_rolx_scale():
    """1 M x 6 node-role factor with 64 levels (the size that takes sklearn ~30 s on the host)."""
    from graphrole_amd.roles import factor
    rng = np.random.RandomState(1)
    G = rng.gamma(0.7, 1.0, size=(1_000_000, 6))
    enc = factor.encode(G, 64)
    assert enc.shape == G.shape and len(np.unique(enc)) <= 64
    rel = np.sqrt(((G - enc) ** 2).mean()) / G.std()
    assert rel < 0.05


def test_role_extractor_many_levels():
    """RoleExtractor(n_roles=k) on a wide table asks for 2**int(log2(k * F)) >= 512 levels
    (roles/extract.py:69-72); the reference works there, so must this."""
    import pandas as pd
    from graphrole_amd import RoleExtractor
    rng = np.random.RandomState(5)
    X = pd.DataFrame(np.abs(rng.randn(4000, 125)) * np.linspace(1, 20, 125))
    np.random.seed(0)
    rx = RoleExtractor(n_roles=5)                                 # 5 * 125 = 625 -> 9 bits -> 512 levels
    rx.extract_role_factors(X)
    assert rx.node_role_factor.shape == (4000, 5) and rx.role_feature_factor.shape == (5, 125)
    assert len(np.unique(rx.node_role_factor.values)) <= 512
    assert len(np.unique(rx.role_feature_factor.values)) <= 512


# ---------------------------------------------------------------- the reference's quantiser, numerically
def _kmeans_cases():
    rng = np.random.RandomState(0)
    from oracle import rolx
    X = np.abs(rng.randn(20000, 10)) * np.linspace(1, 30, 10)
    np.random.seed(0)
    W, H, _ = rolx.nmf(X, 6)
    return [
        ('u600_k8', rng.rand(600), 8),
        ('u600_k64', rng.rand(600), 64),
        ('gamma20k_k64', rng.gamma(0.5, 2.0, 20000), 64),
        ('spike_k32', np.concatenate([rng.exponential(1, 3000), np.full(1000, 1e-3)]), 32),
        ('ints_k8', rng.randint(0, 20, 5000).astype(float), 8),
        ('ints_k30_more_levels_than_values', rng.randint(0, 20, 500).astype(float), 30),
        ('k_eq_m', rng.rand(40), 40),
        ('k512_of_690', rng.gamma(0.8, 3.0, 690), 512),
        ('signed_k16', rng.randn(20000) * 3, 16),
        ('lognormal200k_k128', rng.lognormal(0, 2, 200000), 128),
        ('nmf_W_k64', W.ravel(), 64),
        ('nmf_H_k32', H.ravel(), 32),
        ('k1', rng.rand(50), 1),
        ('k2_two_values', np.where(rng.rand(1000) < 0.3, 1.0, 4.0), 2),
        ('k256_1m', rng.gamma(2.0, 1.0, 1000000), 256),
    ]


@pytest.mark.parametrize('case', _kmeans_cases(), ids=lambda c: c[0])
def test_kmeans_quantizer_equals_sklearn(case):
    """grx_kmeans1d against sklearn.cluster.KMeans(n_clusters=k, random_state=1) -- the reference's quantiser
    (graphrole/roles/factor.py:41-48) -- run here on the same values: same n_iter_, same number of distinct
    levels, every entry's level within 1e-9 (relative to the data's scale)."""
    from sklearn.cluster import KMeans
    from graphrole_amd import kernels as K
    name, data, k = case
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        km = KMeans(n_clusters=k, random_state=1).fit(data.reshape(-1, 1))
    ref = km.cluster_centers_[km.labels_].ravel()
    q, centres, info = K.kmeans1d(K.to_device(data), k)
    q, info = K.to_host(q), K.to_host(info)
    scale = max(np.abs(data).max(), 1e-300)
    if len(np.unique(data)) >= k:
        # (more levels than distinct values: sklearn relocates empty clusters to points that are all at distance
        # zero from their centres -- an arbitrary choice that only moves the iteration count)
        assert int(info[0]) == km.n_iter_, (name, int(info[0]), km.n_iter_)
    assert np.abs(q - ref).max() <= 1e-9 * scale, (name, np.abs(q - ref).max())
    assert int(info[2]) == len(np.unique(ref)) == len(np.unique(q))
    assert int(info[3]) == 0                                     # the seeding's internal consistency checks
    if len(np.unique(data)) >= k:
        # centres in seed order, like cluster_centers_
        np.testing.assert_allclose(K.to_host(centres), km.cluster_centers_[:, 0], rtol=0, atol=1e-9 * scale)


def test_kmeans_quantizer_equals_the_oracle_restatement():
    from graphrole_amd.roles import factor
    from oracle import kmeans1d
    rng = np.random.RandomState(9)
    X = rng.gamma(0.7, 2.0, size=(3000, 5))
    for n_bins in (4, 16, 100):
        enc = factor.encode(X, n_bins)                           # default quantizer = 'kmeans'
        ref, _, _ = kmeans1d.kmeans_quantize(X, n_bins)
        assert np.abs(enc - ref).max() <= 1e-9 * np.abs(X).max()


_AB_DRIVER = '''
import sys
sys.path.insert(0, ROOT)
import numpy as np, torch
from graphrole_amd import kernels as K
out = {}
for tag, m, k, seed in (('a', 50_000, 64, 0), ('b', 300_007, 256, 1), ('c', 4_099, 17, 2), ('d', 1_200_000, 512, 3),
                        ('e', 200_000, 32, 4), ('f', 90_000, 1024, 5), ('g', 130, 100, 6), ('h', 3_000, 64, 7),
                        ('i', 900, 512, 8), ('j', 4_096, 700, 9), ('k', 20_000, 2_500, 10), ('l', 2_100_000, 40, 11)):
    rng = np.random.default_rng(seed)
    v = np.abs(rng.standard_normal(m)) * rng.choice([1e-3, 1.0, 40.0], size=m)
    if tag == 'e':
        v = np.round(v, 1)                                       # few distinct values: long runs of ties
    if tag == 'f':
        v = rng.lognormal(0.0, 4.0, size=m)                      # fourteen orders of magnitude
    q, centers, info = K.kmeans1d(K.to_device(v), k)
    out[tag + '_q'] = K.to_host(q); out[tag + '_c'] = K.to_host(centers); out[tag + '_i'] = K.to_host(info)
np.savez(OUT, **out)
'''


def test_interval_seeding_equals_the_full_pass(tmp_path):
    """The one-dimensional k-means++ (csrc/grx_kmeans.hip): a candidate's potential and the update only visit the range
    of sorted values between the midpoints to the neighbouring seeds (widened by a rounding bound).  With
    GRX_KMEANS_FULL_RANGE=1 every range is [0, m) -- sklearn's own formulation, every value tested for every candidate --
    in the same exact integer arithmetic: the ranges are supersets of what can change, so seeds, centres and quantised
    values are identical bits; and the internal consistency checks report nothing in either mode.  Likewise the
    one-workgroup seeding of few values (m <= 4096, km_seed_small_kernel) against the many-launch path (GRX_KMEANS_SMALL=0),
    and the default evaluation of the candidates' potentials from sums over blocks of sorted values (closed form, fp64)
    against the pass over each range in integer arithmetic (GRX_KMEANS_GAIN_PASS=1): the two can only choose differently
    where two potentials agree to 1e-12, which none of these inputs has.  And the update inside the pick kernel (late
    seeds) against km_update_kernel for every seed (GRX_KMEANS_MERGE=0) and against the merged update for every seed."""
    import os
    import subprocess
    import sys
    import textwrap
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    results = []
    # default; every range [0, m); few values (m <= 4096) through the many-launch path instead of the one-workgroup kernel
    for tag, extra in (('default', {}), ('full', {'GRX_KMEANS_FULL_RANGE': '1'}), ('nosmall', {'GRX_KMEANS_SMALL': '0'}),
                       ('slowpick', {'GRX_KMEANS_SLOW_PICK': '1', 'GRX_KMEANS_SMALL': '0'}),
                       ('nomerge', {'GRX_KMEANS_MERGE': '0', 'GRX_KMEANS_SMALL': '0'}),
                       ('merge_all', {'GRX_KMEANS_MERGE': '1e9', 'GRX_KMEANS_SMALL': '0'}),
                       ('gainpass', {'GRX_KMEANS_GAIN_PASS': '1'}), ('gainpass_nosmall', {'GRX_KMEANS_GAIN_PASS': '1',
                                                                                          'GRX_KMEANS_SMALL': '0'})):
        out = tmp_path / f'km_{tag}.npz'
        code = 'ROOT = %r\nOUT = %r\n' % (root, str(out)) + textwrap.dedent(_AB_DRIVER)
        env = dict(os.environ, **extra)
        res = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=900, env=env)
        assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-3000:]
        results.append(np.load(out))
    a = results[0]
    for b in results[1:]:
        for key in a.files:
            assert np.array_equal(a[key], b[key]), key
            if key.endswith('_i'):
                assert int(a[key][3]) == 0 and int(b[key][3]) == 0, (key, a[key], b[key])


def test_kmeans_runs_are_bitwise_repeatable():
    """All sums of the seeding are integers: no result depends on the order of an atomic or a reduction."""
    from graphrole_amd import kernels as K
    rng = np.random.default_rng(11)
    v = K.to_device(np.abs(rng.standard_normal(700_001)) * rng.choice([1e-3, 1.0, 40.0], size=700_001))
    first = [K.to_host(t) for t in K.kmeans1d(v, 128)]
    for _ in range(3):
        again = [K.to_host(t) for t in K.kmeans1d(v, 128)]
        for x, y in zip(first, again):
            assert np.array_equal(x, y)


def test_a_seeding_fault_is_an_error_not_a_result():
    """grx_kmeans1d checks itself while it seeds (include/grx.h, d_info[3]); the host code turns any report into an
    exception.  The one report an input can provoke: values beyond 1e144, whose squared distances leave the fixed-point
    range of the exact sums (sklearn's own distances overflow a little further up)."""
    from graphrole_amd.roles import factor
    x = np.array([[1.0, 2.0, 3.0e150], [4.0, 5.0, 6.0]])
    with pytest.raises(factor.QuantizerFault):
        factor.encode(x, 2)
    assert factor.encode(x / 1e150, 2).shape == x.shape


def test_an_expired_wait_ends_the_run_at_once_and_is_reported(tmp_path):
    """The workgroups of a pick launch wait for each other's sums (tagged words, csrc/grx_kmeans.hip).  The waits are
    bounded; GRX_KMEANS_LOSE_A_SUM=<seed> makes one workgroup of that seed's launch withhold its sum: the others' wait
    must expire (a fraction of a second), the remaining launches of the run must return at once, d_info[3] must carry
    bit 4 and encode() must raise -- no hang, no result."""
    import os
    import subprocess
    import sys
    import textwrap
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = 'ROOT = %r\n' % root + textwrap.dedent('''
        import sys, time
        sys.path.insert(0, ROOT)
        import numpy as np, torch
        from graphrole_amd import kernels as K
        from graphrole_amd.roles import factor
        v = np.abs(np.random.default_rng(0).standard_normal(300_000))
        K.kmeans1d(K.to_device(v[:5000]), 8)          # (load the library, warm up: the small path is not affected)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        q, c, info = K.kmeans1d(K.to_device(v), 200)
        info = K.to_host(info)
        dt = time.perf_counter() - t0
        print('FAULTS', int(info[3]), 'SECONDS', round(dt, 2))
        try:
            factor.encode(v.reshape(-1, 3), 200)
            print('RAISED no')
        except factor.QuantizerFault as e:
            print('RAISED yes')
    ''')
    t0 = time.perf_counter()
    res = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=240,
                         env=dict(os.environ, GRX_KMEANS_LOSE_A_SUM='7'))
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-3000:]
    out = res.stdout
    faults = int(out.split('FAULTS')[1].split()[0])
    seconds = float(out.split('SECONDS')[1].split()[0])
    assert faults & 16, out
    assert seconds < 20.0, out
    assert 'RAISED yes' in out, out
    assert time.perf_counter() - t0 < 200
