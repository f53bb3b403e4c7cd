"""Shared helpers for the tests: golden-fixture loading and synthetic graphs."""
import json
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
#: THE tolerance of weighted graphs (README): the reference adds edge weights in the iteration order
#: of Python sets, the device in CSR order -- the tables agree to this relative error, everything else is bit-exact
WEIGHTED_RTOL = 1e-11
#: ... and the tighter bound that holds on the reference-generated fixtures (<= 2 000 nodes, rows of <= a few hundred
#: terms): the loosening to 1e-11 is for hubs that add ~10^5 weights (5 M / 100 M graph), not for these
WEIGHTED_RTOL_FIXTURES = 1e-12
GOLDEN = os.path.join(ROOT, 'tests', 'golden')

REFEX_CASES = ['karate', 'karate_weighted', 'er300', 'ba300', 'dw200_attrs', 'loops_dangling150',
               'directed120', 'path4', 'iface7', 'iface7_dw', 'er2000', 'ba2000',
               'karate_minmax', 'ba300_maxsum', 'dw200_minmax', 'loops_dangling150_minmax', 'ba300_stdvar',
               'karate_sumstd', 'iface7_prod', 'dw200_prod', 'path4_prod']
# round 3: aggregation lists with 'prod' over integer columns (wrapping int64), 'median', 'count' / 'size'
TYPED_CASES = ['karate_prodwrap', 'karate_sumprodwrap', 'ba300_prodmean', 'karate_summedian', 'ba300_medianmean',
               'dw200_medianmax', 'loops_dangling150_median', 'karate_sumcount', 'er300_meansize',
               'loops_dangling150_countmax', 'directed120_prodcount']
ROLES_CASES = ['karate', 'karate_weighted', 'er300', 'ba300', 'dw200_attrs', 'directed120', 'loops_dangling150']
NMF_CASES = ['rand20x30_r3', 'rand500x12_r6', 'rand800x40_r6', 'rand3000x9_r2', 'karate_r4', 'er2000_r6',
             'ba2000_r6', 'dw200_r5']


class Golden:
    def __init__(self, path):
        self.z = np.load(path, allow_pickle=False)

    def __getitem__(self, k):
        return self.z[k]

    def js(self, k):
        return json.loads(str(self.z[k + '_json']))

    def __contains__(self, k):
        return k in self.z.files


def golden_aggs(g):
    """aggs list a ReFeX fixture was generated with (older fixtures: the reference default)."""
    return g.js('aggs') if 'aggs_json' in g else ['sum', 'mean']


def load_refex(name):
    return Golden(os.path.join(GOLDEN, f'refex_{name}.npz'))


def golden_path(filename):
    return os.path.join(GOLDEN, filename)


def load_roles(name):
    return Golden(os.path.join(GOLDEN, f'roles_{name}.npz'))


def load_nmf(name):
    return Golden(os.path.join(GOLDEN, f'nmf_{name}.npz'))


def oracle_graph_from_golden(g):
    from oracle import refex
    w = g['w'] if len(g['w']) else None
    og = refex.graph_from_arrays(int(g['n']), g['src'], g['dst'], w, bool(g['directed']), g.js('labels'))
    og.num_edges = int(g['num_edges'])
    if 'adj_idx' in g:
        # the adjacency (insertion) order of the graph the reference ran on
        assert np.array_equal(g['adj_ptr'], og.row_ptr)
        og.adj_col = g['adj_idx'].astype(np.int32)
    attach_orders(og, g)
    return og


def attach_orders(og, z):
    """node iteration order / predecessor insertion order of the graph the reference ran on (tools/make_golden_order.py):
    what the ORDER of the reference's generation-0 additions depends on besides the adjacency order"""
    if 'node_order' in z:
        og.node_order = z['node_order']
    if 'pred_ptr' in z:
        og.pred_ptr, og.pred_col = z['pred_ptr'], z['pred_idx'].astype(np.int64)


WEIGHTED_ORDER_CASES = ['karate_weighted', 'dw200_attrs', 'iface7_dw', 'dw200_minmax', 'dw200_prod', 'dw200_medianmax']
GEN0W_CASES = ['rw60', 'rwd80', 'rw400']


def typed_gen0(g):
    """Generation-0 columns of a fixture with the reference's dtypes (int64 where its frame had integers)."""
    names = g.js('gen0_names')
    kinds = g.js('gen0_dtypes') if 'gen0_dtypes_json' in g else ['int64' if bool(g['gen0_is_int']) else 'float64'] * len(names)
    vals = g['gen0_values']
    return names, [vals[:, j].astype(np.int64) if 'int' in kinds[j] else vals[:, j].copy() for j in range(len(names))]


def assert_typed_final_equal(g, columns, arrays, rtol=0.0):
    """Final table of a typed run against the fixture: column order, dtypes, integer columns bit for bit (int64, also
    beyond 2^53), float columns exactly (unweighted) or to rtol."""
    assert list(columns) == g.js('final_columns')
    kinds = g.js('final_dtypes')
    for j, nm in enumerate(columns):
        got = np.asarray(arrays[nm])
        assert str(got.dtype) == kinds[j], f'{nm}: dtype {got.dtype}, reference {kinds[j]}'
        if 'int' in kinds[j]:
            assert np.array_equal(got, g['final_values_i64'][:, j]), nm
        elif rtol:
            np.testing.assert_allclose(got, g['final_values'][:, j], rtol=rtol, atol=0, err_msg=nm)
        else:
            assert np.array_equal(got, g['final_values'][:, j]), nm


def random_graph(n, m, seed, directed=False, weighted=False, self_loops=0):
    """Unique random edges (each undirected edge listed once)."""
    rng = np.random.default_rng(seed)
    src = rng.integers(0, n, size=3 * m)
    dst = rng.integers(0, n, size=3 * m)
    keep = src != dst
    src, dst = src[keep], dst[keep]
    if not directed:
        lo, hi = np.minimum(src, dst), np.maximum(src, dst)
        src, dst = lo, hi
    key = src.astype(np.int64) * n + dst
    _, idx = np.unique(key, return_index=True)
    idx = np.sort(idx)[:m]
    src, dst = src[idx], dst[idx]
    if self_loops:
        loops = rng.choice(n, size=self_loops, replace=False)
        src = np.concatenate([src, loops])
        dst = np.concatenate([dst, loops])
    w = rng.uniform(0.1, 5.0, size=len(src)) if weighted else None
    return src, dst, w


def powerlaw_graph(n, m, seed):
    """Barabasi-Albert style preferential attachment (repeated-nodes method), undirected, unique edges."""
    rng = np.random.default_rng(seed)
    src = np.empty((n - m) * m, dtype=np.int64)
    dst = np.empty((n - m) * m, dtype=np.int64)
    repeated = np.empty(2 * (n - m) * m + m, dtype=np.int64)
    repeated[:m] = np.arange(m)
    fill = m
    pos = 0
    targets = np.arange(m)
    for v in range(m, n):
        src[pos:pos + m] = v
        dst[pos:pos + m] = targets
        pos += m
        repeated[fill:fill + m] = targets
        repeated[fill + m:fill + 2 * m] = v
        fill += 2 * m
        chosen = set()
        while len(chosen) < m:
            chosen.update(repeated[rng.integers(0, fill, size=m - len(chosen))].tolist())
        targets = np.fromiter(chosen, dtype=np.int64, count=m)
    return src, dst, None
