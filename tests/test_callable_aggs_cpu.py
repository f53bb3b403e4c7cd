"""
-m "not gpu": user-callable aggregations (graphrole/features/extract.py:26,47,111 -- the aggs list goes straight to
DataFrame.agg) through the drop-in class on the CPU test double, against tables the REFERENCE produced with the same
functions (tests/golden/refex_callable_*.npz, tools/make_golden_callables.py).  The GPU twin is
tests/test_gpu_callable_aggs.py.
"""
import json

import numpy as np
import pytest

from tests import graphs as G_
from tests import util


def build_extractor(name):
    import networkx as nx
    from graphrole_amd import RecursiveFeatureExtractor
    g = util.Golden(util.golden_path(f'refex_callable_{name}.npz'))
    graph, spec, max_generations = G_.CALLABLE_CASES[name]
    if graph == 'karate':
        # the fixture's own edges in the adjacency order of the graph the reference ran on: order-sensitive
        # aggregations ('skew', 'sem': floating-point sums over G[node]) see the neighbours as the reference did
        labels = g.js('labels')
        G = nx.Graph()
        G.add_nodes_from(labels)
        G.add_edges_from((labels[a], labels[b]) for a, b in zip(g['src'], g['dst']))
        adj_ptr, adj_idx = g['adj_ptr'], g['adj_idx']
        for i, lab in enumerate(labels):
            G._adj[lab] = {labels[j]: G._adj[lab][labels[j]] for j in adj_idx[adj_ptr[i]:adj_ptr[i + 1]]}
        kwargs = {}
    else:
        G, kwargs = G_.BUILDERS[graph]()
    return g, RecursiveFeatureExtractor(G, max_generations=max_generations, aggs=G_.resolve_aggs(spec), **kwargs)


def check_against_golden(g, fe):
    X = fe.extract_features()
    assert list(X.columns) == g.js('final_columns')
    assert fe.generation_count == int(g['generation_count'])
    assert [str(t) for t in X.dtypes] == g.js('final_dtypes')
    assert list(X.index) == g.js('labels')
    weighted = len(g['w']) > 0
    if weighted:
        np.testing.assert_allclose(X.values.astype(float), g['final_values'], rtol=1e-12, atol=0)
    else:
        assert np.array_equal(X.values.astype(float), g['final_values'])


@pytest.mark.parametrize('name', list(G_.CALLABLE_CASES))
def test_callable_aggs_equal_reference_through_the_test_double(name):
    from graphrole_amd import backend
    from tests import fake_kernels
    backend.use(fake_kernels)
    try:
        g, fe = build_extractor(name)
        check_against_golden(g, fe)
    finally:
        backend.use(None)


def test_callable_rules():
    import networkx as nx
    import pandas as pd
    from graphrole_amd import RecursiveFeatureExtractor, backend
    from graphrole_amd.features.extract import _agg_name, _has_kernel
    from tests import fake_kernels

    def sum(s):                                            # a user function that merely shares a kernel's name
        return float(s.to_numpy()[0])                       # ... and returns the FIRST neighbour's value
    assert [_has_kernel(a) for a in ['sum', np.sum, pd.DataFrame.mean, max, np.median, pd.Series.std]] == [True] * 6
    assert [_has_kernel(a) for a in [np.ptp, sum, G_.spread, (lambda s: 1.0)]] == [False] * 4
    assert _agg_name(np.max) == 'max' and _agg_name(G_.spread) == 'spread' and _agg_name(lambda s: 0) == '<lambda>'
    backend.use(fake_kernels)
    try:
        G = nx.path_graph(6)
        with pytest.raises(AttributeError, match='not a valid function'):
            RecursiveFeatureExtractor(G, aggs=['no_such_agg']).extract_features()   # pandas' own error, as in the reference
        with pytest.raises(ValueError, match='unique'):
            RecursiveFeatureExtractor(G, aggs=[lambda s: s.max(), lambda s: s.min()]).extract_features()
        with pytest.raises(TypeError, match='one row per function'):
            RecursiveFeatureExtractor(G, aggs=[np.ptp]).extract_features()           # pandas 2 applies it element-wise
        K = nx.karate_club_graph()
        for _, _, d in K.edges(data=True):
            d.clear()
        X = RecursiveFeatureExtractor(K, max_generations=2, aggs=[sum]).extract_features()
        made = [c for c in X.columns if c.endswith('(sum)')]
        assert made                                           # the user's function ran, not the kernel:
        for c in made:
            parent = c[:-len('(sum)')]
            assert [X.loc[v, c] for v in K.nodes] == [float(X.loc[next(iter(K[v])), parent]) for v in K.nodes]
    finally:
        backend.use(None)
