"""
-m gpu: end-to-end parity of graphrole_amd.RecursiveFeatureExtractor (HIP path) with the reference,
through golden vectors the reference produced (tests/golden, tools/make_golden.py), plus the
reference's own unit-test cases (tests/test_features/test_extract.py, tests/test_graph/
test_interface.py) restated against the drop-in classes.

Tolerances: column lists, retained sets per generation, dtypes, index order: exact.
Values: BIT-EXACT for unweighted graphs (the neighbour sums follow numpy's pairwise tree in
adjacency order, like the reference's Series.sum()); rtol util.WEIGHTED_RTOL (1e-11) for weighted graphs, whose
generation-0 columns the reference sums with Python's sum() over ego-graph edge views.
"""
import networkx as nx
import numpy as np
import pandas as pd
import pytest

from tests import graphs, util

pytestmark = pytest.mark.gpu

RTOL = util.WEIGHTED_RTOL        # one tolerance for weighted graphs, defined once (tests/util.py)
RTOL_FIXTURES = util.WEIGHTED_RTOL_FIXTURES     # the reference-generated fixtures hold the tighter bound


def _check_values(actual, expected, weighted):
    if weighted:
        np.testing.assert_allclose(actual, expected, rtol=RTOL_FIXTURES, atol=0)
    else:
        assert np.array_equal(actual, expected), f'{int((actual != expected).sum())} entries differ'


def _graph_for(name, g):
    """networkx graph of a golden case: builders for synthetic cases, fixture arrays for karate
    (with the adjacency order of the graph the reference ran on restored)."""
    name = graphs.CASE_AGGS.get(name, (name,))[0]
    if name in graphs.BUILDERS:
        G, kwargs = graphs.BUILDERS[name]()
        return G, kwargs
    labels = g.js('labels')
    G = nx.DiGraph() if bool(g['directed']) else nx.Graph()
    G.add_nodes_from(labels)
    w = g['w']
    all_int = bool(len(w)) and all(float(x).is_integer() for x in w)     # karate weights are Python ints
    for k, (s, d) in enumerate(zip(g['src'], g['dst'])):
        if len(w):
            G.add_edge(labels[s], labels[d], weight=int(w[k]) if all_int else float(w[k]))
        else:
            G.add_edge(labels[s], labels[d])
    adj_ptr, adj_idx = g['adj_ptr'], g['adj_idx']
    for i, lab in enumerate(labels):                          # same edge-data dicts, golden order
        want = [labels[j] for j in adj_idx[adj_ptr[i]:adj_ptr[i + 1]]]
        G._adj[lab] = {v: G._adj[lab][v] for v in want}
    return G, g.js('kwargs')


@pytest.mark.parametrize('native', [True, False], ids=['grx_refex_run', 'per_kernel'])
@pytest.mark.parametrize('name', util.REFEX_CASES)
def test_extract_features_matches_reference(name, native):
    """Both drivers of the generation loop against the reference's golden tables: the whole loop below the
    ABI (grx_refex_run, the one-GPU product path) and the per-kernel sequence from Python (the path a
    ShardPlan uses)."""
    from graphrole_amd import RecursiveFeatureExtractor
    g = util.load_refex(name)
    G, kwargs = _graph_for(name, g)
    kwargs = dict(kwargs, native_loop=native)
    aggs = util.golden_aggs(g)
    if aggs == ['sum', 'mean']:
        fe = RecursiveFeatureExtractor(G, **kwargs)          # default aggs, like the reference's example
    elif int(g['max_generations']) != 10:
        fe = RecursiveFeatureExtractor(G, aggs=aggs, max_generations=int(g['max_generations']), **kwargs)
    else:
        fe = RecursiveFeatureExtractor(G, aggs=aggs, **kwargs)
    X = fe.extract_features()
    assert list(X.columns) == g.js('final_columns')
    assert list(X.index) == g.js('labels')
    assert fe.generation_count == int(g['generation_count'])
    assert [str(t) for t in X.dtypes] == g.js('final_dtypes')
    _check_values(X.values.astype(np.float64), g['final_values'], weighted=bool(len(g['w'])))
    for gen in range(int(g['n_generations_recorded'])):
        assert fe._final_names[gen] == g.js(f'g{gen}_retained'), f'generation {gen}'
    # the working set after the last generation (extract.py:135-141), in the reference's order
    last = int(g['n_generations_recorded']) - 1
    assert list(fe._features.columns) == g.js(f'g{last}_working_after')
    # per-generation counts
    for gen in range(int(g['n_generations_recorded'])):
        st = fe.stats[gen]
        assert st['candidates'] == len(g.js(f'g{gen}_cand_names')) and st['retained'] == len(g.js(f'g{gen}_retained'))
        assert st['working'] == len(g.js(f'g{gen}_working_before')) and st['dropped'] == len(g.js(f'g{gen}_dropped'))
    # memoised second call is identical (reference test_extract_features_back_to_back)
    pd.testing.assert_frame_equal(X, fe.extract_features())
    # a second run of the same instance reuses its arena and reproduces the table bit for bit
    fe.reset()
    pd.testing.assert_frame_equal(X, fe.extract_features())


@pytest.mark.parametrize('name', ['karate', 'er300', 'dw200_attrs', 'directed120'])
def test_generation_trace_matches_reference(name):
    """Drive the generations one by one, like tools/make_golden.py drove the reference."""
    from graphrole_amd import RecursiveFeatureExtractor
    g = util.load_refex(name)
    G, kwargs = _graph_for(name, g)
    fe = RecursiveFeatureExtractor(G, aggs=['sum', 'mean'], **kwargs)
    labels = g.js('labels')
    feats = fe.graph.get_neighborhood_features()
    assert list(feats.columns) == g.js('gen0_names')
    assert list(feats.index) == labels
    np.testing.assert_allclose(feats.values.astype(float), g['gen0_values'], rtol=RTOL_FIXTURES)
    fe._update(feats)
    assert list(fe._features.columns) == g.js('g0_working_after')
    for gen in range(1, int(g['n_generations_recorded'])):
        fe.generation_count = gen
        fe._feature_group_thresh = gen
        cand = fe._get_next_features()
        assert list(cand.columns) == g.js(f'g{gen}_cand_names')
        _check_values(cand.values, g[f'g{gen}_cand_values'], weighted=bool(len(g['w'])))
        fe._update(cand)
        assert list(fe._final_features[gen].keys()) == g.js(f'g{gen}_retained')
        assert list(fe._features.columns) == g.js(f'g{gen}_working_after')


def test_csr_graph_input_equals_networkx_input():
    from graphrole_amd import RecursiveFeatureExtractor
    from graphrole_amd.graph import CSRGraph
    g = util.load_refex('er2000')
    G, _ = _graph_for('er2000', g)
    X1 = RecursiveFeatureExtractor(G).extract_features()
    C = CSRGraph(int(g['n']), g['src'], g['dst'], adjacency=g['adj_idx'])
    X2 = RecursiveFeatureExtractor(C).extract_features()
    pd.testing.assert_frame_equal(X1, X2, check_exact=True)
    # without an explicit adjacency the sums run in order of appearance: same table up to re-association
    X3 = RecursiveFeatureExtractor(CSRGraph(int(g['n']), g['src'], g['dst'])).extract_features()
    pd.testing.assert_frame_equal(X1, X3, rtol=RTOL)
    # weighted + directed + attributes through arrays
    g = util.load_refex('dw200_attrs')
    G, kwargs = _graph_for('dw200_attrs', g)
    X1 = RecursiveFeatureExtractor(G, **kwargs).extract_features()
    labels = g.js('labels')
    attrs = {}
    for a in ('a_uniform', 'a_poisson', 'a_sparse'):
        attrs[a] = np.array([G.nodes[v].get(a, 0) for v in labels], dtype=float)
    C = CSRGraph(int(g['n']), g['src'], g['dst'], weights=g['w'], directed=True, attributes=attrs,
                 adjacency=g['adj_idx'])
    X2 = RecursiveFeatureExtractor(C, attributes=True).extract_features()
    assert list(X1.columns) == list(X2.columns)
    np.testing.assert_allclose(X1.values.astype(float), X2.values.astype(float), rtol=0, atol=0)


# ------------------------------------------------------------------ reference unit-test cases
class TestExtractorLikeReference:
    """tests/test_features/test_extract.py of the reference, against the drop-in class."""

    edges = [('a', 'b'), ('a', 'c'), ('c', 'd')]

    def _rfe(self):
        from graphrole_amd import RecursiveFeatureExtractor
        rfe = RecursiveFeatureExtractor(nx.Graph(self.edges), aggs=[np.sum, np.mean])
        rfe._features = rfe.graph.get_neighborhood_features()
        rfe._final_features = {0: rfe._features.to_dict()}
        rfe.generation_count = 1
        return rfe

    def test_unknown_graph_type_raises_type_error(self):
        from graphrole_amd import RecursiveFeatureExtractor

        class SomeGraph:
            pass
        with pytest.raises(TypeError, match='supported libraries'):
            RecursiveFeatureExtractor(SomeGraph)

    def test_empty_graph_raises_value_error(self):
        from graphrole_amd import RecursiveFeatureExtractor
        with pytest.raises(ValueError, match='at least one edge'):
            RecursiveFeatureExtractor(nx.Graph())

    def test_get_next_features_known_answers(self):
        expected = {                                      # test_extract.py:104-122
            'external_edges(sum)':  {'a': 2.0, 'b': 1.0, 'c': 2.0, 'd': 1.0},
            'degree(sum)':          {'a': 3.0, 'b': 2.0, 'c': 3.0, 'd': 2.0},
            'internal_edges(sum)':  {'a': 3.0, 'b': 2.0, 'c': 3.0, 'd': 2.0},
            'external_edges(mean)': {'a': 1.0, 'b': 1.0, 'c': 1.0, 'd': 1.0},
            'degree(mean)':         {'a': 1.5, 'b': 2.0, 'c': 1.5, 'd': 2.0},
            'internal_edges(mean)': {'a': 1.5, 'b': 2.0, 'c': 1.5, 'd': 2.0},
        }
        got = self._rfe()._get_next_features()
        assert list(got.columns) == ['degree(sum)', 'internal_edges(sum)', 'external_edges(sum)',
                                     'degree(mean)', 'internal_edges(mean)', 'external_edges(mean)']
        exp = pd.DataFrame(expected)
        assert np.array_equal(got.sort_index(axis=1).sort_index(axis=0).values,
                              exp.sort_index(axis=1).sort_index(axis=0).values)

    def test_update_prunes_duplicate_and_keeps_recorded(self):
        rfe = self._rfe()
        existing = rfe._features
        rng = np.random.RandomState(0)
        new = pd.concat([
            pd.DataFrame(existing['degree'].values, columns=['degree2'], index=existing.index),
            pd.DataFrame(rng.randn(existing.shape[0], 2), columns=['a', 'b'], index=existing.index),
        ], axis=1)
        rfe._update(new)
        expected = pd.concat([existing[['degree', 'external_edges']], new[['a', 'b']]], axis=1)
        pd.testing.assert_frame_equal(rfe._features, expected)
        final = rfe._finalize_features()
        expected_final = pd.concat([existing, new[['a', 'b']]], axis=1)
        pd.testing.assert_frame_equal(final.sort_index(axis=1), expected_final.sort_index(axis=1))

    def test_aggregated_df_to_dict(self):
        rfe = self._rfe()
        cols = ['feature1', 'feature2', 'feature3']
        df = pd.DataFrame(np.arange(6).reshape(2, 3), columns=cols, index=['sum', 'mean'])
        assert rfe._aggregated_df_to_dict(df) == {
            'feature1(sum)': 0, 'feature2(sum)': 1, 'feature3(sum)': 2,
            'feature1(mean)': 3, 'feature2(mean)': 4, 'feature3(mean)': 5}
        ser = pd.Series([6, 7, 8], index=cols, name='prod')
        assert rfe._aggregated_df_to_dict(ser) == {'feature1(prod)': 6, 'feature2(prod)': 7, 'feature3(prod)': 8}

    def test_finalize_features_merges_generations(self):
        rfe = self._rfe()
        data = {'a': {'a': 0, 'b': 5, 'c': 1, 'd': 2}, 'b': {'a': 1, 'b': 6, 'c': 2, 'd': 3},
                'c': {'a': 2, 'b': 7, 'c': 3, 'd': 4}, 'e': {'a': 4, 'b': 9, 'c': 5, 'd': 6}}
        expected = pd.DataFrame(data)
        rfe._final_features = {0: expected[['a', 'b']].to_dict(), 1: expected[['c']].to_dict(),
                               2: expected['e'].to_frame().to_dict()}
        final = rfe._finalize_features()
        assert list(final.columns) == ['e', 'c', 'a', 'b']           # latest generation first
        pd.testing.assert_frame_equal(final.sort_index(axis=1), expected.sort_index(axis=1))

    def test_dangling_nodes(self):
        from graphrole_amd import RecursiveFeatureExtractor
        G = nx.Graph()
        G.add_nodes_from(['a', 'b', 'c', 'd'])
        G.add_edge('a', 'c')
        rfe = RecursiveFeatureExtractor(G)
        feats = rfe.extract_features()
        assert feats.index.tolist() == ['a', 'b', 'c', 'd']
        assert feats.notnull().all().all()
        rfe2 = RecursiveFeatureExtractor(G)
        nf = rfe2.graph.get_neighborhood_features()
        rfe2._features = nf
        rfe2._final_features = {0: nf.to_dict()}
        rfe2.generation_count = 1
        nxt = rfe2._get_next_features()
        assert nxt.notnull().all().all()
        assert (nxt.loc[['b', 'd']] == 0).all().all()                 # neighbourless nodes -> 0

    def test_unsupported_aggregation_fails_loudly(self):
        from graphrole_amd import RecursiveFeatureExtractor
        # a NAME pandas does not know: pandas' own error, as in the reference (the list goes to DataFrame.agg);
        # names it knows but the device does not ('sem', 'nunique', ...) run on the host: tests/test_gpu_callable_aggs.py
        with pytest.raises(AttributeError, match='not a valid function'):
            RecursiveFeatureExtractor(nx.Graph(self.edges), aggs=['sum', 'no_such_agg']).extract_features()
        # a CALLABLE without a kernel is evaluated by pandas on the host (tests/test_gpu_callable_aggs.py): the same
        # numbers as the kernel-backed 'sum' here, under the name pandas gives a lambda
        X = RecursiveFeatureExtractor(nx.Graph(self.edges), aggs=[lambda s: s.sum()]).extract_features()
        Y = RecursiveFeatureExtractor(nx.Graph(self.edges), aggs=['sum']).extract_features()
        assert [c.replace('<lambda>', 'sum') for c in X.columns] == list(Y.columns)
        assert np.array_equal(X.values.astype(float), Y.values.astype(float))

    def test_agg_order_follows_aggs(self):
        from graphrole_amd import RecursiveFeatureExtractor
        rfe = RecursiveFeatureExtractor(nx.Graph(self.edges), aggs=['mean', 'sum'])
        nf = rfe.graph.get_neighborhood_features()
        rfe._features = nf
        rfe._final_features = {0: nf.to_dict()}
        rfe.generation_count = 1
        assert list(rfe._get_next_features().columns)[:2] == ['degree(mean)', 'internal_edges(mean)']


class TestInterfaceLikeReference:
    """tests/test_graph/test_interface.py of the reference (networkx adapter)."""

    def test_get_interface(self):
        from graphrole_amd.graph import interface
        klass = interface.get_interface(nx.Graph())
        assert isinstance(klass(nx.Graph()), interface.BaseGraphInterface)
        assert interface.get_interface(str) is None
        assert interface.get_interface('str') is None

    def test_counts_nodes_neighbors(self):
        from graphrole_amd.graph.interface import NetworkxInterface
        G, _ = graphs.iface7()
        t = NetworkxInterface(G)
        assert t.get_num_edges() == 7
        assert NetworkxInterface(nx.Graph()).get_num_edges() == 0
        assert set(t.get_nodes()) == set(range(7))
        nbrs = {0: {1, 2, 3}, 1: {0}, 2: {0}, 3: {0, 6}, 4: {5, 6}, 5: {4, 6}, 6: {3, 4, 5}}
        for node, exp in nbrs.items():
            assert set(t.get_neighbors(node)) == exp

    def test_neighborhood_features_undirected(self):
        from graphrole_amd.graph.interface import NetworkxInterface
        G, _ = graphs.iface7()
        exp = pd.DataFrame({                              # test_interface.py:124-148
            'degree': {0: 3, 1: 1, 2: 1, 3: 2, 4: 2, 5: 2, 6: 3},
            'internal_edges': {0: 3, 1: 1, 2: 1, 3: 2, 4: 3, 5: 3, 6: 4},
            'external_edges': {0: 1, 1: 2, 2: 2, 3: 4, 4: 1, 5: 1, 6: 1}})
        pd.testing.assert_frame_equal(NetworkxInterface(G).get_neighborhood_features(), exp)

    def test_neighborhood_features_directed_weighted(self):
        from graphrole_amd.graph.interface import NetworkxInterface
        G, _ = graphs.iface7_directed_weighted()
        exp = pd.DataFrame({                              # test_interface.py:150-186
            'in_degree': {0: 0.00, 1: 2.00, 2: 1.50, 3: 3.00, 4: 0.00, 5: 0.75, 6: 3.75},
            'out_degree': {0: 6.50, 1: 0.00, 2: 0.00, 3: 0.25, 4: 3.25, 5: 1.00, 6: 0.00},
            'total_degree': {0: 6.50, 1: 2.00, 2: 1.50, 3: 3.25, 4: 3.25, 5: 1.75, 6: 3.75},
            'internal_edges': {0: 6.50, 1: 0.00, 2: 0.00, 3: 0.25, 4: 4.25, 5: 1.00, 6: 0.00},
            'external_edges': {0: 0.25, 1: 0.00, 2: 0.00, 3: 0.00, 4: 0.00, 5: 0.00, 6: 0.00}})
        pd.testing.assert_frame_equal(NetworkxInterface(G).get_neighborhood_features(), exp)

    def test_neighborhood_features_with_attributes(self):
        from graphrole_amd.graph.interface import NetworkxInterface
        a1, a2 = 'attribute_attr1', 'attribute_attr2'
        full = pd.DataFrame({                             # test_interface.py:188-221
            'degree': {0: 3, 1: 1, 2: 1, 3: 2, 4: 2, 5: 2, 6: 3},
            a1: {0: 1.0, 1: 0.0, 2: 0.0, 3: 0.0, 4: 0.0, 5: 0.0, 6: 0.0},
            a2: {0: 0.0, 1: 1.0, 2: 2.0, 3: 3.0, 4: 4.0, 5: 5.0, 6: 6.0},
            'internal_edges': {0: 3, 1: 1, 2: 1, 3: 2, 4: 3, 5: 3, 6: 4},
            'external_edges': {0: 1, 1: 2, 2: 2, 3: 4, 4: 1, 5: 1, 6: 1}})
        plain, _ = graphs.iface7()
        attrs = graphs.iface7_attrs()
        table = [                                         # test_interface.py:223-322
            (attrs, dict(attributes=True), []),
            (attrs, dict(), [a1, a2]),
            (plain, dict(attributes=True), [a1, a2]),
            (attrs, dict(attributes=True, attributes_include=['attr1', 'attr2']), []),
            (attrs, dict(attributes=True, attributes_include=['attr1']), [a2]),
            (attrs, dict(attributes=True, attributes_exclude=['attr1', 'attr2']), [a1, a2]),
            (attrs, dict(attributes=True, attributes_exclude=['attr2']), [a2]),
            (attrs, dict(attributes=True, attributes_include=['attr1'], attributes_exclude=['attr2']), [a2]),
            (attrs, dict(attributes=True, attributes_include=['attr1', 'attr2'], attributes_exclude=['attr2']), [a2]),
            (attrs, dict(attributes=True, attributes_include=['attr2'], attributes_exclude=['attr2']), [a1, a2]),
        ]
        for G, kwargs, dropped in table:
            got = NetworkxInterface(G, **kwargs).get_neighborhood_features()
            pd.testing.assert_frame_equal(got, full.drop(dropped, axis=1), obj=str(kwargs))


class TestPrunerLikeReference:
    """tests/test_features/test_prune.py:107-225 of the reference."""

    def _pruner(self):
        from graphrole_amd.features.prune import FeaturePruner
        return FeaturePruner({0: {'b': {}, 'a': {}}, 1: {'c': {}, 'd': {}}}, 1)

    def test_vertical_log_binning_function(self):
        from graphrole_amd.features.prune import vertical_log_binning
        assert vertical_log_binning(np.array([])).tolist() == []
        assert vertical_log_binning(np.array(range(10))).tolist() == [0, 0, 0, 0, 0, 1, 1, 2, 3, 4]
        assert vertical_log_binning(pd.Series(range(10)), frac=0.25).tolist() == [0, 0, 1, 1, 2, 3, 4, 5, 6, 7]
        with pytest.raises(ValueError):
            vertical_log_binning(np.array([1.0]), frac=1.0)

    def test_prune_features(self):
        pruner = self._pruner()
        features = pd.DataFrame({'a': [1, 2, 3, 10], 'b': [1, 2, 3, 1], 'c': [2, 1, 1, 4],
                                 'd': [1, 1, 1, 1], 'e': [1, 1, 2, 0]})
        pruner._generation_dict = {0: {'a': {}, 'b': {}, 'c': {}}, 1: {'d': {}, 'e': {}}}
        for thresh, expected in [(0, []), (1, ['c', 'd', 'e']), (2, ['b', 'c', 'd', 'e'])]:
            pruner._feature_group_thresh = thresh
            assert set(pruner.prune_features(features)) == set(expected)

    def test_group_features(self):
        pruner = self._pruner()
        features = pd.DataFrame({'a': [1, 2, 3], 'b': [1, 2, 3], 'c': [2, 1, 1], 'd': [1, 1, 1]})
        table = [(0, [{'a', 'b'}]), (1, [{'a', 'b'}, {'c', 'd'}]), (2, [{'a', 'b', 'c', 'd'}]), (-1, [])]
        for thresh, expected in table:
            pruner._feature_group_thresh = thresh
            assert list(pruner._group_features(features)) == expected

    def test_get_oldest_feature(self):
        pruner = self._pruner()
        assert pruner._get_oldest_feature({'a', 'c', 'f'}) == 'a'
        assert pruner._get_oldest_feature({'a', 'b', 'f'}) == 'a'
        assert pruner._get_oldest_feature({'x', 'd', 'f', 'aa'}) == 'd'
        assert pruner._get_oldest_feature({'y', 'x', 'z'}) == 'x'

    def test_set_getitem(self):
        pruner = self._pruner()
        for _ in range(10):
            assert pruner._set_getitem({3, 2, 5, 6}) == 2
            assert pruner._set_getitem({'d', 'b', 'a', 'c'}) == 'a'


def test_large_powerlaw_matches_oracle_and_is_reproducible():
    """50k-node power-law graph: full pipeline vs the C oracle (retained sets exact, values util.WEIGHTED_RTOL)."""
    from graphrole_amd import RecursiveFeatureExtractor
    from graphrole_amd.graph import CSRGraph
    from oracle import refex
    n, m = 50000, 8
    src, dst, _ = util.powerlaw_graph(n, m, seed=11)
    fe = RecursiveFeatureExtractor(CSRGraph(n, src, dst), max_generations=4)
    X = fe.extract_features()
    og = refex.graph_from_arrays(n, src, dst)
    ref = refex.extract_features(og, max_generations=4, fast=True)
    assert list(X.columns) == ref.columns
    assert fe.generation_count == ref.generation_count
    for gen, tr in enumerate(ref.trace):
        assert fe._final_names[gen] == tr.retained
    np.testing.assert_allclose(X.values.astype(float), ref.values, rtol=RTOL, atol=0)
    X2 = RecursiveFeatureExtractor(CSRGraph(n, src, dst), max_generations=4).extract_features()
    assert np.array_equal(X.values, X2.values)                        # bitwise run-to-run


# ------------------------------------------------------------------ randomized sweep against the oracle
_FUZZ = [dict(n=n, m=m, seed=seed, directed=d, weighted=w, self_loops=sl)
         for seed, (n, m, d, w, sl) in enumerate([
             (2, 1, False, False, 0), (3, 2, True, False, 0), (9, 8, False, False, 2), (40, 39, False, False, 0),
             (60, 300, False, False, 5), (60, 300, True, False, 5), (60, 300, False, True, 3), (60, 300, True, True, 3),
             (500, 600, False, False, 0), (500, 5000, False, False, 20), (500, 5000, True, False, 20),
             (800, 9000, True, True, 10), (1500, 2000, False, False, 0), (1500, 40000, False, False, 0),
             (3000, 3100, True, False, 1), (2500, 60000, True, True, 0), (300, 20000, False, False, 0),
             (129, 128 * 64, False, False, 0), (4000, 5000, False, True, 7), (1000, 30000, False, False, 100)])]


@pytest.mark.parametrize('spec', _FUZZ, ids=[f"n{s['n']}_m{s['m']}_{'d' if s['directed'] else 'u'}{'w' if s['weighted'] else ''}" for s in _FUZZ])
@pytest.mark.parametrize('aggs', [['sum', 'mean'], ['max', 'sum', 'min'], ['std', 'mean', 'var']],
                         ids=['summean', 'maxsummin', 'stdmeanvar'])
def test_random_graphs_match_oracle(spec, aggs):
    """Whole pipeline on random graphs of every kind (sparse, dense, directed, weighted, self-loops,
    isolated nodes, rows longer than 128) against the oracle: column lists, per-generation retained
    lists and generation count exact; values bit-exact on unweighted graphs, util.WEIGHTED_RTOL on weighted."""
    from graphrole_amd import RecursiveFeatureExtractor
    from graphrole_amd.graph import CSRGraph
    from oracle import refex
    src, dst, w = util.random_graph(**spec)
    n = spec['n'] + 3                                              # three isolated nodes at the end
    G = CSRGraph(n, src, dst, weights=w, directed=spec['directed'])
    fe = RecursiveFeatureExtractor(G, max_generations=5, aggs=aggs)
    X = fe.extract_features()
    og = refex.graph_from_arrays(n, src, dst, w, spec['directed'])
    ref = refex.extract_features(og, max_generations=5, fast=True, aggs=aggs)
    assert list(X.columns) == ref.columns
    assert fe.generation_count == ref.generation_count
    for gen, tr in enumerate(ref.trace):
        assert fe._final_names[gen] == tr.retained, f'generation {gen}'
    got = X.values.astype(np.float64)
    if spec['weighted'] and ('std' in aggs or 'var' in aggs):
        # generation 0 of a weighted graph agrees to util.WEIGHTED_RTOL only, and a variance of nearly equal
        # values amplifies that by its cancellation: compare against the scale of each column
        scale = np.abs(ref.values).max(axis=0, keepdims=True) + 1e-300
        assert np.abs(got - ref.values).max() <= 1e-9 * scale.max()
        np.testing.assert_allclose(got / scale, ref.values / scale, rtol=0, atol=1e-7)
    elif spec['weighted']:
        np.testing.assert_allclose(got, ref.values, rtol=RTOL, atol=0)
    else:
        assert np.array_equal(got, ref.values), f'{int((got != ref.values).sum())} entries differ'


@pytest.mark.parametrize('native', [True, False], ids=['grx_refex_run', 'per_kernel'])
@pytest.mark.parametrize('name', util.TYPED_CASES)
def test_any_aggregation_list_matches_reference(name, native):
    """aggs with 'prod' over integer columns (the reference's WRAPPING int64 arithmetic, values far beyond 2^53 and
    negative), 'median', 'count' / 'size', mixed with the others -- against tables the reference produced with those
    aggs: columns, dtypes, generation count, integer columns bit for bit (int64), float columns exactly on unweighted
    graphs.  native: the loop below the ABI takes these aggregations too, except where int64 semantics are needed
    (then both parametrisations run the per-kernel driver)."""
    from graphrole_amd import RecursiveFeatureExtractor
    g = util.load_refex(name)
    G, kwargs = _graph_for(name, g)
    fe = RecursiveFeatureExtractor(G, aggs=util.golden_aggs(g), max_generations=int(g['max_generations']),
                                  native_loop=native, **kwargs)
    X = fe.extract_features()
    assert list(X.index) == g.js('labels')
    assert fe.generation_count == int(g['generation_count'])
    weighted = bool(len(g['w']))
    util.assert_typed_final_equal(g, list(X.columns), {c: X[c].to_numpy() for c in X.columns}, rtol=RTOL if weighted else 0.0)
    for gen in range(int(g['n_generations_recorded'])):
        assert fe._final_names[gen] == g.js(f'g{gen}_retained'), f'generation {gen}'
    pd.testing.assert_frame_equal(X, fe.extract_features())


def test_prod_on_integer_features_wraps_like_the_reference():
    """The reference multiplies int64 columns in int64 and wraps silently past 2^63 (numpy): so do the int64-bits
    columns here (csrc/grx_aggx.hip) -- on a graph where products overflow in the first generation."""
    from graphrole_amd import RecursiveFeatureExtractor
    from oracle import refex
    G, _ = graphs.BUILDERS['ba300']()
    fe = RecursiveFeatureExtractor(G, aggs=['sum', 'prod'], max_generations=3)
    X = fe.extract_features()
    assert all(dt == np.int64 for dt in X.dtypes)
    assert np.abs(X.to_numpy().astype(np.float64)).max() > 2.0 ** 60          # wrapped territory
    og = refex.graph_from_networkx(G)
    names0, block0 = refex.neighborhood_features(og)
    ref = refex.extract_features_typed(og, names0, [block0[:, j].astype(np.int64) for j in range(len(names0))], 3,
                                       ['sum', 'prod'])
    assert list(X.columns) == ref.columns and fe.generation_count == ref.generation_count
    for c in X.columns:
        assert np.array_equal(X[c].to_numpy(), ref.arrays[c]), c


@pytest.mark.parametrize('aggs', [['median', 'sum'], ['count', 'mean', 'max'], ['prod', 'median', 'size']],
                         ids=['mediansum', 'countmeanmax', 'prodmediansize'])
@pytest.mark.parametrize('spec', [dict(n=700, m=9000, seed=1, directed=False, weighted=False, self_loops=4),
                                  dict(n=500, m=3000, seed=2, directed=True, weighted=True, self_loops=3),
                                  dict(n=129, m=128 * 64, seed=3, directed=False, weighted=False, self_loops=0)],
                         ids=['u700', 'dw500', 'dense129'])
def test_any_aggregation_list_on_random_graphs_matches_typed_oracle(spec, aggs):
    """Random graphs (rows of 100+ neighbours: the radix-selection branch of the median; isolated nodes) against
    oracle.refex.extract_features_typed, itself pinned on the reference's tables for such aggs."""
    from graphrole_amd import RecursiveFeatureExtractor
    from graphrole_amd.graph import CSRGraph
    from oracle import refex
    src, dst, w = util.random_graph(**spec)
    n = spec['n'] + 2
    G = CSRGraph(n, src, dst, weights=w, directed=spec['directed'])
    fe = RecursiveFeatureExtractor(G, max_generations=3, aggs=aggs)
    X = fe.extract_features()
    og = refex.graph_from_arrays(n, src, dst, w, spec['directed'])
    names0, block0 = refex.neighborhood_features(og, fast=True)
    int0 = w is None
    cols0 = [block0[:, j].astype(np.int64) if int0 else block0[:, j] for j in range(len(names0))]
    ref = refex.extract_features_typed(og, names0, cols0, 3, aggs)
    assert list(X.columns) == ref.columns and fe.generation_count == ref.generation_count
    for c in X.columns:
        got, want = X[c].to_numpy(), ref.arrays[c]
        assert got.dtype == want.dtype, c
        if spec['weighted']:
            np.testing.assert_allclose(got, want, rtol=1e-9, atol=0, err_msg=c)
        else:
            assert np.array_equal(got, want), c


def test_prod_on_float_features_matches_oracle():
    from graphrole_amd import RecursiveFeatureExtractor
    from graphrole_amd.graph import CSRGraph
    from oracle import refex
    rng = np.random.default_rng(12)
    n, m = 400, 1600
    src, dst = rng.integers(0, n, m), rng.integers(0, n, m)
    keep = src != dst
    key = np.minimum(src, dst) * n + np.maximum(src, dst)
    _, first = np.unique(key, return_index=True)
    sel = np.intersect1d(first, np.flatnonzero(keep))
    src, dst = src[sel], dst[sel]
    w = rng.uniform(0.05, 0.3, len(src))                       # weighted degrees around 1: products stay finite
    G = CSRGraph(n, src, dst, weights=w)
    aggs = ['prod', 'mean']
    fe = RecursiveFeatureExtractor(G, max_generations=3, aggs=aggs)
    X = fe.extract_features()
    og = refex.graph_from_arrays(n, src, dst, w, False)
    ref = refex.extract_features(og, max_generations=3, fast=True, aggs=aggs)
    assert list(X.columns) == ref.columns and any('(prod)' in c for c in X.columns)
    np.testing.assert_allclose(X.values.astype(float), ref.values, rtol=1e-9, atol=0)


def test_arena_grows_on_demand_and_chunks_are_reused(monkeypatch):
    """Round 5: the arena of grx_refex_run is a chain of chunks -- a first block far too small makes the library ask
    for more through the grow callback (no repeated run), the table equals the one from a roomy arena, and a second
    run of the same extractor takes the chunks it already has."""
    from graphrole_amd import RecursiveFeatureExtractor, kernels, synth
    G = synth.ba_graph(400_000, 5, seed=3)        # ~0.3 GB of columns and workspace: several 64 MB chunks
    ref = RecursiveFeatureExtractor(G, max_generations=4).extract_features()
    monkeypatch.setattr(kernels, '_refex_arena_guess', lambda *a, **k: 1 << 16)
    fe = RecursiveFeatureExtractor(G, max_generations=4)
    X = fe.extract_features()
    assert kernels.refex_run.attempts == 1
    grown = [what for what, _ in kernels.refex_run.trace if what.startswith('grow')]
    assert len(grown) >= 2 and len(fe._arena) == 1 + len(grown)
    pd.testing.assert_frame_equal(X, ref, check_exact=True)
    held = sorted(c.data_ptr() for c in fe._arena)
    fe.reset()
    X2 = fe.extract_features()
    pd.testing.assert_frame_equal(X2, ref, check_exact=True)
    assert not [what for what, _ in kernels.refex_run.trace if what.startswith('arena')]
    assert set(held) <= {c.data_ptr() for c in fe._arena}
