"""-m gpu: the igraph adapter's loop / parallel-edge conventions on the real kernels (the neighbour multiset goes
through grx_aggregate, the edge-dict graph through the ego-net kernels) against oracle/igraph_path.py."""
import numpy as np
import pytest

from tests.test_igraph_adapter_cpu import _random_multigraph, _stand_in_graph_class

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('directed', [False, True])
@pytest.mark.parametrize('weights', [None, 'int'])
def test_igraph_multigraph_on_device(directed, weights):
    from graphrole_amd import RecursiveFeatureExtractor
    from oracle import igraph_path
    rng = np.random.default_rng(5 + directed + (0 if weights is None else 2))
    n = 300
    edges = _random_multigraph(rng, n, 1500, directed, True, True)
    edges += [(0, j) for j in range(1, 200)]                       # a hub: more than 128 (multiset) neighbours
    w = None if weights is None else [int(x) for x in rng.integers(1, 6, len(edges))]
    Graph = _stand_in_graph_class()
    fe = RecursiveFeatureExtractor(Graph(n, edges, directed, w), max_generations=4)
    X = fe.extract_features()
    ref = igraph_path.extract_features(n, edges, directed, w, max_generations=4)
    assert list(X.columns) == ref.columns
    assert fe.generation_count == ref.generation_count
    assert np.array_equal(X.values.astype(float), ref.values)
