"""
-m gpu: user-callable aggregations on the HIP backend (kernel-backed entries of the list on the device, callables
evaluated by pandas over device-gathered neighbour rows) against the reference's tables for the same functions
(tests/golden/refex_callable_*.npz).
"""
import pytest

from tests import graphs as G_
from tests.test_callable_aggs_cpu import build_extractor, check_against_golden

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('name', list(G_.CALLABLE_CASES))
def test_callable_aggs_equal_reference(name):
    g, fe = build_extractor(name)
    check_against_golden(g, fe)


def test_callables_refuse_the_sharded_path():
    import networkx as nx
    from graphrole_amd import RecursiveFeatureExtractor
    fe = RecursiveFeatureExtractor(nx.path_graph(8), aggs=[G_.spread], distributed=None)
    assert fe.extract_features().shape[0] == 8
