"""-m gpu: differential fuzz of RecursiveFeatureExtractor against the oracle on random small graphs
(tools/fuzz_refex.py): sizes 5..6000, sparse to dense, hubs with more than 128 neighbours, directed / integer
weights / self-loops / isolated nodes, three aggregation sets, 2..5 generations -- columns and values bit for bit."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('seed', [11, 12])
def test_refex_random_graphs_equal_oracle(seed):
    from tools import fuzz_refex
    rng = np.random.default_rng(seed)
    done = sum(fuzz_refex.one(rng, case) != 'skip' for case in range(40))
    assert done >= 30
