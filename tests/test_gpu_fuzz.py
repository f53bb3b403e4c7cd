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


@pytest.mark.parametrize('seed', [21, 22])
def test_rolx_random_tables_equal_oracle(seed):
    """tools/fuzz_rolx.py: random non-negative tables (F = 2..140, r = 2..8, dense / graded / sparse / rank-deficient):
    the GPU factorisation stops at the oracle's iteration and gives its factors to 1e-7."""
    from tools import fuzz_rolx
    rng = np.random.default_rng(seed)
    for case in range(25):
        fuzz_rolx.one(rng, case)
