"""-m gpu: networkx MultiGraph / MultiDiGraph inputs on the real kernels against the reference's own output
(tests/golden/multigraph_*.npz, see tests/test_multigraph_cpu.py)."""
import pytest

from tests.test_multigraph_cpu import CASES, check

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('name', CASES)
def test_multigraph_equals_reference_on_device(name):
    check(name)
