"""
-m "not gpu": the host logic of the drop-in classes (naming, ordering, pruning decisions, dtype
handling, NNDSVDa small-space algebra, convergence rule, error behaviour) exercised WITHOUT a
GPU by injecting the CPU test double tests/fake_kernels.py (built on oracle/) as the kernel
backend.  The same test bodies run against the real HIP backend in test_gpu_refex.py /
test_gpu_rolx.py (-m gpu); nothing here measures or ships the double.
"""
import pytest

from graphrole_amd import backend
from tests import fake_kernels
from tests import test_gpu_refex as R
from tests import test_gpu_rolx as X


@pytest.fixture(autouse=True)
def _cpu_double():
    backend.use(fake_kernels)
    yield
    backend.use(None)


# ReFeX host logic
test_extract_features_matches_reference = R.test_extract_features_matches_reference
test_generation_trace_matches_reference = R.test_generation_trace_matches_reference
test_csr_graph_input_equals_networkx_input = R.test_csr_graph_input_equals_networkx_input
test_prod_on_integer_features_wraps_like_the_reference = R.test_prod_on_integer_features_wraps_like_the_reference
test_any_aggregation_list_matches_reference = R.test_any_aggregation_list_matches_reference
test_prod_on_float_features_matches_oracle = R.test_prod_on_float_features_matches_oracle


class TestExtractorLikeReference(R.TestExtractorLikeReference):
    pass


class TestInterfaceLikeReference(R.TestInterfaceLikeReference):
    pass


class TestPrunerLikeReference(R.TestPrunerLikeReference):
    pass


# RolX host logic
test_nmf_matches_reference_golden = X.test_nmf_matches_reference_golden
test_nmf_consumes_global_rng_like_sklearn = X.test_nmf_consumes_global_rng_like_sklearn
test_nmf_rejects_negative_input = X.test_nmf_rejects_negative_input
test_nmf_rank_deficient_features = X.test_nmf_rank_deficient_features
test_end_to_end_karate_roles = X.test_end_to_end_karate_roles


class TestFactorLikeReference(X.TestFactorLikeReference):
    pass


class TestDescriptionLengthLikeReference(X.TestDescriptionLengthLikeReference):
    pass


class TestRoleExtractorLikeReference(X.TestRoleExtractorLikeReference):
    pass
