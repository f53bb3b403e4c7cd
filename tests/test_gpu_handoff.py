"""
-m gpu: the host <-> device boundary of the two public calls (csrc/grx_hostio.hip, features/handoff.py).
extract_features() hands out a DataFrame AND remembers the device block behind it; extract_role_factors() on the
unmodified table reuses that block, on anything else it uploads -- either way the factors are those of the values
the caller passed (reference: graphrole/roles/extract.py:59-93).
"""
import numpy as np
import pandas as pd
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def table():
    from graphrole_amd import RecursiveFeatureExtractor, synth
    G = synth.ba_graph(30_000, 6, seed=5)
    fe = RecursiveFeatureExtractor(G, max_generations=4)
    return fe, fe.extract_features()


def _fit(X, n_roles=4):
    from graphrole_amd import RoleExtractor
    np.random.seed(3)
    rx = RoleExtractor(n_roles=n_roles)
    rx.extract_role_factors(X)
    return rx.node_role_factor, rx.role_feature_factor


def test_frame_has_the_reference_dtypes_and_wraps_the_block(table):
    fe, X = table
    assert X.dtypes['degree'] == np.int64 and X.dtypes['internal_edges'] == np.int64
    assert all(dt == np.float64 for nm, dt in X.dtypes.items() if '(' in nm)
    assert list(X.columns) == fe.final_columns()
    # memoised (extract.py:70-71): the second call returns an equal table
    assert fe.extract_features().equals(X)


def test_unmodified_table_is_handed_over_in_hbm_and_gives_the_upload_results(table):
    from graphrole_amd import kernels as K
    from graphrole_amd.features import handoff
    _, X = table
    block = handoff.lookup(K, X)
    assert block is not None and tuple(block.shape) == (X.shape[1], X.shape[0])
    assert np.array_equal(K.to_host(block).T, X.to_numpy(dtype=np.float64))
    G1, F1 = _fit(X)                                   # hand-off
    Xc = X.copy()
    assert handoff.lookup(K, Xc) is None               # a copy is another object: ordinary upload path
    G2, F2 = _fit(Xc)
    assert np.array_equal(G1.values, G2.values) and np.array_equal(F1.values, F2.values)
    assert list(G1.index) == list(X.index) and list(F1.columns) == list(X.columns)


@pytest.mark.parametrize('edit', ['cell', 'int_cell', 'column', 'swap_rows', 'rename_only'])
def test_any_edit_of_the_table_is_seen(edit):
    from graphrole_amd import RecursiveFeatureExtractor, kernels as K, synth
    from graphrole_amd.features import handoff
    X = RecursiveFeatureExtractor(synth.ba_graph(20_000, 5, seed=6), max_generations=3).extract_features()
    assert handoff.lookup(K, X) is not None
    if edit == 'cell':
        X.iloc[17, 1] = X.iloc[17, 1] + 0.5
    elif edit == 'int_cell':
        X.loc[X.index[3], 'degree'] += 1
    elif edit == 'column':
        X[X.columns[0]] = X[X.columns[0]] * 2.0
    elif edit == 'swap_rows':
        v = X.iloc[:, 2].to_numpy()
        v[[10, 11]] = v[[11, 10]]
    elif edit == 'rename_only':
        X.columns = [f'c{j}' for j in range(X.shape[1])]           # values untouched: still a hit
        assert handoff.lookup(K, X) is not None
        return
    assert handoff.lookup(K, X) is None
    # and the factors are those of the edited values
    G1, F1 = _fit(X, 3)
    G2, F2 = _fit(pd.DataFrame(X.to_numpy(dtype=np.float64), index=X.index, columns=X.columns), 3)
    assert np.array_equal(G1.values, G2.values) and np.array_equal(F1.values, F2.values)


def test_negative_and_nan_tables_raise_like_sklearn(table):
    from graphrole_amd import RoleExtractor
    _, X = table
    bad = X.copy().astype(np.float64)
    bad.iloc[5, 0] = -1.0
    with pytest.raises(ValueError, match='Negative values'):
        RoleExtractor(n_roles=3).extract_role_factors(bad)
    bad.iloc[5, 0] = np.nan
    with pytest.raises(ValueError):
        RoleExtractor(n_roles=3).extract_role_factors(bad)


@pytest.mark.parametrize('count', [1, 1000, (1 << 17) + 3, (8 << 20) // 8, 5 * (8 << 20) // 8 + 12345])
def test_bulk_copies_round_trip(count):
    from graphrole_amd import kernels as K
    rng = np.random.default_rng(count)
    a = rng.random(count)
    d = K.to_device(a)
    assert np.array_equal(K.to_host(d), a)
    e = rng.integers(0, 2 ** 31 - 1, size=count, dtype=np.int64)
    assert np.array_equal(K.to_host(K.edges_to_device(e)), e.astype(np.int32))


def test_edge_ids_beyond_32_bits_are_refused():
    from graphrole_amd import _lib, kernels as K
    e = np.arange(1 << 18, dtype=np.int64)
    e[77] = 1 << 31
    with pytest.raises(_lib.GrxInvalid):
        K.edges_to_device(e)
