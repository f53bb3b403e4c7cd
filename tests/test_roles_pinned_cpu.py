"""
-m "not gpu": the oracle's restatement of RoleExtractor.roles / role_percentage
(graphrole/roles/extract.py:38-57) against what the REFERENCE returned for its own fitted factors
(tests/golden/roles_*.npz, roles_wide.npz -- tools/make_golden_roles.py), and the drop-in properties driven
through the CPU test double.
"""
import json

import numpy as np
import pandas as pd
import pytest

from tests import util


def _cases():
    out = []
    for name in util.ROLES_CASES:
        out.append((name, 'node_role_factor', 'roles_index', 'role_percentage'))
        out.append((name, 'fixed3_node_role_factor', 'fixed3_roles_index', 'fixed3_role_percentage'))
    return out


def wide_cases():
    z = np.load(util.golden_path('roles_wide.npz'))
    return [(f'{name}_r{r}', z) for name, r in json.loads(str(z['cases']))]


@pytest.mark.parametrize('name,factor_key,roles_key,share_key', _cases())
def test_oracle_roles_equal_reference(name, factor_key, roles_key, share_key):
    from oracle import rolx
    ref = util.load_roles(name)
    G = ref[factor_key]
    assert np.array_equal(rolx.dominant_role_index(G), ref[roles_key])
    assert np.array_equal(rolx.role_percentage(G), ref[share_key], equal_nan=True)


def test_oracle_roles_equal_reference_on_wide_factors():
    from oracle import rolx
    for key, z in wide_cases():
        G = z[f'{key}_node_role_factor']
        assert G.shape[1] >= 8                                    # the eight-accumulator branch of numpy's pairwise sum
        assert np.array_equal(rolx.dominant_role_index(G), z[f'{key}_roles_index'])
        assert np.array_equal(rolx.role_percentage(G), z[f'{key}_role_percentage'], equal_nan=True)


def test_goldens_hold_ties():
    """quantised factors are full of exact ties: the first-maximum rule decides real rows of the fixtures"""
    tied = 0
    for name in util.ROLES_CASES:
        G = np.sort(util.load_roles(name)['fixed3_node_role_factor'], axis=1)
        tied += int((G[:, -1] == G[:, -2]).sum())
    assert tied >= 20, tied


def test_properties_through_the_test_double():
    """roles / role_percentage / dominant_role_index of the drop-in class: labels, index, None before fitting, NaN
    handling -- the same answers pandas gives for the reference's expressions"""
    from graphrole_amd import RoleExtractor, backend
    from tests import fake_kernels
    backend.use(fake_kernels)
    try:
        rx = RoleExtractor(n_roles=2)
        assert rx.roles is None and rx.role_percentage is None and rx.dominant_role_index() is None
        G = np.array([[1.0, 1.0, 0.5], [0.0, 0.0, 0.0], [0.25, np.nan, 0.75], [3.0, 1.0, 3.0]])
        frame = pd.DataFrame(G, index=['d', 'a', 'c', 'b'], columns=['role_0', 'role_1', 'role_2'])
        rx.node_role_factor = frame
        assert rx.roles == frame.idxmax(axis=1).to_dict() == {'d': 'role_0', 'a': 'role_0', 'c': 'role_2', 'b': 'role_0'}
        expect = frame.apply(lambda row: row / row.sum(), axis=1)
        got = rx.role_percentage
        assert list(got.index) == list(expect.index) and list(got.columns) == list(expect.columns)
        assert np.array_equal(got.values, expect.values, equal_nan=True)
        assert rx.dominant_role_index().tolist() == [0, 0, 2, 0]
        # integer node labels come back as Python ints, like Series.to_dict()
        rx.node_role_factor = pd.DataFrame(G, index=np.arange(4) * 10, columns=['role_0', 'role_1', 'role_2'])
        assert all(type(k) is int for k in rx.roles)
        # a frame wider than any fit (the reference's properties are idxmax / apply on ANY frame): no limit (round 5)
        wide = pd.DataFrame(np.arange(120.0).reshape(3, 40) % 7)
        rx.node_role_factor = wide
        assert rx.roles == wide.idxmax(axis=1).to_dict()
        assert np.array_equal(rx.role_percentage.values, wide.apply(lambda row: row / row.sum(), axis=1).values)
    finally:
        backend.use(None)
