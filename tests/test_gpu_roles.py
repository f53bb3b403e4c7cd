"""
-m gpu: RoleExtractor.roles / role_percentage (graphrole/roles/extract.py:38-57) on the device
(csrc/grx_roles.hip), pinned on what the REFERENCE returned: tests/golden/roles_*.npz / roles_wide.npz hold,
for every fitted factor, the reference's ``roles`` dict (as label positions) and its ``role_percentage`` table
(tools/make_golden_roles.py).  The kernels are first fed the reference's own factor -- ``roles`` must be equal
and ``role_percentage`` bit-equal -- then the whole drop-in call chain is compared.
"""
import json

import numpy as np
import pandas as pd
import pytest

from tests import util
from tests.test_roles_pinned_cpu import _cases, wide_cases

pytestmark = pytest.mark.gpu


def _device_answers(G):
    from graphrole_amd import kernels as K
    Gd = K.to_device(np.ascontiguousarray(G, dtype=np.float64))
    return K.to_host(K.role_argmax(Gd)), K.to_host(K.row_normalise(Gd))


@pytest.mark.parametrize('name,factor_key,roles_key,share_key', _cases())
def test_kernels_on_the_reference_factor(name, factor_key, roles_key, share_key):
    ref = util.load_roles(name)
    first, share = _device_answers(ref[factor_key])
    assert np.array_equal(first, ref[roles_key])
    assert np.array_equal(share, ref[share_key], equal_nan=True)                     # bit-equal


def test_kernels_on_wide_reference_factors():
    for key, z in wide_cases():
        first, share = _device_answers(z[f'{key}_node_role_factor'])
        assert np.array_equal(first, z[f'{key}_roles_index']), key
        assert np.array_equal(share, z[f'{key}_role_percentage'], equal_nan=True), key


@pytest.mark.parametrize('n,r', [(n, r) for r in (1, 2, 3, 5, 7, 8, 9, 15, 16, 17, 24, 31, 32) for n in (1, 127, 128, 129, 70001)] + [
                                 # wider than any fitted factor (a frame the caller assigned): fewer rows per LDS tile,
                                 # numpy's pairwise recursion beyond 128 values, rows that do not fit LDS at all
                                 (5000, 33), (3001, 59), (3001, 60), (2000, 128), (2000, 129), (700, 300), (300, 1000),
                                 (40, 7679), (40, 7680), (9, 8193), (9, 20000)])
def test_kernels_equal_the_oracle(n, r):
    """ranks 1 .. GRX_MAX_ROLES and beyond, ragged row counts around the 128-row tile; quantised-like values (many exact
    ties), all-zero rows (0 / 0 = NaN), NaN entries and all-NaN rows"""
    from oracle import rolx
    rng = np.random.RandomState(1000 * r + n % 997)
    levels = np.sort(rng.gamma(0.7, 2.0, 6))
    G = levels[rng.randint(0, 6, size=(n, r))]
    G[rng.rand(n) < 0.05] = 0.0
    G[rng.rand(n, r) < 0.01] = np.nan
    if n > 3:
        G[3] = np.nan
    first, share = _device_answers(G)
    assert np.array_equal(first, rolx.dominant_role_index(G))
    assert np.array_equal(share, rolx.role_percentage(G), equal_nan=True)


def test_kernels_take_empty_factors_and_user_assigned_wide_frames():
    from graphrole_amd import RoleExtractor, kernels as K
    assert K.to_host(K.role_argmax(K.to_device(np.ones((0, 4))))).shape == (0,)
    # the reference's properties are idxmax / apply on ANY frame (roles/extract.py:38-57): a 40-column frame assigned by
    # the caller works although no fit produces more than GRX_MAX_ROLES roles
    rng = np.random.RandomState(3)
    frame = pd.DataFrame(rng.rand(50, 40), index=[f'n{i}' for i in range(50)], columns=[f'role_{i}' for i in range(40)])
    rx = RoleExtractor(n_roles=2)
    rx.node_role_factor = frame
    assert rx.roles == frame.idxmax(axis=1).to_dict()
    expect = frame.apply(lambda row: row / row.sum(), axis=1)
    assert np.array_equal(rx.role_percentage.values, expect.values)


@pytest.mark.parametrize('name', util.ROLES_CASES)
def test_fixed_rank_roles_equal_reference(name):
    """RoleExtractor(n_roles=3).roles / .role_percentage end to end (HIP NMF + quantiser + row kernels) against the
    reference's properties for the same table and seed"""
    from graphrole_amd import RoleExtractor
    ref = util.load_roles(name)
    g = util.load_refex(name)
    X = pd.DataFrame(g['final_values'], index=g.js('labels'), columns=g.js('final_columns'))
    np.random.seed(int(ref['seed']))
    rx = RoleExtractor(n_roles=3)
    rx.extract_role_factors(X)
    roles = rx.roles
    labels = list(rx.node_role_factor.columns)
    assert list(roles) == list(X.index)
    assert [labels.index(roles[node]) for node in X.index] == ref['fixed3_roles_index'].tolist()
    assert rx.dominant_role_index().tolist() == ref['fixed3_roles_index'].tolist()
    share = rx.role_percentage
    assert list(share.index) == list(X.index) and list(share.columns) == labels
    # our factor equals the reference's to ~1e-14 (test_fixed_rank_role_factors_equal_reference): so do the shares
    assert np.allclose(share.values, ref['fixed3_role_percentage'], rtol=1e-11, atol=0, equal_nan=True)
    # and they are bit-equal to the reference's expression evaluated on OUR factor
    expect = rx.node_role_factor.apply(lambda row: row / row.sum(), axis=1)
    assert np.array_equal(share.values, expect.values, equal_nan=True)
    assert roles == rx.node_role_factor.idxmax(axis=1).to_dict()


def test_roles_at_a_million_rows_are_fast():
    """1 M x 6 factor: device part in milliseconds (the dict of ``roles`` is the reference's return type and costs
    what a million-entry Python dict costs)"""
    import time
    import torch
    from graphrole_amd import RoleExtractor
    rng = np.random.RandomState(0)
    levels = np.sort(rng.gamma(0.7, 2.0, 64))
    G = levels[rng.randint(0, 64, size=(1_000_000, 6))]
    rx = RoleExtractor(n_roles=6)
    rx.node_role_factor = pd.DataFrame(G, columns=[f'role_{i}' for i in range(6)])
    rx.dominant_role_index()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    first = rx.dominant_role_index()
    t1 = time.perf_counter()
    share = rx.role_percentage
    t2 = time.perf_counter()
    assert np.array_equal(first, np.argmax(G, axis=1))
    assert np.array_equal(share.values, G / G.sum(axis=1, keepdims=True))           # r < 8: left-to-right sums
    assert t1 - t0 < 0.1 and t2 - t1 < 0.25, (t1 - t0, t2 - t1)
