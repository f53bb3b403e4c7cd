#!/usr/bin/env python3
"""
tools/make_golden_roles.py -- model-selection fixtures (tests/golden/roles_<name>.npz) by RUNNING THE
REFERENCE's RoleExtractor in the build container on feature tables that tools/make_golden.py already captured
(the final ReFeX tables of tests/golden/refex_<name>.npz).

    PYTHONDONTWRITEBYTECODE=1 python tools/make_golden_roles.py

Per table: the MDL grid of RoleExtractor._select_model (graphrole/roles/extract.py:98-142) -- encoding and
error costs of every (n_roles, n_bits) cell as the reference computes them (its quantiser is sklearn
KMeans(random_state=1), graphrole/roles/factor.py:41-48) with numpy's global RNG seeded once before the grid --
the selected cell, the selected factors, and the fixed-rank result for n_roles = 3; for both fits the reference's
``roles`` dict (stored as the position of every node's label in the factor's columns) and its ``role_percentage``
table (graphrole/roles/extract.py:38-57).  tests/golden/roles_wide.npz adds fits with 9 and 12 roles (rows of 8
and more values take the other branch of numpy's pairwise sum).
The reference is imported here and only here; the fixtures are data.
"""
import json
import os
import sys
import warnings

import numpy as np

REF = '/root/reference'
sys.dont_write_bytecode = True
sys.path.insert(0, REF)
warnings.simplefilter('ignore')

import pandas as pd                                              # noqa: E402
from graphrole import RoleExtractor                              # noqa: E402
from graphrole.roles.description_length import get_description_length_costs   # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, 'tests', 'golden')
WIDE = [('er2000', 12), ('dw200_attrs', 9), ('er300', 8)]
CASES = ['karate', 'karate_weighted', 'er300', 'ba300', 'dw200_attrs', 'directed120', 'loops_dangling150']
SEED = 0


def load_table(name):
    z = np.load(os.path.join(OUT, f'refex_{name}.npz'))
    cols = json.loads(str(z['final_columns_json']))
    labels = json.loads(str(z['labels_json']))
    return pd.DataFrame(z['final_values'], index=labels, columns=cols)


def roles_of(rx):
    """(position of every node's role label in node_role_factor.columns, role_percentage values), rows in the
    factor's index order -- from the reference's own properties"""
    frame = rx.node_role_factor
    roles = rx.roles
    pos = {label: i for i, label in enumerate(frame.columns)}
    index = np.array([pos[roles[node]] for node in frame.index], dtype=np.int32)
    share = rx.role_percentage
    assert list(share.index) == list(frame.index) and list(share.columns) == list(frame.columns)
    return index, np.ascontiguousarray(share.values, dtype=np.float64)


def environment():
    """what the tied cells of the MDL grid depend on: sklearn's k-means++ breaks mathematical ties by the last-bit
    rounding of its BLAS dot products -- recorded with every fixture so that a changed allow-list
    (tests/test_gpu_rolx.py::TIED_CELLS) can be traced to a changed build"""
    import numpy
    import pandas
    import scipy
    import sklearn
    from threadpoolctl import threadpool_info
    pools = [{k: p.get(k) for k in ('user_api', 'internal_api', 'version', 'num_threads', 'threading_layer', 'architecture')}
             for p in threadpool_info()]
    return json.dumps({'sklearn': sklearn.__version__, 'numpy': numpy.__version__, 'scipy': scipy.__version__,
                       'pandas': pandas.__version__, 'python': sys.version.split()[0], 'threadpools': pools})


def main():
    wide = {}
    for name, r in WIDE:
        X = load_table(name)
        np.random.seed(SEED)
        rx = RoleExtractor(n_roles=r)
        rx.extract_role_factors(X)
        idx, share = roles_of(rx)
        wide[f'{name}_r{r}_node_role_factor'] = rx.node_role_factor.values
        wide[f'{name}_r{r}_role_feature_factor'] = rx.role_feature_factor.values
        wide[f'{name}_r{r}_roles_index'] = idx
        wide[f'{name}_r{r}_role_percentage'] = share
        print(f'roles_wide {name} r={r}: ties in {int((np.sort(rx.node_role_factor.values, axis=1)[:, -1] == np.sort(rx.node_role_factor.values, axis=1)[:, -2]).sum())} of {len(idx)} rows')
    np.savez_compressed(os.path.join(OUT, 'roles_wide.npz'), seed=SEED, environment_json=environment(), cases=json.dumps([[n, r] for n, r in WIDE]), **wide)
    for name in CASES:
        X = load_table(name)
        # the grid exactly as RoleExtractor._select_model walks it (one RNG stream for the whole grid)
        rx = RoleExtractor()
        bit_stop = rx.max_bits + 1
        role_stop = min(min(X.shape), rx.max_roles) + 1
        enc = np.full((role_stop, bit_stop), np.nan)
        err = np.full((role_stop, bit_stop), np.nan)
        np.random.seed(SEED)
        for roles in range(rx.min_roles, role_stop):
            for bits in range(rx.min_bits, bit_stop):
                try:
                    model = rx._get_encoded_role_factors(X, roles, bits)
                    e, r = get_description_length_costs(X, model)
                except ValueError:
                    continue
                enc[roles, bits], err[roles, bits] = e, r
        costs = rx._rescale_costs(enc) + rx._rescale_costs(err)
        sel = np.argwhere(costs == np.nanmin(costs))[0]
        # the public call must pick the same cell
        np.random.seed(SEED)
        rx2 = RoleExtractor()
        rx2.extract_role_factors(X)
        assert rx2.node_role_factor.shape[1] == int(sel[0]), (name, rx2.node_role_factor.shape, sel)
        # fixed rank
        np.random.seed(SEED)
        rx3 = RoleExtractor(n_roles=3)
        rx3.extract_role_factors(X)
        old = np.load(os.path.join(OUT, f'roles_{name}.npz')) if os.path.exists(os.path.join(OUT, f'roles_{name}.npz')) else None
        if old is not None:                       # a regenerated fixture must reproduce what was pinned before
            # (to rounding: sklearn's threaded KMeans reductions move the centres by an ulp or two from run to run)
            assert np.allclose(old['node_role_factor'], rx2.node_role_factor.values, rtol=1e-12, atol=0), name
            assert np.allclose(old['fixed3_node_role_factor'], rx3.node_role_factor.values, rtol=1e-12, atol=0), name
        sel_idx, sel_share = roles_of(rx2)
        fix_idx, fix_share = roles_of(rx3)
        np.savez_compressed(
            os.path.join(OUT, f'roles_{name}.npz'), seed=SEED, environment_json=environment(), roles_index=sel_idx, role_percentage=sel_share,
            fixed3_roles_index=fix_idx, fixed3_role_percentage=fix_share, encoding_costs=enc, error_costs=err,
            selected=np.array(sel, dtype=np.int64), node_role_factor=rx2.node_role_factor.values,
            role_feature_factor=rx2.role_feature_factor.values, fixed3_node_role_factor=rx3.node_role_factor.values,
            fixed3_role_feature_factor=rx3.role_feature_factor.values,
            fixed3_n_bits=int(np.log2(3 * min(X.shape))))
        print(f'roles_{name}: table {X.shape} -> selected (n_roles, n_bits) = {tuple(int(v) for v in sel)}')


if __name__ == '__main__':
    main()
