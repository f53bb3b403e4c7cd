"""
-m "not gpu": the pruner inside grx_refex_run (C++: feature graph, components, oldest member by recorded generation
then by name -- csrc/grx_refex.hip::prune, reached through the host-only entry point grx_host_prune) against the
Python FeaturePruner, on the reference's known-answer tables (graphrole tests/test_features/test_prune.py:119-208),
on the generation traces the reference itself produced (tests/golden/refex_*.npz) and on random feature graphs.
"""
import ctypes

import numpy as np
import pytest

from graphrole_amd import _lib
from graphrole_amd.features.prune import FeaturePruner
from tests import util


def native_drop(names, dist, thresh, generation_dict):
    F = len(names)
    rec = np.full(F, -1, dtype=np.int32)
    for gen in sorted(generation_dict):
        for j, nm in enumerate(names):
            if nm in generation_dict[gen] and rec[j] < 0:
                rec[j] = gen
    c_names = (ctypes.c_char_p * max(F, 1))(*[nm.encode() for nm in names])
    d = np.ascontiguousarray(dist, dtype=np.int32)
    drop = np.zeros(max(F, 1), dtype=np.int32)
    _lib.call('grx_host_prune', F, c_names, rec.ctypes.data_as(ctypes.c_void_p), len(generation_dict),
              d.ctypes.data_as(ctypes.c_void_p), int(thresh), drop.ctypes.data_as(ctypes.c_void_p))
    return sorted(nm for j, nm in enumerate(names) if drop[j])


def python_drop(names, dist, thresh, generation_dict):
    return sorted(FeaturePruner(generation_dict, thresh).prune_from_distances(list(names), np.asarray(dist)))


def test_known_answer_groups_of_the_reference():
    # test_prune.py:119-147: columns a..e binned; a ~ b ~ c within 1, d ~ e within 1; a, d recorded in generation 0
    names = ['a', 'b', 'c', 'd', 'e']
    dist = np.full((5, 5), 9, dtype=np.int32)
    for p, q in ((0, 1), (1, 2), (3, 4)):
        dist[p, q] = dist[q, p] = 1
    gens = {0: {'a', 'd'}, 1: {'b'}}
    assert native_drop(names, dist, 1, gens) == python_drop(names, dist, 1, gens) == ['b', 'c', 'e']
    # nobody recorded: the smallest name of each group survives
    assert native_drop(names, dist, 1, {}) == python_drop(names, dist, 1, {}) == ['b', 'c', 'e']
    # threshold 0: no edges, nothing dropped
    assert native_drop(names, dist, 0, gens) == []
    # the oldest member wins over a smaller name
    gens = {0: {'c'}, 1: {'a'}}
    assert native_drop(names, dist, 1, gens) == python_drop(names, dist, 1, gens) == ['a', 'b', 'e']


@pytest.mark.parametrize('case', ['karate', 'er300', 'ba300', 'dw200_attrs', 'directed120', 'ba300_stdvar', 'dw200_minmax'])
def test_generation_traces_of_the_reference(case):
    """Every generation of a golden run: working set, Chebyshev matrix and recorded generations as the reference had
    them -> the reference's drop list."""
    g = util.load_refex(case)
    recorded = {}
    gen = 0
    while f'g{gen}_cheb' in g:
        names = g.js(f'g{gen}_working_before')
        dist = np.asarray(g[f'g{gen}_cheb'])
        want = sorted(g.js(f'g{gen}_dropped'))
        assert native_drop(names, dist, gen, recorded) == python_drop(names, dist, gen, recorded) == want
        recorded[gen] = set(g.js(f'g{gen}_retained'))
        gen += 1
    assert gen >= 2


@pytest.mark.parametrize('seed', range(40))
def test_random_feature_graphs(seed):
    rng = np.random.default_rng(seed)
    F = int(rng.integers(1, 40))
    pool = [f'{base}({agg})' * int(rng.integers(1, 3)) for base in ('degree', 'x', 'internal_edges', 'a(sum)')
            for agg in ('sum', 'mean', 'max')] + [f'f{j}' for j in range(60)]
    names = list(rng.choice(pool, size=F, replace=False))
    dist = rng.integers(0, 6, size=(F, F)).astype(np.int32)
    dist = np.minimum(dist, dist.T)
    np.fill_diagonal(dist, 0)
    n_gen = int(rng.integers(0, 4))
    gens = {}
    free = list(names)
    for gen in range(n_gen):
        take = [free.pop(int(rng.integers(len(free)))) for _ in range(min(len(free), int(rng.integers(0, 5))))]
        gens[gen] = set(take)
    thresh = int(rng.integers(0, 4))
    assert native_drop(names, dist, thresh, gens) == python_drop(names, dist, thresh, gens)
