"""Differential fuzz on the GPU box: random small graphs (sizes, densities, directed / weighted, self-loops,
isolated nodes, aggs) through RecursiveFeatureExtractor against the oracle -- bit-exact on unweighted and integer-weighted graphs, generation 0 to
1e-11 with non-integer weights.  Usage: PYTHONPATH=. [FUZZ_BIG=1] python tools/fuzz_refex.py [cases] [seed]"""
import os
import sys

import numpy as np

from graphrole_amd import RecursiveFeatureExtractor
from graphrole_amd.graph import CSRGraph
from oracle import refex


def one(rng, case):
    sizes = [5, 17, 64, 300, 1500, 6000]
    if os.environ.get('FUZZ_BIG') == '1':                       # medium graphs: the sampled bucket map of the binning at work
        sizes = [6000, 40_000, 120_000, 400_000]
    n = int(rng.choice(sizes))
    directed = bool(rng.integers(0, 2))
    weighted = bool(rng.integers(0, 2))
    m = int(n * rng.choice([0.5, 1.5, 4, 12]))
    if rng.random() < 0.3:                                      # power-law-ish targets: hubs (rows > 128 neighbours)
        dst = np.minimum((rng.pareto(1.0, m) * 2).astype(np.int64), n - 1)
    else:
        dst = rng.integers(0, n, m)
    if rng.random() < 0.3:                                      # power-law-ish SOURCES too: long rows (round 5: the ego-net
        src = np.minimum((rng.pareto(0.8, m) * 2).astype(np.int64), n - 1)     # kernels for 33 - 64, 65 - 511, 512+ neighbours)
    else:
        src = rng.integers(0, n, m)
    if rng.random() < 0.7:                                      # mostly without self-loops
        keep = src != dst
        src, dst = src[keep], dst[keep]
    key = src * n + dst if directed else np.minimum(src, dst) * n + np.maximum(src, dst)
    _, first = np.unique(key, return_index=True)
    first.sort()
    src, dst = src[first], dst[first]
    if len(src) == 0:
        return 'skip'
    w = None
    if weighted:
        w = rng.integers(1, 6, len(src)).astype(np.float64) if rng.random() < 0.5 else rng.random(len(src)) + 0.1
    aggs = [['sum', 'mean'], ['sum', 'mean', 'max'], ['mean', 'min', 'std']][int(rng.integers(0, 3))]
    gens = int(rng.integers(2, 6))
    G = CSRGraph(n, src, dst, weights=w, directed=directed)
    X = RecursiveFeatureExtractor(G, max_generations=gens, aggs=aggs).extract_features()
    og = refex.graph_from_arrays(n, src, dst, w, directed, list(range(n)))
    ref = refex.extract_features(og, max_generations=gens, aggs=aggs, fast=True)
    float_weights = weighted and not np.all(w == np.round(w))
    if float_weights:
        # generation 0 of a graph with non-integer weights is within 1e-12, not bit-exact (the weight sums run in
        # another order than Python's sum()): near-equal values may then fall on different sides
        # of a bin edge and the pruning decisions -- hence the column lists -- may differ.  Checked: generation 0.
        for name in ref.columns:
            if '(' not in name and name in X.columns:
                a, b = X[name].values.astype(float), ref.values[:, ref.columns.index(name)]
                assert np.all(np.abs(a - b) <= 1e-11 * max(np.abs(b).max(), 1.0)), (case, name)
    else:
        assert list(X.columns) == ref.columns, (case, n, len(src), directed, weighted, aggs, gens, list(X.columns), ref.columns)
        got, exp = X.values.astype(float), ref.values
        assert np.array_equal(got, exp), (case, n, directed, weighted, aggs, gens, int((got != exp).sum()))
    return f'n={n} m={len(src)} dir={directed} w={weighted} aggs={aggs} gens={gens} F={X.shape[1]}'


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    rng = np.random.default_rng(seed)
    done = 0
    for case in range(cases):
        r = one(rng, case)
        if r != 'skip':
            done += 1
            if case % 10 == 0:
                print(case, r, flush=True)
    print('fuzz ok:', done, 'graphs')


if __name__ == '__main__':
    main()
