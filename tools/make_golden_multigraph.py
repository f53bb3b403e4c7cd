#!/usr/bin/env python3
"""
tools/make_golden_multigraph.py -- tests/golden/multigraph_<name>.npz by RUNNING THE REFERENCE (build container only)
on networkx MultiGraph / MultiDiGraph inputs: parallel edges, self-loops, weights on some / all / no edges.

    PYTHONDONTWRITEBYTECODE=1 python tools/make_golden_multigraph.py

Each fixture holds the edge list in insertion order (src, dst, w with NaN = no weight attribute), the reference's
generation-0 table, neighbour lists, final table (columns, values, dtypes) and generation_count.
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

import networkx as nx                                    # noqa: E402
from graphrole import RecursiveFeatureExtractor          # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden')


def random_edges(seed, n, m, loops, weights):
    rng = np.random.default_rng(seed)
    src, dst, w = [], [], []
    while len(src) < m:
        a, b = (int(x) for x in rng.integers(0, n, 2))
        if a == b and not loops:
            continue
        reps = 1 + int(rng.random() < 0.3) + int(rng.random() < 0.1)          # parallel edges
        for _ in range(reps):
            if rng.random() < 0.5:
                a, b = b, a                                                   # either orientation
            src.append(a); dst.append(b)
            if weights == 'all' or (weights == 'some' and rng.random() < 0.5):
                w.append(float(rng.integers(1, 24)) / 4.0 if weights != 'int' else float(rng.integers(1, 6)))
            elif weights == 'int':
                w.append(float(rng.integers(1, 6)))
            else:
                w.append(np.nan)
    return np.array(src), np.array(dst), np.array(w)


CASES = {
    'undirected_unweighted': dict(seed=1, n=40, m=90, directed=False, loops=True, weights=None),
    'undirected_some_weights': dict(seed=2, n=40, m=90, directed=False, loops=True, weights='some'),
    'undirected_int_weights': dict(seed=3, n=30, m=70, directed=False, loops=False, weights='int'),
    'directed_unweighted': dict(seed=4, n=40, m=100, directed=True, loops=True, weights=None),
    'directed_all_weights': dict(seed=5, n=40, m=100, directed=True, loops=True, weights='all'),
}


def build(n, src, dst, w, directed, int_weights):
    G = nx.MultiDiGraph() if directed else nx.MultiGraph()
    G.add_nodes_from(range(n))
    for a, b, x in zip(src, dst, w):
        if np.isnan(x):
            G.add_edge(int(a), int(b))
        else:
            G.add_edge(int(a), int(b), weight=int(x) if int_weights else float(x))
    return G


def main():
    for name, c in CASES.items():
        src, dst, w = random_edges(c['seed'], c['n'], c['m'], c['loops'], c['weights'])
        G = build(c['n'], src, dst, w, c['directed'], c['weights'] == 'int')
        fe = RecursiveFeatureExtractor(G, max_generations=4, aggs=['sum', 'mean'])
        X = fe.extract_features()
        gen0 = fe.graph.get_neighborhood_features()
        nbrs = [list(fe.graph.get_neighbors(v)) for v in range(c['n'])]
        np.savez_compressed(
            os.path.join(OUT, f'multigraph_{name}.npz'),
            n=c['n'], directed=c['directed'], int_weights=c['weights'] == 'int', src=src, dst=dst, w=w,
            gen0_names_json=json.dumps(list(gen0.columns)), gen0_values=gen0.values.astype(float),
            gen0_dtypes_json=json.dumps([str(t) for t in gen0.dtypes]),
            neighbours_json=json.dumps(nbrs),
            final_columns_json=json.dumps(list(X.columns)), final_values=X.values.astype(float),
            final_dtypes_json=json.dumps([str(t) for t in X.dtypes]), generation_count=fe.generation_count)
        print(name, X.shape, fe.generation_count, [str(t) for t in gen0.dtypes])


if __name__ == '__main__':
    main()
