#!/usr/bin/env python3
"""
tools/make_golden_order.py -- what the ORDER of the reference's generation-0 additions depends on, for the weighted
fixtures: the graph's node iteration order and (directed) every node's predecessor list in insertion order.

  tests/golden/order_<name>.npz    for the weighted cases of tools/make_golden.py (same graphs, rebuilt by the same
                                   builders): node_order, pred_ptr, pred_idx -- side files, the refex_<name>.npz
                                   fixtures are untouched
  tests/golden/gen0w_<name>.npz    new small weighted graphs with SHUFFLED node / edge insertion order, a self-loop,
                                   egos larger than half the graph: arrays + orders + the REFERENCE's generation-0
                                   table (NetworkxInterface.get_neighborhood_features) and final ReFeX table

The reference is imported here and only here; the fixtures are data.
    PYTHONDONTWRITEBYTECODE=1 python tools/make_golden_order.py
"""
import json
import os
import sys
import warnings

import networkx as nx
import numpy as np

REF = '/root/reference'
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.dont_write_bytecode = True
sys.path.insert(0, os.path.join(REF, 'examples'))
sys.path.insert(0, REF)
sys.path.insert(0, ROOT)
warnings.simplefilter('ignore')

from graphrole import RecursiveFeatureExtractor                                  # noqa: E402
from graphrole.graph.interface.networkx import NetworkxInterface               # noqa: E402
from tools.make_golden import REFEX_CASES, adjacency_arrays, graph_arrays        # noqa: E402

OUT = os.path.join(ROOT, 'tests', 'golden')


def orders(G, labels):
    index = {lab: i for i, lab in enumerate(labels)}
    out = {'node_order': np.array([index[x] for x in G.nodes], dtype=np.int64)}
    if G.is_directed():
        ptr, idx = [0], []
        for lab in labels:
            idx.extend(index[u] for u in G.pred[lab])
            ptr.append(len(idx))
        out['pred_ptr'] = np.array(ptr, dtype=np.int64)
        out['pred_idx'] = np.array(idx, dtype=np.int32)
    return out


def random_weighted(n, m, directed, seed):
    G = nx.gnm_random_graph(n, m, seed=seed, directed=directed)
    rng = np.random.default_rng(seed)
    for _, _, d in G.edges(data=True):
        d['weight'] = float(rng.uniform(0.1, 5.0))
    H = (nx.DiGraph if directed else nx.Graph)()
    nodes = list(G.nodes)
    rng.shuffle(nodes)
    H.add_nodes_from(nodes)
    edges = list(G.edges(data=True))
    rng.shuffle(edges)
    H.add_edges_from(edges)
    H.add_edge(3, 3, weight=2.5)
    return H


NEW_CASES = {'rw60': (60, 300, False, 1), 'rwd80': (80, 700, True, 2), 'rw400': (400, 3000, False, 3)}


def main():
    import glob
    for path in sorted(glob.glob(os.path.join(OUT, 'refex_*.npz'))):
        name = os.path.basename(path)[len('refex_'):-len('.npz')]
        z = np.load(path)
        if not len(z['w']) or name not in REFEX_CASES:
            continue
        G, _ = REFEX_CASES[name]()
        labels, src, dst, w = graph_arrays(G)
        assert np.array_equal(src, z['src']) and np.array_equal(dst, z['dst']) and np.array_equal(w, z['w']), name
        np.savez_compressed(os.path.join(OUT, f'order_{name}.npz'), **orders(G, labels))
        print(f'order_{name}: n={len(labels)}')
    for name, spec in NEW_CASES.items():
        G = random_weighted(*spec)
        labels, src, dst, w = graph_arrays(G)
        adj_ptr, adj_idx = adjacency_arrays(G, labels)
        gen0 = NetworkxInterface(G).get_neighborhood_features().loc[labels]
        fe = RecursiveFeatureExtractor(G, max_generations=3, aggs=['sum', 'mean'])
        final = fe.extract_features().loc[labels]
        np.savez_compressed(
            os.path.join(OUT, f'gen0w_{name}.npz'), n=len(labels), src=src, dst=dst, w=w, directed=G.is_directed(),
            labels_json=json.dumps(labels), num_edges=G.number_of_edges(), adj_ptr=adj_ptr, adj_idx=adj_idx,
            gen0_names_json=json.dumps(list(gen0.columns)), gen0_values=gen0.values.astype(np.float64),
            max_generations=3, generation_count=fe.generation_count, final_columns_json=json.dumps(list(final.columns)),
            final_values=final.values.astype(np.float64), **orders(G, labels))
        big = sum(1 for v in G.nodes if 2 * (len(set(G[v]) | {v})) >= len(G))
        print(f'gen0w_{name}: n={len(labels)} m={G.number_of_edges()} egos of at least half the graph: {big}')


if __name__ == '__main__':
    main()
