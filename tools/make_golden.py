#!/usr/bin/env python3
"""
tools/make_golden.py -- generate tests/golden/* by RUNNING THE REFERENCE in the build container.

    PYTHONDONTWRITEBYTECODE=1 python tools/make_golden.py [--only NAME ...]

The reference (/root/reference, read-only) is imported here and ONLY here; the fixtures it
writes are data (inputs + the reference's outputs).  The reference never travels to the GPU box.
The reference is driven with aggs=['sum','mean'] because its default aggs raise TypeError on
pandas >= 2; the column names are identical.

Each ReFeX fixture (refex_<name>.npz) holds
  graph      : n, src, dst, w (empty if unweighted), directed, labels_json, attr_names_json, attr_values
  kwargs_json: constructor kwargs (attributes=...)
  gen0_names_json / gen0_values                          (get_neighborhood_features)
  per generation g: cand_names, cand_values (the DataFrame handed to _update), working_before,
                    binned (vertical_log_binning of every pruner input column), cheb (pdist),
                    dropped, retained, working_after
  final_columns_json / final_values / generation_count
NMF fixtures (nmf_<name>.npz): X, r, seed, omega, W0, H0, W, H, n_iter.
"""
import argparse
import json
import os
import sys
import warnings

import numpy as np

REF = '/root/reference'
sys.dont_write_bytecode = True
sys.path.insert(0, REF)
sys.path.insert(0, os.path.join(REF, 'examples'))
warnings.simplefilter('ignore')

import networkx as nx                                    # noqa: E402
import pandas as pd                                      # noqa: E402
from scipy.spatial.distance import pdist, squareform     # noqa: E402
from graphrole import RecursiveFeatureExtractor          # noqa: E402
from graphrole.features.prune import FeaturePruner, vertical_log_binning   # noqa: E402
from graphrole.roles.factor import get_nmf_decomposition                   # noqa: E402
from data import load_nx_karate_club_graph               # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden')
AGGS = ['sum', 'mean']


# --------------------------------------------------------------------------- graphs
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import graphs as G_                              # noqa: E402


def g_karate(weighted):
    """The reference's own copy of the karate-club graph (examples/data/data.py:6-15)."""
    return load_nx_karate_club_graph(weighted=weighted), {}


REFEX_CASES = {
    'karate': lambda: g_karate(False),
    'karate_weighted': lambda: g_karate(True),
}
REFEX_CASES.update(G_.BUILDERS)
# the same graphs with other aggregation lists (extract.py:36-47 `aggs`; names as pandas labels them)
CASE_AGGS = G_.CASE_AGGS
for _name, _spec in CASE_AGGS.items():
    REFEX_CASES[_name] = REFEX_CASES[_spec[0]]


# --------------------------------------------------------------------------- ReFeX capture
def graph_arrays(G):
    labels = sorted(G.nodes)
    index = {lab: i for i, lab in enumerate(labels)}
    weighted = any('weight' in d for _, _, d in G.edges(data=True))
    src = np.array([index[u] for u, v in G.edges], dtype=np.int64)
    dst = np.array([index[v] for u, v in G.edges], dtype=np.int64)
    w = (np.array([d.get('weight', 1) for _, _, d in G.edges(data=True)], dtype=np.float64)
         if weighted else np.zeros(0))
    return labels, src, dst, w


def adjacency_arrays(G, labels):
    """G[node] iteration order of every node (rows = sorted labels): the order in which the
    reference's reindex(nbrs) lines up the neighbour rows it sums (features/extract.py:108-110)."""
    index = {lab: i for i, lab in enumerate(labels)}
    adj_ptr = np.zeros(len(labels) + 1, dtype=np.int64)
    adj_idx = []
    for i, lab in enumerate(labels):
        nbrs = [index[v] for v in G[lab]]
        adj_idx.extend(nbrs)
        adj_ptr[i + 1] = len(adj_idx)
    return adj_ptr, np.array(adj_idx, dtype=np.int32)


def capture_refex(name, G, kwargs, max_generations=10):
    """Drive the reference's own methods in the order extract_features does (extract.py:65-89)."""
    aggs = CASE_AGGS[name][1] if name in CASE_AGGS else AGGS
    fe = RecursiveFeatureExtractor(G, max_generations=max_generations, aggs=aggs, **kwargs)
    labels, src, dst, w = graph_arrays(G)
    out = dict(n=len(labels), src=src, dst=dst, w=w, directed=G.is_directed(),
               labels_json=json.dumps(labels), kwargs_json=json.dumps(kwargs),
               num_edges=G.number_of_edges(), max_generations=max_generations, aggs_json=json.dumps(aggs))
    out['adj_ptr'], out['adj_idx'] = adjacency_arrays(G, labels)

    def record(gen, cand, thresh):
        cand = cand.reindex(labels).fillna(0) if len(cand.index) != len(labels) else cand.loc[labels]
        pre = pd.concat([fe._features, cand], axis=1, sort=True).fillna(0)
        binned = pre.apply(vertical_log_binning)
        cheb = squareform(pdist(binned.T.values, metric='chebychev')) if pre.shape[1] > 1 \
            else np.zeros((pre.shape[1], pre.shape[1]))
        pruner = FeaturePruner(fe._final_features, thresh)
        dropped = sorted(pruner.prune_features(pre))
        rec = {
            f'g{gen}_cand_names_json': json.dumps(list(cand.columns)),
            f'g{gen}_cand_values': cand.values.astype(np.float64),
            f'g{gen}_working_before_json': json.dumps(list(pre.columns)),
            f'g{gen}_binned': binned.values.astype(np.int16),
            f'g{gen}_cheb': cheb.astype(np.int64),
            f'g{gen}_dropped_json': json.dumps(dropped),
        }
        return rec

    feats = fe.graph.get_neighborhood_features()
    out['gen0_names_json'] = json.dumps(list(feats.columns))
    out['gen0_values'] = feats.loc[labels].values.astype(np.float64)
    out['gen0_is_int'] = bool(all(np.issubdtype(t, np.integer) for t in feats.dtypes))
    out.update(record(0, feats, 0))
    fe._update(feats)
    out['g0_retained_json'] = json.dumps(list(fe._final_features[0].keys()))
    out['g0_working_after_json'] = json.dumps(list(fe._features.columns))
    n_gen = 1
    for gen in range(1, fe.max_generations):
        fe.generation_count = gen
        fe._feature_group_thresh = gen
        feats = fe._get_next_features()
        out.update(record(gen, feats, gen))
        fe._update(feats)
        out[f'g{gen}_retained_json'] = json.dumps(list(fe._final_features[gen].keys()))
        out[f'g{gen}_working_after_json'] = json.dumps(list(fe._features.columns))
        n_gen = gen + 1
        if not fe._final_features[gen]:
            break
    final = fe._finalize_features()
    out['n_generations_recorded'] = n_gen
    out['generation_count'] = fe.generation_count
    out['final_columns_json'] = json.dumps(list(final.columns))
    out['final_values'] = final.loc[labels].values.astype(np.float64)
    out['final_dtypes_json'] = json.dumps([str(t) for t in final.dtypes])
    # integer columns exactly (fp64 loses the low bits of wrapped int64 products): int64 per column, 0 where float
    ordered = final.loc[labels]
    out['final_values_i64'] = np.column_stack([ordered[c].to_numpy().astype(np.int64) if ordered[c].dtype.kind in 'iu'
                                               else np.zeros(len(labels), dtype=np.int64) for c in ordered.columns]) \
        if ordered.shape[1] else np.zeros((len(labels), 0), dtype=np.int64)
    gen0 = fe.graph.get_neighborhood_features().loc[labels]
    out['gen0_dtypes_json'] = json.dumps([str(t) for t in gen0.dtypes])

    # cross-check: an untouched instance run through the public entry point agrees exactly
    fe2 = RecursiveFeatureExtractor(G, max_generations=max_generations, aggs=aggs, **kwargs)
    final2 = fe2.extract_features()
    assert list(final2.columns) == list(final.columns)
    assert fe2.generation_count == fe.generation_count
    assert np.array_equal(final2.values, final.values)
    np.savez_compressed(os.path.join(OUT, f'refex_{name}.npz'), **out)
    print(f'refex_{name}: n={len(labels)} final={final.shape} generations={fe.generation_count}')
    return final


# --------------------------------------------------------------------------- NMF capture
def capture_nmf(name, X, r, seed):
    from sklearn.decomposition import NMF
    from sklearn.decomposition._nmf import _initialize_nmf
    X = np.ascontiguousarray(X, dtype=np.float64)
    n, f = X.shape
    inner = n if n < f else f
    np.random.seed(seed)
    omega = np.random.normal(size=(inner, r + 10))
    np.random.seed(seed)
    W0, H0 = _initialize_nmf(X, r, init='nndsvda', random_state=None)
    np.random.seed(seed)
    G, F = get_nmf_decomposition(X, r)                         # the reference's call site
    np.random.seed(seed)
    model = NMF(n_components=r, solver='mu', init='nndsvda')
    G2 = model.fit_transform(X)
    assert np.array_equal(G, G2) and np.array_equal(F, model.components_)
    np.savez_compressed(os.path.join(OUT, f'nmf_{name}.npz'), X=X, r=r, seed=seed, omega=omega,
                        W0=W0, H0=H0, W=G, H=F, n_iter=model.n_iter_)
    print(f'nmf_{name}: X={X.shape} r={r} n_iter={model.n_iter_}')


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--only', nargs='*', default=None)
    args = ap.parse_args()
    os.makedirs(OUT, exist_ok=True)
    finals = {}
    for name, make in REFEX_CASES.items():
        if args.only and name not in args.only:
            continue
        G, kwargs = make()
        spec = CASE_AGGS.get(name, ())
        finals[name] = capture_refex(name, G, kwargs, *([spec[2]] if len(spec) > 2 else []))

    if args.only and not any(x.startswith('nmf') for x in args.only):
        return
    rs = np.random.RandomState(7)
    nmf_cases = {
        'rand20x30_r3': (np.random.RandomState(0).rand(20, 30), 3, 0),       # transposed branch
        'rand500x12_r6': (np.abs(rs.randn(500, 12)) * np.arange(1, 13), 6, 1),   # F <= r+10: exact
        'rand800x40_r6': (np.abs(rs.randn(800, 40)) * np.linspace(1, 9, 40), 6, 2),  # F > r+10
        'rand3000x9_r2': (rs.rand(3000, 9) ** 3, 2, 3),                      # n_iter = 7 branch
    }
    if 'karate' in finals:
        nmf_cases['karate_r4'] = (finals['karate'].values, 4, 0)
    if 'er2000' in finals:
        nmf_cases['er2000_r6'] = (finals['er2000'].values, 6, 0)
    if 'ba2000' in finals:
        nmf_cases['ba2000_r6'] = (finals['ba2000'].values, 6, 5)
    if 'dw200_attrs' in finals:
        nmf_cases['dw200_r5'] = (finals['dw200_attrs'].values, 5, 11)
    for name, (X, r, seed) in nmf_cases.items():
        if args.only and ('nmf_' + name) not in args.only and 'nmf' not in args.only:
            continue
        capture_nmf(name, X, r, seed)


if __name__ == '__main__':
    main()
