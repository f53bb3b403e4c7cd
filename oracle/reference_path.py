"""
oracle/reference_path.py -- TEST INFRASTRUCTURE / CPU BASELINE ONLY (see oracle/__init__.py).

The reference-faithful CPU formulation of the hot loops: the same third-party calls the
reference makes (pandas reindex/agg per node, sklearn NMF) -- what BASELINE.md section 3 calls
"baseline (1)".  It is interpreter bound (about 3 ms per node per generation), so bench.py times
it on contiguous node samples and labels the full-graph figure as extrapolated.  Equality with
the reference itself (golden tables produced by the imported reference, tools/make_golden.py):
tests/test_oracle_pinned.py::test_reference_path_equals_golden.

Legs (each the third-party call sequence of one reference function):
  aggregate_rows_pandas    features/extract.py:104-119   per-node reindex -> agg -> fillna
  egonet_rows_networkx     graph/interface/networkx.py:71-83,115-123   nx.ego_graph + nx.edge_boundary
  prune_distances          features/prune.py:13-56,94-116   np.unique binning per column + scipy pdist
  sklearn_nmf              roles/factor.py:19-25
"""
from __future__ import annotations

import time
from typing import Sequence, Tuple

import numpy as np
import pandas as pd


def aggregate_rows_pandas(row_ptr: np.ndarray, col: np.ndarray, features: pd.DataFrame,
                          rows: Sequence[int]) -> pd.DataFrame:
    """
    graphrole/features/extract.py:104-119 for the given rows: per node, select the neighbours'
    feature rows, aggregate with ['sum', 'mean'], NaN -> 0, name the results '<feature>(<agg>)'.
    `features` is indexed 0..n-1.
    """
    out = {}
    for v in rows:
        nbrs = col[row_ptr[v]:row_ptr[v + 1]]
        agg = features.reindex(index=nbrs).agg(['sum', 'mean']).fillna(0)
        flat = {}
        for how, series in agg.iterrows():
            for feat, val in series.items():
                flat[f'{feat}({how})'] = val
        out[v] = flat
    return pd.DataFrame.from_dict(out, orient='index')


def time_aggregate_sample(row_ptr, col, X: np.ndarray, names: Sequence[str], first_row: int,
                          n_rows: int) -> Tuple[float, int]:
    """Seconds and edges consumed for a contiguous block of rows (one generation)."""
    frame = pd.DataFrame(X, columns=list(names))
    rows = range(first_row, min(first_row + n_rows, len(row_ptr) - 1))
    t0 = time.perf_counter()
    aggregate_rows_pandas(row_ptr, col, frame, rows)
    dt = time.perf_counter() - t0
    edges = int(row_ptr[rows[-1] + 1] - row_ptr[rows[0]])
    return dt, edges


def egonet_rows_networkx(G, nodes) -> pd.DataFrame:
    """
    graphrole/graph/interface/networkx.py:71-83 for the given nodes of a networkx graph: the 1-hop
    ego network and its edge boundary are materialised per node, both edge sets are summed by weight
    (1 for an edge without one, :115-123).
    """
    import networkx as nx

    def weight_sum(edges):
        return sum(G.get_edge_data(*e, default={}).get('weight', 1) for e in edges)

    out = {}
    for v in nodes:
        ego = nx.ego_graph(G, v, radius=1)
        out[v] = {'internal_edges': weight_sum(ego.edges),
                  'external_edges': weight_sum(list(nx.edge_boundary(G, ego.nodes)))}
    return pd.DataFrame.from_dict(out, orient='index')


def bin_column_numpy(arr: np.ndarray, frac: float = 0.5) -> np.ndarray:
    """graphrole/features/prune.py:13-56 with the reference's own primitives: np.unique + cumulative
    counts, one searchsorted per bin, one boolean mask over the whole column per bin (:44-48 -- the
    reference counts the mask with Python's sum(); np.count_nonzero here, which only makes this
    baseline faster than the original)."""
    n = len(arr)
    out = np.zeros(n, dtype=int)
    uniq, counts = np.unique(arr, return_counts=True)
    cum = np.cumsum(counts)
    done, lo = 0, -np.inf
    for b in range(n):
        size = max(int(frac * (n - done)), 1)
        hi = uniq[np.searchsorted(cum, done + size)]
        mask = np.logical_and(arr > lo, arr <= hi)
        out[mask] = b
        done += int(np.count_nonzero(mask))
        lo = hi
        if done == n:
            break
    return out


def prune_distances(features: pd.DataFrame) -> np.ndarray:
    """graphrole/features/prune.py:94-108: every column binned, then the condensed Chebyshev
    distance vector of scipy's pdist over the binned columns."""
    from scipy.spatial.distance import pdist
    binned = features.apply(bin_column_numpy)
    return pdist(binned.T, metric='chebychev')


def networkx_sample_graph(row_ptr: np.ndarray, col: np.ndarray, first_row: int, n_rows: int):
    """Undirected unweighted networkx graph holding every edge incident to the rows
    [first_row, first_row + n_rows) and every edge among their neighbours' rows that the ego
    networks of those rows can see (all edges incident to a neighbour of a sampled row).
    ego_graph / edge_boundary of a sampled row on this graph equal those on the whole graph."""
    import networkx as nx
    rows = np.arange(first_row, min(first_row + n_rows, len(row_ptr) - 1))
    seen = np.unique(np.concatenate([rows] + [col[row_ptr[v]:row_ptr[v + 1]] for v in rows]))
    G = nx.Graph()
    G.add_nodes_from(int(v) for v in seen)
    for a in seen:
        nb = col[row_ptr[a]:row_ptr[a + 1]]
        G.add_edges_from((int(a), int(b)) for b in nb)
    return G, [int(v) for v in rows]


def sklearn_nmf(X: np.ndarray, n_roles: int):
    """graphrole/roles/factor.py:19-25 verbatim in effect: sklearn NMF(mu, nndsvda)."""
    import warnings
    from sklearn.decomposition import NMF
    model = NMF(n_components=n_roles, solver='mu', init='nndsvda')
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        t0 = time.perf_counter()
        W = model.fit_transform(X)
        dt = time.perf_counter() - t0
    return W, model.components_, model.n_iter_, dt
