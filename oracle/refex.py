"""
oracle/refex.py -- TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Numpy restatement of the ReFeX half of the GraphRole hot path.  Citations are
``/root/reference/<file>:<line>``.  Parity: pinned by tests/test_oracle_pinned.py against the
reference's own known-answer tables and against golden vectors produced by the reference
(tools/make_golden.py).

Conventions:
  * rows  = node labels sorted ascending            (graph/interface/base.py:24-25)
  * neighbours of v = out-adjacency *set* of v, self-loop included, unweighted in the recursion
                                                     (graph/interface/networkx.py:42-46)
  * the recursion sums a node's neighbours in ADJACENCY order (G[node] iteration order =
    order in which the incident edges were added, features/extract.py:108-110) with numpy's
    pairwise summation (Series.sum -> ndarray.sum): `adj_col` keeps that order, `col` is the
    same rows sorted ascending for the set-based generation-0 features.
  * every feature value is carried as float64; integer-valued gen-0 columns of unweighted graphs
    are cast to int64 only when a DataFrame is produced.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from numbers import Number
from typing import Dict, List, Optional, Sequence, Set, Tuple

import os

import numpy as np

#: threads of the column binning of the full-size runs (0 = up to 32, 1 = serial: bench.py's single-core timing leg)
BIN_THREADS = 0


# --------------------------------------------------------------------------------------
# graph container
# --------------------------------------------------------------------------------------
@dataclass
class OracleGraph:
    """CSR of the out-adjacency with rows in sorted-label order."""
    labels: list
    row_ptr: np.ndarray            # int64 [n+1]
    col: np.ndarray                # int32 [nnz], ascending inside each row
    w: Optional[np.ndarray]        # float64 [nnz] or None (implicit weight 1)
    directed: bool
    num_edges: int                 # G.number_of_edges()
    # transpose (in-adjacency) -- only needed for directed in_degree
    t_row_ptr: Optional[np.ndarray] = None
    t_col: Optional[np.ndarray] = None
    t_w: Optional[np.ndarray] = None
    attrs: Dict[str, np.ndarray] = field(default_factory=dict)   # 'attribute_<name>' -> float64[n]
    adj_col: Optional[np.ndarray] = None   # int32 [nnz]: rows of `col` in adjacency (insertion) order
    # what only matters for the ORDER in which the reference adds edge weights (neighborhood_features_networkx_order):
    node_order: Optional[np.ndarray] = None     # row indices in the order the graph iterates its nodes (default: sorted)
    pred_ptr: Optional[np.ndarray] = None        # directed: in-neighbours of every row in insertion order (G.pred[v])
    pred_col: Optional[np.ndarray] = None

    @property
    def n(self) -> int:
        return len(self.row_ptr) - 1

    @property
    def nnz(self) -> int:
        return int(self.row_ptr[-1])

    def row(self, v: int) -> np.ndarray:
        return self.col[self.row_ptr[v]:self.row_ptr[v + 1]]

    def adj_row(self, v: int) -> np.ndarray:
        """Neighbours of v in the order the reference visits them (falls back to ascending)."""
        c = self.col if self.adj_col is None else self.adj_col
        return c[self.row_ptr[v]:self.row_ptr[v + 1]]

    def row_w(self, v: int) -> np.ndarray:
        s, e = self.row_ptr[v], self.row_ptr[v + 1]
        if self.w is None:
            return np.ones(e - s, dtype=np.float64)
        return self.w[s:e]


def _csr_from_coo(n: int, src: np.ndarray, dst: np.ndarray, w: Optional[np.ndarray]):
    order = np.lexsort((dst, src))
    src, dst = src[order], dst[order]
    row_ptr = np.zeros(n + 1, dtype=np.int64)
    np.add.at(row_ptr, src + 1, 1)
    row_ptr = np.cumsum(row_ptr)
    return row_ptr, dst.astype(np.int32), (None if w is None else w[order].astype(np.float64))


def adjacency_order(n: int, src: np.ndarray, dst: np.ndarray, directed: bool) -> np.ndarray:
    """
    Rows of the out-adjacency in networkx insertion order for a graph built by adding the edges
    (src[i], dst[i]) one after the other: add_edge(u, v) appends v to adj[u] and u to adj[v]
    (once for a self-loop), so row x lists its neighbours by the index of the incident edge.
    """
    m = len(src)
    if directed:
        rows, cols = src, dst
    else:
        rows = np.empty(2 * m, dtype=np.int64)
        cols = np.empty(2 * m, dtype=np.int64)
        rows[0::2], rows[1::2] = src, dst
        cols[0::2], cols[1::2] = dst, src
        keep = np.ones(2 * m, dtype=bool)
        keep[1::2] = src != dst
        rows, cols = rows[keep], cols[keep]
    order = np.argsort(rows, kind='stable')
    return cols[order].astype(np.int32)


def graph_from_arrays(n: int, src, dst, w=None, directed: bool = False, labels=None) -> OracleGraph:
    """
    Build from unique edge arrays (row indices already refer to sorted labels).
    Undirected input lists every edge once (either orientation); self-loops allowed.
    The neighbour order of the recursion is the order of appearance in the arrays.
    """
    src = np.asarray(src, dtype=np.int64)
    dst = np.asarray(dst, dtype=np.int64)
    w = None if w is None else np.asarray(w, dtype=np.float64)
    m = len(src)
    if directed:
        row_ptr, col, ww = _csr_from_coo(n, src, dst, w)
        t_row_ptr, t_col, t_w = _csr_from_coo(n, dst, src, w)
    else:
        loop = src == dst
        s2 = np.concatenate([src, dst[~loop]])
        d2 = np.concatenate([dst, src[~loop]])
        w2 = None if w is None else np.concatenate([w, w[~loop]])
        row_ptr, col, ww = _csr_from_coo(n, s2, d2, w2)
        t_row_ptr = t_col = t_w = None
    return OracleGraph(labels=list(range(n)) if labels is None else list(labels),
                       row_ptr=row_ptr, col=col, w=ww, directed=directed, num_edges=m,
                       t_row_ptr=t_row_ptr, t_col=t_col, t_w=t_w,
                       adj_col=adjacency_order(n, src, dst, directed))


def graph_from_networkx(G, attributes: bool = False, attributes_include: Sequence[str] = (),
                        attributes_exclude: Sequence[str] = ()) -> OracleGraph:
    """
    networkx (Di)Graph -> OracleGraph.  Follows NetworkxInterface: get_nodes / get_neighbors
    (graph/interface/networkx.py:36-46), edge weight default 1 (:115-123), attribute selection
    (:87-113, base.py:28-48).
    """
    labels = sorted(G.nodes)
    index = {lab: i for i, lab in enumerate(labels)}
    directed = G.is_directed()
    weighted = any('weight' in d for _, _, d in G.edges(data=True))
    src, dst, ww = [], [], []
    for u, v, d in G.edges(data=True):
        src.append(index[u])
        dst.append(index[v])
        ww.append(d.get('weight', 1))
    g = graph_from_arrays(len(labels), src, dst, ww if weighted else None, directed, labels)
    g.num_edges = G.number_of_edges()
    # G.edges lists edges grouped by node, not in insertion order: take the adjacency order itself
    g.adj_col = np.array([index[v] for lab in labels for v in G.adj[lab]], dtype=np.int32)
    if attributes:
        g.attrs = _attribute_columns(G, labels, attributes_include, attributes_exclude)
    return g


def _attribute_columns(G, labels, include, exclude) -> Dict[str, np.ndarray]:
    """networkx.py:87-113: numeric attrs only, missing -> 0, exclude beats include."""
    exclude = set(exclude)
    cols: Dict[str, Dict] = {}
    if include:
        for name in include:
            if name in exclude:
                continue
            cols['attribute_' + name] = {
                node: a.get(name, 0) for node, a in G.nodes(data=True)
                if isinstance(a.get(name, 0), Number)
            }
    else:
        for node, a in G.nodes(data=True):
            for name, val in a.items():
                if name in exclude or not isinstance(val, Number):
                    continue
                cols.setdefault('attribute_' + name, {})[node] = val
    return {k: np.array([float(d.get(lab, 0)) for lab in labels]) for k, d in cols.items()}


# --------------------------------------------------------------------------------------
# generation 0: local + ego-net features
# --------------------------------------------------------------------------------------
def local_features(g: OracleGraph) -> Dict[str, np.ndarray]:
    """
    graph/interface/networkx.py:48-69.  Undirected: weighted degree, a self-loop counts twice
    (networkx convention).  Directed: in/out/total weighted degree.
    """
    n = g.n
    rowsum = np.array([g.row_w(v).sum() for v in range(n)], dtype=np.float64)
    if g.directed:
        indeg = np.zeros(n)
        for v in range(n):
            s, e = g.t_row_ptr[v], g.t_row_ptr[v + 1]
            indeg[v] = (e - s) if g.t_w is None else g.t_w[s:e].sum()
        out = {'in_degree': indeg, 'out_degree': rowsum, 'total_degree': rowsum + indeg}
    else:
        loopw = np.zeros(n)
        for v in range(n):
            r = g.row(v)
            hit = np.nonzero(r == v)[0]
            if len(hit):
                loopw[v] = g.row_w(v)[hit[0]]
        out = {'degree': rowsum + loopw}
    out.update(g.attrs)
    return out


def egonet_features(g: OracleGraph) -> Dict[str, np.ndarray]:
    """
    graph/interface/networkx.py:71-83,115-123.  ego(v) = {v} U nbrs(v).
    internal = sum of w over edges with both ends in ego (undirected: each edge once, self-loops
    once; directed: every arc).  external = sum of w over edge_boundary(G, ego): arcs/edges
    leaving the ego set.
    """
    n = g.n
    internal = np.zeros(n)
    external = np.zeros(n)
    for v in range(n):
        ego = np.union1d(g.row(v), [v])
        ins = 0.0
        ext = 0.0
        for a in ego:
            r = g.row(int(a))
            wr = g.row_w(int(a))
            inside = np.isin(r, ego, assume_unique=True)
            if g.directed:
                ins += wr[inside].sum()
            else:
                ins += wr[inside & (r >= a)].sum()
            ext += wr[~inside].sum()
        internal[v] = ins
        external[v] = ext
    return {'internal_edges': internal, 'external_edges': external}


# --------------------------------------------------------------------------------------
# generation 0 of WEIGHTED graphs in the reference's own order of additions
# --------------------------------------------------------------------------------------
def _weight_lookup(g: OracleGraph):
    n = g.n
    key = np.repeat(np.arange(n, dtype=np.int64), np.diff(g.row_ptr)) * n + g.col.astype(np.int64)
    w = np.ones(g.nnz) if g.w is None else g.w
    table = dict(zip(key.tolist(), w.tolist()))
    return lambda a, b: table[a * n + b]


def neighborhood_features_networkx_order(g: OracleGraph) -> Tuple[List[str], np.ndarray]:
    """
    Generation 0 (graph/interface/networkx.py:48-83,115-123) with every sum evaluated in the order networkx 3.x
    hands the terms to Python's left-to-right ``sum()`` -- the weighted columns then equal the reference's bit for bit.
    What that order is (networkx 3.4.2, restated from its published source; labels must be the graph's node objects,
    here: the sorted labels of an integer-labelled graph, for which Python's set layout does not depend on the process):

      degree            sum over G.adj[v] in insertion order, + the self-loop weight once more (reportviews.py DegreeView);
                        DiGraph: successors' sum + predecessors' sum, in_degree over G.pred[v] in insertion order
      internal_edges    ego = nx.ego_graph(G, v): BFS order [v] + G.adj[v] -> ``set`` (filters.show_nodes) -> the COPY's
                        node order is that set's iteration order when the ego set is smaller than half the graph
                        (coreviews.FilterAtlas.__iter__), else the graph's node order filtered; the copy's adjacency
                        lists are filled by add_edge in that node order x G.adj order; ``ego.edges`` walks the copy's
                        nodes and adjacency lists, an undirected edge reported at its first endpoint
      external_edges    nx.edge_boundary(G, ego.nodes): a second ``set`` built from the copy's node order, G.edges(nbunch)
                        over that set x G.adj order, undirected edges deduplicated by ``seen``, kept when exactly one end
                        is inside

    The two sets are real Python sets here: their iteration order IS the specification (a hash-table layout), which is
    why no array order reproduces it and why the device path keeps a tolerance on weighted generation-0 sums.  With
    string labels the layout depends on PYTHONHASHSEED: the reference does not reproduce its own last bits there.
    """
    n = g.n
    labels = list(g.labels)
    index = {lab: i for i, lab in enumerate(labels)}
    weight = _weight_lookup(g)
    order = list(range(n)) if g.node_order is None else [int(i) for i in g.node_order]
    adj = [[int(u) for u in g.adj_row(v)] for v in range(n)]                       # successors / neighbours, insertion order
    if g.directed:
        if g.pred_ptr is not None:
            pred = [[int(u) for u in g.pred_col[g.pred_ptr[v]:g.pred_ptr[v + 1]]] for v in range(n)]
        else:
            pred = [[int(u) for u in g.t_col[g.t_row_ptr[v]:g.t_row_ptr[v + 1]]] for v in range(n)]
    out: Dict[str, list] = {}
    if g.directed:
        ins = [sum(weight(u, v) for u in pred[v]) for v in range(n)]
        outs = [sum(weight(v, u) for u in adj[v]) for v in range(n)]
        # DiDegreeView.__iter__: sum over successors + sum over predecessors
        tot = [sum(weight(v, u) for u in adj[v]) + sum(weight(u, v) for u in pred[v]) for v in range(n)]
        out['in_degree'], out['out_degree'], out['total_degree'] = ins, outs, tot
    else:
        out['degree'] = [sum(weight(v, u) for u in adj[v]) + (weight(v, v) if v in adj[v] else 0) for v in range(n)]
    internal, external = [0] * n, [0] * n
    for v in range(n):
        bfs = [labels[v]] + [labels[u] for u in adj[v] if u != v]
        inside_set = set(bfs)                                                       # filters.show_nodes: set(nodes)
        if 2 * len(inside_set) < n:
            ego_nodes = [index[lab] for lab in inside_set]                           # FilterAtlas: the set's own order
        else:
            ego_nodes = [i for i in order if labels[i] in inside_set]
        member = {i for i in ego_nodes}
        # Graph.copy(): add_edges_from over (u in ego_nodes) x (G.adj[u] inside) fills the copy's adjacency lists
        cadj: Dict[int, Dict[int, None]] = {i: {} for i in ego_nodes}
        for a in ego_nodes:
            for b in adj[a]:
                if b in member:
                    cadj[a][b] = None
                    if not g.directed:
                        cadj[b][a] = None
        terms = []
        if g.directed:
            for a in ego_nodes:                                                      # OutEdgeView
                terms.extend(weight(a, b) for b in cadj[a])
        else:
            seen = set()
            for a in ego_nodes:                                                      # EdgeView: each edge at its first endpoint
                for b in cadj[a]:
                    if b not in seen:
                        terms.append(weight(a, b))
                seen.add(a)
        internal[v] = sum(terms)
        # edge_boundary(G, ego.nodes): nset1 = {n for n in ego.nodes if n in G}
        nset1 = {labels[i] for i in ego_nodes}
        terms = []
        if g.directed:
            for lab in nset1:                                                        # G.edges(nset1): OutEdgeView over the set
                a = index[lab]
                terms.extend(weight(a, b) for b in adj[a] if b not in member)
        else:
            seen = set()
            for lab in nset1:
                a = index[lab]
                for b in adj[a]:
                    if b not in seen and b not in member:
                        terms.append(weight(a, b))
                seen.add(a)
        external[v] = sum(terms)
    out['internal_edges'], out['external_edges'] = internal, external
    cols = dict(out)
    local_names = [k for k in cols if k not in ('internal_edges', 'external_edges')]
    names = local_names + list(g.attrs) + ['internal_edges', 'external_edges']
    data = [np.asarray(cols[k], dtype=np.float64) for k in local_names] + [np.asarray(g.attrs[k], dtype=np.float64) for k in g.attrs] + \
           [np.asarray(cols['internal_edges'], dtype=np.float64), np.asarray(cols['external_edges'], dtype=np.float64)]
    return names, np.column_stack(data)


def local_features_c(g: OracleGraph) -> Dict[str, np.ndarray]:
    """local_features through the plain-C twins (oracle/csrc/oracle_kernels.c)."""
    from . import ckernels
    if g.directed:
        outd = ckernels.rowsum(g.row_ptr, g.col, g.w, False)
        ind = ckernels.rowsum(g.t_row_ptr, g.t_col, g.t_w, False)
        out = {'in_degree': ind, 'out_degree': outd, 'total_degree': outd + ind}
    else:
        out = {'degree': ckernels.rowsum(g.row_ptr, g.col, g.w, True)}
    out.update(g.attrs)
    return out


def egonet_features_c(g: OracleGraph) -> Dict[str, np.ndarray]:
    from . import ckernels
    i, e = ckernels.egonet(g.row_ptr, g.col, g.w, g.directed)
    return {'internal_edges': i, 'external_edges': e}


def neighborhood_features(g: OracleGraph, fast: bool = False) -> Tuple[List[str], np.ndarray]:
    """graph/interface/base.py:18-26: concat([local, ego], axis=1)."""
    loc = local_features_c(g) if fast else local_features(g)
    ego = egonet_features_c(g) if fast else egonet_features(g)
    names = list(loc) + list(ego)
    X = np.column_stack([loc[k] for k in loc] + [ego[k] for k in ego])
    return names, X


# --------------------------------------------------------------------------------------
# recursion: neighbour aggregation
# --------------------------------------------------------------------------------------
PAIRWISE_BLOCK = 128     # numpy PW_BLOCKSIZE
REDUCE_CHUNK = 8192      # numpy's default ufunc buffer size (np.getbufsize()), in elements


def pairwise_sum(A: np.ndarray) -> np.ndarray:
    """
    Column sums of the rows of A [k, f] in the association order of numpy's pairwise summation
    (numpy/_core/src/umath/loops_utils.h.src, pairwise_sum_DOUBLE), which is what
    ``Series.sum()`` of the reference's ``features.reindex(nbrs).agg(...)`` executes per column
    (features/extract.py:110-113 -> pandas nanops.nansum -> ndarray.sum on a contiguous column):
      k < 8     sequential, left to right
      k <= 128  eight accumulators r[j] = A[j] + A[j+8] + ..., combined as
                ((r0+r1)+(r2+r3)) + ((r4+r5)+(r6+r7)), then the k % 8 trailing rows one by one
      k > 128   split at k2 = k//2 rounded down to a multiple of 8, recurse, add the halves
    """
    k = A.shape[0]
    if k < 8:
        res = np.zeros(A.shape[1])
        for i in range(k):
            res = res + A[i]
        return res
    if k <= PAIRWISE_BLOCK:
        r = [A[j].copy() for j in range(8)]
        k8 = k - k % 8
        for i in range(8, k8, 8):
            for j in range(8):
                r[j] = r[j] + A[i + j]
        res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]))
        for i in range(k8, k):
            res = res + A[i]
        return res
    k2 = k // 2
    k2 -= k2 % 8
    return pairwise_sum(A[:k2]) + pairwise_sum(A[k2:])


def ndarray_sum(A: np.ndarray) -> np.ndarray:
    """
    Column sums of A [k, f] exactly as ``ndarray.sum()`` evaluates a contiguous column: the
    reduction walks the column in chunks of 8192 elements (the ufunc buffer size), each chunk
    is summed by pairwise_sum and added to the running total:  ((0 + p(c0)) + p(c1)) + ...
    (checked against numpy and against the reference on a node with 12 000 neighbours).
    """
    total = np.zeros(A.shape[1])
    for i in range(0, A.shape[0], REDUCE_CHUNK):
        total = total + pairwise_sum(A[i:i + REDUCE_CHUNK])
    return total


def aggregate(g: OracleGraph, X: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """
    features/extract.py:98-119.  For every node v and column c:
        sum[v,c]  = sum_{u in nbrs(v)} X[u,c]     neighbours in adjacency order, ndarray_sum
        mean[v,c] = sum[v,c] / |nbrs(v)|          (0 when v has no neighbours, :113 fillna)
    """
    n, f = X.shape
    s = np.zeros((n, f))
    m = np.zeros((n, f))
    for v in range(n):
        nb = g.adj_row(v)
        if len(nb):
            acc = ndarray_sum(X[nb])
            s[v] = acc
            m[v] = acc / len(nb)
    return s, m


def aggregate_var(g: OracleGraph, X: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """
    features/extract.py:98-119 with 'var' / 'std' in `aggs`: pandas' nanvar with ddof = 1
    (pandas/core/nanops.py: avg = values.sum() / count; sqr = (avg - values) ** 2;
    var = sqr.sum() / (count - 1)), both sums by ndarray_sum in adjacency order; std = sqrt(var);
    fewer than two neighbours -> NaN -> 0 (:113).
    """
    n, f = X.shape
    var = np.zeros((n, f))
    std = np.zeros((n, f))
    for v in range(n):
        nb = g.adj_row(v)
        k = len(nb)
        if k >= 2:
            A = X[nb]
            avg = ndarray_sum(A) / k
            sq = ndarray_sum((avg - A) ** 2)
            var[v] = sq / (k - 1)
            std[v] = np.sqrt(var[v])
    return var, std


def aggregate_prod(g: OracleGraph, X: np.ndarray) -> np.ndarray:
    """
    features/extract.py:98-119 with 'prod' in `aggs`: Series.prod() = np.multiply.reduce, a plain
    left-to-right product over the neighbours in adjacency order (no pairwise tree for multiply);
    the product over no neighbours is 1, not NaN.  (On int64 columns pandas multiplies in int64
    and wraps silently; the restatement is in fp64, exact below 2**53 -- the engine refuses beyond.)
    """
    n, f = X.shape
    out = np.ones((n, f))
    for v in range(n):
        for u in g.adj_row(v):
            out[v] = out[v] * X[u]
    return out


def aggregate_minmax(g: OracleGraph, X: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """
    features/extract.py:98-119 with 'min' / 'max' in `aggs`: column-wise minimum / maximum over
    the neighbours' rows; a node without neighbours aggregates an empty frame -> NaN -> 0 (:113).
    """
    n, f = X.shape
    lo = np.zeros((n, f))
    hi = np.zeros((n, f))
    for v in range(n):
        nb = g.row(v)
        if len(nb):
            lo[v] = X[nb].min(axis=0)
            hi[v] = X[nb].max(axis=0)
    return lo, hi


def aggregate_fast(g: OracleGraph, X: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """Same values up to fp64 re-association; scipy CSR @ dense.  Used for large parity cases."""
    import scipy.sparse as sp
    A = sp.csr_matrix((np.ones(g.nnz), g.col, g.row_ptr), shape=(g.n, g.n))
    s = A @ X
    deg = np.diff(g.row_ptr).astype(np.float64)
    m = np.divide(s, deg[:, None], out=np.zeros_like(s), where=deg[:, None] > 0)
    return s, m


# --------------------------------------------------------------------------------------
# pruning
# --------------------------------------------------------------------------------------
def vertical_log_binning(arr, frac: float = 0.5) -> np.ndarray:
    """
    features/prune.py:13-56.  Sorted uniques + cumulative counts; repeatedly take
    max(int(frac * unbinned), 1) more values, extend to the end of the tie run, label that
    half-open value interval (bin_min, bin_max] with the running bin index.
    """
    if not 0 < frac < 1:
        raise ValueError('must specify frac in interval (0, 1)')
    arr = np.asarray(arr)
    n = len(arr)
    binned = np.zeros(n, dtype=np.int64)
    if n == 0:
        return binned
    uniq, counts = np.unique(arr, return_counts=True)
    cum = np.cumsum(counts)
    done = 0
    lo = -np.inf
    b = 0
    while done < n:
        size = max(int(frac * (n - done)), 1)
        j = int(np.searchsorted(cum, done + size, side='left'))
        hi = uniq[j]
        binned[(arr > lo) & (arr <= hi)] = b
        done = int(cum[j])
        lo = hi
        b += 1
    return binned


def chebyshev_matrix(B: np.ndarray) -> np.ndarray:
    """features/prune.py:108: max_i |B[i,p] - B[i,q]| for every column pair (int64 F x F)."""
    F = B.shape[1]
    D = np.zeros((F, F), dtype=np.int64)
    Bi = B.astype(np.int64)
    for p in range(F):
        for q in range(p + 1, F):
            d = int(np.abs(Bi[:, p] - Bi[:, q]).max()) if B.shape[0] else 0
            D[p, q] = D[q, p] = d
    return D


def connected_components(edges: Sequence[Tuple]) -> List[Set]:
    """graph/graph.py:18-57: components over nodes that appear in at least one edge."""
    adj: Dict = {}
    for a, b in edges:
        adj.setdefault(a, set()).add(b)
        adj.setdefault(b, set()).add(a)
    seen: Set = set()
    comps = []
    for start in adj:
        if start in seen:
            continue
        comp = set()
        stack = [start]
        while stack:
            x = stack.pop()
            if x in comp:
                continue
            comp.add(x)
            stack.extend(adj[x] - comp)
        seen |= comp
        comps.append(comp)
    return comps


def group_features(names: Sequence[str], D: np.ndarray, thresh: int) -> List[Set[str]]:
    """features/prune.py:94-116: edge (p,q) iff D[p,q] <= thresh; connected components."""
    F = len(names)
    edges = [(names[p], names[q]) for p in range(F) for q in range(p + 1, F) if D[p, q] <= thresh]
    return connected_components(edges)


def oldest_feature(group: Set[str], generation_names: Dict[int, Sequence[str]]) -> str:
    """features/prune.py:118-139: earliest recorded generation, ties -> smallest name."""
    for gen in range(len(generation_names)):
        cur = group.intersection(generation_names[gen])
        if cur:
            return min(cur)
    return min(group)


def prune_features(names: Sequence[str], D: np.ndarray, thresh: int,
                   generation_names: Dict[int, Sequence[str]]) -> Set[str]:
    """features/prune.py:76-92: per component keep the oldest member, drop the rest."""
    drop: Set[str] = set()
    for group in group_features(names, D, thresh):
        if len(group) == 1:
            continue
        drop |= group - {oldest_feature(group, generation_names)}
    return drop


# --------------------------------------------------------------------------------------
# driver
# --------------------------------------------------------------------------------------
@dataclass
class GenerationTrace:
    generation: int
    candidates: List[str]          # new columns offered this generation, in offered order
    working_before: List[str]      # pruner input columns (old working set + candidates)
    dropped: List[str]             # sorted
    retained: List[str]            # recorded for this generation (order as the reference records)
    working_after: List[str]


@dataclass
class RefexResult:
    labels: list
    columns: List[str]             # final column order (latest generation first)
    values: np.ndarray             # n x len(columns) float64
    generation_count: int
    trace: List[GenerationTrace]
    gen0_is_int: bool


def extract_features(g: OracleGraph, max_generations: int = 10, fast: bool = False,
                     gen0: Optional[Tuple[List[str], np.ndarray]] = None,
                     aggs: Sequence[str] = ('sum', 'mean')) -> RefexResult:
    """
    features/extract.py:65-142.  ``fast=True`` swaps the numpy loops for the plain-C twins.  Working set = dict name -> column; ``final[g]`` = names recorded
    at generation g (their values are frozen copies).
    """
    if fast:
        from . import ckernels
        agg_fn = lambda gg, X: ckernels.aggregate(gg.row_ptr, gg.col if gg.adj_col is None else gg.adj_col, X)
        minmax_fn = lambda gg, X: ckernels.aggregate_minmax(gg.row_ptr, gg.col, X)
        var_fn = lambda gg, X: ckernels.aggregate_var(gg.row_ptr, gg.col if gg.adj_col is None else gg.adj_col, X)
        prod_fn = lambda gg, X: ckernels.aggregate_prod(gg.row_ptr, gg.col if gg.adj_col is None else gg.adj_col, X)
        bin_fn = ckernels.vertical_log_binning
        cheb_fn = lambda B: ckernels.chebyshev(B.T)
    else:
        agg_fn, bin_fn, cheb_fn = aggregate, vertical_log_binning, chebyshev_matrix
        minmax_fn = aggregate_minmax
        var_fn = aggregate_var
        prod_fn = aggregate_prod
    names0, X0 = gen0 if gen0 is not None else neighborhood_features(g, fast)
    work: Dict[str, np.ndarray] = {}
    final_names: Dict[int, List[str]] = {}
    final_vals: Dict[str, np.ndarray] = {}
    trace: List[GenerationTrace] = []

    def update(gen: int, cand_names: List[str], cand_vals: np.ndarray, thresh: int):
        before = list(work)                      # extract.py:128-133 concat keeps old then new
        for j, nm in enumerate(cand_names):
            work[nm] = cand_vals[:, j]
        cols = list(work)
        if fast and g.n >= 200_000 and len(cols) > 1 and BIN_THREADS != 1:
            # the columns are binned independently (each call sorts a private copy in C, the GIL is released): threads
            # change the wall-clock of the full-size tests, not a value
            from concurrent.futures import ThreadPoolExecutor
            with ThreadPoolExecutor(max_workers=BIN_THREADS or min(32, os.cpu_count() or 1)) as pool:
                binned = list(pool.map(lambda c: bin_fn(work[c]), cols))
            B = np.column_stack(binned)
        else:
            B = np.column_stack([bin_fn(work[c]) for c in cols]) if cols else np.zeros((g.n, 0))
        D = cheb_fn(B)
        drop = prune_features(cols, D, thresh, final_names)          # prune.py:76
        for nm in drop:
            del work[nm]                                              # extract.py:137
        # extract.py:140 Index.difference: name-sorted iff something was dropped (pandas 2.x)
        kept = [c for c in cand_names if c not in drop]
        retained = sorted(kept) if drop else kept
        final_names[gen] = retained
        for nm in retained:
            final_vals[nm] = work[nm].copy()
        trace.append(GenerationTrace(gen, list(cand_names), before + list(cand_names),
                                     sorted(drop), list(retained), list(work)))

    update(0, names0, X0, 0)
    generation_count = 0
    for gen in range(1, max_generations):                             # extract.py:77
        generation_count = gen
        prev = final_names[gen - 1]
        Xp = np.column_stack([work[c] for c in prev]) if prev else np.zeros((g.n, 0))
        s, m = agg_fn(g, Xp)
        blocks = {'sum': s, 'mean': m}
        if 'min' in aggs or 'max' in aggs:
            blocks['min'], blocks['max'] = minmax_fn(g, Xp)
        if 'var' in aggs or 'std' in aggs:
            blocks['var'], blocks['std'] = var_fn(g, Xp)
        if 'prod' in aggs:
            blocks['prod'] = prod_fn(g, Xp)
        cand_names = [f'{c}({a})' for a in aggs for c in prev]        # extract.py:152-162
        cand_vals = np.column_stack([blocks[a] for a in aggs]) if prev else np.zeros((g.n, 0))
        update(gen, cand_names, cand_vals, gen)
        if not final_names[gen]:                                      # extract.py:86-87
            break
    columns: List[str] = []
    for gen in sorted(final_names, reverse=True):                     # extract.py:95 ChainMap order
        columns.extend(final_names[gen])
    values = np.column_stack([final_vals[c] for c in columns]) if columns else np.zeros((g.n, 0))
    return RefexResult(g.labels, columns, values, generation_count, trace, g.w is None)


# --------------------------------------------------------------------------------------
# any aggregation list, with the reference's dtypes (test infrastructure like the rest of this file)
# --------------------------------------------------------------------------------------
FLOAT_AGGS = ('mean', 'std', 'var', 'median')
INT_SAFE_ON_EMPTY = ('sum', 'prod', 'count', 'size')


def _agg_one(vals: np.ndarray, agg: str):
    """One pandas aggregation of one node's neighbour values (features/extract.py:108-113 ->
    DataFrame.agg -> Series.<agg>), restated with the numpy calls pandas' nanops make (pandas 2.x without
    bottleneck): integer columns add and multiply in wrapping int64; mean / var / std / median work on the fp64
    conversion; NaN where pandas gives NaN (no neighbours)."""
    n = len(vals)
    is_int = vals.dtype.kind in 'iu'
    if agg == 'sum':
        return vals.sum() if n else (0 if is_int else 0.0)
    if agg == 'prod':
        return np.multiply.reduce(vals) if n else (1 if is_int else 1.0)
    if agg in ('count', 'size'):
        return n
    if n == 0:
        return np.nan
    if agg == 'min':
        return vals.min()
    if agg == 'max':
        return vals.max()
    x = np.ascontiguousarray(vals, dtype=np.float64)
    if agg == 'mean':
        return x.sum() / n
    if agg == 'median':
        return float(np.median(x))
    if agg in ('var', 'std'):
        if n < 2:
            return np.nan
        avg = x.sum() / n
        var = np.ascontiguousarray((avg - x) ** 2).sum() / (n - 1)
        return var if agg == 'var' else float(np.sqrt(var))
    raise ValueError(agg)


@dataclass
class TypedResult:
    columns: List[str]
    arrays: Dict[str, np.ndarray]      # final columns, int64 or float64 like the reference's frame
    generation_count: int
    retained: Dict[int, List[str]]


def extract_features_typed(g: OracleGraph, names0: Sequence[str], cols0: Sequence[np.ndarray],
                           max_generations: int = 10, aggs: Sequence[str] = ('sum', 'mean')) -> TypedResult:
    """
    features/extract.py:65-142 for ANY list of pandas aggregation names, keeping the dtypes the reference's frames
    have: cols0 are the generation-0 columns with their reference dtypes (int64 for the degree / ego-net columns of an
    unweighted graph).  A candidate column is int64 iff its parent is and every entry of every node's agg frame for
    it is an integer (no mean / std / var / median among the aggs, and no node without neighbours unless all aggs are
    integer-valued on nothing); integer sums and products WRAP modulo 2^64 (numpy int64).  Pure-Python loop over the
    nodes: small cases only.
    """
    n = g.n
    deg = np.diff(g.row_ptr)
    no_empty = bool(n == 0 or deg.min() > 0)
    work: Dict[str, np.ndarray] = {}
    final_names: Dict[int, List[str]] = {}
    final_vals: Dict[str, np.ndarray] = {}

    def update(gen, cand_names, cand_cols, thresh):
        for nm, col in zip(cand_names, cand_cols):
            work[nm] = col
        cols = list(work)
        B = np.column_stack([vertical_log_binning(work[c]) for c in cols]) if cols else np.zeros((n, 0))
        D = chebyshev_matrix(B)
        drop = prune_features(cols, D, thresh, final_names)
        for nm in drop:
            del work[nm]
        kept = [c for c in cand_names if c not in drop]
        final_names[gen] = sorted(kept) if drop else kept
        for nm in final_names[gen]:
            final_vals[nm] = work[nm].copy()

    update(0, list(names0), [np.asarray(c) for c in cols0], 0)
    generation_count = 0
    for gen in range(1, max_generations):
        generation_count = gen
        prev = final_names[gen - 1]
        cand_names = [f'{c}({a})' for a in aggs for c in prev]
        cand_cols = []
        for a in aggs:
            for c in prev:
                parent = work[c]
                keeps_int = (parent.dtype.kind in 'iu' and not (set(FLOAT_AGGS) & set(aggs))
                             and (no_empty or set(aggs) <= set(INT_SAFE_ON_EMPTY)))
                out = np.empty(n, dtype=np.int64 if keeps_int else np.float64)
                for v in range(n):
                    val = _agg_one(parent[g.adj_row(v)], a)
                    if isinstance(val, float) and val != val:
                        val = 0                                   # fillna(0), extract.py:113
                    out[v] = val                                  # an int64 result in a float frame: cast like pandas
                cand_cols.append(out)
        update(gen, cand_names, cand_cols, gen)
        if not final_names[gen]:
            break
    columns: List[str] = []
    for gen in sorted(final_names, reverse=True):
        columns.extend(final_names[gen])
    return TypedResult(columns, {c: final_vals[c] for c in columns}, generation_count, final_names)
