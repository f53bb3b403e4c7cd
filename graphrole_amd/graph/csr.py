"""
CSRGraph -- array-native graph input for graphs that are too large to hold as networkx objects
(BASELINE configs 3-5: 1 M - 5 M nodes, 10 M - 100 M edges).  It is a *new* adapter key of the
reference's plug-in seam (graphrole/graph/interface/__init__.py:12-17): the extractor accepts it
wherever a networkx graph is accepted.

Host-side ingest: edge arrays -> CSR of the out-adjacency with rows in sorted-label
order and ascending column indices (+ the transposed CSR for directed graphs, used for the
weighted in-degree), plus ``adj_col``: the same rows with every node's neighbours in ADJACENCY
order -- the order the reference's ``G[node]`` lists them, which is the order its neighbour
sums are evaluated in (graphrole/features/extract.py:108-113).  For edge arrays that is the order
of appearance (what networkx would hold after add_edge(src[i], dst[i]) for i = 0, 1, ...).
"""
from __future__ import annotations

from typing import Dict, Optional, Sequence

import numpy as np


def _csr_from_coo(n: int, src: np.ndarray, dst: np.ndarray, w: Optional[np.ndarray]):
    """Rows ascending, columns ascending inside a row.  One int64 key sort (src * n + dst)."""
    key = src * np.int64(n) + dst
    if w is None:
        key.sort()
        w_sorted = None
    else:
        order = np.argsort(key, kind='stable')
        key = key[order]
        w_sorted = np.ascontiguousarray(w[order], dtype=np.float64)
    rows = key // n
    row_ptr = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(np.bincount(rows, minlength=n), out=row_ptr[1:])
    col = (key - rows * n).astype(np.int32)
    return row_ptr, col, w_sorted


def adjacency_order(n: int, src: np.ndarray, dst: np.ndarray, directed: bool) -> np.ndarray:
    """Out-adjacency rows (sorted-label row order) with neighbours by index of the incident edge:
    add_edge(u, v) appends v to adj[u] and u to adj[v] (once for a self-loop)."""
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


class CSRGraph:
    """
    :param n: number of nodes (row i <-> labels[i]; labels default to 0..n-1 and must be sorted)
    :param src, dst: unique edges as row indices; an undirected edge is listed once
    :param weights: optional edge weights (None = every edge has the implicit weight 1); an
      integer-typed array keeps the degree / ego-net columns int64 like networkx does
    :param directed: arcs src -> dst when True
    :param attributes: optional {name: array of n numbers} numeric node attributes
    :param adjacency: optional int array [nnz]: every row's neighbours (row indices) in the order
      the neighbour sums must be evaluated in, rows concatenated in row order; default = order of
      appearance in (src, dst).  The networkx adapter passes ``G[node]`` order.
    """

    def __init__(self, n: int, src, dst, weights=None, directed: bool = False,
                 labels: Optional[Sequence] = None, attributes: Optional[Dict[str, np.ndarray]] = None,
                 validate: bool = True, adjacency=None) -> None:
        src = np.ascontiguousarray(src, dtype=np.int64)
        dst = np.ascontiguousarray(dst, dtype=np.int64)
        if src.shape != dst.shape or src.ndim != 1:
            raise ValueError('src and dst must be 1-d arrays of equal length')
        # integer-typed weights keep generation-0 features integer (networkx sums Python ints)
        self.integral = weights is None or np.asarray(weights).dtype.kind in 'iub'
        w = None if weights is None else np.ascontiguousarray(weights, dtype=np.float64)
        if w is not None and w.shape != src.shape:
            raise ValueError('weights must match the edge arrays')
        if n >= 2 ** 31:
            raise ValueError('CSRGraph supports fewer than 2^31 nodes (int32 column indices)')
        if validate and len(src):
            if src.min() < 0 or dst.min() < 0 or src.max() >= n or dst.max() >= n:
                raise ValueError('edge endpoint outside [0, n)')
            a, b = (src, dst) if directed else (np.minimum(src, dst), np.maximum(src, dst))
            if len(np.unique(a * n + b)) != len(src):
                raise ValueError('duplicate edges: merge parallel edges before building a CSRGraph')
        self.n = int(n)
        self.directed = bool(directed)
        self.weighted = w is not None
        self.num_edges = int(len(src))
        # default labels stay a range: a million-entry list (and the pandas Index built from it for every result
        # table) costs ~0.1 s that the rest of the pipeline no longer has
        self.labels = range(int(n)) if labels is None else list(labels)
        if len(self.labels) != n:
            raise ValueError('labels must have n entries')
        self._label_index = None
        # The host CSR is built on first use: the device path ingests the edge arrays directly
        # (grx_ingest, graphrole_amd/kernels.py::device_ingest) and never needs it.
        self._edges = (src, dst, w)
        self._adjacency = None if adjacency is None else np.ascontiguousarray(adjacency, dtype=np.int32)
        self._host = None
        n_loops = int(np.count_nonzero(src == dst))
        self.n_loops = n_loops                              # 0: the degree kernels skip their search for the diagonal
        self._nnz = int(len(src)) if directed else 2 * int(len(src)) - n_loops
        if self._adjacency is not None and self._adjacency.shape != (self._nnz,):
            raise ValueError('adjacency must list every neighbour of every row exactly once')
        if validate and self._adjacency is not None and self._nnz:
            rows = np.repeat(np.arange(n, dtype=np.int64), np.diff(self.row_ptr))
            if not np.array_equal(np.sort(rows * n + self.adj_col), rows * n + self.col):
                raise ValueError('adjacency rows must be permutations of the neighbour sets')
        self.attributes: Dict[str, np.ndarray] = {}
        for name, values in (attributes or {}).items():
            values = np.asarray(values)
            if values.shape != (n,):
                raise ValueError(f'attribute {name!r} must have shape ({n},)')
            self.attributes[name] = values

    # ------------------------------------------------------------------ host CSR (lazy)
    def _build_host(self) -> dict:
        if self._host is None:
            n, (src, dst, w) = self.n, self._edges
            h = {}
            if self.directed:
                h['row_ptr'], h['col'], h['w'] = _csr_from_coo(n, src, dst, w)
                h['t_row_ptr'], h['t_col'], h['t_w'] = _csr_from_coo(n, dst, src, w)
            else:
                off = src != dst
                s2 = np.concatenate([src, dst[off]])
                d2 = np.concatenate([dst, src[off]])
                w2 = None if w is None else np.concatenate([w, w[off]])
                h['row_ptr'], h['col'], h['w'] = _csr_from_coo(n, s2, d2, w2)
                h['t_row_ptr'] = h['t_col'] = h['t_w'] = None
            h['adj_col'] = self._adjacency if self._adjacency is not None else adjacency_order(n, src, dst, self.directed)
            self._host = h
        return self._host

    row_ptr = property(lambda self: self._build_host()['row_ptr'])
    col = property(lambda self: self._build_host()['col'])
    w = property(lambda self: self._build_host()['w'])
    t_row_ptr = property(lambda self: self._build_host()['t_row_ptr'])
    t_col = property(lambda self: self._build_host()['t_col'])
    t_w = property(lambda self: self._build_host()['t_w'])

    @property
    def adj_col(self) -> np.ndarray:
        return self._build_host()['adj_col']

    @adj_col.setter
    def adj_col(self, value) -> None:
        value = np.ascontiguousarray(value, dtype=np.int32)
        self._build_host()['adj_col'] = value
        self._adjacency = value                       # an explicit order: the device ingest must not assume appearance order

    def edge_arrays(self):
        """(src, dst, weights or None) as given -- what the device ingest consumes -- or None when an explicit
        neighbour order was supplied (then the host CSR is the source of truth)."""
        return None if self._adjacency is not None else self._edges

    @property
    def nnz(self) -> int:
        return self._nnz

    def neighbors(self, row: int) -> np.ndarray:
        return self.col[self.row_ptr[row]:self.row_ptr[row + 1]]

    # ------------------------------------------------------------------ other array-native inputs
    def label_index(self):
        """pandas Index of the node labels (row order), built once."""
        if self._label_index is None:
            import pandas as pd
            if isinstance(self.labels, range):
                self._label_index = pd.Index(np.arange(self.n, dtype=np.int64))
            else:
                self._label_index = pd.Index(self.labels)
        return self._label_index

    @classmethod
    def from_scipy_sparse(cls, A, directed: bool = False, weighted: Optional[bool] = None,
                          labels: Optional[Sequence] = None,
                          attributes: Optional[Dict[str, np.ndarray]] = None) -> 'CSRGraph':
        """
        Adjacency matrix (any scipy.sparse format, square; entry (i, j) = weight of i -> j).
        Undirected: the matrix must be symmetric, every edge is taken once from the upper triangle.
        ``weighted=None`` treats the matrix as unweighted iff every stored value equals 1.
        The neighbour order of the sums is the stored order of each row of ``A.tocsr()``.
        """
        import scipy.sparse as sp
        A = sp.csr_matrix(A)
        n = A.shape[0]
        if A.shape[0] != A.shape[1]:
            raise ValueError('adjacency matrix must be square')
        A.sum_duplicates()
        rows = np.repeat(np.arange(n, dtype=np.int64), np.diff(A.indptr))
        cols = A.indices.astype(np.int64)
        data = np.asarray(A.data)
        if weighted is None:
            weighted = bool(len(data)) and not np.all(data == 1)
        if directed:
            src, dst, w = rows, cols, data
        else:
            if (abs(A - A.T) > 0).nnz:
                raise ValueError('undirected input needs a symmetric adjacency matrix')
            upper = rows <= cols
            src, dst, w = rows[upper], cols[upper], data[upper]
        return cls(n, src, dst, weights=w if weighted else None, directed=directed, labels=labels,
                   attributes=attributes, validate=False, adjacency=A.indices)

    @classmethod
    def from_edgelist_file(cls, path: str, directed: bool = False, weighted: bool = False,
                           comments: str = '#', delimiter: Optional[str] = None) -> 'CSRGraph':
        """
        Text edge list, one ``u v [w]`` per line with integer node ids (any integers: the sorted unique
        ids become the labels).  Parallel edges are merged by summing their weights (unweighted: dropped),
        an undirected edge may appear in either orientation.
        """
        table = np.loadtxt(path, comments=comments, delimiter=delimiter, ndmin=2,
                           dtype=np.float64 if weighted else np.int64)
        if table.shape[1] < (3 if weighted else 2):
            raise ValueError(f'{path}: expected {"u v w" if weighted else "u v"} per line')
        u, v = table[:, 0].astype(np.int64), table[:, 1].astype(np.int64)
        labels, inverse = np.unique(np.concatenate([u, v]), return_inverse=True)
        n = len(labels)
        src, dst = inverse[:len(u)].astype(np.int64), inverse[len(u):].astype(np.int64)
        w = table[:, 2] if weighted else None
        a, b = (src, dst) if directed else (np.minimum(src, dst), np.maximum(src, dst))
        key = a * n + b
        first = np.unique(key, return_index=True)[1]
        if len(first) != len(key):                                  # merge parallel edges, keep first position
            order = np.sort(first)
            if weighted:
                uniq, inv = np.unique(key, return_inverse=True)
                wsum = np.bincount(inv, weights=w, minlength=len(uniq))
                w = wsum[np.searchsorted(uniq, key[order])]
            src, dst = a[order], b[order]
        return cls(n, src, dst, weights=w, directed=directed, labels=[int(x) for x in labels], validate=False)


class InternalGraph:
    """
    The device-side view of a CSRGraph: rows relabelled in DEGREE-DESCENDING order (ties by
    label order).  High-degree rows -- the hot gather targets and the hub work list -- become a
    contiguous prefix, and consecutive rows have similar degrees, which is what lets the
    aggregation kernels slice the adjacency without padding blow-up.  ``perm[i]`` is the
    label-order row of internal row i, ``inv`` its inverse; every per-node device array is in
    internal order and is mapped back with ``to_label_order`` at the host boundary only.
    """

    def __init__(self, g: CSRGraph) -> None:
        n = g.n
        deg = np.diff(g.row_ptr)
        self.perm = np.argsort(-deg, kind='stable').astype(np.int64)
        self.inv = np.empty(n, dtype=np.int64)
        self.inv[self.perm] = np.arange(n, dtype=np.int64)
        self.n, self.directed, self.weighted, self.integral = n, g.directed, g.weighted, g.integral
        self.labels, self.num_edges = g.labels, g.num_edges
        rows = np.repeat(np.arange(n, dtype=np.int64), deg)
        self.row_ptr, self.col, self.w = _csr_from_coo(n, self.inv[rows], self.inv[g.col], g.w)
        # neighbour lists in the reference's visiting order, rows permuted, ids relabelled
        new_deg = deg[self.perm]
        src_pos = (np.arange(len(g.col), dtype=np.int64) - np.repeat(self.row_ptr[:-1], new_deg)
                   + np.repeat(g.row_ptr[:-1][self.perm], new_deg))
        self.agg_col = self.inv[g.adj_col[src_pos]].astype(np.int32)
        if g.directed:
            trows = np.repeat(np.arange(n, dtype=np.int64), np.diff(g.t_row_ptr))
            self.t_row_ptr, self.t_col, self.t_w = _csr_from_coo(n, self.inv[trows], self.inv[g.t_col], g.t_w)
        else:
            self.t_row_ptr = self.t_col = self.t_w = None

    @property
    def nnz(self) -> int:
        return int(self.row_ptr[-1])

    def to_internal(self, values: np.ndarray) -> np.ndarray:
        """label-order per-node array -> internal order"""
        return np.ascontiguousarray(np.asarray(values)[self.perm])

    def to_label_order(self, values: np.ndarray) -> np.ndarray:
        """internal-order per-node array -> label order"""
        return np.ascontiguousarray(np.asarray(values)[..., self.inv])


class DeviceBuiltGraph:
    """
    Host view of a graph whose CSR was built in HBM (kernels.device_ingest): the same attributes the engine
    reads from an InternalGraph -- shape, flags, labels, the internal order and the row pointers -- without the
    column arrays, which exist on the device only.
    """

    def __init__(self, g: CSRGraph, perm, inv, row_ptr: np.ndarray) -> None:
        # perm / inv: int32 DEVICE tensors (the internal order never has to leave HBM on the hot path: the result table
        # is permuted on the device); host copies are fetched on first use (attribute columns, adapter views)
        self.perm_dev, self.inv_dev, self.row_ptr = perm, inv, row_ptr
        self._perm = self._inv = None
        self.n, self.directed, self.weighted, self.integral = g.n, g.directed, g.weighted, g.integral
        self.labels, self.num_edges = g.labels, g.num_edges

    @staticmethod
    def _fetch(t) -> np.ndarray:
        return np.ascontiguousarray(t.cpu().numpy() if hasattr(t, 'cpu') else t).astype(np.int64)

    @property
    def perm(self) -> np.ndarray:
        if self._perm is None:
            self._perm = self._fetch(self.perm_dev)
        return self._perm

    @property
    def inv(self) -> np.ndarray:
        if self._inv is None:
            self._inv = self._fetch(self.inv_dev)
        return self._inv

    @property
    def nnz(self) -> int:
        return int(self.row_ptr[-1])

    def to_internal(self, values: np.ndarray) -> np.ndarray:
        return np.ascontiguousarray(np.asarray(values)[self.perm])

    def to_label_order(self, values: np.ndarray) -> np.ndarray:
        return np.ascontiguousarray(np.asarray(values)[..., self.inv])
