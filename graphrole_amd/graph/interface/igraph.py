"""
igraph adapter (reference: graphrole/graph/interface/igraph.py:19-205), written against igraph's
public Graph API (is_directed, is_weighted, vcount, ecount, get_edgelist, es['weight'],
vs.attribute_names, vs[name]); python-igraph itself is NOT imported -- it is absent from this
image, so the reference's igraph path could not be run here and this adapter's parity is
UNPINNED (tests drive it with a duck-typed stand-in and compare with the networkx adapter).

Graphs with self-loops or parallel edges follow the reference's igraph conventions as its code reads
(igraph.py:36-39, 55-59, 78-98, 129-205) on top of igraph's documented behaviour -- ASSUMED, since igraph is not
here to confirm it: ``Graph.neighbors(v, mode='out')`` lists neighbours in ascending vertex order, once per
parallel edge, an undirected loop twice; ``Vertex.degree(mode)`` counts a loop twice (once each for 'in' / 'out');
``Edge.tuple`` reports an undirected edge with a fixed orientation.  Then:
  * neighbour aggregation runs over that neighbour MULTISET (extract.py:107-110 reindexes with repeated labels);
  * ``edge_weights`` is a dict keyed by the edge tuple: parallel edges collapse to one entry holding the LAST
    weight (1 when unweighted), and the ego-net sums and the weighted degrees read that dict -- a loop counts once
    in the weighted total degree (``node in edge``) and once in each of in / out;
  * unweighted degrees count edge ends of the multigraph.
The oracle restatement of exactly this reading is oracle/igraph_path.py (tests/test_igraph_adapter_cpu.py).
On simple graphs both reference adapters define the same features; nodes are the vertex indices
(igraph.py:49-53) and the reserved vertex attribute 'name' is never a feature (igraph.py:14-16).
"""
from __future__ import annotations

from numbers import Integral, Number
from typing import Dict, Iterable, Optional

import numpy as np
import pandas as pd

from graphrole_amd.graph.csr import CSRGraph
from graphrole_amd.graph.interface.base import DeviceGraphInterface
from graphrole_amd.types import Node

IGRAPH_RESERVED_ATTRIBUTE_NAMES = {'name'}


class IgraphInterface(DeviceGraphInterface):

    def __init__(self, G, **kwargs) -> None:
        """
        :param G: igraph Graph (edge attribute 'weight' is honoured, default 1)
        :kwarg attributes / attributes_include / attributes_exclude: as for the networkx adapter
        """
        self.G = G
        self.directed = bool(G.is_directed())
        self.weighted = bool(G.is_weighted())
        self._set_attribute_kwargs(**kwargs)
        self._csr: Optional[CSRGraph] = None

    def get_num_edges(self) -> int:
        return int(self.G.ecount())

    def get_nodes(self) -> Iterable[Node]:
        return list(range(int(self.G.vcount())))

    def get_neighbors(self, node: Node) -> Iterable[Node]:
        m_ptr, m_col = self._neighbour_multiset()
        return [int(j) for j in m_col[m_ptr[int(node)]:m_ptr[int(node) + 1]]]

    # ------------------------------------------------------------------ edge views
    def _edges(self):
        """(src, dst, weights or None) of every edge as igraph lists them; undirected tuples as (min, max)."""
        if getattr(self, '_edge_cache', None) is None:
            edges = np.asarray(self.G.get_edgelist(), dtype=np.int64).reshape(-1, 2)
            src, dst = edges[:, 0].copy(), edges[:, 1].copy()
            if not self.directed:
                src, dst = np.minimum(src, dst), np.maximum(src, dst)
            wts = None
            if self.weighted:
                raw = list(self.G.es['weight'])
                wts = np.asarray(raw, dtype=np.float64)
                if all(isinstance(x, Integral) for x in raw):
                    wts = wts.astype(np.int64)
            self._edge_cache = (src, dst, wts)
        return self._edge_cache

    def _is_simple(self) -> bool:
        """No loops, no parallel edges -- a property of the wrapped graph, computed once (the O(m log m) np.unique ran
        on every _device_graph() call of a generation before)."""
        if getattr(self, '_simple', None) is None:
            src, dst, _ = self._edges()
            n = int(self.G.vcount())
            self._simple = bool(not np.any(src == dst) and len(np.unique(src * n + dst)) == len(src))
        return self._simple

    def _dict_edges(self):
        """The reference's ``edge_weights`` dict (igraph.py:36-39): one entry per distinct tuple, the weight of its
        LAST occurrence (1 when the graph is unweighted)."""
        src, dst, wts = self._edges()
        n = int(self.G.vcount())
        key = src * n + dst
        # last occurrence of every key, entries kept in order of first appearance (dict semantics)
        _, first = np.unique(key, return_index=True)
        _, last_rev = np.unique(key[::-1], return_index=True)
        last = len(key) - 1 - last_rev                       # np.unique sorts by key: both arrays align
        order = np.argsort(first, kind='stable')
        pick_first, pick_last = first[order], last[order]
        w = None if wts is None else wts[pick_last]
        return src[pick_first], dst[pick_first], w

    def to_csr(self) -> CSRGraph:
        """Simple graphs: the graph itself.  With loops / parallel edges: the graph of the ``edge_weights`` dict (what
        the ego-net sums and weighted degrees read); the neighbour multiset of the aggregation is separate
        (_device_graph)."""
        if self._csr is None:
            n = int(self.G.vcount())
            if self._is_simple():
                src, dst, wts = self._edges()
            else:
                src, dst, wts = self._dict_edges()
            # Graph.neighbors(v, mode='out') -- the order the reference's neighbour sums run in (igraph.py:55-59,
            # features/extract.py:108-110) -- lists the neighbours in ascending vertex order whatever the order
            # of the edge list: that is the CSR's own column order, not the order of appearance CSRGraph
            # assumes for plain edge arrays
            csr = CSRGraph(n, src, dst, wts, self.directed, labels=list(range(n)), validate=False)
            csr.adj_col = csr.col.copy()
            self._csr = csr
        return self._csr

    # ------------------------------------------------------------------ loops / parallel edges
    def _neighbour_multiset(self):
        """CSR (row_ptr, col) of ``Graph.neighbors(v, mode='out')`` for every v: ascending ids, one entry per
        parallel edge, an undirected loop twice."""
        src, dst, _ = self._edges()
        n = int(self.G.vcount())
        if self.directed:
            rows, cols = src, dst
        else:
            rows, cols = np.concatenate([src, dst]), np.concatenate([dst, src])    # a loop contributes (v, v) twice
        order = np.lexsort((cols, rows))
        rows, cols = rows[order], cols[order]
        row_ptr = np.zeros(n + 1, dtype=np.int64)
        np.cumsum(np.bincount(rows, minlength=n), out=row_ptr[1:])
        return row_ptr, cols.astype(np.int64)

    def _device_graph(self):
        if self._is_simple():
            return super()._device_graph()
        if getattr(self, '_dev', None) is None:
            K = self._K()
            from graphrole_amd.graph.csr import InternalGraph
            host = InternalGraph(self.to_csr())                # internal order of the dict graph
            self._struct = K.DeviceCSR(host.row_ptr, host.col, host.w, agg_col=host.agg_col)
            # the neighbour multiset in the same internal order: row i = label perm[i], ids relabelled, igraph's order
            m_ptr, m_col = self._neighbour_multiset()
            deg = np.diff(m_ptr)[host.perm]
            a_ptr = np.zeros(host.n + 1, dtype=np.int64)
            np.cumsum(deg, out=a_ptr[1:])
            pos = (np.arange(int(a_ptr[-1]), dtype=np.int64) - np.repeat(a_ptr[:-1], deg)
                   + np.repeat(m_ptr[:-1][host.perm], deg))
            a_col = host.inv[m_col[pos]].astype(np.int32)
            agg = K.DeviceCSR(a_ptr, a_col, None, agg_col=a_col)
            self._dev = (host, agg, None)
        return self._dev

    def local_feature_columns(self):
        if self._is_simple():
            return super().local_feature_columns()
        K = self._K()
        host = self._device_graph()[0]
        n = host.n
        src, dst, wts = self._edges()
        if self.weighted:
            # igraph.py:155-162 on the edge_weights dict
            ds, dd, dw = self._dict_edges()
            dw = dw.astype(np.float64)
            outd = np.bincount(ds, weights=dw, minlength=n)
            ind = np.bincount(dd, weights=dw, minlength=n)
            loop = np.bincount(ds[ds == dd], weights=dw[ds == dd], minlength=n)
            total = outd + ind - loop                          # ``node in edge``: a loop once
            integral = bool(np.issubdtype(wts.dtype, np.integer))
        else:
            outd = np.bincount(src, minlength=n).astype(np.float64)
            ind = np.bincount(dst, minlength=n).astype(np.float64)
            total = outd + ind                                 # Vertex.degree(): a loop twice
            integral = True
        dt = np.dtype('int64') if integral else np.dtype('float64')
        if self.directed:
            names, arrays = ['in_degree', 'out_degree', 'total_degree'], [ind, outd, total]
        else:
            names, arrays = ['degree'], [total]
        cols = [K.to_device(host.to_internal(a)) for a in arrays]
        dtypes = [dt] * len(names)
        if self._attrs:
            attr = self._attribute_frame()
            if attr is not None and attr.shape[1]:
                attr = attr.reindex(host.labels).fillna(0)
                for name in attr.columns:
                    values = attr[name].to_numpy()
                    names.append(name)
                    dtypes.append(values.dtype if values.dtype.kind in 'iu' else np.dtype('float64'))
                    cols.append(K.to_device(host.to_internal(values.astype(np.float64))))
        return names, cols, dtypes

    def egonet_feature_columns(self):
        if self._is_simple():
            return super().egonet_feature_columns()
        K = self._K()
        host = self._device_graph()[0]
        st = self._struct                                     # the edge_weights dict as a graph (igraph.py:78-98)
        rowsum = K.row_sums(st, False) if host.weighted else None
        internal, external = K.egonet_features(st, host.directed, rowsum, 0, host.n)
        dt = np.dtype('int64') if host.integral else np.dtype('float64')
        return ['internal_edges', 'external_edges'], [internal, external], [dt, dt]

    def _attribute_frame(self) -> Optional[pd.DataFrame]:
        """igraph.py:100-127: numeric values only, vertex by vertex; with an include list a missing
        value counts as 0; exclude beats include; 'name' is reserved."""
        banned = set(self._attrs_exclude) | IGRAPH_RESERVED_ATTRIBUTE_NAMES
        present = list(self.G.vs.attribute_names())
        n = int(self.G.vcount())
        table: Dict[str, Dict[Node, Number]] = {}
        wanted = [a for a in (self._attrs_include or present) if a not in banned]
        for attr in wanted:
            if attr in present:
                values = list(self.G.vs[attr])
            elif self._attrs_include:
                values = [0] * n                                   # attributes().get(attr_name, 0)
            else:
                continue
            column = {i: v for i, v in enumerate(values) if isinstance(v, Number)}
            if column or self._attrs_include:
                table[self._attribute_feature_name(attr)] = column
        return pd.DataFrame(table, index=pd.RangeIndex(n)).fillna(0) if table else pd.DataFrame(index=pd.RangeIndex(n))
