"""
igraph adapter (reference: graphrole/graph/interface/igraph.py:19-205), written against igraph's
public Graph API (is_directed, is_weighted, vcount, ecount, get_edgelist, es['weight'],
vs.attribute_names, vs[name]); python-igraph itself is NOT imported -- it is absent from this
image, so the reference's igraph path could not be run here and this adapter's parity is
UNPINNED (tests drive it with a duck-typed stand-in and compare with the networkx adapter).

Scope: simple graphs.  The reference's igraph conventions differ from its networkx ones on
self-loops (``Graph.neighbors`` lists an undirected loop twice, igraph.py:59; a weighted loop
counts once in the degree, igraph.py:160-162) and on parallel edges (repeated neighbours are
aggregated repeatedly, the weight dict keeps the last one, igraph.py:36-39); neither can be checked
without igraph, so such graphs raise NotImplementedError instead of guessing.
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
        return [int(j) for j in self.to_csr().neighbors(int(node))]

    def to_csr(self) -> CSRGraph:
        if self._csr is None:
            n = int(self.G.vcount())
            edges = np.asarray(self.G.get_edgelist(), dtype=np.int64).reshape(-1, 2)
            src, dst = edges[:, 0], edges[:, 1]
            if np.any(src == dst):
                raise NotImplementedError('igraph graphs with self-loops: the reference counts them differently '
                                          'from its networkx adapter and igraph is not available to pin that')
            key = (src * n + dst) if self.directed else (np.minimum(src, dst) * n + np.maximum(src, dst))
            if len(np.unique(key)) != len(key):
                raise NotImplementedError('igraph graphs with parallel edges are not supported (simple graphs only)')
            wts = None
            if self.weighted:
                raw = list(self.G.es['weight'])
                wts = np.asarray(raw, dtype=np.float64)
                if all(isinstance(x, Integral) for x in raw):
                    wts = wts.astype(np.int64)
            # Graph.neighbors(v, mode='out') -- the order the reference's neighbour sums run in (igraph.py:55-59,
            # features/extract.py:108-110) -- lists the neighbours in ascending vertex order whatever the order
            # of the edge list: that is the CSR's own column order, not the order of appearance CSRGraph
            # assumes for plain edge arrays
            csr = CSRGraph(n, src, dst, wts, self.directed, labels=list(range(n)), validate=False)
            csr.adj_col = csr.col.copy()
            self._csr = csr
        return self._csr

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
