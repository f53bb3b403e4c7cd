"""
NetworkX adapter (reference: graphrole/graph/interface/networkx.py:13-123).  The graph is read
ONCE into a host CSR (no per-node Python during extraction); degree and ego-net features come
from HIP kernels.  Attribute selection is host-side metadata handling.
"""
from __future__ import annotations

from numbers import Integral, Number
from typing import Dict, Iterable, Optional

import networkx as nx
import numpy as np
import pandas as pd

from graphrole_amd.graph.csr import CSRGraph
from graphrole_amd.graph.interface.base import DeviceGraphInterface
from graphrole_amd.types import Node


class NetworkxInterface(DeviceGraphInterface):

    def __init__(self, G: nx.Graph, **kwargs) -> None:
        """
        :param G: networkx Graph or DiGraph (edge attribute 'weight' is honoured, default 1)
        :kwarg attributes: use numeric node attributes as features
        :kwarg attributes_include: only these attributes (default: all)
        :kwarg attributes_exclude: never these attributes (wins over include)
        """
        if G.is_multigraph():
            # the reference runs on a Multi(Di)Graph but mixes conventions there (weighted degrees sum the
            # parallel edges, its ego-net sums count each of them as 1, networkx.py:54-62,115-123)
            raise NotImplementedError('networkx MultiGraph / MultiDiGraph input: merge the parallel edges first '
                                      '(e.g. nx.Graph(G), or sum their weights into one edge)')
        self.G = G
        self.directed = G.is_directed()
        self._set_attribute_kwargs(**kwargs)
        self._csr: Optional[CSRGraph] = None

    def get_num_edges(self) -> int:
        return self.G.number_of_edges()

    def get_nodes(self) -> Iterable[Node]:
        return self.G.nodes

    def get_neighbors(self, node: Node) -> Iterable[Node]:
        return self.G[node].keys()

    def to_csr(self) -> CSRGraph:
        if self._csr is None:
            labels = sorted(self.G.nodes)
            row_of = {label: i for i, label in enumerate(labels)}
            m = self.G.number_of_edges()
            src = np.empty(m, dtype=np.int64)
            dst = np.empty(m, dtype=np.int64)
            wts = np.ones(m, dtype=np.float64)
            weighted = False
            integral = True
            for k, (u, v, weight) in enumerate(self.G.edges(data='weight')):
                src[k] = row_of[u]
                dst[k] = row_of[v]
                if weight is not None:
                    wts[k] = weight
                    weighted = True
                    integral = integral and isinstance(weight, Integral)
            if weighted and integral:
                wts = wts.astype(np.int64)
            # G[node] iteration order = the order the reference's reindex(nbrs) sums in
            # (features/extract.py:108-110); G.edges is grouped by node, so read adj itself
            adjacency = np.fromiter((row_of[v] for label in labels for v in self.G.adj[label]), dtype=np.int32)
            self._csr = CSRGraph(len(labels), src, dst, wts if weighted else None, self.directed,
                                 labels=labels, validate=False, adjacency=adjacency)
        return self._csr

    def _attribute_frame(self) -> Optional[pd.DataFrame]:
        """networkx.py:87-113: numeric values only; with an include list a missing value counts
        as 0, otherwise missing entries are filled with 0 later; exclude beats include."""
        banned = set(self._attrs_exclude)
        table: Dict[str, Dict[Node, Number]] = {}
        if self._attrs_include:
            for attr in self._attrs_include:
                if attr in banned:
                    continue
                column = {}
                for node, data in self.G.nodes(data=True):
                    value = data.get(attr, 0)
                    if isinstance(value, Number):
                        column[node] = value
                table[self._attribute_feature_name(attr)] = column
        else:
            for node, data in self.G.nodes(data=True):
                for attr, value in data.items():
                    if attr in banned or not isinstance(value, Number):
                        continue
                    table.setdefault(self._attribute_feature_name(attr), {})[node] = value
        return pd.DataFrame(table).fillna(0)
