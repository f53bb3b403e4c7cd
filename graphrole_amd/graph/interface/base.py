"""
Graph adapters: the plug-in seam of the reference (graphrole/graph/interface/base.py:9-83).

``BaseGraphInterface`` keeps the reference's five abstract methods and the concrete
``get_neighborhood_features``.  New here: every adapter exposes its graph in bulk through
``to_csr()`` so the engine never calls ``get_neighbors`` node by node, and the generation-0
features are computed by HIP kernels (grx_row_sums / grx_egonet_features) on the CSR in HBM.
"""
from __future__ import annotations

from abc import ABC, abstractmethod
from typing import Iterable, List, Optional, Tuple

import numpy as np
import pandas as pd

from graphrole_amd.graph.csr import CSRGraph, DeviceBuiltGraph, InternalGraph
from graphrole_amd.types import Node


class BaseGraphInterface(ABC):

    # prefix of feature names derived from node attributes (base.py:16)
    attribute_feature_prefix = 'attribute'

    # ------------------------------------------------------------------ reference API
    def get_neighborhood_features(self) -> pd.DataFrame:
        """Generation-0 block = local features then ego-net features, rows sorted by node
        (base.py:18-26)."""
        local = self._get_local_features()
        ego = self._get_egonet_features()
        return pd.concat([local, ego], axis=1, sort=True).sort_index()

    def _set_attribute_kwargs(self, **kwargs) -> None:
        """attributes / attributes_include / attributes_exclude (base.py:28-39)."""
        self._attrs: bool = kwargs.get('attributes', False)
        self._attrs_include: List[str] = kwargs.get('attributes_include', [])
        self._attrs_exclude: List[str] = kwargs.get('attributes_exclude', [])

    @classmethod
    def _attribute_feature_name(cls, attr_name: str) -> str:
        return f'{cls.attribute_feature_prefix}_{attr_name}'

    @abstractmethod
    def get_num_edges(self) -> int:
        """Number of edges of the wrapped graph."""

    @abstractmethod
    def get_nodes(self) -> Iterable[Node]:
        """Iterable over node labels."""

    @abstractmethod
    def get_neighbors(self, node: Node) -> Iterable[Node]:
        """Out-neighbours of one node."""

    @abstractmethod
    def _get_local_features(self) -> pd.DataFrame:
        """Degree (and optional attribute) features, one row per node."""

    @abstractmethod
    def _get_egonet_features(self) -> pd.DataFrame:
        """internal_edges / external_edges of every node's 1-hop ego-net."""

    # ------------------------------------------------------------------ engine API (new)
    @abstractmethod
    def to_csr(self) -> CSRGraph:
        """The whole graph as a host CSR (rows = sorted labels)."""


class DeviceGraphInterface(BaseGraphInterface):
    """
    Shared implementation of the generation-0 features on the GPU.  Subclasses provide
    ``to_csr()`` and ``_attribute_frame()``; everything numerical happens in libgrx.so.
    """

    def _K(self):
        from graphrole_amd import backend
        return backend.get()

    # -- device state -----------------------------------------------------------------
    def _device_graph(self):
        """(InternalGraph, device CSR, device transposed CSR or None); rows in internal
        (degree-descending) order -- see graphrole_amd.graph.csr.InternalGraph."""
        if getattr(self, '_dev', None) is None:
            K = self._K()
            g = self.to_csr()
            edges = g.edge_arrays() if hasattr(K, 'device_ingest') else None
            # grx_ingest packs (row, slot) into 64-bit sort keys: fewer than 2^30 edges; larger graphs (and explicit
            # neighbour orders) take the host construction + upload
            if edges is not None and 0 < g.num_edges < (1 << 30) and self._device_ingest:
                # the whole construction -- degree-descending relabelling, CSR in adjacency and in ascending
                # column order, weights, transposed CSR -- in HBM (grx_ingest); no host CSR is ever built
                src, dst, w = edges
                perm, inv, row_ptr, out, tr = K.device_ingest(g.n, src, dst, w, g.directed, g.nnz)
                host = DeviceBuiltGraph(g, perm, inv, row_ptr)
            else:
                host = InternalGraph(g)
                out = K.DeviceCSR(host.row_ptr, host.col, host.w, agg_col=host.agg_col)
                tr = K.DeviceCSR(host.t_row_ptr, host.t_col, host.t_w) if host.directed else None
            self._dev = (host, out, tr)
            # networkx counts an undirected self-loop twice in the degree (networkx.py:54): the kernel looks every row's
            # own index up -- a binary search per row that a graph without loops does not need
            self._has_loops = getattr(g, 'n_loops', 1) > 0
        return self._dev

    #: False forces the host-side construction (InternalGraph + upload); tests compare the two
    _device_ingest = True
    #: set by _device_graph; True = look the diagonal up (always right, slower)
    _has_loops = True

    def _row_range(self) -> Tuple[int, int]:
        """Rows this rank computes (whole graph unless a ShardPlan was attached)."""
        plan = getattr(self, '_shard_plan', None)
        host = self._device_graph()[0]
        return (0, host.n) if plan is None else (plan.row_begin, plan.row_end)

    def _finish_columns(self, cols):
        """Exchange row slices between ranks (no-op on a single GPU)."""
        plan = getattr(self, '_shard_plan', None)
        if plan is None:
            return cols
        return plan.all_gather_columns(cols)

    def local_feature_columns(self) -> Tuple[List[str], list, List[np.dtype]]:
        """
        Device columns of networkx.py:48-63: weighted ``degree`` (self-loop twice) or
        ``in_degree, out_degree, total_degree``; then the attribute columns.
        Returns (names, device columns, pandas dtypes of the reference's frame).
        """
        K = self._K()
        host, out, tr = self._device_graph()
        rb, re = self._row_range()
        int_dtype = np.dtype('int64') if host.integral else np.dtype('float64')
        if host.directed:
            outd = K.row_sums(out, False, rb, re)
            ind = K.row_sums(tr, False, rb, re)
            outd, ind = self._finish_columns([outd, ind])
            # the ego-net kernel wants the same out-weight sums: hand them over instead of a second pass
            self._out_rowsum = outd if host.weighted else None
            names = ['in_degree', 'out_degree', 'total_degree']
            cols = [ind, outd, K.add_columns(outd, ind)]
        else:
            deg, = self._finish_columns([K.row_sums(out, self._has_loops, rb, re)])
            names, cols = ['degree'], [deg]
        dtypes = [int_dtype] * len(names)
        if self._attrs:
            # attribute columns are graph data: uploaded once and cached on the adapter
            if getattr(self, '_attr_cols', None) is None:
                a_names, a_cols, a_dtypes = [], [], []
                arrays = self._attribute_arrays()
                if arrays is None:
                    attr = self._attribute_frame()
                    if attr is not None and attr.shape[1]:
                        attr = attr.reindex(host.labels).fillna(0)
                        arrays = {name: attr[name].to_numpy() for name in attr.columns}
                for name, values in (arrays or {}).items():
                    values = np.asarray(values)
                    a_names.append(name)
                    a_dtypes.append(values.dtype if values.dtype.kind in 'iu' else np.dtype('float64'))
                    perm_dev = getattr(host, 'perm_dev', None)
                    if perm_dev is not None and hasattr(K, 'permute_columns'):
                        # label order up, internal order by one gather in HBM (a host fancy-index of a few million
                        # doubles per attribute cost more than the whole generation loop)
                        up = K.to_device(np.ascontiguousarray(values, dtype=np.float64))
                        a_cols.append(K.permute_columns([up], perm_dev, host.n)[0])
                    else:
                        a_cols.append(K.to_device(host.to_internal(values.astype(np.float64))))
                self._attr_cols = (a_names, a_cols, a_dtypes)
            names += self._attr_cols[0]
            cols += self._attr_cols[1]
            dtypes += self._attr_cols[2]
        return names, cols, dtypes

    def egonet_feature_columns(self) -> Tuple[List[str], list, List[np.dtype]]:
        """Device columns of networkx.py:71-83."""
        K = self._K()
        host, out, _ = self._device_graph()
        rb, re = self._row_range()
        rowsum = self.__dict__.pop('_out_rowsum', None) if host.weighted else None
        if rowsum is None and host.weighted:
            rowsum = K.row_sums(out, False)
        internal, external = K.egonet_features(out, host.directed, rowsum, rb, re,
                                               shard=getattr(self, '_shard_plan', None))
        internal, external = self._finish_columns([internal, external])
        dt = np.dtype('int64') if host.integral else np.dtype('float64')
        return ['internal_edges', 'external_edges'], [internal, external], [dt, dt]

    def neighborhood_feature_columns(self):
        n1, c1, d1 = self.local_feature_columns()
        n2, c2, d2 = self.egonet_feature_columns()
        return n1 + n2, c1 + c2, d1 + d2

    def int32_exact_flags(self, names, dtypes) -> List[bool]:
        """Per generation-0 column: every value an exact integer in [0, 2^31)?  True for the degree / ego-net columns
        of an unweighted graph with fewer than 2^31 adjacency entries (each is bounded by nnz: a degree, a number of
        edges, a sum of degrees over distinct nodes) and for integer attributes inside that range -- the condition
        under which a generation may gather int32 rows (grx_aggregate_i32)."""
        host = self._device_graph()[0]
        structural = bool(host.integral and not host.weighted and host.nnz < 2 ** 31)
        attr_ok = {}
        cached = getattr(self, '_attr_cols', None)
        if cached is not None:
            arrays = self._attribute_arrays() or {}
            for name, dt in zip(cached[0], cached[2]):
                values = arrays.get(name)
                attr_ok[name] = bool(np.dtype(dt).kind in 'iu' and values is not None and len(values) and
                                     int(np.min(values)) >= 0 and int(np.max(values)) < 2 ** 31)
        prefix = self.attribute_feature_prefix + '_'
        return [attr_ok.get(nm, False) if nm.startswith(prefix) else (structural and np.dtype(dt).kind in 'iu')
                for nm, dt in zip(names, dtypes)]

    def _frame(self, names, cols, dtypes) -> pd.DataFrame:
        K = self._K()
        host = self._device_graph()[0]
        data = {nm: host.to_label_order(K.to_host(c)).astype(dt) for nm, c, dt in zip(names, cols, dtypes)}
        csr = self.to_csr()
        index = csr.label_index() if hasattr(csr, 'label_index') else pd.Index(host.labels)
        return pd.DataFrame(data, index=index, columns=names)

    # -- reference API on top ------------------------------------------------------------
    def _get_local_features(self) -> pd.DataFrame:
        return self._frame(*self.local_feature_columns())

    def _get_egonet_features(self) -> pd.DataFrame:
        return self._frame(*self.egonet_feature_columns())

    def _attribute_frame(self) -> Optional[pd.DataFrame]:
        """Numeric node attributes as a frame indexed by node label (columns already prefixed)."""
        return None

    def _attribute_arrays(self):
        """Optional fast path: {prefixed name: array in to_csr() row order} without pandas."""
        return None
