"""
Thin Python wrappers over the libgrx.so C ABI (include/grx.h) for torch-held device memory.

torch is plumbing only: tensors provide HBM allocations (caching allocator) and the current
HIP stream; every computation below is a hand-written HIP kernel in graphrole_amd/csrc.
There is no CPU path in this module -- it raises when no GPU / no libgrx.so is present.
"""
from __future__ import annotations

import ctypes
from typing import Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib

c_void_p = ctypes.c_void_p
HUB_FACTOR = 32          # GRX_HUB_FACTOR in csrc/grx_common.h


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else c_void_p(t.data_ptr())


def _stream():
    return c_void_p(torch.cuda.current_stream().cuda_stream)


def _hptr(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(c_void_p)


def device() -> torch.device:
    if not torch.cuda.is_available():
        raise _lib.GrxError('graphrole_amd needs a HIP device (torch.cuda.is_available() is False); '
                            'there is no CPU fallback')
    return torch.device('cuda', torch.cuda.current_device())


_NUM_CUS = None


def num_cus() -> int:
    global _NUM_CUS
    if _NUM_CUS is None:
        cu = ctypes.c_int(0)
        _lib.call('grx_device_info', ctypes.byref(cu), None, None, 0)
        _NUM_CUS = int(cu.value) or 256
    return _NUM_CUS


_BULK_BYTES = 1 << 20          # above this, host <-> device copies go through the pipelined staging ring


def to_device(a: np.ndarray) -> torch.Tensor:
    a = np.ascontiguousarray(a)
    if a.nbytes < _BULK_BYTES or a.dtype.kind not in 'fiu':
        return torch.from_numpy(a).to(device())
    t = torch.empty(a.shape, dtype=torch.from_numpy(a[:0]).dtype, device=device())
    _lib.call('grx_upload', _ptr(t), _hptr(a), a.nbytes, _stream())
    return t


def upload_into(dst: torch.Tensor, a: np.ndarray) -> None:
    """Host array -> an existing contiguous device tensor of the same byte size (pipelined pinned staging)."""
    a = np.ascontiguousarray(a)
    assert dst.is_contiguous() and dst.numel() * dst.element_size() == a.nbytes
    _lib.call('grx_upload', _ptr(dst), _hptr(a), a.nbytes, _stream())


def edges_to_device(a: np.ndarray) -> torch.Tensor:
    """Edge endpoint array -> int32 device tensor; int64 input (numpy's default) is narrowed while staging."""
    a = np.ascontiguousarray(a)
    if a.dtype == np.int64 and a.nbytes >= _BULK_BYTES:
        t = torch.empty(a.shape, dtype=torch.int32, device=device())
        _lib.call('grx_upload_i64_as_i32', _ptr(t), _hptr(a), a.size, _stream())
        return t
    return to_device(np.ascontiguousarray(a, dtype=np.int32))


def host_checksums(base_ptr: int, ncols: int, col_bytes: int, stride_bytes: int) -> np.ndarray:
    """64-bit content hash of ncols host columns (grx_host_checksums)."""
    out = np.empty(ncols, dtype=np.uint64)
    _lib.call('grx_host_checksums', c_void_p(base_ptr), int(ncols), int(col_bytes), int(stride_bytes), _hptr(out))
    return out


def min_value(X: torch.Tensor, n: int) -> float:
    """min over the n valid rows of a feature-major device matrix [F, ld]; NaN if any entry is NaN."""
    ws_bytes = _lib.load().grx_min_value_workspace_bytes()
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=device())
    out = torch.empty(1, dtype=torch.float64, device=device())
    _lib.call('grx_min_value', n, X.shape[0], _ptr(X), _ld(X), _ptr(out), _ptr(ws), ws_bytes, _stream())
    return float(to_host(out)[0])


import threading

_PINNED = threading.local()          # per-thread staging buffers: distinct extractor instances may run in distinct threads
_PINNED_MAX_BYTES = 1 << 20


def to_host(t: torch.Tensor) -> np.ndarray:
    """Device -> host copy of a result.  Small tensors (everything the drivers read back between
    launches: distance matrices, Gram matrices, residuals) go through a pinned staging buffer with
    an asynchronous copy + stream synchronise: in isolation 23 us behind a queued kernel against 110 us
    for Tensor.cpu() (synchronous hipMemcpy); inside the pipeline, where the stream holds milliseconds
    of queued work at every read-back, the step time did not change measurably."""
    t = t.detach()
    nbytes = t.numel() * t.element_size()
    if t.is_cuda and nbytes > _PINNED_MAX_BYTES and t.is_contiguous():
        # bulk results (the feature table, the node-role factor): chunks through the pinned staging ring, host memcpy
        # threaded (grx_download) -- a plain Tensor.cpu() into pageable memory runs at a fraction of the link rate
        out = np.empty(tuple(t.shape), dtype=torch.empty(0, dtype=t.dtype).numpy().dtype)
        _lib.call('grx_download', _hptr(out), _ptr(t), nbytes, _stream())
        return out
    if not t.is_cuda or nbytes == 0 or nbytes > _PINNED_MAX_BYTES:
        return t.cpu().numpy()
    bufs = getattr(_PINNED, 'bufs', None)
    if bufs is None:
        bufs = _PINNED.bufs = {}
    buf = bufs.get(t.dtype)
    if buf is None or buf.numel() < t.numel():
        buf = torch.empty(max(t.numel(), 4096), dtype=t.dtype, pin_memory=True)
        bufs[t.dtype] = buf
    view = buf[:t.numel()]
    view.copy_(t.contiguous().view(-1), non_blocking=True)
    torch.cuda.current_stream().synchronize()
    return view.numpy().reshape(tuple(t.shape)).copy()


def synchronize() -> None:
    """Wait for the current HIP stream (phase timing of the API calls)."""
    torch.cuda.current_stream().synchronize()


def empty(shape, dtype=torch.float64) -> torch.Tensor:
    return torch.empty(shape, dtype=dtype, device=device())


def zeros(shape, dtype=torch.float64) -> torch.Tensor:
    """Zero-filled device tensor without a torch kernel: allocation + hipMemsetAsync on the current stream (torch
    is plumbing here -- no torch compute op runs on the product path, tests/test_gpu_no_torch_ops.py)."""
    t = torch.empty(shape, dtype=dtype, device=device())
    if t.numel():
        _lib.call('grx_memset', _ptr(t), 0, t.numel() * t.element_size(), _stream())
    return t


NATIVE_SHARDING = True      # grx_refex_run / grx_nmf_fit take a communicator: a ShardPlan runs below the ABI


def _shard_args(shard):
    """(grx_comm handle, host bounds pointer) of a ShardPlan, (None, None) on one GPU."""
    if shard is None:
        return None, None
    comm = shard.comm()
    if comm is None:
        return None, None
    return comm, shard.bounds_ptr()


def _ld(t: torch.Tensor) -> int:
    """Leading dimension of a [rows, n] block; a single-row tensor may carry any stride(0)."""
    return t.stride(0) if t.shape[0] > 1 else max(t.stride(0), t.shape[1], 1)


def ptr_array(tensors: Sequence[torch.Tensor]):
    """Host table of device pointers (the `const T* const* h_*` arguments of the ABI)."""
    return (c_void_p * max(len(tensors), 1))(*[t.data_ptr() for t in tensors])


class AggregatePlan:
    """Owner of a grx_aggregate_plan (per-graph preprocessing of grx_aggregate, see grx.h)."""

    def __init__(self, row_ptr: np.ndarray) -> None:
        import ctypes
        row_ptr = np.ascontiguousarray(row_ptr, dtype=np.int64)
        handle = ctypes.c_void_p()
        _lib.call('grx_aggregate_plan_create', len(row_ptr) - 1, row_ptr.ctypes.data_as(c_void_p),
                  ctypes.byref(handle))
        self.handle = handle
        n_long, n_blocks, lanes = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int()
        _lib.call('grx_aggregate_plan_info', handle, ctypes.byref(n_long), ctypes.byref(n_blocks), ctypes.byref(lanes))
        self.n_long_rows, self.n_blocks, self.lanes_per_row = n_long.value, n_blocks.value, lanes.value

    def set_lanes(self, lanes_per_row: int) -> None:
        _lib.call('grx_aggregate_plan_set_lanes', self.handle, int(lanes_per_row))
        self.lanes_per_row = int(lanes_per_row)

    def __del__(self):
        try:
            if self.handle:
                _lib.load().grx_aggregate_plan_destroy(self.handle)
                self.handle = None
        except Exception:
            pass


class DeviceCSR:
    """CSR adjacency resident in HBM (row_ptr int64[n+1], col int32[nnz] ascending inside a row,
    optional w fp64[nnz]).  agg_col (optional): the same rows with the neighbours in the order the
    reference visits them -- the summation order of grx_aggregate (defaults to col)."""

    def __init__(self, row_ptr: np.ndarray, col: np.ndarray, w: Optional[np.ndarray] = None,
                 agg_col: Optional[np.ndarray] = None):
        dev = device()
        self.n = int(len(row_ptr) - 1)
        self.nnz = int(row_ptr[-1]) if len(row_ptr) else 0
        self.row_ptr = torch.from_numpy(np.ascontiguousarray(row_ptr, dtype=np.int64)).to(dev)
        col = np.ascontiguousarray(col, dtype=np.int32)
        self.col = torch.from_numpy(col if len(col) else np.zeros(1, np.int32)).to(dev)
        self.w = None
        if w is not None:
            w = np.ascontiguousarray(w, dtype=np.float64)
            self.w = torch.from_numpy(w if len(w) else np.zeros(1)).to(dev)
        avg = self.nnz / max(self.n, 1)
        # lane-group width of the ego-net finish kernel; rows above 32 * lanes_per_row neighbours
        # get a workgroup each there (grx_egonet_unweighted hub list)
        self.lanes_per_row = 4 if avg < 12 else 8 if avg < 24 else 16 if avg < 48 else 32
        deg = np.diff(np.asarray(row_ptr, dtype=np.int64))
        hubs = np.nonzero(deg > HUB_FACTOR * self.lanes_per_row)[0].astype(np.int32)
        self.n_hubs = int(len(hubs))
        self.hub_rows = torch.from_numpy(hubs).to(dev) if self.n_hubs else None
        self._host = (np.asarray(row_ptr, dtype=np.int64), col)
        self._oriented = None
        self.agg_col = self.col
        if agg_col is not None and len(agg_col):
            self.agg_col = torch.from_numpy(np.ascontiguousarray(agg_col, dtype=np.int32)).to(dev)
        self._plan = None
        self.degree_sorted = bool(self.n < 2 or np.all(deg[1:] <= deg[:-1]))

    @classmethod
    def from_device(cls, row_ptr: torch.Tensor, col: torch.Tensor, w: Optional[torch.Tensor],
                    agg_col: Optional[torch.Tensor], host_row_ptr: np.ndarray) -> 'DeviceCSR':
        """A CSR that is already in HBM (device_ingest); only the row pointers exist on the host."""
        self = cls.__new__(cls)
        host_row_ptr = np.ascontiguousarray(host_row_ptr, dtype=np.int64)
        self.n = int(len(host_row_ptr) - 1)
        self.nnz = int(host_row_ptr[-1])
        self.row_ptr, self.col, self.w = row_ptr, col, w
        avg = self.nnz / max(self.n, 1)
        self.lanes_per_row = 4 if avg < 12 else 8 if avg < 24 else 16 if avg < 48 else 32
        deg = np.diff(host_row_ptr)
        hubs = np.nonzero(deg > HUB_FACTOR * self.lanes_per_row)[0].astype(np.int32)
        self.n_hubs = int(len(hubs))
        self.hub_rows = torch.from_numpy(hubs).to(device()) if self.n_hubs else None
        self._host = (host_row_ptr, None)
        self._oriented = None
        self.agg_col = col if agg_col is None else agg_col
        self._plan = None
        self.degree_sorted = bool(self.n < 2 or np.all(deg[1:] <= deg[:-1]))
        return self

    def plan(self) -> AggregatePlan:
        if self._plan is None:
            self._plan = AggregatePlan(self._host[0])
        return self._plan

    def _oriented_on_device(self) -> 'DeviceCSR':
        """grx_orient_count / grx_orient_fill: the orientation without a host copy of the columns."""
        n = self.n
        lib = _lib.load()
        ws_bytes = lib.grx_orient_workspace_bytes(n)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=device())
        o_ptr = torch.empty(n + 1, dtype=torch.int64, device=device())
        _lib.call('grx_orient_count', n, _ptr(self.row_ptr), _ptr(self.col), _ptr(o_ptr), _ptr(ws), ws_bytes, _stream())
        h_ptr = o_ptr.cpu().numpy()
        o_nnz = int(h_ptr[-1])
        o_col = torch.empty(max(o_nnz, 1), dtype=torch.int32, device=device())
        arc = torch.empty(max(o_nnz, 1), dtype=torch.int64, device=device())
        _lib.call('grx_orient_fill', n, _ptr(self.row_ptr), _ptr(self.col), _ptr(o_ptr), o_nnz, _ptr(o_col), _ptr(arc),
                  _ptr(ws), ws_bytes, _stream())
        o = DeviceCSR.from_device(o_ptr, o_col, None, None, h_ptr)
        o.arc = arc
        return o

    def oriented(self) -> 'DeviceCSR':
        """Degree-oriented copy (arc u->v iff (d'(u),u) < (d'(v),v)) for grx_triangle_counts."""
        if self._oriented is None and self._host[1] is None:
            self._oriented = self._oriented_on_device()
        if self._oriented is None:
            row_ptr, col = self._host
            n = self.n
            rows = np.repeat(np.arange(n, dtype=np.int64), np.diff(row_ptr))
            colv = col.astype(np.int64)
            loops = np.bincount(rows[rows == colv], minlength=n)
            dprime = np.diff(row_ptr) - loops
            keep = (dprime[rows] < dprime[colv]) | ((dprime[rows] == dprime[colv]) & (rows < colv))
            o_ptr = np.zeros(n + 1, dtype=np.int64)
            np.cumsum(np.bincount(rows[keep], minlength=n), out=o_ptr[1:])
            self._oriented = DeviceCSR(o_ptr, col[keep])
            # per oriented arc k = u->v: begin of N+(v) | d+(v) << 32 | d+(u) << 42 | (k - begin of N+(u)) << 52, the
            # 10-bit fields saturating at 1023 (see grx_triangle_counts)
            tgt = col[keep].astype(np.int64)
            src = rows[keep]
            k = np.arange(len(tgt), dtype=np.int64)
            sat = lambda x: np.minimum(x, 1023)
            arc = (o_ptr[tgt] | (sat(o_ptr[tgt + 1] - o_ptr[tgt]) << 32) | (sat(o_ptr[src + 1] - o_ptr[src]) << 42)
                   | (sat(k - o_ptr[src]) << 52))
            self._oriented.arc = torch.from_numpy(arc if len(arc) else np.zeros(1, np.int64)).to(device())
        return self._oriented

    def triangle_split(self, rank: int, world: int) -> Tuple[int, int]:
        """Source-row range of the oriented CSR that `rank` of `world` counts triangles for:
        cuts balance sum(d+ * (d+ + 1)), the size of the list intersections a source row starts."""
        cache = self.__dict__.setdefault('_tri_split', {})
        if (rank, world) in cache:                              # a property of the graph: computed once
            return cache[(rank, world)]
        o = self.oriented()
        dplus = np.diff(o._host[0])
        work = np.cumsum(dplus * (dplus + 1) + 1)
        total = int(work[-1]) if len(work) else 0
        cuts = [0] + [int(np.searchsorted(work, total * p / world, side='left')) for p in range(1, world)] + [self.n]
        cuts = np.maximum.accumulate(np.array(cuts, dtype=np.int64))
        cache[(rank, world)] = (int(cuts[rank]), int(cuts[rank + 1]))
        return cache[(rank, world)]


def device_ingest(n: int, src: np.ndarray, dst: np.ndarray, w: Optional[np.ndarray], directed: bool, nnz: int):
    """
    grx_ingest: edge arrays -> (perm, inv, host row pointers [, host transposed row pointers], DeviceCSR of the
    out-adjacency, DeviceCSR of the in-adjacency or None) with rows in the internal (degree-descending) order --
    what InternalGraph + DeviceCSR build on the host, without the host.
    """
    dev = device()
    m = int(len(src))
    d_src = edges_to_device(src)
    d_dst = edges_to_device(dst)
    d_w = None if w is None else to_device(np.ascontiguousarray(w, dtype=np.float64))
    perm = torch.empty(n, dtype=torch.int32, device=dev)
    inv = torch.empty(n, dtype=torch.int32, device=dev)
    row_ptr = torch.empty(n + 1, dtype=torch.int64, device=dev)
    col = torch.empty(max(nnz, 1), dtype=torch.int32, device=dev)
    # zero-filled: with validate=False a duplicate edge writes one slot only -- the others must not be garbage
    wcol = None if w is None else zeros(max(nnz, 1))
    agg_col = torch.empty(max(nnz, 1), dtype=torch.int32, device=dev)
    t_row_ptr = t_col = t_w = None
    if directed:
        t_row_ptr = torch.empty(n + 1, dtype=torch.int64, device=dev)
        t_col = torch.empty(max(m, 1), dtype=torch.int32, device=dev)
        t_w = None if w is None else zeros(max(m, 1))
    ws_bytes = _lib.load().grx_ingest_workspace_bytes(n, m, int(directed))
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    _lib.call('grx_ingest', n, m, _ptr(d_src), _ptr(d_dst), _ptr(d_w), int(directed), int(nnz), _ptr(perm), _ptr(inv),
              _ptr(row_ptr), _ptr(col), _ptr(wcol), _ptr(agg_col), _ptr(t_row_ptr), _ptr(t_col), _ptr(t_w), _ptr(ws),
              ws_bytes, _stream())
    del ws
    h_row_ptr = to_host(row_ptr)
    out = DeviceCSR.from_device(row_ptr, col, wcol, agg_col, h_row_ptr)
    tr = None
    if directed:
        tr = DeviceCSR.from_device(t_row_ptr, t_col, t_w, None, to_host(t_row_ptr))
    # the internal order stays on the device (int32); the host copies are fetched only if something asks
    return perm, inv, h_row_ptr, out, tr


def row_sums(csr: DeviceCSR, add_self_loop: bool, row_begin: int = 0, row_end: Optional[int] = None,
             out: Optional[torch.Tensor] = None) -> torch.Tensor:
    row_end = csr.n if row_end is None else row_end
    if out is None:
        out = zeros(csr.n, dtype=torch.float64)
    _lib.call('grx_row_sums', csr.n, _ptr(csr.row_ptr), _ptr(csr.col), _ptr(csr.w), int(add_self_loop),
              row_begin, row_end, _ptr(out), _stream())
    return out


def add_columns(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    out = torch.empty_like(a)
    _lib.call('grx_add_columns', a.numel(), _ptr(a), _ptr(b), _ptr(out), _stream())
    return out


def _egonet_call(csr: 'DeviceCSR', directed: bool, rowsum, row_begin: int, row_end: int, internal, external) -> None:
    nnz = int(csr.nnz)
    ws_bytes = int(_lib.load().grx_egonet_workspace_bytes(csr.n, nnz))
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=device())
    _lib.call('grx_egonet_features', csr.n, nnz, _ptr(csr.row_ptr), _ptr(csr.col), _ptr(csr.w), _ptr(rowsum),
              int(directed), row_begin, row_end, _ptr(internal), _ptr(external), _ptr(ws), ws_bytes, _stream())


def egonet_features(csr: DeviceCSR, directed: bool, rowsum: Optional[torch.Tensor] = None,
                    row_begin: int = 0, row_end: Optional[int] = None, shard=None) -> Tuple[torch.Tensor, torch.Tensor]:
    """
    internal / external ego-net edge weight of rows [row_begin,row_end).
    shard: a ShardPlan (rank, world, all_reduce_sum_) -- the triangle counts of the unweighted
    undirected path are then split over the ranks by oriented source row and summed once.
    """
    row_end = csr.n if row_end is None else row_end
    if csr.w is None and not directed:
        # exact integer fast path through per-node triangle counts
        if shard is None:
            T = triangle_counts(csr)
        else:
            T = triangle_counts(csr, *csr.triangle_split(shard.rank, shard.world))
            shard.all_reduce_sum_(T)
        return egonet_from_triangles(csr, T, row_begin, row_end)
    internal = zeros(csr.n, dtype=torch.float64)
    external = zeros(csr.n, dtype=torch.float64)
    if csr.w is not None and rowsum is None:
        rowsum = row_sums(csr, False)
    _egonet_call(csr, directed, rowsum, row_begin, row_end, internal, external)
    return internal, external


def triangle_counts(csr: DeviceCSR, row_begin: int = 0, row_end: Optional[int] = None) -> torch.Tensor:
    """int64 [n] triangle counts through every node, accumulated over oriented source rows
    [row_begin,row_end) (all-reduce SUM across ranks when the rows are split)."""
    row_end = csr.n if row_end is None else row_end
    o = csr.oriented()
    T = zeros(max(csr.n, 1), dtype=torch.int64)
    _lib.call('grx_triangle_counts', csr.n, _ptr(o.row_ptr), _ptr(o.col), _ptr(o.arc), row_begin, row_end, _ptr(T),
              _stream())
    return T


def egonet_from_triangles(csr: DeviceCSR, T: torch.Tensor, row_begin: int = 0,
                          row_end: Optional[int] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    row_end = csr.n if row_end is None else row_end
    # a partial row range leaves the other rows zero (callers complete them by an exchange, tests read them)
    alloc = (lambda k: torch.empty(k, dtype=torch.float64, device=device())) if (row_begin == 0 and row_end == csr.n) else zeros
    internal, external = alloc(csr.n), alloc(csr.n)
    scratch = torch.empty(max(csr.n, 1), dtype=torch.int32, device=device())
    _lib.call('grx_egonet_unweighted', csr.n, _ptr(csr.row_ptr), _ptr(csr.col), _ptr(T), row_begin, row_end,
              _ptr(internal), _ptr(external), _ptr(scratch), _ptr(csr.hub_rows), csr.n_hubs,
              HUB_FACTOR * csr.lanes_per_row, _stream())
    return internal, external


def egonet_features_general(csr: DeviceCSR, directed: bool, rowsum: Optional[torch.Tensor] = None,
                            row_begin: int = 0, row_end: Optional[int] = None):
    """The gather kernel for any graph (weighted / directed); also valid for unweighted undirected."""
    row_end = csr.n if row_end is None else row_end
    internal = zeros(csr.n, dtype=torch.float64)
    external = zeros(csr.n, dtype=torch.float64)
    if csr.w is not None and rowsum is None:
        rowsum = row_sums(csr, False)
    _egonet_call(csr, directed, rowsum, row_begin, row_end, internal, external)
    return internal, external


def pack_rows(cols: Sequence[torch.Tensor], n: int) -> Tuple[torch.Tensor, int]:
    """Column tensors -> row-major [n, ldr] gather source, ldr = grx_aggregate_ldr(f) (cache-line
    friendly row size: 16 / 32 / 64 bytes or whole 128-byte lines)."""
    f = len(cols)
    ldr = _lib.load().grx_aggregate_ldr(f)
    rows = torch.empty((max(n, 1), ldr), dtype=torch.float64, device=device())
    if f and n:
        ptrs = ptr_array(cols)
        _lib.call('grx_pack_rows', n, f, ptrs, _ptr(rows), ldr, _stream())
    return rows, ldr


def aggregate_i32_ok(csr: DeviceCSR, f: int) -> bool:
    """Can f exact-int32 columns be aggregated through the integer kernel on this graph (grx_aggregate_i32_ok)?"""
    return bool(_lib.load().grx_aggregate_i32_ok(csr.plan().handle, int(f)))


def pack_rows_i32(cols: Sequence[torch.Tensor], n: int) -> Tuple[torch.Tensor, int]:
    """fp64 columns holding exact integers in [0, 2^31) -> row-major int32 [n, ldi] gather source (16- / 32-byte rows)."""
    f = len(cols)
    ldi = _lib.load().grx_aggregate_ldi(f)
    rows = torch.empty((max(n, 1), ldi), dtype=torch.int32, device=device())
    if n:
        _lib.call('grx_pack_rows_i32', n, f, ptr_array(cols), _ptr(rows), ldi, _stream())
    return rows, ldi


def aggregate_i32(csr: DeviceCSR, rows: torch.Tensor, f: int, ldi: int, row_begin: int = 0, row_end: Optional[int] = None,
                  want_sum: bool = True, want_mean: bool = True) -> torch.Tensor:
    """grx_aggregate_i32: the [2f, n] sums / means block of aggregate() from an int32 gather source."""
    n = csr.n
    row_end = n if row_end is None else row_end
    out = torch.empty((2 * f, n), dtype=torch.float64, device=device())
    s_ptr = c_void_p(out.data_ptr()) if want_sum else None
    m_ptr = c_void_p(out.data_ptr() + f * n * 8) if want_mean else None
    _lib.call('grx_aggregate_i32', csr.plan().handle, _ptr(csr.row_ptr), _ptr(csr.agg_col), f, _ptr(rows), ldi,
              row_begin, row_end, s_ptr, m_ptr, n, _stream())
    return out


def aggregate(csr: DeviceCSR, rows: torch.Tensor, f: int, ldr: int, row_begin: int = 0,
              row_end: Optional[int] = None, want_sum: bool = True, want_mean: bool = True,
              out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """
    Returns a [2f, n] block: rows 0..f-1 = neighbour sums, rows f..2f-1 = neighbour means
    (features/extract.py:152-162 orders all sums before all means).  Bit-exact with the
    reference's Series.sum() when csr.agg_col lists the neighbours in adjacency order.
    """
    n = csr.n
    row_end = n if row_end is None else row_end
    if out is None:
        full = row_begin == 0 and row_end == n and want_sum and want_mean
        out = torch.empty((2 * f, n), dtype=torch.float64, device=device()) if full else zeros((2 * f, n))
    if f == 0:
        return out
    s_ptr = c_void_p(out.data_ptr()) if want_sum else None
    m_ptr = c_void_p(out.data_ptr() + f * n * 8) if want_mean else None
    _lib.call('grx_aggregate', csr.plan().handle, _ptr(csr.row_ptr), _ptr(csr.agg_col), f, _ptr(rows), ldr,
              row_begin, row_end, s_ptr, m_ptr, n, _stream())
    return out


def aggregate_var(csr: DeviceCSR, rows: torch.Tensor, f: int, ldr: int, mean: torch.Tensor, row_begin: int = 0,
                  row_end: Optional[int] = None, want_var: bool = True, want_std: bool = True) -> torch.Tensor:
    """[2f, n] block: rows 0..f-1 = sample variance (ddof = 1) of the neighbours' values, rows f..2f-1 =
    its square root; `mean` = the [f, n] neighbour means of the same rows (aggregate()[f:])."""
    n = csr.n
    row_end = n if row_end is None else row_end
    out = zeros((2 * f, n), dtype=torch.float64)
    if f == 0:
        return out
    assert mean.shape == (f, n) and mean.is_contiguous()
    v_ptr = c_void_p(out.data_ptr()) if want_var else None
    s_ptr = c_void_p(out.data_ptr() + f * n * 8) if want_std else None
    _lib.call('grx_aggregate_var', csr.plan().handle, _ptr(csr.row_ptr), _ptr(csr.agg_col), f, _ptr(rows), ldr,
              row_begin, row_end, _ptr(mean), v_ptr, s_ptr, n, _stream())
    return out


def aggregate_prod(csr: DeviceCSR, rows: torch.Tensor, f: int, ldr: int, row_begin: int = 0,
                   row_end: Optional[int] = None) -> torch.Tensor:
    """[f, n] block: left-to-right product of the neighbours' values (agg 'prod'); only rows
    [row_begin, row_end) are written."""
    n = csr.n
    row_end = n if row_end is None else row_end
    out = torch.empty((f, n), dtype=torch.float64, device=device())
    if f:
        _lib.call('grx_aggregate_prod', _ptr(csr.row_ptr), _ptr(csr.agg_col), f, _ptr(rows), ldr, row_begin, row_end,
                  _ptr(out), n, _stream())
    return out


def aggregate_minmax(csr: DeviceCSR, rows: torch.Tensor, f: int, ldr: int, row_begin: int = 0,
                     row_end: Optional[int] = None, want_min: bool = True, want_max: bool = True) -> torch.Tensor:
    """[2f, n] block: rows 0..f-1 = neighbour minima, rows f..2f-1 = neighbour maxima."""
    n = csr.n
    row_end = n if row_end is None else row_end
    out = zeros((2 * f, n), dtype=torch.float64)
    if f == 0:
        return out
    lo = c_void_p(out.data_ptr()) if want_min else None
    hi = c_void_p(out.data_ptr() + f * n * 8) if want_max else None
    _lib.call('grx_aggregate_minmax', csr.plan().handle, _ptr(csr.row_ptr), _ptr(csr.agg_col), f, _ptr(rows), ldr,
              row_begin, row_end, lo, hi, n, _stream())
    return out


# ---- integer (int64-bits) columns, median, count: csrc/grx_aggx.hip -------------------------------------------------
def packed_layout(field_bits: Sequence[int], degree_bits: int, out_field: Sequence[int], out_is_mean: Sequence[bool]):
    """grx_packed_layout + the row width it needs (8 / 16 bytes; 0: does not fit)."""
    L = _lib.PackedLayout()
    L.n_fields, L.degree_bits, L.n_out = len(field_bits), int(degree_bits), len(out_field)
    for k, b in enumerate(field_bits):
        L.field_bits[k] = int(b)
    for j, (k, m) in enumerate(zip(out_field, out_is_mean)):
        L.out_field[j], L.out_is_mean[j] = int(k), int(bool(m))
    return L, int(_lib.load().grx_packed_row_bytes(ctypes.byref(L)))


def column_bits(block: torch.Tensor, n: int, int_mask: int, row_begin: int = 0, row_end: Optional[int] = None) -> torch.Tensor:
    """int32[ncols]: bits of the maximum of every flagged column of a [ncols, >= n] fp64 block (grx_column_bits)."""
    ncols = block.shape[0]
    out = zeros(max(ncols, 1), dtype=torch.int32)
    _lib.call('grx_column_bits', n, ncols, _ptr(block), _ld(block), row_begin, n if row_end is None else row_end,
              int(int_mask), _ptr(out), _stream())
    return out[:ncols]


def pack_fields(csr: DeviceCSR, layout, row_bytes: int, field_cols: Sequence[torch.Tensor]) -> torch.Tensor:
    """Bit-packed gather source of grx_aggregate_packed: uint8 [n * row_bytes]."""
    rows = torch.empty(max(csr.n, 1) * row_bytes, dtype=torch.uint8, device=device())
    _lib.call('grx_pack_fields', csr.n, ctypes.byref(layout), ptr_array(list(field_cols)), _ptr(csr.row_ptr), _ptr(rows), _stream())
    return rows


def aggregate_packed(csr: DeviceCSR, layout, rows: torch.Tensor, row_begin: int = 0, row_end: Optional[int] = None,
                     want_sum: bool = True, want_mean: bool = True) -> torch.Tensor:
    """[2 * n_out, n] (sums, then means) from bit-packed integer rows: bit-identical to aggregate() on the fp64 columns."""
    n, f = csr.n, layout.n_out
    out = torch.zeros((2 * f, max(n, 1)), dtype=torch.float64, device=device())
    _lib.call('grx_aggregate_packed', csr.plan().handle, _ptr(csr.row_ptr), _ptr(csr.agg_col), ctypes.byref(layout), _ptr(rows),
              row_begin, n if row_end is None else row_end, _ptr(out[:f]) if want_sum else None,
              _ptr(out[f:]) if want_mean else None, out.stride(0), _stream())
    return out[:, :n]


def convert_i64_to_f64(col: torch.Tensor) -> torch.Tensor:
    """A column of int64 BITS (stored in an fp64 tensor) -> the fp64 values (numpy astype(float64))."""
    out = torch.empty_like(col)
    _lib.call('grx_convert_i64_to_f64', col.numel(), _ptr(col), _ptr(out), _stream())
    return out


def convert_f64_to_i64(col: torch.Tensor) -> torch.Tensor:
    """fp64 values (exact integers) -> int64 bits in an fp64 tensor."""
    out = torch.empty_like(col)
    _lib.call('grx_convert_f64_to_i64', col.numel(), _ptr(col), _ptr(out), _stream())
    return out


def aggregate_i64(csr: DeviceCSR, rows: torch.Tensor, f: int, ldr: int, row_begin: int = 0, row_end: Optional[int] = None,
                  want: Sequence[str] = ('sum', 'prod', 'min', 'max')) -> dict:
    """Wrapping int64 sum / prod and min / max over the neighbours; `rows` = pack_rows() of int64-bits columns.
    Returns {agg: [f, n] tensor of int64 bits}."""
    n = csr.n
    row_end = n if row_end is None else row_end
    outs = {a: torch.empty((f, n), dtype=torch.float64, device=device()) for a in want}
    if f:
        _lib.call('grx_aggregate_i64', _ptr(csr.row_ptr), _ptr(csr.agg_col), f, _ptr(rows), ldr, row_begin, row_end,
                  _ptr(outs.get('sum')), _ptr(outs.get('prod')), _ptr(outs.get('min')), _ptr(outs.get('max')), n, _stream())
    return outs


def aggregate_count(csr: DeviceCSR, f: int, row_begin: int = 0, row_end: Optional[int] = None,
                    as_i64: bool = False) -> torch.Tensor:
    """[f, n] block: the number of neighbours of every row, as fp64 values or int64 bits (aggs 'count' / 'size')."""
    n = csr.n
    row_end = n if row_end is None else row_end
    out = torch.empty((f, n), dtype=torch.float64, device=device())
    if f:
        _lib.call('grx_aggregate_count', _ptr(csr.row_ptr), f, row_begin, row_end, int(as_i64), _ptr(out), n, _stream())
    return out


def aggregate_median(csr: DeviceCSR, rows: torch.Tensor, f: int, ldr: int, row_begin: int = 0,
                     row_end: Optional[int] = None) -> torch.Tensor:
    """[f, n] block: numpy's median of the neighbours' values (agg 'median'); 0 for rows without neighbours."""
    n = csr.n
    row_end = n if row_end is None else row_end
    out = torch.empty((f, n), dtype=torch.float64, device=device())
    if f and row_end > row_begin:
        ws_bytes = _lib.load().grx_aggregate_median_workspace_bytes(csr.nnz)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=device())
        _lib.call('grx_aggregate_median', _ptr(csr.row_ptr), _ptr(csr.agg_col), f, _ptr(rows), ldr, row_begin, row_end,
                  _ptr(out), n, _ptr(ws), ws_bytes, _stream())
    return out


def sort_columns(block: torch.Tensor) -> torch.Tensor:
    """Ascending sort of every row of a [ncols, n] fp64 block (each row is one feature column)."""
    ncols, n = block.shape
    out = torch.empty_like(block)
    if ncols == 0 or n == 0:
        return out
    ws_bytes = _lib.load().grx_sort_workspace_bytes(n, ncols)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=device())
    _lib.call('grx_sort_columns', n, ncols, _ptr(block), block.stride(0), _ptr(out), _ld(out), _ptr(ws),
              ws_bytes, _stream())
    return out


def vertical_log_bin(block: torch.Tensor, frac: float = 0.5, out: Optional[torch.Tensor] = None,
                     is_i64: Optional[Sequence[bool]] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """Bin every row of a [ncols, n] fp64 block -> (uint8 [ncols, n], int32 [ncols] bin counts).
    `block` (and `out`) may be row-strided views (e.g. every P-th column of a candidate block).
    is_i64[j]: row j holds int64 bits and is ordered as integers (grx_vertical_log_bin_typed)."""
    ncols, n = block.shape
    assert n <= 1 or block.stride(1) == 1
    bins = out if out is not None else zeros((ncols, max(n, 1)), dtype=torch.uint8)[:, :n]
    nbins = zeros(max(ncols, 1), dtype=torch.int32)
    lib = _lib.load()
    ws_bytes = lib.grx_log_bin_workspace_bytes(n, ncols)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=device())
    flags = None
    if is_i64 is not None and any(is_i64):
        flags = np.ascontiguousarray(np.asarray(list(is_i64), dtype=np.uint8))
    _lib.call('grx_vertical_log_bin_typed', n, ncols, _ptr(block), block.stride(0) if ncols else n, _hptr(flags),
              float(frac), _ptr(bins), bins.stride(0) if ncols else max(n, 1), _ptr(nbins), _ptr(ws), ws_bytes, _stream())
    return bins, nbins[:ncols]


def chebyshev(bin_cols: Sequence[torch.Tensor], n: int, first_new: int = 0, row_begin: int = 0,
              row_end: Optional[int] = None, cap: int = 255) -> torch.Tensor:
    """int32 [F, F] pairwise max |bin difference| over rows [row_begin,row_end).  Entries <= cap are
    exact, larger ones are some value > cap (cap = 255: all exact)."""
    F = len(bin_cols)
    row_end = n if row_end is None else row_end
    dist = zeros((F, F), dtype=torch.int32)
    if F >= 2 and row_end > row_begin:
        ptrs = ptr_array(bin_cols)
        _lib.call('grx_chebyshev', row_begin, row_end, F, first_new, ptrs, _ptr(dist), int(cap), _stream())
    return dist


# ------------------------------------------------------------------------------- whole ReFeX loop
def _refex_arena_guess(n: int, f0: int, n_aggs: int, max_gens: int) -> int:
    """First size of the grx_refex_run arena: a wrong guess costs a whole second run (the call reports what it needed
    and is repeated), an oversized one a larger hipMalloc.  Model: a generation keeps ~90 % of n_aggs x its parents
    (what the benchmark graphs do), every candidate block, its bins and the binning workspace of the widest generation
    live at once."""
    lib = _lib.load()
    growth = min(0.9 * n_aggs, 4.0)
    per_gen = [f0 * growth ** g for g in range(max(1, min(max_gens, 6)))]
    cols = sum(min(c, 400.0) for c in per_gen)
    widest = int(min(max(per_gen) + 1, 400))
    blocks = int(cols * max(n, 1) * 9.2)                               # fp64 columns + uint8 bins, 256-byte slack
    scratch = max(lib.grx_log_bin_workspace_bytes(n, lib.grx_refex_bin_batch(n, widest)), max(n, 1) * 8 * lib.grx_aggregate_ldr(widest))
    return blocks + int(scratch) + (64 << 20)



_AGG_NAMES = {v: k for k, v in _lib.AGG_IDS.items()}


def refex_run(csr: DeviceCSR, gen0_cols: Sequence[torch.Tensor], gen0_names: Sequence[str], max_generations: int,
              aggs: Sequence[str], arena=None, shard=None,
              gen0_int32: Optional[Sequence[bool]] = None):
    """
    grx_refex_run: the generation loop of RecursiveFeatureExtractor below the ABI (one call, one GPU).
    Returns (columns, generations, generation_count, arena): `columns` = one dict per RECORDED feature in
    record order {generation, parent, agg (name or None), gen0_index, work_position, col (fp64[n] tensor --
    a view into a chunk of `arena`, or the caller's generation-0 column)}; `generations` = per-generation counts.
    `arena`: a list of uint8 device tensors (chunks).  The first one is sized by a model of the run; when the run
    needs more the library asks for it through a grow callback and a chunk is appended -- a run is never repeated
    because the first size was a guess (round 5; it used to be, and at config 5 always was).  Pass the list back in to
    reuse the memory.
    shard: a ShardPlan -- the loop aggregates this rank's rows only and issues its own exchanges (RCCL or the
    plan's callback transport) between the kernels; every rank gets the complete columns.  Every allocation of the
    loop is sized from rank-independent bounds, so all ranks grow at the same points: no agreement is needed.
    """
    import time as _time
    comm, bounds = _shard_args(shard)
    n = csr.n
    f0 = len(gen0_cols)
    int_flags = None if gen0_int32 is None else (ctypes.c_int * max(f0, 1))(*[int(bool(b)) for b in gen0_int32])
    agg_ids = (ctypes.c_int * len(aggs))(*[_lib.AGG_IDS[a] for a in aggs])
    names = (ctypes.c_char_p * f0)(*[nm.encode('utf-8') for nm in gen0_names])
    col_ptrs = ptr_array(list(gen0_cols))
    max_gens = max(int(max_generations), 1)
    refex_run.trace = []                       # diagnostics: (what, seconds) of every allocation and library call
    if isinstance(arena, torch.Tensor):
        arena = [arena]
    chunks = list(arena) if arena else []
    if not chunks:
        _t0 = _time.perf_counter()
        chunks.append(torch.empty(_refex_arena_guess(n, f0, len(aggs), max_gens), dtype=torch.uint8, device=device()))
        refex_run.trace.append(('arena %d bytes (first chunk)' % chunks[0].numel(), _time.perf_counter() - _t0))
    # the library walks the chunks in order: the first is passed as the arena, the others are handed out again by the
    # grow callback before any new memory is allocated
    spare = chunks[1:]
    chunks = chunks[:1]

    def _grow(nbytes, _user):
        try:
            _t0 = _time.perf_counter()
            for k, t in enumerate(spare):
                if t.numel() >= nbytes:
                    chunk = spare.pop(k)
                    break
            else:
                chunk = torch.empty(int(nbytes), dtype=torch.uint8, device=device())
            chunks.append(chunk)
            refex_run.trace.append(('grow %d bytes' % nbytes, _time.perf_counter() - _t0))
            return chunk.data_ptr()
        except Exception:                       # out of device memory: the library reports GRX_ERR_WORKSPACE
            grow_failed.append(int(nbytes))
            return None

    grow_failed: list = []                     # a failed grow is an exhausted ARENA, whatever the byte counts below say
    grow_cb = _lib.GROW_FN(_grow)
    max_columns = 256
    refex_run.attempts = 0                     # diagnostics: how often the library was entered (column table grown)
    while True:
        refex_run.attempts += 1
        table = (_lib.RefexColumn * max_columns)()
        gens = (_lib.RefexGeneration * max_gens)()
        n_cols, gen_count, needed = ctypes.c_int(0), ctypes.c_int(0), ctypes.c_size_t(0)
        lib = _lib.load()
        _t0 = _time.perf_counter()
        rc = lib.grx_refex_run(csr.plan().handle, n, _ptr(csr.row_ptr), _ptr(csr.agg_col), f0, col_ptrs, names, int_flags,
                               int(max_generations), len(aggs), agg_ids, comm, bounds, _ptr(chunks[0]), chunks[0].numel(),
                               grow_cb, None, max_columns, table,
                               ctypes.byref(n_cols), max_gens, gens, ctypes.byref(gen_count), ctypes.byref(needed),
                               _stream())
        refex_run.trace.append(('grx_refex_run rc=%d chunks=%d needed=%d' % (rc, len(chunks), needed.value),
                                _time.perf_counter() - _t0))
        # GRX_ERR_WORKSPACE is returned for a full column table AND for an exhausted arena.  The table is the cause only
        # when no grow failed (a reused spare chunk can be larger than what the library asked for, so the byte counts alone
        # do not tell) -- and the table is not grown without bound: 256 -> 65 536 columns is far beyond any feature set
        if rc == -3 and not grow_failed and needed.value <= sum(c.numel() for c in chunks) and max_columns < 65536:
            max_columns *= 4                            # GRX_ERR_WORKSPACE from the column table, not from the arena
            spare[:0] = chunks[1:]
            chunks = chunks[:1]
            continue
        _lib.check(rc, 'grx_refex_run')
        break
    agg_names = _AGG_NAMES
    # one reinterpretation per chunk, one slice per column: tensor views cost ~2 us each on the host, and this loop
    # sits between the last kernel of the generation loop and whatever the caller launches next
    spans = [(c.data_ptr(), c.data_ptr() + c.numel(), c[:c.numel() & ~7].view(torch.float64)) for c in chunks]
    columns = []
    for i in range(n_cols.value):
        c = table[i]
        if c.gen0_index >= 0:
            col = gen0_cols[c.gen0_index]
        else:
            ptr = int(c.d_col)
            for lo, hi, view in spans:
                if lo <= ptr < hi:
                    col = view.narrow(0, (ptr - lo) >> 3, n)       # (narrow: a third of the host cost of a slice)
                    break
            else:
                raise _lib.GrxError('grx_refex_run returned a column outside every arena chunk')
        columns.append(dict(generation=c.generation, parent=c.parent, agg=agg_names.get(c.agg), gen0_index=c.gen0_index,
                            work_position=c.work_position, col=col))
    generations = [dict(generation=g, candidates=gens[g].candidates, working=gens[g].working, dropped=gens[g].dropped,
                        retained=gens[g].retained, gather_row_bytes=gens[g].gather_row_bytes)
                   for g in range(gen_count.value + 1)]
    return columns, generations, int(gen_count.value), chunks + spare


# ------------------------------------------------------------------------------- NMF
def gather_columns(cols: Sequence[torch.Tensor], n: int) -> torch.Tensor:
    F = len(cols)
    out = torch.empty((F, max(n, 1)), dtype=torch.float64, device=device())
    if F and n:
        ptrs = ptr_array(cols)
        _lib.call('grx_gather_columns', n, F, ptrs, _ptr(out), _ld(out), _stream())
    return out


def gram(X: torch.Tensor, n: int, T: Optional[np.ndarray] = None, row_begin: int = 0,
         row_end: Optional[int] = None) -> Tuple[np.ndarray, float]:
    """(X T)^T (X T) as a host k x k array plus sum(X) (valid when T is None)."""
    F = X.shape[0]
    row_end = n if row_end is None else row_end
    k = F if T is None else T.shape[1]
    if T is not None:
        T = np.ascontiguousarray(T, dtype=np.float64)
        assert T.shape[0] == F
    lib = _lib.load()
    ws_bytes = lib.grx_gram_workspace_bytes(n, k)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=device())
    out = torch.empty(k * k + 1, dtype=torch.float64, device=device())
    _lib.call('grx_gram', n, F, _ptr(X), _ld(X), row_begin, row_end, _hptr(T), k, _ptr(out), _ptr(ws),
              ws_bytes, _stream())
    host = to_host(out)
    return host[:k * k].reshape(k, k).copy(), float(host[k * k])


def project(X: torch.Tensor, n: int, Z: np.ndarray, row_begin: int = 0,
            row_end: Optional[int] = None, out: Optional[torch.Tensor] = None):
    """U = X Z (feature-major [r, ld]) and host stats [r,4] (see grx.h)."""
    F = X.shape[0]
    row_end = n if row_end is None else row_end
    Z = np.ascontiguousarray(Z, dtype=np.float64)
    r = Z.shape[1]
    if out is None:
        out = zeros((r, X.shape[1]), dtype=torch.float64)
    lib = _lib.load()
    ws_bytes = lib.grx_project_workspace_bytes(n, r)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=device())
    stats = torch.empty(r * 4, dtype=torch.float64, device=device())
    _lib.call('grx_project', n, F, _ptr(X), _ld(X), row_begin, row_end, _hptr(Z), r, _ptr(out),
              _ld(out), _ptr(stats), _ptr(ws), ws_bytes, _stream())
    return out, to_host(stats).reshape(r, 4)


def nndsvd_apply(U: torch.Tensor, n: int, sign: np.ndarray, scale: np.ndarray, eps: float, fill: float,
                 row_begin: int = 0, row_end: Optional[int] = None) -> None:
    row_end = n if row_end is None else row_end
    sign = np.ascontiguousarray(sign, dtype=np.float64)
    scale = np.ascontiguousarray(scale, dtype=np.float64)
    _lib.call('grx_nndsvd_apply', n, U.shape[0], _ptr(U), _ld(U), row_begin, row_end, _hptr(sign),
              _hptr(scale), float(eps), float(fill), _stream())


# ---- host-side small dense algebra of the NNDSVDa initialisation (grx.h: grx_host_*) ----------
def _hp(a: np.ndarray):
    return a.ctypes.data_as(c_void_p)


def host_whiten(G1: np.ndarray, rank: int = 0) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """Eigen-pairs of the Gram matrix above the numerical floor: (T1 [F, k], lam [k], V [F, k]).  rank > 0
    (grx_host_whiten_for_rank): when fewer than min(rank, F) directions survive the floor -- a graded table whose small
    columns fell under it -- the columns are equilibrated exactly before the eigen-decomposition."""
    import ctypes
    F = G1.shape[0]
    G1 = np.ascontiguousarray(G1, dtype=np.float64)
    T1, lam, V = np.empty(F * F), np.empty(F), np.empty(F * F)
    k = ctypes.c_int(0)
    if rank > 0:
        _lib.call('grx_host_whiten_for_rank', F, _hp(G1), int(rank), _hp(T1), _hp(lam), _hp(V), ctypes.byref(k))
    else:
        _lib.call('grx_host_whiten', F, _hp(G1), _hp(T1), _hp(lam), _hp(V), ctypes.byref(k))
    k = k.value
    return T1[:F * k].reshape(F, k), lam[:k], V[:F * k].reshape(F, k)


def host_range_finder(T1: np.ndarray, lam: np.ndarray, V: np.ndarray, G2: np.ndarray, omega: np.ndarray, r: int,
                      n_iter: int) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """sklearn's randomized_svd in the whitened k x F space: (Z [F, r], S [r], Vt [r, F])."""
    F, k = T1.shape
    omega = np.ascontiguousarray(omega, dtype=np.float64)
    assert omega.shape[0] == F
    Z, S, Vt = np.empty((F, r)), np.empty(r), np.empty((r, F))
    _lib.call('grx_host_range_finder', F, k, _hp(np.ascontiguousarray(T1)), _hp(np.ascontiguousarray(lam)),
              _hp(np.ascontiguousarray(V)), _hp(np.ascontiguousarray(G2, dtype=np.float64)), _hp(omega),
              omega.shape[1], int(r), int(n_iter), _hp(Z), _hp(S), _hp(Vt))
    return Z, S, Vt


def host_small_svd(X: np.ndarray, omega: np.ndarray, r: int, n_iter: int) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """n < F: sklearn's transposed randomized_svd on the small host matrix (grx_host_small_svd): (U [n, r], S [r],
    V [r, F]) before svd_flip."""
    X = np.ascontiguousarray(X, dtype=np.float64)
    omega = np.ascontiguousarray(omega, dtype=np.float64)
    n, F = X.shape
    assert omega.shape[0] == n
    U, S, V = np.empty((n, r)), np.empty(r), np.empty((r, F))
    _lib.call('grx_host_small_svd', n, F, _hp(X), _hp(omega), omega.shape[1], int(r), int(n_iter), _hp(U), _hp(S), _hp(V))
    return U, S, V


def host_nndsvd_plan(S: np.ndarray, Vt: np.ndarray, stats: np.ndarray):
    r, F = Vt.shape
    sign, scale, H = np.empty(r), np.empty(r), np.empty((r, F))
    _lib.call('grx_host_nndsvd_plan', r, F, _hp(np.ascontiguousarray(S)), _hp(np.ascontiguousarray(Vt)),
              _hp(np.ascontiguousarray(stats, dtype=np.float64)), _hp(sign), _hp(scale), _hp(H))
    return sign, scale, H


def lloyd_max(values: torch.Tensor, n_bins: int, max_iter: int = 300):
    """1-D Lloyd-Max quantiser of a flat fp64 device tensor -> (quantised tensor, centres [n_bins],
    info int32[3] = {iterations, non-empty cells, distinct output values})."""
    m = values.numel()
    out = torch.empty_like(values)
    centers = torch.empty(max(n_bins, 1), dtype=torch.float64, device=device())
    info = zeros(3, dtype=torch.int32)
    ws_bytes = _lib.load().grx_lloyd_max_workspace_bytes(m)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=device())
    _lib.call('grx_lloyd_max', m, _ptr(values), int(n_bins), int(max_iter), _ptr(out), _ptr(centers), _ptr(info),
              _ptr(ws), ws_bytes, _stream())
    return out, centers, info


# ---- the reference's quantiser: sklearn KMeans(n_clusters, random_state=1) reproduced (grx_kmeans1d) ----
_KMEANS_DRAWS: dict = {}


def kmeans_draws(m: int, n_bins: int):
    """The random numbers sklearn's KMeans(random_state=1) consumes for m samples and n_bins clusters -- they do
    not depend on the data: (index of the first seed, (n_bins - 1) x n_trials uniforms).
    _kmeans_plusplus: center_id = random_state.choice(n_samples, p=sample_weight / sample_weight.sum()), i.e.
    searchsorted(cumsum(p) / cumsum(p)[-1], random_sample(), side='right') with p = 1/m (what numpy's choice
    does, without its argument checks); then random_state.uniform(size=2 + int(log(k))) per further seed."""
    key = (int(m), int(n_bins))
    hit = _KMEANS_DRAWS.get(key)
    if hit is None:
        rs = np.random.RandomState(1)
        u = rs.random_sample()
        # searchsorted(cumsum(full(m, 1 / m)) / total, u, 'right') without the m-element arrays (grx_host_uniform_choice)
        idx = ctypes.c_int64(0)
        _lib.call('grx_host_uniform_choice', int(m), float(u), ctypes.byref(idx))
        first = int(idx.value)
        trials = 2 + int(np.log(n_bins))
        uniform = rs.uniform(size=(max(n_bins - 1, 0), trials))
        hit = (first, np.ascontiguousarray(uniform), trials)
        if len(_KMEANS_DRAWS) > 64:
            _KMEANS_DRAWS.clear()
        _KMEANS_DRAWS[key] = hit
    return hit


def kmeans1d(values: torch.Tensor, n_bins: int, max_iter: int = 300, rel_tol: float = 1e-4):
    """encode() as the reference computes it: `values` flat fp64 device tensor IN THE REFERENCE'S FLATTEN ORDER ->
    (quantised tensor, centres [n_bins] in seed order, info int32[4] = {n_iter_, non-empty clusters, distinct
    output values, seeding faults (0: see include/grx.h)})."""
    m = values.numel()
    first, uniform, trials = kmeans_draws(m, int(n_bins))
    out = torch.empty_like(values)
    centers = torch.empty(max(int(n_bins), 1), dtype=torch.float64, device=device())
    info = zeros(4, dtype=torch.int32)
    ws_bytes = _lib.load().grx_kmeans1d_workspace_bytes(m, int(n_bins))
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=device())
    _lib.call('grx_kmeans1d', m, _ptr(values), int(n_bins), first, _hptr(uniform), trials, int(max_iter), float(rel_tol),
              _ptr(out), _ptr(centers), _ptr(info), _ptr(ws), ws_bytes, _stream())
    return out, centers, info


def permute_columns(cols: Sequence[torch.Tensor], index: torch.Tensor, n: int) -> torch.Tensor:
    """[F, n] block with out[c][i] = cols[c][index[i]] (index int32 on the device)."""
    F = len(cols)
    out = torch.empty((F, max(n, 1)), dtype=torch.float64, device=device())
    if F and n:
        _lib.call('grx_permute_columns', n, F, ptr_array(cols), _ptr(index), _ptr(out), _ld(out), _stream())
    return out[:, :n]


def transpose(src: torch.Tensor, rows: int, cols: int) -> torch.Tensor:
    """[rows, >= cols] row-major (leading dimension src.stride(0)) -> contiguous [cols, rows]."""
    out = torch.empty((cols, rows), dtype=torch.float64, device=device())
    _lib.call('grx_transpose', rows, cols, _ptr(src), _ld(src), _ptr(out), rows, _stream())
    return out


def role_argmax(G: torch.Tensor) -> torch.Tensor:
    """n x r row-major node-role factor -> int32[n]: column of the first maximum of every row, -1 for an all-NaN
    row (grx_role_argmax; RoleExtractor.roles, graphrole/roles/extract.py:38-47)."""
    n, r = G.shape
    assert G.is_contiguous() and G.dtype == torch.float64
    out = torch.empty(n, dtype=torch.int32, device=device())
    _lib.call('grx_role_argmax', n, r, _ptr(G), _ptr(out), _stream())
    return out


def row_normalise(G: torch.Tensor) -> torch.Tensor:
    """n x r row-major node-role factor -> every row divided by its sum, summed in Series.sum()'s order
    (grx_row_normalise; RoleExtractor.role_percentage, graphrole/roles/extract.py:49-57)."""
    n, r = G.shape
    assert G.is_contiguous() and G.dtype == torch.float64
    out = torch.empty((n, r), dtype=torch.float64, device=device())
    _lib.call('grx_row_normalise', n, r, _ptr(G), _ptr(out), _stream())
    return out


class NmfState:
    """Device buffers of one multiplicative-update run (X, W feature-major; H r x F).
    H: host array or device tensor [r, F]; x_sq_norm: ||X||_F^2 when known (lets the convergence
    checks use the trace identity of the W-pass outputs instead of a pass over X)."""

    def __init__(self, X: torch.Tensor, n: int, W: torch.Tensor, H, x_sq_norm: Optional[float] = None):
        self.X, self.n, self.W = X, n, W
        self.x_sq_norm = x_sq_norm
        self.info = None               # NmfInfo of the last grx_nmf_mu / grx_nmf_fit on this state
        self.F, self.r = X.shape[0], W.shape[0]
        dev = device()
        if isinstance(H, torch.Tensor):
            self.H = H.to(dev).contiguous()
        else:
            self.H = torch.from_numpy(np.ascontiguousarray(H, dtype=np.float64)).to(dev)
        self.AB = zeros(self.r * self.F + self.r * self.r)
        self.err = zeros(1)
        self.ws_bytes = _lib.load().grx_nmf_workspace_bytes(n, self.F, self.r)
        self.ws = torch.empty(self.ws_bytes, dtype=torch.uint8, device=dev)

    def w_pass(self, row_begin: int = 0, row_end: Optional[int] = None) -> None:
        row_end = self.n if row_end is None else row_end
        _lib.call('grx_nmf_w_pass', self.n, self.F, self.r, _ptr(self.X), _ld(self.X), _ptr(self.W),
                  _ld(self.W), row_begin, row_end, _ptr(self.H), _ptr(self.AB), _ptr(self.ws),
                  self.ws_bytes, _stream())

    def w_pass_next(self, row_begin: int = 0, row_end: Optional[int] = None) -> None:
        """h_update() + w_pass() in one launch less (grx_nmf_w_pass_next): self.AB must hold the (all-reduced) sums
        of the previous pass; afterwards self.H is the updated H and self.AB this rank's new partial sums."""
        row_end = self.n if row_end is None else row_end
        spare = self.__dict__.get('_H_spare')
        if spare is None:
            spare = self._H_spare = torch.empty_like(self.H)
        _lib.call('grx_nmf_w_pass_next', self.n, self.F, self.r, _ptr(self.X), _ld(self.X), _ptr(self.W), _ld(self.W),
                  row_begin, row_end, _ptr(self.H), _ptr(self.AB), _ptr(spare), _ptr(self.AB), _ptr(self.ws),
                  self.ws_bytes, _stream())
        self.H, self._H_spare = spare, self.H

    def h_update(self) -> None:
        _lib.call('grx_nmf_h_update', self.F, self.r, _ptr(self.H), _ptr(self.AB), _stream())

    def residual_sq(self, row_begin: int = 0, row_end: Optional[int] = None) -> torch.Tensor:
        row_end = self.n if row_end is None else row_end
        _lib.call('grx_nmf_residual', self.n, self.F, self.r, _ptr(self.X), _ld(self.X), _ptr(self.W),
                  _ld(self.W), row_begin, row_end, _ptr(self.H), _ptr(self.err), _ptr(self.ws),
                  self.ws_bytes, _stream())
        return self.err

    def kl_cost(self, W: torch.Tensor, H: torch.Tensor, row_begin: int = 0, row_end: Optional[int] = None) -> float:
        """Generalised-KL error cost of X against the (encoded) factors W [r, ld], H [r, F]."""
        row_end = self.n if row_end is None else row_end
        out = zeros(1, dtype=torch.float64)
        _lib.call('grx_nmf_kl_cost', self.n, self.F, self.r, _ptr(self.X), _ld(self.X), _ptr(W), _ld(W),
                  row_begin, row_end, _ptr(H), _ptr(out), _ptr(self.ws), self.ws_bytes, _stream())
        return float(to_host(out)[0])

    def iterate(self, iters: int, with_residual: bool = True) -> None:
        _lib.call('grx_nmf_iterate', self.n, self.F, self.r, _ptr(self.X), _ld(self.X), _ptr(self.W),
                  _ld(self.W), _ptr(self.H), _ptr(self.AB), _ptr(self.err) if with_residual else None,
                  int(iters), _ptr(self.ws), self.ws_bytes, _stream())


# ---- the whole factorisation below the ABI (grx.h: grx_nmf_init / grx_nmf_mu / grx_nmf_fit) ----
def _fit_workspace(n: int, F: int, r: int) -> Tuple[torch.Tensor, int]:
    ws_bytes = _lib.load().grx_nmf_fit_workspace_bytes(n, F, r)
    return torch.empty(ws_bytes, dtype=torch.uint8, device=device()), ws_bytes


def _omega_arg(omega: np.ndarray, F: int) -> np.ndarray:
    omega = np.ascontiguousarray(omega, dtype=np.float64)
    if omega.ndim != 2 or omega.shape[0] != F:
        raise _lib.GrxInvalid(f'omega must be [F, r + 10] with F = {F}, got {omega.shape}')
    return omega


def nmf_init(X: torch.Tensor, n: int, r: int, omega: np.ndarray, shard=None):
    """NNDSVDa start of the feature-major device matrix X [F, ld] (n valid rows, n >= F) in one call:
    (W0 [r, ld] device, H0 [r, F] device, ||X||_F^2)."""
    F = X.shape[0]
    omega = _omega_arg(omega, F)
    W = zeros((r, X.shape[1]), dtype=torch.float64)
    H = torch.empty((r, F), dtype=torch.float64, device=device())
    ws, ws_bytes = _fit_workspace(n, F, r)
    xx = ctypes.c_double(0.0)
    comm, bounds = _shard_args(shard)
    _lib.call('grx_nmf_init', n, F, r, _ptr(X), _ld(X), _hptr(omega), omega.shape[1], _ptr(W), _ld(W), _ptr(H),
              ctypes.byref(xx), comm, bounds, _ptr(ws), ws_bytes, _stream())
    return W, H, float(xx.value)


def nmf_mu(state: NmfState, tol: float, max_iter: int, shard=None) -> int:
    """Multiplicative updates with sklearn's stopping rule on the state's W, H in place; returns n_iter.
    shard: a ShardPlan -- W rows of this rank only, one all-reduce per iteration below the ABI, W completed at the end."""
    comm, bounds = _shard_args(shard)
    ws, ws_bytes = _fit_workspace(state.n, state.F, state.r)
    info = _lib.NmfInfo()
    xx = state.x_sq_norm if state.x_sq_norm is not None else -1.0
    _lib.call('grx_nmf_mu', state.n, state.F, state.r, _ptr(state.X), _ld(state.X), _ptr(state.W), _ld(state.W),
              _ptr(state.H), float(xx), float(tol), int(max_iter), ctypes.byref(info), comm, bounds, _ptr(ws), ws_bytes,
              _stream())
    state.info = info
    return int(info.n_iter)


def nmf_fit(X: torch.Tensor, n: int, r: int, omega: np.ndarray, tol: float, max_iter: int, shard=None):
    """grx_nmf_fit: NNDSVDa + multiplicative updates in ONE call; returns (NmfState, n_iter).  shard: a ShardPlan --
    the row passes cover this rank's rows, the exchanges run below the ABI, W and H end complete on every rank."""
    comm, bounds = _shard_args(shard)
    F = X.shape[0]
    omega = _omega_arg(omega, F)
    W = zeros((r, X.shape[1]), dtype=torch.float64)
    H = torch.empty((r, F), dtype=torch.float64, device=device())
    ws, ws_bytes = _fit_workspace(n, F, r)
    info = _lib.NmfInfo()
    _lib.call('grx_nmf_fit', n, F, r, _ptr(X), _ld(X), _hptr(omega), omega.shape[1], float(tol), int(max_iter),
              _ptr(W), _ld(W), _ptr(H), ctypes.byref(info), comm, bounds, _ptr(ws), ws_bytes, _stream())
    del ws
    state = NmfState(X, n, W, H, x_sq_norm=float(info.x_sq_norm))
    state.info = info
    return state, int(info.n_iter)
