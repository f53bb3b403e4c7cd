"""
Node-range sharding across GPUs (one process per GPU, torch.distributed; backend "nccl" is RCCL
over xGMI on ROCm, "gloo" in the CPU tests).  The reference has no distributed code at all
 -- this is new.

Scheme:
  * the CSR is replicated; rank p computes rows [bounds[p], bounds[p+1]) of every per-node
    kernel (ego-net, aggregation, NMF row passes).  Bounds balance nnz + n, not n.
  * per ReFeX generation the candidate block (all aggregations of all retained columns) never
    travels whole.  Rank p owns candidate columns p, p+P, ...:
      1. all-to-all: every rank sends its row slice of every column to the column's owner
         (volume per rank: candidates x n x 8 / P bytes), the owner bins whole columns;
      2. all-to-all back: the owner returns the uint8 bins of the receiver's rows (each rank only
         ever scans its own rows for the Chebyshev matrix), all-reduce(MAX) of the F x F matrix,
         identical pruning decisions everywhere;
      3. all-gather of the RETAINED new columns only (they are the next generation's gather
         source and part of the result) -- typically 1/3 to 1/6 of the candidates.
    Generation 0 (a handful of columns, complete on every rank) bins by owner and combines the
    bins with all-reduce(MAX).
  * NMF: W rows sharded, H replicated; one all-reduce(SUM) of [W^T X | W^T W] per iteration and
    of the residual at convergence checks.

Where the exchanges run: on the product path (HIP kernels) a ShardPlan is a row partition plus a ``grx_comm``
(include/grx.h, csrc/grx_comm.hip) -- RCCL bound directly from libgrx.so when torch.distributed's backend is
"nccl", a callback transport staged through gloo otherwise (the two-ranks-on-one-GPU test) -- and the whole-loop
drivers grx_refex_run / grx_nmf_fit issue every exchange themselves, below the ABI, between their kernels.  The
methods of this class are thin wrappers over the same C entry points for device tensors (the per-kernel driver,
RoleExtractor's model selection): nothing is packed, no torch compute op runs.  For host tensors (the CPU test double
of tests/, gloo) the same protocol is spelled with torch.distributed calls.
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import ctypes

import numpy as np
import torch
import torch.distributed as dist

_NP_DTYPES = {0: np.float64, 1: np.int32, 2: np.int64, 3: np.uint8}            # grx_dtype


def row_cuts(row_ptr: np.ndarray, world: int) -> np.ndarray:
    """int64[world + 1]: contiguous row ranges with equal shares of nnz + n (the work of a neighbour aggregation: one
    gather per adjacency entry, one output row per node).  The partition of every ShardPlan; tools/project_scaling.py
    times every rank's share of it on one GPU."""
    n = len(row_ptr) - 1
    work = np.asarray(row_ptr[1:], dtype=np.int64) + np.arange(1, n + 1, dtype=np.int64)
    total = int(work[-1]) if n else 0
    cuts = [0]
    for p in range(1, world):
        cuts.append(int(np.searchsorted(work, total * p / world, side='left')))
    cuts.append(n)
    return np.maximum.accumulate(np.array(cuts, dtype=np.int64))


class ShardPlan:

    def __init__(self, row_ptr: np.ndarray, group=None) -> None:
        if not dist.is_available() or not dist.is_initialized():
            raise RuntimeError('ShardPlan needs an initialised torch.distributed process group')
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        n = len(row_ptr) - 1
        self.n = n
        self.bounds = row_cuts(row_ptr, self.world)
        self.row_begin = int(self.bounds[self.rank])
        self.row_end = int(self.bounds[self.rank + 1])
        self.max_rows = int(np.diff(self.bounds).max()) if n else 0
        # a one-rank group skips every exchange -- unless GRX_FORCE_COLLECTIVES=1 (test hook: drives the
        # real RCCL calls, dtypes and split sizes on a one-GPU box, tests/test_gpu_sharded.py)
        self._solo = self.world == 1 and not _force_collectives()
        self._perm_cache: dict = {}
        self._bounds_c = (ctypes.c_int64 * (self.world + 1))(*[int(b) for b in self.bounds])
        self._comm = None               # grx_comm handle (device exchanges), looked up on first use
        #: exchange timing (bench.py --gpus N): name -> [calls, device ms]; off by default
        self._timing = False
        self._timed_events: list = []
        self.exchange_stats: dict = {}

    # ------------------------------------------------------------------ the transport below the ABI
    def bounds_ptr(self):
        """Host int64[world + 1] row partition, as the h_bounds argument of the C entry points."""
        return ctypes.cast(self._bounds_c, ctypes.c_void_p)

    def comm(self):
        """grx_comm handle of this plan (None for a one-rank group without forced exchanges); one communicator per
        process group, shared by every plan of the group (the row partition is an argument of the calls, not a
        property of the transport)."""
        if self._comm is not None or self._solo:
            return self._comm
        self._comm = _communicator(self.group)
        if self._timing:
            from graphrole_amd import _lib
            _lib.call('grx_comm_timing', self._comm, 1)
        return self._comm

    @property
    def timing(self) -> bool:
        return self._timing

    @timing.setter
    def timing(self, on: bool) -> None:
        self._timing = bool(on)
        if self._comm is not None:
            from graphrole_amd import _lib
            _lib.call('grx_comm_timing', self._comm, int(self._timing))

    # ------------------------------------------------------------------ exchange timing
    def _time(self, name: str):
        """Context manager: HIP events around one exchange on the current stream (device time of the packing
        kernels + the collective); a no-op unless self.timing."""
        plan = self

        class _Scope:
            def __enter__(self_inner):
                self_inner.on = plan.timing and torch.cuda.is_available() and plan._comm is None
                if self_inner.on:
                    self_inner.start = torch.cuda.Event(enable_timing=True)
                    self_inner.stop = torch.cuda.Event(enable_timing=True)
                    self_inner.start.record()
                return self_inner

            def __exit__(self_inner, *exc):
                if self_inner.on:
                    self_inner.stop.record()
                    plan._timed_events.append((name, self_inner.start, self_inner.stop))
                return False

        return _Scope()

    def collect_timing(self) -> dict:
        """Synchronise and fold the recorded events into exchange_stats {name: [calls, ms]}."""
        if self._timed_events:
            torch.cuda.synchronize()
            for name, start, stop in self._timed_events:
                cell = self.exchange_stats.setdefault(name, [0, 0.0])
                cell[0] += 1
                cell[1] += start.elapsed_time(stop)
            self._timed_events = []
        if self._comm is not None:
            from graphrole_amd import _lib
            for kind, name in enumerate(_lib.COMM_KINDS):
                calls, ms = ctypes.c_longlong(0), ctypes.c_double(0.0)
                _lib.call('grx_comm_timing_read', self._comm, kind, ctypes.byref(calls), ctypes.byref(ms))
                if calls.value:
                    cell = self.exchange_stats.setdefault(name, [0, 0.0])
                    cell[0] += calls.value
                    cell[1] += ms.value
            _lib.call('grx_comm_timing_reset', self._comm)
        return self.exchange_stats

    def reset_timing(self) -> None:
        self._timed_events = []
        self.exchange_stats = {}
        if self._comm is not None:
            from graphrole_amd import _lib
            _lib.call('grx_comm_timing_reset', self._comm)

    # ------------------------------------------------------------------ collectives
    @staticmethod
    def _stream():
        return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

    @staticmethod
    def _col_ptrs(tensors):
        return (ctypes.c_void_p * max(len(tensors), 1))(*[t.data_ptr() for t in tensors])

    def all_gather_block(self, block: torch.Tensor) -> torch.Tensor:
        """block [ncols, n] with only this rank's row slice valid -> every slice valid, in place."""
        if self._solo:
            return block
        ncols = block.shape[0]
        if ncols == 0 or self.n == 0:
            return block
        if block.is_cuda:
            assert block.stride(1) == 1
            self.all_gather_columns([block[j] for j in range(ncols)])
            return block
        with self._time('all_gather_block'):
            return self._all_gather_block(block, ncols)

    def _all_gather_block(self, block: torch.Tensor, ncols: int) -> torch.Tensor:
        # host tensors (CPU test double)
        send = torch.zeros((ncols, self.max_rows), dtype=block.dtype)
        send[:, :self.row_end - self.row_begin] = block[:, self.row_begin:self.row_end]
        flat = torch.empty((self.world * ncols, self.max_rows), dtype=block.dtype)
        dist.all_gather_into_tensor(flat, send, group=self.group)      # concatenates along dim 0
        recv = flat.view(self.world, ncols, self.max_rows)
        # one concatenation + one copy instead of a slice copy per rank (every small launch is host time
        # on the critical path of a sharded generation)
        pieces = [recv[p, :, :int(self.bounds[p + 1] - self.bounds[p])] for p in range(self.world)
                  if self.bounds[p + 1] > self.bounds[p]]
        block.copy_(torch.cat(pieces, dim=1))
        return block

    def _all_to_all(self, recv: torch.Tensor, send: torch.Tensor, out_split, in_split) -> None:
        dist.all_to_all_single(recv, send, output_split_sizes=out_split, input_split_sizes=in_split, group=self.group)

    def columns_to_owners(self, block: torch.Tensor) -> torch.Tensor:
        """block [ncols, n] with only this rank's row slice valid -> the WHOLE columns this rank owns
        (columns rank, rank + world, ...) as a [n_owned, n] tensor."""
        if block.is_cuda and self._solo:
            return block[:, :self.n]                       # a one-rank group owns every column, rows complete
        if block.is_cuda:
            from graphrole_amd import _lib
            ncols = block.shape[0]
            n_owned = len(range(self.rank, ncols, self.world))
            owned = torch.empty((n_owned, max(self.n, 1)), dtype=block.dtype, device=block.device)
            assert ncols <= 1 or block.stride(1) == 1
            _lib.call('grx_comm_columns_to_owners', self.comm(), self.bounds_ptr(), ncols, ctypes.c_void_p(block.data_ptr()),
                      block.stride(0) if ncols > 1 else max(self.n, 1), block.element_size(),
                      ctypes.c_void_p(owned.data_ptr()), owned.stride(0) if n_owned else max(self.n, 1), self._stream())
            return owned[:, :self.n]
        with self._time('columns_to_owners'):
            return self._columns_to_owners(block)

    def _columns_to_owners(self, block: torch.Tensor) -> torch.Tensor:
        ncols = block.shape[0]
        rb, re = self.row_begin, self.row_end
        counts, perm = self._owner_order(ncols, block.device)
        # columns in owner-major order (owner 0's columns, then owner 1's, ...): one gather launch
        send = block[:, rb:re].index_select(0, perm).reshape(-1)
        in_split = [c * (re - rb) for c in counts]
        n_owned = counts[self.rank]
        rows = [int(self.bounds[p + 1] - self.bounds[p]) for p in range(self.world)]
        out_split = [n_owned * r for r in rows]
        recv = torch.empty(sum(out_split), dtype=block.dtype, device=block.device)
        self._all_to_all(recv, send, out_split, in_split)
        if n_owned == 0:
            return torch.empty((0, self.n), dtype=block.dtype, device=block.device)
        pieces, off = [], 0
        for r in rows:
            if r:
                pieces.append(recv[off:off + n_owned * r].view(n_owned, r))
            off += n_owned * r
        return torch.cat(pieces, dim=1)                             # [n_owned, n] in one launch

    def owned_to_rows(self, owned: torch.Tensor, ncols: int) -> torch.Tensor:
        """Inverse direction for per-column results (uint8 bins): owned [n_owned, n] whole columns ->
        [ncols, n] with this rank's row slice of EVERY column valid (other rows: unspecified on the device path,
        zero on the host path -- each rank only ever scans its own rows)."""
        if owned.is_cuda and self._solo:
            assert owned.shape[0] == ncols
            return owned[:, :self.n]
        if owned.is_cuda:
            from graphrole_amd import _lib
            out = torch.empty((ncols, max(self.n, 1)), dtype=owned.dtype, device=owned.device)
            assert owned.shape[0] <= 1 or owned.stride(1) == 1
            _lib.call('grx_comm_owned_to_rows', self.comm(), self.bounds_ptr(), ncols, ctypes.c_void_p(owned.data_ptr()),
                      owned.stride(0) if owned.shape[0] > 1 else max(self.n, 1), owned.element_size(),
                      ctypes.c_void_p(out.data_ptr()), out.stride(0), self._stream())
            return out[:, :self.n]
        with self._time('owned_to_rows'):
            return self._owned_to_rows(owned, ncols)

    def _owned_to_rows(self, owned: torch.Tensor, ncols: int) -> torch.Tensor:
        rb, re = self.row_begin, self.row_end
        n_owned = owned.shape[0]
        counts, perm = self._owner_order(ncols, owned.device)
        assert n_owned == counts[self.rank]
        rows = [int(self.bounds[p + 1] - self.bounds[p]) for p in range(self.world)]
        in_split = [n_owned * r for r in rows]
        send = torch.empty(sum(in_split), dtype=owned.dtype, device=owned.device)
        off = 0
        for p, r in enumerate(rows):                                # the row slice each rank gets back
            if n_owned and r:
                send[off:off + n_owned * r].view(n_owned, r).copy_(owned[:, int(self.bounds[p]):int(self.bounds[p + 1])])
            off += n_owned * r
        out_split = [c * (re - rb) for c in counts]
        recv = torch.empty(sum(out_split), dtype=owned.dtype, device=owned.device)
        self._all_to_all(recv, send, out_split, in_split)
        out = torch.zeros((ncols, self.n), dtype=owned.dtype, device=owned.device)
        if ncols and re > rb:
            # the received chunks are the columns in owner-major order: one scatter back to column order
            out[:, rb:re].index_copy_(0, perm, recv.view(ncols, re - rb))
        return out

    def _owner_order(self, ncols: int, device) -> Tuple[List[int], torch.Tensor]:
        """(columns per owner, column indices in owner-major order) for the round-robin column split."""
        key = (ncols, str(device))
        hit = self._perm_cache.get(key)
        if hit is None:
            counts = [len(range(q, ncols, self.world)) for q in range(self.world)]
            order = [c for q in range(self.world) for c in range(q, ncols, self.world)]
            hit = (counts, torch.tensor(order, dtype=torch.int64, device=device))
            self._perm_cache[key] = hit
        return hit

    def all_gather_columns(self, cols: Sequence[torch.Tensor]) -> List[torch.Tensor]:
        """Same for a list of separate [n] columns (device columns: completed in place, where they lie)."""
        if self._solo or not cols:
            return list(cols)
        if cols[0].is_cuda:
            from graphrole_amd import _lib
            cols = list(cols)
            assert all(c.is_contiguous() and c.dtype == cols[0].dtype for c in cols)
            _lib.call('grx_comm_all_gather_rows', self.comm(), self.bounds_ptr(), len(cols), self._col_ptrs(cols),
                      cols[0].element_size(), self._stream())
            return cols
        block = torch.stack(list(cols))
        self.all_gather_block(block)
        return [block[j] for j in range(len(cols))]

    def _all_reduce_(self, t: torch.Tensor, op) -> torch.Tensor:
        if not self._solo and t.numel():
            if t.is_cuda:
                from graphrole_amd import _lib
                assert t.is_contiguous()
                dtype = _lib.DTYPE_IDS[str(t.dtype).replace('torch.', '')]
                _lib.call('grx_comm_all_reduce', self.comm(), ctypes.c_void_p(t.data_ptr()), t.numel(), dtype,
                          0 if op == dist.ReduceOp.SUM else 1, self._stream())
            elif t.is_contiguous():
                dist.all_reduce(t, op=op, group=self.group)
            else:
                c = t.contiguous()
                dist.all_reduce(c, op=op, group=self.group)
                t.copy_(c)
        return t

    def _small_device(self) -> torch.device:
        return torch.device('cuda', torch.cuda.current_device()) if dist.get_backend(self.group) == 'nccl' \
            else torch.device('cpu')

    def all_reduce_sum_host(self, a: np.ndarray) -> np.ndarray:
        """Sum of a small host array over the ranks (identical bits on every rank)."""
        if self._solo:
            return a
        t = torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).to(self._small_device())
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        from graphrole_amd import kernels as K
        return (K.to_host(t) if t.is_cuda else t.numpy()).reshape(a.shape)

    def all_gather_host(self, a: np.ndarray) -> np.ndarray:
        """[world, *a.shape]: the small host array of every rank."""
        a = np.ascontiguousarray(a, dtype=np.float64)
        if self._solo:
            return a[None]
        t = torch.from_numpy(a).to(self._small_device()).reshape(1, -1)
        out = torch.empty((self.world, t.shape[1]), dtype=t.dtype, device=t.device)
        dist.all_gather_into_tensor(out, t, group=self.group)
        from graphrole_amd import kernels as K
        return (K.to_host(out) if out.is_cuda else out.numpy()).reshape((self.world,) + a.shape)

    def all_reduce_max_(self, t: torch.Tensor) -> torch.Tensor:
        with self._time('all_reduce_max'):
            return self._all_reduce_(t, dist.ReduceOp.MAX)

    def all_reduce_sum_(self, t: torch.Tensor) -> torch.Tensor:
        with self._time('all_reduce_sum'):
            return self._all_reduce_(t, dist.ReduceOp.SUM)


class _StagedTransport:
    """Callback transport for torch.distributed backends without device collectives (gloo): device buffers are
    staged through host memory with the library's own copies.  Used by the two-ranks-on-one-GPU tests; the
    product transport is RCCL (grx_comm_create_rccl)."""

    def __init__(self, group) -> None:
        from graphrole_amd import _lib
        self.group = group
        self.world = dist.get_world_size(group)
        self.callbacks = (_lib.ALL_REDUCE_FN(self.all_reduce), _lib.EXCHANGE_FN(self.exchange))   # kept alive here

    def all_reduce(self, user, d_buf, count, dtype, op, stream) -> int:
        try:
            from graphrole_amd import _lib
            host = np.empty(count, dtype=_NP_DTYPES[dtype])
            _lib.call('grx_memcpy_d2h', host.ctypes.data_as(ctypes.c_void_p), d_buf, host.nbytes, stream)
            _lib.call('grx_stream_sync', stream)
            dist.all_reduce(torch.from_numpy(host), op=dist.ReduceOp.SUM if op == 0 else dist.ReduceOp.MAX,
                            group=self.group)
            _lib.call('grx_memcpy_h2d', d_buf, host.ctypes.data_as(ctypes.c_void_p), host.nbytes, stream)
            _lib.call('grx_stream_sync', stream)
            return 0
        except Exception:                                    # an exception must not unwind through the C frames
            import traceback
            traceback.print_exc()
            return -2

    def exchange(self, user, n_ops, ops, stream) -> int:
        try:
            from graphrole_amd import _lib
            sends = [[] for _ in range(self.world)]
            recvs = [[] for _ in range(self.world)]
            for i in range(n_ops):
                op = ops[i]
                if op.is_recv:
                    recvs[op.peer].append((op.d_ptr, int(op.bytes)))
                else:
                    buf = np.empty(int(op.bytes), dtype=np.uint8)
                    _lib.call('grx_memcpy_d2h', buf.ctypes.data_as(ctypes.c_void_p), op.d_ptr, buf.nbytes, stream)
                    sends[op.peer].append(buf)
            _lib.call('grx_stream_sync', stream)
            in_split = [sum(b.nbytes for b in sends[q]) for q in range(self.world)]
            out_split = [sum(nb for _, nb in recvs[q]) for q in range(self.world)]
            flat = [b for q in range(self.world) for b in sends[q]]
            send_t = torch.from_numpy(np.concatenate(flat) if flat else np.empty(0, dtype=np.uint8))
            recv_t = torch.empty(sum(out_split), dtype=torch.uint8)
            # between one pair of ranks, transfers are matched in op order: concatenation keeps that order
            dist.all_to_all_single(recv_t, send_t, output_split_sizes=out_split, input_split_sizes=in_split,
                                   group=self.group)
            host = recv_t.numpy()
            off = 0
            for q in range(self.world):
                for d_ptr, nb in recvs[q]:
                    _lib.call('grx_memcpy_h2d', d_ptr, host[off:off + nb].ctypes.data_as(ctypes.c_void_p), nb, stream)
                    off += nb
            _lib.call('grx_stream_sync', stream)
            return 0
        except Exception:
            import traceback
            traceback.print_exc()
            return -2


_COMMUNICATORS: dict = {}


def _communicator(group):
    """The grx_comm of a torch.distributed process group, created once per process.  Backend "nccl": RCCL bound inside
    libgrx.so -- rank 0 draws the unique id, torch.distributed carries it to the others (the NCCL bootstrap pattern).
    Any other backend: the staged callback transport."""
    from graphrole_amd import _lib
    backend = dist.get_backend(group)
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    forced = _force_collectives()
    # keyed on the process-group OBJECT (the default group resolved to its object), which the cache entry keeps alive:
    # an id() cannot be recycled by a group created after this one was destroyed, and a re-initialised default group
    # is a new object -- it gets a new communicator instead of the stale one
    pg = group if group is not None else dist.group.WORLD
    key = (id(pg), backend, rank, world, forced)
    hit = _COMMUNICATORS.get(key)
    if hit is not None:
        return hit[0]
    handle = ctypes.c_void_p()
    keep = None
    if backend == 'nccl':
        box = [None]
        if rank == 0:
            buf = (ctypes.c_char * _lib.COMM_ID_BYTES)()
            _lib.call('grx_comm_rccl_unique_id', ctypes.cast(buf, ctypes.c_void_p))
            box = [bytes(buf)]
        src = 0 if group is None else dist.get_global_rank(group, 0)
        dist.broadcast_object_list(box, src=src, group=group)
        ident = ctypes.create_string_buffer(box[0], _lib.COMM_ID_BYTES)
        _lib.call('grx_comm_create_rccl', ctypes.cast(ident, ctypes.c_void_p), rank, world,
                  _lib.COMM_SELF_VIA_TRANSPORT if forced else 0, ctypes.byref(handle))
    else:
        keep = _StagedTransport(group)
        _lib.call('grx_comm_create_callbacks', rank, world, keep.callbacks[0], keep.callbacks[1], None,
                  ctypes.byref(handle))
    _COMMUNICATORS[key] = (handle, keep, pg)
    return handle


def _force_collectives() -> bool:
    import os
    return os.environ.get('GRX_FORCE_COLLECTIVES') == '1'


def maybe_plan(row_ptr: np.ndarray, distributed) -> Optional[ShardPlan]:
    """
    distributed: None/False -> single GPU; True -> default process group (if initialised with
    world size > 1); a ProcessGroup -> that group.
    """
    if not distributed:
        return None
    if not (dist.is_available() and dist.is_initialized()):
        if distributed is True:
            return None
        raise RuntimeError('distributed= was given but torch.distributed is not initialised')
    group = None if distributed is True else distributed
    if dist.get_world_size(group) == 1 and not _force_collectives():
        return None
    return ShardPlan(row_ptr, group)
