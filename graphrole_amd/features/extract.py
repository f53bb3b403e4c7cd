"""
ReFeX recursive feature extraction on MI355X (reference: graphrole/features/extract.py).

Same public surface as the reference's ``RecursiveFeatureExtractor`` -- constructor signature,
``extract_features()``, ``generation_count``, error types, column names / order, row order,
dtypes, memoisation -- but all features live as fp64 columns in HBM from generation 0 until the
final DataFrame is assembled: one device column per feature, never a dict of dicts.

Per generation (device kernels in graphrole_amd/csrc, host decisions in features/prune.py):
    pack retained columns -> grx_aggregate (sum, mean over neighbours) -> grx_vertical_log_bin of
    the new columns (bins of older columns are cached: re-binning is result-identical,
    prune.py:101-104) -> grx_chebyshev -> tiny host step: feature-graph components, drop list.
"""
from __future__ import annotations

import time
from collections import OrderedDict
from typing import Dict, List, Optional, Sequence

import numpy as np
import pandas as pd

from graphrole_amd.features.prune import FeaturePruner
from graphrole_amd.graph import interface
from graphrole_amd.types import DataFrameDict, DataFrameLike

_SUPPORTED_AGGS = ('sum', 'mean', 'min', 'max', 'var', 'std', 'prod', 'median', 'count', 'size')
_FAST_AGGS = ('sum', 'mean', 'min', 'max', 'var', 'std')             # the tuned pairwise-order kernels
_FLOAT_AGGS = ('mean', 'std', 'var', 'median')                       # their result makes the whole agg frame float64
_INT_SAFE_ON_EMPTY = ('sum', 'prod', 'count', 'size')                # integers even for a node without neighbours


def _agg_name(agg) -> str:
    """'sum' / np.sum / pd.DataFrame.sum / a user function -> the name pandas puts in the result index
    (pandas.core.common.get_callable_name: __name__, the wrapped function of a functools.partial, else the class)."""
    if isinstance(agg, str):
        return agg
    import functools
    if hasattr(agg, '__name__'):
        return agg.__name__
    if isinstance(agg, functools.partial):
        return _agg_name(agg.func)
    if callable(agg):
        return type(agg).__name__
    raise TypeError(f'cannot interpret aggregation {agg!r}')


def _kernel_functions() -> dict:
    """id(function) -> name for the function OBJECTS that stand for the ten kernel-backed aggregations: the numpy,
    pandas and builtin functions pandas itself maps to those names.  Matched by identity: a function that merely
    carries the name 'sum' in a module starting with numpy / pandas (np.ma.sum, a pandas-internal helper with other NaN
    or ddof rules, a user function) is NOT one of them and runs on the host like any other callable."""
    import builtins
    table = {}
    for name in _SUPPORTED_AGGS:
        for owner in (np, pd.DataFrame, pd.Series, builtins):
            fn = getattr(owner, name, None)
            if callable(fn) and getattr(fn, '__name__', None) == name:
                table[id(fn)] = name
    # (np.amin / np.amax carry their own __name__ -- the label pandas puts in the result -- and therefore run on the host
    # like any other callable: the kernels are selected by the names above)
    return table


_KERNEL_FUNCTIONS = _kernel_functions()


def _has_kernel(agg) -> bool:
    """True for the spellings of the ten aggregations that have device kernels: their names as strings, and the numpy /
    pandas / builtin function objects pandas itself maps to those names (np.sum, pd.DataFrame.mean, max, ...).  Any
    OTHER callable -- also a user function that happens to be called 'sum' -- is evaluated on the host (extract.py:111)."""
    if isinstance(agg, str):
        return agg in _SUPPORTED_AGGS
    return callable(agg) and _KERNEL_FUNCTIONS.get(id(agg)) == getattr(agg, '__name__', None)


class RecursiveFeatureExtractor:

    """ Compute recursive features for nodes of a graph """

    supported_graph_libs = interface.get_supported_graph_libraries()

    default_aggs = [
        pd.DataFrame.sum,
        pd.DataFrame.mean,
    ]

    def __init__(
        self,
        G,
        max_generations: int = 10,
        aggs: Optional[List] = None,
        **kwargs
    ) -> None:
        """
        :param G: graph object from a supported graph package (networkx, graphrole_amd.CSRGraph)
        :param max_generations: maximum levels of recursion
        :param aggs: optional list of aggregations for each recursive generation
          ('sum', 'mean', 'min', 'max', 'std', 'var', 'prod', 'median', 'count', 'size' in any spelling pandas
          accepts run as device kernels; the remaining pandas NAMES raise NotImplementedError).  Any other CALLABLE
          -- what the reference hands to ``DataFrame.agg`` (extract.py:26,111) -- is evaluated by pandas on the
          host, node by node, over neighbour rows the device gathered: the reference's own procedure at the
          reference's speed (~0.5 ms per node and generation), single GPU only, documented in DESIGN.md section 7.
          With 'prod' the integer columns of an unweighted graph follow the reference's wrapping int64
          arithmetic (csrc/grx_aggx.hip)
        :kwargs: attributes / attributes_include / attributes_exclude for the graph interface;
          distributed=True|ProcessGroup shards node ranges over the ranks of torch.distributed;
          native_loop=False drives the generations from Python even on one GPU (tests)
        """
        distributed = kwargs.pop('distributed', None)
        kwargs_native = bool(kwargs.pop('native_loop', True))
        graph_class = interface.get_interface(G)
        if graph_class is None:
            raise TypeError(f'Input graph G must be from one of the following '
                            f'supported libraries: {self.supported_graph_libs}')

        graph = graph_class(G, **kwargs)
        if graph.get_num_edges() == 0:
            raise ValueError('Input graph G must contain at least one edge')

        self.graph = graph
        self.max_generations = max_generations
        self.aggs = aggs if aggs else self.default_aggs
        self._distributed = distributed

        # current generation
        self.generation_count = 0
        # distance threshold for grouping binned features; equals generation_count
        self._feature_group_thresh = 0

        # device feature store --------------------------------------------------------
        self._work: 'OrderedDict[str, object]' = OrderedDict()     # working set: name -> fp64 column
        self._work_bins: Dict[str, object] = {}                    # name -> uint8 binned column
        self._dtypes: Dict[str, np.dtype] = {}                     # pandas dtype of each column
        self._final_names: Dict[int, List[str]] = {}               # generation -> recorded names
        self._final_cols: Dict[str, object] = {}                   # recorded name -> fp64 column
        self._int32_exact: set = set()                             # generation-0 names that may travel as int32 rows
        self._i64: set = set()        # columns stored as int64 BITS (integer columns of a run with 'prod': they wrap)
        self._plan = None
        self._plan_ready = False
        self._arena = None            # device memory of grx_refex_run, reused by later runs of this instance
        #: False drives the generation loop from Python, kernel by kernel (the path a ShardPlan uses)
        self._native_loop = kwargs_native
        #: per-generation statistics (candidates, retained, widths) for benchmarks
        self.stats: List[Dict] = []
        #: wall-clock breakdown of the last extract_features() call (seconds): device run, permutation, download,
        #: DataFrame construction, hand-off hashes.  Set ``time_phases = True`` to have the device synchronised
        #: between the phases (otherwise the asynchronous kernels are charged to the download that waits for them)
        self.wall: Dict[str, float] = {}
        self.time_phases = False

    # ------------------------------------------------------------------ plumbing
    def _K(self):
        return self.graph._K()

    def _labels(self) -> list:
        return self.graph.to_csr().labels

    def _n(self) -> int:
        return self.graph.to_csr().n

    def _order(self):
        """InternalGraph of the adapter: device rows are in degree-descending internal order."""
        return self.graph._device_graph()[0]

    def _shard(self):
        if not self._plan_ready:
            from graphrole_amd.parallel import maybe_plan
            self._plan = maybe_plan(self._order().row_ptr, self._distributed) if self._distributed else None
            self.graph._shard_plan = self._plan
            self._plan_ready = True
        return self._plan

    def _agg_names(self) -> List[str]:
        names = [_agg_name(a) for a in self.aggs]
        if self._host_callables() and len(set(names)) != len(names):
            # pandas: SpecificationError('Function names must be unique if there is no new column names assigned')
            raise ValueError(f'Function names must be unique: {names}')
        return names

    def _host_callables(self) -> Dict[str, object]:
        """name -> entry for the entries of ``aggs`` that have no device kernel: callables, and any other aggregation
        NAME pandas knows ('sem', 'skew', 'nunique', 'first', ...) -- the reference hands the list to DataFrame.agg
        as it is (extract.py:26,47,111), so these are evaluated by pandas on the host over device-gathered neighbour
        rows; a name pandas does not know fails there with pandas' own AttributeError, as in the reference."""
        return {_agg_name(a): a for a in self.aggs if not _has_kernel(a)}

    # ------------------------------------------------------------------ public API
    def extract_features(self) -> DataFrameLike:
        """
        Perform recursive feature extraction to return DataFrame of features
        """
        # return already calculated features if stored in state (extract.py:70-71)
        self.wall = {}
        if not self._final_names:
            t0 = time.perf_counter()
            self.run_on_device()
            self._phase_sync()
            self.wall['device_run_s'] = time.perf_counter() - t0
        return self._finalize_features()

    def _phase_sync(self) -> None:
        if self.time_phases:
            sync = getattr(self._K(), 'synchronize', None)
            if sync is not None:
                sync()

    def run_on_device(self) -> None:
        """
        The generation loop without the final host DataFrame: afterwards every recorded feature
        is an fp64 column in HBM (``device_features()``).  Benchmarks and device-to-device
        hand-off to the NMF use this entry point.
        """
        if self._final_names:
            return
        aggs = self._agg_names()
        self._shard()

        # generation 0: neighbourhood (local + ego-net) features
        t_gen0 = time.perf_counter()
        names, cols, dtypes = self.graph.neighborhood_feature_columns()
        K = self._K()
        if self.time_phases:
            self._phase_sync()
            self.wall['generation0_s'] = time.perf_counter() - t_gen0
        native = hasattr(K, 'refex_run') and (self._plan is None or getattr(K, 'NATIVE_SHARDING', False))
        # 'prod' over INTEGER columns follows the reference's wrapping int64 arithmetic: those columns are carried as
        # int64 bits from generation 0 on, and the generations are driven from here (the per-kernel driver)
        wrapping = 'prod' in aggs and any(np.dtype(dt).kind in 'iu' for dt in dtypes)
        if wrapping:
            if not hasattr(K, 'convert_f64_to_i64'):
                raise NotImplementedError("'prod' over integer features needs the int64 kernels of libgrx.so")
            cols = [K.convert_f64_to_i64(c) if np.dtype(dt).kind in 'iu' else c for c, dt in zip(cols, dtypes)]
            self._i64 = {nm for nm, dt in zip(names, dtypes) if np.dtype(dt).kind in 'iu'}
        if native and self._native_loop and not wrapping and len(set(aggs)) == len(aggs) and not self._host_callables():
            # the whole generation loop runs below the ABI (grx_refex_run); with a ShardPlan it aggregates this rank's
            # rows and issues the exchanges itself
            self._run_native(names, cols, dtypes, aggs)
            return
        flags = self._int32_flags(names, dtypes)
        self._int32_exact = {nm for nm, ok in zip(names, flags or []) if ok}
        self._update_columns(names, cols, dtypes)

        for generation in range(1, self.max_generations):

            self.generation_count = generation
            self._feature_group_thresh = generation

            names, cols, dtypes, block = self._next_feature_columns()
            self._update_columns(names, cols, dtypes, block)

            # stop if an iteration results in no features retained
            if not self._final_names[generation]:
                break

    def _run_native(self, names0, cols0, dtypes0, aggs) -> None:
        """grx_refex_run + the bookkeeping the DataFrame views need: names from the lineage table, pandas
        dtypes, the final working set, per-generation statistics."""
        K = self._K()
        _, dev_graph, _ = self.graph._device_graph()
        flags = self._int32_flags(names0, dtypes0)
        t_loop = time.perf_counter()
        columns, generations, gen_count, self._arena = K.refex_run(dev_graph, cols0, names0, self.max_generations,
                                                                   aggs, self._arena, shard=self._plan, gen0_int32=flags)
        if self.time_phases:
            self._phase_sync()
            self.wall['generation_loop_s'] = time.perf_counter() - t_loop
            self.wall['arena_bytes'] = int(sum(c.numel() for c in self._arena)) if self._arena else 0
            self.wall['generation_loop_attempts'] = int(getattr(K.refex_run, 'attempts', 1))
            self.wall['generation_loop_trace'] = [[what, round(sec, 6)] for what, sec in getattr(K.refex_run, 'trace', [])]
        host = self.graph._device_graph()[0]
        no_empty_rows = self._no_empty_rows(host)
        names: List[str] = []
        for c in columns:
            if c['gen0_index'] >= 0:
                nm = names0[c['gen0_index']]
                self._dtypes[nm] = dtypes0[c['gen0_index']]
            else:
                parent = names[c['parent']]
                nm = f"{parent}({c['agg']})"
                self._dtypes[nm] = self._candidate_dtype(self._dtypes[parent], aggs, no_empty_rows)
            names.append(nm)
            self._final_names.setdefault(c['generation'], []).append(nm)
            self._final_cols[nm] = c['col']
        for g in range(gen_count + 1):
            self._final_names.setdefault(g, [])
        in_work = sorted((c['work_position'], nm) for c, nm in zip(columns, names) if c['work_position'] >= 0)
        for _, nm in in_work:
            self._work[nm] = self._final_cols[nm]
        self.generation_count = gen_count
        self._feature_group_thresh = gen_count
        self.stats = generations

    def _int32_flags(self, names, dtypes):
        """Which generation-0 columns hold exact integers in [0, 2^31) (adapter knowledge; None = unknown)."""
        probe = getattr(self.graph, 'int32_exact_flags', None)
        return None if probe is None else probe(names, dtypes)

    @staticmethod
    def _no_empty_rows(host) -> bool:
        """Every node has a neighbour -- a property of the graph, computed once (np.diff over millions of rows
        costs milliseconds per call)."""
        hit = host.__dict__.get('_no_empty_rows')
        if hit is None:
            hit = host.__dict__['_no_empty_rows'] = bool(host.n == 0 or np.diff(host.row_ptr).min() > 0)
        return hit

    @staticmethod
    def _candidate_dtype(parent_dtype, aggs, no_empty_rows: bool):
        """pandas dtype of a candidate column in the reference's frame (extract.py:104-119): the per-node
        agg frame of an integer column stays integer unless 'mean' / 'std' / 'var' / 'median' is among the aggs or
        a node without neighbours turns min / max into NaN -> 0.0 (sum, prod, count and size of nothing are the
        integers 0, 1, 0, 0); one float value makes the column float64."""
        keeps_int = not (set(_FLOAT_AGGS) & set(aggs)) and (no_empty_rows or set(aggs) <= set(_INT_SAFE_ON_EMPTY))
        f64 = np.dtype('float64')
        return np.dtype('int64') if keeps_int and np.dtype(parent_dtype).kind in 'iu' else f64

    def reset(self) -> None:
        """Forget all computed features (the graph stays resident in HBM)."""
        self.generation_count = 0
        self._feature_group_thresh = 0
        self._work.clear()
        self._work_bins.clear()
        self._final_names = {}
        self._final_cols = {}
        self._i64 = set()
        self.stats = []

    def final_columns(self) -> List[str]:
        """Recorded feature names in output order: latest generation first (extract.py:95)."""
        columns: List[str] = []
        for gen in sorted(self._final_names, reverse=True):
            columns.extend(nm for nm in self._final_names[gen] if nm not in columns)
        return columns

    def device_features(self):
        """(names, device columns) of the final features, in output order, without leaving HBM."""
        names = self.final_columns()
        return names, [self._final_cols[nm] for nm in names]

    # ------------------------------------------------------------------ engine
    def _next_feature_columns(self, complete: bool = False):
        """Candidate columns of the current generation: every aggregation of every column
        retained in the previous generation (extract.py:98-119,152-162)."""
        K = self._K()
        plan = self._shard()
        n = self._n()
        prev = list(self._final_names[self.generation_count - 1])
        aggs = self._agg_names()
        f = len(prev)
        if f == 0:
            return [], [], [], None
        _, dev_graph, _ = self.graph._device_graph()
        rb, re = (0, n) if plan is None else (plan.row_begin, plan.row_end)
        host = self.graph._device_graph()[0]
        no_empty_rows = self._no_empty_rows(host)
        f64 = np.dtype('float64')
        names = [f'{c}({a})' for a in aggs for c in prev]
        dtypes = [self._candidate_dtype(self._dtypes.get(c, f64), aggs, no_empty_rows) for a in aggs for c in prev]
        callables = self._host_callables()
        if callables:
            if plan is not None:
                raise NotImplementedError('callable aggregations are evaluated on the host of ONE process; '
                                          'distributed= is not supported with them')
            sub, dtypes = self._aggregate_with_callables(K, dev_graph, prev, aggs, callables, names, dtypes, n)
        elif set(aggs) <= set(_FAST_AGGS) and not (self._i64 & set(prev)):
            sub = self._aggregate_fast(K, dev_graph, prev, aggs, n, rb, re)
        else:
            sub = self._aggregate_general(K, dev_graph, prev, aggs, names, dtypes, n, rb, re)
        # sharded: only this rank's rows of the candidate block are valid from here on; the
        # exchanges happen in _update_columns (owners bin whole columns, only retained ones are
        # gathered everywhere) -- or right here when a caller wants complete columns
        if plan is not None and complete:
            plan.all_gather_block(sub)
        self._partial_block = sub if (plan is not None and not complete) else None
        cols = [sub[j] for j in range(len(names))]
        return names, cols, dtypes, sub

    def _aggregate_fast(self, K, dev_graph, prev, aggs, n, rb, re):
        """sum / mean / min / max / var / std of fp64 columns: the tuned kernels (numpy's pairwise order); the
        [len(aggs) * f, n] candidate block in aggregation-major order."""
        f = len(prev)
        pieces = {}
        need_var = 'var' in aggs or 'std' in aggs
        # generation 1 of an unweighted graph: every parent an exact int32 column and only sums / means wanted -> the
        # integer gather source (the same choice grx_refex_run makes)
        int_rows = (set(aggs) <= {'sum', 'mean'} and hasattr(K, 'aggregate_i32') and
                    all(c in self._int32_exact for c in prev) and K.aggregate_i32_ok(dev_graph, f))
        block = None
        if int_rows:
            irows, ldi = K.pack_rows_i32([self._work[c] for c in prev], n)
            block = K.aggregate_i32(dev_graph, irows, f, ldi, rb, re, want_sum='sum' in aggs, want_mean='mean' in aggs)
            pieces['sum'], pieces['mean'] = block[:f], block[f:]
            rows = ldr = None
        else:
            rows, ldr = K.pack_rows([self._work[c] for c in prev], n)
        if not int_rows and ('sum' in aggs or 'mean' in aggs or need_var):
            block = K.aggregate(dev_graph, rows, f, ldr, rb, re,
                                want_sum='sum' in aggs, want_mean='mean' in aggs or need_var)
            pieces['sum'], pieces['mean'] = block[:f], block[f:]
        if need_var:
            # pandas' nanvar: squared deviations from the neighbour mean, ddof = 1
            vs = K.aggregate_var(dev_graph, rows, f, ldr, pieces['mean'], rb, re,
                                 want_var='var' in aggs, want_std='std' in aggs)
            pieces['var'], pieces['std'] = vs[:f], vs[f:]
        if 'min' in aggs or 'max' in aggs:
            mm = K.aggregate_minmax(dev_graph, rows, f, ldr, rb, re,
                                    want_min='min' in aggs, want_max='max' in aggs)
            pieces['min'], pieces['max'] = mm[:f], mm[f:]
        # candidate order: every column under the first aggregation, then the second, ... (:158-162)
        if list(aggs) == ['sum', 'mean']:
            return block
        return self._as_block([pieces[a][j] for a in aggs for j in range(f)], n)

    def _aggregate_general(self, K, dev_graph, prev, aggs, names, dtypes, n, rb, re):
        """Any supported aggregation list, including 'prod', 'median', 'count' / 'size' and parents that are carried as
        int64 bits (the reference's integer columns in a run with 'prod': numpy multiplies and adds them in wrapping
        int64 arithmetic, extract.py:111).  Per parent column:
          * integer aggregations (sum, prod, min, max) of an int64 parent: wrapping integer kernels (grx_aggregate_i64);
          * everything else on the fp64 VALUES of the parent (an int64 parent converted like astype(float64), which is
            what pandas does before mean / var / median): the pairwise-order kernels, grx_aggregate_prod,
            grx_aggregate_median, grx_aggregate_count.
        A candidate whose reference dtype is int64 stays in int64 bits, any other is fp64 (an integer result cast)."""
        f = len(prev)
        cols = [self._work[c] for c in prev]
        is_i64 = [c in self._i64 for c in prev]
        fp_cols = [K.convert_i64_to_f64(col) if i64 else col for col, i64 in zip(cols, is_i64)]
        rows, ldr = K.pack_rows(fp_cols, n)
        pieces = {}
        need_var = 'var' in aggs or 'std' in aggs
        if 'sum' in aggs or 'mean' in aggs or need_var:
            block = K.aggregate(dev_graph, rows, f, ldr, rb, re, want_sum=True, want_mean=True)
            pieces['sum'], pieces['mean'] = block[:f], block[f:]
        if need_var:
            vs = K.aggregate_var(dev_graph, rows, f, ldr, pieces['mean'], rb, re, want_var=True, want_std=True)
            pieces['var'], pieces['std'] = vs[:f], vs[f:]
        if 'min' in aggs or 'max' in aggs:
            mm = K.aggregate_minmax(dev_graph, rows, f, ldr, rb, re, want_min=True, want_max=True)
            pieces['min'], pieces['max'] = mm[:f], mm[f:]
        if 'prod' in aggs:
            pieces['prod'] = K.aggregate_prod(dev_graph, rows, f, ldr, rb, re)
        if 'median' in aggs:
            pieces['median'] = K.aggregate_median(dev_graph, rows, f, ldr, rb, re)
        if 'count' in aggs or 'size' in aggs:
            pieces['count'] = pieces['size'] = K.aggregate_count(dev_graph, f, rb, re, as_i64=False)
        ints = {}
        if any(is_i64):
            # the integer parents once more, as integers: their sums and products wrap modulo 2^64 like numpy's
            idx = [j for j in range(f) if is_i64[j]]
            irows, ildr = K.pack_rows([cols[j] for j in idx], n)
            want = [a for a in ('sum', 'prod', 'min', 'max') if a in aggs]
            if want:
                got = K.aggregate_i64(dev_graph, irows, len(idx), ildr, rb, re, want=want)
                for a in want:
                    ints[a] = {j: got[a][k] for k, j in enumerate(idx)}
            if 'count' in aggs or 'size' in aggs:
                cnt = K.aggregate_count(dev_graph, 1, rb, re, as_i64=True)[0]
                ints['count'] = ints['size'] = {j: cnt for j in idx}
        picked, flags = [], []
        for k, a in enumerate(aggs):
            for j in range(f):
                keep_int = np.dtype(dtypes[k * f + j]).kind in 'iu' and is_i64[j]
                if is_i64[j] and a in ints:
                    col = ints[a][j]                              # int64 bits
                    if not keep_int:
                        col = K.convert_i64_to_f64(col)          # the agg frame became float64: cast like pandas
                else:
                    col = pieces[a][j]
                    assert not keep_int, (a, prev[j])
                picked.append(col)
                flags.append(keep_int)
        for nm, flag in zip(names, flags):
            if flag:
                self._i64.add(nm)
        return self._as_block(picked, n)

    def _aggregate_with_callables(self, K, dev_graph, prev, aggs, callables, names, dtypes, n):
        """
        ``aggs`` holds callables without a device kernel.  The kernel-backed entries run on the device as always; for
        the callables the reference's own expression is evaluated (extract.py:104-113): per node, the frame of its
        neighbours' previous-generation features -- index = neighbour labels in adjacency order, columns = feature
        names -- goes through ``DataFrame.agg([f])`` and ``fillna(0)``.  The neighbour rows are gathered ON THE
        DEVICE (grx_permute_columns over the adjacency-ordered index list) and downloaded once; pandas then works
        on slices of that table.  This is the reference's per-node pandas loop at the reference's speed; it exists
        so that no constructor input of the reference raises here, not to be fast.  Never routed through oracle/.
        A callable pandas can only apply ELEMENT-wise (np.ptp under pandas 2: the result is not one row per
        function) is refused with a TypeError instead of reproducing the reference's garbled column names.
        """
        import torch
        f = len(prev)
        host = self.graph._device_graph()[0]
        known = [a for a in aggs if a not in callables]
        pieces = {}
        if known:
            if any(c in self._i64 for c in prev):
                raise NotImplementedError("callable aggregations together with wrapping integer 'prod' columns")
            known_names = [f'{c}({a})' for a in known for c in prev]
            known_dtypes = [np.dtype('float64')] * len(known_names)
            if set(known) <= set(_FAST_AGGS):
                blk = self._aggregate_fast(K, dev_graph, prev, known, n, 0, n)
            else:
                blk = self._aggregate_general(K, dev_graph, prev, known, known_names, known_dtypes, n, 0, n)
            for k, a in enumerate(known):
                pieces[a] = [blk[k * f + j] for j in range(f)]
        # neighbour rows in adjacency order, gathered on the device: E[c, e] = X_prev[c][agg_col[e]]
        cols = [self._work[c] for c in prev]
        nnz = int(host.row_ptr[-1])
        E = K.to_host(K.permute_columns(cols, dev_graph.agg_col, nnz)) if nnz else np.zeros((f, 0))
        nbr = dev_graph.agg_col
        nbr = (K.to_host(nbr) if hasattr(nbr, 'detach') else np.asarray(nbr))[:nnz]
        labels = np.asarray(list(host.labels), dtype=object)[np.asarray(host.perm)]      # label of every internal row
        funcs = [callables[a] for a in aggs if a in callables]
        order = [a for a in aggs if a in callables]
        out = {a: np.zeros((f, n)) for a in order}
        all_int = {a: np.ones(f, dtype=bool) for a in order}
        prev_dt = [np.dtype(self._dtypes.get(c, 'float64')) for c in prev]
        row_ptr = host.row_ptr
        # A node without neighbours: the reference aggregates an EMPTY frame with the whole list (extract.py:108-113).
        # pandas either raises there (a list mixing names and callables: "cannot combine transform and aggregation
        # operations" -- the reference raises the same, so this propagates), or hands back nothing for the callables
        # (-> the node's candidates are NaN -> 0, extract.py:132), or real rows (a callable that copes with an empty Series).
        empty_rows = None
        if n and int(np.diff(row_ptr).min()) == 0:
            res0 = pd.DataFrame({c: np.zeros(0, dtype=prev_dt[j]) for j, c in enumerate(prev)}, columns=prev).agg(list(self.aggs))
            empty_rows = {a: (res0.loc[a].to_numpy(dtype=np.float64, na_value=0.0)
                              if isinstance(res0, pd.DataFrame) and a in res0.index and res0.shape[1] == f else np.zeros(f))
                          for a in order}
        # The reference hands the WHOLE list to DataFrame.agg (extract.py:111); here only the callables go through pandas
        # node by node.  Whatever pandas has to say about the full list -- names mixed with callables that transform
        # ("cannot combine transform and aggregation operations"), duplicate result names -- it says on the first node
        # with neighbours, as in the reference: one probe with the real list on that node's frame
        for v in range(n):
            b, e = int(row_ptr[v]), int(row_ptr[v + 1])
            if e > b:
                pd.DataFrame({c: E[j, b:e].astype(prev_dt[j], copy=False) for j, c in enumerate(prev)},
                             index=labels[nbr[b:e]], columns=prev).agg(list(self.aggs))
                break
        for v in range(n):
            b, e = int(row_ptr[v]), int(row_ptr[v + 1])
            if e == b:
                for a in order:
                    out[a][:, v] = empty_rows[a]
                    all_int[a][:] = False                                   # NaN -> 0.0: the column is float
                continue
            frame = pd.DataFrame({c: E[j, b:e].astype(prev_dt[j], copy=False) for j, c in enumerate(prev)},
                                 index=labels[nbr[b:e]], columns=prev)
            res = frame.agg(funcs)
            if not isinstance(res, pd.DataFrame) or list(res.index) != order:
                raise TypeError(
                    f'aggregation callables {order}: DataFrame.agg did not return one row per function (pandas applied a '
                    f'function element-wise: it does not reduce a Series) -- pass a function of a Series that returns a scalar')
            for a in order:
                vals = res.loc[a]
                for j in range(f):
                    x = vals.iloc[j]
                    if not isinstance(x, (int, np.integer)) or isinstance(x, (bool, np.bool_)):
                        all_int[a][j] = False
                out[a][:, v] = vals.to_numpy(dtype=np.float64, na_value=0.0)      # .fillna(0), extract.py:113
        for a in order:
            block = K.to_device(np.ascontiguousarray(out[a]))
            pieces[a] = [block[j] for j in range(f)]
        # dtypes as pandas infers them from the per-node dicts: a candidate stays integer only if every function of the
        # list returned integers for its parent (the agg frame's column is one dtype) -- see _candidate_dtype
        f64 = np.dtype('float64')
        base = self._candidate_dtype(np.dtype('int64'), known, self._no_empty_rows(host)) if known else np.dtype('int64')
        new_dtypes = []
        for a in aggs:
            for j in range(f):
                ints = (prev_dt[j].kind in 'iu' and base.kind in 'iu' and all(all_int[c][j] for c in order))
                new_dtypes.append(np.dtype('int64') if ints else f64)
        return self._as_block([pieces[a][j] for a in aggs for j in range(f)], n), new_dtypes

    def _update_columns(self, names: Sequence[str], cols: Sequence, dtypes: Sequence[np.dtype],
                        block=None) -> None:
        """Add candidate columns, prune across the whole working set, record what this
        generation retains (extract.py:121-142)."""
        K = self._K()
        plan = self._shard()
        n = self._n()
        names = list(names)
        for nm, col, dt in zip(names, cols, dtypes):
            self._work[nm] = col
            self._dtypes[nm] = dt
        # bin the columns that have no cached bins yet
        fresh = [nm for nm in self._work if nm not in self._work_bins]
        partial = plan is not None and block is not None and getattr(self, '_partial_block', None) is block \
            and fresh == names
        if fresh:
            if block is None or fresh != names:
                block = self._as_block([self._work[nm] for nm in fresh], n)
            typed = [nm in self._i64 for nm in fresh] if self._i64 else None
            kw = {'is_i64': typed} if typed and any(typed) else {}
            if plan is None:
                bins, _ = K.vertical_log_bin(block, **kw)
            elif partial:
                # candidate block with only this rank's rows: whole columns to their owners, the
                # owners' bins of this rank's rows back (parallel.py, steps 1 and 2)
                owned = plan.columns_to_owners(block)
                owned_bins = K.zeros((owned.shape[0], max(n, 1)), dtype=self._uint8())[:, :n]
                if owned.shape[0]:
                    own_kw = {'is_i64': typed[plan.rank::plan.world]} if kw else {}
                    K.vertical_log_bin(owned, out=owned_bins, **own_kw)
                bins = plan.owned_to_rows(owned_bins, len(fresh))
            else:
                bins = K.zeros((len(fresh), max(n, 1)), dtype=self._uint8())[:, :n]
                mine = slice(plan.rank, len(fresh), plan.world)
                if len(range(len(fresh))[mine]):
                    own_kw = {'is_i64': typed[mine]} if kw else {}
                    K.vertical_log_bin(block[mine], out=bins[mine], **own_kw)
                plan.all_reduce_max_(bins)
            for j, nm in enumerate(fresh):
                self._work_bins[nm] = bins[j]
        work_names = list(self._work)
        rb, re = (0, n) if plan is None else (plan.row_begin, plan.row_end)
        # the pruner only asks whether a distance is <= the generation number (prune.py:110-113)
        dist = K.chebyshev([self._work_bins[nm] for nm in work_names], n, 0, rb, re, cap=self._feature_group_thresh)
        if plan is not None:
            plan.all_reduce_max_(dist)
        dist_host = K.to_host(dist)
        # host: feature graph -> groups -> drop list (a few dozen nodes)
        pruner = FeaturePruner(self._final_names, self._feature_group_thresh)
        features_to_drop = pruner.prune_from_distances(work_names, dist_host)
        dropped = set(features_to_drop)
        for nm in dropped:
            del self._work[nm]
            del self._work_bins[nm]
        # extract.py:140 Index.difference: name-sorted iff the drop list is non-empty (pandas 2)
        kept = list(dict.fromkeys(nm for nm in names if nm not in dropped))
        if partial and kept:
            # step 3: the retained new columns, and only those, become complete on every rank
            # (columns are completed where they lie: no copy, no packing)
            for nm, col in zip(kept, plan.all_gather_columns([self._work[nm] for nm in kept])):
                self._work[nm] = col
        self._partial_block = None
        retained = sorted(kept) if dropped else kept
        self._final_names[self.generation_count] = retained
        for nm in retained:
            self._final_cols[nm] = self._work[nm]
        self.stats.append(dict(generation=self.generation_count, candidates=len(names),
                               working=len(work_names), dropped=len(dropped), retained=len(retained)))

    def _uint8(self):
        import torch
        return torch.uint8

    def _as_block(self, cols: Sequence, n: int):
        """[len(cols), n] contiguous fp64 block assembled by grx_gather_columns."""
        K = self._K()
        return K.gather_columns(list(cols), n)[:, :n] if n else K.zeros((len(cols), 0))

    def _finalize_features(self) -> DataFrameLike:
        """DataFrame of every recorded feature, latest generation first (extract.py:91-96): the columns are put
        back into label order on the device (one permutation kernel for the whole table), copied out as one
        block at link rate, and wrapped without another copy.  The device block is remembered next to the frame
        (features/handoff.py): RoleExtractor.extract_role_factors(X) on the unmodified table skips the upload."""
        columns = self.final_columns()
        frame = self._frame_of(columns, [self._final_cols[nm] for nm in columns], handoff=True)
        return frame

    def _frame_of(self, names: Sequence[str], cols: Sequence, handoff: bool = False) -> pd.DataFrame:
        K = self._K()
        n = self._n()
        csr = self.graph.to_csr()
        labels = csr.label_index() if hasattr(csr, 'label_index') else pd.Index(self._labels())
        if not names:
            return pd.DataFrame(index=labels)
        names = list(names)
        t0 = time.perf_counter()
        dev_block = K.permute_columns(list(cols), self._inv_device(), n)                  # [F, n], label order
        self._phase_sync()
        t1 = time.perf_counter()
        block = K.to_host(dev_block)
        t2 = time.perf_counter()
        # one frame per run of equally typed columns, each wrapping its rows of the block (integer runs -- generation 0
        # of unweighted graphs -- as int64 copies), joined without copying: assigning the integer columns one by one
        # into a single float frame cost a block split per column
        kinds = [np.dtype(self._dtypes.get(nm, 'float64')) for nm in names]
        parts, a = [], 0
        while a < len(names):
            b = a + 1
            while b < len(names) and kinds[b] == kinds[a]:
                b += 1
            if kinds[a] == np.dtype('float64'):
                values = block[a:b]
            else:
                # integer columns: fp64 values of exact integers -> astype; int64 BITS (runs with 'prod') -> a view
                values = np.stack([block[j].view(np.int64) if names[j] in self._i64 else block[j].astype(kinds[a])
                                   for j in range(a, b)])
            parts.append(pd.DataFrame(values.T, index=labels, columns=names[a:b], copy=False))
            a = b
        frame = parts[0] if len(parts) == 1 else pd.concat(parts, axis=1, copy=False)
        t3 = time.perf_counter()
        if handoff and hasattr(K, 'host_checksums') and not (self._i64 & set(names)):
            from graphrole_amd.features import handoff as _handoff
            _handoff.register(K, frame, dev_block)
        if handoff:
            self.wall.update(permute_s=t1 - t0, download_s=t2 - t1, frame_s=t3 - t2,
                             handoff_hash_s=time.perf_counter() - t3, table_bytes=int(block.nbytes))
        return frame

    def _inv_device(self):
        """inv (label row -> internal row) as an int32 device tensor (already there after a device ingest)."""
        if getattr(self, '_inv_dev', None) is None:
            order = self._order()
            dev = getattr(order, 'inv_dev', None)
            self._inv_dev = dev if dev is not None else self._K().to_device(order.inv.astype(np.int32))
        return self._inv_dev

    # ------------------------------------------------------------------ reference-compatible internals
    # The reference's tests drive these private members with DataFrames / dicts of dicts
    # (tests/test_features/test_extract.py:87-214).  They are thin views over the device store.
    @property
    def _features(self) -> pd.DataFrame:
        if not self._work:
            return pd.DataFrame()
        return self._frame_of(list(self._work), list(self._work.values()))

    @_features.setter
    def _features(self, frame: pd.DataFrame) -> None:
        self._work.clear()
        self._work_bins.clear()
        for nm, col, dt in zip(*self._columns_from_frame(frame)):
            self._work[nm] = col
            self._dtypes[nm] = dt

    @property
    def _final_features(self) -> Dict[int, DataFrameDict]:
        out: Dict[int, DataFrameDict] = {}
        for gen, names in self._final_names.items():
            out[gen] = self._frame_of(list(names), [self._final_cols[nm] for nm in names]).to_dict()
        return out

    @_final_features.setter
    def _final_features(self, value: Dict[int, DataFrameDict]) -> None:
        self._final_names = {}
        self._final_cols = {}
        for gen, table in value.items():
            frame = pd.DataFrame(table)
            names, cols, dtypes = self._columns_from_frame(frame)
            self._final_names[gen] = list(names)
            for nm, col, dt in zip(names, cols, dtypes):
                # share the working-set column when it is the same feature
                self._final_cols[nm] = self._work.get(nm, col)
                self._dtypes.setdefault(nm, dt)

    def _columns_from_frame(self, frame: pd.DataFrame):
        K = self._K()
        order = self._order()
        frame = frame.reindex(self._labels()).fillna(0)     # nodes missing from the frame -> 0 (:132)
        names = [str(c) if not isinstance(c, str) else c for c in frame.columns]
        cols = [K.to_device(order.to_internal(frame[c].to_numpy(dtype=np.float64))) for c in frame.columns]
        dtypes = [frame[c].dtype if frame[c].dtype.kind in 'iu' else np.dtype('float64') for c in frame.columns]
        return names, cols, dtypes

    def _get_next_features(self) -> DataFrameLike:
        """Next level of recursive features as a DataFrame (extract.py:98-119)."""
        self._shard()
        names, cols, _, _ = self._next_feature_columns(complete=True)
        K = self._K()
        order = self._order()
        data = {nm: order.to_label_order(K.to_host(c)) for nm, c in zip(names, cols)}
        return pd.DataFrame(data, index=pd.Index(self._labels()), columns=names)

    def _update(self, features: DataFrameLike) -> None:
        """Add candidate features given as a DataFrame and prune (extract.py:121-142)."""
        self._shard()
        names, cols, dtypes = self._columns_from_frame(as_frame(features))
        self._update_columns(names, cols, dtypes)

    @staticmethod
    def _aggregated_df_to_dict(agg_df: DataFrameLike) -> Dict[str, float]:
        """{'<feature>(<agg>)': value} from the result of DataFrame.agg (extract.py:144-163);
        kept for API compatibility -- the device path names its columns directly."""
        if isinstance(agg_df, pd.Series):
            agg_df = agg_df.to_frame().T
        flat: Dict[str, float] = {}
        for agg, row in agg_df.to_dict(orient='index').items():
            for feature, value in row.items():
                flat[f'{feature}({agg})'] = value
        return flat


# Helper functions

def as_frame(df_like: DataFrameLike) -> pd.DataFrame:
    """pd.Series -> one-column DataFrame; DataFrames pass through (extract.py:168-177)."""
    return df_like.to_frame() if isinstance(df_like, pd.Series) else df_like
