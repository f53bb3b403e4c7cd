"""
ctypes binding of libgrx.so (the C ABI declared in include/grx.h).

The product path has NO CPU fallback: if the shared library is missing or a call fails, an
exception is raised.  PyTorch is imported first so that libgrx.so resolves ``libamdhip64.so.7``
to the HIP runtime PyTorch already loaded (one runtime per process); PyTorch itself is only
plumbing here (device memory, streams, torch.distributed).
"""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, c_char_p, c_double, c_float, c_int, c_int64, c_size_t, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
# GRX_LIB_PATH: load another build of the same ABI (tuning A/B runs of tools/; never set in production)
LIB_PATH = os.environ.get('GRX_LIB_PATH') or os.path.join(_HERE, 'libgrx.so')


class GrxError(RuntimeError):
    """A libgrx.so call returned a non-zero status."""


class GrxInvalid(GrxError, ValueError):
    """GRX_ERR_INVALID: the reference raises ValueError for the same condition."""


class GrxDegenerate(GrxError, ValueError):
    """GRX_ERR_DEGENERATE: numerically degenerate input (an all-zero feature matrix)."""


class NmfInfo(ctypes.Structure):
    """grx_nmf_info of include/grx.h."""
    _fields_ = [('n_iter', c_int), ('direct_residuals', c_int), ('err_init', c_double), ('err_last', c_double),
                ('x_sq_norm', c_double)]


class RefexColumn(ctypes.Structure):
    """grx_refex_column of include/grx.h."""
    _fields_ = [('generation', c_int), ('parent', c_int), ('agg', c_int), ('gen0_index', c_int),
                ('work_position', c_int), ('d_col', c_void_p)]


class RefexGeneration(ctypes.Structure):
    """grx_refex_generation of include/grx.h."""
    _fields_ = [('candidates', c_int), ('working', c_int), ('dropped', c_int), ('retained', c_int),
                ('gather_row_bytes', c_int)]


AGG_IDS = {'sum': 0, 'mean': 1, 'min': 2, 'max': 3, 'var': 4, 'std': 5, 'prod': 6, 'median': 7, 'count': 8,
           'size': 9}                                                           # grx_agg


class PackedLayout(ctypes.Structure):
    """grx_packed_layout of include/grx.h."""
    _fields_ = [('n_fields', c_int), ('field_bits', c_int * 8), ('degree_bits', c_int), ('n_out', c_int),
                ('out_field', c_int * 8), ('out_is_mean', c_int * 8)]


class P2pOp(ctypes.Structure):
    """grx_p2p_op of include/grx.h."""
    _fields_ = [('is_recv', c_int), ('peer', c_int), ('d_ptr', c_void_p), ('bytes', c_size_t)]


# transport callbacks of grx_comm_create_callbacks
ALL_REDUCE_FN = ctypes.CFUNCTYPE(c_int, c_void_p, c_void_p, c_size_t, c_int, c_int, c_void_p)
EXCHANGE_FN = ctypes.CFUNCTYPE(c_int, c_void_p, c_int, POINTER(P2pOp), c_void_p)
GROW_FN = ctypes.CFUNCTYPE(c_void_p, c_size_t, c_void_p)          # grx_grow_fn: more arena for grx_refex_run
COMM_ID_BYTES = 128
COMM_SELF_VIA_TRANSPORT = 1
DTYPE_IDS = {'float64': 0, 'int32': 1, 'int64': 2, 'uint8': 3}                 # grx_dtype
COMM_KINDS = ('all_reduce', 'exchange', 'all_gather_rows', 'columns_to_owners', 'owned_to_rows')   # grx_comm_kind

_lib = None

# name -> (restype, argtypes); every entry mirrors include/grx.h
_SIGNATURES = {
    'grx_version': (c_int, []),
    'grx_last_error': (c_char_p, []),
    'grx_device_info': (c_int, [POINTER(c_int), POINTER(c_int), c_char_p, c_size_t]),
    'grx_dev_malloc': (c_int, [POINTER(c_void_p), c_size_t]),
    'grx_dev_free': (c_int, [c_void_p]),
    'grx_memcpy_h2d': (c_int, [c_void_p, c_void_p, c_size_t, c_void_p]),
    'grx_memcpy_d2h': (c_int, [c_void_p, c_void_p, c_size_t, c_void_p]),
    'grx_memset': (c_int, [c_void_p, c_int, c_size_t, c_void_p]),
    'grx_stream_sync': (c_int, [c_void_p]),
    'grx_download': (c_int, [c_void_p, c_void_p, c_size_t, c_void_p]),
    'grx_upload': (c_int, [c_void_p, c_void_p, c_size_t, c_void_p]),
    'grx_upload_i64_as_i32': (c_int, [c_void_p, c_void_p, c_size_t, c_void_p]),
    'grx_host_checksums': (c_int, [c_void_p, c_int, c_size_t, c_size_t, c_void_p]),
    'grx_host_uniform_choice': (c_int, [c_int64, c_double, POINTER(c_int64)]),
    'grx_min_value_workspace_bytes': (c_size_t, []),
    'grx_min_value': (c_int, [c_int64, c_int, c_void_p, c_int64, c_void_p, c_void_p, c_size_t, c_void_p]),
    'grx_event_create': (c_int, [POINTER(c_void_p)]),
    'grx_event_destroy': (c_int, [c_void_p]),
    'grx_event_record': (c_int, [c_void_p, c_void_p]),
    'grx_event_elapsed_ms': (c_int, [c_void_p, c_void_p, POINTER(c_float)]),
    'grx_trace_marker': (c_int, [c_int, c_void_p]),
    'grx_profile_enable': (c_int, [c_int]),
    'grx_profile_enabled': (c_int, []),
    'grx_profile_select': (c_int, [ctypes.c_uint64]),
    'grx_profile_reset': (c_int, []),
    'grx_profile_kernel_count': (c_int, []),
    'grx_profile_kernel_name': (c_char_p, [c_int]),
    'grx_profile_read': (c_int, [c_int, POINTER(c_double), POINTER(ctypes.c_longlong)]),
    'grx_comm_rccl_unique_id': (c_int, [c_void_p]),
    'grx_comm_create_rccl': (c_int, [c_void_p, c_int, c_int, c_int, POINTER(c_void_p)]),
    'grx_comm_create_callbacks': (c_int, [c_int, c_int, ALL_REDUCE_FN, EXCHANGE_FN, c_void_p, POINTER(c_void_p)]),
    'grx_comm_destroy': (c_int, [c_void_p]),
    'grx_comm_rank': (c_int, [c_void_p]),
    'grx_comm_world': (c_int, [c_void_p]),
    'grx_comm_all_reduce': (c_int, [c_void_p, c_void_p, c_size_t, c_int, c_int, c_void_p]),
    'grx_comm_exchange': (c_int, [c_void_p, c_int, c_void_p, c_void_p]),
    'grx_comm_all_gather_rows': (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p]),
    'grx_comm_columns_to_owners': (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int64, c_int, c_void_p, c_int64,
                                           c_void_p]),
    'grx_comm_owned_to_rows': (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int64, c_int, c_void_p, c_int64,
                                       c_void_p]),
    'grx_comm_timing': (c_int, [c_void_p, c_int]),
    'grx_comm_timing_read': (c_int, [c_void_p, c_int, POINTER(ctypes.c_longlong), POINTER(c_double)]),
    'grx_comm_timing_reset': (c_int, [c_void_p]),
    'grx_row_sums': (c_int, [c_int64, c_void_p, c_void_p, c_void_p, c_int, c_int64, c_int64, c_void_p, c_void_p]),
    'grx_add_columns': (c_int, [c_int64, c_void_p, c_void_p, c_void_p, c_void_p]),
    'grx_egonet_workspace_bytes': (c_size_t, [c_int64, c_int64]),
    'grx_egonet_features': (c_int, [c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int64, c_int64,
                                    c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    'grx_pack_rows': (c_int, [c_int64, c_int, c_void_p, c_void_p, c_int, c_void_p]),
    'grx_aggregate_ldr': (c_int, [c_int]),
    'grx_aggregate_plan_create': (c_int, [c_int64, c_void_p, c_void_p]),
    'grx_aggregate_plan_destroy': (None, [c_void_p]),
    'grx_aggregate_plan_info': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p]),
    'grx_aggregate_plan_set_lanes': (c_int, [c_void_p, c_int]),
    'grx_aggregate': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int, c_int64, c_int64, c_void_p,
                              c_void_p, c_int64, c_void_p]),
    'grx_aggregate_var': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int, c_int64, c_int64, c_void_p,
                                  c_void_p, c_void_p, c_int64, c_void_p]),
    'grx_aggregate_minmax': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int, c_int64, c_int64,
                                     c_void_p, c_void_p, c_int64, c_void_p]),
    'grx_aggregate_ldi': (c_int, [c_int]),
    'grx_aggregate_i32_ok': (c_int, [c_void_p, c_int]),
    'grx_pack_rows_i32': (c_int, [c_int64, c_int, c_void_p, c_void_p, c_int, c_void_p]),
    'grx_aggregate_i32': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int, c_int64, c_int64, c_void_p,
                                  c_void_p, c_int64, c_void_p]),
    'grx_convert_i64_to_f64': (c_int, [c_int64, c_void_p, c_void_p, c_void_p]),
    'grx_convert_f64_to_i64': (c_int, [c_int64, c_void_p, c_void_p, c_void_p]),
    'grx_aggregate_i64': (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int, c_int64, c_int64, c_void_p, c_void_p,
                                  c_void_p, c_void_p, c_int64, c_void_p]),
    'grx_aggregate_count': (c_int, [c_void_p, c_int, c_int64, c_int64, c_int, c_void_p, c_int64, c_void_p]),
    'grx_aggregate_median_workspace_bytes': (c_size_t, [c_int64]),
    'grx_aggregate_median': (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int, c_int64, c_int64, c_void_p, c_int64,
                                     c_void_p, c_size_t, c_void_p]),
    'grx_aggregate_prod': (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int, c_int64, c_int64, c_void_p, c_int64,
                                   c_void_p]),
    'grx_triangle_counts': (c_int, [c_int64, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_void_p]),
    'grx_egonet_unweighted': (c_int, [c_int64, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_void_p,
                                      c_void_p, c_void_p, c_int64, c_int64, c_void_p]),
    'grx_log_bin_workspace_bytes': (c_size_t, [c_int64, c_int]),
    'grx_vertical_log_bin': (c_int, [c_int64, c_int, c_void_p, c_int64, c_double, c_void_p, c_int64, c_void_p,
                                     c_void_p, c_size_t, c_void_p]),
    'grx_vertical_log_bin_typed': (c_int, [c_int64, c_int, c_void_p, c_int64, c_void_p, c_double, c_void_p, c_int64, c_void_p,
                                           c_void_p, c_size_t, c_void_p]),
    'grx_sort_workspace_bytes': (c_size_t, [c_int64, c_int]),
    'grx_sort_columns': (c_int, [c_int64, c_int, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_size_t, c_void_p]),
    'grx_chebyshev': (c_int, [c_int64, c_int64, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p]),
    'grx_gather_columns': (c_int, [c_int64, c_int, c_void_p, c_void_p, c_int64, c_void_p]),
    'grx_host_whiten': (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    'grx_host_whiten_for_rank': (c_int, [c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    'grx_host_range_finder': (c_int, [c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int,
                                      c_int, c_void_p, c_void_p, c_void_p]),
    'grx_host_small_svd': (c_int, [c_int, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    'grx_host_nndsvd_plan': (c_int, [c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    'grx_gram_workspace_bytes': (c_size_t, [c_int64, c_int]),
    'grx_gram': (c_int, [c_int64, c_int, c_void_p, c_int64, c_int64, c_int64, c_void_p, c_int, c_void_p, c_void_p,
                         c_size_t, c_void_p]),
    'grx_project_workspace_bytes': (c_size_t, [c_int64, c_int]),
    'grx_project': (c_int, [c_int64, c_int, c_void_p, c_int64, c_int64, c_int64, c_void_p, c_int, c_void_p, c_int64,
                            c_void_p, c_void_p, c_size_t, c_void_p]),
    'grx_nndsvd_apply': (c_int, [c_int64, c_int, c_void_p, c_int64, c_int64, c_int64, c_void_p, c_void_p, c_double,
                                 c_double, c_void_p]),
    'grx_nmf_workspace_bytes': (c_size_t, [c_int64, c_int, c_int]),
    'grx_nmf_w_pass': (c_int, [c_int64, c_int, c_int, c_void_p, c_int64, c_void_p, c_int64, c_int64, c_int64,
                               c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    'grx_nmf_w_pass_next': (c_int, [c_int64, c_int, c_int, c_void_p, c_int64, c_void_p, c_int64, c_int64, c_int64,
                                    c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    'grx_nmf_h_update': (c_int, [c_int, c_int, c_void_p, c_void_p, c_void_p]),
    'grx_nmf_residual': (c_int, [c_int64, c_int, c_int, c_void_p, c_int64, c_void_p, c_int64, c_int64, c_int64,
                                 c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    'grx_lloyd_max_workspace_bytes': (c_size_t, [c_int64]),
    'grx_lloyd_max': (c_int, [c_int64, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t,
                              c_void_p]),
    'grx_nmf_kl_cost': (c_int, [c_int64, c_int, c_int, c_void_p, c_int64, c_void_p, c_int64, c_int64, c_int64,
                                c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    'grx_host_prune': (c_int, [c_int, c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p]),
    'grx_refex_bin_batch': (c_int, [c_int64, c_int]),
    'grx_refex_run': (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int,
                              c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p, c_void_p, c_int, c_void_p, POINTER(c_int),
                              c_int, c_void_p, POINTER(c_int), POINTER(c_size_t), c_void_p]),
    'grx_kmeans1d_workspace_bytes': (c_size_t, [c_int64, c_int]),
    'grx_kmeans1d': (c_int, [c_int64, c_void_p, c_int, c_int64, c_void_p, c_int, c_int, c_double, c_void_p, c_void_p,
                             c_void_p, c_void_p, c_size_t, c_void_p]),
    'grx_transpose': (c_int, [c_int64, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_void_p]),
    'grx_ingest_workspace_bytes': (c_size_t, [c_int64, c_int64, c_int]),
    'grx_ingest': (c_int, [c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_int, c_int64, c_void_p, c_void_p, c_void_p,
                           c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    'grx_permute_columns': (c_int, [c_int64, c_int, c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    'grx_orient_workspace_bytes': (c_size_t, [c_int64]),
    'grx_orient_count': (c_int, [c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    'grx_orient_fill': (c_int, [c_int64, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_size_t,
                                c_void_p]),
    'grx_host_eigh': (c_int, [c_int, c_void_p, c_void_p, c_void_p]),
    'grx_nmf_fit_workspace_bytes': (c_size_t, [c_int64, c_int, c_int]),
    'grx_nmf_init': (c_int, [c_int64, c_int, c_int, c_void_p, c_int64, c_void_p, c_int, c_void_p, c_int64, c_void_p,
                             POINTER(c_double), c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    'grx_nmf_mu': (c_int, [c_int64, c_int, c_int, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_double, c_double,
                           c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    'grx_nmf_fit': (c_int, [c_int64, c_int, c_int, c_void_p, c_int64, c_void_p, c_int, c_double, c_int, c_void_p,
                            c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    'grx_nmf_iterate_rows': (c_int, [c_int64, c_int, c_int, c_void_p, c_int64, c_void_p, c_int64, c_int64, c_int64,
                                     c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_size_t, c_void_p]),
    'grx_packed_row_bytes': (c_int, [c_void_p]),
    'grx_column_bits': (c_int, [c_int64, c_int, c_void_p, c_int64, c_int64, c_int64, ctypes.c_uint64, c_void_p, c_void_p]),
    'grx_pack_fields': (c_int, [c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    'grx_aggregate_packed': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_void_p,
                                     c_int64, c_void_p]),
    'grx_role_argmax': (c_int, [c_int64, c_int, c_void_p, c_void_p, c_void_p]),
    'grx_row_normalise': (c_int, [c_int64, c_int, c_void_p, c_void_p, c_void_p]),
    'grx_nmf_iterate': (c_int, [c_int64, c_int, c_int, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p,
                                c_void_p, c_int, c_void_p, c_size_t, c_void_p]),
}

EXPORTED_SYMBOLS = tuple(_SIGNATURES)


def load() -> ctypes.CDLL:
    """Load libgrx.so once; raise (never fall back) when it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise GrxError(
            f'{LIB_PATH} is missing: build it with `python -c "import __graft_entry__ as g; g.build()"` '
            f'or `make -C graphrole_amd/csrc`.  graphrole_amd has no CPU fallback.')
    import torch  # noqa: F401  -- loads the HIP runtime libgrx.so links against
    lib = ctypes.CDLL(LIB_PATH)
    for name, (restype, argtypes) in _SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the header and the .so disagree
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = lib
    return lib


def check(status: int, what: str = '') -> None:
    if status == 0:
        return
    msg = load().grx_last_error().decode('utf-8', 'replace')
    text = f'{what}: {msg}' if what else msg
    if status == -1:
        raise GrxInvalid(text)
    if status == -5:
        raise GrxDegenerate(text)
    raise GrxError(f'[status {status}] {text}')


def call(name: str, *args):
    """Invoke an int-returning entry point and raise on a non-zero status."""
    fn = getattr(load(), name)
    check(fn(*args), name)
