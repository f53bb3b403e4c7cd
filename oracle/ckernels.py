"""
oracle/ckernels.py -- TEST INFRASTRUCTURE ONLY.  ctypes loader for oracle/liboracle.so
(plain-C twins of the heavy loops in oracle/refex.py; built by ``make -C oracle``).
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, 'liboracle.so')
_lib = None

_i64p = np.ctypeslib.ndpointer(dtype=np.int64, flags='C_CONTIGUOUS')
_i32p = np.ctypeslib.ndpointer(dtype=np.int32, flags='C_CONTIGUOUS')
_f64p = np.ctypeslib.ndpointer(dtype=np.float64, flags='C_CONTIGUOUS')


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, 'csrc', 'oracle_kernels.c')
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(['make', '-C', _HERE, '-B', 'liboracle.so'], stdout=subprocess.DEVNULL)
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_SO)
        L.orc_aggregate.argtypes = [ctypes.c_int64, _i64p, _i32p, ctypes.c_int, _f64p, _f64p, _f64p]
        L.orc_aggregate.restype = None
        L.orc_aggregate_var.argtypes = [ctypes.c_int64, _i64p, _i32p, ctypes.c_int, _f64p, _f64p, _f64p]
        L.orc_aggregate_var.restype = None
        L.orc_aggregate_prod.argtypes = [ctypes.c_int64, _i64p, _i32p, ctypes.c_int, _f64p, _f64p]
        L.orc_aggregate_prod.restype = None
        L.orc_aggregate_minmax.argtypes = [ctypes.c_int64, _i64p, _i32p, ctypes.c_int, _f64p, _f64p, _f64p]
        L.orc_aggregate_minmax.restype = None
        L.orc_rowsum.argtypes = [ctypes.c_int64, _i64p, _i32p, ctypes.c_void_p, ctypes.c_int, _f64p]
        L.orc_rowsum.restype = None
        L.orc_egonet.argtypes = [ctypes.c_int64, _i64p, _i32p, ctypes.c_void_p, ctypes.c_int, _f64p, _f64p]
        L.orc_egonet.restype = ctypes.c_int
        L.orc_vertical_log_binning.argtypes = [ctypes.c_int64, _f64p, ctypes.c_double, _i32p]
        L.orc_vertical_log_binning.restype = ctypes.c_int64
        L.orc_chebyshev.argtypes = [ctypes.c_int64, ctypes.c_int, _i32p, _i64p]
        L.orc_chebyshev.restype = None
        L.orc_set_threads.argtypes = [ctypes.c_int]
        L.orc_set_threads.restype = None
        _lib = L
    return _lib


def set_threads(t: int) -> None:
    """Threads of the row-parallel C loops (0 = up to 32): any count gives the same values; the CPU-baseline leg of
    bench.py times with 1."""
    lib().orc_set_threads(int(t))


def _wptr(w):
    return None if w is None else w.ctypes.data_as(ctypes.c_void_p)


def aggregate(row_ptr, col, X):
    X = np.ascontiguousarray(X, dtype=np.float64)
    n, f = X.shape
    S = np.empty_like(X)
    M = np.empty_like(X)
    lib().orc_aggregate(n, row_ptr, col, f, X, S, M)
    return S, M


def aggregate_var(row_ptr, col, X):
    X = np.ascontiguousarray(X, dtype=np.float64)
    n, f = X.shape
    var = np.empty_like(X)
    std = np.empty_like(X)
    lib().orc_aggregate_var(n, row_ptr, col, f, X, var, std)
    return var, std


def aggregate_prod(row_ptr, col, X):
    X = np.ascontiguousarray(X, dtype=np.float64)
    n, f = X.shape
    out = np.empty_like(X)
    lib().orc_aggregate_prod(n, row_ptr, col, f, X, out)
    return out


def aggregate_minmax(row_ptr, col, X):
    X = np.ascontiguousarray(X, dtype=np.float64)
    n, f = X.shape
    lo = np.empty_like(X)
    hi = np.empty_like(X)
    lib().orc_aggregate_minmax(n, row_ptr, col, f, X, lo, hi)
    return lo, hi


def rowsum(row_ptr, col, w, add_self_loop: bool):
    n = len(row_ptr) - 1
    out = np.empty(n)
    lib().orc_rowsum(n, row_ptr, col, _wptr(w), int(add_self_loop), out)
    return out


def egonet(row_ptr, col, w, directed: bool):
    n = len(row_ptr) - 1
    internal = np.empty(n)
    external = np.empty(n)
    rc = lib().orc_egonet(n, row_ptr, col, _wptr(w), int(directed), internal, external)
    if rc != 0:
        raise MemoryError('orc_egonet')
    return internal, external


def vertical_log_binning(arr, frac: float = 0.5):
    arr = np.ascontiguousarray(arr, dtype=np.float64)
    out = np.empty(len(arr), dtype=np.int32)
    nb = lib().orc_vertical_log_binning(len(arr), arr, frac, out)
    if nb < 0:
        raise ValueError('must specify frac in interval (0, 1)')
    return out


def chebyshev(B_colmajor):
    """B_colmajor: int32 [F, n] (one binned column per row of the array)."""
    B = np.ascontiguousarray(B_colmajor, dtype=np.int32)
    F, n = B.shape
    D = np.zeros((F, F), dtype=np.int64)
    lib().orc_chebyshev(n, F, B, D)
    return D
