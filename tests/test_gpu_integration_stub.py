"""
-m gpu: INTEGRATION.md section 2 executed as written -- a numpy-only host binds libgrx.so with ctypes and its own
allocator helpers (grx_dev_malloc / grx_memcpy_* / grx_stream_sync); NO torch in the process.  The DeviceArray
stub is taken verbatim from the document's first code block; the calls of 2a-2e follow the document on a small
graph and are checked against the oracle.
"""
import os
import re
import subprocess
import sys
import textwrap

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

DRIVER = r'''
import ctypes as C, json, re, sys
import numpy as np
sys.path.insert(0, ROOT)
assert 'torch' not in sys.modules
doc = open(ROOT + '/INTEGRATION.md').read()
stub = re.search(r"```python\n(import ctypes as C, numpy as np.*?)```", doc, re.S).group(1)
stub = stub.replace("C.CDLL('libgrx.so')", "C.CDLL(ROOT + '/graphrole_amd/libgrx.so')")
exec(stub)                                             # _lib, _vp, _i64, _i32, _f64, _check, DeviceArray
assert 'torch' not in sys.modules
_lib.grx_last_error.restype = C.c_char_p
_lib.grx_log_bin_workspace_bytes.restype = C.c_size_t
_lib.grx_nmf_fit_workspace_bytes.restype = C.c_size_t

from graphrole_amd import synth                         # numpy-only graph generator
from oracle import refex, rolx
G = synth.ba_graph(3000, 5, seed=2)
n = G.n
og = refex.OracleGraph(labels=G.labels, row_ptr=G.row_ptr, col=G.col, w=None, directed=False,
                       num_edges=G.num_edges, adj_col=G.adj_col)
names0, X0 = refex.neighborhood_features(og, fast=True)

# ---- 2c: generation-0 degree (weighted row sums, self-loops twice)
d_row_ptr, d_col, d_adj = DeviceArray(G.row_ptr), DeviceArray(G.col), DeviceArray(G.adj_col)
deg = DeviceArray(nbytes=n * 8)
_check(_lib.grx_row_sums(_i64(n), d_row_ptr.ptr, d_col.ptr, None, _i32(1), _i64(0), _i64(n), deg.ptr, None))
assert np.array_equal(deg.to_host(np.float64, (n,)), X0[:, 0])
internal, external = DeviceArray(np.zeros(n)), DeviceArray(np.zeros(n))
_lib.grx_egonet_workspace_bytes.restype = C.c_size_t
nnz = int(len(G.col))
ego_bytes = _lib.grx_egonet_workspace_bytes(_i64(n), _i64(nnz))
ego_ws = DeviceArray(nbytes=ego_bytes)
_check(_lib.grx_egonet_features(_i64(n), _i64(nnz), d_row_ptr.ptr, d_col.ptr, None, None, _i32(0), _i64(0), _i64(n),
                                internal.ptr, external.ptr, ego_ws.ptr, C.c_size_t(ego_bytes), None))
assert np.array_equal(internal.to_host(np.float64, (n,)), X0[:, 1])
assert np.array_equal(external.to_host(np.float64, (n,)), X0[:, 2])

# ---- 2a: neighbour aggregation of the three generation-0 columns
plan = _vp()
_check(_lib.grx_aggregate_plan_create(_i64(n), G.row_ptr.ctypes.data_as(_vp), C.byref(plan)))
f = 3
ldr = _lib.grx_aggregate_ldr(f)
cols = [DeviceArray(np.ascontiguousarray(X0[:, j])) for j in range(f)]
ptrs = (_vp * f)(*[c.ptr.value for c in cols])
rows = DeviceArray(nbytes=n * ldr * 8)
_check(_lib.grx_pack_rows(_i64(n), _i32(f), ptrs, rows.ptr, _i32(ldr), None))
out = DeviceArray(nbytes=2 * f * n * 8)
mean_ptr = _vp(out.ptr.value + f * n * 8)
_check(_lib.grx_aggregate(plan, d_row_ptr.ptr, d_adj.ptr, _i32(f), rows.ptr, _i32(ldr), _i64(0), _i64(n), out.ptr,
                          mean_ptr, _i64(n), None))
block = out.to_host(np.float64, (2 * f, n))
S, M = refex.aggregate_fast(og, X0)
assert np.array_equal(block[:f].T, S) and np.array_equal(block[f:].T, M)          # bit-exact

# ---- 2b: binning + Chebyshev of the nine columns
feats = np.ascontiguousarray(np.vstack([X0.T, block]))                              # [F, n]
F = feats.shape[0]
X = DeviceArray(feats)
bins = DeviceArray(nbytes=F * n)
ws_bytes = _lib.grx_log_bin_workspace_bytes(_i64(n), _i32(F)); ws = DeviceArray(nbytes=ws_bytes)
_check(_lib.grx_vertical_log_bin(_i64(n), _i32(F), X.ptr, _i64(n), _f64(0.5), bins.ptr, _i64(n), None,
                                 ws.ptr, C.c_size_t(ws_bytes), None))
got_bins = bins.to_host(np.uint8, (F, n))
exp_bins = np.stack([refex.vertical_log_binning(feats[j]) for j in range(F)])
assert np.array_equal(got_bins, exp_bins)
bptrs = (_vp * F)(*[bins.ptr.value + j * n for j in range(F)])
dist = DeviceArray(np.zeros((F, F), dtype=np.int32))
_check(_lib.grx_chebyshev(_i64(0), _i64(n), _i32(F), _i32(0), bptrs, dist.ptr, _i32(255), None))
assert np.array_equal(dist.to_host(np.int32, (F, F)), refex.chebyshev_matrix(exp_bins.T))

# ---- 2e: the whole factorisation in one call
r = 4
Xn = np.abs(feats.T) + 0.0                                                           # n x F, non-negative
omega = np.random.RandomState(0).normal(size=(F, r + 10))
dX = DeviceArray(np.ascontiguousarray(Xn.T))
dW, dH = DeviceArray(nbytes=r * n * 8), DeviceArray(nbytes=r * F * 8)
class Info(C.Structure):
    _fields_ = [('n_iter', C.c_int), ('direct_residuals', C.c_int), ('err_init', C.c_double),
                ('err_last', C.c_double), ('x_sq_norm', C.c_double)]
info = Info()
fit_bytes = _lib.grx_nmf_fit_workspace_bytes(_i64(n), _i32(F), _i32(r)); fit_ws = DeviceArray(nbytes=fit_bytes)
_check(_lib.grx_nmf_fit(_i64(n), _i32(F), _i32(r), dX.ptr, _i64(n), omega.ctypes.data_as(_vp), _i32(r + 10),
                        _f64(1e-4), _i32(200), dW.ptr, _i64(n), dH.ptr, C.byref(info), None, None, fit_ws.ptr,
                        C.c_size_t(fit_bytes), None))
We, He, it = rolx.nmf(Xn, r, omega)
W = dW.to_host(np.float64, (r, n)).T
H = dH.to_host(np.float64, (r, F))
assert info.n_iter == it, (info.n_iter, it)
assert np.abs(W - We).max() / np.abs(We).max() < 1e-8 and np.abs(H - He).max() / np.abs(He).max() < 1e-8
# ---- 2a (packed rows): the same sums / means from 8-byte bit-packed integer rows
class Layout(C.Structure):
    _fields_ = [('n_fields', C.c_int), ('field_bits', C.c_int * 8), ('degree_bits', C.c_int), ('n_out', C.c_int),
                ('out_field', C.c_int * 8), ('out_is_mean', C.c_int * 8)]
d_block = DeviceArray(np.ascontiguousarray(X0.T))
d_bits = DeviceArray(np.zeros(f, dtype=np.int32))
_check(_lib.grx_column_bits(_i64(n), _i32(f), d_block.ptr, _i64(n), _i64(0), _i64(n), C.c_uint64(7), d_bits.ptr, None))
widths = d_bits.to_host(np.int32, (f,))
assert list(widths) == [max(int(X0[:, j].max()).bit_length(), 1) for j in range(f)]
lay = Layout()
lay.n_fields, lay.degree_bits, lay.n_out = f, 0, f
for j in range(f):
    lay.field_bits[j], lay.out_field[j], lay.out_is_mean[j] = int(widths[j]), j, 0
row_bytes = _lib.grx_packed_row_bytes(C.byref(lay))
assert row_bytes in (8, 16)
prow = DeviceArray(nbytes=n * row_bytes)
_check(_lib.grx_pack_fields(_i64(n), C.byref(lay), ptrs, d_row_ptr.ptr, prow.ptr, None))
out2 = DeviceArray(nbytes=2 * f * n * 8)
_check(_lib.grx_aggregate_packed(plan, d_row_ptr.ptr, d_adj.ptr, C.byref(lay), prow.ptr, _i64(0), _i64(n), out2.ptr,
                                 _vp(out2.ptr.value + f * n * 8), _i64(n), None))
assert np.array_equal(out2.to_host(np.float64, (2 * f, n)), block)                    # bit-identical to 2a

# ---- 2h: roles / role_percentage of a quantised factor
levels = np.sort(np.random.RandomState(1).gamma(0.7, 2.0, 8))
Gq = np.ascontiguousarray(levels[np.random.RandomState(2).randint(0, 8, size=(n, r))])
dG = DeviceArray(Gq)
first = DeviceArray(nbytes=4 * n)
_check(_lib.grx_role_argmax(_i64(n), _i32(r), dG.ptr, first.ptr, None))
share = DeviceArray(nbytes=8 * n * r)
_check(_lib.grx_row_normalise(_i64(n), _i32(r), dG.ptr, share.ptr, None))
assert np.array_equal(first.to_host(np.int32, (n,)), rolx.dominant_role_index(Gq))
assert np.array_equal(share.to_host(np.float64, (n, r)), rolx.role_percentage(Gq))
assert 'torch' not in sys.modules
print('INTEGRATION_STUB_OK', info.n_iter)
'''


def test_integration_md_stub_runs_without_torch():
    code = 'ROOT = %r\n' % ROOT + textwrap.dedent(DRIVER)
    env = dict(os.environ)
    env.setdefault('LD_LIBRARY_PATH', '/opt/rocm/lib')
    res = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=600, env=env)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    assert 'INTEGRATION_STUB_OK' in res.stdout
