"""
CPU test double of graphrole_amd.kernels (same function signatures, torch CPU tensors), built on
oracle/.  Used ONLY by the `-m "not gpu"` tests to exercise the host logic (drop-in classes,
pruning decisions, naming, sharding over gloo) in a container without a GPU.  Never shipped,
never measured.
"""


import numpy as np
import torch

from oracle import ckernels, rolx


def device():
    return torch.device('cpu')


def to_device(a):
    return torch.from_numpy(np.ascontiguousarray(a).copy())


def to_host(t):
    return t.detach().cpu().numpy()


def empty(shape, dtype=torch.float64):
    return torch.empty(shape, dtype=dtype)


def zeros(shape, dtype=torch.float64):
    return torch.zeros(shape, dtype=dtype)


class DeviceCSR:
    def __init__(self, row_ptr, col, w=None, agg_col=None):
        self.n = len(row_ptr) - 1
        self.nnz = int(row_ptr[-1])
        self.row_ptr = np.ascontiguousarray(row_ptr, dtype=np.int64)
        self.col = np.ascontiguousarray(col, dtype=np.int32)
        self.w = None if w is None else np.ascontiguousarray(w, dtype=np.float64)
        self.lanes_per_row = 8
        self.agg_col = self.col if agg_col is None else np.ascontiguousarray(agg_col, dtype=np.int32)


def row_sums(csr, add_self_loop, row_begin=0, row_end=None, out=None):
    row_end = csr.n if row_end is None else row_end
    full = ckernels.rowsum(csr.row_ptr, csr.col, csr.w, add_self_loop)
    res = torch.zeros(csr.n, dtype=torch.float64) if out is None else out
    res[row_begin:row_end] = torch.from_numpy(full[row_begin:row_end])
    return res


def add_columns(a, b):
    return a + b


def egonet_features(csr, directed, rowsum=None, row_begin=0, row_end=None, shard=None):
    row_end = csr.n if row_end is None else row_end
    i, e = ckernels.egonet(csr.row_ptr, csr.col, csr.w, directed)
    if shard is not None and csr.w is None and not directed:
        # same collective as the device path: per-rank partial integer counts, summed once
        part = torch.zeros(csr.n, dtype=torch.int64)
        mine = slice(shard.rank, csr.n, shard.world)
        part[mine] = torch.from_numpy(np.rint(i).astype(np.int64))[mine]
        shard.all_reduce_sum_(part)
        assert np.array_equal(part.numpy(), np.rint(i).astype(np.int64))
    internal = torch.zeros(csr.n, dtype=torch.float64)
    external = torch.zeros(csr.n, dtype=torch.float64)
    internal[row_begin:row_end] = torch.from_numpy(i[row_begin:row_end])
    external[row_begin:row_end] = torch.from_numpy(e[row_begin:row_end])
    return internal, external


def pack_rows(cols, n):
    f = len(cols)
    ldr = max(2, (f + 1) & ~1)
    rows = torch.zeros((max(n, 1), ldr), dtype=torch.float64)
    for c, col in enumerate(cols):
        rows[:n, c] = col[:n]
    return rows, ldr


def aggregate(csr, rows, f, ldr, row_begin=0, row_end=None, want_sum=True, want_mean=True, out=None):
    n = csr.n
    row_end = n if row_end is None else row_end
    res = torch.zeros((2 * f, n), dtype=torch.float64) if out is None else out
    if f == 0:
        return res
    S, M = ckernels.aggregate(csr.row_ptr, csr.agg_col, np.ascontiguousarray(rows.numpy()[:n, :f]))
    if want_sum:
        res[:f, row_begin:row_end] = torch.from_numpy(S.T[:, row_begin:row_end].copy())
    if want_mean:
        res[f:, row_begin:row_end] = torch.from_numpy(M.T[:, row_begin:row_end].copy())
    return res


def aggregate_var(csr, rows, f, ldr, mean, row_begin=0, row_end=None, want_var=True, want_std=True):
    n = csr.n
    row_end = n if row_end is None else row_end
    res = torch.zeros((2 * f, n), dtype=torch.float64)
    if f == 0:
        return res
    var, std = ckernels.aggregate_var(csr.row_ptr, csr.agg_col, np.ascontiguousarray(rows.numpy()[:n, :f]))
    if want_var:
        res[:f, row_begin:row_end] = torch.from_numpy(var.T[:, row_begin:row_end].copy())
    if want_std:
        res[f:, row_begin:row_end] = torch.from_numpy(std.T[:, row_begin:row_end].copy())
    return res


def aggregate_prod(csr, rows, f, ldr, row_begin=0, row_end=None):
    n = csr.n
    row_end = n if row_end is None else row_end
    res = torch.ones((f, n), dtype=torch.float64)
    if f:
        P = ckernels.aggregate_prod(csr.row_ptr, csr.agg_col, np.ascontiguousarray(rows.numpy()[:n, :f]))
        res[:, row_begin:row_end] = torch.from_numpy(P.T[:, row_begin:row_end].copy())
    return res


def aggregate_minmax(csr, rows, f, ldr, row_begin=0, row_end=None, want_min=True, want_max=True):
    n = csr.n
    row_end = n if row_end is None else row_end
    res = torch.zeros((2 * f, n), dtype=torch.float64)
    if f == 0:
        return res
    lo, hi = ckernels.aggregate_minmax(csr.row_ptr, csr.agg_col, np.ascontiguousarray(rows.numpy()[:n, :f]))
    if want_min:
        res[:f, row_begin:row_end] = torch.from_numpy(lo.T[:, row_begin:row_end].copy())
    if want_max:
        res[f:, row_begin:row_end] = torch.from_numpy(hi.T[:, row_begin:row_end].copy())
    return res


# ---- int64-bits columns, median, count (csrc/grx_aggx.hip) on numpy -------------------------------------------------
def convert_i64_to_f64(col):
    return torch.from_numpy(col.numpy().view(np.int64).astype(np.float64))


def convert_f64_to_i64(col):
    return torch.from_numpy(col.numpy().astype(np.int64).view(np.float64).copy())


def _rows_of(csr, row_begin, row_end):
    return [(v, csr.agg_col[csr.row_ptr[v]:csr.row_ptr[v + 1]]) for v in range(row_begin, row_end)]


def aggregate_i64(csr, rows, f, ldr, row_begin=0, row_end=None, want=('sum', 'prod', 'min', 'max')):
    n = csr.n
    row_end = n if row_end is None else row_end
    X = rows.numpy()[:n, :f].view(np.int64)
    outs = {a: np.zeros((f, n), dtype=np.int64) for a in want}
    with np.errstate(over='ignore'):
        for v, nb in _rows_of(csr, row_begin, row_end):
            vals = X[nb]
            if 'sum' in outs:
                outs['sum'][:, v] = vals.sum(axis=0) if len(nb) else 0
            if 'prod' in outs:
                outs['prod'][:, v] = np.multiply.reduce(vals, axis=0) if len(nb) else 1
            if 'min' in outs:
                outs['min'][:, v] = vals.min(axis=0) if len(nb) else 0
            if 'max' in outs:
                outs['max'][:, v] = vals.max(axis=0) if len(nb) else 0
    return {a: torch.from_numpy(o.view(np.float64).copy()) for a, o in outs.items()}


def aggregate_count(csr, f, row_begin=0, row_end=None, as_i64=False):
    n = csr.n
    row_end = n if row_end is None else row_end
    deg = np.diff(csr.row_ptr).astype(np.int64)
    out = np.zeros((f, n), dtype=np.int64 if as_i64 else np.float64)
    out[:, row_begin:row_end] = deg[row_begin:row_end]
    return torch.from_numpy(out.view(np.float64).copy() if as_i64 else out)


def aggregate_median(csr, rows, f, ldr, row_begin=0, row_end=None):
    n = csr.n
    row_end = n if row_end is None else row_end
    X = rows.numpy()[:n, :f]
    out = np.zeros((f, n))
    for v, nb in _rows_of(csr, row_begin, row_end):
        if len(nb):
            out[:, v] = np.median(X[nb], axis=0)
    return torch.from_numpy(out)


def sort_columns(block):
    return torch.from_numpy(np.sort(block.numpy(), axis=1))


def vertical_log_bin(block, frac=0.5, out=None, is_i64=None):
    ncols, n = block.shape
    bins = out if out is not None else torch.zeros((ncols, n), dtype=torch.uint8)
    nb = torch.zeros(ncols, dtype=torch.int32)
    if not 0 < frac < 1:
        raise ValueError('must specify frac in interval (0, 1)')
    for j in range(ncols):
        if is_i64 is not None and is_i64[j]:
            from oracle import refex
            b = refex.vertical_log_binning(block[j].numpy().view(np.int64), frac)
        else:
            b = ckernels.vertical_log_binning(np.ascontiguousarray(block[j].numpy()), frac)
        bins[j] = torch.from_numpy(b.astype(np.uint8))
        nb[j] = int(b.max()) + 1 if n else 0
    return bins, nb


def chebyshev(bin_cols, n, first_new=0, row_begin=0, row_end=None, cap=255):
    F = len(bin_cols)
    row_end = n if row_end is None else row_end
    D = torch.zeros((F, F), dtype=torch.int32)
    if F >= 2 and row_end > row_begin:
        B = np.stack([c.numpy()[row_begin:row_end].astype(np.int32) for c in bin_cols])
        full = ckernels.chebyshev(B)
        mask = np.zeros((F, F), dtype=bool)
        mask[:, first_new:] = True
        mask[first_new:, :] = True
        D = torch.from_numpy(np.where(mask, full, 0).astype(np.int32))
    return D


def gather_columns(cols, n):
    F = len(cols)
    out = torch.zeros((F, max(n, 1)), dtype=torch.float64)
    for j, c in enumerate(cols):
        out[j, :n] = c[:n]
    return out


def gram(X, n, T=None, row_begin=0, row_end=None):
    row_end = n if row_end is None else row_end
    A = X.numpy()[:, row_begin:row_end].T
    Y = A if T is None else A @ T
    return Y.T @ Y, float(A.sum())


def project(X, n, Z, row_begin=0, row_end=None, out=None):
    row_end = n if row_end is None else row_end
    A = X.numpy()[:, :n].T
    U = A @ Z
    r = Z.shape[1]
    res = torch.zeros((r, X.shape[1]), dtype=torch.float64) if out is None else out
    res[:, row_begin:row_end] = torch.from_numpy(U.T[:, row_begin:row_end].copy())
    Us = U[row_begin:row_end]
    idx = np.argmax(np.abs(Us), axis=0)
    stats = np.stack([Us[idx, np.arange(r)], (idx + row_begin).astype(float),
                      (np.maximum(Us, 0) ** 2).sum(0), (np.minimum(Us, 0) ** 2).sum(0)], axis=1)
    return res, stats


def nndsvd_apply(U, n, sign, scale, eps, fill, row_begin=0, row_end=None):
    row_end = n if row_end is None else row_end
    u = U.numpy()
    for j in range(u.shape[0]):
        x = u[j, row_begin:row_end]
        v = np.abs(x) if sign[j] == 0 else np.maximum(sign[j] * x, 0)
        v = v * scale[j]
        u[j, row_begin:row_end] = np.where(v < eps, fill, v)


def lloyd_max(values, n_bins, max_iter=300):
    """Reference behaviour of encode(): sklearn KMeans(random_state=1) on the flat values."""
    import warnings
    from sklearn.cluster import KMeans
    data = values.numpy().reshape(-1, 1)
    km = KMeans(n_clusters=n_bins, random_state=1)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        km.fit(data)
    centres = km.cluster_centers_[:, 0]
    out = torch.from_numpy(centres[km.labels_].copy())
    nonempty = len(np.unique(km.labels_))
    return out, torch.from_numpy(np.sort(centres)), torch.tensor([km.n_iter_, nonempty, len(np.unique(out.numpy()))],
                                                                  dtype=torch.int32)


def kmeans1d(values, n_bins, max_iter=300, rel_tol=1e-4):
    """The reference's quantiser: sklearn KMeans(random_state=1) on the flat values."""
    return lloyd_max(values, n_bins)


def permute_columns(cols, index, n):
    idx = torch.as_tensor(np.asarray(index)).long() if not hasattr(index, 'long') else index.long()
    return torch.stack([c[idx] for c in cols]) if len(cols) else torch.zeros((0, n), dtype=torch.float64)


def transpose(src, rows, cols):
    return src[:rows, :cols].t().contiguous()


class NmfState:
    def __init__(self, X, n, W, H, x_sq_norm=None):
        self.X, self.n, self.W = X, n, W
        self.x_sq_norm = x_sq_norm
        self.F, self.r = X.shape[0], W.shape[0]
        H = H.numpy() if isinstance(H, torch.Tensor) else H
        self.H = torch.from_numpy(np.ascontiguousarray(H, dtype=np.float64).copy())
        self.AB = torch.zeros(self.r * self.F + self.r * self.r, dtype=torch.float64)
        self.err = torch.zeros(1, dtype=torch.float64)

    def w_pass(self, row_begin=0, row_end=None):
        row_end = self.n if row_end is None else row_end
        X = self.X.numpy()[:, row_begin:row_end].T
        W = self.W.numpy()[:, row_begin:row_end].T
        H = self.H.numpy()
        den = W @ (H @ H.T)
        den[den == 0] = rolx.EPSILON
        W1 = W * ((X @ H.T) / den)
        self.W.numpy()[:, row_begin:row_end] = W1.T
        self.AB[:self.r * self.F] = torch.from_numpy((W1.T @ X).ravel())
        self.AB[self.r * self.F:] = torch.from_numpy((W1.T @ W1).ravel())

    def h_update(self):
        A = self.AB.numpy()[:self.r * self.F].reshape(self.r, self.F)
        B = self.AB.numpy()[self.r * self.F:].reshape(self.r, self.r)
        H = self.H.numpy()
        den = B @ H
        den[den == 0] = rolx.EPSILON
        H *= A / den

    def residual_sq(self, row_begin=0, row_end=None):
        row_end = self.n if row_end is None else row_end
        X = self.X.numpy()[:, row_begin:row_end].T
        W = self.W.numpy()[:, row_begin:row_end].T
        R = X - W @ self.H.numpy()
        self.err[0] = float((R * R).sum())
        return self.err

    def w_pass_next(self, row_begin=0, row_end=None):
        self.h_update()
        self.w_pass(row_begin, row_end)

    def kl_cost(self, W, H, row_begin=0, row_end=None):
        row_end = self.n if row_end is None else row_end
        V = self.X.numpy()[:, row_begin:row_end].T
        A = W.numpy()[:, row_begin:row_end].T @ H.numpy()
        return rolx.error_cost(V, A)

    def iterate(self, iters, with_residual=True):
        for _ in range(iters):
            self.w_pass()
            self.h_update()
        if with_residual:
            self.residual_sq()


# ---- whole-loop entry points: the test double runs the per-kernel sequence of roles/factor.py (the
# sharded product path) without exchanges, on the fake kernels above
def nmf_init(X, n, r, omega):
    from graphrole_amd.roles import factor
    W0, H0, xx = factor._init_orchestrated(X, n, r, omega, None)
    return W0, torch.from_numpy(H0.copy()), xx


def nmf_mu(state, tol, max_iter):
    from graphrole_amd.roles import factor
    return factor._mu_orchestrated(state, tol, max_iter, None)


def nmf_fit(X, n, r, omega, tol, max_iter):
    W0, H0, xx = nmf_init(X, n, r, omega)
    state = NmfState(X, n, W0, H0, x_sq_norm=xx)
    return state, nmf_mu(state, tol, max_iter)


def role_argmax(G):
    return torch.from_numpy(rolx.dominant_role_index(G.numpy()))


def row_normalise(G):
    return torch.from_numpy(rolx.role_percentage(G.numpy()))
