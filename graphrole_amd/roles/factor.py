"""
RolX factorisation (reference: graphrole/roles/factor.py).

``get_nmf_decomposition`` reproduces ``sklearn.decomposition.NMF(n_components=r, solver='mu',
init='nndsvda')`` (factor.py:19) with every O(N) pass on the GPU:

  init  (sklearn _nmf.py:316-359, extmath.py:531-604)
        X = Q M with Q orthonormal is obtained from two Gram passes (grx_gram): the first gives a
        whitening transform T1 from eigh(X^T X), the second re-orthogonalises (CholeskyQR2-style),
        M = pinv(T) is formed from the factors (no inversion).  Because Q has orthonormal columns,
        sklearn's randomised range finder on X maps to the same algorithm on the small matrix M
        (same Gaussian test matrix, drawn from numpy's global RNG exactly as sklearn does), so the
        singular triplets agree to rounding.  U = X Z, the svd_flip signs and the +/- part norms of
        NNDSVD come from one projection pass (grx_project); the element-wise NNDSVDa transform is
        grx_nndsvd_apply.  All k x F algebra is the library's own host code (grx_host_*).
  loop  (_nmf.py:815-885)  grx_nmf_iterate: fused W-update + W^T X / W^T W reduction pass,
        H-update kernel; the host reads one small block per convergence check.
  On one GPU both halves run below the ABI in one call (grx_nmf_fit, csrc/grx_fit.hip); with a ShardPlan
  the same sequence is driven from here, kernel by kernel, with the exchanges between the passes.

``encode`` (factor.py:29-49: sklearn KMeans(n_clusters, random_state=1) on the flattened entries) runs on the
GPU as the same procedure (grx_kmeans1d, csrc/grx_kmeans.hip); the deterministic Lloyd-Max solver
grx_lloyd_max (csrc/grx_quant.hip) is available as quantizer='lloyd_max' (SURVEY.md section 8(f) rank 1).
"""
from __future__ import annotations

from typing import Tuple

import numpy as np

from graphrole_amd.types import FactorTuple

class TooFewSamples(ValueError):
    """More quantisation levels requested than there are factor entries: sklearn's KMeans raises ValueError
    ("n_samples=.. should be >= n_clusters=..") in the reference's encode(); the model-selection grid of
    RoleExtractor skips exactly these cells (roles/extract.py:127-129) and nothing else."""


NMF_TOL = 1e-4          # sklearn NMF defaults (_nmf.py:1538-1553)
NMF_MAX_ITER = 200
NNDSVD_EPS = 1e-6


def _kernels():
    from graphrole_amd import backend
    return backend.get()


def _host():
    """The library's HOST routines for the k x F algebra of the initialisation (grx_host_*; no device
    work, usable without a GPU)."""
    from graphrole_amd import kernels
    return kernels


def _host_init(X: np.ndarray, r: int, omega: np.ndarray):
    """N < F (fewer nodes than features): every matrix of the initialisation is small -- sklearn's transposed
    randomized_svd branch (extmath.py:565-569, 587-604) runs in the library's host routines (grx_host_small_svd:
    LU-normalised power iterations, Householder QR, one-sided Jacobi SVD), then the NNDSVD column choices."""
    n, F = X.shape
    n_iter = 7 if r < 0.1 * min(X.shape) else 4                      # extmath.py:557-560
    U, S, V = _host().host_small_svd(X, omega, r, n_iter)
    # sklearn flips on the rows of its "Vt" = the columns of U here
    idx = np.argmax(np.abs(U), axis=0)
    stats = np.stack([U[idx, np.arange(r)], idx.astype(float), (np.maximum(U, 0) ** 2).sum(0),
                      (np.minimum(U, 0) ** 2).sum(0)], axis=1)
    sign, scale, H = _host().host_nndsvd_plan(S, V, stats)
    W = np.empty_like(U)
    for j in range(r):
        col = np.abs(U[:, j]) if sign[j] == 0 else np.maximum(sign[j] * U[:, j], 0)
        W[:, j] = col * scale[j]
    return W, H


def _merge_project_stats(per_rank: np.ndarray) -> np.ndarray:
    """Combine grx_project column statistics of disjoint row ranges ([world, r, 4]): the entry of
    largest magnitude (first row wins ties, like argmax) and the sums of squares."""
    world, r, _ = per_rank.shape
    out = np.zeros((r, 4))
    for j in range(r):
        best = 0
        for p in range(1, world):
            a, b = abs(per_rank[p, j, 0]), abs(per_rank[best, j, 0])
            if a > b or (a == b and per_rank[p, j, 1] < per_rank[best, j, 1]):
                best = p
        out[j, 0:2] = per_rank[best, j, 0:2]
        out[j, 2] = per_rank[:, j, 2].sum()
        out[j, 3] = per_rank[:, j, 3].sum()
    return out


def _init_orchestrated(Xd, n: int, r: int, omega: np.ndarray, plan=None):
    """
    The initialisation as a sequence of per-kernel calls with the exchanges of a ShardPlan between them
    (the multi-GPU path): every rank scans only its rows, the two Gram matrices are summed over the ranks,
    the projection statistics merged, and W0 is filled for the rank's own rows only.  On one GPU the
    same sequence runs inside libgrx.so (grx_nmf_init).
    """
    K = _kernels()
    H_ = _host()
    F = Xd.shape[0]
    rb, re = (0, n) if plan is None else (plan.row_begin, plan.row_end)

    def gram(T=None):
        G, xs = K.gram(Xd, n, T, rb, re)
        if plan is None:
            return G, xs
        packed = plan.all_reduce_sum_host(np.concatenate([G.reshape(-1), [xs]]))
        return packed[:-1].reshape(G.shape), float(packed[-1])

    G1, xsum = gram()
    x_mean = xsum / (n * F)
    x_sq_norm = float(np.trace(G1))                                  # ||X||_F^2
    n_iter = 7 if r < 0.1 * min(n, F) else 4                          # extmath.py:557-560
    T1, lam_keep, V_keep = H_.host_whiten(G1, r)
    if T1.shape[1] == 0:
        raise ValueError('NMF initialisation: the feature matrix is numerically zero')
    G2, _ = gram(T1)
    Z, S, Vt = H_.host_range_finder(T1, lam_keep, V_keep, G2, omega, r, n_iter)
    U, stats = K.project(Xd, n, Z, rb, re)
    if plan is not None:
        stats = _merge_project_stats(plan.all_gather_host(stats))
    sign, scale, H = H_.host_nndsvd_plan(S, Vt, stats)
    K.nndsvd_apply(U, n, sign, scale, NNDSVD_EPS, x_mean, rb, re)    # W[W < eps] = 0; W[W == 0] = mean
    H[H < NNDSVD_EPS] = 0
    H[H == 0] = x_mean
    return U, H, x_sq_norm


def nndsvda_init_device(Xd, n: int, r: int, omega: np.ndarray, plan=None):
    """
    NNDSVDa start for the feature-major device matrix Xd [F, ld] with n valid rows (N >= F):
    (W0 on the device, feature-major r x ld; H0 [r, F]; ||X||_F^2).  One GPU: a single call below the
    ABI (grx_nmf_init).  With a ShardPlan: the per-kernel sequence with its exchanges.
    """
    if plan is None:
        return _kernels().nmf_init(Xd, n, r, omega)
    return _init_orchestrated(Xd, n, r, omega, plan)


def draw_omega(shape: Tuple[int, int], n_roles: int) -> np.ndarray:
    """Gaussian test matrix exactly as sklearn draws it: global numpy RNG (random_state=None,
    factor.py:19), shape (min(N, F), r + 10) (extmath.py:297, 565-569)."""
    n, F = shape
    return np.random.normal(size=(n if n < F else F, n_roles + 10))


def _mu_orchestrated(state, tol: float, max_iter: int, plan=None) -> int:
    """
    sklearn's _fit_multiplicative_update driver (_nmf.py:815-885) over the per-kernel entry points, with
    the per-iteration all-reduce of a ShardPlan: ||X - WH||_F is read once per ten iterations and the
    reference's stopping rule (prev_err - err) / err_init < tol applied.  On one GPU the same loop runs
    inside libgrx.so (grx_nmf_mu).
    """
    K = _kernels()

    def residual_norm():
        if plan is not None:
            plan.all_reduce_sum_(state.err)
        return float(np.sqrt(K.to_host(state.err)[0]))

    def residual_from_identity():
        """||X - W H||_F from the W-pass outputs A = W^T X, B = W^T W (already summed over the ranks)
        and the updated H:  ||X||^2 - 2 <A, H> + <B, H H^T>.  No pass over X.  The identity cancels
        when the fit is nearly exact, so it is only trusted for a relative squared residual above 1e-8
        (its rounding error is then far under the 1e-4 stopping tolerance)."""
        xx = getattr(state, 'x_sq_norm', None)
        if xx is None or xx <= 0.0:
            return None
        r_, F_ = state.r, state.F
        AB = K.to_host(state.AB)
        H = K.to_host(state.H)
        A, B = AB[:r_ * F_].reshape(r_, F_), AB[r_ * F_:].reshape(r_, r_)
        sq = xx - 2.0 * float((A * H).sum()) + float((B * (H @ H.T)).sum())
        return float(np.sqrt(sq)) if sq > 1e-8 * xx else None

    rb, re = (0, state.n) if plan is None else (plan.row_begin, plan.row_end)
    state.residual_sq(rb, re)
    err_init = residual_norm()                                        # _nmf.py:826
    prev = err_init
    n_iter = 0
    while n_iter < max_iter:
        step = min(10, max_iter - n_iter)
        check = tol > 0 and (n_iter + step) % 10 == 0
        if plan is None:
            state.iterate(step, with_residual=False)
        else:
            _sharded_mu_block(state, plan, rb, re, step)
        n_iter += step
        if check:                                                     # _nmf.py:872-885
            err = residual_from_identity()
            if err is None:
                state.residual_sq(rb, re)
                err = residual_norm()
            if (prev - err) / err_init < tol:
                break
            prev = err
    return n_iter


def _sharded_mu_block(state, plan, rb: int, re: int, steps: int) -> None:
    """`steps` multiplicative updates of a row shard: W pass over the own rows, ONE all-reduce of the fused
    [W^T X ; W^T W] buffer, H update.  (Capturing the block -- kernels and the RCCL all-reduce -- as a HIP graph was
    measured on a one-rank RCCL group and removed: 8.80 ms per step against 8.00 for these plain calls, the
    capture per fit costs more than the launches it saves.)"""
    for it in range(steps):
        # iterations after the first fold the H update of their predecessor into the W pass (one launch and one
        # host call fewer); the update of the last one closes the block
        if it == 0:
            state.w_pass(rb, re)
        else:
            state.w_pass_next(rb, re)
        plan.all_reduce_sum_(state.AB)
    if steps:
        state.h_update()


def run_mu_loop(state, tol: float = NMF_TOL, max_iter: int = NMF_MAX_ITER, plan=None):
    """Multiplicative updates on an NmfState until sklearn's stopping rule fires; (state, n_iter)."""
    if plan is None:
        return state, _kernels().nmf_mu(state, tol, max_iter)
    return state, _mu_orchestrated(state, tol, max_iter, plan)


def nmf_device(Xd, n: int, n_roles: int, omega: np.ndarray,
               tol: float = NMF_TOL, max_iter: int = NMF_MAX_ITER, plan=None):
    """NNDSVDa + multiplicative updates on a device matrix; returns (NmfState, n_iter).  One GPU: ONE
    call below the ABI (grx_nmf_fit)."""
    K = _kernels()
    if plan is None:
        return K.nmf_fit(Xd, n, n_roles, omega, tol, max_iter)
    if getattr(K, 'NATIVE_SHARDING', False):
        # the sharded fit below the ABI: row passes on the rank's rows, exchanges issued by the C++ driver, W
        # completed on every rank at the end
        return K.nmf_fit(Xd, n, n_roles, omega, tol, max_iter, shard=plan)
    W0, H0, xx = _init_orchestrated(Xd, n, n_roles, omega, plan)
    state = K.NmfState(Xd, n, W0, H0, x_sq_norm=xx)
    return run_mu_loop(state, tol, max_iter, plan)


def get_nmf_decomposition(X: np.ndarray, n_roles: int) -> FactorTuple:
    """
    Compute NMF decomposition
    :param X: matrix to factor (n_nodes x n_features, non-negative)
    :param n_roles: rank of decomposition
    """
    G, F, _ = nmf_with_info(X, n_roles)
    return G, F


def _checked_matrix(X) -> np.ndarray:
    X = np.ascontiguousarray(np.asarray(X), dtype=np.float64)
    if X.ndim != 2:
        raise ValueError('X must be 2-dimensional')
    if X.size and X.min() < 0:
        raise ValueError('Negative values in data passed to NMF (input X)')        # _nmf.py:283
    return X


def _shape_of(X) -> Tuple[int, int]:
    return (int(X[0]), int(X[1])) if isinstance(X, tuple) else tuple(X.shape)


def nmf_state(Xd, X, n_roles: int, plan=None):
    """
    Run the factorisation of the matrix whose feature-major copy Xd [F, n] is already in HBM; X is the host matrix
    (n x F) or just its shape (n, F) when the table never left the device (features/handoff.py).
    Returns (NmfState with W [r, n] and H [r, F] on the device, n_iter).  Consumes numpy's
    global RNG exactly like sklearn (one Gaussian test matrix).  With a ShardPlan the row passes cover the
    rank's rows only (every rank must hold the same X and draw the same test matrix: seed numpy alike) and the
    rows of W are gathered at the end, so that every rank returns the complete factor.
    """
    K = _kernels()
    n, F = _shape_of(X)
    if n_roles > min(n, F):
        raise ValueError("init = 'nndsvda' can only be used when n_components <= min(n_samples, n_features)")
    omega = draw_omega((n, F), n_roles)
    if plan is not None and n >= F:
        state, n_iter = nmf_device(Xd, n, n_roles, omega, plan=plan)
        if not getattr(K, 'NATIVE_SHARDING', False):
            plan.all_gather_block(state.W[:, :n])
        return state, n_iter
    if n < F:
        # fewer nodes than features: every matrix of the initialisation is small (k x F algebra)
        if isinstance(X, tuple):
            X = np.ascontiguousarray(K.to_host(Xd)[:, :n].T)
        W0h, H0 = _host_init(X, n_roles, omega)
        H0 = H0.copy()
        W0h[W0h < NNDSVD_EPS] = 0
        H0[H0 < NNDSVD_EPS] = 0
        W0h[W0h == 0] = X.mean()
        H0[H0 == 0] = X.mean()
        return run_mu_loop(K.NmfState(Xd, n, K.to_device(np.ascontiguousarray(W0h.T)), H0))
    return nmf_device(Xd, n, n_roles, omega)


def feature_major(X: np.ndarray):
    """The n x F host table as the feature-major device matrix [F, n] the NMF kernels read: uploaded as it is
    (row-major) and transposed in HBM -- a strided host transpose of a multi-GB table costs seconds."""
    K = _kernels()
    n, F = X.shape
    return K.transpose(K.to_device(X), n, F)


def device_matrix(features):
    """
    (Xd [F, n] feature-major on the device, (n, F)) of the table handed to RoleExtractor -- a DataFrame or an array.
      * a DataFrame that RecursiveFeatureExtractor.extract_features() returned and nobody modified: the device block
        it was copied from (features/handoff.py) -- no upload, no transpose;
      * any other DataFrame: column by column (pandas stores columns contiguously) through the pipelined upload,
        straight into the feature-major layout;
      * arrays: upload + transpose in HBM.
    Negative or NaN entries raise ValueError like sklearn's NMF (_nmf.py:283, check_array); the check is one pass in
    HBM instead of a host pass.
    """
    import pandas as pd
    K = _kernels()
    if isinstance(features, pd.DataFrame) and hasattr(K, 'upload_into'):
        n, F = features.shape
        from graphrole_amd.features import handoff
        Xd = handoff.lookup(K, features)
        if Xd is None:
            Xd = K.empty((F, max(n, 1)))
            for j in range(F):
                col = features.iloc[:, j].to_numpy()
                if col.dtype != np.float64 or not col.flags.c_contiguous:
                    col = np.ascontiguousarray(col, dtype=np.float64)
                K.upload_into(Xd[j, :n], col) if n else None
        if n and F:
            lo = K.min_value(Xd, n)
            if lo != lo:
                raise ValueError('Input X contains NaN.')
            if lo < 0:
                raise ValueError('Negative values in data passed to NMF (input X)')        # _nmf.py:283
        return Xd, (n, F)
    X = _checked_matrix(features.values if isinstance(features, pd.DataFrame) else features)
    return feature_major(X), tuple(X.shape)


def nmf_with_info(X: np.ndarray, n_roles: int):
    """(G, F, n_iter); G = W (n x r), F = H (r x n_features)."""
    K = _kernels()
    X = _checked_matrix(X)
    n = X.shape[0]
    Xd = feature_major(X)
    state, n_iter = nmf_state(Xd, X, n_roles)
    return K.to_host(state.W)[:, :n].T.copy(), K.to_host(state.H).copy(), n_iter


QUANTIZERS = ('kmeans', 'lloyd_max')


def _quantize_flat(flat, n_bins: int, quantizer: str):
    """Quantise a flat device tensor (reference flatten order) -> (quantised, distinct output values)."""
    K = _kernels()
    if quantizer == 'kmeans':
        q, _, info = K.kmeans1d(flat, n_bins)
    elif quantizer == 'lloyd_max':
        q, _, info = K.lloyd_max(flat, n_bins)
    else:
        raise ValueError(f'quantizer must be one of {QUANTIZERS}, got {quantizer!r}')
    return q, info


class QuantizerFault(RuntimeError):
    """grx_kmeans1d reported a failed internal consistency check (include/grx.h, d_info[3])."""


def _checked_info(info, quantizer: str):
    """Host copy of a quantiser's info; the k-means seeding checks itself as it goes (its potential after every update,
    every search inside its block, every wait between workgroups): a non-zero report is an error, never a result."""
    info = _kernels().to_host(info)
    if quantizer == 'kmeans' and len(info) > 3 and int(info[3]) != 0:
        raise QuantizerFault(f'grx_kmeans1d: seeding fault bits {int(info[3]):#x} (include/grx.h: d_info[3])')
    return info


def encoded_factors_device(Xd, X, n_roles: int, n_bits: int, quantizer: str = 'kmeans', plan=None,
                           want_node_major: bool = False):
    """
    NMF of X (host matrix, or its shape (n, F) when only the device copy Xd exists) followed by the quantisation
    of both factors with 2**n_bits levels, without leaving HBM (roles/extract.py:144-161).  Returns (state,
    Wq [r, n], Hq [r, F], distinct values of Wq, distinct values of Hq); raises TooFewSamples (a ValueError, like the
    reference) when there are fewer factor entries than levels.  want_node_major: Wq is returned as the n x r
    row-major matrix the caller copies out (the layout the quantiser worked in) instead of [r, n].
    quantizer='kmeans' reproduces the reference's sklearn KMeans(random_state=1) (grx_kmeans1d);
    'lloyd_max' is the deterministic optimum-seeking quantiser (grx_lloyd_max): lower error, other numbers.
    """
    K = _kernels()
    n, F = _shape_of(X)
    state, _ = nmf_state(Xd, X, n_roles, plan)
    n_bins = int(2 ** n_bits)
    for size in (n_roles * n, n_roles * F):               # encode(G) first, then encode(F)
        if n_bins > size:
            raise TooFewSamples(f'n_samples={size} should be >= n_clusters={n_bins}.')
    # the reference quantises G.reshape(G.size, 1) with G = W as an n x r row-major matrix: that order is the
    # order of its cumulative sums, so the feature-major device factor is transposed first
    G_flat = K.transpose(state.W, n_roles, n).reshape(-1)
    Gq_flat, info_w = _quantize_flat(G_flat, n_bins, quantizer)
    Wq = Gq_flat.view(n, n_roles) if want_node_major else K.transpose(Gq_flat.view(n, n_roles), n, n_roles)
    Hq, info_h = _quantize_flat(state.H.reshape(-1), n_bins, quantizer)
    info_w, info_h = _checked_info(info_w, quantizer), _checked_info(info_h, quantizer)
    return state, Wq, Hq.view(n_roles, F), int(info_w[2]), int(info_h[2])


def encode(X: np.ndarray, n_bins: int, quantizer: str = 'kmeans') -> np.ndarray:
    """
    Encode (quantize) a matrix X using a specified number of bins: every entry is replaced by the centre of its
    cluster (factor.py:29-49).  quantizer='kmeans' (default): the reference's quantiser itself,
    sklearn KMeans(n_clusters=n_bins, random_state=1) on the flattened entries, reproduced on the GPU
    (grx_kmeans1d: same seeding draws, same Lloyd iterations and stopping rule; centres equal sklearn's to ~1e-12).
    quantizer='lloyd_max': grx_lloyd_max, a deterministic 1-D Lloyd-Max solver (exact DP start + Lloyd
    iterations) -- the same fixed-point conditions with an error not above KMeans', but other numbers.
    :param X: matrix to encode
    :param n_bins: number of bins for encoding
    """
    K = _kernels()
    X = np.asarray(X, dtype=np.float64)
    if n_bins > X.size:
        # sklearn raises the same for KMeans(n_clusters > n_samples); the model-selection loop
        # of RoleExtractor relies on it (roles/extract.py:127-129)
        raise TooFewSamples(f'n_samples={X.size} should be >= n_clusters={n_bins}.')
    flat = K.to_device(np.ascontiguousarray(X).reshape(-1))
    quantized, info = _quantize_flat(flat, int(n_bins), quantizer)
    _checked_info(info, quantizer)
    return K.to_host(quantized).reshape(X.shape)
