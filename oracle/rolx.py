"""
oracle/rolx.py -- TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Numpy/scipy restatement of the RolX half of the GraphRole hot path:
``graphrole/roles/factor.py:10-26`` calls ``sklearn.decomposition.NMF(n_components=r,
solver='mu', init='nndsvda')`` -- a third-party dependency that is not under /root/reference.
Pinned version in this container: scikit-learn 1.7.2 (reference requires >=1.3.1,
requirements.txt:4).  The published algorithm is restated here from
``sklearn/decomposition/_nmf.py`` (:221-375 init, :526-556/:620-641/:706-727 updates,
:815-885 loop) and ``sklearn/utils/extmath.py`` (:287-357 range finder, :531-604 randomized
SVD, :895-953 svd_flip).

Parity: pinned by tests/test_oracle_pinned.py (test_oracle_nmf_matches_reference_golden) against factors produced by the reference's
``get_nmf_decomposition`` (tests/golden/nmf_*.npz, generated with ``np.random.seed`` fixed).
"""
from __future__ import annotations

from typing import Optional, Tuple

import numpy as np
from scipy import linalg

EPSILON = float(np.finfo(np.float32).eps)        # _nmf.py:39


def draw_omega(shape: Tuple[int, int], n_components: int, rng=None) -> np.ndarray:
    """
    The Gaussian test matrix of the range finder (extmath.py:297).  sklearn draws it from the
    GLOBAL numpy RNG (random_state=None, factor.py:19) with shape (min(N,F), r+10) -- the
    matrix is transposed first when N < F (extmath.py:565-569).
    """
    n, f = shape
    inner = n if n < f else f
    rng = np.random if rng is None else rng
    return rng.normal(size=(inner, n_components + 10))


def randomized_svd(X: np.ndarray, r: int, omega: np.ndarray):
    """extmath.py:531-604 with n_oversamples=10, n_iter='auto', normalizer 'auto' (LU)."""
    n, f = X.shape
    n_iter = 7 if r < 0.1 * min(n, f) else 4                       # extmath.py:557-560
    transpose = n < f
    M = X.T if transpose else X
    Q = omega
    if n_iter <= 2:
        norm = lambda a: a
    else:
        norm = lambda a: linalg.lu(a, permute_l=True, check_finite=False)[0]
    for _ in range(n_iter):                                          # extmath.py:349-351
        Q = norm(M @ Q)
        Q = norm(M.T @ Q)
    Q, _ = linalg.qr(M @ Q, mode='economic', check_finite=False)    # extmath.py:355
    B = Q.T @ M
    Uhat, s, Vt = linalg.svd(B, full_matrices=False, lapack_driver='gesdd')
    U = Q @ Uhat
    if not transpose:                                                # svd_flip u-based
        idx = np.argmax(np.abs(U), axis=0)
        signs = np.sign(U[idx, np.arange(U.shape[1])])
    else:                                                            # decide on rows of Vt
        idx = np.argmax(np.abs(Vt), axis=1)
        signs = np.sign(Vt[np.arange(Vt.shape[0]), idx])
    U = U * signs[None, :]
    Vt = Vt * signs[:, None]
    if transpose:
        return Vt[:r, :].T, s[:r], U[:, :r].T
    return U[:, :r], s[:r], Vt[:r, :]


def nndsvda_from_svd(U, S, V, x_mean: float, eps: float = 1e-6):
    """_nmf.py:316-359 (Boutsidis & Gallopoulos NNDSVD, zeros filled with mean(X))."""
    r = len(S)
    W = np.zeros_like(U)
    H = np.zeros_like(V)
    W[:, 0] = np.sqrt(S[0]) * np.abs(U[:, 0])
    H[0, :] = np.sqrt(S[0]) * np.abs(V[0, :])
    for j in range(1, r):
        x, y = U[:, j], V[j, :]
        x_p, y_p = np.maximum(x, 0), np.maximum(y, 0)
        x_n, y_n = np.abs(np.minimum(x, 0)), np.abs(np.minimum(y, 0))
        x_p_nrm, y_p_nrm = linalg.norm(x_p), linalg.norm(y_p)
        x_n_nrm, y_n_nrm = linalg.norm(x_n), linalg.norm(y_n)
        m_p, m_n = x_p_nrm * y_p_nrm, x_n_nrm * y_n_nrm
        if m_p > m_n:
            u, v, sigma = x_p / x_p_nrm, y_p / y_p_nrm, m_p
        else:
            u, v, sigma = x_n / x_n_nrm, y_n / y_n_nrm, m_n
        lbd = np.sqrt(S[j] * sigma)
        W[:, j] = lbd * u
        H[j, :] = lbd * v
    W[W < eps] = 0
    H[H < eps] = 0
    W[W == 0] = x_mean
    H[H == 0] = x_mean
    return W, H


def nndsvda_init(X: np.ndarray, r: int, omega: np.ndarray):
    if (X < 0).any():
        raise ValueError('Negative values in data passed to NMF initialization')   # _nmf.py:283
    if r > min(X.shape):
        raise ValueError("init = 'nndsvda' can only be used when n_components <= min(n_samples, n_features)")
    U, S, V = randomized_svd(X, r, omega)
    return nndsvda_from_svd(U, S, V, X.mean())


def frobenius_error(X, W, H) -> float:
    """_nmf.py:120-133 with square_root=True: ||X - WH||_F."""
    R = X - W @ H
    return float(np.sqrt(np.sum(R * R)))


def mu_iterations(X, W, H, tol: float = 1e-4, max_iter: int = 200):
    """_nmf.py:815-885 (beta_loss=2, no regularisation, gamma=1)."""
    W = W.copy()
    H = H.copy()
    err_init = frobenius_error(X, W, H)
    prev = err_init
    n_iter = 0
    for n_iter in range(1, max_iter + 1):
        numer = X @ H.T                                   # _nmf.py:541-544
        denom = W @ (H @ H.T)                             # :553-556
        denom[denom == 0] = EPSILON                       # :632
        W *= numer / denom
        numer = W.T @ X                                   # :706
        denom = np.linalg.multi_dot([W.T, W, H])          # :707
        denom[denom == 0] = EPSILON                       # :720
        H *= numer / denom
        if tol > 0 and n_iter % 10 == 0:                  # :872-885
            err = frobenius_error(X, W, H)
            if (prev - err) / err_init < tol:
                break
            prev = err
    return W, H, n_iter


def nmf(X: np.ndarray, r: int, omega: Optional[np.ndarray] = None, tol: float = 1e-4,
        max_iter: int = 200):
    """get_nmf_decomposition (roles/factor.py:10-26): returns (G=W, F=H, n_iter)."""
    X = np.asarray(X, dtype=np.float64)
    if omega is None:
        omega = draw_omega(X.shape, r)
    W0, H0 = nndsvda_init(X, r, omega)
    return mu_iterations(X, W0, H0, tol, max_iter)


# ----- model-selection pieces (roles/description_length.py, roles/extract.py) ---------------
def encoding_cost(G_enc, F_enc) -> float:
    """roles/description_length.py:32-41."""
    n_bins = max(len(np.unique(G_enc)), len(np.unique(F_enc)))
    return float(np.ceil(np.log2(n_bins)) * (G_enc.size + F_enc.size))


def error_cost(V, V_approx) -> float:
    """roles/description_length.py:44-61 (generalised KL with zero masking)."""
    v1 = np.asarray(V, dtype=np.float64).ravel()
    v2 = np.asarray(V_approx, dtype=np.float64).ravel()
    mask = v1 != 0
    logs = np.zeros_like(v1)
    np.log(v1 / v2, where=mask, out=logs)
    return float(np.sum(np.where(mask, v1 * logs - v1 + v2, 0)))


def rescale_costs(costs: np.ndarray) -> np.ndarray:
    """roles/extract.py:163-173."""
    norms = np.sqrt(np.nansum(np.square(costs), axis=1))
    return costs / norms.reshape(costs.shape[0], 1)


# ----- roles / role_percentage (roles/extract.py:38-57) --------------------------------------
def dominant_role_index(G: np.ndarray) -> np.ndarray:
    """Column of the first maximum of every row of the node-role factor: what DataFrame.idxmax(axis=1) resolves
    to (roles/extract.py:43-45; pandas nanargmax: NaN -> -inf, then numpy argmax = first maximum).  -1 marks a
    row of NaNs only."""
    G = np.asarray(G, dtype=np.float64)
    nan = np.isnan(G)
    idx = np.where(nan, -np.inf, G).argmax(axis=1).astype(np.int32)
    idx[nan.all(axis=1)] = -1
    return idx


def role_percentage(G: np.ndarray) -> np.ndarray:
    """row / row.sum() for every row (roles/extract.py:55).  Series.sum() of r float64 values is numpy's add.reduce
    over the row with NaN counted as 0 (pandas nanops.nansum): pairwise_sum -- left to right below 8 values, eight
    strided accumulators from 8 on; one reduce call per row keeps exactly that order."""
    G = np.ascontiguousarray(G, dtype=np.float64)
    filled = np.where(np.isnan(G), 0.0, G)
    out = np.empty_like(G)
    with np.errstate(invalid='ignore', divide='ignore'):
        for i in range(G.shape[0]):
            out[i] = G[i] / np.add.reduce(filled[i])
    return out
