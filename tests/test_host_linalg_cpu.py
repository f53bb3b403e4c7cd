"""
-m "not gpu": the host-side small dense algebra of libgrx.so (grx_host_whiten, grx_host_range_finder,
grx_host_nndsvd_plan; no device work) against the numpy / scipy formulation it replaces
(restated below with scipy) and against sklearn's randomized_svd; the symmetric eigen-solver against numpy.
"""
import ctypes

import numpy as np
import pytest
from scipy import linalg


def _vp(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _range_finder_svd(M, r, omega, shape):
    """sklearn's randomized_svd (extmath.py:531-604, n_oversamples=10, n_iter='auto', normalizer 'auto' = LU)
    applied to the k x F matrix M that stands for X = Q M -- the scipy formulation grx_host_range_finder
    replaces.  Returns (Us [k x r], S [r], Vt [r x F]) before svd_flip."""
    n_iter = 7 if r < 0.1 * min(shape) else 4

    def lu_norm(a):
        return linalg.lu(a, permute_l=True, check_finite=False)[0]

    Qs = omega
    for _ in range(n_iter):
        Qs = lu_norm(M @ Qs)
        Qs = lu_norm(M.T @ Qs)
    Qs, _ = linalg.qr(M @ Qs, mode='economic', check_finite=False)
    Uhat, s, Vt = linalg.svd(Qs.T @ M, full_matrices=False, lapack_driver='gesdd')
    Us = Qs @ Uhat
    k = Us.shape[1]
    if k < r:
        Us = np.hstack([Us, np.zeros((Us.shape[0], r - k))])
        s = np.concatenate([s, np.zeros(r - k)])
        Vt = np.vstack([Vt, np.zeros((r - k, Vt.shape[1]))])
    return Us[:, :r], s[:r], Vt[:r]


def _nndsvd_plan(S, Vt, stats):
    """Column choices of NNDSVD (_nmf.py:324-352) in numpy -- what grx_host_nndsvd_plan replaces."""
    r, F = Vt.shape
    flip = np.sign(stats[:, 0])
    flip[flip == 0] = 1.0
    sign, scale, H = np.zeros(r), np.zeros(r), np.zeros((r, F))
    scale[0] = np.sqrt(S[0])
    H[0] = np.sqrt(S[0]) * np.abs(Vt[0])
    for j in range(1, r):
        y = Vt[j] * flip[j]
        x_p_nrm = np.sqrt(stats[j, 2] if flip[j] > 0 else stats[j, 3])
        x_n_nrm = np.sqrt(stats[j, 3] if flip[j] > 0 else stats[j, 2])
        y_p, y_n = np.maximum(y, 0), np.abs(np.minimum(y, 0))
        y_p_nrm, y_n_nrm = linalg.norm(y_p), linalg.norm(y_n)
        m_p, m_n = x_p_nrm * y_p_nrm, x_n_nrm * y_n_nrm
        with np.errstate(invalid='ignore', divide='ignore'):
            if m_p > m_n:
                x_nrm, v, sigma, part = x_p_nrm, y_p / y_p_nrm, m_p, 1.0
            else:
                x_nrm, v, sigma, part = x_n_nrm, y_n / y_n_nrm, m_n, -1.0
        lbd = np.sqrt(S[j] * sigma)
        if not np.isfinite(lbd) or x_nrm == 0:
            sign[j], scale[j] = 1.0, 0.0
            continue
        sign[j] = flip[j] * part
        scale[j] = lbd / x_nrm
        H[j] = lbd * v
    return sign, scale, H


@pytest.mark.parametrize('n,F,r,deficient', [(5000, 20, 6, False), (3000, 12, 6, True), (4000, 40, 4, False),
                                             (2000, 7, 3, False), (3000, 9, 2, False), (800, 64, 8, False),
                                             (500, 5, 5, True), (1500, 115, 6, False), (1200, 200, 6, True)])
def test_native_small_space_equals_scipy_formulation(n, F, r, deficient):
    from graphrole_amd import kernels as K
    rng = np.random.RandomState(n + F)
    X = np.abs(rng.randn(n, F)) * np.linspace(1, 50, F)
    if deficient:
        X[:, F - 1] = X[:, 0] + X[:, 1]                        # exact linear dependency
    omega = rng.normal(size=(F, r + 10))
    G1 = X.T @ X
    lam, V1 = linalg.eigh(G1)
    keep = lam > max(lam.max(), 0.0) * F * np.finfo(np.float64).eps * 16
    T1, lam_keep, V_keep = K.host_whiten(G1)
    assert T1.shape == (F, int(keep.sum())) and lam_keep.shape == (int(keep.sum()),)
    np.testing.assert_allclose(lam_keep, lam[keep], rtol=1e-10)
    # (lam_keep, V_keep) reproduce G1 on the kept subspace, which is all grx_host_range_finder needs of them
    np.testing.assert_allclose((V_keep * lam_keep) @ V_keep.T, G1, atol=1e-9 * np.abs(G1).max())
    Y = X @ T1
    np.testing.assert_allclose(Y.T @ Y, np.eye(T1.shape[1]), atol=1e-8)          # whitened
    # the scipy formulation on its own basis
    T1p = V1[:, keep] / np.sqrt(lam[keep])
    Yp = X @ T1p
    lam2, V2 = linalg.eigh(Yp.T @ Yp)
    Tp = (T1p @ V2) / np.sqrt(lam2)
    Mp = (np.sqrt(lam2)[:, None] * V2.T) @ (np.sqrt(lam[keep])[:, None] * V1[:, keep].T)
    Usp, Sp, Vtp = _range_finder_svd(Mp, r, omega, (n, F))
    n_iter = 7 if r < 0.1 * min(n, F) else 4
    Z, S, Vt = K.host_range_finder(T1, lam_keep, V_keep, Y.T @ Y, omega, r, n_iter)
    U, Up = X @ Z, X @ (Tp @ Usp)
    sg = np.sign((U * Up).sum(axis=0))
    sg[sg == 0] = 1
    live = Sp > 1e-9 * Sp.max()                                 # columns beyond the rank are arbitrary
    np.testing.assert_allclose(S[live], Sp[live], rtol=1e-9)
    np.testing.assert_allclose((U * sg)[:, live], Up[:, live], atol=1e-9 * np.abs(Up).max())
    np.testing.assert_allclose((Vt * sg[:, None])[live], Vtp[live], atol=1e-9)
    # ... which is sklearn's randomized_svd of X itself (same omega: a RandomState subclass hands it out)
    from sklearn.utils.extmath import randomized_svd

    class _Rs(np.random.RandomState):
        def normal(self, loc=0.0, scale=1.0, size=None):
            assert tuple(size) == omega.shape
            return omega.copy()

    _, Ssk, _ = randomized_svd(X, r, n_oversamples=10, random_state=_Rs(0), flip_sign=False)
    np.testing.assert_allclose(S[live], Ssk[live], rtol=1e-8)
    # NNDSVD choices from the projection statistics
    idx = np.argmax(np.abs(U), axis=0)
    stats = np.stack([U[idx, np.arange(r)], idx.astype(float), (np.maximum(U, 0) ** 2).sum(0),
                      (np.minimum(U, 0) ** 2).sum(0)], axis=1)
    sign_p, scale_p, H_p = _nndsvd_plan(S, Vt, stats)
    sign, scale, H = K.host_nndsvd_plan(S, Vt, stats)
    assert np.array_equal(sign, sign_p)
    np.testing.assert_allclose(scale, scale_p, rtol=1e-14)
    np.testing.assert_allclose(H, np.nan_to_num(H_p), rtol=1e-14, atol=0)


def test_host_routines_validate_arguments():
    from graphrole_amd import _lib
    lib = _lib.load()
    assert lib.grx_host_whiten(0, None, None, None, None, None) == -1
    assert lib.grx_host_range_finder(4, 5, None, None, None, None, None, 3, 2, 4, None, None, None) == -1   # k > F
    assert lib.grx_host_nndsvd_plan(2, 3, None, None, None, None, None, None) == -1


@pytest.mark.parametrize('n', [1, 2, 7, 33, 64, 65, 130, 257])
def test_host_eigh_matches_numpy(n):
    """grx_host_eigh: cyclic Jacobi up to 32 columns, Householder tridiagonalisation + implicit QL above."""
    from graphrole_amd import _lib
    lib = _lib.load()
    rng = np.random.RandomState(n)
    X = np.abs(rng.randn(3 * n + 5, n)) * np.linspace(1, 30, n)
    if n > 4:
        X[:, n - 1] = X[:, 0] - 2 * X[:, 1]                      # rank deficient
    A = X.T @ X
    w, V = np.empty(n), np.empty((n, n))
    assert lib.grx_host_eigh(n, _vp(A), _vp(w), _vp(V)) == 0
    we = np.linalg.eigvalsh(A)
    scale = max(we.max(), 1e-300)
    assert np.all(np.diff(w) >= 0)
    np.testing.assert_allclose(w, we, atol=1e-13 * scale * n)
    np.testing.assert_allclose(A @ V, V * w, atol=1e-13 * scale * n)
    np.testing.assert_allclose(V.T @ V, np.eye(n), atol=1e-12)


@pytest.mark.parametrize('n,F,r', [(13, 30, 3), (5, 8, 3), (3, 40, 2), (12, 13, 6), (18, 100, 8), (7, 9, 7)])
def test_small_svd_of_tables_with_fewer_nodes_than_features(n, F, r):
    """grx_host_small_svd (n < F: sklearn's transposed randomized_svd branch on the library's own LU / QR / Jacobi
    routines) against numpy's SVD: when r + 10 >= n the range finder spans the whole row space, so the leading r
    singular triplets are exact (up to the sign of a pair)."""
    from graphrole_amd import kernels as K
    rng = np.random.default_rng(n * 100 + F)
    X = np.abs(rng.standard_normal((n, F))) * np.linspace(1, 5, F)
    omega = rng.standard_normal((n, r + 10))
    U, S, V = K.host_small_svd(X, omega, r, 4)
    Ue, Se, Vte = np.linalg.svd(X, full_matrices=False)
    np.testing.assert_allclose(S, Se[:r], rtol=1e-10)
    for j in range(r):
        sgn = np.sign(U[:, j] @ Ue[:, j])
        np.testing.assert_allclose(sgn * U[:, j], Ue[:, j], atol=1e-8)
        np.testing.assert_allclose(sgn * V[j], Vte[j], atol=1e-8)


@pytest.mark.parametrize('seed', [0, 1, 2, 3])
def test_whitening_keeps_every_direction_of_a_graded_table(seed):
    """Column norms six decades apart (a degree column next to a mean of means) and r = F: the plain eigen-decomposition
    of the Gram matrix drops the smallest directions (its floor is set by the largest column); asked for rank r,
    grx_host_whiten_for_rank then equilibrates the columns exactly (powers of two) and keeps them all
    (tools/fuzz_rolx.py, seed 403 case 103)."""
    from graphrole_amd import kernels as K
    rng = np.random.RandomState(seed)
    n, F = 18, 17
    X = np.abs(rng.randn(n, F)) * 10.0 ** rng.uniform(-2, 4, F)
    G1 = X.T @ X
    k_plain = K.host_whiten(G1)[0].shape[1]
    assert K.host_whiten(G1, rank=min(6, k_plain))[0].shape[1] == k_plain    # enough directions for the rank: the plain result
    if seed == 0:
        assert k_plain < F                                        # the plain decomposition loses directions here ...
    T1, lam_keep, V_keep = K.host_whiten(G1, rank=F)
    assert T1.shape == (F, F)                                     # ... and all are kept when all are asked for
    np.testing.assert_allclose((V_keep * lam_keep) @ V_keep.T, G1, atol=1e-9 * np.abs(G1).max())
    Y = X @ T1
    np.testing.assert_allclose(Y.T @ Y, np.eye(F), atol=1e-6)
    r = F
    omega = rng.normal(size=(F, r + 10))
    Z, S, Vt = K.host_range_finder(T1, lam_keep, V_keep, Y.T @ Y, omega, r, 4)
    s_true = np.linalg.svd(X, compute_uv=False)
    np.testing.assert_allclose(S, s_true, rtol=1e-6)             # all seventeen, down to 1e-7 of the largest
    U = X @ Z
    np.testing.assert_allclose(U.T @ U, np.eye(r), atol=1e-6)
