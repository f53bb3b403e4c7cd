"""
-m "not gpu": the host-side small dense algebra of libgrx.so (grx_host_whiten, grx_host_range_finder,
grx_host_nndsvd_plan; no device work) against the numpy / scipy formulation it replaces
(graphrole_amd/roles/factor.py::_range_finder_svd, _nndsvd_plan) and against sklearn's randomized_svd.
"""
import ctypes

import numpy as np
import pytest
from scipy import linalg


def _vp(a):
    return a.ctypes.data_as(ctypes.c_void_p)


@pytest.mark.parametrize('n,F,r,deficient', [(5000, 20, 6, False), (3000, 12, 6, True), (4000, 40, 4, False),
                                             (2000, 7, 3, False), (3000, 9, 2, False), (800, 64, 8, False),
                                             (500, 5, 5, True)])
def test_native_small_space_equals_scipy_formulation(n, F, r, deficient):
    from graphrole_amd import kernels as K
    from graphrole_amd.roles import factor
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
    Y = X @ T1
    np.testing.assert_allclose(Y.T @ Y, np.eye(T1.shape[1]), atol=1e-8)          # whitened
    # the scipy formulation on its own basis
    T1p = V1[:, keep] / np.sqrt(lam[keep])
    Yp = X @ T1p
    lam2, V2 = linalg.eigh(Yp.T @ Yp)
    Tp = (T1p @ V2) / np.sqrt(lam2)
    Mp = (np.sqrt(lam2)[:, None] * V2.T) @ (np.sqrt(lam[keep])[:, None] * V1[:, keep].T)
    Usp, Sp, Vtp = factor._range_finder_svd(Mp, r, omega, (n, F))
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
    sign_p, scale_p, H_p = factor._nndsvd_plan(S, Vt, stats)
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
