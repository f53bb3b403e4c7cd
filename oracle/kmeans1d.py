"""
oracle/kmeans1d.py -- TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Numpy restatement of the quantiser behind the reference's ``encode`` (graphrole/roles/factor.py:29-49):
``sklearn.cluster.KMeans(n_clusters=k, random_state=1).fit(values.reshape(-1, 1))`` followed by
``cluster_centers_[labels_]`` -- scikit-learn is a third-party dependency that is not under /root/reference
(requirements.txt:4 ``>=1.3.1``; 1.7.2 in this container).  Restated from its published source for ONE feature
column and unit sample weights:
  sklearn/cluster/_kmeans.py   KMeans.fit (:1445-1545: tolerance, mean-centring, n_init='auto' -> 1 run),
                               _kmeans_plusplus (:163-257), _kmeans_single_lloyd (:620-720)
  sklearn/cluster/_k_means_lloyd.pyx / _k_means_common.pyx   E step (first minimum of c^2 - 2 x c), M step,
                               _relocate_empty_clusters_dense, _average_centers, _center_shift
  sklearn/metrics/pairwise.py  _euclidean_distances (:370-420: -2 x.y + |x|^2 + |y|^2, clipped at 0)
Parity: pinned against sklearn itself by tests/test_oracle_pinned.py::test_oracle_kmeans1d_equals_sklearn.
"""
from __future__ import annotations

from typing import Tuple

import numpy as np


def _sq_dist(c: float, x: np.ndarray, xsq: np.ndarray) -> np.ndarray:
    """_euclidean_distances of one candidate against every value, in sklearn's operation order."""
    d = -2.0 * (x * c)
    d += c * c
    d += xsq
    np.maximum(d, 0, out=d)
    return d


def kmeans_plusplus(x: np.ndarray, k: int, rs: np.random.RandomState) -> np.ndarray:
    """_kmeans_plusplus for mean-centred 1-D data x: indices of the chosen seeds, in order."""
    m = len(x)
    xsq = x * x
    w = np.ones(m)
    trials = 2 + int(np.log(k))
    picked = np.empty(k, dtype=np.int64)
    picked[0] = rs.choice(m, p=w / w.sum())
    d = _sq_dist(x[picked[0]], x, xsq)
    pot = d @ w
    for c in range(1, k):
        rand_vals = rs.uniform(size=trials) * pot
        cand = np.searchsorted(np.cumsum(w * d, dtype=np.float64), rand_vals)
        np.clip(cand, None, m - 1, out=cand)
        D = np.stack([_sq_dist(x[j], x, xsq) for j in cand])
        np.minimum(d, D, out=D)
        pots = D @ w
        best = int(np.argmin(pots))
        pot, d = pots[best], D[best]
        picked[c] = cand[best]
    return picked


def lloyd(x: np.ndarray, centers: np.ndarray, tol: float, max_iter: int = 300) -> Tuple[np.ndarray, np.ndarray, int]:
    """_kmeans_single_lloyd: (labels, centers, n_iter)."""
    m, k = len(x), len(centers)
    centers = centers.copy()
    labels_old = np.full(m, -1, dtype=np.int32)

    def e_step(c):
        # first minimum of |c|^2 - 2 x c (the |x|^2 term is common to a row)
        out = np.empty(m, dtype=np.int32)
        csq = c * c
        for lo in range(0, m, 65536):
            xs = x[lo:lo + 65536]
            out[lo:lo + 65536] = np.argmin(csq[None, :] - 2.0 * (xs[:, None] * c[None, :]), axis=1)
        return out

    strict = False
    n_iter = 0
    labels = labels_old
    for it in range(max_iter):
        n_iter = it + 1
        labels = e_step(centers)
        sums = np.bincount(labels, weights=x, minlength=k)
        counts = np.bincount(labels, minlength=k).astype(np.float64)
        empty = np.where(counts == 0)[0]
        if len(empty):
            # _relocate_empty_clusters_dense: the points farthest from their centre seed the empty clusters
            dist = (x - centers[labels]) ** 2
            far = np.argpartition(dist, -len(empty))[:-len(empty) - 1:-1]
            for new_id, idx in zip(empty, far):
                old_id = labels[idx]
                sums[old_id] -= x[idx]
                sums[new_id] = x[idx]
                counts[new_id] = 1.0
                counts[old_id] -= 1.0
        new = np.where(counts > 0, sums * (1.0 / np.where(counts > 0, counts, 1.0)), 0.0)
        shift = np.abs(new - centers)
        centers = new
        if np.array_equal(labels, labels_old):
            strict = True
            break
        if (shift ** 2).sum() <= tol:
            break
        labels_old = labels
    if not strict:
        labels = e_step(centers)
    return labels, centers, n_iter


def kmeans_quantize(values: np.ndarray, k: int, seed: int = 1, max_iter: int = 300, rel_tol: float = 1e-4):
    """encode(): every value replaced by the centre of its cluster; (quantised, centres, n_iter)."""
    v = np.ascontiguousarray(values, dtype=np.float64).reshape(-1)
    if k > len(v):
        raise ValueError(f'n_samples={len(v)} should be >= n_clusters={k}.')
    rs = np.random.RandomState(seed)
    tol = float(np.mean(np.var(v.reshape(-1, 1), axis=0)) * rel_tol)         # _tolerance
    mean = v.reshape(-1, 1).mean(axis=0)[0]
    x = v - mean
    seeds = kmeans_plusplus(x, k, rs)
    labels, centers, n_iter = lloyd(x, x[seeds], tol, max_iter)
    centers = centers + mean
    return centers[labels].reshape(np.shape(values)), centers, n_iter
