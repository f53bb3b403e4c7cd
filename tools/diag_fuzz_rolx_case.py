"""Replay one case of tools/fuzz_rolx.py (same generator, same rng stream) and print where the device factors leave
the oracle's.  Usage: PYTHONPATH=. [FUZZ_WIDE_RANK=1] python tools/diag_fuzz_rolx_case.py <seed> <case>"""
import os
import sys

import numpy as np

from graphrole_amd.roles import factor
from oracle import rolx


def draw(rng):
    F = int(rng.choice([2, 3, 5, 8, 12, 16, 17, 20, 31, 33, 48, 49, 64, 70, 90, 115, 128, 129, 140]))
    n = int(rng.choice([F + 1, 2 * F + 3, 500, 3000, 20000]))
    n = max(n, F)
    r = int(rng.integers(2, min(8, F) + 1))
    if os.environ.get('FUZZ_WIDE_RANK') == '1' and F >= 12:
        r = int(rng.integers(9, min(32, F) + 1))
    kind = int(rng.integers(0, 4))
    X = np.abs(rng.standard_normal((n, F)))
    if kind == 1:
        X *= 10.0 ** rng.uniform(-2, 4, F)
    elif kind == 2:
        X *= rng.random((n, F)) < 0.2
        X[0] += 0.1
    elif kind == 3:
        k = max(2, F // 3)
        X = np.abs(rng.standard_normal((n, k))) @ np.abs(rng.standard_normal((k, F)))
    seed = int(rng.integers(0, 2 ** 31 - 1))
    return X, r, kind, seed


def main():
    seed0, case = int(sys.argv[1]), int(sys.argv[2])
    rng = np.random.default_rng(seed0)
    for _ in range(case):
        draw(rng)
    X, r, kind, seed = draw(rng)
    n, F = X.shape
    print(f'case {case}: n={n} F={F} r={r} kind={kind} cond(X)={np.linalg.cond(X):.3e}')
    np.random.seed(seed)
    G, H, n_iter = factor.nmf_with_info(X, r)
    np.random.seed(seed)
    omega = rolx.draw_omega(X.shape, r)
    W0, H0 = rolx.nndsvda_init(X, r, omega)
    We, He, it = rolx.mu_iterations(X, W0.copy(), H0.copy())
    print('n_iter', n_iter, it)
    print('W rel', np.abs(G - We).max() / np.abs(We).max(), 'H rel', np.abs(H - He).max() / np.abs(He).max())
    # where does it start: the initialisation, or the iterations?
    from graphrole_amd import kernels as K
    Xd = K.to_device(np.ascontiguousarray(X.T))
    W0d, H0d, _ = factor.nndsvda_init_device(Xd, n, r, omega)
    W0h = K.to_host(W0d)[:, :n].T
    H0h = K.to_host(H0d) if hasattr(H0d, 'is_cuda') else np.asarray(H0d)
    print('init W0 rel', np.abs(W0h - W0).max() / np.abs(W0).max(), 'H0 rel', np.abs(H0h - H0).max() / np.abs(H0).max())
    # oracle iterations started from the DEVICE init: how much of the gap is the init's
    Wd, Hd, itd = rolx.mu_iterations(X, W0h.copy(), H0h.copy())
    print('oracle MU from the device init: n_iter', itd, 'W rel vs device result', np.abs(G - Wd).max() / np.abs(Wd).max(),
          'vs oracle result', np.abs(Wd - We).max() / np.abs(We).max())
    s = np.linalg.svd(X, compute_uv=False)
    print('singular values', s[:3], '...', s[-3:])


if __name__ == '__main__':
    main()
