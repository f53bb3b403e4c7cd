"""Differential fuzz of the RolX factorisation on the GPU box: random non-negative tables of many shapes
(n x F, F = 2..140 -- every Gram / W-pass instantiation family -- r = 2..8, with FUZZ_WIDE_RANK=1 r = 9..32: the
second half of one role tile and the composed update beyond it; dense / sparse / graded / rank-deficient)
through factor.nmf_with_info against oracle.rolx.nmf with the same numpy seed: equal iteration counts, factors to
1e-7 of their scale.  Usage: PYTHONPATH=. [FUZZ_WIDE_RANK=1] python tools/fuzz_rolx.py [cases] [seed]"""
import os
import sys

import numpy as np

from graphrole_amd.roles import factor
from oracle import rolx


def one(rng, case):
    F = int(rng.choice([2, 3, 5, 8, 12, 16, 17, 20, 31, 33, 48, 49, 64, 70, 90, 115, 128, 129, 140]))
    n = int(rng.choice([F + 1, 2 * F + 3, 500, 3000, 20000]))
    n = max(n, F)
    r = int(rng.integers(2, min(8, F) + 1))
    if os.environ.get('FUZZ_WIDE_RANK') == '1' and F >= 12:
        r = int(rng.integers(9, min(32, F) + 1))
    kind = int(rng.integers(0, 4))
    X = np.abs(rng.standard_normal((n, F)))
    if kind == 1:                                               # graded columns (degree-like scales)
        X *= 10.0 ** rng.uniform(-2, 4, F)
    elif kind == 2:                                             # sparse
        X *= rng.random((n, F)) < 0.2
        X[0] += 0.1                                             # no all-zero column
    elif kind == 3:                                             # rank deficient: duplicated / combined columns
        k = max(2, F // 3)
        X = np.abs(rng.standard_normal((n, k))) @ np.abs(rng.standard_normal((k, F)))
    seed = int(rng.integers(0, 2 ** 31 - 1))
    np.random.seed(seed)
    G, H, n_iter = factor.nmf_with_info(X, r)
    np.random.seed(seed)
    We, He, it = rolx.nmf(X, r)
    assert n_iter == it, (case, n, F, r, kind, n_iter, it)
    assert np.abs(G - We).max() <= 1e-7 * max(np.abs(We).max(), 1e-300), (case, n, F, r, kind)
    assert np.abs(H - He).max() <= 1e-7 * max(np.abs(He).max(), 1e-300), (case, n, F, r, kind)
    return f'n={n} F={F} r={r} kind={kind} iters={it}'


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    rng = np.random.default_rng(seed)
    for case in range(cases):
        r = one(rng, case)
        if case % 10 == 0:
            print(case, r, flush=True)
    print('fuzz ok:', cases, 'tables')


if __name__ == '__main__':
    main()
