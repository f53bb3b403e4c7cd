"""Differential fuzz of grx_vertical_log_bin (the sort-free default path) against the oracle's vertical_log_binning on
the GPU box: random heights, value distributions aimed at the bucket map (ties, clusters at ulp spacing, huge dynamic
range, outliers off the sample grid, sorted input, int64-bits columns), random `frac`.
Usage: PYTHONPATH=. python tools/fuzz_binning.py [cases] [seed]"""
import sys

import numpy as np
import torch

from graphrole_amd import kernels as K
from oracle import ckernels, refex


def column(rng, n):
    kind = int(rng.integers(0, 12))
    if kind == 0:
        x = rng.pareto(rng.uniform(0.8, 3.0), n).round(int(rng.integers(0, 4)))
    elif kind == 1:
        x = rng.integers(0, int(rng.choice([2, 7, 100, 5000, 10 ** 6])), n).astype(np.float64)
    elif kind == 2:
        x = rng.standard_normal(n) * 10.0 ** rng.integers(-8, 8)
    elif kind == 3:
        x = np.where(rng.random(n) < rng.uniform(0.05, 0.99), 0.0, rng.lognormal(0, 3, n))
    elif kind == 4:
        x = np.sort(rng.pareto(1.2, n))
        if rng.random() < 0.5:
            x = x[::-1].copy()
    elif kind == 5:                                           # clusters at ulp spacing + far outliers
        x = 1.0 + rng.integers(0, int(rng.choice([3, 1 << 8, 1 << 20, 1 << 40])), n) * 2.0 ** -52
        x[rng.integers(0, n, max(n // 1000, 1))] = rng.choice([1e6, -1e6, 1e300, 5e-324])
    elif kind == 6:
        x = rng.standard_normal(n) * 10.0 ** rng.integers(-300, 300, n)
    elif kind == 7:                                           # few heavy values + continuous noise
        vals = rng.standard_normal(int(rng.integers(1, 6)))
        x = np.where(rng.random(n) < 0.8, vals[rng.integers(0, len(vals), n)], rng.standard_normal(n))
    elif kind == 8:                                           # tie runs of a chosen length everywhere
        run = int(rng.choice([2, 50, 300, 5000]))
        x = (rng.integers(0, max(n // run, 2), n) * rng.uniform(1e-6, 10.0)).astype(np.float64)
    elif kind == 9:                                           # outliers exactly off the sample grid
        x = rng.random(n)
        stride = max(n // 4095, 1)
        idx = np.arange(1 if stride > 1 else 0, n, max(stride * int(rng.integers(1, 9)), 1))
        x[idx[:max(len(idx) // 50, 1)]] = rng.choice([1e9, -1e9])
    elif kind == 10:
        x = -np.round(rng.pareto(1.1, n) * rng.choice([1, 20, 1000]))
    else:
        x = np.full(n, rng.standard_normal())
        if rng.random() < 0.5:
            x[rng.integers(0, n, 3)] += 1.0
    return np.ascontiguousarray(x, dtype=np.float64), False


def int_column(rng, n):
    kind = int(rng.integers(0, 4))
    if kind == 0:
        x = rng.integers(-2 ** 63, 2 ** 63 - 1, n, dtype=np.int64, endpoint=True)
    elif kind == 1:
        x = rng.integers(-5, 5, n).astype(np.int64)
    elif kind == 2:
        x = (2 ** 53 + rng.integers(0, 5, n)).astype(np.int64)
    else:
        x = (rng.pareto(1.0, n) * 1e12).astype(np.int64) * rng.choice([-1, 1], n)
    return x, True


def one(rng):
    n = int(rng.choice([3, 40, 1000, 4095, 4097, 70_000, 300_000, 1_200_000], p=[.05, .05, .15, .1, .1, .25, .2, .1]))
    ncols = int(rng.integers(1, 9))
    cols = [(int_column if rng.random() < 0.2 else column)(rng, n) for _ in range(ncols)]
    frac = float(rng.choice([0.5, 0.5, 0.3, 0.7, 0.25]))
    block = np.stack([c.view(np.float64) if is_int else c for c, is_int in cols])
    flags = [is_int for _, is_int in cols]
    bins, nb = K.vertical_log_bin(torch.from_numpy(block).cuda(), frac, is_i64=flags)
    got = bins.cpu().numpy()
    for j, (c, is_int) in enumerate(cols):
        exp = refex.vertical_log_binning(c, frac) if is_int else ckernels.vertical_log_binning(c, frac)
        if exp.max() >= 128:
            assert int(nb[j]) < 0, (n, j, frac)
            continue
        if not np.array_equal(got[j], exp) or int(nb[j]) != exp.max() + 1:
            np.save('/tmp/fuzz_binning_fail.npy', block[j])
            raise AssertionError(f'mismatch: n={n} col={j} int={is_int} frac={frac} nb={int(nb[j])} expected {exp.max() + 1}')
    return n * ncols


if __name__ == '__main__':
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    rng = np.random.default_rng(seed)
    total = 0
    for case in range(cases):
        total += one(rng)
    print(f'fuzz_binning: {cases} cases, {total} keys, all equal')
