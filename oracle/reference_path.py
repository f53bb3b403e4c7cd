"""
oracle/reference_path.py -- TEST INFRASTRUCTURE / CPU BASELINE ONLY (see oracle/__init__.py).

The reference-faithful CPU formulation of the hot loops: the same third-party calls the
reference makes (pandas reindex/agg per node, sklearn NMF) -- what BASELINE.md section 3 calls
"baseline (1)".  It is interpreter bound (about 3 ms per node per generation), so bench.py times
it on a contiguous node sample and labels the full-graph figure as extrapolated.  Equality with
the reference itself: tests/test_oracle_pinned.py::test_reference_path_equals_golden.
"""
from __future__ import annotations

import time
from typing import Sequence, Tuple

import numpy as np
import pandas as pd


def aggregate_rows_pandas(row_ptr: np.ndarray, col: np.ndarray, features: pd.DataFrame,
                          rows: Sequence[int]) -> pd.DataFrame:
    """
    graphrole/features/extract.py:104-119 for the given rows: per node, select the neighbours'
    feature rows, aggregate with ['sum', 'mean'], NaN -> 0, name the results '<feature>(<agg>)'.
    `features` is indexed 0..n-1.
    """
    out = {}
    for v in rows:
        nbrs = col[row_ptr[v]:row_ptr[v + 1]]
        agg = features.reindex(index=nbrs).agg(['sum', 'mean']).fillna(0)
        flat = {}
        for how, series in agg.iterrows():
            for feat, val in series.items():
                flat[f'{feat}({how})'] = val
        out[v] = flat
    return pd.DataFrame.from_dict(out, orient='index')


def time_aggregate_sample(row_ptr, col, X: np.ndarray, names: Sequence[str], first_row: int,
                          n_rows: int) -> Tuple[float, int]:
    """Seconds and edges consumed for a contiguous block of rows (one generation)."""
    frame = pd.DataFrame(X, columns=list(names))
    rows = range(first_row, min(first_row + n_rows, len(row_ptr) - 1))
    t0 = time.perf_counter()
    aggregate_rows_pandas(row_ptr, col, frame, rows)
    dt = time.perf_counter() - t0
    edges = int(row_ptr[rows[-1] + 1] - row_ptr[rows[0]])
    return dt, edges


def sklearn_nmf(X: np.ndarray, n_roles: int):
    """graphrole/roles/factor.py:19-25 verbatim in effect: sklearn NMF(mu, nndsvda)."""
    import warnings
    from sklearn.decomposition import NMF
    model = NMF(n_components=n_roles, solver='mu', init='nndsvda')
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        t0 = time.perf_counter()
        W = model.fit_transform(X)
        dt = time.perf_counter() - t0
    return W, model.components_, model.n_iter_, dt
