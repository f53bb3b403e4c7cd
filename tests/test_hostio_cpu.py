"""-m "not gpu": the host half of csrc/grx_hostio.hip -- the content hashes behind the device-resident hand-off."""
import ctypes

import numpy as np


def _sums(a):
    from graphrole_amd import _lib
    a = np.ascontiguousarray(a)
    out = np.zeros(a.shape[0], dtype=np.uint64)
    _lib.call('grx_host_checksums', a.ctypes.data_as(ctypes.c_void_p), a.shape[0], a.shape[1] * 8, a.strides[0],
              out.ctypes.data_as(ctypes.c_void_p))
    return out


def test_checksums_see_bit_flips_swaps_and_shifts():
    rng = np.random.default_rng(0)
    a = rng.random((7, 200_003))
    base = _sums(a)
    assert len(set(base.tolist())) == 7
    assert np.array_equal(base, _sums(a.copy()))
    b = a.copy()
    b[3, 77] = np.nextafter(b[3, 77], 2)                         # one bit
    assert (base != _sums(b)).tolist() == [False, False, False, True, False, False, False]
    b = a.copy()
    b[5, [10, 11]] = b[5, [11, 10]]                               # two neighbours swapped
    assert (base != _sums(b)).nonzero()[0].tolist() == [5]
    b = a.copy()
    b[1, 64:128], b[1, 128:192] = a[1, 128:192], a[1, 64:128]     # two whole cache-line groups swapped
    assert (base != _sums(b)).nonzero()[0].tolist() == [1]
    b = a.copy()
    b[6] = np.roll(a[6], 8)                                       # shifted by one step of the hash
    assert (base != _sums(b)).nonzero()[0].tolist() == [6]


def test_checksum_of_a_column_does_not_depend_on_the_batch():
    rng = np.random.default_rng(1)
    a = rng.random((5, 131_072 + 9))
    together = _sums(a)
    for j in range(5):
        assert _sums(a[j:j + 1])[0] == together[j]
    ints = rng.integers(0, 1 << 40, size=(3, 70_000), dtype=np.int64)
    assert np.array_equal(_sums(ints.view(np.float64)), _sums(ints.view(np.float64).copy()))


def test_uniform_choice_equals_numpy_searchsorted_of_the_cumulative_sum():
    """grx_host_uniform_choice: the index RandomState.choice(m, p=uniform) derives from its one draw (sklearn's first
    k-means++ seed) -- numpy's searchsorted(cumsum(full(m, 1/m)) / total, u, 'right') -- incl. draws that sit on or
    next to a boundary of the cumulative sum (the exact-loop branch)."""
    from graphrole_amd import _lib

    def ours(m, u):
        idx = ctypes.c_int64(0)
        _lib.call('grx_host_uniform_choice', int(m), float(u), ctypes.byref(idx))
        return idx.value

    def numpy_way(m, u):
        cdf = np.cumsum(np.full(m, 1.0 / m))
        cdf /= cdf[-1]
        return int(np.searchsorted(cdf, u, side='right'))

    u1 = np.random.RandomState(1).random_sample()
    for m in list(range(1, 120)) + [1000, 4099, 65536, 100003, 1234567]:
        draws = [u1, 0.0, 0.5, 0.999999, 1.0 / 3]
        draws += [(k + eps) / m for k in (0, 1, m // 2, m - 1) for eps in (0.0, 1e-12, -1e-12, 0.5) if 0 <= (k + eps) / m < 1]
        for u in draws:
            assert ours(m, u) == numpy_way(m, u), (m, u)


def test_pool_back_to_back_fork_joins_stay_exact():
    """The fork-join pool behind grx_host_checksums / grx_download is re-armed thousands of times with CHANGING task
    counts (a straggler of call N must never run a task of call N + 1 twice: the hashes of a call would then be
    combined from the wrong parts or returned while a part is still being written).  Two Python threads hammer it
    with tables of different widths; every answer must equal the single-column answer computed up front."""
    import threading
    rng = np.random.default_rng(7)
    tables = [rng.random((w, 40_000 + 64 * w)) for w in (1, 2, 3, 5, 8, 13)]
    expect = [np.array([_sums(t[j:j + 1])[0] for j in range(t.shape[0])], dtype=np.uint64) for t in tables]
    errors = []

    def hammer(offset):
        try:
            for it in range(400):
                k = (it + offset) % len(tables)
                if not np.array_equal(_sums(tables[k]), expect[k]):
                    errors.append((offset, it, k))
                    return
        except Exception as exc:                                  # pragma: no cover
            errors.append(repr(exc))

    threads = [threading.Thread(target=hammer, args=(o,)) for o in (0, 3)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors[:3]
