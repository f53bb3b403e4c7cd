"""Prototype (numpy) of the multi-level threshold selection that replaces the full radix sort in
grx_vertical_log_bin: 8 levels of 8-bit histograms restricted to candidate prefixes, with interval
arithmetic over the unknown tie counts.  Prints candidate counts per level and checks the thresholds
against the sort-based walk (prune.py:33-54)."""
import sys
import numpy as np

sys.path.insert(0, 'tests')
sys.path.insert(0, '.')


def keys_of(x):
    b = (x + 0.0).view(np.uint64)
    neg = (b >> np.uint64(63)).astype(bool)
    return np.where(neg, ~b, b ^ np.uint64(0x8000000000000000))


def walk_sorted(s, frac):
    n = len(s)
    done, thr = 0, []
    while done < n:
        size = max(int(frac * (n - done)), 1)
        hi = s[done + size - 1]
        done = int(np.searchsorted(s, hi, side='right'))
        thr.append(hi)
    return thr


def T(b, n, frac):
    if b >= n:
        return n
    return b + max(int(frac * (n - b)), 1)


def select(keys, frac, cap=10**9):
    n = len(keys)
    cands = [(0, 0)]                      # (prefix, base): keys with smaller prefix
    counts_per_level = []
    for level in range(8):
        shift_p = 64 - 8 * level
        shift_d = 56 - 8 * level
        pref = (keys >> np.uint64(shift_p)) if level else np.zeros(n, np.uint64)
        dig = ((keys >> np.uint64(shift_d)) & np.uint64(255)).astype(np.int64)
        # sub-bucket table: list of (prefix', start, count) in key order
        subs = []
        for p, base in cands:
            m = pref == np.uint64(p)
            h = np.bincount(dig[m], minlength=256)
            st = base + np.concatenate([[0], np.cumsum(h)[:-1]])
            for d in np.flatnonzero(h):
                subs.append(((p << 8) | int(d), int(st[d]), int(h[d])))
        starts = np.array([s[1] for s in subs])
        ends = np.array([s[1] + s[2] for s in subs])

        def locate(r):                    # sub-bucket holding 1-based rank r
            i = int(np.searchsorted(ends, r, side='left'))
            assert i < len(subs) and starts[i] < r <= ends[i], (level, r)
            return i
        marked = set()
        lo = hi = 0
        thr = []
        steps = 0
        while lo < n and steps < 200:
            a, b = locate(T(lo, n, frac)), locate(T(hi, n, frac))
            marked.update(range(a, b + 1))
            lo, hi = max(T(lo, n, frac), starts[a] + 1), int(ends[b])
            if level == 7:
                assert a == b
                lo = hi = int(ends[a])
                thr.append(subs[a][0])
            steps += 1
        counts_per_level.append(len(marked))
        cands = [(subs[i][0], subs[i][1]) for i in sorted(marked)]
    return thr, counts_per_level


def main():
    import util
    from oracle import refex, ckernels
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
    src, dst, _ = util.powerlaw_graph(n, 10, 0)
    g = refex.graph_from_arrays(n, src, dst, None, False)
    names, X = refex.neighborhood_features(g, True)
    cols = [X[:, j].astype(np.float64) for j in range(X.shape[1])]
    cur = X.astype(np.float64)
    for gen in range(2):
        S, M = ckernels.aggregate(g.row_ptr, g.adj_col, np.ascontiguousarray(cur[:, :6]))
        cur = np.concatenate([S, M], axis=1)
        cols += [cur[:, j] for j in range(cur.shape[1])]
    rng = np.random.default_rng(0)
    cols.append(rng.random(n))
    cols.append(rng.standard_normal(n))
    cols.append(np.floor(rng.pareto(1.5, n)))
    cols.append(np.zeros(n))
    worst = np.zeros(8, int)
    for j, c in enumerate(cols):
        k = keys_of(c)
        ref = walk_sorted(np.sort(c), 0.5)
        thr, cnt = select(k, 0.5)
        back = np.array(thr, dtype=np.uint64)
        neg = (back >> np.uint64(63)).astype(bool)
        vals = np.where(neg, back ^ np.uint64(0x8000000000000000), ~back).view(np.float64)
        assert len(ref) == len(vals) and np.array_equal(np.array(ref), vals), j
        worst = np.maximum(worst, cnt)
        print(j, len(ref), cnt)
    print('worst', worst)


if __name__ == '__main__':
    main()
