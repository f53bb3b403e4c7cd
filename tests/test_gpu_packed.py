"""
-m gpu: neighbour aggregation from bit-packed integer rows (grx_column_bits / grx_pack_fields / grx_aggregate_packed)
is bit-identical to grx_aggregate on the fp64 columns it replaces -- generation-1 shape (integer summands only, 8-byte
rows) and generation-2 shape (sums S and means fl(S / d) rebuilt in registers, 16-byte rows) -- on graphs with hubs
(rows of more than 128 and of more than 8192 neighbours), dangling nodes and ragged row ranges.
"""
import time

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _setup(G):
    from graphrole_amd import RecursiveFeatureExtractor
    fe = RecursiveFeatureExtractor(G, max_generations=2)
    names, cols, _ = fe.graph.neighborhood_feature_columns()
    dev = fe.graph._device_graph()[1]
    return fe, dev, names, cols


def _bits(K, cols, n):
    block = K.gather_columns(cols, n)
    return [int(b) for b in K.to_host(K.column_bits(block, n, (1 << len(cols)) - 1))]


def _check_generation_shapes(G, row_ranges=((0, None),)):
    from graphrole_amd import kernels as K
    fe, dev, names, cols = _setup(G)
    n = dev.n
    # ---- generation 1: three integer columns, sums and means of integers
    bits0 = _bits(K, cols, n)
    host_max = [int(K.to_host(c)[:n].max()) for c in cols]
    assert bits0 == [max(m.bit_length(), 1) for m in host_max]
    L1, rb1 = K.packed_layout(bits0, 0, list(range(len(cols))), [False] * len(cols))
    assert rb1 in (8, 16)
    rows1 = K.pack_fields(dev, L1, rb1, cols)
    fp_rows, ldr = K.pack_rows(cols, n)
    for b, e in row_ranges:
        ref = K.aggregate(dev, fp_rows, len(cols), ldr, b, e)
        got = K.aggregate_packed(dev, L1, rows1, b, e)
        e = n if e is None else e
        assert torch.equal(got[:, b:e], ref[:, b:e])
    # ---- generation 2: parents = sums S_k and means fl(S_k / d) of generation 1
    f = len(cols)
    gen1 = K.aggregate(dev, fp_rows, f, ldr)
    sums = [gen1[j] for j in range(f)]
    means = [gen1[f + j] for j in range(f)]
    bits1 = _bits(K, sums, n)
    deg_bits = max(int(np.diff(K.to_host(dev.row_ptr)).max()).bit_length(), 1)
    out_field = [0, f - 1] + list(range(f))
    out_is_mean = [False, False] + [True] * f
    L2, rb2 = K.packed_layout(bits1, deg_bits, out_field, out_is_mean)
    assert rb2 in (8, 16), (bits1, deg_bits)
    rows2 = K.pack_fields(dev, L2, rb2, sums)
    parents = [sums[0], sums[f - 1]] + means
    fp2, ldr2 = K.pack_rows(parents, n)
    for b, e in row_ranges:
        ref = K.aggregate(dev, fp2, len(parents), ldr2, b, e)
        got = K.aggregate_packed(dev, L2, rows2, b, e)
        e = n if e is None else e
        assert torch.equal(got[:, b:e], ref[:, b:e])
    return rb1, rb2


def test_packed_equals_fp64_on_a_power_law_graph():
    from graphrole_amd import synth
    G = synth.ba_graph(60_000, 8, seed=3)
    rb1, rb2 = _check_generation_shapes(G, row_ranges=((0, None), (0, 777), (12_345, 40_001), (59_990, None)))
    assert rb1 == 8 and rb2 == 16


def test_packed_equals_fp64_with_a_very_long_row_and_dangling_nodes():
    """a star of 20 000 leaves (one row beyond numpy's 8192-element chunk), a clique, isolated-from-the-rest leaves"""
    from graphrole_amd.graph.csr import CSRGraph
    rng = np.random.default_rng(5)
    n = 30_000
    src = [np.zeros(20_000, dtype=np.int64)]
    dst = [np.arange(1, 20_001, dtype=np.int64)]
    a = rng.integers(1, n, 60_000)
    b = rng.integers(1, n, 60_000)
    keep = a != b
    key = np.unique(np.minimum(a[keep], b[keep]) * n + np.maximum(a[keep], b[keep]))
    src.append(key // n)
    dst.append(key % n)
    s, d = np.concatenate(src), np.concatenate(dst)
    lo, hi = np.minimum(s, d), np.maximum(s, d)
    uniq = np.unique(lo * n + hi)
    G = CSRGraph(n, uniq // n, uniq % n, validate=False)
    _check_generation_shapes(G, row_ranges=((0, None), (0, 1), (1, 5000)))


@pytest.mark.parametrize('n_out', [1, 2, 4, 7, 8])
def test_every_output_count_and_both_row_widths(n_out):
    """random integer tables: F = 1 .. 8 outputs, 8- and 16-byte rows, means and sums mixed"""
    from graphrole_amd import kernels as K, synth
    G = synth.er_graph(20_000, 150_000, seed=n_out)
    _, dev, _, _ = _setup(G)
    n = dev.n
    rng = np.random.RandomState(n_out)
    for widths in ([9, 7, 11], [30, 28, 25, 20], [17] * 6):
        cols = [K.to_device(rng.randint(0, 2 ** w, size=n).astype(np.float64)) for w in widths]
        deg_bits = max(int(np.diff(K.to_host(dev.row_ptr)).max()).bit_length(), 1)
        out_field = [j % len(widths) for j in range(n_out)]
        out_is_mean = [bool((j // len(widths) + j) & 1) for j in range(n_out)]
        L, rb = K.packed_layout(widths, deg_bits, out_field, out_is_mean)
        assert rb in (8, 16)
        rows = K.pack_fields(dev, L, rb, cols)
        deg = K.to_host(dev.row_ptr)
        cnt = np.diff(deg).astype(np.float64)
        parents = []
        for k, m in zip(out_field, out_is_mean):
            v = K.to_host(cols[k])[:n]
            parents.append(K.to_device(np.where(cnt > 0, v / np.where(cnt > 0, cnt, 1.0), 0.0) if m else v))
        fp, ldr = K.pack_rows(parents, n)
        assert torch.equal(K.aggregate_packed(dev, L, rows), K.aggregate(dev, fp, n_out, ldr))


def test_layout_limits():
    from graphrole_amd import _lib, kernels as K
    assert K.packed_layout([20, 20, 20], 0, [0, 1, 2], [0, 0, 0])[1] == 8
    assert K.packed_layout([20, 20, 20], 14, [0, 1, 2], [0, 0, 1])[1] == 16
    assert K.packed_layout([62, 62], 10, [0, 1], [0, 0])[1] == 0          # 62 + 62 does not fit one word, + 10 not two
    assert K.packed_layout([40, 40, 40, 40], 0, [0], [0])[1] == 0         # 160 bits
    assert K.packed_layout([63], 0, [0], [0])[1] == 0


def test_packed_rows_are_faster_at_ba1m():
    """BA 1 M / 10 M (BASELINE config 3): the launches grx_refex_run issues for generations 1 and 2, old rows vs packed"""
    from graphrole_amd import kernels as K, synth
    import json
    import os
    G = synth.ba_graph(1_000_000, 10, seed=0)
    fe, dev, names, cols = _setup(G)
    n = dev.n
    f = len(cols)

    def timed(fn, reps=10):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e3

    bits0 = _bits(K, cols, n)
    L1, rb1 = K.packed_layout(bits0, 0, list(range(f)), [False] * f)
    rows1 = K.pack_fields(dev, L1, rb1, cols)
    irows, ldi = K.pack_rows_i32(cols, n)
    t_i32 = timed(lambda: K.aggregate_i32(dev, irows, f, ldi))
    t_p1 = timed(lambda: K.aggregate_packed(dev, L1, rows1))
    fp_rows, ldr = K.pack_rows(cols, n)
    gen1 = K.aggregate(dev, fp_rows, f, ldr)
    sums = [gen1[j] for j in range(f)]
    bits1 = _bits(K, sums, n)
    deg_bits = max(int(np.diff(K.to_host(dev.row_ptr)).max()).bit_length(), 1)
    # the five parents BA 1 M retains in generation 1: three sums, two means
    out_field, out_is_mean = [0, 1, 2, 0, 2], [False, False, False, True, True]
    L2, rb2 = K.packed_layout(bits1, deg_bits, out_field, out_is_mean)
    rows2 = K.pack_fields(dev, L2, rb2, sums)
    parents = [gen1[0], gen1[1], gen1[2], gen1[f + 0], gen1[f + 2]]
    fp2, ldr2 = K.pack_rows(parents, n)
    t_fp = timed(lambda: K.aggregate(dev, fp2, 5, ldr2))
    t_p2 = timed(lambda: K.aggregate_packed(dev, L2, rows2))
    assert torch.equal(K.aggregate_packed(dev, L2, rows2), K.aggregate(dev, fp2, 5, ldr2))
    rec = {'gen1_int32_rows_ms': t_i32, 'gen1_packed_ms': t_p1, 'gen1_row_bytes': rb1, 'gen1_bits': bits0,
           'gen2_fp64_rows_ms': t_fp, 'gen2_packed_ms': t_p2, 'gen2_row_bytes': rb2, 'gen2_bits': bits1, 'degree_bits': deg_bits}
    out_dir = os.path.join(os.environ.get('GRAFT_REPO_ROOT', '.'), 'gpurun_out')
    os.makedirs(out_dir, exist_ok=True)
    json.dump(rec, open(os.path.join(out_dir, 'packed_rows_ba1m.json'), 'w'), indent=1)
    print(rec)
    assert t_p1 < t_i32 and t_p2 < t_fp, rec
