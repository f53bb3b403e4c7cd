"""
-m gpu: kernel-level parity of the HIP path (through the libgrx.so C ABI) against the oracle and
the golden vectors produced by the reference.  Integer / index results are compared bit-exactly;
fp64 sums within RTOL (re-association of fp64 additions only -- stated in DESIGN.md).
"""
import os

import numpy as np
import pytest

from tests import util

pytestmark = pytest.mark.gpu

RTOL = 1e-12


@pytest.fixture(scope='module')
def K():
    import torch
    assert torch.cuda.is_available(), 'gpu tests need a GPU'
    from graphrole_amd import kernels
    return kernels


def _oracle_graph(n, src, dst, w, directed):
    from oracle import refex
    return refex.graph_from_arrays(n, src, dst, w, directed)


def _dev_csr(K, og, transpose=False):
    if transpose:
        return K.DeviceCSR(og.t_row_ptr, og.t_col, og.t_w)
    return K.DeviceCSR(og.row_ptr, og.col, og.w, agg_col=og.adj_col)


GRAPHS = [
    dict(n=50, m=120, seed=1, directed=False, weighted=False, self_loops=0),
    dict(n=300, m=2000, seed=2, directed=False, weighted=True, self_loops=5),
    dict(n=300, m=2500, seed=3, directed=True, weighted=True, self_loops=4),
    dict(n=1000, m=3000, seed=4, directed=True, weighted=False, self_loops=3),
    dict(n=5000, m=60000, seed=5, directed=False, weighted=False, self_loops=10),
]


@pytest.mark.parametrize('spec', GRAPHS)
def test_row_sums_and_egonet_vs_oracle(K, spec):
    from oracle import refex
    src, dst, w = util.random_graph(**spec)
    og = _oracle_graph(spec['n'], src, dst, w, spec['directed'])
    csr = _dev_csr(K, og)
    loc = refex.local_features_c(og)
    ego = refex.egonet_features_c(og)
    if spec['directed']:
        out = K.row_sums(csr, False).cpu().numpy()
        ind = K.row_sums(_dev_csr(K, og, True), False).cpu().numpy()
        np.testing.assert_allclose(out, loc['out_degree'], rtol=RTOL)
        np.testing.assert_allclose(ind, loc['in_degree'], rtol=RTOL)
    else:
        deg = K.row_sums(csr, True).cpu().numpy()
        np.testing.assert_allclose(deg, loc['degree'], rtol=RTOL)
    internal, external = K.egonet_features(csr, spec['directed'])
    np.testing.assert_allclose(internal.cpu().numpy(), ego['internal_edges'], rtol=RTOL, atol=0)
    got_ext = external.cpu().numpy()
    np.testing.assert_allclose(got_ext, ego['external_edges'], rtol=1e-11, atol=0)
    # exact zeros stay exact zeros (boundary-free ego-nets)
    assert np.array_equal(got_ext == 0, ego['external_edges'] == 0)
    if not spec['weighted']:
        assert np.array_equal(internal.cpu().numpy(), ego['internal_edges'])
        assert np.array_equal(got_ext, ego['external_edges'])
    if not spec['weighted'] and not spec['directed']:
        # K.egonet_features took the triangle-count fast path; the general gather kernel must agree
        i2, e2 = K.egonet_features_general(csr, False)
        assert np.array_equal(i2.cpu().numpy(), ego['internal_edges'])
        assert np.array_equal(e2.cpu().numpy(), ego['external_edges'])


def test_egonet_hub_rows_powerlaw(K):
    """Power-law graph: rows above the hub threshold go through the workgroup-per-node kernel."""
    from oracle import refex
    n, m = 30000, 6
    src, dst, _ = util.powerlaw_graph(n, m, seed=0)
    og = _oracle_graph(n, src, dst, None, False)
    assert np.diff(og.row_ptr).max() > 600
    csr = _dev_csr(K, og)
    ego = refex.egonet_features_c(og)
    internal, external = K.egonet_features(csr, False)                 # triangle-count fast path
    assert np.array_equal(internal.cpu().numpy(), ego['internal_edges'])
    assert np.array_equal(external.cpu().numpy(), ego['external_edges'])
    internal, external = K.egonet_features_general(csr, False)         # gather kernel, hub workgroups
    assert np.array_equal(internal.cpu().numpy(), ego['internal_edges'])
    assert np.array_equal(external.cpu().numpy(), ego['external_edges'])
    T = K.triangle_counts(csr).cpu().numpy()
    Ta = K.triangle_counts(csr, 0, 12345).cpu().numpy() + K.triangle_counts(csr, 12345, None).cpu().numpy()
    assert np.array_equal(T[:n], Ta[:n])                                # arc ranges add up (all-reduce SUM)
    # node-range slices reproduce the full result
    i2, e2 = K.egonet_features(csr, False, row_begin=1000, row_end=17000)
    assert np.array_equal(i2.cpu().numpy()[1000:17000], ego['internal_edges'][1000:17000])
    assert np.all(i2.cpu().numpy()[:1000] == 0) and np.all(e2.cpu().numpy()[17000:] == 0)


def _clique_graph(seed, directed, leak):
    """Disjoint cliques of 3 .. 31 nodes (every ego-net closed: external exactly 0) with weights over twelve decades;
    `leak` adds a few arcs of tiny weight between cliques -- rows whose boundary weight is 1e-13 of their row sum, the
    case in which rowsum - matched would be rounding noise and the kernel has to add the leaving arcs one by one."""
    rng = np.random.default_rng(seed)
    src, dst, w = [], [], []
    start, firsts = 0, []
    for k in list(rng.integers(3, 32, size=60)) + [2, 2, 33, 40]:
        ids = np.arange(start, start + k)
        firsts.append((start, int(k)))
        a, b = np.meshgrid(ids, ids, indexing='ij')
        keep = (a != b) if directed else (a < b)
        src.append(a[keep]); dst.append(b[keep])
        w.append(10.0 ** rng.uniform(-6, 6, size=int(keep.sum())))
        start += k
    n = start
    src, dst, w = np.concatenate(src), np.concatenate(dst), np.concatenate(w)
    if leak:
        ls = np.array([firsts[i][0] for i in range(0, 40, 2)])
        ld = np.array([firsts[i + 1][0] + 1 for i in range(0, 40, 2)])
        src, dst = np.concatenate([src, ls]), np.concatenate([dst, ld])
        w = np.concatenate([w, np.full(len(ls), 1e-7)])
    return n, src, dst, w


@pytest.mark.parametrize('directed', [False, True])
@pytest.mark.parametrize('leak', [False, True])
def test_egonet_closed_rows_and_wide_weight_range(K, directed, leak):
    """Round 5: the ego-net kernel reads weights for MATCHED arcs only and takes what leaves the ego set as
    rowsum - matched; closed rows must give exactly 0, and nearly closed rows must not lose their digits."""
    from oracle import refex
    n, src, dst, w = _clique_graph(11 + directed, directed, leak)
    og = _oracle_graph(n, src, dst, w, directed)
    csr = _dev_csr(K, og)
    ego = refex.egonet_features_c(og)
    internal, external = K.egonet_features(csr, directed)
    np.testing.assert_allclose(internal.cpu().numpy(), ego['internal_edges'], rtol=RTOL, atol=0)
    got = external.cpu().numpy()
    assert np.array_equal(got == 0, ego['external_edges'] == 0)
    np.testing.assert_allclose(got, ego['external_edges'], rtol=util.WEIGHTED_RTOL, atol=0)
    if not leak:
        assert not got.any()
    # row ranges reproduce the full result bit for bit (no dependence on the launch geometry)
    i2, e2 = K.egonet_features(csr, directed, row_begin=n // 3, row_end=n - 7)
    assert np.array_equal(i2.cpu().numpy()[n // 3:n - 7], internal.cpu().numpy()[n // 3:n - 7])
    assert np.array_equal(e2.cpu().numpy()[n // 3:n - 7], got[n // 3:n - 7])


def test_egonet_member_rows_of_every_length(K):
    """Directed weighted graph whose member rows span 0 .. 700 arcs: the prefetched four chunks, the chunk loop
    beyond them, the binary-search branch for hub members, and the wavefront / workgroup kernels for long ego sets
    (their rows come from the lists of egonet_prepare_kernel)."""
    from oracle import refex
    rng = np.random.default_rng(5)
    n = 4000
    deg = np.concatenate([np.arange(0, 80), rng.integers(0, 40, size=n - 80 - 9), [300, 511, 512, 513, 600, 700,
                                                                                    1024, 2500, 3900]])     # hub rows in 1 - 4 parts
    src = np.repeat(np.arange(n), deg)
    dst = np.concatenate([rng.choice(n, size=int(d), replace=False) for d in deg]) if deg.sum() else np.zeros(0, int)
    # make the hubs popular members, and close a few triangles
    extra_s = rng.integers(0, n, size=3000)
    extra_d = rng.choice(np.arange(n - 9, n), size=3000)
    key = np.unique(np.concatenate([src * n + dst, extra_s * n + extra_d]))
    src, dst = key // n, key % n
    w = rng.uniform(0.1, 5.0, size=len(src))
    og = _oracle_graph(n, src, dst, w, True)
    csr = _dev_csr(K, og)
    ego = refex.egonet_features_c(og)
    internal, external = K.egonet_features(csr, True)
    np.testing.assert_allclose(internal.cpu().numpy(), ego['internal_edges'], rtol=RTOL, atol=0)
    np.testing.assert_allclose(external.cpu().numpy(), ego['external_edges'], rtol=util.WEIGHTED_RTOL, atol=0)
    # the same rows without weights: exact integers
    og1 = _oracle_graph(n, src, dst, None, True)
    ego1 = refex.egonet_features_c(og1)
    i1, e1 = K.egonet_features(_dev_csr(K, og1), True)
    assert np.array_equal(i1.cpu().numpy(), ego1['internal_edges'])
    assert np.array_equal(e1.cpu().numpy(), ego1['external_edges'])
    # the same edges as an UNDIRECTED weighted graph (every member's row holds the arc back: the undirected fast path of
    # the group kernel, hub rows whose members are hubs themselves)
    lo, hi = np.minimum(src, dst), np.maximum(src, dst)
    key2, first = np.unique(lo * n + hi, return_index=True)
    og2 = _oracle_graph(n, key2 // n, key2 % n, w[first], False)
    ego2 = refex.egonet_features_c(og2)
    i2, e2 = K.egonet_features(_dev_csr(K, og2), False)
    np.testing.assert_allclose(i2.cpu().numpy(), ego2['internal_edges'], rtol=RTOL, atol=0)
    np.testing.assert_allclose(e2.cpu().numpy(), ego2['external_edges'], rtol=util.WEIGHTED_RTOL, atol=0)
    assert np.array_equal(e2.cpu().numpy() == 0, ego2['external_edges'] == 0)


@pytest.mark.parametrize('n', [1, 63, 64, 65, 4099, 100003])
def test_pack_rows_of_64_bytes(K, n):
    """5 - 8 columns become rows of 8 doubles (pack_rows8_kernel: 64 rows x 8 columns through the wave's LDS slice):
    values in place, pad columns zero, row counts that are no multiple of the 64-row block."""
    import torch
    rng = np.random.default_rng(n)
    for f in (5, 6, 7, 8):
        X = rng.standard_normal((n, f))
        Xd = torch.from_numpy(np.ascontiguousarray(X.T)).cuda()
        rows, ldr = K.pack_rows([Xd[c] for c in range(f)], n)
        assert ldr == 8
        got = rows.cpu().numpy()
        assert np.array_equal(got[:n, :f], X)
        assert not got[:n, f:].any()


def test_triangle_counts_long_oriented_lists(K):
    """The per-arc table of the oriented graph keeps list lengths in 10-bit fields; a clique of 1 200 nodes (oriented
    out-degrees 0 .. 1199) saturates them and takes the row-pointer path, lists of 17 .. 1022 ids take the chunked
    group path.  T(v) of a clique is C(n - 1, 2); a sparse random part around it keeps the ordinary path busy.  Both
    builders of the table (device and host) must agree, and source-row ranges must add up."""
    import torch
    rng = np.random.default_rng(7)
    nc, n = 1200, 4000
    iu = np.triu_indices(nc, 1)
    extra_s = rng.integers(0, n, 20000)
    extra_d = rng.integers(nc, n, 20000)                                  # never inside the clique
    keep = extra_s != extra_d
    src = np.concatenate([iu[0], extra_s[keep]]).astype(np.int64)
    dst = np.concatenate([iu[1], extra_d[keep]]).astype(np.int64)
    key = np.unique(np.minimum(src, dst) * n + np.maximum(src, dst))
    src, dst = key // n, key % n
    og = _oracle_graph(n, src, dst, None, False)
    csr = _dev_csr(K, og)
    T = K.triangle_counts(csr).cpu().numpy()[:n]
    # reference: triangles through v from the dense adjacency
    A = np.zeros((n, n), dtype=np.float32)
    A[src, dst] = 1; A[dst, src] = 1
    At = torch.from_numpy(A).cuda()
    exp = ((At @ At) * At).sum(1).cpu().numpy().astype(np.int64) // 2
    assert exp[:nc].min() >= (nc - 1) * (nc - 2) // 2
    assert np.array_equal(T.astype(np.int64), exp)
    cut = 700
    Ta = K.triangle_counts(csr, 0, cut).cpu().numpy() + K.triangle_counts(csr, cut, None).cpu().numpy()
    assert np.array_equal(Ta[:n], T)
    # the device-built table equals the host-built one
    dev = K.DeviceCSR.from_device(csr.row_ptr, csr.col, None, None, og.row_ptr) if hasattr(K.DeviceCSR, 'from_device') else None
    if dev is not None:
        o = dev.oriented()
        ref = csr.oriented()
        o_nnz = int(np.asarray(og.row_ptr)[-1]) // 2
        assert np.array_equal(o.arc.cpu().numpy()[:o_nnz], ref.arc.cpu().numpy()[:o_nnz])


def test_triangle_counts_concentrated_on_hubs(K):
    """Every corner of every triangle is an atomic increment of T[vertex], and atomics to one address queue up: the
    counters of the first 256 vertices of the degree-descending order are collected per workgroup in LDS
    (csrc/grx_graph.hip, triangle_count_arcs_kernel) and added once per workgroup.  Graphs that send almost every
    increment to a handful of hubs, with counts known in closed form:
      * h hubs joined to each other and to every one of m leaves (no leaf-leaf edges): a leaf closes C(h, 2) triangles,
        a hub (h - 1) m + C(h - 1, 2) -- hub counters far beyond 2^16, leaves beyond index 256 untouched by LDS;
      * a wheel: the hub closes m triangles, every rim vertex 2;
    and source-row ranges (the sharded path) must add up to the whole."""
    for h, m in ((2, 150_000), (5, 60_000), (40, 3_000)):
        n = h + m
        hubs = np.arange(h)
        leaves = np.arange(h, n)
        hs, hd = np.triu_indices(h, 1)
        src = np.concatenate([hs, np.repeat(hubs, m)]).astype(np.int64)
        dst = np.concatenate([hd, np.tile(leaves, h)]).astype(np.int64)
        og = _oracle_graph(n, src, dst, None, False)
        csr = _dev_csr(K, og)
        T = K.triangle_counts(csr).cpu().numpy()[:n].astype(np.int64)
        exp_hub = (h - 1) * m + (h - 1) * (h - 2) // 2
        exp_leaf = h * (h - 1) // 2
        # (rows stay in label order here: the hubs are rows 0 .. h - 1, and the rows of largest degree)
        assert np.array_equal(T[:h], np.full(h, exp_hub)), (h, m)
        assert np.array_equal(T[h:], np.full(m, exp_leaf)), (h, m)
        cut = n // 3
        Ta = K.triangle_counts(csr, 0, cut).cpu().numpy() + K.triangle_counts(csr, cut, None).cpu().numpy()
        assert np.array_equal(Ta[:n].astype(np.int64), T)
    m = 200_000                                                           # wheel: hub 0, rim 1 .. m
    rim = np.arange(1, m + 1)
    src = np.concatenate([np.zeros(m, dtype=np.int64), rim])
    dst = np.concatenate([rim, np.roll(rim, -1)])
    og = _oracle_graph(m + 1, src, dst, None, False)
    csr = _dev_csr(K, og)
    T = K.triangle_counts(csr).cpu().numpy()[:m + 1].astype(np.int64)
    assert T[0] == m and np.array_equal(T[1:], np.full(m, 2))


@pytest.mark.parametrize('name', ['karate', 'karate_weighted', 'dw200_attrs', 'loops_dangling150', 'directed120',
                                  'iface7', 'iface7_dw', 'path4', 'er2000', 'ba2000'])
def test_gen0_vs_reference_golden(K, name):
    g = util.load_refex(name)
    og = util.oracle_graph_from_golden(g)
    names = g.js('gen0_names')
    vals = g['gen0_values']
    csr = _dev_csr(K, og)
    got = {}
    if og.directed:
        got['out_degree'] = K.row_sums(csr, False).cpu().numpy()
        got['in_degree'] = K.row_sums(_dev_csr(K, og, True), False).cpu().numpy()
        got['total_degree'] = got['out_degree'] + got['in_degree']
    else:
        got['degree'] = K.row_sums(csr, True).cpu().numpy()
    i, e = K.egonet_features(csr, og.directed)
    got['internal_edges'], got['external_edges'] = i.cpu().numpy(), e.cpu().numpy()
    for j, nm in enumerate(names):
        if nm.startswith('attribute_'):
            continue
        np.testing.assert_allclose(got[nm], vals[:, j], rtol=1e-12, atol=0, err_msg=f'{name}:{nm}')
        assert np.array_equal(got[nm] == 0, vals[:, j] == 0), f'{name}:{nm} zero pattern'


@pytest.mark.parametrize('lanes', [None, 4, 8, 16, 32])
@pytest.mark.parametrize('f', [1, 2, 3, 4, 5, 7, 8, 13, 16, 17, 26, 40])
def test_aggregate_bit_exact_vs_oracle(K, f, lanes):
    """Sums and means equal the oracle's numpy-pairwise sums in adjacency order BIT FOR BIT, for
    every lane-group width, with rows of every length class (< 8, 8..128, > 128 = block tree)."""
    import torch
    from oracle import ckernels
    if lanes is not None and f not in (3, 5, 8, 17):
        pytest.skip('lane-group sweep on a few widths')
    n, m = 20000, 6
    src, dst, _ = util.powerlaw_graph(n, m, seed=f)
    og = _oracle_graph(n, src, dst, None, False)
    deg = np.diff(og.row_ptr)
    assert deg.max() > 300 and (deg < 8).any() and ((deg >= 8) & (deg <= 128)).any()
    # a few dangling nodes at the end
    og.row_ptr = np.concatenate([og.row_ptr, np.full(7, og.row_ptr[-1])])
    n2 = og.n
    X = np.abs(np.random.default_rng(f).standard_normal((n2, f))) * 10.0 ** np.arange(f).clip(0, 6)
    S, M = ckernels.aggregate(og.row_ptr, og.adj_col, X)
    csr = _dev_csr(K, og)
    if lanes is not None:
        csr.plan().set_lanes(lanes)
    assert csr.plan().n_long_rows == int((deg > 128).sum()) and csr.plan().n_blocks >= 2 * csr.plan().n_long_rows
    Xd = torch.from_numpy(np.ascontiguousarray(X.T)).cuda()
    rows, ldr = K.pack_rows([Xd[c] for c in range(f)], n2)
    assert np.array_equal(rows.cpu().numpy()[:, :f], X)
    blk = K.aggregate(csr, rows, f, ldr)
    got = blk.cpu().numpy()
    assert np.array_equal(got[:f].T, S), f'{int((got[:f].T != S).sum())} sums differ'
    assert np.array_equal(got[f:].T, M), f'{int((got[f:].T != M).sum())} means differ'
    assert np.all(got[:, n:] == 0)                              # dangling rows: sum 0, mean 0 (not NaN)
    # row-range slices agree bitwise with the full run
    blk3 = K.aggregate(csr, rows, f, ldr, row_begin=123, row_end=15001)
    assert torch.equal(blk3[:, 123:15001], blk[:, 123:15001])
    assert float(blk3[:, :123].abs().sum()) == 0.0 and float(blk3[:, 15001:].abs().sum()) == 0.0
    # only one of the outputs
    only_mean = K.aggregate(csr, rows, f, ldr, want_sum=False)
    assert torch.equal(only_mean[f:], blk[f:]) and float(only_mean[:f].abs().sum()) == 0.0


@pytest.mark.parametrize('lanes', [None, 4, 16, 32])
@pytest.mark.parametrize('f', [1, 2, 3, 4, 5, 8])
def test_aggregate_i32_bit_exact_vs_oracle(K, f, lanes):
    """The integer gather source (grx_aggregate_i32: 16- / 32-byte int32 rows, any summation order, int64
    accumulators) gives the oracle's pairwise-order fp64 sums and means bit for bit whenever the columns hold exact
    integers below 2^31 -- rows of every length class, blocks of long rows, row ranges, dangling rows."""
    import torch
    from oracle import ckernels
    if lanes is not None and f not in (3, 5):
        pytest.skip('lane-group sweep on two widths')
    n, m = 20000, 6
    src, dst, _ = util.powerlaw_graph(n, m, seed=40 + f)
    og = _oracle_graph(n, src, dst, None, False)
    deg = np.diff(og.row_ptr)
    assert deg.max() > 300 and (deg < 8).any()
    og.row_ptr = np.concatenate([og.row_ptr, np.full(5, og.row_ptr[-1])])
    n2 = og.n
    rng = np.random.default_rng(f)
    X = rng.integers(0, 2 ** 31, size=(n2, f)).astype(np.float64)
    X[:, 0] = rng.integers(0, 50, size=n2)                       # a degree-like column next to full-range ones
    S, M = ckernels.aggregate(og.row_ptr, og.adj_col, X)
    csr = _dev_csr(K, og)
    if lanes is not None:
        csr.plan().set_lanes(lanes)
    assert K.aggregate_i32_ok(csr, f)
    Xd = torch.from_numpy(np.ascontiguousarray(X.T)).cuda()
    rows, ldi = K.pack_rows_i32([Xd[c] for c in range(f)], n2)
    assert ldi == (4 if f <= 4 else 8) and np.array_equal(rows.cpu().numpy()[:, :f], X.astype(np.int32))
    got = K.aggregate_i32(csr, rows, f, ldi).cpu().numpy()
    assert np.array_equal(got[:f].T, S), f'{int((got[:f].T != S).sum())} sums differ'
    assert np.array_equal(got[f:].T, M), f'{int((got[f:].T != M).sum())} means differ'
    part = K.aggregate_i32(csr, rows, f, ldi, row_begin=77, row_end=12345).cpu().numpy()
    assert np.array_equal(part[:, 77:12345], got[:, 77:12345])
    # and it is what the fp64 kernel computes
    rows64, ldr = K.pack_rows([Xd[c] for c in range(f)], n2)
    assert torch.equal(K.aggregate(csr, rows64, f, ldr), torch.from_numpy(got).cuda())
    assert not K.aggregate_i32_ok(csr, 9)


def test_aggregate_i32_on_huge_row(K):
    """70 001 neighbours of full-range int32 values: 1 024 integer block sums, the total near 2^47."""
    import torch
    n = 70002
    src = np.zeros(n - 1, dtype=np.int64)
    dst = np.arange(1, n, dtype=np.int64)
    og = _oracle_graph(n, src, dst, None, False)
    csr = _dev_csr(K, og)
    x = np.random.default_rng(9).integers(2 ** 30, 2 ** 31, size=n).astype(np.float64)
    rows, ldi = K.pack_rows_i32([torch.from_numpy(x).cuda()], n)
    got = K.aggregate_i32(csr, rows, 1, ldi).cpu().numpy()
    total = int(x[1:].astype(np.int64).sum())
    assert got[0, 0] == float(total) and got[1, 0] == float(total) / (n - 1)
    assert np.array_equal(got[0, 1:], np.full(n - 1, x[0]))


def test_aggregate_equals_numpy_sum_on_huge_row(K):
    """A star: the centre has 70 001 neighbours -> 1 024 blocks, 10 tree levels; the device sum
    must equal ndarray.sum() of the neighbour values in adjacency order."""
    import torch
    n = 70002
    src = np.zeros(n - 1, dtype=np.int64)
    dst = np.arange(1, n, dtype=np.int64)
    rng = np.random.default_rng(5)
    perm = rng.permutation(n - 1)
    og = _oracle_graph(n, src[perm], dst[perm], None, False)
    csr = _dev_csr(K, og)
    x = rng.random(n) ** 4 * 1e3
    col = torch.from_numpy(x).cuda()
    rows, ldr = K.pack_rows([col], n)
    got = K.aggregate(csr, rows, 1, ldr).cpu().numpy()
    nb = og.adj_row(0)
    assert np.array_equal(nb, dst[perm])                        # adjacency order = order of appearance
    assert got[0, 0] == np.ascontiguousarray(x[nb]).sum()
    assert got[1, 0] == np.ascontiguousarray(x[nb]).sum() / len(nb)
    assert np.array_equal(got[0, 1:], np.full(n - 1, x[0]))


@pytest.mark.parametrize('lanes', [None, 4, 16])
@pytest.mark.parametrize('f', [1, 2, 3, 5, 8, 13, 17, 26])
def test_aggregate_var_bit_exact_vs_oracle(K, f, lanes):
    """var / std = pandas' nanvar (ddof 1): the squared deviations from the neighbour mean are summed
    with the same numpy-pairwise order as the sums; rows with < 2 neighbours give NaN -> 0."""
    import torch
    from oracle import ckernels
    if lanes is not None and f not in (3, 8, 17):
        pytest.skip('lane-group sweep on a few widths')
    n, m = 20000, 1 if f == 2 else 6
    src, dst, _ = util.powerlaw_graph(n, m, seed=100 + f)
    og = _oracle_graph(n, src, dst, None, False)
    deg = np.diff(og.row_ptr)
    assert deg.max() > (300 if m == 6 else 64) and (m == 6 or (deg == 1).any())
    og.row_ptr = np.concatenate([og.row_ptr, np.full(3, og.row_ptr[-1])])
    n2 = og.n
    X = np.random.default_rng(f).standard_normal((n2, f)) * 10.0 ** np.arange(f).clip(0, 6)
    V, S = ckernels.aggregate_var(og.row_ptr, og.adj_col, X)
    csr = _dev_csr(K, og)
    if lanes is not None:
        csr.plan().set_lanes(lanes)
    Xd = torch.from_numpy(np.ascontiguousarray(X.T)).cuda()
    rows, ldr = K.pack_rows([Xd[c] for c in range(f)], n2)
    mean = K.aggregate(csr, rows, f, ldr, want_sum=False)[f:].contiguous()
    got = K.aggregate_var(csr, rows, f, ldr, mean).cpu().numpy()
    assert np.array_equal(got[:f].T, V), f'{int((got[:f].T != V).sum())} variances differ'
    assert np.array_equal(got[f:].T, S), f'{int((got[f:].T != S).sum())} stds differ'
    one = np.flatnonzero(np.diff(og.row_ptr) < 2)
    assert not got[:, one].any()
    part = K.aggregate_var(csr, rows, f, ldr, mean, row_begin=50, row_end=7000, want_var=False).cpu().numpy()
    assert np.array_equal(part[f:, 50:7000], got[f:, 50:7000]) and not part[:f].any() and not part[f:, 7000:].any()


def test_aggregate_var_equals_pandas_on_huge_row(K):
    """70 001 neighbours: block tree + 8192-element chunks, against DataFrame.var() itself."""
    import pandas as pd
    import torch
    n = 70002
    src = np.zeros(n - 1, dtype=np.int64)
    dst = np.arange(1, n, dtype=np.int64)
    rng = np.random.default_rng(6)
    perm = rng.permutation(n - 1)
    og = _oracle_graph(n, src[perm], dst[perm], None, False)
    csr = _dev_csr(K, og)
    x = rng.random((n, 2)) ** 3 * 1e2
    Xd = torch.from_numpy(np.ascontiguousarray(x.T)).cuda()
    rows, ldr = K.pack_rows([Xd[0], Xd[1]], n)
    mean = K.aggregate(csr, rows, 2, ldr, want_sum=False)[2:].contiguous()
    got = K.aggregate_var(csr, rows, 2, ldr, mean).cpu().numpy()
    frame = pd.DataFrame(x[og.adj_row(0)])
    assert np.array_equal(got[:2, 0], frame.var().to_numpy())
    assert np.array_equal(got[2:, 0], frame.std().to_numpy())
    assert not got[:, 1:].any()                                 # leaves have one neighbour


@pytest.mark.parametrize('f', [1, 3, 8, 20])
def test_aggregate_prod_bit_exact_vs_oracle(K, f):
    """agg 'prod': left-to-right product in adjacency order (np.multiply.reduce), 1 for no neighbours."""
    import torch
    from oracle import ckernels
    n, m = 20000, 6
    src, dst, _ = util.powerlaw_graph(n, m, seed=200 + f)
    og = _oracle_graph(n, src, dst, None, False)
    og.row_ptr = np.concatenate([og.row_ptr, np.full(4, og.row_ptr[-1])])
    n2 = og.n
    X = np.random.default_rng(f).uniform(0.7, 1.4, (n2, f))
    X[:, 0] *= np.where(np.random.default_rng(1).random(n2) < 0.3, -1.0, 1.0)      # signs too
    P = ckernels.aggregate_prod(og.row_ptr, og.adj_col, X)
    csr = _dev_csr(K, og)
    Xd = torch.from_numpy(np.ascontiguousarray(X.T)).cuda()
    rows, ldr = K.pack_rows([Xd[c] for c in range(f)], n2)
    got = K.aggregate_prod(csr, rows, f, ldr).cpu().numpy()
    assert np.array_equal(got.T, P)
    assert np.all(got[:, n:] == 1.0)
    part = K.aggregate_prod(csr, rows, f, ldr, row_begin=10, row_end=5000).cpu().numpy()
    assert np.array_equal(part[:, 10:5000], got[:, 10:5000])          # only the rows of the range are written


@pytest.mark.parametrize('f', [1, 3, 6, 8, 20])
def test_aggregate_minmax_vs_oracle(K, f):
    import torch
    from oracle import ckernels
    n, m = 20000, 6
    src, dst, _ = util.powerlaw_graph(n, m, seed=40 + f)
    og = _oracle_graph(n, src, dst, None, False)
    og.row_ptr = np.concatenate([og.row_ptr, np.full(5, og.row_ptr[-1])])
    n2 = og.n
    X = np.random.default_rng(f).standard_normal((n2, f)) * 10.0 ** np.arange(f).clip(0, 6)   # negatives too
    lo, hi = ckernels.aggregate_minmax(og.row_ptr, og.adj_col, X)
    csr = _dev_csr(K, og)
    Xd = torch.from_numpy(np.ascontiguousarray(X.T)).cuda()
    rows, ldr = K.pack_rows([Xd[c] for c in range(f)], n2)
    got = K.aggregate_minmax(csr, rows, f, ldr).cpu().numpy()
    assert np.array_equal(got[:f].T, lo) and np.array_equal(got[f:].T, hi)
    assert np.all(got[:, n:] == 0)                              # no neighbours -> NaN -> 0
    part = K.aggregate_minmax(csr, rows, f, ldr, row_begin=77, row_end=9000, want_min=False).cpu().numpy()
    assert np.array_equal(part[f:, 77:9000], got[f:, 77:9000]) and not part[:f].any() and not part[f:, 9000:].any()


def test_aggregate_equal_columns_give_equal_outputs(K):
    """Pruning relies on it: identical input columns -> bitwise identical aggregated columns."""
    import torch
    n, m = 10000, 8
    src, dst, _ = util.powerlaw_graph(n, m, seed=3)
    og = _oracle_graph(n, src, dst, None, False)
    csr = _dev_csr(K, og)
    base = torch.rand(n, dtype=torch.float64, device='cuda') * 3.7
    other = torch.rand(n, dtype=torch.float64, device='cuda')
    rows, ldr = K.pack_rows([base, other, base.clone(), other, base], n)
    blk = K.aggregate(csr, rows, 5, ldr)
    for a, b in [(0, 2), (0, 4), (1, 3), (5, 7), (5, 9), (6, 8)]:
        assert torch.equal(blk[a], blk[b])


def test_aggregate_integer_columns_exact(K):
    import torch
    from oracle import ckernels
    src, dst, w = util.random_graph(3000, 40000, seed=9)
    og = _oracle_graph(3000, src, dst, None, False)
    csr = _dev_csr(K, og)
    X = np.random.default_rng(1).integers(0, 1000, size=(3000, 3)).astype(np.float64)
    S, M = ckernels.aggregate(og.row_ptr, og.adj_col, X)
    Xd = torch.from_numpy(np.ascontiguousarray(X.T)).cuda()
    rows, ldr = K.pack_rows([Xd[c] for c in range(3)], 3000)
    got = K.aggregate(csr, rows, 3, ldr).cpu().numpy()
    assert np.array_equal(got[:3].T, S)          # integer sums are exact in any order
    assert np.array_equal(got[3:].T, M)          # and so are their IEEE quotients


@pytest.mark.parametrize('n', [1, 2, 63, 64, 65, 4095, 4096, 4097, 100003, 1 << 20])
def test_sort_columns_exact(K, n):
    import torch
    rng = np.random.default_rng(n)
    cols = [
        rng.standard_normal(n) * 1e3,
        rng.integers(-5, 6, size=n).astype(np.float64),
        np.abs(rng.standard_normal(n)) * 10.0 ** rng.integers(-300, 300, size=n),
        np.where(rng.random(n) < 0.3, 0.0, rng.random(n)),
    ]
    cols[3][::7] = -0.0
    if n > 10:
        cols[2][3] = np.inf
        cols[2][5] = -np.inf
        cols[0][1] = 5e-324
    block = torch.from_numpy(np.stack(cols)).cuda()
    got = K.sort_columns(block).cpu().numpy()
    for j, c in enumerate(cols):
        exp = np.sort(c)
        assert np.array_equal(got[j], exp), f'col {j}'


REFERENCE_BINNING_TABLE = [   # /root/reference/tests/test_features/test_prune.py:17-85
    ([0], 0.5, [0]),
    ([1], 0.5, [0]),
    ([1, 1], 0.5, [0, 0]),
    ([1, 2], 0.5, [0, 1]),
    ([1, 2, 1], 0.5, [0, 1, 0]),
    ([1, 2, 2], 0.5, [0, 1, 1]),
    ([-1, 0, 0], 0.5, [0, 1, 1]),
    ([1, 2, 3, 4], 0.5, [0, 0, 1, 2]),
    ([1, 2, 3, 4, 5], 0.5, [0, 0, 1, 2, 3]),
    ([1, 2, 3, 4, 5, 6], 0.5, [0, 0, 0, 1, 2, 3]),
    (list(range(10)), 0.5, [0, 0, 0, 0, 0, 1, 1, 2, 3, 4]),
    ([-x for x in range(10)], 0.5, [0, 0, 0, 0, 0, 1, 1, 2, 3, 4][::-1]),
    ([-0.1 * x for x in range(10)], 0.5, [0, 0, 0, 0, 0, 1, 1, 2, 3, 4][::-1]),
    (list(range(10)), 0.1, list(range(10))),
    (list(range(10)), 0.25, [0, 0, 1, 1, 2, 3, 4, 5, 6, 7]),
]


def test_log_bin_reference_known_answers(K):
    import torch
    for arr, frac, expected in REFERENCE_BINNING_TABLE:
        block = torch.tensor([arr], dtype=torch.float64, device='cuda')
        bins, nb = K.vertical_log_bin(block, frac)
        assert bins.cpu().numpy()[0].tolist() == expected, (arr, frac)
        assert int(nb[0]) == max(expected) + 1
    # empty input -> empty output (test_prune.py:18-21)
    bins, _ = K.vertical_log_bin(torch.zeros((1, 0), dtype=torch.float64, device='cuda'))
    assert bins.shape == (1, 0)


def test_log_bin_bad_frac_raises_value_error(K):
    import torch
    block = torch.ones((1, 4), dtype=torch.float64, device='cuda')
    for frac in (0.0, 1.0, -0.5, 1.5):
        with pytest.raises(ValueError, match='frac'):        # prune.py:20-21
            K.vertical_log_bin(block, frac)


@pytest.mark.parametrize('name', ['karate', 'er300', 'ba300', 'dw200_attrs', 'er2000', 'ba2000', 'directed120'])
def test_log_bin_and_chebyshev_vs_reference_golden(K, name):
    """Bins / distances of the reference's own pruner inputs, bit-exact."""
    import torch
    g = util.load_refex(name)
    n = int(g['n'])
    for gen in range(int(g['n_generations_recorded'])):
        names = g.js(f'g{gen}_working_before')
        exp_bins = g[f'g{gen}_binned']
        # rebuild the pruner input columns: previous working set + this generation's candidates
        cols = _working_columns(g, gen)
        block = torch.from_numpy(np.ascontiguousarray(np.stack([cols[nm] for nm in names]))).cuda()
        bins, nb = K.vertical_log_bin(block)
        got = bins.cpu().numpy().T
        assert np.array_equal(got, exp_bins), f'{name} gen {gen}'
        D = K.chebyshev([bins[j] for j in range(len(names))], n).cpu().numpy()
        assert np.array_equal(D, g[f'g{gen}_cheb']), f'{name} gen {gen}'


def _working_columns(g, gen):
    """name -> reference values for every column that is in the pruner input of `gen`."""
    cols = {}
    for gg in range(gen + 1):
        names = g.js(f'g{gg}_cand_names')
        vals = g[f'g{gg}_cand_values']
        for j, nm in enumerate(names):
            cols[nm] = vals[:, j]
    return cols


@pytest.mark.parametrize('n', [5000, 300001])
def test_log_bin_pass_skipping_patterns(K, n):
    """grx_vertical_log_bin skips the radix passes of byte positions that are constant over a column;
    every pattern of constant / varying bytes must give the oracle's bins."""
    import torch
    from oracle import ckernels
    rng = np.random.default_rng(n)
    ints = rng.integers(0, 200, n).astype(np.float64)
    cols = [
        ints,                                                           # low five mantissa bytes constant
        ints + 2.0 ** 40,                                               # top bytes constant too
        np.full(n, 7.25),                                               # every byte constant: no pass at all
        np.where(rng.random(n) < 0.5, 3.0, 5.0),                        # two values
        2.0 ** rng.integers(-20, 20, n).astype(np.float64),             # only exponent bytes vary
        rng.standard_normal(n),                                         # sign + everything varies
        -ints - 1.0,                                                    # negative: complemented keys
        rng.integers(0, 2 ** 31, n).astype(np.float64),                 # one constant byte (the lowest)
        (rng.integers(0, 256, n) * 2.0 ** -52 + 1.0),                   # only the LOWEST byte varies
        rng.random(n).astype(np.float32).astype(np.float64),            # low 29 mantissa bits zero
        np.where(rng.random(n) < 0.5, -0.0, 0.0),                       # -0.0 == 0.0: one bin
    ]
    for k in (1, 4, len(cols)):                                         # column subsets -> other launch shapes
        block = torch.from_numpy(np.stack(cols[:k] if k < len(cols) else cols)).cuda()
        bins, nb = K.vertical_log_bin(block)
        got = bins.cpu().numpy()
        for j in range(block.shape[0]):
            exp = ckernels.vertical_log_binning(cols[j])
            assert np.array_equal(got[j], exp), f'col {j} of {k}'
            assert int(nb[j]) == exp.max() + 1


@pytest.mark.parametrize('n', [70, 5000, 400001])
def test_log_bin_window_sort_adversarial(K, n):
    """The binning sort orders WIDE columns by their top four varying key bytes only and resolves runs of
    keys that agree in those bytes where a threshold lands: runs of 2..64 keys (shuffle ranking), long runs of
    equal keys (heavy ties), long runs of unequal keys (workgroup radix selection), and NARROW 32-bit columns
    next to them in one launch -- all must give the oracle's bins."""
    import torch
    from oracle import ckernels
    rng = np.random.default_rng(n + 7)
    tiny = 2.0 ** -40
    base = 1.0 + rng.integers(0, 1 << 20, n) * tiny                  # differ only in key bytes 1..3
    far = np.where(rng.random(n) < 0.01, rng.random(n) * 1e12, 0.0)   # a few huge values: byte 7 varies -> WIDE
    cols = [
        np.where(far > 0, far, base),                                 # ONE long run of unequal keys + outliers
        np.where(far > 0, far, 1.0 + (rng.integers(0, 3, n)) * tiny),  # three values in one long run
        np.where(rng.random(n) < 0.4, 0.1, rng.random(n)) + np.where(rng.random(n) < 0.001, 1e9, 0.0),   # heavy ties, WIDE
        np.round(rng.random(n) * 4096) / 4096 + rng.integers(0, 4, n) * 2.0 ** -45 + np.where(rng.random(n) < 0.01, 1e6, 0),
        rng.integers(0, 1 << 19, n).astype(np.float64),               # NARROW: integers below 2^20
        rng.integers(0, 1 << 30, n).astype(np.float64),               # integers up to 2^30: five varying bytes -> WIDE
        (rng.integers(0, 1 << 30, n) // 7 * 7).astype(np.float64) * 1024.0,
        rng.pareto(1.2, n),                                           # heavy tail, all distinct
        -rng.pareto(1.2, n).round(1),                                 # negative, ties
        np.where(rng.random(n) < 0.3, -0.0, rng.standard_normal(n).round(1)),
        # integer-key mode (non-negative integers below 2^32 sorted by the integer itself) and its boundaries
        rng.integers(0, 300, n).astype(np.float64),                   # two varying bytes as integers, four as fp64
        np.minimum(rng.pareto(1.1, n) * 50, 4294967295.0).round(),    # degree-like, up to the last admissible value
        np.where(rng.random(n) < 0.001, 4294967296.0, rng.integers(0, 1 << 16, n).astype(np.float64)),   # 2^32: not admissible
        np.where(rng.random(n) < 0.001, 0.5, rng.integers(0, 1 << 16, n).astype(np.float64)),            # one non-integer
        np.where(rng.random(n) < 0.001, -3.0, rng.integers(0, 1 << 16, n).astype(np.float64)),           # one negative
        np.where(rng.random(n) < 0.5, -0.0, rng.integers(0, 5, n).astype(np.float64)),                   # -0.0 is the integer 0
        (rng.integers(0, 1 << 12, n) * 65536).astype(np.float64),     # integers varying in bytes 2..3 only
    ]
    for frac in (0.5, 0.3, 0.9):
        block = torch.from_numpy(np.stack(cols)).cuda()
        bins, nb = K.vertical_log_bin(block, frac)
        got = bins.cpu().numpy()
        for j in range(len(cols)):
            exp = ckernels.vertical_log_binning(cols[j], frac)
            if exp.max() >= 128:
                assert int(nb[j]) < 0                                 # reported, never silently wrong
                continue
            assert np.array_equal(got[j], exp), f'col {j} frac {frac}'
            assert int(nb[j]) == exp.max() + 1


@pytest.mark.parametrize('n', [3, 64, 65, 1023, 1025, 50_000, 700_001])
def test_log_bin_bucket_select_adversarial(K, n):
    """The default binning does not sort: 4095 strided samples fix a monotone map of the key range onto <= 4096
    buckets, one pass counts (and certifies blocks of ties), an interval walk marks the buckets a threshold can fall
    into, one pass collects their keys, the segments are sorted in LDS, the exact walk reads thresholds and tie-run
    ends from them.  Patterns aimed at every step: values outside the sampled range (at positions the samples skip),
    sorted and reverse-sorted input (samples = quantiles), one dense non-uniform cluster (segments beyond the LDS
    capacity: radix selection), tie runs longer than a wavefront inside sorted segments and inside oversized ones, a
    column whose every bucket is a candidate, two-valued and constant columns, subnormals / huge magnitudes /
    negatives, and int64-bits columns over the whole int64 range."""
    import torch
    from oracle import ckernels, refex
    rng = np.random.default_rng(n + 99)
    base = rng.random(n)
    out_hi = base.copy(); out_hi[1 % n::max(n // 7, 1)] = 1e300                      # huge outliers off the sample grid
    out_lo = base + 10.0; out_lo[2 % n::max(n // 5, 1)] = -1e-300
    cluster = 1.0 + rng.integers(0, 1 << 30, n) * 2.0 ** -52                         # one bucket, 2^30 distinct keys
    cluster[::max(n // 3, 1)] = 1e6
    runs = 1.0 + rng.integers(0, max(n // 200, 2), n) * 1e-3                          # tie runs of ~200 in every segment
    tied_cluster = 1.0 + rng.integers(0, 1 << 12, n) * 2.0 ** -52                     # one bucket, 4096 values, long runs
    tied_cluster[::max(n // 3, 1)] = 1e6
    cols = [
        runs, tied_cluster,
        out_hi, out_lo,
        np.sort(rng.pareto(1.3, n)), np.sort(rng.pareto(1.3, n))[::-1].copy(),
        cluster,
        np.arange(n, dtype=np.float64),                                               # every rank its own key
        np.where(rng.random(n) < 0.5, 1.0, 2.0), np.full(n, -3.5),
        rng.standard_normal(n) * 10.0 ** rng.integers(-300, 300, n),                  # the whole exponent range
        np.where(rng.random(n) < 0.2, 5e-324, 0.0) * rng.integers(0, 50, n),          # subnormals and zeros
        -np.round(rng.pareto(1.1, n) * 20),                                           # negative integers, heavy ties
    ]
    wrapped = (rng.integers(-2 ** 63, 2 ** 63 - 1, n, dtype=np.int64, endpoint=True))
    small_int = rng.integers(-5, 5, n).astype(np.int64)
    near = (2 ** 53 + rng.integers(0, 3, n)).astype(np.int64)                         # distinct as int64, equal as fp64
    icols = [wrapped, small_int, near]
    for frac in (0.5, 0.25):
        block = np.stack(cols + [c.view(np.float64) for c in icols])
        flags = [False] * len(cols) + [True] * len(icols)
        bins, nb = K.vertical_log_bin(torch.from_numpy(block).cuda(), frac, is_i64=flags)
        got = bins.cpu().numpy()
        for j, c in enumerate(cols):
            exp = ckernels.vertical_log_binning(c, frac)
            if exp.max() >= 128:
                assert int(nb[j]) < 0
                continue
            assert np.array_equal(got[j], exp), f'col {j} frac {frac}'
            assert int(nb[j]) == exp.max() + 1
        for j, c in enumerate(icols):
            exp = refex.vertical_log_binning(c, frac)                                 # numpy on the int64 array
            if exp.max() >= 128:
                assert int(nb[len(cols) + j]) < 0
                continue
            assert np.array_equal(got[len(cols) + j], exp), f'int64 col {j} frac {frac}'


def test_log_bin_too_many_bins_is_reported(K):
    """frac so small that more than 128 bins are needed: the reference returns them, the device labels are
    7-bit -- the standalone API raises instead of returning saturated labels."""
    from graphrole_amd.features.prune import vertical_log_binning
    with pytest.raises(NotImplementedError, match='128 bins'):
        vertical_log_binning(np.arange(300.0), frac=0.001)
    assert vertical_log_binning(np.arange(10.0)).tolist() == [0, 0, 0, 0, 0, 1, 1, 2, 3, 4]


@pytest.mark.parametrize('n', [1000, 123457, 1 << 20])
def test_log_bin_vs_oracle_random(K, n):
    import torch
    from oracle import ckernels
    rng = np.random.default_rng(n)
    cols = [
        rng.pareto(1.5, n).round(2),                                   # heavy ties + heavy tail
        rng.integers(0, 30, n).astype(np.float64),                      # few distinct values
        rng.standard_normal(n),                                         # all distinct
        np.zeros(n),                                                    # constant
        np.where(rng.random(n) < 0.9, 0.0, rng.random(n)),              # mostly zeros
        np.arange(n, dtype=np.float64)[::-1].copy(),
    ]
    block = torch.from_numpy(np.stack(cols)).cuda()
    bins, nb = K.vertical_log_bin(block)
    got = bins.cpu().numpy()
    for j, c in enumerate(cols):
        exp = ckernels.vertical_log_binning(c)
        assert np.array_equal(got[j], exp), f'col {j}'
        assert int(nb[j]) == exp.max() + 1
    D = K.chebyshev([bins[j] for j in range(len(cols))], n).cpu().numpy()
    exp_D = ckernels.chebyshev(np.stack([ckernels.vertical_log_binning(c) for c in cols]))
    assert np.array_equal(D, exp_D)
    # only pairs that involve the last two ("new") columns
    D2 = K.chebyshev([bins[j] for j in range(len(cols))], n, first_new=4).cpu().numpy()
    mask = np.zeros_like(exp_D, dtype=bool)
    mask[:, 4:] = True
    mask[4:, :] = True
    np.fill_diagonal(mask, False)
    assert np.array_equal(D2[mask], exp_D[mask]) and np.all(D2[~mask] == 0)
    # row-range partial maxima combine by elementwise max (multi-GPU all-reduce MAX)
    h = n // 3
    Da = K.chebyshev([bins[j] for j in range(len(cols))], n, row_begin=0, row_end=h).cpu().numpy()
    Db = K.chebyshev([bins[j] for j in range(len(cols))], n, row_begin=h, row_end=n).cpu().numpy()
    assert np.array_equal(np.maximum(Da, Db), exp_D)


def test_chebyshev_many_columns(K):
    import torch
    rng = np.random.default_rng(0)
    n, F = 5000, 70
    B = rng.integers(0, 25, size=(F, n)).astype(np.uint8)
    Bd = torch.from_numpy(B).cuda()
    D = K.chebyshev([Bd[j] for j in range(F)], n).cpu().numpy()
    exp = np.abs(B[:, None, :].astype(np.int32) - B[None, :, :].astype(np.int32)).max(axis=2)
    assert np.array_equal(D, exp)


@pytest.mark.parametrize('F,n', [(40, 200_000), (96, 30_000), (97, 20_000), (230, 9_000)])
def test_chebyshev_cap_semantics_and_column_groups(K, F, n):
    """cap: entries <= cap exact, larger ones only known to be > cap (what the pruner needs);
    more than 96 columns run as launches over pairs of 48-column groups."""
    import torch
    rng = np.random.default_rng(F)
    base = rng.integers(0, 20, size=(8, n)).astype(np.uint8)
    B = np.empty((F, n), dtype=np.uint8)
    for j in range(F):                                          # families of near-identical columns
        B[j] = base[j % 8]
        flip = rng.integers(0, n, size=(j // 8) * 3)           # j // 8 controls how far a copy drifts
        B[j, flip] = np.minimum(B[j, flip] + (j // 8) % 4, 127)
    Bd = torch.from_numpy(B).cuda()
    exp = np.abs(B[:, None, :].astype(np.int32) - B[None, :, :].astype(np.int32)).max(axis=2)
    exact = K.chebyshev([Bd[j] for j in range(F)], n).cpu().numpy()
    assert np.array_equal(exact, exp)
    for cap in (0, 1, 2, 5):
        D = K.chebyshev([Bd[j] for j in range(F)], n, cap=cap).cpu().numpy()
        small = exp <= cap
        assert np.array_equal(D[small], exp[small])
        assert np.all(D[~small] > cap) and np.all(D[~small] <= exp[~small])
        assert np.array_equal(D, D.T)
    # row ranges combine by elementwise max also under a cap
    h = n // 2 + 7
    Da = K.chebyshev([Bd[j] for j in range(F)], n, row_begin=0, row_end=h, cap=1).cpu().numpy()
    Db = K.chebyshev([Bd[j] for j in range(F)], n, row_begin=h, row_end=n, cap=1).cpu().numpy()
    both = np.maximum(Da, Db)
    assert np.array_equal(both <= 1, exp <= 1) and np.array_equal(both[exp <= 1], exp[exp <= 1])


# ------------------------------------------------------------------------------------- NMF
@pytest.mark.parametrize('shape', [(1000, 7, 4), (50000, 12, 6), (3000, 40, 6), (2000, 64, 8), (700, 100, 16),
                                   (100, 3, 2), (4099, 48, 5), (1001, 17, 3), (333, 33, 2), (17, 1, 1), (5000, 49, 4),
                                   (1200, 121, 4), (900, 200, 7), (600, 480, 16),
                                   (777, 70, 5), (1500, 90, 3), (2049, 128, 6), (130, 113, 4)])
def test_nmf_building_blocks_vs_numpy(K, shape):
    import torch
    from oracle import rolx
    n, F, r = shape
    rng = np.random.default_rng(n + F)
    X = np.abs(rng.standard_normal((n, F))) * np.linspace(1, 50, F)
    Xd = torch.from_numpy(np.ascontiguousarray(X.T)).cuda()
    # gram, identity and transformed
    G, xsum = K.gram(Xd, n)
    np.testing.assert_allclose(G, X.T @ X, rtol=1e-12)
    np.testing.assert_allclose(xsum, X.sum(), rtol=1e-12)
    k = max(1, F - 2)
    T = rng.standard_normal((F, k))
    G2, _ = K.gram(Xd, n, T)
    Y = X @ T
    np.testing.assert_allclose(G2, Y.T @ Y, rtol=1e-10, atol=1e-10 * np.abs(Y.T @ Y).max())
    # row-range partials add up
    Ga, sa = K.gram(Xd, n, None, 0, n // 2)
    Gb, sb = K.gram(Xd, n, None, n // 2, n)
    np.testing.assert_allclose(Ga + Gb, X.T @ X, rtol=1e-12)
    # project + stats
    Z = rng.standard_normal((F, r))
    U, stats = K.project(Xd, n, Z)
    Ue = X @ Z
    np.testing.assert_allclose(U.cpu().numpy()[:, :n].T, Ue, rtol=1e-11, atol=1e-11 * np.abs(Ue).max())
    Ug = U.cpu().numpy()[:, :n].T
    idx = np.argmax(np.abs(Ug), axis=0)
    assert np.array_equal(stats[:, 1].astype(np.int64), idx)
    assert np.array_equal(stats[:, 0], Ug[idx, np.arange(r)])
    np.testing.assert_allclose(stats[:, 2], (np.maximum(Ug, 0) ** 2).sum(axis=0), rtol=1e-12)
    np.testing.assert_allclose(stats[:, 3], (np.minimum(Ug, 0) ** 2).sum(axis=0), rtol=1e-12)
    # one multiplicative update against the numpy restatement of sklearn's update
    W0 = np.abs(rng.standard_normal((n, r))) + 0.1
    H0 = np.abs(rng.standard_normal((r, F))) + 0.1
    W0[::17, 1 % r] = 0.0
    W0[::29, :] = 0.0                                         # all-zero rows: zero denominators -> EPSILON
    Wd = torch.from_numpy(np.ascontiguousarray(W0.T)).cuda()
    st = K.NmfState(Xd, n, Wd, H0)
    st.w_pass()
    W1 = W0 * ((X @ H0.T) / np.where(W0 @ (H0 @ H0.T) == 0, rolx.EPSILON, W0 @ (H0 @ H0.T)))
    np.testing.assert_allclose(st.W.cpu().numpy().T, W1, rtol=1e-12, atol=0)
    AB = st.AB.cpu().numpy()
    np.testing.assert_allclose(AB[:r * F].reshape(r, F), W1.T @ X, rtol=1e-12)
    np.testing.assert_allclose(AB[r * F:].reshape(r, r), W1.T @ W1, rtol=1e-12)
    st.h_update()
    den = (W1.T @ W1) @ H0
    H1 = H0 * ((W1.T @ X) / np.where(den == 0, rolx.EPSILON, den))
    np.testing.assert_allclose(st.H.cpu().numpy(), H1, rtol=1e-12)
    err = float(st.residual_sq().cpu()[0])
    np.testing.assert_allclose(err, ((X - W1 @ H1) ** 2).sum(), rtol=1e-11)


def test_nmf_iterate_matches_oracle_loop(K):
    import torch
    from oracle import rolx
    rng = np.random.default_rng(5)
    n, F, r = 20000, 10, 6
    X = np.abs(rng.standard_normal((n, F))) * np.linspace(1, 20, F)
    W0 = np.abs(rng.standard_normal((n, r))) + 0.05
    H0 = np.abs(rng.standard_normal((r, F))) + 0.05
    We, He, _ = rolx.mu_iterations(X, W0, H0, tol=0.0, max_iter=20)
    st = K.NmfState(torch.from_numpy(np.ascontiguousarray(X.T)).cuda(), n,
                    torch.from_numpy(np.ascontiguousarray(W0.T)).cuda(), H0)
    st.iterate(20)
    np.testing.assert_allclose(st.W.cpu().numpy().T, We, rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(st.H.cpu().numpy(), He, rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(np.sqrt(float(st.err.cpu()[0])), rolx.frobenius_error(X, We, He), rtol=1e-9)
    # bitwise reproducible
    st2 = K.NmfState(torch.from_numpy(np.ascontiguousarray(X.T)).cuda(), n,
                     torch.from_numpy(np.ascontiguousarray(W0.T)).cuda(), H0)
    st2.iterate(20)
    assert torch.equal(st.W, st2.W) and torch.equal(st.H, st2.H)


def test_nndsvd_apply(K):
    import torch
    rng = np.random.default_rng(2)
    n, r = 5000, 5
    U = rng.standard_normal((r, n)) * 1e-3
    Ud = torch.from_numpy(U.copy()).cuda()
    sign = np.array([0.0, 1.0, -1.0, 1.0, -1.0])
    scale = np.array([2.0, 3.0, 0.5, 1e-4, 7.0])
    K.nndsvd_apply(Ud, n, sign, scale, 1e-6, 0.123)
    exp = np.empty_like(U)
    for j in range(r):
        v = np.abs(U[j]) if sign[j] == 0 else np.maximum(sign[j] * U[j], 0)
        v = v * scale[j]
        exp[j] = np.where(v < 1e-6, 0.123, v)
    assert np.array_equal(Ud.cpu().numpy(), exp)


def test_unsupported_shapes_fail_loudly(K):
    import torch
    from graphrole_amd._lib import GrxError
    X = torch.zeros((481, 10), dtype=torch.float64, device='cuda')          # GRX_MAX_NMF_FEATURES = 480
    with pytest.raises(GrxError):
        K.gram(X, 10)


@pytest.mark.parametrize('f', [1, 3, 8, 9, 16, 23, 48, 130])
@pytest.mark.parametrize('n', [1, 63, 64, 1000 + 37])
def test_pack_rows_every_width(K, f, n):
    """grx_pack_rows: the thread-per-row kernel (rows of 16 / 32 / 64 bytes) and the LDS-tiled one (whole
    lines, pointer table split at 128 columns) give the zero-padded row-major block."""
    import torch
    X = np.random.default_rng(100 * f + n).standard_normal((f, n))
    Xd = torch.from_numpy(X).cuda()
    rows, ldr = K.pack_rows([Xd[c] for c in range(f)], n)
    torch.cuda.synchronize()
    got = rows.cpu().numpy()
    assert got.shape == (n, ldr) and ldr >= f
    assert np.array_equal(got[:, :f], X.T)
    assert np.all(got[:, f:] == 0.0)


def test_log_bin_stored_bucket_id_path_equals_the_default():
    """Columns of 2.5 M rows and more keep the bucket ids of the histogram pass so that the collect and label passes read
    2 bytes per key instead of 8 (csrc/grx_prune.hip sel_assign_kernel).  GRX_BIN_BID_MIN_N=0 sends EVERY column that
    way: the whole binning / Chebyshev part of this file -- the reference's known answers, the goldens' per-generation
    bins, the adversarial and random cases against the oracle -- must pass unchanged (bins are integers: equal, not close)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, GRX_BIN_BID_MIN_N='0')
    res = subprocess.run([sys.executable, '-m', 'pytest', os.path.join(root, 'tests', 'test_gpu_kernels.py'), '-q', '-m', 'gpu', '-x',
                          '-k', '(log_bin or chebyshev) and not stored_bucket_id', '-p', 'no:cacheprovider'],
                         capture_output=True, text=True, timeout=1500, env=env, cwd=root)
    assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-2000:]
    assert ' passed' in res.stdout

