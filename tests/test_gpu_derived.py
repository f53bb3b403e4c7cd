"""
-m gpu: grx_aggregate_derived -- neighbour sums / means of base columns P_k and of fl(P_k / d) formed in registers --
is bit-identical to grx_aggregate on the columns P_k and fl(P_k / d) themselves: every row width (1 .. 15 bases), hubs
beyond 128 and beyond 8192 neighbours, dangling nodes, ragged row ranges, values from 1e-300 to 1e300, zeros of both
signs, and the shape it is for (weighted directed graph with 13 generation-0 columns).
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _device_graph(G):
    from graphrole_amd import RecursiveFeatureExtractor
    fe = RecursiveFeatureExtractor(G, max_generations=2, attributes=bool(G.attributes))
    return fe, fe.graph._device_graph()[1]


def _reference(K, dev, cols, n, b=0, e=None):
    rows, ldr = K.pack_rows(cols, n)
    blk = K.aggregate(dev, rows, len(cols), ldr, b, e)
    return blk[:len(cols)], blk[len(cols):]


def _check(G, base_values, ranges=((0, None),)):
    from graphrole_amd import kernels as K
    fe, dev = _device_graph(G)
    n = dev.n
    cnt = np.diff(K.to_host(dev.row_ptr)).astype(np.float64)
    bases = [K.to_device(np.ascontiguousarray(v)) for v in base_values]
    with np.errstate(divide='ignore', invalid='ignore'):
        divided = [K.to_device(np.where(cnt > 0, v / np.where(cnt > 0, cnt, 1.0), 0.0)) for v in base_values]
    nb = len(bases)
    for b, e in ranges:
        got = K.aggregate_derived(dev, bases, [(True, True, True, True)] * nb, b, e)
        ps, pm = _reference(K, dev, bases, n, b, e)
        qs, qm = _reference(K, dev, divided, n, b, e)
        hi = n if e is None else e
        for c in range(nb):
            for k, ref in enumerate((ps[c], pm[c], qs[c], qm[c])):
                assert torch.equal(got[c][k][b:hi], ref[b:hi]), (nb, c, k, b, e)


@pytest.mark.parametrize('nb', [1, 2, 3, 4, 7, 8, 13, 15])
def test_derived_equals_plain_for_every_row_width(nb):
    from graphrole_amd import synth
    G = synth.ba_graph(30_000, 6, seed=nb)
    rng = np.random.RandomState(nb)
    vals = [np.abs(rng.randn(G.n)) * 10.0 ** rng.uniform(-3, 9) for _ in range(nb)]
    _check(G, vals, ranges=((0, None), (0, 513), (7_001, 22_222)))


def test_derived_on_a_weighted_directed_graph_with_hubs_and_dangling_nodes():
    from graphrole_amd import synth
    G = synth.directed_weighted_graph(60_000, 900_000, seed=5)           # power-law in-degree; some nodes have no out-arcs
    rng = np.random.RandomState(0)
    vals = [rng.gamma(0.5, 10.0 ** rng.uniform(-2, 6), G.n) for _ in range(13)]      # config 5's generation-0 width
    vals[3] = -vals[3]                                                                 # attributes may be negative
    _check(G, vals, ranges=((0, None), (59_000, None)))


def test_derived_with_a_very_long_row():
    from graphrole_amd.graph.csr import CSRGraph
    n = 25_000
    src = np.concatenate([np.zeros(20_000, dtype=np.int64), np.arange(1, 3_000, dtype=np.int64)])
    dst = np.concatenate([np.arange(1, 20_001, dtype=np.int64), np.arange(2, 3_001, dtype=np.int64)])
    G = CSRGraph(n, src, dst, validate=False)
    rng = np.random.RandomState(1)
    _check(G, [rng.rand(n) * 1e5, rng.rand(n)], ranges=((0, None), (0, 1)))


def test_derived_division_on_extreme_values():
    """the in-register quotient equals the division for tiny, huge, zero (both signs) and non-finite bases"""
    from graphrole_amd import synth
    G = synth.er_graph(8_000, 40_000, seed=2)
    rng = np.random.RandomState(2)
    n = G.n
    tiny = 10.0 ** rng.uniform(-320, -280, n)                # subnormals and the slow path's range
    huge = 10.0 ** rng.uniform(280, 305, n)
    mixed = rng.randn(n) * 10.0 ** rng.uniform(-20, 20, n)
    mixed[::7] = 0.0
    mixed[3::11] = -0.0
    _check(G, [tiny, huge, mixed])


def test_derived_rejects_bad_shapes():
    from graphrole_amd import _lib, kernels as K, synth
    _, dev = _device_graph(synth.er_graph(500, 2_000, seed=0))
    cols = [K.to_device(np.ones(dev.n)) for _ in range(16)]
    with pytest.raises(ValueError):
        K.aggregate_derived(dev, cols, [(True, False, False, False)] * 16)
    assert _lib.load().grx_aggregate_derived_ldr(15) == 16 and _lib.load().grx_aggregate_derived_ldr(1) == 2
    assert _lib.load().grx_aggregate_derived_ldr(0) == 0 and _lib.load().grx_aggregate_derived_ldr(16) == 0
