"""
-m gpu: graph ingest on the device (grx_ingest, grx_orient_*) against the host construction it replaces
(graphrole_amd/graph/csr.py: CSRGraph + InternalGraph, kernels.DeviceCSR.oriented): every array -- internal
order, row pointers, ascending and adjacency-order columns, weights, the transposed CSR, the oriented graph with
its per-arc table -- must be identical, and the extractor must return the same table through both.
"""
import numpy as np
import pytest

from tests import util

pytestmark = pytest.mark.gpu

SPECS = [
    dict(n=50, m=120, seed=1),
    dict(n=300, m=2500, seed=2, self_loops=7),
    dict(n=200, m=900, seed=3, directed=True, weighted=True, self_loops=3),
    dict(n=150, m=700, seed=4, weighted=True, self_loops=2),
    dict(n=400, m=3000, seed=5, directed=True),
    dict(n=20000, m=150000, seed=6, self_loops=11),
]


def _both(spec):
    from graphrole_amd import kernels as K
    from graphrole_amd.graph.csr import CSRGraph, InternalGraph
    src, dst, w = util.random_graph(**spec)
    G = CSRGraph(spec['n'], src, dst, weights=w, directed=spec.get('directed', False))
    host = InternalGraph(G)
    perm, inv, row_ptr, out, tr = K.device_ingest(G.n, src, dst, w, G.directed, G.nnz)
    return G, host, perm, inv, row_ptr, out, tr


@pytest.mark.parametrize('spec', SPECS, ids=lambda s: f"n{s['n']}_{'d' if s.get('directed') else 'u'}{'w' if s.get('weighted') else ''}")
def test_device_ingest_equals_host_construction(spec):
    from graphrole_amd import kernels as K
    G, host, perm, inv, row_ptr, out, tr = _both(spec)
    # (the internal order comes back as int32 device tensors: it never has to leave HBM on the hot path)
    assert np.array_equal(perm.cpu().numpy(), host.perm) and np.array_equal(inv.cpu().numpy(), host.inv)
    assert np.array_equal(row_ptr, host.row_ptr)
    nnz = host.nnz
    assert out.nnz == nnz == G.nnz
    assert np.array_equal(out.col.cpu().numpy()[:nnz], host.col)
    assert np.array_equal(out.agg_col.cpu().numpy()[:nnz], host.agg_col)
    if host.weighted:
        assert np.array_equal(out.w.cpu().numpy()[:nnz], host.w)
    if host.directed:
        assert np.array_equal(tr.row_ptr.cpu().numpy(), host.t_row_ptr)
        assert np.array_equal(tr.col.cpu().numpy()[:len(host.t_col)], host.t_col)
        if host.weighted:
            assert np.array_equal(tr.w.cpu().numpy()[:len(host.t_w)], host.t_w)
    else:
        # orientation for triangle counting: device (no host columns) against the numpy construction
        ref = K.DeviceCSR(host.row_ptr, host.col, host.w, agg_col=host.agg_col).oriented()
        got = out.oriented()
        assert np.array_equal(got.row_ptr.cpu().numpy(), ref.row_ptr.cpu().numpy())
        o_nnz = ref.nnz
        assert got.nnz == o_nnz
        assert np.array_equal(got.col.cpu().numpy()[:o_nnz], ref.col.cpu().numpy()[:o_nnz])
        assert np.array_equal(got.arc.cpu().numpy()[:o_nnz], ref.arc.cpu().numpy()[:o_nnz])


@pytest.mark.parametrize('spec', SPECS[:5], ids=lambda s: f"n{s['n']}")
def test_extractor_same_table_through_both_ingests(spec):
    from graphrole_amd import RecursiveFeatureExtractor
    from graphrole_amd.graph.csr import CSRGraph
    src, dst, w = util.random_graph(**spec)

    def run(device_ingest):
        G = CSRGraph(spec['n'], src, dst, weights=w, directed=spec.get('directed', False))
        fe = RecursiveFeatureExtractor(G, max_generations=4)
        fe.graph._device_ingest = device_ingest
        return fe.extract_features()

    a, b = run(True), run(False)
    assert list(a.columns) == list(b.columns) and [str(t) for t in a.dtypes] == [str(t) for t in b.dtypes]
    assert np.array_equal(a.values, b.values)


def test_device_ingest_at_bench_scale_matches_oracle():
    """BA 200 k / 2 M through the device ingest: the table equals the oracle's bit for bit (the adjacency order
    of the sums is the edge-appearance order, features/extract.py:108-110)."""
    from graphrole_amd import RecursiveFeatureExtractor, synth
    from oracle import refex
    G = synth.ba_graph(200_000, 10, seed=4)
    fe = RecursiveFeatureExtractor(G, max_generations=4)
    X = fe.extract_features()
    assert type(fe.graph._device_graph()[0]).__name__ == 'DeviceBuiltGraph'
    og = refex.OracleGraph(labels=G.labels, row_ptr=G.row_ptr, col=G.col, w=None, directed=False,
                           num_edges=G.num_edges, adj_col=G.adj_col)
    ref = refex.extract_features(og, max_generations=4, fast=True)
    assert list(X.columns) == ref.columns
    assert np.array_equal(X.values.astype(float), ref.values)
