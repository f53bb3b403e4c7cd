"""
-m gpu: BASELINE.json full-size cases.  Parity with the C oracle where it finishes in seconds, and
size-independent properties (sortedness, permutation checksums, linearity, idempotence, bitwise
reproducibility) on the 1 M-node / 10 M-edge power-law graph of configs 3/4.
"""
import os

import numpy as np
import pytest

from tests import util

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def ba1m():
    from graphrole_amd import synth
    return synth.ba_graph(1_000_000, 10, seed=0)


def _oracle_graph(G):
    from oracle import refex
    return refex.OracleGraph(labels=G.labels, row_ptr=G.row_ptr, col=G.col, w=G.w, directed=G.directed,
                             num_edges=G.num_edges, t_row_ptr=G.t_row_ptr, t_col=G.t_col, t_w=G.t_w,
                             adj_col=G.adj_col)


def test_config2_er_100k_1m_matches_oracle():
    """BASELINE config 2: Erdos-Renyi 100k / 1M, 4 generations, n_roles = 6."""
    from graphrole_amd import RecursiveFeatureExtractor, synth
    from graphrole_amd.roles import factor
    from oracle import refex, rolx
    G = synth.er_graph(100_000, 1_000_000, seed=0)
    fe = RecursiveFeatureExtractor(G, max_generations=4)
    X = fe.extract_features()
    ref = refex.extract_features(_oracle_graph(G), max_generations=4, fast=True)
    assert list(X.columns) == ref.columns and fe.generation_count == ref.generation_count
    for gen, tr in enumerate(ref.trace):
        assert fe._final_names[gen] == tr.retained
    assert np.array_equal(X.values.astype(float), ref.values)          # unweighted: bit-exact
    np.random.seed(0)
    Gf, Ff, n_iter = factor.nmf_with_info(X.values.astype(float), 6)
    np.random.seed(0)
    We, He, it = rolx.nmf(X.values.astype(float), 6)
    assert n_iter == it
    assert np.abs(Gf - We).max() / np.abs(We).max() < 1e-8
    assert np.abs(Ff - He).max() / np.abs(He).max() < 1e-8


def test_config3_ba_1m_10m_matches_oracle(ba1m):
    """BASELINE config 3: the full 1 M / 10 M graph against the oracle's C port (retained sets exact)."""
    from graphrole_amd import RecursiveFeatureExtractor
    from oracle import refex
    fe = RecursiveFeatureExtractor(ba1m, max_generations=4)
    X = fe.extract_features()
    ref = refex.extract_features(_oracle_graph(ba1m), max_generations=4, fast=True)
    assert list(X.columns) == ref.columns and fe.generation_count == ref.generation_count
    for gen, tr in enumerate(ref.trace):
        assert fe._final_names[gen] == tr.retained
    assert np.array_equal(X.values.astype(float), ref.values)          # unweighted: bit-exact
    # gen-0 integer columns are exact
    for col in ('degree', 'internal_edges', 'external_edges'):
        assert np.array_equal(X[col].values, ref.values[:, ref.columns.index(col)].astype(np.int64))
    # BASELINE config 3's RolX half: NMF of the 1 M x F table, n_roles = 6, against the oracle's restatement
    # of sklearn NMF(mu, nndsvda) (graphrole/roles/factor.py:10-26): equal iteration count, factors 1e-8
    from graphrole_amd.roles import factor
    from oracle import rolx
    Xv = X.values.astype(float)
    np.random.seed(0)
    Gf, Ff, n_iter = factor.nmf_with_info(Xv, 6)
    np.random.seed(0)
    We, He, it = rolx.nmf(Xv, 6)
    assert n_iter == it
    assert np.abs(Gf - We).max() / np.abs(We).max() < 1e-8
    assert np.abs(Ff - He).max() / np.abs(He).max() < 1e-8


def test_fullsize_properties(ba1m):
    import torch
    from graphrole_amd import kernels as K
    n = ba1m.n
    csr = K.DeviceCSR(ba1m.row_ptr, ba1m.col, agg_col=ba1m.adj_col)
    deg = torch.from_numpy(np.diff(ba1m.row_ptr).astype(np.float64)).cuda()
    rng = torch.Generator(device='cuda').manual_seed(1)
    a = torch.rand(n, dtype=torch.float64, device='cuda', generator=rng)
    b = torch.rand(n, dtype=torch.float64, device='cuda', generator=rng) * 100
    ones = torch.ones(n, dtype=torch.float64, device='cuda')
    rows, ldr = K.pack_rows([a, b, a + b, ones], n)
    blk = K.aggregate(csr, rows, 4, ldr)
    # sum of ones = degree, mean of ones = 1 (all nodes have neighbours here), both exact
    assert torch.equal(blk[3], deg)
    assert torch.equal(blk[7], ones)
    # linearity: agg(a + b) = agg(a) + agg(b) up to re-association
    torch.testing.assert_close(blk[2], blk[0] + blk[1], rtol=1e-12, atol=0)
    # mean = sum / degree exactly (IEEE division of the same sum)
    assert torch.equal(blk[4], blk[0] / deg)
    # bitwise reproducible
    assert torch.equal(K.aggregate(csr, rows, 4, ldr), blk)
    # sort: sorted, a permutation (order-independent checksums), idempotent
    block = torch.stack([a, b, blk[0], blk[4]])
    srt = K.sort_columns(block)
    assert bool((srt[:, 1:] >= srt[:, :-1]).all())
    assert torch.equal(srt.view(torch.int64).sum(dim=1), block.contiguous().view(torch.int64).sum(dim=1))
    assert torch.equal(srt.view(torch.int64).bitwise_xor(0x5555).sum(dim=1),
                       block.contiguous().view(torch.int64).bitwise_xor(0x5555).sum(dim=1))
    assert torch.equal(K.sort_columns(srt), srt)
    # binning: monotone in the value, bin 0 holds at least half of the column, last bin non-empty
    bins, nb = K.vertical_log_bin(block)
    for j in range(4):
        order = torch.argsort(block[j])
        bj = bins[j][order].to(torch.int32)
        assert bool((bj[1:] >= bj[:-1]).all())
        assert int((bins[j] == 0).sum()) >= n // 2
        assert int(bins[j].max()) == int(nb[j]) - 1
    # binning is idempotent on its own output ordering: equal values share a bin
    vals, inv = torch.unique(block[1], return_inverse=True)
    first = torch.zeros(len(vals), dtype=torch.uint8, device='cuda').scatter_(0, inv, bins[1])
    assert torch.equal(first[inv], bins[1])
    # Chebyshev matrix is symmetric with a zero diagonal and invariant to the row split
    D = K.chebyshev([bins[j] for j in range(4)], n)
    assert torch.equal(D, D.T) and int(D.diagonal().abs().sum()) == 0
    Dh = torch.maximum(K.chebyshev([bins[j] for j in range(4)], n, 0, 0, n // 3),
                       K.chebyshev([bins[j] for j in range(4)], n, 0, n // 3, n))
    assert torch.equal(D, Dh)


def test_config5_like_directed_weighted_attributes_matches_oracle():
    """Scaled-down config 5 (directed, weighted, power-law in-degree, 8 attributes): 200k / 4M."""
    from graphrole_amd import RecursiveFeatureExtractor, synth
    from oracle import refex
    G = synth.directed_weighted_graph(200_000, 4_000_000, seed=0)
    fe = RecursiveFeatureExtractor(G, max_generations=3, attributes=True)
    X = fe.extract_features()
    og = _oracle_graph(G)
    og.attrs = {'attribute_' + k: np.asarray(v, dtype=np.float64) for k, v in G.attributes.items()}
    ref = refex.extract_features(og, max_generations=3, fast=True)
    assert list(X.columns) == ref.columns and fe.generation_count == ref.generation_count
    for gen, tr in enumerate(ref.trace):
        assert fe._final_names[gen] == tr.retained
    np.testing.assert_allclose(X.values.astype(float), ref.values, rtol=util.WEIGHTED_RTOL, atol=0)


@pytest.mark.skipif(os.environ.get('GRX_SKIP_CONFIG5') == '1', reason='GRX_SKIP_CONFIG5=1')
def test_config5_full_size_matches_oracle():
    """BASELINE config 5 at its real size on one GPU: directed, weighted, 5 M nodes / 100 M arcs, 8 numeric
    node attributes, attributes=True, max_generations=4 -- ReFeX against the oracle's C port (retained lists
    exact, values util.WEIGHTED_RTOL (1e-11): weighted generation-0 sums run in another order than networkx's),
    then the RolX NMF of the resulting wide table (n_roles = 6) against oracle.rolx.nmf
    (graphrole/features/extract.py:65-89, graphrole/roles/factor.py:10-26)."""
    from graphrole_amd import RecursiveFeatureExtractor, synth
    from graphrole_amd.roles import factor
    from oracle import refex, rolx
    G = synth.directed_weighted_graph(5_000_000, 100_000_000, seed=0)
    fe = RecursiveFeatureExtractor(G, max_generations=4, attributes=True)
    X = fe.extract_features()
    og = _oracle_graph(G)
    og.attrs = {'attribute_' + k: np.asarray(v, dtype=np.float64) for k, v in G.attributes.items()}
    ref = refex.extract_features(og, max_generations=4, fast=True)
    assert list(X.columns) == ref.columns and fe.generation_count == ref.generation_count
    for gen, tr in enumerate(ref.trace):
        assert fe._final_names[gen] == tr.retained
    Xv = X.values.astype(float)
    np.testing.assert_allclose(Xv, ref.values, rtol=util.WEIGHTED_RTOL, atol=0)
    del fe, og, ref
    assert Xv.shape[1] > 64                      # the wide-table NMF path (F above the 64-column fast path)
    np.random.seed(0)
    Gf, Ff, n_iter = factor.nmf_with_info(Xv, 6)
    np.random.seed(0)
    We, He, it = rolx.nmf(Xv, 6)
    assert n_iter == it
    assert np.abs(Gf - We).max() / np.abs(We).max() < 1e-8
    assert np.abs(Ff - He).max() / np.abs(He).max() < 1e-8
