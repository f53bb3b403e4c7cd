"""
-m "not gpu": the neighbour ORDER is part of the reference's behaviour (its sums run in G[node]
order).  Host-side checks that every graph class carries that order to the
device layout: CSRGraph (order of appearance / explicit adjacency), the networkx adapter (G.adj
iteration order), InternalGraph (degree-descending relabelling keeps each row's sequence).
"""
import networkx as nx
import numpy as np
import pytest

from graphrole_amd.graph.csr import CSRGraph, InternalGraph
from graphrole_amd.graph.interface.networkx import NetworkxInterface


def _rows(ptr, col):
    return [list(col[ptr[i]:ptr[i + 1]]) for i in range(len(ptr) - 1)]


def test_csr_graph_adjacency_is_order_of_appearance():
    #            e0      e1      e2      e3      e4 (self-loop)  e5
    src = np.array([3, 0, 2, 0, 1, 3])
    dst = np.array([0, 2, 3, 1, 1, 1])
    g = CSRGraph(4, src, dst)
    assert _rows(g.row_ptr, g.col) == [[1, 2, 3], [0, 1, 3], [0, 3], [0, 1, 2]]          # ascending
    # what networkx holds after add_edge(src[i], dst[i]) for i = 0, 1, ...
    G = nx.Graph()
    G.add_nodes_from(range(4))
    G.add_edges_from(zip(src.tolist(), dst.tolist()))
    assert _rows(g.row_ptr, g.adj_col) == [list(G.adj[v]) for v in range(4)]
    # directed: successors in arc order
    gd = CSRGraph(4, src, dst, directed=True)
    D = nx.DiGraph()
    D.add_nodes_from(range(4))
    D.add_edges_from(zip(src.tolist(), dst.tolist()))
    assert _rows(gd.row_ptr, gd.adj_col) == [list(D.adj[v]) for v in range(4)]


def test_explicit_adjacency_is_validated():
    src, dst = np.array([0, 1, 2]), np.array([1, 2, 0])
    good = np.array([2, 1, 0, 2, 1, 0])                                    # rows: [2,1] [0,2] [1,0]
    g = CSRGraph(3, src, dst, adjacency=good)
    assert _rows(g.row_ptr, g.adj_col) == [[2, 1], [0, 2], [1, 0]]
    with pytest.raises(ValueError, match='adjacency'):
        CSRGraph(3, src, dst, adjacency=np.array([2, 1, 0, 2, 1]))         # wrong length
    with pytest.raises(ValueError, match='permutations'):
        CSRGraph(3, src, dst, adjacency=np.array([2, 2, 0, 2, 1, 0]))      # row 0 lists 2 twice


def test_networkx_adapter_reads_adjacency_order():
    rng = np.random.default_rng(3)
    edges = [(int(a), int(b)) for a, b in rng.integers(0, 40, size=(300, 2)) if a != b]
    G = nx.Graph()
    G.add_nodes_from(range(40))
    for k in rng.permutation(len(edges)):                                  # insertion order != sorted order
        G.add_edge(*edges[k])
    csr = NetworkxInterface(G).to_csr()
    assert _rows(csr.row_ptr, csr.adj_col) == [list(G.adj[v]) for v in sorted(G.nodes)]
    assert _rows(csr.row_ptr, csr.col) == [sorted(G.adj[v]) for v in sorted(G.nodes)]
    # labels that are not 0..n-1
    H = nx.relabel_nodes(G, {v: f'n{v:02d}' for v in G.nodes})
    csr2 = NetworkxInterface(H).to_csr()
    labels = sorted(H.nodes)
    index = {lab: i for i, lab in enumerate(labels)}
    assert _rows(csr2.row_ptr, csr2.adj_col) == [[index[u] for u in H.adj[lab]] for lab in labels]


def test_internal_graph_keeps_each_rows_sequence():
    rng = np.random.default_rng(5)
    n, m = 200, 1500
    src, dst = rng.integers(0, n, m), rng.integers(0, n, m)
    keep = src != dst
    key = np.minimum(src, dst)[keep] * n + np.maximum(src, dst)[keep]
    _, idx = np.unique(key, return_index=True)
    src, dst = src[keep][np.sort(idx)], dst[keep][np.sort(idx)]
    g = CSRGraph(n, src, dst)
    ig = InternalGraph(g)
    deg = np.diff(g.row_ptr)
    assert np.all(np.diff(deg[ig.perm]) <= 0)                              # degree-descending rows
    assert np.array_equal(ig.inv[ig.perm], np.arange(n))
    for i in range(n):                                                     # internal row i = label row perm[i]
        lab = ig.perm[i]
        want_sorted = sorted(ig.inv[g.col[g.row_ptr[lab]:g.row_ptr[lab + 1]]])
        want_order = list(ig.inv[g.adj_col[g.row_ptr[lab]:g.row_ptr[lab + 1]]])
        assert list(ig.col[ig.row_ptr[i]:ig.row_ptr[i + 1]]) == want_sorted
        assert list(ig.agg_col[ig.row_ptr[i]:ig.row_ptr[i + 1]]) == want_order
    x = rng.random(n)
    assert np.array_equal(ig.to_label_order(ig.to_internal(x)), x)


def test_csr_graph_from_scipy_sparse_and_edgelist_file(tmp_path):
    import scipy.sparse as sp
    rng = np.random.default_rng(1)
    n = 30
    src, dst = rng.integers(0, n, 120), rng.integers(0, n, 120)
    key = np.minimum(src, dst) * n + np.maximum(src, dst)
    _, idx = np.unique(key, return_index=True)
    src, dst = src[np.sort(idx)], dst[np.sort(idx)]
    w = rng.uniform(0.5, 2.0, len(src))
    ref = CSRGraph(n, src, dst, weights=w)
    A = sp.coo_matrix((np.concatenate([w, w[src != dst]]),
                       (np.concatenate([src, dst[src != dst]]), np.concatenate([dst, src[src != dst]]))), shape=(n, n))
    g = CSRGraph.from_scipy_sparse(A)
    assert g.weighted and not g.directed and g.num_edges == ref.num_edges
    assert np.array_equal(g.row_ptr, ref.row_ptr) and np.array_equal(g.col, ref.col)
    np.testing.assert_allclose(g.w, ref.w)
    assert sorted(g.adj_col[g.row_ptr[3]:g.row_ptr[4]]) == list(ref.col[ref.row_ptr[3]:ref.row_ptr[4]])
    gu = CSRGraph.from_scipy_sparse((A != 0).astype(float))
    assert not gu.weighted and gu.integral
    gd = CSRGraph.from_scipy_sparse(sp.csr_matrix(([1.0, 2.0], ([0, 2], [1, 0])), shape=(3, 3)), directed=True)
    assert gd.directed and gd.num_edges == 2 and list(gd.col) == [1, 0]
    with pytest.raises(ValueError, match='symmetric'):
        CSRGraph.from_scipy_sparse(sp.csr_matrix(([1.0], ([0], [1])), shape=(2, 2)))
    # text edge list with arbitrary integer ids, a duplicate and a comment line
    path = tmp_path / 'edges.txt'
    path.write_text('# u v\n10 20\n20 30\n30 10\n20 10\n40 40\n')
    ge = CSRGraph.from_edgelist_file(str(path))
    assert ge.labels == [10, 20, 30, 40] and ge.num_edges == 4              # 20-10 merged with 10-20
    assert _rows(ge.row_ptr, ge.col) == [[1, 2], [0, 2], [0, 1], [3]]
    pathw = tmp_path / 'edges_w.txt'
    pathw.write_text('1 2 0.5\n2 1 1.5\n2 3 2.0\n')
    gw = CSRGraph.from_edgelist_file(str(pathw), weighted=True)
    assert gw.num_edges == 2 and gw.weighted
    np.testing.assert_allclose(gw.w[gw.row_ptr[0]:gw.row_ptr[1]], [2.0])      # 0.5 + 1.5 merged
