"""
-m gpu: parity of the RolX half (graphrole_amd.roles, HIP NMF) with the reference's
``get_nmf_decomposition`` (sklearn NMF(solver='mu', init='nndsvda')) through golden factors the
reference produced with a fixed numpy seed (tests/golden/nmf_*.npz), plus the reference's own
unit tests for roles (tests/test_roles/*) restated against the drop-in classes.

Tolerance: factors within 1e-8 relative (max-norm per factor), equal
iteration count.  The reference's tests pin only shapes / non-negativity / unique counts.
"""
import numpy as np
import pandas as pd
import pytest

from tests import util

pytestmark = pytest.mark.gpu

FACTOR_RTOL = 1e-8


def _relmax(a, b):
    return np.abs(a - b).max() / np.abs(b).max()


@pytest.mark.parametrize('name', util.NMF_CASES)
def test_nmf_matches_reference_golden(name):
    from graphrole_amd.roles import factor
    g = util.load_nmf(name)
    X, r, seed = g['X'], int(g['r']), int(g['seed'])
    np.random.seed(seed)
    G, F, n_iter = factor.nmf_with_info(X, r)
    assert G.shape == g['W'].shape and F.shape == g['H'].shape
    assert n_iter == int(g['n_iter'])
    assert (G >= 0).all() and (F >= 0).all()
    assert _relmax(G, g['W']) < FACTOR_RTOL, _relmax(G, g['W'])
    assert _relmax(F, g['H']) < FACTOR_RTOL, _relmax(F, g['H'])


@pytest.mark.parametrize('name', ['rand500x12_r6', 'rand800x40_r6', 'rand3000x9_r2', 'er2000_r6', 'dw200_r5'])
def test_nndsvda_init_matches_reference_golden(name):
    from graphrole_amd import kernels as K
    from graphrole_amd.roles import factor
    g = util.load_nmf(name)
    X, r = g['X'], int(g['r'])
    n = X.shape[0]
    Xd = K.to_device(np.ascontiguousarray(X.T))
    W0, H0, xx = factor.nndsvda_init_device(Xd, n, r, g['omega'])          # one call below the ABI
    assert _relmax(K.to_host(W0)[:, :n].T, g['W0']) < 1e-9
    assert _relmax(K.to_host(H0), g['H0']) < 1e-9
    assert abs(xx - float((X * X).sum())) <= 1e-12 * xx
    # the per-kernel sequence the sharded path drives (same kernels, host algebra, no exchanges)
    W0o, H0o, xxo = factor._init_orchestrated(Xd, n, r, g['omega'])
    assert _relmax(K.to_host(W0o)[:, :n].T, g['W0']) < 1e-9
    assert _relmax(H0o, g['H0']) < 1e-9 and abs(xxo - xx) <= 1e-12 * xx


def test_nmf_consumes_global_rng_like_sklearn():
    """random_state=None in the reference (factor.py:19): one normal(size=(min(N,F), r+10)) draw."""
    from graphrole_amd.roles import factor
    X = np.abs(np.random.RandomState(3).randn(200, 9))
    np.random.seed(42)
    factor.get_nmf_decomposition(X, 3)
    after = np.random.rand()
    np.random.seed(42)
    np.random.normal(size=(9, 13))
    assert after == np.random.rand()


def test_nmf_rejects_negative_input():
    from graphrole_amd.roles import factor
    with pytest.raises(ValueError, match='Negative'):
        factor.get_nmf_decomposition(np.array([[1.0, -1.0], [2.0, 3.0]]), 1)


@pytest.mark.parametrize('n,F,r', [(3000, 125, 6), (1500, 260, 4)])
def test_nmf_more_than_120_features_matches_oracle(n, F, r):
    """BASELINE config 5 (directed, weighted, 8 attributes) yields 125 features: the wide kernels
    (column-group Gram, chunked fp64-MFMA W-pass) against the oracle's sklearn restatement."""
    from graphrole_amd.roles import factor
    from oracle import rolx
    rng = np.random.RandomState(F)
    base = np.abs(rng.randn(n, 12)) * np.linspace(1, 30, 12)
    mix = np.abs(rng.randn(12, F))
    X = base @ mix + 0.05 * np.abs(rng.randn(n, F))
    np.random.seed(3)
    G, Fm, n_iter = factor.nmf_with_info(X, r)
    np.random.seed(3)
    We, He, it = rolx.nmf(X, r)
    assert n_iter == it
    assert _relmax(G, We) < FACTOR_RTOL and _relmax(Fm, He) < FACTOR_RTOL


@pytest.mark.parametrize('n,F,r', [(3000, 24, 17), (2000, 40, 32), (1500, 130, 20)])
def test_nmf_more_than_16_roles_matches_oracle(n, F, r):
    """Ranks beyond one MFMA tile of roles (the reference accepts any n_roles, roles/extract.py:22-33): NNDSVDa
    (projection in two role chunks) + the composed multiplicative update against the oracle's sklearn restatement --
    same iteration count, factors to the tolerance of the fused path."""
    from graphrole_amd.roles import factor
    from oracle import rolx
    rng = np.random.RandomState(n + r)
    base = np.abs(rng.randn(n, r + 3)) * np.linspace(1, 20, r + 3)
    mix = np.abs(rng.randn(r + 3, F))
    X = base @ mix + 0.05 * np.abs(rng.randn(n, F))
    np.random.seed(5)
    G, Fm, n_iter = factor.nmf_with_info(X, r)
    np.random.seed(5)
    We, He, it = rolx.nmf(X, r)
    assert n_iter == it
    assert _relmax(G, We) < FACTOR_RTOL and _relmax(Fm, He) < FACTOR_RTOL


def test_role_extractor_with_24_roles():
    """The public call with a rank above 16: encoded factors of the right shape, roles assigned, and the limit
    (32) refused at construction."""
    from graphrole_amd import RoleExtractor
    import pandas as pd
    rng = np.random.RandomState(2)
    X = pd.DataFrame(np.abs(rng.randn(4000, 30)) @ np.abs(rng.randn(30, 36)))
    np.random.seed(0)
    rx = RoleExtractor(n_roles=24)
    rx.extract_role_factors(X)
    assert rx.node_role_factor.shape == (4000, 24) and rx.role_feature_factor.shape == (24, 36)
    assert len(rx.roles) == 4000
    with pytest.raises(ValueError, match='at most 32'):
        RoleExtractor(n_roles=33)
    with pytest.raises(ValueError, match='at most 32'):
        RoleExtractor(n_role_range=(2, 40))


def test_nmf_rank_deficient_features():
    """total_degree = in_degree + out_degree is an exact linear dependency (directed graphs)."""
    from graphrole_amd.roles import factor
    from oracle import rolx
    rng = np.random.RandomState(0)
    a, b = np.abs(rng.randn(3000)), np.abs(rng.randn(3000))
    X = np.column_stack([a, b, a + b, np.abs(rng.randn(3000, 4))])
    np.random.seed(1)
    G, F, n_iter = factor.nmf_with_info(X, 4)
    np.random.seed(1)
    We, He, it = rolx.nmf(X, 4)
    assert n_iter == it
    assert _relmax(G, We) < FACTOR_RTOL and _relmax(F, He) < FACTOR_RTOL


@pytest.mark.parametrize('n,F,r', [(1, 1, 1), (15, 3, 2), (16, 4, 4), (17, 5, 3), (1000, 16, 6), (1001, 17, 7),
                                   (4099, 20, 6), (5000, 33, 16), (3000, 64, 5), (2500, 100, 9), (2049, 120, 16),
                                   (3001, 121, 6), (2000, 125, 16), (1500, 250, 5), (900, 257, 12), (700, 480, 16),
                                   (1000, 5, 17), (4099, 20, 32), (2500, 100, 24), (1200, 130, 20), (600, 448, 32)])
def test_mu_iteration_kernels_vs_numpy(n, F, r):
    """One multiplicative update (sklearn _nmf.py:540-702, beta = 2) computed by grx_nmf_w_pass (fp64
    MFMA tiles; F > 120: the chunked wide kernel) + grx_nmf_h_update against numpy, over the shape
    limits (F <= 480, r <= 16 fused; 17 <= r <= 32 with r + F <= 480: the composed update), row
    counts that are not multiples of the 16-row sub-tile, zero rows / zero denominators, and a
    row range (sharded use)."""
    import torch
    from graphrole_amd import kernels as K
    eps = float(np.finfo(np.float32).eps)
    rng = np.random.default_rng(n * 131 + F * 7 + r)
    X = np.abs(rng.standard_normal((n, F))) * 10.0 ** rng.integers(-2, 3, size=F)
    W = np.abs(rng.standard_normal((n, r)))
    H = np.abs(rng.standard_normal((r, F)))
    if n > 20:
        X[5] = 0.0
        W[7] = 0.0                                     # zero denominator -> EPSILON (_nmf.py:583)
    ld = n + 3                                         # leading dimension larger than n
    Xd = torch.zeros((F, ld), dtype=torch.float64, device='cuda')
    Xd[:, :n] = torch.from_numpy(np.ascontiguousarray(X.T))
    Wd = torch.zeros((r, ld), dtype=torch.float64, device='cuda')
    Wd[:, :n] = torch.from_numpy(np.ascontiguousarray(W.T))
    st = K.NmfState(Xd, n, Wd.clone(), H)
    st.w_pass()
    st.h_update()
    den = W @ (H @ H.T)
    den[den == 0] = eps
    W1 = W * ((X @ H.T) / den)
    A, B = W1.T @ X, W1.T @ W1
    denh = B @ H
    denh[denh == 0] = eps
    H1 = H * (A / denh)
    got_W = st.W.cpu().numpy()[:, :n].T
    np.testing.assert_allclose(got_W, W1, rtol=1e-12, atol=1e-300)
    assert not st.W.cpu().numpy()[:, n:].any()         # padding columns untouched
    AB = st.AB.cpu().numpy()
    np.testing.assert_allclose(AB[:r * F].reshape(r, F), A, rtol=1e-11)
    np.testing.assert_allclose(AB[r * F:].reshape(r, r), B, rtol=1e-11)
    np.testing.assert_allclose(st.H.cpu().numpy(), H1, rtol=1e-10)
    # bitwise reproducible, and a row range updates exactly its rows
    st2 = K.NmfState(Xd, n, Wd.clone(), H)
    st2.w_pass()
    assert torch.equal(st2.W, st.W) and torch.equal(st2.AB, st.AB)
    if n > 100:
        rb, re = 37, n - 41
        st3 = K.NmfState(Xd, n, Wd.clone(), H)
        st3.w_pass(rb, re)
        w3 = st3.W.cpu().numpy()
        assert np.array_equal(w3[:, rb:re], st.W.cpu().numpy()[:, rb:re])
        assert np.array_equal(w3[:, :rb], Wd.cpu().numpy()[:, :rb]) and np.array_equal(w3[:, re:], Wd.cpu().numpy()[:, re:])
        A3 = W1[rb:re].T @ X[rb:re]
        np.testing.assert_allclose(st3.AB.cpu().numpy()[:r * F].reshape(r, F), A3, rtol=1e-11)


class TestFactorLikeReference:
    """tests/test_roles/test_factor.py of the reference."""

    def test_get_nmf_decomposition_shapes(self):
        from graphrole_amd.roles import factor
        np.random.seed(0)
        X = np.random.rand(20, 30)
        for n_roles in range(2, 8):
            G, F = factor.get_nmf_decomposition(X, n_roles)
            assert G.shape == (20, n_roles) and F.shape == (n_roles, 30)
            assert (G >= 0).all() and (F >= 0).all()

    def test_encode(self):
        from graphrole_amd.roles import factor
        np.random.seed(0)
        X = np.random.rand(20, 30)
        for n_bins in range(1, 8):
            assert len(np.unique(factor.encode(X, n_bins))) <= n_bins


class TestDescriptionLengthLikeReference:
    """tests/test_roles/test_description_length.py of the reference (host arithmetic)."""

    def test_costs(self):
        from graphrole_amd.roles import description_length as dl
        G = np.array([[1, 2, 3], [1, 2, 4]])
        F = np.array([[1, 2, 3], [4, 5, 5]])
        assert dl.get_encoding_cost((G, F)) == 3 * (G.size + F.size)
        np.random.seed(0)
        X = np.random.rand(20, 30)
        assert dl.get_error_cost(X, abs(X - np.random.randn(*X.shape))) > 0
        assert dl.get_error_cost(X, X) == 0
        assert len(dl.get_description_length_costs(X, (np.random.rand(20, 4), np.random.rand(4, 30)))) == 2


class TestRoleExtractorLikeReference:
    """tests/test_roles/test_extract.py of the reference."""

    def setup_method(self):
        np.random.seed(0)
        self.n_nodes, self.n_features = 20, 30
        names = [f'feature{i + 1}' for i in range(self.n_features)]
        self.features = pd.DataFrame(np.random.rand(self.n_nodes, self.n_features), columns=names,
                                     index=range(self.n_nodes))

    def test_init(self):
        from graphrole_amd import RoleExtractor
        re_ = RoleExtractor()
        assert re_.n_roles is None
        assert (re_.min_roles, re_.max_roles) == RoleExtractor.N_ROLE_RANGE == (2, 8)
        assert (re_.min_bits, re_.max_bits) == RoleExtractor.N_BIT_RANGE == (1, 8)
        assert RoleExtractor(n_roles=5).n_roles == 5
        re_ = RoleExtractor(n_role_range=(3, 5), n_bit_range=(2, 6))
        assert (re_.min_roles, re_.max_roles, re_.min_bits, re_.max_bits) == (3, 5, 2, 6)

    def test_extract_role_factors(self):
        from graphrole_amd import RoleExtractor
        for n_roles in range(2, 6):
            re_ = RoleExtractor(n_roles=n_roles)
            re_.extract_role_factors(self.features)
            roles = {f'role_{i}' for i in range(n_roles)}
            assert re_.node_role_factor.shape == (self.n_nodes, n_roles)
            assert re_.role_feature_factor.shape == (n_roles, self.n_features)
            assert set(re_.node_role_factor.index) == set(self.features.index)
            assert set(re_.node_role_factor.columns) == roles
            assert set(re_.role_feature_factor.index) == roles
            assert set(re_.role_feature_factor.columns) == set(self.features.columns)

    def test_roles_and_percentage(self):
        from graphrole_amd import RoleExtractor
        re_ = RoleExtractor()
        assert re_.roles is None and re_.role_percentage is None
        re_ = RoleExtractor(n_roles=3)
        re_.extract_role_factors(self.features)
        names = {f'role_{i}' for i in range(3)}
        assert set(re_.roles.keys()) == set(self.features.index)
        assert set(re_.roles.values()) <= names
        pct = re_.role_percentage
        assert set(pct.columns) == names
        assert np.allclose(pct.sum(axis=1).values, 1.0)

    def test_explain(self):
        from graphrole_amd import RoleExtractor
        with pytest.raises(NotImplementedError):
            RoleExtractor().explain()

    def test_select_model_picks_two_roles(self):
        from graphrole_amd import RoleExtractor
        re_ = RoleExtractor(n_role_range=(2, 5), n_bit_range=(2, 5))
        G, F = re_._select_model(self.features)
        assert G.shape[1] == F.shape[0] == 2                  # test_extract.py:81-88

    def test_get_encoded_role_factors(self):
        from graphrole_amd import RoleExtractor
        re_ = RoleExtractor()
        min_shape = min(self.features.shape)
        for n_roles in range(2, 4):
            for n_bits in range(1, 6):
                if 2 ** n_bits <= n_roles * min_shape:
                    G, F = re_._get_encoded_role_factors(self.features, n_roles, n_bits)
                    assert G.shape == (self.n_nodes, n_roles) and F.shape == (n_roles, self.n_features)
                    assert len(np.unique(G)) <= 2 ** n_bits and len(np.unique(F)) <= 2 ** n_bits
                else:
                    with pytest.raises(ValueError):
                        re_._get_encoded_role_factors(self.features, n_roles, n_bits)

    def test_rescale_costs(self):
        from graphrole_amd import RoleExtractor
        costs = np.full((3, 3), np.nan)
        costs[1, 1] = 0.37
        costs[2, :] = [0.1, 0.5, 0.9]
        out = RoleExtractor._rescale_costs(costs)
        assert np.isnan(out[0]).all()
        assert np.isnan(out[1, 0]) and np.isnan(out[1, 2]) and out[1, 1] == pytest.approx(1.0)
        assert np.linalg.norm(out[2]) == pytest.approx(1.0)


def test_end_to_end_karate_roles():
    """BASELINE config 1 plumbing: karate club, ReFeX then RolX with MDL model selection."""
    import networkx as nx
    from graphrole_amd import RecursiveFeatureExtractor, RoleExtractor
    g = util.load_refex('karate')
    labels = g.js('labels')
    G = nx.Graph()
    G.add_nodes_from(labels)
    G.add_edges_from((labels[s], labels[d]) for s, d in zip(g['src'], g['dst']))
    features = RecursiveFeatureExtractor(G).extract_features()
    assert features.shape == (34, 7)
    np.random.seed(0)
    re_ = RoleExtractor(n_roles=None)
    re_.extract_role_factors(features)
    k = re_.node_role_factor.shape[1]
    assert 2 <= k <= 7
    assert set(re_.roles.keys()) == set(labels)
    assert np.allclose(re_.role_percentage.sum(axis=1).values, 1.0)


@pytest.mark.parametrize('F,r', [(9, 4), (40, 20)])
def test_device_mdl_costs_equal_host_formulas(F, r):
    """Encoding / error cost computed in HBM (grx_lloyd_max info, grx_nmf_kl_cost) against the host
    restatement of graphrole/roles/description_length.py:32-61 on the same encoded factors (r = 20: the kernels'
    instantiation for more than 16 roles)."""
    from graphrole_amd import kernels as K
    from graphrole_amd.roles import description_length as dl
    from graphrole_amd.roles import factor
    rng = np.random.RandomState(4)
    V = np.abs(rng.randn(5000, F)) * np.linspace(1, 12, F)
    V[rng.rand(*V.shape) < 0.1] = 0.0                         # zero entries are masked by the KL cost
    Vd = K.to_device(np.ascontiguousarray(V.T))
    np.random.seed(3)
    state, Wq, Hq, uniq_g, uniq_f = factor.encoded_factors_device(Vd, V, r, 5)
    G, F = K.to_host(Wq).T, K.to_host(Hq)
    assert uniq_g == len(np.unique(G)) and uniq_f == len(np.unique(F))
    enc_host, err_host = dl.get_description_length_costs(V, (G, F))
    assert enc_host == np.ceil(np.log2(max(uniq_g, uniq_f))) * (G.size + F.size)
    np.testing.assert_allclose(state.kl_cost(Wq, Hq), err_host, rtol=1e-10)
    # row-range partials add up (multi-GPU all-reduce SUM)
    np.testing.assert_allclose(state.kl_cost(Wq, Hq, 0, 1234) + state.kl_cost(Wq, Hq, 1234, 5000), err_host,
                               rtol=1e-10)


def test_model_selection_on_a_larger_matrix_runs_in_hbm():
    """n_roles=None on 50k x 10 features: 7 x 8 grid of NMF + 2 quantisations + MDL costs."""
    import time
    from graphrole_amd import RoleExtractor
    rng = np.random.RandomState(0)
    base = np.abs(rng.randn(50000, 3))
    feats = pd.DataFrame(np.abs(base @ np.abs(rng.randn(3, 10)) + 0.05 * rng.randn(50000, 10)),
                         columns=[f'f{i}' for i in range(10)])
    np.random.seed(0)
    re_ = RoleExtractor()
    t0 = time.perf_counter()
    re_.extract_role_factors(feats)
    dt = time.perf_counter() - t0
    k = re_.node_role_factor.shape[1]
    assert 2 <= k <= 8 and re_.role_feature_factor.shape == (k, 10)
    assert np.allclose(re_.role_percentage.sum(axis=1).values, 1.0)
    assert dt < 60


# ---------------------------------------------------------------- model selection vs the reference (fixtures)
# (n_roles, n_bits) cells of the MDL grid whose k-means++ seeding meets candidates of mathematically EQUAL potential:
# sklearn's pick there is decided by the last-bit rounding of its BLAS dot product, not by the algorithm
TIED_CELLS = {
    'karate': [], 'er300': [], 'ba300': [],
    'karate_weighted': [(6, 4)], 'dw200_attrs': [(4, 8), (7, 8), (8, 7), (8, 8)], 'directed120': [(8, 6)],
    'loops_dangling150': [(8, 6)],
}
def _selection_record(name, quantizer):
    """(reference grid + selection, ours) for one golden feature table."""
    from graphrole_amd import RoleExtractor
    ref = util.load_roles(name)
    g = util.load_refex(name)
    X = pd.DataFrame(g['final_values'], index=g.js('labels'), columns=g.js('final_columns'))
    np.random.seed(int(ref['seed']))
    rx = RoleExtractor()
    rx.quantizer = quantizer
    rx.extract_role_factors(X)
    return ref, rx, X


def _record(path_name, name, record):
    import json
    import os
    out_dir = os.path.join(os.environ.get('GRAFT_REPO_ROOT', util.ROOT), 'gpurun_out')
    os.makedirs(out_dir, exist_ok=True)
    path = os.path.join(out_dir, path_name)
    table = json.load(open(path)) if os.path.exists(path) else {}
    table[name] = record
    json.dump(table, open(path, 'w'), indent=1, sort_keys=True)


@pytest.mark.parametrize('name', util.ROLES_CASES)
def test_model_selection_vs_reference(name):
    """RoleExtractor(n_roles=None) with the reference's quantiser reproduced (quantizer='kmeans', the default)
    against the MDL grid the reference computed for the same table (tools/make_golden_roles.py;
    graphrole/roles/extract.py:98-142): the same cells are skipped, every cell's encoding cost is EQUAL, and the
    error cost agrees to 1e-6 in all cells but a few.  The few: k-means++ on a small factor often meets two
    candidate seeds with mathematically EQUAL potentials (two isolated entries that only capture themselves and
    each other); sklearn's choice between them is decided by the last-bit rounding of its BLAS dot product, i.e.
    it is not a property of the algorithm (it changes with the BLAS build) and cannot be reproduced -- such a cell
    ends with another, equally good, seed set and an error cost that differs by 1e-4 .. a few percent.  The
    selected cell is recorded next to the reference's (gpurun_out/model_selection.json -> profiles/)."""
    ref, rx, X = _selection_record(name, 'kmeans')
    ours = rx.model_selection_
    enc_r, err_r = ref['encoding_costs'], ref['error_costs']
    enc_o, err_o = ours['encoding_costs'], ours['error_costs']
    assert np.array_equal(np.isnan(enc_o), np.isnan(enc_r)) and np.array_equal(np.isnan(err_o), np.isnan(err_r))
    live = ~np.isnan(enc_r)
    assert np.array_equal(enc_o[live], enc_r[live])
    rel_grid = np.abs(err_o - err_r) / np.abs(err_r)
    rel = rel_grid[live]
    same = rel <= 1e-6
    tied = sorted([int(r), int(b)] for r, b in np.argwhere(live & ~(rel_grid <= 1e-6)))
    record = {'table': name, 'cells': int(live.sum()), 'cells_equal_1e-6': int(same.sum()),
              'max_rel_diff_of_the_others': float(rel[~same].max()) if (~same).any() else 0.0,
              'cells_that_differ': tied,
              'reference_selected': [int(v) for v in ref['selected']], 'ours_selected': list(ours['selected'])}
    _record('model_selection.json', name, record)
    # the cells whose seeding meets mathematically tied candidates are KNOWN (an explicit allow-list, observed on
    # MI355X and stable: the draws do not depend on the data); every other cell must agree to 1e-6, the tied ones to
    # 10 x the largest deviation ever observed (0.26 %), and the reference's cell is selected on every table
    allowed = [list(c) for c in TIED_CELLS[name]]
    assert all(c in allowed for c in tied), record
    assert rel.max() < 0.03, record
    assert list(ours['selected']) == [int(v) for v in ref['selected']], record
    if list(ours['selected']) not in tied:
        scale = np.abs(ref['node_role_factor']).max()
        assert np.abs(rx.node_role_factor.values - ref['node_role_factor']).max() <= 1e-7 * scale
        assert np.abs(rx.role_feature_factor.values - ref['role_feature_factor']).max() <= \
            1e-7 * np.abs(ref['role_feature_factor']).max()


@pytest.mark.parametrize('name', util.ROLES_CASES)
def test_fixed_rank_role_factors_equal_reference(name):
    """RoleExtractor(n_roles=3) (roles/extract.py:69-77) against the reference's encoded factors for the same
    table: NMF + KMeans quantisation of both factors, to rounding (2e-13 of the factor's scale)."""
    from graphrole_amd import RoleExtractor
    ref = util.load_roles(name)
    g = util.load_refex(name)
    X = pd.DataFrame(g['final_values'], index=g.js('labels'), columns=g.js('final_columns'))
    np.random.seed(int(ref['seed']))
    rx3 = RoleExtractor(n_roles=3)
    rx3.extract_role_factors(X)
    G, F = rx3.node_role_factor.values, rx3.role_feature_factor.values
    assert len(np.unique(G)) == len(np.unique(ref['fixed3_node_role_factor']))
    assert len(np.unique(F)) == len(np.unique(ref['fixed3_role_feature_factor']))
    dG = np.abs(G - ref['fixed3_node_role_factor']).max() / np.abs(ref['fixed3_node_role_factor']).max()
    dF = np.abs(F - ref['fixed3_role_feature_factor']).max() / np.abs(ref['fixed3_role_feature_factor']).max()
    _record('fixed_rank_factors.json', name, {'node_role_max_rel_diff': float(dG), 'role_feature_max_rel_diff': float(dF)})
    assert dG < 2e-13 and dF < 2e-13            # observed <= 2e-14 on all seven tables (profiles/r02_fixed_rank_factors.json)


@pytest.mark.parametrize('name', util.ROLES_CASES)
def test_model_selection_with_the_lloyd_max_quantizer(name):
    """quantizer='lloyd_max' (the deterministic Lloyd-Max solver) is NOT the reference's quantiser: the same
    cells are skipped and the encoding cost never exceeds the reference's, but the KL error cost of a cell
    differs in either direction and the selected cell can move by one role.  The table is recorded in
    gpurun_out/model_selection_lloyd_max.json (copied to profiles/)."""
    ref, rx, _ = _selection_record(name, 'lloyd_max')
    ours = rx.model_selection_
    enc_r, err_r = ref['encoding_costs'], ref['error_costs']
    enc_o, err_o = ours['encoding_costs'], ours['error_costs']
    assert np.array_equal(np.isnan(enc_o), np.isnan(enc_r)) and np.array_equal(np.isnan(err_o), np.isnan(err_r))
    live = ~np.isnan(enc_r)
    assert np.all(enc_o[live] <= enc_r[live])
    rel = (err_o[live] - err_r[live]) / np.abs(err_r[live])
    record = {'table': name, 'reference_selected': [int(v) for v in ref['selected']],
              'lloyd_max_selected': list(ours['selected']), 'error_cost_rel_diff_max': float(rel.max()),
              'error_cost_rel_diff_min': float(rel.min()), 'error_cost_rel_diff_median': float(np.median(rel)),
              'cells': int(live.sum())}
    _record('model_selection_lloyd_max.json', name, record)
    assert abs(float(np.median(rel))) < 0.05, record
    assert abs(ours['selected'][0] - int(ref['selected'][0])) <= 1 and abs(ours['selected'][1] - int(ref['selected'][1])) <= 1


# ---------------------------------------------------------------- stopping rule under adversarial margins
def _stop_rule_case(seed):
    """Matrices of tools-style families; the seeds below were picked (offline search over 1500 seeds with the
    oracle) because their STOPPING decision (prev - err) / err_init < 1e-4 is made with the smallest margins:
    1e-8 .. 2e-7 away from the tolerance at some convergence check."""
    rng = np.random.RandomState(seed)
    n = int(rng.choice([300, 800, 2000]))
    F = int(rng.choice([8, 12, 20, 30]))
    r = int(rng.choice([2, 3, 4, 6]))
    kind = seed % 4
    if kind == 0:
        X = np.abs(rng.randn(n, F)) * np.linspace(1, 20, F)
    elif kind == 1:
        X = np.abs(rng.randn(n, r)) @ np.abs(rng.randn(r, F)) + (10.0 ** -rng.randint(1, 7)) * np.abs(rng.randn(n, F))
    elif kind == 2:
        X = rng.gamma(0.5, 2.0, (n, F))
    else:
        X = np.abs(rng.randn(n, r)) @ np.abs(rng.randn(r, F))               # exactly rank r: near-exact fit
    omega = rng.normal(size=(F, r + 10))
    return X, r, omega


@pytest.mark.parametrize('seed', [220, 1262, 1452, 1044, 1429, 666, 1032, 1288, 994, 1050,     # smallest margins
                                  3, 7, 11, 15, 19, 23,                                        # exactly rank r
                                  1, 5, 9, 13, 17, 21])                                        # rank r + noise 1e-1..1e-6
def test_stopping_rule_adversarial(seed):
    """The convergence checks read ||X - WH|| from the trace identity of the W-pass outputs and fall back to the
    direct pass when it cancels (grx_nmf_mu, csrc/grx_fit.hip).  Same stopping iteration as the oracle's direct
    residual (sklearn _nmf.py:872-885) where the decision is closest to the tolerance, on near-exact fits (the
    identity must NOT be trusted there), and through both drivers of the loop."""
    from graphrole_amd import kernels as K
    from graphrole_amd.roles import factor
    from oracle import rolx
    X, r, omega = _stop_rule_case(seed)
    n = X.shape[0]
    W0, H0 = rolx.nndsvda_init(X, r, omega)
    We, He, it = rolx.mu_iterations(X, W0, H0)
    Xd = K.to_device(np.ascontiguousarray(X.T))
    xx = float((X * X).sum())
    for driver in ('grx_nmf_mu', 'per_kernel'):
        state = K.NmfState(Xd, n, K.to_device(np.ascontiguousarray(W0.T)), H0, x_sq_norm=xx)
        if driver == 'grx_nmf_mu':
            _, n_iter = factor.run_mu_loop(state)
        else:
            n_iter = factor._mu_orchestrated(state, factor.NMF_TOL, factor.NMF_MAX_ITER, None)
        assert n_iter == it, (driver, seed, n_iter, it)
        assert _relmax(K.to_host(state.W)[:, :n].T, We) < 1e-7
        assert _relmax(K.to_host(state.H), He) < 1e-7
    if seed % 4 == 3:
        # exactly rank r: relative residual far below 1e-4 -> every check after the first ones ran the direct pass
        assert state.x_sq_norm == xx
        st2 = K.NmfState(Xd, n, K.to_device(np.ascontiguousarray(W0.T)), H0, x_sq_norm=xx)
        factor.run_mu_loop(st2)
        rel = np.linalg.norm(X - K.to_host(st2.W)[:, :n].T @ K.to_host(st2.H)) / np.linalg.norm(X)
        if rel < 1e-5:
            assert st2.info.direct_residuals >= 1


# ---------------------------------------------------------------- conditioning of the initialisation
def _ill_conditioned(kind, seed):
    """Feature tables that stress the Gram-matrix + eigh whitening of the NNDSVDa start (it squares the condition
    number of X; sklearn's range finder works on X itself): near-duplicate columns and extreme column scales."""
    rng = np.random.RandomState(seed)
    n, F, r = 4000, 14, 6
    base = np.abs(rng.randn(n, r)) @ np.abs(rng.randn(r, F)) + 0.05 * np.abs(rng.randn(n, F))
    if kind == 'near_duplicate_columns':
        # two pairs of columns equal to 1e-7 relative: kappa(X) ~ 1e7, kappa(X^T X) ~ 1e14
        base[:, 5] = base[:, 2] * (1.0 + 1e-7 * rng.randn(n))
        base[:, 11] = base[:, 7] * (1.0 + 1e-7 * rng.randn(n))
    elif kind == 'scale_spread_1e10':
        base = base * np.logspace(-5, 5, F)                           # column scales from 1e-5 to 1e5
    elif kind == 'both':
        base[:, 5] = base[:, 2] * (1.0 + 1e-7 * rng.randn(n))
        base = base * np.logspace(-5, 5, F)
    elif kind == 'degree_like':
        # what ReFeX tables look like: heavy-tailed integer-ish columns, one a near-multiple of another
        base = np.floor(rng.pareto(1.5, (n, F)) * 10.0)
        base[:, 3] = 2.0 * base[:, 1] + (rng.rand(n) < 1e-3)
    return np.ascontiguousarray(base), r


@pytest.mark.parametrize('kind', ['near_duplicate_columns', 'scale_spread_1e10', 'both', 'degree_like'])
@pytest.mark.parametrize('seed', [0, 1, 2])
def test_nmf_on_ill_conditioned_tables_matches_oracle(kind, seed):
    """equal iteration count and factors within the stated 1e-8 of oracle.rolx.nmf (sklearn's procedure restated:
    randomized range finder on X itself) on tables whose Gram matrix has a condition number up to ~1e14 and beyond"""
    from graphrole_amd.roles import factor
    from oracle import rolx
    X, r = _ill_conditioned(kind, seed)
    omega = np.random.RandomState(100 + seed).normal(size=(X.shape[1], r + 10))
    # the same Gaussian test matrix on both sides (the reference draws it from numpy's global stream)
    from graphrole_amd import kernels as K
    Xd = K.to_device(np.ascontiguousarray(X.T))
    st, n_iter = factor.nmf_device(Xd, X.shape[0], r, omega)
    G = K.to_host(st.W)[:, :X.shape[0]].T
    Fm = K.to_host(st.H)
    We, He, it = rolx.nmf(X, r, omega)
    cond = np.linalg.cond(X)
    _record('nmf_conditioning.json', f'{kind}_{seed}', {'cond_X': float(cond), 'n_iter': int(n_iter), 'oracle_n_iter': int(it),
                                                       'W_rel': float(_relmax(G, We)), 'H_rel': float(_relmax(Fm, He))})
    assert n_iter == it, (n_iter, it, cond)
    assert _relmax(G, We) < FACTOR_RTOL and _relmax(Fm, He) < FACTOR_RTOL, (cond, _relmax(G, We), _relmax(Fm, He))


@pytest.mark.parametrize('seed', [0, 1, 2])
@pytest.mark.parametrize('shape', [(18, 17, 17), (40, 17, 12), (3000, 20, 6)])
def test_nmf_on_graded_tables_with_full_rank_request(shape, seed):
    """column norms six decades apart (what ReFeX tables look like) and up to r = F roles: every singular direction the
    NNDSVDa start needs survives the Gram whitening (exact power-of-two column equilibration, csrc/grx_host_linalg.hip);
    found by tools/fuzz_rolx.py (FUZZ_WIDE_RANK=1, seed 403, case 103: the start differed in its seventeenth direction)"""
    from graphrole_amd.roles import factor
    from oracle import rolx
    n, F, r = shape
    rng = np.random.RandomState(100 * seed + n)
    X = np.abs(rng.randn(n, F)) * 10.0 ** rng.uniform(-2, 4, F)
    np.random.seed(seed)
    G, H, n_iter = factor.nmf_with_info(X, r)
    np.random.seed(seed)
    We, He, it = rolx.nmf(X, r)
    assert n_iter == it
    assert _relmax(G, We) < 1e-7 and _relmax(H, He) < 1e-7, (_relmax(G, We), _relmax(H, He))


def test_read_backs_through_the_copy_engine_give_the_same_results():
    """The values the host decides on (distance matrix of the pruner, Gram matrices, residuals) reach it through mapped
    host memory and a flag (csrc/grx_runtime.hip grx_fetch_begin / grx_fetch_wait); GRX_READBACK=memcpy sends them
    through hipMemcpyAsync + hipStreamSynchronize as before round 6.  Same bytes either way: the golden NMF factors, the
    stopping rule and the end-to-end run must pass unchanged, and so must the generation loop on the reference's tables."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, GRX_READBACK='memcpy')
    res = subprocess.run([sys.executable, '-m', 'pytest', os.path.join(root, 'tests', 'test_gpu_rolx.py'),
                          os.path.join(root, 'tests', 'test_gpu_refex.py'), '-q', '-m', 'gpu', '-x', '-p', 'no:cacheprovider', '-k',
                          'nmf_matches_reference_golden or stopping_rule or end_to_end or fixed_rank or golden or reference'],
                         capture_output=True, text=True, timeout=1500, env=env, cwd=root)
    assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-2000:]
    assert ' passed' in res.stdout
