"""
-m "not gpu": the N > 1 path (node-range sharding, one exchange per generation, column-sharded
binning, all-reduce of the Chebyshev matrix and of the NMF partial sums) run as TWO gloo ranks on
CPU with the test double tests/fake_kernels.py as kernel backend, checked against the
single-process result and the reference's golden vectors.
"""
import os
import socket

import numpy as np
import pytest
import torch.multiprocessing as mp

from tests import util



def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, port, case, out_dir, WORLD):
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=WORLD)
    try:
        from graphrole_amd import RecursiveFeatureExtractor, backend
        from graphrole_amd.graph import CSRGraph
        from graphrole_amd.roles import factor
        from tests import fake_kernels
        backend.use(fake_kernels)
        g = util.load_refex(case)
        w = g['w'] if len(g['w']) else None
        G = CSRGraph(int(g['n']), g['src'], g['dst'], weights=w, directed=bool(g['directed']))
        fe = RecursiveFeatureExtractor(G, max_generations=int(g['max_generations']), distributed=True,
                                      aggs=util.golden_aggs(g))
        X = fe.extract_features()
        plan = fe._shard()
        assert plan is not None and plan.world == WORLD
        assert 0 < plan.row_end - plan.row_begin < G.n
        # sharded NMF: W rows split, AB all-reduced every iteration
        K = backend.get()
        Xd = K.to_device(np.ascontiguousarray(X.values.astype(float).T))
        omega = np.random.RandomState(5).normal(size=(X.shape[1], 4 + 10))
        state, n_iter = factor.nmf_device(Xd, G.n, 4, omega, plan=plan)
        W = K.to_host(state.W)[:, :G.n]
        np.savez(os.path.join(out_dir, f'rank{rank}.npz'), X=X.values.astype(float), cols=np.array(list(X.columns)),
                 gen=fe.generation_count, W=W, H=K.to_host(state.H), n_iter=n_iter,
                 rb=plan.row_begin, re=plan.row_end)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('case,WORLD', [('er300', 2), ('ba2000', 2), ('er300', 3), ('directed120', 3), ('karate', 7),
                                        ('ba300_stdvar', 2), ('loops_dangling150_minmax', 3)])
def test_sharded_pipeline_equals_single_process(case, WORLD, tmp_path):
    """2, 3 and 7 ranks (3: candidate counts that do not divide by the world size -> uneven column
    ownership in the owner all-to-all; 7 > the 6 candidates of generation 1: ranks that own no column
    and send / receive zero-length messages)."""
    port = _free_port()
    mp.spawn(_worker, args=(port, case, str(tmp_path), WORLD), nprocs=WORLD, join=True)
    r0 = np.load(tmp_path / 'rank0.npz')
    r1 = np.load(tmp_path / f'rank{WORLD - 1}.npz')
    g = util.load_refex(case)
    # both ranks hold the full, identical feature table, equal to the reference's
    assert list(r0['cols']) == list(r1['cols']) == g.js('final_columns')
    assert int(r0['gen']) == int(g['generation_count'])
    assert np.array_equal(r0['X'], r1['X'])
    np.testing.assert_allclose(r0['X'], g['final_values'], rtol=1e-12, atol=0)
    # NMF: H replicated and identical, iteration counts equal, W rows owned by each rank agree
    # with a single-process run
    assert int(r0['n_iter']) == int(r1['n_iter'])
    assert np.array_equal(r0['H'], r1['H'])
    from graphrole_amd import backend
    from graphrole_amd.roles import factor
    from tests import fake_kernels
    backend.use(fake_kernels)
    try:
        K = backend.get()
        X = r0['X']
        Xd = K.to_device(np.ascontiguousarray(X.T))
        omega = np.random.RandomState(5).normal(size=(X.shape[1], 14))
        state, n_iter = factor.nmf_device(Xd, X.shape[0], 4, omega)
        Wref, Href = K.to_host(state.W), K.to_host(state.H)
    finally:
        backend.use(None)
    assert n_iter == int(r0['n_iter'])
    np.testing.assert_allclose(r0['H'], Href, rtol=1e-9)
    for r in (r0, r1):
        rb, re = int(r['rb']), int(r['re'])
        np.testing.assert_allclose(r['W'][:, rb:re], Wref[:, rb:re], rtol=1e-9, atol=1e-15)


def test_shard_plan_balances_edges_not_nodes():
    """Bounds follow nnz + n, so a power-law graph's hub-heavy prefix gets fewer rows."""
    import torch.distributed as dist
    from graphrole_amd.parallel import ShardPlan
    port = _free_port()
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=0, world_size=1)
    try:
        deg = np.concatenate([np.full(100, 1000), np.full(9900, 10)])
        row_ptr = np.concatenate([[0], np.cumsum(deg)]).astype(np.int64)
        plan = ShardPlan(row_ptr)
        assert plan.world == 1 and (plan.row_begin, plan.row_end) == (0, 10000)
        plan.world, plan.rank = 4, 0                     # inspect the cut points of a 4-way split
        work = row_ptr[1:] + np.arange(1, 10001)
        cuts = [int(np.searchsorted(work, work[-1] * p / 4)) for p in range(1, 4)]
        assert cuts[0] < 2500                            # first rank owns the hubs -> fewer rows
    finally:
        dist.destroy_process_group()


def _roles_worker(rank, port, case, out_dir, WORLD):
    import pandas as pd
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=WORLD)
    try:
        from graphrole_amd import RoleExtractor, backend
        from tests import fake_kernels
        backend.use(fake_kernels)
        g = util.load_refex(case)
        ref = util.load_roles(case)
        X = pd.DataFrame(g['final_values'], index=g.js('labels'), columns=g.js('final_columns'))
        out = {}
        for tag, kwargs in (('grid', dict(n_role_range=(2, 4), n_bit_range=(1, 4))), ('fixed', dict(n_roles=3))):
            for mode, distributed in (('sharded', True), ('single', None)):
                if mode == 'single' and rank != 0:
                    continue
                np.random.seed(int(ref['seed']))
                rx = RoleExtractor(distributed=distributed, **kwargs)
                rx.extract_role_factors(X)
                out[f'{tag}_{mode}_G'] = rx.node_role_factor.values
                out[f'{tag}_{mode}_F'] = rx.role_feature_factor.values
                if tag == 'grid':
                    out[f'{tag}_{mode}_sel'] = np.array(rx.model_selection_['selected'])
                    out[f'{tag}_{mode}_err'] = rx.model_selection_['error_costs']
        np.savez(os.path.join(out_dir, f'roles{rank}.npz'), **out)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('case,WORLD', [('er300', 2), ('karate', 3)])
def test_sharded_role_extraction_equals_single_process(case, WORLD, tmp_path):
    """RoleExtractor(distributed=True): the row passes of every factorisation of the MDL grid and the KL error cost
    of every cell run on the rank's rows and are summed over the ranks; every rank ends with the factors and the
    cost grids of a single-process fit (same numpy seed on every rank)."""
    mp.spawn(_roles_worker, args=(_free_port(), case, str(tmp_path), WORLD), nprocs=WORLD, join=True)
    r0 = np.load(tmp_path / 'roles0.npz')
    rl = np.load(tmp_path / f'roles{WORLD - 1}.npz')
    for tag in ('grid', 'fixed'):
        for part in ('G', 'F'):
            a, b, s = r0[f'{tag}_sharded_{part}'], rl[f'{tag}_sharded_{part}'], r0[f'{tag}_single_{part}']
            # (the CPU test double quantises with sklearn's threaded KMeans: centres are equal to an ulp, not bitwise)
            np.testing.assert_allclose(a, b, rtol=1e-13, atol=0, err_msg='ranks disagree')
            assert a.shape == s.shape
            np.testing.assert_allclose(a, s, rtol=1e-9, atol=1e-12 * np.abs(s).max())
    assert list(r0['grid_sharded_sel']) == list(rl['grid_sharded_sel']) == list(r0['grid_single_sel'])
    np.testing.assert_allclose(r0['grid_sharded_err'], r0['grid_single_err'], rtol=1e-9, equal_nan=True)
