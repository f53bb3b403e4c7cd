"""
-m gpu: the N > 1 code path on the real kernels.  A gpurun box has ONE GPU, so two ranks share
cuda:0 and talk over gloo (ShardPlan stages device tensors through the host for that backend);
what is exercised is every row-range / column-split argument of the HIP entry points
(grx_row_sums, grx_triangle_counts + all-reduce, grx_egonet_*, grx_aggregate, column-sharded
grx_vertical_log_bin, row-sharded grx_chebyshev, grx_nmf_w_pass + all-reduce) and that the ranks
end up with the table a single process computes -- bit for bit for ReFeX.
"""
import os
import socket

import numpy as np
import pytest
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

WORLD = 2


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _nmf_rank(kind):
    """'<graph>_r<k>' factors with k roles (ranks above 16 take the composed update); the default is 4"""
    return int(kind.rsplit('_r', 1)[1]) if '_r' in kind else 4


def _graph(kind):
    from graphrole_amd import synth
    kind = kind.rsplit('_r', 1)[0] if '_r' in kind else kind
    if kind == 'ba':
        return synth.ba_graph(60_000, 8, seed=3)
    if kind == 'ba1m':
        # BASELINE config 4's graph (the 1 M / 10 M power-law graph of config 3, node-range sharded)
        return synth.ba_graph(1_000_000, 10, seed=0)
    if kind == 'dw1m':
        # BASELINE config 5's shape (weighted directed power-law + 8 attributes) at 1 M nodes / 10 M arcs
        return synth.directed_weighted_graph(1_000_000, 10_000_000, seed=0)
    return synth.directed_weighted_graph(40_000, 400_000, seed=4)


def _worker(rank, port, kind, driver, out_dir, world=WORLD):
    import torch
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    torch.cuda.set_device(0)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from graphrole_amd import RecursiveFeatureExtractor, kernels as K
        from graphrole_amd.roles import factor
        G = _graph(kind)
        # 'native': grx_refex_run / grx_nmf_fit with the plan's communicator (the staged callback transport under
        # gloo) -- the exchanges are issued by the C++ drivers; 'per_kernel': the Python driver over the same C
        # exchange entry points
        fe = RecursiveFeatureExtractor(G, max_generations=4, distributed=True, native_loop=driver == 'native',
                                      attributes=bool(G.attributes))
        X = fe.extract_features()
        plan = fe._shard()
        assert plan is not None and plan.world == world and 0 < plan.row_end - plan.row_begin < G.n
        assert plan.comm() is not None
        Xd = K.gather_columns(fe.device_features()[1], G.n)
        F = X.shape[1]
        rank_ = _nmf_rank(kind)
        omega = np.random.RandomState(5).normal(size=(F, rank_ + 10))
        state, n_iter = factor.nmf_device(Xd, G.n, rank_, omega, plan=plan)
        out = dict(X=X.values.astype(float), cols=np.array(list(X.columns)), gen=fe.generation_count,
                   W=K.to_host(state.W)[:, :G.n], H=K.to_host(state.H), n_iter=n_iter,
                   rb=plan.row_begin, re=plan.row_end)
        if rank == 0:
            # the single-GPU answer, same process, same kernels
            fe1 = RecursiveFeatureExtractor(G, max_generations=4, attributes=bool(G.attributes))
            X1 = fe1.extract_features()
            Xd1 = K.gather_columns(fe1.device_features()[1], G.n)
            s1, it1 = factor.nmf_device(Xd1, G.n, rank_, omega)
            out.update(X1=X1.values.astype(float), cols1=np.array(list(X1.columns)), W1=K.to_host(s1.W)[:, :G.n],
                       H1=K.to_host(s1.H), it1=it1)
        np.savez(os.path.join(out_dir, f'rank{rank}.npz'), **out)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('kind,driver', [('ba', 'native'), ('ba', 'per_kernel'), ('directed_weighted', 'native'),
                                         ('directed_weighted', 'per_kernel'), ('ba1m', 'native'), ('dw1m', 'native'),
                                         ('directed_weighted_r20', 'native'), ('directed_weighted_r20', 'per_kernel')])
def test_two_ranks_one_gpu_equal_single_process(kind, driver, tmp_path):
    mp.spawn(_worker, args=(_free_port(), kind, driver, str(tmp_path)), nprocs=WORLD, join=True)
    r0, r1 = np.load(tmp_path / 'rank0.npz'), np.load(tmp_path / 'rank1.npz')
    assert list(r0['cols']) == list(r1['cols']) == list(r0['cols1'])
    assert np.array_equal(r0['X'], r1['X'])
    assert np.array_equal(r0['X'], r0['X1'])                       # ReFeX: bit-exact vs one GPU
    assert int(r0['n_iter']) == int(r1['n_iter']) == int(r0['it1'])
    assert np.array_equal(r0['H'], r1['H'])
    np.testing.assert_allclose(r0['H'], r0['H1'], rtol=1e-9)       # partial sums combine in another order
    for r in (r0, r1):
        rb, re = int(r['rb']), int(r['re'])
        # entries far below the factor's scale carry the absolute error of the re-ordered partial sums
        np.testing.assert_allclose(r['W'][:, rb:re], r0['W1'][:, rb:re], rtol=1e-9, atol=1e-12 * np.abs(r0['W1']).max())


@pytest.mark.parametrize('kind,driver', [('ba', 'native'), ('directed_weighted', 'native'), ('ba', 'per_kernel')])
def test_eight_ranks_one_gpu_equal_single_process(kind, driver, tmp_path):
    """P = 8 -- the partition of BASELINE configs 4 / 5 -- on the real kernels: eight processes share cuda:0 (callback
    transport).  Generation 0 has 3 (BA) columns for 8 column owners, so five ranks own NO column of the first
    binning exchange, and the nnz-balanced row cuts are uneven on the power-law graph; every rank must end with the
    single-process table, bit for bit."""
    world = 8
    mp.spawn(_worker, args=(_free_port(), kind, driver, str(tmp_path), world), nprocs=world, join=True)
    ranks = [np.load(tmp_path / f'rank{q}.npz') for q in range(world)]
    r0 = ranks[0]
    sizes = [int(r['re']) - int(r['rb']) for r in ranks]
    assert sum(sizes) == r0['X'].shape[0] and [int(r['rb']) for r in ranks[1:]] == [int(r['re']) for r in ranks[:-1]]
    if kind == 'ba':
        assert max(sizes) > 2 * min(sizes)                   # hubs first: the first ranks hold few, long rows
    for r in ranks:
        assert list(r['cols']) == list(r0['cols1'])
        assert np.array_equal(r['X'], r0['X1'])                    # ReFeX: bit-exact vs one GPU
        assert int(r['n_iter']) == int(r0['it1'])
        assert np.array_equal(r['H'], r0['H'])
        rb, re = int(r['rb']), int(r['re'])
        np.testing.assert_allclose(r['W'][:, rb:re], r0['W1'][:, rb:re], rtol=1e-9, atol=1e-12 * np.abs(r0['W1']).max())
    np.testing.assert_allclose(r0['H'], r0['H1'], rtol=1e-9)


def test_bench_launches_itself_on_two_ranks(tmp_path):
    """`python bench.py --gpus 2` with NO launcher (how a driver calls N = 1): the script re-executes itself under
    torch.distributed.run, the synthetic graph is generated once and shared through /dev/shm, and rank 0 prints one
    JSON line.  GRX_BENCH_SHARE_GPU=1: both ranks on cuda:0 over gloo (functional check; numbers are meaningless)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_PORT', 'MASTER_ADDR')}
    env['GRX_BENCH_SHARE_GPU'] = '1'
    proc = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--workload', 'ba100k', '--steps', '2',
                           '--warmup', '1', '--no-sharded-extra'], capture_output=True, text=True, timeout=900, env=env, cwd=str(tmp_path))
    assert proc.returncode == 0, proc.stdout[-2000:] + proc.stderr[-3000:]
    lines = [l for l in proc.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, proc.stdout[-2000:]
    line = json.loads(lines[0])
    assert line['n_gpus'] == 2 and line['steps'] == 2 and line['config']['workload'] == 'ba100k'
    assert len(line['per_rank']) == 2 and {pr['rank'] for pr in line['per_rank']} == {0, 1}
    assert line['value'] > 0 and line['roofline']['frac'] > 0


def _worker_rccl(rank, port, kind, driver, out_dir):
    """One rank, backend 'nccl' (= RCCL), GRX_FORCE_COLLECTIVES=1: every exchange of the N > 1 path
    runs as a real RCCL call on HBM tensors (all_to_all_single on fp64 / uint8, all_gather_into_tensor,
    all_reduce SUM / MAX on fp64 / int32) -- what a gloo test cannot cover."""
    import torch
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    os.environ['GRX_FORCE_COLLECTIVES'] = '1'
    torch.cuda.set_device(0)
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
    try:
        from graphrole_amd import RecursiveFeatureExtractor, kernels as K
        from graphrole_amd.roles import factor
        G = _graph(kind)
        fe = RecursiveFeatureExtractor(G, max_generations=4, distributed=True, aggs=['sum', 'mean', 'max'],
                                      native_loop=driver == 'native')
        X = fe.extract_features()
        plan = fe._shard()
        assert plan is not None and plan.world == 1 and not plan._solo
        plan.timing = True                                    # every exchange of one more pass, by kind
        fe.reset()
        fe.run_on_device()
        kinds = plan.collect_timing()
        plan.timing = False
        assert kinds.get('columns_to_owners', [0])[0] >= 1 and kinds.get('owned_to_rows', [0])[0] >= 1, kinds
        assert kinds.get('all_gather_rows', [0])[0] >= 1 and kinds.get('all_reduce', [0])[0] >= 1, kinds
        Xd = K.gather_columns(fe.device_features()[1], G.n)
        F = X.shape[1]
        omega = np.random.RandomState(5).normal(size=(F, 4 + 10))
        state, n_iter = factor.nmf_device(Xd, G.n, 4, omega, plan=plan)
        os.environ['GRX_FORCE_COLLECTIVES'] = '0'
        fe1 = RecursiveFeatureExtractor(G, max_generations=4, aggs=['sum', 'mean', 'max'])
        X1 = fe1.extract_features()
        Xd1 = K.gather_columns(fe1.device_features()[1], G.n)
        s1, it1 = factor.nmf_device(Xd1, G.n, 4, omega)
        np.savez(os.path.join(out_dir, 'rccl.npz'), X=X.values.astype(float), cols=np.array(list(X.columns)),
                 X1=X1.values.astype(float), cols1=np.array(list(X1.columns)), n_iter=n_iter, it1=it1,
                 H=K.to_host(state.H), H1=K.to_host(s1.H), W=K.to_host(state.W)[:, :G.n], W1=K.to_host(s1.W)[:, :G.n])
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('kind,driver', [('ba', 'native'), ('ba', 'per_kernel'), ('directed_weighted', 'native'),
                                         ('ba1m', 'native')])
def test_rccl_collectives_one_rank(kind, driver, tmp_path):
    mp.spawn(_worker_rccl, args=(_free_port(), kind, driver, str(tmp_path)), nprocs=1, join=True)
    r = np.load(tmp_path / 'rccl.npz')
    assert list(r['cols']) == list(r['cols1'])
    assert np.array_equal(r['X'], r['X1'])
    assert int(r['n_iter']) == int(r['it1'])
    np.testing.assert_allclose(r['H'], r['H1'], rtol=1e-9)
    np.testing.assert_allclose(r['W'], r['W1'], rtol=1e-9, atol=1e-12 * np.abs(r['W1']).max())


def _roles_worker(rank, port, out_dir):
    import pandas as pd
    import torch
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    torch.cuda.set_device(0)
    dist.init_process_group('gloo', rank=rank, world_size=WORLD)
    try:
        from graphrole_amd import RecursiveFeatureExtractor, RoleExtractor
        G = _graph('ba')
        X = RecursiveFeatureExtractor(G, max_generations=3).extract_features()
        out = {}
        for tag, kwargs in (('grid', dict(n_role_range=(2, 4), n_bit_range=(2, 5))), ('fixed', dict(n_roles=4))):
            for mode, distributed in (('sharded', True), ('single', None)):
                if mode == 'single' and rank != 0:
                    continue
                np.random.seed(11)
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


def test_sharded_role_extraction_two_ranks_one_gpu(tmp_path):
    """RoleExtractor(distributed=True) on the real kernels: row-sharded factorisations and KL costs of the MDL grid
    (grx_gram / grx_project / grx_nmf_w_pass / grx_nmf_kl_cost with row ranges + all-reduces), replicated KMeans
    encode; both ranks end with the single-process factors and cost grid."""
    mp.spawn(_roles_worker, args=(_free_port(), str(tmp_path)), nprocs=WORLD, join=True)
    r0, r1 = np.load(tmp_path / 'roles0.npz'), np.load(tmp_path / 'roles1.npz')
    assert list(r0['grid_sharded_sel']) == list(r1['grid_sharded_sel']) == list(r0['grid_single_sel'])
    np.testing.assert_allclose(r0['grid_sharded_err'], r0['grid_single_err'], rtol=1e-8, equal_nan=True)
    for tag in ('grid', 'fixed'):
        for part in ('G', 'F'):
            a, b, s = r0[f'{tag}_sharded_{part}'], r1[f'{tag}_sharded_{part}'], r0[f'{tag}_single_{part}']
            assert np.array_equal(a, b), 'ranks disagree'
            np.testing.assert_allclose(a, s, rtol=1e-8, atol=1e-11 * np.abs(s).max())
