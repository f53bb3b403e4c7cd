"""
-m "not gpu": the host-side pieces of bench.py that a multi-GPU driver run depends on and that need no device --
the self-launch under torch.distributed.run, the one-generation-per-node graph hand-over through /dev/shm, the
gather ceiling looked up from the committed microbenchmark profile.
"""
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _clean_env():
    return {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_PORT', 'MASTER_ADDR')}


def test_gpus_2_without_launcher_re_executes_under_torchrun(tmp_path):
    """no GPU here: every rank must get as far as bench.py's own 'needs a GPU' exit -- which proves the script became
    its launcher (two ranks, WORLD_SIZE = 2) instead of dying on a WORLD_SIZE assertion"""
    proc = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--workload', 'tiny', '--steps', '1',
                           '--warmup', '0'], capture_output=True, text=True, timeout=600, env=_clean_env(), cwd=str(tmp_path))
    text = proc.stdout + proc.stderr
    assert proc.returncode != 0
    assert 'bench.py needs a GPU' in text, text[-3000:]
    assert 'AssertionError' not in text and 'WORLD_SIZE=' not in text, text[-3000:]


def test_world_size_mismatch_is_a_clear_exit(tmp_path):
    env = dict(_clean_env(), WORLD_SIZE='3', RANK='0', LOCAL_RANK='0')
    proc = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--workload', 'tiny'],
                          capture_output=True, text=True, timeout=600, env=env, cwd=str(tmp_path))
    assert proc.returncode != 0
    assert 'needs a GPU' in proc.stderr or '--gpus 2 but the launcher set WORLD_SIZE=3' in proc.stderr


def test_graph_is_generated_once_and_mapped_by_the_other_ranks(tmp_path, monkeypatch):
    sys.path.insert(0, ROOT)
    import bench
    calls = []
    real = bench.generate_graph
    monkeypatch.setattr(bench, 'generate_graph', lambda name: (calls.append(name), real(name))[1])
    monkeypatch.setattr(bench, 'shared_dirs', lambda name: [str(tmp_path / 'shared' / name)])
    g0 = bench.build_graph('tiny', world=2, local_rank=0)
    g1 = bench.build_graph('tiny', world=2, local_rank=1)          # finds the published files, generates nothing
    assert calls == ['tiny']
    ref = real('tiny')
    for g in (g0, g1):
        assert g.n == ref.n and g.num_edges == ref.num_edges
        assert np.array_equal(g.row_ptr, ref.row_ptr) and np.array_equal(g.col, ref.col) and np.array_equal(g.adj_col, ref.adj_col)
    src = g1.edge_arrays()[0]
    assert not src.flags.owndata                                    # a view of the mapped file, not a private copy
    assert bench._PUBLISHED == [str(tmp_path / 'shared' / 'tiny')]
    bench._PUBLISHED.clear()


def test_a_failing_builder_does_not_leave_the_other_ranks_waiting(tmp_path, monkeypatch):
    sys.path.insert(0, ROOT)
    import pytest
    import bench
    monkeypatch.setattr(bench, 'shared_dirs', lambda name: [str(tmp_path / 'shared' / name)])

    def boom(name):
        raise MemoryError('no graph today')
    monkeypatch.setattr(bench, 'generate_graph', boom)
    with pytest.raises(MemoryError):
        bench.build_graph('tiny', world=2, local_rank=0)
    with pytest.raises(SystemExit, match='failed'):
        bench.build_graph('tiny', world=2, local_rank=1)        # returns at once, not after an hour


def test_the_builder_skips_a_location_without_room(tmp_path, monkeypatch):
    sys.path.insert(0, ROOT)
    import collections
    import shutil
    import bench
    small, roomy = tmp_path / 'small' / 'tiny', tmp_path / 'roomy' / 'tiny'
    monkeypatch.setattr(bench, 'shared_dirs', lambda name: [str(small), str(roomy)])
    usage = collections.namedtuple('usage', 'total used free')
    real = shutil.disk_usage
    monkeypatch.setattr(shutil, 'disk_usage', lambda p: usage(1 << 20, 1 << 20, 0) if 'small' in str(p) else real(p))
    g = bench.build_graph('tiny', world=2, local_rank=0)
    assert not (small / 'meta.json').exists() and (roomy / 'meta.json').exists()
    assert bench.build_graph('tiny', world=2, local_rank=1).n == g.n
    bench._PUBLISHED.clear()


def test_weighted_graph_round_trips_through_the_shared_directory(tmp_path):
    sys.path.insert(0, ROOT)
    from graphrole_amd import synth
    g = synth.directed_weighted_graph(3000, 20000, seed=1)
    synth.save_graph(g, str(tmp_path))
    h = synth.load_graph(str(tmp_path))
    assert (h.n, h.directed, h.weighted, h.integral, h.num_edges) == (g.n, g.directed, g.weighted, g.integral, g.num_edges)
    assert np.array_equal(h.row_ptr, g.row_ptr) and np.array_equal(h.col, g.col) and np.array_equal(h.w, g.w)
    assert np.array_equal(h.t_row_ptr, g.t_row_ptr) and np.array_equal(h.t_w, g.t_w)
    assert list(h.attributes) == list(g.attributes)
    assert all(np.array_equal(h.attributes[k], g.attributes[k]) for k in g.attributes)


def test_gather_ceiling_comes_from_the_committed_microbenchmark():
    sys.path.insert(0, ROOT)
    import bench
    cells = [json.loads(l) for l in open(bench.GATHER_PROFILE) if l.strip()]
    best = {(c['table_mb'], c['row_bytes'], c['dist']): c['rows_per_s'] for c in cells if c['kind'] == 'best'}
    assert {mb for mb, _, _ in best} >= {2.0, 17.0, 64.0, 1024.0} and {rb for _, rb, _ in best} == {16, 32, 64}
    # at a measured size the ceiling is the best cell of that size; between sizes it lies between the neighbours
    at64 = max(best[(64.0, rb, 'powerlaw')] for rb in (16, 32, 64))
    assert abs(bench.gather_ceiling(64e6, 'powerlaw') - at64) <= 1e-9 * at64
    mid = bench.gather_ceiling(40e6, 'uniform')
    lo = max(best[(64.0, rb, 'uniform')] for rb in (16, 32, 64))
    hi = max(best[(32.0, rb, 'uniform')] for rb in (16, 32, 64))
    assert lo < mid < hi
    assert bench.gather_ceiling(1e3, 'uniform') == max(best[(2.0, rb, 'uniform')] for rb in (16, 32, 64))
    # the LDS-prefix experiment is part of the same file (kept as evidence)
    assert any(c['kind'] == 'lds_prefix' for c in cells)
