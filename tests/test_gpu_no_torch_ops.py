"""
-m gpu: torch is plumbing (allocations, streams, torch.distributed initialisation) -- no torch COMPUTE kernel may run
on the product path, single-GPU or sharded.  The path is driven under `rocprofv3 --kernel-trace`, delimited by
grx_marker_kernel launches (grx_trace_marker), and every kernel between the markers must be one of this library's,
an RCCL kernel or a runtime copy / fill.
"""
import csv
import glob
import os
import re
import subprocess
import sys
import textwrap

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

DRIVER = '''
import ctypes, os, sys
sys.path.insert(0, ROOT)
os.environ['MASTER_ADDR'] = '127.0.0.1'
os.environ['MASTER_PORT'] = str(PORT)
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
os.environ['GRX_FORCE_COLLECTIVES'] = '1'
import numpy as np
import torch
import torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
from graphrole_amd import RecursiveFeatureExtractor, RoleExtractor, _lib, kernels as K, synth
from graphrole_amd.roles import factor
lib = _lib.load()
stream = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
G = synth.ba_graph(50_000, 8, seed=2)
Gd = synth.directed_weighted_graph(20_000, 200_000, seed=3)
# distributed initialisation (allowed to use torch): the communicator of the sharded extractor
warm = RecursiveFeatureExtractor(G, max_generations=3, distributed=True)
warm._shard().comm()
torch.cuda.synchronize()
lib.grx_trace_marker(1, stream())
for graph, kwargs in ((G, {}), (Gd, dict(attributes=True)), (G, dict(aggs=['sum', 'mean', 'max', 'std'])),
                      (G, dict(native_loop=False))):
    for distributed in (None, True):
        fe = RecursiveFeatureExtractor(graph, max_generations=4, distributed=distributed, **kwargs)
        X = fe.extract_features()
        names, cols = fe.device_features()
        Xd = K.gather_columns(cols, graph.n)
        omega = np.random.RandomState(1).normal(size=(len(names), 14))
        factor.nmf_device(Xd, graph.n, 4, omega, plan=fe._shard())
np.random.seed(0)
rx = RoleExtractor(n_roles=4)
rx.extract_role_factors(X)
rx = RoleExtractor(n_role_range=(2, 3), n_bit_range=(3, 4), distributed=True)
rx.extract_role_factors(X)
torch.cuda.synchronize()
lib.grx_trace_marker(2, stream())
torch.cuda.synchronize()
dist.destroy_process_group()
print('DRIVER_OK')
'''

TORCH_KERNEL = re.compile(r'at::|c10::|elementwise_kernel|vectorized_elementwise|CatArray|index_select|indexSelect|'
                          r'FillFunctor|reduce_kernel|gatherTopK|at_cuda_detail|cunn_|triton')


def _free_port():
    import socket
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def test_no_torch_compute_kernel_between_the_markers(tmp_path):
    script = tmp_path / 'drive.py'
    script.write_text('ROOT = %r\nPORT = %d\n' % (ROOT, _free_port()) + textwrap.dedent(DRIVER))
    out = tmp_path / 'trace'
    env = dict(os.environ, TMPDIR='/tmp')
    cmd = ['rocprofv3', '--kernel-trace', '--output-format', 'csv', '-d', str(out), '-o', 't', '--',
           sys.executable, str(script)]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd='/tmp')
    assert res.returncode == 0 and 'DRIVER_OK' in res.stdout, res.stdout[-3000:] + res.stderr[-3000:]
    files = glob.glob(str(out / '**' / '*kernel_trace.csv'), recursive=True)
    assert files, 'rocprofv3 wrote no kernel trace'
    rows = []
    for path in files:
        with open(path, newline='') as fh:
            rows += list(csv.DictReader(fh))
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    names = [r['Kernel_Name'] for r in rows]
    marks = [i for i, nm in enumerate(names) if 'grx_marker_kernel' in nm]
    assert len(marks) == 2, f'expected two marker launches, found {len(marks)}'
    inside = names[marks[0] + 1:marks[1]]
    assert len(inside) > 500, 'the product path launched suspiciously few kernels'
    bad = sorted({nm for nm in inside if TORCH_KERNEL.search(nm)})
    assert not bad, f'torch compute kernels on the product path: {bad[:10]}'
    # and the exchanges really ran as RCCL kernels (one-rank group with GRX_FORCE_COLLECTIVES=1)
    assert any(re.search(r'nccl|rccl', nm, re.I) for nm in inside), 'no RCCL kernel between the markers'
