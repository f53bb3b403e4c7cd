"""Microbenchmark of grx_triangle_counts on the BASELINE BA graph (tools/pmc_triangles.sh drives it under rocprofv3)."""
import sys, time
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from graphrole_amd import synth, kernels as K

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
G = synth.ba_graph(n, 10, seed=0)
csr = K.DeviceCSR(G.row_ptr, G.col, None, agg_col=G.adj_col)
csr.oriented()
for _ in range(3):
    K.triangle_counts(csr)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    T = K.triangle_counts(csr)
torch.cuda.synchronize()
print('ms per launch (incl. zero fill)', (time.perf_counter() - t0) / reps * 1e3, 'triangles', int(T.sum().item()) // 3)
