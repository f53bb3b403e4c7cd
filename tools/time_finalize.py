#!/usr/bin/env python3
"""Where the wall-clock of extract_features() goes on a large table (run on the GPU box): tools/time_finalize.py [workload]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from graphrole_amd import RecursiveFeatureExtractor, kernels as K
from graphrole_amd.features import handoff

name = sys.argv[1] if len(sys.argv) > 1 else 'dw5m'
G = bench.build_graph(name)
fe = RecursiveFeatureExtractor(G, max_generations=4, attributes=bool(G.attributes))
t0 = time.perf_counter(); fe.run_on_device(); torch.cuda.synchronize(); t1 = time.perf_counter()
print(f'run_on_device (incl. ingest) {t1 - t0:.3f} s')
names, cols = fe.device_features()
n = G.n
for trial in range(2):
    t0 = time.perf_counter()
    dev_block = K.permute_columns(list(cols), fe._inv_device(), n); torch.cuda.synchronize()
    t1 = time.perf_counter()
    block = K.to_host(dev_block)
    t2 = time.perf_counter()
    import pandas as pd
    frame = pd.DataFrame(block.T, columns=names, copy=False)
    t3 = time.perf_counter()
    handoff.register(K, frame, dev_block)
    t4 = time.perf_counter()
    hit = handoff.lookup(K, frame)
    t5 = time.perf_counter()
    gb = block.nbytes / 1e9
    print(f'trial {trial}: permute {t1 - t0:.3f}  download {t2 - t1:.3f} ({gb / (t2 - t1):.1f} GB/s)  frame {t3 - t2:.3f}  '
          f'register {t4 - t3:.3f} ({gb / (t4 - t3):.1f} GB/s)  lookup {t5 - t4:.3f}  hit={hit is not None}')
    # the same copy into an already touched buffer (page faults out of the way)
    t0 = time.perf_counter()
    from graphrole_amd import _lib
    _lib.call('grx_download', K._hptr(block), K._ptr(dev_block), block.nbytes, K._stream())
    t1 = time.perf_counter()
    print(f'         download into touched memory {t1 - t0:.3f} ({gb / (t1 - t0):.1f} GB/s)')
    del frame, block
t0 = time.perf_counter(); X = fe.extract_features(); t1 = time.perf_counter()
print(f'extract_features() after the run: {t1 - t0:.3f} s')
