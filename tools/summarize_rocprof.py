#!/usr/bin/env python3
"""
Condense rocprofv3 output collected by tools/profile_gpu.sh (gpurun_out/prof_<tag>/) into
profiles/<tag>_kernel_stats.csv (per-kernel count / total / average / share) and
profiles/<tag>_pmc.json (per-kernel averages of the PMC counters, with the gfx950 FETCH_SIZE
correction of MI355X_MICROARCH.md: FETCH_SIZE counts 64 B per 128 B request -> doubled).
"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

tag = sys.argv[1] if len(sys.argv) > 1 else 'r02'
src = os.path.join('gpurun_out', f'prof_{tag}')
os.makedirs('profiles', exist_ok=True)


def short(name):
    name = re.sub(r'\(anonymous namespace\)::', '', name)
    name = re.sub(r'^void ', '', name)
    return re.sub(r'\(.*$', '', name)


# ---- kernel trace -> stats
rows = defaultdict(lambda: [0, 0.0])
for path in glob.glob(os.path.join(src, 'trace', '**', '*kernel_trace.csv'), recursive=True):
    with open(path) as fh:
        for r in csv.DictReader(fh):
            dur = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3      # us
            k = short(r['Kernel_Name'])
            rows[k][0] += 1
            rows[k][1] += dur
total = sum(v[1] for v in rows.values()) or 1.0
with open(os.path.join('profiles', f'{tag}_kernel_stats.csv'), 'w') as fh:
    fh.write('kernel,calls,total_us,avg_us,percent\n')
    for k, (c, t) in sorted(rows.items(), key=lambda kv: -kv[1][1]):
        fh.write(f'"{k}",{c},{t:.1f},{t / c:.2f},{100 * t / total:.2f}\n')
print(f'{len(rows)} kernels, total {total / 1e3:.2f} ms')

# ---- pmc passes
pmc = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
for path in glob.glob(os.path.join(src, 'pmc_*', '**', '*counter_collection.csv'), recursive=True):
    with open(path) as fh:
        for r in csv.DictReader(fh):
            k = short(r['Kernel_Name'])
            cell = pmc[k][r['Counter_Name']]
            cell[0] += 1
            cell[1] += float(r['Counter_Value'])
out = {}
for k, ctrs in pmc.items():
    d = {c: v[1] / v[0] for c, v in ctrs.items()}
    d['launches_sampled'] = max(v[0] for v in ctrs.values())
    if 'FETCH_SIZE' in d:
        # rocprofv3 reports FETCH_SIZE/WRITE_SIZE in KiB; gfx950 FETCH_SIZE = 1/2 of the bytes
        d['hbm_read_bytes_per_launch_corrected'] = d['FETCH_SIZE'] * 1024 * 2
    if 'WRITE_SIZE' in d:
        d['hbm_write_bytes_per_launch'] = d['WRITE_SIZE'] * 1024
    if 'TCC_HIT_sum' in d and 'TCC_MISS_sum' in d and d['TCC_HIT_sum'] + d['TCC_MISS_sum'] > 0:
        d['l2_hit_rate'] = d['TCC_HIT_sum'] / (d['TCC_HIT_sum'] + d['TCC_MISS_sum'])
    if d.get('SQ_VALU_MFMA_BUSY_CYCLES') and d.get('SQ_BUSY_CU_CYCLES'):
        # matrix-core utilisation = cycles a SIMD's MFMA pipe was busy / (cycles a CU was busy x 4 SIMDs);
        # both SQ counters are sums over all CUs of the chip
        d['mfma_util'] = d['SQ_VALU_MFMA_BUSY_CYCLES'] / (4.0 * d['SQ_BUSY_CU_CYCLES'])
    if d.get('SQ_VALU_MFMA_BUSY_CYCLES') and d.get('GRBM_GUI_ACTIVE'):
        # the same against the whole chip for the whole launch: 256 CUs x 4 SIMDs x GPU-active cycles
        # (GRBM_GUI_ACTIVE is reported once per XCD and averaged over them by the accumulation above
        # when rocprofv3 emits one row per XCD; a single summed row is 8 x the cycle count)
        d['mfma_util_chip'] = d['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024.0 * d['GRBM_GUI_ACTIVE'])
    out[k] = d
json.dump(out, open(os.path.join('profiles', f'{tag}_pmc.json'), 'w'), indent=1, sort_keys=True)
print(f'pmc kernels: {len(out)}')

# ---- traffic figure bench.py attaches to its roofline object (aggregation kernel).
# Calibration for THIS access pattern (random 32-64 B row gathers): FETCH_SIZE(KiB)*1024 equals
# TCC_MISS_sum * 64 B within 5 %, i.e. the requests are 64-byte and the 2x correction the guide gives
# for wide coalesced streams does not apply; WRITE_SIZE is taken as reported.
agg = {k: d for k, d in out.items() if k.startswith(('aggregate_kernel', 'aggregate_i32_kernel', 'aggregate_packed_kernel'))}
if agg:
    w = sum(d['launches_sampled'] for d in agg.values())
    fetch = sum(d.get('FETCH_SIZE', 0.0) * 1024 * d['launches_sampled'] for d in agg.values()) / w
    write = sum(d.get('WRITE_SIZE', 0.0) * 1024 * d['launches_sampled'] for d in agg.values()) / w
    miss = sum(d.get('TCC_MISS_sum', 0.0) * d['launches_sampled'] for d in agg.values()) / w
    workload = sys.argv[2] if len(sys.argv) > 2 else 'ba1m'
    nmf = {k: d for k, d in out.items() if k.startswith('nmf_w_pass')}
    nmf_traffic = None
    if nmf:
        wn = sum(d['launches_sampled'] for d in nmf.values())
        nmf_fetch = sum(d.get('FETCH_SIZE', 0.0) * 1024 * d['launches_sampled'] for d in nmf.values()) / wn
        nmf_write = sum(d.get('WRITE_SIZE', 0.0) * 1024 * d['launches_sampled'] for d in nmf.values()) / wn
        # coalesced streaming reads: FETCH_SIZE reports half the bytes on gfx950 (MI355X_MICROARCH.md, HBM
        # section); this kernel confirms it on a known byte count (208 MB read, 48 MB written per launch)
        nmf_traffic = 2.0 * nmf_fetch + nmf_write
        nmf_first = max(nmf.values(), key=lambda d: d['launches_sampled'])
    binary = {}
    if os.path.exists(os.path.join(src, 'binary.json')):
        binary = json.load(open(os.path.join(src, 'binary.json')))
    tpath = os.path.join('profiles', 'traffic_latest.json')
    table = {}
    if os.path.exists(tpath):
        table = json.load(open(tpath))
        if 'workload' in table:                     # the one-workload layout of rounds 1-2
            table = {table['workload']: table}
    table[workload] = ({'workload': workload, 'n_gpus': 1, 'source': f'profiles/{tag}_pmc.json',
               # sha256 of the library and of bench.py the counters were collected with: bench.py prints
               # "traffic_stale": true and nulls `traffic` when the binary it runs differs
               'lib_sha256': binary.get('lib_sha256'), 'bench_sha256': binary.get('bench_sha256'),
               'aggregate_kernel_hbm_bytes_per_launch': fetch + write,
               'nmf_w_pass_hbm_bytes_per_launch': nmf_traffic,
               'nmf_w_pass_fetch_reported': nmf_fetch if nmf else None,
               'nmf_w_pass_write_reported': nmf_write if nmf else None,
               'nmf_note': '2 x FETCH_SIZE + WRITE_SIZE: the gfx950 correction for coalesced streaming reads',
               'nmf_w_pass_mfma': ({k: nmf_first.get(k) for k in ('SQ_VALU_MFMA_BUSY_CYCLES', 'SQ_INSTS_VALU_MFMA_MOPS_F64',
                                                               'SQ_INSTS_VALU_MFMA_F64', 'SQ_BUSY_CU_CYCLES', 'GRBM_GUI_ACTIVE',
                                                               'mfma_util', 'mfma_util_chip')} if nmf else None),
               'aggregate_l2_hit_rate': (sum(d.get('l2_hit_rate', 0.0) * d['launches_sampled'] for d in agg.values()) / w),
               'fetch_bytes': fetch, 'write_bytes': write, 'tcc_miss_x64B': miss * 64,
               'note': 'fabric-side bytes (L2 misses; Infinity-Cache hits are counted), FETCH_SIZE uncorrected, see script'})
    json.dump(table, open(tpath, 'w'), indent=1)
    print('traffic', fetch + write)
