#!/usr/bin/env python3
"""
tools/project_scaling.py -- a MEASURED PROJECTION of the 2 / 4 / 8-GPU step, made on ONE GPU (run through gpurun).

No multi-GPU box has ever been available to this repository (SCALE_rNN.json: skipped), so the N > 1 numbers north_star
asks for cannot be measured.  What can be measured today is what decides them: how evenly the node-range partition
(graphrole_amd/parallel.py ShardPlan: contiguous ranges of the degree-descending row order, cut by nnz + n) spreads
every phase of the step.  For P in {1, 2, 4, 8} and every rank's row range this script runs THAT RANK'S SHARE of every
phase ALONE on the one GPU -- the very kernels the sharded path launches, with its (row_begin, row_end) -- and times it:

  gen0        ego-net / triangle kernels over the rank's rows        (graph/interface/networkx.py:71-83)
  aggregate   neighbour aggregation of generation g over the rows    (features/extract.py:98-119)
  binning     vertical_log_binning of the columns the rank OWNS (column c of a generation's candidates belongs to rank
              c mod P; every owner bins whole columns)                (features/prune.py:13-56)
  chebyshev   pairwise bin distances over the rank's rows            (features/prune.py:108)
  nmf         the W pass + residual of the MU loop over the rows, 20 iterations as the bench step runs
              (roles/factor.py:10-26)

and adds the exchange model: calls x per-call latency + bytes / (links x 153 GB/s), the call counts and the one-rank
latency from the forced-collectives bench line (profiles/r05_bench_forced_collectives_*.json), the bytes from the
protocol of parallel.py (whole fresh columns to their owners, bins back, retained columns all-gathered).

Output (stdout, one JSON document; profiles/r06_projected_scaling_<workload>.json): per P the per-rank milliseconds of
every phase, max / mean imbalance per phase, the projected step = sum over phases of the slowest rank + exchange +
the replicated remainder, projected edges/s.  EVERYTHING HERE IS A PROJECTION: nothing ran over xGMI.

    python tools/project_scaling.py [ba1m|dw5m|er100k] [--cuts work|...]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from graphrole_amd import RecursiveFeatureExtractor, synth  # noqa: E402
from graphrole_amd import kernels as K  # noqa: E402
from graphrole_amd.roles import factor  # noqa: E402

MAX_GENERATIONS = 4
N_ROLES = 6
LINK_GBS = 153.0            # one xGMI link, one direction (MI355X_MICROARCH.md)
LINKS = 7


def build(workload):
    if workload == 'ba1m':
        return synth.ba_graph(1_000_000, 10, seed=0)
    if workload == 'er100k':
        return synth.er_graph(100_000, 1_000_000, seed=0)
    if workload == 'dw5m':
        return synth.directed_weighted_graph(5_000_000, 100_000_000, seed=0)
    if workload == 'dw1m':
        return synth.directed_weighted_graph(1_000_000, 20_000_000, seed=0)
    raise SystemExit(f'unknown workload {workload}')


def cuts_work(row_ptr, world):
    """the partition of every ShardPlan (graphrole_amd/parallel.py row_cuts): balanced by nnz + n"""
    from graphrole_amd.parallel import row_cuts
    return row_cuts(np.asarray(row_ptr), world)


def timed(fn, reps=5, inner=5):
    """device milliseconds of one fn(), the fastest of `reps` batches of `inner` calls back to back between two
    synchronisations (the launch and synchronisation overhead of a single call would otherwise be charged to kernels
    of a few microseconds; the minimum discards batches another tenant of the box disturbed)"""
    fn()
    torch.cuda.synchronize()
    out = []
    for _ in range(reps):
        t0 = time.perf_counter()
        for _ in range(inner):
            fn()
        torch.cuda.synchronize()
        out.append((time.perf_counter() - t0) * 1e3 / inner)
    return float(np.min(out))


def cuts_from_work(work_per_row, world):
    """contiguous ranges with equal shares of sum(work_per_row)"""
    n = len(work_per_row)
    cum = np.cumsum(np.asarray(work_per_row, dtype=np.float64))
    total = float(cum[-1]) if n else 0.0
    cuts = [0] + [int(np.searchsorted(cum, total * p / world, side='left')) for p in range(1, world)] + [n]
    return np.maximum.accumulate(np.array(cuts, dtype=np.int64))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('workload', nargs='?', default='ba1m')
    ap.add_argument('--call-us', type=float, default=None, help='per-collective latency (default: from the forced-collectives line)')
    args = ap.parse_args()
    G = build(args.workload)
    fe = RecursiveFeatureExtractor(G, max_generations=MAX_GENERATIONS, attributes=bool(getattr(G, 'attributes', None)))
    fe.run_on_device()
    torch.cuda.synchronize()
    host, dev, _ = fe.graph._device_graph()
    n = host.n
    names, cols = fe.device_features()
    col = dict(zip(names, cols))
    gens = {g: list(v) for g, v in fe._final_names.items()}
    executed = max(gens)
    # the one-GPU step, phase by phase (the bench's own figures)
    t_full = timed(lambda: (fe.reset(), fe.run_on_device()), reps=3)
    Xd = K.gather_columns(cols, n)
    rng = np.random.default_rng(0)
    omega = rng.normal(size=(len(names), N_ROLES + 10))
    t_nmf_full = timed(lambda: factor.nmf_device(Xd, n, N_ROLES, omega), reps=3)
    state, n_iter = factor.nmf_device(Xd, n, N_ROLES, omega)

    # representative inputs of every generation: the columns it aggregates, the candidate block it bins
    prev_cols = {g: [col[nm] for nm in gens[g - 1]] for g in range(1, executed + 1)}
    packed = {g: K.pack_rows(prev_cols[g], n) for g in prev_cols}
    cand = {g: K.aggregate(dev, packed[g][0], len(prev_cols[g]), packed[g][1]) for g in prev_cols}
    work_counts = [s.get('working') if isinstance(s, dict) else None for s in fe.stats]
    rowsum = K.row_sums(dev, False) if host.weighted else None
    bins_full = {}
    for g in prev_cols:
        b, _ = K.vertical_log_bin(cand[g])
        bins_full[g] = b
    gen0_block = K.gather_columns([col[nm] for nm in gens[0]], n)
    bins_gen0, _ = K.vertical_log_bin(gen0_block)

    out = {'workload': args.workload, 'n': int(n), 'nnz': int(host.nnz), 'generations_executed': int(executed),
           'features': len(names), 'one_gpu_ms': {'refex_step': t_full, 'nmf_fit': t_nmf_full, 'nmf_iterations': int(n_iter)},
           'label': 'PROJECTED from single-GPU measurements of every rank\'s share; nothing here ran over xGMI',
           'partition': 'contiguous ranges of the degree-descending row order, cut by nnz + n (parallel.py ShardPlan)'}
    # exchange model
    # calls per step: this workload's forced-collectives line; time per call: the ba1m line's (messages of a few KB to a
    # few MB inside one rank -- the latency of a call; the larger workloads' one-rank figure is mostly local copying of
    # the payload, which the bandwidth term below stands for)
    forced = os.path.join(ROOT, 'profiles', f'r05_bench_forced_collectives_{args.workload}.json')
    small = os.path.join(ROOT, 'profiles', 'r05_bench_forced_collectives_ba1m.json')
    calls, call_us = 42, 16.6
    if os.path.exists(forced):
        calls = int(json.load(open(forced))['per_rank'][0]['exchange']['calls'])
    if os.path.exists(small):
        ex = json.load(open(small))['per_rank'][0]['exchange']
        call_us = 1e3 * ex['ms'] / ex['calls']
    if args.call_us:
        call_us = args.call_us
    out['exchange_model'] = {'calls_per_step': calls, 'us_per_call': call_us, 'links': LINKS, 'link_GBs': LINK_GBS,
                             'source': 'calls and one-rank per-call time: ' + os.path.relpath(forced, ROOT) +
                                       ' (RCCL with every exchange forced on one rank); bytes: protocol of parallel.py'}
    deg = np.diff(np.asarray(host.row_ptr)).astype(np.float64)
    ego_rules = {'nnz+n (the ShardPlan cut)': deg + 1.0, 'd^2+n': deg * deg + 1.0, 'd^1.5+n': deg ** 1.5 + 1.0}
    if not host.directed and not host.weighted:
        # the sharded path (kernels.egonet_features): triangle counts over the rank's share of the ORIENTED source rows
        # (DeviceCSR.triangle_split: its own cut, balanced by d+ (d+ + 1)), summed over the ranks, then the row pass
        def gen0_fn(rb, re, r=0, world=1):
            T = K.triangle_counts(dev, *dev.triangle_split(r, world))
            K.egonet_from_triangles(dev, T, rb, re)
    else:
        def gen0_fn(rb, re, r=0, world=1):
            K.egonet_features_general(dev, host.directed, rowsum, rb, re)
    per_p = {}
    for world in (1, 2, 4, 8):
        bounds = cuts_work(host.row_ptr, world)
        nmf_bounds = cuts_from_work(np.ones(n), world)          # the NMF passes cost the same for every row: equal rows
        ranks = []
        for r in range(world):
            rb, re = int(bounds[r]), int(bounds[r + 1])
            ph = {'rows': re - rb, 'nnz': int(host.row_ptr[re] - host.row_ptr[rb])}
            ph['gen0'] = timed(lambda: gen0_fn(rb, re, r, world))
            per_gen = {}
            # generation 0's own binning + distances
            mine0 = list(range(r, len(gens[0]), world))
            b0 = 0.0
            if mine0:
                sub0 = gen0_block[mine0].contiguous()
                b0 = timed(lambda: K.vertical_log_bin(sub0))
            c0 = timed(lambda: K.chebyshev([bins_gen0[j] for j in range(len(gens[0]))], n, 0, rb, re, cap=0))
            per_gen['0'] = {'aggregate': 0.0, 'binning': b0, 'chebyshev': c0}
            seen_bins = [bins_gen0[j] for j in range(len(gens[0]))]
            for g in sorted(prev_cols):
                f = len(prev_cols[g])
                a_ms = timed(lambda: K.aggregate(dev, packed[g][0], f, packed[g][1], rb, re))
                mine = list(range(r, 2 * f, world))
                b_ms = 0.0
                if mine:
                    subg = cand[g][mine].contiguous()
                    b_ms = timed(lambda: K.vertical_log_bin(subg))
                work = seen_bins + [bins_full[g][j] for j in range(2 * f)]
                c_ms = timed(lambda: K.chebyshev(work, n, len(seen_bins), rb, re, cap=g))
                per_gen[str(g)] = {'aggregate': a_ms, 'binning': b_ms, 'chebyshev': c_ms, 'columns_owned': len(mine)}
                # the next generation compares against what this one retained
                keep = [j for j, nm in enumerate([f'{c}({a})' for a in ('sum', 'mean') for c in gens[g - 1]]) if nm in gens.get(g, [])]
                seen_bins = seen_bins + [bins_full[g][j] for j in keep]
            ph['generations'] = per_gen
            for key in ('aggregate', 'binning', 'chebyshev'):
                ph[key] = sum(v[key] for v in per_gen.values())
            nb, ne = int(nmf_bounds[r]), int(nmf_bounds[r + 1])

            def mu():
                state.w_pass(nb, ne)
                state.h_update()
            t_it = timed(mu)
            t_res = timed(lambda: state.residual_sq(nb, ne))
            ph['nmf'] = n_iter * t_it + max(1, n_iter // 10) * t_res
            ph['nmf_rows'] = ne - nb
            ranks.append(ph)
        phases = ('gen0', 'aggregate', 'binning', 'chebyshev', 'nmf')
        summary = {}
        for p_ in phases:
            v = np.array([rk[p_] for rk in ranks])
            summary[p_] = {'max_ms': float(v.max()), 'mean_ms': float(v.mean()), 'imbalance': float(v.max() / v.mean()) if v.mean() > 0 else 1.0}
        # a step waits for its slowest rank at every exchange: generation 0, then per generation aggregation /
        # binning (the owners) / distances, then the NMF
        sync_sum = max(rk['gen0'] for rk in ranks) + max(rk['nmf'] for rk in ranks)
        for gkey in ranks[0]['generations']:
            for key in ('aggregate', 'binning', 'chebyshev'):
                sync_sum += max(rk['generations'][gkey][key] for rk in ranks)
        # other cut rules for the ego-net / triangle phase alone (its work grows with the square of a row's degree)
        ego_alt = {}
        if world > 1:
            for rule, wv in ego_rules.items():
                bb = cuts_from_work(wv, world)
                tt = [timed(lambda: gen0_fn(int(bb[q]), int(bb[q + 1]), q, world)) for q in range(world)]
                ego_alt[rule] = {'max_ms': max(tt), 'mean_ms': float(np.mean(tt)), 'imbalance': max(tt) / float(np.mean(tt)),
                                 'rows': [int(x) for x in np.diff(bb)]}
        # exchanged bytes per rank and step (fp64 columns to owners + retained columns all-gathered + uint8 bins back)
        if world > 1:
            fresh = len(gens[0]) + sum(2 * len(prev_cols[g]) for g in prev_cols)
            retained = sum(len(gens.get(g, [])) for g in gens)
            own_rows = np.diff(bounds).min()
            recv = ((fresh / world) * (n - own_rows) * 8 + retained * (n - own_rows) * 8 + fresh * own_rows * 1)
            links = min(world - 1, LINKS)
            ex_ms = calls * call_us * 1e-3 + recv / (links * LINK_GBS * 1e9) * 1e3
        else:
            recv, ex_ms = 0, 0.0
        step = sync_sum + ex_ms
        per_p[str(world)] = {'bounds': [int(b) for b in bounds], 'ranks': ranks, 'phases': summary,
                             'egonet_cut_rules': ego_alt, 'sum_of_slowest_ranks_ms': sync_sum,
                             'exchange_ms': ex_ms, 'exchange_bytes_received_per_rank': float(recv),
                             'projected_step_ms': step}
    # the phases timed here are re-runs of single kernels (no launch overlap, separate packing): projected throughput is
    # the ONE-GPU bench step scaled by the ratio of the summed phase maxima
    base = per_p['1']['projected_step_ms']
    one_gpu_step = t_full + t_nmf_full
    for world, rec in per_p.items():
        scale = rec['projected_step_ms'] / base
        rec['projected_bench_step_ms'] = one_gpu_step * scale
        rec['projected_edges_per_s'] = host.nnz * executed / (one_gpu_step * scale * 1e-3)
        rec['projected_speedup'] = 1.0 / scale
    out['per_world'] = per_p
    out['one_gpu_step_ms_measured_here'] = one_gpu_step
    print(json.dumps(out))


if __name__ == '__main__':
    main()
