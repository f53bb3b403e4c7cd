#!/usr/bin/env python3
"""
bench.py -- headline benchmark of the MI355X ReFeX / RolX hot path.

    python bench.py --gpus 1 --steps K --warmup W [--workload ba1m|er100k|ba100k|tiny]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over the synthetic graph that is already resident in HBM:
ReFeX (generation-0 degree/ego-net features + recursive neighbour aggregation with pruning,
max_generations=4) followed by RolX NMF (NNDSVDa + multiplicative updates to convergence,
n_roles=6) on the resulting features, device-to-device.  The timed region is K steps between
barrier + synchronize pairs; the time is the max over ranks.

  value            = edges aggregated per second of the WHOLE step = nnz * executed recursive
                     generations * K / (time of the K timed steps, ReFeX pass + NMF), whole job
  refex.edges_per_s = the same numerator over the time spent in the ReFeX phase only
  nmf.iters_per_s  = multiplicative-update iterations / s over the NMF phase of the same steps
  ms_per_step      = whole step (both phases)
  ms_per_step_without_launch_events = the same K steps timed once more with the per-launch HIP events of the
                     timed region switched off (reported next to ms_per_step, never instead of it)
  roofline_nmf     = the NMF W pass; its launch time comes from the untimed breakdown pass (events
                     around every launch of one step): twenty launches per step with two event records
                     each would make the small workloads host-bound inside the timed region
  roofline         = aggregation kernel: algorithmic bytes (4 B/edge +
                     (8 + 24 f) B/node per launch) / its HIP-event time inside the timed region;
                     `traffic` comes from the committed rocprofv3 --pmc pass of the same command
                     (profiles/traffic_latest.json) and says so in `traffic_source` -- counters cannot
                     be read from inside the process
  cpu_baseline     = the oracle (plain-C port, 1 core) on the same graph, rank 0, N = 1 only
  cpu_reference_path = BASELINE.md baseline (1): the reference's own pandas / networkx / scipy /
                     sklearn call sequences (oracle/reference_path.py) on bounded samples, extrapolated
                     to the whole workload; `speedup_vs_reference_path` compares it with `api_wall_s`
  api_wall_s       = cold wall-clock of the drop-in calls a GraphRole user makes:
                     RecursiveFeatureExtractor(G).extract_features() -> DataFrame and
                     RoleExtractor(6).extract_role_factors(X) (NMF + encode), ingest share broken out

With --gpus N > 1 the same graph is node-range sharded over the ranks (strong scaling): grx_refex_run and
grx_nmf_fit take the run's communicator (csrc/grx_comm.hip: RCCL bound inside libgrx.so) and issue the exchanges
themselves -- per generation the candidate row slices go to the column owners, bins come back, and only the
retained columns are completed on every rank; one all-reduce per NMF iteration.  The line then also carries
`sharded_dw5m`: the same measurement on BASELINE config 5 (5 M nodes / 100 M weighted arcs), the workload where
sharding pays (90 ms per step on one GPU).  `per_rank` lists every rank's aggregation launch time and the device
time of its exchanges by kind.
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (generator, n, m, description)
    'ba1m': ('ba', 1_000_000, 10, 'Barabasi-Albert n=1,000,000 m=10 (~10M edges), seed 0  [BASELINE config 3/4]'),
    'er100k': ('er', 100_000, 1_000_000, 'Erdos-Renyi G(100,000; 1,000,000), seed 0  [BASELINE config 2]'),
    'ba100k': ('ba', 100_000, 10, 'Barabasi-Albert n=100,000 m=10 (reduced; not a headline number)'),
    'ba10m': ('ba', 10_000_000, 10, 'Barabasi-Albert n=10,000,000 m=10 (~100M edges): 10x the BASELINE graph, scale check'),
    'tiny': ('ba', 5_000, 5, 'Barabasi-Albert n=5,000 m=5 (smoke only)'),
    'dw1m': ('dw', 1_000_000, 10_000_000, 'weighted directed power-law, 1 M nodes / 10 M arcs + 8 attributes (config-5 shape, reduced)'),
    'dw5m': ('dw', 5_000_000, 100_000_000, 'weighted directed power-law, 5 M nodes / 100 M arcs + 8 attributes  [BASELINE config 5, on ONE GPU]'),
}
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec
GATHER_PROFILE = os.path.join(ROOT, 'profiles', 'r04_gather_bw.json')   # tools/microbench/gather_bw.hip on MI355X
N_ROLES = 6
MAX_GENERATIONS = 4


_PUBLISHED = []            # shared graph directories this process created (removed when it ends)


def generate_graph(name):
    from graphrole_amd import synth
    kind, n, m, _ = WORKLOADS[name]
    if kind == 'dw':
        return synth.directed_weighted_graph(n, m, seed=0)
    return synth.ba_graph(n, m, seed=0) if kind == 'ba' else synth.er_graph(n, m, seed=0)


def shared_dirs(name):
    """Where the ranks of one node meet, in order of preference (/dev/shm, then /tmp): the generators are seeded, so the
    content of a workload's directory is a function of its name and of graphrole_amd/synth.py (hashed into the path).
    The builder takes the first location with room; the other ranks look in all of them."""
    import hashlib
    from graphrole_amd import synth
    tag = hashlib.sha256(open(synth.__file__, 'rb').read()).hexdigest()[:12]
    bases = [b for b in ('/dev/shm', '/tmp') if os.path.isdir(b) and os.access(b, os.W_OK)]
    return [os.path.join(b, f'grx_bench_{os.getuid()}', f'{name}_{tag}') for b in bases]


def shared_dir(name):
    return shared_dirs(name)[0]


def private_dir(path):
    """The per-user directory in a world-writable place (/dev/shm, /tmp): created 0700, and refused when somebody else
    owns it or others may write to it (the graph files a rank maps come from there)."""
    os.makedirs(path, mode=0o700, exist_ok=True)
    st = os.stat(path)
    if st.st_uid != os.getuid() or (st.st_mode & 0o022):
        raise OSError(f'{path} is not a private directory of uid {os.getuid()} (owner {st.st_uid}, mode {oct(st.st_mode & 0o777)})')
    return path


def build_graph(name, world=1, local_rank=0, share=False):
    """N = 1: generate.  N > 1 (or share=True: the counter passes of --pmc re-run this script): the synthetic graph
    is generated ONCE per node (local rank 0 -> .npy files in /dev/shm, published by one atomic rename), the other
    processes map the same pages -- eight concurrent numpy generations of the 5 M / 100 M graph would cost minutes of
    host time and 8 x the memory before any GPU work.  A builder that fails leaves a marker, so that nobody waits for
    files that will not come."""
    if world == 1 and not share:
        return generate_graph(name)
    import shutil
    from graphrole_amd import synth
    places = shared_dirs(name)
    def found():
        for p in places:
            if os.path.exists(os.path.join(p, 'meta.json')) and os.stat(os.path.dirname(p)).st_uid == os.getuid():
                return p
        return None
    # the failure marker belongs to THIS launch (the launcher's rendezvous port, else the parent process): a marker left
    # behind by an earlier crashed launch cannot make the other ranks give up while rank 0 builds the graph
    launch = os.environ.get('TORCHELASTIC_RUN_ID') or os.environ.get('MASTER_PORT') or str(os.getppid())
    failed = f'{places[-1]}.failed.{launch}'
    if local_rank == 0 and found() is None:
        try:
            G = generate_graph(name)
            src, dst, w = G.edge_arrays()
            need = int(1.25 * (src.nbytes + dst.nbytes + (w.nbytes if w is not None else 0) +
                               sum(np.asarray(v).nbytes for v in G.attributes.values()))) + (1 << 20)
            last_error = None
            for path in places:
                try:
                    private_dir(os.path.dirname(path))
                    if shutil.disk_usage(os.path.dirname(path)).free < need:
                        continue                           # e.g. a container with a 64 MB /dev/shm
                    tmp = f'{path}.tmp{os.getpid()}'
                    os.makedirs(tmp, exist_ok=True)
                    synth.save_graph(G, tmp)
                    try:
                        os.rename(tmp, path)
                        _PUBLISHED.append(path)
                    except OSError:                        # another launch published the same content first
                        shutil.rmtree(tmp, ignore_errors=True)
                    break
                except OSError as exc:
                    last_error = exc
                    shutil.rmtree(f'{path}.tmp{os.getpid()}', ignore_errors=True)
            else:
                raise RuntimeError(f'no room for the shared graph files ({need >> 20} MiB) in {places}: {last_error}')
        except BaseException:
            private_dir(os.path.dirname(failed))
            open(failed, 'w').write('local rank 0 could not publish the graph')
            raise
    deadline = time.time() + 3600
    while found() is None:
        if os.path.exists(failed):
            raise SystemExit(f'bench.py: the rank that builds the {name} graph failed ({failed})')
        if time.time() > deadline:
            raise SystemExit(f'bench.py: rank waited an hour for {places}')
        time.sleep(0.2)
    return synth.load_graph(found())


PMC_PASSES = ('FETCH_SIZE', 'WRITE_SIZE', 'TCC_HIT_sum TCC_MISS_sum',
              'SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_BUSY_CU_CYCLES')


def pmc_counters(workload, timeout_s=240):
    """Same-run hardware counters: this script re-runs itself (two steps of the same workload on the same graph
    files, same libgrx.so) under `rocprofv3 --pmc`, one pass per counter group and nothing but --pmc in the command
    (MI355X_MICROARCH.md, HBM section), and folds the per-launch averages of the aggregation kernels and of the NMF
    W pass into the line.  Returns None when rocprofv3 is missing or a pass fails (the line then falls back to the
    committed profiles/traffic_latest.json and says so)."""
    import csv
    import glob
    import re
    import shutil
    import subprocess
    import tempfile
    from collections import defaultdict
    rocprof = shutil.which('rocprofv3') or ('/opt/rocm/bin/rocprofv3' if os.path.exists('/opt/rocm/bin/rocprofv3') else None)
    if rocprof is None:
        return None
    out_root = tempfile.mkdtemp(prefix='grx_pmc_', dir='/tmp')
    env = dict(os.environ, TMPDIR='/tmp')
    sums = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
    try:
        for i, group in enumerate(PMC_PASSES):
            cmd = [rocprof, '--pmc', *group.split(), '--output-format', 'csv', '-d', os.path.join(out_root, f'p{i}'), '-o', 'pmc',
                   '--', sys.executable, os.path.abspath(__file__), '--workload', workload, '--pmc-child']
            proc = subprocess.run(cmd, cwd='/tmp', env=env, capture_output=True, text=True, timeout=timeout_s)
            if proc.returncode != 0:
                return {'error': f'rocprofv3 --pmc {group}: rc {proc.returncode}: {proc.stderr[-300:]}'}
            for path in glob.glob(os.path.join(out_root, f'p{i}', '**', '*counter_collection.csv'), recursive=True):
                with open(path) as fh:
                    for row in csv.DictReader(fh):
                        k = re.sub(r'\(.*$', '', re.sub(r'^void ', '', re.sub(r'\(anonymous namespace\)::', '', row['Kernel_Name'])))
                        cell = sums[k][row['Counter_Name']]
                        cell[0] += 1
                        cell[1] += float(row['Counter_Value'])
    except Exception as exc:
        return {'error': repr(exc)}
    finally:
        shutil.rmtree(out_root, ignore_errors=True)

    def fold(prefixes):
        tot, launches = defaultdict(float), 0
        for k, ctrs in sums.items():
            if any(k.startswith(p) for p in prefixes):
                launches += max(v[0] for v in ctrs.values())
                for c, v in ctrs.items():
                    tot[c] += v[1]
        return ({c: v / launches for c, v in tot.items()}, launches) if launches else (None, 0)

    agg, agg_n = fold(('aggregate_kernel', 'aggregate_i32_kernel', 'aggregate_packed_kernel'))
    nmf, nmf_n = fold(('nmf_w_pass',))
    if not agg or 'FETCH_SIZE' not in agg:
        return {'error': 'no aggregation launches in the counter output'}
    res = {'passes': list(PMC_PASSES), 'aggregate_launches_sampled': agg_n,
           # random 16..64-byte row gathers: FETCH_SIZE (KiB) x 1024 equals TCC_MISS_sum x 64 B within 5 % on this pattern,
           # i.e. the requests are 64-byte and the x2 correction for wide coalesced streams does not apply here
           'aggregate_fetch_bytes': agg['FETCH_SIZE'] * 1024, 'aggregate_write_bytes': agg.get('WRITE_SIZE', 0.0) * 1024,
           'aggregate_tcc_miss_x64B': agg.get('TCC_MISS_sum', 0.0) * 64}
    res['aggregate_traffic_per_launch'] = res['aggregate_fetch_bytes'] + res['aggregate_write_bytes']
    if agg.get('TCC_HIT_sum', 0) + agg.get('TCC_MISS_sum', 0) > 0:
        res['aggregate_l2_hit_rate'] = agg['TCC_HIT_sum'] / (agg['TCC_HIT_sum'] + agg['TCC_MISS_sum'])
    if nmf and 'FETCH_SIZE' in nmf:
        # wide coalesced streaming reads: FETCH_SIZE reports half the bytes on gfx950 (the guide's correction)
        res['nmf_w_pass_traffic_per_launch'] = 2.0 * nmf['FETCH_SIZE'] * 1024 + nmf.get('WRITE_SIZE', 0.0) * 1024
        res['nmf_w_pass_launches_sampled'] = nmf_n
        if nmf.get('SQ_VALU_MFMA_BUSY_CYCLES') and nmf.get('SQ_BUSY_CU_CYCLES'):
            res['nmf_w_pass_mfma'] = {'SQ_VALU_MFMA_BUSY_CYCLES': nmf['SQ_VALU_MFMA_BUSY_CYCLES'],
                                      'SQ_INSTS_VALU_MFMA_MOPS_F64': nmf.get('SQ_INSTS_VALU_MFMA_MOPS_F64'),
                                      'SQ_BUSY_CU_CYCLES': nmf['SQ_BUSY_CU_CYCLES'],
                                      # cycles a SIMD's matrix pipe was busy / (cycles a CU was busy x 4 SIMDs)
                                      'mfma_util': nmf['SQ_VALU_MFMA_BUSY_CYCLES'] / (4.0 * nmf['SQ_BUSY_CU_CYCLES'])}
    return res


def gather_ceiling(table_bytes, dist):
    """Rows per second the chip gathers from a table of this many bytes with this index distribution
    (profiles/r04_gather_bw.json, measured with tools/microbench/gather_bw.hip: the gather alone -- no summation
    order, no writes; the rate depends on the table's BYTES, not on the row width: a 16-, 32- or 64-byte row is one
    request either way).  Log-linear interpolation between the measured table sizes; None without the file."""
    try:
        cells = [json.loads(l) for l in open(GATHER_PROFILE) if l.strip()]
    except OSError:
        return None
    pts = {}
    for c in cells:
        if c.get('kind') == 'best' and c['dist'] == dist:
            mb = c['table_mb']
            pts[mb] = max(pts.get(mb, 0.0), c['rows_per_s'])
    if not pts:
        return None
    xs = sorted(pts)
    mb = table_bytes / 1e6
    if mb <= xs[0]:
        return pts[xs[0]]
    if mb >= xs[-1]:
        return pts[xs[-1]]
    for lo, hi in zip(xs, xs[1:]):
        if lo <= mb <= hi:
            t = (np.log(mb) - np.log(lo)) / (np.log(hi) - np.log(lo))
            return float(np.exp((1 - t) * np.log(pts[lo]) + t * np.log(pts[hi])))


def profile_totals(lib):
    out = {}
    for kid in range(lib.grx_profile_kernel_count()):
        ms, cnt = ctypes.c_double(), ctypes.c_longlong()
        lib.grx_profile_read(kid, ctypes.byref(ms), ctypes.byref(cnt))
        if cnt.value:
            out[lib.grx_profile_kernel_name(kid).decode()] = (ms.value, cnt.value)
    return out


def cpu_baseline(G, args, X_features):
    """Oracle (plain-C port, single thread) on the same graph once, plus the reference-faithful legs of
    BASELINE.md baseline (1) on bounded samples."""
    from oracle import ckernels, refex, reference_path
    # the workload itself: weights, directions and attribute columns included (config 5 has all three)
    og = refex.OracleGraph(labels=G.labels, row_ptr=G.row_ptr, col=G.col, w=G.w, directed=G.directed,
                           num_edges=G.num_edges, t_row_ptr=getattr(G, 't_row_ptr', None), t_col=getattr(G, 't_col', None),
                           t_w=getattr(G, 't_w', None), adj_col=G.adj_col)
    if G.attributes:
        og.attrs = {'attribute_' + k: np.asarray(v, dtype=np.float64) for k, v in G.attributes.items()}
    # ONE thread: the oracle's row-parallel loops (OpenMP, there for the full-size parity tests) and its threaded
    # binning are switched off for the timing -- `cores` below says 1 and means it
    bin_threads = refex.BIN_THREADS
    ckernels.set_threads(1)
    refex.BIN_THREADS = 1
    try:
        t0 = time.perf_counter()
        res = refex.extract_features(og, max_generations=MAX_GENERATIONS, fast=True)
        dt = time.perf_counter() - t0
    finally:
        ckernels.set_threads(0)
        refex.BIN_THREADS = bin_threads
    gens = res.generation_count
    out = {
        'value': G.nnz * gens / dt, 'unit': 'edges/s', 'cores': 1, 'kind': 'port',
        'sample': f'full workload once: oracle C port, gen-0 + {gens} generations + pruning in {dt:.1f} s',
        'seconds': dt, 'host_cpus': os.cpu_count(),
    }
    extra = {}
    # the same run on every core of the host (OpenMP row loops of oracle/csrc/oracle_kernels.c, threaded binning):
    # what a competent multi-core CPU implementation of this restatement does on this box
    try:
        cores = os.cpu_count() or 1
        ckernels.set_threads(cores)
        refex.BIN_THREADS = cores
        t0 = time.perf_counter()
        res_all = refex.extract_features(og, max_generations=MAX_GENERATIONS, fast=True)
        dt_all = time.perf_counter() - t0
        extra['cpu_baseline_all_cores'] = {
            'value': G.nnz * res_all.generation_count / dt_all, 'unit': 'edges/s', 'cores': cores, 'kind': 'port',
            'sample': f'full workload once: oracle C port with OpenMP row loops ({cores} threads) and one binning thread per '
                      f'column, gen-0 + {res_all.generation_count} generations + pruning in {dt_all:.1f} s',
            'seconds': dt_all,
        }
        del res_all
    except Exception as exc:                       # baseline legs must never kill the bench line
        extra['cpu_baseline_all_cores'] = {'error': repr(exc)}
    finally:
        ckernels.set_threads(0)
        refex.BIN_THREADS = bin_threads
    ref = {'unit': 'edges/s', 'cores': 1, 'kind': 'port',
           'what': 'reference-faithful CPU path (BASELINE.md baseline 1): the pandas / networkx / scipy / sklearn call '
                   'sequences of the reference (oracle/reference_path.py, pinned on the reference\'s golden tables by '
                   'tests/test_oracle_pinned.py::test_reference_path_equals_golden)'}
    try:
        import pandas as pd
        names0 = res.trace[0].retained
        X0 = res.values[:, [res.columns.index(c) for c in names0]]
        first = G.n // 2
        legs = {}
        # (a) neighbour aggregation, features/extract.py:104-119: two contiguous samples (linearity)
        agg = []
        for sample in (min(args.cpu_sample_nodes, G.n), min(2 * args.cpu_sample_nodes, G.n)):
            dt_p, edges_p = reference_path.time_aggregate_sample(G.row_ptr, G.adj_col, X0, names0, first, sample)
            agg.append({'nodes': sample, 'edges': edges_p, 'seconds': dt_p, 'ms_per_node': 1e3 * dt_p / sample})
        legs['aggregate'] = agg
        ms_node = agg[-1]['ms_per_node']
        ref['value'] = agg[-1]['edges'] / agg[-1]['seconds']
        # (b) ego-net features, graph/interface/networkx.py:71-83
        n_ego = min(args.cpu_ego_nodes, G.n)
        Gs, rows = reference_path.networkx_sample_graph(G.row_ptr, G.col, first, n_ego)
        t0 = time.perf_counter()
        reference_path.egonet_rows_networkx(Gs, rows)
        dt_e = time.perf_counter() - t0
        legs['egonet'] = {'nodes': n_ego, 'seconds': dt_e, 'ms_per_node': 1e3 * dt_e / n_ego}
        # (c) pruning, features/prune.py:13-56,94-116: binning of every column + pdist, full height
        width = min(res.values.shape[1], args.cpu_prune_columns)
        frame = pd.DataFrame(res.values[:, :width])
        t0 = time.perf_counter()
        reference_path.prune_distances(frame)
        dt_b = time.perf_counter() - t0
        legs['prune'] = {'rows': G.n, 'columns': width, 'seconds': dt_b,
                         'note': 'one prune call: np.unique binning of every column + scipy pdist(chebychev), all rows'}
        ref['legs'] = legs
        total = ms_node * 1e-3 * G.n * gens + legs['egonet']['ms_per_node'] * 1e-3 * G.n + dt_b * (gens + 1)
        ref['extrapolated_refex_seconds'] = total
        ref['sample'] = (f'aggregation loop on {agg[0]["nodes"]} and {agg[1]["nodes"]} contiguous nodes of generation 1 '
                         f'({agg[0]["ms_per_node"]:.2f} / {agg[1]["ms_per_node"]:.2f} ms per node), nx.ego_graph loop on '
                         f'{n_ego} nodes ({legs["egonet"]["ms_per_node"]:.2f} ms per node), one full-height prune call '
                         f'({dt_b:.1f} s for {width} columns); whole ReFeX pass extrapolated: {ms_node * 1e-3 * G.n * gens / 60:.0f} min '
                         f'aggregation + {legs["egonet"]["ms_per_node"] * 1e-3 * G.n / 60:.0f} min ego-nets + {dt_b * (gens + 1):.0f} s pruning')
    except Exception as exc:                       # baseline legs must never kill the bench line
        ref['error'] = repr(exc)
    extra['cpu_reference_path'] = ref
    if X_features is not None and args.cpu_nmf:
        try:
            _, _, n_iter, dt_n = reference_path.sklearn_nmf(X_features, N_ROLES)
            from threadpoolctl import threadpool_info
            threads = max([p.get('num_threads', 1) for p in threadpool_info()] or [1])
            extra['cpu_baseline_nmf'] = {
                'value': n_iter / dt_n, 'unit': 'iters/s', 'cores': threads, 'kind': 'reference',
                'seconds': dt_n,
                'sample': f'sklearn NMF(mu, nndsvda) full fit on the {X_features.shape[0]}x{X_features.shape[1]} '
                          f'feature matrix: {n_iter} iterations in {dt_n:.1f} s (incl. init)',
            }
        except Exception as exc:
            extra['cpu_baseline_nmf'] = {'error': repr(exc)}
    return out, extra


def api_wall_once(G, with_roles=True):
    """One cold pair of the two calls a GraphRole user makes (graphrole/features/extract.py:65-96,
    graphrole/roles/extract.py:59-93) on the bench graph: a fresh adapter, nothing resident in HBM.  The extractor's
    own phase clock (RecursiveFeatureExtractor.wall, device synchronised between phases) gives the breakdown."""
    import gc
    import torch
    from graphrole_amd import RecursiveFeatureExtractor, RoleExtractor
    out = {}
    fe = RecursiveFeatureExtractor(G, max_generations=MAX_GENERATIONS, attributes=bool(G.attributes))
    fe.time_phases = True
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    host, dev, _ = fe.graph._device_graph()        # degree-descending relabelling, CSR upload
    dev.plan()
    if not host.directed and not host.weighted:
        dev.oriented()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    X = fe.extract_features()
    t2 = time.perf_counter()
    np.random.seed(0)
    rx = RoleExtractor(n_roles=N_ROLES)
    rx.extract_role_factors(X)
    t3 = time.perf_counter()
    out.update({'device_ingest_s': t1 - t0, 'extract_features_s': t2 - t0, 'extract_role_factors_s': t3 - t2,
                'total_s': t3 - t0, 'features_shape': list(X.shape),
                'extract_features_breakdown_s': {k: v for k, v in fe.wall.items()}})
    if with_roles:
        # the two properties a user reads next (graphrole/roles/extract.py:38-57): arg-max / row shares on the device
        # (grx_role_argmax / grx_row_normalise incl. the upload of the factor), then the reference's return types
        first = rx.dominant_role_index()
        t4 = time.perf_counter()
        roles = rx.roles
        t5 = time.perf_counter()
        share = rx.role_percentage
        t6 = time.perf_counter()
        out['roles'] = {'dominant_role_index_s': t4 - t3, 'roles_dict_s': t5 - t4, 'role_percentage_s': t6 - t5,
                        'nodes': len(first), 'roles_entries': len(roles), 'shape': list(share.shape),
                        'what': 'dominant_role_index() = upload + grx_role_argmax + int32 download; roles = the same + a '
                                'Python dict of n entries (the reference\'s return type); role_percentage = upload + '
                                'grx_row_normalise + download + DataFrame'}
        del first, roles, share
        # the reference's DEFAULT RoleExtractor(): MDL model selection over 2..8 roles x 1..8 bits, every cell an NMF fit
        # + two encodes + the KL cost (graphrole/roles/extract.py:98-142), on the same table
        np.random.seed(0)
        t7 = time.perf_counter()
        rsel = RoleExtractor()
        rsel.extract_role_factors(X)
        out['rolx_model_selection_s'] = time.perf_counter() - t7
        out['rolx_model_selection'] = {'selected_roles': int(rsel.node_role_factor.shape[1]),
                                       'cells': int(np.isfinite(rsel.model_selection_['error_costs']).sum())
                                       if getattr(rsel, 'model_selection_', None) else None}
        del rsel
    del X, rx, fe
    gc.collect()
    return out


def api_wall(G, args, reps=3, with_roles=True):
    """`reps` cold pairs in a row (every one a fresh adapter and a fresh result table); the top-level figures are
    those of the MEDIAN repetition by total_s, min / median / max beside them."""
    runs = [api_wall_once(G, with_roles=with_roles and i == reps - 1) for i in range(reps)]
    order = sorted(range(reps), key=lambda i: runs[i]['total_s'])
    med = dict(runs[order[reps // 2]])
    if with_roles and 'roles' not in med:
        med['roles'] = runs[-1].get('roles')
    for key in ('rolx_model_selection_s', 'rolx_model_selection'):
        if with_roles and key in runs[-1]:
            med[key] = runs[-1][key]
    out = {}
    if not G.directed and not G.weighted:
        from graphrole_amd.graph.csr import CSRGraph
        rows = np.repeat(np.arange(G.n, dtype=np.int64), np.diff(G.row_ptr))
        upper = rows <= G.col
        src, dst = rows[upper], G.col[upper].astype(np.int64)
        t0 = time.perf_counter()
        CSRGraph(G.n, src, dst, validate=False)
        out['csr_from_edge_arrays_s'] = time.perf_counter() - t0
    out.update(med)
    totals = [r['total_s'] for r in runs]
    out['total_s_min_median_max'] = [min(totals), sorted(totals)[reps // 2], max(totals)]
    out['repetitions'] = [{k: r[k] for k in ('total_s', 'device_ingest_s', 'extract_features_s', 'extract_role_factors_s',
                                             'extract_features_breakdown_s')} for r in runs]
    out['what'] = ('cold RecursiveFeatureExtractor(CSRGraph).extract_features() -> DataFrame (device_ingest_s = '
                   'relabelling + CSR / plan / oriented-graph upload, included in extract_features_s) and '
                   'RoleExtractor(6).extract_role_factors(X) incl. encode; %d cold pairs in a row, the figures are the '
                   'median pair\'s' % reps)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--workload', default='ba1m', choices=list(WORKLOADS))
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--agg-lanes', type=int, default=0, help='override the lane-group width of grx_aggregate (tuning)')
    ap.add_argument('--cpu-sample-nodes', type=int, default=20000,
                    help='reference-path aggregation samples: this many and twice as many contiguous nodes '
                         '(BASELINE.md section 3 sizes: 20000 / 40000, ~40 s of single-core pandas; the cost per node '
                         'is linear, so smaller samples give the same extrapolation)')
    ap.add_argument('--cpu-ego-nodes', type=int, default=400)
    ap.add_argument('--cpu-prune-columns', type=int, default=24)
    ap.add_argument('--no-api-wall', action='store_true')
    ap.add_argument('--no-sharded-extra', action='store_true',
                    help='N > 1: skip the additional sharded config-5 (dw5m) measurement')
    ap.add_argument('--cpu-nmf', type=int, default=1)
    ap.add_argument('--pmc', default='auto', choices=['auto', 'on', 'off'],
                    help='same-run hardware counters: re-run two steps under rocprofv3 --pmc (one pass per counter group) '
                         'and fold HBM traffic / L2 hit rate / MFMA busy into the line; auto = when N = 1 and rocprofv3 exists')
    ap.add_argument('--pmc-child', action='store_true', help='internal: the process a --pmc pass profiles (two steps, no output)')
    ap.add_argument('--soak-seconds', type=float, default=2.5,
                    help='after the timed region: keep stepping (untimed) for this long so that an external GPU-activity '
                         'sampler sees the device busy; 0 disables')
    args = ap.parse_args()

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        # `python bench.py --gpus N` without a launcher: become the launcher (one rank per GPU, rendezvous on the
        # loopback address -- the container hostname may not resolve)
        import socket
        with socket.socket() as sock:
            sock.bind(('127.0.0.1', 0))
            port = sock.getsockname()[1]
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={args.gpus}',
               '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.stdout.flush()
        os.execv(sys.executable, cmd)

    import torch
    import torch.distributed as dist
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a GPU (the product path has no CPU fallback)')
    # GRX_BENCH_SHARE_GPU=1: functional check of the N > 1 path on a one-GPU box (all ranks on
    # cuda:0, gloo).  Numbers from that mode are meaningless; the driver never sets it.
    share_gpu = os.environ.get('GRX_BENCH_SHARE_GPU') == '1'
    if share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    # GRX_FORCE_COLLECTIVES=1 under torchrun with one rank: the sharded code path with its real RCCL
    # calls on a one-GPU box (functional check, see graphrole_amd/parallel.py)
    multi = world > 1 or (os.environ.get('GRX_FORCE_COLLECTIVES') == '1' and 'RANK' in os.environ)
    if multi:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if share_gpu:
            dist.init_process_group('gloo')
        else:
            dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))
    if world != args.gpus:
        raise SystemExit(f'bench.py: --gpus {args.gpus} but the launcher set WORLD_SIZE={world}')

    import shutil
    have_rocprof = bool(shutil.which('rocprofv3') or os.path.exists('/opt/rocm/bin/rocprofv3'))
    under_profiler = any(k.startswith(('ROCPROF', 'ROCP_')) for k in os.environ)
    use_pmc = (not args.pmc_child and world == 1 and not multi and
               (args.pmc == 'on' or (args.pmc == 'auto' and have_rocprof and not under_profiler)))

    def run_workload(workload, steps, warmup, light):
        """One measured workload; light = the extra sharded line of an N > 1 run (no breakdown pass, no CPU legs)."""
        args_workload = workload
        from graphrole_amd import RecursiveFeatureExtractor, _lib, backend
        from graphrole_amd.roles import factor
        K = backend.get()
        lib = _lib.load()

        G = build_graph(args_workload, world, int(os.environ.get('LOCAL_RANK', '0')), share=use_pmc or args.pmc_child)
        fe = RecursiveFeatureExtractor(G, max_generations=MAX_GENERATIONS, distributed=multi,
                                      attributes=bool(G.attributes))
        dev_graph = fe.graph._device_graph()[1]        # graph resident in HBM before anything is timed
        if args.agg_lanes:
            dev_graph.plan().set_lanes(args.agg_lanes)
        plan = fe._shard()
        # the NMF passes cost the same for every row: its row shards are EQUAL ROW COUNTS, as RoleExtractor._plan cuts
        # them (roles/extract.py), not the nnz-balanced ranges of ReFeX (whose last rank holds 15 x the rows of the
        # first on the power-law graph: profiles/r06_projected_scaling_*.json)
        nmf_plan = None
        if plan is not None:
            from graphrole_amd import parallel
            nmf_plan = parallel.maybe_plan(np.zeros(G.n + 1, dtype=np.int64), True)
        rng = np.random.RandomState(0)

        def barrier():
            torch.cuda.synchronize()
            if multi:
                dist.barrier()
            torch.cuda.synchronize()

        state = {}

        def step(timers):
            fe.reset()
            t0 = time.perf_counter()
            fe.run_on_device()
            names, cols = fe.device_features()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            Xd = K.gather_columns(cols, G.n)
            omega = rng.normal(size=(len(names), N_ROLES + 10))
            nmf_state, n_iter = factor.nmf_device(Xd, G.n, N_ROLES, omega, plan=nmf_plan)
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            timers['refex'] += t1 - t0
            timers['nmf'] += t2 - t1
            timers['nmf_iters'] += n_iter
            state.update(names=names, Xd=Xd, n_iter=n_iter, gens=fe.generation_count, stats=list(fe.stats),
                         F=len(names), W=nmf_state.W)

        warm = dict(refex=0.0, nmf=0.0, nmf_iters=0)
        if args.pmc_child:                              # the process a counter pass profiles: two plain steps
            step(warm)
            step(warm)
            torch.cuda.synchronize()
            return None
        for _ in range(warmup):
            step(warm)
        # untimed breakdown pass: HIP events around EVERY kernel launch (per-kernel ms per step)
        lib.grx_profile_reset()
        lib.grx_profile_select(0)
        lib.grx_profile_enable(1)
        step(dict(refex=0.0, nmf=0.0, nmf_iters=0))
        torch.cuda.synchronize()
        breakdown = profile_totals(lib)
        # timed region: events only around the kernel the `roofline` object is about
        names = {lib.grx_profile_kernel_name(i).decode(): i for i in range(lib.grx_profile_kernel_count())}
        mask = 0
        # (only the dominant kernel: every event record costs the host about as much as a launch, and twenty W-pass
        # launches per step with two events each made the small workloads host-bound; the W pass of `roofline_nmf` is
        # timed in the untimed breakdown pass above)
        for kname in ('aggregate_kernel', 'aggregate_hub_kernel'):
            mask |= 1 << names[kname]
        lib.grx_profile_reset()
        lib.grx_profile_select(mask)
        timers = dict(refex=0.0, nmf=0.0, nmf_iters=0)
        barrier()
        t_start = time.perf_counter()
        for _ in range(steps):
            step(timers)
        barrier()
        elapsed = time.perf_counter() - t_start
        lib.grx_profile_enable(0)
        prof = profile_totals(lib)
        # the same steps once more WITHOUT the per-launch events of the timed region (what a caller runs: no event
        # records between the small launches) -- reported next to `value`,
        # never instead of it
        plain = dict(refex=0.0, nmf=0.0, nmf_iters=0)
        barrier()
        t_plain = time.perf_counter()
        for _ in range(steps):
            step(plain)
        barrier()
        t_plain = time.perf_counter() - t_plain
        # the cold API pairs, first set: right after the timed region (before the soak loop and the counter passes, whose
        # influence on the host side of the calls the second set below makes visible)
        api_before = None
        if rank == 0 and world == 1 and not args.no_api_wall and not light:
            try:
                api_before = api_wall(G, args, reps=3, with_roles=False)
            except Exception as exc:
                api_before = {'error': repr(exc)}
        # untimed soak: the timed region of the small workloads is tens of milliseconds -- an external activity sampler
        # (the driver polls the SMI once a second) would never see the device busy
        soak = None
        # (N = 1 only: the loop is bounded by TIME, and ranks that ran different step counts would leave each other
        # waiting inside an exchange)
        if args.soak_seconds > 0 and not light and not multi:
            t_soak, n_soak = time.perf_counter(), 0
            while time.perf_counter() - t_soak < args.soak_seconds:
                step(dict(refex=0.0, nmf=0.0, nmf_iters=0))
                n_soak += 1
            barrier()
            soak = {'seconds': time.perf_counter() - t_soak, 'steps': n_soak, 'what': 'untimed steps after the timed region'}
        # N > 1: one more (untimed) step with HIP events around every exchange -> exchange share of a step
        exchange = None
        if plan is not None:
            plan.reset_timing()
            plan.timing = True
            barrier()
            t_x = time.perf_counter()
            step(dict(refex=0.0, nmf=0.0, nmf_iters=0))
            barrier()
            t_x = time.perf_counter() - t_x
            plan.timing = False
            stats = plan.collect_timing()
            exchange = {'step_ms': t_x * 1e3, 'ms': sum(v[1] for v in stats.values()), 'calls': sum(v[0] for v in stats.values()),
                        'by_kind': {k: {'calls': v[0], 'ms': v[1]} for k, v in stats.items()}}

        # max over ranks
        red = torch.tensor([elapsed, timers['refex'], timers['nmf']], dtype=torch.float64,
                           device='cpu' if share_gpu else 'cuda')
        if multi:
            dist.all_reduce(red, op=dist.ReduceOp.MAX)
        elapsed, t_refex, t_nmf = [float(x) for x in red.cpu()]
        # N > 1 soak: bounded by a STEP COUNT every rank derives from the same reduced time (a time-bounded loop would
        # leave ranks that ran different counts waiting inside an exchange) -- an external sampler sees N busy devices
        if args.soak_seconds > 0 and not light and multi:
            n_soak = max(1, min(2000, int(np.ceil(args.soak_seconds / max(elapsed / steps, 1e-6)))))
            t_soak = time.perf_counter()
            for _ in range(n_soak):
                step(dict(refex=0.0, nmf=0.0, nmf_iters=0))
            barrier()
            soak = {'seconds': time.perf_counter() - t_soak, 'steps': n_soak,
                    'what': 'untimed steps after the timed region, the same count on every rank'}

        # RolX encode of the node-role factor, outside the timed steps: the reference's quantiser reproduced
        # (grx_kmeans1d, the default of RoleExtractor) and the Lloyd-Max solver (quantizer='lloyd_max')
        encode_info = None
        if rank == 0 and world == 1 and not light:
            Wd = state['W']
            n_bins = 2 ** int(np.log2(N_ROLES * min(G.n, state['F'])))          # roles/extract.py:72
            flat = K.transpose(Wd, N_ROLES, G.n).reshape(-1)                      # the reference's flatten order (n x r)
            encode_info = {'values': int(flat.numel()), 'n_bins': n_bins}
            for label, fn in (('kmeans', K.kmeans1d), ('lloyd_max', K.lloyd_max)):
                fn(flat, n_bins)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                _, _, info = fn(flat, n_bins)
                torch.cuda.synchronize()
                encode_info[label] = {'ms': (time.perf_counter() - t0) * 1e3, 'iterations': int(info[0]),
                                      'distinct_levels': int(info[2])}
            encode_info['what'] = ('encode() of the N x r node-role factor: kmeans = sklearn KMeans(random_state=1) reproduced '
                                   '(grx_kmeans1d), lloyd_max = grx_lloyd_max')
        # RolX's real cost per call (roles/extract.py:59-93,144-161): NMF fit + encode(G) + encode(F) on the same device
        # table the steps factorise, as RoleExtractor(n_roles=6).extract_role_factors runs them -- outside `value`
        rolx_info = None
        if rank == 0 and world == 1 and not light:
            try:
                n_bits = int(np.log2(N_ROLES * min(G.n, state['F'])))               # roles/extract.py:72
                shape = (G.n, state['F'])
                np.random.seed(0)
                factor.encoded_factors_device(state['Xd'], shape, N_ROLES, n_bits, 'kmeans')
                torch.cuda.synchronize()
                reps_r = 3
                t0 = time.perf_counter()
                for _ in range(reps_r):
                    factor.encoded_factors_device(state['Xd'], shape, N_ROLES, n_bits, 'kmeans')
                torch.cuda.synchronize()
                rolx_info = {'ms_per_call': (time.perf_counter() - t0) / reps_r * 1e3, 'n_roles': N_ROLES, 'n_bits': n_bits,
                             'calls_timed': reps_r,
                             'what': 'NMF fit (NNDSVDa + MU loop) + encode(G) + encode(F) with the reference\'s quantiser '
                                     '(grx_kmeans1d), device time, the steps\' own feature table'}
            except Exception as exc:
                rolx_info = {'error': repr(exc)}

        # per-rank figures (N > 1): every rank's aggregation launch time and exchange time, gathered on rank 0
        per_rank = None
        if multi:
            agg_ms_r, agg_cnt_r = prof.get('aggregate_kernel', (0.0, 0))
            hub_ms_r, _ = prof.get('aggregate_hub_kernel', (0.0, 0))
            w_ms_r, w_cnt_r = breakdown.get('nmf_w_pass_kernel', (0.0, 0))
            mine = {'rank': rank, 'rows': (plan.row_end - plan.row_begin) if plan is not None else G.n,
                    'aggregate_avg_launch_ms': (agg_ms_r + hub_ms_r) / agg_cnt_r if agg_cnt_r else None,
                    'w_pass_avg_launch_ms': w_ms_r / w_cnt_r if w_cnt_r else None, 'exchange': exchange}
            gathered = [None] * world
            dist.all_gather_object(gathered, mine)
            per_rank = gathered
        if rank == 0:
            gens = state['gens']
            edges_per_step = G.nnz * gens
            # headline: edges aggregated per second of the WHOLE step (ReFeX pass + NMF), consistent with
            # ms_per_step; the per-phase rates are in refex.edges_per_s and nmf.iters_per_s
            value = edges_per_step * steps / elapsed
            # aggregation-kernel roofline (algorithmic bytes per launch)
            f_per_gen = [s['candidates'] // 2 for s in state['stats'] if s['generation'] >= 1]
            rows_per_rank = G.n / world
            nnz_per_rank = G.nnz / world
            alg_bytes = 0.0
            launches = 0
            for f in f_per_gen:
                for c0 in range(0, f, 16):
                    fc = min(16, f - c0)
                    alg_bytes += nnz_per_rank * 4 + (rows_per_rank + 1) * 8 + G.n * fc * 8 / world + rows_per_rank * 2 * fc * 8
                    launches += 1
            agg_ms, agg_cnt = prof.get('aggregate_kernel', (0.0, 0))
            hub_ms, _ = prof.get('aggregate_hub_kernel', (0.0, 0))
            roofline = None
            if agg_cnt:
                per_launch_ms = (agg_ms + hub_ms) / agg_cnt
                achieved = alg_bytes / launches / (per_launch_ms * 1e-3) / 1e9
                traffic = nmf_traffic = traffic_source = nmf_mfma = agg_l2 = None
                traffic_stale = None
                tpath = os.path.join(ROOT, 'profiles', 'traffic_latest.json')
                pmc = pmc_counters(args_workload) if (use_pmc and not light) else None
                if pmc and 'error' not in pmc:
                    traffic = pmc['aggregate_traffic_per_launch']
                    agg_l2 = pmc.get('aggregate_l2_hit_rate')
                    nmf_traffic = pmc.get('nmf_w_pass_traffic_per_launch')
                    nmf_mfma = pmc.get('nmf_w_pass_mfma')
                    traffic_stale = False
                    traffic_source = ('measured in this run: rocprofv3 --pmc passes (%s) of two steps of this command on the same '
                                      'graph files and libgrx.so, started by bench.py itself after the timed region'
                                      % '; '.join(PMC_PASSES))
                elif os.path.exists(tpath):
                    try:
                        table = json.load(open(tpath))
                        tj = table.get(args_workload) if 'workload' not in table else \
                            (table if table.get('workload') == args_workload else None)
                        if tj and tj.get('n_gpus', 1) == world:
                            # counters cannot be read in-process: they come from the committed --pmc passes, and only
                            # when those passes profiled THIS library and THIS bench.py (sha256 recorded by
                            # tools/profile_gpu.sh); otherwise the line says so instead of carrying stale traffic
                            import hashlib
                            digest = lambda path: hashlib.sha256(open(path, 'rb').read()).hexdigest()
                            traffic_stale = not (tj.get('lib_sha256') == digest(_lib.LIB_PATH) and
                                                 tj.get('bench_sha256') == digest(os.path.abspath(__file__)))
                            traffic_source = (f'static: {tj.get("source")} (committed rocprofv3 --pmc passes of this command, '
                                              'tools/profile_gpu.sh; NOT measured in this run'
                                              + ('; STALE: collected with another libgrx.so / bench.py, not attached)' if traffic_stale
                                                 else '; same libgrx.so and bench.py by sha256)'))
                            if not traffic_stale:
                                traffic = tj.get('aggregate_kernel_hbm_bytes_per_launch')
                                nmf_traffic = tj.get('nmf_w_pass_hbm_bytes_per_launch')
                                nmf_mfma = tj.get('nmf_w_pass_mfma')
                                agg_l2 = tj.get('aggregate_l2_hit_rate')
                    except Exception:
                        traffic = nmf_traffic = None
                roofline = {'bound': 'hbm', 'kernel': 'aggregate_packed_kernel (bit-packed integer rows, generations 1-2 of unweighted graphs) / aggregate_kernel '
                                      '(fp64 rows) + aggregate_combine_kernel for rows longer than 128',
                            'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': achieved / HBM_PEAK_GBS,
                            'traffic': traffic, 'traffic_stale': traffic_stale, 'traffic_source': traffic_source,
                            'l2_hit_rate': agg_l2, 'pmc': pmc,
                            'algorithmic_bytes_per_launch': alg_bytes / launches,
                            'avg_launch_ms': per_launch_ms, 'launches': agg_cnt, 'f_prev_per_generation': f_per_gen}
                # The kernel is a random-row gather: one request per CSR entry, and the chip's request rate -- not its
                # byte rate -- is what bounds it (profiles/r04_gather_bw.json: 16-, 32- and 64-byte rows gather at the same
                # rows/s; the rate follows the TABLE's bytes through the L2 hit rate).  Ceiling of this step = the
                # microbenchmark's time for the same number of gathers from tables of the launches' sizes, with the index
                # distribution of the workload (power-law hubs first for the preferential-attachment graphs).
                dist_kind = 'uniform' if WORKLOADS[args_workload][0] == 'er' else 'powerlaw'
                ceil_s, per_gen = 0.0, []
                for s_gen in state['stats']:
                    rb = s_gen.get('gather_row_bytes', 0)
                    if s_gen['generation'] < 1 or not rb:
                        continue
                    n_launch = -(-rb // 128)                        # rows wider than 128 bytes: one launch per 16 columns
                    rate = gather_ceiling(G.n * min(rb, 128), dist_kind)
                    per_gen.append({'generation': s_gen['generation'], 'row_bytes': rb, 'table_mb': G.n * rb / 1e6,
                                    'launches': n_launch, 'ceiling_rows_per_s': rate})
                    if rate:
                        ceil_s += n_launch * nnz_per_rank / rate
                roofline['gather_rows_per_s'] = nnz_per_rank / (per_launch_ms * 1e-3)
                roofline['gather_ceiling'] = {
                    'source': 'profiles/r04_gather_bw.json (tools/microbench/gather_bw.hip on MI355X: the gather alone, best of '
                              'unroll x grid variants; index distribution: %s)' % dist_kind,
                    'per_generation': per_gen, 'ceiling_ms_per_step': ceil_s * 1e3 if ceil_s else None,
                    'measured_ms_per_step': (agg_ms + hub_ms) / steps}
                roofline['frac_of_gather_ceiling'] = (ceil_s * 1e3) / ((agg_ms + hub_ms) / steps) if ceil_s else None
            F, r = state['F'], N_ROLES
            w_ms, w_cnt = breakdown.get('nmf_w_pass_kernel', (0.0, 0))
            roofline_nmf = None
            if w_cnt:
                nmf_bytes = (G.n / world) * (F * 8 + 2 * r * 8)
                ach = nmf_bytes / (w_ms / w_cnt * 1e-3) / 1e9
                roofline_nmf = {'bound': 'hbm', 'kernel': 'nmf_w_pass_mfma_kernel (fp64 MFMA 16x16x4)', 'achieved': ach,
                                'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': ach / HBM_PEAK_GBS,
                                'traffic': nmf_traffic if agg_cnt else None,
                                'traffic_source': traffic_source if agg_cnt else None,
                                'mfma_util': (nmf_mfma or {}).get('mfma_util') if agg_cnt else None,
                                'mfma_counters': nmf_mfma if agg_cnt else None,
                                'algorithmic_bytes_per_launch': nmf_bytes, 'avg_launch_ms': w_ms / w_cnt,
                                'avg_launch_source': 'HIP events around every launch of one untimed step (breakdown pass)'}
            line = {
                'metric': 'ReFeX edges-aggregated/sec (+ RolX NMF iters/sec in nmf.iters_per_s), 1M-node graph',
                'value': value, 'unit': 'edges/s', 'n_gpus': world, 'steps': steps, 'warmup': warmup,
                'ms_per_step': elapsed / steps * 1e3, 'higher_is_better': True, 'scaling': 'strong',
                'ms_per_step_without_launch_events': t_plain / steps * 1e3,
                'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
                'config': {'workload': args_workload, 'description': WORKLOADS[args_workload][3], 'n_nodes': G.n,
                           'n_edges': G.num_edges, 'nnz': G.nnz, 'max_generations': MAX_GENERATIONS,
                           'recursive_generations_executed': gens, 'n_roles': N_ROLES, 'n_features': F,
                           'sharding': ('node-range x%d, loops and exchanges below the C ABI (grx_comm: RCCL grouped send / recv + '
                                        'all-reduce over xGMI): per generation candidate row slices to the column owners, '
                                        'bins back, all-gather of the retained columns only; one all-reduce per NMF '
                                        'iteration' % world) if world > 1 else 'single GPU'},
                'refex': {'ms_per_step': t_refex / steps * 1e3, 'edges_per_step': edges_per_step,
                          'edges_per_s': edges_per_step * steps / t_refex,
                          'generations': state['stats']},
                'nmf': {'iters_per_s': timers['nmf_iters'] / t_nmf, 'ms_per_step': t_nmf / steps * 1e3,
                        'iterations_per_step': state['n_iter'], 'includes': 'NNDSVDa init + MU loop + convergence checks'},
                'soak': soak, 'encode': encode_info, 'rolx': rolx_info, 'roofline': roofline, 'roofline_nmf': roofline_nmf,
                'kernel_ms_per_step': {k: v[0] for k, v in sorted(breakdown.items(), key=lambda kv: -kv[1][0])},
            }
            if per_rank is not None:
                # per-rank roofline of the aggregation kernel (each rank owns ~1/N of the rows and of the nnz) and the
                # share of a step spent in exchanges: what a scaling curve has to be read against
                for pr in per_rank:
                    if pr['aggregate_avg_launch_ms'] and launches:
                        ach = alg_bytes / launches / (pr['aggregate_avg_launch_ms'] * 1e-3) / 1e9
                        pr['aggregate_achieved_gbs'] = ach
                        pr['aggregate_frac_of_hbm_peak'] = ach / HBM_PEAK_GBS
                    if pr['exchange']:
                        pr['exchange_share_of_step'] = pr['exchange']['ms'] / pr['exchange']['step_ms']
                line['per_rank'] = per_rank
            # the cold API calls come FIRST: the CPU legs below (pandas / networkx / sklearn, their thread pools and
            # multi-GB temporaries) leave the process in a state in which the 4.6 GB result table of dw5m takes 1.3 s
            # instead of 0.3 s to materialise -- not what a user of the two calls sees
            if world == 1 and not args.no_api_wall and not light:
                try:
                    line['api_wall_s'] = api_wall(G, args, reps=3)
                    line['api_wall_s']['when'] = 'after the soak loop and the rocprofv3 --pmc child processes'
                    # the reference's default RoleExtractor() (n_roles=None): wall-clock of the whole MDL grid, cold call
                    line['rolx_model_selection_s'] = line['api_wall_s'].get('rolx_model_selection_s')
                except Exception as exc:
                    line['api_wall_s'] = {'error': repr(exc)}
                line['api_wall_before_soak_s'] = api_before
            if world == 1 and not args.no_cpu_baseline and not light:
                Xh = K.to_host(state['Xd'])[:, :G.n].T.copy() if args.cpu_nmf else None
                base, extra = cpu_baseline(G, args, Xh)
                line['cpu_baseline'] = base
                line.update(extra)
            refp = line.get('cpu_reference_path') or {}
            if 'extrapolated_refex_seconds' in refp and 'total_s' in (line.get('api_wall_s') or {}):
                nmf_s = (line.get('cpu_baseline_nmf') or {}).get('seconds', 0.0)
                line['speedup_vs_reference_path'] = {
                    'value': (refp['extrapolated_refex_seconds'] + nmf_s) / line['api_wall_s']['total_s'],
                    'what': 'reference-faithful CPU path (extrapolated ReFeX pass + measured sklearn NMF, without its '
                            'KMeans encode) / api_wall_s.total_s; north_star target: >= 10'}
            return line
        return None

    line = run_workload(args.workload, args.steps, args.warmup, light=False)
    if multi and not args.no_sharded_extra and args.workload != 'dw5m':
        # the workload where sharding pays (BASELINE config 5: 90 ms per step on one GPU), beside the headline
        extra = run_workload('dw5m', max(2, min(args.steps, 5)), 1, light=True)
        if rank == 0 and extra is not None:
            line['sharded_dw5m'] = {k: extra[k] for k in ('value', 'unit', 'ms_per_step', 'n_gpus', 'steps', 'config', 'refex',
                                                         'nmf', 'roofline', 'per_rank') if k in extra}
    if multi:
        dist.barrier()
        dist.destroy_process_group()
    for path in _PUBLISHED:                             # mapped copies in other processes stay valid after the unlink
        import shutil
        shutil.rmtree(path, ignore_errors=True)
    if rank == 0 and line is not None:
        # the JSON line is the LAST thing on stdout: librccl prints a version banner through C stdio, which would
        # otherwise be flushed after Python's own buffer when the process exits
        sys.stdout.flush()
        try:
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        print(json.dumps(line))
        sys.stdout.flush()


if __name__ == '__main__':
    main()
