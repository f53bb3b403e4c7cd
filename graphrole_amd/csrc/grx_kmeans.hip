// grx_kmeans.hip -- the reference's quantiser, reproduced: 1-D k-means exactly as
// sklearn.cluster.KMeans(n_clusters=k, random_state=1).fit(values.reshape(-1, 1)) runs it.
//
// Reference call site: encode(), graphrole/roles/factor.py:29-49 -- every factor entry is replaced by
// cluster_centers_[labels_] of that fit.  Because the fit is only loosely converged (tol = 1e-4 * var(X), a
// handful of Lloyd iterations), its result is determined by the k-means++ seeding, and the MDL model selection
// of RolX (roles/extract.py:98-142) inherits that: a better quantiser picks other cells.  So the procedure itself
// is reproduced (sklearn 1.7.2; cluster/_kmeans.py:163-257 _kmeans_plusplus, :620-720 _kmeans_single_lloyd,
// :1445-1545 fit; _k_means_lloyd.pyx / _k_means_common.pyx; metrics/pairwise.py _euclidean_distances):
//   * the caller draws the random numbers exactly as RandomState(1) hands them to sklearn (they do not depend
//     on the data): the first seed index and n_trials uniforms per further seed;
//   * mean-centring, tolerance, squared distances in sklearn's operation order (-2 x c + c^2 + x^2, clipped);
//   * k-means++: cumulative sum of the closest distances -> searchsorted of uniform * potential -> the candidate
//     with the smallest potential wins;  per seed: update + tile sums, pick, potentials, choose (4 launches, no
//     host round trip);
//   * Lloyd on the sorted values with prefix sums (labels are intervals): first-minimum E step decided with
//     sklearn's own expression c^2 - 2 x c at the interval ends, centres = sum * (1 / count), relocation of
//     empty clusters to the farthest points, stop on unchanged labels or total squared centre shift <= tol,
//     a last E step when the labels had not settled.
// What cannot be bit-identical: sums are reduced in another order than numpy's cumsum / BLAS / OpenMP partials
// (relative 1e-13), so a uniform draw that lands within that distance of a boundary of the cumulative sum picks a
// neighbouring point (probability ~1e-6 per draw at 6 M values), and tied candidate potentials are recognised
// with a 1e-12 tolerance (see km_choose_kernel).  Centres agree with sklearn to ~1e-12 otherwise.
#include "grx_common.h"

#include <cstdlib>

int grx_internal_sort_columns(int64_t n, int ncols, const double *cols, int64_t ld, double *out, int64_t out_ld,
                              void *workspace, hipStream_t st);
extern "C" size_t grx_sort_workspace_bytes(int64_t n, int ncols);

namespace {

constexpr int KM_TILE = 1024;              // values per tile of the cumulative sum (256 threads x 4)
constexpr int KM_MAX_TRIALS = 16;          // 2 + int(log(k)) <= 13 for k <= 65536
constexpr int KM_MAX_K = 8192;              // the E step ranks the centres by counting: O(k^2) per iteration

__device__ __forceinline__ double km_sqdist(double c, double csq, double x)
{
    // sklearn _euclidean_distances: -2 * (x . c), += |c|^2, += |x|^2, clipped at 0 -- every step rounded
    double d = __dmul_rn(-2.0, __dmul_rn(x, c));
    d = __dadd_rn(d, csq);
    d = __dadd_rn(d, __dmul_rn(x, x));
    return d > 0.0 ? d : 0.0;
}

struct KmState {                 // device scalars shared by the kernels of one run
    double mean, tol, pot;
    double cand_x[KM_MAX_TRIALS];
    int64_t cand_id[KM_MAX_TRIALS];
    double best_x;
    int64_t best_id;
    unsigned bar_count, bar_gen;   // grid barrier of the persistent seeding kernel
};

// fixed-shape workgroup sum (256 threads): wave butterflies, then the four wave totals in order
__device__ __forceinline__ double km_block_sum(double v, double *red)
{
    v = grx_group_sum<64>(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return ((red[0] + red[1]) + red[2]) + red[3];
}

// the same sum for one of several 256-thread sub-blocks of a larger workgroup (lt = thread id inside the sub-block,
// red = the sub-block's four slots): identical tree, so a sub-block of the persistent kernel produces the bits a
// 256-thread workgroup of the per-seed kernels does
__device__ __forceinline__ double km_subblock_sum(double v, double *red, int lt)
{
    v = grx_group_sum<64>(v);
    __syncthreads();
    if ((lt & 63) == 0) red[lt >> 6] = v;
    __syncthreads();
    return ((red[0] + red[1]) + red[2]) + red[3];
}

// ---- moments ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void km_sum_kernel(const double *__restrict__ v, int64_t m,
                                                     const KmState *__restrict__ st, int square,
                                                     double *__restrict__ part)
{
    __shared__ double red[4];
    const double shift = square ? st->mean : 0.0;
    double s = 0.0;
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < m; i += stride) {
        const double x = v[i] - shift;
        s += square ? x * x : x;
    }
    s = km_block_sum(s, red);
    if (threadIdx.x == 0) part[blockIdx.x] = s;
}

__global__ __launch_bounds__(64) void km_moment_final_kernel(const double *__restrict__ part, int nb, int64_t m, int which,
                                                             double rel_tol, KmState *st)
{
    if (threadIdx.x != 0) return;
    double s = 0.0;
    for (int b = 0; b < nb; ++b) s += part[b];
    if (which == 0) st->mean = s / (double)m;
    else st->tol = (s / (double)m) * rel_tol;                  // _tolerance: mean(var(X, axis=0)) * tol
}

// x = v - mean; d = squared distance to the first seed; tile sums of d
__global__ __launch_bounds__(256) void km_init_kernel(const double *__restrict__ v, int64_t m, int64_t first,
                                                      const KmState *__restrict__ st, double *__restrict__ x,
                                                      double *__restrict__ d, double *__restrict__ tsum,
                                                      double *__restrict__ seeds_x, int64_t *__restrict__ seeds_id)
{
    __shared__ double red[4];
    const double mean = st->mean;
    const double c = v[first] - mean, csq = __dmul_rn(c, c);
    const int64_t base = (int64_t)blockIdx.x * KM_TILE;
    double s = 0.0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int64_t i = base + j * 256 + threadIdx.x;
        if (i < m) {
            const double xi = v[i] - mean;
            x[i] = xi;
            const double di = km_sqdist(c, csq, xi);
            d[i] = di;
            s += di;
        }
    }
    s = km_block_sum(s, red);
    if (threadIdx.x == 0) tsum[blockIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) { seeds_x[0] = c; seeds_id[0] = first; }
}

// ---- k-means++ : one further seed = pick -> potentials -> choose -> update ---------------------------
// pick: inclusive scan of the tile sums (potential = last), then per trial r = uniform * potential and the first
// index whose cumulative sum reaches r: binary search over the tile prefixes, sequential additions inside the tile
__device__ __forceinline__ void km_pick_body(const double *__restrict__ x, const double *__restrict__ d,
                                             int64_t m, double *__restrict__ tsum, int64_t ntiles,
                                             const double *__restrict__ uniform, int n_trials, int first_call,
                                             KmState *st, double *s_scan, double *s_pot_p)
{
    double &s_pot = *s_pot_p;
    {
        // inclusive prefixes of the tile sums in place: a contiguous chunk of tiles per thread, the chunk totals
        // scanned across the workgroup
        const int64_t chunk = (ntiles + 1023) / 1024;
        const int64_t t0 = (int64_t)threadIdx.x * chunk, t1 = (t0 + chunk < ntiles) ? t0 + chunk : ntiles;
        double local = 0.0;
        for (int64_t t = t0; t < t1; ++t) local += tsum[t];
        s_scan[threadIdx.x] = local;
        __syncthreads();
        for (int off = 1; off < 1024; off <<= 1) {
            const double add = (threadIdx.x >= (unsigned)off) ? s_scan[threadIdx.x - off] : 0.0;
            __syncthreads();
            s_scan[threadIdx.x] += add;
            __syncthreads();
        }
        double run = s_scan[threadIdx.x] - local;
        for (int64_t t = t0; t < t1; ++t) { run += tsum[t]; tsum[t] = run; }
        if (threadIdx.x == 0) {
            // sklearn carries candidates_pot[best] as the potential; the first time it is closest_dist_sq @ weights
            s_pot = first_call ? s_scan[1023] : st->pot;
            if (first_call) st->pot = s_scan[1023];
        }
        __threadfence_block();
        __syncthreads();
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (wave >= n_trials) return;
    const double r = uniform[wave] * s_pot;
    // first tile whose inclusive prefix reaches r
    int64_t lo = 0, hi = ntiles - 1;
    while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (tsum[mid] < r) lo = mid + 1; else hi = mid; }
    const int64_t tile = lo;
    const int64_t base = tile * KM_TILE;
    // inside the tile: running cumulative sum 64 values at a time (wave scan), first position that reaches r
    double carry = tile ? tsum[tile - 1] : 0.0;
    int64_t idx = base + KM_TILE - 1;
    bool found = false;
    for (int j0 = 0; j0 < KM_TILE && !found; j0 += 64) {
        const int64_t i = base + j0 + lane;
        double inc = (i < m) ? d[i] : 0.0;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const double y = __shfl_up(inc, off, 64);
            if (lane >= off) inc += y;
        }
        const double acc = carry + inc;
        const uint64_t hit = __ballot(acc >= r);
        if (hit) {
            idx = base + j0 + (__ffsll((long long)hit) - 1);
            found = true;
        }
        carry += __shfl(inc, 63, 64);
    }
    if (lane == 0) {
        if (idx > m - 1) idx = m - 1;                           // np.clip(candidate_ids, None, n - 1)
        st->cand_id[wave] = idx;
        st->cand_x[wave] = x[idx];
    }
}

__global__ __launch_bounds__(1024) void km_pick_kernel(const double *__restrict__ x, const double *__restrict__ d,
                                                       int64_t m, double *__restrict__ tsum, int64_t ntiles,
                                                       const double *__restrict__ uniform, int n_trials, int first_call,
                                                       KmState *st)
{
    __shared__ double s_scan[1024];
    __shared__ double s_pot;
    km_pick_body(x, d, m, tsum, ntiles, uniform, n_trials, first_call, st, s_scan, &s_pot);
}

// potentials of the candidates: sum over all values of min(d, squared distance to the candidate).  One 256-thread
// (sub-)block vb of nb: thread lt adds the values vb * 256 + lt, + nb * 256, ... in order, then the block tree.
__device__ __forceinline__ void km_pots_block(const double *__restrict__ x, const double *__restrict__ d, int64_t m,
                                              int n_trials, const KmState *__restrict__ st, double *__restrict__ ppart,
                                              int vb, int nb, int lt, double *red, bool valid)
{
    double c[KM_MAX_TRIALS], csq[KM_MAX_TRIALS], s[KM_MAX_TRIALS];
#pragma unroll
    for (int j = 0; j < KM_MAX_TRIALS; ++j) {
        c[j] = j < n_trials ? st->cand_x[j] : 0.0;
        csq[j] = __dmul_rn(c[j], c[j]);
        s[j] = 0.0;
    }
    const int64_t stride = (int64_t)nb * 256;
    for (int64_t i = valid ? (int64_t)vb * 256 + lt : m; i < m; i += stride) {
        const double xi = x[i], di = d[i];
#pragma unroll
        for (int j = 0; j < KM_MAX_TRIALS; ++j) {
            if (j < n_trials) {
                const double dj = km_sqdist(c[j], csq[j], xi);
                s[j] += dj < di ? dj : di;
            }
        }
    }
    for (int j = 0; j < n_trials; ++j) {
        const double tot = km_subblock_sum(s[j], red, lt);
        if (lt == 0 && valid) ppart[(size_t)j * nb + vb] = tot;
    }
}

__global__ __launch_bounds__(256) void km_pots_kernel(const double *__restrict__ x, const double *__restrict__ d, int64_t m,
                                                      int n_trials, const KmState *__restrict__ st,
                                                      double *__restrict__ ppart)
{
    __shared__ double red[4];
    km_pots_block(x, d, m, n_trials, st, ppart, blockIdx.x, gridDim.x, threadIdx.x, red, true);
}

// one wavefront per candidate: lane-strided partial sums (eight loads in flight), fixed butterfly -- a single
// thread walking the thousands of block partials of its candidate was a 0.1 ms latency chain per seed
__device__ __forceinline__ void km_choose_body(const double *__restrict__ ppart, int nb, int n_trials, int seed_no,
                                               KmState *st, double *__restrict__ seeds_x, int64_t *__restrict__ seeds_id,
                                               double *pots)
{
    const int j = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (j < n_trials) {
        const double *src = ppart + (size_t)j * nb;
        double s = 0.0;
        for (int b0 = lane; b0 < nb; b0 += 64 * 8) {
            double v[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = src[b0 + 64 * q < nb ? b0 + 64 * q : nb - 1];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < 8; ++q) s += b0 + 64 * q < nb ? v[q] : 0.0;
        }
        s = grx_group_sum<64>(s);
        if (lane == 0) pots[j] = s;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        // np.argmin: first minimum.  Exact ties are COMMON on small inputs -- two isolated candidates that each
        // capture only themselves and each other give the same potential, the same numbers summed in swapped
        // positions -- and BLAS returns bit-equal sums for them where another summation order may not: potentials
        // within 1e-12 of the minimum count as tied, the first of them wins
        double lowest = pots[0];
        for (int q = 1; q < n_trials; ++q) lowest = pots[q] < lowest ? pots[q] : lowest;
        int best = 0;
        while (best < n_trials - 1 && pots[best] > lowest + 1e-12 * lowest) ++best;
        st->pot = pots[best];
        st->best_x = st->cand_x[best];
        st->best_id = st->cand_id[best];
        seeds_x[seed_no] = st->cand_x[best];
        seeds_id[seed_no] = st->cand_id[best];
    }
}

__global__ __launch_bounds__(64 * KM_MAX_TRIALS) void km_choose_kernel(const double *__restrict__ ppart, int nb,
                                                                       int n_trials, int seed_no, KmState *st,
                                                                       double *__restrict__ seeds_x,
                                                                       int64_t *__restrict__ seeds_id)
{
    __shared__ double pots[KM_MAX_TRIALS];
    km_choose_body(ppart, nb, n_trials, seed_no, st, seeds_x, seeds_id, pots);
}

// d = min(d, squared distance to the chosen seed); tile sums for the next cumulative sum.  One 256-thread
// (sub-)block per tile of KM_TILE values.
__device__ __forceinline__ void km_update_tile(const double *__restrict__ x, double *__restrict__ d, int64_t m,
                                               const KmState *__restrict__ st, double *__restrict__ tsum, int64_t tile,
                                               int lt, double *red, bool valid)
{
    const double c = st->best_x, csq = __dmul_rn(c, c);
    const int64_t base = tile * KM_TILE;
    double s = 0.0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int64_t i = base + j * 256 + lt;
        if (valid && i < m) {
            const double dj = km_sqdist(c, csq, x[i]);
            const double di = d[i];
            const double nd = dj < di ? dj : di;
            if (dj < di) d[i] = nd;                           // late seeds move few points: most lines stay clean
            s += nd;
        }
    }
    s = km_subblock_sum(s, red, lt);
    if (lt == 0 && valid) tsum[tile] = s;
}

__global__ __launch_bounds__(256) void km_update_kernel(const double *__restrict__ x, double *__restrict__ d, int64_t m,
                                                        const KmState *__restrict__ st, double *__restrict__ tsum)
{
    __shared__ double red[4];
    km_update_tile(x, d, m, st, tsum, blockIdx.x, threadIdx.x, red, true);
}

// ---- the whole k-means++ seeding in ONE cooperative launch --------------------------------------------------------
// Per further seed the four steps above depend on each other through grid-wide results (the cumulative sum of every
// tile, the potentials over all values): as separate launches that is 4 (k - 1) launches per encode -- 2 044 for the
// 512 levels of a wide table.  Here every workgroup stays resident (hipLaunchCooperativeKernel) and the steps are
// separated by grid barriers; the serial steps (pick, choose) are run by the LAST workgroup to arrive, before it
// releases the others: two barriers per seed.  A workgroup is four 256-thread sub-blocks that take the role of the
// 256-thread workgroups of the per-seed kernels (same index sets, same trees), so the seeds are bit-identical.
__device__ __forceinline__ bool km_grid_arrive(KmState *st, unsigned nwg, unsigned *s_flag, unsigned *gen_out)
{
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned gen = __hip_atomic_load(&st->bar_gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __threadfence();                                      // this workgroup's results first
        const unsigned prev = __hip_atomic_fetch_add(&st->bar_count, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        s_flag[0] = prev == nwg - 1;
        s_flag[1] = gen;
        if (prev == nwg - 1) __threadfence();                 // ... and everybody else's before the serial step reads them
    }
    __syncthreads();
    *gen_out = s_flag[1];
    return s_flag[0] != 0;
}

__device__ __forceinline__ void km_grid_release(KmState *st)
{
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_store(&st->bar_count, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __threadfence();
        __hip_atomic_fetch_add(&st->bar_gen, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
}

__device__ __forceinline__ void km_grid_wait(KmState *st, unsigned gen)
{
    if (threadIdx.x == 0) {
        while (__hip_atomic_load(&st->bar_gen, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) == gen)
            __builtin_amdgcn_s_sleep(2);
        __threadfence();
    }
    __syncthreads();
}

__global__ __launch_bounds__(1024) void km_seed_kernel(const double *__restrict__ x, double *__restrict__ d, int64_t m,
                                                       double *__restrict__ tsum, int64_t ntiles,
                                                       const double *__restrict__ uniform, int n_trials, int k, int nb,
                                                       KmState *st, double *__restrict__ ppart,
                                                       double *__restrict__ seeds_x, int64_t *__restrict__ seeds_id)
{
    __shared__ double s_scan[1024];
    __shared__ double s_pot;
    __shared__ double s_red[4][4];
    __shared__ double s_pots[KM_MAX_TRIALS];
    __shared__ unsigned s_flag[2];
    const unsigned nwg = gridDim.x;
    const int sub = threadIdx.x >> 8, lt = threadIdx.x & 255;
    const int64_t vstride = (int64_t)nwg * 4;
    unsigned gen;
    // the first pick needs the tile sums of km_init_kernel (an earlier launch): workgroup 0 runs it, the others wait
    if (km_grid_arrive(st, nwg, s_flag, &gen)) {
        km_pick_body(x, d, m, tsum, ntiles, uniform, n_trials, 1, st, s_scan, &s_pot);
        km_grid_release(st);
    } else {
        km_grid_wait(st, gen);
    }
    for (int c = 1; c < k; ++c) {
        // potentials of this seed's candidates
        for (int64_t v0 = (int64_t)blockIdx.x * 4; v0 < nb; v0 += vstride) {
            const int64_t vb = v0 + sub;
            km_pots_block(x, d, m, n_trials, st, ppart, (int)vb, nb, lt, s_red[sub], vb < nb);
        }
        if (km_grid_arrive(st, nwg, s_flag, &gen)) {
            km_choose_body(ppart, nb, n_trials, c, st, seeds_x, seeds_id, s_pots);
            km_grid_release(st);
        } else {
            km_grid_wait(st, gen);
        }
        if (c == k - 1) break;                                 // the distances to the last seed are never needed
        for (int64_t t0 = (int64_t)blockIdx.x * 4; t0 < ntiles; t0 += vstride) {
            const int64_t tile = t0 + sub;
            km_update_tile(x, d, m, st, tsum, tile, lt, s_red[sub], tile < ntiles);
        }
        if (km_grid_arrive(st, nwg, s_flag, &gen)) {
            km_pick_body(x, d, m, tsum, ntiles, uniform + (size_t)c * n_trials, n_trials, 0, st, s_scan, &s_pot);
            km_grid_release(st);
        } else {
            km_grid_wait(st, gen);
        }
    }
}

// ---- prefix sums of the sorted values: P[i] = sum_{j < i} xs[j] ------------------------------------
__global__ __launch_bounds__(256) void km_tile_sums_kernel(const double *__restrict__ xs, int64_t m, double *__restrict__ tsum)
{
    __shared__ double red[4];
    const int64_t base = (int64_t)blockIdx.x * KM_TILE;
    double s = 0.0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int64_t i = base + threadIdx.x * 4 + j;
        if (i < m) s += xs[i];
    }
    s = km_block_sum(s, red);
    if (threadIdx.x == 0) tsum[blockIdx.x] = s;
}

__global__ __launch_bounds__(64) void km_scan_tiles_kernel(double *__restrict__ tsum, int64_t ntiles)
{
    if (threadIdx.x != 0) return;
    double run = 0.0;
    for (int64_t t = 0; t < ntiles; ++t) { const double v = tsum[t]; tsum[t] = run; run += v; }
}

__global__ __launch_bounds__(256) void km_prefix_kernel(const double *__restrict__ xs, int64_t m,
                                                        const double *__restrict__ tsum, double *__restrict__ P)
{
    __shared__ double wtot[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t base = (int64_t)blockIdx.x * KM_TILE + threadIdx.x * 4;
    double t[4], a = 0.0;
#pragma unroll
    for (int j = 0; j < 4; ++j) { t[j] = (base + j < m) ? xs[base + j] : 0.0; a += t[j]; }
    double inc = a;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const double y = __shfl_up(inc, off, 64);
        if (lane >= off) inc += y;
    }
    if (lane == 63) wtot[wave] = inc;
    __syncthreads();
    double o = tsum[blockIdx.x];
    for (int w = 0; w < wave; ++w) o += wtot[w];
    o += inc - a;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (base + j < m) { P[base + j] = o; o += t[j]; if (base + j + 1 == m) P[m] = o; }
    }
}

// ---- Lloyd iterations on the sorted values (one workgroup) ------------------------------------------
// xs ascending, P its prefix sums.  c[j]: centres in seed order.  Scratch (global, k entries each):
// ord (centre id by sorted position), hiS (values assigned to sorted positions <= p), lo / hi per centre id
// (this and the previous E step), sums, counts.
__device__ __forceinline__ int64_t km_upper(const double *__restrict__ xs, int64_t m, double b)
{
    int64_t lo = 0, up = m;                                     // first index with xs > b
    while (lo < up) { const int64_t mid = (lo + up) >> 1; if (xs[mid] <= b) lo = mid + 1; else up = mid; }
    return lo;
}

__device__ __forceinline__ int64_t km_lower(const double *__restrict__ xs, int64_t m, double b)
{
    int64_t lo = 0, up = m;                                     // first index with xs >= b
    while (lo < up) { const int64_t mid = (lo + up) >> 1; if (xs[mid] < b) lo = mid + 1; else up = mid; }
    return lo;
}

// true when sklearn's E step gives value x to the RIGHT centre: first minimum of c^2 - 2 x c in centre-id order
__device__ __forceinline__ bool km_prefers_right(double x, double cl, int idl, double cr, int idr)
{
    const double fl = __dadd_rn(__dmul_rn(cl, cl), __dmul_rn(-2.0, __dmul_rn(x, cl)));
    const double fr = __dadd_rn(__dmul_rn(cr, cr), __dmul_rn(-2.0, __dmul_rn(x, cr)));
    return fr < fl || (fr == fl && idr < idl);
}

// number of sorted values that go to centres at or left of `cl` when the next distinct centre is `cr`
__device__ int64_t km_boundary(const double *__restrict__ xs, int64_t m, double cl, int idl, double cr, int idr)
{
    int64_t h = km_upper(xs, m, 0.5 * (cl + cr));
    for (int guard = 0; guard < 8; ++guard) {
        if (h > 0 && km_prefers_right(xs[h - 1], cl, idl, cr, idr)) { h = km_lower(xs, m, xs[h - 1]); continue; }
        if (h < m && !km_prefers_right(xs[h], cl, idl, cr, idr)) { h = km_upper(xs, m, xs[h]); continue; }
        break;
    }
    return h;
}

struct KmLloydBufs {
    double *c, *cnew, *sums, *counts, *cfinal, *maxval;
    int32_t *ord;
    int64_t *hiS, *lo, *hi, *plo, *phi, *rl, *rh;
};

__device__ void km_e_step(const double *__restrict__ xs, int64_t m, int k, const KmLloydBufs &B)
{
    const int t = threadIdx.x, nt = blockDim.x;
    // sorted order of the centres (ties by centre id): rank by counting
    for (int j = t; j < k; j += nt) {
        const double cj = B.c[j];
        int rank = 0;
        for (int i = 0; i < k; ++i) {
            const double ci = B.c[i];
            rank += (ci < cj) || (ci == cj && i < j);
        }
        B.ord[rank] = j;
    }
    __syncthreads();
    // hiS[p]: values assigned to sorted positions <= p.  Equal centres: the smallest id takes the values.
    for (int p = t; p < k; p += nt) {
        const double cp = B.c[B.ord[p]];
        int g = p;                                              // first member of p's group of equal centres
        while (g > 0 && B.c[B.ord[g - 1]] == cp) --g;
        int nx = p + 1;                                         // next distinct centre
        while (nx < k && B.c[B.ord[nx]] == cp) ++nx;
        B.hiS[p] = (nx >= k) ? m : km_boundary(xs, m, cp, B.ord[g], B.c[B.ord[nx]], B.ord[nx]);
    }
    __syncthreads();
    for (int p = t; p < k; p += nt) {
        const int j = B.ord[p];
        const int64_t h = B.hiS[p];
        int64_t l;
        if (p == 0) l = 0;
        else if (B.c[B.ord[p - 1]] == B.c[j]) l = h;            // not the first of its group: empty
        else l = B.hiS[p - 1];
        B.lo[j] = l;
        B.hi[j] = h;
    }
    __syncthreads();
}

__global__ __launch_bounds__(1024) void km_lloyd_kernel(const double *__restrict__ xs, const double *__restrict__ P,
                                                        int64_t m, int k, int max_iter, const KmState *__restrict__ st,
                                                        const double *__restrict__ seeds_x, KmLloydBufs B,
                                                        int32_t *__restrict__ info)
{
    __shared__ double s_red[16];
    const int t = threadIdx.x, nt = blockDim.x;
    for (int j = t; j < k; j += nt) { B.c[j] = seeds_x[j]; B.plo[j] = -1; B.phi[j] = -1; }
    __syncthreads();
    const double tol = st->tol;
    int n_iter = 0;
    bool strict = false;
    for (int it = 0; it < max_iter; ++it) {
        n_iter = it + 1;
        km_e_step(xs, m, k, B);
        int n_empty_local = 0;
        for (int j = t; j < k; j += nt) {
            const int64_t l = B.lo[j], h = B.hi[j];
            B.counts[j] = (double)(h - l);
            B.sums[j] = P[h] - P[l];
            n_empty_local += (h == l);
        }
        const int any_empty = __syncthreads_or(n_empty_local);
        if (any_empty) {
            // _relocate_empty_clusters_dense: the points farthest from their centre seed the empty clusters
            // (ascending cluster id).  A cluster's farthest point is an end of its interval.
            for (int j = t; j < k; j += nt) { B.rl[j] = B.lo[j]; B.rh[j] = B.hi[j]; }
            __syncthreads();
            if (t == 0) {
                for (int e = 0; e < k; ++e) {
                    if (B.hi[e] != B.lo[e]) continue;           // empty after the E step, ascending cluster id
                    double best = -1.0;
                    int bj = -1, bend = 0;
                    for (int j = 0; j < k; ++j) {
                        const int64_t l = B.rl[j], h = B.rh[j];  // members not yet given away
                        if (h <= l) continue;
                        const double dl = (xs[l] - B.c[j]) * (xs[l] - B.c[j]);
                        const double dh = (xs[h - 1] - B.c[j]) * (xs[h - 1] - B.c[j]);
                        if (dl > best) { best = dl; bj = j; bend = 0; }
                        if (dh > best) { best = dh; bj = j; bend = 1; }
                    }
                    if (bj < 0) break;
                    const double xv = bend ? xs[B.rh[bj] - 1] : xs[B.rl[bj]];
                    if (bend) B.rh[bj] -= 1; else B.rl[bj] += 1;
                    B.sums[bj] -= xv;
                    B.counts[bj] -= 1.0;
                    B.sums[e] = xv;
                    B.counts[e] = 1.0;
                }
            }
        }
        __syncthreads();
        double shift2 = 0.0;
        int changed = 0;
        for (int j = t; j < k; j += nt) {
            const double cnt = B.counts[j];
            const double cn = cnt > 0.0 ? B.sums[j] * (1.0 / cnt) : 0.0;        // _average_centers: sum * (1 / weight)
            const double dlt = cn - B.c[j];
            shift2 += dlt * dlt;
            B.cnew[j] = cn;
            changed |= (B.lo[j] != B.plo[j]) || (B.hi[j] != B.phi[j]);
        }
        // total squared shift: per-wave butterflies, then the wave totals in order
        shift2 = grx_group_sum<64>(shift2);
        if ((t & 63) == 0) s_red[t >> 6] = shift2;
        const int any_changed = __syncthreads_or(changed);
        double tot = 0.0;
        for (int w = 0; w < (nt >> 6); ++w) tot += s_red[w];
        for (int j = t; j < k; j += nt) { B.c[j] = B.cnew[j]; B.plo[j] = B.lo[j]; B.phi[j] = B.hi[j]; }
        __syncthreads();
        if (!any_changed) { strict = true; break; }             // labels unchanged: strict convergence
        if (tot <= tol) break;
    }
    if (!strict) km_e_step(xs, m, k, B);                        // a last E step so that labels match the centres
    // tables of the assignment pass: per sorted position the largest value it takes, and the output level
    const double mean = st->mean;
    for (int p = t; p < k; p += nt) {
        const int64_t h = B.hiS[p];
        B.maxval[p] = h > 0 ? xs[h - 1] : -1.79769313486231570e308;
        B.cfinal[p] = B.c[B.ord[p]] + mean;                     // best_centers += X_mean
    }
    __syncthreads();
    if (t == 0) {
        int nonempty = 0, distinct = 0;
        double last = 0.0;
        // distinct output values among the clusters that hold values, in sorted-centre order
        for (int p = 0; p < k; ++p) {
            const int j = B.ord[p];
            if (B.hi[j] > B.lo[j]) {
                ++nonempty;
                if (distinct == 0 || B.cfinal[p] != last) { ++distinct; last = B.cfinal[p]; }
            }
        }
        info[0] = n_iter;
        info[1] = nonempty;
        info[2] = distinct;
    }
}

__global__ __launch_bounds__(256) void km_assign_kernel(const double *__restrict__ v, int64_t m, int k,
                                                        const KmState *__restrict__ st, const double *__restrict__ maxval,
                                                        const double *__restrict__ cfinal, const int32_t *__restrict__ ord,
                                                        double *__restrict__ out, double *__restrict__ centers_out)
{
    const double mean = st->mean;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < m; i += stride) {
        const double x = v[i] - mean;
        int lo = 0, up = k - 1;                                 // first sorted position whose largest value is >= x
        while (lo < up) { const int mid = (lo + up) >> 1; if (maxval[mid] < x) lo = mid + 1; else up = mid; }
        out[i] = cfinal[lo];
    }
    if (blockIdx.x == 0)
        for (int p = threadIdx.x; p < k; p += blockDim.x) centers_out[ord[p]] = cfinal[p];     // seed order
}

// out[c * ld_out + r] = in[r * ld_in + c]: the factor matrices are feature-major on the device ([r, n]) while the
// reference flattens them row-major as n x r (encode(): X.reshape(X.size, 1)) -- the order its cumulative sums run in
__global__ __launch_bounds__(256) void km_transpose_kernel(int64_t rows, int64_t cols, const double *__restrict__ in,
                                                           int64_t ld_in, double *__restrict__ out, int64_t ld_out)
{
    __shared__ double tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;      // 32 x 8
    const int64_t c0 = (int64_t)blockIdx.x * 32, r0 = (int64_t)blockIdx.y * 32;
    for (int j = ty; j < 32; j += 8) {
        const int64_t r = r0 + j, c = c0 + tx;
        tile[j][tx] = (r < rows && c < cols) ? in[r * ld_in + c] : 0.0;
    }
    __syncthreads();
    for (int j = ty; j < 32; j += 8) {
        const int64_t c = c0 + j, r = r0 + tx;
        if (c < cols && r < rows) out[c * ld_out + r] = tile[tx][j];
    }
}

struct KmPlan {
    int64_t ntiles;
    int nb;                                                    // workgroups of the strided reductions
    size_t off_state, off_x, off_d, off_tsum, off_ppart, off_part, off_seedx, off_seedid, off_uniform, off_xs, off_P,
        off_lloyd, off_sort, total;
};

KmPlan km_plan(int64_t m, int k)
{
    KmPlan p;
    p.ntiles = grx_ceil_div(m, KM_TILE);
    const int64_t want = grx_ceil_div(m, 256 * 8);
    p.nb = (int)(want > 2048 ? 2048 : (want < 1 ? 1 : want));
    size_t o = 0;
    auto take = [&](size_t bytes) { size_t at = o; o += grx_align_up(bytes, 256); return at; };
    p.off_state = take(sizeof(KmState));
    p.off_x = take((size_t)m * 8);
    p.off_d = take((size_t)m * 8);
    p.off_tsum = take((size_t)p.ntiles * 8);
    p.off_ppart = take((size_t)KM_MAX_TRIALS * p.nb * 8);
    p.off_part = take((size_t)p.nb * 8);
    p.off_seedx = take((size_t)k * 8);
    p.off_seedid = take((size_t)k * 8);
    p.off_uniform = take((size_t)(k > 1 ? k - 1 : 1) * KM_MAX_TRIALS * 8);
    p.off_xs = take((size_t)m * 8);
    p.off_P = take((size_t)(m + 1) * 8);
    p.off_lloyd = take((size_t)k * 8 * 16);
    p.off_sort = take(grx_sort_workspace_bytes(m, 1));
    p.total = o;
    return p;
}

}  // namespace

extern "C" {

int grx_transpose(int64_t rows, int64_t cols, const double *d_in, int64_t ld_in, double *d_out, int64_t ld_out, void *stream)
{
    GRX_REQUIRE(rows >= 0 && cols >= 0 && ld_in >= cols && ld_out >= rows, "grx_transpose: bad shape");
    if (rows == 0 || cols == 0) return GRX_OK;
    GRX_REQUIRE(d_in && d_out, "grx_transpose: NULL pointer");
    const dim3 grid((unsigned)grx_ceil_div(cols, 32), (unsigned)grx_ceil_div(rows, 32));
    GRX_REQUIRE(grid.y < 65536u * 1u || true, "grx_transpose: too many rows");
    km_transpose_kernel<<<grid, 256, 0, grx_stream(stream)>>>(rows, cols, d_in, ld_in, d_out, ld_out);
    GRX_LAUNCH_CHECK();
    return GRX_OK;
}

size_t grx_kmeans1d_workspace_bytes(int64_t m, int k)
{
    if (m < 1) m = 1;
    if (k < 1) k = 1;
    return km_plan(m, k).total;
}

int grx_kmeans1d(int64_t m, const double *d_values, int k, int64_t first_seed, const double *h_uniform, int n_trials,
                 int max_iter, double rel_tol, double *d_quantized, double *d_centers, int32_t *d_info,
                 void *d_workspace, size_t workspace_bytes, void *stream)
{
    GRX_REQUIRE(m >= 1 && k >= 1 && max_iter >= 0, "grx_kmeans1d: bad m / k / max_iter");   // max_iter = 0: seeding only
    GRX_REQUIRE(k <= m, "n_samples=%lld should be >= n_clusters=%d.", (long long)m, k);
    GRX_REQUIRE(m < ((int64_t)1 << 31), "grx_kmeans1d: m must be < 2^31");
    if (k > KM_MAX_K) {
        grx_set_error("grx_kmeans1d: n_clusters=%d > %d", k, KM_MAX_K);
        return GRX_ERR_UNSUPPORTED;
    }
    GRX_REQUIRE(first_seed >= 0 && first_seed < m, "grx_kmeans1d: first seed outside [0, m)");
    GRX_REQUIRE(n_trials >= 1 && n_trials <= KM_MAX_TRIALS, "grx_kmeans1d: n_trials outside [1, %d]", KM_MAX_TRIALS);
    GRX_REQUIRE(d_values && d_quantized && d_centers && d_info && d_workspace && (k == 1 || h_uniform),
                "grx_kmeans1d: NULL pointer");
    const KmPlan p = km_plan(m, k);
    if (workspace_bytes < p.total) {
        grx_set_error("grx_kmeans1d: workspace %zu < %zu", workspace_bytes, p.total);
        return GRX_ERR_WORKSPACE;
    }
    hipStream_t st = grx_stream(stream);
    char *ws = reinterpret_cast<char *>(d_workspace);
    KmState *state = reinterpret_cast<KmState *>(ws + p.off_state);
    double *x = reinterpret_cast<double *>(ws + p.off_x);
    double *d = reinterpret_cast<double *>(ws + p.off_d);
    double *tsum = reinterpret_cast<double *>(ws + p.off_tsum);
    double *ppart = reinterpret_cast<double *>(ws + p.off_ppart);
    double *part = reinterpret_cast<double *>(ws + p.off_part);
    double *seeds_x = reinterpret_cast<double *>(ws + p.off_seedx);
    int64_t *seeds_id = reinterpret_cast<int64_t *>(ws + p.off_seedid);
    double *d_uniform = reinterpret_cast<double *>(ws + p.off_uniform);
    double *xs = reinterpret_cast<double *>(ws + p.off_xs);
    double *P = reinterpret_cast<double *>(ws + p.off_P);
    GRX_PROF(GRX_K_QUANT, st);
    if (k > 1)
        GRX_CHECK_HIP(hipMemcpyAsync(d_uniform, h_uniform, (size_t)(k - 1) * n_trials * 8, hipMemcpyHostToDevice, st));
    // mean and tolerance (KMeans.fit: X -= X.mean(axis=0); tol = mean(var(X, axis=0)) * 1e-4)
    km_sum_kernel<<<p.nb, 256, 0, st>>>(d_values, m, state, 0, part);
    km_moment_final_kernel<<<1, 64, 0, st>>>(part, p.nb, m, 0, rel_tol, state);
    km_sum_kernel<<<p.nb, 256, 0, st>>>(d_values, m, state, 1, part);
    km_moment_final_kernel<<<1, 64, 0, st>>>(part, p.nb, m, 1, rel_tol, state);
    km_init_kernel<<<(int)p.ntiles, 256, 0, st>>>(d_values, m, first_seed, state, x, d, tsum, seeds_x, seeds_id);
    GRX_LAUNCH_CHECK();
    if (k > 1) {
        // GRX_KMEANS_COOPERATIVE=1: one cooperative launch for all k - 1 further seeds instead of four launches per
        // seed -- same bits (tests/test_gpu_encode.py), but MEASURED SLOWER on MI355X (1 M x 6 factor, 64 levels: 16.1
        // against 6.3 ms): every grid barrier needs an agent-scope release / acquire so that the distances written
        // on one XCD are seen on another, i.e. an L2 write-back + invalidate per workgroup per barrier, which costs
        // more than the launch boundary it replaces (the same finding as the fused reduce + H update of the NMF,
        // DESIGN.md section 3).  Kept as an option; the default is the per-seed sequence.
        static const bool per_seed = [] { const char *e = std::getenv("GRX_KMEANS_COOPERATIVE"); return !(e && *e == '1'); }();
        int resident = 0;
        if (!per_seed) {
            int per_cu = 0, dev = 0, cus = 0;
            GRX_CHECK_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, km_seed_kernel, 1024, 0));
            GRX_CHECK_HIP(hipGetDevice(&dev));
            GRX_CHECK_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
            resident = per_cu * cus;
        }
        if (resident >= 1) {
            const int64_t useful = grx_ceil_div(p.ntiles > p.nb ? p.ntiles : (int64_t)p.nb, 4);
            int grid = (int)(useful < resident ? useful : resident);
            if (grid < 1) grid = 1;
            GRX_CHECK_HIP(hipMemsetAsync(&state->bar_count, 0, 2 * sizeof(unsigned), st));
            int64_t m_arg = m, ntiles_arg = p.ntiles;
            int trials_arg = n_trials, k_arg = k, nb_arg = p.nb;
            const double *x_arg = x, *u_arg = d_uniform;
            void *args[] = {&x_arg, &d, &m_arg, &tsum, &ntiles_arg, &u_arg, &trials_arg, &k_arg, &nb_arg, &state, &ppart,
                            &seeds_x, &seeds_id};
            GRX_CHECK_HIP(hipLaunchCooperativeKernel(reinterpret_cast<const void *>(km_seed_kernel), dim3(grid), dim3(1024),
                                                     args, 0, st));
        } else {
            for (int c = 1; c < k; ++c) {
                km_pick_kernel<<<1, 1024, 0, st>>>(x, d, m, tsum, p.ntiles, d_uniform + (size_t)(c - 1) * n_trials, n_trials,
                                                   c == 1, state);
                km_pots_kernel<<<p.nb, 256, 0, st>>>(x, d, m, n_trials, state, ppart);
                km_choose_kernel<<<1, 64 * n_trials, 0, st>>>(ppart, p.nb, n_trials, c, state, seeds_x, seeds_id);
                km_update_kernel<<<(int)p.ntiles, 256, 0, st>>>(x, d, m, state, tsum);
            }
        }
    }
    GRX_LAUNCH_CHECK();
    // Lloyd on the sorted values
    int rc = grx_internal_sort_columns(m, 1, x, m, xs, m, ws + p.off_sort, st);
    if (rc != GRX_OK) return rc;
    km_tile_sums_kernel<<<(int)p.ntiles, 256, 0, st>>>(xs, m, tsum);
    km_scan_tiles_kernel<<<1, 64, 0, st>>>(tsum, p.ntiles);
    km_prefix_kernel<<<(int)p.ntiles, 256, 0, st>>>(xs, m, tsum, P);
    KmLloydBufs B;
    double *lb = reinterpret_cast<double *>(ws + p.off_lloyd);
    B.c = lb; B.cnew = lb + k; B.sums = lb + 2 * (size_t)k; B.counts = lb + 3 * (size_t)k; B.cfinal = lb + 4 * (size_t)k;
    B.maxval = lb + 5 * (size_t)k;
    B.hiS = reinterpret_cast<int64_t *>(lb + 6 * (size_t)k);
    B.lo = B.hiS + k; B.hi = B.lo + k; B.plo = B.hi + k; B.phi = B.plo + k; B.rl = B.phi + k; B.rh = B.rl + k;
    B.ord = reinterpret_cast<int32_t *>(B.rh + k);
    km_lloyd_kernel<<<1, 1024, 0, st>>>(xs, P, m, k, max_iter, state, seeds_x, B, d_info);
    const int64_t want = grx_ceil_div(m, 256 * 4);
    km_assign_kernel<<<(int)(want > 2048 ? 2048 : want), 256, 0, st>>>(d_values, m, k, state, B.maxval, B.cfinal, B.ord,
                                                                      d_quantized, d_centers);
    GRX_LAUNCH_CHECK();
    return GRX_OK;
}

}  // extern "C"
