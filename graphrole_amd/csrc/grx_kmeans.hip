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
//   * k-means++: cumulative sum of the closest distances IN INDEX ORDER -> searchsorted of uniform * potential ->
//     the candidate with the smallest potential wins;
//   * Lloyd on the sorted values with prefix sums (labels are intervals): first-minimum E step decided with
//     sklearn's own expression c^2 - 2 x c at the interval ends, centres = sum * (1 / count), relocation of
//     empty clusters to the farthest points, stop on unchanged labels or total squared centre shift <= tol,
//     a last E step when the labels had not settled.
//
// The seeding in one dimension (round 6).  sklearn's formulation touches every value for every seed (distances to
// the candidates, minimum with the closest distance, potentials): O(m k) -- 0.6 TB of traffic for the 30 M entries
// and 512 levels of a wide table.  On the line a candidate c between the seeds s_L < c < s_R can only lower the
// closest distance of values between the two midpoints, a CONTIGUOUS RANGE of the sorted values.  So:
//   * the values are sorted once together with their indices (perm: sorted position -> index, rank: its inverse);
//     the closest distances live in SORTED order (ds);
//   * the range is a SUPERSET of the values the full pass would change: midpoints widened by a bound on the
//     rounding error of the two fp distances (see km_pick_tail), and inside it every value is tested with the very
//     expression the full pass uses -- GRX_KMEANS_FULL_RANGE=1 makes every range [0, m) and must give the same bits
//     (tests/test_gpu_encode.py);
//   * the sums a candidate INDEX is drawn from are EXACT: a distance is split into three fixed-point limbs (units
//     2^(E-32), 2^(E-64), 2^(E-96), E from the largest possible distance; the rest, < 2^-96 of it, is dropped -- a
//     function of the value alone), limb sums are integers, integer addition is associative: the sums of the closest
//     distances per block of 2^s INDICES (at most 1024 blocks) are kept by integer atomics from the range update, in
//     any order, and the cumulative sum sklearn searches is their prefix; inside the block a trial's r falls in, the
//     distances are gathered through rank.  Nothing depends on the order of a reduction, so grids and chunking are
//     free to change and every run gives the same bits;
//   * the candidates' potentials only feed an argmin: potential(c) = potential - sum over c's range of
//     (d - min(d, dist(c, x))) comes from fp64 sums over blocks of KM_CHUNK SORTED positions (sum of d per block,
//     rewritten by every update, and static moments of the block's values: km_pick_tail, D) plus the two end blocks
//     value by value -- no pass over the ranges.  GRX_KMEANS_GAIN_PASS=1: the pass, in exact integers (km_gain_kernel);
//   * per seed one launch for most seeds, no host round trip: km_prep_kernel (winner of the previous seed, prefix of
//     the block sums, the n_trials candidates -- sixteen workgroups per trial share the gather of a block and hand
//     over through tagged words --, their neighbours among the seeds, their ranges, their gains, and once every
//     trial is recorded the update of the winner's range by all workgroups); the first seeds, whose ranges are long,
//     get their update from km_update_kernel.  m <= 4096: the whole seeding in one workgroup (km_seed_small_kernel).
// What cannot be bit-identical to sklearn: it sums in floating point (cumsum, BLAS), relative 1e-13, so a uniform
// draw that lands within that distance of a boundary of the cumulative sum picks a neighbouring index (probability
// ~1e-6 per draw at 6 M values), and tied candidate potentials are recognised with a 1e-12 tolerance (km_best_of).
// Centres agree with sklearn to ~1e-12 otherwise.
#include "grx_common.h"

#include <cmath>
#include <cstdlib>

// Built with -ffp-contract=off (csrc/Makefile): every product and sum below is rounded on its own, as numpy rounds
// them.  hipcc's default (-ffp-contract=fast) fuses a * b + c differently from one kernel to the next -- HIP's
// __dmul_rn / __dadd_rn are plain * and + and do not prevent it -- and the seeding needs the SAME distance bits
// wherever a distance is evaluated (the consistency check of km_prep_kernel caught exactly that).

int grx_internal_sort_pairs(int64_t n, const double *col, double *out, uint32_t *perm, void *workspace, hipStream_t st);
size_t grx_internal_sort_pairs_workspace_bytes(int64_t n);

namespace {

typedef __int128 i128;
typedef unsigned long long u64;
typedef long long i64;

constexpr int KM_TILE = 1024;                 // values per tile of the prefix sums of the sorted values (256 threads x 4)
#ifndef KM_MAX_BLOCKS_V
#define KM_MAX_BLOCKS_V 1024
#endif
constexpr int KM_MAX_BLOCKS = KM_MAX_BLOCKS_V;           // index blocks of the cumulative sum (four per thread of the prep scan)
constexpr int KM_MIN_BLOCK_SHIFT = 6;
constexpr int KM_SUB = 16;                    // sub-blocks of an index block: one workgroup each in km_prep_kernel
constexpr int KM_MAX_TRIALS = 16;             // 2 + int(log(k)) <= 11 for k <= 8192
constexpr int KM_MAX_K = 8192;                // the E step ranks the centres by counting: O(k^2) per iteration
constexpr int KM_CHUNK = 2048;                // sorted positions per workgroup step of the range kernels (256 x 8)
constexpr int KM_RANGE_GRID = 512;            // workgroups of the gain kernel
constexpr int KM_UPDATE_GRID = 512;           // ... of the update kernel (each flushes its block sums once)
constexpr int KM_E_MIN = -900, KM_E_MAX = 960;
constexpr int KM_UPDATE_THREADS = 1024;  // km_update_kernel: four chunks in flight per workgroup
constexpr int KM_SPIN_LIMIT = 1 << 18;   // reads of a trial's sums before the pick gives up (a fraction of a second)
constexpr int KM_TOP2 = 1024;            // LDS level of the search over the sorted values
constexpr int KM_SEEDS_LDS = 2048;       // sorted seeds the pick keeps in LDS (more: searched in memory)
constexpr int KM_SMALL_M = 4096;              // at most this many values: the seeding runs in one workgroup

__device__ __forceinline__ double km_sqdist(double c, double csq, double x)
{
    // sklearn _euclidean_distances: -2 * (x . c), += |c|^2, += |x|^2, clipped at 0 -- every step rounded
    double d = __dmul_rn(-2.0, __dmul_rn(x, c));
    d = __dadd_rn(d, csq);
    d = __dadd_rn(d, __dmul_rn(x, x));
    return d > 0.0 ? d : 0.0;
}

struct KmLimb { double cA, cB, cC, sA, sB, sC; };

// per-seed records exist twice (index = seed number & 1): the prep workgroups of seed c + 1 read seed c's while they
// write their own
struct KmSeedRec {
    double cand_x[KM_MAX_TRIALS];
    i64 cand_id[KM_MAX_TRIALS], cand_lo[KM_MAX_TRIALS], cand_hi[KM_MAX_TRIALS];   // candidate index; range of sorted positions
    i64 gain[3][KM_MAX_TRIALS];  // limb sums of (d - min(d, dist to candidate)) over the candidate's range
    u64 pot_lo, pot_hi;          // potential before this seed = sum of the closest distances, in quanta of 2^(E-96)
    i64 sub[KM_MAX_TRIALS][KM_SUB][3];   // km_prep_kernel: limb sums of the sub-blocks of the block each trial's r falls in
    double gain_d[KM_MAX_TRIALS];        // gains from the sorted-block sums (km_pick_tail), when no gain pass runs
    u64 pub[KM_MAX_TRIALS][6];           // the trial's range, value and gain for the update inside the pick kernel: 32 or 48 bits
                                         // per word under the seed's number (bits 48..63)
};

struct KmState {                 // device scalars shared by the kernels of one run
    double mean, tol, vmin, vmax;
    double amax;                 // bound on |x - mean|
    double c0;                   // the first seed (centred)
    KmLimb limb;
    int faults, scale_e;
    KmSeedRec rec[2];
#ifdef KM_DBG_TIMING
    long long dbg[32];
#endif
};
#ifdef KM_DBG_TIMING
#define KM_T(i) do { if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0 && seed_no == KM_DBG_TIMING) st->dbg[i] = wall_clock64(); } while (0)
#define KM_TP(i) do { if (trial == 0 && threadIdx.x == 0 && st->dbg[0] != 0 && st->dbg[i] == 0) st->dbg[i] = wall_clock64(); } while (0)
#define KM_TG(i) do { if (blockIdx.x == 0 && threadIdx.x == 0 && seed_no == KM_DBG_TIMING) const_cast<KmState *>(st)->dbg[i] = wall_clock64(); } while (0)
#else
#define KM_T(i) do { } while (0)
#define KM_TP(i) do { } while (0)
#define KM_TG(i) do { } while (0)
#endif

// ---- exact sums: three limbs per distance ----------------------------------------------------------------------
// a = d rounded to a multiple of 2^(E-32), b = (d - a) rounded to 2^(E-64), c = (d - a - b) rounded to 2^(E-96);
// d - a and d - a - b are exact.  0 <= a / 2^(E-32) <= 2^32, |b|, |c| <= 2^31 units: sums of up to 2^20 limbs are
// exact in fp64, sums of up to 2^30 as int64.  value(d) := a + b + c.
__device__ __forceinline__ void km_split(const KmLimb &L, double d, double &a, double &b, double &c)
{
    a = __dadd_rn(__dadd_rn(d, L.cA), -L.cA);
    const double r1 = __dadd_rn(d, -a);
    b = __dadd_rn(__dadd_rn(r1, L.cB), -L.cB);
    const double r2 = __dadd_rn(r1, -b);
    c = __dadd_rn(__dadd_rn(r2, L.cC), -L.cC);
}

__device__ __forceinline__ i128 km_join(i64 a, i64 b, i64 c) { return ((i128)a << 64) + ((i128)b << 32) + (i128)c; }

__device__ __forceinline__ i128 km_quanta(const KmLimb &L, double d)
{
    double a, b, c;
    km_split(L, d, a, b, c);
    return km_join(__double2ll_rn(a * L.sA), __double2ll_rn(b * L.sB), __double2ll_rn(c * L.sC));
}

__device__ __forceinline__ i128 km_make128(u64 lo, u64 hi) { return (i128)(((unsigned __int128)hi << 64) | lo); }

__device__ __forceinline__ double km_to_double(i128 v)          // v >= 0
{
    return (double)(u64)(v >> 64) * 18446744073709551616.0 + (double)(u64)v;
}

__device__ __forceinline__ i128 km_ceil128(double r)            // smallest integer >= r, 0 <= r < 2^127
{
    if (!(r > 0.0)) return 0;
    const u64 bits = (u64)__double_as_longlong(r);
    const int e = (int)((bits >> 52) & 0x7FF);
    const u64 mant = (bits & ((1ull << 52) - 1)) | (e ? (1ull << 52) : 0ull);
    const int sh = (e ? e : 1) - 1075;                          // r = mant * 2^sh
    if (sh >= 0) return (i128)mant << sh;
    if (sh <= -64) return 1;
    const u64 q = mant >> (-sh), rem = mant & ((1ull << (-sh)) - 1);
    return (i128)(q + (rem ? 1 : 0));
}

__device__ __forceinline__ i128 km_shfl_up128(i128 v, int off)
{
    return km_make128(__shfl_up((u64)v, off, 64), __shfl_up((u64)(v >> 64), off, 64));
}

__device__ __forceinline__ i128 km_wave_scan128(i128 v, int lane)      // inclusive
{
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const i128 y = km_shfl_up128(v, off);
        if (lane >= off) v += y;
    }
    return v;
}

__device__ __forceinline__ i64 km_wave_sum_i64(i64 v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

__device__ __forceinline__ void km_atomic_add_i64(i64 *p, i64 v)
{
    if (v) atomicAdd(reinterpret_cast<u64 *>(p), (u64)v);
}

// fixed-shape workgroup sum (256 threads): wave butterflies, then the four wave totals in order
__device__ __forceinline__ double km_block_sum(double v, double *red)
{
    v = grx_group_sum<64>(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return ((red[0] + red[1]) + red[2]) + red[3];
}

// ---- moments ---------------------------------------------------------------------------------------
// part[b] = workgroup sum; pass 0 (square == 0) also part[nb + b] = min, part[2 nb + b] = max
__global__ __launch_bounds__(256) void km_sum_kernel(const double *__restrict__ v, int64_t m,
                                                     const KmState *__restrict__ st, int square,
                                                     double *__restrict__ part)
{
    __shared__ double red[4], rmin[4], rmax[4];
    const double shift = square ? st->mean : 0.0;
    double s = 0.0, lo = v[0], hi = v[0];
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < m; i += stride) {
        const double raw = v[i];
        const double x = raw - shift;
        s += square ? x * x : x;
        lo = raw < lo ? raw : lo;
        hi = raw > hi ? raw : hi;
    }
    s = km_block_sum(s, red);
    if (threadIdx.x == 0) part[blockIdx.x] = s;
    if (!square) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const double a = __shfl_xor(lo, off, 64), b = __shfl_xor(hi, off, 64);
            lo = a < lo ? a : lo;
            hi = b > hi ? b : hi;
        }
        if ((threadIdx.x & 63) == 0) { rmin[threadIdx.x >> 6] = lo; rmax[threadIdx.x >> 6] = hi; }
        __syncthreads();
        if (threadIdx.x == 0) {
            for (int w = 1; w < 4; ++w) { lo = rmin[w] < lo ? rmin[w] : lo; hi = rmax[w] > hi ? rmax[w] : hi; }
            part[gridDim.x + blockIdx.x] = lo;
            part[2 * gridDim.x + blockIdx.x] = hi;
        }
    }
}

// which == 0: mean, min, max.  which == 1: tolerance, and the scale of the exact sums from the first seed.
__global__ __launch_bounds__(256) void km_moment_final_kernel(const double *__restrict__ part, int nb, int64_t m, int which,
                                                              double rel_tol, const double *__restrict__ v, int64_t first,
                                                              KmState *st)
{
    __shared__ double red[4], rmin[4], rmax[4];
    double s = 0.0;
    for (int b = threadIdx.x; b < nb; b += 256) s += part[b];
    s = km_block_sum(s, red);
    if (which == 0) {
        double lo = part[nb], hi = part[2 * nb];
        for (int b = threadIdx.x; b < nb; b += 256) { lo = part[nb + b] < lo ? part[nb + b] : lo; hi = part[2 * nb + b] > hi ? part[2 * nb + b] : hi; }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const double a = __shfl_xor(lo, off, 64), b = __shfl_xor(hi, off, 64);
            lo = a < lo ? a : lo;
            hi = b > hi ? b : hi;
        }
        if ((threadIdx.x & 63) == 0) { rmin[threadIdx.x >> 6] = lo; rmax[threadIdx.x >> 6] = hi; }
        __syncthreads();
        if (threadIdx.x != 0) return;
        for (int w = 1; w < 4; ++w) { lo = rmin[w] < lo ? rmin[w] : lo; hi = rmax[w] > hi ? rmax[w] : hi; }
        st->mean = s / (double)m;
        st->vmin = lo;
        st->vmax = hi;
        return;
    }
    // (a sub-block sum is valid when it carries its seed's number, never 0)
    for (int i = threadIdx.x; i < (int)(2 * sizeof(KmSeedRec) / 8); i += 256) reinterpret_cast<u64 *>(&st->rec[0])[i] = 0;
    if (threadIdx.x != 0) return;
    st->tol = (s / (double)m) * rel_tol;                       // _tolerance: mean(var(X, axis=0)) * tol
    const double mean = st->mean;
    const double xl = st->vmin - mean, xh = st->vmax - mean, c0 = v[first] - mean;
    st->c0 = c0;
    st->amax = fabs(xl) > fabs(xh) ? fabs(xl) : fabs(xh);
    const double reach = fabs(xl - c0) > fabs(xh - c0) ? fabs(xl - c0) : fabs(xh - c0);
    const double bound = reach * reach;                        // no closest distance is ever larger (up to rounding: + 2 below)
    int E = (bound > 0.0 && bound < 1.0e300) ? ilogb(bound) + 2 : (bound > 0.0 ? KM_E_MAX + 1 : KM_E_MIN);
    int faults = 0;
    if (E < KM_E_MIN) E = KM_E_MIN;                            // (values below 1e-135: distances lose their low bits)
    if (E > KM_E_MAX) { E = KM_E_MAX; faults = 2; }            // values beyond 1e144: not representable
    st->scale_e = E;
    st->limb.cA = ldexp(1.5, 52 + E - 32);
    st->limb.cB = ldexp(1.5, 52 + E - 64);
    st->limb.cC = ldexp(1.5, 52 + E - 96);
    st->limb.sA = ldexp(1.0, 32 - E);
    st->limb.sB = ldexp(1.0, 64 - E);
    st->limb.sC = ldexp(1.0, 96 - E);
    st->faults = faults;
#ifdef KM_DBG_TIMING
    for (int j = 0; j < 32; ++j) st->dbg[j] = 0;
#endif
}

// index order: x = v - mean; per block of 2^block_shift indices the exact sum of the squared distances to the first
// seed (a workgroup of 1024 indices touches at most 17 blocks: collected in LDS, then added with integer atomics)
__global__ __launch_bounds__(256) void km_init_kernel(const double *__restrict__ v, int64_t m, int64_t first,
                                                      KmState *st, double *__restrict__ x, i64 *__restrict__ bacc,
                                                      int block_shift, double *__restrict__ seeds_x,
                                                      int64_t *__restrict__ seeds_id, double *__restrict__ sorted)
{
    __shared__ u64 s_acc[3][(KM_TILE >> KM_MIN_BLOCK_SHIFT) + 1];
    const double mean = st->mean;
    const KmLimb L = st->limb;
    const double c = st->c0, csq = __dmul_rn(c, c);
    const int64_t base = (int64_t)blockIdx.x * KM_TILE;
    const int64_t b0 = base >> block_shift;
    if (threadIdx.x < 3 * ((KM_TILE >> KM_MIN_BLOCK_SHIFT) + 1)) (&s_acc[0][0])[threadIdx.x] = 0;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < KM_TILE / 256; ++j) {
        const int64_t i = base + j * 256 + threadIdx.x;
        if (i < m) {
            const double xi = v[i] - mean;
            x[i] = xi;
            double a, b, cc;
            km_split(L, km_sqdist(c, csq, xi), a, b, cc);
            const int slot = (int)((i >> block_shift) - b0);
            atomicAdd(&s_acc[0][slot], (u64)__double2ll_rn(a * L.sA));
            atomicAdd(&s_acc[1][slot], (u64)__double2ll_rn(b * L.sB));
            atomicAdd(&s_acc[2][slot], (u64)__double2ll_rn(cc * L.sC));
        }
    }
    __syncthreads();
    if (threadIdx.x < 3 * ((KM_TILE >> KM_MIN_BLOCK_SHIFT) + 1)) {
        const int limb = threadIdx.x / ((KM_TILE >> KM_MIN_BLOCK_SHIFT) + 1), slot = threadIdx.x % ((KM_TILE >> KM_MIN_BLOCK_SHIFT) + 1);
        if (s_acc[limb][slot]) km_atomic_add_i64(bacc + 4 * (b0 + slot) + limb, (i64)s_acc[limb][slot]);
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        seeds_x[0] = c;
        seeds_id[0] = first;
        sorted[0] = c;
    }
}

// sorted order, one workgroup per block of KM_CHUNK positions: closest distances to the first seed, the inverse
// permutation, and what the gains are later computed from without a pass over the values (km_pick_tail): the block's sum
// of closest distances (kept current by km_update_kernel) and its static moments about a reference value mu inside the
// block -- sum(x - mu), sum((x - mu)^2) -- from which sum((x - c)^2) over the block follows for any c
struct KmSorted {
    double sd;         // sum of the closest distances of the block (fixed-shape tree: reproducible)
    double mu, s1, s2;
};

// the search for a value's position among the sorted values, from the top: xtop2 (every s2-th block start, at most KM_TOP2
// entries: in LDS for the whole launch) -> xtop (the first value of every block, one load per lane) -> the block itself
// (eight values per thread of the workgroup).  Two round trips to memory where a 32-ary search over xs takes five.
struct KmTop {
    double *xtop, *xtop2;
    int64_t nsb;       // blocks of KM_CHUNK sorted positions
    int s2, n2;        // xtop2[j] = xtop[j * s2], j < n2
};

__global__ __launch_bounds__(256) void km_sorted_init_kernel(const double *__restrict__ xs, const uint32_t *__restrict__ perm,
                                                             int64_t m, const KmState *__restrict__ st,
                                                             double *__restrict__ ds, uint32_t *__restrict__ rank,
                                                             KmSorted *__restrict__ sb, KmTop top)
{
    __shared__ double red[4];
    const double c = st->c0, csq = __dmul_rn(c, c);
    const int64_t p0 = (int64_t)blockIdx.x * KM_CHUNK;
    const int64_t p1 = p0 + KM_CHUNK < m ? p0 + KM_CHUNK : m;
    const double mu = xs[p0 + (p1 - p0) / 2];
    if (threadIdx.x == 0) {
        const double x0 = xs[p0];
        top.xtop[blockIdx.x] = x0;
        if (blockIdx.x % top.s2 == 0) top.xtop2[blockIdx.x / top.s2] = x0;
    }
    double sd = 0.0, s1 = 0.0, s2 = 0.0;
#pragma unroll
    for (int u = 0; u < KM_CHUNK / 256; ++u) {
        const int64_t p = p0 + u * 256 + threadIdx.x;
        if (p < p1) {
            const double x = xs[p];
            const double d = km_sqdist(c, csq, x);
            ds[p] = d;
            rank[perm[p]] = (uint32_t)p;
            sd += d;
            s1 += x - mu;
            s2 += (x - mu) * (x - mu);
        }
    }
    sd = km_block_sum(sd, red);
    s1 = km_block_sum(s1, red);
    s2 = km_block_sum(s2, red);
    if (threadIdx.x == 0) { KmSorted o; o.sd = sd; o.mu = mu; o.s1 = s1; o.s2 = s2; sb[blockIdx.x] = o; }
}

// ---- k-means++ : one further seed = km_prep_kernel (pick [+ update]) [-> km_gain_kernel] [-> km_update_kernel] --------
// argmin of the candidates' potentials (= potential - gain), np.argmin's first minimum.  Exact ties are COMMON on
// small inputs -- two isolated candidates that each capture only themselves and each other -- and sklearn's BLAS sums
// return bit-equal potentials for them: potentials within 1e-12 of the minimum count as tied, the first wins.
__device__ int km_best(const KmSeedRec *rec, int n_trials)
{
    const i128 pot = km_make128(rec->pot_lo, rec->pot_hi);
    double pd[KM_MAX_TRIALS];
    double lowest = 0.0;
#pragma unroll
    for (int j = 0; j < KM_MAX_TRIALS; ++j) {
        pd[j] = 0.0;
        if (j < n_trials) {
            pd[j] = km_to_double(pot - km_join(rec->gain[0][j], rec->gain[1][j], rec->gain[2][j]));
            lowest = (j == 0 || pd[j] < lowest) ? pd[j] : lowest;
        }
    }
    int best = n_trials - 1;
#pragma unroll
    for (int j = KM_MAX_TRIALS - 1; j >= 0; --j)
        if (j < n_trials && !(pd[j] > lowest + 1e-12 * lowest)) best = j;
    return best;
}

// Two searches in one wavefront over an ascending array a[0, n): lanes 0-31 find the first index with a >= tlo
// (0 when !has_lo), lanes 32-63 the first index with a > thi (n when !has_hi); 32 probes per step.
__device__ __forceinline__ void km_dual_search(const double *__restrict__ a, int64_t n, double tlo, double thi, bool has_lo,
                                               bool has_hi, int64_t &out_lo, int64_t &out_hi)
{
    const int lane = threadIdx.x & 63, half = lane >> 5, l = lane & 31;
    const double t = half ? thi : tlo;
    int64_t lo = 0, hi = n;                                     // the answer is in [lo, hi]
    if (!(half ? has_hi : has_lo)) lo = hi = half ? n : 0;
    for (;;) {
        const int64_t len = hi - lo;
        if (__ballot(len > 0) == 0) break;
        const int64_t stride = (len + 31) >> 5;
        const int64_t idx = lo + (int64_t)l * stride;
        bool pred = false;
        if (len > 0 && idx < hi) {
            const double val = a[idx];
            pred = half ? (val <= t) : (val < t);
        }
        const uint64_t bal = __ballot(pred);
        const int cnt = __popc((unsigned)(half ? (bal >> 32) : (bal & 0xFFFFFFFFull)));
        if (len > 0) {
            if (cnt == 0) {
                hi = lo;
            } else {
                const int64_t nlo = lo + (int64_t)(cnt - 1) * stride + 1;
                int64_t nhi = lo + (int64_t)cnt * stride;
                if (nhi > hi) nhi = hi;
                lo = nlo;
                hi = nhi;
            }
        }
    }
    out_lo = __shfl(lo, 0, 64);
    out_hi = __shfl(lo, 32, 64);
}

// argmin by one wavefront (lane j = candidate j, pd its potential; lanes >= n_trials are ignored): same rule as km_best
__device__ __forceinline__ int km_best_of(double pd, int n_trials, int lane)
{
    if (lane >= n_trials) pd = 1.79769313486231570e308;
    double lowest = pd;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const double o = __shfl_xor(lowest, off, 64);
        lowest = o < lowest ? o : lowest;
    }
    const uint64_t tied = __ballot(lane < n_trials && !(pd > lowest + 1e-12 * lowest));
    return __ffsll((long long)tied) - 1;
}

__device__ __forceinline__ int km_best_wave(const KmSeedRec *rec, int n_trials, int lane, int closed)
{
    double pd = 1.79769313486231570e308;
    if (lane < n_trials) {
        const i128 pot = km_make128(rec->pot_lo, rec->pot_hi);
        pd = closed ? km_to_double(pot) - rec->gain_d[lane]
                    : km_to_double(pot - km_join(rec->gain[0][lane], rec->gain[1][lane], rec->gain[2][lane]));
    }
    double lowest = pd;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const double o = __shfl_xor(lowest, off, 64);
        lowest = o < lowest ? o : lowest;
    }
    const uint64_t tied = __ballot(lane < n_trials && !(pd > lowest + 1e-12 * lowest));
    return __ffsll((long long)tied) - 1;
}

// a trial's candidate into the seed's record.  merge_tag != 0: workgroups of THIS launch read it as well (the update
// inside the pick kernel), from words that go straight to memory
__device__ __forceinline__ void km_publish(KmSeedRec *cur, int trial, double c, int64_t idx, int64_t lo, int64_t hi, double gain,
                                           u64 merge_tag)
{
    cur->cand_x[trial] = c;
    cur->cand_id[trial] = idx;
    cur->cand_lo[trial] = lo;
    cur->cand_hi[trial] = hi;
    cur->gain_d[trial] = gain;
    if (!merge_tag) return;
    // every word validates itself: no order among the stores, no second read for whoever waits
    const u64 cb = (u64)__double_as_longlong(c), gb = (u64)__double_as_longlong(gain), t = merge_tag << 48;
    u64 *w = cur->pub[trial];
    __hip_atomic_store(w + 0, (u64)lo | t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // positions are below 2^31
    __hip_atomic_store(w + 1, (u64)hi | t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(w + 2, (cb & 0xFFFFFFFFull) | t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(w + 3, (cb >> 32) | t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(w + 4, (gb & 0xFFFFFFFFull) | t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(w + 5, (gb >> 32) | t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// The end of a trial's pick, by the workgroup that holds the chosen index idx (hit_rank = its sorted position when the
// workgroup already knows it, 0xFFFFFFFF otherwise):
//  C: the candidate's neighbours s_L < c < s_R among the seeds and its range of sorted positions.  A value x > c can
//    only get closer to c than it is to its closest seed if (x - c)^2 - err < (x - s_R)^2 + err, err the rounding error
//    of the two evaluations of km_sqdist (<= 11 * 2^-53 * max|x|^2 each: three products and two sums of terms <= 4 max|x|^2),
//    i.e. x < (c + s_R) / 2 + err / (s_R - c);  same on the left.  The range is widened by three times that.
//  D: the candidate's gain.
__device__ void km_pick_tail(const double *__restrict__ xs, const double *__restrict__ ds, const uint32_t *__restrict__ rank,
                             int64_t m, int trial, int64_t idx, uint32_t hit_rank, bool has_newest, double newest,
                             int n_old, const double *__restrict__ sorted_old, int full_range, int closed, int slow_pick,
                             const KmSorted *__restrict__ sb, KmTop top, const double *s_top2, const double *s_seeds,
                             double amax, double quanta_per_unit, u64 merge_tag, KmState *st, KmSeedRec *cur)
{
    __shared__ u64 s_cnt4[4];
    __shared__ int64_t s_bounds[6];
    __shared__ double s_f[5], s_gred[4];
    __shared__ int s_flags;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    // ---- C: neighbours among the seeds (wavefront 0)
    if (wave == 0) {
        KM_TP(6);
        const double cx = xs[hit_rank != 0xFFFFFFFFu ? hit_rank : rank[idx]];
        int64_t pl, pr;                                         // pl seeds < cx, pr seeds <= cx
        double sl = 0.0, sr = 0.0;
        if (s_seeds) {                                          // the sorted seeds are in LDS
            int nl = 0, nr = 0;
            for (int i = lane; i < n_old; i += 64) { const double v = s_seeds[i]; nl += v < cx ? 1 : 0; nr += v <= cx ? 1 : 0; }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) { nl += __shfl_xor(nl, off, 64); nr += __shfl_xor(nr, off, 64); }
            pl = nl; pr = nr;
            if (pl > 0) sl = s_seeds[pl - 1];
            if (pr < n_old) sr = s_seeds[pr];
        } else {
            km_dual_search(sorted_old, n_old, cx, cx, true, true, pl, pr);
            if (pl > 0) sl = sorted_old[pl - 1];
            if (pr < n_old) sr = sorted_old[pr];
        }
        bool has_l = pl > 0, has_r = pr < n_old;
        if (has_newest) {                                       // the seed chosen by this launch, not in sorted_old
            if (newest < cx && (!has_l || newest > sl)) { sl = newest; has_l = true; }
            if (newest > cx && (!has_r || newest < sr)) { sr = newest; has_r = true; }
        }
        KM_TP(7);
        const double err = 64.0 * 1.1102230246251565e-16 * amax * amax, slack = 8.0 * 2.220446049250313e-16 * amax;
        const double wl = has_l ? err / (cx - sl) + slack : 0.0, wr = has_r ? err / (sr - cx) + slack : 0.0;
        // range [lo, hi): first value >= tlo .. first value > thi.  The values that CERTAINLY get closer (the same bound,
        // inward): first value > til .. first value >= tih
        const double inf = __builtin_inf();
        const bool open_l = !has_l || full_range, open_r = !has_r || full_range;
        const double tlo = open_l ? -inf : 0.5 * (sl + cx) - wl, til = has_l ? 0.5 * (sl + cx) + wl : -inf;
        const double thi = open_r ? inf : 0.5 * (cx + sr) + wr, tih = has_r ? 0.5 * (cx + sr) - wr : inf;
        int64_t Blo = 0, Bhi = 0;
        if (closed && !slow_pick) {
            // blocks of sorted positions that hold lo and hi: the last block that starts below tlo (at or below thi)
            int n2l = 0, n2h = 0;
            for (int i = lane; i < top.n2; i += 64) { const double v = s_top2[i]; n2l += v < tlo ? 1 : 0; n2h += v <= thi ? 1 : 0; }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) { n2l += __shfl_xor(n2l, off, 64); n2h += __shfl_xor(n2h, off, 64); }
            const int64_t jl = n2l > 0 ? n2l - 1 : 0, jh = n2h > 0 ? n2h - 1 : 0;
            const int64_t l0 = jl * top.s2, h0 = jh * top.s2;
            const int64_t l1 = l0 + top.s2 < top.nsb ? l0 + top.s2 : top.nsb, h1 = h0 + top.s2 < top.nsb ? h0 + top.s2 : top.nsb;
            int cl = 0, chh = 0;
            for (int64_t o = lane; o < top.s2; o += 64) {
                const double vl = top.xtop[l0 + o < l1 ? l0 + o : l1 - 1], vh = top.xtop[h0 + o < h1 ? h0 + o : h1 - 1];
                cl += (l0 + o < l1 && vl < tlo) ? 1 : 0;
                chh += (h0 + o < h1 && vh <= thi) ? 1 : 0;
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) { cl += __shfl_xor(cl, off, 64); chh += __shfl_xor(chh, off, 64); }
            const int64_t nbl = n2l > 0 ? l0 + cl : 0, nbh = n2h > 0 ? h0 + chh : 0;
            Blo = nbl > 0 ? nbl - 1 : 0;
            Bhi = nbh > 0 ? nbh - 1 : 0;
        }
        KM_TP(8);
        if (lane == 0) {
            s_f[0] = cx; s_f[1] = tlo; s_f[2] = til; s_f[3] = tih; s_f[4] = thi;
            s_bounds[4] = Blo; s_bounds[5] = Bhi;
            s_flags = (has_l ? 1 : 0) | (has_r ? 2 : 0);
        }
    }
    __syncthreads();
    const double c = s_f[0], csq = __dmul_rn(c, c);
    bool slow = !closed || slow_pick;
    if (!slow) {
        // ---- D (the usual way): lo and hi inside their blocks, and the candidate's gain = sum over its range of
        // d - min(d, dist) WITHOUT a pass over the range: the blocks of KM_CHUNK sorted positions strictly between the two
        // contribute sum(d) - sum((x - c)^2) from the block sums of the closest distances and the static block moments;
        // the two end blocks, already here for the counting, value by value with the expression the update uses.
        // fp64, fixed order: the potentials only feed the argmin with its 1e-12 tie rule (sklearn's own are BLAS sums).
        const double tlo = s_f[1], til = s_f[2], tih = s_f[3], thi = s_f[4];
        const int64_t Blo = s_bounds[4], Bhi = s_bounds[5];
        const bool same = Blo == Bhi;
        const int64_t a0 = Blo * KM_CHUNK, a1 = a0 + KM_CHUNK < m ? a0 + KM_CHUNK : m;
        const int64_t b0 = Bhi * KM_CHUNK, b1 = b0 + KM_CHUNK < m ? b0 + KM_CHUNK : m;
        double xa[KM_CHUNK / 256], da[KM_CHUNK / 256], xb[KM_CHUNK / 256], db[KM_CHUNK / 256];
#pragma unroll
        for (int u = 0; u < KM_CHUNK / 256; ++u) {
            const int64_t p = a0 + u * 256 + tid, q = p < a1 ? p : a1 - 1;
            xa[u] = xs[q];
            da[u] = ds[q];
        }
        if (!same) {
#pragma unroll
            for (int u = 0; u < KM_CHUNK / 256; ++u) {
                const int64_t p = b0 + u * 256 + tid, q = p < b1 ? p : b1 - 1;
                xb[u] = xs[q];
                db[u] = ds[q];
            }
        } else {
#pragma unroll
            for (int u = 0; u < KM_CHUNK / 256; ++u) { xb[u] = xa[u]; db[u] = da[u]; }
        }
        const int64_t Bf = Blo + 1 + tid;                       // this thread's first block in between
        KmSorted mid = sb[Bf < Bhi ? Bf : Blo];
        // four counts of at most KM_CHUNK (16 bits each) and the end blocks' part of the gain -- a value at position p is
        // in [lo, hi) exactly when tlo <= x <= thi -- in one reduction
        u64 cnt = 0;
        double g = 0.0;
#pragma unroll
        for (int u = 0; u < KM_CHUNK / 256; ++u) {
            if (a0 + u * 256 + tid < a1) {
                cnt += (xa[u] < tlo ? 1ull : 0ull) + (xa[u] <= til ? 1ull << 16 : 0ull);
                const double dj = km_sqdist(c, csq, xa[u]);
                g += (xa[u] >= tlo && xa[u] <= thi && dj < da[u]) ? da[u] - dj : 0.0;
            }
            if (b0 + u * 256 + tid < b1) {
                cnt += (xb[u] < tih ? 1ull << 32 : 0ull) + (xb[u] <= thi ? 1ull << 48 : 0ull);
                if (!same) {
                    const double dj = km_sqdist(c, csq, xb[u]);
                    g += (xb[u] <= thi && dj < db[u]) ? db[u] - dj : 0.0;
                }
            }
        }
        for (int64_t B = Bf; B < Bhi; B += 256) {              // (dropped below when the band leaves an end block)
            if (B != Bf) mid = sb[B];
            const double t = mid.mu - c;
            g += mid.sd - (mid.s2 + 2.0 * t * mid.s1 + (double)KM_CHUNK * t * t);
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) { cnt += __shfl_xor(cnt, off, 64); g += __shfl_xor(g, off, 64); }
        if (lane == 0) { s_cnt4[wave] = cnt; s_gred[wave] = g; }
        __syncthreads();
        cnt = s_cnt4[0] + s_cnt4[1] + s_cnt4[2] + s_cnt4[3];
        g = ((s_gred[0] + s_gred[1]) + s_gred[2]) + s_gred[3];
        const int64_t lo = a0 + (int64_t)(cnt & 0xFFFF), ilo = a0 + (int64_t)((cnt >> 16) & 0xFFFF);
        const int64_t ihi = b0 + (int64_t)((cnt >> 32) & 0xFFFF), hi = b0 + (int64_t)(cnt >> 48);
        // blocks in between are only summed in closed form when all their values certainly get closer: always, unless the
        // band where that is undecided reaches out of an end block (neighbouring seeds a rounding error apart)
        if (Bhi > Blo + 1 && (ilo > a1 - 1 || ihi <= b0)) {
            slow = true;
            __syncthreads();                                    // (s_gred is used again)
        } else {
            KM_TP(9);
            // (the gain in the quanta of the exact sums: a power of two)
            if (tid == 0) km_publish(cur, trial, c, idx, lo, hi, g * quanta_per_unit, merge_tag);
            return;
        }
    }
    // ---- the general way (and the comparison modes): positions by searches over all the sorted values
    if (wave == 0) {
        const double tlo = s_f[1], til = s_f[2], tih = s_f[3], thi = s_f[4];
        const bool has_l = s_flags & 1, has_r = s_flags & 2;
        int64_t lo = 0, hi = m;
        if (!full_range) km_dual_search(xs, m, tlo, thi, has_l, has_r, lo, hi);
        int64_t ilo = lo, ihi = hi;
        if (closed) {
            int64_t a, b;
            km_dual_search(xs, m, tih, til, has_r, has_l, a, b);    // a: first value >= tih, b: first value > til
            if (has_l) ilo = b < lo ? lo : b;
            if (has_r) ihi = a > hi ? hi : a;
            if (ilo > ihi) ilo = ihi;
        }
        if (lane == 0) {
            if (!closed) km_publish(cur, trial, c, idx, lo, hi, 0.0, 0);
            s_bounds[0] = lo; s_bounds[1] = hi; s_bounds[2] = ilo; s_bounds[3] = ihi;
        }
    }
    if (!closed) return;
    __syncthreads();
    {
        // blocks inside [ilo, ihi) in closed form, the ragged ends value by value
        const int64_t lo = s_bounds[0], hi = s_bounds[1], ilo = s_bounds[2], ihi = s_bounds[3];
        int64_t bl = (ilo + KM_CHUNK - 1) / KM_CHUNK, bh = ihi / KM_CHUNK;
        if (bl >= bh) { bl = 0; bh = 0; }
        const int64_t e0 = bl < bh ? bl * KM_CHUNK : hi;         // [lo, e0) and [e1, hi) value by value
        const int64_t e1 = bl < bh ? bh * KM_CHUNK : hi;
        double g = 0.0;
#pragma unroll 1
        for (int side = 0; side < 2; ++side) {
            const int64_t a = side ? e1 : lo, b = side ? hi : e0;
            for (int64_t base = a; base < b; base += 8 * 256) {
                double xv[8], dv[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int64_t p = base + u * 256 + tid;
                    const int64_t q = p < b ? p : b - 1;
                    xv[u] = xs[q];
                    dv[u] = ds[q];
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const double dj = km_sqdist(c, csq, xv[u]);
                    g += (base + u * 256 + tid < b && dj < dv[u]) ? dv[u] - dj : 0.0;
                }
            }
        }
        for (int64_t B = bl + tid; B < bh; B += 256) {
            const KmSorted o = sb[B];
            const double t = o.mu - c;
            g += o.sd - (o.s2 + 2.0 * t * o.s1 + (double)KM_CHUNK * t * t);
        }
        g = km_block_sum(g, s_gred);
        if (tid == 0) km_publish(cur, trial, c, idx, lo, hi, g * quanta_per_unit, merge_tag);
    }
}

// The pick of seed `seed_no`: KM_SUB workgroups per trial -- one workgroup gathers ~330 scattered values per microsecond,
// so the walk of an index block (32 768 indices at 30 M values) is spread over sixteen.  Each publishes the sum of its
// sixteenth tagged with the seed's number and reads the sixteen sums of its trial until all carry the tag (the 16 x n_trials
// workgroups of a launch are far fewer than the GPU holds at once: nobody waits for a workgroup that cannot start; the
// wait is bounded all the same and reports a fault).  The one whose sixteenth holds the trial's r continues with the
// values it gathered still in registers (km_pick_tail); the others leave.
// Every workgroup (the common part is cheap and repeated):
//  A (choose_prev): the winner among the previous seed's candidates becomes seed seed_no - 1; the first workgroup records
//    it and writes the sorted list of seeds with it inserted into the other buffer.
//  B (do_pick): inclusive prefix of the index-block sums = sklearn's cumulative sum at block ends; total = the potential.
//    The trial's r = uniform * potential lies in the first block whose cumulative sum reaches it (np.searchsorted, left);
//    workgroup `sub` gathers the closest distances of its sixteenth of that block (through rank) and records their sum.
//  C (merge): the update of the closest distances with the seed chosen among this launch's candidates, by all the
//    workgroups once every trial is recorded -- for the seeds whose ranges are short enough for 16 x n_trials workgroups
//    (the host decides by the seed's number); km_update_kernel otherwise.
__global__ __launch_bounds__(256) void km_prep_kernel(const double *__restrict__ xs, double *ds,
                                                      const uint32_t *__restrict__ rank, const uint32_t *__restrict__ perm, int64_t m,
                                                      i64 *bacc, int nblocks, int block_shift,
                                                      const double *__restrict__ uniform, int n_trials, int seed_no,
                                                      int choose_prev, int do_pick, int full_range, int closed, int slow_pick, int merge,
                                                      int lose_a_sum, KmSorted *sb, KmTop top, KmState *st, double *__restrict__ seeds_x,
                                                      int64_t *__restrict__ seeds_id, double *__restrict__ sorted2, int sorted_ld)
{
    // (the LDS of the searches -- top level of the sorted values, sorted seeds -- later holds the update's block sums)
    __shared__ u64 s_big[3 * KM_MAX_BLOCKS > KM_TOP2 + KM_SEEDS_LDS ? 3 * KM_MAX_BLOCKS : KM_TOP2 + KM_SEEDS_LDS];
    double *s_top2 = reinterpret_cast<double *>(s_big), *s_seeds = s_top2 + KM_TOP2;
    __shared__ int64_t s_ulo, s_uhi;
    __shared__ double s_uc, s_ured[4];
    __shared__ u64 s_mine[2];
    __shared__ int64_t s_idx, s_o[2];
    __shared__ double s_dv[8];
    __shared__ uint32_t s_rk[8], s_hit_rank;
    __shared__ int s_first, s_sidx, s_reach;
    __shared__ double s_newest;
    __shared__ i64 s_newid;
    __shared__ u64 s_w_lo[4], s_w_hi[4], s_carry[2], s_expect[2];
    __shared__ double s_expect_d;
    __shared__ int s_cnt[4];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int sub = blockIdx.x, trial = blockIdx.y;
    const bool lead = sub == 0 && trial == 0;
    KmSeedRec *cur = &st->rec[seed_no & 1];
    const KmSeedRec *prev = &st->rec[(seed_no - 1) & 1];
    const int n_old = choose_prev ? seed_no - 1 : seed_no;      // seeds in the sorted list: it lives in buffer (count & 1)
    const double *sorted_old = sorted2 + (size_t)(n_old & 1) * sorted_ld;
    double *sorted_new = sorted2 + (size_t)((n_old + 1) & 1) * sorted_ld;
    // ---- B (first half, issued before the choice so that the two round trips to memory overlap): this thread's eight
    // consecutive block sums
    i128 inc8[KM_MAX_BLOCKS / 256];
    i128 loc = 0;
    const bool seeds_in_lds = n_old <= KM_SEEDS_LDS;
    const int faults_before = st->faults;                       // (branched on below, when it has long arrived)
    // (scalars of the later steps: loaded now, not in the middle of the chain)
    const double u_trial = do_pick ? uniform[trial] : 0.0;
    const KmLimb L = st->limb;
    const double amax = st->amax;
    if (do_pick) {
        // (what the end of the pick searches, into LDS now: used after several barriers)
        if (closed && !slow_pick)
            for (int i = tid; i < top.n2; i += 256) s_top2[i] = top.xtop2[i];
        if (seeds_in_lds)
            for (int i = tid; i < n_old; i += 256) s_seeds[i] = sorted_old[i];
        i64 raw[KM_MAX_BLOCKS / 256][3];
#pragma unroll
        for (int q = 0; q < KM_MAX_BLOCKS / 256; ++q) {
            const int b = tid * (KM_MAX_BLOCKS / 256) + q;
            const int bb = b < nblocks ? b : nblocks - 1;
            raw[q][0] = bacc[4 * bb]; raw[q][1] = bacc[4 * bb + 1]; raw[q][2] = bacc[4 * bb + 2];
        }
#pragma unroll
        for (int q = 0; q < KM_MAX_BLOCKS / 256; ++q) {
            if (tid * (KM_MAX_BLOCKS / 256) + q < nblocks) loc += km_join(raw[q][0], raw[q][1], raw[q][2]);
            inc8[q] = loc;
        }
    }
    double newest = 0.0;                                        // the seed chosen here, not yet in sorted_old
    KM_T(0);
#ifdef KM_DBG_TIMING
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0 && seed_no == KM_DBG_TIMING + 1) st->dbg[20] = wall_clock64();
#endif
    if (choose_prev) {
        if (wave == 0) {
            // (the candidate values ride along with the gains: one round trip to memory instead of two)
            const bool on = lane < n_trials && lane < KM_MAX_TRIALS;
            const double cx = on ? prev->cand_x[lane] : 0.0;
            const i64 cid = on ? prev->cand_id[lane] : 0;
            const int best = km_best_wave(prev, n_trials, lane, closed);
            const double bx = __shfl(cx, best, 64);
            const i64 bid = __shfl(cid, best, 64);
            if (lane == 0) {
                s_newest = bx;
                s_newid = bid;
                const i128 after = km_make128(prev->pot_lo, prev->pot_hi) -
                                   km_join(prev->gain[0][best], prev->gain[1][best], prev->gain[2][best]);
                s_expect[0] = (u64)after;
                s_expect[1] = (u64)(after >> 64);
                s_expect_d = km_to_double(km_make128(prev->pot_lo, prev->pot_hi)) - prev->gain_d[best];
            }
        }
        __syncthreads();
        newest = s_newest;
        if (lead) {
            int pos = 0;                                        // seeds <= newest: it goes behind them
            for (int i0 = 0; i0 < n_old; i0 += 256) pos += __syncthreads_count(i0 + tid < n_old && sorted_old[i0 + tid] <= newest);
            for (int i = tid; i < n_old; i += 256) sorted_new[i < pos ? i : i + 1] = sorted_old[i];
            if (tid == 0) {
                sorted_new[pos] = newest;
                seeds_x[seed_no - 1] = newest;
                seeds_id[seed_no - 1] = s_newid;
            }
        }
    }
    // a wait between workgroups has expired in an earlier launch of this run (fault bit 4): the run is lost and reported
    // as such -- do not wait again, seed after seed
    if (!do_pick || (faults_before & 16)) return;
    KM_T(1);
    const i128 winc = km_wave_scan128(loc, lane);
    if (lane == 63) { s_w_lo[wave] = (u64)winc; s_w_hi[wave] = (u64)(winc >> 64); }
    __syncthreads();
    i128 before = 0, total = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        const i128 wt = km_make128(s_w_lo[w], s_w_hi[w]);
        if (w < wave) before += wt;
        total += wt;
    }
    before += winc - loc;                                       // cumulative sum before this thread's first block
    KM_T(2);
    const double r = u_trial * km_to_double(total);
    const i128 R = km_ceil128(r);
    // the block: number of blocks whose inclusive cumulative sum is < R (they are non-decreasing)
    int below = 0;
#pragma unroll
    for (int q = 0; q < KM_MAX_BLOCKS / 256; ++q)
        below += (tid * (KM_MAX_BLOCKS / 256) + q < nblocks && before + inc8[q] < R) ? 1 : 0;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) below += __shfl_xor(below, off, 64);
    if (lane == 0) s_cnt[wave] = below;
    __syncthreads();
    const int blk = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
    const bool clipped = R > total || blk >= nblocks;           // np.clip(candidate_ids, None, n - 1)
    if (blk == 0 && tid == 0) { s_carry[0] = 0; s_carry[1] = 0; }
#pragma unroll
    for (int q = 0; q < KM_MAX_BLOCKS / 256; ++q)
        if (tid * (KM_MAX_BLOCKS / 256) + q == blk - 1) { const i128 cy = before + inc8[q]; s_carry[0] = (u64)cy; s_carry[1] = (u64)(cy >> 64); }
    __syncthreads();
    KM_T(3);
    if (lead) {
        if (tid == 0) {
            // the block sums after the update = the potential the winner's gain promised: to the last bit with the
            // exact gains of the gain pass, to rounding with the closed-form ones
            if (choose_prev && !closed && ((u64)total != s_expect[0] || (u64)(total >> 64) != s_expect[1])) atomicOr(&st->faults, 1);
            if (choose_prev && closed) {
                const double got = km_to_double(total), prev_pot = km_to_double(km_make128(prev->pot_lo, prev->pot_hi));
                if (!(fabs(got - s_expect_d) <= 1e-9 * prev_pot)) atomicOr(&st->faults, 8);
            }
            cur->pot_lo = (u64)total;
            cur->pot_hi = (u64)(total >> 64);
        }
        if (tid < 3 * KM_MAX_TRIALS) cur->gain[tid / KM_MAX_TRIALS][tid % KM_MAX_TRIALS] = 0;
    }
    const i128 carry = km_make128(s_carry[0], s_carry[1]);
    const double *seeds_lds = seeds_in_lds ? s_seeds : nullptr;
    const u64 merge_tag = merge ? (u64)seed_no : 0;
    if (clipped) {                                              // the last index, whatever the block holds
        if (sub == 0) km_pick_tail(xs, ds, rank, m, trial, m - 1, 0xFFFFFFFFu, choose_prev != 0, newest, n_old, sorted_old, full_range,
                                   closed, slow_pick, sb, top, s_top2, seeds_lds, amax, L.sC, merge_tag, st, cur);
    } else {
        // ---- this workgroup's sixteenth of the block
        const int64_t bsize = (int64_t)1 << block_shift, ssize = bsize / KM_SUB;
        const int64_t per = ssize >= 256 ? ssize >> 8 : 1;
        const int64_t bend = (((int64_t)blk + 1) << block_shift) < m ? (((int64_t)blk + 1) << block_shift) : m;
        const int64_t s0 = ((int64_t)blk << block_shift) + (int64_t)sub * ssize;
        const int64_t s1 = s0 + ssize < bend ? s0 + ssize : bend;
        const int64_t i0 = s0 + (int64_t)tid * per;
        const int64_t i1 = i0 + per < s1 ? i0 + per : s1;
        double wa = 0.0, wb = 0.0, wc = 0.0;                     // limb sums of this thread's values: exact
        uint32_t rk0[8];                                        // the first eight stay in registers for the last step
        double dv0[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {                           // ranks first, then the distances they point at
            const int64_t i = i0 + q < i1 ? i0 + q : (i1 > 0 ? i1 - 1 : 0);
            rk0[q] = rank[i < m ? i : m - 1];
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) dv0[q] = ds[rk0[q]];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            double a, b, cc;
            km_split(L, i0 + q < i1 ? dv0[q] : 0.0, a, b, cc);
            wa += a; wb += b; wc += cc;
        }
        for (int64_t ib = i0 + 8; ib < i1; ib += 8) {           // (only when m is beyond 2^29)
            uint32_t rk[8];
            double dv[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) rk[q] = rank[ib + q < i1 ? ib + q : i1 - 1];
#pragma unroll
            for (int q = 0; q < 8; ++q) dv[q] = ds[rk[q]];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                double a, b, cc;
                km_split(L, ib + q < i1 ? dv[q] : 0.0, a, b, cc);
                wa += a; wb += b; wc += cc;
            }
        }
        KM_T(4);
        // prefix over the threads (wavefront scan, then the four wavefront totals), total of the workgroup
        const i128 lsum = km_join(__double2ll_rn(wa * L.sA), __double2ll_rn(wb * L.sB), __double2ll_rn(wc * L.sC));
        const i128 linc = km_wave_scan128(lsum, lane);
        if (lane == 63) { s_w_lo[wave] = (u64)linc; s_w_hi[wave] = (u64)(linc >> 64); }
        if (tid == 0) { s_first = 256; s_idx = -1; s_hit_rank = 0xFFFFFFFFu; }
        __syncthreads();
        i128 wbefore = 0, wtotal = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const i128 wt = km_make128(s_w_lo[w], s_w_hi[w]);
            if (w < wave) wbefore += wt;
            wtotal += wt;
        }
        // publish: three words of 48 bits under the seed's number (a sum is below 2^112 quanta)
        const u64 mask48 = (1ull << 48) - 1, tag = (u64)(seed_no & 0xFFFF);
        if (wave == 0) {
            u64 *words = reinterpret_cast<u64 *>(&cur->sub[trial][0][0]);
            // (lose_a_sum: the test of the time-out -- one workgroup of the launch never publishes, GRX_KMEANS_LOSE_A_SUM)
            if (lane == 0 && !(lose_a_sum && sub == 3 && trial == 0)) {
                __hip_atomic_store(words + 3 * sub, ((u64)wtotal & mask48) | (tag << 48), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(words + 3 * sub + 1, ((u64)(wtotal >> 48) & mask48) | (tag << 48), __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(words + 3 * sub + 2, ((u64)(wtotal >> 96) & mask48) | (tag << 48), __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
            }
            // the sixteen sums of the trial: lane e reads word e until every word carries the tag
            const int e = lane < 3 * KM_SUB ? lane : 0;
            u64 w = 0;
            int spins = 0;
            for (;;) {
                w = __hip_atomic_load(words + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (__ballot((w >> 48) != tag) == 0) break;
                if (++spins >= KM_SPIN_LIMIT) break;
                __builtin_amdgcn_s_sleep(2);
            }
            const u64 pay = w & mask48;
            const int src = lane < KM_SUB ? 3 * lane : 0;
            const u64 p0 = __shfl(pay, src, 64), p1 = __shfl(pay, src + 1, 64), p2 = __shfl(pay, src + 2, 64);
            const i128 sv = lane < KM_SUB ? (i128)p0 + ((i128)p1 << 48) + ((i128)p2 << 96) : (i128)0;
            const i128 sinc = km_wave_scan128(sv, lane);
            const uint64_t reach = __ballot(lane < KM_SUB && carry + sinc >= R);
            const int sidx = reach ? __ffsll((long long)reach) - 1 : KM_SUB - 1;
            const i128 scarry = carry + km_make128(__shfl((u64)(sinc - sv), sidx, 64), __shfl((u64)((sinc - sv) >> 64), sidx, 64));
            if (lane == 0) {
                if (spins >= KM_SPIN_LIMIT) { atomicOr(&st->faults, 16); s_sidx = -1; }
                else s_sidx = sidx;
                s_reach = reach != 0;
                s_carry[0] = (u64)scarry; s_carry[1] = (u64)(scarry >> 64);
            }
        }
        __syncthreads();
        KM_T(5);
        if (sub == s_sidx) {
            // ---- this workgroup's sixteenth holds the index: the thread, then the value
            KM_TP(5);
            const i128 mine = km_make128(s_carry[0], s_carry[1]) + wbefore + linc - lsum;   // cumulative sum before this thread's first index
            if (s_reach && mine + lsum >= R) atomicMin(&s_first, tid);
            __syncthreads();
            const int owner = s_first;
            if (owner == 256) {
                if (tid == 0) { atomicOr(&st->faults, 4); s_idx = s1 - 1; }
            } else if (per <= 8) {
                // the owner's (at most eight) values through LDS to wavefront 0: one value per lane, inclusive scan, first hit
                if (tid == owner) {
#pragma unroll
                    for (int q = 0; q < 8; ++q) { s_dv[q] = dv0[q]; s_rk[q] = rk0[q]; }
                    s_mine[0] = (u64)mine; s_mine[1] = (u64)(mine >> 64);
                    s_o[0] = i0; s_o[1] = i1;
                }
                __syncthreads();
                if (wave == 0) {
                    const int64_t o0 = s_o[0], o1 = s_o[1];
                    const i128 qv = lane < o1 - o0 ? km_quanta(L, s_dv[lane < 8 ? lane : 0]) : (i128)0;
                    const i128 qinc = km_wave_scan128(qv, lane);
                    const uint64_t ok = __ballot(lane < o1 - o0 && km_make128(s_mine[0], s_mine[1]) + qinc >= R);
                    const int h = ok ? __ffsll((long long)ok) - 1 : (int)(o1 - o0) - 1;
                    if (lane == 0) { s_idx = o0 + h; s_hit_rank = s_rk[h]; }
                }
            } else if (wave == (owner >> 6)) {
                // the owner's wavefront walks the owner's indices together
                const int ol = owner & 63;
                const int64_t o0 = __shfl(i0, ol, 64), o1 = __shfl(i1, ol, 64);
                i128 run = km_make128(__shfl((u64)mine, ol, 64), __shfl((u64)(mine >> 64), ol, 64));
                int64_t hit = o1 - 1;
                for (int64_t ib = o0; ib < o1; ib += 64) {
                    const int64_t i = ib + lane;
                    const i128 qv = i < o1 ? km_quanta(L, ds[rank[i]]) : (i128)0;
                    const i128 qinc = km_wave_scan128(qv, lane);
                    const uint64_t ok = __ballot(i < o1 && run + qinc >= R);
                    if (ok) { hit = ib + __ffsll((long long)ok) - 1; break; }
                    run += km_make128(__shfl((u64)qinc, 63, 64), __shfl((u64)(qinc >> 64), 63, 64));
                }
                if (lane == 0) s_idx = hit;
            }
            __syncthreads();
            int64_t idx = s_idx;
            if (idx > m - 1) idx = m - 1;
            km_pick_tail(xs, ds, rank, m, trial, idx, s_hit_rank, choose_prev != 0, newest, n_old, sorted_old, full_range, closed,
                         slow_pick, sb, top, s_top2, seeds_lds, amax, L.sC, merge_tag, st, cur);
        }
    }
    if (!merge) return;
    // ---- C: every workgroup of the launch takes part in the update once all trials are recorded
    __syncthreads();                                            // (the searches are over: their LDS is free)
    u64 *s_acc = s_big;
    for (int i = tid; i < 3 * nblocks; i += 256) s_acc[i] = 0;
    if (wave == 0) {
        // lane e reads words e and e + 64 of the 6 * n_trials published ones until all carry the seed's number
        const u64 *words = &cur->pub[0][0];
        const int nw = 6 * n_trials, e0 = lane < nw ? lane : 0, e1 = lane + 64 < nw ? lane + 64 : 0;
        u64 r0 = 0, r1 = 0;
        int spins = 0;
        for (;;) {
            r0 = __hip_atomic_load(words + e0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            r1 = __hip_atomic_load(words + e1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (__ballot((r0 >> 48) != merge_tag || (r1 >> 48) != merge_tag) == 0) break;
            if (++spins >= KM_SPIN_LIMIT) break;
            __builtin_amdgcn_s_sleep(2);
        }
        u64 f[6];                                               // lane j: the six words of trial j
#pragma unroll
        for (int q = 0; q < 6; ++q) {
            const int src = lane < n_trials ? 6 * lane + q : q;
            const u64 a = __shfl(r0, src & 63, 64), b = __shfl(r1, src & 63, 64);
            f[q] = (src < 64 ? a : b) & ((1ull << 48) - 1);
        }
        const i64 lo_j = (i64)f[0], hi_j = (i64)f[1];
        const double cx_j = __longlong_as_double((long long)(f[2] | (f[3] << 32)));
        const double g_j = __longlong_as_double((long long)(f[4] | (f[5] << 32)));
        const int best = km_best_of(km_to_double(total) - g_j, n_trials, lane);
        const i64 blo = __shfl(lo_j, best, 64), bhi = __shfl(hi_j, best, 64);
        const double bc = __shfl(cx_j, best, 64);
        if (lane == 0) {
            const bool ok = spins < KM_SPIN_LIMIT;
            if (!ok) atomicOr(&st->faults, 16);
            s_ulo = ok ? blo : 0; s_uhi = ok ? bhi : 0; s_uc = bc;
        }
    }
    __syncthreads();
    {
        const int64_t lo = s_ulo, hi = s_uhi;
        const double c = s_uc, csq = __dmul_rn(c, c);
        const int64_t first = lo / KM_CHUNK;
        const int64_t chunks = hi > lo ? (hi - 1) / KM_CHUNK - first + 1 : 0;
        const int64_t wg = (int64_t)blockIdx.y * gridDim.x + blockIdx.x, nwg = (int64_t)gridDim.x * gridDim.y;
        if (wg >= chunks) return;
        for (int64_t ch = wg; ch < chunks; ch += nwg) {
            const int64_t p0 = (first + ch) * KM_CHUNK;
            const int64_t p1 = p0 + KM_CHUNK < m ? p0 + KM_CHUNK : m;
            double xv[KM_CHUNK / 256], dv[KM_CHUNK / 256];
            uint32_t iv[KM_CHUNK / 256];
#pragma unroll
            for (int u = 0; u < KM_CHUNK / 256; ++u) {
                const int64_t p = p0 + u * 256 + tid;
                const int64_t q = p < p1 ? p : p1 - 1;
                xv[u] = xs[q];
                dv[u] = ds[q];
                iv[u] = perm[q];
            }
            __builtin_amdgcn_sched_barrier(0);
            double left = 0.0;                                  // what stays: the block's new sum
#pragma unroll
            for (int u = 0; u < KM_CHUNK / 256; ++u) {
                const int64_t p = p0 + u * 256 + tid;
                if (p >= p1) continue;
                const double d = dv[u];
                const double dj = km_sqdist(c, csq, xv[u]);
                if (p >= lo && p < hi && dj < d) {
                    ds[p] = dj;
                    left += dj;
                    double a, b, cc, aj, bj, cj;
                    km_split(L, d, a, b, cc);
                    km_split(L, dj, aj, bj, cj);
                    const i64 ua = -__double2ll_rn((a - aj) * L.sA), ub = -__double2ll_rn((b - bj) * L.sB),
                              uc = -__double2ll_rn((cc - cj) * L.sC);
                    const int ib = (int)(iv[u] >> block_shift);
                    if (ua) atomicAdd(&s_acc[3 * ib], (u64)ua);
                    if (ub) atomicAdd(&s_acc[3 * ib + 1], (u64)ub);
                    if (uc) atomicAdd(&s_acc[3 * ib + 2], (u64)uc);
                } else {
                    left += d;
                }
            }
            left = km_block_sum(left, s_ured);                  // (the same shape as km_update_kernel's)
            if (tid == 0) sb[first + ch].sd = left;
        }
        __syncthreads();
        for (int i = tid; i < 3 * nblocks; i += 256) {
            const u64 v = s_acc[i];
            if (v) atomicAdd(reinterpret_cast<u64 *>(bacc) + 4 * (i / 3) + i % 3, v);
        }
    }
}

// union of the candidates' ranges as disjoint intervals in ascending order, and their prefix in chunks of KM_CHUNK
struct KmIntervals {
    int n;
    int64_t lo[KM_MAX_TRIALS], hi[KM_MAX_TRIALS], chunk0[KM_MAX_TRIALS + 1];
};

// by ONE wavefront: lane j < n_trials brings range j.  Bitonic sort of the sixteen (lo, hi) by lo, running maximum of
// hi, a range that starts beyond it opens a new interval -- a single lane doing this through LDS took 4 us per launch.
template <int NT>
__device__ __forceinline__ void km_merge_intervals_wave(int64_t lo64, int64_t hi64, int n_trials, int lane, KmIntervals *out)
{
    // positions are < 2^31 (grx_kmeans1d requires m < 2^31): 32-bit shuffles, and a sorting network of NT lanes only
    const int none = 0x7FFFFFFF;
    int lo = (int)lo64, hi = (int)hi64;
    if (lane >= n_trials || lane >= NT || hi <= lo) lo = hi = none;
#pragma unroll
    for (int k = 2; k <= NT; k <<= 1) {
#pragma unroll
        for (int j = k >> 1; j > 0; j >>= 1) {
            const int olo = __shfl_xor(lo, j, 64), ohi = __shfl_xor(hi, j, 64);
            const bool take_min = ((lane & k) == 0) == ((lane & j) == 0);
            if (take_min ? (olo < lo) : (olo > lo)) { lo = olo; hi = ohi; }
        }
    }
    const bool valid = lane < NT && lo != none;
    int pmax = valid ? hi : (int)0x80000000;
#pragma unroll
    for (int off = 1; off < NT; off <<= 1) {
        const int o = __shfl_up(pmax, off, 64);
        if (lane >= off && o > pmax) pmax = o;
    }
    const int before = __shfl_up(pmax, 1, 64);
    const bool start = valid && (lane == 0 || lo > before);
    const uint64_t starts = __ballot(start), valids = __ballot(valid);
    const int n = __popcll(starts);
    const int gid = __popcll(starts & ((2ull << lane) - 1)) - 1;
    const bool last = valid && (((starts >> (lane + 1)) & 1ull) || !((valids >> (lane + 1)) & 1ull));
    // interval g: its first range's lo, the running maximum at its last range; both to lane g by ballots and shuffles
    int glo = 0, ghi = 0;
    {
        // the lane that starts group `lane` / ends it: the (lane + 1)-th set bit of starts / of lasts
        const uint64_t lasts = __ballot(last);
        uint64_t sbits = starts, lbits = lasts;
        int s_lane = 0, l_lane = 0;
        for (int q = 0; q <= lane && q < n; ++q) {
            s_lane = __ffsll((long long)sbits) - 1; sbits &= sbits - 1;
            l_lane = __ffsll((long long)lbits) - 1; lbits &= lbits - 1;
        }
        glo = __shfl(lo, s_lane, 64);
        ghi = __shfl(pmax, l_lane, 64);
    }
    (void)gid;
    int chunks = 0;
    if (lane < n) chunks = (ghi - glo + KM_CHUNK - 1) / KM_CHUNK;
    int inc = chunks;
#pragma unroll
    for (int off = 1; off < NT; off <<= 1) {
        const int o = __shfl_up(inc, off, 64);
        if (lane >= off) inc += o;
    }
    if (lane < n) { out->lo[lane] = glo; out->hi[lane] = ghi; out->chunk0[lane] = inc - chunks; }
    if (lane == (n > 0 ? n - 1 : 0)) { out->chunk0[n] = n > 0 ? inc : 0; out->n = n; }
}

// gain: for every candidate the exact sum over its range of d - min(d, distance to the candidate).  One pass over the
// union of the ranges; a value is tested against the candidates whose range holds it.
template <int NT>
__global__ __launch_bounds__(256) void km_gain_kernel(const double *__restrict__ xs, const double *__restrict__ ds,
                                                      KmState *st, int seed_no, int n_trials)
{
    __shared__ KmIntervals s_iv;
    __shared__ int64_t s_rng[2][KM_MAX_TRIALS];
    __shared__ double s_cx[KM_MAX_TRIALS];
    __shared__ u64 s_g[3][NT];
    KmSeedRec *rec = &st->rec[seed_no & 1];
    const KmLimb L = st->limb;
    if (threadIdx.x < 64) {                                     // everything this launch needs, in one round trip
        const int j = threadIdx.x;
        const bool on = j < n_trials && j < KM_MAX_TRIALS;
        const int64_t lo_j = on ? rec->cand_lo[j] : 0, hi_j = on ? rec->cand_hi[j] : 0;
        const double cx_j = on ? rec->cand_x[j] : 0.0;
        if (j < KM_MAX_TRIALS) { s_rng[0][j] = lo_j; s_rng[1][j] = hi_j; s_cx[j] = cx_j; }
        if (j < 3 * NT) (&s_g[0][0])[j] = 0;
        KM_TG(10);
        km_merge_intervals_wave<NT>(lo_j, hi_j, n_trials, j, &s_iv);
    }
    KM_TG(11);
    __syncthreads();
    KM_TG(12);
    const int n_iv = s_iv.n;
    const int64_t total_chunks = s_iv.chunk0[n_iv];
    if ((int64_t)blockIdx.x >= total_chunks) return;
    double c[NT], csq[NT], ga[NT], gb[NT], gc[NT];
    int64_t lo[NT], hi[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        c[j] = s_cx[j];
        csq[j] = __dmul_rn(c[j], c[j]);
        lo[j] = j < n_trials ? rec->cand_lo[j] : 0;
        hi[j] = j < n_trials ? rec->cand_hi[j] : 0;
        ga[j] = gb[j] = gc[j] = 0.0;
    }
    int iv = 0;
    for (int64_t ch = blockIdx.x; ch < total_chunks; ch += gridDim.x) {
        while (iv + 1 < n_iv && s_iv.chunk0[iv + 1] <= ch) ++iv;
        const int64_t p0 = s_iv.lo[iv] + (ch - s_iv.chunk0[iv]) * KM_CHUNK;
        const int64_t iv_end = s_iv.hi[iv];
        const int64_t p1 = p0 + KM_CHUNK < iv_end ? p0 + KM_CHUNK : iv_end;
        double xv[KM_CHUNK / 256], dv[KM_CHUNK / 256];
#pragma unroll
        for (int u = 0; u < KM_CHUNK / 256; ++u) {
            const int64_t p = p0 + u * 256 + threadIdx.x;
            const int64_t q = p < p1 ? p : p1 - 1;
            xv[u] = xs[q];
            dv[u] = ds[q];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < KM_CHUNK / 256; ++u) {
            const int64_t p = p0 + u * 256 + threadIdx.x;
            if (p >= p1) continue;
            const double x = xv[u], d = dv[u];
            if (!(d > 0.0)) continue;                          // nothing to gain
            double a, b, cc;
            km_split(L, d, a, b, cc);
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                if (j < n_trials && p >= lo[j] && p < hi[j]) {
                    const double dj = km_sqdist(c[j], csq[j], x);
                    if (dj < d) {
                        double aj, bj, cj;
                        km_split(L, dj, aj, bj, cj);
                        ga[j] += a - aj; gb[j] += b - bj; gc[j] += cc - cj;     // exact: < 2^20 terms per thread
                    }
                }
            }
        }
    }
    KM_TG(13);
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        // (a chunk lies in the ranges of one or two candidates: the others' sums are zero in the whole wavefront)
        if (j < n_trials && __ballot(ga[j] != 0.0 || gb[j] != 0.0 || gc[j] != 0.0) != 0) {
            const i64 ua = km_wave_sum_i64(__double2ll_rn(ga[j] * L.sA));
            const i64 ub = km_wave_sum_i64(__double2ll_rn(gb[j] * L.sB));
            const i64 uc = km_wave_sum_i64(__double2ll_rn(gc[j] * L.sC));
            if (lane == 0) {
                if (ua) atomicAdd(&s_g[0][j], (u64)ua);
                if (ub) atomicAdd(&s_g[1][j], (u64)ub);
                if (uc) atomicAdd(&s_g[2][j], (u64)uc);
            }
        }
    }
    __syncthreads();
    if (threadIdx.x < 3 * NT) {
        const int limb = threadIdx.x / NT, j = threadIdx.x % NT;
        if (j < n_trials) km_atomic_add_i64(&rec->gain[limb][j], (i64)s_g[limb][j]);
    }
    KM_TG(14);
}

// update: d = min(d, distance to the chosen seed) over the winner's range.  What leaves the closest distances leaves
// the sums of the INDEX blocks they belong to: collected per workgroup in LDS (integer limbs, any order), then added
// to the block sums with integer atomics -- a straight atomic per value runs at 23 G atomics/s on MI355X
// (tools/microbench/atomic_scatter.hip: 3.9 ms for the three limbs of 30 M values against 0.11 ms for the pass itself).
__global__ __launch_bounds__(KM_UPDATE_THREADS) void km_update_kernel(const double *__restrict__ xs, double *__restrict__ ds,
                                                        const uint32_t *__restrict__ perm, i64 *__restrict__ bacc,
                                                        int nblocks, int block_shift, int64_t m, KmSorted *__restrict__ sb,
                                                        const KmState *__restrict__ st, int seed_no, int n_trials, int closed)
{
    // KM_UPDATE_THREADS / 256 groups of 256 threads, one chunk each at a time, ONE set of LDS sums: what a workgroup adds to
    // the index-block sums at the end is up to 3 * nblocks atomics whatever it processed (a chunk's 2048 values already
    // touch two thirds of 2048 blocks), so more chunks in flight per workgroup, not more workgroups
    constexpr int GROUPS = KM_UPDATE_THREADS / 256;
    __shared__ u64 s_acc[3 * KM_MAX_BLOCKS];
    __shared__ int64_t s_lo, s_hi;
    __shared__ double s_c, s_red[2][GROUPS * 4];
    const KmSeedRec *rec = &st->rec[seed_no & 1];
    const int grp = threadIdx.x >> 8, t = threadIdx.x & 255;
    KM_TG(15);
    const KmLimb L = st->limb;
    // (the other wavefronts clear the LDS sums while wavefront 0 waits for the candidates' records)
    if (threadIdx.x >= 64)
        for (int i = threadIdx.x - 64; i < 3 * nblocks; i += KM_UPDATE_THREADS - 64) s_acc[i] = 0;
    if (threadIdx.x < 64) {
        // (every candidate's range and value ride along with the gains: one round trip to memory, not two)
        const int lane = threadIdx.x;
        const bool on = lane < n_trials && lane < KM_MAX_TRIALS;
        const int64_t lo_j = on ? rec->cand_lo[lane] : 0, hi_j = on ? rec->cand_hi[lane] : 0;
        const double cx_j = on ? rec->cand_x[lane] : 0.0;
        const int best = km_best_wave(rec, n_trials, lane, closed);
        const int64_t blo = __shfl(lo_j, best, 64), bhi = __shfl(hi_j, best, 64);
        const double bc = __shfl(cx_j, best, 64);
        if (lane == 0) { s_lo = blo; s_hi = bhi; s_c = bc; }
    }
    KM_TG(16);
    __syncthreads();
    const int64_t lo = s_lo, hi = s_hi;
    // one chunk = one block of KM_CHUNK sorted positions (whole: its sum of closest distances is rewritten)
    const int64_t first = lo / KM_CHUNK;
    const int64_t chunks = hi > lo ? (hi - 1) / KM_CHUNK - first + 1 : 0;
    int64_t share = (chunks + GROUPS - 1) / GROUPS;             // workgroups that take part
    if (share > (int64_t)gridDim.x) share = gridDim.x;
    if ((int64_t)blockIdx.x >= share) return;
    // (straight atomics to the block sums for workgroups with a single chunk -- no clearing and scanning of 3 * nblocks
    // LDS words -- were measured SLOWER: 30 M values / 512 levels 32.5 -> 37.0 ms; a chunk's 2048 values share blocks
    // often enough for the LDS stage to save global atomics, which run at 23 G/s whatever their addresses)
    KM_TG(17);
    const double c = s_c, csq = __dmul_rn(c, c);
    const int64_t rounds = (chunks + share * GROUPS - 1) / (share * GROUPS);
    for (int64_t it = 0; it < rounds; ++it) {
        const int64_t ch = (it * share + blockIdx.x) * GROUPS + grp;
        const bool active = ch < chunks;
        const int64_t p0 = (first + (active ? ch : 0)) * KM_CHUNK;
        const int64_t p1 = p0 + KM_CHUNK < m ? p0 + KM_CHUNK : m;
        double left = 0.0;                                      // what stays: the block's new sum
        if (active) {
            double xv[KM_CHUNK / 256], dv[KM_CHUNK / 256];
            uint32_t iv[KM_CHUNK / 256];
#pragma unroll
            for (int u = 0; u < KM_CHUNK / 256; ++u) {
                const int64_t p = p0 + u * 256 + t;
                const int64_t q = p < p1 ? p : p1 - 1;
                xv[u] = xs[q];
                dv[u] = ds[q];
                iv[u] = perm[q];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < KM_CHUNK / 256; ++u) {
                const int64_t p = p0 + u * 256 + t;
                if (p >= p1) continue;
                const double d = dv[u];
                const double dj = km_sqdist(c, csq, xv[u]);
                if (p >= lo && p < hi && dj < d) {
                    ds[p] = dj;
                    left += dj;
                    double a, b, cc, aj, bj, cj;
                    km_split(L, d, a, b, cc);
                    km_split(L, dj, aj, bj, cj);
                    const i64 ua = -__double2ll_rn((a - aj) * L.sA), ub = -__double2ll_rn((b - bj) * L.sB),
                              uc = -__double2ll_rn((cc - cj) * L.sC);
                    const int blk = (int)(iv[u] >> block_shift);
                    if (ua) atomicAdd(&s_acc[3 * blk], (u64)ua);
                    if (ub) atomicAdd(&s_acc[3 * blk + 1], (u64)ub);
                    if (uc) atomicAdd(&s_acc[3 * blk + 2], (u64)uc);
                } else {
                    left += d;
                }
            }
        }
        // the group's sum in a fixed shape: wavefront butterflies, then its four wavefront totals in order
        left = grx_group_sum<64>(left);
        if ((threadIdx.x & 63) == 0) s_red[it & 1][threadIdx.x >> 6] = left;
        __syncthreads();
        if (t == 0 && active) {
            const double *r4 = &s_red[it & 1][grp * 4];
            sb[first + ch].sd = ((r4[0] + r4[1]) + r4[2]) + r4[3];
        }
    }
    __syncthreads();
    KM_TG(18);
    for (int i = threadIdx.x; i < 3 * nblocks; i += KM_UPDATE_THREADS) {
        const u64 v = s_acc[i];
        if (v) atomicAdd(reinterpret_cast<u64 *>(bacc) + 4 * (i / 3) + i % 3, v);
    }
    KM_TG(19);
}

// ---- few values (m <= 4096: the r x F factor of RolX, small graphs): the whole seeding in ONE workgroup -----------------
// The same procedure in the same exact arithmetic -- cumulative sum in index order, candidates by searchsorted, gains,
// first minimum with the 1e-12 tie rule, update -- with the values in registers (VPT consecutive indices per thread) and
// six workgroup barriers per seed instead of three launches (which cost ~45 us per seed whatever m is: 63 seeds of a
// 120-entry factor took 2.3 ms).  Every candidate's range is the whole input.
template <int VPT, int NT>
__global__ __launch_bounds__(1024) void km_seed_small_kernel(const double *__restrict__ v, int m, int64_t first,
                                                             const double *__restrict__ uniform, int n_trials, int k,
                                                             const KmState *__restrict__ st, double *__restrict__ seeds_x,
                                                             int64_t *__restrict__ seeds_id)
{
    __shared__ double s_x[1024 * VPT];
    __shared__ u64 s_w_lo[16], s_w_hi[16];
    __shared__ u64 s_gain[3][KM_MAX_TRIALS];
    __shared__ int s_cand[KM_MAX_TRIALS];
    __shared__ double s_cx[KM_MAX_TRIALS];
    __shared__ int s_best;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const KmLimb L = st->limb;
    const double mean = st->mean, c0 = st->c0, c0sq = __dmul_rn(c0, c0);
    double x[VPT], d[VPT];
#pragma unroll
    for (int q = 0; q < VPT; ++q) {
        const int i = tid * VPT + q;
        x[q] = i < m ? v[i] - mean : 0.0;
        s_x[i] = x[q];
        d[q] = i < m ? km_sqdist(c0, c0sq, x[q]) : 0.0;
    }
    if (tid == 0) { seeds_x[0] = c0; seeds_id[0] = first; }
    const int nwaves = (int)(blockDim.x >> 6);                  // (launched with just enough wavefronts for m values)
    double u = (lane < n_trials && k > 1) ? uniform[lane] : 0.0;   // lane j: the uniform of trial j
    for (int c = 1; c < k; ++c) {
        // cumulative sum of the closest distances in index order
        i128 inc[VPT];
        i128 lsum = 0;
#pragma unroll
        for (int q = 0; q < VPT; ++q) {
            lsum += (tid * VPT + q < m) ? km_quanta(L, d[q]) : (i128)0;
            inc[q] = lsum;
        }
        const i128 winc = km_wave_scan128(lsum, lane);
        if (lane == 63) { s_w_lo[wave] = (u64)winc; s_w_hi[wave] = (u64)(winc >> 64); }
        if (tid < KM_MAX_TRIALS) s_cand[tid] = 0x7FFFFFFF;
        if (tid < 3 * KM_MAX_TRIALS) (&s_gain[0][0])[tid] = 0;
        __syncthreads();
        i128 mine = winc - lsum, total = 0;
        for (int w = 0; w < nwaves; ++w) {
            const i128 wt = km_make128(s_w_lo[w], s_w_hi[w]);
            if (w < wave) mine += wt;
            total += wt;
        }
        const double potd = km_to_double(total);
        // candidates: first index whose cumulative sum reaches uniform * potential (lane j works out trial j's target)
        const i128 Rl = km_ceil128(u * potd);
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            if (j < n_trials) {
                const i128 R = km_make128(__shfl((u64)Rl, j, 64), __shfl((u64)(Rl >> 64), j, 64));
                if (mine < R && mine + lsum >= R) {             // (at most one thread per trial)
                    int hit = VPT - 1;
#pragma unroll
                    for (int q = VPT - 1; q >= 0; --q)
                        if (mine + inc[q] >= R) hit = q;
                    atomicMin(&s_cand[j], tid * VPT + hit);
                } else if (R == 0 && tid == 0) {
                    atomicMin(&s_cand[j], 0);
                }
            }
        }
        // (the next seed's uniforms while this one is worked on)
        const double un = (lane < n_trials && c + 1 < k) ? uniform[(size_t)c * n_trials + lane] : 0.0;
        __syncthreads();
        if (tid < n_trials) {
            const int idx = s_cand[tid] < m - 1 ? s_cand[tid] : m - 1;      // np.clip(candidate_ids, None, n - 1)
            s_cand[tid] = idx;
            s_cx[tid] = s_x[idx];
        }
        __syncthreads();
        // gains
        double cj[NT], ga[NT], gb[NT], gc[NT];
#pragma unroll
        for (int j = 0; j < NT; ++j) { cj[j] = j < n_trials ? s_cx[j] : 0.0; ga[j] = gb[j] = gc[j] = 0.0; }
#pragma unroll
        for (int q = 0; q < VPT; ++q) {
            if (tid * VPT + q < m && d[q] > 0.0) {
                double a, b, cc;
                km_split(L, d[q], a, b, cc);
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    if (j < n_trials) {
                        const double dj = km_sqdist(cj[j], __dmul_rn(cj[j], cj[j]), x[q]);
                        if (dj < d[q]) {
                            double aj, bj, cc2;
                            km_split(L, dj, aj, bj, cc2);
                            ga[j] += a - aj; gb[j] += b - bj; gc[j] += cc - cc2;
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            if (j < n_trials && __ballot(ga[j] != 0.0 || gb[j] != 0.0 || gc[j] != 0.0) != 0) {
                const i64 ua = km_wave_sum_i64(__double2ll_rn(ga[j] * L.sA));
                const i64 ub = km_wave_sum_i64(__double2ll_rn(gb[j] * L.sB));
                const i64 uc = km_wave_sum_i64(__double2ll_rn(gc[j] * L.sC));
                if (lane == 0) {
                    if (ua) atomicAdd(&s_gain[0][j], (u64)ua);
                    if (ub) atomicAdd(&s_gain[1][j], (u64)ub);
                    if (uc) atomicAdd(&s_gain[2][j], (u64)uc);
                }
            }
        }
        __syncthreads();
        // first minimum of potential - gain, potentials within 1e-12 tied (km_best)
        if (wave == 0) {
            double pd = 1.79769313486231570e308;
            if (lane < n_trials) pd = km_to_double(total - km_join((i64)s_gain[0][lane], (i64)s_gain[1][lane], (i64)s_gain[2][lane]));
            double lowest = pd;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                const double o = __shfl_xor(lowest, off, 64);
                lowest = o < lowest ? o : lowest;
            }
            const uint64_t tied = __ballot(lane < n_trials && !(pd > lowest + 1e-12 * lowest));
            const int best = __ffsll((long long)tied) - 1;
            if (lane == 0) {
                s_best = best;
                seeds_x[c] = s_cx[best];
                seeds_id[c] = s_cand[best];
            }
        }
        __syncthreads();
        const double cb = s_cx[s_best], cbsq = __dmul_rn(cb, cb);
#pragma unroll
        for (int q = 0; q < VPT; ++q) {
            const double dj = km_sqdist(cb, cbsq, x[q]);
            d[q] = dj < d[q] ? dj : d[q];
        }
        u = un;
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

// exclusive scan of the tile sums in place: a contiguous chunk of tiles per thread, the chunk totals scanned across
// the workgroup (fixed shape)
__global__ __launch_bounds__(1024) void km_scan_tiles_kernel(double *__restrict__ tsum, int64_t ntiles)
{
    __shared__ double s_w[16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t chunk = (ntiles + 1023) / 1024;
    const int64_t t0 = (int64_t)threadIdx.x * chunk, t1 = (t0 + chunk < ntiles) ? t0 + chunk : ntiles;
    double local = 0.0;
    for (int64_t t = t0; t < t1; ++t) local += tsum[t];
    double inc = local;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const double y = __shfl_up(inc, off, 64);
        if (lane >= off) inc += y;
    }
    if (lane == 63) s_w[wave] = inc;
    __syncthreads();
    double run = 0.0;
    for (int w = 0; w < wave; ++w) run += s_w[w];
    run += inc - local;
    for (int64_t t = t0; t < t1; ++t) { const double v = tsum[t]; tsum[t] = run; run += v; }
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

// The Lloyd kernel's view of the sorted values from the top (km_sorted_init's block starts): tops[j] = xs[j * step] for
// j < n_tops in LDS (every block start when there are at most KM_LLOYD_TOPS of them, every s2-th otherwise), so that a
// search for a boundary reads memory only inside [tops interval] -- 11 to 15 dependent loads instead of 23 to 25.
constexpr int KM_LLOYD_TOPS = 4096;
struct KmTopLds { const double *tops; int n_tops; int64_t step; };     // n_tops == 0: no index (few values)

// first index with xs > b, like km_upper
__device__ __forceinline__ int64_t km_upper_top(const double *__restrict__ xs, int64_t m, double b, const KmTopLds &T)
{
    if (T.n_tops == 0) return km_upper(xs, m, b);
    int lo = 0, up = T.n_tops;                                  // tops <= b: the answer lies behind the last of them
    while (lo < up) { const int mid = (lo + up) >> 1; if (T.tops[mid] <= b) lo = mid + 1; else up = mid; }
    if (lo == 0) return 0;                                      // xs[0] > b
    int64_t a = (int64_t)(lo - 1) * T.step + 1, e = (int64_t)lo * T.step < m ? (int64_t)lo * T.step : m;
    while (a < e) { const int64_t mid = (a + e) >> 1; if (xs[mid] <= b) a = mid + 1; else e = mid; }
    return a;
}

// number of sorted values that go to centres at or left of `cl` when the next distinct centre is `cr`
__device__ int64_t km_boundary(const double *__restrict__ xs, int64_t m, double cl, int idl, double cr, int idr,
                               const KmTopLds &T)
{
    int64_t h = km_upper_top(xs, m, 0.5 * (cl + cr), T);
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

// (s_c: LDS room for k centres when k <= KM_LLOYD_C_LDS, else nullptr -- the ranking reads every centre k times)
constexpr int KM_LLOYD_C_LDS = 2048;
__device__ void km_e_step(const double *__restrict__ xs, int64_t m, int k, const KmLloydBufs &B, const KmTopLds &T, double *s_c)
{
    const int t = threadIdx.x, nt = blockDim.x;
    const double *cc = B.c;
    if (s_c) {
        for (int j = t; j < k; j += nt) s_c[j] = B.c[j];
        __syncthreads();
        cc = s_c;
    }
    // sorted order of the centres (ties by centre id): rank by counting
    for (int j = t; j < k; j += nt) {
        const double cj = cc[j];
        int rank = 0;
        for (int i = 0; i < k; ++i) {
            const double ci = cc[i];
            rank += (ci < cj) || (ci == cj && i < j);
        }
        B.ord[rank] = j;
    }
    __syncthreads();
    // hiS[p]: values assigned to sorted positions <= p.  Equal centres: the smallest id takes the values.
    for (int p = t; p < k; p += nt) {
        const double cp = cc[B.ord[p]];
        int g = p;                                              // first member of p's group of equal centres
        while (g > 0 && cc[B.ord[g - 1]] == cp) --g;
        int nx = p + 1;                                         // next distinct centre
        while (nx < k && cc[B.ord[nx]] == cp) ++nx;
        B.hiS[p] = (nx >= k) ? m : km_boundary(xs, m, cp, B.ord[g], cc[B.ord[nx]], B.ord[nx], T);
    }
    __syncthreads();
    for (int p = t; p < k; p += nt) {
        const int j = B.ord[p];
        const int64_t h = B.hiS[p];
        int64_t l;
        if (p == 0) l = 0;
        else if (cc[B.ord[p - 1]] == cc[j]) l = h;            // not the first of its group: empty
        else l = B.hiS[p - 1];
        B.lo[j] = l;
        B.hi[j] = h;
    }
    __syncthreads();
}

__global__ __launch_bounds__(1024) void km_lloyd_kernel(const double *__restrict__ xs, const double *__restrict__ P,
                                                        int64_t m, int k, int max_iter, const KmState *__restrict__ st,
                                                        const double *__restrict__ seeds_x, KmLloydBufs B,
                                                        int32_t *__restrict__ info, const double *__restrict__ xtop,
                                                        int64_t nsb, const double *__restrict__ xtop2, int n2, int s2)
{
    __shared__ double s_red[16];
    __shared__ double s_tops[KM_LLOYD_TOPS], s_cbuf[KM_LLOYD_C_LDS];
    const int t = threadIdx.x, nt = blockDim.x;
    // block starts of the sorted values (km_sorted_init_kernel) into LDS: all of them, or the coarser level
    KmTopLds T;
    T.tops = s_tops; T.n_tops = 0; T.step = 0;
    if (xtop && nsb > 0 && nsb <= KM_LLOYD_TOPS) {
        for (int i = t; i < (int)nsb; i += nt) s_tops[i] = xtop[i];
        T.n_tops = (int)nsb; T.step = KM_CHUNK;
    } else if (xtop2 && n2 > 0 && n2 <= KM_LLOYD_TOPS) {
        for (int i = t; i < n2; i += nt) s_tops[i] = xtop2[i];
        T.n_tops = n2; T.step = (int64_t)s2 * KM_CHUNK;
    }
    double *s_c = k <= KM_LLOYD_C_LDS ? s_cbuf : nullptr;
    for (int j = t; j < k; j += nt) { B.c[j] = seeds_x[j]; B.plo[j] = -1; B.phi[j] = -1; }
    __syncthreads();
    const double tol = st->tol;
    int n_iter = 0;
    bool strict = false;
    for (int it = 0; it < max_iter; ++it) {
        n_iter = it + 1;
        km_e_step(xs, m, k, B, T, s_c);
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
    if (!strict) km_e_step(xs, m, k, B, T, s_c);                // a last E step so that labels match the centres
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
        info[3] = k > 1 ? st->faults : 0;
#ifdef KM_DBG_TIMING
        for (int q = 1; q < 10; ++q) printf("prep phase %d: %lld ns\n", q, (st->dbg[q] - st->dbg[q - 1]) * 10);
        printf("prep end -> update start: %lld ns\n", (st->dbg[15] - st->dbg[9]) * 10);
        for (int q = 16; q < 21; ++q) printf("update phase %d: %lld ns\n", q, (st->dbg[q] - st->dbg[q - 1]) * 10);
        printf("seed to seed: %lld ns\n", (st->dbg[20] - st->dbg[0]) * 10);
#endif
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
    int block_shift, nblocks;                                  // index blocks of the cumulative sum
    size_t off_state, off_x, off_ds, off_bacc, off_tsum, off_part, off_seedx, off_seedid, off_sorted, off_uniform,
        off_xs, off_perm, off_rank, off_sb, off_top, off_top2, off_P, off_lloyd, off_sort, total;
    size_t sorted_ld;
};

KmPlan km_plan(int64_t m, int k)
{
    KmPlan p;
    p.ntiles = grx_ceil_div(m, KM_TILE);
    const int64_t want = grx_ceil_div(m, 256 * 8);
    p.nb = (int)(want > 2048 ? 2048 : (want < 1 ? 1 : want));
    p.block_shift = KM_MIN_BLOCK_SHIFT;
    while (grx_ceil_div(m, (int64_t)1 << p.block_shift) > KM_MAX_BLOCKS) ++p.block_shift;
    p.nblocks = (int)grx_ceil_div(m, (int64_t)1 << p.block_shift);
    p.sorted_ld = grx_align_up((size_t)(k + 1) * 8, 256) / 8;
    size_t o = 0;
    auto take = [&](size_t bytes) { size_t at = o; o += grx_align_up(bytes, 256); return at; };
    p.off_state = take(sizeof(KmState));
    p.off_x = take((size_t)m * 8);
    p.off_ds = take((size_t)m * 8);
    p.off_bacc = take((size_t)(KM_MAX_BLOCKS + 32) * 32);
    p.off_tsum = take((size_t)p.ntiles * 8);
    p.off_part = take((size_t)3 * p.nb * 8);
    p.off_seedx = take((size_t)k * 8);
    p.off_seedid = take((size_t)k * 8);
    p.off_sorted = take(2 * p.sorted_ld * 8);
    p.off_uniform = take((size_t)(k > 1 ? k - 1 : 1) * KM_MAX_TRIALS * 8);
    p.off_xs = take((size_t)m * 8);
    p.off_perm = take((size_t)m * 4);
    p.off_rank = take((size_t)m * 4);
    p.off_sb = take((size_t)grx_ceil_div(m, KM_CHUNK) * sizeof(KmSorted));
    p.off_top = take((size_t)grx_ceil_div(m, KM_CHUNK) * 8);
    p.off_top2 = take((size_t)KM_TOP2 * 8);
    p.off_P = take((size_t)(m + 1) * 8);
    p.off_lloyd = take((size_t)k * 8 * 16);
    p.off_sort = take(grx_internal_sort_pairs_workspace_bytes(m));
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
    double *ds = reinterpret_cast<double *>(ws + p.off_ds);
    i64 *bacc = reinterpret_cast<i64 *>(ws + p.off_bacc);
    double *tsum = reinterpret_cast<double *>(ws + p.off_tsum);
    double *part = reinterpret_cast<double *>(ws + p.off_part);
    double *seeds_x = reinterpret_cast<double *>(ws + p.off_seedx);
    int64_t *seeds_id = reinterpret_cast<int64_t *>(ws + p.off_seedid);
    double *sorted2 = reinterpret_cast<double *>(ws + p.off_sorted);
    double *d_uniform = reinterpret_cast<double *>(ws + p.off_uniform);
    double *xs = reinterpret_cast<double *>(ws + p.off_xs);
    uint32_t *perm = reinterpret_cast<uint32_t *>(ws + p.off_perm);
    uint32_t *rank = reinterpret_cast<uint32_t *>(ws + p.off_rank);
    KmSorted *sb = reinterpret_cast<KmSorted *>(ws + p.off_sb);
    double *P = reinterpret_cast<double *>(ws + p.off_P);
    GRX_PROF(GRX_K_QUANT, st);
    if (k > 1)
        GRX_CHECK_HIP(hipMemcpyAsync(d_uniform, h_uniform, (size_t)(k - 1) * n_trials * 8, hipMemcpyHostToDevice, st));
    GRX_CHECK_HIP(hipMemsetAsync(bacc, 0, (size_t)(KM_MAX_BLOCKS + 32) * 32, st));
    // mean and tolerance (KMeans.fit: X -= X.mean(axis=0); tol = mean(var(X, axis=0)) * 1e-4)
    km_sum_kernel<<<p.nb, 256, 0, st>>>(d_values, m, state, 0, part);
    km_moment_final_kernel<<<1, 256, 0, st>>>(part, p.nb, m, 0, rel_tol, d_values, first_seed, state);
    km_sum_kernel<<<p.nb, 256, 0, st>>>(d_values, m, state, 1, part);
    km_moment_final_kernel<<<1, 256, 0, st>>>(part, p.nb, m, 1, rel_tol, d_values, first_seed, state);
    // (the sorted list of seeds with n entries lives in buffer n & 1: the first seed goes to buffer 1)
    km_init_kernel<<<(int)p.ntiles, 256, 0, st>>>(d_values, m, first_seed, state, x, bacc, p.block_shift, seeds_x, seeds_id,
                                                  sorted2 + p.sorted_ld);
    GRX_LAUNCH_CHECK();
    // the values in ascending order with the index each came from (the Lloyd iterations need the order as well)
    int rc = grx_internal_sort_pairs(m, x, xs, perm, ws + p.off_sort, st);
    if (rc != GRX_OK) return rc;
    const int64_t want = grx_ceil_div(m, 256 * 4);
    const int stream_grid = (int)(want > 2048 ? 2048 : want);
    // GRX_KMEANS_FULL_RANGE=1: every candidate's range is [0, m) -- sklearn's own O(m k) formulation in the same exact
    // arithmetic; the ranges are supersets of what can change, so both give the same bits (tests/test_gpu_encode.py).
    // GRX_KMEANS_SMALL=0: few values take the many-launch path as well (the same test: same bits again).
    static const int full_range = [] { const char *e = std::getenv("GRX_KMEANS_FULL_RANGE"); return (e && *e == '1') ? 1 : 0; }();
    static const int small_ok = [] { const char *e = std::getenv("GRX_KMEANS_SMALL"); return (e && *e == '0') ? 0 : 1; }();
    // (block starts of the sorted values for the Lloyd kernel's searches: built by the many-launch seeding only)
    const double *l_xtop = nullptr, *l_xtop2 = nullptr;
    int64_t l_nsb = 0;
    int l_n2 = 0, l_s2 = 0;
    if (k > 1 && m <= KM_SMALL_M && small_ok) {
        const int vpt = m <= 1024 ? 1 : KM_SMALL_M / 1024;
        const int threads = (int)grx_align_up((size_t)grx_ceil_div(m, vpt), 64);
        if (vpt == 1 && n_trials <= 8)
            km_seed_small_kernel<1, 8><<<1, threads, 0, st>>>(d_values, (int)m, first_seed, d_uniform, n_trials, k, state, seeds_x, seeds_id);
        else if (vpt == 1)
            km_seed_small_kernel<1, KM_MAX_TRIALS><<<1, threads, 0, st>>>(d_values, (int)m, first_seed, d_uniform, n_trials, k, state,
                                                                          seeds_x, seeds_id);
        else if (n_trials <= 8)
            km_seed_small_kernel<KM_SMALL_M / 1024, 8><<<1, threads, 0, st>>>(d_values, (int)m, first_seed, d_uniform, n_trials, k, state,
                                                                              seeds_x, seeds_id);
        else
            km_seed_small_kernel<KM_SMALL_M / 1024, KM_MAX_TRIALS><<<1, threads, 0, st>>>(d_values, (int)m, first_seed, d_uniform, n_trials,
                                                                                          k, state, seeds_x, seeds_id);
    } else if (k > 1) {
        // GRX_KMEANS_GAIN_PASS=1 (and the full ranges): every candidate's gain by a pass over its range in exact integer
        // arithmetic (km_gain_kernel) instead of the closed form over sorted blocks inside the pick (km_pick_tail, D):
        // same seeds unless two potentials agree to 1e-12 (tests/test_gpu_encode.py compares the modes)
        static const int gain_pass = [] { const char *e = std::getenv("GRX_KMEANS_GAIN_PASS"); return (e && *e == '1') ? 1 : 0; }();
        const int closed = (full_range || gain_pass) ? 0 : 1;
        // GRX_KMEANS_SLOW_PICK=1: the positions of every range by searches over all the sorted values (what the pick falls
        // back to when a range's undecided band leaves its end blocks) -- same positions, the same test again
        static const int slow_pick = [] { const char *e = std::getenv("GRX_KMEANS_SLOW_PICK"); return (e && *e == '1') ? 1 : 0; }();
        const int64_t max_chunks = grx_ceil_div(m, KM_CHUNK);
        KmTop top;
        top.xtop = reinterpret_cast<double *>(ws + p.off_top);
        top.xtop2 = reinterpret_cast<double *>(ws + p.off_top2);
        top.nsb = max_chunks;
        top.s2 = (int)grx_ceil_div(max_chunks, KM_TOP2);
        top.n2 = (int)grx_ceil_div(max_chunks, top.s2);
        km_sorted_init_kernel<<<(int)max_chunks, 256, 0, st>>>(xs, perm, m, state, ds, rank, sb, top);
        l_xtop = top.xtop; l_xtop2 = top.xtop2; l_nsb = top.nsb; l_n2 = top.n2; l_s2 = top.s2;
        const int range_grid = (int)(max_chunks < KM_RANGE_GRID ? max_chunks : KM_RANGE_GRID);
        static const int update_grid_max = [] { const char *e = std::getenv("GRX_KMEANS_UPDATE_GRID"); return e ? atoi(e) : KM_UPDATE_GRID; }();
        const int update_grid = (int)(max_chunks < update_grid_max ? max_chunks : update_grid_max);
        // The update with seed c inside its pick kernel (GRX_KMEANS_MERGE=0: never) once the ranges are short: a range
        // holds about m / c values, the launch has 16 * n_trials workgroups, and up to GRX_KMEANS_MERGE chunks (default 2)
        // per workgroup still beat km_update_kernel's launch
        static const double merge_chunks = [] { const char *e = std::getenv("GRX_KMEANS_MERGE"); return e ? atof(e) : 2.0; }();
        const double merge_from = merge_chunks > 0.0 ? (double)m / ((double)KM_CHUNK * merge_chunks * KM_SUB * n_trials) : 1e300;
        // GRX_KMEANS_LOSE_A_SUM=<seed>: at that seed one workgroup of the pick withholds its sum (tests: the others' wait
        // must expire, the run must end at once with fault bit 4, nothing may hang)
        static const int lose_at = [] { const char *e = std::getenv("GRX_KMEANS_LOSE_A_SUM"); return e ? atoi(e) : -1; }();
        for (int c = 1; c < k; ++c) {
            const int merge = (closed && c < k - 1 && (double)c >= merge_from) ? 1 : 0;
            km_prep_kernel<<<dim3(KM_SUB, n_trials), 256, 0, st>>>(xs, ds, rank, perm, m, bacc, p.nblocks, p.block_shift,
                                                                   d_uniform + (size_t)(c - 1) * n_trials, n_trials, c, c >= 2, 1,
                                                                   full_range, closed, slow_pick, merge, c == lose_at ? 1 : 0, sb, top,
                                                                   state, seeds_x, seeds_id, sorted2, (int)p.sorted_ld);
            if (!closed) {
                if (n_trials <= 8) km_gain_kernel<8><<<range_grid, 256, 0, st>>>(xs, ds, state, c, n_trials);
                else km_gain_kernel<KM_MAX_TRIALS><<<range_grid, 256, 0, st>>>(xs, ds, state, c, n_trials);
            }
            if (c < k - 1 && !merge)                               // the distances to the last seed are never needed
                km_update_kernel<<<update_grid, KM_UPDATE_THREADS, 0, st>>>(xs, ds, perm, bacc, p.nblocks, p.block_shift, m, sb, state,
                                                                            c, n_trials, closed);
        }
        km_prep_kernel<<<dim3(1, 1), 256, 0, st>>>(xs, ds, rank, perm, m, bacc, p.nblocks, p.block_shift, d_uniform, n_trials, k, 1,
                                                   0, full_range, closed, slow_pick, 0, 0, sb, top, state, seeds_x, seeds_id, sorted2,
                                                   (int)p.sorted_ld);
    }
    GRX_LAUNCH_CHECK();
    // Lloyd on the sorted values
    km_tile_sums_kernel<<<(int)p.ntiles, 256, 0, st>>>(xs, m, tsum);
    km_scan_tiles_kernel<<<1, 1024, 0, st>>>(tsum, p.ntiles);
    km_prefix_kernel<<<(int)p.ntiles, 256, 0, st>>>(xs, m, tsum, P);
    KmLloydBufs B;
    double *lb = reinterpret_cast<double *>(ws + p.off_lloyd);
    B.c = lb; B.cnew = lb + k; B.sums = lb + 2 * (size_t)k; B.counts = lb + 3 * (size_t)k; B.cfinal = lb + 4 * (size_t)k;
    B.maxval = lb + 5 * (size_t)k;
    B.hiS = reinterpret_cast<int64_t *>(lb + 6 * (size_t)k);
    B.lo = B.hiS + k; B.hi = B.lo + k; B.plo = B.hi + k; B.phi = B.plo + k; B.rl = B.phi + k; B.rh = B.rl + k;
    B.ord = reinterpret_cast<int32_t *>(B.rh + k);
    km_lloyd_kernel<<<1, 1024, 0, st>>>(xs, P, m, k, max_iter, state, seeds_x, B, d_info, l_xtop, l_nsb, l_xtop2, l_n2, l_s2);
    km_assign_kernel<<<stream_grid, 256, 0, st>>>(d_values, m, k, state, B.maxval, B.cfinal, B.ord, d_quantized, d_centers);
    GRX_LAUNCH_CHECK();
    return GRX_OK;
}

}  // extern "C"
