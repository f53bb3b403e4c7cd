// grx_quant.hip -- 1-D Lloyd-Max quantiser for RolX `encode` (graphrole/roles/factor.py:29-49).
//
// The reference quantises all entries of a factor matrix with sklearn KMeans(n_clusters=n_bins,
// random_state=1) on the flattened values ("Lloyd-Max quantizer which can be computed using
// kmeans", factor.py:38-39).  k-means++ seeding from a host RNG cannot be reproduced on a device,
// so parity is by property: <= n_bins distinct output values, every value is
// replaced by the mean of its cell, cells are nearest-centre cells, and the quantisation error is
// not worse than sklearn's.  To meet the last point deterministically the start is not random:
//   1. sort the values (batched radix sort of grx_prune.hip), prefix sums of s, s^2 and of
//      cbrt(gap)^2 (Panter-Dite: optimal cell density ~ pdf^(1/3))
//   2. cut the sorted values into <= 1024 micro-cells of equal cbrt-density mass (or one value per
//      cell when there are few values) and solve the k-cluster problem on the cells EXACTLY by
//      dynamic programming (one workgroup; O(k * cells^2))
//   3. refine with Lloyd iterations on the full sorted array (cluster sums from the prefix sums:
//      O(k log m) per iteration) until no boundary moves
//   4. map every input value to its cell centre.
// With <= 1024 values step 2 is the exact optimum of the k-means objective (<= any k-means run).
#include "grx_common.h"

int grx_internal_sort_columns(int64_t n, int ncols, const double *cols, int64_t ld, double *out, int64_t out_ld,
                              void *workspace, hipStream_t st);
extern "C" size_t grx_sort_workspace_bytes(int64_t n, int ncols);

namespace {

constexpr int Q_TILE = 2048;            // elements per scan workgroup (256 threads x 8)
constexpr int Q_CELLS = 1024;           // micro-cells of the DP stage
constexpr int Q_MAX_BINS = 256;          // exact-DP start + one-thread-per-cluster Lloyd kernel
constexpr int QC_BIG = 8192;             // exact DP with one cell per value up to this many values
constexpr int Q_BIG_MAX_BINS = 65536;    // above Q_MAX_BINS: companding start, strided Lloyd kernel, tables in global memory

// ---- three fused prefix sums over the sorted values: s, s^2, cbrt(s[i+1]-s[i])^2 -----------
__device__ __forceinline__ void q_terms(const double *__restrict__ s, int64_t m, int64_t i, double &a, double &b,
                                        double &c)
{
    const double x = s[i];
    a = x;
    b = x * x;
    const double gap = (i + 1 < m) ? (s[i + 1] - x) : 0.0;
    const double r = cbrt(gap);
    c = r * r;
}

__global__ __launch_bounds__(256) void q_tile_sums_kernel(const double *__restrict__ s, int64_t m,
                                                          double *__restrict__ tsum)
{
    __shared__ double red[3][4];
    const int64_t base = (int64_t)blockIdx.x * Q_TILE + threadIdx.x * 8;
    double a = 0.0, b = 0.0, c = 0.0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        if (base + j < m) {
            double x, y, z;
            q_terms(s, m, base + j, x, y, z);
            a += x; b += y; c += z;
        }
    }
    a = grx_group_sum<64>(a); b = grx_group_sum<64>(b); c = grx_group_sum<64>(c);
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = a; red[1][threadIdx.x >> 6] = b; red[2][threadIdx.x >> 6] = c; }
    __syncthreads();
    if (threadIdx.x < 3)
        tsum[(size_t)blockIdx.x * 3 + threadIdx.x] =
            ((red[threadIdx.x][0] + red[threadIdx.x][1]) + red[threadIdx.x][2]) + red[threadIdx.x][3];
}

// exclusive scan of the tile sums, one workgroup, sequential per quantity (ntiles is small)
__global__ __launch_bounds__(64) void q_scan_tiles_kernel(double *__restrict__ tsum, int64_t ntiles)
{
    if (threadIdx.x < 3) {
        double run = 0.0;
        for (int64_t t = 0; t < ntiles; ++t) {
            const double v = tsum[(size_t)t * 3 + threadIdx.x];
            tsum[(size_t)t * 3 + threadIdx.x] = run;
            run += v;
        }
    }
}

// P[i] = sum_{j<i} term_j for i in [0, m]; three arrays of m+1 entries
__global__ __launch_bounds__(256) void q_prefix_kernel(const double *__restrict__ s, int64_t m,
                                                       const double *__restrict__ tsum, double *__restrict__ P,
                                                       double *__restrict__ P2, double *__restrict__ Gp)
{
    __shared__ double wtot[3][4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t base = (int64_t)blockIdx.x * Q_TILE + threadIdx.x * 8;
    double ta[8], tb[8], tc[8];
    double a = 0.0, b = 0.0, c = 0.0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        ta[j] = tb[j] = tc[j] = 0.0;
        if (base + j < m) q_terms(s, m, base + j, ta[j], tb[j], tc[j]);
        a += ta[j]; b += tb[j]; c += tc[j];
    }
    // inclusive wave scan of the per-thread sums
    double ia = a, ib = b, ic = c;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const double xa = __shfl_up(ia, off, 64), xb = __shfl_up(ib, off, 64), xc = __shfl_up(ic, off, 64);
        if (lane >= off) { ia += xa; ib += xb; ic += xc; }
    }
    if (lane == 63) { wtot[0][wave] = ia; wtot[1][wave] = ib; wtot[2][wave] = ic; }
    __syncthreads();
    double oa = tsum[(size_t)blockIdx.x * 3 + 0], ob = tsum[(size_t)blockIdx.x * 3 + 1],
           oc = tsum[(size_t)blockIdx.x * 3 + 2];
    for (int w = 0; w < wave; ++w) { oa += wtot[0][w]; ob += wtot[1][w]; oc += wtot[2][w]; }
    oa += ia - a; ob += ib - b; oc += ic - c;          // exclusive prefix of this thread
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        if (base + j < m) {
            P[base + j] = oa; P2[base + j] = ob; Gp[base + j] = oc;
            oa += ta[j]; ob += tb[j]; oc += tc[j];
            if (base + j + 1 == m) { P[m] = oa; P2[m] = ob; Gp[m] = oc; }
        }
    }
}

// ---- micro-cells --------------------------------------------------------------------------------
// ce[0..nb]: element index of the cell edges (strictly increasing, ce[0] = 0, ce[nb] = m)
__global__ __launch_bounds__(Q_CELLS) void q_cells_kernel(const double *__restrict__ Gp, int64_t m, int want_cells,
                                                          int64_t *__restrict__ ce, int *__restrict__ nb_out)
{
    __shared__ int64_t raw[Q_CELLS + 1];
    const int t = threadIdx.x;
    if (m <= want_cells) {
        for (int64_t i = t; i <= m; i += blockDim.x) ce[i] = i;
        if (t == 0) *nb_out = (int)m;
        return;
    }
    const double total = Gp[m];
    for (int b = t; b <= want_cells; b += blockDim.x) {
        int64_t e;
        if (b == 0) e = 0;
        else if (b == want_cells) e = m;
        else {
            const double target = total * (double)b / (double)want_cells;
            int64_t lo = 0, hi = m;                    // first i with Gp[i] >= target
            while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (Gp[mid] < target) lo = mid + 1; else hi = mid; }
            e = lo < 1 ? 1 : (lo > m - 1 ? m - 1 : lo);
        }
        raw[b] = e;
    }
    __syncthreads();
    if (t == 0) {                                       // strictly increasing, duplicates removed
        int nb = 0;
        ce[0] = 0;
        for (int b = 1; b <= want_cells; ++b)
            if (raw[b] > ce[nb]) ce[++nb] = raw[b];
        if (ce[nb] != m) ce[++nb] = m;
        *nb_out = nb;
    }
}

// ---- exact k-clustering of the cells by dynamic programming -------------------------------------
// D_j[i] = min_{j <= mm < i} D_{j-1}[mm] + cost(cells mm..i-1);  arg[j][i] = minimiser (smallest
// mm on ties).  One launch per layer j, one wavefront per i (lanes stride over mm, fixed-order
// argmin butterfly), so a layer is spread over the whole chip instead of one workgroup.
__global__ __launch_bounds__(256) void q_dp_init_kernel(const int64_t *__restrict__ ce, const int *__restrict__ nb_ptr,
                                                        const double *__restrict__ P, const double *__restrict__ P2,
                                                        double *__restrict__ cell, double *__restrict__ D0, int cs)
{
    const int nb = *nb_ptr;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i <= nb; i += gridDim.x * blockDim.x) {
        const int64_t e = ce[i];
        cell[i] = P[e];
        cell[cs + i] = P2[e];
        cell[2 * cs + i] = (double)e;
        D0[i] = (i == 0) ? 0.0 : 1e300;
    }
}

__global__ __launch_bounds__(256) void q_dp_layer_kernel(const int *__restrict__ nb_ptr, int j, int k,
                                                         const double *__restrict__ cell,
                                                         const double *__restrict__ Dprev, double *__restrict__ Dcur,
                                                         int32_t *__restrict__ arg, int cs)
{
    const int nb = *nb_ptr;
    if (j >= (k < nb ? k : nb)) return;                 // surplus layers (more bins than cells)
    const double *cp = cell, *cp2 = cell + cs, *cn = cell + 2 * cs;
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);  // one wavefront per i
    if (i > nb) return;
    double best = 1e300;
    int bm = 0x7fffffff;
    if (i >= j + 1) {
        const double pi = cp[i], p2i = cp2[i], ni = cn[i];
        for (int mm = j + lane; mm < i; mm += 64) {
            const double dprev = Dprev[mm];
            if (dprev >= 1e300) continue;
            const double n = ni - cn[mm], sm = pi - cp[mm];
            double cst = (p2i - cp2[mm]) - sm * sm / n;
            if (cst < 0.0) cst = 0.0;
            const double v = dprev + cst;
            if (v < best) { best = v; bm = mm; }          // ascending mm per lane: first minimum kept
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const double ov = __shfl_xor(best, off, 64);
        const int om = __shfl_xor(bm, off, 64);
        if (ov < best || (ov == best && om < bm)) { best = ov; bm = om; }
    }
    if (lane == 0) {
        Dcur[i] = (i == 0) ? 1e300 : best;
        arg[(size_t)j * cs + i] = (bm == 0x7fffffff) ? j : bm;
    }
}

__global__ void q_dp_backtrack_kernel(const int64_t *__restrict__ ce, const int *__restrict__ nb_ptr, int k,
                                      const int32_t *__restrict__ arg, int64_t *__restrict__ edges, int cs)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const int nb = *nb_ptr;
    const int kk = k < nb ? k : nb;
    int i = nb;
    edges[kk] = ce[nb];
    for (int j = kk - 1; j >= 0; --j) {
        i = arg[(size_t)j * cs + i];
        edges[j] = ce[i];
    }
    for (int j = kk + 1; j <= k; ++j) edges[j] = ce[nb];      // surplus clusters stay empty
}

// ---- Lloyd refinement on the sorted array ------------------------------------------------------
// thread j owns cluster j.  hi[j] = number of sorted values in clusters 0..j.
__global__ __launch_bounds__(Q_MAX_BINS) void q_lloyd_kernel(const double *__restrict__ s, int64_t m,
                                                             const double *__restrict__ P, int k, int max_iter,
                                                             const int64_t *__restrict__ edges,
                                                             double *__restrict__ centers,
                                                             double *__restrict__ bounds, int32_t *__restrict__ info)
{
    __shared__ double c[Q_MAX_BINS];
    __shared__ int64_t hi[Q_MAX_BINS + 1];
    __shared__ int live[Q_MAX_BINS];
    const int j = threadIdx.x;
    if (j == 0) hi[0] = 0;
    if (j < k) {
        const int64_t a = edges[j], b = edges[j + 1];
        hi[j + 1] = b;
        live[j] = b > a;
        c[j] = (b > a) ? (P[b] - P[a]) / (double)(b - a) : 0.0;
    }
    __syncthreads();
    // empty clusters (fewer distinct cells than bins) inherit the centre below them
    if (j < k && !live[j]) { int q = j; while (q > 0 && !live[q]) --q; c[j] = c[q]; }
    __syncthreads();
    int it = 0;
    for (; it < max_iter; ++it) {
        int64_t nh = m;
        if (j < k - 1) {
            const double bnd = 0.5 * (c[j] + c[j + 1]);
            int64_t lo = 0, up = m;                     // first index with s > bnd  (upper_bound)
            while (lo < up) { const int64_t mid = (lo + up) >> 1; if (s[mid] <= bnd) lo = mid + 1; else up = mid; }
            nh = lo;
        }
        const int changed = (j < k) && (nh != hi[j + 1]);
        const int any = __syncthreads_or(changed);
        if (!any) break;
        if (j < k) hi[j + 1] = nh;
        __syncthreads();
        if (j < k) {
            const int64_t a = hi[j], b = hi[j + 1];
            if (b > a) c[j] = (P[b] - P[a]) / (double)(b - a);
        }
        __syncthreads();
    }
    if (j < k) centers[j] = c[j];
    if (j < k - 1) bounds[j] = 0.5 * (c[j] + c[j + 1]);
    if (j == 0) {
        info[0] = it;
        int nonempty = 0, distinct = 0;
        double last = 0.0;
        for (int q = 0; q < k; ++q) {
            if (hi[q + 1] > hi[q]) {
                ++nonempty;
                if (distinct == 0 || c[q] != last) { ++distinct; last = c[q]; }
            }
        }
        info[1] = nonempty;
        info[2] = distinct;
    }
}

__global__ __launch_bounds__(256) void q_assign_kernel(const double *__restrict__ x, int64_t m, int k,
                                                       const double *__restrict__ centers,
                                                       const double *__restrict__ bounds, double *__restrict__ out)
{
    __shared__ double c[Q_MAX_BINS], b[Q_MAX_BINS];
    if (threadIdx.x < k) c[threadIdx.x] = centers[threadIdx.x];
    if (threadIdx.x < k - 1) b[threadIdx.x] = bounds[threadIdx.x];
    __syncthreads();
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < m; i += stride) {
        const double v = x[i];
        int lo = 0, up = k - 1;                         // first boundary >= v  -> cluster index
        while (lo < up) { const int mid = (lo + up) >> 1; if (b[mid] < v) lo = mid + 1; else up = mid; }
        out[i] = c[lo];
    }
}

// ---- more than Q_MAX_BINS levels (n_bits = int(log2(n_roles * min(shape))) reaches 9-12 bits on wide feature
// tables, roles/extract.py:72) ---------------------------------------------------------------------------------
// The exact DP over micro-cells costs O(k cells^2) and its tables live in LDS; with this many levels the
// high-resolution approximation is already close to optimal, so the start is the companding partition itself --
// k cells of equal cbrt-density mass (Panter-Dite) -- refined by the same Lloyd iterations.  One workgroup,
// clusters strided over its threads, all tables in global memory.
__global__ __launch_bounds__(1024) void q_cells_big_kernel(const double *__restrict__ Gp, int64_t m, int k,
                                                           int64_t *__restrict__ raw, int64_t *__restrict__ edges)
{
    const int t = threadIdx.x;
    const double total = Gp[m];
    for (int b = t; b <= k; b += blockDim.x) {
        int64_t e;
        if (b == 0) e = 0;
        else if (b == k) e = m;
        else if (m <= k) e = b < m ? b : m;               // no more values than levels: one value per cell
        else {
            const double target = total * (double)b / (double)k;
            int64_t lo = 0, hi = m;                       // first i with Gp[i] >= target
            while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (Gp[mid] < target) lo = mid + 1; else hi = mid; }
            e = lo < 1 ? 1 : (lo > m - 1 ? m - 1 : lo);
        }
        raw[b] = e;
    }
    __syncthreads();
    // monotone edges (duplicates = empty clusters, which inherit the centre below them in the Lloyd kernel)
    for (int b = t; b <= k; b += blockDim.x) edges[b] = raw[b];
    __syncthreads();
    if (t == 0)
        for (int b = 1; b <= k; ++b)
            if (edges[b] < edges[b - 1]) edges[b] = edges[b - 1];
}

__global__ __launch_bounds__(1024) void q_lloyd_big_kernel(const double *__restrict__ s, int64_t m,
                                                           const double *__restrict__ P, int k, int max_iter,
                                                           const int64_t *__restrict__ edges, double *__restrict__ c,
                                                           int64_t *__restrict__ hi, int64_t *__restrict__ nh,
                                                           double *__restrict__ bounds, int32_t *__restrict__ info)
{
    const int t = threadIdx.x, nt = blockDim.x;
    for (int j = t; j < k; j += nt) {
        const int64_t a = edges[j], b = edges[j + 1];
        hi[j + 1] = b;
        c[j] = (b > a) ? (P[b] - P[a]) / (double)(b - a) : -1e308;      // marker: empty
    }
    if (t == 0) hi[0] = 0;
    __syncthreads();
    if (t == 0) {                                          // empty clusters inherit the centre below them
        double last = 0.0;
        bool have = false;
        for (int j = 0; j < k; ++j) {
            if (c[j] == -1e308) c[j] = have ? last : 0.0;
            else { last = c[j]; have = true; }
        }
    }
    __syncthreads();
    int it = 0;
    for (; it < max_iter; ++it) {
        int changed = 0;
        for (int j = t; j < k; j += nt) {
            int64_t e = m;
            if (j < k - 1) {
                const double bnd = 0.5 * (c[j] + c[j + 1]);
                int64_t lo = 0, up = m;                   // first index with s > bnd
                while (lo < up) { const int64_t mid = (lo + up) >> 1; if (s[mid] <= bnd) lo = mid + 1; else up = mid; }
                e = lo;
            }
            nh[j] = e;
            changed |= (e != hi[j + 1]);
        }
        if (!__syncthreads_or(changed)) break;
        for (int j = t; j < k; j += nt) hi[j + 1] = nh[j];
        __syncthreads();
        for (int j = t; j < k; j += nt) {
            const int64_t a = hi[j], b = hi[j + 1];
            if (b > a) c[j] = (P[b] - P[a]) / (double)(b - a);
        }
        __syncthreads();
    }
    for (int j = t; j < k - 1; j += nt) bounds[j] = 0.5 * (c[j] + c[j + 1]);
    if (t == 0) {
        info[0] = it;
        int nonempty = 0, distinct = 0;
        double last = 0.0;
        for (int q = 0; q < k; ++q) {
            if (hi[q + 1] > hi[q]) {
                ++nonempty;
                if (distinct == 0 || c[q] != last) { ++distinct; last = c[q]; }
            }
        }
        info[1] = nonempty;
        info[2] = distinct;
    }
}

__global__ __launch_bounds__(256) void q_assign_big_kernel(const double *__restrict__ x, int64_t m, int k,
                                                           const double *__restrict__ centers,
                                                           const double *__restrict__ bounds, double *__restrict__ out)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < m; i += stride) {
        const double v = x[i];
        int lo = 0, up = k - 1;                           // first boundary >= v  -> cluster index (tables are L2 hits)
        while (lo < up) { const int mid = (lo + up) >> 1; if (bounds[mid] < v) lo = mid + 1; else up = mid; }
        out[i] = centers[lo];
    }
}

struct QuantPlan {
    int64_t ntiles;
    size_t off_sorted, off_P, off_P2, off_G, off_tsum, off_ce, off_arg, off_edges, off_bounds, off_nb, off_cell, off_D,
        off_big, off_sort_ws, total;
    int64_t kmax;
    int cell_cap;          // micro-cells of the DP stage: Q_CELLS, or QC_BIG when every one of <= QC_BIG values is a cell
};

QuantPlan q_plan(int64_t m)
{
    QuantPlan p;
    p.ntiles = grx_ceil_div(m, Q_TILE);
    size_t o = 0;
    auto take = [&](size_t bytes) { size_t at = o; o += grx_align_up(bytes, 256); return at; };
    p.off_sorted = take((size_t)m * 8);
    p.off_P = take((size_t)(m + 1) * 8);
    p.off_P2 = take((size_t)(m + 1) * 8);
    p.off_G = take((size_t)(m + 1) * 8);
    p.off_tsum = take((size_t)p.ntiles * 3 * 8);
    // tables of the many-level path are sized by the largest admissible level count (n_bins <= m)
    p.kmax = m < Q_BIG_MAX_BINS ? m : Q_BIG_MAX_BINS;
    if (p.kmax < Q_MAX_BINS) p.kmax = Q_MAX_BINS;
    // exact start whenever every value can be its own cell: the r x F factor has m/2 < k <= m levels
    // (k = 2**int(log2(m))) and r x F <= 16 x 480 values, where no density-based start is any good
    p.cell_cap = (m > Q_CELLS && m <= QC_BIG) ? QC_BIG : Q_CELLS;
    const size_t layers = (p.cell_cap == QC_BIG) ? (size_t)m : (size_t)Q_CELLS;
    p.off_ce = take((size_t)(p.cell_cap + 2) * 8);
    p.off_arg = take(layers * (size_t)(p.cell_cap + 1) * 4);
    p.off_edges = take((size_t)(p.kmax + 1) * 8);
    p.off_bounds = take((size_t)p.kmax * 8);
    p.off_big = take((size_t)(p.kmax + 2) * 8 * 3);        // raw edges / running edges / new edges
    p.off_nb = take(256);
    p.off_cell = take((size_t)3 * (p.cell_cap + 1) * 8);
    p.off_D = take((size_t)2 * (p.cell_cap + 1) * 8);
    p.off_sort_ws = take(grx_sort_workspace_bytes(m, 1));
    p.total = o;
    return p;
}

}  // namespace

extern "C" {

size_t grx_lloyd_max_workspace_bytes(int64_t m)
{
    return q_plan(m < 1 ? 1 : m).total;
}

int grx_lloyd_max(int64_t m, const double *d_values, int n_bins, int max_iter, double *d_quantized,
                  double *d_centers, int32_t *d_info, void *d_workspace, size_t workspace_bytes, void *stream)
{
    GRX_REQUIRE(m >= 1, "grx_lloyd_max: no values");
    GRX_REQUIRE(n_bins >= 1 && max_iter >= 0, "grx_lloyd_max: bad n_bins / max_iter");
    GRX_REQUIRE(n_bins <= m, "n_samples=%lld should be >= n_clusters=%d.", (long long)m, n_bins);
    if (n_bins > Q_BIG_MAX_BINS) {
        grx_set_error("grx_lloyd_max: n_bins=%d > %d", n_bins, Q_BIG_MAX_BINS);
        return GRX_ERR_UNSUPPORTED;
    }
    GRX_REQUIRE(m < ((int64_t)1 << 31), "grx_lloyd_max: m must be < 2^31");
    GRX_REQUIRE(d_values && d_quantized && d_centers && d_info && d_workspace, "grx_lloyd_max: NULL pointer");
    const QuantPlan p = q_plan(m);
    if (workspace_bytes < p.total) {
        grx_set_error("grx_lloyd_max: workspace %zu < %zu", workspace_bytes, p.total);
        return GRX_ERR_WORKSPACE;
    }
    hipStream_t st = grx_stream(stream);
    char *ws = reinterpret_cast<char *>(d_workspace);
    double *sorted = reinterpret_cast<double *>(ws + p.off_sorted);
    double *P = reinterpret_cast<double *>(ws + p.off_P);
    double *P2 = reinterpret_cast<double *>(ws + p.off_P2);
    double *Gp = reinterpret_cast<double *>(ws + p.off_G);
    double *tsum = reinterpret_cast<double *>(ws + p.off_tsum);
    int64_t *ce = reinterpret_cast<int64_t *>(ws + p.off_ce);
    int32_t *arg = reinterpret_cast<int32_t *>(ws + p.off_arg);
    int64_t *edges = reinterpret_cast<int64_t *>(ws + p.off_edges);
    double *bounds = reinterpret_cast<double *>(ws + p.off_bounds);
    int *nb = reinterpret_cast<int *>(ws + p.off_nb);
    double *cell = reinterpret_cast<double *>(ws + p.off_cell);
    double *Dbuf = reinterpret_cast<double *>(ws + p.off_D);
    int rc = grx_internal_sort_columns(m, 1, d_values, m, sorted, m, ws + p.off_sort_ws, st);
    if (rc != GRX_OK) return rc;
    {
        GRX_PROF(GRX_K_QUANT, st);
        q_tile_sums_kernel<<<(int)p.ntiles, 256, 0, st>>>(sorted, m, tsum);
        q_scan_tiles_kernel<<<1, 64, 0, st>>>(tsum, p.ntiles);
        q_prefix_kernel<<<(int)p.ntiles, 256, 0, st>>>(sorted, m, tsum, P, P2, Gp);
        int64_t *raw = reinterpret_cast<int64_t *>(ws + p.off_big);
        int64_t *hi = raw + (p.kmax + 2), *nh = hi + (p.kmax + 2);
        if (n_bins <= Q_MAX_BINS || m <= QC_BIG) {
            // exact start: dynamic programming over the micro-cells (every value its own cell when m <= 8192)
            const int cs = p.cell_cap + 1;
            q_cells_kernel<<<1, Q_CELLS, 0, st>>>(Gp, m, p.cell_cap, ce, nb);
            q_dp_init_kernel<<<(p.cell_cap + 256) / 256, 256, 0, st>>>(ce, nb, P, P2, cell, Dbuf, cs);
            for (int j = 0; j < n_bins; ++j) {
                double *Dprev = Dbuf + (size_t)(j & 1) * cs;
                double *Dcur = Dbuf + (size_t)((j + 1) & 1) * cs;
                q_dp_layer_kernel<<<(cs + 3) / 4, 256, 0, st>>>(nb, j, n_bins, cell, Dprev, Dcur, arg, cs);
            }
            q_dp_backtrack_kernel<<<1, 64, 0, st>>>(ce, nb, n_bins, arg, edges, cs);
        } else {
            q_cells_big_kernel<<<1, 1024, 0, st>>>(Gp, m, n_bins, raw, edges);
        }
        if (n_bins > Q_MAX_BINS) {
            q_lloyd_big_kernel<<<1, 1024, 0, st>>>(sorted, m, P, n_bins, max_iter, edges, d_centers, hi, nh, bounds, d_info);
            const int64_t want_big = grx_ceil_div(m, 256 * 4);
            q_assign_big_kernel<<<(int)(want_big > 2048 ? 2048 : want_big), 256, 0, st>>>(d_values, m, n_bins, d_centers,
                                                                                        bounds, d_quantized);
            GRX_LAUNCH_CHECK();
            return GRX_OK;
        }
        q_lloyd_kernel<<<1, Q_MAX_BINS, 0, st>>>(sorted, m, P, n_bins, max_iter, edges, d_centers, bounds, d_info);
        const int64_t want = grx_ceil_div(m, 256 * 4);
        q_assign_kernel<<<(int)(want > 2048 ? 2048 : want), 256, 0, st>>>(d_values, m, n_bins, d_centers, bounds,
                                                                          d_quantized);
    }
    GRX_LAUNCH_CHECK();
    return GRX_OK;
}

}  // extern "C"
