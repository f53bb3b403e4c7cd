// grx_ingest.hip -- graph ingest on the device: edge list -> the CSR structures the ReFeX kernels walk.
//
// Reference: the graph adapters hand the engine a node list and per-node neighbour lists
// (graphrole/graph/interface/networkx.py:36-46, base.py:18-26: rows in sorted-label order; the neighbour sums run
// in G[node] order, features/extract.py:108-110).  On the host that is a CSR build + a degree-descending relabelling
// + an orientation for triangle counting -- ~1 s of numpy per 10 M edges, two orders of magnitude more than the
// step it feeds.  Here the same structures are built from the uploaded edge arrays by a dozen kernels and three
// radix sorts of 64-bit keys:
//   degrees (integer atomics) -> internal order = degree-descending, ties by label (one key sort)
//   -> row pointers (scan) -> adjacency-order neighbour lists: sort (row, edge sequence), the neighbour is the other
//   end of that edge -> ascending neighbour lists: sort (row, column) -> weights by binary search in the sorted row
//   -> directed graphs: the same for the transposed (in-) adjacency.
//   grx_orient_*: the degree-oriented copy (arc u -> v iff (d'(u), u) < (d'(v), v)) with its per-arc table.
// All results are integers (and copied weights): bit-identical to the host construction (tests compare them).
#include "grx_common.h"

int grx_internal_sort_u64(int64_t n, const uint64_t *keys, uint64_t *out, void *workspace, hipStream_t st);
extern "C" size_t grx_sort_workspace_bytes(int64_t n, int ncols);

namespace {

constexpr uint64_t ING_SENTINEL = ~0ull;
constexpr int ING_SCAN_TILE = 2048;

__global__ __launch_bounds__(256) void ing_degree_kernel(int64_t m, const int32_t *__restrict__ src,
                                                         const int32_t *__restrict__ dst, int directed,
                                                         int32_t *__restrict__ deg, int32_t *__restrict__ deg_in)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < m; e += stride) {
        const int32_t u = src[e], v = dst[e];
        atomicAdd(&deg[u], 1);
        if (directed) atomicAdd(&deg_in[v], 1);
        else if (u != v) atomicAdd(&deg[v], 1);
    }
}

// internal order = out-degree descending, ties by label: ascending key ((2^31 - 1 - deg) << 32) | label
__global__ __launch_bounds__(256) void ing_node_keys_kernel(int64_t n, const int32_t *__restrict__ deg,
                                                            uint64_t *__restrict__ keys)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) keys[i] = ((uint64_t)(0x7FFFFFFF - deg[i]) << 32) | (uint64_t)i;
}

__global__ __launch_bounds__(256) void ing_perm_kernel(int64_t n, const uint64_t *__restrict__ sorted,
                                                       const int32_t *__restrict__ deg, const int32_t *__restrict__ deg_in,
                                                       int32_t *__restrict__ perm, int32_t *__restrict__ inv,
                                                       int64_t *__restrict__ cnt, int64_t *__restrict__ cnt_in)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int32_t label = (int32_t)(sorted[i] & 0xFFFFFFFFull);
    perm[i] = label;
    inv[label] = (int32_t)i;
    cnt[i] = deg[label];
    if (cnt_in) cnt_in[i] = deg_in[label];
}

// ---- exclusive scan of int64 counts: out[0..n], out[n] = total -----------------------------------------------
__global__ __launch_bounds__(256) void ing_scan_tiles_kernel(int64_t n, const int64_t *__restrict__ in,
                                                             int64_t *__restrict__ tsum)
{
    __shared__ int64_t red[4];
    const int64_t base = (int64_t)blockIdx.x * ING_SCAN_TILE + threadIdx.x * 8;
    int64_t s = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j)
        if (base + j < n) s += in[base + j];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) tsum[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

__global__ __launch_bounds__(1024) void ing_scan_top_kernel(int64_t ntiles, int64_t *__restrict__ tsum)
{
    __shared__ int64_t s_scan[1024];
    const int64_t chunk = (ntiles + 1023) / 1024;
    const int64_t t0 = (int64_t)threadIdx.x * chunk, t1 = (t0 + chunk < ntiles) ? t0 + chunk : ntiles;
    int64_t local = 0;
    for (int64_t t = t0; t < t1; ++t) local += tsum[t];
    s_scan[threadIdx.x] = local;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        const int64_t add = (threadIdx.x >= (unsigned)off) ? s_scan[threadIdx.x - off] : 0;
        __syncthreads();
        s_scan[threadIdx.x] += add;
        __syncthreads();
    }
    int64_t run = s_scan[threadIdx.x] - local;                  // exclusive prefix of this thread's chunk
    for (int64_t t = t0; t < t1; ++t) { const int64_t v = tsum[t]; tsum[t] = run; run += v; }
}

__global__ __launch_bounds__(256) void ing_scan_apply_kernel(int64_t n, const int64_t *__restrict__ in,
                                                             const int64_t *__restrict__ tsum, int64_t *__restrict__ out)
{
    __shared__ int64_t wtot[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t base = (int64_t)blockIdx.x * ING_SCAN_TILE + threadIdx.x * 8;
    int64_t v[8], a = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) { v[j] = (base + j < n) ? in[base + j] : 0; a += v[j]; }
    int64_t inc = a;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int64_t y = __shfl_up(inc, off, 64);
        if (lane >= off) inc += y;
    }
    if (lane == 63) wtot[wave] = inc;
    __syncthreads();
    int64_t o = tsum[blockIdx.x];
    for (int w = 0; w < wave; ++w) o += wtot[w];
    o += inc - a;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        if (base + j < n) { out[base + j] = o; o += v[j]; if (base + j + 1 == n) out[n] = o; }
    }
}

int ing_scan(int64_t n, const int64_t *in, int64_t *out, int64_t *tsum, hipStream_t st)
{
    const int64_t ntiles = grx_ceil_div(n, ING_SCAN_TILE);
    ing_scan_tiles_kernel<<<(int)ntiles, 256, 0, st>>>(n, in, tsum);
    ing_scan_top_kernel<<<1, 1024, 0, st>>>(ntiles, tsum);
    ing_scan_apply_kernel<<<(int)ntiles, 256, 0, st>>>(n, in, tsum, out);
    GRX_LAUNCH_CHECK();
    return GRX_OK;
}

// ---- arc keys --------------------------------------------------------------------------------------------
// slot 2e: the arc src -> dst of edge e; slot 2e + 1: dst -> src (undirected, not for a self-loop).  For the
// transposed adjacency of a directed graph the roles of the ends are swapped.
//   key_adj = (row << 33) | slot        rows in internal order, neighbours by edge appearance
//   key_col = (row << 32) | column      rows in internal order, columns ascending
__global__ __launch_bounds__(256) void ing_arc_keys_kernel(int64_t m, const int32_t *__restrict__ src,
                                                           const int32_t *__restrict__ dst, const int32_t *__restrict__ inv,
                                                           int mode /* 0 undirected, 1 directed out, 2 directed in */,
                                                           uint64_t *__restrict__ key_adj, uint64_t *__restrict__ key_col)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < m; e += stride) {
        const uint64_t u = (uint64_t)inv[src[e]], v = (uint64_t)inv[dst[e]];
        if (mode == 0) {
            if (key_adj) { key_adj[2 * e] = (u << 33) | (uint64_t)(2 * e); key_adj[2 * e + 1] = (u == v) ? ING_SENTINEL : ((v << 33) | (uint64_t)(2 * e + 1)); }
            key_col[2 * e] = (u << 32) | v;
            key_col[2 * e + 1] = (u == v) ? ING_SENTINEL : ((v << 32) | u);
        } else if (mode == 1) {
            if (key_adj) key_adj[e] = (u << 33) | (uint64_t)e;
            key_col[e] = (u << 32) | v;
        } else {
            key_col[e] = (v << 32) | u;
        }
    }
}

__global__ __launch_bounds__(256) void ing_adj_cols_kernel(int64_t nnz, const uint64_t *__restrict__ sorted_adj,
                                                           const int32_t *__restrict__ src, const int32_t *__restrict__ dst,
                                                           const int32_t *__restrict__ inv, int undirected,
                                                           int32_t *__restrict__ agg_col)
{
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= nnz) return;
    const uint64_t slot = sorted_adj[p] & ((1ull << 33) - 1);
    int32_t other;
    if (undirected) {
        const int64_t e = (int64_t)(slot >> 1);
        other = (slot & 1) ? src[e] : dst[e];
    } else {
        other = dst[slot];
    }
    agg_col[p] = inv[other];
}

__global__ __launch_bounds__(256) void ing_low32_kernel(int64_t nnz, const uint64_t *__restrict__ sorted,
                                                        int32_t *__restrict__ col)
{
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p < nnz) col[p] = (int32_t)(sorted[p] & 0xFFFFFFFFull);
}

// weight of every arc to its place in the sorted row (edges are unique: one position per (row, column))
__global__ __launch_bounds__(256) void ing_weights_kernel(int64_t m, const int32_t *__restrict__ src,
                                                          const int32_t *__restrict__ dst, const double *__restrict__ w,
                                                          const int32_t *__restrict__ inv, int mode,
                                                          const int64_t *__restrict__ row_ptr, const int32_t *__restrict__ col,
                                                          double *__restrict__ wcol)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < m; e += stride) {
        const int32_t u = inv[src[e]], v = inv[dst[e]];
        const double we = w[e];
        const int narcs = (mode == 0 && u != v) ? 2 : 1;
        for (int a = 0; a < narcs; ++a) {
            int32_t r, c;
            if (mode == 2) { r = v; c = u; }
            else if (a == 0) { r = u; c = v; }
            else { r = v; c = u; }
            int64_t lo = row_ptr[r], hi = row_ptr[r + 1];
            while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (col[mid] < c) lo = mid + 1; else hi = mid; }
            wcol[lo] = we;
        }
    }
}

// ---- orientation -------------------------------------------------------------------------------------------
// d'(v) = degree without the self-loop; arc u -> v kept iff (d'(u), u) < (d'(v), v)
__global__ __launch_bounds__(256) void orient_dprime_kernel(int64_t n, const int64_t *__restrict__ row_ptr,
                                                            const int32_t *__restrict__ col, int32_t *__restrict__ dprime)
{
    const int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= n) return;
    int64_t lo = row_ptr[v];
    const int64_t b = lo, e = row_ptr[v + 1];
    int64_t hi = e;
    while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (col[mid] < (int32_t)v) lo = mid + 1; else hi = mid; }
    const int loop = (lo < e && col[lo] == (int32_t)v) ? 1 : 0;
    dprime[v] = (int32_t)(e - b) - loop;
}

// One wavefront per row (rows are in degree-descending order: a thread per row left the 10 k-neighbour hubs to
// single lanes, 4 ms per kernel on BA 1 M): lanes take consecutive arcs, the kept ones are counted / placed with a
// ballot, so the oriented list keeps the ascending order of the row.
__device__ __forceinline__ bool orient_keeps(int32_t du, int32_t u, int32_t dv, int32_t v)
{
    return (du < dv) || (du == dv && u < v);
}

__global__ __launch_bounds__(256) void orient_count_kernel(int64_t n, const int64_t *__restrict__ row_ptr,
                                                           const int32_t *__restrict__ col, const int32_t *__restrict__ dprime,
                                                           int64_t *__restrict__ cnt)
{
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t u = wave; u < n; u += nwaves) {
        const int32_t du = dprime[u];
        const int64_t b = row_ptr[u], e = row_ptr[u + 1];
        int64_t c = 0;
        for (int64_t k0 = b; k0 < e; k0 += 64) {
            const int64_t k = k0 + lane;
            bool keep = false;
            if (k < e) {
                const int32_t v = col[k];
                keep = orient_keeps(du, (int32_t)u, dprime[v], v);
            }
            c += __popcll(__ballot(keep));
        }
        if (lane == 0) cnt[u] = c;
    }
}

__global__ __launch_bounds__(256) void orient_fill_kernel(int64_t n, const int64_t *__restrict__ row_ptr,
                                                          const int32_t *__restrict__ col, const int32_t *__restrict__ dprime,
                                                          const int64_t *__restrict__ o_row_ptr, int32_t *__restrict__ o_col)
{
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    const uint64_t below = lane == 0 ? 0ull : (~0ull >> (64 - lane));
    for (int64_t u = wave; u < n; u += nwaves) {
        const int32_t du = dprime[u];
        const int64_t b = row_ptr[u], e = row_ptr[u + 1];
        int64_t at = o_row_ptr[u];
        for (int64_t k0 = b; k0 < e; k0 += 64) {
            const int64_t k = k0 + lane;
            bool keep = false;
            int32_t v = 0;
            if (k < e) {
                v = col[k];
                keep = orient_keeps(du, (int32_t)u, dprime[v], v);
            }
            const uint64_t kept = __ballot(keep);
            if (keep) o_col[at + __popcll(kept & below)] = v;
            at += __popcll(kept);
        }
    }
}

// per oriented arc k = u->v (grx.h): begin of N+(v) | min(|N+(v)|, 1023) << 32 | min(|N+(u)|, 1023) << 42 |
// min(k - begin of N+(u), 1023) << 52
__global__ __launch_bounds__(256) void orient_arc_kernel(int64_t n, const int64_t *__restrict__ o_row_ptr,
                                                         const int32_t *__restrict__ o_col, uint64_t *__restrict__ arc)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t u = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; u < n; u += stride) {
        const int64_t ub = o_row_ptr[u], ue = o_row_ptr[u + 1];
        const uint64_t ulen = (uint64_t)(ue - ub) < 1023 ? (uint64_t)(ue - ub) : 1023;
        for (int64_t k = ub; k < ue; ++k) {
            const int32_t v = o_col[k];
            const uint64_t b = (uint64_t)o_row_ptr[v], len = (uint64_t)(o_row_ptr[v + 1] - o_row_ptr[v]);
            const uint64_t pos = (uint64_t)(k - ub) < 1023 ? (uint64_t)(k - ub) : 1023;
            arc[k] = b | ((len < 1023 ? len : 1023) << 32) | (ulen << 42) | (pos << 52);
        }
    }
}

// out[c][i] = col_c[idx[i]]: feature columns from the internal row order back to label order, all columns of the
// result table in one launch (the host then needs a single copy instead of a fancy-index gather per column)
__global__ __launch_bounds__(256) void permute_columns_kernel(int64_t n, GrxPtrTable ptr_tab, const int32_t *__restrict__ idx,
                                                              double *__restrict__ out, int64_t ld)
{
    const double *src = reinterpret_cast<const double *>(ptr_tab.p[blockIdx.y]);
    double *dst = out + (size_t)blockIdx.y * ld;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) dst[i] = src[idx[i]];
}

struct IngestPlan {
    size_t off_deg, off_deg_in, off_keys_n, off_sorted_n, off_cnt, off_cnt_in, off_tsum, off_key_a, off_key_b, off_out, off_sort,
        total;
    int64_t slots;
};

IngestPlan ingest_plan(int64_t n, int64_t m, int directed)
{
    IngestPlan p;
    p.slots = directed ? m : 2 * m;
    size_t o = 0;
    auto take = [&](size_t bytes) { size_t at = o; o += grx_align_up(bytes ? bytes : 8, 256); return at; };
    p.off_deg = take((size_t)n * 4);
    p.off_deg_in = take((size_t)n * 4);
    p.off_keys_n = take((size_t)n * 8);
    p.off_sorted_n = take((size_t)n * 8);
    p.off_cnt = take((size_t)(n + 1) * 8);
    p.off_cnt_in = take((size_t)(n + 1) * 8);
    p.off_tsum = take((size_t)(grx_ceil_div(n > 1 ? n : 1, ING_SCAN_TILE) + 1) * 8);
    p.off_key_a = take((size_t)p.slots * 8);
    p.off_key_b = take((size_t)p.slots * 8);
    p.off_out = take((size_t)p.slots * 8);
    const size_t s1 = grx_sort_workspace_bytes(n > 1 ? n : 1, 1), s2 = grx_sort_workspace_bytes(p.slots > 1 ? p.slots : 1, 1);
    p.off_sort = take(s1 > s2 ? s1 : s2);
    p.total = o;
    return p;
}

}  // namespace

extern "C" {

size_t grx_ingest_workspace_bytes(int64_t n, int64_t m, int directed)
{
    return ingest_plan(n < 1 ? 1 : n, m < 1 ? 1 : m, directed).total;
}

int grx_ingest(int64_t n, int64_t m, const int32_t *d_src, const int32_t *d_dst, const double *d_w, int directed,
               int64_t nnz, int32_t *d_perm, int32_t *d_inv, int64_t *d_row_ptr, int32_t *d_col, double *d_wcol,
               int32_t *d_agg_col, int64_t *d_t_row_ptr, int32_t *d_t_col, double *d_t_w, void *d_workspace,
               size_t workspace_bytes, void *stream)
{
    GRX_REQUIRE(n >= 1 && m >= 1 && n < ((int64_t)1 << 31) && m < ((int64_t)1 << 30), "grx_ingest: bad n / m");
    GRX_REQUIRE(d_src && d_dst && d_perm && d_inv && d_row_ptr && d_col && d_agg_col && d_workspace, "grx_ingest: NULL pointer");
    GRX_REQUIRE(!d_w || d_wcol, "grx_ingest: weights need an output array");
    GRX_REQUIRE(!directed || (d_t_row_ptr && d_t_col && (!d_w || d_t_w)), "grx_ingest: directed graphs need the transposed outputs");
    GRX_REQUIRE(nnz >= 1 && nnz <= (directed ? m : 2 * m), "grx_ingest: nnz outside (0, %s]", directed ? "m" : "2m");
    const IngestPlan p = ingest_plan(n, m, directed);
    if (workspace_bytes < p.total) {
        grx_set_error("grx_ingest: workspace %zu < %zu", workspace_bytes, p.total);
        return GRX_ERR_WORKSPACE;
    }
    hipStream_t st = grx_stream(stream);
    char *ws = reinterpret_cast<char *>(d_workspace);
    int32_t *deg = reinterpret_cast<int32_t *>(ws + p.off_deg);
    int32_t *deg_in = reinterpret_cast<int32_t *>(ws + p.off_deg_in);
    uint64_t *keys_n = reinterpret_cast<uint64_t *>(ws + p.off_keys_n);
    uint64_t *sorted_n = reinterpret_cast<uint64_t *>(ws + p.off_sorted_n);
    int64_t *cnt = reinterpret_cast<int64_t *>(ws + p.off_cnt);
    int64_t *cnt_in = reinterpret_cast<int64_t *>(ws + p.off_cnt_in);
    int64_t *tsum = reinterpret_cast<int64_t *>(ws + p.off_tsum);
    uint64_t *key_a = reinterpret_cast<uint64_t *>(ws + p.off_key_a);
    uint64_t *key_b = reinterpret_cast<uint64_t *>(ws + p.off_key_b);
    void *sort_ws = ws + p.off_sort;
    const int egrid = (int)(grx_ceil_div(m, 256) > 4096 ? 4096 : grx_ceil_div(m, 256));
    const int ngrid = (int)grx_ceil_div(n, 256);
    GRX_CHECK_HIP(hipMemsetAsync(deg, 0, (size_t)n * 4, st));
    GRX_CHECK_HIP(hipMemsetAsync(deg_in, 0, (size_t)n * 4, st));
    ing_degree_kernel<<<egrid, 256, 0, st>>>(m, d_src, d_dst, directed, deg, deg_in);
    ing_node_keys_kernel<<<ngrid, 256, 0, st>>>(n, deg, keys_n);
    GRX_LAUNCH_CHECK();
    int rc = grx_internal_sort_u64(n, keys_n, sorted_n, sort_ws, st);
    if (rc != GRX_OK) return rc;
    ing_perm_kernel<<<ngrid, 256, 0, st>>>(n, sorted_n, deg, deg_in, d_perm, d_inv, cnt, directed ? cnt_in : nullptr);
    GRX_LAUNCH_CHECK();
    rc = ing_scan(n, cnt, d_row_ptr, tsum, st);
    if (rc != GRX_OK) return rc;
    if (directed) {
        rc = ing_scan(n, cnt_in, d_t_row_ptr, tsum, st);
        if (rc != GRX_OK) return rc;
    }
    const int agrid = (int)grx_ceil_div(nnz, 256);
    // adjacency order, then ascending columns, of the out-adjacency
    ing_arc_keys_kernel<<<egrid, 256, 0, st>>>(m, d_src, d_dst, d_inv, directed ? 1 : 0, key_a, key_b);
    GRX_LAUNCH_CHECK();
    uint64_t *out = reinterpret_cast<uint64_t *>(ws + p.off_out);
    rc = grx_internal_sort_u64(p.slots, key_a, out, sort_ws, st);          // sentinels (second arc of a loop) sort last
    if (rc != GRX_OK) return rc;
    ing_adj_cols_kernel<<<agrid, 256, 0, st>>>(nnz, out, d_src, d_dst, d_inv, directed ? 0 : 1, d_agg_col);
    GRX_LAUNCH_CHECK();
    rc = grx_internal_sort_u64(p.slots, key_b, out, sort_ws, st);
    if (rc != GRX_OK) return rc;
    ing_low32_kernel<<<agrid, 256, 0, st>>>(nnz, out, d_col);
    if (d_w) ing_weights_kernel<<<egrid, 256, 0, st>>>(m, d_src, d_dst, d_w, d_inv, directed ? 1 : 0, d_row_ptr, d_col, d_wcol);
    GRX_LAUNCH_CHECK();
    if (directed) {
        // the transposed (in-) adjacency: rows = targets, columns = sources, ascending
        ing_arc_keys_kernel<<<egrid, 256, 0, st>>>(m, d_src, d_dst, d_inv, 2, nullptr, key_b);
        GRX_LAUNCH_CHECK();
        rc = grx_internal_sort_u64(m, key_b, out, sort_ws, st);
        if (rc != GRX_OK) return rc;
        ing_low32_kernel<<<(int)grx_ceil_div(m, 256), 256, 0, st>>>(m, out, d_t_col);
        if (d_w) ing_weights_kernel<<<egrid, 256, 0, st>>>(m, d_src, d_dst, d_w, d_inv, 2, d_t_row_ptr, d_t_col, d_t_w);
        GRX_LAUNCH_CHECK();
    }
    return GRX_OK;
}

int grx_permute_columns(int64_t n, int F, const double *const *h_col_ptrs, const int32_t *d_index, double *d_out,
                        int64_t ld, void *stream)
{
    GRX_REQUIRE(n >= 0 && F >= 0 && ld >= n, "grx_permute_columns: bad shape");
    if (n == 0 || F == 0) return GRX_OK;
    GRX_REQUIRE(h_col_ptrs && d_index && d_out, "grx_permute_columns: NULL pointer");
    const int64_t want = grx_ceil_div(n, 256 * 4);
    for (int c0 = 0; c0 < F; c0 += GRX_MAX_PTRS) {
        const int fc = (F - c0 < GRX_MAX_PTRS) ? F - c0 : GRX_MAX_PTRS;
        GrxPtrTable tab;
        for (int c = 0; c < fc; ++c) tab.p[c] = h_col_ptrs[c0 + c];
        const dim3 grid((unsigned)(want > 1024 ? 1024 : want), fc);
        permute_columns_kernel<<<grid, 256, 0, grx_stream(stream)>>>(n, tab, d_index, d_out + (size_t)c0 * ld, ld);
        GRX_LAUNCH_CHECK();
    }
    return GRX_OK;
}

size_t grx_orient_workspace_bytes(int64_t n)
{
    if (n < 1) n = 1;
    return grx_align_up((size_t)n * 4, 256) + grx_align_up((size_t)(n + 1) * 8, 256) +
           grx_align_up((size_t)(grx_ceil_div(n, ING_SCAN_TILE) + 1) * 8, 256);
}

int grx_orient_count(int64_t n, const int64_t *d_row_ptr, const int32_t *d_col, int64_t *d_o_row_ptr, void *d_workspace,
                     size_t workspace_bytes, void *stream)
{
    GRX_REQUIRE(n >= 1 && d_row_ptr && d_col && d_o_row_ptr && d_workspace, "grx_orient_count: bad arguments");
    if (workspace_bytes < grx_orient_workspace_bytes(n)) {
        grx_set_error("grx_orient_count: workspace too small");
        return GRX_ERR_WORKSPACE;
    }
    hipStream_t st = grx_stream(stream);
    char *ws = reinterpret_cast<char *>(d_workspace);
    int32_t *dprime = reinterpret_cast<int32_t *>(ws);
    int64_t *cnt = reinterpret_cast<int64_t *>(ws + grx_align_up((size_t)n * 4, 256));
    int64_t *tsum = reinterpret_cast<int64_t *>(ws + grx_align_up((size_t)n * 4, 256) + grx_align_up((size_t)(n + 1) * 8, 256));
    const int ngrid = (int)grx_ceil_div(n, 256);
    orient_dprime_kernel<<<ngrid, 256, 0, st>>>(n, d_row_ptr, d_col, dprime);
    {
        const int64_t want = grx_ceil_div(n, 4);                         // a wavefront per row, four per workgroup
        const int wgrid = (int)(want > GRX_NUM_CU * 32 ? GRX_NUM_CU * 32 : want);
        orient_count_kernel<<<wgrid, 256, 0, st>>>(n, d_row_ptr, d_col, dprime, cnt);
    }
    GRX_LAUNCH_CHECK();
    return ing_scan(n, cnt, d_o_row_ptr, tsum, st);
}

int grx_orient_fill(int64_t n, const int64_t *d_row_ptr, const int32_t *d_col, const int64_t *d_o_row_ptr, int64_t o_nnz,
                    int32_t *d_o_col, uint64_t *d_o_arc, void *d_workspace, size_t workspace_bytes, void *stream)
{
    GRX_REQUIRE(n >= 1 && d_row_ptr && d_col && d_o_row_ptr && d_workspace && o_nnz >= 0, "grx_orient_fill: bad arguments");
    GRX_REQUIRE(o_nnz == 0 || (d_o_col && d_o_arc), "grx_orient_fill: NULL output");
    if (workspace_bytes < grx_orient_workspace_bytes(n)) {
        grx_set_error("grx_orient_fill: workspace too small");
        return GRX_ERR_WORKSPACE;
    }
    hipStream_t st = grx_stream(stream);
    const int32_t *dprime = reinterpret_cast<const int32_t *>(d_workspace);      // left there by grx_orient_count
    const int64_t want = grx_ceil_div(n, 4);
    const int wgrid = (int)(want > GRX_NUM_CU * 32 ? GRX_NUM_CU * 32 : want);
    orient_fill_kernel<<<wgrid, 256, 0, st>>>(n, d_row_ptr, d_col, dprime, d_o_row_ptr, d_o_col);
    if (o_nnz) {
        const int64_t awant = grx_ceil_div(n, 256);
        orient_arc_kernel<<<(int)(awant > GRX_NUM_CU * 32 ? GRX_NUM_CU * 32 : awant), 256, 0, st>>>(n, d_o_row_ptr, d_o_col, d_o_arc);
    }
    GRX_LAUNCH_CHECK();
    return GRX_OK;
}

}  // extern "C"
