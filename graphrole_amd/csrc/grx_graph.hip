// grx_graph.hip -- generation-0 features and the ReFeX neighbour aggregation on a CSR graph.
//
// Kernels (all HBM/L2-gather bound; no MFMA -- this is integer-indexed fp64 streaming work):
//   row_sums_kernel        weighted degree            networkx.py:48-63
//   egonet_kernel          ego-net internal/external  networkx.py:71-83,115-123
//   pack_rows_kernel       column-major -> row-major gather source
//   aggregate_kernel       sum / mean over neighbours  features/extract.py:98-119
//   aggregate_hub_kernel   same, one workgroup per high-degree row
//
// Determinism: every reduction is "per-lane sequential in CSR order, then a fixed butterfly /
// fixed-order LDS sum", so the addition tree of a row depends only on its degree and the launch
// geometry chosen from the graph's average degree.
#include "grx_common.h"

namespace {

// ---------------------------------------------------------------------------------------
// small device helpers
// ---------------------------------------------------------------------------------------
// position of key in the ascending slice col[b,e), or -1
__device__ __forceinline__ int64_t find_in_row(const int32_t *__restrict__ col, int64_t b,
                                               int64_t e, int32_t key)
{
    while (b < e) {
        int64_t mid = (b + e) >> 1;
        int32_t c = col[mid];
        if (c < key) b = mid + 1;
        else if (c > key) e = mid;
        else return mid;
    }
    return -1;
}

// first position in col[b,e) with col[pos] >= key
__device__ __forceinline__ int64_t lower_bound_row(const int32_t *__restrict__ col, int64_t b,
                                                   int64_t e, int32_t key)
{
    while (b < e) {
        int64_t mid = (b + e) >> 1;
        if (col[mid] < key) b = mid + 1;
        else e = mid;
    }
    return b;
}

__device__ __forceinline__ int ilog2_i64(int64_t x) { return 63 - __clzll((unsigned long long)(x | 1)); }

// ---------------------------------------------------------------------------------------
// weighted row sums
// ---------------------------------------------------------------------------------------
template <int G>
__global__ __launch_bounds__(256) void row_sums_kernel(const int64_t *__restrict__ row_ptr,
                                                       const int32_t *__restrict__ col,
                                                       const double *__restrict__ w, int add_loop,
                                                       int64_t row_begin, int64_t row_end,
                                                       double *__restrict__ out)
{
    const int lane = threadIdx.x % G;
    const int64_t group = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / G;
    const int64_t ngroups = (int64_t)gridDim.x * blockDim.x / G;
    for (int64_t v = row_begin + group; v < row_end; v += ngroups) {
        const int64_t b = row_ptr[v], e = row_ptr[v + 1];
        if (w == nullptr) {                                 // implicit weight 1: degree from row_ptr
            if (lane == 0)
                out[v] = (double)((e - b) + ((add_loop && find_in_row(col, b, e, (int32_t)v) >= 0) ? 1 : 0));
            continue;
        }
        double s = 0.0, loop = 0.0;
        for (int64_t k = b + lane; k < e; k += G) {
            const double x = w ? w[k] : 1.0;
            s += x;
            if (add_loop && col[k] == v) loop = x;
        }
        s = grx_group_sum<G>(s);
        loop = grx_group_sum<G>(loop);
        if (lane == 0) out[v] = s + loop;
    }
}

// ---------------------------------------------------------------------------------------
// ego-net features
// ---------------------------------------------------------------------------------------
// One node per TPN threads.  Each lane owns ego members m = lane, lane+TPN, ... and for member a
// picks the cheaper of
//   S1: walk row(a), test membership of each entry in ego(v)        cost deg(a) * log deg(v)
//   S2: walk ego(v), look each member up in row(a)                   cost deg(v) * log deg(a)
// S2 obtains the boundary weight of a as rowsum(a) - matched weight; when every entry of row(a)
// matched, the boundary contribution is exactly 0 (no cancellation residue).
template <int TPN>
__global__ __launch_bounds__(TPN == 64 ? 256 : TPN) void egonet_kernel(
    const int64_t *__restrict__ row_ptr, const int32_t *__restrict__ col,
    const double *__restrict__ w, const double *__restrict__ rowsum, int directed,
    int64_t row_begin, int64_t row_end, int64_t deg_lo, int64_t deg_hi,
    double *__restrict__ internal, double *__restrict__ external)
{
    constexpr int BLOCK = (TPN == 64) ? 256 : TPN;
    constexpr int NODES_PER_BLOCK = BLOCK / TPN;
    constexpr int WAVES = TPN / 64;
    __shared__ double red[2][WAVES > 1 ? WAVES : 1];
    const int lane = threadIdx.x % TPN;
    const int64_t slot = (int64_t)blockIdx.x * NODES_PER_BLOCK + threadIdx.x / TPN;
    const int64_t nslots = (int64_t)gridDim.x * NODES_PER_BLOCK;

    for (int64_t v = row_begin + slot; v < row_end; v += nslots) {
        const int64_t vb = row_ptr[v], ve = row_ptr[v + 1];
        const int64_t dv = ve - vb;
        if (dv < deg_lo || dv >= deg_hi) continue;          // uniform over the TPN threads
        const bool v_in_row = find_in_row(col, vb, ve, (int32_t)v) >= 0;
        const int64_t members = dv + (v_in_row ? 0 : 1);
        const int lg_dv = ilog2_i64(members) + 2;
        double ins = 0.0, ext = 0.0;
        for (int64_t m = lane; m < members; m += TPN) {
            const int64_t a = (m < dv) ? (int64_t)col[vb + m] : v;
            const int64_t ab = row_ptr[a], ae = row_ptr[a + 1];
            const int64_t da = ae - ab;
            if (da * lg_dv <= members * (int64_t)(ilog2_i64(da) + 2)) {
                for (int64_t j = ab; j < ae; ++j) {
                    const int32_t b = col[j];
                    const double x = w ? w[j] : 1.0;
                    const bool inside = (b == (int32_t)v) || find_in_row(col, vb, ve, b) >= 0;
                    if (inside) {
                        if (directed || b >= a) ins += x;
                    } else {
                        ext += x;
                    }
                }
            } else {
                int64_t matched = 0;
                double in_all = 0.0;
                int64_t lo = ab;
                // look one ego member up in the not-yet-passed tail of row(a)
                auto probe = [&](int32_t b) {
                    const int64_t pos = lower_bound_row(col, lo, ae, b);
                    if (pos < ae && col[pos] == b) {
                        const double x = w ? w[pos] : 1.0;
                        ++matched;
                        in_all += x;
                        if (directed || b >= a) ins += x;
                        lo = pos + 1;
                    } else {
                        lo = pos;
                    }
                };
                bool v_pending = !v_in_row;                  // v merged at its sorted position
                for (int64_t t = 0; t < dv && lo < ae; ++t) {
                    const int32_t b = col[vb + t];
                    if (v_pending && (int32_t)v < b) {
                        probe((int32_t)v);
                        v_pending = false;
                        if (lo >= ae) break;
                    }
                    probe(b);
                }
                if (v_pending && lo < ae) probe((int32_t)v);
                if (matched != da) ext += (w ? rowsum[a] : (double)da) - in_all;
            }
        }
        ins = grx_group_sum<64>(ins);
        ext = grx_group_sum<64>(ext);
        if constexpr (WAVES > 1) {
            const int wv = threadIdx.x / 64;
            if ((threadIdx.x & 63) == 0) { red[0][wv] = ins; red[1][wv] = ext; }
            __syncthreads();
            if (threadIdx.x == 0) {
                double si = 0.0, se = 0.0;
                for (int i = 0; i < WAVES; ++i) { si += red[0][i]; se += red[1][i]; }
                internal[v] = si;
                external[v] = se;
            }
            __syncthreads();
        } else {
            if (lane == 0) { internal[v] = ins; external[v] = ext; }
        }
    }
}

// ---------------------------------------------------------------------------------------
// ego-net features of UNWEIGHTED UNDIRECTED graphs through per-node triangle counts
// ---------------------------------------------------------------------------------------
// With A' = adjacency without the diagonal, d'(v) its degrees, L(v) the self-loop flags and
// T(v) the number of triangles through v (= edges among the neighbours of v):
//     internal(v) = d'(v) + T(v) + sum_{a in ego(v)} L(a)
//     external(v) = sum_{a in ego(v)} d'(a) - 2 (d'(v) + T(v))
// (networkx.py:71-83 counted edge by edge; all quantities are integers, so this is exact.)
// T comes from the degree-oriented graph (arc u->v iff (d'(u),u) < (d'(v),v)): every triangle
// is found exactly once as a common out-neighbour of the two ends of its lowest arc; oriented
// lists are short even for power-law hubs, so a two-pointer merge per arc is cheap.
// An 8-lane group owns one source u and walks its oriented arcs u->v TOGETHER: the group reads
// N+(v) with one coalesced 32-byte load per 8 elements (instead of eight lanes chasing eight
// different lists), keeps N+(u) in registers (3 entries per lane = 24 per pass; degree ordering
// keeps oriented lists <= ~20 on the BASELINE graphs) and tests membership all-to-all with
// in-group shuffles.  L2 requests per arc drop from ~16 to ~4, which is what bounded this kernel
// (rocprof: 165 M TCP->TCC requests per launch).  Counting is integer atomics: exact, any order.
__global__ __launch_bounds__(256) void triangle_count_kernel(
    const int64_t *__restrict__ o_row_ptr, const int32_t *__restrict__ o_col, int64_t row_begin,
    int64_t row_end, unsigned long long *__restrict__ T)
{
    constexpr int G = 8;
    const int lane = threadIdx.x % G;
    const int gshift = (threadIdx.x & 63) & ~(G - 1);          // bit offset of this group in a ballot
    const int64_t group = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / G;
    const int64_t ngroups = (int64_t)gridDim.x * blockDim.x / G;
    for (int64_t u = row_begin + group; u < row_end; u += ngroups) {
        const int64_t ub = o_row_ptr[u], ue = o_row_ptr[u + 1];
        unsigned long long cu = 0;
        for (int64_t base = ub; base < ue; base += 3 * G) {     // usually a single pass
            int32_t uu[3];
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const int64_t idx = base + lane + (int64_t)G * i;
                uu[i] = (idx < ue) ? o_col[idx] : -2;
            }
            for (int64_t k = ub; k < ue; ++k) {
                const int32_t v = o_col[k];                     // same address in the group
                const int64_t vb = o_row_ptr[v], ve = o_row_ptr[v + 1];
                unsigned c_arc = 0;
                for (int64_t j0 = vb; j0 < ve; j0 += G) {
                    const int32_t y = (j0 + lane < ve) ? o_col[j0 + lane] : -1;
                    unsigned match = 0;
#pragma unroll
                    for (int sidx = 0; sidx < G; ++sidx) {
                        const int32_t ys = __shfl(y, sidx, G);
                        const bool hit = (ys == uu[0]) | (ys == uu[1]) | (ys == uu[2]);
                        const unsigned long long bal = __ballot(hit);
                        if ((bal >> gshift) & 0xFFull) match |= 1u << sidx;
                    }
                    if ((match >> lane) & 1u) atomicAdd(&T[y], 1ull);
                    c_arc += __popc(match);
                }
                if (lane == 0 && c_arc) atomicAdd(&T[v], (unsigned long long)c_arc);
                cu += c_arc;
            }
        }
        if (lane == 0 && cu) atomicAdd(&T[u], cu);
    }
}

// info[v] = (d'(v) << 1) | L(v)   (int32: the whole table is 4 B/node and stays L2-resident)
__global__ __launch_bounds__(256) void node_info_kernel(int64_t n, const int64_t *__restrict__ row_ptr,
                                                        const int32_t *__restrict__ col,
                                                        int32_t *__restrict__ info)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < n; v += stride) {
        const int64_t b = row_ptr[v], e = row_ptr[v + 1];
        const int64_t loop = find_in_row(col, b, e, (int32_t)v) >= 0 ? 1 : 0;
        info[v] = (int32_t)((((e - b) - loop) << 1) | loop);
    }
}

__device__ __forceinline__ void egonet_finish_row(int64_t v, long long sum_d, long long loops,
                                                  const int32_t *__restrict__ info,
                                                  const unsigned long long *__restrict__ T,
                                                  double *__restrict__ internal, double *__restrict__ external)
{
    const int32_t iv = info[v];
    const long long dv = iv >> 1;
    const long long core = dv + (long long)T[v];
    internal[v] = (double)(core + loops + (iv & 1));
    external[v] = (double)(sum_d + dv - 2 * core);
}

// G = 8 lanes per row; rows with more than hub_deg neighbours are left to the workgroup-per-row
// variant below (integer sums: any order is exact).
__global__ __launch_bounds__(256) void egonet_from_triangles_kernel(
    const int64_t *__restrict__ row_ptr, const int32_t *__restrict__ col,
    const int32_t *__restrict__ info, const unsigned long long *__restrict__ T, int64_t row_begin,
    int64_t row_end, int64_t hub_deg, double *__restrict__ internal, double *__restrict__ external)
{
    constexpr int G = 8;
    const int lane = threadIdx.x % G;
    const int64_t group = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / G;
    const int64_t ngroups = (int64_t)gridDim.x * blockDim.x / G;
    for (int64_t v = row_begin + group; v < row_end; v += ngroups) {
        const int64_t b = row_ptr[v], e = row_ptr[v + 1];
        if (e - b > hub_deg) continue;
        long long sum_d = 0, loops = 0;
        for (int64_t k = b + lane; k < e; k += G) {
            const int32_t a = col[k];
            if (a != (int32_t)v) {
                const int32_t ia = info[a];
                sum_d += ia >> 1;
                loops += ia & 1;
            }
        }
#pragma unroll
        for (int off = G / 2; off > 0; off >>= 1) {
            sum_d += __shfl_xor(sum_d, off, G);
            loops += __shfl_xor(loops, off, G);
        }
        if (lane == 0) egonet_finish_row(v, sum_d, loops, info, T, internal, external);
    }
}

__global__ __launch_bounds__(256) void egonet_from_triangles_hub_kernel(
    const int64_t *__restrict__ row_ptr, const int32_t *__restrict__ col,
    const int32_t *__restrict__ info, const unsigned long long *__restrict__ T, int64_t row_begin,
    int64_t row_end, const int32_t *__restrict__ hub_rows, int64_t n_hubs,
    double *__restrict__ internal, double *__restrict__ external)
{
    __shared__ long long red[2][4];
    for (int64_t h = blockIdx.x; h < n_hubs; h += gridDim.x) {
        const int64_t v = hub_rows[h];
        if (v < row_begin || v >= row_end) continue;
        const int64_t b = row_ptr[v], e = row_ptr[v + 1];
        long long sum_d = 0, loops = 0;
        for (int64_t k = b + threadIdx.x; k < e; k += 256) {
            const int32_t a = col[k];
            if (a != (int32_t)v) {
                const int32_t ia = info[a];
                sum_d += ia >> 1;
                loops += ia & 1;
            }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            sum_d += __shfl_xor(sum_d, off, 64);
            loops += __shfl_xor(loops, off, 64);
        }
        if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = sum_d; red[1][threadIdx.x >> 6] = loops; }
        __syncthreads();
        if (threadIdx.x == 0)
            egonet_finish_row(v, red[0][0] + red[0][1] + red[0][2] + red[0][3],
                              red[1][0] + red[1][1] + red[1][2] + red[1][3], info, T, internal, external);
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------
// pack: column-major columns -> row-major n x ldr (zero padded)
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pack_rows_kernel(int64_t n, int f, int ldr, GrxPtrTable cols_tab,
                                                        double *__restrict__ rows)
{
    const double *const *cols = reinterpret_cast<const double *const *>(cols_tab.p);
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        double *dst = rows + i * ldr;
        for (int c = 0; c < f; ++c) dst[c] = cols[c][i];
        for (int c = f; c < ldr; ++c) dst[c] = 0.0;
    }
}

__global__ __launch_bounds__(256) void add_columns_kernel(int64_t n, const double *__restrict__ a,
                                                          const double *__restrict__ b,
                                                          double *__restrict__ out)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) out[i] = a[i] + b[i];
}

// ---------------------------------------------------------------------------------------
// neighbour aggregation
// ---------------------------------------------------------------------------------------
// G lanes cooperate on one output row.  A neighbour's feature row is padded to LDR doubles
// (LDR*8 = 16/32/64/128 bytes, so a row never straddles a 128-byte line and 64-byte rows are
// exactly one cache line) and is fetched by CL = LDR/2 ADJACENT lanes, 16 bytes each: one
// wave-level load touches 64/CL distinct lines instead of 64, which is what the address
// units / L1 care about for a random gather.  Lane = (slot, part): slot = lane / CL walks the
// neighbour list (two neighbours per trip for memory-level parallelism), part = lane % CL owns
// columns 2*part, 2*part+1.  The slots are combined by a fixed butterfly, so the addition tree
// of a row depends only on its degree and the launch geometry.  Rows with degree > hub_deg are
// left to aggregate_hub_kernel.
template <int LDR, int G>
__global__ __launch_bounds__(256) void aggregate_kernel(
    const int64_t *__restrict__ row_ptr, const int32_t *__restrict__ col,
    const double *__restrict__ rows, int64_t row_stride, int f, int64_t row_begin, int64_t row_end,
    int64_t hub_deg, double *__restrict__ out_sum, double *__restrict__ out_mean, int64_t ld)
{
    constexpr int CL = LDR / 2;           // lanes per neighbour row
    constexpr int S = G / CL;             // neighbour slots per output row
    static_assert(S >= 1 && (S & (S - 1)) == 0, "G must be a power-of-two multiple of LDR/2");
    const int lane = threadIdx.x % G;
    const int part = lane % CL, slot = lane / CL;
    const int64_t group = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / G;
    const int64_t ngroups = (int64_t)gridDim.x * blockDim.x / G;
    for (int64_t v = row_begin + group; v < row_end; v += ngroups) {
        const int64_t b = row_ptr[v], e = row_ptr[v + 1];
        const int64_t d = e - b;
        if (d > hub_deg) continue;
        double a0 = 0.0, a1 = 0.0;
        int64_t k = b + slot;
        for (; k + S < e; k += 2 * S) {
            const int64_t u0 = col[k], u1 = col[k + S];
            const double2 x0 = *reinterpret_cast<const double2 *>(rows + u0 * row_stride + 2 * part);
            const double2 x1 = *reinterpret_cast<const double2 *>(rows + u1 * row_stride + 2 * part);
            a0 += x0.x; a1 += x0.y;
            a0 += x1.x; a1 += x1.y;
        }
        if (k < e) {
            const int64_t u0 = col[k];
            const double2 x0 = *reinterpret_cast<const double2 *>(rows + u0 * row_stride + 2 * part);
            a0 += x0.x; a1 += x0.y;
        }
#pragma unroll
        for (int off = CL; off < G; off <<= 1) {
            a0 += __shfl_xor(a0, off, G);
            a1 += __shfl_xor(a1, off, G);
        }
        if (slot == 0) {
            const double cnt = (double)d;
            const int c0 = 2 * part, c1 = 2 * part + 1;
            if (c0 < f) {
                if (out_sum) out_sum[(int64_t)c0 * ld + v] = a0;
                if (out_mean) out_mean[(int64_t)c0 * ld + v] = (d > 0) ? a0 / cnt : 0.0;
            }
            if (c1 < f) {
                if (out_sum) out_sum[(int64_t)c1 * ld + v] = a1;
                if (out_mean) out_mean[(int64_t)c1 * ld + v] = (d > 0) ? a1 / cnt : 0.0;
            }
        }
    }
}

// One workgroup per high-degree row.  The rows come from a host-built list (hub_rows, ascending)
// so that the hubs -- which cluster at low indices in preferential-attachment graphs -- spread
// over the whole chip instead of queueing behind each other.
template <int FP>
__global__ __launch_bounds__(256) void aggregate_hub_kernel(
    const int64_t *__restrict__ row_ptr, const int32_t *__restrict__ col,
    const double *__restrict__ rows, int64_t ldr, int f, int64_t row_begin, int64_t row_end,
    const int32_t *__restrict__ hub_rows, int64_t n_hubs, double *__restrict__ out_sum,
    double *__restrict__ out_mean, int64_t ld)
{
    __shared__ double red[4][FP];
    for (int64_t h = blockIdx.x; h < n_hubs; h += gridDim.x) {
        const int64_t v = hub_rows[h];
        if (v < row_begin || v >= row_end) continue;            // uniform over the workgroup
        const int64_t b = row_ptr[v], e = row_ptr[v + 1];
        double acc[FP];
#pragma unroll
        for (int c = 0; c < FP; ++c) acc[c] = 0.0;
        for (int64_t k = b + threadIdx.x; k < e; k += 256) {
            const int64_t u = col[k];
            const double2 *p = reinterpret_cast<const double2 *>(rows + u * ldr);
#pragma unroll
            for (int c = 0; c < FP / 2; ++c) {
                const double2 x = p[c];
                acc[2 * c] += x.x; acc[2 * c + 1] += x.y;
            }
        }
#pragma unroll
        for (int c = 0; c < FP; ++c) acc[c] = grx_group_sum<64>(acc[c]);
        if ((threadIdx.x & 63) == 0) {
#pragma unroll
            for (int c = 0; c < FP; ++c) red[threadIdx.x >> 6][c] = acc[c];
        }
        __syncthreads();
        if (threadIdx.x < f) {
            const int c = threadIdx.x;
            const double sm = ((red[0][c] + red[1][c]) + red[2][c]) + red[3][c];
            if (out_sum) out_sum[(int64_t)c * ld + v] = sm;
            if (out_mean) out_mean[(int64_t)c * ld + v] = sm / (double)(e - b);
        }
        __syncthreads();
    }
}

template <int LDR>
int launch_aggregate(int G, const int64_t *row_ptr, const int32_t *col, const double *rows,
                     int64_t row_stride, int f, int64_t rb, int64_t re, int64_t hub_deg,
                     const int32_t *hub_rows, int64_t n_hubs, double *s, double *m, int64_t ld,
                     hipStream_t st)
{
    constexpr int CL = LDR / 2;
    if (G < CL) G = CL;
    const int64_t nrows = re - rb;
    const int64_t want = grx_ceil_div(nrows * G, 256);
    const int grid = (int)(want < 1 ? 1 : (want > GRX_NUM_CU * 32 ? GRX_NUM_CU * 32 : want));
    {
        GRX_PROF(GRX_K_AGGREGATE, st);
        if (G <= 4 && CL <= 4) aggregate_kernel<LDR, (CL > 4 ? CL : 4)><<<grid, 256, 0, st>>>(row_ptr, col, rows, row_stride, f, rb, re, hub_deg, s, m, ld);
        else if (G <= 8 && CL <= 8) aggregate_kernel<LDR, (CL > 8 ? CL : 8)><<<grid, 256, 0, st>>>(row_ptr, col, rows, row_stride, f, rb, re, hub_deg, s, m, ld);
        else if (G <= 16) aggregate_kernel<LDR, 16><<<grid, 256, 0, st>>>(row_ptr, col, rows, row_stride, f, rb, re, hub_deg, s, m, ld);
        else aggregate_kernel<LDR, 32><<<grid, 256, 0, st>>>(row_ptr, col, rows, row_stride, f, rb, re, hub_deg, s, m, ld);
    }
    GRX_LAUNCH_CHECK();
    if (n_hubs > 0) {
        const int hgrid = (int)(n_hubs > GRX_NUM_CU * 8 ? GRX_NUM_CU * 8 : n_hubs);
        GRX_PROF(GRX_K_AGGREGATE_HUB, st);
        aggregate_hub_kernel<LDR><<<hgrid, 256, 0, st>>>(row_ptr, col, rows, row_stride, f, rb, re, hub_rows, n_hubs, s, m, ld);
    }
    GRX_LAUNCH_CHECK();
    return GRX_OK;
}

}  // namespace

extern "C" {

int grx_row_sums(int64_t n, const int64_t *d_row_ptr, const int32_t *d_col, const double *d_w,
                 int add_self_loop, int64_t row_begin, int64_t row_end, double *d_out, void *stream)
{
    GRX_REQUIRE(n >= 0 && row_begin >= 0 && row_begin <= row_end && row_end <= n,
                "grx_row_sums: bad row range [%lld,%lld) for n=%lld", (long long)row_begin,
                (long long)row_end, (long long)n);
    if (row_end == row_begin) return GRX_OK;
    GRX_REQUIRE(d_row_ptr && d_col && d_out, "grx_row_sums: NULL pointer");
    const int64_t nrows = row_end - row_begin;
    const int64_t want = grx_ceil_div(nrows * 8, 256);
    const int grid = (int)(want > GRX_NUM_CU * 16 ? GRX_NUM_CU * 16 : want);
    { GRX_PROF(GRX_K_ROW_SUMS, grx_stream(stream));
    row_sums_kernel<8><<<grid, 256, 0, grx_stream(stream)>>>(d_row_ptr, d_col, d_w, add_self_loop,
                                                            row_begin, row_end, d_out);
    }
    GRX_LAUNCH_CHECK();
    return GRX_OK;
}

int grx_add_columns(int64_t n, const double *d_a, const double *d_b, double *d_out, void *stream)
{
    GRX_REQUIRE(n >= 0, "grx_add_columns: n < 0");
    if (n == 0) return GRX_OK;
    GRX_REQUIRE(d_a && d_b && d_out, "grx_add_columns: NULL pointer");
    const int64_t want = grx_ceil_div(n, 256 * 4);
    { GRX_PROF(GRX_K_ADD_COLUMNS, grx_stream(stream));
    add_columns_kernel<<<(int)(want > 2048 ? 2048 : want), 256, 0, grx_stream(stream)>>>(n, d_a, d_b, d_out);
    }
    GRX_LAUNCH_CHECK();
    return GRX_OK;
}

int grx_egonet_features(int64_t n, const int64_t *d_row_ptr, const int32_t *d_col,
                        const double *d_w, const double *d_rowsum, int directed,
                        int64_t row_begin, int64_t row_end, double *d_internal,
                        double *d_external, void *stream)
{
    GRX_REQUIRE(n >= 0 && row_begin >= 0 && row_begin <= row_end && row_end <= n,
                "grx_egonet_features: bad row range");
    if (row_end == row_begin) return GRX_OK;
    GRX_REQUIRE(d_row_ptr && d_col && d_internal && d_external, "grx_egonet_features: NULL pointer");
    GRX_REQUIRE(d_w == nullptr || d_rowsum != nullptr,
                "grx_egonet_features: weighted graphs need d_rowsum (grx_row_sums, add_self_loop=0)");
    const int64_t nrows = row_end - row_begin;
    constexpr int64_t HUB = 512;             // members handled by a 512-thread workgroup above this
    {
        const int64_t want = grx_ceil_div(nrows, 4);
        const int grid = (int)(want > GRX_NUM_CU * 32 ? GRX_NUM_CU * 32 : want);
        { GRX_PROF(GRX_K_EGONET_WAVE, grx_stream(stream));
        egonet_kernel<64><<<grid, 256, 0, grx_stream(stream)>>>(
            d_row_ptr, d_col, d_w, d_rowsum, directed, row_begin, row_end, 0, HUB, d_internal,
            d_external);
        }
        GRX_LAUNCH_CHECK();
    }
    {
        const int grid = (int)(nrows > GRX_NUM_CU * 8 ? GRX_NUM_CU * 8 : nrows);
        { GRX_PROF(GRX_K_EGONET_BLOCK, grx_stream(stream));
        egonet_kernel<512><<<grid, 512, 0, grx_stream(stream)>>>(
            d_row_ptr, d_col, d_w, d_rowsum, directed, row_begin, row_end, HUB,
            (int64_t)1 << 62, d_internal, d_external);
        }
        GRX_LAUNCH_CHECK();
    }
    return GRX_OK;
}

int grx_triangle_counts(int64_t n, const int64_t *d_o_row_ptr, const int32_t *d_o_col, int64_t row_begin,
                        int64_t row_end, uint64_t *d_T, void *stream)
{
    GRX_REQUIRE(n >= 0 && row_begin >= 0 && row_begin <= row_end && row_end <= n, "grx_triangle_counts: bad row range");
    if (row_end == row_begin) return GRX_OK;
    GRX_REQUIRE(d_o_row_ptr && d_o_col && d_T, "grx_triangle_counts: NULL pointer");
    const int64_t want = grx_ceil_div((row_end - row_begin) * 8, 256);
    const int grid = (int)(want > GRX_NUM_CU * 32 ? GRX_NUM_CU * 32 : want);
    { GRX_PROF(GRX_K_TRIANGLES, grx_stream(stream));
    triangle_count_kernel<<<grid, 256, 0, grx_stream(stream)>>>(d_o_row_ptr, d_o_col, row_begin, row_end,
                                                                reinterpret_cast<unsigned long long *>(d_T));
    }
    GRX_LAUNCH_CHECK();
    return GRX_OK;
}

int grx_egonet_unweighted(int64_t n, const int64_t *d_row_ptr, const int32_t *d_col, const uint64_t *d_T,
                          int64_t row_begin, int64_t row_end, double *d_internal, double *d_external,
                          int32_t *d_scratch, const int32_t *d_hub_rows, int64_t n_hub_rows, int64_t hub_degree,
                          void *stream)
{
    GRX_REQUIRE(n >= 0 && row_begin >= 0 && row_begin <= row_end && row_end <= n, "grx_egonet_unweighted: bad row range");
    if (n == 0) return GRX_OK;
    GRX_REQUIRE(d_row_ptr && d_col && d_T && d_internal && d_external && d_scratch, "grx_egonet_unweighted: NULL pointer");
    hipStream_t st = grx_stream(stream);
    {
        const int64_t want = grx_ceil_div(n, 256);
        GRX_PROF(GRX_K_EGONET_FINISH, st);
        node_info_kernel<<<(int)(want > GRX_NUM_CU * 16 ? GRX_NUM_CU * 16 : want), 256, 0, st>>>(n, d_row_ptr, d_col, d_scratch);
    }
    GRX_LAUNCH_CHECK();
    if (row_end > row_begin) {
        const int64_t want = grx_ceil_div((row_end - row_begin) * 8, 256);
        GRX_PROF(GRX_K_EGONET_FINISH, st);
        const int64_t hub_deg = (d_hub_rows && n_hub_rows > 0) ? hub_degree : ((int64_t)1 << 62);
        egonet_from_triangles_kernel<<<(int)(want > GRX_NUM_CU * 32 ? GRX_NUM_CU * 32 : want), 256, 0, st>>>(
            d_row_ptr, d_col, d_scratch, reinterpret_cast<const unsigned long long *>(d_T), row_begin, row_end,
            hub_deg, d_internal, d_external);
        if (d_hub_rows && n_hub_rows > 0) {
            const int hgrid = (int)(n_hub_rows > GRX_NUM_CU * 8 ? GRX_NUM_CU * 8 : n_hub_rows);
            egonet_from_triangles_hub_kernel<<<hgrid, 256, 0, st>>>(
                d_row_ptr, d_col, d_scratch, reinterpret_cast<const unsigned long long *>(d_T), row_begin,
                row_end, d_hub_rows, n_hub_rows, d_internal, d_external);
        }
    }
    GRX_LAUNCH_CHECK();
    return GRX_OK;
}

int grx_pack_rows(int64_t n, int f, const double *const *h_col_ptrs, double *d_rows, int ldr,
                  void *stream)
{
    if (f > GRX_MAX_PTRS) {
        grx_set_error("grx_pack_rows: f=%d > %d columns per call", f, GRX_MAX_PTRS);
        return GRX_ERR_UNSUPPORTED;
    }
    GRX_REQUIRE(n >= 0 && f >= 0 && ldr >= f, "grx_pack_rows: bad shape n=%lld f=%d ldr=%d",
                (long long)n, f, ldr);
    if (n == 0 || ldr == 0) return GRX_OK;
    GRX_REQUIRE(h_col_ptrs && d_rows, "grx_pack_rows: NULL pointer");
    GrxPtrTable tab;
    for (int c = 0; c < f; ++c) tab.p[c] = h_col_ptrs[c];
    const int64_t want = grx_ceil_div(n, 256);
    const int grid = (int)(want > GRX_NUM_CU * 16 ? GRX_NUM_CU * 16 : want);
    { GRX_PROF(GRX_K_PACK_ROWS, grx_stream(stream));
    pack_rows_kernel<<<grid, 256, 0, grx_stream(stream)>>>(n, f, ldr, tab, d_rows);
    }
    GRX_LAUNCH_CHECK();
    return GRX_OK;
}

int grx_aggregate(int64_t n, const int64_t *d_row_ptr, const int32_t *d_col, int f,
                  const double *d_rows, int ldr, int64_t row_begin, int64_t row_end,
                  double *d_sum, double *d_mean, int64_t ld, int lanes_per_row,
                  const int32_t *d_hub_rows, int64_t n_hub_rows, void *stream)
{
    GRX_REQUIRE(n >= 0 && row_begin >= 0 && row_begin <= row_end && row_end <= n,
                "grx_aggregate: bad row range");
    GRX_REQUIRE(f >= 0 && ldr >= f, "grx_aggregate: ldr=%d < f=%d", ldr, f);
    GRX_REQUIRE(ldr == 2 || ldr == 4 || ldr == 8 || (ldr >= 16 && ldr % 16 == 0),
                "grx_aggregate: ldr=%d must be 2, 4, 8 or a multiple of 16 (use grx_aggregate_ldr)", ldr);
    if (row_end == row_begin || f == 0) return GRX_OK;
    GRX_REQUIRE(d_row_ptr && d_col && d_rows, "grx_aggregate: NULL pointer");
    GRX_REQUIRE(ld >= n, "grx_aggregate: ld < n");
    GRX_REQUIRE((reinterpret_cast<uintptr_t>(d_rows) & 127) == 0, "grx_aggregate: d_rows must be 128-byte aligned");
    int G = lanes_per_row;
    if (G != 4 && G != 8 && G != 16 && G != 32) G = 8;
    GRX_REQUIRE(n_hub_rows >= 0 && (n_hub_rows == 0 || d_hub_rows != nullptr), "grx_aggregate: bad hub list");
    // rows longer than this are expected in d_hub_rows; without a list every row takes the
    // lane-group path (correct, but hubs then serialise on one lane group)
    const int64_t hub_deg = d_hub_rows ? (int64_t)G * GRX_HUB_FACTOR : ((int64_t)1 << 62);
    hipStream_t st = grx_stream(stream);
    if (ldr < 16)
        switch (ldr) {
        case 2:  return launch_aggregate<2>(G, d_row_ptr, d_col, d_rows, ldr, f, row_begin, row_end, hub_deg, d_hub_rows, n_hub_rows, d_sum, d_mean, ld, st);
        case 4:  return launch_aggregate<4>(G, d_row_ptr, d_col, d_rows, ldr, f, row_begin, row_end, hub_deg, d_hub_rows, n_hub_rows, d_sum, d_mean, ld, st);
        default: return launch_aggregate<8>(G, d_row_ptr, d_col, d_rows, ldr, f, row_begin, row_end, hub_deg, d_hub_rows, n_hub_rows, d_sum, d_mean, ld, st);
        }
    // wide rows: 16 columns (one 128-byte segment of every row) per launch
    for (int c0 = 0; c0 < f; c0 += 16) {
        const int fc = (f - c0 < 16) ? (f - c0) : 16;
        double *s = d_sum ? d_sum + (int64_t)c0 * ld : nullptr;
        double *m = d_mean ? d_mean + (int64_t)c0 * ld : nullptr;
        int rc = launch_aggregate<16>(G, d_row_ptr, d_col, d_rows + c0, ldr, fc, row_begin, row_end, hub_deg,
                                      d_hub_rows, n_hub_rows, s, m, ld, st);
        if (rc != GRX_OK) return rc;
    }
    return GRX_OK;
}

/* row stride (in doubles) grx_pack_rows / grx_aggregate use for f columns */
int grx_aggregate_ldr(int f)
{
    if (f <= 2) return 2;
    if (f <= 4) return 4;
    if (f <= 8) return 8;
    return (f + 15) / 16 * 16;
}

}  // extern "C"
