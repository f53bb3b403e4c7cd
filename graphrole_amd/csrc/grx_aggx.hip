// grx_aggx.hip -- the aggregations of RecursiveFeatureExtractor(aggs=[...]) beyond sum / mean / min / max / var / std
// (graphrole/features/extract.py:26,47,111 hands ANY pandas-aggregatable to DataFrame.agg):
//
//   * INTEGER feature columns with the reference's int64 semantics.  With 'prod' among the aggs the reference's
//     integer columns (generation 0 of an unweighted graph and everything derived from it by sum / prod / min / max /
//     count) overflow 2^53 within a generation or two and WRAP modulo 2^64 silently (numpy int64 arithmetic); an fp64
//     column cannot follow that.  Such columns are carried as int64 BITS in the 8-byte column slots (pack, exchange
//     and permute kernels move bits) and aggregated here in wrapping integer arithmetic; grx_convert_* switches
//     representation (numpy astype: round to nearest), the binning kernels take a per-column "is int64" flag.
//   * 'median' (pandas nanmedian -> numpy median: middle element, or (a + b) / 2 of the two middle elements), by
//     gathering every row's neighbour values into edge order once and selecting inside each segment: rank counting
//     in a wavefront up to 64 neighbours, an 8-pass radix selection (LDS histogram per wavefront) above.
//   * 'count' / 'size': the number of neighbours.
// API completeness, not a tuned path: one lane per (row, column) for the integer
// aggregations, one wavefront per row for the median.
#include "grx_common.h"

namespace {

__global__ __launch_bounds__(256) void convert_i64_f64_kernel(int64_t n, const long long *__restrict__ in, double *__restrict__ out)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) out[i] = (double)in[i];
}

__global__ __launch_bounds__(256) void convert_f64_i64_kernel(int64_t n, const double *__restrict__ in, long long *__restrict__ out)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) out[i] = (long long)in[i];
}

// one lane per (row, column): wrapping sum / product, min, max over the neighbours' int64 values
__global__ __launch_bounds__(256) void aggregate_i64_kernel(const int64_t *__restrict__ row_ptr, const int32_t *__restrict__ col,
                                                            const long long *__restrict__ rows, int64_t row_stride, int f,
                                                            int64_t row_begin, int64_t row_end, long long *__restrict__ out_sum,
                                                            long long *__restrict__ out_prod, long long *__restrict__ out_min,
                                                            long long *__restrict__ out_max, int64_t ld)
{
    const int64_t total = (row_end - row_begin) * f;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int64_t v = row_begin + i / f;
        const int c = (int)(i % f);
        unsigned long long s = 0, p = 1;                         // unsigned: wrapping is defined
        long long lo = 0x7fffffffffffffffll, hi = -0x7fffffffffffffffll - 1;
        const int64_t b = row_ptr[v], e = row_ptr[v + 1];
        for (int64_t k = b; k < e; ++k) {
            const long long x = rows[(int64_t)col[k] * row_stride + c];
            s += (unsigned long long)x;
            p *= (unsigned long long)x;
            lo = x < lo ? x : lo;
            hi = x > hi ? x : hi;
        }
        const int64_t o = (int64_t)c * ld + v;
        if (out_sum) out_sum[o] = (long long)s;
        if (out_prod) out_prod[o] = (long long)p;
        if (out_min) out_min[o] = e > b ? lo : 0;                // min / max of nothing: NaN -> fillna(0), extract.py:113
        if (out_max) out_max[o] = e > b ? hi : 0;
    }
}

__global__ __launch_bounds__(256) void count_kernel(const int64_t *__restrict__ row_ptr, int64_t row_begin, int64_t row_end,
                                                    int f, int as_i64, double *__restrict__ out, int64_t ld)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t v = row_begin + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < row_end; v += stride) {
        const long long d = row_ptr[v + 1] - row_ptr[v];
        for (int c = 0; c < f; ++c) {
            if (as_i64) reinterpret_cast<long long *>(out)[(int64_t)c * ld + v] = d;
            else out[(int64_t)c * ld + v] = (double)d;
        }
    }
}

// ---- median -----------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t med_key(double x)
{
    const uint64_t b = (uint64_t)__double_as_longlong(x);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);        // total order of the doubles as unsigned integers
}

__device__ __forceinline__ double med_val(uint64_t k)
{
    const uint64_t b = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
    return __longlong_as_double((long long)b);
}

// vals[e] = order key of column c of neighbour col[e] for the adjacency entries of rows [row_begin, row_end)
__global__ __launch_bounds__(256) void med_gather_kernel(const int32_t *__restrict__ col, const double *__restrict__ rows,
                                                         int64_t row_stride, int c, int64_t e_begin, int64_t e_end,
                                                         uint64_t *__restrict__ vals)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t e = e_begin + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < e_end; e += stride)
        vals[e - e_begin] = med_key(rows[(int64_t)col[e] * row_stride + c] + 0.0);
}

// one wavefront per row: the k-th smallest (0-based) of keys[0..d), and how many keys are <= it
__device__ __forceinline__ uint64_t med_select(const uint64_t *__restrict__ keys, int64_t d, int64_t k, int64_t *n_le,
                                               uint32_t *hist /* 256 per wavefront */)
{
    const int lane = threadIdx.x & 63;
    if (d <= 64) {
        const uint64_t mine = lane < d ? keys[lane] : ~0ull;
        int lt = 0, le = 0;
        for (int j = 0; j < (int)d; ++j) {
            const uint64_t o = __shfl(mine, j, 64);
            lt += o < mine;
            le += o <= mine;
        }
        const uint64_t hit = __ballot(lane < d && lt <= k && k < le);
        const int src = __ffsll((long long)hit) - 1;
        *n_le = __shfl(le, src, 64);
        return __shfl(mine, src, 64);
    }
    // radix selection, most significant byte first; `prefix` fixes the bytes above the current one
    uint64_t prefix = 0;
    int64_t below = 0, krem = k, equal = d;
    for (int byte = 7; byte >= 0; --byte) {
        for (int j = lane; j < 256; j += 64) hist[j] = 0;
        __threadfence_block();
        const int hs = 8 * (byte + 1);
        for (int64_t i = lane; i < d; i += 64) {
            const uint64_t v = keys[i];
            if (byte == 7 || (v >> hs) == (prefix >> hs)) atomicAdd(&hist[(uint32_t)(v >> (8 * byte)) & 0xFF], 1u);
        }
        __threadfence_block();
        // every lane walks the histogram alike (256 LDS reads; the rows that come here are long)
        int64_t cum = 0;
        int dsel = 0;
        for (; dsel < 255; ++dsel) {
            const int64_t h = hist[dsel];
            if (cum + h > krem) break;
            cum += h;
        }
        prefix |= (uint64_t)dsel << (8 * byte);
        below += cum;
        krem -= cum;
        equal = hist[dsel];
        __threadfence_block();
    }
    *n_le = below + equal;
    return prefix;
}

__global__ __launch_bounds__(256) void med_select_kernel(const int64_t *__restrict__ row_ptr, int64_t row_begin, int64_t row_end,
                                                         int64_t e_begin, const uint64_t *__restrict__ vals,
                                                         double *__restrict__ out)
{
    __shared__ uint32_t s_hist[4][256];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t nw = (int64_t)gridDim.x * 4;
    for (int64_t v = row_begin + (int64_t)blockIdx.x * 4 + wave; v < row_end; v += nw) {
        const int64_t b = row_ptr[v], d = row_ptr[v + 1] - b;
        double res = 0.0;                                        // no neighbours: NaN -> fillna(0)
        if (d > 0) {
            const uint64_t *keys = vals + (b - e_begin);
            int64_t n_le = 0;
            const int64_t k = (d - 1) / 2;                       // lower middle
            const uint64_t ka = med_select(keys, d, k, &n_le, s_hist[wave]);
            const double a = med_val(ka);
            if (d & 1) {
                res = a;
            } else {
                double bval = a;
                if (n_le < k + 2) {                              // the upper middle is the smallest key above ka
                    uint64_t mn = ~0ull;
                    for (int64_t i = lane; i < d; i += 64) {
                        const uint64_t x = keys[i];
                        if (x > ka && x < mn) mn = x;
                    }
#pragma unroll
                    for (int off = 32; off > 0; off >>= 1) {
                        const uint64_t o = __shfl_xor(mn, off, 64);
                        mn = o < mn ? o : mn;
                    }
                    bval = med_val(mn);
                }
                res = (a + bval) / 2.0;                          // numpy: mean of the two middle elements
            }
        }
        if (lane == 0) out[v] = res;
    }
}

int grid_for(int64_t items)
{
    const int64_t want = grx_ceil_div(items, 256);
    return (int)(want < 1 ? 1 : (want > GRX_NUM_CU * 32 ? GRX_NUM_CU * 32 : want));
}

}  // namespace

extern "C" {

int grx_convert_i64_to_f64(int64_t n, const int64_t *d_in, double *d_out, void *stream)
{
    if (n <= 0) return GRX_OK;
    GRX_REQUIRE(d_in && d_out, "grx_convert_i64_to_f64: NULL pointer");
    convert_i64_f64_kernel<<<grid_for(n), 256, 0, grx_stream(stream)>>>(n, reinterpret_cast<const long long *>(d_in), d_out);
    GRX_LAUNCH_CHECK();
    return GRX_OK;
}

int grx_convert_f64_to_i64(int64_t n, const double *d_in, int64_t *d_out, void *stream)
{
    if (n <= 0) return GRX_OK;
    GRX_REQUIRE(d_in && d_out, "grx_convert_f64_to_i64: NULL pointer");
    convert_f64_i64_kernel<<<grid_for(n), 256, 0, grx_stream(stream)>>>(n, d_in, reinterpret_cast<long long *>(d_out));
    GRX_LAUNCH_CHECK();
    return GRX_OK;
}

int grx_aggregate_i64(const int64_t *d_row_ptr, const int32_t *d_col, int f, const int64_t *d_rows, int ldr,
                      int64_t row_begin, int64_t row_end, int64_t *d_sum, int64_t *d_prod, int64_t *d_min, int64_t *d_max,
                      int64_t ld, void *stream)
{
    GRX_REQUIRE(f >= 0 && ldr >= f && row_begin >= 0 && row_begin <= row_end && ld >= row_end, "grx_aggregate_i64: bad shape");
    if (f == 0 || row_end == row_begin) return GRX_OK;
    GRX_REQUIRE(d_row_ptr && d_col && d_rows, "grx_aggregate_i64: NULL pointer");
    aggregate_i64_kernel<<<grid_for((row_end - row_begin) * f), 256, 0, grx_stream(stream)>>>(
        d_row_ptr, d_col, reinterpret_cast<const long long *>(d_rows), ldr, f, row_begin, row_end,
        reinterpret_cast<long long *>(d_sum), reinterpret_cast<long long *>(d_prod), reinterpret_cast<long long *>(d_min),
        reinterpret_cast<long long *>(d_max), ld);
    GRX_LAUNCH_CHECK();
    return GRX_OK;
}

int grx_aggregate_count(const int64_t *d_row_ptr, int f, int64_t row_begin, int64_t row_end, int as_i64, double *d_out,
                        int64_t ld, void *stream)
{
    GRX_REQUIRE(f >= 0 && row_begin >= 0 && row_begin <= row_end && ld >= row_end, "grx_aggregate_count: bad shape");
    if (f == 0 || row_end == row_begin) return GRX_OK;
    GRX_REQUIRE(d_row_ptr && d_out, "grx_aggregate_count: NULL pointer");
    count_kernel<<<grid_for(row_end - row_begin), 256, 0, grx_stream(stream)>>>(d_row_ptr, row_begin, row_end, f, as_i64, d_out, ld);
    GRX_LAUNCH_CHECK();
    return GRX_OK;
}

size_t grx_aggregate_median_workspace_bytes(int64_t nnz) { return (size_t)(nnz > 0 ? nnz : 1) * 8 + 256; }

int grx_aggregate_median(const int64_t *d_row_ptr, const int32_t *d_col, int f, const double *d_rows, int ldr,
                         int64_t row_begin, int64_t row_end, double *d_median, int64_t ld, void *d_workspace,
                         size_t workspace_bytes, void *stream)
{
    GRX_REQUIRE(f >= 0 && ldr >= f && row_begin >= 0 && row_begin <= row_end && ld >= row_end, "grx_aggregate_median: bad shape");
    if (f == 0 || row_end == row_begin) return GRX_OK;
    GRX_REQUIRE(d_row_ptr && d_col && d_rows && d_median && d_workspace, "grx_aggregate_median: NULL pointer");
    // the adjacency range of the rows (two row pointers): one small read-back, this is not a tuned path
    int64_t range[2] = {0, 0};
    GRX_CHECK_HIP(hipMemcpyAsync(&range[0], d_row_ptr + row_begin, 8, hipMemcpyDeviceToHost, grx_stream(stream)));
    GRX_CHECK_HIP(hipMemcpyAsync(&range[1], d_row_ptr + row_end, 8, hipMemcpyDeviceToHost, grx_stream(stream)));
    GRX_CHECK_HIP(hipStreamSynchronize(grx_stream(stream)));
    const int64_t e_begin = range[0], e_end = range[1];
    GRX_REQUIRE(e_end >= e_begin, "grx_aggregate_median: bad adjacency range");
    GRX_REQUIRE(workspace_bytes >= grx_aggregate_median_workspace_bytes(e_end - e_begin), "grx_aggregate_median: workspace too small");
    hipStream_t st = grx_stream(stream);
    uint64_t *vals = reinterpret_cast<uint64_t *>(d_workspace);
    const int64_t want = grx_ceil_div(row_end - row_begin, 4);
    const int sgrid = (int)(want > GRX_NUM_CU * 32 ? GRX_NUM_CU * 32 : (want < 1 ? 1 : want));
    for (int c = 0; c < f; ++c) {
        if (e_end > e_begin)
            med_gather_kernel<<<grid_for(e_end - e_begin), 256, 0, st>>>(d_col, d_rows, ldr, c, e_begin, e_end, vals);
        med_select_kernel<<<sgrid, 256, 0, st>>>(d_row_ptr, row_begin, row_end, e_begin, vals, d_median + (int64_t)c * ld);
        GRX_LAUNCH_CHECK();
    }
    return GRX_OK;
}

}  // extern "C"
