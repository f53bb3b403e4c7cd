// grx_roles.hip -- the two row passes over the node-role factor that finish RolX:
//   grx_role_argmax     RoleExtractor.roles            graphrole/roles/extract.py:38-47  (DataFrame.idxmax(axis=1))
//   grx_row_normalise   RoleExtractor.role_percentage  graphrole/roles/extract.py:49-57  (row / row.sum() per row)
//
// The factor is the n x r row-major matrix the reference holds (any r: a fitted factor has r <= GRX_MAX_ROLES, a frame
// the caller assigned may be wider -- round 5): 8 r bytes per node in, 4 (or
// 8 r) bytes out -- pure HBM streaming.  A workgroup stages 128 consecutive rows through LDS so that the global
// reads and writes are coalesced 16-byte-per-lane streams whatever r is; one lane then owns one row in LDS (row
// stride padded to an odd number of doubles: conflict-free ds_read_b64).
//
// Exactness.  With 2^n_bits-level quantised factors most rows hold exact ties, so "first maximum wins" IS the
// result (pandas: nanargmax = NaN -> -inf, then numpy argmax).  The row sum follows the order Series.sum() uses
// for r contiguous doubles -- numpy's pairwise_sum: fewer than 8 values are added left to right, 8 to 128 run
// eight strided accumulators, ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)), then the r % 8 trailing values one by one;
// more than 128 values are halved recursively (numpy_row_sum);
// NaN counts as 0 in the sum (nanops.nansum) and stays NaN in the quotient.  0 / 0 is NaN like the reference's.
#include "grx_common.h"

#include <cmath>

namespace {

constexpr int ROWS_PER_WG = 128;        // 128 x (32 | 1) doubles = 33 KB of LDS at r = 32; wider factors: fewer rows per tile
constexpr size_t TILE_BYTES_MAX = 60u << 10;
constexpr int PW_BLOCK = 128;           // numpy's PW_BLOCKSIZE

// numpy's pairwise_sum for at most PW_BLOCK values (NaN counts as 0: nanops.nansum)
__device__ __forceinline__ double numpy_block_sum(const double *__restrict__ a, int r)
{
#pragma clang fp contract(off)
    auto val = [&](int i) { const double x = a[i]; return x != x ? 0.0 : x; };
    if (r < 8) {
        double res = 0.0;
        for (int i = 0; i < r; ++i) res += val(i);
        return res;
    }
    double acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = val(j);
    int i = 8;
    for (; i < r - (r % 8); i += 8) {
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] += val(i + j);
    }
    double res = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
    for (; i < r; ++i) res += val(i);
    return res;
}

// ... and beyond: pairwise_sum(a, n) = pairwise_sum(a, n2) + pairwise_sum(a + n2, n - n2) with n2 = n / 2 rounded down
// to a multiple of 8, as a post-order walk with an explicit stack (a row of a wide, user-assigned factor)
__device__ double numpy_pairwise_sum(const double *__restrict__ a, int r)
{
#pragma clang fp contract(off)
    if (r <= PW_BLOCK) return numpy_block_sum(a, r);
    int off[32], len[32];
    unsigned char state[32];              // 0: descend left, 1: left done, 2: right done
    double left[32];
    int sp = 0;
    off[0] = 0; len[0] = r; state[0] = 0;
    double ret = 0.0;
    while (sp >= 0) {
        if (len[sp] <= PW_BLOCK) { ret = numpy_block_sum(a + off[sp], len[sp]); --sp; continue; }
        int n2 = len[sp] / 2;
        n2 -= n2 % 8;
        if (state[sp] == 0) {
            state[sp] = 1;
            off[sp + 1] = off[sp]; len[sp + 1] = n2; state[sp + 1] = 0;
            ++sp;
        } else if (state[sp] == 1) {
            left[sp] = ret;
            state[sp] = 2;
            off[sp + 1] = off[sp] + n2; len[sp + 1] = len[sp] - n2; state[sp + 1] = 0;
            ++sp;
        } else {
            ret = left[sp] + ret;
            --sp;
        }
    }
    return ret;
}

// ndarray.sum() / add.reduce of r contiguous doubles: pairwise_sum over chunks of the ufunc buffer (8192 elements),
// the chunk sums added to a running total in order (measured against numpy 2.2: rows of 20 000 values follow this, not
// one recursion over the whole row) -- the same rule grx_aggregate applies to rows beyond 8192 neighbours
constexpr int UFUNC_BUFFER = 8192;
__device__ double numpy_row_sum(const double *__restrict__ a, int r)
{
#pragma clang fp contract(off)
    double total = numpy_pairwise_sum(a, r < UFUNC_BUFFER ? r : UFUNC_BUFFER);
    for (int at = UFUNC_BUFFER; at < r; at += UFUNC_BUFFER)
        total += numpy_pairwise_sum(a + at, r - at < UFUNC_BUFFER ? r - at : UFUNC_BUFFER);
    return total;
}

// ARGMAX: d_first_max[v] = column of the first maximum of row v (-1: every entry is NaN)
// NORMALISE: d_share[v, :] = row v / its sum
template <bool ARGMAX, bool NORMALISE>
__device__ __forceinline__ void role_one_row(double *row, const double *in, int r, int32_t *first_max)
{
    if (ARGMAX) {
        double best = -INFINITY;
        int arg = 0;
        bool any = false;
        for (int c = 0; c < r; ++c) {
            const double x = in[c];
            const bool nan = x != x;
            any |= !nan;
            const double v = nan ? -INFINITY : x;
            if (c == 0 || v > best) { best = v; arg = c; }
        }
        *first_max = any ? arg : -1;
    }
    if (NORMALISE) {
        const double s = numpy_row_sum(in, r);
        for (int c = 0; c < r; ++c) row[c] = in[c] / s;
    }
}

// tile_rows rows per workgroup pass (ROWS_PER_WG unless the factor is wider than 59 columns); tile_rows == 0: rows that
// do not fit LDS at all (r > 7679: a user-assigned frame) are processed straight from global memory, one lane per row
template <bool ARGMAX, bool NORMALISE>
__global__ __launch_bounds__(ROWS_PER_WG) void role_rows_kernel(int64_t n, int r, int tile_rows, const double *__restrict__ G,
                                                                int32_t *__restrict__ d_first_max,
                                                                double *__restrict__ d_share)
{
    extern __shared__ double tile[];
    if (tile_rows == 0) {
        for (int64_t v = (int64_t)blockIdx.x * ROWS_PER_WG + threadIdx.x; v < n; v += (int64_t)gridDim.x * ROWS_PER_WG)
            role_one_row<ARGMAX, NORMALISE>(NORMALISE ? d_share + v * r : nullptr, G + v * r, r, ARGMAX ? d_first_max + v : nullptr);
        return;
    }
    const int stride = r | 1;
    for (int64_t base = (int64_t)blockIdx.x * tile_rows; base < n; base += (int64_t)gridDim.x * tile_rows) {
        const int rows = (int)((n - base) < tile_rows ? (n - base) : tile_rows);
        const double *src = G + base * r;
        for (int i = threadIdx.x; i < rows * r; i += ROWS_PER_WG) tile[(i / r) * stride + (i % r)] = src[i];
        __syncthreads();
        if ((int)threadIdx.x < rows) {
            double *row = tile + threadIdx.x * stride;
            role_one_row<ARGMAX, NORMALISE>(row, row, r, ARGMAX ? d_first_max + base + threadIdx.x : nullptr);
        }
        if (NORMALISE) {
            __syncthreads();
            double *dst = d_share + base * r;
            for (int i = threadIdx.x; i < rows * r; i += ROWS_PER_WG) dst[i] = tile[(i / r) * stride + (i % r)];
        }
        __syncthreads();
    }
}

template <bool ARGMAX, bool NORMALISE>
int launch_role_rows(int64_t n, int r, const double *d_G, int32_t *d_first_max, double *d_share, hipStream_t st)
{
    const size_t row_bytes = (size_t)(r | 1) * sizeof(double);
    int tile_rows = (int)(TILE_BYTES_MAX / row_bytes);
    if (tile_rows > ROWS_PER_WG) tile_rows = ROWS_PER_WG;
    const size_t shmem = (size_t)tile_rows * row_bytes;
    const int64_t want = grx_ceil_div(n, tile_rows ? tile_rows : ROWS_PER_WG);
    const int grid = (int)(want > GRX_NUM_CU * 16 ? GRX_NUM_CU * 16 : want);
    GRX_PROF(GRX_K_ROLE_ROWS, st);
    role_rows_kernel<ARGMAX, NORMALISE><<<grid, ROWS_PER_WG, shmem, st>>>(n, r, tile_rows, d_G, d_first_max, d_share);
    GRX_LAUNCH_CHECK();
    return GRX_OK;
}

}  // namespace

extern "C" {

int grx_role_argmax(int64_t n, int r, const double *d_G, int32_t *d_first_max, void *stream)
{
    GRX_REQUIRE(n >= 0 && r >= 1, "grx_role_argmax: n=%lld r=%d", (long long)n, r);
    if (n == 0) return GRX_OK;
    GRX_REQUIRE(d_G && d_first_max, "grx_role_argmax: NULL pointer");
    return launch_role_rows<true, false>(n, r, d_G, d_first_max, nullptr, grx_stream(stream));
}

int grx_row_normalise(int64_t n, int r, const double *d_G, double *d_share, void *stream)
{
    GRX_REQUIRE(n >= 0 && r >= 1, "grx_row_normalise: n=%lld r=%d", (long long)n, r);
    if (n == 0) return GRX_OK;
    GRX_REQUIRE(d_G && d_share, "grx_row_normalise: NULL pointer");
    return launch_role_rows<false, true>(n, r, d_G, nullptr, d_share, grx_stream(stream));
}

}  // extern "C"
