// Microbenchmark: random cache-line gather rate of MI355X (what bounds grx_aggregate).
// Each lane group of CL lanes reads one random LINE-byte row (16 B per lane) of a table of
// `rows` rows; UNROLL independent rows in flight per lane.  Prints G rows/s and GB/s of useful
// bytes for several table sizes (L2-resident .. beyond Infinity Cache).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <int CL, int UNROLL>
__global__ __launch_bounds__(256) void gather(const double2 *__restrict__ table, const int *__restrict__ idx,
                                             long n_idx, double *__restrict__ out)
{
    const long lane_global = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int part = threadIdx.x % CL;
    const long slot = lane_global / CL, nslots = (long)gridDim.x * blockDim.x / CL;
    double a = 0.0, b = 0.0;
    for (long k = slot * UNROLL; k + UNROLL <= n_idx; k += nslots * UNROLL) {
        int u[UNROLL];
#pragma unroll
        for (int j = 0; j < UNROLL; ++j) u[j] = idx[k + j];
        double2 x[UNROLL];
#pragma unroll
        for (int j = 0; j < UNROLL; ++j) x[j] = table[(long)u[j] * CL + part];
#pragma unroll
        for (int j = 0; j < UNROLL; ++j) { a += x[j].x; b += x[j].y; }
    }
    if (a + b == 12345.678) out[0] = a + b;
}

template <int CL, int UNROLL>
void run(long rows, long n_idx, const int *d_idx, const double2 *d_table, double *d_out)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int grid = 256 * 32;
    gather<CL, UNROLL><<<grid, 256>>>(d_table, d_idx, n_idx, d_out);
    hipEventRecord(e0);
    for (int it = 0; it < 5; ++it) gather<CL, UNROLL><<<grid, 256>>>(d_table, d_idx, n_idx, d_out);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    printf("rows=%9ld (%6.1f MB)  line=%3dB unroll=%d : %7.3f ms  %6.1f Grows/s  %7.1f GB/s useful\n", rows,
           rows * CL * 16 / 1e6, CL * 16, UNROLL, ms, n_idx / ms / 1e6, n_idx * CL * 16.0 / ms / 1e6);
}

int main()
{
    const long n_idx = 20000000;
    for (long rows : {32768L, 262144L, 1000000L, 4000000L, 16000000L}) {
        std::vector<int> h(n_idx);
        unsigned long long s = 88172645463325252ull;
        for (long i = 0; i < n_idx; ++i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; h[i] = (int)(s % rows); }
        int *d_idx; double2 *d_table; double *d_out;
        hipMalloc(&d_idx, n_idx * 4); hipMalloc(&d_table, rows * 64); hipMalloc(&d_out, 8);
        hipMemcpy(d_idx, h.data(), n_idx * 4, hipMemcpyHostToDevice);
        hipMemset(d_table, 0, rows * 64);
        run<4, 2>(rows, n_idx, d_idx, d_table, d_out);
        run<4, 4>(rows, n_idx, d_idx, d_table, d_out);
        run<4, 8>(rows, n_idx, d_idx, d_table, d_out);
        run<2, 4>(rows, n_idx, d_idx, d_table, d_out);
        run<1, 8>(rows, n_idx, d_idx, d_table, d_out);
        hipFree(d_idx); hipFree(d_table); hipFree(d_out);
    }
    return 0;
}
