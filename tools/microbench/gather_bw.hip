// Microbenchmark: the random-row gather rate of MI355X -- the pattern that bounds grx_aggregate.
//
//   hipcc -O3 --offload-arch=gfx950 -o gather_bw gather_bw.hip && ./gather_bw > profiles/r04_gather_bw.json
//
// A lane group of CL lanes reads one ROW-byte row (16 B per lane) of a table; UNROLL independent rows are in
// flight per lane; the index stream (int32, 4 B per gathered row) is read sequentially like a CSR column array.
// Nothing is summed in order and nothing is written: this is the gather alone, i.e. an upper bound for any kernel
// that pulls one table row per CSR entry.  Two index distributions:
//   uniform   every row equally likely (Erdos-Renyi-like adjacency)
//   powerlaw  row = floor(N * U^2): on a degree-descending table of a Barabasi-Albert graph the share of edge
//             endpoints that fall into the first x rows is sqrt(x / N) (p(k) ~ 2 m^2 / k^3) -- the hubs form a hot
//             prefix, which is what lets the real kernel hit its XCD's L2 more often than a uniform stream does
// and one variant with the hot prefix of the table staged in LDS (the north_star's "LDS-staged feature tiles"):
// every workgroup copies the first K rows into its LDS, a gather of row < K is an LDS read.
//
// One JSON object per line on stdout; "best" lines carry the fastest (unroll, grid) variant of a
// (table MB, row bytes, distribution) cell -- the figure bench.py uses as the gather ceiling.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

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

// the first k_lds rows of the table live in LDS (dynamic shared memory: k_lds * CL * 16 bytes per workgroup)
template <int CL, int UNROLL>
__global__ __launch_bounds__(1024) void gather_lds(const double2 *__restrict__ table, const int *__restrict__ idx,
                                                  long n_idx, int k_lds, double *__restrict__ out)
{
    extern __shared__ double2 hot[];
    for (int i = threadIdx.x; i < k_lds * CL; i += blockDim.x) hot[i] = table[i];
    __syncthreads();
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
        for (int j = 0; j < UNROLL; ++j) {
            // the far rows first (long latency), the LDS rows are filled in below
            if (u[j] >= k_lds) x[j] = table[(long)u[j] * CL + part];
        }
#pragma unroll
        for (int j = 0; j < UNROLL; ++j) {
            if (u[j] < k_lds) x[j] = hot[u[j] * CL + part];
        }
#pragma unroll
        for (int j = 0; j < UNROLL; ++j) { a += x[j].x; b += x[j].y; }
    }
    if (a + b == 12345.678) out[0] = a + b;
}

struct Result { double ms; int unroll, grid; };

template <typename F>
double time_ms(F launch)
{
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    launch(); launch();
    CHECK(hipEventRecord(e0));
    const int reps = 6;
    for (int it = 0; it < reps; ++it) launch();
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    CHECK(hipEventDestroy(e0)); CHECK(hipEventDestroy(e1));
    return ms / reps;
}

template <int CL, int UNROLL>
void try_plain(const double2 *t, const int *idx, long n_idx, double *out, Result &best, const char *dist, double mb)
{
    for (int wg_per_cu : {8, 16, 32}) {
        const int grid = 256 * wg_per_cu;
        const double ms = time_ms([&] { gather<CL, UNROLL><<<grid, 256>>>(t, idx, n_idx, out); });
        printf("{\"kind\": \"variant\", \"table_mb\": %.1f, \"row_bytes\": %d, \"dist\": \"%s\", \"unroll\": %d, \"grid\": %d, "
               "\"ms\": %.4f, \"rows_per_s\": %.4e}\n", mb, CL * 16, dist, UNROLL, grid, ms, n_idx / (ms * 1e-3));
        if (ms < best.ms) best = {ms, UNROLL, grid};
    }
}

template <int CL>
void cell(double mb, const char *dist, long n_idx, const std::vector<int> &h_idx, long rows)
{
    int *d_idx; double2 *d_table; double *d_out;
    CHECK(hipMalloc(&d_idx, n_idx * 4)); CHECK(hipMalloc(&d_table, rows * CL * 16)); CHECK(hipMalloc(&d_out, 8));
    CHECK(hipMemcpy(d_idx, h_idx.data(), n_idx * 4, hipMemcpyHostToDevice));
    CHECK(hipMemset(d_table, 0, rows * CL * 16));
    Result best{1e30, 0, 0};
    try_plain<CL, 2>(d_table, d_idx, n_idx, d_out, best, dist, mb);
    try_plain<CL, 4>(d_table, d_idx, n_idx, d_out, best, dist, mb);
    try_plain<CL, 8>(d_table, d_idx, n_idx, d_out, best, dist, mb);
    if (CL <= 2) try_plain<CL, 16>(d_table, d_idx, n_idx, d_out, best, dist, mb);
    printf("{\"kind\": \"best\", \"table_mb\": %.1f, \"rows\": %ld, \"row_bytes\": %d, \"dist\": \"%s\", \"unroll\": %d, \"grid\": %d, "
           "\"ms\": %.4f, \"rows_per_s\": %.4e, \"useful_gbs\": %.1f, \"line_gbs\": %.1f, \"n_idx\": %ld}\n",
           mb, rows, CL * 16, dist, best.unroll, best.grid, best.ms, n_idx / (best.ms * 1e-3),
           n_idx * (CL * 16.0 + 4) / best.ms / 1e6, n_idx * 64.0 / best.ms / 1e6, n_idx);
    // hot prefix in LDS: only where the distribution has one
    if (!strcmp(dist, "powerlaw")) {
        for (int lds_kb : {64, 128, 156}) {
            const int k_lds = (int)std::min<long>(rows, (long)lds_kb * 1024 / (CL * 16));
            long hits = 0;
            for (long i = 0; i < n_idx; ++i) hits += h_idx[i] < k_lds;
            const size_t shmem = (size_t)k_lds * CL * 16;
            CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&gather_lds<CL, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
            CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&gather_lds<CL, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
            double best_ms = 1e30; int bu = 0, bg = 0;
            for (int per_cu : {1, 2, 4}) {                   // one 1024-thread workgroup owns the CU's LDS; more only queue
                const int grid = 256 * per_cu;
                double ms = time_ms([&] { gather_lds<CL, 4><<<grid, 1024, shmem>>>(d_table, d_idx, n_idx, k_lds, d_out); });
                if (ms < best_ms) { best_ms = ms; bu = 4; bg = grid; }
                ms = time_ms([&] { gather_lds<CL, 8><<<grid, 1024, shmem>>>(d_table, d_idx, n_idx, k_lds, d_out); });
                if (ms < best_ms) { best_ms = ms; bu = 8; bg = grid; }
            }
            printf("{\"kind\": \"lds_prefix\", \"table_mb\": %.1f, \"rows\": %ld, \"row_bytes\": %d, \"dist\": \"%s\", \"lds_kb\": %d, "
                   "\"lds_rows\": %d, \"lds_hit_share\": %.4f, \"unroll\": %d, \"grid\": %d, \"ms\": %.4f, \"rows_per_s\": %.4e, "
                   "\"plain_ms\": %.4f, \"speedup_vs_plain\": %.3f}\n",
                   mb, rows, CL * 16, dist, lds_kb, k_lds, (double)hits / n_idx, bu, bg, best_ms, n_idx / (best_ms * 1e-3), best.ms,
                   best.ms / best_ms);
        }
    }
    fflush(stdout);
    CHECK(hipFree(d_idx)); CHECK(hipFree(d_table)); CHECK(hipFree(d_out));
}

int main(int argc, char **argv)
{
    const long n_idx = argc > 1 ? atol(argv[1]) : 20000000;
    const double sizes_mb[] = {2, 16, 17, 32, 64, 256, 1024};
    for (double mb : sizes_mb) {
        for (int row_bytes : {16, 32, 64}) {
            const long rows = (long)(mb * 1e6 / row_bytes);
            for (const char *dist : {"uniform", "powerlaw"}) {
                std::vector<int> h(n_idx);
                unsigned long long s = 88172645463325252ull;
                for (long i = 0; i < n_idx; ++i) {
                    s ^= s << 13; s ^= s >> 7; s ^= s << 17;
                    const double u = (double)(s >> 11) * (1.0 / 9007199254740992.0);
                    long r = !strcmp(dist, "uniform") ? (long)(u * rows) : (long)(u * u * rows);
                    h[i] = (int)(r < rows ? r : rows - 1);
                }
                if (row_bytes == 16) cell<1>(mb, dist, n_idx, h, rows);
                else if (row_bytes == 32) cell<2>(mb, dist, n_idx, h, rows);
                else cell<4>(mb, dist, n_idx, h, rows);
            }
        }
    }
    return 0;
}
