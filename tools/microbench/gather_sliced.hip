// Microbenchmark (round 5, review item 3b): does slicing the gather SOURCE by XCD pay?
//
//   hipcc -O3 --offload-arch=gfx950 -o gather_sliced gather_sliced.hip && ./gather_sliced > profiles/r05_gather_sliced.json
//
// The aggregation gathers one table row per CSR entry; its rate follows the table's bytes because every XCD's 4 MiB
// L2 sees the whole table (profiles/r04_gather_bw.json: 250 G rows/s L2-resident, 92 G at 16 MB, 62 G at 64 MB).  For
// the order-free integer generation the entries could be bucketed by SOURCE slice at plan time and slice s served
// only by the workgroups with blockIdx % 8 == s (workgroups are dealt round-robin to the 8 XCDs), so that an XCD's L2
// holds 1/8 of the table.  This measures the gather alone, both ways, on the bench graph's shape: 1 M rows, 20 M
// power-law entries (row = floor(N U^2)), rows of 8 / 16 / 64 bytes (generations 1 / 2 / 3 of the BASELINE graph).
// Slices: equal shares of the ENTRIES (boundaries at N (s/8)^2 -- equal row counts would give slice 0 35 % of the work).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// WORDS 8-byte words per row, one lane per word; UNROLL rows in flight per lane group
template <int WORDS, int UNROLL>
__global__ __launch_bounds__(256) void gather_plain(const double *__restrict__ table, const int *__restrict__ idx, long n_idx,
                                                    double *__restrict__ out)
{
    const long lane_global = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int part = threadIdx.x % WORDS;
    const long slot = lane_global / WORDS, nslots = (long)gridDim.x * blockDim.x / WORDS;
    double a = 0.0;
    for (long k = slot * UNROLL; k + UNROLL <= n_idx; k += nslots * UNROLL) {
        int u[UNROLL];
#pragma unroll
        for (int j = 0; j < UNROLL; ++j) u[j] = idx[k + j];
#pragma unroll
        for (int j = 0; j < UNROLL; ++j) a += table[(long)u[j] * WORDS + part];
    }
    if (a == 12345.678) out[0] = a;
}

// workgroup b serves slice b % 8: entries [off[s], off[s + 1]) of the bucketed stream
template <int WORDS, int UNROLL>
__global__ __launch_bounds__(256) void gather_sliced(const double *__restrict__ table, const int *__restrict__ idx,
                                                     const long *__restrict__ off, double *__restrict__ out)
{
    const int s = blockIdx.x & 7;
    const long wg = blockIdx.x >> 3, nwg = gridDim.x >> 3;
    const int part = threadIdx.x % WORDS;
    const long slot = (wg * blockDim.x + threadIdx.x) / WORDS, nslots = nwg * blockDim.x / WORDS;
    const long b = off[s], e = off[s + 1];
    double a = 0.0;
    for (long k = b + slot * UNROLL; k + UNROLL <= e; k += nslots * UNROLL) {
        int u[UNROLL];
#pragma unroll
        for (int j = 0; j < UNROLL; ++j) u[j] = idx[k + j];
#pragma unroll
        for (int j = 0; j < UNROLL; ++j) a += table[(long)u[j] * WORDS + part];
    }
    if (a == 12345.678) out[0] = a;
}

template <typename F>
double time_ms(F launch)
{
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    launch(); launch();
    CHECK(hipEventRecord(e0));
    const int reps = 8;
    for (int it = 0; it < reps; ++it) launch();
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    return ms / reps;
}

template <int WORDS>
void cell(long rows, long n_idx, const std::vector<int> &h_idx, bool equal_rows)
{
    // bucket the stream by slice (stable: inside a slice the stream keeps its order)
    long bound[9];
    for (int s = 0; s <= 8; ++s) bound[s] = equal_rows ? rows * s / 8 : (long)((double)rows * s * s / 64.0);
    bound[8] = rows;
    std::vector<long> off(9, 0);
    std::vector<int> slice_of(n_idx);
    for (long i = 0; i < n_idx; ++i) {
        int s = 0;
        while (s < 7 && h_idx[i] >= bound[s + 1]) ++s;
        slice_of[i] = s;
        ++off[s + 1];
    }
    for (int s = 0; s < 8; ++s) off[s + 1] += off[s];
    std::vector<int> bucketed(n_idx);
    std::vector<long> cur(off.begin(), off.end() - 1);
    for (long i = 0; i < n_idx; ++i) bucketed[cur[slice_of[i]]++] = h_idx[i];
    int *d_idx, *d_bucketed; long *d_off; double *d_table, *d_out;
    CHECK(hipMalloc(&d_idx, n_idx * 4)); CHECK(hipMalloc(&d_bucketed, n_idx * 4)); CHECK(hipMalloc(&d_off, 9 * 8));
    CHECK(hipMalloc(&d_table, rows * WORDS * 8)); CHECK(hipMalloc(&d_out, 8));
    CHECK(hipMemcpy(d_idx, h_idx.data(), n_idx * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_bucketed, bucketed.data(), n_idx * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_off, off.data(), 9 * 8, hipMemcpyHostToDevice));
    CHECK(hipMemset(d_table, 0, rows * WORDS * 8));
    double best_plain = 1e30, best_sliced = 1e30, best_bucketed_plain = 1e30;
    for (int per_cu : {8, 16, 32}) {
        const int grid = 256 * per_cu;
        best_plain = std::min(best_plain, time_ms([&] { gather_plain<WORDS, 4><<<grid, 256>>>(d_table, d_idx, n_idx, d_out); }));
        best_plain = std::min(best_plain, time_ms([&] { gather_plain<WORDS, 8><<<grid, 256>>>(d_table, d_idx, n_idx, d_out); }));
        // the bucketed stream read by ALL workgroups in order: time locality without the XCD affinity
        best_bucketed_plain = std::min(best_bucketed_plain, time_ms([&] { gather_plain<WORDS, 8><<<grid, 256>>>(d_table, d_bucketed, n_idx, d_out); }));
        best_sliced = std::min(best_sliced, time_ms([&] { gather_sliced<WORDS, 4><<<grid, 256>>>(d_table, d_bucketed, d_off, d_out); }));
        best_sliced = std::min(best_sliced, time_ms([&] { gather_sliced<WORDS, 8><<<grid, 256>>>(d_table, d_bucketed, d_off, d_out); }));
    }
    printf("{\"rows\": %ld, \"row_bytes\": %d, \"table_mb\": %.1f, \"entries\": %ld, \"slices\": \"%s\", \"largest_slice_mb\": %.2f, "
           "\"plain_ms\": %.4f, \"plain_rows_per_s\": %.4e, \"bucketed_stream_all_workgroups_ms\": %.4f, \"sliced_ms\": %.4f, "
           "\"sliced_rows_per_s\": %.4e, \"speedup\": %.3f}\n",
           rows, WORDS * 8, rows * WORDS * 8 / 1e6, n_idx, equal_rows ? "equal rows" : "equal entry shares",
           (bound[8] - bound[7]) * WORDS * 8 / 1e6, best_plain, n_idx / (best_plain * 1e-3), best_bucketed_plain, best_sliced,
           n_idx / (best_sliced * 1e-3), best_plain / best_sliced);
    fflush(stdout);
    CHECK(hipFree(d_idx)); CHECK(hipFree(d_bucketed)); CHECK(hipFree(d_off)); CHECK(hipFree(d_table)); CHECK(hipFree(d_out));
}

int main()
{
    const long rows = 1000000, n_idx = 20000000;
    std::vector<int> h(n_idx);
    unsigned long long s = 88172645463325252ull;
    for (long i = 0; i < n_idx; ++i) {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        const double u = (double)(s >> 11) * (1.0 / 9007199254740992.0);
        const long r = (long)(u * u * rows);
        h[i] = (int)(r < rows ? r : rows - 1);
    }
    for (bool equal_rows : {false, true}) {
        cell<1>(rows, n_idx, h, equal_rows);
        cell<2>(rows, n_idx, h, equal_rows);
        cell<8>(rows, n_idx, h, equal_rows);
    }
    return 0;
}
