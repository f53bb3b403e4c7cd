// Microbenchmark (round 5): can the time-ordered gather of gather_sliced.hip be had for ORDER-SENSITIVE sums?
//
// The neighbour sums of generations 2 / 3 add doubles in numpy's pairwise order per destination row, so the ADDITIONS
// cannot follow a source-slice order -- but the FETCHES can if the fetched values are parked until the row is summed.
// Here every workgroup owns a tile of E consecutive entries (its rows' neighbour lists), fetches them in slice order
// (the tile's entries sorted by source slice) into LDS slots, then reads the slots back in entry order.  Whether the
// workgroups of the chip stay in the same slice phase without any barrier is what this measures:
//   plain        entries in adjacency order, straight accumulate (the aggregation kernel's pattern)
//   tiles        one launch over all tiles (the hardware refills workgroup slots as tiles finish: phases drift)
//   tile_rounds  one launch per resident set of tiles (every round starts in phase)
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

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

// tile t = entries [t E, (t + 1) E): src[] sorted by slice inside the tile, slot[] = the entry's position in the tile
template <int WORDS, int UNROLL>
__global__ __launch_bounds__(256) void gather_tiles(const double *__restrict__ table, const int *__restrict__ src,
                                                    const unsigned short *__restrict__ slot, int E, long first_tile,
                                                    double *__restrict__ out)
{
    extern __shared__ double lds[];
    const long t = first_tile + blockIdx.x;
    const int *s = src + t * E;
    const unsigned short *p = slot + t * E;
    const int part = threadIdx.x % WORDS;
    const int lane = threadIdx.x / WORDS, lanes = 256 / WORDS;
    for (int k = lane * UNROLL; k + UNROLL <= E; k += lanes * UNROLL) {
        int u[UNROLL];
        int q[UNROLL];
#pragma unroll
        for (int j = 0; j < UNROLL; ++j) { u[j] = s[k + j]; q[j] = p[k + j]; }
        double x[UNROLL];
#pragma unroll
        for (int j = 0; j < UNROLL; ++j) x[j] = table[(long)u[j] * WORDS + part];
#pragma unroll
        for (int j = 0; j < UNROLL; ++j) lds[q[j] * WORDS + part] = x[j];
    }
    __syncthreads();
    double a = 0.0;
    for (int k = threadIdx.x; k < E * WORDS; k += 256) a += lds[k];
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
void cell(long rows, long n_idx, const std::vector<int> &h_idx, int lds_kb, int nslices)
{
    const int E = lds_kb * 1024 / (WORDS * 8);
    const long tiles = n_idx / E;
    std::vector<long> bound(nslices + 1);
    for (int s = 0; s <= nslices; ++s) bound[s] = (long)((double)rows * s * s / ((double)nslices * nslices));
    bound[nslices] = rows;
    std::vector<int> src(tiles * E);
    std::vector<unsigned short> slot(tiles * E);
    std::vector<std::pair<int, int>> tmp(E);
    for (long t = 0; t < tiles; ++t) {
        for (int k = 0; k < E; ++k) {
            const int r = h_idx[t * E + k];
            int s = 0;
            while (s < nslices - 1 && r >= bound[s + 1]) ++s;
            tmp[k] = {s, k};
        }
        std::stable_sort(tmp.begin(), tmp.end(), [](const std::pair<int, int> &a, const std::pair<int, int> &b) { return a.first < b.first; });
        for (int k = 0; k < E; ++k) { src[t * E + k] = h_idx[t * E + tmp[k].second]; slot[t * E + k] = (unsigned short)tmp[k].second; }
    }
    int *d_idx, *d_src; unsigned short *d_slot; double *d_table, *d_out;
    CHECK(hipMalloc(&d_idx, n_idx * 4)); CHECK(hipMalloc(&d_src, tiles * E * 4)); CHECK(hipMalloc(&d_slot, tiles * E * 2));
    CHECK(hipMalloc(&d_table, rows * WORDS * 8)); CHECK(hipMalloc(&d_out, 8));
    CHECK(hipMemcpy(d_idx, h_idx.data(), n_idx * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_src, src.data(), tiles * E * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_slot, slot.data(), tiles * E * 2, hipMemcpyHostToDevice));
    CHECK(hipMemset(d_table, 0, rows * WORDS * 8));
    const size_t shmem = (size_t)E * WORDS * 8;
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&gather_tiles<WORDS, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&gather_tiles<WORDS, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
    double best_plain = 1e30;
    for (int per_cu : {8, 16, 32})
        best_plain = std::min(best_plain, time_ms([&] { gather_plain<WORDS, 8><<<256 * per_cu, 256>>>(d_table, d_idx, n_idx, d_out); }));
    double t4 = time_ms([&] { gather_tiles<WORDS, 4><<<(int)tiles, 256, shmem>>>(d_table, d_src, d_slot, E, 0, d_out); });
    double t8 = time_ms([&] { gather_tiles<WORDS, 8><<<(int)tiles, 256, shmem>>>(d_table, d_src, d_slot, E, 0, d_out); });
    const int per_cu = std::max(1, std::min(8, 160 / lds_kb));
    const long resident = 256L * per_cu;
    double r8 = time_ms([&] {
        for (long f = 0; f < tiles; f += resident)
            gather_tiles<WORDS, 8><<<(int)std::min(resident, tiles - f), 256, shmem>>>(d_table, d_src, d_slot, E, f, d_out);
    });
    printf("{\"row_bytes\": %d, \"table_mb\": %.1f, \"lds_kb\": %d, \"entries_per_tile\": %d, \"tiles\": %ld, \"slices\": %d, "
           "\"plain_ms\": %.4f, \"tiles_ms\": %.4f, \"tile_rounds_ms\": %.4f, \"rounds\": %ld, \"speedup_tiles\": %.3f, \"speedup_rounds\": %.3f}\n",
           WORDS * 8, rows * WORDS * 8 / 1e6, lds_kb, E, tiles, nslices, best_plain, std::min(t4, t8), r8,
           (tiles + resident - 1) / resident, best_plain / std::min(t4, t8), best_plain / r8);
    fflush(stdout);
    CHECK(hipFree(d_idx)); CHECK(hipFree(d_src)); CHECK(hipFree(d_slot)); CHECK(hipFree(d_table)); CHECK(hipFree(d_out));
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
    for (int nslices : {8, 16})
        for (int lds_kb : {16, 32, 64}) {
            cell<1>(rows, n_idx, h, lds_kb, nslices);
            cell<2>(rows, n_idx, h, lds_kb, nslices);
            if (lds_kb >= 32) cell<8>(rows, n_idx, h, lds_kb, nslices);
        }
    return 0;
}
