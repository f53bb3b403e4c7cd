// Microbenchmark (round 6): what do the integer atomics of the 1-D k-means++ range update cost?
//
// km_update_kernel (graphrole_amd/csrc/grx_kmeans.hip) walks a range of SORTED positions and subtracts every changed
// distance from the sum of the tile its INDEX lies in: three 64-bit non-returning atomic adds (the limbs of the exact
// sum) to a random one of `tiles` addresses per value, plus the same for the super-tile (tiles / 64 addresses).
// Variants over n values:
//   stream        the pass without atomics (two 8-byte loads, one 4-byte load, one store): the floor
//   tile3         + 3 atomics to the value's tile
//   tile3_super3  + 3 atomics to its super-tile, straight to memory
//   tile3_lds     + super-tile sums collected in LDS per workgroup and flushed once
//   tile1         + 1 atomic to the tile (what a single-limb sum would cost)
// Build: hipcc -O3 --offload-arch=gfx950 atomic_scatter.hip -o atomic_scatter ; run: ./atomic_scatter [n] [tile_shift]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef unsigned long long u64;
constexpr int CHUNK = 2048;

template <int MODE>
__global__ __launch_bounds__(256) void pass_kernel(const double *__restrict__ xs, double *__restrict__ ds,
                                                   const unsigned *__restrict__ perm, long n, int tile_shift,
                                                   u64 *__restrict__ tacc, u64 *__restrict__ sacc, int nsup)
{
    extern __shared__ u64 s_sup[];
    if (MODE == 3) {
        for (int i = threadIdx.x; i < 3 * nsup; i += 256) s_sup[i] = 0;
        __syncthreads();
    }
    const long chunks = (n + CHUNK - 1) / CHUNK;
    for (long ch = blockIdx.x; ch < chunks; ch += gridDim.x) {
        const long p0 = ch * CHUNK;
        double xv[8], dv[8];
        unsigned iv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const long p = p0 + u * 256 + threadIdx.x;
            const long q = p < n ? p : n - 1;
            xv[u] = xs[q]; dv[u] = ds[q]; iv[u] = perm[q];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const long p = p0 + u * 256 + threadIdx.x;
            if (p >= n) continue;
            const double dj = xv[u] * xv[u];
            if (dj < dv[u]) {
                ds[p] = dj;
                const u64 a = (u64)(dv[u] - dj), b = a ^ 0x55, c = a + 3;
                const long tile = iv[u] >> tile_shift;
                if (MODE >= 1) {
                    atomicAdd(&tacc[4 * tile], a);
                    if (MODE != 4) { atomicAdd(&tacc[4 * tile + 1], b); atomicAdd(&tacc[4 * tile + 2], c); }
                }
                if (MODE == 2) {
                    const long s = tile >> 6;
                    atomicAdd(&sacc[4 * s], a); atomicAdd(&sacc[4 * s + 1], b); atomicAdd(&sacc[4 * s + 2], c);
                }
                if (MODE == 3) {
                    const long s = tile >> 6;
                    atomicAdd(&s_sup[3 * s], a); atomicAdd(&s_sup[3 * s + 1], b); atomicAdd(&s_sup[3 * s + 2], c);
                }
            }
        }
    }
    if (MODE == 3) {
        __syncthreads();
        for (int i = threadIdx.x; i < 3 * nsup; i += 256)
            if (s_sup[i]) atomicAdd(&sacc[4 * (i / 3) + i % 3], s_sup[i]);
    }
}

int main(int argc, char **argv)
{
    const long n = argc > 1 ? atol(argv[1]) : 30000000L;
    const int tile_shift = argc > 2 ? atoi(argv[2]) : 10;
    const long tiles = ((n - 1) >> tile_shift) + 1;
    const int nsup = (int)((tiles + 63) / 64);
    std::vector<double> hx(n), hd(n);
    std::vector<unsigned> hp(n);
    for (long i = 0; i < n; ++i) { hx[i] = 1.0 + (double)(i % 97); hd[i] = 1.0e6; hp[i] = (unsigned)i; }
    unsigned long long state = 88172645463325252ull;
    for (long i = n - 1; i > 0; --i) {                          // a random permutation: sorted position -> index
        state ^= state << 13; state ^= state >> 7; state ^= state << 17;
        const long j = (long)(state % (unsigned long long)(i + 1));
        std::swap(hp[i], hp[j]);
    }
    double *xs, *ds;
    unsigned *perm;
    u64 *tacc, *sacc;
    CHECK(hipMalloc(&xs, n * 8)); CHECK(hipMalloc(&ds, n * 8)); CHECK(hipMalloc(&perm, n * 4));
    CHECK(hipMalloc(&tacc, tiles * 32)); CHECK(hipMalloc(&sacc, (size_t)nsup * 32));
    CHECK(hipMemcpy(xs, hx.data(), n * 8, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(perm, hp.data(), n * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const char *names[5] = {"stream", "tile3", "tile3_super3", "tile3_lds", "tile1"};
    printf("{\"n\": %ld, \"tiles\": %ld, \"super_tiles\": %d, \"results\": [\n", n, tiles, nsup);
    bool first = true;
    for (long len : {n, n / 8, n / 64, n / 512}) {
        for (int mode = 0; mode < 5; ++mode) {
            float best = 1e30f;
            for (int rep = 0; rep < 5; ++rep) {
                CHECK(hipMemcpy(ds, hd.data(), len * 8, hipMemcpyHostToDevice));
                CHECK(hipMemset(tacc, 0, tiles * 32)); CHECK(hipMemset(sacc, 0, (size_t)nsup * 32));
                CHECK(hipDeviceSynchronize());
                CHECK(hipEventRecord(e0));
                const int grid = 2048;
                const size_t lds = mode == 3 ? (size_t)nsup * 24 : 0;
                switch (mode) {
                case 0: pass_kernel<0><<<grid, 256, lds>>>(xs, ds, perm, len, tile_shift, tacc, sacc, nsup); break;
                case 1: pass_kernel<1><<<grid, 256, lds>>>(xs, ds, perm, len, tile_shift, tacc, sacc, nsup); break;
                case 2: pass_kernel<2><<<grid, 256, lds>>>(xs, ds, perm, len, tile_shift, tacc, sacc, nsup); break;
                case 3: pass_kernel<3><<<grid, 256, lds>>>(xs, ds, perm, len, tile_shift, tacc, sacc, nsup); break;
                default: pass_kernel<4><<<grid, 256, lds>>>(xs, ds, perm, len, tile_shift, tacc, sacc, nsup); break;
                }
                CHECK(hipEventRecord(e1));
                CHECK(hipEventSynchronize(e1));
                float ms;
                CHECK(hipEventElapsedTime(&ms, e0, e1));
                best = ms < best ? ms : best;
            }
            printf("%s  {\"values\": %ld, \"variant\": \"%s\", \"ms\": %.4f, \"values_per_us\": %.1f}", first ? "" : ",\n", len,
                   names[mode], best, len / (best * 1000.0));
            first = false;
        }
    }
    printf("\n]}\n");
    return 0;
}
