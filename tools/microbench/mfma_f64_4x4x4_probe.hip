// Probe of v_mfma_f64_4x4x4_4b_f64 on gfx950: operand / result lane layout and issue cost.
//   hipcc -O3 --offload-arch=gfx950 -o mfma_probe mfma_f64_4x4x4_probe.hip && ./mfma_probe
// Layout: for every (lane_a, lane_b) a one-hot A (1.0 in lane_a) and one-hot B (1.0 in lane_b) are multiplied;
// the lanes of D that become 1.0 tell which (block, i, k) / (block, k, j) the two input lanes are.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void layout_kernel(int *out)
{
    const int lane = threadIdx.x;
    for (int la = 0; la < 64; ++la)
        for (int lb = 0; lb < 64; ++lb) {
            const double a = lane == la ? 1.0 : 0.0, b = lane == lb ? 1.0 : 0.0;
            const double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
            const unsigned long long m = __ballot(d != 0.0);
            if (lane == 0) { out[(la * 64 + lb) * 2] = (int)(m & 0xFFFFFFFFull); out[(la * 64 + lb) * 2 + 1] = (int)(m >> 32); }
        }
}

template <int WHICH>
__global__ void rate_kernel(double *sink, int iters)
{
    double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
    if (WHICH == 0) {
        double c0 = 0, c1 = 0, c2 = 0, c3 = 0;
        for (int i = 0; i < iters; ++i) {
            c0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c3, 0, 0, 0);
        }
        sink[blockIdx.x * blockDim.x + threadIdx.x] = c0 + c1 + c2 + c3;
    } else {
        typedef double v4d __attribute__((ext_vector_type(4)));
        v4d c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
        for (int i = 0; i < iters; ++i) {
            c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
        }
        sink[blockIdx.x * blockDim.x + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
    }
}

int main()
{
    int *d_out; hipMalloc(&d_out, 64 * 64 * 2 * sizeof(int));
    layout_kernel<<<1, 64>>>(d_out);
    std::vector<int> h(64 * 64 * 2);
    hipMemcpy(h.data(), d_out, h.size() * sizeof(int), hipMemcpyDeviceToHost);
    // for lane_a = 0..63: which lane_b give a product, and where it lands
    for (int la = 0; la < 64; la += 1) {
        printf("A lane %2d:", la);
        for (int lb = 0; lb < 64; ++lb) {
            const unsigned long long m = (unsigned)h[(la * 64 + lb) * 2] | ((unsigned long long)(unsigned)h[(la * 64 + lb) * 2 + 1] << 32);
            if (m) { printf(" B%d->D", lb); for (int l = 0; l < 64; ++l) if ((m >> l) & 1) printf("%d,", l); }
        }
        printf("\n");
    }
    double *sink; hipMalloc(&sink, 256 * 1024 * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    for (int which = 0; which < 2; ++which) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            if (which == 0) rate_kernel<0><<<256 * 4, 256>>>(sink, iters); else rate_kernel<1><<<256 * 4, 256>>>(sink, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            // 4 waves per WG, 4 WGs per CU -> 4 waves per SIMD; MFMAs per SIMD = 4 waves * 4 * iters
            const double per = ms * 1e-3 * 2.4e9 / (4.0 * 4 * iters);
            if (rep) printf("%s: %.3f ms, ~%.1f cycles per MFMA per SIMD (at 2.4 GHz)\n", which ? "16x16x4" : "4x4x4-4b", ms, per);
        }
    }
    return 0;
}
