// Microbenchmark (round 5): what does a small device -> host read-back between dependent kernels cost?
//
//   hipcc -O3 --offload-arch=gfx950 -o readback_latency readback_latency.hip && ./readback_latency
//
// The generation loop and the NMF fit read a few KB back several times per step (distance matrix, Gram matrices,
// residual) before the host can decide what to launch next.  Variants, each timed as "producer kernel -> host has the
// bytes -> consumer kernel launched and finished" over many repetitions:
//   memcpy_sync     hipMemcpyAsync to pinned memory + hipStreamSynchronize                 (what the library does)
//   memcpy_event    hipMemcpyAsync + hipEventRecord, host spins on hipEventQuery
//   mapped_flag     the producer's data are published by a tiny kernel into MAPPED pinned host memory, then a sequence
//                   flag; the host spins on the flag in its own memory (no copy engine, no interrupt, no runtime call)
//   mapped_fused    the producer kernel itself stores to mapped host memory and raises the flag (no extra launch)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <immintrin.h>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void producer(int *out, int words, int seq)
{
    for (int i = threadIdx.x; i < words; i += blockDim.x) out[i] = seq + i;
}
__global__ void consumer(const int *in, int *out, int words)
{
    int s = 0;
    for (int i = threadIdx.x; i < words; i += blockDim.x) s += in[i];
    if (s == 123456789) out[0] = s;
}
__global__ void publish(const int *src, int *host_dst, int words, volatile int *flag, int seq)
{
    for (int i = threadIdx.x; i < words; i += blockDim.x) host_dst[i] = src[i];
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) { *flag = seq; __threadfence_system(); }
}
__global__ void producer_fused(int *out, int *host_dst, int words, volatile int *flag, int seq)
{
    for (int i = threadIdx.x; i < words; i += blockDim.x) { out[i] = seq + i; host_dst[i] = seq + i; }
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) { *flag = seq; __threadfence_system(); }
}

static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main()
{
    const int reps = 2000;
    for (int words : {64, 1369, 25000}) {
        int *d_a, *d_b, *h_pinned, *h_mapped, *d_mapped;
        volatile int *h_flag; int *d_flag;
        CHECK(hipMalloc(&d_a, words * 4)); CHECK(hipMalloc(&d_b, 64));
        CHECK(hipHostMalloc(&h_pinned, words * 4, hipHostMallocDefault));
        CHECK(hipHostMalloc(&h_mapped, words * 4, hipHostMallocMapped));
        CHECK(hipHostGetDevicePointer((void **)&d_mapped, h_mapped, 0));
        CHECK(hipHostMalloc((void **)&h_flag, 64, hipHostMallocMapped));
        CHECK(hipHostGetDevicePointer((void **)&d_flag, (void *)h_flag, 0));
        *h_flag = 0;
        hipStream_t st; CHECK(hipStreamCreate(&st));
        hipEvent_t ev; CHECK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        double t[4];
        for (int variant = 0; variant < 4; ++variant) {
            CHECK(hipStreamSynchronize(st));
            long check = 0;
            const double t0 = now_us();
            for (int r = 1; r <= reps; ++r) {
                if (variant == 0) {
                    producer<<<1, 256, 0, st>>>(d_a, words, r);
                    CHECK(hipMemcpyAsync(h_pinned, d_a, words * 4, hipMemcpyDeviceToHost, st));
                    CHECK(hipStreamSynchronize(st));
                    check += h_pinned[0];
                } else if (variant == 1) {
                    producer<<<1, 256, 0, st>>>(d_a, words, r);
                    CHECK(hipMemcpyAsync(h_pinned, d_a, words * 4, hipMemcpyDeviceToHost, st));
                    CHECK(hipEventRecord(ev, st));
                    while (hipEventQuery(ev) == hipErrorNotReady) _mm_pause();
                    check += h_pinned[0];
                } else if (variant == 2) {
                    producer<<<1, 256, 0, st>>>(d_a, words, r);
                    publish<<<1, 256, 0, st>>>(d_a, d_mapped, words, d_flag, r);
                    while (*h_flag != r) _mm_pause();
                    check += h_mapped[0];
                } else {
                    producer_fused<<<1, 256, 0, st>>>(d_a, d_mapped, words, d_flag, r);
                    while (*h_flag != r) _mm_pause();
                    check += h_mapped[0];
                }
                consumer<<<1, 256, 0, st>>>(d_a, d_b, words);          // the launch that depended on the host's decision
            }
            CHECK(hipStreamSynchronize(st));
            t[variant] = (now_us() - t0) / reps;
            if (check == 42) printf("!");
        }
        printf("{\"bytes\": %d, \"memcpy_sync_us\": %.2f, \"memcpy_event_spin_us\": %.2f, \"mapped_flag_us\": %.2f, \"mapped_fused_us\": %.2f}\n",
               words * 4, t[0], t[1], t[2], t[3]);
        fflush(stdout);
    }
    return 0;
}
