// grx_common.h -- internal helpers shared by the libgrx.so translation units (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>

#include "grx.h"

void grx_set_error(const char *fmt, ...);

#define GRX_CHECK_HIP(expr)                                                                   \
    do {                                                                                      \
        hipError_t e__ = (expr);                                                              \
        if (e__ != hipSuccess) {                                                              \
            grx_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(e__)); \
            return GRX_ERR_HIP;                                                               \
        }                                                                                     \
    } while (0)

#define GRX_REQUIRE(cond, ...)                                                                \
    do {                                                                                      \
        if (!(cond)) {                                                                        \
            grx_set_error(__VA_ARGS__);                                                       \
            return GRX_ERR_INVALID;                                                           \
        }                                                                                     \
    } while (0)

#define GRX_LAUNCH_CHECK() GRX_CHECK_HIP(hipGetLastError())

static inline hipStream_t grx_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }

// Per-kernel event timing (grx_profile_* in grx.h).  No-ops unless enabled.
enum GrxKernelId {
    GRX_K_ROW_SUMS = 0, GRX_K_EGONET_WAVE, GRX_K_EGONET_BLOCK, GRX_K_PACK_ROWS, GRX_K_AGGREGATE,
    GRX_K_AGGREGATE_HUB, GRX_K_SORT_COUNT, GRX_K_SORT_SCAN, GRX_K_SORT_SCATTER, GRX_K_BIN_THRESHOLD,
    GRX_K_BIN_ASSIGN, GRX_K_CHEBYSHEV, GRX_K_GATHER_COLUMNS, GRX_K_GRAM, GRX_K_PROJECT,
    GRX_K_NNDSVD_APPLY, GRX_K_NMF_W_PASS, GRX_K_REDUCE_PARTIALS, GRX_K_NMF_H_UPDATE,
    GRX_K_NMF_RESIDUAL, GRX_K_ADD_COLUMNS, GRX_K_TRIANGLES, GRX_K_EGONET_FINISH, GRX_K_QUANT, GRX_K_KEY_BITS,
    GRX_K_SEL_MAP, GRX_K_SEL_HIST, GRX_K_SEL_WALK1, GRX_K_SEL_COLLECT, GRX_K_SEL_SEGSORT, GRX_K_SEL_WALK2, GRX_K_ROLE_ROWS, GRX_K_COUNT
};
bool grx_prof_is_on();
void grx_prof_begin(int id, hipStream_t st);
void grx_prof_end(int id, hipStream_t st);
struct GrxProfScope {
    int id; hipStream_t st;
    GrxProfScope(int i, hipStream_t s) : id(i), st(s) { grx_prof_begin(id, st); }
    ~GrxProfScope() { grx_prof_end(id, st); }
};
#define GRX_PROF(id, st) GrxProfScope grx_prof_scope_##id(id, st)

// Column-pointer tables travel as kernel arguments (no host->device copy per call).
constexpr int GRX_MAX_PTRS = 128;
struct GrxPtrTable { const void *p[GRX_MAX_PTRS]; };

constexpr int GRX_HUB_FACTOR = 32;  // grx_aggregate: rows longer than lanes_per_row * 32 are hubs
constexpr int GRX_WAVE = 64;       // CDNA4 wavefront
constexpr int GRX_NUM_CU = 256;    // MI355X

static inline int64_t grx_ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

// Small device -> host read-backs the host's next launch depends on (distance matrix of the pruner, Gram matrices,
// residuals), without the copy engine and without an interrupt: a one-workgroup kernel stores the bytes into the
// caller's PINNED host buffer (hipHostMalloc: device-visible) and then a sequence number into a pinned flag; the host
// spins on the flag in its own memory.  hipMemcpyAsync + hipStreamSynchronize takes 18 us, this 14
// (tools/microbench/readback_latency.hip, profiles/r05_readback_latency.json); a step has nine such points.
// grx_fetch_begin queues one copy (any number before a wait); grx_fetch_wait returns when all queued copies of the
// calling thread are in host memory -- like hipStreamSynchronize it implies that everything queued on the stream before
// them has finished.  More than 32 KB, sizes that are not multiples of 4, GRX_READBACK=memcpy: the copy engine.
int grx_fetch_begin(void *h_dst_pinned, const void *d_src, size_t bytes, hipStream_t st);
int grx_fetch_wait(hipStream_t st);
static inline size_t grx_align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// Fixed-shape butterfly: every lane ends with the same total, the addition tree depends only
// on WIDTH, so results are bitwise reproducible.
template <int WIDTH>
__device__ __forceinline__ double grx_group_sum(double v)
{
#pragma unroll
    for (int off = WIDTH / 2; off > 0; off >>= 1) v += __shfl_xor(v, off, WIDTH);
    return v;
}
