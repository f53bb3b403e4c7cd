// grx_prune.hip -- vertical logarithmic binning and pairwise Chebyshev distance.
//
// graphrole/features/prune.py:13-56 needs, per column, the sorted values (np.unique + cumsum),
// a short walk that picks the bin thresholds, and a relabelling pass.  On the GPU:
//   1. batched LSD radix sort of the fp64 columns (8 passes x 8 bits, keys only)
//        tile_count_kernel -> scan_kernel -> scatter_kernel        (HBM bound, wave ballots)
//   2. bin_threshold_kernel : one wavefront per column walks the sorted column; the end of a
//        tie run is found with a 64-ary ballot search                (latency bound, tiny)
//   3. bin_assign_kernel    : each value -> lower_bound over <=128 thresholds held in LDS
//   4. chebyshev_kernel     : max_i |bin_p[i]-bin_q[i]| for column pairs, LDS-tiled rows,
//        integer atomicMax into the F x F matrix (prune.py:108)
// All integer work: results are bit-exact functions of the input columns.
#include "grx_common.h"

#include <cstdlib>

namespace {

constexpr int SORT_THREADS = 256;
constexpr int SORT_ITEMS = 16;
constexpr int SORT_TILE = SORT_THREADS * SORT_ITEMS;     // 4096 keys per workgroup
constexpr int RADIX = 256;

__device__ __forceinline__ uint64_t f64_to_key(double x)
{
    const uint64_t b = (uint64_t)__double_as_longlong(x);
    return b ^ ((b >> 63) ? 0xFFFFFFFFFFFFFFFFull : 0x8000000000000000ull);
}

__device__ __forceinline__ double key_to_f64(uint64_t k)
{
    const uint64_t b = k ^ ((k >> 63) ? 0x8000000000000000ull : 0xFFFFFFFFFFFFFFFFull);
    return __longlong_as_double((long long)b);
}

// Pass skipping (grx_vertical_log_bin only): a byte position that is constant over a whole column
// makes its LSD pass the identity permutation.  Pass 0's counting kernel also accumulates the OR of
// the keys and of their complements; scan pass 0 turns them into flags[pass][col].  A skipped pass
// launches nothing useful (every workgroup returns at once) and does not flip the ping-pong, so scan
// pass 0 also records the buffer each column is in before every pass: 0 = the fp64 input, 1 = bufA,
// 2 = bufB (one byte for a workgroup to read before it can issue its loads).  flags == nullptr: plain
// sort, explicit src / dst.
struct SkipCtl {
    uint8_t *flags;            // [9][ncols]: PASS_SKIPPED, or the buffer the column is in before pass p (row 8: at the end)
    uint64_t *bits;            // [ncols][ntiles][2]: per tile, OR of the keys and OR of their complements
    const double *cols;
    int64_t cols_ld;
    uint64_t *buf_a, *buf_b;   // column stride n
    int ncols;
};

constexpr uint8_t PASS_SKIPPED = 0xFF;

// Load the ITEMS keys of this thread.  Wave w of the tile owns the contiguous slice
// [w*64*ITEMS, (w+1)*64*ITEMS); item i of lane l is element i*64 + l of that slice, so
// (wave, item, lane) order == memory order (needed for LSD stability) and loads coalesce.
// The loads are issued back to back from clamped addresses and converted / masked afterwards, behind a scheduling
// barrier: a load inside `if (idx < n)` -- or a conversion next to it -- makes hipcc wait for every load before it
// issues the next one (s_waitcnt vmcnt(0) sixteen times per thread; found in the ISA).
template <bool from_f64>
__device__ __forceinline__ void load_keys(const void *__restrict__ src, int64_t n, int64_t tile_base,
                                          uint64_t (&keys)[SORT_ITEMS], uint32_t &valid_mask)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t base = tile_base + (int64_t)wave * 64 * SORT_ITEMS + lane;
    const int64_t last = n > 0 ? n - 1 : 0;
    uint64_t raw[SORT_ITEMS];
#pragma unroll
    for (int i = 0; i < SORT_ITEMS; ++i) {
        const int64_t idx = base + (int64_t)i * 64;
        raw[i] = reinterpret_cast<const uint64_t *>(src)[idx < n ? idx : last];
    }
    __builtin_amdgcn_sched_barrier(0);
    valid_mask = 0;
#pragma unroll
    for (int i = 0; i < SORT_ITEMS; ++i) {
        const bool ok = base + (int64_t)i * 64 < n;
        valid_mask |= ok ? 1u << i : 0u;
        const uint64_t k = from_f64 ? f64_to_key(__longlong_as_double((long long)raw[i])) : raw[i];
        keys[i] = ok ? k : 0xFFFFFFFFFFFFFFFFull;
    }
}

// hist layout per column: [RADIX][ntiles] (digit-major) so one flat exclusive scan yields the
// global output offset of (digit, tile).
template <bool FROM_F64>
__global__ __launch_bounds__(SORT_THREADS) void tile_count_kernel(
    const void *__restrict__ src, int64_t src_ld, int64_t n, int shift, int ntiles,
    uint32_t *__restrict__ hist, SkipCtl ctl)
{
    __shared__ uint32_t cnt[RADIX];
    const int col = blockIdx.y, tile = blockIdx.x;
    const int pass = shift >> 3;
    const char *csrc = reinterpret_cast<const char *>(src) + (size_t)col * src_ld * 8;
    bool from_f64 = FROM_F64;
    if (ctl.flags && pass > 0) {
        const int cur = ctl.flags[pass * ctl.ncols + col];
        if (cur == PASS_SKIPPED) return;                        // constant byte: nothing to count
        from_f64 = cur == 0;
        csrc = cur == 0 ? reinterpret_cast<const char *>(ctl.cols + (size_t)col * ctl.cols_ld)
                        : reinterpret_cast<const char *>((cur == 1 ? ctl.buf_a : ctl.buf_b) + (size_t)col * n);
    }
    cnt[threadIdx.x] = 0;
    __syncthreads();
    uint64_t keys[SORT_ITEMS];
    uint32_t vm;
    // one uniform branch around the sixteen loads (fp64 input after pass 0 only when every lower byte was constant)
    if (FROM_F64 || from_f64) load_keys<true>(csrc, n, (int64_t)tile * SORT_TILE, keys, vm);
    else load_keys<false>(csrc, n, (int64_t)tile * SORT_TILE, keys, vm);
    const int lane = threadIdx.x & 63;
    if (ctl.flags && pass == 0) {
        // which bit positions vary over the column: OR of the keys and OR of their complements
        uint64_t o = 0, z = 0;
#pragma unroll
        for (int i = 0; i < SORT_ITEMS; ++i)
            if ((vm >> i) & 1u) { o |= keys[i]; z |= ~keys[i]; }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            o |= __shfl_xor(o, off, 64);
            z |= __shfl_xor(z, off, 64);
        }
        // per-tile pair, OR-ed over the tiles by scan pass 0 (no atomics: every wave of the first
        // resident batch would hit the same address at once)
        __shared__ uint64_t wbits[4][2];
        if (lane == 0) { wbits[threadIdx.x >> 6][0] = o; wbits[threadIdx.x >> 6][1] = z; }
        __syncthreads();
        if (threadIdx.x < 2) {
            const int k = threadIdx.x;
            ctl.bits[((size_t)col * ntiles + tile) * 2 + k] = wbits[0][k] | wbits[1][k] | wbits[2][k] | wbits[3][k];
        }
    }
#pragma unroll
    for (int i = 0; i < SORT_ITEMS; ++i) {
        // same-address LDS atomics serialise lane by lane; columns of small integers (degrees, their
        // sums) have whole wavefronts agreeing on most digits: count those with one atomic
        const bool valid = (vm >> i) & 1u;
        const uint32_t d = (uint32_t)(keys[i] >> shift) & 0xFF;
        const uint64_t active = __ballot(valid);
        if (active == 0) continue;                              // uniform over the wave
        const int leader = __ffsll((long long)active) - 1;
        const uint32_t d0 = __shfl(d, leader, 64);
        if (__ballot(valid && d != d0) == 0) {
            if (lane == leader) atomicAdd(&cnt[d0], (uint32_t)__popcll(active));
        } else if (valid) {
            atomicAdd(&cnt[d], 1u);
        }
    }
    __syncthreads();
    hist[((size_t)col * RADIX + threadIdx.x) * ntiles + tile] = cnt[threadIdx.x];
}

// scan_rows_kernel: per (column, digit) exclusive scan of the digit-major counter table
// hist[RADIX][ntiles] across tiles, digit total -> tot.  scatter_kernel scans the 256 digit totals
// itself (digit base) and adds it to the per-tile offset.
__global__ __launch_bounds__(64) void scan_rows_kernel(uint32_t *__restrict__ hist, int ntiles,
                                                       uint32_t *__restrict__ tot, SkipCtl ctl, int pass)
{
    const int d = blockIdx.x, col = blockIdx.y, lane = threadIdx.x;
    if (ctl.flags) {
        if (pass > 0 && ctl.flags[pass * ctl.ncols + col] == PASS_SKIPPED) return;
        if (pass == 0 && d == 0) {
            // a bit varies iff it is set in some key and clear in another
            uint64_t o = 0, z = 0;
            for (int t = lane; t < ntiles; t += 64) {
                o |= ctl.bits[((size_t)col * ntiles + t) * 2];
                z |= ctl.bits[((size_t)col * ntiles + t) * 2 + 1];
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                o |= __shfl_xor(o, off, 64);
                z |= __shfl_xor(z, off, 64);
            }
            const uint64_t varying = o & z;
            if (lane == 0) {
                int cur = 0;
                for (int p = 0; p < 8; ++p) {
                    const bool skip = ((varying >> (8 * p)) & 0xFF) == 0;
                    ctl.flags[p * ctl.ncols + col] = skip ? PASS_SKIPPED : (uint8_t)cur;
                    if (!skip) cur = (cur == 1) ? 2 : 1;
                }
                ctl.flags[8 * ctl.ncols + col] = (uint8_t)cur;
            }
        }
    }
    uint32_t *row = hist + ((size_t)col * RADIX + d) * ntiles;
    uint32_t carry = 0;
    for (int t0 = 0; t0 < ntiles; t0 += 64 * 4) {
        uint32_t x[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {                       // issue the loads together
            const int t = t0 + j * 64 + lane;
            x[j] = (t < ntiles) ? row[t] : 0u;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int t = t0 + j * 64 + lane;
            uint32_t inc = x[j];
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const uint32_t y = __shfl_up(inc, off, 64);
                if (lane >= off) inc += y;
            }
            if (t < ntiles) row[t] = carry + inc - x[j];
            carry += __shfl(inc, 63, 64);
        }
    }
    if (lane == 0) tot[(size_t)col * RADIX + d] = carry;
}

template <bool FROM_F64, bool TO_F64>
__global__ __launch_bounds__(SORT_THREADS) void scatter_kernel(
    const void *__restrict__ src, int64_t src_ld, void *__restrict__ dst, int64_t dst_ld, int64_t n,
    int shift, int ntiles, const uint32_t *__restrict__ offsets, const uint32_t *__restrict__ digit_tot,
    SkipCtl ctl)
{
    __shared__ uint32_t cnt[4][RADIX];
    __shared__ uint32_t gdelta[RADIX];
    __shared__ uint32_t wsum[8];
    __shared__ uint64_t stage[SORT_TILE];
    const int col = blockIdx.y, tile = blockIdx.x;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int cur = ctl.flags ? ctl.flags[(shift >> 3) * ctl.ncols + col] : 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) cnt[w][threadIdx.x] = 0;
    __syncthreads();
    const char *csrc = reinterpret_cast<const char *>(src) + (size_t)col * src_ld * 8;
    char *cdst = reinterpret_cast<char *>(dst) + (size_t)col * dst_ld * 8;
    bool from_f64 = FROM_F64;
    if (ctl.flags) {
        if (cur == PASS_SKIPPED) return;                        // identity permutation: the keys stay put
        from_f64 = cur == 0;
        csrc = cur == 0 ? reinterpret_cast<const char *>(ctl.cols + (size_t)col * ctl.cols_ld)
                        : reinterpret_cast<const char *>((cur == 1 ? ctl.buf_a : ctl.buf_b) + (size_t)col * n);
        cdst = reinterpret_cast<char *>((cur == 1 ? ctl.buf_b : ctl.buf_a) + (size_t)col * n);
    }
    uint64_t keys[SORT_ITEMS];
    uint32_t vm;
    // one uniform branch around the sixteen loads (fp64 input after pass 0 only when every lower byte was constant)
    if (FROM_F64 || from_f64) load_keys<true>(csrc, n, (int64_t)tile * SORT_TILE, keys, vm);
    else load_keys<false>(csrc, n, (int64_t)tile * SORT_TILE, keys, vm);
    const uint64_t lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    uint32_t rank[SORT_ITEMS];
#pragma unroll
    for (int i = 0; i < SORT_ITEMS; ++i) {
        const bool valid = (vm >> i) & 1u;
        const uint32_t d = (uint32_t)(keys[i] >> shift) & 0xFF;
        uint64_t peers = __ballot(valid);
        // lanes holding the same digit: eight ballots -- unless the whole wavefront agrees (every
        // constant digit position of an integer-valued column), which one shuffle + ballot detects
        const uint32_t d0 = __shfl(d, peers ? __ffsll((long long)peers) - 1 : 0, 64);
        if (__ballot(valid && d != d0) != 0) {
#pragma unroll
            for (int bit = 0; bit < 8; ++bit) {
                const bool set = (d >> bit) & 1u;
                const uint64_t m = __ballot(set);
                peers &= set ? m : ~m;
            }
        }
        uint32_t r = 0;
        if (valid) {
            const uint32_t before = cnt[wave][d];
            const uint32_t in_group = (uint32_t)__popcll(peers & lt_mask);
            r = before + in_group;
            __builtin_amdgcn_wave_barrier();
            if (in_group == 0) cnt[wave][d] = before + (uint32_t)__popcll(peers);
        }
        __builtin_amdgcn_wave_barrier();
        rank[i] = r;
    }
    __syncthreads();
    // The keys are first placed in digit order in LDS (tile-local position = exclusive digit prefix
    // + wave offset + rank), then written out by consecutive lanes: a digit's run of keys (16 on
    // average) becomes one contiguous global store instead of 8-byte stores to 64 places.
    {
        const int d = threadIdx.x;
        const uint32_t c0 = cnt[0][d], c1 = cnt[1][d], c2 = cnt[2][d], c3 = cnt[3][d];
        const uint32_t total = c0 + c1 + c2 + c3;
        // two exclusive prefixes over the 256 digits: the tile-local one (digit totals of this tile) and
        // the global digit base (digit totals of the whole column, from scan_rows_kernel)
        const uint32_t gtot = digit_tot[(size_t)col * RADIX + d];
        uint32_t inc = total, ginc = gtot;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t y = __shfl_up(inc, off, 64), gy = __shfl_up(ginc, off, 64);
            if (lane >= off) { inc += y; ginc += gy; }
        }
        if (lane == 63) { wsum[wave] = inc; wsum[4 + wave] = ginc; }
        __syncthreads();
        uint32_t lp = inc - total, gbase = ginc - gtot;
        for (int w = 0; w < wave; ++w) { lp += wsum[w]; gbase += wsum[4 + w]; }
        const uint32_t g = offsets[((size_t)col * RADIX + d) * ntiles + tile] + gbase;
        gdelta[d] = g - lp;                                     // global position = gdelta[digit] + local position
        cnt[0][d] = lp;
        cnt[1][d] = lp + c0;
        cnt[2][d] = lp + c0 + c1;
        cnt[3][d] = lp + c0 + c1 + c2;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < SORT_ITEMS; ++i) {
        if ((vm >> i) & 1u) {
            const uint32_t d = (uint32_t)(keys[i] >> shift) & 0xFF;
            stage[cnt[wave][d] + rank[i]] = keys[i];
        }
    }
    __syncthreads();
    const int64_t left = n - (int64_t)tile * SORT_TILE;
    const int nv = (int)(left < SORT_TILE ? left : SORT_TILE);
    for (int j = threadIdx.x; j < nv; j += SORT_THREADS) {
        const uint64_t key = stage[j];
        const uint32_t pos = gdelta[(uint32_t)(key >> shift) & 0xFF] + (uint32_t)j;
        if (TO_F64) reinterpret_cast<double *>(cdst)[pos] = key_to_f64(key);
        else reinterpret_cast<uint64_t *>(cdst)[pos] = key;
    }
}

// scatter_kernel with a 32-bit payload travelling with every key (grx_kmeans.hip: the index of the value, so that the
// sort also yields the permutation).  Single column, no pass skipping.  pay_src == nullptr: the payload is the
// position (pass 0).  Stable like scatter_kernel: equal keys keep their index order.
template <bool FROM_F64, bool TO_F64>
__global__ __launch_bounds__(SORT_THREADS) void scatter_pairs_kernel(
    const void *__restrict__ src, void *__restrict__ dst, const uint32_t *__restrict__ pay_src,
    uint32_t *__restrict__ pay_dst, int64_t n, int shift, int ntiles, const uint32_t *__restrict__ offsets,
    const uint32_t *__restrict__ digit_tot)
{
    __shared__ uint32_t cnt[4][RADIX];
    __shared__ uint32_t gdelta[RADIX];
    __shared__ uint32_t wsum[8];
    __shared__ uint64_t stage[SORT_TILE];
    __shared__ uint32_t stage_pay[SORT_TILE];
    const int tile = blockIdx.x;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
    for (int w = 0; w < 4; ++w) cnt[w][threadIdx.x] = 0;
    __syncthreads();
    uint64_t keys[SORT_ITEMS];
    uint32_t pay[SORT_ITEMS];
    uint32_t vm;
    load_keys<FROM_F64>(src, n, (int64_t)tile * SORT_TILE, keys, vm);
    {
        const int64_t base = (int64_t)tile * SORT_TILE + (int64_t)wave * 64 * SORT_ITEMS + lane;
        const int64_t last = n > 0 ? n - 1 : 0;
#pragma unroll
        for (int i = 0; i < SORT_ITEMS; ++i) {
            const int64_t idx = base + (int64_t)i * 64;
            pay[i] = pay_src ? pay_src[idx < n ? idx : last] : (uint32_t)idx;
        }
    }
    const uint64_t lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    uint32_t rank[SORT_ITEMS];
#pragma unroll
    for (int i = 0; i < SORT_ITEMS; ++i) {
        const bool valid = (vm >> i) & 1u;
        const uint32_t d = (uint32_t)(keys[i] >> shift) & 0xFF;
        uint64_t peers = __ballot(valid);
        const uint32_t d0 = __shfl(d, peers ? __ffsll((long long)peers) - 1 : 0, 64);
        if (__ballot(valid && d != d0) != 0) {
#pragma unroll
            for (int bit = 0; bit < 8; ++bit) {
                const bool set = (d >> bit) & 1u;
                const uint64_t m = __ballot(set);
                peers &= set ? m : ~m;
            }
        }
        uint32_t r = 0;
        if (valid) {
            const uint32_t before = cnt[wave][d];
            const uint32_t in_group = (uint32_t)__popcll(peers & lt_mask);
            r = before + in_group;
            __builtin_amdgcn_wave_barrier();
            if (in_group == 0) cnt[wave][d] = before + (uint32_t)__popcll(peers);
        }
        __builtin_amdgcn_wave_barrier();
        rank[i] = r;
    }
    __syncthreads();
    {
        const int d = threadIdx.x;
        const uint32_t c0 = cnt[0][d], c1 = cnt[1][d], c2 = cnt[2][d], c3 = cnt[3][d];
        const uint32_t total = c0 + c1 + c2 + c3;
        const uint32_t gtot = digit_tot[d];
        uint32_t inc = total, ginc = gtot;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t y = __shfl_up(inc, off, 64), gy = __shfl_up(ginc, off, 64);
            if (lane >= off) { inc += y; ginc += gy; }
        }
        if (lane == 63) { wsum[wave] = inc; wsum[4 + wave] = ginc; }
        __syncthreads();
        uint32_t lp = inc - total, gbase = ginc - gtot;
        for (int w = 0; w < wave; ++w) { lp += wsum[w]; gbase += wsum[4 + w]; }
        const uint32_t g = offsets[(size_t)d * ntiles + tile] + gbase;
        gdelta[d] = g - lp;
        cnt[0][d] = lp;
        cnt[1][d] = lp + c0;
        cnt[2][d] = lp + c0 + c1;
        cnt[3][d] = lp + c0 + c1 + c2;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < SORT_ITEMS; ++i) {
        if ((vm >> i) & 1u) {
            const uint32_t d = (uint32_t)(keys[i] >> shift) & 0xFF;
            const uint32_t at = cnt[wave][d] + rank[i];
            stage[at] = keys[i];
            stage_pay[at] = pay[i];
        }
    }
    __syncthreads();
    const int64_t left = n - (int64_t)tile * SORT_TILE;
    const int nv = (int)(left < SORT_TILE ? left : SORT_TILE);
    for (int j = threadIdx.x; j < nv; j += SORT_THREADS) {
        const uint64_t key = stage[j];
        const uint32_t pos = gdelta[(uint32_t)(key >> shift) & 0xFF] + (uint32_t)j;
        if (TO_F64) reinterpret_cast<double *>(dst)[pos] = key_to_f64(key);
        else reinterpret_cast<uint64_t *>(dst)[pos] = key;
        pay_dst[pos] = stage_pay[j];
    }
}

// One wavefront per column.  prune.py:33-54: repeat { size = max(int(frac*unbinned),1);
// hi = value at sorted position done+size-1; extend to the end of hi's tie run }.
__global__ __launch_bounds__(64) void bin_threshold_kernel(const double *__restrict__ sorted,
                                                           int64_t ld, int64_t n, double frac,
                                                           double *__restrict__ thr,
                                                           int32_t *__restrict__ nbins, SkipCtl ctl)
{
    // with pass skipping the sorted column is wherever its last executed pass left it, as keys
    // (buffer 0 = the input itself: every byte constant, i.e. all values equal)
    const int where = ctl.flags ? ctl.flags[8 * ctl.ncols + blockIdx.x] : 0;
    const double *sd = ctl.flags ? ctl.cols + (size_t)blockIdx.x * ctl.cols_ld : sorted + (size_t)blockIdx.x * ld;
    const uint64_t *sk = (where == 1 ? ctl.buf_a : ctl.buf_b) + (size_t)blockIdx.x * n;
    struct { const double *d; const uint64_t *k; bool keys;
             __device__ double operator[](int64_t i) const { return keys ? key_to_f64(k[i]) : d[i]; } }
        s{sd, sk, where != 0};
    double *t = thr + (size_t)blockIdx.x * GRX_MAX_BINS;
    const int lane = threadIdx.x;
    int64_t done = 0;
    int nb = 0;
    while (done < n && nb < GRX_MAX_BINS) {
        int64_t size = (int64_t)(frac * (double)(n - done));
        if (size < 1) size = 1;
        const int64_t pos = done + size - 1;
        const double hi = s[pos];
        int64_t L = pos + 1, R = n;
        while (L < R) {
            const int64_t len = R - L;
            const int64_t step = (len + 63) >> 6;
            const int64_t idx = L + (int64_t)lane * step;
            const bool eq = (idx < R) && (s[idx] == hi);
            const int c = __popcll(__ballot(eq));
            if (c == 0) {
                R = L;
            } else {
                const int64_t nL = L + (int64_t)(c - 1) * step + 1;
                const int64_t cap = L + (int64_t)c * step;
                R = (cap < R) ? cap : R;
                L = nL;
            }
        }
        if (lane == 0) t[nb] = hi;
        ++nb;
        done = L;
    }
    if (lane == 0) nbins[blockIdx.x] = nb;
}

// =======================================================================================
// Binning-specific sort ("window" sort).  The threshold walk of prune.py:33-54 needs ~20 order
// statistics and the ends of their tie runs, not a fully sorted column, and most columns vary in few
// of their 64 key bits.  One light pass finds which key bytes vary over each column
// (key_bits_kernel -> bin_plan_kernel); with lo / hi the lowest / highest varying byte:
//   NARROW (hi - lo < 4)  the column is sorted as 32-bit keys (key >> 8 lo): exact, 4-byte elements,
//                          one LSD round per varying byte (integer-valued columns below 2^20 take 2-4
//                          rounds of 12 bytes per key instead of 5 of 24);
//   WIDE   (otherwise)     64-bit keys sorted by their top four window bytes only (bytes hi-3 .. hi):
//                          at most four rounds; keys that agree in those bytes form short runs whose
//                          order is resolved lazily, only where a bin threshold lands
//                          (bin_threshold2_kernel: shuffle ranking for runs up to 64 keys, workgroup
//                          radix selection for longer ones).
// Exactly four rounds are launched whatever the columns need (a round a column does not use returns at
// once): 13 launches per call instead of 24, and about half the key bytes moved.
// =======================================================================================
struct BinPlanCol {
    uint64_t const_bits;      // NARROW: the (constant) key bits outside the 32 sorted ones
    uint8_t narrow;
    uint8_t npass;            // rounds this column takes part in (0: every key equal)
    uint8_t lo_shift;         // NARROW: bit offset of the window; WIDE: bit offset of the sorted part
    uint8_t pad0;
    uint8_t dshift[4];        // per round: shift of the stored key that exposes the round's digit
};

// Columns that hold int64 BITS instead of doubles (the reference's integer columns once 'prod' made them overflow
// 2^53, csrc/grx_aggx.hip): their order key is the two's-complement value with the sign bit flipped.  One bit per
// column of the call, passed by value.
struct ColFlags { uint64_t w[8]; };
constexpr int MAX_TYPED_COLS = 512;
__device__ __forceinline__ bool col_is_i64(const ColFlags &f, int col) { return col < MAX_TYPED_COLS && ((f.w[col >> 6] >> (col & 63)) & 1ull); }
__device__ __forceinline__ uint64_t i64_bits_to_key(double raw) { return (uint64_t)__double_as_longlong(raw) ^ 0x8000000000000000ull; }
// order key of a stored value of either kind (-0.0 and 0.0 are one value for doubles, like np.unique)
__device__ __forceinline__ uint64_t value_key(double raw, bool is_i64) { return is_i64 ? i64_bits_to_key(raw) : f64_to_key(raw + 0.0); }

constexpr int BITS_TILE = 256 * 32;
// INTEGER columns (every value a non-negative integer below 2^32: degrees, ego-net counts, their neighbour sums) may
// be sorted by the integer itself instead of the fp64 bit pattern: a degree-like column below 2^16 varies in two
// bytes of the integer but in four of the fp64 key (exponent + leading mantissa bits).  Marked by lo_shift.
constexpr int LO_SHIFT_INT = 255;

__global__ __launch_bounds__(256) void key_bits_kernel(const double *__restrict__ cols, int64_t ld, int64_t n,
                                                       int ntiles, uint64_t *__restrict__ bits, ColFlags flags)
{
    const int col = blockIdx.y, tile = blockIdx.x;
    const bool i64 = col_is_i64(flags, col);
    const double *x = cols + (size_t)col * ld;
    const int64_t base = (int64_t)tile * BITS_TILE;
    uint64_t o = 0, z = 0;
    uint32_t oi = 0, zi = 0, notint = 0;
    const int64_t last = n > 0 ? n - 1 : 0;
    for (int i0 = 0; i0 < 32; i0 += 8) {
        // eight loads in flight (clamped addresses), then the conversions (see load_keys)
        double raw[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int64_t idx = base + (int64_t)(i0 + j) * 256 + threadIdx.x;
            raw[j] = x[idx < n ? idx : last];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int64_t idx = base + (int64_t)(i0 + j) * 256 + threadIdx.x;
            const double v = raw[j] + 0.0;                    // + 0.0: -0.0 and 0.0 are one value (np.unique)
            const uint64_t k = i64 ? i64_bits_to_key(raw[j]) : f64_to_key(v);
            o |= idx < n ? k : 0;
            z |= idx < n ? ~k : 0;
            const bool in_range = !i64 && v >= 0.0 && v < 4294967296.0;  // false for NaN; int64 columns keep their own key
            const uint32_t iv = in_range ? (uint32_t)v : 0u;
            const bool isint = in_range && (double)iv == v;
            oi |= idx < n ? iv : 0u;
            zi |= idx < n ? ~iv : 0u;
            notint |= (idx < n && !isint) ? 1u : 0u;
        }
    }
    uint64_t w2 = (uint64_t)oi | ((uint64_t)notint << 32), w3 = zi;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        o |= __shfl_xor(o, off, 64);
        z |= __shfl_xor(z, off, 64);
        w2 |= __shfl_xor(w2, off, 64);
        w3 |= __shfl_xor(w3, off, 64);
    }
    __shared__ uint64_t wb[4][4];
    if ((threadIdx.x & 63) == 0) {
        wb[threadIdx.x >> 6][0] = o; wb[threadIdx.x >> 6][1] = z; wb[threadIdx.x >> 6][2] = w2; wb[threadIdx.x >> 6][3] = w3;
    }
    __syncthreads();
    if (threadIdx.x < 4)
        bits[((size_t)col * ntiles + tile) * 4 + threadIdx.x] =
            wb[0][threadIdx.x] | wb[1][threadIdx.x] | wb[2][threadIdx.x] | wb[3][threadIdx.x];
}

__global__ __launch_bounds__(64) void bin_plan_kernel(const uint64_t *__restrict__ bits, int ntiles,
                                                      BinPlanCol *__restrict__ plan)
{
    const int col = blockIdx.x, lane = threadIdx.x;
    uint64_t o = 0, z = 0, w2 = 0, w3 = 0;
    for (int t = lane; t < ntiles; t += 64) {
        o |= bits[((size_t)col * ntiles + t) * 4];
        z |= bits[((size_t)col * ntiles + t) * 4 + 1];
        w2 |= bits[((size_t)col * ntiles + t) * 4 + 2];
        w3 |= bits[((size_t)col * ntiles + t) * 4 + 3];
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        o |= __shfl_xor(o, off, 64);
        z |= __shfl_xor(z, off, 64);
        w2 |= __shfl_xor(w2, off, 64);
        w3 |= __shfl_xor(w3, off, 64);
    }
    if (lane != 0) return;
    const uint64_t varying = o & z;                           // a bit varies iff set in one key and clear in another
    BinPlanCol p;
    p.const_bits = 0;
    p.narrow = 1; p.npass = 0; p.lo_shift = 0; p.pad0 = 0;
    for (int j = 0; j < 4; ++j) p.dshift[j] = 0;
    int lo = -1, hi = -1;
    for (int b = 0; b < 8; ++b)
        if ((varying >> (8 * b)) & 0xFF) { if (lo < 0) lo = b; hi = b; }
    if (lo >= 0) {
        if (hi - lo < 4) {
            p.narrow = 1;
            p.lo_shift = (uint8_t)(8 * lo);
            // the 32 key bits from the window's lowest byte upwards are carried in the sorted key; every other
            // bit of the key is constant over the column and kept aside (o holds its value)
            p.const_bits = o & ~(0xFFFFFFFFull << (8 * lo));
            for (int b = lo; b <= hi; ++b)
                if ((varying >> (8 * b)) & 0xFF) p.dshift[p.npass++] = (uint8_t)(8 * (b - lo));
        } else {
            p.narrow = 0;
            p.lo_shift = (uint8_t)(8 * (hi - 3));
            for (int b = hi - 3; b <= hi; ++b)
                if ((varying >> (8 * b)) & 0xFF) p.dshift[p.npass++] = (uint8_t)(8 * b);
        }
        // integer column: the integer key when it needs fewer rounds than the fp64 pattern
        if (((w2 >> 32) & 1ull) == 0) {
            const uint32_t vi = (uint32_t)w2 & (uint32_t)w3;
            int rounds = 0;
            for (int b = 0; b < 4; ++b) rounds += ((vi >> (8 * b)) & 0xFF) ? 1 : 0;
            if (rounds < p.npass) {
                p.narrow = 1;
                p.lo_shift = (uint8_t)LO_SHIFT_INT;
                p.const_bits = 0;
                p.npass = 0;
                for (int j = 0; j < 4; ++j) p.dshift[j] = 0;
                for (int b = 0; b < 4; ++b)
                    if ((vi >> (8 * b)) & 0xFF) p.dshift[p.npass++] = (uint8_t)(8 * b);
            }
        }
    }
    plan[col] = p;
}

// keys of one thread (wave-contiguous slices like load_keys); round 0 converts the fp64 input
// src_kind: 0 = stored keys of an earlier round, 1 = the fp64 input column, 2 = an int64 input column
template <typename KeyT>
__device__ __forceinline__ void load_keys2(const void *__restrict__ src, int src_kind, int lo_shift, int64_t n,
                                           int64_t tile_base, KeyT (&keys)[SORT_ITEMS], uint32_t &valid_mask)
{
    const bool from_f64 = src_kind != 0;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t base = tile_base + (int64_t)wave * 64 * SORT_ITEMS + lane;
    const int64_t last = n > 0 ? n - 1 : 0;
    valid_mask = 0;
    if (from_f64) {
        // loads first, conversion behind a scheduling barrier (see load_keys)
        double raw[SORT_ITEMS];
#pragma unroll
        for (int i = 0; i < SORT_ITEMS; ++i) {
            const int64_t idx = base + (int64_t)i * 64;
            raw[i] = reinterpret_cast<const double *>(src)[idx < n ? idx : last];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < SORT_ITEMS; ++i) {
            const bool ok = base + (int64_t)i * 64 < n;
            valid_mask |= ok ? 1u << i : 0u;
            const KeyT k = lo_shift == LO_SHIFT_INT ? (KeyT)(uint32_t)(raw[i] + 0.0)
                                                    : (KeyT)(value_key(raw[i], src_kind == 2) >> lo_shift);
            keys[i] = ok ? k : (KeyT)~(KeyT)0;
        }
    } else {
        KeyT raw[SORT_ITEMS];
#pragma unroll
        for (int i = 0; i < SORT_ITEMS; ++i) {
            const int64_t idx = base + (int64_t)i * 64;
            raw[i] = reinterpret_cast<const KeyT *>(src)[idx < n ? idx : last];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < SORT_ITEMS; ++i) {
            const bool ok = base + (int64_t)i * 64 < n;
            valid_mask |= ok ? 1u << i : 0u;
            keys[i] = ok ? raw[i] : (KeyT)~(KeyT)0;
        }
    }
}

template <typename KeyT>
__device__ __forceinline__ void count_body(const void *src, int src_kind, int lo_shift, int dshift, int64_t n,
                                           int tile, uint32_t *cnt)
{
    KeyT keys[SORT_ITEMS];
    uint32_t vm;
    if (src_kind) load_keys2<KeyT>(src, src_kind, lo_shift, n, (int64_t)tile * SORT_TILE, keys, vm);
    else load_keys2<KeyT>(src, 0, 0, n, (int64_t)tile * SORT_TILE, keys, vm);
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int i = 0; i < SORT_ITEMS; ++i) {
        const bool valid = (vm >> i) & 1u;
        const uint32_t d = (uint32_t)(keys[i] >> dshift) & 0xFF;
        const uint64_t active = __ballot(valid);
        if (active == 0) continue;
        const int leader = __ffsll((long long)active) - 1;
        const uint32_t d0 = __shfl(d, leader, 64);
        if (__ballot(valid && d != d0) == 0) {
            if (lane == leader) atomicAdd(&cnt[d0], (uint32_t)__popcll(active));
        } else if (valid) {
            atomicAdd(&cnt[d], 1u);
        }
    }
}

// source / destination of round `round` for a column: round 0 reads the fp64 input and writes buffer A,
// then A -> B -> A -> B (each column has an n * 8 byte slot in both buffers; NARROW uses half of it)
__device__ __forceinline__ const void *round_src(int round, int col, const double *cols, int64_t ld, int64_t n,
                                                 const uint64_t *buf_a, const uint64_t *buf_b)
{
    if (round == 0) return cols + (size_t)col * ld;
    return ((round & 1) ? buf_a : buf_b) + (size_t)col * n;
}

__global__ __launch_bounds__(SORT_THREADS) void tile_count2_kernel(const double *__restrict__ cols, int64_t ld,
                                                                   int64_t n, int round, int ntiles,
                                                                   const BinPlanCol *__restrict__ plan,
                                                                   const uint64_t *buf_a, const uint64_t *buf_b,
                                                                   uint32_t *__restrict__ hist, ColFlags flags)
{
    __shared__ uint32_t cnt[RADIX];
    const int col = blockIdx.y, tile = blockIdx.x;
    if (round >= plan[col].npass) return;
    const int src_kind = round == 0 ? (col_is_i64(flags, col) ? 2 : 1) : 0;
    const int narrow = plan[col].narrow, lo_shift = plan[col].lo_shift, dshift = plan[col].dshift[round];
    cnt[threadIdx.x] = 0;
    __syncthreads();
    const void *src = round_src(round, col, cols, ld, n, buf_a, buf_b);
    if (narrow) count_body<uint32_t>(src, src_kind, lo_shift, dshift, n, tile, cnt);
    else count_body<uint64_t>(src, src_kind, 0, dshift, n, tile, cnt);
    __syncthreads();
    hist[((size_t)col * RADIX + threadIdx.x) * ntiles + tile] = cnt[threadIdx.x];
}

__global__ __launch_bounds__(64) void scan_rows2_kernel(uint32_t *__restrict__ hist, int ntiles,
                                                        uint32_t *__restrict__ tot,
                                                        const BinPlanCol *__restrict__ plan, int round)
{
    const int d = blockIdx.x, col = blockIdx.y, lane = threadIdx.x;
    if (round >= plan[col].npass) return;
    uint32_t *row = hist + ((size_t)col * RADIX + d) * ntiles;
    uint32_t carry = 0;
    for (int t0 = 0; t0 < ntiles; t0 += 64 * 4) {
        uint32_t x[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int t = t0 + j * 64 + lane;
            x[j] = (t < ntiles) ? row[t] : 0u;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int t = t0 + j * 64 + lane;
            uint32_t inc = x[j];
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const uint32_t y = __shfl_up(inc, off, 64);
                if (lane >= off) inc += y;
            }
            if (t < ntiles) row[t] = carry + inc - x[j];
            carry += __shfl(inc, 63, 64);
        }
    }
    if (lane == 0) tot[(size_t)col * RADIX + d] = carry;
}

template <typename KeyT>
__device__ __forceinline__ void scatter_body(const void *src, int src_kind, int lo_shift, int dshift, void *dst,
                                             int64_t n, int tile, int ntiles, const uint32_t *offsets_col,
                                             const uint32_t *digit_tot_col, uint32_t (*cnt)[RADIX], uint32_t *gdelta,
                                             uint32_t *wsum, KeyT *stage)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    KeyT keys[SORT_ITEMS];
    uint32_t vm;
    if (src_kind) load_keys2<KeyT>(src, src_kind, lo_shift, n, (int64_t)tile * SORT_TILE, keys, vm);
    else load_keys2<KeyT>(src, 0, 0, n, (int64_t)tile * SORT_TILE, keys, vm);
    const uint64_t lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    uint32_t rank[SORT_ITEMS];
#pragma unroll
    for (int i = 0; i < SORT_ITEMS; ++i) {
        const bool valid = (vm >> i) & 1u;
        const uint32_t d = (uint32_t)(keys[i] >> dshift) & 0xFF;
        uint64_t peers = __ballot(valid);
        const uint32_t d0 = __shfl(d, peers ? __ffsll((long long)peers) - 1 : 0, 64);
        if (__ballot(valid && d != d0) != 0) {
#pragma unroll
            for (int bit = 0; bit < 8; ++bit) {
                const bool set = (d >> bit) & 1u;
                const uint64_t m = __ballot(set);
                peers &= set ? m : ~m;
            }
        }
        uint32_t r = 0;
        if (valid) {
            const uint32_t before = cnt[wave][d];
            const uint32_t in_group = (uint32_t)__popcll(peers & lt_mask);
            r = before + in_group;
            __builtin_amdgcn_wave_barrier();
            if (in_group == 0) cnt[wave][d] = before + (uint32_t)__popcll(peers);
        }
        __builtin_amdgcn_wave_barrier();
        rank[i] = r;
    }
    __syncthreads();
    {
        const int d = threadIdx.x;
        const uint32_t c0 = cnt[0][d], c1 = cnt[1][d], c2 = cnt[2][d], c3 = cnt[3][d];
        const uint32_t total = c0 + c1 + c2 + c3;
        const uint32_t gtot = digit_tot_col[d];
        uint32_t inc = total, ginc = gtot;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t y = __shfl_up(inc, off, 64), gy = __shfl_up(ginc, off, 64);
            if (lane >= off) { inc += y; ginc += gy; }
        }
        if (lane == 63) { wsum[wave] = inc; wsum[4 + wave] = ginc; }
        __syncthreads();
        uint32_t lp = inc - total, gbase = ginc - gtot;
        for (int w = 0; w < wave; ++w) { lp += wsum[w]; gbase += wsum[4 + w]; }
        const uint32_t g = offsets_col[(size_t)d * ntiles + tile] + gbase;
        gdelta[d] = g - lp;
        cnt[0][d] = lp;
        cnt[1][d] = lp + c0;
        cnt[2][d] = lp + c0 + c1;
        cnt[3][d] = lp + c0 + c1 + c2;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < SORT_ITEMS; ++i) {
        if ((vm >> i) & 1u) {
            const uint32_t d = (uint32_t)(keys[i] >> dshift) & 0xFF;
            stage[cnt[wave][d] + rank[i]] = keys[i];
        }
    }
    __syncthreads();
    const int64_t left = n - (int64_t)tile * SORT_TILE;
    const int nv = (int)(left < SORT_TILE ? left : SORT_TILE);
    for (int j = threadIdx.x; j < nv; j += SORT_THREADS) {
        const KeyT key = stage[j];
        const uint32_t pos = gdelta[(uint32_t)(key >> dshift) & 0xFF] + (uint32_t)j;
        reinterpret_cast<KeyT *>(dst)[pos] = key;
    }
}

__global__ __launch_bounds__(SORT_THREADS) void scatter2_kernel(const double *__restrict__ cols, int64_t ld,
                                                                int64_t n, int round, int ntiles,
                                                                const BinPlanCol *__restrict__ plan, uint64_t *buf_a,
                                                                uint64_t *buf_b, const uint32_t *__restrict__ offsets,
                                                                const uint32_t *__restrict__ digit_tot, ColFlags flags)
{
    __shared__ uint32_t cnt[4][RADIX];
    __shared__ uint32_t gdelta[RADIX];
    __shared__ uint32_t wsum[8];
    __shared__ uint64_t stage[SORT_TILE];
    const int col = blockIdx.y, tile = blockIdx.x;
    if (round >= plan[col].npass) return;
    const int narrow = plan[col].narrow, lo_shift = plan[col].lo_shift, dshift = plan[col].dshift[round];
    const int src_kind = round == 0 ? (col_is_i64(flags, col) ? 2 : 1) : 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) cnt[w][threadIdx.x] = 0;
    __syncthreads();
    const void *src = round_src(round, col, cols, ld, n, buf_a, buf_b);
    void *dst = ((round & 1) ? buf_b : buf_a) + (size_t)col * n;
    const uint32_t *off_col = offsets + (size_t)col * RADIX * ntiles;
    const uint32_t *tot_col = digit_tot + (size_t)col * RADIX;
    if (narrow)
        scatter_body<uint32_t>(src, src_kind, lo_shift, dshift, dst, n, tile, ntiles, off_col, tot_col, cnt, gdelta, wsum,
                               reinterpret_cast<uint32_t *>(stage));
    else
        scatter_body<uint64_t>(src, src_kind, 0, dshift, dst, n, tile, ntiles, off_col, tot_col, cnt, gdelta, wsum, stage);
}

// ---- threshold walk on a window-sorted column --------------------------------------------------
// One workgroup per column.  prune.py:33-54: repeat { size = max(int(frac * unbinned), 1); hi = value at
// sorted position done + size - 1; the bin ends at the end of hi's tie run }.  The searches are executed by
// every wavefront alike (their loads hit the same lines); only the scan of a long unresolved run is split
// over the waves.
template <typename KeyT>
struct KeyArray {
    const KeyT *k;
    __device__ __forceinline__ KeyT operator[](int64_t i) const { return k[i]; }
};

// first index in [L, R) whose key's high part differs from h, given that keys are sorted by high part,
// that the high part at L - 1 equals h and that every key in [L, R) has a high part >= h
template <typename KeyT>
__device__ __forceinline__ int64_t run_end(const KeyArray<KeyT> s, int shift, uint64_t h, int64_t L, int64_t R)
{
    const int lane = threadIdx.x & 63;
    // most runs end within a few keys: probe the next 64 positions with one load per lane
    {
        const int64_t idx = L + lane;
        const bool eq = (idx < R) && (((uint64_t)s[idx] >> shift) == h);
        const uint64_t m = __ballot(eq);
        const int c = (m == ~0ull) ? 64 : __ffsll((long long)~m) - 1;       // leading lanes that still match
        if (c < 64 || L + 64 >= R) return (L + c < R) ? L + c : R;
        L += 64;
    }
    while (L < R) {
        const int64_t len = R - L;
        const int64_t step = (len + 63) >> 6;
        const int64_t idx = L + (int64_t)lane * step;
        const bool eq = (idx < R) && (((uint64_t)s[idx] >> shift) == h);
        const int c = __popcll(__ballot(eq));
        if (c == 0) {
            R = L;
        } else {
            const int64_t nL = L + (int64_t)(c - 1) * step + 1;
            const int64_t cap = L + (int64_t)c * step;
            R = (cap < R) ? cap : R;
            L = nL;
        }
    }
    return L;
}

// first index in [L, R] whose high part equals h, given that it does at R and keys are sorted by high part
template <typename KeyT>
__device__ __forceinline__ int64_t run_begin(const KeyArray<KeyT> s, int shift, uint64_t h, int64_t L, int64_t R)
{
    const int lane = threadIdx.x & 63;
    {
        const int64_t idx = R - 1 - lane;
        const bool eq = (idx >= L) && (((uint64_t)s[idx] >> shift) == h);
        const uint64_t m = __ballot(eq);
        const int c = (m == ~0ull) ? 64 : __ffsll((long long)~m) - 1;
        if (c < 64 || R - 64 <= L) return (R - c > L) ? R - c : L;
        R -= 64;
    }
    while (L < R) {
        const int64_t len = R - L;
        const int64_t step = (len + 63) >> 6;
        const int64_t idx = L + (int64_t)lane * step;
        const bool below = (idx < R) && (((uint64_t)s[idx] >> shift) < h);
        const int c = __popcll(__ballot(below));                          // the first c sample points are below
        if (c == 0) {
            R = L;
        } else {
            const int64_t nL = L + (int64_t)(c - 1) * step + 1;
            const int64_t cap = L + (int64_t)c * step;
            R = (cap < R) ? cap : R;
            L = nL;
        }
    }
    return L;
}

__global__ __launch_bounds__(256) void bin_threshold2_kernel(const double *__restrict__ cols, int64_t ld, int64_t n,
                                                             double frac, const BinPlanCol *__restrict__ plan,
                                                             const uint64_t *__restrict__ buf_a,
                                                             const uint64_t *__restrict__ buf_b,
                                                             uint64_t *__restrict__ thr, int32_t *__restrict__ nbins,
                                                             ColFlags flags)
{
    // thresholds are stored as ORDER KEYS (value_key): one comparison rule for fp64 and int64 columns
    const int col = blockIdx.x;
    const BinPlanCol p = plan[col];
    uint64_t *t = thr + (size_t)col * GRX_MAX_BINS;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (p.npass == 0) {                                       // every value equal: one bin
        if (threadIdx.x == 0) { t[0] = value_key(cols[(size_t)col * ld], col_is_i64(flags, col)); nbins[col] = 1; }
        return;
    }
    const uint64_t *buf = ((p.npass & 1) ? buf_a : buf_b) + (size_t)col * n;
    __shared__ uint64_t s_red[8];
    __shared__ uint32_t s_hist[RADIX];
    __shared__ uint64_t s_pick[2];
    int64_t done = 0;
    int nb = 0;
    if (p.narrow) {
        const KeyArray<uint32_t> s{reinterpret_cast<const uint32_t *>(buf)};
        while (done < n && nb < GRX_MAX_BINS) {
            int64_t size = (int64_t)(frac * (double)(n - done));
            if (size < 1) size = 1;
            const int64_t pos = done + size - 1;
            const uint32_t k = s[pos];
            const int64_t end = run_end<uint32_t>(s, 0, (uint64_t)k, pos + 1, n);
            if (threadIdx.x == 0)
                t[nb] = p.lo_shift == LO_SHIFT_INT ? f64_to_key((double)k) : (((uint64_t)k << p.lo_shift) | p.const_bits);
            ++nb;
            done = end;
        }
    } else {
        const KeyArray<uint64_t> s{buf};
        const int shift = p.lo_shift;
        uint64_t prev_h = 0;
        int64_t prev_a = 0;
        while (done < n && nb < GRX_MAX_BINS) {
            int64_t size = (int64_t)(frac * (double)(n - done));
            if (size < 1) size = 1;
            const int64_t pos = done + size - 1;
            const uint64_t kpos = s[pos];
            const uint64_t h = kpos >> shift;
            const int64_t b = run_end<uint64_t>(s, shift, h, pos + 1, n);
            // a run is unordered inside: positions do not tell which of its keys earlier bins took, so a
            // threshold that lands in the run of the previous one is ranked against the WHOLE run again
            const int64_t a = (nb > 0 && h == prev_h) ? prev_a : run_begin<uint64_t>(s, shift, h, done, pos);
            prev_h = h;
            prev_a = a;
            const int64_t len = b - a, q = pos - a;           // the q-th smallest key of the run is the threshold
            uint64_t tk;
            int64_t end;
            if (len == 1) {
                tk = kpos;
                end = b;
            } else if (len <= 64) {
                // rank every key of the run against the others (one key per lane, 64-bit shuffles)
                const uint64_t mine = (lane < len) ? s[a + lane] : ~0ull;
                int lt = 0, le = 0;
                for (int j = 0; j < (int)len; ++j) {
                    const uint64_t other = __shfl(mine, j, 64);
                    lt += other < mine;
                    le += other <= mine;
                }
                const uint64_t hit = __ballot(lane < len && lt <= q && q < le);
                const int src = __ffsll((long long)hit) - 1;
                tk = __shfl(mine, src, 64);
                end = a + __shfl(le, src, 64);
            } else {
                // long run: all keys equal (heavy ties)?  min / max over the run, split over the workgroup
                uint64_t mn = ~0ull, mx = 0;
                for (int64_t i = a + threadIdx.x; i < b; i += 256) {
                    const uint64_t v = s[i];
                    mn = v < mn ? v : mn;
                    mx = v > mx ? v : mx;
                }
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) {
                    const uint64_t o1 = __shfl_xor(mn, off, 64), o2 = __shfl_xor(mx, off, 64);
                    mn = o1 < mn ? o1 : mn;
                    mx = o2 > mx ? o2 : mx;
                }
                __syncthreads();
                if (lane == 0) { s_red[wave] = mn; s_red[4 + wave] = mx; }
                __syncthreads();
                mn = s_red[0]; mx = s_red[4];
                for (int w = 1; w < 4; ++w) { mn = s_red[w] < mn ? s_red[w] : mn; mx = s_red[4 + w] > mx ? s_red[4 + w] : mx; }
                if (mn == mx) {
                    tk = mn;
                    end = b;
                } else {
                    // radix selection of the q-th smallest key over the unsorted low bytes, most significant first
                    uint64_t prefix = h << shift;
                    int64_t below = 0, qrem = q, equal = len;
                    for (int byte = shift / 8 - 1; byte >= 0; --byte) {
                        __syncthreads();
                        s_hist[threadIdx.x] = 0;
                        __syncthreads();
                        const int hs = 8 * (byte + 1);
                        for (int64_t i = a + threadIdx.x; i < b; i += 256) {
                            const uint64_t v = s[i];
                            if ((v >> hs) == (prefix >> hs)) atomicAdd(&s_hist[(uint32_t)(v >> (8 * byte)) & 0xFF], 1u);
                        }
                        __syncthreads();
                        if (threadIdx.x == 0) {
                            int64_t cum = 0;
                            int d = 0;
                            for (; d < 255; ++d) {
                                if (cum + (int64_t)s_hist[d] > qrem) break;
                                cum += s_hist[d];
                            }
                            s_pick[0] = (uint64_t)d;
                            s_pick[1] = (uint64_t)cum;
                        }
                        __syncthreads();
                        const int d = (int)s_pick[0];
                        prefix |= (uint64_t)d << (8 * byte);
                        below += (int64_t)s_pick[1];
                        qrem -= (int64_t)s_pick[1];
                        equal = (int64_t)s_hist[d];
                    }
                    tk = prefix;
                    end = a + below + equal;
                }
            }
            if (threadIdx.x == 0) t[nb] = tk;
            ++nb;
            done = end;
        }
    }
    // more than GRX_MAX_BINS bins (only reachable with a tiny frac): reported as a negative count
    if (threadIdx.x == 0) nbins[col] = (done < n) ? -nb : nb;
}

__global__ __launch_bounds__(256) void bin_assign_kernel(const double *__restrict__ cols, int64_t ld,
                                                         int64_t n, const uint64_t *__restrict__ thr,
                                                         const int32_t *__restrict__ nbins,
                                                         uint8_t *__restrict__ bins, int64_t ld_bins, ColFlags flags,
                                                         const int32_t *__restrict__ fault = nullptr,
                                                         int32_t *__restrict__ status = nullptr)
{
    __shared__ uint64_t t[GRX_MAX_BINS];
    const int col = blockIdx.y;
    const bool i64 = col_is_i64(flags, col);
    int nb = nbins[col];
    // outcome flags for a caller that asked for them (grx_internal_vertical_log_bin): status[0] = 1 when the sort-free
    // walk met a bucket it had not marked (an invariant of the interval walk broke), status[1] = 1 when a column needs
    // more than GRX_MAX_BINS bins (its labels saturate below).  Two WORDS, not two bits of one: sharded runs combine
    // them across ranks with a MAX reduction, which keeps each word exact
    if (status && blockIdx.x == 0 && threadIdx.x == 0) {
        if (col == 0 && fault && *fault != 0) status[0] = 1;
        if (nb < 0) status[1] = 1;
    }
    if (nb < 0) nb = GRX_MAX_BINS;                    // more than GRX_MAX_BINS bins: labels saturate, the caller is told
    if (threadIdx.x < GRX_MAX_BINS)
        t[threadIdx.x] = (threadIdx.x < nb) ? thr[(size_t)col * GRX_MAX_BINS + threadIdx.x] : 0ull;
    __syncthreads();
    const double *x = cols + (size_t)col * ld;
    uint8_t *o = bins + (size_t)col * ld_bins;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const uint64_t v = value_key(x[i], i64);
        int lo = 0, hi = nb;                 // first threshold >= v (order keys compare like the values)
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (t[mid] < v) lo = mid + 1; else hi = mid;
        }
        o[i] = (uint8_t)lo;
    }
}

// ---------------------------------------------------------------------------------------
// Chebyshev distance between binned columns
// ---------------------------------------------------------------------------------------
constexpr int CH_ROWS = 512;                 // rows per LDS tile
constexpr int CH_STRIDE = CH_ROWS + 8;       // +8 bytes: consecutive columns start two banks apart
constexpr int CH_MAX_F = 96;                 // F * CH_STRIDE + 3 bytes per pair <= 64 KiB of LDS

// max over the 4 bytes of |x - y| for bytes <= 127 (bin labels are < 128)
__device__ __forceinline__ uint32_t ch_absdiff_max4(uint32_t x, uint32_t y)
{
    uint32_t m = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int xa = (x >> (8 * j)) & 0xFF, yb = (y >> (8 * j)) & 0xFF;
        const uint32_t d = (uint32_t)(xa > yb ? xa - yb : yb - xa);
        m = d > m ? d : m;
    }
    return m;
}

// One workgroup walks a strip of row tiles; a tile of 512 rows x F byte columns sits in LDS.
// One WAVEFRONT per column pair: lane l compares rows 8l..8l+7 of the tile, a butterfly gives the
// pair's maximum over the tile, and the pair is dropped from the workgroup's list as soon as its
// maximum exceeds `cap` -- the pruner only asks "distance <= generation number ?" (prune.py:110-113),
// and all but a handful of pairs exceed that within the first tile.  Distances <= cap are exact;
// larger ones are reported as some value > cap (cap = 255: everything exact).
// The F local columns are the global columns [a0, a0+na) followed by [b0, b0+F-na); dist is the
// global ldF x ldF matrix (more than CH_MAX_F columns are covered by several launches).
__global__ __launch_bounds__(256) void chebyshev_kernel(int64_t row_begin, int64_t row_end, int F,
                                                        int first_new, GrxPtrTable ptr_tab,
                                                        int32_t *__restrict__ dist, int ldF, int a0, int na,
                                                        int b0, int cap, int tiles_per_block, int filter, int psplit,
                                                        int64_t tile_stride)
{
    // tile_stride > 1 (the sample stage): the block's tile t lies at rows row_begin + t * tile_stride * CH_ROWS -- a
    // sample SPREAD over the row range instead of its first rows
    // psplit > 1 (the sample stage: few tiles, every pair): the grid is psplit times the tile blocks, block b takes
    // the tiles of block b % (gridDim.x / psplit) and the pairs with id % psplit == b / (gridDim.x / psplit) -- sixteen
    // workgroups walking thousands of pairs each were 1.6 of the 4.8 ms this kernel cost at config 5
    const uint8_t *const *ptrs = reinterpret_cast<const uint8_t *const *>(ptr_tab.p);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char *tile = smem;                                            // F * CH_STRIDE
    const int q0 = first_new > 1 ? first_new : 1;
    // pairs (p,q), q in [q0,F), p in [0,q): id = tri(q) - tri(q0) + p, tri(q) = q(q-1)/2
    const int tri0 = q0 * (q0 - 1) / 2;
    const int all_pairs = F * (F - 1) / 2 - tri0;
    uint16_t *pair_pq = reinterpret_cast<uint16_t *>(smem + (size_t)F * CH_STRIDE);   // (p << 8) | q
    uint8_t *pair_max = reinterpret_cast<uint8_t *>(pair_pq + all_pairs);              // running maximum
    __shared__ int n_listed;
    if (threadIdx.x == 0) n_listed = 0;
    __syncthreads();
    const int tile_blocks = (int)gridDim.x / psplit;
    const int part = (int)blockIdx.x / tile_blocks;
    for (int id = threadIdx.x; id < all_pairs; id += 256) {
        if (psplit > 1 && id % psplit != part) continue;
        const int target = id + tri0;
        int q = (int)((1.0f + sqrtf(1.0f + 8.0f * (float)target)) * 0.5f);
        while (q * (q - 1) / 2 > target) --q;
        while ((q + 1) * q / 2 <= target) ++q;
        const int p = target - q * (q - 1) / 2;
        int slot = psplit > 1 ? id / psplit : id;
        if (filter) {
            // second stage: only the pairs a first stage over a sample of rows left at <= cap
            const int gp = p < na ? a0 + p : b0 + (p - na), gq = q < na ? a0 + q : b0 + (q - na);
            slot = (dist[(size_t)gp * ldF + gq] <= cap) ? atomicAdd(&n_listed, 1) : -1;
        }
        if (slot >= 0) {
            pair_pq[slot] = (uint16_t)((p << 8) | q);
            pair_max[slot] = 0;
        }
    }
    __syncthreads();
    const int npairs = filter ? n_listed : (psplit > 1 ? (all_pairs - part + psplit - 1) / psplit : all_pairs);
    if (npairs <= 0) return;                                               // second stage: the sample settled every pair
    // second stage (round 5): only the columns that still have a pair within the cap are staged -- after the sample
    // stage that is a handful of the up to 128 columns of the set, and filling the LDS tile was what the stage spent its
    // time on (config 5: 0.6 - 0.8 ms per launch at 1 TB/s)
    __shared__ uint8_t col_needed[CH_MAX_F];
    __shared__ uint8_t col_list[CH_MAX_F];
    __shared__ int n_cols;
    if (filter) {
        for (int c = threadIdx.x; c < F; c += 256) col_needed[c] = 0;
        if (threadIdx.x == 0) n_cols = 0;
        __syncthreads();
        for (int id = threadIdx.x; id < npairs; id += 256) {
            col_needed[pair_pq[id] >> 8] = 1;
            col_needed[pair_pq[id] & 0xFF] = 1;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            int k = 0;
            for (int c = 0; c < F; ++c)
                if (col_needed[c]) col_list[k++] = (uint8_t)c;
            n_cols = k;
        }
        __syncthreads();
    }
    const int n_stage = filter ? n_cols : F;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    int64_t align_or = 0;                                                  // low address bits of all columns
    for (int c = 0; c < F; ++c) align_or |= (int64_t)(reinterpret_cast<uintptr_t>(ptrs[c]) & 3);
    const int64_t first_tile = (int64_t)(blockIdx.x % tile_blocks) * tiles_per_block;
    for (int tl = 0; tl < tiles_per_block; ++tl) {
        const int64_t r0 = row_begin + (first_tile + tl) * tile_stride * CH_ROWS;
        if (r0 >= row_end) break;
        const int rows = (int)((row_end - r0 < CH_ROWS) ? (row_end - r0) : CH_ROWS);
        __syncthreads();
        if (rows == CH_ROWS && ((r0 | align_or) & 3) == 0) {
            // full tile, 4-byte aligned columns: element e = (column, dword) over all 256 lanes; eight loads per
            // thread are issued before the first LDS store (a load-store loop waits for every load in turn)
            const int total = n_stage * (CH_ROWS / 4);
            for (int e0 = threadIdx.x; e0 < total; e0 += 256 * 8) {
                uint32_t v[8];
                int cc[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int e = e0 + 256 * j < total ? e0 + 256 * j : total - 1;
                    cc[j] = filter ? (int)col_list[e / (CH_ROWS / 4)] : e / (CH_ROWS / 4);
                    v[j] = reinterpret_cast<const uint32_t *>(ptrs[cc[j]] + r0)[e % (CH_ROWS / 4)];
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int e = e0 + 256 * j;
                    if (e < total) reinterpret_cast<uint32_t *>(tile + cc[j] * CH_STRIDE)[e % (CH_ROWS / 4)] = v[j];
                }
            }
        } else {
            for (int e = threadIdx.x; e < n_stage * CH_ROWS; e += 256) {
                const int c = filter ? (int)col_list[e / CH_ROWS] : e / CH_ROWS, i = e % CH_ROWS;
                tile[c * CH_STRIDE + i] = (i < rows) ? ptrs[c][r0 + i] : 0;
            }
        }
        // rows beyond `rows` are zero in every column -> contribute distance 0
        __syncthreads();
        // four pairs per wavefront and trip: their state and tile reads are issued together (with 64 KB of LDS per
        // workgroup two waves share a SIMD, and one pair at a time was a chain of four dependent LDS round trips)
        constexpr int U = 4;
        for (int id0 = wave; id0 < npairs; id0 += 4 * U) {
            uint32_t cur[U], pq[U];
            bool act[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int id = id0 + 4 * u;
                act[u] = id < npairs;
                cur[u] = act[u] ? pair_max[id] : 0u;
                pq[u] = act[u] ? pair_pq[id] : 0u;
            }
            uint2 x[U], y[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                act[u] = act[u] && (int)cur[u] <= cap;                     // uniform over the wavefront
                x[u] = *reinterpret_cast<const uint2 *>(tile + (pq[u] >> 8) * CH_STRIDE + lane * 8);
                y[u] = *reinterpret_cast<const uint2 *>(tile + (pq[u] & 0xFF) * CH_STRIDE + lane * 8);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (!act[u]) continue;
                const int id = id0 + 4 * u;
                uint32_t m = ch_absdiff_max4(x[u].x, y[u].x);
                const uint32_t m2 = ch_absdiff_max4(x[u].y, y[u].y);
                m = m2 > m ? m2 : m;
                if (__ballot((int)m > cap) != 0) {
                    // beyond the cap: the exact value is not needed, one ballot settles the pair
                    if (lane == 0) pair_max[id] = (uint8_t)(cap + 1);
                    continue;
                }
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) {
                    const uint32_t o = __shfl_xor(m, off, 64);
                    m = o > m ? o : m;
                }
                if (lane == 0 && m > cur[u]) pair_max[id] = (uint8_t)m;
            }
        }
    }
    __syncthreads();
    for (int id = threadIdx.x; id < npairs; id += 256) {
        const int mx = pair_max[id];
        if (mx == 0) continue;
        const int lp = pair_pq[id] >> 8, lq = pair_pq[id] & 0xFF;
        const int p = lp < na ? a0 + lp : b0 + (lp - na), q = lq < na ? a0 + lq : b0 + (lq - na);
        // hundreds of workgroups report the same pair and same-address read-modify-writes serialise at
        // the memory side: look first (agent-scope load, coherent across XCDs) and skip the atomic when
        // the matrix already holds a value at least as large
        if (__hip_atomic_load(&dist[(size_t)p * ldF + q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < mx) {
            atomicMax(&dist[(size_t)p * ldF + q], mx);
            atomicMax(&dist[(size_t)q * ldF + p], mx);
        }
    }
}

// =======================================================================================
// Binning WITHOUT sorting (round 3).  prune.py:13-56 needs ~20 order statistics of a column and the ends of their
// tie runs -- not a sorted column.  Per column:
//   sel_map_kernel      4095 strided samples -> [kmin, kmax] of the order keys -> a linear map of the key range onto
//                       4096 cells, refined by a LUT: a cell with s samples is split into 2^floor(log2 s) buckets,
//                       empty cells merge into their successor (keys outside the sampled range fall into the end
//                       buckets: the map is monotone, the counts exact).  Buckets whose samples are all equal are
//                       tie-block CANDIDATES.
//   sel_hist_kernel     ONE pass: bucket histogram (LDS, wave-aggregated atomics); a candidate bucket that receives a
//                       key other than its sampled value is struck off -- the rest are CERTIFIED blocks of ties
//   sel_walk1_kernel    prefix sums; the threshold walk in INTERVAL arithmetic over the bucket boundaries (the exact
//                       end of a tie run is unknown yet, so `done` is an interval [lo, hi] -- except in a certified
//                       block of ties, where the bin ends with the bucket): every bucket a threshold rank can fall
//                       into is marked -- typically one or two per threshold
//   sel_collect_kernel  ONE pass: the keys of the marked buckets (certified tie blocks excepted), appended to
//                       per-bucket segments, with each segment's smallest and largest key
//   sel_sort_kernel     every segment in ascending order, one workgroup each, in LDS
//   sel_walk2_kernel    the exact walk: the threshold is the q-th key of its bucket's segment and the bin ends with the
//                       threshold's tie run -- one trip to memory per threshold (segments too large for the LDS sort
//                       fall back to a workgroup radix selection) -> thresholds as order keys
//   bin_assign_kernel   as before.
// Three streaming passes of 8 bytes per key and 7 launches per call instead of key_bits + 4 x (count, scan, scatter) +
// walk + assign (~70 bytes per key, 15 launches).  Measured per bench step (profiles/r03_binning_ab.txt): BA 1 M
// 1.23 -> 0.67 ms, ER 100 k 0.64 -> 0.41 ms, directed weighted 5 M / 40 M 25.9 -> 7.3 ms.  The bins are identical
// (both paths are exact); the sort path stays behind GRX_BIN_SORT=1 and in the A/B tests.
// =======================================================================================
constexpr int SEL_NB = 4096;
constexpr int SEL_MAX_IDS = 512;                           // marked buckets with an LDS slot in the collect pass
constexpr int SEL_SORT_CAP = 8192;                         // keys of a segment the segment sort holds in LDS
constexpr uint16_t SEL_MARK_SORTED = 0x8000;               // mark bit: the segment is in ascending order
constexpr uint16_t SEL_MARK_TIES = 0xFFFF;                 // mark: a certified block of ties, nothing collected
constexpr uint16_t SEL_THR_EXACT = 0x8000;                 // thr_bucket bit: keys of the bucket need the exact compare
// columns from this height on keep their bucket ids (see sel_assign_kernel); GRX_BIN_BID_MIN_N overrides (tests: 0)
inline int64_t sel_bid_min_n()
{
    static const int64_t v = [] { const char *e = std::getenv("GRX_BIN_BID_MIN_N"); return e ? (int64_t)std::atoll(e) : (int64_t)2500000; }();
    return v;
}
__host__ __device__ inline int64_t sel_bid_stride(int64_t n) { return (n + 3) & ~(int64_t)3; }   // 8-byte aligned columns
constexpr int SEL_HIST_ITEMS = 32;                         // keys per thread of the histogram pass
constexpr int SEL_HIST_TILE = 256 * SEL_HIST_ITEMS;        // 8192 keys per workgroup
// The bucket map of a column: a LINEAR map of the sampled key range onto 4096 cells, refined by a look-up table built
// from 4095 strided samples -- a cell that holds s samples is split into 2^floor(log2 s) buckets (<= 4095 in total),
// empty cells share the first bucket of the next occupied cell.  One LDS read and a few shifts per key (a binary search
// over sorted sample splitters, tried first, cost 12 dependent LDS reads per key and tripled both streaming passes),
// and a bucket holds ~n / 4096 keys wherever the data is dense -- unless they are ties.  Monotone in the key, so
// every bucket is a contiguous range of the order.
struct SelMap { uint64_t kmin; int shift; int ncells; int nb; };
constexpr int SEL_NSAMPLE = SEL_NB - 1;

// LUT entry of a cell: first bucket | log2(number of buckets) << 16
__device__ __forceinline__ int sel_bucket(const SelMap &m, const uint32_t *__restrict__ lut, uint64_t key)
{
    if (key <= m.kmin) return 0;
    const uint64_t rel = key - m.kmin;
    uint64_t c = rel >> m.shift;
    if (c >= (uint64_t)m.ncells) return m.nb - 1;               // above the sampled range: the last bucket
    const uint32_t e = lut[c];
    const int k = (int)(e >> 16);
    const uint64_t within = rel - (c << m.shift);               // offset inside the cell, < 2^shift
    return (int)(e & 0xFFFFu) + (int)(within >> (m.shift - k));
}

constexpr int SEL_MAX_TIES = 511;                          // tie-block candidates per column (ids 1 .. 511)

__global__ __launch_bounds__(1024) void sel_map_kernel(const double *__restrict__ cols, int64_t ld, int64_t n,
                                                       SelMap *__restrict__ maps, uint32_t *__restrict__ luts,
                                                       uint16_t *__restrict__ tie_of_bucket, uint64_t *__restrict__ tie_value,
                                                       uint32_t *__restrict__ hist, uint32_t *__restrict__ cursor,
                                                       unsigned long long *__restrict__ bmin,
                                                       unsigned long long *__restrict__ bmax, uint8_t *__restrict__ tie_broken,
                                                       int32_t *__restrict__ fault, ColFlags flags)
{
    // the tables the later passes accumulate into, cleared here (the first kernel of the call, one workgroup per column)
    // instead of by three fill launches in front of it: 5 us each, four calls per ReFeX pass
    for (int b = threadIdx.x; b < SEL_NB; b += 1024) {
        const size_t cell = (size_t)blockIdx.x * SEL_NB + b;
        hist[cell] = 0; cursor[cell] = 0; bmax[cell] = 0ull; bmin[cell] = ~0ull; tie_broken[cell] = 0;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) *fault = 0;
    __shared__ unsigned long long smin[SEL_NB];
    __shared__ unsigned long long smax[SEL_NB];
    __shared__ uint32_t scnt[SEL_NB];
    __shared__ uint32_t s_lut[SEL_NB];
    __shared__ uint32_t cnt[SEL_NB];
    __shared__ uint64_t red[32];
    __shared__ uint32_t wsum[16];
    __shared__ SelMap s_map;
    const int col = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const bool i64 = col_is_i64(flags, col);
    const double *x = cols + (size_t)col * ld;
    uint64_t key[4];
    bool have[4];
    uint64_t mn = ~0ull, mx = 0ull;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int64_t k = (int64_t)t * 4 + j;                  // sample number
        have[j] = k < SEL_NSAMPLE && (n > SEL_NSAMPLE || k < n);
        key[j] = 0;
        if (have[j]) {
            key[j] = value_key(x[n > SEL_NSAMPLE ? (k * (n - 1)) / (SEL_NSAMPLE - 1) : k], i64);
            mn = key[j] < mn ? key[j] : mn;
            mx = key[j] > mx ? key[j] : mx;
        }
        cnt[4 * t + j] = 0;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const uint64_t a = __shfl_xor(mn, off, 64), b = __shfl_xor(mx, off, 64);
        mn = a < mn ? a : mn;
        mx = b > mx ? b : mx;
    }
    if (lane == 0) { red[wave] = mn; red[16 + wave] = mx; }
    __syncthreads();
    if (t == 0) {
        for (int w = 1; w < 16; ++w) { mn = red[w] < mn ? red[w] : mn; mx = red[16 + w] > mx ? red[16 + w] : mx; }
        const uint64_t span = mx - mn;
        const int bits = span ? 64 - __clzll((long long)span) : 0;
        s_map.kmin = mn;
        s_map.shift = bits > 12 ? bits - 12 : 0;
        s_map.ncells = (int)(span >> s_map.shift) + 1;
        s_map.nb = 0;
    }
    __syncthreads();
    const SelMap m = s_map;
#pragma unroll
    for (int j = 0; j < 4; ++j)
        if (have[j]) atomicAdd(&cnt[(key[j] - m.kmin) >> m.shift], 1u);
    __syncthreads();
    // buckets per cell: 2^min(floor(log2 s), shift) for s samples, none for an empty cell; exclusive prefix = first bucket
    uint32_t alloc[4], kk[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const uint32_t sc = cnt[4 * t + j];
        int k = sc ? 31 - __clz((int)sc) : 0;
        if (k > m.shift) k = m.shift;
        kk[j] = (uint32_t)k;
        alloc[j] = sc ? (1u << k) : 0u;
    }
    const uint32_t local = alloc[0] + alloc[1] + alloc[2] + alloc[3];
    uint32_t inc = local;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t y = __shfl_up(inc, off, 64);
        if (lane >= off) inc += y;
    }
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    uint32_t run = inc - local;
    for (int w = 0; w < wave; ++w) run += wsum[w];
    uint32_t *lut = luts + (size_t)col * SEL_NB;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const uint32_t e = run | (kk[j] << 16);
        lut[4 * t + j] = e;
        s_lut[4 * t + j] = e;
        run += alloc[j];
        smin[4 * t + j] = ~0ull; smax[4 * t + j] = 0ull; scnt[4 * t + j] = 0;
    }
    if (t == 1023) {
        // one more bucket behind the last cell's: keys above the sampled range and trailing empty cells end there
        s_map.nb = (int)run + 1;
        maps[col] = s_map;
    }
    __syncthreads();
    // Tie-block candidates: a bucket whose samples (two or more) are all equal.  The histogram pass CERTIFIES them --
    // it compares every key that lands in such a bucket with the sampled value and flags the bucket on a mismatch --
    // so that the interval walk knows exactly where a bin ends when its threshold falls into a block of ties (a
    // bucket of 30 % equal keys would otherwise leave the next thresholds anywhere in hundreds of buckets).
    const SelMap mm = s_map;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (have[j]) {
            const int b = sel_bucket(mm, s_lut, key[j]);
            atomicAdd(&scnt[b], 1u);
            atomicMin(&smin[b], (unsigned long long)key[j]);
            atomicMax(&smax[b], (unsigned long long)key[j]);
        }
    }
    __syncthreads();
    uint32_t cand[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) { const int b = 4 * t + j; cand[j] = (scnt[b] >= 2 && smin[b] == smax[b]) ? 1u : 0u; }
    const uint32_t lc = cand[0] + cand[1] + cand[2] + cand[3];
    uint32_t ic = lc;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t y = __shfl_up(ic, off, 64);
        if (lane >= off) ic += y;
    }
    __syncthreads();
    if (lane == 63) wsum[wave] = ic;
    __syncthreads();
    uint32_t id = ic - lc;
    for (int w = 0; w < wave; ++w) id += wsum[w];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int b = 4 * t + j;
        uint16_t tid = 0;
        if (cand[j]) {
            ++id;
            if (id <= SEL_MAX_TIES) { tid = (uint16_t)id; tie_value[(size_t)col * (SEL_MAX_TIES + 1) + id] = smin[b]; }
        }
        tie_of_bucket[(size_t)col * SEL_NB + b] = tid;
    }
}

template <bool STORE_BID>
__global__ __launch_bounds__(256) void sel_hist_kernel(const double *__restrict__ cols, int64_t ld, int64_t n,
                                                       const SelMap *__restrict__ maps, const uint32_t *__restrict__ luts,
                                                       const uint16_t *__restrict__ tie_of_bucket,
                                                       const uint64_t *__restrict__ tie_value, uint8_t *__restrict__ tie_broken,
                                                       uint32_t *__restrict__ hist, uint16_t *__restrict__ bid, ColFlags flags,
                                                       int tiles_per_wg)
{
    __shared__ uint32_t h[SEL_NB];
    __shared__ uint32_t lut[SEL_NB];
    __shared__ uint16_t tieb[SEL_NB];
    __shared__ uint64_t tiev[SEL_MAX_TIES + 1];
    const int col = blockIdx.y;
    const bool i64 = col_is_i64(flags, col);
    const SelMap m = maps[col];
    for (int b = threadIdx.x; b < SEL_NB; b += 256) {
        h[b] = 0;
        lut[b] = luts[(size_t)col * SEL_NB + b];
        tieb[b] = tie_of_bucket[(size_t)col * SEL_NB + b];
    }
    for (int k = threadIdx.x; k <= SEL_MAX_TIES; k += 256) tiev[k] = tie_value[(size_t)col * (SEL_MAX_TIES + 1) + k];
    __syncthreads();
    const double *x = cols + (size_t)col * ld;
    const int64_t last = n - 1;
    const int lane = threadIdx.x & 63;
    // (tall columns: several tiles per workgroup -- the tables above and the 4096 counters flushed below are a fixed
    // cost per workgroup, about as much LDS and atomic traffic as one tile of keys)
    for (int tile = 0; tile < tiles_per_wg; ++tile) {
    const int64_t base = ((int64_t)blockIdx.x * tiles_per_wg + tile) * SEL_HIST_TILE;
    if (base >= n) break;
    for (int i0 = 0; i0 < SEL_HIST_ITEMS; i0 += 8) {
        double raw[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int64_t idx = base + (int64_t)(i0 + j) * 256 + threadIdx.x;
            raw[j] = x[idx < n ? idx : last];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const bool valid = base + (int64_t)(i0 + j) * 256 + threadIdx.x < n;
            const uint64_t key = value_key(raw[j], i64);
            const int b = sel_bucket(m, lut, key);
            if (STORE_BID && valid)                               // for the two passes below (2 bytes per key instead of 8)
                bid[(size_t)col * sel_bid_stride(n) + base + (int64_t)(i0 + j) * 256 + threadIdx.x] = (uint16_t)b;
            const int tid = valid ? (int)tieb[b] : 0;
            if (tid && tiev[tid] != key) tie_broken[(size_t)col * SEL_NB + b] = 1;     // not a block of ties after all
            // heavy ties put a whole wavefront into one bucket: one atomic for all of it
            const uint64_t active = __ballot(valid);
            if (active == 0) continue;
            const int b0 = __shfl(b, __ffsll((long long)active) - 1, 64);
            if (__ballot(valid && b != b0) == 0) {
                if (lane == __ffsll((long long)active) - 1) atomicAdd(&h[b0], (uint32_t)__popcll(active));
            } else if (valid) {
                atomicAdd(&h[b], 1u);
            }
        }
    }
    }
    __syncthreads();
    uint32_t *out = hist + (size_t)col * SEL_NB;
    for (int b = threadIdx.x; b < SEL_NB; b += 256)
        if (h[b]) atomicAdd(&out[b], h[b]);
}

// prefix sums of the bucket counts, the interval walk that marks the buckets a threshold can fall into, and the
// segment offsets of the marked buckets.  One workgroup of 1024 threads per column (four buckets per thread).
__global__ __launch_bounds__(1024) void sel_walk1_kernel(int64_t n, double frac,
                                                         const uint32_t *__restrict__ hist, const uint16_t *__restrict__ tie_of_bucket,
                                                         const uint8_t *__restrict__ tie_broken,
                                                         const uint64_t *__restrict__ tie_value, uint32_t *__restrict__ cum,
                                                         uint16_t *__restrict__ mark, uint32_t *__restrict__ seg_off,
                                                         uint16_t *__restrict__ idlist, int32_t *__restrict__ nids,
                                                         unsigned long long *__restrict__ bmin,
                                                         unsigned long long *__restrict__ bmax)
{
    __shared__ uint8_t TIE[SEL_NB];                              // certified block of ties
    __shared__ uint32_t C[SEL_NB];
    __shared__ uint32_t S[SEL_NB];
    __shared__ uint8_t M[SEL_NB];
    __shared__ uint32_t wsum[16];
    const int col = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
    constexpr int nb = SEL_NB;
    const uint32_t *h = hist + (size_t)col * SEL_NB;
    auto scan4 = [&](uint32_t (&v)[4], uint32_t *dst) {            // inclusive scan of 4096 values, 4 per thread, into dst
        const uint32_t local = v[0] + v[1] + v[2] + v[3];
        uint32_t inc = local;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t y = __shfl_up(inc, off, 64);
            if (lane >= off) inc += y;
        }
        __syncthreads();
        if (lane == 63) wsum[wave] = inc;
        __syncthreads();
        uint32_t before = inc - local;
        for (int w = 0; w < wave; ++w) before += wsum[w];
        uint32_t run = before;
#pragma unroll
        for (int j = 0; j < 4; ++j) { run += v[j]; dst[4 * t + j] = run; }
        __syncthreads();
    };
    uint32_t v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int b = 4 * t + j;
        v[j] = b < nb ? h[b] : 0u;
        M[b] = 0;
        TIE[b] = (tie_of_bucket[(size_t)col * SEL_NB + b] != 0 && tie_broken[(size_t)col * SEL_NB + b] == 0) ? 1 : 0;
    }
    scan4(v, C);
    if (wave == 0) {
        // first bucket whose inclusive prefix exceeds the rank: two 64-ary ballot steps (the whole wavefront walks)
        auto bucket_of = [&](int64_t pos) {
            const int coarse = lane * 64 + 63;
            const bool gt = coarse < nb ? (int64_t)C[coarse] > pos : true;
            const int blk = __ffsll((long long)__ballot(gt)) - 1;
            const int fine = blk * 64 + lane;
            const bool gt2 = fine < nb ? (int64_t)C[fine] > pos : true;
            return blk * 64 + __ffsll((long long)__ballot(gt2)) - 1;
        };
        int64_t dlo = 0, dhi = 0;
        for (int step = 0; step < GRX_MAX_BINS && dlo < n; ++step) {
            int64_t slo = (int64_t)(frac * (double)(n - dlo));
            if (slo < 1) slo = 1;
            const int64_t plo = dlo + slo - 1;
            const int64_t dh = dhi < n ? dhi : n - 1;           // states that are already done need no threshold
            int64_t shi = (int64_t)(frac * (double)(n - dh));
            if (shi < 1) shi = 1;
            int64_t phi = dh + shi - 1;
            if (phi < plo) phi = plo;
            const int jlo = bucket_of(plo), jhi = bucket_of(phi);
            for (int j = jlo + lane; j <= jhi; j += 64) M[j] = 1;
            // the bin ends at or after its threshold's rank and inside the threshold's bucket -- at the END of the bucket
            // when that is a certified block of ties
            dlo = TIE[jlo] ? (int64_t)C[jlo] : plo + 1;
            dhi = (int64_t)C[jhi];
            if (dhi < dlo) dhi = dlo;
        }
    }
    __syncthreads();
    // a marked bucket that is a certified block of ties needs no keys: its one value is the smallest and the largest
    bool collect[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int b = 4 * t + j;
        const uint32_t cnt = b < nb ? (C[b] - (b ? C[b - 1] : 0u)) : 0u;
        collect[j] = M[b] && !TIE[b];
        v[j] = collect[j] ? cnt : 0u;
        if (M[b] && TIE[b]) {
            const size_t cell = (size_t)col * SEL_NB + b;
            const uint64_t value = tie_value[(size_t)col * (SEL_MAX_TIES + 1) + tie_of_bucket[cell]];
            bmin[cell] = value;
            bmax[cell] = value;
        }
    }
    scan4(v, S);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int b = 4 * t + j;
        if (b < nb) {
            cum[(size_t)col * SEL_NB + b] = C[b];
            seg_off[(size_t)col * SEL_NB + b] = S[b] - v[j];    // exclusive
        }
    }
    // compact ids of the collected buckets (1-based; SEL_MAX_IDS and beyond share the last id: the collect pass serves
    // those through global atomics) and the list of them for the segment sort
    uint32_t m[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) m[j] = collect[j] ? 1u : 0u;
    scan4(m, S);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int b = 4 * t + j;
        if (b < nb) {
            const uint32_t id = S[b];                            // inclusive count = 1-based id
            uint16_t mk = 0;
            if (m[j]) {
                mk = (uint16_t)(id < SEL_MAX_IDS ? id : SEL_MAX_IDS);
                if (id < SEL_MAX_IDS) idlist[(size_t)col * SEL_MAX_IDS + id - 1] = (uint16_t)b;
            } else if (M[b]) {
                mk = SEL_MARK_TIES;
            }
            mark[(size_t)col * SEL_NB + b] = mk;
        }
    }
    if (t == 1023) nids[col] = (int32_t)(S[nb - 1] < (uint32_t)SEL_MAX_IDS ? S[nb - 1] : SEL_MAX_IDS - 1);
}

// The collected segments in ascending order, each by one workgroup in LDS (bitonic network).  The exact walk then reads
// a threshold and the end of its tie run with ONE trip to memory instead of a multi-pass selection per threshold -- the
// walk is a serial chain of ~20 thresholds per column, its latency is the launch's duration.  Blocks of ties, segments
// beyond the LDS capacity and those of the buckets without a compact id stay unordered: the walk ranks up to 64 keys by
// shuffles and selects by radix passes beyond.
__global__ __launch_bounds__(512) void sel_sort_kernel(const uint32_t *__restrict__ cum, const uint32_t *__restrict__ seg_off,
                                                       const uint16_t *__restrict__ idlist, const int32_t *__restrict__ nids,
                                                       const unsigned long long *__restrict__ bmin,
                                                       const unsigned long long *__restrict__ bmax, int64_t n,
                                                       uint64_t *__restrict__ coll, uint16_t *__restrict__ mark)
{
    __shared__ uint64_t s[SEL_SORT_CAP];
    const int col = blockIdx.y;
    const int count = nids[col];
    for (int id = blockIdx.x; id < count; id += gridDim.x) {
        const int b = idlist[(size_t)col * SEL_MAX_IDS + id];
        const size_t cell = (size_t)col * SEL_NB + b;
        const int64_t len = (int64_t)cum[cell] - (b ? (int64_t)cum[cell - 1] : 0);
        if (len < 2 || len > SEL_SORT_CAP || bmin[cell] == bmax[cell]) continue;
        uint64_t *seg = coll + (size_t)col * n + seg_off[cell];
        int P = 2;
        while (P < len) P <<= 1;
        __syncthreads();
        for (int i = threadIdx.x; i < P; i += 512) s[i] = i < len ? seg[i] : ~0ull;
        for (int k = 2; k <= P; k <<= 1) {
            for (int j = k >> 1; j > 0; j >>= 1) {
                __syncthreads();
                for (int i = threadIdx.x; i < P / 2; i += 512) {
                    const int lo = 2 * i - (i & (j - 1)), hi = lo + j;
                    const uint64_t a = s[lo], c = s[hi];
                    const bool up = (lo & k) == 0;
                    if ((a > c) == up) { s[lo] = c; s[hi] = a; }
                }
            }
        }
        __syncthreads();
        for (int i = threadIdx.x; i < len; i += 512) seg[i] = s[i];
        if (threadIdx.x == 0) mark[cell] |= SEL_MARK_SORTED;
    }
}

// FROM_BID: the bucket of a key is read from the ids the histogram pass stored; the key itself only when its bucket is marked
template <bool FROM_BID>
__global__ __launch_bounds__(256) void sel_collect_kernel(const double *__restrict__ cols, int64_t ld, int64_t n,
                                                          const uint16_t *__restrict__ bid,
                                                          const SelMap *__restrict__ maps, const uint32_t *__restrict__ luts,
                                                          const uint16_t *__restrict__ mark,
                                                          const uint32_t *__restrict__ seg_off, uint32_t *__restrict__ cursor,
                                                          uint64_t *__restrict__ coll, unsigned long long *__restrict__ bmin,
                                                          unsigned long long *__restrict__ bmax, ColFlags flags,
                                                          int tiles_per_wg)
{
    // per marked bucket (compact id): how many keys of this tile, their smallest and largest, the bucket itself
    __shared__ uint32_t cnt[SEL_MAX_IDS];
    __shared__ uint32_t basev[SEL_MAX_IDS];
    __shared__ uint16_t bucket_of_id[SEL_MAX_IDS];
    __shared__ unsigned long long lo_id[SEL_MAX_IDS];
    __shared__ unsigned long long hi_id[SEL_MAX_IDS];
    __shared__ uint16_t M[SEL_NB];
    __shared__ uint32_t lut[SEL_NB];
    const int col = blockIdx.y;
    const bool i64 = col_is_i64(flags, col);
    const SelMap m = maps[col];
    for (int b = threadIdx.x; b < SEL_NB; b += 256) { M[b] = mark[(size_t)col * SEL_NB + b]; lut[b] = FROM_BID ? 0u : luts[(size_t)col * SEL_NB + b]; }
    const double *x = cols + (size_t)col * ld;
    const int64_t last = n - 1;
    // (tall columns: several tiles per workgroup, the 4096 marks are loaded once)
    for (int tile = 0; tile < tiles_per_wg; ++tile) {
    const int64_t base = ((int64_t)blockIdx.x * tiles_per_wg + tile) * SORT_TILE;
    if (base >= n) break;
    __syncthreads();                                            // (the previous tile's slots are no longer read)
    for (int k = threadIdx.x; k < SEL_MAX_IDS; k += 256) { cnt[k] = 0; lo_id[k] = ~0ull; hi_id[k] = 0ull; }
    __syncthreads();
    double raw[SORT_ITEMS];
    uint16_t bv[SORT_ITEMS];
    const uint16_t *bx = FROM_BID ? bid + (size_t)col * sel_bid_stride(n) : nullptr;
#pragma unroll
    for (int i = 0; i < SORT_ITEMS; ++i) {
        const int64_t idx = base + (int64_t)i * 256 + threadIdx.x;
        if (FROM_BID) { bv[i] = bx[idx < n ? idx : last]; raw[i] = 0.0; }
        else { raw[i] = x[idx < n ? idx : last]; bv[i] = 0; }
    }
    __builtin_amdgcn_sched_barrier(0);
    uint64_t *dst = coll + (size_t)col * n;
    uint32_t *cur = cursor + (size_t)col * SEL_NB;
    const uint32_t *off = seg_off + (size_t)col * SEL_NB;
    uint64_t keys[SORT_ITEMS];
    int rank[SORT_ITEMS], id[SORT_ITEMS];
    bool any = false;
#pragma unroll
    for (int i = 0; i < SORT_ITEMS; ++i) {
        const int64_t idx = base + (int64_t)i * 256 + threadIdx.x;
        const bool valid = idx < n;
        keys[i] = FROM_BID ? 0ull : value_key(raw[i], i64);
        const int b = FROM_BID ? (int)bv[i] : sel_bucket(m, lut, keys[i]);
        const int mk = valid ? (int)M[b] : 0;
        if (FROM_BID && mk > 0) keys[i] = value_key(x[idx], i64);
        id[i] = mk - 1;
        rank[i] = -1;
        if (mk > 0 && mk < SEL_MAX_IDS) {
            rank[i] = (int)atomicAdd(&cnt[mk - 1], 1u);
            atomicMin(&lo_id[mk - 1], (unsigned long long)keys[i]);
            atomicMax(&hi_id[mk - 1], (unsigned long long)keys[i]);
            bucket_of_id[mk - 1] = (uint16_t)b;                 // the same value from every writer
            any = true;
        } else if (mk == SEL_MAX_IDS) {
            // more marked buckets than LDS slots (a huge unresolved bucket followed by thousands of thin ones):
            // straight to the segment through global atomics
            const size_t cell = (size_t)col * SEL_NB + b;
            dst[off[b] + atomicAdd(&cur[b], 1u)] = keys[i];
            atomicMin(&bmin[cell], (unsigned long long)keys[i]);
            atomicMax(&bmax[cell], (unsigned long long)keys[i]);
        }
    }
    if (__syncthreads_or(any) == 0) continue;                   // nothing of this tile has an LDS slot
    for (int k = threadIdx.x; k < SEL_MAX_IDS; k += 256) {
        if (cnt[k]) {
            const int b = bucket_of_id[k];
            const size_t cell = (size_t)col * SEL_NB + b;
            basev[k] = off[b] + atomicAdd(&cur[b], cnt[k]);
            atomicMin(&bmin[cell], lo_id[k]);
            atomicMax(&bmax[cell], hi_id[k]);
        }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < SORT_ITEMS; ++i)
        if (rank[i] >= 0) dst[basev[id[i]] + (uint32_t)rank[i]] = keys[i];
    }
}

// the q-th smallest (0-based) key of an unordered segment whose smallest / largest keys are mn / mx, and the number of
// its keys <= that key; whole workgroup (256 threads), every thread returns the same values
// q-th smallest key (0-based) of an UNORDERED segment of more than 64 keys and the number of keys <= it
__device__ void sel_segment_select(const uint64_t *__restrict__ seg, int64_t len, int64_t q, uint64_t mn, uint64_t mx,
                                   uint64_t *tk_out, int64_t *le_out, uint32_t *s_hist, uint32_t *s_wsum, uint64_t *s_pick)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // radix selection over the bytes in which the segment's keys can differ, most significant first
    const int top = (63 - __clzll((long long)(mn ^ mx))) >> 3;
    uint64_t prefix = top < 7 ? (mn >> (8 * (top + 1))) << (8 * (top + 1)) : 0ull;
    int64_t below = 0, qrem = q, equal = len;
    for (int byte = top; byte >= 0 && equal > 1; --byte) {
        __syncthreads();
        s_hist[threadIdx.x] = 0;
        __syncthreads();
        const int hs = 8 * (byte + 1);
        for (int64_t i0 = threadIdx.x; i0 < len; i0 += 256 * 8) {        // eight loads in flight
            uint64_t v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) { const int64_t i = i0 + (int64_t)j * 256; v[j] = seg[i < len ? i : len - 1]; }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const bool in = i0 + (int64_t)j * 256 < len && (byte == 7 || (v[j] >> hs) == (prefix >> hs));
                if (in) atomicAdd(&s_hist[(uint32_t)(v[j] >> (8 * byte)) & 0xFF], 1u);
            }
        }
        __syncthreads();
        // digit d with cum(d - 1) <= qrem < cum(d): inclusive scan of the 256 counts, one per thread
        const uint32_t cntd = s_hist[threadIdx.x];
        uint32_t inc = cntd;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t y = __shfl_up(inc, off, 64);
            if (lane >= off) inc += y;
        }
        if (lane == 63) s_wsum[wave] = inc;
        __syncthreads();
        uint32_t before = inc - cntd;
        for (int w = 0; w < wave; ++w) before += s_wsum[w];
        if ((int64_t)before <= qrem && qrem < (int64_t)before + (int64_t)cntd) {   // exactly one thread
            s_pick[0] = (uint64_t)threadIdx.x;
            s_pick[1] = (uint64_t)before;
            s_pick[2] = (uint64_t)cntd;
        }
        __syncthreads();
        prefix |= s_pick[0] << (8 * byte);
        below += (int64_t)s_pick[1];
        qrem -= (int64_t)s_pick[1];
        equal = (int64_t)s_pick[2];
        if (equal == 1 && byte > 0) {
            // a single key carries the prefix: fetch it instead of resolving its remaining bytes one by one
            __syncthreads();
            const int hs2 = 8 * byte;
            for (int64_t i = threadIdx.x; i < len; i += 256) {
                const uint64_t v = seg[i];
                if ((v >> hs2) == (prefix >> hs2)) s_pick[3] = v;
            }
            __syncthreads();
            prefix = s_pick[3];
        }
    }
    *tk_out = prefix;
    *le_out = below + equal;
}

__global__ __launch_bounds__(256) void sel_walk2_kernel(int64_t n, double frac,
                                                        const uint32_t *__restrict__ cum, const uint16_t *__restrict__ mark,
                                                        const uint32_t *__restrict__ seg_off, const uint64_t *__restrict__ coll,
                                                        const unsigned long long *__restrict__ bmin,
                                                        const unsigned long long *__restrict__ bmax,
                                                        uint64_t *__restrict__ thr, uint16_t *__restrict__ thr_bucket,
                                                        int32_t *__restrict__ nbins, int32_t *__restrict__ fault)
{
    __shared__ uint32_t C[SEL_NB];
    __shared__ uint32_t SO[SEL_NB];
    __shared__ uint16_t MK[SEL_NB];
    __shared__ uint32_t s_hist[RADIX];
    __shared__ uint32_t s_wsum[4];
    __shared__ uint64_t s_pick[4];
    const int col = blockIdx.x;
    constexpr int nb_buckets = SEL_NB;
    for (int b = threadIdx.x; b < nb_buckets; b += 256) {
        C[b] = cum[(size_t)col * SEL_NB + b];
        SO[b] = seg_off[(size_t)col * SEL_NB + b];
        MK[b] = mark[(size_t)col * SEL_NB + b];
    }
    __syncthreads();
    const uint64_t *segs = coll + (size_t)col * n;
    uint64_t *t = thr + (size_t)col * GRX_MAX_BINS;
    const int lane = threadIdx.x & 63;
    int64_t done = 0;
    int nb = 0;
    while (done < n && nb < GRX_MAX_BINS) {
        int64_t size = (int64_t)(frac * (double)(n - done));
        if (size < 1) size = 1;
        const int64_t pos = done + size - 1;
        // first bucket whose inclusive prefix exceeds pos: two 64-ary ballot steps over the 4096 prefixes
        int j;
        {
            const int coarse = lane * 64 + 63;
            const bool gt = coarse < nb_buckets ? (int64_t)C[coarse] > pos : true;
            const int blk = __ffsll((long long)__ballot(gt)) - 1;
            const int fine = blk * 64 + lane;
            const bool gt2 = fine < nb_buckets ? (int64_t)C[fine] > pos : true;
            j = blk * 64 + __ffsll((long long)__ballot(gt2)) - 1;
        }
        const int64_t before = j ? (int64_t)C[j - 1] : 0;
        const int64_t len = (int64_t)C[j] - before;
        const int64_t q = pos - before;
        const size_t cell = (size_t)col * SEL_NB + j;
        const uint16_t mk = MK[j];
        if (!mk) {                                              // cannot happen: the interval walk covers every exact walk
            if (threadIdx.x == 0) atomicAdd(fault, 1);
            break;
        }
        // ONE trip to memory serves the common cases: the bucket's smallest and largest key, and 64 keys of its segment
        // -- from rank q on when the segment is sorted, all of it when it holds no more than 64
        const bool ties = mk == SEL_MARK_TIES;
        const bool sorted = !ties && (mk & SEL_MARK_SORTED);
        const uint64_t *seg = segs + SO[j];
        const int64_t at = sorted ? q + lane : lane;
        const uint64_t mine = ties ? 0ull : seg[at < len ? at : len - 1];
        const uint64_t mn = bmin[cell], mx = bmax[cell];
        uint64_t tk;
        int64_t le;
        if (mn == mx) {                                         // a block of ties
            tk = mn;
            le = len;
        } else if (sorted) {
            tk = __shfl(mine, 0, 64);
            int64_t base = q;
            uint64_t diff = __ballot(mine != tk || at >= len);
            while (diff == 0) {                                 // a run of ties longer than a wavefront: keep scanning
                base += 64;
                const int64_t i = base + lane;
                diff = __ballot(i >= len || seg[i < len ? i : len - 1] != tk);
            }
            le = base + __ffsll((long long)diff) - 1;
        } else if (len <= 64) {
            const uint64_t key = lane < len ? mine : ~0ull;
            int lt = 0, lq = 0;
            for (int i = 0; i < (int)len; ++i) {
                const uint64_t other = __shfl(key, i, 64);
                lt += other < key;
                lq += other <= key;
            }
            const uint64_t hit = __ballot(lane < len && lt <= q && q < lq);
            const int src = __ffsll((long long)hit) - 1;
            tk = __shfl(key, src, 64);
            le = __shfl(lq, src, 64);
        } else {
            sel_segment_select(seg, len, q, mn, mx, &tk, &le, s_hist, s_wsum, s_pick);
        }
        if (threadIdx.x == 0) {
            t[nb] = tk;
            // bucket of the threshold; bit 15: the bucket holds other keys too, so its keys are compared with the thresholds
            if (thr_bucket) thr_bucket[(size_t)col * GRX_MAX_BINS + nb] = (uint16_t)(j | (mn != mx ? SEL_THR_EXACT : 0));
        }
        ++nb;
        done = before + le;
    }
    if (threadIdx.x == 0) nbins[col] = (done < n) ? -nb : nb;
}

// Labels from the stored bucket ids (columns of sel_bid_min_n() = 2.5 M rows and more: round 3 measured the three streaming
// passes of config 5 at 7.3 -> 6.3 ms with it, and the 1 M-row columns of BASELINE's headline graph 0.05 ms SLOWER --
// hence the switch on the height).  The bucket map is monotone, so every key of a bucket WITHOUT a threshold lies
// between the same two thresholds -- its bin is the number of thresholds in earlier buckets -- and so does every key
// of a bucket that is one block of ties.  Only the keys of the buckets that hold a threshold among other keys (~20 of
// 4096) are read and compared.  2 + 1 bytes per key instead of 8 + 1.
__global__ __launch_bounds__(256) void sel_assign_kernel(const double *__restrict__ cols, int64_t ld, int64_t n,
                                                         const uint16_t *__restrict__ bid,
                                                         const uint64_t *__restrict__ thr,
                                                         const uint16_t *__restrict__ thr_bucket,
                                                         const int32_t *__restrict__ nbins,
                                                         uint8_t *__restrict__ bins, int64_t ld_bins, ColFlags flags,
                                                         const int32_t *__restrict__ fault, int32_t *__restrict__ status)
{
    __shared__ uint64_t t[GRX_MAX_BINS];
    __shared__ uint16_t tb[GRX_MAX_BINS];
    __shared__ uint8_t lut[SEL_NB];                              // bin of the bucket | 0x80: compare exactly
    const int col = blockIdx.y;
    const bool i64 = col_is_i64(flags, col);
    int nb = nbins[col];
    if (status && blockIdx.x == 0 && threadIdx.x == 0) {         // (as bin_assign_kernel)
        if (col == 0 && fault && *fault != 0) status[0] = 1;
        if (nb < 0) status[1] = 1;
    }
    const bool saturated = nb < 0;                               // more than GRX_MAX_BINS bins: the walk stopped early
    if (saturated) nb = GRX_MAX_BINS;
    if (threadIdx.x < GRX_MAX_BINS) {
        t[threadIdx.x] = (threadIdx.x < nb) ? thr[(size_t)col * GRX_MAX_BINS + threadIdx.x] : 0ull;
        tb[threadIdx.x] = (threadIdx.x < nb) ? thr_bucket[(size_t)col * GRX_MAX_BINS + threadIdx.x] : (uint16_t)0x7FFF;
    }
    __syncthreads();
    for (int b = threadIdx.x; b < SEL_NB; b += 256) {
        int lo = 0, hi = nb;                                     // thresholds in earlier buckets (ascending with the index)
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if ((tb[mid] & 0x7FFF) < b) lo = mid + 1; else hi = mid;
        }
        bool exact = saturated;
        for (int k = lo; k < nb && (tb[k] & 0x7FFF) == b; ++k) exact |= (tb[k] & SEL_THR_EXACT) != 0;
        lut[b] = (uint8_t)(lo | (exact ? 0x80 : 0));
    }
    __syncthreads();
    const double *x = cols + (size_t)col * ld;
    const uint16_t *bx = bid + (size_t)col * sel_bid_stride(n);
    uint8_t *o = bins + (size_t)col * ld_bins;
    const bool word_stores = (reinterpret_cast<uintptr_t>(o) & 3) == 0;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x * 4;
    for (int64_t i0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; i0 < n; i0 += stride) {
        // four consecutive keys per thread: one 8-byte load of bucket ids, one 4-byte store of labels
        uint16_t b4[4];
        if (i0 + 3 < n) {
            const uint2 raw = *reinterpret_cast<const uint2 *>(bx + i0);
            b4[0] = (uint16_t)(raw.x & 0xFFFF); b4[1] = (uint16_t)(raw.x >> 16);
            b4[2] = (uint16_t)(raw.y & 0xFFFF); b4[3] = (uint16_t)(raw.y >> 16);
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) b4[j] = i0 + j < n ? bx[i0 + j] : (uint16_t)0;
        }
        uint32_t packed = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            uint32_t lab = lut[b4[j]];
            if ((lab & 0x80) && i0 + j < n) {
                const uint64_t v = value_key(x[i0 + j], i64);
                int lo = 0, hi = nb;                             // first threshold >= v
                while (lo < hi) {
                    const int mid = (lo + hi) >> 1;
                    if (t[mid] < v) lo = mid + 1; else hi = mid;
                }
                lab = (uint32_t)lo;
            }
            packed |= (lab & 0xFF) << (8 * j);
        }
        if (i0 + 3 < n && word_stores) {
            *reinterpret_cast<uint32_t *>(o + i0) = packed;
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (i0 + j < n) o[i0 + j] = (uint8_t)(packed >> (8 * j));
        }
    }
}

struct SortPlan {
    int ntiles;
    size_t keys_bytes;      // one key buffer: ncols * n * 8
    size_t hist_bytes;      // ncols * RADIX * ntiles * 4
};

SortPlan make_plan(int64_t n, int ncols)
{
    SortPlan p;
    p.ntiles = (int)grx_ceil_div(n, SORT_TILE);
    p.keys_bytes = grx_align_up((size_t)ncols * (size_t)n * 8, 256);
    // per-tile counters + digit totals + digit bases
    p.hist_bytes = grx_align_up((size_t)ncols * RADIX * (size_t)p.ntiles * 4, 256) +
                   2 * grx_align_up((size_t)ncols * RADIX * 4, 256);
    return p;
}

// sort ncols columns; keysA/hist are scratch; result (fp64 ascending) in out (column stride out_ld)
int sort_columns(int64_t n, int ncols, const double *cols, int64_t ld, double *out, int64_t out_ld,
                 uint64_t *keysA, uint32_t *hist, hipStream_t st, SkipCtl ctl = SkipCtl{}, bool raw_u64 = false)
{
    const SortPlan p = make_plan(n, ncols);
    const bool skipping = ctl.flags != nullptr;             // then out is a key buffer with stride n
    const dim3 grid(p.ntiles, ncols);
    uint32_t *tot = hist + grx_align_up((size_t)ncols * RADIX * (size_t)p.ntiles * 4, 256) / 4;
    for (int pass = 0; pass < 8; ++pass) {
        const int shift = 8 * pass;
        // ping-pong: pass 0 cols->A, odd A->out, even out->A; pass 7 writes fp64 into out
        const void *src;
        int64_t sld;
        void *dst;
        int64_t dld;
        if (pass == 0) { src = cols; sld = ld; }
        else if (pass & 1) { src = keysA; sld = n; }
        else { src = out; sld = out_ld; }
        if (pass & 1) { dst = out; dld = out_ld; }
        else { dst = keysA; dld = n; }
        {
            GRX_PROF(GRX_K_SORT_COUNT, st);
            if (pass == 0 && !raw_u64) tile_count_kernel<true><<<grid, SORT_THREADS, 0, st>>>(src, sld, n, shift, p.ntiles, hist, ctl);
            else tile_count_kernel<false><<<grid, SORT_THREADS, 0, st>>>(src, sld, n, shift, p.ntiles, hist, ctl);
        }
        GRX_LAUNCH_CHECK();
        {
            GRX_PROF(GRX_K_SORT_SCAN, st);
            scan_rows_kernel<<<dim3(RADIX, ncols), 64, 0, st>>>(hist, p.ntiles, tot, ctl, pass);
        }
        GRX_LAUNCH_CHECK();
        {
            GRX_PROF(GRX_K_SORT_SCATTER, st);
            if (pass == 0 && !raw_u64) scatter_kernel<true, false><<<grid, SORT_THREADS, 0, st>>>(src, sld, dst, dld, n, shift, p.ntiles, hist, tot, ctl);
            else if (pass == 7 && !skipping && !raw_u64) scatter_kernel<false, true><<<grid, SORT_THREADS, 0, st>>>(src, sld, dst, dld, n, shift, p.ntiles, hist, tot, ctl);
            else scatter_kernel<false, false><<<grid, SORT_THREADS, 0, st>>>(src, sld, dst, dld, n, shift, p.ntiles, hist, tot, ctl);
        }
        GRX_LAUNCH_CHECK();
    }
    return GRX_OK;
}

}  // namespace

// internal entry for other translation units (grx_quant.hip): workspace laid out as in
// grx_sort_columns (key buffer, then counters)
int grx_internal_sort_columns(int64_t n, int ncols, const double *cols, int64_t ld, double *out, int64_t out_ld,
                              void *workspace, hipStream_t st)
{
    const SortPlan p = make_plan(n, ncols);
    char *ws = reinterpret_cast<char *>(workspace);
    return sort_columns(n, ncols, cols, ld, out, out_ld, reinterpret_cast<uint64_t *>(ws),
                        reinterpret_cast<uint32_t *>(ws + p.keys_bytes), st);
}

// one fp64 column -> ascending values in `out` and, in `perm`, the index each sorted position came from (stable).
// workspace: grx_sort_pairs_workspace_bytes(n) = key buffer, payload buffer, counters
int grx_internal_sort_pairs(int64_t n, const double *col, double *out, uint32_t *perm, void *workspace, hipStream_t st)
{
    if (n <= 0) return GRX_OK;
    const SortPlan p = make_plan(n, 1);
    char *ws = reinterpret_cast<char *>(workspace);
    uint64_t *keysA = reinterpret_cast<uint64_t *>(ws);
    uint32_t *payA = reinterpret_cast<uint32_t *>(ws + p.keys_bytes);
    uint32_t *hist = reinterpret_cast<uint32_t *>(ws + p.keys_bytes + grx_align_up((size_t)n * 4, 256));
    uint32_t *tot = hist + grx_align_up((size_t)RADIX * (size_t)p.ntiles * 4, 256) / 4;
    const dim3 grid(p.ntiles, 1);
    for (int pass = 0; pass < 8; ++pass) {
        const int shift = 8 * pass;
        const void *src = pass == 0 ? (const void *)col : ((pass & 1) ? (const void *)keysA : (const void *)out);
        void *dst = (pass & 1) ? (void *)out : (void *)keysA;
        const uint32_t *psrc = pass == 0 ? nullptr : ((pass & 1) ? payA : perm);
        uint32_t *pdst = (pass & 1) ? perm : payA;
        {
            GRX_PROF(GRX_K_SORT_COUNT, st);
            if (pass == 0) tile_count_kernel<true><<<grid, SORT_THREADS, 0, st>>>(src, n, n, shift, p.ntiles, hist, SkipCtl{});
            else tile_count_kernel<false><<<grid, SORT_THREADS, 0, st>>>(src, n, n, shift, p.ntiles, hist, SkipCtl{});
        }
        GRX_LAUNCH_CHECK();
        {
            GRX_PROF(GRX_K_SORT_SCAN, st);
            scan_rows_kernel<<<dim3(RADIX, 1), 64, 0, st>>>(hist, p.ntiles, tot, SkipCtl{}, pass);
        }
        GRX_LAUNCH_CHECK();
        {
            GRX_PROF(GRX_K_SORT_SCATTER, st);
            if (pass == 0) scatter_pairs_kernel<true, false><<<grid, SORT_THREADS, 0, st>>>(src, dst, psrc, pdst, n, shift, p.ntiles, hist, tot);
            else if (pass == 7) scatter_pairs_kernel<false, true><<<grid, SORT_THREADS, 0, st>>>(src, dst, psrc, pdst, n, shift, p.ntiles, hist, tot);
            else scatter_pairs_kernel<false, false><<<grid, SORT_THREADS, 0, st>>>(src, dst, psrc, pdst, n, shift, p.ntiles, hist, tot);
        }
        GRX_LAUNCH_CHECK();
    }
    return GRX_OK;
}

size_t grx_internal_sort_pairs_workspace_bytes(int64_t n)
{
    if (n <= 0) return 256;
    const SortPlan p = make_plan(n, 1);
    return p.keys_bytes + grx_align_up((size_t)n * 4, 256) + p.hist_bytes;
}

// raw 64-bit keys (graph ingest: (row, column) / (row, edge sequence) pairs), ascending; same workspace
int grx_internal_sort_u64(int64_t n, const uint64_t *keys, uint64_t *out, void *workspace, hipStream_t st)
{
    if (n <= 0) return GRX_OK;
    const SortPlan p = make_plan(n, 1);
    char *ws = reinterpret_cast<char *>(workspace);
    return sort_columns(n, 1, reinterpret_cast<const double *>(keys), n, reinterpret_cast<double *>(out), n,
                        reinterpret_cast<uint64_t *>(ws), reinterpret_cast<uint32_t *>(ws + p.keys_bytes), st, SkipCtl{}, true);
}

extern "C" {

size_t grx_sort_workspace_bytes(int64_t n, int ncols)
{
    if (n <= 0 || ncols <= 0) return 256;
    const SortPlan p = make_plan(n, ncols);
    return p.keys_bytes + p.hist_bytes;
}

namespace {
struct SelLayout { size_t maps, luts, tieb, tiev, hist, cum, seg_off, cursor, bmax, tbroken, bmin, mark, idlist, nids, coll, bid, thr, thrb, nbins, fault, total; };
SelLayout sel_layout(int64_t n, int ncols)
{
    SelLayout L;
    size_t o = 0;
    auto take = [&](size_t bytes) { const size_t at = o; o += grx_align_up(bytes, 256); return at; };
    L.maps = take((size_t)ncols * sizeof(SelMap));
    L.luts = take((size_t)ncols * SEL_NB * 4);
    L.tieb = take((size_t)ncols * SEL_NB * 2);
    L.tiev = take((size_t)ncols * (SEL_MAX_TIES + 1) * 8);
    L.hist = take((size_t)ncols * SEL_NB * 4);                  // hist, cursor, bmax (zeroed together)
    L.cursor = take((size_t)ncols * SEL_NB * 4);
    L.bmax = take((size_t)ncols * SEL_NB * 8);
    L.tbroken = take((size_t)ncols * SEL_NB);                   // (zeroed with the three before it)
    L.bmin = take((size_t)ncols * SEL_NB * 8);                  // all ones
    L.cum = take((size_t)ncols * SEL_NB * 4);
    L.seg_off = take((size_t)ncols * SEL_NB * 4);
    L.mark = take((size_t)ncols * SEL_NB * 2);
    L.idlist = take((size_t)ncols * SEL_MAX_IDS * 2);
    L.nids = take((size_t)ncols * 4);
    L.coll = take((size_t)ncols * (size_t)n * 8);
    L.bid = take(n >= sel_bid_min_n() ? (size_t)ncols * (size_t)sel_bid_stride(n) * 2 : 0);
    L.thr = take((size_t)ncols * GRX_MAX_BINS * 8);
    L.thrb = take((size_t)ncols * GRX_MAX_BINS * 2);
    L.nbins = take((size_t)ncols * 4);
    L.fault = take(4);
    L.total = o;
    return L;
}
}  // namespace

size_t grx_log_bin_workspace_bytes(int64_t n, int ncols)
{
    if (n <= 0 || ncols <= 0) return 256;
    const SortPlan p = make_plan(n, ncols);
    // keysA + sorted + hist + thresholds + nbins + pass-skipping state (key bits, flags)
    const size_t sort_path = 2 * p.keys_bytes + p.hist_bytes + grx_align_up((size_t)ncols * GRX_MAX_BINS * 8, 256) +
                             grx_align_up((size_t)ncols * 4, 256) + grx_align_up((size_t)ncols * p.ntiles * 32, 256) +
                             grx_align_up((size_t)ncols * sizeof(BinPlanCol), 256);
    const size_t select_path = sel_layout(n, ncols).total;
    return sort_path > select_path ? sort_path : select_path;
}

int grx_sort_columns(int64_t n, int ncols, const double *d_cols, int64_t ld, double *d_sorted,
                     int64_t ld_sorted, void *d_workspace, size_t workspace_bytes, void *stream)
{
    GRX_REQUIRE(n >= 0 && ncols >= 0 && ld >= n && ld_sorted >= n, "grx_sort_columns: bad shape");
    GRX_REQUIRE(n < ((int64_t)1 << 31), "grx_sort_columns: n must be < 2^31");
    if (n == 0 || ncols == 0) return GRX_OK;
    GRX_REQUIRE(d_cols && d_sorted && d_workspace, "grx_sort_columns: NULL pointer");
    if (workspace_bytes < grx_sort_workspace_bytes(n, ncols)) {
        grx_set_error("grx_sort_columns: workspace %zu < %zu", workspace_bytes, grx_sort_workspace_bytes(n, ncols));
        return GRX_ERR_WORKSPACE;
    }
    const SortPlan p = make_plan(n, ncols);
    char *ws = reinterpret_cast<char *>(d_workspace);
    return sort_columns(n, ncols, d_cols, ld, d_sorted, ld_sorted, reinterpret_cast<uint64_t *>(ws),
                        reinterpret_cast<uint32_t *>(ws + p.keys_bytes), grx_stream(stream));
}

}  // extern "C"

// Internal (grx_refex.hip): grx_vertical_log_bin_typed with an explicit place for the outcome flags -- d_status[0] (the
// sort-free threshold walk met an unmarked bucket) and d_status[1] (a column needs more than GRX_MAX_BINS bins) are
// set from inside the call's last kernel: no launch of its own, no synchronisation, the caller reads the two words
// with whatever it copies back next.  (Round 4 passed the pointer through a thread-local that the next binning call of
// the thread consumed -- a hidden coupling between two calls; gone.)
int grx_internal_vertical_log_bin(int64_t n, int ncols, const double *d_cols, int64_t ld, const uint8_t *h_is_i64, double frac,
                                  uint8_t *d_bins, int64_t ld_bins, int32_t *d_nbins, void *d_workspace,
                                  size_t workspace_bytes, int32_t *d_status, void *stream);

extern "C" {

int grx_vertical_log_bin(int64_t n, int ncols, const double *d_cols, int64_t ld, double frac,
                         uint8_t *d_bins, int64_t ld_bins, int32_t *d_nbins, void *d_workspace,
                         size_t workspace_bytes, void *stream)
{
    return grx_vertical_log_bin_typed(n, ncols, d_cols, ld, nullptr, frac, d_bins, ld_bins, d_nbins, d_workspace,
                                      workspace_bytes, stream);
}

int grx_vertical_log_bin_typed(int64_t n, int ncols, const double *d_cols, int64_t ld, const uint8_t *h_is_i64, double frac,
                               uint8_t *d_bins, int64_t ld_bins, int32_t *d_nbins, void *d_workspace,
                               size_t workspace_bytes, void *stream)
{
    return grx_internal_vertical_log_bin(n, ncols, d_cols, ld, h_is_i64, frac, d_bins, ld_bins, d_nbins, d_workspace,
                                         workspace_bytes, nullptr, stream);
}

}  // extern "C"

int grx_internal_vertical_log_bin(int64_t n, int ncols, const double *d_cols, int64_t ld, const uint8_t *h_is_i64, double frac,
                                  uint8_t *d_bins, int64_t ld_bins, int32_t *d_nbins, void *d_workspace,
                                  size_t workspace_bytes, int32_t *status, void *stream)
{
    ColFlags flags;
    for (int j = 0; j < 8; ++j) flags.w[j] = 0;
    if (h_is_i64)
        for (int c = 0; c < ncols; ++c)
            if (h_is_i64[c]) {
                GRX_REQUIRE(c < MAX_TYPED_COLS, "grx_vertical_log_bin_typed: int64 columns beyond the first %d of a call", MAX_TYPED_COLS);
                flags.w[c >> 6] |= 1ull << (c & 63);
            }
    GRX_REQUIRE(frac > 0.0 && frac < 1.0, "must specify frac in interval (0, 1)");
    GRX_REQUIRE(n >= 0 && ncols >= 0 && ld >= n && ld_bins >= n, "grx_vertical_log_bin: bad shape");
    GRX_REQUIRE(n < ((int64_t)1 << 31), "grx_vertical_log_bin: n must be < 2^31");
    if (n == 0 || ncols == 0) return GRX_OK;
    GRX_REQUIRE(d_cols && d_bins && d_workspace, "grx_vertical_log_bin: NULL pointer");
    if (workspace_bytes < grx_log_bin_workspace_bytes(n, ncols)) {
        grx_set_error("grx_vertical_log_bin: workspace %zu < %zu", workspace_bytes, grx_log_bin_workspace_bytes(n, ncols));
        return GRX_ERR_WORKSPACE;
    }
    hipStream_t st = grx_stream(stream);
    const SortPlan p = make_plan(n, ncols);
    char *ws = reinterpret_cast<char *>(d_workspace);
    static const bool use_sort = [] { const char *e = std::getenv("GRX_BIN_SORT"); return e && *e == '1'; }();
    if (!use_sort) {
        // binning without sorting: sample range -> bucket histogram -> interval walk -> collect -> exact walk -> assign
        const SelLayout L = sel_layout(n, ncols);
        SelMap *maps = reinterpret_cast<SelMap *>(ws + L.maps);
        uint32_t *luts = reinterpret_cast<uint32_t *>(ws + L.luts);
        uint16_t *tieb = reinterpret_cast<uint16_t *>(ws + L.tieb);
        uint64_t *tiev = reinterpret_cast<uint64_t *>(ws + L.tiev);
        uint8_t *tbroken = reinterpret_cast<uint8_t *>(ws + L.tbroken);
        uint32_t *hist = reinterpret_cast<uint32_t *>(ws + L.hist);
        uint32_t *cursor = reinterpret_cast<uint32_t *>(ws + L.cursor);
        uint32_t *cum = reinterpret_cast<uint32_t *>(ws + L.cum);
        uint32_t *seg_off = reinterpret_cast<uint32_t *>(ws + L.seg_off);
        uint16_t *mark = reinterpret_cast<uint16_t *>(ws + L.mark);
        uint16_t *idlist = reinterpret_cast<uint16_t *>(ws + L.idlist);
        int32_t *nids = reinterpret_cast<int32_t *>(ws + L.nids);
        uint64_t *coll = reinterpret_cast<uint64_t *>(ws + L.coll);
        uint64_t *thr = reinterpret_cast<uint64_t *>(ws + L.thr);
        uint16_t *bid = reinterpret_cast<uint16_t *>(ws + L.bid);
        uint16_t *thrb = reinterpret_cast<uint16_t *>(ws + L.thrb);
        // tall columns keep their bucket ids: passes 2 and 3 read 2 bytes per key instead of 8
        const bool use_bid = n >= sel_bid_min_n();
        int32_t *nb_ws = reinterpret_cast<int32_t *>(ws + L.nbins);
        int32_t *fault = reinterpret_cast<int32_t *>(ws + L.fault);
        unsigned long long *bmin = reinterpret_cast<unsigned long long *>(ws + L.bmin);
        unsigned long long *bmax = reinterpret_cast<unsigned long long *>(ws + L.bmax);
        {
            GRX_PROF(GRX_K_SEL_MAP, st);
            sel_map_kernel<<<ncols, 1024, 0, st>>>(d_cols, ld, n, maps, luts, tieb, tiev, hist, cursor, bmin, bmax, tbroken, fault, flags);
        }
        {
            GRX_PROF(GRX_K_SEL_HIST, st);
            const int hist_tiles = use_bid ? 4 : 1;
            const dim3 hgrid((unsigned)grx_ceil_div(n, (int64_t)SEL_HIST_TILE * hist_tiles), ncols);
            if (use_bid) sel_hist_kernel<true><<<hgrid, 256, 0, st>>>(d_cols, ld, n, maps, luts, tieb, tiev, tbroken, hist, bid, flags, hist_tiles);
            else sel_hist_kernel<false><<<hgrid, 256, 0, st>>>(d_cols, ld, n, maps, luts, tieb, tiev, tbroken, hist, nullptr, flags, hist_tiles);
        }
        {
            GRX_PROF(GRX_K_SEL_WALK1, st);
            sel_walk1_kernel<<<ncols, 1024, 0, st>>>(n, frac, hist, tieb, tbroken, tiev, cum, mark, seg_off, idlist, nids, bmin, bmax);
        }
        {
            GRX_PROF(GRX_K_SEL_COLLECT, st);
            const int coll_tiles = use_bid ? 8 : 1;
            const dim3 cgrid((unsigned)grx_ceil_div(p.ntiles, coll_tiles), ncols);
            if (use_bid) sel_collect_kernel<true><<<cgrid, 256, 0, st>>>(d_cols, ld, n, bid, maps, luts, mark, seg_off, cursor, coll, bmin, bmax, flags, coll_tiles);
            else sel_collect_kernel<false><<<cgrid, 256, 0, st>>>(d_cols, ld, n, nullptr, maps, luts, mark, seg_off, cursor, coll, bmin, bmax, flags, coll_tiles);
        }
        {
            GRX_PROF(GRX_K_SEL_SEGSORT, st);
            sel_sort_kernel<<<dim3(64, ncols), 512, 0, st>>>(cum, seg_off, idlist, nids, bmin, bmax, n, coll, mark);
        }
        {
            GRX_PROF(GRX_K_SEL_WALK2, st);
            sel_walk2_kernel<<<ncols, 256, 0, st>>>(n, frac, cum, mark, seg_off, coll, bmin, bmax, thr, use_bid ? thrb : nullptr, nb_ws, fault);
        }
        GRX_LAUNCH_CHECK();
        const int64_t want = grx_ceil_div(n, 256 * 4);
        const dim3 grid((unsigned)(want > 2048 ? 2048 : want), ncols);
        {
            GRX_PROF(GRX_K_BIN_ASSIGN, st);
            // (sel_assign builds a 4096-entry bucket -> label table per workgroup: sixteen strides of keys per workgroup,
            // not two and a half, or the table costs more than the pass -- measured 2.6 ms against 1.8 at config 5)
            const int64_t want_a = grx_ceil_div(n, 256 * 4 * 16);
            const dim3 agrid((unsigned)(want_a > 2048 ? 2048 : want_a), ncols);
            if (use_bid) sel_assign_kernel<<<agrid, 256, 0, st>>>(d_cols, ld, n, bid, thr, thrb, nb_ws, d_bins, ld_bins, flags, fault, status);
            else bin_assign_kernel<<<grid, 256, 0, st>>>(d_cols, ld, n, thr, nb_ws, d_bins, ld_bins, flags, fault, status);
        }
        GRX_LAUNCH_CHECK();
        if (d_nbins)
            GRX_CHECK_HIP(hipMemcpyAsync(d_nbins, nb_ws, (size_t)ncols * 4, hipMemcpyDeviceToDevice, st));
        return GRX_OK;
    }
    uint64_t *buf_a = reinterpret_cast<uint64_t *>(ws);
    uint64_t *buf_b = reinterpret_cast<uint64_t *>(ws + p.keys_bytes);
    uint32_t *hist = reinterpret_cast<uint32_t *>(ws + 2 * p.keys_bytes);
    uint32_t *tot = hist + grx_align_up((size_t)ncols * RADIX * (size_t)p.ntiles * 4, 256) / 4;
    uint64_t *thr = reinterpret_cast<uint64_t *>(ws + 2 * p.keys_bytes + p.hist_bytes);
    int32_t *nb_ws = reinterpret_cast<int32_t *>(ws + 2 * p.keys_bytes + p.hist_bytes +
                                                 grx_align_up((size_t)ncols * GRX_MAX_BINS * 8, 256));
    char *plan_ws = reinterpret_cast<char *>(nb_ws) + grx_align_up((size_t)ncols * 4, 256);
    uint64_t *bits = reinterpret_cast<uint64_t *>(plan_ws);
    BinPlanCol *plan = reinterpret_cast<BinPlanCol *>(plan_ws + grx_align_up((size_t)ncols * p.ntiles * 32, 256));
    // which key bytes vary -> per-column plan (NARROW 32-bit keys / WIDE top-four-bytes sort)
    const int nbt = (int)grx_ceil_div(n, BITS_TILE);
    {
        GRX_PROF(GRX_K_KEY_BITS, st);
        key_bits_kernel<<<dim3(nbt, ncols), 256, 0, st>>>(d_cols, ld, n, nbt, bits, flags);
        bin_plan_kernel<<<ncols, 64, 0, st>>>(bits, nbt, plan);
    }
    GRX_LAUNCH_CHECK();
    const dim3 sgrid(p.ntiles, ncols);
    for (int round = 0; round < 4; ++round) {
        {
            GRX_PROF(GRX_K_SORT_COUNT, st);
            tile_count2_kernel<<<sgrid, SORT_THREADS, 0, st>>>(d_cols, ld, n, round, p.ntiles, plan, buf_a, buf_b, hist, flags);
        }
        {
            GRX_PROF(GRX_K_SORT_SCAN, st);
            scan_rows2_kernel<<<dim3(RADIX, ncols), 64, 0, st>>>(hist, p.ntiles, tot, plan, round);
        }
        {
            GRX_PROF(GRX_K_SORT_SCATTER, st);
            scatter2_kernel<<<sgrid, SORT_THREADS, 0, st>>>(d_cols, ld, n, round, p.ntiles, plan, buf_a, buf_b, hist, tot, flags);
        }
        GRX_LAUNCH_CHECK();
    }
    { GRX_PROF(GRX_K_BIN_THRESHOLD, st);
    bin_threshold2_kernel<<<ncols, 256, 0, st>>>(d_cols, ld, n, frac, plan, buf_a, buf_b, thr, nb_ws, flags);
    }
    GRX_LAUNCH_CHECK();
    const int64_t want = grx_ceil_div(n, 256 * 4);
    const dim3 grid((unsigned)(want > 2048 ? 2048 : want), ncols);
    { GRX_PROF(GRX_K_BIN_ASSIGN, st);
    bin_assign_kernel<<<grid, 256, 0, st>>>(d_cols, ld, n, thr, nb_ws, d_bins, ld_bins, flags, nullptr, status);
    }
    GRX_LAUNCH_CHECK();
    if (d_nbins)
        GRX_CHECK_HIP(hipMemcpyAsync(d_nbins, nb_ws, (size_t)ncols * 4, hipMemcpyDeviceToDevice, st));
    return GRX_OK;
}

extern "C" {

int grx_chebyshev(int64_t row_begin, int64_t row_end, int F, int first_new,
                  const uint8_t *const *h_bin_ptrs, int32_t *d_dist, int cap, void *stream)
{
    GRX_REQUIRE(row_begin >= 0 && row_begin <= row_end, "grx_chebyshev: bad row range");
    GRX_REQUIRE(F >= 0 && first_new >= 0, "grx_chebyshev: bad F/first_new");
    if (F < 2 || first_new >= F || row_end == row_begin) return GRX_OK;
    GRX_REQUIRE(h_bin_ptrs && d_dist, "grx_chebyshev: NULL pointer");
    hipStream_t st = grx_stream(stream);
    if (cap < 0 || cap > 254) cap = 254;                                   // bin labels are < 128: 254 = everything exact
    const int64_t tiles = grx_ceil_div(row_end - row_begin, CH_ROWS);
    // a strip of consecutive tiles per workgroup: a pair that exceeded the cap in the first tile of
    // the strip costs nothing in the others.  Two workgroups per CU keep the chip busy.
    const int64_t max_grid = GRX_NUM_CU * 4;
    const int tiles_per_block = (int)(tiles <= max_grid ? 1 : grx_ceil_div(tiles, max_grid));
    const int grid = (int)grx_ceil_div(tiles, tiles_per_block);
    auto lds_bytes = [](int Fl, int fn) {
        const int q0l = fn > 1 ? fn : 1;
        const int np = Fl * (Fl - 1) / 2 - q0l * (q0l - 1) / 2;
        return (size_t)Fl * CH_STRIDE + (size_t)(np > 0 ? np : 0) * 3 + 16;
    };
    // one column set (<= CH_MAX_F columns: the global columns [a0, a0+na) then [b0, ...)): pairs with
    // q >= fn.  Two stages when a cap is given and the range is long: all pairs on a sample of sixteen tiles SPREAD
    // over the range (round 5; rounds 2 - 4 took the first rows -- with the degree-descending node order the hubs,
    // where the features of a power-law graph differ most; on config 5 the sums of every generation grow with the
    // degree together, their bins agree on the hubs, and thousands of pairs survived to die on the other rows), then
    // the whole range for the pairs that are still within the cap.
    constexpr int SAMPLE_TILES = 16;
    auto run = [&](int Fl, int fn, const GrxPtrTable &tab, int a0, int na, int b0) -> int {
        GRX_PROF(GRX_K_CHEBYSHEV, st);
        const size_t lds = lds_bytes(Fl, fn);
        if (cap < 254 && tiles > 8 * SAMPLE_TILES && Fl > 4) {
            const int64_t sample_stride = tiles / SAMPLE_TILES;               // >= 8
            const int q0l = fn > 1 ? fn : 1;
            const int np = Fl * (Fl - 1) / 2 - q0l * (q0l - 1) / 2;
            const int psplit = np >= 256 ? 16 : (np >= 32 ? 4 : 1);        // the sample stage's pairs over 16 x psplit workgroups
            chebyshev_kernel<<<SAMPLE_TILES * psplit, 256, lds, st>>>(row_begin, row_end, Fl, fn, tab, d_dist, F, a0, na, b0, cap, 1,
                                                                      0, psplit, sample_stride);
            // (the sampled tiles are simply visited again: sixteen of thousands)
            const int tpb = (int)(tiles <= GRX_NUM_CU * 8 ? 1 : grx_ceil_div(tiles, GRX_NUM_CU * 8));
            chebyshev_kernel<<<(int)grx_ceil_div(tiles, tpb), 256, lds, st>>>(row_begin, row_end, Fl, fn, tab, d_dist, F, a0, na,
                                                                              b0, cap, tpb, 1, 1, 1);
        } else {
            chebyshev_kernel<<<grid, 256, lds, st>>>(row_begin, row_end, Fl, fn, tab, d_dist, F, a0, na, b0, cap,
                                                     tiles_per_block, 0, 1, 1);
        }
        GRX_LAUNCH_CHECK();
        return GRX_OK;
    };
    if (F <= CH_MAX_F) {
        GrxPtrTable tab;
        for (int c = 0; c < F; ++c) tab.p[c] = h_bin_ptrs[c];
        return run(F, first_new, tab, 0, F, 0);
    }
    // more columns than one LDS tile holds: column groups of CH_MAX_F/2, one column set per pair of
    // groups (A, A): all pairs inside A;  (A, B), A < B: the pairs between A and B (q in B, p < q;
    // the pairs inside B are recomputed, harmless under atomicMax)
    constexpr int GROUP = CH_MAX_F / 2;
    const int ngroups = (F + GROUP - 1) / GROUP;
    for (int A = 0; A < ngroups; ++A) {
        const int a0 = A * GROUP, na = (F - a0 < GROUP) ? F - a0 : GROUP;
        for (int B = A; B < ngroups; ++B) {
            const int b0 = B * GROUP, nb = (B == A) ? 0 : ((F - b0 < GROUP) ? F - b0 : GROUP);
            GrxPtrTable tab;
            for (int c = 0; c < na; ++c) tab.p[c] = h_bin_ptrs[a0 + c];
            for (int c = 0; c < nb; ++c) tab.p[na + c] = h_bin_ptrs[b0 + c];
            if (na + nb < 2) continue;
            int rc = run(na + nb, (B == A) ? 0 : na, tab, a0, na, b0);
            if (rc != GRX_OK) return rc;
        }
    }
    return GRX_OK;
}

}  // extern "C"
