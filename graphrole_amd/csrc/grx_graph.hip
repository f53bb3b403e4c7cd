// grx_graph.hip -- generation-0 features and the ReFeX neighbour aggregation on a CSR graph.
//
// Kernels (all HBM/L2-gather bound; no MFMA -- this is integer-indexed fp64 streaming work):
//   row_sums_kernel        weighted degree            networkx.py:48-63
//   egonet_group_kernel / egonet_big_kernel   ego-net internal/external  networkx.py:71-83,115-123
//   pack_rows_kernel       column-major -> row-major gather source
//   aggregate_kernel       sum / mean over neighbours  features/extract.py:98-119
//   aggregate_combine_kernel                             rows with > 128 neighbours: block sums -> row sums
//   aggregate_minmax_kernel   min / max over neighbours
//
// Determinism: the neighbour sums follow numpy's pairwise-summation tree (a function of the row
// length only), so they are bitwise equal to the reference's Series.sum() for any launch
// geometry; the other reductions are "per-lane sequential, then a fixed butterfly".
#include "grx_common.h"

#include <cstdlib>

// Streams that are read exactly once per launch (the neighbour index list, the oriented arc tables, the output
// columns) can carry the non-temporal hint so that they do not evict the hot rows / hub lists the random gathers of
// the same kernel hit in L2.  MEASURED WITHOUT GAIN on MI355X (round 3, -DGRX_NT_STREAMS=1 against 0: aggregation
// 1.011 vs 0.995 ms per step at BA 1 M, 12.95 vs 12.33 ms at config 5, triangle counting 0.588 vs 0.585): the L2 hit
// rate of these kernels is set by how many gather rows fit (sqrt(C / N) on a power-law graph), not by what the
// streams displace.  Off by default; the macro stays for the next experiment.
#ifndef GRX_NT_STREAMS
#define GRX_NT_STREAMS 0
#endif
#if GRX_NT_STREAMS
#define GRX_STREAM_LD(p) __builtin_nontemporal_load(&(p))
#define GRX_STREAM_ST(p, v) __builtin_nontemporal_store((v), &(p))
#else
#define GRX_STREAM_LD(p) (p)
#define GRX_STREAM_ST(p, v) ((p) = (v))
#endif

#include <vector>

namespace {

// ---------------------------------------------------------------------------------------
// small device helpers
// ---------------------------------------------------------------------------------------
// position of key in the ascending slice col[b,e), or -1
__device__ __forceinline__ int64_t find_in_row(const int32_t *__restrict__ col, int64_t b,
                                               int64_t e, int32_t key)
{
    while (b < e) {
        int64_t mid = (b + e) >> 1;
        int32_t c = col[mid];
        if (c < key) b = mid + 1;
        else if (c > key) e = mid;
        else return mid;
    }
    return -1;
}

// first position in col[b,e) with col[pos] >= key
__device__ __forceinline__ int64_t lower_bound_row(const int32_t *__restrict__ col, int64_t b,
                                                   int64_t e, int32_t key)
{
    while (b < e) {
        int64_t mid = (b + e) >> 1;
        if (col[mid] < key) b = mid + 1;
        else e = mid;
    }
    return b;
}

__device__ __forceinline__ int ilog2_i64(int64_t x) { return 63 - __clzll((unsigned long long)(x | 1)); }

// ---------------------------------------------------------------------------------------
// weighted row sums
// ---------------------------------------------------------------------------------------
// G lanes per row, R rows per group in flight (rows v, v + ngroups, ...): the dependent row_ptr -> weights chain
// of one short row leaves the memory system idle, R independent chains keep it busy.  Per row the additions run
// in the same order whatever R is: lane-strided partial sums, then the butterfly.
template <int G, int R>
__global__ __launch_bounds__(256) void row_sums_kernel(const int64_t *__restrict__ row_ptr,
                                                       const int32_t *__restrict__ col,
                                                       const double *__restrict__ w, int add_loop,
                                                       int64_t row_begin, int64_t row_end,
                                                       double *__restrict__ out)
{
    const int lane = threadIdx.x % G;
    const int64_t group = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / G;
    const int64_t ngroups = (int64_t)gridDim.x * blockDim.x / G;
    for (int64_t v0 = row_begin + group; v0 < row_end; v0 += ngroups * R) {
        int64_t b[R], e[R];
        int64_t longest = 0;
#pragma unroll
        for (int i = 0; i < R; ++i) {
            const int64_t v = v0 + ngroups * i;
            const bool live = v < row_end;
            b[i] = live ? row_ptr[v] : 0;
            e[i] = live ? row_ptr[v + 1] : 0;
            longest = e[i] - b[i] > longest ? e[i] - b[i] : longest;
        }
        if (w == nullptr) {                                 // implicit weight 1: degree from row_ptr
            if (lane == 0) {
#pragma unroll
                for (int i = 0; i < R; ++i) {
                    const int64_t v = v0 + ngroups * i;
                    if (v < row_end)
                        out[v] = (double)((e[i] - b[i]) + ((add_loop && find_in_row(col, b[i], e[i], (int32_t)v) >= 0) ? 1 : 0));
                }
            }
            continue;
        }
        double s[R], loop[R];
#pragma unroll
        for (int i = 0; i < R; ++i) s[i] = loop[i] = 0.0;
        for (int64_t off = lane; off < longest; off += G) {
#pragma unroll
            for (int i = 0; i < R; ++i) {
                const int64_t k = b[i] + off;
                if (k < e[i]) {
                    const double x = w[k];
                    s[i] += x;
                    if (add_loop && col[k] == v0 + ngroups * i) loop[i] = x;
                }
            }
        }
#pragma unroll
        for (int i = 0; i < R; ++i) {
            const double si = grx_group_sum<G>(s[i]);
            const double li = grx_group_sum<G>(loop[i]);
            const int64_t v = v0 + ngroups * i;
            if (lane == 0 && v < row_end) out[v] = si + li;
        }
    }
}

// ---------------------------------------------------------------------------------------
// ego-net features
// ---------------------------------------------------------------------------------------
// One node per TPN threads.  Each lane owns ego members m = lane, lane+TPN, ... and for member a
// picks the cheaper of
//   S1: walk row(a), test membership of each entry in ego(v)        cost deg(a) * log deg(v)
//   S2: walk ego(v), look each member up in row(a)                   cost deg(v) * log deg(a)
// S2 obtains the boundary weight of a as rowsum(a) - matched weight; when every entry of row(a)
// matched, the boundary contribution is exactly 0 (no cancellation residue).
// The same features for nodes with at most EGO_GROUP_MAX out-neighbours (almost every node of a
// sparse graph): EIGHT lanes per node instead of a wavefront.  The ego set sits in registers (three
// ids per lane); the members are visited one after the other and their rows are either scanned in
// coalesced chunks of eight arcs, membership by the all-pairs shuffle compare of the triangle
// kernel, or -- long rows, i.e. hub neighbours -- probed by binary search for the (at most 25) ego
// members.  Per lane sequential sums, fixed butterfly: bitwise reproducible.
constexpr int EGO_SLOTS = 4;                                     // ids per lane: nodes with at most 32 out-neighbours
constexpr int EGO_GROUP_MAX = 8 * EGO_SLOTS;
constexpr int EGO_SLOTS_WIDE = 8;                                // a second instance of the kernel: 33 .. 64 out-neighbours
constexpr int EGO_GROUP_MAX_WIDE = 8 * EGO_SLOTS_WIDE;

// Round 5.  What a member a of an ego set contributes needs the ids of row(a), where it begins (weights of matched
// arcs), its length and its weighted row sum.  Read from the CSR that is two row_ptr entries, rowsum[a] and an
// unaligned run of ids: 3 - 4 cache-line requests per (v, a) pair, and the REQUEST rate (~50 G/s beyond the caches,
// profiles/r04_gather_bw.json) is what bounds this kernel.  A streaming pre-pass therefore lays every row out as one
// aligned 128-byte SLOT -- row sum, begin | min(length, SAT) << 40, the first 28 ids (-1 padded) -- so that a pair
// costs ONE aligned request that eight lanes read with one 16-byte load each; rows longer than 28 ids continue in the CSR.
constexpr int EGO_DEG_SHIFT = 40;
constexpr unsigned long long EGO_BEGIN_MASK = (1ull << EGO_DEG_SHIFT) - 1;
constexpr unsigned EGO_DEG_SAT = (1u << 24) - 1;                 // longer rows: length from row_ptr
constexpr int EGO_SLOT_IDS = 28;
struct __align__(16) EgoSlot {
    double rs;
    unsigned long long bd;
    int32_t ids[EGO_SLOT_IDS];
};
static_assert(sizeof(EgoSlot) == 128, "one slot = one 128-byte line");

// slots of all n rows + the rows of [row_begin, row_end) beyond the 8-lane kernel's 32 neighbours, appended to lists by
// out-degree: 33 .. 64 (the wide instance of the group kernel), 65 .. hub - 1 (a wavefront per row), hub and more: a
// workgroup per PART of EGO_PART members (entries {row, part, parts}: a node with 10 000 neighbours is ten work items,
// not one workgroup that finishes long after the others).  The order of the lists does not matter: every row's result
// is computed independently of the others, the parts of a row are summed in part order by egonet_combine_kernel.
constexpr int EGO_PART = 1024;
__global__ __launch_bounds__(256) void egonet_prepare_kernel(
    int64_t n, const int64_t *__restrict__ row_ptr, const int32_t *__restrict__ col, const double *__restrict__ rowsum,
    int64_t row_begin, int64_t row_end, int64_t hub, EgoSlot *__restrict__ slots, int32_t *__restrict__ wide_rows,
    int32_t *__restrict__ mid_rows, int32_t *__restrict__ hub_parts, unsigned *__restrict__ counts)
{
    for (int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x; v < ((n + 63) & ~(int64_t)63); v += (int64_t)gridDim.x * 256) {
        const int64_t d = v < n ? row_ptr[v + 1] - row_ptr[v] : 0;
        const bool owned = v >= row_begin && v < row_end;
#pragma unroll
        for (int which = 0; which < 2; ++which) {
            const bool take = owned && (which == 0 ? (d > EGO_GROUP_MAX && d <= EGO_GROUP_MAX_WIDE) : (d > EGO_GROUP_MAX_WIDE && d < hub));
            const unsigned long long bal = __ballot(take);
            if (bal) {
                const int lane = threadIdx.x & 63;
                unsigned base = 0;
                if (lane == 0) base = atomicAdd(&counts[which], (unsigned)__popcll(bal));
                base = __shfl(base, 0, 64);
                if (take) (which == 0 ? wide_rows : mid_rows)[base + __popcll(bal & ((1ull << lane) - 1))] = (int32_t)v;
            }
        }
        if (owned && d >= hub) {
            const unsigned parts = (unsigned)((d + EGO_PART - 1) / EGO_PART);
            const unsigned base = atomicAdd(&counts[2], parts);
            for (unsigned p = 0; p < parts; ++p) {
                hub_parts[3 * (size_t)(base + p)] = (int32_t)v;
                hub_parts[3 * (size_t)(base + p) + 1] = (int32_t)p;
                hub_parts[3 * (size_t)(base + p) + 2] = (int32_t)parts;
            }
        }
    }
    // one thread per 4-byte word of a slot: 32 consecutive threads write one line
    uint32_t *words = reinterpret_cast<uint32_t *>(slots);
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < n * 32; idx += (int64_t)gridDim.x * 256) {
        const int64_t r = idx >> 5;
        const int wi = (int)(idx & 31);
        const int64_t b = row_ptr[r];
        const int64_t d = row_ptr[r + 1] - b;
        uint32_t word;
        if (wi < 2) {
            const double rs = rowsum ? rowsum[r] : (double)d;
            const unsigned long long bits = (unsigned long long)__double_as_longlong(rs);
            word = wi == 0 ? (uint32_t)bits : (uint32_t)(bits >> 32);
        } else if (wi < 4) {
            const unsigned long long bd =
                (unsigned long long)b | ((unsigned long long)(d < (int64_t)EGO_DEG_SAT ? d : (int64_t)EGO_DEG_SAT) << EGO_DEG_SHIFT);
            word = wi == 2 ? (uint32_t)bd : (uint32_t)(bd >> 32);
        } else {
            const int k = wi - 4;
            word = k < d ? (uint32_t)col[b + k] : 0xffffffffu;
        }
        words[idx] = word;
    }
}

// The membership filter of ego(v): a Bloom filter of EGO_FILTER_WORDS x 32 bits with TWO bits per member -- bit
// (id mod 4096) and bit ((id >> 12) mod 4096) -- in the group's LDS slice.  With at most 33 members a foreign id passes
// both probes with probability ~1e-4 (one probe: 0.8 %, which sent two thirds of all WAVEFRONT steps down the general
// path: a wavefront takes it when any of its eight groups does).  Graphs below 4096 nodes: the first probe is exact.
constexpr int EGO_FILTER_WORDS = 128;
__device__ __forceinline__ unsigned ego_filter_bit(const unsigned *flt, int32_t b)
{
    const unsigned u = (unsigned)b;
    const unsigned w1 = flt[(u >> 5) & (EGO_FILTER_WORDS - 1)] >> (u & 31u);
    const unsigned w2 = flt[(u >> 17) & (EGO_FILTER_WORDS - 1)] >> ((u >> 12) & 31u);
    return w1 & w2 & 1u;
}
__device__ __forceinline__ void ego_filter_set(unsigned *flt, int32_t b)
{
    const unsigned u = (unsigned)b;
    atomicOr(&flt[(u >> 5) & (EGO_FILTER_WORDS - 1)], 1u << (u & 31u));
    atomicOr(&flt[(u >> 17) & (EGO_FILTER_WORDS - 1)], 1u << ((u >> 12) & 31u));
}

// the byte of a wavefront ballot that belongs to this lane's group of eight (gshift = first lane of the group)
__device__ __forceinline__ unsigned ego_group_bits(unsigned long long ballot, int gshift)
{
    const unsigned half = (gshift & 32) ? (unsigned)(ballot >> 32) : (unsigned)ballot;
    return __builtin_amdgcn_ubfe(half, (unsigned)gshift & 31u, 8u);
}

// membership of the eight ids b (one per lane of the group, -1 = none) in ego(v) = {v} U {uu[0..3] of the group's
// lanes}: the filter decides whether the exact all-pairs shuffle compare has to run at all
template <int SLOTS>
__device__ __forceinline__ bool ego_chunk_inside(int32_t b, int32_t v, const int32_t (&uu)[SLOTS],
                                                 const unsigned *flt, int gshift, int lane)
{
    constexpr int G = 8;
    const bool live = b >= 0;
    const bool maybe = live && ego_filter_bit(flt, b);
    unsigned match = 0;
    if (ego_group_bits(__ballot(maybe), gshift)) {              // uniform over the group
#pragma unroll
        for (int sidx = 0; sidx < G; ++sidx) {
            const int32_t bs = __shfl(b, sidx, G);
            bool hit = false;
#pragma unroll
            for (int i = 0; i < SLOTS; ++i) hit |= bs == uu[i];
            if (ego_group_bits(__ballot(hit), gshift)) match |= 1u << sidx;
        }
    }
    return live && (((match >> lane) & 1u) || b == v);
}

// Nodes with at most EGO_GROUP_MAX out-neighbours: eight lanes per node.  Per member a of ego(v) the group reads the
// slot of a with one 16-byte load per lane (lane 0: row sum, begin | length; lanes 1 - 7: four ids each) and tests
// the ids for membership; WEIGHTS are read for the MATCHED arcs only (the first version scanned ids + 8-byte weights
// of every member row from the CSR: 25-50x the compulsory traffic, profiles/r05_dw5m_pmc.json):
//     internal += w(a -> b)                       for b in ego(v)   (undirected: b >= a, every edge once)
//     external += rowsum(a)                       no arc of row(a) ends in ego(v)       -- no subtraction
//              += 0                               every arc does                         -- exactly 0
//              += rowsum(a) - matched weight      otherwise, unless that difference lost more than six bits to
//                                                 cancellation: then the unmatched weights are added one by one
// The members are taken BATCH at a time: the slots of a whole batch are requested before the first is looked at.
// FAST PATH (the kernel was bound by its VALU instruction stream -- 313 instructions per member, every SIMD 100 % busy,
// the memory system at 17 G requests/s, profiles/r05_egonet.txt): a member whose ids all miss the filter and whose
// row fits its slot contributes rowsum(a) to `external` and nothing else -- four filter probes per lane, one ballot,
// no header broadcast (lane 0 holds the row sum itself).  Everything else -- a filter hit, a row beyond 28 ids, a
// hub row -- takes the general path below.
// The external shares are added by lane 0 in member order, the rest are per-lane sequential sums and a fixed
// butterfly: bitwise reproducible, independent of the launch geometry.
template <int BATCH, int SLOTS, bool DIRECTED>
__global__ __launch_bounds__(256) void egonet_group_kernel(
    const int64_t *__restrict__ row_ptr, const int32_t *__restrict__ col,
    const double *__restrict__ w, const EgoSlot *__restrict__ slots,
    int64_t row_begin, int64_t row_end, const int32_t *__restrict__ rows, const unsigned *__restrict__ n_rows,
    double *__restrict__ internal, double *__restrict__ external)
{
    constexpr bool directed = DIRECTED;      // two instances: the undirected one carries the weights of row(v) in LDS
    // rows == nullptr: the nodes of [row_begin, row_end) with at most 8 SLOTS neighbours; else: the listed nodes
    constexpr int EGO_SLOTS = SLOTS;
    constexpr int EGO_GROUP_MAX = 8 * SLOTS;
    constexpr int G = 8;
    __shared__ unsigned ego_filter[256 / G][EGO_FILTER_WORDS];
    __shared__ int32_t ego_id[256 / G][EGO_GROUP_MAX];
    extern __shared__ double ego_w[];      // undirected graphs only (dynamic: 0 bytes otherwise): the weights of row(v) = w(a -> v)
    const int lane = threadIdx.x % G;
    const int gshift = (threadIdx.x & 63) & ~(G - 1);
    const int64_t group = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / G;
    const int64_t ngroups = (int64_t)gridDim.x * blockDim.x / G;
    unsigned *flt = ego_filter[threadIdx.x / G];
    int32_t *mid = ego_id[threadIdx.x / G];
    double *mw = ego_w + (threadIdx.x / G) * EGO_GROUP_MAX;
    const int4 *slot16 = reinterpret_cast<const int4 *>(slots);
    const int64_t first = rows ? 0 : row_begin, last = rows ? (int64_t)n_rows[0] : row_end;
    for (int64_t it = first + group; it < last; it += ngroups) {
        const int64_t v = rows ? (int64_t)rows[it] : it;
        const int64_t vb = row_ptr[v], ve = row_ptr[v + 1];
        const int dv = (int)(ve - vb);
        if (ve - vb > EGO_GROUP_MAX) continue;                  // uniform over the group
        int32_t uu[EGO_SLOTS];
#pragma unroll
        for (int i = 0; i < EGO_SLOTS; ++i) {
            const int64_t idx = vb + lane + (int64_t)G * i;
            uu[i] = (idx < ve) ? col[idx] : -2;
        }
        bool mine_v = false;
#pragma unroll
        for (int i = 0; i < EGO_SLOTS; ++i) mine_v |= uu[i] == (int32_t)v;
        const bool v_in_row = ego_group_bits(__ballot(mine_v), gshift) != 0u;
        __builtin_amdgcn_wave_barrier();                       // the previous node's readers are done
#pragma unroll
        for (int i = 0; i < EGO_SLOTS; ++i)
            if (uu[i] >= 0) mid[lane + G * i] = uu[i];
#pragma unroll
        for (int i = 0; i < EGO_FILTER_WORDS / (4 * G); ++i)
            reinterpret_cast<uint4 *>(flt)[lane + G * i] = make_uint4(0u, 0u, 0u, 0u);
        __builtin_amdgcn_wave_barrier();
        // UNDIRECTED graphs: every member's row holds the arc back to v (the CSR is symmetric), so v stays OUT of the
        // filter -- a member whose only arc into ego(v) is that one still takes the fast path below, with the arc's
        // weight from row(v) (w(a -> v) = w(v -> a), the same double); the exact test finds b == v without the filter
#pragma unroll
        for (int i = 0; i <= EGO_SLOTS; ++i) {
            const int32_t id = i < EGO_SLOTS ? uu[i < EGO_SLOTS ? i : 0] : ((lane == 0 && directed) ? (int32_t)v : -2);
            if (id >= 0 && (directed || id != (int32_t)v)) ego_filter_set(flt, id);
        }
        double ins = 0.0, ext = 0.0;
        // member v itself: every arc of row(v) ends in ego(v)
#pragma unroll
        for (int i = 0; i < EGO_SLOTS; ++i)
            if (uu[i] >= 0) {
                const double x = w ? w[vb + lane + (int64_t)G * i] : 1.0;
                if (!directed) mw[lane + G * i] = x;
                if (directed || uu[i] >= (int32_t)v) ins += x;
            }
        __builtin_amdgcn_wave_barrier();

        for (int mb = 0; mb < dv; mb += BATCH) {
            int4 q[BATCH];
#pragma unroll
            for (int k = 0; k < BATCH; ++k) {
                q[k] = make_int4(-1, -1, -1, -1);
                if (mb + k < dv) {
                    const int32_t a = mid[mb + k];
                    if (a != (int32_t)v) q[k] = slot16[(int64_t)a * G + lane];
                }
            }
#pragma unroll
            for (int k = 0; k < BATCH; ++k) {
                const int m = mb + k;
                if (m >= dv) break;
                const int32_t a = mid[m];
                if (a == (int32_t)v) continue;                  // a self-loop: row(v) is counted above
                {
                    // fast path: lanes 1 - 7 probe the filter with their four ids (a pad id -1 probes like any other: a hit
                    // only costs the general path); lane 0 contributes "the row does not fit its slot" (length in bits
                    // 8 .. 31 of its fourth word)
                    const unsigned hit = lane ? (ego_filter_bit(flt, q[k].x) | ego_filter_bit(flt, q[k].y) |
                                                 ego_filter_bit(flt, q[k].z) | ego_filter_bit(flt, q[k].w))
                                              : (unsigned)(((unsigned)q[k].w >> 8) > (unsigned)EGO_SLOT_IDS);
                    if (ego_group_bits(__ballot(hit != 0u), gshift) == 0u) {
                        const double rs0 = __longlong_as_double((long long)(((unsigned long long)(unsigned)q[k].y << 32) | (unsigned)q[k].x));
                        if (directed) {
                            if (lane == 0) ext += rs0;
                            continue;
                        }
                        // undirected: exactly one arc of row(a) ends in ego(v), the one back to v
                        const double wva = mw[m];
                        const double e = rs0 - wva;
                        const bool cancelled = lane == 0 && !(e * 64.0 >= rs0);
                        if (ego_group_bits(__ballot(cancelled), gshift) == 0u) {
                            if (lane == 0) {
                                ext += e;
                                if ((int32_t)v >= a) ins += wva;
                            }
                            continue;
                        }
                    }
                }
                // lane 0 of the group holds the header of the slot
                const unsigned rs_lo = (unsigned)__shfl(q[k].x, 0, G), rs_hi = (unsigned)__shfl(q[k].y, 0, G);
                const unsigned bd_lo = (unsigned)__shfl(q[k].z, 0, G), bd_hi = (unsigned)__shfl(q[k].w, 0, G);
                const double rs = __longlong_as_double((long long)(((unsigned long long)rs_hi << 32) | rs_lo));
                const unsigned long long bd = ((unsigned long long)bd_hi << 32) | bd_lo;
                const int64_t ab = (int64_t)(bd & EGO_BEGIN_MASK);
                int64_t da = (int64_t)(bd >> EGO_DEG_SHIFT);
                if (da == (int64_t)EGO_DEG_SAT) da = row_ptr[a + 1] - ab;
                if (da == 0) continue;
                const int64_t ae = ab + da;
                if (da <= (int64_t)EGO_SLOTS * G * (ilog2_i64(da) + 2)) {
                    int cnt = 0;
                    double msum = 0.0;
                    auto chunk = [&](int32_t b, int64_t j) {
                        const bool inside = ego_chunk_inside<SLOTS>(b, (int32_t)v, uu, flt, gshift, lane);
                        cnt += __popc(ego_group_bits(__ballot(inside), gshift));
                        if (inside) {
                            const double x = w ? w[j] : 1.0;
                            msum += x;
                            if (directed || b >= a) ins += x;
                        }
                    };
                    // ids 4 (lane - 1) .. 4 (lane - 1) + 3 of the row sit in this lane's quarter of the slot
                    const int64_t j4 = ab + 4 * (lane - 1);
                    chunk(lane ? q[k].x : -1, j4);
                    if (da > 1) chunk(lane ? q[k].y : -1, j4 + 1);
                    if (da > 2) chunk(lane ? q[k].z : -1, j4 + 2);
                    if (da > 3) chunk(lane ? q[k].w : -1, j4 + 3);
                    for (int64_t j0 = ab + EGO_SLOT_IDS; j0 < ae; j0 += G)
                        chunk(j0 + lane < ae ? col[j0 + lane] : -1, j0 + lane);
                    if (cnt == 0) {
                        if (lane == 0) ext += rs;
                    } else if (cnt != da) {
#pragma unroll
                        for (int off = 1; off < G; off <<= 1) msum += __shfl_xor(msum, off, G);
                        const double e = rs - msum;
                        if (e * 64.0 >= rs) {
                            if (lane == 0) ext += e;
                        } else {
                            // nearly closed row: the difference would carry the rounding of the two sums; add the arcs
                            // that leave the ego set one by one instead
                            for (int64_t j0 = ab; j0 < ae; j0 += G) {
                                const int32_t b = j0 + lane < ae ? col[j0 + lane] : -1;
                                const bool inside = ego_chunk_inside<SLOTS>(b, (int32_t)v, uu, flt, gshift, lane);
                                if (b >= 0 && !inside) ext += w ? w[j0 + lane] : 1.0;
                            }
                        }
                    }
                } else {
                    // long row (a hub): look the ego members up in it
                    int matched = 0;
                    double in_all = 0.0;
#pragma unroll
                    for (int i = 0; i <= EGO_SLOTS; ++i) {
                        int32_t key = -2;
                        if (i < EGO_SLOTS) key = uu[i < EGO_SLOTS ? i : 0];
                        else if (lane == 0 && !v_in_row) key = (int32_t)v;
                        if (key >= 0) {
                            const int64_t pos = lower_bound_row(col, ab, ae, key);
                            if (pos < ae && col[pos] == key) {
                                const double x = w ? w[pos] : 1.0;
                                ++matched;
                                in_all += x;
                                if (directed || key >= a) ins += x;
                            }
                        }
                    }
#pragma unroll
                    for (int off = 1; off < G; off <<= 1) {
                        matched += __shfl_xor(matched, off, G);
                        in_all += __shfl_xor(in_all, off, G);
                    }
                    if (lane == 0 && matched != da) ext += rs - in_all;
                }
            }
        }
#pragma unroll
        for (int off = 1; off < G; off <<= 1) {
            ins += __shfl_xor(ins, off, G);
            ext += __shfl_xor(ext, off, G);
        }
        if (lane == 0) { internal[v] = ins; external[v] = ext; }
    }
}

// Nodes with more than 64 out-neighbours (round 5; replaces the wavefront / workgroup kernels of rounds 1 - 4, which
// searched ego(v) in global memory for every arc of every member: 38 ms for the hubs of a weighted BA 1 M / 10 M graph).
// WAVES = 1: a wavefront per node (four nodes per workgroup), WAVES = 4: a 256-thread workgroup per node.
//   * ego(v) as a Bloom filter in LDS (two bits per member, 16+ bits of filter per member while it fits); an id that
//     passes both probes is confirmed by a binary search in row(v) itself (ascending ids);
//   * ONE LANE PER MEMBER a: the lane reads the member's slot (row sum, begin | length, first 28 ids) and tests the
//     ids one after the other -- a filter hit costs that lane a search, not the whole group (eight lanes per member
//     made every group wait for the group with a hit);
//   * members whose rows do not fit a slot are taken afterwards by the whole wavefront, 64 ids per step -- or, when the
//     member is the far bigger hub, by looking ego(v) up in row(a), as before;
//   * weights only for the arcs that end in ego(v); external = rowsum(a) - matched with the guards of the group kernel.
// Lane <-> member and lane <-> chunk position are functions of the row alone: bitwise reproducible.
// ... confirmed through a two-level search: every `stride`-th id of row(v) sits in LDS (samp[0 .. ns)), the binary search
// over the samples costs no memory round trip, the remaining `stride` ids are searched in row(v) itself -- fourteen
// dependent global loads per confirmed id (a hub's row) made the hub-to-hub pairs of a power-law graph the whole cost
struct EgoBigSet {
    const unsigned *flt;
    unsigned bit_mask;
    const int32_t *samp;
    int ns, stride;
    const int32_t *col;
    int64_t vb, ve;
    int32_t v;
};
__device__ __forceinline__ bool ego_big_member(const EgoBigSet &S, int32_t b)
{
    const unsigned h1 = (unsigned)b & S.bit_mask, h2 = (((unsigned)b * 0x9E3779B1u) >> 7) & S.bit_mask;
    if (!((S.flt[h1 >> 5] >> (h1 & 31u)) & (S.flt[h2 >> 5] >> (h2 & 31u)) & 1u)) return false;
    if (b == S.v) return true;
    int lo = 0, hi = S.ns;                                      // first sample > b
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (S.samp[mid] <= b) lo = mid + 1; else hi = mid;
    }
    if (lo == 0) return false;
    const int64_t begin = S.vb + (int64_t)(lo - 1) * S.stride;
    const int64_t end = begin + S.stride < S.ve ? begin + S.stride : S.ve;
    return find_in_row(S.col, begin, end, b) >= 0;
}

template <int WAVES>
__global__ __launch_bounds__(256) void egonet_big_kernel(
    const int64_t *__restrict__ row_ptr, const int32_t *__restrict__ col, const double *__restrict__ w,
    const EgoSlot *__restrict__ slots, int directed, const int32_t *__restrict__ rows, const unsigned *__restrict__ n_rows,
    int filter_words, double *__restrict__ internal, double *__restrict__ external, double *__restrict__ part_out)
{
    // WAVES == 1: rows = node ids, results to internal / external.  WAVES == 4: rows = {node, part, parts} triples, the
    // members [part EGO_PART, (part + 1) EGO_PART) of the node, results to part_out[2 entry], [2 entry + 1]
    extern __shared__ unsigned ego_big_lds[];
    __shared__ double red[2][4];
    constexpr int T = 64 * WAVES;                               // lanes per node
    constexpr int NODES = 4 / WAVES;                            // nodes per workgroup
    const int wlane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int tid = threadIdx.x % T;
    constexpr int SAMPLES = WAVES > 1 ? 1024 : 64;              // ids of row(v) kept in LDS for the two-level search
    unsigned *flt = ego_big_lds + (size_t)(threadIdx.x / T) * (filter_words + SAMPLES);
    int32_t *samp = reinterpret_cast<int32_t *>(flt + filter_words);
    const unsigned bit_mask = (unsigned)filter_words * 32u - 1u;
    const int4 *slot16 = reinterpret_cast<const int4 *>(slots);
    const int64_t count = (int64_t)n_rows[0];
    auto node_sync = [&] { if (WAVES > 1) __syncthreads(); else __builtin_amdgcn_wave_barrier(); };
    for (int64_t it = (int64_t)blockIdx.x * NODES + threadIdx.x / T; it < count; it += (int64_t)gridDim.x * NODES) {
        const int32_t v = WAVES > 1 ? rows[3 * it] : rows[it];
        const int64_t vb = row_ptr[v], ve = row_ptr[v + 1], dv = ve - vb;
        const int64_t m_lo = WAVES > 1 ? (int64_t)rows[3 * it + 1] * EGO_PART : 0;
        const int64_t m_hi = WAVES > 1 ? (m_lo + EGO_PART < dv ? m_lo + EGO_PART : dv) : dv;
        node_sync();                                            // the previous node's readers are done
        for (int i = tid; i < filter_words; i += T) flt[i] = 0u;
        node_sync();
        for (int64_t m = tid; m <= dv; m += T) {
            const int32_t id = m < dv ? col[vb + m] : v;
            const unsigned h1 = (unsigned)id & bit_mask, h2 = (((unsigned)id * 0x9E3779B1u) >> 7) & bit_mask;
            atomicOr(&flt[h1 >> 5], 1u << (h1 & 31u));
            atomicOr(&flt[h2 >> 5], 1u << (h2 & 31u));
        }
        const int stride = (int)((dv + SAMPLES - 1) / SAMPLES) > 0 ? (int)((dv + SAMPLES - 1) / SAMPLES) : 1;
        const int ns = (int)((dv + stride - 1) / stride);
        for (int i = tid; i < ns; i += T) samp[i] = col[vb + (int64_t)i * stride];
        node_sync();
        const EgoBigSet S{flt, bit_mask, samp, ns, stride, col, vb, ve, v};
        double ins = 0.0, ext = 0.0;
        // member v itself: every arc of row(v) ends in ego(v)
        for (int64_t m = m_lo + tid; m < m_hi; m += T)
            if (directed || col[vb + m] >= v) ins += w ? w[vb + m] : 1.0;
        for (int64_t m0 = m_lo; m0 < m_hi; m0 += T) {
            const int64_t m = m0 + tid;
            int32_t a = m < m_hi ? col[vb + m] : -1;
            if (a == v) a = -1;                                 // a self-loop: counted above
            int64_t ab = 0, da = 0;
            double rs = 0.0;
            if (a >= 0) {
                const int4 h = slot16[(int64_t)a * 8];
                rs = __longlong_as_double((long long)(((unsigned long long)(unsigned)h.y << 32) | (unsigned)h.x));
                const unsigned long long bd = ((unsigned long long)(unsigned)h.w << 32) | (unsigned)h.z;
                ab = (int64_t)(bd & EGO_BEGIN_MASK);
                da = (int64_t)(bd >> EGO_DEG_SHIFT);
                if (da == (int64_t)EGO_DEG_SAT) da = row_ptr[a + 1] - ab;
                // the first id sits behind the header in the same quarter of the slot
            }
            const bool is_long = a >= 0 && da > EGO_SLOT_IDS;
            if (a >= 0 && !is_long && da > 0) {
                int cnt = 0;
                double msum = 0.0;
                for (int q = 0; q * 4 < da; ++q) {
                    const int4 ids = slot16[(int64_t)a * 8 + 1 + q];
                    const int32_t b4[4] = {ids.x, ids.y, ids.z, ids.w};
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int k = q * 4 + j;
                        if (k < da && ego_big_member(S, b4[j])) {
                            const double x = w ? w[ab + k] : 1.0;
                            ++cnt;
                            msum += x;
                            if (directed || b4[j] >= a) ins += x;
                        }
                    }
                }
                if (cnt == 0) ext += rs;
                else if (cnt != da) {
                    const double e = rs - msum;
                    if (e * 64.0 >= rs) ext += e;
                    else {
                        // nearly closed row: add the arcs that leave the ego set one by one
                        for (int64_t k = 0; k < da; ++k) {
                            const int32_t b = col[ab + k];
                            if (!ego_big_member(S, b)) ext += w ? w[ab + k] : 1.0;
                        }
                    }
                }
            }
            // members whose rows do not fit a slot: the whole wavefront takes them one after the other
            unsigned long long todo = __ballot(is_long);
            while (todo) {
                const int src = __ffsll((long long)todo) - 1;
                todo &= todo - 1;
                const int32_t a_s = __shfl(a, src, 64);
                const int64_t ab_s = __shfl(ab, src, 64), da_s = __shfl(da, src, 64);
                const double rs_s = __shfl(rs, src, 64);
                const int64_t members = dv + 1;
                if (da_s * (ilog2_i64(members) + 2) <= members * (int64_t)(ilog2_i64(da_s) + 2) * 4) {
                    // scan row(a), 64 ids per step
                    unsigned long long cnt = 0;
                    double msum = 0.0;
                    for (int64_t k0 = 0; k0 < da_s; k0 += 64) {
                        const int64_t k = k0 + wlane;
                        const int32_t b = k < da_s ? col[ab_s + k] : -1;
                        const bool inside = b >= 0 && ego_big_member(S, b);
                        cnt += (unsigned long long)__popcll(__ballot(inside));
                        if (inside) {
                            const double x = w ? w[ab_s + k] : 1.0;
                            msum += x;
                            if (directed || b >= a_s) ins += x;
                        }
                    }
                    if (cnt != 0 && (int64_t)cnt != da_s) {
                        msum = grx_group_sum<64>(msum);
                        const double e = rs_s - msum;
                        if (e * 64.0 >= rs_s) {
                            if (wlane == src) ext += e;
                        } else {
                            for (int64_t k0 = 0; k0 < da_s; k0 += 64) {
                                const int64_t k = k0 + wlane;
                                const int32_t b = k < da_s ? col[ab_s + k] : -1;
                                if (b >= 0 && !ego_big_member(S, b)) ext += w ? w[ab_s + k] : 1.0;
                            }
                        }
                    } else if (cnt == 0 && wlane == src) {
                        ext += rs_s;
                    }
                } else {
                    // a is by far the bigger hub: look the members of ego(v) up in row(a)
                    long long matched = 0;
                    double in_all = 0.0;
                    for (int64_t t0 = 0; t0 <= dv; t0 += 64) {
                        const int64_t t = t0 + wlane;
                        int32_t key = -1;
                        if (t < dv) key = col[vb + t];
                        else if (t == dv && find_in_row(col, vb, ve, v) < 0) key = v;      // v itself, once
                        if (key >= 0) {
                            const int64_t pos = find_in_row(col, ab_s, ab_s + da_s, key);
                            if (pos >= 0) {
                                const double x = w ? w[pos] : 1.0;
                                ++matched;
                                in_all += x;
                                if (directed || key >= a_s) ins += x;
                            }
                        }
                    }
                    matched = (long long)grx_group_sum<64>((double)matched);
                    in_all = grx_group_sum<64>(in_all);
                    if (wlane == src && matched != da_s) ext += rs_s - in_all;
                }
            }
        }
        ins = grx_group_sum<64>(ins);
        ext = grx_group_sum<64>(ext);
        if constexpr (WAVES > 1) {
            if (wlane == 0) { red[0][wave] = ins; red[1][wave] = ext; }
            __syncthreads();
            if (threadIdx.x == 0) {
                double si = 0.0, se = 0.0;
                for (int i = 0; i < WAVES; ++i) { si += red[0][i]; se += red[1][i]; }
                part_out[2 * it] = si;
                part_out[2 * it + 1] = se;
            }
        } else {
            if (wlane == 0) { internal[v] = ins; external[v] = ext; }
        }
    }
}

// the parts of a hub row, added in part order (the entries of a row are consecutive in the list)
__global__ __launch_bounds__(256) void egonet_combine_kernel(const int32_t *__restrict__ parts, const unsigned *__restrict__ n_parts,
                                                             const double *__restrict__ part_out, double *__restrict__ internal,
                                                             double *__restrict__ external)
{
    const int64_t count = (int64_t)n_parts[0];
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < count; e += (int64_t)gridDim.x * 256) {
        if (parts[3 * e + 1] != 0) continue;
        double si = 0.0, se = 0.0;
        for (int p = 0; p < parts[3 * e + 2]; ++p) { si += part_out[2 * (e + p)]; se += part_out[2 * (e + p) + 1]; }
        internal[parts[3 * e]] = si;
        external[parts[3 * e]] = se;
    }
}

// ---------------------------------------------------------------------------------------
// ego-net features of UNWEIGHTED UNDIRECTED graphs through per-node triangle counts
// ---------------------------------------------------------------------------------------
// With A' = adjacency without the diagonal, d'(v) its degrees, L(v) the self-loop flags and
// T(v) the number of triangles through v (= edges among the neighbours of v):
//     internal(v) = d'(v) + T(v) + sum_{a in ego(v)} L(a)
//     external(v) = sum_{a in ego(v)} d'(a) - 2 (d'(v) + T(v))
// (networkx.py:71-83 counted edge by edge; all quantities are integers, so this is exact.)
// T comes from the degree-oriented graph (arc u->v iff (d'(u),u) < (d'(v),v)): every triangle
// is found exactly once as a common out-neighbour of the two ends of its lowest arc; oriented
// lists are short even for power-law hubs.
// The ARCS are the work items (round 3).  Rounds 1-2 gave every source u an 8-lane group that walked its arcs: N+(u)
// in registers, a 256-bit membership filter, an all-pairs shuffle compare of 8-id chunks.  Its counters showed the
// instruction stream and the per-source dependency chain as the bound, not the memory system: the exact test (eight
// shuffles, each followed by three compares and a ballot) ran whenever ANY of the eight groups of a wavefront had a
// filter hit, i.e. nearly always (SQ_ACTIVE_INST_ANY 7x the aggregate kernel's for half its memory traffic), a
// wavefront waited for its longest source row, every source cost three dependent round trips, and only ~4 memory
// instructions were in flight per CU.  0.58 ms at BA 1 M / 10 M.  Steps from there (profiles/r03_triangles.txt):
//   N+(u) broadcast into registers, compares instead of shuffles                     0.47 ms
//   arcs as work items, all-pairs compare by DPP lane rotations, 8 lanes per arc      0.43 ms
//   16 lanes per arc, binary search across the lanes (ds_bpermute)                   0.44 ms
//   + a lane-per-arc pass that touches the lists first (64 random lines in flight)   0.52 ms  (rejected)
//   8-byte table entries read by one lane per arc and passed on by ds_bpermute        0.41 ms
//   the four searches of a group interleaved, unconditional loads (no exec juggling)  0.39 ms
// Now: a wavefront takes 64 consecutive arcs u->v; a 16-lane group handles four of them at a time, has the first
// sixteen ids of all eight lists in flight at once -- degree ordering keeps 99.5 % of the oriented lists of the
// BASELINE graphs that short -- and intersects by BINARY SEARCH: the lists are ascending, every lane looks its id of
// N+(v) up among the sixteen ids of N+(u) spread over the group's lanes (five ds_bpermute probes).  Two round trips
// per arc, no per-source loop, every group always has work; VALU 48 % busy, LDS 29 %, 22 G L2 misses/s.
constexpr int TRI_AG = 16;                   // lanes per arc
constexpr int TRI_ARCS = 4;                  // arcs per group and iteration
constexpr int32_t TRI_PAD = 0x7fffffff;      // pads N+(u) to sixteen ascending ids

// is y one of the sixteen ascending ids the group's lanes hold in a?  group_byte = 4 * (first lane of the group)
__device__ __forceinline__ bool tri_search16(int32_t y, int32_t a, int group_byte)
{
    int pos = group_byte;                                       // byte address of lane `lower bound so far`
    int32_t t = __builtin_amdgcn_ds_bpermute(pos + 7 * 4, a);
    pos += (t < y) ? 8 * 4 : 0;
    t = __builtin_amdgcn_ds_bpermute(pos + 3 * 4, a);
    pos += (t < y) ? 4 * 4 : 0;
    t = __builtin_amdgcn_ds_bpermute(pos + 1 * 4, a);
    pos += (t < y) ? 2 * 4 : 0;
    t = __builtin_amdgcn_ds_bpermute(pos, a);
    pos += (t < y) ? 4 : 0;
    t = __builtin_amdgcn_ds_bpermute(pos, a);
    return t == y;
}

__device__ __forceinline__ unsigned long long tri_bperm64(unsigned long long x, int src_byte)
{
    const int lo = __builtin_amdgcn_ds_bpermute(src_byte, (int)(unsigned)x);
    const int hi = __builtin_amdgcn_ds_bpermute(src_byte, (int)(unsigned)(x >> 32));
    return ((unsigned long long)(unsigned)hi << 32) | (unsigned)lo;
}

// o_arc[k] for the k-th oriented arc u->v (grx.h): begin of N+(v) | |N+(v)| << 32 | |N+(u)| << 42 | (k - begin of
// N+(u)) << 52, the three 10-bit fields saturating at 1023 -- such arcs (hubs of the ORIENTED graph: out-degree is at
// most sqrt(2 m)) are looked up from o_row_ptr instead.
constexpr int TRI_FIELD = 10;
constexpr unsigned TRI_SAT = (1u << TRI_FIELD) - 1;

// A wavefront takes 64 consecutive arcs per iteration: one LANE per arc reads the 8-byte table entry (a coalesced
// 512-byte read), then one 16-lane GROUP per arc, four arcs at a time, receives the entries from the lanes that read
// them, loads the lists and intersects them by the binary search above.
// The counters of the first TRI_HUBS vertices (rows are in degree-descending order: the hubs) are kept per workgroup in
// LDS and added to T once at the end.  Every corner of every triangle is one atomic increment, and on a power-law graph
// a few vertices take most of them (BA 1 M / 10 M: 12 275 of 180 k on vertex 0, 35 k on the first sixteen); atomics to
// ONE address are served one after the other (~9 ns each): 0.15 of the kernel's 0.41 ms was that queue (measured by
// dropping the atomics below an index: 0.41 -> 0.30 without vertex 0, 0.25 without the first sixteen).  With the LDS
// counters 0.27 ms; workgroups of 512 / 1024 threads, which collect more per flush, were slower (0.28 / 0.30).
constexpr int TRI_HUBS = 256;
constexpr int TRI_THREADS = 256;

__device__ __forceinline__ void tri_add(unsigned long long *__restrict__ T, unsigned long long *s_hub, int32_t idx, unsigned c)
{
    if (idx < TRI_HUBS) atomicAdd(&s_hub[idx], (unsigned long long)c);
    else atomicAdd(&T[idx], (unsigned long long)c);
}

__global__ __launch_bounds__(TRI_THREADS) void triangle_count_arcs_kernel(
    const int64_t *__restrict__ o_row_ptr, const int32_t *__restrict__ o_col,
    const unsigned long long *__restrict__ o_arc, int64_t row_begin, int64_t row_end,
    unsigned long long *__restrict__ T)
{
    __shared__ unsigned long long s_hub[TRI_HUBS];
    for (int i = threadIdx.x; i < TRI_HUBS; i += blockDim.x) s_hub[i] = 0;
    __syncthreads();
    constexpr int G = TRI_AG;
    constexpr unsigned long long GMASK = (1ull << G) - 1;
    const int wlane = threadIdx.x & 63;
    const int lane = wlane % G;
    const int gshift = wlane & ~(G - 1);                       // first lane of this group in the wavefront
    const int group_byte = gshift * 4;
    const int g = wlane / G;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    const int64_t kb = o_row_ptr[row_begin], ke = o_row_ptr[row_end];
    for (int64_t base = kb + wave * 64; base < ke; base += nwaves * 64) {
        const int64_t k_mine = base + wlane;
        unsigned long long d_mine = (k_mine < ke) ? o_arc[k_mine] : 0ull;
        // ---- arcs with a saturated field (rare): the whole wavefront serves them one by one from o_row_ptr
        {
            const unsigned vl = (unsigned)(d_mine >> 32) & TRI_SAT, ul = (unsigned)(d_mine >> 42) & TRI_SAT,
                           ps = (unsigned)(d_mine >> 52) & TRI_SAT;
            unsigned long long todo = __ballot(vl == TRI_SAT || ul == TRI_SAT || ps == TRI_SAT);
            if (vl == TRI_SAT || ul == TRI_SAT || ps == TRI_SAT) d_mine = 0ull;      // not for the group phase
            while (todo) {
                const int src = __ffsll((long long)todo) - 1;
                todo &= todo - 1;
                const int64_t k = base + src;
                const int32_t v = o_col[k];
                int64_t lo = row_begin, hi = row_end;           // u: last row with o_row_ptr[row] <= k
                while (hi - lo > 1) {
                    const int64_t mid = (lo + hi) >> 1;
                    if (o_row_ptr[mid] <= k) lo = mid; else hi = mid;
                }
                const int64_t ub = o_row_ptr[lo], ue = o_row_ptr[lo + 1], vb = o_row_ptr[v], ve = o_row_ptr[v + 1];
                unsigned long long c = 0;
                for (int64_t j0 = vb; j0 < ve; j0 += 64) {      // every id of N+(v): binary search in N+(u)
                    const bool have = j0 + wlane < ve;
                    const int32_t y = have ? o_col[j0 + wlane] : -1;
                    int64_t a = ub, e = ue;
                    while (have && a < e) {
                        const int64_t mid = (a + e) >> 1;
                        if (o_col[mid] < y) a = mid + 1; else e = mid;
                    }
                    const bool hit = have && a < ue && o_col[a] == y;
                    if (hit) tri_add(T, s_hub, y, 1u);
                    c += (unsigned long long)__popcll(__ballot(hit));
                }
                if (c && wlane == 0) { tri_add(T, s_hub, v, (unsigned)c); tri_add(T, s_hub, (int32_t)lo, (unsigned)c); }
            }
        }
        // ---- the group phase
#pragma unroll 1
        for (int sub = 0; sub < 64 / ((64 / G) * TRI_ARCS); ++sub) {
            uint32_t vb[TRI_ARCS], ub[TRI_ARCS];
            int vlen[TRI_ARCS], ulen[TRI_ARCS];
#pragma unroll
            for (int j = 0; j < TRI_ARCS; ++j) {
                const int src = sub * 16 + g * TRI_ARCS + j;    // the lane that holds this arc's entry
                const unsigned long long d = tri_bperm64(d_mine, src * 4);
                vb[j] = (uint32_t)d;
                vlen[j] = (int)((unsigned)(d >> 32) & TRI_SAT);
                ulen[j] = (int)((unsigned)(d >> 42) & TRI_SAT);
                ub[j] = (uint32_t)(base + src) - ((unsigned)(d >> 52) & TRI_SAT);
                if (vlen[j] == 0) ulen[j] = 0;                 // nothing to intersect with: do not fetch N+(u) either
                if (ulen[j] == 0) ub[j] = 0;                   // (no arc here: keep the unconditional loads in bounds)
            }
            // eight independent loads per lane (unconditional -- lanes beyond a list read its first id and discard it:
            // a branch around every load cost more than the redundant reads)
            int32_t y0[TRI_ARCS], a0[TRI_ARCS];
#pragma unroll
            for (int j = 0; j < TRI_ARCS; ++j) {
                const bool yv = lane < vlen[j] && ulen[j] > 0, av = lane < ulen[j];
                const int32_t yr = o_col[vb[j] + (yv ? lane : 0)];
                const int32_t ar = o_col[ub[j] + (av ? lane : 0)];
                y0[j] = yv ? yr : -1;
                a0[j] = av ? ar : TRI_PAD;
            }
            // the four binary searches step by step TOGETHER: four independent ds_bpermute in flight per step (one
            // search after the other was twenty serial LDS round trips per group of arcs)
            bool h[TRI_ARCS];
            {
                int pos[TRI_ARCS];
                int32_t t[TRI_ARCS];
#pragma unroll
                for (int j = 0; j < TRI_ARCS; ++j) t[j] = __builtin_amdgcn_ds_bpermute(group_byte + 7 * 4, a0[j]);
#pragma unroll
                for (int j = 0; j < TRI_ARCS; ++j) pos[j] = group_byte + ((t[j] < y0[j]) ? 8 * 4 : 0);
#pragma unroll
                for (int j = 0; j < TRI_ARCS; ++j) t[j] = __builtin_amdgcn_ds_bpermute(pos[j] + 3 * 4, a0[j]);
#pragma unroll
                for (int j = 0; j < TRI_ARCS; ++j) pos[j] += (t[j] < y0[j]) ? 4 * 4 : 0;
#pragma unroll
                for (int j = 0; j < TRI_ARCS; ++j) t[j] = __builtin_amdgcn_ds_bpermute(pos[j] + 1 * 4, a0[j]);
#pragma unroll
                for (int j = 0; j < TRI_ARCS; ++j) pos[j] += (t[j] < y0[j]) ? 2 * 4 : 0;
#pragma unroll
                for (int j = 0; j < TRI_ARCS; ++j) t[j] = __builtin_amdgcn_ds_bpermute(pos[j], a0[j]);
#pragma unroll
                for (int j = 0; j < TRI_ARCS; ++j) pos[j] += (t[j] < y0[j]) ? 4 : 0;
#pragma unroll
                for (int j = 0; j < TRI_ARCS; ++j) t[j] = __builtin_amdgcn_ds_bpermute(pos[j], a0[j]);
#pragma unroll
                for (int j = 0; j < TRI_ARCS; ++j) h[j] = t[j] == y0[j];
            }
            bool longer = false;
#pragma unroll
            for (int j = 0; j < TRI_ARCS; ++j) longer |= (ulen[j] > G || vlen[j] > G) && ulen[j] > 0;
            if (__ballot(h[0] | h[1] | h[2] | h[3] | longer) == 0) continue;     // the usual case: no triangle here
#pragma unroll
            for (int j = 0; j < TRI_ARCS; ++j) {
                unsigned c_arc = 0;
                const unsigned long long b0 = __ballot(h[j]);
                if (b0) {
                    if (h[j]) tri_add(T, s_hub, y0[j], 1u);
                    c_arc = (unsigned)__popcll((b0 >> gshift) & GMASK);
                }
                if (__ballot((ulen[j] > G || vlen[j] > G) && ulen[j] > 0) != 0) {
                    // lists beyond sixteen ids: the remaining chunk pairs, from memory (degree ordering keeps them rare)
                    for (int ja = 0; __ballot(ja < ulen[j]) != 0; ja += G) {
                        const int32_t a = (ja + lane < ulen[j]) ? o_col[ub[j] + ja + lane] : TRI_PAD;
                        for (int jb = (ja == 0) ? G : 0; __ballot(jb < vlen[j] && ja < ulen[j]) != 0; jb += G) {
                            const int32_t y = (jb + lane < vlen[j] && ja < ulen[j]) ? o_col[vb[j] + jb + lane] : -1;
                            const bool hh = tri_search16(y, a, group_byte);
                            const unsigned long long bh = __ballot(hh);
                            if (bh) {
                                if (hh) tri_add(T, s_hub, y, 1u);
                                c_arc += (unsigned)__popcll((bh >> gshift) & GMASK);
                            }
                        }
                    }
                }
                if (__ballot(c_arc != 0) != 0) {
                    if (c_arc && lane == 0) {
                        // the arc's two ends: the target from the column array, the source = the row that owns position k
                        const int64_t k = base + sub * 16 + g * TRI_ARCS + j;
                        const int32_t v = o_col[k];
                        int64_t lo = row_begin, hi = row_end;   // last row with o_row_ptr[row] <= k
                        while (hi - lo > 1) {
                            const int64_t mid = (lo + hi) >> 1;
                            if (o_row_ptr[mid] <= k) lo = mid; else hi = mid;
                        }
                        tri_add(T, s_hub, v, c_arc);
                        tri_add(T, s_hub, (int32_t)lo, c_arc);
                    }
                }
            }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < TRI_HUBS; i += blockDim.x)
        if (s_hub[i]) atomicAdd(&T[i], s_hub[i]);
}

// info[v] = (d'(v) << 1) | L(v)   (int32: the whole table is 4 B/node and stays L2-resident)
__global__ __launch_bounds__(256) void node_info_kernel(int64_t n, const int64_t *__restrict__ row_ptr,
                                                        const int32_t *__restrict__ col,
                                                        int32_t *__restrict__ info)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < n; v += stride) {
        const int64_t b = row_ptr[v], e = row_ptr[v + 1];
        const int64_t loop = find_in_row(col, b, e, (int32_t)v) >= 0 ? 1 : 0;
        info[v] = (int32_t)((((e - b) - loop) << 1) | loop);
    }
}

__device__ __forceinline__ void egonet_finish_row(int64_t v, long long sum_d, long long loops,
                                                  const int32_t *__restrict__ info,
                                                  const unsigned long long *__restrict__ T,
                                                  double *__restrict__ internal, double *__restrict__ external)
{
    const int32_t iv = info[v];
    const long long dv = iv >> 1;
    const long long core = dv + (long long)T[v];
    internal[v] = (double)(core + loops + (iv & 1));
    external[v] = (double)(sum_d + dv - 2 * core);
}

// G = 8 lanes per row; rows with more than hub_deg neighbours are left to the workgroup-per-row
// variant below (integer sums: any order is exact).
__device__ __forceinline__ void egonet_rows_body(
    const int64_t *__restrict__ row_ptr, const int32_t *__restrict__ col,
    const int32_t *__restrict__ info, const unsigned long long *__restrict__ T, int64_t row_begin,
    int64_t row_end, int64_t hub_deg, double *__restrict__ internal, double *__restrict__ external, int64_t block,
    int64_t nblocks)
{
    constexpr int G = 8;
    const int lane = threadIdx.x % G;
    const int64_t group = (block * blockDim.x + threadIdx.x) / G;
    const int64_t ngroups = nblocks * blockDim.x / G;
    for (int64_t v = row_begin + group; v < row_end; v += ngroups) {
        const int64_t b = row_ptr[v], e = row_ptr[v + 1];
        if (e - b > hub_deg) continue;
        long long sum_d = 0, loops = 0;
        for (int64_t k = b + lane; k < e; k += G) {
            const int32_t a = col[k];
            if (a != (int32_t)v) {
                const int32_t ia = info[a];
                sum_d += ia >> 1;
                loops += ia & 1;
            }
        }
#pragma unroll
        for (int off = G / 2; off > 0; off >>= 1) {
            sum_d += __shfl_xor(sum_d, off, G);
            loops += __shfl_xor(loops, off, G);
        }
        if (lane == 0) egonet_finish_row(v, sum_d, loops, info, T, internal, external);
    }
}

__device__ __forceinline__ void egonet_hubs_body(
    const int64_t *__restrict__ row_ptr, const int32_t *__restrict__ col,
    const int32_t *__restrict__ info, const unsigned long long *__restrict__ T, int64_t row_begin,
    int64_t row_end, const int32_t *__restrict__ hub_rows, int64_t n_hubs,
    double *__restrict__ internal, double *__restrict__ external, int64_t block, int64_t nblocks)
{
    __shared__ long long red[2][4];
    for (int64_t h = block; h < n_hubs; h += nblocks) {
        const int64_t v = hub_rows[h];
        if (v < row_begin || v >= row_end) continue;
        const int64_t b = row_ptr[v], e = row_ptr[v + 1];
        long long sum_d = 0, loops = 0;
        for (int64_t k = b + threadIdx.x; k < e; k += 256) {
            const int32_t a = col[k];
            if (a != (int32_t)v) {
                const int32_t ia = info[a];
                sum_d += ia >> 1;
                loops += ia & 1;
            }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            sum_d += __shfl_xor(sum_d, off, 64);
            loops += __shfl_xor(loops, off, 64);
        }
        if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = sum_d; red[1][threadIdx.x >> 6] = loops; }
        __syncthreads();
        if (threadIdx.x == 0)
            egonet_finish_row(v, red[0][0] + red[0][1] + red[0][2] + red[0][3],
                              red[1][0] + red[1][1] + red[1][2] + red[1][3], info, T, internal, external);
        __syncthreads();
    }
}

// ONE launch for both: the first hub_blocks workgroups take the hub rows (a workgroup per row: a chain of dependent
// loads 40 deep for a 10 k-neighbour hub), the others the eight-lanes-per-row pass.  As two launches the hub kernel ran
// alone on a few CUs AFTER the row pass (0.04 ms of a 0.155 ms phase at BA 1 M / 10 M); now the chains start first and
// hide behind the row pass.
__global__ __launch_bounds__(256) void egonet_from_triangles_kernel(
    const int64_t *__restrict__ row_ptr, const int32_t *__restrict__ col,
    const int32_t *__restrict__ info, const unsigned long long *__restrict__ T, int64_t row_begin,
    int64_t row_end, int64_t hub_deg, const int32_t *__restrict__ hub_rows, int64_t n_hubs, int hub_blocks,
    double *__restrict__ internal, double *__restrict__ external)
{
    if ((int)blockIdx.x < hub_blocks)
        egonet_hubs_body(row_ptr, col, info, T, row_begin, row_end, hub_rows, n_hubs, internal, external, blockIdx.x, hub_blocks);
    else
        egonet_rows_body(row_ptr, col, info, T, row_begin, row_end, hub_deg, internal, external,
                         (int64_t)blockIdx.x - hub_blocks, (int64_t)gridDim.x - hub_blocks);
}

// ---------------------------------------------------------------------------------------
// pack: column-major columns -> row-major n x ldr (zero padded)
// ---------------------------------------------------------------------------------------
// columns [c_off, c_off + f) of the row-major block; the pad columns [pad_from, ldr) are zeroed
__global__ __launch_bounds__(256) void pack_rows_kernel(int64_t n, int f, int ldr, GrxPtrTable cols_tab,
                                                        double *__restrict__ rows, int c_off, int pad_from)
{
    const double *const *cols = reinterpret_cast<const double *const *>(cols_tab.p);
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        double *dst = rows + i * ldr;
        for (int c = 0; c < f; ++c) dst[c_off + c] = cols[c][i];
        for (int c = pad_from; c < ldr; ++c) dst[c] = 0.0;
    }
}

// Rows of exactly 64 bytes (ldr = 8: the 5 - 8 retained columns of a generation on the BASELINE graphs).  A wavefront
// turns 64 rows x 8 columns through its LDS slice so that every store instruction writes 1 KiB of consecutive bytes
// (lane l: 16 bytes at l * 16 + k * 1024) -- the thread-per-row form writes 8 bytes per lane at a stride of 64, eight
// times over the same 64 lines.
__global__ __launch_bounds__(256) void pack_rows8_kernel(int64_t n, int f, GrxPtrTable cols_tab, double *__restrict__ rows)
{
    __shared__ double tile[4][64 * 9];                           // [wave][row * 9 + column]: padded against bank conflicts
    const double *const *cols = reinterpret_cast<const double *const *>(cols_tab.p);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    double *mine = tile[wave];
    const int64_t nblocks = (n + 63) / 64;
    for (int64_t blk = (int64_t)blockIdx.x * 4 + wave; blk < nblocks; blk += (int64_t)gridDim.x * 4) {
        const int64_t row0 = blk * 64, i = row0 + lane;
        double v[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) v[c] = (c < f && i < n) ? cols[c][i] : 0.0;
#pragma unroll
        for (int c = 0; c < 8; ++c) mine[lane * 9 + c] = v[c];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int piece = k * 64 + lane;                     // 16-byte piece of the 4 KiB block
            const int r = piece >> 2, c = (piece & 3) * 2;
            if (row0 + r < n) {
                double2 out;
                out.x = mine[r * 9 + c];
                out.y = mine[r * 9 + c + 1];
                *reinterpret_cast<double2 *>(rows + (row0 + r) * 8 + c) = out;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
}

// The same for ldr a multiple of 16 (rows of whole 128-byte lines): a 64-row x 16-column tile goes through LDS so
// that both sides are coalesced -- columns are read 64 rows (512 bytes) at a time, rows written a line at a time.
constexpr int PK_LD = 65;
__global__ __launch_bounds__(256) void pack_rows_tiled_kernel(int64_t n, int f, int ldr, GrxPtrTable cols_tab,
                                                              double *__restrict__ rows, int c_off, int pad_from)
{
    __shared__ double tile[16 * PK_LD];
    const double *const *cols = reinterpret_cast<const double *const *>(cols_tab.p);
    const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
    const int c_end = pad_from < ldr ? ldr : c_off + f;          // the last launch also zeroes the pad columns
    const int64_t ntiles = (n + 63) / 64;
    for (int64_t tile_i = blockIdx.x; tile_i < ntiles; tile_i += gridDim.x) {
        const int64_t row0 = tile_i * 64;
        for (int cb = c_off; cb < c_end; cb += 16) {
            double v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int c = cb + wave + 4 * u - c_off;         // column of this launch's table
                const int64_t i = row0 + lane;
                v[u] = (c < f && i < n) ? cols[c][i] : 0.0;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) tile[(wave + 4 * u) * PK_LD + lane] = v[u];
            __syncthreads();
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int r = (t >> 4) + 16 * u, c = t & 15;
                if (row0 + r < n && cb + c < c_end) rows[(row0 + r) * ldr + cb + c] = tile[c * PK_LD + r];
            }
            __syncthreads();
        }
    }
}

__global__ __launch_bounds__(256) void add_columns_kernel(int64_t n, const double *__restrict__ a,
                                                          const double *__restrict__ b,
                                                          double *__restrict__ out)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) out[i] = a[i] + b[i];
}

// ---------------------------------------------------------------------------------------
// neighbour aggregation
// ---------------------------------------------------------------------------------------
// The reference sums a node's neighbour rows column by column with Series.sum(), i.e. with
// numpy's pairwise summation (numpy/_core/src/umath/loops_utils.h.src, pairwise_sum_DOUBLE) in
// the order G[node] lists the neighbours (features/extract.py:108-113).  The kernels reproduce
// that association bit for bit:
//     cnt < 8     sequential
//     cnt <= 128  r[j] = x[j] + x[j+8] + ...;  ((r0+r1)+(r2+r3)) + ((r4+r5)+(r6+r7));  then the
//                 cnt % 8 trailing elements one by one
//     cnt > 128   binary tree over blocks: split at cnt/2 rounded down to a multiple of 8
//     cnt > 8192  ndarray.sum() walks the column in chunks of 8192 elements (the ufunc buffer
//                 size); each chunk is summed as above and added to the running total
// which happens to be a good GPU shape: the eight accumulators are eight independent gathers.
//
// G lanes cooperate on one output row.  A neighbour's feature row is padded to LDR doubles
// (16/32/64/128 bytes, never straddling a 128-byte line) and fetched by CL = LDR/2 ADJACENT
// lanes, 16 bytes each, so one wave-level load touches 64/CL lines.  Lane = (slot, part):
// part = lane % CL owns columns 2*part, 2*part+1; slot = lane / CL owns the accumulators
// r[slot], r[slot+S], ... (S = G/CL slots, A = 8/S accumulators per lane).  The tree levels
// that pair accumulators of different slots are xor-shuffles, the others are local adds.
// Rows with more than 128 neighbours are cut into the blocks of numpy's recursion by the host
// (AggregatePlan): aggregate_kernel sums each block like a short row and
// aggregate_combine_kernel adds the block sums along the same binary tree.
constexpr int PW_BLOCK = 128;
constexpr int PW_CHUNK = 8192;

// SQDEV: the summands are (m - x)^2 with the per-column values m0, m1 (pandas' nanvar:
// avg = sum / count; ((avg - values) ** 2).sum() -- the same pairwise tree over the transformed values).
template <int LDR, int G, bool SQDEV = false>
__device__ __forceinline__ void pairwise_segment(const int32_t *__restrict__ col, const double *__restrict__ rows,
                                                 int64_t row_stride, int64_t b, int cnt, int part, int slot,
                                                 double &o0, double &o1, double m0 = 0.0, double m1 = 0.0)
{
    // the squares must be rounded before they are added, as numpy does: no fma contraction in here
#pragma clang fp contract(off)
    auto tr = [&](double2 x) {
#pragma clang fp contract(off)
        if (SQDEV) { const double t0 = m0 - x.x, t1 = m1 - x.y; x.x = t0 * t0; x.y = t1 * t1; }
        return x;
    };
    constexpr int CL = (LDR >= 16 ? 16 : LDR) / 2;
    constexpr int S = G / CL, A = 8 / S;
    static_assert(S >= 1 && S <= 8 && S * A == 8, "lane group must hold 1..8 neighbour slots");
    const double *base = rows + 2 * part;
    double res0 = 0.0, res1 = 0.0;
    const int c8 = cnt & ~7;
    // The cnt % 8 trailing neighbours are ADDED last, one by one, but nothing stops their loads from being issued
    // first: indices, then rows, all in flight with the main part's gathers (inside `if (idx < cnt)` each index
    // load was waited for before its gather, and each gather before the next index: up to 2 A round trips in a
    // row).  Clamped indices re-read the last neighbour; the mask is applied where the values are used.
    const int rem = cnt - c8;
    double2 xt[A];
    if (rem) {                                            // uniform over the lane group
        int64_t ut[A];
#pragma unroll
        for (int t = 0; t < A; ++t) {
            const int idx = c8 + slot + t * S;
            ut[t] = GRX_STREAM_LD(col[b + (idx < cnt ? idx : cnt - 1)]);
        }
#pragma unroll
        for (int t = 0; t < A; ++t) xt[t] = *reinterpret_cast<const double2 *>(base + ut[t] * row_stride);
    }
    if (c8) {
        double r0[A], r1[A];
#pragma unroll
        for (int t = 0; t < A; ++t) {
            const int64_t u = GRX_STREAM_LD(col[b + slot + t * S]);
            const double2 x = tr(*reinterpret_cast<const double2 *>(base + u * row_stride));
            r0[t] = x.x; r1[t] = x.y;
        }
        int i = 8;
        if constexpr (A <= 2) {                           // two trips of 8 per iteration: 2A gathers in flight
            for (; i + 8 < c8; i += 16) {
                double2 x[2 * A];
#pragma unroll
                for (int t = 0; t < A; ++t) {
                    const int64_t u0 = GRX_STREAM_LD(col[b + i + slot + t * S]), u1 = GRX_STREAM_LD(col[b + i + 8 + slot + t * S]);
                    x[t] = tr(*reinterpret_cast<const double2 *>(base + u0 * row_stride));
                    x[A + t] = tr(*reinterpret_cast<const double2 *>(base + u1 * row_stride));
                }
#pragma unroll
                for (int t = 0; t < A; ++t) { r0[t] += x[t].x; r1[t] += x[t].y; }
#pragma unroll
                for (int t = 0; t < A; ++t) { r0[t] += x[A + t].x; r1[t] += x[A + t].y; }
            }
        }
        for (; i < c8; i += 8) {
#pragma unroll
            for (int t = 0; t < A; ++t) {
                const int64_t u = GRX_STREAM_LD(col[b + i + slot + t * S]);
                const double2 x = tr(*reinterpret_cast<const double2 *>(base + u * row_stride));
                r0[t] += x.x; r1[t] += x.y;
            }
        }
        // ((r0+r1)+(r2+r3)) + ((r4+r5)+(r6+r7)): residue j = slot + t*S, level `bit` pairs j ^ (1 << bit)
#pragma unroll
        for (int bit = 0; bit < 3; ++bit) {
            if ((1 << bit) < S) {
#pragma unroll
                for (int t = 0; t < A; ++t) {
                    r0[t] += __shfl_xor(r0[t], CL << bit, G);
                    r1[t] += __shfl_xor(r1[t], CL << bit, G);
                }
            } else {
                const int step = (1 << bit) / S;              // distance between partners in r[]
#pragma unroll
                for (int t = 0; t < A; t += 2 * step) {
                    if (t + step < A) { r0[t] += r0[t + step]; r1[t] += r1[t + step]; }
                }
            }
        }
        res0 = r0[0]; res1 = r1[0];
    }
    if (rem) {
        double2 x[A];
#pragma unroll
        for (int t = 0; t < A; ++t) {
            const int idx = c8 + slot + t * S;
            const double2 v = tr(xt[t]);
            x[t] = idx < cnt ? v : make_double2(0.0, 0.0);
        }
#pragma unroll
        for (int i = 0; i < 7; ++i) {
            const int src = (i % S) * CL + part;
            const double v0 = __shfl(x[i / S].x, src, G), v1 = __shfl(x[i / S].y, src, G);
            if (i < rem) { res0 += v0; res1 += v1; }
        }
    }
    o0 = res0; o1 = res1;
}

// Blocks of the rows with more than PW_BLOCK neighbours (grx_aggregate_plan): handled by the same launch
// as the short rows, one lane group per block of 57..128 neighbours -> blk_sums[blk][16]; the
// combine kernel adds them along numpy's recursion afterwards.
struct BlockWork {
    const int32_t *long_rows;
    const int64_t *blk_begin;
    const int32_t *blk_len;
    const int32_t *blk_row;
    int64_t n_blocks;
    double *blk_sums;
};

// VAR: out_sum / out_mean become out_var / out_std -- the sample variance (ddof = 1, pandas' default)
// of the neighbours' values around mean_in (the 'mean' output of a previous launch) and its root.
template <int LDR, int G, bool VAR = false>
__global__ __launch_bounds__(256) void aggregate_kernel(
    const int64_t *__restrict__ row_ptr, const int32_t *__restrict__ col,
    const double *__restrict__ rows, int64_t row_stride, int f, int64_t row_begin, int64_t row_end,
    double *__restrict__ out_sum, double *__restrict__ out_mean, int64_t ld,
    const double *__restrict__ mean_in, BlockWork bw)
{
    constexpr int CL = (LDR >= 16 ? 16 : LDR) / 2;
    const int lane = threadIdx.x % G;
    const int part = lane % CL, slot = lane / CL;
    const int64_t group = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / G;
    const int64_t ngroups = (int64_t)gridDim.x * blockDim.x / G;
    // the long rows' blocks first (the longest work items of the launch), then the short rows
    for (int64_t k = group; k < bw.n_blocks; k += ngroups) {
        const int64_t v = bw.long_rows[bw.blk_row[k]];
        if (v < row_begin || v >= row_end) continue;
        double a0, a1;
        if (VAR) {
            const int c0 = 2 * part, c1 = 2 * part + 1;
            const double m0 = c0 < f ? mean_in[(int64_t)c0 * ld + v] : 0.0;
            const double m1 = c1 < f ? mean_in[(int64_t)c1 * ld + v] : 0.0;
            pairwise_segment<LDR, G, true>(col, rows, row_stride, bw.blk_begin[k], bw.blk_len[k], part, slot, a0, a1, m0, m1);
        } else {
            pairwise_segment<LDR, G>(col, rows, row_stride, bw.blk_begin[k], bw.blk_len[k], part, slot, a0, a1);
        }
        if (slot == 0) {
            bw.blk_sums[k * 16 + 2 * part] = a0;
            bw.blk_sums[k * 16 + 2 * part + 1] = a1;
        }
    }
    for (int64_t v = row_begin + group; v < row_end; v += ngroups) {
        const int64_t b = row_ptr[v], e = row_ptr[v + 1];
        const int64_t d = e - b;
        if (d > PW_BLOCK) continue;                       // the block loop above + aggregate_combine_kernel
        double a0, a1;
        const int c0 = 2 * part, c1 = 2 * part + 1;
        if (VAR) {
            const double m0 = c0 < f ? mean_in[(int64_t)c0 * ld + v] : 0.0;
            const double m1 = c1 < f ? mean_in[(int64_t)c1 * ld + v] : 0.0;
            pairwise_segment<LDR, G, true>(col, rows, row_stride, b, (int)d, part, slot, a0, a1, m0, m1);
        } else {
            pairwise_segment<LDR, G>(col, rows, row_stride, b, (int)d, part, slot, a0, a1);
        }
        if (slot == 0) {
            const double cnt = (double)d;
            if (VAR) {
                // count - ddof <= 0 -> NaN -> fillna(0) (extract.py:113)
                const double v0 = (d > 1) ? a0 / (cnt - 1.0) : 0.0, v1 = (d > 1) ? a1 / (cnt - 1.0) : 0.0;
                if (c0 < f) {
                    if (out_sum) out_sum[(int64_t)c0 * ld + v] = v0;
                    if (out_mean) out_mean[(int64_t)c0 * ld + v] = sqrt(v0);
                }
                if (c1 < f) {
                    if (out_sum) out_sum[(int64_t)c1 * ld + v] = v1;
                    if (out_mean) out_mean[(int64_t)c1 * ld + v] = sqrt(v1);
                }
            } else {
                if (c0 < f) {
                    if (out_sum) GRX_STREAM_ST(out_sum[(int64_t)c0 * ld + v], a0);
                    if (out_mean) GRX_STREAM_ST(out_mean[(int64_t)c0 * ld + v], (d > 0) ? a0 / cnt : 0.0);
                }
                if (c1 < f) {
                    if (out_sum) GRX_STREAM_ST(out_sum[(int64_t)c1 * ld + v], a1);
                    if (out_mean) GRX_STREAM_ST(out_mean[(int64_t)c1 * ld + v], (d > 0) ? a1 / cnt : 0.0);
                }
            }
        }
    }
}

// Sixteen lanes per long row (lane c = column c): add the block sums along numpy's recursion
//   sum(n) = n <= 128 ? block : sum(n2) + sum(n - n2),  n2 = n/2 - (n/2) % 8
// chunk by chunk (8192 neighbours), running total over the chunks.  The recursion is flattened by the
// host into its post-order program: blk_ops[i] & 0x7F = number of pending additions after pushing
// block i, bit 7 = last block of a chunk.  Every lane runs the stack machine of its column on a
// private LDS stack (no barriers, no staging: the block sums of a row are read once, 128 bytes per
// block and group, sixteen blocks in flight at a time).
// (First version: one 64-lane workgroup per row with the block sums staged in LDS -- 42 us per launch
// for the 6.7 k long rows of BA 1 M, a chain of five dependent round trips per workgroup.)
constexpr int PW_MAX_DEPTH = 12;

__global__ __launch_bounds__(256) void aggregate_combine_kernel(
    const int64_t *__restrict__ row_ptr, int f, int64_t row_begin, int64_t row_end,
    const int32_t *__restrict__ long_rows, const int64_t *__restrict__ blk_ptr, int64_t n_long,
    const uint8_t *__restrict__ blk_ops, const double *__restrict__ blk_sums, double *__restrict__ out_sum,
    double *__restrict__ out_mean, int64_t ld, int var_mode)
{
    __shared__ double stk[16][PW_MAX_DEPTH][16];
    const int c = threadIdx.x & 15, grp = threadIdx.x >> 4;
    const int64_t gstride = (int64_t)gridDim.x * 16;
    for (int64_t h = (int64_t)blockIdx.x * 16 + grp; h < n_long; h += gstride) {
        const int64_t v = long_rows[h];
        if (v < row_begin || v >= row_end) continue;               // uniform over the 16 lanes
        const int64_t n = row_ptr[v + 1] - row_ptr[v];
        const int64_t leaf_end = blk_ptr[h + 1];
        int64_t leaf = blk_ptr[h];
        double total = 0.0;
        int sp = 0;
        // batches of 16 blocks: the loads of a batch are independent (hubs have ~100 blocks -- one load
        // at a time would be a 100-deep latency chain), the folding is sequential
        while (leaf < leaf_end) {
            const int m = (int)((leaf_end - leaf) < 16 ? (leaf_end - leaf) : 16);
            double buf[16];
            int ops[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                buf[j] = (j < m) ? blk_sums[(leaf + j) * 16 + c] : 0.0;
                ops[j] = (j < m) ? (int)blk_ops[leaf + j] : 0;
            }
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                if (j < m) {
                    double val = buf[j];
                    for (int k = ops[j] & 0x7F; k > 0; --k) val = stk[grp][--sp][c] + val;   // left + right
                    if (ops[j] & 0x80) total += val;             // chunk complete (sp == 0 here)
                    else stk[grp][sp++][c] = val;
                }
            }
            leaf += m;
        }
        if (c < f) {
            if (var_mode) {                                     // long rows have n > 128 >= 2
                const double var = total / ((double)n - 1.0);
                if (out_sum) out_sum[(int64_t)c * ld + v] = var;
                if (out_mean) out_mean[(int64_t)c * ld + v] = sqrt(var);
            } else {
                if (out_sum) out_sum[(int64_t)c * ld + v] = total;
                if (out_mean) out_mean[(int64_t)c * ld + v] = total / (double)n;
            }
        }
    }
}


// ---------------------------------------------------------------------------------------
// neighbour aggregation of INTEGER rows (generation 1 of unweighted graphs)
// ---------------------------------------------------------------------------------------
// When every source column holds exact non-negative integers below 2^31 (degrees, ego-net edge counts: the whole
// generation-0 block of an unweighted graph) the sums are integers below 2^53, so ANY order of additions gives the
// bits numpy's pairwise tree gives -- and the gather source shrinks from 8 to 4 bytes per column: three columns
// are a 16-byte row, four rows per 64-byte request line, a quarter of the table a 4 MiB L2 has to hold (the hit
// rate of the gather follows sqrt(rows that fit / N) on a power-law graph).  Lane = (slot,
// part): CL = LDI / 4 adjacent lanes fetch one neighbour row (int4 each), slot s takes neighbours s, s + S, ...;
// int64 accumulators, xor-butterfly over the slots, mean = double(sum) / count like the fp64 kernel.
// Rows with more than 128 neighbours reuse the block list of the plan: one lane group per block -> int64 partial
// sums (the blk_sums scratch), added per row by aggregate_i32_combine_kernel.
__global__ __launch_bounds__(256) void pack_rows_i32_kernel(int64_t n, int f, int ldi, GrxPtrTable cols_tab,
                                                            int32_t *__restrict__ rows)
{
    const double *const *cols = reinterpret_cast<const double *const *>(cols_tab.p);
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        int32_t *dst = rows + i * ldi;
        for (int c = 0; c < f; ++c) dst[c] = (int32_t)cols[c][i];
        for (int c = f; c < ldi; ++c) dst[c] = 0;
    }
}

template <int LDI, int G>
__device__ __forceinline__ void i32_segment(const int32_t *__restrict__ col, const int32_t *__restrict__ rows, int64_t b,
                                            int cnt, int part, int slot, long long (&acc)[4])
{
    constexpr int CL = LDI / 4, S = G / CL;
    const int32_t *base = rows + 4 * part;
    acc[0] = acc[1] = acc[2] = acc[3] = 0;
    // four neighbours of this slot per trip: indices first (clamped), then the four row loads, all independent
    for (int k0 = slot; k0 < cnt; k0 += 4 * S) {
        int64_t u[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int k = k0 + t * S;
            u[t] = GRX_STREAM_LD(col[b + (k < cnt ? k : cnt - 1)]);
        }
        int4 x[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) x[t] = *reinterpret_cast<const int4 *>(base + u[t] * LDI);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            if (k0 + t * S < cnt) { acc[0] += x[t].x; acc[1] += x[t].y; acc[2] += x[t].z; acc[3] += x[t].w; }
        }
    }
#pragma unroll
    for (int off = CL; off < G; off <<= 1) {
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] += __shfl_xor(acc[j], off, G);
    }
}

template <int LDI, int G>
__global__ __launch_bounds__(256) void aggregate_i32_kernel(
    const int64_t *__restrict__ row_ptr, const int32_t *__restrict__ col, const int32_t *__restrict__ rows, int f,
    int64_t row_begin, int64_t row_end, double *__restrict__ out_sum, double *__restrict__ out_mean, int64_t ld,
    BlockWork bw)
{
    constexpr int CL = LDI / 4;
    const int lane = threadIdx.x % G;
    const int part = lane % CL, slot = lane / CL;
    const int64_t group = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / G;
    const int64_t ngroups = (int64_t)gridDim.x * blockDim.x / G;
    long long acc[4];
    long long *blk = reinterpret_cast<long long *>(bw.blk_sums);
    for (int64_t k = group; k < bw.n_blocks; k += ngroups) {
        const int64_t v = bw.long_rows[bw.blk_row[k]];
        if (v < row_begin || v >= row_end) continue;
        i32_segment<LDI, G>(col, rows, bw.blk_begin[k], bw.blk_len[k], part, slot, acc);
        if (slot == 0) {
#pragma unroll
            for (int j = 0; j < 4; ++j) blk[k * 16 + 4 * part + j] = acc[j];
        }
    }
    for (int64_t v = row_begin + group; v < row_end; v += ngroups) {
        const int64_t b = row_ptr[v], e = row_ptr[v + 1];
        const int64_t d = e - b;
        if (d > PW_BLOCK) continue;
        i32_segment<LDI, G>(col, rows, b, (int)d, part, slot, acc);
        if (slot == 0) {
            const double cnt = (double)d;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int c = 4 * part + j;
                if (c < f) {
                    const double sum = (double)acc[j];
                    if (out_sum) GRX_STREAM_ST(out_sum[(int64_t)c * ld + v], sum);
                    if (out_mean) GRX_STREAM_ST(out_mean[(int64_t)c * ld + v], (d > 0) ? sum / cnt : 0.0);
                }
            }
        }
    }
}

// sixteen lanes per long row (lane c = column c): integer partial sums of its blocks, any order
__global__ __launch_bounds__(256) void aggregate_i32_combine_kernel(
    const int64_t *__restrict__ row_ptr, int f, int64_t row_begin, int64_t row_end, const int32_t *__restrict__ long_rows,
    const int64_t *__restrict__ blk_ptr, int64_t n_long, const double *__restrict__ blk_sums, double *__restrict__ out_sum,
    double *__restrict__ out_mean, int64_t ld)
{
    const long long *blk = reinterpret_cast<const long long *>(blk_sums);
    const int c = threadIdx.x & 15, grp = threadIdx.x >> 4;
    const int64_t gstride = (int64_t)gridDim.x * 16;
    for (int64_t h = (int64_t)blockIdx.x * 16 + grp; h < n_long; h += gstride) {
        const int64_t v = long_rows[h];
        if (v < row_begin || v >= row_end) continue;
        const int64_t n = row_ptr[v + 1] - row_ptr[v];
        long long total = 0;
        // sixteen block sums in flight at a time (hubs have ~100 blocks: one load at a time is a 100-deep latency chain)
        const int64_t kb = blk_ptr[h], ke = blk_ptr[h + 1];
        for (int64_t k0 = kb; k0 < ke; k0 += 16) {
            long long part[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) part[j] = blk[(k0 + j < ke ? k0 + j : ke - 1) * 16 + c];
#pragma unroll
            for (int j = 0; j < 16; ++j) total += k0 + j < ke ? part[j] : 0;
        }
        if (c < f) {
            const double sum = (double)total;
            if (out_sum) out_sum[(int64_t)c * ld + v] = sum;
            if (out_mean) out_mean[(int64_t)c * ld + v] = sum / (double)n;
        }
    }
}

// ---------------------------------------------------------------------------------------
// neighbour aggregation from BIT-PACKED integer rows (generations 1 and 2 of unweighted graphs)
// ---------------------------------------------------------------------------------------
// The gather is bound by the chip's REQUEST rate, and how many requests miss an XCD's 4 MiB L2 is set by the bytes of
// the gather table (profiles/r04_gather_bw.json: 16-, 32- and 64-byte rows gather at the same rows/s; a 64 MB table
// at 62 G rows/s, a 16 MB one at 92, an L2-resident one at 250).  So the table is made as small as exactness allows:
//   * every summand the reference adds in generation g is either an exact integer S (a degree / ego-net count of
//     generation 0, or a neighbour SUM of such a column) or the mean fl(S / d) of one over d neighbours;
//   * the row of node u therefore only has to carry the integers S_k(u) of the distinct base columns and d(u), each in
//     as many bits as the column's maximum needs (grx_column_bits) -- 8 or 16 bytes instead of 16 / 64;
//   * the kernel rebuilds every summand in registers -- double(S), or double(S) / double(d) with the same correctly
//     rounded division that produced the stored mean -- and adds them in numpy's pairwise order exactly like
//     aggregate_kernel does (for integer summands any order gives the same bits; one code path serves both).
// Lane = slot: G = 8 lanes per output row, lane s owns the strided accumulator r[s] of EVERY output column (a whole
// neighbour row is one 8- / 16-byte load of one lane).  Rows longer than 128 neighbours reuse the plan's block list and
// aggregate_combine_kernel unchanged.
struct PackedDesc {
    int n_out;
    uint8_t word[8], shift[8], bits[8], is_mean[8];    // per output: where its source field sits
    uint8_t d_word, d_shift, d_bits;                     // the neighbour-count field (d_bits = 0: none)
};

template <int WORDS>
struct PackedRow { unsigned long long w[WORDS]; };

template <int WORDS>
__device__ __forceinline__ PackedRow<WORDS> packed_load(const unsigned long long *__restrict__ rows, int64_t u)
{
    PackedRow<WORDS> r;
    if constexpr (WORDS == 1) {
        r.w[0] = rows[u];
    } else {
        const ulonglong2 v = *reinterpret_cast<const ulonglong2 *>(rows + 2 * u);
        r.w[0] = v.x; r.w[1] = v.y;
    }
    return r;
}

template <int WORDS>
__device__ __forceinline__ unsigned long long packed_field(const PackedRow<WORDS> &r, int word, int shift, int bits)
{
    const unsigned long long w = (WORDS == 2 && word) ? r.w[WORDS - 1] : r.w[0];
    return bits >= 64 ? w : ((w >> shift) & ((1ull << bits) - 1ull));
}

// the F summands of one neighbour row
// Means: fl(S / d) for F columns that share the divisor.  The compiler's fp64 division is, for operands that need no
// scaling (here 0 <= S < 2^53, 1 <= d < 2^31), exactly: y = rcp(d) refined by two Newton steps, q = S * y,
// r = fma(-d, q, S), q' = fma(r, y, q) -- correctly rounded.  The reciprocal and its refinement do not depend on S, so
// they are done ONCE per neighbour row and every column pays three instructions instead of a whole division; the
// result is the same correctly rounded quotient (tests/test_gpu_packed.py compares ~10^8 quotients with the divisions
// aggregate_kernel's sources were produced by).
template <int WORDS, int F>
__device__ __forceinline__ void packed_values(const PackedRow<WORDS> &r, const PackedDesc &d, double (&x)[F])
{
#pragma clang fp contract(off)
    double cnt = 1.0, y = 0.0;
    if (d.d_bits) {
        cnt = (double)(int)packed_field<WORDS>(r, d.d_word, d.d_shift, d.d_bits);       // < 2^31 (place_fields)
        y = __builtin_amdgcn_rcp(cnt);
        double e = __builtin_fma(-cnt, y, 1.0);
        y = __builtin_fma(y, e, y);
        e = __builtin_fma(-cnt, y, 1.0);
        y = __builtin_fma(y, e, y);
    }
#pragma unroll
    for (int j = 0; j < F; ++j) {
        const unsigned long long f = packed_field<WORDS>(r, d.word[j], d.shift[j], d.bits[j]);
        const double s = d.bits[j] <= 31 ? (double)(int)f : (double)(long long)f;
        if (d.is_mean[j]) {
            // the stored mean of a node without neighbours is 0 (NaN -> 0, extract.py:113)
            const double q = s * y;
            const double rem = __builtin_fma(-cnt, q, s);
            x[j] = cnt > 0.0 ? __builtin_fma(rem, y, q) : 0.0;
        } else {
            x[j] = s;
        }
    }
}

// numpy's pairwise sum of one segment of cnt <= 128 neighbours for F columns by S lanes (slot = lane % S), A = 8 / S of the
// eight strided accumulators per lane (residue j = slot + t * S):
//   r[j] = x[j] + x[j+8] + ...;  ((r0+r1)+(r2+r3)) + ((r4+r5)+(r6+r7));  then the cnt % 8 trailing elements one by one
// Fewer lanes per row = more independent gathers per lane and trip (A of them) and more rows per wavefront.
template <int WORDS, int F, int S>
__device__ __forceinline__ void packed_segment(const int32_t *__restrict__ col, const unsigned long long *__restrict__ rows,
                                               const PackedDesc &d, int64_t b, int cnt, int slot, double (&res)[F])
{
#pragma clang fp contract(off)
    constexpr int A = 8 / S;
    static_assert(S * A == 8, "S must be 1, 2, 4 or 8");
    const int c8 = cnt & ~7, rem = cnt - c8;
#pragma unroll
    for (int j = 0; j < F; ++j) res[j] = 0.0;
    // the trailing neighbours of this lane: indices and rows are requested first, used last
    PackedRow<WORDS> tail[A];
    if (rem) {
        int64_t ut[A];
#pragma unroll
        for (int t = 0; t < A; ++t) {
            const int idx = c8 + slot + t * S;
            ut[t] = GRX_STREAM_LD(col[b + (idx < cnt ? idx : cnt - 1)]);
        }
#pragma unroll
        for (int t = 0; t < A; ++t) tail[t] = packed_load<WORDS>(rows, ut[t]);
    }
    if (c8) {
        double r[A][F];
        {
            int64_t u[A];
#pragma unroll
            for (int t = 0; t < A; ++t) u[t] = GRX_STREAM_LD(col[b + slot + t * S]);
            PackedRow<WORDS> pr[A];
#pragma unroll
            for (int t = 0; t < A; ++t) pr[t] = packed_load<WORDS>(rows, u[t]);
#pragma unroll
            for (int t = 0; t < A; ++t) packed_values<WORDS, F>(pr[t], d, r[t]);
        }
        int i = 8;
        if constexpr (A <= 2) {                               // two trips of 8 per iteration: 2 A gathers in flight per lane
            for (; i + 8 < c8; i += 16) {
                int64_t u[2 * A];
#pragma unroll
                for (int t = 0; t < A; ++t) {
                    u[t] = GRX_STREAM_LD(col[b + i + slot + t * S]);
                    u[A + t] = GRX_STREAM_LD(col[b + i + 8 + slot + t * S]);
                }
                PackedRow<WORDS> pr[2 * A];
#pragma unroll
                for (int t = 0; t < 2 * A; ++t) pr[t] = packed_load<WORDS>(rows, u[t]);
#pragma unroll
                for (int h = 0; h < 2; ++h) {
#pragma unroll
                    for (int t = 0; t < A; ++t) {
                        double x[F];
                        packed_values<WORDS, F>(pr[h * A + t], d, x);
#pragma unroll
                        for (int j = 0; j < F; ++j) r[t][j] += x[j];
                    }
                }
            }
        }
        for (; i < c8; i += 8) {
            int64_t u[A];
#pragma unroll
            for (int t = 0; t < A; ++t) u[t] = GRX_STREAM_LD(col[b + i + slot + t * S]);
            PackedRow<WORDS> pr[A];
#pragma unroll
            for (int t = 0; t < A; ++t) pr[t] = packed_load<WORDS>(rows, u[t]);
#pragma unroll
            for (int t = 0; t < A; ++t) {
                double x[F];
                packed_values<WORDS, F>(pr[t], d, x);
#pragma unroll
                for (int j = 0; j < F; ++j) r[t][j] += x[j];
            }
        }
        // ((r0+r1)+(r2+r3)) + ((r4+r5)+(r6+r7)): residue j = slot + t*S, level `bit` pairs j ^ (1 << bit)
#pragma unroll
        for (int bit = 0; bit < 3; ++bit) {
            if ((1 << bit) < S) {
#pragma unroll
                for (int t = 0; t < A; ++t)
#pragma unroll
                    for (int j = 0; j < F; ++j) r[t][j] += __shfl_xor(r[t][j], 1 << bit, S);
            } else {
                const int step = (1 << bit) / S;              // distance between partners in r[]
#pragma unroll
                for (int t = 0; t < A; t += 2 * step) {
                    if (t + step < A) {
#pragma unroll
                        for (int j = 0; j < F; ++j) r[t][j] += r[t + step][j];
                    }
                }
            }
        }
#pragma unroll
        for (int j = 0; j < F; ++j) res[j] = r[0][j];
    }
    if (rem) {
        double x[A][F];
#pragma unroll
        for (int t = 0; t < A; ++t) packed_values<WORDS, F>(tail[t], d, x[t]);
#pragma unroll
        for (int i = 0; i < 7; ++i) {
#pragma unroll
            for (int j = 0; j < F; ++j) {
                const double v = S > 1 ? __shfl(x[i / S][j], i % S, S) : x[i / S][j];
                if (i < rem) res[j] += v;
            }
        }
    }
}

template <int WORDS, int F, int S>
__global__ __launch_bounds__(256) void aggregate_packed_kernel(
    const int64_t *__restrict__ row_ptr, const int32_t *__restrict__ col, const unsigned long long *__restrict__ rows,
    PackedDesc d, int64_t row_begin, int64_t row_end, double *__restrict__ out_sum, double *__restrict__ out_mean,
    int64_t ld, BlockWork bw)
{
    const int slot = threadIdx.x % S;
    const int64_t group = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / S;
    const int64_t ngroups = (int64_t)gridDim.x * blockDim.x / S;
    double a[F];
    for (int64_t k = group; k < bw.n_blocks; k += ngroups) {
        const int64_t v = bw.long_rows[bw.blk_row[k]];
        if (v < row_begin || v >= row_end) continue;
        packed_segment<WORDS, F, S>(col, rows, d, bw.blk_begin[k], bw.blk_len[k], slot, a);
        if (slot == 0) {
#pragma unroll
            for (int j = 0; j < F; ++j) bw.blk_sums[k * 16 + j] = a[j];
        }
    }
    for (int64_t v = row_begin + group; v < row_end; v += ngroups) {
        const int64_t b = row_ptr[v], e = row_ptr[v + 1];
        const int64_t cntl = e - b;
        if (cntl > PW_BLOCK) continue;                     // the block loop above + aggregate_combine_kernel
        packed_segment<WORDS, F, S>(col, rows, d, b, (int)cntl, slot, a);
        // every lane of the group holds the totals: lane s stores columns s, s + S, ...
        const double cnt = (double)cntl;
#pragma unroll
        for (int j = 0; j < F; ++j) {
            if (slot == j % S) {
                if (out_sum) GRX_STREAM_ST(out_sum[(int64_t)j * ld + v], a[j]);
                if (out_mean) GRX_STREAM_ST(out_mean[(int64_t)j * ld + v], (cntl > 0) ? a[j] / cnt : 0.0);
            }
        }
    }
}

// bit-packed gather source: row u = the fields' integers of node u (+ its neighbour count), see PackedDesc
struct PackFieldsArgs {
    const double *src[8];
    uint8_t word[8], shift[8];
    int n_fields;
    uint8_t d_word, d_shift, d_bits;
};

template <int WORDS>
__global__ __launch_bounds__(256) void pack_fields_kernel(int64_t n, PackFieldsArgs a, const int64_t *__restrict__ row_ptr,
                                                          unsigned long long *__restrict__ rows)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        unsigned long long w[2] = {0ull, 0ull};
        for (int k = 0; k < a.n_fields; ++k)
            w[a.word[k]] |= (unsigned long long)(long long)a.src[k][i] << a.shift[k];
        if (a.d_bits) w[a.d_word] |= (unsigned long long)(row_ptr[i + 1] - row_ptr[i]) << a.d_shift;
        if constexpr (WORDS == 1) rows[i] = w[0];
        else *reinterpret_cast<ulonglong2 *>(rows + 2 * i) = make_ulonglong2(w[0], w[1]);
    }
}

// bits[c] = max(bits[c], number of bits of max(column c over the workgroup's slice of rows [rb, re))) for the columns
// flagged in `mask` (exact non-negative integers by construction); bits[] starts at 0.  The width is monotone in the
// value, so the maximum over the slices' widths is the width of the column maximum -- and it can be max-reduced over
// ranks as a 32-bit integer.  grid = (row slices, columns).
__global__ __launch_bounds__(256) void column_bits_kernel(const double *__restrict__ block, int64_t ld, int64_t rb, int64_t re,
                                                          unsigned long long mask, int32_t *__restrict__ bits)
{
    const int c = blockIdx.y;
    if (!((mask >> c) & 1ull)) return;
    const double *x = block + (int64_t)c * ld;
    double m = 0.0;
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = rb + (int64_t)blockIdx.x * 256 + threadIdx.x; i < re; i += stride) m = fmax(m, x[i]);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmax(m, __shfl_xor(m, off, 64));
    __shared__ double part[4];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = fmax(fmax(part[0], part[1]), fmax(part[2], part[3]));
        // a column whose maximum is not below 2^62 (or not finite) cannot be packed: 64 says so
        const unsigned long long v = (m >= 0.0 && m < 4.6e18) ? (unsigned long long)m : ~0ull;
        atomicMax(bits + c, v ? 64 - __clzll((long long)v) : 1);
    }
}

// product over the neighbours (agg 'prod'): np.multiply.reduce is a plain left-to-right product, so
// one lane per (row, column) multiplies in adjacency order; the empty product is 1.  Not a tuned
// kernel: the lanes of a row read adjacent doubles of each neighbour row, nothing more.
__global__ __launch_bounds__(256) void aggregate_prod_kernel(
    const int64_t *__restrict__ row_ptr, const int32_t *__restrict__ col, const double *__restrict__ rows,
    int64_t row_stride, int f, int64_t row_begin, int64_t row_end, double *__restrict__ out, int64_t ld)
{
    const int64_t total = (row_end - row_begin) * f;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int64_t v = row_begin + i / f;
        const int c = (int)(i % f);
        double p = 1.0;
        for (int64_t k = row_ptr[v]; k < row_ptr[v + 1]; ++k) p *= rows[(int64_t)col[k] * row_stride + c];
        out[(int64_t)c * ld + v] = p;
    }
}

// min / max over the neighbours (aggs 'min', 'max' of features/extract.py:36-47); order-free.
template <int LDR, int G>
__global__ __launch_bounds__(256) void aggregate_minmax_kernel(
    const int64_t *__restrict__ row_ptr, const int32_t *__restrict__ col,
    const double *__restrict__ rows, int64_t row_stride, int f, int64_t row_begin, int64_t row_end,
    double *__restrict__ out_min, double *__restrict__ out_max, int64_t ld)
{
    constexpr int CL = (LDR >= 16 ? 16 : LDR) / 2;
    constexpr int S = G / CL;
    const int lane = threadIdx.x % G;
    const int part = lane % CL, slot = lane / CL;
    const int64_t group = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / G;
    const int64_t ngroups = (int64_t)gridDim.x * blockDim.x / G;
    const double inf = __builtin_huge_val();
    for (int64_t v = row_begin + group; v < row_end; v += ngroups) {
        const int64_t b = row_ptr[v], e = row_ptr[v + 1];
        double lo0 = inf, lo1 = inf, hi0 = -inf, hi1 = -inf;
        for (int64_t k = b + slot; k < e; k += S) {
            const int64_t u = col[k];
            const double2 x = *reinterpret_cast<const double2 *>(rows + u * row_stride + 2 * part);
            lo0 = fmin(lo0, x.x); hi0 = fmax(hi0, x.x);
            lo1 = fmin(lo1, x.y); hi1 = fmax(hi1, x.y);
        }
#pragma unroll
        for (int off = CL; off < G; off <<= 1) {
            lo0 = fmin(lo0, __shfl_xor(lo0, off, G)); hi0 = fmax(hi0, __shfl_xor(hi0, off, G));
            lo1 = fmin(lo1, __shfl_xor(lo1, off, G)); hi1 = fmax(hi1, __shfl_xor(hi1, off, G));
        }
        if (slot == 0) {
            const bool any = e > b;                       // no neighbours -> NaN -> fillna(0) (:113)
            const int c0 = 2 * part, c1 = 2 * part + 1;
            if (c0 < f) {
                if (out_min) out_min[(int64_t)c0 * ld + v] = any ? lo0 : 0.0;
                if (out_max) out_max[(int64_t)c0 * ld + v] = any ? hi0 : 0.0;
            }
            if (c1 < f) {
                if (out_min) out_min[(int64_t)c1 * ld + v] = any ? lo1 : 0.0;
                if (out_max) out_max[(int64_t)c1 * ld + v] = any ? hi1 : 0.0;
            }
        }
    }
}

// Per-graph preprocessing of grx_aggregate: lane-group width and the block list of the long rows.
}  // namespace

struct grx_aggregate_plan {
    int64_t n = 0;
    int lanes_per_row = 8;
    int64_t n_long = 0, n_blocks = 0;
    int64_t max_degree = 0;             // longest row (bounds the integer sums of grx_aggregate_i32)
    int32_t *d_long_rows = nullptr;     // [n_long] ascending
    int64_t *d_blk_ptr = nullptr;       // [n_long + 1]
    int64_t *d_blk_begin = nullptr;     // [n_blocks] position in d_col
    int32_t *d_blk_len = nullptr;       // [n_blocks]
    int32_t *d_blk_row = nullptr;       // [n_blocks] index into d_long_rows
    uint8_t *d_blk_ops = nullptr;       // [n_blocks] post-order program of the combine step
    double *d_blk_sums = nullptr;       // [n_blocks * 16] scratch
};

namespace {

// Blocks of numpy's recursion over [begin, begin+n) in order, with the post-order program of the
// additions: ops[i] = how many times "pop left, add" runs after block i has been pushed.
void pairwise_blocks(int64_t begin, int64_t n, std::vector<int64_t> &b, std::vector<int32_t> &len,
                     std::vector<uint8_t> &ops)
{
    if (n <= PW_BLOCK) { b.push_back(begin); len.push_back((int32_t)n); ops.push_back(0); return; }
    int64_t n2 = n / 2;
    n2 -= n2 % 8;
    pairwise_blocks(begin, n2, b, len, ops);
    pairwise_blocks(begin + n2, n - n2, b, len, ops);
    ++ops.back();
}

template <int LDR, int G>
int launch_aggregate_g(const grx_aggregate_plan *p, const int64_t *row_ptr, const int32_t *col, const double *rows,
                       int64_t row_stride, int f, int64_t rb, int64_t re, double *s, double *m, int64_t ld,
                       hipStream_t st, const double *mean_in = nullptr)
{
    const int64_t nrows = re - rb;
    const int64_t want = grx_ceil_div(nrows * G, 256);
    const int grid = (int)(want < 1 ? 1 : (want > GRX_NUM_CU * 32 ? GRX_NUM_CU * 32 : want));
    const BlockWork bw{p->d_long_rows, p->d_blk_begin, p->d_blk_len, p->d_blk_row, p->n_long > 0 ? p->n_blocks : 0,
                       p->d_blk_sums};
    {
        GRX_PROF(GRX_K_AGGREGATE, st);
        if (mean_in) aggregate_kernel<LDR, G, true><<<grid, 256, 0, st>>>(row_ptr, col, rows, row_stride, f, rb, re, s, m, ld, mean_in, bw);
        else aggregate_kernel<LDR, G><<<grid, 256, 0, st>>>(row_ptr, col, rows, row_stride, f, rb, re, s, m, ld, nullptr, bw);
    }
    GRX_LAUNCH_CHECK();
    if (p->n_long > 0) {
        GRX_PROF(GRX_K_AGGREGATE_HUB, st);
        const int64_t cwant = grx_ceil_div(p->n_long, 16);
        aggregate_combine_kernel<<<(unsigned)(cwant > GRX_NUM_CU * 32 ? GRX_NUM_CU * 32 : cwant), 256, 0, st>>>(
            row_ptr, f, rb, re, p->d_long_rows, p->d_blk_ptr, p->n_long, p->d_blk_ops, p->d_blk_sums, s, m, ld,
            mean_in ? 1 : 0);
    }
    GRX_LAUNCH_CHECK();
    return GRX_OK;
}

template <int LDR, int G>
int launch_minmax_g(const int64_t *row_ptr, const int32_t *col, const double *rows, int64_t row_stride, int f,
                    int64_t rb, int64_t re, double *lo, double *hi, int64_t ld, hipStream_t st)
{
    const int64_t want = grx_ceil_div((re - rb) * G, 256);
    const int grid = (int)(want < 1 ? 1 : (want > GRX_NUM_CU * 32 ? GRX_NUM_CU * 32 : want));
    {
        GRX_PROF(GRX_K_AGGREGATE, st);
        aggregate_minmax_kernel<LDR, G><<<grid, 256, 0, st>>>(row_ptr, col, rows, row_stride, f, rb, re, lo, hi, ld);
    }
    GRX_LAUNCH_CHECK();
    return GRX_OK;
}

// lanes per row G for a row stride: S = G / (LDR/2) must be 1, 2, 4 or 8
template <int LDR>
int launch_aggregate(bool minmax, const grx_aggregate_plan *p, const int64_t *row_ptr, const int32_t *col,
                     const double *rows, int64_t row_stride, int f, int64_t rb, int64_t re, double *a, double *b,
                     int64_t ld, hipStream_t st, const double *mean_in = nullptr)
{
    constexpr int CL = (LDR >= 16 ? 16 : LDR) / 2;
    int G = p->lanes_per_row;
    if (G < CL) G = CL;
    if (G > 8 * CL) G = 8 * CL;
    if (G < 4) G = 4;
#define GRX_AGG_CASE(GG)                                                                                              \
    case GG:                                                                                                          \
        if constexpr (GG >= CL && GG <= 8 * CL)                                                                       \
            return minmax ? launch_minmax_g<LDR, GG>(row_ptr, col, rows, row_stride, f, rb, re, a, b, ld, st)         \
                          : launch_aggregate_g<LDR, GG>(p, row_ptr, col, rows, row_stride, f, rb, re, a, b, ld, st,   \
                                                        mean_in);                                                     \
        break;
    switch (G) {
        GRX_AGG_CASE(4)
        GRX_AGG_CASE(8)
        GRX_AGG_CASE(16)
        GRX_AGG_CASE(32)
    default: break;
    }
#undef GRX_AGG_CASE
    grx_set_error("grx_aggregate: no kernel for ldr=%d lanes_per_row=%d", LDR, G);
    return GRX_ERR_UNSUPPORTED;
}

int aggregate_dispatch(bool minmax, const grx_aggregate_plan *plan, const int64_t *d_row_ptr, const int32_t *d_col,
                       int f, const double *d_rows, int ldr, int64_t row_begin, int64_t row_end, double *d_a,
                       double *d_b, int64_t ld, void *stream, const double *d_mean_in = nullptr)
{
    const char *who = minmax ? "grx_aggregate_minmax" : (d_mean_in ? "grx_aggregate_var" : "grx_aggregate");
    GRX_REQUIRE(plan != nullptr, "%s: NULL plan (grx_aggregate_plan_create)", who);
    const int64_t n = plan->n;
    GRX_REQUIRE(row_begin >= 0 && row_begin <= row_end && row_end <= n, "%s: bad row range", who);
    GRX_REQUIRE(f >= 0 && ldr >= f, "%s: ldr=%d < f=%d", who, ldr, f);
    GRX_REQUIRE(ldr == 2 || ldr == 4 || ldr == 8 || (ldr >= 16 && ldr % 16 == 0),
                "%s: ldr=%d must be 2, 4, 8 or a multiple of 16 (use grx_aggregate_ldr)", who, ldr);
    if (row_end == row_begin || f == 0) return GRX_OK;
    GRX_REQUIRE(d_row_ptr && d_col && d_rows, "%s: NULL pointer", who);
    GRX_REQUIRE(ld >= n, "%s: ld < n", who);
    GRX_REQUIRE((reinterpret_cast<uintptr_t>(d_rows) & 127) == 0, "%s: d_rows must be 128-byte aligned", who);
    hipStream_t st = grx_stream(stream);
    if (ldr < 16)
        switch (ldr) {
        case 2:  return launch_aggregate<2>(minmax, plan, d_row_ptr, d_col, d_rows, ldr, f, row_begin, row_end, d_a, d_b, ld, st, d_mean_in);
        case 4:  return launch_aggregate<4>(minmax, plan, d_row_ptr, d_col, d_rows, ldr, f, row_begin, row_end, d_a, d_b, ld, st, d_mean_in);
        default: return launch_aggregate<8>(minmax, plan, d_row_ptr, d_col, d_rows, ldr, f, row_begin, row_end, d_a, d_b, ld, st, d_mean_in);
        }
    // wide rows: 16 columns (one 128-byte segment of every row) per launch
    for (int c0 = 0; c0 < f; c0 += 16) {
        const int fc = (f - c0 < 16) ? (f - c0) : 16;
        double *a = d_a ? d_a + (int64_t)c0 * ld : nullptr;
        double *b = d_b ? d_b + (int64_t)c0 * ld : nullptr;
        int rc = launch_aggregate<16>(minmax, plan, d_row_ptr, d_col, d_rows + c0, ldr, fc, row_begin, row_end, a, b, ld, st,
                                      d_mean_in ? d_mean_in + (int64_t)c0 * ld : nullptr);
        if (rc != GRX_OK) return rc;
    }
    return GRX_OK;
}

}  // namespace

namespace {
struct PackedPlacement { int row_bytes; uint8_t word[8], shift[8]; uint8_t d_word, d_shift; };

// fields in order, first into word 0 while they fit, then word 1; the neighbour count last; no field straddles a word
bool place_fields(const grx_packed_layout *L, PackedPlacement *P)
{
    if (!L || L->n_fields < 1 || L->n_fields > 7 || L->n_out < 1 || L->n_out > 8 || L->degree_bits < 0 || L->degree_bits > 31)
        return false;
    int used[2] = {0, 0}, w = 0;
    auto put = [&](int bits, uint8_t *word, uint8_t *shift) {
        if (bits < 1 || bits > 62) return false;
        if (used[w] + bits > 64) { if (w == 1) return false; w = 1; }
        *word = (uint8_t)w; *shift = (uint8_t)used[w];
        used[w] += bits;
        return true;
    };
    for (int k = 0; k < L->n_fields; ++k)
        if (!put(L->field_bits[k], &P->word[k], &P->shift[k])) return false;
    P->d_word = P->d_shift = 0;
    if (L->degree_bits && !put(L->degree_bits, &P->d_word, &P->d_shift)) return false;
    for (int j = 0; j < L->n_out; ++j)
        if (L->out_field[j] < 0 || L->out_field[j] >= L->n_fields) return false;
    P->row_bytes = used[1] ? 16 : 8;
    return true;
}

// lanes per output row (S): fewer lanes = more gathers in flight per lane (8 / S per trip) and more rows per wavefront
int packed_slots()
{
    static const int s = [] {
        const char *e = std::getenv("GRX_PACKED_SLOTS");
        const int v = e ? std::atoi(e) : 0;
        return (v == 2 || v == 4 || v == 8) ? v : 4;
    }();
    return s;
}

template <int WORDS, int F>
void launch_packed(hipStream_t st, const int64_t *row_ptr, const int32_t *col, const void *rows, const PackedDesc &d,
                   int64_t rb, int64_t re, double *s, double *m, int64_t ld, const BlockWork &bw)
{
    const int S = packed_slots();
    const int64_t want = grx_ceil_div((re - rb) * S, 256);
    const int grid = (int)(want < 1 ? 1 : (want > GRX_NUM_CU * 32 ? GRX_NUM_CU * 32 : want));
    const unsigned long long *r = reinterpret_cast<const unsigned long long *>(rows);
    if (S == 2) aggregate_packed_kernel<WORDS, F, 2><<<grid, 256, 0, st>>>(row_ptr, col, r, d, rb, re, s, m, ld, bw);
    else if (S == 4) aggregate_packed_kernel<WORDS, F, 4><<<grid, 256, 0, st>>>(row_ptr, col, r, d, rb, re, s, m, ld, bw);
    else aggregate_packed_kernel<WORDS, F, 8><<<grid, 256, 0, st>>>(row_ptr, col, r, d, rb, re, s, m, ld, bw);
}
}  // namespace

extern "C" {

int grx_row_sums(int64_t n, const int64_t *d_row_ptr, const int32_t *d_col, const double *d_w,
                 int add_self_loop, int64_t row_begin, int64_t row_end, double *d_out, void *stream)
{
    GRX_REQUIRE(n >= 0 && row_begin >= 0 && row_begin <= row_end && row_end <= n,
                "grx_row_sums: bad row range [%lld,%lld) for n=%lld", (long long)row_begin,
                (long long)row_end, (long long)n);
    if (row_end == row_begin) return GRX_OK;
    GRX_REQUIRE(d_row_ptr && d_col && d_out, "grx_row_sums: NULL pointer");
    const int64_t nrows = row_end - row_begin;
    const int64_t want = grx_ceil_div(nrows * 8, 256);
    const int grid = (int)(want > GRX_NUM_CU * 16 ? GRX_NUM_CU * 16 : want);
    { GRX_PROF(GRX_K_ROW_SUMS, grx_stream(stream));
    row_sums_kernel<8, 4><<<grid, 256, 0, grx_stream(stream)>>>(d_row_ptr, d_col, d_w, add_self_loop,
                                                            row_begin, row_end, d_out);
    }
    GRX_LAUNCH_CHECK();
    return GRX_OK;
}

int grx_add_columns(int64_t n, const double *d_a, const double *d_b, double *d_out, void *stream)
{
    GRX_REQUIRE(n >= 0, "grx_add_columns: n < 0");
    if (n == 0) return GRX_OK;
    GRX_REQUIRE(d_a && d_b && d_out, "grx_add_columns: NULL pointer");
    const int64_t want = grx_ceil_div(n, 256 * 4);
    { GRX_PROF(GRX_K_ADD_COLUMNS, grx_stream(stream));
    add_columns_kernel<<<(int)(want > 2048 ? 2048 : want), 256, 0, grx_stream(stream)>>>(n, d_a, d_b, d_out);
    }
    GRX_LAUNCH_CHECK();
    return GRX_OK;
}

static size_t ego_parts_cap(int64_t nnz) { return (size_t)(nnz > 0 ? nnz : 0) / 512 + (size_t)(nnz > 0 ? nnz : 0) / EGO_PART + 64; }

size_t grx_egonet_workspace_bytes(int64_t n, int64_t nnz)
{
    const size_t rows = (size_t)((n > 0 ? n : 0) + 64);
    return rows * sizeof(EgoSlot) + 2 * rows * sizeof(int32_t) + 256 + ego_parts_cap(nnz) * (3 * sizeof(int32_t) + 2 * sizeof(double)) + 64;
}

int grx_egonet_features(int64_t n, int64_t nnz, const int64_t *d_row_ptr, const int32_t *d_col,
                        const double *d_w, const double *d_rowsum, int directed,
                        int64_t row_begin, int64_t row_end, double *d_internal,
                        double *d_external, void *d_workspace, size_t workspace_bytes, void *stream)
{
    GRX_REQUIRE(n >= 0 && row_begin >= 0 && row_begin <= row_end && row_end <= n,
                "grx_egonet_features: bad row range");
    if (row_end == row_begin) return GRX_OK;
    GRX_REQUIRE(d_row_ptr && d_col && d_internal && d_external, "grx_egonet_features: NULL pointer");
    GRX_REQUIRE(d_w == nullptr || d_rowsum != nullptr,
                "grx_egonet_features: weighted graphs need d_rowsum (grx_row_sums, add_self_loop=0)");
    GRX_REQUIRE(nnz >= 0 && nnz < ((int64_t)1 << EGO_DEG_SHIFT), "grx_egonet_features: nnz must be in [0, 2^40) (a slot keeps a row's begin in 40 bits)");
    GRX_REQUIRE(d_workspace != nullptr && workspace_bytes >= grx_egonet_workspace_bytes(n, nnz),
                "grx_egonet_features: workspace too small (grx_egonet_workspace_bytes)");
    GRX_REQUIRE(n < ((int64_t)1 << 31), "grx_egonet_features: more than 2^31 - 1 nodes");
    hipStream_t st = grx_stream(stream);
    const int64_t nrows = row_end - row_begin;
    constexpr int64_t HUB = 512;             // out-degree from which a row goes to the workgroup kernel (egonet_big_kernel, 256 threads, split into parts)
    // workspace: row slots | counters (256 bytes) | rows of the wide group kernel, the wavefront and the workgroup kernel
    const size_t rows_cap = (size_t)(n + 64);
    EgoSlot *slots = reinterpret_cast<EgoSlot *>(d_workspace);
    unsigned *counts = reinterpret_cast<unsigned *>(slots + rows_cap);
    int32_t *wide_rows = reinterpret_cast<int32_t *>(counts + 64);
    int32_t *mid_rows = wide_rows + rows_cap;
    double *part_out = reinterpret_cast<double *>(reinterpret_cast<char *>(mid_rows + rows_cap) + ((16 - (((size_t)(mid_rows + rows_cap)) & 15)) & 15));
    int32_t *hub_parts = reinterpret_cast<int32_t *>(part_out + 2 * ego_parts_cap(nnz));
    GRX_CHECK_HIP(hipMemsetAsync(counts, 0, 256, st));
    {
        const int64_t want = grx_ceil_div(n * 32, 256);
        GRX_PROF(GRX_K_EGONET_WAVE, st);
        // (one word per thread, no grid-stride cap: the row_ptr -> col chain of a thread is two dependent round trips)
        egonet_prepare_kernel<<<(int)(want > ((int64_t)1 << 30) ? ((int64_t)1 << 30) : want), 256, 0, st>>>(
            n, d_row_ptr, d_col, d_w ? d_rowsum : nullptr, row_begin, row_end, HUB, slots, wide_rows, mid_rows, hub_parts, counts);
        GRX_LAUNCH_CHECK();
        // nodes with at most EGO_GROUP_MAX neighbours: eight lanes each; the rest: a wavefront each
        const int64_t gwant = grx_ceil_div(nrows * 8, 256);
        const int ggrid = (int)(gwant > GRX_NUM_CU * 32 ? GRX_NUM_CU * 32 : gwant);
        const int64_t wwant = grx_ceil_div(nrows * 8, 256 * 16);               // a few per cent of the rows at most
        const int wgrid = (int)(wwant > GRX_NUM_CU * 8 ? GRX_NUM_CU * 8 : (wwant < 1 ? 1 : wwant));
        if (directed) {
            egonet_group_kernel<2, EGO_SLOTS, true><<<ggrid, 256, 0, st>>>(d_row_ptr, d_col, d_w, slots, row_begin, row_end, nullptr,
                                                                           nullptr, d_internal, d_external);
            GRX_LAUNCH_CHECK();
            egonet_group_kernel<2, EGO_SLOTS_WIDE, true><<<wgrid, 256, 0, st>>>(d_row_ptr, d_col, d_w, slots, row_begin, row_end,
                                                                                wide_rows, counts + 0, d_internal, d_external);
        } else {
            egonet_group_kernel<2, EGO_SLOTS, false><<<ggrid, 256, (size_t)(256 / 8) * EGO_GROUP_MAX * sizeof(double), st>>>(
                d_row_ptr, d_col, d_w, slots, row_begin, row_end, nullptr, nullptr, d_internal, d_external);
            GRX_LAUNCH_CHECK();
            egonet_group_kernel<2, EGO_SLOTS_WIDE, false><<<wgrid, 256, (size_t)(256 / 8) * EGO_GROUP_MAX_WIDE * sizeof(double), st>>>(
                d_row_ptr, d_col, d_w, slots, row_begin, row_end, wide_rows, counts + 0, d_internal, d_external);
        }
        GRX_LAUNCH_CHECK();
        // 65 .. HUB - 1 neighbours: a wavefront per node, 32 K filter bits each (>= 64 per member)
        const int64_t want4 = grx_ceil_div(nrows, 4 * 16);
        const int grid = (int)(want4 > GRX_NUM_CU * 16 ? GRX_NUM_CU * 16 : (want4 < 1 ? 1 : want4));
        egonet_big_kernel<1><<<grid, 256, 4 * (1024 + 64) * sizeof(unsigned), st>>>(d_row_ptr, d_col, d_w, slots, directed, mid_rows,
                                                                             counts + 1, 1024, d_internal, d_external, nullptr);
        GRX_LAUNCH_CHECK();
    }
    {
        // HUB and more: a workgroup per part of 1024 members, 256 K filter bits (16 per member up to 16 K neighbours; beyond
        // that more ids pass the filter and are turned away by the search in row(v)); then the parts of a row in order
        const int64_t hwant = grx_ceil_div(nrows, 16);
        const int grid = (int)(hwant > GRX_NUM_CU * 8 ? GRX_NUM_CU * 8 : (hwant < 1 ? 1 : hwant));
        GRX_PROF(GRX_K_EGONET_BLOCK, st);
        egonet_big_kernel<4><<<grid, 256, (8192 + 1024) * sizeof(unsigned), st>>>(d_row_ptr, d_col, d_w, slots, directed, hub_parts,
                                                                         counts + 2, 8192, d_internal, d_external, part_out);
        GRX_LAUNCH_CHECK();
        egonet_combine_kernel<<<64, 256, 0, st>>>(hub_parts, counts + 2, part_out, d_internal, d_external);
        GRX_LAUNCH_CHECK();
    }
    return GRX_OK;
}

int grx_triangle_counts(int64_t n, const int64_t *d_o_row_ptr, const int32_t *d_o_col, const uint64_t *d_o_arc,
                        int64_t row_begin, int64_t row_end, uint64_t *d_T, void *stream)
{
    GRX_REQUIRE(n >= 0 && row_begin >= 0 && row_begin <= row_end && row_end <= n, "grx_triangle_counts: bad row range");
    if (row_end == row_begin) return GRX_OK;
    GRX_REQUIRE(d_o_row_ptr && d_o_col && d_o_arc && d_T, "grx_triangle_counts: NULL pointer");
    const int64_t want = grx_ceil_div((row_end - row_begin) * 8, TRI_THREADS);   // ~64 arcs per wavefront and sweep at 8 arcs per row
    static const int rounds = [] { const char *e = std::getenv("GRX_TRI_ROUNDS"); return e ? atoi(e) : 4; }();
    const int64_t cap = (int64_t)GRX_NUM_CU * (2048 / TRI_THREADS) * rounds;   // workgroups that fill the chip, times rounds
    const int grid = (int)(want > cap ? cap : (want < 1 ? 1 : want));
    { GRX_PROF(GRX_K_TRIANGLES, grx_stream(stream));
    triangle_count_arcs_kernel<<<grid, TRI_THREADS, 0, grx_stream(stream)>>>(
        d_o_row_ptr, d_o_col, reinterpret_cast<const unsigned long long *>(d_o_arc), row_begin, row_end,
        reinterpret_cast<unsigned long long *>(d_T));
    }
    GRX_LAUNCH_CHECK();
    return GRX_OK;
}

int grx_egonet_unweighted(int64_t n, const int64_t *d_row_ptr, const int32_t *d_col, const uint64_t *d_T,
                          int64_t row_begin, int64_t row_end, double *d_internal, double *d_external,
                          int32_t *d_scratch, const int32_t *d_hub_rows, int64_t n_hub_rows, int64_t hub_degree,
                          void *stream)
{
    GRX_REQUIRE(n >= 0 && row_begin >= 0 && row_begin <= row_end && row_end <= n, "grx_egonet_unweighted: bad row range");
    if (n == 0) return GRX_OK;
    GRX_REQUIRE(d_row_ptr && d_col && d_T && d_internal && d_external && d_scratch, "grx_egonet_unweighted: NULL pointer");
    hipStream_t st = grx_stream(stream);
    {
        const int64_t want = grx_ceil_div(n, 256);
        GRX_PROF(GRX_K_EGONET_FINISH, st);
        node_info_kernel<<<(int)(want > GRX_NUM_CU * 16 ? GRX_NUM_CU * 16 : want), 256, 0, st>>>(n, d_row_ptr, d_col, d_scratch);
    }
    GRX_LAUNCH_CHECK();
    if (row_end > row_begin) {
        const int64_t want = grx_ceil_div((row_end - row_begin) * 8, 256);
        GRX_PROF(GRX_K_EGONET_FINISH, st);
        const int64_t hub_deg = (d_hub_rows && n_hub_rows > 0) ? hub_degree : ((int64_t)1 << 62);
        const bool hubs = d_hub_rows && n_hub_rows > 0;
        const int hub_blocks = hubs ? (int)(n_hub_rows > GRX_NUM_CU * 8 ? GRX_NUM_CU * 8 : n_hub_rows) : 0;
        const int row_blocks = (int)(want > GRX_NUM_CU * 32 ? GRX_NUM_CU * 32 : want);
        egonet_from_triangles_kernel<<<hub_blocks + row_blocks, 256, 0, st>>>(
            d_row_ptr, d_col, d_scratch, reinterpret_cast<const unsigned long long *>(d_T), row_begin, row_end,
            hub_deg, d_hub_rows, hubs ? n_hub_rows : 0, hub_blocks, d_internal, d_external);
    }
    GRX_LAUNCH_CHECK();
    return GRX_OK;
}

int grx_pack_rows(int64_t n, int f, const double *const *h_col_ptrs, double *d_rows, int ldr,
                  void *stream)
{
    GRX_REQUIRE(n >= 0 && f >= 0 && ldr >= f, "grx_pack_rows: bad shape n=%lld f=%d ldr=%d",
                (long long)n, f, ldr);
    if (n == 0 || ldr == 0) return GRX_OK;
    GRX_REQUIRE(h_col_ptrs && d_rows, "grx_pack_rows: NULL pointer");
    const int64_t want = grx_ceil_div(n, 256);
    const int grid = (int)(want > GRX_NUM_CU * 16 ? GRX_NUM_CU * 16 : want);
    // the pointer table travels as a kernel argument, GRX_MAX_PTRS columns per launch
    for (int c0 = 0; c0 < f || c0 == 0; c0 += GRX_MAX_PTRS) {
        const int fc = (f - c0 < GRX_MAX_PTRS) ? f - c0 : GRX_MAX_PTRS;
        const bool last = c0 + fc >= f;
        GrxPtrTable tab;
        for (int c = 0; c < fc; ++c) tab.p[c] = h_col_ptrs[c0 + c];
        {
            GRX_PROF(GRX_K_PACK_ROWS, grx_stream(stream));
            if (ldr % 16 == 0) {
                const int64_t tiles = grx_ceil_div(n, 64);
                const int tgrid = (int)(tiles > GRX_NUM_CU * 16 ? GRX_NUM_CU * 16 : tiles);
                pack_rows_tiled_kernel<<<tgrid, 256, 0, grx_stream(stream)>>>(n, fc, ldr, tab, d_rows, c0, last ? f : ldr);
            } else if (ldr == 8 && c0 == 0 && last && (reinterpret_cast<uintptr_t>(d_rows) & 15) == 0) {
                const int64_t blocks = grx_ceil_div(grx_ceil_div(n, 64), 4);
                pack_rows8_kernel<<<(int)(blocks > GRX_NUM_CU * 16 ? GRX_NUM_CU * 16 : blocks), 256, 0, grx_stream(stream)>>>(
                    n, fc, tab, d_rows);
            } else {
                pack_rows_kernel<<<grid, 256, 0, grx_stream(stream)>>>(n, fc, ldr, tab, d_rows, c0, last ? f : ldr);
            }
        }
        GRX_LAUNCH_CHECK();
        if (last) break;
    }
    return GRX_OK;
}

int grx_aggregate_plan_create(int64_t n, const int64_t *h_row_ptr, grx_aggregate_plan **out)
{
    GRX_REQUIRE(n >= 0 && out != nullptr && (n == 0 || h_row_ptr != nullptr), "grx_aggregate_plan_create: bad arguments");
    auto *p = new grx_aggregate_plan();
    p->n = n;
    const double avg = n ? (double)(h_row_ptr[n] - h_row_ptr[0]) / (double)n : 0.0;
    p->lanes_per_row = avg < 12 ? 4 : avg < 24 ? 8 : avg < 48 ? 16 : 32;
    std::vector<int32_t> long_rows, blk_len, blk_row;
    std::vector<int64_t> blk_ptr, blk_begin;
    std::vector<uint8_t> blk_ops;
    for (int64_t v = 0; v < n; ++v) {
        const int64_t d = h_row_ptr[v + 1] - h_row_ptr[v];
        if (d > p->max_degree) p->max_degree = d;
        if (d <= PW_BLOCK) continue;
        blk_ptr.push_back((int64_t)blk_begin.size());
        const size_t before = blk_begin.size();
        for (int64_t c0 = 0; c0 < d; c0 += PW_CHUNK) {
            pairwise_blocks(h_row_ptr[v] + c0, (d - c0 < PW_CHUNK) ? d - c0 : PW_CHUNK, blk_begin, blk_len, blk_ops);
            blk_ops.back() |= 0x80;                              // end of a chunk
        }
        blk_row.insert(blk_row.end(), blk_begin.size() - before, (int32_t)long_rows.size());
        long_rows.push_back((int32_t)v);
    }
    blk_ptr.push_back((int64_t)blk_begin.size());
    p->n_long = (int64_t)long_rows.size();
    p->n_blocks = (int64_t)blk_begin.size();
    auto upload = [](void **dst, const void *src, size_t bytes) -> hipError_t {
        hipError_t e = hipMalloc(dst, bytes ? bytes : 8);
        if (e == hipSuccess && bytes) e = hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice);
        return e;
    };
    hipError_t e = hipSuccess;
    if (p->n_long) {
        e = upload((void **)&p->d_long_rows, long_rows.data(), long_rows.size() * 4);
        if (e == hipSuccess) e = upload((void **)&p->d_blk_ptr, blk_ptr.data(), blk_ptr.size() * 8);
        if (e == hipSuccess) e = upload((void **)&p->d_blk_begin, blk_begin.data(), blk_begin.size() * 8);
        if (e == hipSuccess) e = upload((void **)&p->d_blk_len, blk_len.data(), blk_len.size() * 4);
        if (e == hipSuccess) e = upload((void **)&p->d_blk_row, blk_row.data(), blk_row.size() * 4);
        if (e == hipSuccess) e = upload((void **)&p->d_blk_ops, blk_ops.data(), blk_ops.size());
        if (e == hipSuccess) e = hipMalloc((void **)&p->d_blk_sums, (size_t)p->n_blocks * 16 * 8);
    }
    if (e != hipSuccess) {
        grx_set_error("grx_aggregate_plan_create: %s", hipGetErrorString(e));
        grx_aggregate_plan_destroy(p);
        return GRX_ERR_HIP;
    }
    *out = p;
    return GRX_OK;
}

void grx_aggregate_plan_destroy(grx_aggregate_plan *p)
{
    if (!p) return;
    (void)hipFree(p->d_long_rows); (void)hipFree(p->d_blk_ptr); (void)hipFree(p->d_blk_begin);
    (void)hipFree(p->d_blk_len); (void)hipFree(p->d_blk_row); (void)hipFree(p->d_blk_ops);
    (void)hipFree(p->d_blk_sums);
    delete p;
}

int grx_aggregate_plan_info(const grx_aggregate_plan *p, int64_t *n_long_rows, int64_t *n_blocks, int *lanes_per_row)
{
    GRX_REQUIRE(p != nullptr, "grx_aggregate_plan_info: NULL plan");
    if (n_long_rows) *n_long_rows = p->n_long;
    if (n_blocks) *n_blocks = p->n_blocks;
    if (lanes_per_row) *lanes_per_row = p->lanes_per_row;
    return GRX_OK;
}

int grx_aggregate_plan_set_lanes(grx_aggregate_plan *p, int lanes_per_row)
{
    GRX_REQUIRE(p != nullptr, "grx_aggregate_plan_set_lanes: NULL plan");
    GRX_REQUIRE(lanes_per_row == 4 || lanes_per_row == 8 || lanes_per_row == 16 || lanes_per_row == 32,
                "grx_aggregate_plan_set_lanes: lanes_per_row must be 4, 8, 16 or 32");
    p->lanes_per_row = lanes_per_row;
    return GRX_OK;
}

int grx_aggregate(const grx_aggregate_plan *plan, const int64_t *d_row_ptr, const int32_t *d_col, int f,
                  const double *d_rows, int ldr, int64_t row_begin, int64_t row_end,
                  double *d_sum, double *d_mean, int64_t ld, void *stream)
{
    return aggregate_dispatch(false, plan, d_row_ptr, d_col, f, d_rows, ldr, row_begin, row_end, d_sum, d_mean, ld, stream);
}

int grx_aggregate_var(const grx_aggregate_plan *plan, const int64_t *d_row_ptr, const int32_t *d_col, int f,
                      const double *d_rows, int ldr, int64_t row_begin, int64_t row_end, const double *d_mean,
                      double *d_var, double *d_std, int64_t ld, void *stream)
{
    GRX_REQUIRE(d_mean != nullptr || f == 0 || row_begin == row_end, "grx_aggregate_var: needs the neighbour means (grx_aggregate)");
    return aggregate_dispatch(false, plan, d_row_ptr, d_col, f, d_rows, ldr, row_begin, row_end, d_var, d_std, ld, stream,
                              d_mean);
}

int grx_aggregate_minmax(const grx_aggregate_plan *plan, const int64_t *d_row_ptr, const int32_t *d_col, int f,
                         const double *d_rows, int ldr, int64_t row_begin, int64_t row_end,
                         double *d_min, double *d_max, int64_t ld, void *stream)
{
    return aggregate_dispatch(true, plan, d_row_ptr, d_col, f, d_rows, ldr, row_begin, row_end, d_min, d_max, ld, stream);
}

int grx_aggregate_prod(const int64_t *d_row_ptr, const int32_t *d_col, int f, const double *d_rows, int ldr,
                       int64_t row_begin, int64_t row_end, double *d_prod, int64_t ld, void *stream)
{
    GRX_REQUIRE(f >= 0 && ldr >= f && row_begin >= 0 && row_begin <= row_end && ld >= row_end,
                "grx_aggregate_prod: bad shape");
    if (f == 0 || row_end == row_begin) return GRX_OK;
    GRX_REQUIRE(d_row_ptr && d_col && d_rows && d_prod, "grx_aggregate_prod: NULL pointer");
    const int64_t want = grx_ceil_div((row_end - row_begin) * f, 256);
    const int grid = (int)(want > GRX_NUM_CU * 32 ? GRX_NUM_CU * 32 : want);
    aggregate_prod_kernel<<<grid, 256, 0, grx_stream(stream)>>>(d_row_ptr, d_col, d_rows, ldr, f, row_begin, row_end,
                                                               d_prod, ld);
    GRX_LAUNCH_CHECK();
    return GRX_OK;
}

/* row stride in int32 of the integer gather source: 16- or 32-byte rows; 0 = no integer kernel for that many columns */
int grx_aggregate_ldi(int f) { return f <= 0 ? 0 : f <= 4 ? 4 : f <= 8 ? 8 : 0; }

int grx_pack_rows_i32(int64_t n, int f, const double *const *h_col_ptrs, int32_t *d_rows, int ldi, void *stream)
{
    GRX_REQUIRE(n >= 0 && f >= 1 && ldi == grx_aggregate_ldi(f), "grx_pack_rows_i32: ldi must be grx_aggregate_ldi(f)");
    if (n == 0) return GRX_OK;
    GRX_REQUIRE(h_col_ptrs && d_rows, "grx_pack_rows_i32: NULL pointer");
    GrxPtrTable tab;
    for (int c = 0; c < f; ++c) {
        GRX_REQUIRE(h_col_ptrs[c] != nullptr, "grx_pack_rows_i32: column %d is NULL", c);
        tab.p[c] = h_col_ptrs[c];
    }
    const int64_t want = grx_ceil_div(n, 256);
    { GRX_PROF(GRX_K_PACK_ROWS, grx_stream(stream));
    pack_rows_i32_kernel<<<(int)(want > GRX_NUM_CU * 16 ? GRX_NUM_CU * 16 : want), 256, 0, grx_stream(stream)>>>(
        n, f, ldi, tab, d_rows);
    }
    GRX_LAUNCH_CHECK();
    return GRX_OK;
}

int grx_aggregate_i32_ok(const grx_aggregate_plan *plan, int f)
{
    // integer sums stay exact in fp64 while max_degree * 2^31 <= 2^53
    return plan != nullptr && grx_aggregate_ldi(f) != 0 && plan->max_degree < ((int64_t)1 << 22);
}

int grx_aggregate_i32(const grx_aggregate_plan *plan, const int64_t *d_row_ptr, const int32_t *d_col, int f,
                      const int32_t *d_rows, int ldi, int64_t row_begin, int64_t row_end, double *d_sum, double *d_mean,
                      int64_t ld, void *stream)
{
    GRX_REQUIRE(plan != nullptr, "grx_aggregate_i32: NULL plan");
    GRX_REQUIRE(grx_aggregate_i32_ok(plan, f) && ldi == grx_aggregate_ldi(f),
                "grx_aggregate_i32: f=%d / ldi=%d / max degree %lld outside the integer kernel's range", f, ldi,
                (long long)plan->max_degree);
    const int64_t n = plan->n;
    GRX_REQUIRE(row_begin >= 0 && row_begin <= row_end && row_end <= n && ld >= n, "grx_aggregate_i32: bad row range");
    if (row_end == row_begin) return GRX_OK;
    GRX_REQUIRE(d_row_ptr && d_col && d_rows, "grx_aggregate_i32: NULL pointer");
    GRX_REQUIRE((reinterpret_cast<uintptr_t>(d_rows) & 63) == 0, "grx_aggregate_i32: d_rows must be 64-byte aligned");
    hipStream_t st = grx_stream(stream);
    const int CL = ldi / 4;
    int G = plan->lanes_per_row;
    if (G < 4) G = 4;
    if (G > 16) G = 16;
    if (G < CL) G = CL;
    const int64_t want = grx_ceil_div((row_end - row_begin) * G, 256);
    const int grid = (int)(want < 1 ? 1 : (want > GRX_NUM_CU * 32 ? GRX_NUM_CU * 32 : want));
    const BlockWork bw{plan->d_long_rows, plan->d_blk_begin, plan->d_blk_len, plan->d_blk_row,
                       plan->n_long > 0 ? plan->n_blocks : 0, plan->d_blk_sums};
    {
        GRX_PROF(GRX_K_AGGREGATE, st);
#define GRX_I32_CASE(LL, GG)                                                                                          \
        if (ldi == LL && G == GG)                                                                                     \
            aggregate_i32_kernel<LL, GG><<<grid, 256, 0, st>>>(d_row_ptr, d_col, d_rows, f, row_begin, row_end, d_sum, \
                                                               d_mean, ld, bw);
        GRX_I32_CASE(4, 4) GRX_I32_CASE(4, 8) GRX_I32_CASE(4, 16) GRX_I32_CASE(8, 4) GRX_I32_CASE(8, 8) GRX_I32_CASE(8, 16)
#undef GRX_I32_CASE
    }
    GRX_LAUNCH_CHECK();
    if (plan->n_long > 0) {
        GRX_PROF(GRX_K_AGGREGATE_HUB, st);
        const int64_t cwant = grx_ceil_div(plan->n_long, 16);
        aggregate_i32_combine_kernel<<<(unsigned)(cwant > GRX_NUM_CU * 32 ? GRX_NUM_CU * 32 : cwant), 256, 0, st>>>(
            d_row_ptr, f, row_begin, row_end, plan->d_long_rows, plan->d_blk_ptr, plan->n_long, plan->d_blk_sums, d_sum,
            d_mean, ld);
    }
    GRX_LAUNCH_CHECK();
    return GRX_OK;
}

/* ---- bit-packed integer rows (see PackedDesc above) ---------------------------------------------------------- */

}  // extern "C"
int64_t grx_internal_plan_max_degree(const grx_aggregate_plan *plan) { return plan ? plan->max_degree : 0; }
extern "C" {

int grx_packed_row_bytes(const grx_packed_layout *layout)
{
    PackedPlacement P;
    return place_fields(layout, &P) ? P.row_bytes : 0;
}

int grx_column_bits(int64_t n, int ncols, const double *d_block, int64_t ld, int64_t row_begin, int64_t row_end,
                    uint64_t int_mask, int32_t *d_bits, void *stream)
{
    GRX_REQUIRE(n >= 0 && ncols >= 0 && ncols <= 64 && row_begin >= 0 && row_begin <= row_end && row_end <= n && ld >= n,
                "grx_column_bits: bad shape (at most 64 columns per call)");
    if (ncols == 0) return GRX_OK;
    GRX_REQUIRE(d_block && d_bits, "grx_column_bits: NULL pointer");
    // d_bits accumulates by atomicMax: the CALLER zeroes it (grx_refex_run clears it with its distance matrix)
    const int64_t want = grx_ceil_div(row_end - row_begin, 256 * 16);
    const dim3 grid((unsigned)(want < 1 ? 1 : (want > 256 ? 256 : want)), (unsigned)ncols);
    column_bits_kernel<<<grid, 256, 0, grx_stream(stream)>>>(d_block, ld, row_begin, row_end, int_mask, d_bits);
    GRX_LAUNCH_CHECK();
    return GRX_OK;
}

int grx_pack_fields(int64_t n, const grx_packed_layout *layout, const double *const *h_field_cols, const int64_t *d_row_ptr,
                    void *d_rows, void *stream)
{
    PackedPlacement P;
    GRX_REQUIRE(place_fields(layout, &P), "grx_pack_fields: the fields do not fit two 64-bit words (grx_packed_row_bytes)");
    GRX_REQUIRE(n >= 0 && h_field_cols && d_rows && (d_row_ptr || !layout->degree_bits), "grx_pack_fields: NULL pointer");
    GRX_REQUIRE((reinterpret_cast<uintptr_t>(d_rows) & 15) == 0, "grx_pack_fields: d_rows must be 16-byte aligned");
    if (n == 0) return GRX_OK;
    PackFieldsArgs a{};
    a.n_fields = layout->n_fields;
    for (int k = 0; k < layout->n_fields; ++k) {
        GRX_REQUIRE(h_field_cols[k] != nullptr, "grx_pack_fields: field column %d is NULL", k);
        a.src[k] = h_field_cols[k];
        a.word[k] = P.word[k];
        a.shift[k] = P.shift[k];
    }
    a.d_word = P.d_word; a.d_shift = P.d_shift; a.d_bits = (uint8_t)layout->degree_bits;
    hipStream_t st = grx_stream(stream);
    const int64_t want = grx_ceil_div(n, 256);
    const int grid = (int)(want > GRX_NUM_CU * 16 ? GRX_NUM_CU * 16 : want);
    {
        GRX_PROF(GRX_K_PACK_ROWS, st);
        if (P.row_bytes == 8) pack_fields_kernel<1><<<grid, 256, 0, st>>>(n, a, d_row_ptr, reinterpret_cast<unsigned long long *>(d_rows));
        else pack_fields_kernel<2><<<grid, 256, 0, st>>>(n, a, d_row_ptr, reinterpret_cast<unsigned long long *>(d_rows));
    }
    GRX_LAUNCH_CHECK();
    return GRX_OK;
}

int grx_aggregate_packed(const grx_aggregate_plan *plan, const int64_t *d_row_ptr, const int32_t *d_col,
                         const grx_packed_layout *layout, const void *d_rows, int64_t row_begin, int64_t row_end,
                         double *d_sum, double *d_mean, int64_t ld, void *stream)
{
    GRX_REQUIRE(plan != nullptr, "grx_aggregate_packed: NULL plan");
    PackedPlacement P;
    GRX_REQUIRE(place_fields(layout, &P), "grx_aggregate_packed: the fields do not fit two 64-bit words");
    const int64_t n = plan->n;
    GRX_REQUIRE(row_begin >= 0 && row_begin <= row_end && row_end <= n && ld >= n, "grx_aggregate_packed: bad row range");
    if (row_end == row_begin) return GRX_OK;
    GRX_REQUIRE(d_row_ptr && d_col && d_rows, "grx_aggregate_packed: NULL pointer");
    PackedDesc d{};
    d.n_out = layout->n_out;
    bool any_mean = false;
    for (int j = 0; j < layout->n_out; ++j) {
        const int k = layout->out_field[j];
        d.word[j] = P.word[k]; d.shift[j] = P.shift[k]; d.bits[j] = (uint8_t)layout->field_bits[k];
        d.is_mean[j] = layout->out_is_mean[j] ? 1 : 0;
        any_mean = any_mean || d.is_mean[j];
    }
    GRX_REQUIRE(!any_mean || layout->degree_bits > 0, "grx_aggregate_packed: mean summands need the neighbour-count field");
    d.d_word = P.d_word; d.d_shift = P.d_shift; d.d_bits = (uint8_t)layout->degree_bits;
    hipStream_t st = grx_stream(stream);
    const BlockWork bw{plan->d_long_rows, plan->d_blk_begin, plan->d_blk_len, plan->d_blk_row,
                       plan->n_long > 0 ? plan->n_blocks : 0, plan->d_blk_sums};
    {
        GRX_PROF(GRX_K_AGGREGATE, st);
#define GRX_PACKED_CASE(FF)                                                                                           \
        case FF:                                                                                                      \
            if (P.row_bytes == 8) launch_packed<1, FF>(st, d_row_ptr, d_col, d_rows, d, row_begin, row_end, d_sum, d_mean, ld, bw); \
            else launch_packed<2, FF>(st, d_row_ptr, d_col, d_rows, d, row_begin, row_end, d_sum, d_mean, ld, bw); \
            break;
        switch (layout->n_out) {
            GRX_PACKED_CASE(1) GRX_PACKED_CASE(2) GRX_PACKED_CASE(3) GRX_PACKED_CASE(4)
            GRX_PACKED_CASE(5) GRX_PACKED_CASE(6) GRX_PACKED_CASE(7) GRX_PACKED_CASE(8)
        default: break;
        }
#undef GRX_PACKED_CASE
    }
    GRX_LAUNCH_CHECK();
    if (plan->n_long > 0) {
        GRX_PROF(GRX_K_AGGREGATE_HUB, st);
        const int64_t cwant = grx_ceil_div(plan->n_long, 16);
        aggregate_combine_kernel<<<(unsigned)(cwant > GRX_NUM_CU * 32 ? GRX_NUM_CU * 32 : cwant), 256, 0, st>>>(
            d_row_ptr, layout->n_out, row_begin, row_end, plan->d_long_rows, plan->d_blk_ptr, plan->n_long, plan->d_blk_ops,
            plan->d_blk_sums, d_sum, d_mean, ld, 0);
    }
    GRX_LAUNCH_CHECK();
    return GRX_OK;
}

/* row stride (in doubles) grx_pack_rows / grx_aggregate use for f columns */
int grx_aggregate_ldr(int f)
{
    if (f <= 2) return 2;
    if (f <= 4) return 4;
    if (f <= 8) return 8;
    return (f + 15) / 16 * 16;
}

}  // extern "C"
