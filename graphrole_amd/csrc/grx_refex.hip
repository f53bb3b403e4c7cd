// grx_refex.hip -- the ReFeX generation loop below the C ABI (host driver, no device code of its own).
//
// Reference: RecursiveFeatureExtractor.extract_features / _get_next_features / _update
// (graphrole/features/extract.py:65-142) with FeaturePruner (graphrole/features/prune.py:76-139) and the
// feature graph's connected components (graphrole/graph/graph.py:7-57).
//
// Every generation is "pack the retained columns -> aggregate over neighbours -> bin the new columns ->
// Chebyshev distances over the working set -> prune", and the next generation cannot start before the
// pruning decision of this one is known.  Driven from Python that dependency chain costs ~10 boundary
// crossings, half a dozen allocations and an interpreter-speed pruner per generation -- more than the
// kernels themselves on a 100 k-node graph.  Here the whole chain is C++: kernels are enqueued back to
// back into one caller-provided arena, the only host <-> device traffic is the F x F distance matrix of
// each generation, and the pruner is a few dozen string compares.
#include "grx_common.h"

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

int64_t grx_internal_plan_max_degree(const grx_aggregate_plan *plan);                                          // grx_graph.hip
int grx_internal_vertical_log_bin(int64_t n, int ncols, const double *d_cols, int64_t ld, const uint8_t *h_is_i64, double frac,
                                  uint8_t *d_bins, int64_t ld_bins, int32_t *d_nbins, void *d_workspace,
                                  size_t workspace_bytes, int32_t *d_status, void *stream);                  // grx_prune.hip

namespace {

const char *const AGG_NAMES[] = {"sum", "mean", "min", "max", "var", "std", "prod", "median", "count", "size"};
constexpr int N_AGG_KINDS = 10;

struct Column {
    std::string name;
    int generation;
    int parent;               // index into the column list, -1 for generation 0
    int agg;                  // grx_agg, -1 for generation 0
    const double *data;
    const uint8_t *bins;
    int record_index;         // position in the output table, -1 while not recorded
    bool int32_exact = false; // every value an exact integer in [0, 2^31): may travel as an int32 gather source
    // what the values ARE, for the bit-packed gather source (grx_aggregate_packed):
    //   kind 1: exact non-negative integers S (a generation-0 count or a neighbour sum of one); bits = width of the maximum
    //   kind 2: fl(S / d) with S = column `base` (the sum candidate of the same parent) and d the neighbour count
    //   kind 0: anything else
    int kind = 0;
    int base = -1;
    int bits = -1;            // known after the generation's read-back (-1: not known)
};

// Bump allocator over the caller's arena.  The arena is a CHAIN of chunks (round 5): the first is the block the caller
// passed; when a request does not fit, the caller's grow function hands out another chunk (allocations only need to be
// contiguous in themselves, never across chunks), so a run is never repeated because a first size was a guess.
// Positions are LOGICAL offsets (chunk i covers [start_i, start_i + cap_i)): marks taken for scratch stay valid, a
// request that does not fit the rest of a chunk skips to the next one.  Without a grow function (or when it fails)
// the allocator keeps counting past the end so that the caller learns how much would have been needed.
struct Arena {
    struct Chunk { char *base; size_t cap, start; };
    std::vector<Chunk> chunks;
    size_t cur = 0, top = 0, peak = 0, grow_min = 0;
    bool overflow = false;
    grx_grow_fn grow = nullptr;
    void *grow_user = nullptr;
    Arena(void *base, size_t cap, grx_grow_fn fn, void *user) : grow(fn), grow_user(user)
    {
        chunks.push_back({reinterpret_cast<char *>(base), base ? cap : 0, 0});
        grow_min = cap / 4 > ((size_t)64 << 20) ? cap / 4 : ((size_t)64 << 20);
    }
    size_t capacity() const { return chunks.back().start + chunks.back().cap; }
    void rewind(size_t mark)
    {
        top = mark;
        while (cur > 0 && chunks[cur].start > mark) --cur;
        while (cur + 1 < chunks.size() && chunks[cur + 1].start <= mark) ++cur;
    }
    void *take(size_t bytes)
    {
        for (;;) {
            const Chunk &c = chunks[cur];
            const size_t at = grx_align_up(top > c.start ? top - c.start : 0, 256);
            if (!overflow && at + bytes <= c.cap) {
                top = c.start + at + bytes;
                if (top > peak) peak = top;
                return c.base + at;
            }
            if (!overflow && cur + 1 < chunks.size()) { ++cur; top = chunks[cur].start; continue; }
            if (!overflow && grow) {
                const size_t want = grx_align_up(bytes + 256 > grow_min ? bytes + 256 : grow_min, 256);
                void *p = grow(want, grow_user);
                if (p) {
                    chunks.push_back({reinterpret_cast<char *>(p), want, capacity()});
                    ++cur;
                    top = chunks[cur].start;
                    continue;
                }
            }
            // no room and nobody to ask: count on
            overflow = true;
            top = grx_align_up(top, 256) + bytes;
            if (top > peak) peak = top;
            return nullptr;
        }
    }
};

thread_local void *g_pinned = nullptr;
thread_local size_t g_pinned_bytes = 0;

int pinned(size_t bytes, void **out)
{
    if (g_pinned_bytes < bytes) {
        if (g_pinned) (void)hipHostFree(g_pinned);
        g_pinned = nullptr;
        g_pinned_bytes = 0;
        const size_t want = bytes < (256u << 10) ? (256u << 10) : bytes;
        GRX_CHECK_HIP(hipHostMalloc(&g_pinned, want, hipHostMallocMapped));
        g_pinned_bytes = want;
    }
    *out = g_pinned;
    return GRX_OK;
}

#define GRX_TRY(expr) do { int rc__ = (expr); if (rc__ != GRX_OK) return rc__; } while (0)


// prune.py:76-130 on the distance matrix of the working set: indices (into `work`) to drop
std::vector<int> prune(const std::vector<Column> &cols, const std::vector<int> &work, const int32_t *dist, int thresh,
                       const std::vector<std::vector<int>> &recorded)
{
    const int F = (int)work.size();
    std::vector<std::vector<int>> adj(F);
    for (int p = 0; p < F; ++p)
        for (int q = p + 1; q < F; ++q)
            if (dist[(size_t)p * F + q] <= thresh) { adj[p].push_back(q); adj[q].push_back(p); }
    std::vector<int> comp(F, -1), drop;
    std::vector<int> stack, members;
    for (int s = 0; s < F; ++s) {
        if (comp[s] >= 0 || adj[s].empty()) continue;             // isolated features are never pruned
        members.clear();
        stack.assign(1, s);
        comp[s] = s;
        while (!stack.empty()) {
            const int v = stack.back();
            stack.pop_back();
            members.push_back(v);
            for (int u : adj[v])
                if (comp[u] < 0) { comp[u] = s; stack.push_back(u); }
        }
        // the member recorded in the earliest generation survives; ties and never-recorded members by name
        int keep = -1;
        for (size_t gen = 0; gen < recorded.size() && keep < 0; ++gen)
            for (int m : members) {
                const int c = work[m];
                if (std::find(recorded[gen].begin(), recorded[gen].end(), c) == recorded[gen].end()) continue;
                if (keep < 0 || cols[c].name < cols[work[keep]].name) keep = m;
            }
        if (keep < 0)
            for (int m : members)
                if (keep < 0 || cols[work[m]].name < cols[work[keep]].name) keep = m;
        for (int m : members)
            if (m != keep) drop.push_back(m);
    }
    return drop;
}

}  // namespace

extern "C" {

// columns the generation loop bins per call of the binning (its workspace is sized for this many)
int grx_refex_bin_batch(int64_t n, int ncols)
{
    if (ncols <= 0) return 0;
    const size_t per = grx_log_bin_workspace_bytes(n, 1);
    int64_t b = per ? (int64_t)(((size_t)4 << 30) / per) : ncols;          // 4 GiB: two calls for the widest generation of config 5
    if (b < 8) b = 8;
    return (int)(ncols < b ? ncols : b);
}

// The pruning decision of grx_refex_run as a host-only entry point (no device work): tests drive it without a GPU
// against the Python FeaturePruner and the reference's known-answer tables (tests/test_host_logic_cpu.py).
int grx_host_prune(int F, const char *const *h_names, const int *h_recorded_generation, int n_generations,
                   const int32_t *h_dist, int thresh, int *h_drop)
{
    GRX_REQUIRE(F >= 0 && n_generations >= 0 && (F == 0 || (h_names && h_recorded_generation && h_dist && h_drop)),
                "grx_host_prune: bad arguments");
    std::vector<Column> cols;
    std::vector<int> work;
    std::vector<std::vector<int>> recorded(n_generations);
    for (int j = 0; j < F; ++j) {
        GRX_REQUIRE(h_names[j] != nullptr, "grx_host_prune: name %d is NULL", j);
        cols.push_back({h_names[j], 0, -1, -1, nullptr, nullptr, -1});
        work.push_back(j);
        const int g = h_recorded_generation[j];
        GRX_REQUIRE(g < n_generations, "grx_host_prune: column %d recorded in generation %d of %d", j, g, n_generations);
        if (g >= 0) recorded[g].push_back(j);
        h_drop[j] = 0;
    }
    for (int m : prune(cols, work, h_dist, thresh, recorded)) h_drop[m] = 1;
    return GRX_OK;
}

int grx_refex_run(const grx_aggregate_plan *plan, int64_t n, const int64_t *d_row_ptr, const int32_t *d_agg_col,
                  int f0, const double *const *h_gen0_cols, const char *const *h_gen0_names, const int *h_gen0_int32,
                  int max_generations, int n_aggs, const int *h_aggs, grx_comm *comm, const int64_t *h_bounds, void *d_arena,
                  size_t arena_bytes, grx_grow_fn grow, void *grow_user, int max_columns,
                  grx_refex_column *h_columns, int *n_columns, int max_gens, grx_refex_generation *h_gens,
                  int *generation_count, size_t *arena_needed, void *stream)
{
    GRX_REQUIRE(plan && d_row_ptr && d_agg_col && n >= 1 && f0 >= 1 && h_gen0_cols && h_gen0_names,
                "grx_refex_run: bad graph / generation-0 arguments");
    GRX_REQUIRE(n_aggs >= 1 && h_aggs && h_columns && n_columns && h_gens && generation_count && max_columns >= f0 &&
                max_gens >= 1, "grx_refex_run: bad output arguments");
    bool has[N_AGG_KINDS] = {};
    for (int a = 0; a < n_aggs; ++a) {
        GRX_REQUIRE(h_aggs[a] >= 0 && h_aggs[a] < N_AGG_KINDS, "grx_refex_run: unknown aggregation id %d", h_aggs[a]);
        has[h_aggs[a]] = true;
    }
    hipStream_t st = grx_stream(stream);
    // node-range sharding (include/grx.h): this rank computes rows [rb, re) of every per-node kernel; a candidate
    // block holds only those rows until the exchanges below complete what the next step reads
    const int P = comm ? grx_comm_world(comm) : 1, me = comm ? grx_comm_rank(comm) : 0;
    GRX_REQUIRE(!comm || h_bounds, "grx_refex_run: a communicator needs the row partition h_bounds");
    if (comm)
        GRX_REQUIRE(h_bounds[0] == 0 && h_bounds[P] == n, "grx_refex_run: h_bounds must run from 0 to n");
    const int64_t rb = comm ? h_bounds[me] : 0, re = comm ? h_bounds[me + 1] : n;
    Arena arena(d_arena, d_arena ? arena_bytes : 0, grow, grow_user);
    // Every allocation below is sized from RANK-INDEPENDENT upper bounds (ceil(count / P) owned columns, the largest
    // nnz slice of the partition): every rank asks for the same sizes at the same points.  WITHOUT a grow function --
    // and with arenas of one capacity on every rank -- every rank's arena overflows at the same allocation or not at
    // all, so the `if (!arena.overflow)` guards around the exchanges below take the same branch on every rank and a
    // too-small arena is a joint GRX_ERR_WORKSPACE.  WITH a grow function an overflow means that THIS rank's grow()
    // failed (the device is out of memory): that is not agreed between the ranks -- the peers wait inside the next
    // exchange until the transport's own timeout ends the job.  A joint error would cost one more latency-sized
    // collective per generation on every run; an out-of-memory rank ends the job either way (include/grx.h says so).
    int64_t nnz_rows = 0;                                    // adjacency entries of the longest row slice (median workspace)
    if (has[GRX_AGG_MEDIAN]) {
        std::vector<int64_t> ends(P + 1, 0);
        for (int q = 0; q <= P; ++q)
            GRX_CHECK_HIP(hipMemcpyAsync(&ends[q], d_row_ptr + (comm ? h_bounds[q] : (q ? n : 0)), 8, hipMemcpyDeviceToHost, st));
        GRX_CHECK_HIP(hipStreamSynchronize(st));
        for (int q = 0; q < P; ++q) nnz_rows = std::max(nnz_rows, ends[q + 1] - ends[q]);
    }
    std::vector<Column> cols;
    std::vector<int> work;                                   // working set, insertion order (extract.py:128-133)
    std::vector<std::vector<int>> recorded;                  // generation -> recorded columns (_final_features)
    int n_out = 0;
    bool table_full = false;

    // add the new columns to the working set, bin them, prune across the set, record what survives
    // (extract.py:121-142); `block` = the new columns as one contiguous [count, n] block
    int deg_bits = 1;
    while (deg_bits < 62 && (grx_internal_plan_max_degree(plan) >> deg_bits) != 0) ++deg_bits;
    static const bool packed_allowed = [] { const char *e = std::getenv("GRX_NO_PACKED_ROWS"); return !(e && *e == '1'); }();
    int gather_row_bytes = 0;                                // row width of the gather source of the generation at hand
    auto update = [&](int first_new, int count, const double *block, int generation, bool partial) -> int {
        const size_t scratch_mark = arena.top;
        uint8_t *bins = nullptr;
        if (count) {
            // bins are cached for the life of a column: re-binning is result-identical (prune.py:101-104)
            arena.rewind(scratch_mark);
            bins = reinterpret_cast<uint8_t *>(arena.take((size_t)count * n));
        }
        const size_t persistent_top = arena.top;
        // sharded: rank q bins columns q, q + P, ... of the new block as WHOLE columns (the threshold walk needs every
        // row), every rank gets back the bins of its own rows of every column -- the only rows its Chebyshev pass reads
        const int n_owned = comm ? (count > me ? (count - me + P - 1) / P : 0) : count;
        const int n_owned_max = comm ? (count + P - 1) / P : count;       // what rank 0 owns: the allocation size everywhere
        // columns are binned independently, so a wide candidate block goes through the binning in batches: the workspace
        // (two key buffers per column and more) stays below 4 GiB instead of growing with the block (config 5: 6.7 GB for
        // 82 columns of 5 M keys -- a third of the run's memory); small graphs keep the single call
        const int bin_batch = grx_refex_bin_batch(n, n_owned_max);
        const size_t ws_bytes = grx_log_bin_workspace_bytes(n, bin_batch);
        void *ws = n_owned_max ? arena.take(ws_bytes) : nullptr;
        auto bin_columns = [&](int ncols, const double *src_cols, int64_t ld_cols, uint8_t *dst_bins, int32_t *d_status) -> int {
            for (int c0 = 0; c0 < ncols; c0 += bin_batch) {
                const int b = ncols - c0 < bin_batch ? ncols - c0 : bin_batch;
                GRX_TRY(grx_internal_vertical_log_bin(n, b, src_cols + (size_t)c0 * ld_cols, ld_cols, nullptr, 0.5,
                                                      dst_bins + (size_t)c0 * n, n, nullptr, ws, ws_bytes, d_status, stream));
            }
            return GRX_OK;
        };
        double *owned = (comm && partial && n_owned_max) ? reinterpret_cast<double *>(arena.take((size_t)n_owned_max * n * 8)) : nullptr;
        uint8_t *owned_bins = (comm && n_owned_max) ? reinterpret_cast<uint8_t *>(arena.take((size_t)n_owned_max * n)) : nullptr;
        for (int j = 0; j < count; ++j) work.push_back(first_new + j);
        const int F = (int)work.size();
        // (a whole number of 256-byte units: the runtime clears an unaligned tail with a second fill launch)
        // + two status words behind the matrix (threshold walk failed | too many bins): the outcome flags of the binning
        // travel (and, sharded, are max-reduced word by word) with the distances, so a failed binning is a joint error
        // on every rank instead of a silent wrong drop list
        // ... and one word per new column: the bit width of its maximum when it holds exact integers (kind 1)
        // (not for the last generation the loop can reach: nothing will be gathered from its columns)
        const bool want_bits = count <= 64 && generation + 1 < max_generations && packed_allowed;
        const size_t tail_words = 2 + (size_t)(want_bits ? count : 0);
        const size_t dist_bytes = grx_align_up(((size_t)F * F + tail_words) * 4, 256);
        int32_t *d_dist = reinterpret_cast<int32_t *>(arena.take(dist_bytes));
        std::vector<int> drop_idx;
        if (!arena.overflow) {
            if (F >= 2) GRX_CHECK_HIP(hipMemsetAsync(d_dist, 0, dist_bytes, st));
            if (count && !comm) {
                GRX_TRY(bin_columns(count, block, n, bins, F >= 2 ? d_dist + (size_t)F * F : nullptr));
            } else if (count) {
                const double *src = block + (size_t)me * n;       // complete columns: the owned ones are a strided view
                int64_t ld_src = (int64_t)P * n;
                if (partial) {
                    // step 1: every rank's row slice of a column travels to the column's owner
                    GRX_TRY(grx_comm_columns_to_owners(comm, h_bounds, count, block, n, 8, owned, n, stream));
                    src = owned;
                    ld_src = n;
                }
                if (n_owned) {
                    GRX_TRY(bin_columns(n_owned, src, ld_src, owned_bins, F >= 2 ? d_dist + (size_t)F * F : nullptr));
                }
                // step 2: the owners' bins of this rank's rows come back
                GRX_TRY(grx_comm_owned_to_rows(comm, h_bounds, count, owned_bins, n, 1, bins, n, stream));
            }
            for (int j = 0; j < count; ++j) cols[first_new + j].bins = bins + (size_t)j * n;
            if (F >= 2 && tail_words > 2 && re > rb) {
                uint64_t mask = 0;
                for (int j = 0; j < count; ++j)
                    if (cols[first_new + j].kind == 1) mask |= 1ull << j;
                if (mask) GRX_TRY(grx_column_bits(n, count, block, n, rb, re, mask, d_dist + (size_t)F * F + 2, stream));
            }
            if (F >= 2) {
                std::vector<const uint8_t *> ptrs(F);
                for (int j = 0; j < F; ++j) ptrs[j] = cols[work[j]].bins;
                // the pruner only asks "distance <= generation number?" (prune.py:110-113)
                if (re > rb) GRX_TRY(grx_chebyshev(rb, re, F, 0, ptrs.data(), d_dist, generation, stream));
                if (comm) GRX_TRY(grx_comm_all_reduce(comm, d_dist, (size_t)F * F + tail_words, GRX_I32, GRX_MAX, stream));
                void *host = nullptr;
                GRX_TRY(pinned(((size_t)F * F + tail_words) * 4, &host));
                GRX_TRY(grx_fetch_begin(host, d_dist, ((size_t)F * F + tail_words) * 4, st));
                GRX_TRY(grx_fetch_wait(st));
                if (tail_words > 2)
                    for (int j = 0; j < count; ++j) {
                        const int32_t b = reinterpret_cast<const int32_t *>(host)[(size_t)F * F + 2 + j];
                        if (cols[first_new + j].kind == 1 && b >= 1 && b <= 62) cols[first_new + j].bits = b;
                    }
                const int32_t walk_failed = reinterpret_cast<const int32_t *>(host)[(size_t)F * F];
                const int32_t bins_overflow = reinterpret_cast<const int32_t *>(host)[(size_t)F * F + 1];
                if (walk_failed != 0 || bins_overflow != 0) {
                    grx_set_error("grx_refex_run: generation %d: vertical log binning failed on some rank (%s%s); "
                                  "GRX_BIN_SORT=1 selects the sort-based binning", generation,
                                  walk_failed ? "the sort-free threshold walk met an unmarked bucket" : "",
                                  bins_overflow ? " a column needs more than GRX_MAX_BINS bins" : "");
                    return GRX_ERR_UNSUPPORTED;
                }
                // identical distances on every rank -> identical decisions, no further agreement needed
                drop_idx = prune(cols, work, reinterpret_cast<const int32_t *>(host), generation, recorded);
            }
        }
        arena.rewind(persistent_top);                         // workspace and distance matrix are scratch
        std::vector<char> dropped(cols.size(), 0);
        for (int m : drop_idx) dropped[work[m]] = 1;
        const int working_before = F;
        work.erase(std::remove_if(work.begin(), work.end(), [&](int c) { return dropped[c] != 0; }), work.end());
        std::vector<int> kept;
        for (int j = 0; j < count; ++j)
            if (!dropped[first_new + j]) kept.push_back(first_new + j);
        if (comm && partial && !kept.empty() && !arena.overflow) {
            // step 3: the retained new columns -- the next generation's gather source and part of the result -- and
            // only those become complete on every rank (typically 1/3 to 1/6 of the candidates)
            std::vector<void *> ptrs;
            for (int c : kept) ptrs.push_back(const_cast<double *>(cols[c].data));
            GRX_TRY(grx_comm_all_gather_rows(comm, h_bounds, (int)ptrs.size(), ptrs.data(), 8, stream));
        }
        // extract.py:140 Index.difference: name-sorted iff the drop list is non-empty (pandas 2)
        if (!drop_idx.empty())
            std::stable_sort(kept.begin(), kept.end(), [&](int a, int b) { return cols[a].name < cols[b].name; });
        for (int c : kept) {
            if (n_out >= max_columns) { table_full = true; break; }
            cols[c].record_index = n_out;
            grx_refex_column &o = h_columns[n_out++];
            o.generation = generation;
            o.parent = cols[c].parent >= 0 ? cols[cols[c].parent].record_index : -1;
            o.agg = cols[c].agg;
            o.gen0_index = cols[c].generation == 0 ? c : -1;
            o.work_position = -1;
            o.d_col = cols[c].data;
        }
        recorded.push_back(kept);
        if (generation < max_gens) {
            grx_refex_generation &g = h_gens[generation];
            g.candidates = count;
            g.working = working_before;
            g.dropped = (int)drop_idx.size();
            g.retained = (int)kept.size();
            g.gather_row_bytes = gather_row_bytes;
        }
        return GRX_OK;
    };

    // ---- generation 0: the neighbourhood features the caller computed (base.py:18-26)
    for (int j = 0; j < f0; ++j) {
        GRX_REQUIRE(h_gen0_cols[j] && h_gen0_names[j], "grx_refex_run: generation-0 column %d is NULL", j);
        cols.push_back({h_gen0_names[j], 0, -1, -1, h_gen0_cols[j], nullptr, -1});
        cols.back().int32_exact = h_gen0_int32 && h_gen0_int32[j] != 0;
        if (cols.back().int32_exact) { cols.back().kind = 1; cols.back().base = j; }
    }

    {
        // binning wants one contiguous block: a scratch copy of the (separately allocated) input columns
        const size_t mark = arena.top;
        double *copy = reinterpret_cast<double *>(arena.take((size_t)f0 * n * 8));
        if (!arena.overflow) GRX_TRY(grx_gather_columns(n, f0, h_gen0_cols, copy, n, stream));
        (void)mark;
        // (the copy stays below the bins in the arena: a few columns, once per run)
        GRX_TRY(update(0, f0, copy, 0, false));
    }
    int generation = 0;
    for (int g = 1; g < max_generations && !arena.overflow && !table_full; ++g) {
        generation = g;
        const std::vector<int> &prev = recorded[g - 1];
        const int f = (int)prev.size();
        const int count = n_aggs * f;
        const int first_new = (int)cols.size();
        double *block = count ? reinterpret_cast<double *>(arena.take((size_t)count * n * 8)) : nullptr;
        if (count) {
            // candidate order: every column under the first aggregation, then the second, ... (extract.py:158-162)
            int a_sum = -1;
            for (int a = 0; a < n_aggs; ++a)
                if (h_aggs[a] == GRX_AGG_SUM) a_sum = a;
            for (int a = 0; a < n_aggs; ++a)
                for (int j = 0; j < f; ++j) {
                    Column child{cols[prev[j]].name + "(" + AGG_NAMES[h_aggs[a]] + ")", g, prev[j], h_aggs[a],
                                 block ? block + ((size_t)a * f + j) * n : nullptr, nullptr, -1};
                    // a neighbour sum of exact integers is an exact integer while it stays below 2^53; its mean is
                    // fl(S / d) of that sum (the sum candidate of the same parent exists when 'sum' is among the aggs)
                    const Column &par = cols[prev[j]];
                    const bool exact_sum = par.kind == 1 && par.bits >= 1 && par.bits + deg_bits <= 53;
                    if (exact_sum && h_aggs[a] == GRX_AGG_SUM) { child.kind = 1; child.base = first_new + a * f + j; }
                    if (exact_sum && h_aggs[a] == GRX_AGG_MEAN && a_sum >= 0) { child.kind = 2; child.base = first_new + a_sum * f + j; }
                    cols.push_back(std::move(child));
                }
            const size_t mark = arena.top;
            const int ldr = grx_aggregate_ldr(f);
            const bool need_var = has[GRX_AGG_VAR] || has[GRX_AGG_STD];
            // integer gather source (16- / 32-byte rows) when every parent is an exact int32 column and only sums /
            // means are wanted: any summation order gives the reference's bits there (grx.h, grx_aggregate_i32)
            bool only_sum_mean = true;
            for (int a = 0; a < n_aggs; ++a) only_sum_mean = only_sum_mean && (h_aggs[a] == GRX_AGG_SUM || h_aggs[a] == GRX_AGG_MEAN);
            bool int_rows = only_sum_mean && grx_aggregate_i32_ok(plan, f);
            for (int j = 0; j < f && int_rows; ++j) int_rows = cols[prev[j]].int32_exact;
            const int ldi = int_rows ? grx_aggregate_ldi(f) : 0;
            // bit-packed integer rows (grx_aggregate_packed): every parent an exact integer column or the mean of one,
            // column maxima known, fields within two 64-bit words; sharded: the base columns must be complete on this
            // rank (generation 0, or retained -- those were all-gathered)
            grx_packed_layout layout{};
            std::vector<const double *> field_cols;
            int packed_bytes = 0;
            if (packed_allowed && only_sum_mean && f <= 8) {
                std::vector<int> fields;
                bool ok = true, any_mean = false;
                for (int j = 0; j < f && ok; ++j) {
                    const Column &par = cols[prev[j]];
                    const int b = par.kind == 1 ? prev[j] : (par.kind == 2 ? par.base : -1);
                    ok = b >= 0 && cols[b].bits >= 1 && cols[b].bits + deg_bits <= 53 &&
                         (!comm || cols[b].generation == 0 || cols[b].record_index >= 0);
                    if (!ok) break;
                    int k = (int)(std::find(fields.begin(), fields.end(), b) - fields.begin());
                    if (k == (int)fields.size()) { fields.push_back(b); ok = fields.size() <= 7; }
                    layout.out_field[j] = k;
                    layout.out_is_mean[j] = par.kind == 2;
                    any_mean = any_mean || par.kind == 2;
                }
                if (ok) {
                    layout.n_fields = (int)fields.size();
                    layout.n_out = f;
                    layout.degree_bits = any_mean ? deg_bits : 0;
                    for (size_t k = 0; k < fields.size(); ++k) {
                        layout.field_bits[k] = cols[fields[k]].bits;
                        field_cols.push_back(cols[fields[k]].data);
                    }
                    packed_bytes = grx_packed_row_bytes(&layout);
                }
            }
            gather_row_bytes = packed_bytes ? packed_bytes : (int_rows ? ldi * 4 : ldr * 8);
            double *rows = reinterpret_cast<double *>(arena.take(packed_bytes ? (size_t)n * packed_bytes
                                                                 : int_rows ? (size_t)n * ldi * 4 : (size_t)n * ldr * 8));
            double *mean_scratch = (need_var && !has[GRX_AGG_MEAN]) ? reinterpret_cast<double *>(arena.take((size_t)f * n * 8))
                                                                    : nullptr;
            const size_t med_bytes = has[GRX_AGG_MEDIAN] ? grx_aggregate_median_workspace_bytes(nnz_rows) : 0;
            void *med_ws = has[GRX_AGG_MEDIAN] ? arena.take(med_bytes) : nullptr;
            if (!arena.overflow) {
                auto out_of = [&](int agg) -> double * {
                    for (int a = 0; a < n_aggs; ++a)
                        if (h_aggs[a] == agg) return block + (size_t)a * f * n;
                    return nullptr;
                };
                std::vector<const double *> ptrs(f);
                for (int j = 0; j < f; ++j) ptrs[j] = cols[prev[j]].data;
                double *d_mean = has[GRX_AGG_MEAN] ? out_of(GRX_AGG_MEAN) : mean_scratch;
                if (packed_bytes) {
                    GRX_TRY(grx_pack_fields(n, &layout, field_cols.data(), d_row_ptr, rows, stream));
                    GRX_TRY(grx_aggregate_packed(plan, d_row_ptr, d_agg_col, &layout, rows, rb, re, out_of(GRX_AGG_SUM), d_mean, n,
                                                 stream));
                } else if (int_rows) {
                    int32_t *irows = reinterpret_cast<int32_t *>(rows);
                    GRX_TRY(grx_pack_rows_i32(n, f, ptrs.data(), irows, ldi, stream));
                    GRX_TRY(grx_aggregate_i32(plan, d_row_ptr, d_agg_col, f, irows, ldi, rb, re, out_of(GRX_AGG_SUM), d_mean, n,
                                              stream));
                } else {
                    GRX_TRY(grx_pack_rows(n, f, ptrs.data(), rows, ldr, stream));
                    if (has[GRX_AGG_SUM] || d_mean)
                        GRX_TRY(grx_aggregate(plan, d_row_ptr, d_agg_col, f, rows, ldr, rb, re, out_of(GRX_AGG_SUM), d_mean, n, stream));
                }
                if (need_var)
                    GRX_TRY(grx_aggregate_var(plan, d_row_ptr, d_agg_col, f, rows, ldr, rb, re, d_mean, out_of(GRX_AGG_VAR),
                                              out_of(GRX_AGG_STD), n, stream));
                if (has[GRX_AGG_MIN] || has[GRX_AGG_MAX])
                    GRX_TRY(grx_aggregate_minmax(plan, d_row_ptr, d_agg_col, f, rows, ldr, rb, re, out_of(GRX_AGG_MIN),
                                                 out_of(GRX_AGG_MAX), n, stream));
                // the aggregations beyond the tuned six (csrc/grx_aggx.hip, fp64 columns; integer columns with 'prod' need
                // int64 arithmetic and are driven per kernel by the host, features/extract.py)
                if (has[GRX_AGG_PROD])
                    GRX_TRY(grx_aggregate_prod(d_row_ptr, d_agg_col, f, rows, ldr, rb, re, out_of(GRX_AGG_PROD), n, stream));
                if (has[GRX_AGG_MEDIAN])
                    GRX_TRY(grx_aggregate_median(d_row_ptr, d_agg_col, f, rows, ldr, rb, re, out_of(GRX_AGG_MEDIAN), n, med_ws,
                                                 med_bytes, stream));
                if (has[GRX_AGG_COUNT])
                    GRX_TRY(grx_aggregate_count(d_row_ptr, f, rb, re, 0, out_of(GRX_AGG_COUNT), n, stream));
                if (has[GRX_AGG_SIZE])
                    GRX_TRY(grx_aggregate_count(d_row_ptr, f, rb, re, 0, out_of(GRX_AGG_SIZE), n, stream));
            }
            arena.rewind(mark);                               // the gather source is scratch
        }
        GRX_TRY(update(first_new, count, block, g, comm != nullptr));
        if (recorded[g].empty()) break;                       // extract.py:86-87
    }
    if (arena_needed) *arena_needed = arena.peak;
    if (arena.overflow) {
        grx_set_error("grx_refex_run: arena of %zu bytes is too small (needs at least %zu so far)", arena.capacity(), arena.peak);
        return GRX_ERR_WORKSPACE;
    }
    if (table_full) {
        grx_set_error("grx_refex_run: more than %d recorded columns", max_columns);
        return GRX_ERR_WORKSPACE;
    }
    for (size_t pos = 0; pos < work.size(); ++pos)
        if (cols[work[pos]].record_index >= 0) h_columns[cols[work[pos]].record_index].work_position = (int)pos;
    *n_columns = n_out;
    *generation_count = generation;
    GRX_CHECK_HIP(hipStreamSynchronize(st));
    return GRX_OK;
}

}  // extern "C"
