/*
 * grx.h -- C ABI of libgrx.so, the MI355X (gfx950) ReFeX / RolX hot-path library.
 *
 * The reference (dkaslovsky/GraphRole) is pure Python and has no FFI of its own; the seam this
 * library sits under is the pair of public classes and the graph-adapter ABC
 * (SURVEY.md section 8b).  Every entry point below names the reference lines it replaces.
 * graphrole_amd/ binds these symbols with ctypes; INTEGRATION.md shows the stub a GraphRole
 * maintainer would add.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes only.  Every function returns 0 on success and a
 *     negative grx_status otherwise; grx_last_error() returns a thread-local message.
 *   - Pointers named d_* are DEVICE pointers (hipMalloc'ed by the caller, by PyTorch's caching
 *     allocator, or by grx_dev_malloc).  h_* are host pointers (small tables / matrices that
 *     are consumed before the call returns).
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream).  All work is
 *     stream-ordered; nothing here synchronises unless the name says so.
 *   - Feature columns are column-major: one contiguous fp64 array of n values per column.
 *     "rows" buffers are row-major n x f (the gather source of the aggregation kernel).
 *   - Graph = CSR of the out-adjacency, rows in sorted-label order, column indices ascending in
 *     each row, int64 row_ptr[n+1], int32 col[nnz], optional fp64 w[nnz] (NULL = weight 1).
 *   - Node-range sharding: kernels that produce one value per node take [row_begin,row_end)
 *     and touch only those output rows, so a rank computes its own slice of a replicated graph.
 */
#ifndef GRX_H
#define GRX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GRX_VERSION 600          /* 0.6.0: grx_kmeans1d reports into int32[4] (d_info[3] = consistency faults of the seeding) */
#define GRX_MAX_BINS 128         /* upper bound on vertical-log bins (n < 2^63 gives < 70) */
#define GRX_MAX_ROLES 32         /* NMF rank limit of the device kernels: 1 .. 16 fused fp64-MFMA passes; 17 .. 32
                                    a composed update (several times the traffic), then with n_roles + features <= 480 */
#define GRX_MAX_NMF_FEATURES 480 /* NMF feature-count limit of the device kernels (fast paths: 120) */

typedef enum {
    GRX_OK = 0,
    GRX_ERR_INVALID = -1,        /* bad argument */
    GRX_ERR_HIP = -2,            /* HIP runtime error (message has hipGetErrorString) */
    GRX_ERR_WORKSPACE = -3,      /* workspace too small */
    GRX_ERR_UNSUPPORTED = -4,    /* shape outside the compiled limits */
    GRX_ERR_DEGENERATE = -5      /* numerically degenerate input (e.g. an all-zero feature matrix) */
} grx_status;

/* Environment switches, each read once per process; all of them select between formulations that the tests compare
 * on the same inputs (same results), none changes what is computed:
 *   GRX_READBACK=memcpy        the few KB grx_refex_run / grx_nmf_fit read back before they decide what to launch next
 *                              go through hipMemcpyAsync + hipStreamSynchronize instead of mapped host memory and a flag
 *   GRX_BIN_BID_MIN_N=<rows>   column height from which the binning keeps 12-bit bucket ids (default 2 500 000)
 *   GRX_TRI_ROUNDS=<k>         workgroups of grx_triangle_counts = k x what fills the chip (default 4)
 *   GRX_KMEANS_*               see grx_kmeans1d below
 *   GRX_FORCE_COLLECTIVES=1    a one-rank communicator issues every exchange of the sharded path (bench.py)
 */
/* ------------------------------------------------------------------ runtime helpers ---- */
int grx_version(void);
const char *grx_last_error(void);
/* Device properties: CU count, wavefront size, gcn arch name (buf may be NULL). */
int grx_device_info(int *cu_count, int *wave_size, char *arch_buf, size_t arch_buf_len);
/* For callers without their own allocator (numpy-only hosts). */
int grx_dev_malloc(void **d_out, size_t bytes);
int grx_dev_free(void *d_ptr);
int grx_memcpy_h2d(void *d_dst, const void *h_src, size_t bytes, void *stream);
int grx_memcpy_d2h(void *h_dst, const void *d_src, size_t bytes, void *stream);
int grx_memset(void *d_dst, int value, size_t bytes, void *stream);
int grx_stream_sync(void *stream);
/*
 * Bulk host <-> device copies for PAGEABLE host memory (numpy arrays) at close to link rate: chunks through a ring
 * of pinned staging buffers, the host-side memcpy on a small thread pool while the next chunk is on the link.  The
 * only bulk data of the two public calls that crosses PCIe is the result table extract_features() hands to
 * extract_role_factors() (graphrole/roles/extract.py:59-93) and the edge arrays.  Synchronous on return.
 * grx_upload_i64_as_i32 narrows int64 edge arrays (numpy's default) while staging; GRX_ERR_INVALID if a value
 * does not fit.  grx_host_checksums: 64-bit content hash of each of ncols host columns (col_bytes each, a multiple
 * of 8; column c at h_base + c * stride_bytes) -- how a result table is recognised as unmodified when it comes back.
 * grx_min_value: min over the n x F entries of a feature-major device matrix (NaN if any entry is NaN): sklearn's
 * "Negative values in data passed to NMF" check (_nmf.py:283) without a host pass.
 */
int grx_download(void *h_dst, const void *d_src, size_t bytes, void *stream);
int grx_upload(void *d_dst, const void *h_src, size_t bytes, void *stream);
int grx_upload_i64_as_i32(int32_t *d_dst, const int64_t *h_src, size_t count, void *stream);
int grx_host_checksums(const void *h_base, int ncols, size_t col_bytes, size_t stride_bytes, uint64_t *h_out);
/* The index numpy's RandomState.choice(m, p = uniform) returns for its one uniform draw u -- searchsorted(cumsum(full(m,
 * 1 / m)) / total, u, 'right') -- without the three m-element arrays: the first k-means++ seed of sklearn's KMeans
 * (_kmeans.py:221) as grx_kmeans1d needs it. */
int grx_host_uniform_choice(int64_t m, double u, int64_t *index);
size_t grx_min_value_workspace_bytes(void);
int grx_min_value(int64_t n, int F, const double *d_X, int64_t ld, double *d_out, void *d_workspace,
                  size_t workspace_bytes, void *stream);

/* Stream-ordered event timing helpers (bench.py measures kernels on the launch stream). */
int grx_event_create(void **event_out);
int grx_event_destroy(void *event);
int grx_event_record(void *event, void *stream);
int grx_event_elapsed_ms(void *start, void *stop, float *ms_out);   /* synchronises on stop */

/* Launches the one-thread `grx_marker_kernel` on `stream`: a named fence in a rocprofv3 kernel trace (tests
 * delimit the product path with it, tools/step_timeline.py a step). */
int grx_trace_marker(int tag, void *stream);

/*
 * Per-kernel timing with HIP events recorded on the launch stream around every kernel launch
 * of this library (bench.py's roofline numbers).  Off by default.  Kernel ids are dense in
 * [0, grx_profile_kernel_count()); grx_profile_kernel_name gives the name rocprofv3 shows.
 * grx_profile_read synchronises on the recorded events and returns the totals since the last
 * grx_profile_reset.
 */
int grx_profile_enable(int on);
int grx_profile_enabled(void);                  /* 1 while the event profiler is on */
int grx_profile_select(uint64_t kernel_mask);   /* bit i = time kernel id i; 0 = all (default) */
int grx_profile_reset(void);
int grx_profile_kernel_count(void);
const char *grx_profile_kernel_name(int id);
int grx_profile_read(int id, double *total_ms, long long *launches);

/* ------------------------------------------------------------------ node-range sharding -- */
/*
 * NEW (the reference is single-process): one process per GPU, the graph and the
 * feature columns replicated, rank p computes node rows [h_bounds[p], h_bounds[p+1]) of every per-node
 * kernel.  A grx_comm is the transport between the ranks; the whole-loop drivers (grx_refex_run,
 * grx_nmf_fit) take one and issue their exchanges themselves, stream-ordered between the kernels.
 *
 *   grx_comm_create_rccl       RCCL over xGMI.  librccl.so.1 is resolved at run time (the copy the process
 *                              already loaded -- e.g. PyTorch's -- else the system one; GRX_RCCL_PATH overrides),
 *                              so single-GPU users carry no RCCL dependency.  Bootstrap as with NCCL: rank 0
 *                              calls grx_comm_rccl_unique_id, hands the GRX_COMM_ID_BYTES to every rank through
 *                              any channel it has (torch.distributed, MPI, a file), every rank calls create.
 *                              flags & GRX_COMM_SELF_VIA_TRANSPORT: transfers of a rank to itself go through
 *                              ncclSend / ncclRecv as well (functional test of the RCCL calls on a one-GPU box).
 *   grx_comm_create_callbacks  any other transport (the gloo-staged test transport of tests/, MPI ...): the two
 *                              primitives below as function pointers.
 * Primitives (stream-ordered; d_* are device pointers):
 *   grx_comm_all_reduce        in place over `count` elements of grx_dtype with GRX_SUM / GRX_MAX
 *   grx_comm_exchange          a group of point-to-point transfers.  Between one pair of ranks, sends and
 *                              receives are matched in the order they appear in `ops`.
 * Composites over a row partition h_bounds (host int64[world + 1]); column j of a block = base + j * ld
 * elements; rank q OWNS columns q, q + world, ... of a block.  None of them packs: every transfer reads /
 * writes the row slice of a column where it lies.
 *   grx_comm_all_gather_rows     every column (host table of device pointers) holds this rank's rows ->
 *                                complete on every rank (the next generation's gather source, W of the NMF)
 *   grx_comm_columns_to_owners   d_block [ncols x ld] with this rank's rows valid -> d_owned [n_owned x ld_owned]:
 *                                the owned columns with ALL rows (the owner bins whole columns)
 *   grx_comm_owned_to_rows       the inverse for per-column results (uint8 bins): d_owned whole owned columns
 *                                -> d_block [ncols x ld] with this rank's rows of EVERY column
 * Timing (bench.py --gpus N): grx_comm_timing(comm, 1) records HIP events around every primitive;
 * grx_comm_timing_read folds them into calls / milliseconds per kind (grx_comm_kind) and synchronises.
 */
typedef struct grx_comm grx_comm;
typedef enum { GRX_F64 = 0, GRX_I32 = 1, GRX_I64 = 2, GRX_U8 = 3 } grx_dtype;
typedef enum { GRX_SUM = 0, GRX_MAX = 1 } grx_reduce_op;
typedef enum { GRX_COMM_ALL_REDUCE = 0, GRX_COMM_EXCHANGE = 1, GRX_COMM_ALL_GATHER_ROWS = 2,
               GRX_COMM_COLUMNS_TO_OWNERS = 3, GRX_COMM_OWNED_TO_ROWS = 4, GRX_COMM_KINDS = 5 } grx_comm_kind;
typedef struct { int is_recv; int peer; void *d_ptr; size_t bytes; } grx_p2p_op;
typedef int (*grx_all_reduce_fn)(void *user, void *d_buf, size_t count, int dtype, int op, void *stream);
typedef int (*grx_exchange_fn)(void *user, int n_ops, const grx_p2p_op *ops, void *stream);
#define GRX_COMM_ID_BYTES 128
#define GRX_COMM_SELF_VIA_TRANSPORT 1
int grx_comm_rccl_unique_id(void *h_id);
int grx_comm_create_rccl(const void *h_id, int rank, int world, int flags, grx_comm **out);
int grx_comm_create_callbacks(int rank, int world, grx_all_reduce_fn all_reduce, grx_exchange_fn exchange, void *user,
                              grx_comm **out);
int grx_comm_destroy(grx_comm *comm);
int grx_comm_rank(const grx_comm *comm);
int grx_comm_world(const grx_comm *comm);
int grx_comm_all_reduce(grx_comm *comm, void *d_buf, size_t count, int dtype, int op, void *stream);
int grx_comm_exchange(grx_comm *comm, int n_ops, const grx_p2p_op *ops, void *stream);
int grx_comm_all_gather_rows(grx_comm *comm, const int64_t *h_bounds, int ncols, void *const *h_col_ptrs,
                             int elem_bytes, void *stream);
int grx_comm_columns_to_owners(grx_comm *comm, const int64_t *h_bounds, int ncols, const void *d_block, int64_t ld,
                               int elem_bytes, void *d_owned, int64_t ld_owned, void *stream);
int grx_comm_owned_to_rows(grx_comm *comm, const int64_t *h_bounds, int ncols, const void *d_owned, int64_t ld_owned,
                           int elem_bytes, void *d_block, int64_t ld, void *stream);
int grx_comm_timing(grx_comm *comm, int on);
int grx_comm_timing_read(grx_comm *comm, int kind, long long *calls, double *ms);
int grx_comm_timing_reset(grx_comm *comm);

/* ------------------------------------------------------------------ graph ingest -------- */
/*
 * Edge list -> the CSR structures of this library, on the device.  Replaces the host-side construction the
 * adapters need before anything can run (graphrole/graph/interface/networkx.py:36-46 get_nodes / get_neighbors,
 * base.py:18-26 row order): ~1 s of numpy per 10 M edges against milliseconds here.
 *   d_src / d_dst  int32[m] row indices (0..n-1 = sorted node labels); every edge once (undirected: either
 *                  orientation; a self-loop once); d_w fp64[m] or NULL.
 *   internal order: rows sorted by out-degree descending, ties by label (hub rows become a prefix):
 *                  d_perm[i] = label row of internal row i, d_inv its inverse.
 *   d_row_ptr int64[n+1], d_col int32[nnz] ascending per row, d_wcol fp64[nnz] (aligned with d_col, NULL if
 *                  unweighted), d_agg_col int32[nnz] = the same rows with the neighbours in ADJACENCY order (the
 *                  order the incident edges appear in the edge list = G[node] order of a graph built by
 *                  add_edge in that order): the summation order of grx_aggregate.
 *   nnz            2m - #self-loops (undirected) or m (directed): the caller counts the loops.
 *   directed:      d_t_row_ptr / d_t_col / d_t_w = the transposed (in-) adjacency, ascending (weighted in-degree).
 * grx_orient_count / grx_orient_fill: the degree-oriented copy for grx_triangle_counts (arc u -> v kept iff
 * (d'(u), u) < (d'(v), v), d' = degree without the self-loop) and its per-arc table; count first (o_row_ptr,
 * o_nnz = o_row_ptr[n] read back by the caller), then fill with the same workspace.
 */
size_t grx_ingest_workspace_bytes(int64_t n, int64_t m, int directed);
int grx_ingest(int64_t n, int64_t m, const int32_t *d_src, const int32_t *d_dst, const double *d_w, int directed,
               int64_t nnz, int32_t *d_perm, int32_t *d_inv, int64_t *d_row_ptr, int32_t *d_col, double *d_wcol,
               int32_t *d_agg_col, int64_t *d_t_row_ptr, int32_t *d_t_col, double *d_t_w, void *d_workspace,
               size_t workspace_bytes, void *stream);
/* out[c * ld + i] = col_c[d_index[i]] for F columns (host table of device pointers): the result table from the
 * internal row order back to label order (d_index = inv), one contiguous F x ld block for a single copy out. */
int grx_permute_columns(int64_t n, int F, const double *const *h_col_ptrs, const int32_t *d_index, double *d_out,
                        int64_t ld, void *stream);
size_t grx_orient_workspace_bytes(int64_t n);
int grx_orient_count(int64_t n, const int64_t *d_row_ptr, const int32_t *d_col, int64_t *d_o_row_ptr, void *d_workspace,
                     size_t workspace_bytes, void *stream);
int grx_orient_fill(int64_t n, const int64_t *d_row_ptr, const int32_t *d_col, const int64_t *d_o_row_ptr, int64_t o_nnz,
                    int32_t *d_o_col, uint64_t *d_o_arc, void *d_workspace, size_t workspace_bytes, void *stream);

/* ------------------------------------------------------------------ generation 0 -------- */
/*
 * Weighted row sums of a CSR.  Replaces NetworkxInterface._get_local_features
 * (graphrole/graph/interface/networkx.py:48-63): out-degree / undirected degree from the CSR,
 * in-degree from the transposed CSR.  add_self_loop != 0 adds the diagonal entry a second time
 * (networkx counts an undirected self-loop twice).  Deterministic summation order.
 */
int grx_row_sums(int64_t n, const int64_t *d_row_ptr, const int32_t *d_col, const double *d_w,
                 int add_self_loop, int64_t row_begin, int64_t row_end, double *d_out,
                 void *stream);

/* out[i] = a[i] + b[i]  (total_degree = in_degree + out_degree, networkx.py:57). */
int grx_add_columns(int64_t n, const double *d_a, const double *d_b, double *d_out, void *stream);

/*
 * Ego-net features.  Replaces NetworkxInterface._get_egonet_features + _get_edge_sum
 * (graphrole/graph/interface/networkx.py:71-83,115-123).  ego(v) = {v} U row(v);
 * internal[v] = weight of edges with both ends in ego(v) (undirected: each edge and self-loop
 * once; directed: every arc); external[v] = weight of edges leaving ego(v).
 * d_rowsum: plain weighted row sums of the same CSR (grx_row_sums with add_self_loop = 0, all n
 * rows); required when d_w != NULL, ignored otherwise.
 * nnz: number of CSR entries (d_row_ptr[n]); d_workspace: grx_egonet_workspace_bytes(n, nnz) bytes of device scratch (round 5: one aligned 128-byte slot per node --
 * its row sum, where its row begins, how long it is and its first 28 ids -- so that a member of an ego set costs one
 * aligned request instead of three or four, and the lists of the rows the wavefront / workgroup kernels own).
 * Weights are read for the arcs that END inside the ego set only; what leaves it is rowsum(a) minus that, exactly 0
 * for a closed row, and added arc by arc when the difference would cancel more than six bits.
 */
size_t grx_egonet_workspace_bytes(int64_t n, int64_t nnz);
int grx_egonet_features(int64_t n, int64_t nnz, const int64_t *d_row_ptr, const int32_t *d_col,
                        const double *d_w, const double *d_rowsum, int directed,
                        int64_t row_begin, int64_t row_end, double *d_internal,
                        double *d_external, void *d_workspace, size_t workspace_bytes, void *stream);

/*
 * Fast path of the same ego-net features for UNWEIGHTED UNDIRECTED graphs (BASELINE configs
 * 1-4), exact in integer arithmetic:  with d' the loop-free degrees, L the self-loop flags and
 * T(v) the number of triangles through v,
 *     internal(v) = d'(v) + T(v) + sum_{a in ego(v)} L(a),
 *     external(v) = sum_{a in ego(v)} d'(a) - 2 (d'(v) + T(v)).
 * grx_triangle_counts accumulates T (caller zero-fills d_T, uint64[n]) from the degree-oriented
 * graph (CSR that keeps arc u->v iff (d'(u),u) < (d'(v),v), columns ascending); only source rows
 * [row_begin,row_end) are processed, so ranks can split the arcs and all-reduce(SUM) d_T.
 * d_o_arc: one uint64 per oriented arc k = u->v (the arcs are the kernel's work items; the table is read sequentially
 * instead of dependent random lookups): o_row_ptr[v] | min(d+(v), 1023) << 32 | min(d+(u), 1023) << 42 |
 * min(k - o_row_ptr[u], 1023) << 52 -- where the target's own list lies and how long it is, how long the source's
 * list is and where in it the arc stands (requires o_nnz < 2^32; arcs with a field at 1023 are looked up from
 * o_row_ptr).
 * grx_egonet_unweighted then writes rows [row_begin,row_end); d_scratch: int32[n] (degree < 2^30).
 * d_hub_rows / n_hub_rows (optional): ascending rows with more than hub_degree neighbours; they get a
 * workgroup each instead of an 8-lane group.
 */
int grx_triangle_counts(int64_t n, const int64_t *d_o_row_ptr, const int32_t *d_o_col,
                        const uint64_t *d_o_arc, int64_t row_begin, int64_t row_end, uint64_t *d_T,
                        void *stream);
int grx_egonet_unweighted(int64_t n, const int64_t *d_row_ptr, const int32_t *d_col,
                          const uint64_t *d_T, int64_t row_begin, int64_t row_end,
                          double *d_internal, double *d_external, int32_t *d_scratch,
                          const int32_t *d_hub_rows, int64_t n_hub_rows, int64_t hub_degree, void *stream);

/* ------------------------------------------------------------------ recursion ----------- */
/*
 * Pack f feature columns into the row-major gather source of grx_aggregate.
 * h_col_ptrs: HOST array of f device pointers (each an fp64 column of n values); the table
 * travels as a kernel argument, 128 columns per launch.
 * d_rows: n x ldr row-major, ldr >= f; columns f..ldr-1 are zero-filled.  grx_aggregate wants
 * ldr = grx_aggregate_ldr(f): 2, 4, 8 or a multiple of 16 doubles, so that a feature row is a
 * 16/32/64-byte slice of one cache line or a whole number of 128-byte lines.
 */
int grx_aggregate_ldr(int f);
int grx_pack_rows(int64_t n, int f, const double *const *h_col_ptrs, double *d_rows, int ldr,
                  void *stream);

/*
 * ReFeX neighbour aggregation.  Replaces RecursiveFeatureExtractor._get_next_features
 * (graphrole/features/extract.py:98-119):
 *     sum[c][v]  = sum_{u in row(v)} rows[u][c]
 *     mean[c][v] = sum[c][v] / |row(v)|      (0 when row(v) is empty)
 * BIT-EXACT with the reference: the reference evaluates every sum with Series.sum(), i.e. numpy's
 * pairwise summation over the neighbours in G[node] order (extract.py:108-113); the kernels add
 * in exactly that tree (8 interleaved accumulators, ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)), trailing
 * elements one by one, rows with more than 128 neighbours as a binary tree of blocks, rows with
 * more than 8192 as a running total over 8192-element chunks like ndarray.sum()).  d_col must
 * therefore list every row's neighbours in the order the reference visits them (adjacency /
 * insertion order; any order gives a correct sum, only this one gives the reference's bits).
 *
 * grx_aggregate_plan: per-graph preprocessing (lane-group width from the mean degree, block list
 * of the rows with more than 128 neighbours).  h_row_ptr is a HOST array int64[n+1].  The plan owns
 * device memory (block list + scratch); one stream at a time per plan.
 *
 * d_rows: n x ldr row-major, ldr = grx_aggregate_ldr(f), 128-byte aligned (all n rows are
 * needed: neighbours may live on any rank's slice).  A neighbour row is fetched by ldr/2
 * adjacent lanes (16 bytes each) so one request covers the whole row.
 * d_sum / d_mean: column-major, column c at d_sum + c*ld (ld >= n); only rows
 * [row_begin,row_end) are written.  Either output may be NULL.
 * grx_aggregate_minmax: column-wise minimum / maximum over the neighbours (aggs 'min' / 'max',
 * extract.py:36-47); 0 for a row without neighbours (fillna, :113).
 */
typedef struct grx_aggregate_plan grx_aggregate_plan;
int grx_aggregate_plan_create(int64_t n, const int64_t *h_row_ptr, grx_aggregate_plan **plan);
void grx_aggregate_plan_destroy(grx_aggregate_plan *plan);
int grx_aggregate_plan_info(const grx_aggregate_plan *plan, int64_t *n_long_rows, int64_t *n_blocks,
                            int *lanes_per_row);
int grx_aggregate_plan_set_lanes(grx_aggregate_plan *plan, int lanes_per_row);   /* 4, 8, 16 or 32 */
int grx_aggregate(const grx_aggregate_plan *plan, const int64_t *d_row_ptr, const int32_t *d_col, int f,
                  const double *d_rows, int ldr, int64_t row_begin, int64_t row_end,
                  double *d_sum, double *d_mean, int64_t ld, void *stream);
/* aggs 'var' / 'std' (pandas: sample variance, ddof = 1): with d_mean the 'mean' output of grx_aggregate for
 * the same rows, var = pairwise-sum((mean - x)^2) / (count - 1) and std = sqrt(var) -- pandas' nanvar, bit for
 * bit; 0 for rows with fewer than two neighbours (NaN -> fillna(0)).  Either output may be NULL. */
int grx_aggregate_var(const grx_aggregate_plan *plan, const int64_t *d_row_ptr, const int32_t *d_col, int f,
                      const double *d_rows, int ldr, int64_t row_begin, int64_t row_end, const double *d_mean,
                      double *d_var, double *d_std, int64_t ld, void *stream);
int grx_aggregate_minmax(const grx_aggregate_plan *plan, const int64_t *d_row_ptr, const int32_t *d_col, int f,
                         const double *d_rows, int ldr, int64_t row_begin, int64_t row_end,
                         double *d_min, double *d_max, int64_t ld, void *stream);
/*
 * The same sums / means for an INTEGER gather source: when every source column holds exact non-negative integers
 * below 2^31 (the generation-0 block of an unweighted graph: degrees and ego-net edge counts are bounded by nnz), the
 * neighbour sums are integers below 2^53 and ANY order of additions gives the bits of the reference's pairwise tree.
 * The gather source is then n x ldi int32 (ldi = grx_aggregate_ldi(f): 4 or 8, i.e. 16- / 32-byte rows instead of 32 /
 * 64): four times as many rows per cache line and per MiB of L2.  grx_aggregate_i32_ok: the plan's longest row keeps
 * max_degree * 2^31 <= 2^53 and f <= 8.  Outputs as grx_aggregate.
 */
int grx_aggregate_ldi(int f);
int grx_aggregate_i32_ok(const grx_aggregate_plan *plan, int f);
int grx_pack_rows_i32(int64_t n, int f, const double *const *h_col_ptrs, int32_t *d_rows, int ldi, void *stream);
int grx_aggregate_i32(const grx_aggregate_plan *plan, const int64_t *d_row_ptr, const int32_t *d_col, int f,
                      const int32_t *d_rows, int ldi, int64_t row_begin, int64_t row_end, double *d_sum, double *d_mean,
                      int64_t ld, void *stream);
/*
 * INTEGER feature columns with the reference's int64 semantics, 'median', 'count' / 'size' (csrc/grx_aggx.hip;
 * extract.py:26,47,111 passes any pandas-aggregatable to DataFrame.agg).
 *   grx_convert_*          switch a column between fp64 values and int64 bits (numpy astype semantics)
 *   grx_aggregate_i64      d_rows: n x ldr row-major int64 (grx_pack_rows moves bits); wrapping sum / product, min, max
 *                          over the neighbours -> int64 columns (any output may be NULL); min / max of no neighbours: 0
 *   grx_aggregate_count    number of neighbours into f columns, as fp64 values or int64 bits
 *   grx_aggregate_median   numpy's median of the neighbours' values (middle element, or (a + b) / 2 of the two middle
 *                          ones; 0 for no neighbours); workspace grx_aggregate_median_workspace_bytes(nnz of the rows)
 */
int grx_convert_i64_to_f64(int64_t n, const int64_t *d_in, double *d_out, void *stream);
int grx_convert_f64_to_i64(int64_t n, const double *d_in, int64_t *d_out, void *stream);
int grx_aggregate_i64(const int64_t *d_row_ptr, const int32_t *d_col, int f, const int64_t *d_rows, int ldr,
                      int64_t row_begin, int64_t row_end, int64_t *d_sum, int64_t *d_prod, int64_t *d_min, int64_t *d_max,
                      int64_t ld, void *stream);
int grx_aggregate_count(const int64_t *d_row_ptr, int f, int64_t row_begin, int64_t row_end, int as_i64, double *d_out,
                        int64_t ld, void *stream);
size_t grx_aggregate_median_workspace_bytes(int64_t nnz);
int grx_aggregate_median(const int64_t *d_row_ptr, const int32_t *d_col, int f, const double *d_rows, int ldr,
                         int64_t row_begin, int64_t row_end, double *d_median, int64_t ld, void *d_workspace,
                         size_t workspace_bytes, void *stream);
/* agg 'prod' (extract.py:36-47 with Series.prod): left-to-right product over the neighbours in the order
 * of d_col, 1 for a row without neighbours.  fp64 arithmetic: exact for integer columns below 2^53 (the
 * reference multiplies int64 columns in int64 and wraps silently beyond 2^63; the caller checks). */
int grx_aggregate_prod(const int64_t *d_row_ptr, const int32_t *d_col, int f, const double *d_rows, int ldr,
                       int64_t row_begin, int64_t row_end, double *d_prod, int64_t ld, void *stream);

/*
 * The whole generation loop below the ABI (single GPU).  Replaces RecursiveFeatureExtractor.extract_features /
 * _get_next_features / _update (graphrole/features/extract.py:65-142) together with FeaturePruner
 * (graphrole/features/prune.py:76-139) and the feature graph's components (graphrole/graph/graph.py:7-57):
 * generation 0 = the caller's neighbourhood-feature columns (grx_row_sums / grx_egonet_* outputs and
 * attribute columns; names as the reference spells them), then per generation: pack the retained columns,
 * aggregate over the neighbours (every aggregation of h_aggs, in that order: the candidate '<parent>(<agg>)'
 * columns are ordered aggregation-major like extract.py:158-162), bin the new columns, Chebyshev distances
 * over the working set with threshold = generation number, drop every member of a feature group but its
 * oldest, record what this generation retains (name-sorted iff something was dropped), stop when a
 * generation retains nothing or at max_generations.
 *
 * d_arena / arena_bytes: caller-provided device memory all columns, bins and scratch buffers are carved
 * from (no allocation inside); too small -> GRX_ERR_WORKSPACE with *arena_needed = bytes needed up to the
 * generation that failed (grow and call again; the run is deterministic).
 * h_columns (capacity max_columns): the RECORDED columns in record order -- generation 0's first;
 * `parent` indexes this table, names are rebuilt by the caller as name[parent] + '(' + agg + ')'.
 * h_gens (capacity max_gens >= max_generations): per-generation counts.  *generation_count = the last
 * executed generation (the reference's generation_count).  Synchronises `stream` before returning.
 *
 * h_gen0_int32 (may be NULL): per generation-0 column, non-zero if every value is an exact integer in [0, 2^31).
 * The loop then tracks what every later column IS -- an exact integer S (a neighbour sum of such a column, while
 * bits(S) + bits(longest row) <= 53), the mean fl(S / d) of one, or anything else -- learns the bit width of every
 * integer column's maximum with the read-back each generation makes anyway, and gathers generations whose parents are
 * all of the first two kinds from bit-packed rows (grx_aggregate_packed: 8 / 16 bytes per node; generations 1 and 2
 * of an unweighted graph) when only sums and means are asked for; the fall-back for integer parents whose widths are
 * not known is the int32 row of grx_aggregate_i32.  GRX_NO_PACKED_ROWS=1 (environment) switches the packed rows off.
 *
 * comm / h_bounds (NULL / NULL = one GPU): node-range sharding.  Every rank calls with the same arguments and the
 * COMPLETE generation-0 columns; it aggregates rows [h_bounds[rank], h_bounds[rank + 1]) only and the loop issues
 * its own exchanges on `stream`: candidate row slices to the column owners (grx_comm_columns_to_owners), the owners'
 * bins of the rank's rows back (grx_comm_owned_to_rows), all-reduce(MAX) of the F x F distance matrix (identical
 * pruning decisions everywhere), then the RETAINED new columns -- and only those -- completed on every rank
 * (grx_comm_all_gather_rows).  Every recorded column is complete on every rank on return; the neighbour sums
 * follow a tree that depends on the row length only, so the bits equal those of a one-GPU run.
 */
/*
 * Neighbour sums / means from BIT-PACKED integer rows.  The gather of grx_aggregate is bound by the chip's request
 * rate, and the share of requests that miss an XCD's L2 follows the BYTES of the gather table (DESIGN.md section 3,
 * profiles/r04_gather_bw.json).  On an unweighted graph every summand of generations 1 and 2
 * (graphrole/features/extract.py:104-119) is an exact integer S -- a generation-0 count or a neighbour sum of one --
 * or the mean fl(S / d) of one; a row that carries only the distinct S_k and the neighbour count d, each in the bits
 * its column maximum needs, is 8 or 16 bytes instead of 16 / 64, and the kernel rebuilds the summands in registers
 * (double(S), or double(S) / double(d): the correctly rounded division that produced the stored mean) and adds them
 * in numpy's pairwise order like grx_aggregate.  Results are bit-identical to grx_aggregate on the fp64 columns.
 *   grx_column_bits       d_bits[c] = max(d_bits[c], bits of max(column c, rows [row_begin, row_end))) for the columns
 *                         flagged in int_mask (exact non-negative integers); the caller zeroes d_bits first (the widths
 *                         of row slices combine by max, also across ranks); <= 64 columns per call
 *   grx_packed_row_bytes  8 / 16, or 0 when the fields do not fit two 64-bit words (no field straddles a word; every
 *                         field 1..62 bits, the neighbour count at most 31)
 *   grx_pack_fields       row u = fields S_k(u) = (int64) h_field_cols[k][u], then the neighbour count
 *                         d_row_ptr[u + 1] - d_row_ptr[u] when degree_bits > 0; d_rows 16-byte aligned
 *   grx_aggregate_packed  output j (column j of d_sum / d_mean, leading dimension ld) sums field out_field[j] of the
 *                         neighbours, as S or -- out_is_mean[j] -- as fl(S / d); rows of more than 128 neighbours go
 *                         through the plan's block list like grx_aggregate.  The caller guarantees
 *                         bits(S) + bits(longest row) <= 53 (sums stay exact integers).
 */
typedef struct {
    int n_fields;              /* 1 .. 7 integer source columns */
    int field_bits[8];
    int degree_bits;           /* width of the neighbour-count field, 0 = none (then no output may be a mean) */
    int n_out;                 /* 1 .. 8 outputs */
    int out_field[8];
    int out_is_mean[8];
} grx_packed_layout;
int grx_packed_row_bytes(const grx_packed_layout *layout);
int grx_column_bits(int64_t n, int ncols, const double *d_block, int64_t ld, int64_t row_begin, int64_t row_end,
                    uint64_t int_mask, int32_t *d_bits, void *stream);
int grx_pack_fields(int64_t n, const grx_packed_layout *layout, const double *const *h_field_cols, const int64_t *d_row_ptr,
                    void *d_rows, void *stream);
int grx_aggregate_packed(const grx_aggregate_plan *plan, const int64_t *d_row_ptr, const int32_t *d_col,
                         const grx_packed_layout *layout, const void *d_rows, int64_t row_begin, int64_t row_end,
                         double *d_sum, double *d_mean, int64_t ld, void *stream);

/* The pruning decision of the loop alone, on the host (no device work): FeaturePruner.prune_features given the
 * Chebyshev matrix (prune.py:76-130).  h_recorded_generation[j]: generation that recorded column j, -1 if none (a
 * new candidate); h_dist F x F; h_drop[j] = 1 for every member of a feature group but its oldest. */
int grx_host_prune(int F, const char *const *h_names, const int *h_recorded_generation, int n_generations,
                   const int32_t *h_dist, int thresh, int *h_drop);
typedef enum { GRX_AGG_SUM = 0, GRX_AGG_MEAN = 1, GRX_AGG_MIN = 2, GRX_AGG_MAX = 3, GRX_AGG_VAR = 4, GRX_AGG_STD = 5,
               GRX_AGG_PROD = 6, GRX_AGG_MEDIAN = 7, GRX_AGG_COUNT = 8, GRX_AGG_SIZE = 9 } grx_agg;
typedef struct {
    int generation;            /* generation that recorded the column */
    int parent;                /* index of the parent column in this table, -1 for generation 0 */
    int agg;                   /* grx_agg, -1 for generation 0 */
    int gen0_index;            /* index into h_gen0_cols for generation-0 columns, else -1 */
    int work_position;         /* position in the final working set (extract.py:135-141), -1 if pruned from it */
    const double *d_col;       /* fp64[n]: inside the arena, or the caller's generation-0 column */
} grx_refex_column;
typedef struct {
    int candidates, working, dropped, retained;
    int gather_row_bytes;      /* bytes per row of the generation's gather source (8 / 16: bit-packed integer rows,
                                  16 / 32: int32 rows, 8 * ldr: fp64 rows; 0 for generation 0) -- what a gather-rate
                                  ceiling has to be looked up with */
} grx_refex_generation;
/* grow (may be NULL): called when the arena is full -- returns `bytes` more bytes of device memory (256-byte aligned)
 * that stay valid as long as the caller uses the returned columns, or NULL.  With it the run never fails for want of
 * room (round 5: a first size is a guess, and a wrong guess used to cost a whole second run); columns may then lie
 * in any chunk -- d_col is an absolute pointer either way.  Without it a too-small arena is GRX_ERR_WORKSPACE and
 * *arena_needed a lower bound of what the run takes.  GRX_ERR_WORKSPACE is also what a full column table returns
 * (max_columns): a caller that grows tells the two apart by whether its grow ever returned NULL.
 * Sharded (comm != NULL): a grow that returns NULL on ONE rank (that device is out of memory) is not agreed with the
 * peers -- they wait in the next exchange until the transport's timeout ends the job. */
typedef void *(*grx_grow_fn)(size_t bytes, void *user);
/* Columns the loop hands to one call of the binning (workspace = grx_log_bin_workspace_bytes(n, this), at most about 4 GiB for
 * wide blocks of long columns; all of them for small graphs): what a caller's first arena size should assume. */
int grx_refex_bin_batch(int64_t n, int ncols);
int grx_refex_run(const grx_aggregate_plan *plan, int64_t n, const int64_t *d_row_ptr, const int32_t *d_agg_col,
                  int f0, const double *const *h_gen0_cols, const char *const *h_gen0_names, const int *h_gen0_int32,
                  int max_generations, int n_aggs, const int *h_aggs, grx_comm *comm, const int64_t *h_bounds, void *d_arena,
                  size_t arena_bytes, grx_grow_fn grow, void *grow_user, int max_columns, grx_refex_column *h_columns,
                  int *n_columns, int max_gens, grx_refex_generation *h_gens, int *generation_count, size_t *arena_needed,
                  void *stream);

/* ------------------------------------------------------------------ pruning ------------- */
/*
 * Vertical logarithmic binning of ncols columns.  Replaces vertical_log_binning
 * (graphrole/features/prune.py:13-56) as FeaturePruner._group_features applies it (:105).
 * d_cols: column j at d_cols + j*ld.  d_bins: uint8 bin ids, column j at d_bins + j*ld_bins.
 * d_nbins: int32[ncols] number of bins used per column (may be NULL).
 * Workspace: grx_log_bin_workspace_bytes(n, ncols) bytes.
 * 0 < frac < 1 (the reference raises ValueError otherwise -> GRX_ERR_INVALID).
 */
size_t grx_log_bin_workspace_bytes(int64_t n, int ncols);
int grx_vertical_log_bin(int64_t n, int ncols, const double *d_cols, int64_t ld, double frac,
                         uint8_t *d_bins, int64_t ld_bins, int32_t *d_nbins,
                         void *d_workspace, size_t workspace_bytes, void *stream);

/* The same with a per-column representation flag (host array, may be NULL): h_is_i64[j] != 0 -- column j holds int64
 * BITS (the reference's integer columns under aggs with 'prod', see grx_aggregate_i64) and is ordered as integers:
 * np.unique on an int64 array, prune.py:27.  At most the first 512 columns of a call may be flagged. */
int grx_vertical_log_bin_typed(int64_t n, int ncols, const double *d_cols, int64_t ld, const uint8_t *h_is_i64, double frac,
                               uint8_t *d_bins, int64_t ld_bins, int32_t *d_nbins, void *d_workspace,
                               size_t workspace_bytes, void *stream);

/* Batched ascending sort of fp64 columns (the first stage of the binning; exposed for tests).
 * Workspace: grx_sort_workspace_bytes(n, ncols). */
size_t grx_sort_workspace_bytes(int64_t n, int ncols);
int grx_sort_columns(int64_t n, int ncols, const double *d_cols, int64_t ld, double *d_sorted,
                     int64_t ld_sorted, void *d_workspace, size_t workspace_bytes, void *stream);

/*
 * Pairwise Chebyshev distance between binned columns.  Replaces
 * pdist(binned.T, metric='chebychev') (graphrole/features/prune.py:108).
 * h_bin_ptrs: HOST array of F device pointers to uint8 columns.  Only rows
 * [row_begin,row_end) are scanned (multi-GPU: all-reduce(MAX) the result).  d_dist: int32
 * F x F, must be zero-filled by the caller; pairs (p,q) with q >= first_new are computed
 * (first_new = 0: all pairs), the matrix is written symmetrically.  Any F: up to 96 columns are
 * one launch (one LDS tile), more are covered by one launch per pair of 48-column groups.
 * cap: the caller only needs distances up to `cap` exactly (the pruner compares with the generation
 * number, prune.py:110-113): entries <= cap are exact, larger ones are reported as SOME value > cap and
 * cost almost nothing (a pair leaves the work list once it exceeds the cap).  cap = 255: all exact.
 */
int grx_chebyshev(int64_t row_begin, int64_t row_end, int F, int first_new,
                  const uint8_t *const *h_bin_ptrs, int32_t *d_dist, int cap, void *stream);

/* ------------------------------------------------------------------ RolX NMF ------------ */
/*
 * HOST-side small dense algebra of the NNDSVDa initialisation (no device work; plain pointers to host
 * arrays, row-major).  Between its device passes (grx_gram twice, grx_project) sklearn's
 * initialisation (_nmf.py:324-359 via randomized_svd, extmath.py:531-604) only touches k x F matrices
 * with k <= F; these three calls replace ~25 numpy / LAPACK wrapper calls (0.6 ms per fit).  Any
 * F <= GRX_MAX_NMF_FEATURES: cyclic Jacobi up to 32 columns, tridiagonal QL above.
 *   grx_host_whiten        G1 = X^T X -> eigen-pairs above the numerical floor: lam_keep [k],
 *                          V_keep [F x k], T1 = V_keep / sqrt(lam_keep) [F x k]; *k = 0: X is zero
 *   grx_host_range_finder  G2 = (X T1)^T (X T1) -> T (X T orthonormal), M = (X T)^T X, randomized_svd of
 *                          M with the F x n_over test matrix omega and n_iter LU-normalised power
 *                          iterations -> Z [F x r] (U = X Z), S [r], Vt [r x F] (before svd_flip)
 *   grx_host_nndsvd_plan   per-column choices of NNDSVD from the grx_project statistics -> sign [r],
 *                          scale [r] for grx_nndsvd_apply and H [r x F] before thresholding
 */
 /* grx_host_eigh: the symmetric eigen-solver behind the two calls below (cyclic Jacobi up to 32 columns,
  * Householder tridiagonalisation + implicit QL above); h_A n x n row-major, h_w ascending, eigenvectors in
  * the columns of h_V.  Exposed for tests. */
int grx_host_eigh(int n, const double *h_A, double *h_w, double *h_V);
int grx_host_whiten(int F, const double *h_G1, double *h_T1, double *h_lam_keep, double *h_V_keep, int *k);
/* the same for a factorisation of rank r: when the plain decomposition keeps fewer than min(r, F) directions (a graded
 * table: column norms decades apart), the columns are equilibrated exactly (powers of two) before eigh */
int grx_host_whiten_for_rank(int F, const double *G1, int r, double *T1, double *lam_keep, double *V_keep, int *k_out);
int grx_host_range_finder(int F, int k, const double *h_T1, const double *h_lam_keep, const double *h_V_keep,
                          const double *h_G2, const double *h_omega, int n_over, int r, int n_iter,
                          double *h_Z, double *h_S, double *h_Vt);
int grx_host_nndsvd_plan(int r, int F, const double *h_S, const double *h_Vt, const double *h_stats,
                         double *h_sign, double *h_scale, double *h_H);
/* n < F (fewer nodes than features): sklearn's transposed randomized_svd branch (extmath.py:565-569) on the small
 * host matrix itself -- h_X n x F row-major, h_omega n x n_over; h_U n x r (before svd_flip), h_S [r], h_V r x F. */
int grx_host_small_svd(int n, int F, const double *h_X, const double *h_omega, int n_over, int r, int n_iter,
                       double *h_U, double *h_S, double *h_V);

/*
 * All NMF matrices are "feature-major": X is F x ldx (row c = feature column c, n valid
 * entries), W is r x ldw.  H is r x F row-major, HHt is r x r.  Rows [row_begin,row_end) of
 * the node axis are processed (multi-GPU: partial results are summed by the caller).
 */

/* Copy F columns given by a HOST array of device pointers into a contiguous F x ld matrix (F <= 128). */
int grx_gather_columns(int64_t n, int F, const double *const *h_col_ptrs, double *d_out,
                       int64_t ld, void *stream);

/*
 * G = (X T)^T (X T), T = h_T (host, F x k row-major; NULL = identity, k = F), k <= 128.
 * Also returns sum of all entries of X in d_out[k*k] (for X.mean()).  d_out: fp64 [k*k + 1].
 * Used for the rank-revealing orthogonal factorisation behind the NNDSVDa initialisation that
 * sklearn's NMF(init='nndsvda') performs (sklearn/decomposition/_nmf.py:316-359,
 * sklearn/utils/extmath.py:531-604) at graphrole/roles/factor.py:19.
 */
size_t grx_gram_workspace_bytes(int64_t n, int k);
int grx_gram(int64_t n, int F, const double *d_X, int64_t ldx, int64_t row_begin, int64_t row_end,
             const double *h_T, int k, double *d_out, void *d_workspace, size_t workspace_bytes,
             void *stream);

/*
 * U = X Z (Z = h_Z host, F x r row-major) written feature-major to d_U (r x ldu), plus per
 * column j statistics in d_stats (fp64 [r*4]): {max|U_j| entry with its sign, index of it,
 * sum of squares of positive part, sum of squares of negative part}.
 */
size_t grx_project_workspace_bytes(int64_t n, int r);
int grx_project(int64_t n, int F, const double *d_X, int64_t ldx, int64_t row_begin,
                int64_t row_end, const double *h_Z, int r, double *d_U, int64_t ldu,
                double *d_stats, void *d_workspace, size_t workspace_bytes, void *stream);

/*
 * NNDSVDa column transform (sklearn/decomposition/_nmf.py:324-359), in place on d_U:
 *   W[j][i] = scale[j] * max(sign[j] * U[j][i], 0)   (column 0: scale*|U|),
 *   values < eps -> 0 -> fill.   h_sign: +1/-1 per column (0 = take |.|).
 */
int grx_nndsvd_apply(int64_t n, int r, double *d_U, int64_t ldu, int64_t row_begin,
                     int64_t row_end, const double *h_sign, const double *h_scale, double eps,
                     double fill, void *stream);

/*
 * Multiplicative-update state.  Replaces the loop body of sklearn's
 * _fit_multiplicative_update (sklearn/decomposition/_nmf.py:831-868) reached from
 * graphrole/roles/factor.py:24:
 *     W <- W * (X H^T) / (W (H H^T));   H <- H * (W^T X) / ((W^T W) H);   zero denominators
 *     are replaced by float32 eps (_nmf.py:39,632,720).
 * grx_nmf_w_pass   : updates W rows [row_begin,row_end) in place and writes this rank's
 *                    partial sums  d_AB = [ W'^T X (r x F) | W'^T W' (r x r) ].
 * grx_nmf_h_update : H <- H * A / (B H) from the (all-reduced) d_AB.
 * grx_nmf_residual : d_out[0] = sum_i ||x_i - w_i H||^2 over the row range
 *                    (_beta_divergence, _nmf.py:120-133, before the square root).
 */
size_t grx_nmf_workspace_bytes(int64_t n, int F, int r);
/* grx_nmf_w_pass_next: the W pass of the NEXT iteration with the H update of the previous one folded in --
 * H = d_H_prev * A / (B d_H_prev) from the (all-reduced) d_AB_prev is computed in the kernel's prologue, used for the
 * pass and stored in d_H_out (a different buffer); equal to grx_nmf_h_update followed by grx_nmf_w_pass, one launch
 * fewer per iteration.  d_AB_prev may be d_AB (it is read before the partial sums are reduced into d_AB). */
int grx_nmf_w_pass_next(int64_t n, int F, int r, const double *d_X, int64_t ldx, double *d_W, int64_t ldw,
                        int64_t row_begin, int64_t row_end, const double *d_H_prev, const double *d_AB_prev,
                        double *d_H_out, double *d_AB, void *d_workspace, size_t workspace_bytes, void *stream);
int grx_nmf_w_pass(int64_t n, int F, int r, const double *d_X, int64_t ldx, double *d_W,
                   int64_t ldw, int64_t row_begin, int64_t row_end, const double *d_H,
                   double *d_AB, void *d_workspace, size_t workspace_bytes, void *stream);
int grx_nmf_h_update(int F, int r, double *d_H, const double *d_AB, void *stream);
int grx_nmf_residual(int64_t n, int F, int r, const double *d_X, int64_t ldx, const double *d_W,
                     int64_t ldw, int64_t row_begin, int64_t row_end, const double *d_H,
                     double *d_out, void *d_workspace, size_t workspace_bytes, void *stream);
/*
 * MDL error cost of an (encoded) factor pair: generalised KL divergence of X from W H over the
 * non-zero entries of X (graphrole/roles/description_length.py:44-61), rows [row_begin,row_end).
 * d_out[0] receives the sum.  Workspace as for the other NMF passes.
 */
int grx_nmf_kl_cost(int64_t n, int F, int r, const double *d_X, int64_t ldx, const double *d_W,
                    int64_t ldw, int64_t row_begin, int64_t row_end, const double *d_H, double *d_out,
                    void *d_workspace, size_t workspace_bytes, void *stream);
/*
 * Enqueue `iters` full single-GPU iterations followed (d_err != NULL) by one residual evaluation into
 * d_err[0]; no host synchronisation.  Same results as `iters` x (grx_nmf_w_pass, grx_nmf_h_update), two
 * launches per iteration instead of three: for F <= 120 the H update of an iteration runs in the prologue of the
 * next W pass (two scratch copies of H at the end of the workspace), d_H holds the final H on completion.
 */
int grx_nmf_iterate(int64_t n, int F, int r, const double *d_X, int64_t ldx, double *d_W,
                    int64_t ldw, double *d_H, double *d_AB, double *d_err, int iters,
                    void *d_workspace, size_t workspace_bytes, void *stream);
/* The same block of iterations on a row shard: W rows [row_begin,row_end) are updated, the partial sums of every
 * pass are summed over the ranks of `comm` (NULL: no exchange) before the H update that consumes them -- one
 * small all-reduce per iteration, enqueued between the launches.  H ends identical on every rank. */
int grx_nmf_iterate_rows(int64_t n, int F, int r, const double *d_X, int64_t ldx, double *d_W, int64_t ldw,
                         int64_t row_begin, int64_t row_end, double *d_H, double *d_AB, int iters, grx_comm *comm,
                         void *d_workspace, size_t workspace_bytes, void *stream);

/*
 * The whole factorisation below the ABI.  Replaces get_nmf_decomposition (graphrole/roles/factor.py:10-26),
 * i.e. sklearn NMF(n_components=r, solver='mu', init='nndsvda').fit_transform(X) with its defaults
 * (tol 1e-4, max_iter 200; sklearn/decomposition/_nmf.py:1538-1553), on a feature-major device matrix:
 *   grx_nmf_init  NNDSVDa start (_nmf.py:316-359 via randomized_svd, extmath.py:531-604): two Gram
 *                 passes, the k x F host algebra (grx_host_*), one projection pass, the element-wise
 *                 transform.  h_omega: the F x n_over Gaussian test matrix the caller drew exactly like
 *                 sklearn does (global numpy RNG, n_over = r + 10).  Needs n >= F >= r.  Writes W0 to d_W
 *                 (r x ldw), H0 to d_H (r x F, device) and ||X||_F^2 to *x_sq_norm.
 *                 GRX_ERR_DEGENERATE: X is numerically zero.
 *   grx_nmf_mu    multiplicative updates from the W, H in place with sklearn's stopping rule
 *                 (_nmf.py:815-885: error every 10 iterations, stop when (prev - err) / err_init < tol).
 *                 x_sq_norm > 0 lets the convergence checks use ||X||^2 - 2<W^T X, H> + <W^T W, H H^T>
 *                 from the W-pass outputs whenever the relative squared residual exceeds 1e-8 (direct pass
 *                 over X otherwise; x_sq_norm <= 0: always direct).
 *   grx_nmf_fit   both.
 * Kernels are enqueued on `stream`; the calls synchronise it at every small read-back (three in the
 * initialisation, one per ten iterations) and return with the results complete in d_W / d_H.
 * comm / h_bounds (NULL / NULL = one GPU): row-sharded fit.  Every rank holds the same X and passes the same
 * omega; the O(n) passes cover rows [h_bounds[rank], h_bounds[rank + 1]) and the drivers issue the exchanges:
 * all-reduce(SUM) of the two Gram matrices, the projection statistics of every rank merged on the host, one
 * all-reduce of [W^T X | W^T W] per iteration (grx_nmf_iterate_rows), of the residual at direct convergence
 * checks; grx_nmf_mu / grx_nmf_fit finish by completing W on every rank (grx_comm_all_gather_rows).  H and n_iter
 * are identical on every rank.
 */
typedef struct {
    int n_iter;               /* executed iterations: sklearn's n_iter_ */
    int direct_residuals;     /* convergence checks that needed the direct ||X - W H|| pass */
    double err_init;          /* ||X - W0 H0||_F */
    double err_last;          /* error at the last convergence check */
    double x_sq_norm;         /* ||X||_F^2 as used by the checks */
} grx_nmf_info;
size_t grx_nmf_fit_workspace_bytes(int64_t n, int F, int r);
int grx_nmf_init(int64_t n, int F, int r, const double *d_X, int64_t ldx, const double *h_omega, int n_over,
                 double *d_W, int64_t ldw, double *d_H, double *x_sq_norm, grx_comm *comm, const int64_t *h_bounds,
                 void *d_workspace, size_t workspace_bytes, void *stream);
int grx_nmf_mu(int64_t n, int F, int r, const double *d_X, int64_t ldx, double *d_W, int64_t ldw, double *d_H,
               double x_sq_norm, double tol, int max_iter, grx_nmf_info *info, grx_comm *comm, const int64_t *h_bounds,
               void *d_workspace, size_t workspace_bytes, void *stream);
int grx_nmf_fit(int64_t n, int F, int r, const double *d_X, int64_t ldx, const double *h_omega, int n_over,
                double tol, int max_iter, double *d_W, int64_t ldw, double *d_H, grx_nmf_info *info, grx_comm *comm,
                const int64_t *h_bounds, void *d_workspace, size_t workspace_bytes, void *stream);

/* ------------------------------------------------------------------ RolX encode --------- */
/*
 * 1-D Lloyd-Max quantiser.  Replaces encode() (graphrole/roles/factor.py:29-49: sklearn KMeans on
 * the flattened factor entries, every entry replaced by its cluster centre).  Deterministic:
 * sort -> prefix sums -> exact DP over <= 1024 equal cbrt-density micro-cells -> Lloyd refinement
 * -> assignment.  d_values / d_quantized: fp64[m]; d_centers: fp64[n_bins] (ascending);
 * d_info: int32[3] = {Lloyd iterations, non-empty cells, distinct output values}.  Up to 256 levels the start
 * is the exact DP; 257..65536 levels (9..16 bits: roles/extract.py:72 asks for 2**int(log2(n_roles * min(shape)))
 * on wide tables) start from the companding partition itself (equal cbrt-density cells).
 * n_bins > m is the reference's ValueError (sklearn: "n_samples=.. should be >= n_clusters=..")
 * -> GRX_ERR_INVALID.
 */
size_t grx_lloyd_max_workspace_bytes(int64_t m);
int grx_lloyd_max(int64_t m, const double *d_values, int n_bins, int max_iter, double *d_quantized,
                  double *d_centers, int32_t *d_info, void *d_workspace, size_t workspace_bytes,
                  void *stream);

/*
 * The reference's quantiser itself.  Replaces encode() (graphrole/roles/factor.py:29-49) with the SAME procedure
 * sklearn runs there: KMeans(n_clusters=k, random_state=1) on the flattened entries -- k-means++ seeding driven by
 * the caller's random numbers (drawn exactly as RandomState(1) hands them to sklearn: they do not depend on the
 * data), Lloyd iterations with sklearn's stopping rule (labels unchanged, or total squared centre shift <=
 * 1e-4 * var), empty-cluster relocation, every entry replaced by its cluster centre.  Centres agree with sklearn's
 * to ~1e-12 (sums are reduced in another order); the model selection of RolX then picks the reference's cell.
 * The seeding does not stream all m values per seed as sklearn's _kmeans_plusplus (:163-257) does: on the line a
 * candidate changes closest distances only in a contiguous range of the sorted values, and the sums the candidate
 * INDEX is drawn from (sklearn's cumulative sum in index order) are exact fixed-point integers (csrc/grx_kmeans.hip)
 * -- same seeds, O(m log k) instead of O(m k) work.  The candidates' potentials, which only feed an argmin, come from
 * fp64 sums over blocks of sorted values (first minimum, potentials within 1e-12 count as tied: sklearn's own are BLAS
 * sums in another order).
 *   d_values      fp64[m] in the order the reference flattens the matrix (row-major n x r for the node-role
 *                 factor: grx_transpose from the feature-major device layout)
 *   first_seed    RandomState(1).choice(m, p = uniform)            (sklearn _kmeans_plusplus, first centre)
 *   h_uniform     (k - 1) x n_trials doubles, n_trials = 2 + int(log(k)): RandomState.uniform(size=n_trials) per seed
 *   max_iter, rel_tol   sklearn defaults 300, 1e-4
 *   d_centers     fp64[k] in seed order;  d_info int32[4] = {Lloyd iterations (n_iter_), non-empty clusters,
 *                 distinct output values, seeding faults (0 unless an internal consistency check failed: bit 0 the
 *                 block sums after an update differ from potential - gain (exact gains, GRX_KMEANS_GAIN_PASS=1), bit 1
 *                 values beyond 1e144, bit 2 a cumulative-sum search left its block, bit 3 the potential after an
 *                 update is further than 1e-9 from potential - gain (fp64 gains, the default), bit 4 a workgroup of
 *                 the pick waited in vain for the others of its launch)}.
 *                 k <= 8192 (GRX_ERR_UNSUPPORTED above), k <= m (GRX_ERR_INVALID).
 * Environment (read once; for tests/test_gpu_encode.py, which runs every mode on the same inputs and compares bits):
 * GRX_KMEANS_FULL_RANGE=1 every range [0, m); GRX_KMEANS_GAIN_PASS=1 gains by an integer pass over each range;
 * GRX_KMEANS_SLOW_PICK=1 range positions by searches over all values; GRX_KMEANS_MERGE=0 the update always in its own
 * kernel; GRX_KMEANS_SMALL=0 no one-workgroup seeding of m <= 4096; GRX_KMEANS_LOSE_A_SUM=<seed> (test of the bounded
 * waits) one workgroup of that seed's launch withholds its sum: the run ends within a second with fault bit 4.
 * grx_transpose: out[c * ld_out + r] = in[r * ld_in + c].
 */
size_t grx_kmeans1d_workspace_bytes(int64_t m, int k);
int grx_kmeans1d(int64_t m, const double *d_values, int k, int64_t first_seed, const double *h_uniform, int n_trials,
                 int max_iter, double rel_tol, double *d_quantized, double *d_centers, int32_t *d_info,
                 void *d_workspace, size_t workspace_bytes, void *stream);
int grx_transpose(int64_t rows, int64_t cols, const double *d_in, int64_t ld_in, double *d_out, int64_t ld_out,
                  void *stream);


/* ------------------------------------------------------------------ RolX roles ---------- */
/*
 * The two row passes over the fitted node-role factor.  d_G: fp64 n x r row-major (the layout of the reference's
 * node_role_factor.values), any r >= 1 (the limit of the NMF kernels, GRX_MAX_ROLES, does not apply: a caller may
 * assign a wider frame).
 *   grx_role_argmax    replaces RoleExtractor.roles (graphrole/roles/extract.py:38-47, DataFrame.idxmax(axis=1)):
 *                      d_first_max int32[n] = column of the FIRST maximum of every row (quantised factors are full
 *                      of exact ties), NaN entries skipped, -1 for a row of NaNs only.
 *   grx_row_normalise  replaces RoleExtractor.role_percentage (:49-57, apply(lambda row: row / row.sum(), axis=1)):
 *                      d_share fp64 n x r = every row divided by its sum, the sum evaluated in the order
 *                      Series.sum() adds r doubles (numpy pairwise_sum: left to right below 8 values, eight strided
 *                      accumulators from 8 on; NaN counts as 0) -- bit-equal to the reference; 0 / 0 = NaN as there.
 */
int grx_role_argmax(int64_t n, int r, const double *d_G, int32_t *d_first_max, void *stream);
int grx_row_normalise(int64_t n, int r, const double *d_G, double *d_share, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* GRX_H */
