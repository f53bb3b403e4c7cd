// grx_comm.hip -- the transport between the ranks of a node-range sharded run (host code; no kernels).
//
// The reference is single-process: everything here is new.  One process per GPU, CSR and
// feature columns replicated, rank p computes node rows [bounds[p], bounds[p+1]) -- see include/grx.h, "node-range
// sharding".  Two transports behind one handle:
//   * RCCL over xGMI, bound at run time with dlopen/dlsym (librccl.so.1 of the process -- PyTorch ships one -- or
//     the system copy): all-reduce for the small reductions, grouped ncclSend / ncclRecv for everything that moves
//     row slices.  xGMI is point-to-point (7 links per GPU): a group of direct sends keeps every link busy with
//     exactly the bytes its peer needs, no ring, no staging.
//   * caller-provided callbacks (tests: gloo staged through the host; any MPI-like library).
// The composites never pack: a column's row slice is contiguous (column-major features), so each transfer names
// the slice where it lies -- in the candidate block, the owner's whole-column buffer or the bin block.
#include "grx_common.h"

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>

namespace {

struct RcclApi {
    void *handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
};

RcclApi g_rccl;
std::mutex g_rccl_mu;

int load_rccl()
{
    std::lock_guard<std::mutex> lk(g_rccl_mu);
    if (g_rccl.handle) return GRX_OK;
    void *h = nullptr;
    const char *env = std::getenv("GRX_RCCL_PATH");
    if (env && *env) h = dlopen(env, RTLD_NOW | RTLD_GLOBAL);
    // the copy this process already maps (one RCCL per process: PyTorch's, when the host is PyTorch)
    if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_NOLOAD);
    if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) {
        grx_set_error("grx_comm: librccl.so.1 not found (%s); set GRX_RCCL_PATH", dlerror());
        return GRX_ERR_UNSUPPORTED;
    }
    RcclApi api;
    api.handle = h;
#define GRX_SYM(field, name)                                                                  \
    api.field = reinterpret_cast<decltype(api.field)>(dlsym(h, name));                        \
    if (!api.field) { grx_set_error("grx_comm: librccl has no symbol %s", name); return GRX_ERR_UNSUPPORTED; }
    GRX_SYM(GetUniqueId, "ncclGetUniqueId")
    GRX_SYM(CommInitRank, "ncclCommInitRank")
    GRX_SYM(CommDestroy, "ncclCommDestroy")
    GRX_SYM(AllReduce, "ncclAllReduce")
    GRX_SYM(Send, "ncclSend")
    GRX_SYM(Recv, "ncclRecv")
    GRX_SYM(GroupStart, "ncclGroupStart")
    GRX_SYM(GroupEnd, "ncclGroupEnd")
    GRX_SYM(GetErrorString, "ncclGetErrorString")
#undef GRX_SYM
    g_rccl = api;
    return GRX_OK;
}

#define GRX_CHECK_RCCL(expr)                                                                   \
    do {                                                                                       \
        ncclResult_t r__ = (expr);                                                             \
        if (r__ != ncclSuccess) {                                                              \
            grx_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, g_rccl.GetErrorString(r__)); \
            return GRX_ERR_HIP;                                                                \
        }                                                                                      \
    } while (0)

struct TimedCall { int kind; hipEvent_t start, stop; };

}  // namespace

struct grx_comm {
    int rank = 0, world = 1;
    int flags = 0;
    ncclComm_t rccl = nullptr;                       // RCCL transport
    grx_all_reduce_fn cb_all_reduce = nullptr;       // callback transport
    grx_exchange_fn cb_exchange = nullptr;
    void *user = nullptr;
    bool timing = false;
    int depth = 0;                                   // a composite times itself, not the primitives inside it
    std::vector<TimedCall> pending;
    long long calls[GRX_COMM_KINDS] = {0, 0, 0, 0, 0};
    double ms[GRX_COMM_KINDS] = {0, 0, 0, 0, 0};
};

namespace {

struct TimeScope {
    grx_comm *c; int kind; hipStream_t st; hipEvent_t start = nullptr; bool on;
    TimeScope(grx_comm *comm, int k, hipStream_t s) : c(comm), kind(k), st(s)
    {
        on = c->timing && c->depth == 0;
        c->depth += 1;
        if (on && hipEventCreate(&start) == hipSuccess) (void)hipEventRecord(start, st); else on = false;
    }
    ~TimeScope()
    {
        c->depth -= 1;
        if (!on) return;
        hipEvent_t stop = nullptr;
        if (hipEventCreate(&stop) != hipSuccess) { (void)hipEventDestroy(start); return; }
        (void)hipEventRecord(stop, st);
        c->pending.push_back({kind, start, stop});
    }
};


ncclDataType_t rccl_dtype(int dtype)
{
    switch (dtype) {
    case GRX_F64: return ncclFloat64;
    case GRX_I32: return ncclInt32;
    case GRX_I64: return ncclInt64;
    default: return ncclUint8;
    }
}

int exchange_impl(grx_comm *c, int n_ops, const grx_p2p_op *ops, hipStream_t st)
{
    // transfers of a rank to itself: device-to-device copies, matched in order (unless the caller wants even
    // those to exercise the transport)
    const bool self_local = !(c->rccl && (c->flags & GRX_COMM_SELF_VIA_TRANSPORT));
    std::vector<grx_p2p_op> remote;
    remote.reserve(n_ops);
    std::vector<const grx_p2p_op *> self_send, self_recv;
    for (int i = 0; i < n_ops; ++i) {
        const grx_p2p_op &op = ops[i];
        GRX_REQUIRE(op.peer >= 0 && op.peer < c->world, "grx_comm_exchange: peer %d outside the group of %d", op.peer,
                    c->world);
        if (op.bytes == 0) continue;
        GRX_REQUIRE(op.d_ptr != nullptr, "grx_comm_exchange: NULL buffer in op %d", i);
        if (op.peer == c->rank && self_local) (op.is_recv ? self_recv : self_send).push_back(&op);
        else remote.push_back(op);
    }
    GRX_REQUIRE(self_send.size() == self_recv.size(), "grx_comm_exchange: %zu sends to self but %zu receives",
                self_send.size(), self_recv.size());
    for (size_t i = 0; i < self_send.size(); ++i) {
        GRX_REQUIRE(self_send[i]->bytes == self_recv[i]->bytes, "grx_comm_exchange: self transfer %zu: %zu bytes sent, %zu expected",
                    i, self_send[i]->bytes, self_recv[i]->bytes);
        if (self_send[i]->d_ptr != self_recv[i]->d_ptr)
            GRX_CHECK_HIP(hipMemcpyAsync(self_recv[i]->d_ptr, self_send[i]->d_ptr, self_send[i]->bytes,
                                         hipMemcpyDeviceToDevice, st));
    }
    // RCCL sends / receives are point to point: nothing to do without remote transfers.  A callback transport may be
    // built on a collective (the gloo one is an all_to_all_single): it is entered by every rank of every exchange,
    // also by a rank that has nothing to move (no rows and no owned column), or its peers would wait for it forever.
    if (remote.empty() && (c->rccl || c->world == 1)) return GRX_OK;
    if (c->rccl) {
        GRX_CHECK_RCCL(g_rccl.GroupStart());
        for (const grx_p2p_op &op : remote) {
            ncclResult_t r = op.is_recv ? g_rccl.Recv(op.d_ptr, op.bytes, ncclUint8, op.peer, c->rccl, st)
                                        : g_rccl.Send(op.d_ptr, op.bytes, ncclUint8, op.peer, c->rccl, st);
            if (r != ncclSuccess) {
                (void)g_rccl.GroupEnd();
                grx_set_error("grx_comm_exchange: ncclSend/ncclRecv -> %s", g_rccl.GetErrorString(r));
                return GRX_ERR_HIP;
            }
        }
        GRX_CHECK_RCCL(g_rccl.GroupEnd());
        return GRX_OK;
    }
    GRX_REQUIRE(c->cb_exchange != nullptr, "grx_comm_exchange: the communicator has no transport");
    const int rc = c->cb_exchange(c->user, (int)remote.size(), remote.data(), st);
    if (rc != GRX_OK) { grx_set_error("grx_comm_exchange: the transport callback returned %d", rc); return rc < 0 ? rc : GRX_ERR_HIP; }
    return GRX_OK;
}

// columns owned by rank q of ncols: q, q + world, ...
inline int owned_count(int ncols, int q, int world) { return ncols > q ? (ncols - q + world - 1) / world : 0; }

}  // namespace

extern "C" {

int grx_comm_rccl_unique_id(void *h_id)
{
    GRX_REQUIRE(h_id != nullptr, "grx_comm_rccl_unique_id: NULL");
    int rc = load_rccl();
    if (rc != GRX_OK) return rc;
    static_assert(sizeof(ncclUniqueId) == GRX_COMM_ID_BYTES, "GRX_COMM_ID_BYTES must equal sizeof(ncclUniqueId)");
    ncclUniqueId id;
    GRX_CHECK_RCCL(g_rccl.GetUniqueId(&id));
    std::memcpy(h_id, &id, sizeof(id));
    return GRX_OK;
}

int grx_comm_create_rccl(const void *h_id, int rank, int world, int flags, grx_comm **out)
{
    GRX_REQUIRE(h_id && out && world >= 1 && rank >= 0 && rank < world, "grx_comm_create_rccl: bad arguments");
    int rc = load_rccl();
    if (rc != GRX_OK) return rc;
    ncclUniqueId id;
    std::memcpy(&id, h_id, sizeof(id));
    grx_comm *c = new grx_comm;
    c->rank = rank;
    c->world = world;
    c->flags = flags;
    ncclResult_t r = g_rccl.CommInitRank(&c->rccl, world, id, rank);
    if (r != ncclSuccess) {
        grx_set_error("grx_comm_create_rccl: ncclCommInitRank(rank %d of %d) -> %s", rank, world, g_rccl.GetErrorString(r));
        delete c;
        return GRX_ERR_HIP;
    }
    *out = c;
    return GRX_OK;
}

int grx_comm_create_callbacks(int rank, int world, grx_all_reduce_fn all_reduce, grx_exchange_fn exchange, void *user,
                              grx_comm **out)
{
    GRX_REQUIRE(out && world >= 1 && rank >= 0 && rank < world, "grx_comm_create_callbacks: bad arguments");
    GRX_REQUIRE(world == 1 || (all_reduce && exchange), "grx_comm_create_callbacks: a group of %d needs both callbacks", world);
    grx_comm *c = new grx_comm;
    c->rank = rank;
    c->world = world;
    c->cb_all_reduce = all_reduce;
    c->cb_exchange = exchange;
    c->user = user;
    *out = c;
    return GRX_OK;
}

int grx_comm_destroy(grx_comm *comm)
{
    if (!comm) return GRX_OK;
    for (auto &t : comm->pending) { (void)hipEventDestroy(t.start); (void)hipEventDestroy(t.stop); }
    if (comm->rccl) (void)g_rccl.CommDestroy(comm->rccl);
    delete comm;
    return GRX_OK;
}

int grx_comm_rank(const grx_comm *comm) { return comm ? comm->rank : 0; }
int grx_comm_world(const grx_comm *comm) { return comm ? comm->world : 1; }

int grx_comm_all_reduce(grx_comm *comm, void *d_buf, size_t count, int dtype, int op, void *stream)
{
    GRX_REQUIRE(comm != nullptr, "grx_comm_all_reduce: NULL communicator");
    GRX_REQUIRE(dtype >= 0 && dtype <= GRX_U8 && (op == GRX_SUM || op == GRX_MAX), "grx_comm_all_reduce: bad dtype / op");
    if (count == 0) return GRX_OK;
    GRX_REQUIRE(d_buf != nullptr, "grx_comm_all_reduce: NULL buffer");
    hipStream_t st = grx_stream(stream);
    TimeScope scope(comm, GRX_COMM_ALL_REDUCE, st);
    if (comm->rccl) {
        if (comm->world == 1 && !(comm->flags & GRX_COMM_SELF_VIA_TRANSPORT)) return GRX_OK;
        GRX_CHECK_RCCL(g_rccl.AllReduce(d_buf, d_buf, count, rccl_dtype(dtype), op == GRX_SUM ? ncclSum : ncclMax, comm->rccl, st));
        return GRX_OK;
    }
    if (comm->world == 1) return GRX_OK;
    GRX_REQUIRE(comm->cb_all_reduce != nullptr, "grx_comm_all_reduce: the communicator has no transport");
    const int rc = comm->cb_all_reduce(comm->user, d_buf, count, dtype, op, stream);
    if (rc != GRX_OK) { grx_set_error("grx_comm_all_reduce: the transport callback returned %d", rc); return rc < 0 ? rc : GRX_ERR_HIP; }
    return GRX_OK;
}

int grx_comm_exchange(grx_comm *comm, int n_ops, const grx_p2p_op *ops, void *stream)
{
    GRX_REQUIRE(comm != nullptr && n_ops >= 0 && (ops || n_ops == 0), "grx_comm_exchange: bad arguments");
    hipStream_t st = grx_stream(stream);
    TimeScope scope(comm, GRX_COMM_EXCHANGE, st);
    return exchange_impl(comm, n_ops, ops, st);
}

int grx_comm_all_gather_rows(grx_comm *comm, const int64_t *h_bounds, int ncols, void *const *h_col_ptrs,
                             int elem_bytes, void *stream)
{
    GRX_REQUIRE(comm && h_bounds && ncols >= 0 && (h_col_ptrs || ncols == 0) && elem_bytes >= 1,
                "grx_comm_all_gather_rows: bad arguments");
    hipStream_t st = grx_stream(stream);
    TimeScope scope(comm, GRX_COMM_ALL_GATHER_ROWS, st);
    const int P = comm->world, me = comm->rank;
    const size_t mine = (size_t)(h_bounds[me + 1] - h_bounds[me]) * elem_bytes;
    std::vector<grx_p2p_op> ops;
    ops.reserve((size_t)2 * ncols * P);
    const bool self_too = comm->rccl && (comm->flags & GRX_COMM_SELF_VIA_TRANSPORT);
    for (int q = 0; q < P; ++q) {
        if (q == me && !self_too) continue;                   // the own slice is already in place
        const size_t theirs = (size_t)(h_bounds[q + 1] - h_bounds[q]) * elem_bytes;
        for (int j = 0; j < ncols; ++j) {
            char *col = reinterpret_cast<char *>(h_col_ptrs[j]);
            GRX_REQUIRE(col != nullptr, "grx_comm_all_gather_rows: column %d is NULL", j);
            if (q == me) {
                // in place: a send and a receive of the same bytes (RCCL handles the aliasing as a no-op copy)
                ops.push_back({0, q, col + (size_t)h_bounds[me] * elem_bytes, mine});
                ops.push_back({1, q, col + (size_t)h_bounds[me] * elem_bytes, mine});
                continue;
            }
            ops.push_back({0, q, col + (size_t)h_bounds[me] * elem_bytes, mine});
            ops.push_back({1, q, col + (size_t)h_bounds[q] * elem_bytes, theirs});
        }
    }
    return exchange_impl(comm, (int)ops.size(), ops.data(), st);
}

int grx_comm_columns_to_owners(grx_comm *comm, const int64_t *h_bounds, int ncols, const void *d_block, int64_t ld,
                               int elem_bytes, void *d_owned, int64_t ld_owned, void *stream)
{
    GRX_REQUIRE(comm && h_bounds && ncols >= 0 && elem_bytes >= 1, "grx_comm_columns_to_owners: bad arguments");
    hipStream_t st = grx_stream(stream);
    TimeScope scope(comm, GRX_COMM_COLUMNS_TO_OWNERS, st);
    const int P = comm->world, me = comm->rank;
    const int n_owned = owned_count(ncols, me, P);
    GRX_REQUIRE((d_block || ncols == 0) && (d_owned || n_owned == 0), "grx_comm_columns_to_owners: NULL buffer");
    const char *block = reinterpret_cast<const char *>(d_block);
    char *owned = reinterpret_cast<char *>(d_owned);
    const size_t mine = (size_t)(h_bounds[me + 1] - h_bounds[me]) * elem_bytes;
    std::vector<grx_p2p_op> ops;
    ops.reserve((size_t)ncols + (size_t)n_owned * P);
    // pair (me -> q): my rows of q's columns, ascending column; pair (q -> me): q's rows of my columns, ascending
    for (int c = 0; c < ncols; ++c)
        ops.push_back({0, c % P, const_cast<char *>(block) + ((size_t)c * ld + h_bounds[me]) * elem_bytes, mine});
    for (int q = 0; q < P; ++q) {
        const size_t theirs = (size_t)(h_bounds[q + 1] - h_bounds[q]) * elem_bytes;
        for (int j = 0; j < n_owned; ++j)
            ops.push_back({1, q, owned + ((size_t)j * ld_owned + h_bounds[q]) * elem_bytes, theirs});
    }
    return exchange_impl(comm, (int)ops.size(), ops.data(), st);
}

int grx_comm_owned_to_rows(grx_comm *comm, const int64_t *h_bounds, int ncols, const void *d_owned, int64_t ld_owned,
                           int elem_bytes, void *d_block, int64_t ld, void *stream)
{
    GRX_REQUIRE(comm && h_bounds && ncols >= 0 && elem_bytes >= 1, "grx_comm_owned_to_rows: bad arguments");
    hipStream_t st = grx_stream(stream);
    TimeScope scope(comm, GRX_COMM_OWNED_TO_ROWS, st);
    const int P = comm->world, me = comm->rank;
    const int n_owned = owned_count(ncols, me, P);
    GRX_REQUIRE((d_block || ncols == 0) && (d_owned || n_owned == 0), "grx_comm_owned_to_rows: NULL buffer");
    const char *owned = reinterpret_cast<const char *>(d_owned);
    char *block = reinterpret_cast<char *>(d_block);
    const size_t mine = (size_t)(h_bounds[me + 1] - h_bounds[me]) * elem_bytes;
    std::vector<grx_p2p_op> ops;
    ops.reserve((size_t)ncols + (size_t)n_owned * P);
    for (int q = 0; q < P; ++q) {
        const size_t theirs = (size_t)(h_bounds[q + 1] - h_bounds[q]) * elem_bytes;
        for (int j = 0; j < n_owned; ++j)
            ops.push_back({0, q, const_cast<char *>(owned) + ((size_t)j * ld_owned + h_bounds[q]) * elem_bytes, theirs});
    }
    for (int c = 0; c < ncols; ++c)
        ops.push_back({1, c % P, block + ((size_t)c * ld + h_bounds[me]) * elem_bytes, mine});
    return exchange_impl(comm, (int)ops.size(), ops.data(), st);
}

int grx_comm_timing(grx_comm *comm, int on)
{
    GRX_REQUIRE(comm != nullptr, "grx_comm_timing: NULL communicator");
    comm->timing = on != 0;
    return GRX_OK;
}

int grx_comm_timing_reset(grx_comm *comm)
{
    GRX_REQUIRE(comm != nullptr, "grx_comm_timing_reset: NULL communicator");
    for (auto &t : comm->pending) { (void)hipEventDestroy(t.start); (void)hipEventDestroy(t.stop); }
    comm->pending.clear();
    for (int k = 0; k < GRX_COMM_KINDS; ++k) { comm->calls[k] = 0; comm->ms[k] = 0.0; }
    return GRX_OK;
}

int grx_comm_timing_read(grx_comm *comm, int kind, long long *calls, double *ms)
{
    GRX_REQUIRE(comm != nullptr && kind >= 0 && kind < GRX_COMM_KINDS, "grx_comm_timing_read: bad arguments");
    for (auto &t : comm->pending) {
        GRX_CHECK_HIP(hipEventSynchronize(t.stop));
        float e = 0.f;
        GRX_CHECK_HIP(hipEventElapsedTime(&e, t.start, t.stop));
        comm->ms[t.kind] += e;
        comm->calls[t.kind] += 1;
        (void)hipEventDestroy(t.start);
        (void)hipEventDestroy(t.stop);
    }
    comm->pending.clear();
    if (calls) *calls = comm->calls[kind];
    if (ms) *ms = comm->ms[kind];
    return GRX_OK;
}

}  // extern "C"
