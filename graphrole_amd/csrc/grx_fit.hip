// grx_fit.hip -- whole-loop drivers of the RolX half below the C ABI (no device code of its own).
//
// The per-kernel entry points of grx_nmf.hip are enough to drive the factorisation from any host
// language, but every call crossing the boundary costs interpreter time (ctypes, allocator, one
// synchronous copy per read-back): on the bench graph the initialisation alone spent ~0.45 ms of
// host time between its three device passes.  grx_nmf_fit / grx_nmf_mu run the same sequence --
// sklearn's NMF(solver='mu', init='nndsvda') as graphrole/roles/factor.py:19-25 calls it -- from
// C++: kernels are enqueued back to back, the small matrices come back through one pinned staging
// buffer, the k x F algebra between the passes is the library's own (grx_host_linalg.hip), and the
// stopping rule (_nmf.py:872-885) reads one small block per ten iterations.
#include "grx_common.h"

#include <cmath>
#include <cstring>
#include <vector>

namespace {

constexpr int MAX_WORLD = 64;     // ranks of a sharded fit (8 per MI355X node)

// Pinned staging buffer for the small device -> host read-backs (Gram matrices, projection
// statistics, the [A | B] block and H at a convergence check).  One per host thread, grown on demand.
struct Staging {
    void *ptr = nullptr;
    size_t bytes = 0;
    ~Staging() { if (ptr) (void)hipHostFree(ptr); }
};
thread_local Staging g_stage;

int staging(size_t bytes, double **out)
{
    if (g_stage.bytes < bytes) {
        if (g_stage.ptr) (void)hipHostFree(g_stage.ptr);
        g_stage.ptr = nullptr;
        g_stage.bytes = 0;
        size_t want = bytes < (1u << 20) ? (1u << 20) : bytes;
        GRX_CHECK_HIP(hipHostMalloc(&g_stage.ptr, want, hipHostMallocMapped));
        g_stage.bytes = want;
    }
    *out = reinterpret_cast<double *>(g_stage.ptr);
    return GRX_OK;
}

// one size for every driver of a fit: the buffer is then never reallocated while a copy of an earlier stage is
// still in flight (init: F*F + 1 + 4 r doubles; MU loop: [A | B], H and two scalars)
size_t staging_bytes(int F, int r)
{
    const size_t a = (size_t)F * F + 1 + (size_t)MAX_WORLD * r * 4;
    const size_t b = (size_t)2 * r * F + (size_t)r * r + 8;
    return (a > b ? a : b) * 8;
}

// device -> pinned host, then wait for the stream: the values are valid on return
int fetch(double *h_dst, const double *d_src, size_t count, hipStream_t st)
{
    const int rc = grx_fetch_begin(h_dst, d_src, count * 8, st);
    return rc != GRX_OK ? rc : grx_fetch_wait(st);
}

struct FitLayout {
    size_t gram, project, mu, ab, small, total;      // byte offsets / sizes inside the workspace
};

FitLayout fit_layout(int64_t n, int F, int r)
{
    FitLayout L;
    size_t g = 0;                                      // the second Gram pass has k <= F output columns
    for (int k = 1; k <= F; ++k) { const size_t b = grx_gram_workspace_bytes(n, k); if (b > g) g = b; }
    const size_t p = grx_project_workspace_bytes(n, r);
    const size_t m = grx_nmf_workspace_bytes(n, F, r);
    // the three phases never overlap in time: one region sized for the largest
    size_t big = g > p ? g : p;
    if (m > big) big = m;
    L.gram = L.project = L.mu = 0;
    L.ab = grx_align_up(big, 256);
    const size_t ab_bytes = grx_align_up(((size_t)r * F + (size_t)r * r) * 8, 256);
    L.small = L.ab + ab_bytes;
    // small outputs: Gram (F*F + 1), projection statistics (r * 4; sharded: one set per rank), error scalars
    const size_t small_bytes = grx_align_up(((size_t)F * F + 1 + (size_t)(MAX_WORLD + 1) * r * 4 + 8) * 8, 256);
    L.total = L.small + small_bytes;
    return L;
}

#define GRX_TRY(expr) do { int rc__ = (expr); if (rc__ != GRX_OK) return rc__; } while (0)

// node-range shard of a fit: this rank's rows of every O(n) pass (include/grx.h, "node-range sharding")
struct Shard {
    grx_comm *comm;
    int world, rank;
    int64_t rb, re;
};

int make_shard(const char *who, int64_t n, grx_comm *comm, const int64_t *h_bounds, Shard *out)
{
    out->comm = comm;
    out->world = comm ? grx_comm_world(comm) : 1;
    out->rank = comm ? grx_comm_rank(comm) : 0;
    out->rb = 0;
    out->re = n;
    if (!comm) return GRX_OK;
    GRX_REQUIRE(h_bounds != nullptr && h_bounds[0] == 0 && h_bounds[out->world] == n,
                "%s: a communicator needs the row partition h_bounds (0 .. n)", who);
    GRX_REQUIRE(out->world <= MAX_WORLD, "%s: at most %d ranks", who, MAX_WORLD);
    out->rb = h_bounds[out->rank];
    out->re = h_bounds[out->rank + 1];
    return GRX_OK;
}

int sum_over_ranks(const Shard &sh, double *d_buf, size_t count, void *stream)
{
    return sh.comm ? grx_comm_all_reduce(sh.comm, d_buf, count, GRX_F64, GRX_SUM, stream) : GRX_OK;
}

}  // namespace

extern "C" {

size_t grx_nmf_fit_workspace_bytes(int64_t n, int F, int r)
{
    if (F < 1) F = 1;
    if (r < 1) r = 1;
    return fit_layout(n, F, r).total;
}

static int nmf_mu_impl(int64_t n, int F, int r, const double *d_X, int64_t ldx, double *d_W, int64_t ldw, double *d_H,
                       double x_sq_norm, double tol, int max_iter, grx_nmf_info *info, const Shard &sh,
                       const int64_t *h_bounds, void *d_workspace, size_t workspace_bytes, void *stream)
{
    GRX_REQUIRE(n >= 1 && max_iter >= 0 && info != nullptr, "grx_nmf_mu: bad arguments");
    GRX_REQUIRE(d_X && d_W && d_H && d_workspace, "grx_nmf_mu: NULL pointer");
    const FitLayout L = fit_layout(n, F, r);
    if (workspace_bytes < L.total) {
        grx_set_error("grx_nmf_mu: workspace %zu < %zu", workspace_bytes, L.total);
        return GRX_ERR_WORKSPACE;
    }
    hipStream_t st = grx_stream(stream);
    char *ws = reinterpret_cast<char *>(d_workspace);
    void *mu_ws = ws + L.mu;
    const size_t mu_bytes = L.ab - L.mu;
    double *d_AB = reinterpret_cast<double *>(ws + L.ab);
    double *d_err = reinterpret_cast<double *>(ws + L.small);
    const size_t nA = (size_t)r * F, nB = (size_t)r * r;
    double *host = nullptr;
    GRX_TRY(staging(staging_bytes(F, r), &host));

    info->n_iter = 0;
    info->direct_residuals = 0;
    info->x_sq_norm = x_sq_norm;
    // initial error, _nmf.py:826 (always the direct pass: W0 H0 is far from X, but the identity's
    // inputs A, B do not exist yet).  It is only needed at the first convergence check: the scalar stays
    // on the device (d_err[1]) and comes back with that check's block -- one synchronisation fewer per fit.
    double *d_err_init = d_err + 1;
    GRX_TRY(grx_nmf_residual(n, F, r, d_X, ldx, d_W, ldw, sh.rb, sh.re, d_H, d_err_init, mu_ws, mu_bytes, stream));
    GRX_TRY(sum_over_ranks(sh, d_err_init, 1, stream));
    double *h_err_init = host + nA + nB + nA;
    bool have_init = false;
    double err_init = 0.0, prev = 0.0;
    int n_iter = 0;
    while (n_iter < max_iter) {
        const int step = (max_iter - n_iter) < 10 ? (max_iter - n_iter) : 10;
        const bool check = tol > 0.0 && (n_iter + step) % 10 == 0;
        // (a replayed HIP graph of the block was measured and removed: capture + instantiation cost 0.4 ms per fit,
        // and even a cached executable graph replayed slower than these launches -- DESIGN.md section 3)
        GRX_TRY(grx_nmf_iterate_rows(n, F, r, d_X, ldx, d_W, ldw, sh.rb, sh.re, d_H, d_AB, step, sh.comm, mu_ws, mu_bytes,
                                     stream));
        n_iter += step;
        if (!check) continue;
        // _nmf.py:872-885.  ||X - W H||_F from the W-pass outputs A = W^T X, B = W^T W and the updated H:
        // ||X||^2 - 2 <A, H> + <B, H H^T>  -- no pass over X.  The identity cancels when the fit is nearly
        // exact, so it is only trusted for a relative squared residual above 1e-8 (its own rounding error
        // is then far under the 1e-4 stopping tolerance); otherwise the direct kernel runs.
        double err = -1.0;
        if (!have_init) GRX_TRY(grx_fetch_begin(h_err_init, d_err_init, 8, st));
        if (x_sq_norm > 0.0) {
            GRX_TRY(grx_fetch_begin(host, d_AB, (nA + nB) * 8, st));
            GRX_TRY(fetch(host + nA + nB, d_H, nA, st));
            const double *A = host, *B = host + nA, *H = host + nA + nB;
            double ah = 0.0;
            for (size_t i = 0; i < nA; ++i) ah += A[i] * H[i];
            double bh = 0.0;
            for (int a = 0; a < r; ++a)
                for (int b = 0; b < r; ++b) {
                    double hh = 0.0;
                    for (int c = 0; c < F; ++c) hh += H[(size_t)a * F + c] * H[(size_t)b * F + c];
                    bh += B[(size_t)a * r + b] * hh;
                }
            const double sq = x_sq_norm - 2.0 * ah + bh;
            if (sq > 1e-8 * x_sq_norm) err = std::sqrt(sq);
        }
        if (err < 0.0) {
            GRX_TRY(grx_nmf_residual(n, F, r, d_X, ldx, d_W, ldw, sh.rb, sh.re, d_H, d_err, mu_ws, mu_bytes, stream));
            GRX_TRY(sum_over_ranks(sh, d_err, 1, stream));
            GRX_TRY(fetch(host, d_err, 1, st));
            err = std::sqrt(host[0]);
            info->direct_residuals += 1;
        }
        if (!have_init) {                                     // arrived with one of the fetches above
            err_init = prev = std::sqrt(*h_err_init);
            have_init = true;
        }
        info->err_last = err;
        if ((prev - err) / err_init < tol) break;
        prev = err;
    }
    if (!have_init) {                                         // no convergence check took place
        GRX_TRY(fetch(h_err_init, d_err_init, 1, st));
        err_init = std::sqrt(*h_err_init);
        info->err_last = err_init;
    }
    info->err_init = err_init;
    info->n_iter = n_iter;
    if (sh.comm && h_bounds) {
        // every rank returns the complete factor: the row slices of the r columns of W travel once per fit
        std::vector<void *> ptrs(r);
        for (int j = 0; j < r; ++j) ptrs[j] = d_W + (size_t)j * ldw;
        GRX_TRY(grx_comm_all_gather_rows(sh.comm, h_bounds, r, ptrs.data(), 8, stream));
        GRX_CHECK_HIP(hipStreamSynchronize(st));
    }
    return GRX_OK;
}

int grx_nmf_mu(int64_t n, int F, int r, const double *d_X, int64_t ldx, double *d_W, int64_t ldw, double *d_H,
               double x_sq_norm, double tol, int max_iter, grx_nmf_info *info, grx_comm *comm, const int64_t *h_bounds,
               void *d_workspace, size_t workspace_bytes, void *stream)
{
    Shard sh;
    GRX_TRY(make_shard("grx_nmf_mu", n, comm, h_bounds, &sh));
    return nmf_mu_impl(n, F, r, d_X, ldx, d_W, ldw, d_H, x_sq_norm, tol, max_iter, info, sh, h_bounds, d_workspace,
                       workspace_bytes, stream);
}

// wait_for_upload = false (grx_nmf_fit): the caller goes on enqueueing on the same stream and does not touch the
// staging buffer from the host before its next synchronisation
static int nmf_init_impl(int64_t n, int F, int r, const double *d_X, int64_t ldx, const double *h_omega, int n_over,
                         double *d_W, int64_t ldw, double *d_H, double *x_sq_norm, const Shard &sh, void *d_workspace,
                         size_t workspace_bytes, void *stream, bool wait_for_upload)
{
    GRX_REQUIRE(n >= 1 && F >= 1 && r >= 1 && n >= F && r <= F, "grx_nmf_init: needs n >= F >= r >= 1 (n=%lld F=%d r=%d)",
                (long long)n, F, r);
    GRX_REQUIRE(d_X && h_omega && d_W && d_H && d_workspace && n_over >= r, "grx_nmf_init: bad arguments");
    const FitLayout L = fit_layout(n, F, r);
    if (workspace_bytes < L.total) {
        grx_set_error("grx_nmf_init: workspace %zu < %zu", workspace_bytes, L.total);
        return GRX_ERR_WORKSPACE;
    }
    hipStream_t st = grx_stream(stream);
    char *ws = reinterpret_cast<char *>(d_workspace);
    double *d_small = reinterpret_cast<double *>(ws + L.small);
    const size_t big_bytes = L.ab;
    double *host = nullptr;
    GRX_TRY(staging(staging_bytes(F, r), &host));

    // X = Q M with orthonormal Q from two Gram passes: eigh(X^T X) whitens, the second pass
    // re-orthogonalises; sklearn's randomized_svd of X (extmath.py:531-604) then runs on the k x F
    // matrix M with the caller's Gaussian test matrix (roles/factor.py docstring)
    // sharded: every rank scans its own rows, the Gram matrices (<= F x F doubles) are summed over the ranks
    GRX_TRY(grx_gram(n, F, d_X, ldx, sh.rb, sh.re, nullptr, F, d_small, ws + L.gram, big_bytes, stream));
    GRX_TRY(sum_over_ranks(sh, d_small, (size_t)F * F + 1, stream));
    GRX_TRY(fetch(host, d_small, (size_t)F * F + 1, st));
    const double x_mean = host[(size_t)F * F] / ((double)n * (double)F);
    double trace = 0.0;
    for (int i = 0; i < F; ++i) trace += host[(size_t)i * F + i];
    if (x_sq_norm) *x_sq_norm = trace;
    std::vector<double> T1((size_t)F * F), lam(F), V((size_t)F * F);
    int k = 0;
    GRX_TRY(grx_host_whiten_for_rank(F, host, r, T1.data(), lam.data(), V.data(), &k));
    if (k == 0) {
        grx_set_error("NMF initialisation: the feature matrix is numerically zero");
        return GRX_ERR_DEGENERATE;
    }
    GRX_TRY(grx_gram(n, F, d_X, ldx, sh.rb, sh.re, T1.data(), k, d_small, ws + L.gram, big_bytes, stream));
    GRX_TRY(sum_over_ranks(sh, d_small, (size_t)k * k, stream));
    GRX_TRY(fetch(host, d_small, (size_t)k * k, st));
    const int lo = n < F ? (int)n : F;
    const int n_iter = (r < 0.1 * lo) ? 7 : 4;                       // extmath.py:557-560
    std::vector<double> Z((size_t)F * r), S(r), Vt((size_t)r * F);
    GRX_TRY(grx_host_range_finder(F, k, T1.data(), lam.data(), V.data(), host, h_omega, n_over, r, n_iter, Z.data(),
                                  S.data(), Vt.data()));
    double *d_stats = d_small;
    GRX_TRY(grx_project(n, F, d_X, ldx, sh.rb, sh.re, Z.data(), r, d_W, ldw, d_stats, ws + L.project, big_bytes, stream));
    if (sh.comm) {
        // the column statistics of disjoint row ranges: every rank collects all of them and merges on the host -- the
        // entry of largest magnitude (the lowest row wins a tie, like argmax over the whole column), sums of squares
        const size_t cnt = (size_t)r * 4;
        double *d_all = d_stats + cnt;
        std::vector<grx_p2p_op> ops;
        for (int q = 0; q < sh.world; ++q) {
            ops.push_back({0, q, d_stats, cnt * 8});
            ops.push_back({1, q, d_all + (size_t)q * cnt, cnt * 8});
        }
        GRX_TRY(grx_comm_exchange(sh.comm, (int)ops.size(), ops.data(), stream));
        GRX_TRY(fetch(host, d_all, cnt * sh.world, st));
        std::vector<double> merged(cnt);
        for (int j = 0; j < r; ++j) {
            int best = 0;
            for (int p = 1; p < sh.world; ++p) {
                const double a = std::fabs(host[(size_t)p * cnt + j * 4]), b = std::fabs(host[(size_t)best * cnt + j * 4]);
                if (a > b || (a == b && host[(size_t)p * cnt + j * 4 + 1] < host[(size_t)best * cnt + j * 4 + 1])) best = p;
            }
            merged[j * 4 + 0] = host[(size_t)best * cnt + j * 4 + 0];
            merged[j * 4 + 1] = host[(size_t)best * cnt + j * 4 + 1];
            double pos = 0.0, neg = 0.0;
            for (int p = 0; p < sh.world; ++p) { pos += host[(size_t)p * cnt + j * 4 + 2]; neg += host[(size_t)p * cnt + j * 4 + 3]; }
            merged[j * 4 + 2] = pos;
            merged[j * 4 + 3] = neg;
        }
        std::memcpy(host, merged.data(), cnt * 8);
    } else {
        GRX_TRY(fetch(host, d_stats, (size_t)r * 4, st));
    }
    std::vector<double> sign(r), scale(r), H((size_t)r * F);
    GRX_TRY(grx_host_nndsvd_plan(r, F, S.data(), Vt.data(), host, sign.data(), scale.data(), H.data()));
    const double eps = 1e-6;                                          // _nmf.py:354-359
    GRX_TRY(grx_nndsvd_apply(n, r, d_W, ldw, sh.rb, sh.re, sign.data(), scale.data(), eps, x_mean, stream));
    for (size_t i = 0; i < (size_t)r * F; ++i) {
        if (H[i] < eps) H[i] = 0.0;
        if (H[i] == 0.0) H[i] = x_mean;
    }
    // through the pinned buffer: a pageable source would make the asynchronous copy synchronous anyway
    std::memcpy(host, H.data(), (size_t)r * F * 8);
    GRX_CHECK_HIP(hipMemcpyAsync(d_H, host, (size_t)r * F * 8, hipMemcpyHostToDevice, st));
    if (wait_for_upload) GRX_CHECK_HIP(hipStreamSynchronize(st));     // `host` is reused by the next call
    return GRX_OK;
}

int grx_nmf_init(int64_t n, int F, int r, const double *d_X, int64_t ldx, const double *h_omega, int n_over,
                 double *d_W, int64_t ldw, double *d_H, double *x_sq_norm, grx_comm *comm, const int64_t *h_bounds,
                 void *d_workspace, size_t workspace_bytes, void *stream)
{
    Shard sh;
    GRX_TRY(make_shard("grx_nmf_init", n, comm, h_bounds, &sh));
    return nmf_init_impl(n, F, r, d_X, ldx, h_omega, n_over, d_W, ldw, d_H, x_sq_norm, sh, d_workspace, workspace_bytes,
                         stream, true);
}

int grx_nmf_fit(int64_t n, int F, int r, const double *d_X, int64_t ldx, const double *h_omega, int n_over,
                double tol, int max_iter, double *d_W, int64_t ldw, double *d_H, grx_nmf_info *info, grx_comm *comm,
                const int64_t *h_bounds, void *d_workspace, size_t workspace_bytes, void *stream)
{
    GRX_REQUIRE(info != nullptr, "grx_nmf_fit: info is NULL");
    Shard sh;
    GRX_TRY(make_shard("grx_nmf_fit", n, comm, h_bounds, &sh));
    double xx = -1.0;
    GRX_TRY(nmf_init_impl(n, F, r, d_X, ldx, h_omega, n_over, d_W, ldw, d_H, &xx, sh, d_workspace, workspace_bytes, stream,
                          false));
    return nmf_mu_impl(n, F, r, d_X, ldx, d_W, ldw, d_H, xx, tol, max_iter, info, sh, h_bounds, d_workspace,
                       workspace_bytes, stream);
}

}  // extern "C"
