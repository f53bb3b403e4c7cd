// grx_hostio.hip -- the host <-> device boundary of the two public calls (host code + two trivial kernels).
//
// The reference hands a pandas DataFrame from extract_features() to extract_role_factors()
// (graphrole/roles/extract.py:59-93): on this device path that table is the only bulk data that ever crosses PCIe.
// A plain hipMemcpy to / from pageable memory runs at a fraction of the link rate (the runtime stages through one
// pinned buffer and the host-side memcpy, single-threaded, also takes the first-touch page faults of a fresh
// destination).  Here:
//   grx_download / grx_upload   chunks through a ring of pinned staging buffers, the device copies asynchronous, the
//                               host-side memcpy of chunk i on a small thread pool while chunk i + 1 is on the link
//   grx_upload_i64_as_i32       the same for edge arrays given as int64 (numpy's default): narrowed while staging
//   grx_host_checksums          per-column content hashes of a host table, threaded -- how a result table is
//                               recognised as unmodified when it comes back (features/handoff.py)
//   grx_min_value               min over a feature-major device matrix (sklearn's "Negative values in data" check,
//                               _nmf.py:283, without a host pass)
#include "grx_common.h"

#include <atomic>
#include <cmath>
#include <condition_variable>
#include <cstring>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

namespace {

// ---------------------------------------------------------------------------------------------------------------
// a tiny fork-join pool: run(k, fn) calls fn(0..k-1) on the workers + the caller and returns when all are done
// ---------------------------------------------------------------------------------------------------------------
class Pool {
public:
    static Pool &get()
    {
        static Pool *p = new Pool;       // never destroyed: the workers are detached and may outlive static teardown
        return *p;
    }
    int size() const { return (int)workers_.size() + 1; }
    void run(int tasks, const std::function<void(int)> &fn)
    {
        if (tasks <= 0) return;
        if (tasks == 1 || workers_.empty()) { for (int i = 0; i < tasks; ++i) fn(i); return; }
        std::unique_lock<std::mutex> call(call_mu_);           // one fork-join at a time
        {
            std::unique_lock<std::mutex> lk(mu_);
            // Nobody may be between "draw an index" and "compare it with the bound" while the bound changes: a
            // straggler of the previous call holding a stale index (>= the old task count) that resumed after the new
            // bound was stored would run a task a second time -- pending_ would reach 0 with a task still in flight
            // and run() would return under it.  Entering and leaving work() is counted under this mutex, and the job
            // is replaced only while the count is zero; whoever enters afterwards draws from the new counter and
            // sees the new function and bound.
            idle_.wait(lk, [&] { return active_ == 0; });
            fn_.store(&fn);
            total_.store(tasks);
            pending_ = tasks;
            next_.store(0);
            ++epoch_;
        }
        cv_.notify_all();
        work();
        std::unique_lock<std::mutex> lk(mu_);
        done_.wait(lk, [&] { return pending_ == 0; });
        total_.store(0);
    }

private:
    Pool()
    {
        unsigned hw = std::thread::hardware_concurrency();
        int n = hw >= 32 ? 11 : hw >= 8 ? 5 : hw >= 4 ? 2 : 0;   // + the caller
        for (int i = 0; i < n; ++i) workers_.emplace_back([this] { loop(); });
        for (auto &t : workers_) t.detach();
    }
    void work()
    {
        { std::lock_guard<std::mutex> lk(mu_); ++active_; }
        for (;;) {
            const int i = next_.fetch_add(1);
            if (i >= total_.load()) break;
            (*fn_.load())(i);
            std::lock_guard<std::mutex> lk(mu_);
            if (--pending_ == 0) done_.notify_all();
        }
        std::lock_guard<std::mutex> lk(mu_);
        if (--active_ == 0) idle_.notify_all();
    }
    void loop()
    {
        unsigned long seen = 0;
        for (;;) {
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_.wait(lk, [&] { return epoch_ != seen; });
                seen = epoch_;
            }
            work();
        }
    }
    std::vector<std::thread> workers_;
    std::mutex mu_, call_mu_;
    std::condition_variable cv_, done_, idle_;
    int active_ = 0;                                           // threads inside work() (guarded by mu_)
    std::atomic<const std::function<void(int)> *> fn_{nullptr};
    std::atomic<int> next_{0}, total_{0};
    int pending_ = 0;
    unsigned long epoch_ = 0;
};

void parallel_memcpy(void *dst, const void *src, size_t bytes)
{
    Pool &pool = Pool::get();
    const size_t grain = 1u << 20;
    int parts = (int)((bytes + grain - 1) / grain);
    if (parts > pool.size()) parts = pool.size();
    if (parts <= 1) { std::memcpy(dst, src, bytes); return; }
    const size_t per = ((bytes + parts - 1) / parts + 63) & ~(size_t)63;
    pool.run(parts, [&](int i) {
        const size_t off = (size_t)i * per;
        if (off >= bytes) return;
        const size_t len = bytes - off < per ? bytes - off : per;
        std::memcpy((char *)dst + off, (const char *)src + off, len);
    });
}

// ---------------------------------------------------------------------------------------------------------------
// pinned staging ring (one per host thread)
// ---------------------------------------------------------------------------------------------------------------
constexpr int RING = 4;
constexpr size_t CHUNK = 8u << 20;

struct Ring {
    void *buf[RING] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t ev[RING] = {nullptr, nullptr, nullptr, nullptr};
    bool ready = false;
    ~Ring()
    {
        for (int i = 0; i < RING; ++i) {
            if (buf[i]) (void)hipHostFree(buf[i]);
            if (ev[i]) (void)hipEventDestroy(ev[i]);
        }
    }
};
thread_local Ring g_ring;

int ring_init()
{
    if (g_ring.ready) return GRX_OK;
    for (int i = 0; i < RING; ++i) {
        GRX_CHECK_HIP(hipHostMalloc(&g_ring.buf[i], CHUNK, hipHostMallocDefault));
        GRX_CHECK_HIP(hipEventCreateWithFlags(&g_ring.ev[i], hipEventDisableTiming));
    }
    g_ring.ready = true;
    return GRX_OK;
}

// 64-bit mixer (splitmix64 finaliser)
inline uint64_t mix64(uint64_t x)
{
    x ^= x >> 30; x *= 0xbf58476d1ce4e5b9ull;
    x ^= x >> 27; x *= 0x94d049bb133111ebull;
    x ^= x >> 31;
    return x;
}

// position-dependent content hash of `words` 8-byte words starting at word index `first`: eight lanes (one cache
// line per step), one 32 x 32 -> 64 multiply per word (vpmuludq under -mavx2), a key that advances every step so that
// equal blocks at different positions hash differently
uint64_t hash_words(const uint64_t *p, size_t words, size_t first)
{
    static const uint64_t KEY[8] = {0xa0761d6478bd642full, 0xe7037ed1a0b428dbull, 0x8ebc6af09c88c6e3ull, 0x589965cc75374cc3ull,
                                    0x1d8e4e27c47d124full, 0xeb44accab455d165ull, 0x2d358dccaa6c78a5ull, 0x8bb84b93962eacc9ull};
    uint64_t acc[8], key[8];
    for (int l = 0; l < 8; ++l) { acc[l] = KEY[7 - l] ^ first; key[l] = KEY[l] + (uint64_t)first * 0x9e3779b97f4a7c15ull; }
    size_t i = 0;
    for (; i + 8 <= words; i += 8) {
        for (int l = 0; l < 8; ++l) {
            const uint64_t d = p[i + l];
            const uint64_t k = d ^ key[l];
            acc[l] += (uint64_t)(uint32_t)k * (k >> 32) + ((d << 32) | (d >> 32));
            key[l] += 0x9e3779b97f4a7c15ull;
        }
    }
    uint64_t h = 0;
    for (; i < words; ++i) h = mix64(h ^ p[i] ^ (uint64_t)(first + i));
    for (int l = 0; l < 8; ++l) h = mix64(h ^ acc[l]) + (uint64_t)l;
    return h;
}

__global__ __launch_bounds__(256) void min_value_kernel(int64_t n, int F, const double *__restrict__ X, int64_t ld,
                                                        double *__restrict__ partial)
{
    double m = INFINITY;
    bool nan = false;
    for (int c = blockIdx.y; c < F; c += gridDim.y) {
        const double *col = X + (size_t)c * ld;
        for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
            const double v = col[i];
            nan |= v != v;
            m = v < m ? v : m;
        }
    }
    if (nan) m = NAN;
    // wave minimum (NaN sticks: x < NaN is false, so carry it explicitly)
    for (int off = 32; off > 0; off >>= 1) {
        const double o = __shfl_xor(m, off, 64);
        m = (m != m || o != o) ? NAN : (o < m ? o : m);
    }
    __shared__ double s[4];
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w) m = (m != m || s[w] != s[w]) ? NAN : (s[w] < m ? s[w] : m);
        partial[blockIdx.y * gridDim.x + blockIdx.x] = m;
    }
}

__global__ __launch_bounds__(256) void min_finish_kernel(int count, const double *__restrict__ partial, double *__restrict__ out)
{
    double m = INFINITY;
    for (int i = threadIdx.x; i < count; i += 256) {
        const double v = partial[i];
        m = (m != m || v != v) ? NAN : (v < m ? v : m);
    }
    for (int off = 32; off > 0; off >>= 1) {
        const double o = __shfl_xor(m, off, 64);
        m = (m != m || o != o) ? NAN : (o < m ? o : m);
    }
    __shared__ double s[4];
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w) m = (m != m || s[w] != s[w]) ? NAN : (s[w] < m ? s[w] : m);
        out[0] = m;
    }
}

constexpr int MIN_GRID_X = 256, MIN_GRID_Y = 4;

}  // namespace

extern "C" {

int grx_download(void *h_dst, const void *d_src, size_t bytes, void *stream)
{
    if (bytes == 0) return GRX_OK;
    GRX_REQUIRE(h_dst && d_src, "grx_download: NULL pointer");
    hipStream_t st = grx_stream(stream);
    if (bytes <= (256u << 10)) {
        GRX_CHECK_HIP(hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, st));
        GRX_CHECK_HIP(hipStreamSynchronize(st));
        return GRX_OK;
    }
    int rc = ring_init();
    if (rc != GRX_OK) return rc;
    const size_t chunks = (bytes + CHUNK - 1) / CHUNK;
    // chunk c: device -> staging[c % RING] on the stream; the host copies staging -> destination once its event fired,
    // RING - 1 chunks behind the enqueue front, so the link and the host memcpy overlap
    for (size_t c = 0; c < chunks + RING - 1; ++c) {
        if (c >= (size_t)(RING - 1)) {
            const size_t d = c - (RING - 1);
            const size_t off = d * CHUNK, len = bytes - off < CHUNK ? bytes - off : CHUNK;
            GRX_CHECK_HIP(hipEventSynchronize(g_ring.ev[d % RING]));
            parallel_memcpy((char *)h_dst + off, g_ring.buf[d % RING], len);
        }
        if (c < chunks) {
            const size_t off = c * CHUNK, len = bytes - off < CHUNK ? bytes - off : CHUNK;
            GRX_CHECK_HIP(hipMemcpyAsync(g_ring.buf[c % RING], (const char *)d_src + off, len, hipMemcpyDeviceToHost, st));
            GRX_CHECK_HIP(hipEventRecord(g_ring.ev[c % RING], st));
        }
    }
    return GRX_OK;
}

static int upload_impl(void *d_dst, const void *h_src, size_t bytes, bool narrow, hipStream_t st)
{
    // narrow: h_src holds int64, the device receives int32 (bytes = device bytes)
    int rc = ring_init();
    if (rc != GRX_OK) return rc;
    const size_t chunks = (bytes + CHUNK - 1) / CHUNK;
    std::atomic<int> bad{0};
    for (size_t c = 0; c < chunks; ++c) {
        const size_t off = c * CHUNK, len = bytes - off < CHUNK ? bytes - off : CHUNK;
        if (c >= RING) GRX_CHECK_HIP(hipEventSynchronize(g_ring.ev[c % RING]));     // staging buffer free again
        void *stage = g_ring.buf[c % RING];
        if (!narrow) {
            parallel_memcpy(stage, (const char *)h_src + off, len);
        } else {
            const int64_t *src = reinterpret_cast<const int64_t *>(h_src) + off / 4;
            int32_t *dst = reinterpret_cast<int32_t *>(stage);
            const size_t cnt = len / 4;
            Pool &pool = Pool::get();
            const int parts = pool.size();
            const size_t per = (cnt + parts - 1) / parts;
            pool.run(parts, [&](int i) {
                const size_t b = (size_t)i * per, e = b + per < cnt ? b + per : cnt;
                int64_t any = 0;
                for (size_t k = b; k < e; ++k) { const int64_t v = src[k]; any |= v ^ (int64_t)(int32_t)v; dst[k] = (int32_t)v; }
                if (any) bad.store(1);
            });
        }
        GRX_CHECK_HIP(hipMemcpyAsync((char *)d_dst + off, stage, len, hipMemcpyHostToDevice, st));
        GRX_CHECK_HIP(hipEventRecord(g_ring.ev[c % RING], st));
    }
    GRX_CHECK_HIP(hipStreamSynchronize(st));
    GRX_REQUIRE(!bad.load(), "grx_upload_i64_as_i32: a value does not fit 32 bits");
    return GRX_OK;
}

int grx_upload(void *d_dst, const void *h_src, size_t bytes, void *stream)
{
    if (bytes == 0) return GRX_OK;
    GRX_REQUIRE(d_dst && h_src, "grx_upload: NULL pointer");
    hipStream_t st = grx_stream(stream);
    if (bytes <= (256u << 10)) {
        GRX_CHECK_HIP(hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, st));
        GRX_CHECK_HIP(hipStreamSynchronize(st));
        return GRX_OK;
    }
    return upload_impl(d_dst, h_src, bytes, false, st);
}

int grx_upload_i64_as_i32(int32_t *d_dst, const int64_t *h_src, size_t count, void *stream)
{
    if (count == 0) return GRX_OK;
    GRX_REQUIRE(d_dst && h_src, "grx_upload_i64_as_i32: NULL pointer");
    return upload_impl(d_dst, h_src, count * 4, true, grx_stream(stream));
}

int grx_host_checksums(const void *h_base, int ncols, size_t col_bytes, size_t stride_bytes, uint64_t *h_out)
{
    GRX_REQUIRE(ncols >= 0 && (ncols == 0 || (h_base && h_out)) && col_bytes % 8 == 0, "grx_host_checksums: bad arguments");
    if (ncols == 0) return GRX_OK;
    Pool &pool = Pool::get();
    const size_t words = col_bytes / 8;
    // fixed 512 KiB pieces: the hash of a column does not depend on how many columns or threads share the call
    const size_t per = 65536;
    const int pieces = (int)((words + per - 1) / per);
    if (pieces == 0) { for (int c = 0; c < ncols; ++c) h_out[c] = 0x243f6a8885a308d3ull; return GRX_OK; }
    std::vector<uint64_t> part((size_t)ncols * pieces, 0);
    pool.run(ncols * pieces, [&](int t) {
        const int c = t / pieces, k = t % pieces;
        const size_t b = (size_t)k * per;
        const size_t e = b + per < words ? b + per : words;
        const uint64_t *p = reinterpret_cast<const uint64_t *>(reinterpret_cast<const char *>(h_base) + (size_t)c * stride_bytes);
        part[t] = hash_words(p + b, e - b, b);
    });
    for (int c = 0; c < ncols; ++c) {
        uint64_t h = 0x243f6a8885a308d3ull ^ (uint64_t)words;
        for (int k = 0; k < pieces; ++k) h = mix64(h ^ part[(size_t)c * pieces + k]) + (uint64_t)k;
        h_out[c] = h;
    }
    return GRX_OK;
}

// RandomState.choice(m, p = uniform) for the ONE draw u it makes (sklearn's first k-means++ seed, _kmeans.py:221):
// numpy computes cdf = cumsum(full(m, 1 / m)); cdf /= cdf[-1]; index = searchsorted(cdf, u, side='right').  The running
// sums are plain sequential fp64 additions; restated as the same additions without materialising the three m-element
// arrays (0.1 s at 30 M samples in numpy).  Far from an integer crossing the index is floor-determined: the running
// sum is within (i + 1) * 2^-52 relative of (i + 1) / m, so the exact loops only run when u * m lies that close to one.
int grx_host_uniform_choice(int64_t m, double u, int64_t *index)
{
    GRX_REQUIRE(m >= 1 && index != nullptr && u >= 0.0 && u < 1.0, "grx_host_uniform_choice: bad arguments");
    const double p = 1.0 / (double)m;
    const double x = u * (double)m;
    const double slack = 4.0 * (double)m * 2.220446049250313e-16 * (x + 1.0) + 1e-6;   // index uncertainty, generous
    const double fl = std::floor(x);
    if (x - fl > slack && (fl + 1.0) - x > slack) { *index = (int64_t)fl; return GRX_OK; }
    double total = 0.0;
    for (int64_t i = 0; i < m; ++i) total += p;               // cdf[-1]
    double s = 0.0;
    int64_t count = 0;                                        // entries with cdf[i] / total <= u
    for (int64_t i = 0; i < m; ++i) {
        s += p;
        if (s / total <= u) count = i + 1; else break;        // cdf is non-decreasing
    }
    *index = count;
    return GRX_OK;
}

size_t grx_min_value_workspace_bytes(void) { return (size_t)MIN_GRID_X * MIN_GRID_Y * 8 + 256; }

int grx_min_value(int64_t n, int F, const double *d_X, int64_t ld, double *d_out, void *d_workspace, size_t workspace_bytes,
                  void *stream)
{
    GRX_REQUIRE(n >= 1 && F >= 1 && d_X && d_out && d_workspace && ld >= n, "grx_min_value: bad arguments");
    GRX_REQUIRE(workspace_bytes >= grx_min_value_workspace_bytes(), "grx_min_value: workspace too small");
    hipStream_t st = grx_stream(stream);
    double *partial = reinterpret_cast<double *>(d_workspace);
    int64_t want = grx_ceil_div(n, 256 * 8);
    const int gx = (int)(want > MIN_GRID_X ? MIN_GRID_X : want);
    const int gy = F < MIN_GRID_Y ? F : MIN_GRID_Y;
    min_value_kernel<<<dim3(gx, gy), 256, 0, st>>>(n, F, d_X, ld, partial);
    GRX_LAUNCH_CHECK();
    min_finish_kernel<<<1, 256, 0, st>>>(gx * gy, partial, d_out);
    GRX_LAUNCH_CHECK();
    return GRX_OK;
}

}  // extern "C"
