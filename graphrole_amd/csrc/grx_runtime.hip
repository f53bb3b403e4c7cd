// grx_runtime.hip -- error state, device info, memory / stream / event helpers of the C ABI.
#include <cstdarg>
#include <cstring>

#include "grx_common.h"

static thread_local char g_err[1024] = "";

void grx_set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" {

int grx_version(void) { return GRX_VERSION; }

const char *grx_last_error(void) { return g_err; }

int grx_device_info(int *cu_count, int *wave_size, char *arch_buf, size_t arch_buf_len)
{
    int dev = 0;
    GRX_CHECK_HIP(hipGetDevice(&dev));
    hipDeviceProp_t prop;
    GRX_CHECK_HIP(hipGetDeviceProperties(&prop, dev));
    if (cu_count) *cu_count = prop.multiProcessorCount;
    if (wave_size) *wave_size = prop.warpSize;
    if (arch_buf && arch_buf_len) {
        strncpy(arch_buf, prop.gcnArchName, arch_buf_len - 1);
        arch_buf[arch_buf_len - 1] = 0;
    }
    return GRX_OK;
}

int grx_dev_malloc(void **d_out, size_t bytes)
{
    GRX_REQUIRE(d_out != nullptr, "grx_dev_malloc: d_out is NULL");
    GRX_CHECK_HIP(hipMalloc(d_out, bytes ? bytes : 1));
    return GRX_OK;
}

int grx_dev_free(void *d_ptr)
{
    GRX_CHECK_HIP(hipFree(d_ptr));
    return GRX_OK;
}

int grx_memcpy_h2d(void *d_dst, const void *h_src, size_t bytes, void *stream)
{
    GRX_CHECK_HIP(hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, grx_stream(stream)));
    return GRX_OK;
}

int grx_memcpy_d2h(void *h_dst, const void *d_src, size_t bytes, void *stream)
{
    GRX_CHECK_HIP(hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, grx_stream(stream)));
    return GRX_OK;
}

int grx_memset(void *d_dst, int value, size_t bytes, void *stream)
{
    GRX_CHECK_HIP(hipMemsetAsync(d_dst, value, bytes, grx_stream(stream)));
    return GRX_OK;
}

int grx_stream_sync(void *stream)
{
    GRX_CHECK_HIP(hipStreamSynchronize(grx_stream(stream)));
    return GRX_OK;
}

int grx_event_create(void **event_out)
{
    GRX_REQUIRE(event_out != nullptr, "grx_event_create: NULL");
    hipEvent_t ev;
    GRX_CHECK_HIP(hipEventCreate(&ev));
    *event_out = ev;
    return GRX_OK;
}

int grx_event_destroy(void *event)
{
    GRX_CHECK_HIP(hipEventDestroy(reinterpret_cast<hipEvent_t>(event)));
    return GRX_OK;
}

int grx_event_record(void *event, void *stream)
{
    GRX_CHECK_HIP(hipEventRecord(reinterpret_cast<hipEvent_t>(event), grx_stream(stream)));
    return GRX_OK;
}

int grx_event_elapsed_ms(void *start, void *stop, float *ms_out)
{
    GRX_REQUIRE(ms_out != nullptr, "grx_event_elapsed_ms: NULL");
    GRX_CHECK_HIP(hipEventSynchronize(reinterpret_cast<hipEvent_t>(stop)));
    GRX_CHECK_HIP(hipEventElapsedTime(ms_out, reinterpret_cast<hipEvent_t>(start),
                                      reinterpret_cast<hipEvent_t>(stop)));
    return GRX_OK;
}

}  // extern "C"
