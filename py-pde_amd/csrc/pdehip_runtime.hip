// pdehip_runtime.hip — device/stream/memory plumbing of the C ABI (include/pdehip.h).
#include "pdehip_common.h"

namespace pdehip {
static thread_local std::string g_last_error;
void set_error(const std::string &msg) { g_last_error = msg; }
}  // namespace pdehip

using namespace pdehip;

extern "C" {

const char *pdehip_last_error(void) { return g_last_error.c_str(); }
int pdehip_abi_version(void) { return PDEHIP_ABI_VERSION; }

int pdehip_device_count(int *count)
{
    if (!count) PDEHIP_FAIL(E_VALUE, "count is NULL");
    *count = 0;
    PDEHIP_HIP(hipGetDeviceCount(count));
    return 0;
}

int pdehip_set_device(int device)
{
    PDEHIP_HIP(hipSetDevice(device));
    return 0;
}

int pdehip_device_name(char *buf, size_t len)
{
    if (!buf || len == 0) PDEHIP_FAIL(E_VALUE, "buffer is NULL");
    int dev = 0;
    PDEHIP_HIP(hipGetDevice(&dev));
    hipDeviceProp_t prop;
    PDEHIP_HIP(hipGetDeviceProperties(&prop, dev));
    snprintf(buf, len, "%s (%s, %d CUs)", prop.name, prop.gcnArchName, prop.multiProcessorCount);
    return 0;
}

int pdehip_malloc(void **ptr, size_t bytes)
{
    if (!ptr) PDEHIP_FAIL(E_VALUE, "ptr is NULL");
    *ptr = nullptr;
    if (bytes == 0) bytes = 16;
    PDEHIP_HIP(hipMalloc(ptr, bytes));
    PDEHIP_HIP(hipMemset(*ptr, 0, bytes));
    return 0;
}

int pdehip_free(void *ptr)
{
    if (ptr) PDEHIP_HIP(hipFree(ptr));
    return 0;
}

int pdehip_memset(void *ptr, int value, size_t bytes, void *stream)
{
    PDEHIP_HIP(hipMemsetAsync(ptr, value, bytes, as_stream(stream)));
    return 0;
}

int pdehip_memcpy_h2d(void *dst, const void *src_host, size_t bytes, void *stream)
{
    PDEHIP_HIP(hipMemcpyAsync(dst, src_host, bytes, hipMemcpyHostToDevice, as_stream(stream)));
    PDEHIP_HIP(hipStreamSynchronize(as_stream(stream)));  // pageable host memory: keep it simple & safe
    return 0;
}

int pdehip_memcpy_d2h(void *dst_host, const void *src, size_t bytes, void *stream)
{
    PDEHIP_HIP(hipMemcpyAsync(dst_host, src, bytes, hipMemcpyDeviceToHost, as_stream(stream)));
    PDEHIP_HIP(hipStreamSynchronize(as_stream(stream)));
    return 0;
}

int pdehip_memcpy_d2d(void *dst, const void *src, size_t bytes, void *stream)
{
    PDEHIP_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, as_stream(stream)));
    return 0;
}

int pdehip_stream_create(void **stream)
{
    if (!stream) PDEHIP_FAIL(E_VALUE, "stream is NULL");
    hipStream_t s;
    PDEHIP_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    *stream = s;
    return 0;
}
int pdehip_stream_destroy(void *stream)
{
    if (stream) PDEHIP_HIP(hipStreamDestroy(as_stream(stream)));
    return 0;
}
int pdehip_stream_synchronize(void *stream)
{
    PDEHIP_HIP(hipStreamSynchronize(as_stream(stream)));
    return 0;
}
int pdehip_stream_wait_event(void *stream, void *event)
{
    PDEHIP_HIP(hipStreamWaitEvent(as_stream(stream), reinterpret_cast<hipEvent_t>(event), 0));
    return 0;
}
int pdehip_event_create(void **event)
{
    if (!event) PDEHIP_FAIL(E_VALUE, "event is NULL");
    hipEvent_t e;
    PDEHIP_HIP(hipEventCreate(&e));
    *event = e;
    return 0;
}
int pdehip_event_destroy(void *event)
{
    if (event) PDEHIP_HIP(hipEventDestroy(reinterpret_cast<hipEvent_t>(event)));
    return 0;
}
int pdehip_event_record(void *event, void *stream)
{
    PDEHIP_HIP(hipEventRecord(reinterpret_cast<hipEvent_t>(event), as_stream(stream)));
    return 0;
}
int pdehip_event_synchronize(void *event)
{
    PDEHIP_HIP(hipEventSynchronize(reinterpret_cast<hipEvent_t>(event)));
    return 0;
}
int pdehip_event_elapsed_ms(void *start, void *stop, float *ms)
{
    if (!ms) PDEHIP_FAIL(E_VALUE, "ms is NULL");
    PDEHIP_HIP(hipEventElapsedTime(ms, reinterpret_cast<hipEvent_t>(start), reinterpret_cast<hipEvent_t>(stop)));
    return 0;
}

int pdehip_layout(const pdehip_grid_t *g, int64_t *out8)
{
    NGrid n;
    PDEHIP_TRY(norm_grid(g, &n));
    if (!out8) PDEHIP_FAIL(E_VALUE, "out is NULL");
    out8[0] = n.p[0]; out8[1] = n.p[1]; out8[2] = n.pc; out8[3] = n.off; out8[4] = n.lpad;
    out8[5] = n.pc + kAllocSlack;          // elements to allocate per component-stack: ncomp*pc + slack
    out8[6] = kAllocSlack;
    out8[7] = n.p[3 - n.ndim];             // pitch of one layer along the grid's axis 0
    return 0;
}

}  // extern "C"
