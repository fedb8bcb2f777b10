// pdehip_runtime.hip — device/stream/memory plumbing of the C ABI (include/pdehip.h).
#include <cstdarg>
#include "pdehip_common.h"
#include <cstring>
#include <mutex>
#include <thread>

namespace pdehip {
static thread_local std::string g_last_error;
void set_error(const std::string &msg) { g_last_error = msg; }
// the stencil kernel instance the calling thread launched last (pdehip_last_kernel_name: bench.py labels its roofline with what actually ran)
static thread_local char g_last_kernel[192] = "";
void note_kernel(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    const int len = vsnprintf(g_last_kernel, sizeof(g_last_kernel), fmt, ap);
    va_end(ap);
    if (fastmath_on() && len > 0 && len < (int)sizeof(g_last_kernel) - 40) snprintf(g_last_kernel + len, sizeof(g_last_kernel) - len, " [fastmath: FMA contraction]");
}
}  // namespace pdehip

using namespace pdehip;

// ---------------------------------------------------------------------------------------------
// Host <-> device transfers of a field's valid data where it lies in host memory: the reference's fields are numpy arrays with
// ghost cells, `field.data` is a strided window of them (pde/fields/base.py:116-160), and every eq.solve starts with an upload
// and ends with a download of that window.  hipMemcpy moves CONTIGUOUS pageable memory at 56 GB/s on the MI355X boxes
// (profiles/r02_time_transfers.md) - nothing to add there - but a window first needs a contiguous host copy (numpy: 105 ms per
// GB in, 49 ms per GB out).  Here the rows of the window are gathered / scattered straight into pinned chunks while the DMA
// engine moves the previous chunk: kCopyThreads threads for large fields, each with two pinned chunks and a stream.  The call
// returns when the whole transfer is complete.
// ---------------------------------------------------------------------------------------------
namespace {
constexpr size_t kPipelinedMin = 32u << 20;   // below this one lane on the calling thread
constexpr size_t kChunk = 4u << 20;
constexpr int kCopyThreads = 4;

struct CopyLane {
    void *pinned[2] = {nullptr, nullptr};
    hipStream_t st = nullptr;
    hipEvent_t ev[2] = {nullptr, nullptr};
};
struct CopyPool {
    std::mutex m;   // one pipelined transfer at a time
    int device = -1;
    CopyLane lane[kCopyThreads];
};
CopyPool g_pool;

int pool_prepare(int dev)
{
    if (g_pool.device == dev) return 0;
    if (g_pool.device >= 0) PDEHIP_FAIL(E_RUNTIME, "pinned transfer pool belongs to device %d (one device per process)", g_pool.device);
    for (auto &l : g_pool.lane) {
        for (int b = 0; b < 2; b++) {
            PDEHIP_HIP(hipHostMalloc(&l.pinned[b], kChunk, hipHostMallocDefault));
            PDEHIP_HIP(hipEventCreateWithFlags(&l.ev[b], hipEventDisableTiming));
        }
        PDEHIP_HIP(hipStreamCreateWithFlags(&l.st, hipStreamNonBlocking));
    }
    g_pool.device = dev;
    return 0;
}

// the host side of a transfer: a byte stream of `rows` rows of `row` bytes, each contiguous in host memory
struct HostView {
    char *base;
    long n[3];          // components, axis 0, axis 1 (rows = n[0] * n[1] * n[2])
    int64_t s[3];       // byte strides of the three
    size_t row;         // bytes of one row (fastest axis)
    bool contiguous;
};
void host_copy(const HostView &h, size_t off, size_t len, char *pinned, bool to_host)
{
    if (h.contiguous) {
        if (to_host) memcpy(h.base + off, pinned, len); else memcpy(pinned, h.base + off, len);
        return;
    }
    while (len) {
        const size_t r = off / h.row, in = off % h.row, take = (len < h.row - in) ? len : h.row - in;
        const size_t j = r % h.n[2], t = r / h.n[2], i = t % h.n[1], c = t / h.n[1];
        char *p = h.base + (int64_t)c * h.s[0] + (int64_t)i * h.s[1] + (int64_t)j * h.s[2] + in;
        if (to_host) memcpy(p, pinned, take); else memcpy(pinned, p, take);
        off += take; len -= take; pinned += take;
    }
}

// chunks c = lane, lane + nlanes, ... of the transfer; returns a HIP error code (0 = fine)
hipError_t lane_run(int dev, CopyLane &l, int lane, int nlanes, char *devp, const HostView &host, size_t bytes, bool h2d)
{
    hipError_t rc = hipSetDevice(dev);   // a new thread starts on device 0
    if (rc != hipSuccess) return rc;
    const size_t nchunks = (bytes + kChunk - 1) / kChunk;
    int b = 0;
    size_t pending_off[2] = {0, 0}, pending_len[2] = {0, 0};   // d2h: chunk whose DMA is in flight in buffer b
    for (size_t c = lane; c < nchunks; c += nlanes, b ^= 1) {
        const size_t off = c * kChunk, len = (off + kChunk <= bytes) ? kChunk : bytes - off;
        // the buffer is free again once the transfer issued two rounds ago is done
        if ((rc = hipEventSynchronize(l.ev[b])) != hipSuccess) return rc;
        if (h2d) {
            host_copy(host, off, len, (char *)l.pinned[b], false);
            if ((rc = hipMemcpyAsync(devp + off, l.pinned[b], len, hipMemcpyHostToDevice, l.st)) != hipSuccess) return rc;
        } else {
            if (pending_len[b]) host_copy(host, pending_off[b], pending_len[b], (char *)l.pinned[b], true);
            if ((rc = hipMemcpyAsync(l.pinned[b], devp + off, len, hipMemcpyDeviceToHost, l.st)) != hipSuccess) return rc;
            pending_off[b] = off; pending_len[b] = len;
        }
        if ((rc = hipEventRecord(l.ev[b], l.st)) != hipSuccess) return rc;
    }
    if ((rc = hipStreamSynchronize(l.st)) != hipSuccess) return rc;
    if (!h2d) {
        // the (up to) two chunks still in the pinned buffers
        for (int k = 0; k < 2; k++, b ^= 1)
            if (pending_len[b]) { host_copy(host, pending_off[b], pending_len[b], (char *)l.pinned[b], true); pending_len[b] = 0; }
    }
    return hipSuccess;
}

// `devp` <-> host view, complete on return.  Large transfers: kCopyThreads lanes; small ones: one lane on the calling thread.
int pipelined_copy(void *devp, const HostView &host, size_t bytes, bool h2d, hipStream_t user_stream)
{
    int dev = 0;
    PDEHIP_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> guard(g_pool.m);
    PDEHIP_TRY(pool_prepare(dev));
    // everything queued on the caller's stream (producers of the source, readers of the destination) first
    PDEHIP_HIP(hipStreamSynchronize(user_stream));
    if (bytes < kPipelinedMin) {
        PDEHIP_HIP(lane_run(dev, g_pool.lane[0], 0, 1, (char *)devp, host, bytes, h2d));
        return 0;
    }
    hipError_t rcs[kCopyThreads];
    std::thread th[kCopyThreads];
    for (int t = 0; t < kCopyThreads; t++)
        th[t] = std::thread([&, t] { rcs[t] = lane_run(dev, g_pool.lane[t], t, kCopyThreads, (char *)devp, host, bytes, h2d); });
    for (auto &t : th) t.join();
    for (int t = 0; t < kCopyThreads; t++) PDEHIP_HIP(rcs[t]);
    return 0;
}
// valid cells of a field in host memory (any strides with a contiguous fastest axis) <-> the device's full layout
int transfer_valid(const pdehip_grid_t *g, int ncomp, void *host, const int64_t *hs, void *full, bool upload, void *stream)
{
    NGrid n;
    PDEHIP_TRY(norm_grid(g, &n));
    if (!host || !hs || !full) PDEHIP_FAIL(E_VALUE, "transfer of valid data: NULL pointer");
    if (ncomp < 1) PDEHIP_FAIL(E_VALUE, "number of components must be positive (%d)", ncomp);
    const long es = elem_size(n.dtype);
    if (hs[3] != es) PDEHIP_FAIL(E_VALUE, "host array must be contiguous along the fastest axis (stride %ld, element %ld bytes)", (long)hs[3], es);
    HostView h;
    h.base = (char *)host;
    h.row = (size_t)(n.n[2] * es);
    // rows: (component, axis 0, axis 1) of the normalised 3-D shape; 2-D grids have n[0] == 1, 1-D grids n[0] == n[1] == 1
    if (n.ndim == 3) { h.n[0] = ncomp; h.n[1] = n.n[0]; h.n[2] = n.n[1]; h.s[0] = hs[0]; h.s[1] = hs[1]; h.s[2] = hs[2]; }
    else if (n.ndim == 2) { h.n[0] = 1; h.n[1] = ncomp; h.n[2] = n.n[1]; h.s[0] = 0; h.s[1] = hs[0]; h.s[2] = hs[2]; }
    else { h.n[0] = 1; h.n[1] = 1; h.n[2] = ncomp; h.s[0] = 0; h.s[1] = 0; h.s[2] = hs[0]; }
    const size_t bytes = (size_t)ncomp * n.n[0] * n.n[1] * h.row;
    h.contiguous = (h.n[2] == 1 || h.s[2] == (int64_t)h.row) && (h.n[1] == 1 || h.s[1] == (int64_t)(h.n[2] * h.row)) &&
                   (h.n[0] == 1 || h.s[0] == (int64_t)(h.n[1] * h.n[2] * h.row));
    void *stage = nullptr;
    PDEHIP_HIP(hipMalloc(&stage, bytes ? bytes : 16));
    int rc = 0;
    if (upload) {
        rc = h.contiguous ? pdehip_memcpy_h2d(stage, host, bytes, stream) : pipelined_copy(stage, h, bytes, true, as_stream(stream));
        if (rc == 0) rc = pdehip_valid_to_full(g, ncomp, stage, full, stream);
        if (rc == 0 && hipStreamSynchronize(as_stream(stream)) != hipSuccess) { set_error("hipStreamSynchronize failed after upload"); rc = E_RUNTIME; }
    } else {
        rc = pdehip_full_to_valid(g, ncomp, full, stage, stream);
        if (rc == 0) rc = h.contiguous ? pdehip_memcpy_d2h(host, stage, bytes, stream) : pipelined_copy(stage, h, bytes, false, as_stream(stream));
    }
    (void)hipFree(stage);
    return rc;
}
}  // namespace

extern "C" {

const char *pdehip_last_error(void) { return g_last_error.c_str(); }
const char *pdehip_last_kernel_name(void) { return g_last_kernel; }
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
    static std::mutex m;
    static bool loaded = false;
    std::lock_guard<std::mutex> guard(m);
    if (!loaded) {
        PDEHIP_TRY(preload_stencil_kernels());
        PDEHIP_TRY(preload_e2_kernels());
        PDEHIP_TRY(preload_t2_kernels());
        PDEHIP_TRY(preload_shell_kernels());
        loaded = true;
    }
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
    // The fill runs on the null stream and has not necessarily finished when hipMemset returns; the callers' streams are non-blocking
    // (pdehip_stream_create) and do not order themselves after it.  Without this wait, work on such a stream that writes a fresh
    // buffer can be overtaken by the fill: measured with tools/stress_slab_1d.py as all-zero results in 15-24 % of tiny slab runs
    // (none in 7500 with the wait; profiles/r03_malloc_fill_race.md).
    PDEHIP_HIP(hipStreamSynchronize(nullptr));
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

int pdehip_upload_valid(const pdehip_grid_t *g, int ncomp, const void *host, const int64_t *host_strides, void *full, void *stream)
{ return transfer_valid(g, ncomp, const_cast<void *>(host), host_strides, full, true, stream); }
int pdehip_download_valid(const pdehip_grid_t *g, int ncomp, const void *full, void *host, const int64_t *host_strides, void *stream)
{ return transfer_valid(g, ncomp, host, host_strides, const_cast<void *>(full), false, stream); }

int pdehip_memcpy_d2d(void *dst, const void *src, size_t bytes, void *stream)
{
    PDEHIP_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, as_stream(stream)));
    return 0;
}

// device-to-device copy by a kernel: one 16-byte vector per thread, workgroups in address order, streaming (non-temporal) stores - the
// access order that reaches the highest copy rate on MI355X (6.2-6.55 TB/s at 1 GiB: profiles/r03_microbench4_copy_ceiling.log; hipMemcpyDtoD:
// 4.8-5.2).  bench.py prices the stencil kernels against it (`frac_of_nt_copy`); transfers between resident arrays may use it as well.
namespace {
typedef float pdehip_f4 __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(256) copy_nt_kernel(const pdehip_f4 *__restrict__ in, pdehip_f4 *__restrict__ out, long n)
{
    const long i = blockIdx.x * 256L + threadIdx.x;
    if (i < n) __builtin_nontemporal_store(in[i], out + i);
}
}  // namespace
int pdehip_copy_nt(void *dst, const void *src, size_t bytes, void *stream)
{
    if (!dst || !src) PDEHIP_FAIL(E_VALUE, "copy_nt: NULL pointer");
    if (bytes % 16 || (uintptr_t)dst % 16 || (uintptr_t)src % 16) PDEHIP_FAIL(E_VALUE, "copy_nt: 16-byte aligned pointers and a multiple of 16 bytes");
    const long n = (long)(bytes / 16);
    if (!n) return 0;
    hipLaunchKernelGGL(copy_nt_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, as_stream(stream), (const pdehip_f4 *)src, (pdehip_f4 *)dst, n);
    PDEHIP_HIP(hipGetLastError());
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
