// microbench3.hip — HBM write/read/copy structure probes on MI355X (tool, not product).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                         \
    do {                                                                              \
        hipError_t e = (x);                                                           \
        if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } \
    } while (0)

typedef double d2 __attribute__((ext_vector_type(2)));

template <bool NT>
__global__ void __launch_bounds__(256) w_stride(d2 *out, long n, d2 v)
{
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += stride) {
        if (NT) __builtin_nontemporal_store(v, out + i);
        else out[i] = v;
    }
}
// each block owns a contiguous chunk
template <bool NT>
__global__ void __launch_bounds__(256) w_chunk(d2 *out, long n, d2 v)
{
    const long per = (n + gridDim.x - 1) / gridDim.x;
    const long b0 = blockIdx.x * per, b1 = (b0 + per < n) ? b0 + per : n;
    for (long i = b0 + threadIdx.x; i < b1; i += 256) {
        if (NT) __builtin_nontemporal_store(v, out + i);
        else out[i] = v;
    }
}
// each wave owns a contiguous chunk (wave-private streams)
__global__ void __launch_bounds__(256) w_wavechunk(d2 *out, long n, d2 v)
{
    const long nw = (long)gridDim.x * 4;
    const long wid = blockIdx.x * 4 + (threadIdx.x >> 6);
    const long per = (n + nw - 1) / nw;
    const long b0 = wid * per, b1 = (b0 + per < n) ? b0 + per : n;
    for (long i = b0 + (threadIdx.x & 63); i < b1; i += 64) out[i] = v;
}
// 64 B per lane per iteration (4 x dwordx4 to consecutive addresses)
__global__ void __launch_bounds__(256) w_wide(d2 *out, long n, d2 v)
{
    const long stride = (long)gridDim.x * blockDim.x * 4;
    for (long i = (blockIdx.x * (long)blockDim.x + threadIdx.x) * 4; i + 3 < n; i += stride) {
        out[i] = v; out[i + 1] = v; out[i + 2] = v; out[i + 3] = v;
    }
}
__global__ void __launch_bounds__(256) w_data(d2 *out, long n)
{
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += stride) {
        d2 v;
        v[0] = (double)(i * 2654435761u % 1000) * 1e-3;
        v[1] = v[0] * 1.37 + 0.11;
        out[i] = v;
    }
}
__global__ void __launch_bounds__(256) r_stride(const d2 *in, double *sink, long n)
{
    const long stride = (long)gridDim.x * blockDim.x;
    d2 acc = {0, 0};
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += stride) acc += in[i];
    if (acc[0] + acc[1] == 1.2345) sink[0] = acc[0];
}
template <bool NT>
__global__ void __launch_bounds__(256) c_chunk(const d2 *in, d2 *out, long n)
{
    const long per = (n + gridDim.x - 1) / gridDim.x;
    const long b0 = blockIdx.x * per, b1 = (b0 + per < n) ? b0 + per : n;
    long i = b0 + threadIdx.x;
    for (; i + 7 * 256 < b1; i += 8 * 256) {
        d2 v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) v[u] = NT ? __builtin_nontemporal_load(in + i + u * 256) : in[i + u * 256];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            if (NT) __builtin_nontemporal_store(v[u], out + i + u * 256);
            else out[i + u * 256] = v[u];
        }
    }
    for (; i < b1; i += 256) out[i] = in[i];
}

template <typename F>
static double time_it(F launch, int reps = 20)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; i++) launch();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < reps; i++) launch();
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipGetLastError());
    return ms / reps * 1e-3;
}

int main()
{
    const long nv = 68157440;  // ~1.09 GB of double2
    const size_t bytes = nv * 16;
    d2 *a, *b;
    double *sink;
    CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes)); CK(hipMalloc(&sink, 64));
    hipLaunchKernelGGL(w_data, dim3(4096), dim3(256), 0, 0, a, nv);
    CK(hipDeviceSynchronize());
    const d2 one = {1.0, 2.0}, zero = {0.0, 0.0};
    printf("buffer %.1f MB\n", bytes / 1e6);
    for (int blocks : {256, 512, 1024, 2048, 4096, 8192, 16384}) {
        double t1 = time_it([&] { hipLaunchKernelGGL(w_stride<false>, dim3(blocks), dim3(256), 0, 0, b, nv, one); });
        double t2 = time_it([&] { hipLaunchKernelGGL(w_stride<true>, dim3(blocks), dim3(256), 0, 0, b, nv, one); });
        double t3 = time_it([&] { hipLaunchKernelGGL(w_chunk<false>, dim3(blocks), dim3(256), 0, 0, b, nv, one); });
        double t4 = time_it([&] { hipLaunchKernelGGL(w_chunk<true>, dim3(blocks), dim3(256), 0, 0, b, nv, one); });
        double t5 = time_it([&] { hipLaunchKernelGGL(w_wavechunk, dim3(blocks), dim3(256), 0, 0, b, nv, one); });
        double t6 = time_it([&] { hipLaunchKernelGGL(w_wide, dim3(blocks), dim3(256), 0, 0, b, nv, one); });
        double t7 = time_it([&] { hipLaunchKernelGGL(w_stride<false>, dim3(blocks), dim3(256), 0, 0, b, nv, zero); });
        double t8 = time_it([&] { hipLaunchKernelGGL(w_data, dim3(blocks), dim3(256), 0, 0, b, nv); });
        printf("write blocks=%5d: stride %.0f | stride-nt %.0f | chunk %.0f | chunk-nt %.0f | wavechunk %.0f | wide64B %.0f | zeros %.0f | random-data %.0f GB/s\n",
               blocks, bytes / t1 / 1e9, bytes / t2 / 1e9, bytes / t3 / 1e9, bytes / t4 / 1e9, bytes / t5 / 1e9, bytes / t6 / 1e9, bytes / t7 / 1e9, bytes / t8 / 1e9);
        fflush(stdout);
    }
    {
        double t = time_it([&] { CK(hipMemsetAsync(b, 0, bytes, 0)); });
        printf("hipMemsetAsync(0)      : %.0f GB/s\n", bytes / t / 1e9);
        t = time_it([&] { CK(hipMemsetD32Async((hipDeviceptr_t)b, 0x3ff12345, bytes / 4, 0)); });
        printf("hipMemsetD32Async(pat) : %.0f GB/s\n", bytes / t / 1e9);
    }
    for (int blocks : {512, 1024, 2048, 4096, 8192}) {
        double t = time_it([&] { hipLaunchKernelGGL(r_stride, dim3(blocks), dim3(256), 0, 0, a, sink, nv); });
        double t2 = time_it([&] { hipLaunchKernelGGL(c_chunk<false>, dim3(blocks), dim3(256), 0, 0, a, b, nv); });
        double t3 = time_it([&] { hipLaunchKernelGGL(c_chunk<true>, dim3(blocks), dim3(256), 0, 0, a, b, nv); });
        printf("blocks=%5d: read %.0f GB/s | copy chunk8 %.0f | copy chunk8 nt %.0f GB/s (r+w)\n", blocks, bytes / t / 1e9, 2.0 * bytes / t2 / 1e9, 2.0 * bytes / t3 / 1e9);
    }
    {
        double t = time_it([&] { CK(hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, 0)); });
        printf("hipMemcpy D2D (random data): %.0f GB/s (r+w)\n", 2.0 * bytes / t / 1e9);
    }
    return 0;
}
