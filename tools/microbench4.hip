// microbench4.hip — calibration of the copy ceiling on MI355X (tool, not product; VERDICT r2 "next" #5).
// Question: MI355X_MICROARCH.md quotes 6.29 TB/s for a float4 copy; round 1 measured 5.2-5.6 TB/s with per-block chunked copies and
// hipMemcpyDtoD.  Which access ORDER reaches the guide's figure?  Variants:
//   simple   : one 16-byte vector per thread, grid = n / 256 (workgroups in flight touch one contiguous window)
//   stride U : persistent grid-stride loop with U independent vectors in flight per thread (window = grid x U x 4 KiB)
//   chunk  U : every workgroup owns one contiguous chunk (round 1's pattern; many distant windows)
//   march    : the stencil kernels' order: a wave owns R rows of 4 KiB and marches through planes 2 MiB apart
// each with plain and non-temporal accesses, for buffers of 128 MiB .. 1 GiB (a 256 MiB Infinity Cache sits in front of HBM).
// build: hipcc --offload-arch=gfx950 -O3 -o tools/microbench4 tools/microbench4.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CK(x)                                                                                                        \
    do {                                                                                                             \
        hipError_t e = (x);                                                                                          \
        if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } \
    } while (0)

typedef float f4 __attribute__((ext_vector_type(4)));

template <bool NT> __device__ __forceinline__ f4 ld(const f4 *p) { return NT ? __builtin_nontemporal_load(p) : *p; }
template <bool NT> __device__ __forceinline__ void st(f4 *p, f4 v)
{
    if (NT) __builtin_nontemporal_store(v, p);
    else *p = v;
}

template <bool NT> __global__ void __launch_bounds__(256) c_simple(const f4 *in, f4 *out, long n)
{
    const long i = blockIdx.x * 256L + threadIdx.x;
    if (i < n) st<NT>(out + i, ld<NT>(in + i));
}

template <int U, bool NT> __global__ void __launch_bounds__(256) c_stride(const f4 *in, f4 *out, long n)
{
    const long stride = (long)gridDim.x * 256 * U;
    for (long i = blockIdx.x * 256L * U + threadIdx.x; i + (U - 1) * 256 < n; i += stride) {
        f4 v[U];
#pragma unroll
        for (int u = 0; u < U; u++) v[u] = ld<NT>(in + i + u * 256);
#pragma unroll
        for (int u = 0; u < U; u++) st<NT>(out + i + u * 256, v[u]);
    }
}

template <int U, bool NT> __global__ void __launch_bounds__(256) c_chunk(const f4 *in, f4 *out, long n)
{
    const long per = n / gridDim.x;
    const long b0 = blockIdx.x * per, b1 = b0 + per;
    for (long i = b0 + threadIdx.x; i + (U - 1) * 256 < b1; i += 256 * U) {
        f4 v[U];
#pragma unroll
        for (int u = 0; u < U; u++) v[u] = ld<NT>(in + i + u * 256);
#pragma unroll
        for (int u = 0; u < U; u++) st<NT>(out + i + u * 256, v[u]);
    }
}

// stencil order: the buffer is [planes][rows][256 vectors] (a row = 4 KiB, like a 512-cell fp64 row); a WAVE owns R rows and
// marches through `planes_per` planes; wave tiles are numbered rows-fastest, then x-chunks (like the product's xcd_swizzle-free map)
template <int R, bool NT> __global__ void __launch_bounds__(64) c_march(const f4 *in, f4 *out, int planes, int rows, int planes_per)
{
    const int tiles_per_plane = rows / R;
    const int tile = blockIdx.x % tiles_per_plane, chunk = blockIdx.x / tiles_per_plane;
    const long row_v = 256;
    const long plane_v = (long)rows * row_v;
    const int p0 = chunk * planes_per;
    for (int p = p0; p < p0 + planes_per && p < planes; p++) {
        f4 v[R][4];
#pragma unroll
        for (int r = 0; r < R; r++)
#pragma unroll
            for (int c = 0; c < 4; c++) v[r][c] = ld<false>(in + p * plane_v + (tile * R + r) * row_v + c * 64 + threadIdx.x);
#pragma unroll
        for (int r = 0; r < R; r++)
#pragma unroll
            for (int c = 0; c < 4; c++) st<NT>(out + p * plane_v + (tile * R + r) * row_v + c * 64 + threadIdx.x, v[r][c]);
    }
}

__global__ void __launch_bounds__(256) fill(f4 *out, long n)
{
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        float x = (float)((i * 2654435761u) % 1000) * 1e-3f;
        out[i] = f4{x, x + 1, x + 2, x + 3};
    }
}

template <typename F> static double time_it(F launch, int reps)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; i++) launch();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < reps; i++) launch();
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipGetLastError());
    CK(hipEventDestroy(e0));
    CK(hipEventDestroy(e1));
    return ms / reps * 1e-3;
}

int main()
{
    const size_t max_bytes = 1ull << 30;
    f4 *a, *b;
    CK(hipMalloc(&a, max_bytes));
    CK(hipMalloc(&b, max_bytes));
    hipLaunchKernelGGL(fill, dim3(4096), dim3(256), 0, 0, a, (long)(max_bytes / 16));
    CK(hipDeviceSynchronize());
    printf("copy rates in GB/s (read + write bytes / time)\n");
    for (size_t mib : {128, 512, 1024}) {
        const size_t bytes = mib << 20;
        const long n = bytes / 16;
        const int reps = mib >= 1024 ? 20 : 40;
        auto rate = [&](double t) { return 2.0 * bytes / t / 1e9; };
        printf("== buffer %zu MiB (src) + %zu MiB (dst)\n", mib, mib);
        printf("simple            : plain %.0f | nt %.0f\n",
               rate(time_it([&] { hipLaunchKernelGGL(c_simple<false>, dim3(n / 256), dim3(256), 0, 0, a, b, n); }, reps)),
               rate(time_it([&] { hipLaunchKernelGGL(c_simple<true>, dim3(n / 256), dim3(256), 0, 0, a, b, n); }, reps)));
        for (int blocks : {256, 512, 1024, 2048, 4096, 8192}) {
            printf("stride blocks=%5d: U1 %.0f | U2 %.0f | U4 %.0f | U8 %.0f | U4nt %.0f | U8nt %.0f    chunk: U4 %.0f | U8 %.0f | U8nt %.0f\n", blocks,
                   rate(time_it([&] { hipLaunchKernelGGL((c_stride<1, false>), dim3(blocks), dim3(256), 0, 0, a, b, n); }, reps)),
                   rate(time_it([&] { hipLaunchKernelGGL((c_stride<2, false>), dim3(blocks), dim3(256), 0, 0, a, b, n); }, reps)),
                   rate(time_it([&] { hipLaunchKernelGGL((c_stride<4, false>), dim3(blocks), dim3(256), 0, 0, a, b, n); }, reps)),
                   rate(time_it([&] { hipLaunchKernelGGL((c_stride<8, false>), dim3(blocks), dim3(256), 0, 0, a, b, n); }, reps)),
                   rate(time_it([&] { hipLaunchKernelGGL((c_stride<4, true>), dim3(blocks), dim3(256), 0, 0, a, b, n); }, reps)),
                   rate(time_it([&] { hipLaunchKernelGGL((c_stride<8, true>), dim3(blocks), dim3(256), 0, 0, a, b, n); }, reps)),
                   rate(time_it([&] { hipLaunchKernelGGL((c_chunk<4, false>), dim3(blocks), dim3(256), 0, 0, a, b, n); }, reps)),
                   rate(time_it([&] { hipLaunchKernelGGL((c_chunk<8, false>), dim3(blocks), dim3(256), 0, 0, a, b, n); }, reps)),
                   rate(time_it([&] { hipLaunchKernelGGL((c_chunk<8, true>), dim3(blocks), dim3(256), 0, 0, a, b, n); }, reps)));
            fflush(stdout);
        }
        // stencil order on the same bytes: rows of 4 KiB, 512 rows per plane (2 MiB planes)
        const int rows = 512, planes = (int)(bytes / (512 * 4096));
        for (int per : {planes / 8, planes / 4, planes / 2, planes}) {
            if (per < 1) continue;
            const int chunks = (planes + per - 1) / per;
            printf("march planes/wave=%4d: R2 %.0f | R2nt %.0f | R4 %.0f | R4nt %.0f   (wave tiles: %d / %d)\n", per,
                   rate(time_it([&] { hipLaunchKernelGGL((c_march<2, false>), dim3(rows / 2 * chunks), dim3(64), 0, 0, a, b, planes, rows, per); }, reps)),
                   rate(time_it([&] { hipLaunchKernelGGL((c_march<2, true>), dim3(rows / 2 * chunks), dim3(64), 0, 0, a, b, planes, rows, per); }, reps)),
                   rate(time_it([&] { hipLaunchKernelGGL((c_march<4, false>), dim3(rows / 4 * chunks), dim3(64), 0, 0, a, b, planes, rows, per); }, reps)),
                   rate(time_it([&] { hipLaunchKernelGGL((c_march<4, true>), dim3(rows / 4 * chunks), dim3(64), 0, 0, a, b, planes, rows, per); }, reps)),
                   rows / 2 * chunks, rows / 4 * chunks);
            fflush(stdout);
        }
        printf("hipMemcpyDtoD     : %.0f\n", rate(time_it([&] { CK(hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, 0)); }, reps)));
    }
    return 0;
}
