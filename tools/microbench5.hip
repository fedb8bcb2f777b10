// microbench5.hip — does the 256 MB Infinity Cache serve the START of a sweep when the sweep direction alternates?  (tool)
// A time loop ping-pongs two 1 GiB fields: sweep n writes B front to back, sweep n+1 reads B.  Read in the SAME direction the
// first bytes sweep n+1 wants were written longest ago (evicted); read in the OPPOSITE direction they are the most recent
// ones.  Copy kernels in the simplest streaming order (one 16-byte vector per thread), plain / non-temporal loads and stores.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/microbench5 tools/microbench5.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CK(x)                                                                                                        \
    do {                                                                                                             \
        hipError_t e = (x);                                                                                          \
        if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } \
    } while (0)

typedef float f4 __attribute__((ext_vector_type(4)));

template <bool NTL, bool NTS> __global__ void __launch_bounds__(256) copy_dir(const f4 *in, f4 *out, long nblocks, int reverse)
{
    const long b = reverse ? nblocks - 1 - (long)blockIdx.x : (long)blockIdx.x;
    const long i = b * 256 + threadIdx.x;
    const f4 v = NTL ? __builtin_nontemporal_load(in + i) : in[i];
    if (NTS) __builtin_nontemporal_store(v, out + i);
    else out[i] = v;
}

template <bool NTL, bool NTS> static double run(f4 *a, f4 *b, long n, bool alternate, int reps)
{
    const long nblocks = n / 256;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    auto sweep = [&](int s) {
        const int rev = alternate ? (s & 1) : 0;
        if (s & 1) hipLaunchKernelGGL((copy_dir<NTL, NTS>), dim3(nblocks), dim3(256), 0, 0, b, a, nblocks, rev);
        else hipLaunchKernelGGL((copy_dir<NTL, NTS>), dim3(nblocks), dim3(256), 0, 0, a, b, nblocks, rev);
    };
    for (int s = 0; s < 4; s++) sweep(s);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    for (int s = 0; s < reps; s++) sweep(s);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipGetLastError());
    return ms / reps * 1e-3;
}

int main()
{
    for (size_t mib : {256, 512, 1024}) {
        const size_t bytes = mib << 20;
        const long n = bytes / 16;
        f4 *a, *b;
        CK(hipMalloc(&a, bytes));
        CK(hipMalloc(&b, bytes));
        CK(hipMemset(a, 0x11, bytes));
        CK(hipMemset(b, 0x22, bytes));
        auto r = [&](double t) { return 2.0 * bytes / t / 1e9; };
        printf("== ping-pong copies of %zu MiB fields, GB/s (read + write): same direction | alternating direction\n", mib);
        printf("plain load, plain store : %.0f | %.0f\n", r(run<false, false>(a, b, n, false, 20)), r(run<false, false>(a, b, n, true, 20)));
        printf("plain load, nt store    : %.0f | %.0f\n", r(run<false, true>(a, b, n, false, 20)), r(run<false, true>(a, b, n, true, 20)));
        printf("nt load,    plain store : %.0f | %.0f\n", r(run<true, false>(a, b, n, false, 20)), r(run<true, false>(a, b, n, true, 20)));
        printf("nt load,    nt store    : %.0f | %.0f\n", r(run<true, true>(a, b, n, false, 20)), r(run<true, true>(a, b, n, true, 20)));
        fflush(stdout);
        CK(hipFree(a));
        CK(hipFree(b));
    }
    return 0;
}
