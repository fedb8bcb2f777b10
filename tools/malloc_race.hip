// Is a fresh allocation's zero fill (hipMalloc + hipMemset on the null stream) ordered before work on a NON-BLOCKING stream?
// Two uses of a fresh 64-byte buffer, as in SlabStepper.set_local / gather_local: (A) small upload from pageable memory into it,
// (B) a kernel writes it and the result is downloaded.  Each with and without hipStreamSynchronize(nullptr) after the fill.
// Counts how often the zero fill wins (the buffer reads back as zeros).   hipcc --offload-arch=gfx950 -O2 -o malloc_race malloc_race.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <cstdlib>

__global__ void fill_kernel(double *p, int n, double v) { int i = threadIdx.x; if (i < n) p[i] = v + i; }
__global__ void copy_kernel(const double *a, double *b, int n) { int i = threadIdx.x; if (i < n) b[i] = a[i]; }

int main(int argc, char **argv)
{
    const int reps = argc > 1 ? atoi(argv[1]) : 20000;
    hipStream_t s;
    hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    double *keep;
    hipMalloc(&keep, 64);
    double src[8], dst[8];
    for (int sync = 0; sync < 2; sync++) {
        long lostA = 0, lostB = 0;
        for (int r = 0; r < reps; r++) {
            for (int i = 0; i < 8; i++) src[i] = 1.0 + r + i;
            double *p;
            hipMalloc(&p, 64);
            hipMemset(p, 0, 64);
            if (sync) hipStreamSynchronize(nullptr);
            hipMemcpyAsync(p, src, 64, hipMemcpyHostToDevice, s);          // (A) upload into the fresh buffer
            copy_kernel<<<1, 64, 0, s>>>(p, keep, 8);
            hipMemcpyAsync(dst, keep, 64, hipMemcpyDeviceToHost, s);
            hipStreamSynchronize(s);
            if (memcmp(src, dst, 64)) lostA++;
            hipFree(p);
            hipMalloc(&p, 64);
            hipMemset(p, 0, 64);
            if (sync) hipStreamSynchronize(nullptr);
            fill_kernel<<<1, 64, 0, s>>>(p, 8, 1.0 + r);                   // (B) a kernel writes the fresh buffer
            hipMemcpyAsync(dst, p, 64, hipMemcpyDeviceToHost, s);
            hipStreamSynchronize(s);
            if (dst[0] != 1.0 + r) lostB++;
            hipFree(p);
        }
        printf("MALLOCRACE sync_after_fill=%d reps=%d upload_lost=%ld kernel_write_lost=%ld\n", sync, reps, lostA, lostB);
    }
    return 0;
}
