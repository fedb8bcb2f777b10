// tools/e2_bench.hip - tile-shape experiments for the two-level kernel (pdehip_march2.inc) outside the library: the body is
// instantiated for several (cells per lane, rows per tile, waves per SIMD) and timed on a periodic n^3 fp64 grid with unit
// spacing; every variant's output is compared bit for bit with the library's shape (2 cells per lane, 4 rows, 2 waves).
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Ipy-pde_amd/csrc tools/e2_bench.hip -o tools/e2_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "pdehip_device.h"
namespace pdehip {
#include "pdehip_march2.inc"
template <typename T, int VEC, int RY, int M2, bool RAGGED, bool NT, int WAVES, int NB>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WAVES))) e2v_kernel(LapArgs a)
{
    euler2_body<T, VEC, RY, M2, true, RAGGED, false, NT, NB>(a);
}
}
using namespace pdehip;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

struct Geo { long n, p1, p0, off, total; };

template <int VEC, int RY, bool NT, int WAVES, int NB = 3>
static double run(const char *name, const Geo &g, const double *in, double *out, long cap, int reps, int nwz_want)
{
    LapArgs a;
    memset(&a, 0, sizeof(a));
    constexpr long CW = 64 * VEC;
    a.in = in; a.out = out; a.y = in;
    a.n0 = a.n1 = a.n2 = g.n; a.p0 = g.p0; a.p1 = g.p1; a.off = g.off; a.o_off = g.off; a.o_s0 = g.p0; a.o_s1 = g.p1;
    a.sx = a.sy = a.sz = 1.0; a.s1 = 1.0; a.s2 = 0.1; a.ndim = 3; a.any_ibc = 1;
    for (int k = 0; k < 3; k++) a.per[k] = 1;
    a.ntz = (g.n + CW - 1) / CW; a.nty = (g.n + RY - 1) / RY;
    const long tiles = a.ntz * a.nty;
    long nxc = cap / tiles; if (nxc < 1) nxc = 1;
    const long lx = (g.n + nxc - 1) / nxc;
    a.lx = (int)lx; a.nxc = (g.n + lx - 1) / lx; a.xstride = lx;
    int nwz = nwz_want; while (a.ntz % nwz) nwz /= 2;
    a.nwy = 1; a.nblocks = a.nxc * tiles / nwz; a.no_swizzle = 0;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto launch = [&]() { hipLaunchKernelGGL((e2v_kernel<double, VEC, RY, E2_DIFFUSION_UNIT, false, NT, WAVES, NB>), dim3((unsigned)a.nblocks), dim3(64 * nwz), 0, 0, a); };
    launch(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    for (int r = 0; r < reps; r++) launch();
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double per = ms / reps, bytes = 16.0 * (double)g.n * g.n * g.n;
    printf("%-28s waves %ld (x-chunks %ld of %ld planes, %d waves per block)  %.4f ms per launch  %.3f TB/s  %.1f Gcell-steps/s\n", name, a.nxc * tiles, (long)a.nxc, lx, nwz, per,
           bytes / per * 1e-9, 2.0 * g.n * g.n * g.n / per * 1e-6);
    return per;
}

int main(int argc, char **argv)
{
    const long n = argc > 1 ? atol(argv[1]) : 512;
    const int reps = argc > 2 ? atoi(argv[2]) : 20;
    // argv[5]: elements added to the row pitch, argv[6]: elements added to the plane pitch (layout experiments)
    const long pad1 = argc > 5 ? atol(argv[5]) : 0, pad0 = argc > 6 ? atol(argv[6]) : 0;
    Geo g; g.n = n; g.p1 = (n + 2 + 1) / 2 * 2 + pad1; g.p0 = g.p1 * (n + 2) + pad0; g.off = g.p0 + g.p1 + 2; g.total = g.p0 * (n + 2) + 64;
    printf("row pitch %ld elements, plane pitch %ld elements (%ld B)\n", g.p1, g.p0, g.p0 * 8);
    std::vector<double> h((size_t)g.total);
    unsigned long long s = 88172645463325252ULL;
    for (auto &v : h) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; v = (double)(s >> 11) * (1.0 / 9007199254740992.0); }
    double *in, *ref, *out;
    // argv[3] = 1: physically contiguous allocations (hipExtMallocWithFlags, hipDeviceMallocContiguous) - the rate of the sweep depends on
    // how the arrays are backed (profiles/r03_timing_modes.log)
    const bool contig = argc > 3 && atoi(argv[3]) == 1;
    auto alloc = [&](double **p) { return contig ? hipExtMallocWithFlags((void **)p, g.total * 8, hipDeviceMallocContiguous) : hipMalloc(p, g.total * 8); };
    CK(alloc(&in)); CK(alloc(&ref)); CK(alloc(&out));
    printf("allocations: %s  in %p  ref %p  out %p\n", contig ? "contiguous" : "default", (void *)in, (void *)ref, (void *)out);
    CK(hipMemcpy(in, h.data(), g.total * 8, hipMemcpyHostToDevice));
    CK(hipMemset(ref, 0, g.total * 8));
    std::vector<double> href((size_t)g.total), hout((size_t)g.total);
    auto check = [&](const char *name) {
        CK(hipMemcpy(hout.data(), out, g.total * 8, hipMemcpyDeviceToHost));
        long bad = 0;
        for (long i = 0; i < n; i++) for (long j = 0; j < n; j++) {
            const long o = g.off + i * g.p0 + j * g.p1;
            bad += memcmp(&hout[o], &href[o], n * 8) != 0;
        }
        if (bad) printf("  !! %s: %ld rows differ from the library's tile\n", name, bad);
        CK(hipMemset(out, 0, g.total * 8));
    };
    for (int round = 0; round < 2; round++) {
        run<2, 4, true, 2>("2 cells x 4 rows, 2 waves", g, in, ref, 2048, reps, 4);
        if (round == 0) CK(hipMemcpy(href.data(), ref, g.total * 8, hipMemcpyDeviceToHost));
        run<2, 4, false, 2>("same, plain stores", g, in, out, 2048, reps, 4); check("plain");
        if (argc > 4) continue;   // argv[4]: only the library's tile
        run<2, 2, true, 2>("2 cells x 2 rows, 2 waves", g, in, out, 2048, reps, 4); check("2x2w2");
        run<2, 2, true, 2, 4>("2x2, 2 waves, 4 buffers", g, in, out, 2048, reps, 4); check("2x2w2b4");
        run<2, 2, false, 2, 4>("2x2, 2w, 4 buf, plain st", g, in, out, 2048, reps, 4); check("2x2w2b4p");
        run<2, 2, true, 2, 4>("2x2, 2w, 4 buf, 4096 waves", g, in, out, 4096, reps, 4); check("2x2w2b4x");
        run<1, 4, true, 3, 4>("1x4, 3 waves, 4 buffers", g, in, out, 3072, reps, 4); check("1x4w3b4");
        run<1, 4, true, 2, 4>("1x4, 2 waves, 4 buffers", g, in, out, 2048, reps, 4); check("1x4w2b4");
        run<1, 2, true, 4, 4>("1x2, 4 waves, 4 buffers", g, in, out, 4096, reps, 4); check("1x2w4b4");
        run<2, 4, true, 2, 4>("2x4, 2 waves, 4 buffers", g, in, out, 2048, reps, 4); check("2x4w2b4");
        run<2, 4, true, 1, 4>("2x4, 1 wave, 4 buffers", g, in, out, 1024, reps, 4); check("2x4w1b4");
        run<2, 8, true, 1, 3>("2x8, 1 wave, 3 buffers", g, in, out, 1024, reps, 4); check("2x8w1b3");
        run<2, 8, true, 1, 4>("2x8, 1 wave, 4 buffers", g, in, out, 1024, reps, 4); check("2x8w1b4");
        run<2, 8, false, 1, 4>("2x8, 1w, 4 buf, plain st", g, in, out, 1024, reps, 4); check("2x8w1b4p");
        run<2, 8, true, 1, 4>("2x8, 1w, 4 buf, 2048 waves", g, in, out, 2048, reps, 4); check("2x8w1b4x");
    }
    return 0;
}
