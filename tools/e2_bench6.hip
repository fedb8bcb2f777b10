// tools/e2_bench6.hip - round 6: instruction-diet experiments for the two-step sweep (pdehip_march2.inc) outside the library, on the
// library's row layout (first interior cell of every row on a 128-byte line, pitch a multiple of 128 bytes).  Periodic n^3 fp64 grid, unit
// spacing; every variant's output is compared bit for bit with the library's tall tile.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Ipy-pde_amd/csrc tools/e2_bench6.hip -o tools/e2_bench6
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "pdehip_device.h"
namespace pdehip {
#include "pdehip_march2.inc"
template <typename T, int VEC, int RY, int M2, bool NT, int WAVES, int NB, bool PER3, int TWEAK>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WAVES, WAVES))) e2v_kernel(LapArgs a)
{
    euler2_body<T, VEC, RY, M2, true, false, false, NT, NB, PER3, TWEAK>(a);
}
}
using namespace pdehip;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

struct Geo { long n, p1, p0, off, total; };
static double *g_pong = nullptr;   // argv[4] = 1: launches alternate in -> out, out -> pong (the time loop's access pattern: the output of a sweep is the next input)

template <int RY, bool NT, int WAVES, int NB, bool PER3, int M2 = E2_DIFFUSION_UNIT, int TWEAK = 0>
static double run(const char *name, const Geo &g, const double *in, double *out, long cap, int reps, int nwz_want)
{
    LapArgs a;
    memset(&a, 0, sizeof(a));
    constexpr long CW = 128;
    a.in = in; a.out = out; a.y = in;
    a.n0 = a.n1 = a.n2 = g.n; a.p0 = g.p0; a.p1 = g.p1; a.off = g.off; a.o_off = g.off; a.o_s0 = g.p0; a.o_s1 = g.p1;
    a.sx = a.sy = a.sz = 1.0; a.s1 = 1.0; a.s2 = 0.1; a.ndim = 3; a.any_ibc = 1;
    for (int k = 0; k < 3; k++) {
        a.per[k] = 1;
        for (int side = 0; side < 2; side++) { a.ibc[k][side].on = 1; a.ibc[k][side].idx = side ? 0 : g.n - 1; a.ibc[k][side].c = 0; a.ibc[k][side].f = 1; a.ibc1[k][side] = a.ibc[k][side]; }
    }
    a.ntz = (g.n + CW - 1) / CW; a.nty = (g.n + RY - 1) / RY;
    const long tiles = a.ntz * a.nty;
    long nxc = cap / tiles; if (nxc < 1) nxc = 1;
    const long lx = (g.n + nxc - 1) / nxc;
    a.lx = (int)lx; a.nxc = (g.n + lx - 1) / lx; a.xstride = lx;
    int nwz = nwz_want; while (a.ntz % nwz) nwz /= 2;
    a.nwy = 1; a.nblocks = a.nxc * tiles / nwz; a.no_swizzle = 0;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    int flip = 0;
    auto launch = [&]() { if (g_pong) { a.in = flip ? out : in; a.y = a.in; a.out = flip ? g_pong : out; flip ^= 1; } hipLaunchKernelGGL((e2v_kernel<double, 2, RY, M2, NT, WAVES, NB, PER3, TWEAK>), dim3((unsigned)a.nblocks), dim3(64 * nwz), 0, 0, a); };
    launch(); CK(hipDeviceSynchronize()); flip = 0;
    CK(hipEventRecord(e0, 0));
    for (int r = 0; r < reps; r++) launch();
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double per = ms / reps, bytes = 16.0 * (double)g.n * g.n * g.n;
    printf("%-34s waves %5ld  %.4f ms per launch  %.3f TB/s = %.3f of 8  %.1f Gcell-steps/s\n", name, a.nxc * tiles, per, bytes / per * 1e-9, bytes / per * 1e-9 / 8.0,
           2.0 * g.n * g.n * g.n / per * 1e-6);
    return per;
}

int main(int argc, char **argv)
{
    const long n = argc > 1 ? atol(argv[1]) : 512;
    const int reps = argc > 2 ? atoi(argv[2]) : 20;
    const int rounds = argc > 3 ? atoi(argv[3]) : 3;
    Geo g; g.n = n;
    const long lpad = 16;
    g.p1 = (lpad + n + 1 + lpad - 1) / lpad * lpad; g.p0 = g.p1 * (n + 2); g.off = g.p0 + g.p1 + lpad; g.total = g.p0 * (n + 2) + 4096;
    printf("row pitch %ld elements, plane pitch %ld elements (%ld B)\n", g.p1, g.p0, g.p0 * 8);
    std::vector<double> h((size_t)g.total);
    unsigned long long s = 88172645463325252ULL;
    for (auto &v : h) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; v = (double)(s >> 11) * (1.0 / 9007199254740992.0); }
    double *in, *ref, *out;
    CK(hipMalloc(&in, g.total * 8)); CK(hipMalloc(&ref, g.total * 8)); CK(hipMalloc(&out, g.total * 8));
    CK(hipMemcpy(in, h.data(), g.total * 8, hipMemcpyHostToDevice));
    CK(hipMemset(ref, 0, g.total * 8));
    if (argc > 4 && atoi(argv[4]) == 1) { CK(hipMalloc(&g_pong, g.total * 8)); CK(hipMemset(g_pong, 0, g.total * 8)); }
    std::vector<double> href((size_t)g.total), hout((size_t)g.total);
    auto check = [&](const char *name) {
        CK(hipMemcpy(hout.data(), out, g.total * 8, hipMemcpyDeviceToHost));
        long bad = 0;
        for (long i = 0; i < n; i++) for (long j = 0; j < n; j++) {
            const long o = g.off + i * g.p0 + j * g.p1;
            bad += memcmp(&hout[o], &href[o], n * 8) != 0;
        }
        if (bad && !g_pong) printf("  !! %s: %ld rows differ from the library's tile\n", name, bad);
        CK(hipMemset(out, 0, g.total * 8));
    };
    for (int round = 0; round < rounds; round++) {
        run<8, true, 1, 4, false>("tall 2x8 1w 4buf (library)", g, in, ref, 1024, reps, 4);
        if (round == 0) CK(hipMemcpy(href.data(), ref, g.total * 8, hipMemcpyDeviceToHost));
        run<8, true, 1, 3, true>("2x8 1w 3buf, all-periodic", g, in, out, 1024, reps, 4); check("2x8 3buf per3");
        run<6, true, 1, 3, true>("2x6 1w 3buf, all-periodic", g, in, out, 1024, reps, 4); check("2x6 3buf per3");
        run<6, true, 1, 3, true, E2_DIFFUSION_UNIT, 1>("2x6 1w 3buf, per, late", g, in, out, 1024, reps, 4); check("2x6 3buf per3 late");
        run<6, true, 1, 4, true>("2x6 1w 4buf, all-periodic", g, in, out, 1024, reps, 4); check("2x6 4buf per3");
        run<6, true, 1, 3, true>("2x6 1w 3buf, per, 1376 waves", g, in, out, 1376, reps, 4); check("2x6 3buf per3 x");
        run<4, true, 2, 3, true, E2_DIFFUSION_UNIT, 1>("2x4 2w NT per, late", g, in, out, 2048, reps, 4); check("4-row per3 late");
    }
    return 0;
}
