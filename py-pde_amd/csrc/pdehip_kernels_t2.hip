// pdehip_kernels_t2.hip - 2-D grids: K Euler steps per launch with the time levels in LDS (pdehip_tile2d.inc).  Split from pdehip_kernels.hip
// (round 6): its own translation unit.  Same compile flags (-ffp-contract=off).
#include "pdehip_common.h"

// Compiled TWICE (py-pde_amd/Makefile): as pdehip::exactv with -ffp-contract=off (bit-identical to the CPU oracle; the default) and, with
// -DPDEHIP_FAST_VARIANT -ffp-contract=fast, as pdehip::fastv (FMA contraction like numba's default fastmath, pde/backends/numba/utils.py:330-336;
// opt-in through pdehip_set_fastmath, results within 1e-10 of the exact build).  pdehip_dispatch.hip picks one per call.
#ifdef PDEHIP_FAST_VARIANT
#define PDEHIP_VARIANT_NS fastv
#else
#define PDEHIP_VARIANT_NS exactv
#endif
namespace pdehip {
namespace PDEHIP_VARIANT_NS {

#include "pdehip_tile2d.inc"

// K Euler steps of a 2-D grid per launch with the time levels in LDS (pdehip_tile2d.inc).  mode 0: diffusion (s1 = D), 1:
// Cahn-Hilliard (gamma; fm = faces of mu).  *done = false when grid / faces / step count are not covered.
constexpr int kTile2Halo = 8;
int tile2d_max_steps(int mode) { return (mode == 1 || mode == 4) ? kTile2Halo / 2 : kTile2Halo; }

// arguments and launch geometry of tile2d_kernel; mode 2 = the run-time built instance (pdehip_jit.hip fills par / scales)
int plan_tile2d(const NGrid &n, const void *in, void *out, int mode, double s1, double s2, double gamma, const InputBCs &fc,
                const InputBCs *fm, int nsteps, Tile2Args *pa, unsigned *nblocks, int *ptcw, bool *done)
{
    *done = false;
    if (n.ndim != 2 || in == out || nsteps < 1 || nsteps > tile2d_max_steps(mode) || force_generic_kernels()) return 0;
    if (n.n[1] >= (1L << 30) || n.n[2] >= (1L << 30)) return 0;   // 32-bit window arithmetic
    const bool two = mode == 1 || mode == 3 || mode == 4;   // a second table of conditions: mu (Cahn-Hilliard) / the second field (mode 3) / the temporary (mode 4)
    if (two && !fm) PDEHIP_FAIL(E_RUNTIME, "internal: tile sweep of two fields without the second table of conditions");
    Tile2Args &a = *pa;
    memset(&a, 0, sizeof(a));
    for (int k = 0; k < 2; k++) {
        const int ax = 1 + k;
        const int cls = classify_axis(fc, ax, n.n[ax]);
        if (cls < 0 || (two && classify_axis(*fm, ax, n.n[ax]) != cls)) return 0;
        a.per[k] = cls;
        for (int side = 0; side < 2; side++) {
            a.c[0][k][side] = fc.c[ax][side]; a.f[0][k][side] = fc.f[ax][side];
            if (two) { a.c[1][k][side] = fm->c[ax][side]; a.f[1][k][side] = fm->f[ax][side]; }
        }
    }
    a.in = in; a.out = out;
    a.n0 = n.n[1]; a.n1 = n.n[2]; a.p1 = n.p[1]; a.off = n.off; a.pc = n.pc;
    a.sx = n.lap_scale[1]; a.sy = n.lap_scale[2];
    a.s1 = s1; a.s2 = s2; a.gamma = gamma; a.nsteps = nsteps;
    for (int k = 0; k < 2; k++) {   // scales of the generated epilogue's inputs (as in jit_apply_impl)
        const double dx = n.dx[1 + k];
        a.gs[k] = 0.25 / (dx * dx); a.dd1[k] = 2 * dx; a.dd2[k] = 1 / (dx * dx); a.dg[k] = 0.5 / dx;
    }
    // tile 32 x 64 (halo redundancy 1.9 x at H = 8); grids that would give fewer than one workgroup per CU take 32 x 32 tiles
    // (2.25 x): a workgroup's K levels run one after the other on ONE CU, so spreading wins over redundancy there
    const long tiles64 = ((n.n[2] + 63) / 64) * ((n.n[1] + 31) / 32);
    const int tcw = tiles64 >= 256 ? 64 : 32;
    a.tiles1 = (int)((n.n[2] + tcw - 1) / tcw);
    *nblocks = (unsigned)(a.tiles1 * ((n.n[1] + 31) / 32));
    *ptcw = tcw;
    *done = true;
    return 0;
}

int launch_tile2d(const NGrid &n, const void *in, void *out, int mode, double s1, double s2, double gamma, const InputBCs &fc,
                  const InputBCs *fm, int nsteps, hipStream_t st, bool *done)
{
    Tile2Args a;
    unsigned nblocks = 0;
    int tcw = 0;
    PDEHIP_TRY(PDEHIP_VARIANT_NS::plan_tile2d(n, in, out, mode, s1, s2, gamma, fc, fm, nsteps, &a, &nblocks, &tcw, done));
    if (!*done) return 0;
    *done = false;
    const dim3 grid(nblocks), block(1024);
    note_kernel("tile2d_kernel<%s,mode=%d,32x%d tile> (%d steps per launch, time levels in LDS)", n.dtype == PDEHIP_F64 ? "double" : "float", mode, tcw, nsteps);
#define PDEHIP_T2(T, M)                                                                                            \
    do {                                                                                                           \
        if (tcw == 64) hipLaunchKernelGGL((tile2d_kernel<T, M, 32, 64, kTile2Halo>), grid, block, 0, st, a);       \
        else hipLaunchKernelGGL((tile2d_kernel<T, M, 32, 32, kTile2Halo>), grid, block, 0, st, a);                 \
    } while (0)
    if (n.dtype == PDEHIP_F64) {
        if (mode == 0) PDEHIP_T2(double, 0); else PDEHIP_T2(double, 1);
    } else {
        if (mode == 0) PDEHIP_T2(float, 0); else PDEHIP_T2(float, 1);
    }
#undef PDEHIP_T2
    PDEHIP_HIP(hipGetLastError());
    *done = true;
    return 0;
}

// (see preload_stencil_kernels, pdehip_kernels.hip)
int preload_t2_kernels()
{
    hipFuncAttributes attr;
    PDEHIP_HIP(hipFuncGetAttributes(&attr, reinterpret_cast<const void *>(&tile2d_kernel<double, 0, 32, 64, kTile2Halo>)));
    return 0;
}

}  // namespace PDEHIP_VARIANT_NS
}  // namespace pdehip
