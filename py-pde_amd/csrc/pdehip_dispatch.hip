// pdehip_dispatch.hip - one entry per launcher of the stencil kernels: the exact build (-ffp-contract=off: every rounding of the reference's
// expression order, bit-identical to the CPU oracle) or, after pdehip_set_fastmath(1), the same kernels compiled with FMA contraction.
// Reference: numba compiles py-pde's operators with fastmath {nsz, arcp, contract, afn, reassoc} by default (pde/backends/numba/utils.py:330-336,
// config `backend.numba.fastmath`, pde/backends/numba/config.py:20-26); here the default is the exact build and contraction is the opt-in.
#include <atomic>

#include "pdehip_common.h"

namespace pdehip {

static std::atomic<int> g_fastmath{0};
bool fastmath_on() { return g_fastmath.load(std::memory_order_relaxed) != 0; }

#define PDEHIP_PICK(call) (fastmath_on() ? fastv::call : exactv::call)

int launch_laplace(const NGrid &n, const void *in, void *out, const OutStr &o, int mode, double s1, double s2, double gamma, const void *y,
                   hipStream_t st, const InputBCs *fg, const StageFuse *stage)
{
    return PDEHIP_PICK(launch_laplace(n, in, out, o, mode, s1, s2, gamma, y, st, fg, stage));
}
int launch_deriv_march(const NGrid &n, const void *in, void *out, const OutStr &o, int mode, const double *gs, hipStream_t st, bool *done)
{
    return PDEHIP_PICK(launch_deriv_march(n, in, out, o, mode, gs, st, done));
}
int launch_div_march(const NGrid &n, int method, const void *in, void *out, const OutStr &o, hipStream_t st, bool *done)
{
    return PDEHIP_PICK(launch_div_march(n, method, in, out, o, st, done));
}
bool laplace_can_fuse_bcs(const NGrid &n, const void *in, const void *out, const void *y) { return exactv::laplace_can_fuse_bcs(n, in, out, y); }
int launch_ghosts(const NGrid &n, int ncomp, const pdehip_bc_face_t *faces, void *data, hipStream_t st)
{
    return PDEHIP_PICK(launch_ghosts(n, ncomp, faces, data, st));
}
int launch_euler2(const NGrid &n, const void *in, void *out, double s1, double s2, const InputBCs &fg, int xplain, hipStream_t st, bool *done,
                  bool dry_run, int ends, int m2, const InputBCs *fg1, double gamma, Euler2Plan *plan, const StageFuse *stage, int yzplain)
{
    return PDEHIP_PICK(launch_euler2(n, in, out, s1, s2, fg, xplain, st, done, dry_run, ends, m2, fg1, gamma, plan, stage, yzplain));
}
int tile2d_max_steps(int mode) { return exactv::tile2d_max_steps(mode); }
int plan_tile2d(const NGrid &n, const void *in, void *out, int mode, double s1, double s2, double gamma, const InputBCs &fc, const InputBCs *fm,
                int nsteps, Tile2Args *args, unsigned *nblocks, int *tile_columns, bool *done)
{
    return exactv::plan_tile2d(n, in, out, mode, s1, s2, gamma, fc, fm, nsteps, args, nblocks, tile_columns, done);   // (geometry only: no kernel)
}
int launch_tile2d(const NGrid &n, const void *in, void *out, int mode, double s1, double s2, double gamma, const InputBCs &fc, const InputBCs *fm,
                  int nsteps, hipStream_t st, bool *done)
{
    return PDEHIP_PICK(launch_tile2d(n, in, out, mode, s1, s2, gamma, fc, fm, nsteps, st, done));
}
bool force_generic_kernels() { return exactv::force_generic_kernels(); }
// the code objects of BOTH builds are loaded when the device is selected (see exactv::preload_stencil_kernels)
int preload_stencil_kernels() { PDEHIP_TRY(exactv::preload_stencil_kernels()); return fastv::preload_stencil_kernels(); }
int preload_e2_kernels() { PDEHIP_TRY(exactv::preload_e2_kernels()); return fastv::preload_e2_kernels(); }
int preload_t2_kernels() { PDEHIP_TRY(exactv::preload_t2_kernels()); return fastv::preload_t2_kernels(); }

}  // namespace pdehip

extern "C" {

int pdehip_set_fastmath(int on)
{
    pdehip::g_fastmath.store(on ? 1 : 0, std::memory_order_relaxed);
    return 0;
}

int pdehip_get_fastmath(int *on)
{
    using namespace pdehip;
    if (!on) PDEHIP_FAIL(E_VALUE, "get_fastmath: NULL pointer");
    *on = pdehip::fastmath_on() ? 1 : 0;
    return 0;
}

}  // extern "C"
