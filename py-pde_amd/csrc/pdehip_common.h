// pdehip_common.h — internal helpers shared by the translation units of libpdehip.so
// (gfx950 only; no CUDA-compat paths).
#pragma once

#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "../../include/pdehip.h"
#include "pdehip_device.h"
#include "pdehip_slab_loops.h"

namespace pdehip {

// ---- error plumbing: int status + thread-local message (SURVEY.md §8b "error convention") --
void set_error(const std::string &msg);
#define PDEHIP_FAIL(code, ...)                         \
    do {                                               \
        char _buf[512];                                \
        snprintf(_buf, sizeof(_buf), __VA_ARGS__);     \
        ::pdehip::set_error(_buf);                     \
        return (code);                                 \
    } while (0)
#define PDEHIP_HIP(expr)                                                                     \
    do {                                                                                     \
        hipError_t _e = (expr);                                                              \
        if (_e != hipSuccess)                                                                \
            PDEHIP_FAIL(100 + (int)_e, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), \
                        __FILE__, __LINE__);                                                 \
    } while (0)
#define PDEHIP_TRY(expr)          \
    do {                          \
        int _rc = (expr);         \
        if (_rc != 0) return _rc; \
    } while (0)

// error codes (mapped to Python exceptions by pde_hip/_lib.py)
enum { E_OK = 0, E_VALUE = 1, E_NOTIMPL = 2, E_RUNTIME = 3 };

// ---- device layout of a "full" array ---------------------------------------------------------
// A grid is normalised to three axes (an n-D grid occupies the trailing n axes; padded axes
// have n = 1 and no ghost layer).  The device layout differs from the reference's host layout
// (shape + 2, compact) in ONE respect: every row of the fastest axis is shifted and padded so
// that the first interior cell of each row sits on a 128-byte line and the pitch is a multiple of
// 128 bytes.  Every interior vector (double2 / float4) is an aligned dwordx4 access, and - round 5 - the
// 1 KiB a wave moves per row piece covers exactly 8 cache lines: with the 16-byte alignment of rounds 1-4
// (pitch n2 + 4) every such piece straddled 9 lines and the pieces of neighbouring waves / rows shared a
// line, i.e. partial-line writes and double fetches at every seam.  Measured in tools/microbench6.hip
// (profiles/r05_microbench6_row_alignment.md): copies of the interior rows 5.0 -> 5.9-6.0 TB/s, chunk-tiled
// stencil marches +10-25 %.  PDEHIP_ROW_ALIGN=<bytes> (16 ... 256, A/B aid) selects the alignment.
//   row:  [pad .. pad, ghost, interior(n2) ..., ghost, pad ..]     interior starts at `lpad`
struct NGrid {
    int ndim;
    int dtype;
    long n[3];         // valid cells
    long gh[3];        // ghost width (1 on used axes, 0 on padded axes)
    long p[3];         // pitches in elements (p[2] == 1)
    long pc;           // elements of one component
    long off;          // offset of interior cell (0,0,0)
    long lpad;         // column of the first interior cell in a row
    double dx[3];
    double lap_scale[3];  // dx**-2, numpy semantics (pow(dx, -2))
};

inline long elem_size(int dtype) { return dtype == PDEHIP_F64 ? 8 : 4; }

inline long row_align_bytes()
{
    static const long bytes = [] {
        const char *e = getenv("PDEHIP_ROW_ALIGN");
        const long v = e ? atol(e) : 128;
        return (v == 16 || v == 32 || v == 64 || v == 128 || v == 256) ? v : 128L;
    }();
    return bytes;
}

inline int norm_grid(const pdehip_grid_t *g, NGrid *n)
{
    if (!g) PDEHIP_FAIL(E_VALUE, "grid descriptor is NULL");
    if (g->ndim < 1 || g->ndim > 3) PDEHIP_FAIL(E_NOTIMPL, "unsupported number of axes %d", g->ndim);
    if (g->dtype != PDEHIP_F64 && g->dtype != PDEHIP_F32) PDEHIP_FAIL(E_NOTIMPL, "unsupported dtype code %d", g->dtype);
    n->ndim = g->ndim;
    n->dtype = g->dtype;
    for (int ax = 0; ax < 3; ax++) { n->n[ax] = 1; n->gh[ax] = 0; n->dx[ax] = 1; n->lap_scale[ax] = 1; }
    for (int a = 0; a < g->ndim; a++) {
        int ax = 3 - g->ndim + a;
        if (g->shape[a] < 1) PDEHIP_FAIL(E_VALUE, "grid shape must be positive (axis %d: %ld)", a, (long)g->shape[a]);
        n->n[ax] = g->shape[a];
        n->gh[ax] = 1;
        n->dx[ax] = g->dx[a];
        n->lap_scale[ax] = std::pow(g->dx[a], -2.0);
    }
    n->lpad = row_align_bytes() / elem_size(g->dtype);
    n->p[2] = 1;
    // pitch: a multiple of the alignment.  "tight" (PDEHIP_ROW_PITCH=tight, A/B aid): n2 + 2 rounded up - the upper ghost cell of a row may
    // then be element 0 of the next row's segment (its padding), 3 % fewer bytes of footprint at 512 cells per row
    static const bool tight = getenv("PDEHIP_ROW_PITCH") && !strcmp(getenv("PDEHIP_ROW_PITCH"), "tight");
    n->p[1] = tight && n->lpad >= 4 ? ((n->n[2] + 2 + n->lpad - 1) / n->lpad) * n->lpad : ((n->lpad + n->n[2] + 1 + n->lpad - 1) / n->lpad) * n->lpad;
    n->p[0] = n->p[1] * (n->n[1] + 2 * n->gh[1]);
    n->pc = n->p[0] * (n->n[0] + 2 * n->gh[0]);
    n->off = n->gh[0] * n->p[0] + n->gh[1] * n->p[1] + n->lpad;
    return 0;
}

// slack elements appended to every allocation so vector loads of clamped lanes stay in bounds
constexpr long kAllocSlack = 2048;

struct OutStr { long off, s0, s1, sc; };
inline OutStr out_strides(const NGrid &n, int layout)
{
    OutStr o;
    if (layout == PDEHIP_OUT_FULL) { o.off = n.off; o.s0 = n.p[0]; o.s1 = n.p[1]; o.sc = n.pc; }
    else { o.off = 0; o.s1 = n.n[2]; o.s0 = n.n[1] * n.n[2]; o.sc = n.n[0] * n.n[1] * n.n[2]; }
    return o;
}

inline hipStream_t as_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }

// kernel launchers implemented in pdehip_kernels.hip
// input-side BCs the stencil kernel evaluates on the fly (scalar first-order conditions)
struct InputBCs {
    int on[3][2];      // [normalised axis][lower, upper]
    long idx[3][2];
    double c[3][2], f[3][2];
};
// periodic (1) / local (0) / not covered (-1) classification of the two faces of one axis
inline int classify_axis(const InputBCs &fg, int ax, long n)
{
    const bool on = fg.on[ax][0] && fg.on[ax][1];
    const bool per = on && fg.idx[ax][0] == n - 1 && fg.idx[ax][1] == 0 && fg.c[ax][0] == 0 && fg.c[ax][1] == 0 &&
                     fg.f[ax][0] == 1 && fg.f[ax][1] == 1;
    const bool loc = on && fg.idx[ax][0] == 0 && fg.idx[ax][1] == n - 1;
    return per ? 1 : (loc ? 0 : -1);
}
// every array of a Runge-Kutta stage epilogue on a 16-byte boundary (the vector accesses of the stage sweeps)
inline bool stage_aligned(const LapArgs &a)
{
    uintptr_t bits = (uintptr_t)a.st_y | (uintptr_t)a.st_out;
    for (int m = 0; m < 5; m++) bits |= (uintptr_t)a.st_k[m];
    return bits % 16 == 0;
}
void note_kernel(const char *fmt, ...);   // pdehip_runtime.hip: name of the kernel instance launched last (pdehip_last_kernel_name)
// StageFuse (what follows a slope k = dt*rhs in a Runge-Kutta scheme, fused into the sweep that computes k): pdehip_slab_loops.h
// launch configuration of the two-level kernel handed to a caller that launches a run-time compiled instance itself
struct Euler2Plan {
    LapArgs a;
    unsigned grid, block;
    int ry;
    bool has_y;
};
int preload_shell_kernels();     // pdehip_shell.hip: the same for its code object
#include "pdehip_launchers.h"
namespace exactv {
#include "pdehip_launchers.h"
}
namespace fastv {
#include "pdehip_launchers.h"
}
bool fastmath_on();   // pdehip_dispatch.hip: pdehip_set_fastmath (process-wide; default off)
int euler_multi_2d(const pdehip_grid_t *g, const pdehip_rhs_t *rhs, const void *in, void *out, double dt, int nsteps, void *stream,
                   bool *done);
// two Euler steps of the diffusion equation in one sweep, BCs of both levels on the fly; *done = false
// (nothing launched) when the grid / faces are not covered by the kernel (see pdehip_march2.inc)
// two Euler steps per sweep with faces given as coefficient arrays (pdehip_shell.hip) and the two coefficient sets of a program of
// expression conditions (pdehip_jit.hip)
int euler2_timed_faces(const pdehip_grid_t *g, const void *in, void *out, double s1, double s2, const pdehip_bc_face_t *faces,
                       void *bc_program, void *stream, bool *done);
int shell_open_rows(const NGrid &n, const LapArgs &a, int columns, int rows, hipStream_t st);   // the columns / rows behind the tiles of a two-step sweep over open rows
bool bcprog_reads(void *handle);
bool bcprog_second_set(void *handle, const double *const_arr, const double **c2, const double **f2);
int bcprog_run_pair(void *handle, double t0, double t1, void *stream);
int euler2_with_input_bcs(const pdehip_grid_t *g, const void *in, void *out, double s1, double s2,
                          const pdehip_bc_face_t *faces, void *stream, bool *done, int xplain = 0, bool dry_run = false, int ends = 0);
// two Euler steps of the diffusion equation on a box of a larger array with two real halo layers on the cut axes (pdehip_ops.hip)
int euler2_box(const pdehip_grid_t *g_box, const pdehip_bc_face_t *faces, const int *cut3, const void *in_ext, void *out_ext, double s1,
               double s2, void *stream, bool *done, bool dry_run = false, const long *lo3 = nullptr, const long *n3 = nullptr);
// one Cahn-Hilliard sweep: mu = c^3 - c - gamma*lap(c) with the faces of c, then (euler) out = c + dt*lap(mu) or
// (!euler) out = dt*lap(mu) with the faces of mu — mu never leaves the registers; *done as above
int cahn_hilliard_fused(const pdehip_grid_t *g, const void *in, void *out, double gamma, double dt, bool euler,
                        const pdehip_bc_face_t *faces_c, const pdehip_bc_face_t *faces_mu, void *stream, bool *done,
                        int xplain = 0, bool dry_run = false, const StageFuse *stage = nullptr);
// (stage: `out` = dt*lap(mu) is followed by the Runge-Kutta combination of StageFuse in the same sweep; `euler` must be false)
// xplain: 1 = the slowest axis has two real halo layers on either side (slab decomposition) instead of BCs; 2 = on its upper
// side only (the lower face keeps its local BC: first slab of a non-periodic axis); 3 = on its lower side only (last slab); `in` / `out` then
// point one layer before the first layer to update, like every sub-slab launch (pdehip_comm.hip)
// BCs of `in` (on the fly where possible) + stencil (mode LAP_*) into the FULL array `out`
// (see pdehip_ops.hip)
int laplace_with_input_bcs(const pdehip_grid_t *g, void *in, const void *y, void *out, int mode, double s1,
                           double s2, double gamma, const pdehip_bc_face_t *in_faces, void *stream,
                           const StageFuse *stage = nullptr);

}  // namespace pdehip
