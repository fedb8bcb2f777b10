// pdehip_device.h — device-side definitions shared by the offline build (hipcc) and the run-time
// build (hiprtc, pdehip_jit.hip) of the register-pipelined stencil kernel.  No host headers.
#pragma once

namespace pdehip {

// epilogue selector of lap_march_kernel
enum { LAP_PLAIN = 0, LAP_SCALED = 1, LAP_EULER = 2, LAP_CH_MU = 3,
       LAP_GRAD_C = 4, LAP_GRAD_F = 5, LAP_GRAD_B = 6, LAP_GRADSQ_C = 7, LAP_GRADSQ_N = 8,
       LAP_CUSTOM = 9 /* epilogue generated at run time: pde_epilogue() */,
       LAP_STAGE = 10 /* Runge-Kutta stage: slope dt*(D*lap) AND the pointwise combination that follows it (LapArgs::st*) */ };

// the two fused levels of euler2_kernel (pdehip_march2.inc)
enum { E2_DIFFUSION = 0, E2_CH_EULER = 1, E2_CH_SCALED = 2, E2_CUSTOM = 3 /* run-time generated: pde_epilogue() twice */,
       E2_CUSTOM2 = 4 /* run-time generated: level 1 = pde_epilogue(), level 2 = pde_epilogue2() (two-pass expressions) */,
       E2_CH_STAGE = 5 /* E2_CH_SCALED + the Runge-Kutta combination that follows the slope (LapArgs::st*, like LAP_STAGE) */,
       E2_DIFFUSION_UNIT = 6 /* E2_DIFFUSION on a grid with dx = 1 on every axis and D = 1 (UnitGrid, the benchmark configuration):
                                the four multiplications by 1.0 per update are left out - x * 1.0 == x bit for bit */ };

// ---------------------------------------------------------------------------------------------
// small device helpers
// ---------------------------------------------------------------------------------------------
template <typename T, int VEC> struct VecT;
template <> struct VecT<double, 2> { typedef double type __attribute__((ext_vector_type(2))); };
template <> struct VecT<double, 1> { typedef double type __attribute__((ext_vector_type(1))); };   // narrow fp64 tiles of euler2_kernel (8 B per lane)
template <> struct VecT<float, 4> { typedef float type __attribute__((ext_vector_type(4))); };
template <> struct VecT<float, 2> { typedef float type __attribute__((ext_vector_type(2))); };   // narrow fp32 tiles of euler2_kernel (8 B per lane)

// streaming loads of the plane data of the march kernels.  PDEHIP_NT_LOADS (build-time A/B switch, tools/build_variant.sh):
// 1 = non-temporal loads, 2 = non-temporal loads and plain stores everywhere.  Measured on copy kernels (profiles/
// r03_microbench5_infinity_cache_direction.log): nt load + plain store 6.49 TB/s vs plain load + nt store 6.24 at 1 GiB.
#ifdef PDEHIP_NT_LOADS
#define PDEHIP_LDV(ptr) __builtin_nontemporal_load(ptr)
#else
#define PDEHIP_LDV(ptr) (*(ptr))
#endif

// wavefront shift by one lane through DPP (gfx9 wave_shr:1 / wave_shl:1).  Lane 0 (resp. lane
// 63) has no source lane and keeps `old`, which carries the value from outside the chunk.
__device__ __forceinline__ double wave_shr1(double old, double src)
{
    unsigned long long o = __double_as_longlong(old), s = __double_as_longlong(src);
    int lo = __builtin_amdgcn_update_dpp((int)(o & 0xffffffffu), (int)(s & 0xffffffffu), 0x138, 0xf, 0xf, false);
    int hi = __builtin_amdgcn_update_dpp((int)(o >> 32), (int)(s >> 32), 0x138, 0xf, 0xf, false);
    return __longlong_as_double(((unsigned long long)(unsigned)hi << 32) | (unsigned)lo);
}
__device__ __forceinline__ double wave_shl1(double old, double src)
{
    unsigned long long o = __double_as_longlong(old), s = __double_as_longlong(src);
    int lo = __builtin_amdgcn_update_dpp((int)(o & 0xffffffffu), (int)(s & 0xffffffffu), 0x130, 0xf, 0xf, false);
    int hi = __builtin_amdgcn_update_dpp((int)(o >> 32), (int)(s >> 32), 0x130, 0xf, 0xf, false);
    return __longlong_as_double(((unsigned long long)(unsigned)hi << 32) | (unsigned)lo);
}
__device__ __forceinline__ double readlane_d(double v, int lane)
{
    unsigned long long s = __double_as_longlong(v);
    int lo = __builtin_amdgcn_readlane((int)(s & 0xffffffffu), lane);
    int hi = __builtin_amdgcn_readlane((int)(s >> 32), lane);
    return __longlong_as_double(((unsigned long long)(unsigned)hi << 32) | (unsigned)lo);
}

// max of non-negative doubles through their bit patterns: NaN (canonicalised by abs_nan_canon) is the largest, so a
// NaN anywhere surfaces in the error norm (solvers/base.py:577-590)
__device__ __forceinline__ double abs_nan_canon(double e)
{
    e = fabs(e);
    return (e != e) ? __longlong_as_double(0x7ff8000000000000LL) : e;
}
__device__ __forceinline__ double max_nan(double a, double b)
{
    const unsigned long long x = (unsigned long long)__double_as_longlong(a), y = (unsigned long long)__double_as_longlong(b);
    return __longlong_as_double((long long)((x > y) ? x : y));
}

// XCD-aware block -> work-item map: hardware places block b on XCD b % 8 (speed only, never
// correctness).  Every XCD gets a contiguous range of the tile list so that tiles sharing halo
// rows also share an L2.
__device__ __forceinline__ long xcd_swizzle(long bid, long nblocks)
{
    const long per = nblocks / 8;
    if (bid >= per * 8) return bid;  // tail: identity
    return (bid % 8) * per + bid / 8;
}

template <int MODE>
__device__ __forceinline__ double epilogue(double lap, double c, double yv, double s1, double s2, double gamma)
{
    if (MODE == LAP_PLAIN) return lap;
    if (MODE == LAP_SCALED || MODE == LAP_STAGE) return s2 * (s1 * lap);   // dt * (D * lap)
    if (MODE == LAP_EULER) return yv + s2 * (s1 * lap);    // pde/solvers/euler.py:174
    return c * c * c - c - gamma * lap;                    // pde/pdes/cahn_hilliard.py:116-120
}

// boundary condition evaluated on the fly on the input side:  virtual = c + f * in[idx]
struct InBC {
    int on;
    long idx;      // valid index along the axis the virtual value is computed from
    double c, f;
};

struct LapArgs {
    const void *in;
    void *out;
    const void *y;
    long n0, n1, n2;
    long p0, p1, off;
    long o_off, o_s0, o_s1;
    double sx, sy, sz, s1, s2, gamma;
    int ndim;
    int lx;           // planes per x-chunk
    long nxc, nty, ntz, nblocks;
    int no_swizzle;
    long o_sc;        // component stride of the output (gradient modes)
    double gs[3];     // per-axis scale of the derivative modes
    int any_ibc;
    const void *ex[3];   // LAP_CUSTOM: up to three extra input arrays (centre values only)
    double par[12];      // LAP_CUSTOM: run-time scalars of the generated epilogue (dt, constants, t, ...)
    InBC ibc[3][2];   // [normalised axis][lower, upper]
    int per[3];       // euler2_kernel: axis is periodic (else both faces are local first-order BCs)
    long xstride;     // euler2_kernel: first plane of x-chunk xc is xc * xstride (== lx, or n0 - lx for the two-ended boundary sweep)
    int nwy;          // euler2_kernel: waves of a workgroup stacked along the rows (others: along the fastest axis)
    InBC ibc1[3][2];  // euler2_kernel: faces of the intermediate level when it is another field (Cahn-Hilliard: mu)
    // LAP_STAGE: what follows the slope k (still in registers) in a Runge-Kutta scheme, all arrays in the layout of `in`
    //   st_kind 0: out = k, st_out = y + sum_m st_c[m]*st_k[m] + st_c[5]*k      (input of the next stage, runge_kutta.py:135-145)
    //   st_kind 1: st_out = y + (k1 + 2*k2 + 2*k3 + k)/6, `out` is not written     (new state of RK4, runge_kutta.py:60)
    //   st_kind 2: st_out = y + c1*k1 + c3*k3 + c4*k4 + c5*k5 and *st_err = max |error estimate| with k6 = k; st_k = {k1, k3, k4, k5},
    //              `out` is not written                                          (end of an RKF45 attempt, runge_kutta.py:147-150)
    //   st_kind 3: out = k (the rate, s2 = 1), st_out = y + st_c[5] * (1.5*k - 0.5*st_k[0])   (Adams-Bashforth step, adams_bashforth.py:44)
    int st_kind;
    double *st_err;
    const void *st_y;
    const void *st_k[5];   // earlier slopes, NULL-terminated
    double st_c[6];
    void *st_out;
    // LAP_CUSTOM: per-axis derivative scales of the generated epilogue: d1 = (r - l) / dd1 with dd1 = 2 dx (operators/common.py:60-110),
    // d2 = (r - 2c + l) * dd2 with dd2 = 1 / dx^2 (:150-190).  (At the end: the kernarg offsets of everything else stay.)
    double dd1[3], dd2[3];
    double dg[3];   // LAP_CUSTOM: 0.5 / dx, the scale of the components of `gradient` / `divergence` (cartesian.py:451-454, :876-879)
    // lap_march_kernel, split rows (launch_laplace_t): the first `strip_blocks` workgroups compute the `strip_n2` columns behind the n2
    // columns of the vectorised part, one cell per thread and 8 rows per thread (lap_strip)
    long strip_blocks;
    int strip_n2;
    // euler2_kernel, "open" rows (launch_euler2_tv): the tiles cover whole chunks only, the row goes on for one or two cells that another
    // kernel computes (pdehip_shell.hip) - the last tile is not moved back to the end of the row
    int z_open;
};

// per-axis central first and second derivatives at a cell, by normalised axis (a 2-D grid uses entries 1 and 2)
struct PdeDer { double d1[3], d2[3], gr[3]; };   // gr: components of the central gradient, (r - l) * (0.5 / dx)

// arguments of tile2d_kernel (pdehip_tile2d.inc): K Euler steps of a 2-D grid per launch, time levels in LDS
struct Tile2Args {
    const void *in;
    void *out;
    long n0, n1;           // rows (first grid axis), columns (fastest axis)
    long p1, off;          // row pitch, offset of cell (0, 0)
    double sx, sy;         // dx^-2 of the row axis / the column axis
    double s1, s2, gamma;  // diffusion: s1 = D, s2 = dt;  Cahn-Hilliard: gamma, s2 = dt
    int nsteps;            // K
    int per[2];            // periodic axes
    double c[2][2][2], f[2][2][2];   // [field: 0 = state, 1 = mu][axis][side]: ghost = c + f * adjacent cell (local faces)
    int tiles1;            // tiles along the columns
    // MODE 2 (run-time generated update): parameters and derivative scales of the generated epilogue, [row axis, column axis]
    double par[12];
    double gs[2], dd1[2], dd2[2], dg[2];
    long pc;               // MODE 3 (two fields): elements between the two component arrays
    double t0;             // MODE 2-4: time of step 0 of the run and index of this launch's first step: level l is evaluated at
    long step0;            //   t0 + (step0 + l) * dt, the expression of the step loop (numba/_solvers.py:100), in p[1]
};

// LAP_CUSTOM: the pointwise epilogue is generated at run time (pde_hip/expr.py -> pdehip_jit.hip); the
// offline build never instantiates that mode and only needs the declaration to parse.
#ifdef PDEHIP_JIT
__device__ __forceinline__ double pde_epilogue(double c, double lap, double gsq, double e0, double e1, double e2, const double *p, const PdeDer &d);
__device__ __forceinline__ double pde_epilogue2(double c, double lap, double gsq, double e0, double e1, double e2, const double *p, const PdeDer &d);
#else
__device__ __forceinline__ double pde_epilogue(double, double, double, double, double, double, const double *, const PdeDer &) { return 0.0; }
__device__ __forceinline__ double pde_epilogue2(double, double, double, double, double, double, const double *, const PdeDer &) { return 0.0; }
#endif

}  // namespace pdehip
