// pdehip_shell.hip — two Euler steps of the diffusion equation per sweep with faces whose coefficients change from step to step.
//
// Conditions given as expressions of time and position (pde/grids/boundaries/local.py:766-1150, `value_expression: "sin(t) * y"`) reach
// the device as per-cell coefficient arrays that a run-time compiled program rewrites for every evaluation of the right-hand side
// (include/pdehip.h: pdehip_bcprog_*).  The two-step sweep (pdehip_march2.inc) applies its conditions on the fly from SCALAR coefficients,
// the same for both levels - so such runs took one step per sweep, with a refresh launch and a ghost-cell launch per step (0.52 against
// 0.22 ms per step at 512^3, profiles/r03_time_bc_program.log).
//
// Here: the second level of a two-step sweep needs the faces at t + dt, which depend on nothing but t + dt and the position (conditions
// that read the FIELD stay on the one-step path): the program writes a second coefficient set.  The sweep itself runs with stand-in
// coefficients on those faces; what it computes wrongly is exactly the cells whose two-step domain of dependence contains a virtual point
// of such a face - the two layers of cells next to it.  `shell_kernel` recomputes those cells, one per thread, from the input field with
// the true coefficients of both levels and overwrites them (same stream, behind the sweep).  Every operation is the one of the one-step
// path: virtual point = (T)(c + f * adjacent), level 1 rounded to the storage type, the Laplacian summed axis by axis
// (cartesian.py:147-151 / :220-227), `u + dt * (D * lap)` (pde/solvers/euler.py:172-175) - bit-identical to two single steps.

#include "pdehip_common.h"

namespace pdehip {
namespace {

struct ShellFace {
    int mode;               // 1 periodic (wrap), 2 local with scalar coefficients, 3 local with coefficient arrays
    double c[2], f[2];      // [level]
    const double *ca[2], *fa[2];
};
// one face given as arrays: its two layers of cells as a list of workgroup tiles
struct ShellJob { int ax, side; long first, nb0, nb1, nb2; int origin; };   // first workgroup; workgroups per (normalised) axis; first of the two layers
struct ShellArgs {
    const void *in;
    void *out;
    long n[3], p[3], off;
    int ni[3], pi[2];       // the same as 32-bit values (the index arithmetic of the kernel; launch_euler2 admits pitches < 2^31 only)
    int first_axis;         // 0 for 3-D, 1 for 2-D grids (normalised axes)
    int per[3];             // the axis is periodic
    double sc[3], s1, s2;
    ShellFace face[3][2];
    int njobs;
    ShellJob job[6];
};

// A workgroup recomputes a box of S0 x S1 x S2 cells (two layers along the face's axis): it stages the input it needs (the box widened by
// two cells) in LDS, computes level 1 ONCE per cell of the box widened by one cell into LDS, and takes level 2 from there.  One cell per
// thread straight from memory - seven evaluations of level 1 per output, ~60 operands - took 0.97 ms at 512^3, 0.8 of it at the faces of
// the fastest axis, where every operand of a thread is a cache line of its own; with the input staged 0.19 ms; level 1 shared through LDS 0.126 ms
// (profiles/r04_time_bc_program.md).
template <int AX> struct TileDims {
    static constexpr int S0 = AX == 0 ? 2 : (AX == 1 ? 4 : 8);
    static constexpr int S1 = AX == 0 ? 4 : (AX == 1 ? 2 : 16);
    static constexpr int S2 = AX == 2 ? 2 : 32;
    static_assert(S0 * S1 * S2 == 256, "one output cell per thread");
};
constexpr int kBox0 = 1728;   // max (S0 + 4) * (S1 + 4) * (S2 + 4): 6 x 8 x 36
constexpr int kBox1 = 816;    // max (S0 + 2) * (S1 + 2) * (S2 + 2): 4 x 6 x 34

// (every axis index below is a compile-time constant: the cell coordinates stay in registers and the face table is read with constant
// offsets - with run-time axis indices the coordinates lived in scratch memory)
struct Cell { int x0, x1, x2; };
template <int AX> __device__ __forceinline__ int coord(const Cell &x) { return AX == 0 ? x.x0 : (AX == 1 ? x.x1 : x.x2); }
__device__ __forceinline__ long elem(const ShellArgs &a, const Cell &x) { return a.off + (long)x.x0 * a.pi[0] + (long)x.x1 * a.pi[1] + x.x2; }   // (p[2] == 1)
__device__ __forceinline__ bool inside(const ShellArgs &a, const Cell &x)
{
    return (unsigned)x.x0 < (unsigned)a.ni[0] && (unsigned)x.x1 < (unsigned)a.ni[1] && (unsigned)x.x2 < (unsigned)a.ni[2];
}
// A tile works in UNWRAPPED coordinates: next to the end of a periodic axis its boxes reach one or two cells beyond the grid, and those
// entries hold the cells of the far side (staged from there, level 1 computed like for any other cell).  So every operand of a cell of the
// level-1 box is in the input box and every operand of an output is in the level-1 box: plain LDS reads at the centre's index +- the
// compile-time stride of the axis - no range checks, no second code path, no index arithmetic per operand.  (With a fallback path inlined
// the kernel was 317 KB of code for a 64 KB instruction cache: 0.145 ms; with coordinates -> index per operand it issued ~1090 lane
// operations per output and was bound by them: 0.126 ms, VALU busy 66 %, profiles/r04_shell_kernel_counters.md.)
template <int AX> __device__ __forceinline__ int phys(const ShellArgs &a, int q)   // the cell of the grid behind an unwrapped coordinate
{
    return q < 0 ? q + a.ni[AX] : (q >= a.ni[AX] ? q - a.ni[AX] : q);
}
// inside the grid, or an image across a periodic axis - the two layers of images a tile of the last cells can need, not more: a box may reach
// further out than that when the axis is shorter than the tile (6 rows under a 16-row tile), and `phys` wraps once
template <int AX> __device__ __forceinline__ bool exists1(const ShellArgs &a, int q)
{
    return (unsigned)q < (unsigned)a.ni[AX] || (a.per[AX] && (unsigned)(q + 2) < (unsigned)(a.ni[AX] + 4));
}
__device__ __forceinline__ bool exists(const ShellArgs &a, const Cell &x) { return exists1<0>(a, x.x0) && exists1<1>(a, x.x1) && exists1<2>(a, x.x2); }

// extents of the two boxes of a tile next to a face of axis JAX: the input box (level 0) and the level-1 box
template <int JAX, int LV> struct BoxDims {
    typedef TileDims<JAX> D;
    static constexpr int W = LV == 0 ? 4 : 2;
    static constexpr int B0 = D::S0 + W, B1 = D::S1 + W, B2 = D::S2 + W;
    template <int AX> static constexpr int stride() { return AX == 0 ? B1 * B2 : (AX == 1 ? B2 : 1); }
};

// coefficients of face (AX, SIDE) at the face cell of `x`, level lv
template <int AX, int SIDE>
__device__ __forceinline__ void face_coef(const ShellArgs &a, int lv, const Cell &x, double *c, double *f)
{
    const ShellFace &F = a.face[AX][SIDE];
    if (F.mode == 3) {
        constexpr int o1 = (AX == 0) ? 1 : 0, o2 = (AX == 2) ? 1 : 2;
        const long e = (long)phys<o1>(a, coord<o1>(x)) * a.ni[o2] + phys<o2>(a, coord<o2>(x));
        *c = F.ca[lv][e];
        *f = F.fa[lv][e];
    } else {
        *c = F.c[lv];
        *f = F.f[lv];
    }
}

// level LV at the neighbour of the cell x (box entry ci) on side SIDE of axis AX: a cell (of the grid, or its image across a periodic axis)
// or the virtual point `c + f * (adjacent cell = x)` (local.py:1636)
template <typename T, int JAX, int LV, int AX, int SIDE>
__device__ __forceinline__ double neighbour(const ShellArgs &a, const T *box, const Cell &x, int ci, double cen)
{
    const int q = coord<AX>(x) + (SIDE ? 1 : -1);
    // (x itself may be an image: then q is further out still - a cell as well)
    if ((unsigned)q < (unsigned)a.ni[AX] || a.per[AX]) return (double)box[ci + (SIDE ? 1 : -1) * BoxDims<JAX, LV>::template stride<AX>()];
    double c, f;
    face_coef<AX, SIDE>(a, LV, x, &c, &f);
    return (double)(T)(c + f * cen);
}

template <typename T, int JAX, int LV, int AX>
__device__ __forceinline__ void add_axis(const ShellArgs &a, const T *box, const Cell &x, int ci, double cen, double vm, double &lap, bool &first)
{
    if (AX < a.first_axis) return;
    const double lm = neighbour<T, JAX, LV, AX, 0>(a, box, x, ci, cen), lp = neighbour<T, JAX, LV, AX, 1>(a, box, x, ci, cen);
    const double l = (lm - vm + lp) * a.sc[AX];
    lap = first ? l : lap + l;
    first = false;
}

// one Euler step at the cell x (entry ci of the box of level LV), rounded to the storage type like a stored field
template <typename T, int JAX, int LV>
__device__ __forceinline__ T euler_at(const ShellArgs &a, const T *box, const Cell &x, int ci)
{
    const double cen = (double)box[ci];
    const double vm = 2 * cen;
    double lap = 0;
    bool first = true;
    add_axis<T, JAX, LV, 0>(a, box, x, ci, cen, vm, lap, first);
    add_axis<T, JAX, LV, 1>(a, box, x, ci, cen, vm, lap, first);
    add_axis<T, JAX, LV, 2>(a, box, x, ci, cen, vm, lap, first);
    return (T)(cen + a.s2 * (a.s1 * lap));
}

// the tile (b0, b1, b2) of the two layers next to a face of axis AX (first layer: `origin`)
template <typename T, int AX>
__device__ __forceinline__ void shell_tile(const ShellArgs &a, int origin, int b0, int b1, int b2, T *lds0, T *lds1)
{
    typedef TileDims<AX> D;
    typedef BoxDims<AX, 0> G;
    typedef BoxDims<AX, 1> H;
    constexpr int S[3] = {D::S0, D::S1, D::S2};
    static_assert(G::B0 * G::B1 * G::B2 <= kBox0 && H::B0 * H::B1 * H::B2 <= kBox1, "LDS boxes");
    int o[3] = {b0 * S[0], b1 * S[1], b2 * S[2]};
    o[AX] = origin;   // (n >= 4 along every axis: launch_euler2)
    {   // the input box; all loads of the thread first, then the stores: one memory round trip per workgroup instead of one per pass
        constexpr int NG = G::B0 * G::B1 * G::B2, NA = (NG + 255) / 256;
        T v[NA];
#pragma unroll
        for (int m = 0; m < NA; m++) {
            const int e = threadIdx.x + m * 256;
            const int c2 = e % G::B2, c1 = (e / G::B2) % G::B1, c0 = e / (G::B2 * G::B1);
            const Cell x = {o[0] - 2 + c0, o[1] - 2 + c1, o[2] - 2 + c2};
            v[m] = 0;
            if (e < NG && exists(a, x)) {
                const Cell y = {phys<0>(a, x.x0), phys<1>(a, x.x1), phys<2>(a, x.x2)};
                v[m] = ((const T *)a.in)[elem(a, y)];
            }
        }
#pragma unroll
        for (int m = 0; m < NA; m++) {
            const int e = threadIdx.x + m * 256;
            if (e < NG) lds0[e] = v[m];
        }
    }
    __syncthreads();
    {   // level 1, once per cell of the box widened by one
        constexpr int NH = H::B0 * H::B1 * H::B2, NE = (NH + 255) / 256;
#pragma unroll
        for (int m = 0; m < NE; m++) {   // (unrolled: the coefficient loads of the passes overlap)
            const int e = threadIdx.x + m * 256;
            const int c2 = e % H::B2, c1 = (e / H::B2) % H::B1, c0 = e / (H::B2 * H::B1);
            const Cell x = {o[0] - 1 + c0, o[1] - 1 + c1, o[2] - 1 + c2};
            T v = 0;
            if (e < NH && exists(a, x)) v = euler_at<T, AX, 0>(a, lds0, x, ((c0 + 1) * G::B1 + (c1 + 1)) * G::B2 + (c2 + 1));
            if (e < NH) lds1[e] = v;
        }
    }
    __syncthreads();
    const int t2 = threadIdx.x % S[2], t1 = (threadIdx.x / S[2]) % S[1], t0 = threadIdx.x / (S[2] * S[1]);
    const Cell x = {o[0] + t0, o[1] + t1, o[2] + t2};
    if (!inside(a, x)) return;
    ((T *)a.out)[elem(a, x)] = euler_at<T, AX, 1>(a, lds1, x, ((t0 + 1) * H::B1 + (t1 + 1)) * H::B2 + (t2 + 1));
}

// The arguments - six face descriptors among them - are read from LDS: as scalar registers they did not fit (106 SGPRs and ~3000 `v_readlane`
// in the code: values spilled into lanes of vector registers, half of the instructions a wave issued).  The first lanes copy the kernel
// argument segment word by word.
template <typename T>
__global__ void __launch_bounds__(256) shell_kernel(ShellArgs by_value)
{
    __shared__ T lds0[kBox0];
    __shared__ T lds1[kBox1];
    __shared__ ShellArgs a;
    {
        const int *src = (const int *)__builtin_amdgcn_kernarg_segment_ptr();   // (`by_value` sits at offset 0 of the segment)
        int *dst = (int *)&a;
        for (int w = threadIdx.x; w < (int)(sizeof(ShellArgs) / sizeof(int)); w += 256) dst[w] = src[w];
        (void)by_value;
    }
    __syncthreads();
    int q = 0;
#pragma unroll
    for (int m = 1; m < 6; m++)
        if (m < a.njobs && (long)blockIdx.x >= a.job[m].first) q = m;
    long first = a.job[0].first, nb1 = a.job[0].nb1, nb2 = a.job[0].nb2;
    int ax = a.job[0].ax, origin = a.job[0].origin;
#pragma unroll
    for (int m = 1; m < 6; m++)
        if (q == m) { first = a.job[m].first; nb1 = a.job[m].nb1; nb2 = a.job[m].nb2; ax = a.job[m].ax; origin = a.job[m].origin; }
    long b = (long)blockIdx.x - first;
    const long b2 = b % nb2;
    b /= nb2;
    const long b1 = b % nb1, b0 = b / nb1;
    if (ax == 0) shell_tile<T, 0>(a, origin, (int)b0, (int)b1, (int)b2, lds0, lds1);
    else if (ax == 1) shell_tile<T, 1>(a, origin, (int)b0, (int)b1, (int)b2, lds0, lds1);
    else shell_tile<T, 2>(a, origin, (int)b0, (int)b1, (int)b2, lds0, lds1);
}

}  // namespace

// Two Euler steps of `in` into `out` with the faces `faces` (grid axes; level 0: their own coefficients, level 1: for array faces the
// second set `second(const_arr) -> (const, factor)` of the program, for scalar faces the same scalars).  *done = false, nothing launched,
// when the grid or a face is not covered (the caller steps once per sweep).
int euler2_timed_faces(const pdehip_grid_t *g, const void *in, void *out, double s1, double s2, const pdehip_bc_face_t *faces,
                       void *bc_program, void *stream, bool *done)
{
    *done = false;
    NGrid n;
    PDEHIP_TRY(norm_grid(g, &n));
    if (!in || !out || !faces) PDEHIP_FAIL(E_VALUE, "euler2: NULL pointer");
    if (n.ndim < 2) return 0;
    InputBCs fg;
    memset(&fg, 0, sizeof(fg));
    ShellArgs a;
    memset(&a, 0, sizeof(a));
    long total = 0;   // workgroups
    for (int ar = 0; ar < n.ndim; ar++)
        for (int side = 0; side < 2; side++) {
            const int ax = 3 - n.ndim + ar;
            const pdehip_bc_face_t &r = faces[2 * ar + side];
            if (r.kind != PDEHIP_BC_ORDER1 || r.index1 < 0 || r.index1 >= n.n[ax]) return 0;
            ShellFace &F = a.face[ax][side];
            fg.on[ax][side] = 1;
            fg.idx[ax][side] = r.index1;
            if (r.flags == 0) {
                fg.c[ax][side] = r.const_v;
                fg.f[ax][side] = r.factor1;
                const bool wraps = r.index1 == (side ? 0 : n.n[ax] - 1) && r.const_v == 0 && r.factor1 == 1 && n.n[ax] > 1;
                F.mode = wraps ? 1 : 2;
                if (!wraps && r.index1 != (side ? n.n[ax] - 1 : 0)) return 0;
                F.c[0] = F.c[1] = r.const_v;
                F.f[0] = F.f[1] = r.factor1;
            } else if (r.flags == PDEHIP_BCF_ARRAYS) {
                if (r.index1 != (side ? n.n[ax] - 1 : 0) || !r.const_arr || !r.factor1_arr) return 0;
                // stand-in for the sweep: a local face (what it computes next to it is overwritten below)
                fg.c[ax][side] = 0;
                fg.f[ax][side] = 1;
                F.mode = 3;
                F.ca[0] = r.const_arr;
                F.fa[0] = r.factor1_arr;
                // (arrays no program rewrites - conditions that depend on the position only - serve both levels)
                if (!bc_program || !bcprog_second_set(bc_program, r.const_arr, &F.ca[1], &F.fa[1])) { F.ca[1] = F.ca[0]; F.fa[1] = F.fa[0]; }
                ShellJob &J = a.job[a.njobs++];
                const long tile0 = ax == 0 ? 2 : (ax == 1 ? 4 : 8), tile1 = ax == 0 ? 4 : (ax == 1 ? 2 : 16), tile2 = ax == 2 ? 2 : 32;   // TileDims
                J.ax = ax; J.side = side; J.first = total; J.origin = side ? (int)n.n[ax] - 2 : 0;
                J.nb0 = ax == 0 ? 1 : (n.n[0] + tile0 - 1) / tile0;
                J.nb1 = ax == 1 ? 1 : (n.n[1] + tile1 - 1) / tile1;
                J.nb2 = ax == 2 ? 1 : (n.n[2] + tile2 - 1) / tile2;
                total += J.nb0 * J.nb1 * J.nb2;
            } else {
                return 0;
            }
        }
    // (an axis whose two sides differ in kind - periodic on one side only - is refused by the sweep)
    PDEHIP_TRY(launch_euler2(n, in, out, s1, s2, fg, 0, as_stream(stream), done, false, 0));
    if (!*done || a.njobs == 0) return 0;
    a.in = in; a.out = out; a.off = n.off; a.first_axis = 3 - n.ndim;
    for (int k = 0; k < 3; k++) { a.n[k] = n.n[k]; a.p[k] = n.p[k]; a.sc[k] = n.lap_scale[k]; a.ni[k] = (int)n.n[k]; }
    a.pi[0] = (int)n.p[0]; a.pi[1] = (int)n.p[1];
    for (int k = 0; k < 3; k++) a.per[k] = a.face[k][0].mode == 1 && a.face[k][1].mode == 1;
    if (n.p[0] >= (1L << 31) || n.p[2] != 1) PDEHIP_FAIL(E_RUNTIME, "internal: pitch of the slowest axis beyond 2^31 elements");
    a.s1 = s1; a.s2 = s2;
    if (total >= (1L << 31)) PDEHIP_FAIL(E_RUNTIME, "internal: too many workgroups for the cells next to the faces");
    if (n.dtype == PDEHIP_F64) hipLaunchKernelGGL((shell_kernel<double>), dim3((unsigned)total), dim3(256), 0, as_stream(stream), a);
    else hipLaunchKernelGGL((shell_kernel<float>), dim3((unsigned)total), dim3(256), 0, as_stream(stream), a);
    PDEHIP_HIP(hipGetLastError());
    return 0;
}

// (see preload_stencil_kernels: the code object of this translation unit is loaded when the device is selected as well, not in the middle of a run)
int preload_shell_kernels()
{
    hipFuncAttributes attr;
    PDEHIP_HIP(hipFuncGetAttributes(&attr, reinterpret_cast<const void *>(&shell_kernel<double>)));
    return 0;
}

// The last one to eight columns of every row behind a two-step sweep whose tiles cover whole chunks only (launch_euler2_tv, "open" rows):
// the two layers of cells next to the upper face of the fastest axis, with the scalar conditions of the sweep (`a`: kernel axes).
// `rows` (round 6, "open" columns of tiles): the same for the last one to seven ROWS of every plane - the two layers next to the upper face of the rows.
int shell_open_rows(const NGrid &n, const LapArgs &la, int columns, int rows, hipStream_t st)
{
    ShellArgs a;
    memset(&a, 0, sizeof(a));
    a.first_axis = 3 - n.ndim;
    for (int ax = a.first_axis; ax < 3; ax++) {
        const int k = n.ndim == 3 ? ax : (ax == 1 ? 0 : 2);   // kernel axis of the normalised axis (2-D: the march axis is the first grid axis)
        for (int side = 0; side < 2; side++) {
            ShellFace &F = a.face[ax][side];
            F.mode = la.per[k] == 1 ? 1 : 2;
            if (la.per[k] != 0 && la.per[k] != 1) PDEHIP_FAIL(E_RUNTIME, "internal: open rows of a slab sweep");
            F.c[0] = F.c[1] = la.ibc[k][side].c;
            F.f[0] = F.f[1] = la.ibc[k][side].f;
        }
    }
    // (Round 6 tried a transposed kernel here - a lane = a row, strips of six cells of six planes in registers, row neighbours by DPP, like
    // rimz_kernel of the fast block loop: bit-exact and SLOWER than the LDS boxes below, 0.2333 against 0.2229 ms per step at 512 x 512 x 513,
    // 0.2507 against 0.2300 at 512 x 512 x 520 - eighteen loads per lane, each a cache line of its own: profiles/r06_call12_time_sizes.md.  Removed.)
    long total = 0;
    if ((columns + 1) / 2 + (rows + 1) / 2 > 6) PDEHIP_FAIL(E_RUNTIME, "internal: open rows of %d columns, %d rows: more than six jobs", columns, rows);
    for (int c = 0; c < columns; c += 2) {   // two layers per job, from the end of the row inwards
        ShellJob &J = a.job[a.njobs++];
        J.ax = 2; J.side = 1; J.first = total; J.origin = (int)n.n[2] - 2 - c;
        J.nb0 = (n.n[0] + 7) / 8; J.nb1 = (n.n[1] + 15) / 16; J.nb2 = 1;   // TileDims<2>
        total += J.nb0 * J.nb1;
    }
    for (int c = 0; c < rows; c += 2) {
        ShellJob &J = a.job[a.njobs++];
        J.ax = 1; J.side = 1; J.first = total; J.origin = (int)n.n[1] - 2 - c;
        J.nb0 = (n.n[0] + 3) / 4; J.nb1 = 1; J.nb2 = (n.n[2] + 31) / 32;   // TileDims<1>
        total += J.nb0 * J.nb2;
    }
    if (columns < 0 || columns > 8 || rows < 0 || rows > 7 || a.njobs > 6 || columns + rows < 1 || n.n[2] < 16 || (rows && (n.ndim != 3 || n.n[1] < 16)))
        PDEHIP_FAIL(E_RUNTIME, "internal: open rows of %d columns, %d rows", columns, rows);
    a.in = la.in; a.out = la.out; a.off = n.off;
    for (int k = 0; k < 3; k++) { a.n[k] = n.n[k]; a.p[k] = n.p[k]; a.sc[k] = n.lap_scale[k]; a.ni[k] = (int)n.n[k]; }
    a.pi[0] = (int)n.p[0]; a.pi[1] = (int)n.p[1];
    for (int k = 0; k < 3; k++) a.per[k] = a.face[k][0].mode == 1 && a.face[k][1].mode == 1;
    a.s1 = la.s1; a.s2 = la.s2;
    if (total >= (1L << 31) || n.p[0] >= (1L << 31) || n.p[2] != 1) PDEHIP_FAIL(E_RUNTIME, "internal: grid beyond the index range of the shell kernel");
    if (n.dtype == PDEHIP_F64) hipLaunchKernelGGL((shell_kernel<double>), dim3((unsigned)total), dim3(256), 0, st, a);
    else hipLaunchKernelGGL((shell_kernel<float>), dim3((unsigned)total), dim3(256), 0, st, a);
    PDEHIP_HIP(hipGetLastError());
    return 0;
}

}  // namespace pdehip
