// pdehip_ops.hip — gradient / divergence / gradient_squared, layout conversion and the
// pointwise kernels of the Runge–Kutta steppers (all bandwidth-bound, 16-byte vector access
// over the interior rows of the aligned device layout).
#include "pdehip_common.h"

namespace pdehip {

struct DevGrid {
    long n0, n1, n2;
    long p0, p1, pc, off;
    int ndim;
};
static DevGrid dev_grid(const NGrid &n)
{
    DevGrid d;
    d.n0 = n.n[0]; d.n1 = n.n[1]; d.n2 = n.n[2];
    d.p0 = n.p[0]; d.p1 = n.p[1]; d.pc = n.pc; d.off = n.off; d.ndim = n.ndim;
    return d;
}

template <typename T, int VEC> struct VecOf;
template <> struct VecOf<double, 2> { typedef double type __attribute__((ext_vector_type(2))); };
template <> struct VecOf<float, 4> { typedef float type __attribute__((ext_vector_type(4))); };
template <> struct VecOf<double, 1> { typedef double type __attribute__((ext_vector_type(1))); };
template <> struct VecOf<float, 1> { typedef float type __attribute__((ext_vector_type(1))); };

// Iterate over all interior cells of `ncomp` components in chunks of VEC cells along the
// fastest axis.  body(comp, i, j, k, e) with e = element offset inside the full array.
template <int VEC, typename F>
__device__ __forceinline__ void for_each_chunk(const DevGrid &g, int ncomp, F body)
{
    const long kc = g.n2 / VEC;
    const long total = (long)ncomp * g.n0 * g.n1 * kc;
    for (long t = blockIdx.x * (long)blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
        long r = t;
        const long k = (r % kc) * VEC;
        r /= kc;
        const long j = r % g.n1;
        r /= g.n1;
        const long i = r % g.n0;
        const int comp = (int)(r / g.n0);
        body(comp, i, j, k, (long)comp * g.pc + g.off + i * g.p0 + j * g.p1 + k);
    }
}

static inline unsigned grid_blocks(long items)
{
    long b = (items + 255) / 256;
    if (b < 1) b = 1;
    if (b > 16384) b = 16384;
    return (unsigned)b;
}

// ---- derivative operators (one cell per thread; neighbours through L1/L2) ------------------------
struct DerivArgs {
    DevGrid g;
    const void *in;
    void *out;
    long o_off, o_s0, o_s1, o_sc;
    double dx[3];
    int method;   // gradient/divergence: PDEHIP_CENTRAL..; gradient_squared: central flag
};

template <typename T>
__global__ void __launch_bounds__(256) gradient_kernel(DerivArgs a)
{
    const T *in = (const T *)a.in;
    T *out = (T *)a.out;
    const DevGrid &g = a.g;
    for_each_chunk<1>(g, 1, [&](int, long i, long j, long k, long e) {
        const double mid = (double)in[e];
        for (int c = 0; c < g.ndim; c++) {
            const int ax = 3 - g.ndim + c;
            const long pa = (ax == 0) ? g.p0 : (ax == 1) ? g.p1 : 1;
            const double dx = a.dx[ax];
            const double hi = (double)in[e + pa], lo = (double)in[e - pa];
            const double d = (a.method == PDEHIP_CENTRAL) ? hi - lo : (a.method == PDEHIP_FORWARD) ? hi - mid : mid - lo;
            double r;
            if (g.ndim == 1)  // cartesian.py:418-422 divides in 1-D
                r = (a.method == PDEHIP_CENTRAL) ? d / (2 * dx) : d / dx;
            else              // cartesian.py:464-474, :516-548 multiply by 0.5/dx or 1/dx
                r = d * ((a.method == PDEHIP_CENTRAL) ? 0.5 / dx : 1 / dx);
            out[(long)c * a.o_sc + a.o_off + i * a.o_s0 + j * a.o_s1 + k] = (T)r;
        }
    });
}

template <typename T>
__global__ void __launch_bounds__(256) divergence_kernel(DerivArgs a)
{
    const T *in = (const T *)a.in;
    T *out = (T *)a.out;
    const DevGrid &g = a.g;
    for_each_chunk<1>(g, 1, [&](int, long i, long j, long k, long e) {
        double acc = 0;
        for (int c = 0; c < g.ndim; c++) {
            const int ax = 3 - g.ndim + c;
            const long pa = (ax == 0) ? g.p0 : (ax == 1) ? g.p1 : 1;
            const double dx = a.dx[ax];
            const T *ca = in + (long)c * g.pc;
            const double hi = (double)ca[e + pa], lo = (double)ca[e - pa], mid = (double)ca[e];
            const double d = (a.method == PDEHIP_CENTRAL) ? hi - lo : (a.method == PDEHIP_FORWARD) ? hi - mid : mid - lo;
            double t;
            if (g.ndim == 1)  // cartesian.py:843-848
                t = (a.method == PDEHIP_CENTRAL) ? d / (2 * dx) : d / dx;
            else              // cartesian.py:889-900, :942-957
                t = d * ((a.method == PDEHIP_CENTRAL) ? 0.5 / dx : 1 / dx);
            acc = (c == 0) ? t : acc + t;
        }
        out[a.o_off + i * a.o_s0 + j * a.o_s1 + k] = (T)acc;
    });
}

template <typename T>
__global__ void __launch_bounds__(256) gradsq_kernel(DerivArgs a)
{
    const T *in = (const T *)a.in;
    T *out = (T *)a.out;
    const DevGrid &g = a.g;
    for_each_chunk<1>(g, 1, [&](int, long i, long j, long k, long e) {
        const double mid = (double)in[e];
        double acc = 0;
        for (int c = 0; c < g.ndim; c++) {
            const int ax = 3 - g.ndim + c;
            const long pa = (ax == 0) ? g.p0 : (ax == 1) ? g.p1 : 1;
            const double dx = a.dx[ax];
            const double hi = (double)in[e + pa], lo = (double)in[e - pa];
            double t;
            if (a.method) {  // cartesian.py:665-668 central
                const double d = hi - lo;
                t = d * d * (0.25 / (dx * dx));
            } else {         // cartesian.py:674-688 mean of forward/backward squares
                const double dl = hi - mid, dr = mid - lo;
                t = (dl * dl + dr * dr) * (0.5 / (dx * dx));
            }
            acc = (c == 0) ? t : acc + t;
        }
        out[a.o_off + i * a.o_s0 + j * a.o_s1 + k] = (T)acc;
    });
}

struct AxisArgs {
    DevGrid g;
    const void *in;
    void *out;
    long o_off, o_s0, o_s1;
    long pa;        // pitch along the axis
    double dx;
    int order, method;
};

template <typename T>
__global__ void __launch_bounds__(256) axis_derivative_kernel(AxisArgs a)
{
    const T *in = (const T *)a.in;
    T *out = (T *)a.out;
    for_each_chunk<1>(a.g, 1, [&](int, long i, long j, long k, long e) {
        const double c = (double)in[e], l = (double)in[e - a.pa], r = (double)in[e + a.pa];
        double v;
        if (a.order == 2) v = (r - 2 * c + l) * (1 / (a.dx * a.dx));              // common.py:150-190
        else if (a.method == PDEHIP_CENTRAL) v = (r - l) / (2 * a.dx);            // common.py:60-110
        else if (a.method == PDEHIP_FORWARD) v = (r - c) / a.dx;
        else v = (c - l) / a.dx;
        out[a.o_off + i * a.o_s0 + j * a.o_s1 + k] = (T)v;
    });
}

// ---- integral over the grid: two deterministic passes --------------------------------------------------------------
struct SumArgs {
    DevGrid g;
    const void *in;
    double *partial;   // [ncomp][nblocks]
    double *out;       // [ncomp]
    double vol;
    int comp, nblocks;
};
__device__ __forceinline__ double block_sum(double v)
{
#pragma unroll
    for (int ofs = 32; ofs >= 1; ofs >>= 1) v += __shfl_xor(v, ofs, 64);
    __shared__ double part[4];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) part[w] = v;
    __syncthreads();
    double s = 0;
    if (threadIdx.x == 0)
        for (int q = 0; q < (int)(blockDim.x >> 6); q++) s += part[q];
    return s;   // valid in thread 0
}
template <typename T>
__global__ void __launch_bounds__(256) partial_sum_kernel(SumArgs a)
{
    const T *in = (const T *)a.in + (long)a.comp * a.g.pc;
    double acc = 0;
    for_each_chunk<1>(a.g, 1, [&](int, long, long, long, long e) { acc += a.vol * (double)in[e]; });
    const double s = block_sum(acc);
    if (threadIdx.x == 0) a.partial[(long)a.comp * a.nblocks + blockIdx.x] = s;
}
// number of non-finite cells (NaN, +-inf) of one component: the partial / final sum skeleton with a predicate
template <typename T>
__global__ void __launch_bounds__(256) partial_nonfinite_kernel(SumArgs a)
{
    const T *in = (const T *)a.in + (long)a.comp * a.g.pc;
    double acc = 0;
    for_each_chunk<1>(a.g, 1, [&](int, long, long, long, long e) {
        const double v = (double)in[e];
        acc += (v - v == 0.0) ? 0.0 : 1.0;   // x - x is 0 for finite x, NaN for NaN and +-inf
    });
    const double s = block_sum(acc);
    if (threadIdx.x == 0) a.partial[(long)a.comp * a.nblocks + blockIdx.x] = s;
}
__global__ void __launch_bounds__(256) final_sum_kernel(SumArgs a)
{
    double acc = 0;
    for (int q = threadIdx.x; q < a.nblocks; q += blockDim.x) acc += a.partial[(long)a.comp * a.nblocks + q];
    const double s = block_sum(acc);
    if (threadIdx.x == 0) a.out[a.comp] = s;
}

// ---- Gaussian white noise increment of an Euler-Maruyama step (pde/solvers/euler.py:66-147) -----------------------------
// Philox4x32-10 (counter = {cell, call number}, key = seed) + Box-Muller, one draw per cell: reproducible for a given seed,
// independent of the launch geometry; twin of oracle_add_gaussian_noise.
struct NoiseArgs {
    DevGrid g;
    void *y;
    int ncomp;
    double scale;
    unsigned long long seed, counter, cell_offset;
};

__device__ inline double philox_normal(unsigned long long cell, unsigned long long counter, unsigned long long seed)
{
    unsigned c0 = (unsigned)cell, c1 = (unsigned)(cell >> 32), c2 = (unsigned)counter, c3 = (unsigned)(counter >> 32);
    unsigned k0 = (unsigned)seed, k1 = (unsigned)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; r++) {
        const unsigned long long p0 = 0xD2511F53ull * c0, p1 = 0xCD9E8D57ull * c2;
        const unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0, n1 = (unsigned)p1, n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1, n3 = (unsigned)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    const double u1 = ((double)((((unsigned long long)c0 << 32) | c1) >> 11) + 1.0) * (1.0 / 9007199254740992.0);
    const double u2 = (double)((((unsigned long long)c2 << 32) | c3) >> 11) * (1.0 / 9007199254740992.0);
    return sqrt(-2.0 * log(u1)) * cos(6.283185307179586476925286766559 * u2);
}

template <typename T>
__global__ void __launch_bounds__(256) gaussian_noise_kernel(NoiseArgs a)
{
    T *y = (T *)a.y;
    const long cells = a.g.n0 * a.g.n1 * a.g.n2;
    for_each_chunk<1>(a.g, a.ncomp, [&](int c, long i, long j, long k, long e) {
        const unsigned long long q = (unsigned long long)(c * cells + (i * a.g.n1 + j) * a.g.n2 + k) + a.cell_offset;
        y[e] = (T)((double)y[e] + a.scale * philox_normal(q, a.counter, a.seed));
    });
}

// ---- 2-D nine-point Laplacian (pde/backends/numba/operators/cartesian.py:153-190) ------------------------------------
struct Lap9Args {
    const void *in;
    void *out;
    long p1, off, nx, ny, o_off, o_s1;
    double st[3][3];
    int per_x, per_y;
};

// the four corner ghost cells, make_corner_point_setter_2d (cartesian.py:36-78), restated as written there
template <typename T>
__global__ void corner_points_kernel(Lap9Args a)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    T *arr = (T *)a.in;               // full indices (i, j) = 0 .. nx+1, 0 .. ny+1; `off` is the first interior cell (1, 1)
    const long p1 = a.p1, X = a.nx + 1, Y = a.ny + 1;
    auto at = [&](long i, long j) -> T & { return arr[a.off + (i - 1) * p1 + (j - 1)]; };
    if (a.per_x) {
        at(0, 0) = at(X - 1, 0); at(X, 0) = at(1, 0);
        at(0, Y) = at(X - 1, Y); at(X, Y) = at(1, Y);
    } else if (a.per_y) {
        at(0, 0) = at(0, Y - 1); at(X, 0) = at(X, 1);
        at(0, Y) = at(0, Y - 1); at(X, Y) = at(X, 1);
    } else {
        at(0, 0) = (T)(0.5 * ((double)at(0, 1) + (double)at(1, 0)));
        at(X, 0) = (T)(0.5 * ((double)at(X, 1) + (double)at(X - 1, 0)));
        at(0, Y) = (T)(0.5 * ((double)at(0, Y - 1) + (double)at(1, Y)));
        at(X, Y) = (T)(0.5 * ((double)at(X, Y - 1) + (double)at(X - 1, Y)));
    }
}

// one cell per thread, rows along threadIdx.x (coalesced); the 3 x 3 neighbourhood comes out of L1/L2
template <typename T>
__global__ void __launch_bounds__(256) laplace9_kernel(Lap9Args a)
{
    const T *in = (const T *)a.in;
    T *out = (T *)a.out;
    const long bpr = (a.ny + 255) / 256;   // blocks per row
    const long i = blockIdx.x / bpr, j = (blockIdx.x % bpr) * 256 + threadIdx.x;
    if (j >= a.ny) return;
    const T *c = in + a.off + i * a.p1 + j;
    double value = 0;   // accumulation order of cartesian.py:184-188
#pragma unroll
    for (int x = 0; x < 3; x++)
#pragma unroll
        for (int y = 0; y < 3; y++) value += (double)c[(x - 1) * a.p1 + (y - 1)] * a.st[x][y];
    out[a.o_off + i * a.o_s1 + j] = (T)value;
}

static int launch_deriv(int which, const pdehip_grid_t *g, int method, const void *in, void *out,
                        int layout, void *stream)
{
    NGrid n;
    PDEHIP_TRY(norm_grid(g, &n));
    if (!in || !out) PDEHIP_FAIL(E_VALUE, "operator: NULL array pointer");
    if (layout != PDEHIP_OUT_VALID && layout != PDEHIP_OUT_FULL) PDEHIP_FAIL(E_VALUE, "unknown output layout %d", layout);
    if (which != 2 && (method < 0 || method > 2)) PDEHIP_FAIL(E_VALUE, "Unknown derivative type `%d`", method);
    OutStr o = out_strides(n, layout);
    if (which != 1) {
        // gradient / gradient_squared: same loads as the Laplacian -> register-pipelined kernel
        double gs[3];
        for (int q = 0; q < 3; q++) {
            const double dx = n.dx[q];
            if (which == 0) gs[q] = (method == PDEHIP_CENTRAL) ? 0.5 / dx : 1 / dx;       // cartesian.py:451-454, :503-506
            else gs[q] = method ? 0.25 / (dx * dx) : 0.5 / (dx * dx);                      // cartesian.py:661, :672
        }
        const int mode = (which == 0) ? (method == PDEHIP_CENTRAL ? LAP_GRAD_C : method == PDEHIP_FORWARD ? LAP_GRAD_F : LAP_GRAD_B)
                                      : (method ? LAP_GRADSQ_C : LAP_GRADSQ_N);
        bool done = false;
        PDEHIP_TRY(launch_deriv_march(n, in, out, o, mode, gs, as_stream(stream), &done));
        if (done) return 0;
    } else {
        bool done = false;
        PDEHIP_TRY(launch_div_march(n, method, in, out, o, as_stream(stream), &done));
        if (done) return 0;
    }
    DerivArgs a;
    a.g = dev_grid(n);
    a.in = in; a.out = out;
    a.o_off = o.off; a.o_s0 = o.s0; a.o_s1 = o.s1; a.o_sc = o.sc;
    for (int q = 0; q < 3; q++) a.dx[q] = n.dx[q];
    a.method = method;
    const unsigned blocks = grid_blocks(n.n[0] * n.n[1] * n.n[2]);
    hipStream_t st = as_stream(stream);
#define PDEHIP_LAUNCH(KERNEL)                                                                  \
    if (n.dtype == PDEHIP_F64) hipLaunchKernelGGL((KERNEL<double>), dim3(blocks), dim3(256), 0, st, a); \
    else hipLaunchKernelGGL((KERNEL<float>), dim3(blocks), dim3(256), 0, st, a);
    if (which == 0) { PDEHIP_LAUNCH(gradient_kernel) }
    else if (which == 1) { PDEHIP_LAUNCH(divergence_kernel) }
    else { PDEHIP_LAUNCH(gradsq_kernel) }
#undef PDEHIP_LAUNCH
    PDEHIP_HIP(hipGetLastError());
    return 0;
}

// ---- layout conversion ---------------------------------------------------------------------------
struct CopyArgs {
    DevGrid g;
    const void *src;
    void *dst;
    int ncomp;
    int mode;  // 0 valid->full, 1 full->valid, 2 hostfull->full (all cells), 3 full->hostfull
};

template <typename T>
__global__ void __launch_bounds__(256) layout_copy_kernel(CopyArgs a)
{
    const DevGrid &g = a.g;
    if (a.mode < 2) {
        const T *src = (const T *)a.src;
        T *dst = (T *)a.dst;
        for_each_chunk<1>(g, a.ncomp, [&](int comp, long i, long j, long k, long e) {
            const long v = (((long)comp * g.n0 + i) * g.n1 + j) * g.n2 + k;
            if (a.mode == 0) dst[e] = src[v];
            else dst[v] = src[e];
        });
    } else {
        // compact host-full layout: shape (ncomp, [n0+2], [n1+2], n2+2) for the used axes
        const long g0 = (g.ndim >= 3) ? 1 : 0, g1 = (g.ndim >= 2) ? 1 : 0;
        const long m0 = g.n0 + 2 * g0, m1 = g.n1 + 2 * g1, m2 = g.n2 + 2;
        const long total = (long)a.ncomp * m0 * m1 * m2;
        const T *src = (const T *)a.src;
        T *dst = (T *)a.dst;
        const long lpad = g.off - g0 * g.p0 - g1 * g.p1;  // column of the first interior cell
        for (long t = blockIdx.x * (long)blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
            long r = t;
            const long k = r % m2; r /= m2;
            const long j = r % m1; r /= m1;
            const long i = r % m0;
            const long comp = r / m0;
            const long e = comp * g.pc + i * g.p0 + j * g.p1 + (lpad - 1) + k;
            if (a.mode == 2) dst[e] = src[t];
            else dst[t] = src[e];
        }
    }
}

static int launch_copy(const pdehip_grid_t *g, int ncomp, const void *src, void *dst, int mode, void *stream)
{
    NGrid n;
    PDEHIP_TRY(norm_grid(g, &n));
    if (!src || !dst) PDEHIP_FAIL(E_VALUE, "layout copy: NULL array pointer");
    if (ncomp < 1) PDEHIP_FAIL(E_VALUE, "ncomp must be >= 1");
    CopyArgs a;
    a.g = dev_grid(n); a.src = src; a.dst = dst; a.ncomp = ncomp; a.mode = mode;
    const unsigned blocks = grid_blocks((long)ncomp * (n.n[0] + 2) * (n.n[1] + 2) * (n.n[2] + 2));
    if (n.dtype == PDEHIP_F64) hipLaunchKernelGGL((layout_copy_kernel<double>), dim3(blocks), dim3(256), 0, as_stream(stream), a);
    else hipLaunchKernelGGL((layout_copy_kernel<float>), dim3(blocks), dim3(256), 0, as_stream(stream), a);
    PDEHIP_HIP(hipGetLastError());
    return 0;
}

// ---- pointwise kernels of the RK steppers ----------------------------------------------------------
struct LinArgs {
    DevGrid g;
    int ncomp, n;
    void *out;
    const void *y;
    const void *k[6];
    double coef[6];
};

template <typename T, int VEC>
__global__ void __launch_bounds__(256) lincomb_kernel(LinArgs a)
{
    typedef typename VecOf<T, VEC>::type V;
    for_each_chunk<VEC>(a.g, a.ncomp, [&](int, long, long, long, long e) {
        // evaluated left to right like `y + b1*k1 + b2*k2 + ...` (runge_kutta.py:135-145)
        double acc[VEC];
        int s = 0;
        if (a.y) {
            const V yv = *(const V *)((const T *)a.y + e);
#pragma unroll
            for (int q = 0; q < VEC; q++) acc[q] = (double)yv[q];
        } else {
            const V k0 = *(const V *)((const T *)a.k[0] + e);
#pragma unroll
            for (int q = 0; q < VEC; q++) acc[q] = a.coef[0] * (double)k0[q];
            s = 1;
        }
        for (int m = s; m < a.n; m++) {
            const V kv = *(const V *)((const T *)a.k[m] + e);
#pragma unroll
            for (int q = 0; q < VEC; q++) acc[q] = acc[q] + a.coef[m] * (double)kv[q];
        }
        V o;
#pragma unroll
        for (int q = 0; q < VEC; q++) o[q] = (T)acc[q];
        *(V *)((T *)a.out + e) = o;
    });
}

struct Rk4Args {
    DevGrid g;
    int ncomp;
    void *y;
    const void *k1, *k2, *k3, *k4;
};
template <typename T, int VEC>
__global__ void __launch_bounds__(256) rk4_combine_kernel(Rk4Args a)
{
    typedef typename VecOf<T, VEC>::type V;
    for_each_chunk<VEC>(a.g, a.ncomp, [&](int, long, long, long, long e) {
        const V y = *(const V *)((const T *)a.y + e);
        const V k1 = *(const V *)((const T *)a.k1 + e), k2 = *(const V *)((const T *)a.k2 + e);
        const V k3 = *(const V *)((const T *)a.k3 + e), k4 = *(const V *)((const T *)a.k4 + e);
        V o;
#pragma unroll
        for (int q = 0; q < VEC; q++) {
            // pde/solvers/runge_kutta.py:60
            const double s = ((double)k1[q] + 2 * (double)k2[q] + 2 * (double)k3[q] + (double)k4[q]) / 6;
            o[q] = (T)((double)y[q] + s);
        }
        *(V *)((T *)a.y + e) = o;
    });
}

struct Ab2Args {
    DevGrid g;
    int ncomp;
    void *y;
    const void *rc, *rp;
    double dt;
};
template <typename T, int VEC>
__global__ void __launch_bounds__(256) ab2_combine_kernel(Ab2Args a)
{
    typedef typename VecOf<T, VEC>::type V;
    for_each_chunk<VEC>(a.g, a.ncomp, [&](int, long, long, long, long e) {
        const V y = *(const V *)((const T *)a.y + e);
        const V rc = *(const V *)((const T *)a.rc + e), rp = *(const V *)((const T *)a.rp + e);
        V o;
#pragma unroll
        for (int q = 0; q < VEC; q++) {
            // pde/solvers/adams_bashforth.py:44
            const double s = a.dt * (1.5 * (double)rc[q] - 0.5 * (double)rp[q]);
            o[q] = (T)((double)y[q] + s);
        }
        *(V *)((T *)a.y + e) = o;
    });
}

// block-wide max of non-negative doubles (NaN propagates: its bit pattern is the largest)
__device__ __forceinline__ void block_max_to(double v, double *dst)
{
    unsigned long long b = (unsigned long long)__double_as_longlong(v);
#pragma unroll
    for (int ofs = 32; ofs >= 1; ofs >>= 1) {
        const unsigned long long o = __shfl_xor(b, ofs, 64);
        b = (o > b) ? o : b;
    }
    __shared__ unsigned long long part[4];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) part[w] = b;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long m = part[0];
        for (int q = 1; q < (int)(blockDim.x >> 6); q++) m = (part[q] > m) ? part[q] : m;
        // 16384 workgroups hitting one address serialise in the L2 atomic unit (~10 ns each: 200 us for a 256^3 fp32
        // field whose data moves in 100 us).  The cell only grows, so a (possibly stale) read that is already >= m
        // makes the atomic unnecessary; after the first few workgroups almost all of them are.
        if (m > *(volatile unsigned long long *)dst) atomicMax((unsigned long long *)dst, m);
    }
}

struct Rkf45Args {
    DevGrid g;
    int ncomp;
    const void *y;
    void *ynew;
    const void *k[6];
    double *err;
};
template <typename T, int VEC>
__global__ void __launch_bounds__(256) rkf45_combine_kernel(Rkf45Args a)
{
    typedef typename VecOf<T, VEC>::type V;
    // pde/solvers/runge_kutta.py:117-125 (same quotients as the reference source)
    const double r1 = 1.0 / 360, r3 = -128.0 / 4275, r4 = -2197.0 / 75240, r5 = 1.0 / 50, r6 = 2.0 / 55;
    const double c1 = 25.0 / 216, c3 = 1408.0 / 2565, c4 = 2197.0 / 4104, c5 = -1.0 / 5;
    double emax = 0;
    for_each_chunk<VEC>(a.g, a.ncomp, [&](int, long, long, long, long e) {
        const V y = *(const V *)((const T *)a.y + e);
        const V k1 = *(const V *)((const T *)a.k[0] + e), k3 = *(const V *)((const T *)a.k[2] + e);
        const V k4 = *(const V *)((const T *)a.k[3] + e), k5 = *(const V *)((const T *)a.k[4] + e);
        const V k6 = *(const V *)((const T *)a.k[5] + e);
        V o;
#pragma unroll
        for (int q = 0; q < VEC; q++) {
            const double d1 = k1[q], d3 = k3[q], d4 = k4[q], d5 = k5[q], d6 = k6[q];
            const double el = r1 * d1 + r3 * d3 + r4 * d4 + r5 * d5 + r6 * d6;   // :147
            emax = max_nan(emax, abs_nan_canon(el));                              // :148
            o[q] = (T)((double)y[q] + c1 * d1 + c3 * d3 + c4 * d4 + c5 * d5);    // :150
        }
        *(V *)((T *)a.ynew + e) = o;
    });
    block_max_to(emax, a.err);
}

// end of an adaptive Euler attempt (pde/backends/numba/_solvers.py:381-394; StageFuse kind 4 is the same arithmetic inside a sweep):
//   out = half + k   (`step_small += 0.5 * dt * rate_midpoint`, k = that product),   *err = max |(y + dt * rate) - out|
struct EulerAdaptArgs {
    DevGrid g;
    int ncomp;
    const void *y, *rate, *half, *k;
    void *out;
    double dt;
    double *err;
};
template <typename T, int VEC>
__global__ void __launch_bounds__(256) euler_adaptive_combine_kernel(EulerAdaptArgs a)
{
    typedef typename VecOf<T, VEC>::type V;
    double emax = 0;
    for_each_chunk<VEC>(a.g, a.ncomp, [&](int, long, long, long, long e) {
        const V y = *(const V *)((const T *)a.y + e), r = *(const V *)((const T *)a.rate + e);
        const V h = *(const V *)((const T *)a.half + e), k = *(const V *)((const T *)a.k + e);
        V o;
#pragma unroll
        for (int q = 0; q < VEC; q++) {
            o[q] = (T)((double)h[q] + (double)k[q]);                       // :391
            const double large = (double)(T)((double)y[q] + a.dt * (double)r[q]);   // :381 (an array of the state's type in the reference)
            emax = max_nan(emax, abs_nan_canon(large - (double)o[q]));     // :394
        }
        *(V *)((T *)a.out + e) = o;
    });
    block_max_to(emax, a.err);
}

struct DiffArgs {
    DevGrid g;
    int ncomp;
    const void *a, *b;
    double *err;
};
template <typename T, int VEC>
__global__ void __launch_bounds__(256) max_abs_diff_kernel(DiffArgs a)
{
    typedef typename VecOf<T, VEC>::type V;
    double emax = 0;
    for_each_chunk<VEC>(a.g, a.ncomp, [&](int, long, long, long, long e) {
        const V x = *(const V *)((const T *)a.a + e), y = *(const V *)((const T *)a.b + e);
#pragma unroll
        for (int q = 0; q < VEC; q++) emax = max_nan(emax, abs_nan_canon((double)x[q] - (double)y[q]));
    });
    block_max_to(emax, a.err);
}

// max |z| over the cells of complex data held as pairs of real components (2p = real part, 2p + 1 = imaginary part): the error norm
// `np.abs(err).max()` of the adaptive schemes for complex states (pde/solvers/runge_kutta.py:147-148, pde/solvers/euler.py:253);
// hypot like numpy's absolute value of a complex number
template <typename T, int VEC>
__global__ void __launch_bounds__(256) max_abs_pairs_kernel(DiffArgs a)
{
    double emax = 0;
    for_each_chunk<1>(a.g, a.ncomp, [&](int comp, long, long, long, long e) {
        const T *p = (const T *)a.a + e + (long)comp * a.g.pc;   // (e already carries comp * pc: component 2 * comp)
        const double re = (double)p[0], im = (double)p[a.g.pc];
        emax = max_nan(emax, abs_nan_canon(hypot(re, im)));   // (hypot: inf wins over nan, like numpy's |z|)
    });
    block_max_to(emax, a.err);
}

template <typename T> static constexpr int vec_of() { return 16 / sizeof(T); }

#define PDEHIP_VEC_LAUNCH(KERNEL, ARGS, ITEMS)                                                         \
    do {                                                                                               \
        hipStream_t _st = as_stream(stream);                                                           \
        if (n.dtype == PDEHIP_F64) {                                                                   \
            if (n.n[2] % 2 == 0) hipLaunchKernelGGL((KERNEL<double, 2>), dim3(grid_blocks((ITEMS) / 2)), dim3(256), 0, _st, ARGS); \
            else hipLaunchKernelGGL((KERNEL<double, 1>), dim3(grid_blocks(ITEMS)), dim3(256), 0, _st, ARGS); \
        } else {                                                                                       \
            if (n.n[2] % 4 == 0) hipLaunchKernelGGL((KERNEL<float, 4>), dim3(grid_blocks((ITEMS) / 4)), dim3(256), 0, _st, ARGS); \
            else hipLaunchKernelGGL((KERNEL<float, 1>), dim3(grid_blocks(ITEMS)), dim3(256), 0, _st, ARGS); \
        }                                                                                              \
        PDEHIP_HIP(hipGetLastError());                                                                 \
    } while (0)

}  // namespace pdehip

using namespace pdehip;

namespace pdehip {
// Apply the BCs `in_faces` to `in` and evaluate the stencil (mode LAP_*) into the FULL array `out`.
// Scalar first-order faces are evaluated on the fly inside the stencil kernel (the ghost cells of
// `in` are neither written nor read for them); array-valued / second-order faces are written into
// the ghost cells of `in` by the ghost kernel first (numba/backend.py:501-517 order: BCs, then stencil).
int laplace_with_input_bcs(const pdehip_grid_t *g, void *in, const void *y, void *out, int mode, double s1,
                           double s2, double gamma, const pdehip_bc_face_t *in_faces, void *stream, const StageFuse *stage)
{
    NGrid n;
    PDEHIP_TRY(norm_grid(g, &n));
    if (!in_faces) PDEHIP_FAIL(E_VALUE, "input faces are NULL");
    InputBCs fg;
    memset(&fg, 0, sizeof(fg));
    pdehip_bc_face_t rest[2 * PDEHIP_MAX_DIM];
    int n_rest = 0, n_fused = 0;
    static int fuse_axes = -1;   // PDEHIP_FUSE_AXES: bit per normalised axis (tuning / testing aid)
    if (fuse_axes < 0) { const char *e = getenv("PDEHIP_FUSE_AXES"); fuse_axes = e ? atoi(e) : 7; }
    // (1-D grids: the one-cell-per-thread kernel evaluates its two virtual points itself for the plain epilogues)
    const bool can_fuse = laplace_can_fuse_bcs(n, in, out, mode == LAP_EULER ? y : nullptr) || (n.ndim == 1 && mode <= LAP_CH_MU && !stage);
    for (int a = 0; a < PDEHIP_MAX_DIM; a++)
        for (int side = 0; side < 2; side++) {
            pdehip_bc_face_t &r = rest[2 * a + side];
            if (a >= n.ndim) { memset(&r, 0, sizeof(r)); continue; }
            r = in_faces[2 * a + side];
            if (r.kind == PDEHIP_BC_SKIP) continue;
            const int ax = 3 - n.ndim + a;
            if (can_fuse && ((fuse_axes >> ax) & 1) && r.kind == PDEHIP_BC_ORDER1 && r.flags == 0 && r.index1 >= 0 && r.index1 < n.n[ax]) {
                fg.on[ax][side] = 1; fg.idx[ax][side] = r.index1; fg.c[ax][side] = r.const_v; fg.f[ax][side] = r.factor1;
                r.kind = PDEHIP_BC_SKIP;
                n_fused++;
            } else {
                n_rest++;
            }
        }
    if (n_rest) PDEHIP_TRY(launch_ghosts(n, 1, rest, in, as_stream(stream)));
    return launch_laplace(n, in, out, out_strides(n, PDEHIP_OUT_FULL), mode, s1, s2, gamma, y, as_stream(stream), n_fused ? &fg : nullptr, stage);
}

// faces (grid axes) -> on-the-fly BC table (normalised axes); false when a face is not a scalar first-order condition
static bool faces_to_input_bcs(const NGrid &n, const pdehip_bc_face_t *faces, InputBCs *fg, int first_axis = 0, int xplain = 0)
{
    memset(fg, 0, sizeof(*fg));
    if (xplain > 1) {   // one local face on the slowest axis (first / last slab of a non-periodic axis)
        const int side = xplain == 2 ? 0 : 1, ax = 3 - n.ndim;
        const pdehip_bc_face_t &r = faces[side];
        if (r.kind != PDEHIP_BC_ORDER1 || r.flags != 0 || r.index1 < 0 || r.index1 >= n.n[ax]) return false;
        fg->on[ax][side] = 1; fg->idx[ax][side] = r.index1; fg->c[ax][side] = r.const_v; fg->f[ax][side] = r.factor1;
    }
    for (int a = first_axis; a < n.ndim; a++)
        for (int side = 0; side < 2; side++) {
            const int ax = 3 - n.ndim + a;
            const pdehip_bc_face_t &r = faces[2 * a + side];
            if (r.kind != PDEHIP_BC_ORDER1 || r.flags != 0 || r.index1 < 0 || r.index1 >= n.n[ax]) return false;
            fg->on[ax][side] = 1; fg->idx[ax][side] = r.index1; fg->c[ax][side] = r.const_v; fg->f[ax][side] = r.factor1;
        }
    return true;
}

int euler2_with_input_bcs(const pdehip_grid_t *g, const void *in, void *out, double s1, double s2,
                          const pdehip_bc_face_t *faces, void *stream, bool *done, int xplain, bool dry_run, int ends)
{
    *done = false;
    NGrid n;
    PDEHIP_TRY(norm_grid(g, &n));
    if (!in || !out || !faces) PDEHIP_FAIL(E_VALUE, "euler2: NULL pointer");
    if (n.ndim < 2) return 0;
    InputBCs fg;
    if (!faces_to_input_bcs(n, faces, &fg, xplain ? 1 : 0, xplain)) return 0;
    return launch_euler2(n, in, out, s1, s2, fg, xplain, as_stream(stream), done, dry_run, ends);
}

// Two Euler steps on a BOX of a larger array (fast block decomposition, pdehip_block2_loops.h).  `g_box`: the own cells of the rank;
// `in_ext` / `out_ext`: full arrays of the grid two cells larger along the first two axes (own cell (0, 0, 0) = its interior cell
// (1, 1, 0)), i.e. two halo planes / rows on either side; along the fastest axis the two halo cells sit in the row padding.  Cut axes
// read those halos as they are ("plain"), the others must be periodic and wrap inside the kernel.
int euler2_box(const pdehip_grid_t *g_box, const pdehip_bc_face_t *faces, const int *cut3, const void *in_ext, void *out_ext, double s1,
               double s2, void *stream, bool *done, bool dry_run, const long *lo3, const long *n3)
{
    *done = false;
    if (!g_box || !faces || !cut3 || !in_ext || !out_ext) PDEHIP_FAIL(E_VALUE, "euler2_box: NULL pointer");
    if (g_box->ndim != 3) return 0;
    pdehip_grid_t ge = *g_box;
    ge.shape[0] += 2; ge.shape[1] += 2;
    NGrid ne;
    PDEHIP_TRY(norm_grid(&ge, &ne));
    NGrid nb = ne;
    for (int a = 0; a < 3; a++) nb.n[a] = g_box->shape[a];
    nb.off = ne.off + ne.p[0] + ne.p[1];
    // a part of the box (lo3 / n3: the interior two layers behind the cut faces): its halo cells are own cells - only along cut axes
    if (lo3 && n3) {
        for (int a = 0; a < 3; a++) {
            if (!cut3[a] && (lo3[a] != 0 || n3[a] != nb.n[a])) PDEHIP_FAIL(E_RUNTIME, "internal: a part of a box along an axis that wraps");
            if (lo3[a] < 0 || n3[a] < 1 || lo3[a] + n3[a] > nb.n[a]) return 0;
            nb.off += lo3[a] * ne.p[a];
            nb.n[a] = n3[a];
        }
    }
    InputBCs fg;
    memset(&fg, 0, sizeof(fg));
    for (int a = 0; a < 3; a++) {
        if (cut3[a]) continue;
        for (int side = 0; side < 2; side++) {
            const pdehip_bc_face_t &r = faces[2 * a + side];
            const long want = side ? 0 : nb.n[a] - 1;
            if (r.kind != PDEHIP_BC_ORDER1 || r.flags != 0 || r.index1 != want || r.const_v != 0.0 || r.factor1 != 1.0) return 0;   // periodic only
            fg.on[a][side] = 1; fg.idx[a][side] = want; fg.c[a][side] = 0.0; fg.f[a][side] = 1.0;
        }
    }
    // the two halo cells of a cut fastest axis live in the padding of the rows
    if (cut3[2] && !(ne.lpad >= 2 && ne.p[1] >= ne.lpad + nb.n[2] + 2)) return 0;
    return launch_euler2(nb, in_ext, out_ext, s1, s2, fg, cut3[0] ? 1 : 0, as_stream(stream), done, dry_run, 0, E2_DIFFUSION, nullptr, 0.0,
                         nullptr, nullptr, (cut3[1] ? 1 : 0) | (cut3[2] ? 2 : 0));
}

int euler_multi_2d(const pdehip_grid_t *g, const pdehip_rhs_t *rhs, const void *in, void *out, double dt, int nsteps, void *stream,
                   bool *done)
{
    *done = false;
    NGrid n;
    PDEHIP_TRY(norm_grid(g, &n));
    if (!in || !out || !rhs) PDEHIP_FAIL(E_VALUE, "euler_multi_2d: NULL pointer");
    if (n.ndim != 2) return 0;
    InputBCs fc, fm;
    if (!faces_to_input_bcs(n, rhs->bc_c, &fc)) return 0;
    if (rhs->kind == PDEHIP_RHS_DIFFUSION)
        return launch_tile2d(n, in, out, 0, rhs->param, dt, 0.0, fc, nullptr, nsteps, as_stream(stream), done);
    if (rhs->kind != PDEHIP_RHS_CAHN_HILLIARD || !faces_to_input_bcs(n, rhs->bc_mu, &fm)) return 0;
    return launch_tile2d(n, in, out, 1, 1.0, dt, rhs->param, fc, &fm, nsteps, as_stream(stream), done);
}

int cahn_hilliard_fused(const pdehip_grid_t *g, const void *in, void *out, double gamma, double dt, bool euler,
                        const pdehip_bc_face_t *faces_c, const pdehip_bc_face_t *faces_mu, void *stream, bool *done,
                        int xplain, bool dry_run, const StageFuse *stage)
{
    *done = false;
    NGrid n;
    PDEHIP_TRY(norm_grid(g, &n));
    if (!in || (!out && !(stage && (stage->kind == 1 || stage->kind == 2 || stage->kind == 4))) || !faces_c || !faces_mu) PDEHIP_FAIL(E_VALUE, "cahn_hilliard_fused: NULL pointer");
    if (stage && euler) PDEHIP_FAIL(E_RUNTIME, "internal: a Runge-Kutta stage sweep computes the scaled slope");
    if (n.ndim < 2) return 0;
    InputBCs fc, fm;
    if (!faces_to_input_bcs(n, faces_c, &fc, xplain ? 1 : 0, xplain) || !faces_to_input_bcs(n, faces_mu, &fm, xplain ? 1 : 0, xplain)) return 0;
    // level 2 is `y + s2 * (s1 * lap(mu))` resp. `s2 * (s1 * lap(mu))` with s1 = 1 like the two-kernel path (pdehip_steppers.hip)
    return launch_euler2(n, in, out, 1.0, dt, fc, xplain, as_stream(stream), done, dry_run, 0,
                         stage ? E2_CH_STAGE : (euler ? E2_CH_EULER : E2_CH_SCALED), &fm, gamma, nullptr, stage);
}
}  // namespace pdehip

extern "C" {

int pdehip_diffusion_euler2(const pdehip_grid_t *g, const pdehip_bc_face_t *faces, const void *in_full, void *out_full,
                            double diffusivity, double dt, int *done, void *stream)
{
    if (!done) PDEHIP_FAIL(E_VALUE, "diffusion_euler2: NULL pointer");
    bool d = false;
    *done = 0;
    PDEHIP_TRY(euler2_with_input_bcs(g, in_full, out_full, diffusivity, dt, faces, stream, &d));
    *done = d ? 1 : 0;
    return 0;
}

int pdehip_euler_multi_2d(const pdehip_grid_t *g, const pdehip_rhs_t *rhs, const void *in_full, void *out_full, double dt, int nsteps,
                          int *done, void *stream)
{
    if (!done) PDEHIP_FAIL(E_VALUE, "euler_multi_2d: NULL pointer");
    bool d = false;
    *done = 0;
    PDEHIP_TRY(euler_multi_2d(g, rhs, in_full, out_full, dt, nsteps, stream, &d));
    *done = d ? 1 : 0;
    return 0;
}

int pdehip_diffusion_euler2_slab(const pdehip_grid_t *g_sub, const pdehip_bc_face_t *faces, const void *in_full, void *out_full,
                                 double diffusivity, double dt, int halo_sides, int *done, void *stream)
{
    if (!done) PDEHIP_FAIL(E_VALUE, "diffusion_euler2_slab: NULL pointer");
    if (halo_sides < 1 || halo_sides > 3) PDEHIP_FAIL(E_VALUE, "diffusion_euler2_slab: halo_sides must be 1 (both), 2 (upper) or 3 (lower)");
    bool d = false;
    *done = 0;
    PDEHIP_TRY(euler2_with_input_bcs(g_sub, in_full, out_full, diffusivity, dt, faces, stream, &d, halo_sides, false, 0));
    *done = d ? 1 : 0;
    return 0;
}

int pdehip_cahn_hilliard_fused(const pdehip_grid_t *g, const pdehip_bc_face_t *faces_c, const pdehip_bc_face_t *faces_mu,
                               const void *c_full, void *out_full, double gamma, double dt, int euler, int *done, void *stream)
{
    if (!done) PDEHIP_FAIL(E_VALUE, "cahn_hilliard_fused: NULL pointer");
    bool d = false;
    *done = 0;
    PDEHIP_TRY(cahn_hilliard_fused(g, c_full, out_full, gamma, dt, euler != 0, faces_c, faces_mu, stream, &d));
    *done = d ? 1 : 0;
    return 0;
}

int pdehip_valid_to_full(const pdehip_grid_t *g, int ncomp, const void *valid, void *full, void *stream)
{ return launch_copy(g, ncomp, valid, full, 0, stream); }
int pdehip_full_to_valid(const pdehip_grid_t *g, int ncomp, const void *full, void *valid, void *stream)
{ return launch_copy(g, ncomp, full, valid, 1, stream); }
int pdehip_hostfull_to_full(const pdehip_grid_t *g, int ncomp, const void *hostfull_dev, void *full, void *stream)
{ return launch_copy(g, ncomp, hostfull_dev, full, 2, stream); }
int pdehip_full_to_hostfull(const pdehip_grid_t *g, int ncomp, const void *full, void *hostfull_dev, void *stream)
{ return launch_copy(g, ncomp, full, hostfull_dev, 3, stream); }

int pdehip_set_ghost_cells(const pdehip_grid_t *g, int ncomp, const pdehip_bc_face_t *faces, void *data_full, void *stream)
{
    NGrid n;
    PDEHIP_TRY(norm_grid(g, &n));
    return launch_ghosts(n, ncomp, faces, data_full, as_stream(stream));
}

static int lap_entry(const pdehip_grid_t *g, const void *in, void *out, int layout, int mode, double s1,
                     double s2, double gamma, const void *y, void *stream)
{
    NGrid n;
    PDEHIP_TRY(norm_grid(g, &n));
    if (layout != PDEHIP_OUT_VALID && layout != PDEHIP_OUT_FULL) PDEHIP_FAIL(E_VALUE, "unknown output layout %d", layout);
    return launch_laplace(n, in, out, out_strides(n, layout), mode, s1, s2, gamma, y, as_stream(stream));
}


int pdehip_laplace(const pdehip_grid_t *g, const void *in_full, void *out, int out_layout, void *stream)
{ return lap_entry(g, in_full, out, out_layout, LAP_PLAIN, 0, 0, 0, nullptr, stream); }
int pdehip_laplace_scaled(const pdehip_grid_t *g, const void *in_full, void *out_full, double s1, double s2, void *stream)
{ return lap_entry(g, in_full, out_full, PDEHIP_OUT_FULL, LAP_SCALED, s1, s2, 0, nullptr, stream); }
int pdehip_laplace_euler(const pdehip_grid_t *g, const void *in_full, const void *y_full, void *out_full, double s1, double s2, void *stream)
{ return lap_entry(g, in_full, out_full, PDEHIP_OUT_FULL, LAP_EULER, s1, s2, 0, y_full, stream); }
int pdehip_cahn_hilliard_mu(const pdehip_grid_t *g, const void *c_full, void *mu_full, double gamma, void *stream)
{ return lap_entry(g, c_full, mu_full, PDEHIP_OUT_FULL, LAP_CH_MU, 0, 0, gamma, nullptr, stream); }

int pdehip_axis_derivative(const pdehip_grid_t *g, int axis, int order, int method, const void *in_full, void *out,
                           int out_layout, void *stream)
{
    NGrid n;
    PDEHIP_TRY(norm_grid(g, &n));
    if (!in_full || !out) PDEHIP_FAIL(E_VALUE, "axis_derivative: NULL array pointer");
    if (axis < 0 || axis >= n.ndim) PDEHIP_FAIL(E_VALUE, "axis %d out of range for a %d-dimensional grid", axis, n.ndim);
    if (order != 1 && order != 2) PDEHIP_FAIL(E_VALUE, "derivative order must be 1 or 2 (got %d)", order);
    if (method < 0 || method > 2) PDEHIP_FAIL(E_VALUE, "Unknown derivative type `%d`", method);
    if (out_layout != PDEHIP_OUT_VALID && out_layout != PDEHIP_OUT_FULL) PDEHIP_FAIL(E_VALUE, "unknown output layout %d", out_layout);
    const OutStr o = out_strides(n, out_layout);
    const int ax = 3 - n.ndim + axis;
    AxisArgs a;
    a.g = dev_grid(n); a.in = in_full; a.out = out;
    a.o_off = o.off; a.o_s0 = o.s0; a.o_s1 = o.s1;
    a.pa = n.p[ax]; a.dx = n.dx[ax]; a.order = order; a.method = method;
    const unsigned blocks = grid_blocks(n.n[0] * n.n[1] * n.n[2]);
    if (n.dtype == PDEHIP_F64) hipLaunchKernelGGL((axis_derivative_kernel<double>), dim3(blocks), dim3(256), 0, as_stream(stream), a);
    else hipLaunchKernelGGL((axis_derivative_kernel<float>), dim3(blocks), dim3(256), 0, as_stream(stream), a);
    PDEHIP_HIP(hipGetLastError());
    return 0;
}

}  // extern "C"

// ---- products of tensor fields at every cell (numpy/backend.py:285-363: np.einsum per rank combination) --------------------------
namespace pdehip {
struct ProdArgs {
    const void *a, *b;
    void *out;
    long pc;        // elements of one component (full array)
    int dim, kind, cplx, conj;
};
// kind 0: v.v -> s   1: T.v -> v   2: v.T -> v   3: T.T -> T   4: outer v (x) v -> T.   Complex data: planar (re, im) pairs per tensor
// entry (component index = entry * 2 + part); `conj`: the second operand is conjugated.  Sums run over the contracted index in order.
template <typename T>
__global__ void __launch_bounds__(256) field_product_kernel(ProdArgs p)
{
    const long e = blockIdx.x * 256L + threadIdx.x;
    if (e >= p.pc) return;
    const int d = p.dim, w = p.cplx ? 2 : 1;
    const T *a = (const T *)p.a + e, *b = (const T *)p.b + e;
    T *out = (T *)p.out + e;
    auto term = [&](int ia, int ib, double &re, double &im) {   // += a[ia] * (conj) b[ib]
        const double ar = (double)a[(long)(ia * w) * p.pc], br = (double)b[(long)(ib * w) * p.pc];
        if (!p.cplx) { re = re + ar * br; return; }
        const double ai = (double)a[(long)(ia * w + 1) * p.pc];
        double bi = (double)b[(long)(ib * w + 1) * p.pc];
        if (p.conj) bi = -bi;
        re = re + (ar * br - ai * bi);
        im = im + (ar * bi + ai * br);
    };
    auto put = [&](int io, double re, double im) {
        out[(long)(io * w) * p.pc] = (T)re;
        if (p.cplx) out[(long)(io * w + 1) * p.pc] = (T)im;
    };
    if (p.kind == 0) {
        double re = 0, im = 0;
        for (int i = 0; i < d; i++) term(i, i, re, im);
        put(0, re, im);
    } else if (p.kind == 1) {
        for (int i = 0; i < d; i++) {
            double re = 0, im = 0;
            for (int j = 0; j < d; j++) term(i * d + j, j, re, im);
            put(i, re, im);
        }
    } else if (p.kind == 2) {
        for (int j = 0; j < d; j++) {
            double re = 0, im = 0;
            for (int i = 0; i < d; i++) term(i, i * d + j, re, im);
            put(j, re, im);
        }
    } else if (p.kind == 3) {
        for (int i = 0; i < d; i++)
            for (int k = 0; k < d; k++) {
                double re = 0, im = 0;
                for (int j = 0; j < d; j++) term(i * d + j, j * d + k, re, im);
                put(i * d + k, re, im);
            }
    } else {
        for (int i = 0; i < d; i++)
            for (int j = 0; j < d; j++) {
                double re = 0, im = 0;
                term(i, j, re, im);
                put(i * d + j, re, im);
            }
    }
}
}  // namespace pdehip

extern "C" {

int pdehip_field_product(const pdehip_grid_t *g, int kind, int complex_pairs, int conjugate, const void *a_full, const void *b_full, void *out_full,
                         void *stream)
{
    if (!a_full || !b_full || !out_full) PDEHIP_FAIL(E_VALUE, "field_product: NULL pointer");
    if (kind < 0 || kind > 4) PDEHIP_FAIL(E_VALUE, "field_product: kind 0 (v.v), 1 (T.v), 2 (v.T), 3 (T.T) or 4 (outer)");
    if (out_full == a_full || out_full == b_full) PDEHIP_FAIL(E_VALUE, "field_product: the output must not be an operand");
    NGrid n;
    PDEHIP_TRY(norm_grid(g, &n));
    ProdArgs p = {a_full, b_full, out_full, n.pc, g->ndim, kind, complex_pairs ? 1 : 0, conjugate ? 1 : 0};
    const unsigned blocks = (unsigned)((n.pc + 255) / 256);
    if (n.dtype == PDEHIP_F64) hipLaunchKernelGGL(field_product_kernel<double>, dim3(blocks), dim3(256), 0, as_stream(stream), p);
    else hipLaunchKernelGGL(field_product_kernel<float>, dim3(blocks), dim3(256), 0, as_stream(stream), p);
    PDEHIP_HIP(hipGetLastError());
    return 0;
}

int pdehip_integrate(const pdehip_grid_t *g, int ncomp, const void *arr_full, double cell_volume, double *out_dev, void *stream)
{
    NGrid n;
    PDEHIP_TRY(norm_grid(g, &n));
    if (!arr_full || !out_dev) PDEHIP_FAIL(E_VALUE, "integrate: NULL pointer");
    if (ncomp < 1 || ncomp > 64) PDEHIP_FAIL(E_VALUE, "integrate: 1..64 components");
    static double *partial = nullptr;   // 64 components x 1024 workgroups, allocated once per process
    constexpr int kBlocks = 1024;
    if (!partial) PDEHIP_HIP(hipMalloc(&partial, sizeof(double) * 64 * kBlocks));
    SumArgs a;
    a.g = dev_grid(n); a.in = arr_full; a.partial = partial; a.out = out_dev; a.vol = cell_volume;
    long blocks = (n.n[0] * n.n[1] * n.n[2] + 255) / 256;
    a.nblocks = (int)(blocks < kBlocks ? (blocks < 1 ? 1 : blocks) : kBlocks);
    hipStream_t st = as_stream(stream);
    for (int c = 0; c < ncomp; c++) {
        a.comp = c;
        if (n.dtype == PDEHIP_F64) hipLaunchKernelGGL((partial_sum_kernel<double>), dim3(a.nblocks), dim3(256), 0, st, a);
        else hipLaunchKernelGGL((partial_sum_kernel<float>), dim3(a.nblocks), dim3(256), 0, st, a);
        hipLaunchKernelGGL(final_sum_kernel, dim3(1), dim3(256), 0, st, a);
    }
    PDEHIP_HIP(hipGetLastError());
    return 0;
}

int pdehip_count_nonfinite(const pdehip_grid_t *g, int ncomp, const void *arr_full, double *out_dev, void *stream)
{
    NGrid n;
    PDEHIP_TRY(norm_grid(g, &n));
    if (!arr_full || !out_dev) PDEHIP_FAIL(E_VALUE, "count_nonfinite: NULL pointer");
    if (ncomp < 1 || ncomp > 64) PDEHIP_FAIL(E_VALUE, "count_nonfinite: 1..64 components");
    static double *partial = nullptr;
    constexpr int kBlocks = 1024;
    if (!partial) PDEHIP_HIP(hipMalloc(&partial, sizeof(double) * 64 * kBlocks));
    SumArgs a;
    a.g = dev_grid(n); a.in = arr_full; a.partial = partial; a.out = out_dev; a.vol = 1.0;
    long blocks = (n.n[0] * n.n[1] * n.n[2] + 255) / 256;
    a.nblocks = (int)(blocks < kBlocks ? (blocks < 1 ? 1 : blocks) : kBlocks);
    hipStream_t st = as_stream(stream);
    for (int c = 0; c < ncomp; c++) {
        a.comp = c;
        if (n.dtype == PDEHIP_F64) hipLaunchKernelGGL((partial_nonfinite_kernel<double>), dim3(a.nblocks), dim3(256), 0, st, a);
        else hipLaunchKernelGGL((partial_nonfinite_kernel<float>), dim3(a.nblocks), dim3(256), 0, st, a);
        hipLaunchKernelGGL(final_sum_kernel, dim3(1), dim3(256), 0, st, a);
    }
    PDEHIP_HIP(hipGetLastError());
    return 0;
}

int pdehip_add_gaussian_noise(const pdehip_grid_t *g, int ncomp, void *y_full, double scale, uint64_t seed, uint64_t counter,
                              uint64_t cell_offset, void *stream)
{
    NGrid n;
    PDEHIP_TRY(norm_grid(g, &n));
    if (!y_full) PDEHIP_FAIL(E_VALUE, "add_gaussian_noise: NULL pointer");
    if (ncomp < 1) PDEHIP_FAIL(E_VALUE, "add_gaussian_noise: ncomp must be positive");
    NoiseArgs a;
    a.g = dev_grid(n); a.y = y_full; a.ncomp = ncomp; a.scale = scale; a.seed = seed; a.counter = counter; a.cell_offset = cell_offset;
    const unsigned blocks = grid_blocks((long)ncomp * n.n[0] * n.n[1] * n.n[2]);
    if (n.dtype == PDEHIP_F64) hipLaunchKernelGGL((gaussian_noise_kernel<double>), dim3(blocks), dim3(256), 0, as_stream(stream), a);
    else hipLaunchKernelGGL((gaussian_noise_kernel<float>), dim3(blocks), dim3(256), 0, as_stream(stream), a);
    PDEHIP_HIP(hipGetLastError());
    return 0;
}

int pdehip_laplace9(const pdehip_grid_t *g, const int *periodic2, double corner_weight, void *in_full, void *out, int out_layout,
                    void *stream)
{
    NGrid n;
    PDEHIP_TRY(norm_grid(g, &n));
    if (!in_full || !out || !periodic2) PDEHIP_FAIL(E_VALUE, "laplace9: NULL pointer");
    if (n.ndim != 2) PDEHIP_FAIL(E_VALUE, "the nine-point stencil is defined for 2-D grids (got %d axes)", n.ndim);
    if (out_layout != PDEHIP_OUT_VALID && out_layout != PDEHIP_OUT_FULL) PDEHIP_FAIL(E_VALUE, "unknown output layout %d", out_layout);
    const OutStr o = out_strides(n, out_layout);
    Lap9Args a;
    a.in = in_full; a.out = out;
    a.p1 = n.p[1]; a.off = n.off; a.nx = n.n[1]; a.ny = n.n[2]; a.o_off = o.off; a.o_s1 = o.s1;
    a.per_x = periodic2[0] != 0; a.per_y = periodic2[1] != 0;
    const double w = corner_weight, dxm2 = n.lap_scale[1], dym2 = n.lap_scale[2], dm2 = dxm2 + dym2;
    const double st[3][3] = {{0.25 * dm2 * w, dxm2 * (1 - w), 0.25 * dm2 * w},
                             {dym2 * (1 - w), (dxm2 + dym2) * (w - 2), dym2 * (1 - w)},
                             {0.25 * dm2 * w, dxm2 * (1 - w), 0.25 * dm2 * w}};
    memcpy(a.st, st, sizeof(st));
    hipStream_t s = as_stream(stream);
    const dim3 grid((unsigned)(((a.ny + 255) / 256) * a.nx));
    if (n.dtype == PDEHIP_F64) {
        hipLaunchKernelGGL((corner_points_kernel<double>), dim3(1), dim3(64), 0, s, a);
        hipLaunchKernelGGL((laplace9_kernel<double>), grid, dim3(256), 0, s, a);
    } else {
        hipLaunchKernelGGL((corner_points_kernel<float>), dim3(1), dim3(64), 0, s, a);
        hipLaunchKernelGGL((laplace9_kernel<float>), grid, dim3(256), 0, s, a);
    }
    PDEHIP_HIP(hipGetLastError());
    return 0;
}

int pdehip_gradient(const pdehip_grid_t *g, int method, const void *in_full, void *out, int out_layout, void *stream)
{ return launch_deriv(0, g, method, in_full, out, out_layout, stream); }
int pdehip_divergence(const pdehip_grid_t *g, int method, const void *in_full, void *out, int out_layout, void *stream)
{ return launch_deriv(1, g, method, in_full, out, out_layout, stream); }
int pdehip_gradient_squared(const pdehip_grid_t *g, int central, const void *in_full, void *out, int out_layout, void *stream)
{ return launch_deriv(2, g, central ? 1 : 0, in_full, out, out_layout, stream); }

int pdehip_lincomb(const pdehip_grid_t *g, int ncomp, void *out_full, const void *y_full, int nk,
                   const double *coef_host, const void *const *k_full_host, void *stream)
{
    NGrid n;
    PDEHIP_TRY(norm_grid(g, &n));
    if (nk < 1 || nk > 6) PDEHIP_FAIL(E_VALUE, "lincomb supports 1..6 terms (got %d)", nk);
    if (!out_full || !coef_host || !k_full_host) PDEHIP_FAIL(E_VALUE, "lincomb: NULL pointer");
    LinArgs a;
    a.g = dev_grid(n); a.ncomp = ncomp; a.n = nk; a.out = out_full; a.y = y_full;
    for (int q = 0; q < 6; q++) { a.k[q] = q < nk ? k_full_host[q] : nullptr; a.coef[q] = q < nk ? coef_host[q] : 0; }
    const long items = (long)ncomp * n.n[0] * n.n[1] * n.n[2];
    PDEHIP_VEC_LAUNCH(lincomb_kernel, a, items);
    return 0;
}

int pdehip_rk4_combine(const pdehip_grid_t *g, int ncomp, void *y_full, const void *k1, const void *k2,
                       const void *k3, const void *k4, void *stream)
{
    NGrid n;
    PDEHIP_TRY(norm_grid(g, &n));
    if (!y_full || !k1 || !k2 || !k3 || !k4) PDEHIP_FAIL(E_VALUE, "rk4_combine: NULL pointer");
    Rk4Args a;
    a.g = dev_grid(n); a.ncomp = ncomp; a.y = y_full; a.k1 = k1; a.k2 = k2; a.k3 = k3; a.k4 = k4;
    const long items = (long)ncomp * n.n[0] * n.n[1] * n.n[2];
    PDEHIP_VEC_LAUNCH(rk4_combine_kernel, a, items);
    return 0;
}

int pdehip_ab2_combine(const pdehip_grid_t *g, int ncomp, void *y_full, const void *rate_cur, const void *rate_prev,
                       double dt, void *stream)
{
    NGrid n;
    PDEHIP_TRY(norm_grid(g, &n));
    if (!y_full || !rate_cur || !rate_prev) PDEHIP_FAIL(E_VALUE, "ab2_combine: NULL pointer");
    Ab2Args a;
    a.g = dev_grid(n); a.ncomp = ncomp; a.y = y_full; a.rc = rate_cur; a.rp = rate_prev; a.dt = dt;
    const long items = (long)ncomp * n.n[0] * n.n[1] * n.n[2];
    PDEHIP_VEC_LAUNCH(ab2_combine_kernel, a, items);
    return 0;
}

int pdehip_euler_adaptive_combine(const pdehip_grid_t *g, int ncomp, const void *y_full, const void *rate_full, double dt, const void *half_full,
                                  const void *k_full, void *out_full, double *err_dev, void *stream)
{
    NGrid n;
    PDEHIP_TRY(norm_grid(g, &n));
    if (!y_full || !rate_full || !half_full || !k_full || !out_full || !err_dev) PDEHIP_FAIL(E_VALUE, "euler_adaptive_combine: NULL pointer");
    if (ncomp < 1) PDEHIP_FAIL(E_VALUE, "ncomp must be >= 1");
    PDEHIP_HIP(hipMemsetAsync(err_dev, 0, sizeof(double), as_stream(stream)));
    EulerAdaptArgs a;
    a.g = dev_grid(n); a.ncomp = ncomp; a.y = y_full; a.rate = rate_full; a.half = half_full; a.k = k_full; a.out = out_full; a.dt = dt; a.err = err_dev;
    const long items = (long)ncomp * n.n[0] * n.n[1] * n.n[2];
    PDEHIP_VEC_LAUNCH(euler_adaptive_combine_kernel, a, items);
    return 0;
}

int pdehip_max_abs_pairs(const pdehip_grid_t *g, int npairs, const void *arr_full, double *out_dev, void *stream)
{
    NGrid n;
    PDEHIP_TRY(norm_grid(g, &n));
    if (!arr_full || !out_dev || npairs < 1) PDEHIP_FAIL(E_VALUE, "max_abs_pairs: NULL pointer or no pairs");
    PDEHIP_HIP(hipMemsetAsync(out_dev, 0, sizeof(double), as_stream(stream)));
    DiffArgs a;
    a.g = dev_grid(n); a.ncomp = npairs; a.a = arr_full; a.b = nullptr; a.err = out_dev;
    const long items = (long)npairs * n.n[0] * n.n[1] * n.n[2];
    hipStream_t st = as_stream(stream);
    if (n.dtype == PDEHIP_F64) hipLaunchKernelGGL((max_abs_pairs_kernel<double, 1>), dim3(grid_blocks(items)), dim3(256), 0, st, a);
    else hipLaunchKernelGGL((max_abs_pairs_kernel<float, 1>), dim3(grid_blocks(items)), dim3(256), 0, st, a);
    PDEHIP_HIP(hipGetLastError());
    return 0;
}

int pdehip_rkf45_combine(const pdehip_grid_t *g, int ncomp, const void *y, void *ynew,
                         const void *const *k6_host, double *err_dev, void *stream)
{
    NGrid n;
    PDEHIP_TRY(norm_grid(g, &n));
    if (!y || !ynew || !k6_host || !err_dev) PDEHIP_FAIL(E_VALUE, "rkf45_combine: NULL pointer");
    PDEHIP_HIP(hipMemsetAsync(err_dev, 0, sizeof(double), as_stream(stream)));
    Rkf45Args a;
    a.g = dev_grid(n); a.ncomp = ncomp; a.y = y; a.ynew = ynew; a.err = err_dev;
    for (int q = 0; q < 6; q++) a.k[q] = k6_host[q];
    const long items = (long)ncomp * n.n[0] * n.n[1] * n.n[2];
    PDEHIP_VEC_LAUNCH(rkf45_combine_kernel, a, items);
    return 0;
}

int pdehip_max_abs_diff(const pdehip_grid_t *g, int ncomp, const void *a_full, const void *b_full,
                        double *out_dev, void *stream)
{
    NGrid n;
    PDEHIP_TRY(norm_grid(g, &n));
    if (!a_full || !b_full || !out_dev) PDEHIP_FAIL(E_VALUE, "max_abs_diff: NULL pointer");
    PDEHIP_HIP(hipMemsetAsync(out_dev, 0, sizeof(double), as_stream(stream)));
    DiffArgs a;
    a.g = dev_grid(n); a.ncomp = ncomp; a.a = a_full; a.b = b_full; a.err = out_dev;
    const long items = (long)ncomp * n.n[0] * n.n[1] * n.n[2];
    PDEHIP_VEC_LAUNCH(max_abs_diff_kernel, a, items);
    return 0;
}

}  // extern "C"
