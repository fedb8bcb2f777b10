// microbench6.hip — access ORDER of the 3-D fp64 Laplacian on MI355X (tool, not product; VERDICT r4 "next" #2 / #3).
// Round 3 found that a copy reaches 6.2-6.55 TB/s only when short-lived workgroups sweep ONE contiguous window in address order and
// that the plane march of lap_march_kernel (thousands of long-lived waves, each hopping 2 MiB per plane, 8 x-chunks = 8 fronts) is
// bound to 5.4-5.8 as a pure copy.  Question here: does a stencil with ONE front reach more than the march?
//   slab  R,P : short-lived workgroups (256 threads = one 512-cell row of 16-byte vectors), each computes P planes x R rows; the planes
//               i-1 .. i+P it reads come from the XCD's L2 except the newest one (block -> XCD b % 8 owns a fixed range of rows, so the
//               8 XCDs move through the planes in step: one read front, one write front)
//   march1 RY,PF,nxc : long-lived waves like the product, but narrow tiles (RY rows x one 128-cell chunk) so that 2048-4096 waves fit
//               into ONE x-chunk (nxc = 1: a single front), PF planes of prefetch
//   copy      : the simple copy (ceiling) and the product-like march copy
// every stencil variant is checked bit for bit against a one-cell-per-thread kernel with the reference's expression order.
// build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o tools/microbench6 tools/microbench6.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x)                                                                                                        \
    do {                                                                                                             \
        hipError_t e = (x);                                                                                          \
        if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } \
    } while (0)

typedef double d2 __attribute__((ext_vector_type(2)));

struct G {
    long n0, n1, n2;   // cells per axis (n2 = 512: one workgroup row)
    long p0, p1, off;  // plane pitch, row pitch, index of interior cell (0, 0, 0); the arrays are ghost-padded
    double sx, sy, sz;
};

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

template <bool NT> __device__ __forceinline__ void stv(double *p, d2 v)
{
    if (NT) __builtin_nontemporal_store(v, (d2 *)p);
    else *(d2 *)p = v;
}

// pde/backends/numba/operators/cartesian.py:220-227
__device__ __forceinline__ double lap7(double c, double xm, double xp, double ym, double yp, double zm, double zp, const G &g)
{
    const double vm = 2 * c;
    const double lx = (xm - vm + xp) * g.sx;
    const double ly = (ym - vm + yp) * g.sy;
    const double lz = (zm - vm + zp) * g.sz;
    return lx + ly + lz;
}

__global__ void __launch_bounds__(256) k_ref(const double *in, double *out, G g)
{
    const long t = blockIdx.x * 256L + threadIdx.x;
    if (t >= g.n0 * g.n1 * g.n2) return;
    const long k = t % g.n2, j = (t / g.n2) % g.n1, i = t / (g.n2 * g.n1);
    const double *c = in + g.off + i * g.p0 + j * g.p1 + k;
    out[g.off + i * g.p0 + j * g.p1 + k] = lap7(c[0], c[-g.p0], c[g.p0], c[-g.p1], c[g.p1], c[-1], c[1], g);
}

// ---------------------------------------------------------------------------------------------------------------------------------
// slab: one workgroup = P planes x R rows x the whole 512-cell row
// ---------------------------------------------------------------------------------------------------------------------------------
template <int R, int P, bool NT, bool SWZ>
__global__ void __launch_bounds__(256) k_slab(const double *__restrict__ in, double *__restrict__ out, G g)
{
    const long b = blockIdx.x;
    const long tiles = g.n1 / R;   // row tiles per plane
    long pi, tile;
    if (SWZ) {
        const long xcd = b & 7, q = b >> 3, tpx = tiles / 8;
        pi = q / tpx;
        tile = xcd * tpx + q % tpx;
    } else {
        pi = b / tiles;
        tile = b % tiles;
    }
    const long i0 = pi * P, j0 = tile * R;
    const int t = threadIdx.x, lane = t & 63;
    const double *base = in + g.off + i0 * g.p0 + j0 * g.p1 + 2 * t;
    // the cell left of the wave's chunk (lanes < 32) or right of it (lanes >= 32): one broadcast load per row
    const long zh_rel = (lane < 32) ? -(long)(2 * lane) - 1 : (long)(2 * (63 - lane)) + 2;

    d2 pl[P + 2][R + 2];
    double zh[P][R];
#pragma unroll
    for (int p = 0; p < P + 2; p++) {
        const bool inner = (p >= 1 && p <= P);
#pragma unroll
        for (int r = 0; r < R + 2; r++) {
            if (!inner && (r == 0 || r == R + 1)) continue;   // outer planes: centre rows only
            pl[p][r] = *(const d2 *)(base + (long)(p - 1) * g.p0 + (long)(r - 1) * g.p1);
        }
    }
#pragma unroll
    for (int p = 0; p < P; p++)
#pragma unroll
        for (int r = 0; r < R; r++) zh[p][r] = (base + (long)p * g.p0 + (long)r * g.p1)[zh_rel];

    double *ob = out + g.off + i0 * g.p0 + j0 * g.p1 + 2 * t;
#pragma unroll
    for (int p = 0; p < P; p++)
#pragma unroll
        for (int r = 0; r < R; r++) {
            const d2 cc = pl[p + 1][r + 1];
            const double zl = wave_shr1(zh[p][r], cc[1]);
            const double zr = wave_shl1(zh[p][r], cc[0]);
            d2 res;
            res[0] = lap7(cc[0], pl[p][r + 1][0], pl[p + 2][r + 1][0], pl[p + 1][r][0], pl[p + 1][r + 2][0], zl, cc[1], g);
            res[1] = lap7(cc[1], pl[p][r + 1][1], pl[p + 2][r + 1][1], pl[p + 1][r][1], pl[p + 1][r + 2][1], cc[0], zr, g);
            stv<NT>(ob + (long)p * g.p0 + (long)r * g.p1, res);
        }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// march1: a wave owns RY rows x one 128-cell chunk and marches lx planes with PF planes of prefetch; workgroup = the 4 chunks of the rows
// ---------------------------------------------------------------------------------------------------------------------------------
template <int RY, int PF, bool NT>
__global__ void __launch_bounds__(256) k_march1(const double *__restrict__ in, double *__restrict__ out, G g, int lx)
{
    const long tiles = g.n1 / RY;           // workgroups per x-chunk
    const long nb = gridDim.x;
    // XCD b % 8 gets a contiguous range of the (x-chunk, row tile) list
    const long per = nb / 8;
    long bid = blockIdx.x;
    if (bid < per * 8) bid = (bid % 8) * per + bid / 8;
    const long xc = bid / tiles, tile = bid % tiles;
    const int t = threadIdx.x, lane = t & 63;
    const long i0 = xc * lx, i1 = (i0 + lx < g.n0) ? i0 + lx : g.n0;
    const long j0 = tile * RY;
    const double *base = in + g.off + j0 * g.p1 + 2 * t;
    const long zh_rel = (lane < 32) ? -(long)(2 * lane) - 1 : (long)(2 * (63 - lane)) + 2;

    d2 prev[RY];
    d2 pl[PF + 1][RY + 2];
    double zh[PF + 1][RY];
    auto load_plane = [&](long i, int slot) {
        const double *p = base + i * g.p0;
#pragma unroll
        for (int r = 0; r < RY + 2; r++) pl[slot][r] = *(const d2 *)(p + (long)(r - 1) * g.p1);
#pragma unroll
        for (int r = 0; r < RY; r++) zh[slot][r] = (p + (long)r * g.p1)[zh_rel];
    };
#pragma unroll
    for (int r = 0; r < RY; r++) prev[r] = *(const d2 *)(base + (i0 - 1) * g.p0 + (long)r * g.p1);
#pragma unroll
    for (int q = 0; q < PF; q++) load_plane((i0 + q > g.n0) ? g.n0 : i0 + q, q);

    double *ob = out + g.off + j0 * g.p1 + 2 * t;
    for (long i = i0; i < i1; i++) {
        load_plane((i + PF > g.n0) ? g.n0 : i + PF, PF);
#pragma unroll
        for (int r = 0; r < RY; r++) {
            const d2 cc = pl[0][r + 1];
            const double zl = wave_shr1(zh[0][r], cc[1]);
            const double zr = wave_shl1(zh[0][r], cc[0]);
            d2 res;
            res[0] = lap7(cc[0], prev[r][0], pl[1][r + 1][0], pl[0][r][0], pl[0][r + 2][0], zl, cc[1], g);
            res[1] = lap7(cc[1], prev[r][1], pl[1][r + 1][1], pl[0][r][1], pl[0][r + 2][1], cc[0], zr, g);
            stv<NT>(ob + i * g.p0 + (long)r * g.p1, res);
        }
#pragma unroll
        for (int r = 0; r < RY; r++) prev[r] = pl[0][r + 1];
#pragma unroll
        for (int q = 0; q < PF; q++) {
#pragma unroll
            for (int r = 0; r < RY + 2; r++) pl[q][r] = pl[q + 1][r];
#pragma unroll
            for (int r = 0; r < RY; r++) zh[q][r] = zh[q + 1][r];
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// copies in the same layout (interior rows only): the ceiling of each access order
// ---------------------------------------------------------------------------------------------------------------------------------
template <int R, bool NT, bool SWZ> __global__ void __launch_bounds__(256) k_copy_slab(const double *__restrict__ in, double *__restrict__ out, G g)
{
    const long b = blockIdx.x;
    const long tiles = g.n1 / R;
    long pi, tile;
    if (SWZ) {
        const long xcd = b & 7, q = b >> 3, tpx = tiles / 8;
        pi = q / tpx;
        tile = xcd * tpx + q % tpx;
    } else {
        pi = b / tiles;
        tile = b % tiles;
    }
    const long o = g.off + pi * g.p0 + tile * R * g.p1 + 2 * threadIdx.x;
    d2 v[R];
#pragma unroll
    for (int r = 0; r < R; r++) v[r] = *(const d2 *)(in + o + (long)r * g.p1);
#pragma unroll
    for (int r = 0; r < R; r++) stv<NT>(out + o + (long)r * g.p1, v[r]);
}

template <bool NT> __global__ void __launch_bounds__(256) k_copy_simple(const d2 *__restrict__ in, d2 *__restrict__ out, long n)
{
    const long i = blockIdx.x * 256L + threadIdx.x;
    if (i < n) {
        if (NT) __builtin_nontemporal_store(in[i], out + i);
        else out[i] = in[i];
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
static double *d_in, *d_out, *d_ref;
static G g;
static long total;
static hipEvent_t e0, e1;

template <typename F> static float time_it(F launch, int reps = 20)
{
    for (int i = 0; i < 3; i++) launch();
    CK(hipDeviceSynchronize());
    float best = 1e9f;
    for (int s = 0; s < 3; s++) {
        CK(hipEventRecord(e0));
        for (int i = 0; i < reps; i++) launch();
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms / reps < best) best = ms / reps;
    }
    return best;
}

static long mismatches()
{
    static std::vector<double> a, b;
    a.resize(total);
    b.resize(total);
    CK(hipMemcpy(a.data(), d_out, total * 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(b.data(), d_ref, total * 8, hipMemcpyDeviceToHost));
    long bad = 0;
    for (long i = 0; i < g.n0; i++)
        for (long j = 0; j < g.n1; j++)
            bad += memcmp(&a[g.off + i * g.p0 + j * g.p1], &b[g.off + i * g.p0 + j * g.p1], g.n2 * 8) != 0;
    return bad;
}

template <typename F> static void run(const char *name, F launch, bool check = true)
{
    long bad = -1;
    if (check) {
        CK(hipMemset(d_out, 0xff, total * 8));
        launch();
        CK(hipDeviceSynchronize());
        bad = mismatches();
    }
    const float ms = time_it(launch);
    const double gb = (double)g.n0 * g.n1 * g.n2 * 16 / 1e9;
    printf("%-34s %8.4f ms  %6.0f GB/s  %.3f of 8 TB/s  %s\n", name, ms, gb / ms * 1e3, gb / ms * 1e3 / 8000,
           check ? (bad == 0 ? "bit-exact" : "MISMATCH") : "");
    fflush(stdout);
}

int main(int argc, char **argv)
{
    const long n = argc > 1 ? atol(argv[1]) : 512;
    g.n0 = n; g.n1 = n; g.n2 = 512;
    // row layout: A = 2 is the product's (first interior cell 16-byte aligned, pitch n2 + 4); A = 16: the first interior cell of EVERY row
    // on a 128-byte line (pitch n2 + 16: the upper ghost cell of a row is element 0 of the next row's segment, the lower one element 15)
    const long A = argc > 2 ? atol(argv[2]) : 2;
    g.p1 = g.n2 + (A == 2 ? 4 : A);
    g.p0 = (g.n1 + 2) * g.p1;
    g.off = g.p0 + g.p1 + A;
    g.sx = 1.0; g.sy = 1.0; g.sz = 1.0;
    total = (g.n0 + 2) * g.p0 + 64;
    CK(hipMalloc(&d_in, total * 8));
    CK(hipMalloc(&d_out, total * 8));
    CK(hipMalloc(&d_ref, total * 8));
    {
        std::vector<double> h(total);
        unsigned long long s = 88172645463325252ULL;
        for (long i = 0; i < total; i++) {
            s ^= s << 13; s ^= s >> 7; s ^= s << 17;
            h[i] = (double)(s >> 11) / 9007199254740992.0;
        }
        CK(hipMemcpy(d_in, h.data(), total * 8, hipMemcpyHostToDevice));
    }
    CK(hipMemset(d_ref, 0xff, total * 8));
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const long cells = g.n0 * g.n1 * g.n2;
    hipLaunchKernelGGL(k_ref, dim3((unsigned)((cells + 255) / 256)), dim3(256), 0, 0, d_in, d_ref, g);
    CK(hipDeviceSynchronize());
    printf("# Laplacian %ld x %ld x %ld fp64, ghost-padded arrays (pitch %ld), algorithmic bytes 16 B/cell\n", g.n0, g.n1, g.n2, g.p1);

    // ceilings
    {
        const long nv = total / 2;
        run("copy simple plain", [&] { hipLaunchKernelGGL(k_copy_simple<false>, dim3((unsigned)((nv + 255) / 256)), dim3(256), 0, 0, (const d2 *)d_in, (d2 *)d_out, nv); }, false);
        run("copy simple nt", [&] { hipLaunchKernelGGL(k_copy_simple<true>, dim3((unsigned)((nv + 255) / 256)), dim3(256), 0, 0, (const d2 *)d_in, (d2 *)d_out, nv); }, false);
    }
#define COPY_SLAB(R, NT, SWZ) \
    run("copy slab R=" #R " nt=" #NT " swz=" #SWZ, [&] { hipLaunchKernelGGL((k_copy_slab<R, NT, SWZ>), dim3((unsigned)(g.n0 * g.n1 / R)), dim3(256), 0, 0, d_in, d_out, g); }, false)
    COPY_SLAB(4, true, true);
    COPY_SLAB(4, true, false);
    COPY_SLAB(4, false, true);
    COPY_SLAB(8, true, true);
    COPY_SLAB(2, true, true);
    COPY_SLAB(1, true, true);
    COPY_SLAB(1, true, false);
    COPY_SLAB(2, true, false);

#define SLAB(R, P, NT, SWZ) \
    run("slab R=" #R " P=" #P " nt=" #NT " swz=" #SWZ, [&] { hipLaunchKernelGGL((k_slab<R, P, NT, SWZ>), dim3((unsigned)(g.n0 / P * g.n1 / R)), dim3(256), 0, 0, d_in, d_out, g); })
    SLAB(4, 1, true, true);
    SLAB(4, 1, true, false);
    SLAB(4, 1, false, true);
    SLAB(2, 1, true, true);
    SLAB(8, 1, true, true);
    SLAB(16, 1, true, true);
    SLAB(2, 2, true, true);
    SLAB(4, 2, true, true);
    SLAB(8, 2, true, true);
    SLAB(2, 4, true, true);
    SLAB(4, 4, true, true);
    SLAB(4, 4, false, true);
    SLAB(2, 8, true, true);
    SLAB(4, 8, true, true);

#define MARCH1(RY, PF, NT, NXC) \
    run("march1 RY=" #RY " PF=" #PF " nt=" #NT " nxc=" #NXC, [&] { hipLaunchKernelGGL((k_march1<RY, PF, NT>), dim3((unsigned)(NXC * g.n1 / RY)), dim3(256), 0, 0, d_in, d_out, g, (int)(g.n0 / NXC)); })
    MARCH1(1, 1, true, 1);
    MARCH1(1, 2, true, 1);
    MARCH1(1, 3, true, 1);
    MARCH1(1, 2, true, 2);
    MARCH1(1, 3, false, 1);
    MARCH1(2, 1, true, 1);
    MARCH1(2, 2, true, 1);
    MARCH1(2, 3, true, 1);
    MARCH1(2, 1, true, 2);
    MARCH1(2, 2, true, 2);
    MARCH1(2, 1, true, 4);
    MARCH1(2, 2, true, 4);
    MARCH1(2, 1, true, 8);
    MARCH1(4, 1, true, 4);
    MARCH1(4, 2, true, 4);
    MARCH1(4, 1, true, 8);
    MARCH1(4, 1, true, 16);
    return 0;
}
