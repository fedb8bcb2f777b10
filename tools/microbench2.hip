// microbench2.hip — second tuning sweep (tool, not product): wider wave tiles (CZ chunks per row),
// prefetch depth, copy-kernel structure.  All variants are checked against the generic kernel.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off tools/microbench2.hip -o tools/microbench2
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                         \
    do {                                                                              \
        hipError_t e = (x);                                                           \
        if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } \
    } while (0)

typedef double d2 __attribute__((ext_vector_type(2)));

struct A {
    const double *in;
    double *out;
    long n0, n1, n2, p0, p1, off;
    double sx, sy, sz;
    int lx;
    long nxc, nty, ntz, nblocks;
    int swz;
};

__device__ __forceinline__ double dpp_shr1(double old, double src)
{
    unsigned long long o = __double_as_longlong(old), s = __double_as_longlong(src);
    int lo = __builtin_amdgcn_update_dpp((int)(o & 0xffffffffu), (int)(s & 0xffffffffu), 0x138, 0xf, 0xf, false);
    int hi = __builtin_amdgcn_update_dpp((int)(o >> 32), (int)(s >> 32), 0x138, 0xf, 0xf, false);
    return __longlong_as_double(((unsigned long long)(unsigned)hi << 32) | (unsigned)lo);
}
__device__ __forceinline__ double dpp_shl1(double old, double src)
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
__device__ __forceinline__ long swz(long bid, long nb)
{
    const long per = nb / 8;
    if (bid >= per * 8) return bid;
    return (bid % 8) * per + bid / 8;
}

// WY waves per block stacked along y; each wave: RY rows x (CZ chunks of 128 cells)
template <int RY, int CZ, int WY, int PF, bool NT>
__global__ void __launch_bounds__(64 * WY) march(A a)
{
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    long bid = a.swz ? swz(blockIdx.x, a.nblocks) : (long)blockIdx.x;
    const long tz = bid % a.ntz; bid /= a.ntz;
    const long ty = bid % a.nty;
    const long xc = bid / a.nty;
    const long kw = tz * (128 * CZ);
    const long j0 = ty * (WY * RY) + (long)w * RY;
    if (j0 >= a.n1) return;
    const long i0 = xc * a.lx, i1 = (i0 + a.lx < a.n0) ? i0 + a.lx : a.n0;
    long kc[CZ];
    bool ok[CZ];
#pragma unroll
    for (int c = 0; c < CZ; c++) { long k = kw + c * 128 + lane * 2; ok[c] = k < a.n2; kc[c] = k > a.n2 ? a.n2 : k; }
    long roff[RY + 2];
#pragma unroll
    for (int r = 0; r < RY + 2; r++) { long j = j0 + r - 1; if (j > a.n1) j = a.n1; roff[r] = a.off + j * a.p1; }
    long zh_k = (lane < 32) ? kw - 1 : kw + 128 * CZ;
    if (zh_k > a.n2) zh_k = a.n2;

    d2 pl[PF + 1][RY + 2][CZ];   // planes in flight: [0] = cur, [1..PF] = prefetched
    double zh[PF + 1][RY];
    d2 prev[RY][CZ];
    const double *in = a.in;
    {
        const double *p = in + (i0 - 1) * a.p0;
#pragma unroll
        for (int r = 0; r < RY; r++)
#pragma unroll
            for (int c = 0; c < CZ; c++) prev[r][c] = *(const d2 *)(p + roff[r + 1] + kc[c]);
    }
#pragma unroll
    for (int q = 0; q < PF; q++) {
        long ii = i0 + q; if (ii > a.n0) ii = a.n0;
        const double *p = in + ii * a.p0;
#pragma unroll
        for (int r = 0; r < RY + 2; r++)
#pragma unroll
            for (int c = 0; c < CZ; c++) pl[q][r][c] = *(const d2 *)(p + roff[r] + kc[c]);
#pragma unroll
        for (int r = 0; r < RY; r++) zh[q][r] = p[roff[r + 1] + zh_k];
    }
    for (long i = i0; i < i1; i++) {
        {   // prefetch plane i+PF into slot PF
            long ii = i + PF; if (ii > a.n0) ii = a.n0;
            const double *p = in + ii * a.p0;
#pragma unroll
            for (int r = 0; r < RY + 2; r++)
#pragma unroll
                for (int c = 0; c < CZ; c++) pl[PF][r][c] = *(const d2 *)(p + roff[r] + kc[c]);
#pragma unroll
            for (int r = 0; r < RY; r++) zh[PF][r] = p[roff[r + 1] + zh_k];
        }
#pragma unroll
        for (int r = 0; r < RY; r++) {
#pragma unroll
            for (int c = 0; c < CZ; c++) {
                const d2 cc = pl[0][r + 1][c], up = pl[0][r][c], dn = pl[0][r + 2][c], xp = pl[1][r + 1][c], xm = prev[r][c];
                double oldl = zh[0][r], oldr = zh[0][r];
                if (c > 0) oldl = readlane_d(pl[0][r + 1][c > 0 ? c - 1 : 0][1], 63);
                if (c < CZ - 1) oldr = readlane_d(pl[0][r + 1][c < CZ - 1 ? c + 1 : c][0], 0);
                const double zl = dpp_shr1(oldl, cc[1]);
                const double zr = dpp_shl1(oldr, cc[0]);
                d2 res;
                {
                    const double vm = 2 * cc[0];
                    const double lx = (xm[0] - vm + xp[0]) * a.sx, ly = (up[0] - vm + dn[0]) * a.sy, lz = (zl - vm + cc[1]) * a.sz;
                    res[0] = lx + ly + lz;
                }
                {
                    const double vm = 2 * cc[1];
                    const double lx = (xm[1] - vm + xp[1]) * a.sx, ly = (up[1] - vm + dn[1]) * a.sy, lz = (cc[0] - vm + zr) * a.sz;
                    res[1] = lx + ly + lz;
                }
                if (ok[c] && (j0 + r) < a.n1) {
                    d2 *po = (d2 *)(a.out + a.off + i * a.p0 + (j0 + r) * a.p1 + kc[c]);
                    if (NT) __builtin_nontemporal_store(res, po);
                    else *po = res;
                }
            }
        }
        // rotate
#pragma unroll
        for (int r = 0; r < RY; r++)
#pragma unroll
            for (int c = 0; c < CZ; c++) prev[r][c] = pl[0][r + 1][c];
#pragma unroll
        for (int q = 0; q < PF; q++) {
#pragma unroll
            for (int r = 0; r < RY + 2; r++)
#pragma unroll
                for (int c = 0; c < CZ; c++) pl[q][r][c] = pl[q + 1][r][c];
#pragma unroll
            for (int r = 0; r < RY; r++) zh[q][r] = zh[q + 1][r];
        }
    }
}

__global__ void __launch_bounds__(256) generic(A a)
{
    const long total = a.n0 * a.n1 * a.n2;
    for (long t = blockIdx.x * (long)blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
        const long k = t % a.n2, j = (t / a.n2) % a.n1, i = t / (a.n2 * a.n1);
        const double *c = a.in + a.off + i * a.p0 + j * a.p1 + k;
        const double vm = 2 * c[0];
        a.out[a.off + i * a.p0 + j * a.p1 + k] = (c[-a.p0] - vm + c[a.p0]) * a.sx + (c[-a.p1] - vm + c[a.p1]) * a.sy + (c[-1] - vm + c[1]) * a.sz;
    }
}

template <int U>
__global__ void __launch_bounds__(256) copy_u(const d2 *in, d2 *out, long n)
{
    const long stride = (long)gridDim.x * blockDim.x;
    long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
    for (; i + (U - 1) * stride < n; i += U * stride) {
        d2 v[U];
#pragma unroll
        for (int u = 0; u < U; u++) v[u] = in[i + u * stride];
#pragma unroll
        for (int u = 0; u < U; u++) out[i + u * stride] = v[u];
    }
    for (; i < n; i += stride) out[i] = in[i];
}
// block-contiguous copy: every block streams its own contiguous chunk
template <int U>
__global__ void __launch_bounds__(256) copy_chunk(const d2 *in, d2 *out, long n)
{
    const long per = (n + gridDim.x - 1) / gridDim.x;
    const long b0 = blockIdx.x * per, b1 = (b0 + per < n) ? b0 + per : n;
    long i = b0 + threadIdx.x;
    for (; i + (U - 1) * 256 < b1; i += U * 256) {
        d2 v[U];
#pragma unroll
        for (int u = 0; u < U; u++) v[u] = in[i + u * 256];
#pragma unroll
        for (int u = 0; u < U; u++) out[i + u * 256] = v[u];
    }
    for (; i < b1; i += 256) out[i] = in[i];
}
__global__ void __launch_bounds__(256) read_only(const d2 *in, double *sink, long n)
{
    const long stride = (long)gridDim.x * blockDim.x;
    d2 acc = {0, 0};
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += stride) acc += in[i];
    if (acc[0] + acc[1] == 1.2345) sink[0] = acc[0];
}
__global__ void __launch_bounds__(256) write_only(d2 *out, long n)
{
    const long stride = (long)gridDim.x * blockDim.x;
    d2 v = {1.0, 2.0};
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += stride) out[i] = v;
}

template <typename F>
static double time_it(F launch, int reps = 20)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; i++) launch();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < reps; i++) launch();
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipGetLastError());
    return ms / reps * 1e-3;
}

static double *g_ref = nullptr;
static size_t g_elems = 0;
static std::vector<double> g_h, g_r;

template <int RY, int CZ, int WY, int PF, bool NT>
static void run(A a, long want_blocks, int swz_on)
{
    a.swz = swz_on;
    a.ntz = (a.n2 + 128 * CZ - 1) / (128 * CZ);
    a.nty = (a.n1 + WY * RY - 1) / (WY * RY);
    long tiles = a.ntz * a.nty;
    long nxc = (want_blocks + tiles - 1) / tiles;
    if (nxc < 1) nxc = 1;
    if (nxc > a.n0) nxc = a.n0;
    long lx = (a.n0 + nxc - 1) / nxc;
    a.lx = (int)lx;
    a.nxc = (a.n0 + lx - 1) / lx;
    a.nblocks = a.nxc * tiles;
    CK(hipMemset(a.out, 0, g_elems * 8));
    double t = time_it([&] { hipLaunchKernelGGL((march<RY, CZ, WY, PF, NT>), dim3((unsigned)a.nblocks), dim3(64 * WY), 0, 0, a); });
    // verify against the generic kernel (interior only)
    CK(hipMemcpy(g_h.data(), a.out, g_elems * 8, hipMemcpyDeviceToHost));
    long bad = 0;
    for (long i = 0; i < a.n0 && bad < 5; i += 37)
        for (long j = 0; j < a.n1; j += 11)
            for (long k = 0; k < a.n2; k++) {
                size_t e = a.off + i * a.p0 + j * a.p1 + k;
                if (g_h[e] != g_r[e]) bad++;
            }
    double cells = (double)a.n0 * a.n1 * a.n2;
    printf("march RY=%d CZ=%d WY=%d PF=%d nt=%d swz=%d blocks=%5ld lx=%3d : %7.3f ms %6.1f Gcells/s %6.1f GB/s (%4.1f%%) %s\n", RY, CZ, WY, PF, (int)NT,
           swz_on, a.nblocks, a.lx, t * 1e3, cells / t / 1e9, cells * 16 / t / 1e9, cells * 16 / t / 8e12 * 100, bad ? "MISMATCH" : "ok");
    fflush(stdout);
}

int main(int argc, char **argv)
{
    long N = argc > 1 ? atol(argv[1]) : 512;
    A a;
    a.n0 = a.n1 = a.n2 = N;
    a.p1 = ((2 + N + 1 + 1) / 2) * 2;
    a.p0 = a.p1 * (N + 2);
    long pc = a.p0 * (N + 2);
    a.off = a.p0 + a.p1 + 2;
    a.sx = 1.0; a.sy = 0.5; a.sz = 0.25;
    g_elems = pc + 64;
    CK(hipMalloc((void **)&a.in, g_elems * 8));
    CK(hipMalloc(&a.out, g_elems * 8));
    CK(hipMalloc(&g_ref, g_elems * 8));
    g_h.resize(g_elems); g_r.resize(g_elems);
    for (size_t i = 0; i < g_elems; i++) g_h[i] = (double)((i * 2654435761u) % 1000) / 1000.0;
    CK(hipMemcpy((void *)a.in, g_h.data(), g_elems * 8, hipMemcpyHostToDevice));
    CK(hipMemset(g_ref, 0, g_elems * 8));
    {
        A r = a; r.out = g_ref;
        hipLaunchKernelGGL(generic, dim3(8192), dim3(256), 0, 0, r);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(g_r.data(), g_ref, g_elems * 8, hipMemcpyDeviceToHost));
    }
    printf("grid %ld^3 fp64 (%.1f MB per array)\n", N, g_elems * 8 / 1e6);
    const long nv = pc / 2;
    const double cb = 2.0 * nv * 16;
    for (int blocks : {1024, 2048, 4096}) {
        double t;
        t = time_it([&] { hipLaunchKernelGGL(copy_u<1>, dim3(blocks), dim3(256), 0, 0, (const d2 *)a.in, (d2 *)a.out, nv); });
        printf("copy_u<1> blocks=%d : %.3f ms %.0f GB/s\n", blocks, t * 1e3, cb / t / 1e9);
        t = time_it([&] { hipLaunchKernelGGL(copy_u<4>, dim3(blocks), dim3(256), 0, 0, (const d2 *)a.in, (d2 *)a.out, nv); });
        printf("copy_u<4> blocks=%d : %.3f ms %.0f GB/s\n", blocks, t * 1e3, cb / t / 1e9);
        t = time_it([&] { hipLaunchKernelGGL(copy_u<8>, dim3(blocks), dim3(256), 0, 0, (const d2 *)a.in, (d2 *)a.out, nv); });
        printf("copy_u<8> blocks=%d : %.3f ms %.0f GB/s\n", blocks, t * 1e3, cb / t / 1e9);
        t = time_it([&] { hipLaunchKernelGGL(copy_chunk<4>, dim3(blocks), dim3(256), 0, 0, (const d2 *)a.in, (d2 *)a.out, nv); });
        printf("copy_chunk<4> blocks=%d : %.3f ms %.0f GB/s\n", blocks, t * 1e3, cb / t / 1e9);
        t = time_it([&] { hipLaunchKernelGGL(copy_chunk<8>, dim3(blocks), dim3(256), 0, 0, (const d2 *)a.in, (d2 *)a.out, nv); });
        printf("copy_chunk<8> blocks=%d : %.3f ms %.0f GB/s\n", blocks, t * 1e3, cb / t / 1e9);
    }
    {
        double t = time_it([&] { hipLaunchKernelGGL(read_only, dim3(4096), dim3(256), 0, 0, (const d2 *)a.in, g_ref, nv); });
        printf("read_only  : %.3f ms %.0f GB/s\n", t * 1e3, nv * 16.0 / t / 1e9);
        t = time_it([&] { hipLaunchKernelGGL(write_only, dim3(4096), dim3(256), 0, 0, (d2 *)a.out, nv); });
        printf("write_only : %.3f ms %.0f GB/s\n", t * 1e3, nv * 16.0 / t / 1e9);
    }
    for (long wb : {256L, 512L, 1024L, 2048L}) {
        if (argc > 2) break;
        run<2, 1, 4, 1, false>(a, wb, 1);
        run<1, 1, 4, 1, false>(a, wb, 1);
        run<2, 2, 4, 1, false>(a, wb, 1);
        run<1, 2, 4, 1, false>(a, wb, 1);
        run<2, 4, 4, 1, false>(a, wb, 1);
        run<1, 4, 4, 1, false>(a, wb, 1);
        run<2, 1, 4, 2, false>(a, wb, 1);
        run<2, 2, 4, 2, false>(a, wb, 1);
        run<1, 4, 4, 2, false>(a, wb, 1);
        run<2, 4, 2, 1, false>(a, wb, 1);
        run<2, 4, 1, 1, false>(a, wb, 1);
        run<4, 4, 1, 1, false>(a, wb, 1);
        run<2, 2, 8, 1, false>(a, wb, 1);
    }
    run<2, 1, 4, 1, true>(a, 512, 1);
    run<2, 4, 4, 1, true>(a, 512, 1);
    run<2, 4, 4, 1, false>(a, 512, 0);
    // does the relative placement of the input and the output array matter (HBM channel / bank overlap)?
    {
        double *big;
        const size_t extra = 64u << 20;
        CK(hipMalloc(&big, g_elems * 8 + extra));
        for (size_t off_bytes : {(size_t)0, (size_t)256, (size_t)4096, (size_t)(64 << 10), (size_t)(1 << 20), (size_t)(2 << 20) + 4096, (size_t)(8 << 20) + 12288, (size_t)(32 << 20) + 65536}) {
            A b = a;
            b.out = (double *)((char *)big + off_bytes);
            b.swz = 1;
            b.ntz = 1; b.nty = (b.n1 + 1) / 2;
            long tiles = b.nty, nxc = (1024 + tiles - 1) / tiles;
            long lx = (b.n0 + nxc - 1) / nxc;
            b.lx = (int)lx; b.nxc = (b.n0 + lx - 1) / lx; b.nblocks = b.nxc * tiles;
            double t = time_it([&] { hipLaunchKernelGGL((march<2, 4, 1, 1, false>), dim3((unsigned)b.nblocks), dim3(64), 0, 0, b); });
            printf("out offset %10zu B (out-in = %ld B): %.4f ms  %.1f GB/s\n", off_bytes, (long)((char *)b.out - (char *)b.in), t * 1e3, 16.0 * b.n0 * b.n1 * b.n2 / t / 1e9);
        }
    }
    return 0;
}
