// microbench.hip — kernel-variant sweep for the Laplacian on MI355X (tuning tool, not product).
// Build:  hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off tools/microbench.hip -o tools/microbench
// Run  :  tools/microbench [n]      (n = cells per axis, default 512)
// Prints one line per variant: time per launch, Gcells/s, algorithmic GB/s (16 B per cell fp64).
#include "../py-pde_amd/csrc/pdehip_runtime.hip"
#include "../py-pde_amd/csrc/pdehip_kernels.hip"

#include <vector>

using namespace pdehip;

#define CK(x)                                                                         \
    do {                                                                              \
        hipError_t e = (x);                                                           \
        if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } \
    } while (0)

typedef double d2 __attribute__((ext_vector_type(2)));

__global__ void __launch_bounds__(256) copy_kernel(const d2 *in, d2 *out, long n)
{
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) out[i] = in[i];
}
__global__ void __launch_bounds__(256) copy_nt_kernel(const d2 *in, d2 *out, long n)
{
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        __builtin_nontemporal_store(in[i], out + i);
}
// copy with a source shifted by 8 bytes: probes the cost of 16-byte loads at 8-byte alignment
typedef double d2u __attribute__((ext_vector_type(2), aligned(8)));
__global__ void __launch_bounds__(256) copy_misaligned_kernel(const double *in, d2 *out, long n)
{
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        d2u v = *(const d2u *)(in + 2 * i + 1);
        d2 w; w[0] = v[0]; w[1] = v[1];
        out[i] = w;
    }
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

static LapArgs make_args(const NGrid &n, const void *in, void *out)
{
    LapArgs a;
    memset(&a, 0, sizeof(a));
    a.in = in; a.out = out; a.y = nullptr;
    a.n0 = n.n[0]; a.n1 = n.n[1]; a.n2 = n.n[2];
    a.p0 = n.p[0]; a.p1 = n.p[1]; a.off = n.off;
    a.o_off = n.off; a.o_s0 = n.p[0]; a.o_s1 = n.p[1];
    a.sx = n.lap_scale[0]; a.sy = n.lap_scale[1]; a.sz = n.lap_scale[2];
    a.s1 = 1.0; a.s2 = 0.1; a.gamma = 1;
    a.ndim = n.ndim;
    return a;
}

template <typename T, int VEC, int RY, int MODE, bool NT>
static void run_march(const char *tag, const NGrid &n, const void *in, void *out, long want_blocks, int no_swz, double bytes_per_cell)
{
    LapArgs a = make_args(n, in, out);
    a.no_swizzle = no_swz;
    a.ntz = (a.n2 + 64 * VEC - 1) / (64 * VEC);
    a.nty = (a.n1 + 4 * RY - 1) / (4 * RY);
    long tiles = a.ntz * a.nty;
    long nxc = (want_blocks + tiles - 1) / tiles;
    if (nxc < 1) nxc = 1;
    if (nxc > a.n0) nxc = a.n0;
    long lx = (a.n0 + nxc - 1) / nxc;
    a.lx = (int)lx;
    a.nxc = (a.n0 + lx - 1) / lx;
    a.nblocks = a.nxc * tiles;
    double t = time_it([&] { hipLaunchKernelGGL((lap_march_kernel<T, VEC, RY, MODE, true, true, NT>), dim3((unsigned)a.nblocks), dim3(256), 0, 0, a); });
    double cells = (double)a.n0 * a.n1 * a.n2;
    printf("%-28s RY=%d blocks=%5ld lx=%3d nt=%d swz=%d : %8.3f ms  %7.1f Gcells/s  %7.1f GB/s (%.1f%% of 8 TB/s)\n", tag, RY, a.nblocks,
           a.lx, (int)NT, !no_swz, t * 1e3, cells / t / 1e9, cells * bytes_per_cell / t / 1e9, cells * bytes_per_cell / t / 8e12 * 100);
    fflush(stdout);
}

int main(int argc, char **argv)
{
    long N = argc > 1 ? atol(argv[1]) : 512;
    pdehip_grid_t g;
    g.ndim = 3; g.dtype = PDEHIP_F64;
    for (int a = 0; a < 3; a++) { g.shape[a] = N; g.dx[a] = 1.0; }
    NGrid n;
    if (norm_grid(&g, &n)) { printf("norm_grid failed\n"); return 1; }
    size_t bytes = (size_t)(n.pc + kAllocSlack) * 8;
    double *in, *out;
    CK(hipMalloc(&in, bytes)); CK(hipMalloc(&out, bytes));
    std::vector<double> h((size_t)n.pc + kAllocSlack);
    for (size_t i = 0; i < h.size(); i++) h[i] = (double)((i * 2654435761u) % 1000) / 1000.0;
    CK(hipMemcpy(in, h.data(), bytes, hipMemcpyHostToDevice));
    CK(hipMemset(out, 0, bytes));
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    printf("device: %s (%s) CUs=%d  grid %ld^3 fp64, full array %.1f MB\n", prop.name, prop.gcnArchName, prop.multiProcessorCount, N, bytes / 1e6);

    // copy ceilings
    long nv = n.pc / 2;
    for (int blocks : {2048, 4096, 8192}) {
        double t = time_it([&] { hipLaunchKernelGGL(copy_kernel, dim3(blocks), dim3(256), 0, 0, (const d2 *)in, (d2 *)out, nv); });
        printf("copy double2 blocks=%d        : %8.3f ms  %7.1f GB/s (read+write)\n", blocks, t * 1e3, 2.0 * nv * 16 / t / 1e9);
    }
    {
        double t = time_it([&] { hipLaunchKernelGGL(copy_nt_kernel, dim3(4096), dim3(256), 0, 0, (const d2 *)in, (d2 *)out, nv); });
        printf("copy double2 nontemporal      : %8.3f ms  %7.1f GB/s\n", t * 1e3, 2.0 * nv * 16 / t / 1e9);
        t = time_it([&] { hipLaunchKernelGGL(copy_misaligned_kernel, dim3(4096), dim3(256), 0, 0, (const double *)in, (d2 *)out, nv - 1); });
        printf("copy double2 src+8B misaligned: %8.3f ms  %7.1f GB/s\n", t * 1e3, 2.0 * nv * 16 / t / 1e9);
        t = time_it([&] { CK(hipMemcpyAsync(out, in, bytes, hipMemcpyDeviceToDevice, 0)); });
        printf("hipMemcpy D2D                 : %8.3f ms  %7.1f GB/s\n", t * 1e3, 2.0 * bytes / t / 1e9);
    }

    // generic kernel
    {
        LapArgs a = make_args(n, in, out);
        double t = time_it([&] { hipLaunchKernelGGL((lap_generic_kernel<double, LAP_PLAIN>), dim3(8192), dim3(256), 0, 0, a); });
        double cells = (double)N * N * N;
        printf("generic 1 cell/thread         : %8.3f ms  %7.1f Gcells/s  %7.1f GB/s\n", t * 1e3, cells / t / 1e9, cells * 16 / t / 1e9);
    }
    // register-pipelined variants
    for (long wb : {512L, 1024L, 2048L, 4096L, 8192L}) {
        run_march<double, 2, 2, LAP_PLAIN, false>("march plain", n, in, out, wb, 0, 16);
        run_march<double, 2, 4, LAP_PLAIN, false>("march plain", n, in, out, wb, 0, 16);
        run_march<double, 2, 8, LAP_PLAIN, false>("march plain", n, in, out, wb, 0, 16);
    }
    for (long wb : {1024L, 2048L, 4096L}) {
        run_march<double, 2, 4, LAP_PLAIN, true>("march plain", n, in, out, wb, 0, 16);
        run_march<double, 2, 4, LAP_PLAIN, false>("march plain", n, in, out, wb, 1, 16);
        run_march<double, 2, 8, LAP_PLAIN, true>("march plain", n, in, out, wb, 0, 16);
    }
    run_march<double, 2, 4, LAP_EULER, false>("march euler (y=in)", n, in, out, 2048, 0, 16);
    run_march<double, 2, 4, LAP_EULER, true>("march euler (y=in)", n, in, out, 2048, 0, 16);
    run_march<double, 2, 4, LAP_CH_MU, false>("march ch_mu", n, in, out, 2048, 0, 16);
    return 0;
}
