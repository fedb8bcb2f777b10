// pdehip_comm.hip — slab-parallel halo exchange and the slab Euler loop, straight on RCCL.
//
// Replaces the reference's MPI face exchange inside every right-hand side
// (pde/backends/numba_mpi/backend.py:30-194, pde/grids/boundaries/local.py:561-662) and the
// MAX all-reduce of the adaptive error (pde/backends/base.py:678-712).  One process per GPU; the
// communicator is created from an ncclUniqueId that the host distributes (torch.distributed).
//
// librccl is NOT linked: the host passes the path of the librccl.so that is already loaded in the
// process (torch ships its own copy) and the entry points are resolved with dlsym, so there is
// exactly one RCCL in the address space.
//
// Measured motivation (profiles/r01_probe_slab.log): driving send/recv through
// torch.distributed.batch_isend_irecv costs ~260 us of host time per step, 5x the 53 us a
// 64x512x512 slab needs on the GPU; the loop below enqueues a step in a few tens of us and never
// synchronises with the host.
#include <dlfcn.h>

#include <rccl/rccl.h>

#include <chrono>
#include <vector>

#include "pdehip_common.h"
#include "pdehip_slab_loops.h"
#include "pdehip_block_loops.h"

using namespace pdehip;

namespace {

struct Rccl {
    void *handle = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    // (optional: what the communicator reports about itself, pdehip_comm_info)
    decltype(&ncclCommCount) CommCount = nullptr;
    decltype(&ncclCommUserRank) CommUserRank = nullptr;
    decltype(&ncclCommCuDevice) CommCuDevice = nullptr;
    decltype(&ncclGetVersion) GetVersion = nullptr;
};
Rccl g_rccl;

int load_rccl(const char *path)
{
    if (g_rccl.handle) return 0;
    void *h = dlopen((path && path[0]) ? path : "librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) PDEHIP_FAIL(E_RUNTIME, "cannot load RCCL (%s): %s", path ? path : "librccl.so", dlerror());
#define PDEHIP_SYM(field, name)                                                          \
    g_rccl.field = reinterpret_cast<decltype(g_rccl.field)>(dlsym(h, name));             \
    if (!g_rccl.field) PDEHIP_FAIL(E_RUNTIME, "RCCL symbol %s not found", name)
    PDEHIP_SYM(GetUniqueId, "ncclGetUniqueId");
    PDEHIP_SYM(CommInitRank, "ncclCommInitRank");
    PDEHIP_SYM(CommDestroy, "ncclCommDestroy");
    PDEHIP_SYM(GetErrorString, "ncclGetErrorString");
    PDEHIP_SYM(GroupStart, "ncclGroupStart");
    PDEHIP_SYM(GroupEnd, "ncclGroupEnd");
    PDEHIP_SYM(Send, "ncclSend");
    PDEHIP_SYM(Recv, "ncclRecv");
    PDEHIP_SYM(AllReduce, "ncclAllReduce");
#undef PDEHIP_SYM
    g_rccl.CommCount = reinterpret_cast<decltype(g_rccl.CommCount)>(dlsym(h, "ncclCommCount"));
    g_rccl.CommUserRank = reinterpret_cast<decltype(g_rccl.CommUserRank)>(dlsym(h, "ncclCommUserRank"));
    g_rccl.CommCuDevice = reinterpret_cast<decltype(g_rccl.CommCuDevice)>(dlsym(h, "ncclCommCuDevice"));
    g_rccl.GetVersion = reinterpret_cast<decltype(g_rccl.GetVersion)>(dlsym(h, "ncclGetVersion"));
    g_rccl.handle = h;
    return 0;
}

#define PDEHIP_NCCL(expr)                                                                                  \
    do {                                                                                                   \
        ncclResult_t _r = (expr);                                                                          \
        if (_r != ncclSuccess) PDEHIP_FAIL(E_RUNTIME, "%s failed: %s", #expr, g_rccl.GetErrorString(_r)); \
    } while (0)

struct Block2Ctx;   // scratch of the fast block loop (defined next to it, below); owned by the communicator
void block2_release(Block2Ctx *x);

struct Comm {
    ncclComm_t comm = nullptr;
    int rank = 0, size = 1;
    hipStream_t halo = nullptr;  // stream of the exchange + boundary-layer kernels
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};   // slab::EV_COMP, EV_HALO, EV_BND, EV_BND2
    double *scratch2 = nullptr;  // device: {value, nan flag} for the MAX all-reduce
    void *ext[4] = {nullptr, nullptr, nullptr, nullptr};   // private slab arrays with two / four halo layers per side (two-steps-per-sweep loops)
    size_t ext_bytes = 0;
    int ext_count = 0;
    // block decomposition: contiguous staging buffers of the packed faces, [axis][side][0 send / 1 receive]
    void *stg[3][2][2] = {{{nullptr, nullptr}, {nullptr, nullptr}}, {{nullptr, nullptr}, {nullptr, nullptr}}, {{nullptr, nullptr}, {nullptr, nullptr}}};
    size_t stg_bytes[3] = {0, 0, 0};
    Block2Ctx *b2 = nullptr;     // (was a process-wide table keyed by the Comm pointer and never released)
    // The scratch above (ext, stg, b2) is shared by every run on this context - all steppers without a communicator share the serial
    // context - and those runs enqueue on their own streams: a run waits for the end of the run before it (ScratchTurn).
    hipEvent_t ev_done = nullptr;
    bool ev_done_recorded = false;
    // streams found to share a hardware queue with a compute stream (separate_queues): kept alive so that their queue stays taken
    std::vector<hipStream_t> parked;
    hipStream_t probed_for = nullptr;   // the compute stream the halo stream was last checked against
    bool probed = false;
};

// one run's turn on the scratch of a context: waits (on the device) for the run before it, marks its own end
struct ScratchTurn {
    Comm *c;
    hipStream_t st;
    ScratchTurn(Comm *c_, hipStream_t st_) : c(c_), st(st_)
    {
        if (c->ev_done && c->ev_done_recorded) (void)hipStreamWaitEvent(st, c->ev_done, 0);
    }
    ~ScratchTurn()
    {
        if (c->ev_done && hipEventRecord(c->ev_done, st) == hipSuccess) c->ev_done_recorded = true;
    }
};

// frees every scratch buffer of a context (after the work that uses them)
void release_scratch(Comm *c)
{
    if (c->ev_done && c->ev_done_recorded) (void)hipEventSynchronize(c->ev_done);
    if (c->halo) (void)hipStreamSynchronize(c->halo);
    for (auto st : c->parked) (void)hipStreamDestroy(st);
    c->parked.clear();
    c->probed = false;
    for (auto &e : c->ext) { (void)hipFree(e); e = nullptr; }
    c->ext_bytes = 0; c->ext_count = 0;
    for (int a = 0; a < 3; a++) {
        for (int side = 0; side < 2; side++)
            for (int r = 0; r < 2; r++) { (void)hipFree(c->stg[a][side][r]); c->stg[a][side][r] = nullptr; }
        c->stg_bytes[a] = 0;
    }
    block2_release(c->b2);
    c->b2 = nullptr;
}

// serial use of the slab loops (comm == NULL, no neighbours): streams / events of a process-wide context without RCCL
Comm *serial_context()
{
    static Comm ctx;
    return &ctx;
}

constexpr int kHaloPriorityDefault = 0;

int ensure_streams(Comm *c)
{
    if (!c->halo) {
        // The halo stream carries the boundary sweeps and the RCCL kernels.  A HIGH-PRIORITY stream was measured SLOWER (64 x 512 x
        // 512 slab, halo to self: 0.060 vs 0.043 ms per step, profiles/r03_probe_slab.md), so it is a plain stream unless
        // PDEHIP_HALO_PRIORITY=1 asks for the experiment.
        // PDEHIP_HALO_PRIORITY: 0 plain stream (default), 1 highest, -1 lowest priority.  Streams of different priority never share a hardware queue:
        // round 6 found the compute stream and a plain halo stream of the same process on ONE hardware queue whenever the process had created an odd
        // number of other streams before (0.054 instead of 0.040 ms per step on a 64-layer slab: the two chains of the loop serialise,
        // profiles/r06_probe_slab.md).
        int lo = 0, hi = 0;
        const int want = getenv("PDEHIP_HALO_PRIORITY") ? atoi(getenv("PDEHIP_HALO_PRIORITY")) : kHaloPriorityDefault;
        if (want == 0 || hipDeviceGetStreamPriorityRange(&lo, &hi) != hipSuccess || hi == lo) PDEHIP_HIP(hipStreamCreateWithFlags(&c->halo, hipStreamNonBlocking));
        else PDEHIP_HIP(hipStreamCreateWithPriority(&c->halo, hipStreamNonBlocking, want > 0 ? hi : lo));   // (hi = the numerically lowest value = highest priority)
    }
    for (auto &e : c->ev)
        if (!e) PDEHIP_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    if (!c->ev_done) PDEHIP_HIP(hipEventCreateWithFlags(&c->ev_done, hipEventDisableTiming));
    return 0;
}

// ~`ticks` of the 100 MHz wall clock on one lane (separate_queues)
__global__ void spin_kernel(long ticks)
{
    const long t0 = (long)wall_clock64();
    while ((long)wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
}

// The overlapped loops need the compute stream and the halo stream on DIFFERENT hardware queues.  The HIP runtime maps streams onto a handful
// of queues by creation order; round 6 found the two on ONE queue whenever the process had created an odd number of other streams before
// (bench.py after its side measurements: 0.054 instead of 0.040 ms per step on a 64-layer slab - the two chains of the loop serialise;
// priorities move the collision elsewhere: profiles/r06_probe_slab.md).  So: measured, once per pair of streams.  Two 200 us spin kernels,
// one per stream; if they take the time of both, the halo stream is parked (kept alive: its queue stays taken) and a new one is created,
// up to 6 times.  ~1 ms per communicator.  PDEHIP_QUEUE_PROBE=0: off.
int separate_queues(Comm *c, hipStream_t comp)
{
    static const bool off = getenv("PDEHIP_QUEUE_PROBE") && getenv("PDEHIP_QUEUE_PROBE")[0] == '0';
    if (off || (c->probed && c->probed_for == comp)) return 0;
    c->probed = true; c->probed_for = comp;
    const long ticks = 20000;   // 200 us
    for (int attempt = 0; attempt < 6; attempt++) {
        PDEHIP_HIP(hipStreamSynchronize(comp));
        PDEHIP_HIP(hipStreamSynchronize(c->halo));
        hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(1), 0, comp, 100L);      // (code object loaded, both queues awake)
        hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(1), 0, c->halo, 100L);
        PDEHIP_HIP(hipStreamSynchronize(comp));
        PDEHIP_HIP(hipStreamSynchronize(c->halo));
        const auto t0 = std::chrono::steady_clock::now();
        hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(1), 0, comp, ticks);
        hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(1), 0, c->halo, ticks);
        PDEHIP_HIP(hipStreamSynchronize(comp));
        PDEHIP_HIP(hipStreamSynchronize(c->halo));
        const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
        if (us < 330.0) return 0;   // overlapped (two in a row would take >= 400)
        c->parked.push_back(c->halo);
        c->halo = nullptr;
        PDEHIP_HIP(hipStreamCreateWithFlags(&c->halo, hipStreamNonBlocking));
    }
    return 0;   // (no separate queue found: the loops are still correct, only slower)
}

__global__ void pack_nan_kernel(const double *in, double *out2)
{
    const double v = in[0];
    const bool isn = (v != v);
    out2[0] = isn ? 0.0 : v;
    out2[1] = isn ? 1.0 : 0.0;
}
__global__ void unpack_nan_kernel(const double *in2, double *out)
{
    out[0] = (in2[1] > 0.0) ? __longlong_as_double(0x7ff8000000000000LL) : in2[0];
}

// face <-> contiguous staging buffer (block decomposition): layer `idx` along axis `a` (normalised axes), interior of the face only
struct FaceCopy {
    void *buf, *packed;
    long base;       // element offset of cell (idx; 0, 0) of the face
    long m1, m2;     // face extents
    long q1, q2;     // element pitches of the two face axes
};
struct FaceCopyMany {
    FaceCopy f[6];
    long start[7];   // first work item of face k (start[n] = total)
    int n;
};
template <typename T, bool PACK>
__global__ void __launch_bounds__(256) face_copy_kernel(FaceCopyMany a)
{
    const long total = a.start[a.n];
    for (long t = blockIdx.x * 256L + threadIdx.x; t < total; t += (long)gridDim.x * 256L) {
        int k = 0;
#pragma unroll
        for (int q = 1; q < 6; q++)
            if (q < a.n && t >= a.start[q]) k = q;
        const FaceCopy &f = a.f[k];
        const long loc = t - a.start[k];
        const long u = loc / f.m2, v = loc % f.m2;
        const long e = f.base + u * f.q1 + v * f.q2;
        T *buf = (T *)f.buf;
        T *pk = (T *)f.packed;
        if (PACK) pk[loc] = buf[e];
        else buf[e] = pk[loc];
    }
}

// (defined below, next to rim2_kernel) the two-ended boundary sweep of a slab - own layers 0, 1 and n-2, n-1 - by the rim kernel of the fast
// block loop: no march, every operand of a tile requested at once.  *done = false: not covered (the two-level kernel takes it).
int slab_rim2(const pdehip_grid_t *gs, const void *in, void *out, double D, double dt, const pdehip_bc_face_t *faces, void *st, bool *done);

// The `Ops` policy of pdehip_slab_loops.h on the device: HIP streams / events, RCCL point-to-point over xGMI, gfx950 kernels.
struct HipOps {
    Comm *c;
    void *halo() { return c->halo; }
    // layers per side of the thick boundary chunks of slab::euler2_run (PDEHIP_SLAB_THICK; default 0: the thin boundary sweep on the halo
    // stream - the thick schedule measured SLOWER to self, 0.047-0.048 against 0.043 ms per step at 64 x 512 x 512: the RCCL kernel
    // starves next to the inner launch and the next boundary chunks wait for it, profiles/r05_probe_block.md)
    long slab_thick()
    {
        const char *e = getenv("PDEHIP_SLAB_THICK");   // (read per run: a test switches it inside one process)
        const long v = e ? atol(e) : 0;
        return v < 0 ? 0 : v;
    }
    // schedule of pdehip_slab_euler4_run (PDEHIP_SLAB_DEEP_MODE): 1 = the first sweep of a group cut in two (interior / boundary), 2 = both sweeps cut,
    // 3 = the boundary layers on the halo stream, a group ahead (slab::euler4p_run), 4 = the same with the second boundary pass behind the first
    // interior sweep of its group (less redundant work)
    static constexpr int kDeepModeDefault = 3;
    int deep_mode()
    {
        const char *e = getenv("PDEHIP_SLAB_DEEP_MODE");   // (read per run: a test switches it inside one process)
        const int v = e ? atoi(e) : 0;
        return (v >= 1 && v <= 4) ? v : kDeepModeDefault;
    }
    int record(int ev, void *st) { PDEHIP_HIP(hipEventRecord(c->ev[ev], as_stream(st))); return 0; }
    int wait(void *st, int ev) { PDEHIP_HIP(hipStreamWaitEvent(as_stream(st), c->ev[ev], 0)); return 0; }
    int group_start() { PDEHIP_NCCL(g_rccl.GroupStart()); return 0; }
    int group_end() { PDEHIP_NCCL(g_rccl.GroupEnd()); return 0; }
    int send(const void *p, size_t bytes, int peer, void *st)
    {
        PDEHIP_NCCL(g_rccl.Send(p, bytes, ncclInt8, peer, c->comm, as_stream(st)));
        return 0;
    }
    int recv(void *p, size_t bytes, int peer, void *st)
    {
        PDEHIP_NCCL(g_rccl.Recv(p, bytes, ncclInt8, peer, c->comm, as_stream(st)));
        return 0;
    }
    int copy(void *dst, const void *src, size_t bytes, void *st)
    {
        PDEHIP_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, as_stream(st)));
        return 0;
    }
    int zero(void *p, size_t bytes, void *st) { PDEHIP_HIP(hipMemsetAsync(p, 0, bytes, as_stream(st))); return 0; }
    int refresh(void *bc_program, double t, const void *in, void *st) { return pdehip_bcprog_run(bc_program, t, in, st); }
    int fail(const char *msg) { PDEHIP_FAIL(E_NOTIMPL, "%s", msg); }
    int fail_runtime(const char *fmt, double v) { PDEHIP_FAIL(E_RUNTIME, fmt, v); }
    // BCs of `in` (faces not marked SKIP) + stencil into the full array `out`
    int lap(const pdehip_grid_t *gs, void *in, const void *y, void *out, int kind, double s1, double s2, double gamma,
            const pdehip_bc_face_t *faces, void *st, const StageFuse *sf)
    {
        const int mode = kind == slab::K_EULER ? LAP_EULER : kind == slab::K_SCALED ? LAP_SCALED : kind == slab::K_CH_MU ? LAP_CH_MU : LAP_STAGE;
        return laplace_with_input_bcs(gs, in, y, out, mode, s1, s2, gamma, faces, st, sf);
    }
    int euler2(const pdehip_grid_t *gs, const void *in, void *out, double D, double dt, const pdehip_bc_face_t *faces, void *st, bool *done,
               int xplain, bool dry, int ends)
    {
        // The two-ended boundary sweep of a slab between two neighbours: 2 + 2 planes.  As a march of the two-level kernel it is a chain of
        // seven dependent plane loads per wave (41 us for 512 x 512 planes next to the interior sweep, profiles/r03_probe_slab.md) and
        // heads the critical path boundary -> send / receive -> next boundary; the rim kernel asks for everything at once.
        if (ends == 2 && xplain == 1 && !dry) {
            PDEHIP_TRY(slab_rim2(gs, in, out, D, dt, faces, st, done));
            if (*done) return 0;
        }
        return euler2_with_input_bcs(gs, in, out, D, dt, faces, st, done, xplain, dry, ends);
    }
    int ch_fused(const pdehip_grid_t *gs, const void *in, void *out, double gamma, double dt, bool euler, const pdehip_bc_face_t *fc,
                 const pdehip_bc_face_t *fm, void *st, bool *done, int xplain, bool dry, const StageFuse *sf)
    {
        return cahn_hilliard_fused(gs, in, out, gamma, dt, euler, fc, fm, st, done, xplain, dry, sf);
    }
    // the pointwise Runge-Kutta combination `sf` with the slope k already stored (unfused stages)
    int combine(const pdehip_grid_t *g, void *k, const StageFuse &sf, void *st)
    {
        if (sf.kind == 0) {
            const void *ks[6];
            double cf[6];
            int n = 0;
            for (; n < 5 && sf.k[n]; n++) { ks[n] = sf.k[n]; cf[n] = sf.c[n]; }
            ks[n] = k; cf[n] = sf.c_new;
            return pdehip_lincomb(g, 1, sf.out2, sf.y, n + 1, cf, ks, st);
        }
        if (sf.kind == 1) {
            if (sf.out2 != sf.y) PDEHIP_FAIL(E_RUNTIME, "internal: the RK4 update works in place");
            return pdehip_rk4_combine(g, 1, sf.out2, sf.k[0], sf.k[1], sf.k[2], k, st);
        }
        if (sf.kind == 2) {
            const void *k6[6] = {sf.k[0], sf.k[0] /* k2 does not enter */, sf.k[1], sf.k[2], sf.k[3], k};
            return pdehip_rkf45_combine(g, 1, sf.y, sf.out2, k6, sf.err, st);
        }
        if (sf.kind == 4) return pdehip_euler_adaptive_combine(g, 1, sf.y, sf.k[0], sf.c[0], sf.k[1], k, sf.out2, sf.err, st);
        PDEHIP_FAIL(E_NOTIMPL, "internal: unknown stage kind %d", sf.kind);
    }
    // in-place MAX over all ranks of one fp64 device scalar; NaN wins like numpy.max
    int allreduce_max(double *dev_scalar, void *stream)
    {
        if (c->size == 1) return 0;
        hipStream_t st = as_stream(stream);
        hipLaunchKernelGGL(pack_nan_kernel, dim3(1), dim3(1), 0, st, dev_scalar, c->scratch2);
        PDEHIP_NCCL(g_rccl.AllReduce(c->scratch2, c->scratch2, 2, ncclFloat64, ncclMax, c->comm, st));
        hipLaunchKernelGGL(unpack_nan_kernel, dim3(1), dim3(1), 0, st, c->scratch2, dev_scalar);
        PDEHIP_HIP(hipGetLastError());
        return 0;
    }
    int read_scalar(double *host, const double *dev, void *st)
    {
        PDEHIP_HIP(hipMemcpyAsync(host, dev, sizeof(double), hipMemcpyDeviceToHost, as_stream(st)));
        PDEHIP_HIP(hipStreamSynchronize(as_stream(st)));
        return 0;
    }
    // --- block decomposition (pdehip_block_loops.h) ---
    const NGrid *bn = nullptr;   // normalised local grid of the block run in progress
    void *stage(int axis, int side, bool recv) { return c->stg[axis][side][recv ? 1 : 0]; }
    int face_copy(const block::Geo &q, void *buf, const block::FaceJob *jobs, int njobs, bool pack, void *st)
    {
        const int o = 3 - q.ndim;   // grid axis -> normalised axis
        FaceCopyMany m;
        m.n = njobs;
        long total = 0;
        for (int j = 0; j < njobs; j++) {
            const int axis = jobs[j].axis;
            int others[2], k = 0;
            for (int a = 0; a < q.ndim; a++)
                if (a != axis) others[k++] = a;
            FaceCopy &f = m.f[j];
            f.buf = buf; f.packed = jobs[j].packed;
            f.base = bn->off + jobs[j].index * bn->p[o + axis];
            f.m1 = k == 2 ? q.n[others[0]] : 1;
            f.m2 = q.n[others[k - 1]];
            f.q1 = k == 2 ? bn->p[o + others[0]] : 0;
            f.q2 = bn->p[o + others[k - 1]];
            m.start[j] = total;
            total += f.m1 * f.m2;
        }
        for (int j = njobs; j < 7; j++) m.start[j] = total;
        const unsigned blocks = (unsigned)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
        hipStream_t s = as_stream(st);
        if (q.esz == 8) {
            if (pack) hipLaunchKernelGGL((face_copy_kernel<double, true>), dim3(blocks), dim3(256), 0, s, m);
            else hipLaunchKernelGGL((face_copy_kernel<double, false>), dim3(blocks), dim3(256), 0, s, m);
        } else {
            if (pack) hipLaunchKernelGGL((face_copy_kernel<float, true>), dim3(blocks), dim3(256), 0, s, m);
            else hipLaunchKernelGGL((face_copy_kernel<float, false>), dim3(blocks), dim3(256), 0, s, m);
        }
        PDEHIP_HIP(hipGetLastError());
        return 0;
    }
    int lincomb(const pdehip_grid_t *g, void *out, const void *y, int n, const double *cf, const void *const *k, void *st) { return pdehip_lincomb(g, 1, out, y, n, cf, k, st); }
    int rk4_combine(const pdehip_grid_t *g, void *y, const void *k1, const void *k2, const void *k3, const void *k4, void *st) { return pdehip_rk4_combine(g, 1, y, k1, k2, k3, k4, st); }
    int rkf45_combine(const pdehip_grid_t *g, const void *y, void *ynew, const void *const *k6, double *err, void *st) { return pdehip_rkf45_combine(g, 1, y, ynew, k6, err, st); }
    int euler_adaptive_combine(const pdehip_grid_t *g, const void *y, const void *rate, double dt, const void *half, const void *k, void *out, double *err,
                               void *st)
    {
        return pdehip_euler_adaptive_combine(g, 1, y, rate, dt, half, k, out, err, st);
    }
};

// geometry of a block + its staging buffers; nb6[2 * axis + side] = neighbour rank or -1
int make_block(Comm *c, const pdehip_grid_t *g, const NGrid &n, const int *nb6, block::Geo *q)
{
    if (g->ndim < 2) PDEHIP_FAIL(E_NOTIMPL, "block decomposition: 2-D and 3-D grids");
    q->ndim = g->ndim;
    q->esz = (size_t)elem_size(n.dtype);
    for (int a = 0; a < 3; a++) { q->n[a] = a < g->ndim ? g->shape[a] : 1; q->nb[a][0] = q->nb[a][1] = -1; }
    for (int a = 0; a < g->ndim; a++)
        for (int side = 0; side < 2; side++) {
            const int peer = nb6[2 * a + side];
            if (peer >= c->size) PDEHIP_FAIL(E_VALUE, "block: neighbour rank %d outside of world size %d", peer, c->size);
            q->nb[a][side] = peer < 0 ? -1 : peer;
        }
    for (int a = 0; a < g->ndim; a++) {
        const size_t need = q->face_elems(a) * q->esz;
        if ((q->nb[a][0] >= 0 || q->nb[a][1] >= 0) && c->stg_bytes[a] < need) {
            if (c->ev_done && c->ev_done_recorded) PDEHIP_HIP(hipEventSynchronize(c->ev_done));
            PDEHIP_HIP(hipStreamSynchronize(c->halo));
            for (int side = 0; side < 2; side++)
                for (int r = 0; r < 2; r++) {
                    if (c->stg[a][side][r]) (void)hipFree(c->stg[a][side][r]);
                    PDEHIP_HIP(hipMalloc(&c->stg[a][side][r], need));
                }
            c->stg_bytes[a] = need;
        }
    }
    return 0;
}

int make_geo(const pdehip_grid_t *g, NGrid *n, slab::Geo *q)
{
    PDEHIP_TRY(norm_grid(g, n));
    q->nloc = g->shape[0];
    q->esz = (size_t)elem_size(n->dtype);
    q->lp = (size_t)n->p[3 - n->ndim] * q->esz;
    return 0;
}

// comm handle -> context; NULL is allowed for a process without neighbours (serial use of the loops)
int context(void *comm, int lower, int upper, Comm **out)
{
    Comm *c = static_cast<Comm *>(comm);
    if (!c) {
        if (lower >= 0 || upper >= 0) PDEHIP_FAIL(E_VALUE, "a slab with neighbours needs a communicator");
        c = serial_context();
    }
    PDEHIP_TRY(ensure_streams(c));
    *out = c;
    return 0;
}

// private arrays of the two-step slab loops: the layout of a slab of nloc + 2 * (depth - 1) layers, i.e. `depth` halo layers per side
int slab_private_arrays(Comm *c, const pdehip_grid_t *g_local, const slab::Geo &q, long depth, hipStream_t comp, int count = 2)
{
    pdehip_grid_t ge = *g_local;
    ge.shape[0] = q.nloc + 2 * (depth - 1);
    NGrid ne;
    PDEHIP_TRY(norm_grid(&ge, &ne));
    const size_t need = (size_t)(ne.pc + kAllocSlack) * q.esz;
    if (c->ext_bytes < need || c->ext_count < count) {
        if (c->ev_done && c->ev_done_recorded) PDEHIP_HIP(hipEventSynchronize(c->ev_done));
        PDEHIP_HIP(hipStreamSynchronize(c->halo));
        const size_t bytes = need > c->ext_bytes ? need : c->ext_bytes;
        for (auto &e : c->ext) { (void)hipFree(e); e = nullptr; }
        c->ext_bytes = 0; c->ext_count = 0;
        for (int k = 0; k < count; k++) {
            PDEHIP_HIP(hipMalloc(&c->ext[k], bytes));
            PDEHIP_HIP(hipMemsetAsync(c->ext[k], 0, bytes, comp));
        }
        c->ext_bytes = bytes; c->ext_count = count;
    }
    return 0;
}

int check_rhs(const pdehip_rhs_t *rhs)
{
    if (!rhs) PDEHIP_FAIL(E_VALUE, "rhs descriptor is NULL");
    if (rhs->kind != PDEHIP_RHS_DIFFUSION && rhs->kind != PDEHIP_RHS_CAHN_HILLIARD) PDEHIP_FAIL(E_NOTIMPL, "unknown rhs kind %d", rhs->kind);
    return 0;
}

}  // namespace

extern "C" {

int pdehip_comm_unique_id(const char *librccl_path, void *id128)
{
    PDEHIP_TRY(load_rccl(librccl_path));
    if (!id128) PDEHIP_FAIL(E_VALUE, "id buffer is NULL");
    ncclUniqueId id;
    PDEHIP_NCCL(g_rccl.GetUniqueId(&id));
    memcpy(id128, id.internal, NCCL_UNIQUE_ID_BYTES);
    return 0;
}

int pdehip_comm_create(const char *librccl_path, const void *id128, int rank, int size, void **comm)
{
    PDEHIP_TRY(load_rccl(librccl_path));
    if (!id128 || !comm) PDEHIP_FAIL(E_VALUE, "comm_create: NULL pointer");
    if (rank < 0 || rank >= size) PDEHIP_FAIL(E_VALUE, "comm_create: rank %d outside of world size %d", rank, size);
    ncclUniqueId id;
    memcpy(id.internal, id128, NCCL_UNIQUE_ID_BYTES);
    Comm *c = new Comm();
    c->rank = rank; c->size = size;
    PDEHIP_NCCL(g_rccl.CommInitRank(&c->comm, size, id, rank));
    PDEHIP_TRY(ensure_streams(c));
    PDEHIP_HIP(hipMalloc(&c->scratch2, 2 * sizeof(double)));
    *comm = c;
    return 0;
}

int pdehip_comm_destroy(void *comm)
{
    if (!comm) return 0;
    Comm *c = static_cast<Comm *>(comm);
    release_scratch(c);
    g_rccl.CommDestroy(c->comm);
    (void)hipStreamDestroy(c->halo);
    for (auto &e : c->ev) (void)hipEventDestroy(e);
    (void)hipEventDestroy(c->ev_done);
    (void)hipFree(c->scratch2);
    delete c;
    return 0;
}

// the scratch buffers of the process-wide context that serves runs WITHOUT a communicator (steppers of one process, no neighbours): a
// stepper that closes hands them back (they are allocated again on demand)
int pdehip_release_scratch(void)
{
    release_scratch(serial_context());
    return 0;
}

// what RCCL itself says about the communicator: out5 = {ncclCommCount, ncclCommUserRank, ncclCommCuDevice, ncclGetVersion, HIP device of the
// calling thread} (-1 where the library lacks the entry point), pci_bus_id of that device ("" if unknown)
int pdehip_comm_info(void *comm, int *out5, char *pci_bus_id, size_t n)
{
    if (!comm || !out5) PDEHIP_FAIL(E_VALUE, "comm_info: NULL pointer");
    Comm *c = static_cast<Comm *>(comm);
    for (int k = 0; k < 5; k++) out5[k] = -1;
    if (g_rccl.CommCount) PDEHIP_NCCL(g_rccl.CommCount(c->comm, &out5[0]));
    if (g_rccl.CommUserRank) PDEHIP_NCCL(g_rccl.CommUserRank(c->comm, &out5[1]));
    if (g_rccl.CommCuDevice) PDEHIP_NCCL(g_rccl.CommCuDevice(c->comm, &out5[2]));
    if (g_rccl.GetVersion) PDEHIP_NCCL(g_rccl.GetVersion(&out5[3]));
    int dev = -1;
    if (hipGetDevice(&dev) == hipSuccess) out5[4] = dev;
    if (pci_bus_id && n > 0) {
        pci_bus_id[0] = 0;
        if (dev >= 0 && hipDeviceGetPCIBusId(pci_bus_id, (int)n, dev) != hipSuccess) pci_bus_id[0] = 0;
    }
    return 0;
}

int pdehip_halo_exchange(void *comm, const pdehip_grid_t *g_local, void *buf_full, int lower, int upper, void *stream)
{
    if (!comm || !buf_full) PDEHIP_FAIL(E_VALUE, "halo_exchange: NULL pointer");
    NGrid n;
    slab::Geo q;
    PDEHIP_TRY(make_geo(g_local, &n, &q));
    HipOps ops{static_cast<Comm *>(comm)};
    return slab::exchange(ops, q, buf_full, lower, upper, stream);
}

int pdehip_allreduce_max(void *comm, double *dev_scalar, void *stream)
{
    if (!comm || !dev_scalar) PDEHIP_FAIL(E_VALUE, "allreduce_max: NULL pointer");
    HipOps ops{static_cast<Comm *>(comm)};
    return ops.allreduce_max(dev_scalar, stream);
}

// nsteps explicit Euler steps of the diffusion equation on one slab, exchange overlapped with the interior kernel
// (loop: slab::euler_run in pdehip_slab_loops.h)
int pdehip_slab_euler_run(void *comm, const pdehip_grid_t *g_local, const pdehip_rhs_t *rhs, int lower, int upper,
                          void *buf_a, void *buf_b, double dt, int64_t nsteps, void **result, void *stream)
{
    if (!comm || !rhs || !buf_a || !buf_b || !result) PDEHIP_FAIL(E_VALUE, "slab_euler_run: NULL pointer");
    if (rhs->kind != PDEHIP_RHS_DIFFUSION) PDEHIP_FAIL(E_NOTIMPL, "slab_euler_run implements the diffusion right-hand side (others: pdehip_slab_euler_sweeps)");
    NGrid n;
    slab::Geo q;
    PDEHIP_TRY(make_geo(g_local, &n, &q));
    HipOps ops{static_cast<Comm *>(comm)};
    if (lower >= 0 || upper >= 0) PDEHIP_TRY(separate_queues(ops.c, as_stream(stream)));
    // (two streams per step: faces that change with time take the single-stream sweeps, pdehip_slab_euler_sweeps)
    if (rhs->bc_program) PDEHIP_FAIL(E_NOTIMPL, "slab_euler_run: time-dependent boundary conditions run through pdehip_slab_euler_sweeps");
    return slab::euler_run(ops, g_local, q, rhs, lower, upper, buf_a, buf_b, dt, nsteps, result, stream);
}

int pdehip_slab_euler2_supported(const pdehip_grid_t *g_local, const pdehip_rhs_t *rhs, int *ok)
{
    if (!g_local || !rhs || !ok) PDEHIP_FAIL(E_VALUE, "slab_euler2_supported: NULL pointer");
    *ok = 0;
    if (rhs->kind != PDEHIP_RHS_DIFFUSION || g_local->ndim != 3 || g_local->shape[0] < 4) return 0;
    bool done = false;
    pdehip_grid_t gs = *g_local;
    gs.shape[0] = 2;   // the smallest launch of the loop
    PDEHIP_TRY(euler2_with_input_bcs(&gs, (const void *)16, (void *)32, rhs->param, 0.0, rhs->bc_c, nullptr, &done, 1, true));
    // sides without a neighbour (face not marked SKIP) must be local first-order faces the kernel can apply itself
    const bool phys0 = rhs->bc_c[0].kind != PDEHIP_BC_SKIP, phys1 = rhs->bc_c[1].kind != PDEHIP_BC_SKIP;
    if (done && (phys0 || phys1)) {
        gs.shape[0] = g_local->shape[0];
        PDEHIP_TRY(euler2_with_input_bcs(&gs, (const void *)16, (void *)32, rhs->param, 0.0, rhs->bc_c, nullptr, &done,
                                         (phys0 && phys1) ? 0 : (phys0 ? 2 : 3), true));
    }
    *ok = done ? 1 : 0;
    return 0;
}

// two Euler steps per sweep on a slab (loop: slab::euler2_run); the private arrays with two halo layers per side live in
// the communicator and are re-used from call to call
int pdehip_slab_euler2_run(void *comm, const pdehip_grid_t *g_local, const pdehip_rhs_t *rhs, int lower, int upper,
                           void *buf_a, void *buf_b, double dt, int64_t nsteps, void **result, void *stream)
{
    if (!comm || !rhs || !buf_a || !buf_b || !result) PDEHIP_FAIL(E_VALUE, "slab_euler2_run: NULL pointer");
    int ok = 0;
    PDEHIP_TRY(pdehip_slab_euler2_supported(g_local, rhs, &ok));
    if (!ok) PDEHIP_FAIL(E_NOTIMPL, "slab_euler2_run: grid or faces are not covered by the two-step kernel");
    Comm *c = static_cast<Comm *>(comm);
    NGrid n;
    slab::Geo q;
    PDEHIP_TRY(make_geo(g_local, &n, &q));
    PDEHIP_TRY(slab_private_arrays(c, g_local, q, 2, as_stream(stream)));
    if (lower >= 0 || upper >= 0) PDEHIP_TRY(separate_queues(c, as_stream(stream)));
    ScratchTurn turn(c, as_stream(stream));
    HipOps ops{c};
    if (rhs->bc_program) PDEHIP_FAIL(E_NOTIMPL, "slab_euler2_run: time-dependent boundary conditions run through pdehip_slab_euler_sweeps");
    return slab::euler2_run(ops, g_local, q, rhs, lower, upper, buf_a, c->ext[0], c->ext[1], dt, nsteps, result, stream);
}

int pdehip_slab_euler4_supported(const pdehip_grid_t *g_local, const pdehip_rhs_t *rhs, int *ok)
{
    if (!g_local || !rhs || !ok) PDEHIP_FAIL(E_VALUE, "slab_euler4_supported: NULL pointer");
    *ok = 0;
    if (g_local->shape[0] < 8) return 0;
    return pdehip_slab_euler2_supported(g_local, rhs, ok);   // the same launches, other layer ranges
}

// four steps per exchange on a slab (loop: slab::euler4_run); the private arrays with four halo layers per side live in the communicator
int pdehip_slab_euler4_run(void *comm, const pdehip_grid_t *g_local, const pdehip_rhs_t *rhs, int lower, int upper,
                           void *buf_a, void *buf_b, double dt, int64_t nsteps, void **result, void *stream)
{
    if (!comm || !rhs || !buf_a || !buf_b || !result) PDEHIP_FAIL(E_VALUE, "slab_euler4_run: NULL pointer");
    if (nsteps < 0) PDEHIP_FAIL(E_VALUE, "slab_euler4_run: negative step count");
    int ok = 0;
    PDEHIP_TRY(pdehip_slab_euler4_supported(g_local, rhs, &ok));
    if (!ok) PDEHIP_FAIL(E_NOTIMPL, "slab_euler4_run: grid or faces are not covered by the two-step kernel, or fewer than 8 local layers");
    if (rhs->bc_program) PDEHIP_FAIL(E_NOTIMPL, "slab_euler4_run: time-dependent boundary conditions run through pdehip_slab_euler_sweeps");
    Comm *c = static_cast<Comm *>(comm);
    NGrid n;
    slab::Geo q;
    PDEHIP_TRY(make_geo(g_local, &n, &q));
    HipOps ops{c};
    const bool piped = ops.deep_mode() >= 3;
    PDEHIP_TRY(slab_private_arrays(c, g_local, q, 4, as_stream(stream), piped ? 4 : 2));
    if (lower >= 0 || upper >= 0) PDEHIP_TRY(separate_queues(c, as_stream(stream)));
    ScratchTurn turn(c, as_stream(stream));
    if (piped) return slab::euler4p_run(ops, g_local, q, rhs, lower, upper, buf_a, c->ext, dt, nsteps, result, stream);
    return slab::euler4_run(ops, g_local, q, rhs, lower, upper, buf_a, c->ext[0], c->ext[1], dt, nsteps, result, stream);
}

int pdehip_slab_ch_supported(const pdehip_grid_t *g_local, const pdehip_rhs_t *rhs, int *ok)
{
    if (!g_local || !rhs || !ok) PDEHIP_FAIL(E_VALUE, "slab_ch_supported: NULL pointer");
    *ok = 0;
    if (rhs->kind != PDEHIP_RHS_CAHN_HILLIARD || g_local->ndim != 3 || g_local->shape[0] < 2) return 0;
    bool done = false;
    PDEHIP_TRY(cahn_hilliard_fused(g_local, (const void *)16, (void *)32, rhs->param, 0.0, false, rhs->bc_c, rhs->bc_mu, nullptr,
                                   &done, 1, true));
    // sides without a neighbour (faces not marked SKIP): local first-order faces of c AND mu that the kernel applies itself
    const bool phys0 = rhs->bc_c[0].kind != PDEHIP_BC_SKIP || rhs->bc_mu[0].kind != PDEHIP_BC_SKIP;
    const bool phys1 = rhs->bc_c[1].kind != PDEHIP_BC_SKIP || rhs->bc_mu[1].kind != PDEHIP_BC_SKIP;
    if (done && (phys0 || phys1))
        PDEHIP_TRY(cahn_hilliard_fused(g_local, (const void *)16, (void *)32, rhs->param, 0.0, false, rhs->bc_c, rhs->bc_mu, nullptr,
                                       &done, (phys0 && phys1) ? 0 : (phys0 ? 2 : 3), true));
    *ok = done ? 1 : 0;
    return 0;
}

// Cahn-Hilliard right-hand side on a slab in ONE sweep after ONE exchange of two layers of c (slab::rhs_sweep with
// F_FUSED_CH); `c_ext` / `out_ext` are the arrays with two halo layers per side
int pdehip_slab_ch_sweep(void *comm, const pdehip_grid_t *g_local, const pdehip_rhs_t *rhs, int lower, int upper,
                         void *c_ext, void *out_ext, double dt, int euler, void *stream)
{
    if (!comm || !rhs || !c_ext || !out_ext) PDEHIP_FAIL(E_VALUE, "slab_ch_sweep: NULL pointer");
    if (rhs->kind != PDEHIP_RHS_CAHN_HILLIARD) PDEHIP_FAIL(E_VALUE, "slab_ch_sweep: not a Cahn-Hilliard right-hand side");
    NGrid n;
    slab::Geo q;
    PDEHIP_TRY(make_geo(g_local, &n, &q));
    HipOps ops{static_cast<Comm *>(comm)};
    return slab::rhs_sweep(ops, g_local, q, rhs, lower, upper, slab::F_FUSED_CH, slab::layer(c_ext, q, 1), slab::layer(out_ext, q, 1), dt,
                           euler != 0, nullptr, stream);
}

// which of the PDEHIP_SLAB_* paths this rank could take for (grid, right-hand side, neighbours): the caller ANDs the
// answers of all ranks.  Faces towards neighbours must already be marked PDEHIP_BC_SKIP in rhs->bc_c / bc_mu or are
// treated as exchanged through lower / upper.
int pdehip_slab_flags_supported(const pdehip_grid_t *g_local, const pdehip_rhs_t *rhs, int lower, int upper, int *flags)
{
    if (!g_local || !flags) PDEHIP_FAIL(E_VALUE, "slab_flags_supported: NULL pointer");
    PDEHIP_TRY(check_rhs(rhs));
    *flags = 0;
    NGrid n;
    PDEHIP_TRY(norm_grid(g_local, &n));
    pdehip_rhs_t r = *rhs;
    slab::local_faces(rhs->bc_c, lower, upper, r.bc_c);
    slab::local_faces(rhs->bc_mu, lower, upper, r.bc_mu);
    if (rhs->kind == PDEHIP_RHS_CAHN_HILLIARD) {
        bool done = false;
        if (n.ndim >= 2 && g_local->shape[0] >= 2)
            PDEHIP_TRY(cahn_hilliard_fused(g_local, (const void *)16, (void *)32, rhs->param, 0.0, false, r.bc_c, r.bc_mu, nullptr, &done,
                                           slab::xends(lower, upper), true));
        if (done) *flags |= PDEHIP_SLAB_FUSED_CH;
        // the stage epilogue rides on the two-level sweep: same coverage with a stage descriptor (dry run)
        if (done) {
            StageFuse sf;
            memset(&sf, 0, sizeof(sf));
            sf.y = (const void *)48; sf.out2 = (void *)64;
            bool d2 = false;
            PDEHIP_TRY(cahn_hilliard_fused(g_local, (const void *)16, (void *)32, rhs->param, 0.0, false, r.bc_c, r.bc_mu, nullptr, &d2,
                                           slab::xends(lower, upper), true, &sf));
            if (d2) *flags |= PDEHIP_SLAB_FUSED_STAGE;
        }
    } else if (laplace_can_fuse_bcs(n, (const void *)16, (const void *)32, nullptr)) {
        *flags |= PDEHIP_SLAB_FUSED_STAGE;
    }
    return 0;
}

int pdehip_slab_rhs_scaled(void *comm, const pdehip_grid_t *g_local, const pdehip_rhs_t *rhs, int lower, int upper, int flags,
                           void *y_full, void *k_out_full, double dt, void *stream)
{
    if (!y_full || !k_out_full) PDEHIP_FAIL(E_VALUE, "slab_rhs_scaled: NULL pointer");
    PDEHIP_TRY(check_rhs(rhs));
    Comm *c;
    PDEHIP_TRY(context(comm, lower, upper, &c));
    NGrid n;
    slab::Geo q;
    PDEHIP_TRY(make_geo(g_local, &n, &q));
    HipOps ops{c};
    return slab::rhs_sweep(ops, g_local, q, rhs, lower, upper, flags, y_full, k_out_full, dt, false, nullptr, stream, rhs->t);
}

int pdehip_slab_euler_sweeps(void *comm, const pdehip_grid_t *g_local, const pdehip_rhs_t *rhs, int lower, int upper, int flags,
                             void *buf_a, void *buf_b, double dt, int64_t nsteps, void **result, void *stream)
{
    if (!buf_a || !buf_b || !result) PDEHIP_FAIL(E_VALUE, "slab_euler_sweeps: NULL pointer");
    if (nsteps < 0) PDEHIP_FAIL(E_VALUE, "slab_euler_sweeps: negative step count");
    PDEHIP_TRY(check_rhs(rhs));
    Comm *c;
    PDEHIP_TRY(context(comm, lower, upper, &c));
    NGrid n;
    slab::Geo q;
    PDEHIP_TRY(make_geo(g_local, &n, &q));
    HipOps ops{c};
    return slab::euler_sweeps(ops, g_local, q, rhs, lower, upper, flags, buf_a, buf_b, dt, nsteps, result, stream);
}

int pdehip_slab_rk4_run(void *comm, const pdehip_grid_t *g_local, const pdehip_rhs_t *rhs, int lower, int upper, int flags,
                        void *y_full, void *const *work5_host, double dt, int64_t nsteps, void *stream)
{
    if (!y_full || !work5_host) PDEHIP_FAIL(E_VALUE, "slab_rk4_run: NULL pointer");
    if (nsteps < 0) PDEHIP_FAIL(E_VALUE, "slab_rk4_run: negative step count");
    PDEHIP_TRY(check_rhs(rhs));
    Comm *c;
    PDEHIP_TRY(context(comm, lower, upper, &c));
    NGrid n;
    slab::Geo q;
    PDEHIP_TRY(make_geo(g_local, &n, &q));
    HipOps ops{c};
    for (int64_t s = 0; s < nsteps; s++)
        PDEHIP_TRY(slab::rk4_step(ops, g_local, q, rhs, lower, upper, flags, y_full, work5_host, dt, stream, rhs->t + (double)s * dt));
    return 0;
}

int pdehip_slab_rkf45_run(void *comm, const pdehip_grid_t *g_local, const pdehip_rhs_t *rhs, int lower, int upper, int flags,
                          void *y_full, void *ynew_full, void *const *work7_host, double *err_dev, pdehip_adaptive_t *ctl,
                          void **result, void *stream)
{
    if (!y_full || !ynew_full || !work7_host || !err_dev || !ctl || !result) PDEHIP_FAIL(E_VALUE, "slab_rkf45_run: NULL pointer");
    if (!(ctl->tolerance > 0) || !(ctl->dt > 0)) PDEHIP_FAIL(E_VALUE, "slab_rkf45_run: tolerance and dt must be positive");
    PDEHIP_TRY(check_rhs(rhs));
    Comm *c;
    PDEHIP_TRY(context(comm, lower, upper, &c));
    NGrid n;
    slab::Geo q;
    PDEHIP_TRY(make_geo(g_local, &n, &q));
    HipOps ops{c};
    return slab::rkf45_run(ops, g_local, q, rhs, lower, upper, flags, y_full, ynew_full, work7_host, err_dev, ctl, result, stream);
}

int pdehip_slab_euler_adaptive_run(void *comm, const pdehip_grid_t *g_local, const pdehip_rhs_t *rhs, int lower, int upper, int flags,
                                   void *y_full, void *ynew_full, void *const *work3_host, double *err_dev, pdehip_adaptive_t *ctl,
                                   void **result, void *stream)
{
    if (!y_full || !ynew_full || !work3_host || !err_dev || !ctl || !result) PDEHIP_FAIL(E_VALUE, "slab_euler_adaptive_run: NULL pointer");
    if (!(ctl->tolerance > 0) || !(ctl->dt > 0)) PDEHIP_FAIL(E_VALUE, "slab_euler_adaptive_run: tolerance and dt must be positive");
    PDEHIP_TRY(check_rhs(rhs));
    Comm *c;
    PDEHIP_TRY(context(comm, lower, upper, &c));
    NGrid n;
    slab::Geo q;
    PDEHIP_TRY(make_geo(g_local, &n, &q));
    HipOps ops{c};
    return slab::euler_adaptive_run(ops, g_local, q, rhs, lower, upper, flags, y_full, ynew_full, work3_host, err_dev, ctl, result, stream);
}

int pdehip_euler_adaptive_run(const pdehip_grid_t *g, const pdehip_rhs_t *rhs, void *y_full, void *ynew_full, void *const *work3_host,
                              double *err_dev, pdehip_adaptive_t *ctl, void **result, void *stream)
{
    int flags = 0;
    PDEHIP_TRY(pdehip_slab_flags_supported(g, rhs, -1, -1, &flags));
    return pdehip_slab_euler_adaptive_run(nullptr, g, rhs, -1, -1, flags, y_full, ynew_full, work3_host, err_dev, ctl, result, stream);
}

// ---- block decomposition (pdehip_block_loops.h) ------------------------------------------------------------------------------
static int block_context(void *comm, const pdehip_grid_t *g_local, const int *nb6, Comm **c, NGrid *n, block::Geo *q)
{
    if (!g_local || !nb6) PDEHIP_FAIL(E_VALUE, "block: NULL pointer");
    bool any = false;
    for (int i = 0; i < 2 * g_local->ndim; i++) any |= nb6[i] >= 0;
    *c = static_cast<Comm *>(comm);
    if (!*c) {
        if (any) PDEHIP_FAIL(E_VALUE, "a block with neighbours needs a communicator");
        *c = serial_context();
    }
    PDEHIP_TRY(ensure_streams(*c));
    PDEHIP_TRY(norm_grid(g_local, n));
    return make_block(*c, g_local, *n, nb6, q);
}

int pdehip_block_exchange(void *comm, const pdehip_grid_t *g_local, const int *nb6, void *buf_full, void *stream)
{
    if (!buf_full) PDEHIP_FAIL(E_VALUE, "block_exchange: NULL pointer");
    Comm *c;
    NGrid n;
    block::Geo q;
    PDEHIP_TRY(block_context(comm, g_local, nb6, &c, &n, &q));
    ScratchTurn turn(c, as_stream(stream));
    HipOps ops{c};
    ops.bn = &n;
    return block::exchange(ops, q, buf_full, stream);
}

int pdehip_block_run(void *comm, const pdehip_grid_t *g_local, const pdehip_rhs_t *rhs, const int *nb6, int fuse_stage, int scheme, void *y_full,
                     void *ynew_full, void *const *work_host, double *err_dev, double dt, int64_t nsteps, pdehip_adaptive_t *ctl, void **result,
                     void *stream)
{
    if (!y_full || !result) PDEHIP_FAIL(E_VALUE, "block_run: NULL pointer");
    if (scheme < 0 || scheme > 3) PDEHIP_FAIL(E_VALUE, "block_run: scheme 0 (Euler), 1 (RK4), 2 (adaptive RKF45) or 3 (adaptive Euler)");
    if (scheme != 1 && !ynew_full) PDEHIP_FAIL(E_VALUE, "block_run: the scheme needs a second state array");
    if (scheme >= 1 && !work_host) PDEHIP_FAIL(E_VALUE, "block_run: the scheme needs work arrays");
    if (scheme >= 2 && (!ctl || !err_dev || !(ctl->tolerance > 0) || !(ctl->dt > 0))) PDEHIP_FAIL(E_VALUE, "block_run: the adaptive loop needs ctl, err_dev, tolerance > 0, dt > 0");
    if (nsteps < 0) PDEHIP_FAIL(E_VALUE, "block_run: negative step count");
    PDEHIP_TRY(check_rhs(rhs));
    Comm *c;
    NGrid n;
    block::Geo q;
    PDEHIP_TRY(block_context(comm, g_local, nb6, &c, &n, &q));
    ScratchTurn turn(c, as_stream(stream));
    HipOps ops{c};
    ops.bn = &n;
    return block::run(ops, g_local, q, rhs, fuse_stage != 0, scheme, y_full, ynew_full, work_host, err_dev, dt, nsteps, ctl, result, stream);
}

}  // extern "C"

// ======================================================================================================================================
// The FAST block decomposition (pdehip_block2_loops.h): two Euler steps per sweep on a box, one message per neighbouring rank, the
// exchange hidden behind the next sweep, the rim recomputed when the halos have landed.
// ======================================================================================================================================
#include "pdehip_block2_loops.h"

namespace {

// ---- box copies: pack / unpack of all regions of an exchange in ONE launch, and the own cells between the state array and `ext` ----
struct BoxJob {
    const void *src;
    void *dst;
    long sbase, s0, s1;   // element offset of the box's first cell and the pitches of its two slow axes (source)
    long dbase, d0, d1;   // ... destination
    long n1, nv;          // rows per plane, VECTORS per row
    long start;           // first work item of this job
};
struct BoxJobs {
    int n;
    long total;
    BoxJob j[block2::kMaxRegions];
};
template <typename T, int VEC>
__global__ void __launch_bounds__(256) box_copy_kernel(BoxJobs a)
{
    typedef T V __attribute__((ext_vector_type(VEC)));
    for (long t = blockIdx.x * 256L + threadIdx.x; t < a.total; t += (long)gridDim.x * 256L) {
        int k = 0;
#pragma unroll
        for (int q = 1; q < block2::kMaxRegions; q++)
            if (q < a.n && t >= a.j[q].start) k = q;
        const BoxJob &b = a.j[k];
        const long loc = t - b.start;
        const long v = loc % b.nv, r = loc / b.nv;
        const long j = r % b.n1, i = r / b.n1;
        const T *s = (const T *)b.src + b.sbase + i * b.s0 + j * b.s1 + v * VEC;
        T *d = (T *)b.dst + b.dbase + i * b.d0 + j * b.d1 + v * VEC;
        *(V *)d = *(const V *)s;
    }
}

// ---- the rim: two Euler steps for the cells less than two layers from a cut face, recomputed from the sweep's input ------------------
// No march: a wave owns an output tile of 2 planes x 2 rows x 60 lanes and requests every operand of it at once (24 row vectors of the
// input: the tile widened by two cells, without the corners a 7-point stencil never reaches).  Neighbours along the fastest axis are the
// neighbouring lanes (DPP wave shifts); lanes 0, 1, 62, 63 only feed their neighbours (level 1 is valid on lanes 1 .. 62, level 2 on
// 2 .. 61).  Same expressions in the same order as the two-level sweep (pdehip_march2.inc: `laplace`, `update`; cartesian.py:220-227,
// euler.py:172-175), the intermediate level rounded to the storage type: bit-identical to it and to two single steps.
struct Rim2Args {
    const void *in;
    void *out;
    long n0, n1, n2;
    long p0, p1, off;       // pitches of `ext`, element offset of own cell (0, 0, 0)
    int wrap[3];            // the axis wraps (one block along a periodic axis); else two real halo layers on either side
    double sx, sy, sz, s1, s2;
    int njobs;
    struct Job { long i0, j0, ni, nj; long start; } job[6];   // box of own cells, tiles of 2 x 2 (the last one moved back)
    long nzseg;             // lane segments per row
    long total;             // wave tiles
    // DIRECT (the fastest axis not cut): the halo cells are read from the RECEIVE buffer where the messages land (no unpack launch) and the
    // results are stored into the SEND buffer as well (no pack launch) - a pair of steps then has rim -> send / receive on its critical
    // path instead of rim -> pack -> send / receive -> unpack.  Regions by direction code (dx + 1) * 3 + (dy + 1); rows of n2 cells.
    const void *recv;
    void *send;
    struct Reg { long off, lo0, lo1, rows; } rreg[9], sreg[9];   // off < 0: no such region
};

template <typename T, int VEC, bool DIRECT = false>
__global__ void __launch_bounds__(256) rim2_kernel(Rim2Args a)
{
    typedef typename VecT<T, VEC>::type V;
    const int lane = threadIdx.x & 63;
    const long w = blockIdx.x * 4L + (threadIdx.x >> 6);
    if (w >= a.total) return;
    int k = 0;
#pragma unroll
    for (int q = 1; q < 6; q++)
        if (q < a.njobs && w >= a.job[q].start) k = q;
    const long loc = w - a.job[k].start;
    const long zseg = loc % a.nzseg;
    const long ti = (loc / a.nzseg) % ((a.job[k].ni + 1) / 2), tj = (loc / a.nzseg) / ((a.job[k].ni + 1) / 2);
    long i0 = a.job[k].i0 + 2 * ti, j0 = a.job[k].j0 + 2 * tj;
    if (i0 + 2 > a.job[k].i0 + a.job[k].ni) i0 = a.job[k].i0 + a.job[k].ni - 2;
    if (j0 + 2 > a.job[k].j0 + a.job[k].nj) j0 = a.job[k].j0 + a.job[k].nj - 2;
    const long nv = a.n2 / VEC;
    const long vi = zseg * 60 - 2 + lane;     // the lane's vector of the row
    long vs = vi;
    if (a.wrap[2]) vs = ((vi % nv) + nv) % nv;
    else vs = vi < -1 ? -1 : (vi > nv ? nv : vi);   // one vector of halo on either side; lanes further out only feed lanes that store nothing
    auto xs = [&](long i) { return a.wrap[0] ? ((i % a.n0) + a.n0) % a.n0 : i; };
    auto ys = [&](long j) { return a.wrap[1] ? ((j % a.n1) + a.n1) % a.n1 : j; };
    const T *base = (const T *)a.in + a.off + vs * VEC;

    V u0[6][6];
#pragma unroll
    for (int pp = 0; pp < 6; pp++)
#pragma unroll
        for (int rr = 0; rr < 6; rr++) {
            const int dp = pp < 2 ? 2 - pp : (pp > 3 ? pp - 3 : 0), dr = rr < 2 ? 2 - rr : (rr > 3 ? rr - 3 : 0);
            if (dp + dr > 2) continue;
            if constexpr (DIRECT) {
                const long i = i0 - 2 + pp, j = j0 - 2 + rr;
                const int di = a.wrap[0] ? 0 : (i < 0 ? -1 : (i >= a.n0 ? 1 : 0)), dj = a.wrap[1] ? 0 : (j < 0 ? -1 : (j >= a.n1 ? 1 : 0));
                if (di != 0 || dj != 0) {   // uniform: a halo row - out of the message it arrived in
                    const Rim2Args::Reg &r = a.rreg[(di + 1) * 3 + (dj + 1)];
                    u0[pp][rr] = *(const V *)((const T *)a.recv + r.off + ((xs(i) - r.lo0) * r.rows + (ys(j) - r.lo1)) * a.n2 + vs * VEC);
                    continue;
                }
            }
            u0[pp][rr] = *(const V *)(base + xs(i0 - 2 + pp) * a.p0 + ys(j0 - 2 + rr) * a.p1);
        }
    auto step = [&](double xm, double xp, double up, double dn, double left, double right, double cen) {
        const double vm = 2 * cen;
        const double lx = (xm - vm + xp) * a.sx;
        const double ly = (up - vm + dn) * a.sy;
        const double lz = (left - vm + right) * a.sz;
        const double lap = lx + ly + lz;
        return cen + a.s2 * (a.s1 * lap);
    };
    V u1[4][4];
#pragma unroll
    for (int q = 0; q < 4; q++)
#pragma unroll
        for (int s = 0; s < 4; s++) {
            if (((q == 0 || q == 3) ? 1 : 0) + ((s == 0 || s == 3) ? 1 : 0) > 1) continue;
            const V cc = u0[q + 1][s + 1];
            const double zl = wave_shr1(0.0, (double)cc[VEC - 1]), zr = wave_shl1(0.0, (double)cc[0]);
            V res;
#pragma unroll
            for (int e = 0; e < VEC; e++) {
                const double left = (e == 0) ? zl : (double)cc[e > 0 ? e - 1 : 0];
                const double right = (e == VEC - 1) ? zr : (double)cc[e < VEC - 1 ? e + 1 : e];
                res[e] = (T)step((double)u0[q][s + 1][e], (double)u0[q + 2][s + 1][e], (double)u0[q + 1][s][e], (double)u0[q + 1][s + 2][e], left, right, (double)cc[e]);
            }
            u1[q][s] = res;
        }
    T *ob = (T *)a.out + a.off + vi * VEC;
    const bool store = lane >= 2 && lane <= 61 && vi >= 0 && vi < nv;
#pragma unroll
    for (int q = 0; q < 2; q++)
#pragma unroll
        for (int s = 0; s < 2; s++) {
            const V cc = u1[q + 1][s + 1];
            const double zl = wave_shr1(0.0, (double)cc[VEC - 1]), zr = wave_shl1(0.0, (double)cc[0]);
            V res;
#pragma unroll
            for (int e = 0; e < VEC; e++) {
                const double left = (e == 0) ? zl : (double)cc[e > 0 ? e - 1 : 0];
                const double right = (e == VEC - 1) ? zr : (double)cc[e < VEC - 1 ? e + 1 : e];
                res[e] = (T)step((double)u1[q][s + 1][e], (double)u1[q + 2][s + 1][e], (double)u1[q + 1][s][e], (double)u1[q + 1][s + 2][e], left, right, (double)cc[e]);
            }
            if (store) *(V *)(ob + (i0 + q) * a.p0 + (j0 + s) * a.p1) = res;
            if constexpr (DIRECT) {
                // ... and into every message this cell travels in: the face(s) it lies behind and the edge between them
                const long i = i0 + q, j = j0 + s;
                const int ox = a.wrap[0] ? 0 : (i < 2 ? -1 : (i >= a.n0 - 2 ? 1 : 0)), oy = a.wrap[1] ? 0 : (j < 2 ? -1 : (j >= a.n1 - 2 ? 1 : 0));
                auto put = [&](int dx, int dy) {
                    const Rim2Args::Reg &r = a.sreg[(dx + 1) * 3 + (dy + 1)];
                    if (store && r.off >= 0) *(V *)((T *)a.send + r.off + ((i - r.lo0) * r.rows + (j - r.lo1)) * a.n2 + vi * VEC) = res;
                };
                if (ox) put(ox, 0);
                if (oy) put(0, oy);
                if (ox && oy) put(ox, oy);
            }
        }
}

// The rim behind a cut face of the FASTEST axis: two columns of every row.  Lanes along that axis would idle (2 cells = one lane's vector), so
// the layout is transposed: a lane stands for a ROW and holds the strip k0-2 .. k0+3 of six planes (three 2-element vectors per plane and
// row); row neighbours are the neighbouring lanes (DPP), plane neighbours other registers.  A wave = 2 planes x 60 rows (lanes 0, 1, 62, 63
// only feed their neighbours).  Same expressions in the same order as rim2_kernel.
struct RimzArgs {
    const void *in;
    void *out;
    long n0, n1, n2;
    long p0, p1, off;
    int wrap[3];            // (wrap[2] is 0: this kernel exists for cut fastest axes)
    double sx, sy, sz, s1, s2;
    long nrseg;             // row segments of 60 rows
    long total;             // wave tiles: 2 sides x plane pairs x row segments
};
template <typename T>
__global__ void __launch_bounds__(256) rimz_kernel(RimzArgs a)
{
    typedef T V2 __attribute__((ext_vector_type(2)));
    const int lane = threadIdx.x & 63;
    const long w = blockIdx.x * 4L + (threadIdx.x >> 6);
    if (w >= a.total) return;
    const long npair = (a.n0 + 1) / 2;
    const long rseg = w % a.nrseg, ti = (w / a.nrseg) % npair, side = w / (a.nrseg * npair);
    long i0 = 2 * ti;
    if (i0 + 2 > a.n0) i0 = a.n0 - 2;
    const long k0 = side ? a.n2 - 2 : 0;
    const long jr = rseg * 60 - 2 + lane;      // the lane's row
    long js = jr;
    if (a.wrap[1]) js = ((jr % a.n1) + a.n1) % a.n1;
    else js = jr < -2 ? -2 : (jr > a.n1 + 1 ? a.n1 + 1 : jr);   // two halo rows on either side; lanes further out store nothing
    auto xs = [&](long i) { return a.wrap[0] ? ((i % a.n0) + a.n0) % a.n0 : i; };
    const T *base = (const T *)a.in + a.off + js * a.p1 + (k0 - 2);
    double u0[6][6];
#pragma unroll
    for (int pp = 0; pp < 6; pp++) {
        const T *row = base + xs(i0 - 2 + pp) * a.p0;
#pragma unroll
        for (int v = 0; v < 3; v++) {
            // strip cells 2v, 2v+1; skipped when both lie in a corner of the diamond
            const int dp = pp < 2 ? 2 - pp : (pp > 3 ? pp - 3 : 0);
            const int dk0 = (2 * v) < 2 ? 2 - 2 * v : ((2 * v) > 3 ? 2 * v - 3 : 0), dk1 = (2 * v + 1) < 2 ? 1 - 2 * v : ((2 * v + 1) > 3 ? 2 * v - 2 : 0);
            if (dp + dk0 > 2 && dp + dk1 > 2) { u0[pp][2 * v] = 0; u0[pp][2 * v + 1] = 0; continue; }
            const V2 x = *(const V2 *)(row + 2 * v);
            u0[pp][2 * v] = (double)x[0];
            u0[pp][2 * v + 1] = (double)x[1];
        }
    }
    auto step = [&](double xm, double xp, double up, double dn, double left, double right, double cen) {
        const double vm = 2 * cen;
        const double lx = (xm - vm + xp) * a.sx;
        const double ly = (up - vm + dn) * a.sy;
        const double lz = (left - vm + right) * a.sz;
        const double lap = lx + ly + lz;
        return cen + a.s2 * (a.s1 * lap);
    };
    double u1[4][4];
#pragma unroll
    for (int q = 0; q < 4; q++)
#pragma unroll
        for (int kk = 0; kk < 4; kk++) {
            u1[q][kk] = 0;
            if (((q == 0 || q == 3) ? 1 : 0) + ((kk == 0 || kk == 3) ? 1 : 0) > 1) continue;
            const double cen = u0[q + 1][kk + 1];
            const double up = wave_shr1(0.0, cen), dn = wave_shl1(0.0, cen);
            u1[q][kk] = (double)(T)step(u0[q][kk + 1], u0[q + 2][kk + 1], up, dn, u0[q + 1][kk], u0[q + 1][kk + 2], cen);
        }
    T *ob = (T *)a.out + a.off + jr * a.p1 + k0;
    const bool store = lane >= 2 && lane <= 61 && jr >= 0 && jr < a.n1;
#pragma unroll
    for (int q = 0; q < 2; q++) {
        V2 res;
#pragma unroll
        for (int kk = 0; kk < 2; kk++) {
            const double cen = u1[q + 1][kk + 1];
            const double up = wave_shr1(0.0, cen), dn = wave_shl1(0.0, cen);
            res[kk] = (T)step(u1[q][kk + 1], u1[q + 2][kk + 1], up, dn, u1[q + 1][kk], u1[q + 1][kk + 2], cen);
        }
        if (store) *(V2 *)(ob + (i0 + q) * a.p0) = res;
    }
}

int slab_rim2(const pdehip_grid_t *gs, const void *in, void *out, double D, double dt, const pdehip_bc_face_t *faces, void *st, bool *done)
{
    *done = false;
    static const bool off = getenv("PDEHIP_SLAB_RIM") && getenv("PDEHIP_SLAB_RIM")[0] == '0';   // A/B aid
    if (off || gs->ndim != 3 || gs->shape[0] < 4 || gs->shape[1] < 2) return 0;
    NGrid n;
    PDEHIP_TRY(norm_grid(gs, &n));
    const long vec = 16 / elem_size(n.dtype);
    if (n.n[2] % vec) return 0;
    for (int a = 1; a < 3; a++)
        for (int side = 0; side < 2; side++) {
            const pdehip_bc_face_t &r = faces[2 * a + side];
            if (r.kind != PDEHIP_BC_ORDER1 || r.flags != 0 || r.index1 != (side ? 0 : n.n[a] - 1) || r.const_v != 0.0 || r.factor1 != 1.0) return 0;   // periodic rows / columns only
        }
    Rim2Args a;
    memset(&a, 0, sizeof(a));
    a.in = in; a.out = out;
    a.n0 = n.n[0]; a.n1 = n.n[1]; a.n2 = n.n[2];
    a.p0 = n.p[0]; a.p1 = n.p[1]; a.off = n.off;
    a.wrap[0] = 0; a.wrap[1] = 1; a.wrap[2] = 1;
    a.sx = n.lap_scale[0]; a.sy = n.lap_scale[1]; a.sz = n.lap_scale[2];
    a.s1 = D; a.s2 = dt;
    a.nzseg = (n.n[2] / vec + 59) / 60;
    const long per = ((n.n[1] + 1) / 2) * a.nzseg;
    a.job[0] = {0, 0, 2, n.n[1], 0};
    a.job[1] = {n.n[0] - 2, 0, 2, n.n[1], per};
    a.njobs = 2;
    a.total = 2 * per;
    const unsigned blocks = (unsigned)((a.total + 3) / 4);
    if (n.dtype == PDEHIP_F64) hipLaunchKernelGGL((rim2_kernel<double, 2>), dim3(blocks), dim3(256), 0, as_stream(st), a);
    else hipLaunchKernelGGL((rim2_kernel<float, 4>), dim3(blocks), dim3(256), 0, as_stream(st), a);
    PDEHIP_HIP(hipGetLastError());
    *done = true;
    return 0;
}

struct Block2Ctx {
    void *ext[2] = {nullptr, nullptr};
    size_t ext_bytes = 0;
    void *msg[2] = {nullptr, nullptr};   // send / receive buffer (all peers, contiguous)
    size_t msg_bytes = 0;
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};   // block2::EV_*, [3]: hand-over between the caller's stream and the masked one
    // PDEHIP_BLOCK2_CUS=<R>: the halo stream (rim / pack / RCCL / unpack) owns R compute units, the sweeps the other 256 - R.  The waves
    // of a sweep live for the whole sweep and hold every register of the chip: a small kernel enqueued next to it waits for its END
    // (kernel timeline in profiles/r05_probe_block.md: the 7 us pack took 57 us), so without the partition nothing overlaps.
    hipStream_t comp_masked = nullptr, halo_masked = nullptr;
    int reserved = -1;
};
// (one context per communicator, owned by it and released with it; the serial context of a process without neighbours has its own)
Block2Ctx *block2_ctx(Comm *c)
{
    if (!c->b2) c->b2 = new Block2Ctx();
    return c->b2;
}

struct HipOps2 {
    Comm *c;
    Block2Ctx *x;
    const pdehip_grid_t *g_box;
    NGrid ne;              // the grid of `ext`
    long own_off;          // element offset of own cell (0, 0, 0) in `ext`
    const pdehip_bc_face_t *faces;
    double D, dt;
    size_t es;
    void *halo_st = nullptr;
    void *halo() { return halo_st; }
    size_t esz() const { return es; }
    void *msg(bool send, size_t elem_off) { return static_cast<char *>(x->msg[send ? 0 : 1]) + elem_off * es; }
    int record2(int ev, void *st) { PDEHIP_HIP(hipEventRecord(x->ev[ev], as_stream(st))); return 0; }
    int wait2(void *st, int ev) { PDEHIP_HIP(hipStreamWaitEvent(as_stream(st), x->ev[ev], 0)); return 0; }
    int group_start() { PDEHIP_NCCL(g_rccl.GroupStart()); return 0; }
    int group_end() { PDEHIP_NCCL(g_rccl.GroupEnd()); return 0; }
    int send(const void *p, size_t bytes, int peer, void *st) { PDEHIP_NCCL(g_rccl.Send(p, bytes, ncclInt8, peer, c->comm, as_stream(st))); return 0; }
    int recv(void *p, size_t bytes, int peer, void *st) { PDEHIP_NCCL(g_rccl.Recv(p, bytes, ncclInt8, peer, c->comm, as_stream(st))); return 0; }

    long elem_of(const long *lo) const { return own_off + lo[0] * ne.p[0] + lo[1] * ne.p[1] + lo[2]; }
    int launch_copy(BoxJobs &jobs, int vec, void *st)
    {
        if (!jobs.total) return 0;
        const unsigned blocks = (unsigned)((jobs.total + 255) / 256 < 4096 ? (jobs.total + 255) / 256 : 4096);
        hipStream_t s = as_stream(st);
        if (es == 8) {
            if (vec == 2) hipLaunchKernelGGL((box_copy_kernel<double, 2>), dim3(blocks), dim3(256), 0, s, jobs);
            else hipLaunchKernelGGL((box_copy_kernel<double, 1>), dim3(blocks), dim3(256), 0, s, jobs);
        } else {
            if (vec == 4) hipLaunchKernelGGL((box_copy_kernel<float, 4>), dim3(blocks), dim3(256), 0, s, jobs);
            else if (vec == 2) hipLaunchKernelGGL((box_copy_kernel<float, 2>), dim3(blocks), dim3(256), 0, s, jobs);
            else hipLaunchKernelGGL((box_copy_kernel<float, 1>), dim3(blocks), dim3(256), 0, s, jobs);
        }
        PDEHIP_HIP(hipGetLastError());
        return 0;
    }
    // widest vector (elements) that every box of the list allows on both sides
    static int common_vec(const BoxJobs &jobs, const long *n2s, int maxvec)
    {
        int vec = maxvec;
        for (int k = 0; k < jobs.n; k++) {
            const BoxJob &b = jobs.j[k];
            while (vec > 1 && (n2s[k] % vec || b.sbase % vec || b.dbase % vec || b.s0 % vec || b.s1 % vec || b.d0 % vec || b.d1 % vec)) vec /= 2;
        }
        return vec;
    }
    // DIRECT (schedules 1 / 2, the fastest axis not cut): the rim kernel reads the halo cells out of the receive buffer and writes its
    // results into the send buffer itself - only the very first exchange of a run packs, nothing is ever unpacked
    bool direct = false, send_ready = false;
    // all regions of an exchange: own cells -> send buffer (is_pack) / receive buffer -> halo cells
    int pack(const block2::Plan &p, void *ext, bool is_pack, void *st)
    {
        if (direct && (!is_pack || send_ready)) return 0;
        BoxJobs jobs;
        long n2s[block2::kMaxRegions];
        const int nreg = is_pack ? p.nsend : p.nrecv;
        jobs.n = nreg;
        for (int k = 0; k < nreg; k++) {
            const block2::Region &r = is_pack ? p.send[k] : p.recv[k];
            BoxJob &b = jobs.j[k];
            const long e = elem_of(r.box.lo);
            const long m0 = r.box.n[1] * r.box.n[2], m1 = r.box.n[2];
            if (is_pack) { b.src = ext; b.dst = x->msg[0]; b.sbase = e; b.s0 = ne.p[0]; b.s1 = ne.p[1]; b.dbase = (long)r.offset; b.d0 = m0; b.d1 = m1; }
            else { b.src = x->msg[1]; b.dst = ext; b.sbase = (long)r.offset; b.s0 = m0; b.s1 = m1; b.dbase = e; b.d0 = ne.p[0]; b.d1 = ne.p[1]; }
            b.n1 = r.box.n[1];
            n2s[k] = r.box.n[2];
        }
        const int vec = common_vec(jobs, n2s, (int)(16 / es));
        long total = 0;
        for (int k = 0; k < nreg; k++) {
            const block2::Region &r = is_pack ? p.send[k] : p.recv[k];
            jobs.j[k].nv = n2s[k] / vec;
            jobs.j[k].start = total;
            total += r.box.n[0] * r.box.n[1] * jobs.j[k].nv;
        }
        jobs.total = total;
        return launch_copy(jobs, vec, st);
    }
    // own cells between the state array (full array of g_box) and `ext`
    int copy_own(const NGrid &nl, void *state, void *ext, bool to_ext, void *st)
    {
        BoxJobs jobs;
        jobs.n = 1;
        BoxJob &b = jobs.j[0];
        const long lo[3] = {0, 0, 0};
        if (to_ext) { b.src = state; b.dst = ext; b.sbase = nl.off; b.s0 = nl.p[0]; b.s1 = nl.p[1]; b.dbase = elem_of(lo); b.d0 = ne.p[0]; b.d1 = ne.p[1]; }
        else { b.src = ext; b.dst = state; b.sbase = elem_of(lo); b.s0 = ne.p[0]; b.s1 = ne.p[1]; b.dbase = nl.off; b.d0 = nl.p[0]; b.d1 = nl.p[1]; }
        b.n1 = nl.n[1];
        const long n2s[1] = {nl.n[2]};
        const int vec = common_vec(jobs, n2s, (int)(16 / es));
        b.nv = nl.n[2] / vec;
        b.start = 0;
        jobs.total = nl.n[0] * nl.n[1] * b.nv;
        return launch_copy(jobs, vec, st);
    }
    int sweep2(const block2::Plan &p, void *cur, void *nxt, bool interior, void *st)
    {
        bool done = false;
        long lo[3] = {0, 0, 0}, n[3] = {p.n[0], p.n[1], p.n[2]};
        if (interior)
            for (int a = 0; a < 3; a++)
                if (p.cut[a]) { lo[a] = 2; n[a] -= 4; }
        PDEHIP_TRY(euler2_box(g_box, faces, p.cut, cur, nxt, D, dt, st, &done, false, lo, n));
        if (!done) PDEHIP_FAIL(E_RUNTIME, "internal: the two-step kernel refused a box it accepted in the dry run");
        return 0;
    }
    int rim2(const block2::Plan &p, void *cur, void *nxt, void *st)
    {
        if (!p.nrim) return 0;
        Rim2Args a;
        memset(&a, 0, sizeof(a));
        a.in = cur; a.out = nxt;
        a.n0 = p.n[0]; a.n1 = p.n[1]; a.n2 = p.n[2];
        a.p0 = ne.p[0]; a.p1 = ne.p[1]; a.off = own_off;
        for (int k = 0; k < 3; k++) a.wrap[k] = p.cut[k] ? 0 : 1;
        a.sx = ne.lap_scale[0]; a.sy = ne.lap_scale[1]; a.sz = ne.lap_scale[2];
        a.s1 = D; a.s2 = dt;
        const long vec = (long)(16 / es);
        a.nzseg = (p.n[2] / vec + 59) / 60;
        long total = 0;
        for (int k = 0; k < p.nrim; k++) {
            const block2::Box &b = p.rim[k];
            if (b.n[2] != p.n[2]) continue;   // the rims of the fastest axis: rimz_kernel below
            a.job[a.njobs] = {b.lo[0], b.lo[1], b.n[0], b.n[1], total};
            a.njobs++;
            total += ((b.n[0] + 1) / 2) * ((b.n[1] + 1) / 2) * a.nzseg;
        }
        a.total = total;
        if (direct) {
            a.recv = x->msg[1]; a.send = x->msg[0];
            for (int dx = -1; dx <= 1; dx++)
                for (int dy = -1; dy <= 1; dy++) {
                    const int d[3] = {dx, dy, 0};
                    const int c9 = (dx + 1) * 3 + (dy + 1), ri = p.recv_dir[block2::dir_code(d)], si = p.send_dir[block2::dir_code(d)];
                    a.rreg[c9] = {-1, 0, 0, 0};
                    a.sreg[c9] = {-1, 0, 0, 0};
                    if (ri >= 0) a.rreg[c9] = {(long)p.recv[ri].offset, p.recv[ri].box.lo[0], p.recv[ri].box.lo[1], p.recv[ri].box.n[1]};
                    if (si >= 0) a.sreg[c9] = {(long)p.send[si].offset, p.send[si].box.lo[0], p.send[si].box.lo[1], p.send[si].box.n[1]};
                }
        }
        if (total) {
            const unsigned blocks = (unsigned)((total + 3) / 4);
            if (direct) {
                if (es == 8) hipLaunchKernelGGL((rim2_kernel<double, 2, true>), dim3(blocks), dim3(256), 0, as_stream(st), a);
                else hipLaunchKernelGGL((rim2_kernel<float, 4, true>), dim3(blocks), dim3(256), 0, as_stream(st), a);
                send_ready = true;
            } else {
                if (es == 8) hipLaunchKernelGGL((rim2_kernel<double, 2>), dim3(blocks), dim3(256), 0, as_stream(st), a);
                else hipLaunchKernelGGL((rim2_kernel<float, 4>), dim3(blocks), dim3(256), 0, as_stream(st), a);
            }
            PDEHIP_HIP(hipGetLastError());
        }
        if (p.cut[2]) {
            RimzArgs z;
            memset(&z, 0, sizeof(z));
            z.in = cur; z.out = nxt;
            z.n0 = p.n[0]; z.n1 = p.n[1]; z.n2 = p.n[2];
            z.p0 = ne.p[0]; z.p1 = ne.p[1]; z.off = own_off;
            for (int k = 0; k < 3; k++) z.wrap[k] = p.cut[k] ? 0 : 1;
            z.sx = a.sx; z.sy = a.sy; z.sz = a.sz; z.s1 = D; z.s2 = dt;
            z.nrseg = (p.n[1] + 59) / 60;
            z.total = 2 * ((p.n[0] + 1) / 2) * z.nrseg;
            const unsigned blocks = (unsigned)((z.total + 3) / 4);
            if (es == 8) hipLaunchKernelGGL((rimz_kernel<double>), dim3(blocks), dim3(256), 0, as_stream(st), z);
            else hipLaunchKernelGGL((rimz_kernel<float>), dim3(blocks), dim3(256), 0, as_stream(st), z);
            PDEHIP_HIP(hipGetLastError());
        }
        return 0;
    }
};

// what the fast block loop covers: 3-D, diffusion, every face of an uncut axis periodic, boxes and rows the two-step kernel and the 16-byte
// vectors of the rim kernel take (a cut fastest axis: its two halo cells must fit the padding of the rows, euler2_box checks)
int block2_check(const pdehip_grid_t *g_local, const pdehip_rhs_t *rhs, const int *cut3, bool *ok)
{
    *ok = false;
    if (g_local->ndim != 3 || rhs->kind != PDEHIP_RHS_DIFFUSION || rhs->bc_program) return 0;
    const long vec = 16 / elem_size(g_local->dtype);
    if (g_local->shape[2] % vec || g_local->shape[0] < 4 || g_local->shape[1] < 4 || (cut3[2] && g_local->shape[2] < 8)) return 0;
    // (fp32 with a cut fastest axis: the interior box starts two cells into a four-cell vector - the narrow tile's 8-byte vectors take it since
    // the end of round 6, launch_euler2: narrow_only)
    bool done = false;
    PDEHIP_TRY(euler2_box(g_local, rhs->bc_c, cut3, (const void *)16, (void *)32, rhs->param, 0.0, nullptr, &done, true));
    if (done) {   // ... and the interior box of the boundary-first schedules
        long lo[3] = {0, 0, 0}, n[3] = {g_local->shape[0], g_local->shape[1], g_local->shape[2]};
        for (int a = 0; a < 3; a++)
            if (cut3[a]) { lo[a] = 2; n[a] -= 4; }
        if (n[0] < 1 || n[1] < 4) done = false;
        else PDEHIP_TRY(euler2_box(g_local, rhs->bc_c, cut3, (const void *)16, (void *)32, rhs->param, 0.0, nullptr, &done, true, lo, n));
    }
    *ok = done;
    return 0;
}

}  // namespace

namespace {
void block2_release(Block2Ctx *x)
{
    if (!x) return;
    (void)hipFree(x->ext[0]); (void)hipFree(x->ext[1]);
    (void)hipFree(x->msg[0]); (void)hipFree(x->msg[1]);
    for (auto &e : x->ev)
        if (e) (void)hipEventDestroy(e);
    if (x->comp_masked) (void)hipStreamDestroy(x->comp_masked);
    if (x->halo_masked) (void)hipStreamDestroy(x->halo_masked);
    delete x;
}
}  // namespace

extern "C" {

int pdehip_block2_supported(const pdehip_grid_t *g_local, const pdehip_rhs_t *rhs, const int *cut3, int *ok)
{
    if (!g_local || !rhs || !cut3 || !ok) PDEHIP_FAIL(E_VALUE, "block2_supported: NULL pointer");
    bool b = false;
    PDEHIP_TRY(block2_check(g_local, rhs, cut3, &b));
    *ok = b ? 1 : 0;
    return 0;
}

int pdehip_block2_euler_run(void *comm, const pdehip_grid_t *g_local, const pdehip_rhs_t *rhs, const int *dims3, const int *coords3,
                            const int *cut3, void *buf_a, void *buf_b, double dt, int64_t nsteps, void **result, void *stream)
{
    if (!g_local || !rhs || !dims3 || !coords3 || !cut3 || !buf_a || !buf_b || !result) PDEHIP_FAIL(E_VALUE, "block2_euler_run: NULL pointer");
    if (nsteps < 0 || nsteps % 2) PDEHIP_FAIL(E_VALUE, "block2_euler_run: the step count must be even (two steps per sweep)");
    bool ok = false;
    PDEHIP_TRY(block2_check(g_local, rhs, cut3, &ok));
    if (!ok) PDEHIP_FAIL(E_NOTIMPL, "block2_euler_run: grid, equation or faces are not covered (ask pdehip_block2_supported)");
    const bool any = cut3[0] || cut3[1] || cut3[2];
    Comm *c = static_cast<Comm *>(comm);
    if (!c) {
        if (any) PDEHIP_FAIL(E_VALUE, "a block with neighbours needs a communicator");
        c = serial_context();
    }
    PDEHIP_TRY(ensure_streams(c));
    for (int a = 0; a < 3; a++) {
        if (dims3[a] < 1 || coords3[a] < 0 || coords3[a] >= dims3[a]) PDEHIP_FAIL(E_VALUE, "block2_euler_run: bad decomposition");
        if (dims3[a] > 1 && !cut3[a]) PDEHIP_FAIL(E_VALUE, "block2_euler_run: an axis with several blocks must be exchanged");
    }
    if (any && (long)dims3[0] * dims3[1] * dims3[2] != c->size) PDEHIP_FAIL(E_VALUE, "block2_euler_run: the decomposition does not match the world size");
    block2::Plan plan;
    const long n[3] = {g_local->shape[0], g_local->shape[1], g_local->shape[2]};
    if (block2::make_plan(n, dims3, coords3, cut3, &plan) != 0) PDEHIP_FAIL(E_VALUE, "block2_euler_run: the box is too small for two halo layers");
    Block2Ctx *x = block2_ctx(c);
    if (any) PDEHIP_TRY(separate_queues(c, as_stream(stream)));
    ScratchTurn turn(c, as_stream(stream));
    for (auto &e : x->ev)
        if (!e) PDEHIP_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    HipOps2 ops;
    ops.c = c; ops.x = x; ops.g_box = g_local; ops.faces = rhs->bc_c; ops.D = rhs->param; ops.dt = dt;
    pdehip_grid_t ge = *g_local;
    ge.shape[0] += 2; ge.shape[1] += 2;
    PDEHIP_TRY(norm_grid(&ge, &ops.ne));
    ops.own_off = ops.ne.off + ops.ne.p[0] + ops.ne.p[1];
    ops.es = (size_t)elem_size(ops.ne.dtype);
    NGrid nl;
    PDEHIP_TRY(norm_grid(g_local, &nl));
    hipStream_t comp = as_stream(stream);
    // schedule and partition of the compute units (tuning aids; the defaults are what measured best on one MI355X, profiles/r05_probe_block.md)
    static const int mode = getenv("PDEHIP_BLOCK2_MODE") ? atoi(getenv("PDEHIP_BLOCK2_MODE")) : 2;
    static const int want_cus = getenv("PDEHIP_BLOCK2_CUS") ? atoi(getenv("PDEHIP_BLOCK2_CUS")) : 0;
    if (mode < 0 || mode > 2) PDEHIP_FAIL(E_VALUE, "PDEHIP_BLOCK2_MODE: 0, 1 or 2");
    ops.halo_st = c->halo;
    hipStream_t comp2 = comp;
    if (want_cus > 0 && any) {
        if (x->reserved != want_cus) {
            int ncu = 0;
            PDEHIP_HIP(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0));
            if (want_cus >= ncu || ncu > 1024) PDEHIP_FAIL(E_VALUE, "PDEHIP_BLOCK2_CUS: between 1 and the number of compute units - 1");
            uint32_t mc[32], mh[32];
            const int words = (ncu + 31) / 32;
            for (int w = 0; w < words; w++) { mc[w] = 0; mh[w] = 0; }
            for (int b = 0; b < ncu; b++) (b < want_cus ? mh : mc)[b / 32] |= 1u << (b % 32);
            if (x->comp_masked) { (void)hipStreamDestroy(x->comp_masked); (void)hipStreamDestroy(x->halo_masked); }
            PDEHIP_HIP(hipExtStreamCreateWithCUMask(&x->comp_masked, (uint32_t)words, mc));
            PDEHIP_HIP(hipExtStreamCreateWithCUMask(&x->halo_masked, (uint32_t)words, mh));
            x->reserved = want_cus;
        }
        comp2 = x->comp_masked;
        ops.halo_st = x->halo_masked;
    }
    static const bool direct_off = getenv("PDEHIP_BLOCK2_DIRECT") && getenv("PDEHIP_BLOCK2_DIRECT")[0] == '0';   // A/B aid
    ops.direct = mode >= 1 && !cut3[2] && any && !direct_off && g_local->shape[2] % (16 / elem_size(g_local->dtype)) == 0;
    const size_t need = (size_t)(ops.ne.pc + kAllocSlack) * ops.es;
    if (x->ext_bytes < need) {
        if (c->ev_done && c->ev_done_recorded) PDEHIP_HIP(hipEventSynchronize(c->ev_done));
        PDEHIP_HIP(hipStreamSynchronize(c->halo));
        PDEHIP_HIP(hipStreamSynchronize(comp));
        (void)hipFree(x->ext[0]); (void)hipFree(x->ext[1]);
        x->ext[0] = x->ext[1] = nullptr; x->ext_bytes = 0;
        PDEHIP_HIP(hipMalloc(&x->ext[0], need));
        PDEHIP_HIP(hipMalloc(&x->ext[1], need));
        x->ext_bytes = need;
        PDEHIP_HIP(hipMemsetAsync(x->ext[0], 0, need, comp));
        PDEHIP_HIP(hipMemsetAsync(x->ext[1], 0, need, comp));
    }
    const size_t mneed = (plan.send_total > plan.recv_total ? plan.send_total : plan.recv_total) * ops.es + 256;
    if (x->msg_bytes < mneed) {
        if (c->ev_done && c->ev_done_recorded) PDEHIP_HIP(hipEventSynchronize(c->ev_done));
        PDEHIP_HIP(hipStreamSynchronize(c->halo));
        (void)hipFree(x->msg[0]); (void)hipFree(x->msg[1]);
        x->msg[0] = x->msg[1] = nullptr; x->msg_bytes = 0;
        PDEHIP_HIP(hipMalloc(&x->msg[0], mneed));
        PDEHIP_HIP(hipMalloc(&x->msg[1], mneed));
        x->msg_bytes = mneed;
    }
    PDEHIP_TRY(ops.copy_own(nl, buf_a, x->ext[0], true, comp));
    void *res = x->ext[0];
    if (comp2 != comp) { PDEHIP_HIP(hipEventRecord(x->ev[3], comp)); PDEHIP_HIP(hipStreamWaitEvent(comp2, x->ev[3], 0)); }
    if (nsteps > 0) PDEHIP_TRY(block2::euler2_run(ops, plan, x->ext[0], x->ext[1], nsteps, &res, comp2, mode));
    if (comp2 != comp) { PDEHIP_HIP(hipEventRecord(x->ev[3], comp2)); PDEHIP_HIP(hipStreamWaitEvent(comp, x->ev[3], 0)); }
    PDEHIP_TRY(ops.copy_own(nl, buf_a, res, false, comp));
    *result = buf_a;
    return 0;
}

}  // extern "C"
