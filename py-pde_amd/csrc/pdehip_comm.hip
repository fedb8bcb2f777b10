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

#include "pdehip_common.h"

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
    g_rccl.handle = h;
    return 0;
}

#define PDEHIP_NCCL(expr)                                                                                  \
    do {                                                                                                   \
        ncclResult_t _r = (expr);                                                                          \
        if (_r != ncclSuccess) PDEHIP_FAIL(E_RUNTIME, "%s failed: %s", #expr, g_rccl.GetErrorString(_r)); \
    } while (0)

struct Comm {
    ncclComm_t comm;
    int rank, size;
    hipStream_t halo;            // stream of the exchange + boundary-layer kernels
    hipEvent_t ev_comp, ev_halo, ev_bnd;
    double *scratch2;            // device: {value, nan flag} for the MAX all-reduce
    void *ext[2] = {nullptr, nullptr};   // slab copies with TWO halo layers per side (two-steps-per-sweep loop)
    size_t ext_bytes = 0;
};

// pointer to full layer `layer` (0 = lower ghost layer) of a slab
inline char *layer_ptr(const NGrid &n, void *buf, long layer)
{
    return static_cast<char *>(buf) + layer * n.p[3 - n.ndim] * elem_size(n.dtype);
}

// Post the halo exchange of `buf` on `st`.  Order per peer: the "downward" pair first, then the
// "upward" pair — RCCL matches sends and receives to one peer in issue order, so the 2-rank periodic
// ring and the 1-rank self exchange pair up correctly (same order as pde_hip/distributed.py, which is
// tested on CPU with gloo at world sizes 2 and 3).
int exchange(Comm *c, const NGrid &n, void *buf, int lower, int upper, hipStream_t st)
{
    if (lower < 0 && upper < 0) return 0;
    const long nloc = n.n[3 - n.ndim];
    const size_t bytes = (size_t)n.p[3 - n.ndim] * elem_size(n.dtype);
    PDEHIP_NCCL(g_rccl.GroupStart());
    if (lower >= 0) PDEHIP_NCCL(g_rccl.Send(layer_ptr(n, buf, 1), bytes, ncclInt8, lower, c->comm, st));
    if (upper >= 0) {
        PDEHIP_NCCL(g_rccl.Recv(layer_ptr(n, buf, nloc + 1), bytes, ncclInt8, upper, c->comm, st));
        PDEHIP_NCCL(g_rccl.Send(layer_ptr(n, buf, nloc), bytes, ncclInt8, upper, c->comm, st));
    }
    if (lower >= 0) PDEHIP_NCCL(g_rccl.Recv(layer_ptr(n, buf, 0), bytes, ncclInt8, lower, c->comm, st));
    PDEHIP_NCCL(g_rccl.GroupEnd());
    return 0;
}

// faces of a sub-slab of layers [first, first+count) (1-based valid layers of the slab): the
// inter-layer faces inside the slab are real data (SKIP); physical / exchanged faces keep the slab's
// descriptor with the index translated into the sub-slab
void sub_faces(const pdehip_bc_face_t *faces, long nloc, long first, long count, pdehip_bc_face_t *out)
{
    for (int i = 0; i < 2 * PDEHIP_MAX_DIM; i++) out[i] = faces[i];
    if (first > 1) out[0].kind = PDEHIP_BC_SKIP;
    else out[0].index1 -= (first - 1), out[0].index2 -= (first - 1);
    if (first + count - 1 < nloc) out[1].kind = PDEHIP_BC_SKIP;
    else out[1].index1 -= (first - 1), out[1].index2 -= (first - 1);
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
    PDEHIP_HIP(hipStreamCreateWithFlags(&c->halo, hipStreamNonBlocking));
    PDEHIP_HIP(hipEventCreateWithFlags(&c->ev_comp, hipEventDisableTiming));
    PDEHIP_HIP(hipEventCreateWithFlags(&c->ev_halo, hipEventDisableTiming));
    PDEHIP_HIP(hipEventCreateWithFlags(&c->ev_bnd, hipEventDisableTiming));
    PDEHIP_HIP(hipMalloc(&c->scratch2, 2 * sizeof(double)));
    *comm = c;
    return 0;
}

int pdehip_comm_destroy(void *comm)
{
    if (!comm) return 0;
    Comm *c = static_cast<Comm *>(comm);
    (void)hipStreamSynchronize(c->halo);
    g_rccl.CommDestroy(c->comm);
    (void)hipStreamDestroy(c->halo);
    (void)hipEventDestroy(c->ev_comp); (void)hipEventDestroy(c->ev_halo); (void)hipEventDestroy(c->ev_bnd);
    (void)hipFree(c->scratch2);
    (void)hipFree(c->ext[0]); (void)hipFree(c->ext[1]);
    delete c;
    return 0;
}

int pdehip_halo_exchange(void *comm, const pdehip_grid_t *g_local, void *buf_full, int lower, int upper, void *stream)
{
    if (!comm || !buf_full) PDEHIP_FAIL(E_VALUE, "halo_exchange: NULL pointer");
    NGrid n;
    PDEHIP_TRY(norm_grid(g_local, &n));
    return exchange(static_cast<Comm *>(comm), n, buf_full, lower, upper, as_stream(stream));
}

int pdehip_allreduce_max(void *comm, double *dev_scalar, void *stream)
{
    if (!comm || !dev_scalar) PDEHIP_FAIL(E_VALUE, "allreduce_max: NULL pointer");
    Comm *c = static_cast<Comm *>(comm);
    if (c->size == 1) return 0;
    hipStream_t st = as_stream(stream);
    // NaN must win like numpy's max: reduce {value with NaN -> 0, NaN flag}
    hipLaunchKernelGGL(pack_nan_kernel, dim3(1), dim3(1), 0, st, dev_scalar, c->scratch2);
    PDEHIP_NCCL(g_rccl.AllReduce(c->scratch2, c->scratch2, 2, ncclFloat64, ncclMax, c->comm, st));
    hipLaunchKernelGGL(unpack_nan_kernel, dim3(1), dim3(1), 0, st, c->scratch2, dev_scalar);
    PDEHIP_HIP(hipGetLastError());
    return 0;
}

// nsteps explicit Euler steps of the diffusion equation on one slab; all work is enqueued without
// host synchronisation:
//   comp stream : interior kernel (layers 2..n-1)   ............................ | next step
//   halo stream : boundary kernels (layers 1, n) - send/recv of the new layers 1, n
// The exchange of step s+1's input overlaps the interior kernel of step s.  BCs of the faces this
// rank owns are evaluated on the fly inside the kernels; exchanged faces read the received layers.
int pdehip_slab_euler_run(void *comm, const pdehip_grid_t *g_local, const pdehip_rhs_t *rhs, int lower, int upper,
                          void *buf_a, void *buf_b, double dt, int64_t nsteps, void **result, void *stream)
{
    if (!comm || !rhs || !buf_a || !buf_b || !result) PDEHIP_FAIL(E_VALUE, "slab_euler_run: NULL pointer");
    if (rhs->kind != PDEHIP_RHS_DIFFUSION) PDEHIP_FAIL(E_NOTIMPL, "slab_euler_run implements the diffusion right-hand side");
    Comm *c = static_cast<Comm *>(comm);
    NGrid n;
    PDEHIP_TRY(norm_grid(g_local, &n));
    const long nloc = g_local->shape[0];
    hipStream_t comp = as_stream(stream), halo = c->halo;
    // faces: exchanged ones are read from the ghost layers
    pdehip_bc_face_t faces[2 * PDEHIP_MAX_DIM];
    for (int i = 0; i < 2 * PDEHIP_MAX_DIM; i++) faces[i] = rhs->bc_c[i];
    if (lower >= 0) faces[0].kind = PDEHIP_BC_SKIP;
    if (upper >= 0) faces[1].kind = PDEHIP_BC_SKIP;
    const size_t esz = elem_size(n.dtype);
    const size_t lp = (size_t)n.p[3 - n.ndim] * esz;   // bytes per layer

    auto sub_step = [&](hipStream_t st, void *cur, void *nxt, long first, long count) -> int {
        if (count <= 0) return 0;
        pdehip_grid_t gs = *g_local;
        gs.shape[0] = count;
        pdehip_bc_face_t sf[2 * PDEHIP_MAX_DIM];
        sub_faces(faces, nloc, first, count, sf);
        char *pc = static_cast<char *>(cur) + (first - 1) * lp, *pn = static_cast<char *>(nxt) + (first - 1) * lp;
        return laplace_with_input_bcs(&gs, pc, pc, pn, LAP_EULER, rhs->param, dt, 0, sf, st);
    };

    void *cur = buf_a, *nxt = buf_b;
    // ghost layers of the initial state
    PDEHIP_HIP(hipEventRecord(c->ev_comp, comp));
    PDEHIP_HIP(hipStreamWaitEvent(halo, c->ev_comp, 0));
    PDEHIP_TRY(exchange(c, n, cur, lower, upper, halo));
    for (int64_t s = 0; s < nsteps; s++) {
        // interior layers need no exchanged data; they must wait for the boundary layers of `cur`
        // (written on the halo stream in the previous step)
        if (s > 0) PDEHIP_HIP(hipStreamWaitEvent(comp, c->ev_bnd, 0));
        PDEHIP_TRY(sub_step(comp, cur, nxt, 2, nloc - 2));
        PDEHIP_HIP(hipEventRecord(c->ev_comp, comp));
        // boundary layers: the received ghost layers are ordered by the halo stream itself
        PDEHIP_TRY(sub_step(halo, cur, nxt, 1, 1));
        if (nloc > 1) PDEHIP_TRY(sub_step(halo, cur, nxt, nloc, 1));
        PDEHIP_HIP(hipEventRecord(c->ev_bnd, halo));
        PDEHIP_TRY(exchange(c, n, nxt, lower, upper, halo));   // overlaps the interior kernel
        // the next step overwrites `cur`: its interior kernel (comp) and boundary kernels (halo, in
        // order) must be done; the halo stream additionally waits for this step's interior kernel
        PDEHIP_HIP(hipStreamWaitEvent(halo, c->ev_comp, 0));
        void *t = cur; cur = nxt; nxt = t;
    }
    // make the compute stream see everything
    PDEHIP_HIP(hipEventRecord(c->ev_halo, halo));
    PDEHIP_HIP(hipStreamWaitEvent(comp, c->ev_halo, 0));
    *result = cur;
    return 0;
}

// ---------------------------------------------------------------------------------------------------------
// Two Euler steps per sweep on a slab (temporal blocking, pdehip_march2.inc) — halves both the HBM traffic per
// step and the NUMBER of halo exchanges: two layers per side are exchanged once per two steps.
//
// The slab is copied into a private array with two halo layers per side (layers 0,1 | own 2..n+1 | n+2,n+3):
//   comp stream : interior sweep (own layers 4..n-1; reads own layers only)        ............ | next pair
//   halo stream : boundary sweeps (layers 2,3 and n,n+1; read the received halos) - send/recv of the new
//                 boundary layers, overlapping the interior sweep
// Requires: both neighbours present on EVERY rank (periodic slowest axis), >= 4 local layers on every rank and a
// grid / faces the kernel covers (pdehip_slab_euler2_supported) — the caller decides globally, all ranks alike.
// ---------------------------------------------------------------------------------------------------------
static int exchange2(Comm *c, size_t lp, long nloc, void *ext, int lower, int upper, hipStream_t st)
{
    char *b = static_cast<char *>(ext);
    if (lower < 0 && upper < 0) return 0;
    PDEHIP_NCCL(g_rccl.GroupStart());
    if (lower >= 0) PDEHIP_NCCL(g_rccl.Send(b + 2 * lp, 2 * lp, ncclInt8, lower, c->comm, st));            // own first two layers -> lower
    if (upper >= 0) PDEHIP_NCCL(g_rccl.Recv(b + (nloc + 2) * lp, 2 * lp, ncclInt8, upper, c->comm, st));   // upper halo <- upper
    if (upper >= 0) PDEHIP_NCCL(g_rccl.Send(b + nloc * lp, 2 * lp, ncclInt8, upper, c->comm, st));         // own last two layers -> upper
    if (lower >= 0) PDEHIP_NCCL(g_rccl.Recv(b, 2 * lp, ncclInt8, lower, c->comm, st));                     // lower halo <- lower
    PDEHIP_NCCL(g_rccl.GroupEnd());
    return 0;
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

int pdehip_slab_euler2_run(void *comm, const pdehip_grid_t *g_local, const pdehip_rhs_t *rhs, int lower, int upper,
                           void *buf_a, void *buf_b, double dt, int64_t nsteps, void **result, void *stream)
{
    if (!comm || !rhs || !buf_a || !buf_b || !result) PDEHIP_FAIL(E_VALUE, "slab_euler2_run: NULL pointer");
    int ok = 0;
    PDEHIP_TRY(pdehip_slab_euler2_supported(g_local, rhs, &ok));
    if (!ok) PDEHIP_FAIL(E_NOTIMPL, "slab_euler2_run: grid or faces are not covered by the two-step kernel");
    // sides of the slowest axis without a neighbour keep their (local, first-order) physical face:
    // 0 both physical, 1 both exchanged, 2 lower physical, 3 upper physical  (xplain codes of launch_euler2)
    const int xends = (lower >= 0 && upper >= 0) ? 1 : (lower < 0 && upper < 0) ? 0 : (lower < 0 ? 2 : 3);
    Comm *c = static_cast<Comm *>(comm);
    NGrid n;
    PDEHIP_TRY(norm_grid(g_local, &n));
    const long nloc = g_local->shape[0];
    hipStream_t comp = as_stream(stream), halo = c->halo;
    const size_t esz = elem_size(n.dtype);
    const size_t lp = (size_t)n.p[0] * esz;   // bytes per layer
    // private arrays with two halo layers per side: the layout of a slab of nloc+2 layers
    pdehip_grid_t ge = *g_local;
    ge.shape[0] = nloc + 2;
    NGrid ne;
    PDEHIP_TRY(norm_grid(&ge, &ne));
    const size_t need = (size_t)(ne.pc + kAllocSlack) * esz;
    if (c->ext_bytes < need) {
        PDEHIP_HIP(hipStreamSynchronize(halo));
        (void)hipFree(c->ext[0]); (void)hipFree(c->ext[1]);
        c->ext[0] = c->ext[1] = nullptr; c->ext_bytes = 0;
        PDEHIP_HIP(hipMalloc(&c->ext[0], need));
        PDEHIP_HIP(hipMalloc(&c->ext[1], need));
        c->ext_bytes = need;
        PDEHIP_HIP(hipMemsetAsync(c->ext[0], 0, need, comp));
        PDEHIP_HIP(hipMemsetAsync(c->ext[1], 0, need, comp));
    }
    char *cur = static_cast<char *>(c->ext[0]), *nxt = static_cast<char *>(c->ext[1]);
    // own layers: slab layers 1..nloc -> private layers 2..nloc+1 (same row layout, one layer further in)
    PDEHIP_HIP(hipMemcpyAsync(cur + 2 * lp, static_cast<char *>(buf_a) + lp, (size_t)nloc * lp, hipMemcpyDeviceToDevice, comp));

    pdehip_bc_face_t faces[2 * PDEHIP_MAX_DIM];
    for (int i = 0; i < 2 * PDEHIP_MAX_DIM; i++) faces[i] = rhs->bc_c[i];
    if (lower >= 0) faces[0].kind = PDEHIP_BC_SKIP;   // exchanged sides: real layers
    if (upper >= 0) faces[1].kind = PDEHIP_BC_SKIP;

    // two steps on private layers [first, first+count)
    // (ends > 0: the first and the last `ends` layers of the range in one launch)
    auto sweep2 = [&](hipStream_t st, long first, long count, int ends) -> int {
        if (count <= 0) return 0;
        pdehip_grid_t gs = *g_local;
        gs.shape[0] = count;
        bool done = false;
        // the interior sweep reads own layers only (plain on both sides); the two-ended boundary sweep meets the physical faces
        PDEHIP_TRY(euler2_with_input_bcs(&gs, cur + (first - 1) * lp, nxt + (first - 1) * lp, rhs->param, dt, faces, st, &done,
                                         ends ? xends : 1, false, ends));
        if (!done) PDEHIP_FAIL(E_RUNTIME, "internal: two-step kernel refused a sub-slab");
        return 0;
    };

    // comp stream : interior sweep (reads own layers only)                          | interior sweep ...
    // halo stream : [wait interior s-1] boundary sweep - send/recv new boundary layers | [wait] boundary sweep ...
    // Measured alternatives (profiles/r01_probe_slab_euler2.md): boundary sweep first on the compute stream, then the
    // interior sweep — the RCCL kernel then crawls behind the full-occupancy interior sweep and ends with it (worse at
    // every slab thickness); capping the interior sweep at 75 % of the wave slots helps only thin slabs.
    PDEHIP_HIP(hipEventRecord(c->ev_comp, comp));
    PDEHIP_HIP(hipStreamWaitEvent(halo, c->ev_comp, 0));
    PDEHIP_TRY(exchange2(c, lp, nloc, cur, lower, upper, halo));
    int64_t s = 0;
    bool first_pair = true;
    for (; s + 2 <= nsteps; s += 2) {
        if (!first_pair) PDEHIP_HIP(hipStreamWaitEvent(comp, c->ev_bnd, 0));   // boundary layers of `cur` (halo stream)
        first_pair = false;
        PDEHIP_TRY(sweep2(comp, 4, nloc - 4, 0));
        PDEHIP_HIP(hipEventRecord(c->ev_comp, comp));
        PDEHIP_TRY(sweep2(halo, 2, nloc, 2));   // own layers 2,3 and nloc,nloc+1 (needs the received halo layers)
        PDEHIP_HIP(hipEventRecord(c->ev_bnd, halo));
        if (s + 2 < nsteps) PDEHIP_TRY(exchange2(c, lp, nloc, nxt, lower, upper, halo));   // overlaps the interior sweep
        // the next pair overwrites `cur` and its boundary sweep reads the interior layers written now
        PDEHIP_HIP(hipStreamWaitEvent(halo, c->ev_comp, 0));
        char *t = cur; cur = nxt; nxt = t;
    }
    PDEHIP_HIP(hipEventRecord(c->ev_halo, halo));
    PDEHIP_HIP(hipStreamWaitEvent(comp, c->ev_halo, 0));
    if (s < nsteps) {
        // odd step count: one single step; layers 1 and nloc+2 act as its ghost layers (already exchanged)
        pdehip_grid_t gs = *g_local;
        PDEHIP_TRY(laplace_with_input_bcs(&gs, cur + lp, cur + lp, nxt + lp, LAP_EULER, rhs->param, dt, 0, faces, comp));
        char *t = cur; cur = nxt; nxt = t;
    }
    PDEHIP_HIP(hipMemcpyAsync(static_cast<char *>(buf_a) + lp, cur + 2 * lp, (size_t)nloc * lp, hipMemcpyDeviceToDevice, comp));
    *result = buf_a;
    return 0;
}

// ---------------------------------------------------------------------------------------------------------
// Cahn-Hilliard right-hand side on a slab in ONE sweep (fused two-level kernel, mu in registers): two layers of c per
// side are exchanged, mu needs no exchange of its own (the reference exchanges c AND mu, one layer each:
// pde/grids/boundaries/local.py:561-662 per operator application).  `c_ext` / `out_ext` are slab arrays with TWO halo
// layers per side (layers 0,1 | own 2..n+1 | n+2,n+3), i.e. the layout of a slab of n+2 layers.
//   euler != 0: out = c + dt * laplace(mu)        euler == 0: out = dt * laplace(mu)
// Preconditions as for pdehip_slab_euler2_run (periodic slowest axis, >= 2 own layers on every rank, *ok from
// pdehip_slab_ch_supported), checked globally by the caller.
// ---------------------------------------------------------------------------------------------------------
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

int pdehip_slab_ch_sweep(void *comm, const pdehip_grid_t *g_local, const pdehip_rhs_t *rhs, int lower, int upper,
                         void *c_ext, void *out_ext, double dt, int euler, void *stream)
{
    if (!comm || !rhs || !c_ext || !out_ext) PDEHIP_FAIL(E_VALUE, "slab_ch_sweep: NULL pointer");
    // sides without a neighbour keep their physical faces (xplain codes of launch_euler2)
    const int xmode = (lower >= 0 && upper >= 0) ? 1 : (lower < 0 && upper < 0) ? 0 : (lower < 0 ? 2 : 3);
    Comm *c = static_cast<Comm *>(comm);
    NGrid n;
    PDEHIP_TRY(norm_grid(g_local, &n));
    const long nloc = g_local->shape[0];
    const size_t lp = (size_t)n.p[0] * elem_size(n.dtype);
    hipStream_t st = as_stream(stream);
    PDEHIP_TRY(exchange2(c, lp, nloc, c_ext, lower, upper, st));
    pdehip_bc_face_t fc[2 * PDEHIP_MAX_DIM], fm[2 * PDEHIP_MAX_DIM];
    for (int i = 0; i < 2 * PDEHIP_MAX_DIM; i++) { fc[i] = rhs->bc_c[i]; fm[i] = rhs->bc_mu[i]; }
    if (lower >= 0) fc[0].kind = fm[0].kind = PDEHIP_BC_SKIP;
    if (upper >= 0) fc[1].kind = fm[1].kind = PDEHIP_BC_SKIP;
    bool done = false;
    PDEHIP_TRY(cahn_hilliard_fused(g_local, static_cast<char *>(c_ext) + lp, static_cast<char *>(out_ext) + lp, rhs->param, dt,
                                   euler != 0, fc, fm, stream, &done, xmode));
    if (!done) PDEHIP_FAIL(E_NOTIMPL, "slab_ch_sweep: grid or faces are not covered by the two-level kernel");
    return 0;
}

}  // extern "C"
