// pdehip_slab_loops.h — the slab-parallel time loops, written ONCE against an `Ops` policy.
//
// One process per GPU owns an axis-0 slab of the grid (pde_hip/mesh.py; partition rule of pde/grids/_mesh.py:96-111).
// This header holds every piece of control flow that decides WHO sends WHAT to WHOM and in WHICH ORDER — the halo
// exchanges, the stream/event choreography that overlaps them with the interior sweeps, the Runge-Kutta stage sequence
// and the adaptive accept/reject loop.  It replaces the reference's MPI path: blocking face exchange inside every
// right-hand side (pde/backends/numba_mpi/backend.py:30-194, pde/grids/boundaries/local.py:561-662), the MAX all-reduce of
// the adaptive error (pde/backends/base.py:678-712; loop pde/backends/numba/_solvers.py:249-281) and the stepper wrapper
// pde/solvers/explicit_mpi.py:133-226.
//
// The code is a template over `Ops` (streams, events, point-to-point transport, kernels) and has two instantiations:
//   * csrc/pdehip_comm.hip      HipOps: HIP streams/events, RCCL ncclSend/ncclRecv groups over xGMI, the gfx950 kernels — the product;
//   * tests/shim/pdehip_shim_comm.cpp  HostOps: host memory, a file-mailbox transport between the ranks of a CPU job and
//                               the CPU oracle as kernels — TESTS ONLY, so that world sizes 2..4 execute exactly this
//                               call sequence on a box without GPUs (tests/test_distributed_gloo.py).
// Plain C++17, no HIP types: a stream is an opaque `void *`.
//
// Transport contract (what RCCL gives and the mailbox transport reproduces): sends and receives issued between
// group_start() and group_end() progress together (no ordering deadlock inside a group); messages between one pair of
// ranks match in issue order; a rank may be its own peer (world size 1 with a periodic axis).
#pragma once

#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstring>

#include "../../include/pdehip.h"

namespace pdehip {

// what follows a slope k = dt*rhs in a Runge-Kutta scheme, fused into the sweep that computes k (mode LAP_STAGE)
struct StageFuse {
    int kind;            // 0: next stage input  out2 = y + sum_m c[m]*k[m] + c_new*k  (k is also stored);  1: RK4 update
                         //    out2 = y + (k[0] + 2*k[1] + 2*k[2] + k)/6  (k is not stored; out2 may be y itself)
                         // 2: end of an RKF45 attempt  out2 = 4th-order state from y and k = {k1, k3, k4, k5}, *err = max-norm
                         //    of the error estimate with k6 = k  (k is not stored; *err must be zero before the launch)
                         // 3: Adams-Bashforth step  out2 = y + c_new * (1.5*k - 0.5*k[0])  with k = the rate (also stored), c_new = dt
                         // 4: end of an adaptive Euler attempt (pde/backends/numba/_solvers.py:380-395): k = dt/2 * rhs(step_half) with
                         //    step_half = k[1] (REQUIRED to be the input array of the sweep), out2 = step_half + k (the double step),
                         //    *err = max |(y + c[0]*k[0]) - out2| with k[0] = the rate carried from the last accepted state and
                         //    c[0] = dt (the single step; never stored)  (k is not stored; *err must be zero before the launch)
    const void *y;
    const void *k[5];    // earlier slopes, NULL-terminated
    double c[5], c_new;
    void *out2;
    double *err;
};

namespace slab {

enum { EV_COMP = 0, EV_HALO = 1, EV_BND = 2, EV_BND2 = 3 };   // EV_BND / EV_BND2: boundary sweeps of even / odd pairs (euler2_run)
// stencil flavours the loops ask the kernels for (HipOps maps them onto LAP_* of pdehip_device.h)
enum { K_SCALED = 1, K_EULER = 2, K_CH_MU = 3, K_STAGE = 10 };
// how the loops were told to run (decided GLOBALLY by the caller: every rank must take the same path)
enum {
    F_FUSED_CH = 1,     // Cahn-Hilliard right-hand side as ONE two-level sweep after ONE exchange of two layers (else two kernels, two exchanges)
    F_FUSED_STAGE = 2,  // Runge-Kutta stage epilogue inside the sweep (else slope kernel + pointwise combination)
};

// RKF45 tableau, pde/solvers/runge_kutta.py:92-112 (identical quotients)
inline const double *rkf45_row(int s)
{
    static const double B2[] = {1.0 / 4};
    static const double B3[] = {3.0 / 32, 9.0 / 32};
    static const double B4[] = {1932.0 / 2197, -7200.0 / 2197, 7296.0 / 2197};
    static const double B5[] = {439.0 / 216, -8.0, 3680.0 / 513, -845.0 / 4104};
    static const double B6[] = {-8.0 / 27, 2.0, -3544.0 / 2565, 1859.0 / 4104, -11.0 / 40};
    static const double *const rows[5] = {B2, B3, B4, B5, B6};
    return rows[s];
}

#define SLAB_TRY(expr)            \
    do {                          \
        int _rc = (expr);         \
        if (_rc != 0) return _rc; \
    } while (0)

// geometry of the local slab in bytes / layers
struct Geo {
    long nloc;      // own layers
    size_t lp;      // bytes of one layer (incl. its rows' padding)
    size_t esz;
};

inline char *layer(void *buf, const Geo &q, long index) { return static_cast<char *>(buf) + index * (long)q.lp; }

// Exchange ONE layer per side of `buf` (a slab array: ghost layer 0, own layers 1..n, ghost layer n+1).
// Order per peer: the "downward" pair first, then the "upward" pair — messages to one peer match in issue order, so the
// 2-rank periodic ring and the 1-rank self exchange pair up correctly.
template <class Ops>
int exchange(Ops &ops, const Geo &q, void *buf, int lower, int upper, void *st)
{
    if (lower < 0 && upper < 0) return 0;
    SLAB_TRY(ops.group_start());
    if (lower >= 0) SLAB_TRY(ops.send(layer(buf, q, 1), q.lp, lower, st));
    if (upper >= 0) {
        SLAB_TRY(ops.recv(layer(buf, q, q.nloc + 1), q.lp, upper, st));
        SLAB_TRY(ops.send(layer(buf, q, q.nloc), q.lp, upper, st));
    }
    if (lower >= 0) SLAB_TRY(ops.recv(layer(buf, q, 0), q.lp, lower, st));
    return ops.group_end();
}

// Exchange TWO layers per side of `ext` (layers 0,1 | own 2..n+1 | n+2,n+3): the halo of a two-level sweep.
template <class Ops>
int exchange2(Ops &ops, const Geo &q, void *ext, int lower, int upper, void *st)
{
    if (lower < 0 && upper < 0) return 0;
    SLAB_TRY(ops.group_start());
    if (lower >= 0) SLAB_TRY(ops.send(layer(ext, q, 2), 2 * q.lp, lower, st));             // own first two layers -> lower
    if (upper >= 0) SLAB_TRY(ops.recv(layer(ext, q, q.nloc + 2), 2 * q.lp, upper, st));    // upper halo <- upper
    if (upper >= 0) SLAB_TRY(ops.send(layer(ext, q, q.nloc), 2 * q.lp, upper, st));        // own last two layers -> upper
    if (lower >= 0) SLAB_TRY(ops.recv(layer(ext, q, 0), 2 * q.lp, lower, st));             // lower halo <- lower
    return ops.group_end();
}

// faces of a sub-slab of layers [first, first+count) (1-based valid layers of the slab): the inter-layer faces inside the
// slab are real data (SKIP); physical / exchanged faces keep the slab's descriptor with the index translated
inline void sub_faces(const pdehip_bc_face_t *faces, long nloc, long first, long count, pdehip_bc_face_t *out)
{
    for (int i = 0; i < 2 * PDEHIP_MAX_DIM; i++) out[i] = faces[i];
    if (first > 1) out[0].kind = PDEHIP_BC_SKIP;
    else out[0].index1 -= (first - 1), out[0].index2 -= (first - 1);
    if (first + count - 1 < nloc) out[1].kind = PDEHIP_BC_SKIP;
    else out[1].index1 -= (first - 1), out[1].index2 -= (first - 1);
}

inline void local_faces(const pdehip_bc_face_t *src, int lower, int upper, pdehip_bc_face_t *dst)
{
    for (int i = 0; i < 2 * PDEHIP_MAX_DIM; i++) dst[i] = src[i];
    if (lower >= 0) dst[0].kind = PDEHIP_BC_SKIP;   // exchanged sides: the ghost layer holds real data
    if (upper >= 0) dst[1].kind = PDEHIP_BC_SKIP;
}

// xplain code of the two-level kernels: 0 both ends physical, 1 both exchanged, 2 lower physical, 3 upper physical
inline int xends(int lower, int upper) { return (lower >= 0 && upper >= 0) ? 1 : (lower < 0 && upper < 0) ? 0 : (lower < 0 ? 2 : 3); }

// ---------------------------------------------------------------------------------------------------------
// nsteps explicit Euler steps of the diffusion equation; everything is enqueued without host synchronisation:
//   comp stream : interior kernel (layers 2..n-1)   ............................ | next step
//   halo stream : boundary kernels (layers 1, n) - send/recv of the new layers 1, n
// The exchange of step s+1's input overlaps the interior kernel of step s.  BCs of the faces this rank owns are
// evaluated inside the kernels; exchanged faces read the received layers.
// ---------------------------------------------------------------------------------------------------------
template <class Ops>
int euler_run(Ops &ops, const pdehip_grid_t *g, const Geo &q, const pdehip_rhs_t *rhs, int lower, int upper, void *buf_a, void *buf_b,
              double dt, int64_t nsteps, void **result, void *comp)
{
    void *halo = ops.halo();
    pdehip_bc_face_t faces[2 * PDEHIP_MAX_DIM];
    local_faces(rhs->bc_c, lower, upper, faces);
    auto sub_step = [&](void *st, void *cur, void *nxt, long first, long count) -> int {
        if (count <= 0) return 0;
        pdehip_grid_t gs = *g;
        gs.shape[0] = count;
        pdehip_bc_face_t sf[2 * PDEHIP_MAX_DIM];
        sub_faces(faces, q.nloc, first, count, sf);
        char *pc = layer(cur, q, first - 1), *pn = layer(nxt, q, first - 1);
        return ops.lap(&gs, pc, pc, pn, K_EULER, rhs->param, dt, 0.0, sf, st, nullptr);
    };
    void *cur = buf_a, *nxt = buf_b;
    // ghost layers of the initial state
    SLAB_TRY(ops.record(EV_COMP, comp));
    SLAB_TRY(ops.wait(halo, EV_COMP));
    SLAB_TRY(exchange(ops, q, cur, lower, upper, halo));
    for (int64_t s = 0; s < nsteps; s++) {
        // interior layers need no exchanged data; they must wait for the boundary layers of `cur` (written on the halo
        // stream in the previous step)
        if (s > 0) SLAB_TRY(ops.wait(comp, EV_BND));
        SLAB_TRY(sub_step(comp, cur, nxt, 2, q.nloc - 2));
        SLAB_TRY(ops.record(EV_COMP, comp));
        // boundary layers: the received ghost layers are ordered by the halo stream itself
        SLAB_TRY(sub_step(halo, cur, nxt, 1, 1));
        if (q.nloc > 1) SLAB_TRY(sub_step(halo, cur, nxt, q.nloc, 1));
        SLAB_TRY(ops.record(EV_BND, halo));
        SLAB_TRY(exchange(ops, q, nxt, lower, upper, halo));   // overlaps the interior kernel
        // the next step overwrites `cur`: its interior kernel (comp) and boundary kernels (halo, in order) must be done;
        // the halo stream additionally waits for this step's interior kernel
        SLAB_TRY(ops.wait(halo, EV_COMP));
        void *t = cur; cur = nxt; nxt = t;
    }
    SLAB_TRY(ops.record(EV_HALO, halo));   // make the compute stream see everything
    SLAB_TRY(ops.wait(comp, EV_HALO));
    *result = cur;
    return 0;
}

// ---------------------------------------------------------------------------------------------------------
// Two Euler steps per sweep (temporal blocking) — halves both the HBM traffic per step and the NUMBER of halo exchanges:
// two layers per side are exchanged once per two steps.  The slab is copied into private arrays `ext[0/1]` with two
// halo layers per side (layers 0,1 | own 2..n+1 | n+2,n+3):
//   comp stream : interior sweep (own layers 4..n-1; reads own layers only)        ............ | next pair
//   halo stream : boundary sweeps (layers 2,3 and n,n+1; read the received halos) - send/recv of the new boundary layers
// Requires >= 4 own layers on EVERY rank and grid / faces the two-level kernel covers (the caller decides globally).
// ---------------------------------------------------------------------------------------------------------
template <class Ops>
int euler2_run(Ops &ops, const pdehip_grid_t *g, const Geo &q, const pdehip_rhs_t *rhs, int lower, int upper, void *buf_a, void *ext0,
               void *ext1, double dt, int64_t nsteps, void **result, void *comp)
{
    void *halo = ops.halo();
    const int xe = xends(lower, upper);
    char *cur = static_cast<char *>(ext0), *nxt = static_cast<char *>(ext1);
    // own layers: slab layers 1..nloc -> private layers 2..nloc+1 (same row layout, one layer further in)
    SLAB_TRY(ops.copy(layer(cur, q, 2), layer(buf_a, q, 1), (size_t)q.nloc * q.lp, comp));
    pdehip_bc_face_t faces[2 * PDEHIP_MAX_DIM];
    local_faces(rhs->bc_c, lower, upper, faces);
    // two steps on private layers [first, first+count)  (ends > 0: the first and the last `ends` layers of the range in one launch)
    auto sweep2 = [&](void *st, long first, long count, int ends) -> int {
        if (count <= 0) return 0;
        pdehip_grid_t gs = *g;
        gs.shape[0] = count;
        bool done = false;
        // the interior sweep reads own layers only (plain on both sides); the two-ended boundary sweep meets the physical faces
        SLAB_TRY(ops.euler2(&gs, layer(cur, q, first - 1), layer(nxt, q, first - 1), rhs->param, dt, faces, st, &done, ends ? xe : 1, false, ends));
        if (!done) return ops.fail("internal: two-step kernel refused a sub-slab");
        return 0;
    };
    SLAB_TRY(ops.record(EV_COMP, comp));
    SLAB_TRY(ops.wait(halo, EV_COMP));
    SLAB_TRY(exchange2(ops, q, cur, lower, upper, halo));
    int64_t s = 0;
    int pair = 0;
    // Round 5 (PDEHIP_SLAB_THICK=<layers>, off by default: measured slower): THICK boundary chunks on the compute stream.  The thin boundary sweep below (2 + 2 layers on the halo
    // stream) starves next to the interior sweep - workgroups of a full-occupancy sweep hold their registers for the whole sweep - and
    // ends ~12 us after it; the next interior sweep waits for it (profiles/r05_probe_block.md).  Here the sweep itself is cut in two
    // launches on ONE stream: the first and the last `thick` layers (they need the received halos; as chunks of the ordinary march they
    // cost what they cost inside the whole sweep), then the layers in between while the halo stream sends / receives the new boundary
    // layers.  The critical path of a pair was meant to be the sweep, with the whole inner launch for the exchange to hide in.  Measured
    // (profiles/r05_probe_block.md): the RCCL kernel, dispatched next to the inner launch, takes 46 us instead of 12 and ends after it, the
    // next boundary chunks follow a stream hand-over later - 99 us per pair against 86 with the thin sweep below.
    long thick = ops.slab_thick();
    if (thick > (q.nloc - 4) / 2) thick = (q.nloc - 4) / 2;
    if (thick >= 2 && (lower >= 0 || upper >= 0)) {
        // (both launches must be ones the two-step kernel takes: asked with a dry run, else the schedule below)
        auto covered = [&](long first, long count, int ends) -> bool {
            pdehip_grid_t gs = *g;
            gs.shape[0] = count;
            bool done = false;
            return ops.euler2(&gs, layer(cur, q, first - 1), layer(nxt, q, first - 1), rhs->param, dt, faces, comp, &done, ends ? xe : 1, true, ends) == 0 && done;
        };
        if (!covered(2, q.nloc, (int)thick) || !covered(2 + thick, q.nloc - 2 * thick, 0)) thick = 0;
    }
    if (thick >= 2 && (lower >= 0 || upper >= 0)) {
        SLAB_TRY(ops.record(EV_HALO, halo));
        for (; s + 2 <= nsteps; s += 2, pair++) {
            SLAB_TRY(ops.wait(comp, EV_HALO));                 // halos of `cur` (exchange of the previous pair)
            SLAB_TRY(sweep2(comp, 2, q.nloc, (int)thick));     // own layers 2 .. thick+1 and nloc+2-thick .. nloc+1
            SLAB_TRY(ops.record(EV_BND, comp));
            if (s + 2 < nsteps) {
                SLAB_TRY(ops.wait(halo, EV_BND));
                SLAB_TRY(exchange2(ops, q, nxt, lower, upper, halo));   // overlaps the inner launch
                SLAB_TRY(ops.record(EV_HALO, halo));
            }
            SLAB_TRY(sweep2(comp, 2 + thick, q.nloc - 2 * thick, 0));
            char *t = cur; cur = nxt; nxt = t;
        }
        SLAB_TRY(ops.record(EV_COMP, comp));
        SLAB_TRY(ops.wait(halo, EV_COMP));
    }
    for (; s + 2 <= nsteps; s += 2, pair++) {
        // The boundary sweep and the exchange are ENQUEUED FIRST, on the (high-priority) halo stream: their few workgroups are
        // dispatched before the interior sweep fills the chip with workgroups that live for the whole sweep - enqueued behind it
        // they waited for its end and the exchange was exposed (0.043 vs 0.031 ms per step on a 64 x 512 x 512 slab).
        // Dependences: boundary(p) <- exchange(p-1) (stream order) and interior(p-1) (EV_COMP, awaited at the end of the last
        // turn); interior(p) <- boundary(p-1): it reads the layers that sweep wrote and overwrites layers it read - two events in
        // turn, so that interior(p) does not wait for boundary(p), which was recorded before it.
        SLAB_TRY(sweep2(halo, 2, q.nloc, 2));   // own layers 2,3 and nloc,nloc+1 (needs the received halo layers)
        SLAB_TRY(ops.record(pair % 2 ? EV_BND2 : EV_BND, halo));
        if (s + 2 < nsteps) SLAB_TRY(exchange2(ops, q, nxt, lower, upper, halo));   // overlaps the interior sweep
        if (pair > 0) SLAB_TRY(ops.wait(comp, pair % 2 ? EV_BND : EV_BND2));        // boundary sweep of the PREVIOUS pair
        SLAB_TRY(sweep2(comp, 4, q.nloc - 4, 0));
        SLAB_TRY(ops.record(EV_COMP, comp));
        // the next pair overwrites `cur` and its boundary sweep reads the interior layers written now
        SLAB_TRY(ops.wait(halo, EV_COMP));
        char *t = cur; cur = nxt; nxt = t;
    }
    SLAB_TRY(ops.record(EV_HALO, halo));
    SLAB_TRY(ops.wait(comp, EV_HALO));
    if (s < nsteps) {
        // odd step count: one single step; layers 1 and nloc+2 act as its ghost layers (already exchanged)
        SLAB_TRY(ops.lap(g, layer(cur, q, 1), layer(cur, q, 1), layer(nxt, q, 1), K_EULER, rhs->param, dt, 0.0, faces, comp, nullptr));
        char *t = cur; cur = nxt; nxt = t;
    }
    SLAB_TRY(ops.copy(layer(buf_a, q, 1), layer(cur, q, 2), (size_t)q.nloc * q.lp, comp));
    *result = buf_a;
    return 0;
}

// `depth` layers per side of `ext` (layers 0..depth-1 | own depth..n+depth-1 | n+depth..n+2*depth-1): halos of several sweeps at once
template <class Ops>
int exchange_deep(Ops &ops, const Geo &q, void *ext, long depth, int lower, int upper, void *st)
{
    if (lower < 0 && upper < 0) return 0;
    SLAB_TRY(ops.group_start());
    if (lower >= 0) SLAB_TRY(ops.send(layer(ext, q, depth), (size_t)depth * q.lp, lower, st));                    // own first layers -> lower
    if (upper >= 0) SLAB_TRY(ops.recv(layer(ext, q, q.nloc + depth), (size_t)depth * q.lp, upper, st));           // upper halo <- upper
    if (upper >= 0) SLAB_TRY(ops.send(layer(ext, q, q.nloc), (size_t)depth * q.lp, upper, st));                   // own last layers -> upper
    if (lower >= 0) SLAB_TRY(ops.recv(layer(ext, q, 0), (size_t)depth * q.lp, lower, st));                        // lower halo <- lower
    return ops.group_end();
}

// ---------------------------------------------------------------------------------------------------------
// Communication-avoiding variant of euler2_run (round 6): FOUR halo layers per side, ONE exchange per FOUR steps (two two-step sweeps).
//   private arrays (layers 0..3 | own 4..n+3 | n+4..n+7), group of four steps, everything that computes on ONE stream:
//     A : cur -> nxt, own layers and two more per exchanged side ([2, n+6): valid because cur holds four halo layers)
//     B : nxt -> cur, own layers ([4, n+4): reads the two extra layers A produced)
//     X : send / receive the four outermost own layers of cur (halo stream)
//   A is cut in two launches: A_int = the layers that read own cells only ([6, n+2)) runs while X of the group before is in flight, A_bnd
//   (4 + 4 layers, two-ended launch) waits for it.  So the exchange has a whole sweep to hide behind, nothing small ever has to find wave
//   slots NEXT to a sweep except the RCCL kernel itself (the boundary -> send / receive -> boundary cycle of euler2_run - 88 us per pair
//   against 71 us of interior, profiles/r05_probe_block.md - is gone), and the number of messages, RCCL groups, events and stream
//   hand-overs per step halves.  Price: A computes 4 extra layers per group (+3 % at 64 own layers).  mode 2 also cuts B (B_bnd first, X
//   starts behind it and has B_int and the next A_int to hide behind).
// Same arithmetic as every other path: the redundant layers are computed from the same inputs in the same order on both ranks.
// Requires >= 8 own layers on EVERY rank (the caller decides globally, pdehip_slab_euler4_supported).
// Replaces: the per-step blocking exchange of pde/solvers/explicit_mpi.py:133-226 + pde/backends/numba_mpi/backend.py:163-194.
// ---------------------------------------------------------------------------------------------------------
template <class Ops>
int euler4_run(Ops &ops, const pdehip_grid_t *g, const Geo &q, const pdehip_rhs_t *rhs, int lower, int upper, void *buf_a, void *ext0,
               void *ext1, double dt, int64_t nsteps, void **result, void *comp)
{
    constexpr long D = 4;   // halo depth
    void *halo = ops.halo();
    const int xe = xends(lower, upper);
    const long n = q.nloc;
    const int mode = ops.deep_mode();
    char *cur = static_cast<char *>(ext0), *nxt = static_cast<char *>(ext1);
    SLAB_TRY(ops.copy(layer(cur, q, D), layer(buf_a, q, 1), (size_t)n * q.lp, comp));
    pdehip_bc_face_t faces[2 * PDEHIP_MAX_DIM];
    local_faces(rhs->bc_c, lower, upper, faces);
    // two steps src -> dst on private layers [first, first + count); xplain as in euler2_run (1: real layers beyond both ends);
    // ends > 0: the first and the last `ends` layers of the range only
    auto sweep2 = [&](void *st, char *src, char *dst, long first, long count, int xplain, int ends) -> int {
        if (count <= 0) return 0;
        pdehip_grid_t gs = *g;
        gs.shape[0] = count;
        pdehip_bc_face_t sf[2 * PDEHIP_MAX_DIM];
        for (int i = 0; i < 2 * PDEHIP_MAX_DIM; i++) sf[i] = faces[i];
        // a physical upper face: its cell indices count from the first layer of the range
        sf[1].index1 += count - n; sf[1].index2 += count - n;
        bool done = false;
        SLAB_TRY(ops.euler2(&gs, layer(src, q, first - 1), layer(dst, q, first - 1), rhs->param, dt, sf, st, &done, xplain, false, ends));
        if (!done) return ops.fail("internal: two-step kernel refused a sub-slab");
        return 0;
    };
    const long lo = lower >= 0 ? D - 2 : D, hi = upper >= 0 ? n + D + 2 : n + D;   // layers A produces
    SLAB_TRY(ops.record(EV_COMP, comp));
    SLAB_TRY(ops.wait(halo, EV_COMP));
    SLAB_TRY(exchange_deep(ops, q, cur, D, lower, upper, halo));
    SLAB_TRY(ops.record(EV_HALO, halo));
    int64_t s = 0;
    for (; s + 4 <= nsteps; s += 4) {
        SLAB_TRY(sweep2(comp, cur, nxt, lo + D, hi - lo - 2 * D, 1, 0));   // A_int: reads own layers only
        SLAB_TRY(ops.wait(comp, EV_HALO));                                 // halos of cur have landed
        SLAB_TRY(sweep2(comp, cur, nxt, lo, hi - lo, xe, (int)D));         // A_bnd
        if (mode == 2 && n > 2 * D) {
            SLAB_TRY(sweep2(comp, nxt, cur, D, n, xe, (int)D));            // B_bnd: the layers the exchange sends
            SLAB_TRY(ops.record(EV_BND, comp));
            SLAB_TRY(sweep2(comp, nxt, cur, 2 * D, n - 2 * D, 1, 0));      // B_int
        } else {
            SLAB_TRY(sweep2(comp, nxt, cur, D, n, xe, 0));                 // B
            SLAB_TRY(ops.record(EV_BND, comp));
        }
        if (s + 4 < nsteps) {   // (nothing reads the halos after the last step)
            SLAB_TRY(ops.wait(halo, EV_BND));
            SLAB_TRY(exchange_deep(ops, q, cur, D, lower, upper, halo));
            SLAB_TRY(ops.record(EV_HALO, halo));
        }
    }
    SLAB_TRY(ops.wait(comp, EV_HALO));
    const int64_t rest = nsteps - s;
    if (rest >= 2) {
        // two steps: the own layers, and one more per exchanged side when a single step follows
        const long lo2 = (rest == 3 && lower >= 0) ? D - 1 : D, hi2 = (rest == 3 && upper >= 0) ? n + D + 1 : n + D;
        SLAB_TRY(sweep2(comp, cur, nxt, lo2, hi2 - lo2, xe, 0));
        char *t = cur; cur = nxt; nxt = t;
    }
    if (rest % 2) {
        // one single step; layers D-1 and n+D act as its ghost layers (exchanged, or computed by the sweep above)
        SLAB_TRY(ops.lap(g, layer(cur, q, D - 1), layer(cur, q, D - 1), layer(nxt, q, D - 1), K_EULER, rhs->param, dt, 0.0, faces, comp, nullptr));
        char *t = cur; cur = nxt; nxt = t;
    }
    // the halo stream is idle here (its last exchange was awaited above); the next run may reuse the private arrays
    SLAB_TRY(ops.copy(layer(buf_a, q, 1), layer(cur, q, D), (size_t)n * q.lp, comp));
    *result = buf_a;
    return 0;
}

// ---------------------------------------------------------------------------------------------------------
// Four steps per exchange with the BOUNDARY WORK OFF THE CRITICAL PATH (schedule 3 of pdehip_slab_euler4_run).
// In schedules 1 / 2 above the boundary parts of the sweeps are small launches IN the chain of sweeps (17 us each for 8 us of traffic: a
// short march is latency-bound) and the stream hand-over behind the RCCL kernel is exposed (profiles/r06_probe_slab.md).  Here a group of
// four steps is two INDEPENDENT chains that meet once per group:
//   comp stream : A_int (cur -> mid, the layers that read own cells only), B_int (mid -> nxt, the layers that read only what A_int wrote)
//   halo stream : P = the boundary layers of nxt (own layers < 4 from an exchanged side, four steps ahead) straight from cur and its four
//                 halo layers - two short two-step passes through a scratch array (12 -> 8 -> 4 layers per side); X = send / receive of
//                 those layers and of the halos of nxt.
//   A_int(g+1) waits for P(g) (recorded a whole group earlier: no hand-over latency); P(g+1) for B_int(g) and, in stream order, X(g).
// P recomputes what A_int / B_int also touch at the seam (12 + 8 layers per side and group instead of 4 + 4: +6 % of a 64-layer slab) - from
// the same inputs in the same order, so every cell still gets the bits of the serial run.  Three state arrays in rotation (cur -> mid -> nxt):
// X(g) writes halos nobody reads any more, P(g) writes layers of nxt that B_int(g-1) has finished reading.
// ---------------------------------------------------------------------------------------------------------
template <class Ops>
int euler4p_run(Ops &ops, const pdehip_grid_t *g, const Geo &q, const pdehip_rhs_t *rhs, int lower, int upper, void *buf_a, void *const *ext4,
                double dt, int64_t nsteps, void **result, void *comp)
{
    constexpr long D = 4;
    void *halo = ops.halo();
    const int xe = xends(lower, upper);
    const long n = q.nloc;
    char *cur = static_cast<char *>(ext4[0]), *mid = static_cast<char *>(ext4[1]), *nxt = static_cast<char *>(ext4[2]);
    char *tmp = static_cast<char *>(ext4[3]);
    SLAB_TRY(ops.copy(layer(cur, q, D), layer(buf_a, q, 1), (size_t)n * q.lp, comp));
    pdehip_bc_face_t faces[2 * PDEHIP_MAX_DIM];
    local_faces(rhs->bc_c, lower, upper, faces);
    auto sweep2 = [&](void *st, char *src, char *dst, long first, long count, int xplain, int ends) -> int {
        if (count <= 0) return 0;
        pdehip_grid_t gs = *g;
        gs.shape[0] = count;
        pdehip_bc_face_t sf[2 * PDEHIP_MAX_DIM];
        for (int i = 0; i < 2 * PDEHIP_MAX_DIM; i++) sf[i] = faces[i];
        sf[1].index1 += count - n; sf[1].index2 += count - n;   // a physical upper face: indices count from the first layer of the range
        bool done = false;
        SLAB_TRY(ops.euler2(&gs, layer(src, q, first - 1), layer(dst, q, first - 1), rhs->param, dt, sf, st, &done, xplain, false, ends));
        if (!done) return ops.fail("internal: two-step kernel refused a sub-slab");
        return 0;
    };
    // `depth` layers next to every EXCHANGED side of the range [first, first + count), real layers beyond both ends of each piece
    auto seams = [&](void *st, char *src, char *dst, long first, long count, long depth) -> int {
        if (lower >= 0 && upper >= 0) return sweep2(st, src, dst, first, count, 1, (int)depth);   // both in ONE launch
        if (lower >= 0) return sweep2(st, src, dst, first, depth, 1, 0);
        if (upper >= 0) return sweep2(st, src, dst, first + count - depth, depth, 1, 0);
        return 0;
    };
    // layers the interior sweeps produce: a physical side is part of them (its face is applied by the kernel)
    const long a0 = lower >= 0 ? D + 2 : D, a1 = upper >= 0 ? n + D - 2 : n + D;       // A_int
    const long b0 = lower >= 0 ? 2 * D : D, b1 = upper >= 0 ? n : n + D;               // B_int
    const long lo = lower >= 0 ? D - 2 : D, hi = upper >= 0 ? n + D + 2 : n + D;       // pass 1 of P spans [lo, hi)
    SLAB_TRY(ops.record(EV_COMP, comp));
    SLAB_TRY(ops.wait(halo, EV_COMP));
    SLAB_TRY(exchange_deep(ops, q, cur, D, lower, upper, halo));
    const int64_t groups = nsteps / 4, rest = nsteps - 4 * groups;
    // Schedule 4 ("lean", PDEHIP_SLAB_DEEP_MODE=4): the second boundary pass reads the layers A_int produces anyway instead of recomputing them -
    // pass 1 only the 4 layers per side that need the halos (straight into `mid`, next to A_int's range), pass 2 behind A_int of the SAME group.
    // 4 + 4 layers of boundary work per side and group instead of 8 + 4; the exchange then has B_int and the next A_int to hide behind.
    const bool lean = ops.deep_mode() == 4;
    for (int64_t k = 0; k < groups; k++) {
        if (k > 0) {
            SLAB_TRY(ops.wait(comp, EV_BND2));   // P(k-1): the seam layers of cur
            SLAB_TRY(ops.wait(halo, EV_BND));    // B_int(k-1): cur next to the seams, and it has let go of what P(k) overwrites
        }
        // boundary chain first: its few workgroups should be dispatched before the sweeps fill the chip
        if (lean) {
            SLAB_TRY(seams(halo, cur, mid, lo, hi - lo, D));          // two steps: the 4 layers per exchanged side that read the halos
            SLAB_TRY(sweep2(comp, cur, mid, a0, a1 - a0, xe, 0));     // A_int
            SLAB_TRY(ops.record(EV_COMP, comp));
            SLAB_TRY(ops.wait(halo, EV_COMP));
            SLAB_TRY(seams(halo, mid, nxt, D, n, D));                 // two more: the 4 own layers per exchanged side
        } else {
            SLAB_TRY(seams(halo, cur, tmp, lo, hi - lo, 2 * D));      // two steps: 8 layers per exchanged side
            SLAB_TRY(seams(halo, tmp, nxt, D, n, D));                 // two more: the 4 own layers per exchanged side
        }
        SLAB_TRY(ops.record(EV_BND2, halo));
        if (k + 1 < groups || rest > 0) SLAB_TRY(exchange_deep(ops, q, nxt, D, lower, upper, halo));
        if (!lean) SLAB_TRY(sweep2(comp, cur, mid, a0, a1 - a0, xe, 0));     // A_int
        SLAB_TRY(sweep2(comp, mid, nxt, b0, b1 - b0, xe, 0));     // B_int
        SLAB_TRY(ops.record(EV_BND, comp));
        char *t = cur; cur = nxt; nxt = mid; mid = t;
    }
    SLAB_TRY(ops.record(EV_HALO, halo));
    SLAB_TRY(ops.wait(comp, EV_HALO));
    if (rest >= 2) {
        const long lo2 = (rest == 3 && lower >= 0) ? D - 1 : D, hi2 = (rest == 3 && upper >= 0) ? n + D + 1 : n + D;
        SLAB_TRY(sweep2(comp, cur, mid, lo2, hi2 - lo2, xe, 0));
        char *t = cur; cur = mid; mid = t;
    }
    if (rest % 2) {
        SLAB_TRY(ops.lap(g, layer(cur, q, D - 1), layer(cur, q, D - 1), layer(mid, q, D - 1), K_EULER, rhs->param, dt, 0.0, faces, comp, nullptr));
        char *t = cur; cur = mid; mid = t;
    }
    SLAB_TRY(ops.copy(layer(buf_a, q, 1), layer(cur, q, D), (size_t)n * q.lp, comp));
    *result = buf_a;
    return 0;
}

// ---------------------------------------------------------------------------------------------------------
// k_out = dt * rhs(in) on the slab, followed in the same sweep (flags & F_FUSED_STAGE) by the Runge-Kutta combination
// `sf`; Euler form (out = in + dt*rhs(in)) with euler = true and sf = NULL.  `in` (and for Cahn-Hilliard `out`) are slab
// arrays with ONE SPARE LAYER of allocated memory beyond each ghost layer, so that the same memory is the two-halo-layer
// array of the fused Cahn-Hilliard sweep (`in - lp`).  The halo of `in` is exchanged first (one layer; two layers for the
// fused Cahn-Hilliard sweep, whose mu then needs no exchange of its own — the reference exchanges c AND mu).
// Everything on `st`; no overlap (the Runge-Kutta loops are not exchange-bound at one exchange per full-slab sweep).
// ---------------------------------------------------------------------------------------------------------
template <class Ops>
int rhs_sweep(Ops &ops, const pdehip_grid_t *g, const Geo &q, const pdehip_rhs_t *rhs, int lower, int upper, int flags, void *in, void *out,
              double dt, bool euler, const StageFuse *sf, void *st, double t = 0.0)
{
    // faces with explicit time dependence: their coefficient arrays for the time of THIS evaluation (pdehip_rhs_t::bc_program)
    if (rhs->bc_program) SLAB_TRY(ops.refresh(rhs->bc_program, t, in, st));
    pdehip_bc_face_t fc[2 * PDEHIP_MAX_DIM], fm[2 * PDEHIP_MAX_DIM];
    local_faces(rhs->bc_c, lower, upper, fc);
    const bool fuse_stage = sf && (flags & F_FUSED_STAGE);
    if (rhs->kind == PDEHIP_RHS_DIFFUSION) {
        SLAB_TRY(exchange(ops, q, in, lower, upper, st));
        if (euler) return ops.lap(g, in, in, out, K_EULER, rhs->param, dt, 0.0, fc, st, nullptr);
        if (fuse_stage) return ops.lap(g, in, nullptr, out, K_STAGE, rhs->param, dt, 0.0, fc, st, sf);
        SLAB_TRY(ops.lap(g, in, nullptr, out, K_SCALED, rhs->param, dt, 0.0, fc, st, nullptr));
        return sf ? ops.combine(g, out, *sf, st) : 0;
    }
    local_faces(rhs->bc_mu, lower, upper, fm);
    if (flags & F_FUSED_CH) {
        SLAB_TRY(exchange2(ops, q, layer(in, q, -1), lower, upper, st));
        bool done = false;
        // a kind-1/2/4 stage does not store the slope: the kernel gets no `out`
        SLAB_TRY(ops.ch_fused(g, in, fuse_stage && sf->kind != 0 && sf->kind != 3 ? nullptr : out, rhs->param, dt, euler, fc, fm, st, &done,
                              xends(lower, upper), false, fuse_stage ? sf : nullptr));
        if (!done && fuse_stage && out) {
            // The sweep itself is covered, its stage epilogue is not: on grids that need overlapping tiles (an odd row count, rows
            // that end inside a vector) cells are computed twice, so an epilogue that overwrites one of its own operands - the RK4
            // update writes the new state over y - is refused (launch_euler2_tv).  Slope alone into `out` (every caller passes a
            // free array, also for the kinds that do not store it), then the pointwise combination: what pdehip_rk4_step does.
            SLAB_TRY(ops.ch_fused(g, in, out, rhs->param, dt, euler, fc, fm, st, &done, xends(lower, upper), false, nullptr));
            if (done) return ops.combine(g, out, *sf, st);
        }
        if (!done) return ops.fail("slab sweep: grid or faces are not covered by the two-level kernel (flags were decided wrongly)");
        return (sf && !fuse_stage) ? ops.combine(g, out, *sf, st) : 0;
    }
    // two kernels, two exchanges (c, then mu) — the reference's sequence
    if (!rhs->scratch_mu) return ops.fail("slab sweep: Cahn-Hilliard needs a scratch slab for mu");
    SLAB_TRY(exchange(ops, q, in, lower, upper, st));
    SLAB_TRY(ops.lap(g, in, nullptr, rhs->scratch_mu, K_CH_MU, 0.0, 0.0, rhs->param, fc, st, nullptr));
    SLAB_TRY(exchange(ops, q, rhs->scratch_mu, lower, upper, st));
    if (euler) return ops.lap(g, rhs->scratch_mu, in, out, K_EULER, 1.0, dt, 0.0, fm, st, nullptr);
    SLAB_TRY(ops.lap(g, rhs->scratch_mu, nullptr, out, K_SCALED, 1.0, dt, 0.0, fm, st, nullptr));
    return sf ? ops.combine(g, out, *sf, st) : 0;
}

// one classical RK4 step in place on y (pde/solvers/runge_kutta.py:52-61); w = k1..k4, tmp (slab arrays; y, k4 and tmp
// serve as stage inputs and need the spare layers); same stage sequence as pdehip_rk4_step
template <class Ops>
int rk4_step(Ops &ops, const pdehip_grid_t *g, const Geo &q, const pdehip_rhs_t *rhs, int lower, int upper, int flags, void *y, void *const *w,
             double dt, void *st, double t = 0.0)
{
    void *k1 = w[0], *k2 = w[1], *k3 = w[2], *k4 = w[3], *tmp = w[4];
    StageFuse sf;
    memset(&sf, 0, sizeof(sf));
    sf.y = y; sf.c_new = 0.5; sf.out2 = tmp;
    SLAB_TRY(rhs_sweep(ops, g, q, rhs, lower, upper, flags, y, k1, dt, false, &sf, st, t));
    sf.out2 = k4;   // the array of k4 doubles as the second stage-input buffer; k4 itself is consumed by the last sweep
    SLAB_TRY(rhs_sweep(ops, g, q, rhs, lower, upper, flags, tmp, k2, dt, false, &sf, st, t + 0.5 * dt));
    sf.c_new = 1.0; sf.out2 = tmp;
    SLAB_TRY(rhs_sweep(ops, g, q, rhs, lower, upper, flags, k4, k3, dt, false, &sf, st, t + 0.5 * dt));
    sf.kind = 1; sf.k[0] = k1; sf.k[1] = k2; sf.k[2] = k3; sf.out2 = y;
    // unfused combination needs k4 stored: its array is free again (its role as stage input ended with the third sweep)
    return rhs_sweep(ops, g, q, rhs, lower, upper, flags, tmp, k4, dt, false, &sf, st, t + dt);
}

// one RKF45 attempt (pde/solvers/runge_kutta.py:135-153): ynew and *err_dev (this rank's max-norm; the caller reduces);
// w = k1..k6, tmp; y, ynew and tmp serve as stage inputs (spare layers); same stage sequence as pdehip_rkf45_attempt
template <class Ops>
int rkf45_attempt(Ops &ops, const pdehip_grid_t *g, const Geo &q, const pdehip_rhs_t *rhs, int lower, int upper, int flags, void *y, void *ynew,
                  void *const *w, double dt, double *err_dev, void *st, double t = 0.0)
{
    static const double A[6] = {0.0, 1.0 / 4, 3.0 / 8, 12.0 / 13, 1.0, 1.0 / 2};   // stage times t + a_s * dt (runge_kutta.py:92-98)
    void *tmp = w[6];
    void *t_in = y, *t_out = tmp;
    for (int s = 0; s < 5; s++) {
        StageFuse sf;
        memset(&sf, 0, sizeof(sf));
        sf.y = y; sf.out2 = t_out;
        const double *row = rkf45_row(s);
        for (int m = 0; m < s; m++) { sf.k[m] = w[m]; sf.c[m] = row[m]; }
        sf.c_new = row[s];
        SLAB_TRY(rhs_sweep(ops, g, q, rhs, lower, upper, flags, t_in, w[s], dt, false, &sf, st, t + A[s] * dt));
        t_in = t_out;
        t_out = (t_out == tmp) ? ynew : tmp;
    }
    StageFuse sf;
    memset(&sf, 0, sizeof(sf));
    sf.kind = 2; sf.y = y; sf.out2 = ynew; sf.err = err_dev;
    sf.k[0] = w[0]; sf.k[1] = w[2]; sf.k[2] = w[3]; sf.k[3] = w[4];
    SLAB_TRY(ops.zero(err_dev, sizeof(double), st));
    return rhs_sweep(ops, g, q, rhs, lower, upper, flags, t_in, w[5], dt, false, &sf, st, t + A[5] * dt);
}

// time-step controller, pde/solvers/base.py:572-592 (`_make_dt_adjuster`); returns 0, or 1 / 2 = below dt_min (with / without NaN)
inline int adjust_dt(double *dt, double error_rel, double dt_min, double dt_max)
{
    if (error_rel < 0.00057665) *dt *= 4.0;
    else if (std::isnan(error_rel)) *dt *= 0.25;
    else *dt *= std::fmax(0.9 * std::pow(error_rel, -0.2), 0.1);
    if (*dt > dt_max) *dt = dt_max;
    else if (*dt < dt_min) return std::isnan(error_rel) ? 1 : 2;
    return 0;
}

// nsteps Euler steps of a right-hand side without a dedicated overlapped loop (Cahn-Hilliard): one rhs_sweep per step
template <class Ops>
int euler_sweeps(Ops &ops, const pdehip_grid_t *g, const Geo &q, const pdehip_rhs_t *rhs, int lower, int upper, int flags, void *buf_a, void *buf_b,
                 double dt, int64_t nsteps, void **result, void *st)
{
    void *cur = buf_a, *nxt = buf_b;
    for (int64_t s = 0; s < nsteps; s++) {
        SLAB_TRY(rhs_sweep(ops, g, q, rhs, lower, upper, flags, cur, nxt, dt, true, nullptr, st, rhs->t + (double)s * dt));
        void *t = cur; cur = nxt; nxt = t;
    }
    *result = cur;
    return 0;
}

// The adaptive loop of pde/backends/numba/_solvers.py:249-281 around RKF45 attempts, with the MAX all-reduce of the error
// (pde/backends/base.py:678-712) — all of it here, so that a slab-parallel run costs the host ONE 8-byte read per attempt
// (the accept/reject decision) and nothing per stage.  Accepted attempts swap the roles of y / ynew (no copy); *result
// names the array holding the final state.  Returns 0, or ops.fail(...) with the reference's messages
// (pde/solvers/base.py:583-590) when the step size falls below dt_min.
template <class Ops>
int rkf45_run(Ops &ops, const pdehip_grid_t *g, const Geo &q, const pdehip_rhs_t *rhs, int lower, int upper, int flags, void *y, void *ynew,
              void *const *w, double *err_dev, pdehip_adaptive_t *a, void **result, void *st)
{
    double dt_opt = a->dt, t = a->t_start;
    void *cur = y, *nxt = ynew;
    while (true) {
        const double dt_step = std::fmax(std::fmin(dt_opt, a->t_end - t), a->dt_min);
        SLAB_TRY(rkf45_attempt(ops, g, q, rhs, lower, upper, flags, cur, nxt, w, dt_step, err_dev, st, t));
        SLAB_TRY(ops.allreduce_max(err_dev, st));
        double err = 0;
        SLAB_TRY(ops.read_scalar(&err, err_dev, st));
        const double error_rel = err / a->tolerance;
        a->attempts++;
        if (error_rel <= 1) {   // accept (false for NaN)
            a->steps++;
            t += dt_step;
            void *tmp = cur; cur = nxt; nxt = tmp;
            // running statistics of the accepted step sizes (pde/tools/math.py:125-174, Welford)
            a->stat_min = a->stat_count ? std::fmin(a->stat_min, dt_step) : dt_step;
            a->stat_max = a->stat_count ? std::fmax(a->stat_max, dt_step) : dt_step;
            const double delta = dt_step - a->stat_mean;
            a->stat_count++;
            a->stat_mean += delta / (double)a->stat_count;
            a->stat_m2 += delta * (dt_step - a->stat_mean);
        }
        if (t < a->t_end) {
            double d = dt_step;
            const int bad = adjust_dt(&d, error_rel, a->dt_min, a->dt_max);
            if (bad == 1) return ops.fail_runtime("Encountered NaN even though dt < %g", a->dt_min);
            if (bad == 2) return ops.fail_runtime("Time step below %g", a->dt_min);
            dt_opt = d;
        } else {
            break;
        }
    }
    a->dt = dt_opt;
    a->t_last = t;
    *result = cur;
    return 0;
}

}  // namespace slab
}  // namespace pdehip
