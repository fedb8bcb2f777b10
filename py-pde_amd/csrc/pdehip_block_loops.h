// pdehip_block_loops.h — BLOCK decomposition (e.g. 2 x 2 x 2 for 512^3 on 8 GPUs): one process per GPU owns a box of the grid and
// exchanges ONE ghost layer with up to six face neighbours.  Written once against an `Ops` policy, like pdehip_slab_loops.h.
//
// Reference: GridMesh (pde/grids/_mesh.py:168-806; `_get_optimal_decomposition` :59-93 picks 2 x 2 x 2 for 512^3 on 8 nodes; neighbours
// incl. the periodic wrap :401-444), the face exchange `_MPIBC` (pde/grids/boundaries/local.py:561-662: blocking send / recv of one
// layer per face inside every right-hand side), the MAX all-reduce of the adaptive error (pde/backends/base.py:678-712).
//
// Why blocks: a slab of 512^3 on 8 ranks sends 2 x 2 MiB per rank and step over the TWO xGMI links to its ring neighbours; a
// 256^3 block sends 6 x 0.5 MiB over THREE pairs of links (xGMI is point to point: the per-link bytes drop by 4 x).  The price: faces
// normal to the two fast axes are strided in memory - they are packed into contiguous staging buffers by small kernels
// (Ops::face_copy: all faces of an exchange in one launch) - and the stencil kernels read the received ghost cells from memory on every exchanged face.
//
// Only faces are exchanged (7-point stencils never read edge or corner ghosts), ONE layer per face: the two-steps-per-sweep and
// the fused Cahn-Hilliard sweeps (halo width 2, diagonal dependences) are slab-only; a block run takes one right-hand side per
// sweep and exchanges c and mu separately, like the reference.
//
// Transport contract as in pdehip_slab_loops.h: operations of a group progress together, messages between one pair of ranks match
// in issue order (with two blocks along a periodic axis the lower and the upper neighbour are the SAME rank: the order
// send-down, recv-from-up, send-up, recv-from-down pairs the messages correctly, as in slab::exchange).
#pragma once

#include "pdehip_rk_loops.h"

namespace pdehip {
namespace block {

struct Geo {
    int ndim;          // grid axes (2 or 3); arrays below are indexed by GRID axis 0 .. ndim-1
    long n[3];         // own cells per axis
    int nb[3][2];      // neighbour rank per (axis, side: 0 lower / 1 upper), -1: physical face (kept in the face table)
    size_t esz;        // bytes per element
    size_t face_elems(int axis) const
    {
        size_t m = 1;
        for (int a = 0; a < ndim; a++)
            if (a != axis) m *= (size_t)n[a];
        return m;
    }
};

// Fill the ghost layers of `buf` (a full array of the local block) on every face that has a neighbour.
// Ops::stage(axis, side, recv) returns a contiguous device / host buffer of face_elems(axis) elements (owned by the context).
struct FaceJob {
    int axis;
    long index;     // layer along `axis` (own layers 0 .. n-1, ghost layers -1 and n)
    void *packed;   // contiguous staging buffer of face_elems(axis) elements
};

template <class Ops>
int exchange(Ops &ops, const Geo &q, void *buf, void *st)
{
    FaceJob out[6], in[6];
    int n = 0;
    for (int a = 0; a < q.ndim; a++) {
        for (int side = 0; side < 2; side++) {
            if (q.nb[a][side] < 0) continue;
            // own boundary layer towards that neighbour: first own layer (index 0) goes down, last (n - 1) goes up; what comes
            // back from it lands in the ghost layer on that side (-1 / n)
            out[n] = {a, side ? q.n[a] - 1 : 0, ops.stage(a, side, false)};
            in[n] = {a, side ? q.n[a] : -1, ops.stage(a, side, true)};
            n++;
        }
    }
    if (!n) return 0;
    SLAB_TRY(ops.face_copy(q, buf, out, n, true, st));    // all faces in ONE launch
    SLAB_TRY(ops.group_start());
    for (int a = 0; a < q.ndim; a++) {
        const size_t bytes = q.face_elems(a) * q.esz;
        const int lower = q.nb[a][0], upper = q.nb[a][1];
        if (lower >= 0) SLAB_TRY(ops.send(ops.stage(a, 0, false), bytes, lower, st));
        if (upper >= 0) SLAB_TRY(ops.recv(ops.stage(a, 1, true), bytes, upper, st));
        if (upper >= 0) SLAB_TRY(ops.send(ops.stage(a, 1, false), bytes, upper, st));
        if (lower >= 0) SLAB_TRY(ops.recv(ops.stage(a, 0, true), bytes, lower, st));
    }
    SLAB_TRY(ops.group_end());
    return ops.face_copy(q, buf, in, n, false, st);
}

// face table of the block: exchanged faces hold real data (SKIP), physical faces keep their condition
inline void local_faces(const pdehip_bc_face_t *src, const Geo &q, pdehip_bc_face_t *dst)
{
    for (int i = 0; i < 2 * PDEHIP_MAX_DIM; i++) dst[i] = src[i];
    for (int a = 0; a < q.ndim; a++)
        for (int side = 0; side < 2; side++)
            if (q.nb[a][side] >= 0) dst[2 * a + side].kind = PDEHIP_BC_SKIP;
}

// The evaluator of pdehip_rk_loops.h for a block: slope = exchange + stencil sweep (with the stage epilogue where `fuse` allows),
// error norm reduced over all ranks.  `Ops` additionally provides lap / combine / allreduce_max / read_scalar / zero / refresh as in
// pdehip_slab_loops.h.
template <class Ops>
struct Eval {
    Ops &ops;
    const pdehip_grid_t *g;
    const Geo &q;
    const pdehip_rhs_t *rhs;
    bool fuse;   // Runge-Kutta stage epilogue inside the diffusion sweep (decided globally by the caller)

    int rhs_sweep(void *in, void *out, double dt, double t, bool euler, const StageFuse *sf, bool *fused, void *st)
    {
        *fused = false;
        if (rhs->bc_program) SLAB_TRY(ops.refresh(rhs->bc_program, t, in, st));
        pdehip_bc_face_t fc[2 * PDEHIP_MAX_DIM], fm[2 * PDEHIP_MAX_DIM];
        local_faces(rhs->bc_c, q, fc);
        SLAB_TRY(exchange(ops, q, in, st));
        if (rhs->kind == PDEHIP_RHS_DIFFUSION) {
            if (euler) return ops.lap(g, in, in, out, slab::K_EULER, rhs->param, dt, 0.0, fc, st, nullptr);
            if (sf && fuse) { *fused = true; return ops.lap(g, in, nullptr, out, slab::K_STAGE, rhs->param, dt, 0.0, fc, st, sf); }
            return ops.lap(g, in, nullptr, out, slab::K_SCALED, rhs->param, dt, 0.0, fc, st, nullptr);
        }
        // Cahn-Hilliard: two kernels, two exchanges (c, then mu) - the reference's sequence (pde/pdes/cahn_hilliard.py:115-122)
        if (!rhs->scratch_mu) return ops.fail("block sweep: Cahn-Hilliard needs a scratch block for mu");
        local_faces(rhs->bc_mu, q, fm);
        SLAB_TRY(ops.lap(g, in, nullptr, rhs->scratch_mu, slab::K_CH_MU, 0.0, 0.0, rhs->param, fc, st, nullptr));
        SLAB_TRY(exchange(ops, q, rhs->scratch_mu, st));
        if (euler) return ops.lap(g, rhs->scratch_mu, in, out, slab::K_EULER, 1.0, dt, 0.0, fm, st, nullptr);
        return ops.lap(g, rhs->scratch_mu, nullptr, out, slab::K_SCALED, 1.0, dt, 0.0, fm, st, nullptr);
    }
    // --- interface of rk::rk4_step / rkf45_attempt / rkf45_run ---
    int slope(void *in, void *k_out, double dt, double t, const StageFuse *sf, bool *fused, void *st)
    {
        // a fused kind-1 / kind-2 stage does not store the slope; the sweep then needs no `out`
        return rhs_sweep(in, k_out, dt, t, false, sf, fused, st);
    }
    int lincomb(void *out, const void *y, int n, const double *c, const void *const *k, void *st) { return ops.lincomb(g, out, y, n, c, k, st); }
    int rk4_combine(void *y, const void *k1, const void *k2, const void *k3, const void *k4, void *st) { return ops.rk4_combine(g, y, k1, k2, k3, k4, st); }
    int rkf45_combine(const void *y, void *ynew, const void *const *k6, double *err, void *st) { return ops.rkf45_combine(g, y, ynew, k6, err, st); }
    int euler_adaptive_combine(const void *y, const void *rate, double dt, const void *half, const void *k, void *out, double *err, void *st)
    {
        return ops.euler_adaptive_combine(g, y, rate, dt, half, k, out, err, st);
    }
    int zero(void *p, size_t bytes, void *st) { return ops.zero(p, bytes, st); }
    int reduce_error(double *err_dev, void *st) { return ops.allreduce_max(err_dev, st); }
    int read_scalar(double *host, const double *dev, void *st) { return ops.read_scalar(host, dev, st); }
    int fail_runtime(const char *fmt, double v) { return ops.fail_runtime(fmt, v); }
};

// scheme 0: `nsteps` Euler steps ping-ponging y / ynew; 1: `nsteps` RK4 steps in place on y; 2: the adaptive RKF45 loop `ctl`;
// 3: the reference's adaptive Euler loop `ctl` (work: rate, step_half, slope scratch)
template <class Ops>
int run(Ops &ops, const pdehip_grid_t *g, const Geo &q, const pdehip_rhs_t *rhs, bool fuse, int scheme, void *y, void *ynew, void *const *work,
        double *err_dev, double dt, int64_t nsteps, pdehip_adaptive_t *ctl, void **result, void *st)
{
    Eval<Ops> ev{ops, g, q, rhs, fuse};
    if (scheme == 0) {
        void *cur = y, *nxt = ynew;
        bool fused = false;
        for (int64_t s = 0; s < nsteps; s++) {
            SLAB_TRY(ev.rhs_sweep(cur, nxt, dt, rhs->t + (double)s * dt, true, nullptr, &fused, st));
            void *t = cur; cur = nxt; nxt = t;
        }
        *result = cur;
        return 0;
    }
    if (scheme == 1) {
        for (int64_t s = 0; s < nsteps; s++) SLAB_TRY(rk::rk4_step(ev, y, work, dt, rhs->t + (double)s * dt, st));
        *result = y;
        return 0;
    }
    if (scheme == 3) return rk::euler_adaptive_run(ev, y, ynew, work, err_dev, ctl, result, st);
    return rk::rkf45_run(ev, y, ynew, work, err_dev, ctl, result, st);
}

}  // namespace block
}  // namespace pdehip
