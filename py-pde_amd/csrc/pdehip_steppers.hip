// pdehip_steppers.hip — fused right-hand sides and explicit time steppers (host-side launch
// sequences on one stream; no host synchronisation inside a step loop).
//
// Kernel budget per step (algorithmic HBM traffic, values per cell):
//   diffusion Euler        : ghosts + laplace_euler                     = 1 read + 1 write
//   Cahn–Hilliard Euler    : ghosts + ch_mu (1r+1w) + ghosts + laplace_euler(mu; y=c) (2r+1w) = 5 values
//   RK stage               : lincomb (1+j reads, 1 write) + rhs_scaled; diffusion: ONE sweep per stage (rhs_stage)
#include "pdehip_common.h"

using namespace pdehip;

namespace {

// RKF45 tableau, pde/solvers/runge_kutta.py:92-112 (identical quotients)
const double B2[] = {1.0 / 4};
const double B3[] = {3.0 / 32, 9.0 / 32};
const double B4[] = {1932.0 / 2197, -7200.0 / 2197, 7296.0 / 2197};
const double B5[] = {439.0 / 216, -8.0, 3680.0 / 513, -845.0 / 4104};
const double B6[] = {-8.0 / 27, 2.0, -3544.0 / 2565, 1859.0 / 4104, -11.0 / 40};
const double A45[] = {0.0, 1.0 / 4, 3.0 / 8, 12.0 / 13, 1.0, 1.0 / 2};   // runge_kutta.py:92-98

int check_rhs(const pdehip_rhs_t *rhs)
{
    if (!rhs) PDEHIP_FAIL(E_VALUE, "rhs descriptor is NULL");
    if (rhs->kind != PDEHIP_RHS_DIFFUSION && rhs->kind != PDEHIP_RHS_CAHN_HILLIARD)
        PDEHIP_FAIL(E_NOTIMPL, "unknown rhs kind %d", rhs->kind);
    if (rhs->kind == PDEHIP_RHS_CAHN_HILLIARD && !rhs->scratch_mu)
        PDEHIP_FAIL(E_VALUE, "Cahn-Hilliard rhs needs a scratch array for mu");
    return 0;
}

// cache of captured Euler-step graphs (see pdehip_euler_run)
constexpr int64_t kGraphSteps = 32;   // even: the ping-pong buffers are back in place after a block
constexpr int kGraphCache = 8;
struct GraphKey {
    pdehip_grid_t g;
    pdehip_rhs_t rhs;
    void *buf[8];   // every array the captured launches touch (Euler: the two ping-pong buffers; RK4: y and the work arrays)
    double dt;
    int kind;       // 0 Euler block, 1 RK4 block
};
struct GraphEntry {
    GraphKey key;
    hipGraphExec_t exec = nullptr;
    hipStream_t cap = nullptr;
    hipEvent_t ev = nullptr;
};
GraphEntry g_graphs[kGraphCache];
unsigned g_graph_next = 0;

// Replays a cached graph of `block` steps as often as it fits into `nsteps`, building it first when the run is long
// enough to pay for the build (milliseconds).  `capture(stream)` issues one block onto the capture stream.  The
// replays are ordered after the user's stream and the result is handed back to it (no host synchronisation).
// *done = steps taken through replays (0: nothing cached / built; the caller launches every step itself).
template <class Capture>
int replay_graph(const GraphKey &key, int64_t nsteps, int64_t block, int64_t build_from, void *stream, Capture &&capture, int64_t *done)
{
    *done = 0;
    if (nsteps < block) return 0;
    GraphEntry *entry = nullptr;
    for (auto &e : g_graphs)
        if (e.exec && memcmp(&e.key, &key, sizeof(key)) == 0) { entry = &e; break; }
    if (!entry && nsteps >= build_from) {
        GraphEntry &e = g_graphs[g_graph_next++ % kGraphCache];
        if (e.exec) { (void)hipGraphExecDestroy(e.exec); e.exec = nullptr; }
        if (!e.cap) PDEHIP_HIP(hipStreamCreateWithFlags(&e.cap, hipStreamNonBlocking));
        if (!e.ev) PDEHIP_HIP(hipEventCreateWithFlags(&e.ev, hipEventDisableTiming));
        hipGraph_t graph = nullptr;
        PDEHIP_HIP(hipStreamBeginCapture(e.cap, hipStreamCaptureModeThreadLocal));
        const int rc = capture((void *)e.cap);
        const hipError_t ce = hipStreamEndCapture(e.cap, &graph);
        if (rc == 0 && ce == hipSuccess && hipGraphInstantiate(&e.exec, graph, nullptr, nullptr, 0) == hipSuccess) {
            e.key = key;
            entry = &e;
        } else {
            e.exec = nullptr;
        }
        if (graph) (void)hipGraphDestroy(graph);
        if (rc) return rc;
    }
    if (entry) {
        hipStream_t user = as_stream(stream);
        PDEHIP_HIP(hipEventRecord(entry->ev, user));
        PDEHIP_HIP(hipStreamWaitEvent(entry->cap, entry->ev, 0));
        int64_t s = 0;
        for (; s + block <= nsteps; s += block) PDEHIP_HIP(hipGraphLaunch(entry->exec, entry->cap));
        PDEHIP_HIP(hipEventRecord(entry->ev, entry->cap));
        PDEHIP_HIP(hipStreamWaitEvent(user, entry->ev, 0));
        *done = s;
    }
    return 0;
}

bool graphs_enabled()
{
    static int use_graph = -1;
    if (use_graph < 0) { const char *e = getenv("PDEHIP_GRAPH"); use_graph = e ? atoi(e) : 1; }
    return use_graph != 0;
}

// One Runge-Kutta stage in ONE sweep: the slope k = dt*rhs(in) and, from the slope still in registers, the pointwise
// combination that follows it (the input of the next stage, or the new state).  Saves the separate lincomb /
// rk4_combine pass: a stage moves (1 + earlier slopes + 2 or 3) arrays instead of 2 + (earlier slopes + 3).
// *fused = false (nothing launched) when the sweep is not available: diffusion needs the vectorised kernel,
// Cahn-Hilliard the two-level kernel (pdehip_march2.inc).
// faces that depend on the time or on the field: their coefficient arrays for the time and the input `in` of THIS evaluation
// (pdehip_rhs_t::bc_program)
int refresh_bcs(const pdehip_rhs_t *rhs, double t, const void *in, void *stream)
{
    return rhs->bc_program ? pdehip_bcprog_run(rhs->bc_program, t, in, stream) : 0;
}

int rhs_stage(const pdehip_grid_t *g, const pdehip_rhs_t *rhs, void *in, void *k_out, double dt, const StageFuse &sf,
              void *stream, bool *fused)
{
    *fused = false;
    static int on = -1;   // PDEHIP_RK_FUSE=0: separate kernels (tuning / testing aid)
    if (on < 0) { const char *e = getenv("PDEHIP_RK_FUSE"); on = e ? atoi(e) : 1; }
    if (!on) return 0;
    if (rhs->kind == PDEHIP_RHS_CAHN_HILLIARD)   // two-level sweep (mu in registers) with the stage epilogue, where it covers grid and faces
        return cahn_hilliard_fused(g, in, k_out, rhs->param, dt, false, rhs->bc_c, rhs->bc_mu, stream, fused, 0, false, &sf);
    NGrid n;
    PDEHIP_TRY(norm_grid(g, &n));
    uintptr_t bits = (uintptr_t)sf.y | (uintptr_t)sf.out2;
    for (int m = 0; m < 5; m++) bits |= (uintptr_t)sf.k[m];
    if (bits % 16 != 0 || !laplace_can_fuse_bcs(n, in, k_out, nullptr)) return 0;
    *fused = true;
    return laplace_with_input_bcs(g, in, nullptr, k_out, LAP_STAGE, rhs->param, dt, 0, rhs->bc_c, stream, &sf);
}

}  // namespace

extern "C" {

static int rhs_scaled_at(const pdehip_grid_t *g, const pdehip_rhs_t *rhs, void *y_full, void *k_out_full, double dt, double t, void *stream);
int pdehip_rhs_scaled(const pdehip_grid_t *g, const pdehip_rhs_t *rhs, void *y_full, void *k_out_full,
                      double dt, void *stream)
{
    PDEHIP_TRY(check_rhs(rhs));
    return rhs_scaled_at(g, rhs, y_full, k_out_full, dt, rhs->t, stream);
}

static int rhs_scaled_at(const pdehip_grid_t *g, const pdehip_rhs_t *rhs, void *y_full, void *k_out_full, double dt, double t, void *stream)
{
    PDEHIP_TRY(refresh_bcs(rhs, t, y_full, stream));
    // numba/backend.py:501-517: BCs, then the stencil — here one kernel (BCs evaluated on the fly)
    if (rhs->kind == PDEHIP_RHS_DIFFUSION)   // dt * (D * lap)
        return laplace_with_input_bcs(g, y_full, nullptr, k_out_full, LAP_SCALED, rhs->param, dt, 0, rhs->bc_c, stream);
    // mu = c^3 - c - g*lap(c) with bc_c;  k = dt * lap(mu) with bc_mu — ONE sweep with mu in registers where the
    // two-level kernel covers grid and faces (16 instead of 32 B per cell), else two kernels through scratch_mu
    bool fused = false;
    PDEHIP_TRY(cahn_hilliard_fused(g, y_full, k_out_full, rhs->param, dt, false, rhs->bc_c, rhs->bc_mu, stream, &fused));
    if (fused) return 0;
    PDEHIP_TRY(laplace_with_input_bcs(g, y_full, nullptr, rhs->scratch_mu, LAP_CH_MU, 0, 0, rhs->param, rhs->bc_c, stream));
    return laplace_with_input_bcs(g, rhs->scratch_mu, nullptr, k_out_full, LAP_SCALED, 1.0, dt, 0, rhs->bc_mu, stream);
}

int pdehip_euler_run(const pdehip_grid_t *g, const pdehip_rhs_t *rhs, void *buf_a, void *buf_b, double dt,
                     int64_t nsteps, void **result, void *stream)
{
    PDEHIP_TRY(check_rhs(rhs));
    if (!buf_a || !buf_b || !result) PDEHIP_FAIL(E_VALUE, "euler_run: NULL pointer");
    if (nsteps < 0) PDEHIP_FAIL(E_VALUE, "euler_run: negative step count");
    void *cur = buf_a, *nxt = buf_b;
    bool ch_fused = true;
    const bool timed = rhs->bc_program != nullptr;   // faces that change with time: one step per sweep, refreshed before every step
    auto one_step = [&](void *c, void *n, void *st) -> int {
        if (rhs->kind == PDEHIP_RHS_DIFFUSION)
            // state + dt * (D * laplace(state))   euler.py:174 with diffusion.py:121 — ONE kernel per step
            return laplace_with_input_bcs(g, c, c, n, LAP_EULER, rhs->param, dt, 0, rhs->bc_c, st);
        if (ch_fused) {   // mu in registers: 16 instead of 40 B per cell and step
            bool fused = false;
            PDEHIP_TRY(cahn_hilliard_fused(g, c, n, rhs->param, dt, true, rhs->bc_c, rhs->bc_mu, st, &fused));
            if (fused) return 0;
            ch_fused = false;
        }
        PDEHIP_TRY(laplace_with_input_bcs(g, c, nullptr, rhs->scratch_mu, LAP_CH_MU, 0, 0, rhs->param, rhs->bc_c, st));
        return laplace_with_input_bcs(g, rhs->scratch_mu, c, n, LAP_EULER, 1.0, dt, 0, rhs->bc_mu, st);
    };
    // Two steps per sweep where the temporal-blocking kernel covers the grid and its BCs (the intermediate
    // level never touches HBM: 16 B per cell for two steps), else one step per sweep.
    bool two_ok = rhs->kind == PDEHIP_RHS_DIFFUSION && !timed;   // (the second level would need the faces at t + dt)
    // ... which a program of conditions that depend on time and position (not on the field) writes as a second coefficient set: two
    // steps per sweep with stand-in faces, then the two layers of cells next to those faces again with the true coefficients of both
    // levels (pdehip_shell.hip).  The same for faces given as arrays that nothing rewrites (conditions that depend on the position).
    // PDEHIP_TIMED_TWO_STEP=0: one step per sweep (A/B, tests).
    bool array_faces = false;
    for (int q = 0; q < 2 * g->ndim; q++) array_faces = array_faces || (rhs->bc_c[q].flags & PDEHIP_BCF_ARRAYS) != 0;
    const char *two_env = getenv("PDEHIP_TIMED_TWO_STEP");
    bool two_arr = rhs->kind == PDEHIP_RHS_DIFFUSION && (timed || array_faces) && !(timed && bcprog_reads(rhs->bc_program)) &&
                   !(two_env && two_env[0] == '0');
    // 2-D grids of a few MB: K steps per launch with the time levels in LDS (pdehip_tile2d.inc) — such grids are bound by
    // launch / cache latency per step, not by HBM.  PDEHIP_TILE2D=off disables it, PDEHIP_TILE2D=<k> caps K,
    // PDEHIP_TILE2D_CELLS=<n> moves the size limit (default 2^21 cells).
    static int tile_k = -1;
    static long tile_cells = 1L << 21;   // measured (profiles/r02_time_sizes.md): wins up to 1024^2, loses at 2048^2 (HBM-bound there: the register kernel moves fewer bytes)
    if (tile_k < 0) {
        const char *e = getenv("PDEHIP_TILE2D");
        tile_k = (e && !strcmp(e, "off")) ? 0 : (e && atoi(e) > 0 ? atoi(e) : 64);
        const char *c = getenv("PDEHIP_TILE2D_CELLS");
        if (c && atol(c) > 0) tile_cells = atol(c);
    }
    long ncells = 1;
    for (int q = 0; q < g->ndim; q++) ncells *= g->shape[q];
    bool tile_ok = g->ndim == 2 && tile_k > 0 && ncells <= tile_cells && !timed;
    auto advance = [&](void *c, void *n, void *st, int64_t left, int *took, int64_t step = 0) -> int {
        if (two_arr && left >= 2) {
            if (timed) PDEHIP_TRY(bcprog_run_pair(rhs->bc_program, rhs->t + (double)step * dt, rhs->t + (double)(step + 1) * dt, st));
            bool done = false;
            PDEHIP_TRY(euler2_timed_faces(g, c, n, rhs->param, dt, rhs->bc_c, rhs->bc_program, st, &done));
            if (done) { *took = 2; return 0; }
            two_arr = false;
        }
        if (timed) PDEHIP_TRY(refresh_bcs(rhs, rhs->t + (double)step * dt, c, st));   // _solvers.py:100: t = t_start + i * dt
        if (tile_ok) {
            int k = tile2d_max_steps(rhs->kind == PDEHIP_RHS_DIFFUSION ? 0 : 1);
            if (k > tile_k) k = tile_k;
            if (k > left) k = (int)left;
            bool done = false;
            PDEHIP_TRY(euler_multi_2d(g, rhs, c, n, dt, k, st, &done));
            if (done) { *took = k; return 0; }
            tile_ok = false;
        }
        if (two_ok && left >= 2) {
            bool done = false;
            PDEHIP_TRY(euler2_with_input_bcs(g, c, n, rhs->param, dt, rhs->bc_c, st, &done));
            if (done) { *took = 2; return 0; }
            two_ok = false;
        }
        *took = 1;
        return one_step(c, n, st);
    };
    // Launch-bound regime (small grids: a 512^2 step is ~3 us of GPU work, the host needs ~3.5 us per
    // launch): capture a block of steps into a hipGraph and replay it, so the inner loop costs one graph
    // launch per kGraphSteps steps instead of 1-2 kernel launches per step.  Building a graph costs
    // milliseconds, so graphs are cached (same grid, rhs, buffers and dt -> same graph) and only built
    // for long runs; a stepper that is called once per tracker interrupt re-uses its graph.
    int64_t s = 0;
    long cells = 1;
    for (int a = 0; a < g->ndim; a++) cells *= g->shape[a];
    if (graphs_enabled() && !timed && cells <= (1L << 22) && nsteps >= kGraphSteps) {
        GraphKey key;
        memset(&key, 0, sizeof(key));
        key.g = *g; key.rhs = *rhs; key.rhs.t = 0; key.buf[0] = buf_a; key.buf[1] = buf_b; key.dt = dt; key.kind = 0;
        PDEHIP_TRY(replay_graph(key, nsteps, kGraphSteps, 2048, stream, [&](void *cap) -> int {
            for (int64_t q = 0; q < kGraphSteps;) {
                int took = 0;
                PDEHIP_TRY(advance(cur, nxt, cap, kGraphSteps - q, &took));
                q += took;
                void *t = cur; cur = nxt; nxt = t;   // even number of swaps (16 double steps or 32 single steps): back in place
            }
            return 0;
        }, &s));
    }
    while (s < nsteps) {
        int took = 0;
        PDEHIP_TRY(advance(cur, nxt, stream, nsteps - s, &took, s));
        s += took;
        void *t = cur; cur = nxt; nxt = t;
    }
    *result = cur;
    return 0;
}

static int rk4_step_at(const pdehip_grid_t *g, const pdehip_rhs_t *rhs, void *y, void *const *w, double dt, double t, void *stream);
int pdehip_rk4_step(const pdehip_grid_t *g, const pdehip_rhs_t *rhs, void *y, void *const *w, double dt, void *stream)
{
    PDEHIP_TRY(check_rhs(rhs));
    if (!y || !w) PDEHIP_FAIL(E_VALUE, "rk4_step: NULL pointer");
    return rk4_step_at(g, rhs, y, w, dt, rhs->t, stream);
}

static int rk4_step_at(const pdehip_grid_t *g, const pdehip_rhs_t *rhs, void *y, void *const *w, double dt, double t, void *stream)
{
    void *k1 = w[0], *k2 = w[1], *k3 = w[2], *k4 = w[3], *tmp = w[4];
    const double half = 0.5, one = 1.0;
    const void *kk[1];
    // runge_kutta.py:52-61.  Fused form: every sweep also writes what the next one reads (k4's array serves as the
    // second stage input; k4 itself never leaves the registers) - 17 instead of 23 arrays moved per step.
    bool fused = false;
    StageFuse sf;
    memset(&sf, 0, sizeof(sf));
    sf.y = y; sf.c_new = half; sf.out2 = tmp;
    // stage times t, t + dt/2, t + dt/2, t + dt (runge_kutta.py:52-59): time-dependent faces are refreshed before each stage
    PDEHIP_TRY(refresh_bcs(rhs, t, y, stream));
    PDEHIP_TRY(rhs_stage(g, rhs, y, k1, dt, sf, stream, &fused));
    if (fused) {
        sf.out2 = k4;
        PDEHIP_TRY(refresh_bcs(rhs, t + 0.5 * dt, tmp, stream));
        PDEHIP_TRY(rhs_stage(g, rhs, tmp, k2, dt, sf, stream, &fused));
        sf.c_new = one; sf.out2 = tmp;
        PDEHIP_TRY(refresh_bcs(rhs, t + 0.5 * dt, k4, stream));   // (the same time, another input: conditions that read the field)
        if (fused) PDEHIP_TRY(rhs_stage(g, rhs, k4, k3, dt, sf, stream, &fused));
        sf.kind = 1; sf.k[0] = k1; sf.k[1] = k2; sf.k[2] = k3; sf.out2 = y;
        PDEHIP_TRY(refresh_bcs(rhs, t + dt, tmp, stream));
        if (!fused) PDEHIP_FAIL(E_RUNTIME, "internal: fused Runge-Kutta stage refused after the first one was taken");
        PDEHIP_TRY(rhs_stage(g, rhs, tmp, nullptr, dt, sf, stream, &fused));
        if (fused) return 0;
        // the last sweep writes the new state over the old one, which sweeps with overlapping tiles (row lengths / counts that
        // no tile divides, pdehip_march2.inc) cannot do: slope into the array of k4 (free since the third stage), then combine
        PDEHIP_TRY(rhs_scaled_at(g, rhs, tmp, k4, dt, t + dt, stream));
        return pdehip_rk4_combine(g, 1, y, k1, k2, k3, k4, stream);
    }
    PDEHIP_TRY(rhs_scaled_at(g, rhs, y, k1, dt, t, stream));
    kk[0] = k1; PDEHIP_TRY(pdehip_lincomb(g, 1, tmp, y, 1, &half, kk, stream));
    PDEHIP_TRY(rhs_scaled_at(g, rhs, tmp, k2, dt, t + 0.5 * dt, stream));
    kk[0] = k2; PDEHIP_TRY(pdehip_lincomb(g, 1, tmp, y, 1, &half, kk, stream));
    PDEHIP_TRY(rhs_scaled_at(g, rhs, tmp, k3, dt, t + 0.5 * dt, stream));
    kk[0] = k3; PDEHIP_TRY(pdehip_lincomb(g, 1, tmp, y, 1, &one, kk, stream));
    PDEHIP_TRY(rhs_scaled_at(g, rhs, tmp, k4, dt, t + dt, stream));
    return pdehip_rk4_combine(g, 1, y, k1, k2, k3, k4, stream);
}

int pdehip_ab2_step(const pdehip_grid_t *g, const pdehip_rhs_t *rhs, void *y_in, void *y_out, void *rate_cur,
                    const void *rate_prev, double dt, int *fused, void *stream)
{
    PDEHIP_TRY(check_rhs(rhs));
    if (!y_in || !y_out || !rate_cur || !rate_prev || !fused) PDEHIP_FAIL(E_VALUE, "ab2_step: NULL pointer");
    if (y_in == y_out || rate_cur == rate_prev) PDEHIP_FAIL(E_VALUE, "ab2_step: aliased arrays");
    *fused = 0;
    // adams_bashforth.py:40-47 in ONE sweep: the rate of y_in (stored for the next step) and the new state from it
    StageFuse sf;
    memset(&sf, 0, sizeof(sf));
    sf.kind = 3; sf.y = y_in; sf.k[0] = rate_prev; sf.c_new = dt; sf.out2 = y_out;
    bool done = false;
    PDEHIP_TRY(refresh_bcs(rhs, rhs->t, y_in, stream));
    PDEHIP_TRY(rhs_stage(g, rhs, y_in, rate_cur, 1.0, sf, stream, &done));
    *fused = done ? 1 : 0;
    return 0;
}

int pdehip_rk4_run(const pdehip_grid_t *g, const pdehip_rhs_t *rhs, void *y, void *const *w, double dt, int64_t nsteps,
                   void *stream)
{
    PDEHIP_TRY(check_rhs(rhs));
    if (!y || !w) PDEHIP_FAIL(E_VALUE, "rk4_run: NULL pointer");
    if (nsteps < 0) PDEHIP_FAIL(E_VALUE, "rk4_run: negative step count");
    // launch-bound regime (a 512^2 stage is a few us of GPU work): blocks of 8 steps (32 sweeps) as a cached hipGraph
    constexpr int64_t kBlock = 8;
    int64_t s = 0;
    long cells = 1;
    for (int a = 0; a < g->ndim; a++) cells *= g->shape[a];
    if (graphs_enabled() && !rhs->bc_program && cells <= (1L << 22) && nsteps >= kBlock) {
        GraphKey key;
        memset(&key, 0, sizeof(key));
        key.g = *g; key.rhs = *rhs; key.rhs.t = 0; key.dt = dt; key.kind = 1;
        key.buf[0] = y;
        for (int q = 0; q < 5; q++) key.buf[1 + q] = w[q];
        PDEHIP_TRY(replay_graph(key, nsteps, kBlock, 512, stream, [&](void *cap) -> int {
            for (int64_t q = 0; q < kBlock; q++) PDEHIP_TRY(pdehip_rk4_step(g, rhs, y, w, dt, cap));
            return 0;
        }, &s));
    }
    for (; s < nsteps; s++) PDEHIP_TRY(rk4_step_at(g, rhs, y, w, dt, rhs->t + (double)s * dt, stream));   // _solvers.py:100
    return 0;
}

int pdehip_rkf45_attempt(const pdehip_grid_t *g, const pdehip_rhs_t *rhs, void *y, void *ynew, void *const *w,
                         double dt, double *err_dev, void *stream)
{
    PDEHIP_TRY(check_rhs(rhs));
    if (!y || !ynew || !w || !err_dev) PDEHIP_FAIL(E_VALUE, "rkf45_attempt: NULL pointer");
    void *tmp = w[6];
    const void *k[6] = {w[0], w[1], w[2], w[3], w[4], w[5]};
    const double t0 = rhs->t;   // the stages are evaluated at t0 + a_s * dt (runge_kutta.py:135-145)
    // runge_kutta.py:135-145.  Fused form: stages 1-5 also write the input of the next stage (alternating between
    // tmp and ynew), the sixth computes the new state and the error norm from k6 in registers - 36 instead of 45
    // arrays moved per attempt.
    {
        const double *tab[5] = {B2, B3, B4, B5, B6};
        void *t_in = y, *t_out = tmp;
        bool fused = true;
        for (int s = 0; s < 5 && fused; s++) {
            StageFuse sf;
            memset(&sf, 0, sizeof(sf));
            sf.y = y; sf.out2 = t_out;
            for (int m = 0; m < s; m++) { sf.k[m] = k[m]; sf.c[m] = tab[s][m]; }
            sf.c_new = tab[s][s];
            PDEHIP_TRY(refresh_bcs(rhs, t0 + A45[s] * dt, t_in, stream));
            PDEHIP_TRY(rhs_stage(g, rhs, t_in, w[s], dt, sf, stream, &fused));
            if (!fused && s > 0) PDEHIP_FAIL(E_RUNTIME, "internal: fused Runge-Kutta stage refused after the first one was taken");
            t_in = t_out;
            t_out = (t_out == tmp) ? ynew : tmp;
        }
        if (fused) {   // five stages done: the last input is in tmp, ynew is free again; k6 stays in registers
            StageFuse sf;
            memset(&sf, 0, sizeof(sf));
            sf.kind = 2; sf.y = y; sf.out2 = ynew; sf.err = err_dev;
            sf.k[0] = k[0]; sf.k[1] = k[2]; sf.k[2] = k[3]; sf.k[3] = k[4];
            PDEHIP_HIP(hipMemsetAsync(err_dev, 0, sizeof(double), as_stream(stream)));
            PDEHIP_TRY(refresh_bcs(rhs, t0 + A45[5] * dt, t_in, stream));
            PDEHIP_TRY(rhs_stage(g, rhs, t_in, nullptr, dt, sf, stream, &fused));
            if (!fused) PDEHIP_FAIL(E_RUNTIME, "internal: fused Runge-Kutta stage refused after the first one was taken");
            return 0;
        }
    }
    PDEHIP_TRY(rhs_scaled_at(g, rhs, y, w[0], dt, t0, stream));
    PDEHIP_TRY(pdehip_lincomb(g, 1, tmp, y, 1, B2, k, stream));
    PDEHIP_TRY(rhs_scaled_at(g, rhs, tmp, w[1], dt, t0 + A45[1] * dt, stream));
    PDEHIP_TRY(pdehip_lincomb(g, 1, tmp, y, 2, B3, k, stream));
    PDEHIP_TRY(rhs_scaled_at(g, rhs, tmp, w[2], dt, t0 + A45[2] * dt, stream));
    PDEHIP_TRY(pdehip_lincomb(g, 1, tmp, y, 3, B4, k, stream));
    PDEHIP_TRY(rhs_scaled_at(g, rhs, tmp, w[3], dt, t0 + A45[3] * dt, stream));
    PDEHIP_TRY(pdehip_lincomb(g, 1, tmp, y, 4, B5, k, stream));
    PDEHIP_TRY(rhs_scaled_at(g, rhs, tmp, w[4], dt, t0 + A45[4] * dt, stream));
    PDEHIP_TRY(pdehip_lincomb(g, 1, tmp, y, 5, B6, k, stream));
    PDEHIP_TRY(rhs_scaled_at(g, rhs, tmp, w[5], dt, t0 + A45[5] * dt, stream));
    // runge_kutta.py:147-150
    return pdehip_rkf45_combine(g, 1, y, ynew, k, err_dev, stream);
}

}  // extern "C"
