// pdehip_rk_loops.h — Runge-Kutta step sequences and the adaptive loop, written ONCE against an evaluator policy `Eval`.
//
// The reference jit-compiles these loops around whatever right-hand side the PDE provides: RK4 pde/solvers/runge_kutta.py:29-66,
// RKF45 :68-156, the adaptive loop pde/backends/numba/_solvers.py:199-319 with the controller pde/solvers/base.py:533-594.
// Here the right-hand side of an expression PDE is a list of kernel passes (pdehip_jit.hip); this header holds the control flow
// around it, so that a whole run is ONE C call: no Python per step or per stage, 8 bytes to the host per adaptive attempt.
// Two instantiations: csrc/pdehip_jit.hip (JitEval: run-time built HIP kernels — the product) and tests/shim/pdehip_shim_comm.cpp
// (host memory, gcc-built epilogues, oracle kernels — TESTS ONLY), so the CPU test-suite executes exactly this sequence.
// (The slab-parallel twins for the built-in right-hand sides, with their halo exchanges, are in pdehip_slab_loops.h.)
//
// Eval provides:
//   int slope(void *in, void *k_out, double dt, double t, const StageFuse *sf, bool *fused, void *st)
//        k_out = dt * F(in; t); if it can, also the combination `sf` in the same sweep (*fused = true) — then k_out is written
//        only for sf->kind 0 / 3;   int lincomb / rk4_combine / rkf45_combine(...): the pointwise kernels of include/pdehip.h;
//   int zero(void *, size_t, void *st);   int read_scalar(double *host, const double *dev, void *st);
//   int reduce_error(double *err_dev, void *st)   (in-place MAX over all ranks, NaN wins; a no-op on one device)
//   int fail_runtime(const char *fmt, double value)   (sets the error message, returns the RuntimeError status)
#pragma once

#include <cmath>
#include <cstring>

#include "pdehip_slab_loops.h"   // StageFuse, slab::rkf45_row, slab::adjust_dt, SLAB_TRY

namespace pdehip {
namespace rk {

// one classical RK4 step in place on y; w = k1..k4, tmp (runge_kutta.py:52-61; same stage sequence as pdehip_rk4_step: the array
// of k4 doubles as the second stage-input buffer, a fused last sweep never stores k4)
template <class Eval>
int rk4_step(Eval &ev, void *y, void *const *w, double dt, double t, void *st)
{
    void *k1 = w[0], *k2 = w[1], *k3 = w[2], *k4 = w[3], *tmp = w[4];
    const double half = 0.5, one = 1.0;
    const void *kk[1];
    bool fused = false;
    StageFuse sf;
    memset(&sf, 0, sizeof(sf));
    sf.y = y; sf.c_new = half; sf.out2 = tmp;
    SLAB_TRY(ev.slope(y, k1, dt, t, &sf, &fused, st));
    if (!fused) { kk[0] = k1; SLAB_TRY(ev.lincomb(tmp, y, 1, &half, kk, st)); }
    sf.out2 = k4;
    SLAB_TRY(ev.slope(tmp, k2, dt, t + 0.5 * dt, &sf, &fused, st));
    if (!fused) { kk[0] = k2; SLAB_TRY(ev.lincomb(k4, y, 1, &half, kk, st)); }
    sf.c_new = one; sf.out2 = tmp;
    SLAB_TRY(ev.slope(k4, k3, dt, t + 0.5 * dt, &sf, &fused, st));
    if (!fused) { kk[0] = k3; SLAB_TRY(ev.lincomb(tmp, y, 1, &one, kk, st)); }
    sf.kind = 1; sf.k[0] = k1; sf.k[1] = k2; sf.k[2] = k3; sf.c_new = 0.0; sf.out2 = y;
    SLAB_TRY(ev.slope(tmp, k4, dt, t + dt, &sf, &fused, st));
    if (!fused) SLAB_TRY(ev.rk4_combine(y, k1, k2, k3, k4, st));
    return 0;
}

// one RKF45 attempt: ynew and *err_dev from y (runge_kutta.py:135-153); w = k1..k6, tmp; stage inputs alternate between tmp and
// ynew (free until the last sweep), like pdehip_rkf45_attempt
template <class Eval>
int rkf45_attempt(Eval &ev, void *y, void *ynew, void *const *w, double dt, double t, double *err_dev, void *st)
{
    static const double A[6] = {0.0, 1.0 / 4, 3.0 / 8, 12.0 / 13, 1.0, 1.0 / 2};   // runge_kutta.py:92-98
    void *tmp = w[6];
    void *t_in = y, *t_out = tmp;
    const void *k[6] = {w[0], w[1], w[2], w[3], w[4], w[5]};
    bool fused = false;
    for (int s = 0; s < 5; s++) {
        StageFuse sf;
        memset(&sf, 0, sizeof(sf));
        sf.y = y; sf.out2 = t_out;
        const double *row = slab::rkf45_row(s);
        for (int m = 0; m < s; m++) { sf.k[m] = w[m]; sf.c[m] = row[m]; }
        sf.c_new = row[s];
        SLAB_TRY(ev.slope(t_in, w[s], dt, t + A[s] * dt, &sf, &fused, st));
        if (!fused) SLAB_TRY(ev.lincomb(t_out, y, s + 1, row, k, st));
        t_in = t_out;
        t_out = (t_out == tmp) ? ynew : tmp;
    }
    StageFuse sf;
    memset(&sf, 0, sizeof(sf));
    sf.kind = 2; sf.y = y; sf.out2 = ynew; sf.err = err_dev;
    sf.k[0] = w[0]; sf.k[1] = w[2]; sf.k[2] = w[3]; sf.k[3] = w[4];
    SLAB_TRY(ev.zero(err_dev, sizeof(double), st));
    SLAB_TRY(ev.slope(t_in, w[5], dt, t + A[5] * dt, &sf, &fused, st));
    if (!fused) SLAB_TRY(ev.rkf45_combine(y, ynew, k, err_dev, st));
    return 0;
}

// the adaptive loop of pde/backends/numba/_solvers.py:249-281; accepted attempts swap
// the roles of y / ynew; *result names the array holding the final state
template <class Eval>
int rkf45_run(Eval &ev, void *y, void *ynew, void *const *w, double *err_dev, pdehip_adaptive_t *a, void **result, void *st)
{
    double dt_opt = a->dt, t = a->t_start;
    void *cur = y, *nxt = ynew;
    while (true) {
        const double dt_step = std::fmax(std::fmin(dt_opt, a->t_end - t), a->dt_min);
        SLAB_TRY(rkf45_attempt(ev, cur, nxt, w, dt_step, t, err_dev, st));
        SLAB_TRY(ev.reduce_error(err_dev, st));   // MAX over all ranks (pde/backends/base.py:678-712); nothing on one device
        double err = 0;
        SLAB_TRY(ev.read_scalar(&err, err_dev, st));
        const double error_rel = err / a->tolerance;
        a->attempts++;
        if (error_rel <= 1) {   // accept (false for NaN)
            a->steps++;
            t += dt_step;
            void *tmp = cur; cur = nxt; nxt = tmp;
            a->stat_min = a->stat_count ? std::fmin(a->stat_min, dt_step) : dt_step;
            a->stat_max = a->stat_count ? std::fmax(a->stat_max, dt_step) : dt_step;
            const double delta = dt_step - a->stat_mean;
            a->stat_count++;
            a->stat_mean += delta / (double)a->stat_count;
            a->stat_m2 += delta * (dt_step - a->stat_mean);
        }
        if (t < a->t_end) {
            double d = dt_step;
            const int bad = slab::adjust_dt(&d, error_rel, a->dt_min, a->dt_max);
            if (bad) {
                a->dt = dt_opt; a->t_last = t; *result = cur;
                return bad == 1 ? ev.fail_runtime("Encountered NaN even though dt < %g", a->dt_min) : ev.fail_runtime("Time step below %g", a->dt_min);
            }
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

// The reference's ADAPTIVE EULER loop, pde/backends/numba/_solvers.py:322-466 (loop :374-433; numpy twin pde/solvers/euler.py:222-280).
// It is NOT the generic "one step vs two half steps" estimate of pde/solvers/base.py:393-425: the rate of the current state is CARRIED
// from attempt to attempt,
//     rate       = rhs(state, t_start)                                   once per call (:373)
//     step_large = state + dt * rate,   step_small = state + dt/2 * rate               (:381-383)
//     step_small += dt/2 * rhs(step_small, t + dt/2);  error = max |step_large - step_small|   (:387-394)
//     accept:  rate = rhs(step_small, t)   - evaluated at the time BEFORE `t += dt` (:402-407) -,  state = step_small
//     reject:  the rate is kept, only dt changes
// so an accepted attempt costs two right-hand sides, a rejected one a single one, and with time-dependent conditions / explicit
// time in the equation the carried rate belongs to the OLD time.  Here the rate of an accepted state is evaluated lazily at the
// start of the next attempt (its time is remembered; the rate after the last step of a call is never needed) in a sweep that also
// writes the half step (StageFuse kind 0), and the second half step, the double-step state and the error norm come out of ONE
// sweep (kind 4): 7 array passes per accepted step where the sequence above, statement by statement, moves 14.
// w = rate, step_half, k (slope scratch of unfused sweeps); y / ynew swap on acceptance; *result names the final state.
// Eval additionally provides  int euler_adaptive_combine(y, rate, dt, half, k, out, err_dev, st)  (the pointwise form of kind 4).
template <class Eval>
int euler_adaptive_run(Eval &ev, void *y, void *ynew, void *const *w, double *err_dev, pdehip_adaptive_t *a, void **result, void *st)
{
    void *rate = w[0], *half = w[1], *kmid = w[2];
    double dt_opt = a->dt, t = a->t_start, t_rate = a->t_start;
    void *cur = y, *nxt = ynew;
    bool have_rate = false, fused = false;
    const double one = 1.0;
    while (true) {
        const double dt_step = std::fmax(std::fmin(dt_opt, a->t_end - t), a->dt_min);
        const double h = 0.5 * dt_step;
        const void *kk[1] = {rate};
        if (!have_rate) {
            // rate of the current state at the time it was accepted from + the first half step with it
            StageFuse sf;
            memset(&sf, 0, sizeof(sf));
            sf.y = cur; sf.c_new = h; sf.out2 = half;
            SLAB_TRY(ev.slope(cur, rate, one, t_rate, &sf, &fused, st));
            if (!fused) SLAB_TRY(ev.lincomb(half, cur, 1, &h, kk, st));
            have_rate = true;
        } else {
            SLAB_TRY(ev.lincomb(half, cur, 1, &h, kk, st));   // after a rejection: same rate, smaller step
        }
        StageFuse sf;
        memset(&sf, 0, sizeof(sf));
        sf.kind = 4; sf.y = cur; sf.k[0] = rate; sf.c[0] = dt_step; sf.k[1] = half; sf.out2 = nxt; sf.err = err_dev;
        SLAB_TRY(ev.zero(err_dev, sizeof(double), st));
        SLAB_TRY(ev.slope(half, kmid, h, t + h, &sf, &fused, st));
        if (!fused) SLAB_TRY(ev.euler_adaptive_combine(cur, rate, dt_step, half, kmid, nxt, err_dev, st));
        SLAB_TRY(ev.reduce_error(err_dev, st));
        double err = 0;
        SLAB_TRY(ev.read_scalar(&err, err_dev, st));
        const double error_rel = err / a->tolerance;
        a->attempts++;
        if (error_rel <= 1) {   // accept (false for NaN)
            a->steps++;
            t_rate = t;         // the reference evaluates the new rate BEFORE advancing the time (_solvers.py:402-407)
            t += dt_step;
            void *tmp = cur; cur = nxt; nxt = tmp;
            have_rate = false;
            a->stat_min = a->stat_count ? std::fmin(a->stat_min, dt_step) : dt_step;
            a->stat_max = a->stat_count ? std::fmax(a->stat_max, dt_step) : dt_step;
            const double delta = dt_step - a->stat_mean;
            a->stat_count++;
            a->stat_mean += delta / (double)a->stat_count;
            a->stat_m2 += delta * (dt_step - a->stat_mean);
        }
        if (t < a->t_end) {
            double d = dt_step;
            const int bad = slab::adjust_dt(&d, error_rel, a->dt_min, a->dt_max);
            if (bad) {
                a->dt = dt_opt; a->t_last = t; *result = cur;
                return bad == 1 ? ev.fail_runtime("Encountered NaN even though dt < %g", a->dt_min) : ev.fail_runtime("Time step below %g", a->dt_min);
            }
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

}  // namespace rk

namespace slab {

// The evaluator of the loops above for an axis-0 slab (or the whole grid on one device: lower = upper = -1, no communicator):
// slab::rhs_sweep exchanges the halo of its input, sweeps, and applies the combination itself - inside the sweep where
// `flags` allow, with the pointwise kernels otherwise - so every slope reports "fused".
template <class Ops>
struct Eval {
    Ops &ops;
    const pdehip_grid_t *g;
    const Geo &q;
    const pdehip_rhs_t *rhs;
    int lower, upper, flags;

    int slope(void *in, void *k_out, double dt, double t, const StageFuse *sf, bool *fused, void *st)
    {
        *fused = true;
        return rhs_sweep(ops, g, q, rhs, lower, upper, flags, in, k_out, dt, false, sf, st, t);
    }
    int lincomb(void *out, const void *y, int n, const double *c, const void *const *k, void *st) { return ops.lincomb(g, out, y, n, c, k, st); }
    int zero(void *p, size_t bytes, void *st) { return ops.zero(p, bytes, st); }
    int reduce_error(double *err_dev, void *st) { return ops.allreduce_max(err_dev, st); }
    int read_scalar(double *host, const double *dev, void *st) { return ops.read_scalar(host, dev, st); }
    int fail_runtime(const char *fmt, double v) { return ops.fail_runtime(fmt, v); }
    int euler_adaptive_combine(const void *, const void *, double, const void *, const void *, void *, double *, void *)
    {
        return ops.fail("internal: slab sweeps combine themselves");
    }
};

// adaptive Euler on a slab: rk::euler_adaptive_run with the exchanges of slab::rhs_sweep and the MAX all-reduce of the error
// (w = rate, step_half, slope scratch; y, ynew and step_half serve as sweep inputs: spare layers)
template <class Ops>
int euler_adaptive_run(Ops &ops, const pdehip_grid_t *g, const Geo &q, const pdehip_rhs_t *rhs, int lower, int upper, int flags, void *y,
                       void *ynew, void *const *w, double *err_dev, pdehip_adaptive_t *a, void **result, void *st)
{
    Eval<Ops> ev{ops, g, q, rhs, lower, upper, flags};
    return rk::euler_adaptive_run(ev, y, ynew, w, err_dev, a, result, st);
}

}  // namespace slab
}  // namespace pdehip
