/*
 * pdehip_shim.c — HOST implementation of the C ABI of include/pdehip.h.  TESTS ONLY.
 *
 * ******************************************************************************
 * TEST INFRASTRUCTURE.  This library exists so that the Python host side of the hip backend
 * (py-pde_amd/pde_hip: plugin class, BC conversion, stepper loops, solver.info bookkeeping) can
 * be driven END TO END through the REAL py-pde in a container without a GPU.  It is built into
 * tests/shim/_build/ by tests/shim/build.py and loaded only by tests (tests/shimlib.py sets
 * PDEHIP_LIB before the package binds its library).  The product never builds, ships, finds or
 * loads it: py-pde_amd/pde_hip/_lib.py resolves py-pde_amd/lib/libpdehip.so and raises when
 * that library or a HIP device is missing.
 * ******************************************************************************
 *
 * "Device" memory is host memory, streams and events are dummies, every compute entry point is
 * the CPU oracle (oracle/pde_oracle.c, compiled into this file) behind the pdehip_* name.  The
 * full layout of this library is the reference's compact layout (pitch N+2), reported through
 * pdehip_layout() like the device library reports its padded one — the host side must not care.
 *
 * Entry points that exist only on the device (two steps per sweep, fused Cahn-Hilliard sweep,
 * fused Runge-Kutta stages) report "not covered" (*done = 0) by default — the host side then
 * takes its unfused branch — or, with PDEHIP_SHIM_FUSED=1, are composed of oracle calls so that
 * the fused branches of the host code run as well (the device versions are bit-identical to the
 * composition by design; tests/test_hip_*.py check that on the GPU).
 *
 * Run-time specialised right-hand sides (pdehip_jit_*): the epilogue body is plain C, so it is
 * compiled with gcc into a small shared object and applied to laplace / gradient_squared
 * arrays computed by the oracle.
 *
 * Slab-parallel layer (pdehip_comm_*, pdehip_slab_*): the communicator is a directory of files
 * (one mailbox per rank and message), so several processes — the ranks of a torch.distributed
 * gloo job — can run the SAME slab loops as pdehip_comm.hip on the CPU (tests/test_distributed_*).
 */
#define _GNU_SOURCE
#include "../../oracle/pde_oracle.c"

#include <dlfcn.h>
#include <errno.h>
#include <fcntl.h>
#include <stdarg.h>
#include <stdio.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

enum { E_OK = 0, E_VALUE = 1, E_NOTIMPL = 2, E_RUNTIME = 3 };

static __thread char g_err[512] = "";
static int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}
/* error channel for the C++ half of the shim (pdehip_shim_comm.cpp) */
int shim_set_error(int code, const char *msg) { return fail(code, "%s", msg); }

#define TRY(expr)                                                                       \
    do {                                                                                \
        int _rc = (expr);                                                               \
        if (_rc) return _rc > 99 ? _rc : fail(_rc == 2 ? E_NOTIMPL : (_rc == 1 ? E_VALUE : E_RUNTIME), \
                                              "shim: %s failed with code %d", #expr, _rc); \
    } while (0)

static int fused_enabled(void)
{
    const char *e = getenv("PDEHIP_SHIM_FUSED");
    return e && e[0] == '1';
}

/* ---- runtime -------------------------------------------------------------------------- */
const char *pdehip_last_error(void) { return g_err; }
int pdehip_abi_version(void) { return PDEHIP_ABI_VERSION; }
static int g_shim_fastmath = 0;   /* recorded only: the oracle has one arithmetic */
int pdehip_set_fastmath(int on) { g_shim_fastmath = on ? 1 : 0; return 0; }
int pdehip_get_fastmath(int *on) { *on = g_shim_fastmath; return 0; }
const char *pdehip_last_kernel_name(void) { return "host shim (tests only): the CPU oracle stands in for every kernel"; }
int pdehip_device_count(int *count)
{
    const char *e = getenv("PDEHIP_SHIM_DEVICES");
    *count = e ? atoi(e) : 1;
    return 0;
}
static int g_device = 0;
int pdehip_set_device(int device)
{
    int n;
    pdehip_device_count(&n);
    if (device < 0 || device >= n) return fail(E_RUNTIME + 100, "hipSetDevice(%d) failed: invalid device ordinal", device);
    g_device = device;
    return 0;
}
int pdehip_get_device(int *device) { *device = g_device; return 0; }
int pdehip_device_name(char *buf, size_t len)
{
    snprintf(buf, len, "host shim (tests only), device %d", g_device);
    return 0;
}
int pdehip_malloc(void **ptr, size_t bytes)
{
    *ptr = calloc(1, bytes ? bytes : 1);
    return *ptr ? 0 : fail(E_RUNTIME, "shim: out of memory (%zu bytes)", bytes);
}
int pdehip_free(void *ptr) { free(ptr); return 0; }
int pdehip_memset(void *ptr, int value, size_t bytes, void *stream) { (void)stream; memset(ptr, value, bytes); return 0; }
int pdehip_memcpy_h2d(void *dst, const void *src, size_t bytes, void *stream) { (void)stream; memmove(dst, src, bytes); return 0; }
int pdehip_memcpy_d2h(void *dst, const void *src, size_t bytes, void *stream) { (void)stream; memmove(dst, src, bytes); return 0; }
int pdehip_memcpy_d2d(void *dst, const void *src, size_t bytes, void *stream) { (void)stream; memmove(dst, src, bytes); return 0; }
int pdehip_copy_nt(void *dst, const void *src, size_t bytes, void *stream) { (void)stream; memmove(dst, src, bytes); return 0; }
int pdehip_host_alloc(void **ptr, size_t bytes) { return pdehip_malloc(ptr, bytes); }
int pdehip_host_free(void *ptr) { free(ptr); return 0; }
int pdehip_stream_create(void **stream) { *stream = malloc(8); return 0; }
int pdehip_stream_destroy(void *stream) { free(stream); return 0; }
int pdehip_stream_synchronize(void *stream) { (void)stream; return 0; }
int pdehip_stream_wait_event(void *stream, void *event) { (void)stream; (void)event; return 0; }
int pdehip_event_create(void **event) { *event = calloc(1, sizeof(struct timespec)); return 0; }
int pdehip_event_destroy(void *event) { free(event); return 0; }
int pdehip_event_record(void *event, void *stream) { (void)stream; clock_gettime(CLOCK_MONOTONIC, (struct timespec *)event); return 0; }
int pdehip_event_synchronize(void *event) { (void)event; return 0; }
int pdehip_event_elapsed_ms(void *start, void *stop, float *ms)
{
    const struct timespec *a = start, *b = stop;
    *ms = (float)((b->tv_sec - a->tv_sec) * 1e3 + (b->tv_nsec - a->tv_nsec) * 1e-6);
    return 0;
}

int pdehip_layout(const pdehip_grid_t *g, int64_t *out8)
{
    ngrid_t n;
    if (!g || g->ndim < 1 || g->ndim > 3) return fail(E_NOTIMPL, "unsupported number of axes");
    if (g->dtype != PDEHIP_F64 && g->dtype != PDEHIP_F32) return fail(E_NOTIMPL, "unsupported dtype code %d", g->dtype);
    for (int a = 0; a < g->ndim; a++)
        if (g->shape[a] < 1) return fail(E_VALUE, "grid shape must be positive (axis %d: %ld)", a, (long)g->shape[a]);
    if (norm_grid(g, &n)) return fail(E_VALUE, "bad grid");
    out8[0] = n.p[0]; out8[1] = n.p[1]; out8[2] = n.pc; out8[3] = n.off; out8[4] = 1;
    out8[5] = n.pc; out8[6] = 0; out8[7] = n.p[3 - n.ndim];
    return 0;
}

static size_t full_bytes(const pdehip_grid_t *g, int ncomp)
{
    ngrid_t n;
    norm_grid(g, &n);
    return (size_t)ncomp * (size_t)n.pc * (g->dtype == PDEHIP_F64 ? 8 : 4);
}

static int check_grid(const pdehip_grid_t *g)
{
    int64_t lay[8];
    return pdehip_layout(g, lay);
}
#define GRID(g) do { int _rc = check_grid(g); if (_rc) return _rc; } while (0)

/* ---- layout conversion ------------------------------------------------------------------ */
int pdehip_valid_to_full(const pdehip_grid_t *g, int ncomp, const void *valid, void *full, void *stream)
{ (void)stream; GRID(g); TRY(oracle_valid_to_full(g, ncomp, valid, full)); return 0; }
int pdehip_full_to_valid(const pdehip_grid_t *g, int ncomp, const void *full, void *valid, void *stream)
{ (void)stream; GRID(g); TRY(oracle_full_to_valid(g, ncomp, full, valid)); return 0; }
/* strided host view <-> full layout: gather / scatter the rows into a contiguous valid image, then the oracle's copy */
static int transfer_valid(const pdehip_grid_t *g, int ncomp, char *host, const int64_t *hs, void *full, int upload)
{
    GRID(g);
    if (!host || !hs || !full) return fail(E_VALUE, "transfer of valid data: NULL pointer");
    if (ncomp < 1) return fail(E_VALUE, "number of components must be positive (%d)", ncomp);
    const long es = g->dtype == PDEHIP_F64 ? 8 : 4;
    if (hs[3] != es) return fail(E_VALUE, "host array must be contiguous along the fastest axis (stride %ld, element %ld bytes)", (long)hs[3], es);
    long n[3] = {1, 1, 1};
    int64_t st[3] = {0, 0, 0};
    for (int a = 0; a < g->ndim; a++) { n[3 - g->ndim + a] = g->shape[a]; st[3 - g->ndim + a] = hs[1 + 3 - g->ndim + a]; }
    const size_t row = (size_t)(n[2] * es);
    char *valid = malloc((size_t)ncomp * n[0] * n[1] * row + 16);
    if (!valid) return fail(E_RUNTIME, "out of memory");
    if (!upload) TRY(oracle_full_to_valid(g, ncomp, full, valid));
    char *v = valid;
    for (long c = 0; c < ncomp; c++)
        for (long i = 0; i < n[0]; i++)
            for (long j = 0; j < n[1]; j++, v += row) {
                char *h = host + c * hs[0] + i * st[0] + j * st[1];
                if (upload) memcpy(v, h, row); else memcpy(h, v, row);
            }
    if (upload) TRY(oracle_valid_to_full(g, ncomp, valid, full));
    free(valid);
    return 0;
}
int pdehip_upload_valid(const pdehip_grid_t *g, int ncomp, const void *host, const int64_t *host_strides, void *full, void *stream)
{ (void)stream; return transfer_valid(g, ncomp, (char *)host, host_strides, full, 1); }
int pdehip_download_valid(const pdehip_grid_t *g, int ncomp, const void *full, void *host, const int64_t *host_strides, void *stream)
{ (void)stream; return transfer_valid(g, ncomp, (char *)host, host_strides, (void *)full, 0); }
int pdehip_hostfull_to_full(const pdehip_grid_t *g, int ncomp, const void *hostfull, void *full, void *stream)
{ (void)stream; GRID(g); memmove(full, hostfull, full_bytes(g, ncomp)); return 0; }
int pdehip_full_to_hostfull(const pdehip_grid_t *g, int ncomp, const void *full, void *hostfull, void *stream)
{ (void)stream; GRID(g); memmove(hostfull, full, full_bytes(g, ncomp)); return 0; }

/* ---- ghost cells and operators ---------------------------------------------------------- */
int pdehip_set_ghost_cells(const pdehip_grid_t *g, int ncomp, const pdehip_bc_face_t *faces, void *data_full, void *stream)
{ (void)stream; GRID(g); TRY(oracle_set_ghost_cells(g, ncomp, faces, data_full)); return 0; }
int pdehip_laplace(const pdehip_grid_t *g, const void *in_full, void *out, int out_layout, void *stream)
{ (void)stream; GRID(g); TRY(oracle_laplace(g, in_full, out, out_layout)); return 0; }
int oracle_laplace_spectral(const pdehip_grid_t *g, const void *in_full, void *out, int out_layout);
int pdehip_laplace_spectral(const pdehip_grid_t *g, const void *in_full, void *out, int out_layout, void *stream)
{
    (void)stream; GRID(g);
    if (g->ndim > 2) return fail(E_NOTIMPL, "Spectral Laplace operator not implemented for %d dimensions", g->ndim);
    TRY(oracle_laplace_spectral(g, in_full, out, out_layout));
    return 0;
}
int pdehip_gradient(const pdehip_grid_t *g, int method, const void *in_full, void *out, int out_layout, void *stream)
{ (void)stream; GRID(g); TRY(oracle_gradient(g, method, in_full, out, out_layout)); return 0; }
int pdehip_divergence(const pdehip_grid_t *g, int method, const void *in_full, void *out, int out_layout, void *stream)
{ (void)stream; GRID(g); TRY(oracle_divergence(g, method, in_full, out, out_layout)); return 0; }
int pdehip_gradient_squared(const pdehip_grid_t *g, int central, const void *in_full, void *out, int out_layout, void *stream)
{ (void)stream; GRID(g); TRY(oracle_gradient_squared(g, central, in_full, out, out_layout)); return 0; }
int pdehip_axis_derivative(const pdehip_grid_t *g, int axis, int order, int method, const void *in_full, void *out,
                           int out_layout, void *stream)
{
    (void)stream; GRID(g);
    if (axis < 0 || axis >= g->ndim) return fail(E_VALUE, "axis %d out of range", axis);
    if (order != 1 && order != 2) return fail(E_VALUE, "derivative order must be 1 or 2");
    TRY(oracle_axis_derivative(g, axis, order, method, in_full, out, out_layout));
    return 0;
}
#ifdef ORACLE_HAS_LAPLACE9
int pdehip_laplace9(const pdehip_grid_t *g, const int *periodic2, double corner_weight, void *in_full, void *out,
                    int out_layout, void *stream)
{
    (void)stream; GRID(g);
    if (g->ndim != 2) return fail(E_VALUE, "the 9-point stencil needs a 2-D grid");
    TRY(oracle_laplace9(g, periodic2, corner_weight, in_full, out, out_layout));
    return 0;
}
#endif
int pdehip_laplace_scaled(const pdehip_grid_t *g, const void *in_full, void *out_full, double s1, double s2, void *stream)
{ (void)stream; GRID(g); TRY(oracle_laplace_scaled(g, in_full, out_full, s1, s2)); return 0; }
int pdehip_laplace_euler(const pdehip_grid_t *g, const void *in_full, const void *y_full, void *out_full, double s1, double s2,
                         void *stream)
{ (void)stream; GRID(g); TRY(oracle_laplace_euler(g, in_full, y_full, out_full, s1, s2)); return 0; }
int pdehip_cahn_hilliard_mu(const pdehip_grid_t *g, const void *c_full, void *mu_full, double gamma, void *stream)
{ (void)stream; GRID(g); TRY(oracle_cahn_hilliard_mu(g, c_full, mu_full, gamma)); return 0; }

/* ---- pointwise helpers -------------------------------------------------------------------- */
int pdehip_lincomb(const pdehip_grid_t *g, int ncomp, void *out_full, const void *y_full, int n, const double *coef_host,
                   const void *const *k_full_host, void *stream)
{
    (void)stream; GRID(g);
    if (n < 0 || n > 6) return fail(E_VALUE, "lincomb: n must be 0..6");
    TRY(oracle_lincomb(g, ncomp, out_full, y_full, n, coef_host, k_full_host));
    return 0;
}
/* products of tensor fields, cell by cell (the arithmetic of the device kernel: sums over the contracted index in order) */
#define FIELD_PRODUCT(T)                                                                                                      \
    for (long e = 0; e < pc; e++) {                                                                                           \
        const T *a = (const T *)a_full + e, *b = (const T *)b_full + e;                                                       \
        T *out = (T *)out_full + e;                                                                                           \
        const int nout = kind == 0 ? 1 : (kind == 1 || kind == 2 ? d : d * d);                                                \
        for (int io = 0; io < nout; io++) {                                                                                   \
            double re = 0, im = 0;                                                                                            \
            const int nsum = kind == 4 ? 1 : d;                                                                               \
            for (int m = 0; m < nsum; m++) {                                                                                  \
                int ia, ib;                                                                                                   \
                if (kind == 0) { ia = m; ib = m; }                                                                            \
                else if (kind == 1) { ia = io * d + m; ib = m; }                                                              \
                else if (kind == 2) { ia = m; ib = m * d + io; }                                                              \
                else if (kind == 3) { ia = (io / d) * d + m; ib = m * d + io % d; }                                           \
                else { ia = io / d; ib = io % d; }                                                                            \
                const double ar = (double)a[(long)(ia * w) * pc], br = (double)b[(long)(ib * w) * pc];                       \
                if (!complex_pairs) { re = re + ar * br; continue; }                                                          \
                const double ai = (double)a[(long)(ia * w + 1) * pc];                                                         \
                double bi = (double)b[(long)(ib * w + 1) * pc];                                                               \
                if (conjugate) bi = -bi;                                                                                      \
                re = re + (ar * br - ai * bi);                                                                                \
                im = im + (ar * bi + ai * br);                                                                                \
            }                                                                                                                 \
            out[(long)(io * w) * pc] = (T)re;                                                                                 \
            if (complex_pairs) out[(long)(io * w + 1) * pc] = (T)im;                                                          \
        }                                                                                                                     \
    }
int pdehip_field_product(const pdehip_grid_t *g, int kind, int complex_pairs, int conjugate, const void *a_full, const void *b_full, void *out_full,
                         void *stream)
{
    (void)stream; GRID(g);
    if (!a_full || !b_full || !out_full) return fail(E_VALUE, "field_product: NULL pointer");
    if (kind < 0 || kind > 4) return fail(E_VALUE, "field_product: kind 0 (v.v), 1 (T.v), 2 (v.T), 3 (T.T) or 4 (outer)");
    int64_t lay[8];
    pdehip_layout(g, lay);
    const long pc = (long)lay[2];
    const int d = g->ndim, w = complex_pairs ? 2 : 1;
    if (g->dtype == PDEHIP_F64) { FIELD_PRODUCT(double) } else { FIELD_PRODUCT(float) }
    return 0;
}
int pdehip_rk4_combine(const pdehip_grid_t *g, int ncomp, void *y, const void *k1, const void *k2, const void *k3, const void *k4,
                       void *stream)
{ (void)stream; GRID(g); TRY(oracle_rk4_combine(g, ncomp, y, k1, k2, k3, k4)); return 0; }
int pdehip_ab2_combine(const pdehip_grid_t *g, int ncomp, void *y, const void *rate_cur, const void *rate_prev, double dt, void *stream)
{ (void)stream; GRID(g); TRY(oracle_ab2_combine(g, ncomp, y, rate_cur, rate_prev, dt)); return 0; }
int pdehip_rkf45_combine(const pdehip_grid_t *g, int ncomp, const void *y, void *ynew, const void *const *k6_host, double *err_dev,
                         void *stream)
{ (void)stream; GRID(g); TRY(oracle_rkf45_combine(g, ncomp, y, ynew, k6_host, err_dev)); return 0; }
int oracle_euler_adaptive_combine(const pdehip_grid_t *g, int ncomp, const void *y, const void *rate, double dt, const void *half, const void *k,
                                  void *out, double *err);
int pdehip_euler_adaptive_combine(const pdehip_grid_t *g, int ncomp, const void *y, const void *rate, double dt, const void *half, const void *k,
                                  void *out, double *err_dev, void *stream)
{ (void)stream; GRID(g); TRY(oracle_euler_adaptive_combine(g, ncomp, y, rate, dt, half, k, out, err_dev)); return 0; }
int oracle_max_abs_pairs(const pdehip_grid_t *g, int npairs, const void *a, double *out);
int pdehip_max_abs_pairs(const pdehip_grid_t *g, int npairs, const void *arr_full, double *out_dev, void *stream)
{ (void)stream; GRID(g); TRY(oracle_max_abs_pairs(g, npairs, arr_full, out_dev)); return 0; }
int pdehip_max_abs_diff(const pdehip_grid_t *g, int ncomp, const void *a, const void *b, double *out_dev, void *stream)
{ (void)stream; GRID(g); TRY(oracle_max_abs_diff(g, ncomp, a, b, out_dev)); return 0; }

int pdehip_integrate(const pdehip_grid_t *g, int ncomp, const void *arr_full, double cell_volume, double *out_dev, void *stream)
{ (void)stream; GRID(g); TRY(oracle_integrate(g, ncomp, arr_full, cell_volume, out_dev)); return 0; }
int oracle_count_nonfinite(const pdehip_grid_t *g, int ncomp, const void *arr_full, double *out);
int pdehip_count_nonfinite(const pdehip_grid_t *g, int ncomp, const void *arr_full, double *out_dev, void *stream)
{ (void)stream; GRID(g); TRY(oracle_count_nonfinite(g, ncomp, arr_full, out_dev)); return 0; }
int pdehip_add_gaussian_noise(const pdehip_grid_t *g, int ncomp, void *y_full, double scale, uint64_t seed, uint64_t counter,
                              uint64_t cell_offset, void *stream)
{ (void)stream; GRID(g); TRY(oracle_add_gaussian_noise(g, ncomp, y_full, scale, seed, counter, cell_offset)); return 0; }

/* ---- device-only fused sweeps: "not covered" unless PDEHIP_SHIM_FUSED=1 --------------------- */
static int faces_simple(const pdehip_bc_face_t *f, int ndim)
{
    /* the device kernels evaluate scalar first-order faces on the fly; mimic that coverage rule */
    for (int i = 0; i < 2 * ndim; i++)
        if (f[i].kind != PDEHIP_BC_ORDER1 || (f[i].flags & PDEHIP_BCF_ARRAYS)) return 0;
    return 1;
}
int pdehip_diffusion_euler2(const pdehip_grid_t *g, const pdehip_bc_face_t *faces, const void *in_full, void *out_full,
                            double diffusivity, double dt, int *done, void *stream)
{
    (void)stream; GRID(g);
    *done = 0;
    if (!fused_enabled() || g->ndim < 2 || !faces_simple(faces, g->ndim)) return 0;
    size_t nb = full_bytes(g, 1);
    void *tmp = malloc(nb), *src = malloc(nb);
    memcpy(src, in_full, nb);   /* ghost cells of `in` are neither read nor written */
    int rc = oracle_set_ghost_cells(g, 1, faces, src);
    if (!rc) rc = oracle_laplace_euler(g, src, src, tmp, diffusivity, dt);
    if (!rc) rc = oracle_set_ghost_cells(g, 1, faces, tmp);
    if (!rc) rc = oracle_laplace_euler(g, tmp, tmp, out_full, diffusivity, dt);
    free(tmp); free(src);
    TRY(rc);
    *done = 1;
    return 0;
}
int pdehip_euler_multi_2d(const pdehip_grid_t *g, const pdehip_rhs_t *rhs, const void *in_full, void *out_full, double dt, int nsteps,
                          int *done, void *stream)
{
    (void)stream; GRID(g);
    *done = 0;
    if (!fused_enabled() || g->ndim != 2 || nsteps < 1 || nsteps > (rhs->kind == PDEHIP_RHS_DIFFUSION ? 8 : 4)) return 0;
    if (!faces_simple(rhs->bc_c, 2) || (rhs->kind == PDEHIP_RHS_CAHN_HILLIARD && !faces_simple(rhs->bc_mu, 2))) return 0;
    size_t nb = full_bytes(g, 1);
    void *a = malloc(nb), *b = calloc(1, nb), *res = NULL;
    memcpy(a, in_full, nb);
    pdehip_rhs_t r = *rhs;
    void *mu = NULL;
    if (rhs->kind == PDEHIP_RHS_CAHN_HILLIARD) r.scratch_mu = mu = calloc(1, nb);
    int rc = oracle_euler_run(g, &r, a, b, dt, nsteps, &res);
    if (!rc) {   /* interior only: the ghost cells of `out` are not part of the contract */
        int64_t lay[8];
        pdehip_layout(g, lay);
        const int esz = g->dtype == PDEHIP_F64 ? 8 : 4;
        for (int64_t i = 0; i < g->shape[0]; i++)
            memcpy((char *)out_full + (lay[3] + i * lay[1]) * esz, (char *)res + (lay[3] + i * lay[1]) * esz, (size_t)g->shape[1] * esz);
    }
    free(a); free(b); free(mu);
    TRY(rc);
    *done = 1;
    return 0;
}
int pdehip_diffusion_euler2_slab(const pdehip_grid_t *g_sub, const pdehip_bc_face_t *faces, const void *in_full, void *out_full,
                                 double diffusivity, double dt, int halo_sides, int *done, void *stream)
{
    (void)g_sub; (void)faces; (void)in_full; (void)out_full; (void)diffusivity; (void)dt; (void)halo_sides; (void)stream;
    *done = 0;
    return 0;
}
int pdehip_cahn_hilliard_fused(const pdehip_grid_t *g, const pdehip_bc_face_t *faces_c, const pdehip_bc_face_t *faces_mu,
                               const void *c_full, void *out_full, double gamma, double dt, int euler, int *done, void *stream)
{
    (void)stream; GRID(g);
    *done = 0;
    if (!fused_enabled() || g->ndim < 2 || !faces_simple(faces_c, g->ndim) || !faces_simple(faces_mu, g->ndim)) return 0;
    size_t nb = full_bytes(g, 1);
    void *mu = calloc(1, nb), *src = malloc(nb);
    memcpy(src, c_full, nb);
    int rc = oracle_set_ghost_cells(g, 1, faces_c, src);
    if (!rc) rc = oracle_cahn_hilliard_mu(g, src, mu, gamma);
    if (!rc) rc = oracle_set_ghost_cells(g, 1, faces_mu, mu);
    if (!rc) rc = euler ? oracle_laplace_euler(g, mu, src, out_full, 1.0, dt) : oracle_laplace_scaled(g, mu, out_full, 1.0, dt);
    free(mu); free(src);
    TRY(rc);
    *done = 1;
    return 0;
}

/* ---- fused steppers ------------------------------------------------------------------------ */
/* faces with explicit time dependence (pdehip_rhs_t::bc_program): refreshed for the time of every evaluation; the Runge-Kutta
 * sequences with a time per stage are the generic loops of csrc/pdehip_rk_loops.h (pdehip_shim_comm.cpp) */
int shim_timed_rk4_step(const pdehip_grid_t *g, const pdehip_rhs_t *rhs, void *y, void *const *w, double dt, double t);
int shim_timed_rkf45_attempt(const pdehip_grid_t *g, const pdehip_rhs_t *rhs, void *y, void *ynew, void *const *w, double dt, double t, double *err);
int pdehip_rhs_scaled(const pdehip_grid_t *g, const pdehip_rhs_t *rhs, void *y_full, void *k_out_full, double dt, void *stream)
{
    (void)stream; GRID(g);
    if (rhs->bc_program) { int rc = pdehip_bcprog_run(rhs->bc_program, rhs->t, y_full, stream); if (rc) return rc; }
    TRY(oracle_rhs_scaled(g, rhs, y_full, k_out_full, dt));
    return 0;
}
int pdehip_euler_run(const pdehip_grid_t *g, const pdehip_rhs_t *rhs, void *buf_a, void *buf_b, double dt, int64_t nsteps,
                     void **result, void *stream)
{
    (void)stream; GRID(g);
    if (nsteps < 0) return fail(E_VALUE, "nsteps must be >= 0");
    if (rhs->bc_program) {
        void *cur = buf_a, *nxt = buf_b, *res = NULL;
        for (int64_t s = 0; s < nsteps; s++) {
            int rc = pdehip_bcprog_run(rhs->bc_program, rhs->t + (double)s * dt, cur, stream);
            if (rc) return rc;
            TRY(oracle_euler_run(g, rhs, cur, nxt, dt, 1, &res));
            void *t = cur; cur = nxt; nxt = t;
        }
        *result = cur;
        return 0;
    }
    TRY(oracle_euler_run(g, rhs, buf_a, buf_b, dt, nsteps, result));
    return 0;
}
int pdehip_rk4_step(const pdehip_grid_t *g, const pdehip_rhs_t *rhs, void *y_full, void *const *work5_host, double dt, void *stream)
{
    (void)stream; GRID(g);
    if (rhs->bc_program) return shim_timed_rk4_step(g, rhs, y_full, work5_host, dt, rhs->t);
    TRY(oracle_rk4_step(g, rhs, y_full, work5_host, dt));
    return 0;
}
int pdehip_rk4_run(const pdehip_grid_t *g, const pdehip_rhs_t *rhs, void *y_full, void *const *work5_host, double dt, int64_t nsteps,
                   void *stream)
{
    (void)stream; GRID(g);
    for (int64_t s = 0; s < nsteps; s++) {
        if (rhs->bc_program) { int rc = shim_timed_rk4_step(g, rhs, y_full, work5_host, dt, rhs->t + (double)s * dt); if (rc) return rc; }
        else TRY(oracle_rk4_step(g, rhs, y_full, work5_host, dt));
    }
    return 0;
}
int pdehip_rkf45_attempt(const pdehip_grid_t *g, const pdehip_rhs_t *rhs, void *y_full, void *ynew_full, void *const *work7_host,
                         double dt, double *err_dev, void *stream)
{
    (void)stream; GRID(g);
    if (rhs->bc_program) return shim_timed_rkf45_attempt(g, rhs, y_full, ynew_full, work7_host, dt, rhs->t, err_dev);
    TRY(oracle_rkf45_attempt(g, rhs, y_full, ynew_full, work7_host, dt, err_dev));
    return 0;
}
int pdehip_ab2_step(const pdehip_grid_t *g, const pdehip_rhs_t *rhs, void *y_in_full, void *y_out_full, void *rate_cur_full,
                    const void *rate_prev_full, double dt, int *fused, void *stream)
{
    (void)stream; GRID(g);
    *fused = 0;
    if (!fused_enabled() || g->ndim < 2) return 0;
    if (rhs->bc_program) { int rc = pdehip_bcprog_run(rhs->bc_program, rhs->t, y_in_full, stream); if (rc) return rc; }
    TRY(oracle_rhs_scaled(g, rhs, y_in_full, rate_cur_full, 1.0));
    memcpy(y_out_full, y_in_full, full_bytes(g, 1));
    TRY(oracle_ab2_combine(g, 1, y_out_full, rate_cur_full, rate_prev_full, dt));
    *fused = 1;
    return 0;
}

/* ---- boundary-condition programs (pdehip_bcprog_*): gcc instead of hiprtc, a host loop instead of a kernel ---------------- */
typedef void (*bc_face_fn)(int, double, double, double, double, double, double, double *, double *);
typedef struct {
    void *dl;
    bc_face_fn fn;
    int nfaces;
    pdehip_bcprog_face_t *faces;
    int64_t (*sgeo)[3];   /* faces that read the field: element offset of face cell (0, 0) and the two pitches */
    int reads, esz;
} shim_bcprog_t;
int pdehip_bcprog_create(const char *source, int nfaces, const pdehip_bcprog_face_t *faces, const pdehip_grid_t *grid, void **handle)
{
    static int counter = 0;
    if (!source || !faces || !handle || nfaces < 1 || nfaces > 64) return fail(E_VALUE, "bcprog_create: NULL pointer or bad face count");
    int reads = 0;
    for (int f = 0; f < nfaces; f++) reads |= faces[f].reads_value != 0;
    if (reads && !grid) return fail(E_VALUE, "bcprog_create: a face reads the field but no grid is given");
    char dir[] = "/tmp/pdehip_shimbc_XXXXXX";
    if (!mkdtemp(dir)) return fail(E_RUNTIME, "shim: mkdtemp failed");
    char src[600], so[600], cmd[2000];
    snprintf(src, sizeof(src), "%s/b%d.c", dir, counter);
    snprintf(so, sizeof(so), "%s/b%d.so", dir, counter++);
    FILE *f = fopen(src, "w");
    if (!f) return fail(E_RUNTIME, "shim: cannot write %s", src);
    fprintf(f, "#include <math.h>\n#include <stdbool.h>\n#define PDEHIP_BC_FN static inline\n%s\n"
               "void bc_face_entry(int face, double value, double dx, double c0, double c1, double c2, double t, double *A, double *B)\n"
               "{ bc_face(face, value, dx, c0, c1, c2, t, A, B); }\n", source);
    fclose(f);
    snprintf(cmd, sizeof(cmd), "gcc -O2 -fPIC -shared -std=gnu11 -ffp-contract=off -fno-fast-math -o %s %s -lm 2> %s/err.txt", so, src, dir);
    if (system(cmd) != 0) {
        char msg[300] = "";
        snprintf(cmd, sizeof(cmd), "%s/err.txt", dir);
        FILE *e = fopen(cmd, "r");
        if (e) { size_t k = fread(msg, 1, sizeof(msg) - 1, e); msg[k] = 0; fclose(e); }
        return fail(E_VALUE, "boundary-condition program does not compile: %s", msg);
    }
    shim_bcprog_t *b = calloc(1, sizeof(*b));
    b->dl = dlopen(so, RTLD_NOW | RTLD_LOCAL);
    unlink(src); unlink(so);
    snprintf(cmd, sizeof(cmd), "%s/err.txt", dir); unlink(cmd);
    rmdir(dir);
    if (!b->dl) { free(b); return fail(E_RUNTIME, "shim: dlopen failed: %s", dlerror()); }
    b->fn = (bc_face_fn)dlsym(b->dl, "bc_face_entry");
    if (!b->fn) { dlclose(b->dl); free(b); return fail(E_RUNTIME, "shim: bc_face_entry not found"); }
    b->nfaces = nfaces;
    b->faces = malloc(sizeof(*faces) * (size_t)nfaces);
    memcpy(b->faces, faces, sizeof(*faces) * (size_t)nfaces);
    b->sgeo = calloc((size_t)nfaces, sizeof(*b->sgeo));
    b->reads = reads;
    b->esz = reads && grid->dtype == PDEHIP_F32 ? 4 : 8;
    if (reads) {
        /* full arrays of the shim: C order, one ghost layer per axis, no padding */
        const int nd = grid->ndim;
        int64_t pitch[3] = {0, 0, 0}, pc = 1, off = 0;
        for (int a = nd - 1; a >= 0; a--) { pitch[a] = pc; pc *= grid->shape[a] + 2; }
        for (int a = 0; a < nd; a++) off += pitch[a];
        for (int q = 0; q < nfaces; q++) {
            const pdehip_bcprog_face_t *F = &faces[q];
            if (!F->reads_value) continue;
            if (F->axis < 0 || F->axis >= nd || F->value_index < 0 || F->value_index >= grid->shape[F->axis] || F->component < 0)
                return fail(E_VALUE, "bcprog_create: face %d: bad axis / value cell / component", q);
            int others[2] = {0, 0}, no = 0;
            for (int a = 0; a < nd; a++) if (a != F->axis) others[no++] = a;
            if ((no >= 1 ? grid->shape[others[0]] : 1) != F->m1 || (no >= 2 ? grid->shape[others[1]] : 1) != F->m2)
                return fail(E_VALUE, "bcprog_create: face %d: extents do not match the grid", q);
            b->sgeo[q][0] = off + (int64_t)F->component * pc + F->value_index * pitch[F->axis];
            b->sgeo[q][1] = no >= 1 ? pitch[others[0]] : 0;
            b->sgeo[q][2] = no >= 2 ? pitch[others[1]] : 0;
        }
    }
    *handle = b;
    return 0;
}
int pdehip_bcprog_run(void *handle, double t, const void *state_full, void *stream)
{
    (void)stream;
    shim_bcprog_t *b = handle;
    if (!b) return fail(E_VALUE, "bcprog_run: NULL handle");
    if (b->reads && !state_full) return fail(E_VALUE, "bcprog_run: the conditions read the field, but no field is given");
    for (int f = 0; f < b->nfaces; f++) {
        const pdehip_bcprog_face_t *F = &b->faces[f];
        for (int64_t i1 = 0; i1 < F->m1; i1++)
            for (int64_t i2 = 0; i2 < F->m2; i2++) {
                double c[3];
                for (int k = 0; k < 3; k++)
                    c[k] = F->index[k] == 0 ? F->origin[k] : ((double)(F->first[k] + (F->index[k] == 1 ? i1 : i2)) + 0.5) * F->step[k] + F->origin[k];
                double value = 0;
                if (F->reads_value) {
                    const int64_t o = b->sgeo[f][0] + i1 * b->sgeo[f][1] + i2 * b->sgeo[f][2];
                    value = b->esz == 8 ? ((const double *)state_full)[o] : (double)((const float *)state_full)[o];
                }
                double a = 0, bb = 0;
                b->fn(f, value, F->dx, c[0], c[1], c[2], t, &a, &bb);
                F->const_arr[i1 * F->m2 + i2] = a;
                F->factor_arr[i1 * F->m2 + i2] = bb;
            }
    }
    return 0;
}
int pdehip_bcprog_destroy(void *handle)
{
    shim_bcprog_t *b = handle;
    if (!b) return 0;
    if (b->dl) dlclose(b->dl);
    free(b->faces);
    free(b->sgeo);
    free(b);
    return 0;
}

/* ---- run-time specialised right-hand sides: gcc instead of hiprtc ---------------------------- */
typedef struct { double d1[3], d2[3], gr[3]; } shim_der_t;   /* PdeDer of csrc/pdehip_device.h */
typedef double (*epilogue_fn)(double, double, double, double, double, double, const double *, shim_der_t);
typedef struct {
    void *dl[2];
    epilogue_fn fn[2];
    int nbody;
} shim_jit_t;

static int compile_epilogue(const char *body, void **dl, epilogue_fn *fn)
{
    static int counter = 0;
    char dir[] = "/tmp/pdehip_shim_XXXXXX";
    if (!mkdtemp(dir)) return fail(E_RUNTIME, "shim: mkdtemp failed");
    char src[600], so[600], cmd[2000];
    snprintf(src, sizeof(src), "%s/e%d.c", dir, counter);
    snprintf(so, sizeof(so), "%s/e%d.so", dir, counter++);
    FILE *f = fopen(src, "w");
    if (!f) return fail(E_RUNTIME, "shim: cannot write %s", src);
    fprintf(f, "#include <math.h>\n#include <stdbool.h>\ntypedef struct { double d1[3], d2[3], gr[3]; } PdeDer;\n"
               "double pde_epilogue(double c, double lap, double gsq, double e0, double e1, double e2, const double *p, PdeDer d)\n{\n"
               "(void)c; (void)lap; (void)gsq; (void)e0; (void)e1; (void)e2; (void)p; (void)d;\n%s\n}\n", body);
    fclose(f);
    snprintf(cmd, sizeof(cmd), "gcc -O2 -fPIC -shared -std=gnu11 -ffp-contract=off -fno-fast-math -o %s %s -lm 2> %s/err.txt", so, src, dir);
    if (system(cmd) != 0) {
        char msg[300] = "";
        snprintf(cmd, sizeof(cmd), "%s/err.txt", dir);
        FILE *e = fopen(cmd, "r");
        if (e) { size_t k = fread(msg, 1, sizeof(msg) - 1, e); msg[k] = 0; fclose(e); }
        return fail(E_VALUE, "shim: epilogue does not compile: %s", msg);
    }
    *dl = dlopen(so, RTLD_NOW | RTLD_LOCAL);
    if (!*dl) return fail(E_RUNTIME, "shim: dlopen failed: %s", dlerror());
    *fn = (epilogue_fn)dlsym(*dl, "pde_epilogue");
    unlink(src); unlink(so);
    snprintf(cmd, sizeof(cmd), "%s/err.txt", dir); unlink(cmd);
    rmdir(dir);
    return *fn ? 0 : fail(E_RUNTIME, "shim: pde_epilogue not found");
}
int pdehip_jit_create(const char *epilogue_body, void **handle)
{
    shim_jit_t *j = calloc(1, sizeof(*j));
    j->nbody = 1;
    int rc = compile_epilogue(epilogue_body, &j->dl[0], &j->fn[0]);
    if (rc) { free(j); return rc; }
    *handle = j;
    return 0;
}
int pdehip_jit_create2(const char *body1, const char *body2, void **handle)
{
    shim_jit_t *j = calloc(1, sizeof(*j));
    j->nbody = 2;
    int rc = compile_epilogue(body1, &j->dl[0], &j->fn[0]);
    if (!rc) rc = compile_epilogue(body2, &j->dl[1], &j->fn[1]);
    if (rc) { free(j); return rc; }
    *handle = j;
    return 0;
}
int pdehip_jit_destroy(void *handle)
{
    shim_jit_t *j = handle;
    if (!j) return 0;
    for (int i = 0; i < 2; i++) if (j->dl[i]) dlclose(j->dl[i]);
    free(j);
    return 0;
}
int pdehip_jit_check(void *handle, int dtype, int ndim) { (void)handle; (void)dtype; (void)ndim; return 0; }

/* out = fn(in, laplace(in), gradient_squared(in), extras; params) on the interior; `in` must carry valid ghost cells */
static int shim_pointwise(epilogue_fn fn, const pdehip_grid_t *g, const void *in, const void *const *extra3, void *out,
                          const double *params)
{
    ngrid_t n;
    norm_grid(g, &n);
    size_t nb = full_bytes(g, 1);
    void *lap = calloc(1, nb), *gsq = calloc(1, nb);
    void *der[2][3] = {{NULL, NULL, NULL}, {NULL, NULL, NULL}};   /* [order - 1][normalised axis] */
    int rc = oracle_laplace(g, in, lap, PDEHIP_OUT_FULL);
    if (!rc) rc = oracle_gradient_squared(g, 1, in, gsq, PDEHIP_OUT_FULL);
    for (int a = 0; a < g->ndim && !rc; a++)
        for (int order = 1; order <= 2 && !rc; order++) {
            der[order - 1][3 - g->ndim + a] = calloc(1, nb);
            rc = oracle_axis_derivative(g, a, order, PDEHIP_CENTRAL, in, der[order - 1][3 - g->ndim + a], PDEHIP_OUT_FULL);
        }
    /* central gradient: all components at once (component-major full arrays) */
    void *grad = calloc((size_t)(g->ndim > 0 ? g->ndim : 1), nb);
    if (!rc) rc = oracle_gradient(g, PDEHIP_CENTRAL, in, grad, PDEHIP_OUT_FULL);
    if (rc) { free(lap); free(gsq); free(grad); for (int q = 0; q < 6; q++) free(der[q / 3][q % 3]); return rc; }
    /* results go to a scratch first: `out` may alias an extra array */
    void *res = malloc(nb);
    memcpy(res, out, nb);
    for (int64_t i = 0; i < n.n[0]; i++)
        for (int64_t j = 0; j < n.n[1]; j++)
            for (int64_t k = 0; k < n.n[2]; k++) {
                int64_t at = n.off + i * n.p[0] + j * n.p[1] + k;
                double e[3] = {0, 0, 0};
                shim_der_t d = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
                if (g->dtype == PDEHIP_F64) {
                    for (int m = 0; m < 3; m++) if (extra3 && extra3[m]) e[m] = ((const double *)extra3[m])[at];
                    for (int q = 0; q < 3; q++) {
                        if (der[0][q]) d.d1[q] = ((double *)der[0][q])[at];
                        if (der[1][q]) d.d2[q] = ((double *)der[1][q])[at];
                    }
                    for (int a = 0; a < g->ndim; a++) d.gr[3 - g->ndim + a] = ((double *)grad)[(int64_t)a * n.pc + at];
                    ((double *)res)[at] = fn(((const double *)in)[at], ((double *)lap)[at], ((double *)gsq)[at], e[0], e[1], e[2], params, d);
                } else {
                    for (int m = 0; m < 3; m++) if (extra3 && extra3[m]) e[m] = ((const float *)extra3[m])[at];
                    for (int q = 0; q < 3; q++) {
                        if (der[0][q]) d.d1[q] = ((float *)der[0][q])[at];
                        if (der[1][q]) d.d2[q] = ((float *)der[1][q])[at];
                    }
                    for (int a = 0; a < g->ndim; a++) d.gr[3 - g->ndim + a] = ((float *)grad)[(int64_t)a * n.pc + at];
                    ((float *)res)[at] = (float)fn(((const float *)in)[at], ((float *)lap)[at], ((float *)gsq)[at], e[0], e[1], e[2], params, d);
                }
            }
    for (int q = 0; q < 6; q++) free(der[q / 3][q % 3]);
    free(grad);
    /* interior only: the ghost cells of `out` are left untouched like the device kernel does */
    int esz = g->dtype == PDEHIP_F64 ? 8 : 4;
    for (int64_t i = 0; i < n.n[0]; i++)
        for (int64_t j = 0; j < n.n[1]; j++) {
            int64_t at = n.off + i * n.p[0] + j * n.p[1];
            memcpy((char *)out + at * esz, (char *)res + at * esz, (size_t)n.n[2] * esz);
        }
    free(res); free(lap); free(gsq);
    return 0;
}
int pdehip_jit_apply(void *handle, const pdehip_grid_t *g, void *in_full, const void *const *extra3_host, void *out_full,
                     const double *params_host, int nparams, const pdehip_bc_face_t *in_faces, void *stream)
{
    (void)stream; (void)nparams; GRID(g);
    shim_jit_t *j = handle;
    if (!j || j->nbody != 1) return fail(E_VALUE, "jit_apply: bad handle");
    if (in_faces) TRY(oracle_set_ghost_cells(g, 1, in_faces, in_full));
    TRY(shim_pointwise(j->fn[0], g, in_full, extra3_host, out_full, params_host));
    return 0;
}
int pdehip_jit_apply_stage(void *handle, const pdehip_grid_t *g, void *in_full, const void *const *extra3_host, void *k_out_full,
                           const double *params_host, int nparams, const pdehip_bc_face_t *in_faces, int kind, const void *y_full,
                           int nk, const void *const *k_prev_host, const double *coef_host, double c_new, void *out2_full,
                           double *err_dev, int *done, void *stream)
{
    (void)stream; GRID(g);
    *done = 0;
    if (!fused_enabled() || g->ndim < 2) return 0;
    size_t nb = full_bytes(g, 1);
    void *k = kind == 0 ? k_out_full : calloc(1, nb);
    int rc = pdehip_jit_apply(handle, g, in_full, extra3_host, k, params_host, nparams, in_faces, NULL);
    if (!rc && kind == 0) {
        const void *ks[6]; double cf[6];
        for (int m = 0; m < nk; m++) { ks[m] = k_prev_host[m]; cf[m] = coef_host[m]; }
        ks[nk] = k; cf[nk] = c_new;
        rc = oracle_lincomb(g, 1, out2_full, y_full, nk + 1, cf, ks);
    } else if (!rc && kind == 1) {
        if (out2_full != y_full) memcpy(out2_full, y_full, nb);
        rc = oracle_rk4_combine(g, 1, out2_full, k_prev_host[0], k_prev_host[1], k_prev_host[2], k);
    } else if (!rc && kind == 2) {
        /* k2 does not enter the RKF45 tail: any valid array stands in for it */
        const void *k6[6] = {k_prev_host[0], k_prev_host[0], k_prev_host[1], k_prev_host[2], k_prev_host[3], k};
        rc = oracle_rkf45_combine(g, 1, y_full, out2_full, k6, err_dev);
    } else if (!rc && kind == 4) {
        rc = oracle_euler_adaptive_combine(g, 1, y_full, k_prev_host[0], coef_host[0], k_prev_host[1], k, out2_full, err_dev);
    } else if (!rc) {
        rc = 1;
    }
    if (kind != 0) free(k);
    if (rc > 99) return rc;
    TRY(rc);
    *done = 1;
    return 0;
}
int pdehip_jit_euler2(void *handle, const pdehip_grid_t *g, const void *in_full, void *out_full, const double *params_host,
                      int nparams, const pdehip_bc_face_t *faces, int *done, void *stream)
{
    (void)stream; (void)nparams; GRID(g);
    shim_jit_t *j = handle;
    *done = 0;
    if (!fused_enabled() || g->ndim < 2 || !faces_simple(faces, g->ndim)) return 0;
    size_t nb = full_bytes(g, 1);
    void *src = malloc(nb), *tmp = calloc(1, nb);
    memcpy(src, in_full, nb);
    int rc = oracle_set_ghost_cells(g, 1, faces, src);
    if (!rc) rc = shim_pointwise(j->fn[0], g, src, NULL, tmp, params_host);
    if (!rc) rc = oracle_set_ghost_cells(g, 1, faces, tmp);
    if (!rc) rc = shim_pointwise(j->fn[0], g, tmp, NULL, out_full, params_host);
    free(src); free(tmp);
    TRY(rc);
    *done = 1;
    return 0;
}
int pdehip_jit_fused2(void *handle, const pdehip_grid_t *g, const void *in_full, void *out_full, const double *params_host,
                      int nparams, const pdehip_bc_face_t *faces_u, const pdehip_bc_face_t *faces_tmp, int *done, void *stream)
{
    (void)stream; (void)nparams; GRID(g);
    shim_jit_t *j = handle;
    *done = 0;
    if (!j || j->nbody != 2) return fail(E_VALUE, "jit_fused2: bad handle");
    if (!fused_enabled() || g->ndim < 2 || !faces_simple(faces_u, g->ndim) || !faces_simple(faces_tmp, g->ndim)) return 0;
    size_t nb = full_bytes(g, 1);
    void *src = malloc(nb), *tmp = calloc(1, nb);
    memcpy(src, in_full, nb);
    const void *ex[3] = {src, NULL, NULL};
    int rc = oracle_set_ghost_cells(g, 1, faces_u, src);
    if (!rc) rc = shim_pointwise(j->fn[0], g, src, NULL, tmp, params_host);
    if (!rc) rc = oracle_set_ghost_cells(g, 1, faces_tmp, tmp);
    if (!rc) rc = shim_pointwise(j->fn[1], g, tmp, ex, out_full, params_host);
    free(src); free(tmp);
    TRY(rc);
    *done = 1;
    return 0;
}

/* the Euler loop over the passes of an expression PDE: the same sequence of pdehip_jit_apply calls */
int pdehip_jit_euler_run(const pdehip_grid_t *g, const pdehip_jit_pass_t *passes, int npasses, void *const *fixed, int nfixed,
                         void *state_a, void *state_b, int ncomp, double dt, double t0, int uses_time, int64_t nsteps,
                         void *bc_program, void **result, void *stream)
{
    (void)uses_time; GRID(g);
    if (!passes || !state_a || !state_b || !result || (nfixed > 0 && !fixed)) return fail(E_VALUE, "jit_euler_run: NULL pointer");
    if (npasses < 1 || ncomp < 1 || nsteps < 0) return fail(E_VALUE, "jit_euler_run: bad pass / component / step count");
    const size_t comp_bytes = full_bytes(g, 1);
    char *cur = state_a, *nxt = state_b;
    for (int q = 0; q < npasses; q++) {
        const int32_t idx[5] = {passes[q].src, passes[q].extras[0], passes[q].extras[1], passes[q].extras[2], passes[q].out};
        for (int m = 0; m < 5; m++) {
            if (idx[m] == PDEHIP_JIT_NONE && m != 0 && m != 4) continue;
            if (idx[m] == PDEHIP_JIT_NONE || idx[m] >= nfixed || idx[m] < -ncomp)
                return fail(E_VALUE, "jit_euler_run: pass %d refers to array %d (fixed: %d, components: %d)", q, (int)idx[m], nfixed, ncomp);
        }
    }
    for (int64_t s = 0; s < nsteps; s++) {
        const double params[2] = {dt, t0 + (double)s * dt};
        if (bc_program) { int rc = pdehip_bcprog_run(bc_program, params[1], cur, stream); if (rc) return rc; }
        for (int q = 0; q < npasses; q++) {
            const pdehip_jit_pass_t *p = &passes[q];
            const void *ex[3];
            for (int m = 0; m < 3; m++)
                ex[m] = p->extras[m] == PDEHIP_JIT_NONE ? NULL : (p->extras[m] >= 0 ? fixed[p->extras[m]] : (void *)(cur + (size_t)(-1 - p->extras[m]) * comp_bytes));
            void *src = p->src >= 0 ? fixed[p->src] : (void *)(cur + (size_t)(-1 - p->src) * comp_bytes);
            void *out = p->out >= 0 ? fixed[p->out] : (void *)(nxt + (size_t)(-1 - p->out) * comp_bytes);
            if (p->exchange) {   /* decomposed grids: the ghost layers of the operand travel first (pdehip_exchange_t) */
                const pdehip_exchange_t *x = p->exchange;
                int nb6[6], rc;
                for (int i = 0; i < 6; i++) nb6[i] = x->nb6[i];
                rc = x->blocks ? pdehip_block_exchange(x->comm, g, nb6, src, stream) : pdehip_halo_exchange(x->comm, g, src, x->lower, x->upper, stream);
                if (rc) return rc;
            }
            TRY(pdehip_jit_apply(p->handle, g, src, ex, out, params, 2, p->faces, stream));
        }
        char *t = cur; cur = nxt; nxt = t;
    }
    *result = cur;
    return 0;
}

#ifndef SHIM_WITH_COMM
/* slab-parallel layer: provided by pdehip_shim_comm.cpp when it is linked in */
#define NOCOMM(name, ...) int pdehip_##name(__VA_ARGS__) { return fail(E_NOTIMPL, "shim: pdehip_" #name " needs the comm part of the shim"); }
NOCOMM(comm_unique_id, const char *a, void *b)
NOCOMM(comm_create, const char *a, const void *b, int c, int d, void **e)
NOCOMM(comm_destroy, void *a)
NOCOMM(halo_exchange, void *a, const pdehip_grid_t *b, void *c, int d, int e, void *f)
NOCOMM(allreduce_max, void *a, double *b, void *c)
NOCOMM(slab_euler_run, void *a, const pdehip_grid_t *b, const pdehip_rhs_t *c, int d, int e, void *f, void *g, double h, int64_t i, void **j, void *k)
NOCOMM(slab_euler2_supported, const pdehip_grid_t *a, const pdehip_rhs_t *b, int *c)
NOCOMM(slab_euler2_run, void *a, const pdehip_grid_t *b, const pdehip_rhs_t *c, int d, int e, void *f, void *g, double h, int64_t i, void **j, void *k)
NOCOMM(slab_ch_supported, const pdehip_grid_t *a, const pdehip_rhs_t *b, int *c)
NOCOMM(slab_ch_sweep, void *a, const pdehip_grid_t *b, const pdehip_rhs_t *c, int d, int e, void *f, void *g, double h, int i, void *j)
#endif
