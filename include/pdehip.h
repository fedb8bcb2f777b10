/*
 * pdehip.h — C ABI of libpdehip.so, the MI355X (gfx950) backend for py-pde's
 * Cartesian finite-difference operators + explicit time steppers.
 *
 * This is the drop-in boundary: the entry points below are what a
 * `pde.backends` plugin binds through ctypes.  Every function cites the reference
 * interface it replaces (paths relative to the py-pde source tree).  There are no
 * torch / numpy types in any signature: plain pointers, sizes and POD structs.
 *
 * Conventions
 *   - every function returns 0 on success, non-zero on failure; the message of the
 *     last failure (per thread) is returned by pdehip_last_error().
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream).
 *   - all array pointers are DEVICE pointers unless the name says `host`.
 *   - a "full" array is the ghost-padded C-contiguous array of py-pde
 *     (pde/grids/base.py:329-337, pde/fields/base.py:116-160): shape
 *     (ncomp, N0+2, N1+2, N2+2) (unused axes dropped), interior cell (i,j,k) lives
 *     at [i+1, j+1, k+1].  A "valid" array is the C-contiguous interior
 *     (ncomp, N0, N1, N2).
 *   - ON THE DEVICE a full array keeps the ghost-padded index space of the reference but
 *     every row of the fastest axis is shifted/padded so that its first interior cell is
 *     16-byte aligned and the row pitch is a multiple of 16 bytes (all hot-kernel accesses
 *     are aligned dwordx4).  pdehip_layout() reports pitches and the allocation size;
 *     pdehip_valid_to_full / pdehip_hostfull_to_full convert from the host layouts.  Layers
 *     along axis 0 stay contiguous (one block per slab face for the halo exchange).
 *   - dtype PDEHIP_F32 stores fp32 but all arithmetic between a load and the store
 *     of one kernel runs in fp64 registers (numba promotes float32 array elements
 *     against float64 closure constants the same way, SURVEY.md §7 "fp32").
 *   - the arithmetic of every kernel follows the reference expression order and is
 *     compiled with -ffp-contract=off, so results are bit-identical to the CPU
 *     oracle (oracle/pde_oracle.c) and to the reference's eager torch-CPU backend.
 */
#ifndef PDEHIP_H
#define PDEHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PDEHIP_MAX_DIM 3
#define PDEHIP_ABI_VERSION 7

enum { PDEHIP_F64 = 0, PDEHIP_F32 = 1 };
/* derivative flavour, pde/backends/numba/operators/cartesian.py:386-587 `method` */
enum { PDEHIP_CENTRAL = 0, PDEHIP_FORWARD = 1, PDEHIP_BACKWARD = 2 };
/* layout of an operator's output array */
enum { PDEHIP_OUT_VALID = 0, PDEHIP_OUT_FULL = 1 };

/* Geometry of a CartesianGrid / UnitGrid (pde/grids/cartesian.py:36-146, :473-507).
 * `shape` and `dx` mirror grid.shape / grid.discretization. */
typedef struct pdehip_grid {
    int32_t ndim;                  /* 1, 2 or 3 */
    int32_t dtype;                 /* PDEHIP_F64 | PDEHIP_F32 */
    int64_t shape[PDEHIP_MAX_DIM]; /* valid cells per axis; unused axes must be 1 */
    double dx[PDEHIP_MAX_DIM];     /* grid.discretization */
} pdehip_grid_t;

/* One side of one axis of a BoundariesList, already reduced to virtual-point data
 * (pde/grids/boundaries/local.py:1611-1636 ConstBC1stOrderBase.set_ghost_cells,
 *  :2022-2061 ConstBC2ndOrderBase.set_ghost_cells, data from get_virtual_point_data
 *  :1728-1731 periodic, :1749-1753 Dirichlet, :1773-1778 Neumann, :1927-1938 Mixed,
 *  :2081-2103 Curvature):
 *      ghost = const + factor1 * full[index1 + 1]  (+ factor2 * full[index2 + 1])
 * evaluated on the interior of the face only (corners/edges are never written).
 */
enum { PDEHIP_BC_SKIP = 0, PDEHIP_BC_ORDER1 = 1, PDEHIP_BC_ORDER2 = 2 };
enum {
    PDEHIP_BCF_ARRAYS = 1, /* const/factor given per (component, face cell) as fp64 device
                              arrays of shape (ncomp, face cells in C order) */
    PDEHIP_BCF_NORMAL = 2  /* `bc.normal`: only vector component == axis is written */
};
typedef struct pdehip_bc_face {
    int32_t kind;  /* PDEHIP_BC_* */
    int32_t flags; /* PDEHIP_BCF_* */
    int64_t index1, index2; /* indices into the VALID array along the axis */
    double const_v, factor1, factor2; /* used when !(flags & PDEHIP_BCF_ARRAYS) */
    const double *const_arr, *factor1_arr, *factor2_arr; /* used when ARRAYS */
} pdehip_bc_face_t;
/* Faces are stored as faces[2*axis + upper] (lower = 0, upper = 1); they are applied in
 * the reference order axis 0 (upper, lower), axis 1 (upper, lower), ...
 * (pde/backends/numba/backend.py:335-340, pde/grids/boundaries/axis.py:236-238). */

/* Right-hand side of the PDEs on the hot path, evaluated by the fused steppers. */
enum {
    PDEHIP_RHS_DIFFUSION = 0,    /* D * laplace(c)            pde/pdes/diffusion.py:119-121 */
    PDEHIP_RHS_CAHN_HILLIARD = 1 /* laplace(c**3 - c - g*laplace(c)) pde/pdes/cahn_hilliard.py:115-122 */
};
typedef struct pdehip_rhs {
    int32_t kind;
    int32_t reserved;
    double param;                   /* diffusivity D, or interface_width g */
    pdehip_bc_face_t bc_c[2 * PDEHIP_MAX_DIM];  /* BCs of the state field */
    pdehip_bc_face_t bc_mu[2 * PDEHIP_MAX_DIM]; /* BCs of mu (Cahn-Hilliard only) */
    void *scratch_mu;               /* full array (Cahn-Hilliard only) */
    /* Faces whose coefficient arrays depend on time (expression conditions that contain `t`) or on the field (conditions that
     * are not affine in the adjacent value): a program from pdehip_bcprog_create that rewrites the const / factor arrays the face
     * tables above point to, or NULL.  Every entry point that evaluates the right-hand side runs it for the time AND the input
     * field of THAT evaluation first, on the same stream (only faces of bc_c may read the field) (Euler: t + i*dt,
     * Runge-Kutta: the stage times t + a_s*dt; pde/solvers/euler.py:172-175, pde/solvers/runge_kutta.py:52-59, :135-145 pass
     * `t` to the right-hand side, which hands it to the conditions as args={"t": t}: pde/pdes/diffusion.py:119-121). */
    void *bc_program;
    double t;                       /* time of a single evaluation (rhs_scaled, rk4_step, rkf45_attempt, ab2_step) / of the
                                       first step of a fixed-step run (euler_run, rk4_run); ignored without bc_program */
} pdehip_rhs_t;

/* ---- boundary conditions given as expressions, evaluated ON THE DEVICE --------------------------------------------------------
 * Replaces the per-call evaluation of ExpressionBC (pde/grids/boundaries/local.py:766-1150; numba twin
 * pde/backends/numba/_boundaries.py:256-394, torch twin pde/backends/torch/_boundaries.py:258-345).  A condition whose virtual
 * point F(value, dx, coords, t) is affine in the adjacent value is  ghost = A(dx, coords, t) + B(dx, coords, t) * value, i.e. a
 * first-order face with per-cell coefficient arrays (PDEHIP_BCF_ARRAYS).  A condition that is NOT affine in `value` is the same
 * face with A = F(value now, dx, coords, t), B = 0, rewritten from the field every time the conditions are applied
 * (`reads_value`: the program reads the cell `value_index` of the field along the face's axis, like the reference's
 * `arr[..., value_cell]`, local.py:1089-1135).  A program is C source (compiled with hiprtc) that defines
 *     PDEHIP_BC_FN void bc_face(int face, double value, double dx, double c0, double c1, double c2, double t, double *A, double *B)
 * (c0..c2 = coordinates of the wall point along the grid axes; value = 0 for faces that do not read the field), plus one
 * descriptor per face: where A and B go and how the face cell (i1, i2) maps to coordinates.  pdehip_bcprog_run evaluates all
 * faces of the program for time t - and the field `state_full` - in ONE launch; the stencil kernels and the ghost kernel then
 * read the arrays as usual. */
typedef struct pdehip_bcprog_face {
    double *const_arr, *factor_arr;   /* fp64 device arrays of m1 * m2 face cells (C order), written by the program */
    int64_t m1, m2;                   /* face extents along the remaining grid axes in grid order (1 where there is none) */
    double origin[3], step[3];        /* per coordinate k: index[k] == 0: c[k] = origin[k] (the wall, or an unused slot);        */
    int32_t index[3];                 /*   index[k] == 1 / 2: c[k] = (first[k] + i1 / i2 + 0.5) * step[k] + origin[k] - cell centres of */
    int32_t reads_value;              /*   the reference (discretize_interval, pde/grids/base.py:88-113), operation by operation */
    double dx;                        /* spacing normal to the wall */
    int32_t axis;                     /* reads_value: grid axis normal to the face, ...                                      */
    int32_t component;                /*   ... component of the field (0 for a scalar field) and ...                          */
    int64_t value_index;              /*   ... valid index along `axis` of the cells whose values are handed to bc_face        */
    int64_t first[3];                 /* per coordinate k with index[k] != 0: global index of the face's first cell - the face of a slab / block of a
                                         decomposed grid evaluates c[k] = ((first[k] + i) + 0.5) * step[k] + origin[k] with the bounds of the WHOLE grid,
                                         bit-identical to the undecomposed run (0 for a whole grid)                              */
} pdehip_bcprog_face_t;
/* `grid`: layout of the fields handed to pdehip_bcprog_run (NULL when no face reads the field) */
int pdehip_bcprog_create(const char *source, int nfaces, const pdehip_bcprog_face_t *faces, const pdehip_grid_t *grid, void **handle);
/* `state_full`: the field the conditions are about to be applied to (may be NULL when no face reads it) */
int pdehip_bcprog_run(void *handle, double t, const void *state_full, void *stream);
int pdehip_bcprog_destroy(void *handle);

/* ---- runtime ------------------------------------------------------------------ */
const char *pdehip_last_error(void);
/* name of the stencil-kernel instance the calling thread launched last (template name with its tile shape and switches), "" before the first
 * launch; ABI version 7.  Measurement aid: bench.py labels its roofline with the instance that ran.  (No counterpart in the reference.) */
const char *pdehip_last_kernel_name(void);
/* Arithmetic mode of the stencil kernels (Laplacian / gradient / divergence, the fused right-hand sides and time steps, ghost cells, run-time
 * compiled expression kernels), process-wide; ABI version 7.  0 (default): compiled with -ffp-contract=off - every rounding of the reference's
 * expression order, results bit-identical to its numpy / torch-CPU evaluation.  1: the same kernels compiled with FMA contraction - what
 * numba's default `fastmath` (pde/backends/numba/utils.py:330-336; config `backend.numba.fastmath`, pde/backends/numba/config.py:20-26) allows
 * the reference's own numba backend; results agree with the exact build within 1e-10 relative (tests/test_hip_fastmath.py), ~10-20 % fewer
 * fp64 operations per cell.  Set it before building steppers: kernels of expression PDEs are compiled for the mode current at their first use. */
int pdehip_set_fastmath(int on);
int pdehip_get_fastmath(int *on);
int pdehip_abi_version(void);
int pdehip_device_count(int *count);
int pdehip_set_device(int device);
int pdehip_device_name(char *buf, size_t len);
/* memory: replaces numpy allocation in NumbaBackend.make_operator
 * (pde/backends/numba/backend.py:460-517) and TorchBackend.numpy_to_native
 * (pde/backends/torch/backend.py:217-238); memory is zero-initialised */
int pdehip_malloc(void **ptr, size_t bytes);
int pdehip_free(void *ptr);
int pdehip_memset(void *ptr, int value, size_t bytes, void *stream);
int pdehip_memcpy_h2d(void *dst, const void *src_host, size_t bytes, void *stream);
int pdehip_memcpy_d2h(void *dst_host, const void *src, size_t bytes, void *stream);
int pdehip_memcpy_d2d(void *dst, const void *src, size_t bytes, void *stream);
/* the same by a kernel with streaming stores (16-byte aligned pointers, a multiple of 16 bytes): the fastest copy order on MI355X, the
 * yardstick bench.py prices the stencil kernels against (no reference counterpart: numpy copies on the host) */
int pdehip_copy_nt(void *dst, const void *src, size_t bytes, void *stream);
int pdehip_stream_create(void **stream);
int pdehip_stream_destroy(void *stream);
int pdehip_stream_synchronize(void *stream);
int pdehip_stream_wait_event(void *stream, void *event);
int pdehip_event_create(void **event);
int pdehip_event_destroy(void *event);
int pdehip_event_record(void *event, void *stream);
int pdehip_event_synchronize(void *event);
int pdehip_event_elapsed_ms(void *start, void *stop, float *ms);

/* Device layout of a full array of this grid.  out8 = { pitch of normalised axis 0, pitch of
 * normalised axis 1, elements per component, offset of interior cell (0,..,0), column of the
 * first interior cell in a row, elements to allocate for ONE component incl. tail slack,
 * tail slack, pitch (elements) of one layer along the grid's axis 0 }.  A device full array
 * of `ncomp` components needs (ncomp * out8[2] + out8[6]) elements. */
int pdehip_layout(const pdehip_grid_t *g, int64_t *out8);

/* ---- data movement between the valid and the full layout ----------------------
 * replaces NumpyBackend.make_valid_data_setter (pde/backends/numpy/backend.py:72-115) */
int pdehip_valid_to_full(const pdehip_grid_t *g, int ncomp, const void *valid, void *full,
                         void *stream);
int pdehip_full_to_valid(const pdehip_grid_t *g, int ncomp, const void *full, void *valid,
                         void *stream);
/* The same between HOST memory and a device full array, complete on return: the valid data of a field as the
 * reference holds it — `field.data` is a strided view of the ghost-padded host array (pde/fields/base.py:116-160,
 * pde/grids/base.py:329-337) — uploaded from / downloaded into that view without a contiguous host copy.
 * host_strides[4] (bytes): { between components, along the grid's axes right-aligned to three (entries of axes the
 * grid does not have are ignored) }; the fastest axis must be contiguous (host_strides[3] == element size, else
 * PDEHIP_E_VALUE).  Contiguous host memory: one hipMemcpy; windows: rows gathered / scattered through pinned chunks
 * while the DMA engine works, large fields on several threads.  Replaces the host side of TorchBackend.numpy_to_native /
 * native_to_numpy (pde/backends/torch/backend.py:217-260). */
int pdehip_upload_valid(const pdehip_grid_t *g, int ncomp, const void *host, const int64_t *host_strides,
                        void *full, void *stream);
int pdehip_download_valid(const pdehip_grid_t *g, int ncomp, const void *full, void *host,
                          const int64_t *host_strides, void *stream);
/* `hostfull` = a device copy of the reference's compact host full array (shape + 2 per axis,
 * `field._data_full`, pde/fields/datafield_base.py:93-126), ghost cells included */
int pdehip_hostfull_to_full(const pdehip_grid_t *g, int ncomp, const void *hostfull_dev,
                            void *full, void *stream);
int pdehip_full_to_hostfull(const pdehip_grid_t *g, int ncomp, const void *full,
                            void *hostfull_dev, void *stream);

/* ---- ghost cells ---------------------------------------------------------------
 * replaces NumbaBackend.make_ghost_cell_setter (pde/backends/numba/backend.py:184-404) */
int pdehip_set_ghost_cells(const pdehip_grid_t *g, int ncomp, const pdehip_bc_face_t *faces,
                           void *data_full, void *stream);

/* ---- operators (no BCs: the caller guarantees ghost cells are set) ---------------
 * replace the closures returned by make_laplace / make_gradient / make_divergence
 * (pde/backends/numba/operators/cartesian.py:332-383, :553-587, :962-996), signature
 * (arr_full, out_valid) of BackendBase.make_operator_no_bc (pde/backends/base.py:482-521).
 * `out_layout` selects whether `out` is a valid or a full array (the interior of a full
 * array is written, its ghost cells are left untouched). */
int pdehip_laplace(const pdehip_grid_t *g, const void *in_full, void *out, int out_layout,
                   void *stream);
/* The SPECTRAL Laplacian of periodic 1-D / 2-D grids: out = ifft(factor * fft(in)).real with factor = -(2 pi fftfreq)^2 summed over the
 * axes - `_make_laplace_numba_spectral_1d` / `_2d`, pde/backends/numba/operators/cartesian.py:232-330, chosen by `spectral=True` /
 * `use_spectral` (:359-372).  The transform is hipFFT's (loaded at run time); ghost cells of in_full are not read (every axis is
 * periodic: the caller checks).  3-D: error "not implemented", like the reference (:369-370). */
int pdehip_laplace_spectral(const pdehip_grid_t *g, const void *in_full, void *out, int out_layout, void *stream);
int pdehip_gradient(const pdehip_grid_t *g, int method, const void *in_full, void *out,
                    int out_layout, void *stream);
int pdehip_divergence(const pdehip_grid_t *g, int method, const void *in_full, void *out,
                      int out_layout, void *stream);
/* sum_a (d_a f)^2, pde/backends/numba/operators/cartesian.py:590-809; central != 0 selects
 * the central-difference form */
int pdehip_gradient_squared(const pdehip_grid_t *g, int central, const void *in_full, void *out,
                            int out_layout, void *stream);

/* 2-D Laplacian with the nine-point stencil (pde/backends/numba/operators/cartesian.py:153-190; `corner_weight` w: 1/2
 * Oono-Puri, 1/3 Mehrstellen; w = 0 is pdehip_laplace).  The four CORNER ghost cells of `in_full` are written first, as the
 * reference's make_corner_point_setter_2d does (:36-78: copies across a periodic axis — periodic2 = grid.periodic —, else
 * the mean of the two adjacent face ghosts); the face ghost cells must be set by the caller.  This is the one operator
 * that writes its input (SURVEY.md 8b "ownership"). */
int pdehip_laplace9(const pdehip_grid_t *g, const int *periodic2, double corner_weight, void *in_full, void *out,
                    int out_layout, void *stream);

/* first (order = 1, `method` as above) or second (order = 2, central) derivative along ONE axis:
 * the `d_dx`, `d_dy_forward`, `d2_dx2`, ... pattern operators of NumbaBackend.get_operator_info
 * (pde/backends/numba/backend.py:119-182), formulas of make_derivative / make_derivative2
 * (pde/backends/numba/operators/common.py:19-193): (r - l) / (2 dx), (r - c) / dx, (c - l) / dx,
 * (r - 2 c + l) * (1 / dx**2) */
int pdehip_axis_derivative(const pdehip_grid_t *g, int axis, int order, int method, const void *in_full,
                           void *out, int out_layout, void *stream);

/* ---- fused stencil + pointwise kernels (all arrays full) --------------------------
 * out = s2 * (s1 * laplace(in))                       [k = dt * (D * lap), RK stages] */
int pdehip_laplace_scaled(const pdehip_grid_t *g, const void *in_full, void *out_full, double s1,
                          double s2, void *stream);
/* out = y + s2 * (s1 * laplace(in)); y may alias in    [Euler: pde/solvers/euler.py:172-175] */
int pdehip_laplace_euler(const pdehip_grid_t *g, const void *in_full, const void *y_full,
                         void *out_full, double s1, double s2, void *stream);
/* TWO explicit Euler steps of the diffusion equation in one sweep (temporal blocking):
 *     out = E(E(in)),  E(u) = u + dt * (D * laplace(u)) with the BCs `faces` applied to u
 * i.e. two iterations of the loop body of pde/backends/numba/_solvers.py:98-108 with
 * pde/solvers/euler.py:172-175 and pde/pdes/diffusion.py:119-121, bit-identical to two single steps;
 * the intermediate level lives in registers only, ghost cells of `in` are neither read nor written.
 * Covers 3-D grids whose fastest axis is a multiple of 64 x 16 bytes, an even number of rows, >= 4 cells
 * per axis, and faces that are periodic or scalar first-order (Dirichlet / Neumann / mixed) per axis pair.
 * *done = 0 and nothing is written when the case is not covered (pdehip_euler_run then falls back to
 * single steps by itself). */
int pdehip_diffusion_euler2(const pdehip_grid_t *g, const pdehip_bc_face_t *faces, const void *in_full,
                            void *out_full, double diffusivity, double dt, int *done, void *stream);
/* `nsteps` (1..8 for the diffusion equation, 1..4 for Cahn-Hilliard) explicit Euler steps of a 2-D grid in ONE launch: a
 * workgroup keeps a tile plus halo in LDS and advances it through all time levels (temporal blocking), BCs applied to every
 * level — bit-identical to `nsteps` iterations of the loop body of pde/backends/numba/_solvers.py:98-108 (euler.py:172-175 with
 * diffusion.py:119-121 / cahn_hilliard.py:115-122).  For grids of a few MB (BASELINE configs 1-3), where a step per launch is
 * bound by launch and cache latency, not by HBM.  *done = 0 (nothing written) when the grid is not 2-D, a face is not periodic
 * / scalar first-order with the virtual point from the adjacent cell, or `nsteps` is out of range; pdehip_euler_run uses it by
 * itself and falls back to the register-pipelined kernels. */
int pdehip_euler_multi_2d(const pdehip_grid_t *g, const pdehip_rhs_t *rhs, const void *in_full, void *out_full, double dt,
                          int nsteps, int *done, void *stream);
/* The same two steps on a sub-slab of layers of a larger array (building block of pdehip_slab_euler2_run): `g_sub` has
 * the extent of the sub-slab along axis 0, `in_full` / `out_full` point ONE LAYER BEFORE its first layer, and the array
 * holds TWO real layers beyond the sub-slab on the sides named by `halo_sides` (1 = both sides, 2 = upper side only, the
 * lower face is the physical face `faces[0]`; 3 = lower side only, upper face `faces[1]`, index1 = extent - 1).  Replaces
 * the _MPIBC virtual points of pde/grids/boundaries/local.py:561-662 for both time levels. */
int pdehip_diffusion_euler2_slab(const pdehip_grid_t *g_sub, const pdehip_bc_face_t *faces, const void *in_full,
                                 void *out_full, double diffusivity, double dt, int halo_sides, int *done, void *stream);
/* ONE sweep for the Cahn-Hilliard right-hand side (pde/pdes/cahn_hilliard.py:115-122): mu = c^3 - c - gamma*laplace(c)
 * with the faces of c, then laplace(mu) with the faces of mu; mu lives in registers only (same two-level kernel as
 * pdehip_diffusion_euler2, bit-identical to pdehip_cahn_hilliard_mu + pdehip_laplace_euler / _scaled):
 *     euler != 0: out = c + dt * laplace(mu)   (one explicit Euler step)      euler == 0: out = dt * laplace(mu)
 * *done = 0 and nothing written when grid / faces are not covered (same rules as pdehip_diffusion_euler2, and the
 * faces of c and mu must be periodic on the same axes); pdehip_rhs_scaled / pdehip_euler_run use it by themselves. */
int pdehip_cahn_hilliard_fused(const pdehip_grid_t *g, const pdehip_bc_face_t *faces_c,
                               const pdehip_bc_face_t *faces_mu, const void *c_full, void *out_full, double gamma,
                               double dt, int euler, int *done, void *stream);
/* mu = c*c*c - c - gamma * laplace(c)                  [pde/pdes/cahn_hilliard.py:116-120] */
int pdehip_cahn_hilliard_mu(const pdehip_grid_t *g, const void *c_full, void *mu_full,
                            double gamma, void *stream);

/* ---- pointwise helpers over the interior of full arrays ----------------------------
 * out = y + sum_j coef[j] * k[j], evaluated left to right (pde/solvers/runge_kutta.py:135-153);
 * n <= 6; y == NULL drops the leading term (then out = coef[0]*k[0] + ...) */
int pdehip_lincomb(const pdehip_grid_t *g, int ncomp, void *out_full, const void *y_full, int n,
                   const double *coef_host, const void *const *k_full_host, void *stream);
/* y += (k1 + 2*k2 + 2*k3 + k4) / 6                     [pde/solvers/runge_kutta.py:60] */
int pdehip_rk4_combine(const pdehip_grid_t *g, int ncomp, void *y_full, const void *k1,
                       const void *k2, const void *k3, const void *k4, void *stream);
/* y += dt * (1.5 * rate_cur - 0.5 * rate_prev)        [Adams-Bashforth: pde/solvers/adams_bashforth.py:41-44,
 * pde/backends/numba/_solvers.py:160-167; the rates are unscaled right-hand sides = pdehip_rhs_scaled with dt = 1] */
int pdehip_ab2_combine(const pdehip_grid_t *g, int ncomp, void *y_full, const void *rate_cur,
                       const void *rate_prev, double dt, void *stream);
/* RKF45 tail (pde/solvers/runge_kutta.py:146-152):
 *   err  = max | r1 k1 + r3 k3 + r4 k4 + r5 k5 + r6 k6 |   -> *err_dev (fp64 device scalar)
 *   ynew = y + c1 k1 + c3 k3 + c4 k4 + c5 k5
 * *err_dev is reset by the call; NaN propagates like numpy's max. */
int pdehip_rkf45_combine(const pdehip_grid_t *g, int ncomp, const void *y, void *ynew,
                         const void *const *k6_host, double *err_dev, void *stream);
/* max |a - b| over the interior -> *out_dev (generic error estimate pde/solvers/base.py:416) */
/* *out_dev = max |z| over complex data held as pairs of real components of the full array (component 2p = real part, 2p + 1 =
 * imaginary part): the error norm `np.abs(error).max()` of the adaptive schemes when the state is complex
 * (pde/solvers/runge_kutta.py:147-148, pde/solvers/euler.py:253; complex states: pde/solvers/controller.py:430-432).  NaN wins. */
int pdehip_max_abs_pairs(const pdehip_grid_t *g, int npairs, const void *arr_full, double *out_dev, void *stream);
/* End of an adaptive Euler attempt, pde/backends/numba/_solvers.py:381-394 (numpy twin pde/solvers/euler.py:238-256):
 *   out = half + k                       `step_small += 0.5 * dt * rate_midpoint`   (k = that product, half = state + dt/2 * rate)
 *   *err_dev = max |(y + dt * rate) - out|      `np.abs(step_large - step_small).max()`, step_large is never stored
 * The loops below compute the same inside the sweep that produces k (stage kind 4). */
int pdehip_euler_adaptive_combine(const pdehip_grid_t *g, int ncomp, const void *y_full, const void *rate_full, double dt,
                                  const void *half_full, const void *k_full, void *out_full, double *err_dev, void *stream);
int pdehip_max_abs_diff(const pdehip_grid_t *g, int ncomp, const void *a_full, const void *b_full,
                        double *out_dev, void *stream);

/* y += scale * xi over the interior, xi ~ N(0, 1) independent per cell: the noise increment of an Euler-Maruyama step
 * (pde/solvers/euler.py:66-147: `state += sqrt(dt) * sqrt(noise_variance / cell_volume) * dW`; the reference draws dW with
 * numba's / torch's generator, pde/backends/numba/backend.py make_gaussian_noise, pde/backends/torch/backend.py:603-625).
 * Counter-based generator: xi of cell q (C-order index of the VALID array, component-major) in call number `counter` is
 * Box-Muller(Philox4x32-10(key = seed, counter = {q_lo, q_hi, counter_lo, counter_hi})) — reproducible for a given seed, independent of the
 * launch geometry and of how the grid is split over devices (pass the global cell offset of the slab in `cell_offset`). */
int pdehip_add_gaussian_noise(const pdehip_grid_t *g, int ncomp, void *y_full, double scale, uint64_t seed, uint64_t counter,
                              uint64_t cell_offset, void *stream);

/* out_dev[c] = cell_volume * sum over the interior of component c (fp64 device scalars, one per component): the integral of
 * a field on a Cartesian grid (uniform cell volumes) — NumbaBackend.make_integrator (pde/backends/numba/backend.py:555-652),
 * used by conservation checks and post-step logic without moving the field to the host.  Two passes (per-workgroup partial
 * sums, then one workgroup): the result does not depend on scheduling; it differs from the reference's sequential sum by
 * the usual reordering error (relative 1e-13 at 512^3). */
int pdehip_integrate(const pdehip_grid_t *g, int ncomp, const void *arr_full, double cell_volume, double *out_dev, void *stream);
/* out_dev[c] = number of cells of component c that are NaN or +-inf (as a double): the check of the reference's
 * ConsistencyTracker (`np.all(np.isfinite(field.data))`, pde/trackers/trackers.py:974-1003) without moving the field to the host -
 * 8 bytes per component cross PCIe instead of the whole state at every interrupt. */
int pdehip_count_nonfinite(const pdehip_grid_t *g, int ncomp, const void *arr_full, double *out_dev, void *stream);

/* ---- fused time steppers ----------------------------------------------------------
 * k_out = dt * rhs(y): applies the BCs of y (and of mu) — on the fly inside the stencil kernel where the
 * faces allow it, else by setting the ghost cells in place — then evaluates the RHS;
 * replaces NumbaBackend.make_pde_rhs + the `dt * rhs(...)` temporaries
 * (pde/backends/numba/backend.py:1158-1196, pde/solvers/runge_kutta.py:52-59) */
int pdehip_rhs_scaled(const pdehip_grid_t *g, const pdehip_rhs_t *rhs, void *y_full,
                      void *k_out_full, double dt, void *stream);
/* `nsteps` explicit Euler steps with fixed dt, ping-ponging between buf_a (initial state)
 * and buf_b; *result receives the buffer holding the final state (which of the two depends on the number
 * of sweeps: the diffusion RHS advances two steps per sweep where pdehip_diffusion_euler2 covers the grid).
 * Replaces the jitted
 * fixed-step loop (pde/backends/numba/_solvers.py:93-104) around
 * EulerSolver._make_single_step_fixed_dt (pde/solvers/euler.py:149-179). */
int pdehip_euler_run(const pdehip_grid_t *g, const pdehip_rhs_t *rhs, void *buf_a, void *buf_b,
                     double dt, int64_t nsteps, void **result, void *stream);
/* one classical RK4 step in place on y (pde/solvers/runge_kutta.py:29-66);
 * work = 5 full arrays (scratch: slopes and stage inputs; their contents after the call are unspecified).
 * Where the stencil kernels cover grid and faces every stage is ONE sweep: the slope and, from the slope still in
 * registers, the input of the next stage / the new state (17 instead of 23 arrays moved per step); the result is
 * bit-identical to the sequence rhs_scaled + lincomb + rk4_combine, which remains the fallback. */
int pdehip_rk4_step(const pdehip_grid_t *g, const pdehip_rhs_t *rhs, void *y_full,
                    void *const *work5_host, double dt, void *stream);
/* One Adams-Bashforth step (pde/solvers/adams_bashforth.py:40-47) in ONE sweep: rate_cur = rhs(y_in) (kept for the
 * next step) and y_out = y_in + dt * (1.5 * rate_cur - 0.5 * rate_prev), bit-identical to pdehip_rhs_scaled(dt = 1) +
 * pdehip_ab2_combine.  *fused = 0 and nothing launched where the stencil kernels do not carry the stage epilogue
 * (1-D, odd rows, faces the two-level kernel does not cover for Cahn-Hilliard): the caller runs those two. */
int pdehip_ab2_step(const pdehip_grid_t *g, const pdehip_rhs_t *rhs, void *y_in_full, void *y_out_full,
                    void *rate_cur_full, const void *rate_prev_full, double dt, int *fused, void *stream);
/* `nsteps` RK4 steps with fixed dt in place on y: the fixed-step loop (pde/backends/numba/_solvers.py:93-104) around
 * pdehip_rk4_step.  Small grids (launch-bound) replay a cached hipGraph of 8 steps, like pdehip_euler_run. */
int pdehip_rk4_run(const pdehip_grid_t *g, const pdehip_rhs_t *rhs, void *y_full, void *const *work5_host,
                   double dt, int64_t nsteps, void *stream);
/* one RKF45 attempt (pde/solvers/runge_kutta.py:68-156): ynew and *err_dev are produced,
 * y is unchanged apart from its ghost cells; work = 7 full arrays (scratch: k1..k6, tmp; contents unspecified after
 * the call, ynew doubles as a stage input until the last sweep).  Fused like pdehip_rk4_step: six sweeps, the last
 * one computes the new state and the max-norm of the error estimate from k6 in registers (36 instead of 45 arrays). */
int pdehip_rkf45_attempt(const pdehip_grid_t *g, const pdehip_rhs_t *rhs, void *y_full,
                         void *ynew_full, void *const *work7_host, double dt, double *err_dev,
                         void *stream);

/* ---- slab-parallel layer: one process per GPU, axis-0 slabs, RCCL over xGMI -----------------------
 * Replaces the reference's MPI path: face exchange inside every right-hand side
 * (pde/backends/numba_mpi/backend.py:30-194, pde/grids/boundaries/local.py:561-662), the MAX
 * all-reduce of the adaptive error (pde/backends/base.py:678-712) and the MPI stepper wrapper
 * (pde/solvers/explicit_mpi.py:133-226).  `librccl_path` names the librccl.so already loaded in the
 * process (RCCL is resolved with dlsym, not linked).  Rank 0 creates a 128-byte unique id, the host
 * distributes it, every rank calls pdehip_comm_create.  `lower` / `upper` are the neighbour ranks
 * along axis 0 (-1 = physical boundary). */
int pdehip_comm_unique_id(const char *librccl_path, void *id128);
int pdehip_comm_create(const char *librccl_path, const void *id128, int rank, int size, void **comm);
int pdehip_comm_destroy(void *comm);
/* What RCCL reports about the communicator (ABI version 7; measurement aid: the N > 1 bench line records that RCCL really spans N ranks on N
 * devices): out5 = {ncclCommCount, ncclCommUserRank, ncclCommCuDevice, ncclGetVersion, HIP device of the caller}, -1 where unavailable;
 * pci_bus_id (>= 16 bytes) = hipDeviceGetPCIBusId of that device.  Reference counterpart: MPI.COMM_WORLD.size / rank of pde/tools/mpi.py:40-61. */
int pdehip_comm_info(void *comm, int *out5, char *pci_bus_id, size_t n);
/* send valid layers 1 / n of the local slab to the neighbours, receive their layers into the ghost
 * layers n+1 / 0 (ncclSend/ncclRecv in one group, enqueued on `stream`) */
int pdehip_halo_exchange(void *comm, const pdehip_grid_t *g_local, void *buf_full, int lower, int upper,
                         void *stream);
/* in-place MAX over all ranks of one fp64 device scalar; NaN wins like numpy.max */
int pdehip_allreduce_max(void *comm, double *dev_scalar, void *stream);
/* `nsteps` Euler steps of the diffusion right-hand side on the local slab: the exchange of the new
 * boundary layers overlaps the interior kernel (second HIP stream); no host synchronisation */
int pdehip_slab_euler_run(void *comm, const pdehip_grid_t *g_local, const pdehip_rhs_t *rhs, int lower,
                          int upper, void *buf_a, void *buf_b, double dt, int64_t nsteps, void **result,
                          void *stream);
/* Same contract, two steps per sweep (see pdehip_diffusion_euler2): TWO layers per side are exchanged once
 * per two steps, so both the HBM traffic per step and the number of exchanges are halved.  The slab is worked
 * on in a private copy with two halo layers per side; the result is written back to `buf_a` (*result).
 * Preconditions the CALLER checks globally (all ranks must take the same path): both neighbours exist on every
 * rank (periodic slowest axis; replaces the _MPIBC exchange of pde/grids/boundaries/local.py:561-662 for both
 * levels), >= 4 local layers on every rank, and *ok != 0 from pdehip_slab_euler2_supported (grid shape, dtype
 * and the faces of the two other axes are covered by the kernel).  E_NOTIMPL otherwise. */
int pdehip_slab_euler2_supported(const pdehip_grid_t *g_local, const pdehip_rhs_t *rhs, int *ok);
/* Cahn-Hilliard right-hand side on a slab in ONE sweep (pdehip_cahn_hilliard_fused with real halo layers on the slowest
 * axis): two layers of c per side are exchanged, mu needs no exchange of its own (the reference exchanges c and mu,
 * one layer each, per evaluation).  `c_ext` / `out_ext` are slab arrays with TWO halo layers per side (the layout of a
 * slab of n+2 layers; own layers 2..n+1).  euler != 0: out = c + dt*laplace(mu); euler == 0: out = dt*laplace(mu).
 * Same global preconditions as pdehip_slab_euler2_run, with >= 2 own layers per rank and pdehip_slab_ch_supported. */
int pdehip_slab_ch_supported(const pdehip_grid_t *g_local, const pdehip_rhs_t *rhs, int *ok);
int pdehip_slab_ch_sweep(void *comm, const pdehip_grid_t *g_local, const pdehip_rhs_t *rhs, int lower, int upper,
                         void *c_ext, void *out_ext, double dt, int euler, void *stream);
int pdehip_slab_euler2_run(void *comm, const pdehip_grid_t *g_local, const pdehip_rhs_t *rhs, int lower,
                           int upper, void *buf_a, void *buf_b, double dt, int64_t nsteps, void **result,
                           void *stream);
/* Scratch buffers of the process-wide context that serves the slab / block loops WITHOUT a communicator (comm == NULL: one process, no
 * neighbours): freed here (after the runs that use them), allocated again on demand.  A communicator frees its own in pdehip_comm_destroy.
 * (ABI version 7; the reference has no counterpart: numpy arrays are garbage-collected.) */
int pdehip_release_scratch(void);
/* Communication-avoiding variant (ABI version 7): FOUR halo layers per side, ONE exchange per FOUR steps (two two-step sweeps; the first
 * computes the own layers and two more per exchanged side, the second the own layers).  The part of the first sweep that reads own cells only
 * runs while the exchange of the group before is in flight; the boundary part waits for it - all sweeps on ONE stream, the halo stream
 * carries ncclSend / ncclRecv only.  Same arithmetic, bit-identical results.  Replaces the blocking per-step exchange of
 * pde/solvers/explicit_mpi.py:133-226 + pde/backends/numba_mpi/backend.py:163-194.  Same contract and preconditions as
 * pdehip_slab_euler2_run, with >= 8 local layers on EVERY rank (the caller checks globally) and *ok != 0 from pdehip_slab_euler4_supported. */
int pdehip_slab_euler4_supported(const pdehip_grid_t *g_local, const pdehip_rhs_t *rhs, int *ok);
int pdehip_slab_euler4_run(void *comm, const pdehip_grid_t *g_local, const pdehip_rhs_t *rhs, int lower,
                           int upper, void *buf_a, void *buf_b, double dt, int64_t nsteps, void **result,
                           void *stream);

/* ---- slab-parallel Runge-Kutta and the adaptive loop (one C call per run: no host work per stage) ----------------
 * `flags` (PDEHIP_SLAB_*) select code paths that every rank must take alike; the caller ANDs the local answers of
 * pdehip_slab_flags_supported over all ranks (MIN all-reduce) before the first call.  Arrays that serve as stage inputs
 * (y, ynew, work[3] (k4), the last work array (tmp)) — simplest: ALL slab arrays — must be allocated with one spare layer
 * (pdehip_layout out8[7] elements) before and after, because the fused Cahn-Hilliard sweep reads TWO halo layers per side
 * (exchanged into that spare room).  `comm` may be NULL when lower == upper == -1 (single process, no exchange): the same
 * loops then serve the serial adaptive stepper. */
enum {
    PDEHIP_SLAB_FUSED_CH = 1,    /* Cahn-Hilliard right-hand side as one two-level sweep after ONE exchange of two layers of c */
    PDEHIP_SLAB_FUSED_STAGE = 2  /* Runge-Kutta combination inside the sweep that computes the slope */
};
int pdehip_slab_flags_supported(const pdehip_grid_t *g_local, const pdehip_rhs_t *rhs, int lower, int upper, int *flags);
/* k_out = dt * rhs(y) on the slab incl. the halo exchange(s): replaces NumbaMPIBackend's operator + _MPIBC exchange per
 * right-hand side (pde/backends/numba_mpi/backend.py:30-194) */
int pdehip_slab_rhs_scaled(void *comm, const pdehip_grid_t *g_local, const pdehip_rhs_t *rhs, int lower, int upper, int flags,
                           void *y_full, void *k_out_full, double dt, void *stream);
/* `nsteps` Euler steps of ANY fused right-hand side, one exchange + sweep per step (Cahn-Hilliard; the diffusion equation has
 * the overlapped loops pdehip_slab_euler_run / pdehip_slab_euler2_run) */
int pdehip_slab_euler_sweeps(void *comm, const pdehip_grid_t *g_local, const pdehip_rhs_t *rhs, int lower, int upper, int flags,
                             void *buf_a, void *buf_b, double dt, int64_t nsteps, void **result, void *stream);
/* `nsteps` classical RK4 steps in place on y (pde/solvers/runge_kutta.py:29-66, loop pde/backends/numba/_solvers.py:93-104) */
int pdehip_slab_rk4_run(void *comm, const pdehip_grid_t *g_local, const pdehip_rhs_t *rhs, int lower, int upper, int flags,
                        void *y_full, void *const *work5_host, double dt, int64_t nsteps, void *stream);
/* Adaptive RKF45 from t_start to t_end: the loop of pde/backends/numba/_solvers.py:249-281 with the controller of
 * pde/solvers/base.py:572-592 and the MAX all-reduce of the error estimate (pde/backends/base.py:678-712).  In: t_start, t_end,
 * dt (first trial step), tolerance, dt_min, dt_max; the counters and statistics are ACCUMULATED (zero them before the first
 * call of a run).  Out: dt (next trial step), t_last, steps, attempts, statistics of the accepted steps.  *result = y or ynew,
 * whichever holds the final state.  A step size below dt_min is reported as an error (code 3) with the reference's message. */
typedef struct pdehip_adaptive {
    double t_start, t_end, dt, tolerance, dt_min, dt_max;
    double t_last;
    int64_t steps, attempts;
    int64_t stat_count;
    double stat_min, stat_max, stat_mean, stat_m2;
} pdehip_adaptive_t;
int pdehip_slab_rkf45_run(void *comm, const pdehip_grid_t *g_local, const pdehip_rhs_t *rhs, int lower, int upper, int flags,
                          void *y_full, void *ynew_full, void *const *work7_host, double *err_dev, pdehip_adaptive_t *ctl,
                          void **result, void *stream);

/* ADAPTIVE EULER from t_start to t_end - the reference's own loop `_make_adaptive_stepper_euler`, pde/backends/numba/_solvers.py:322-466
 * (numpy twin pde/solvers/euler.py:181-283), NOT the generic one-step-versus-two-half-steps estimate: the rate of the current state
 * is carried from attempt to attempt (a rejected attempt re-uses it), and after an accepted attempt the new rate is evaluated at the
 * time BEFORE `t += dt` (:402-407) - with time-dependent conditions or explicit time in the equation results and step counts depend
 * on that.  Control block and error handling as pdehip_slab_rkf45_run; work3 = rate, half step, slope scratch (slab arrays: the
 * half step is a sweep input).  Two sweeps per accepted attempt where the stage epilogues cover the grid (kinds 0 and 4). */
int pdehip_slab_euler_adaptive_run(void *comm, const pdehip_grid_t *g_local, const pdehip_rhs_t *rhs, int lower, int upper, int flags,
                                   void *y_full, void *ynew_full, void *const *work3_host, double *err_dev, pdehip_adaptive_t *ctl,
                                   void **result, void *stream);
/* the same on one device (no communicator, flags decided here) */
int pdehip_euler_adaptive_run(const pdehip_grid_t *g, const pdehip_rhs_t *rhs, void *y_full, void *ynew_full, void *const *work3_host,
                              double *err_dev, pdehip_adaptive_t *ctl, void **result, void *stream);

/* ---- BLOCK decomposition: one process per GPU owns a box of the grid (e.g. 2 x 2 x 2 for 512^3 on 8 GPUs) -----------------------
 * Replaces GridMesh with a multi-axis decomposition (pde/grids/_mesh.py:59-93 `_get_optimal_decomposition`, :401-444 neighbours) and the
 * face exchange `_MPIBC` of every decomposed axis (pde/grids/boundaries/local.py:561-662).  nb6[2 * axis + side] = rank of the neighbour
 * across that face of the local block, or -1 for a physical face (whose condition stays in the face tables of `rhs`; periodic axes
 * with ONE block keep their periodic condition).  One ghost layer per face; faces normal to the fast axes travel through packed
 * staging buffers.  Why: per xGMI link a 256^3 block moves 0.5 MiB per face where a 64 x 512 x 512 slab moves 2 MiB (xGMI is point
 * to point), see csrc/pdehip_block_loops.h. */
/* fill the ghost layers of buf_full on all faces that have a neighbour (one RCCL group) */
int pdehip_block_exchange(void *comm, const pdehip_grid_t *g_local, const int *nb6, void *buf_full, void *stream);
/* time loops on a block, ONE call per run: scheme 0 = `nsteps` explicit Euler steps (y_full / ynew_full ping-pong, *result names
 * the final one), 1 = `nsteps` RK4 steps in place (work: k1..k4, tmp), 2 = the adaptive RKF45 loop `ctl` incl. the MAX all-reduce
 * of the error (work: k1..k6, tmp), 3 = the reference's adaptive Euler loop `ctl` (pdehip_slab_euler_adaptive_run; work: rate, half
 * step, slope scratch).  Every right-hand side exchanges the faces of its input first (Cahn-Hilliard: c, then mu - the
 * reference's sequence); fuse_stage != 0: diffusion stages carry their Runge-Kutta combination (decide it for ALL ranks alike).
 * The time of the first step is rhs->t. */
int pdehip_block_run(void *comm, const pdehip_grid_t *g_local, const pdehip_rhs_t *rhs, const int *nb6, int fuse_stage, int scheme,
                     void *y_full, void *ynew_full, void *const *work_host, double *err_dev, double dt, int64_t nsteps,
                     pdehip_adaptive_t *ctl, void **result, void *stream);

/* ---- products of tensor fields, cell by cell (ABI version 6) ------------------------------------------------------------------------
 * Replaces BackendBase.make_inner_prod_operator / make_outer_prod_operator (pde/backends/base.py:567-610; numpy: np.einsum per rank
 * combination, pde/backends/numpy/backend.py:285-363; numba: pde/backends/numba/backend.py:654-893), which `DataFieldBase.dot` /
 * `VectorField.outer_product` / `make_dot_operator` use (pde/fields/datafield_base.py:965).  kind 0: vector . vector -> scalar, 1: tensor
 * . vector -> vector, 2: vector . tensor -> vector, 3: tensor . tensor -> tensor, 4: outer product of two vectors -> tensor; the vector
 * dimension is g->ndim; all arrays FULL, component-major (tensor entries in C order).  complex_pairs != 0: every tensor entry is a
 * planar (re, im) pair (the layout of complex states, pde_hip/complex_expr.py); conjugate != 0: the second operand is conjugated
 * (the reference's default for `dot`). */
int pdehip_field_product(const pdehip_grid_t *g, int kind, int complex_pairs, int conjugate, const void *a_full, const void *b_full,
                         void *out_full, void *stream);

/* ---- the FAST block decomposition: two Euler steps per sweep on a box, halos two layers deep incl. the edges, ONE message per
 * neighbouring rank, the exchange hidden behind the next sweep (csrc/pdehip_block2_loops.h; ABI version 6) -------------------------
 * Replaces the same reference code as pdehip_block_run (GridMesh pde/grids/_mesh.py:59-93, :401-444; the blocking face exchange inside
 * every right-hand side pde/backends/numba_mpi/backend.py:163-194, pde/grids/boundaries/local.py:561-662) for the case the benchmark
 * runs: DiffusionPDE, fixed-step Euler, a 3-D grid that is periodic on every axis.  dims3 / coords3: blocks per axis and the block of
 * this rank (ranks in C order of the block indices, like GridMesh); cut3[a] != 0: axis a is exchanged (several blocks - or one block
 * that sends its halo to itself: the probe of the exchange path on one device), else it wraps inside the kernels.  A cut of the fastest
 * axis is covered for fp64 (its halo cells live in the padding of the rows); where pdehip_block2_supported answers 0 the caller takes
 * pdehip_block_run.  buf_a: the state (full array of
 * g_local), advanced in place by `nsteps` (EVEN) steps; buf_b is not touched (kept for the signature of the other loops).
 * Bit-identical to single steps. */
int pdehip_block2_supported(const pdehip_grid_t *g_local, const pdehip_rhs_t *rhs, const int *cut3, int *ok);
int pdehip_block2_euler_run(void *comm, const pdehip_grid_t *g_local, const pdehip_rhs_t *rhs, const int *dims3, const int *coords3,
                            const int *cut3, void *buf_a, void *buf_b, double dt, int64_t nsteps, void **result, void *stream);

/* ---- run-time specialised right-hand sides (generic `PDE({...})` expressions) ----------------------
 * Replaces the sympy -> numba code generation of pde/pdes/pde.py:401-499 / pde/tools/expressions.py:
 * 361-388.  `epilogue_body` is the body of
 *     double pde_epilogue(double c, double lap, double gsq, double e0, double e1, double e2, const double *p)
 * (C statements ending in `return ...;`).  pdehip_jit_apply evaluates, for every interior cell,
 *     out = pde_epilogue(in, laplace(in), gradient_squared(in), extra[0], extra[1], extra[2], params)
 * in ONE pass of the register-pipelined stencil kernel compiled around the epilogue with hiprtc
 * (cached per tile shape); `in_faces` are evaluated on the fly / by the ghost kernel first (NULL: ghost
 * cells of `in_full` are already set).  All arrays are FULL arrays of the same grid. */
int pdehip_jit_create(const char *epilogue_body, void **handle);
int pdehip_jit_destroy(void *handle);
/* compile (not load) the kernels of this expression for a dtype / dimension: works without a GPU */
int pdehip_jit_check(void *handle, int dtype, int ndim);
int pdehip_jit_apply(void *handle, const pdehip_grid_t *g, void *in_full, const void *const *extra3_host,
                     void *out_full, const double *params_host, int nparams, const pdehip_bc_face_t *in_faces,
                     void *stream);
/* pdehip_jit_apply whose result is the slope k of a Runge-Kutta stage (the epilogue computes dt * F), followed in the
 * SAME sweep by the pointwise combination of the scheme (pde/solvers/runge_kutta.py:52-61, :135-150):
 *   kind 0: k_out = k,  out2 = y + sum_m coef[m] * k_prev[m] + c_new * k      (input of the next stage)
 *   kind 1: out2 = y + (k_prev[0] + 2 k_prev[1] + 2 k_prev[2] + k) / 6        (RK4 update; k_out unused, out2 may be y)
 *   kind 2: out2 = y + c1 k1 + c3 k3 + c4 k4 + c5 k5 and *err_dev = max |error estimate| with k6 = k,
 *           k_prev = {k1, k3, k4, k5}                                         (end of an RKF45 attempt)
 *   kind 4: out2 = k_prev[1] + k and *err_dev = max |(y + coef[0] * k_prev[0]) - out2| with k_prev = {carried rate, half step},
 *           coef[0] = dt, k = dt/2 * F(half step)                             (end of an adaptive Euler attempt, pdehip_euler_adaptive_combine)
 * Same expressions in the same order as pdehip_lincomb / pdehip_rk4_combine / pdehip_rkf45_combine: bit-identical to
 * pdehip_jit_apply followed by them.  *done = 0 (nothing launched) when only the generic kernel covers the grid. */
int pdehip_jit_apply_stage(void *handle, const pdehip_grid_t *g, void *in_full, const void *const *extra3_host,
                           void *k_out_full, const double *params_host, int nparams, const pdehip_bc_face_t *in_faces,
                           int kind, const void *y_full, int nk, const void *const *k_prev_host, const double *coef_host,
                           double c_new, void *out2_full, double *err_dev, int *done, void *stream);
/* TWO applications in one sweep: out = f(f(in)), f(u) = pde_epilogue(u, laplace(u), gradient_squared(u); params), with
 * the BCs `faces` applied to u before each application and the intermediate level in registers — two explicit Euler
 * steps of a one-pass expression PDE (epilogue = the Euler update `u + dt * F(u, ...)`; no extra arrays, no explicit
 * time).  Replaces two iterations of the fixed-step loop pde/backends/numba/_solvers.py:98-108 around the compiled
 * right-hand side of pde/pdes/pde.py:401-499.  *done = 0 (nothing launched) when grid or faces are not covered: same
 * rules as pdehip_diffusion_euler2; the caller then calls pdehip_jit_apply twice. */
int pdehip_jit_euler2(void *handle, const pdehip_grid_t *g, const void *in_full, void *out_full,
                      const double *params_host, int nparams, const pdehip_bc_face_t *faces, int *done, void *stream);
/* Two-pass expressions in ONE sweep (e.g. `laplace(c**3 - c - laplace(c))`, Swift-Hohenberg): handle from
 * pdehip_jit_create2(body1, body2);  tmp = body1(u, laplace u, gradient_squared u; params) is kept in registers,
 *     out = body2(tmp, laplace tmp, gradient_squared tmp, e0 = u; params)
 * with `faces_u` applied to u and `faces_tmp` to tmp (the reference applies them when it evaluates the inner and the
 * outer operator: pde/pdes/pde.py:299-399).  *done = 0 when grid / faces are not covered: the caller then runs the two
 * passes through pdehip_jit_apply. */
int pdehip_jit_create2(const char *body1, const char *body2, void **handle);
int pdehip_jit_fused2(void *handle, const pdehip_grid_t *g, const void *in_full, void *out_full,
                      const double *params_host, int nparams, const pdehip_bc_face_t *faces_u,
                      const pdehip_bc_face_t *faces_tmp, int *done, void *stream);

/* The fixed-step explicit Euler loop of an expression PDE (one or several scalar components) in ONE call: `nsteps` times the
 * sequence of passes, each pdehip_jit_apply of its handle, the state ping-ponging between two buffers.  Replaces the loop
 * pde/backends/numba/_solvers.py:98-108 (`for i in range(steps): t = t_start + i * dt; single_step(state, t)`) around
 * pde/solvers/euler.py:172-175 for the compiled right-hand sides of pde/pdes/pde.py:401-499 - a Python loop over the passes
 * costs 40-85 us per step on small grids where the kernels need 2-5 us (profiles/r02_time_small_expr.md).
 * Array indices of a pass: >= 0 selects fixed[index] (temporaries, constant arrays, coordinates); -1 - k selects component k
 * of the CURRENT state as `src` / `extras` and component k of the NEXT state as `out`; PDEHIP_JIT_NONE: no array.  The
 * epilogue of the pass that writes the next state must be the Euler update `state + dt * F` (parameters p[0] = dt,
 * p[1] = t).  state_a holds the state on entry; *result is the buffer that holds it afterwards.  uses_time = 0 allows replaying a
 * captured hipGraph for long runs (PDEHIP_JIT_GRAPH=1; measured slower than the plain launches of this loop, so off by default). */
#define PDEHIP_JIT_NONE INT32_MIN
/* DECOMPOSED grids (one box per GPU): a pass whose `exchange` is set fills the ghost layers of its `src` from the neighbours before it
 * runs - pdehip_halo_exchange (slab: lower / upper) or pdehip_block_exchange (blocks: nb6) on the loop's stream - like `_MPIBC` inside
 * every operator of the reference's MPI path (pde/grids/boundaries/local.py:561-662, pde/solvers/explicit_mpi.py:133-226).  The caller
 * sets it on the FIRST pass of an evaluation that applies operators to an array (the operand of a nested operator is an intermediate
 * field: exchanged like the state); the faces towards neighbours are PDEHIP_BC_SKIP in `faces`.  The adaptive loops reduce their
 * error norm over all ranks of `comm` (MAX, NaN wins: pde/backends/base.py:678-712). */
typedef struct {
    void *comm;
    int32_t blocks;          /* 0: axis-0 slab (lower / upper), 1: block decomposition (nb6) */
    int32_t lower, upper;
    int32_t nb6[6];
} pdehip_exchange_t;
typedef struct {
    void *handle;
    int32_t src;
    int32_t extras[3];
    int32_t out;
    const pdehip_bc_face_t *faces;   /* conditions applied to src before the pass (NULL: none) */
    const pdehip_exchange_t *exchange;   /* decomposed grids: exchange the ghost layers of src first (NULL: nothing) */
} pdehip_jit_pass_t;
int pdehip_jit_euler_run(const pdehip_grid_t *g, const pdehip_jit_pass_t *passes, int npasses, void *const *fixed, int nfixed,
                         void *state_a, void *state_b, int ncomp, double dt, double t0, int uses_time, int64_t nsteps,
                         void *bc_program, void **result, void *stream);
/* Runge-Kutta loops of an expression PDE in ONE call: `nsteps` classical RK4 steps with fixed dt (ctl == NULL;
 * pde/solvers/runge_kutta.py:29-66 inside the loop pde/backends/numba/_solvers.py:98-108), or the ADAPTIVE Runge-Kutta-Fehlberg
 * loop from ctl->t_start to ctl->t_end (runge_kutta.py:68-156 inside pde/backends/numba/_solvers.py:249-281 with the controller
 * pde/solvers/base.py:572-592; the host reads 8 bytes per attempt, nothing per stage).  The passes are those of
 * pdehip_jit_euler_run with another meaning of the state indices: -1 - k as `src` / `extras` is component k of the STAGE INPUT,
 * as `out` component k of the SLOPE; the epilogue of the passes that write a slope must compute `dt * F` (p[0] = dt of the
 * step, p[1] = time of the stage).  work: 5 (RK4: k1..k4, tmp) or 7 (RKF45: k1..k6, tmp) arrays of ncomp components each; y
 * is advanced in place (RK4) or ping-pongs with ynew (adaptive: *result names the array holding the final state).  A single-
 * component expression whose last pass runs on the vectorised kernel takes the stage epilogue of pdehip_jit_apply_stage
 * (stage_fuse bit 0); everything else combines with pdehip_lincomb / pdehip_rk4_combine / pdehip_rkf45_combine: bit-identical.
 * stage_fuse bit 1 (adaptive loops only): the components are the planar (re, im) PAIRS of complex fields - the error norm is the
 * modulus, `np.abs(error).max()` of a complex array (runge_kutta.py:147-148, pde/solvers/euler.py:253): new state and an explicit error
 * field through pdehip_lincomb, then pdehip_max_abs_pairs; the error field is ONE MORE work array (work[7]; adaptive Euler: work3[3]).
 * bc_program (or NULL): time-dependent faces, refreshed for every stage time. */
int pdehip_jit_rk_run(const pdehip_grid_t *g, const pdehip_jit_pass_t *passes, int npasses, void *const *fixed, int nfixed,
                      int ncomp, void *y, void *ynew, void *const *work_host, double *err_dev, double dt, double t0, int64_t nsteps,
                      pdehip_adaptive_t *ctl, int stage_fuse, void *bc_program, void **result, void *stream);

/* The reference's adaptive Euler loop (pde/backends/numba/_solvers.py:322-466) for an expression PDE in ONE call: passes as in
 * pdehip_jit_rk_run (the slope passes compute `p[0] * F`; the carried rate is obtained with p[0] = 1), work3 = rate, half step, slope
 * scratch (ncomp components each).  stage_fuse != 0: single-component expressions take the stage epilogues (kinds 0 and 4 of
 * pdehip_jit_apply_stage) where the vectorised kernel covers the grid. */
int pdehip_jit_euler_adaptive_run(const pdehip_grid_t *g, const pdehip_jit_pass_t *passes, int npasses, void *const *fixed, int nfixed,
                                  int ncomp, void *y, void *ynew, void *const *work3_host, double *err_dev, pdehip_adaptive_t *ctl,
                                  int stage_fuse, void *bc_program, void **result, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* PDEHIP_H */
