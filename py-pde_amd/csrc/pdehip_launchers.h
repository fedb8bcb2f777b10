// pdehip_launchers.h - prototypes of the launchers of the stencil-kernel translation units (pdehip_kernels.hip, pdehip_kernels_e2.hip,
// pdehip_kernels_t2.hip).  NO include guard: pdehip_common.h includes it three times - in namespace pdehip (the dispatchers of
// pdehip_dispatch.hip, what everybody calls), in pdehip::exactv (the kernels compiled with -ffp-contract=off: bit-identical to the CPU oracle,
// the default) and in pdehip::fastv (the SAME sources compiled with -ffp-contract=fast: pdehip_set_fastmath, round 6).
int launch_laplace(const NGrid &n, const void *in, void *out, const OutStr &o, int mode, double s1,
                   double s2, double gamma, const void *y, hipStream_t st, const InputBCs *fg = nullptr,
                   const StageFuse *stage = nullptr);
int launch_deriv_march(const NGrid &n, const void *in, void *out, const OutStr &o, int mode, const double *gs, hipStream_t st, bool *done);
int launch_div_march(const NGrid &n, int method, const void *in, void *out, const OutStr &o, hipStream_t st, bool *done);
bool laplace_can_fuse_bcs(const NGrid &n, const void *in, const void *out, const void *y);
int launch_ghosts(const NGrid &n, int ncomp, const pdehip_bc_face_t *faces, void *data, hipStream_t st);
int preload_e2_kernels();        // pdehip_kernels_e2.hip
int preload_t2_kernels();        // pdehip_kernels_t2.hip
int preload_stencil_kernels();   // pdehip_kernels.hip: load the code object of the stencil kernels now (see pdehip_set_device)
int launch_euler2(const NGrid &n, const void *in, void *out, double s1, double s2, const InputBCs &fg, int xplain,
                  hipStream_t st, bool *done, bool dry_run = false, int ends = 0, int m2 = E2_DIFFUSION,
                  const InputBCs *fg1 = nullptr, double gamma = 0, Euler2Plan *plan = nullptr, const StageFuse *stage = nullptr,
                  int yzplain = 0);   // yzplain: bit 0 / bit 1 = the rows / the fastest axis have two real halo layers in memory (a box of a larger array)
// K Euler steps of a 2-D grid per launch, time levels in LDS (pdehip_tile2d.inc): diffusion (rhs->kind 0) or Cahn-Hilliard
int tile2d_max_steps(int mode);
int plan_tile2d(const NGrid &n, const void *in, void *out, int mode, double s1, double s2, double gamma, const InputBCs &fc,
                const InputBCs *fm, int nsteps, Tile2Args *args, unsigned *nblocks, int *tile_columns, bool *done);
int launch_tile2d(const NGrid &n, const void *in, void *out, int mode, double s1, double s2, double gamma, const InputBCs &fc,
                  const InputBCs *fm, int nsteps, hipStream_t st, bool *done);
bool force_generic_kernels();   // PDEHIP_FORCE_GENERIC=1
