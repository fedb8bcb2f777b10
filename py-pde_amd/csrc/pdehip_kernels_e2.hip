// pdehip_kernels_e2.hip - launch logic and offline instances of the two-level sweeps (pdehip_march2.inc): two Euler steps per sweep, the fused
// Cahn-Hilliard right-hand side, Runge-Kutta stage sweeps.  Split from pdehip_kernels.hip (round 6): its own translation unit and code object.
// Same compile flags (-ffp-contract=off: bit-identical to the CPU oracle).
#include "pdehip_common.h"

// Compiled TWICE (py-pde_amd/Makefile): as pdehip::exactv with -ffp-contract=off (bit-identical to the CPU oracle; the default) and, with
// -DPDEHIP_FAST_VARIANT -ffp-contract=fast, as pdehip::fastv (FMA contraction like numba's default fastmath, pde/backends/numba/utils.py:330-336;
// opt-in through pdehip_set_fastmath, results within 1e-10 of the exact build).  pdehip_dispatch.hip picks one per call.
#ifdef PDEHIP_FAST_VARIANT
#define PDEHIP_VARIANT_NS fastv
#else
#define PDEHIP_VARIANT_NS exactv
#endif
namespace pdehip {
namespace PDEHIP_VARIANT_NS {

#include "pdehip_march2.inc"

// ---------------------------------------------------------------------------------------------
// two Euler steps per sweep (pdehip_march2.inc).  *done = false when the grid / BCs are outside
// what the kernel covers; the caller then takes two single steps.
// ---------------------------------------------------------------------------------------------
struct Tune2 { int ry; long blocks; int off; bool set; int order; };
static const Tune2 &tune2()
{
    static Tune2 t = {0, 0, 0, false, -1};
    if (!t.set) {
        t.set = true;
        // PDEHIP_EULER2="ry,blocks,waves" tile rows / wave tiles per sweep / waves per workgroup (tuning aid),
        // PDEHIP_EULER2=off disables the kernel
        const char *e = getenv("PDEHIP_EULER2");
        if (e && !strcmp(e, "off")) t.off = 1;
        else if (e) sscanf(e, "%d,%ld,%d", &t.ry, &t.blocks, &t.order);
    }
    return t;
}

// fp32 tiles (fp32 storage, fp64 registers).  The wide tile - 4 cells per lane (16-byte accesses), 2 rows - is at 252 VGPRs
// without room for the stage epilogue (256 + 64 B of scratch with it).  The NARROW tile - 2 cells per lane (8-byte accesses),
// 4 rows, the shape of the fp64 tile - needs 194 VGPRs (204 with the stage epilogue) and recomputes 1.5 x instead of 2 x of
// the intermediate level.  PDEHIP_F32_TILE="vec,ry[,stage_vec,stage_ry]" overrides the choice (tuning aid).
struct TuneF32 { int vec, ry, svec, sry; };
static const TuneF32 &tune_f32()
{
    static TuneF32 t = {0, 0, 0, 0};
    static bool set = false;
    if (!set) {
        set = true;
        const char *e = getenv("PDEHIP_F32_TILE");
        if (e) sscanf(e, "%d,%d,%d,%d", &t.vec, &t.ry, &t.svec, &t.sry);
    }
    return t;
}

template <typename T, int VEC>
static int launch_euler2_tv(const NGrid &n, LapArgs a, int xplain, hipStream_t st, bool *done, bool dry_run, int ends, int m2,
                            Euler2Plan *plan, int ry_f32)
{
    constexpr int CW = 64 * VEC;
    const Tune2 &t2 = tune2();
    // fp64: 4-row tiles (226 VGPRs, 2 waves per SIMD); fp32: see tune_f32()
    const bool has_y = n.ndim == 3;   // 2-D: march along the first grid axis, a "plane" is one row (a.n1 == 1)
    int ry = (t2.ry && t2.ry != 8) ? t2.ry : 4;   // (8: the tall tile where it applies, see `tall`)
    if (sizeof(T) == 4) ry = ry_f32;
    // the tall tile (8 rows, one wave per SIMD, four plane buffers: pdehip_march2.inc): the plain two-step diffusion sweep of fp64
    // grids whose rows end at chunk boundaries.  PDEHIP_EULER2=8 selects it (measurement: profiles/r03_e2_tile_shapes.log)
    // Round 5 (rows on 128-byte lines): the tall tile wins for fields well beyond the Infinity Cache - 512^3 0.2157 -> 0.2108 ms per step (mean of
    // four alternations), 512 x 512 x 256 +3.7 %, 384^3 +2.6 % - and loses below (256^3 -3 %, 128 x 512 x 512 -0.7 %): profiles/r05_ab_tall_tile.log.
    // PDEHIP_EULER2=8 forces it, PDEHIP_EULER2=4 the 4-row tile.  (The tall tile WITH the ragged-row code, for extents that are not multiples of the tile,
    // was built and measured slower than the 4-row tile everywhere - 513^3 440 against 479, 511^3 538 against 599 Gcell-steps/s: profiles/r05_ab_tall_ragged.log.)
    // Round 6: by default only for all-periodic grids (the 3-buffer instance without the face code, euler2_tall_per_kernel); with faces the 4-row tile
    // with late loads and branches is ahead of the tall one now (512^3: 0.443 against 0.488 ms per launch, profiles/r06_e2_bench6.md)
    const bool all_periodic = xplain == 0 && a.per[0] == 1 && a.per[1] == 1 && a.per[2] == 1;
    const bool tall_auto = t2.ry == 0 && (double)a.n0 * a.n1 * a.n2 * sizeof(T) > 400.0 * 1048576.0 && all_periodic && !(getenv("PDEHIP_E2_PER3") && getenv("PDEHIP_E2_PER3")[0] == '0');
    const bool tall_want = sizeof(T) == 8 && VEC == 2 && (t2.ry == 8 || tall_auto) && has_y && !plan && xplain == 0 && ends == 0 &&
                           (m2 == E2_DIFFUSION) && a.per[1] != 2 && a.per[2] != 2;
    // Row counts that are not a multiple of the tile: the last tile is moved back until it ends with the last row (it
    // recomputes rows of its neighbour, pdehip_march2.inc).  With at least 8 tiles per column the big tile with <= 1/8 of
    // redundant rows beats the exactly fitting smaller one (1.5 x instead of 2 x of the intermediate level); an odd number
    // of non-periodic rows has no exactly fitting tile at all.
    // "Open" rows: a row one to eight cells longer than a whole number of chunks (513 = 4 x 128 + 1) gave the moved last chunk a wave of
    // its own that marched every plane for one vector - 25 % more waves (fp64 513^3 0.281 against 0.228 ms per step at 512^3, fp32 0.268
    // against 0.167).  Instead the tiles cover the whole chunks - the halo columns right of the last one are real cells, or the virtual
    // column through the `zhi2` code of the ragged instances - and the remaining columns are recomputed from the input by the LDS-tiled
    // kernel of pdehip_shell.hip (two layers next to the upper face of the fastest axis).  PDEHIP_OPEN_ROWS=0: off (A/B).
    static const bool open_off = getenv("PDEHIP_OPEN_ROWS") && getenv("PDEHIP_OPEN_ROWS")[0] == '0';
    long open_tail = 0;
    if (!open_off && !plan && xplain == 0 && ends == 0 && m2 == E2_DIFFUSION && a.n2 > CW && a.n2 % CW >= 1 && a.n2 % CW <= 8 && a.per[1] != 2 && a.per[2] != 2) open_tail = a.n2 % CW;
    const long n2t = a.n2 - open_tail;   // the columns the tiles cover
    // "Open" COLUMNS of tiles (round 6): one to four rows beyond a whole number of tiles (513 = 64 x 8 + 1) are left to the same recomputing kernel
    // instead of a moved last tile - the tiles then divide the wave slots like those of the multiple of the tile below (513^3: 516 tiles of 4 rows
    // gave 3 x-chunks = 1548 of 2048 wave slots; 512 rows x 512 columns: the tall tile, 256 x 4 = 1024 of 1024).  The halo rows of the last tiles are
    // real rows (or the wrapped ones: `a.n1` stays the row count of the grid); next to a local upper face the last tile's output row under the
    // virtual row is wrong and recomputed with the rows behind it (the two layers of a job overlap it for an odd remainder).
    // fp64 fields of 8 M cells and more (below, the extra launch costs more than the moved tile).  PDEHIP_OPEN_ROWS=0 / PDEHIP_OPEN_Y=0: off (A/B).
    static const bool open_y_off = getenv("PDEHIP_OPEN_Y") && getenv("PDEHIP_OPEN_Y")[0] == '0';
    long open_y = 0;
    // (fp32: for the wide 4-row tile of all-periodic grids - launch_euler2_t asks for it with ry_f32 = 4)
    const bool open_y_type = (sizeof(T) == 8 && VEC == 2) || (sizeof(T) == 4 && VEC == 4 && ry_f32 == 4);
    if (!open_off && !open_y_off && open_y_type && has_y && !plan && xplain == 0 && ends == 0 && m2 == E2_DIFFUSION && a.per[1] != 2 && a.per[2] != 2 &&
        (double)a.n0 * a.n1 * a.n2 >= 8388608.0 && a.n1 >= 64) {
        // (the recomputing kernel takes six jobs of two layers: the open columns of the fastest axis first)
        const long jobs_left = 6 - (open_tail + 1) / 2;
        const bool tall_rows = tall_want && n2t % CW == 0 && (t2.ry == 8 || (n2t / CW) % 4 == 0) && (a.n1 % 8) >= 1 && ((a.n1 % 8) + 1) / 2 <= jobs_left;
        const long unit_rows = tall_rows ? 8 : 4;
        const long r = a.n1 % unit_rows;
        if (r >= 1 && r <= (unit_rows == 8 ? 7 : 3) && (r + 1) / 2 <= jobs_left) {
            // ... where it fills the wave slots better than the moved last tile does (519 rows = 129 tiles of 4 + 3: as badly quantised as 130 tiles -
            // the extra launch then only costs: 517^3 fp32 664 -> 642 Gcell-steps/s, profiles/r06_call35_sizes.log)
            auto fill = [](long tiles, long slots) { return tiles >= slots ? 1.0 : (double)((slots / tiles) * tiles) / (double)slots; };
            const long ntz_ = (n2t + CW - 1) / CW;
            const long slots_open = (unit_rows == 8 || sizeof(T) == 4) ? 1024 : 2048, slots_moved = sizeof(T) == 4 ? 1024 : 2048;
            const double with_open = fill((a.n1 - r) / unit_rows * ntz_, slots_open), with_moved = fill((a.n1 + 3) / 4 * ntz_, slots_moved);
            if (with_open > with_moved + 0.08) open_y = r;
        }
    }
    const long n1t = a.n1 - open_y;      // the rows the tiles cover
    // (the tall tile has no code for the virtual FAR column of an open row with one more cell: the ragged 4-row instance takes those)
    // (chosen automatically only where the chunks of a row come in fours - workgroups of four waves that stream whole rows: 300 x 512 x 640, five
    // chunks = one-wave workgroups, 561.6 on the tall tile against 585.4 Gcell-steps/s on the 4-row tile, 384 columns 518 against 572:
    // profiles/r06_call32_sizes.log)
    const bool tall = tall_want && n2t % CW == 0 && n1t % 8 == 0 && !(open_tail == 1 && !a.per[2]) && (t2.ry == 8 || (n2t / CW) % 4 == 0);
    const int ry_want = ry;
    while (ry > 1 && n1t % ry) ry /= 2;
    if (has_y && ry < ry_want) {
        int big = ry_want;
        while (big > ry && n1t < 8L * big) big /= 2;
        if (big > ry) ry = big;
        else if (ry == 1) ry = 2;   // (1-row tiles exist for periodic rows of fp32 grids only and recompute 3 x)
    }
    // the stage epilogue (six more streams) does not fit the ragged 4-row fp64 tile without spilling: 2-row tiles there
    const long n2v = (n2t + VEC - 1) / VEC * VEC;   // a row that ends inside a vector: the last chunk is moved back by n2v - n2 cells
    if (m2 == E2_CH_STAGE && sizeof(T) == 8 && ry == 4 && n2v % CW != 0) ry = 2;
    if (!has_y) ry = 1;
    if (tall) ry = 8;
    if ((ry != 1 && ry != 2 && ry != 4 && !tall) || n1t < ry || (n2v != n2t && n2t < CW)) return 0;
    const bool overlap = n2v != n2t || n1t % ry != 0;
    // the wide fp32 tile has no registers for the virtual row / column in a tile's OUTER halo position (next to a moved tile
    // with local faces): the narrow tile takes those grids (launch_euler2_t)
    if (sizeof(T) == 4 && VEC == 4 && ((has_y && n1t % ry != 0 && !a.per[1]) || (a.n2 % CW == 1 && !a.per[2]))) return 0;
    if (overlap && m2 == E2_CH_STAGE) {
        // cells of overlapping tiles are computed and stored twice: nothing a sweep writes may be one of its pointwise inputs
        // (the new state of RK4 written over the old one: those sweeps combine with the pointwise kernels)
        bool alias = a.st_out == a.st_y || a.out == a.st_y;
        for (int m = 0; m < 5; m++) alias = alias || (a.st_k[m] && (a.st_k[m] == a.st_out || a.st_k[m] == a.out));
        if (alias) return 0;
    }
    a.ntz = (n2t + CW - 1) / CW;   // the row may end inside the last chunk
    a.z_open = open_tail > 0;
    a.nty = (n1t + ry - 1) / ry;
    const long tiles = a.ntz * a.nty;
    // every x-chunk recomputes two planes of the intermediate level and re-reads four input planes
    if (ends > 0) {
        // boundary sweep of a slab: the first and the last `ends` planes in ONE launch
        a.lx = ends; a.nxc = 2; a.xstride = a.n0 - ends;
    } else if (!has_y) {
        // 2-D: a wave's march is a chain of dependent row loads (~1 us each out of the Infinity Cache for grids of a few
        // MB), so short chunks win until the chip is full: up to ~4096 waves, chunks of at least `minlx` rows (the
        // 4 overlap rows per chunk cost no HBM traffic for cache-resident grids)
        const long minlx = t2.order > 0 ? t2.order : 2;
        long nxc = (t2.blocks ? t2.blocks : 4096) / tiles;
        if (nxc > a.n0 / minlx) nxc = a.n0 / minlx;
        if (nxc < 1) nxc = 1;
        const long lx = (a.n0 + nxc - 1) / nxc;
        a.lx = (int)lx;
        a.nxc = (a.n0 + lx - 1) / lx;
        a.xstride = lx;
    } else {
        // ONE full round of 2048 wave tiles (256 CUs x 8 wave slots at 2 waves per SIMD): measured best or equal from 64 to
        // 512 planes (0.126 vs 0.131 ms/step at 256 planes, 0.066 vs 0.071 at 128 with 4096 tiles; in the slab loop the
        // boundary sweep and the RCCL kernel otherwise queue up behind the second round:
        // profiles/r01_time_tiles_vs_planes.log).  Interior sweep of a THIN slab (exchange-bound): at most 1536, so that
        // the RCCL kernel of the halo stream finds free wave slots at once — workgroups march for the whole sweep, a kernel
        // launched behind a full round waits for it to end (measured: 90 us for 13 us of work).
        const bool thin = xplain && a.n0 < 96;
        // a box of the fast block loop (plain rows / columns): 7/8 of a round - the rim, pack, RCCL and unpack kernels of the halo stream
        // otherwise wait for the END of the sweep (0.0536 -> 0.0501 ms per step at 256 x 128 x 512, profiles/r05_probe_block.md)
        const bool boxed = a.per[1] == 2 || a.per[2] == 2;
        const bool wide1 = sizeof(T) == 4 && VEC == 4 && (m2 == E2_CH_STAGE ? ry == 2 : ry == 4) && has_y;   // (euler2_stage1w_kernel, euler2_wide4_kernel: one wave per SIMD)
        const long cap = t2.blocks ? t2.blocks : ((tall || wide1) ? 1024 : (thin ? 1536 : (boxed ? 1792 : 2048)));   // (the tall tile runs one wave per SIMD)
        static const long floor_env = getenv("PDEHIP_E2_MINLX") ? atol(getenv("PDEHIP_E2_MINLX")) : 0;   // tuning aid
        long nxc;
        if (thin) {
            nxc = cap / tiles;
            long minlx = 16;
            while (minlx > 2 && tiles * (a.n0 / minlx) < cap) minlx /= 2;
            if (floor_env > 0) minlx = floor_env;
            if (nxc > a.n0 / minlx) nxc = a.n0 / minlx;
            if (nxc < 1) nxc = 1;
        } else {
            // The number of x-chunks by a cost model.  A wave marches lx + 2 planes; the chip holds `cap` of them.  While they
            // fit (W <= cap) the sweep is bound by the bytes (W * L) down to the latency floor of a lone march; beyond, the
            // waves left over for the last round march ALONE at that floor: tile counts just above a divisor of `cap` (512 x 513
            // x 512: 516 tiles, 4 chunks = 2064 waves took 0.307 ms per step against 0.225 for 512^3; 300^3: 225 tiles, 10
            // chunks = 2250 waves) take one chunk less instead.  Chunks shorter than 16 planes (two recomputed planes per
            // chunk: > 12.5 % extra work) only while the first round is not full (100^3: 19.7 -> 8.1 us per step).
            double best = 0;
            nxc = 1;
            for (long c = 1; c <= a.n0 / 2 || c == 1; c++) {
                const long lx = (a.n0 + c - 1) / c, real = (a.n0 + lx - 1) / lx;
                if (real != c) continue;   // the same chunking as a smaller count
                if (floor_env > 0 ? lx < floor_env : (lx < 16 && (c - 1) * tiles >= cap)) break;
                const long W = real * tiles;
                const double full = (double)(W / cap), part = (double)(W % cap) / (double)cap;
                // a wave needs 1.6 - 1.9 us per plane whether the chip is full or not (200^3: 1200 waves of 19 planes took as long
                // per plane as 2000 waves of 12): one round costs its march length, nearly whatever its size; the waves of an
                // incomplete LAST round start while the round before drains (measured: 0.36 of a round for a handful)
                double rounds = W <= cap ? 0.85 + 0.15 * (double)W / (double)cap : full + (part > 0 ? (part > 0.36 ? part : 0.36) : 0.0);
                const double cost = (double)(lx + 2) * rounds;
                if (best == 0 || cost < best) { best = cost; nxc = c; }
            }
        }
        const long lx = (a.n0 + nxc - 1) / nxc;
        a.lx = (int)lx;
        a.nxc = (a.n0 + lx - 1) / lx;
        a.xstride = lx;
    }
    // waves per workgroup = neighbouring chunks of the same rows (1, 2 or 4; PDEHIP_EULER2 third field overrides)
    int nwz = (a.ntz % 4 == 0) ? 4 : (a.ntz % 2 == 0 ? 2 : 1), nwy = 1;
    if (has_y && t2.order > 0) {   // tuning aid: third field = 10 * (waves along the rows) + (waves along the fastest axis)
        const int wz_ = t2.order % 10, wy_ = t2.order / 10 > 0 ? t2.order / 10 : 1;
        if (wz_ > 0 && a.ntz % wz_ == 0 && a.nty % wy_ == 0 && wz_ * wy_ <= 4) { nwz = wz_; nwy = wy_; }
    }
    a.nwy = nwy;
    a.nblocks = a.nxc * tiles / (nwz * nwy);
    a.no_swizzle = 0;
    const dim3 grid((unsigned)a.nblocks), block(64 * nwz * nwy);
    // real halo planes instead of BCs on the slowest axis: both sides (1), upper side only (2), lower side only (3)
    if (xplain) a.per[0] = xplain == 1 ? 2 : (xplain == 2 ? 3 : 4);
    if (plan) {   // the caller launches a run-time compiled instance itself (pdehip_jit.hip)
        if (has_y ? (ry != 2 && ry != 4) : ry != 1) return 0;
        plan->a = a; plan->grid = (unsigned)a.nblocks; plan->block = 64u * nwz * nwy; plan->ry = ry; plan->has_y = has_y;
        *done = true;
        return 0;
    }
    // the stage epilogue exists for real halo layers on BOTH sides (a run-time argument of the plain instances) but not as
    // one-sided (XS) instances: the first / last slab of a non-periodic axis combines with the pointwise kernels
    if (m2 == E2_CH_STAGE && xplain > 1) return 0;
    // every axis periodic: the instances without the code of the local faces (pdehip_march2.inc, PER3).  PDEHIP_E2_PER3=0: off (A/B)
    static const bool per3_off = getenv("PDEHIP_E2_PER3") && getenv("PDEHIP_E2_PER3")[0] == '0';
    const bool per3 = !per3_off && xplain == 0 && a.per[0] == 1 && a.per[1] == 1 && a.per[2] == 1 && m2 == E2_DIFFUSION && sizeof(T) == 8 && VEC == 2 && has_y && !plan;
    if constexpr (sizeof(T) == 8 && VEC == 2) {
        if (tall) {
            if (dry_run) { *done = true; return 0; }
            const bool nt_ = ((double)a.n0 * a.n1 * a.n2 * sizeof(T) > 192.0 * 1048576.0);
            const bool unit_ = a.sx == 1.0 && a.sy == 1.0 && a.sz == 1.0 && a.s1 == 1.0;
            if (per3) {
                note_kernel("euler2_tall_per_kernel<double,2,%s,%s> (8 rows, 3 plane buffers, 1 wave per SIMD, all-periodic)", unit_ ? "E2_DIFFUSION_UNIT" : "E2_DIFFUSION", nt_ ? "NT" : "plain stores");
                if (unit_ && nt_) hipLaunchKernelGGL((euler2_tall_per_kernel<T, VEC, E2_DIFFUSION_UNIT, true>), grid, block, 0, st, a);
                else if (unit_) hipLaunchKernelGGL((euler2_tall_per_kernel<T, VEC, E2_DIFFUSION_UNIT, false>), grid, block, 0, st, a);
                else if (nt_) hipLaunchKernelGGL((euler2_tall_per_kernel<T, VEC, E2_DIFFUSION, true>), grid, block, 0, st, a);
                else hipLaunchKernelGGL((euler2_tall_per_kernel<T, VEC, E2_DIFFUSION, false>), grid, block, 0, st, a);
                PDEHIP_HIP(hipGetLastError());
                if (open_tail || open_y) PDEHIP_TRY(shell_open_rows(n, a, (int)open_tail, (int)open_y, st));
                *done = true;
                return 0;
            }
            note_kernel("euler2_tall_kernel<double,2,8,%s,%s> (8 rows, 4 plane buffers, 1 wave per SIMD)", unit_ ? "E2_DIFFUSION_UNIT" : "E2_DIFFUSION", nt_ ? "NT" : "plain stores");
            if (unit_ && nt_) hipLaunchKernelGGL((euler2_tall_kernel<T, VEC, 8, E2_DIFFUSION_UNIT, true>), grid, block, 0, st, a);
            else if (unit_) hipLaunchKernelGGL((euler2_tall_kernel<T, VEC, 8, E2_DIFFUSION_UNIT, false>), grid, block, 0, st, a);
            else if (nt_) hipLaunchKernelGGL((euler2_tall_kernel<T, VEC, 8, E2_DIFFUSION, true>), grid, block, 0, st, a);
            else hipLaunchKernelGGL((euler2_tall_kernel<T, VEC, 8, E2_DIFFUSION, false>), grid, block, 0, st, a);
            PDEHIP_HIP(hipGetLastError());
            if (open_tail || open_y) PDEHIP_TRY(shell_open_rows(n, a, (int)open_tail, (int)open_y, st));
            *done = true;
            return 0;
        }
    }
    {   // is there an offline instance of this tile?  (the list below, PDEHIP_E2; asked before a dry run answers "covered")
        const bool xs_ = xplain > 1;
        bool have;
        if (!has_y) have = ry == 1 && !xs_ && (sizeof(T) == 8 || VEC == 4);
        else if (sizeof(T) == 8) have = ry == 4 || ry == 2;
        else if (VEC == 4) have = ry == 2 || (ry == 1 && !xs_) || (ry == 4 && all_periodic && m2 == E2_DIFFUSION && !plan && ends == 0);   // (4: euler2_wide4_kernel)
        else have = ry == 4 || ry == 2 || (ry == 1 && !xs_);
        if (m2 == E2_CH_STAGE && sizeof(T) == 4 && VEC == 4 && has_y && ry > 1) have = ry == 2 && !xs_;   // (two waves per SIMD: 256 VGPRs + scratch; ry 2: euler2_stage1w_kernel)
        // a 1-row tile of a 3-D grid is its own neighbour's halo: the tile of row 1 reads the virtual row -1, which only the
        // tile of row 0 transforms (`ylo`) - correct for periodic rows only
        if (has_y && ry == 1 && !a.per[1]) have = false;
        if (!have) return 0;
    }
    if (dry_run) { *done = true; return 0; }
    if (m2 == E2_CUSTOM || m2 == E2_CUSTOM2) PDEHIP_FAIL(E_RUNTIME, "internal: the custom two-level kernel exists only as a run-time build");
    // the variant without the ragged-row code (rows end at chunk boundaries) exists for the 4-row fp64 tile only: there
    // the 5 VGPRs decide whether the loads can be issued early (8-19 % at 256^3 and slab-sized grids)
    // XS: the one-sided halo modes of the first / last slab of a non-periodic axis are separate instances (with the
    // ragged-row code): compiled into the hot instances they cost 5-9 % through register allocation alone
    const bool xs = xplain > 1;
    // (the virtual rows next to a moved last tile - pdehip_march2.inc: ylo2 / yhi2 - are part of the ragged-row code)
    // (... and so is the virtual FAR column right of the last chunk of an open row with one more cell: zhi2)
    const bool ragged = xs || !(sizeof(T) == 8 && ry == 4 && n2v % CW == 0) || (has_y && n1t % ry != 0 && !a.per[1]) || (open_tail == 1 && !a.per[2]);
    // NT: streaming stores, for the hot instance and fields that do not fit the 256 MB Infinity Cache
#if defined(PDEHIP_NT_LOADS) && PDEHIP_NT_LOADS == 2
    const bool nt = false;   // A/B variant: non-temporal loads, plain stores
#else
    // (round 6: also the ragged 4-row fp64 diffusion tile of two-sided grids - 500 x 500 x 300, rows that end inside a chunk - has a streaming-store form)
    const bool nt = (!ragged || (sizeof(T) == 8 && ry == 4 && has_y && !xs && m2 == E2_DIFFUSION)) && m2 != E2_CH_STAGE && ((double)a.n0 * a.n1 * a.n2 * sizeof(T) > 192.0 * 1048576.0);
#endif
    // unit spacing and D = 1 (UnitGrid benchmarks): the 3-D instances exist without the multiplications by 1.0 (fp32 and the
    // cache-resident sizes are VALU-bound: up to 10 %)
    static const bool unit_off = getenv("PDEHIP_NO_UNIT") != nullptr;   // A/B aid
    const bool unit = !unit_off && a.sx == 1.0 && a.sy == 1.0 && a.sz == 1.0 && a.s1 == 1.0;
#define PDEHIP_E2(RY_, HY_, RG_, XS_, NT_)                                                                                               \
    if (!launched && ry == RY_ && has_y == HY_ && ragged == RG_ && xs == XS_ && nt == NT_) {                                             \
        launched = true;                                                                                                                 \
        if (m2 == E2_DIFFUSION) {                                                                                                       \
            if constexpr (!XS_) {   /* every instance except the one-sided slab ends */                                                       \
                if (unit) hipLaunchKernelGGL((euler2_kernel<T, VEC, RY_, E2_DIFFUSION_UNIT, HY_, RG_, XS_, NT_>), grid, block, 0, st, a);  \
                else hipLaunchKernelGGL((euler2_kernel<T, VEC, RY_, E2_DIFFUSION, HY_, RG_, XS_, NT_>), grid, block, 0, st, a);            \
            } else hipLaunchKernelGGL((euler2_kernel<T, VEC, RY_, E2_DIFFUSION, HY_, RG_, XS_, NT_>), grid, block, 0, st, a);             \
        }                                                                                                                                \
        else if (m2 == E2_CH_EULER) hipLaunchKernelGGL((euler2_kernel<T, VEC, RY_, E2_CH_EULER, HY_, RG_, XS_, NT_>), grid, block, 0, st, a); \
        else if (m2 == E2_CH_STAGE) {                                                                                                    \
            if constexpr (!XS_ && !NT_ && !(sizeof(T) == 8 && RY_ == 4 && RG_) && !(sizeof(T) == 4 && VEC == 4 && HY_ && RY_ > 1)) hipLaunchKernelGGL((euler2_kernel<T, VEC, RY_, E2_CH_STAGE, HY_, RG_, false, false>), grid, block, 0, st, a); \
            else return 0;                                                                                                               \
        } else hipLaunchKernelGGL((euler2_kernel<T, VEC, RY_, E2_CH_SCALED, HY_, RG_, XS_, NT_>), grid, block, 0, st, a);                  \
    }
    bool launched = false, noted = false;
    if constexpr (sizeof(T) == 8 && VEC == 2) {
        // a slab of a grid that is periodic along its rows and its fastest axis, real halo planes on both sides (interior and boundary sweeps of the
        // slab loops): the all-periodic 4-row body with the march axis as the arguments say (PER3 = 2).  PDEHIP_E2_PERYZ=0: off (A/B)
        static const bool peryz_off = getenv("PDEHIP_E2_PERYZ") && getenv("PDEHIP_E2_PERYZ")[0] == '0';
        if (!launched && !peryz_off && xplain == 1 && a.per[1] == 1 && a.per[2] == 1 && m2 == E2_DIFFUSION && has_y && !plan && ry == 4 && !ragged && !open_tail && !open_y) {
            note_kernel("euler2_peryz_kernel<double,2,%s,%s> (4 rows, 2 waves per SIMD, rows and fastest axis periodic, halo planes along the march axis)", unit ? "E2_DIFFUSION_UNIT" : "E2_DIFFUSION", nt ? "NT" : "plain stores");
            if (unit && nt) hipLaunchKernelGGL((euler2_peryz_kernel<T, VEC, E2_DIFFUSION_UNIT, true>), grid, block, 0, st, a);
            else if (unit) hipLaunchKernelGGL((euler2_peryz_kernel<T, VEC, E2_DIFFUSION_UNIT, false>), grid, block, 0, st, a);
            else if (nt) hipLaunchKernelGGL((euler2_peryz_kernel<T, VEC, E2_DIFFUSION, true>), grid, block, 0, st, a);
            else hipLaunchKernelGGL((euler2_peryz_kernel<T, VEC, E2_DIFFUSION, false>), grid, block, 0, st, a);
            launched = true;
            noted = true;
        }
        if (per3 && ry == 4 && !ragged) {   // (fp64, 4 rows, rows that end at chunk boundaries - or open rows: their last columns follow below)
            note_kernel("euler2_per_kernel<double,2,%s,%s> (4 rows, 2 waves per SIMD, all-periodic)", unit ? "E2_DIFFUSION_UNIT" : "E2_DIFFUSION", nt ? "NT" : "plain stores");
            if (unit && nt) hipLaunchKernelGGL((euler2_per_kernel<T, VEC, E2_DIFFUSION_UNIT, true>), grid, block, 0, st, a);
            else if (unit) hipLaunchKernelGGL((euler2_per_kernel<T, VEC, E2_DIFFUSION_UNIT, false>), grid, block, 0, st, a);
            else if (nt) hipLaunchKernelGGL((euler2_per_kernel<T, VEC, E2_DIFFUSION, true>), grid, block, 0, st, a);
            else hipLaunchKernelGGL((euler2_per_kernel<T, VEC, E2_DIFFUSION, false>), grid, block, 0, st, a);
            launched = true;
        }
    }
    if constexpr (sizeof(T) == 4 && VEC == 4) {
        if (ry == 4 && m2 == E2_DIFFUSION) {   // fp32 diffusion: the wide 4-row tile at one wave per SIMD (pdehip_march2.inc; `have` above)
            // streaming stores for fields beyond the Infinity Cache (512^3: 874 against 848 Gcell-steps/s)
            const bool nt4 = (double)a.n0 * a.n1 * a.n2 * sizeof(T) > 192.0 * 1048576.0;
            note_kernel("euler2_wide4_kernel<float,4,%s,%s> (4 rows, 1 wave per SIMD, all-periodic)", unit ? "E2_DIFFUSION_UNIT" : "E2_DIFFUSION", nt4 ? "NT" : "plain stores");
#define PDEHIP_W4(M2_, NT_) hipLaunchKernelGGL((euler2_wide4_kernel<T, VEC, M2_, NT_>), grid, block, 0, st, a)
            if (unit && nt4) PDEHIP_W4(E2_DIFFUSION_UNIT, true); else if (unit) PDEHIP_W4(E2_DIFFUSION_UNIT, false);
            else if (nt4) PDEHIP_W4(E2_DIFFUSION, true); else PDEHIP_W4(E2_DIFFUSION, false);
#undef PDEHIP_W4
            launched = true;
            noted = true;
        }
        if (m2 == E2_CH_STAGE && ry == 2 && has_y && !xs) {   // the wide fp32 stage tile at one wave per SIMD (pdehip_march2.inc)
            hipLaunchKernelGGL((euler2_stage1w_kernel<T, VEC, 2, true>), grid, block, 0, st, a);
            launched = true;
        }
    }
    if constexpr (sizeof(T) == 8 || VEC == 4) {
        PDEHIP_E2(1, false, true, false, false)
        PDEHIP_E2(2, true, true, false, false)
        PDEHIP_E2(2, true, true, true, false)
    }
    if constexpr (sizeof(T) == 8) {
        PDEHIP_E2(4, true, true, false, false) PDEHIP_E2(4, true, false, false, false) PDEHIP_E2(4, true, false, false, true)
        PDEHIP_E2(4, true, true, false, true)
        PDEHIP_E2(4, true, true, true, false)
    }
    if constexpr (sizeof(T) == 4 && VEC == 4) { PDEHIP_E2(1, true, true, false, false) }   // 1-row wide tile (with the stage epilogue: 220 VGPRs)
    if constexpr (sizeof(T) == 4 && VEC == 2) {   // narrow fp32 tiles, 3-D only
        PDEHIP_E2(4, true, true, false, false) PDEHIP_E2(2, true, true, false, false) PDEHIP_E2(1, true, true, false, false)
        PDEHIP_E2(4, true, true, true, false) PDEHIP_E2(2, true, true, true, false)
    }
#undef PDEHIP_E2
    if (!launched) return 0;   // no instance of this shape (the caller takes the pass-by-pass path)
    if (!noted && !(per3 && ry == 4 && !ragged) && !(sizeof(T) == 4 && VEC == 4 && m2 == E2_CH_STAGE && ry == 2 && has_y && !xs))
        note_kernel("euler2_kernel<%s,%d,%d,m2=%d%s,%s,%s,%s,%s>", sizeof(T) == 8 ? "double" : "float", VEC, ry, m2, (unit && m2 == E2_DIFFUSION && !xs) ? " unit" : "", has_y ? "3-D" : "2-D",
                    ragged ? "ragged" : "aligned rows", xs ? "one-sided" : "two-sided", nt ? "NT" : "plain stores");
    PDEHIP_HIP(hipGetLastError());
    if (open_tail || open_y) PDEHIP_TRY(shell_open_rows(n, a, (int)open_tail, (int)open_y, st));   // the last columns of every row, the last rows of every plane
    *done = true;
    return 0;
}

template <typename T>
static int launch_euler2_t(const NGrid &n, LapArgs a, int xplain, hipStream_t st, bool *done, bool dry_run, int ends, int m2,
                           Euler2Plan *plan, bool narrow_only)
{
    if constexpr (sizeof(T) == 8) {
        return launch_euler2_tv<double, 2>(n, a, xplain, st, done, dry_run, ends, m2, plan, 0);
    } else {
        const TuneF32 &tf = tune_f32();
        const bool stage = m2 == E2_CH_STAGE;
        // an fp32 box that starts two cells into a four-cell vector (the interior of a block whose fastest axis is cut: pdehip_block2_loops.h):
        // the narrow tile's 8-byte vectors take it
        if (narrow_only) {
            if (n.ndim != 3 || plan || stage) return 0;
            return launch_euler2_tv<float, 2>(n, a, xplain, st, done, dry_run, ends, m2, plan, 4);
        }
        // Measured at 256^3 / 512^3 (profiles/r03_f32_tiles.md): the sweeps without the stage epilogue are fastest on the wide
        // 2-row tile (diffusion 0.0233 vs 0.0249 ms per step, Cahn-Hilliard 0.0575 vs 0.0585); the Runge-Kutta stage sweeps
        // need the narrow 4-row tile to carry their epilogue at all (RKF45 attempt 0.786 -> 0.755 ms).  The run-time built
        // kernels of pdehip_jit.hip keep the wide tile (`plan`).
        int vec = 4, ry = 2;
        // all-periodic diffusion: the wide tile with four rows at one wave per SIMD (round 6; PDEHIP_F32_WIDE4=0: off, A/B).  With faces it was measured
        // too: 441.6-443.1 against 420.5-424.7 us per launch at 512^3, 61.7 against 61.6 at 256^3 in the kernel trace (profiles/r06_f32_wide4.md) - not used there
        static const bool wide4_off = getenv("PDEHIP_F32_WIDE4") && getenv("PDEHIP_F32_WIDE4")[0] == '0';
        // (grids of a few MB are bound by the latency of a march, not by instructions: 64 x 64 x 256 lost 4 %)
        const bool wide4 = !wide4_off && n.ndim == 3 && !plan && !stage && m2 == E2_DIFFUSION && xplain == 0 && ends == 0 && a.per[0] == 1 && a.per[1] == 1 && a.per[2] == 1 &&
                           (a.n1 % 4 == 0 || ((double)a.n0 * a.n1 * a.n2 >= 8388608.0 && a.n1 >= 64)) &&   // (or one to three rows more, left open: launch_euler2_tv)
                           !tf.vec && (double)a.n0 * a.n1 * a.n2 >= 2097152.0;
        // (rows that fill the 256-cell chunks of the wide tile badly go to the narrow tile below: 384 cells = 1.5 chunks lost 24 % here)
        auto fill4 = [&](long cw) {
            const long t = a.n2 % cw;
            return (a.n2 > cw && t >= 1 && t <= 8) ? 1.0 : (double)a.n2 / (double)((a.n2 + cw - 1) / cw * cw);
        };
        if (wide4 && !(fill4(128) > 1.15 * fill4(256))) {
            bool ok4 = false;
            PDEHIP_TRY((launch_euler2_tv<float, 4>(n, a, xplain, st, &ok4, dry_run, ends, m2, plan, 4)));
            if (ok4) { *done = true; return 0; }
        }
        if (n.ndim == 3 && !plan) {
            if (stage) { vec = tf.svec ? tf.svec : 2; ry = tf.svec ? tf.sry : 4; }
            else if (tf.vec) { vec = tf.vec; ry = tf.ry; }
            else {
                // rows that fill the 128-cell chunks of the narrow tile much better than the 256-cell chunks of the wide one
                // (300 cells: 78 % against 59 % of the lanes own cells; 513: 80 % against 67 %)
                // (rows one or two cells beyond whole chunks leave those cells to another kernel: launch_euler2_tv, "open" rows)
                auto fill = [&](long cw) {
                    const long t = a.n2 % cw;
                    return (a.n2 > cw && t >= 1 && t <= 8) ? 1.0 : (double)a.n2 / (double)((a.n2 + cw - 1) / cw * cw);
                };
                const double wide = fill(256), narrow = fill(128);
                if (narrow > 1.15 * wide) { vec = 2; ry = 4; }
            }
        }
        // PDEHIP_F32_STAGE_WIDE=1: the stage sweeps on the wide 2-row tile at ONE wave per SIMD (16-byte accesses; euler2_stage1w_kernel)
        static const int stage_wide = getenv("PDEHIP_F32_STAGE_WIDE") ? atoi(getenv("PDEHIP_F32_STAGE_WIDE")) : 0;
        if (stage && stage_wide && n.ndim == 3 && !plan && !tf.svec) { vec = 4; ry = 2; }
        else if (stage && vec == 4 && ry > 1 && n.ndim == 3) ry = 1;   // the wide tile carries the stage epilogue with one row only
        if (vec == 2) return launch_euler2_tv<float, 2>(n, a, xplain, st, done, dry_run, ends, m2, plan, ry);
        PDEHIP_TRY((launch_euler2_tv<float, 4>(n, a, xplain, st, done, dry_run, ends, m2, plan, ry)));
        // what the wide tile declines (rows shorter than its chunk that end inside a 4-cell vector, moved last tiles next to
        // local faces) the narrow tile (2-cell vectors, 4 rows) may still take
        if (!*done && n.ndim == 3 && !plan) return launch_euler2_tv<float, 2>(n, a, xplain, st, done, dry_run, ends, m2, plan, 4);
        return 0;
    }
}

int launch_euler2(const NGrid &n, const void *in, void *out, double s1, double s2, const InputBCs &fg,
                  int xplain, hipStream_t st, bool *done, bool dry_run, int ends, int m2, const InputBCs *fg1, double gamma,
                  Euler2Plan *plan, const StageFuse *stage, int yzplain)
{
    *done = false;
    if ((m2 == E2_CH_STAGE) != (stage != nullptr)) PDEHIP_FAIL(E_RUNTIME, "internal: stage sweep without / with a stage descriptor");
    const long vec = 16 / elem_size(n.dtype);
    if ((m2 == E2_CH_EULER || m2 == E2_CH_SCALED || m2 == E2_CH_STAGE) && !fg1) PDEHIP_FAIL(E_RUNTIME, "internal: fused Cahn-Hilliard sweep without the faces of mu");
    if (tune2().off || force_generic_kernels() || (n.ndim != 3 && n.ndim != 2) || in == out) return 0;
    // kernel axes (march, rows, lanes) <- normalised grid axes: 3-D (0, 1, 2); 2-D (1, -, 2): the march axis is the first
    // grid axis and there are no rows
    const int am = n.ndim == 3 ? 0 : 1;
    if (n.ndim == 2 && xplain) return 0;
    if (n.n[am] < (xplain ? 1 : 4) || (n.ndim == 3 && n.n[1] < 4) || n.n[2] < 4 || n.p[am] >= (1L << 31)) return 0;
    const bool narrow_only = n.dtype == PDEHIP_F32 && n.off % vec != 0 && n.off % 2 == 0;   // (see launch_euler2_t)
    if ((uintptr_t)in % 16 || (uintptr_t)out % 16 || (n.off % vec && !narrow_only) || n.p[am] % vec || n.p[1] % vec) return 0;
    LapArgs a;
    memset(&a, 0, sizeof(a));
    for (int k = 0; k < 3; k++) {   // k = kernel axis
        if (k == 0 && xplain == 1) continue;
        if (k == 0 && xplain > 1) {
            // first / last slab of a non-periodic axis: ONE local face (the other side has real halo planes)
            const int side = xplain == 2 ? 0 : 1;
            const InputBCs &f1 = fg1 ? *fg1 : fg;
            const long want = side ? n.n[am] - 1 : 0;
            if (!fg.on[am][side] || fg.idx[am][side] != want || !f1.on[am][side] || f1.idx[am][side] != want) return 0;
            a.ibc[0][side].on = 1; a.ibc[0][side].idx = want; a.ibc[0][side].c = fg.c[am][side]; a.ibc[0][side].f = fg.f[am][side];
            a.ibc1[0][side].on = 1; a.ibc1[0][side].idx = want; a.ibc1[0][side].c = f1.c[am][side]; a.ibc1[0][side].f = f1.f[am][side];
            continue;
        }
        if (k == 1 && n.ndim == 2) { a.per[1] = 1; continue; }
        // block decomposition (pdehip_block2_loops.h): `n` describes a BOX of a larger array - two real halo rows (bit 0) / columns
        // (bit 1) on either side in memory; no faces on those axes
        if (k >= 1 && n.ndim == 3 && (yzplain & (1 << (k - 1)))) { a.per[k] = 2; continue; }
        const int ax = (k == 0) ? am : k;
        // both faces periodic, or both local (virtual point from the adjacent cell); the same for both levels
        const int cls = classify_axis(fg, ax, n.n[ax]);
        if (cls < 0 || (fg1 && classify_axis(*fg1, ax, n.n[ax]) != cls)) return 0;
        a.per[k] = cls;
        for (int side = 0; side < 2; side++) {
            a.ibc[k][side].on = 1;
            a.ibc[k][side].idx = fg.idx[ax][side];
            a.ibc[k][side].c = fg.c[ax][side];
            a.ibc[k][side].f = fg.f[ax][side];
            const InputBCs &f1 = fg1 ? *fg1 : fg;
            a.ibc1[k][side].on = 1;
            a.ibc1[k][side].idx = f1.idx[ax][side];
            a.ibc1[k][side].c = f1.c[ax][side];
            a.ibc1[k][side].f = f1.f[ax][side];
        }
    }
    a.gamma = gamma;
    if (stage) {
        if (!stage->y || !stage->out2 || stage->out2 == in || ((stage->kind == 0 || stage->kind == 3) && !out)) PDEHIP_FAIL(E_VALUE, "stage sweep: NULL or aliased array pointer");
        a.st_kind = stage->kind; a.st_y = stage->y; a.st_out = stage->out2; a.st_err = stage->err;
        int nk = 0;
        for (int m = 0; m < 5 && stage->k[m]; m++, nk++) { a.st_k[m] = stage->k[m]; a.st_c[m] = stage->c[m]; }
        if ((stage->kind == 1 && nk != 3) || (stage->kind == 2 && (nk != 4 || !stage->err)) || (stage->kind == 3 && nk != 1) ||
            (stage->kind == 4 && (nk != 2 || !stage->err || stage->k[1] != in)))
            PDEHIP_FAIL(E_RUNTIME, "internal: malformed stage descriptor");
        a.st_c[5] = stage->c_new;
        if (!stage_aligned(a)) return 0;
    }
    a.in = in; a.out = out; a.y = in;
    a.n0 = n.n[am]; a.n1 = n.ndim == 3 ? n.n[1] : 1; a.n2 = n.n[2];
    a.p0 = n.p[am]; a.p1 = n.ndim == 3 ? n.p[1] : 0; a.off = n.off;
    a.o_off = n.off; a.o_s0 = a.p0; a.o_s1 = a.p1;
    a.sx = n.lap_scale[am]; a.sy = n.lap_scale[1]; a.sz = n.lap_scale[2];
    a.s1 = s1; a.s2 = s2;
    a.ndim = n.ndim; a.any_ibc = 1;
    // squared central gradient of the custom epilogue, kernel-axis order (cartesian.py:661: 0.25 / dx**2)
    a.gs[0] = 0.25 / (n.dx[am] * n.dx[am]); a.gs[1] = 0.25 / (n.dx[1] * n.dx[1]); a.gs[2] = 0.25 / (n.dx[2] * n.dx[2]);
    if (n.dtype == PDEHIP_F64) return launch_euler2_t<double>(n, a, xplain, st, done, dry_run, ends, m2, plan, false);
    return launch_euler2_t<float>(n, a, xplain, st, done, dry_run, ends, m2, plan, narrow_only);
}

// (see preload_stencil_kernels, pdehip_kernels.hip: the code object of this translation unit is loaded when the device is selected, not in the middle of a run)
int preload_e2_kernels()
{
    hipFuncAttributes attr;
    PDEHIP_HIP(hipFuncGetAttributes(&attr, reinterpret_cast<const void *>(&euler2_kernel<double, 2, 4, E2_DIFFUSION, true, false, false, false>)));
    return 0;
}

}  // namespace PDEHIP_VARIANT_NS
}  // namespace pdehip
