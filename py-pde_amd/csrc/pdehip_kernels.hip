// pdehip_kernels.hip — hand-written gfx950 (CDNA4) kernels for py-pde's Cartesian
// finite-difference operators.  Bandwidth-bound stencil work: no MFMA.  Design:
//   * every cell of the input is fetched from HBM once: a wavefront marches along the slowest
//     axis keeping three planes of its tile in registers (prev / cur / next),
//   * a wave tile is RY rows x (CZ chunks of 64 lanes x 16 bytes): with CZ covering the whole
//     fastest axis a wave streams RY*row_pitch CONTIGUOUS bytes per plane, which is what the
//     HBM write path wants (measured: scattered 1 KiB pieces 4.0-4.4 TB/s, contiguous 5.6+),
//   * every global access is an aligned 16-byte dwordx4; neighbours along the fastest axis come
//     from wavefront DPP shifts / readlane, never from memory; the two halo cells of a wave
//     tile are one broadcast load,
//   * no LDS, no barriers: halo rows between neighbouring tiles are served by the XCD's L2,
//     which is why the block -> tile map hands each XCD a contiguous range of tiles,
//   * first-order BCs with scalar coefficients are evaluated on the fly on the input side (halo
//     rows/planes/cells are loaded from the source layer and transformed), so that an Euler step
//     is a single kernel and ghost cells are never materialised inside a time loop.
// Arithmetic follows the reference expression order (pde/backends/numba/operators/
// cartesian.py) and the file is compiled with -ffp-contract=off: results are bit-identical to
// the CPU oracle.
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


#include "pdehip_march.inc"
#include "pdehip_div.inc"

// ---------------------------------------------------------------------------------------------
// generic fallback: one cell per thread, direct loads (any shape / any alignment / 1-D).
// ---------------------------------------------------------------------------------------------
template <typename T, int MODE>
__global__ void __launch_bounds__(256) lap_generic_kernel(LapArgs a)
{
    const long total = a.n0 * a.n1 * a.n2;
    const T *in = (const T *)a.in;
    T *out = (T *)a.out;
    const T *yin = (const T *)a.y;
    for (long t = blockIdx.x * (long)blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
        const long k = t % a.n2;
        const long j = (t / a.n2) % a.n1;
        const long i = t / (a.n2 * a.n1);
        const T *c = in + a.off + i * a.p0 + j * a.p1 + k;
        const double mid = (double)c[0];
        double lap;
        if (a.ndim == 1) {
            double l = (double)c[-1], r = (double)c[1];
            if (a.any_ibc) {
                // 1-D: the two virtual points on the fly (`const + factor * in[idx]`, rounded like a stored ghost cell): a step of a
                // small 1-D grid is ONE launch instead of ghost kernel + stencil kernel
                if (k == 0 && a.ibc[2][0].on) l = (double)(T)(a.ibc[2][0].c + a.ibc[2][0].f * (double)in[a.off + a.ibc[2][0].idx]);
                if (k == a.n2 - 1 && a.ibc[2][1].on) r = (double)(T)(a.ibc[2][1].c + a.ibc[2][1].f * (double)in[a.off + a.ibc[2][1].idx]);
            }
            lap = (l - 2 * mid + r) * a.sz;
        } else if (a.ndim == 2) {
            const double lx = ((double)c[-a.p1] - 2 * mid + (double)c[a.p1]) * a.sy;
            const double ly = ((double)c[-1] - 2 * mid + (double)c[1]) * a.sz;
            lap = lx + ly;
        } else {
            const double vm = 2 * mid;
            const double lx = ((double)c[-a.p0] - vm + (double)c[a.p0]) * a.sx;
            const double ly = ((double)c[-a.p1] - vm + (double)c[a.p1]) * a.sy;
            const double lz = ((double)c[-1] - vm + (double)c[1]) * a.sz;
            lap = lx + ly + lz;
        }
        double yy = 0;
        if (MODE == LAP_EULER) yy = (double)yin[a.off + i * a.p0 + j * a.p1 + k];
        out[a.o_off + i * a.o_s0 + j * a.o_s1 + k] = (T)epilogue<MODE>(lap, mid, yy, a.s1, a.s2, a.gamma);
    }
}

// ---------------------------------------------------------------------------------------------
// launch configuration
// ---------------------------------------------------------------------------------------------
struct Tune {
    int ry, cz, wy, pf;
    long blocks;
    bool force_generic, set;
};
static Tune g_tune = {0, 0, 0, 0, 0, false, false};
static const Tune &tune()
{
    if (!g_tune.set) {
        g_tune.set = true;
        const char *e = getenv("PDEHIP_FORCE_GENERIC");
        g_tune.force_generic = (e && e[0] == '1');
        // PDEHIP_TUNE="ry,cz,wy,pf,blocks" selects one of the instantiated tile shapes (tuning aid)
        if ((e = getenv("PDEHIP_TUNE")) != nullptr) sscanf(e, "%d,%d,%d,%d,%ld", &g_tune.ry, &g_tune.cz, &g_tune.wy, &g_tune.pf, &g_tune.blocks);
    }
    return g_tune;
}

template <typename T, int VEC, int RY, int CZ, int WY, int PF, int MODE, bool HAS_X>
static int launch_march(const LapArgs &a0, bool y_is_in, long want_blocks, hipStream_t st)
{
    LapArgs a = a0;
    a.ntz = (a.n2 + 64 * VEC * CZ - 1) / (64 * VEC * CZ);
    a.nty = (a.n1 + WY * RY - 1) / (WY * RY);
    // x-chunk length: enough workgroups to keep 256 CUs streaming, chunks as long as possible
    // (each chunk re-reads two halo planes)
    const long tiles = a.ntz * a.nty;
    long lx = a.n0;
    if (HAS_X) {
        long nxc = (want_blocks + tiles - 1) / tiles;
        if (nxc < 1) nxc = 1;
        if (nxc > a.n0) nxc = a.n0;
        lx = (a.n0 + nxc - 1) / nxc;
    }
    a.lx = (int)lx;
    a.nxc = (a.n0 + lx - 1) / lx;
    a.nblocks = a.nxc * tiles;
    // only the Euler epilogue reads a second array; on-the-fly BCs exist for the four laplace epilogues
    constexpr bool kHasY = (MODE == LAP_EULER);
    constexpr bool kIbc = (MODE <= LAP_CH_MU) || MODE == LAP_STAGE;
    // split rows (launch_laplace_t): the workgroups of the strip in front of those of the main part, a multiple of 8 (XCD mapping)
    a.strip_blocks = 0;
    if (a.strip_n2 > 0) {
        if (!(HAS_X && MODE <= LAP_CH_MU && !a.any_ibc)) PDEHIP_FAIL(E_RUNTIME, "internal: split rows reached a sweep without the strip");
        const long rs = 8L * a.strip_n2;   // rows per thread (lap_strip)
        const long threads = a.n0 * ((a.n1 + rs - 1) / rs) * a.strip_n2;
        a.strip_blocks = ((threads + 64 * WY - 1) / (64 * WY) + 7) / 8 * 8;
    }
    const dim3 grid((unsigned)(a.nblocks + a.strip_blocks)), block(64 * WY);
    if (a.any_ibc && !kIbc) PDEHIP_FAIL(E_RUNTIME, "internal: on-the-fly BCs are not built for the derivative epilogues");
    // rows that end inside a lane's vector, or tiles with whole chunks beyond the row, take the TAILS instance
    const bool tails = (a.n2 % VEC != 0) || (((a.n2 + 64L * VEC - 1) / (64L * VEC)) % CZ != 0);
    // streaming stores: 3-D outputs that do not fit the 256 MB Infinity Cache (a smaller field is re-read from that cache by the
    // next sweep); the whole-row tiles of aligned rows only (the hot instances)
    const int ncomp_out = (MODE == LAP_GRAD_C || MODE == LAP_GRAD_F || MODE == LAP_GRAD_B) ? 3 : (MODE == LAP_STAGE ? 2 : 1);
    static const bool nt_off = getenv("PDEHIP_NO_NT") != nullptr;   // A/B aid
    const bool nt = HAS_X && !tails && !nt_off && ((double)a.n0 * a.n1 * a.n2 * sizeof(T) * ncomp_out > 192.0 * 1048576.0);
    note_kernel("lap_march_kernel<%s,%d,RY=%d,CZ=%d,WY=%d,PF=%d,mode=%d,%s,%s,%s>", sizeof(T) == 8 ? "double" : "float", VEC, RY, CZ, WY, PF, MODE, HAS_X ? "3-D" : "2-D",
                tails ? "tails" : "aligned rows", (!tails && HAS_X && nt) ? "NT" : "plain stores");
#define PDEHIP_MARCH(YIN_, IBC_)                                                                                                       \
    do {                                                                                                                               \
        if (tails) hipLaunchKernelGGL((lap_march_kernel<T, VEC, RY, CZ, WY, PF, MODE, HAS_X, YIN_, IBC_, true>), grid, block, 0, st, a);  \
        else if (HAS_X && nt) hipLaunchKernelGGL((lap_march_kernel<T, VEC, RY, CZ, WY, PF, MODE, HAS_X, YIN_, IBC_, false, HAS_X>), grid, block, 0, st, a); \
        else hipLaunchKernelGGL((lap_march_kernel<T, VEC, RY, CZ, WY, PF, MODE, HAS_X, YIN_, IBC_, false>), grid, block, 0, st, a);       \
    } while (0)
    if (a.any_ibc) {
        if (y_is_in || !kHasY) PDEHIP_MARCH(true, kIbc);
        else PDEHIP_MARCH(!kHasY, kIbc);
    } else {
        if (y_is_in || !kHasY) PDEHIP_MARCH(true, false);
        else PDEHIP_MARCH(!kHasY, false);
    }
#undef PDEHIP_MARCH
    PDEHIP_HIP(hipGetLastError());
    return 0;
}


template <typename T, int MODE>
static int launch_laplace_t(const NGrid &n, const LapArgs &a, const OutStr &o, hipStream_t st)
{
    constexpr int VEC = 16 / sizeof(T);
    const Tune &tn = tune();
    // any row length: a row that ends inside a lane's vector is stored element-wise there (pdehip_march.inc); what the
    // vector kernel needs is 16-byte aligned rows on both sides — always true for full arrays, for valid (compact) output
    // only when the row length is a multiple of the vector
    const bool vec_ok = (o.s1 % VEC == 0) && (o.s0 % VEC == 0) && (o.off % VEC == 0) && (a.o_sc % VEC == 0) &&
                        (((uintptr_t)a.out) % 16 == 0) && (((uintptr_t)a.in) % 16 == 0) &&
                        (a.y == nullptr || ((uintptr_t)a.y) % 16 == 0) && stage_aligned(a);
    if (n.ndim >= 2 && vec_ok && !tn.force_generic) {
        // A row that is a few cells longer than a whole number of 64-lane chunks (513 = 4 x 128 + 1): the extra chunk would march every
        // plane for those few cells AND take the whole-row tile away from the rest (measured at 513^3: 0.48 of the peak against 0.65-0.71
        // at 512^3, profiles/r03_time_sizes.md).  Split: the aligned part of every row through the vectorised kernel - its last halo
        // column is the first cell of the remainder, real data in memory - and the remaining columns through the one-cell-per-thread
        // kernel (same expressions, cartesian.py:147-151 / :220-227).  Ghost cells from memory only (operators on a field whose faces are
        // set; the sweeps with on-the-fly faces of odd rows are the two-step kernel's, which has no such cliff).
        // (measured: 512 x 512 x 514 / 516 / 520 fp64 0.512 / 0.522 / 0.536 -> 0.408 / 0.416 / 0.434 ms next to 0.409 for 512^3 -
        // with a CONSTANT number of strip workgroups, lap_strip; with one workgroup per 64 x 8 cells four columns took 0.67 ms:
        // profiles/r04_time_sizes.md)
        // (wider: 512 x 512 x 524 / 528 / 536 / 544 fp64 0.48-0.50 -> 0.41-0.44 ms, fp32 0.28-0.29 -> 0.22-0.24; 48 and 64 columns still win for
        // fp64 but lose for fp32, 44 of 300 lose for both: 32)
        static const long split_max = getenv("PDEHIP_ROW_SPLIT") ? atol(getenv("PDEHIP_ROW_SPLIT")) : 32;
        // The same for a row that ends inside a lane's vector (511 cells of fp64, 2 per lane): without its last n2 % VEC columns the
        // row takes the instance without the element-wise tail bookkeeping (0.55 -> the rate of 510^3).
        const long cw = 64L * VEC;
        long tail = n.n[2] % cw;
        // (fp64 511^3: 0.584 -> 0.667 of the peak)
        if (tail == 0 || tail > split_max) tail = n.n[2] % VEC;
        // (3-D only: on a 2-D grid the second launch costs more than the narrow tiles - 4095 x 4097: 0.046 against 0.039 ms)
        // (a.strip_n2: this IS the aligned part of a split row - 131 = 130 + 1 by the vector rule must not split its 130 = 128 + 2 again)
        if (n.ndim == 3 && MODE <= LAP_CH_MU && !a.any_ibc && tail > 0 && split_max > 0 && n.n[2] - tail >= cw && !tn.ry && a.strip_n2 == 0) {
            NGrid nm = n;
            nm.n[2] -= tail;
            LapArgs am = a;
            am.n2 = nm.n[2];
            // the strip inside the launch of the main part (its workgroups first: lap_strip) - or, PDEHIP_ROW_SPLIT_SEPARATE=1 (A/B), as
            // a launch of its own behind it (44 us at 513^3 fp64 against ~10 us inside)
            static const bool separate = getenv("PDEHIP_ROW_SPLIT_SEPARATE") != nullptr;
            if (!separate) {
                am.strip_n2 = (int)tail;
                return launch_laplace_t<T, MODE>(nm, am, o, st);
            }
            PDEHIP_TRY((launch_laplace_t<T, MODE>(nm, am, o, st)));
            LapArgs as = a;
            as.n2 = tail;
            as.off += nm.n[2];
            as.o_off += nm.n[2];
            const long total = as.n0 * as.n1 * as.n2;
            const long blocks = (total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192;
            hipLaunchKernelGGL((lap_generic_kernel<T, MODE>), dim3((unsigned)blocks), dim3(256), 0, st, as);
            PDEHIP_HIP(hipGetLastError());
            return 0;
        }
        const bool y_is_in = (a.y == a.in) || a.y == nullptr;
        // chunks per row: cover the whole fastest axis with one wave where possible (contiguous
        // RY x row bytes per wave and plane is what the HBM write path likes) ...
        // ... but a row that needs 5 chunks must not pay for 8: among tiles of 4 / 2 / 1 chunks take the widest one with the
        // fewest chunks in total (513 cells: 5 x 1 instead of 2 x 4, where the second tile would march every plane for ONE
        // cell — measured 0.86 vs 0.48 ms at 513^3 / 512^3)
        const long chunks = (n.n[2] + 64 * VEC - 1) / (64 * VEC);
        int cz = 1;
        {
            long best = -1;
            for (int cand = 4; cand >= 1; cand /= 2) {
                const long padded = (chunks + cand - 1) / cand * cand;
                if (best < 0 || padded < best) { best = padded; cz = cand; }
            }
        }
        // measured on MI355X at 512^3 fp64 (profiles/r01_sweep_tiles.log): 2 rows x whole-row chunks, ~1024
        // single-wave workgroups (4 per CU) gives 0.41 ms per pass = 65 % of the 8 TB/s HBM peak
        // fp32 (fp64 registers, 4 cells per lane) is VALU-heavier: 4-row tiles measured 585 vs 454 Gcells/s at 512^3
        int ry = (n.ndim == 3 && sizeof(T) == 4 && cz >= 2) ? 4 : 2, wy = 1, pf = 1;   // 2-D: 2-row tiles too (4096^2: 79 % vs 57 % with 8 rows)
        long blocks = 1024;
        // ... but never starve the chip: small grids get smaller tiles until there are >= 512 wave
        // tiles (a 512^2 grid as 8-row x 512-cell tiles would be 64 waves on 256 CUs)
        auto n_tiles = [&](int ry_, int cz_) {
            const long per_plane = ((n.n[1] + ry_ - 1) / ry_) * ((n.n[2] + 64L * VEC * cz_ - 1) / (64L * VEC * cz_));
            return n.ndim == 3 ? per_plane * n.n[0] : per_plane;   // 3-D can also split along x
        };
        while (cz > 1 && n_tiles(ry, cz) < 512) cz /= 2;
        if (n.ndim == 3 && ry == 4 && cz < 2) ry = 2;   // only (4,4) and (4,2) are instantiated
        // Runge-Kutta stages with few pointwise streams are latency-bound at one wave per SIMD: two waves measured
        // 0.74 vs 0.91 ms (no earlier slope) and 1.11 vs 1.17 ms (one) at 512^3; from two slopes on one wave wins
        // (profiles/r01_time_rk.md)
        if (MODE == LAP_STAGE && (a.st_kind == 1 || !a.st_k[1])) blocks = 2048;
        // narrow tiles over a long row (e.g. 5 x 1 chunk for 513 cells): a wave carries 1/4 or 1/2 of the bytes of a whole-row
        // tile per plane, so keep the bytes in flight by marching shorter x-chunks with more waves
        if (chunks > cz && cz < 4) blocks *= 4 / cz;
        // Round 5: two planes of prefetch and half the waves (4 x-chunks of 128 planes) for the whole-row fp64 tile on fields beyond the Infinity
        // Cache: 0.3629-0.3662 against 0.3677-0.3800 ms per 512^3 Laplacian, three alternations (profiles/r05_lap_prefetch.log); the plain
        // epilogues only (the stage sweeps carry up to eight more streams: not measured)
        if (n.ndim == 3 && sizeof(T) == 8 && ry == 2 && cz == 4 && wy == 1 && chunks % 4 == 0 && n.n[2] % VEC == 0 && MODE <= LAP_CH_MU &&
            (double)n.n[0] * n.n[1] * n.n[2] * sizeof(T) > 400.0 * 1048576.0) { pf = 2; blocks = 512; }
        if (tn.ry) { ry = tn.ry; cz = tn.cz; wy = tn.wy; pf = tn.pf; blocks = tn.blocks; }
        // (two planes of prefetch exist for fp64 and the plain epilogues only: the fp32 stage instance would spill 20 bytes to scratch)
        constexpr bool kHasPf2 = sizeof(T) == 8 && MODE <= LAP_CH_MU;
        if (!kHasPf2) pf = 1;
#define PDEHIP_CFG3P2(RY_, CZ_) \
    if constexpr (kHasPf2) { if (n.ndim == 3 && ry == RY_ && cz == CZ_ && wy == 1 && pf == 2) return launch_march<T, VEC, RY_, CZ_, 1, 2, MODE, true>(a, y_is_in, blocks, st); }
#define PDEHIP_CFG3(RY_, CZ_, WY_, PF_) \
    if (n.ndim == 3 && ry == RY_ && cz == CZ_ && wy == WY_ && pf == PF_) return launch_march<T, VEC, RY_, CZ_, WY_, PF_, MODE, true>(a, y_is_in, blocks, st);
#define PDEHIP_CFG2(RY_, CZ_) \
    if (n.ndim == 2 && ry == RY_ && cz == CZ_) return launch_march<T, VEC, RY_, CZ_, 1, 1, MODE, false>(a, y_is_in, blocks, st);
        PDEHIP_CFG3(2, 4, 1, 1)
        PDEHIP_CFG3P2(2, 4)       // two planes of prefetch, one wave per SIMD (profiles/r05_lap_prefetch.log)
        PDEHIP_CFG3P2(2, 2)
        PDEHIP_CFG3(2, 2, 1, 1)
        PDEHIP_CFG3(2, 1, 1, 1)
        PDEHIP_CFG3(4, 4, 1, 1)
        PDEHIP_CFG3(4, 2, 1, 1)
        PDEHIP_CFG2(2, 4)
        PDEHIP_CFG2(2, 2)
        PDEHIP_CFG2(2, 1)
#undef PDEHIP_CFG3
#undef PDEHIP_CFG3P2
#undef PDEHIP_CFG2
        PDEHIP_FAIL(E_VALUE, "PDEHIP_TUNE selects a tile shape that is not instantiated (%d,%d,%d,%d)", ry, cz, wy, pf);
    }
    if (a.any_ibc && n.ndim != 1) PDEHIP_FAIL(E_RUNTIME, "internal: on-the-fly BCs requested for the generic kernel");
    if (MODE > LAP_CH_MU) PDEHIP_FAIL(E_RUNTIME, "internal: derivative and stage epilogues need the vectorised kernel");
    const long total = n.n[0] * n.n[1] * n.n[2];
    long blocks = (total + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL((lap_generic_kernel<T, MODE>), dim3((unsigned)blocks), dim3(256), 0, st, a);
    PDEHIP_HIP(hipGetLastError());
    return 0;
}

// true when launch_laplace would take the vectorised kernel (the only one with fused ghosts)
bool laplace_can_fuse_bcs(const NGrid &n, const void *in, const void *out, const void *y)
{
    return n.ndim >= 2 && !tune().force_generic && ((uintptr_t)in % 16 == 0) &&
           ((uintptr_t)out % 16 == 0) && (y == nullptr || (uintptr_t)y % 16 == 0);
}

// divergence through its own register-pipelined kernel (three input components, pdehip_div.inc)
template <typename T, int METHOD>
static int launch_div_t(const NGrid &n, DivArgs a, hipStream_t st)
{
    constexpr int VEC = 16 / sizeof(T);
    const long chunks = (n.n[2] + 64 * VEC - 1) / (64 * VEC);
    int cz = chunks >= 4 ? 4 : (chunks >= 2 ? 2 : 1);
    auto tiles_for = [&](int cz_) { return ((n.n[1] + 1) / 2) * ((n.n[2] + 64L * VEC * cz_ - 1) / (64L * VEC * cz_)); };
    while (cz > 1 && tiles_for(cz) * (n.ndim == 3 ? n.n[0] : 1) < 512) cz /= 2;
    const long tiles = tiles_for(cz);
    a.ntz = (n.n[2] + 64L * VEC * cz - 1) / (64L * VEC * cz);
    a.nty = (n.n[1] + 1) / 2;
    long lx = n.n[0];
    if (n.ndim == 3) {
        long nxc = (1024 + tiles - 1) / tiles;
        if (nxc < 1) nxc = 1;
        if (nxc > n.n[0]) nxc = n.n[0];
        lx = (n.n[0] + nxc - 1) / nxc;
    }
    a.lx = (int)lx;
    a.nxc = (n.n[0] + lx - 1) / lx;
    a.nblocks = a.nxc * tiles;
    const dim3 grid((unsigned)a.nblocks), block(64);
#define PDEHIP_DIV(CZ_)                                                                                             \
    if (cz == CZ_) {                                                                                                \
        if (n.ndim == 3) hipLaunchKernelGGL((div_march_kernel<T, VEC, 2, CZ_, METHOD, true>), grid, block, 0, st, a); \
        else hipLaunchKernelGGL((div_march_kernel<T, VEC, 2, CZ_, METHOD, false>), grid, block, 0, st, a);            \
    }
    PDEHIP_DIV(4) PDEHIP_DIV(2) PDEHIP_DIV(1)
#undef PDEHIP_DIV
    PDEHIP_HIP(hipGetLastError());
    return 0;
}

int launch_div_march(const NGrid &n, int method, const void *in, void *out, const OutStr &o, hipStream_t st, bool *done)
{
    *done = false;
    const long vec = 16 / elem_size(n.dtype);
    const bool vec_ok = n.ndim >= 2 && (n.n[2] % vec == 0) && (o.s1 % vec == 0) && (o.s0 % vec == 0) && (o.off % vec == 0) &&
                        (n.pc % vec == 0) && ((uintptr_t)out % 16 == 0) && ((uintptr_t)in % 16 == 0) && !tune().force_generic;
    if (!vec_ok) return 0;
    DivArgs a;
    memset(&a, 0, sizeof(a));
    a.in = in; a.out = out;
    a.n0 = n.n[0]; a.n1 = n.n[1]; a.n2 = n.n[2];
    a.p0 = n.p[0]; a.p1 = n.p[1]; a.pc = n.pc; a.off = n.off;
    a.o_off = o.off; a.o_s0 = o.s0; a.o_s1 = o.s1;
    for (int q = 0; q < 3; q++) a.gs[q] = (method == PDEHIP_CENTRAL) ? 0.5 / n.dx[q] : 1 / n.dx[q];   // cartesian.py:876-879, :928-931
    *done = true;
#define PDEHIP_DM(T)                                                                   \
    switch (method) {                                                                  \
    case PDEHIP_CENTRAL: return launch_div_t<T, PDEHIP_CENTRAL>(n, a, st);             \
    case PDEHIP_FORWARD: return launch_div_t<T, PDEHIP_FORWARD>(n, a, st);             \
    case PDEHIP_BACKWARD: return launch_div_t<T, PDEHIP_BACKWARD>(n, a, st);           \
    default: PDEHIP_FAIL(E_VALUE, "Unknown derivative type `%d`", method);             \
    }
    if (n.dtype == PDEHIP_F64) { PDEHIP_DM(double) }
    PDEHIP_DM(float)
#undef PDEHIP_DM
}

// gradient / gradient_squared through the register-pipelined kernel (same loads as the Laplacian,
// different epilogue).  Returns 1 without launching when the shape needs the generic derivative kernels.
int launch_deriv_march(const NGrid &n, const void *in, void *out, const OutStr &o, int mode, const double *gs, hipStream_t st, bool *done)
{
    *done = false;
    const long vec = 16 / elem_size(n.dtype);
    const bool vec_ok = n.ndim >= 2 && (o.s1 % vec == 0) && (o.s0 % vec == 0) && (o.off % vec == 0) &&
                        (o.sc % vec == 0) && ((uintptr_t)out % 16 == 0) && ((uintptr_t)in % 16 == 0) && !tune().force_generic;
    if (!vec_ok) return 0;
    LapArgs a;
    memset(&a, 0, sizeof(a));
    a.in = in; a.out = out; a.y = nullptr;
    a.n0 = n.n[0]; a.n1 = n.n[1]; a.n2 = n.n[2];
    a.p0 = n.p[0]; a.p1 = n.p[1]; a.off = n.off;
    a.o_off = o.off; a.o_s0 = o.s0; a.o_s1 = o.s1; a.o_sc = o.sc;
    for (int q = 0; q < 3; q++) a.gs[q] = gs[q];
    a.ndim = n.ndim; a.lx = 1;
    *done = true;
#define PDEHIP_DMODE(T)                                                            \
    switch (mode) {                                                                \
    case LAP_GRAD_C: return launch_laplace_t<T, LAP_GRAD_C>(n, a, o, st);          \
    case LAP_GRAD_F: return launch_laplace_t<T, LAP_GRAD_F>(n, a, o, st);          \
    case LAP_GRAD_B: return launch_laplace_t<T, LAP_GRAD_B>(n, a, o, st);          \
    case LAP_GRADSQ_C: return launch_laplace_t<T, LAP_GRADSQ_C>(n, a, o, st);      \
    case LAP_GRADSQ_N: return launch_laplace_t<T, LAP_GRADSQ_N>(n, a, o, st);      \
    default: PDEHIP_FAIL(E_VALUE, "unknown derivative mode %d", mode);             \
    }
    if (n.dtype == PDEHIP_F64) { PDEHIP_DMODE(double) }
    PDEHIP_DMODE(float)
#undef PDEHIP_DMODE
}

int launch_laplace(const NGrid &n, const void *in, void *out, const OutStr &o, int mode, double s1,
                   double s2, double gamma, const void *y, hipStream_t st, const InputBCs *fg, const StageFuse *stage)
{
    if ((mode == LAP_STAGE) != (stage != nullptr)) PDEHIP_FAIL(E_RUNTIME, "internal: stage epilogue without / with a stage descriptor");
    if (!in || (!out && !(stage && (stage->kind == 1 || stage->kind == 2 || stage->kind == 4)))) PDEHIP_FAIL(E_VALUE, "laplace: NULL array pointer");
    if (mode == LAP_EULER && !y) PDEHIP_FAIL(E_VALUE, "laplace_euler: y is NULL");
    LapArgs a;
    memset(&a, 0, sizeof(a));
    a.in = in; a.out = out; a.y = (mode == LAP_EULER) ? y : nullptr;
    a.n0 = n.n[0]; a.n1 = n.n[1]; a.n2 = n.n[2];
    a.p0 = n.p[0]; a.p1 = n.p[1]; a.off = n.off;
    a.o_off = o.off; a.o_s0 = o.s0; a.o_s1 = o.s1;
    a.sx = n.lap_scale[0]; a.sy = n.lap_scale[1]; a.sz = n.lap_scale[2];
    a.s1 = s1; a.s2 = s2; a.gamma = gamma;
    a.ndim = n.ndim; a.lx = 1;
    if (fg) {
        for (int ax = 0; ax < 3; ax++)
            for (int side = 0; side < 2; side++) {
                a.ibc[ax][side].on = fg->on[ax][side];
                a.ibc[ax][side].idx = fg->idx[ax][side];
                a.ibc[ax][side].c = fg->c[ax][side];
                a.ibc[ax][side].f = fg->f[ax][side];
                a.any_ibc |= fg->on[ax][side];
            }
    }
    if (stage) {
        if (!stage->y || !stage->out2 || stage->out2 == in || out == in) PDEHIP_FAIL(E_VALUE, "stage epilogue: NULL or aliased array pointer");
        a.st_kind = stage->kind; a.st_y = stage->y; a.st_out = stage->out2;
        int nk = 0;
        for (int m = 0; m < 5 && stage->k[m]; m++, nk++) { a.st_k[m] = stage->k[m]; a.st_c[m] = stage->c[m]; }
        if (stage->kind == 1 && nk != 3) PDEHIP_FAIL(E_RUNTIME, "internal: the RK4 update needs three earlier slopes");
        if (stage->kind == 2 && (nk != 4 || !stage->err)) PDEHIP_FAIL(E_RUNTIME, "internal: the RKF45 update needs four earlier slopes and the error cell");
        if (stage->kind == 3 && nk != 1) PDEHIP_FAIL(E_RUNTIME, "internal: the Adams-Bashforth update needs the previous rate");
        if (stage->kind == 4 && (nk != 2 || !stage->err || stage->k[1] != in)) PDEHIP_FAIL(E_RUNTIME, "internal: the adaptive Euler update needs the rate, the half step as the input of the sweep and the error cell");
        a.st_err = stage->err;
        a.st_c[5] = stage->c_new;
    }
#define PDEHIP_MODE_SWITCH(T)                                                     \
    switch (mode) {                                                               \
    case LAP_STAGE: return launch_laplace_t<T, LAP_STAGE>(n, a, o, st);           \
    case LAP_PLAIN: return launch_laplace_t<T, LAP_PLAIN>(n, a, o, st);           \
    case LAP_SCALED: return launch_laplace_t<T, LAP_SCALED>(n, a, o, st);         \
    case LAP_EULER: return launch_laplace_t<T, LAP_EULER>(n, a, o, st);           \
    case LAP_CH_MU: return launch_laplace_t<T, LAP_CH_MU>(n, a, o, st);           \
    default: PDEHIP_FAIL(E_VALUE, "unknown laplace mode %d", mode);               \
    }
    if (n.dtype == PDEHIP_F64) { PDEHIP_MODE_SWITCH(double) }
    PDEHIP_MODE_SWITCH(float)
#undef PDEHIP_MODE_SWITCH
}

// (the two-step sweeps: pdehip_kernels_e2.hip; the LDS-tiled 2-D sweeps: pdehip_kernels_t2.hip - separate translation units, so that an
// edit of one family rebuilds in two minutes instead of six)
bool force_generic_kernels() { return tune().force_generic; }

// ---------------------------------------------------------------------------------------------
// ghost cells: one launch for all faces.  Faces only read interior cells and write disjoint
// ghost cells, so the reference's axis-by-axis order (numba/backend.py:335-340) cannot be
// observed and all faces are processed concurrently.
// ---------------------------------------------------------------------------------------------
struct GhostFace {
    int kind, flags, ax, side;
    long index1, index2;
    double c, f1, f2;
    const double *ca, *f1a, *f2a;
    long m1, m2;          // extent of the two other (normalised) axes
    long q1, q2, g1, g2;  // their pitches and ghost widths
    long pa;              // pitch along the BC axis
    long ghost;           // full index of the ghost layer along the axis
    long start;           // first work item of this face
    int comp_axis;        // vector component addressed by a `normal` BC
};
struct GhostArgs {
    GhostFace f[6];
    int nfaces, ncomp;
    long pc, lpad_off;    // component pitch, offset of column 0 of interior rows (lpad - gh2)
    long total;
};

template <typename T>
__global__ void __launch_bounds__(256) ghost_kernel(GhostArgs a, T *data)
{
    for (long t = blockIdx.x * (long)blockDim.x + threadIdx.x; t < a.total; t += (long)gridDim.x * blockDim.x) {
        int fi = 0;
#pragma unroll
        for (int q = 1; q < 6; q++)
            if (q < a.nfaces && t >= a.f[q].start) fi = q;
        const GhostFace &f = a.f[fi];
        long loc = t - f.start;
        const long v = loc % f.m2;
        loc /= f.m2;
        const long u = loc % f.m1;
        const int comp = (int)(loc / f.m1);
        if ((f.flags & PDEHIP_BCF_NORMAL) && comp != f.comp_axis) continue;
        T *d = data + (long)comp * a.pc + a.lpad_off;
        const long base = (u + f.g1) * f.q1 + (v + f.g2) * f.q2;
        double cst, f1, f2 = 0;
        if (f.flags & PDEHIP_BCF_ARRAYS) {
            const long fc = (long)((f.flags & PDEHIP_BCF_NORMAL) ? 0 : comp) * f.m1 * f.m2 + u * f.m2 + v;
            cst = f.ca[fc];
            f1 = f.f1a[fc];
            if (f.kind == PDEHIP_BC_ORDER2) f2 = f.f2a[fc];
        } else {
            cst = f.c; f1 = f.f1; f2 = f.f2;
        }
        // local.py:1636  const + factor * data[index + 1]
        double r = cst + f1 * (double)d[base + (f.index1 + 1) * f.pa];
        if (f.kind == PDEHIP_BC_ORDER2) r = r + f2 * (double)d[base + (f.index2 + 1) * f.pa];  // local.py:2055-2059
        d[base + f.ghost * f.pa] = (T)r;
    }
}

// The HIP runtime loads a translation unit's code object when its first kernel is launched.  For this file that is ~10 MB
// of code (every tile shape x epilogue of the stencil kernels), and loading it in the middle of a run - after hiprtc
// modules of expression PDEs had been loaded and unloaded - ended in "Memory access fault by GPU ... on address (nil)"
// during the load in about 4 of 10 processes (MI355X, ROCm 7.0.2 runtime; the fault is raised before the kernel that
// triggered the load is dispatched).  Loaded right after the device is selected, it never did: pdehip_set_device calls
// this once per process.
int preload_stencil_kernels()
{
    hipFuncAttributes attr;
    PDEHIP_HIP(hipFuncGetAttributes(&attr, reinterpret_cast<const void *>(&ghost_kernel<double>)));
    return 0;
}

int launch_ghosts(const NGrid &n, int ncomp, const pdehip_bc_face_t *faces, void *data, hipStream_t st)
{
    if (!faces || !data) PDEHIP_FAIL(E_VALUE, "set_ghost_cells: NULL pointer");
    if (ncomp < 1) PDEHIP_FAIL(E_VALUE, "set_ghost_cells: ncomp must be >= 1");
    GhostArgs a;
    memset(&a, 0, sizeof(a));
    a.ncomp = ncomp;
    a.pc = n.pc;
    a.lpad_off = n.lpad - n.gh[2];
    long total = 0;
    int nf = 0;
    for (int ar = 0; ar < n.ndim; ar++) {
        const int ax = 3 - n.ndim + ar;
        for (int side = 1; side >= 0; side--) {
            const pdehip_bc_face_t &s = faces[2 * ar + side];
            if (s.kind == PDEHIP_BC_SKIP) continue;
            if (s.kind != PDEHIP_BC_ORDER1 && s.kind != PDEHIP_BC_ORDER2)
                PDEHIP_FAIL(E_VALUE, "unknown BC kind %d on axis %d", s.kind, ar);
            if (s.index1 < 0 || s.index1 >= n.n[ax] || (s.kind == PDEHIP_BC_ORDER2 && (s.index2 < 0 || s.index2 >= n.n[ax])))
                PDEHIP_FAIL(E_VALUE, "BC index out of range on axis %d", ar);
            if ((s.flags & PDEHIP_BCF_ARRAYS) && (!s.const_arr || !s.factor1_arr || (s.kind == PDEHIP_BC_ORDER2 && !s.factor2_arr)))
                PDEHIP_FAIL(E_VALUE, "BC arrays missing on axis %d", ar);
            GhostFace &f = a.f[nf++];
            const int o1 = (ax == 0) ? 1 : 0, o2 = (ax == 2) ? 1 : 2;
            f.kind = s.kind; f.flags = s.flags; f.ax = ax; f.side = side;
            f.index1 = s.index1; f.index2 = s.index2;
            f.c = s.const_v; f.f1 = s.factor1; f.f2 = s.factor2;
            f.ca = s.const_arr; f.f1a = s.factor1_arr; f.f2a = s.factor2_arr;
            f.m1 = n.n[o1]; f.m2 = n.n[o2];
            f.q1 = n.p[o1]; f.q2 = n.p[o2]; f.g1 = n.gh[o1]; f.g2 = n.gh[o2];
            f.pa = n.p[ax];
            f.ghost = side ? n.n[ax] + 1 : 0;
            f.start = total;
            f.comp_axis = ar;
            total += (long)ncomp * f.m1 * f.m2;
        }
    }
    if (nf == 0) return 0;
    a.nfaces = nf;
    a.total = total;
    // pitches above are relative to column 0 of the *compact* row; the interior columns of the
    // device layout start at lpad, the ghost column at lpad-1: shift via lpad_off = lpad - gh2
    long blocks = (total + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    if (n.dtype == PDEHIP_F64)
        hipLaunchKernelGGL((ghost_kernel<double>), dim3((unsigned)blocks), dim3(256), 0, st, a, (double *)data);
    else
        hipLaunchKernelGGL((ghost_kernel<float>), dim3((unsigned)blocks), dim3(256), 0, st, a, (float *)data);
    PDEHIP_HIP(hipGetLastError());
    return 0;
}

}  // namespace PDEHIP_VARIANT_NS
}  // namespace pdehip
