// pdehip_jit.hip — run-time specialisation of the register-pipelined stencil kernel for arbitrary
// pointwise right-hand sides (the generic `PDE({...})` expressions of pde/pdes/pde.py:299-499).
//
// The reference turns a sympy expression into numba code (pde/tools/expressions.py:361-388,
// pde/pdes/pde.py:401-499).  Here the host (pde_hip/expr.py) turns it into the body of
//     double pde_epilogue(double c, double lap, double gsq, double e0, double e1, double e2, const double *p, const PdeDer &d)
// and this file compiles lap_march_body<..., LAP_CUSTOM, ...> around it with hiprtc: one pass reads
// the stencil array once, evaluates laplace / gradient_squared in registers and applies the generated
// pointwise function (plus up to three more arrays at the same cell) — no temporaries, the same HBM
// traffic as the hand-fused kernels.  Kernels are cached per (tile shape, dtype, dimension, IBC).
// hiprtc is resolved with dlopen at first use; the kernel sources are embedded at build time.
#include <dlfcn.h>

#include <map>
#include <mutex>
#include <vector>

#include "pdehip_common.h"
#include "pdehip_rk_loops.h"
#include "pdehip_sources.h"   // generated: kDeviceH, kMarchInc (raw string literals)

using namespace pdehip;

namespace {

typedef struct _hiprtcProgram *hiprtcProgram;
struct Rtc {
    void *handle = nullptr;
    int (*CreateProgram)(hiprtcProgram *, const char *, const char *, int, const char **, const char **) = nullptr;
    int (*CompileProgram)(hiprtcProgram, int, const char **) = nullptr;
    int (*GetProgramLogSize)(hiprtcProgram, size_t *) = nullptr;
    int (*GetProgramLog)(hiprtcProgram, char *) = nullptr;
    int (*GetCodeSize)(hiprtcProgram, size_t *) = nullptr;
    int (*GetCode)(hiprtcProgram, char *) = nullptr;
    int (*DestroyProgram)(hiprtcProgram *) = nullptr;
};
Rtc g_rtc;

int load_rtc()
{
    if (g_rtc.handle) return 0;
    const char *cands[] = {getenv("PDEHIP_HIPRTC"), "/opt/rocm/lib/libhiprtc.so", "libhiprtc.so", "libhiprtc.so.7"};
    void *h = nullptr;
    for (const char *c : cands)
        if (c && c[0] && (h = dlopen(c, RTLD_NOW | RTLD_LOCAL))) break;
    if (!h) PDEHIP_FAIL(E_RUNTIME, "cannot load hiprtc (set PDEHIP_HIPRTC to libhiprtc.so): %s", dlerror());
#define PDEHIP_SYM(field, name)                                                    \
    g_rtc.field = reinterpret_cast<decltype(g_rtc.field)>(dlsym(h, name));          \
    if (!g_rtc.field) PDEHIP_FAIL(E_RUNTIME, "hiprtc symbol %s not found", name)
    PDEHIP_SYM(CreateProgram, "hiprtcCreateProgram");
    PDEHIP_SYM(CompileProgram, "hiprtcCompileProgram");
    PDEHIP_SYM(GetProgramLogSize, "hiprtcGetProgramLogSize");
    PDEHIP_SYM(GetProgramLog, "hiprtcGetProgramLog");
    PDEHIP_SYM(GetCodeSize, "hiprtcGetCodeSize");
    PDEHIP_SYM(GetCode, "hiprtcGetCode");
    PDEHIP_SYM(DestroyProgram, "hiprtcDestroyProgram");
#undef PDEHIP_SYM
    g_rtc.handle = h;
    return 0;
}

struct Variant {
    hipModule_t module = nullptr;
    hipFunction_t fn = nullptr;
};
struct Jit {
    std::string body;                       // statements of pde_epilogue
    std::string body2;                      // statements of pde_epilogue2 (level 2 of a fused two-pass expression)
    std::map<std::string, Variant> cache;   // key: "T,VEC,RY,CZ,HASX,IBC" or "generic,T"
};
std::map<std::string, Variant> g_shared;   // epilogue source(s) + variant key -> loaded kernel, for the life of the process
std::mutex g_shared_mutex;

// the generated code reads PdeDer (`d.d1[..]`, `d.d2[..]`, `d.gr[..]`): only the one-level kernels supply it
bool uses_axis_derivatives(const Jit *j)
{
    for (const char *key : {"d.d1", "d.d2", "d.gr"})
        if (j->body.find(key) != std::string::npos || j->body2.find(key) != std::string::npos) return true;
    return false;
}

const char *kGenericKernel = R"SRC(
// one cell per thread: any shape / 1-D / odd row lengths (ghost cells must be set)
extern "C" __global__ void __launch_bounds__(256) pde_kernel(pdehip::LapArgs a)
{
    using namespace pdehip;
    typedef PDE_T T;
    const long total = a.n0 * a.n1 * a.n2;
    const T *in = (const T *)a.in;
    T *out = (T *)a.out;
    for (long t = blockIdx.x * (long)blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
        const long k = t % a.n2, j = (t / a.n2) % a.n1, i = t / (a.n2 * a.n1);
        const long e = a.off + i * a.p0 + j * a.p1 + k;
        const T *c = in + e;
        const double mid = (double)c[0];
        double lap, gsq;
        PdeDer d = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
        const double vm = 2 * mid;
        double zl = (double)c[-1], zr = (double)c[1];
        if (a.ndim == 1 && a.any_ibc) {   // 1-D: the two virtual points on the fly (as in lap_generic_kernel)
            if (k == 0 && a.ibc[2][0].on) zl = (double)(T)(a.ibc[2][0].c + a.ibc[2][0].f * (double)in[a.off + a.ibc[2][0].idx]);
            if (k == a.n2 - 1 && a.ibc[2][1].on) zr = (double)(T)(a.ibc[2][1].c + a.ibc[2][1].f * (double)in[a.off + a.ibc[2][1].idx]);
        }
        const double dz = zr - zl;
        d.d1[2] = dz / a.dd1[2];
        d.d2[2] = (zr - vm + zl) * a.dd2[2];
        d.gr[2] = dz * a.dg[2];
        if (a.ndim == 1) {
            lap = (zl - 2 * mid + zr) * a.sz;
            gsq = dz * dz * a.gs[2];
        } else if (a.ndim == 2) {
            const double dy = (double)c[a.p1] - (double)c[-a.p1];
            lap = ((double)c[-a.p1] - 2 * mid + (double)c[a.p1]) * a.sy + ((double)c[-1] - 2 * mid + (double)c[1]) * a.sz;
            gsq = dy * dy * a.gs[1] + dz * dz * a.gs[2];
            d.d1[1] = dy / a.dd1[1];
            d.d2[1] = ((double)c[a.p1] - vm + (double)c[-a.p1]) * a.dd2[1];
            d.gr[1] = dy * a.dg[1];
        } else {
            const double dx = (double)c[a.p0] - (double)c[-a.p0], dy = (double)c[a.p1] - (double)c[-a.p1];
            lap = ((double)c[-a.p0] - vm + (double)c[a.p0]) * a.sx + ((double)c[-a.p1] - vm + (double)c[a.p1]) * a.sy +
                  ((double)c[-1] - vm + (double)c[1]) * a.sz;
            gsq = dx * dx * a.gs[0] + dy * dy * a.gs[1] + dz * dz * a.gs[2];
            d.d1[0] = dx / a.dd1[0];
            d.d2[0] = ((double)c[a.p0] - vm + (double)c[-a.p0]) * a.dd2[0];
            d.d1[1] = dy / a.dd1[1];
            d.d2[1] = ((double)c[a.p1] - vm + (double)c[-a.p1]) * a.dd2[1];
            d.gr[0] = dx * a.dg[0];
            d.gr[1] = dy * a.dg[1];
        }
        double ex[3];
        for (int m = 0; m < 3; m++) ex[m] = a.ex[m] ? (double)((const T *)a.ex[m])[e] : 0.0;
        out[a.o_off + i * a.o_s0 + j * a.o_s1 + k] = (T)pde_epilogue(mid, lap, gsq, ex[0], ex[1], ex[2], a.par, d);
    }
}
)SRC";

const char *kMarchWrapper = R"SRC(
extern "C" __global__ void __launch_bounds__(64 * PDE_WY) pde_kernel(pdehip::LapArgs a)
{
    pdehip::lap_march_body<PDE_T, PDE_VEC, PDE_RY, PDE_CZ, PDE_WY, 1, pdehip::LAP_CUSTOM, PDE_HASX, true, PDE_IBC, PDE_TAILS>(a);
}
)SRC";

// two-level kernel (pdehip_march2.inc) around the same epilogue: two Euler steps of a one-pass expression per sweep
const char *kMarch2Wrapper = R"SRC(
extern "C" __global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) pde_kernel(pdehip::LapArgs a)
{
    pdehip::euler2_body<PDE_T, PDE_VEC, PDE_RY, PDE_M2, PDE_HASX, true, false, false>(a);
}
)SRC";

// the multi-step 2-D kernel (time levels in LDS, pdehip_tile2d.inc) around the generated update; PDE_CZ = tile columns
constexpr int kTileVariant = 100;
const char *kTileWrapper = R"SRC(
extern "C" __global__ void __launch_bounds__(1024) pde_kernel(pdehip::Tile2Args a)
{
    pdehip::tile2d_body<PDE_T, PDE_RY, 32, PDE_CZ, 8>(a);   // PDE_RY carries the mode: 2 one field, 3 two coupled fields
}
)SRC";

// cache key of a variant in the current arithmetic mode (pdehip_set_fastmath)
std::string mode_key(const std::string &key) { return (fastmath_on() && !key.empty()) ? key + "#fast" : key; }

int compile_variant(Jit *j, const std::string &key_in, bool generic, const char *tname, int vec, int ry, int cz, bool hasx, bool ibc, Variant *out,
                    int two_level = 0,    // 0: one-level kernel, E2_CUSTOM / E2_CUSTOM2: two-level kernel
                    bool stage = false,   // one-level kernel followed by the Runge-Kutta stage epilogue (LapArgs::st_*)
                    int wy = 1,           // waves per workgroup (stacked along the rows), one-level kernel
                    bool tails = true)    // rows that end inside a vector / leave idle chunks in a tile (pdehip_march.inc)
{
    PDEHIP_TRY(load_rtc());
    const std::string key = mode_key(key_in);   // (the exact and the contracted build of a variant are different kernels)
    // kernels already built in this process for the same epilogue(s) and variant: every eq.solve creates new handles for the
    // same expressions, and a hiprtc build costs 50-150 ms (a 4000-step run of a 256^2 grid spent 9/10 of its wall time there)
    const std::string shared_key = j->body + '\x01' + j->body2 + '\x01' + key;
    if (out && !key.empty()) {
        std::lock_guard<std::mutex> guard(g_shared_mutex);
        auto it = g_shared.find(shared_key);
        if (it != g_shared.end()) {
            *out = it->second;
            j->cache[key] = *out;
            return 0;
        }
    }
    std::string src = "#define PDEHIP_JIT 1\n#include \"pdehip_device.h\"\nnamespace pdehip {\n"
                      "__device__ __forceinline__ double pde_epilogue(double c, double lap, double gsq, double e0, double e1, double e2, const double *p, const PdeDer &d)\n{\n";
    src += j->body;
    src += "\n}\n";
    if (two_level == E2_CUSTOM2 || (two_level == kTileVariant && !j->body2.empty())) {
        src += "__device__ __forceinline__ double pde_epilogue2(double c, double lap, double gsq, double e0, double e1, double e2, const double *p, const PdeDer &d)\n{\n";
        src += j->body2;
        src += "\n}\n";
    }
    if (two_level == kTileVariant) src += "#include \"pdehip_tile2d.inc\"\n";
    else if (two_level) src += "#include \"pdehip_march2.inc\"\n";   // PDE_HASX carries HAS_Y there
    else if (!generic) src += "#include \"pdehip_march.inc\"\n";
    src += "}  // namespace pdehip\n";
    src += two_level == kTileVariant ? kTileWrapper : (two_level ? kMarch2Wrapper : (generic ? kGenericKernel : kMarchWrapper));
    const char *hdr_src[] = {kDeviceH, kMarchInc, kMarch2Inc, kTile2dInc};
    const char *hdr_name[] = {"pdehip_device.h", "pdehip_march.inc", "pdehip_march2.inc", "pdehip_tile2d.inc"};
    hiprtcProgram prog = nullptr;
    if (g_rtc.CreateProgram(&prog, src.c_str(), "pde_kernel.hip", 4, hdr_src, hdr_name) != 0)
        PDEHIP_FAIL(E_RUNTIME, "hiprtcCreateProgram failed");
    std::vector<std::string> opts = {"--offload-arch=gfx950", "-O3", "-std=c++17", fastmath_on() ? "-ffp-contract=fast" : "-ffp-contract=off",
                                     std::string("-DPDE_T=") + tname, "-DPDE_VEC=" + std::to_string(vec), "-DPDE_RY=" + std::to_string(ry),
                                     "-DPDE_CZ=" + std::to_string(cz), std::string("-DPDE_HASX=") + (hasx ? "true" : "false"),
                                     std::string("-DPDE_IBC=") + (ibc ? "true" : "false"), "-DPDE_M2=" + std::to_string(two_level == kTileVariant ? 0 : two_level),
                                     std::string("-DPDE_STAGE=") + (stage ? "1" : "0"), "-DPDE_WY=" + std::to_string(wy),
                                     std::string("-DPDE_TAILS=") + (tails ? "true" : "false")};
    std::vector<const char *> copts;
    for (auto &o : opts) copts.push_back(o.c_str());
    const int rc = g_rtc.CompileProgram(prog, (int)copts.size(), copts.data());
    if (rc != 0) {
        size_t n = 0;
        g_rtc.GetProgramLogSize(prog, &n);
        std::string log(n + 1, '\0');
        if (n) g_rtc.GetProgramLog(prog, &log[0]);
        g_rtc.DestroyProgram(&prog);
        PDEHIP_FAIL(E_VALUE, "expression kernel does not compile: %.400s", log.c_str());
    }
    size_t n = 0;
    g_rtc.GetCodeSize(prog, &n);
    std::vector<char> code(n);
    g_rtc.GetCode(prog, code.data());
    g_rtc.DestroyProgram(&prog);
    if (!out) return 0;   // compile check only (no device needed)
    PDEHIP_HIP(hipModuleLoadData(&out->module, code.data()));
    PDEHIP_HIP(hipModuleGetFunction(&out->fn, out->module, "pde_kernel"));
    j->cache[key] = *out;
    if (!key.empty()) {
        std::lock_guard<std::mutex> guard(g_shared_mutex);
        g_shared[shared_key] = *out;
    }
    return 0;
}

}  // namespace

extern "C" {

int pdehip_jit_create(const char *epilogue_body, void **handle)
{
    if (!epilogue_body || !handle) PDEHIP_FAIL(E_VALUE, "jit_create: NULL pointer");
    Jit *j = new Jit();
    j->body = epilogue_body;
    *handle = j;
    return 0;
}

// compile (but do not load) the kernels an expression needs: usable without a GPU (hiprtc
// cross-compiles), used by the CPU test-suite to validate generated epilogues
int pdehip_jit_check(void *handle, int dtype, int ndim)
{
    if (!handle) PDEHIP_FAIL(E_VALUE, "jit_check: NULL handle");
    if (dtype != PDEHIP_F64 && dtype != PDEHIP_F32) PDEHIP_FAIL(E_NOTIMPL, "unsupported dtype code %d", dtype);
    Jit *j = static_cast<Jit *>(handle);
    const char *tname = dtype == PDEHIP_F64 ? "double" : "float";
    const int vec = dtype == PDEHIP_F64 ? 2 : 4;
    PDEHIP_TRY(compile_variant(j, "", true, tname, 1, 1, 1, false, false, nullptr));
    if (ndim >= 2) PDEHIP_TRY(compile_variant(j, "", false, tname, vec, 2, 2, ndim == 3, true, nullptr));
    if (ndim >= 2) PDEHIP_TRY(compile_variant(j, "", false, tname, vec, 1, 4, ndim == 3, true, nullptr, 0, true));   // stage sweeps: 1-row tiles
    if (ndim >= 2) PDEHIP_TRY(compile_variant(j, "", false, tname, vec, ndim == 3 ? 2 : 1, 1, ndim == 3, true, nullptr,
                                              j->body2.empty() ? E2_CUSTOM : E2_CUSTOM2));
    if (ndim == 2) PDEHIP_TRY(compile_variant(j, "", false, tname, vec, j->body2.empty() ? 2 : 3, 64, false, true, nullptr, kTileVariant));
    return 0;
}

// a fused two-pass expression: level 1 = body1 applied to the state, level 2 = body2 applied to the result of level 1
// with e0 = the state at the cell (see pdehip_jit_fused2)
int pdehip_jit_create2(const char *body1, const char *body2, void **handle)
{
    if (!body1 || !body2 || !handle) PDEHIP_FAIL(E_VALUE, "jit_create2: NULL pointer");
    Jit *j = new Jit();
    j->body = body1;
    j->body2 = body2;
    *handle = j;
    return 0;
}

int pdehip_jit_destroy(void *handle)
{
    if (!handle) return 0;
    // (the modules stay loaded: they are shared through g_shared by every handle with the same epilogue)
    delete static_cast<Jit *>(handle);
    return 0;
}

// Apply the BCs `in_faces` (NULL: ghost cells are already set) to `in_full`, then evaluate
//   out = pde_epilogue(in, laplace(in), gradient_squared(in), extra[0], extra[1], extra[2], params)
// on every interior cell (all arrays FULL, same grid).
}  // extern "C"

namespace {
// stage != NULL: the generated epilogue is the slope of a Runge-Kutta stage and the combination that follows it is
// computed in the same sweep (StageFuse, pdehip_common.h); *done = 0 and nothing launched when the vectorised kernel
// does not cover the grid (the caller then runs pdehip_jit_apply + lincomb / combine)
int jit_apply_impl(void *handle, const pdehip_grid_t *g, void *in_full, const void *const *extra3_host, void *out_full,
                   const double *params_host, int nparams, const pdehip_bc_face_t *in_faces, void *stream,
                   const StageFuse *stage, int *done)
{
    if (done) *done = 0;
    if (!handle || !in_full || (!out_full && !(stage && (stage->kind == 1 || stage->kind == 2 || stage->kind == 4)))) PDEHIP_FAIL(E_VALUE, "jit_apply: NULL pointer");
    if (nparams < 0 || nparams > 12) PDEHIP_FAIL(E_VALUE, "jit_apply: at most 12 scalar parameters");
    Jit *j = static_cast<Jit *>(handle);
    NGrid n;
    PDEHIP_TRY(norm_grid(g, &n));
    const bool f64 = n.dtype == PDEHIP_F64;
    const int vec = f64 ? 2 : 4;
    const OutStr o = out_strides(n, PDEHIP_OUT_FULL);
    LapArgs a;
    memset(&a, 0, sizeof(a));
    a.in = in_full; a.out = out_full;
    a.n0 = n.n[0]; a.n1 = n.n[1]; a.n2 = n.n[2];
    a.p0 = n.p[0]; a.p1 = n.p[1]; a.off = n.off;
    a.o_off = o.off; a.o_s0 = o.s0; a.o_s1 = o.s1; a.o_sc = o.sc;
    a.sx = n.lap_scale[0]; a.sy = n.lap_scale[1]; a.sz = n.lap_scale[2];
    for (int q = 0; q < 3; q++) a.gs[q] = 0.25 / (n.dx[q] * n.dx[q]);   // gradient_squared, central (cartesian.py:661)
    for (int q = 0; q < 3; q++) { a.dd1[q] = 2 * n.dx[q]; a.dd2[q] = 1 / (n.dx[q] * n.dx[q]); a.dg[q] = 0.5 / n.dx[q]; }   // operators/common.py:60-110, :150-190
    a.ndim = n.ndim; a.lx = 1;
    for (int m = 0; m < 3; m++) a.ex[m] = extra3_host ? extra3_host[m] : nullptr;
    for (int q = 0; q < nparams; q++) a.par[q] = params_host[q];

    bool aligned = ((uintptr_t)in_full % 16 == 0) && ((uintptr_t)out_full % 16 == 0);
    for (int m = 0; m < 3; m++) aligned = aligned && ((uintptr_t)a.ex[m] % 16 == 0);
    if (stage) {
        if (!stage->y || !stage->out2 || stage->out2 == in_full || out_full == in_full) PDEHIP_FAIL(E_VALUE, "jit_apply_stage: NULL or aliased array pointer");
        a.st_kind = stage->kind; a.st_y = stage->y; a.st_out = stage->out2; a.st_err = stage->err;
        int nk = 0;
        for (int m = 0; m < 5 && stage->k[m]; m++, nk++) { a.st_k[m] = stage->k[m]; a.st_c[m] = stage->c[m]; }
        if ((stage->kind == 1 && nk != 3) || (stage->kind == 2 && (nk != 4 || !stage->err)) || (stage->kind == 4 && (nk != 2 || !stage->err)) || stage->kind < 0 ||
            stage->kind == 3 || stage->kind > 4)
            PDEHIP_FAIL(E_VALUE, "jit_apply_stage: malformed stage (kind %d with %d earlier slopes)", stage->kind, nk);
        a.st_c[5] = stage->c_new;
        uintptr_t bits = (uintptr_t)a.st_y | (uintptr_t)a.st_out;
        for (int m = 0; m < 5; m++) bits |= (uintptr_t)a.st_k[m];
        aligned = aligned && bits % 16 == 0;
    }
    const bool fast = n.ndim >= 2 && aligned;   // any row length: rows ending inside a vector are stored element-wise (pdehip_march.inc)
    if (stage && !fast) return 0;   // only the vectorised kernel carries the stage epilogue

    // boundary conditions: on the fly where possible (fast kernel), ghost kernel otherwise
    InputBCs fg;
    memset(&fg, 0, sizeof(fg));
    int n_fused = 0;
    if (in_faces) {
        pdehip_bc_face_t rest[2 * PDEHIP_MAX_DIM];
        int n_rest = 0;
        for (int ar = 0; ar < PDEHIP_MAX_DIM; ar++)
            for (int side = 0; side < 2; side++) {
                pdehip_bc_face_t &r = rest[2 * ar + side];
                if (ar >= n.ndim) { memset(&r, 0, sizeof(r)); continue; }
                r = in_faces[2 * ar + side];
                if (r.kind == PDEHIP_BC_SKIP) continue;
                const int ax = 3 - n.ndim + ar;
                if ((fast || n.ndim == 1) && r.kind == PDEHIP_BC_ORDER1 && r.flags == 0 && r.index1 >= 0 && r.index1 < n.n[ax]) {   // 1-D: the generic kernel does it too
                    fg.on[ax][side] = 1; fg.idx[ax][side] = r.index1; fg.c[ax][side] = r.const_v; fg.f[ax][side] = r.factor1;
                    r.kind = PDEHIP_BC_SKIP;
                    n_fused++;
                } else {
                    n_rest++;
                }
            }
        if (n_rest) PDEHIP_TRY(launch_ghosts(n, 1, rest, in_full, as_stream(stream)));
    }
    for (int ax = 0; ax < 3; ax++)
        for (int side = 0; side < 2; side++) {
            a.ibc[ax][side].on = fg.on[ax][side]; a.ibc[ax][side].idx = fg.idx[ax][side];
            a.ibc[ax][side].c = fg.c[ax][side]; a.ibc[ax][side].f = fg.f[ax][side];
        }
    a.any_ibc = n_fused > 0;

    Variant v;
    unsigned blocks, threads;
    const char *tname = f64 ? "double" : "float";
    if (fast) {
        const long chunks = (n.n[2] + 64L * vec - 1) / (64L * vec);
        int cz = chunks >= 4 ? 4 : (chunks >= 2 ? 2 : 1);
        int ry = 2;   // measured: 2-row tiles win in 2-D as well (4096^2 Allen-Cahn 80 % vs 40 % of the HBM peak with 8 rows)
        static int ry_env = -1;   // PDEHIP_JIT_RY: rows per wave tile (tuning aid; any value works, kernels are built on demand)
        if (ry_env < 0) { const char *e = getenv("PDEHIP_JIT_RY"); ry_env = e ? atoi(e) : 0; }
        // stage sweeps: 1-row tiles.  The 2-row stage kernel (28 KB of code) runs at its one-wave-per-SIMD speed when it
        // is loaded through hipModuleLoadData (4.9 vs 3.7 ms per RK4 step at 512^3; the same code built into the library
        // is not affected, nor is this one once librocprofiler-sdk.so is in the process); with half the code the
        // effect is gone (4.0 ms): profiles/r01_time_expr_rk.md
        if (stage) ry = 1;
        if (ry_env > 0) ry = ry_env;
        static int cz_env = -1;   // PDEHIP_JIT_CZ: chunks per wave tile (tuning aid)
        if (cz_env < 0) { const char *e = getenv("PDEHIP_JIT_CZ"); cz_env = e ? atoi(e) : 0; }
        if (cz_env > 0 && cz_env <= cz) cz = cz_env;
        auto n_tiles = [&](int ry_, int cz_) {
            const long per_plane = ((n.n[1] + ry_ - 1) / ry_) * ((n.n[2] + 64L * vec * cz_ - 1) / (64L * vec * cz_));
            return n.ndim == 3 ? per_plane * n.n[0] : per_plane;
        };
        while (cz > 1 && n_tiles(ry, cz) < 512) cz /= 2;
        const bool hasx = n.ndim == 3, ibc = n_fused > 0;
        static int wy_env = -1;   // PDEHIP_JIT_WY: waves per workgroup of the stage sweeps (tuning aid)
        if (wy_env < 0) { const char *e = getenv("PDEHIP_JIT_WY"); wy_env = e ? atoi(e) : 0; }
        const int wy = (stage && wy_env > 0) ? wy_env : 1;
        const bool tails = (n.n[2] % vec != 0) || (chunks % cz != 0);
        const std::string key = std::string(tname) + "," + std::to_string(ry) + "," + std::to_string(cz) + "," + (hasx ? "x" : "-") + (ibc ? "b" : "-") +
                                (stage ? "s" : "-") + std::to_string(wy) + (tails ? "t" : "-");
        auto it = j->cache.find(mode_key(key));
        if (it != j->cache.end()) v = it->second;
        else PDEHIP_TRY(compile_variant(j, key, false, tname, vec, ry, cz, hasx, ibc, &v, 0, stage != nullptr, wy, tails));
        a.ntz = (a.n2 + 64L * vec * cz - 1) / (64L * vec * cz);
        a.nty = (a.n1 + wy * ry - 1) / (wy * ry);
        const long tiles = a.ntz * a.nty;
        long lx = a.n0;
        if (hasx) {
            // stages with few pointwise streams: two waves per SIMD (see launch_laplace_t, profiles/r01_time_rk.md)
            static long want_env = -1;   // PDEHIP_JIT_BLOCKS: wave tiles per sweep (tuning aid)
            if (want_env < 0) { const char *e = getenv("PDEHIP_JIT_BLOCKS"); want_env = e ? atol(e) : 0; }
            const long want = want_env > 0 ? want_env : ((stage && (stage->kind == 1 || !stage->k[1])) ? 2048 : 1024);
            long nxc = (want / wy + tiles - 1) / tiles;   // `want` counts waves
            if (nxc < 1) nxc = 1;
            if (nxc > a.n0) nxc = a.n0;
            lx = (a.n0 + nxc - 1) / nxc;
        }
        a.lx = (int)lx;
        a.nxc = (a.n0 + lx - 1) / lx;
        a.nblocks = a.nxc * tiles;
        blocks = (unsigned)a.nblocks;
        threads = 64 * wy;
    } else {
        const std::string key = std::string("generic,") + tname;
        auto it = j->cache.find(mode_key(key));
        if (it != j->cache.end()) v = it->second;
        else PDEHIP_TRY(compile_variant(j, key, true, tname, 1, 1, 1, false, false, &v));
        long b = (n.n[0] * n.n[1] * n.n[2] + 255) / 256;
        blocks = (unsigned)(b > 8192 ? 8192 : (b < 1 ? 1 : b));
        threads = 256;
    }
    void *args[] = {&a};
    PDEHIP_HIP(hipModuleLaunchKernel(v.fn, blocks, 1, 1, threads, 1, 1, 0, as_stream(stream), args, nullptr));
    if (done) *done = 1;
    return 0;
}
}  // namespace

extern "C" {

int pdehip_jit_apply(void *handle, const pdehip_grid_t *g, void *in_full, const void *const *extra3_host, void *out_full,
                     const double *params_host, int nparams, const pdehip_bc_face_t *in_faces, void *stream)
{
    return jit_apply_impl(handle, g, in_full, extra3_host, out_full, params_host, nparams, in_faces, stream, nullptr, nullptr);
}

// pdehip_jit_apply whose result is the slope k of a Runge-Kutta stage (the epilogue must compute dt * F), followed in the
// SAME sweep by the pointwise combination of the scheme (all arrays FULL):
//   kind 0: k_out = k,  out2 = y + sum_m coef[m] * k_prev[m] + c_new * k                       (input of the next stage)
//   kind 1: out2 = y + (k_prev[0] + 2 k_prev[1] + 2 k_prev[2] + k) / 6     (RK4 update; k_out unused, out2 may be y)
//   kind 2: out2 = y + c1 k1 + c3 k3 + c4 k4 + c5 k5, *err_dev = max |error estimate| with k6 = k, k_prev = {k1, k3, k4, k5}
//           (end of an RKF45 attempt; *err_dev is zeroed first)
//   kind 4: out2 = k_prev[1] + k, *err_dev = max |(y + coef[0] * k_prev[0]) - out2|: end of an adaptive Euler attempt with k_prev = {carried
//           rate, half step}, coef[0] = dt, k = dt/2 * F(half step)  (pde/backends/numba/_solvers.py:381-394; *err_dev is zeroed first)
// The same expressions in the same order as pdehip_lincomb / pdehip_rk4_combine / pdehip_rkf45_combine.
// *done = 0 and nothing launched when only the generic kernel covers the grid (1-D, odd rows).
int pdehip_jit_apply_stage(void *handle, const pdehip_grid_t *g, void *in_full, const void *const *extra3_host, void *k_out_full,
                           const double *params_host, int nparams, const pdehip_bc_face_t *in_faces, int kind,
                           const void *y_full, int nk, const void *const *k_prev_host, const double *coef_host, double c_new,
                           void *out2_full, double *err_dev, int *done, void *stream)
{
    if (!done) PDEHIP_FAIL(E_VALUE, "jit_apply_stage: NULL pointer");
    if (nk < 0 || nk > 5 || (nk > 0 && !k_prev_host) || ((kind == 0 || kind == 4) && nk > 0 && !coef_host)) PDEHIP_FAIL(E_VALUE, "jit_apply_stage: bad slope table");
    StageFuse sf;
    memset(&sf, 0, sizeof(sf));
    sf.kind = kind; sf.y = y_full; sf.out2 = out2_full; sf.err = err_dev; sf.c_new = c_new;
    for (int m = 0; m < nk; m++) {
        if (!k_prev_host[m]) PDEHIP_FAIL(E_VALUE, "jit_apply_stage: NULL slope array");
        sf.k[m] = k_prev_host[m];
        sf.c[m] = (kind == 0 || (kind == 4 && m == 0)) ? coef_host[m] : 0.0;
    }
    if (kind == 2 || kind == 4) {
        if (!err_dev) PDEHIP_FAIL(E_VALUE, "jit_apply_stage: NULL error cell");
        PDEHIP_HIP(hipMemsetAsync(err_dev, 0, sizeof(double), as_stream(stream)));
    }
    return jit_apply_impl(handle, g, in_full, extra3_host, k_out_full, params_host, nparams, in_faces, stream, &sf, done);
}

// TWO applications of the epilogue in one sweep: out = f(f(in)) with f(u) = pde_epilogue(u, laplace(u), gradient_squared(u);
// params) and the BCs `faces` applied to u before each application — two explicit Euler steps of a one-pass expression PDE
// whose epilogue is the Euler update `u + dt * F`.  The intermediate level lives in registers (pdehip_march2.inc).
// *done = 0 and nothing launched when grid / faces are not covered (same rules as pdehip_diffusion_euler2) — the caller
// then applies pdehip_jit_apply twice.
int pdehip_jit_euler2(void *handle, const pdehip_grid_t *g, const void *in_full, void *out_full, const double *params_host,
                      int nparams, const pdehip_bc_face_t *faces, int *done, void *stream)
{
    if (!handle || !in_full || !out_full || !faces || !done) PDEHIP_FAIL(E_VALUE, "jit_euler2: NULL pointer");
    if (nparams < 0 || nparams > 12) PDEHIP_FAIL(E_VALUE, "jit_euler2: at most 12 scalar parameters");
    *done = 0;
    Jit *j = static_cast<Jit *>(handle);
    if (uses_axis_derivatives(j)) return 0;   // the two-level kernel evaluates laplace / gradient_squared only
    NGrid n;
    PDEHIP_TRY(norm_grid(g, &n));
    if (n.ndim < 2) return 0;
    InputBCs fg;
    memset(&fg, 0, sizeof(fg));
    for (int a = 0; a < n.ndim; a++)
        for (int side = 0; side < 2; side++) {
            const int ax = 3 - n.ndim + a;
            const pdehip_bc_face_t &r = faces[2 * a + side];
            if (r.kind != PDEHIP_BC_ORDER1 || r.flags != 0 || r.index1 < 0 || r.index1 >= n.n[ax]) return 0;
            fg.on[ax][side] = 1; fg.idx[ax][side] = r.index1; fg.c[ax][side] = r.const_v; fg.f[ax][side] = r.factor1;
        }
    Euler2Plan plan;
    bool ok = false;
    PDEHIP_TRY(launch_euler2(n, in_full, out_full, 0.0, 0.0, fg, false, as_stream(stream), &ok, false, 0, E2_CUSTOM, nullptr, 0.0, &plan));
    if (!ok) return 0;
    for (int q = 0; q < nparams; q++) plan.a.par[q] = params_host[q];
    const bool f64 = n.dtype == PDEHIP_F64;
    const char *tname = f64 ? "double" : "float";
    const std::string key = std::string("two,") + tname + "," + std::to_string(plan.ry) + (plan.has_y ? ",y" : ",-");
    Variant v;
    auto it = j->cache.find(mode_key(key));
    if (it != j->cache.end()) v = it->second;
    else PDEHIP_TRY(compile_variant(j, key, false, tname, f64 ? 2 : 4, plan.ry, 1, plan.has_y, true, &v, E2_CUSTOM));
    void *kargs[] = {&plan.a};
    PDEHIP_HIP(hipModuleLaunchKernel(v.fn, plan.grid, 1, 1, plan.block, 1, 1, 0, as_stream(stream), kargs, nullptr));
    *done = 1;
    return 0;
}

// ONE sweep for a two-pass expression  tmp = f1(u, lap u, |grad u|^2),  out = f2(tmp, lap tmp, |grad tmp|^2, e0 = u):
// tmp lives in registers only (two-level kernel, pdehip_march2.inc).  `faces_u` are the BCs of u, `faces_tmp` those of
// tmp (both periodic on the same axes).  *done = 0 when grid / faces are not covered: the caller runs the two passes.
int pdehip_jit_fused2(void *handle, const pdehip_grid_t *g, const void *in_full, void *out_full, const double *params_host,
                      int nparams, const pdehip_bc_face_t *faces_u, const pdehip_bc_face_t *faces_tmp, int *done, void *stream)
{
    if (!handle || !in_full || !out_full || !faces_u || !faces_tmp || !done) PDEHIP_FAIL(E_VALUE, "jit_fused2: NULL pointer");
    if (nparams < 0 || nparams > 12) PDEHIP_FAIL(E_VALUE, "jit_fused2: at most 12 scalar parameters");
    *done = 0;
    Jit *j = static_cast<Jit *>(handle);
    if (j->body2.empty()) PDEHIP_FAIL(E_VALUE, "jit_fused2: handle was not created with pdehip_jit_create2");
    if (uses_axis_derivatives(j)) return 0;
    NGrid n;
    PDEHIP_TRY(norm_grid(g, &n));
    if (n.ndim < 2) return 0;
    InputBCs fg[2];
    const pdehip_bc_face_t *fsrc[2] = {faces_u, faces_tmp};
    for (int l = 0; l < 2; l++) {
        memset(&fg[l], 0, sizeof(fg[l]));
        for (int a = 0; a < n.ndim; a++)
            for (int side = 0; side < 2; side++) {
                const int ax = 3 - n.ndim + a;
                const pdehip_bc_face_t &r = fsrc[l][2 * a + side];
                if (r.kind != PDEHIP_BC_ORDER1 || r.flags != 0 || r.index1 < 0 || r.index1 >= n.n[ax]) return 0;
                fg[l].on[ax][side] = 1; fg[l].idx[ax][side] = r.index1; fg[l].c[ax][side] = r.const_v; fg[l].f[ax][side] = r.factor1;
            }
    }
    Euler2Plan plan;
    bool ok = false;
    PDEHIP_TRY(launch_euler2(n, in_full, out_full, 0.0, 0.0, fg[0], false, as_stream(stream), &ok, false, 0, E2_CUSTOM2, &fg[1], 0.0, &plan));
    if (!ok) return 0;
    for (int q = 0; q < nparams; q++) plan.a.par[q] = params_host[q];
    const bool f64 = n.dtype == PDEHIP_F64;
    const char *tname = f64 ? "double" : "float";
    const std::string key = std::string("fused2,") + tname + "," + std::to_string(plan.ry) + (plan.has_y ? ",y" : ",-");
    Variant v;
    auto it = j->cache.find(mode_key(key));
    if (it != j->cache.end()) v = it->second;
    else PDEHIP_TRY(compile_variant(j, key, false, tname, f64 ? 2 : 4, plan.ry, 1, plan.has_y, true, &v, E2_CUSTOM2));
    void *kargs[] = {&plan.a};
    PDEHIP_HIP(hipModuleLaunchKernel(v.fn, plan.grid, 1, 1, plan.block, 1, 1, 0, as_stream(stream), kargs, nullptr));
    *done = 1;
    return 0;
}


// ---- boundary conditions given as expressions, evaluated on the device (include/pdehip.h: pdehip_bcprog_*) ------------------------
namespace {
struct BcFaceDev {   // device copy of a face descriptor + the start of its cells in the launch
    double *A, *B;
    long m1, m2;
    double origin[3], step[3];
    int index[3];
    long first[3];     // global index of the first face cell per coordinate (slabs / blocks of a decomposed grid)
    int reads;         // the face reads the field: element offset of face cell (i1, i2) = soff + i1 * sp1 + i2 * sp2
    double dx;
    long start;
    long soff, sp1, sp2;
};
struct BcProg {
    hipModule_t module = nullptr;
    hipFunction_t fn = nullptr;
    BcFaceDev *faces_dev = nullptr;
    // the second coefficient set (bcprog_run_pair: the faces of the second level of a two-step sweep), allocated on first use
    std::vector<BcFaceDev> host;
    BcFaceDev *faces_dev2 = nullptr;
    double *second = nullptr;   // 2 * total doubles: per face A then B
    int nfaces = 0;
    long total = 0;
    int reads = 0;     // some face reads the field
    int esz = 8;       // bytes per element of the field
};
const char *kBcKernel = R"SRC(
struct BcFaceDev { double *A, *B; long m1, m2; double origin[3], step[3]; int index[3]; long first[3]; int reads; double dx; long start; long soff, sp1, sp2; };
extern "C" __global__ void __launch_bounds__(256) bc_refresh(const BcFaceDev *faces, int nfaces, long total, double t, const void *state, int esz,
                                                              const BcFaceDev *faces2, double t2)
{
    for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256L) {
        int f = 0;
        for (int q = 1; q < nfaces; q++)
            if (i >= faces[q].start) f = q;
        const BcFaceDev &F = faces[f];
        const long loc = i - F.start, i2 = loc % F.m2, i1 = loc / F.m2;
        double c[3];
        for (int k = 0; k < 3; k++) {
            const int w = F.index[k];
            // the cell centres of the reference, operation by operation: (i + 0.5) * dx + x_min  (pde/grids/base.py:112-113)
            c[k] = w == 0 ? F.origin[k] : ((double)(F.first[k] + (w == 1 ? i1 : i2)) + 0.5) * F.step[k] + F.origin[k];
        }
        double value = 0;
        if (F.reads) {
            const long o = F.soff + i1 * F.sp1 + i2 * F.sp2;
            value = esz == 8 ? ((const double *)state)[o] : (double)((const float *)state)[o];
        }
        double a = 0, b = 0;
        bc_face(f, value, F.dx, c[0], c[1], c[2], t, &a, &b);
        F.A[loc] = a;
        F.B[loc] = b;
        if (faces2) {   // the second coefficient set: the same faces at time t2 (the second level of a two-step sweep)
            bc_face(f, value, F.dx, c[0], c[1], c[2], t2, &a, &b);
            faces2[f].A[loc] = a;
            faces2[f].B[loc] = b;
        }
    }
}
)SRC";
}  // namespace

extern "C" {

int pdehip_bcprog_create(const char *source, int nfaces, const pdehip_bcprog_face_t *faces, const pdehip_grid_t *grid, void **handle)
{
    if (!source || !faces || !handle || nfaces < 1 || nfaces > 64) PDEHIP_FAIL(E_VALUE, "bcprog_create: NULL pointer or bad face count");
    NGrid ng;
    bool reads = false;
    for (int f = 0; f < nfaces; f++) reads = reads || faces[f].reads_value != 0;
    if (reads) {
        if (!grid) PDEHIP_FAIL(E_VALUE, "bcprog_create: a face reads the field but no grid is given");
        PDEHIP_TRY(norm_grid(grid, &ng));
    }
    // One compiled + loaded module per SOURCE STRING for the life of the process: a program is built per face table, per
    // right-hand side and per loop, mostly from identical sources; modules are never unloaded (see pdehip_bcprog_destroy), so without
    // the cache every solver leaked a code object and paid the hiprtc compile again.  The per-handle state is the face table only.
    static std::mutex bc_mu;
    static std::map<std::string, std::pair<hipModule_t, hipFunction_t>> bc_modules;
    const std::string key(source);
    hipModule_t module = nullptr;
    hipFunction_t fn = nullptr;
    {
        std::lock_guard<std::mutex> lock(bc_mu);
        auto it = bc_modules.find(key);
        if (it != bc_modules.end()) { module = it->second.first; fn = it->second.second; }
    }
    if (!module) {
        PDEHIP_TRY(load_rtc());
        // (no host headers in a hiprtc source: constants arrive as literals, pde_hip/expr.py `c_printer`)
        std::string src = "#define PDEHIP_BC_FN __device__ __forceinline__\n";
        src += source;
        src += kBcKernel;
        hiprtcProgram prog = nullptr;
        if (g_rtc.CreateProgram(&prog, src.c_str(), "bc_program.hip", 0, nullptr, nullptr) != 0) PDEHIP_FAIL(E_RUNTIME, "hiprtcCreateProgram failed");
        const char *opts[] = {"--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off"};
        if (g_rtc.CompileProgram(prog, 4, opts) != 0) {
            size_t n = 0;
            g_rtc.GetProgramLogSize(prog, &n);
            std::string log(n + 1, '\0');
            if (n) g_rtc.GetProgramLog(prog, &log[0]);
            g_rtc.DestroyProgram(&prog);
            PDEHIP_FAIL(E_VALUE, "boundary-condition program does not compile: %.400s", log.c_str());
        }
        size_t n = 0;
        g_rtc.GetCodeSize(prog, &n);
        std::vector<char> code(n);
        g_rtc.GetCode(prog, code.data());
        g_rtc.DestroyProgram(&prog);
        hipError_t e = hipModuleLoadData(&module, code.data());
        if (e == hipSuccess) e = hipModuleGetFunction(&fn, module, "bc_refresh");
        if (e != hipSuccess) {
            if (module) (void)hipModuleUnload(module);
            PDEHIP_FAIL(E_RUNTIME, "bcprog_create: %s", hipGetErrorString(e));
        }
        std::lock_guard<std::mutex> lock(bc_mu);
        auto ins = bc_modules.emplace(key, std::make_pair(module, fn));
        if (!ins.second) { module = ins.first->second.first; fn = ins.first->second.second; }   // (another thread was first: its module serves)
    }
    BcProg *b = new BcProg();
    b->module = module;
    b->fn = fn;
    std::vector<BcFaceDev> host((size_t)nfaces);
    long start = 0;
    for (int f = 0; f < nfaces; f++) {
        const pdehip_bcprog_face_t &s = faces[f];
        if (!s.const_arr || !s.factor_arr || s.m1 < 1 || s.m2 < 1) { delete b; PDEHIP_FAIL(E_VALUE, "bcprog_create: face %d has no arrays / cells", f); }
        BcFaceDev &d = host[f];
        d.A = s.const_arr; d.B = s.factor_arr; d.m1 = s.m1; d.m2 = s.m2; d.dx = s.dx; d.start = start;
        for (int k = 0; k < 3; k++) { d.origin[k] = s.origin[k]; d.step[k] = s.step[k]; d.index[k] = s.index[k]; d.first[k] = s.first[k]; }
        d.reads = s.reads_value != 0; d.soff = d.sp1 = d.sp2 = 0;
        if (d.reads) {
            // the cell (value_index along the face's axis, i1, i2 along the others in grid order) of component `component`
            const int nd = ng.ndim, ax = s.axis;
            if (ax < 0 || ax >= nd || s.value_index < 0 || s.value_index >= ng.n[3 - nd + ax] || s.component < 0) { delete b; PDEHIP_FAIL(E_VALUE, "bcprog_create: face %d: bad axis / value cell / component", f); }
            int others[2], no = 0;
            for (int a = 0; a < nd; a++) if (a != ax) others[no++] = a;
            if ((no >= 1 ? ng.n[3 - nd + others[0]] : 1) != s.m1 || (no >= 2 ? ng.n[3 - nd + others[1]] : 1) != s.m2) { delete b; PDEHIP_FAIL(E_VALUE, "bcprog_create: face %d: extents do not match the grid", f); }
            d.soff = ng.off + (long)s.component * ng.pc + s.value_index * ng.p[3 - nd + ax];
            d.sp1 = no >= 1 ? ng.p[3 - nd + others[0]] : 0;
            d.sp2 = no >= 2 ? ng.p[3 - nd + others[1]] : 0;
        }
        start += s.m1 * s.m2;
    }
    b->host = host;
    b->nfaces = nfaces;
    b->reads = reads ? 1 : 0;
    b->esz = reads ? (int)elem_size(ng.dtype) : 8;
    b->total = start;
    hipError_t e = hipMalloc(&b->faces_dev, sizeof(BcFaceDev) * host.size());
    if (e == hipSuccess) e = hipMemcpy(b->faces_dev, host.data(), sizeof(BcFaceDev) * host.size(), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipStreamSynchronize(nullptr);   // the program runs on the callers' non-blocking streams (see pdehip_malloc)
    if (e != hipSuccess) {
        if (b->faces_dev) (void)hipFree(b->faces_dev);
        delete b;
        PDEHIP_FAIL(E_RUNTIME, "bcprog_create: %s", hipGetErrorString(e));
    }
    *handle = b;
    return 0;
}

int pdehip_bcprog_run(void *handle, double t, const void *state_full, void *stream)
{
    BcProg *b = static_cast<BcProg *>(handle);
    if (!b) PDEHIP_FAIL(E_VALUE, "bcprog_run: NULL handle");
    if (b->reads && !state_full) PDEHIP_FAIL(E_VALUE, "bcprog_run: the conditions read the field, but no field is given");
    const BcFaceDev *faces = b->faces_dev;
    int nfaces = b->nfaces, esz = b->esz;
    long total = b->total;
    const BcFaceDev *faces2 = nullptr;
    double t2 = 0;
    void *kargs[] = {&faces, &nfaces, &total, &t, &state_full, &esz, &faces2, &t2};
    const unsigned blocks = (unsigned)((total + 255) / 256 < 1024 ? (total + 255) / 256 : 1024);
    PDEHIP_HIP(hipModuleLaunchKernel(b->fn, blocks, 1, 1, 256, 1, 1, 0, as_stream(stream), kargs, nullptr));
    return 0;
}

int pdehip_bcprog_destroy(void *handle)
{
    BcProg *b = static_cast<BcProg *>(handle);
    if (!b) return 0;
    // (the module belongs to the per-source cache of pdehip_bcprog_create and stays loaded: unloading code objects in the middle of a
    // run was the trigger of the lazy-load fault noted in pdehip_kernels.hip)
    if (b->faces_dev) (void)hipFree(b->faces_dev);
    if (b->faces_dev2) (void)hipFree(b->faces_dev2);
    if (b->second) (void)hipFree(b->second);
    delete b;
    return 0;
}

}  // extern "C"

// ---- two coefficient sets per launch pair: the faces of BOTH levels of a two-step sweep (pdehip_shell.hip) ------------------------
extern "C++" {   // (this part of the file sits inside an extern "C" region)
namespace pdehip {
bool bcprog_reads(void *handle) { return handle && static_cast<BcProg *>(handle)->reads != 0; }

static int bcprog_second_alloc(BcProg *b)
{
    if (b->second) return 0;
    PDEHIP_HIP(hipMalloc(&b->second, sizeof(double) * 2 * (size_t)b->total));
    std::vector<BcFaceDev> h2 = b->host;
    for (auto &d : h2) {
        d.A = b->second + 2 * d.start;
        d.B = d.A + d.m1 * d.m2;
    }
    hipError_t e = hipMalloc(&b->faces_dev2, sizeof(BcFaceDev) * h2.size());
    if (e == hipSuccess) e = hipMemcpy(b->faces_dev2, h2.data(), sizeof(BcFaceDev) * h2.size(), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipStreamSynchronize(nullptr);
    if (e != hipSuccess) PDEHIP_FAIL(E_RUNTIME, "bcprog (second set): %s", hipGetErrorString(e));
    return 0;
}

// the coefficient arrays of the second set that belong to the face whose first-set constants are `const_arr`
bool bcprog_second_set(void *handle, const double *const_arr, const double **c2, const double **f2)
{
    BcProg *b = static_cast<BcProg *>(handle);
    if (!b || bcprog_second_alloc(b) != 0) return false;
    for (const auto &d : b->host)
        if (d.A == const_arr) {
            *c2 = b->second + 2 * d.start;
            *f2 = *c2 + d.m1 * d.m2;
            return true;
        }
    return false;
}

// first set for time t0 (the arrays of the face tables), second set for t1; programs that read the field have no second set
int bcprog_run_pair(void *handle, double t0, double t1, void *stream)
{
    BcProg *b = static_cast<BcProg *>(handle);
    if (!b || b->reads) PDEHIP_FAIL(E_RUNTIME, "internal: two coefficient sets of a program that reads the field");
    PDEHIP_TRY(bcprog_second_alloc(b));
    const BcFaceDev *faces = b->faces_dev, *faces2 = b->faces_dev2;   // both sets in ONE launch
    int nfaces = b->nfaces, esz = b->esz;
    long total = b->total;
    const void *state = nullptr;
    void *kargs[] = {&faces, &nfaces, &total, &t0, &state, &esz, &faces2, &t1};
    const unsigned blocks = (unsigned)((total + 255) / 256 < 1024 ? (total + 255) / 256 : 1024);
    PDEHIP_HIP(hipModuleLaunchKernel(b->fn, blocks, 1, 1, 256, 1, 1, 0, as_stream(stream), kargs, nullptr));
    return 0;
}
}  // namespace pdehip
}  // extern "C++"

// ---- fixed-step Euler loop over the passes of an expression PDE (include/pdehip.h) ---------------------------------------------
namespace {
struct LoopGraph {
    std::vector<char> key;
    hipGraphExec_t exec = nullptr;
};
constexpr int kLoopGraphs = 8;
LoopGraph g_loop_graphs[kLoopGraphs];
unsigned g_loop_next = 0;
hipStream_t g_loop_cap = nullptr;
hipEvent_t g_loop_ev = nullptr;

// all passes of one evaluation: `src` / `extras` index -1 - k = component k of `cur`, `out` index -1 - k = component k of `nxt`
// decomposed grids: the ghost layers of a pass's operand travel before the pass (pdehip_exchange_t, include/pdehip.h)
int exchange_operand(const pdehip_grid_t *g, const pdehip_jit_pass_t &p, void *src, void *stream)
{
    const pdehip_exchange_t *x = p.exchange;
    if (!x) return 0;
    int nb6[6];
    for (int i = 0; i < 6; i++) nb6[i] = x->nb6[i];
    return x->blocks ? pdehip_block_exchange(x->comm, g, nb6, src, stream) : pdehip_halo_exchange(x->comm, g, src, x->lower, x->upper, stream);
}
void *exchange_comm(const pdehip_jit_pass_t *passes, int npasses)
{
    for (int q = 0; q < npasses; q++)
        if (passes[q].exchange && passes[q].exchange->comm) return passes[q].exchange->comm;
    return nullptr;
}

int run_passes(const pdehip_grid_t *g, const pdehip_jit_pass_t *passes, int npasses, void *const *fixed, char *cur, char *nxt,
               size_t comp_bytes, const double *params, void *stream, int skip_last = 0)
{
    for (int q = 0; q < npasses - skip_last; q++) {
        const pdehip_jit_pass_t &p = passes[q];
        auto in = [&](int32_t idx) -> void * {
            if (idx == PDEHIP_JIT_NONE) return nullptr;
            return idx >= 0 ? fixed[idx] : (void *)(cur + (size_t)(-1 - idx) * comp_bytes);
        };
        void *out = p.out >= 0 ? fixed[p.out] : (void *)(nxt + (size_t)(-1 - p.out) * comp_bytes);
        const void *ex[3] = {in(p.extras[0]), in(p.extras[1]), in(p.extras[2])};
        PDEHIP_TRY(exchange_operand(g, p, in(p.src), stream));
        PDEHIP_TRY(jit_apply_impl(p.handle, g, in(p.src), ex, out, params, 2, p.faces, stream, nullptr, nullptr));
    }
    return 0;
}

int loop_steps(const pdehip_grid_t *g, const pdehip_jit_pass_t *passes, int npasses, void *const *fixed, char *cur, char *nxt,
               size_t comp_bytes, double dt, double t0, int64_t first, int64_t count, void *stream, void *bc_program = nullptr)
{
    for (int64_t s = 0; s < count; s++) {
        const double params[2] = {dt, t0 + (double)(first + s) * dt};   // _solvers.py:100: t = t_start + i * dt
        if (bc_program) PDEHIP_TRY(pdehip_bcprog_run(bc_program, params[1], cur, stream));   // the faces of THIS step's time and state
        for (int q = 0; q < npasses; q++) {
            const pdehip_jit_pass_t &p = passes[q];
            auto in = [&](int32_t idx) -> void * {
                if (idx == PDEHIP_JIT_NONE) return nullptr;
                return idx >= 0 ? fixed[idx] : (void *)(cur + (size_t)(-1 - idx) * comp_bytes);
            };
            void *out = p.out >= 0 ? fixed[p.out] : (void *)(nxt + (size_t)(-1 - p.out) * comp_bytes);
            const void *ex[3] = {in(p.extras[0]), in(p.extras[1]), in(p.extras[2])};
            PDEHIP_TRY(exchange_operand(g, p, in(p.src), stream));
            PDEHIP_TRY(jit_apply_impl(p.handle, g, in(p.src), ex, out, params, 2, p.faces, stream, nullptr, nullptr));
        }
        char *t = cur; cur = nxt; nxt = t;
    }
    return 0;
}
}  // namespace

int pdehip_jit_euler_run(const pdehip_grid_t *g, const pdehip_jit_pass_t *passes, int npasses, void *const *fixed, int nfixed,
                         void *state_a, void *state_b, int ncomp, double dt, double t0, int uses_time, int64_t nsteps,
                         void *bc_program, void **result, void *stream)
{
    if (bc_program) uses_time = 1;
    if (!g || !passes || !state_a || !state_b || !result || (nfixed > 0 && !fixed)) PDEHIP_FAIL(E_VALUE, "jit_euler_run: NULL pointer");
    if (npasses < 1 || ncomp < 1 || nsteps < 0) PDEHIP_FAIL(E_VALUE, "jit_euler_run: bad pass / component / step count");
    NGrid n;
    PDEHIP_TRY(norm_grid(g, &n));
    const size_t comp_bytes = (size_t)n.pc * elem_size(n.dtype);
    for (int q = 0; q < npasses; q++) {
        const pdehip_jit_pass_t &p = passes[q];
        if (!p.handle) PDEHIP_FAIL(E_VALUE, "jit_euler_run: pass %d has no handle", q);
        const int32_t idx[5] = {p.src, p.extras[0], p.extras[1], p.extras[2], p.out};
        for (int m = 0; m < 5; m++) {
            if (idx[m] == PDEHIP_JIT_NONE && m != 0 && m != 4) continue;
            if (idx[m] == PDEHIP_JIT_NONE || idx[m] >= nfixed || idx[m] < -ncomp)
                PDEHIP_FAIL(E_VALUE, "jit_euler_run: pass %d refers to array %d (fixed: %d, components: %d)", q, (int)idx[m], nfixed, ncomp);
        }
    }
    char *cur = (char *)state_a, *nxt = (char *)state_b;
    int64_t s = 0;
    // 2-D grids of a few MB and a ONE-pass expression of the state alone: K = 8 steps per launch with the time levels in LDS
    // (pdehip_tile2d.inc around the generated update) - such grids are bound by launch / dependency latency, not by bytes
    {
        static int tile_on = -1;
        static long tile_cells = 1L << 21;
        if (tile_on < 0) {
            const char *e = getenv("PDEHIP_TILE2D");
            tile_on = (e && (e[0] == '0' || !strcmp(e, "off"))) ? 0 : 1;
            if ((e = getenv("PDEHIP_TILE2D_CELLS")) != nullptr) tile_cells = atol(e);
        }
        // one field: ONE pass of the state alone.  Two fields (reaction-diffusion systems): one pass per field, each reading its
        // own component through the stencil and at most the OTHER component's centre value (slot e0): both fields advance in
        // lockstep inside the kernel (pde_epilogue for the first, pde_epilogue2 for the second)
        const pdehip_jit_pass_t &p = passes[0];
        auto none = [](int32_t v) { return v == PDEHIP_JIT_NONE; };
        auto own_pass = [&](const pdehip_jit_pass_t &q, int comp, int other) {
            return q.src == -1 - comp && q.out == -1 - comp && (none(q.extras[0]) || q.extras[0] == -1 - other) && none(q.extras[1]) &&
                   none(q.extras[2]) && q.faces && static_cast<Jit *>(q.handle)->body2.empty();
        };
        const bool one = npasses == 1 && ncomp == 1 && own_pass(p, 0, 0) && none(p.extras[0]);
        const bool two = npasses == 2 && ncomp == 2 && own_pass(passes[0], 0, 1) && own_pass(passes[1], 1, 0);
        // two-pass chain of ONE field: tmp = f1(state), state' = f2(tmp; e0 = state) (mode 4, like the Cahn-Hilliard pair)
        const bool chain = npasses == 2 && ncomp == 1 && p.src == -1 && p.out >= 0 && none(p.extras[0]) && none(p.extras[1]) && none(p.extras[2]) &&
                           p.faces && passes[1].src == p.out && passes[1].out == -1 && (none(passes[1].extras[0]) || passes[1].extras[0] == -1) &&
                           none(passes[1].extras[1]) && none(passes[1].extras[2]) && passes[1].faces &&
                           static_cast<Jit *>(p.handle)->body2.empty() && static_cast<Jit *>(passes[1].handle)->body2.empty();
        if (tile_on && !bc_program && (one || two || chain) && n.ndim == 2 && n.n[1] * n.n[2] <= tile_cells && nsteps >= 2) {   // (explicit time: evaluated per level)
            Jit *j = static_cast<Jit *>(p.handle);
            if (two || chain) {
                // the second field reads the first one as e0 and vice versa: when a pass does not use the other field its slot
                // is simply ignored by its epilogue.  One combined handle per pair of epilogues, for the life of the process.
                static std::map<std::string, Jit *> pairs;
                static std::mutex pairs_mutex;
                Jit *j1 = static_cast<Jit *>(passes[1].handle);
                const std::string pk = j->body + (two ? '\x02' : '\x03') + j1->body;
                std::lock_guard<std::mutex> guard(pairs_mutex);
                auto it = pairs.find(pk);
                if (it == pairs.end()) {
                    Jit *jc = new Jit();
                    jc->body = j->body;
                    jc->body2 = j1->body;
                    it = pairs.emplace(pk, jc).first;
                }
                j = it->second;
            }
            InputBCs fc, fm;
            memset(&fc, 0, sizeof(fc));
            memset(&fm, 0, sizeof(fm));
            bool ok = true;
            for (int f = 0; f < ((two || chain) ? 2 : 1) && ok; f++)
                for (int a = 0; a < 2 && ok; a++)
                    for (int side = 0; side < 2; side++) {
                        const pdehip_bc_face_t &r = passes[f].faces[2 * a + side];
                        const int ax = 1 + a;
                        if (r.kind != PDEHIP_BC_ORDER1 || r.flags != 0 || r.index1 < 0 || r.index1 >= n.n[ax]) { ok = false; break; }
                        InputBCs &t = f ? fm : fc;
                        t.on[ax][side] = 1; t.idx[ax][side] = r.index1; t.c[ax][side] = r.const_v; t.f[ax][side] = r.factor1;
                    }
            const int mode = two ? 3 : (chain ? 4 : 2);
            const int kmax = tile2d_max_steps(mode);
            while (ok && s < nsteps) {
                const int k = (int)((nsteps - s) < kmax ? (nsteps - s) : kmax);
                Tile2Args ta;
                unsigned nblocks = 0;
                int tcw = 0;
                bool done = false;
                PDEHIP_TRY(plan_tile2d(n, cur, nxt, mode, 0.0, 0.0, 0.0, fc, (two || chain) ? &fm : nullptr, k, &ta, &nblocks, &tcw, &done));
                if (!done) { ok = false; break; }
                ta.par[0] = dt; ta.par[1] = t0; ta.t0 = t0; ta.step0 = (long)s;
                const char *tname = n.dtype == PDEHIP_F64 ? "double" : "float";
                const std::string key = std::string(two ? "tile2," : (chain ? "tile4," : "tile,")) + tname + "," + std::to_string(tcw);
                Variant v;
                auto it = j->cache.find(mode_key(key));
                if (it != j->cache.end()) v = it->second;
                else PDEHIP_TRY(compile_variant(j, key, false, tname, n.dtype == PDEHIP_F64 ? 2 : 4, mode, tcw, false, true, &v, kTileVariant));
                void *kargs[] = {&ta};
                PDEHIP_HIP(hipModuleLaunchKernel(v.fn, nblocks, 1, 1, 1024, 1, 1, 0, as_stream(stream), kargs, nullptr));
                s += k;
                char *t = cur; cur = nxt; nxt = t;
            }
            if (ok) { *result = cur; return 0; }
            // (not covered: nothing was launched - `ok` can only turn false before the first launch)
        }
    }
    // the first two steps always run as plain launches: they build whatever kernel is not built yet (no hiprtc inside a capture)
    const int64_t head = nsteps < 2 ? nsteps : 2;
    PDEHIP_TRY(loop_steps(g, passes, npasses, fixed, cur, nxt, comp_bytes, dt, t0, 0, head, stream, bc_program));
    s = head;
    if (head % 2) { char *t = cur; cur = nxt; nxt = t; }
    constexpr int64_t kBlock = 16;   // even: the buffers are back in place after a block
    // measured (256^2 Allen-Cahn, one pass per step): replaying the captured block 4.36 us per step, plain launches from this
    // loop 3.68 us - the run-time built kernels carry 1.3 KB of arguments per node; the graph stays an option (PDEHIP_JIT_GRAPH=1)
    static int graphs = -1;
    if (graphs < 0) { const char *e = getenv("PDEHIP_JIT_GRAPH"); graphs = (e && e[0] == '1') ? 1 : 0; }
    if (graphs && !uses_time && nsteps - s >= 4 * kBlock && !exchange_comm(passes, npasses)) {   // (no RCCL groups inside a captured graph)
        // launch-bound regime: replay a captured block (cached per passes / arrays / dt)
        std::vector<char> key;
        auto put = [&](const void *ptr, size_t len) { key.insert(key.end(), (const char *)ptr, (const char *)ptr + len); };
        put(g, sizeof(*g)); put(passes, sizeof(*passes) * npasses);
        if (nfixed) put(fixed, sizeof(void *) * nfixed);
        put(&cur, sizeof(cur)); put(&nxt, sizeof(nxt)); put(&ncomp, sizeof(ncomp)); put(&dt, sizeof(dt));
        LoopGraph *entry = nullptr;
        for (auto &e : g_loop_graphs)
            if (e.exec && e.key == key) { entry = &e; break; }
        if (!g_loop_cap) PDEHIP_HIP(hipStreamCreateWithFlags(&g_loop_cap, hipStreamNonBlocking));
        if (!g_loop_ev) PDEHIP_HIP(hipEventCreateWithFlags(&g_loop_ev, hipEventDisableTiming));
        if (!entry) {
            LoopGraph &e = g_loop_graphs[g_loop_next++ % kLoopGraphs];
            if (e.exec) { (void)hipGraphExecDestroy(e.exec); e.exec = nullptr; }
            hipGraph_t graph = nullptr;
            PDEHIP_HIP(hipStreamBeginCapture(g_loop_cap, hipStreamCaptureModeThreadLocal));
            const int rc = loop_steps(g, passes, npasses, fixed, cur, nxt, comp_bytes, dt, t0, 0, kBlock, (void *)g_loop_cap);
            const hipError_t ce = hipStreamEndCapture(g_loop_cap, &graph);
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
            PDEHIP_HIP(hipEventRecord(g_loop_ev, user));
            PDEHIP_HIP(hipStreamWaitEvent(g_loop_cap, g_loop_ev, 0));
            for (; s + kBlock <= nsteps; s += kBlock) PDEHIP_HIP(hipGraphLaunch(entry->exec, g_loop_cap));
            PDEHIP_HIP(hipEventRecord(g_loop_ev, g_loop_cap));
            PDEHIP_HIP(hipStreamWaitEvent(user, g_loop_ev, 0));
        }
    }
    PDEHIP_TRY(loop_steps(g, passes, npasses, fixed, cur, nxt, comp_bytes, dt, t0, s, nsteps - s, stream, bc_program));
    if ((nsteps - s) % 2) { char *t = cur; cur = nxt; nxt = t; }
    *result = cur;
    return 0;
}

// ---- Runge-Kutta loops over the passes of an expression PDE: the generic loops of pdehip_rk_loops.h with the passes as evaluator ----
namespace {
struct JitEval {
    const pdehip_grid_t *g;
    const pdehip_jit_pass_t *passes;
    int npasses;
    void *const *fixed;
    int ncomp;
    size_t comp_bytes;
    int stage_fuse;     // 1: try the stage epilogue of the last pass, 0: never, -1: refused once (stays off)
    void *bc_program;
    // complex states as planar (re, im) pairs of components (pde_hip/complex_expr.py): the error norm of the adaptive schemes is the
    // modulus, `np.abs(error).max()` of a complex array (pde/solvers/runge_kutta.py:147-148, pde/solvers/euler.py:253) - new state and an
    // explicit error field through the pointwise kernels, then pdehip_max_abs_pairs (the same calls as the host-driven loop)
    void *efield = nullptr;

    int refresh(double t, const void *in, void *st) { return bc_program ? pdehip_bcprog_run(bc_program, t, in, st) : 0; }
    // k_out = dt * F(in; t) and - where the last pass carries it - the combination `sf` in the same sweep (*fused)
    int slope(void *in, void *k_out, double dt, double t, const StageFuse *sf, bool *fused, void *st)
    {
        *fused = false;
        const double params[2] = {dt, t};
        PDEHIP_TRY(refresh(t, in, st));
        const pdehip_jit_pass_t &last = passes[npasses - 1];
        const bool try_stage = sf && stage_fuse > 0 && ncomp == 1 && last.out == -1;
        PDEHIP_TRY(run_passes(g, passes, npasses, fixed, (char *)in, (char *)k_out, comp_bytes, params, st, try_stage ? 1 : 0));
        if (!try_stage) return 0;
        auto arr = [&](int32_t idx) -> void * {
            if (idx == PDEHIP_JIT_NONE) return nullptr;
            return idx >= 0 ? fixed[idx] : (void *)((char *)in + (size_t)(-1 - idx) * comp_bytes);
        };
        const void *ex[3] = {arr(last.extras[0]), arr(last.extras[1]), arr(last.extras[2])};
        int nk = 0;
        while (nk < 5 && sf->k[nk]) nk++;
        int done = 0;
        PDEHIP_TRY(exchange_operand(g, last, arr(last.src), st));
        PDEHIP_TRY(pdehip_jit_apply_stage(last.handle, g, arr(last.src), ex, k_out, params, 2, last.faces, sf->kind, sf->y, nk, sf->k, sf->c,
                                          sf->c_new, sf->out2, sf->err, &done, st));
        if (done) { *fused = true; return 0; }
        stage_fuse = -1;   // only the generic kernel covers this grid: plain pass + pointwise combination from now on
        pdehip_jit_pass_t again = last;
        again.exchange = nullptr;   // (its operand has just been exchanged)
        return run_passes(g, &again, 1, fixed, (char *)in, (char *)k_out, comp_bytes, params, st);
    }
    int lincomb(void *out, const void *y, int n, const double *c, const void *const *k, void *st) { return pdehip_lincomb(g, ncomp, out, y, n, c, k, st); }
    int rk4_combine(void *y, const void *k1, const void *k2, const void *k3, const void *k4, void *st) { return pdehip_rk4_combine(g, ncomp, y, k1, k2, k3, k4, st); }
    int rkf45_combine(const void *y, void *ynew, const void *const *k6, double *err, void *st)
    {
        if (!efield) return pdehip_rkf45_combine(g, ncomp, y, ynew, k6, err, st);
        const double c[4] = {25.0 / 216, 1408.0 / 2565, 2197.0 / 4104, -1.0 / 5};                        // runge_kutta.py:150
        const double r[5] = {1.0 / 360, -128.0 / 4275, -2197.0 / 75240, 1.0 / 50, 2.0 / 55};             // runge_kutta.py:147
        const void *kc[4] = {k6[0], k6[2], k6[3], k6[4]}, *kr[5] = {k6[0], k6[2], k6[3], k6[4], k6[5]};
        PDEHIP_TRY(pdehip_lincomb(g, ncomp, ynew, y, 4, c, kc, st));
        PDEHIP_TRY(pdehip_lincomb(g, ncomp, efield, nullptr, 5, r, kr, st));
        return pdehip_max_abs_pairs(g, ncomp / 2, efield, err, st);
    }
    int euler_adaptive_combine(const void *y, const void *rate, double dt, const void *half, const void *k, void *out, double *err, void *st)
    {
        if (!efield) return pdehip_euler_adaptive_combine(g, ncomp, y, rate, dt, half, k, out, err, st);
        const double one = 1.0, minus = -1.0;
        const void *kk[1] = {k};
        PDEHIP_TRY(pdehip_lincomb(g, ncomp, out, half, 1, &one, kk, st));         // step_small += 0.5 * dt * rate_midpoint
        kk[0] = rate;
        PDEHIP_TRY(pdehip_lincomb(g, ncomp, efield, y, 1, &dt, kk, st));          // step_large
        kk[0] = out;
        PDEHIP_TRY(pdehip_lincomb(g, ncomp, efield, efield, 1, &minus, kk, st));  // step_large - step_small
        return pdehip_max_abs_pairs(g, ncomp / 2, efield, err, st);
    }
    int zero(void *ptr, size_t bytes, void *st) { PDEHIP_HIP(hipMemsetAsync(ptr, 0, bytes, as_stream(st))); return 0; }
    // decomposed grids: MAX over the ranks of the communicator the passes exchange through (NaN wins); nothing on one device
    int reduce_error(double *err_dev, void *st)
    {
        void *comm = exchange_comm(passes, npasses);
        return comm ? pdehip_allreduce_max(comm, err_dev, st) : 0;
    }
    int read_scalar(double *host, const double *dev, void *st)
    {
        PDEHIP_HIP(hipMemcpyAsync(host, dev, sizeof(double), hipMemcpyDeviceToHost, as_stream(st)));
        PDEHIP_HIP(hipStreamSynchronize(as_stream(st)));
        return 0;
    }
    int fail_runtime(const char *fmt, double v) { PDEHIP_FAIL(E_RUNTIME, fmt, v); }
};
}  // namespace

// scheme 0: RK4 with fixed steps (ctl == NULL) / adaptive RKF45 (ctl), 1: the reference's adaptive Euler loop (ctl)
static int jit_loop_run(const char *who, int scheme, const pdehip_grid_t *g, const pdehip_jit_pass_t *passes, int npasses, void *const *fixed, int nfixed,
                        int ncomp, void *y, void *ynew, void *const *work_host, double *err_dev, double dt, double t0, int64_t nsteps,
                        pdehip_adaptive_t *ctl, int stage_fuse, void *bc_program, void **result, void *stream)
{
    if (!g || !passes || !y || !work_host || !result || (nfixed > 0 && !fixed)) PDEHIP_FAIL(E_VALUE, "%s: NULL pointer", who);
    if (npasses < 1 || ncomp < 1 || nsteps < 0) PDEHIP_FAIL(E_VALUE, "%s: bad pass / component / step count", who);
    if (ctl && (!ynew || !err_dev || !(ctl->tolerance > 0) || !(ctl->dt > 0))) PDEHIP_FAIL(E_VALUE, "%s: the adaptive loop needs ynew, err_dev, tolerance > 0 and dt > 0", who);
    NGrid n;
    PDEHIP_TRY(norm_grid(g, &n));
    for (int q = 0; q < npasses; q++) {
        const pdehip_jit_pass_t &p = passes[q];
        if (!p.handle) PDEHIP_FAIL(E_VALUE, "jit_rk_run: pass %d has no handle", q);
        const int32_t idx[5] = {p.src, p.extras[0], p.extras[1], p.extras[2], p.out};
        for (int m = 0; m < 5; m++) {
            if (idx[m] == PDEHIP_JIT_NONE && m != 0 && m != 4) continue;
            if (idx[m] == PDEHIP_JIT_NONE || idx[m] >= nfixed || idx[m] < -ncomp)
                PDEHIP_FAIL(E_VALUE, "jit_rk_run: pass %d refers to array %d (fixed: %d, components: %d)", q, (int)idx[m], nfixed, ncomp);
        }
    }
    JitEval ev{g, passes, npasses, fixed, ncomp, (size_t)n.pc * elem_size(n.dtype), (stage_fuse & 1) ? 1 : 0, bc_program};
    if (stage_fuse & 2) {   // complex pairs: one more work array, the error field (RKF45: work[7], adaptive Euler: work[3])
        if (ncomp % 2 || !ctl) PDEHIP_FAIL(E_VALUE, "%s: complex pairs need an even number of components and the adaptive loop", who);
        ev.efield = work_host[scheme == 1 ? 3 : 7];
        if (!ev.efield) PDEHIP_FAIL(E_VALUE, "%s: complex pairs need the error field as one more work array", who);
    }
    if (scheme == 1) return rk::euler_adaptive_run(ev, y, ynew, work_host, err_dev, ctl, result, stream);
    if (ctl) return rk::rkf45_run(ev, y, ynew, work_host, err_dev, ctl, result, stream);
    for (int64_t s = 0; s < nsteps; s++) PDEHIP_TRY(rk::rk4_step(ev, y, work_host, dt, t0 + (double)s * dt, stream));
    *result = y;
    return 0;
}

int pdehip_jit_rk_run(const pdehip_grid_t *g, const pdehip_jit_pass_t *passes, int npasses, void *const *fixed, int nfixed,
                      int ncomp, void *y, void *ynew, void *const *work_host, double *err_dev, double dt, double t0, int64_t nsteps,
                      pdehip_adaptive_t *ctl, int stage_fuse, void *bc_program, void **result, void *stream)
{
    return jit_loop_run("jit_rk_run", 0, g, passes, npasses, fixed, nfixed, ncomp, y, ynew, work_host, err_dev, dt, t0, nsteps, ctl, stage_fuse,
                        bc_program, result, stream);
}

int pdehip_jit_euler_adaptive_run(const pdehip_grid_t *g, const pdehip_jit_pass_t *passes, int npasses, void *const *fixed, int nfixed,
                                  int ncomp, void *y, void *ynew, void *const *work3_host, double *err_dev, pdehip_adaptive_t *ctl,
                                  int stage_fuse, void *bc_program, void **result, void *stream)
{
    if (!ctl) PDEHIP_FAIL(E_VALUE, "jit_euler_adaptive_run: NULL pointer");
    return jit_loop_run("jit_euler_adaptive_run", 1, g, passes, npasses, fixed, nfixed, ncomp, y, ynew, work3_host, err_dev, 0.0, 0.0, 0, ctl,
                        stage_fuse, bc_program, result, stream);
}

}  // extern "C"
