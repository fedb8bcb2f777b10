// pdehip_spectral.hip — the SPECTRAL Laplacian of periodic 1-D / 2-D Cartesian grids.
//
// Reference: `_make_laplace_numba_spectral_1d` / `_2d` (pde/backends/numba/operators/cartesian.py:232-330; selected by `spectral=True` or the
// configuration value `use_spectral`, :359-372):   out = ifft(factor * fft(arr)).real,   factor = -(2 pi f)^2 with f = fftfreq(n, dx)  (1-D),
// factor = -4 pi^2 (f0^2 + f1^2) (2-D).  The transform is a library operation (hipFFT / rocFFT, loaded at run time like hiprtc and RCCL:
// the finite-difference path must not depend on it); what is ours: the factor table (same expressions as the reference, in double), the
// multiplication in frequency space with the 1/N of the unnormalised inverse, and the layout conversion between the ghost-padded full
// arrays and the dense arrays the transform wants.  Real-to-complex transforms: the field is real, the result is real by symmetry (the
// reference takes `.real` of the complex inverse).
#include <dlfcn.h>

#include <cstdint>
#include <map>
#include <mutex>
#include <vector>

#include "pdehip_common.h"

using namespace pdehip;

namespace {

typedef struct hipfftHandle_t *hipfftHandle;
enum { FFT_R2C = 0x2a, FFT_C2R = 0x2c, FFT_D2Z = 0x6a, FFT_Z2D = 0x6c };
struct Fft {
    void *handle = nullptr;
    int (*Plan1d)(hipfftHandle *, int, int, int) = nullptr;
    int (*Plan2d)(hipfftHandle *, int, int, int) = nullptr;
    int (*SetStream)(hipfftHandle, hipStream_t) = nullptr;
    int (*ExecD2Z)(hipfftHandle, double *, void *) = nullptr;
    int (*ExecZ2D)(hipfftHandle, void *, double *) = nullptr;
    int (*ExecR2C)(hipfftHandle, float *, void *) = nullptr;
    int (*ExecC2R)(hipfftHandle, void *, float *) = nullptr;
};
Fft g_fft;

int load_fft()
{
    if (g_fft.handle) return 0;
    const char *cands[] = {getenv("PDEHIP_HIPFFT"), "/opt/rocm/lib/libhipfft.so", "libhipfft.so", "libhipfft.so.0"};
    void *h = nullptr;
    for (const char *c : cands)
        if (c && c[0] && (h = dlopen(c, RTLD_NOW | RTLD_LOCAL))) break;
    if (!h) PDEHIP_FAIL(E_RUNTIME, "cannot load hipFFT for the spectral Laplace operator (set PDEHIP_HIPFFT to libhipfft.so): %s", dlerror());
#define PDEHIP_SYM(field, name)                                                    \
    g_fft.field = reinterpret_cast<decltype(g_fft.field)>(dlsym(h, name));          \
    if (!g_fft.field) PDEHIP_FAIL(E_RUNTIME, "hipFFT symbol %s not found", name)
    PDEHIP_SYM(Plan1d, "hipfftPlan1d");
    PDEHIP_SYM(Plan2d, "hipfftPlan2d");
    PDEHIP_SYM(SetStream, "hipfftSetStream");
    PDEHIP_SYM(ExecD2Z, "hipfftExecD2Z");
    PDEHIP_SYM(ExecZ2D, "hipfftExecZ2D");
    PDEHIP_SYM(ExecR2C, "hipfftExecR2C");
    PDEHIP_SYM(ExecC2R, "hipfftExecC2R");
#undef PDEHIP_SYM
    g_fft.handle = h;
    return 0;
}

// one plan pair + factor table + work arrays per (device, stream, shape, spacing, dtype), for the life of the process (plans and
// scratch arrays belong to the device that was current when they were made, and two streams must not share the scratch arrays)
struct Spectral {
    hipfftHandle fwd = nullptr, inv = nullptr;
    double *factor = nullptr;     // n0 * (n1 / 2 + 1) values, already divided by the number of cells
    void *dense = nullptr;        // the valid data without ghost cells
    void *freq = nullptr;         // n0 * (n1 / 2 + 1) complex values
    long nfreq = 0;
};
std::mutex g_mu;
std::map<std::vector<double>, Spectral> g_cache;

template <typename T>
__global__ void __launch_bounds__(256) scale_kernel(T *freq, const double *factor, long n)
{
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256L) {
        const double f = factor[i];
        freq[2 * i] = (T)((double)freq[2 * i] * f);
        freq[2 * i + 1] = (T)((double)freq[2 * i + 1] * f);
    }
}

// numpy.fft.fftfreq(n, d)[k]
inline double fftfreq(long k, long n, double d) { return (double)(k < (n + 1) / 2 ? k : k - n) / ((double)n * d); }

int get_plan(const NGrid &n, void *stream, Spectral **out)
{
    const int nd = n.ndim;
    const long n0 = nd == 2 ? n.n[1] : 1, n1 = n.n[2];     // normalised axes: a 2-D grid uses entries 1 and 2
    int device = 0;
    PDEHIP_HIP(hipGetDevice(&device));
    std::vector<double> key = {(double)nd, (double)n0, (double)n1, n.dx[1], n.dx[2], (double)n.dtype, (double)device, (double)(uintptr_t)stream};
    std::lock_guard<std::mutex> lock(g_mu);
    auto it = g_cache.find(key);
    if (it != g_cache.end()) { *out = &it->second; return 0; }
    PDEHIP_TRY(load_fft());
    Spectral s;
    const bool f64 = n.dtype == PDEHIP_F64;
    const long nh = n1 / 2 + 1;
    s.nfreq = n0 * nh;
    int rc;
    if (nd == 1) {
        rc = g_fft.Plan1d(&s.fwd, (int)n1, f64 ? FFT_D2Z : FFT_R2C, 1);
        if (!rc) rc = g_fft.Plan1d(&s.inv, (int)n1, f64 ? FFT_Z2D : FFT_C2R, 1);
    } else {
        rc = g_fft.Plan2d(&s.fwd, (int)n0, (int)n1, f64 ? FFT_D2Z : FFT_R2C);
        if (!rc) rc = g_fft.Plan2d(&s.inv, (int)n0, (int)n1, f64 ? FFT_Z2D : FFT_C2R);
    }
    if (rc) PDEHIP_FAIL(E_RUNTIME, "hipFFT could not plan a %ld x %ld transform (code %d)", n0, n1, rc);
    // the factor of the reference, expression by expression (cartesian.py:253-254 / :303-304), times 1/N of the unnormalised inverse
    std::vector<double> host((size_t)s.nfreq);
    const double cells = (double)(n0 * n1);
    for (long i = 0; i < n0; i++)
        for (long j = 0; j < nh; j++) {
            double f;
            if (nd == 1) {
                const double ks = 2 * M_PI * fftfreq(j, n1, n.dx[2]);
                f = -(ks * ks);
            } else {
                const double k0 = fftfreq(i, n0, n.dx[1]), k1 = fftfreq(j, n1, n.dx[2]);
                f = -4 * (M_PI * M_PI) * (k0 * k0 + k1 * k1);
            }
            host[(size_t)(i * nh + j)] = f / cells;
        }
    const size_t esz = f64 ? 8 : 4;
    PDEHIP_HIP(hipMalloc(&s.factor, sizeof(double) * host.size()));
    PDEHIP_HIP(hipMemcpy(s.factor, host.data(), sizeof(double) * host.size(), hipMemcpyHostToDevice));
    PDEHIP_HIP(hipMalloc(&s.dense, esz * (size_t)(n0 * n1)));
    PDEHIP_HIP(hipMalloc(&s.freq, 2 * esz * (size_t)s.nfreq));
    rc = g_fft.SetStream(s.fwd, as_stream(stream));
    if (!rc) rc = g_fft.SetStream(s.inv, as_stream(stream));
    if (rc) PDEHIP_FAIL(E_RUNTIME, "hipfftSetStream failed (code %d)", rc);
    *out = &g_cache.emplace(key, s).first->second;
    return 0;
}

}  // namespace

extern "C" {

int pdehip_laplace_spectral(const pdehip_grid_t *g, const void *in_full, void *out, int out_layout, void *stream)
{
    NGrid n;
    PDEHIP_TRY(norm_grid(g, &n));
    if (!in_full || !out) PDEHIP_FAIL(E_VALUE, "laplace_spectral: NULL array pointer");
    if (n.ndim > 2) PDEHIP_FAIL(E_NOTIMPL, "Spectral Laplace operator not implemented for %d dimensions", n.ndim);   // cartesian.py:369-370
    if (out_layout != PDEHIP_OUT_VALID && out_layout != PDEHIP_OUT_FULL) PDEHIP_FAIL(E_VALUE, "laplace_spectral: unknown output layout %d", out_layout);
    Spectral *s = nullptr;
    PDEHIP_TRY(get_plan(n, stream, &s));
    hipStream_t st = as_stream(stream);
    const bool f64 = n.dtype == PDEHIP_F64;
    // full -> dense (plans and scratch arrays belong to this device and stream: calls on one stream are ordered by it)
    PDEHIP_TRY(pdehip_full_to_valid(g, 1, in_full, s->dense, stream));
    int rc = f64 ? g_fft.ExecD2Z(s->fwd, (double *)s->dense, s->freq) : g_fft.ExecR2C(s->fwd, (float *)s->dense, s->freq);
    if (rc) PDEHIP_FAIL(E_RUNTIME, "hipFFT forward transform failed (code %d)", rc);
    const unsigned blocks = (unsigned)((s->nfreq + 255) / 256 < 4096 ? (s->nfreq + 255) / 256 : 4096);
    if (f64) hipLaunchKernelGGL((scale_kernel<double>), dim3(blocks), dim3(256), 0, st, (double *)s->freq, s->factor, s->nfreq);
    else hipLaunchKernelGGL((scale_kernel<float>), dim3(blocks), dim3(256), 0, st, (float *)s->freq, s->factor, s->nfreq);
    PDEHIP_HIP(hipGetLastError());
    rc = f64 ? g_fft.ExecZ2D(s->inv, s->freq, (double *)s->dense) : g_fft.ExecC2R(s->inv, s->freq, (float *)s->dense);
    if (rc) PDEHIP_FAIL(E_RUNTIME, "hipFFT inverse transform failed (code %d)", rc);
    if (out_layout == PDEHIP_OUT_VALID) {
        PDEHIP_HIP(hipMemcpyAsync(out, s->dense, (size_t)elem_size(n.dtype) * (size_t)(n.n[1] * n.n[2]), hipMemcpyDeviceToDevice, st));
        return 0;
    }
    return pdehip_valid_to_full(g, 1, s->dense, out, stream);
}

}  // extern "C"
