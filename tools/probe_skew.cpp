// Probe: does time-skewed slab scheduling (two Euler steps per sweep, intermediate in a small ring
// that stays in the 256 MB Infinity Cache) beat two full sweeps?  Timing only (chunk seams ignored).
// build: g++ -O2 -Iinclude tools/probe_skew.cpp -o tools/probe_skew -Lpy-pde_amd/lib -lpdehip -Wl,-rpath,$PWD/py-pde_amd/lib
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "pdehip.h"

#define CK(x) do { if ((x) != 0) { printf("FAIL %s: %s\n", #x, pdehip_last_error()); return 1; } } while (0)

int main(int argc, char **argv) {
    int N = argc > 1 ? atoi(argv[1]) : 512;
    pdehip_grid_t g = {3, PDEHIP_F64, {N, N, N}, {1.0, 1.0, 1.0}};
    int64_t lay[8];
    CK(pdehip_layout(&g, lay));
    const int64_t L = lay[7];
    const size_t bytes = (size_t)(lay[2] + lay[6]) * 8;
    void *A, *B, *T;
    CK(pdehip_malloc(&A, bytes)); CK(pdehip_malloc(&B, bytes)); CK(pdehip_malloc(&T, bytes));
    void *s1, *s2, *e0, *e1;
    CK(pdehip_stream_create(&s1)); CK(pdehip_stream_create(&s2));
    CK(pdehip_event_create(&e0)); CK(pdehip_event_create(&e1));
    const int reps = 20;
    float ms;
    // baseline: two full sweeps
    for (int w = 0; w < 2; ++w) {
        CK(pdehip_event_record(e0, s1));
        for (int r = 0; r < reps; ++r) {
            CK(pdehip_laplace_euler(&g, A, A, B, 1.0, 0.1, s1));
            CK(pdehip_laplace_euler(&g, B, B, A, 1.0, 0.1, s1));
        }
        CK(pdehip_event_record(e1, s1)); CK(pdehip_event_synchronize(e1));
        CK(pdehip_event_elapsed_ms(e0, e1, &ms));
    }
    printf("baseline 2 sweeps: %.4f ms per 2 steps\n", ms / reps);
    for (int P : {4, 8, 16, 32, 64, 128}) {
        if (N % P) continue;
        const int nb = N / P;
        pdehip_grid_t sg = g; sg.shape[0] = P;
        for (int variant = 0; variant < 3; ++variant) {
            // 0: ring of 2 chunks, one stream; 1: ring of 2 chunks, two streams; 2: full-size intermediate, one stream
            std::vector<void *> ev(nb + 1);
            for (auto &e : ev) CK(pdehip_event_create(&e));
            for (int w = 0; w < 2; ++w) {
                CK(pdehip_event_record(e0, s1));
                if (variant == 1) CK(pdehip_stream_wait_event(s2, e0));
                for (int r = 0; r < reps; ++r) {
                    for (int b = 0; b < nb; ++b) {
                        char *in = (char *)A + (size_t)b * P * L * 8;
                        char *out = (char *)B + (size_t)b * P * L * 8;
                        char *t = (char *)T + (size_t)(variant == 2 ? b : b % 2) * P * L * 8;
                        CK(pdehip_laplace_euler(&sg, in, in, t, 1.0, 0.1, s1));
                        if (variant == 1) {
                            CK(pdehip_event_record(ev[b], s1));
                            CK(pdehip_stream_wait_event(s2, ev[b]));
                            CK(pdehip_laplace_euler(&sg, t, t, out, 1.0, 0.1, s2));
                        } else {
                            CK(pdehip_laplace_euler(&sg, t, t, out, 1.0, 0.1, s1));
                        }
                    }
                    if (variant == 1) { CK(pdehip_event_record(ev[nb], s2)); CK(pdehip_stream_wait_event(s1, ev[nb])); }
                }
                CK(pdehip_event_record(e1, s1)); CK(pdehip_event_synchronize(e1));
                CK(pdehip_event_elapsed_ms(e0, e1, &ms));
            }
            printf("P=%3d planes (%4.0f MiB) variant %d: %.4f ms per 2 steps\n", P, (double)P * L * 8 / 1048576, variant, ms / reps);
            for (auto &e : ev) pdehip_event_destroy(e);
        }
    }
    return 0;
}
