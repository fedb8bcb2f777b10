// pdehip_shim_comm.cpp — the slab-parallel entry points of include/pdehip.h on the HOST.  TESTS ONLY.
//
// ******************************************************************************
// TEST INFRASTRUCTURE (see pdehip_shim.c).  Instantiates the SAME loop templates the product compiles for the GPU
// (py-pde_amd/csrc/pdehip_slab_loops.h: halo exchanges, stream choreography, Runge-Kutta stage sequence, adaptive
// accept/reject) with a host `Ops` policy, so that several CPU processes — the ranks of a torch.distributed gloo job —
// execute exactly the product's call sequence:
//   * transport: a directory of message files.  Rank 0 creates the directory, its path is the "unique id".  A message
//     from rank s to rank d with per-pair sequence number q is the file  m_<s>_<d>_<q>  (written under a temporary
//     name, then renamed).  Like RCCL: operations of one group progress together (all sends are posted before the first
//     receive blocks), messages between one pair of ranks match in issue order, a rank may be its own peer.  A receive
//     that waits longer than PDEHIP_SHIM_COMM_TIMEOUT seconds (default 60) FAILS — a mismatched send/recv pattern shows
//     up as an error, not as a hang.
//   * kernels: the CPU oracle; the two-level sweeps (two Euler steps / fused Cahn-Hilliard on sub-slabs with real halo
//     layers) are composed of two oracle passes over the same layers the device kernel covers.
//   * streams / events: everything is synchronous, events are no-ops.
// ******************************************************************************
#include <cerrno>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <thread>
#include <vector>

#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

#include "../../py-pde_amd/csrc/pdehip_slab_loops.h"
#include "../../py-pde_amd/csrc/pdehip_rk_loops.h"
#include "../../py-pde_amd/csrc/pdehip_block_loops.h"
#include "../../py-pde_amd/csrc/pdehip_block2_loops.h"

using namespace pdehip;

extern "C" {
// the oracle (compiled into pdehip_shim.c)
int oracle_set_ghost_cells(const pdehip_grid_t *g, int ncomp, const pdehip_bc_face_t *faces, void *data_full);
int oracle_laplace_scaled(const pdehip_grid_t *g, const void *in_full, void *out_full, double s1, double s2);
int oracle_laplace_euler(const pdehip_grid_t *g, const void *in_full, const void *y_full, void *out_full, double s1, double s2);
int oracle_cahn_hilliard_mu(const pdehip_grid_t *g, const void *c_full, void *mu_full, double gamma);
int oracle_lincomb(const pdehip_grid_t *g, int ncomp, void *out_full, const void *y_full, int nk, const double *coef, const void *const *k);
int oracle_rk4_combine(const pdehip_grid_t *g, int ncomp, void *y, const void *k1, const void *k2, const void *k3, const void *k4);
int oracle_rkf45_combine(const pdehip_grid_t *g, int ncomp, const void *y, void *ynew, const void *const *k6, double *err);
int oracle_euler_adaptive_combine(const pdehip_grid_t *g, int ncomp, const void *y, const void *rate, double dt, const void *half, const void *k,
                                  void *out, double *err);
int oracle_rhs_scaled(const pdehip_grid_t *g, const pdehip_rhs_t *rhs, void *y_full, void *k_out_full, double dt);
// the rest of the shim
int pdehip_layout(const pdehip_grid_t *g, int64_t *out8);
int shim_set_error(int code, const char *msg);
}

namespace {

enum { E_VALUE = 1, E_NOTIMPL = 2, E_RUNTIME = 3 };

int failf(int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    return shim_set_error(code, buf);
}

#define OTRY(expr)                                                                      \
    do {                                                                                \
        int _rc = (expr);                                                               \
        if (_rc) return failf(E_RUNTIME, "shim: %s failed with code %d", #expr, _rc);   \
    } while (0)

struct Comm {
    std::string dir;
    int rank = 0, size = 1;
    std::vector<long> sent, received;   // per-peer sequence numbers
    void *ext[4] = {nullptr, nullptr, nullptr, nullptr};
    size_t ext_bytes = 0;
    // operations of the open group
    struct Op { bool is_send; void *p; size_t bytes; int peer; };
    std::vector<Op> group;
    bool in_group = false;
    std::vector<char> stg[3][2][2];   // block decomposition: packed faces, [axis][side][0 send / 1 receive]
    std::vector<char> ext2[2], msg2[2];   // fast block loop (pdehip_block2_loops.h): boxes with two halo layers, send / receive buffer
};

double timeout_seconds()
{
    const char *e = getenv("PDEHIP_SHIM_COMM_TIMEOUT");
    return e ? atof(e) : 60.0;
}

int post_send(Comm *c, const void *p, size_t bytes, int peer)
{
    char tmp[700], fin[700];
    const long q = c->sent[peer]++;
    snprintf(tmp, sizeof(tmp), "%s/t_%d_%d_%ld", c->dir.c_str(), c->rank, peer, q);
    snprintf(fin, sizeof(fin), "%s/m_%d_%d_%ld", c->dir.c_str(), c->rank, peer, q);
    FILE *f = fopen(tmp, "wb");
    if (!f) return failf(E_RUNTIME, "shim comm: cannot write %s: %s", tmp, strerror(errno));
    const size_t w = fwrite(p, 1, bytes, f);
    fclose(f);
    if (w != bytes) return failf(E_RUNTIME, "shim comm: short write to %s", tmp);
    if (rename(tmp, fin) != 0) return failf(E_RUNTIME, "shim comm: rename failed: %s", strerror(errno));
    return 0;
}

int wait_recv(Comm *c, void *p, size_t bytes, int peer)
{
    char fin[700];
    const long q = c->received[peer]++;
    snprintf(fin, sizeof(fin), "%s/m_%d_%d_%ld", c->dir.c_str(), peer, c->rank, q);
    const auto t0 = std::chrono::steady_clock::now();
    const double limit = timeout_seconds();
    for (;;) {
        struct stat sb;
        if (stat(fin, &sb) == 0) {
            if ((size_t)sb.st_size != bytes)
                return failf(E_RUNTIME, "shim comm: rank %d expected %zu bytes from rank %d (message %ld) but the peer sent %zu — mismatched send/recv",
                             c->rank, bytes, peer, q, (size_t)sb.st_size);
            FILE *f = fopen(fin, "rb");
            if (!f) return failf(E_RUNTIME, "shim comm: cannot read %s", fin);
            const size_t r = fread(p, 1, bytes, f);
            fclose(f);
            unlink(fin);
            if (r != bytes) return failf(E_RUNTIME, "shim comm: short read from %s", fin);
            return 0;
        }
        if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > limit)
            return failf(E_RUNTIME, "shim comm: rank %d timed out after %.0f s waiting for message %ld from rank %d — mismatched send/recv or a dead peer",
                         c->rank, limit, q, peer);
        std::this_thread::sleep_for(std::chrono::microseconds(200));
    }
}

size_t layer_bytes(const pdehip_grid_t *g)
{
    int64_t lay[8];
    pdehip_layout(g, lay);
    return (size_t)lay[7] * (g->dtype == PDEHIP_F64 ? 8 : 4);
}
size_t full_bytes(const pdehip_grid_t *g)
{
    int64_t lay[8];
    pdehip_layout(g, lay);
    return (size_t)lay[2] * (g->dtype == PDEHIP_F64 ? 8 : 4);
}

struct HostOps {
    Comm *c;
    long slab_thick() { const char *e = getenv("PDEHIP_SLAB_THICK"); const long v = e ? atol(e) : 0; return v < 0 ? 0 : v; }
    int deep_mode() { const char *e = getenv("PDEHIP_SLAB_DEEP_MODE"); const int v = e ? atoi(e) : 0; return (v >= 1 && v <= 4) ? v : 3; }
    void *halo() { return (void *)1; }
    int record(int, void *) { return 0; }
    int wait(void *, int) { return 0; }
    int group_start() { c->in_group = true; c->group.clear(); return 0; }
    int send(const void *p, size_t bytes, int peer, void *) { c->group.push_back({true, const_cast<void *>(p), bytes, peer}); return 0; }
    int recv(void *p, size_t bytes, int peer, void *) { c->group.push_back({false, p, bytes, peer}); return 0; }
    int group_end()
    {
        c->in_group = false;
        for (auto &op : c->group)
            if (op.is_send) SLAB_TRY(post_send(c, op.p, op.bytes, op.peer));
        for (auto &op : c->group)
            if (!op.is_send) SLAB_TRY(wait_recv(c, op.p, op.bytes, op.peer));
        c->group.clear();
        return 0;
    }
    int copy(void *dst, const void *src, size_t bytes, void *) { memmove(dst, src, bytes); return 0; }
    int zero(void *p, size_t bytes, void *) { memset(p, 0, bytes); return 0; }
    int refresh(void *bc_program, double t, const void *in, void *st) { return pdehip_bcprog_run(bc_program, t, in, st); }
    int fail(const char *msg) { return failf(E_NOTIMPL, "%s", msg); }
    int fail_runtime(const char *fmt, double v) { return failf(E_RUNTIME, fmt, v); }

    int lap(const pdehip_grid_t *gs, void *in, const void *y, void *out, int kind, double s1, double s2, double gamma,
            const pdehip_bc_face_t *faces, void *st, const StageFuse *sf)
    {
        OTRY(oracle_set_ghost_cells(gs, 1, faces, in));   // faces marked SKIP keep the (exchanged / neighbouring) layers
        if (kind == slab::K_EULER) { OTRY(oracle_laplace_euler(gs, in, y, out, s1, s2)); return 0; }
        if (kind == slab::K_SCALED) { OTRY(oracle_laplace_scaled(gs, in, out, s1, s2)); return 0; }
        if (kind == slab::K_CH_MU) { OTRY(oracle_cahn_hilliard_mu(gs, in, out, gamma)); return 0; }
        // K_STAGE: slope, then the combination (kinds 1, 2 and 4 do not store the slope)
        void *k = out, *tmp = nullptr;
        if (sf->kind == 1 || sf->kind == 2 || sf->kind == 4) k = tmp = calloc(1, full_bytes(gs));
        int rc = oracle_laplace_scaled(gs, in, k, s1, s2);
        if (!rc) rc = combine(gs, k, *sf, st);
        free(tmp);
        return rc;
    }
    int combine(const pdehip_grid_t *g, void *k, const StageFuse &sf, void *)
    {
        if (sf.kind == 0) {
            const void *ks[6];
            double cf[6];
            int n = 0;
            for (; n < 5 && sf.k[n]; n++) { ks[n] = sf.k[n]; cf[n] = sf.c[n]; }
            ks[n] = k; cf[n] = sf.c_new;
            OTRY(oracle_lincomb(g, 1, sf.out2, sf.y, n + 1, cf, ks));
            return 0;
        }
        if (sf.kind == 1) {
            if (sf.out2 != sf.y) return failf(E_RUNTIME, "internal: the RK4 update works in place");
            OTRY(oracle_rk4_combine(g, 1, sf.out2, sf.k[0], sf.k[1], sf.k[2], k));
            return 0;
        }
        if (sf.kind == 2) {
            const void *k6[6] = {sf.k[0], sf.k[0], sf.k[1], sf.k[2], sf.k[3], k};
            double err = 0;
            OTRY(oracle_rkf45_combine(g, 1, sf.y, sf.out2, k6, &err));
            // like the device kernel: atomic max onto the (zeroed) scalar, NaN wins
            if (err != err || *sf.err != *sf.err) *sf.err = NAN; else if (err > *sf.err) *sf.err = err;
            return 0;
        }
        if (sf.kind == 4) {
            double err = 0;
            OTRY(oracle_euler_adaptive_combine(g, 1, sf.y, sf.k[0], sf.c[0], sf.k[1], k, sf.out2, &err));
            if (err != err || *sf.err != *sf.err) *sf.err = NAN; else if (err > *sf.err) *sf.err = err;
            return 0;
        }
        return failf(E_NOTIMPL, "internal: unknown stage kind %d", sf.kind);
    }

    // Two-level sweep over the sub-slab `gs` (layers r = 1..count relative to `in`, which points ONE LAYER BEFORE the first
    // layer to update): level 1 on every layer level 2 reads, level 2 on the sub-slab (or its first / last `ends` layers).
    // Sides with a halo (xplain 1: both, 2: upper only, 3: lower only) hold TWO real layers beyond the sub-slab; the other
    // sides end in the physical faces faces1[0/1] (level 0 -> 1) and faces2[0/1] (level 1 -> 2).
    template <class L1, class L2>
    int two_level(const pdehip_grid_t *gs, const void *in, const pdehip_bc_face_t *faces1, const pdehip_bc_face_t *faces2, int xplain, int ends,
                  L1 &&level1, L2 &&level2)
    {
        const long count = gs->shape[0];
        const bool lo = xplain == 1 || xplain == 3, hi = xplain == 1 || xplain == 2;
        const size_t lp = layer_bytes(gs);
        const long a0 = lo ? -1 : 0, a1 = hi ? count + 2 : count + 1;   // window: ghost layer a0, own layers a0+1 .. a1-1, ghost a1
        pdehip_grid_t gw = *gs;
        gw.shape[0] = a1 - a0 - 1;
        const size_t wbytes = full_bytes(&gw);
        std::vector<char> src(wbytes), l1(wbytes, 0);
        memcpy(src.data(), static_cast<const char *>(in) + a0 * (long)lp, wbytes);   // `in` itself is never written
        pdehip_bc_face_t f1[2 * PDEHIP_MAX_DIM], f2[2 * PDEHIP_MAX_DIM];
        for (int i = 0; i < 2 * PDEHIP_MAX_DIM; i++) { f1[i] = faces1[i]; f2[i] = faces2[i]; }
        if (lo) f1[0].kind = f2[0].kind = PDEHIP_BC_SKIP;
        if (hi) f1[1].kind = f2[1].kind = PDEHIP_BC_SKIP;
        else if (lo) { f1[1].index1 += 1; f1[1].index2 += 1; }    // the window starts one layer before the sub-slab
        OTRY(oracle_set_ghost_cells(&gw, 1, f1, src.data()));
        SLAB_TRY(level1(&gw, src.data(), l1.data()));
        char *v = l1.data() + (-a0) * (long)lp;                     // level 1 as a full array of the sub-slab: r = 0 is its ghost layer
        OTRY(oracle_set_ghost_cells(gs, 1, f2, v));
        auto range = [&](long first, long cnt) -> int {
            if (cnt <= 0) return 0;
            pdehip_grid_t g2 = *gs;
            g2.shape[0] = cnt;
            return level2(&g2, v + (first - 1) * (long)lp, (first - 1) * (long)lp);
        };
        if (!ends) return range(1, count);
        SLAB_TRY(range(1, ends));
        return range(count - ends + 1, ends);
    }
    int euler2(const pdehip_grid_t *gs, const void *in, void *out, double D, double dt, const pdehip_bc_face_t *faces, void *, bool *done,
               int xplain, bool dry, int ends)
    {
        *done = gs->ndim >= 2;
        if (dry || !*done) return 0;
        return two_level(gs, in, faces, faces, xplain, ends,
                         [&](const pdehip_grid_t *gw, void *src, void *l1) -> int { OTRY(oracle_laplace_euler(gw, src, src, l1, D, dt)); return 0; },
                         [&](const pdehip_grid_t *g2, void *v, long off) -> int {
                             OTRY(oracle_laplace_euler(g2, v, v, static_cast<char *>(out) + off, D, dt));
                             return 0;
                         });
    }
    int ch_fused(const pdehip_grid_t *gs, const void *in, void *out, double gamma, double dt, bool euler, const pdehip_bc_face_t *fc,
                 const pdehip_bc_face_t *fm, void *st, bool *done, int xplain, bool dry, const StageFuse *sf)
    {
        *done = gs->ndim >= 2;
        if (dry || !*done) return 0;
        return two_level(gs, in, fc, fm, xplain, 0,
                         [&](const pdehip_grid_t *gw, void *src, void *l1) -> int { OTRY(oracle_cahn_hilliard_mu(gw, src, l1, gamma)); return 0; },
                         [&](const pdehip_grid_t *g2, void *mu, long off) -> int {
                             if (euler) { OTRY(oracle_laplace_euler(g2, mu, static_cast<const char *>(in) + off, static_cast<char *>(out) + off, 1.0, dt)); return 0; }
                             if (!sf) { OTRY(oracle_laplace_scaled(g2, mu, static_cast<char *>(out) + off, 1.0, dt)); return 0; }
                             void *k = out, *tmp = nullptr;
                             if (sf->kind == 1 || sf->kind == 2 || sf->kind == 4) k = tmp = calloc(1, full_bytes(g2));
                             int rc = oracle_laplace_scaled(g2, mu, k, 1.0, dt);
                             if (!rc) rc = combine(g2, k, *sf, st);
                             free(tmp);
                             return rc;
                         });
    }
    // MAX over all ranks, NaN wins: every rank sends its value to every other rank (one group)
    int allreduce_max(double *scalar, void *st)
    {
        if (c->size == 1) return 0;
        std::vector<double> all(c->size, 0.0);
        all[c->rank] = *scalar;
        SLAB_TRY(group_start());
        for (int p = 0; p < c->size; p++)
            if (p != c->rank) { SLAB_TRY(send(scalar, sizeof(double), p, st)); SLAB_TRY(recv(&all[p], sizeof(double), p, st)); }
        SLAB_TRY(group_end());
        double m = all[0];
        bool nan = false;
        for (double v : all) { if (v != v) nan = true; else if (v > m || m != m) m = v; }
        *scalar = nan ? NAN : m;
        return 0;
    }
    int read_scalar(double *host, const double *dev, void *) { *host = *dev; return 0; }
    // --- block decomposition (csrc/pdehip_block_loops.h) ---
    int64_t lay[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // pdehip_layout of the local block (set by the entry point)
    void *stage(int axis, int side, bool recv) { return c->stg[axis][side][recv ? 1 : 0].data(); }
    int face_copy(const block::Geo &q, void *buf, const block::FaceJob *jobs, int njobs, bool pack, void *)
    {
        const int o = 3 - q.ndim;
        const long pitch[3] = {(long)lay[0], (long)lay[1], 1};
        for (int j = 0; j < njobs; j++) {
            const int axis = jobs[j].axis;
            int others[2], k = 0;
            for (int a = 0; a < q.ndim; a++)
                if (a != axis) others[k++] = a;
            const long m1 = k == 2 ? q.n[others[0]] : 1, m2 = q.n[others[k - 1]];
            const long q1 = k == 2 ? pitch[o + others[0]] : 0, q2 = pitch[o + others[k - 1]];
            const long base = (long)lay[3] + jobs[j].index * pitch[o + axis];
            char *b = static_cast<char *>(buf), *pk = static_cast<char *>(jobs[j].packed);
            for (long u = 0; u < m1; u++)
                for (long v = 0; v < m2; v++) {
                    char *cell = b + (base + u * q1 + v * q2) * (long)q.esz, *slot = pk + (u * m2 + v) * (long)q.esz;
                    if (pack) memcpy(slot, cell, q.esz); else memcpy(cell, slot, q.esz);
                }
        }
        return 0;
    }
    int lincomb(const pdehip_grid_t *g, void *out, const void *y, int n, const double *cf, const void *const *k, void *) { OTRY(oracle_lincomb(g, 1, out, y, n, cf, k)); return 0; }
    int rk4_combine(const pdehip_grid_t *g, void *y, const void *k1, const void *k2, const void *k3, const void *k4, void *) { OTRY(oracle_rk4_combine(g, 1, y, k1, k2, k3, k4)); return 0; }
    int rkf45_combine(const pdehip_grid_t *g, const void *y, void *ynew, const void *const *k6, double *err, void *) { OTRY(oracle_rkf45_combine(g, 1, y, ynew, k6, err)); return 0; }
    int euler_adaptive_combine(const pdehip_grid_t *g, const void *y, const void *rate, double dt, const void *half, const void *k, void *out, double *err, void *)
    {
        OTRY(oracle_euler_adaptive_combine(g, 1, y, rate, dt, half, k, out, err));
        return 0;
    }
};

int make_block(Comm *c, const pdehip_grid_t *g, const int *nb6, block::Geo *q)
{
    if (g->ndim < 2) return failf(E_NOTIMPL, "block decomposition: 2-D and 3-D grids");
    q->ndim = g->ndim;
    q->esz = g->dtype == PDEHIP_F64 ? 8 : 4;
    for (int a = 0; a < 3; a++) { q->n[a] = a < g->ndim ? g->shape[a] : 1; q->nb[a][0] = q->nb[a][1] = -1; }
    for (int a = 0; a < g->ndim; a++)
        for (int side = 0; side < 2; side++) {
            const int peer = nb6[2 * a + side];
            if (peer >= c->size) return failf(E_VALUE, "block: neighbour rank %d outside of world size %d", peer, c->size);
            q->nb[a][side] = peer < 0 ? -1 : peer;
        }
    for (int a = 0; a < g->ndim; a++)
        for (int side = 0; side < 2; side++)
            for (int r = 0; r < 2; r++) c->stg[a][side][r].resize(q->face_elems(a) * q->esz);
    return 0;
}

int make_geo(const pdehip_grid_t *g, slab::Geo *q)
{
    int64_t lay[8];
    int rc = pdehip_layout(g, lay);
    if (rc) return rc;
    q->nloc = g->shape[0];
    q->esz = g->dtype == PDEHIP_F64 ? 8 : 4;
    q->lp = (size_t)lay[7] * q->esz;
    return 0;
}

Comm *serial_context()
{
    static Comm ctx;
    return &ctx;
}
int context(void *comm, int lower, int upper, Comm **out)
{
    Comm *c = static_cast<Comm *>(comm);
    if (!c) {
        if (lower >= 0 || upper >= 0) return failf(E_VALUE, "a slab with neighbours needs a communicator");
        c = serial_context();
    }
    *out = c;
    return 0;
}

}  // namespace

extern "C" {

int pdehip_comm_unique_id(const char *, void *id128)
{
    char dir[] = "/tmp/pdehip_shim_comm_XXXXXX";
    if (!mkdtemp(dir)) return failf(E_RUNTIME, "shim comm: mkdtemp failed");
    memset(id128, 0, 128);
    memcpy(id128, dir, strlen(dir));
    return 0;
}

int pdehip_comm_create(const char *, const void *id128, int rank, int size, void **comm)
{
    if (!id128 || !comm) return failf(E_VALUE, "comm_create: NULL pointer");
    if (rank < 0 || rank >= size) return failf(E_VALUE, "comm_create: rank %d outside of world size %d", rank, size);
    Comm *c = new Comm();
    c->dir = std::string(static_cast<const char *>(id128), strnlen(static_cast<const char *>(id128), 127));
    c->rank = rank; c->size = size;
    c->sent.assign(size, 0);
    c->received.assign(size, 0);
    // (rank 0 may already have finished and removed the directory when no message ever had to travel - an expression without
    // operators under a loaded machine: seen twice with four pytest workers; a late rank then simply creates it again)
    struct stat sb;
    if (stat(c->dir.c_str(), &sb) != 0 && mkdir(c->dir.c_str(), 0700) != 0 && errno != EEXIST) {
        delete c;
        return failf(E_RUNTIME, "shim comm: mailbox directory %s does not exist and cannot be created", static_cast<const char *>(id128));
    }
    *comm = c;
    return 0;
}

int pdehip_comm_destroy(void *comm)
{
    Comm *c = static_cast<Comm *>(comm);
    if (!c) return 0;
    for (auto &e : c->ext) free(e);
    if (c->rank == 0) rmdir(c->dir.c_str());   // succeeds once every message was consumed
    delete c;
    return 0;
}

int pdehip_comm_info(void *comm, int *out5, char *pci_bus_id, size_t n)
{
    if (!comm || !out5) return failf(E_VALUE, "comm_info: NULL pointer");
    Comm *c = static_cast<Comm *>(comm);
    out5[0] = c->size; out5[1] = c->rank; out5[2] = c->rank; out5[3] = 0; out5[4] = c->rank;   // (mailbox transport: one "device" per rank)
    if (pci_bus_id && n > 0) snprintf(pci_bus_id, n, "shim:%02d", c->rank);
    return 0;
}

int pdehip_halo_exchange(void *comm, const pdehip_grid_t *g_local, void *buf_full, int lower, int upper, void *stream)
{
    if (!comm || !buf_full) return failf(E_VALUE, "halo_exchange: NULL pointer");
    slab::Geo q;
    SLAB_TRY(make_geo(g_local, &q));
    HostOps ops{static_cast<Comm *>(comm)};
    return slab::exchange(ops, q, buf_full, lower, upper, stream);
}

int pdehip_allreduce_max(void *comm, double *scalar, void *stream)
{
    if (!comm || !scalar) return failf(E_VALUE, "allreduce_max: NULL pointer");
    HostOps ops{static_cast<Comm *>(comm)};
    return ops.allreduce_max(scalar, stream);
}

int pdehip_slab_euler_run(void *comm, const pdehip_grid_t *g_local, const pdehip_rhs_t *rhs, int lower, int upper, void *buf_a, void *buf_b,
                          double dt, int64_t nsteps, void **result, void *stream)
{
    if (!comm || !rhs || !buf_a || !buf_b || !result) return failf(E_VALUE, "slab_euler_run: NULL pointer");
    if (rhs->kind != PDEHIP_RHS_DIFFUSION) return failf(E_NOTIMPL, "slab_euler_run implements the diffusion right-hand side");
    slab::Geo q;
    SLAB_TRY(make_geo(g_local, &q));
    HostOps ops{static_cast<Comm *>(comm)};
    if (rhs->bc_program) return failf(E_NOTIMPL, "slab_euler_run: time-dependent boundary conditions run through pdehip_slab_euler_sweeps");
    return slab::euler_run(ops, g_local, q, rhs, lower, upper, buf_a, buf_b, dt, nsteps, result, stream);
}

int pdehip_slab_euler2_supported(const pdehip_grid_t *g_local, const pdehip_rhs_t *rhs, int *ok)
{
    // mimic the device rule: 3-D, >= 4 own layers, scalar first-order faces
    *ok = rhs->kind == PDEHIP_RHS_DIFFUSION && g_local->ndim == 3 && g_local->shape[0] >= 4;
    for (int i = 0; i < 2 * g_local->ndim && *ok; i++)
        if (rhs->bc_c[i].kind == PDEHIP_BC_ORDER2 || (rhs->bc_c[i].flags & PDEHIP_BCF_ARRAYS)) *ok = 0;
    return 0;
}

int pdehip_slab_euler2_run(void *comm, const pdehip_grid_t *g_local, const pdehip_rhs_t *rhs, int lower, int upper, void *buf_a, void *buf_b,
                           double dt, int64_t nsteps, void **result, void *stream)
{
    if (!comm || !rhs || !buf_a || !buf_b || !result) return failf(E_VALUE, "slab_euler2_run: NULL pointer");
    int ok = 0;
    pdehip_slab_euler2_supported(g_local, rhs, &ok);
    if (!ok) return failf(E_NOTIMPL, "slab_euler2_run: grid or faces are not covered by the two-step kernel");
    Comm *c = static_cast<Comm *>(comm);
    slab::Geo q;
    SLAB_TRY(make_geo(g_local, &q));
    pdehip_grid_t ge = *g_local;
    ge.shape[0] = q.nloc + 2;
    const size_t need = full_bytes(&ge);
    if (c->ext_bytes < need) {
        for (auto &e : c->ext) { free(e); e = nullptr; }
        c->ext[0] = calloc(1, need); c->ext[1] = calloc(1, need);
        c->ext_bytes = need;
    }
    HostOps ops{c};
    if (rhs->bc_program) return failf(E_NOTIMPL, "slab_euler2_run: time-dependent boundary conditions run through pdehip_slab_euler_sweeps");
    return slab::euler2_run(ops, g_local, q, rhs, lower, upper, buf_a, c->ext[0], c->ext[1], dt, nsteps, result, stream);
}

int pdehip_release_scratch(void)
{
    Comm *c = serial_context();
    for (auto &e : c->ext) { free(e); e = nullptr; }
    c->ext_bytes = 0;
    for (auto &a : c->stg) for (auto &b : a) for (auto &v : b) std::vector<char>().swap(v);
    for (auto &v : c->ext2) std::vector<char>().swap(v);
    for (auto &v : c->msg2) std::vector<char>().swap(v);
    return 0;
}

int pdehip_slab_euler4_supported(const pdehip_grid_t *g_local, const pdehip_rhs_t *rhs, int *ok)
{
    *ok = 0;
    if (g_local->shape[0] < 8) return 0;
    return pdehip_slab_euler2_supported(g_local, rhs, ok);
}

int pdehip_slab_euler4_run(void *comm, const pdehip_grid_t *g_local, const pdehip_rhs_t *rhs, int lower, int upper, void *buf_a, void *buf_b,
                           double dt, int64_t nsteps, void **result, void *stream)
{
    if (!comm || !rhs || !buf_a || !buf_b || !result) return failf(E_VALUE, "slab_euler4_run: NULL pointer");
    if (nsteps < 0) return failf(E_VALUE, "slab_euler4_run: negative step count");
    int ok = 0;
    pdehip_slab_euler4_supported(g_local, rhs, &ok);
    if (!ok) return failf(E_NOTIMPL, "slab_euler4_run: grid or faces are not covered by the two-step kernel, or fewer than 8 local layers");
    if (rhs->bc_program) return failf(E_NOTIMPL, "slab_euler4_run: time-dependent boundary conditions run through pdehip_slab_euler_sweeps");
    Comm *c = static_cast<Comm *>(comm);
    slab::Geo q;
    SLAB_TRY(make_geo(g_local, &q));
    pdehip_grid_t ge = *g_local;
    ge.shape[0] = q.nloc + 6;
    const size_t need = full_bytes(&ge);
    if (c->ext_bytes < need || !c->ext[3]) {
        for (auto &e : c->ext) { free(e); e = calloc(1, need > c->ext_bytes ? need : c->ext_bytes); }
        if (need > c->ext_bytes) c->ext_bytes = need;
    }
    HostOps ops{c};
    if (ops.deep_mode() >= 3) return slab::euler4p_run(ops, g_local, q, rhs, lower, upper, buf_a, c->ext, dt, nsteps, result, stream);
    return slab::euler4_run(ops, g_local, q, rhs, lower, upper, buf_a, c->ext[0], c->ext[1], dt, nsteps, result, stream);
}

int pdehip_slab_ch_supported(const pdehip_grid_t *g_local, const pdehip_rhs_t *rhs, int *ok)
{
    *ok = rhs->kind == PDEHIP_RHS_CAHN_HILLIARD && g_local->ndim == 3 && g_local->shape[0] >= 2;
    for (int i = 0; i < 2 * g_local->ndim && *ok; i++)
        if (rhs->bc_c[i].kind == PDEHIP_BC_ORDER2 || (rhs->bc_c[i].flags & PDEHIP_BCF_ARRAYS) || rhs->bc_mu[i].kind == PDEHIP_BC_ORDER2 ||
            (rhs->bc_mu[i].flags & PDEHIP_BCF_ARRAYS))
            *ok = 0;
    return 0;
}

int pdehip_slab_ch_sweep(void *comm, const pdehip_grid_t *g_local, const pdehip_rhs_t *rhs, int lower, int upper, void *c_ext, void *out_ext,
                         double dt, int euler, void *stream)
{
    if (!comm || !rhs || !c_ext || !out_ext) return failf(E_VALUE, "slab_ch_sweep: NULL pointer");
    slab::Geo q;
    SLAB_TRY(make_geo(g_local, &q));
    HostOps ops{static_cast<Comm *>(comm)};
    return slab::rhs_sweep(ops, g_local, q, rhs, lower, upper, slab::F_FUSED_CH, slab::layer(c_ext, q, 1), slab::layer(out_ext, q, 1), dt, euler != 0,
                           nullptr, stream);
}

int pdehip_slab_flags_supported(const pdehip_grid_t *g_local, const pdehip_rhs_t *rhs, int lower, int upper, int *flags)
{
    // PDEHIP_SHIM_FUSED=1 exercises the fused branches of the loops, else the plain ones
    const char *e = getenv("PDEHIP_SHIM_FUSED");
    *flags = 0;
    if (!(e && e[0] == '1') || g_local->ndim < 2) return 0;
    *flags = PDEHIP_SLAB_FUSED_STAGE;
    if (rhs->kind == PDEHIP_RHS_CAHN_HILLIARD) {
        int ok = 0;
        pdehip_rhs_t r = *rhs;
        slab::local_faces(rhs->bc_c, lower, upper, r.bc_c);
        slab::local_faces(rhs->bc_mu, lower, upper, r.bc_mu);
        pdehip_slab_ch_supported(g_local, &r, &ok);
        if (ok) *flags |= PDEHIP_SLAB_FUSED_CH; else *flags = 0;   // the stage epilogue of Cahn-Hilliard rides on the two-level sweep
    }
    return 0;
}

#define SLAB_ENTRY_PROLOGUE(name)                          \
    Comm *c;                                               \
    SLAB_TRY(context(comm, lower, upper, &c));             \
    slab::Geo q;                                           \
    SLAB_TRY(make_geo(g_local, &q));                       \
    HostOps ops{c};

int pdehip_slab_rhs_scaled(void *comm, const pdehip_grid_t *g_local, const pdehip_rhs_t *rhs, int lower, int upper, int flags, void *y_full,
                           void *k_out_full, double dt, void *stream)
{
    SLAB_ENTRY_PROLOGUE(rhs_scaled)
    return slab::rhs_sweep(ops, g_local, q, rhs, lower, upper, flags, y_full, k_out_full, dt, false, nullptr, stream, rhs->t);
}

int pdehip_slab_euler_sweeps(void *comm, const pdehip_grid_t *g_local, const pdehip_rhs_t *rhs, int lower, int upper, int flags, void *buf_a,
                             void *buf_b, double dt, int64_t nsteps, void **result, void *stream)
{
    SLAB_ENTRY_PROLOGUE(euler_sweeps)
    return slab::euler_sweeps(ops, g_local, q, rhs, lower, upper, flags, buf_a, buf_b, dt, nsteps, result, stream);
}

int pdehip_slab_rk4_run(void *comm, const pdehip_grid_t *g_local, const pdehip_rhs_t *rhs, int lower, int upper, int flags, void *y_full,
                        void *const *work5_host, double dt, int64_t nsteps, void *stream)
{
    SLAB_ENTRY_PROLOGUE(rk4_run)
    for (int64_t s = 0; s < nsteps; s++)
        SLAB_TRY(slab::rk4_step(ops, g_local, q, rhs, lower, upper, flags, y_full, work5_host, dt, stream, rhs->t + (double)s * dt));
    return 0;
}

int pdehip_slab_rkf45_run(void *comm, const pdehip_grid_t *g_local, const pdehip_rhs_t *rhs, int lower, int upper, int flags, void *y_full,
                          void *ynew_full, void *const *work7_host, double *err_dev, pdehip_adaptive_t *ctl, void **result, void *stream)
{
    if (!(ctl->tolerance > 0) || !(ctl->dt > 0)) return failf(E_VALUE, "slab_rkf45_run: tolerance and dt must be positive");
    SLAB_ENTRY_PROLOGUE(rkf45_run)
    return slab::rkf45_run(ops, g_local, q, rhs, lower, upper, flags, y_full, ynew_full, work7_host, err_dev, ctl, result, stream);
}

int pdehip_slab_euler_adaptive_run(void *comm, const pdehip_grid_t *g_local, const pdehip_rhs_t *rhs, int lower, int upper, int flags, void *y_full,
                                   void *ynew_full, void *const *work3_host, double *err_dev, pdehip_adaptive_t *ctl, void **result, void *stream)
{
    if (!(ctl->tolerance > 0) || !(ctl->dt > 0)) return failf(E_VALUE, "slab_euler_adaptive_run: tolerance and dt must be positive");
    SLAB_ENTRY_PROLOGUE(euler_adaptive_run)
    return slab::euler_adaptive_run(ops, g_local, q, rhs, lower, upper, flags, y_full, ynew_full, work3_host, err_dev, ctl, result, stream);
}

int pdehip_euler_adaptive_run(const pdehip_grid_t *g, const pdehip_rhs_t *rhs, void *y_full, void *ynew_full, void *const *work3_host,
                              double *err_dev, pdehip_adaptive_t *ctl, void **result, void *stream)
{
    int flags = 0;
    SLAB_TRY(pdehip_slab_flags_supported(g, rhs, -1, -1, &flags));
    return pdehip_slab_euler_adaptive_run(nullptr, g, rhs, -1, -1, flags, y_full, ynew_full, work3_host, err_dev, ctl, result, stream);
}

}  // extern "C"

// ---- the generic Runge-Kutta loops of csrc/pdehip_rk_loops.h on the host ------------------------------------------------------
namespace {
int shim_read(double *host, const double *dev) { *host = *dev; return 0; }

// built-in right-hand sides whose faces change with time: slope = refresh + oracle right-hand side, never fused
struct SpecEval {
    const pdehip_grid_t *g;
    const pdehip_rhs_t *rhs;
    int slope(void *in, void *k_out, double dt, double t, const StageFuse *, bool *fused, void *st)
    {
        *fused = false;
        if (rhs->bc_program) SLAB_TRY(pdehip_bcprog_run(rhs->bc_program, t, in, st));
        OTRY(oracle_rhs_scaled(g, rhs, in, k_out, dt));
        return 0;
    }
    int lincomb(void *out, const void *y, int n, const double *c, const void *const *k, void *) { OTRY(oracle_lincomb(g, 1, out, y, n, c, k)); return 0; }
    int rk4_combine(void *y, const void *k1, const void *k2, const void *k3, const void *k4, void *) { OTRY(oracle_rk4_combine(g, 1, y, k1, k2, k3, k4)); return 0; }
    int rkf45_combine(const void *y, void *ynew, const void *const *k6, double *err, void *) { OTRY(oracle_rkf45_combine(g, 1, y, ynew, k6, err)); return 0; }
    int euler_adaptive_combine(const void *y, const void *rate, double dt, const void *half, const void *k, void *out, double *err, void *)
    {
        OTRY(oracle_euler_adaptive_combine(g, 1, y, rate, dt, half, k, out, err));
        return 0;
    }
    int zero(void *p, size_t bytes, void *) { memset(p, 0, bytes); return 0; }
    int reduce_error(double *, void *) { return 0; }
    int read_scalar(double *host, const double *dev, void *) { return shim_read(host, dev); }
    int fail_runtime(const char *fmt, double v) { return failf(E_RUNTIME, fmt, v); }
};

// expression right-hand sides: the passes of pdehip_jit_rk_run through pdehip_jit_apply (gcc-built epilogues, oracle stencils)
struct JitEval {
    const pdehip_grid_t *g;
    const pdehip_jit_pass_t *passes;
    int npasses;
    void *const *fixed;
    int ncomp;
    size_t comp_bytes;
    int stage_fuse;
    void *bc_program;
    bool skip_exchange_once = false;
    // decomposed grids: the ghost layers of a pass's operand travel before the pass (pdehip_exchange_t, include/pdehip.h)
    int exchange_operand(const pdehip_grid_t *gg, const pdehip_jit_pass_t &p, void *src, void *st)
    {
        const pdehip_exchange_t *x = p.exchange;
        if (!x) return 0;
        if (skip_exchange_once) { skip_exchange_once = false; return 0; }
        int nb6[6];
        for (int i = 0; i < 6; i++) nb6[i] = x->nb6[i];
        return x->blocks ? pdehip_block_exchange(x->comm, gg, nb6, src, st) : pdehip_halo_exchange(x->comm, gg, src, x->lower, x->upper, st);
    }
    int run(int first, int count, char *in, char *k_out, const double *params, void *st)
    {
        for (int q = first; q < first + count; q++) {
            const pdehip_jit_pass_t &p = passes[q];
            auto arr = [&](int32_t idx) -> void * {
                if (idx == PDEHIP_JIT_NONE) return nullptr;
                return idx >= 0 ? fixed[idx] : (void *)(in + (size_t)(-1 - idx) * comp_bytes);
            };
            void *out = p.out >= 0 ? fixed[p.out] : (void *)(k_out + (size_t)(-1 - p.out) * comp_bytes);
            const void *ex[3] = {arr(p.extras[0]), arr(p.extras[1]), arr(p.extras[2])};
            SLAB_TRY(exchange_operand(g, p, arr(p.src), st));
            SLAB_TRY(pdehip_jit_apply(p.handle, g, arr(p.src), ex, out, params, 2, p.faces, st));
        }
        return 0;
    }
    int slope(void *in, void *k_out, double dt, double t, const StageFuse *sf, bool *fused, void *st)
    {
        *fused = false;
        const double params[2] = {dt, t};
        if (bc_program) SLAB_TRY(pdehip_bcprog_run(bc_program, t, in, st));
        const pdehip_jit_pass_t &last = passes[npasses - 1];
        const bool try_stage = sf && stage_fuse > 0 && ncomp == 1 && last.out == -1;
        SLAB_TRY(run(0, npasses - (try_stage ? 1 : 0), (char *)in, (char *)k_out, params, st));
        if (!try_stage) return 0;
        auto arr = [&](int32_t idx) -> void * {
            if (idx == PDEHIP_JIT_NONE) return nullptr;
            return idx >= 0 ? fixed[idx] : (void *)((char *)in + (size_t)(-1 - idx) * comp_bytes);
        };
        const void *ex[3] = {arr(last.extras[0]), arr(last.extras[1]), arr(last.extras[2])};
        int nk = 0;
        while (nk < 5 && sf->k[nk]) nk++;
        int done = 0;
        SLAB_TRY(exchange_operand(g, last, arr(last.src), st));
        SLAB_TRY(pdehip_jit_apply_stage(last.handle, g, arr(last.src), ex, k_out, params, 2, last.faces, sf->kind, sf->y, nk, sf->k, sf->c, sf->c_new,
                                        sf->out2, sf->err, &done, st));
        if (done) { *fused = true; return 0; }
        stage_fuse = -1;
        skip_exchange_once = true;     // (the operand of the last pass has just been exchanged)
        return run(npasses - 1, 1, (char *)in, (char *)k_out, params, st);
    }
    int lincomb(void *out, const void *y, int n, const double *c, const void *const *k, void *) { OTRY(oracle_lincomb(g, ncomp, out, y, n, c, k)); return 0; }
    int rk4_combine(void *y, const void *k1, const void *k2, const void *k3, const void *k4, void *) { OTRY(oracle_rk4_combine(g, ncomp, y, k1, k2, k3, k4)); return 0; }
    void *efield = nullptr;   // complex pairs (stage_fuse bit 1): the error field; modulus norm through the pointwise entry points
    int rkf45_combine(const void *y, void *ynew, const void *const *k6, double *err, void *st)
    {
        if (!efield) { OTRY(oracle_rkf45_combine(g, ncomp, y, ynew, k6, err)); return 0; }
        const double c[4] = {25.0 / 216, 1408.0 / 2565, 2197.0 / 4104, -1.0 / 5};
        const double r[5] = {1.0 / 360, -128.0 / 4275, -2197.0 / 75240, 1.0 / 50, 2.0 / 55};
        const void *kc[4] = {k6[0], k6[2], k6[3], k6[4]}, *kr[5] = {k6[0], k6[2], k6[3], k6[4], k6[5]};
        if (int rc = pdehip_lincomb(g, ncomp, ynew, y, 4, c, kc, st)) return rc;
        if (int rc = pdehip_lincomb(g, ncomp, efield, nullptr, 5, r, kr, st)) return rc;
        return pdehip_max_abs_pairs(g, ncomp / 2, efield, err, st);
    }
    int euler_adaptive_combine(const void *y, const void *rate, double dt, const void *half, const void *k, void *out, double *err, void *st)
    {
        if (!efield) { OTRY(oracle_euler_adaptive_combine(g, ncomp, y, rate, dt, half, k, out, err)); return 0; }
        const double one = 1.0, minus = -1.0;
        const void *kk[1] = {k};
        if (int rc = pdehip_lincomb(g, ncomp, out, half, 1, &one, kk, st)) return rc;
        kk[0] = rate;
        if (int rc = pdehip_lincomb(g, ncomp, efield, y, 1, &dt, kk, st)) return rc;
        kk[0] = out;
        if (int rc = pdehip_lincomb(g, ncomp, efield, efield, 1, &minus, kk, st)) return rc;
        return pdehip_max_abs_pairs(g, ncomp / 2, efield, err, st);
    }
    int zero(void *p, size_t bytes, void *) { memset(p, 0, bytes); return 0; }
    int reduce_error(double *err_dev, void *st)
    {
        for (int q = 0; q < npasses; q++)
            if (passes[q].exchange && passes[q].exchange->comm) return pdehip_allreduce_max(passes[q].exchange->comm, err_dev, st);
        return 0;
    }
    int read_scalar(double *host, const double *dev, void *) { return shim_read(host, dev); }
    int fail_runtime(const char *fmt, double v) { return failf(E_RUNTIME, fmt, v); }
};
}  // namespace

extern "C" {

int shim_timed_rk4_step(const pdehip_grid_t *g, const pdehip_rhs_t *rhs, void *y, void *const *w, double dt, double t)
{
    SpecEval ev{g, rhs};
    return rk::rk4_step(ev, y, w, dt, t, nullptr);
}

int shim_timed_rkf45_attempt(const pdehip_grid_t *g, const pdehip_rhs_t *rhs, void *y, void *ynew, void *const *w, double dt, double t, double *err)
{
    SpecEval ev{g, rhs};
    return rk::rkf45_attempt(ev, y, ynew, w, dt, t, err, nullptr);
}

static int jit_loop_run(int scheme, const pdehip_grid_t *g, const pdehip_jit_pass_t *passes, int npasses, void *const *fixed, int nfixed,
                      int ncomp, void *y, void *ynew, void *const *work_host, double *err_dev, double dt, double t0, int64_t nsteps,
                      pdehip_adaptive_t *ctl, int stage_fuse, void *bc_program, void **result, void *stream)
{
    if (!g || !passes || !y || !work_host || !result || (nfixed > 0 && !fixed)) return failf(E_VALUE, "jit_rk_run: NULL pointer");
    if (npasses < 1 || ncomp < 1 || nsteps < 0) return failf(E_VALUE, "jit_rk_run: bad pass / component / step count");
    if (ctl && (!ynew || !err_dev || !(ctl->tolerance > 0) || !(ctl->dt > 0))) return failf(E_VALUE, "jit_rk_run: the adaptive loop needs ynew, err_dev, tolerance > 0 and dt > 0");
    for (int q = 0; q < npasses; q++) {
        const int32_t idx[5] = {passes[q].src, passes[q].extras[0], passes[q].extras[1], passes[q].extras[2], passes[q].out};
        for (int m = 0; m < 5; m++) {
            if (idx[m] == PDEHIP_JIT_NONE && m != 0 && m != 4) continue;
            if (idx[m] == PDEHIP_JIT_NONE || idx[m] >= nfixed || idx[m] < -ncomp)
                return failf(E_VALUE, "jit_rk_run: pass %d refers to array %d (fixed: %d, components: %d)", q, (int)idx[m], nfixed, ncomp);
        }
    }
    JitEval ev{g, passes, npasses, fixed, ncomp, full_bytes(g), (stage_fuse & 1) ? 1 : 0, bc_program};
    if (stage_fuse & 2) {
        if (ncomp % 2 || !ctl) return failf(E_VALUE, "jit_rk_run: complex pairs need an even number of components and the adaptive loop");
        ev.efield = work_host[scheme == 1 ? 3 : 7];
        if (!ev.efield) return failf(E_VALUE, "jit_rk_run: complex pairs need the error field as one more work array");
    }
    if (scheme == 1) return rk::euler_adaptive_run(ev, y, ynew, work_host, err_dev, ctl, result, stream);
    if (ctl) return rk::rkf45_run(ev, y, ynew, work_host, err_dev, ctl, result, stream);
    for (int64_t s = 0; s < nsteps; s++) SLAB_TRY(rk::rk4_step(ev, y, work_host, dt, t0 + (double)s * dt, stream));
    *result = y;
    return 0;
}

int pdehip_jit_rk_run(const pdehip_grid_t *g, const pdehip_jit_pass_t *passes, int npasses, void *const *fixed, int nfixed,
                      int ncomp, void *y, void *ynew, void *const *work_host, double *err_dev, double dt, double t0, int64_t nsteps,
                      pdehip_adaptive_t *ctl, int stage_fuse, void *bc_program, void **result, void *stream)
{
    return jit_loop_run(0, g, passes, npasses, fixed, nfixed, ncomp, y, ynew, work_host, err_dev, dt, t0, nsteps, ctl, stage_fuse, bc_program, result, stream);
}

int pdehip_jit_euler_adaptive_run(const pdehip_grid_t *g, const pdehip_jit_pass_t *passes, int npasses, void *const *fixed, int nfixed,
                                  int ncomp, void *y, void *ynew, void *const *work3_host, double *err_dev, pdehip_adaptive_t *ctl,
                                  int stage_fuse, void *bc_program, void **result, void *stream)
{
    if (!ctl) return failf(E_VALUE, "jit_euler_adaptive_run: NULL pointer");
    return jit_loop_run(1, g, passes, npasses, fixed, nfixed, ncomp, y, ynew, work3_host, err_dev, 0.0, 0.0, 0, ctl, stage_fuse, bc_program, result, stream);
}


int pdehip_block_exchange(void *comm, const pdehip_grid_t *g_local, const int *nb6, void *buf_full, void *stream)
{
    if (!g_local || !nb6 || !buf_full) return failf(E_VALUE, "block_exchange: NULL pointer");
    Comm *c = static_cast<Comm *>(comm);
    if (!c) c = serial_context();
    block::Geo q;
    SLAB_TRY(make_block(c, g_local, nb6, &q));
    HostOps ops{c};
    SLAB_TRY(pdehip_layout(g_local, ops.lay));
    return block::exchange(ops, q, buf_full, stream);
}

int pdehip_block_run(void *comm, const pdehip_grid_t *g_local, const pdehip_rhs_t *rhs, const int *nb6, int fuse_stage, int scheme, void *y_full,
                     void *ynew_full, void *const *work_host, double *err_dev, double dt, int64_t nsteps, pdehip_adaptive_t *ctl, void **result,
                     void *stream)
{
    if (!g_local || !nb6 || !rhs || !y_full || !result) return failf(E_VALUE, "block_run: NULL pointer");
    if (scheme < 0 || scheme > 3) return failf(E_VALUE, "block_run: scheme 0 (Euler), 1 (RK4), 2 (adaptive RKF45) or 3 (adaptive Euler)");
    if (scheme != 1 && !ynew_full) return failf(E_VALUE, "block_run: the scheme needs a second state array");
    if (scheme >= 1 && !work_host) return failf(E_VALUE, "block_run: the scheme needs work arrays");
    if (scheme >= 2 && (!ctl || !err_dev || !(ctl->tolerance > 0) || !(ctl->dt > 0))) return failf(E_VALUE, "block_run: the adaptive loop needs ctl, err_dev, tolerance > 0, dt > 0");
    Comm *c = static_cast<Comm *>(comm);
    if (!c) c = serial_context();
    block::Geo q;
    SLAB_TRY(make_block(c, g_local, nb6, &q));
    HostOps ops{c};
    SLAB_TRY(pdehip_layout(g_local, ops.lay));
    return block::run(ops, g_local, q, rhs, fuse_stage != 0, scheme, y_full, ynew_full, work_host, err_dev, dt, nsteps, ctl, result, stream);
}

}  // extern "C"


// ---- the fast block loop (csrc/pdehip_block2_loops.h) on the host: the SAME schedule template, oracle passes as kernels ------------
namespace {
struct HostOps2 {
    Comm *c;
    pdehip_grid_t g_box, g_ext;       // own cells / the grid of `ext` (two cells larger along the first two axes)
    int64_t lay[8];                   // pdehip_layout of g_ext
    const pdehip_bc_face_t *faces;
    double D, dt;
    size_t es;
    void *halo() { return (void *)1; }
    size_t esz() const { return es; }
    void *msg(bool send, size_t elem_off) { return c->msg2[send ? 0 : 1].data() + elem_off * es; }
    int record2(int, void *) { return 0; }
    int wait2(void *, int) { return 0; }
    int group_start() { c->in_group = true; c->group.clear(); return 0; }
    int send(const void *p, size_t bytes, int peer, void *) { c->group.push_back({true, const_cast<void *>(p), bytes, peer}); return 0; }
    int recv(void *p, size_t bytes, int peer, void *) { c->group.push_back({false, p, bytes, peer}); return 0; }
    int group_end()
    {
        c->in_group = false;
        for (auto &op : c->group)
            if (op.is_send) SLAB_TRY(post_send(c, op.p, op.bytes, op.peer));
        for (auto &op : c->group)
            if (!op.is_send) SLAB_TRY(wait_recv(c, op.p, op.bytes, op.peer));
        c->group.clear();
        return 0;
    }
    int zcut = 0;                     // the fastest axis is cut: g_ext is two cells larger along it as well (own cell k = its interior cell k + 1)
    long elem_of(long i, long j, long k) const { return (long)lay[3] + (i + 1) * (long)lay[0] + (j + 1) * (long)lay[1] + k + zcut; }   // own-cell coordinates
    int pack(const block2::Plan &p, void *ext, bool is_pack, void *)
    {
        const int nreg = is_pack ? p.nsend : p.nrecv;
        for (int r = 0; r < nreg; r++) {
            const block2::Region &reg = is_pack ? p.send[r] : p.recv[r];
            char *buf = static_cast<char *>(msg(is_pack, reg.offset));
            size_t m = 0;
            for (long i = 0; i < reg.box.n[0]; i++)
                for (long j = 0; j < reg.box.n[1]; j++)
                    for (long k = 0; k < reg.box.n[2]; k++, m++) {
                        char *cell = static_cast<char *>(ext) + elem_of(reg.box.lo[0] + i, reg.box.lo[1] + j, reg.box.lo[2] + k) * (long)es;
                        if (is_pack) memcpy(buf + m * es, cell, es); else memcpy(cell, buf + m * es, es);
                    }
        }
        return 0;
    }
    // two steps from `cur` with whatever its halo cells hold; cells `only_rim` (or all own cells) of the result into `nxt`
    int two_steps(const block2::Plan &p, const void *cur, void *nxt, int which)   // which: 0 all own cells, 1 the rim, 2 the interior
    {
        const size_t bytes = (size_t)lay[2] * es;
        std::vector<char> u0(bytes), u1(bytes, 0), u2(bytes, 0);
        memcpy(u0.data(), cur, bytes);
        const long hz = zcut ? 2 : 0;      // halo cells of a cut fastest axis travel with their rows (the edges)
        auto at = [&](std::vector<char> &v, long i, long j) { return v.data() + elem_of(i, j, -hz) * (long)es; };
        const long n0 = p.n[0], n1 = p.n[1], rowb = (p.n[2] + 2 * hz) * (long)es;
        // an uncut (periodic) axis wraps: its two halo layers from the own cells - rows first, then whole planes (the edges follow)
        auto wrap_xy = [&](std::vector<char> &v, long depth) {
            if (!p.cut[1])
                for (long i = -depth; i < n0 + depth; i++)      // (the halo planes of a cut first axis hold received rows: they wrap too)
                    for (long h = 1; h <= depth; h++) { memcpy(at(v, i, -h), at(v, i, n1 - h), rowb); memcpy(at(v, i, n1 - 1 + h), at(v, i, h - 1), rowb); }
            if (!p.cut[0])
                for (long h = 1; h <= depth; h++)
                    for (long j = -depth; j < n1 + depth; j++) { memcpy(at(v, -h, j), at(v, n0 - h, j), rowb); memcpy(at(v, n0 - 1 + h, j), at(v, h - 1, j), rowb); }
        };
        pdehip_bc_face_t f[2 * PDEHIP_MAX_DIM];
        memset(f, 0, sizeof(f));                 // SKIP on the first two axes: their ghost layers hold halo data
        if (!zcut) { f[4] = faces[4]; f[5] = faces[5]; }   // the fastest axis: the periodic condition of the grid - or halo data like the others
        wrap_xy(u0, 2);
        OTRY(oracle_set_ghost_cells(&g_ext, 1, f, u0.data()));
        OTRY(oracle_laplace_euler(&g_ext, u0.data(), u0.data(), u1.data(), D, dt));     // level 1 on the own cells widened by one
        OTRY(oracle_set_ghost_cells(&g_ext, 1, f, u1.data()));
        OTRY(oracle_laplace_euler(&g_ext, u1.data(), u1.data(), u2.data(), D, dt));     // level 2: valid on the own cells
        for (long i = 0; i < n0; i++)
            for (long j = 0; j < n1; j++)
                for (long k = 0; k < p.n[2]; k++) {
                    const bool rim = (p.cut[0] && (i < 2 || i >= n0 - 2)) || (p.cut[1] && (j < 2 || j >= n1 - 2)) || (p.cut[2] && (k < 2 || k >= p.n[2] - 2));
                    if ((which == 1 && !rim) || (which == 2 && rim)) continue;
                    memcpy(static_cast<char *>(nxt) + elem_of(i, j, k) * (long)es, u2.data() + elem_of(i, j, k) * (long)es, es);
                }
        return 0;
    }
    int sweep2(const block2::Plan &p, void *cur, void *nxt, bool interior, void *) { return two_steps(p, cur, nxt, interior ? 2 : 0); }
    int rim2(const block2::Plan &p, void *cur, void *nxt, void *) { return p.nrim ? two_steps(p, cur, nxt, 1) : 0; }
};

bool block2_covers(const pdehip_grid_t *g_local, const pdehip_rhs_t *rhs, const int *cut3)
{
    if (g_local->ndim != 3 || rhs->kind != PDEHIP_RHS_DIFFUSION || rhs->bc_program) return false;   // (fp32 with a cut fastest axis: covered by the library since the end of round 6, pdehip_comm.hip: block2_check)
    const long vec = g_local->dtype == PDEHIP_F64 ? 2 : 4;
    if (g_local->shape[2] % vec || g_local->shape[0] < 4 || g_local->shape[1] < 4 || g_local->shape[2] < 4 || (cut3[2] && g_local->shape[2] < 8)) return false;
    for (int a = 0; a < 3; a++) {
        if (cut3[a]) continue;
        for (int side = 0; side < 2; side++) {
            const pdehip_bc_face_t &r = rhs->bc_c[2 * a + side];
            if (r.kind != PDEHIP_BC_ORDER1 || r.flags != 0 || r.index1 != (side ? 0 : g_local->shape[a] - 1) || r.const_v != 0.0 || r.factor1 != 1.0) return false;
        }
    }
    return true;
}
}  // namespace

extern "C" {

int pdehip_block2_supported(const pdehip_grid_t *g_local, const pdehip_rhs_t *rhs, const int *cut3, int *ok)
{
    if (!g_local || !rhs || !cut3 || !ok) return failf(E_VALUE, "block2_supported: NULL pointer");
    *ok = block2_covers(g_local, rhs, cut3) ? 1 : 0;
    return 0;
}

int pdehip_block2_euler_run(void *comm, const pdehip_grid_t *g_local, const pdehip_rhs_t *rhs, const int *dims3, const int *coords3,
                            const int *cut3, void *buf_a, void *buf_b, double dt, int64_t nsteps, void **result, void *stream)
{
    if (!g_local || !rhs || !dims3 || !coords3 || !cut3 || !buf_a || !buf_b || !result) return failf(E_VALUE, "block2_euler_run: NULL pointer");
    if (nsteps < 0 || nsteps % 2) return failf(E_VALUE, "block2_euler_run: the step count must be even (two steps per sweep)");
    if (!block2_covers(g_local, rhs, cut3)) return failf(E_NOTIMPL, "block2_euler_run: grid, equation or faces are not covered (ask pdehip_block2_supported)");
    Comm *c = static_cast<Comm *>(comm);
    const bool any = cut3[0] || cut3[1] || cut3[2];
    if (!c) {
        if (any) return failf(E_VALUE, "a block with neighbours needs a communicator");
        c = serial_context();
    }
    if (any && (long)dims3[0] * dims3[1] * dims3[2] != c->size) return failf(E_VALUE, "block2_euler_run: the decomposition does not match the world size");
    block2::Plan plan;
    const long n[3] = {(long)g_local->shape[0], (long)g_local->shape[1], (long)g_local->shape[2]};
    if (block2::make_plan(n, dims3, coords3, cut3, &plan) != 0) return failf(E_VALUE, "block2_euler_run: the box is too small for two halo layers");
    HostOps2 ops;
    ops.c = c; ops.g_box = *g_local; ops.g_ext = *g_local; ops.faces = rhs->bc_c; ops.D = rhs->param; ops.dt = dt;
    ops.g_ext.shape[0] += 2; ops.g_ext.shape[1] += 2;
    ops.zcut = cut3[2] ? 1 : 0;
    if (ops.zcut) ops.g_ext.shape[2] += 2;
    SLAB_TRY(pdehip_layout(&ops.g_ext, ops.lay));
    ops.es = g_local->dtype == PDEHIP_F64 ? 8 : 4;
    int64_t ll[8];
    SLAB_TRY(pdehip_layout(g_local, ll));
    const size_t bytes = (size_t)ops.lay[2] * ops.es;
    for (auto &e : c->ext2) e.assign(bytes, 0);
    const size_t mbytes = (plan.send_total > plan.recv_total ? plan.send_total : plan.recv_total) * ops.es + 16;
    for (auto &m : c->msg2) m.assign(mbytes, 0);
    const long rowb = n[2] * (long)ops.es;
    auto own_rows = [&](void *ext, bool to_ext) {
        for (long i = 0; i < n[0]; i++)
            for (long j = 0; j < n[1]; j++) {
                char *e = static_cast<char *>(ext) + ops.elem_of(i, j, 0) * (long)ops.es;
                char *st = static_cast<char *>(buf_a) + ((long)ll[3] + i * (long)ll[0] + j * (long)ll[1]) * (long)ops.es;
                if (to_ext) memcpy(e, st, rowb); else memcpy(st, e, rowb);
            }
    };
    own_rows(c->ext2[0].data(), true);
    void *res = c->ext2[0].data();
    const int mode = getenv("PDEHIP_BLOCK2_MODE") ? atoi(getenv("PDEHIP_BLOCK2_MODE")) : 2;
    if (nsteps > 0) SLAB_TRY(block2::euler2_run(ops, plan, c->ext2[0].data(), c->ext2[1].data(), nsteps, &res, stream, mode));
    own_rows(res, false);
    *result = buf_a;
    return 0;
}

}  // extern "C"
