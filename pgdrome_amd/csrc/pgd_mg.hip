// Geometric multigrid preconditioner for pgd_pcg_solve (settings["preconditioner"] of the reference's solver parameters,
// solver.py:593-594 / 634-635 forwards them to PETSc; "amg"-type values map here, PGD_TUNE_PCG_PRECOND).
//
// Where it applies: the diagonally scaled operator of the solve is ONE stencil c[0..7] (dia_classify's stencil form: every row
// verified bit by bit) on an nx x ny x nz lattice whose eliminated (Dirichlet) nodes are exactly the hull of the lattice - cfg4 /
// cfg3 / cfg5: -Laplace + mu on the box with a homogeneous hull.  Everything else keeps the Jacobi-PCG.
//
// The hierarchy needs no matrix at all:
//   * P1 on the 6-tets-per-cube mesh is nested under doubling of the spacing (the cube's long diagonal is an edge of all six
//     tets), so the P1 interpolation P from the lattice of every other node has the operator's own 15-point shape - weight 1 at
//     the node, 1/2 at the 14 neighbours along the mesh edges - and the Galerkin operator P^T A P of a 15-point stencil is again
//     a 15-point stencil: eight numbers per level, computed on the host from the level above (mg_galerkin; entries off the
//     pattern are checked to vanish).
//   * node k of a coarse level is node 2k of the level above; a coarse node is eliminated iff that fine node is.  Where a lattice
//     has an even number of nodes its far hull face has no coarse counterpart: the last coarse node is free and what lies beyond
//     it reads as zero - a homogeneous Dirichlet node one coarse spacing out instead of half a spacing.  The preconditioner stays
//     symmetric positive definite; PCG does the rest (measured: 18 - 20 iterations for rtol 1e-10 from 32^3 to 256^3, even and
//     odd node counts alike).
//   * V(1,1) with damped Jacobi (omega = 6/7, lambda_max(D^-1 A) = 2 for these stencils), the pre-smoothing step from a zero
//     start folded into the residual (x1 = w b:  r = b - w A b) and into the prolongation (x = w b + P e); the coarsest level
//     (at most 4096 nodes) is 24 sweeps inside one workgroup.
// Invariant that keeps the kernels free of range logic: every vector of the cycle is ZERO on eliminated nodes, the faces x = 0,
// y = 0, z = 0 are eliminated on every level, so a neighbour index that runs over the end of a line or a plane lands on such a
// node (or behind the array: tested) and contributes the zero the ghost node would.
//
// Levels with at least mg_march_min nodes along x and y run their two stencil passes in k_spmv_stencil_march (epilogues 1 and 2,
// pgd_spmv.hip: 16 and 24 B per row at the product's rate); the small ones in the plain kernels below.
#include "pgd_internal.h"

#include <cmath>
#include <cstring>

namespace pgd {

struct MgGrid { int nx, ny, nz, fx, fy, fz; };         // f* = 1: the far face along that axis holds eliminated nodes (else: free, zero beyond)
struct MgSt { double c[8]; double w; };                // couplings (slot s = dx + 2 dy + 4 dz) and omega / c[0]

struct MgLevel {
    MgGrid g;
    int64_t n = 0;
    MgSt s;
    double *b = nullptr, *x = nullptr, *t = nullptr;    // right-hand side, result, work (level 0: b is the caller's r)
    uint8_t *cls = nullptr;                             // code byte per node for the march kernel: 1 = eliminated (levels >= 1; level 0: the operator's)
};

struct Mg {
    int nx = 0, ny = 0, nz = 0;
    std::vector<MgLevel> lv;
    double key[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    bool have_key = false;
    int *bad = nullptr;                                 // device flag of the hull check
    const uint8_t *cls0 = nullptr;                      // level 0: the operator's codes of THIS solve
    int ident0 = -1;
    int np0 = 0;                                        // partial sums the level-0 post-smoothing pass leaves
    // a z-slab of a row-sharded lattice (pgd_mg_slab_*): level 0 is the GLOBAL lattice of which this rank holds the local planes
    // [0, nzloc) = global planes [zoff, zoff + nzloc), owns [lz0, lz1) of them; level 0 has no buffers of its own
    bool slab = false;
    int zoff = 0, nzloc = 0, lz0 = 0, lz1 = 0;
    uint8_t *cls_slab = nullptr;                        // code byte per LOCAL node (1 = eliminated: the global hull) for the march kernel
    size_t cls_slab_bytes = 0;
    int cls_zoff = -1, cls_nzloc = -1;
};

struct MgSlab { int zoff, zl0, nzloc; };                // kernels: blockIdx.z = local plane - zl0; global z = local + zoff

__device__ __forceinline__ bool mg_is_free(const MgGrid &g, int x, int y, int z) {
    return x >= 1 && y >= 1 && z >= 1 && x <= g.nx - 1 - g.fx && y <= g.ny - 1 - g.fy && z <= g.nz - 1 - g.fz;
}

// (A v)_i for a free node i: lower neighbours exist (x, y, z >= 1), upper ones may run over the end of the array
__device__ __forceinline__ double mg_apply(const MgSt &S, const double *__restrict__ v, int64_t i, int64_t n, int nx, int64_t P) {
    double acc = S.c[0] * v[i];
#pragma unroll
    for (int s = 1; s < 8; ++s) {
        const int64_t off = (s & 1) + (int64_t)nx * ((s >> 1) & 1) + P * (s >> 2);
        const int64_t j = i + off;
        const double hi = j < n ? v[j] : 0.0;
        acc = fma(S.c[s], v[i - off] + hi, acc);
    }
    return acc;
}

// MODE 0: out = in - w A in            (residual behind the pre-smoothing step x1 = w in from a zero start)
// MODE 1: out = in + w (b - A in)      (post-smoothing step);  DOT: partial sums of b . out per workgroup
// W: the window of a z-slab (whole lattice: {0, 0, nz}) - the planes of blockIdx.z are local planes W.zl0 + blockIdx.z of an array of
// W.nzloc planes that starts at global plane W.zoff; `g` is the GLOBAL lattice (which nodes are free), neighbours are local.
template <int MODE, bool DOT>
__global__ __launch_bounds__(256) void k_mg_pass(MgGrid g, MgSt S, MgSlab W, const double *__restrict__ in, const double *__restrict__ b,
                                                 double *__restrict__ out, double *__restrict__ partials, const int *__restrict__ flags) {
    __shared__ double s_red[4];
    if (flags && flags[0]) return;
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6), zl = W.zl0 + (int)blockIdx.z;
    double d = 0.0;
    if (x < g.nx && y < g.ny) {
        const int64_t P = (int64_t)g.nx * g.ny, n = P * W.nzloc, i = P * zl + (int64_t)g.nx * y + x;
        double o = 0.0;
        if (mg_is_free(g, x, y, zl + W.zoff)) {
            const double a = mg_apply(S, in, i, n, g.nx, P);
            if (MODE == 0) o = fma(-S.w, a, in[i]);
            else { const double bi = b[i]; o = fma(S.w, bi - a, in[i]); if (DOT) d = bi * o; }
        }
        out[i] = o;
    }
    if (DOT) {
        d = block_sum(d, s_red);
        if (threadIdx.x == 0) partials[(int64_t)blockIdx.x + (int64_t)gridDim.x * (blockIdx.y + (int64_t)gridDim.y * blockIdx.z)] = d;
    }
}

// bc = P^T r: the node's own value + half of its 14 neighbours along the mesh edges of the fine lattice.
// Wf: the fine array's slab window (whole lattice: {0, 0, nz}); Z0: first coarse plane of the launch (blockIdx.z = Z - Z0); the coarse
// array is always whole.
__global__ __launch_bounds__(256) void k_mg_restrict(MgGrid gc, MgGrid gf, MgSlab Wf, int Z0, const double *__restrict__ r, double *__restrict__ bc,
                                                     const int *__restrict__ flags) {
    if (flags && flags[0]) return;
    const int X = blockIdx.x * 64 + (threadIdx.x & 63), Y = blockIdx.y * 4 + (threadIdx.x >> 6), Z = Z0 + (int)blockIdx.z;
    if (X >= gc.nx || Y >= gc.ny) return;
    const int64_t Pc = (int64_t)gc.nx * gc.ny, I = Pc * Z + (int64_t)gc.nx * Y + X;
    double o = 0.0;
    if (mg_is_free(gc, X, Y, Z)) {
        const int64_t Pf = (int64_t)gf.nx * gf.ny, nf = Pf * Wf.nzloc, i = Pf * (2 * Z - Wf.zoff) + (int64_t)gf.nx * (2 * Y) + 2 * X;
        double h = 0.0;
#pragma unroll
        for (int s = 1; s < 8; ++s) {
            const int64_t off = (s & 1) + (int64_t)gf.nx * ((s >> 1) & 1) + Pf * (s >> 2);
            const int64_t j = i + off;
            h += r[i - off] + (j < nf ? r[j] : 0.0);
        }
        o = fma(0.5, h, r[i]);
    }
    bc[I] = o;
}

// t = w b + P e on the fine lattice: a fine node is a coarse node (all coordinates even) or the midpoint of ONE coarse edge.
// Wf: the fine array's slab window (blockIdx.z = local plane - Wf.zl0); the coarse array is whole.
__global__ __launch_bounds__(256) void k_mg_prolong(MgGrid gf, MgGrid gc, MgSlab Wf, double w, const double *__restrict__ b, const double *__restrict__ e,
                                                    double *__restrict__ t, const int *__restrict__ flags) {
    if (flags && flags[0]) return;
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6), zl = Wf.zl0 + (int)blockIdx.z;
    if (x >= gf.nx || y >= gf.ny) return;
    const int z = zl + Wf.zoff;
    const int64_t Pf = (int64_t)gf.nx * gf.ny, i = Pf * zl + (int64_t)gf.nx * y + x;
    double o = 0.0;
    if (mg_is_free(gf, x, y, z)) {
        const int64_t Pc = (int64_t)gc.nx * gc.ny, nc = Pc * gc.nz;
        const int64_t ja = Pc * (z >> 1) + (int64_t)gc.nx * (y >> 1) + (x >> 1);
        const int64_t jb = Pc * ((z + 1) >> 1) + (int64_t)gc.nx * ((y + 1) >> 1) + ((x + 1) >> 1);
        const double ea = e[ja], eb = jb < nc ? e[jb] : 0.0;      // (ja == jb on a coarse node: 0.5 (e + e))
        o = fma(w, b[i], 0.5 * (ea + eb));
    }
    t[i] = o;
}

// the coarsest level inside one workgroup: `sweeps` damped-Jacobi steps from a zero start
constexpr int MG_BOTTOM_MAX = 4096;
__global__ __launch_bounds__(1024) void k_mg_bottom(MgGrid g, MgSt S, const double *__restrict__ b, double *__restrict__ x, int sweeps,
                                                    const int *__restrict__ flags) {
    __shared__ double s_v[2][MG_BOTTOM_MAX];
    if (flags && flags[0]) return;
    const int P = g.nx * g.ny, n = P * g.nz;
    double bi[4];
    bool fr[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int i = threadIdx.x + 1024 * k;
        fr[k] = false; bi[k] = 0.0;
        if (i < n) {
            const int z = i / P, rem = i - z * P, y = rem / g.nx, xx = rem - y * g.nx;
            fr[k] = mg_is_free(g, xx, y, z);
            bi[k] = fr[k] ? b[i] : 0.0;
            s_v[0][i] = S.w * bi[k];
        }
    }
    __syncthreads();
    int cur = 0;
    for (int sw = 1; sw < sweeps; ++sw) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int i = threadIdx.x + 1024 * k;
            if (i < n) s_v[cur ^ 1][i] = fr[k] ? fma(S.w, bi[k] - mg_apply(S, s_v[cur], i, n, g.nx, P), s_v[cur][i]) : 0.0;
        }
        __syncthreads();
        cur ^= 1;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int i = threadIdx.x + 1024 * k;
        if (i < n) x[i] = s_v[cur][i];
    }
}

// are the eliminated nodes of the operator exactly the hull of the lattice?
__global__ __launch_bounds__(TPB) void k_mg_hull_check(const uint8_t *__restrict__ cls, int ident, int nx, int ny, int nz, int *__restrict__ bad) {
    const int64_t P = (int64_t)nx * ny, n = P * nz;
    const int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x;
    if (i >= n) return;
    const int z = (int)(i / P), rem = (int)(i - (int64_t)z * P), y = rem / nx, x = rem - y * nx;
    const bool hull = x == 0 || y == 0 || z == 0 || x == nx - 1 || y == ny - 1 || z == nz - 1;
    if (((int)cls[i] == ident) != hull) *bad = 1;
}

__global__ __launch_bounds__(TPB) void k_mg_codes(MgGrid g, uint8_t *__restrict__ cls) {
    const int64_t P = (int64_t)g.nx * g.ny, n = P * g.nz;
    const int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x;
    if (i >= n) return;
    const int z = (int)(i / P), rem = (int)(i - (int64_t)z * P), y = rem / g.nx, x = rem - y * g.nx;
    cls[i] = mg_is_free(g, x, y, z) ? 0 : 1;
}

__global__ __launch_bounds__(TPB) void k_mg_codes_slab(MgGrid g, int zoff, int nzloc, uint8_t *__restrict__ cls) {
    const int64_t P = (int64_t)g.nx * g.ny, n = P * nzloc;
    const int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x;
    if (i >= n) return;
    const int zl = (int)(i / P), rem = (int)(i - (int64_t)zl * P), y = rem / g.nx, x = rem - y * g.nx;
    cls[i] = mg_is_free(g, x, y, zl + zoff) ? 0 : 1;
}

__global__ __launch_bounds__(TPB) void k_mg_fix_start(const uint8_t *__restrict__ cls, int ident, const double *__restrict__ b,
                                                      double *__restrict__ x, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x;
    if (i < n && (int)cls[i] == ident) x[i] = b[i];
}

// Galerkin coarse stencil P^T A P on the infinite lattice.  false: it leaves the 15-point pattern (not a P1 operator of this mesh)
static bool mg_galerkin(const double cf[8], double cc[8]) {
    double S[3][3][3], W[3][3][3];
    std::memset(S, 0, sizeof S);
    std::memset(W, 0, sizeof W);
    for (int s = 0; s < 8; ++s) {
        const int dx = s & 1, dy = (s >> 1) & 1, dz = s >> 2;
        S[1 + dz][1 + dy][1 + dx] = cf[s]; S[1 - dz][1 - dy][1 - dx] = cf[s];
        W[1 + dz][1 + dy][1 + dx] = s ? 0.5 : 1.0; W[1 - dz][1 - dy][1 - dx] = s ? 0.5 : 1.0;
    }
    // y = A (P e_0): support radius 2 around the fine node of coarse node 0
    double Y[5][5][5];
    std::memset(Y, 0, sizeof Y);
    for (int gz = -2; gz <= 2; ++gz) for (int gy = -2; gy <= 2; ++gy) for (int gx = -2; gx <= 2; ++gx) {
        double a = 0.0;
        for (int az = -1; az <= 1; ++az) for (int ay = -1; ay <= 1; ++ay) for (int ax = -1; ax <= 1; ++ax) {
            const int fz = gz + az, fy = gy + ay, fx = gx + ax;
            if (fz < -1 || fz > 1 || fy < -1 || fy > 1 || fx < -1 || fx > 1) continue;
            a += S[1 + az][1 + ay][1 + ax] * W[1 + fz][1 + fy][1 + fx];
        }
        Y[2 + gz][2 + gy][2 + gx] = a;
    }
    double scale = 0.0, leak = 0.0;
    for (int Dz = -1; Dz <= 1; ++Dz) for (int Dy = -1; Dy <= 1; ++Dy) for (int Dx = -1; Dx <= 1; ++Dx) {
        double a = 0.0;
        for (int ez = -1; ez <= 1; ++ez) for (int ey = -1; ey <= 1; ++ey) for (int ex = -1; ex <= 1; ++ex) {
            const int gz = 2 * Dz + ez, gy = 2 * Dy + ey, gx = 2 * Dx + ex;
            if (gz < -2 || gz > 2 || gy < -2 || gy > 2 || gx < -2 || gx > 2) continue;
            a += W[1 + ez][1 + ey][1 + ex] * Y[2 + gz][2 + gy][2 + gx];
        }
        // on the pattern: all components of D in {0, 1} or all in {0, -1}
        const bool up = Dx >= 0 && Dy >= 0 && Dz >= 0, dn = Dx <= 0 && Dy <= 0 && Dz <= 0;
        if (up) cc[Dx + 2 * Dy + 4 * Dz] = a;
        if (!up && !dn) leak = std::max(leak, std::fabs(a));
        scale = std::max(scale, std::fabs(a));
    }
    return cc[0] > 0.0 && leak <= 1e-10 * scale;
}

static void mg_free(Mg *&M) {
    if (!M) return;
    for (MgLevel &L : M->lv) {
        if (L.b) (void)hipFree(L.b);
        if (L.x) (void)hipFree(L.x);
        if (L.t) (void)hipFree(L.t);
        if (L.cls) (void)hipFree(L.cls);
    }
    if (M->bad) (void)hipFree(M->bad);
    if (M->cls_slab) (void)hipFree(M->cls_slab);
    delete M;
    M = nullptr;
}

void mg_release(Ctx *c) {
    mg_free(c->mg);
    mg_free(c->mg_slab);
}

double *mg_result(Ctx *c) { return c->mg && !c->mg->lv.empty() ? c->mg->lv[0].x : nullptr; }

static dim3 mg_grid(const MgGrid &g) { return dim3((unsigned)((g.nx + 63) / 64), (unsigned)((g.ny + 3) / 4), (unsigned)g.nz); }

bool mg_prepare(Ctx *c, const Mesh *m, const Csr *a) {
    if (!m || !a || !a->st_ok || a->st_ident < 0 || !a->cls || m->sym_nx <= 0) return false;
    const int nx = m->sym_nx, ny = m->sym_ny, nz = (int)(m->nv / ((int64_t)nx * ny));
    if ((int64_t)nx * ny * nz != m->nv || std::min(nx, std::min(ny, nz)) < 8 || nz > 65535 || ny > 4 * 65535) return false;
    if (a->st_z0 != 0 || a->st_z1 != nz) return false;                     // (the whole lattice was verified, not a slab of it)
    for (int s = 1; s < 8; ++s) if (!(a->st_c[s] == a->st_c[s])) return false;
    if (!(a->st_c[0] > 0.0)) return false;
    if (!c->mg) c->mg = new Mg();
    Mg *M = c->mg;
    if (!M->bad) {
        void *q = nullptr;
        if (hipMalloc(&q, sizeof(int)) != hipSuccess) return false;
        M->bad = (int *)q;
    }
    // the eliminated nodes must be the hull, nothing else (one pass over the code bytes; read back with the solve's first look)
    int bad = 0;
    if (hipMemsetAsync(M->bad, 0, sizeof(int), c->stream) != hipSuccess) return false;
    k_mg_hull_check<<<(unsigned)((m->nv + TPB - 1) / TPB), TPB, 0, c->stream>>>(a->cls, a->st_ident, nx, ny, nz, M->bad);
    if (hipMemcpyAsync(&bad, M->bad, sizeof(int), hipMemcpyDeviceToHost, c->stream) != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess)
        return false;
    if (bad) return false;
    if (M->nx != nx || M->ny != ny || M->nz != nz) {                       // another lattice: new levels and buffers
        mg_free(c->mg);
        c->mg = new Mg();
        M = c->mg;
        void *q = nullptr;
        if (hipMalloc(&q, sizeof(int)) != hipSuccess) return false;
        M->bad = (int *)q;
        M->nx = nx; M->ny = ny; M->nz = nz;
        MgGrid g{nx, ny, nz, 1, 1, 1};
        for (;;) {
            MgLevel L;
            L.g = g;
            L.n = (int64_t)g.nx * g.ny * g.nz;
            M->lv.push_back(L);
            if (std::min(g.nx, std::min(g.ny, g.nz)) < 8) break;
            MgGrid h;
            h.nx = (g.nx + 1) / 2; h.ny = (g.ny + 1) / 2; h.nz = (g.nz + 1) / 2;
            h.fx = g.fx && (g.nx & 1); h.fy = g.fy && (g.ny & 1); h.fz = g.fz && (g.nz & 1);
            g = h;
        }
        if (M->lv.size() < 2 || M->lv.back().n > MG_BOTTOM_MAX) { M->nx = 0; M->lv.clear(); return false; }
        for (size_t l = 0; l < M->lv.size(); ++l) {
            MgLevel &L = M->lv[l];
            void *q2 = nullptr;
            const size_t bytes = (size_t)L.n * sizeof(double);
            if (l > 0) { if (hipMalloc(&q2, bytes) != hipSuccess) { M->nx = 0; return false; } L.b = (double *)q2; }
            if (hipMalloc(&q2, bytes) != hipSuccess) { M->nx = 0; return false; }
            L.x = (double *)q2;
            if (l + 1 < M->lv.size()) { if (hipMalloc(&q2, bytes) != hipSuccess) { M->nx = 0; return false; } L.t = (double *)q2; }
            if (l > 0 && l + 1 < M->lv.size()) {
                if (hipMalloc(&q2, (size_t)L.n) != hipSuccess) { M->nx = 0; return false; }
                L.cls = (uint8_t *)q2;
                k_mg_codes<<<(unsigned)((L.n + TPB - 1) / TPB), TPB, 0, c->stream>>>(L.g, L.cls);
            }
        }
        M->have_key = false;
    }
    if (!M->have_key || std::memcmp(M->key, a->st_c, sizeof M->key) != 0) {
        double cf[8];
        for (int s = 0; s < 8; ++s) cf[s] = a->st_c[s];
        for (size_t l = 0; l < M->lv.size(); ++l) {
            MgLevel &L = M->lv[l];
            for (int s = 0; s < 8; ++s) L.s.c[s] = cf[s];
            L.s.w = (6.0 / 7.0) / cf[0];      // (prototype scan at 64^3, V(1,1)-PCG iterations: 0.7 -> 20, 0.8 -> 19, 6/7 -> 18, 0.9 -> 18, 0.95 -> 24, 1 -> 100)
            if (l + 1 < M->lv.size()) {
                double cc[8];
                if (!mg_galerkin(cf, cc)) { M->have_key = false; return false; }
                for (int s = 0; s < 8; ++s) cf[s] = cc[s];
            }
        }
        std::memcpy(M->key, a->st_c, sizeof M->key);
        M->have_key = true;
    }
    M->cls0 = a->cls;
    M->ident0 = a->st_ident;
    const dim3 g0 = mg_grid(M->lv[0].g);
    M->np0 = (int)((int64_t)g0.x * g0.y * g0.z);
    if (ensure_partials(c, std::max<int64_t>(M->np0 + 64, 4 * (int64_t)MAX_VEC_BLOCKS)) != PGD_OK) return false;
    return true;
}

int mg_fix_start(Ctx *c, const Csr *a, const double *b, double *x, int64_t n) {
    k_mg_fix_start<<<(unsigned)((n + TPB - 1) / TPB), TPB, 0, c->stream>>>(a->cls, a->st_ident, b, x, n);
    PGD_LAUNCH_CHECK(c);
    return PGD_OK;
}

// The cycle from level l0 down and back up: b0 is level l0's right-hand side, the result lands in z_out (default: that level's x).
// (dot / nparts: only with l0 == 0, the level whose post-smoothing pass the PCG wants r . z from.)
static int mg_cycle(Ctx *c, Mg *M, int l0, const double *b0, bool dot, int *nparts, double *z_out) {
    if (!M || (int)M->lv.size() < l0 + 1) return fail(c, PGD_ERR_INVALID, "mg_cycle: no hierarchy");
    const int nl = (int)M->lv.size();
    const dim3 blk(256, 1, 1);      // 64 x 4 nodes of one plane
    const int *flags = c->flags;
    // a level's stencil passes run in the march kernel of the product where the lattice is wide enough to fill its 64 x 16 patches
    auto march = [&](int l) {
        const MgLevel &L = M->lv[l];
        return c->mg_march_min > 0 && L.g.nx >= c->mg_march_min && L.g.ny >= c->mg_march_min && L.g.nz >= 8 &&
               (int64_t)L.g.nx * L.g.ny < ((int64_t)1 << 26) && ((l == 0 && !M->slab) || L.cls);
    };
    auto pass = [&](int l, int epi, const double *in, const double *b, double *out, bool want_dot, int *np) -> int {
        const MgLevel &L = M->lv[l];
        return launch_stencil_pass(c, l == 0 ? M->cls0 : L.cls, l == 0 ? M->ident0 : 1, L.s.c, L.g.nx, L.g.ny, L.g.nz, 1, L.g.nz - L.g.fz,
                                   in, b, out, L.s.w, epi, want_dot, np);
    };
    auto whole = [&](const MgLevel &L) { return MgSlab{0, 0, L.g.nz}; };
    int np_dot = M->np0;
    if (l0 == nl - 1) {                                   // (a hierarchy whose first replicated level is the coarsest)
        MgLevel &B = M->lv[nl - 1];
        k_mg_bottom<<<1, 1024, 0, c->stream>>>(B.g, B.s, b0, z_out ? z_out : B.x, 24, flags);
        PGD_LAUNCH_CHECK(c);
        return PGD_OK;
    }
    // down: residual behind the folded pre-smoothing step, restriction
    for (int l = l0; l + 1 < nl; ++l) {
        MgLevel &L = M->lv[l], &C = M->lv[l + 1];
        const double *b = l == l0 ? b0 : L.b;
        if (march(l)) PGD_TRY(pass(l, 1, b, nullptr, L.t, false, nullptr));
        else k_mg_pass<0, false><<<mg_grid(L.g), blk, 0, c->stream>>>(L.g, L.s, whole(L), b, nullptr, L.t, nullptr, flags);
        k_mg_restrict<<<mg_grid(C.g), blk, 0, c->stream>>>(C.g, L.g, whole(L), 0, L.t, C.b, flags);
    }
    {
        MgLevel &B = M->lv[nl - 1];
        k_mg_bottom<<<1, 1024, 0, c->stream>>>(B.g, B.s, B.b, B.x, 24, flags);
    }
    // up: x = w b + P e, one more damped-Jacobi step
    for (int l = nl - 2; l >= l0; --l) {
        MgLevel &L = M->lv[l], &C = M->lv[l + 1];
        const double *b = l == l0 ? b0 : L.b;
        k_mg_prolong<<<mg_grid(L.g), blk, 0, c->stream>>>(L.g, C.g, whole(L), L.s.w, b, C.x, L.t, flags);
        double *out = l == l0 && z_out ? z_out : L.x;
        const bool d = l == 0 && dot;
        if (march(l)) PGD_TRY(pass(l, 2, L.t, b, out, d, d ? &np_dot : nullptr));
        else if (d) k_mg_pass<1, true><<<mg_grid(L.g), blk, 0, c->stream>>>(L.g, L.s, whole(L), L.t, b, out, c->partials, flags);
        else k_mg_pass<1, false><<<mg_grid(L.g), blk, 0, c->stream>>>(L.g, L.s, whole(L), L.t, b, out, nullptr, flags);
    }
    if (nparts) *nparts = np_dot;
    PGD_LAUNCH_CHECK(c);
    return PGD_OK;
}

int mg_vcycle(Ctx *c, const double *r, bool dot, int *nparts, double *z_out) {
    Mg *M = c->mg;
    if (!M || M->lv.size() < 2) return fail(c, PGD_ERR_INVALID, "mg_vcycle: no hierarchy");
    return mg_cycle(c, M, 0, r, dot, nparts, z_out);
}

// ------------------------------------------------------------------------------------------------------------------------------
// The V-cycle on a ROW-SHARDED lattice (round 4; VERDICT r03 "missing 2": settings["preconditioner"] = "amg" on a sharded mesh).
// Level 0 - the lattice of the solve, 7/8 of a cycle's work - stays where its rows are: every rank runs the level's two stencil
// passes, the restriction and the prolongation on its own z-slab (one ghost plane per side, filled by the caller's halo exchange).
// Levels >= 1 are WHOLE on every rank: each rank restricts into the coarse planes its slab covers (zero elsewhere), the caller
// sums that vector over the ranks (one all-reduce of n / 8 doubles - every entry has exactly one non-zero contribution, so the sum
// is exact), and every rank runs the rest of the cycle redundantly with the kernels of the unsharded preconditioner.  Every value
// of the cycle is then computed by the same arithmetic as on one GPU; per cycle the ranks exchange two halos (r, t) and one
// coarse vector.  The PCG loop around it, its dots and its collectives belong to the caller (pgdrome_amd/dist.py::pcg_mg).
// Applies where A itself (unscaled) is one stencil + eliminated nodes on the rank's owned planes (dia_classify, every row verified)
// and the eliminated nodes of those planes are exactly the hull of the GLOBAL lattice.

__global__ __launch_bounds__(TPB) void k_mg_hull_check_slab(const uint8_t *__restrict__ cls, int ident, int nx, int ny, int nz_global, int zoff,
                                                            int64_t row0, int64_t row1, int *__restrict__ bad) {
    const int64_t P = (int64_t)nx * ny;
    const int64_t i = row0 + (int64_t)blockIdx.x * TPB + threadIdx.x;
    if (i >= row1) return;
    const int zl = (int)(i / P), rem = (int)(i - (int64_t)zl * P), y = rem / nx, x = rem - y * nx, z = zl + zoff;
    const bool hull = x == 0 || y == 0 || z == 0 || x == nx - 1 || y == ny - 1 || z == nz_global - 1;
    if (((int)cls[i] == ident) != hull) *bad = 1;
}

static void mg_free_levels(Mg *M) {
    for (MgLevel &L : M->lv) {
        if (L.b) (void)hipFree(L.b);
        if (L.x) (void)hipFree(L.x);
        if (L.t) (void)hipFree(L.t);
        if (L.cls) (void)hipFree(L.cls);
    }
    M->lv.clear();
}

}  // namespace pgd

using namespace pgd;

extern "C" {

int pgd_mg_slab_setup(pgd_handle h, pgd_handle oh, int nz_global, int z_first, int64_t own0, int64_t own1, int64_t *n_coarse, int *applies) {
    PGD_CTX(c, h);
    Csr *a = get_csr(c, oh);
    Mesh *m = a ? get_mesh(c, a->mesh) : nullptr;
    if (!a || !m || !n_coarse || !applies) return fail(c, PGD_ERR_INVALID, "mg_slab_setup: invalid arguments");
    *applies = 0; *n_coarse = 0;
    if (m->sym_nx <= 0) return PGD_OK;
    const int nx = m->sym_nx, ny = m->sym_ny;
    const int64_t P = (int64_t)nx * ny;
    const int nzloc = (int)(m->nv / P);
    if ((int64_t)nzloc * P != m->nv || own0 % P != 0 || own1 % P != 0 || own0 < 0 || own1 <= own0 || own1 > m->nv || z_first < 0 ||
        z_first + nzloc > nz_global)
        return fail(c, PGD_ERR_INVALID, "mg_slab_setup: the owned rows are not whole planes of the local lattice, or the slab leaves the global one");
    if (std::min(nx, std::min(ny, nz_global)) < 8 || nz_global > 65535 || ny > 4 * 65535) return PGD_OK;
    const int lz0 = (int)(own0 / P), lz1 = (int)(own1 / P);
    // every owned plane but the global hull's needs its neighbours in the local array (one ghost plane per side)
    if ((z_first + lz0 > 0 && lz0 < 1) || (z_first + lz1 < nz_global && lz1 + 1 > nzloc)) return PGD_OK;
    bool sym = false;
    PGD_TRY(ensure_sym(c, m, a, &sym));
    if (!sym || !a->uvals_valid || a->uvals_scaled) return PGD_OK;
    PGD_TRY(dia_classify(c, m, a, lz0, lz1));
    if (!a->st_ok || a->st_ident < 0 || !a->cls || a->st_z0 != lz0 || a->st_z1 != lz1) return PGD_OK;
    for (int s = 1; s < 8; ++s) if (!(a->st_c[s] == a->st_c[s])) return PGD_OK;
    if (!(a->st_c[0] > 0.0)) return PGD_OK;
    if (!c->mg_slab) c->mg_slab = new Mg();
    Mg *M = c->mg_slab;
    M->slab = true;
    if (!M->bad) {
        void *q = nullptr;
        PGD_HIP(c, hipMalloc(&q, sizeof(int)));
        M->bad = (int *)q;
    }
    int bad = 0;
    PGD_HIP(c, hipMemsetAsync(M->bad, 0, sizeof(int), c->stream));
    k_mg_hull_check_slab<<<(unsigned)((own1 - own0 + TPB - 1) / TPB), TPB, 0, c->stream>>>(a->cls, a->st_ident, nx, ny, nz_global, z_first, own0, own1, M->bad);
    PGD_HIP(c, hipMemcpyAsync(&bad, M->bad, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    PGD_HIP(c, hipStreamSynchronize(c->stream));
    if (bad) return PGD_OK;
    if (M->nx != nx || M->ny != ny || M->nz != nz_global) {
        mg_free_levels(M);
        M->nx = M->ny = M->nz = 0;
        MgGrid g{nx, ny, nz_global, 1, 1, 1};
        for (;;) {
            MgLevel L;
            L.g = g;
            L.n = (int64_t)g.nx * g.ny * g.nz;
            M->lv.push_back(L);
            if (std::min(g.nx, std::min(g.ny, g.nz)) < 8) break;
            MgGrid hh;
            hh.nx = (g.nx + 1) / 2; hh.ny = (g.ny + 1) / 2; hh.nz = (g.nz + 1) / 2;
            hh.fx = g.fx && (g.nx & 1); hh.fy = g.fy && (g.ny & 1); hh.fz = g.fz && (g.nz & 1);
            g = hh;
        }
        if (M->lv.size() < 2 || M->lv.back().n > MG_BOTTOM_MAX) { M->lv.clear(); return PGD_OK; }
        for (size_t l = 1; l < M->lv.size(); ++l) {          // (level 0 lives in the caller's slab vectors)
            MgLevel &L = M->lv[l];
            void *q2 = nullptr;
            const size_t bytes = (size_t)L.n * sizeof(double);
            PGD_HIP(c, hipMalloc(&q2, bytes)); L.b = (double *)q2;
            PGD_HIP(c, hipMalloc(&q2, bytes)); L.x = (double *)q2;
            if (l + 1 < M->lv.size()) {
                PGD_HIP(c, hipMalloc(&q2, bytes)); L.t = (double *)q2;
                PGD_HIP(c, hipMalloc(&q2, (size_t)L.n)); L.cls = (uint8_t *)q2;
                k_mg_codes<<<(unsigned)((L.n + TPB - 1) / TPB), TPB, 0, c->stream>>>(L.g, L.cls);
            }
        }
        PGD_LAUNCH_CHECK(c);
        M->nx = nx; M->ny = ny; M->nz = nz_global;
        M->have_key = false;
    }
    if (!M->have_key || std::memcmp(M->key, a->st_c, sizeof M->key) != 0) {
        double cf[8];
        for (int s = 0; s < 8; ++s) cf[s] = a->st_c[s];
        for (size_t l = 0; l < M->lv.size(); ++l) {
            MgLevel &L = M->lv[l];
            for (int s = 0; s < 8; ++s) L.s.c[s] = cf[s];
            L.s.w = (6.0 / 7.0) / cf[0];
            if (l + 1 < M->lv.size()) {
                double cc[8];
                if (!mg_galerkin(cf, cc)) { M->have_key = false; return PGD_OK; }
                for (int s = 0; s < 8; ++s) cf[s] = cc[s];
            }
        }
        std::memcpy(M->key, a->st_c, sizeof M->key);
        M->have_key = true;
    }
    M->zoff = z_first; M->nzloc = nzloc; M->lz0 = lz0; M->lz1 = lz1;
    if (M->cls_zoff != z_first || M->cls_nzloc != nzloc || M->cls_slab_bytes < (size_t)m->nv) {
        if (M->cls_slab && M->cls_slab_bytes < (size_t)m->nv) { (void)hipFree(M->cls_slab); M->cls_slab = nullptr; M->cls_slab_bytes = 0; }
        if (!M->cls_slab) {
            void *q = nullptr;
            PGD_HIP(c, hipMalloc(&q, (size_t)m->nv + PAD_BYTES));
            M->cls_slab = (uint8_t *)q; M->cls_slab_bytes = (size_t)m->nv;
        }
        k_mg_codes_slab<<<(unsigned)((m->nv + TPB - 1) / TPB), TPB, 0, c->stream>>>(M->lv[0].g, z_first, nzloc, M->cls_slab);
        PGD_LAUNCH_CHECK(c);
        M->cls_zoff = z_first; M->cls_nzloc = nzloc;
    }
    const int64_t np = (int64_t)((nx + 63) / 64) * ((ny + 3) / 4) * (lz1 - lz0);
    PGD_TRY(ensure_partials(c, std::max<int64_t>(np + 64, 4 * (int64_t)MAX_VEC_BLOCKS)));
    *n_coarse = M->lv[1].n;
    *applies = 1;
    return PGD_OK;
}

// level 0 of a slab in the march kernel of the product: wide planes, at least three owned planes; the main run = the local planes
// that are not hull planes of the global lattice (ghost planes included: they hold the neighbour ranks' data)
static bool mg_slab_march(const Ctx *c, const Mg *M, int *zm0, int *zm1) {
    const MgGrid &g = M->lv[0].g;
    *zm0 = std::max(0, 1 - M->zoff);
    *zm1 = std::min(M->nzloc, g.nz - 1 - M->zoff);
    return c->mg_march_min > 0 && g.nx >= c->mg_march_min && g.ny >= c->mg_march_min && M->lz1 - M->lz0 >= 3 && M->cls_slab &&
           (int64_t)g.nx * g.ny < ((int64_t)1 << 26) && *zm1 > *zm0;
}

#define PGD_MG_SLAB(M, what)                                                                                           \
    Mg *M = c->mg_slab;                                                                                                \
    if (!M || !M->slab || M->lv.size() < 2 || !M->have_key) return fail(c, PGD_ERR_INVALID, what ": pgd_mg_slab_setup has not succeeded")

int pgd_mg_slab_fix_start(pgd_handle h, pgd_handle oh, pgd_handle bh, pgd_handle xh, int64_t own0, int64_t own1) {
    PGD_CTX(c, h);
    PGD_MG_SLAB(M, "mg_slab_fix_start");
    Csr *a = get_csr(c, oh);
    Vec *b = get_vec(c, bh), *x = get_vec(c, xh);
    if (!a || !a->cls || !a->st_ok || !b || !x || b->n != x->n || own0 < 0 || own1 > x->n || own0 > own1)
        return fail(c, PGD_ERR_INVALID, "mg_slab_fix_start: invalid arguments");
    if (own1 == own0) return PGD_OK;
    // x = b on the eliminated rows of the owned planes: the residual, and with it every vector of the cycle, vanishes there
    k_mg_fix_start<<<(unsigned)((own1 - own0 + TPB - 1) / TPB), TPB, 0, c->stream>>>(a->cls + own0, a->st_ident, b->d + own0, x->d + own0, own1 - own0);
    PGD_LAUNCH_CHECK(c);
    return PGD_OK;
}

int pgd_mg_slab_down(pgd_handle h, pgd_handle rh, pgd_handle th) {
    PGD_CTX(c, h);
    PGD_MG_SLAB(M, "mg_slab_down");
    Vec *r = get_vec(c, rh), *t = get_vec(c, th);
    const MgLevel &L = M->lv[0];
    const int64_t nloc = (int64_t)L.g.nx * L.g.ny * M->nzloc;
    if (!r || !t || r == t || r->n != nloc || t->n != nloc) return fail(c, PGD_ERR_INVALID, "mg_slab_down: vectors of the local slab expected");
    int zm0 = 0, zm1 = 0;
    if (mg_slab_march(c, M, &zm0, &zm1)) {
        // (the kernels of the cycle leave at once where the context's "done" flag is up: the caller's loop - dist.pcg_mg - resets the
        // flags at the start of a solve and stops calling when the flag rises)
        return launch_stencil_pass(c, M->cls_slab, 1, L.s.c, L.g.nx, L.g.ny, M->nzloc, zm0, zm1, r->d, nullptr, t->d, L.s.w, 1, false, nullptr, M->lz0, M->lz1);
    }
    const dim3 grid((unsigned)((L.g.nx + 63) / 64), (unsigned)((L.g.ny + 3) / 4), (unsigned)(M->lz1 - M->lz0));
    k_mg_pass<0, false><<<grid, dim3(256, 1, 1), 0, c->stream>>>(L.g, L.s, MgSlab{M->zoff, M->lz0, M->nzloc}, r->d, nullptr, t->d, nullptr, nullptr);
    PGD_LAUNCH_CHECK(c);
    return PGD_OK;
}

int pgd_mg_slab_restrict(pgd_handle h, pgd_handle th, pgd_handle bh) {
    PGD_CTX(c, h);
    PGD_MG_SLAB(M, "mg_slab_restrict");
    Vec *t = get_vec(c, th), *b1 = get_vec(c, bh);
    const MgLevel &L = M->lv[0], &C = M->lv[1];
    const int64_t nloc = (int64_t)L.g.nx * L.g.ny * M->nzloc;
    if (!t || !b1 || t->n != nloc || b1->n != C.n) return fail(c, PGD_ERR_INVALID, "mg_slab_restrict: slab vector and whole level-1 vector expected");
    PGD_HIP(c, hipMemsetAsync(b1->d, 0, (size_t)C.n * sizeof(double), c->stream));
    // coarse plane Z belongs to the rank that owns fine plane 2 Z
    const int g0 = M->zoff + M->lz0, g1 = M->zoff + M->lz1, Z0 = (g0 + 1) / 2, Z1 = (g1 + 1) / 2;
    if (Z1 > Z0) {
        const dim3 grid((unsigned)((C.g.nx + 63) / 64), (unsigned)((C.g.ny + 3) / 4), (unsigned)(Z1 - Z0));
        k_mg_restrict<<<grid, dim3(256, 1, 1), 0, c->stream>>>(C.g, L.g, MgSlab{M->zoff, 0, M->nzloc}, Z0, t->d, b1->d, nullptr);
        PGD_LAUNCH_CHECK(c);
    }
    return PGD_OK;
}

int pgd_mg_coarse(pgd_handle h, pgd_handle bh, pgd_handle xh) {
    PGD_CTX(c, h);
    PGD_MG_SLAB(M, "mg_coarse");
    Vec *b1 = get_vec(c, bh), *x1 = get_vec(c, xh);
    if (!b1 || !x1 || b1 == x1 || b1->n != M->lv[1].n || x1->n != M->lv[1].n) return fail(c, PGD_ERR_INVALID, "mg_coarse: whole level-1 vectors expected");
    return mg_cycle(c, M, 1, b1->d, false, nullptr, x1->d);
}

int pgd_mg_slab_up(pgd_handle h, pgd_handle rh, pgd_handle xh, pgd_handle th, pgd_handle zh, int slot, double *dot) {
    PGD_CTX(c, h);
    PGD_MG_SLAB(M, "mg_slab_up");
    Vec *r = get_vec(c, rh), *x1 = get_vec(c, xh), *t = get_vec(c, th), *z = get_vec(c, zh);
    const MgLevel &L = M->lv[0], &C = M->lv[1];
    const int64_t nloc = (int64_t)L.g.nx * L.g.ny * M->nzloc;
    if (!r || !x1 || !t || !z || (!dot && slot < 0) || slot >= PGD_NSLOTS || r->n != nloc || t->n != nloc || z->n != nloc || x1->n != C.n ||
        t == r || z == r || z == t)
        return fail(c, PGD_ERR_INVALID, "mg_slab_up: invalid vectors or slot");
    const dim3 blk(256, 1, 1);
    // t = w r + P e on ALL local planes (the ghost planes of r are current and e is whole: no exchange of t is needed)
    const dim3 gall((unsigned)((L.g.nx + 63) / 64), (unsigned)((L.g.ny + 3) / 4), (unsigned)M->nzloc);
    k_mg_prolong<<<gall, blk, 0, c->stream>>>(L.g, C.g, MgSlab{M->zoff, 0, M->nzloc}, L.s.w, r->d, x1->d, t->d, nullptr);
    // z = t + w (r - A t) on the owned planes, partial sums of r . z
    const dim3 gown(gall.x, gall.y, (unsigned)(M->lz1 - M->lz0));
    int np = (int)((int64_t)gown.x * gown.y * gown.z);
    int zm0 = 0, zm1 = 0;
    const double *parts = c->partials;
    if (mg_slab_march(c, M, &zm0, &zm1)) {
        PGD_LAUNCH_CHECK(c);
        PGD_TRY(launch_stencil_pass(c, M->cls_slab, 1, L.s.c, L.g.nx, L.g.ny, M->nzloc, zm0, zm1, t->d, r->d, z->d, L.s.w, 2, true, &np, M->lz0, M->lz1));
        parts = c->partials + c->partials_off;
    } else {
        PGD_TRY(ensure_partials(c, std::max<int64_t>(np + 64, 4 * (int64_t)MAX_VEC_BLOCKS)));
        parts = c->partials;
        k_mg_pass<1, true><<<gown, blk, 0, c->stream>>>(L.g, L.s, MgSlab{M->zoff, M->lz0, M->nzloc}, t->d, r->d, z->d, c->partials, nullptr);
        PGD_LAUNCH_CHECK(c);
    }
    if (slot >= 0) return reduce_partials(c, parts, np, 1, slot, -1, 0, 0);      // r . z stays on the device (a host-driven loop on the slot bank)
    PGD_TRY(ensure_work(c, 6, 256));
    PGD_TRY(reduce_partials_to(c, parts, np, 1, c->work[6]));
    PGD_HIP(c, hipMemcpyAsync(dot, c->work[6], sizeof(double), hipMemcpyDeviceToHost, c->stream));
    PGD_HIP(c, hipStreamSynchronize(c->stream));
    return PGD_OK;
}

}  // extern "C"

