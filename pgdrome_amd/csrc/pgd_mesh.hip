// Mesh upload, vertex->cell adjacency, CSR pattern and P1 element assembly.
//
// Assembly is OWNER-COMPUTES: the workgroup that owns 256 consecutive rows
// gathers, per row, the element-matrix rows of all incident cells in a fixed
// (sorted) order and accumulates them in an LDS image of the rows' contiguous
// CSR segment, which is then written out with coalesced stores.  No atomics,
// bitwise reproducible, and the nodal coordinates are read from SoA arrays so
// that neighbouring rows read neighbouring addresses.  This is the LDS-staged
// scatter into CSR of the north star, turned into a gather.
#include "pgd_internal.h"

#include <cstring>

namespace pgd {

constexpr int MAX_ROW = 64;         // max row length for int4 cell records (P1 simplices, P2 intervals)
constexpr int MAX_ROW_P2 = 128;     // ... for P2 triangles / tetrahedra
constexpr int ASM_CAP = 6144;       // CSR entries staged per 256-row workgroup (72 KiB LDS)

// ------------------------------------------------------------------ adjacency
// Cell records: one int4 per cell for up to 4 nodes (P1 simplices, P2 intervals) - a single 16-byte load -
// or flat records of nvpc ints for P2 triangles (6 nodes) / tetrahedra (10 nodes).
template <int NVMAX>
__device__ __forceinline__ void load_cell(const void *__restrict__ cells, int64_t e, int nvpc, int *u) {
    if constexpr (NVMAX <= 4) {
        const int4 c = ((const int4 *)cells)[e];
        u[0] = c.x; u[1] = c.y; u[2] = c.z; u[3] = c.w;
    } else {
        const int *rec = (const int *)cells + e * nvpc;
#pragma unroll
        for (int j = 0; j < NVMAX; ++j) u[j] = j < nvpc ? rec[j] : -1;
    }
}

template <int NVMAX>
__global__ __launch_bounds__(TPB) void k_v2c_count(const void *__restrict__ cells, int64_t nc, int nvpc,
                                                   int *__restrict__ cnt) {
    for (int64_t e = (int64_t)blockIdx.x * TPB + threadIdx.x; e < nc; e += (int64_t)gridDim.x * TPB) {
        int v[NVMAX];
        load_cell<NVMAX>(cells, e, nvpc, v);
#pragma unroll
        for (int j = 0; j < NVMAX; ++j) if (j < nvpc) atomicAdd(&cnt[v[j]], 1);
    }
}

template <int NVMAX>
__global__ __launch_bounds__(TPB) void k_v2c_fill(const void *__restrict__ cells, int64_t nc, int nvpc,
                                                  const int *__restrict__ ptr, int *__restrict__ cursor,
                                                  int *__restrict__ v2c) {
    for (int64_t e = (int64_t)blockIdx.x * TPB + threadIdx.x; e < nc; e += (int64_t)gridDim.x * TPB) {
        int v[NVMAX];
        load_cell<NVMAX>(cells, e, nvpc, v);
#pragma unroll
        for (int j = 0; j < NVMAX; ++j) {
            if (j < nvpc) {
                const int pos = atomicAdd(&cursor[v[j]], 1);
                v2c[ptr[v[j]] + pos] = (int)e;
            }
        }
    }
}

// atomics filled the lists in arrival order: sort each (short) list so that every
// later pass sees the incident cells in ascending cell id - fixed summation order.
__global__ __launch_bounds__(TPB) void k_v2c_sort(const int *__restrict__ ptr, int *__restrict__ v2c, int64_t nv) {
    for (int64_t v = (int64_t)blockIdx.x * TPB + threadIdx.x; v < nv; v += (int64_t)gridDim.x * TPB) {
        const int a = ptr[v], b = ptr[v + 1];
        for (int i = a + 1; i < b; ++i) {
            const int key = v2c[i];
            int j = i - 1;
            while (j >= a && v2c[j] > key) { v2c[j + 1] = v2c[j]; --j; }
            v2c[j + 1] = key;
        }
    }
}

// --------------------------------------------------------------------- pattern
// Sorted-unique neighbour list of one node into s_row (LDS, one row per thread,
// stride MAXR+1 to keep the banks apart).  Returns the length, or -1 on overflow.
template <int NVMAX, int MAXR>
__device__ __forceinline__ int gather_row(int v, const void *__restrict__ cells, int nvpc,
                                          const int *__restrict__ v2c_ptr, const int *__restrict__ v2c,
                                          int *s_row) {
    int len = 0;
    const int a = v2c_ptr[v], b = v2c_ptr[v + 1];
    for (int k = a; k < b; ++k) {
        int u[NVMAX];
        load_cell<NVMAX>(cells, v2c[k], nvpc, u);
#pragma unroll
        for (int j = 0; j < NVMAX; ++j) {
            if (j >= nvpc) continue;
            const int w = u[j];
            int pos = 0;
            while (pos < len && s_row[pos] < w) ++pos;
            if (pos < len && s_row[pos] == w) continue;
            if (len >= MAXR) return -1;
            for (int t = len; t > pos; --t) s_row[t] = s_row[t - 1];
            s_row[pos] = w;
            ++len;
        }
    }
    return len;
}

// NT threads per workgroup, each builds one row in its own LDS strip: <4, 64, 256> for int4 records,
// <10, 128, 64> for P2 simplices (a P2 tetrahedron vertex on a structured box couples to 65 nodes).
template <bool FILL, int NVMAX, int MAXR, int NT>
__global__ __launch_bounds__(NT) void k_pattern(const void *__restrict__ cells, int nvpc,
                                                const int *__restrict__ v2c_ptr, const int *__restrict__ v2c,
                                                int64_t nv, int *__restrict__ row_len,
                                                const int *__restrict__ row_ptr, int *__restrict__ cols,
                                                int *__restrict__ stats /* [0] overflow, [1] max_row, [2] kl, [3] ku */) {
    __shared__ int s_rows[NT * (MAXR + 1)];
    const int64_t v = (int64_t)blockIdx.x * NT + threadIdx.x;
    if (v >= nv) return;
    int *s_row = s_rows + threadIdx.x * (MAXR + 1);
    const int len = gather_row<NVMAX, MAXR>((int)v, cells, nvpc, v2c_ptr, v2c, s_row);
    if (len < 0) { atomicExch(&stats[0], 1); if (!FILL) row_len[v] = 0; return; }
    if (!FILL) {
        row_len[v] = len;
        atomicMax(&stats[1], len);
        if (len > 0) {
            atomicMax(&stats[2], (int)v - s_row[0]);
            atomicMax(&stats[3], s_row[len - 1] - (int)v);
        }
    } else {
        const int base = row_ptr[v];
        for (int k = 0; k < len; ++k) cols[base + k] = s_row[k];
    }
}

// -------------------------------------------------------------------- uniform lattices
// A structured vertex grid (Mesh::sym_nx > 0: row = x + nx y + nx ny z) whose coordinates are origin + index * step per
// axis to within the rounding of that formula: the assembly then takes edge vectors as whole steps (p1_geometry).
__global__ __launch_bounds__(TPB) void k_lattice_verify(const double *__restrict__ cx, const double *__restrict__ cy,
                                                        const double *__restrict__ cz, int64_t nv, int nx, int ny, int nz,
                                                        double *__restrict__ h_out, int *__restrict__ flag) {
    const int64_t v = (int64_t)blockIdx.x * TPB + threadIdx.x;
    if (v >= nv) return;
    const int64_t P = (int64_t)nx * ny;
    const int iz = (int)(v / P), rem = (int)(v - (int64_t)iz * P), iy = rem / nx, ix = rem - iy * nx;
    const double *cc[3] = {cx, cy, cz};
    const int idx[3] = {ix, iy, iz}, cnt[3] = {nx, ny, nz};
    bool ok = true;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const double o = cc[a][0], f = cc[a][nv - 1];
        const double h = (f - o) / (double)(cnt[a] - 1);
        const double scale = fmax(fmax(fabs(o), fabs(f)), fabs(f - o));
        if (!(fabs(h) > 0.0) || !(fabs(cc[a][v] - fma((double)idx[a], h, o)) <= 8.0 * 2.220446049250313e-16 * scale)) ok = false;
        if (v == 0) h_out[a] = h;
    }
    if (!ok) flag[0] = 1;
}

// every cell of a lattice mesh joins vertices that differ by at most one lattice step along every axis: then a cell's edge vectors
// follow from its vertex INDICES alone (k_assemble_p1<3>: no coordinate is read)
__global__ __launch_bounds__(TPB) void k_lattice_cells_verify(const int4 *__restrict__ cells, int64_t nc, int nx, int ny, int *__restrict__ flag) {
    const int64_t k = (int64_t)blockIdx.x * TPB + threadIdx.x;
    if (k >= nc) return;
    const int4 c4 = cells[k];
    const int u[4] = {c4.x, c4.y, c4.z, c4.w};
    const int P = nx * ny;
    int lo[3], hi[3];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int iz = u[t] / P, rem = u[t] - iz * P, iy = rem / nx, ix = rem - iy * nx;
        const int id[3] = {ix, iy, iz};
#pragma unroll
        for (int a = 0; a < 3; ++a) { lo[a] = t ? min(lo[a], id[a]) : id[a]; hi[a] = t ? max(hi[a], id[a]) : id[a]; }
    }
    if (hi[0] - lo[0] > 1 || hi[1] - lo[1] > 1 || hi[2] - lo[2] > 1) flag[0] = 1;
}

// cube corners v0..v7 = (dx, dy, dz) in {0, 1}^3, corner id = dx + 2 dy + 4 dz; tetrahedra of a cube in cell order:
//   (v0, v1, v3, v7), (v0, v1, v7, v5), (v0, v5, v7, v4), (v0, v3, v2, v7), (v0, v6, v4, v7), (v0, v2, v6, v7)
__host__ __device__ constexpr int box_corner(int t, int a) {
    return t == 0 ? (a == 0 ? 0 : a == 1 ? 1 : a == 2 ? 3 : 7) : t == 1 ? (a == 0 ? 0 : a == 1 ? 1 : a == 2 ? 7 : 5) :
           t == 2 ? (a == 0 ? 0 : a == 1 ? 5 : a == 2 ? 7 : 4) : t == 3 ? (a == 0 ? 0 : a == 1 ? 3 : a == 2 ? 2 : 7) :
           t == 4 ? (a == 0 ? 0 : a == 1 ? 6 : a == 2 ? 4 : 7) : (a == 0 ? 0 : a == 1 ? 2 : a == 2 ? 6 : 7);
}
__host__ __device__ constexpr int box_local(int t, int corner) {      // local index of that corner in tetrahedron t, -1: not a vertex of it
    return box_corner(t, 0) == corner ? 0 : box_corner(t, 1) == corner ? 1 : box_corner(t, 2) == corner ? 2 : box_corner(t, 3) == corner ? 3 : -1;
}
// slot of the neighbour at lattice offset (dx, dy, dz) in a row's ascending columns (all 15 present): -1 off the pattern
__host__ __device__ constexpr int box_slot(int dx, int dy, int dz) {
    return (dz == -1) ? ((dy == -1) ? (dx == -1 ? 0 : dx == 0 ? 1 : -1) : (dy == 0) ? (dx == -1 ? 2 : dx == 0 ? 3 : -1) : -1) :
           (dz == 0) ? ((dy == -1) ? (dx == -1 ? 4 : dx == 0 ? 5 : -1) : (dy == 0) ? (dx == -1 ? 6 : dx == 0 ? 7 : 8) : (dx == 0 ? 9 : dx == 1 ? 10 : -1)) :
           ((dy == 0) ? (dx == 0 ? 11 : dx == 1 ? 12 : -1) : (dy == 1) ? (dx == 0 ? 13 : dx == 1 ? 14 : -1) : -1);
}

// cell 6 q + t = (first vertex of cube q) + the offsets of cell t (cube 0): the 6-tetrahedra box mesh in its natural numbering
struct RegPat { int off[6][4]; int loc[6][8]; };
__global__ __launch_bounds__(TPB) void k_lattice_regular_verify(const int4 *__restrict__ cells, int64_t nc, int nx, int ny, RegPat R, int *__restrict__ flag) {
    const int64_t k = (int64_t)blockIdx.x * TPB + threadIdx.x;
    if (k >= nc) return;
    const int64_t q = k / 6;
    const int t = (int)(k - 6 * q), ncx = nx - 1, ncy = ny - 1;
    const int cz = (int)(q / ((int64_t)ncx * ncy)), rem = (int)(q - (int64_t)cz * ncx * ncy), cy = rem / ncx, cx = rem - cy * ncx;
    const int base = cx + nx * cy + nx * ny * cz;
    const int4 c4 = cells[k];
    if (c4.x != base + R.off[t][0] || c4.y != base + R.off[t][1] || c4.z != base + R.off[t][2] || c4.w != base + R.off[t][3]) flag[0] = 1;
}

static int detect_regular_cells(Ctx *c, Mesh *m) {
    m->lattice_regular = false;
    const int nx = m->sym_nx, ny = m->sym_ny;
    const int64_t P = (int64_t)nx * ny;
    const int nz = (int)(m->nv / P);
    if (m->nc != (int64_t)6 * (nx - 1) * (ny - 1) * (nz - 1) || m->nc < 6) return PGD_OK;
    int4 first[6];
    PGD_HIP(c, hipMemcpyAsync(first, m->cells, sizeof first, hipMemcpyDeviceToHost, c->stream));
    PGD_HIP(c, hipStreamSynchronize(c->stream));
    RegPat R;
    for (int t = 0; t < 6; ++t) {
        const int u[4] = {first[t].x, first[t].y, first[t].z, first[t].w};
        for (int q = 0; q < 8; ++q) R.loc[t][q] = -1;
        for (int a = 0; a < 4; ++a) {
            if (u[a] < 0) return PGD_OK;
            const int dz = (int)(u[a] / P), rem = (int)(u[a] - dz * P), dy = rem / nx, dx = rem - dy * nx;
            if (dx > 1 || dy > 1 || dz > 1 || R.loc[t][dx + 2 * dy + 4 * dz] >= 0) return PGD_OK;      // not a corner of cube 0 (or twice the same)
            R.off[t][a] = u[a];
            R.loc[t][dx + 2 * dy + 4 * dz] = a;
        }
    }
    void *q;
    PGD_TRY(dev_alloc(c, &q, sizeof(int)));
    int *fl = (int *)q;
    PGD_HIP(c, hipMemsetAsync(fl, 0, sizeof(int), c->stream));
    k_lattice_regular_verify<<<(int)((m->nc + TPB - 1) / TPB), TPB, 0, c->stream>>>(m->cells, m->nc, nx, ny, R, fl);
    int bad = 1;
    PGD_HIP(c, hipMemcpyAsync(&bad, fl, sizeof bad, hipMemcpyDeviceToHost, c->stream));
    PGD_HIP(c, hipStreamSynchronize(c->stream));
    (void)hipFree(fl);
    PGD_LAUNCH_CHECK(c);
    bool box = bad == 0;
    for (int t = 0; t < 6 && box; ++t)
        for (int q = 0; q < 8; ++q) box = box && R.loc[t][q] == box_local(t, q);      // ... cut like dolfin's BoxMesh (k_assemble_p1_regular's pattern)
    if (box) {
        m->lattice_regular = true;
        memcpy(m->pat_off, R.off, sizeof R.off);
        memcpy(m->pat_loc, R.loc, sizeof R.loc);
    }
    return PGD_OK;
}

static int detect_lattice(Ctx *c, Mesh *m) {
    m->lattice = false;
    m->lattice_unit = false;
    if (m->sym_nx <= 0 || m->gdim != 3 || m->ncomp != 1 || !m->coords || m->cellsN) return PGD_OK;
    const int64_t plane = (int64_t)m->sym_nx * m->sym_ny;
    const int nz = (int)(m->nv / plane);
    if (m->sym_nx < 2 || m->sym_ny < 2 || nz < 2) return PGD_OK;
    void *p;
    PGD_TRY(dev_alloc(c, &p, 4 * sizeof(double)));
    double *buf = (double *)p;                               // h[3], then the flag
    PGD_HIP(c, hipMemsetAsync(buf, 0, 4 * sizeof(double), c->stream));
    k_lattice_verify<<<(int)((m->nv + TPB - 1) / TPB), TPB, 0, c->stream>>>(m->coords, m->coords + m->nv, m->coords + 2 * m->nv, m->nv,
                                                                        m->sym_nx, m->sym_ny, nz, buf, (int *)(buf + 3));
    double host[4] = {0, 0, 0, 0};
    PGD_HIP(c, hipMemcpyAsync(host, buf, sizeof host, hipMemcpyDeviceToHost, c->stream));
    PGD_HIP(c, hipStreamSynchronize(c->stream));
    (void)hipFree(buf);
    PGD_LAUNCH_CHECK(c);
    int bad;
    memcpy(&bad, &host[3], sizeof bad);
    if (bad == 0) {
        m->lattice = true;
        for (int k = 0; k < 3; ++k) m->lat_h[k] = host[k];
        // ... and its cells are unit cells (P1 tetrahedra, int32 vertex ids, planes of at least 3 x 3 vertices)
        if (m->cells && m->nvpc == 4 && m->sym_nx >= 3 && m->sym_ny >= 3 && m->nv < ((int64_t)1 << 31)) {
            void *q;
            PGD_TRY(dev_alloc(c, &q, sizeof(int)));
            int *fl = (int *)q;
            PGD_HIP(c, hipMemsetAsync(fl, 0, sizeof(int), c->stream));
            k_lattice_cells_verify<<<(int)((m->nc + TPB - 1) / TPB), TPB, 0, c->stream>>>(m->cells, m->nc, m->sym_nx, m->sym_ny, fl);
            int far = 1;
            PGD_HIP(c, hipMemcpyAsync(&far, fl, sizeof far, hipMemcpyDeviceToHost, c->stream));
            PGD_HIP(c, hipStreamSynchronize(c->stream));
            (void)hipFree(fl);
            PGD_LAUNCH_CHECK(c);
            m->lattice_unit = far == 0;
            if (m->lattice_unit) PGD_TRY(detect_regular_cells(c, m));
        }
    }
    return PGD_OK;
}

// -------------------------------------------------------------------- assembly
struct AsmArgs {
    const double *cx, *cy, *cz;   // SoA coordinates
    const int4 *cells;
    const int *v2c_ptr, *v2c, *row_ptr, *cols;
    const double *w;              // vertex weights (weighted kinds)
    double *vals;
    int64_t nv;
    int kind, da, db;
    int lattice;                  // Mesh::lattice: edge vectors are whole lattice steps
    double lat_h[3], lat_inv[3];
    int lat_unit, nx, ny;         // Mesh::lattice_unit: ... of at most one step per axis, taken from the vertex indices (row = x + nx y + nx ny z)
};

// lattice offset (dx, dy, dz), each in {-1, 0, 1}, of vertex v relative to vertex v0 (unit cells, nx, ny >= 3)
__device__ __forceinline__ void lattice_step(int d, int nx, int P, int &dx, int &dy, int &dz) {
    dz = d > P / 2 ? 1 : (d < -(P / 2) ? -1 : 0);
    const int r = d - dz * P;
    dy = 2 * r > nx ? 1 : (2 * r < -nx ? -1 : 0);
    dx = r - dy * nx;
}

template <int D>
__device__ __forceinline__ void p1_geometry(const AsmArgs &A, const int *u, double &vol, double g[D + 1][D]) {
    if constexpr (D == 1) {
        const double e = A.cx[u[1]] - A.cx[u[0]];
        vol = fabs(e);
        g[1][0] = 1.0 / e;
        g[0][0] = -g[1][0];
    } else if constexpr (D == 2) {
        const double x0 = A.cx[u[0]], y0 = A.cy[u[0]];
        const double ax = A.cx[u[1]] - x0, ay = A.cy[u[1]] - y0;
        const double bx = A.cx[u[2]] - x0, by = A.cy[u[2]] - y0;
        const double det = ax * by - ay * bx, inv = 1.0 / det;
        vol = 0.5 * fabs(det);
        g[1][0] = by * inv;  g[1][1] = -bx * inv;
        g[2][0] = -ay * inv; g[2][1] = ax * inv;
        g[0][0] = -(g[1][0] + g[2][0]);
        g[0][1] = -(g[1][1] + g[2][1]);
    } else {
        double ax, ay, az, bx, by, bz, cx, cy, cz;
        if (A.lat_unit) {
            // unit cells of a uniform lattice: the edge vectors are (dx hx, dy hy, dz hz) with the steps read off the vertex INDICES -
            // the very numbers the rounding below produces from the coordinates, without the twelve scattered coordinate reads per
            // cell visit (r04: 63 KB pulled through the L1 per row against 400 B of unique data, profiles/r03_assembly_counters.txt)
            const int P = A.nx * A.ny;
            int i1, j1, k1, i2, j2, k2, i3, j3, k3;
            lattice_step(u[1] - u[0], A.nx, P, i1, j1, k1);
            lattice_step(u[2] - u[0], A.nx, P, i2, j2, k2);
            lattice_step(u[3] - u[0], A.nx, P, i3, j3, k3);
            ax = (double)i1 * A.lat_h[0]; ay = (double)j1 * A.lat_h[1]; az = (double)k1 * A.lat_h[2];
            bx = (double)i2 * A.lat_h[0]; by = (double)j2 * A.lat_h[1]; bz = (double)k2 * A.lat_h[2];
            cx = (double)i3 * A.lat_h[0]; cy = (double)j3 * A.lat_h[1]; cz = (double)k3 * A.lat_h[2];
        } else {
            const double x0 = A.cx[u[0]], y0 = A.cy[u[0]], z0 = A.cz[u[0]];
            ax = A.cx[u[1]] - x0; ay = A.cy[u[1]] - y0; az = A.cz[u[1]] - z0;
            bx = A.cx[u[2]] - x0; by = A.cy[u[2]] - y0; bz = A.cz[u[2]] - z0;
            cx = A.cx[u[3]] - x0; cy = A.cy[u[3]] - y0; cz = A.cz[u[3]] - z0;
        }
        if (A.lattice && !A.lat_unit) {
            // vertices on a uniform lattice (to the rounding of their coordinates, checked at upload): every edge component
            // is a whole number of steps - taken as exactly that, so congruent cells get IDENTICAL local matrices and the
            // assembled rows of a uniform grid repeat bit for bit (what the row-class dictionary of the products lives on).
            // The coordinate differences carry the rounding of i * h; this removes it (relative change ~ 1e-16).
            ax = rint(ax * A.lat_inv[0]) * A.lat_h[0]; ay = rint(ay * A.lat_inv[1]) * A.lat_h[1]; az = rint(az * A.lat_inv[2]) * A.lat_h[2];
            bx = rint(bx * A.lat_inv[0]) * A.lat_h[0]; by = rint(by * A.lat_inv[1]) * A.lat_h[1]; bz = rint(bz * A.lat_inv[2]) * A.lat_h[2];
            cx = rint(cx * A.lat_inv[0]) * A.lat_h[0]; cy = rint(cy * A.lat_inv[1]) * A.lat_h[1]; cz = rint(cz * A.lat_inv[2]) * A.lat_h[2];
        }
        // cross products: b x c, c x a, a x b
        const double n1x = by * cz - bz * cy, n1y = bz * cx - bx * cz, n1z = bx * cy - by * cx;
        const double n2x = cy * az - cz * ay, n2y = cz * ax - cx * az, n2z = cx * ay - cy * ax;
        const double n3x = ay * bz - az * by, n3y = az * bx - ax * bz, n3z = ax * by - ay * bx;
        const double det = ax * n1x + ay * n1y + az * n1z, inv = 1.0 / det;
        vol = fabs(det) * (1.0 / 6.0);
        g[1][0] = n1x * inv; g[1][1] = n1y * inv; g[1][2] = n1z * inv;
        g[2][0] = n2x * inv; g[2][1] = n2y * inv; g[2][2] = n2z * inv;
        g[3][0] = n3x * inv; g[3][1] = n3y * inv; g[3][2] = n3z * inv;
        for (int k = 0; k < 3; ++k) g[0][k] = -(g[1][k] + g[2][k] + g[3][k]);
    }
}

// entry (i = test, j = trial) of the local matrix; closed forms for P1 simplices
// gi = gradient of the row's own basis function (selected by the caller without dynamic register indexing: a private array
// indexed with a run-time index lives in SCRATCH memory - 112 bytes per lane here, 40 GB of scratch traffic per atom at 256^3,
// profiles/r03_assembly_counters.txt), gj = that of column j (compile-time j)
template <int D>
__device__ __forceinline__ double p1_entry(int kind, int da, int db, int i, int j, double vol, const double (&gi)[D],
                                           const double (&gj)[D], const double (&wl)[D + 1]) {
    constexpr double MFAC = 1.0 / ((D + 1) * (D + 2));
    constexpr double WFAC = (D == 1) ? 1.0 / 24.0 : (D == 2) ? 2.0 / 120.0 : 6.0 / 720.0;   // D!/(D+3)!
    double gib = gi[0], gja = gj[0];                         // components db of gi and da of gj
#pragma unroll
    for (int k = 1; k < D; ++k) { if (k == db) gib = gi[k]; if (k == da) gja = gj[k]; }
    switch (kind) {
        case PGD_ATOM_MASS: return vol * MFAC * (i == j ? 2.0 : 1.0);
        case PGD_ATOM_STIFF: {
            double s = 0.0;
#pragma unroll
            for (int k = 0; k < D; ++k) s += gi[k] * gj[k];
            return vol * s;
        }
        case PGD_ATOM_DUDV: return vol * gib * gja;
        case PGD_ATOM_CONV: return vol * (1.0 / (D + 1)) * gja;
        case PGD_ATOM_CONVT: return vol * (1.0 / (D + 1)) * gib;
        case PGD_ATOM_WMASS: {
            double s = 0.0;
#pragma unroll
            for (int k = 0; k < D + 1; ++k) {
                const double cijk = (i == j) ? (k == i ? 6.0 : 2.0) : ((k == i || k == j) ? 2.0 : 1.0);
                s += cijk * wl[k];
            }
            return vol * WFAC * s;
        }
        case PGD_ATOM_WSTIFF: {
            double s = 0.0, wb = 0.0;
#pragma unroll
            for (int k = 0; k < D; ++k) s += gi[k] * gj[k];
#pragma unroll
            for (int k = 0; k < D + 1; ++k) wb += wl[k];
            return vol * (wb * (1.0 / (D + 1))) * s;
        }
    }
    return 0.0;
}
template <int D>
__global__ __launch_bounds__(TPB) void k_assemble_p1(AsmArgs A) {
    __shared__ double s_acc[ASM_CAP];
    __shared__ int s_cols[ASM_CAP];
    __shared__ int s_rp[TPB + 1];
    const int tid = threadIdx.x;
    const int64_t r0 = (int64_t)blockIdx.x * TPB;
    const int nr = (int)min((int64_t)TPB, A.nv - r0);
    if (tid < nr) s_rp[tid] = A.row_ptr[r0 + tid];
    if (tid == 0) s_rp[nr] = A.row_ptr[r0 + nr];
    __syncthreads();
    const int s = s_rp[0], e = s_rp[nr];
    const bool staged = (e - s) <= ASM_CAP;   // uniform
    if (staged) {
        for (int k = tid; k < e - s; k += TPB) { s_acc[k] = 0.0; s_cols[k] = A.cols[s + k]; }
    } else if (tid < nr) {
        for (int k = s_rp[tid]; k < s_rp[tid + 1]; ++k) A.vals[k] = 0.0;
    }
    __syncthreads();
    if (tid < nr) {
        const int r = (int)(r0 + tid);
        const int ra = s_rp[tid], len = s_rp[tid + 1] - ra;
        double *acc = staged ? (s_acc + (ra - s)) : (A.vals + ra);
        const int *rc = staged ? (s_cols + (ra - s)) : (A.cols + ra);
        const int ca = A.v2c_ptr[r], cb = A.v2c_ptr[r + 1];
        for (int k = ca; k < cb; ++k) {
            const int4 c4 = A.cells[A.v2c[k]];
            const int u[4] = {c4.x, c4.y, c4.z, c4.w};
            int i = 0;
#pragma unroll
            for (int t = 0; t < D + 1; ++t) if (u[t] == r) i = t;
            double vol, g[D + 1][D], wl[D + 1], gi[D];
            p1_geometry<D>(A, u, vol, g);
#pragma unroll
            for (int t = 0; t < D + 1; ++t) wl[t] = A.w ? A.w[u[t]] : 0.0;
#pragma unroll
            for (int k = 0; k < D; ++k) {
                gi[k] = g[0][k];
#pragma unroll
                for (int t = 1; t < D + 1; ++t) if (i == t) gi[k] = g[t][k];
            }
#pragma unroll
            for (int j = 0; j < D + 1; ++j) {
                const double val = p1_entry<D>(A.kind, A.da, A.db, i, j, vol, gi, g[j], wl);
                // cols are sorted and contain u[j]: branch-free lower bound, log2(len) steps (the linear walk it replaces was ~1400
                // LDS reads per row - 24 cells x 4 entries x ~15 - and most of the kernel's instructions)
                int pos = 0;
                for (int nleft = len; nleft > 1;) {
                    const int half = nleft >> 1;
                    pos += rc[pos + half - 1] < u[j] ? half : 0;
                    nleft -= half;
                }
                acc[pos] += val;
            }
        }
    }
    __syncthreads();
    if (staged)
        for (int k = tid; k < e - s; k += TPB) A.vals[s + k] = s_acc[k];
}


// Regularly numbered unit-cell lattices cut like dolfin's BoxMesh (Mesh::lattice_regular), unweighted kinds: NOTHING of the mesh is
// read but the row pointers.  The cells around vertex (x, y, z) are the tetrahedra of its up to eight cubes that have it as a corner -
// the cutting pattern below, verified for every cell at upload (k_lattice_regular_verify) - visited in ascending cell number like the
// vertex->cell list of the general kernel; a cell's local matrix depends on its type t alone and comes from a table of 6 x 16 entries
// that every workgroup computes first with the general kernel's own arithmetic (p1_geometry on the type's steps, p1_entry); with the
// pattern known at compile time every contribution goes to one of the row's 15 stencil slots by a CONSTANT index - registers, no
// search - and a slot belongs to the row's CSR entries iff one of the cubes that feed it exists (that is how the pattern was built).
// Same values, same order of summation: bit-identical atoms.  (r04: the general kernel is bound by its gathers - a cell record per
// visit, 24 visits per row; a table behind one more gather, the cell's type byte, was SLOWER; gather-free with the binary search into
// the LDS image of the CSR rows: 5.6 ms at 256^3; HISTORY.md.)  Reads 4 B, writes 8 x 15 B per row.
__global__ __launch_bounds__(TPB) void k_assemble_p1_regular(AsmArgs A, int nz) {
    __shared__ double s_acc[TPB * 15];
    __shared__ int s_rp[TPB + 1];
    __shared__ double s_loc[6 * 16];
    const int tid = threadIdx.x;
    const int64_t r0 = (int64_t)blockIdx.x * TPB;
    const int nr = (int)min((int64_t)TPB, A.nv - r0);
    const int nx = A.nx, P = A.nx * A.ny;
    if (tid < nr) s_rp[tid] = A.row_ptr[r0 + tid];
    if (tid == 0) s_rp[nr] = A.row_ptr[r0 + nr];
    if (tid < 6) {
        int u[4];
#pragma unroll
        for (int a = 0; a < 4; ++a) {                     // (the steps between the vertices are all p1_geometry takes)
            int cn = 0;
#pragma unroll
            for (int t = 0; t < 6; ++t) if (t == tid) cn = box_corner(t, a);
            u[a] = 4 * P + 4 * nx + 4 + (cn & 1) + nx * ((cn >> 1) & 1) + P * (cn >> 2);
        }
        double vol, g[4][3];
        const double wl[4] = {0.0, 0.0, 0.0, 0.0};
        p1_geometry<3>(A, u, vol, g);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) s_loc[tid * 16 + i * 4 + j] = p1_entry<3>(A.kind, A.da, A.db, i, j, vol, g[i], g[j], wl);
    }
    __syncthreads();
    const int s = s_rp[0];
    if (tid < nr) {
        const int r = (int)(r0 + tid);
        const int z = r / P, rem = r - z * P, y = rem / nx, x = rem - y * nx;
        double acc[15];
        unsigned present = 0;
#pragma unroll
        for (int q = 0; q < 15; ++q) acc[q] = 0.0;
#pragma unroll
        for (int o = 7; o >= 0; --o) {                      // the vertex as corner o of the cube at (x - ox, y - oy, z - oz): ascending cube number
            const int ox = o & 1, oy = (o >> 1) & 1, oz = o >> 2;
            const int cx = x - ox, cy = y - oy, cz = z - oz;
            const bool have = cx >= 0 && cy >= 0 && cz >= 0 && cx < nx - 1 && cy < A.ny - 1 && cz < nz - 1;
#pragma unroll
            for (int t = 0; t < 6; ++t) {
                const int i = box_local(t, o);
                if (i < 0) continue;                        // (compile time)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int cj = box_corner(t, j);
                    const int q = box_slot((cj & 1) - ox, ((cj >> 1) & 1) - oy, (cj >> 2) - oz);
                    const double val = s_loc[t * 16 + 4 * i + j];
                    if (have) { acc[q] += val; present |= 1u << q; }
                }
            }
        }
        // the row's CSR entries: the slots that got a contribution, in ascending column order
        double *out = s_acc + (s_rp[tid] - s);
        int k = 0;
#pragma unroll
        for (int q = 0; q < 15; ++q)
            if (present & (1u << q)) out[k++] = acc[q];
    }
    __syncthreads();
    const int e = s_rp[nr];
    for (int k = tid; k < e - s; k += TPB) A.vals[s + k] = s_acc[k];
}

// Quadratic Lagrange elements on intervals (cell record = v0, v1, midpoint node): owner-computes like
// the P1 kernel, local 3x3 entries by 4-point Gauss quadrature (exact to degree 7).  These systems are
// small (time / parameter dimensions), so every lane accumulates straight into its own CSR row.
__global__ __launch_bounds__(TPB) void k_assemble_p2_interval(AsmArgs A) {
    const double GX[4] = {0.06943184420297371, 0.33000947820757187, 0.6699905217924281, 0.9305681557970262};
    const double GW[4] = {0.17392742256872692, 0.32607257743127305, 0.32607257743127305, 0.17392742256872692};
    const int64_t r = (int64_t)blockIdx.x * TPB + threadIdx.x;
    if (r >= A.nv) return;
    const int ra = A.row_ptr[r], len = A.row_ptr[r + 1] - ra;
    for (int k = 0; k < len; ++k) A.vals[ra + k] = 0.0;
    for (int k = A.v2c_ptr[r]; k < A.v2c_ptr[r + 1]; ++k) {
        const int4 c4 = A.cells[A.v2c[k]];
        const int u[3] = {c4.x, c4.y, c4.z};
        const int i = (u[0] == (int)r) ? 0 : (u[1] == (int)r) ? 1 : 2;
        const double hs = A.cx[u[1]] - A.cx[u[0]], inv = 1.0 / hs, ah = fabs(hs);
        double wl[3] = {1.0, 1.0, 1.0};
        const bool weighted = A.kind == PGD_ATOM_WMASS || A.kind == PGD_ATOM_WSTIFF;
        if (weighted) { wl[0] = A.w[u[0]]; wl[1] = A.w[u[1]]; wl[2] = A.w[u[2]]; }
        double loc[3] = {0.0, 0.0, 0.0};
        for (int q = 0; q < 4; ++q) {
            const double s = GX[q];
            const double N[3] = {(1 - s) * (1 - 2 * s), s * (2 * s - 1), 4 * s * (1 - s)};
            const double dN[3] = {(4 * s - 3) * inv, (4 * s - 1) * inv, (4 - 8 * s) * inv};
            double jac = ah * GW[q];
            if (weighted) jac *= wl[0] * N[0] + wl[1] * N[1] + wl[2] * N[2];
            for (int j = 0; j < 3; ++j) {
                double f;
                switch (A.kind) {
                    case PGD_ATOM_MASS: case PGD_ATOM_WMASS: f = N[i] * N[j]; break;
                    case PGD_ATOM_CONV: f = N[i] * dN[j]; break;
                    case PGD_ATOM_CONVT: f = dN[i] * N[j]; break;
                    default: f = dN[i] * dN[j]; break;      // STIFF, DUDV(0,0), WSTIFF
                }
                loc[j] = fma(jac, f, loc[j]);
            }
        }
        for (int j = 0; j < 3; ++j) {
            int pos = 0;
            while (pos < len - 1 && A.cols[ra + pos] < u[j]) ++pos;
            A.vals[ra + pos] += loc[j];
        }
    }
}

// Quadratic Lagrange elements on triangles / tetrahedra.  Cell record = (vertices, then one node per edge in
// the UFC local edge order); shape functions in barycentric coordinates: vertex i: l_i (2 l_i - 1), edge
// (a, b): 4 l_a l_b.  Local entries by a conical-product Gauss-Jacobi rule (4 points per direction, exact to
// degree 7 - the highest integrand, weight * N_i * N_j, has degree 6).  One lane per row, accumulating
// into its own CSR row in ascending cell order (no atomics, reproducible).
__constant__ double GJ_X[3][4] = {
    {0.06943184420297371, 0.33000947820757187, 0.6699905217924281, 0.9305681557970262},
    {0.057104196114517725, 0.2768430136381238, 0.5835904323689168, 0.8602401356562195},
    {0.048500549446997276, 0.23860073755186234, 0.5170472951043674, 0.7958514178967728}};
__constant__ double GJ_W[3][4] = {   // weight (1 - t)^alpha on [0, 1], alpha = 0, 1, 2
    {0.1739274225687269, 0.3260725774312731, 0.3260725774312731, 0.1739274225687269},
    {0.13550691343148852, 0.2034645680102711, 0.12984754760823233, 0.031180970950008085},
    {0.11088841561127774, 0.14345878979921445, 0.0686338871729231, 0.010352240749918081}};
__constant__ int P2_EA[2][6] = {{1, 0, 0, 0, 0, 0}, {2, 1, 1, 0, 0, 0}};
__constant__ int P2_EB[2][6] = {{2, 2, 1, 0, 0, 0}, {3, 3, 2, 3, 2, 1}};

template <int D>
__global__ __launch_bounds__(64) void k_assemble_p2_simplex(AsmArgs A, const int *__restrict__ cellsN) {
    constexpr int NN = (D + 1) * (D + 2) / 2, NE = NN - (D + 1), NQ = (D == 2) ? 16 : 64;
    const int64_t r = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (r >= A.nv) return;
    const int ra = A.row_ptr[r], len = A.row_ptr[r + 1] - ra;
    for (int k = 0; k < len; ++k) A.vals[ra + k] = 0.0;
    const bool weighted = A.kind == PGD_ATOM_WMASS || A.kind == PGD_ATOM_WSTIFF;
    for (int k = A.v2c_ptr[r]; k < A.v2c_ptr[r + 1]; ++k) {
        const int *rec = cellsN + (int64_t)A.v2c[k] * NN;
        int u[NN], i = 0;
#pragma unroll
        for (int t = 0; t < NN; ++t) { u[t] = rec[t]; if (u[t] == (int)r) i = t; }
        double vol, g[D + 1][D], wl[NN], loc[NN];
        p1_geometry<D>(A, u, vol, g);
#pragma unroll
        for (int t = 0; t < NN; ++t) { wl[t] = weighted ? A.w[u[t]] : 0.0; loc[t] = 0.0; }
        for (int q = 0; q < NQ; ++q) {
            double lam[D + 1], wq;
            if constexpr (D == 2) {
                const int qu = q >> 2, qv = q & 3;
                const double a = GJ_X[1][qu], b = GJ_X[0][qv];
                lam[1] = a; lam[2] = b * (1.0 - a); lam[0] = 1.0 - lam[1] - lam[2];
                wq = 2.0 * GJ_W[1][qu] * GJ_W[0][qv];
            } else {
                const int qu = q >> 4, qv = (q >> 2) & 3, qw = q & 3;
                const double a = GJ_X[2][qu], b = GJ_X[1][qv], cc = GJ_X[0][qw];
                lam[1] = a; lam[2] = b * (1.0 - a); lam[3] = cc * (1.0 - a) * (1.0 - b);
                lam[0] = 1.0 - lam[1] - lam[2] - lam[3];
                wq = 6.0 * GJ_W[2][qu] * GJ_W[1][qv] * GJ_W[0][qw];
            }
            double N[NN], dN[NN][D];
#pragma unroll
            for (int t = 0; t < D + 1; ++t) {
                N[t] = lam[t] * (2.0 * lam[t] - 1.0);
#pragma unroll
                for (int d = 0; d < D; ++d) dN[t][d] = (4.0 * lam[t] - 1.0) * g[t][d];
            }
#pragma unroll
            for (int e = 0; e < NE; ++e) {
                const int a = P2_EA[D - 2][e], b = P2_EB[D - 2][e];
                N[D + 1 + e] = 4.0 * lam[a] * lam[b];
#pragma unroll
                for (int d = 0; d < D; ++d) dN[D + 1 + e][d] = 4.0 * (lam[b] * g[a][d] + lam[a] * g[b][d]);
            }
            double jac = vol * wq;
            if (weighted) {
                double wv = 0.0;
#pragma unroll
                for (int t = 0; t < NN; ++t) wv = fma(wl[t], N[t], wv);
                jac *= wv;
            }
            // select row i without dynamic register indexing
            double Ni = 0.0, dNi[D];
#pragma unroll
            for (int d = 0; d < D; ++d) dNi[d] = 0.0;
#pragma unroll
            for (int t = 0; t < NN; ++t) if (t == i) {
                Ni = N[t];
#pragma unroll
                for (int d = 0; d < D; ++d) dNi[d] = dN[t][d];
            }
            double dNi_b = 0.0;
#pragma unroll
            for (int d = 0; d < D; ++d) if (d == A.db) dNi_b = dNi[d];
#pragma unroll
            for (int j = 0; j < NN; ++j) {
                double f = 0.0, dNj_a = 0.0;
#pragma unroll
                for (int d = 0; d < D; ++d) if (d == A.da) dNj_a = dN[j][d];
                switch (A.kind) {
                    case PGD_ATOM_MASS: case PGD_ATOM_WMASS: f = Ni * N[j]; break;
                    case PGD_ATOM_CONV: f = Ni * dNj_a; break;
                    case PGD_ATOM_CONVT: f = dNi_b * N[j]; break;
                    case PGD_ATOM_DUDV: f = dNi_b * dNj_a; break;
                    default:
#pragma unroll
                        for (int d = 0; d < D; ++d) f = fma(dNi[d], dN[j][d], f);
                        break;
                }
                loc[j] = fma(jac, f, loc[j]);
            }
        }
#pragma unroll
        for (int j = 0; j < NN; ++j) {
            int lo = 0, hi = len - 1;            // binary search: cols are sorted and contain u[j]
            while (lo < hi) { const int mid = (lo + hi) >> 1; if (A.cols[ra + mid] < u[j]) lo = mid + 1; else hi = mid; }
            A.vals[ra + lo] += loc[j];
        }
    }
}

// --------------------------------------------------------------------- host side
static int build_topology(Ctx *c, Mesh *m) {
    void *p;
    const int64_t nv = m->nv, nc = m->nc;
    int *cnt = nullptr, *stats = nullptr;
    PGD_TRY(dev_alloc(c, &p, (size_t)(nv + 1) * sizeof(int))); cnt = (int *)p;
    PGD_TRY(dev_alloc(c, &p, (size_t)(nv + 1) * sizeof(int))); m->v2c_ptr = (int *)p;
    PGD_TRY(dev_alloc(c, &p, (size_t)(nc * m->nvpc) * sizeof(int))); m->v2c = (int *)p;
    PGD_TRY(dev_alloc(c, &p, 8 * sizeof(int))); stats = (int *)p;
    auto cleanup = [&]() { (void)hipFree(cnt); (void)hipFree(stats); };
    hipStream_t st = c->stream;
    PGD_HIP(c, hipMemsetAsync(cnt, 0, (size_t)(nv + 1) * sizeof(int), st));
    PGD_HIP(c, hipMemsetAsync(stats, 0, 8 * sizeof(int), st));
    const bool flat = m->cellsN != nullptr;
    const void *recs = flat ? (const void *)m->cellsN : (const void *)m->cells;
    if (flat) k_v2c_count<10><<<grid_for(nc), TPB, 0, st>>>(recs, nc, m->nvpc, cnt);
    else k_v2c_count<4><<<grid_for(nc), TPB, 0, st>>>(recs, nc, m->nvpc, cnt);
    int rc = scan_exclusive_i32(c, cnt, m->v2c_ptr, nv);
    if (rc != PGD_OK) { cleanup(); return rc; }
    PGD_HIP(c, hipMemsetAsync(cnt, 0, (size_t)(nv + 1) * sizeof(int), st));
    if (flat) k_v2c_fill<10><<<grid_for(nc), TPB, 0, st>>>(recs, nc, m->nvpc, m->v2c_ptr, cnt, m->v2c);
    else k_v2c_fill<4><<<grid_for(nc), TPB, 0, st>>>(recs, nc, m->nvpc, m->v2c_ptr, cnt, m->v2c);
    k_v2c_sort<<<grid_for(nv), TPB, 0, st>>>(m->v2c_ptr, m->v2c, nv);
    // pattern: count, scan, fill
    PGD_TRY(dev_alloc(c, &p, (size_t)(nv + 1) * sizeof(int))); m->row_ptr = (int *)p;
    const int gb = flat ? (int)((nv + 63) / 64) : (int)((nv + TPB - 1) / TPB);
    if (flat) k_pattern<false, 10, MAX_ROW_P2, 64><<<gb, 64, 0, st>>>(recs, m->nvpc, m->v2c_ptr, m->v2c, nv, cnt, nullptr, nullptr, stats);
    else k_pattern<false, 4, MAX_ROW, TPB><<<gb, TPB, 0, st>>>(recs, m->nvpc, m->v2c_ptr, m->v2c, nv, cnt, nullptr, nullptr, stats);
    rc = scan_exclusive_i32(c, cnt, m->row_ptr, nv);
    if (rc != PGD_OK) { cleanup(); return rc; }
    int hstats[4] = {0, 0, 0, 0};
    int total = 0;
    PGD_HIP(c, hipMemcpyAsync(hstats, stats, sizeof hstats, hipMemcpyDeviceToHost, st));
    PGD_HIP(c, hipMemcpyAsync(&total, m->row_ptr + nv, sizeof(int), hipMemcpyDeviceToHost, st));
    PGD_HIP(c, hipStreamSynchronize(st));
    if (hstats[0]) { cleanup(); return fail(c, PGD_ERR_LIMIT, "mesh: a node has more than %d neighbours", flat ? MAX_ROW_P2 : MAX_ROW); }
    if (total < 0) { cleanup(); return fail(c, PGD_ERR_LIMIT, "mesh: nnz overflows int32"); }
    m->nnz = total;
    m->max_row = hstats[1];
    m->kl = hstats[2];
    m->ku = hstats[3];
    PGD_TRY(dev_alloc(c, &p, (size_t)(m->nnz > 0 ? m->nnz : 1) * sizeof(int))); m->cols = (int *)p;
    PGD_HIP(c, hipMemsetAsync(m->cols, 0, (size_t)(m->nnz > 0 ? m->nnz : 1) * sizeof(int) + PAD_BYTES, st));
    if (flat) k_pattern<true, 10, MAX_ROW_P2, 64><<<gb, 64, 0, st>>>(recs, m->nvpc, m->v2c_ptr, m->v2c, nv, nullptr, m->row_ptr, m->cols, stats);
    else k_pattern<true, 4, MAX_ROW, TPB><<<gb, TPB, 0, st>>>(recs, m->nvpc, m->v2c_ptr, m->v2c, nv, nullptr, m->row_ptr, m->cols, stats);
    PGD_HIP(c, hipStreamSynchronize(st));
    cleanup();
    PGD_LAUNCH_CHECK(c);
    return PGD_OK;
}


// ---------------------------------------------------------------- column dictionary
constexpr int DICT_TABLE = 4096;   // open-addressing table of 64-bit pattern hashes

__device__ __forceinline__ unsigned long long row_hash(const int *__restrict__ row_ptr,
                                                       const int *__restrict__ cols, int r) {
    const int a = row_ptr[r], b = row_ptr[r + 1];
    unsigned long long h = 1469598103934665603ULL ^ (unsigned)(b - a);
    h *= 1099511628211ULL;
    for (int k = a; k < b; ++k) { h ^= (unsigned)(cols[k] - r); h *= 1099511628211ULL; }
    return h ? h : 1ULL;
}

__device__ __forceinline__ int dict_find(const unsigned long long *keys, unsigned long long h) {
    int slot = (int)(h & (DICT_TABLE - 1));
    for (int probe = 0; probe < DICT_TABLE; ++probe) {
        const unsigned long long cur = keys[slot];
        if (cur == h) return slot;
        if (cur == 0ULL) return -1;
        slot = (slot + 1) & (DICT_TABLE - 1);
    }
    return -1;
}

// pass A: every distinct hash gets a table slot (ids are handed out after the kernel boundary)
__global__ __launch_bounds__(TPB) void k_dict_insert(const int *__restrict__ row_ptr, const int *__restrict__ cols,
                                                     int64_t nv, unsigned long long *keys, int *flags) {
    for (int64_t r = (int64_t)blockIdx.x * TPB + threadIdx.x; r < nv; r += (int64_t)gridDim.x * TPB) {
        if (row_ptr[r + 1] - row_ptr[r] > DICT_DLEN) { flags[0] = 1; continue; }
        const unsigned long long h = row_hash(row_ptr, cols, (int)r);
        int slot = (int)(h & (DICT_TABLE - 1));
        bool done = false;
        for (int probe = 0; probe < DICT_TABLE && !done; ++probe) {
            unsigned long long cur = keys[slot];          // cheap look first: almost always already there
            if (cur == 0ULL) cur = atomicCAS(&keys[slot], 0ULL, h);
            if (cur == 0ULL || cur == h) done = true;
            else slot = (slot + 1) & (DICT_TABLE - 1);
        }
        if (!done) flags[0] = 1;                          // table full: far too many patterns
    }
}

// pass B: number the occupied slots
__global__ void k_dict_ids(const unsigned long long *keys, int *ids, int *count) {
    const int slot = blockIdx.x * blockDim.x + threadIdx.x;
    if (slot < DICT_TABLE) ids[slot] = keys[slot] ? atomicAdd(count, 1) : -1;
}

// pass C: pattern id per row + the lowest row of every pattern as its representative
__global__ __launch_bounds__(TPB) void k_dict_assign(const int *__restrict__ row_ptr, const int *__restrict__ cols,
                                                     int64_t nv, const unsigned long long *keys, const int *ids,
                                                     uint16_t *pids, int *rep, int *flags) {
    for (int64_t r = (int64_t)blockIdx.x * TPB + threadIdx.x; r < nv; r += (int64_t)gridDim.x * TPB) {
        const int slot = dict_find(keys, row_hash(row_ptr, cols, (int)r));
        const int id = slot >= 0 ? ids[slot] : -1;
        if (id < 0 || id >= DICT_MAXP) { flags[0] = 1; pids[r] = 0; continue; }
        pids[r] = (uint16_t)id;
        if (rep[id] > (int)r) atomicMin(&rep[id], (int)r);
    }
}

// pass D: the table of relative offsets, from the representatives
__global__ void k_dict_build(const int *__restrict__ row_ptr, const int *__restrict__ cols, const int *rep,
                             int count, int *dict_off) {
    const int id = blockIdx.x, k = threadIdx.x;       // DICT_DLEN threads
    if (id >= count) return;
    const int r = rep[id], a = row_ptr[r], len = row_ptr[r + 1] - a;
    dict_off[id * DICT_DLEN + k] = (k < len) ? cols[a + k] - r : 0;
}

// pass E: every row must decode to exactly its column ids (guards against hash collisions)
__global__ __launch_bounds__(TPB) void k_dict_verify(const int *__restrict__ row_ptr, const int *__restrict__ cols,
                                                     int64_t nv, const uint16_t *pids, const int *dict_off, int *flags) {
    for (int64_t r = (int64_t)blockIdx.x * TPB + threadIdx.x; r < nv; r += (int64_t)gridDim.x * TPB) {
        const int a = row_ptr[r], len = row_ptr[r + 1] - a;
        const int *off = dict_off + (int)pids[r] * DICT_DLEN;
        bool ok = len <= DICT_DLEN;
        for (int k = 0; ok && k < len; ++k) ok = (cols[a + k] == (int)r + off[k]);
        if (!ok) flags[0] = 1;
    }
}

static int build_dictionary(Ctx *c, Mesh *m) {
    m->dict_count = 0;
    if (m->nnz == 0 || m->max_row > DICT_DLEN) return PGD_OK;
    void *p;
    unsigned long long *keys = nullptr;
    int *ibuf = nullptr;   // ids[DICT_TABLE], rep[DICT_MAXP], count, flags
    PGD_TRY(dev_alloc(c, &p, DICT_TABLE * sizeof(unsigned long long))); keys = (unsigned long long *)p;
    PGD_TRY(dev_alloc(c, &p, (DICT_TABLE + DICT_MAXP + 8) * sizeof(int))); ibuf = (int *)p;
    int *ids = ibuf, *rep = ibuf + DICT_TABLE, *count = rep + DICT_MAXP, *flags = count + 1;
    PGD_TRY(dev_alloc(c, &p, (size_t)m->nv * sizeof(uint16_t))); m->pids = (uint16_t *)p;
    PGD_TRY(dev_alloc(c, &p, (size_t)DICT_MAXP * DICT_DLEN * sizeof(int))); m->dict_off = (int *)p;
    hipStream_t st = c->stream;
    PGD_HIP(c, hipMemsetAsync(keys, 0, DICT_TABLE * sizeof(unsigned long long), st));
    PGD_HIP(c, hipMemsetAsync(ibuf, 0x7f, (DICT_TABLE + DICT_MAXP) * sizeof(int), st));   // rep = large
    PGD_HIP(c, hipMemsetAsync(count, 0, 8 * sizeof(int), st));
    PGD_HIP(c, hipMemsetAsync(m->dict_off, 0, (size_t)DICT_MAXP * DICT_DLEN * sizeof(int), st));
    const int g = grid_for(m->nv);
    k_dict_insert<<<g, TPB, 0, st>>>(m->row_ptr, m->cols, m->nv, keys, flags);
    k_dict_ids<<<DICT_TABLE / TPB, TPB, 0, st>>>(keys, ids, count);
    int h[2] = {0, 0};
    PGD_HIP(c, hipMemcpyAsync(h, count, sizeof h, hipMemcpyDeviceToHost, st));
    PGD_HIP(c, hipStreamSynchronize(st));
    bool ok = h[1] == 0 && h[0] >= 1 && h[0] <= DICT_MAXP;
    if (ok) {
        k_dict_assign<<<g, TPB, 0, st>>>(m->row_ptr, m->cols, m->nv, keys, ids, m->pids, rep, flags);
        k_dict_build<<<h[0], DICT_DLEN, 0, st>>>(m->row_ptr, m->cols, rep, h[0], m->dict_off);
        k_dict_verify<<<g, TPB, 0, st>>>(m->row_ptr, m->cols, m->nv, m->pids, m->dict_off, flags);
        int f = 1;
        PGD_HIP(c, hipMemcpyAsync(&f, flags, sizeof f, hipMemcpyDeviceToHost, st));
        PGD_HIP(c, hipStreamSynchronize(st));
        ok = f == 0;
    }
    (void)hipFree(keys);
    (void)hipFree(ibuf);
    if (ok) {
        m->dict_count = h[0];
    } else {
        (void)hipFree(m->pids); m->pids = nullptr;
        (void)hipFree(m->dict_off); m->dict_off = nullptr;
    }
    PGD_LAUNCH_CHECK(c);
    return PGD_OK;
}


// ---------------------------------------------------------------- blocked (vector-valued) layouts
// A vector-valued Lagrange space with NC components on a scalar layout: dof (node i, component c) has the
// number NC i + c, and dof row NC i + c couples to BOTH components of every node j of scalar row i - the
// scalar pattern with each entry widened to NC columns.  Entry k of scalar row i therefore sits at
// position NC k + cu of every dof row NC i + cv: embedding a scalar atom into a block needs no search.
__global__ __launch_bounds__(TPB) void k_block_rowptr(const int *__restrict__ rp, int64_t n, int nc, int *__restrict__ brp) {
    const int64_t r = (int64_t)blockIdx.x * TPB + threadIdx.x;     // dof row
    if (r > n * nc) return;
    if (r == n * nc) { brp[r] = rp[n] * nc * nc; return; }
    const int64_t i = r / nc, c = r % nc;
    brp[r] = (rp[i] * nc + (int)c * (rp[i + 1] - rp[i])) * nc;
}

__global__ __launch_bounds__(TPB) void k_block_cols(const int *__restrict__ rp, const int *__restrict__ cols, int64_t n,
                                                    int nc, const int *__restrict__ brp, int *__restrict__ bcols) {
    const int64_t r = (int64_t)blockIdx.x * TPB + threadIdx.x;
    if (r >= n * nc) return;
    const int64_t i = r / nc;
    const int a = rp[i], len = rp[i + 1] - a, base = brp[r];
    for (int k = 0; k < len; ++k)
        for (int c = 0; c < nc; ++c) bcols[base + k * nc + c] = cols[a + k] * nc + c;
}

// dst[(row NC i + cv), (col NC j + cu)] += coef * src[i, j]
__global__ __launch_bounds__(TPB) void k_block_embed(const int *__restrict__ rp, const double *__restrict__ src, int64_t n,
                                                     int nc, int cv, int cu, double coef, const int *__restrict__ brp,
                                                     double *__restrict__ dst) {
    const int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x;
    if (i >= n) return;
    const int a = rp[i], len = rp[i + 1] - a, base = brp[i * nc + cv];
    for (int k = 0; k < len; ++k) dst[base + k * nc + cu] += coef * src[a + k];
}

// tetrahedra arrive in their record layout: vertex ids checked on the device (flag + one offending id)
__global__ __launch_bounds__(TPB) void k_cells_validate(const int4 *__restrict__ cells, int64_t nc, int64_t nv, int *__restrict__ bad) {
    for (int64_t e = (int64_t)blockIdx.x * TPB + threadIdx.x; e < nc; e += (int64_t)gridDim.x * TPB) {
        const int4 u = cells[e];
        const int ids[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (ids[j] < 0 || ids[j] >= nv) { bad[0] = 1; bad[1] = ids[j]; }
    }
}

}  // namespace pgd

using namespace pgd;

extern "C" {

int pgd_mesh_upload(pgd_handle h, const double *coords, int64_t nv, int gdim, const int32_t *cells,
                    int64_t nc, int nvpc, pgd_handle *out) {
    PGD_CTX(c, h);
    // quadratic elements: cell record = (vertices, then edge nodes): 3 / 6 / 10 nodes in 1-D / 2-D / 3-D
    const bool p2 = gdim >= 1 && gdim <= 3 && nvpc == (gdim + 1) * (gdim + 2) / 2;
    if (!coords || !cells || !out || nv < 2 || nc < 1 || gdim < 1 || gdim > 3 || (nvpc != gdim + 1 && !p2))
        return fail(c, PGD_ERR_INVALID, "mesh_upload: need P1 simplices (nvpc == gdim + 1, gdim in 1..3) or P2 simplices (nvpc 3 / 6 / 10)");
    if (nv >= (int64_t)1 << 31 || nc * nvpc >= (int64_t)1 << 31)
        return fail(c, PGD_ERR_LIMIT, "mesh_upload: index range exceeds int32");
    // an out-of-range vertex id would fault on the device: connectivity is validated before any kernel follows it - on the host
    // for the small layouts that are repacked there anyway, by k_cells_validate for tetrahedra (12.6 M cells at 128^3: the host
    // loops over them cost 70 ms of every first solve)
    const bool tets = nvpc == 4;
    if (!tets)
        for (int64_t i = 0; i < nc * nvpc; ++i)
            if (cells[i] < 0 || cells[i] >= nv) return fail(c, PGD_ERR_INVALID, "mesh_upload: cell vertex id %d out of range", cells[i]);
    std::unique_ptr<Mesh> m(new Mesh);
    m->kind = Obj::MESH;
    m->gdim = gdim; m->nvpc = nvpc; m->nv = nv; m->nc = nc;
    void *p;
    PGD_TRY(dev_alloc(c, &p, (size_t)nv * 3 * sizeof(double))); m->coords = (double *)p;
    {   // host-side repack: AoS -> SoA coordinates
        std::vector<double> soa((size_t)nv * 3, 0.0);
        for (int64_t v = 0; v < nv; ++v)
            for (int k = 0; k < gdim; ++k) soa[(size_t)k * nv + v] = coords[v * gdim + k];
        PGD_HIP(c, hipMemcpyAsync(m->coords, soa.data(), soa.size() * sizeof(double), hipMemcpyHostToDevice, c->stream));
        PGD_HIP(c, hipStreamSynchronize(c->stream));
    }
    if (nvpc > 4) {   // P2 triangles / tetrahedra: flat records
        PGD_TRY(dev_alloc(c, &p, (size_t)nc * nvpc * sizeof(int))); m->cellsN = (int *)p;
        PGD_HIP(c, hipMemcpyAsync(m->cellsN, cells, (size_t)nc * nvpc * sizeof(int), hipMemcpyHostToDevice, c->stream));
        PGD_HIP(c, hipStreamSynchronize(c->stream));
    } else {          // one int4 record per cell
        PGD_TRY(dev_alloc(c, &p, (size_t)nc * sizeof(int4))); m->cells = (int4 *)p;
        if (tets) {   // the caller's array IS the record layout
            PGD_HIP(c, hipMemcpyAsync(m->cells, cells, (size_t)nc * sizeof(int4), hipMemcpyHostToDevice, c->stream));
            int *bad = nullptr;
            PGD_TRY(dev_alloc(c, &p, 64)); bad = (int *)p;
            PGD_HIP(c, hipMemsetAsync(bad, 0, 64, c->stream));
            k_cells_validate<<<grid_for(nc), TPB, 0, c->stream>>>(m->cells, nc, nv, bad);
            int hb[2] = {0, 0};
            PGD_HIP(c, hipMemcpyAsync(hb, bad, sizeof hb, hipMemcpyDeviceToHost, c->stream));
            PGD_HIP(c, hipStreamSynchronize(c->stream));
            (void)hipFree(bad);
            PGD_LAUNCH_CHECK(c);
            if (hb[0]) return fail(c, PGD_ERR_INVALID, "mesh_upload: cell vertex id %d out of range", hb[1]);
        } else {
            std::vector<int4> rec((size_t)nc);
            for (int64_t e = 0; e < nc; ++e) {
                int u[4] = {-1, -1, -1, -1};
                for (int j = 0; j < nvpc; ++j) u[j] = cells[e * nvpc + j];
                rec[e] = make_int4(u[0], u[1], u[2], u[3]);
            }
            PGD_HIP(c, hipMemcpyAsync(m->cells, rec.data(), rec.size() * sizeof(int4), hipMemcpyHostToDevice, c->stream));
            PGD_HIP(c, hipStreamSynchronize(c->stream));
        }
    }
    PGD_TRY(build_topology(c, m.get()));
    PGD_TRY(build_dictionary(c, m.get()));
    PGD_TRY(build_sym_tables(c, m.get()));
    PGD_TRY(detect_lattice(c, m.get()));
    *out = put_obj(c, m.release());
    return PGD_OK;
}

int pgd_mesh_lattice(pgd_handle h, pgd_handle mh, int32_t *is_lattice, double *steps) {
    PGD_CTX(c, h);
    Mesh *m = get_mesh(c, mh);
    if (!m) return fail(c, PGD_ERR_INVALID, "mesh_lattice: invalid handle");
    if (is_lattice) *is_lattice = m->lattice ? 1 : 0;
    if (steps) for (int k = 0; k < 3; ++k) steps[k] = m->lattice ? m->lat_h[k] : 0.0;
    return PGD_OK;
}

int pgd_mesh_info(pgd_handle h, pgd_handle mh, int64_t *nv, int64_t *nc, int64_t *nnz, int32_t *max_row,
                  int32_t *kl, int32_t *ku) {
    PGD_CTX(c, h);
    Mesh *m = get_mesh(c, mh);
    if (!m) return fail(c, PGD_ERR_INVALID, "mesh_info: invalid handle");
    if (nv) *nv = m->nv;
    if (nc) *nc = m->nc;
    if (nnz) *nnz = m->nnz;
    if (max_row) *max_row = m->max_row;
    if (kl) *kl = m->kl;
    if (ku) *ku = m->ku;
    return PGD_OK;
}

int pgd_mesh_pattern_download(pgd_handle h, pgd_handle mh, int32_t *row_ptr, int32_t *cols) {
    PGD_CTX(c, h);
    Mesh *m = get_mesh(c, mh);
    if (!m) return fail(c, PGD_ERR_INVALID, "mesh_pattern_download: invalid handle");
    if (row_ptr) PGD_HIP(c, hipMemcpyAsync(row_ptr, m->row_ptr, (size_t)(m->nv + 1) * sizeof(int), hipMemcpyDeviceToHost, c->stream));
    if (cols && m->nnz) PGD_HIP(c, hipMemcpyAsync(cols, m->cols, (size_t)m->nnz * sizeof(int), hipMemcpyDeviceToHost, c->stream));
    PGD_HIP(c, hipStreamSynchronize(c->stream));
    return PGD_OK;
}

int pgd_mesh_dict_count(pgd_handle h, pgd_handle mh, int32_t *count) {
    PGD_CTX(c, h);
    Mesh *m = get_mesh(c, mh);
    if (!m || !count) return fail(c, PGD_ERR_INVALID, "mesh_dict_count: invalid handle");
    *count = m->dict_count;
    return PGD_OK;
}

int pgd_mesh_sym_info(pgd_handle h, pgd_handle mh, int32_t *slots, int32_t *nx, int32_t *ny) {
    PGD_CTX(c, h);
    Mesh *m = get_mesh(c, mh);
    if (!m) return fail(c, PGD_ERR_INVALID, "mesh_sym_info: invalid handle");
    if (slots) *slots = m->sym_w;
    if (nx) *nx = m->sym_nx;
    if (ny) *ny = m->sym_ny;
    return PGD_OK;
}

int pgd_mesh_free(pgd_handle h, pgd_handle mh) {
    PGD_CTX(c, h);
    return free_obj(c, mh, Obj::MESH);
}

static int new_csr(Ctx *c, pgd_handle mh, Mesh *m, pgd_handle *out, Csr **res) {
    std::unique_ptr<Csr> a(new Csr);
    a->kind = Obj::CSR;
    a->mesh = mh;
    void *p;
    a->vals_bytes = (size_t)(m->nnz > 0 ? m->nnz : 1) * sizeof(double);
    PGD_TRY(dev_alloc(c, &p, a->vals_bytes));
    a->vals = (double *)p;
    PGD_HIP(c, hipMemsetAsync(a->vals, 0, (size_t)(m->nnz > 0 ? m->nnz : 1) * sizeof(double) + PAD_BYTES, c->stream));
    a->immutable = true;       // an atom: written once, here or by its caller (pgd_atom_embed(dst) accumulates and says so)
    *res = a.get();
    *out = put_obj(c, a.release());
    return PGD_OK;
}

int pgd_atom_assemble(pgd_handle h, pgd_handle mh, int kind, int da, int db, pgd_handle wh, pgd_handle *out) {
    PGD_CTX(c, h);
    Mesh *m = get_mesh(c, mh);
    if (!m || !out) return fail(c, PGD_ERR_INVALID, "atom_assemble: invalid mesh handle");
    if (m->ncomp != 1) return fail(c, PGD_ERR_INVALID, "atom_assemble: blocked layouts take their atoms from pgd_atom_embed");
    if (kind < PGD_ATOM_MASS || kind > PGD_ATOM_WSTIFF) return fail(c, PGD_ERR_INVALID, "atom_assemble: unknown kind %d", kind);
    if (da < 0 || da >= m->gdim || db < 0 || db >= m->gdim) return fail(c, PGD_ERR_INVALID, "atom_assemble: derivative axis out of range");
    const double *w = nullptr;
    if (kind == PGD_ATOM_WMASS || kind == PGD_ATOM_WSTIFF) {
        Vec *wv = get_vec(c, wh);
        if (!wv || wv->n != m->nv) return fail(c, PGD_ERR_INVALID, "atom_assemble: weighted kind needs a vertex weight vector");
        w = wv->d;
    }
    Csr *a = nullptr;
    PGD_TRY(new_csr(c, mh, m, out, &a));
    AsmArgs A;
    A.cx = m->coords; A.cy = m->coords + m->nv; A.cz = m->coords + 2 * m->nv;
    A.cells = m->cells; A.v2c_ptr = m->v2c_ptr; A.v2c = m->v2c; A.row_ptr = m->row_ptr; A.cols = m->cols;
    A.w = w; A.vals = a->vals; A.nv = m->nv; A.kind = kind; A.da = da; A.db = db;
    A.lattice = (m->lattice && c->asm_lattice) ? 1 : 0;
    for (int k = 0; k < 3; ++k) { A.lat_h[k] = m->lat_h[k]; A.lat_inv[k] = m->lattice ? 1.0 / m->lat_h[k] : 0.0; }
    // (PGD_TUNE_ASM_LATTICE = 2: steps from the coordinates, the r03 form; 3: steps from the indices, in the general kernel)
    A.lat_unit = (A.lattice && m->lattice_unit && (c->asm_lattice == 1 || c->asm_lattice == 3)) ? 1 : 0;
    A.nx = m->sym_nx; A.ny = m->sym_ny;
    const int gb = (int)((m->nv + TPB - 1) / TPB);
    if (m->cellsN && m->gdim == 2) k_assemble_p2_simplex<2><<<(int)((m->nv + 63) / 64), 64, 0, c->stream>>>(A, m->cellsN);
    else if (m->cellsN) k_assemble_p2_simplex<3><<<(int)((m->nv + 63) / 64), 64, 0, c->stream>>>(A, m->cellsN);
    else if (m->gdim == 1 && m->nvpc == 3) k_assemble_p2_interval<<<gb, TPB, 0, c->stream>>>(A);
    else if (m->gdim == 1) k_assemble_p1<1><<<gb, TPB, 0, c->stream>>>(A);
    else if (m->gdim == 2) k_assemble_p1<2><<<gb, TPB, 0, c->stream>>>(A);
    else if (A.lat_unit && m->lattice_regular && !w && c->asm_lattice == 1 && m->max_row <= 15)
        k_assemble_p1_regular<<<gb, TPB, 0, c->stream>>>(A, (int)(m->nv / ((int64_t)m->sym_nx * m->sym_ny)));
    else k_assemble_p1<3><<<gb, TPB, 0, c->stream>>>(A);
    PGD_LAUNCH_CHECK(c);
    return PGD_OK;
}

int pgd_mesh_blocked(pgd_handle h, pgd_handle mh, int ncomp, pgd_handle *out) {
    PGD_CTX(c, h);
    Mesh *m = get_mesh(c, mh);
    if (!m || !out || ncomp < 2 || ncomp > 3) return fail(c, PGD_ERR_INVALID, "mesh_blocked: need a scalar layout and 2 or 3 components");
    if (m->ncomp != 1) return fail(c, PGD_ERR_INVALID, "mesh_blocked: the base layout must be scalar");
    if ((int64_t)m->nnz * ncomp * ncomp >= (int64_t)1 << 31 || m->nv * ncomp >= (int64_t)1 << 31)
        return fail(c, PGD_ERR_LIMIT, "mesh_blocked: index range exceeds int32");
    std::unique_ptr<Mesh> b(new Mesh);
    b->kind = Obj::MESH;
    b->gdim = m->gdim; b->nvpc = m->nvpc; b->ncomp = ncomp; b->base = mh;
    b->nv = m->nv * ncomp; b->nc = m->nc; b->nnz = m->nnz * ncomp * ncomp;
    b->max_row = m->max_row * ncomp;
    b->kl = m->kl * ncomp + ncomp - 1; b->ku = m->ku * ncomp + ncomp - 1;
    void *p;
    PGD_TRY(dev_alloc(c, &p, (size_t)(b->nv + 1) * sizeof(int))); b->row_ptr = (int *)p;
    PGD_TRY(dev_alloc(c, &p, (size_t)(b->nnz > 0 ? b->nnz : 1) * sizeof(int))); b->cols = (int *)p;
    PGD_HIP(c, hipMemsetAsync(b->cols, 0, (size_t)(b->nnz > 0 ? b->nnz : 1) * sizeof(int) + PAD_BYTES, c->stream));
    k_block_rowptr<<<(int)((b->nv + TPB) / TPB), TPB, 0, c->stream>>>(m->row_ptr, m->nv, ncomp, b->row_ptr);
    k_block_cols<<<(int)((b->nv + TPB - 1) / TPB), TPB, 0, c->stream>>>(m->row_ptr, m->cols, m->nv, ncomp, b->row_ptr, b->cols);
    PGD_LAUNCH_CHECK(c);
    PGD_TRY(build_dictionary(c, b.get()));
    PGD_TRY(build_sym_tables(c, b.get()));
    *out = put_obj(c, b.release());
    return PGD_OK;
}

int pgd_atom_embed(pgd_handle h, pgd_handle bmh, pgd_handle src, int cv, int cu, double coef, pgd_handle dst, pgd_handle *out) {
    PGD_CTX(c, h);
    Mesh *b = get_mesh(c, bmh);
    Csr *a = get_csr(c, src);
    Mesh *m = a ? get_mesh(c, a->mesh) : nullptr;
    if (!b || !a || !m || !out || b->ncomp < 2 || b->base != a->mesh)
        return fail(c, PGD_ERR_INVALID, "atom_embed: need a blocked layout and an atom of its scalar base layout");
    if (cv < 0 || cv >= b->ncomp || cu < 0 || cu >= b->ncomp) return fail(c, PGD_ERR_INVALID, "atom_embed: component out of range");
    Csr *d = nullptr;
    if (dst) {
        d = get_csr(c, dst);
        if (!d || d->mesh != bmh) return fail(c, PGD_ERR_INVALID, "atom_embed: destination is not an atom of the blocked layout");
        *out = dst;
    } else {
        PGD_TRY(new_csr(c, bmh, b, out, &d));      // zero-filled
    }
    PGD_TRY(ensure_vals(c, m, a));
    d->version += 1;           // new values: forms derived from the old ones are gone
    d->uvals_valid = false; d->uvals_scaled = false; d->cls_count = 0; d->cls_tried = false; d->dinv_valid = false;
    k_block_embed<<<(int)((m->nv + TPB - 1) / TPB), TPB, 0, c->stream>>>(m->row_ptr, a->vals, m->nv, b->ncomp, cv, cu, coef, b->row_ptr, d->vals);
    PGD_LAUNCH_CHECK(c);
    return PGD_OK;
}

int pgd_atom_upload(pgd_handle h, pgd_handle mh, const double *vals, pgd_handle *out) {
    PGD_CTX(c, h);
    Mesh *m = get_mesh(c, mh);
    if (!m || !out || !vals) return fail(c, PGD_ERR_INVALID, "atom_upload: invalid arguments");
    Csr *a = nullptr;
    PGD_TRY(new_csr(c, mh, m, out, &a));
    if (m->nnz) PGD_HIP(c, hipMemcpyAsync(a->vals, vals, (size_t)m->nnz * sizeof(double), hipMemcpyHostToDevice, c->stream));
    PGD_HIP(c, hipStreamSynchronize(c->stream));
    return PGD_OK;
}

int pgd_atom_download(pgd_handle h, pgd_handle ah, double *vals) {
    PGD_CTX(c, h);
    Csr *a = get_csr(c, ah);
    Mesh *m = a ? get_mesh(c, a->mesh) : nullptr;
    if (!a || !m || !vals) return fail(c, PGD_ERR_INVALID, "atom_download: invalid handle");
    PGD_TRY(ensure_vals(c, m, a));
    if (m->nnz) PGD_HIP(c, hipMemcpyAsync(vals, a->vals, (size_t)m->nnz * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    PGD_HIP(c, hipStreamSynchronize(c->stream));
    return PGD_OK;
}

int pgd_atom_free(pgd_handle h, pgd_handle ah) {
    PGD_CTX(c, h);
    return free_obj(c, ah, Obj::CSR);
}

}  // extern "C"
