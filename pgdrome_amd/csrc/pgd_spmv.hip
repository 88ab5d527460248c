// k_spmv_csr - the gated kernel: y = A x for fp64 CSR with int32 columns.
//
// Roofline: HBM.  Algorithmic bytes per launch = 12 nnz + 20 n (values + column
// ids, row_ptr, one read of x, one write of y).
//
// FEM rows are short (3 / 7 / 15 entries), so lane-per-row straight from global
// memory strides by 180 B and wave-per-row wastes 49 lanes.  Instead a
// workgroup owns 256 consecutive rows whose values and column ids form ONE
// contiguous CSR segment: that segment is streamed into LDS with 16-byte-per-
// lane loads (every byte of every fetched line is used), all loads of the
// segment are in flight before the first LDS store, and then each lane walks
// its own row out of LDS.  Consecutive rows of a structured-order FEM matrix
// read consecutive x entries for the same entry index k, so the x gather is
// coalesced too; x is served from L2 / Infinity Cache (3 grid planes live).
// Row strides of 15, 7 and 3 entries are odd, so the LDS reads are conflict-free
// (ds_read_b64 banks (2 i stride) mod 64, ds_read_b32 banks (i stride) mod 32).
//
// The logical row-block index is remapped so each XCD walks a contiguous range
// of row blocks (neighbouring blocks share x planes in that XCD's L2).
#include <chrono>
#include <cstring>

#include "pgd_internal.h"

#include <algorithm>

namespace pgd {

constexpr int SPMV_CAP = 4096;                       // staged entries per 256-row workgroup (k_spmv_multi)
constexpr int SPMV_VROUNDS = 8;                      // double2 loads per lane: 16 entries per row
constexpr int SPMV_CROUNDS = 4;                      // int4 loads per lane
constexpr int MAXY = 8;                              // vectors per pass of k_spmv_multi

struct SpmvArgs {
    const int *row_ptr, *cols;
    const double *vals, *x, *w;
    double *y, *partials;
    const int *flags;
    int row_begin, row_end;
};

typedef double d2_t __attribute__((ext_vector_type(2)));
typedef int i4_t __attribute__((ext_vector_type(4)));

// R = rows (= threads) per workgroup; the staging capacity scales with it (16 entries per row).
template <bool DOT, bool STORE, int R>
__global__ __launch_bounds__(R) void k_spmv_csr(SpmvArgs A) {
    constexpr int CAP = 16 * R;
    if (A.flags && A.flags[0]) return;   // PCG already converged: uniform early exit
    __shared__ __align__(16) double s_vals[CAP + 2];
    __shared__ __align__(16) int s_cols[CAP + 4];
    __shared__ int s_rp[R + 1];
    __shared__ double s_red[R / 64];
    const int tid = threadIdx.x;
    const int b = xcd_remap(blockIdx.x, gridDim.x);
    const int r0 = A.row_begin + b * R;
    const int nr = min(R, A.row_end - r0);
    if (tid < nr) s_rp[tid] = A.row_ptr[r0 + tid];
    if (tid == 0) s_rp[nr] = A.row_ptr[r0 + nr];
    __syncthreads();
    const int s = s_rp[0], e = s_rp[nr];
    const int sv = s & ~1, sc = s & ~3;     // 16-byte aligned starts of the two streams
    double acc = 0.0;
    if (e - sc <= CAP) {
        d2_t v[SPMV_VROUNDS];
        i4_t c[SPMV_CROUNDS];
        const double *gv = A.vals + sv;
        const int *gc = A.cols + sc;
        const int nvv = e - sv, ncc = e - sc;
        // every lane always loads (out-of-range lanes re-read offset 0): no predicated
        // register writes, so the staging registers stay in VGPRs and all 12 loads of
        // the segment are in flight before the first LDS store
#pragma unroll
        for (int i = 0; i < SPMV_VROUNDS; ++i) {
            const int k = (tid + i * R) * 2;
            v[i] = *reinterpret_cast<const d2_t *>(gv + (k < nvv ? k : 0));
        }
#pragma unroll
        for (int i = 0; i < SPMV_CROUNDS; ++i) {
            const int k = (tid + i * R) * 4;
            c[i] = *reinterpret_cast<const i4_t *>(gc + (k < ncc ? k : 0));
        }
        // stores are unconditional as well (slots past the segment are never read): the
        // staging phase is straight-line code, 12 loads then 12 LDS stores
#pragma unroll
        for (int i = 0; i < SPMV_VROUNDS; ++i)
            *reinterpret_cast<d2_t *>(s_vals + (tid + i * R) * 2) = v[i];
#pragma unroll
        for (int i = 0; i < SPMV_CROUNDS; ++i)
            *reinterpret_cast<i4_t *>(s_cols + (tid + i * R) * 4) = c[i];
        __syncthreads();
        if (tid < nr) {
            const int a = s_rp[tid], bnd = s_rp[tid + 1];
            for (int k = a; k < bnd; k += 8) {
                // branch-free chunk of 8 entries: clamp the LDS index, zero the value
                double vv[8], xv[8];
                int cc[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int kk = (k + u < bnd) ? k + u : bnd - 1;
                    cc[u] = s_cols[kk - sc];
                    const double t = s_vals[kk - sv];
                    vv[u] = (k + u < bnd) ? t : 0.0;
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) xv[u] = A.x[cc[u]];
#pragma unroll
                for (int u = 0; u < 8; ++u) acc = fma(vv[u], xv[u], acc);
            }
        }
    } else if (tid < nr) {
        // segment does not fit the staging buffers (long rows): straight from global
        for (int k = s_rp[tid]; k < s_rp[tid + 1]; ++k) acc = fma(A.vals[k], A.x[A.cols[k]], acc);
    }
    if (STORE && tid < nr) A.y[r0 + tid] = acc;
    if (DOT) {
        const double t = (tid < nr) ? acc * A.w[r0 + tid] : 0.0;
        const double sum = block_sum_n<R / 64>(t, s_red);
        if (tid == 0) A.partials[b] = sum;
    }
}

// Column-dictionary form of the same product (Mesh::pids / dict_off, pgd_internal.h): the
// values are staged exactly as above, but no column id is read from HBM - a row decodes its
// columns as row + dict_off[pattern][k].  A wave of interior rows shares one pattern, so the
// table reads are L1 broadcasts.  Traffic per row drops from 180 + 20 B to 120 + 22 B; the
// result is bit-identical to k_spmv_csr (same products, same order).
template <bool DOT, bool STORE, int R>
__global__ __launch_bounds__(R) void k_spmv_csr_dict(SpmvArgs A, const uint16_t *__restrict__ pids,
                                                      const int *__restrict__ dict_off) {
    constexpr int CAP = 16 * R;
    if (A.flags && A.flags[0]) return;
    __shared__ __align__(16) double s_vals[CAP + 2];
    __shared__ int s_rp[R + 1];
    __shared__ double s_red[R / 64];
    const int tid = threadIdx.x;
    const int b = xcd_remap(blockIdx.x, gridDim.x);
    const int r0 = A.row_begin + b * R;
    const int nr = min(R, A.row_end - r0);
    if (tid < nr) s_rp[tid] = A.row_ptr[r0 + tid];
    if (tid == 0) s_rp[nr] = A.row_ptr[r0 + nr];
    const int pid = (tid < nr) ? (int)pids[r0 + tid] : 0;
    __syncthreads();
    const int s = s_rp[0], e = s_rp[nr];
    const int sv = s & ~1;
    const bool staged = (e - sv) <= CAP;
    if (staged) {
        d2_t v[SPMV_VROUNDS];
        const double *gv = A.vals + sv;
        const int nvv = e - sv;
#pragma unroll
        for (int i = 0; i < SPMV_VROUNDS; ++i) {
            const int k = (tid + i * R) * 2;
            v[i] = *reinterpret_cast<const d2_t *>(gv + (k < nvv ? k : 0));
        }
#pragma unroll
        for (int i = 0; i < SPMV_VROUNDS; ++i) *reinterpret_cast<d2_t *>(s_vals + (tid + i * R) * 2) = v[i];
    }
    __syncthreads();
    double acc = 0.0;
    if (tid < nr) {
        const int a = s_rp[tid], len = s_rp[tid + 1] - a, r = r0 + tid;
        const int *off = dict_off + pid * DICT_DLEN;
        for (int k0 = 0; k0 < len; k0 += 8) {
            // 8 entries per round: two 16-byte table reads, 8 gathers in flight
            const i4_t o0 = *reinterpret_cast<const i4_t *>(off + k0);
            const i4_t o1 = *reinterpret_cast<const i4_t *>(off + k0 + 4);
            const int oo[8] = {o0.x, o0.y, o0.z, o0.w, o1.x, o1.y, o1.z, o1.w};
            double vv[8], xv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const bool ok = k0 + u < len;
                const int kk = ok ? k0 + u : len - 1;
                const double tv = staged ? s_vals[a + kk - sv] : A.vals[a + kk];
                vv[u] = ok ? tv : 0.0;
                xv[u] = A.x[r + (ok ? oo[u] : 0)];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) acc = fma(vv[u], xv[u], acc);
        }
    }
    if (STORE && tid < nr) A.y[r0 + tid] = acc;
    if (DOT) {
        const double t = (tid < nr) ? acc * A.w[r0 + tid] : 0.0;
        const double sum = block_sum_n<R / 64>(t, s_red);
        if (tid == 0) A.partials[b] = sum;
    }
}

// Rows of at most 16 entries (every P1 mesh): the x gathers need only the pattern id, not the staged
// values, so both memory phases are put in flight together - 8 value loads from HBM and 16 gathers
// from L1/L2 per lane - and the wave waits once.  Segment bounds come from two uniform loads, so the
// value stream starts without waiting for the per-lane row pointers.
template <bool DOT, bool STORE>
__global__ __launch_bounds__(64) void k_spmv_csr_dict16(SpmvArgs A, const uint16_t *__restrict__ pids,
                                                        const int *__restrict__ dict_off) {
    constexpr int R = 64, CAP = 16 * R;
    if (A.flags && A.flags[0]) return;
    __shared__ __align__(16) double s_vals[CAP + 2];
    const int tid = threadIdx.x;
    const int b = xcd_remap(blockIdx.x, gridDim.x);
    const int r0 = A.row_begin + b * R;
    const int nr = min(R, A.row_end - r0);
    const int s = A.row_ptr[r0], e = A.row_ptr[r0 + nr];          // uniform
    const int sv = s & ~1, nvv = e - sv;
    d2_t v[SPMV_VROUNDS];
    const double *gv = A.vals + sv;
#pragma unroll
    for (int i = 0; i < SPMV_VROUNDS; ++i) {
        const int k = (tid + i * R) * 2;
        v[i] = *reinterpret_cast<const d2_t *>(gv + (k < nvv ? k : 0));
    }
    const int row = r0 + (tid < nr ? tid : 0);
    const int a = A.row_ptr[row], len = (tid < nr) ? A.row_ptr[row + 1] - a : 0;
    const int *off = dict_off + (int)pids[row] * DICT_DLEN;
    int oo[16];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const i4_t o = *reinterpret_cast<const i4_t *>(off + 4 * q);
        oo[4 * q] = o.x; oo[4 * q + 1] = o.y; oo[4 * q + 2] = o.z; oo[4 * q + 3] = o.w;
    }
    double xv[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) xv[k] = A.x[row + (k < len ? oo[k] : 0)];
#pragma unroll
    for (int i = 0; i < SPMV_VROUNDS; ++i) *reinterpret_cast<d2_t *>(s_vals + (tid + i * R) * 2) = v[i];
    __syncthreads();
    double acc = 0.0;
    const int base = a - sv;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const double vk = s_vals[base + (k < len ? k : 0)];
        acc = fma(k < len ? vk : 0.0, xv[k], acc);
    }
    if (STORE && tid < nr) A.y[r0 + tid] = acc;
    if (DOT) {
        const double t = (tid < nr) ? acc * A.w[r0 + tid] : 0.0;
        const double sum = wave_sum(t);
        if (tid == 0) A.partials[b] = sum;
    }
}

// out partial[b*ny + m] = sum over the block's rows of x_i (A y_m)_i : one pass
// over the matrix for up to MAXY stored modes (batched scalar functionals).
struct SpmvMultiArgs {
    const int *row_ptr, *cols;
    const double *vals, *x;
    const double *ys[MAXY];
    double *partials;
    int row_begin, row_end, ny;
};

__global__ __launch_bounds__(TPB) void k_spmv_multi(SpmvMultiArgs A) {
    __shared__ __align__(16) double s_vals[SPMV_CAP + 2];
    __shared__ __align__(16) int s_cols[SPMV_CAP + 4];
    __shared__ int s_rp[TPB + 1];
    __shared__ double s_red[4];
    const int tid = threadIdx.x;
    const int b = xcd_remap(blockIdx.x, gridDim.x);
    const int r0 = A.row_begin + b * TPB;
    const int nr = min(TPB, A.row_end - r0);
    if (tid < nr) s_rp[tid] = A.row_ptr[r0 + tid];
    if (tid == 0) s_rp[nr] = A.row_ptr[r0 + nr];
    __syncthreads();
    const int s = s_rp[0], e = s_rp[nr];
    const int sv = s & ~1, sc = s & ~3;
    const bool staged = (e - sc) <= SPMV_CAP;
    if (staged) {
        for (int k = tid * 2; k < e - sv; k += TPB * 2)
            *reinterpret_cast<double2 *>(s_vals + k) = *reinterpret_cast<const double2 *>(A.vals + sv + k);
        for (int k = tid * 4; k < e - sc; k += TPB * 4)
            *reinterpret_cast<int4 *>(s_cols + k) = *reinterpret_cast<const int4 *>(A.cols + sc + k);
    }
    __syncthreads();
    double acc[MAXY];
#pragma unroll
    for (int m = 0; m < MAXY; ++m) acc[m] = 0.0;
    if (tid < nr) {
        const int a = s_rp[tid], bnd = s_rp[tid + 1];
        for (int k = a; k < bnd; ++k) {
            const double v = staged ? s_vals[k - sv] : A.vals[k];
            const int cc = staged ? s_cols[k - sc] : A.cols[k];
#pragma unroll
            for (int m = 0; m < MAXY; ++m)
                if (m < A.ny) acc[m] = fma(v, A.ys[m][cc], acc[m]);
        }
    }
    const double xi = (tid < nr) ? A.x[r0 + tid] : 0.0;
#pragma unroll
    for (int m = 0; m < MAXY; ++m) {
        if (m < A.ny) {
            const double sum = block_sum(acc[m] * xi, s_red);
            if (tid == 0) A.partials[(int64_t)b * A.ny + m] = sum;
        }
    }
}

// HIP-event timing of one launch in four (the events perturb the stream by ~5 us each side); the byte count is
// the CSR formula of SURVEY 8d for the rows covered, whatever internal form of the operator the kernel reads
static int prof_begin(Ctx *c, bool dot, bool store, bool *timed) {
    const bool candidate = c->prof && (!c->prof_pcg_only || (dot && store));
    *timed = candidate && ((c->prof_seen++ & 3) == 0);
    if (*timed) {
        if (c->ev_used + 2 > c->ev.size()) prof_flush(c);
        c->ev_rec[c->ev_used / 2] = Ctx::ProfRec{0, c->prof_iter, 0.0, 0.0, 0.0};
        PGD_HIP(c, hipEventRecord(c->ev[c->ev_used], c->stream));
    }
    return PGD_OK;
}

// own_per_row > 0: bytes per row the kernel that ran must move at least (its own storage form); < 0: -(bytes per
// stored entry) of a CSR form, plus row pointer / pattern id, x and y per row
static int prof_end(Ctx *c, const Mesh *m, int64_t nrows, double own_per_row) {
    PGD_HIP(c, hipEventRecord(c->ev[c->ev_used + 1], c->stream));
    Ctx::ProfRec &rec = c->ev_rec[c->ev_used / 2];
    c->ev_used += 2;
    const double frac = m->nv > 0 ? (double)nrows / (double)m->nv : 0.0;
    rec.bytes = 12.0 * (double)m->nnz * frac + 20.0 * (double)nrows;
    rec.own = own_per_row > 0 ? own_per_row * (double)nrows
                              : -own_per_row * (double)m->nnz * frac + (own_per_row < -10.0 ? 20.0 : 22.0) * (double)nrows;
    return PGD_OK;
}

int launch_spmv(Ctx *c, const Mesh *m, const double *vals, const double *x, double *y, const double *w,
                int64_t r0, int64_t r1, bool dot, bool store, const int *flags, int *nparts_out) {
    if (r1 < 0) r1 = m->nv;
    if (r0 < 0 || r0 > r1 || r1 > m->nv) return fail(c, PGD_ERR_INVALID, "spmv: bad row range");
    const int64_t nrows = r1 - r0;
    const int R = c->spmv_rows;
    const int nblk = (int)((nrows + R - 1) / R);
    if (nparts_out) *nparts_out = nblk;
    if (nblk == 0) return PGD_OK;
    if (dot) PGD_TRY(ensure_partials(c, std::max<int64_t>(c->partials_off + nblk, 4 * MAX_VEC_BLOCKS)));
    SpmvArgs A;
    A.row_ptr = m->row_ptr; A.cols = m->cols; A.vals = vals; A.x = x; A.w = w; A.y = y;
    A.partials = c->partials + c->partials_off; A.flags = flags; A.row_begin = (int)r0; A.row_end = (int)r1;
    bool timed = false;
    PGD_TRY(prof_begin(c, dot, store, &timed));
    const bool use_dict = c->spmv_dict && m->dict_count > 0;
#define PGD_SPMV_LAUNCH(D, S)                                                                             \
    do {                                                                                                  \
        if (use_dict && R == 64 && m->max_row <= 16 && c->spmv_dict == 1) {                               \
            k_spmv_csr_dict16<D, S><<<nblk, 64, 0, c->stream>>>(A, m->pids, m->dict_off);                 \
        } else if (use_dict) {                                                                            \
            if (R == 256) k_spmv_csr_dict<D, S, 256><<<nblk, 256, 0, c->stream>>>(A, m->pids, m->dict_off);      \
            else if (R == 128) k_spmv_csr_dict<D, S, 128><<<nblk, 128, 0, c->stream>>>(A, m->pids, m->dict_off); \
            else k_spmv_csr_dict<D, S, 64><<<nblk, 64, 0, c->stream>>>(A, m->pids, m->dict_off);                \
        } else {                                                                                          \
            if (R == 256) k_spmv_csr<D, S, 256><<<nblk, 256, 0, c->stream>>>(A);                          \
            else if (R == 128) k_spmv_csr<D, S, 128><<<nblk, 128, 0, c->stream>>>(A);                     \
            else k_spmv_csr<D, S, 64><<<nblk, 64, 0, c->stream>>>(A);                                     \
        }                                                                                                 \
    } while (0)
    if (dot && store) PGD_SPMV_LAUNCH(true, true);
    else if (dot) PGD_SPMV_LAUNCH(true, false);
    else PGD_SPMV_LAUNCH(false, true);
#undef PGD_SPMV_LAUNCH
    c->kcount[use_dict ? KC_CSR_DICT : KC_CSR] += 1;
    if (timed) PGD_TRY(prof_end(c, m, nrows, -(use_dict ? 8.0 : 12.0)));
    PGD_LAUNCH_CHECK(c);
    return PGD_OK;
}

// ------------------------------------------------------------------ symmetric half storage
// An SPD finite-element operator is stored ONCE per unordered pair: per row the diagonal and the entries
// right of it, slot by slot in sym_w arrays of n doubles (ELL, structure of arrays).  Row i needs a_ij for
// its lower neighbours j < i too: those sit in row j's slots, and for a mesh whose rows repeat a few relative
// patterns the slot is a function of row i's pattern - so lane i READS uvals[slot * n + (i - d)], a shifted,
// perfectly coalesced stream like every other access of this kernel (own slots, x at i + offset).  No scatter,
// no atomics, fixed summation order.  Every value leaves HBM once (8 (nnz + n) / 2 bytes instead of 12 nnz); the
// second use, at most one grid plane (~4 MB of values) later, is served by the Infinity Cache.
struct SymArgs {
    const double *uvals, *x, *w;
    double *y, *partials;
    const int *flags, *tab;
    const uint16_t *pids;
    int64_t n;              // slot stride in doubles (rows + padding)
    int row_begin, row_end;
    int qq;                 // DOT launches: partial sums in pairs (w . y, y . y) per workgroup (single-sync recurrence)
};

struct SymRec { int v[16]; };

__device__ __forceinline__ SymRec sym_load(const int *tab, int pid) {
    SymRec r;
    const i4_t *q = reinterpret_cast<const i4_t *>(tab + pid * 16);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const i4_t o = q[k];
        r.v[4 * k] = o.x; r.v[4 * k + 1] = o.y; r.v[4 * k + 2] = o.z; r.v[4 * k + 3] = o.w;
    }
    return r;
}

// (ulen incl. diagonal, llen, upper offsets, lower distances and slots) of one CSR row; false when the row
// does not fit the record (no diagonal, more than 7 + 8 off-diagonal entries, a slot beyond 7)
__device__ bool sym_describe(const int *__restrict__ row_ptr, const int *__restrict__ cols, int r, int *rec) {
    for (int k = 0; k < 16; ++k) rec[k] = 0;
    const int a = row_ptr[r], b = row_ptr[r + 1];
    int nl = 0, nu = 0, slots = 0;
    bool diag = false;
    for (int k = a; k < b; ++k) {
        const int col = cols[k];
        if (col < r) {
            int posd = -1, posr = -1;
            for (int t = row_ptr[col]; t < row_ptr[col + 1]; ++t) {
                if (cols[t] == col) posd = t;
                if (cols[t] == r) posr = t;
            }
            const int slot = posr - posd;
            if (posd < 0 || posr < 0 || slot < 1 || slot > 7 || nl >= 8) return false;
            rec[8 + nl] = r - col;
            slots |= slot << (3 * nl);
            ++nl;
        } else if (col == r) {
            diag = true;
        } else {
            if (nu >= 7) return false;
            ++nu;
            rec[nu] = col - r;
        }
    }
    if (!diag) return false;
    rec[0] = (nu + 1) | (nl << 4) | (slots << 8);
    return true;
}

__global__ __launch_bounds__(TPB) void k_sym_rep(const uint16_t *__restrict__ pids, int64_t n, int *__restrict__ rep) {
    const int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x;
    if (i < n && (int)i < rep[pids[i]]) atomicMin(&rep[pids[i]], (int)i);     // almost every row loses the first test
}

__global__ void k_sym_build(const int *__restrict__ row_ptr, const int *__restrict__ cols, const int *__restrict__ rep,
                            int npat, int64_t nv, int *__restrict__ tab, int *__restrict__ flags) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= npat) return;
    int rec[16];
    if (rep[p] < 0 || rep[p] >= nv || !sym_describe(row_ptr, cols, rep[p], rec)) { flags[0] = 1; return; }
    for (int k = 0; k < 16; ++k) tab[p * 16 + k] = rec[k];
}

__global__ __launch_bounds__(TPB) void k_sym_verify(const int *__restrict__ row_ptr, const int *__restrict__ cols,
                                                    const uint16_t *__restrict__ pids, const int *__restrict__ tab,
                                                    int64_t n, int *__restrict__ flags) {
    const int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x;
    if (i >= n) return;
    int rec[16];
    bool ok = sym_describe(row_ptr, cols, (int)i, rec);
    const int *t = tab + (int)pids[i] * 16;
    for (int k = 0; ok && k < 16; ++k) ok = rec[k] == t[k];
    if (!ok) flags[0] = 1;
}

// CSR values -> the slot arrays; also checks a_ij == a_ji to rounding (flags[1] counts violations)
template <int W>
__global__ __launch_bounds__(TPB) void k_csr_to_sym(const int *__restrict__ row_ptr, const double *__restrict__ vals,
                                                    const uint16_t *__restrict__ pids, const int *__restrict__ tab,
                                                    int64_t n, int64_t stride, double *__restrict__ uvals,
                                                    int *__restrict__ flags) {
    const int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x;
    if (i >= n) return;
    const int *t = tab + (int)pids[i] * 16;
    const int hdr = t[0], ulen = hdr & 15, llen = (hdr >> 4) & 15;
    const int a = row_ptr[i];
#pragma unroll
    for (int s = 0; s < W; ++s) uvals[(int64_t)s * stride + i] = s < ulen ? vals[a + llen + s] : 0.0;
    bool asym = false;
    for (int m = 0; m < llen; ++m) {
        const int64_t src = i - t[8 + m];
        const int slot = (hdr >> (8 + 3 * m)) & 7;
        const int ls = (tab[(int)pids[src] * 16] >> 4) & 15;
        const double aij = vals[a + m], aji = vals[row_ptr[src] + ls + slot];
        if (fabs(aij - aji) > 1e-12 * (fabs(aij) + fabs(aji))) asym = true;
    }
    if (asym) atomicAdd(&flags[1], 1);
}

// One row of the product from the slot arrays: 2 W value loads, 2 W x loads, all issued before the first use.
template <int W>
__device__ __forceinline__ double sym_row_rec(const SymArgs &A, int64_t row, const SymRec &t) {
    const int hdr = t.v[0], ulen = hdr & 15, llen = (hdr >> 4) & 15;
    double uv[W], ux[W], lv[W], lx[W];
#pragma unroll
    for (int s = 0; s < W; ++s) {
        uv[s] = A.uvals[(int64_t)s * A.n + row];                      // zero beyond ulen
        ux[s] = A.x[row + ((s > 0 && s < ulen) ? t.v[s] : 0)];
    }
#pragma unroll
    for (int m = 0; m < W; ++m) {
        const bool on = m < llen;
        const int64_t src = row - (on ? t.v[8 + m] : 0);
        const int slot = on ? (hdr >> (8 + 3 * m)) & 7 : 0;
        lv[m] = A.uvals[(int64_t)slot * A.n + src];
        lx[m] = A.x[src];
    }
    double acc = 0.0;
#pragma unroll
    for (int m = 0; m < W; ++m) acc = fma(m < llen ? lv[m] : 0.0, lx[m], acc);     // ascending columns: lower first
#pragma unroll
    for (int s = 0; s < W; ++s) acc = fma(uv[s], ux[s], acc);
    return acc;
}

template <int W>
__device__ __forceinline__ double sym_row(const SymArgs &A, int64_t row) {
    return sym_row_rec<W>(A, row, sym_load(A.tab, (int)A.pids[row]));
}

template <bool DOT, bool STORE, int W>
__global__ __launch_bounds__(64) void k_spmv_sym(SymArgs A) {
    if (A.flags && A.flags[0]) return;
    const int tid = threadIdx.x;
    const int b = xcd_remap(blockIdx.x, gridDim.x);
    const int r0 = A.row_begin + b * 64;
    const int nr = min(64, A.row_end - r0);
    const int64_t row = r0 + (tid < nr ? tid : 0);
    const double acc = sym_row<W>(A, row);
    if (STORE && tid < nr) A.y[r0 + tid] = acc;
    if (DOT) {
        const double v = (tid < nr) ? acc * A.w[r0 + tid] : 0.0;
        const double sum = wave_sum(v);
        if (A.qq) {
            const double s2 = wave_sum((tid < nr) ? acc * acc : 0.0);
            if (tid == 0) { A.partials[2 * b] = sum; A.partials[2 * b + 1] = s2; }
        } else if (tid == 0) A.partials[b] = sum;
    }
}

// ------------------------------------------------------------------ structured vertex grids: diagonal form
// A full structured vertex grid (row = x + nx y + nx ny z; the 15-point pattern of the 6-tetrahedra-per-cube BoxMesh)
// keeps the symmetric half storage in DIAGONAL form: slot s = dx + 2 dy + 4 dz of row i holds a(i, i + dx + nx dy + P dz),
// (dx, dy, dz) in {0, 1}^3, P = nx ny; slots whose neighbour lies outside the grid hold 0.  Same 8 arrays of n doubles as
// the position-ordered form of k_spmv_sym, but the slot of a coupling no longer depends on the row's pattern: no pattern
// id, no table, no selects in the product - and the four couplings to the plane below are exactly what the workgroup
// marching along z loaded one step earlier (its own slots 4..7), so they are handed on through LDS instead of being
// fetched a second time over the fabric (r01: 0.63 GB of 2.0 GB per launch at 256^3 were those re-reads).
// Summation order per row = ascending columns, like the sorted CSR row; absent couplings add an exact 0.
struct DiaArgs {
    const double *uvals, *x, *w;
    double *y, *partials;
    const int *flags;
    int64_t n;              // slot stride in doubles
    int nx, ny, nz;         // vertex grid of the (local) mesh
    int row_begin, row_end; // k_spmv_dia_rows
    int nblk1, row_begin2, row_end2;   // ... with a second row range from workgroup nblk1 on (nblk1 < 0: one range)
    int z0, z1, zchunk, tiles_x, tiles_y;   // k_spmv_dia_march
    int unit_diag;          // the operator is D^-1/2 A D^-1/2 of the scaled recurrence: its diagonal is 1 and is not loaded
    int qq;                 // DOT launches: partial sums in pairs (w . y, y . y) per workgroup (single-sync recurrence)
};

// row order, any row range: small grids, ranges that are not whole planes, products with w != x
template <bool DOT, bool STORE>
__global__ __launch_bounds__(64) void k_spmv_dia_rows(DiaArgs A) {
    if (A.flags && A.flags[0]) return;
    const int tid = threadIdx.x;
    const int b = xcd_remap(blockIdx.x, gridDim.x);
    const bool second = A.nblk1 >= 0 && b >= A.nblk1;       // the two boundary planes of a sharded rank in one launch
    const int r0 = second ? A.row_begin2 + (b - A.nblk1) * 64 : A.row_begin + b * 64;
    const int nr = min(64, (second ? A.row_end2 : A.row_end) - r0);
    const int64_t row = r0 + (tid < nr ? tid : 0);
    const int64_t P = (int64_t)A.nx * A.ny;
    const int z = (int)(row / P);
    const int rem = (int)(row - (int64_t)z * P);
    const int y = rem / A.nx, x = rem - y * A.nx;
    double uv[8], ux[8], lv[8], lx[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        const int dx = s & 1, dy = (s >> 1) & 1, dz = s >> 2;
        const int64_t off = dx + (int64_t)A.nx * dy + P * dz;
        const bool up = x + dx < A.nx && y + dy < A.ny && z + dz < A.nz;
        const bool lo = s > 0 && x >= dx && y >= dy && z >= dz;
        uv[s] = (s == 0 && A.unit_diag) ? 1.0 : A.uvals[(int64_t)s * A.n + row];   // 0 where the neighbour does not exist
        ux[s] = A.x[up ? row + off : row];
        const double t = A.uvals[(int64_t)s * A.n + (lo ? row - off : row)];
        lv[s] = lo ? t : 0.0;
        lx[s] = A.x[lo ? row - off : row];
    }
    double acc = 0.0;
#pragma unroll
    for (int s = 7; s >= 1; --s) acc = fma(lv[s], lx[s], acc);          // ascending columns: lower entries first
#pragma unroll
    for (int s = 0; s < 8; ++s) acc = fma(uv[s], ux[s], acc);
    if (STORE && tid < nr) A.y[r0 + tid] = acc;
    if (DOT) {
        const double v = (tid < nr) ? acc * A.w[r0 + tid] : 0.0;
        const double sum = wave_sum(v);
        if (A.qq) {
            const double s2 = wave_sum((tid < nr) ? acc * acc : 0.0);
            if (tid == 0) { A.partials[2 * b] = sum; A.partials[2 * b + 1] = s2; }
        } else if (tid == 0) A.partials[b] = sum;
    }
}

// A 256-thread workgroup owns a 64 x 4 patch of (x, y) and MARCHES along z through its chunk of planes.
// rocprofv3 --pmc (r01g) showed the L1 address/data path, not HBM, pacing every product that gathers x through the L1;
// so the three planes of x the patch touches (patch + one halo cell each way) live in an LDS ring - each x value enters
// the L1 once per patch and plane, the next plane's cells are fetched while the current plane is computed - and every
// neighbour read is a ds_read_b64 at centre + constant.  The plane-below couplings come from the LDS copy the workgroup
// made of its own slots 4..7 one step earlier; only the patch's low-x lane and low-y wave fetch theirs (they belong to
// the neighbouring patch).  The vector-memory path carries the own slot values (64 B per row), the three in-plane lower
// couplings (served by L1 / L2: the neighbouring lanes and waves load them as their own in the same step) and y.
constexpr int DM_HX = 66;            // cells per line of the staged x patch: 64 + one halo cell each way

// Workgroup barrier that orders LDS traffic only.  __syncthreads() is a workgroup-scope fence + s_barrier, and on gfx9
// the fence waits with vmcnt(0), i.e. also for the acknowledgement of the y store issued just before it in every step
// of the march.  Nothing another wave reads through LDS depends on that store, so the march waits for its LDS
// operations only (measured A/B in one process at 256^3: no difference with five workgroups per CU to overlap the
// wait; kept because it is the weaker - and sufficient - ordering).
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <bool DOT, bool STORE, int WY>
__global__ __launch_bounds__(64 * WY) void k_spmv_dia_march(DiaArgs A) {
    constexpr int NT = 64 * WY, HY = WY + 2, SLICE = DM_HX * HY;     // 64 x WY patch + halo
    static_assert(2 * NT >= SLICE, "two staged cells per thread must cover the halo patch");
    __shared__ double s_x[3 * SLICE];
    __shared__ double s_lo[4 * NT];                         // slots 4..7 of the plane below, one cell per thread
    __shared__ double s_red[WY];
    if (A.flags && A.flags[0]) return;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int b = xcd_remap(blockIdx.x, gridDim.x);
    const int per_chunk = A.tiles_x * A.tiles_y;
    const int chunk = b / per_chunk, tile = b - chunk * per_chunk;
    const int ty = tile / A.tiles_x, tx = tile - ty * A.tiles_x;
    const int x0 = tx * 64, y0 = ty * WY;
    const int x = x0 + lane, y = y0 + wv;
    const bool live = x < A.nx && y < A.ny;
    const bool inx = live && x > 0, iny = live && y > 0, inxy = inx && iny;
    const bool ldx = lane > 0, ldy = wv > 0;               // the lower neighbour in x / y is a thread of this workgroup
    const int64_t nx = A.nx, P = (int64_t)A.nx * A.ny, n = A.n;
    const int64_t base = live ? x + nx * y : 0;
    const int centre = (wv + 1) * DM_HX + lane + 1;
    const int za = A.z0 + chunk * A.zchunk, zb = min(A.z1, za + A.zchunk);
    // the two halo-patch cells this thread stages per plane
    int64_t goff[2];
    bool gok[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int i = tid + q * NT;
        const int ly = i / DM_HX, lx = i - ly * DM_HX;
        const int gx = x0 - 1 + lx, gy = y0 - 1 + ly;
        gok[q] = i < SLICE && gx >= 0 && gx < A.nx && gy >= 0 && gy < A.ny;
        goff[q] = gok[q] ? gx + nx * gy : 0;
    }
    auto fetch = [&](int z, double v[2]) {
        const bool zok = z >= 0 && z < A.nz;
#pragma unroll
        for (int q = 0; q < 2; ++q) v[q] = (zok && gok[q]) ? A.x[goff[q] + P * z] : 0.0;
    };
    auto put = [&](int z, const double v[2]) {
        const int sl = ((z % 3) + 3) % 3;
#pragma unroll
        for (int q = 0; q < 2; ++q)
            if (tid + q * NT < SLICE) s_x[sl * SLICE + tid + q * NT] = v[q];
    };
    double dot = 0.0, dot2 = 0.0;
    if (za < zb) {
        double v[2];
        for (int z = za - 1; z <= za + 1; ++z) { fetch(z, v); put(z, v); }
        if (za > 0) {
            const int64_t rb = base + P * (za - 1);
#pragma unroll
            for (int s = 0; s < 4; ++s) s_lo[s * NT + tid] = A.uvals[(int64_t)(4 + s) * n + rb];
        }
    }
    __syncthreads();
    for (int z = za; z < zb; ++z) {
        double vn[2] = {0.0, 0.0};
        if (z + 2 <= zb) fetch(z + 2, vn);                  // in flight while this plane is computed (plane zb + 1 is never read)
        const int64_t row = base + P * z;
        double uv[8];
        uv[0] = 1.0;                                        // unit diagonal of the scaled operator: 56 instead of 64 B of slots per row
        if (!A.unit_diag) uv[0] = A.uvals[row];
#pragma unroll
        for (int s = 1; s < 8; ++s) uv[s] = A.uvals[(int64_t)s * n + row];
        // in-plane lower couplings: slot s of the row s names (L1 / L2 hits, except across the patch border)
        const double t1 = A.uvals[1 * n + (inx ? row - 1 : row)];
        const double t2 = A.uvals[2 * n + (iny ? row - nx : row)];
        const double t3 = A.uvals[3 * n + (inxy ? row - nx - 1 : row)];
        double l4 = 0.0, l5 = 0.0, l6 = 0.0, l7 = 0.0;
        if (z > 0) {                                        // uniform
            l4 = live ? s_lo[tid] : 0.0;
            if (inx) l5 = ldx ? s_lo[NT + tid - 1] : A.uvals[5 * n + row - P - 1];
            if (iny) l6 = ldy ? s_lo[2 * NT + tid - 64] : A.uvals[6 * n + row - P - nx];
            if (inxy) l7 = (ldx && ldy) ? s_lo[3 * NT + tid - 65] : A.uvals[7 * n + row - P - nx - 1];
        }
        const double l1 = inx ? t1 : 0.0, l2 = iny ? t2 : 0.0, l3 = inxy ? t3 : 0.0;
        const int sl0 = ((z - 1) % 3 + 3) % 3;             // slice of plane z - 1; planes z, z + 1 follow cyclically
        const double *xm = s_x + sl0 * SLICE + centre;
        const double *xc = s_x + ((sl0 + 1) % 3) * SLICE + centre;
        const double *xp = s_x + ((sl0 + 2) % 3) * SLICE + centre;
        const double x00 = xc[0];
        double acc = l7 * xm[-DM_HX - 1];                   // ascending columns: lower entries first
        acc = fma(l6, xm[-DM_HX], acc);
        acc = fma(l5, xm[-1], acc);
        acc = fma(l4, xm[0], acc);
        acc = fma(l3, xc[-DM_HX - 1], acc);
        acc = fma(l2, xc[-DM_HX], acc);
        acc = fma(l1, xc[-1], acc);
        acc = fma(uv[0], x00, acc);
        acc = fma(uv[1], xc[1], acc);
        acc = fma(uv[2], xc[DM_HX], acc);
        acc = fma(uv[3], xc[DM_HX + 1], acc);
        acc = fma(uv[4], xp[0], acc);
        acc = fma(uv[5], xp[1], acc);
        acc = fma(uv[6], xp[DM_HX], acc);
        acc = fma(uv[7], xp[DM_HX + 1], acc);
        if (STORE && live) A.y[row] = acc;
        if (DOT && live) { dot = fma(acc, x00, dot); dot2 = fma(acc, acc, dot2); }   // the PCG product: w is x itself (checked by the launcher)
        lds_barrier();                                      // everyone is done with plane z - 1 and with s_lo
        put(z + 2, vn);                                     // ... whose slice receives plane z + 2
#pragma unroll
        for (int s = 0; s < 4; ++s) s_lo[s * NT + tid] = uv[4 + s];
        lds_barrier();
    }
    if (DOT) {
        for (int pass = 0; pass < (A.qq ? 2 : 1); ++pass) {
            const double sum = wave_sum(pass ? dot2 : dot);
            __syncthreads();
            if (lane == 0) s_red[wv] = sum;
            __syncthreads();
            if (tid == 0) {
                double t = 0.0;
#pragma unroll
                for (int k = 0; k < WY; k += 4) t += (s_red[k] + s_red[k + 1]) + (s_red[k + 2] + s_red[k + 3]);
                A.partials[A.qq ? 2 * b + pass : b] = t;
            }
        }
    }
}

// Two rows per thread: the same march on a 64 x 8 patch with 256 threads - thread (lane, wave) owns the rows y0 + 2 wave
// and y0 + 2 wave + 1.  Half the barriers and patch-border rows per row of work, twice the loads in flight per wave; the
// upper row takes its (0, -1, 0) coupling from the lower row's registers and its plane-below (0, -1, -1) coupling from the
// thread's own LDS cell.  Same per-row arithmetic and order: bit-identical to k_spmv_dia_march.
template <bool DOT, bool STORE>
__global__ __launch_bounds__(256) void k_spmv_dia_march2(DiaArgs A) {
    constexpr int NT = 256, PY = 8, HY = PY + 2, SLICE = DM_HX * HY;        // 660 cells per plane
    __shared__ double s_x[3 * SLICE];
    __shared__ double s_lo[2 * 4 * NT];                     // [row of the pair][slot 4..7][thread]
    __shared__ double s_red[4];
    if (A.flags && A.flags[0]) return;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int b = xcd_remap(blockIdx.x, gridDim.x);
    const int per_chunk = A.tiles_x * A.tiles_y;
    const int chunk = b / per_chunk, tile = b - chunk * per_chunk;
    const int ty = tile / A.tiles_x, tx = tile - ty * A.tiles_x;
    const int x0 = tx * 64, y0 = ty * PY;
    const int x = x0 + lane, ya = y0 + 2 * wv;
    const bool live0 = x < A.nx && ya < A.ny, live1 = x < A.nx && ya + 1 < A.ny;
    const bool inx0 = live0 && x > 0, inx1 = live1 && x > 0;
    const bool iny0 = live0 && ya > 0;                      // the upper row of the pair always has its y - 1 neighbour: the lower row
    const bool ldx = lane > 0, ldy = wv > 0;
    const int64_t nx = A.nx, P = (int64_t)A.nx * A.ny, n = A.n;
    const int64_t base0 = live0 ? x + nx * ya : 0, base1 = live1 ? x + nx * (ya + 1) : 0;
    const int centre = (2 * wv + 1) * DM_HX + lane + 1;     // the lower row of the pair; the upper one at + DM_HX
    const int za = A.z0 + chunk * A.zchunk, zb = min(A.z1, za + A.zchunk);
    int64_t goff[3];
    bool gok[3];
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        const int i = tid + q * NT;
        const int ly = i / DM_HX, lx = i - ly * DM_HX;
        const int gx = x0 - 1 + lx, gy = y0 - 1 + ly;
        gok[q] = i < SLICE && gx >= 0 && gx < A.nx && gy >= 0 && gy < A.ny;
        goff[q] = gok[q] ? gx + nx * gy : 0;
    }
    auto fetch = [&](int z, double v[3]) {
        const bool zok = z >= 0 && z < A.nz;
#pragma unroll
        for (int q = 0; q < 3; ++q) v[q] = (zok && gok[q]) ? A.x[goff[q] + P * z] : 0.0;
    };
    auto put = [&](int z, const double v[3]) {
        const int sl = ((z % 3) + 3) % 3;
#pragma unroll
        for (int q = 0; q < 3; ++q)
            if (tid + q * NT < SLICE) s_x[sl * SLICE + tid + q * NT] = v[q];
    };
    double dot = 0.0, dot2 = 0.0;
    if (za < zb) {
        double v[3];
        for (int z = za - 1; z <= za + 1; ++z) { fetch(z, v); put(z, v); }
        if (za > 0) {
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                s_lo[s * NT + tid] = A.uvals[(int64_t)(4 + s) * n + base0 + P * (za - 1)];
                s_lo[(4 + s) * NT + tid] = A.uvals[(int64_t)(4 + s) * n + base1 + P * (za - 1)];
            }
        }
    }
    __syncthreads();
    for (int z = za; z < zb; ++z) {
        double vn[3] = {0.0, 0.0, 0.0};
        if (z + 2 <= zb) fetch(z + 2, vn);                  // plane zb + 1 is never read
        const int64_t r0 = base0 + P * z, r1 = base1 + P * z;
        double u0[8], u1[8];
        u0[0] = u1[0] = 1.0;                                // unit diagonal of the scaled operator: 56 instead of 64 B of slots per row
        if (!A.unit_diag) { u0[0] = A.uvals[r0]; u1[0] = A.uvals[r1]; }
#pragma unroll
        for (int s = 1; s < 8; ++s) { u0[s] = A.uvals[(int64_t)s * n + r0]; u1[s] = A.uvals[(int64_t)s * n + r1]; }
        // in-plane lower couplings from the neighbouring rows' slots (L1 / L2); (0, -1) of the upper row = u0[2]
        const double t1a = A.uvals[1 * n + (inx0 ? r0 - 1 : r0)];
        const double t2a = A.uvals[2 * n + (iny0 ? r0 - nx : r0)];
        const double t3a = A.uvals[3 * n + ((inx0 && iny0) ? r0 - nx - 1 : r0)];
        const double t1b = A.uvals[1 * n + (inx1 ? r1 - 1 : r1)];
        const double t3b = A.uvals[3 * n + (inx1 ? r1 - nx - 1 : r1)];
        double a4 = 0.0, a5 = 0.0, a6 = 0.0, a7 = 0.0, b4 = 0.0, b5 = 0.0, b6 = 0.0, b7 = 0.0;
        if (z > 0) {                                        // uniform
            a4 = live0 ? s_lo[tid] : 0.0;
            b4 = live1 ? s_lo[4 * NT + tid] : 0.0;
            if (inx0) a5 = ldx ? s_lo[NT + tid - 1] : A.uvals[5 * n + r0 - P - 1];
            if (inx1) b5 = ldx ? s_lo[5 * NT + tid - 1] : A.uvals[5 * n + r1 - P - 1];
            if (iny0) a6 = ldy ? s_lo[6 * NT + tid - 64] : A.uvals[6 * n + r0 - P - nx];      // row below the pair: upper row of wave - 1
            if (live1) b6 = s_lo[2 * NT + tid];                                                // the pair's own lower row
            if (inx0 && iny0) a7 = (ldx && ldy) ? s_lo[7 * NT + tid - 65] : A.uvals[7 * n + r0 - P - nx - 1];
            if (inx1) b7 = ldx ? s_lo[3 * NT + tid - 1] : A.uvals[7 * n + r1 - P - nx - 1];
        }
        const double a1 = inx0 ? t1a : 0.0, a2 = iny0 ? t2a : 0.0, a3 = (inx0 && iny0) ? t3a : 0.0;
        const double b1 = inx1 ? t1b : 0.0, b2 = live1 ? u0[2] : 0.0, b3 = inx1 ? t3b : 0.0;
        const int sl0 = ((z - 1) % 3 + 3) % 3;
        const double *xm = s_x + sl0 * SLICE + centre;
        const double *xc = s_x + ((sl0 + 1) % 3) * SLICE + centre;
        const double *xp = s_x + ((sl0 + 2) % 3) * SLICE + centre;
        const double xa = xc[0], xb = xc[DM_HX];
        double acc0 = a7 * xm[-DM_HX - 1];
        acc0 = fma(a6, xm[-DM_HX], acc0);
        acc0 = fma(a5, xm[-1], acc0);
        acc0 = fma(a4, xm[0], acc0);
        acc0 = fma(a3, xc[-DM_HX - 1], acc0);
        acc0 = fma(a2, xc[-DM_HX], acc0);
        acc0 = fma(a1, xc[-1], acc0);
        acc0 = fma(u0[0], xa, acc0);
        acc0 = fma(u0[1], xc[1], acc0);
        acc0 = fma(u0[2], xc[DM_HX], acc0);
        acc0 = fma(u0[3], xc[DM_HX + 1], acc0);
        acc0 = fma(u0[4], xp[0], acc0);
        acc0 = fma(u0[5], xp[1], acc0);
        acc0 = fma(u0[6], xp[DM_HX], acc0);
        acc0 = fma(u0[7], xp[DM_HX + 1], acc0);
        double acc1 = b7 * xm[-1];
        acc1 = fma(b6, xm[0], acc1);
        acc1 = fma(b5, xm[DM_HX - 1], acc1);
        acc1 = fma(b4, xm[DM_HX], acc1);
        acc1 = fma(b3, xc[-1], acc1);
        acc1 = fma(b2, xc[0], acc1);
        acc1 = fma(b1, xc[DM_HX - 1], acc1);
        acc1 = fma(u1[0], xb, acc1);
        acc1 = fma(u1[1], xc[DM_HX + 1], acc1);
        acc1 = fma(u1[2], xc[2 * DM_HX], acc1);
        acc1 = fma(u1[3], xc[2 * DM_HX + 1], acc1);
        acc1 = fma(u1[4], xp[DM_HX], acc1);
        acc1 = fma(u1[5], xp[DM_HX + 1], acc1);
        acc1 = fma(u1[6], xp[2 * DM_HX], acc1);
        acc1 = fma(u1[7], xp[2 * DM_HX + 1], acc1);
        if (STORE && live0) A.y[r0] = acc0;
        if (STORE && live1) A.y[r1] = acc1;
        if (DOT && live0) { dot = fma(acc0, xa, dot); dot2 = fma(acc0, acc0, dot2); }
        if (DOT && live1) { dot = fma(acc1, xb, dot); dot2 = fma(acc1, acc1, dot2); }
        lds_barrier();
        put(z + 2, vn);
#pragma unroll
        for (int s = 0; s < 4; ++s) { s_lo[s * NT + tid] = u0[4 + s]; s_lo[(4 + s) * NT + tid] = u1[4 + s]; }
        lds_barrier();
    }
    if (DOT) {
        for (int pass = 0; pass < (A.qq ? 2 : 1); ++pass) {
            const double sum = wave_sum(pass ? dot2 : dot);
            __syncthreads();
            if (lane == 0) s_red[wv] = sum;
            __syncthreads();
            if (tid == 0) A.partials[A.qq ? 2 * b + pass : b] = (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
        }
    }
}

// The same march with the r03 techniques of the stencil kernel (round 4, VERDICT r03 "next" #4): every slot value, every x cell
// and every y row goes through a BUFFER descriptor with a 32-bit byte offset per lane that never changes during the march - the
// plane is carried by the scalar offset (slot s of plane z: 8 (s n + P z)) or by a per-plane descriptor built from scalars - and
// a lane that must not read (outside the grid, no neighbour on that side, not a border lane) carries an offset beyond the buffer:
// the range check returns 0.0 / drops the store, so there are no exec masks, no selects and no 64-bit address arithmetic in the
// step.  The registers that frees hold the NEXT plane's slot values: the loads of plane z + 1 are issued before plane z is
// multiplied (two register sets that swap roles, the loop is unrolled by two), so a wave always has a plane of slot loads in
// flight behind the one it is consuming.  Same values, same 15 fused multiply-adds per row in the same order: bit-identical to
// k_spmv_dia_march2.  Needs 64 n < 2^31 (one descriptor over the eight slot arrays) and P < 2^27.
typedef int dm_v2i __attribute__((ext_vector_type(2)));
struct Dm3Plane { double u0[8], u1[8], t1a, t2a, t3a, t1b, t3b, g5a, g5b, g6a, g7a, g7b; };

template <bool DOT, bool STORE>
__global__ __launch_bounds__(256) void k_spmv_dia_march3(DiaArgs A) {
    constexpr int NT = 256, PY = 8, HY = PY + 2, SLICE = DM_HX * HY;        // 660 cells per plane
    constexpr int OOB = (int)0x80000000;                                    // byte offset beyond every buffer of the kernel
    __shared__ double s_x[3 * SLICE];
    __shared__ double s_lo[2 * 4 * NT];                     // [row of the pair][slot 4..7][thread]
    __shared__ double s_red[4];
    if (A.flags && A.flags[0]) return;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int b = xcd_remap(blockIdx.x, gridDim.x);
    const int per_chunk = A.tiles_x * A.tiles_y;
    const int chunk = b / per_chunk, tile = b - chunk * per_chunk;
    const int ty = tile / A.tiles_x, tx = tile - ty * A.tiles_x;
    const int x0 = tx * 64, y0 = ty * PY;
    const int x = x0 + lane, ya = y0 + 2 * wv;
    const bool live0 = x < A.nx && ya < A.ny, live1 = x < A.nx && ya + 1 < A.ny;
    const bool inx0 = live0 && x > 0, inx1 = live1 && x > 0;
    const bool iny0 = live0 && ya > 0;
    const bool ldx = lane > 0, ldy = wv > 0;
    const int nx = A.nx, P = A.nx * A.ny;
    const int P8 = 8 * P, n8 = (int)(8 * A.n);
    const int base0 = x + nx * ya, base1 = base0 + nx;
    const int centre = (2 * wv + 1) * DM_HX + lane + 1;
    const int za = A.z0 + chunk * A.zchunk, zb = min(A.z1, za + A.zchunk);
    // lane offsets, fixed for the whole march
    const int o0 = live0 ? 8 * base0 : OOB, o1 = live1 ? 8 * base1 : OOB;
    const int o1a = inx0 ? 8 * (base0 - 1) : OOB, o2a = iny0 ? 8 * (base0 - nx) : OOB, o3a = (inx0 && iny0) ? 8 * (base0 - nx - 1) : OOB;
    const int o1b = inx1 ? 8 * (base1 - 1) : OOB, o3b = inx1 ? 8 * (base1 - nx - 1) : OOB;
    // plane-below couplings of the patch's low-x lane / low-y wave belong to the neighbouring patch: only those lanes load
    const int q5a = (inx0 && !ldx) ? 8 * (base0 - 1) : OOB, q5b = (inx1 && !ldx) ? 8 * (base1 - 1) : OOB;
    const int q6a = (iny0 && !ldy) ? 8 * (base0 - nx) : OOB;
    const int q7a = (inx0 && iny0 && !(ldx && ldy)) ? 8 * (base0 - nx - 1) : OOB, q7b = (inx1 && !ldx) ? 8 * (base1 - nx - 1) : OOB;
    int gv[3];
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        const int i = tid + q * NT;
        const int ly = i / DM_HX, lx = i - ly * DM_HX;
        const int gx = x0 - 1 + lx, gy = y0 - 1 + ly;
        gv[q] = (i < SLICE && gx >= 0 && gx < A.nx && gy >= 0 && gy < A.ny) ? 8 * (gx + nx * gy) : OOB;
    }
    const auto rsU = __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(A.uvals), 0, 8 * n8, 0x00020000);
    auto ldu = [&](int voff, int soff) { return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rsU, voff, soff, 0)); };
    auto fetch = [&](int z, double (&v)[3]) {
        const bool zok = z >= 0 && z < A.nz;
        const auto r = __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(A.x + (int64_t)P * (zok ? z : 0)), 0, zok ? P8 : 0, 0x00020000);
#pragma unroll
        for (int q = 0; q < 3; ++q) v[q] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, gv[q], 0, 0));
    };
    auto put = [&](int z, const double (&v)[3]) {
        const int sl = ((z % 3) + 3) % 3;
#pragma unroll
        for (int q = 0; q < 3; ++q)
            if (tid + q * NT < SLICE) s_x[sl * SLICE + tid + q * NT] = v[q];
    };
    // all vector-memory reads of plane z (and, for its border lanes, of the couplings up from plane z - 1)
    auto load_plane = [&](int z, Dm3Plane &R) {
        const int zo = P8 * z;
        R.u0[0] = R.u1[0] = 1.0;
        if (!A.unit_diag) { R.u0[0] = ldu(o0, zo); R.u1[0] = ldu(o1, zo); }
#pragma unroll
        for (int s = 1; s < 8; ++s) { R.u0[s] = ldu(o0, s * n8 + zo); R.u1[s] = ldu(o1, s * n8 + zo); }
        R.t1a = ldu(o1a, 1 * n8 + zo); R.t2a = ldu(o2a, 2 * n8 + zo); R.t3a = ldu(o3a, 3 * n8 + zo);
        R.t1b = ldu(o1b, 1 * n8 + zo); R.t3b = ldu(o3b, 3 * n8 + zo);
        R.g5a = R.g5b = R.g6a = R.g7a = R.g7b = 0.0;
        if (z > 0) {                                        // uniform
            const int zm = zo - P8;
            R.g5a = ldu(q5a, 5 * n8 + zm); R.g5b = ldu(q5b, 5 * n8 + zm); R.g6a = ldu(q6a, 6 * n8 + zm);
            R.g7a = ldu(q7a, 7 * n8 + zm); R.g7b = ldu(q7b, 7 * n8 + zm);
        }
    };
    double dot = 0.0, dot2 = 0.0;
    Dm3Plane Ra, Rb;
    if (za < zb) {
        double v[3];
        for (int z = za - 1; z <= za + 1; ++z) { fetch(z, v); put(z, v); }
        if (za > 0) {
            const int zm = P8 * (za - 1);
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                s_lo[s * NT + tid] = ldu(o0, (4 + s) * n8 + zm);
                s_lo[(4 + s) * NT + tid] = ldu(o1, (4 + s) * n8 + zm);
            }
        }
        load_plane(za, Ra);
    }
    __syncthreads();
    auto step = [&](int z, Dm3Plane &C, Dm3Plane &N) {
        double vn[3] = {0.0, 0.0, 0.0};
        if (z + 2 <= zb) fetch(z + 2, vn);                  // plane zb + 1 is never read
        if (z + 1 < zb) load_plane(z + 1, N);               // the next plane's slot values travel while this one is multiplied
        double a4 = 0.0, a5 = 0.0, a6 = 0.0, a7 = 0.0, b4 = 0.0, b5 = 0.0, b6 = 0.0, b7 = 0.0;
        if (z > 0) {                                        // uniform
            a4 = live0 ? s_lo[tid] : 0.0;
            b4 = live1 ? s_lo[4 * NT + tid] : 0.0;
            if (inx0) a5 = ldx ? s_lo[NT + tid - 1] : C.g5a;
            if (inx1) b5 = ldx ? s_lo[5 * NT + tid - 1] : C.g5b;
            if (iny0) a6 = ldy ? s_lo[6 * NT + tid - 64] : C.g6a;
            if (live1) b6 = s_lo[2 * NT + tid];
            if (inx0 && iny0) a7 = (ldx && ldy) ? s_lo[7 * NT + tid - 65] : C.g7a;
            if (inx1) b7 = ldx ? s_lo[3 * NT + tid - 1] : C.g7b;
        }
        const double a1 = C.t1a, a2 = C.t2a, a3 = C.t3a;    // (0.0 where there is no such neighbour: the range check)
        const double b1 = C.t1b, b2 = live1 ? C.u0[2] : 0.0, b3 = C.t3b;
        const int sl0 = ((z - 1) % 3 + 3) % 3;
        const double *xm = s_x + sl0 * SLICE + centre;
        const double *xc = s_x + ((sl0 + 1) % 3) * SLICE + centre;
        const double *xp = s_x + ((sl0 + 2) % 3) * SLICE + centre;
        const double xa = xc[0], xb = xc[DM_HX];
        double acc0 = a7 * xm[-DM_HX - 1];
        acc0 = fma(a6, xm[-DM_HX], acc0);
        acc0 = fma(a5, xm[-1], acc0);
        acc0 = fma(a4, xm[0], acc0);
        acc0 = fma(a3, xc[-DM_HX - 1], acc0);
        acc0 = fma(a2, xc[-DM_HX], acc0);
        acc0 = fma(a1, xc[-1], acc0);
        acc0 = fma(C.u0[0], xa, acc0);
        acc0 = fma(C.u0[1], xc[1], acc0);
        acc0 = fma(C.u0[2], xc[DM_HX], acc0);
        acc0 = fma(C.u0[3], xc[DM_HX + 1], acc0);
        acc0 = fma(C.u0[4], xp[0], acc0);
        acc0 = fma(C.u0[5], xp[1], acc0);
        acc0 = fma(C.u0[6], xp[DM_HX], acc0);
        acc0 = fma(C.u0[7], xp[DM_HX + 1], acc0);
        double acc1 = b7 * xm[-1];
        acc1 = fma(b6, xm[0], acc1);
        acc1 = fma(b5, xm[DM_HX - 1], acc1);
        acc1 = fma(b4, xm[DM_HX], acc1);
        acc1 = fma(b3, xc[-1], acc1);
        acc1 = fma(b2, xc[0], acc1);
        acc1 = fma(b1, xc[DM_HX - 1], acc1);
        acc1 = fma(C.u1[0], xb, acc1);
        acc1 = fma(C.u1[1], xc[DM_HX + 1], acc1);
        acc1 = fma(C.u1[2], xc[2 * DM_HX], acc1);
        acc1 = fma(C.u1[3], xc[2 * DM_HX + 1], acc1);
        acc1 = fma(C.u1[4], xp[DM_HX], acc1);
        acc1 = fma(C.u1[5], xp[DM_HX + 1], acc1);
        acc1 = fma(C.u1[6], xp[2 * DM_HX], acc1);
        acc1 = fma(C.u1[7], xp[2 * DM_HX + 1], acc1);
        if (STORE) {
            const auto ry = __builtin_amdgcn_make_buffer_rsrc(A.y + (int64_t)P * z, 0, P8, 0x00020000);
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(dm_v2i, acc0), ry, o0, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(dm_v2i, acc1), ry, o1, 0, 0);
        }
        if (DOT && live0) { dot = fma(acc0, xa, dot); dot2 = fma(acc0, acc0, dot2); }
        if (DOT && live1) { dot = fma(acc1, xb, dot); dot2 = fma(acc1, acc1, dot2); }
        lds_barrier();
        put(z + 2, vn);
#pragma unroll
        for (int s = 0; s < 4; ++s) { s_lo[s * NT + tid] = C.u0[4 + s]; s_lo[(4 + s) * NT + tid] = C.u1[4 + s]; }
        lds_barrier();
    };
    for (int z = za; z < zb; z += 2) {
        step(z, Ra, Rb);
        if (z + 1 < zb) step(z + 1, Rb, Ra);
    }
    if (DOT) {
        for (int pass = 0; pass < (A.qq ? 2 : 1); ++pass) {
            const double sum = wave_sum(pass ? dot2 : dot);
            __syncthreads();
            if (lane == 0) s_red[wv] = sum;
            __syncthreads();
            if (tid == 0) A.partials[A.qq ? 2 * b + pass : b] = (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
        }
    }
}

// ------------------------------------------------------------------ diagonal form with a row-class dictionary
// On a uniform grid with piecewise-constant coefficients the rows of the (scaled) operator repeat a few 8-tuples of slot
// values: every interior row carries the same one, the rows next to a face, an edge, a corner or a Dirichlet row a few dozen
// others.  dia_classify() (below) finds them per solve, LOSSLESSLY - the class table holds the slot values bit by bit and
// every row is compared with its class's tuple before the code is trusted - and the product then streams ONE BYTE per row
// instead of 56: 17 B per row (code + x + y).  The table lives in LDS; a row of it is [s1 s5 | s3 s7 | s2 s6 | s4 s0], so the
// pairs a neighbour's row contributes - (-1, 0) gives slots 1 and 5, (-1, -1) gives 3 and 7, (0, -1) gives 2 and 6 - are
// one ds_read_b128 each, and the second value of each pair is the coupling to the plane ABOVE the neighbour, i.e. exactly
// what the march needs one step later for its plane below: the eight plane-below couplings of a thread's two rows are
// carried in registers from step to step, nothing of the operator is re-read.  Class id `ncls` is the all-zero row: cells
// outside the grid carry it, so the boundary needs no predicates (0 * x adds an exact 0, as the masked form does).
// Same values, same order of the 15 fused multiply-adds per row: bit-identical to k_spmv_dia_march2.
constexpr int CLS_MAX = 255;             // classes per operator (+ the zero class = 256 table rows of 64 B)
constexpr int CLS_SLOTS = 1024;          // hash slots of the classification
__host__ __device__ constexpr int cls_pos(int s) {      // position of slot s in a table row
    return s == 1 ? 0 : s == 5 ? 1 : s == 3 ? 2 : s == 7 ? 3 : s == 2 ? 4 : s == 6 ? 5 : s == 4 ? 6 : 7;
}

struct DiacArgs {
    const uint8_t *cls;
    const int *same;        // [nz]: plane z carries, row by row, the codes of plane z - 1 (scalar loads: ints)
    const double *table;    // [ncls + 1][8]
    int ncls;
    const double *x;
    double *y, *partials;
    const int *flags;
    int nx, ny, nz;
    int z0, z1, zchunk, tiles_x, tiles_y;
    int qq;
    int nt_y;               // y is stored with the non-temporal hint (the launcher picks the NTY instantiation)
};

// A kernel this light is paced by instruction issue and by the bytes a CU keeps in flight, so the march is built around
// what does NOT change from plane to plane:
// * the couplings live in registers and are looked up again only where the class codes of a plane differ from the plane
//   before (dia_classify flags those planes: on a uniform grid the planes next to the z faces, and the first plane of a
//   march).  In between, a step touches no code and no table, and the couplings to the plane below are the upward halves of
//   the pairs it already holds.
// * D = 6 (or 3) plane fetches are in flight per workgroup - D register sets that rotate by name, the march being unrolled
//   D steps at a time - and nothing touches a fetched value before it is staged (the zero of the cells outside the grid
//   is selected then): a select or a lane predicate at the load makes the wave wait for the fetch it has just issued.
// * run-time switches cost taken branches in every step: the non-temporal y store is a template parameter (NTY), the
//   "this step looks its couplings up" flags are one wave-uniform 64-bit mask per 64 steps.
// * x is read from LDS (22 ds_read_b64 for a thread's two rows); four slices - the planes z - 1, z, z + 1 being read and
//   z + 2 being staged - make ONE barrier per step enough.
using d2 = __attribute__((ext_vector_type(2))) double;
struct DiacT { d2 A15, A37, A26, A40, B15, B37, B26, B40, L15, L37, LB15, D26, LD37; };
constexpr int DIAC_MAXCHUNK = 1024;    // planes per march at most

template <bool DOT, bool STORE, int D, bool NTY>
__global__ __launch_bounds__(256) void k_spmv_diac_march2(DiacArgs A) {
    constexpr int NT = 256, PY = 8, HY = PY + 2, SLICE = DM_HX * HY;        // 660 cells per plane
    constexpr int SLOT = 3 * NT;                            // ... in slots of 768: every thread stages three cells, no predicates
    __shared__ double s_x[4 * SLOT];
    __shared__ __attribute__((aligned(16))) double s_t[(CLS_MAX + 1) * 8];
    __shared__ uint8_t s_c[4 * SLOT];
    __shared__ double s_red[4];
    if (A.flags && A.flags[0]) return;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int b = xcd_remap(blockIdx.x, gridDim.x);
    const int per_chunk = A.tiles_x * A.tiles_y;
    const int chunk = b / per_chunk, tile = b - chunk * per_chunk;
    const int ty = tile / A.tiles_x, tx = tile - ty * A.tiles_x;
    const int x0 = tx * 64, y0 = ty * PY;
    const int x = x0 + lane, ya = y0 + 2 * wv;
    const bool live0 = x < A.nx && ya < A.ny, live1 = x < A.nx && ya + 1 < A.ny;
    const int64_t P = (int64_t)A.nx * A.ny;
    const int base0 = live0 ? x + A.nx * ya : 0, base1 = live1 ? x + A.nx * (ya + 1) : 0;      // offsets within a plane (P < 2^31)
    const int centre = (2 * wv + 1) * DM_HX + lane + 1;     // the lower row of the pair; the upper one at + DM_HX
    const int za = A.z0 + chunk * A.zchunk, zb = min(A.z1, za + A.zchunk);
    const unsigned zero_cls = (unsigned)A.ncls;
    for (int i = tid; i < (A.ncls + 1) * 8; i += NT) s_t[i] = A.table[i];
    // A.same[z] != 0: every row of plane z carries the code of the row below it (plane z - 1).  A step looks codes up where
    // that does not hold, and at the first plane of the march: one bit per step, 64 steps per (wave-uniform) mask - no memory
    // access and no wait in the steps themselves.
    auto slow_mask = [&](int zbase) -> unsigned long long {
        const int z = zbase + lane;
        return __ballot(z == za || !(z > 0 && z < A.nz && A.same[z] != 0));
    };
    unsigned long long look_mask = slow_mask(za);
    if (za >= zb) {                                         // uniform; the launcher sizes the grid so that no chunk is empty
        if (DOT && tid == 0) { if (A.qq) { A.partials[2 * b] = 0.0; A.partials[2 * b + 1] = 0.0; } else A.partials[b] = 0.0; }
        return;
    }
    int goff[3];
    bool gok[3];
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        const int i = tid + q * NT;
        const int ly = i / DM_HX, lx = i - ly * DM_HX;
        const int gx = x0 - 1 + lx, gy = y0 - 1 + ly;
        gok[q] = i < SLICE && gx >= 0 && gx < A.nx && gy >= 0 && gy < A.ny;
        goff[q] = gok[q] ? gx + A.nx * gy : 0;
    }
    // Loads are unconditional (cells outside the grid read a valid address, planes outside it the nearest plane).  Planes
    // beyond zb are never used by the march (zb itself is: the plane above the last one).
    auto fetch = [&](int z, double (&v)[3], unsigned (&cc)[3]) {
        const int zc = min(max(z, 0), A.nz - 1);
        const double *xz = A.x + P * zc;                    // uniform base + 32-bit lane offset
        const uint8_t *cz = A.cls + P * zc;
#pragma unroll
        for (int q = 0; q < 3; ++q) { v[q] = xz[goff[q]]; cc[q] = cz[goff[q]]; }
    };
    auto put = [&](int z, const double (&v)[3], const unsigned (&cc)[3]) {
        const bool zok = z >= 0 && z < A.nz && z <= zb;
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            s_x[(z & 3) * SLOT + tid + q * NT] = (zok && gok[q]) ? v[q] : 0.0;
            s_c[(z & 3) * SLOT + tid + q * NT] = (uint8_t)((zok && gok[q]) ? cc[q] : zero_cls);
        }
    };
    auto pair = [&](int code, int k) -> d2 { return *reinterpret_cast<const d2 *>(s_t + code * 8 + 2 * k); };
    DiacT T;
    double dot = 0.0, dot2 = 0.0;
    double rr[D][3];                                        // the D plane fetches in flight (D = 3 or 6 register sets, rotating by name)
    unsigned kk[D][3];
    fetch(za - 1, rr[0], kk[0]);
    fetch(za, rr[1], kk[1]);
    fetch(za + 1, rr[2], kk[2]);
    put(za - 1, rr[0], kk[0]);
    put(za, rr[1], kk[1]);
    put(za + 1, rr[2], kk[2]);
#pragma unroll
    for (int s = 0; s < D; ++s) fetch(za + 2 + s, rr[s], kk[s]);
    __syncthreads();
    double a4 = 0.0, a5 = 0.0, a6 = 0.0, a7 = 0.0, b4 = 0.0, b5 = 0.0, b6 = 0.0, b7 = 0.0;      // couplings to the plane below
    // one step: rows of plane z; `rv / rk` hold plane z + 2 (staged at the end of the step) and then take the fetch of z + 2 + D
    auto step = [&](int z, double (&rv)[3], unsigned (&rk)[3]) {
        const int zi = z - za;
        if (zi > 0 && (zi & 63) == 0) look_mask = slow_mask(z);       // uniform; marches longer than 64 planes
        const bool look = (look_mask >> (zi & 63)) & 1ull;
        if (__builtin_expect(look, 0)) {                                         // uniform: look the couplings up (codes of planes z - 1 and z are staged)
            const uint8_t *cm = s_c + ((z - 1) & 3) * SLOT + centre;
            const int mA = cm[0], mB = cm[DM_HX], mL = cm[-1], mLB = cm[DM_HX - 1], mD = cm[-DM_HX], mLD = cm[-DM_HX - 1];
            a4 = pair(mA, 3).x; b6 = pair(mA, 2).y; b4 = pair(mB, 3).x; a5 = pair(mL, 0).y; b7 = pair(mL, 1).y;
            b5 = pair(mLB, 0).y; a6 = pair(mD, 2).y; a7 = pair(mLD, 1).y;
            const uint8_t *cz = s_c + (z & 3) * SLOT + centre;
            const int cA = cz[0], cB = cz[DM_HX], cL = cz[-1], cLB = cz[DM_HX - 1], cD = cz[-DM_HX], cLD = cz[-DM_HX - 1];
            T.A15 = pair(cA, 0); T.A37 = pair(cA, 1); T.A26 = pair(cA, 2); T.A40 = pair(cA, 3);
            T.B15 = pair(cB, 0); T.B37 = pair(cB, 1); T.B26 = pair(cB, 2); T.B40 = pair(cB, 3);
            T.L15 = pair(cL, 0); T.L37 = pair(cL, 1); T.LB15 = pair(cLB, 0); T.D26 = pair(cD, 2); T.LD37 = pair(cLD, 1);
        }
        const double *xm = s_x + ((z - 1) & 3) * SLOT + centre;
        const double *xc = s_x + (z & 3) * SLOT + centre;
        const double *xp = s_x + ((z + 1) & 3) * SLOT + centre;
        const double xa = xc[0], xb = xc[DM_HX];
        double acc0 = a7 * xm[-DM_HX - 1];
        acc0 = fma(a6, xm[-DM_HX], acc0);
        acc0 = fma(a5, xm[-1], acc0);
        acc0 = fma(a4, xm[0], acc0);
        acc0 = fma(T.LD37.x, xc[-DM_HX - 1], acc0);
        acc0 = fma(T.D26.x, xc[-DM_HX], acc0);
        acc0 = fma(T.L15.x, xc[-1], acc0);
        acc0 = fma(T.A40.y, xa, acc0);
        acc0 = fma(T.A15.x, xc[1], acc0);
        acc0 = fma(T.A26.x, xb, acc0);
        acc0 = fma(T.A37.x, xc[DM_HX + 1], acc0);
        acc0 = fma(T.A40.x, xp[0], acc0);
        acc0 = fma(T.A15.y, xp[1], acc0);
        acc0 = fma(T.A26.y, xp[DM_HX], acc0);
        acc0 = fma(T.A37.y, xp[DM_HX + 1], acc0);
        double acc1 = b7 * xm[-1];
        acc1 = fma(b6, xm[0], acc1);
        acc1 = fma(b5, xm[DM_HX - 1], acc1);
        acc1 = fma(b4, xm[DM_HX], acc1);
        acc1 = fma(T.L37.x, xc[-1], acc1);
        acc1 = fma(T.A26.x, xa, acc1);
        acc1 = fma(T.LB15.x, xc[DM_HX - 1], acc1);
        acc1 = fma(T.B40.y, xb, acc1);
        acc1 = fma(T.B15.x, xc[DM_HX + 1], acc1);
        acc1 = fma(T.B26.x, xc[2 * DM_HX], acc1);
        acc1 = fma(T.B37.x, xc[2 * DM_HX + 1], acc1);
        acc1 = fma(T.B40.x, xp[DM_HX], acc1);
        acc1 = fma(T.B15.y, xp[DM_HX + 1], acc1);
        acc1 = fma(T.B26.y, xp[2 * DM_HX], acc1);
        acc1 = fma(T.B37.y, xp[2 * DM_HX + 1], acc1);
        const bool on = z < zb;                             // (the march runs in groups of D steps: idle steps behind the last plane)
        double *yz = A.y + P * (on ? z : za);
        if (NTY) {
            if (STORE && live0 && on) __builtin_nontemporal_store(acc0, yz + base0);
            if (STORE && live1 && on) __builtin_nontemporal_store(acc1, yz + base1);
        } else {
            if (STORE && live0 && on) yz[base0] = acc0;
            if (STORE && live1 && on) yz[base1] = acc1;
        }
        if (DOT) {
            const double d0 = (live0 && on) ? acc0 : 0.0, d1 = (live1 && on) ? acc1 : 0.0;      // + 0 * x: the sums are unchanged
            dot = fma(d0, xa, dot); dot2 = fma(d0, d0, dot2);
            dot = fma(d1, xb, dot); dot2 = fma(d1, d1, dot2);
        }
        // the plane below the next step is this one: its upward couplings are the second halves of the pairs just used - and
        // stay that until the codes change again
        if (look) { a4 = T.A40.x; b6 = T.A26.y; b4 = T.B40.x; a5 = T.L15.y; b7 = T.L37.y; b5 = T.LB15.y; a6 = T.D26.y; a7 = T.LD37.y; }
        put(z + 2, rv, rk);                                 // slot (z + 2) & 3: the plane z - 2 that nobody reads any more
        fetch(z + 2 + D, rv, rk);
        lds_barrier();
    };
#pragma clang loop unroll(disable)
    for (int z = za; z < zb; z += D) {
#pragma unroll
        for (int s = 0; s < D; ++s) step(z + s, rr[s], kk[s]);
    }
    if (DOT) {
        for (int pass = 0; pass < (A.qq ? 2 : 1); ++pass) {
            const double sum = wave_sum(pass ? dot2 : dot);
            __syncthreads();
            if (lane == 0) s_red[wv] = sum;
            __syncthreads();
            if (tid == 0) A.partials[A.qq ? 2 * b + pass : b] = (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
        }
    }
}

// ------------------------------------------------------------------ STENCIL form of the row-class dictionary (r03)
// Where dia_classify finds that the classes of an operator are ONE tuple (c0 .. c7) - the "base class" - plus the rows it
// becomes next to eliminated (Dirichlet) nodes and next to the rim of the grid, i.e. every coupling a(i, j) is either c_s
// bit for bit or an exact 0 where j is an identity row or lies outside the grid (k_stencil_verify checks EVERY row of the
// planes a launch may touch, every slot, bitwise), the product needs no table at all:
//     y_i = x_i                               for an identity row i,
//     y_i = sum over the 15 neighbours of c_s * x~_j      otherwise,      x~_j = 0 for identity rows and outside the grid,
// the same 15 fused multiply-adds in the same order as k_spmv_diac_march2 with the same factors - c_s * 0 adds the exact 0
// that 0 * x_j adds there - so y is bit-identical for finite x.  That is the constant-coefficient operator on a uniform
// lattice with Dirichlet nodes (cfg4: -Laplace + mu on the box, homogeneous hull); natural boundaries change the tuples of the
// rim rows and keep the dictionary form.  One more condition keeps the kernel free of per-plane bookkeeping: the planes of the
// verified range are [planes of identity rows only]* [a run of planes with identical codes: the MAIN run] [identity planes]*
// (k_stencil_planes); anything else - Dirichlet nodes that change from plane to plane - stays with the dictionary form.
// What it buys: a step of k_spmv_diac_march2 is ~150 instructions per wave for 128 rows, a third of them the product itself.
// Here the couplings are 8 kernel arguments = SCALAR registers (v_fma_f64 takes one scalar operand), which frees the 68 vector
// registers they took and pays for FOUR rows per thread (a 64 x 16 patch per 256 threads: 16 % instead of 29 % halo cells, half
// the barriers, scalar work and staging per row, 9 instead of 11 LDS reads per row), and every per-lane condition of a step is
// folded into state that is computed ONCE per march from the codes of a main plane:
//   * loads and stores are raw buffer accesses through a descriptor of ONE plane: cells outside the grid, the halo cell of an
//     eliminated node, rows that are not stored carry an offset beyond the plane - the range check returns 0 / drops the store;
//     no exec masks, no 64-bit address arithmetic, no selects at staging;
//   * the own cells are loaded raw, staged through a multiplier m in {0, 1} and KEPT in registers until their plane is
//     multiplied two steps later (three planes, rotating by name): the y = x of an eliminated row leaves in the same full-line
//     store as the sums of its neighbours (written on its own, early, it was a partial-line write per boundary row: 12 us of a
//     65 us launch), and the fused dots take x from there;
//   * planes of identity rows only are copies out of those registers and are staged as zeros; no code byte is read in the
//     march: 16 B per row.
struct StencilArgs {
    const uint8_t *cls;
    double c[8];            // c[s] = coupling of slot s = dx + 2 dy + 4 dz (c[0]: the diagonal)
    int ident;              // class id of the identity rows (-1: none)
    const double *x;
    double *y, *partials;
    const int *flags;
    int nx, ny, nz;
    int zv0, zv1;           // planes the form was verified on (the grid, or the owned planes of a sharded slab)
    int zm0, zm1;           // the main run within them; the other planes of [zv0, zv1) hold identity rows only
    int zs0, zs1;           // planes whose x is fetched and staged like a main plane's: the main run, and a GHOST plane next to it whose
                            // eliminated nodes are the main planes' (a sharded slab: the neighbour rank's boundary plane - data, not rim)
    int z0, z1, zchunk, tiles_x, tiles_y;
    int qq;
    int whatif;             // instrumented builds only (PGD_STENCIL_TIMING): 1 no y stores, 2 no x fetches, 4 no LDS reads / FMAs
    const double *b;        // epilogues of the multigrid passes (EPI 1: y = x - w A x;  EPI 2: y = x + w (b - A x), fused dot b . y)
    double w;
};

#if defined(PGD_STENCIL_TIMING) || defined(PGD_STENCIL_WHATIF)
#define PGD_ST_WHATIF(bit) (A.whatif & (bit))
#else
#define PGD_ST_WHATIF(bit) 0
#endif

typedef int st_v2i __attribute__((ext_vector_type(2)));

// RWT rows per thread: 4 (a 64 x 16 patch) or 2 (64 x 8: twice the patches per plane - thin z-slabs of a sharded solve then fill the
// chip with marches twice as long: a 256 x 256 x 32 slab is 4 marches of 8 planes instead of 8 of 4, r04)
template <bool DOT, bool STORE, int D, bool NTY, int OCC, int EPI = 0, int RWT = 4>
__global__ __launch_bounds__(256, OCC) void k_spmv_stencil_march(StencilArgs A) {
    static_assert(RWT == 4 || RWT == 2, "rows per thread");
    constexpr int NT = 256, RW = RWT, PY = 4 * RW, HY = PY + 2;             // 64 x 16 patch, four rows per thread (64 x 8, two)
    constexpr int NQ = RW + 1;                                              // cells a thread stages per plane: its own rows + one halo cell
    constexpr int SLOT = NQ * NT;                                           // 1280 >= 66 * 18 = 1188 cells (768 >= 66 * 10 = 660) + dump cells of idle stagers
    constexpr int OOB = (int)0x40000000;                                    // byte offset no plane reaches (planes < 2^27 rows: launcher)
    constexpr int AUX_ST = NTY ? 2 : 0;                                     // nt
    __shared__ double s_x[4 * SLOT];
    __shared__ double s_red[4];
    if (A.flags && A.flags[0]) return;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int b = xcd_remap(blockIdx.x, gridDim.x);
    const int per_chunk = A.tiles_x * A.tiles_y;
    const int chunk = b / per_chunk, tile = b - chunk * per_chunk;
    const int ty = tile / A.tiles_x, tx = tile - ty * A.tiles_x;
    const int x0 = tx * 64, y0 = ty * PY;
    const int64_t P = (int64_t)A.nx * A.ny;
    const unsigned P8 = (unsigned)(P * 8);
    const int centre = (RW * wv + 1) * DM_HX + lane + 1;                    // row 0 of the thread's four; row r at + r * DM_HX
    const int za = A.z0 + chunk * A.zchunk, zb = min(A.z1, za + A.zchunk);
    if (za >= zb) {                                                         // uniform; the launcher sizes the grid so that no chunk is empty
        if (DOT && tid == 0) { if (A.qq) { A.partials[2 * b] = 0.0; A.partials[2 * b + 1] = 0.0; } else A.partials[b] = 0.0; }
        return;
    }
    // ---- per-thread state of the whole march, from the codes of one main plane
    // cells this thread stages: q = 0..3 its own rows, q = 4 one cell of the halo ring (threads 164.. : a dump cell)
    int lv[NQ], sv[RW], cell4;                                              // byte offsets of the loads / stores within a plane (or OOB)
    double m[RW];                                                           // 1: the row is live and free, 0: eliminated or outside the grid
    unsigned fix = 0;                                                       // bit r: row r is live and an identity row of the main planes
    {
        const uint8_t *cz = A.cls + P * min(max(A.zm0, 0), A.nz - 1);
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            int lx, ly;
            if (q < RW) { lx = lane + 1; ly = RW * wv + q + 1; }
            else if (tid < DM_HX) { lx = tid; ly = 0; }
            else if (tid < 2 * DM_HX) { lx = tid - DM_HX; ly = HY - 1; }
            else if (tid < 2 * DM_HX + PY) { lx = 0; ly = tid - 2 * DM_HX + 1; }
            else { lx = DM_HX - 1; ly = tid - 2 * DM_HX - PY + 1; }
            const bool ring = q < RW || tid < 2 * DM_HX + 2 * PY;
            const int gx = x0 - 1 + lx, gy = y0 - 1 + ly;
            const bool in = ring && gx >= 0 && gx < A.nx && gy >= 0 && gy < A.ny;
            const int off = in ? gx + A.nx * gy : 0;
            const bool masked = in && A.ident >= 0 && (int)cz[off] == A.ident;
            if (q < RW) {
                lv[q] = in ? 8 * off : OOB;                                 // raw: an identity row's own value is its y
                sv[q] = lv[q];                                              // every live row is stored: its sum, or its x
                m[q] = (in && !masked) ? 1.0 : 0.0;
                if (in && masked) fix |= 1u << q;
            } else {
                lv[q] = (in && !masked) ? 8 * off : OOB;                    // the halo cell of an eliminated node reads as 0
                cell4 = ring ? ly * DM_HX + lx : DM_HX * HY + (tid - (2 * DM_HX + 2 * PY));
            }
        }
    }
    bool fixr[RW];
#pragma unroll
    for (int r = 0; r < RW; ++r) fixr[r] = ((fix >> r) & 1u) != 0;
    const double c0 = A.c[0], c1 = A.c[1], c2 = A.c[2], c3 = A.c[3], c4 = A.c[4], c5 = A.c[5], c6 = A.c[6], c7 = A.c[7];
    double dot = 0.0, dot2 = 0.0;
    // A plane is read and written through a descriptor of its own - base + one plane of records, none at all where the plane
    // must not be touched - built from RUNNING pointers: a step's scalar work is two 64-bit additions, not three 64-bit
    // multiplications (every wave executes the scalar stream of its workgroup again, in order with its vector instructions).
    auto rsrc_of = [&](const double *plane_ptr, bool live) {
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(plane_ptr), 0, live ? (int)P8 : 0, 0x00020000);
    };
    auto fetch = [&](const double *xz, bool live, double (&v)[NQ]) {
        if (PGD_ST_WHATIF(2)) return;
        const auto r = rsrc_of(xz, live);
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const st_v2i t = __builtin_amdgcn_raw_buffer_load_b64(r, lv[q], 0, 0);
            v[q] = __builtin_bit_cast(double, t);
        }
    };
    const int zd0 = min(A.zv0, A.zs0), zd1 = max(A.zv1, A.zs1);             // planes that hold data: the verified ones and such ghost planes
    auto fetch_live = [&](int z) { return z >= zd0 && z < zd1 && z <= zb; };          // (planes behind the march's upper halo plane: no records)
    // stage plane zz (values fetched D + 2 steps ago): the masked values for the neighbours, the raw own values into `own`
    auto stage_main = [&](int zz, const double (&v)[NQ], double (&own)[RW]) {
        double *dst = s_x + (zz & 3) * SLOT;
#pragma unroll
        for (int q = 0; q < RW; ++q) { dst[centre + q * DM_HX] = v[q] * m[q]; own[q] = v[q]; }
        dst[cell4] = v[RW];
    };
    auto stage = [&](int zz, const double (&v)[NQ], double (&own)[RW]) {
        if (zz >= A.zs0 && zz < A.zs1) { stage_main(zz, v, own); return; }  // uniform
        // identity rows only (or outside the verified planes: nothing was fetched): zeros for the neighbours
        double *dst = s_x + (zz & 3) * SLOT;
#pragma unroll
        for (int q = 0; q < RW; ++q) { dst[centre + q * DM_HX] = 0.0; own[q] = v[q]; }
        dst[cell4] = 0.0;
    };
    const double *xq = A.x + P * (int64_t)(za - 1);                         // running pointers: the next plane to fetch ...
    double *yq = A.y + P * (int64_t)(za - 1);                               // ... and the plane being staged / multiplied
    const double *bq = EPI == 2 ? A.b + P * (int64_t)za : nullptr;          // (EPI 2: the right-hand side's plane beside it)
    double rr[D][NQ];                                                       // the D plane fetches in flight, rotating by name
    double own[3][RW];                                                      // raw own values of the planes z, z + 1, z + 2: own[(plane - za) % 3]
    static_assert(D % 3 == 0, "the ring of own values rotates by name: the march is unrolled in multiples of three steps");
    {
        double t0[NQ], t1[NQ], t2[NQ];
        fetch(xq, fetch_live(za - 1), t0); fetch(xq + P, fetch_live(za), t1); fetch(xq + 2 * P, fetch_live(za + 1), t2);
        stage(za - 1, t0, own[2]); stage(za, t1, own[0]); stage(za + 1, t2, own[1]);
    }
    xq += 3 * P;
#pragma unroll
    for (int s = 0; s < D; ++s) { fetch(xq, fetch_live(za + 2 + s), rr[s]); xq += P; }
    yq += P;                                                                // = y + P za
    __syncthreads();
#ifdef PGD_STENCIL_TIMING
    // debug build (tools/stencil_timing.py): s_memtime stamps of four workgroups' wave 0 at the phases of every step
    long long st_prev = 0;
    const int st_slot = b == 0 ? 0 : b == 100 ? 1 : b == 300 ? 2 : b == (int)gridDim.x - 1 ? 3 : -1;
    int st_n = 0;
#define PGD_ST_STAMP(k)                                                                                   \
    do {                                                                                                  \
        if (st_slot >= 0 && tid == 0 && st_n < 48 * 4) {                                                  \
            long long t_;                                                                                 \
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                   \
            A.partials[4096 + st_slot * 256 + st_n] = (double)(t_ - st_prev);                             \
            st_prev = t_;                                                                                 \
            st_n += 1;                                                                                    \
        }                                                                                                 \
    } while (0)
#else
#define PGD_ST_STAMP(k) do { } while (0)
#endif
    // the product of the rows of plane z (a main plane), yz = y + P z, xo = the plane's raw own values
    auto rows = [&](int z, double *yz, bool live, const double (&xo)[RW]) {
        double bv[RW];
        if (EPI == 2) {                                                     // issued first: in flight behind the LDS reads and the chains
            const auto rb = rsrc_of(bq, live);
#pragma unroll
            for (int r = 0; r < RW; ++r) bv[r] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rb, sv[r], 0, 0));
        }
        const double *xm = s_x + ((z - 1) & 3) * SLOT + centre;
        const double *xc = s_x + (z & 3) * SLOT + centre;
        const double *xp = s_x + ((z + 1) & 3) * SLOT + centre;
        // ALL 36 neighbour values of the four rows first (18 ds_read2_b64 in flight behind the barrier), then the four chains of 15
        // fused multiply-adds side by side: left to itself the compiler reads just in time and runs one chain after the other -
        // 60 dependent DP operations with an LDS round trip between every few of them.
        double M[RW + 1][2], C[RW + 2][3], Q[RW + 1][2];
        double acc[RW];
        if (PGD_ST_WHATIF(4)) {
#pragma unroll
            for (int r = 0; r < RW; ++r) { acc[r] = c7; C[r + 1][1] = c6; }
        } else {
#pragma unroll
            for (int j = 0; j <= RW; ++j) { M[j][0] = xm[(j - 1) * DM_HX - 1]; M[j][1] = xm[(j - 1) * DM_HX]; }
#pragma unroll
            for (int j = 0; j <= RW + 1; ++j) {
                if (j <= RW) C[j][0] = xc[(j - 1) * DM_HX - 1];
                C[j][1] = xc[(j - 1) * DM_HX];
                if (j >= 1) C[j][2] = xc[(j - 1) * DM_HX + 1];
            }
#pragma unroll
            for (int j = 0; j <= RW; ++j) { Q[j][0] = xp[j * DM_HX]; Q[j][1] = xp[j * DM_HX + 1]; }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int r = 0; r < RW; ++r) acc[r] = c7 * M[r][0];
#define PGD_ST_ROWS(cc, V)                           \
    _Pragma("unroll") for (int r = 0; r < RW; ++r) acc[r] = fma(cc, V, acc[r])
            PGD_ST_ROWS(c6, M[r][1]);
            PGD_ST_ROWS(c5, M[r + 1][0]);
            PGD_ST_ROWS(c4, M[r + 1][1]);
            PGD_ST_ROWS(c3, C[r][0]);
            PGD_ST_ROWS(c2, C[r][1]);
            PGD_ST_ROWS(c1, C[r + 1][0]);
            PGD_ST_ROWS(c0, C[r + 1][1]);
            PGD_ST_ROWS(c1, C[r + 1][2]);
            PGD_ST_ROWS(c2, C[r + 2][1]);
            PGD_ST_ROWS(c3, C[r + 2][2]);
            PGD_ST_ROWS(c4, Q[r][0]);
            PGD_ST_ROWS(c5, Q[r][1]);
            PGD_ST_ROWS(c6, Q[r + 1][0]);
            PGD_ST_ROWS(c7, Q[r + 1][1]);
#undef PGD_ST_ROWS
        }
        const auto ry = rsrc_of(yz, live && STORE && !PGD_ST_WHATIF(1));
#pragma unroll
        for (int r = 0; r < RW; ++r) {
            const double ev = EPI == 0 ? acc[r] : EPI == 1 ? fma(-A.w, acc[r], xo[r]) : fma(A.w, bv[r] - acc[r], xo[r]);
            const double yv = fixr[r] ? xo[r] : ev;            // an eliminated row: y = x, in the same full-line store as its neighbours' sums
            if (STORE) __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(st_v2i, yv), ry, sv[r], 0, AUX_ST);
            if (DOT && EPI == 2) dot = fma(yv, bv[r], dot);    // (rows outside the grid: b = 0 was loaded)
            else if (DOT) {                                    // (a row outside the grid: x = 0 was loaded for it, and its sum is one of zeros)
                dot = fma(yv, xo[r], dot);
                const double t = sv[r] != OOB ? yv : 0.0;
                dot2 = fma(t, t, dot2);
            }
        }
    };
    // a plane of identity rows only: y = x out of the registers
    auto copy_rows = [&](double *yz, const double (&xo)[RW]) {
        const auto ry = rsrc_of(yz, STORE);
#pragma unroll
        for (int r = 0; r < RW; ++r) {
            if (STORE) __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(st_v2i, xo[r]), ry, sv[r], 0, AUX_ST);
            if (DOT && EPI == 0) { dot = fma(xo[r], xo[r], dot); dot2 = fma(xo[r], xo[r], dot2); }      // (EPI 2: b . y with y = x = 0 there)
        }
    };
    // PURE: every plane the march touches - its own, its two halo planes - is a main plane: no plane tests in the steps
    const bool pure = za - 1 >= A.zs0 && zb < A.zs1;                         // (zs = zm + at most one ghost plane either side: [za, zb) lies in the main run)
    int z = za;
    if (pure) {
#pragma clang loop unroll(disable)
        for (; z + D <= zb; z += D) {
#pragma unroll
            for (int s = 0; s < D; ++s) {
                // one step: rows of plane z + s; rr[s] holds plane z + s + 2 (staged now) and then takes the fetch of z + s + 2 + D
                rows(z + s, yq, true, own[s % 3]);
                PGD_ST_STAMP(0);
                stage_main(z + s + 2, rr[s], own[(s + 2) % 3]);      // slot (z + 2) & 3 held plane z - 2, which nobody reads any more
                PGD_ST_STAMP(1);
                fetch(xq, z + s + 2 + D <= zb, rr[s]);
                PGD_ST_STAMP(2);
                xq += P; yq += P;
                if (EPI == 2) bq += P;
                if (!PGD_ST_WHATIF(8)) lds_barrier();
                PGD_ST_STAMP(3);
            }
        }
    }
    // the same steps with every plane test in them: marches that touch planes of identity rows or the rim of the verified
    // planes, and the last (incomplete) group of steps of any march
#pragma clang loop unroll(disable)
    for (; z < zb; z += D) {
#pragma unroll
        for (int s = 0; s < D; ++s) {
            const int zs = z + s;
            if (zs < zb) {                                                    // uniform (idle steps behind the march's last plane)
                if (zs >= A.zm0 && zs < A.zm1) rows(zs, yq, true, own[s % 3]);
                else if (zs >= A.zv0 && zs < A.zv1) copy_rows(yq, own[s % 3]);
            }
            stage(zs + 2, rr[s], own[(s + 2) % 3]);
            fetch(xq, fetch_live(zs + 2 + D), rr[s]);
            xq += P; yq += P;
            if (EPI == 2) bq += P;
            lds_barrier();
        }
    }
    if (DOT) {
        for (int pass = 0; pass < (A.qq ? 2 : 1); ++pass) {
            const double sum = wave_sum(pass ? dot2 : dot);
            __syncthreads();
            if (lane == 0) s_red[wv] = sum;
            __syncthreads();
            if (tid == 0) A.partials[A.qq ? 2 * b + pass : b] = (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
        }
    }
}
#undef PGD_ST_STAMP
#undef PGD_ST_WHATIF

// Does the dictionary reduce to one stencil?  k_stencil_pick: base class = the class of the row in the middle of the planes
// [z_lo, z_hi), identity class = the tuple (1, 0 .. 0).  A row's tuple holds only the UPPER half of its row: a free node whose
// upper neighbours are all eliminated (the last free node before three faces) carries the identity tuple too.  k_stencil_split
// gives those rows a class id of their own (same tuple, a duplicate table row: the dictionary product does not care), so that
// "class == identity class" means an eliminated node: every coupling TO it is zero as well.  k_stencil_verify: every row of those
// planes, every slot, bit for bit.
struct StencilInfo { int ok, ident, base, split; double c[8]; int zm0, zm1, g_lo, g_hi; };     // g_lo / g_hi: the plane below / above the verified planes is a ghost DATA plane (k_stencil_ghost)

__global__ void k_stencil_pick(const uint8_t *__restrict__ cls, double *__restrict__ table, const int *__restrict__ info,
                               int nx, int ny, int z_lo, int z_hi, StencilInfo *S) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    S->ok = 0; S->ident = -1; S->base = -1; S->split = 0;
    const int ncls = info[0];
    if (info[1] != 0 || ncls <= 0 || ncls >= CLS_MAX || z_hi <= z_lo) return;      // (room for one more class id)
    const int64_t mid = (int64_t)nx * ny * ((z_lo + z_hi) / 2) + (int64_t)nx * (ny / 2) + nx / 2;
    for (int k = 0; k < ncls; ++k) {
        bool id = table[k * 8 + cls_pos(0)] == 1.0;
        for (int s = 1; s < 8; ++s) id = id && table[k * 8 + cls_pos(s)] == 0.0;
        if (id) S->ident = k;
    }
    // base class: the tuple with the most couplings (the row in the middle of a thin grid sits next to a face); the middle row's on ties
    int base = cls[mid], best = -1;
    for (int k = 0; k < ncls; ++k) {
        int nzc = 0;
        for (int s = 1; s < 8; ++s) nzc += table[k * 8 + cls_pos(s)] != 0.0 ? 1 : 0;
        if (nzc > best || (nzc == best && k == (int)cls[mid])) { best = nzc; base = k; }
    }
    if (base == S->ident) return;                            // (nothing but identity rows: keep the dictionary form)
    S->base = base;
    for (int s = 0; s < 8; ++s) S->c[s] = table[base * 8 + cls_pos(s)];
    if (S->ident >= 0) {                                     // the duplicate row of the free nodes with the identity tuple, then the zero class
        for (int s = 0; s < 8; ++s) { table[ncls * 8 + s] = table[S->ident * 8 + s]; table[(ncls + 1) * 8 + s] = 0.0; }
    }
    S->ok = 1;
}

__global__ __launch_bounds__(TPB) void k_stencil_split(uint8_t *__restrict__ cls, const double *__restrict__ table, const int *__restrict__ info,
                                                       int nx, int ny, int64_t nv, StencilInfo *S) {
    if (!S->ok || S->ident < 0) return;
    const int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x;
    if (i >= nv || (int)cls[i] != S->ident) return;
    const int64_t P = (int64_t)nx * ny;
    const int z = (int)(i / P), rem = (int)(i - (int64_t)z * P), y = rem / nx, x = rem - y * nx;
    bool coupled = false;
#pragma unroll
    for (int s = 1; s < 8; ++s) {                            // the lower neighbour whose slot s points at this row
        const int dx = s & 1, dy = (s >> 1) & 1, dz = s >> 2;
        if (x - dx < 0 || y - dy < 0 || z - dz < 0) continue;
        const int kc = cls[i - dx - (int64_t)nx * dy - P * dz];      // (a neighbour re-coded meanwhile carries the same tuple)
        coupled = coupled || table[kc * 8 + cls_pos(s)] != 0.0;
    }
    if (coupled) { cls[i] = (uint8_t)info[0]; S->split = 1; }
}

__global__ void k_stencil_commit(double *__restrict__ table, int *__restrict__ info, StencilInfo *S) {
    if (threadIdx.x != 0 || blockIdx.x != 0 || !S->ok || S->ident < 0) return;
    const int n = info[0];
    if (S->split) info[0] = n + 1;                           // one more class; its zero class sits behind it
    else for (int s = 0; s < 8; ++s) table[n * 8 + s] = 0.0;      // unused: row n is the zero class again
}

// allid[z] = 1 iff plane z holds identity rows only; samem[z] = 1 iff the identity rows of plane z sit where those of plane z - 1 do
__global__ __launch_bounds__(TPB) void k_stencil_allid(const uint8_t *__restrict__ cls, int64_t plane, const StencilInfo *S, int *__restrict__ allid,
                                                       int *__restrict__ samem) {
    __shared__ int s_other, s_diff;
    const int z = blockIdx.x;
    if (threadIdx.x == 0) { s_other = 0; s_diff = 0; }
    __syncthreads();
    const uint8_t *a = cls + plane * z;
    const int ident = S->ident;
    bool other = false, diff = z == 0;
    for (int64_t i = threadIdx.x; i < plane; i += TPB) {
        const bool id = (int)a[i] == ident;
        other = other || !id;
        if (z > 0) diff = diff || id != ((int)a[i - plane] == ident);
    }
    if (other) s_other = 1;
    if (diff) s_diff = 1;
    __syncthreads();
    if (threadIdx.x == 0) { allid[z] = s_other ? 0 : 1; samem[z] = s_diff ? 0 : 1; }
}

// the planes of [z_lo, z_hi) must read [identity planes]* [one run of planes with the SAME identity rows] [identity planes]*
__global__ void k_stencil_planes(const int *__restrict__ same, const int *__restrict__ allid, int z_lo, int z_hi, StencilInfo *S) {
    if (threadIdx.x != 0 || blockIdx.x != 0 || !S->ok) return;
    int a = z_lo, b = z_hi;
    while (a < z_hi && allid[a]) ++a;
    while (b > a && allid[b - 1]) --b;
    bool simple = true;
    for (int z = a; z < b; ++z) simple = simple && !allid[z] && (z == a || same[z] != 0);
    S->zm0 = a; S->zm1 = b;
    if (!simple) S->ok = 0;
}

// A sharded slab's ghost planes: the plane below z_lo (above z_hi - 1) lies inside the array, next to a main plane, and its
// eliminated nodes are exactly those of the main planes - then the march may stage it like a main plane (its rows hold the
// neighbour rank's x; couplings to its eliminated nodes are the exact zeros k_stencil_verify found in the owned rows next to it)
__global__ void k_stencil_ghost_init(StencilInfo *S, int z_lo, int z_hi, int nz, int zm0, int zm1) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    if (zm0 < 0) { zm0 = S->zm0; zm1 = S->zm1; }
    S->g_lo = (S->ok && z_lo > 0 && zm0 == z_lo && zm1 > zm0) ? 1 : 0;
    S->g_hi = (S->ok && z_hi < nz && zm1 == z_hi && zm1 > zm0) ? 1 : 0;
}
// The couplings between the LOWER ghost plane and the first owned plane live in the ghost rows' dz = 1 slots (half storage: a
// row holds its couplings to the rows behind it), which k_stencil_verify - owned rows only - never sees: they are compared here,
// bitwise, with the stencil's couplings (exact zeros where the target is eliminated or outside), so that the march applies
// c[s] to the neighbour rank's x on evidence, like everywhere else.  (The upper ghost plane is reached through
// the owned rows' own dz = 1 slots, which k_stencil_verify has compared.)
__global__ __launch_bounds__(TPB) void k_stencil_ghost(const uint8_t *__restrict__ cls, const double *__restrict__ table, int nx, int ny,
                                                       int z_lo, int z_hi, int zm0, StencilInfo *S) {
    const int64_t plane = (int64_t)nx * ny;
    const int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x;
    if (i >= plane || !S->ok) return;
    if (zm0 < 0) zm0 = S->zm0;
    const int ident = S->ident;
    const bool id = (int)cls[plane * zm0 + i] == ident;
    if (S->g_lo) {
        const int k = cls[plane * (z_lo - 1) + i];
        bool good = (k == ident) == id;
        if (good && k != ident) {
            const int y = (int)(i / nx), x = (int)(i - (int64_t)y * nx);
            const double *t = table + k * 8;
#pragma unroll
            for (int s = 4; s < 8; ++s) {
                const int dx = s & 1, dy = (s >> 1) & 1;
                const bool live = x + dx < nx && y + dy < ny && (int)cls[plane * z_lo + i + dx + (int64_t)nx * dy] != ident;
                const double v = t[cls_pos(s)];
                good = good && (live ? __double_as_longlong(v) == __double_as_longlong(S->c[s]) : v == 0.0);
            }
        }
        if (!good) S->g_lo = 0;
    }
    if (S->g_hi && ((int)cls[plane * z_hi + i] == ident) != id) S->g_hi = 0;
}

__global__ __launch_bounds__(TPB) void k_stencil_verify(const uint8_t *__restrict__ cls, const double *__restrict__ table, int nx, int ny,
                                                        int nz, int z_lo, int z_hi, StencilInfo *S) {
    if (!S->ok) return;
    const int64_t P = (int64_t)nx * ny;
    const int64_t i = P * z_lo + (int64_t)blockIdx.x * TPB + threadIdx.x;
    if (i >= P * z_hi) return;
    const int k = cls[i];
    if (k == S->ident) return;
    const int z = (int)(i / P), rem = (int)(i - (int64_t)z * P), y = rem / nx, x = rem - y * nx;
    const double *t = table + k * 8;
    bool good = __double_as_longlong(t[cls_pos(0)]) == __double_as_longlong(S->c[0]);
#pragma unroll
    for (int s = 1; s < 8; ++s) {
        const int dx = s & 1, dy = (s >> 1) & 1, dz = s >> 2;
        const bool inside = x + dx < nx && y + dy < ny && z + dz < nz;
        const bool live = inside && (int)cls[i + dx + (int64_t)nx * dy + P * dz] != S->ident;
        const double v = t[cls_pos(s)];
        good = good && (live ? __double_as_longlong(v) == __double_as_longlong(S->c[s]) : v == 0.0);
    }
    if (!good) S->ok = 0;
}

// The stencil march with an epilogue, over the whole lattice (pgd_mg.hip: the two stencil passes of a multigrid level):
//   epi 1: y = x - w A x;   epi 2: y = x + w (b - A x), dot: partial sums of b . y (one per workgroup, *nparts of them in c->partials).
// cls / ident as in the product: code byte per node, `ident` marks eliminated nodes; planes [zm0, zm1) share one pattern of
// eliminated nodes, the planes outside hold eliminated nodes only.  x must vanish on eliminated nodes (then y does).
// [z0, z1): the planes the pass computes (default: all) - a z-slab of a row-sharded lattice computes its owned planes only; the planes
// next to them inside the array (ghost planes, filled by the caller) are read like any other plane of the main run.
int launch_stencil_pass(Ctx *c, const uint8_t *cls, int ident, const double cst[8], int nx, int ny, int nz, int zm0, int zm1,
                        const double *x, const double *b, double *y, double w, int epi, bool dot, int *nparts, int z0, int z1) {
    if (z1 < 0) { z0 = 0; z1 = nz; }
    StencilArgs F;
    F.cls = cls; F.ident = ident; F.x = x; F.y = y; F.b = b; F.w = w; F.flags = c->flags;
    for (int s2 = 0; s2 < 8; ++s2) F.c[s2] = cst[s2];
    F.nx = nx; F.ny = ny; F.nz = nz; F.zv0 = 0; F.zv1 = nz; F.zm0 = zm0; F.zm1 = zm1; F.zs0 = zm0; F.zs1 = zm1; F.z0 = z0; F.z1 = z1;
    F.tiles_x = (nx + 63) / 64; F.tiles_y = (ny + 15) / 16;
    F.qq = 0; F.whatif = 0;
    const int64_t tiles = (int64_t)F.tiles_x * F.tiles_y, slots = (int64_t)c->stencil_wg_per_cu * c->num_cu;
    const int64_t marches = std::max<int64_t>(1, slots / tiles);
    const int zc = std::max(3, (int)((z1 - z0 + marches - 1) / marches));
    F.zchunk = zc;
    const int wgs = (int)(((z1 - z0 + zc - 1) / zc) * tiles);
    if (nparts) *nparts = wgs;
    if (dot) PGD_TRY(ensure_partials(c, std::max<int64_t>(c->partials_off + (int64_t)wgs, 4 * MAX_VEC_BLOCKS)));
    F.partials = c->partials + c->partials_off;
    if (epi == 1) k_spmv_stencil_march<false, true, 3, false, 2, 1><<<wgs, 256, 0, c->stream>>>(F);
    else if (epi == 2 && dot) k_spmv_stencil_march<true, true, 3, false, 2, 2><<<wgs, 256, 0, c->stream>>>(F);
    else if (epi == 2) k_spmv_stencil_march<false, true, 3, false, 2, 2><<<wgs, 256, 0, c->stream>>>(F);
    else return fail(c, PGD_ERR_INVALID, "launch_stencil_pass: epilogue %d", epi);
    c->kcount[KC_STENCIL_MARCH] += 1;
    PGD_LAUNCH_CHECK(c);
    return PGD_OK;
}

// --- the classification (per solve, after the slot arrays got their final values)
__device__ __forceinline__ unsigned long long cls_hash(const double *__restrict__ uvals, int64_t stride, int64_t i) {
    unsigned long long h = 0x9e3779b97f4a7c15ull;
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        unsigned long long v = (unsigned long long)__double_as_longlong(uvals[(int64_t)s * stride + i]);
        v *= 0xff51afd7ed558ccdull; v ^= v >> 32;
        h = (h ^ v) * 0xc4ceb9fe1a85ec53ull; h ^= h >> 29;
    }
    return h ? h : 1ull;
}

struct ClsScratch {
    unsigned long long keys[CLS_SLOTS];     // 0 = empty
    int rep[CLS_SLOTS];                     // lowest row with that key
    int id[CLS_SLOTS];                      // class id of the slot (rank of its rep row)
    int info[4];                            // [0] distinct keys, [1] != 0: no dictionary (too many classes / a hash collision)
};

__global__ __launch_bounds__(TPB) void k_cls_insert(const double *__restrict__ uvals, int64_t stride, int64_t nv, ClsScratch *S) {
    const int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x;
    if (*(volatile int *)&S->info[1]) return;
    const bool active = i < nv;
    const unsigned long long h = active ? cls_hash(uvals, stride, i) : 0ull;
    const int lane = threadIdx.x & 63;
    unsigned long long todo = __ballot(active);
    while (todo) {                                          // one insertion per distinct key of the wavefront
        if (*(volatile int *)&S->info[1]) return;           // too many classes already: no point in probing a full table
        const int leader = __ffsll((long long)todo) - 1;
        const unsigned long long h0 = __shfl(h, leader);
        todo &= ~__ballot(active && h == h0);
        if (lane != leader) continue;
        int slot = (int)(h0 & (CLS_SLOTS - 1));
        for (int probe = 0; probe < CLS_SLOTS; ++probe, slot = (slot + 1) & (CLS_SLOTS - 1)) {
            unsigned long long k = *(volatile unsigned long long *)&S->keys[slot];
            if (k == 0ull) {
                k = atomicCAS(&S->keys[slot], 0ull, h0);
                if (k == 0ull) {
                    k = h0;
                    if (atomicAdd(&S->info[0], 1) + 1 > CLS_MAX) atomicExch(&S->info[1], 1);
                }
            }
            if (k == h0) {
                if (*(volatile int *)&S->rep[slot] > (int)i) atomicMin(&S->rep[slot], (int)i);
                break;
            }
        }
    }
}

// class ids in the order of the representative rows (deterministic), the table in the products' layout
__global__ __launch_bounds__(CLS_SLOTS) void k_cls_table(const double *__restrict__ uvals, int64_t stride, ClsScratch *S,
                                                          double *__restrict__ table) {
    __shared__ int s_rep[CLS_SLOTS];
    const int t = threadIdx.x;
    const bool used = S->keys[t] != 0ull;
    s_rep[t] = used ? S->rep[t] : 0x7fffffff;
    __syncthreads();
    if (S->info[1]) return;
    int rank = 0;
    for (int k = 0; k < CLS_SLOTS; ++k) rank += (s_rep[k] < s_rep[t]) ? 1 : 0;
    S->id[t] = used ? rank : -1;
    if (used && rank <= CLS_MAX) {
#pragma unroll
        for (int s = 0; s < 8; ++s) table[rank * 8 + cls_pos(s)] = uvals[(int64_t)s * stride + s_rep[t]];
    }
    if (t < 8) table[min(S->info[0], CLS_MAX) * 8 + t] = 0.0;        // the zero class
}

__global__ __launch_bounds__(TPB) void k_cls_assign(const double *__restrict__ uvals, int64_t stride, int64_t nv, ClsScratch *S,
                                                    const double *__restrict__ table, uint8_t *__restrict__ cls) {
    const int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x;
    if (i >= nv || S->info[1]) return;
    const unsigned long long h = cls_hash(uvals, stride, i);
    int slot = (int)(h & (CLS_SLOTS - 1)), id = -1;
    for (int probe = 0; probe < CLS_SLOTS; ++probe, slot = (slot + 1) & (CLS_SLOTS - 1)) {
        const unsigned long long k = S->keys[slot];
        if (k == h) { id = S->id[slot]; break; }
        if (k == 0ull) break;
    }
    bool same = id >= 0 && id <= CLS_MAX;
    if (same) {
#pragma unroll
        for (int s = 0; s < 8; ++s)
            same = same && __double_as_longlong(uvals[(int64_t)s * stride + i]) == __double_as_longlong(table[id * 8 + cls_pos(s)]);
    }
    if (!same) { atomicExch(&S->info[1], 2); return; }       // two tuples under one key: keep the plain diagonal form
    cls[i] = (uint8_t)id;
}

// same[z] = 1 iff every row of plane z has the code of the row below it in plane z - 1 (same[0] = 0): the march keeps its
// couplings across such planes without looking at a code
__global__ __launch_bounds__(TPB) void k_cls_planes(const uint8_t *__restrict__ cls, int64_t plane, int nz, int *__restrict__ same) {
    __shared__ int s_diff;
    const int z = blockIdx.x;
    if (threadIdx.x == 0) s_diff = 0;
    __syncthreads();
    bool diff = z == 0;
    if (z > 0) {
        const uint8_t *a = cls + plane * z, *b = a - plane;
        for (int64_t i = threadIdx.x; i < plane; i += TPB) diff = diff || a[i] != b[i];
    }
    if (diff) s_diff = 1;
    __syncthreads();
    if (threadIdx.x == 0) same[z] = s_diff ? 0 : 1;
}

// ---- classification with known codes (Mesh::cls_cache)
__global__ void k_cls_table_known(const double *__restrict__ uvals, int64_t stride, int64_t nv, const int *__restrict__ reps, int ncls,
                                  double *__restrict__ table) {
    const int k = threadIdx.x;
    if (k < ncls) {
        const int64_t r = reps[k];
#pragma unroll
        for (int s = 0; s < 8; ++s) table[k * 8 + cls_pos(s)] = (r >= 0 && r < nv) ? uvals[(int64_t)s * stride + r] : 0.0;
    } else if (k == ncls) {
#pragma unroll
        for (int s = 0; s < 8; ++s) table[k * 8 + s] = 0.0;                 // the zero class
    }
}

__global__ __launch_bounds__(TPB) void k_cls_verify_known(const double *__restrict__ uvals, int64_t stride, int64_t nv,
                                                          const double *__restrict__ table, const uint8_t *__restrict__ cls, int ncls,
                                                          int *__restrict__ info) {
    const int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x;
    if (i >= nv) return;
    const int k = cls[i];
    bool same = k < ncls;
    if (same) {
#pragma unroll
        for (int s = 0; s < 8; ++s)
            same = same && __double_as_longlong(uvals[(int64_t)s * stride + i]) == __double_as_longlong(table[k * 8 + cls_pos(s)]);
    }
    if (!same) info[1] = 2;
}

struct ZeroPat { uint8_t b[256]; };

// the class-level relations of the stencil form for known codes: the identity tuple, and every class = the base tuple with
// exact zeros where (and only where) the structure - verified row by row when the codes were first seen - has them
// (`use`: the classes that occur on the verified planes - the incomplete ghost rows of a sharded slab have tuples of their own, which
// the full classification never holds against the stencil either: k_stencil_verify walks the verified planes only)
__global__ void k_stencil_known(const double *__restrict__ table, int ncls, int ident, int base, ZeroPat pat, ZeroPat use, StencilInfo *S) {
    __shared__ int s_bad;
    if (threadIdx.x == 0) s_bad = 0;
    __syncthreads();
    const int k = threadIdx.x;
    if (k < ncls && use.b[k]) {
        bool good = true;
        if (k == ident) {
            good = table[k * 8 + cls_pos(0)] == 1.0;
            for (int s = 1; s < 8; ++s) good = good && table[k * 8 + cls_pos(s)] == 0.0;
        } else {
            good = __double_as_longlong(table[k * 8 + cls_pos(0)]) == __double_as_longlong(table[base * 8 + cls_pos(0)]);
            for (int s = 1; s < 8; ++s) {
                const double v = table[k * 8 + cls_pos(s)];
                good = good && (((pat.b[k] >> s) & 1) ? v == 0.0
                                                      : __double_as_longlong(v) == __double_as_longlong(table[base * 8 + cls_pos(s)]));
            }
        }
        if (!good) s_bad = 1;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        S->ok = s_bad ? 0 : 1; S->ident = ident; S->base = base; S->split = 0;
        for (int s = 0; s < 8; ++s) S->c[s] = table[base * 8 + cls_pos(s)];
    }
}

// after a full classification: a representative row per class, the zero patterns
__global__ __launch_bounds__(TPB) void k_cls_reps(const uint8_t *__restrict__ cls, int64_t nv, int *__restrict__ reps) {
    const int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x;
    const bool active = i < nv;
    const int code = active ? (int)cls[i] : -1;
    // one atomic per distinct code of the wavefront (its lowest lane holds the lowest row), and none where the class already has
    // a representative further down: 16.7 M atomics on nine addresses took 188 ms
    unsigned long long todo = __ballot(active);
    const int lane = threadIdx.x & 63;
    while (todo) {
        const int leader = __ffsll((long long)todo) - 1;
        const int c0 = __shfl(code, leader);
        todo &= ~__ballot(active && code == c0);
        if (lane == leader && reps[c0] > (int)i) atomicMin(&reps[c0], (int)i);
    }
}

// Row classes of the operator's CURRENT slot values (a->uvals, diagonal form).  a->cls_count > 0 afterwards when the
// dictionary exists; any later change of the slot values must reset it (sym_scale, combine_dia, ensure_sym do).
int dia_classify(Ctx *c, const Mesh *m, Csr *a, int zrange_lo, int zrange_hi) {
    a->cls_count = 0;
    a->st_ok = false;
    a->st_g_lo = a->st_g_hi = false;
    if (!c->spmv_classes || m->sym_nx <= 0 || !(a->uvals && a->uvals_valid) || m->nv >= ((int64_t)1 << 31)) return PGD_OK;
    const int64_t plane = (int64_t)m->sym_nx * m->sym_ny;
    if (m->nv / plane < 3 || plane * 64 < c->spmv_grid_min_plane_bytes) return PGD_OK;          // no march on this grid anyway
    void *p;
    if (m->nv / plane > 65536) return PGD_OK;                                // (plane flags of the stencil check: room for 2^16 planes)
    if (!c->cls_scratch) { PGD_TRY(dev_alloc(c, &p, sizeof(ClsScratch) + sizeof(StencilInfo) + 2 * 65536 * sizeof(int))); c->cls_scratch = p; }
    const int nzp = (int)(m->nv / plane);
    const size_t same_bytes = ((size_t)nzp * sizeof(int) + 63) / 64 * 64;
    if (!a->cls) {
        a->cls_bytes = (size_t)m->nv + (size_t)(CLS_MAX + 1) * 8 * sizeof(double) + same_bytes + 64;
        PGD_TRY(dev_alloc(c, &p, a->cls_bytes));
        a->cls_table = (double *)p;                          // the table first (aligned), then the plane flags, then the codes
        a->cls_same = (int *)((uint8_t *)p + (size_t)(CLS_MAX + 1) * 8 * sizeof(double));
        a->cls = (uint8_t *)a->cls_same + same_bytes;
    }
    ClsScratch *S = (ClsScratch *)c->cls_scratch;
    hipStream_t st = c->stream;
    const int g = (int)((m->nv + TPB - 1) / TPB);
    int z_lo = 0, z_hi = nzp;
    if (zrange_lo >= 0) { z_lo = std::max(0, zrange_lo); z_hi = std::min(nzp, zrange_hi); }
    StencilInfo *SI = reinterpret_cast<StencilInfo *>(S + 1);
    const bool try_stencil = c->spmv_stencil && z_hi - z_lo >= 3;
    struct { int info[4]; StencilInfo si; } host;
    const int scaled_key = (a->uvals_scaled ? 1 : 0) + (a->uvals_unit ? 2 : 0) + (try_stencil ? 4 : 0);
    // ---- known structure: codes copied, table from the representative rows, EVERY row compared with its class bit by bit
    if (c->cls_cache_on) {
        for (size_t e = 0; e < m->cls_cache.size(); ++e) {
            Mesh::ClsCache &E = m->cls_cache[e];
            if (E.sig != a->bc_sig || E.scaled != scaled_key || E.z_lo != z_lo || E.z_hi != z_hi || !E.codes) continue;
            PGD_HIP(c, hipMemsetAsync(S->info, 0, sizeof S->info, st));
            PGD_HIP(c, hipMemcpyAsync(a->cls, E.codes, (size_t)m->nv, hipMemcpyDeviceToDevice, st));
            PGD_HIP(c, hipMemcpyAsync(a->cls_same, E.same, (size_t)nzp * sizeof(int), hipMemcpyDeviceToDevice, st));
            k_cls_table_known<<<1, 256, 0, st>>>(a->uvals, a->uvals_stride, m->nv, E.reps, E.ncls, a->cls_table);
            k_cls_verify_known<<<g, TPB, 0, st>>>(a->uvals, a->uvals_stride, m->nv, a->cls_table, a->cls, E.ncls, S->info);
            host.si.ok = 0;
            if (E.st_ok) {
                ZeroPat zp;
                memcpy(zp.b, E.zero_pat, sizeof zp.b);
                ZeroPat use;
                memcpy(use.b, E.in_range, sizeof use.b);
                k_stencil_known<<<1, 256, 0, st>>>(a->cls_table, E.ncls, E.ident, E.base, zp, use, SI);
                k_stencil_ghost_init<<<1, 1, 0, st>>>(SI, z_lo, z_hi, nzp, E.zm0, E.zm1);
                if (z_lo > 0 || z_hi < nzp) k_stencil_ghost<<<(int)((plane + TPB - 1) / TPB), TPB, 0, st>>>(a->cls, a->cls_table, m->sym_nx, m->sym_ny, z_lo, z_hi, E.zm0, SI);
            }
            host.info[1] = 1;
            PGD_HIP(c, hipMemcpyAsync(host.info, S->info, sizeof host.info, hipMemcpyDeviceToHost, st));
            if (E.st_ok) PGD_HIP(c, hipMemcpyAsync(&host.si, SI, sizeof(StencilInfo), hipMemcpyDeviceToHost, st));
            PGD_HIP(c, hipStreamSynchronize(st));
            PGD_LAUNCH_CHECK(c);
            if (getenv("PGD_DEBUG_CLS"))
                fprintf(stderr, "[dia_classify] known structure %zu of %zu: z %d..%d mismatches %d (info %d %d %d %d) stencil %d -> %d\n", e, m->cls_cache.size(),
                        z_lo, z_hi, host.info[1], host.info[0], host.info[1], host.info[2], host.info[3], (int)E.st_ok, host.si.ok);
            if (host.info[1] != 0 || (E.st_ok && !host.si.ok)) {          // not that structure after all: forget it, classify in full
                (void)hipFree(E.same);
                m->cls_cache.erase(m->cls_cache.begin() + (long)e);
                break;
            }
            E.used = ++m->cls_clock;
            c->cls_fast += 1;
            a->cls_count = E.ncls;
            if (E.st_ok) {
                a->st_ok = true;
                a->st_ident = E.ident;
                a->st_z0 = z_lo; a->st_z1 = z_hi; a->st_zm0 = E.zm0; a->st_zm1 = E.zm1;
                a->st_g_lo = host.si.g_lo != 0; a->st_g_hi = host.si.g_hi != 0;
                for (int s2 = 0; s2 < 8; ++s2) a->st_c[s2] = host.si.c[s2];
            }
            return PGD_OK;
        }
    }
    c->cls_full += 1;
    PGD_HIP(c, hipMemsetAsync(S->keys, 0, sizeof S->keys, st));
    PGD_HIP(c, hipMemsetAsync(S->rep, 0x7f, sizeof S->rep, st));
    PGD_HIP(c, hipMemsetAsync(S->info, 0, sizeof S->info, st));
    k_cls_insert<<<g, TPB, 0, st>>>(a->uvals, a->uvals_stride, m->nv, S);
    k_cls_table<<<1, CLS_SLOTS, 0, st>>>(a->uvals, a->uvals_stride, S, a->cls_table);
    k_cls_assign<<<g, TPB, 0, st>>>(a->uvals, a->uvals_stride, m->nv, S, a->cls_table, a->cls);
    // ... and whether the classes are ONE stencil with eliminated nodes (k_spmv_stencil_march) on the planes [z_lo, z_hi) - the
    // whole grid, or the owned planes of a sharded rank's slab, whose ghost-plane rows are incomplete by construction
    if (try_stencil) {
        k_stencil_pick<<<1, 64, 0, st>>>(a->cls, a->cls_table, S->info, m->sym_nx, m->sym_ny, z_lo, z_hi, SI);
        k_stencil_split<<<g, TPB, 0, st>>>(a->cls, a->cls_table, S->info, m->sym_nx, m->sym_ny, m->nv, SI);
        k_stencil_commit<<<1, 64, 0, st>>>(a->cls_table, S->info, SI);
    }
    k_cls_planes<<<nzp, TPB, 0, st>>>(a->cls, plane, nzp, a->cls_same);
    if (try_stencil) {
        const int64_t rows = plane * (z_hi - z_lo);
        k_stencil_verify<<<(int)((rows + TPB - 1) / TPB), TPB, 0, st>>>(a->cls, a->cls_table, m->sym_nx, m->sym_ny, nzp, z_lo, z_hi, SI);
        int *allid = reinterpret_cast<int *>(SI + 1);
        k_stencil_allid<<<nzp, TPB, 0, st>>>(a->cls, plane, SI, allid, allid + 65536);
        k_stencil_planes<<<1, 64, 0, st>>>(allid + 65536, allid, z_lo, z_hi, SI);
        k_stencil_ghost_init<<<1, 1, 0, st>>>(SI, z_lo, z_hi, nzp, -1, -1);
        if (z_lo > 0 || z_hi < nzp) k_stencil_ghost<<<(int)((plane + TPB - 1) / TPB), TPB, 0, st>>>(a->cls, a->cls_table, m->sym_nx, m->sym_ny, z_lo, z_hi, -1, SI);
    }
    host.info[0] = 0; host.info[1] = 1; host.si.ok = 0;
    PGD_HIP(c, hipMemcpyAsync(host.info, S->info, sizeof host.info, hipMemcpyDeviceToHost, st));
    if (try_stencil) PGD_HIP(c, hipMemcpyAsync(&host.si, SI, sizeof(StencilInfo), hipMemcpyDeviceToHost, st));
    PGD_HIP(c, hipStreamSynchronize(st));
    PGD_LAUNCH_CHECK(c);
    // (an operator without classes - a variable coefficient - costs 0.55 ms at 256^3 here: the insert pass gives up as soon as the
    // 256th class appears)
    if (host.info[1] != 0 || host.info[0] <= 0 || host.info[0] > CLS_MAX) return PGD_OK;
    a->cls_count = host.info[0];
    if (getenv("PGD_DEBUG_STENCIL"))
        fprintf(stderr, "[dia_classify] classes %d info1 %d try %d ok %d ident %d base %d c = %.17g %.17g %.17g %.17g %.17g %.17g %.17g %.17g  z [%d, %d)\n",
                host.info[0], host.info[1], (int)try_stencil, host.si.ok, host.si.ident, host.si.base, host.si.c[0], host.si.c[1], host.si.c[2],
                host.si.c[3], host.si.c[4], host.si.c[5], host.si.c[6], host.si.c[7], z_lo, z_hi);
    if (getenv("PGD_DEBUG_STENCIL")) fprintf(stderr, "[dia_classify] main run [%d, %d)\n", host.si.zm0, host.si.zm1);
    if (try_stencil && host.si.ok) {
        a->st_ok = true;
        a->st_ident = host.si.ident;
        a->st_z0 = z_lo; a->st_z1 = z_hi;
        a->st_zm0 = host.si.zm0; a->st_zm1 = host.si.zm1;
        a->st_g_lo = host.si.g_lo != 0; a->st_g_hi = host.si.g_hi != 0;
        for (int s2 = 0; s2 < 8; ++s2) a->st_c[s2] = host.si.c[s2];
    }
    // ---- remember the structure for the next operator with this signature (at most four per mesh, least recently used out)
    if (c->cls_cache_on) {
        const int ncls = a->cls_count;
        Mesh::ClsCache E;
        E.sig = a->bc_sig; E.scaled = scaled_key; E.z_lo = z_lo; E.z_hi = z_hi; E.ncls = ncls;
        const size_t same_b = ((size_t)nzp * sizeof(int) + 63) / 64 * 64, reps_b = 256 * sizeof(int);
        E.bytes = (size_t)m->nv + same_b + reps_b + 64;
        void *q = nullptr;
        if (hipMalloc(&q, E.bytes + PAD_BYTES) == hipSuccess) {
            E.same = (int *)q;
            E.reps = (int *)((uint8_t *)q + same_b);
            E.codes = (uint8_t *)q + same_b + reps_b;          // (freed through this pointer's base: keep the base first)
            // layout note: the allocation starts at E.same; hipFree takes that address
            double tab[(CLS_MAX + 1) * 8];
            bool okc = hipMemcpyAsync(E.codes, a->cls, (size_t)m->nv, hipMemcpyDeviceToDevice, st) == hipSuccess &&
                       hipMemcpyAsync(E.same, a->cls_same, (size_t)nzp * sizeof(int), hipMemcpyDeviceToDevice, st) == hipSuccess &&
                       hipMemsetAsync(E.reps, 0x7f, reps_b, st) == hipSuccess;
            int reps_rng[256];
            if (okc) {
                // (first the classes that occur on the verified planes - representatives within them, through the same kernel on
                // that range - then the representatives of ALL classes, which is what the entry keeps)
                const int64_t r0 = (int64_t)z_lo * plane, r1 = std::min<int64_t>((int64_t)z_hi * plane, m->nv);
                if (r1 > r0) k_cls_reps<<<(int)((r1 - r0 + TPB - 1) / TPB), TPB, 0, st>>>(a->cls + r0, r1 - r0, E.reps);
                okc = hipMemcpyAsync(reps_rng, E.reps, sizeof reps_rng, hipMemcpyDeviceToHost, st) == hipSuccess &&
                      hipStreamSynchronize(st) == hipSuccess && hipMemsetAsync(E.reps, 0x7f, reps_b, st) == hipSuccess;
            }
            if (okc) {
                k_cls_reps<<<g, TPB, 0, st>>>(a->cls, m->nv, E.reps);
                okc = hipMemcpyAsync(tab, a->cls_table, (size_t)(ncls + 1) * 8 * sizeof(double), hipMemcpyDeviceToHost, st) == hipSuccess &&
                      hipStreamSynchronize(st) == hipSuccess;
            }
            if (okc) {
                E.st_ok = a->st_ok; E.ident = a->st_ident; E.base = try_stencil ? host.si.base : -1; E.zm0 = a->st_zm0; E.zm1 = a->st_zm1;
                for (int k = 0; k < 256; ++k) E.in_range[k] = (k < ncls && reps_rng[k] >= 0 && reps_rng[k] < 0x7f000000) ? 1 : 0;
                for (int k = 0; k < 256; ++k) {
                    uint8_t b = 0;
                    if (k < ncls) for (int s2 = 1; s2 < 8; ++s2) if (tab[k * 8 + cls_pos(s2)] == 0.0) b |= (uint8_t)(1u << s2);
                    E.zero_pat[k] = b;
                }
                // (a structural zero of the BASE tuple itself is no pattern bit: such a slot is compared with the base value)
                if (E.base >= 0) for (int k = 0; k < ncls; ++k) E.zero_pat[k] &= (uint8_t)~E.zero_pat[E.base];
                E.codes = (uint8_t *)q + same_b + reps_b;
                E.used = ++m->cls_clock;
                // keep the allocation's base in `same` (first member of the block); free through it
                if (m->cls_cache.size() >= 4) {
                    size_t old = 0;
                    for (size_t e = 1; e < m->cls_cache.size(); ++e) if (m->cls_cache[e].used < m->cls_cache[old].used) old = e;
                    (void)hipFree(m->cls_cache[old].same);
                    m->cls_cache.erase(m->cls_cache.begin() + (long)old);
                }
                m->cls_cache.push_back(E);
            } else {
                (void)hipFree(q);
                (void)hipGetLastError();
            }
        } else {
            (void)hipGetLastError();
        }
    }
    return PGD_OK;
}

// every column of every row must be a grid neighbour (dx, dy, dz) in {-1, 0, 1}^3 of the row with |offset| one of the
// eight diagonals: only then is the diagonal form lossless.  Checked once per mesh.
__device__ __forceinline__ int dia_slot(int64_t d, int64_t nx, int64_t P, int *dx, int *dy, int *dz) {
    const int64_t z = d / P, rem = d - z * P, yy = rem / nx, xx = rem - yy * nx;
    *dx = (int)xx; *dy = (int)yy; *dz = (int)z;
    return (z <= 1 && yy <= 1 && xx <= 1) ? (int)(xx + 2 * yy + 4 * z) : -1;
}

__global__ __launch_bounds__(TPB) void k_dia_verify(const int *__restrict__ row_ptr, const int *__restrict__ cols, int64_t nv,
                                                    int nx, int ny, int nz, int *__restrict__ flags) {
    const int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x;
    if (i >= nv) return;
    const int64_t P = (int64_t)nx * ny;
    const int z = (int)(i / P), rem = (int)(i - (int64_t)z * P), y = rem / nx, x = rem - y * nx;
    bool ok = true, diag = false;
    for (int k = row_ptr[i]; k < row_ptr[i + 1]; ++k) {
        const int64_t d = (int64_t)cols[k] - i;
        int dx, dy, dz;
        const int s = dia_slot(d < 0 ? -d : d, nx, P, &dx, &dy, &dz);
        if (s < 0) { ok = false; continue; }
        if (d == 0) diag = true;
        if (d > 0 && !(x + dx < nx && y + dy < ny && z + dz < nz)) ok = false;
        if (d < 0 && !(x >= dx && y >= dy && z >= dz)) ok = false;
    }
    if (!ok || !diag) flags[0] = 1;
}

// CSR values -> the eight diagonals; also checks a_ij == a_ji to rounding (flags[1] counts violations)
__global__ __launch_bounds__(TPB) void k_csr_to_dia(const int *__restrict__ row_ptr, const int *__restrict__ cols,
                                                    const double *__restrict__ vals, int64_t nv, int nx, int ny,
                                                    int64_t stride, double *__restrict__ uvals, int *__restrict__ flags) {
    const int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x;
    if (i >= nv) return;
    const int64_t P = (int64_t)nx * ny;
    double out[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) out[s] = 0.0;
    bool asym = false;
    for (int k = row_ptr[i]; k < row_ptr[i + 1]; ++k) {
        const int col = cols[k];
        const int64_t d = (int64_t)col - i;
        int dx, dy, dz;
        if (d >= 0) {
            const int s = dia_slot(d, nx, P, &dx, &dy, &dz);
#pragma unroll
            for (int q = 0; q < 8; ++q) if (q == s) out[q] = vals[k];
        } else {
            double aji = 0.0;
            for (int t = row_ptr[col]; t < row_ptr[col + 1]; ++t) if (cols[t] == (int)i) aji = vals[t];
            const double aij = vals[k];
            if (fabs(aij - aji) > 1e-12 * (fabs(aij) + fabs(aji))) asym = true;
        }
    }
#pragma unroll
    for (int s = 0; s < 8; ++s) uvals[(int64_t)s * stride + i] = out[s];
    if (asym) atomicAdd(&flags[1], 1);
}

// slot s of row i holds a(i, i + off_s): times s_i s_{i + off_s}.  The diagonal of D^-1/2 A D^-1/2 is 1: it is SET to exactly 1
// (s_i^2 a_ii differs from it by the rounding of s_i, a relative perturbation of the operator at the level of the rounding of
// every product with it), and the products of the scaled recurrence do not load it (DiaArgs::unit_diag).
__global__ __launch_bounds__(TPB) void k_dia_scale(double *__restrict__ uvals, int64_t stride, const double *__restrict__ sc,
                                                   int64_t nv, int nx, int ny, int unit) {
    const int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x;
    if (i >= nv) return;
    const int64_t P = (int64_t)nx * ny;
    const double si = sc[i];
    uvals[i] = unit ? 1.0 : uvals[i] * si * si;
#pragma unroll
    for (int s = 1; s < 8; ++s) {
        const int64_t j = i + (s & 1) + (int64_t)nx * ((s >> 1) & 1) + P * (s >> 2);
        if (j < nv) uvals[(int64_t)s * stride + i] *= si * sc[j];      // slots without a neighbour hold 0 and keep it
    }
}

int build_sym_tables(Ctx *c, Mesh *m) {
    m->sym_w = 0;
    if (m->dict_count <= 0 || m->max_row > 15 || m->nnz == 0) return PGD_OK;
    void *p;
    int *ibuf = nullptr;       // rep[DICT_MAXP], flags[8]
    PGD_TRY(dev_alloc(c, &p, (DICT_MAXP + 8) * sizeof(int))); ibuf = (int *)p;
    int *rep = ibuf, *flags = ibuf + DICT_MAXP;
    PGD_TRY(dev_alloc(c, &p, (size_t)DICT_MAXP * 16 * sizeof(int))); m->sym_tab = (int *)p;
    hipStream_t st = c->stream;
    PGD_HIP(c, hipMemsetAsync(rep, 0x7f, DICT_MAXP * sizeof(int), st));
    PGD_HIP(c, hipMemsetAsync(flags, 0, 8 * sizeof(int), st));
    PGD_HIP(c, hipMemsetAsync(m->sym_tab, 0, (size_t)DICT_MAXP * 16 * sizeof(int), st));
    k_sym_rep<<<(int)((m->nv + TPB - 1) / TPB), TPB, 0, st>>>(m->pids, m->nv, rep);      // one thread per row, no grid cap
    k_sym_build<<<1, DICT_MAXP, 0, st>>>(m->row_ptr, m->cols, rep, m->dict_count, m->nv, m->sym_tab, flags);
    k_sym_verify<<<(int)((m->nv + TPB - 1) / TPB), TPB, 0, st>>>(m->row_ptr, m->cols, m->pids, m->sym_tab, m->nv, flags);
    std::vector<int> tab((size_t)m->dict_count * 16);
    int f = 1;
    PGD_HIP(c, hipMemcpyAsync(&f, flags, sizeof f, hipMemcpyDeviceToHost, st));
    PGD_HIP(c, hipMemcpyAsync(tab.data(), m->sym_tab, tab.size() * sizeof(int), hipMemcpyDeviceToHost, st));
    PGD_HIP(c, hipStreamSynchronize(st));
    (void)hipFree(ibuf);
    PGD_LAUNCH_CHECK(c);
    if (f != 0) { (void)hipFree(m->sym_tab); m->sym_tab = nullptr; return PGD_OK; }
    int maxu = 0;
    for (int q = 0; q < m->dict_count; ++q) maxu = std::max(maxu, tab[(size_t)q * 16] & 15);
    int maxl = 0;
    for (int q = 0; q < m->dict_count; ++q) maxl = std::max(maxl, (tab[(size_t)q * 16] >> 4) & 15);
    m->sym_w = (maxu <= 4 && maxl <= 4) ? 4 : 8;
    // sorted lower distances of the richest pattern
    int best = 0;
    for (int q = 1; q < m->dict_count; ++q)
        if (((tab[(size_t)q * 16] >> 4) & 15) > ((tab[(size_t)best * 16] >> 4) & 15)) best = q;
    std::vector<int> d;
    for (int k = 0; k < ((tab[(size_t)best * 16] >> 4) & 15); ++k) d.push_back(tab[(size_t)best * 16 + 8 + k]);
    std::sort(d.begin(), d.end());
    // a full structured vertex grid (row = x + nx y + nx ny z, the 15-point pattern of the 6-tetrahedra-per-cube mesh):
    // lower distances 1, nx, nx + 1, P, P + 1, P + nx, P + nx + 1 with P = nx ny dividing the row count
    m->sym_nx = m->sym_ny = 0;
    if (d.size() == 7 && d[0] == 1 && d[2] == d[1] + 1 && d[4] == d[3] + 1 && d[5] == d[3] + d[1] && d[6] == d[5] + 1 &&
        d[3] % d[1] == 0 && m->nv % d[3] == 0 && d[1] >= 2) {
        m->sym_nx = d[1];
        m->sym_ny = d[3] / d[1];
    }
    if (m->sym_nx > 0) {
        // the diagonal form needs every column of every row to be a grid neighbour on one of the eight diagonals
        const int nzg = (int)(m->nv / ((int64_t)m->sym_nx * m->sym_ny));
        int *vf = nullptr;
        PGD_TRY(dev_alloc(c, &p, 8 * sizeof(int))); vf = (int *)p;
        PGD_HIP(c, hipMemsetAsync(vf, 0, 8 * sizeof(int), st));
        k_dia_verify<<<(int)((m->nv + TPB - 1) / TPB), TPB, 0, st>>>(m->row_ptr, m->cols, m->nv, m->sym_nx, m->sym_ny, nzg, vf);
        int bad = 1;
        PGD_HIP(c, hipMemcpyAsync(&bad, vf, sizeof bad, hipMemcpyDeviceToHost, st));
        PGD_HIP(c, hipStreamSynchronize(st));
        (void)hipFree(vf);
        PGD_LAUNCH_CHECK(c);
        if (bad) m->sym_nx = m->sym_ny = 0;
    }
    return PGD_OK;
}

// slot s of row i holds a(i, i + off_s): times s_i s_{i + off_s}
template <int W>
__global__ __launch_bounds__(TPB) void k_sym_scale(double *__restrict__ uvals, int64_t stride, const uint16_t *__restrict__ pids,
                                                   const int *__restrict__ tab, const double *__restrict__ sc, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x;
    if (i >= n) return;
    const int *t = tab + (int)pids[i] * 16;
    const int ulen = t[0] & 15;
    const double si = sc[i];
#pragma unroll
    for (int s = 0; s < W; ++s)
        if (s < ulen) uvals[(int64_t)s * stride + i] *= si * sc[i + (s > 0 ? t[s] : 0)];
}

int sym_scale(Ctx *c, const Mesh *m, Csr *a, const double *sc) {
    if (!(a->uvals && a->uvals_valid) || a->uvals_scaled) return fail(c, PGD_ERR_INVALID, "sym_scale: no unscaled symmetric copy");
    a->cls_count = 0;
    const int g = (int)((m->nv + TPB - 1) / TPB);
    if (m->sym_nx > 0) k_dia_scale<<<g, TPB, 0, c->stream>>>(a->uvals, a->uvals_stride, sc, m->nv, m->sym_nx, m->sym_ny, c->spmv_unit_diag);
    else if (m->sym_w == 4) k_sym_scale<4><<<g, TPB, 0, c->stream>>>(a->uvals, a->uvals_stride, m->pids, m->sym_tab, sc, m->nv);
    else k_sym_scale<8><<<g, TPB, 0, c->stream>>>(a->uvals, a->uvals_stride, m->pids, m->sym_tab, sc, m->nv);
    PGD_LAUNCH_CHECK(c);
    a->uvals_scaled = true;
    a->uvals_unit = m->sym_nx > 0 && c->spmv_unit_diag;
    return PGD_OK;
}

int ensure_sym(Ctx *c, const Mesh *m, Csr *a, bool *usable) {
    *usable = false;
    if (!c->spmv_sym || m->sym_w == 0) return PGD_OK;
    if (a->uvals_valid && !a->uvals_scaled) { *usable = a->uvals != nullptr; return PGD_OK; }
    a->uvals_scaled = false;
    a->cls_count = 0;
    a->uvals_valid = true;                       // decided for this set of values, whatever the outcome
    const int64_t stride = m->nv;      // (padding the arrays apart - 2^27-byte strides at 256^3 - measured no difference)
    if (a->uvals && a->uvals_stride != stride) { dev_release(c, a->uvals, a->uvals_bytes); a->uvals = nullptr; }
    if (!a->uvals) {
        void *p;
        a->uvals_bytes = (size_t)m->sym_w * (size_t)stride * sizeof(double);
        PGD_TRY(dev_alloc(c, &p, a->uvals_bytes));
        a->uvals = (double *)p;
        a->uvals_stride = stride;
    }
    PGD_TRY(ensure_vals(c, m, a));
    a->cls_tried = false;
    PGD_HIP(c, hipMemsetAsync(c->flags + 4, 0, 4 * sizeof(int), c->stream));
    const int g = (int)((m->nv + TPB - 1) / TPB);
    if (m->sym_nx > 0) k_csr_to_dia<<<g, TPB, 0, c->stream>>>(m->row_ptr, m->cols, a->vals, m->nv, m->sym_nx, m->sym_ny, stride, a->uvals, c->flags + 4);
    else if (m->sym_w == 4) k_csr_to_sym<4><<<g, TPB, 0, c->stream>>>(m->row_ptr, a->vals, m->pids, m->sym_tab, m->nv, stride, a->uvals, c->flags + 4);
    else k_csr_to_sym<8><<<g, TPB, 0, c->stream>>>(m->row_ptr, a->vals, m->pids, m->sym_tab, m->nv, stride, a->uvals, c->flags + 4);
    int f[2] = {0, 0};
    PGD_HIP(c, hipMemcpyAsync(f, c->flags + 4, sizeof f, hipMemcpyDeviceToHost, c->stream));
    PGD_HIP(c, hipStreamSynchronize(c->stream));
    PGD_LAUNCH_CHECK(c);
    if (f[1] != 0) {      // not symmetric: keep the general kernel for this operator
        dev_release(c, a->uvals, a->uvals_bytes);
        a->uvals = nullptr;
        return PGD_OK;
    }
    *usable = true;
    return PGD_OK;
}

// Structured grids: the diagonal form of A = sum_t c_t A_t straight from the atoms' diagonal forms (built once per atom
// and kept), with the symmetric Dirichlet elimination and 1 / diagonal in the same pass - 8 (T + 1) n doubles of
// streaming instead of the per-solve conversion from CSR (k_csr_to_dia: 7.3 ms at 256^3, uncoalesced 180-byte rows and
// a search in every lower neighbour's row) and the diagonal search of k_diag_inv.  Same products in the same order as
// k_combine, so the slot values are bit-identical to the converted ones.
constexpr int DIA_MAXT = 8;
struct CombineDiaArgs {
    const double *in[DIA_MAXT];
    double coef[DIA_MAXT];
    int n;
};

__global__ __launch_bounds__(TPB) void k_combine_dia(CombineDiaArgs A, double *__restrict__ out, int64_t stride,
                                                     const uint8_t *__restrict__ mask, double *__restrict__ dinv, int64_t nv,
                                                     int nx, int ny) {
    const int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x;
    if (i >= nv) return;
    const int64_t P = (int64_t)nx * ny;
    const bool bi = mask && mask[i];
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        double v = 0.0;
#pragma unroll
        for (int t = 0; t < DIA_MAXT; ++t)
            if (t < A.n) v = fma(A.coef[t], A.in[t][(int64_t)s * stride + i], v);
        if (mask) {
            const int64_t j = i + (s & 1) + (int64_t)nx * ((s >> 1) & 1) + P * (s >> 2);
            const bool bj = j < nv && mask[j];
            if (s == 0) { if (bi) v = 1.0; }
            else if (bi || bj) v = 0.0;
        }
        out[(int64_t)s * stride + i] = v;
        if (s == 0 && dinv) dinv[i] = 1.0 / v;
    }
}

// mask: Dirichlet flags per row (or null) - only valid on the LAST pass, like k_combine's column mask
int combine_dia(Ctx *c, const Mesh *m, Csr *o, Csr *const *atoms, const double *coefs, int n, const uint8_t *mask) {
    o->uvals_valid = false;
    o->cls_count = 0;
    o->cls_tried = false;
    if (!c->spmv_sym || m->sym_nx <= 0 || !c->spmv_combine_dia) return PGD_OK;
    for (int t = 0; t < n; ++t) {
        bool usable = false;
        PGD_TRY(ensure_sym(c, m, atoms[t], &usable));     // once per atom: conversion + symmetry check, then cached
        if (!usable) return PGD_OK;                       // a non-symmetric atom: the operator keeps the CSR kernels
    }
    const int64_t stride = m->nv;
    if (o->uvals && o->uvals_stride != stride) { dev_release(c, o->uvals, o->uvals_bytes); o->uvals = nullptr; }
    if (!o->uvals) {
        void *p;
        o->uvals_bytes = (size_t)8 * (size_t)stride * sizeof(double);
        PGD_TRY(dev_alloc(c, &p, o->uvals_bytes));
        o->uvals = (double *)p;
        o->uvals_stride = stride;
    }
    if (!o->dinv) {
        void *p;
        o->dinv_bytes = (size_t)m->nv * sizeof(double);
        PGD_TRY(dev_alloc(c, &p, o->dinv_bytes));
        o->dinv = (double *)p;
    }
    const int g = (int)((m->nv + TPB - 1) / TPB);
    for (int t = 0, pass = 0; t < n; ++pass) {
        CombineDiaArgs A;
        int cnt = 0;
        if (pass > 0) { A.in[0] = o->uvals; A.coef[0] = 1.0; cnt = 1; }
        while (t < n && cnt < DIA_MAXT) { A.in[cnt] = atoms[t]->uvals; A.coef[cnt] = coefs[t]; ++cnt; ++t; }
        for (int k = cnt; k < DIA_MAXT; ++k) { A.in[k] = atoms[0]->uvals; A.coef[k] = 0.0; }
        A.n = cnt;
        const bool last = t >= n;
        k_combine_dia<<<g, TPB, 0, c->stream>>>(A, o->uvals, stride, last ? mask : nullptr, last ? o->dinv : nullptr, m->nv,
                                                m->sym_nx, m->sym_ny);
    }
    PGD_LAUNCH_CHECK(c);
    o->uvals_valid = true;
    o->uvals_scaled = false;
    o->dinv_valid = true;
    return PGD_OK;
}

// Products with an ATOM (immutable values) on a structured grid: once its diagonal form exists - pgd_op_combine converted
// it for the first operator it went into - plane-aligned products take the z-march instead of the CSR kernels (bit-identical
// y), and the atom's own row classes are looked for ONCE (0.65 ms at 256^3): mass and stiffness of a uniform grid have them, so
// the functionals and right-hand-side products of the fixed-point loop read one code byte per row like the PCG product.
bool atom_fast_form(Ctx *c, const Mesh *m, Csr *a, int64_t r0, int64_t r1) {
    if (!c->atom_fast || !c->spmv_sym || m->sym_nx <= 0 || !a->immutable || !(a->uvals && a->uvals_valid) || a->uvals_scaled) return false;
    const int64_t plane = (int64_t)m->sym_nx * m->sym_ny;
    if (r1 < 0) r1 = m->nv;
    if (r0 < 0 || r0 > r1 || r1 > m->nv || r0 % plane != 0 || r1 % plane != 0 || (r1 - r0) / plane < 3) return false;
    if (plane * 64 < c->spmv_grid_min_plane_bytes || c->spmv_zchunk <= 0) return false;
    if (!a->cls_tried) {
        a->cls_tried = true;
        if (dia_classify(c, m, a) != PGD_OK) { a->cls_count = 0; return false; }
    }
    return true;
}

// planes per march of the plain z-march over the planes [z0, z1) of a structured grid (0 marches: row order)
static int dia_march_chunks(const Ctx *c, const Mesh *m, int z0, int z1, int wy, int *zchunk_out) {
    const int64_t plane = (int64_t)m->sym_nx * m->sym_ny;
    const int tiles_x = (m->sym_nx + 63) / 64, tiles_y = (m->sym_ny + wy - 1) / wy;
    int zchunk = 0, chunks = 0;
    if (c->spmv_zchunk > 0 && plane * 64 >= c->spmv_grid_min_plane_bytes) {
        const int64_t tile_planes = (int64_t)tiles_x * tiles_y * (z1 - z0);
        zchunk = (int)std::min<int64_t>(c->spmv_zchunk, tile_planes * (wy / 4) / (4 * (int64_t)c->num_cu));
        if (c->spmv_zchunk_force > 0) zchunk = c->spmv_zchunk_force;       // tests: the march on any grid size
        chunks = (zchunk >= 3 || c->spmv_zchunk_force > 0) ? (z1 - z0 + zchunk - 1) / zchunk : 0;
    }
    *zchunk_out = zchunk;
    return chunks;
}

// would a product over ALL rows of this operator (a PCG product: w = x, or no dot) run in k_spmv_stencil_march?  The conditions of
// launch_spmv_op, for the solve that wants to hold the scaled operator as a stencil only (Csr::st_virtual)
bool stencil_whole_grid(const Ctx *c, const Mesh *m, const Csr *a) {
    if (!(c->spmv_sym && m->sym_w && a->uvals_valid && a->uvals) || m->sym_nx <= 0) return false;
    const int64_t plane = (int64_t)m->sym_nx * m->sym_ny;
    const int nz = (int)(m->nv / plane);
    int zchunk = 0;
    if (dia_march_chunks(c, m, 0, nz, c->spmv_variant <= 1 ? 8 : 4, &zchunk) <= 0) return false;
    const bool coded = c->spmv_variant == 0 && c->spmv_classes && a->cls_count > 0;
    return coded && c->spmv_stencil && a->st_ok && (c->spmv_zchunk_force <= 0 || c->spmv_zchunk_stencil > 0) &&
           a->st_z0 == 0 && a->st_z1 == nz && plane < ((int64_t)1 << 26);
}

// ... and over the whole planes [r0, r1) of a sharded slab, ghost planes as halo included
bool stencil_row_range(const Ctx *c, const Mesh *m, const Csr *a, int64_t r0, int64_t r1) {
    if (!(c->spmv_sym && m->sym_w && a->uvals_valid && a->uvals) || m->sym_nx <= 0) return false;
    const int64_t plane = (int64_t)m->sym_nx * m->sym_ny;
    if (r0 < 0 || r1 <= r0 || r1 > m->nv || r0 % plane != 0 || r1 % plane != 0) return false;
    const int nz = (int)(m->nv / plane), z0 = (int)(r0 / plane), z1 = (int)(r1 / plane);
    int zchunk = 0;
    if (dia_march_chunks(c, m, z0, z1, c->spmv_variant <= 1 ? 8 : 4, &zchunk) <= 0) return false;
    const bool coded = c->spmv_variant == 0 && c->spmv_classes && a->cls_count > 0;
    const bool st_lower = z0 == 0 ? a->st_z0 == 0 : (z0 - 1 >= a->st_z0 || (z0 == a->st_z0 && a->st_g_lo));
    const bool st_upper = z1 == nz ? a->st_z1 == nz : (z1 + 1 <= a->st_z1 || (z1 == a->st_z1 && a->st_g_hi));
    return coded && c->spmv_stencil && a->st_ok && (c->spmv_zchunk_force <= 0 || c->spmv_zchunk_stencil > 0) && st_lower && st_upper &&
           plane < ((int64_t)1 << 26);
}

int launch_spmv_op(Ctx *c, const Mesh *m, const Csr *a, const double *x, double *y, const double *w, int64_t r0,
                   int64_t r1, bool dot, bool store, const int *flags, int *nparts_out) {
    // (st_virtual: the stencil couplings describe D^-1/2 A D^-1/2 while the slot arrays and the CSR values hold A - inside
    // pgd_pcg_solve, which has checked that its products take the stencil form; anything else must not compute with the wrong numbers)
    auto not_virtual = [&]() -> int {
        return a->st_virtual ? fail(c, PGD_ERR_INVALID, "spmv: the operator is held as a scaled stencil; this launch cannot take that form") : PGD_OK;
    };
    if (!(c->spmv_sym && m->sym_w && a->uvals_valid && a->uvals)) {
        PGD_TRY(not_virtual());
        PGD_TRY(ensure_vals(c, m, const_cast<Csr *>(a)));
        return launch_spmv(c, m, a->vals, x, y, w, r0, r1, dot, store, flags, nparts_out);
    }
    if (r1 < 0) r1 = m->nv;
    if (r0 < 0 || r0 > r1 || r1 > m->nv) return fail(c, PGD_ERR_INVALID, "spmv: bad row range");
    const int64_t nrows = r1 - r0;
    const int nblk = (int)((nrows + 63) / 64);
    if (nparts_out) *nparts_out = nblk;
    if (nblk == 0) return PGD_OK;
    if (dot) PGD_TRY(ensure_partials(c, std::max<int64_t>(c->partials_off + 2 * (int64_t)nblk, 4 * MAX_VEC_BLOCKS)));
    SymArgs A;
    A.uvals = a->uvals; A.x = x; A.w = w; A.y = y; A.partials = c->partials + c->partials_off; A.flags = flags;
    A.tab = m->sym_tab; A.pids = m->pids; A.n = a->uvals_stride; A.row_begin = (int)r0; A.row_end = (int)r1;
    A.qq = (dot && c->spmv_qq) ? 1 : 0;
    if (m->sym_nx > 0) {
        // structured vertex grid: the operator is held in diagonal form
        const int64_t plane = (int64_t)m->sym_nx * m->sym_ny;
        DiaArgs D;
        D.uvals = a->uvals; D.x = x; D.w = w; D.y = y; D.partials = c->partials + c->partials_off; D.flags = flags; D.n = a->uvals_stride;
        D.nx = m->sym_nx; D.ny = m->sym_ny; D.nz = (int)(m->nv / plane);
        D.row_begin = (int)r0; D.row_end = (int)r1;
        D.nblk1 = -1; D.row_begin2 = D.row_end2 = 0;
        D.z0 = (int)(r0 / plane); D.z1 = (int)(r1 / plane);
        const int wy = c->spmv_variant <= 1 ? 8 : 4;        // patch rows (0: 4 waves x two rows per thread; 1: 8 waves; 2: 4 waves, one row)
        D.tiles_x = (D.nx + 63) / 64; D.tiles_y = (D.ny + wy - 1) / wy; D.zchunk = 0;
        D.unit_diag = (a->uvals_scaled && a->uvals_unit) ? 1 : 0;
        D.qq = (dot && c->spmv_qq) ? 1 : 0;                  // set by the single-sync recurrence around its product launches
        // plane-aligned row range, PCG product (w = x) or plain product, planes large enough: the LDS march
        int chunks = 0;
        if ((!dot || w == x) && r0 % plane == 0 && r1 % plane == 0) {
            // planes per march: as long as the launch still has ~4 workgroups per CU (measured on z-slabs of the 256 x 256
            // grid, tools/bench_spmv_slab.py: 30 planes - the interior of an 8-GPU rank - 31.9 us marching 8 planes vs 41.4 us
            // in row order; a march of fewer than 3 planes pays its prologue too often, and a single plane - the boundary
            // launches of the sharded solve - is faster in row order: 7.1 vs 8-14 us)
            chunks = dia_march_chunks(c, m, D.z0, D.z1, wy, &D.zchunk);
        }
        const int64_t gg = (int64_t)chunks * D.tiles_x * D.tiles_y;
        bool timed2 = false;
        if (gg > 0 && gg < ((int64_t)1 << 30)) {
            const int wgs = (int)gg;
            if (nparts_out) *nparts_out = wgs;
            if (dot) PGD_TRY(ensure_partials(c, std::max<int64_t>(c->partials_off + (D.qq ? 2 : 1) * (int64_t)wgs, 4 * MAX_VEC_BLOCKS)));
            D.partials = c->partials + c->partials_off;
            PGD_TRY(prof_begin(c, dot, store, &timed2));
#define PGD_MARCH(WY)                                                                                  \
    do {                                                                                               \
        if (dot && store) k_spmv_dia_march<true, true, WY><<<wgs, 64 * WY, 0, c->stream>>>(D);         \
        else if (dot) k_spmv_dia_march<true, false, WY><<<wgs, 64 * WY, 0, c->stream>>>(D);            \
        else k_spmv_dia_march<false, true, WY><<<wgs, 64 * WY, 0, c->stream>>>(D);                     \
    } while (0)
            bool coded = c->spmv_variant == 0 && c->spmv_classes && a->cls_count > 0;
            int wgs_c = wgs, zchunk_c = D.zchunk;
            if (coded && c->spmv_zchunk_force <= 0) {
                // the coded march is unrolled six (three) planes at a time and its steps are light: longer marches (fewer halo planes
                // and prologues per row; 256^3: 24 planes 81 us, 12 planes 89 us, 6 planes 103 us), in whole sixes (threes)
                const int64_t tile_planes = (int64_t)D.tiles_x * D.tiles_y * (D.z1 - D.z0);
                // (about one workgroup per CU is enough for this kernel - 128^3: marches of 18 planes = 256 workgroups 61.6 passes/s of
                // cfg3, 12 planes 59.9, 3 planes (four workgroups per CU, the rule of the plain march) 55.7, 24 planes 58.4)
                zchunk_c = (int)std::min<int64_t>(c->spmv_zchunk_coded, (tile_planes + c->num_cu / 2) / (int64_t)c->num_cu);
                zchunk_c = zchunk_c >= 9 ? std::max(12, (zchunk_c + 3) / 6 * 6) : std::max(3, zchunk_c / 3 * 3);
                // Large grids: a CU holds TWO of these workgroups (246 VGPRs), so with marches of 24 planes 256^3 is 1408 workgroups on
                // 512 slots - 2.75 rounds, the last one three quarters empty, and eleven prologues per column.  Where there is work for
                // at least 24 planes per slot the march is as long as it takes to fill every slot exactly once: 256^3 = 4 marches of 66
                // planes = 512 workgroups, 71 -> 63 us per product (8.87 -> 9.15 passes/s; one workgroup per CU - marches of 126 planes -
                // 102 us; 36 planes 68 us, 48 planes 72 us: `PGD_TUNE=21=...` before this rule)
                // (slabs of the sharded solve, tools/bench_coded_march.py 256x256xNZ: 32 planes 21.0 -> 17.9 us with 4 marches of 9,
                // 64 planes 29.9 -> 25.5 us with 4 of 18, 128 planes 41 us with 4 of 36 against 45-46 us with 18 or 24: the rule holds
                // from about 8 planes per slot on; marches under 12 planes run three steps at a time)
                const int64_t per_slot2 = (tile_planes + c->num_cu) / (2 * (int64_t)c->num_cu);
                if (c->spmv_zchunk_coded2 > 0 && per_slot2 >= 8)
                    zchunk_c = (int)std::min<int64_t>(c->spmv_zchunk_coded2, per_slot2 >= 12 ? (per_slot2 + 5) / 6 * 6 : (per_slot2 + 2) / 3 * 3);
                wgs_c = (D.z1 - D.z0 + zchunk_c - 1) / zchunk_c * D.tiles_x * D.tiles_y;
            }
            coded = coded && zchunk_c <= DIAC_MAXCHUNK;
            // one stencil + eliminated nodes (dia_classify verified every row of the planes this launch reads couplings of: its
            // own and the plane below its first): couplings in scalar registers, four rows per thread, no table
            // (tests force the march onto small grids with PGD_TUNE_SPMV_ZCHUNK_FORCE, which selects the dictionary kernel - unless
            // PGD_TUNE_SPMV_ZCHUNK_STENCIL names a march length for this one)
            // Every plane the launch READS - its own and one halo plane either way - must be a verified plane or lie outside the grid
            // (the kernel stages what is outside the verified planes as zeros: right for the rim of the grid, wrong for a ghost plane
            // of a sharded slab, whose rows hold the neighbour rank's x)
            // (... or a ghost plane whose eliminated nodes are the main planes': staged as data, k_stencil_ghost)
            const bool st_lower = D.z0 == 0 ? a->st_z0 == 0 : (D.z0 - 1 >= a->st_z0 || (D.z0 == a->st_z0 && a->st_g_lo));
            const bool st_upper = D.z1 == D.nz ? a->st_z1 == D.nz : (D.z1 + 1 <= a->st_z1 || (D.z1 == a->st_z1 && a->st_g_hi));
            if (coded && c->spmv_stencil && a->st_ok && (c->spmv_zchunk_force <= 0 || c->spmv_zchunk_stencil > 0) && st_lower && st_upper &&
                plane < ((int64_t)1 << 26)) {
                StencilArgs F;
                F.cls = a->cls; F.ident = a->st_ident; F.x = x; F.y = y; F.flags = flags;
                F.zv0 = a->st_z0; F.zv1 = a->st_z1; F.zm0 = a->st_zm0; F.zm1 = a->st_zm1;
                F.zs0 = a->st_zm0 - (a->st_g_lo ? 1 : 0); F.zs1 = a->st_zm1 + (a->st_g_hi ? 1 : 0);
                for (int s2 = 0; s2 < 8; ++s2) F.c[s2] = a->st_c[s2];
                F.nx = D.nx; F.ny = D.ny; F.nz = D.nz; F.z0 = D.z0; F.z1 = D.z1; F.tiles_x = D.tiles_x; F.tiles_y = (D.ny + 15) / 16;
                // rows per thread: four; two where the launch is so thin that marches of four-row patches would be shorter than 8
                // planes (a z-slab of a sharded solve: 256 x 256 x 32 = 8 marches of 4 planes against 4 of 8) - PGD_TUNE_STENCIL_ROWS
                int rows_per_thread = 4;
                {
                    const int64_t slots4 = (int64_t)c->stencil_wg_per_cu * c->num_cu, tiles4 = (int64_t)F.tiles_x * F.tiles_y;
                    const int64_t marches4 = std::max<int64_t>(1, slots4 / tiles4);
                    const int zc4 = (int)((D.z1 - D.z0 + marches4 - 1) / marches4);
                    if (c->stencil_rows == 2 || (c->stencil_rows == 0 && zc4 < 8 && D.ny >= 16 && c->spmv_zchunk_stencil <= 0 && c->stencil_depth != 6)) rows_per_thread = 2;
                }
                if (rows_per_thread == 2) F.tiles_y = (D.ny + 7) / 8;
                F.qq = D.qq;
                F.whatif = 0;
                F.b = nullptr; F.w = 0.0;
#if defined(PGD_STENCIL_TIMING) || defined(PGD_STENCIL_WHATIF)
                if (const char *wi = getenv("PGD_STENCIL_WHATIF")) F.whatif = atoi(wi);
#endif
                const bool nty = D.qq && c->pcg_stream_hints;
                // march length: every resident workgroup slot filled once (two workgroups per CU), whole groups of the fetch depth
                const int64_t tiles = (int64_t)F.tiles_x * F.tiles_y, slots = (int64_t)c->stencil_wg_per_cu * c->num_cu;
                const int planes = D.z1 - D.z0;
                int64_t marches = std::max<int64_t>(1, slots / tiles);
                int zc = (int)((planes + marches - 1) / marches);
                if (c->spmv_zchunk_stencil > 0) zc = c->spmv_zchunk_stencil;
                int depth = 3;                                 // (six plane fetches in flight: no faster, 60 registers more)
                if (c->stencil_depth > 0) depth = c->stencil_depth;
                zc = std::max(depth, zc);                      // (an incomplete last group of steps idles behind the march's last plane)
                F.zchunk = zc;
                const int wgs_s = (int)(((planes + zc - 1) / zc) * tiles);
                if (nparts_out) *nparts_out = wgs_s;
                if (dot) PGD_TRY(ensure_partials(c, std::max<int64_t>(c->partials_off + (D.qq ? 2 : 1) * (int64_t)wgs_s, 4 * MAX_VEC_BLOCKS)));
                F.partials = c->partials + c->partials_off;
#define PGD_STENCIL(DD, OC)                                                                                   \
    do {                                                                                                      \
        if (dot && store && nty) k_spmv_stencil_march<true, true, DD, true, OC><<<wgs_s, 256, 0, c->stream>>>(F);      \
        else if (dot && store) k_spmv_stencil_march<true, true, DD, false, OC><<<wgs_s, 256, 0, c->stream>>>(F);       \
        else if (dot) k_spmv_stencil_march<true, false, DD, false, OC><<<wgs_s, 256, 0, c->stream>>>(F);               \
        else k_spmv_stencil_march<false, true, DD, false, OC><<<wgs_s, 256, 0, c->stream>>>(F);                        \
    } while (0)
                if (rows_per_thread == 2 && depth != 6) {
                    if (dot && store && nty) k_spmv_stencil_march<true, true, 3, true, 2, 0, 2><<<wgs_s, 256, 0, c->stream>>>(F);
                    else if (dot && store) k_spmv_stencil_march<true, true, 3, false, 2, 0, 2><<<wgs_s, 256, 0, c->stream>>>(F);
                    else if (dot) k_spmv_stencil_march<true, false, 3, false, 2, 0, 2><<<wgs_s, 256, 0, c->stream>>>(F);
                    else k_spmv_stencil_march<false, true, 3, false, 2, 0, 2><<<wgs_s, 256, 0, c->stream>>>(F);
                } else if (depth == 6) PGD_STENCIL(6, 2); else PGD_STENCIL(3, 2);
#undef PGD_STENCIL
                c->kcount[KC_STENCIL_MARCH] += 1;
                if (timed2) PGD_TRY(prof_end(c, m, nrows, 16.0));
                PGD_LAUNCH_CHECK(c);
                return PGD_OK;
            }
            PGD_TRY(not_virtual());
            if (coded) {
                // the operator has a row-class dictionary (dia_classify): one byte per row instead of the slot values
                DiacArgs E;
                E.cls = a->cls; E.same = a->cls_same; E.table = a->cls_table; E.ncls = a->cls_count; E.x = x; E.y = y; E.partials = D.partials; E.flags = flags;
                E.nx = D.nx; E.ny = D.ny; E.nz = D.nz; E.z0 = D.z0; E.z1 = D.z1; E.zchunk = zchunk_c; E.tiles_x = D.tiles_x; E.tiles_y = D.tiles_y;
                E.qq = D.qq;
                E.nt_y = (D.qq && c->pcg_stream_hints) ? 1 : 0;      // the PCG's q: read once, by the vector update (+ 1 %)
                if (nparts_out) *nparts_out = wgs_c;
                if (dot) PGD_TRY(ensure_partials(c, std::max<int64_t>(c->partials_off + (D.qq ? 2 : 1) * (int64_t)wgs_c, 4 * MAX_VEC_BLOCKS)));
                E.partials = c->partials + c->partials_off;
                // six plane fetches in flight where the march is long enough to run in sixes (a light kernel is paced by the bytes
                // a CU keeps in flight), three otherwise
                const bool six = zchunk_c >= 12 && zchunk_c % 6 == 0 && c->spmv_fetch_depth != 3;
#define PGD_DIAC(DD)                                                                                       \
    do {                                                                                                   \
        if (dot && store && E.nt_y) k_spmv_diac_march2<true, true, DD, true><<<wgs_c, 256, 0, c->stream>>>(E);   \
        else if (dot && store) k_spmv_diac_march2<true, true, DD, false><<<wgs_c, 256, 0, c->stream>>>(E);       \
        else if (dot) k_spmv_diac_march2<true, false, DD, false><<<wgs_c, 256, 0, c->stream>>>(E);               \
        else k_spmv_diac_march2<false, true, DD, false><<<wgs_c, 256, 0, c->stream>>>(E);                        \
    } while (0)
                if (six) PGD_DIAC(6); else PGD_DIAC(3);
#undef PGD_DIAC
                c->kcount[KC_DIAC_MARCH] += 1;
                if (timed2) PGD_TRY(prof_end(c, m, nrows, 17.0));
                PGD_LAUNCH_CHECK(c);
                return PGD_OK;
            }
            if (c->spmv_variant == 0 && c->dia_march3 && (int64_t)64 * D.n < ((int64_t)1 << 31) && plane < ((int64_t)1 << 27)) {
                if (dot && store) k_spmv_dia_march3<true, true><<<wgs, 256, 0, c->stream>>>(D);
                else if (dot) k_spmv_dia_march3<true, false><<<wgs, 256, 0, c->stream>>>(D);
                else k_spmv_dia_march3<false, true><<<wgs, 256, 0, c->stream>>>(D);
            } else if (c->spmv_variant == 0) {
                if (dot && store) k_spmv_dia_march2<true, true><<<wgs, 256, 0, c->stream>>>(D);
                else if (dot) k_spmv_dia_march2<true, false><<<wgs, 256, 0, c->stream>>>(D);
                else k_spmv_dia_march2<false, true><<<wgs, 256, 0, c->stream>>>(D);
            } else if (wy == 8) PGD_MARCH(8); else PGD_MARCH(4);
#undef PGD_MARCH
            c->kcount[KC_DIA_MARCH] += 1;
        } else {
            PGD_TRY(not_virtual());
            PGD_TRY(prof_begin(c, dot, store, &timed2));
            if (dot && store) k_spmv_dia_rows<true, true><<<nblk, 64, 0, c->stream>>>(D);
            else if (dot) k_spmv_dia_rows<true, false><<<nblk, 64, 0, c->stream>>>(D);
            else k_spmv_dia_rows<false, true><<<nblk, 64, 0, c->stream>>>(D);
            c->kcount[KC_DIA_ROWS] += 1;
        }
        if (timed2) PGD_TRY(prof_end(c, m, nrows, 8.0 * (D.unit_diag ? 7 : 8) + 16));
        PGD_LAUNCH_CHECK(c);
        return PGD_OK;
    }
    PGD_TRY(not_virtual());
    const int grid = nblk;
    bool timed = false;
    PGD_TRY(prof_begin(c, dot, store, &timed));
#define PGD_SYM_LAUNCH(D, S)                                                          \
    do {                                                                              \
        if (m->sym_w == 4) k_spmv_sym<D, S, 4><<<grid, 64, 0, c->stream>>>(A);        \
        else k_spmv_sym<D, S, 8><<<grid, 64, 0, c->stream>>>(A);                      \
    } while (0)
    if (dot && store) PGD_SYM_LAUNCH(true, true);
    else if (dot) PGD_SYM_LAUNCH(true, false);
    else PGD_SYM_LAUNCH(false, true);
#undef PGD_SYM_LAUNCH
    c->kcount[KC_SYM_ROWS] += 1;
    if (timed) PGD_TRY(prof_end(c, m, nrows, 8.0 * m->sym_w + 18));
    PGD_LAUNCH_CHECK(c);
    return PGD_OK;
}

// Gram matrix of the Galerkin start of a PCG solve (fem._rescale_start): column j needs w = A v_j, then the dots
// v_i . w for i <= j and v_j . b - one pass over w and the v_i, partial sums per workgroup, fixed order.
constexpr int GRAM_MAXV = 17;
struct MultiDotArgs {
    const double *w, *b;
    const double *v[GRAM_MAXV];
    int nv;                      // v[0..nv): dots with w; the last one also with b
    int64_t lo, hi;
    double *partials;            // [block][nv + 1]
};

__global__ __launch_bounds__(TPB) void k_multidot(MultiDotArgs A) {
    __shared__ double s_red[4];
    double acc[GRAM_MAXV + 1];
#pragma unroll
    for (int m = 0; m <= GRAM_MAXV; ++m) acc[m] = 0.0;
    for (int64_t i = A.lo + (int64_t)blockIdx.x * TPB + threadIdx.x; i < A.hi; i += (int64_t)gridDim.x * TPB) {
        const double wi = A.w[i];
        double last = 0.0;
#pragma unroll
        for (int m = 0; m < GRAM_MAXV; ++m)
            if (m < A.nv) { last = A.v[m][i]; acc[m] = fma(last, wi, acc[m]); }
        acc[GRAM_MAXV] = fma(last, A.b[i], acc[GRAM_MAXV]);
    }
#pragma unroll
    for (int m = 0; m <= GRAM_MAXV; ++m) {
        if (m < A.nv || m == GRAM_MAXV) {
            const double sum = block_sum(acc[m], s_red);
            if (threadIdx.x == 0) A.partials[(int64_t)blockIdx.x * (A.nv + 1) + (m == GRAM_MAXV ? A.nv : m)] = sum;
        }
    }
}

// Two left factors against the same vectors: acc[l][m] = w_l . v_m - the functionals of an iterate F against the stored modes
// of its dimension under two atoms, taken as (K F) . m_j and (M F) . m_j: the modes are read once for both, where
// F . (K m_j) and F . (M m_j) read a stored product per mode and atom (pgd_vec_multidot_pair).
constexpr int PAIR_MAXV = 16;
struct MultiDot2Args {
    const double *w0, *w1;
    const double *v[PAIR_MAXV];
    int nv;
    int64_t lo, hi;
    double *partials;            // [block][2 nv]: the w0 dots, then the w1 dots
};

__global__ __launch_bounds__(TPB) void k_multidot2(MultiDot2Args A) {
    __shared__ double s_red[4];
    double a0[PAIR_MAXV], a1[PAIR_MAXV];
#pragma unroll
    for (int m = 0; m < PAIR_MAXV; ++m) a0[m] = a1[m] = 0.0;
    for (int64_t i = A.lo + (int64_t)blockIdx.x * TPB + threadIdx.x; i < A.hi; i += (int64_t)gridDim.x * TPB) {
        const double x0 = A.w0[i], x1 = A.w1[i];
#pragma unroll
        for (int m = 0; m < PAIR_MAXV; ++m)
            if (m < A.nv) { const double y = A.v[m][i]; a0[m] = fma(y, x0, a0[m]); a1[m] = fma(y, x1, a1[m]); }
    }
#pragma unroll
    for (int m = 0; m < PAIR_MAXV; ++m) {
        if (m < A.nv) {                                      // uniform
            const double s0 = block_sum(a0[m], s_red);
            const double s1 = block_sum(a1[m], s_red);
            if (threadIdx.x == 0) {
                A.partials[(int64_t)blockIdx.x * (2 * A.nv) + m] = s0;
                A.partials[(int64_t)blockIdx.x * (2 * A.nv) + A.nv + m] = s1;
            }
        }
    }
}

// All of the start's Gram data in ONE pass over the vectors once the products W_j = A v_j are stored (up to 9 vectors):
// G[i][j] = v_i . w_j for i <= j (A is symmetric) and g[j] = v_j . b - 2 k + 1 vector reads instead of the (k + 2)(k + 3) / 2 of
// a k_multidot per column.  Partial sums per workgroup: value q = j (j + 1) / 2 + i for the pair (i <= j), then the k values of g.
constexpr int GRAM9 = 9, GRAM9_NV = GRAM9 * (GRAM9 + 1) / 2 + GRAM9;
struct Gram9Args {
    const double *v[GRAM9], *w[GRAM9], *b;
    int k;
    int64_t lo, hi;
    double *partials;            // [block][k (k + 1) / 2 + k]
};

__global__ __launch_bounds__(TPB) void k_gram9(Gram9Args A) {
    __shared__ double s_red[4];
    double acc[GRAM9_NV];
#pragma unroll
    for (int q = 0; q < GRAM9_NV; ++q) acc[q] = 0.0;
    for (int64_t r = A.lo + (int64_t)blockIdx.x * TPB + threadIdx.x; r < A.hi; r += (int64_t)gridDim.x * TPB) {
        double vv[GRAM9], ww[GRAM9];
        const double br = A.b[r];
#pragma unroll
        for (int t = 0; t < GRAM9; ++t) { vv[t] = t < A.k ? A.v[t][r] : 0.0; ww[t] = t < A.k ? A.w[t][r] : 0.0; }
#pragma unroll
        for (int j = 0; j < GRAM9; ++j) {
#pragma unroll
            for (int i = 0; i <= j; ++i) acc[j * (j + 1) / 2 + i] = fma(vv[i], ww[j], acc[j * (j + 1) / 2 + i]);
            acc[GRAM9 * (GRAM9 + 1) / 2 + j] = fma(vv[j], br, acc[GRAM9 * (GRAM9 + 1) / 2 + j]);
        }
    }
    const int npair = A.k * (A.k + 1) / 2, nv = npair + A.k;
#pragma unroll
    for (int q = 0; q < GRAM9_NV; ++q) {
        const bool pair = q < GRAM9 * (GRAM9 + 1) / 2;
        const int dst = pair ? q : npair + (q - GRAM9 * (GRAM9 + 1) / 2);       // pairs of columns j < k come first in q as well
        const bool used = pair ? q < npair : (q - GRAM9 * (GRAM9 + 1) / 2) < A.k;
        if (used) {                                          // uniform
            const double sum = block_sum(acc[q], s_red);
            if (threadIdx.x == 0) A.partials[(int64_t)blockIdx.x * nv + dst] = sum;
        }
    }
}

// Two row ranges in ONE row-order launch of the diagonal form (the low and the high boundary plane of a row-sharded
// rank: two 7 us launches become one).  *done = false when the operator is not held in that form - the caller then
// launches the ranges one by one.  Partial sums: the first range's workgroups, then the second's.
int launch_spmv_dia_rows2(Ctx *c, const Mesh *m, const Csr *a, const double *x, double *y, const double *w, int64_t r0a,
                          int64_t r1a, int64_t r0b, int64_t r1b, bool dot, const int *flags, int *nparts_out, bool *done) {
    *done = false;
    if (!(c->spmv_sym && m->sym_nx > 0 && a->uvals_valid && a->uvals) || r1a <= r0a || r1b <= r0b) return PGD_OK;
    if (r0a < 0 || r1a > r0b || r1b > m->nv) return fail(c, PGD_ERR_INVALID, "spmv: bad row ranges");
    const int n1 = (int)((r1a - r0a + 63) / 64), n2 = (int)((r1b - r0b + 63) / 64);
    const int64_t plane = (int64_t)m->sym_nx * m->sym_ny;
    DiaArgs D;
    D.qq = (dot && c->spmv_qq) ? 1 : 0;
    if (dot) PGD_TRY(ensure_partials(c, std::max<int64_t>(c->partials_off + (D.qq ? 2 : 1) * (int64_t)(n1 + n2), 4 * MAX_VEC_BLOCKS)));
    D.uvals = a->uvals; D.x = x; D.w = w; D.y = y; D.partials = c->partials + c->partials_off; D.flags = flags; D.n = a->uvals_stride;
    D.nx = m->sym_nx; D.ny = m->sym_ny; D.nz = (int)(m->nv / plane);
    D.row_begin = (int)r0a; D.row_end = (int)r1a; D.nblk1 = n1; D.row_begin2 = (int)r0b; D.row_end2 = (int)r1b;
    D.z0 = D.z1 = D.zchunk = D.tiles_x = D.tiles_y = 0;
    D.unit_diag = (a->uvals_scaled && a->uvals_unit) ? 1 : 0;
    if (dot) k_spmv_dia_rows<true, true><<<n1 + n2, 64, 0, c->stream>>>(D);
    else k_spmv_dia_rows<false, true><<<n1 + n2, 64, 0, c->stream>>>(D);
    c->kcount[KC_DIA_ROWS] += 1;
    PGD_LAUNCH_CHECK(c);
    if (nparts_out) *nparts_out = n1 + n2;
    *done = true;
    return PGD_OK;
}

int launch_spmv_multi(Ctx *c, const Mesh *m, const double *vals, const double *x, const double *const *ys,
                      int ny, int64_t r0, int64_t r1, double *out_host) {
    if (r1 < 0) r1 = m->nv;
    if (r0 < 0 || r0 > r1 || r1 > m->nv) return fail(c, PGD_ERR_INVALID, "bilinear_many: bad row range");
    const int nblk = (int)((r1 - r0 + TPB - 1) / TPB);
    for (int first = 0; first < ny; first += MAXY) {
        const int cnt = (ny - first < MAXY) ? ny - first : MAXY;
        if (nblk == 0) { for (int k = 0; k < cnt; ++k) out_host[first + k] = 0.0; continue; }
        PGD_TRY(ensure_partials(c, (int64_t)nblk * MAXY > 4 * MAX_VEC_BLOCKS ? (int64_t)nblk * MAXY : 4 * MAX_VEC_BLOCKS));
        SpmvMultiArgs A;
        A.row_ptr = m->row_ptr; A.cols = m->cols; A.vals = vals; A.x = x; A.partials = c->partials;
        A.row_begin = (int)r0; A.row_end = (int)r1; A.ny = cnt;
        for (int k = 0; k < MAXY; ++k) A.ys[k] = ys[first + (k < cnt ? k : 0)];
        k_spmv_multi<<<nblk, TPB, 0, c->stream>>>(A);
        c->kcount[KC_MULTI] += 1;
        PGD_LAUNCH_CHECK(c);
        PGD_TRY(reduce_partials(c, c->partials, nblk, cnt, S_TMP, -1, 0, 0));
        PGD_HIP(c, hipMemcpyAsync(out_host + first, c->slots + S_TMP, cnt * sizeof(double), hipMemcpyDeviceToHost, c->stream));
        PGD_HIP(c, hipStreamSynchronize(c->stream));
    }
    return PGD_OK;
}

}  // namespace pgd

using namespace pgd;

extern "C" {

#if defined(PGD_STENCIL_TIMING) || defined(PGD_STENCIL_WHATIF)
int pgd_debug_read_partials(pgd_handle h, double *out, int first, int count) {
    PGD_CTX(c, h);
    PGD_HIP(c, hipMemcpyAsync(out, c->partials + first, (size_t)count * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    PGD_HIP(c, hipStreamSynchronize(c->stream));
    return PGD_OK;
}
#endif

int pgd_mg_counts(pgd_handle h, int64_t *solves, int64_t *fallbacks) {
    PGD_CTX(c, h);
    if (solves) *solves = c->mg_solves;
    if (fallbacks) *fallbacks = c->mg_fallbacks;
    return PGD_OK;
}

int pgd_classify_counts(pgd_handle h, int64_t *full, int64_t *cached) {
    PGD_CTX(c, h);
    if (full) *full = c->cls_full;
    if (cached) *cached = c->cls_fast;
    return PGD_OK;
}

int pgd_tune(pgd_handle h, int knob, int64_t value) {
    PGD_CTX(c, h);
    if (knob == PGD_TUNE_SPMV_ROWS && (value == 64 || value == 128 || value == 256)) { c->spmv_rows = (int)value; return PGD_OK; }
    if (knob == PGD_TUNE_SPMV_DICT && value >= 0 && value <= 2) { c->spmv_dict = (int)value; return PGD_OK; }
    if (knob == PGD_TUNE_SPMV_SYM && value >= 0 && value <= 1) { c->spmv_sym = (int)value; return PGD_OK; }
    if (knob == PGD_TUNE_SPMV_GRID_MIN_BYTES && value >= 0) { c->spmv_grid_min_plane_bytes = value; return PGD_OK; }
    if (knob == PGD_TUNE_SPMV_ZCHUNK_FORCE && value >= 0 && value <= 65536) { c->spmv_zchunk_force = (int)value; return PGD_OK; }
    if (knob == PGD_TUNE_PCG_FOLD_REDUCE && value >= 0 && value <= 1) { c->pcg_fold_reduce = (int)value; return PGD_OK; }
    if (knob == PGD_TUNE_PCG_SCALED && value >= 0 && value <= 1) { c->pcg_scaled = (int)value; return PGD_OK; }
    if (knob == PGD_TUNE_SPMV_ZCHUNK && value >= 0 && value <= 65536) { c->spmv_zchunk = (int)value; return PGD_OK; }
    if (knob == PGD_TUNE_PCG_SINGLE_SYNC && value >= 0 && value <= 1) { c->pcg_single_sync = (int)value; return PGD_OK; }
    if (knob == PGD_TUNE_UNIT_DIAG && value >= 0 && value <= 1) { c->spmv_unit_diag = (int)value; return PGD_OK; }
    if (knob == PGD_TUNE_PCG_DEFER_X && value >= 0 && value <= 1) { c->pcg_defer_x = (int)value; return PGD_OK; }
    if (knob == PGD_TUNE_FAULT_ITERATION && value >= -1 && value <= (1 << 30)) { c->fault_iteration = (int)value; return PGD_OK; }
    if (knob == PGD_TUNE_CLS_CACHE && value >= 0 && value <= 1) { c->cls_cache_on = (int)value; return PGD_OK; }
    if (knob == PGD_TUNE_SPMV_STENCIL && value >= 0 && value <= 1) { c->spmv_stencil = (int)value; return PGD_OK; }
    if (knob == PGD_TUNE_SPMV_ZCHUNK_STENCIL && value >= 0 && value <= 1024) { c->spmv_zchunk_stencil = (int)value; return PGD_OK; }
    if (knob == PGD_TUNE_STENCIL_DEPTH && (value == 0 || value == 3 || value == 6)) { c->stencil_depth = (int)value; return PGD_OK; }
    if (knob == PGD_TUNE_STENCIL_WG_PER_CU && value >= 1 && value <= 8) { c->stencil_wg_per_cu = (int)value; return PGD_OK; }
    if (knob == PGD_TUNE_FAULT_STAGE && value >= 0 && value <= 4) { c->fault_stage = (int)value; return PGD_OK; }
    if (knob == PGD_TUNE_FAULT_STALL_MS && value >= 0 && value <= 20000) { c->fault_stall_ms = (int)value; return PGD_OK; }
    if (knob == PGD_TUNE_COMBINE_DIA && value >= 0 && value <= 1) { c->spmv_combine_dia = (int)value; return PGD_OK; }
    if (knob == PGD_TUNE_SPMV_VARIANT && value >= 0 && value <= 2) { c->spmv_variant = (int)value; return PGD_OK; }
    if (knob == PGD_TUNE_SPMV_ROW_CLASSES && value >= 0 && value <= 1) { c->spmv_classes = (int)value; return PGD_OK; }
    if (knob == PGD_TUNE_ASM_LATTICE && value >= 0 && value <= 3) { c->asm_lattice = (int)value; return PGD_OK; }
    if (knob == PGD_TUNE_PCG_LAG_X && value >= 0 && value <= 1) { c->pcg_lag_x = (int)value; return PGD_OK; }
    if (knob == PGD_TUNE_PCG_STREAM_HINTS && value >= 0 && value <= 1) { c->pcg_stream_hints = (int)value; return PGD_OK; }
    if (knob == PGD_TUNE_PCG_SMALL_SINGLE_SYNC && value >= 0 && value <= 1) { c->pcg_small_ss = (int)value; return PGD_OK; }
    if (knob == PGD_TUNE_PCG_SMALL_ROWS && value >= 0) { c->pcg_small_ss_rows = value; return PGD_OK; }
    if (knob == PGD_TUNE_SPMV_FETCH_DEPTH && (value == 3 || value == 6)) { c->spmv_fetch_depth = (int)value; return PGD_OK; }
    if (knob == PGD_TUNE_SPMV_ZCHUNK_CODED2 && value >= 0 && value <= 1020) { c->spmv_zchunk_coded2 = (int)value / 6 * 6; return PGD_OK; }
    if (knob == PGD_TUNE_PCG_PIPELINE && value >= 0 && value <= 1) { c->pcg_pipeline = (int)value; return PGD_OK; }
    if (knob == PGD_TUNE_PCG_EXACT_PHASE && value >= 0 && value <= 1) { c->pcg_exact_phase = (int)value; return PGD_OK; }
    if (knob == PGD_TUNE_PCG_FOLD_FINISH && value >= 0 && value <= 1) { c->pcg_fold_finish = (int)value; return PGD_OK; }
    if (knob == PGD_TUNE_ATOM_FAST && value >= 0 && value <= 1) { c->atom_fast = (int)value; return PGD_OK; }
    if (knob == PGD_TUNE_LAZY_CSR && value >= 0 && value <= 1) { c->lazy_csr = (int)value; return PGD_OK; }
    if (knob == PGD_TUNE_SPMV_ZCHUNK_CODED && value >= 3 && value <= 1024) { c->spmv_zchunk_coded = (int)value; return PGD_OK; }
    if (knob == PGD_TUNE_PCG_PRECOND && value >= 0 && value <= 1) { c->pcg_precond = (int)value; return PGD_OK; }
    if (knob == PGD_TUNE_PCG_DERIVE_SCALED && value >= 0 && value <= 1) { c->pcg_derive_scaled = (int)value; return PGD_OK; }
    if (knob == PGD_TUNE_COMM_SELF_PERIODIC && value >= 0 && value <= 1) { c->comm.self_periodic = value != 0; return PGD_OK; }
    if (knob == PGD_TUNE_HALO_OVERLAP_MIN_ROWS && value >= 0) { c->comm.overlap_min_rows = value; return PGD_OK; }
    if (knob == PGD_TUNE_SHARD_ONE_MARCH && value >= 0 && value <= 1) { c->shard_one_march = (int)value; return PGD_OK; }
    if (knob == PGD_TUNE_DIA_MARCH3 && value >= 0 && value <= 1) { c->dia_march3 = (int)value; return PGD_OK; }
    if (knob == PGD_TUNE_STENCIL_ROWS && (value == 0 || value == 2 || value == 4)) { c->stencil_rows = (int)value; return PGD_OK; }
    if (knob == PGD_TUNE_PUSH_IN_UPDATE && (value == 0 || value == 1)) { c->push_in_update = (int)value; return PGD_OK; }
    if (knob == PGD_TUNE_MG_CHUNK && value >= 2 && value <= 16 && value % 2 == 0) { c->mg_chunk = (int)value; return PGD_OK; }
    if (knob == PGD_TUNE_MG_MARCH_MIN && value >= 0 && value <= 1 << 20) { c->mg_march_min = (int)value; return PGD_OK; }
    return fail(c, PGD_ERR_INVALID, "tune: unknown knob %d or value out of range", knob);
}

int pgd_spmv(pgd_handle h, pgd_handle ah, pgd_handle xh, pgd_handle yh, int64_t r0, int64_t r1) {
    PGD_CTX(c, h);
    Csr *a = get_csr(c, ah);
    Mesh *m = a ? get_mesh(c, a->mesh) : nullptr;
    Vec *x = get_vec(c, xh), *y = get_vec(c, yh);
    if (!a || !m || !x || !y || x->n != m->nv || y->n != m->nv || x == y)
        return fail(c, PGD_ERR_INVALID, "spmv: invalid handles, size mismatch or x aliases y");
    if (atom_fast_form(c, m, a, r0, r1)) return launch_spmv_op(c, m, a, x->d, y->d, nullptr, r0, r1, false, true, nullptr, nullptr);
    // an OPERATOR that holds its (unscaled) diagonal form - pgd_op_combine made it, nothing has formed its CSR values yet: the
    // product from that form (bit-identical y) instead of 8 nnz (T + 1) bytes of forming the CSR values first (r04: the residual
    // of the spectral start, the host-driven sharded loops)
    if (c->spmv_sym && c->atom_fast && m->sym_nx > 0 && !a->immutable && a->uvals && a->uvals_valid && !a->uvals_scaled && !a->st_virtual) {
        const int64_t plane = (int64_t)m->sym_nx * m->sym_ny;
        const int64_t q0 = r0 < 0 ? 0 : r0, q1 = r1 < 0 ? m->nv : r1;
        if (q0 >= 0 && q1 <= m->nv && q0 <= q1 && q0 % plane == 0 && q1 % plane == 0)
            return launch_spmv_op(c, m, a, x->d, y->d, nullptr, q0, q1, false, true, nullptr, nullptr);
    }
    PGD_TRY(ensure_vals(c, m, a));
    return launch_spmv(c, m, a->vals, x->d, y->d, nullptr, r0, r1, false, true, nullptr, nullptr);
}

int pgd_bilinear(pgd_handle h, pgd_handle ah, pgd_handle xh, pgd_handle yh, int64_t r0, int64_t r1, double *out) {
    PGD_CTX(c, h);
    Csr *a = get_csr(c, ah);
    Mesh *m = a ? get_mesh(c, a->mesh) : nullptr;
    Vec *x = get_vec(c, xh), *y = get_vec(c, yh);
    if (!a || !m || !x || !y || !out || x->n != m->nv || y->n != m->nv)
        return fail(c, PGD_ERR_INVALID, "bilinear: invalid handles or size mismatch");
    int nparts = 0;
    PGD_TRY(ensure_vals(c, m, a));
    PGD_TRY(launch_spmv(c, m, a->vals, y->d, nullptr, x->d, r0, r1, true, false, nullptr, &nparts));
    if (nparts == 0) { *out = 0.0; return PGD_OK; }
    PGD_TRY(reduce_partials(c, c->partials, nparts, 1, S_TMP, -1, 0, 0));
    PGD_HIP(c, hipMemcpyAsync(out, c->slots + S_TMP, sizeof(double), hipMemcpyDeviceToHost, c->stream));
    PGD_HIP(c, hipStreamSynchronize(c->stream));
    return PGD_OK;
}

int pgd_bilinear_many(pgd_handle h, pgd_handle ah, pgd_handle xh, const pgd_handle *yhs, int ny, int64_t r0,
                      int64_t r1, double *out) {
    PGD_CTX(c, h);
    Csr *a = get_csr(c, ah);
    Mesh *m = a ? get_mesh(c, a->mesh) : nullptr;
    Vec *x = get_vec(c, xh);
    if (!a || !m || !x || !out || ny < 0 || (ny > 0 && !yhs) || x->n != m->nv)
        return fail(c, PGD_ERR_INVALID, "bilinear_many: invalid handles or size mismatch");
    std::vector<const double *> ys((size_t)ny);
    for (int k = 0; k < ny; ++k) {
        Vec *y = get_vec(c, yhs[k]);
        if (!y || y->n != m->nv) return fail(c, PGD_ERR_INVALID, "bilinear_many: invalid vector %d", k);
        ys[k] = y->d;
    }
    if (ny == 0) return PGD_OK;
    PGD_TRY(ensure_vals(c, m, a));
    return launch_spmv_multi(c, m, a->vals, x->d, ys.data(), ny, r0, r1, out);
}

int pgd_op_symmetrize(pgd_handle h, pgd_handle ah, int *used) {
    PGD_CTX(c, h);
    Csr *a = get_csr(c, ah);
    Mesh *m = a ? get_mesh(c, a->mesh) : nullptr;
    if (!a || !m) return fail(c, PGD_ERR_INVALID, "op_symmetrize: invalid handle");
    bool usable = false;
    PGD_TRY(ensure_sym(c, m, a, &usable));
    if (used) *used = usable ? 1 : 0;
    return PGD_OK;
}

int pgd_atom_product_form(pgd_handle h, pgd_handle ah, int *form) {
    PGD_CTX(c, h);
    Csr *a = get_csr(c, ah);
    Mesh *m = a ? get_mesh(c, a->mesh) : nullptr;
    if (!a || !m || !form) return fail(c, PGD_ERR_INVALID, "atom_product_form: invalid handle");
    *form = 0;
    if (atom_fast_form(c, m, a, 0, m->nv)) *form = a->cls_count > 0 ? 2 : 1;
    return PGD_OK;
}

int pgd_op_classify(pgd_handle h, pgd_handle ah, int *classes) {
    PGD_CTX(c, h);
    Csr *a = get_csr(c, ah);
    Mesh *m = a ? get_mesh(c, a->mesh) : nullptr;
    if (!a || !m) return fail(c, PGD_ERR_INVALID, "op_classify: invalid handle");
    PGD_TRY(dia_classify(c, m, a));
    if (classes) *classes = a->cls_count;
    return PGD_OK;
}

int pgd_spmv_dot_slot(pgd_handle h, pgd_handle ah, pgd_handle xh, pgd_handle yh, pgd_handle wh, int64_t r0,
                      int64_t r1, int slot) {
    PGD_CTX(c, h);
    Csr *a = get_csr(c, ah);
    Mesh *m = a ? get_mesh(c, a->mesh) : nullptr;
    Vec *x = get_vec(c, xh), *y = get_vec(c, yh), *w = get_vec(c, wh);
    if (!a || !m || !x || !y || !w || x->n != m->nv || y->n != m->nv || w->n != m->nv || x == y ||
        slot < 0 || slot >= PGD_NSLOTS)
        return fail(c, PGD_ERR_INVALID, "spmv_dot_slot: invalid handles, sizes or slot");
    int nparts = 0;
    PGD_TRY(launch_spmv_op(c, m, a, x->d, y->d, w->d, r0, r1, true, true, c->flags, &nparts));
    if (nparts == 0) {   // empty row range: its partial is 0 (callers all-reduce the slot unconditionally)
        PGD_HIP(c, hipMemsetAsync(c->slots + slot, 0, sizeof(double), c->stream));
        return PGD_OK;
    }
    return reduce_partials(c, c->partials, nparts, 1, slot, 0, 0, 0);
}

// x . y_j for up to 256 vectors y_j at once: ceil(k / 17) passes over x, one host synchronisation.  The driver's
// functionals of one iterate against all stored modes of its dimension (fem._bilinear_scalar) come through here.
int pgd_vec_multidot(pgd_handle h, pgd_handle xh, const pgd_handle *yhs, int k, int64_t lo, int64_t hi, double *out) {
    PGD_CTX(c, h);
    Vec *x = get_vec(c, xh);
    if (!x || !yhs || !out || k < 1 || k > 256) return fail(c, PGD_ERR_INVALID, "vec_multidot: invalid handles or count (1..256)");
    if (hi < 0) hi = x->n;
    if (lo < 0 || lo > hi || hi > x->n) return fail(c, PGD_ERR_INVALID, "vec_multidot: bad range");
    std::vector<const double *> y((size_t)k);
    for (int j = 0; j < k; ++j) {
        Vec *v = get_vec(c, yhs[j]);
        if (!v || v->n != x->n) return fail(c, PGD_ERR_INVALID, "vec_multidot: invalid vector %d", j);
        y[(size_t)j] = v->d;
    }
    for (int j = 0; j < k; ++j) out[j] = 0.0;
    if (hi == lo) return PGD_OK;
    PGD_TRY(ensure_work(c, 6, 2 * (int64_t)MAX_VEC_BLOCKS > 512 ? 2 * (int64_t)MAX_VEC_BLOCKS : 512));
    double *res = c->work[6];
    const int g = grid_for(hi - lo);
    PGD_TRY(ensure_partials(c, (int64_t)g * (GRAM_MAXV + 1) > 4 * MAX_VEC_BLOCKS ? (int64_t)g * (GRAM_MAXV + 1) : 4 * MAX_VEC_BLOCKS));
    int off = 0;
    for (int j0 = 0; j0 < k; j0 += GRAM_MAXV) {
        const int nv = std::min(GRAM_MAXV, k - j0);
        MultiDotArgs A;
        A.w = x->d; A.b = x->d; A.nv = nv; A.lo = lo; A.hi = hi; A.partials = c->partials;
        for (int q = 0; q < GRAM_MAXV; ++q) A.v[q] = y[(size_t)(j0 + std::min(q, nv - 1))];
        k_multidot<<<g, TPB, 0, c->stream>>>(A);
        PGD_LAUNCH_CHECK(c);
        PGD_TRY(reduce_partials_to(c, c->partials, g, nv + 1, res + off));     // nv dots, then y_last . x again (unused)
        off += nv + 1;
    }
    std::vector<double> host((size_t)off);
    PGD_HIP(c, hipMemcpyAsync(host.data(), res, (size_t)off * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    PGD_HIP(c, hipStreamSynchronize(c->stream));
    off = 0;
    for (int j0 = 0; j0 < k; j0 += GRAM_MAXV) {
        const int nv = std::min(GRAM_MAXV, k - j0);
        for (int q = 0; q < nv; ++q) out[j0 + q] = host[(size_t)(off + q)];
        off += nv + 1;
    }
    return PGD_OK;
}

int pgd_vec_multidot_pair(pgd_handle h, pgd_handle x0h, pgd_handle x1h, const pgd_handle *yhs, int k, int64_t lo, int64_t hi, double *out) {
    PGD_CTX(c, h);
    Vec *x0 = get_vec(c, x0h), *x1 = get_vec(c, x1h);
    if (!x0 || !x1 || x0->n != x1->n || !yhs || !out || k < 1 || k > 128)
        return fail(c, PGD_ERR_INVALID, "vec_multidot_pair: invalid handles or count (1..128)");
    if (hi < 0) hi = x0->n;
    if (lo < 0 || lo > hi || hi > x0->n) return fail(c, PGD_ERR_INVALID, "vec_multidot_pair: bad range");
    std::vector<const double *> y((size_t)k);
    for (int j = 0; j < k; ++j) {
        Vec *v = get_vec(c, yhs[j]);
        if (!v || v->n != x0->n) return fail(c, PGD_ERR_INVALID, "vec_multidot_pair: invalid vector %d", j);
        y[(size_t)j] = v->d;
    }
    for (int j = 0; j < 2 * k; ++j) out[j] = 0.0;
    if (hi == lo) return PGD_OK;
    PGD_TRY(ensure_work(c, 6, 2 * (int64_t)MAX_VEC_BLOCKS > 512 ? 2 * (int64_t)MAX_VEC_BLOCKS : 512));
    double *res = c->work[6];
    const int g = grid_for(hi - lo);
    PGD_TRY(ensure_partials(c, std::max<int64_t>((int64_t)g * 2 * PAIR_MAXV, 4 * MAX_VEC_BLOCKS)));
    int off = 0;
    for (int j0 = 0; j0 < k; j0 += PAIR_MAXV) {
        const int nv = std::min(PAIR_MAXV, k - j0);
        MultiDot2Args A;
        A.w0 = x0->d; A.w1 = x1->d; A.nv = nv; A.lo = lo; A.hi = hi; A.partials = c->partials;
        for (int q = 0; q < PAIR_MAXV; ++q) A.v[q] = y[(size_t)(j0 + std::min(q, nv - 1))];
        k_multidot2<<<g, TPB, 0, c->stream>>>(A);
        PGD_LAUNCH_CHECK(c);
        PGD_TRY(reduce_partials_to(c, c->partials, g, 2 * nv, res + off));
        off += 2 * nv;
    }
    std::vector<double> host((size_t)off);
    PGD_HIP(c, hipMemcpyAsync(host.data(), res, (size_t)off * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    PGD_HIP(c, hipStreamSynchronize(c->stream));
    off = 0;
    for (int j0 = 0; j0 < k; j0 += PAIR_MAXV) {
        const int nv = std::min(PAIR_MAXV, k - j0);
        for (int q = 0; q < nv; ++q) { out[j0 + q] = host[(size_t)(off + q)]; out[k + j0 + q] = host[(size_t)(off + nv + q)]; }
        off += 2 * nv;
    }
    return PGD_OK;
}

int pgd_start_gram(pgd_handle h, pgd_handle ah, const pgd_handle *vhs, int k, pgd_handle bh, int64_t r0, int64_t r1,
                   double *out) {
    PGD_CTX(c, h);
    Csr *a = get_csr(c, ah);
    Mesh *m = a ? get_mesh(c, a->mesh) : nullptr;
    Vec *b = get_vec(c, bh);
    if (!a || !m || !b || !vhs || !out || k < 1 || k > GRAM_MAXV || b->n != m->nv)
        return fail(c, PGD_ERR_INVALID, "start_gram: invalid handles or more than %d vectors", GRAM_MAXV);
    if (r1 < 0) r1 = m->nv;
    if (r0 < 0 || r0 > r1 || r1 > m->nv) return fail(c, PGD_ERR_INVALID, "start_gram: bad row range");
    const double *v[GRAM_MAXV];
    for (int j = 0; j < k; ++j) {
        Vec *x = get_vec(c, vhs[j]);
        if (!x || x->n != m->nv) return fail(c, PGD_ERR_INVALID, "start_gram: invalid vector %d", j);
        v[j] = x->d;
    }
    for (int q = 0; q < k * k + k; ++q) out[q] = 0.0;
    c->gram_op = 0; c->gram_k = 0; c->gram_n = 0;       // (the products of an earlier call are about to be overwritten)
    if (r1 == r0) return PGD_OK;
    bool sym = false;
    PGD_TRY(ensure_sym(c, m, a, &sym));                 // the SPD solve that follows reads the same copy
    PGD_TRY(ensure_work(c, 3, m->nv));                  // w: the PCG's q buffer (no solve is running)
    PGD_TRY(ensure_work(c, 6, 2 * (int64_t)MAX_VEC_BLOCKS > 256 ? 2 * (int64_t)MAX_VEC_BLOCKS : 256));
    double *w = c->work[3], *res = c->work[6];          // res: k columns of (j + 2) values, packed
    // uniform grids: the k products below read a code byte per row or take the stencil form (the classification is a copy + one
    // verifying pass once the mesh has seen the structure: worth it from two products on)
    const bool dbg_t = getenv("PGD_DEBUG_PCG") != nullptr;
    auto dbg_now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double dbg_t0 = dbg_now();
    if (k >= 2) PGD_TRY(dia_classify(c, m, a));
    if (dbg_t) fprintf(stderr, "[start_gram] k %d classify %.2f ms (cls %d stencil %d)\n", k, 1e3 * (dbg_now() - dbg_t0), a->cls_count, (int)a->st_ok);
    if (k <= GRAM9) {
        // the products first, each into its own vector; then all dots in one pass
        const size_t need = (size_t)k * (size_t)m->nv * sizeof(double);
        if (c->gram_w_bytes < need) {
            if (c->gram_w) dev_release(c, c->gram_w, c->gram_w_bytes);
            c->gram_w = nullptr; c->gram_w_bytes = 0;
            void *pw;
            PGD_TRY(dev_alloc(c, &pw, need));
            c->gram_w = (double *)pw; c->gram_w_bytes = need;
        }
        Gram9Args A;
        for (int j = 0; j < k; ++j) {
            double *wj = c->gram_w + (size_t)j * m->nv;
            PGD_TRY(launch_spmv_op(c, m, a, v[j], wj, nullptr, r0, r1, false, true, nullptr, nullptr));
        }
        for (int t = 0; t < GRAM9; ++t) { A.v[t] = v[t < k ? t : 0]; A.w[t] = c->gram_w + (size_t)(t < k ? t : 0) * m->nv; }
        A.b = b->d; A.k = k; A.lo = r0; A.hi = r1;
        const int nv = k * (k + 1) / 2 + k;
        const int gg = grid_for(r1 - r0, TPB, 1024);
        PGD_TRY(ensure_partials(c, std::max<int64_t>((int64_t)gg * nv, 4 * MAX_VEC_BLOCKS)));
        A.partials = c->partials;
        k_gram9<<<gg, TPB, 0, c->stream>>>(A);
        PGD_LAUNCH_CHECK(c);
        PGD_TRY(reduce_partials_to(c, c->partials, gg, nv, res));
        std::vector<double> host((size_t)nv);
        PGD_HIP(c, hipMemcpyAsync(host.data(), res, (size_t)nv * sizeof(double), hipMemcpyDeviceToHost, c->stream));
        PGD_HIP(c, hipStreamSynchronize(c->stream));       // the one host synchronisation of the call
        for (int j = 0; j < k; ++j) {
            for (int i = 0; i <= j; ++i) out[i * k + j] = out[j * k + i] = host[(size_t)(j * (j + 1) / 2 + i)];
            out[k * k + j] = host[(size_t)(k * (k + 1) / 2 + j)];
        }
        if (r0 == 0 && r1 == m->nv) { c->gram_op = ah; c->gram_k = k; c->gram_n = m->nv; }     // whole products, kept for pgd_start_residual
        return PGD_OK;
    }
    const int g = grid_for(r1 - r0);
    int off = 0;
    for (int j = 0; j < k; ++j) {
        PGD_TRY(launch_spmv_op(c, m, a, v[j], w, nullptr, r0, r1, false, true, nullptr, nullptr));
        PGD_TRY(ensure_partials(c, (int64_t)g * (GRAM_MAXV + 1) > 4 * MAX_VEC_BLOCKS ? (int64_t)g * (GRAM_MAXV + 1) : 4 * MAX_VEC_BLOCKS));
        MultiDotArgs A;
        A.w = w; A.b = b->d; A.nv = j + 1; A.lo = r0; A.hi = r1; A.partials = c->partials;
        for (int q = 0; q < GRAM_MAXV; ++q) A.v[q] = v[q <= j ? q : j];
        k_multidot<<<g, TPB, 0, c->stream>>>(A);
        PGD_LAUNCH_CHECK(c);
        PGD_TRY(reduce_partials_to(c, c->partials, g, j + 2, res + off));
        off += j + 2;
    }
    std::vector<double> host((size_t)off);
    PGD_HIP(c, hipMemcpyAsync(host.data(), res, (size_t)off * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    PGD_HIP(c, hipStreamSynchronize(c->stream));       // the one host synchronisation of the call
    off = 0;
    for (int j = 0; j < k; ++j) {
        for (int i = 0; i <= j; ++i) out[i * k + j] = out[j * k + i] = host[(size_t)off + i];
        out[k * k + j] = host[(size_t)off + j + 1];
        off += j + 2;
    }
    return PGD_OK;
}

}  // extern "C"
