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
#include "pgd_internal.h"

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

int launch_spmv(Ctx *c, const Mesh *m, const double *vals, const double *x, double *y, const double *w,
                int64_t r0, int64_t r1, bool dot, bool store, const int *flags, int *nparts_out) {
    if (r1 < 0) r1 = m->nv;
    if (r0 < 0 || r0 > r1 || r1 > m->nv) return fail(c, PGD_ERR_INVALID, "spmv: bad row range");
    const int64_t nrows = r1 - r0;
    const int R = c->spmv_rows;
    const int nblk = (int)((nrows + R - 1) / R);
    if (nparts_out) *nparts_out = nblk;
    if (nblk == 0) return PGD_OK;
    if (dot) PGD_TRY(ensure_partials(c, (int64_t)nblk > 4 * MAX_VEC_BLOCKS ? nblk : 4 * MAX_VEC_BLOCKS));
    SpmvArgs A;
    A.row_ptr = m->row_ptr; A.cols = m->cols; A.vals = vals; A.x = x; A.w = w; A.y = y;
    A.partials = c->partials; A.flags = flags; A.row_begin = (int)r0; A.row_end = (int)r1;
    // events perturb the stream (~5 us each side): time one launch in four
    const bool candidate = c->prof && (!c->prof_pcg_only || (dot && store));
    const bool timed = candidate && ((c->prof_seen++ & 3) == 0);
    if (timed) {
        if (c->ev_used + 2 > c->ev.size()) prof_flush(c);
        PGD_HIP(c, hipEventRecord(c->ev[c->ev_used], c->stream));
    }
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
    if (timed) {
        PGD_HIP(c, hipEventRecord(c->ev[c->ev_used + 1], c->stream));
        c->ev_used += 2;
        c->prof_launches += 1;
        // algorithmic bytes of the rows this launch covers: 12 B per entry + 20 B per row
        const double frac = m->nv > 0 ? (double)nrows / (double)m->nv : 0.0;
        c->prof_bytes += 12.0 * (double)m->nnz * frac + 20.0 * (double)nrows;
    }
    PGD_LAUNCH_CHECK(c);
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

int pgd_tune(pgd_handle h, int knob, int64_t value) {
    PGD_CTX(c, h);
    if (knob == PGD_TUNE_SPMV_ROWS && (value == 64 || value == 128 || value == 256)) { c->spmv_rows = (int)value; return PGD_OK; }
    if (knob == PGD_TUNE_SPMV_DICT && value >= 0 && value <= 2) { c->spmv_dict = (int)value; return PGD_OK; }
    return fail(c, PGD_ERR_INVALID, "tune: unknown knob %d or value out of range", knob);
}

int pgd_spmv(pgd_handle h, pgd_handle ah, pgd_handle xh, pgd_handle yh, int64_t r0, int64_t r1) {
    PGD_CTX(c, h);
    Csr *a = get_csr(c, ah);
    Mesh *m = a ? get_mesh(c, a->mesh) : nullptr;
    Vec *x = get_vec(c, xh), *y = get_vec(c, yh);
    if (!a || !m || !x || !y || x->n != m->nv || y->n != m->nv || x == y)
        return fail(c, PGD_ERR_INVALID, "spmv: invalid handles, size mismatch or x aliases y");
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
    return launch_spmv_multi(c, m, a->vals, x->d, ys.data(), ny, r0, r1, out);
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
    PGD_TRY(launch_spmv(c, m, a->vals, x->d, y->d, w->d, r0, r1, true, true, c->flags, &nparts));
    if (nparts == 0) {   // empty row range: its partial is 0 (callers all-reduce the slot unconditionally)
        PGD_HIP(c, hipMemsetAsync(c->slots + slot, 0, sizeof(double), c->stream));
        return PGD_OK;
    }
    return reduce_partials(c, c->partials, nparts, 1, slot, 0, 0, 0);
}

}  // extern "C"
