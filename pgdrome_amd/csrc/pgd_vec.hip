// Vector kernels: fill / scale / axpy / scatter-set, the deterministic two-stage
// dot (wavefront shuffle -> LDS -> fixed-order final pass), and an int32 scan.
// All are HBM-bound streams: 8 B/lane loads, grid capped at 8 workgroups per CU
// with a grid-stride loop (guide: Guideline 11).
#include "pgd_internal.h"

#include <cstring>

namespace pgd {

__global__ __launch_bounds__(TPB) void k_fill(double *__restrict__ v, double a, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x; i < n; i += (int64_t)gridDim.x * TPB) v[i] = a;
}

__global__ __launch_bounds__(TPB) void k_scale(double *__restrict__ v, double a, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x; i < n; i += (int64_t)gridDim.x * TPB) v[i] *= a;
}

__global__ __launch_bounds__(TPB) void k_axpy(double *__restrict__ y, double a, const double *__restrict__ x, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x; i < n; i += (int64_t)gridDim.x * TPB)
        y[i] = fma(a, x[i], y[i]);
}

__global__ __launch_bounds__(TPB) void k_set(double *__restrict__ v, const int *__restrict__ idx,
                                             const double *__restrict__ val, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x; i < n; i += (int64_t)gridDim.x * TPB) v[idx[i]] = val[i];
}

// y = sum_k c_k x_k for up to 8 vectors per pass (one write per pass instead of one per term):
// the online reconstruction u(x) = sum_k [prod_i F_i^k(mu_i)] F_x^k of a PGD solution
// (reference model.py:805-842) is this tall-skinny product; 8 (K + 1) n bytes per call.
struct LincombArgs {
    const double *x[8];
    double c[8];
    int k;
    int accumulate;   // 1: y += ..., 0: y = ...
};

__global__ __launch_bounds__(TPB) void k_lincomb(double *__restrict__ y, LincombArgs A, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x; i < n; i += (int64_t)gridDim.x * TPB) {
        double s = A.accumulate ? y[i] : 0.0;
#pragma unroll
        for (int t = 0; t < 8; ++t)
            if (t < A.k) s = fma(A.c[t], A.x[t][i], s);
        y[i] = s;
    }
}

// partial[b] = sum over the block's grid-stride share of x_i y_i (fixed order)
__global__ __launch_bounds__(TPB) void k_dot(const double *__restrict__ x, const double *__restrict__ y,
                                             int64_t lo, int64_t hi, double *__restrict__ partials) {
    __shared__ double s_red[4];
    double acc = 0.0;
    for (int64_t i = lo + (int64_t)blockIdx.x * TPB + threadIdx.x; i < hi; i += (int64_t)gridDim.x * TPB)
        acc = fma(x[i], y[i], acc);
    acc = block_sum(acc, s_red);
    if (threadIdx.x == 0) partials[blockIdx.x] = acc;
}

// Final pass of every reduction: one 1024-thread workgroup adds `nparts` partial
// sums per value in a fixed order and writes slots[slot0 + v].  check_mode 1 adds
// the PCG convergence test (library-driven loop): iters += 1, done <- rr <= tol2.
__global__ __launch_bounds__(1024) void k_reduce_partials(const double *__restrict__ partials, int nparts,
                                                          int nvals, double *__restrict__ slots, int slot0,
                                                          int check_mode, int slot_rr, int slot_tol2,
                                                          int *__restrict__ flags) {
    __shared__ double s_w[16];
    if (flags && check_mode >= 0 && flags[0]) return;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int v = 0; v < nvals; ++v) {
        // 8 independent accumulators: 8 loads in flight per lane instead of a dependent chain
        // (65536 SpMV partials took 30 us as a chain); the order is fixed, so still reproducible
        double a8[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        int i = threadIdx.x;
        for (; i + 7 * 1024 < nparts; i += 8 * 1024) {
#pragma unroll
            for (int u = 0; u < 8; ++u) a8[u] += partials[(int64_t)(i + u * 1024) * nvals + v];
        }
        for (int u = 0; i < nparts; i += 1024, ++u) a8[u & 7] += partials[(int64_t)i * nvals + v];
        double acc = ((a8[0] + a8[1]) + (a8[2] + a8[3])) + ((a8[4] + a8[5]) + (a8[6] + a8[7]));
        acc = wave_sum(acc);
        __syncthreads();
        if (lane == 0) s_w[wv] = acc;
        __syncthreads();
        if (threadIdx.x == 0) {
            double t = 0.0;
#pragma unroll
            for (int k = 0; k < 16; ++k) t += s_w[k];
            slots[slot0 + v] = t;
        }
    }
    if (check_mode == 1 && threadIdx.x == 0) {
        __threadfence_block();
        const double rr = slots[slot_rr], tol2 = slots[slot_tol2];
        flags[1] += 1;
        if (!(rr == rr)) { flags[0] = 1; flags[2] = PGD_ERR_SINGULAR; }   // NaN: breakdown
        else if (rr <= tol2) flags[0] = 1;
    }
    // check_mode 2: the diagonally scaled recurrence (pgd_pcg.hip).  slots[slot0] = r~.r~; slots[slot_rr] is the TRUE
    // r.r only once flags[3] (exact phase) is set - until then r.r >= d_min r~.r~ proves that the test cannot pass yet,
    // and the phase is entered two orders of magnitude (in the norm) before it could.
    if (check_mode == 2 && threadIdx.x == 0) {
        __threadfence_block();
        const double rz = slots[slot0], rr = slots[slot_rr], tol2 = slots[slot_tol2];
        flags[1] += 1;
        if (!(rz == rz) || !(rr == rr)) { flags[0] = 1; flags[2] = PGD_ERR_SINGULAR; }
        else if (flags[3]) { if (rr <= tol2) flags[0] = 1; }
        else if (rz * slots[S_DMIN] <= 1e4 * tol2) flags[3] = 1;
    }
}

// ---- int32 exclusive scan (setup only): 1024 items per workgroup, recursive on block sums
constexpr int SCAN_ITEMS = 4;   // per thread

__global__ __launch_bounds__(TPB) void k_scan_block(const int *__restrict__ in, int *__restrict__ out,
                                                    int *__restrict__ block_sums, int64_t n) {
    __shared__ int s_wave[4];
    const int64_t base = ((int64_t)blockIdx.x * TPB + threadIdx.x) * SCAN_ITEMS;
    int v[SCAN_ITEMS], run = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) {
        v[k] = (base + k < n) ? in[base + k] : 0;
        run += v[k];
    }
    // inclusive scan of `run` across the wave, then across the 4 waves
    int inc = run;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        int t = __shfl_up(inc, off, 64);
        if (lane >= off) inc += t;
    }
    if (lane == 63) s_wave[wv] = inc;
    __syncthreads();
    int wave_off = 0;
    for (int w = 0; w < wv; ++w) wave_off += s_wave[w];
    int excl = wave_off + inc - run;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) {
        if (base + k < n) out[base + k] = excl;
        excl += v[k];
    }
    if (threadIdx.x == TPB - 1 && block_sums) block_sums[blockIdx.x] = wave_off + inc;
}

__global__ __launch_bounds__(TPB) void k_scan_add(int *__restrict__ out, const int *__restrict__ block_off, int64_t n) {
    const int64_t base = ((int64_t)blockIdx.x * TPB + threadIdx.x) * SCAN_ITEMS;
    const int off = block_off[blockIdx.x];
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k)
        if (base + k < n) out[base + k] += off;
}

__global__ void k_scan_total(const int *__restrict__ in, int *__restrict__ out, int64_t n) {
    // out[n] = out[n-1] + in[n-1]  (total), single thread
    if (threadIdx.x == 0 && blockIdx.x == 0) out[n] = (n > 0) ? out[n - 1] + in[n - 1] : 0;
}

static int scan_rec(Ctx *c, const int *in, int *out, int64_t n) {
    const int64_t per_block = (int64_t)TPB * SCAN_ITEMS;
    const int64_t nb = (n + per_block - 1) / per_block;
    if (nb <= 1) {
        k_scan_block<<<1, TPB, 0, c->stream>>>(in, out, nullptr, n);
        PGD_LAUNCH_CHECK(c);
        return PGD_OK;
    }
    int *sums = nullptr, *offs = nullptr;
    void *p;
    PGD_TRY(dev_alloc(c, &p, (size_t)nb * sizeof(int)));
    sums = (int *)p;
    if (dev_alloc(c, &p, (size_t)(nb + 1) * sizeof(int)) != PGD_OK) { (void)hipFree(sums); return PGD_ERR_NOMEM; }
    offs = (int *)p;
    k_scan_block<<<(int)nb, TPB, 0, c->stream>>>(in, out, sums, n);
    int rc = scan_rec(c, sums, offs, nb);
    if (rc == PGD_OK) k_scan_add<<<(int)nb, TPB, 0, c->stream>>>(out, offs, n);
    hipError_t e = hipStreamSynchronize(c->stream);
    (void)hipFree(sums);
    (void)hipFree(offs);
    if (rc != PGD_OK) return rc;
    if (e != hipSuccess) return fail(c, PGD_ERR_HIP, "scan: %s", hipGetErrorString(e));
    return PGD_OK;
}

int scan_exclusive_i32(Ctx *c, const int *in, int *out, int64_t n) {
    if (n > 0) PGD_TRY(scan_rec(c, in, out, n));
    k_scan_total<<<1, 64, 0, c->stream>>>(in, out, n);
    PGD_LAUNCH_CHECK(c);
    return PGD_OK;
}

// First stage for long partial lists (a 64-row-per-workgroup SpMV leaves 262 144 of them):
// workgroup b adds partials [1024 b, 1024 b + 1024) of every value in a fixed order.
__global__ __launch_bounds__(TPB) void k_reduce_stage1(const double *__restrict__ partials, int nparts, int nvals,
                                                       double *__restrict__ out, const int *__restrict__ flags,
                                                       int check_mode) {
    if (flags && check_mode >= 0 && flags[0]) return;
    __shared__ double s_red[4];
    const int base = blockIdx.x * 1024;
    for (int v = 0; v < nvals; ++v) {
        double a[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = base + threadIdx.x + u * TPB;
            a[u] = (i < nparts) ? partials[(int64_t)i * nvals + v] : 0.0;
        }
        const double t = block_sum((a[0] + a[1]) + (a[2] + a[3]), s_red);
        if (threadIdx.x == 0) out[(int64_t)blockIdx.x * nvals + v] = t;
    }
}

int reduce_partials(Ctx *c, const double *partials, int nparts, int nvals, int slot0, int check_mode,
                    int slot_rr, int slot_tol2) {
    if (nparts > 8192) {
        const int nb = (nparts + 1023) / 1024;
        PGD_TRY(ensure_work(c, 5, (int64_t)nb * nvals > 4096 ? (int64_t)nb * nvals : 4096));
        k_reduce_stage1<<<nb, TPB, 0, c->stream>>>(partials, nparts, nvals, c->work[5], c->flags, check_mode);
        partials = c->work[5];
        nparts = nb;
    }
    k_reduce_partials<<<1, 1024, 0, c->stream>>>(partials, nparts, nvals, c->slots, slot0, check_mode,
                                                 slot_rr, slot_tol2, c->flags);
    PGD_LAUNCH_CHECK(c);
    return PGD_OK;
}

// first stage alone (long partial lists in front of a reduction kernel of the caller's own); no-op once the done flag is set
int k_reduce_stage1_pub(Ctx *c, const double *partials, int nparts, int nvals, double *out) {
    const int nb = (nparts + 1023) / 1024;
    k_reduce_stage1<<<nb, TPB, 0, c->stream>>>(partials, nparts, nvals, out, c->flags, 0);
    PGD_LAUNCH_CHECK(c);
    return PGD_OK;
}

// the same final pass into any device array (dest[0..nvals))
int reduce_partials_to(Ctx *c, const double *partials, int nparts, int nvals, double *dest) {
    if (nparts > 8192) {
        const int nb = (nparts + 1023) / 1024;
        PGD_TRY(ensure_work(c, 5, (int64_t)nb * nvals > 4096 ? (int64_t)nb * nvals : 4096));
        k_reduce_stage1<<<nb, TPB, 0, c->stream>>>(partials, nparts, nvals, c->work[5], nullptr, -1);
        partials = c->work[5];
        nparts = nb;
    }
    k_reduce_partials<<<1, 1024, 0, c->stream>>>(partials, nparts, nvals, dest, 0, -1, 0, 0, nullptr);
    PGD_LAUNCH_CHECK(c);
    return PGD_OK;
}

int vec_dot_range(Ctx *c, const double *x, const double *y, int64_t lo, int64_t hi, int slot) {
    const int g = grid_for(hi - lo);
    PGD_TRY(ensure_partials(c, 4 * (int64_t)MAX_VEC_BLOCKS));
    k_dot<<<g, TPB, 0, c->stream>>>(x, y, lo, hi, c->partials);
    PGD_LAUNCH_CHECK(c);
    return reduce_partials(c, c->partials, g, 1, slot, -1, 0, 0);
}

// Calibration stream for the PMC byte model (tools/pmc_calib.py): one pass over a buffer with 8- or 16-byte loads
// per lane, or one pass of 8- / 16-byte stores; the byte count is known, FETCH_SIZE / WRITE_SIZE are read beside it.
template <int W, bool STORE>
__global__ __launch_bounds__(TPB) void k_calib_stream(double *__restrict__ v, int64_t n, double *__restrict__ partials) {
    typedef double d2_t __attribute__((ext_vector_type(2)));
    const int64_t stride = (int64_t)gridDim.x * TPB;
    double acc = 0.0;
    if (W == 8) {
        for (int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x; i < n; i += stride) {
            if (STORE) v[i] = 1.0; else acc += v[i];
        }
    } else {
        d2_t *v2 = reinterpret_cast<d2_t *>(v);
        for (int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x; i < n / 2; i += stride) {
            if (STORE) { d2_t o; o.x = 1.0; o.y = 1.0; v2[i] = o; } else { const d2_t t = v2[i]; acc += t.x + t.y; }
        }
    }
    if (!STORE) {
        __shared__ double s_red[4];
        const double sum = block_sum(acc, s_red);
        if (threadIdx.x == 0) partials[blockIdx.x] = sum;
    }
}

}  // namespace pgd

using namespace pgd;

extern "C" {

int pgd_vec_fill(pgd_handle h, pgd_handle vh, double a) {
    PGD_CTX(c, h);
    Vec *v = get_vec(c, vh);
    if (!v) return fail(c, PGD_ERR_INVALID, "vec_fill: invalid handle");
    if (v->n == 0) return PGD_OK;
    k_fill<<<grid_for(v->n), TPB, 0, c->stream>>>(v->d, a, v->n);
    PGD_LAUNCH_CHECK(c);
    return PGD_OK;
}

int pgd_vec_scale(pgd_handle h, pgd_handle vh, double a) {
    PGD_CTX(c, h);
    Vec *v = get_vec(c, vh);
    if (!v) return fail(c, PGD_ERR_INVALID, "vec_scale: invalid handle");
    if (v->n == 0) return PGD_OK;
    k_scale<<<grid_for(v->n), TPB, 0, c->stream>>>(v->d, a, v->n);
    PGD_LAUNCH_CHECK(c);
    return PGD_OK;
}

// y = a .* x, entry by entry (y may be x or a)
__global__ __launch_bounds__(TPB) void k_vec_mul(double *y, const double *a, const double *x, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x; i < n; i += (int64_t)gridDim.x * TPB) y[i] = a[i] * x[i];
}

int pgd_vec_mul(pgd_handle h, pgd_handle yh, pgd_handle ah, pgd_handle xh) {
    PGD_CTX(c, h);
    Vec *y = get_vec(c, yh), *a = get_vec(c, ah), *x = get_vec(c, xh);
    if (!x || !y || !a || x->n != y->n || a->n != y->n) return fail(c, PGD_ERR_INVALID, "vec_mul: invalid handles or size mismatch");
    if (y->n == 0) return PGD_OK;
    k_vec_mul<<<grid_for(y->n), TPB, 0, c->stream>>>(y->d, a->d, x->d, y->n);
    PGD_LAUNCH_CHECK(c);
    return PGD_OK;
}

int pgd_vec_axpy(pgd_handle h, pgd_handle yh, double a, pgd_handle xh) {
    PGD_CTX(c, h);
    Vec *y = get_vec(c, yh), *x = get_vec(c, xh);
    if (!x || !y || x->n != y->n) return fail(c, PGD_ERR_INVALID, "vec_axpy: invalid handles or size mismatch");
    if (y->n == 0) return PGD_OK;
    k_axpy<<<grid_for(y->n), TPB, 0, c->stream>>>(y->d, a, x->d, y->n);
    PGD_LAUNCH_CHECK(c);
    return PGD_OK;
}

int pgd_vec_lincomb(pgd_handle h, pgd_handle yh, const pgd_handle *xs, const double *coefs, int k) {
    PGD_CTX(c, h);
    Vec *y = get_vec(c, yh);
    if (!y || k < 0 || (k > 0 && (!xs || !coefs))) return fail(c, PGD_ERR_INVALID, "vec_lincomb: bad arguments");
    std::vector<const double *> px((size_t)k);
    for (int t = 0; t < k; ++t) {
        Vec *x = get_vec(c, xs[t]);
        if (!x || x->n != y->n || x == y) return fail(c, PGD_ERR_INVALID, "vec_lincomb: vector %d invalid, of another size or aliasing y", t);
        px[t] = x->d;
    }
    if (y->n == 0) return PGD_OK;
    if (k == 0) {
        k_fill<<<grid_for(y->n), TPB, 0, c->stream>>>(y->d, 0.0, y->n);
        PGD_LAUNCH_CHECK(c);
        return PGD_OK;
    }
    for (int first = 0; first < k; first += 8) {
        LincombArgs A;
        A.k = (k - first < 8) ? k - first : 8;
        A.accumulate = first > 0;
        for (int t = 0; t < 8; ++t) { A.x[t] = px[first + (t < A.k ? t : 0)]; A.c[t] = (t < A.k) ? coefs[first + t] : 0.0; }
        k_lincomb<<<grid_for(y->n), TPB, 0, c->stream>>>(y->d, A, y->n);
    }
    PGD_LAUNCH_CHECK(c);
    return PGD_OK;
}

int pgd_start_residual(pgd_handle h, pgd_handle ah, int k, const double *coefs, pgd_handle bh, pgd_handle rh) {
    PGD_CTX(c, h);
    Vec *b = get_vec(c, bh), *r = get_vec(c, rh);
    if (!b || !r || b == r || !coefs || k < 1) return fail(c, PGD_ERR_INVALID, "start_residual: bad arguments");
    if (!c->gram_op || c->gram_op != ah || c->gram_k != k || c->gram_n != b->n || r->n != b->n || !c->gram_w)
        return fail(c, PGD_ERR_INVALID, "start_residual: the library does not hold the %d products of this operator "
                                        "(pgd_start_gram over all rows with at most 9 vectors must come right before)", k);
    // r = b - sum_j coefs[j] (A v_j): 8 terms per pass, b is the first term of the first pass
    int done = 0;
    bool first = true;
    while (done < k || first) {
        LincombArgs A;
        int t = 0;
        if (first) { A.x[0] = b->d; A.c[0] = 1.0; t = 1; }
        for (; t < 8 && done < k; ++t, ++done) { A.x[t] = c->gram_w + (size_t)done * (size_t)c->gram_n; A.c[t] = -coefs[done]; }
        A.k = t;
        A.accumulate = first ? 0 : 1;
        for (; t < 8; ++t) { A.x[t] = A.x[0]; A.c[t] = 0.0; }
        k_lincomb<<<grid_for(r->n), TPB, 0, c->stream>>>(r->d, A, r->n);
        first = false;
    }
    PGD_LAUNCH_CHECK(c);
    return PGD_OK;
}

int pgd_vec_set(pgd_handle h, pgd_handle vh, const int32_t *idx, const double *val, int64_t n) {
    PGD_CTX(c, h);
    Vec *v = get_vec(c, vh);
    if (!v || n < 0 || (n > 0 && (!idx || !val))) return fail(c, PGD_ERR_INVALID, "vec_set: bad arguments");
    if (n == 0) return PGD_OK;
    // Large lists (the Dirichlet values of a right-hand side, the same in every solve of a fixed-point pass) are kept: indices
    // and values that equal the last call's, word by word, are on the device already - no range check, no upload, no
    // host synchronisation.  Small ones go the direct way through the shared scratch.
    if (n < 4096) {
        for (int64_t i = 0; i < n; ++i)
            if (idx[i] < 0 || idx[i] >= v->n) return fail(c, PGD_ERR_INVALID, "vec_set: index %d out of range", idx[i]);
        PGD_TRY(ensure_ibuf(c, n));
        PGD_TRY(ensure_work(c, 5, n));
        PGD_HIP(c, hipMemcpyAsync(c->ibuf, idx, n * sizeof(int), hipMemcpyHostToDevice, c->stream));
        PGD_HIP(c, hipMemcpyAsync(c->work[5], val, n * sizeof(double), hipMemcpyHostToDevice, c->stream));
        k_set<<<grid_for(n), TPB, 0, c->stream>>>(v->d, c->ibuf, c->work[5], n);
        PGD_LAUNCH_CHECK(c);
        PGD_HIP(c, hipStreamSynchronize(c->stream));   // host buffers are caller-owned
        return PGD_OK;
    }
    const bool same_idx = c->set_idx_on_dev && c->set_idx_host.size() == (size_t)n &&
                          std::memcmp(c->set_idx_host.data(), idx, (size_t)n * sizeof(int32_t)) == 0;
    // (the largest index the kept list was checked against may exceed a shorter vector: the check below covers it)
    bool uploaded = false;
    auto grow = [&](void **buf, int64_t *cap, size_t elem) -> int {       // (contents are not kept: the caller uploads anew)
        if (*cap >= n) return PGD_OK;
        PGD_HIP(c, hipStreamSynchronize(c->stream));
        if (*buf) (void)hipFree(*buf);
        *buf = nullptr; *cap = 0;
        if (hipMalloc(buf, (size_t)n * elem + PAD_BYTES) != hipSuccess) { (void)hipGetLastError(); return fail(c, PGD_ERR_NOMEM, "vec_set: out of device memory"); }
        *cap = n;
        return PGD_OK;
    };
    // The kept lists count as "on the device" only once the copies AND the launch that follow have completed without error
    // (ADVICE r03): until then both flags are down, so a call that fails half way leaves nothing behind that the next call
    // with the same lists would trust - it checks, uploads and synchronises again.
    const bool had_idx = same_idx, had_val_flag = c->set_val_on_dev;
    if (!same_idx) {
        for (int64_t i = 0; i < n; ++i)
            if (idx[i] < 0 || idx[i] >= v->n) return fail(c, PGD_ERR_INVALID, "vec_set: index %d out of range", idx[i]);
        c->set_idx_on_dev = false;
        PGD_TRY(grow(reinterpret_cast<void **>(&c->set_idx), &c->set_idx_cap, sizeof(int)));
        PGD_HIP(c, hipMemcpyAsync(c->set_idx, idx, n * sizeof(int), hipMemcpyHostToDevice, c->stream));
        c->set_idx_host.assign(idx, idx + n);
        c->set_idx_max = 0;
        for (int64_t i = 0; i < n; ++i) if (idx[i] > c->set_idx_max) c->set_idx_max = idx[i];
        uploaded = true;
    } else if (c->set_idx_max >= v->n) {
        return fail(c, PGD_ERR_INVALID, "vec_set: index %d out of range", (int)c->set_idx_max);
    }
    const bool same_val = had_val_flag && c->set_val_host.size() == (size_t)n &&
                          std::memcmp(c->set_val_host.data(), val, (size_t)n * sizeof(double)) == 0;
    c->set_idx_on_dev = false;
    c->set_val_on_dev = false;
    if (!same_val) {
        PGD_TRY(grow(reinterpret_cast<void **>(&c->set_vals), &c->set_vals_cap, sizeof(double)));
        PGD_HIP(c, hipMemcpyAsync(c->set_vals, val, n * sizeof(double), hipMemcpyHostToDevice, c->stream));
        c->set_val_host.assign(val, val + n);
        uploaded = true;
    }
    k_set<<<grid_for(n), TPB, 0, c->stream>>>(v->d, c->set_idx, c->set_vals, n);
    PGD_LAUNCH_CHECK(c);
    if (uploaded) PGD_HIP(c, hipStreamSynchronize(c->stream));   // host buffers are caller-owned
    (void)had_idx;
    c->set_idx_on_dev = true;
    c->set_val_on_dev = true;
    return PGD_OK;
}

int pgd_vec_dot(pgd_handle h, pgd_handle xh, pgd_handle yh, int64_t lo, int64_t hi, double *out) {
    PGD_CTX(c, h);
    Vec *x = get_vec(c, xh), *y = get_vec(c, yh);
    if (!x || !y || x->n != y->n || !out) return fail(c, PGD_ERR_INVALID, "vec_dot: invalid handles or size mismatch");
    if (hi < 0) hi = x->n;
    if (lo < 0 || lo > hi || hi > x->n) return fail(c, PGD_ERR_INVALID, "vec_dot: bad range");
    if (hi == lo) { *out = 0.0; return PGD_OK; }
    PGD_TRY(vec_dot_range(c, x->d, y->d, lo, hi, S_TMP));
    PGD_HIP(c, hipMemcpyAsync(out, c->slots + S_TMP, sizeof(double), hipMemcpyDeviceToHost, c->stream));
    PGD_HIP(c, hipStreamSynchronize(c->stream));
    return PGD_OK;
}

int pgd_calib_stream(pgd_handle h, pgd_handle vh, int bytes_per_lane, int store) {
    PGD_CTX(c, h);
    Vec *v = get_vec(c, vh);
    if (!v || (bytes_per_lane != 8 && bytes_per_lane != 16) || (v->n & 1))
        return fail(c, PGD_ERR_INVALID, "calib_stream: invalid vector or width");
    const int g = 4 * MAX_VEC_BLOCKS;
    PGD_TRY(ensure_partials(c, g));
    if (bytes_per_lane == 8) {
        if (store) k_calib_stream<8, true><<<g, TPB, 0, c->stream>>>(v->d, v->n, c->partials);
        else k_calib_stream<8, false><<<g, TPB, 0, c->stream>>>(v->d, v->n, c->partials);
    } else {
        if (store) k_calib_stream<16, true><<<g, TPB, 0, c->stream>>>(v->d, v->n, c->partials);
        else k_calib_stream<16, false><<<g, TPB, 0, c->stream>>>(v->d, v->n, c->partials);
    }
    PGD_LAUNCH_CHECK(c);
    return PGD_OK;
}

}  // extern "C"
