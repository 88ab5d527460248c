// Operator build (A = sum_t c_t A_t with symmetric Dirichlet elimination), the
// Jacobi-PCG recurrence and the small banded direct solve.
//
// PCG iteration = 3 streaming kernels + 2 one-workgroup reductions:
//   k_spmv_csr<dot>   q = A p, partial p.q                  (12 nnz + 28 n bytes)
//   k_pcg_xr          x += a p; r -= a q; z = r/diag; partial r.z, r.r   (56 n bytes)
//   k_pcg_p           p = z + b p                            (24 n bytes)
// alpha and beta never leave the device: each kernel forms them from the scalar
// bank, and a `done` flag turns every later launch into a no-op, so the host
// only looks at the flag every CHECK_EVERY iterations.  All reductions are
// wavefront-shuffle -> LDS -> fixed-order final pass: bitwise reproducible.
#include <chrono>

#include "pgd_internal.h"

#include <cmath>

#include <cstring>

#include <algorithm>

namespace pgd {

constexpr int MAXT = 8;          // atoms per combine launch
constexpr int CHECK_EVERY = 16;  // PCG iterations enqueued between two host looks at the flag
constexpr int PROF_EAGER_EVERY = 8;   // with launch timing on, one chunk in this many is issued eagerly (with its events)

struct CombineArgs {
    const double *in[MAXT];
    double coef[MAXT];
    int n;
};

// out[k] = sum_t c_t in_t[k]; entries in a Dirichlet column are zeroed
__global__ __launch_bounds__(TPB) void k_combine(CombineArgs A, double *__restrict__ out, const int *__restrict__ cols,
                                                 const uint8_t *__restrict__ colmask, int64_t nnz) {
    for (int64_t k = (int64_t)blockIdx.x * TPB + threadIdx.x; k < nnz; k += (int64_t)gridDim.x * TPB) {
        double s = 0.0;
#pragma unroll
        for (int t = 0; t < MAXT; ++t)
            if (t < A.n) s = fma(A.coef[t], A.in[t][k], s);
        if (colmask && colmask[cols[k]]) s = 0.0;
        out[k] = s;
    }
}

__global__ __launch_bounds__(TPB) void k_mask_set(uint8_t *__restrict__ mask, const int *__restrict__ dofs, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x; i < n; i += (int64_t)gridDim.x * TPB) mask[dofs[i]] = 1;
}

// Dirichlet rows -> identity rows
__global__ __launch_bounds__(TPB) void k_dirichlet_rows(const int *__restrict__ dofs, int64_t n,
                                                        const int *__restrict__ row_ptr, const int *__restrict__ cols,
                                                        double *__restrict__ vals) {
    for (int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x; i < n; i += (int64_t)gridDim.x * TPB) {
        const int r = dofs[i];
        for (int k = row_ptr[r]; k < row_ptr[r + 1]; ++k) vals[k] = (cols[k] == r) ? 1.0 : 0.0;
    }
}

__global__ __launch_bounds__(TPB) void k_diag_inv(const int *__restrict__ row_ptr, const int *__restrict__ cols,
                                                  const double *__restrict__ vals, double *__restrict__ dinv, int64_t n) {
    for (int64_t r = (int64_t)blockIdx.x * TPB + threadIdx.x; r < n; r += (int64_t)gridDim.x * TPB) {
        double d = 0.0;
        for (int k = row_ptr[r]; k < row_ptr[r + 1]; ++k)
            if (cols[k] == (int)r) d = vals[k];
        dinv[r] = 1.0 / d;
    }
}

// r = b - q; z = dinv r; p = z; partials (r.z, r.r, b.b)
__global__ __launch_bounds__(TPB) void k_pcg_init(const double *__restrict__ b, const double *__restrict__ q,
                                                  const double *__restrict__ dinv, double *__restrict__ r,
                                                  double *__restrict__ z, double *__restrict__ p, int64_t lo,
                                                  int64_t hi, double *__restrict__ partials) {
    __shared__ double s_red[4];
    double rz = 0.0, rr = 0.0, bb = 0.0;
    for (int64_t i = lo + (int64_t)blockIdx.x * TPB + threadIdx.x; i < hi; i += (int64_t)gridDim.x * TPB) {
        const double bi = b[i], ri = bi - q[i], zi = dinv[i] * ri;
        r[i] = ri; z[i] = zi; p[i] = zi;
        rz = fma(ri, zi, rz); rr = fma(ri, ri, rr); bb = fma(bi, bi, bb);
    }
    rz = block_sum(rz, s_red);
    rr = block_sum(rr, s_red);
    bb = block_sum(bb, s_red);
    if (threadIdx.x == 0) {
        partials[3 * blockIdx.x + 0] = rz;
        partials[3 * blockIdx.x + 1] = rr;
        partials[3 * blockIdx.x + 2] = bb;
    }
}

__global__ void k_pcg_tol(double *__restrict__ slots, int *__restrict__ flags, double rtol, double atol, int slot_rr,
                          int slot_bb, int slot_tol2) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const double t1 = rtol * rtol * slots[slot_bb], t2 = atol * atol;
    const double tol2 = t1 > t2 ? t1 : t2;
    slots[slot_tol2] = tol2;
    const double rr = slots[slot_rr];
    if (!(rr == rr)) { flags[0] = 1; flags[2] = PGD_ERR_SINGULAR; }
    else if (rr <= tol2) flags[0] = 1;
}

__global__ void k_pcg_check(double *__restrict__ slots, int *__restrict__ flags, int slot_rr, int slot_tol2) {
    if (threadIdx.x != 0 || blockIdx.x != 0 || flags[0]) return;
    const double rr = slots[slot_rr];
    slots[6] = rr;      // last live r.r: later all-reduces of the (frozen) pair slots cannot disturb it
    flags[1] += 1;
    if (!(rr == rr)) { flags[0] = 1; flags[2] = PGD_ERR_SINGULAR; }
    else if (rr <= slots[slot_tol2]) flags[0] = 1;
}

// V2: two consecutive entries per lane (16-byte loads/stores); needs an even `lo`, the odd tail entry
// is handled by one extra lane.
template <bool V2>
__global__ __launch_bounds__(TPB) void k_pcg_xr(double *__restrict__ x, double *__restrict__ r,
                                                const double *__restrict__ p, const double *__restrict__ q,
                                                const double *__restrict__ dinv, double *__restrict__ z, int64_t lo,
                                                int64_t hi, const double *__restrict__ slots, int slot_rz, int slot_pq,
                                                double *__restrict__ partials, const int *__restrict__ flags) {
    if (flags[0]) return;
    __shared__ double s_red[4];
    const double alpha = slots[slot_rz] / slots[slot_pq];
    double rz = 0.0, rr = 0.0;
    if (V2) {
        typedef double d2 __attribute__((ext_vector_type(2)));
        const int64_t npair = (hi - lo) >> 1;
        for (int64_t k = (int64_t)blockIdx.x * TPB + threadIdx.x; k < npair; k += (int64_t)gridDim.x * TPB) {
            const int64_t i = lo + 2 * k;
            const d2 pi = *reinterpret_cast<const d2 *>(p + i), qi = *reinterpret_cast<const d2 *>(q + i);
            const d2 di = *reinterpret_cast<const d2 *>(dinv + i);
            d2 xi = *reinterpret_cast<d2 *>(x + i), ri = *reinterpret_cast<d2 *>(r + i), zi;
            xi.x = fma(alpha, pi.x, xi.x); xi.y = fma(alpha, pi.y, xi.y);
            ri.x = fma(-alpha, qi.x, ri.x); ri.y = fma(-alpha, qi.y, ri.y);
            zi.x = di.x * ri.x; zi.y = di.y * ri.y;
            *reinterpret_cast<d2 *>(x + i) = xi;
            *reinterpret_cast<d2 *>(r + i) = ri;
            *reinterpret_cast<d2 *>(z + i) = zi;
            rz = fma(ri.x, zi.x, rz); rz = fma(ri.y, zi.y, rz);
            rr = fma(ri.x, ri.x, rr); rr = fma(ri.y, ri.y, rr);
        }
        if (((hi - lo) & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
            const int64_t i = hi - 1;
            x[i] = fma(alpha, p[i], x[i]);
            const double ri = fma(-alpha, q[i], r[i]), zi = dinv[i] * ri;
            r[i] = ri; z[i] = zi;
            rz = fma(ri, zi, rz); rr = fma(ri, ri, rr);
        }
    } else {
        for (int64_t i = lo + (int64_t)blockIdx.x * TPB + threadIdx.x; i < hi; i += (int64_t)gridDim.x * TPB) {
            x[i] = fma(alpha, p[i], x[i]);
            const double ri = fma(-alpha, q[i], r[i]), zi = dinv[i] * ri;
            r[i] = ri; z[i] = zi;
            rz = fma(ri, zi, rz); rr = fma(ri, ri, rr);
        }
    }
    rz = block_sum(rz, s_red);
    rr = block_sum(rr, s_red);
    if (threadIdx.x == 0) { partials[2 * blockIdx.x] = rz; partials[2 * blockIdx.x + 1] = rr; }
}

template <bool V2>
__global__ __launch_bounds__(TPB) void k_pcg_p(double *__restrict__ p, const double *__restrict__ z, int64_t lo,
                                               int64_t hi, const double *__restrict__ slots, int slot_num, int slot_den,
                                               const int *__restrict__ flags) {
    if (flags[0]) return;
    const double beta = slots[slot_num] / slots[slot_den];
    if (V2) {
        typedef double d2 __attribute__((ext_vector_type(2)));
        const int64_t npair = (hi - lo) >> 1;
        for (int64_t k = (int64_t)blockIdx.x * TPB + threadIdx.x; k < npair; k += (int64_t)gridDim.x * TPB) {
            const int64_t i = lo + 2 * k;
            const d2 zi = *reinterpret_cast<const d2 *>(z + i);
            d2 pi = *reinterpret_cast<d2 *>(p + i);
            pi.x = fma(beta, pi.x, zi.x); pi.y = fma(beta, pi.y, zi.y);
            *reinterpret_cast<d2 *>(p + i) = pi;
        }
        if (((hi - lo) & 1) && blockIdx.x == 0 && threadIdx.x == 0) p[hi - 1] = fma(beta, p[hi - 1], z[hi - 1]);
    } else {
        for (int64_t i = lo + (int64_t)blockIdx.x * TPB + threadIdx.x; i < hi; i += (int64_t)gridDim.x * TPB)
            p[i] = fma(beta, p[i], z[i]);
    }
}

// ---- single-reduction (Chronopoulos-Gear) form of the same Jacobi-PCG, used by the row-sharded
// solve: ONE all-reduce per iteration instead of two.  Scalars: S[b+0..4] = (r.u, r.r, w.u interior,
// w.u low boundary rows, w.u high boundary rows) - reduced across ranks together - then
// S[b+5] alpha, S[b+6] beta, S[b+7] previous r.u, S[b+8] b.b.
//   p = u + beta p;  s = w + beta s;  x += alpha p;  r -= alpha s;  u = dinv r;  partial (r.u, r.r)
__global__ __launch_bounds__(TPB) void k_cg_update(double *__restrict__ x, double *__restrict__ r, double *__restrict__ u,
                                                   const double *__restrict__ w, double *__restrict__ p,
                                                   double *__restrict__ s, const double *__restrict__ dinv, int64_t lo,
                                                   int64_t hi, const double *__restrict__ slots, int base,
                                                   double *__restrict__ partials, const int *__restrict__ flags) {
    if (flags[0]) return;
    __shared__ double s_red[4];
    const double alpha = slots[base + 5], beta = slots[base + 6];
    double ru = 0.0, rr = 0.0;
    for (int64_t i = lo + (int64_t)blockIdx.x * TPB + threadIdx.x; i < hi; i += (int64_t)gridDim.x * TPB) {
        const double pi = fma(beta, p[i], u[i]), si = fma(beta, s[i], w[i]);
        p[i] = pi; s[i] = si;
        x[i] = fma(alpha, pi, x[i]);
        const double ri = fma(-alpha, si, r[i]), ui = dinv[i] * ri;
        r[i] = ri; u[i] = ui;
        ru = fma(ri, ui, ru); rr = fma(ri, ri, rr);
    }
    ru = block_sum(ru, s_red);
    rr = block_sum(rr, s_red);
    if (threadIdx.x == 0) { partials[2 * blockIdx.x] = ru; partials[2 * blockIdx.x + 1] = rr; }
}

// r = b - q; u = dinv r; p = s = 0; partials (r.u, r.r, b.b)
__global__ __launch_bounds__(TPB) void k_cg_init(const double *__restrict__ b, const double *__restrict__ q,
                                                 const double *__restrict__ dinv, double *__restrict__ r,
                                                 double *__restrict__ u, double *__restrict__ p, double *__restrict__ s,
                                                 int64_t lo, int64_t hi, double *__restrict__ partials) {
    __shared__ double s_red[4];
    double ru = 0.0, rr = 0.0, bb = 0.0;
    for (int64_t i = lo + (int64_t)blockIdx.x * TPB + threadIdx.x; i < hi; i += (int64_t)gridDim.x * TPB) {
        const double bi = b[i], ri = bi - q[i], ui = dinv[i] * ri;
        r[i] = ri; u[i] = ui; p[i] = 0.0; s[i] = 0.0;
        ru = fma(ri, ui, ru); rr = fma(ri, ri, rr); bb = fma(bi, bi, bb);
    }
    ru = block_sum(ru, s_red);
    rr = block_sum(rr, s_red);
    bb = block_sum(bb, s_red);
    if (threadIdx.x == 0) {
        partials[3 * blockIdx.x + 0] = ru;
        partials[3 * blockIdx.x + 1] = rr;
        partials[3 * blockIdx.x + 2] = bb;
    }
}

// The single-reduction recurrence on the diagonally scaled system (see k_scale_in): u = r, so no dinv read and no u
// pass - 10 vector passes instead of 12; sc = d^-1/2 is read only for the true residual norm.
//   p = r + beta p;  s = w + beta s;  x += alpha p;  r -= alpha s;  partial (r.r, sum r^2 / sc^2)
__global__ __launch_bounds__(TPB) void k_cg_update_s(double *__restrict__ x, double *__restrict__ r, const double *__restrict__ w,
                                                     double *__restrict__ p, double *__restrict__ s, const double *__restrict__ sc,
                                                     int64_t lo, int64_t hi, const double *__restrict__ slots, int base,
                                                     double *__restrict__ partials, const int *__restrict__ flags) {
    if (flags[0]) return;
    __shared__ double s_red[4];
    const double alpha = slots[base + 5], beta = slots[base + 6];
    double ru = 0.0, rr = 0.0;
    for (int64_t i = lo + (int64_t)blockIdx.x * TPB + threadIdx.x; i < hi; i += (int64_t)gridDim.x * TPB) {
        const double pi = fma(beta, p[i], r[i]), si = fma(beta, s[i], w[i]);
        p[i] = pi; s[i] = si;
        x[i] = fma(alpha, pi, x[i]);
        const double ri = fma(-alpha, si, r[i]), ti = ri / sc[i];
        r[i] = ri;
        ru = fma(ri, ri, ru); rr = fma(ti, ti, rr);
    }
    ru = block_sum(ru, s_red);
    rr = block_sum(rr, s_red);
    if (threadIdx.x == 0) { partials[2 * blockIdx.x] = ru; partials[2 * blockIdx.x + 1] = rr; }
}

// r = sc b - q; p = s = 0; partials (r.r, true r.r, b.b)
__global__ __launch_bounds__(TPB) void k_cg_init_s(const double *__restrict__ b, const double *__restrict__ q,
                                                   const double *__restrict__ sc, double *__restrict__ r, double *__restrict__ p,
                                                   double *__restrict__ s, int64_t lo, int64_t hi, double *__restrict__ partials) {
    __shared__ double s_red[4];
    double ru = 0.0, rr = 0.0, bb = 0.0;
    for (int64_t i = lo + (int64_t)blockIdx.x * TPB + threadIdx.x; i < hi; i += (int64_t)gridDim.x * TPB) {
        const double bi = b[i], si = sc[i], ri = si * bi - q[i], ti = ri / si;
        r[i] = ri; p[i] = 0.0; s[i] = 0.0;
        ru = fma(ri, ri, ru); rr = fma(ti, ti, rr); bb = fma(bi, bi, bb);
    }
    ru = block_sum(ru, s_red);
    rr = block_sum(rr, s_red);
    bb = block_sum(bb, s_red);
    if (threadIdx.x == 0) {
        partials[3 * blockIdx.x + 0] = ru;
        partials[3 * blockIdx.x + 1] = rr;
        partials[3 * blockIdx.x + 2] = bb;
    }
}

// v <- sqrt(v) (dinv -> d^-1/2) / x <- x / sc / x <- x sc on a row range
__global__ __launch_bounds__(TPB) void k_vec_sqrt(double *__restrict__ v, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x; i < n; i += (int64_t)gridDim.x * TPB) v[i] = sqrt(v[i]);
}

__global__ __launch_bounds__(TPB) void k_vec_div_mul(double *__restrict__ x, const double *__restrict__ sc, int64_t n, int mul) {
    for (int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x; i < n; i += (int64_t)gridDim.x * TPB)
        x[i] = mul ? x[i] * sc[i] : x[i] / sc[i];
}

// The scaled single-reduction recurrence with the scalar step folded in: every workgroup forms alpha / beta itself from
// the all-reduced slots (pure functions of them), workgroup 0 also keeps the iteration count, the stop flag and the
// two scalars the NEXT iteration's formula needs - in the slot set of the other parity, so nobody reads what it
// writes.  S[b..b+4] all-reduced (r.r, true r.r, w.r, 0, 0); set q at b + 9 + 2 q: (previous alpha, previous r.r).
__global__ __launch_bounds__(TPB) void k_cg_update_s2(double *__restrict__ x, double *__restrict__ r, const double *__restrict__ w,
                                                      double *__restrict__ p, double *__restrict__ s, const double *__restrict__ sc,
                                                      int64_t lo, int64_t hi, double *__restrict__ slots, int base, int parity,
                                                      double *__restrict__ partials, int *__restrict__ flags) {
    if (flags[0]) return;
    __shared__ double s_red[4];
    const double g = slots[base], rr_in = slots[base + 1], d = slots[base + 2] + slots[base + 3] + slots[base + 4];
    const double a_prev = slots[base + 9 + 2 * parity], g_prev = slots[base + 10 + 2 * parity];
    const bool bad = !(rr_in == rr_in) || !(d == d), done = rr_in <= slots[S_TOL2];
    const double beta = g / g_prev, alpha = g / (d - beta * g / a_prev);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        flags[1] += 1;
        slots[6] = rr_in;
        if (bad) { flags[0] = 1; flags[2] = PGD_ERR_SINGULAR; }
        else if (done) flags[0] = 1;
        slots[base + 9 + 2 * (1 - parity)] = alpha;
        slots[base + 10 + 2 * (1 - parity)] = g;
    }
    if (bad || done) return;                                   // uniform: every workgroup sees the same slots
    double ru = 0.0, rr = 0.0;
    for (int64_t i = lo + (int64_t)blockIdx.x * TPB + threadIdx.x; i < hi; i += (int64_t)gridDim.x * TPB) {
        const double pi = fma(beta, p[i], r[i]), si = fma(beta, s[i], w[i]);
        p[i] = pi; s[i] = si;
        x[i] = fma(alpha, pi, x[i]);
        const double ri = fma(-alpha, si, r[i]), ti = ri / sc[i];
        r[i] = ri;
        ru = fma(ri, ri, ru); rr = fma(ti, ti, rr);
    }
    ru = block_sum(ru, s_red);
    rr = block_sum(rr, s_red);
    if (threadIdx.x == 0) { partials[2 * blockIdx.x] = ru; partials[2 * blockIdx.x + 1] = rr; }
}

// S[b], S[b+1] <- sums of the update's partial pairs; S[b+2] <- sum of the product's partials; S[b+3] = S[b+4] = 0
__global__ __launch_bounds__(1024) void k_reduce_two(const double *__restrict__ pa, int na, const double *__restrict__ pb, int nb,
                                                     double *__restrict__ slots, int base, const int *__restrict__ flags) {
    __shared__ double s_w[16];
    if (flags[0]) return;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int v = 0; v < 3; ++v) {
        const double *src = v < 2 ? pa : pb;
        const int n = v < 2 ? na : nb, stride = v < 2 ? 2 : 1, off = v < 2 ? v : 0;
        double acc = 0.0;
        for (int i = threadIdx.x; i < n; i += 1024) acc += src[(int64_t)i * stride + off];
        acc = wave_sum(acc);
        __syncthreads();
        if (lane == 0) s_w[wv] = acc;
        __syncthreads();
        if (threadIdx.x == 0) {
            double t = 0.0;
#pragma unroll
            for (int k = 0; k < 16; ++k) t += s_w[k];
            slots[base + v] = t;
        }
    }
    if (threadIdx.x == 0) { slots[base + 3] = 0.0; slots[base + 4] = 0.0; }
}

// after the all-reduce: next alpha / beta, iteration count, convergence
__global__ void k_cg_scalars(double *__restrict__ slots, int *__restrict__ flags, int base, int init, double rtol,
                             double atol) {
    if (threadIdx.x != 0 || blockIdx.x != 0 || flags[0]) return;
    const double g = slots[base], rr = slots[base + 1], d = slots[base + 2] + slots[base + 3] + slots[base + 4];
    if (init) {
        const double t1 = rtol * rtol * slots[base + 8], t2 = atol * atol;
        slots[S_TOL2] = t1 > t2 ? t1 : t2;
    } else {
        flags[1] += 1;
    }
    slots[6] = rr;
    if (!(rr == rr) || !(d == d)) { flags[0] = 1; flags[2] = PGD_ERR_SINGULAR; return; }
    if (rr <= slots[S_TOL2]) { flags[0] = 1; return; }
    const double beta = init ? 0.0 : g / slots[base + 7];
    const double alpha = init ? g / d : g / (d - beta * g / slots[base + 5]);
    slots[base + 5] = alpha;
    slots[base + 6] = beta;
    slots[base + 7] = g;
}

// ---------------------------------------------------------------- banded LU (small systems)
// One workgroup; the band lives in LDS when it fits.  ab(i,j) is stored at
// W[(i - j + kl + ku) + j * ld], ld = 2 kl + ku + 1 (LAPACK dgbtrf layout: kl extra
// super-diagonals for the fill of partial pivoting).  The elimination itself is a
// dependency chain of n steps with O(kl (kl+ku)) work each: one lane does it.
constexpr int BAND_LDS_DOUBLES = 16384;   // 128 KiB

__global__ __launch_bounds__(TPB) void k_band_solve(const int *__restrict__ row_ptr, const int *__restrict__ cols,
                                                    const double *__restrict__ vals, const double *__restrict__ b,
                                                    double *__restrict__ x, int n, int kl, int ku,
                                                    double *__restrict__ gwork, int use_lds, int *__restrict__ flags) {
    __shared__ double s_w[BAND_LDS_DOUBLES];
    const int ld = 2 * kl + ku + 1;
    double *W = use_lds ? s_w : gwork;          // n*ld band entries, then n rhs entries
    double *rhs = W + (int64_t)n * ld;
    for (int64_t k = threadIdx.x; k < (int64_t)n * ld; k += TPB) W[k] = 0.0;
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += TPB) {
        rhs[i] = b[i];
        for (int k = row_ptr[i]; k < row_ptr[i + 1]; ++k) {
            const int j = cols[k];
            W[(i - j + kl + ku) + (int64_t)j * ld] = vals[k];
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const int kv = kl + ku;
        int singular = 0;
        for (int j = 0; j < n; ++j) {
            const int imax = min(j + kl, n - 1), cmax = min(j + kv, n - 1);
            int piv = j;
            double best = fabs(W[kv + (int64_t)j * ld]);
            for (int i = j + 1; i <= imax; ++i) {
                const double t = fabs(W[(i - j + kv) + (int64_t)j * ld]);
                if (t > best) { best = t; piv = i; }
            }
            if (best == 0.0) { singular = 1; break; }
            if (piv != j) {
                for (int cc = j; cc <= cmax; ++cc) {
                    const int64_t a1 = (j - cc + kv) + (int64_t)cc * ld, a2 = (piv - cc + kv) + (int64_t)cc * ld;
                    const double t = W[a1]; W[a1] = W[a2]; W[a2] = t;
                }
                const double t = rhs[j]; rhs[j] = rhs[piv]; rhs[piv] = t;
            }
            const double dinv = 1.0 / W[kv + (int64_t)j * ld];
            for (int i = j + 1; i <= imax; ++i) {
                const double l = W[(i - j + kv) + (int64_t)j * ld] * dinv;
                if (l != 0.0) {
                    for (int cc = j + 1; cc <= cmax; ++cc)
                        W[(i - cc + kv) + (int64_t)cc * ld] -= l * W[(j - cc + kv) + (int64_t)cc * ld];
                    rhs[i] -= l * rhs[j];
                }
            }
        }
        if (!singular) {
            for (int j = n - 1; j >= 0; --j) {
                const int cmax = min(j + kv, n - 1);
                double s = rhs[j];
                for (int cc = j + 1; cc <= cmax; ++cc) s -= W[(j - cc + kv) + (int64_t)cc * ld] * rhs[cc];
                rhs[j] = s / W[kv + (int64_t)j * ld];
            }
        } else {
            flags[2] = PGD_ERR_SINGULAR;
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += TPB) x[i] = rhs[i];
}

// ---- the same Jacobi-PCG on the symmetrically scaled system  A~ = D^-1/2 A D^-1/2,  x~ = D^1/2 x,  b~ = D^-1/2 b:
// plain CG on A~ walks exactly the iterates of Jacobi-PCG on A (x_k = D^-1/2 x~_k, r~ = D^-1/2 r, r~.r~ = r.z), but no
// kernel reads dinv or writes z any more: 9 vector passes per iteration instead of 11.  The stop test stays the TRUE
// residual norm r.r = sum d_i r~_i^2; it needs one more read (s), so it is only formed in the "exact phase", entered
// when d_min r~.r~ - a lower bound of r.r - comes within 10^4 of the tolerance (see k_reduce_partials, check_mode 2).

// s = sqrt(dinv) (= d^-1/2); x <- x / s; the largest dinv (1 / d_min) by an ordered-bits atomic max
__global__ __launch_bounds__(TPB) void k_scale_in(const double *__restrict__ dinv, double *__restrict__ s,
                                                  double *__restrict__ x, int64_t n, unsigned long long *__restrict__ dmax_bits) {
    double mx = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x; i < n; i += (int64_t)gridDim.x * TPB) {
        const double di = dinv[i], si = sqrt(di);
        s[i] = si;
        x[i] = x[i] / si;
        mx = fmax(mx, di);
    }
    for (int o = 32; o > 0; o >>= 1) mx = fmax(mx, __shfl_xor(mx, o));
    if ((threadIdx.x & 63) == 0) {
        // positive doubles order like integers.  A look first: 8192 waves taking turns at one address cost 80 us of the kernel's 100
        // at 128^3, and on a uniform grid all but the first few hold a value that is already there
        const unsigned long long mine = (unsigned long long)__double_as_longlong(mx);
        if (mine > __hip_atomic_load(dmax_bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(dmax_bits, mine);
    }
}

// couplings of D^-1/2 A D^-1/2 from those of a stencil A (pgd_pcg_solve): the arithmetic of k_combine_dia (1 / diagonal),
// k_scale_in (its root) and k_dia_scale (value times the product of the two scale factors; the diagonal: set to 1, or value
// times s times s) on one free row
struct StencilTuple { double c[8]; };
__global__ void k_stencil_derive(StencilTuple A, int unit, double *__restrict__ out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const double dinv = 1.0 / A.c[0], si = sqrt(dinv);
    out[0] = unit ? 1.0 : A.c[0] * si * si;
    for (int s = 1; s < 8; ++s) { double v = A.c[s]; v *= si * si; out[s] = v; }
}

__global__ void k_dmin_slot(const unsigned long long *__restrict__ dmax_bits, double *__restrict__ slots, int *__restrict__ flags) {
    const double dmax_inv = __longlong_as_double((long long)dmax_bits[0]);
    slots[S_DMIN] = dmax_inv > 0.0 ? 1.0 / dmax_inv : 0.0;
    if (!(dmax_inv > 0.0) || !(dmax_inv < 1e300)) { flags[0] = 1; flags[2] = PGD_ERR_SINGULAR; }   // a non-positive diagonal
}

// r = s b - q; p = r; partials (r.r, sum r^2 / s^2 = true r.r, b.b)
__global__ __launch_bounds__(TPB) void k_pcg_init_s(const double *__restrict__ b, const double *__restrict__ q,
                                                    const double *__restrict__ s, double *__restrict__ r,
                                                    double *__restrict__ p, int64_t n, double *__restrict__ partials) {
    __shared__ double s_red[4];
    double rz = 0.0, rr = 0.0, bb = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x; i < n; i += (int64_t)gridDim.x * TPB) {
        const double bi = b[i], si = s[i], ri = si * bi - q[i], ti = ri / si;
        r[i] = ri; p[i] = ri;
        rz = fma(ri, ri, rz); rr = fma(ti, ti, rr); bb = fma(bi, bi, bb);
    }
    rz = block_sum(rz, s_red);
    rr = block_sum(rr, s_red);
    bb = block_sum(bb, s_red);
    if (threadIdx.x == 0) {
        partials[3 * blockIdx.x + 0] = rz;
        partials[3 * blockIdx.x + 1] = rr;
        partials[3 * blockIdx.x + 2] = bb;
    }
}

// alpha = S[rz] / S[pq]; x += alpha p; r -= alpha q; partials (r.r, exact phase ? sum r^2 / s^2 : r.r)
__global__ __launch_bounds__(TPB) void k_pcg_xr_s(double *__restrict__ x, double *__restrict__ r, const double *__restrict__ p,
                                                  const double *__restrict__ q, const double *__restrict__ s, int64_t n,
                                                  const double *__restrict__ slots, int slot_rz, int slot_pq,
                                                  double *__restrict__ partials, const int *__restrict__ flags) {
    if (flags[0]) return;
    __shared__ double s_red[4];
    typedef double d2 __attribute__((ext_vector_type(2)));
    const double alpha = slots[slot_rz] / slots[slot_pq];
    const bool exact = flags[3] != 0;
    double rz = 0.0, rr = 0.0;
    const int64_t npair = n >> 1;
    for (int64_t k = (int64_t)blockIdx.x * TPB + threadIdx.x; k < npair; k += (int64_t)gridDim.x * TPB) {
        const int64_t i = 2 * k;
        const d2 pi = *reinterpret_cast<const d2 *>(p + i), qi = *reinterpret_cast<const d2 *>(q + i);
        d2 xi = *reinterpret_cast<d2 *>(x + i), ri = *reinterpret_cast<d2 *>(r + i);
        xi.x = fma(alpha, pi.x, xi.x); xi.y = fma(alpha, pi.y, xi.y);
        ri.x = fma(-alpha, qi.x, ri.x); ri.y = fma(-alpha, qi.y, ri.y);
        *reinterpret_cast<d2 *>(x + i) = xi;
        *reinterpret_cast<d2 *>(r + i) = ri;
        rz = fma(ri.x, ri.x, rz); rz = fma(ri.y, ri.y, rz);
        if (exact) {
            const d2 si = *reinterpret_cast<const d2 *>(s + i);
            const double tx = ri.x / si.x, ty = ri.y / si.y;
            rr = fma(tx, tx, rr); rr = fma(ty, ty, rr);
        }
    }
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
        const int64_t i = n - 1;
        x[i] = fma(alpha, p[i], x[i]);
        const double ri = fma(-alpha, q[i], r[i]);
        r[i] = ri;
        rz = fma(ri, ri, rz);
        if (exact) { const double t = ri / s[i]; rr = fma(t, t, rr); }
    }
    rz = block_sum(rz, s_red);
    rr = block_sum(rr, s_red);
    if (threadIdx.x == 0) { partials[2 * blockIdx.x] = rz; partials[2 * blockIdx.x + 1] = exact ? rr : rz; }
}

// The same step with the x update moved into the p kernel (large systems, where bytes set the pace): this kernel only
// forms r -= alpha q and the partial sums - 24 B per row instead of 48 ...
__global__ __launch_bounds__(TPB) void k_pcg_r_s(double *__restrict__ r, const double *__restrict__ q, const double *__restrict__ s,
                                                 int64_t n, const double *__restrict__ slots, int slot_rz, int slot_pq,
                                                 double *__restrict__ partials, const int *__restrict__ flags) {
    if (flags[0]) return;
    __shared__ double s_red[4];
    typedef double d2 __attribute__((ext_vector_type(2)));
    const double alpha = slots[slot_rz] / slots[slot_pq];
    const bool exact = flags[3] != 0;
    double rz = 0.0, rr = 0.0;
    const int64_t npair = n >> 1;
    for (int64_t k = (int64_t)blockIdx.x * TPB + threadIdx.x; k < npair; k += (int64_t)gridDim.x * TPB) {
        const int64_t i = 2 * k;
        const d2 qi = *reinterpret_cast<const d2 *>(q + i);
        d2 ri = *reinterpret_cast<d2 *>(r + i);
        ri.x = fma(-alpha, qi.x, ri.x); ri.y = fma(-alpha, qi.y, ri.y);
        *reinterpret_cast<d2 *>(r + i) = ri;
        rz = fma(ri.x, ri.x, rz); rz = fma(ri.y, ri.y, rz);
        if (exact) {
            const d2 si = *reinterpret_cast<const d2 *>(s + i);
            const double tx = ri.x / si.x, ty = ri.y / si.y;
            rr = fma(tx, tx, rr); rr = fma(ty, ty, rr);
        }
    }
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
        const int64_t i = n - 1;
        const double ri = fma(-alpha, q[i], r[i]);
        r[i] = ri;
        rz = fma(ri, ri, rz);
        if (exact) { const double t = ri / s[i]; rr = fma(t, t, rr); }
    }
    rz = block_sum(rz, s_red);
    rr = block_sum(rr, s_red);
    if (threadIdx.x == 0) { partials[2 * blockIdx.x] = rz; partials[2 * blockIdx.x + 1] = exact ? rr : rz; }
}

// ... and this one x += alpha p (the alpha of the iteration that is ending; p is read here anyway) before p = r + beta p:
// 40 B per row instead of 24.  9 vector passes per iteration become 8; the same operations on the same operands, so x,
// r and p are bit-identical to the k_pcg_xr_s / k_pcg_p pair.  When the convergence test of this iteration has set the
// done flag the kernel is a no-op like every later launch - the last x update is then applied by k_scale_out.
__global__ __launch_bounds__(TPB) void k_pcg_px_s(double *__restrict__ x, double *__restrict__ p, const double *__restrict__ r,
                                                  int64_t n, const double *__restrict__ slots, int slot_new, int slot_old,
                                                  int slot_pq, const int *__restrict__ flags) {
    if (flags[0]) return;
    typedef double d2 __attribute__((ext_vector_type(2)));
    const double alpha = slots[slot_old] / slots[slot_pq], beta = slots[slot_new] / slots[slot_old];
    const int64_t npair = n >> 1;
    for (int64_t k = (int64_t)blockIdx.x * TPB + threadIdx.x; k < npair; k += (int64_t)gridDim.x * TPB) {
        const int64_t i = 2 * k;
        const d2 ri = *reinterpret_cast<const d2 *>(r + i);
        d2 pi = *reinterpret_cast<d2 *>(p + i), xi = *reinterpret_cast<d2 *>(x + i);
        xi.x = fma(alpha, pi.x, xi.x); xi.y = fma(alpha, pi.y, xi.y);
        pi.x = fma(beta, pi.x, ri.x); pi.y = fma(beta, pi.y, ri.y);
        *reinterpret_cast<d2 *>(x + i) = xi;
        *reinterpret_cast<d2 *>(p + i) = pi;
    }
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
        const int64_t i = n - 1;
        x[i] = fma(alpha, p[i], x[i]);
        p[i] = fma(beta, p[i], r[i]);
    }
}

// ---- single-sync form of the same scaled recurrence (structured grids, large systems).
// Textbook CG has two reductions per iteration: p.q (for alpha) and r'.r' of the new residual (for beta).  With
// alpha = r.r / p.q and r' = r - alpha q:  r'.r' = r.r - 2 alpha r.q + alpha^2 q.q = alpha^2 q.q - r.r  (r.q = p.q by the
// conjugacy of p), so beta = r'.r' / r.r is known as soon as the product has left p.q and q.q - and ONE vector kernel can
// apply x += alpha p, r -= alpha q, p = r + beta p (7 vector passes, 3 launches per iteration instead of 8 and 5).  Only
// beta uses the predicted r'.r' (relative error ~ 2 eps / beta); the MEASURED r.r of every new residual, summed by the
// same vector kernel, feeds the next alpha and the stop test, so nothing accumulates.  Iterates equal the textbook ones
// up to rounding.  Slots: S1_RZ (measured r~.r~ of the current residual), S1_RR (its true r.r, exact phase),
// S1_ALPHA, S1_BETA.
enum { S1_RZ = 24, S1_RR = 25, S1_ALPHA = 26, S1_BETA = 27, S1_PQ = 28, S1_QQ = 29,
       S1_ALPHA_PREV = 40, S1_BETA_PREV = 41, S1_PEND = 42 };      // the lagged x update (below)
constexpr double LAG_MIN_BETA = 0.01;
enum { S1F_ALPHA = 44 /* +parity */, S1F_BETA = 46 /* +parity */, S1F_EXACT = 48 /* +parity */ };   // scalar step inside the update kernel

__device__ __forceinline__ void pcg1_finish(double *slots, int *flags, double pq, double qq, double rz, double rr, int slot_alpha,
                                            int slot_beta, bool keep_prev = false) {
    const double tol2 = slots[S_TOL2];
    if (!(rz == rz) || !(pq == pq)) { flags[0] = 1; flags[2] = PGD_ERR_SINGULAR; return; }
    if (flags[3]) { if (rr <= tol2) { flags[0] = 1; return; } }
    else if (rz * slots[S_DMIN] <= 1e4 * tol2) flags[3] = 1;
    if (!(rz > 0.0)) { flags[0] = 1; return; }          // the residual vanished exactly: nothing left to do (and no 0 / 0 below)
    const double alpha = rz / pq;
    double rnew = alpha * alpha * qq - rz;
    if (!(rnew > 0.0)) rnew = 0.0;                      // rounding below zero (beta at the eps level): a steepest-descent restart
    if (keep_prev) { slots[S1_ALPHA_PREV] = slots[slot_alpha]; slots[S1_BETA_PREV] = slots[slot_beta]; }   // the lagged x update needs them
    slots[slot_alpha] = alpha;
    slots[slot_beta] = rnew / rz;
    flags[1] += 1;
}

// one workgroup: sums the product's (p.q, q.q) pairs and the previous vector kernel's (r~.r~, true r.r) pairs, runs the stop
// test on that residual and, if the solve goes on, forms alpha and beta for the update that follows
__global__ __launch_bounds__(1024) void k_pcg1_scalars(const double *__restrict__ prod, int nprod, const double *__restrict__ vecp,
                                                       int nvec, double *__restrict__ slots, int *__restrict__ flags) {
    __shared__ double s_w[16];
    if (flags[0]) return;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    // the four sums side by side: waves 4 v .. 4 v + 3 add value v (fixed order: 256 strided lanes, 8 accumulators each)
    {
        const int v = wv >> 2, t = threadIdx.x & 255;
        const double *src = v < 2 ? prod : vecp;
        const int n = v < 2 ? nprod : nvec, off = v & 1;
        double a8[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        int i = t;
        for (; i + 7 * 256 < n; i += 8 * 256) {
#pragma unroll
            for (int u = 0; u < 8; ++u) a8[u] += src[2 * (int64_t)(i + u * 256) + off];
        }
        for (int u = 0; i < n; i += 256, ++u) a8[u & 7] += src[2 * (int64_t)i + off];
        const double acc = wave_sum(((a8[0] + a8[1]) + (a8[2] + a8[3])) + ((a8[4] + a8[5]) + (a8[6] + a8[7])));
        if (lane == 0) s_w[wv] = acc;
    }
    __syncthreads();
    double out[4];
#pragma unroll
    for (int v = 0; v < 4; ++v) out[v] = (s_w[4 * v] + s_w[4 * v + 1]) + (s_w[4 * v + 2] + s_w[4 * v + 3]);
    if (threadIdx.x != 0) return;
    slots[S1_PQ] = out[0]; slots[S1_QQ] = out[1]; slots[S1_RZ] = out[2]; slots[S1_RR] = out[3];
    pcg1_finish(slots, flags, out[0], out[1], out[2], out[3], S1_ALPHA, S1_BETA, true);
}

// the row-sharded solve splits the same step around its all-reduce: local sums -> slots[base .. base + 3] (+ a zero) ...
__global__ __launch_bounds__(1024) void k_pcg1_sums(const double *__restrict__ prod, int nprod, const double *__restrict__ vecp,
                                                    int nvec, double *__restrict__ slots, int base, const int *__restrict__ flags) {
    __shared__ double s_w[16];
    if (flags[0]) return;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int v = wv >> 2, t = threadIdx.x & 255;
    const double *src = v < 2 ? prod : vecp;
    const int n = v < 2 ? nprod : nvec, off = v & 1;
    double a8[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int i = t;
    for (; i + 7 * 256 < n; i += 8 * 256) {
#pragma unroll
        for (int u = 0; u < 8; ++u) a8[u] += src[2 * (int64_t)(i + u * 256) + off];
    }
    for (int u = 0; i < n; i += 256, ++u) a8[u & 7] += src[2 * (int64_t)i + off];
    const double acc = wave_sum(((a8[0] + a8[1]) + (a8[2] + a8[3])) + ((a8[4] + a8[5]) + (a8[6] + a8[7])));
    if (lane == 0) s_w[wv] = acc;
    __syncthreads();
    if (threadIdx.x < 4) slots[base + threadIdx.x] = (s_w[4 * threadIdx.x] + s_w[4 * threadIdx.x + 1]) + (s_w[4 * threadIdx.x + 2] + s_w[4 * threadIdx.x + 3]);
    if (threadIdx.x == 4) slots[base + 4] = 0.0;
}

// ... and, after the all-reduce of those slots, the stop test, alpha and beta (into slots[base + 5], [base + 6])
__global__ void k_pcg1_finish(double *__restrict__ slots, int *__restrict__ flags, int base) {
    if (threadIdx.x != 0 || blockIdx.x != 0 || flags[0]) return;
    slots[6] = slots[base + 3];                          // the last measured true r.r, for the report
    pcg1_finish(slots, flags, slots[base], slots[base + 1], slots[base + 2], slots[base + 3], base + 5, base + 6, true);
}

// initial residual of the sharded solve: tolerance from the all-reduced b.b, first stop test.
// EXACT PHASE (slot_p8 >= 0): like pgd_pcg_solve the sharded single-sync loop measures the TRUE r.r = sum r~_i^2 / s_i^2 (one more
// vector read per row in the update) only near the end - once d_min r~.r~, a lower bound of it, comes within 10^4 of the
// tolerance.  d_min must be the same number on every rank and the bindings all-reduce SUMS: slots[slot_p8] holds the
// all-reduced sum of s_i^16 = (1 / d_i)^8 over all owned rows, and 1 / d_min = max 1 / d_i <= (sum (1 / d_i)^8)^(1/8), so
// (sum)^(-1/8) is a lower bound of d_min (at most n^(1/8) below it: the exact phase starts a few dozen iterations early).  A
// sum that is not a positive finite number means "exact from the start", which is also what slot_p8 < 0 asks for.
__global__ void k_pcg1_tol(double *__restrict__ slots, int *__restrict__ flags, int base, double rtol, double atol, int slot_p8) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const double t1 = rtol * rtol * slots[base + 8], t2 = atol * atol, rr = slots[base + 1];
    slots[S_TOL2] = t1 > t2 ? t1 : t2;
    slots[6] = rr;
    int exact = 1;
    if (slot_p8 >= 0) {
        const double p8 = slots[slot_p8];
        if (p8 > 0.0 && p8 < 1e300) { slots[S_DMIN] = 1.0 / sqrt(sqrt(sqrt(p8))); exact = 0; }
        slots[slot_p8] = 0.0;                            // the slot is a zero in the loop's all-reduces
    }
    flags[3] = exact;
    slots[S1F_EXACT] = slots[S1F_EXACT + 1] = exact ? 1.0 : 0.0;
    if (!(rr == rr)) { flags[0] = 1; flags[2] = PGD_ERR_SINGULAR; }
    else if (rr <= slots[S_TOL2]) flags[0] = 1;
}

// partial sums of s_i^16 over [lo, hi) (s = d^-1/2: the 8-norm of 1 / d, k_pcg1_tol) and of r~_i^2 / s_i^2 (the true r.r of
// the final residual, for the report of a solve that ended outside its exact phase)
__global__ __launch_bounds__(TPB) void k_pcg1_aux(const double *__restrict__ s, const double *__restrict__ r, int64_t lo, int64_t hi,
                                                  double *__restrict__ partials) {
    __shared__ double s_red[4];
    double acc = 0.0;
    for (int64_t i = lo + (int64_t)blockIdx.x * TPB + threadIdx.x; i < hi; i += (int64_t)gridDim.x * TPB) {
        if (r) { const double t = r[i] / s[i]; acc = fma(t, t, acc); }
        else { double t = s[i] * s[i]; t *= t; t *= t; acc += t * t; }
    }
    acc = block_sum(acc, s_red);
    if (threadIdx.x == 0) partials[blockIdx.x] = acc;
}

// x += alpha p; r -= alpha q; p = r + beta p; partial sums (r~.r~, exact phase ? sum r^2 / s^2 : r~.r~) of the new residual.
// LAGGED x UPDATE (lag = 1 / 2, pgd_pcg_solve on large structured systems): x is an output accumulator - nothing of the
// recurrence (r, p, q, alpha, beta, the stop test) reads it - so it need not be touched in every iteration.  Iterations with an
// even index (lag = 1) leave x alone when beta >= LAG_MIN_BETA: 5 vector passes instead of 7; the next iteration (lag = 2) adds
// both terms, x += alpha' p' + alpha p, with the previous direction taken back out of the recurrence that formed the present
// one: p = r + beta' p'  =>  p' = (p - r) / beta' (r and p being what this kernel reads anyway, alpha' and beta' kept by
// k_pcg1_scalars).  The rounding of p' is eps |p| / beta' <= 100 eps |p|: the two-term update differs from two single ones at
// the level of the rounding of x itself.
// NT: q, r and x are streamed with non-temporal loads and stores - they are touched by no other kernel of the iteration (q is
// read here once) - while p, which the product reads next, keeps the default policy and with it its place in the Infinity Cache:
// 145 -> 124 us for this kernel AND 88 -> 81 us for the product behind it (7.4 -> 8.3 passes/s; loads alone + 5 %, stores alone 0,
// p streamed as well - 3 %).  slots[S1_PEND] (written by workgroup 0 of the lag = 1 launches, read by the lag = 2
// ones and by k_scale_out when the solve ends between the two) says whether a term is outstanding.  lag = 0: every iteration.
// FOLD (the row-sharded loop, whose sums arrive all-reduced in slots[fold_base .. + 3]: p.q, q.q, r~.r~ and the true r.r of the
// residual this launch starts from): EVERY workgroup runs the scalar step itself - stop test, alpha, beta from those four
// numbers, bit for bit the same in all of them - and workgroup 0 keeps the books (iteration count, flags, the report's r.r):
// k_pcg1_finish and its launch are gone from the iteration.  Nothing a workgroup reads is written by another one of the same
// launch: alpha / beta alternate between two slot pairs with the parity `par` of the iteration (the two-term x update reads the
// previous pair), the sums are rewritten by the NEXT iteration's k_pcg1_sums, and the sharded form is in its exact phase
// (flags[3] = 1, the true r.r in every iteration) from the start.  A workgroup that starts late and finds the done flag set by
// workgroup 0 returns - what its own test would have told it.
// PUSH (the direct halo of the sharded loop, pgd_comm.hip; FOLD launches with an even first row and even plane sizes): the boundary
// planes of the NEW p go straight into the neighbours' ghost planes as well - write-through stores at system scope - and the last of
// the workgroups that hold such rows (a ticket) posts the sequence number and polls this rank's own: k_halo_push without its launch.
// A launch that returns early (the solve is done - on every rank alike) still posts, so that a neighbour which failed locally and
// keeps pushing from launches of its own is never left waiting; it does not wait itself.
__device__ __forceinline__ void push_store(double *dst, double v) {
    __hip_atomic_store(reinterpret_cast<unsigned long long *>(dst), (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ void push_post(const PushArgs &P) {
    if (P.post_lo) __hip_atomic_store(P.post_lo, P.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    if (P.post_hi) __hip_atomic_store(P.post_hi, P.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

template <bool NT, bool FOLD, bool PUSH = false>
__global__ __launch_bounds__(TPB) void k_pcg1_update(double *__restrict__ x, double *__restrict__ r, double *__restrict__ p,
                                                     const double *__restrict__ q, const double *__restrict__ s, int64_t lo,
                                                     int64_t hi, double *__restrict__ slots, int slot_alpha, int slot_beta,
                                                     double *__restrict__ partials, int *__restrict__ flags, int lag,
                                                     int fold_base, int par, PushArgs P) {
    if (flags[0]) {
        if (PUSH && blockIdx.x == 0 && threadIdx.x == 0) push_post(P);
        return;
    }
    __shared__ double s_red[4];
    typedef double d2 __attribute__((ext_vector_type(2)));
    double alpha, beta;
    bool fold_exact = true;
    if (FOLD) {
        const double pq = slots[fold_base], qq = slots[fold_base + 1], rz0 = slots[fold_base + 2], rr0 = slots[fold_base + 3];
        const bool lead = blockIdx.x == 0 && threadIdx.x == 0;
        if (lead) slots[6] = rr0;                                  // the last measured true r.r, for the report
        int done = 0, status = 0;
        const double tol2 = slots[S_TOL2];
        fold_exact = slots[S1F_EXACT + (par ^ 1)] != 0.0;         // (pcg1_finish, on values every workgroup holds)
        if (!(rz0 == rz0) || !(pq == pq)) { done = 1; status = PGD_ERR_SINGULAR; }
        else if (fold_exact && rr0 <= tol2) done = 1;
        else {
            if (!fold_exact && rz0 * slots[S_DMIN] <= 1e4 * tol2) fold_exact = true;
            if (!(rz0 > 0.0)) done = 1;
        }
        if (lead) { slots[S1F_EXACT + par] = fold_exact ? 1.0 : 0.0; flags[3] = fold_exact ? 1 : 0; }
        if (done) {
            if (lead) { if (status) flags[2] = status; flags[0] = 1; if (PUSH) push_post(P); }
            return;                                                // uniform over the whole launch
        }
        alpha = rz0 / pq;
        double rnew = alpha * alpha * qq - rz0;
        if (!(rnew > 0.0)) rnew = 0.0;
        beta = rnew / rz0;
        if (lead) { slots[S1F_ALPHA + par] = alpha; slots[S1F_BETA + par] = beta; flags[1] += 1; }
    } else {
        alpha = slots[slot_alpha];
        beta = slots[slot_beta];
    }
    const bool exact = FOLD ? fold_exact : flags[3] != 0;
    const bool skip_x = lag == 1 && beta >= LAG_MIN_BETA;
    const bool two = lag == 2 && slots[S1_PEND] != 0.0;
    const double alpha_p = two ? slots[FOLD ? S1F_ALPHA + (par ^ 1) : S1_ALPHA_PREV] : 0.0;
    const double ibeta_p = two ? 1.0 / slots[FOLD ? S1F_BETA + (par ^ 1) : S1_BETA_PREV] : 0.0;
    double rz = 0.0, rr = 0.0;
    int pushed = 0;
    if ((lo & 1) == 0) {                                  // 16-byte accesses (row ranges of the sharded solve may start odd)
        const int64_t npair = (hi - lo) >> 1;
        for (int64_t k = (int64_t)blockIdx.x * TPB + threadIdx.x; k < npair; k += (int64_t)gridDim.x * TPB) {
            const int64_t i = lo + 2 * k;
            const d2 qi = NT ? __builtin_nontemporal_load(reinterpret_cast<const d2 *>(q + i)) : *reinterpret_cast<const d2 *>(q + i);
            d2 pi = *reinterpret_cast<d2 *>(p + i);
            d2 ri = NT ? __builtin_nontemporal_load(reinterpret_cast<d2 *>(r + i)) : *reinterpret_cast<d2 *>(r + i);
            if (!skip_x) {                                // uniform
                d2 xi = NT ? __builtin_nontemporal_load(reinterpret_cast<d2 *>(x + i)) : *reinterpret_cast<d2 *>(x + i);
                if (two) { xi.x = fma(alpha_p, (pi.x - ri.x) * ibeta_p, xi.x); xi.y = fma(alpha_p, (pi.y - ri.y) * ibeta_p, xi.y); }
                xi.x = fma(alpha, pi.x, xi.x); xi.y = fma(alpha, pi.y, xi.y);
                if (NT) __builtin_nontemporal_store(xi, reinterpret_cast<d2 *>(x + i)); else *reinterpret_cast<d2 *>(x + i) = xi;
            }
            ri.x = fma(-alpha, qi.x, ri.x); ri.y = fma(-alpha, qi.y, ri.y);
            pi.x = fma(beta, pi.x, ri.x); pi.y = fma(beta, pi.y, ri.y);
            if (NT) __builtin_nontemporal_store(ri, reinterpret_cast<d2 *>(r + i)); else *reinterpret_cast<d2 *>(r + i) = ri;
            *reinterpret_cast<d2 *>(p + i) = pi;
            if (PUSH) {                                   // (even plane sizes: a pair never straddles a range)
                if (i < P.lo_end) { push_store(P.dst_lo + (i - lo), pi.x); push_store(P.dst_lo + (i - lo) + 1, pi.y); pushed = 1; }
                if (i >= P.hi_begin) { push_store(P.dst_hi + (i - P.hi_begin), pi.x); push_store(P.dst_hi + (i - P.hi_begin) + 1, pi.y); pushed = 1; }
            }
            rz = fma(ri.x, ri.x, rz); rz = fma(ri.y, ri.y, rz);
            if (exact) {
                const d2 si = *reinterpret_cast<const d2 *>(s + i);
                const double tx = ri.x / si.x, ty = ri.y / si.y;
                rr = fma(tx, tx, rr); rr = fma(ty, ty, rr);
            }
        }
    }
    const int64_t tail0 = (lo & 1) == 0 ? lo + 2 * ((hi - lo) >> 1) : lo;      // what the pair loop left: one row, or all of them
    for (int64_t i = tail0 + (int64_t)blockIdx.x * TPB + threadIdx.x; i < hi; i += (int64_t)gridDim.x * TPB) {
        const double pi = p[i], r0 = r[i];
        if (!skip_x) {
            double xi = x[i];
            if (two) xi = fma(alpha_p, (pi - r0) * ibeta_p, xi);
            x[i] = fma(alpha, pi, xi);
        }
        const double ri = fma(-alpha, q[i], r0);
        r[i] = ri;
        p[i] = fma(beta, pi, ri);
        rz = fma(ri, ri, rz);
        if (exact) { const double t = ri / s[i]; rr = fma(t, t, rr); }
    }
    rz = block_sum(rz, s_red);
    rr = block_sum(rr, s_red);
    if (threadIdx.x == 0) { partials[2 * blockIdx.x] = rz; partials[2 * blockIdx.x + 1] = exact ? rr : rz; }
    if (lag == 1 && blockIdx.x == 0 && threadIdx.x == 0) slots[S1_PEND] = skip_x ? 1.0 : 0.0;
    if (PUSH) {
        __builtin_amdgcn_s_waitcnt(0);                    // this wave's pushed rows are acknowledged
        if (!__syncthreads_or(pushed)) return;            // a workgroup without boundary rows (uniform)
        if (threadIdx.x != 0) return;
        const unsigned long long t = __hip_atomic_fetch_add(P.ticket, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (t + 1 != P.nblocks) return;
        __hip_atomic_store(P.ticket, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        push_post(P);
        const long long t0 = wall_clock64();
        for (int which = 0; which < 2; ++which) {
            const unsigned long long *f = which ? P.wait_b : P.wait_a;
            if (!f) continue;
            while (__hip_atomic_load(f, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < P.seq) {
                if (wall_clock64() - t0 > P.ticks) { flags[2] = PGD_ERR_TIMEOUT; flags[0] = 1; return; }
                __builtin_amdgcn_s_sleep(4);
            }
        }
    }
}

// ---- small systems (up to 2^20 rows): the scalar step folded into the vector update, 2 launches per iteration.
// Where launches and not bytes set the pace (256^2 rows: three 3 us kernels and their gaps per iteration) EVERY workgroup of
// the update sums the product's (p.q, q.q) pairs and the previous update's (r~.r~, r.r) pairs itself - in one fixed order
// (256 strided lanes, wave sums, the four waves), so every workgroup holds the
// same alpha, beta and stop decision bit for bit - and workgroup 0 keeps the scalar bank and the flags for the host and the
// kernels that follow.  Nothing a workgroup reads is written by another workgroup of the same launch: the update's own partial
// sums, alpha / beta (the lagged x update reads the previous pair) and the "exact phase" bit alternate between two buffers with
// the parity of the iteration; a workgroup that starts late and finds the done flag already set by workgroup 0 returns, which
// is what its own test would have told it.  (Measured on larger systems too: at 128^3 a wash, at 256^3 the redundant sums cost
// more than the launch they save - those keep k_pcg1_scalars + k_pcg1_update.)
struct Pcg1Scalars { double alpha, beta; int exact, done, status; double pq, qq, rz, rr; };

__device__ __forceinline__ Pcg1Scalars pcg1_wg_scalars(const double *__restrict__ prod, int nprod, const double *__restrict__ vecp, int nvec,
                                                       const double *__restrict__ slots, int exact_cur, double *s_w /* >= 16 */) {
    // all four sums in one sweep (16-byte loads of the pairs, every load independent), one exchange through LDS: a fixed
    // order - lane t adds the pairs t, t + 256, ... - so every workgroup, and every run, gets the same bits
    typedef double d2 __attribute__((ext_vector_type(2)));
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    double a[4] = {0.0, 0.0, 0.0, 0.0};
    for (int i = t; i < nprod; i += 256) { const d2 v = *reinterpret_cast<const d2 *>(prod + 2 * (int64_t)i); a[0] += v.x; a[1] += v.y; }
    for (int i = t; i < nvec; i += 256) { const d2 v = *reinterpret_cast<const d2 *>(vecp + 2 * (int64_t)i); a[2] += v.x; a[3] += v.y; }
#pragma unroll
    for (int v = 0; v < 4; ++v) a[v] = wave_sum(a[v]);
    __syncthreads();
    if (lane == 0) {
#pragma unroll
        for (int v = 0; v < 4; ++v) s_w[4 * v + wv] = a[v];
    }
    __syncthreads();
    double out[4];
#pragma unroll
    for (int v = 0; v < 4; ++v) out[v] = (s_w[4 * v] + s_w[4 * v + 1]) + (s_w[4 * v + 2] + s_w[4 * v + 3]);
    __syncthreads();                                     // s_w is the caller's reduction scratch as well
    Pcg1Scalars S;
    S.pq = out[0]; S.qq = out[1]; S.rz = out[2]; S.rr = out[3];
    S.alpha = S.beta = 0.0; S.exact = exact_cur; S.done = 0; S.status = 0;
    // (pcg1_finish, on local values)
    const double tol2 = slots[S_TOL2];
    if (!(S.rz == S.rz) || !(S.pq == S.pq)) { S.done = 1; S.status = PGD_ERR_SINGULAR; return S; }
    if (exact_cur) { if (S.rr <= tol2) { S.done = 1; return S; } }
    else if (S.rz * slots[S_DMIN] <= 1e4 * tol2) S.exact = 1;
    if (!(S.rz > 0.0)) { S.done = 1; return S; }
    S.alpha = S.rz / S.pq;
    double rnew = S.alpha * S.alpha * S.qq - S.rz;
    if (!(rnew > 0.0)) rnew = 0.0;
    S.beta = rnew / S.rz;
    return S;
}

// par = parity of the iteration; vec_in / vec_out: the previous / this update's partial-sum pairs
__global__ __launch_bounds__(TPB) void k_pcg1_step(double *__restrict__ x, double *__restrict__ r, double *__restrict__ p,
                                                   const double *__restrict__ q, const double *__restrict__ s, int64_t n,
                                                   const double *__restrict__ prod, int nprod, const double *__restrict__ vec_in, int nvec,
                                                   double *__restrict__ vec_out, double *__restrict__ slots, int *__restrict__ flags,
                                                   int par, int lag) {
    // The exit on the done flag must be WORKGROUP-UNIFORM: workgroup 0 of this very launch sets the flag when the stop test fires,
    // and the waves of a workgroup that starts late could otherwise read different values - one leaves, the others go on into
    // pcg1_wg_scalars with that wave's LDS entries never written, and update x / r / p of the converged solve with garbage.
    __shared__ double s_red[16];
    __shared__ int s_done;
    if (threadIdx.x == 0) s_done = flags[0];
    __syncthreads();
    if (s_done) return;
    typedef double d2 __attribute__((ext_vector_type(2)));
    const Pcg1Scalars S = pcg1_wg_scalars(prod, nprod, vec_in, nvec, slots, slots[S1F_EXACT + (par ^ 1)] != 0.0, s_red);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        slots[S1_PQ] = S.pq; slots[S1_QQ] = S.qq; slots[S1_RZ] = S.rz; slots[S1_RR] = S.rr;
        slots[S1F_EXACT + par] = S.exact ? 1.0 : 0.0;
        flags[3] = S.exact;
        if (S.done) { if (S.status) flags[2] = S.status; flags[0] = 1; }
        else { slots[S1F_ALPHA + par] = S.alpha; slots[S1F_BETA + par] = S.beta; flags[1] += 1; }
    }
    if (S.done) return;                                  // uniform over the whole launch
    const double alpha = S.alpha, beta = S.beta;
    const bool exact = S.exact != 0;
    const bool skip_x = lag == 1 && beta >= LAG_MIN_BETA;
    const bool two = lag == 2 && slots[S1_PEND] != 0.0;
    const double alpha_p = two ? slots[S1F_ALPHA + (par ^ 1)] : 0.0, ibeta_p = two ? 1.0 / slots[S1F_BETA + (par ^ 1)] : 0.0;
    double rz = 0.0, rr = 0.0;
    const int64_t npair = n >> 1;
    for (int64_t k = (int64_t)blockIdx.x * TPB + threadIdx.x; k < npair; k += (int64_t)gridDim.x * TPB) {
        const int64_t i = 2 * k;
        const d2 qi = *reinterpret_cast<const d2 *>(q + i);
        d2 pi = *reinterpret_cast<d2 *>(p + i), ri = *reinterpret_cast<d2 *>(r + i);
        if (!skip_x) {                                    // uniform
            d2 xi = *reinterpret_cast<d2 *>(x + i);
            if (two) { xi.x = fma(alpha_p, (pi.x - ri.x) * ibeta_p, xi.x); xi.y = fma(alpha_p, (pi.y - ri.y) * ibeta_p, xi.y); }
            xi.x = fma(alpha, pi.x, xi.x); xi.y = fma(alpha, pi.y, xi.y);
            *reinterpret_cast<d2 *>(x + i) = xi;
        }
        ri.x = fma(-alpha, qi.x, ri.x); ri.y = fma(-alpha, qi.y, ri.y);
        pi.x = fma(beta, pi.x, ri.x); pi.y = fma(beta, pi.y, ri.y);
        *reinterpret_cast<d2 *>(r + i) = ri;
        *reinterpret_cast<d2 *>(p + i) = pi;
        rz = fma(ri.x, ri.x, rz); rz = fma(ri.y, ri.y, rz);
        if (exact) {
            const d2 si = *reinterpret_cast<const d2 *>(s + i);
            const double tx = ri.x / si.x, ty = ri.y / si.y;
            rr = fma(tx, tx, rr); rr = fma(ty, ty, rr);
        }
    }
    for (int64_t i = 2 * npair + (int64_t)blockIdx.x * TPB + threadIdx.x; i < n; i += (int64_t)gridDim.x * TPB) {
        const double pi = p[i], r0 = r[i];
        if (!skip_x) {
            double xi = x[i];
            if (two) xi = fma(alpha_p, (pi - r0) * ibeta_p, xi);
            x[i] = fma(alpha, pi, xi);
        }
        const double ri = fma(-alpha, q[i], r0);
        r[i] = ri;
        p[i] = fma(beta, pi, ri);
        rz = fma(ri, ri, rz);
        if (exact) { const double t = ri / s[i]; rr = fma(t, t, rr); }
    }
    rz = block_sum(rz, s_red);
    rr = block_sum(rr, s_red);
    if (threadIdx.x == 0) { vec_out[2 * blockIdx.x] = rz; vec_out[2 * blockIdx.x + 1] = exact ? rr : rz; }
    if (lag == 1 && blockIdx.x == 0 && threadIdx.x == 0) slots[S1_PEND] = skip_x ? 1.0 : 0.0;
}

// before the first iteration: the initial residual's (r~.r~, true r.r) as the one non-zero pair of the vector partials
__global__ void k_pcg1_seed(double *__restrict__ partials, int npairs, const double *__restrict__ slots, int slot_rz, int slot_rr) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < npairs; i += gridDim.x * blockDim.x) {
        partials[2 * i] = i == 0 ? slots[slot_rz] : 0.0;
        partials[2 * i + 1] = i == 0 ? slots[slot_rr] : 0.0;
    }
}

// x <- s x (back to the unscaled unknown); partial sum r^2 / s^2 (the true r.r, for the report)
// p != nullptr: the x update of the last iteration is still pending (k_pcg_px_s was a no-op once the done flag was set):
// x <- s (x + alpha p) with alpha = S[slot_rz] / S[slot_pq] of that iteration
// lagged != 0 (with p): the solve ended after a k_pcg1_update that left x alone (slots[S1_PEND] says whether it did): the
// outstanding term is alpha p' with p' = (p - r) / beta of that iteration, whose alpha and beta are still in their slots
__global__ __launch_bounds__(TPB) void k_scale_out(double *__restrict__ x, const double *__restrict__ r, const double *__restrict__ s,
                                                   int64_t n, double *__restrict__ partials, const double *__restrict__ p,
                                                   const double *__restrict__ slots, int slot_rz, int slot_pq, int lagged,
                                                   int slot_alpha, int slot_beta) {
    __shared__ double s_red[4];
    double rr = 0.0;
    const bool lag_term = lagged && p && slots[S1_PEND] != 0.0;
    const double alpha = lagged ? (lag_term ? slots[slot_alpha] : 0.0) : p ? slots[slot_rz] / slots[slot_pq] : 0.0;
    const double ibeta = lag_term ? 1.0 / slots[slot_beta] : 0.0;
    for (int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x; i < n; i += (int64_t)gridDim.x * TPB) {
        const double si = s[i], ri = r[i], t = ri / si;
        double xi = x[i];
        if (lag_term) xi = fma(alpha, (p[i] - ri) * ibeta, xi);
        else if (p && !lagged) xi = fma(alpha, p[i], xi);
        x[i] = xi * si;
        rr = fma(t, t, rr);
    }
    rr = block_sum(rr, s_red);
    if (threadIdx.x == 0) partials[blockIdx.x] = rr;
}

// ---- the scaled recurrence with the final reduction passes folded into their consumers: every workgroup of the
// x / r update sums the product's partials itself (<= 8192 of them, the same fixed order in every workgroup: one
// value, bit for bit), every workgroup of the p update sums the x / r update's partials, and its workgroup 0 also
// keeps the scalar bank and the convergence test.  3 dependent launches per iteration instead of 5.
__device__ __forceinline__ double block_total(const double *__restrict__ part, int n, int stride, int off, double *s_red) {
    double a = 0.0;
    for (int i = threadIdx.x; i < n; i += TPB) a += part[(int64_t)i * stride + off];
    a = block_sum(a, s_red);
    __shared__ double s_bc;
    __syncthreads();
    if (threadIdx.x == 0) s_bc = a;
    __syncthreads();
    const double v = s_bc;
    __syncthreads();
    return v;
}

__global__ __launch_bounds__(TPB) void k_pcg_xr_s2(double *__restrict__ x, double *__restrict__ r, const double *__restrict__ p,
                                                   const double *__restrict__ q, const double *__restrict__ s, int64_t n,
                                                   const double *__restrict__ slots, int slot_rz,
                                                   const double *__restrict__ pq_part, int npq,
                                                   double *__restrict__ partials, const int *__restrict__ flags) {
    if (flags[0]) return;
    __shared__ double s_red[4];
    typedef double d2 __attribute__((ext_vector_type(2)));
    const double pq = block_total(pq_part, npq, 1, 0, s_red);
    const double alpha = slots[slot_rz] / pq;
    const bool exact = flags[3] != 0;
    double rz = 0.0, rr = 0.0;
    const int64_t npair = n >> 1;
    for (int64_t k = (int64_t)blockIdx.x * TPB + threadIdx.x; k < npair; k += (int64_t)gridDim.x * TPB) {
        const int64_t i = 2 * k;
        const d2 pi = *reinterpret_cast<const d2 *>(p + i), qi = *reinterpret_cast<const d2 *>(q + i);
        d2 xi = *reinterpret_cast<d2 *>(x + i), ri = *reinterpret_cast<d2 *>(r + i);
        xi.x = fma(alpha, pi.x, xi.x); xi.y = fma(alpha, pi.y, xi.y);
        ri.x = fma(-alpha, qi.x, ri.x); ri.y = fma(-alpha, qi.y, ri.y);
        *reinterpret_cast<d2 *>(x + i) = xi;
        *reinterpret_cast<d2 *>(r + i) = ri;
        rz = fma(ri.x, ri.x, rz); rz = fma(ri.y, ri.y, rz);
        if (exact) {
            const d2 si = *reinterpret_cast<const d2 *>(s + i);
            const double tx = ri.x / si.x, ty = ri.y / si.y;
            rr = fma(tx, tx, rr); rr = fma(ty, ty, rr);
        }
    }
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
        const int64_t i = n - 1;
        x[i] = fma(alpha, p[i], x[i]);
        const double ri = fma(-alpha, q[i], r[i]);
        r[i] = ri;
        rz = fma(ri, ri, rz);
        if (exact) { const double t = ri / s[i]; rr = fma(t, t, rr); }
    }
    rz = block_sum(rz, s_red);
    rr = block_sum(rr, s_red);
    if (threadIdx.x == 0) { partials[2 * blockIdx.x] = rz; partials[2 * blockIdx.x + 1] = exact ? rr : rz; }
}

// p = r + beta p with beta = (r.r) / S[rz_old]; workgroup 0 stores (r.r, true r.r) in S[out], S[out+1], counts the
// iteration and runs the stop test (the logic of k_reduce_partials' check_mode 2)
__global__ __launch_bounds__(TPB) void k_pcg_p_s2(double *__restrict__ p, const double *__restrict__ r, int64_t n,
                                                  const double *__restrict__ part, int npart, double *__restrict__ slots,
                                                  int out, int rz_old, int slot_tol2, int *__restrict__ flags) {
    if (flags[0]) return;
    __shared__ double s_red[4];
    typedef double d2 __attribute__((ext_vector_type(2)));
    const double rz = block_total(part, npart, 2, 0, s_red);
    const double rr = block_total(part, npart, 2, 1, s_red);
    const double beta = rz / slots[rz_old];
    const int64_t npair = n >> 1;
    for (int64_t k = (int64_t)blockIdx.x * TPB + threadIdx.x; k < npair; k += (int64_t)gridDim.x * TPB) {
        const int64_t i = 2 * k;
        const d2 ri = *reinterpret_cast<const d2 *>(r + i);
        d2 pi = *reinterpret_cast<d2 *>(p + i);
        pi.x = fma(beta, pi.x, ri.x); pi.y = fma(beta, pi.y, ri.y);
        *reinterpret_cast<d2 *>(p + i) = pi;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        if (n & 1) p[n - 1] = fma(beta, p[n - 1], r[n - 1]);
        slots[out] = rz; slots[out + 1] = rr;
        const double tol2 = slots[slot_tol2];
        flags[1] += 1;
        if (!(rz == rz) || !(rr == rr)) { flags[0] = 1; flags[2] = PGD_ERR_SINGULAR; }
        else if (flags[3]) { if (rr <= tol2) flags[0] = 1; }
        else if (rz * slots[S_DMIN] <= 1e4 * tol2) flags[3] = 1;
    }
}

// CSR values of an operator whose combine was deferred (pgd_op_combine with the diagonal form in place): A = sum_t c_t A_t,
// Dirichlet columns zeroed, identity rows - from the recorded atoms, which must still be the objects they were.
int ensure_vals(Ctx *c, const Mesh *m, Csr *o) {
    if (!o->vals_pending) return o->vals ? PGD_OK : fail(c, PGD_ERR_INVALID, "operator without values (its pgd_op_combine failed?)");
    const int n = (int)o->rec_atoms.size();
    if (!o->vals) {
        void *p;
        const size_t cnt = (size_t)(m->nnz > 0 ? m->nnz : 1);
        o->vals_bytes = cnt * sizeof(double);
        PGD_TRY(dev_alloc(c, &p, o->vals_bytes));
        o->vals = (double *)p;
        // k_combine writes every entry; the padding behind them (16-byte loads of the CSR kernels read past the end) is zero
        const size_t used = (size_t)(m->nnz > 0 ? m->nnz : 0);
        PGD_HIP(c, hipMemsetAsync(o->vals + used, 0, (cnt - used) * sizeof(double) + PAD_BYTES, c->stream));
    }
    std::vector<const double *> in((size_t)n);
    for (int t = 0; t < n; ++t) {
        Csr *a = get_csr(c, o->rec_atoms[(size_t)t]);
        if (!a || a->serial != o->rec_serials[(size_t)t] || a->version != o->rec_versions[(size_t)t])
            return fail(c, PGD_ERR_INVALID, "operator: its CSR values were left to the first reader and atom %d has been freed or rewritten since", t);
        PGD_TRY(ensure_vals(c, m, a));             // an operator as an input of another one
        in[(size_t)t] = a->vals;
    }
    const uint8_t *mask = nullptr;
    if (o->rec_nbc > 0) {
        PGD_TRY(ensure_mask(c, m->nv));
        PGD_HIP(c, hipMemsetAsync(c->mask, 0, (size_t)m->nv, c->stream));
        k_mask_set<<<grid_for(o->rec_nbc), TPB, 0, c->stream>>>(c->mask, o->rec_bc, o->rec_nbc);
        mask = c->mask;
    }
    const int g = grid_for(m->nnz, TPB, 4 * MAX_VEC_BLOCKS);
    for (int t = 0, pass = 0; t < n; ++pass) {
        // up to MAXT inputs per launch; later launches accumulate onto the running output
        CombineArgs A;
        int cnt = 0;
        if (pass > 0) { A.in[0] = o->vals; A.coef[0] = 1.0; cnt = 1; }
        while (t < n && cnt < MAXT) { A.in[cnt] = in[(size_t)t]; A.coef[cnt] = o->rec_coefs[(size_t)t]; ++cnt; ++t; }
        for (int k = cnt; k < MAXT; ++k) { A.in[k] = in[0]; A.coef[k] = 0.0; }
        A.n = cnt;
        k_combine<<<g, TPB, 0, c->stream>>>(A, o->vals, m->cols, (t >= n) ? mask : nullptr, m->nnz);
    }
    if (o->rec_nbc > 0) k_dirichlet_rows<<<grid_for(o->rec_nbc), TPB, 0, c->stream>>>(o->rec_bc, o->rec_nbc, m->row_ptr, m->cols, o->vals);
    PGD_LAUNCH_CHECK(c);
    o->vals_pending = false;
    return PGD_OK;
}

int csr_diag_inv(Ctx *c, const Mesh *m, Csr *a) {
    if (a->dinv_valid) return PGD_OK;
    if (!a->dinv) {
        void *p;
        a->dinv_bytes = (size_t)m->nv * sizeof(double);
        PGD_TRY(dev_alloc(c, &p, a->dinv_bytes));
        a->dinv = (double *)p;
    }
    PGD_TRY(ensure_vals(c, m, a));
    k_diag_inv<<<grid_for(m->nv), TPB, 0, c->stream>>>(m->row_ptr, m->cols, a->vals, a->dinv, m->nv);
    PGD_LAUNCH_CHECK(c);
    a->dinv_valid = true;
    return PGD_OK;
}

// shared by pgd_pcg_solve and the *_slot entry points
static int pcg_init(Ctx *c, const double *b, const double *q, const double *dinv, double *r, double *z, double *p,
                    int64_t lo, int64_t hi, int slot) {
    const int g = grid_for(hi - lo);
    PGD_TRY(ensure_partials(c, 4 * (int64_t)MAX_VEC_BLOCKS));
    k_pcg_init<<<g, TPB, 0, c->stream>>>(b, q, dinv, r, z, p, lo, hi, c->partials);
    PGD_LAUNCH_CHECK(c);
    return reduce_partials(c, c->partials, g, 3, slot, -1, 0, 0);
}

// launchers of the scaled sharded recurrence (pgd_comm.hip)
int cg_init_s(Ctx *c, const double *b, const double *q, const double *sc, double *r, double *p, double *s, int64_t lo,
              int64_t hi, int base) {
    if (hi == lo) return PGD_OK;
    const int g = grid_for(hi - lo);
    PGD_TRY(ensure_partials(c, 4 * (int64_t)MAX_VEC_BLOCKS));
    k_cg_init_s<<<g, TPB, 0, c->stream>>>(b, q, sc, r, p, s, lo, hi, c->partials);
    PGD_LAUNCH_CHECK(c);
    PGD_TRY(reduce_partials(c, c->partials, g, 3, 40, -1, 0, 0));
    PGD_HIP(c, hipMemcpyAsync(c->slots + base, c->slots + 40, 2 * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
    PGD_HIP(c, hipMemcpyAsync(c->slots + base + 8, c->slots + 42, sizeof(double), hipMemcpyDeviceToDevice, c->stream));
    return PGD_OK;
}

int cg_update_s(Ctx *c, double *x, double *r, const double *w, double *p, double *s, const double *sc, int64_t lo, int64_t hi,
                int base) {
    if (hi == lo) return PGD_OK;
    const int g = grid_for(hi - lo);
    PGD_TRY(ensure_partials(c, 4 * (int64_t)MAX_VEC_BLOCKS));
    k_cg_update_s<<<g, TPB, 0, c->stream>>>(x, r, w, p, s, sc, lo, hi, c->slots, base, c->partials, c->flags);
    PGD_LAUNCH_CHECK(c);
    return reduce_partials(c, c->partials, g, 2, base, 0, 0, 0);
}

// one iteration's vector step of the folded form; its partials stay in work[6] until reduce_two_slots
int cg_update_s2(Ctx *c, double *x, double *r, const double *w, double *p, double *s, const double *sc, int64_t lo, int64_t hi,
                 int base, int parity, int *nblocks) {
    *nblocks = 0;
    if (hi == lo) return PGD_OK;
    const int g = grid_for(hi - lo);
    PGD_TRY(ensure_work(c, 6, 2 * (int64_t)MAX_VEC_BLOCKS));
    k_cg_update_s2<<<g, TPB, 0, c->stream>>>(x, r, w, p, s, sc, lo, hi, c->slots, base, parity, c->work[6], c->flags);
    PGD_LAUNCH_CHECK(c);
    *nblocks = g;
    return PGD_OK;
}

int reduce_two_slots(Ctx *c, int na, int nb, int base) {
    k_reduce_two<<<1, 1024, 0, c->stream>>>(c->work[6], na, c->partials, nb, c->slots, base, c->flags);
    PGD_LAUNCH_CHECK(c);
    return PGD_OK;
}

int vec_sqrt(Ctx *c, double *v, int64_t n) {
    k_vec_sqrt<<<grid_for(n), TPB, 0, c->stream>>>(v, n);
    PGD_LAUNCH_CHECK(c);
    return PGD_OK;
}

int vec_div_mul(Ctx *c, double *x, const double *sc, int64_t n, int mul) {
    k_vec_div_mul<<<grid_for(n), TPB, 0, c->stream>>>(x, sc, n, mul);
    PGD_LAUNCH_CHECK(c);
    return PGD_OK;
}

// launchers of the single-sync recurrence for the row-sharded solve (pgd_comm.hip)
int pcg1_seed(Ctx *c, int npairs, int slot_rz, int slot_rr) {
    PGD_TRY(ensure_work(c, 6, 2 * (int64_t)MAX_VEC_BLOCKS));
    k_pcg1_seed<<<8, TPB, 0, c->stream>>>(c->work[6], npairs, c->slots, slot_rz, slot_rr);
    PGD_LAUNCH_CHECK(c);
    return PGD_OK;
}

int pcg1_tol(Ctx *c, int base, double rtol, double atol, int slot_p8) {
    k_pcg1_tol<<<1, 64, 0, c->stream>>>(c->slots, c->flags, base, rtol, atol, slot_p8);
    PGD_LAUNCH_CHECK(c);
    return PGD_OK;
}

// slots[slot] <- sum over [lo, hi) of s_i^16 (r == nullptr) or of r_i^2 / s_i^2 (k_pcg1_aux); 0 for an empty range
int pcg1_aux(Ctx *c, const double *s, const double *r, int64_t lo, int64_t hi, int slot) {
    if (hi == lo) {
        PGD_HIP(c, hipMemsetAsync(c->slots + slot, 0, sizeof(double), c->stream));
        return PGD_OK;
    }
    const int g = grid_for(hi - lo);
    PGD_TRY(ensure_partials(c, 4 * (int64_t)MAX_VEC_BLOCKS));
    k_pcg1_aux<<<g, TPB, 0, c->stream>>>(s, r, lo, hi, c->partials);
    PGD_LAUNCH_CHECK(c);
    return reduce_partials(c, c->partials, g, 1, slot, -1, 0, 0);
}

int pcg1_sums(Ctx *c, int nprod, int nvec, int base) {
    const double *prod = c->partials;
    if (nprod > 8192) {
        const int nb = (nprod + 1023) / 1024;
        PGD_TRY(ensure_work(c, 5, (int64_t)nb * 2 > 4096 ? (int64_t)nb * 2 : 4096));
        PGD_TRY(k_reduce_stage1_pub(c, c->partials, nprod, 2, c->work[5]));
        prod = c->work[5];
        nprod = nb;
    }
    k_pcg1_sums<<<1, 1024, 0, c->stream>>>(prod, nprod, c->work[6], nvec, c->slots, base, c->flags);
    PGD_LAUNCH_CHECK(c);
    return PGD_OK;
}

int pcg1_finish_slots(Ctx *c, int base) {
    k_pcg1_finish<<<1, 64, 0, c->stream>>>(c->slots, c->flags, base);
    PGD_LAUNCH_CHECK(c);
    return PGD_OK;
}

// the workgroups of a k_pcg1_update launch over [lo, hi) that hold a row of [lo, lo_end) or [hi_begin, hi): the kernel's own loop
unsigned int pcg1_update_push_blocks(int g, int64_t lo, int64_t hi, int64_t lo_end, int64_t hi_begin) {
    const int64_t npair = (hi - lo) >> 1, a1 = (lo_end - lo) >> 1, b0 = (hi_begin - lo) >> 1;
    unsigned int count = 0;
    for (int b = 0; b < g; ++b) {
        bool any = false;
        for (int64_t k0 = (int64_t)b * TPB; k0 < npair && !any; k0 += (int64_t)g * TPB) {
            const int64_t k1 = std::min<int64_t>(k0 + TPB, npair);
            any = k0 < a1 || k1 > b0;
        }
        count += any ? 1u : 0u;
    }
    return count;
}

int pcg1_update(Ctx *c, double *x, double *r, double *p, const double *q, const double *sc, int64_t lo, int64_t hi, int base,
                int *nblocks, int lag, int fold_par, const PushArgs *push) {       // fold_par >= 0: the scalar step in every workgroup (parity of the iteration)
    *nblocks = 0;
    if (hi == lo) return push ? fail(c, PGD_ERR_INVALID, "pcg1_update: a push from an empty slab") : PGD_OK;
    const int g = grid_for((hi - lo + 1) / 2);
    if (push) {
        // the caller has checked: fold, lo even, (hi - lo), (lo_end - lo), (hi - hi_begin) even
        if (fold_par < 0 || (lo & 1) || ((hi - lo) & 1) || ((push->lo_end - lo) & 1) || ((hi - push->hi_begin) & 1))
            return fail(c, PGD_ERR_INVALID, "pcg1_update: this launch cannot carry the direct halo");
    }
    PGD_TRY(ensure_work(c, 6, 2 * (int64_t)MAX_VEC_BLOCKS));
    const bool timed_u = c->prof && ((c->prof_upd_seen++ % 3) == 0);      // launch timing, as in pgd_pcg_solve
    if (timed_u) {
        if (c->ev_used + 2 > c->ev.size()) prof_flush(c);
        c->ev_rec[c->ev_used / 2] = Ctx::ProfRec{1, c->prof_iter, 0.0, 0.0, 0.0};
        PGD_HIP(c, hipEventRecord(c->ev[c->ev_used], c->stream));
    }
    if (push) {
        PushArgs P = *push;
        if (!P.nblocks) P.nblocks = pcg1_update_push_blocks(g, lo, hi, P.lo_end, P.hi_begin);
        if (c->pcg_stream_hints) k_pcg1_update<true, true, true><<<g, TPB, 0, c->stream>>>(x, r, p, q, sc, lo, hi, c->slots, 0, 0, c->work[6], c->flags, lag, base, fold_par, P);
        else k_pcg1_update<false, true, true><<<g, TPB, 0, c->stream>>>(x, r, p, q, sc, lo, hi, c->slots, 0, 0, c->work[6], c->flags, lag, base, fold_par, P);
    } else if (fold_par >= 0) {
        if (c->pcg_stream_hints) k_pcg1_update<true, true><<<g, TPB, 0, c->stream>>>(x, r, p, q, sc, lo, hi, c->slots, 0, 0, c->work[6], c->flags, lag, base, fold_par, PushArgs());
        else k_pcg1_update<false, true><<<g, TPB, 0, c->stream>>>(x, r, p, q, sc, lo, hi, c->slots, 0, 0, c->work[6], c->flags, lag, base, fold_par, PushArgs());
    } else if (c->pcg_stream_hints) k_pcg1_update<true, false><<<g, TPB, 0, c->stream>>>(x, r, p, q, sc, lo, hi, c->slots, base + 5, base + 6, c->work[6], c->flags, lag, 0, 0, PushArgs());
    else k_pcg1_update<false, false><<<g, TPB, 0, c->stream>>>(x, r, p, q, sc, lo, hi, c->slots, base + 5, base + 6, c->work[6], c->flags, lag, 0, 0, PushArgs());
    if (timed_u) {
        PGD_HIP(c, hipEventRecord(c->ev[c->ev_used + 1], c->stream));
        // (outside the exact phase - all but the last few dozen iterations - the kernel does not read s; the host cannot see the flag)
        c->ev_rec[c->ev_used / 2].bytes = ((lag == 1 ? 40.0 : 56.0) + (c->pcg_exact_phase ? 0.0 : 8.0)) * (double)(hi - lo);
        c->ev_used += 2;
    }
    PGD_LAUNCH_CHECK(c);
    *nblocks = g;
    return PGD_OK;
}

// the sharded solve's way out when its last update left a term of x outstanding (lagged x update): x += alpha (p - r) / beta on
// the owned rows, alpha and beta of that iteration still in their slots; a no-op when slots[S1_PEND] is 0
__global__ __launch_bounds__(TPB) void k_pcg1_flush_x(double *__restrict__ x, const double *__restrict__ p, const double *__restrict__ r,
                                                      int64_t lo, int64_t hi, const double *__restrict__ slots, int slot_alpha,
                                                      int slot_beta) {
    if (slots[S1_PEND] == 0.0) return;
    const double alpha = slots[slot_alpha], ibeta = 1.0 / slots[slot_beta];
    for (int64_t i = lo + (int64_t)blockIdx.x * TPB + threadIdx.x; i < hi; i += (int64_t)gridDim.x * TPB)
        x[i] = fma(alpha, (p[i] - r[i]) * ibeta, x[i]);
}

int pcg1_flush_x(Ctx *c, double *x, const double *p, const double *r, int64_t lo, int64_t hi, int base, int fold_par) {
    if (hi == lo) return PGD_OK;
    const int sa = fold_par >= 0 ? S1F_ALPHA + fold_par : base + 5, sb = fold_par >= 0 ? S1F_BETA + fold_par : base + 6;
    k_pcg1_flush_x<<<grid_for(hi - lo), TPB, 0, c->stream>>>(x, p, r, lo, hi, c->slots, sa, sb);
    PGD_LAUNCH_CHECK(c);
    return PGD_OK;
}

static int pcg_xr(Ctx *c, double *x, double *r, const double *p, const double *q, const double *dinv, double *z,
                  int64_t lo, int64_t hi, int slot_rz, int slot_pq, int slot_out, int check_mode, int slot_tol2) {
    const int g = grid_for(hi - lo);
    PGD_TRY(ensure_partials(c, 4 * (int64_t)MAX_VEC_BLOCKS));
    if ((lo & 1) == 0) {
        const int g2 = grid_for((hi - lo + 1) / 2);
        k_pcg_xr<true><<<g2, TPB, 0, c->stream>>>(x, r, p, q, dinv, z, lo, hi, c->slots, slot_rz, slot_pq, c->partials, c->flags);
        PGD_LAUNCH_CHECK(c);
        return reduce_partials(c, c->partials, g2, 2, slot_out, check_mode, slot_out + 1, slot_tol2);
    }
    k_pcg_xr<false><<<g, TPB, 0, c->stream>>>(x, r, p, q, dinv, z, lo, hi, c->slots, slot_rz, slot_pq, c->partials, c->flags);
    PGD_LAUNCH_CHECK(c);
    return reduce_partials(c, c->partials, g, 2, slot_out, check_mode, slot_out + 1, slot_tol2);
}

}  // namespace pgd

using namespace pgd;

extern "C" {

int pgd_op_combine(pgd_handle h, pgd_handle mh, const pgd_handle *atoms, const double *coefs, int n,
                   const int32_t *bc_dofs, int64_t nbc, pgd_handle *op) {
    PGD_CTX(c, h);
    Mesh *m = get_mesh(c, mh);
    if (!m || !atoms || !coefs || !op || n < 1 || nbc < 0 || (nbc > 0 && !bc_dofs))
        return fail(c, PGD_ERR_INVALID, "op_combine: invalid arguments");
    std::vector<Csr *> atom_objs((size_t)n);
    for (int t = 0; t < n; ++t) {
        Csr *a = get_csr(c, atoms[t]);
        if (!a || a->mesh != mh) return fail(c, PGD_ERR_INVALID, "op_combine: atom %d is not on this mesh", t);
        if (*op && atoms[t] == *op) return fail(c, PGD_ERR_INVALID, "op_combine: output aliases an input");
        atom_objs[t] = a;
    }
    // the same Dirichlet list as the last operator on this mesh (checked word by word): it was range-checked then and is on the device
    const bool bc_known = nbc > 0 && m->bc_dev && m->bc_host.size() == (size_t)nbc &&
                          std::memcmp(m->bc_host.data(), bc_dofs, (size_t)nbc * sizeof(int32_t)) == 0;
    if (!bc_known)
        for (int64_t i = 0; i < nbc; ++i)
            if (bc_dofs[i] < 0 || bc_dofs[i] >= m->nv) return fail(c, PGD_ERR_INVALID, "op_combine: bc dof out of range");
    Csr *o = nullptr;
    if (*op) {
        o = get_csr(c, *op);
        if (!o || o->mesh != mh) return fail(c, PGD_ERR_INVALID, "op_combine: *op is not an operator on this mesh");
    } else {
        std::unique_ptr<Csr> a(new Csr);       // (its CSR array - 2 GB at 256^3 - is allocated by ensure_vals, i.e. usually never)
        a->kind = Obj::CSR;
        a->mesh = mh;
        o = a.get();
        *op = put_obj(c, a.release());
    }
    o->dinv_valid = false;
    o->uvals_valid = false;      // new values: the symmetric copy is rebuilt by the next solve
    o->vals_pending = false;
    o->version += 1;
    // the recipe: what ensure_vals needs to form the CSR values later (and what it forms them from right away otherwise)
    o->rec_atoms.assign(atoms, atoms + n);
    o->rec_coefs.assign(coefs, coefs + n);
    o->rec_serials.resize((size_t)n);
    o->rec_versions.resize((size_t)n);
    for (int t = 0; t < n; ++t) { o->rec_serials[(size_t)t] = atom_objs[t]->serial; o->rec_versions[(size_t)t] = atom_objs[t]->version; }
    if (o->rec_bc && o->rec_bc_bytes < (size_t)nbc * sizeof(int)) { dev_release(c, o->rec_bc, o->rec_bc_bytes); o->rec_bc = nullptr; }
    o->rec_nbc = nbc;
    {   // signature of the Dirichlet set + the atoms' identities: only a HINT for the mesh's classification cache (a sample of the
        // list: what the cache hands back is verified row by row)
        uint64_t hsig = 0x9e3779b97f4a7c15ull ^ (uint64_t)nbc;
        auto mix = [&](uint64_t v) { hsig = (hsig ^ v) * 0xff51afd7ed558ccdull; hsig ^= hsig >> 32; };
        const int64_t step = nbc > 4096 ? nbc / 4096 : 1;
        for (int64_t i = 0; i < nbc; i += step) mix((uint64_t)(uint32_t)bc_dofs[i]);
        if (nbc > 0) mix((uint64_t)(uint32_t)bc_dofs[nbc - 1]);
        for (int t = 0; t < n; ++t) mix(atom_objs[t]->serial);
        o->bc_sig = hsig ? hsig : 1;
    }
    const uint8_t *mask = nullptr;
    if (nbc > 0) {
        if (!o->rec_bc) {
            void *p;
            o->rec_bc_bytes = (size_t)nbc * sizeof(int);
            PGD_TRY(dev_alloc(c, &p, o->rec_bc_bytes));
            o->rec_bc = (int *)p;
        }
        PGD_TRY(ensure_mask(c, m->nv));
        PGD_HIP(c, hipMemsetAsync(c->mask, 0, (size_t)m->nv, c->stream));
        if (!bc_known) {
            m->bc_host.clear();
            if (m->bc_dev && m->bc_dev_bytes < (size_t)nbc * sizeof(int)) { (void)hipFree(m->bc_dev); m->bc_dev = nullptr; }
            if (!m->bc_dev) {
                void *p = nullptr;
                m->bc_dev_bytes = (size_t)nbc * sizeof(int);
                if (hipMalloc(&p, m->bc_dev_bytes + PAD_BYTES) != hipSuccess) { (void)hipGetLastError(); m->bc_dev_bytes = 0; }
                m->bc_dev = (int *)p;
            }
            PGD_HIP(c, hipMemcpyAsync(o->rec_bc, bc_dofs, (size_t)nbc * sizeof(int), hipMemcpyHostToDevice, c->stream));
            if (m->bc_dev) {
                PGD_HIP(c, hipMemcpyAsync(m->bc_dev, o->rec_bc, (size_t)nbc * sizeof(int), hipMemcpyDeviceToDevice, c->stream));
                m->bc_host.assign(bc_dofs, bc_dofs + nbc);
            }
        } else {
            PGD_HIP(c, hipMemcpyAsync(o->rec_bc, m->bc_dev, (size_t)nbc * sizeof(int), hipMemcpyDeviceToDevice, c->stream));
        }
        k_mask_set<<<grid_for(nbc), TPB, 0, c->stream>>>(c->mask, o->rec_bc, nbc);
        mask = c->mask;
    }
    o->vals_pending = true;      // from here on the recipe is complete: whatever fails below, a later reader can still form the values
    // structured grids: the diagonal form of the same operator from the atoms' diagonal forms (no per-solve conversion)
    PGD_TRY(combine_dia(c, m, o, atom_objs.data(), coefs, n, mask));
    // ... and where that form exists the solve, its start and its products read nothing else: the CSR values (8 nnz (T + 1)
    // bytes of streaming, 1.5 ms at 256^3) are formed by the first reader that asks for them, usually nobody
    if (!(c->lazy_csr && o->uvals_valid)) PGD_TRY(ensure_vals(c, m, o));
    if (nbc > 0 && !bc_known) PGD_HIP(c, hipStreamSynchronize(c->stream));   // bc_dofs is caller-owned
    return PGD_OK;
}

int pgd_op_diag_inv(pgd_handle h, pgd_handle oh, pgd_handle dh) {
    PGD_CTX(c, h);
    Csr *o = get_csr(c, oh);
    Mesh *m = o ? get_mesh(c, o->mesh) : nullptr;
    Vec *d = get_vec(c, dh);
    if (!o || !m || !d || d->n != m->nv) return fail(c, PGD_ERR_INVALID, "op_diag_inv: invalid handles");
    if (o->dinv_valid && o->dinv) {      // combine_dia formed it with the diagonal form (the same quotient): no pass over the CSR values
        PGD_HIP(c, hipMemcpyAsync(d->d, o->dinv, (size_t)m->nv * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
        return PGD_OK;
    }
    PGD_TRY(ensure_vals(c, m, o));
    k_diag_inv<<<grid_for(m->nv), TPB, 0, c->stream>>>(m->row_ptr, m->cols, o->vals, d->d, m->nv);
    PGD_LAUNCH_CHECK(c);
    return PGD_OK;
}

int pgd_pcg_solve(pgd_handle h, pgd_handle oh, pgd_handle bh, pgd_handle xh, double rtol, double atol, int maxit,
                  int *iters, double *relres) {
    PGD_CTX(c, h);
    Csr *o = get_csr(c, oh);
    Mesh *m = o ? get_mesh(c, o->mesh) : nullptr;
    Vec *b = get_vec(c, bh), *x = get_vec(c, xh);
    if (!o || !m || !b || !x || b->n != m->nv || x->n != m->nv || b == x || maxit < 0)
        return fail(c, PGD_ERR_INVALID, "pcg_solve: invalid handles or size mismatch");
    const int64_t n = m->nv;
    const bool dbg_t = getenv("PGD_DEBUG_PCG") != nullptr;
    auto dbg_now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double dbg_t0 = dbg_now();
    double dbg_t1 = 0, dbg_t2 = 0, dbg_t3 = 0;
    struct ProfIterGuard { Ctx *c; ~ProfIterGuard() { c->prof_iter = -1; } } prof_iter_guard{c};
    c->prof_pend.clear();
    PGD_TRY(csr_diag_inv(c, m, o));
    bool sym = false;
    PGD_TRY(ensure_sym(c, m, o, &sym));       // SPD solve: read every off-diagonal value once per product
    for (int i = 0; i < 4; ++i) PGD_TRY(ensure_work(c, i, n));
    double *r = c->work[0], *z = c->work[1], *p = c->work[2], *q = c->work[3];
    PGD_HIP(c, hipMemsetAsync(c->flags, 0, 8 * sizeof(int), c->stream));
    // r0 = b - A x0; (r.z, r.r, b.b) land in slots 18..20.  r.z / r.r of iteration k live in slots
    // 16 + 2 (k & 1), +1, so iteration 0 finds "the previous r.z" in slot 18 like every even iteration:
    // all 16-iteration chunks are identical and can be replayed as one hipGraph.
    constexpr int S_INIT = 18, S_PAIR = 16;
    const bool scaled = sym && c->pcg_scaled && n >= 2;
    bool mg_on = false;
    double *sc = z;                     // the scaled recurrence has no z: its buffer holds s = d^-1/2
    // On EVERY exit after x and the slot arrays were scaled (a failing launch, graph replay or copy included): x back to
    // D^-1/2 x~ and the slot arrays no longer taken for A - a later product with this operator must not read the scaled
    // matrix, and the caller must not get x in scaled coordinates.
    struct ScaleGuard {
        Ctx *c; Csr *o; double *x; const double *sc; int64_t n; bool active; bool slots_scaled;
        ~ScaleGuard() {
            if (!active) return;
            (void)vec_div_mul(c, x, sc, n, 1);
            if (!slots_scaled) return;      // (the scaled operator was held as a derived stencil only: the slot arrays still hold A)
            o->uvals_valid = false;
            o->uvals_scaled = false;
        }
    } guard{c, o, x->d, sc, n, false, true};
    // ... and the stencil couplings back to those of A where the scaled operator was held as a derived stencil only (declared after
    // `guard`: destroyed first; the slot arrays were never scaled then)
    struct VirtGuard {
        Csr *o; bool active; double saved[8];
        ~VirtGuard() {
            if (!active) return;
            for (int s2 = 0; s2 < 8; ++s2) o->st_c[s2] = saved[s2];
            o->st_virtual = false;
        }
    } virt{o, false, {0, 0, 0, 0, 0, 0, 0, 0}};
    if (scaled) {
        PGD_TRY(ensure_work(c, 5, 4096));
        unsigned long long *bits = reinterpret_cast<unsigned long long *>(c->work[5]);
        PGD_HIP(c, hipMemsetAsync(bits, 0, sizeof(unsigned long long), c->stream));
        k_scale_in<<<grid_for(n), TPB, 0, c->stream>>>(o->dinv, sc, x->d, n, bits);
        k_dmin_slot<<<1, 1, 0, c->stream>>>(bits, c->slots, c->flags);
        PGD_LAUNCH_CHECK(c);
        guard.active = true;            // x is scaled from here on
        // Where A itself is ONE stencil + eliminated nodes on the whole grid (dia_classify: every row and slot verified - the
        // Galerkin start has classified A from two vectors on) the scaled operator is known without touching a slot: every free
        // row has the diagonal c0, s_i = (1 / c0)^1/2 there and 1 on the eliminated rows, and k_dia_scale would write
        // c_s (s_i s_j) - one product, the same for every pair of free nodes - and keep the exact zeros.  The couplings of
        // D^-1/2 A D^-1/2 are DERIVED with that very arithmetic (k_stencil_derive), the codes are A's: no scaling pass over the
        // slot arrays (0.40 ms at 256^3), no second classification (0.25 ms), and the slot arrays still hold A afterwards.
        if (c->pcg_derive_scaled && m->sym_nx > 0) {
            if (o->cls_count <= 0) PGD_TRY(dia_classify(c, m, o));
            if (stencil_whole_grid(c, m, o) && o->st_ident >= 0 && o->st_c[0] > 0.0) {
                PGD_TRY(ensure_work(c, 5, 4096));
                StencilTuple in;
                for (int s2 = 0; s2 < 8; ++s2) in.c[s2] = o->st_c[s2];
                k_stencil_derive<<<1, 1, 0, c->stream>>>(in, c->spmv_unit_diag, c->work[5] + 16);
                PGD_LAUNCH_CHECK(c);
                double out8[8];
                PGD_HIP(c, hipMemcpyAsync(out8, c->work[5] + 16, sizeof out8, hipMemcpyDeviceToHost, c->stream));
                PGD_HIP(c, hipStreamSynchronize(c->stream));
                bool fin = true;
                for (double v : out8) fin = fin && std::isfinite(v);
                if (fin) {
                    for (int s2 = 0; s2 < 8; ++s2) { virt.saved[s2] = o->st_c[s2]; o->st_c[s2] = out8[s2]; }
                    o->st_virtual = true;
                    virt.active = true;
                    guard.slots_scaled = false;
                }
            }
        }
        if (!virt.active) {
            PGD_TRY(sym_scale(c, m, o, sc));
            PGD_TRY(dia_classify(c, m, o));         // uniform grids: a code byte per row instead of its slot values
        }
        // multigrid preconditioner (PGD_TUNE_PCG_PRECOND): one stencil on a lattice whose eliminated nodes are its hull, else Jacobi.
        // Its cycle works on vectors that vanish on the eliminated rows: x = b there from the start (their exact solution, s = 1)
        if (c->pcg_precond == 1) {
            mg_on = mg_prepare(c, m, o);
            if (mg_on) { c->mg_solves += 1; PGD_TRY(mg_fix_start(c, o, b->d, x->d, n)); }
            else c->mg_fallbacks += 1;
        }
        PGD_TRY(launch_spmv_op(c, m, o, x->d, q, nullptr, 0, n, false, true, nullptr, nullptr));
        const int g = grid_for(n);
        PGD_TRY(ensure_partials(c, 4 * (int64_t)MAX_VEC_BLOCKS));
        k_pcg_init_s<<<g, TPB, 0, c->stream>>>(b->d, q, sc, r, p, n, c->partials);
        PGD_LAUNCH_CHECK(c);
        PGD_TRY(reduce_partials(c, c->partials, g, 3, S_INIT, -1, 0, 0));
    } else {
        if (c->pcg_precond == 1) c->mg_fallbacks += 1;      // (no symmetric storage, or the scaled recurrence switched off: Jacobi)
        PGD_TRY(launch_spmv_op(c, m, o, x->d, q, nullptr, 0, n, false, true, nullptr, nullptr));
        PGD_TRY(pcg_init(c, b->d, q, o->dinv, r, z, p, 0, n, S_INIT));
    }
    k_pcg_tol<<<1, 64, 0, c->stream>>>(c->slots, c->flags, rtol, atol, S_INIT + 1, S_INIT + 2, S_TOL2);
    PGD_LAUNCH_CHECK(c);
    if (mg_on) {                        // p0 = z0 = M r0; the first "previous r.z"
        int np = 0;
        PGD_TRY(mg_vcycle(c, r, true, &np, p));
        PGD_TRY(reduce_partials(c, c->partials, np, 1, S_INIT, -1, 0, 0));
    }

    // second partials buffer for the folded reductions (the x / r update reads the product's partials while writing its own)
    PGD_TRY(ensure_work(c, 6, 4 * (int64_t)MAX_VEC_BLOCKS));
    double *part2 = c->work[6];                         // (the two-launch form of small systems alternates between this half and the next)
    // large systems: the x update rides in the p kernel (PGD_TUNE_PCG_DEFER_X); the folded small-system form keeps its own kernels
    // systems of up to 2^20 rows: single-sync recurrence in two launches per iteration (k_pcg1_step) ...
    const bool fold = !mg_on && scaled && c->pcg_single_sync && c->pcg_small_ss && (n <= ((int64_t)1 << 20) || (n <= c->pcg_small_ss_rows && m->sym_nx > 0));
    // ... or the two-reduction recurrence with its final reduction passes folded into their consumers (three launches)
    const bool folded_form = !mg_on && scaled && c->pcg_fold_reduce && n <= ((int64_t)1 << 20) && !fold;
    const bool single_sync = !mg_on && (fold || (scaled && c->pcg_single_sync && !folded_form && m->sym_nx > 0));
    double *part2b = part2 + 2 * (int64_t)MAX_VEC_BLOCKS;
    const bool deferred_x = !mg_on && scaled && c->pcg_defer_x && !folded_form && !single_sync;
    const bool lag_x = single_sync && c->pcg_lag_x;
    // (two-launch form above 2^20 rows: 512 workgroups in the update, every one of which sums all partial sums)
    const int g2v = grid_for((n + 1) / 2, TPB, (n > ((int64_t)1 << 20) && n <= c->pcg_small_ss_rows && c->pcg_small_ss) ? 512 : MAX_VEC_BLOCKS);
    if (single_sync) {
        // the first look at the residual happens in the first k_pcg1_scalars: hand it the initial residual's sums
        // (two-launch form: iteration k reads the pairs of parity (k - 1) & 1, so the seed goes to the second half)
        k_pcg1_seed<<<8, TPB, 0, c->stream>>>(fold ? part2b : part2, g2v, c->slots, S_INIT, S_INIT + 1);
        if (fold) PGD_HIP(c, hipMemsetAsync(c->slots + S1F_ALPHA, 0, 6 * sizeof(double), c->stream));     // alpha, beta, exact-phase bit x 2 parities
        PGD_LAUNCH_CHECK(c);
    }
    auto enqueue = [&](int start, int count) -> int {
        for (int k = 0; k < count; ++k) {
            const int out = S_PAIR + 2 * ((start + k) & 1), rz_old = S_PAIR + 2 * ((start + k + 1) & 1);
            int nparts = 0;
            c->prof_iter = start + k;                    // launch timing: which iteration a sample belongs to (prof_commit)
            if (single_sync) {
                c->spmv_qq = 1;
                const int rc = launch_spmv_op(c, m, o, p, q, p, 0, n, true, true, c->flags, &nparts);
                c->spmv_qq = 0;
                PGD_TRY(rc);
                const double *prod = c->partials;
                if (nparts > 8192) {
                    const int nb = (nparts + 1023) / 1024;
                    PGD_TRY(ensure_work(c, 5, (int64_t)nb * 2 > 4096 ? (int64_t)nb * 2 : 4096));
                    PGD_TRY(k_reduce_stage1_pub(c, c->partials, nparts, 2, c->work[5]));
                    prod = c->work[5];
                    nparts = nb;
                }
                if (!fold) k_pcg1_scalars<<<1, 1024, 0, c->stream>>>(prod, nparts, part2, g2v, c->slots, c->flags);
                // launch timing on: every third update between HIP events (5 or 7 vector passes = 40 or 56 B per row)
                const bool timed_u = c->prof && ((c->prof_upd_seen++ % 3) == 0);        // one in three: both halves of the x-update pairs get sampled
                if (timed_u) {
                    if (c->ev_used + 2 > c->ev.size()) prof_flush(c);
                    c->ev_rec[c->ev_used / 2] = Ctx::ProfRec{1, c->prof_iter, 0.0, 0.0, 0.0};
                    PGD_HIP(c, hipEventRecord(c->ev[c->ev_used], c->stream));
                }
                // the x update lags behind by one iteration in every other one (chunks start at even iteration indices)
                const int lag = lag_x ? 1 + ((start + k) & 1) : 0;
                if (fold) {
                    const int par = (start + k) & 1;
                    k_pcg1_step<<<g2v, TPB, 0, c->stream>>>(x->d, r, p, q, sc, n, prod, nparts, par ? part2 : part2b, g2v, par ? part2b : part2,
                                                            c->slots, c->flags, par, lag);
                } else if (c->pcg_stream_hints) k_pcg1_update<true, false><<<g2v, TPB, 0, c->stream>>>(x->d, r, p, q, sc, 0, n, c->slots, S1_ALPHA, S1_BETA, part2, c->flags, lag, 0, 0, PushArgs());
                else k_pcg1_update<false, false><<<g2v, TPB, 0, c->stream>>>(x->d, r, p, q, sc, 0, n, c->slots, S1_ALPHA, S1_BETA, part2, c->flags, lag, 0, 0, PushArgs());
                if (timed_u) {
                    PGD_HIP(c, hipEventRecord(c->ev[c->ev_used + 1], c->stream));
                    c->ev_rec[c->ev_used / 2].bytes = (lag == 1 ? 40.0 : 56.0) * (double)n;      // (a lag = 1 launch that meets beta < 0.01 moves 56)
                    c->ev_used += 2;
                }
                PGD_LAUNCH_CHECK(c);
                continue;
            }
            PGD_TRY(launch_spmv_op(c, m, o, p, q, p, 0, n, true, true, c->flags, &nparts));
            if (mg_on) {
                // textbook PCG with z = M r from the V-cycle: the stop test stays the one of the Jacobi form (true r.r in the exact phase)
                const int g2 = grid_for((n + 1) / 2);
                int np = 0;
                PGD_TRY(reduce_partials(c, c->partials, nparts, 1, S_PQ, 0, 0, 0));
                k_pcg_xr_s<<<g2, TPB, 0, c->stream>>>(x->d, r, p, q, sc, n, c->slots, rz_old, S_PQ, c->partials, c->flags);
                PGD_LAUNCH_CHECK(c);
                PGD_TRY(reduce_partials(c, c->partials, g2, 2, out, 2, out + 1, S_TOL2));      // counts the iteration, tests
                PGD_TRY(mg_vcycle(c, r, true, &np));
                PGD_TRY(reduce_partials(c, c->partials, np, 1, out, -1, 0, 0));                // r.z over the r~.r~ the test has used
                k_pcg_p<true><<<g2, TPB, 0, c->stream>>>(p, mg_result(c), 0, n, c->slots, out, rz_old, c->flags);
                PGD_LAUNCH_CHECK(c);
                continue;
            }
            // (pays only where the launches, not the bytes, set the pace: 256^2 rows +22 %, 128^3 +-0, 256^3 -2 %)
            if (scaled && c->pcg_fold_reduce && nparts > 0 && nparts <= 8192 && n <= ((int64_t)1 << 20)) {
                const int g2 = grid_for((n + 1) / 2);
                k_pcg_xr_s2<<<g2, TPB, 0, c->stream>>>(x->d, r, p, q, sc, n, c->slots, rz_old, c->partials, nparts, part2, c->flags);
                k_pcg_p_s2<<<g2, TPB, 0, c->stream>>>(p, r, n, part2, g2, c->slots, out, rz_old, S_TOL2, c->flags);
                PGD_LAUNCH_CHECK(c);
                continue;
            }
            PGD_TRY(reduce_partials(c, c->partials, nparts, 1, S_PQ, 0, 0, 0));
            if (scaled && deferred_x) {
                // x += alpha p rides in the p kernel (8 vector passes per iteration instead of 9)
                const int g2 = grid_for((n + 1) / 2);
                k_pcg_r_s<<<g2, TPB, 0, c->stream>>>(r, q, sc, n, c->slots, rz_old, S_PQ, c->partials, c->flags);
                PGD_LAUNCH_CHECK(c);
                PGD_TRY(reduce_partials(c, c->partials, g2, 2, out, 2, out + 1, S_TOL2));
                k_pcg_px_s<<<g2, TPB, 0, c->stream>>>(x->d, p, r, n, c->slots, out, rz_old, S_PQ, c->flags);
                continue;
            }
            if (scaled) {
                const int g2 = grid_for((n + 1) / 2);
                k_pcg_xr_s<<<g2, TPB, 0, c->stream>>>(x->d, r, p, q, sc, n, c->slots, rz_old, S_PQ, c->partials, c->flags);
                PGD_LAUNCH_CHECK(c);
                PGD_TRY(reduce_partials(c, c->partials, g2, 2, out, 2, out + 1, S_TOL2));
                k_pcg_p<true><<<grid_for((n + 1) / 2), TPB, 0, c->stream>>>(p, r, 0, n, c->slots, out, rz_old, c->flags);
                continue;
            }
            // x, r, z update; the final reduction also runs the convergence test on r.r
            PGD_TRY(pcg_xr(c, x->d, r, p, q, o->dinv, z, 0, n, rz_old, S_PQ, out, 1, S_TOL2));
            k_pcg_p<true><<<grid_for((n + 1) / 2), TPB, 0, c->stream>>>(p, z, 0, n, c->slots, out, rz_old, c->flags);
        }
        return PGD_OK;
    };

    // The 16-iteration chunk (80 dependent launches) is replayed as a hipGraph: small systems are launch-bound, and at
    // 256^3 the replay still saves ~2 % (549 vs 560 us per iteration).  With launch timing on, every PROF_EAGER_EVERY-th
    // chunk is issued eagerly so that its products carry their HIP events; the capture itself records none.
    hipGraphExec_t gexec = nullptr;
    if (dbg_t) { (void)hipStreamSynchronize(c->stream); dbg_t1 = dbg_now(); }
    // (a multigrid iteration is ~50 launches and a solve ~20 iterations: shorter chunks, less queued behind the converged one)
    const int CE = mg_on ? std::max(2, std::min(c->mg_chunk & ~1, CHECK_EVERY)) : CHECK_EVERY;      // (even: iteration k uses the slot pair of parity k & 1, and the chunk is replayed)
    if (maxit >= CE) {
        // everything a chunk allocates lazily must exist before the capture starts
        PGD_TRY(ensure_partials(c, std::max<int64_t>(4 * (int64_t)MAX_VEC_BLOCKS, 2 * ((n + 63) / 64) + 64)));
        PGD_TRY(ensure_work(c, 5, 4096));
        const bool prof_saved = c->prof;
        c->prof = false;
        hipGraph_t graph = nullptr;
        if (hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal) == hipSuccess) {
            const int rc = enqueue(0, CE);
            const hipError_t e = hipStreamEndCapture(c->stream, &graph);
            if (rc != PGD_OK || e != hipSuccess || !graph ||
                hipGraphInstantiate(&gexec, graph, nullptr, nullptr, 0) != hipSuccess)
                gexec = nullptr;
            if (graph) (void)hipGraphDestroy(graph);
            (void)hipGetLastError();
        }
        c->prof = prof_saved;
    }
    int f[4] = {0, 0, 0, 0};
    int enq = 0, rc_loop = PGD_OK;
    if (dbg_t) dbg_t2 = dbg_now();
    auto issue = [&](int chunk) -> int {
        const bool eager_for_timing = c->prof && ((enq / CE) % PROF_EAGER_EVERY == 0);
        if (gexec && chunk == CE && !eager_for_timing) {
            if (hipGraphLaunch(gexec, c->stream) != hipSuccess) return fail(c, PGD_ERR_HIP, "pcg_solve: hipGraphLaunch failed");
            return PGD_OK;
        }
        return enqueue(enq, chunk);
    };
    if (c->pcg_pipeline && pcg_flag_snapshots(c) == PGD_OK) {
        // PIPELINED: the next chunk is queued BEFORE the host waits for the flags of the one before it - a snapshot of the flags
        // into pinned memory + an event behind every chunk - so the GPU never idles through the host's round trip (copy, wake-up,
        // graph launch: 50 - 130 us per 16 iterations, i.e. 2 % of a chunk at 256^3 and a third of one on a 256^2 grid).  A chunk
        // queued behind the iteration that converged is 48 no-op launches (every kernel returns on the done flag), once per solve.
        auto snap = [&](int slot) -> hipError_t {
            hipError_t e = hipMemcpyAsync(c->flags_host + 4 * slot, c->flags, 4 * sizeof(int), hipMemcpyDeviceToHost, c->stream);
            if (e == hipSuccess) e = hipEventRecord(c->flag_ev[slot], c->stream);
            return e;
        };
        int cur = 0;
        hipError_t e = snap(0);                           // the flags after the initial residual
        while (e == hipSuccess) {
            const int chunk = (maxit - enq < CE) ? maxit - enq : CE;
            if (chunk > 0) {
                if ((rc_loop = issue(chunk)) != PGD_OK) break;
                enq += chunk;
                if ((e = snap(cur ^ 1)) != hipSuccess) break;
            }
            if ((e = hipEventSynchronize(c->flag_ev[cur])) != hipSuccess) break;
            for (int i = 0; i < 4; ++i) f[i] = c->flags_host[4 * cur + i];
            if (f[0] || chunk <= 0) break;                // converged (what is queued behind it does nothing), or nothing more to queue
            cur ^= 1;
        }
        if (e != hipSuccess && rc_loop == PGD_OK) rc_loop = fail(c, PGD_ERR_HIP, "pcg_solve: %s", hipGetErrorString(e));
    } else {
        while (true) {
            hipError_t e = hipMemcpyAsync(f, c->flags, sizeof f, hipMemcpyDeviceToHost, c->stream);
            if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
            if (e != hipSuccess) { rc_loop = fail(c, PGD_ERR_HIP, "pcg_solve: %s", hipGetErrorString(e)); break; }
            if (f[0] || enq >= maxit) break;
            const int chunk = (maxit - enq < CE) ? maxit - enq : CE;
            if ((rc_loop = issue(chunk)) != PGD_OK) break;
            enq += chunk;
        }
    }
    if (dbg_t) { (void)hipStreamSynchronize(c->stream); dbg_t3 = dbg_now(); }
    if (dbg_t) fprintf(stderr, "[pcg_solve] n %lld graph %s iterations queued %d counted %d stencil %d cls %d | setup %.2f ms capture %.2f ms loop %.2f ms = %.1f us/it\n", (long long)n, gexec ? "yes" : "NO", enq, f[1], (int)o->st_ok, o->cls_count, 1e3 * (dbg_t1 - dbg_t0), 1e3 * (dbg_t2 - dbg_t1), 1e3 * (dbg_t3 - dbg_t2), 1e6 * (dbg_t3 - dbg_t2) / (f[1] > 0 ? f[1] : 1));
    if (gexec) (void)hipGraphExecDestroy(gexec);
    c->prof_iter = -1;
    // launch timing: the product of iteration k ran if no earlier iteration had set the done flag - f[1] iterations were counted, and in
    // the single-sync form the product of the iteration that NOTICED convergence ran as well; its update, and everything queued
    // behind it, did nothing
    if (c->prof) prof_commit(c, rc_loop == PGD_OK ? f[1] + (single_sync ? 1 : 0) : 0, rc_loop == PGD_OK ? f[1] : 0);
    if (rc_loop != PGD_OK) return rc_loop;
    if (scaled) {      // x = D^-1/2 x~, and the true r.r of the last iterate for the report
        const int g = grid_for(n);
        guard.active = false;
        // converged (or broke down) inside an iteration whose p kernel was a no-op: its x update is still to come
        const bool pending = deferred_x && f[0] != 0 && f[1] > 0;
        // lagged x update: f[1] update kernels ran; if the last one had an even index it may have left its term outstanding
        const bool lag_pending = lag_x && f[1] > 0 && ((f[1] - 1) & 1) == 0;
        // (alpha and beta of that last update: in the two-launch form they sit in the slots of its parity)
        const int last_par = f[1] > 0 ? (f[1] - 1) & 1 : 0;
        k_scale_out<<<g, TPB, 0, c->stream>>>(x->d, r, sc, n, c->partials, (pending || lag_pending) ? p : nullptr, c->slots,
                                              S_PAIR + 2 * (f[1] & 1), S_PQ, lag_pending ? 1 : 0,
                                              fold ? S1F_ALPHA + last_par : S1_ALPHA, fold ? S1F_BETA + last_par : S1_BETA);
        PGD_LAUNCH_CHECK(c);
        PGD_TRY(reduce_partials(c, c->partials, g, 1, S_TMP, -1, 0, 0));
        if (!virt.active) {
            o->uvals_valid = false;    // the slot arrays hold the scaled operator: nobody else may take them for A
            o->uvals_scaled = false;
        }
    }
    PGD_LAUNCH_CHECK(c);
    double s[PGD_NSLOTS];
    PGD_HIP(c, hipMemcpyAsync(s, c->slots, sizeof s, hipMemcpyDeviceToHost, c->stream));
    PGD_HIP(c, hipStreamSynchronize(c->stream));
    if (iters) *iters = f[1];
    const double rr = scaled ? s[S_TMP] : (f[1] > 0) ? s[S_PAIR + 2 * ((f[1] - 1) & 1) + 1] : s[S_INIT + 1];
    const double bb = s[S_INIT + 2];
    if (relres) *relres = (bb > 0.0) ? sqrt(rr / bb) : 0.0;
    if (f[2] != 0) return fail(c, f[2], "pcg_solve: breakdown (NaN residual) after %d iterations", f[1]);
    return PGD_OK;
}

int pgd_band_solve(pgd_handle h, pgd_handle oh, pgd_handle bh, pgd_handle xh) {
    PGD_CTX(c, h);
    Csr *o = get_csr(c, oh);
    Mesh *m = o ? get_mesh(c, o->mesh) : nullptr;
    Vec *b = get_vec(c, bh), *x = get_vec(c, xh);
    if (!o || !m || !b || !x || b->n != m->nv || x->n != m->nv)
        return fail(c, PGD_ERR_INVALID, "band_solve: invalid handles or size mismatch");
    const int64_t n = m->nv, ld = 2 * (int64_t)m->kl + m->ku + 1;
    const int64_t need = n * ld + n;
    if (need > ((int64_t)1 << 27)) return fail(c, PGD_ERR_LIMIT, "band_solve: system too large for the direct path (n=%lld, band=%lld)", (long long)n, (long long)ld);
    const int use_lds = need <= BAND_LDS_DOUBLES;
    if (!use_lds) PGD_TRY(ensure_work(c, 4, need));
    PGD_TRY(ensure_vals(c, m, o));
    PGD_HIP(c, hipMemsetAsync(c->flags, 0, 8 * sizeof(int), c->stream));
    k_band_solve<<<1, TPB, 0, c->stream>>>(m->row_ptr, m->cols, o->vals, b->d, x->d, (int)n, m->kl, m->ku,
                                           use_lds ? nullptr : c->work[4], use_lds, c->flags);
    PGD_LAUNCH_CHECK(c);
    int f[4];
    PGD_HIP(c, hipMemcpyAsync(f, c->flags, sizeof f, hipMemcpyDeviceToHost, c->stream));
    PGD_HIP(c, hipStreamSynchronize(c->stream));
    if (f[2] != 0) return fail(c, PGD_ERR_SINGULAR, "band_solve: zero pivot");
    return PGD_OK;
}

// ---------------------------------------------------------- distributed PCG pieces
static int get3(Ctx *c, pgd_handle a, pgd_handle b, pgd_handle d, Vec **A, Vec **B, Vec **D) {
    *A = get_vec(c, a); *B = get_vec(c, b); *D = get_vec(c, d);
    return (*A && *B && *D && (*A)->n == (*B)->n && (*A)->n == (*D)->n) ? PGD_OK : PGD_ERR_INVALID;
}

static int range_ok(int64_t n, int64_t &lo, int64_t &hi) {
    if (hi < 0) hi = n;
    return lo >= 0 && lo <= hi && hi <= n;
}

int pgd_pcg_init_slot(pgd_handle h, pgd_handle bh, pgd_handle qh, pgd_handle dh, pgd_handle rh, pgd_handle zh,
                      pgd_handle ph, int64_t lo, int64_t hi, int slot) {
    PGD_CTX(c, h);
    Vec *b, *q, *d, *r, *z, *p;
    if (get3(c, bh, qh, dh, &b, &q, &d) != PGD_OK || get3(c, rh, zh, ph, &r, &z, &p) != PGD_OK || b->n != r->n ||
        !range_ok(b->n, lo, hi) || slot < 0 || slot + 3 > PGD_NSLOTS)
        return fail(c, PGD_ERR_INVALID, "pcg_init_slot: invalid arguments");
    if (hi == lo) return PGD_OK;
    return pcg_init(c, b->d, q->d, d->d, r->d, z->d, p->d, lo, hi, slot);
}

int pgd_pcg_tol_slot(pgd_handle h, double rtol, double atol, int slot_rr, int slot_bb, int slot_tol2) {
    PGD_CTX(c, h);
    k_pcg_tol<<<1, 64, 0, c->stream>>>(c->slots, c->flags, rtol, atol, slot_rr, slot_bb, slot_tol2);
    PGD_LAUNCH_CHECK(c);
    return PGD_OK;
}

int pgd_pcg_xr_slot(pgd_handle h, pgd_handle xh, pgd_handle rh, pgd_handle ph, pgd_handle qh, pgd_handle dh,
                    pgd_handle zh, int64_t lo, int64_t hi, int slot_rz, int slot_pq, int slot_out) {
    PGD_CTX(c, h);
    Vec *x, *r, *p, *q, *d, *z;
    if (get3(c, xh, rh, ph, &x, &r, &p) != PGD_OK || get3(c, qh, dh, zh, &q, &d, &z) != PGD_OK || x->n != q->n ||
        !range_ok(x->n, lo, hi) || slot_out < 0 || slot_out + 2 > PGD_NSLOTS)
        return fail(c, PGD_ERR_INVALID, "pcg_xr_slot: invalid arguments");
    if (hi == lo) return PGD_OK;
    return pcg_xr(c, x->d, r->d, p->d, q->d, d->d, z->d, lo, hi, slot_rz, slot_pq, slot_out, 0, 0);
}

int pgd_pcg_check_slot(pgd_handle h, int slot_rr, int slot_tol2) {
    PGD_CTX(c, h);
    k_pcg_check<<<1, 64, 0, c->stream>>>(c->slots, c->flags, slot_rr, slot_tol2);
    PGD_LAUNCH_CHECK(c);
    return PGD_OK;
}

int pgd_pcg_p_slot(pgd_handle h, pgd_handle ph, pgd_handle zh, int64_t lo, int64_t hi, int slot_num, int slot_den) {
    PGD_CTX(c, h);
    Vec *p = get_vec(c, ph), *z = get_vec(c, zh);
    if (!p || !z || p->n != z->n || !range_ok(p->n, lo, hi)) return fail(c, PGD_ERR_INVALID, "pcg_p_slot: invalid arguments");
    if (hi == lo) return PGD_OK;
    if ((lo & 1) == 0) k_pcg_p<true><<<grid_for((hi - lo + 1) / 2), TPB, 0, c->stream>>>(p->d, z->d, lo, hi, c->slots, slot_num, slot_den, c->flags);
    else k_pcg_p<false><<<grid_for(hi - lo), TPB, 0, c->stream>>>(p->d, z->d, lo, hi, c->slots, slot_num, slot_den, c->flags);
    PGD_LAUNCH_CHECK(c);
    return PGD_OK;
}

int pgd_cg_init_slot(pgd_handle h, pgd_handle bh, pgd_handle qh, pgd_handle dh, pgd_handle rh, pgd_handle uh,
                     pgd_handle ph, pgd_handle sh, int64_t lo, int64_t hi, int base) {
    PGD_CTX(c, h);
    Vec *b, *q, *d, *r, *u, *p;
    Vec *s = get_vec(c, sh);
    if (get3(c, bh, qh, dh, &b, &q, &d) != PGD_OK || get3(c, rh, uh, ph, &r, &u, &p) != PGD_OK || !s ||
        b->n != r->n || s->n != b->n || !range_ok(b->n, lo, hi) || base < 0 || base + 9 > PGD_NSLOTS)
        return fail(c, PGD_ERR_INVALID, "cg_init_slot: invalid arguments");
    if (hi == lo) return PGD_OK;
    const int g = grid_for(hi - lo);
    PGD_TRY(ensure_partials(c, 4 * (int64_t)MAX_VEC_BLOCKS));
    k_cg_init<<<g, TPB, 0, c->stream>>>(b->d, q->d, d->d, r->d, u->d, p->d, s->d, lo, hi, c->partials);
    PGD_LAUNCH_CHECK(c);
    // (r.u, r.r) -> S[base], S[base+1]; b.b -> S[base+8]: reduce 3 values to a scratch triple, then place b.b
    PGD_TRY(reduce_partials(c, c->partials, g, 3, 40, -1, 0, 0));
    PGD_HIP(c, hipMemcpyAsync(c->slots + base, c->slots + 40, 2 * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
    PGD_HIP(c, hipMemcpyAsync(c->slots + base + 8, c->slots + 42, sizeof(double), hipMemcpyDeviceToDevice, c->stream));
    return PGD_OK;
}

int pgd_cg_update_slot(pgd_handle h, pgd_handle xh, pgd_handle rh, pgd_handle uh, pgd_handle wh, pgd_handle ph,
                       pgd_handle sh, pgd_handle dh, int64_t lo, int64_t hi, int base) {
    PGD_CTX(c, h);
    Vec *x, *r, *u, *w, *p, *s;
    Vec *d = get_vec(c, dh);
    if (get3(c, xh, rh, uh, &x, &r, &u) != PGD_OK || get3(c, wh, ph, sh, &w, &p, &s) != PGD_OK || !d ||
        x->n != w->n || d->n != x->n || !range_ok(x->n, lo, hi) || base < 0 || base + 9 > PGD_NSLOTS)
        return fail(c, PGD_ERR_INVALID, "cg_update_slot: invalid arguments");
    if (hi == lo) return PGD_OK;
    const int g = grid_for(hi - lo);
    PGD_TRY(ensure_partials(c, 4 * (int64_t)MAX_VEC_BLOCKS));
    k_cg_update<<<g, TPB, 0, c->stream>>>(x->d, r->d, u->d, w->d, p->d, s->d, d->d, lo, hi, c->slots, base,
                                          c->partials, c->flags);
    PGD_LAUNCH_CHECK(c);
    return reduce_partials(c, c->partials, g, 2, base, 0, 0, 0);
}

int pgd_cg_scalars_slot(pgd_handle h, int base, int init, double rtol, double atol) {
    PGD_CTX(c, h);
    if (base < 0 || base + 9 > PGD_NSLOTS) return fail(c, PGD_ERR_INVALID, "cg_scalars_slot: bad slot base");
    k_cg_scalars<<<1, 64, 0, c->stream>>>(c->slots, c->flags, base, init, rtol, atol);
    PGD_LAUNCH_CHECK(c);
    return PGD_OK;
}

}  // extern "C"
