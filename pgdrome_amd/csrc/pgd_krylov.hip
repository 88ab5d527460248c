// BiCGStab with Jacobi scaling for the systems that are NOT symmetric (round 4; VERDICT r03 "missing 3").
//
// The reference hands every per-dimension system to LinearVariationalSolver + MUMPS, which solves whatever the callbacks
// produce (/root/reference/pgdrome/solver.py:627-636, 704-716); a convection atom  u.dx(a) * v * dx  on a 2-D / 3-D space
// (or on a 1-D mesh too long for the banded LU) gives a non-symmetric operator that PCG cannot take.  This is van der Vorst's
// BiCGStab with right Jacobi preconditioning on the CSR product (k_spmv_csr / k_spmv_csr_dict16 through launch_spmv - the
// symmetric half storage and its marches do not apply), the vector work fused into three kernels:
//
//     k_bi_p :  p = r + beta (p - omega v),   y = D^-1 p                      (4 reads, 2 writes)
//     v = A y,  rhat . v
//     k_bi_s :  s = r - alpha v  (in r),      z = D^-1 s,    partial s . s    (3 reads, 2 writes)
//     t = A z,  t . s, t . t
//     k_bi_x :  x += alpha y + omega z,  r = s - omega t,  partial r . r, rhat . r      (6 reads, 2 writes)
//
// Two products and three host synchronisations per iteration (the scalars alpha, omega, beta are formed on the host from
// device reductions in fixed order: bitwise reproducible).  Same stop test as pgd_pcg_solve: ||b - A x|| <= max(rtol ||b||, atol)
// on the recurrence residual, confirmed on the TRUE residual at the end (one more product); breakdown (rhat . v = 0 or
// t . t = 0) restarts from the current x with rhat = r, at most 4 times.  HBM-bound: 2 x (12 nnz + 20 n) + 176 n bytes
// per iteration.
#include "pgd_internal.h"

#include <cmath>

namespace pgd {

__global__ __launch_bounds__(TPB) void k_bi_p(double *__restrict__ p, const double *__restrict__ r, const double *__restrict__ v,
                                              const double *__restrict__ dinv, double *__restrict__ y, double beta, double omega, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x; i < n; i += (int64_t)gridDim.x * TPB) {
        const double pi = fma(beta, fma(-omega, v[i], p[i]), r[i]);
        p[i] = pi;
        y[i] = dinv[i] * pi;
    }
}

__global__ __launch_bounds__(TPB) void k_bi_s(double *__restrict__ r, const double *__restrict__ v, const double *__restrict__ dinv,
                                              double *__restrict__ z, double alpha, int64_t n, double *__restrict__ partials) {
    __shared__ double s_red[4];
    double acc = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x; i < n; i += (int64_t)gridDim.x * TPB) {
        const double s = fma(-alpha, v[i], r[i]);
        r[i] = s;
        z[i] = dinv[i] * s;
        acc = fma(s, s, acc);
    }
    acc = block_sum(acc, s_red);
    if (threadIdx.x == 0) partials[blockIdx.x] = acc;
}

// two dots in one pass: partials[2 b] = sum a_i c_i, partials[2 b + 1] = sum a_i a_i
__global__ __launch_bounds__(TPB) void k_bi_dots(const double *__restrict__ a, const double *__restrict__ c2, int64_t n,
                                                 double *__restrict__ partials) {
    __shared__ double s_red[4];
    double ac = 0.0, aa = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x; i < n; i += (int64_t)gridDim.x * TPB) {
        const double ai = a[i];
        ac = fma(ai, c2[i], ac);
        aa = fma(ai, ai, aa);
    }
    ac = block_sum(ac, s_red);
    __syncthreads();
    aa = block_sum(aa, s_red);
    if (threadIdx.x == 0) { partials[2 * blockIdx.x] = ac; partials[2 * blockIdx.x + 1] = aa; }
}

__global__ __launch_bounds__(TPB) void k_bi_x(double *__restrict__ x, double *__restrict__ r, const double *__restrict__ y,
                                              const double *__restrict__ z, const double *__restrict__ t, const double *__restrict__ rhat,
                                              double alpha, double omega, int64_t n, double *__restrict__ partials) {
    __shared__ double s_red[4];
    double rr = 0.0, hr = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x; i < n; i += (int64_t)gridDim.x * TPB) {
        x[i] = fma(omega, z[i], fma(alpha, y[i], x[i]));
        const double ri = fma(-omega, t[i], r[i]);
        r[i] = ri;
        rr = fma(ri, ri, rr);
        hr = fma(rhat[i], ri, hr);
    }
    rr = block_sum(rr, s_red);
    __syncthreads();
    hr = block_sum(hr, s_red);
    if (threadIdx.x == 0) { partials[2 * blockIdx.x] = rr; partials[2 * blockIdx.x + 1] = hr; }
}

// r = b - q, partials of r . r and b . b
__global__ __launch_bounds__(TPB) void k_bi_res(const double *__restrict__ b, const double *__restrict__ q, double *__restrict__ r, int64_t n,
                                                double *__restrict__ partials) {
    __shared__ double s_red[4];
    double rr = 0.0, bb = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x; i < n; i += (int64_t)gridDim.x * TPB) {
        const double bi = b[i], ri = bi - q[i];
        r[i] = ri;
        rr = fma(ri, ri, rr);
        bb = fma(bi, bi, bb);
    }
    rr = block_sum(rr, s_red);
    __syncthreads();
    bb = block_sum(bb, s_red);
    if (threadIdx.x == 0) { partials[2 * blockIdx.x] = rr; partials[2 * blockIdx.x + 1] = bb; }
}

__global__ __launch_bounds__(TPB) void k_bi_axpy(double *__restrict__ x, const double *__restrict__ y, double a, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x; i < n; i += (int64_t)gridDim.x * TPB) x[i] = fma(a, y[i], x[i]);
}

}  // namespace pgd

using namespace pgd;

extern "C" int pgd_bicgstab_solve(pgd_handle h, pgd_handle oh, pgd_handle bh, pgd_handle xh, double rtol, double atol, int maxit,
                                  int *iters, double *relres) {
    PGD_CTX(c, h);
    Csr *o = get_csr(c, oh);
    Mesh *m = o ? get_mesh(c, o->mesh) : nullptr;
    Vec *b = get_vec(c, bh), *x = get_vec(c, xh);
    if (!o || !m || !b || !x || b->n != m->nv || x->n != m->nv || b == x || maxit < 0 || !iters || !relres)
        return fail(c, PGD_ERR_INVALID, "bicgstab_solve: invalid handles or size mismatch");
    const int64_t n = m->nv;
    *iters = 0; *relres = 0.0;
    if (n == 0) return PGD_OK;
    PGD_TRY(csr_diag_inv(c, m, o));           // (forms the CSR values of a deferred combine as well)
    // seven work vectors of the call's own (the context's work buffers belong to the PCG and the Galerkin start)
    struct Work {
        Ctx *c; double *p[7]; size_t bytes;
        ~Work() { for (double *q : p) if (q) dev_release(c, q, bytes); }
    } W{c, {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr}, (size_t)n * sizeof(double)};
    for (int i = 0; i < 7; ++i) { void *q; PGD_TRY(dev_alloc(c, &q, W.bytes)); W.p[i] = (double *)q; }
    double *r = W.p[0], *rhat = W.p[1], *p = W.p[2], *v = W.p[3], *y = W.p[4], *z = W.p[5], *t = W.p[6];
    const int g = grid_for(n);
    PGD_TRY(ensure_partials(c, 4 * (int64_t)MAX_VEC_BLOCKS));
    PGD_TRY(ensure_work(c, 6, 256));
    double *res = c->work[6];
    double host[2];
    auto two = [&](int nparts) -> int {        // the two sums of the partials -> host
        PGD_TRY(reduce_partials_to(c, c->partials, nparts, 2, res));
        PGD_HIP(c, hipMemcpyAsync(host, res, 2 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
        PGD_HIP(c, hipStreamSynchronize(c->stream));
        return PGD_OK;
    };
    auto product = [&](const double *in, double *out) -> int {
        return launch_spmv(c, m, o->vals, in, out, nullptr, 0, n, false, true, nullptr, nullptr);
    };
    auto residual = [&](double *rr, double *bb) -> int {      // r = b - A x
        PGD_TRY(product(x->d, v));
        k_bi_res<<<g, TPB, 0, c->stream>>>(b->d, v, r, n, c->partials);
        PGD_LAUNCH_CHECK(c);
        PGD_TRY(two(g));
        *rr = host[0]; *bb = host[1];
        return PGD_OK;
    };
    double rr = 0.0, bb = 0.0;
    PGD_TRY(residual(&rr, &bb));
    const double bnorm = std::sqrt(bb), tol = std::max(rtol * bnorm, atol);
    auto rel = [&](double q) { return bnorm > 0.0 ? std::sqrt(q) / bnorm : std::sqrt(q); };
    if (!(rr == rr)) return fail(c, PGD_ERR_INVALID, "bicgstab_solve: the start residual is not finite");
    int it = 0, restarts = 0;
    bool fresh = true;                // rhat = r, p = v = 0 to be set up
    double rho = 1.0, alpha = 1.0, omega = 1.0, rho_new = rr;
    while (std::sqrt(rr) > tol && it < maxit) {
        if (fresh) {
            PGD_HIP(c, hipMemcpyAsync(rhat, r, W.bytes, hipMemcpyDeviceToDevice, c->stream));
            PGD_HIP(c, hipMemsetAsync(p, 0, W.bytes, c->stream));
            PGD_HIP(c, hipMemsetAsync(v, 0, W.bytes, c->stream));
            rho = alpha = omega = 1.0;
            rho_new = rr;             // rhat . r with rhat = r
            fresh = false;
        }
        const double beta = (rho_new / rho) * (alpha / omega);
        k_bi_p<<<g, TPB, 0, c->stream>>>(p, r, v, o->dinv, y, beta, omega, n);
        PGD_LAUNCH_CHECK(c);
        PGD_TRY(product(y, v));
        PGD_TRY(vec_dot_range(c, rhat, v, 0, n, S_TMP));
        PGD_HIP(c, hipMemcpyAsync(host, c->slots + S_TMP, sizeof(double), hipMemcpyDeviceToHost, c->stream));
        PGD_HIP(c, hipStreamSynchronize(c->stream));
        const double hv = host[0];
        if (hv == 0.0 || !(hv == hv)) {                       // breakdown: restart from the current x
            if (++restarts > 4) return fail(c, PGD_ERR_SINGULAR, "bicgstab_solve: breakdown (rhat . v = 0) after %d iterations", it);
            PGD_TRY(residual(&rr, &bb));
            fresh = true;
            continue;
        }
        alpha = rho_new / hv;
        k_bi_s<<<g, TPB, 0, c->stream>>>(r, v, o->dinv, z, alpha, n, c->partials);       // r now holds s
        PGD_LAUNCH_CHECK(c);
        PGD_TRY(reduce_partials_to(c, c->partials, g, 1, res));
        PGD_HIP(c, hipMemcpyAsync(host, res, sizeof(double), hipMemcpyDeviceToHost, c->stream));
        PGD_HIP(c, hipStreamSynchronize(c->stream));
        const double ss = host[0];
        ++it;
        if (std::sqrt(ss) <= tol) {                            // converged in the half step: x += alpha y
            k_bi_axpy<<<g, TPB, 0, c->stream>>>(x->d, y, alpha, n);
            PGD_LAUNCH_CHECK(c);
            rr = ss;
            break;
        }
        PGD_TRY(product(z, t));
        k_bi_dots<<<g, TPB, 0, c->stream>>>(t, r, n, c->partials);
        PGD_LAUNCH_CHECK(c);
        PGD_TRY(two(g));
        const double ts = host[0], tt = host[1];
        if (tt == 0.0 || !(tt == tt)) {
            k_bi_axpy<<<g, TPB, 0, c->stream>>>(x->d, y, alpha, n);
            PGD_LAUNCH_CHECK(c);
            if (++restarts > 4) return fail(c, PGD_ERR_SINGULAR, "bicgstab_solve: breakdown (t . t = 0) after %d iterations", it);
            PGD_TRY(residual(&rr, &bb));
            fresh = true;
            continue;
        }
        omega = ts / tt;
        k_bi_x<<<g, TPB, 0, c->stream>>>(x->d, r, y, z, t, rhat, alpha, omega, n, c->partials);
        PGD_LAUNCH_CHECK(c);
        PGD_TRY(two(g));
        rr = host[0];
        rho = rho_new;
        rho_new = host[1];
        if (!(rr == rr)) return fail(c, PGD_ERR_INVALID, "bicgstab_solve: the residual is not finite after %d iterations", it);
        if (omega == 0.0 || rho_new == 0.0) {                  // stagnation of the stabilising step / serious breakdown: restart
            if (std::sqrt(rr) <= tol) break;
            if (++restarts > 4) return fail(c, PGD_ERR_SINGULAR, "bicgstab_solve: breakdown (omega or rho = 0) after %d iterations", it);
            PGD_TRY(residual(&rr, &bb));
            fresh = true;
        }
    }
    // the recurrence residual drifts from the true one: confirm on b - A x, and go on from there if it is not there yet
    for (int pass = 0; pass < 3; ++pass) {
        double rt = 0.0;
        PGD_TRY(residual(&rt, &bb));
        rr = rt;
        if (std::sqrt(rr) <= tol * 1.0000001 || it >= maxit) break;
        // a short second leg from the true residual (same loop, restarted)
        int it2 = 0;
        PGD_HIP(c, hipMemcpyAsync(rhat, r, W.bytes, hipMemcpyDeviceToDevice, c->stream));
        PGD_HIP(c, hipMemsetAsync(p, 0, W.bytes, c->stream));
        PGD_HIP(c, hipMemsetAsync(v, 0, W.bytes, c->stream));
        rho = alpha = omega = 1.0; rho_new = rr;
        while (std::sqrt(rr) > tol && it < maxit && it2 < 50) {
            const double beta = (rho_new / rho) * (alpha / omega);
            k_bi_p<<<g, TPB, 0, c->stream>>>(p, r, v, o->dinv, y, beta, omega, n);
            PGD_LAUNCH_CHECK(c);
            PGD_TRY(product(y, v));
            PGD_TRY(vec_dot_range(c, rhat, v, 0, n, S_TMP));
            PGD_HIP(c, hipMemcpyAsync(host, c->slots + S_TMP, sizeof(double), hipMemcpyDeviceToHost, c->stream));
            PGD_HIP(c, hipStreamSynchronize(c->stream));
            if (host[0] == 0.0 || !(host[0] == host[0])) break;
            alpha = rho_new / host[0];
            k_bi_s<<<g, TPB, 0, c->stream>>>(r, v, o->dinv, z, alpha, n, c->partials);
            PGD_LAUNCH_CHECK(c);
            PGD_TRY(product(z, t));
            k_bi_dots<<<g, TPB, 0, c->stream>>>(t, r, n, c->partials);
            PGD_LAUNCH_CHECK(c);
            PGD_TRY(two(g));
            if (host[1] == 0.0 || !(host[1] == host[1])) { k_bi_axpy<<<g, TPB, 0, c->stream>>>(x->d, y, alpha, n); break; }
            omega = host[0] / host[1];
            k_bi_x<<<g, TPB, 0, c->stream>>>(x->d, r, y, z, t, rhat, alpha, omega, n, c->partials);
            PGD_LAUNCH_CHECK(c);
            PGD_TRY(two(g));
            rr = host[0]; rho = rho_new; rho_new = host[1];
            ++it; ++it2;
            if (omega == 0.0 || rho_new == 0.0 || !(rr == rr)) break;
        }
    }
    *iters = it;
    *relres = rel(rr);
    return PGD_OK;
}
