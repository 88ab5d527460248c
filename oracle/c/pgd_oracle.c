/*
 * pgd_oracle.c - ORACLE, TEST INFRASTRUCTURE ONLY (not product code).
 *
 * Plain-C restatement (OpenMP over the host cores) of the per-dimension linear
 * solve the reference delegates to PETSc through
 *     solver.parameters[...] = settings          (pgdrome/solver.py:593-594, 634-635)
 * with settings = {"linear_solver": "cg", "preconditioner": "jacobi"}: CSR
 * sparse matrix-vector product and the Jacobi-preconditioned conjugate-gradient
 * recurrence, identical to oracle/fem_numpy.py:pcg_jacobi (which the tests pin).
 * Used (a) to check that recurrence in C against the numpy one and (b) as the
 * timed CPU baseline of bench.py at BASELINE.json's full size, where numpy/scipy
 * is single-threaded.  PETSc/FEniCS themselves are absent from this image:
 * this is a port of the algorithm, not the reference binary ("kind": "port").
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#ifdef _OPENMP
#include <omp.h>
#endif

void orc_set_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

int orc_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

void orc_spmv(int64_t n, const int32_t *rp, const int32_t *cols, const double *vals, const double *x, double *y) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) {
        double s = 0.0;
        for (int32_t k = rp[i]; k < rp[i + 1]; ++k) s += vals[k] * x[cols[k]];
        y[i] = s;
    }
}

static double dot(int64_t n, const double *a, const double *b) {
    double s = 0.0;
#pragma omp parallel for reduction(+ : s) schedule(static)
    for (int64_t i = 0; i < n; ++i) s += a[i] * b[i];
    return s;
}

/* returns 0, or -1 on allocation failure; x holds the start vector and the solution */
int orc_pcg_jacobi(int64_t n, const int32_t *rp, const int32_t *cols, const double *vals, const double *b,
                   double *x, double rtol, double atol, int maxit, int *iters, double *relres) {
    double *r = malloc(n * sizeof(double)), *z = malloc(n * sizeof(double)), *p = malloc(n * sizeof(double)),
           *q = malloc(n * sizeof(double)), *dinv = malloc(n * sizeof(double));
    if (!r || !z || !p || !q || !dinv) { free(r); free(z); free(p); free(q); free(dinv); return -1; }
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) {
        double d = 0.0;
        for (int32_t k = rp[i]; k < rp[i + 1]; ++k)
            if (cols[k] == i) d = vals[k];
        dinv[i] = 1.0 / d;
    }
    orc_spmv(n, rp, cols, vals, x, q);
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) { r[i] = b[i] - q[i]; z[i] = dinv[i] * r[i]; p[i] = z[i]; }
    const double bnorm = sqrt(dot(n, b, b));
    double tol = rtol * bnorm; if (atol > tol) tol = atol;
    double rr = dot(n, r, r), rz = dot(n, r, z);
    int it = 0;
    if (sqrt(rr) > tol) {
        while (it < maxit) {
            orc_spmv(n, rp, cols, vals, p, q);
            const double alpha = rz / dot(n, p, q);
            double rz_new = 0.0; rr = 0.0;
#pragma omp parallel for reduction(+ : rz_new, rr) schedule(static)
            for (int64_t i = 0; i < n; ++i) {
                x[i] += alpha * p[i];
                r[i] -= alpha * q[i];
                z[i] = dinv[i] * r[i];
                rz_new += r[i] * z[i];
                rr += r[i] * r[i];
            }
            ++it;
            if (sqrt(rr) <= tol) break;
            const double beta = rz_new / rz;
#pragma omp parallel for schedule(static)
            for (int64_t i = 0; i < n; ++i) p[i] = z[i] + beta * p[i];
            rz = rz_new;
        }
    }
    if (iters) *iters = it;
    if (relres) *relres = bnorm > 0.0 ? sqrt(rr) / bnorm : 0.0;
    free(r); free(z); free(p); free(q); free(dinv);
    return 0;
}
