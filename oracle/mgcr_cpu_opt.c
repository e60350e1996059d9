/* TEST / BENCH INFRASTRUCTURE — an OPTIMISED CPU port of the hot path, used only as the second CPU
 * baseline of bench.py ("an 'optimised CPU' row (OpenMP over all host cores, fused
 * ops) so the GPU speed-up is not judged only against a strawman").  It is NOT the oracle (its dot
 * products are summed by an OpenMP reduction, in no fixed order) and never part of the product path.
 *
 * What it does: restarted GCR (the algorithm of src/GCR.h:158-302, same alpha/beta conjugation) on the
 * 3-D 7-point Poisson matrix in CSR with int32 columns, with the passes of one iteration fused the way
 * the HIP kernels fuse them (SURVEY §8(d) accounting: B_spmv + (13 + 3 lim) V per iteration):
 *   update   x += a p, r -= a Ap, |r|^2                     one pass
 *   spmv     Ar = A r
 *   dots     <Ar, Aps_j> for all stored j                    one pass
 *   build    p' = r - sum b_j ps_j, Ap' = Ar - sum b_j Aps_j, <r,Ap'>, <Ap',Ap'>   one pass
 * all loops `omp parallel for` with static scheduling and first-touch initialisation. */
#include <complex.h>
#include <math.h>
#include <omp.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef double complex cplx;

static uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

/* Runs `iters` iterations of GCR(restart) on Poisson n^3 with the repo's deterministic RHS (seed 0),
 * x0 = 0.  Returns seconds spent in the iterations (set-up excluded); hist[0..iters] = |r|/|b|. */
double orc_opt_gcr_poisson(int64_t n, int restart, int iters, int nthreads, double *hist) {
    if (nthreads > 0) omp_set_num_threads(nthreads);
    const int64_t N = n * n * n;
    int32_t *rowptr = (int32_t *)malloc(sizeof(int32_t) * (size_t)(N + 1));
    int32_t *col = (int32_t *)malloc(sizeof(int32_t) * (size_t)N * 7);
    double *val = (double *)malloc(sizeof(double) * (size_t)N * 7); /* the matrix is real: 12 B per entry, like the GPU slab */
    cplx *x = (cplx *)malloc(sizeof(cplx) * (size_t)N), *r = (cplx *)malloc(sizeof(cplx) * (size_t)N);
    cplx *ar = (cplx *)malloc(sizeof(cplx) * (size_t)N), *b = (cplx *)malloc(sizeof(cplx) * (size_t)N);
    cplx **ps = (cplx **)malloc(sizeof(cplx *) * (size_t)restart), **aps = (cplx **)malloc(sizeof(cplx *) * (size_t)restart);
    for (int s = 0; s < restart; s++) {
        ps[s] = (cplx *)malloc(sizeof(cplx) * (size_t)N);
        aps[s] = (cplx *)malloc(sizeof(cplx) * (size_t)N);
    }
    /* row pointers (serial prefix over n^3 rows is cheap), then parallel first-touch fill */
    rowptr[0] = 0;
    for (int64_t i = 0; i < n; i++)
        for (int64_t j = 0; j < n; j++)
            for (int64_t k = 0; k < n; k++) {
                int cnt = 1 + (i > 0) + (j > 0) + (k > 0) + (k < n - 1) + (j < n - 1) + (i < n - 1);
                int64_t row = (i * n + j) * n + k;
                rowptr[row + 1] = rowptr[row] + cnt;
            }
#pragma omp parallel for schedule(static)
    for (int64_t row = 0; row < N; row++) {
        int64_t i = row / (n * n), j = (row / n) % n, k = row % n;
        int32_t p = rowptr[row];
        if (i > 0) { col[p] = (int32_t)(row - n * n); val[p++] = -1.; }
        if (j > 0) { col[p] = (int32_t)(row - n); val[p++] = -1.; }
        if (k > 0) { col[p] = (int32_t)(row - 1); val[p++] = -1.; }
        col[p] = (int32_t)row; val[p++] = 6.;
        if (k < n - 1) { col[p] = (int32_t)(row + 1); val[p++] = -1.; }
        if (j < n - 1) { col[p] = (int32_t)(row + n); val[p++] = -1.; }
        if (i < n - 1) { col[p] = (int32_t)(row + n * n); val[p++] = -1.; }
        uint64_t a = splitmix64(2 * (uint64_t)row), c = splitmix64(2 * (uint64_t)row + 1);
        b[row] = ((double)(a % 2000) / 1000. - 1.) + I * ((double)(c % 2000) / 1000. - 1.);
        x[row] = 0.;
        r[row] = b[row];
        ar[row] = 0.;
        for (int s = 0; s < restart; s++) { ps[s][row] = 0.; aps[s][row] = 0.; }
    }
    double bb = 0.;
#pragma omp parallel for schedule(static) reduction(+ : bb)
    for (int64_t i = 0; i < N; i++) bb += creal(b[i]) * creal(b[i]) + cimag(b[i]) * cimag(b[i]);
    /* p0 = r, Ap0 = A p0 */
    double n_re = 0., n_im = 0., d_re = 0.;
#pragma omp parallel for schedule(static) reduction(+ : n_re, n_im, d_re)
    for (int64_t row = 0; row < N; row++) {
        cplx s = 0.;
        for (int32_t l = rowptr[row]; l < rowptr[row + 1]; l++) s += val[l] * r[col[l]];
        ps[0][row] = r[row];
        aps[0][row] = s;
        cplx t = conj(r[row]) * s;
        n_re += creal(t); n_im += cimag(t);
        d_re += creal(s) * creal(s) + cimag(s) * cimag(s);
    }
    cplx num = n_re + I * n_im;
    double *den = (double *)calloc((size_t)restart, sizeof(double));
    double dcur = d_re;
    if (hist) hist[0] = 1.;
    int cur = 0, iter_count = 0;
    const double t0 = omp_get_wtime();
    for (int it = 1; it <= iters; it++) {
        iter_count++;
        const cplx alpha = num / dcur;
        den[cur] = dcur;
        double rr = 0.;
        cplx *p = ps[cur], *ap = aps[cur];
#pragma omp parallel for schedule(static) reduction(+ : rr)
        for (int64_t i = 0; i < N; i++) {
            x[i] += alpha * p[i];
            cplx rn = r[i] - alpha * ap[i];
            r[i] = rn;
            rr += creal(rn) * creal(rn) + cimag(rn) * cimag(rn);
        }
        if (hist) hist[it] = sqrt(rr) / sqrt(bb);
        const int lim = restart < iter_count ? restart : iter_count;
        /* SpMV fused with the beta numerators */
        double bre[16], bim[16];
        for (int j = 0; j < lim; j++) bre[j] = bim[j] = 0.;
#pragma omp parallel
        {
            double lre[16], lim_[16];
            for (int j = 0; j < lim; j++) lre[j] = lim_[j] = 0.;
#pragma omp for schedule(static) nowait
            for (int64_t row = 0; row < N; row++) {
                cplx s = 0.;
                for (int32_t l = rowptr[row]; l < rowptr[row + 1]; l++) s += val[l] * r[col[l]];
                ar[row] = s;
                for (int j = 0; j < lim; j++) {
                    cplx t = conj(s) * aps[j][row];
                    lre[j] += creal(t); lim_[j] += cimag(t);
                }
            }
#pragma omp critical
            for (int j = 0; j < lim; j++) { bre[j] += lre[j]; bim[j] += lim_[j]; }
        }
        cplx beta[16];
        for (int j = 0; j < lim; j++) beta[j] = (bre[j] + I * bim[j]) / den[j];
        int ic_next = iter_count;
        if (iter_count % restart == 0) ic_next = 0;
        const int nxt = ic_next % restart;
        n_re = n_im = d_re = 0.;
        cplx *pn = ps[nxt], *apn = aps[nxt];
#pragma omp parallel for schedule(static) reduction(+ : n_re, n_im, d_re)
        for (int64_t i = 0; i < N; i++) {
            cplx pc = r[i], ac = ar[i];
            for (int j = 0; j < lim; j++) { pc -= beta[j] * ps[j][i]; ac -= beta[j] * aps[j][i]; }
            pn[i] = pc;
            apn[i] = ac;
            cplx t = conj(r[i]) * ac;
            n_re += creal(t); n_im += cimag(t);
            d_re += creal(ac) * creal(ac) + cimag(ac) * cimag(ac);
        }
        num = n_re + I * n_im;
        dcur = d_re;
        iter_count = ic_next;
        cur = nxt;
    }
    const double dt = omp_get_wtime() - t0;
    for (int s = 0; s < restart; s++) { free(ps[s]); free(aps[s]); }
    free(ps); free(aps); free(den); free(rowptr); free(col); free(val); free(x); free(r); free(ar); free(b);
    return dt;
}
