// Direct solve of a small coarsest multigrid level (BASELINE.json north_star: "the coarsest-level dense solve if it
// degenerates"; SURVEY.md config 3: "coarsest solve = GCR ... or dense when tiny").  The reference always hands the
// coarsest system to its coarse_solver, a GCR (src/MG.h:424); this is an opt-in alternative
// (mgcr_mg_param.coarse_direct_rows) for hierarchies whose last level has at most a few thousand unknowns: the level's
// operator is written out as a dense matrix (one apply per unit vector), inverted once at set-up by Gauss-Jordan
// elimination with partial pivoting — two launches per pivot, all rows eliminated in parallel — and the coarsest
// "solve" of every V-cycle is ONE dense mat-vec with that inverse instead of up to max_iter latency-bound GCR
// iterations.  Complex fp64 like everything else; no MFMA: at these sizes the mat-vec is a few microseconds of
// streaming, and an fp64 MFMA formulation would need a batch of right-hand sides the cycle does not have.
#include "internal.h"
#include "reduce.h"

namespace mgcr {

__global__ void __launch_bounds__(256) dense_unit_kernel(cplx *e, int64_t n, int64_t j) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) e[i] = make_double2(i == j ? 1. : 0., 0.);
}
// column j of the augmented matrix M = [A | I] (n x 2n, row-major)
__global__ void __launch_bounds__(256) dense_column_kernel(cplx *M, const cplx *y, int64_t n, int64_t j) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) {
        M[i * 2 * n + j] = y[i];
        M[i * 2 * n + n + j] = make_double2(i == j ? 1. : 0., 0.);
    }
}
// pivot step k: the row p >= k with the largest |M[p][k]| (smallest p among equals: deterministic) is swapped into
// row k and normalised; the column of multipliers is saved for the elimination kernel.  One workgroup.
__global__ void __launch_bounds__(1024) dense_pivot_kernel(cplx *M, int64_t n, int64_t k, cplx *colk, int *singular) {
    __shared__ double sbest[1024];
    __shared__ int64_t sidx[1024];
    const int t = threadIdx.x;
    double best = -1.;
    int64_t bi = k;
    for (int64_t i = k + t; i < n; i += 1024) {
        const cplx v = M[i * 2 * n + k];
        const double a = v.x * v.x + v.y * v.y;
        if (a > best) { best = a; bi = i; }
    }
    sbest[t] = best; sidx[t] = bi;
    __syncthreads();
    for (int s = 512; s >= 1; s >>= 1) {
        if (t < s) {
            const double ob = sbest[t + s];
            const int64_t oi = sidx[t + s];
            if (ob > sbest[t] || (ob == sbest[t] && oi < sidx[t])) { sbest[t] = ob; sidx[t] = oi; }
        }
        __syncthreads();
    }
    const int64_t p = sidx[0];
    if (!(sbest[0] > 0.)) { if (t == 0) *singular = 1; return; }
    const cplx piv = M[p * 2 * n + k];
    __syncthreads();
    for (int64_t j = t; j < 2 * n; j += 1024) {   // swap rows k and p, normalise the new row k
        const cplx a = M[p * 2 * n + j], b = M[k * 2 * n + j];
        if (p != k) M[p * 2 * n + j] = b;
        M[k * 2 * n + j] = cdiv(a, piv);
    }
    __syncthreads();   // (row k's column-k entry is now 1; the other rows' multipliers are their column-k entries)
    for (int64_t i = t; i < n; i += 1024) colk[i] = i == k ? make_double2(0., 0.) : M[i * 2 * n + k];
}
// rows i != k: M[i][:] -= colk[i] * M[k][:]
__global__ void __launch_bounds__(256) dense_elim_kernel(cplx *M, int64_t n, int64_t k, const cplx *__restrict__ colk, const int *singular) {
    if (*singular) return;
    const int64_t i = blockIdx.y;
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i == k || j >= 2 * n) return;
    const cplx f = colk[i];
    if (f.x == 0. && f.y == 0.) return;
    M[i * 2 * n + j] = csub(M[i * 2 * n + j], cmul(f, M[k * 2 * n + j]));
}
__global__ void __launch_bounds__(256) dense_extract_kernel(const cplx *M, cplx *inv, int64_t n) {
    const int64_t i = blockIdx.y;
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (j < n) inv[i * n + j] = M[i * 2 * n + n + j];
}
// x = inv b: one wave per row, lanes stride the columns, wave64 tree (fixed order: reproducible)
__global__ void __launch_bounds__(256) dense_matvec_kernel(const cplx *__restrict__ inv, int64_t n, const cplx *__restrict__ b, cplx *__restrict__ x,
                                                           const int *__restrict__ skip, int skip_it) {
    if (skip && skip[0] < skip[1] + skip_it) return;  // {stop_at, base}: see gcr.hip DevState
    const int64_t row = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    if (row >= n) return;
    cplx s = make_double2(0., 0.);
    for (int64_t j = lane; j < n; j += 64) s = cadd(s, cmul(inv[row * n + j], b[j]));
    s.x = wave_sum(s.x);
    s.y = wave_sum(s.y);
    if (lane == 0) x[row] = s;
}

static unsigned g256(int64_t n) { return (unsigned)((n + 255) / 256); }

// inverse of the n x n operator A (any kind that op_apply_raw can apply), on the device; *inv_out owns n*n entries
int dense_inverse_of(Op *A, int64_t n, cplx **inv_out) {
    MGCR_CHECK(n >= 1 && n <= DENSE_MAX_ROWS, MGCR_ERR_UNSUPPORTED, "dense coarsest solve: %lld rows (limit %d)", (long long)n, DENSE_MAX_ROWS);
    hipStream_t st = ctx().stream;
    cplx *M = nullptr, *e = nullptr, *y = nullptr, *colk = nullptr, *inv = nullptr;
    int *d_sing = nullptr;
    auto done = [&](int rc) {
        hipFree(M); hipFree(e); hipFree(y); hipFree(colk); hipFree(d_sing);
        if (rc != MGCR_OK) hipFree(inv);
        return rc;
    };
    if (hipMalloc((void **)&M, sizeof(cplx) * (size_t)n * 2 * n) != hipSuccess || hipMalloc((void **)&e, sizeof(cplx) * (size_t)n) != hipSuccess ||
        hipMalloc((void **)&y, sizeof(cplx) * (size_t)n) != hipSuccess || hipMalloc((void **)&colk, sizeof(cplx) * (size_t)n) != hipSuccess ||
        hipMalloc((void **)&inv, sizeof(cplx) * (size_t)n * n) != hipSuccess || hipMalloc((void **)&d_sing, sizeof(int)) != hipSuccess) {
        set_error("dense coarsest solve: device allocation failed");
        return done(MGCR_ERR_ALLOC);
    }
    hipMemsetAsync(d_sing, 0, sizeof(int), st);
    const SkipRef keep = get_apply_skip();
    set_apply_skip(SkipRef{});   // set-up applies are never skipped
    int rc = MGCR_OK;
    for (int64_t j = 0; j < n && rc == MGCR_OK; j++) {
        hipLaunchKernelGGL(dense_unit_kernel, dim3(g256(n)), dim3(256), 0, st, e, n, j);
        rc = op_apply_raw(A, e, y, n);
        hipLaunchKernelGGL(dense_column_kernel, dim3(g256(n)), dim3(256), 0, st, M, (const cplx *)y, n, j);
    }
    set_apply_skip(keep);
    if (rc != MGCR_OK) return done(rc);
    for (int64_t k = 0; k < n; k++) {
        hipLaunchKernelGGL(dense_pivot_kernel, dim3(1), dim3(1024), 0, st, M, n, k, colk, d_sing);
        hipLaunchKernelGGL(dense_elim_kernel, dim3(g256(2 * n), (unsigned)n), dim3(256), 0, st, M, n, k, (const cplx *)colk, (const int *)d_sing);
    }
    hipLaunchKernelGGL(dense_extract_kernel, dim3(g256(n), (unsigned)n), dim3(256), 0, st, (const cplx *)M, inv, n);
    int sing = 0;
    MGCR_HIP(hipMemcpyAsync(&sing, d_sing, sizeof(int), hipMemcpyDeviceToHost, st));
    MGCR_HIP(hipStreamSynchronize(st));
    if (hipGetLastError() != hipSuccess) { set_error("dense coarsest solve: kernel launch failed"); return done(MGCR_ERR_HIP); }
    if (sing) { set_error("dense coarsest solve: the coarsest operator is singular"); return done(MGCR_ERR_INVALID); }
    *inv_out = inv;
    return done(MGCR_OK);
}

int dense_apply(const cplx *inv, int64_t n, const cplx *b, cplx *x) {
    const SkipRef sk = get_apply_skip();
    hipLaunchKernelGGL(dense_matvec_kernel, dim3(g256(n * 64)), dim3(256), 0, ctx().stream, inv, n, b, x, sk.p, sk.it);
    MGCR_HIP(hipGetLastError());
    return MGCR_OK;
}

}  // namespace mgcr
