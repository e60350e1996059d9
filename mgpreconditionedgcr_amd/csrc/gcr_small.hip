// One-workgroup GCR for small systems (latency regime).
//
// The multi-kernel loop of gcr.hip costs 4 launches per iteration; on a 3072-row system (the
// reference's own data/sample_matrix case, BASELINE config 1) or on the coarsest multigrid level
// every one of those kernels runs for a few microseconds and the solve is bound by launch latency
// and by the start/end of each grid, not by bandwidth.  Here the WHOLE solve — set-up, every
// iteration's SpMV, dot products, updates, history and the convergence test of GCR::solve
// (src/GCR.h:158-302) — is one launch of one 1024-thread workgroup: phases are separated by
// workgroup barriers instead of kernel boundaries, all vectors stay in L2, scalars in registers / LDS.
// Thread t owns elements t, t+1024, ... of every vector and the same rows of the matrix, so the only
// cross-thread dependency is the x-gather of the SpMV (one barrier) and the reductions.
//
// Same arithmetic as the multi-kernel path (same element-wise operation order, same conjugation
// quirks, -ffp-contract=off); the dot products are summed in yet another fixed order (per-thread
// strided partial sums, wave shuffle tree, waves in index order), i.e. results differ from the
// reference only by re-association, like everywhere else (tests/test_gpu_parity.py docstring).
//
// Scope: Sparse / DiracOp operators (ELL slab + CSR tail), restart or truncation mode with at most
// SMALL_MAX_DIRS stored directions, no preconditioner, single GPU.
//
// Larger systems (up to one row per thread of the CHIP) have their own one-launch solver, gcr_resident.hip.  Round 1's
// attempt at it (hipLaunchCooperativeKernel, grid barriers between the phases, vectors in memory: 200 us per iteration
// on a 262 144-row coarsest level against 45 us for the multi-kernel loop) failed on two counts that file's header
// explains: fenced barriers, and Krylov vectors that were re-read from memory every phase instead of living in registers.
#include <climits>

#include "internal.h"
#include "reduce.h"

namespace mgcr {

constexpr int SMALL_MAX_DIRS = 8;
constexpr int SMALL_THREADS = 1024;

struct SmallArgs {
    // operator
    int64_t n;
    int32_t W, L, nchunk;
    int64_t npad;
    const cplx *ell_val;
    const double *ell_val_re;
    const int32_t *ell_col;
    int64_t n_tail_rows;
    const int32_t *tail_rows, *tail_ptr, *tail_col;
    const cplx *tail_val;
    int shift;
    cplx k;
    // solver
    int storage, restart, max_iter, use_x0;
    double tol2;
    const cplx *rhs;
    cplx *x, *r, *ar;
    cplx *ps[SMALL_MAX_DIRS], *aps[SMALL_MAX_DIRS];
    double *hist;
    int hist_cap;
    int *state;  // DevState prefix of gcr.hip: [0] stop_at (INT_MAX while running / iteration it ended), [1] base, [2] iterations done
    const int *outer_skip;
    int outer_it;
};

__device__ __forceinline__ cplx small_row(const SmallArgs &a, int64_t row, const cplx *__restrict__ x) {
    cplx sum = make_double2(0., 0.);
    for (int32_t w = 0; w < a.nchunk * a.L; w++) {
        int64_t idx = ((int64_t)(w / a.L) * a.npad + row) * a.L + (w % a.L);
        cplx xv = x[a.ell_col[idx]];
        if (a.ell_val_re) {
            double v = a.ell_val_re[idx];
            sum = cadd(sum, make_double2(v * xv.x, v * xv.y));
        } else {
            sum = cadd(sum, cmul(a.ell_val[idx], xv));
        }
    }
    if (a.n_tail_rows) {  // tail_rows is ascending: binary search
        int64_t lo = 0, hi = a.n_tail_rows;
        while (lo < hi) {
            int64_t mid = (lo + hi) >> 1;
            if (a.tail_rows[mid] < row) lo = mid + 1; else hi = mid;
        }
        if (lo < a.n_tail_rows && a.tail_rows[lo] == row)
            for (int32_t i = a.tail_ptr[lo]; i < a.tail_ptr[lo + 1]; i++) sum = cadd(sum, cmul(a.tail_val[i], x[a.tail_col[i]]));
    }
    return a.shift ? csub(x[row], cmul(a.k, sum)) : sum;
}

__global__ void __launch_bounds__(SMALL_THREADS) gcr_small_kernel(SmallArgs a) {
    __shared__ double lds[2 * (SMALL_MAX_DIRS + 2) * 17];
    const int tid = threadIdx.x;
    const int64_t n = a.n;
    if (a.outer_skip && a.outer_skip[0] < a.outer_skip[1] + a.outer_it) {  // an outer solve that is already over silences this one
        if (tid == 0) { a.state[0] = -1; a.state[1] = 0; a.state[2] = 0; }
        return;
    }
    // ---- set-up (src/GCR.h:189-216): r = rhs (or rhs - A x0); p = r; Ap = A p; slot 0 <- (p, Ap)
    if (a.use_x0) {
        for (int64_t i = tid; i < n; i += SMALL_THREADS) a.ar[i] = small_row(a, i, a.x);
        __syncthreads();
        for (int64_t i = tid; i < n; i += SMALL_THREADS) a.r[i] = csub(a.rhs[i], a.ar[i]);
    } else {
        for (int64_t i = tid; i < n; i += SMALL_THREADS) a.r[i] = a.rhs[i];
    }
    for (int64_t i = tid; i < n; i += SMALL_THREADS) a.ps[0][i] = a.r[i];
    __syncthreads();
    double v4[4] = {0., 0., 0., 0.}, v2[2] = {0., 0.};
    for (int64_t i = tid; i < n; i += SMALL_THREADS) {
        cplx ap = small_row(a, i, a.ps[0]);
        a.aps[0][i] = ap;
        cplx rv = a.r[i], bv = a.rhs[i];
        cplx t = cconj_mul(rv, ap), u = cconj_mul(ap, ap);
        v4[0] += t.x; v4[1] += t.y; v4[2] += u.x; v4[3] += u.y;
        v2[0] += bv.x * bv.x + bv.y * bv.y;
        v2[1] += rv.x * rv.x + rv.y * rv.y;
    }
    block_sum_bcast<4>(v4, lds);
    block_sum_bcast<2>(v2, lds);
    const double bnorm2 = v2[0];
    if (tid == 0) a.hist[0] = sqrt(v2[1]) / sqrt(bnorm2);
    cplx num = make_double2(v4[0], v4[1]), den = make_double2(v4[2], v4[3]);
    cplx dens[SMALL_MAX_DIRS];
#pragma unroll
    for (int j = 0; j < SMALL_MAX_DIRS; j++) dens[j] = make_double2(1., 0.);

    const int max_it = a.max_iter > 0 ? a.max_iter : 1;  // do..while
    int iter_count = 0, cur = 0, global = 0, stop = INT_MAX;
    while (global < max_it) {
        global++;
        iter_count++;
        // alpha; x += alpha p; r -= alpha Ap; |r|^2       (src/GCR.h:230-233)
        const cplx alpha = cdiv(num, den);
#pragma unroll
        for (int j = 0; j < SMALL_MAX_DIRS; j++)
            if (j == cur) dens[j] = den;
        double rr1[1] = {0.};
        {
            const cplx *p = a.ps[cur], *ap = a.aps[cur];
            for (int64_t i = tid; i < n; i += SMALL_THREADS) {
                a.x[i] = cadd(a.x[i], cmul(alpha, p[i]));
                cplx rn = csub(a.r[i], cmul(alpha, ap[i]));
                a.r[i] = rn;
                rr1[0] += rn.x * rn.x + rn.y * rn.y;
            }
        }
        block_sum_bcast<1>(rr1, lds);  // (its barriers also publish r for the gather below)
        if (global == max_it) {
            // last iteration the solve can run: the direction the reference goes on to build (src/GCR.h:236-287) is
            // never used — record the step and stop (gcr.hip: finish_step_kernel)
            if (tid == 0 && global < a.hist_cap) a.hist[global] = sqrt(rr1[0]) / sqrt(bnorm2);
            if (!((rr1[0] / bnorm2) > a.tol2)) stop = global;
            break;
        }
        // Ar = A r; <Ar, Aps_j>                             (src/GCR.h:242,258)
        const int lim = a.storage < iter_count ? a.storage : iter_count;
        double vb[2 * SMALL_MAX_DIRS];
#pragma unroll
        for (int j = 0; j < 2 * SMALL_MAX_DIRS; j++) vb[j] = 0.;
        for (int64_t i = tid; i < n; i += SMALL_THREADS) {
            cplx arv = small_row(a, i, a.r);
            a.ar[i] = arv;
#pragma unroll
            for (int j = 0; j < SMALL_MAX_DIRS; j++)
                if (j < lim) {
                    cplx t = cconj_mul(arv, a.aps[j][i]);
                    vb[2 * j] += t.x;
                    vb[2 * j + 1] += t.y;
                }
        }
        block_sum_bcast<2 * SMALL_MAX_DIRS>(vb, lds);
        // history / convergence of this step (src/GCR.h:270-274,288)
        if (tid == 0 && global < a.hist_cap) a.hist[global] = sqrt(rr1[0]) / sqrt(bnorm2);
        if (!((rr1[0] / bnorm2) > a.tol2)) stop = global;
        // new direction into the ring slot (src/GCR.h:257-266,277-287)
        int ic_next = iter_count;
        if (iter_count % a.restart == 0) ic_next = 0;
        const int nxt = ic_next % a.storage;
        cplx beta[SMALL_MAX_DIRS];
#pragma unroll
        for (int j = 0; j < SMALL_MAX_DIRS; j++)
            beta[j] = j < lim ? cdiv(make_double2(vb[2 * j], vb[2 * j + 1]), dens[j]) : make_double2(0., 0.);
        double vn[4] = {0., 0., 0., 0.};
        for (int64_t i = tid; i < n; i += SMALL_THREADS) {
            cplx pc = make_double2(0., 0.), ac = make_double2(0., 0.);
#pragma unroll
            for (int j = 0; j < SMALL_MAX_DIRS; j++)
                if (j < lim) {
                    pc = csub(pc, cmul(beta[j], a.ps[j][i]));
                    ac = csub(ac, cmul(beta[j], a.aps[j][i]));
                }
            cplx rv = a.r[i];
            cplx pn = cadd(rv, pc), an = cadd(a.ar[i], ac);
            a.ps[nxt][i] = pn;
            a.aps[nxt][i] = an;
            cplx t = cconj_mul(rv, an), u = cconj_mul(an, an);
            vn[0] += t.x; vn[1] += t.y; vn[2] += u.x; vn[3] += u.y;
        }
        block_sum_bcast<4>(vn, lds);
        num = make_double2(vn[0], vn[1]);
        den = make_double2(vn[2], vn[3]);
        iter_count = ic_next;
        cur = nxt;
        if (stop != INT_MAX) break;
    }
    if (tid == 0) { a.state[0] = stop; a.state[1] = 0; a.state[2] = global; }
}

static int64_t g_small_limit = -1;
static int64_t g_small_solves = 0;
int64_t gcr_small_solve_count() { return g_small_solves; }
void gcr_small_set_limit(int64_t rows) { g_small_limit = rows; }

// can this solve take the one-workgroup path?
bool gcr_small_eligible(const Op *A, const mgcr_gcr_param &p, int storage, int64_t n) {
    // Default 1024 rows.  One workgroup = one CU: it only wins while the whole system streams through a
    // single CU faster than four kernel launches take (~30 us): measured 141 us/iteration on the 3072-row,
    // 39-nnz/row sample (multi-kernel path: 29.6 us), so the default stays well below that.
    if (g_small_limit < 0) g_small_limit = getenv("MGCR_SMALL_SOLVE_ROWS") ? atoll(getenv("MGCR_SMALL_SOLVE_ROWS")) : 1024;
    if (n > g_small_limit || n < 1) return false;
    {
        const Op *b0 = A->kind == OP_DIRAC ? A->base : A;
        if (b0->kind == OP_CSR && (int64_t)b0->csr.nchunk * b0->csr.L * n > 16 * g_small_limit) return false;  // too many entries for one CU
    }
    if (p.left_precond || p.right_precond || p.profile_spmv) return false;
    if (storage > SMALL_MAX_DIRS) return false;
    const Op *base = A->kind == OP_DIRAC ? A->base : A;
    if (base->kind != OP_CSR || base->dist || base->comm) return false;
    if (base->csr.nrow != base->csr.ncol) return false;
    if (base->csr.pat_mode) return false;  // pattern-dictionary storage (>= 2^15 rows) has no slab to walk
    return true;
}

int gcr_small_run(Op *A, const mgcr_gcr_param &p, int storage, int restart, const cplx *rhs, cplx *x, cplx *r, cplx *ar,
                  cplx *const *ps, cplx *const *aps, double *hist, int hist_cap, int *state) {
    const Op *base = A->kind == OP_DIRAC ? A->base : A;
    const CsrDev &M = base->csr;
    SmallArgs a;
    a.n = M.nrow; a.W = M.W; a.L = M.L; a.nchunk = M.nchunk; a.npad = M.npad;
    a.ell_val = M.ell_val; a.ell_val_re = M.ell_val_re; a.ell_col = M.ell_col;
    a.n_tail_rows = M.n_tail_rows; a.tail_rows = M.tail_rows; a.tail_ptr = M.tail_ptr; a.tail_col = M.tail_col; a.tail_val = M.tail_val;
    a.shift = A->kind == OP_DIRAC ? 1 : 0;
    a.k = A->k;
    a.storage = storage; a.restart = restart; a.max_iter = p.max_iter; a.use_x0 = p.use_x0;
    a.tol2 = p.tol * p.tol;
    a.rhs = rhs; a.x = x; a.r = r; a.ar = ar;
    for (int j = 0; j < SMALL_MAX_DIRS; j++) { a.ps[j] = ps[j < storage ? j : 0]; a.aps[j] = aps[j < storage ? j : 0]; }
    a.hist = hist; a.hist_cap = hist_cap; a.state = state;
    SkipRef outer = get_apply_skip();
    a.outer_skip = outer.p; a.outer_it = outer.it;
    hipLaunchKernelGGL(gcr_small_kernel, dim3(1), dim3(SMALL_THREADS), 0, ctx().stream, a);
    MGCR_HIP(hipGetLastError());
    g_small_solves++;
    return MGCR_OK;
}

}  // namespace mgcr
