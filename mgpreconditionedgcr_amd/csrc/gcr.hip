// Device-resident GCR (restarted / truncated / full) — the reference's GCR<T>::solve
// (src/GCR.h:158-302) re-designed for MI355X:
//
//   * x, r and the stored directions live in HBM for the whole solve; p and Ap are not separate
//     vectors but the ring slot that was written last (the reference copies them, :286-287);
//   * per iteration three kernels instead of ~90 vector passes — one launch per iteration where A r fits
//     the chip's LDS (gcr_stepbuild.hip), one launch per solve for small systems (gcr_resident.hip):
//        xr_update   x += a p, r -= a Ap, |r|^2 partials                         6 V  (3 V when the
//                    x update is deferred to the end of the restart cycle, see xr_update_kernel)
//        apply+dots  Ar = A r and the <Ar, Aps[i]> partials for all stored i      B_matrix + (2+lim) V
//                    (gcr_fused.hip: one kernel for a Sparse / DiracOp stored one thread per row; else
//                    the operator's own apply followed by multidot_kernel, (1+lim) V more)
//        build       p' = r - sum b_i ps[i], Ap' = Ar - sum b_i Aps[i] written straight into the
//                    ring slot, plus <r,Ap'> and <Ap',Ap'> partials and the step's bookkeeping
//                    (history entry, convergence predicate)                        (4+2 lim) V
//     = B_matrix + (9 + 3 lim) V of HBM traffic per iteration in the classic form, and
//   * restart mode (restart <= 16) without the literal preconditioner hooks runs LEAN: inside a restart cycle the
//     directions p_k are never formed.  Only x needs them, and x is updated once per cycle, so the
//     solver keeps what p_k is a combination of — the cycle's first direction P0 and the residuals
//     (or M r, flexible mode) D_1..D_k the later directions were started from — plus the small
//     triangular table of coefficients p_k = t_k P0 + sum_m T_km D_m on the device.  The residual ring
//     costs nothing (xr_update writes r into the slot instead of in place), build shrinks from
//     (4+2 lim) V to (3+lim) V, and the step that closes the cycle applies x += sum_k alpha_k p_k and
//     forms the next P0 from the same streams the classic closing step reads.  r, Ap and every scalar
//     follow the same recurrences with the same bits; x differs by rounding only (MGCR_LEAN=0 selects
//     the classic kernels);
//   * all scalars (alpha, beta_i, norms, the iteration counter, the convergence flag and the
//     residual history) stay on the device.  Reductions are two-stage and deterministic: producers
//     write per-workgroup partials, consumers fold them in a fixed order (reduce.h), so the
//     history is reproducible run to run.  The host only looks at the device-side stop predicate
//     every `check_every` iterations; once it is set every later kernel returns immediately, so x, r
//     and the history are exactly those of the converged step;
//   * the work vectors belong to the solver object and are re-used from solve to solve;
//   * streams touched once per iteration (old direction slots; the matrix in spmv.hip) are accessed
//     non-temporally so that r, Ar and the newest Ap survive in L2 / Infinity Cache between kernels;
//   * used as an Operator (smoother / coarse solver / preconditioner, src/GCR.h:62-68) a solve
//     is enqueued without any host round trip.
//
// Arithmetic follows the reference's order and conjugation (alpha = <r,Ap>/<Ap,Ap> with conj on
// r, beta = <Ar,Aps_i>/<Aps_i,Aps_i> with conj on Ar: SURVEY.md §0 fact 2); the library is built
// with -ffp-contract=off so element-wise results round like the reference's.
#include <climits>
#include <cmath>

#include "internal.h"
#include "reduce.h"
#include "gcr_dev.h"
#include "pw_tail_dev.h"

namespace mgcr {

static int g_lean = -1;
static bool lean_enabled() {
    if (g_lean < 0) g_lean = !(getenv("MGCR_LEAN") && atoi(getenv("MGCR_LEAN")) == 0);
    return g_lean != 0;
}
bool set_lean_enabled(bool on) {
    bool prev = lean_enabled();
    g_lean = on ? 1 : 0;
    return prev;
}

// profile_spmv: hipEvents around the three phases of every iteration of the last profiled solve:
// 0 = alpha/r update (+ preconditioner), 1 = operator apply + beta dot products, 2 = direction build
static double g_prof_phase_ms[3] = {0., 0., 0.};
static int g_prof_iters = 0;
static int g_prof_fused = 0;   // 0: apply and dots separate; 1: one kernel; 2: + the direction build in that launch (gcr_stepbuild.hip); 3: + the next step's residual update; 4: the step that closes a cycle as well; 5: apply + dots in one kernel that also forms the residual update (gcr_fused.hip step_apply_xr*_kernel), build separate
void gcr_last_profile(double *phase_ms_total, int *n_iter, int *fused) {
    for (int k = 0; k < 3; k++) phase_ms_total[k] = g_prof_phase_ms[k];
    *n_iter = g_prof_iters;
    *fused = g_prof_fused;
}

struct GcrState {
    Op *A = nullptr;
    mgcr_gcr_param p{};
    int x0_mode = 1;
    std::vector<const cplx *> r_after;  // r_after[k]: where step k of the last solve left its residual (empty: not tracked)
    bool keep_pending = false;      // solves that only ever assign x (assign_x below) leave that to the caller: gcr_take_pending
    bool has_pending = false;
    PendingX pending{};
    bool defer_residual = false;    // nested: the caller forms the last step's residual itself (ResidualSel: r_prev - alpha ap)
    const cplx *defer_r_prev = nullptr, *defer_ap = nullptr, *defer_alpha = nullptr;
    int defer_it = 0;
    bool discard_residual = false;  // nested solves whose caller only wants x (post-smoother, coarsest solve): see alpha_only_kernel
    bool x_from_zero = false;  // next gcr_run: x0 = 0 and x's content is garbage (gcr_run_from_zero)
    int64_t n = 0;
    int storage = 0, restart = 0;
    std::vector<cplx *> ps, aps;
    cplx *r = nullptr, *ar = nullptr, *z = nullptr, *tmp = nullptr, *accp = nullptr, *accap = nullptr;
    cplx *x0 = nullptr;
    cplx *res_ring = nullptr;   // gcr_resident.hip
    cplx *xbak = nullptr;       // copy of the caller's x while a one-launch path may have to be abandoned (gcr_run)
    int64_t xbak_n = 0;
    DevState *st = nullptr;
    double *partsA = nullptr, *partsR = nullptr, *partsN = nullptr, *partsB = nullptr;
    // |b|^2 between the two smoothers of a V-cycle level (csrc/mg.hip): both get the same right-hand side, the kernels that embed the
    // apply sum |b|^2 over the same rows in the same order — the post-smoother folds the partials the pre-smoother's pass left
    // (partsN_of: the vector they belong to, nullptr = none; partsN_g: how many) instead of streaming b once more
    GcrState *bnorm_src = nullptr;
    const cplx *partsN_of = nullptr;
    int partsN_g = 0;
    cplx *den = nullptr;  // cached <Aps[i],Aps[i]> per slot
    double *hist = nullptr;
    int hist_cap = 0;
    int partsB_dirs = 0;
    // multi-GPU: folded scalars that are all-reduced over the ranks
    double *dA = nullptr;   // [4]
    double *dRB = nullptr;  // [1 + 2 * dirs]: |r|^2, then the beta numerators
    double *dN = nullptr;   // [2]: |b|^2, |r0|^2
    cplx *alphas = nullptr; // [storage]: alpha of the deferred x updates of the current restart cycle
    LeanCoef *lc = nullptr;
    // captured restart cycle (see gcr_run)
    hipGraphExec_t graph_exec = nullptr;
    const cplx *graph_x = nullptr;
    int graph_R = 0;
    int64_t graph_n = 0;
};

constexpr int64_t GRAPH_MAX_ROWS = 1 << 18;

static bool bnorm_reuse_enabled() {
    static const bool on = !(getenv("MGCR_BNORM_REUSE") && atoi(getenv("MGCR_BNORM_REUSE")) == 0);
    return on;
}
static bool fuse_init_enabled() {
    static const bool on = !(getenv("MGCR_FUSE_INIT") && atoi(getenv("MGCR_FUSE_INIT")) == 0);
    return on;
}
static bool stepbuild_xr_enabled() {
    static const bool on = !(getenv("MGCR_STEPBUILD_XR") && atoi(getenv("MGCR_STEPBUILD_XR")) == 0);
    return on;
}
static bool stepbuild_close_enabled() {
    static const bool on = !(getenv("MGCR_STEPBUILD_CLOSE") && atoi(getenv("MGCR_STEPBUILD_CLOSE")) == 0);
    return on;
}
static int g_graph = -1;
static bool graphs_enabled() {
    // opt-in (MGCR_GRAPH=1 / mgcr_set_option("graph_replay")): see the measurements at the capture site in gcr_run
    if (g_graph < 0) g_graph = getenv("MGCR_GRAPH") && atoi(getenv("MGCR_GRAPH")) != 0;
    return g_graph != 0;
}
bool set_graph_enabled(bool on) {
    bool prev = graphs_enabled();
    g_graph = on ? 1 : 0;
    return prev;
}

// ------------------------------------------------------------------------------------------------
// kernels
// ------------------------------------------------------------------------------------------------
#define GRID_STRIDE(i, n) \
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (n); i += (int64_t)gridDim.x * blockDim.x)

__global__ void reset_kernel(DevState *st, const int *inherit, int inherit_it, double tol2) {
    st->stop_at = (inherit && inherit[0] < inherit[1] + inherit_it) ? -1 : INT_MAX;  // an outer solve that is over silences this one
    st->base = 0;
    st->iter = 0;
    st->npend = 0;
    st->bnorm2 = 0.;
    st->rr = 0.;
    st->tol2 = tol2;
}


// partials of |a|^2 -> parts[blk]
__global__ void __launch_bounds__(RED_THREADS) norm_partials_kernel(const cplx *__restrict__ a, int64_t n,
                                                                   double *__restrict__ parts, const DevState *st, int it) {
    __shared__ double lds[17];
    if (st && st->stop_at < st->base + it) return;
    double v[1] = {0.};
    GRID_STRIDE(i, n) {
        cplx t = a[i];
        v[0] += t.x * t.x + t.y * t.y;  // Re(conj(a) a), src/Fields.h:228-235
    }
    const double tot = block_sum_owner<1>(v, lds);
    if (threadIdx.x == 0) parts[blockIdx.x] = tot;
}

// partials of <r,Ap> (conj on r) and <Ap,Ap>  ->  partsA[0..3][blk]
__global__ void __launch_bounds__(RED_THREADS) dot2_partials_kernel(const cplx *__restrict__ r, const cplx *__restrict__ ap,
                                                                   int64_t n, double *__restrict__ parts, const DevState *st, int it) {
    __shared__ double lds[4 * 17];
    if (st->stop_at < st->base + it) return;
    double v[4] = {0., 0., 0., 0.};
    GRID_STRIDE(i, n) {
        cplx a = ap[i];
        cplx t = cconj_mul(r[i], a);
        v[0] += t.x; v[1] += t.y;
        cplx u = cconj_mul(a, a);
        v[2] += u.x; v[3] += u.y;
    }
    const double tot = block_sum_owner<4>(v, lds);
    if (threadIdx.x < 4) parts[threadIdx.x * RED_MAX_BLOCKS + blockIdx.x] = tot;
}

// |r|^2 (written twice: it is |b|^2 as well when r0 = b), <r,Ap> and <Ap,Ap> in one pass; same sums, in the
// same order, as norm_partials_kernel + dot2_partials_kernel
__global__ void __launch_bounds__(RED_THREADS) init3_partials_kernel(const cplx *__restrict__ r, const cplx *__restrict__ ap,
                                                                    int64_t n, double *__restrict__ partsN,
                                                                    double *__restrict__ partsR, double *__restrict__ partsA,
                                                                    const DevState *st, int it) {
    __shared__ double lds[5 * 17];
    if (st->stop_at < st->base + it) return;  // partsN == nullptr: r is not b (use_x0), |b|^2 is taken separately
    double v[5] = {0., 0., 0., 0., 0.};
    GRID_STRIDE(i, n) {
        cplx rv = r[i], a = ap[i];
        v[4] += rv.x * rv.x + rv.y * rv.y;
        cplx t = cconj_mul(rv, a);
        v[0] += t.x; v[1] += t.y;
        cplx u = cconj_mul(a, a);
        v[2] += u.x; v[3] += u.y;
    }
    const double tot = block_sum_owner<5>(v, lds);
    if (threadIdx.x < 4) partsA[threadIdx.x * RED_MAX_BLOCKS + blockIdx.x] = tot;
    if (threadIdx.x == 4) {
        partsR[blockIdx.x] = tot;
        if (partsN) partsN[blockIdx.x] = tot;
    }
}

// step 0 bookkeeping (src/GCR.h:213-216)
__global__ void __launch_bounds__(RED_THREADS) init_kernel(DevState *st, const double *__restrict__ partsN, int nblkN,
                                                           int strideN, const double *__restrict__ partsR, int nblkR,
                                                           int strideR, double *__restrict__ hist) {
    __shared__ double lds[17];
    if (st->stop_at < 0) return;
    double b[1], r[1];
    fold_partials<1>(partsN, nblkN, strideN, b, lds);
    fold_partials<1>(partsR, nblkR, strideR, r, lds);
    if (threadIdx.x == 0) {
        st->bnorm2 = b[0];
        st->rr = r[0];
        hist[0] = sqrt(r[0]) / sqrt(b[0]);
    }
}

// alpha = <r,Ap>/<Ap,Ap>;  x = x + p*alpha;  r = r - Ap*alpha  (src/GCR.h:230-233) + |r|^2 partials.
// DEFER (restart mode): x is not touched here; alpha is parked in alphas[slot] and the pending
// updates x += alpha_j p_j of a whole restart cycle are applied, in iteration order, by the
// build_kernel that closes the cycle (which streams those p_j anyway) or by flush_x_kernel at the
// end of the solve — 3 V less traffic per iteration, bit-identical x.
template <bool DEFER, bool LEAN>
__global__ void __launch_bounds__(RED_THREADS) xr_update_kernel(DevState *__restrict__ st, int it, const double *__restrict__ partsA,
                                                                int nblkA, int strideA, const cplx *__restrict__ p,
                                                                const cplx *__restrict__ ap, cplx *__restrict__ x,
                                                                const cplx *r_in, cplx *r_out, int64_t n, double *__restrict__ partsR,
                                                                cplx *__restrict__ den_slot, cplx *__restrict__ alphas, int slot,
                                                                LeanCoef *__restrict__ lc) {
    __shared__ double lds[4 * 17];
    if (st->stop_at < st->base + it) return;
    // the first trip's operands are requested BEFORE alpha is folded from the partials: one memory round trip of the
    // kernel's prologue (partials -> alpha -> first loads) disappears behind the other
    const int64_t i0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    cplx r0 = make_double2(0., 0.), a0 = r0;
    if (i0 < n) { r0 = r_in[i0]; a0 = ap[i0]; }
    double s[4];
    fold_partials<4>(partsA, nblkA, strideA, s, lds);
    const cplx num = make_double2(s[0], s[1]), den = make_double2(s[2], s[3]);
    const cplx alpha = to_sgpr(cdiv(num, den));
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        *den_slot = den;
        if (DEFER) {
            st->npend = slot + 1;
            if (!LEAN) alphas[slot] = alpha;
        }
    }
    if (DEFER && LEAN && blockIdx.x == 0 && (int)threadIdx.x < LND) lean_pending_update(lc, slot, alpha, (int)threadIdx.x);
    double v[1] = {0.};
    GRID_STRIDE(i, n) {
        if (!DEFER) x[i] = cadd(x[i], cmul(alpha, p[i]));
        const bool first = i == i0;
        cplx rn = csub(first ? r0 : r_in[i], cmul(alpha, first ? a0 : ap[i]));
        r_out[i] = rn;  // LEAN: the residual ring (r_out != r_in inside a cycle); else in place
        v[0] += rn.x * rn.x + rn.y * rn.y;
    }
    const double tot = block_sum_owner<1>(v, lds);
    if (threadIdx.x == 0) partsR[blockIdx.x] = tot;
}

// applies the x updates still pending when the solve ends (never skipped).  The usual counts (a smoother's 2-3 sweeps, up to a
// restart-5 cycle) get their loads issued together; same sums, same order.
template <int NP>
__device__ __forceinline__ void flush_x_rows(const cplx *__restrict__ alphas, const DirPtrs &d0, cplx *__restrict__ x, int64_t n, int assign) {
    cplx al[NP];
#pragma unroll
    for (int j = 0; j < NP; j++) al[j] = to_sgpr(alphas[j]);
    GRID_STRIDE(i, n) {
        cplx pv[NP];
#pragma unroll
        for (int j = 0; j < NP; j++) pv[j] = d0.ps[j][i];
        cplx xv = assign ? make_double2(0., 0.) : x[i];   // assign: x0 = 0 was never materialised (gcr_run: assign_x)
#pragma unroll
        for (int j = 0; j < NP; j++) xv = cadd(xv, cmul(al[j], pv[j]));
        x[i] = xv;
    }
}
__global__ void __launch_bounds__(RED_THREADS) flush_x_kernel(DevState *__restrict__ st, const cplx *__restrict__ alphas, DirPtrs d0,
                                                              cplx *__restrict__ x, int64_t n, int assign) {
    const int np = st->npend;
    if (np <= 0) return;   // (also when an outer solver's stop predicate turned this whole solve into a no-op)
    switch (np) {
        case 1: flush_x_rows<1>(alphas, d0, x, n, assign); return;
        case 2: flush_x_rows<2>(alphas, d0, x, n, assign); return;
        case 3: flush_x_rows<3>(alphas, d0, x, n, assign); return;
        case 4: flush_x_rows<4>(alphas, d0, x, n, assign); return;
        case 5: flush_x_rows<5>(alphas, d0, x, n, assign); return;
        default: break;
    }
    GRID_STRIDE(i, n) {
        cplx xv = assign ? make_double2(0., 0.) : x[i];
        for (int j = 0; j < np && j < LND; j++) xv = cadd(xv, cmul(alphas[j], d0.ps[j][i]));
        x[i] = xv;
    }
}
// r = p0 = rhs at the start of a solve (src/GCR.h:189-190): one read, two writes instead of two copies
__global__ void __launch_bounds__(RED_THREADS) copy2_kernel(cplx *__restrict__ a, cplx *__restrict__ b, const cplx *__restrict__ src, int64_t n,
                                                            const DevState *st) {
    if (st->stop_at < 0) return;
    GRID_STRIDE(i, n) {
        const cplx v = src[i];
        a[i] = v;
        b[i] = v;
    }
}
__global__ void clear_pending_kernel(DevState *st) { st->npend = 0; }
__global__ void advance_kernel(DevState *st, int by) { st->base += by; }

// <Ar, Aps[j]> for j < NDT (conj on Ar, src/GCR.h:258) -> partsB[(base+j)*2 + {0,1}][blk].
// NDT is a template parameter so that the (1 + NDT) * U loads of one trip are issued back to back
// with no branch between them: the kernel is latency-bound otherwise (a wave with a single 1-KiB
// load in flight cannot cover HBM latency, even at 32 waves per CU).
template <int NDT, int U>
__global__ void __launch_bounds__(RED_THREADS, (NDT <= 6 ? 8 : 4)) multidot_kernel(const DevState *__restrict__ st, int it, const cplx *__restrict__ ar,
                                                               DirPtrs d, int base, int64_t n, RowMap rm,
                                                               double *__restrict__ partsB) {
    __shared__ double lds[2 * NDT * 17];
    if (st->stop_at < st->base + it) return;
    double v[2 * NDT];
#pragma unroll
    for (int j = 0; j < 2 * NDT; j++) v[j] = 0.;
    // rows of this workgroup's threads: RowMap (gcr_dev.h), shared with gcr_fused.hip's step_apply_kernel so that
    // either kernel yields the same partial sums
    int64_t first, end, step;
    row_range(rm, (int)blockIdx.x, (int)gridDim.x, n, &first, &end, &step);
    for (int64_t i0 = first; i0 < end; i0 += step * U) {
        cplx a[U], b[U][NDT];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int64_t i = i0 + u * step;
            if (i < end) {
                a[u] = ar[i];
#pragma unroll
                for (int j = 0; j < NDT; j++) b[u][j] = ld_stream<NTS>(d.aps[j] + i);
            } else {
                a[u] = make_double2(0., 0.);
#pragma unroll
                for (int j = 0; j < NDT; j++) b[u][j] = make_double2(0., 0.);
            }
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
#pragma unroll
            for (int j = 0; j < NDT; j++) {
                cplx t = cconj_mul(a[u], b[u][j]);
                v[2 * j] += t.x;
                v[2 * j + 1] += t.y;
            }
        }
    }
    const double mine = block_sum_owner<2 * NDT>(v, lds);
    if (threadIdx.x < 2 * NDT) partsB[(size_t)(2 * base + threadIdx.x) * RED_MAX_BLOCKS + blockIdx.x] = mine;
}

// beta_j = <Ar,Aps_j>/<Aps_j,Aps_j>;  p_corr -= ps_j*beta_j;  Ap_corr -= Aps_j*beta_j  (src/GCR.h:257-262)
// FIRST: accumulators start at 0 (else read from accp/accap); LAST: p' = dir + p_corr, Ap' = Ar + Ap_corr
// are written to the ring slot (src/GCR.h:265-266,286-287) and <r,Ap'>, <Ap',Ap'> partials emitted.
// RDIR: r is a different vector from dir (flexible preconditioning) and has to be loaded as well.
// When `book` is set, workgroup 0 also closes the step (src/GCR.h:270-274,288): folds |r|^2, bumps
// the iteration counter, records the history entry and raises the convergence flag.  The flag only
// takes effect from the next kernel on, so this step's directions are still built, as in the
// reference.
// XUPD (closing step of a restart cycle, NDT == restart): also applies the cycle's deferred
// x += alpha_j ps_j, j ascending = iteration order.
template <int NDT, bool FIRST, bool LAST, bool RDIR, bool XUPD>
__global__ void __launch_bounds__(RED_THREADS) build_kernel(DevState *__restrict__ st, int it, const double *__restrict__ partsB,
                                                            int nblkB, int strideB, int book,
                                                            const double *__restrict__ partsR, int nblkR, int strideR,
                                                            double *__restrict__ hist, int hist_cap,
                                                            const cplx *__restrict__ den, DirPtrs d, int base,
                                                            const cplx *__restrict__ dir, const cplx *__restrict__ r,
                                                            const cplx *__restrict__ ar, cplx *accp, cplx *accap,
                                                            cplx *p_out, cplx *ap_out, int64_t n,
                                                            double *__restrict__ partsA, cplx *__restrict__ x,
                                                            const cplx *__restrict__ alphas) {
    __shared__ double lds[2 * NDT * 17 > 4 * 17 ? 2 * NDT * 17 : 4 * 17];
    __shared__ cplx sbeta[NDT];
    if (st->stop_at < st->base + it) return;
    double s[2 * NDT];
    fold_partials<2 * NDT>(partsB + (size_t)(2 * base) * strideB, nblkB, strideB, s, lds);
    if (book && blockIdx.x == 0) {
        double rr[1];
        fold_partials<1>(partsR, nblkR, strideR, rr, lds);
        if (threadIdx.x == 0) {
            const int git = st->base + it;  // global_count
            st->iter = git;
            st->rr = rr[0];
            if (git < hist_cap) hist[git] = sqrt(rr[0]) / sqrt(st->bnorm2);
            // continue while |r|^2/|b|^2 > tol^2 (src/GCR.h:288); NaN compares false -> stop, like the reference
            if (!((rr[0] / st->bnorm2) > st->tol2)) st->stop_at = git;
            if (XUPD) st->npend = 0;
        }
    }
    if (threadIdx.x < NDT) {
        // s[] is identical in every thread; pick this thread's pair without dynamic register indexing
        cplx num = make_double2(0., 0.);
#pragma unroll
        for (int j = 0; j < NDT; j++)
            if (j == (int)threadIdx.x) num = make_double2(s[2 * j], s[2 * j + 1]);
        sbeta[threadIdx.x] = cdiv(num, den[d.slot[threadIdx.x]]);
    }
    __syncthreads();
    cplx beta[NDT], alf[NDT];
#pragma unroll
    for (int j = 0; j < NDT; j++) {
        beta[j] = to_sgpr(sbeta[j]);
        alf[j] = XUPD ? to_sgpr(alphas[j]) : make_double2(0., 0.);
    }
    double v[4] = {0., 0., 0., 0.};
    GRID_STRIDE(i, n) {
        // all loads of the trip first, no branches in between
        cplx pj[NDT], aj[NDT];
#pragma unroll
        for (int j = 0; j < NDT; j++) {
            pj[j] = ld_stream<NTS>(d.ps[j] + i);
            aj[j] = ld_stream<NTS>(d.aps[j] + i);
        }
        cplx pc = FIRST ? make_double2(0., 0.) : accp[i];
        cplx ac = FIRST ? make_double2(0., 0.) : accap[i];
        cplx dv = make_double2(0., 0.), av = make_double2(0., 0.), rv = make_double2(0., 0.);
        if (LAST) {
            dv = dir[i];
            av = ar[i];
            rv = RDIR ? r[i] : dv;
        }
        if (XUPD) {
            cplx xv = x[i];
#pragma unroll
            for (int j = 0; j < NDT; j++) xv = cadd(xv, cmul(alf[j], pj[j]));
            x[i] = xv;
        }
#pragma unroll
        for (int j = 0; j < NDT; j++) {
            pc = csub(pc, cmul(beta[j], pj[j]));
            ac = csub(ac, cmul(beta[j], aj[j]));
        }
        if (LAST) {
            cplx pn = cadd(dv, pc);
            cplx an = cadd(av, ac);
            st_stream<NTS>(p_out + i, pn);  // p is next read as an "old slot"; Ap' is read by the very next kernel
            ap_out[i] = an;
            cplx t = cconj_mul(rv, an);
            v[0] += t.x; v[1] += t.y;
            cplx u = cconj_mul(an, an);
            v[2] += u.x; v[3] += u.y;
        } else {
            accp[i] = pc;
            accap[i] = ac;
        }
    }
    if (LAST) {
        const double mine = block_sum_owner<4>(v, lds);
        if (threadIdx.x < 4) partsA[threadIdx.x * RED_MAX_BLOCKS + blockIdx.x] = mine;
    }
}

// step bookkeeping shared by the build kernels (src/GCR.h:270-274,288)
__device__ __forceinline__ void close_step(DevState *st, int it, double rr, double *hist, int hist_cap, bool clear_pending) {
    const int git = st->base + it;  // global_count
    st->iter = git;
    st->rr = rr;
    if (git < hist_cap) hist[git] = sqrt(rr) / sqrt(st->bnorm2);
    // continue while |r|^2/|b|^2 > tol^2 (src/GCR.h:288); NaN compares false -> stop, like the reference
    if (!((rr / st->bnorm2) > st->tol2)) st->stop_at = git;
    if (clear_pending) st->npend = 0;
}

// The last step of a nested solve whose caller only wants x (V-cycle post-smoother, coarsest solve): alpha and the
// pending-x bookkeeping of xr_update_kernel<true, true>, without its pass over r and Ap — nobody reads that residual.
__global__ void __launch_bounds__(RED_THREADS) alpha_only_kernel(DevState *st, int it, const double *__restrict__ partsA, int nblkA,
                                                                 int strideA, cplx *__restrict__ den_slot, int slot,
                                                                 LeanCoef *__restrict__ lc, cplx *__restrict__ alpha_out) {
    __shared__ double lds[4 * 17];
    if (st->stop_at < st->base + it) return;
    double s[4];
    fold_partials<4>(partsA, nblkA, strideA, s, lds);
    const cplx num = make_double2(s[0], s[1]), den = make_double2(s[2], s[3]);
    const cplx alpha = to_sgpr(cdiv(num, den));
    if (threadIdx.x == 0) {
        *den_slot = den;
        st->npend = slot + 1;
        st->iter = st->base + it;
        if (alpha_out) *alpha_out = alpha;
    }
    if ((int)threadIdx.x < LND) lean_pending_update(lc, slot, alpha, (int)threadIdx.x);
}

// Bookkeeping of a step without the direction build (src/GCR.h:270-274,288): the LAST iteration a solve can run
// (count == max_iter) still updates x and r and records |r|, but the next search direction the reference goes on to
// build (src/GCR.h:236-287: M r, A r, the beta dots, p, Ap) is never used — gcr_run stops after this kernel.
__global__ void __launch_bounds__(RED_THREADS) finish_step_kernel(DevState *st, int it, const double *__restrict__ partsR, int nblkR,
                                                                  int strideR, double *__restrict__ hist, int hist_cap) {
    __shared__ double lds[17];
    if (st->stop_at < st->base + it) return;
    double rr[1];
    fold_partials<1>(partsR, nblkR, strideR, rr, lds);
    if (threadIdx.x == 0) close_step(st, it, rr[0], hist, hist_cap, false);
}

// LEAN, inside a restart cycle: direction k = NDT is started from D_k (the residual, or M r) and only
// its image is formed:  Ap_k = Ar - sum_{j<k} beta_j Ap_j  (same order as build_kernel), with the
// <r,Ap_k>, <Ap_k,Ap_k> partials; workgroup 0 extends the coefficient table by row k:
//   p_k = D_k - sum_j beta_j p_j   =>   t_k = -sum_j beta_j t_j ,  T_km = -sum_{j>=m} beta_j T_jm ,  T_kk = 1.
// (second launch bound: up to NDT = 5 the kernel fits 64 VGPRs, i.e. two 1024-thread workgroups per CU; left
// alone the compiler takes 84 for NDT = 4 and halves the residency: 48.4 us against 42 us at 128^3)
// PW (multi-GPU, peer-write scalars): the last workgroup folds the <r,Ap'>, <Ap',Ap'> partials and sums them over the ranks (pw_tail_dev.h)
template <int NDT, bool PW = false>
__global__ void __launch_bounds__(RED_THREADS, (NDT <= 5 ? 8 : 4)) build_lean_kernel(DevState *__restrict__ st, int it, const double *__restrict__ partsB,
                                                                 int nblkB, int strideB, const double *__restrict__ partsR, int nblkR,
                                                                 int strideR, double *__restrict__ hist, int hist_cap,
                                                                 const cplx *__restrict__ den, DirPtrs d, const cplx *__restrict__ r,
                                                                 const cplx *__restrict__ ar, cplx *ap_out, int64_t n,
                                                                 double *__restrict__ partsA, LeanCoef *__restrict__ lc, int closing, PwTail pw) {
    __shared__ double lds[2 * NDT * 17 > 4 * 17 ? 2 * NDT * 17 : 4 * 17];
    __shared__ cplx sbeta[NDT];
    if (st->stop_at < st->base + it) {
        if (PW) pw_tail(pw, (int)gridDim.x);   // a stopped solve: nothing to do, but the ranks' exchanges stay in lockstep (gcr_fused.hip step_apply_kernel)
        return;
    }
    double s[2 * NDT];
    fold_partials<2 * NDT>(partsB, nblkB, strideB, s, lds);
    if (blockIdx.x == 0) {
        double rr[1];
        fold_partials<1>(partsR, nblkR, strideR, rr, lds);
        if (threadIdx.x == 0) close_step(st, it, rr[0], hist, hist_cap, closing != 0);
    }
    if (threadIdx.x < NDT) {
        cplx num = make_double2(0., 0.);
#pragma unroll
        for (int j = 0; j < NDT; j++)
            if (j == (int)threadIdx.x) num = make_double2(s[2 * j], s[2 * j + 1]);
        sbeta[threadIdx.x] = cdiv(num, den[d.slot[threadIdx.x]]);
    }
    __syncthreads();
    // closing (restart > 8: the cycle-closing step is close_x_kernel + this kernel with NDT = restart, writing
    // Ap_0' over slot 0 in place): no table row — the next cycle starts a new table
    if constexpr (NDT < LND)   // NDT == LND only ever runs as the closing step of a restart-16 cycle
    if (!closing && blockIdx.x == 0 && (int)threadIdx.x <= NDT) {
        // one thread per column of the new table row (thread 0: t_k, thread k: the unit diagonal), so that the
        // loads of a column are independent and the whole row costs one memory round trip, not k^2 / 2 of them —
        // this sits on the critical path of the short kernels of small systems
        constexpr int k = NDT;
        const int m = threadIdx.x;
        cplx a = make_double2(0., 0.);
        if (m == 0) {
            for (int j = 0; j < k; j++) a = csub(a, cmul(sbeta[j], j == 0 ? make_double2(1., 0.) : lc->t[j]));
            lc->t[k] = a;
        } else if (m < k) {
            for (int j = m; j < k; j++) a = csub(a, cmul(sbeta[j], j == m ? make_double2(1., 0.) : lc->T[j * LND + m]));
            lc->T[k * LND + m] = a;
        } else {
            lc->T[k * LND + k] = make_double2(1., 0.);
        }
    }
    cplx beta[NDT];
#pragma unroll
    for (int j = 0; j < NDT; j++) beta[j] = to_sgpr(sbeta[j]);
    double v[4] = {0., 0., 0., 0.};
    GRID_STRIDE(i, n) {
        cplx aj[NDT];
#pragma unroll
        for (int j = 0; j < NDT; j++) aj[j] = ld_stream<NTS>(d.aps[j] + i);
        const cplx av = ar[i], rv = r[i];
        cplx ac = make_double2(0., 0.);
#pragma unroll
        for (int j = 0; j < NDT; j++) ac = csub(ac, cmul(beta[j], aj[j]));
        const cplx an = cadd(av, ac);
        if (ap_out) ap_out[i] = an;   // (nullptr: only <r,Ap_k>, <Ap_k,Ap_k> are wanted — the solve's last step follows and its caller discards the residual)
        cplx t = cconj_mul(rv, an);
        v[0] += t.x; v[1] += t.y;
        cplx u = cconj_mul(an, an);
        v[2] += u.x; v[3] += u.y;
    }
    const double mine = block_sum_owner<4>(v, lds);
    if (threadIdx.x < 4) partsA[threadIdx.x * RED_MAX_BLOCKS + blockIdx.x] = mine;
    if (PW) pw_tail(pw, (int)gridDim.x);
}

// LEAN, restart > 8: the x / P0 half of the cycle-closing step (the Ap half is build_lean_kernel with
// closing = 1).  Same sums, in the same order, as build_close_kernel; the 2 * NDT coefficients stay in LDS.
// Launched with RED_THREADS / 2 threads per workgroup (256 VGPRs each: 16 direction loads in flight plus the
// coefficients the compiler hoists out of the loop do not fit 128); it produces no partial sums, and
// fold_partials only needs blockDim >= the number of partials (<= RED_MAX_BLOCKS = RED_THREADS / 2).
template <int NDT>
__global__ void __launch_bounds__(RED_THREADS / 2) close_x_kernel(DevState *__restrict__ st, int it, const double *__restrict__ partsB,
                                                              int nblkB, int strideB, const cplx *__restrict__ den, DirPtrs d,
                                                              const cplx *__restrict__ dir, cplx *p_out, int64_t n,
                                                              cplx *__restrict__ x, const LeanCoef *__restrict__ lc) {
    __shared__ double lds[2 * NDT * 17];
    __shared__ cplx sbeta[NDT], scp[NDT], scx[NDT];
    if (st->stop_at < st->base + it) return;
    double s[2 * NDT];
    fold_partials<2 * NDT>(partsB, nblkB, strideB, s, lds);
    if (threadIdx.x < NDT) {
        cplx num = make_double2(0., 0.);
#pragma unroll
        for (int j = 0; j < NDT; j++)
            if (j == (int)threadIdx.x) num = make_double2(s[2 * j], s[2 * j + 1]);
        sbeta[threadIdx.x] = cdiv(num, den[d.slot[threadIdx.x]]);
    }
    __syncthreads();
    if (threadIdx.x < NDT) {
        const int m = threadIdx.x;
        cplx a = make_double2(0., 0.);
        if (m == 0) {
            for (int j = 0; j < NDT; j++) a = cadd(a, cmul(sbeta[j], j == 0 ? make_double2(1., 0.) : lc->t[j]));
        } else {
            for (int j = m; j < NDT; j++) a = cadd(a, cmul(sbeta[j], j == m ? make_double2(1., 0.) : lc->T[j * LND + m]));
        }
        scp[m] = a;
        scx[m] = lc->cx[m];
    }
    __syncthreads();
    GRID_STRIDE(i, n) {
        cplx pj[NDT];
#pragma unroll
        for (int j = 0; j < NDT; j++) pj[j] = ld_stream<NTS>(d.ps[j] + i);
        const cplx dv = dir[i];
        cplx xv = x[i];
#pragma unroll
        for (int j = 0; j < NDT; j++) xv = cadd(xv, cmul(scx[j], pj[j]));
        x[i] = xv;
        cplx pc = make_double2(0., 0.);
#pragma unroll
        for (int j = 0; j < NDT; j++) pc = csub(pc, cmul(scp[j], pj[j]));
        st_stream<NTS>(p_out + i, cadd(dv, pc));
    }
}

// LEAN, the step that closes a restart cycle (NDT = restart = lim).  d.ps[0] is P0, d.ps[m] the D_m;
// `dir` is D_R, what the new direction is started from.  In one pass:
//   x    += cx_0 P0 + sum_m cx_m D_m                        (the cycle's x updates, = sum_k alpha_k p_k)
//   P0'   = dir - cp_0 P0 - sum_m cp_m D_m                   (= dir - sum_j beta_j p_j, src/GCR.h:257-266)
//   Ap_0' = Ar - sum_j beta_j Ap_j                           with cp_m = sum_{j>=m} beta_j T_jm, cp_0 = sum_j beta_j t_j
// written in place over slot 0, plus the <r,Ap'>, <Ap',Ap'> partials and the step's bookkeeping.
template <int NDT, bool RDIR, bool PW = false>
__global__ void __launch_bounds__(RED_THREADS, (NDT <= 2 ? 8 : 4)) build_close_kernel(DevState *__restrict__ st, int it, const double *__restrict__ partsB,
                                                                  int nblkB, int strideB, const double *__restrict__ partsR, int nblkR,
                                                                  int strideR, double *__restrict__ hist, int hist_cap,
                                                                  const cplx *__restrict__ den, DirPtrs d, const cplx *__restrict__ dir,
                                                                  const cplx *__restrict__ r, const cplx *__restrict__ ar, cplx *p_out,
                                                                  cplx *ap_out, int64_t n, double *__restrict__ partsA,
                                                                  cplx *__restrict__ x, const LeanCoef *__restrict__ lc, PwTail pw) {
    __shared__ double lds[2 * NDT * 17 > 4 * 17 ? 2 * NDT * 17 : 4 * 17];
    __shared__ cplx sbeta[NDT], scp[NDT];
    if (st->stop_at < st->base + it) {
        if (PW) pw_tail(pw, (int)gridDim.x);   // (see build_lean_kernel)
        return;
    }
    double s[2 * NDT];
    fold_partials<2 * NDT>(partsB, nblkB, strideB, s, lds);
    if (blockIdx.x == 0) {
        double rr[1];
        fold_partials<1>(partsR, nblkR, strideR, rr, lds);
        if (threadIdx.x == 0) close_step(st, it, rr[0], hist, hist_cap, true);
    }
    if (threadIdx.x < NDT) {
        cplx num = make_double2(0., 0.);
#pragma unroll
        for (int j = 0; j < NDT; j++)
            if (j == (int)threadIdx.x) num = make_double2(s[2 * j], s[2 * j + 1]);
        sbeta[threadIdx.x] = cdiv(num, den[d.slot[threadIdx.x]]);
    }
    __syncthreads();
    if (threadIdx.x < NDT) {
        const int m = threadIdx.x;
        cplx a = make_double2(0., 0.);
        if (m == 0) {
            for (int j = 0; j < NDT; j++) a = cadd(a, cmul(sbeta[j], j == 0 ? make_double2(1., 0.) : lc->t[j]));
        } else {
            for (int j = m; j < NDT; j++) a = cadd(a, cmul(sbeta[j], j == m ? make_double2(1., 0.) : lc->T[j * LND + m]));
        }
        scp[m] = a;
    }
    __syncthreads();
    cplx beta[NDT], cp[NDT], cx[NDT];
#pragma unroll
    for (int j = 0; j < NDT; j++) {
        beta[j] = to_sgpr(sbeta[j]);
        cp[j] = to_sgpr(scp[j]);
        cx[j] = to_sgpr(lc->cx[j]);
    }
    double v[4] = {0., 0., 0., 0.};
    GRID_STRIDE(i, n) {
        cplx pj[NDT], aj[NDT];
#pragma unroll
        for (int j = 0; j < NDT; j++) {
            pj[j] = ld_stream<NTS>(d.ps[j] + i);
            aj[j] = ld_stream<NTS>(d.aps[j] + i);
        }
        const cplx dv = dir[i], av = ar[i];
        const cplx rv = RDIR ? r[i] : dv;
        cplx xv = x[i];
#pragma unroll
        for (int j = 0; j < NDT; j++) xv = cadd(xv, cmul(cx[j], pj[j]));
        x[i] = xv;
        cplx pc = make_double2(0., 0.), ac = make_double2(0., 0.);
#pragma unroll
        for (int j = 0; j < NDT; j++) {
            pc = csub(pc, cmul(cp[j], pj[j]));
            ac = csub(ac, cmul(beta[j], aj[j]));
        }
        const cplx pn = cadd(dv, pc), an = cadd(av, ac);
        st_stream<NTS>(p_out + i, pn);
        ap_out[i] = an;
        cplx t = cconj_mul(rv, an);
        v[0] += t.x; v[1] += t.y;
        cplx u = cconj_mul(an, an);
        v[2] += u.x; v[3] += u.y;
    }
    const double mine = block_sum_owner<4>(v, lds);
    if (threadIdx.x < 4) partsA[threadIdx.x * RED_MAX_BLOCKS + blockIdx.x] = mine;
    if (PW) pw_tail(pw, (int)gridDim.x);
}

// ------------------------------------------------------------------------------------------------
// host driver
// ------------------------------------------------------------------------------------------------
template <typename T>
static int dalloc(T **p, size_t count) {
    hipError_t e = hipMalloc((void **)p, sizeof(T) * (count ? count : 1));
    if (e != hipSuccess) {
        set_error("hipMalloc of %zu bytes failed: %s", sizeof(T) * count, hipGetErrorString(e));
        return MGCR_ERR_ALLOC;
    }
    return MGCR_OK;
}

int op_apply_raw(Op *op, const cplx *x, cplx *y, int64_t n) {
    MGCR_CHECK(op, MGCR_ERR_INVALID, "null operator");
    switch (op->kind) {
        case OP_CSR:
            MGCR_CHECK((op->dist ? op->csr.nrow : op->csr.ncol) == n, MGCR_ERR_INVALID, "Sparse matrix dimension does not match Field dimension!");
            return csr_apply(op->csr, x, y, false, make_double2(0., 0.), op->dist);
        case OP_DIRAC:
            MGCR_CHECK(op->base->csr.nrow == n && (op->base->dist || op->base->csr.ncol == n), MGCR_ERR_INVALID,
                       "DiracOp needs a square matrix matching the Field dimension");
            // assertm(k != 0., ...) src/Operator.h:571
            MGCR_CHECK(op->k.x != 0. || op->k.y != 0., MGCR_ERR_INVALID, "No k value supplied for Dirac Operator!");
            return csr_apply(op->base->csr, x, y, true, op->k, op->base->dist);
        case OP_BCSR:
            if (op->dist) {   // row block of a distributed HierarchicalSparse: halo exchange, then the blocks read x or the halo segment
                MGCR_CHECK((int64_t)op->bcsr.nbrow * op->bcsr.bs == n, MGCR_ERR_INVALID, "Sparse matrix dimension does not match Field dimension!");
                MGCR_TRY(dist_halo_begin(op->dist, x));
                MGCR_TRY(dist_halo_end(op->dist));
                return bcsr_apply(op->bcsr, x, y, dist_halo_ptr(op->dist), op->bcsr.nbrow);
            }
            MGCR_CHECK((int64_t)op->bcsr.nbcol * op->bcsr.bs == n, MGCR_ERR_INVALID,
                       "Sparse matrix dimension does not match Field dimension!");
            return bcsr_apply(op->bcsr, x, y);
        case OP_GCR:
            return gcr_apply_as_operator(op->gcr, x, y);
        case OP_MG:
            return mg_apply(op->mg, x, y);
        default:
            set_error("operator kind %d cannot be applied", (int)op->kind);
            return MGCR_ERR_UNSUPPORTED;
    }
}

// r = b - op(x): one pass for a Sparse (the SpMV epilogue takes b), apply + subtract otherwise
__global__ void __launch_bounds__(RED_THREADS) resid_sub_kernel(cplx *r, const cplx *__restrict__ b, int64_t n,
                                                                const int *__restrict__ skip, int skip_it) {
    if (skip && skip[0] < skip[1] + skip_it) return;
    GRID_STRIDE(i, n) r[i] = csub(b[i], r[i]);
}
int op_residual_raw(Op *op, const cplx *x, const cplx *b, cplx *r, int64_t n) {
    MGCR_CHECK(op, MGCR_ERR_INVALID, "null operator");
    if (op->kind == OP_CSR && op->csr.nrow == n && (op->dist ? true : op->csr.ncol == n) && b != r)
        return csr_apply(op->csr, x, r, true, make_double2(1., 0.), op->dist, b);
    MGCR_TRY(op_apply_raw(op, x, r, n));
    const SkipRef sk = get_apply_skip();
    hipLaunchKernelGGL(resid_sub_kernel, dim3(red_grid(n)), dim3(RED_THREADS), 0, ctx().stream, r, b, n, sk.p, sk.it);
    MGCR_HIP(hipGetLastError());
    return MGCR_OK;
}

static void gcr_free_vectors(GcrState *s) {
    if (s->graph_exec) { hipGraphExecDestroy(s->graph_exec); s->graph_exec = nullptr; }
    for (cplx *p : s->ps) hipFree(p);
    for (cplx *p : s->aps) hipFree(p);
    s->ps.clear(); s->aps.clear();
    hipFree(s->r); hipFree(s->ar); hipFree(s->z); hipFree(s->tmp); hipFree(s->accp); hipFree(s->accap);
    hipFree(s->den); hipFree(s->hist); hipFree(s->partsB); hipFree(s->dRB); hipFree(s->alphas); hipFree(s->res_ring);
    s->res_ring = nullptr;
    s->r = s->ar = s->z = s->tmp = s->accp = s->accap = nullptr;
    s->den = nullptr; s->hist = nullptr; s->partsB = nullptr; s->dRB = nullptr; s->alphas = nullptr;
    s->n = 0; s->partsB_dirs = 0; s->hist_cap = 0;
}

void gcr_state_destroy(GcrState *s) {
    if (!s) return;
    if (ctx().ready) hipStreamSynchronize(ctx().stream);
    gcr_free_vectors(s);
    hipFree(s->xbak);
    hipFree(s->x0); hipFree(s->st); hipFree(s->partsA); hipFree(s->partsR); hipFree(s->partsN);
    hipFree(s->dA); hipFree(s->dN); hipFree(s->lc);
    delete s;
}

int gcr_state_create(Op *A, const mgcr_gcr_param *p, int x0_mode, GcrState **out) {
    MGCR_CHECK(p, MGCR_ERR_INVALID, "null GCR parameters");
    // assertm(param->truncation==0 || param->restart==0, ...) src/GCR.h:165
    MGCR_CHECK(p->truncation == 0 || p->restart == 0, MGCR_ERR_INVALID, "Do not support concurrent restarting and truncation.");
    MGCR_CHECK(p->truncation >= 0 && p->restart >= 0 && p->max_iter >= 0, MGCR_ERR_INVALID, "negative GCR parameter");
    GcrState *s = new GcrState();
    s->A = A;
    s->p = *p;
    s->x0_mode = x0_mode;
    int rc = dalloc(&s->st, 1);
    if (rc == MGCR_OK) rc = dalloc(&s->partsA, 4 * RED_MAX_BLOCKS);
    if (rc == MGCR_OK) rc = dalloc(&s->partsR, RED_MAX_BLOCKS);
    if (rc == MGCR_OK) rc = dalloc(&s->partsN, RED_MAX_BLOCKS);
    if (rc == MGCR_OK) rc = dalloc(&s->dA, 4);
    if (rc == MGCR_OK) rc = dalloc(&s->dN, 2);
    if (rc == MGCR_OK) rc = dalloc(&s->lc, 1);
    if (rc != MGCR_OK) { gcr_state_destroy(s); return rc; }
    *out = s;
    return MGCR_OK;
}

void gcr_state_set_use_x0(GcrState *s, bool use_x0) { s->p.use_x0 = use_x0 ? 1 : 0; }

int gcr_state_set_param(GcrState *s, const mgcr_gcr_param *p) {
    MGCR_CHECK(p->truncation == 0 || p->restart == 0, MGCR_ERR_INVALID, "Do not support concurrent restarting and truncation.");
    MGCR_CHECK(p->truncation >= 0 && p->restart >= 0 && p->max_iter >= 0, MGCR_ERR_INVALID, "negative GCR parameter");
    s->p = *p;
    return MGCR_OK;
}

int gcr_state_set_operator(GcrState *s, Op *A) {
    s->A = A;
    return MGCR_OK;
}

int gcr_state_set_x0(GcrState *s, const cplx *x0, int64_t n) {
    if (s->x0) { hipStreamSynchronize(ctx().stream); hipFree(s->x0); s->x0 = nullptr; }
    if (!x0) return MGCR_OK;
    MGCR_TRY(dalloc(&s->x0, (size_t)n));
    MGCR_TRY(k_copy(s->x0, x0, n));
    return MGCR_OK;
}

static int ensure_slot(GcrState *s, int slot) {
    while ((int)s->ps.size() <= slot) {
        cplx *a = nullptr, *b = nullptr;
        MGCR_TRY(dalloc(&a, (size_t)s->n));
        s->ps.push_back(a);
        MGCR_TRY(dalloc(&b, (size_t)s->n));
        s->aps.push_back(b);
    }
    return MGCR_OK;
}

static int gcr_prepare(GcrState *s, int64_t n) {
    const mgcr_gcr_param &p = s->p;
    // mode selection, src/GCR.h:171-185
    int storage = p.max_iter, restart;
    if (p.truncation != 0) storage = p.truncation;
    if (p.restart != 0) { restart = p.restart; storage = restart; } else restart = p.max_iter;
    // a restart cycle longer than the whole solve never closes: iteration k writes slot k <= max_iter,
    // so max_iter + 1 slots behave exactly like `restart` slots (smoothers: 2 sweeps of GCR(10))
    if (p.restart != 0 && p.max_iter >= 1 && p.max_iter + 1 < storage) storage = p.max_iter + 1;
    if (storage < 1) storage = 1;
    if (restart < 1) restart = 1;
    bool precond = p.left_precond || p.right_precond;
    if (s->n != n || s->storage != storage) {
        if (ctx().ready) hipStreamSynchronize(ctx().stream);
        gcr_free_vectors(s);
        s->n = n;
        s->storage = storage;
        MGCR_TRY(dalloc(&s->r, (size_t)n));
        MGCR_TRY(dalloc(&s->ar, (size_t)n));
        MGCR_TRY(dalloc(&s->den, (size_t)storage));
        MGCR_TRY(dalloc(&s->alphas, (size_t)storage));
        int bd = storage < ND ? ND : (storage + ND - 1) / ND * ND;
        MGCR_TRY(dalloc(&s->partsB, (size_t)2 * bd * RED_MAX_BLOCKS));
        MGCR_HIP(hipMemsetAsync(s->partsB, 0, sizeof(double) * (size_t)2 * bd * RED_MAX_BLOCKS, ctx().stream));
        s->partsB_dirs = bd;
        MGCR_TRY(dalloc(&s->dRB, (size_t)1 + 2 * bd));
        MGCR_HIP(hipMemsetAsync(s->dRB, 0, sizeof(double) * ((size_t)1 + 2 * bd), ctx().stream));
    }
    s->restart = restart;
    {   // history buffer: grows with max_iter, everything else is kept across solves
        int cap = (p.max_iter > 0 ? p.max_iter : 1) + 1;
        if (cap > s->hist_cap) {
            if (s->hist) { hipStreamSynchronize(ctx().stream); hipFree(s->hist); s->hist = nullptr; }
            MGCR_TRY(dalloc(&s->hist, (size_t)cap));
            s->hist_cap = cap;
        }
    }
    if (precond && !s->tmp) MGCR_TRY(dalloc(&s->tmp, (size_t)n));
    if (p.flexible && p.right_precond && !s->z) MGCR_TRY(dalloc(&s->z, (size_t)n));
    if (storage > ND && !s->accp) {
        MGCR_TRY(dalloc(&s->accp, (size_t)n));
        MGCR_TRY(dalloc(&s->accap, (size_t)n));
    }
    // restart / truncated modes: all slots up front; full mode grows on demand
    int want = (p.truncation != 0 || p.restart != 0) ? storage : 1;
    MGCR_TRY(ensure_slot(s, want - 1));
    return MGCR_OK;
}

#define KLAUNCH(kernel, grid, ...)                                                                 \
    do {                                                                                           \
        hipLaunchKernelGGL(kernel, dim3(grid), dim3(RED_THREADS), 0, ctx().stream, __VA_ARGS__);   \
        MGCR_HIP(hipGetLastError());                                                               \
    } while (0)

struct SkipGuard {
    SkipRef prev;
    explicit SkipGuard(SkipRef f) : prev(get_apply_skip()) { set_apply_skip(f); }
    ~SkipGuard() { set_apply_skip(prev); }
};

static int launch_multidot(int g, int nd, const DevState *st, int it, const cplx *ar, const DirPtrs &d, int base, int64_t n, double *partsB,
                           const RowMap &rm) {
#define MD(NDT, U) KLAUNCH((multidot_kernel<NDT, U>), g, st, it, ar, d, base, n, rm, partsB)
    switch (nd) {
        case 1: MD(1, 2); break;
        case 2: MD(2, 2); break;
        case 3: MD(3, 1); break;
        case 4: MD(4, 1); break;
        case 5: MD(5, 1); break;
        case 6: MD(6, 1); break;
        case 7: MD(7, 1); break;
        default: MD(8, 1); break;
    }
#undef MD
    return MGCR_OK;
}

struct RedRef {  // where a consumer finds a reduction: slab of per-workgroup partials, or folded + all-reduced scalars
    const double *p;
    int nblk, stride;
};

struct BuildArgs {
    int g, nd;
    bool first, last, rdir, xupd;
    DevState *st;
    int it;
    RedRef B;
    int book;
    RedRef R;
    double *hist;
    int hist_cap;
    const cplx *den;
    DirPtrs d;
    int base;
    const cplx *dir, *r, *ar;
    cplx *accp, *accap, *p_out, *ap_out;
    int64_t n;
    double *partsA;
    cplx *x;
    const cplx *alphas;
};

template <int NDT>
static int launch_build_n(const BuildArgs &a) {
    const int g = a.g;
#define BK(F, L, R_, X_)                                                                                                       \
    KLAUNCH((build_kernel<NDT, F, L, R_, X_>), g, a.st, a.it, a.B.p, a.B.nblk, a.B.stride, a.book, a.R.p, a.R.nblk, a.R.stride, \
            a.hist, a.hist_cap, a.den, a.d, a.base, a.dir, a.r, a.ar, a.accp, a.accap, a.p_out, a.ap_out, a.n, a.partsA, a.x, a.alphas)
    if (a.first && a.last) {
        if (a.xupd) { if (a.rdir) BK(true, true, true, true); else BK(true, true, false, true); }
        else { if (a.rdir) BK(true, true, true, false); else BK(true, true, false, false); }
    } else if (a.first) BK(true, false, false, false);
    else if (a.last) { if (a.rdir) BK(false, true, true, false); else BK(false, true, false, false); }
    else BK(false, false, false, false);
#undef BK
    return MGCR_OK;
}

static int launch_build(const BuildArgs &a) {
    switch (a.nd) {
        case 1: return launch_build_n<1>(a);
        case 2: return launch_build_n<2>(a);
        case 3: return launch_build_n<3>(a);
        case 4: return launch_build_n<4>(a);
        case 5: return launch_build_n<5>(a);
        case 6: return launch_build_n<6>(a);
        case 7: return launch_build_n<7>(a);
        default: return launch_build_n<8>(a);
    }
}

struct LeanArgs {
    int g, nd;
    bool rdir;
    DevState *st;
    int it;
    RedRef B, R;
    double *hist;
    int hist_cap;
    const cplx *den;
    DirPtrs d;
    const cplx *dir, *r, *ar;
    cplx *p_out, *ap_out;
    int64_t n;
    double *partsA;
    cplx *x;
    LeanCoef *lc;
    const PwTail *pw = nullptr;   // multi-GPU: the kernel's last workgroup folds partsA and sums it over the ranks (up to 8 directions)
};

static int launch_build_lean(const LeanArgs &a, int closing) {
#define BL(NDT)                                                                                                               \
    KLAUNCH((build_lean_kernel<NDT>), a.g, a.st, a.it, a.B.p, a.B.nblk, a.B.stride, a.R.p, a.R.nblk, a.R.stride, a.hist,      \
            a.hist_cap, a.den, a.d, a.r, a.ar, a.ap_out, a.n, a.partsA, a.lc, closing, PwTail{})
#define BLP(NDT)                                                                                                              \
    KLAUNCH((build_lean_kernel<NDT, true>), a.g, a.st, a.it, a.B.p, a.B.nblk, a.B.stride, a.R.p, a.R.nblk, a.R.stride, a.hist, \
            a.hist_cap, a.den, a.d, a.r, a.ar, a.ap_out, a.n, a.partsA, a.lc, closing, *a.pw)
    if (a.pw) {
        switch (a.nd) {
            case 1: BLP(1); break;
            case 2: BLP(2); break;
            case 3: BLP(3); break;
            case 4: BLP(4); break;
            case 5: BLP(5); break;
            case 6: BLP(6); break;
            case 7: BLP(7); break;
            default: BLP(8); break;
        }
        return MGCR_OK;
    }
    switch (a.nd) {
        case 1: BL(1); break;
        case 2: BL(2); break;
        case 3: BL(3); break;
        case 4: BL(4); break;
        case 5: BL(5); break;
        case 6: BL(6); break;
        case 7: BL(7); break;
        case 8: BL(8); break;
        case 9: BL(9); break;
        case 10: BL(10); break;
        case 11: BL(11); break;
        case 12: BL(12); break;
        case 13: BL(13); break;
        case 14: BL(14); break;
        case 15: BL(15); break;
        default: BL(16); break;
    }
#undef BL
#undef BLP
    return MGCR_OK;
}

static int launch_close_x(const LeanArgs &a) {
#define CX(NDT)                                                                                                               \
    do {                                                                                                                      \
        hipLaunchKernelGGL((close_x_kernel<NDT>), dim3(2 * a.g), dim3(RED_THREADS / 2), 0, ctx().stream, a.st, a.it, a.B.p,   \
                           a.B.nblk, a.B.stride, a.den, a.d, a.dir, a.p_out, a.n, a.x, (const LeanCoef *)a.lc);               \
        MGCR_HIP(hipGetLastError());                                                                                          \
    } while (0)
    switch (a.nd) {
        case 9: CX(9); break;
        case 10: CX(10); break;
        case 11: CX(11); break;
        case 12: CX(12); break;
        case 13: CX(13); break;
        case 14: CX(14); break;
        case 15: CX(15); break;
        default: CX(16); break;
    }
#undef CX
    return MGCR_OK;
}

static int launch_build_close(const LeanArgs &a) {
#define BC(NDT)                                                                                                               \
    do {                                                                                                                      \
        if (a.pw && a.rdir)                                                                                                   \
            KLAUNCH((build_close_kernel<NDT, true, true>), a.g, a.st, a.it, a.B.p, a.B.nblk, a.B.stride, a.R.p, a.R.nblk,     \
                    a.R.stride, a.hist, a.hist_cap, a.den, a.d, a.dir, a.r, a.ar, a.p_out, a.ap_out, a.n, a.partsA, a.x,      \
                    (const LeanCoef *)a.lc, *a.pw);                                                                           \
        else if (a.pw)                                                                                                        \
            KLAUNCH((build_close_kernel<NDT, false, true>), a.g, a.st, a.it, a.B.p, a.B.nblk, a.B.stride, a.R.p, a.R.nblk,    \
                    a.R.stride, a.hist, a.hist_cap, a.den, a.d, a.dir, a.r, a.ar, a.p_out, a.ap_out, a.n, a.partsA, a.x,      \
                    (const LeanCoef *)a.lc, *a.pw);                                                                           \
        else if (a.rdir)                                                                                                      \
            KLAUNCH((build_close_kernel<NDT, true>), a.g, a.st, a.it, a.B.p, a.B.nblk, a.B.stride, a.R.p, a.R.nblk,           \
                    a.R.stride, a.hist, a.hist_cap, a.den, a.d, a.dir, a.r, a.ar, a.p_out, a.ap_out, a.n, a.partsA, a.x,      \
                    (const LeanCoef *)a.lc, PwTail{});                                                                        \
        else                                                                                                                  \
            KLAUNCH((build_close_kernel<NDT, false>), a.g, a.st, a.it, a.B.p, a.B.nblk, a.B.stride, a.R.p, a.R.nblk,          \
                    a.R.stride, a.hist, a.hist_cap, a.den, a.d, a.dir, a.r, a.ar, a.p_out, a.ap_out, a.n, a.partsA, a.x,      \
                    (const LeanCoef *)a.lc, PwTail{});                                                                        \
    } while (0)
    switch (a.nd) {
        case 1: BC(1); break;
        case 2: BC(2); break;
        case 3: BC(3); break;
        case 4: BC(4); break;
        case 5: BC(5); break;
        case 6: BC(6); break;
        case 7: BC(7); break;
        default: BC(8); break;
    }
#undef BC
    return MGCR_OK;
}

// read back iteration count / history once the solve has been enqueued (not for nested solves)
static int gcr_finish(GcrState *s, double *hist, int hist_cap, int *n_iter, int *converged) {
    Context &c = ctx();
    const mgcr_gcr_param &p = s->p;
    MGCR_HIP(hipMemcpyAsync(c.h_mail, s->st, sizeof(DevState), hipMemcpyDeviceToHost, c.stream));
    MGCR_HIP(hipStreamSynchronize(c.stream));
    DevState hs = *(const DevState *)c.h_mail;
    int it = hs.iter;
    if (n_iter) *n_iter = it;
    if (converged) *converged = (it == p.max_iter) ? 0 : 1;  // src/GCR.h:294-298
    std::vector<double> hh((size_t)it + 1);
    MGCR_HIP(hipMemcpy(hh.data(), s->hist, sizeof(double) * ((size_t)it + 1), hipMemcpyDeviceToHost));
    if (hist)
        for (int i = 0; i <= it && i < hist_cap; i++) hist[i] = hh[i];
    if (p.verbose) {  // src/GCR.h:213-216,270-274,293-300
        for (int i = 0; i <= it; i++) printf("Step %d residual norm = %.10e\n", i, hh[i]);
        if (it == p.max_iter)
            printf("GCR did not converge after %d steps! Residual norm = %.10e\n", p.max_iter, hh[it]);
        else
            printf("GCR converged after %d steps. Residual norm=%.10e\n", it, hh[it]);
    }
    return MGCR_OK;
}

static int gcr_run_once(GcrState *s, const cplx *rhs, cplx *x, bool nested, double *hist, int hist_cap, int *n_iter, int *converged) {
    Context &c = ctx();
    MGCR_CHECK(s->A, MGCR_ERR_INVALID, "GCR has no operator (call initialise / mgcr_gcr_set_operator first)");
    const int64_t n = s->A->dim;
    MGCR_TRY(gcr_prepare(s, n));
    const mgcr_gcr_param &p = s->p;
    const int g = red_grid(n);
    const bool flex = p.flexible && p.right_precond;
    const SkipRef outer = get_apply_skip();  // outer solver's predicate: if that solve is over, this one is a no-op too
    s->has_pending = false;
    s->defer_it = 0;
    s->partsN_of = nullptr;   // (set again below by the path that leaves |b|^2 partials behind)
    const bool from_zero = s->x_from_zero;   // gcr_run_from_zero: x has to be zeroed here, unless the solve only ever ASSIGNS x
    s->x_from_zero = false;

    // small systems: the whole solve in one launch of one workgroup (gcr_small.hip)
    if (gcr_small_eligible(s->A, p, s->storage, n)) {
        if (from_zero) MGCR_TRY(k_zero_apply(x, n));
        MGCR_TRY(ensure_slot(s, s->storage - 1));
        MGCR_TRY(gcr_small_run(s->A, p, s->storage, s->restart, rhs, x, s->r, s->ar, s->ps.data(), s->aps.data(), s->hist,
                               s->hist_cap, &s->st->stop_at));
        s->r_after.assign((size_t)(p.max_iter > 0 ? p.max_iter : 1) + 1, (const cplx *)s->r);
        if (nested) return MGCR_OK;
        return gcr_finish(s, hist, hist_cap, n_iter, converged);
    }

    // systems of at most one row per thread of the chip: the whole solve in one launch, vectors in registers (gcr_resident.hip)
    {
        const bool lean0 = p.restart != 0 && s->storage <= LND && lean_enabled() && !p.left_precond && (!p.right_precond || flex);
        const bool handoff = nested && (s->keep_pending || s->defer_residual);   // callers that take x or r in pieces (V-cycle pre-smoother)
        if (!graphs_enabled() && gcr_resident_eligible(s->A, p, s->storage, s->restart, n, lean0, handoff)) {
            if (!s->res_ring) MGCR_TRY(dalloc(&s->res_ring, (size_t)11 * n));   // the residual ring + P0 (freed with the other vectors)
            MGCR_TRY(gcr_resident_run(s->A, p, s->storage, s->restart, rhs, x, from_zero, nested && s->discard_residual, s->st, s->hist,
                                      s->hist_cap, s->res_ring, outer));
            s->r_after.clear();
            if (nested) return MGCR_OK;
            MGCR_TRY(gcr_finish(s, hist, hist_cap, n_iter, converged));
            return resident_check(true);
        }
    }
    hipLaunchKernelGGL(reset_kernel, dim3(1), dim3(1), 0, c.stream, s->st, outer.p, outer.it, p.tol * p.tol);
    MGCR_HIP(hipGetLastError());
    SkipGuard guard(SkipRef{&s->st->stop_at, 0});

    // restart mode with all slots handled by one build launch: defer the x updates of a cycle ...
    // ... and without the literal hooks (which replace r itself) the cycle runs lean (up to LND slots): see the
    // file header
    const bool lean = p.restart != 0 && s->storage <= LND && lean_enabled() && !p.left_precond && (!p.right_precond || flex);
    const bool defer = lean || (p.restart != 0 && s->storage <= ND);
    // A lean solve that ends before its first restart cycle closes (smoothers: 2 sweeps of GCR(10)) never
    // overwrites its first direction and never updates r in place (the residual ring takes the updates): P0
    // simply IS r0 — no copy — and from x0 = 0 r0 IS rhs, in which case |b|^2 = |r0|^2 comes out of the pass
    // that takes <r,Ap> and <Ap,Ap>.
    const bool alias_p0 = lean && !flex && p.max_iter >= 1 && p.max_iter < p.restart;
    const bool alias0 = alias_p0 && !p.use_x0;
    const cplx *p0 = alias0 ? rhs : alias_p0 ? (const cplx *)s->r : (const cplx *)s->ps[0];
    // From x0 = 0, a solve that never closes a restart cycle touches x exactly once: flush_x_kernel at the end adds the
    // pending updates.  That kernel can just as well WRITE x = sum alpha_j p_j: no zeroing pass before, no read of x then.
    const bool assign_x = from_zero && alias_p0 && !p.use_x0;
    if (from_zero && !assign_x) MGCR_TRY(k_zero_apply(x, n));
    // operator apply fused with the beta dot products: Sparse / DiracOp in a one-thread-per-row layout
    // row -> workgroup map of the dot-product kernels (gcr_dev.h): depends on how far the operator's rows reach
    int64_t reach = 0;
    {
        const Op *b0 = s->A->kind == OP_DIRAC ? s->A->base : s->A;
        if (b0 && b0->kind == OP_CSR) reach = b0->csr.reach;
    }
    const RowMap rmap = make_row_map(n, g, reach);
    bool fuse_ok = false;
    if ((s->A->kind == OP_CSR || s->A->kind == OP_DIRAC) && !p.left_precond) {
        const Op *b0 = s->A->kind == OP_DIRAC ? s->A->base : s->A;
        fuse_ok = b0->kind == OP_CSR && csr_fusable(b0->csr, b0->dist) && b0->csr.nrow == n;
    }
    bool xr_fuse = false;   // set below once lean / flex / multi are known
    // The plain start of a solve on a fusable Sparse / DiracOp (r0 = p0 = rhs: no x0, no preconditioner): r and p0 are
    // written by ONE kernel from one read of rhs, and Ap_0 = A rhs comes out of the pass that also takes <r0,Ap0>,
    // <Ap0,Ap0> and |r0|^2 = |b|^2 (gcr_fused.hip init_apply_kernel) — 3 launches instead of 6 and 6 V less traffic
    // per solve; same sums in the same order as the separate kernels (test_fused_apply_and_dots_same_bits).  Worth
    // ~45 us per solve, i.e. 2-3 % of a 20-iteration solve at 128^3.
    const bool fuse_start = fuse_ok && !alias_p0 && !flex && !p.use_x0 && !p.left_precond && !p.right_precond && fuse_init_enabled();
    // r = rhs (src/GCR.h:189); the reference ignores x0 here unless use_x0 is requested
    if (p.use_x0) {
        MGCR_TRY(op_residual_raw(s->A, x, rhs, s->r, n));
    } else if (fuse_start) {
        hipLaunchKernelGGL(copy2_kernel, dim3(g), dim3(RED_THREADS), 0, c.stream, s->r, s->ps[0], rhs, n, (const DevState *)s->st);
        MGCR_HIP(hipGetLastError());
    } else if (!alias0) {
        MGCR_TRY(k_copy(s->r, rhs, n));
    }
    // p = r (or M r); Ap = A p; both go straight into slot 0 (src/GCR.h:190-191,208-211)
    if (flex) {
        MGCR_TRY(op_apply_raw((Op *)p.right_precond, s->r, s->z, n));
        MGCR_TRY(k_copy(s->ps[0], s->z, n));
    } else if (!alias_p0 && !fuse_start) {
        MGCR_TRY(k_copy(s->ps[0], s->r, n));
    }
    // step 0 of a smoother-like solve on a fusable Sparse / DiracOp: Ap_0 and its dot products in one pass (gcr_fused.hip)
    const bool fuse_init = fuse_ok && alias_p0 && fuse_init_enabled();
    if (!fuse_init && !fuse_start) MGCR_TRY(op_apply_raw(s->A, p0, s->aps[0], n));
    if (!flex) {  // literal hooks, src/GCR.h:197-204 (after p and Ap were formed)
        if (p.right_precond) { MGCR_TRY(op_apply_raw((Op *)p.right_precond, s->r, s->tmp, n)); std::swap(s->r, s->tmp); }
        if (p.left_precond) { MGCR_TRY(op_apply_raw((Op *)p.left_precond, s->r, s->tmp, n)); std::swap(s->r, s->tmp); }
    }
    // Reductions.  Single GPU: consumers fold the producers' per-workgroup partials themselves.
    // Multi-GPU: a one-workgroup fold writes the local sums, RCCL all-reduces them in place on the
    // compute stream, and consumers read the global scalars (stride 1, one "partial").
    Comm *comm = s->A->kind == OP_DIRAC ? s->A->base->comm : s->A->comm;
    const bool multi = comm_collectives(comm);
    if (fuse_ok && lean && !flex && !multi && s->restart > 1) {
        const Op *b0 = s->A->kind == OP_DIRAC ? s->A->base : s->A;
        xr_fuse = csr_xr_fusable(b0->csr, b0->dist);
    }
    MGCR_CHECK(!multi || (!p.left_precond && (!p.right_precond || flex)), MGCR_ERR_UNSUPPORTED,
               "on a distributed operator only flexible right preconditioning is available (set flexible = 1)");
    const DevState *cst = s->st;
    const double *normN = s->partsN;
    if (fuse_start) {
        const Op *b0 = s->A->kind == OP_DIRAC ? s->A->base : s->A;
        MGCR_TRY(csr_init_apply(b0->csr, rhs, s->aps[0], s->A->kind == OP_DIRAC, s->A->k, (const cplx *)nullptr, s->partsA, s->partsR,
                                s->partsN, b0->dist, rmap));
    } else if (fuse_init) {
        const Op *b0 = s->A->kind == OP_DIRAC ? s->A->base : s->A;
        if (!alias0 && !multi && s->bnorm_src && s->bnorm_src->partsN_of == rhs && s->bnorm_src->partsN_g == g && bnorm_reuse_enabled()) {
            normN = s->bnorm_src->partsN;   // |b|^2: the partials of the solve that ran on this b just before
            MGCR_TRY(csr_init_apply(b0->csr, p0, s->aps[0], s->A->kind == OP_DIRAC, s->A->k, (const cplx *)nullptr, s->partsA, s->partsR,
                                    (double *)nullptr, b0->dist, rmap));
        } else
        MGCR_TRY(csr_init_apply(b0->csr, p0, s->aps[0], s->A->kind == OP_DIRAC, s->A->k, alias0 ? (const cplx *)nullptr : rhs, s->partsA,
                                s->partsR, s->partsN, b0->dist, rmap));
        if (normN == s->partsN) { s->partsN_of = rhs; s->partsN_g = g; }
    } else if (alias0) {
        KLAUNCH(init3_partials_kernel, g, rhs, (const cplx *)s->aps[0], n, s->partsN, s->partsR, s->partsA, cst, 0);
    } else if (alias_p0) {
        KLAUNCH(norm_partials_kernel, g, rhs, n, s->partsN, cst, 0);
        KLAUNCH(init3_partials_kernel, g, (const cplx *)s->r, (const cplx *)s->aps[0], n, (double *)nullptr, s->partsR, s->partsA, cst, 0);
    } else {
        KLAUNCH(norm_partials_kernel, g, rhs, n, s->partsN, cst, 0);
        KLAUNCH(norm_partials_kernel, g, (const cplx *)s->r, n, s->partsR, cst, 0);
        KLAUNCH(dot2_partials_kernel, g, (const cplx *)s->r, (const cplx *)s->aps[0], n, s->partsA, cst, 0);
    }
    RedRef refA = {s->partsA, g, RED_MAX_BLOCKS}, refR = {s->partsR, g, RED_MAX_BLOCKS};
    if (multi) {
        MGCR_TRY(comm_fold_allreduce(comm, s->partsN, 1, s->partsR, 1, s->dN, g));
        MGCR_TRY(comm_fold_allreduce(comm, s->partsA, 4, nullptr, 0, s->dA, g));
        KLAUNCH(init_kernel, 1, s->st, (const double *)s->dN, 1, 1, (const double *)(s->dN + 1), 1, 1, s->hist);
        refA = {s->dA, 1, 1};
        refR = {s->dRB, 1, 1};
    } else {
        KLAUNCH(init_kernel, 1, s->st, normN, g, RED_MAX_BLOCKS, (const double *)s->partsR, g, RED_MAX_BLOCKS, s->hist);
    }

    const int max_it = p.max_iter > 0 ? p.max_iter : 1;  // do..while: at least one iteration
    // the literal r = M(r) of src/GCR.h:236-238 changes the residual that is recorded: that mode keeps the full last step
    const bool skip_tail = !p.right_precond || flex;
    // where each step leaves the (true recurrence) residual — not with the literal hooks, which replace r by M r
    s->r_after.clear();
    if (!p.left_precond && (!p.right_precond || flex) && max_it <= LND) s->r_after.assign((size_t)max_it + 1, nullptr);
    int check_every = p.check_every > 0 ? p.check_every : 10;
    const cplx *rcur = alias0 ? rhs : s->r;  // lean: where the current residual lives (s->r at the start of every cycle)
    int iter_count = 0, cur = 0, global = 0;
    bool done = false;
    std::vector<hipEvent_t> prof_events;
    bool prof_step_build = false, prof_step_build_xr = false, prof_step_build_close = false;
    bool xr_prefetched = false;   // the next iteration's residual update already ran at the end of this one's launch
    // one iteration, enqueued on the library stream; `it` = iteration number relative to DevState::base
    auto one_iteration = [&](int it, bool last = false) -> int {
        iter_count++;
        set_apply_skip(SkipRef{&s->st->stop_at, it});
        auto mark = [&]() -> int {
            if (!(p.profile_spmv && !nested)) return MGCR_OK;
            hipEvent_t e;
            MGCR_HIP(hipEventCreate(&e));
            MGCR_HIP(hipEventRecord(e, c.stream));
            prof_events.push_back(e);
            return MGCR_OK;
        };
        MGCR_TRY(mark());  // 4 events per iteration: | xr (+M) | apply + dots | build |
        // slot the new direction goes to (src/GCR.h:277-287)
        const int lim = s->storage < iter_count ? s->storage : iter_count;  // src/GCR.h:251
        int ic_next = iter_count;
        if (iter_count % s->restart == 0) ic_next = 0;
        const int nxt = ic_next % s->storage;
        MGCR_TRY(ensure_slot(s, nxt));
        // alpha, x, r
        const cplx *dir;
        bool xr_now = false;
        const cplx *xr_in = nullptr;
        if (lean) {
            // D_nxt, what direction nxt is started from, lands in the p slot of that direction (nxt >= 1);
            // the step that closes the cycle only needs it for its own build
            cplx *dslot = nxt >= 1 ? s->ps[nxt] : (flex ? s->z : s->r);
            cplx *r_out = flex ? s->r : dslot;
            if (last && skip_tail && nested && s->defer_residual && !multi && !flex && !s->r_after.empty()) {
                // the caller forms this step's residual itself, from the previous one, Ap and alpha (ResidualSel)
                KLAUNCH(alpha_only_kernel, 1, s->st, it, refA.p, refA.nblk, refA.stride, s->den + cur, cur, s->lc, s->alphas + cur);
                s->defer_r_prev = rcur; s->defer_ap = s->aps[cur]; s->defer_alpha = s->alphas + cur; s->defer_it = global;
                s->r_after[(size_t)global] = rcur;   // placeholder: never read (st->iter == defer_it takes the on-the-fly path)
                for (int k = 0; k < 3; k++) MGCR_TRY(mark());
                iter_count = ic_next;
                cur = nxt;
                return MGCR_OK;
            }
            if (last && skip_tail && nested && s->discard_residual && !multi) {
                KLAUNCH(alpha_only_kernel, 1, s->st, it, refA.p, refA.nblk, refA.stride, s->den + cur, cur, s->lc, (cplx *)nullptr);
                s->r_after.clear();
                for (int k = 0; k < 3; k++) MGCR_TRY(mark());
                iter_count = ic_next;
                cur = nxt;
                return MGCR_OK;
            }
            xr_now = xr_fuse && !(last && skip_tail);   // latency regime: the update runs inside the apply kernel below
            xr_in = rcur;
            if (xr_prefetched) xr_prefetched = false;   // the previous step's launch ended with this update (gcr_stepbuild.hip)
            else if (!xr_now)
            KLAUNCH((xr_update_kernel<true, true>), g, s->st, it, refA.p, refA.nblk, refA.stride, (const cplx *)nullptr,
                    (const cplx *)s->aps[cur], x, rcur, r_out, n, s->partsR, s->den + cur, s->alphas, cur, s->lc);
            rcur = r_out;
            dir = r_out;
            if (flex && !(last && skip_tail)) {
                MGCR_TRY(op_apply_raw((Op *)p.right_precond, s->r, dslot, n));
                dir = dslot;
            }
        } else {
        if (defer)
            KLAUNCH((xr_update_kernel<true, false>), g, s->st, it, refA.p, refA.nblk, refA.stride, (const cplx *)s->ps[cur],
                    (const cplx *)s->aps[cur], x, (const cplx *)s->r, s->r, n, s->partsR, s->den + cur, s->alphas, cur, s->lc);
        else
            KLAUNCH((xr_update_kernel<false, false>), g, s->st, it, refA.p, refA.nblk, refA.stride, (const cplx *)s->ps[cur],
                    (const cplx *)s->aps[cur], x, (const cplx *)s->r, s->r, n, s->partsR, s->den + cur, s->alphas, cur, s->lc);
        dir = s->r;
        if (flex && !(last && skip_tail)) {
            MGCR_TRY(op_apply_raw((Op *)p.right_precond, s->r, s->z, n));
            dir = s->z;
        } else if (p.right_precond && !flex) {  // src/GCR.h:236-238
            MGCR_TRY(op_apply_raw((Op *)p.right_precond, s->r, s->tmp, n));
            std::swap(s->r, s->tmp);
            dir = s->r;
            KLAUNCH(norm_partials_kernel, g, (const cplx *)s->r, n, s->partsR, cst, it);
        }
        }
        if (!s->r_after.empty() && global < (int)s->r_after.size()) s->r_after[(size_t)global] = lean ? rcur : (const cplx *)s->r;
        if (last && skip_tail) {
            // nothing after this iteration: no preconditioner apply, no A r, no beta dots, no direction build — only the
            // step's bookkeeping (the x updates still pending are applied by flush_x_kernel below)
            MGCR_TRY(mark());
            MGCR_TRY(mark());
            RedRef fr = refR;
            if (multi) {
                MGCR_TRY(comm_fold_allreduce(comm, s->partsR, 1, nullptr, 0, s->dRB, g));
                fr = {s->dRB, 1, 1};
            }
            KLAUNCH(finish_step_kernel, 1, s->st, it, fr.p, fr.nblk, fr.stride, s->hist, s->hist_cap);
            MGCR_TRY(mark());
            iter_count = ic_next;
            cur = nxt;
            return MGCR_OK;
        }
        MGCR_TRY(mark());
        const int nchunk = (lim + ND - 1) / ND;
        int ch0 = 0;
        // apply + dots + build in ONE launch (gcr_stepbuild.hip) where A r of a thread's rows fits LDS: not the step that closes a cycle
        bool step_build = false;
        const bool closes = ic_next == 0;   // (lim == restart then; up to 5 directions the closing step has its one-launch form too)
        if (fuse_ok && lean && !flex && !multi && !xr_now && rmap.band == 0 && (!closes || stepbuild_close_enabled()) &&
            !graphs_enabled()) {   // (a captured cycle would replay the exchange's generation numbers)
            const Op *b0 = s->A->kind == OP_DIRAC ? s->A->base : s->A;
            step_build = csr_step_build_eligible(b0->csr, b0->dist, lim);
        }
        if (step_build) {
            prof_step_build = true;
            const Op *b0 = s->A->kind == OP_DIRAC ? s->A->base : s->A;
            const cplx *vecs[FND];
            for (int j = 0; j < FND; j++) vecs[j] = s->aps[j < lim ? j : 0];
            // ... and, unless the next step is the solve's last (whose update takes other kernels), that step's residual update too:
            // r and the new Ap of a thread's rows are on the chip, alpha costs one more exchange instead of a launch
            cplx *xr_out = nullptr;
            const bool next_is_special_last = global + 1 == max_it && nested && (s->defer_residual || s->discard_residual);
            if (global + 1 <= max_it && !next_is_special_last && stepbuild_xr_enabled()) {
                const int ic2 = ((ic_next + 1) % s->restart == 0) ? 0 : ic_next + 1;
                const int nxt2 = ic2 % s->storage;
                MGCR_TRY(ensure_slot(s, nxt2));
                xr_out = nxt2 >= 1 ? s->ps[nxt2] : s->r;
            }
            const cplx *cps[FND];
            for (int j = 0; j < FND; j++) cps[j] = s->ps[j < lim ? j : 0];
            MGCR_TRY(csr_step_build(b0->csr, dir, s->A->kind == OP_DIRAC, s->A->k, vecs, lim, s->st, it, refR.p, refR.nblk, refR.stride, s->hist,
                                    s->hist_cap, s->den, s->aps[nxt], s->partsA, s->lc, rmap, xr_out, s->den + nxt, nxt, s->partsR,
                                    closes ? cps : nullptr, closes ? s->ps[0] : nullptr, closes ? x : nullptr));
            xr_prefetched = xr_out != nullptr;
            prof_step_build_xr = prof_step_build_xr || xr_prefetched;
            prof_step_build_close = prof_step_build_close || closes;
            MGCR_TRY(mark());
            MGCR_TRY(mark());
            iter_count = ic_next;
            cur = nxt;
            return MGCR_OK;
        }
        bool tail_rb = false;   // the fold + exchange of |r|^2 and the beta numerators ran inside the apply kernel
        if (fuse_ok) {
            // Ar = A dir and the <Ar, Aps_j> partials of the first FND = 10 stored directions in one pass (gcr_fused.hip);
            // with more than that (restart > 10) multidot_kernel takes directions 8.. in its chunks of ND = 8 (8 and 9 twice:
            // the same sums, the same bits)
            const int nf = lim < FND ? lim : FND;
            const cplx *vecs[FND];
            for (int j = 0; j < FND; j++) vecs[j] = s->aps[j < nf ? j : 0];
            const Op *b0 = s->A->kind == OP_DIRAC ? s->A->base : s->A;
            if (xr_now)
                MGCR_TRY(csr_step_apply_xr(b0->csr, xr_in, s->aps[cur], const_cast<cplx *>(dir), s->ar, s->A->kind == OP_DIRAC, s->A->k, vecs,
                                           nf, s->partsB, s->partsR, s->st, it, refA.p, refA.nblk, refA.stride, s->den + cur, cur, s->lc, rmap));
            else {
                // multi-GPU, scalars by peer writes: the apply kernel's last workgroup folds |r|^2 (the residual update's partials)
                // and the beta numerators and sums them over the ranks — no fold + exchange launch behind it (pw_tail_dev.h)
                PwTail pw;
                if (multi && lim <= FND && csr_step_apply_has_pw_tail(b0->csr, b0->dist) && comm_pw_tail_begin(comm, &pw)) {
                    pw.pa = s->partsR; pw.na = 1; pw.pb = s->partsB; pw.nb = 2 * lim; pw.out = s->dRB; pw.nblk = g;
                    MGCR_TRY(csr_step_apply(b0->csr, dir, s->ar, s->A->kind == OP_DIRAC, s->A->k, vecs, nf, s->partsB, b0->dist, rmap, &pw));
                    tail_rb = true;
                } else
                MGCR_TRY(csr_step_apply(b0->csr, dir, s->ar, s->A->kind == OP_DIRAC, s->A->k, vecs, nf, s->partsB, b0->dist, rmap));
            }
            ch0 = lim <= FND ? nchunk : 1;
        } else {
            MGCR_TRY(op_apply_raw(s->A, dir, s->ar, n));  // src/GCR.h:242
            if (p.left_precond) {                         // src/GCR.h:245-247
                MGCR_TRY(op_apply_raw((Op *)p.left_precond, s->ar, s->tmp, n));
                std::swap(s->ar, s->tmp);
            }
        }
        for (int ch = ch0; ch < nchunk; ch++) {
            DirPtrs d;
            int nd = lim - ch * ND < ND ? lim - ch * ND : ND;
            for (int j = 0; j < ND; j++) {
                int sl = ch * ND + (j < nd ? j : 0);
                d.ps[j] = s->ps[sl]; d.aps[j] = s->aps[sl]; d.slot[j] = sl;
            }
            MGCR_TRY(launch_multidot(g, nd, cst, it, (const cplx *)s->ar, d, ch * ND, n, s->partsB, rmap));
        }
        MGCR_TRY(mark());
        bool tail_a = false;    // ... and that of <r,Ap'>, <Ap',Ap'> inside the build kernel
        RedRef refB = {s->partsB, g, RED_MAX_BLOCKS};
        if (multi) {  // one all-reduce for |r|^2 and all beta numerators of the step
            if (!tail_rb) MGCR_TRY(comm_fold_allreduce(comm, s->partsR, 1, s->partsB, 2 * lim, s->dRB, g));
            refB = {s->dRB + 1, 1, 1};
        }
        if (lean) {
            LeanArgs a;
            a.g = g; a.nd = lim; a.rdir = flex; a.st = s->st; a.it = it; a.B = refB; a.R = refR; a.hist = s->hist;
            a.hist_cap = s->hist_cap; a.den = s->den;
            for (int j = 0; j < LND; j++) { int sl = j < lim ? j : 0; a.d.ps[j] = s->ps[sl]; a.d.aps[j] = s->aps[sl]; a.d.slot[j] = sl; }
            a.dir = dir; a.r = rcur; a.ar = s->ar; a.p_out = s->ps[0]; a.ap_out = s->aps[nxt]; a.n = n; a.partsA = s->partsA;
            a.x = x; a.lc = s->lc;
            PwTail pwa;
            if (multi && lim <= ND && comm_pw_tail_begin(comm, &pwa)) {   // the build kernel folds and exchanges <r,Ap'>, <Ap',Ap'> itself
                pwa.pa = s->partsA; pwa.na = 4; pwa.pb = nullptr; pwa.nb = 0; pwa.out = s->dA; pwa.nblk = g;
                a.pw = &pwa;
                tail_a = true;
            }
            if (ic_next != 0) {
                // the next step is the solve's last and all it takes from this one are <r,Ap'>, <Ap',Ap'> (alpha_only_kernel: the caller — a
                // post-smoother's, a coarsest solve's — discards the residual): Ap' itself is never read, so it is not written
                static const bool skip_dead_ap = !(getenv("MGCR_SKIP_DEAD_AP") && atoi(getenv("MGCR_SKIP_DEAD_AP")) == 0);
                if (skip_dead_ap && global + 1 == max_it && skip_tail && nested && s->discard_residual && !multi) a.ap_out = nullptr;
                MGCR_TRY(launch_build_lean(a, 0));              // lim == nxt
            } else if (lim <= ND) {
                MGCR_TRY(launch_build_close(a));                // lim == restart: closes the cycle
            } else {                                            // ... in two kernels when restart > 8
                MGCR_TRY(launch_close_x(a));
                MGCR_TRY(launch_build_lean(a, 1));
            }
        } else
        for (int ch = 0; ch < nchunk; ch++) {
            BuildArgs a;
            a.g = g;
            a.nd = lim - ch * ND < ND ? lim - ch * ND : ND;
            for (int j = 0; j < ND; j++) {
                int sl = ch * ND + (j < a.nd ? j : 0);
                a.d.ps[j] = s->ps[sl]; a.d.aps[j] = s->aps[sl]; a.d.slot[j] = sl;
            }
            a.first = ch == 0; a.last = ch == nchunk - 1; a.rdir = dir != s->r;
            // the step that closes a restart cycle streams every ps slot of the cycle: apply the deferred x updates there
            a.xupd = defer && ic_next == 0;
            a.st = s->st; a.it = it; a.B = refB; a.book = a.last ? 1 : 0; a.R = refR; a.hist = s->hist; a.hist_cap = s->hist_cap;
            a.den = s->den; a.base = ch * ND; a.dir = dir; a.r = s->r; a.ar = s->ar; a.accp = s->accp; a.accap = s->accap;
            a.p_out = s->ps[nxt]; a.ap_out = s->aps[nxt]; a.n = n; a.partsA = s->partsA; a.x = x; a.alphas = s->alphas;
            MGCR_TRY(launch_build(a));
        }
        if (multi && !tail_a) {
            MGCR_TRY(comm_fold_allreduce(comm, s->partsA, 4, nullptr, 0, s->dA, g));
        }
        MGCR_TRY(mark());
        iter_count = ic_next;
        cur = nxt;
        return MGCR_OK;
    };

    // hipGraph (opt-in, MGCR_GRAPH=1): in restart mode every cycle of R iterations is the same launch sequence with the
    // same arguments (iteration numbers are base-relative, base lives on the device), so one cycle can be captured once
    // and replayed.  Measured on MI355X: +11 % iterations/s on the 3072-row sample while an iteration was 4 launches;
    // since it shrank to 3 (fused apply + dots) replay no longer pays at any size — eager launches run 1-4 % FASTER
    // from 512 to 1.4 M rows (e.g. 64^3: 33.6 k against 32.3 k it/s), equal on the sample: the loop is bound by the
    // GPU-side dependency between consecutive short kernels (~8 us each), not by the host's launch cost.  Hence off
    // by default, and gated to <= 2^18 rows when on.
    const int R = s->restart;
    bool use_graph = graphs_enabled() && n <= GRAPH_MAX_ROWS && defer && !multi && !p.left_precond && !p.right_precond &&
                     !p.profile_spmv && max_it >= 2 * R && R <= s->storage;
    if (use_graph) s->r_after.clear();   // replayed cycles do not pass through the host-side step counter
    if (use_graph && (s->graph_exec == nullptr || s->graph_x != x || s->graph_R != R || s->graph_n != n)) {
        if (s->graph_exec) { hipGraphExecDestroy(s->graph_exec); s->graph_exec = nullptr; }
        hipGraph_t graph = nullptr;
        MGCR_HIP(hipStreamBeginCapture(c.stream, hipStreamCaptureModeThreadLocal));
        int rc = MGCR_OK;
        for (int i = 1; i <= R && rc == MGCR_OK; i++) rc = one_iteration(i);
        if (rc == MGCR_OK) {
            hipLaunchKernelGGL(advance_kernel, dim3(1), dim3(1), 0, c.stream, s->st, R);
        }
        hipError_t e = hipStreamEndCapture(c.stream, &graph);
        if (rc != MGCR_OK) { if (graph) hipGraphDestroy(graph); return rc; }
        MGCR_HIP(e);
        e = hipGraphInstantiate(&s->graph_exec, graph, nullptr, nullptr, 0);
        hipGraphDestroy(graph);
        MGCR_HIP(e);
        s->graph_x = x; s->graph_R = R; s->graph_n = n;
        // the capture pass walked the host-side cycle state through one full cycle: back at its start
    }
    int rel = 0;  // iterations enqueued eagerly since the last advance of DevState::base
    int last_check = 0;
    while (global < max_it && !done) {
        if (use_graph && iter_count == 0 && rel == 0 && max_it - global >= R) {
            MGCR_HIP(hipGraphLaunch(s->graph_exec, c.stream));
            global += R;
        } else {
            global++;
            rel++;
            MGCR_TRY(one_iteration(rel, global == max_it));
        }
        if (!nested && (global / check_every != last_check || global == max_it)) {
            last_check = global / check_every;
            MGCR_HIP(hipMemcpyAsync(c.h_mail, s->st, sizeof(DevState), hipMemcpyDeviceToHost, c.stream));
            MGCR_HIP(hipStreamSynchronize(c.stream));
            const DevState *hs = (const DevState *)c.h_mail;
            if (hs->stop_at != INT_MAX) done = true;
        }
    }
    if (defer) {  // x updates still pending (solve ended inside a restart cycle); ps[0..npend) hold their directions
        DirPtrs d0;
        for (int j = 0; j < LND; j++) { int sl = j < s->storage ? j : 0; d0.ps[j] = s->ps[sl]; d0.aps[j] = s->aps[sl]; d0.slot[j] = sl; }
        d0.ps[0] = p0;
        if (assign_x && s->keep_pending && nested) {
            // the caller writes x = sum coef_j v_j itself, together with whatever else it has to add (V-cycle: + P x_c)
            for (int j = 0; j < LND; j++) s->pending.v[j] = d0.ps[j];
            s->pending.coef = lean ? (const cplx *)s->lc->cx : (const cplx *)s->alphas;
            s->pending.st = s->st;
            s->has_pending = true;
        } else {
        // (npend is not cleared afterwards: nothing reads it again before the next solve's reset_kernel zeroes it)
        KLAUNCH(flush_x_kernel, g, s->st, lean ? (const cplx *)s->lc->cx : (const cplx *)s->alphas, d0, x, n, assign_x ? 1 : 0);
        }
    }
    if (nested) return MGCR_OK;
    if (!prof_events.empty()) {
        MGCR_HIP(hipStreamSynchronize(c.stream));
        for (int k = 0; k < 3; k++) g_prof_phase_ms[k] = 0.;
        for (size_t i = 0; i + 3 < prof_events.size(); i += 4)
            for (int k = 0; k < 3; k++) {
                float ms = 0.f;
                hipEventElapsedTime(&ms, prof_events[i + k], prof_events[i + k + 1]);
                g_prof_phase_ms[k] += ms;
            }
        g_prof_iters = (int)(prof_events.size() / 4);
        g_prof_fused = prof_step_build_close ? 4 : prof_step_build_xr ? 3 : prof_step_build ? 2 : (fuse_ok && xr_fuse) ? 5 : fuse_ok ? 1 : 0;
        for (hipEvent_t e : prof_events) hipEventDestroy(e);
    }
    const int frc = gcr_finish(s, hist, hist_cap, n_iter, converged);
    if (multi) MGCR_TRY(comm_check(comm));   // a peer-write wait that timed out poisoned the scalars with NaN
    MGCR_TRY(resident_check(true));          // a one-launch step (gcr_stepbuild.hip) that was not co-resident gave up: gcr_run repeats the solve
    return frc;
}

// GCR::solve.  A top-level solve that took a one-launch path (gcr_resident.hip, gcr_stepbuild.hip: workgroups that wait for each
// other) and whose launch gave up — another process's kernels held CUs, so the grid was not co-resident; bounded polls, NaN
// results — is REPEATED here on the multi-kernel path, from the untouched right-hand side and the x the caller handed in: the
// caller gets the solve it asked for from its first call (mgcr_stat "one_launch_fallbacks" counts these).  The one-launch paths
// have switched themselves off by then (resident_check).  x on entry: zero when the caller says so (the Field was zeroed by
// mgcr_vec_zero and not written since: re-zeroed for the second run), otherwise a copy is kept whenever a one-launch path may run.
static int64_t g_fallbacks = 0;
int64_t gcr_fallback_count() { return g_fallbacks; }
int gcr_run(GcrState *s, const cplx *rhs, cplx *x, bool nested, double *hist, int hist_cap, int *n_iter, int *converged, bool x_known_zero) {
    if (nested) return gcr_run_once(s, rhs, x, true, hist, hist_cap, n_iter, converged);
    MGCR_CHECK(s->A, MGCR_ERR_INVALID, "GCR has no operator (call initialise / mgcr_gcr_set_operator first)");
    const int64_t n = s->A->dim;
    // Which solves can meet a one-launch path at all?  The solve itself up to 2^21 rows (the paths' own conditions are narrower —
    // operator kind, storage, restart length — but none of them takes more); at any size when a preconditioner is attached, whose
    // nested solves (an MG cycle's coarsest level) may take one.  A copy too many costs one pass over x.
    const bool risky = one_launch_paths_enabled() && comm_live_count() == 0 &&
                       (n <= ((int64_t)1 << 21) || s->p.left_precond || s->p.right_precond);
    bool have_copy = false;
    if (risky && !x_known_zero) {
        if (s->xbak_n != n) {
            hipStreamSynchronize(ctx().stream);
            hipFree(s->xbak); s->xbak = nullptr; s->xbak_n = 0;
            MGCR_TRY(dalloc(&s->xbak, (size_t)n));
            s->xbak_n = n;
        }
        MGCR_TRY(k_copy(s->xbak, x, n));
        have_copy = true;
    }
    int rc = gcr_run_once(s, rhs, x, false, hist, hist_cap, n_iter, converged);
    if (rc != MGCR_INT_GAVE_UP) return rc;
    if (!have_copy && !x_known_zero) {   // (cannot happen: nothing that waits on other workgroups ran) — report like a host synchronisation point would
        set_error("one-launch solver kernel gave up and no copy of x was kept");
        return MGCR_ERR_HIP;
    }
    if (have_copy) MGCR_TRY(k_copy(x, s->xbak, n));
    else MGCR_TRY(k_zero(x, n));
    g_fallbacks++;
    rc = gcr_run_once(s, rhs, x, false, hist, hist_cap, n_iter, converged);
    return rc == MGCR_INT_GAVE_UP ? MGCR_ERR_HIP : rc;
}

void gcr_set_discard_residual(GcrState *s, bool on) { s->discard_residual = on; }
void gcr_set_bnorm_source(GcrState *s, GcrState *src) { s->bnorm_src = src; }
void gcr_set_keep_pending(GcrState *s, bool on) { s->keep_pending = on; }
// x = sum coef_j v_j for a caller that took the pending update and then needs x after all
int gcr_flush_pending(const PendingX &pd, cplx *x, int64_t n) {
    DirPtrs d0;
    for (int j = 0; j < LND; j++) { d0.ps[j] = pd.v[j]; d0.aps[j] = pd.v[j]; d0.slot[j] = j; }
    hipLaunchKernelGGL(flush_x_kernel, dim3(red_grid(n)), dim3(RED_THREADS), 0, ctx().stream, const_cast<DevState *>(pd.st), pd.coef, d0, x, n, 1);
    MGCR_HIP(hipGetLastError());
    return MGCR_OK;
}
// did the solve that just ran leave x unwritten (see keep_pending)?  Then *out says what x is.
bool gcr_take_pending(GcrState *s, PendingX *out) {
    if (!s->has_pending) return false;
    *out = s->pending;
    s->has_pending = false;
    return true;
}

// the residual of the solve that just ran on `s`, selected on the device by the number of steps it took
bool gcr_last_residual(GcrState *s, ResidualSel *out) {
    if (s->r_after.size() < 2 || s->r_after.size() > (size_t)LND + 1) return false;
    for (size_t k = 1; k < s->r_after.size(); k++)
        if (!s->r_after[k]) return false;
    for (int k = 0; k <= LND; k++) out->r[k] = s->r_after[(size_t)k < s->r_after.size() ? (size_t)k : s->r_after.size() - 1];
    out->r[0] = s->r_after[1];
    out->st = s->st;
    out->r_prev = s->defer_it ? s->defer_r_prev : nullptr;
    out->ap = s->defer_ap;
    out->alpha = s->defer_alpha;
    out->last_it = s->defer_it;
    return true;
}
void gcr_set_defer_residual(GcrState *s, bool on) { s->defer_residual = on; }

// nested solve from x0 = 0 into an x whose content does not matter (smoothers, coarsest solve, GCR as a preconditioner)
int gcr_run_from_zero(GcrState *s, const cplx *rhs, cplx *x) {
    s->x_from_zero = true;
    return gcr_run(s, rhs, x, true, nullptr, 0, nullptr, nullptr);
}

// x = init_rand(2) in the reference (src/GCR.h:63-68); here the caller-provided x0 or zero
int gcr_apply_as_operator(GcrState *s, const cplx *f, cplx *y) {
    MGCR_CHECK(s->A, MGCR_ERR_INVALID, "GCR has no operator (call initialise / mgcr_gcr_set_operator first)");
    const int64_t n = s->A->dim;
    if (s->x0_mode == 0 && s->x0) MGCR_TRY(k_copy_apply(y, s->x0, n));
    else return gcr_run_from_zero(s, f, y);
    return gcr_run(s, f, y, true, nullptr, 0, nullptr, nullptr);
}

}  // namespace mgcr
