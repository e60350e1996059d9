// A lean GCR step as ONE launch (src/GCR.h:230-287), for systems whose A r fits the chip's LDS: at most 4096 rows per
// workgroup of 1024 threads, two workgroups per CU — 2 097 152 rows on MI355X, i.e. up to Poisson 128^3, the metric's
// configuration.
//
// gcr_fused.hip's step_apply_kernel writes Ar (V), its partial sums of <Ar, Ap_j> go to memory, and gcr.hip's
// build_lean_kernel — behind a kernel boundary, because beta needs the sums over ALL workgroups — reads Ar back (V);
// xr_update_kernel, behind another boundary because alpha needs <r,Ap'> and <Ap',Ap'>, reads r and the new Ap' back.
// Here the kernel bodies run in one launch with exchange_dev.h's fence-free exchange (~3 us) where the boundaries were:
//   apply + dots | exchange | direction build (CLOSE: the cycle-closing form, gcr.hip build_close_kernel: x update and the
//   next cycle's first direction in a pass of their own) | exchange | XR: the NEXT step's residual update.
// A thread keeps the Ar of its 4 rows in LDS (64 KB per workgroup, thread-private slots) and overwrites it with the new
// Ap' for the update: Ar is neither written to nor read from memory, Ap' is not read back — per in-cycle step
// B_matrix + 16 ncol + (2 lim + 4) V instead of B_matrix + 16 ncol + (2 lim + 7) V, two kernel boundaries and two folds
// less.  The first trip's streams of the build and the update's first r are requested before an exchange is polled.
// Everything else is the kernels' code: the same rows per thread (RowMap), the same per-thread accumulation order, the
// same fold tree — the same bits (tests/test_gpu_stepbuild.py compares with the three-kernel path bit for bit).
//
// Needs all workgroups co-resident (they wait for each other): 64 VGPRs and <= 80 KB of LDS each, at most 2 x #CU
// workgroups, no other process on the device (no live communicator).  Up to 5 stored directions (beyond that the
// kernels need 128 registers: one workgroup per CU).  Bounded polls as in gcr_resident.hip: a missing workgroup makes the
// others leave with NaN results, the solve returns an error and the one-launch paths switch themselves off.
// The host (gcr.hip gcr_run) knows which launch already performed the next update (xr_prefetched); a solve that stops
// on the device turns the whole launch, or its update part, into a no-op like any other kernel of the solve.
#include <climits>
#include <cstdlib>

#include "internal.h"
#include "reduce.h"
#include "gcr_dev.h"
#include "spmv_dev.h"
#include "exchange_dev.h"

namespace mgcr {

constexpr int SB_MAX_TRIPS = 4;     // rows per thread whose Ar stays in LDS (4 x 1024 x 16 B = 64 KB per workgroup)
constexpr int SB_MAX_ND = 5;

struct StepBuildArgs {
    RowMat m;
    const cplx *x;           // the residual the step applies the operator to (= the direction's start D_k)
    const cplx *aps[SB_MAX_ND];
    int64_t n;
    int nlogical;
    RowMap rm;
    DevState *st;
    int it;
    const double *partsR;    // |r|^2 partials of the residual update that ran before
    int nblkR, strideR;
    double *hist;
    int hist_cap;
    const cplx *den;
    cplx *ap_out;
    double *partsA;
    LeanCoef *lc;
    // CLOSE: the step that closes a restart cycle (gcr.hip build_close_kernel): the cycle's directions, its first one rewritten, x updated
    const cplx *ps[SB_MAX_ND];
    cplx *p_out;
    cplx *xvec;
    // XR: the NEXT step's residual update (gcr.hip xr_update_kernel<true, true>) at the end of this launch
    cplx *xr_out;            // where it leaves r - alpha Ap' (the next ring slot)
    cplx *xr_den_slot;
    int xr_slot;
    double *partsR_out;
    v4i *slots;
    unsigned gen0;
    unsigned *abort_dev;
    int *abort_host;
    int spin_limit;
    int test_stall;          // tests: logical workgroup test_stall - 1 leaves before publishing anything (0: nobody)
};

// step bookkeeping (gcr.hip close_step; kept in step with it by tests/test_gpu_stepbuild.py)
__device__ __forceinline__ void sb_close_step(DevState *st, int it, double rr, double *hist, int hist_cap, bool clear_pending) {
    const int git = st->base + it;
    st->iter = git;
    st->rr = rr;
    if (git < hist_cap) hist[git] = sqrt(rr) / sqrt(st->bnorm2);
    if (!((rr / st->bnorm2) > st->tol2)) st->stop_at = git;
    if (clear_pending) st->npend = 0;
}

// REALC: the instantiation for real stencil coefficients (no per-slot real / complex decision, 14 scalar registers less — the
// kernels' scalar registers spill into vector-register lanes, which a wave then reads back one v_readlane at a time)
template <int MODE, int WT, int NDT, bool XR, bool CLOSE, bool REALC = false>
__global__ void __launch_bounds__(RED_THREADS, 8) step_build_kernel(StepBuildArgs a) {
    __shared__ double lds[(2 * NDT > 4 ? 2 * NDT : 4) * 17];
    __shared__ double lds_pw[2 * SB_MAX_ND * 17], lds_ws[2 * SB_MAX_ND * RES_GRP];
    __shared__ int gave_up;
    __shared__ cplx sbeta[NDT], scp[NDT];
    extern __shared__ __attribute__((aligned(16))) unsigned char sb_smem[];   // Ar of this workgroup's rows: [trip][thread]
    if (a.st->stop_at < a.st->base + a.it) return;
    const int lb = logical_workgroup(a.rm, (int)blockIdx.x, (int)gridDim.x);
    if (lb >= a.nlogical) return;
    cplx *arL = reinterpret_cast<cplx *>(sb_smem);
    ResSync sy;
    sy.slots = res_rsrc(a.slots, (unsigned)RES_SLOT_BYTES);
    sy.gen = a.gen0;
    sy.nblk = a.nlogical;
    sy.lb = lb;
    sy.abort_dev = a.abort_dev;
    sy.spin_limit = a.spin_limit;
    sy.pw = lds_pw;
    sy.ws = lds_ws;
    sy.gave_up = &gave_up;
    if (threadIdx.x == 0) gave_up = 0;
    if (a.test_stall && lb == a.test_stall - 1) return;   // (tests) the others must notice, give up and say so
    int64_t i0, end, stride;
    row_range(a.rm, lb, a.nlogical, a.n, &i0, &end, &stride);
    // ---- apply + dot products (gcr_fused.hip step_apply_kernel) ----
    {
        double v[2 * NDT];
#pragma unroll
        for (int j = 0; j < 2 * NDT; j++) v[j] = 0.;
        int trip = 0;
        for (int64_t i = i0; i < end; i += stride, trip++) {
            const PatLds pl{nullptr, nullptr, nullptr};
            cplx sum;
            if constexpr (REALC) sum = sten_row_product_t<WT, false, 1>(a.m, i, [&](int32_t j) -> cplx { return a.x[j]; });
            else sum = fused_row_product<MODE, WT>(a.m, i, 0, pl, [&](int32_t j) -> cplx { return a.x[j]; });
            const cplx yi = a.m.shift ? csub(a.x[i], cmul(a.m.k, sum)) : sum;
            arL[trip * RED_THREADS + (int)threadIdx.x] = yi;
            __builtin_amdgcn_sched_barrier(0);
            cplx b[NDT];
#pragma unroll
            for (int j = 0; j < NDT; j++) b[j] = ld_stream<true>(a.aps[j] + i);
#pragma unroll
            for (int j = 0; j < NDT; j++) {
                cplx t = cconj_mul(yi, b[j]);
                v[2 * j] += t.x;
                v[2 * j + 1] += t.y;
            }
        }
        // ---- the sums over all workgroups (exchange_dev.h), in place of the partial slab + fold_partials ----
        const double mine = block_sum_owner<2 * NDT>(v, lds);
        if ((int)threadIdx.x < 2 * NDT) {
            const v4i w4 = {__double2loint(mine), __double2hiint(mine), (int)sy.gen, 0};
            __builtin_amdgcn_raw_buffer_store_b128(w4, sy.slots, ((1 * RES_NV + (int)threadIdx.x) * RES_BLK + lb) * 16, 0, RES_SC1);
        }
    }
    // the first trip's streams of the build are requested BEFORE the exchange is waited for (they do not depend on beta): one
    // memory round trip of the launch hides behind the polls
    // (up to 3 stored directions: with more the kernel does not keep them next to everything else in 64 registers)
    constexpr bool PRE = NDT <= 3;
    cplx pre[NDT];
#pragma unroll
    for (int j = 0; j < NDT; j++) pre[j] = make_double2(0., 0.);
    if (PRE && i0 < end) {
#pragma unroll
        for (int j = 0; j < NDT; j++) pre[j] = ld_stream<NTS>((CLOSE ? a.ps[j] : a.aps[j]) + i0);
    }
    const bool ok = res_collect<2 * NDT>(sy, 1);
    if (!ok) {   // somebody is missing: leave, with results nobody can mistake for numbers
        if (threadIdx.x == 0) {
            __hip_atomic_store(a.abort_dev, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(a.abort_host, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        for (int64_t i = i0; i < end; i += stride) a.ap_out[i] = make_double2(__builtin_nan(""), __builtin_nan(""));
        if ((int)threadIdx.x < 4) a.partsA[threadIdx.x * RED_MAX_BLOCKS + lb] = __builtin_nan("");
        return;
    }
    // ---- direction build (gcr.hip build_lean_kernel, not closing) ----
    bool ends_here = false;   // XR: did this step end the solve?  Every workgroup folds |r|^2 itself: same bits, same answer
    if (XR || lb == 0) {
        double rr[1];
        fold_partials<1>(a.partsR, a.nblkR, a.strideR, rr, lds);
        if (lb == 0 && threadIdx.x == 0) sb_close_step(a.st, a.it, rr[0], a.hist, a.hist_cap, CLOSE);
        ends_here = !((rr[0] / a.st->bnorm2) > a.st->tol2);
    }
    if ((int)threadIdx.x < NDT) sbeta[threadIdx.x] = cdiv(make_double2(res_total(sy, 2 * threadIdx.x), res_total(sy, 2 * threadIdx.x + 1)), a.den[threadIdx.x]);
    __syncthreads();
    if constexpr (CLOSE) {   // gcr.hip build_close_kernel: cp_m = sum_{j >= m} beta_j T_jm (cp_0 = sum_j beta_j t_j), in every workgroup
        if ((int)threadIdx.x < NDT) {
            const int m = threadIdx.x;
            const LeanCoef *lc = a.lc;
            cplx c = make_double2(0., 0.);
            if (m == 0) {
                for (int j = 0; j < NDT; j++) c = cadd(c, cmul(sbeta[j], j == 0 ? make_double2(1., 0.) : lc->t[j]));
            } else {
                for (int j = m; j < NDT; j++) c = cadd(c, cmul(sbeta[j], j == m ? make_double2(1., 0.) : lc->T[j * LND + m]));
            }
            scp[m] = c;
        }
        __syncthreads();
    }
    if (!CLOSE && lb == 0 && (int)threadIdx.x <= NDT) {   // row k = NDT of the coefficient table
        constexpr int k = NDT;
        const int m = threadIdx.x;
        LeanCoef *lc = a.lc;
        cplx c = make_double2(0., 0.);
        if (m == 0) {
            for (int j = 0; j < k; j++) c = csub(c, cmul(sbeta[j], j == 0 ? make_double2(1., 0.) : lc->t[j]));
            lc->t[k] = c;
        } else if (m < k) {
            for (int j = m; j < k; j++) c = csub(c, cmul(sbeta[j], j == m ? make_double2(1., 0.) : lc->T[j * LND + m]));
            lc->T[k * LND + m] = c;
        } else {
            lc->T[k * LND + k] = make_double2(1., 0.);
        }
    }
    cplx beta[NDT];
#pragma unroll
    for (int j = 0; j < NDT; j++) beta[j] = to_sgpr(sbeta[j]);
    double v[4] = {0., 0., 0., 0.};
    int trip = 0;
    if constexpr (CLOSE) {
        // x += sum_j cx_j p_j;  P0' = dir - sum_j cp_j p_j  (p_0 = P0, p_m = D_m): a pass of its own over the p streams, so that they
        // and the Ap streams below are never in registers together (64 VGPRs: two workgroups per CU)
        cplx cp[NDT], cx[NDT];
#pragma unroll
        for (int j = 0; j < NDT; j++) {
            cp[j] = to_sgpr(scp[j]);
            cx[j] = to_sgpr(a.lc->cx[j]);
        }
        for (int64_t i = i0; i < end; i += stride) {
            cplx pj[NDT];
#pragma unroll
            for (int j = 0; j < NDT; j++) pj[j] = (PRE && i == i0) ? pre[j] : ld_stream<NTS>(a.ps[j] + i);
            const cplx dv = a.x[i];
            cplx xv = a.xvec[i];
#pragma unroll
            for (int j = 0; j < NDT; j++) xv = cadd(xv, cmul(cx[j], pj[j]));
            a.xvec[i] = xv;
            cplx pc = make_double2(0., 0.);
#pragma unroll
            for (int j = 0; j < NDT; j++) pc = csub(pc, cmul(cp[j], pj[j]));
            st_stream<NTS>(a.p_out + i, cadd(dv, pc));
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    for (int64_t i = i0; i < end; i += stride, trip++) {
        cplx aj[NDT];
#pragma unroll
        for (int j = 0; j < NDT; j++) aj[j] = (PRE && !CLOSE && i == i0) ? pre[j] : ld_stream<NTS>(a.aps[j] + i);
        const cplx av = arL[trip * RED_THREADS + (int)threadIdx.x], rv = a.x[i];
        cplx ac = make_double2(0., 0.);
#pragma unroll
        for (int j = 0; j < NDT; j++) ac = csub(ac, cmul(beta[j], aj[j]));
        const cplx an = cadd(av, ac);
        a.ap_out[i] = an;
        if (XR) arL[trip * RED_THREADS + (int)threadIdx.x] = an;   // (A r is not needed any more; the update below wants Ap')
        cplx t = cconj_mul(rv, an);
        v[0] += t.x; v[1] += t.y;
        cplx u = cconj_mul(an, an);
        v[2] += u.x; v[3] += u.y;
    }
    const double mine = block_sum_owner<4>(v, lds);
    if (threadIdx.x < 4) a.partsA[threadIdx.x * RED_MAX_BLOCKS + lb] = mine;
    if constexpr (XR) {
        // ---- the next step's residual update (gcr.hip xr_update_kernel<true, true>): alpha needs <r,Ap'>, <Ap',Ap'> over ALL
        // workgroups — a second exchange instead of a kernel boundary; r and Ap' of the thread's rows are on the chip ----
        if ((int)threadIdx.x < 4) {
            const v4i w4 = {__double2loint(mine), __double2hiint(mine), (int)sy.gen, 0};
            __builtin_amdgcn_raw_buffer_store_b128(w4, sy.slots, ((2 * RES_NV + (int)threadIdx.x) * RES_BLK + lb) * 16, 0, RES_SC1);
        }
        const cplx xr0 = i0 < end ? a.x[i0] : make_double2(0., 0.);   // (requested before the polls, like `pre` above)
        if (!res_collect<4>(sy, 2)) {
            if (threadIdx.x == 0) {
                __hip_atomic_store(a.abort_dev, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(a.abort_host, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
            for (int64_t i = i0; i < end; i += stride) a.xr_out[i] = make_double2(__builtin_nan(""), __builtin_nan(""));
            if (threadIdx.x == 0) a.partsR_out[lb] = __builtin_nan("");
            return;
        }
        if (ends_here) return;   // the step converged: the next one's kernels are no-ops (gcr_dev.h DevState::stop_at)
        const cplx num = make_double2(res_total(sy, 0), res_total(sy, 1)), den = make_double2(res_total(sy, 2), res_total(sy, 3));
        const cplx alpha = to_sgpr(cdiv(num, den));
        if (lb == 0 && threadIdx.x == 0) {
            *a.xr_den_slot = den;
            a.st->npend = a.xr_slot + 1;
        }
        if (lb == 0 && (int)threadIdx.x < LND) lean_pending_update(a.lc, a.xr_slot, alpha, (int)threadIdx.x);
        double vr[1] = {0.};
        trip = 0;
        for (int64_t i = i0; i < end; i += stride, trip++) {
            const cplx rn = csub(i == i0 ? xr0 : a.x[i], cmul(alpha, arL[trip * RED_THREADS + (int)threadIdx.x]));
            a.xr_out[i] = rn;
            vr[0] += rn.x * rn.x + rn.y * rn.y;
        }
        const double tot = block_sum_owner<1>(vr, lds);
        if (threadIdx.x == 0) a.partsR_out[lb] = tot;
    }
}

static int g_stepbuild = -1;
static bool stepbuild_enabled() {
    if (g_stepbuild < 0) g_stepbuild = !(getenv("MGCR_STEPBUILD") && atoi(getenv("MGCR_STEPBUILD")) == 0);
    return g_stepbuild != 0;
}
bool stepbuild_is_enabled() { return stepbuild_enabled(); }
bool set_stepbuild_enabled(bool on) {
    bool prev = stepbuild_enabled();
    g_stepbuild = on ? 1 : 0;
    return prev;
}
static int64_t g_stepbuild_launches = 0;
int64_t stepbuild_launch_count() { return g_stepbuild_launches; }

// the instantiation a step with `nd` stored directions launches
static bool sb_real_enabled() {
    static const bool on = !(getenv("MGCR_SB_REAL") && atoi(getenv("MGCR_SB_REAL")) == 0);
    return on;
}
static const void *sb_kernel(int nd, bool xr, bool close, bool realc) {
#define SBR(NDT, R) (close ? (xr ? (const void *)step_build_kernel<3, 7, NDT, true, true, R> : (const void *)step_build_kernel<3, 7, NDT, false, true, R>) \
                           : (xr ? (const void *)step_build_kernel<3, 7, NDT, true, false, R> : (const void *)step_build_kernel<3, 7, NDT, false, false, R>))
#define SBK(NDT) (realc ? SBR(NDT, true) : SBR(NDT, false))
    switch (nd) {
        case 1: return SBK(1);
        case 2: return SBK(2);
        case 3: return SBK(3);
        case 4: return SBK(4);
        default: return SBK(5);
    }
#undef SBK
#undef SBR
}
// Do `grid` workgroups of this instantiation, with this much dynamic LDS, fit the chip AT ONCE?  The workgroups wait for each
// other inside the launch, so the answer has to come from the runtime (registers and LDS of the code object that was actually
// built: another compiler, -DMGCR_RES_TIMING, ... change them), not from arithmetic on what the kernel is meant to need.
// Asked once per instantiation and LDS size.  (What the runtime cannot know — CUs held by another process — is what the bounded
// polls and gcr_run's repeat are for.)
bool launch_is_coresident(const void *kernel, int threads, size_t dyn_lds, int grid) {
    struct Key { const void *k; size_t lds; int threads; int per_cu; };
    static std::vector<Key> seen;
    int per_cu = -1;
    for (const Key &e : seen)
        if (e.k == kernel && e.lds == dyn_lds && e.threads == threads) per_cu = e.per_cu;
    if (per_cu < 0) {
        if (dyn_lds > 0) (void)hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(dyn_lds > 64 * 1024 ? dyn_lds : 64 * 1024));
        int nb = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kernel, threads, dyn_lds) != hipSuccess) { (void)hipGetLastError(); nb = 0; }
        per_cu = nb;
        seen.push_back(Key{kernel, dyn_lds, threads, per_cu});
    }
    static const int force = getenv("MGCR_TEST_OCCUPANCY") ? atoi(getenv("MGCR_TEST_OCCUPANCY")) : -1;   // tests: pretend the runtime said so
    if (force >= 0) per_cu = force;
    return exchange_shared_init() == MGCR_OK && (int64_t)per_cu * exchange_shared().cus >= grid;
}

static size_t sb_lds_bytes(const CsrDev &A, int g) {
    return sizeof(cplx) * RED_THREADS * (size_t)((A.nrow + (int64_t)g * RED_THREADS - 1) / ((int64_t)g * RED_THREADS));
}

// can step `lim` of a lean cycle on A run as one launch?  (single GPU, 7-slot stencil view, <= 5 stored directions, A r in LDS)
bool csr_step_build_eligible(const CsrDev &A, const DistCsr *dist, int lim) {
    if (!stepbuild_enabled() || dist || comm_live_count() > 0 || lim < 1 || lim > SB_MAX_ND) return false;
    if (!csr_fusable(A, nullptr) || !csr_stencil_active(A) || A.sten_rare || sten_slots(A) != 7) return false;
    const int g = red_grid(A.nrow);
    if (g < 64 || g % 8 != 0) return false;   // (smaller systems have the resident solver or the xr-fused kernels)
    if ((int64_t)g * RED_THREADS * SB_MAX_TRIPS < A.nrow) return false;
    if (A.reach >= ((int64_t)1 << 15)) return false;   // rows that reach this far take the LDS-window kernels (gcr_fused.hip)
    if (exchange_shared_init() != MGCR_OK) return false;
    if (g > RES_BLK) return false;
    // every form the step may be launched in (with / without the next residual update, closing or not) must be co-resident
    const size_t lds = sb_lds_bytes(A, g);
    const bool realc = sb_real_enabled() && row_mat(A, false, cplx{0., 0.}).realv;
    for (int xr = 0; xr < 2; xr++)
        for (int cl = 0; cl < 2; cl++)
            if (!launch_is_coresident(sb_kernel(lim, xr != 0, cl != 0, realc), RED_THREADS, lds, g)) return false;
    return true;
}

int csr_step_build(const CsrDev &A, const cplx *x, bool shift, cplx k, const cplx *const *aps, int nd, DevState *st, int it, const double *partsR,
                   int nblkR, int strideR, double *hist, int hist_cap, const cplx *den, cplx *ap_out, double *partsA, LeanCoef *lc,
                   const RowMap &rm, cplx *xr_out, cplx *xr_den_slot, int xr_slot, double *partsR_out, const cplx *const *close_ps, cplx *close_p_out,
                   cplx *close_x) {
    MGCR_CHECK(nd >= 1 && nd <= SB_MAX_ND, MGCR_ERR_INVALID, "csr_step_build: 1..5 directions");
    MGCR_TRY(exchange_shared_init());
    ExchangeShared &sh = exchange_shared();
    StepBuildArgs a;
    a.m = row_mat(A, shift, k);
    a.x = x;
    for (int j = 0; j < SB_MAX_ND; j++) a.aps[j] = aps[j < nd ? j : 0];
    a.n = A.nrow;
    const int g = red_grid(A.nrow);
    a.nlogical = g;
    a.rm = rm;
    a.st = st; a.it = it; a.partsR = partsR; a.nblkR = nblkR; a.strideR = strideR; a.hist = hist; a.hist_cap = hist_cap;
    a.den = den; a.ap_out = ap_out; a.partsA = partsA; a.lc = lc;
    a.slots = sh.slots; a.abort_dev = sh.abort_dev; a.abort_host = sh.abort_host;
    for (int j = 0; j < SB_MAX_ND; j++) a.ps[j] = close_ps ? close_ps[j < nd ? j : 0] : nullptr;
    a.p_out = close_p_out; a.xvec = close_x;
    a.xr_out = xr_out; a.xr_den_slot = xr_den_slot; a.xr_slot = xr_slot; a.partsR_out = partsR_out;
    a.gen0 = exchange_take_generations(3);
    a.test_stall = getenv("MGCR_TEST_STEPBUILD_STALL") ? atoi(getenv("MGCR_TEST_STEPBUILD_STALL")) : 0;
    a.spin_limit = getenv("MGCR_TEST_RESIDENT_SPIN_LIMIT") ? atoi(getenv("MGCR_TEST_RESIDENT_SPIN_LIMIT")) : RES_SPIN_LIMIT;
    const unsigned grid = (unsigned)g;
    const size_t lds_bytes = sb_lds_bytes(A, g);
    const void *kernel = sb_kernel(nd, xr_out != nullptr, close_ps != nullptr, sb_real_enabled() && a.m.realv);
    MGCR_CHECK(launch_is_coresident(kernel, RED_THREADS, lds_bytes, g), MGCR_ERR_INVALID, "csr_step_build: launch would not be co-resident");
    void *kargs[1] = {(void *)&a};
    MGCR_HIP(hipLaunchKernel(kernel, dim3(grid), dim3(RED_THREADS), kargs, lds_bytes, ctx().stream));
    MGCR_HIP(hipGetLastError());
    g_stepbuild_launches++;
    return MGCR_OK;
}

}  // namespace mgcr
