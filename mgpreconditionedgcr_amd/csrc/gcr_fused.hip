// GCR step kernel that embeds the operator apply (Sparse / DiracOp stored one thread per row; single GPU, or
// the row block of a distributed matrix once its halo has arrived):
// Ar = A r  and the partial sums of <Ar, Aps_j> (conj on Ar, src/GCR.h:258), j < NDT, in ONE pass — Ar is
// not read back from HBM and the SpMV's dependent id -> table -> gather chain overlaps with the Aps_j
// streams (Poisson 128^3: 36.8 us against 21.9 + 22.6 us for the two kernels).
//
// Same launch shape as gcr.hip's multidot kernel (RED_THREADS-wide workgroups, rows dealt by the shared RowMap,
// per-thread accumulation in ascending row order, block_sum_bcast) and the same per-row arithmetic as the
// SpMV kernels (spmv_dev.h): Ar AND the partial sums have the bits the separate kernels produce
// (tests/test_gpu_parity.py::test_fused_apply_and_dots_same_bits).  Workgroups are renumbered so that
// each XCD works on one contiguous band of rows per trip; partials are indexed by the logical number.  The
// row's pattern id is fetched one trip ahead and the pattern table sits in LDS, so only the gathers are
// a dependent HBM round trip.
//
// Tried and dropped (measured on MI355X, Poisson 128^3): also folding the residual update
// r' = r - alpha Ap into this kernel (r' recomputed for the 7 gathered entries): 65 us against
// 21.9 + 39.8 us — 14 gathers per row and 112-126 VGPRs; staging the workgroup's 1024 entries of r in
// LDS to serve the +-1 / +-n neighbours: slower still (two barriers per trip); temporal instead of
// non-temporal loads of the Aps_j so that build_* finds them in the Infinity Cache: no gain; one contiguous
// chunk of rows per workgroup instead of grid-stride (x then crosses XCD bands less: 123 instead of 134 MB
// of HBM traffic per launch by PMC): 9 % slower — 512 separate 16-KiB-wide fronts instead of one sweep.  What
// does pay, for operators whose rows reach far (256^3), is one band per XCD swept by its workgroups together:
// RowMap in gcr_dev.h.
#include "internal.h"
#include "reduce.h"
#include "spmv_dev.h"
#include "gcr_dev.h"
#include "pw_tail_dev.h"

namespace mgcr {

struct DotVecs {
    const cplx *v[FND];
};

// PW (row block of a distributed matrix whose scalars travel by peer writes): the launch's last workgroup also folds the partials
// — |r|^2 of the residual update that ran before included — and sums them over the ranks (pw_tail_dev.h)
template <int MODE, int WT, int NDT, bool PW = false>
__global__ void __launch_bounds__(RED_THREADS, ((MODE == 1 || (MODE == 3 && WT <= 7) || MODE == 4) && NDT <= 5 ? 8 : 4)) step_apply_kernel(RowMat m, const cplx *__restrict__ x, cplx *__restrict__ y,
                                                                 DotVecs d, int64_t n, int nlogical, RowMap rm,
                                                                 double *__restrict__ parts, const int *__restrict__ skip, int skip_it, PwTail pw) {
    __shared__ double lds[2 * NDT * 17];
    extern __shared__ __attribute__((aligned(16))) unsigned char step_smem[];
    if (!PW && skip && skip[0] < skip[1] + skip_it) return;
    // logical workgroup number: XCD x (physical b & 7) owns the band [x * per, (x + 1) * per)
    const int lb = logical_workgroup(rm, (int)blockIdx.x, (int)gridDim.x);
    if (lb >= nlogical) return;
    if (PW && skip && skip[0] < skip[1] + skip_it) {
        // a stopped solve's launches do nothing — but the ranks' exchanges stay in lockstep: every reduction that is launched is
        // one exchange on every rank, whether it runs here or as a fold launch (which never skips), whatever it carries
        pw_tail(pw, nlogical);
        return;
    }
    const int32_t W = WT ? WT : m.W;
    // rows of this logical workgroup's threads: RowMap (gcr_dev.h), the map multidot_kernel uses
    int64_t i, end, stride;
    row_range(rm, lb, nlogical, n, &i, &end, &stride);
    int32_t t0 = 0;
    if ((MODE == 1 || MODE == 2) && i < end) t0 = (int32_t)__builtin_nontemporal_load(m.pid + i) * W;
    PatLds pl{nullptr, nullptr, nullptr};
    if (MODE == 1) pl = stage_patterns(m, step_smem);
    double v[2 * NDT];
#pragma unroll
    for (int j = 0; j < 2 * NDT; j++) v[j] = 0.;
    for (; i < end; i += stride) {
        int32_t t0_next = 0;
        if ((MODE == 1 || MODE == 2) && i + stride < end) t0_next = (int32_t)__builtin_nontemporal_load(m.pid + i + stride) * W;
        const cplx sum = fused_row_product<MODE, WT>(m, i, t0, pl, [&](int32_t j) -> cplx { return gather_x(x, m.xh, m.n_own, j); });
        const cplx yi = m.shift ? csub(x[i], cmul(m.k, sum)) : sum;
        y[i] = yi;
        // the direction streams are loaded only now (the scheduler must not hoist them): the gathers and
        // these loads then never hold registers at the same time, the kernel fits 64 VGPRs up to NDT = 5
        // and two 1024-thread workgroups share a CU — 36.8 against 39.8 us at 128^3
        __builtin_amdgcn_sched_barrier(0);
        cplx b[NDT];
#pragma unroll
        for (int j = 0; j < NDT; j++) b[j] = ld_stream<true>(d.v[j] + i);
#pragma unroll
        for (int j = 0; j < NDT; j++) {
            cplx t = cconj_mul(yi, b[j]);
            v[2 * j] += t.x;
            v[2 * j + 1] += t.y;
        }
        t0 = t0_next;
    }
    const double mine = block_sum_owner<2 * NDT>(v, lds);
    if (threadIdx.x < 2 * NDT) parts[(size_t)threadIdx.x * RED_MAX_BLOCKS + lb] = mine;
    if (PW) pw_tail(pw, nlogical);
}


// The same step for an operator with a stencil view whose near slots are slots 1..5 of 7 (a 3-D grid up to n = 256: +-1, +-n
// within STEN_TILE / 2 rows; spmv.hip sten_try), with the workgroup's 1024 entries of x per trip staged in an LDS window
// (+ halo) that serves those five slots: a row then issues ONE coalesced load of its own entry, the two far gathers and
// — registers now allowing it — its direction streams TOGETHER, i.e. one memory round trip per trip instead of two, and
// 3.5 instead of 7 loads per row go through L1 / L2.  Two window buffers alternate, so one barrier per trip suffices (a
// thread can only start writing trip k + 1's buffer — trip k - 1's — after every thread has passed trip k's barrier, i.e.
// finished reading trip k - 1).  Same row arithmetic and the same per-thread accumulation order as step_apply_kernel:
// same bits.  Measured on MI355X (fused apply + dots, microseconds): Poisson 256^3 275 against 312 — every plane of x is
// otherwise fetched about twice there —, 128^3 35.7 against 34.1: the window is used where rows reach at least 2^15 rows
// (csr_step_apply).  EARLY (the direction streams requested together with the gathers, one memory round trip per
// trip) spills at 64 VGPRs and loses: 42 us at 128^3; kept as a switch.
// CARRY (banded row map whose step IS the reach of the two far slots — a 256 x 256 x Z grid with 64 workgroups per band: the thread
// that owns row i owned row i - n^2 one trip earlier and will own row i + n^2 one trip later): the far gathers disappear — the entry
// requested for the NEXT trip is this trip's + n^2 neighbour, this trip's own entry is the next trip's - n^2 neighbour.  Each trip
// then requests ONE entry per row (+ the halo) and every plane of x passes through the XCD's L2 once instead of three times.
// Same values in the same places: same bits.
template <int NS, bool RARE, int NDT, bool PW = false, bool CARRY = false>
__global__ void __launch_bounds__(RED_THREADS, (NDT <= 5 ? 8 : 4)) step_apply_tile_kernel(RowMat m, const cplx *__restrict__ x, cplx *__restrict__ y,
                                                                                        DotVecs d, int64_t n, int nlogical, RowMap rm,
                                                                                        double *__restrict__ parts, const int *__restrict__ skip, int skip_it, PwTail pw) {
    __shared__ double lds[2 * NDT * 17];
    extern __shared__ __attribute__((aligned(16))) unsigned char step_smem[];
    constexpr unsigned NEAR = 0x3eu;
    constexpr int NC = STEN_COMMON;
    constexpr bool EARLY = false;
    static_assert(NS == 7 || (RARE && NS == 9), "7 common slots, optionally 2 rare ones behind them");
    if (!PW && skip && skip[0] < skip[1] + skip_it) return;
    const int lb = logical_workgroup(rm, (int)blockIdx.x, (int)gridDim.x);
    if (lb >= nlogical) return;
    if (PW && skip && skip[0] < skip[1] + skip_it) {   // (see step_apply_kernel)
        pw_tail(pw, nlogical);
        return;
    }
    int64_t i, end, stride;
    bool row_ok;   // (false: a thread of a plane's short last tile beyond the plane's end — it fills its window entry, nothing else)
    row_range(rm, lb, nlogical, n, &i, &end, &stride, &row_ok);
    const int32_t H = m.sten_halo_f;
    cplx *win = reinterpret_cast<cplx *>(step_smem);   // 2 x [H + RED_THREADS + H]
    const int wlen = RED_THREADS + 2 * H;
    auto clampj = [&](int64_t j) -> int32_t { return (int32_t)(j < 0 ? 0 : j > m.sten_last ? m.sten_last : j); };
    const int lane = (int)(threadIdx.x & 63);
    double v[2 * NDT];
#pragma unroll
    for (int j = 0; j < 2 * NDT; j++) v[j] = 0.;
    int buf = 0;
    cplx c_prev = make_double2(0., 0.), c_cur = c_prev;
    if (CARRY) {
        c_prev = gather_x(x, m.xh, m.n_own, clampj(i + m.sten_off[0]));
        c_cur = gather_x(x, m.xh, m.n_own, clampj(i));
        // CARRY writes a trip's window ONE TRIP AHEAD (the first one here): the next trip's halo is requested together with the next
        // trip's own entry, i.e. while the workgroup that owns those rows requests them too — one fetch from memory for both (requested a
        // trip later the lines had left the L2 again: PMC 1.08-1.23x the model's bytes at 256^3, 1.03-1.06x this way)
        win[H + threadIdx.x] = c_cur;
        if ((int)threadIdx.x < 2 * H) {
            const int t = (int)threadIdx.x;
            win[t < H ? t : RED_THREADS + t] = gather_x(x, m.xh, m.n_own, clampj(t < H ? i - threadIdx.x - H + t : i - threadIdx.x + RED_THREADS + (t - H)));
        }
    }
    for (int64_t base = i - threadIdx.x; base < end; base += stride, i += stride, buf ^= 1) {   // uniform trip count per workgroup
        const bool live = row_ok && i < end;
        const int32_t wave = __builtin_amdgcn_readfirstlane((int32_t)(i >> 6));
        const sten_planes_ptr pp = sten_wave_planes(m, wave < m.sten_nwaves ? wave : m.sten_nwaves);
        uint64_t pl[NS];
#pragma unroll
        for (int c = 0; c < NS; c++) pl[c] = pp[c];
        const cplx far6 = gather_x(x, m.xh, m.n_own, clampj(i + m.sten_off[6]));   // CARRY: the next trip's own entry
        const cplx own = CARRY ? c_cur : gather_x(x, m.xh, m.n_own, clampj(i));
        const cplx far0 = CARRY ? c_prev : gather_x(x, m.xh, m.n_own, clampj(i + m.sten_off[0]));
        cplx halo = make_double2(0., 0.);
        int hidx = -1;
        if ((int)threadIdx.x < 2 * H) {
            const int t = (int)threadIdx.x;
            const int64_t hb = CARRY ? base + stride : base;   // CARRY: the NEXT trip's halo
            halo = gather_x(x, m.xh, m.n_own, clampj(t < H ? hb - H + t : hb + RED_THREADS + (t - H)));
            hidx = t < H ? t : RED_THREADS + t;
        }
        cplx b[NDT];
        if (EARLY) {
#pragma unroll
            for (int j = 0; j < NDT; j++) b[j] = live ? ld_stream<true>(d.v[j] + i) : make_double2(0., 0.);
        }
        __builtin_amdgcn_sched_barrier(0);   // everything above is in flight before anything is waited for
        cplx *sx = win + buf * wlen;
        if (!CARRY) {
            sx[H + threadIdx.x] = own;
            if (hidx >= 0) sx[hidx] = halo;
        }
        __syncthreads();
        if (CARRY) {   // the next trip's window, into the buffer the previous trip read (every thread is past that)
            cplx *sn = win + (buf ^ 1) * wlen;
            sn[H + threadIdx.x] = far6;
            if (hidx >= 0) sn[hidx] = halo;
        }
        cplx sum = make_double2(0., 0.);
        if constexpr (RARE) sum = sten_pre_sum<(CARRY ? 1 : -1)>(m, i, pl[NC], lane, [&](int32_t j) -> cplx { return gather_x(x, m.xh, m.n_own, j); });
#pragma unroll
        for (int c = 0; c < NC; c++) {
            const cplx xv = (NEAR >> c & 1u) ? sx[H + (int)threadIdx.x + m.sten_off[c]] : (c == 0 ? far0 : far6);
            const bool on = (pl[c] >> lane & 1ull) != 0ull;
            const cplx nsum = cadd(sum, sten_term<(CARRY ? 1 : -1)>(m, c, xv));
            sum.x = on ? nsum.x : sum.x;
            sum.y = on ? nsum.y : sum.y;
        }
        if (RARE) {
#pragma unroll
            for (int c = NC; c < NS; c++)
                if (pl[c] != 0ull && !(c == NC && m.sten_pre)) {   // wave-uniform: a wave of a boundary plane
                    const cplx xr = gather_x(x, m.xh, m.n_own, clampj(i + m.sten_off[c]));
                    const bool on = (pl[c] >> lane & 1ull) != 0ull;
                    const cplx nsum = cadd(sum, sten_term<(CARRY ? 1 : -1)>(m, c, xr));
                    sum.x = on ? nsum.x : sum.x;
                    sum.y = on ? nsum.y : sum.y;
                }
        }
        const cplx yi = m.shift ? csub(own, cmul(m.k, sum)) : sum;
        if (live) y[i] = yi;
        if (!EARLY) {
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < NDT; j++) b[j] = live ? ld_stream<true>(d.v[j] + i) : make_double2(0., 0.);
        }
        if (live) {
#pragma unroll
            for (int j = 0; j < NDT; j++) {
                cplx t = cconj_mul(yi, b[j]);
                v[2 * j] += t.x;
                v[2 * j + 1] += t.y;
            }
        }
        if (CARRY) { c_prev = own; c_cur = far6; }
    }
    const double mine = block_sum_owner<2 * NDT>(v, lds);
    if (threadIdx.x < 2 * NDT) parts[(size_t)threadIdx.x * RED_MAX_BLOCKS + lb] = mine;
    if (PW) pw_tail(pw, nlogical);
}


// Latency regime (systems of up to 2^19 rows: every kernel is one wave of workgroups and costs its launch, a fold,
// one memory round trip and a reduction, ~5 us, whatever it moves): the residual update of the step runs INSIDE the
// apply kernel.  alpha is folded from the <r,Ap>, <Ap,Ap> partials, r' = r - alpha Ap is formed for the row itself
// (stored to the residual ring, |r'|^2 summed) and again, with the same expression hence the same bits, for every
// gathered neighbour; then Ar' and the beta dots as in step_apply_kernel.  One launch less per iteration: worth
// 1.3 us of ~32 per iteration at 64^3 (the kernel itself gets ~3.7 us longer), 1.7 % of a 256^3 V-cycle.  At 128^3 the
// doubled gathers lose (65 us against 21.9 + 39.8), hence the size limit.  Lean cycles only (r' never lands on r),
// single GPU, rows of at most 8 entries.
template <int MODE, int WT, int NDT>
__global__ void __launch_bounds__(RED_THREADS, 4) step_apply_xr_kernel(RowMat m, const cplx *__restrict__ r_in, const cplx *__restrict__ ap,
                                                                       cplx *__restrict__ r_out, cplx *__restrict__ y, DotVecs d, int64_t n,
                                                                       int nlogical, RowMap rm, double *__restrict__ parts,
                                                                       double *__restrict__ partsR, DevState *__restrict__ st, int it,
                                                                       const double *__restrict__ partsA, int nblkA, int strideA,
                                                                       cplx *__restrict__ den_slot, int slot, LeanCoef *__restrict__ lc) {
    __shared__ double lds[(2 * NDT + 1) * 17 > 4 * 17 ? (2 * NDT + 1) * 17 : 4 * 17];
    extern __shared__ __attribute__((aligned(16))) unsigned char step_smem[];
    if (st->stop_at < st->base + it) return;
    const int lb = logical_workgroup(rm, (int)blockIdx.x, (int)gridDim.x);
    // alpha and its bookkeeping: xr_update_kernel<true, true>'s prologue
    double sa[4];
    fold_partials<4>(partsA, nblkA, strideA, sa, lds);
    const cplx num = make_double2(sa[0], sa[1]), den = make_double2(sa[2], sa[3]);
    const cplx alpha = to_sgpr(cdiv(num, den));
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        *den_slot = den;
        st->npend = slot + 1;
    }
    if (blockIdx.x == 0 && (int)threadIdx.x < LND) lean_pending_update(lc, slot, alpha, (int)threadIdx.x);
    if (lb >= nlogical) return;
    const int32_t W = WT ? WT : m.W;
    int64_t i, end, stride;
    row_range(rm, lb, nlogical, n, &i, &end, &stride);
    int32_t t0 = 0;
    if ((MODE == 1 || MODE == 2) && i < end) t0 = (int32_t)__builtin_nontemporal_load(m.pid + i) * W;
    PatLds pl{nullptr, nullptr, nullptr};
    if (MODE == 1) pl = stage_patterns(m, step_smem);
    double v[2 * NDT + 1];
#pragma unroll
    for (int j = 0; j < 2 * NDT + 1; j++) v[j] = 0.;
    for (; i < end; i += stride) {
        int32_t t0_next = 0;
        if ((MODE == 1 || MODE == 2) && i + stride < end) t0_next = (int32_t)__builtin_nontemporal_load(m.pid + i + stride) * W;
        const cplx sum = fused_row_product<MODE, WT>(m, i, t0, pl, [&](int32_t j) -> cplx { return csub(r_in[j], cmul(alpha, ap[j])); });
        const cplx rn = csub(r_in[i], cmul(alpha, ap[i]));
        r_out[i] = rn;
        v[2 * NDT] += rn.x * rn.x + rn.y * rn.y;
        const cplx yi = m.shift ? csub(rn, cmul(m.k, sum)) : sum;
        y[i] = yi;
        cplx b[NDT];
#pragma unroll
        for (int j = 0; j < NDT; j++) b[j] = ld_stream<true>(d.v[j] + i);
#pragma unroll
        for (int j = 0; j < NDT; j++) {
            cplx t = cconj_mul(yi, b[j]);
            v[2 * j] += t.x;
            v[2 * j + 1] += t.y;
        }
        t0 = t0_next;
    }
    const double mine = block_sum_owner<2 * NDT + 1>(v, lds);
    if (threadIdx.x < 2 * NDT) parts[(size_t)threadIdx.x * RED_MAX_BLOCKS + lb] = mine;
    if (threadIdx.x == 2 * NDT) partsR[lb] = mine;
}

#include "gcr_fused_xr_tile.h"

// Step 0 of a solve whose first direction is its start residual (lean smoothers, gcr.hip alias_p0): Ap_0 = A r_0
// AND the partial sums of <r_0,Ap_0>, <Ap_0,Ap_0>, |r_0|^2 — and |b|^2 when b is not r_0 — in ONE pass, instead
// of the operator apply followed by init3_partials_kernel (+ norm_partials_kernel): r_0 is the row's own x entry
// and Ap_0 the row's result.  Same launch shape and row map as step_apply_kernel; while that map is the plain
// grid-stride (gcr_dev.h: everything below 256^3) the sums have the order, hence the bits, of those two kernels.
template <int MODE, int WT>
__global__ void __launch_bounds__(RED_THREADS, ((MODE == 3 && WT <= 7) || MODE == 4 || MODE == 1 ? 8 : 4)) init_apply_kernel(RowMat m, const cplx *__restrict__ x, cplx *__restrict__ y,
                                                                    const cplx *__restrict__ b, int64_t n, int nlogical, RowMap rm,
                                                                    double *__restrict__ partsA, double *__restrict__ partsR,
                                                                    double *__restrict__ partsN, const int *__restrict__ skip, int skip_it) {
    __shared__ double lds[6 * 17];
    extern __shared__ __attribute__((aligned(16))) unsigned char step_smem[];
    if (skip && skip[0] < skip[1] + skip_it) return;
    const int lb = logical_workgroup(rm, (int)blockIdx.x, (int)gridDim.x);
    if (lb >= nlogical) return;
    const int32_t W = WT ? WT : m.W;
    int64_t i, end, stride;
    row_range(rm, lb, nlogical, n, &i, &end, &stride);
    int32_t t0 = 0;
    if ((MODE == 1 || MODE == 2) && i < end) t0 = (int32_t)__builtin_nontemporal_load(m.pid + i) * W;
    PatLds pl{nullptr, nullptr, nullptr};
    if (MODE == 1) pl = stage_patterns(m, step_smem);
    double v[6] = {0., 0., 0., 0., 0., 0.};
    for (; i < end; i += stride) {
        int32_t t0_next = 0;
        if ((MODE == 1 || MODE == 2) && i + stride < end) t0_next = (int32_t)__builtin_nontemporal_load(m.pid + i + stride) * W;
        const cplx sum = fused_row_product<MODE, WT>(m, i, t0, pl, [&](int32_t j) -> cplx { return gather_x(x, m.xh, m.n_own, j); });
        const cplx rv = x[i];
        const cplx yi = m.shift ? csub(rv, cmul(m.k, sum)) : sum;
        y[i] = yi;
        v[4] += rv.x * rv.x + rv.y * rv.y;
        const cplx t = cconj_mul(rv, yi);
        v[0] += t.x; v[1] += t.y;
        const cplx u = cconj_mul(yi, yi);
        v[2] += u.x; v[3] += u.y;
        if (b) {
            const cplx bv = ld_stream<true>(b + i);
            v[5] += bv.x * bv.x + bv.y * bv.y;
        }
        t0 = t0_next;
    }
    const double mine = block_sum_owner<6>(v, lds);
    if (threadIdx.x < 4) partsA[(size_t)threadIdx.x * RED_MAX_BLOCKS + lb] = mine;
    // thread 4 owns |r_0|^2, thread 5 |b|^2 (when b is given)
    if (threadIdx.x == 4) {
        partsR[lb] = mine;
        if (!b && partsN) partsN[lb] = mine;   // (partsN == nullptr: the caller has |b|^2 already — gcr.hip bnorm_src)
    }
    if (threadIdx.x == 5 && b) partsN[lb] = mine;
}


// init_apply_kernel with the LDS window of step_apply_tile_kernel (same conditions, same bits)
template <int NS, bool RARE, bool CARRY = false>
__global__ void __launch_bounds__(RED_THREADS, 8) init_apply_tile_kernel(RowMat m, const cplx *__restrict__ x, cplx *__restrict__ y,
                                                                         const cplx *__restrict__ b, int64_t n, int nlogical, RowMap rm,
                                                                         double *__restrict__ partsA, double *__restrict__ partsR,
                                                                         double *__restrict__ partsN, const int *__restrict__ skip, int skip_it) {
    __shared__ double lds[6 * 17];
    extern __shared__ __attribute__((aligned(16))) unsigned char step_smem[];
    constexpr unsigned NEAR = 0x3eu;
    constexpr int NC = STEN_COMMON;
    static_assert(NS == 7 || (RARE && NS == 9), "7 common slots, optionally 2 rare ones behind them");
    if (skip && skip[0] < skip[1] + skip_it) return;
    const int lb = logical_workgroup(rm, (int)blockIdx.x, (int)gridDim.x);
    if (lb >= nlogical) return;
    int64_t i, end, stride;
    bool row_ok;   // (false: a thread of a plane's short last tile beyond the plane's end — it fills its window entry, nothing else)
    row_range(rm, lb, nlogical, n, &i, &end, &stride, &row_ok);
    const int32_t H = m.sten_halo_f;
    cplx *win = reinterpret_cast<cplx *>(step_smem);
    const int wlen = RED_THREADS + 2 * H;
    auto clampj = [&](int64_t j) -> int32_t { return (int32_t)(j < 0 ? 0 : j > m.sten_last ? m.sten_last : j); };
    const int lane = (int)(threadIdx.x & 63);
    double v[6] = {0., 0., 0., 0., 0., 0.};
    int buf = 0;
    cplx c_prev = make_double2(0., 0.), c_cur = c_prev;
    if (CARRY) {   // (step_apply_tile_kernel: far neighbours carried, windows written one trip ahead)
        c_prev = gather_x(x, m.xh, m.n_own, clampj(i + m.sten_off[0]));
        c_cur = gather_x(x, m.xh, m.n_own, clampj(i));
        win[H + threadIdx.x] = c_cur;
        if ((int)threadIdx.x < 2 * H) {
            const int t = (int)threadIdx.x;
            win[t < H ? t : RED_THREADS + t] = gather_x(x, m.xh, m.n_own, clampj(t < H ? i - threadIdx.x - H + t : i - threadIdx.x + RED_THREADS + (t - H)));
        }
    }
    for (int64_t base = i - threadIdx.x; base < end; base += stride, i += stride, buf ^= 1) {
        const bool live = row_ok && i < end;
        const int32_t wave = __builtin_amdgcn_readfirstlane((int32_t)(i >> 6));
        const sten_planes_ptr pp = sten_wave_planes(m, wave < m.sten_nwaves ? wave : m.sten_nwaves);
        uint64_t pl[NS];
#pragma unroll
        for (int c = 0; c < NS; c++) pl[c] = pp[c];
        const cplx far6 = gather_x(x, m.xh, m.n_own, clampj(i + m.sten_off[6]));   // CARRY: the next trip's own entry
        const cplx own = CARRY ? c_cur : gather_x(x, m.xh, m.n_own, clampj(i));
        const cplx far0 = CARRY ? c_prev : gather_x(x, m.xh, m.n_own, clampj(i + m.sten_off[0]));
        cplx halo = make_double2(0., 0.);
        int hidx = -1;
        if ((int)threadIdx.x < 2 * H) {
            const int t = (int)threadIdx.x;
            const int64_t hb = CARRY ? base + stride : base;   // CARRY: the NEXT trip's halo
            halo = gather_x(x, m.xh, m.n_own, clampj(t < H ? hb - H + t : hb + RED_THREADS + (t - H)));
            hidx = t < H ? t : RED_THREADS + t;
        }
        __builtin_amdgcn_sched_barrier(0);
        cplx *sx = win + buf * wlen;
        if (!CARRY) {
            sx[H + threadIdx.x] = own;
            if (hidx >= 0) sx[hidx] = halo;
        }
        __syncthreads();
        if (CARRY) {   // the next trip's window, into the buffer the previous trip read (every thread is past that)
            cplx *sn = win + (buf ^ 1) * wlen;
            sn[H + threadIdx.x] = far6;
            if (hidx >= 0) sn[hidx] = halo;
        }
        cplx sum = make_double2(0., 0.);
        if constexpr (RARE) sum = sten_pre_sum<(CARRY ? 1 : -1)>(m, i, pl[NC], lane, [&](int32_t j) -> cplx { return gather_x(x, m.xh, m.n_own, j); });
#pragma unroll
        for (int c = 0; c < NC; c++) {
            const cplx xv = (NEAR >> c & 1u) ? sx[H + (int)threadIdx.x + m.sten_off[c]] : (c == 0 ? far0 : far6);
            const bool on = (pl[c] >> lane & 1ull) != 0ull;
            const cplx nsum = cadd(sum, sten_term<(CARRY ? 1 : -1)>(m, c, xv));
            sum.x = on ? nsum.x : sum.x;
            sum.y = on ? nsum.y : sum.y;
        }
        if (RARE) {
#pragma unroll
            for (int c = NC; c < NS; c++)
                if (pl[c] != 0ull && !(c == NC && m.sten_pre)) {
                    const cplx xr = gather_x(x, m.xh, m.n_own, clampj(i + m.sten_off[c]));
                    const bool on = (pl[c] >> lane & 1ull) != 0ull;
                    const cplx nsum = cadd(sum, sten_term<(CARRY ? 1 : -1)>(m, c, xr));
                    sum.x = on ? nsum.x : sum.x;
                    sum.y = on ? nsum.y : sum.y;
                }
        }
        if (live) {
            const cplx rv = own;
            const cplx yi = m.shift ? csub(rv, cmul(m.k, sum)) : sum;
            y[i] = yi;
            v[4] += rv.x * rv.x + rv.y * rv.y;
            const cplx t = cconj_mul(rv, yi);
            v[0] += t.x; v[1] += t.y;
            const cplx u = cconj_mul(yi, yi);
            v[2] += u.x; v[3] += u.y;
            if (b) {
                const cplx bv = ld_stream<true>(b + i);
                v[5] += bv.x * bv.x + bv.y * bv.y;
            }
        }
        if (CARRY) { c_prev = own; c_cur = far6; }
    }
    const double mine = block_sum_owner<6>(v, lds);
    if (threadIdx.x < 4) partsA[(size_t)threadIdx.x * RED_MAX_BLOCKS + lb] = mine;
    if (threadIdx.x == 4) {
        partsR[lb] = mine;
        if (!b && partsN) partsN[lb] = mine;   // (partsN == nullptr: the caller has |b|^2 already — gcr.hip bnorm_src)
    }
    if (threadIdx.x == 5 && b) partsN[lb] = mine;
}

// The stand-alone apply y = A x  /  y = w - k A x  (w == nullptr: w is x) in the carried-window form of the two kernels above (CARRY: a
// 256 x 256 x Z grid, real coefficients): persistent workgroups walk their band plane by plane, every entry of x is requested once
// (+ the halo) and the far neighbours come from the thread's own previous / next trip.  Same row arithmetic as sten_spmv_tile
// (spmv.hip): same bits.
template <bool SHIFT>
__global__ void __launch_bounds__(RED_THREADS, 8) sten_apply_carry_kernel(RowMat m, const cplx *__restrict__ x, cplx *__restrict__ y,
                                                                          const cplx *__restrict__ w, int64_t n, int nlogical, RowMap rm,
                                                                          const int *__restrict__ skip, int skip_it) {
    extern __shared__ __attribute__((aligned(16))) unsigned char step_smem[];
    constexpr unsigned NEAR = 0x3eu;
    constexpr int NC = STEN_COMMON;
    if (skip && skip[0] < skip[1] + skip_it) return;
    const int lb = logical_workgroup(rm, (int)blockIdx.x, (int)gridDim.x);
    if (lb >= nlogical) return;
    int64_t i, end, stride;
    bool row_ok;   // (false: a thread of a plane's short last tile beyond the plane's end — it fills its window entry, nothing else)
    row_range(rm, lb, nlogical, n, &i, &end, &stride, &row_ok);
    const int32_t H = m.sten_halo_f;
    cplx *win = reinterpret_cast<cplx *>(step_smem);
    const int wlen = RED_THREADS + 2 * H;
    auto clampj = [&](int64_t j) -> int32_t { return (int32_t)(j < 0 ? 0 : j > m.sten_last ? m.sten_last : j); };
    const int lane = (int)(threadIdx.x & 63);
    int buf = 0;
    cplx c_prev = x[clampj(i + m.sten_off[0])], c_cur = x[clampj(i)];
    {   // windows are written one trip ahead (step_apply_tile_kernel CARRY): the first one here
        win[H + threadIdx.x] = c_cur;
        if ((int)threadIdx.x < 2 * H) {
            const int t = (int)threadIdx.x;
            win[t < H ? t : RED_THREADS + t] = x[clampj(t < H ? i - threadIdx.x - H + t : i - threadIdx.x + RED_THREADS + (t - H))];
        }
    }
    for (int64_t base = i - threadIdx.x; base < end; base += stride, i += stride, buf ^= 1) {
        const bool live = row_ok && i < end;
        const int32_t wave = __builtin_amdgcn_readfirstlane((int32_t)(i >> 6));
        const sten_planes_ptr pp = sten_wave_planes(m, wave < m.sten_nwaves ? wave : m.sten_nwaves);
        uint64_t pl[NC];
#pragma unroll
        for (int c = 0; c < NC; c++) pl[c] = pp[c];
        const cplx far6 = x[clampj(i + m.sten_off[6])];   // the next trip's own entry
        cplx wv = make_double2(0., 0.);
        if (SHIFT && w && live) wv = ld_stream<true>(w + i);
        cplx halo = make_double2(0., 0.);
        int hidx = -1;
        if ((int)threadIdx.x < 2 * H) {
            const int t = (int)threadIdx.x;
            halo = x[clampj(t < H ? base + stride - H + t : base + stride + RED_THREADS + (t - H))];   // the NEXT trip's halo
            hidx = t < H ? t : RED_THREADS + t;
        }
        __builtin_amdgcn_sched_barrier(0);
        cplx *sx = win + buf * wlen;
        __syncthreads();
        {
            cplx *sn = win + (buf ^ 1) * wlen;
            sn[H + threadIdx.x] = far6;
            if (hidx >= 0) sn[hidx] = halo;
        }
        cplx sum = make_double2(0., 0.);
#pragma unroll
        for (int c = 0; c < NC; c++) {
            const cplx xv = (NEAR >> c & 1u) ? sx[H + (int)threadIdx.x + m.sten_off[c]] : (c == 0 ? c_prev : far6);
            const bool on = (pl[c] >> lane & 1ull) != 0ull;
            const cplx nsum = cadd(sum, sten_term<1>(m, c, xv));
            sum.x = on ? nsum.x : sum.x;
            sum.y = on ? nsum.y : sum.y;
        }
        if (live) y[i] = SHIFT ? csub(w ? wv : c_cur, cmul(m.k, sum)) : sum;
        c_prev = c_cur;
        c_cur = far6;
    }
}

static int g_fuse = -1;
static bool fuse_enabled() {
    if (g_fuse < 0) g_fuse = !(getenv("MGCR_FUSE") && atoi(getenv("MGCR_FUSE")) == 0);
    return g_fuse != 0;
}
bool set_fuse_enabled(bool on) {
    bool prev = fuse_enabled();
    g_fuse = on ? 1 : 0;
    return prev;
}
bool csr_fusable(const CsrDev &A, const DistCsr *dist) {
    if (A.pat_mode == 1 && (int64_t)A.npat * A.W * 20 > 48 * 1024) return false;  // pattern table must fit LDS
    // a row block of a distributed matrix qualifies when its halo exchange is ordered on the compute stream
    // (the default): the one kernel then simply runs after it
    if (dist ? dist_halo_overlaps() : A.nrow != A.ncol) return false;
    return fuse_enabled() && A.L == 1 && A.n_tail_rows == 0 && A.nrow >= 1 && A.W >= 1;
}

template <int MODE, int WT, bool PW = false>
static void launch_nd(int nd, unsigned grid, size_t lds_bytes, const RowMat &m, const cplx *x, cplx *y, const DotVecs &d, int64_t n,
                      int g, double *parts, SkipRef sk, const RowMap &rm, const PwTail &pw = PwTail{}) {
#define SK(NDT)                                                                                                          \
    hipLaunchKernelGGL((step_apply_kernel<MODE, WT, NDT, PW>), dim3(grid), dim3(RED_THREADS), lds_bytes, ctx().stream, m, x, y, d, n, \
                       g, rm, parts, sk.p, sk.it, pw)
    switch (nd) {
        case 1: SK(1); break;
        case 2: SK(2); break;
        case 3: SK(3); break;
        case 4: SK(4); break;
        case 5: SK(5); break;
        case 6: SK(6); break;
        case 7: SK(7); break;
        case 8: SK(8); break;
        case 9: SK(9); break;
        default: SK(10); break;
    }
#undef SK
}

template <int NS, bool RARE, bool PW = false, bool CARRY = false>
static void launch_tile_nd(int nd, unsigned grid, size_t lds_bytes, const RowMat &m, const cplx *x, cplx *y, const DotVecs &d, int64_t n,
                           int g, double *parts, SkipRef sk, const RowMap &rm, const PwTail &pw = PwTail{}) {
#define SKT(NDT)                                                                                                              \
    do {                                                                                                                      \
        static bool big_lds = false;   /* up to 2 x 2048 x 16 B of window + the reduction scratch: above the 64 KiB default */ \
        if (!big_lds) {                                                                                                       \
            hipFuncSetAttribute((const void *)step_apply_tile_kernel<NS, RARE, NDT, PW, CARRY>, hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024); \
            big_lds = true;                                                                                                   \
        }                                                                                                                     \
        hipLaunchKernelGGL((step_apply_tile_kernel<NS, RARE, NDT, PW, CARRY>), dim3(grid), dim3(RED_THREADS), lds_bytes, ctx().stream, m, x, y, d, n, \
                           g, rm, parts, sk.p, sk.it, pw);                                                                    \
    } while (0)
    switch (nd) {
        case 1: SKT(1); break;
        case 2: SKT(2); break;
        case 3: SKT(3); break;
        case 4: SKT(4); break;
        case 5: SKT(5); break;
        case 6: SKT(6); break;
        case 7: SKT(7); break;
        case 8: SKT(8); break;
        case 9: SKT(9); break;
        default: SKT(10); break;
    }
#undef SKT
}
static int64_t fused_tile_min_reach() {
    static const int64_t r = getenv("MGCR_FUSED_TILE_REACH") ? atoll(getenv("MGCR_FUSED_TILE_REACH")) : (int64_t)1 << 15;
    return r;
}
// the far slots of the 7-slot view are exactly one step of the banded row map away (step_apply_tile_kernel: CARRY)
// (the CARRY instantiations are also the ones compiled for real stencil coefficients — no per-slot real / complex decision, 14 scalar
// registers less: what lets them keep 64 vector registers without spills)
static bool tile_carry(const CsrDev &A, const RowMap &rm) {
    static const bool on = !(getenv("MGCR_TILE_CARRY") && atoi(getenv("MGCR_TILE_CARRY")) == 0);
    const int64_t step = rm.plane ? rm.plane : (int64_t)rm.per * RED_THREADS;
    // (7 slots, or the 7 + 2 of a distributed row block: its halo columns are rarely present slots of their own — the carried far
    // values of a boundary plane's rows are masked like the gathered ones were)
    return on && rm.band != 0 && A.sten_off[6] == step && A.sten_off[0] == -step && sten_slots(A) == (A.sten_rare ? 9 : 7) &&
           row_mat(A, false, cplx{0., 0.}).realv;
}
static bool fused_tile_enabled() {
    static const bool on = !(getenv("MGCR_FUSED_TILE") && atoi(getenv("MGCR_FUSED_TILE")) == 0);
    return on;
}

// y = A x (or x - k A x) + partials of <y, vecs_j>, j < nd <= FND, laid out like gcr.hip's partsB;
// dist: A is this rank's row block, the halo exchange of x is enqueued first
// pw != nullptr (the caller got it from comm_pw_tail_begin and filled in the slabs): the kernel's last workgroup folds and exchanges
// — only where csr_step_apply_has_pw_tail(A, dist) says the instantiation exists
bool csr_step_apply_has_pw_tail(const CsrDev &A, const DistCsr *dist) {
    // every storage the fused kernels read (stencil view with or without rare slots, dictionary, slab).  Mixed decisions between ranks
    // would still interoperate — a fold inside a kernel and a fold launch speak the same mailbox protocol, one sequence number per
    // reduction either way — but there is no reason to have them
    (void)A;
    return dist != nullptr;
}
int csr_step_apply(const CsrDev &A, const cplx *x, cplx *y, bool shift, cplx k, const cplx *const *vecs, int nd, double *parts,
                   DistCsr *dist, const RowMap &rm, const PwTail *pw) {
    MGCR_CHECK(x != y, MGCR_ERR_INVALID, "SpMV cannot run in place");
    MGCR_CHECK(nd >= 1 && nd <= FND, MGCR_ERR_INVALID, "csr_step_apply: 1..10 vectors");
    RowMat m = row_mat(A, shift, k);
    if (dist) {
        int64_t ib = 0, ie = 0;
        dist_info(dist, &m.xh, &ib, &ie);
        m.n_own = (int32_t)A.nrow;
        MGCR_TRY(dist_halo_begin(dist, x));
        MGCR_TRY(dist_halo_end(dist));
        m.xh = dist_halo_ptr(dist);
    }
    DotVecs d;
    for (int j = 0; j < FND; j++) d.v[j] = vecs[j < nd ? j : 0];
    const int g = red_grid(A.nrow);
    const unsigned grid = (unsigned)(g >= 64 ? (g + 7) / 8 * 8 : g);  // multiple of 8 => XCD bands
    const size_t lds_bytes = row_mat_lds_bytes(A);
    const SkipRef sk = get_apply_skip();
#define ST_W(MODE)                                                                              \
    do {                                                                                        \
        if (pw && A.W == 7) launch_nd<MODE, 7, true>(nd, grid, lds_bytes, m, x, y, d, A.nrow, g, parts, sk, rm, *pw); \
        else if (pw) launch_nd<MODE, 0, true>(nd, grid, lds_bytes, m, x, y, d, A.nrow, g, parts, sk, rm, *pw);        \
        else if (A.W == 7) launch_nd<MODE, 7>(nd, grid, lds_bytes, m, x, y, d, A.nrow, g, parts, sk, rm); \
        else launch_nd<MODE, 0>(nd, grid, lds_bytes, m, x, y, d, A.nrow, g, parts, sk, rm);          \
    } while (0)
#define ST_S(MODE, NS) launch_nd<MODE, NS>(nd, grid, 0, m, x, y, d, A.nrow, g, parts, sk, rm)
    if (csr_stencil_active(A) && A.sten_near_f == 0x3eu && A.sten_halo_f > 0 && (A.sten_rare || sten_slots(A) == 7) && fused_tile_enabled() &&
        A.reach >= fused_tile_min_reach()) {
        // 3-D stencil: x staged in an LDS window per trip (step_apply_tile_kernel)
        const size_t win = 2 * (size_t)(RED_THREADS + 2 * A.sten_halo_f) * sizeof(cplx);
        const bool carry = tile_carry(A, rm);
        if (A.sten_rare && pw && carry) launch_tile_nd<9, true, true, true>(nd, grid, win, m, x, y, d, A.nrow, g, parts, sk, rm, *pw);
        else if (A.sten_rare && pw) launch_tile_nd<9, true, true>(nd, grid, win, m, x, y, d, A.nrow, g, parts, sk, rm, *pw);
        else if (A.sten_rare && carry) launch_tile_nd<9, true, false, true>(nd, grid, win, m, x, y, d, A.nrow, g, parts, sk, rm);
        else if (A.sten_rare) launch_tile_nd<9, true>(nd, grid, win, m, x, y, d, A.nrow, g, parts, sk, rm);
        else if (pw) launch_tile_nd<7, false, true>(nd, grid, win, m, x, y, d, A.nrow, g, parts, sk, rm, *pw);
        else if (carry) launch_tile_nd<7, false, false, true>(nd, grid, win, m, x, y, d, A.nrow, g, parts, sk, rm);
        else launch_tile_nd<7, false>(nd, grid, win, m, x, y, d, A.nrow, g, parts, sk, rm);
    } else if (csr_stencil_active(A)) {   // MODE 4: rare-tail layout (7 common + 2 rare slots)
        if (A.sten_rare && pw) launch_nd<4, 9, true>(nd, grid, 0, m, x, y, d, A.nrow, g, parts, sk, rm, *pw);
        else if (pw && sten_slots(A) == 7) launch_nd<3, 7, true>(nd, grid, 0, m, x, y, d, A.nrow, g, parts, sk, rm, *pw);
        else if (pw) launch_nd<3, 9, true>(nd, grid, 0, m, x, y, d, A.nrow, g, parts, sk, rm, *pw);
        else if (A.sten_rare) ST_S(4, 9);
        else if (sten_slots(A) == 7) ST_S(3, 7);
        else ST_S(3, 9);
    }
    else if (A.pat_mode == 1) ST_W(1);
    else if (A.pat_mode == 2) ST_W(2);
    else ST_W(0);
#undef ST_S
#undef ST_W
    MGCR_HIP(hipGetLastError());
    return MGCR_OK;
}


template <int MODE, int WT>
static void launch_xr_nd(int nd, unsigned grid, size_t lds_bytes, const RowMat &m, const cplx *r_in, const cplx *ap, cplx *r_out, cplx *y,
                         const DotVecs &d, int64_t n, int g, const RowMap &rm, double *parts, double *partsR, DevState *st, int it,
                         const double *partsA, int nblkA, int strideA, cplx *den_slot, int slot, LeanCoef *lc) {
#define SX(NDT)                                                                                                                       \
    hipLaunchKernelGGL((step_apply_xr_kernel<MODE, WT, NDT>), dim3(grid), dim3(RED_THREADS), lds_bytes, ctx().stream, m, r_in, ap, r_out, y, \
                       d, n, g, rm, parts, partsR, st, it, partsA, nblkA, strideA, den_slot, slot, lc)
    switch (nd) {
        case 1: SX(1); break;
        case 2: SX(2); break;
        case 3: SX(3); break;
        case 4: SX(4); break;
        case 5: SX(5); break;
        case 6: SX(6); break;
        case 7: SX(7); break;
        case 8: SX(8); break;
        case 9: SX(9); break;
        default: SX(10); break;
    }
#undef SX
}

static bool tile_regime(const CsrDev &A) {   // where csr_step_apply / csr_init_apply stage x in the LDS window
    return csr_stencil_active(A) && A.sten_near_f == 0x3eu && A.sten_halo_f > 0 && (A.sten_rare || sten_slots(A) == 7) && fused_tile_enabled() &&
           A.reach >= fused_tile_min_reach();
}
static bool xr_tile_enabled() {
    static const bool on = !(getenv("MGCR_XR_FUSE_TILE") && atoi(getenv("MGCR_XR_FUSE_TILE")) == 0);
    return on;
}
template <int NS, bool RARE>
static void launch_xr_tile_nd(int nd, unsigned grid, size_t lds_bytes, const RowMat &m, const cplx *r_in, const cplx *ap, cplx *r_out, cplx *y,
                              const DotVecs &d, int64_t n, int g, const RowMap &rm, double *parts, double *partsR, DevState *st,
                              int it, const double *partsA, int nblkA, int strideA, cplx *den_slot, int slot, LeanCoef *lc) {
#define SXT(NDT)                                                                                                                       \
    do {                                                                                                                               \
        static bool big_lds = false;                                                                                                   \
        if (!big_lds) {                                                                                                                \
            hipFuncSetAttribute((const void *)step_apply_xr_tile_kernel<NS, RARE, NDT, (NDT <= XR_TILE_APC_NDT)>, hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024); \
            big_lds = true;                                                                                                            \
        }                                                                                                                              \
        hipLaunchKernelGGL((step_apply_xr_tile_kernel<NS, RARE, NDT, (NDT <= XR_TILE_APC_NDT)>), dim3(grid), dim3(RED_THREADS), lds_bytes, ctx().stream, m, r_in, ap, r_out, \
                           y, d, n, g, rm, parts, partsR, st, it, partsA, nblkA, strideA, den_slot, slot, lc);                \
    } while (0)
    switch (nd) {
        case 1: SXT(1); break;
        case 2: SXT(2); break;
        case 3: SXT(3); break;
        case 4: SXT(4); break;
        case 5: SXT(5); break;
        case 6: SXT(6); break;
        case 7: SXT(7); break;
        case 8: SXT(8); break;
        case 9: SXT(9); break;
        default: SXT(10); break;
    }
#undef SXT
}

// 0: the residual update keeps its own launch; 1: latency regime (step_apply_xr_kernel); 2: windowed bandwidth regime
// (step_apply_xr_tile_kernel: |r|^2 is then summed over the apply's row map)
int csr_xr_fuse_kind(const CsrDev &A, const DistCsr *dist) {
    if (!csr_xr_fusable(A, dist)) return 0;
    return tile_regime(A) ? 2 : 1;
}
bool csr_xr_fusable(const CsrDev &A, const DistCsr *dist) {
    static const int64_t limit = getenv("MGCR_XR_FUSE_ROWS") ? atoll(getenv("MGCR_XR_FUSE_ROWS")) : ((int64_t)1 << 19);
    if (!dist && tile_regime(A) && xr_tile_enabled() && csr_fusable(A, dist)) {
        // the windowed form: real stencil coefficients, and the far slots one step of the banded row map away (gcr_fused_xr_tile.h)
        return !A.sten_rare && tile_carry(A, make_row_map(A.nrow, red_grid(A.nrow), A.reach));
    }
    // the gathers double, so short rows only: 64^3 7-point, 19 iterations of GCR(10): 0.633 -> 0.608 ms; the 4x4 sample matrix
    // (39 entries per row) loses 1.6 % and stays with the separate update kernel
    return !dist && A.nrow <= limit && A.W <= 8 && csr_fusable(A, dist);
}

// r_out = r_in - alpha ap (alpha from the partsA partials), y = A r_out (or r_out - k A r_out), partials of <y, vecs_j> and of
// |r_out|^2: xr_update_kernel<true, true> + csr_step_apply in one launch
int csr_step_apply_xr(const CsrDev &A, const cplx *r_in, const cplx *ap, cplx *r_out, cplx *y, bool shift, cplx k, const cplx *const *vecs,
                      int nd, double *parts, double *partsR, DevState *st, int it, const double *partsA, int nblkA, int strideA,
                      cplx *den_slot, int slot, LeanCoef *lc, const RowMap &rm) {
    MGCR_CHECK(r_in != r_out && r_out != y && r_in != y, MGCR_ERR_INVALID, "csr_step_apply_xr: operands must be distinct");
    MGCR_CHECK(nd >= 1 && nd <= FND, MGCR_ERR_INVALID, "csr_step_apply_xr: 1..10 vectors");
    RowMat m = row_mat(A, shift, k);
    DotVecs d;
    for (int j = 0; j < FND; j++) d.v[j] = vecs[j < nd ? j : 0];
    const int g = red_grid(A.nrow);
    const unsigned grid = (unsigned)(g >= 64 ? (g + 7) / 8 * 8 : g);
    const size_t lds_bytes = row_mat_lds_bytes(A);
#define SXW(MODE)                                                                                                                          \
    do {                                                                                                                                   \
        if (A.W == 7) launch_xr_nd<MODE, 7>(nd, grid, lds_bytes, m, r_in, ap, r_out, y, d, A.nrow, g, rm, parts, partsR, st, it, partsA, nblkA, \
                                            strideA, den_slot, slot, lc);                                                                  \
        else launch_xr_nd<MODE, 0>(nd, grid, lds_bytes, m, r_in, ap, r_out, y, d, A.nrow, g, rm, parts, partsR, st, it, partsA, nblkA, strideA,  \
                                   den_slot, slot, lc);                                                                                    \
    } while (0)
#define SXS(MODE, NS) launch_xr_nd<MODE, NS>(nd, grid, 0, m, r_in, ap, r_out, y, d, A.nrow, g, rm, parts, partsR, st, it, partsA, nblkA, strideA, \
                                             den_slot, slot, lc)
    if (tile_regime(A)) {
        const size_t win = 2 * (size_t)(RED_THREADS + 2 * A.sten_halo_f) * sizeof(cplx);
        // (APC instantiations take the newest direction's A p from the update's operands: it must be the last of the dot-product streams —
        // it is, in a lean cycle of up to FND directions; beyond that the kernel requests its streams like any other)
        MGCR_CHECK(!A.sten_rare && tile_carry(A, rm) && (nd > XR_TILE_APC_NDT || d.v[nd - 1] == ap), MGCR_ERR_INVALID,
                   "csr_step_apply_xr: not the windowed form's case");
        launch_xr_tile_nd<7, false>(nd, grid, win, m, r_in, ap, r_out, y, d, A.nrow, g, rm, parts, partsR, st, it, partsA, nblkA, strideA,
                                         den_slot, slot, lc);
    } else if (csr_stencil_active(A)) {
        if (A.sten_rare) SXS(4, 9);
        else if (sten_slots(A) == 7) SXS(3, 7);
        else SXS(3, 9);
    }
    else if (A.pat_mode == 1) SXW(1);
    else if (A.pat_mode == 2) SXW(2);
    else SXW(0);
#undef SXS
#undef SXW
    MGCR_HIP(hipGetLastError());
    return MGCR_OK;
}

// aps0 = A r0 (or r0 - k A r0) + the partials of <r0,aps0>, <aps0,aps0> (partsA), |r0|^2 (partsR) and |b|^2 (partsN;
// b == nullptr: b IS r0); layouts of gcr.hip's init3_partials_kernel; dist as in csr_step_apply
int csr_init_apply(const CsrDev &A, const cplx *r0, cplx *aps0, bool shift, cplx k, const cplx *b, double *partsA, double *partsR,
                   double *partsN, DistCsr *dist, const RowMap &rm) {
    MGCR_CHECK(r0 != aps0, MGCR_ERR_INVALID, "SpMV cannot run in place");
    RowMat m = row_mat(A, shift, k);
    if (dist) {
        m.n_own = (int32_t)A.nrow;
        MGCR_TRY(dist_halo_begin(dist, r0));
        MGCR_TRY(dist_halo_end(dist));
        m.xh = dist_halo_ptr(dist);
    }
    const int g = red_grid(A.nrow);
    const unsigned grid = (unsigned)(g >= 64 ? (g + 7) / 8 * 8 : g);
    const size_t lds_bytes = row_mat_lds_bytes(A);
    const SkipRef sk = get_apply_skip();
#define IA(MODE, WT)                                                                                                            \
    hipLaunchKernelGGL((init_apply_kernel<MODE, WT>), dim3(grid), dim3(RED_THREADS), lds_bytes, ctx().stream, m, r0, aps0, b, A.nrow, g, \
                       rm, partsA, partsR, partsN, sk.p, sk.it)
    if (csr_stencil_active(A) && A.sten_near_f == 0x3eu && A.sten_halo_f > 0 && (A.sten_rare || sten_slots(A) == 7) && fused_tile_enabled() &&
        A.reach >= fused_tile_min_reach()) {
        const size_t win = 2 * (size_t)(RED_THREADS + 2 * A.sten_halo_f) * sizeof(cplx);
#define IAT(NS, RARE, CARRY)                                                                                                        \
    do {                                                                                                                            \
        static bool big_lds = false;                                                                                                \
        if (!big_lds) {                                                                                                             \
            hipFuncSetAttribute((const void *)init_apply_tile_kernel<NS, RARE, CARRY>, hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024); \
            big_lds = true;                                                                                                         \
        }                                                                                                                           \
        hipLaunchKernelGGL((init_apply_tile_kernel<NS, RARE, CARRY>), dim3(grid), dim3(RED_THREADS), win, ctx().stream, m, r0, aps0, b, A.nrow, g, rm, \
                           partsA, partsR, partsN, sk.p, sk.it);                                                                    \
    } while (0)
        if (A.sten_rare && tile_carry(A, rm)) IAT(9, true, true);
        else if (A.sten_rare) IAT(9, true, false);
        else if (tile_carry(A, rm)) IAT(7, false, true);
        else IAT(7, false, false);
#undef IAT
    } else if (csr_stencil_active(A)) {
#define IAS(MODE, NS) hipLaunchKernelGGL((init_apply_kernel<MODE, NS>), dim3(grid), dim3(RED_THREADS), 0, ctx().stream, m, r0, aps0, b, A.nrow, g, \
                                         rm, partsA, partsR, partsN, sk.p, sk.it)
        if (A.sten_rare) IAS(4, 9);
        else if (sten_slots(A) == 7) IAS(3, 7);
        else IAS(3, 9);
#undef IAS
    }
    else if (A.pat_mode == 1) { if (A.W == 7) IA(1, 7); else IA(1, 0); }
    else if (A.pat_mode == 2) { if (A.W == 7) IA(2, 7); else IA(2, 0); }
    else { if (A.W == 7) IA(0, 7); else IA(0, 0); }
#undef IA
    MGCR_HIP(hipGetLastError());
    return MGCR_OK;
}

// y = A x / y = w - k A x in the carried-window form; false: not this kind of operator (the caller takes its own kernels)
bool csr_apply_carry(const CsrDev &A, const cplx *x, cplx *y, bool shift, cplx k, const cplx *w) {
    static const bool on = !(getenv("MGCR_APPLY_CARRY") && atoi(getenv("MGCR_APPLY_CARRY")) == 0);
    // (planes below 64 x 1024 sites — 192^3: x is small enough for the caches to serve spmv.hip's far gathers, 38.6 against 44.2 us)
    if (!on || !tile_regime(A) || A.nrow != A.ncol || A.n_tail_rows || A.reach < 64 * RED_THREADS) return false;
    const int g = red_grid(A.nrow);
    if (g < 64 || g % 8) return false;
    const RowMap rm = make_row_map(A.nrow, g, A.reach);
    if (A.sten_rare || !tile_carry(A, rm)) return false;
    const RowMat m = row_mat(A, shift, k);
    const size_t win = 2 * (size_t)(RED_THREADS + 2 * A.sten_halo_f) * sizeof(cplx);
    const SkipRef sk = get_apply_skip();
    static bool big_lds = false;
    if (!big_lds) {
        (void)hipFuncSetAttribute((const void *)sten_apply_carry_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024);
        (void)hipFuncSetAttribute((const void *)sten_apply_carry_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024);
        big_lds = true;
    }
    if (shift) hipLaunchKernelGGL((sten_apply_carry_kernel<true>), dim3((unsigned)g), dim3(RED_THREADS), win, ctx().stream, m, x, y, w, A.nrow, g, rm, sk.p, sk.it);
    else hipLaunchKernelGGL((sten_apply_carry_kernel<false>), dim3((unsigned)g), dim3(RED_THREADS), win, ctx().stream, m, x, y, (const cplx *)nullptr, A.nrow, g, rm, sk.p, sk.it);
    return hipGetLastError() == hipSuccess;
}

}  // namespace mgcr
