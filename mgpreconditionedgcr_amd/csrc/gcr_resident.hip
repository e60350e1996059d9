// Resident GCR: a whole lean restarted solve (src/GCR.h:158-302) in ONE launch, for systems of at most 1024 rows per
// compute unit (<= 256 workgroups = 262 144 rows on MI355X) — the latency regime, where every kernel of gcr.hip's path
// costs its launch (~4.5 us in a dependent chain) plus a pass over 4 MB vectors that live in the Infinity Cache: the 49
// iterations of the coarsest solve of a 256^3 V-cycle took 49 x (13.6 + 12.2) us; here 49 x 14.7.
//
// What makes one launch possible (tools/barrier_lab.hip, tools/coherent_lab.hip, profiles/r02_barrier_lab.txt,
// r02_coherent_lab.txt):
//   * a device-wide rendezvous costs 1.7 us when every workgroup only STORES a flag of its own and POLLS the others'
//     with relaxed atomics — the 21-57 us of a counter barrier with release / acquire fences are the fences (256
//     workgroups each writing back and invalidating their XCD's L2) and the contended counter;
//   * without fences the data that crosses workgroups has to be coherent by itself: it is written with
//     buffer_store ... sc1 (write-through to the memory side) and read with buffer_load ... sc1 (misses the
//     non-coherent caches), the agent-scope cache policy of a relaxed atomic, which the compiler schedules like any
//     other load (no inline assembly, no hand-placed waits).  tools/coherent_lab.hip checks exactly this on the part.
//
// Data layout: a workgroup of 512 threads owns 1024 consecutive rows, thread t rows t and t + 512 (RPT = 2: with one
// row per thread every wave repeats the scalar work and ten images do not fit 128 registers; with four, one wave per
// SIMD hides nothing).  Its entries of the stored images Ap_0..Ap_{R-1} and of r are REGISTERS (4 VGPRs each and row);
// x and P0 are touched once per restart cycle and stay in memory (own rows).  The only vector that crosses threads is
// the residual a step hands to the operator apply (the residual ring D_1..D_{R-1} of gcr.hip's lean cycle, needed in
// memory anyway: the step that closes a cycle reads its own rows back); 7-point stencils take the near neighbours of it
// from an LDS window written from registers.
// Reductions: every workgroup stores its partial sums as 16-byte {value, generation} slots.  They are folded in two
// hops — (64 workgroups x scalar) tasks dealt over waves of the first workgroups, whose group sums every workgroup then
// polls from one of 16 replicas — because 256 workgroups polling the same 4 KB made one memory channel serve it all
// (5-6 us per exchange).  Order and tree are those of reduce.h's fold_partials: the scalars, hence every iterate, have
// the bits of gcr.hip's kernels (tests/test_gpu_resident.py compares the two paths bit for bit).  Per iteration: two
// such exchanges (the beta numerators with |r|^2; <r,Ap>, <Ap,Ap>) and one hand-over of the residual that only waits
// for the neighbouring workgroups' slots.
//
// Scope: single GPU (no live communicator: several processes could share the device), a Sparse / DiracOp stored one
// thread per row with rows of at most 16 entries (stencil view, row-pattern dictionaries, ELL slab), lean restart
// cycles of 5 or 10 (or solves that end before their first cycle closes), no preconditioner hooks, x0 = 0 or
// ignored, no piecewise hand-over of x or r to a V-cycle.  Everything else takes gcr.hip's path.  A workgroup that
// polls more than RES_SPIN_LIMIT times for another one (the launch was not co-resident: foreign work on the device)
// raises the abort flag, every workgroup leaves, x is poisoned with NaN and the next host synchronisation reports
// the failure — no wave spins forever (test_resident_gives_up_instead_of_hanging).
#include <algorithm>
#include <climits>
#include <cmath>
#include <cstdlib>

#include "internal.h"
#include "reduce.h"
#include "gcr_dev.h"
#include "spmv_dev.h"
#include "exchange_dev.h"

namespace mgcr {


struct ResidentArgs {
    RowMat m;
    const cplx *rhs;
    cplx *x;
    int64_t n;
    int nlogical;
    RowMap rm;
    DevState *st;
    const int *inherit;
    int inherit_it;
    double tol2;
    int max_it, storage;
    int from_zero, alpha_only_last;
    int nbr;             // a row gathers from at most this many workgroups to either side of its own
    int tile_h;          // MODE 6: halo rows of the LDS window (CsrDev::sten_halo_f)
    int spin_limit;      // polls before a workgroup gives up on the others (RES_SPIN_LIMIT; tests shorten it)
    int test_stall;      // tests: logical workgroup test_stall - 1 leaves right after step 0 without a word (0: nobody)
    double *hist;
    int hist_cap;
    cplx *ring;          // (R + 1) slots of n rows: 0 = the residual a closing step starts the next cycle from, m = D_m, R = P0 after the first cycle
    v4i *slots;          // RES_SLOT_BYTES: {value, generation} slots of both hops
    unsigned gen0;
    unsigned *abort_dev; // raised by a workgroup that gave up waiting; polled by the others
    int *abort_host;     // host-mapped copy for resident_check()
    unsigned long long *dbg;   // MGCR_RES_TIMING: ticks (100 MHz) spent per phase, summed over the steps (workgroup 0)
};

// development aid (build with EXTRA=-DMGCR_RES_TIMING, run with MGCR_RES_TIMING=1): ticks of the 100 MHz clock per phase of
// a step, summed over the solve by workgroup 0 and printed by gcr_resident_run.  Off by default: the 8 accumulators are
// 18 registers the ten-image kernel does not have.
#ifdef MGCR_RES_TIMING
#define RES_TICK(a, S, phase)                                \
    do {                                                     \
        if ((a).dbg) {                                       \
            const unsigned long long t_ = wall_clock64();    \
            (S).tacc[(phase) + 8 * (K >= 5 ? 1 : 0)] += t_ - (S).tprev; \
            (S).tprev = t_;                                  \
        }                                                    \
    } while (0)
#else
#define RES_TICK(a, S, phase) do { } while (0)
#endif

// spmv_dev.h sten_row_product_t<NS, false, -1> for the RPT rows of a thread at once: the gathers of ALL rows are in flight before
// the first one is looked at (row by row they would cost RPT memory round trips); per row the same loads, the same terms in
// the same order, the same bits.
template <int NS, int RPT, class XF>
__device__ __forceinline__ void res_sten_rows(const RowMat &m, const int (&row)[RPT], const bool (&act)[RPT], XF xf, cplx (&sum)[RPT]) {
    uint64_t pl[RPT][NS];
    cplx xv[RPT][NS];
#pragma unroll
    for (int h = 0; h < RPT; h++) {
        // (an inactive row set — past the last row — reads the zero presence row behind the last wave and row 0's neighbours: unused)
        const int r = act[h] ? row[h] : 0;
        const int32_t wave = __builtin_amdgcn_readfirstlane(act[h] ? (int32_t)(row[h] >> 6) : m.sten_nwaves);
        const sten_planes_ptr pp = sten_wave_planes(m, wave);
#pragma unroll
        for (int c = 0; c < NS; c++) pl[h][c] = pp[c];
#pragma unroll
        for (int c = 0; c < NS; c++) {
            int32_t j = (int32_t)r + m.sten_off[c];
            j = j < 0 ? 0 : j > m.sten_last ? m.sten_last : j;
            xv[h][c] = xf(j);
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    const int lane = (int)(threadIdx.x & 63);
#pragma unroll
    for (int h = 0; h < RPT; h++) {
        cplx s = make_double2(0., 0.);
#pragma unroll
        for (int c = 0; c < NS; c++) {
            const bool on = (pl[h][c] >> lane & 1ull) != 0ull;
            const cplx ns = cadd(s, sten_term<-1>(m, c, xv[h][c]));
            s.x = on ? ns.x : s.x;
            s.y = on ? ns.y : s.y;
        }
        sum[h] = s;
    }
}

// res_sten_rows for a 3-D 7-point stencil (slots 1..5 = the row itself, +- 1, +- n: within H <= 512 rows; slots 0 and 6 = +- n^2):
// the near slots come from an LDS window holding the workgroup's 1024 rows (written from registers) and H halo rows to
// either side (fetched by a few threads), only the two far slots are gathered from memory: 2.1 instead of 7 coherent
// loads per row.  Same terms, same order, same bits.
constexpr int RES_TILE_HALO = RED_THREADS / 2;
template <int RPT, class XF>
__device__ __forceinline__ void res_sten_rows_tile(const RowMat &m, const int (&row)[RPT], const bool (&act)[RPT], const cplx *win, int base, int H,
                                                   XF xf, cplx (&sum)[RPT]) {
    constexpr int NS = 7;
    uint64_t pl[RPT][NS];
    cplx far0[RPT], far6[RPT];
#pragma unroll
    for (int h = 0; h < RPT; h++) {
        const int r = act[h] ? row[h] : 0;
        const int32_t wave = __builtin_amdgcn_readfirstlane(act[h] ? (int32_t)(row[h] >> 6) : m.sten_nwaves);
        const sten_planes_ptr pp = sten_wave_planes(m, wave);
#pragma unroll
        for (int c = 0; c < NS; c++) pl[h][c] = pp[c];
        int32_t j0 = (int32_t)r + m.sten_off[0], j6 = (int32_t)r + m.sten_off[6];
        j0 = j0 < 0 ? 0 : j0 > m.sten_last ? m.sten_last : j0;
        j6 = j6 < 0 ? 0 : j6 > m.sten_last ? m.sten_last : j6;
        far0[h] = xf(j0);
        far6[h] = xf(j6);
    }
    __syncthreads();   // the halo rows (written by the caller just before) are in the window
    const int lane = (int)(threadIdx.x & 63);
#pragma unroll
    for (int h = 0; h < RPT; h++) {
        const int r = act[h] ? row[h] : base;
        cplx xv[NS];
        xv[0] = far0[h];
        xv[6] = far6[h];
#pragma unroll
        for (int c = 1; c < 6; c++) xv[c] = win[r - base + H + m.sten_off[c]];
        cplx s = make_double2(0., 0.);
#pragma unroll
        for (int c = 0; c < NS; c++) {
            const bool on = (pl[h][c] >> lane & 1ull) != 0ull;
            const cplx ns = cadd(s, sten_term<-1>(m, c, xv[c]));
            s.x = on ? ns.x : s.x;
            s.y = on ? ns.y : s.y;
        }
        sum[h] = s;
    }
}

template <int R, int RPT>
struct ResState {
    cplx Ap[RPT][R];     // images of the cycle's directions, this thread's rows
    cplx rv[RPT];
    int row[RPT];        // the rows (1024 / RPT apart inside the workgroup's 1024)
    int32_t t0[RPT];     // MODE 1, 2: the row's pattern (id * W), fixed for the solve
    PatLds pl;           // MODE 1: the pattern table in LDS
    cplx *win;           // MODE 6 (7-point stencil through an LDS window): the window, tile_h halo rows to either side
    bool act[RPT];
    cplx num, den;       // <r,Ap_cur>, <Ap_cur,Ap_cur>
    double bnorm2, rr;
    int it, npend, stop_at, iter;
    bool x_live;         // x holds something to add to (else: x0 = 0 that was never written)
    bool p0_rhs;         // first cycle: P0 is the right-hand side itself
    bool aborted;
#ifdef MGCR_RES_TIMING
    unsigned long long tacc[16], tprev;
#endif
};

template <int R>
struct ResTables {       // LDS: gcr_dev.h LeanCoef with rows of R, plus the per-slot denominators and the step's coefficients
    cplx T[R * R], t[R], cx[R], den[R], beta[R], cp[R];
};

// One step at cycle position K (direction K is the current one, lim = K + 1 directions are stored).  Returns false when the
// solve is over (converged, last iteration, abort).
template <int MODE, int NS, int R, int RPT, int K>
__device__ __forceinline__ bool res_step(const ResidentArgs &a, ResState<R, RPT> &S, ResTables<R> &tb, ResSync &sy, __amdgpu_buffer_rsrc_t ring, bool owner0) {
    constexpr int lim = K + 1;
    constexpr bool closing = K + 1 == R;
    constexpr int nxt = closing ? 0 : K + 1;
    const int nwave = (int)blockDim.x >> 6, wave = (int)threadIdx.x >> 6;
    S.it++;
    const int it = S.it;
    RES_TICK(a, S, 7);   // (whatever ran since the last tick: loop overhead)
    const bool last = it == a.max_it;
    // alpha and its bookkeeping (gcr.hip xr_update_kernel<true, true> / alpha_only_kernel)
    const cplx alpha = to_sgpr(cdiv(S.num, S.den));
    if (threadIdx.x == 0) tb.den[K] = S.den;
    if ((int)threadIdx.x < R) {   // gcr_dev.h lean_pending_update, slot = K
        const int m = threadIdx.x;
        if (K == 0) tb.cx[m] = m == 0 ? alpha : make_double2(0., 0.);
        else if (m == 0) tb.cx[0] = cadd(tb.cx[0], cmul(alpha, tb.t[K]));
        else if (m <= K) tb.cx[m] = cadd(tb.cx[m], cmul(alpha, tb.T[K * R + m]));
    }
    S.npend = K + 1;
    if (last && a.alpha_only_last) {   // the caller only wants x: no residual, no history entry (alpha_only_kernel)
        S.iter = it;
        return false;
    }
    const int vbytes = (int)a.n * 16;   // one ring slot
    cplx rn[RPT];
#pragma unroll
    for (int h = 0; h < RPT; h++) {
        rn[h] = csub(S.rv[h], cmul(alpha, S.Ap[h][K]));
        S.rv[h] = rn[h];
        if (S.act[h] && !last) st_coh(ring, S.row[h], nxt * vbytes, rn[h]);
        if (MODE == 6 && !last) S.win[h * (int)blockDim.x + (int)threadIdx.x + a.tile_h] = rn[h];   // (its readers are behind the hand-over's barrier)
    }
    {   // |r|^2; the residual has reached memory when the workgroup's slot says so (every wave waits for its stores first)
        __builtin_amdgcn_s_waitcnt(0);
#pragma unroll
        for (int h = 0; h < RPT; h++) {
            double v[1] = {0.};
            if (S.act[h]) v[0] += rn[h].x * rn[h].x + rn[h].y * rn[h].y;
            res_contrib<1>(sy, 0, v, h * nwave + wave);
        }
        res_publish<1>(sy, 0);
        RES_TICK(a, S, 0);
        if (last) {   // nothing after this step: fold |r|^2 now (gcr.hip finish_step_kernel)
            if (!res_collect<1>(sy, 0)) { S.aborted = true; return false; }
            const double rr = to_sgpr(res_total(sy, 0));
            S.iter = it;
            S.rr = rr;
            if (owner0 && it < a.hist_cap) a.hist[it] = sqrt(rr) / sqrt(S.bnorm2);
            if (!((rr / S.bnorm2) > a.tol2)) S.stop_at = it;
            return false;
        }
        if (!res_neighbour_wait(sy, a.nbr)) { S.aborted = true; return false; }
        RES_TICK(a, S, 1);
    }
    // Ar = A r (or r - k A r) from the neighbours' rows, <Ar, Ap_j> for the stored directions
    cplx ar[RPT];
    {
        cplx sum[RPT];
        if constexpr (MODE == 3) {
            res_sten_rows<NS, RPT>(a.m, S.row, S.act, [&](int32_t j) -> cplx { return ld_coh(ring, j, nxt * vbytes); }, sum);
        } else if constexpr (MODE == 6) {
            const int base = sy.lb * RED_THREADS, H = a.tile_h;
            for (int t = threadIdx.x; t < 2 * H; t += blockDim.x) {   // the halo rows, from the neighbouring workgroups
                const int g = t < H ? base - H + t : base + RED_THREADS + (t - H);
                if (g >= 0 && g < a.n) S.win[t < H ? t : RED_THREADS + t] = ld_coh(ring, g, nxt * vbytes);
            }
            res_sten_rows_tile<RPT>(a.m, S.row, S.act, S.win, base, H, [&](int32_t j) -> cplx { return ld_coh(ring, j, nxt * vbytes); }, sum);
        } else {   // ELL slab / row-pattern dictionaries (spmv_dev.h row_product): NS is the compile-time row width (0: m.W)
#pragma unroll
            for (int h = 0; h < RPT; h++) {
                sum[h] = make_double2(0., 0.);
                if (S.act[h]) sum[h] = fused_row_product<MODE, NS>(a.m, S.row[h], S.t0[h], S.pl, [&](int32_t j) -> cplx { return ld_coh(ring, j, nxt * vbytes); });
            }
        }
#pragma unroll
        for (int h = 0; h < RPT; h++) {
            ar[h] = make_double2(0., 0.);
            if (S.act[h]) ar[h] = a.m.shift ? csub(rn[h], cmul(a.m.k, sum[h])) : sum[h];
        }
    }
    RES_TICK(a, S, 2);
    {
        constexpr int NVB = 2 * lim;
#pragma unroll
        for (int h = 0; h < RPT; h++) {
#pragma unroll
            for (int c0 = 0; c0 + RES_CHUNK <= NVB; c0 += RES_CHUNK) {
                double v[RES_CHUNK];
#pragma unroll
                for (int q = 0; q < RES_CHUNK / 2; q++) {
                    const cplx t = S.act[h] ? cconj_mul(ar[h], S.Ap[h][c0 / 2 + q]) : make_double2(0., 0.);
                    v[2 * q] = 0. + t.x;
                    v[2 * q + 1] = 0. + t.y;
                }
                res_contrib<RES_CHUNK>(sy, c0, v, h * nwave + wave);
            }
            constexpr int tail = NVB % RES_CHUNK;
            if constexpr (tail > 0) {
                constexpr int c0 = NVB - tail;
                double v[tail];
#pragma unroll
                for (int q = 0; q < tail / 2; q++) {
                    const cplx t = S.act[h] ? cconj_mul(ar[h], S.Ap[h][c0 / 2 + q]) : make_double2(0., 0.);
                    v[2 * q] = 0. + t.x;
                    v[2 * q + 1] = 0. + t.y;
                }
                res_contrib<tail>(sy, c0, v, h * nwave + wave);
            }
        }
        res_publish<NVB>(sy, 1);
        RES_TICK(a, S, 3);
        if (!res_collect<NVB, true>(sy, 1)) { S.aborted = true; return false; }
        RES_TICK(a, S, 4);
        // the step's bookkeeping (gcr.hip close_step, inside the build kernels: after the apply, like here)
        const double rr = to_sgpr(res_total(sy, NVB));
        S.iter = it;
        S.rr = rr;
        if (owner0 && it < a.hist_cap) a.hist[it] = sqrt(rr) / sqrt(S.bnorm2);
        if (!((rr / S.bnorm2) > a.tol2)) { S.stop_at = it; return false; }
    }
    // beta_j, the coefficient table, the new image (gcr.hip build_lean_kernel / close_x_kernel / build_close_kernel)
    if ((int)threadIdx.x < lim) tb.beta[threadIdx.x] = cdiv(make_double2(res_total(sy, 2 * threadIdx.x), res_total(sy, 2 * threadIdx.x + 1)), tb.den[threadIdx.x]);
    __syncthreads();
    if constexpr (closing) {
        if ((int)threadIdx.x < lim) {
            const int m = threadIdx.x;
            cplx c = make_double2(0., 0.);
            if (m == 0) {
                for (int j = 0; j < lim; j++) c = cadd(c, cmul(tb.beta[j], j == 0 ? make_double2(1., 0.) : tb.t[j]));
            } else {
                for (int j = m; j < lim; j++) c = cadd(c, cmul(tb.beta[j], j == m ? make_double2(1., 0.) : tb.T[j * R + m]));
            }
            tb.cp[m] = c;
        }
        __syncthreads();
    }
    cplx beta[lim];
#pragma unroll
    for (int j = 0; j < lim; j++) beta[j] = to_sgpr(tb.beta[j]);
    cplx an[RPT];
#pragma unroll
    for (int h = 0; h < RPT; h++) {
        cplx ac = make_double2(0., 0.);
#pragma unroll
        for (int j = 0; j < lim; j++) ac = csub(ac, cmul(beta[j], S.Ap[h][j]));
        an[h] = cadd(ar[h], ac);
    }
    if constexpr (closing) {
        // x += cx_0 P0 + sum_m cx_m D_m;  P0' = D_R - cp_0 P0 - sum_m cp_m D_m  (D_R = this step's residual); own rows only,
        // five vectors at a time
#pragma unroll
        for (int h = 0; h < RPT; h++) {
            const int i = S.row[h];
            const bool act = S.act[h];
            cplx xv = (act && S.x_live) ? a.x[i] : make_double2(0., 0.);
            cplx pc = make_double2(0., 0.);
#pragma unroll
            for (int j0 = 0; j0 < R; j0 += 5) {
                cplx pj[5];
#pragma unroll
                for (int q = 0; q < 5; q++) {
                    const int j = j0 + q;
                    pj[q] = make_double2(0., 0.);
                    if (act && j < R) pj[q] = j == 0 ? (S.p0_rhs ? a.rhs[i] : ld_coh(ring, i, R * vbytes)) : ld_coh(ring, i, j * vbytes);
                }
#pragma unroll
                for (int q = 0; q < 5; q++)
                    if (j0 + q < R) xv = cadd(xv, cmul(to_sgpr(tb.cx[j0 + q]), pj[q]));
#pragma unroll
                for (int q = 0; q < 5; q++)
                    if (j0 + q < R) pc = csub(pc, cmul(to_sgpr(tb.cp[j0 + q]), pj[q]));
                __builtin_amdgcn_sched_barrier(0);
            }
            if (act) {
                a.x[i] = xv;
                st_coh(ring, i, R * vbytes, cadd(rn[h], pc));
            }
        }
        S.x_live = true;
        S.p0_rhs = false;
        S.npend = 0;
    }
#pragma unroll
    for (int h = 0; h < RPT; h++) S.Ap[h][nxt] = an[h];
    {
#pragma unroll
        for (int h = 0; h < RPT; h++) {
            double v[4] = {0., 0., 0., 0.};
            if (S.act[h]) {
                const cplx t = cconj_mul(rn[h], an[h]);
                v[0] += t.x; v[1] += t.y;
                const cplx u = cconj_mul(an[h], an[h]);
                v[2] += u.x; v[3] += u.y;
            }
            res_contrib<4>(sy, 0, v, h * nwave + wave);
        }
        res_publish<4>(sy, 2);
        RES_TICK(a, S, 5);
        // the coefficient table's row k = lim, by the LAST wave while the exchange is under way (the first waves carry the
        // exchange's tasks); its readers (the next step's pending-x update, the closing step) are behind later barriers
        if constexpr (!closing) {
            const int m = (int)threadIdx.x - ((int)blockDim.x - 64);
            if (m >= 0 && m <= lim) {
                constexpr int k = lim;
                cplx c = make_double2(0., 0.);
                if (m == 0) {
                    for (int j = 0; j < k; j++) c = csub(c, cmul(tb.beta[j], j == 0 ? make_double2(1., 0.) : tb.t[j]));
                    tb.t[k] = c;
                } else if (m < k) {
                    for (int j = m; j < k; j++) c = csub(c, cmul(tb.beta[j], j == m ? make_double2(1., 0.) : tb.T[j * R + m]));
                    tb.T[k * R + m] = c;
                } else {
                    tb.T[k * R + k] = make_double2(1., 0.);
                }
            }
        }
        if (!res_collect<4>(sy, 2)) { S.aborted = true; return false; }
        RES_TICK(a, S, 6);
        S.num = to_sgpr(make_double2(res_total(sy, 0), res_total(sy, 1)));
        S.den = to_sgpr(make_double2(res_total(sy, 2), res_total(sy, 3)));
    }
    return true;
}

template <int MODE, int NS, int R, int RPT, int K>
__device__ __forceinline__ bool res_cycle(const ResidentArgs &a, ResState<R, RPT> &S, ResTables<R> &tb, ResSync &sy, __amdgpu_buffer_rsrc_t ring, bool owner0) {
    if constexpr (K < R) {
        if (K >= a.storage) return false;   // (never reached: a solve whose cycle cannot close ends by max_it before)
        if (!res_step<MODE, NS, R, RPT, K>(a, S, tb, sy, ring, owner0)) return false;
        return res_cycle<MODE, NS, R, RPT, K + 1>(a, S, tb, sy, ring, owner0);
    } else {
        return true;
    }
}

// RPT rows per thread: 1024 / RPT threads per workgroup.  The reductions are those of 1024 threads (16 sums of 64 rows per
// workgroup and scalar); what shrinks with the thread count is everything a wave does whatever its rows (alpha, beta, the
// exchanges' polling, addressing): more than half of a step's instructions with one row per thread.
template <int MODE, int NS, int R, int RPT>
__global__ void __launch_bounds__(RED_THREADS / RPT, RED_THREADS / RPT / 256) gcr_resident_kernel(ResidentArgs a) {
    __shared__ double lds_pw[RES_NV * 17], lds_ws[RES_NV * RES_GRP];
    __shared__ int gave_up;
    __shared__ ResTables<R> tb;
    extern __shared__ __attribute__((aligned(16))) unsigned char step_smem[];
    constexpr int T = RED_THREADS / RPT;
    const int lb = logical_workgroup(a.rm, (int)blockIdx.x, (int)gridDim.x);
    const bool owner0 = lb == 0 && threadIdx.x == 0;
    // an outer solve that is over silences this one (gcr.hip reset_kernel)
    if (a.inherit && a.inherit[0] < a.inherit[1] + a.inherit_it) {
        if (owner0) {
            a.st->stop_at = -1; a.st->base = 0; a.st->iter = 0; a.st->npend = 0; a.st->bnorm2 = 0.; a.st->rr = 0.; a.st->tol2 = a.tol2;
        }
        return;
    }
    if (lb >= a.nlogical) return;
    const int vbytes = (int)a.n * 16;
    const __amdgpu_buffer_rsrc_t ring = res_rsrc(a.ring, (unsigned)(R + 1) * (unsigned)vbytes);
    const int nwave = T / 64, wave = (int)threadIdx.x >> 6;
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
    __syncthreads();
    ResState<R, RPT> S;
#pragma unroll
    for (int h = 0; h < RPT; h++) {
        S.row[h] = lb * RED_THREADS + h * T + (int)threadIdx.x;
        S.act[h] = S.row[h] < a.n;
#pragma unroll
        for (int j = 0; j < R; j++) S.Ap[h][j] = make_double2(0., 0.);
        S.rv[h] = S.act[h] ? a.rhs[S.row[h]] : make_double2(0., 0.);
        S.t0[h] = 0;
        if ((MODE == 1 || MODE == 2) && S.act[h]) S.t0[h] = (int32_t)a.m.pid[S.row[h]] * (NS ? NS : a.m.W);
    }
    S.pl = PatLds{nullptr, nullptr, nullptr};
    if (MODE == 1) S.pl = stage_patterns(a.m, step_smem);
    S.win = reinterpret_cast<cplx *>(step_smem);   // MODE 6: (1024 + 2 tile_h) entries
    S.x_live = !a.from_zero;
    S.p0_rhs = true;
    S.it = 0; S.npend = 0; S.stop_at = INT_MAX; S.iter = 0; S.rr = 0.; S.bnorm2 = 0.;
    S.aborted = false;
#ifdef MGCR_RES_TIMING
    for (int q = 0; q < 16; q++) S.tacc[q] = 0;
    S.tprev = a.dbg ? wall_clock64() : 0ull;
#endif
    // step 0: Ap_0 = A r_0 and <r_0,Ap_0>, <Ap_0,Ap_0>, |r_0|^2 = |b|^2 (gcr_fused.hip init_apply_kernel, gcr.hip init_kernel)
    {
#pragma unroll
        for (int h = 0; h < RPT; h++) {
            double v[5] = {0., 0., 0., 0., 0.};
            if (S.act[h]) {
                const cplx sum = fused_row_product<(MODE == 6 ? 3 : MODE), NS>(a.m, S.row[h], S.t0[h], S.pl, [&](int32_t j) -> cplx { return a.rhs[j]; });
                const cplx rv = S.rv[h];
                const cplx yi = a.m.shift ? csub(rv, cmul(a.m.k, sum)) : sum;
                S.Ap[h][0] = yi;
                v[4] += rv.x * rv.x + rv.y * rv.y;
                const cplx t = cconj_mul(rv, yi);
                v[0] += t.x; v[1] += t.y;
                const cplx u = cconj_mul(yi, yi);
                v[2] += u.x; v[3] += u.y;
            }
            res_contrib<5>(sy, 0, v, h * nwave + wave);
        }
        res_publish<5>(sy, 2);
        if (!res_collect<5>(sy, 2)) S.aborted = true;
        S.num = to_sgpr(make_double2(res_total(sy, 0), res_total(sy, 1)));
        S.den = to_sgpr(make_double2(res_total(sy, 2), res_total(sy, 3)));
        S.rr = to_sgpr(res_total(sy, 4));
        S.bnorm2 = S.rr;
        if (owner0 && !S.aborted) a.hist[0] = sqrt(S.rr) / sqrt(S.bnorm2);
    }
    if (a.test_stall && lb == a.test_stall - 1) return;   // (tests) the others must notice, give up and say so
    if (!S.aborted)
        while (res_cycle<MODE, NS, R, RPT, 0>(a, S, tb, sy, ring, owner0)) {}
    __syncthreads();
    if (S.aborted) {
        if (threadIdx.x == 0) {
            __hip_atomic_store(a.abort_dev, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(a.abort_host, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
#pragma unroll
        for (int h = 0; h < RPT; h++)
            if (S.act[h]) a.x[S.row[h]] = make_double2(__builtin_nan(""), __builtin_nan(""));
        if (owner0) { a.st->stop_at = S.it; a.st->base = 0; a.st->iter = S.it; a.st->npend = 0; a.st->bnorm2 = S.bnorm2; a.st->rr = __builtin_nan(""); a.st->tol2 = a.tol2; }
        return;
    }
    // the x updates still pending (gcr.hip flush_x_kernel): x += cx_0 P0 + sum_{1 <= m < npend} cx_m D_m
#pragma unroll
    for (int h = 0; h < RPT; h++) {
        const int i = S.row[h];
        const bool act = S.act[h];
        cplx xv = (act && S.x_live) ? a.x[i] : make_double2(0., 0.);
        const int np = S.npend;
#pragma unroll
        for (int j0 = 0; j0 < R; j0 += 5) {
            cplx pj[5];
#pragma unroll
            for (int q = 0; q < 5; q++) {
                const int j = j0 + q;
                pj[q] = make_double2(0., 0.);
                if (act && j < R && j < np) pj[q] = j == 0 ? (S.p0_rhs ? a.rhs[i] : ld_coh(ring, i, R * vbytes)) : ld_coh(ring, i, j * vbytes);
            }
#pragma unroll
            for (int q = 0; q < 5; q++)
                if (j0 + q < R && j0 + q < np) xv = cadd(xv, cmul(to_sgpr(tb.cx[j0 + q]), pj[q]));
            __builtin_amdgcn_sched_barrier(0);
        }
        if (act) a.x[i] = xv;
    }
#ifdef MGCR_RES_TIMING
    if (owner0 && a.dbg)
        for (int q = 0; q < 16; q++) a.dbg[q] = S.tacc[q];
#endif
    if (owner0) {
        a.st->stop_at = S.stop_at; a.st->base = 0; a.st->iter = S.iter; a.st->npend = 0; a.st->bnorm2 = S.bnorm2; a.st->rr = S.rr; a.st->tol2 = a.tol2;
    }
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
static int64_t g_resident_solves = 0;
int64_t resident_solve_count() { return g_resident_solves; }
static int g_resident = -1;
static bool resident_enabled() {
    if (g_resident < 0) g_resident = !(getenv("MGCR_RESIDENT") && atoi(getenv("MGCR_RESIDENT")) == 0);
    return g_resident != 0;
}
bool set_resident_enabled(bool on) {
    bool prev = resident_enabled();
    g_resident = on ? 1 : 0;
    return prev;
}

ExchangeShared &exchange_shared() {
    static ExchangeShared s;
    return s;
}
int exchange_shared_init() {
    ExchangeShared &s = exchange_shared();
    if (s.slots) return MGCR_OK;
    hipDeviceProp_t prop;
    int dev = 0;
    MGCR_HIP(hipGetDevice(&dev));
    MGCR_HIP(hipGetDeviceProperties(&prop, dev));
    s.cus = prop.multiProcessorCount;
    const size_t bytes = RES_SLOT_BYTES;
    MGCR_HIP(hipMalloc((void **)&s.slots, bytes));
    MGCR_HIP(hipMemset(s.slots, 0, bytes));
    MGCR_HIP(hipMalloc((void **)&s.abort_dev, sizeof(unsigned)));
    MGCR_HIP(hipMemset(s.abort_dev, 0, sizeof(unsigned)));
    MGCR_HIP(hipHostMalloc((void **)&s.abort_host, sizeof(int), hipHostMallocMapped));
    *s.abort_host = 0;
    return MGCR_OK;
}
// generations never repeat: start over on cleared slots before the counter wraps
unsigned exchange_take_generations(unsigned need) {
    ExchangeShared &sh = exchange_shared();
    if (sh.gen > 0xffffffffu - need - 1u) {
        hipMemsetAsync(sh.slots, 0, RES_SLOT_BYTES, ctx().stream);
        sh.gen = 1;
    }
    const unsigned g0 = sh.gen;
    sh.gen += need;
    return g0;
}
void resident_shutdown() {
    ExchangeShared &s = exchange_shared();
    if (s.slots) hipFree(s.slots);
    if (s.abort_dev) hipFree(s.abort_dev);
    if (s.abort_host) hipHostFree(s.abort_host);
    s = ExchangeShared();
}
// did a resident solve give up since the last look?  (called where results are handed back to the host; gcr_run, which can
// repeat the solve, asks with internal = true and gets MGCR_INT_GAVE_UP instead of an error for the caller)
int resident_check(bool internal) {
    ExchangeShared &s = exchange_shared();
    if (s.abort_host && *(volatile int *)s.abort_host != 0) {
        *(volatile int *)s.abort_host = 0;
        hipMemsetAsync(s.abort_dev, 0, sizeof(unsigned), ctx().stream);
        // whatever kept the launch from being co-resident may still be there: this process stays with the multi-kernel paths
        set_resident_enabled(false);
        set_stepbuild_enabled(false);
        if (internal) return MGCR_INT_GAVE_UP;
        set_error("one-launch solver kernel: a workgroup waited too long for the others (the launch was not co-resident); its results were "
                  "poisoned with NaN.  The one-launch paths are now off in this process: repeat the operation (GCR::solve repeats itself; "
                  "this came from an operator apply or a V-cycle whose output the library does not own)");
        return MGCR_ERR_HIP;
    }
    return MGCR_OK;
}
bool one_launch_paths_enabled() { return resident_enabled() || stepbuild_is_enabled(); }

// which instantiation a solve launches, with how much dynamic LDS, on how many workgroups (shared by the eligibility test,
// which asks the runtime whether that launch is co-resident, and the launch itself)
struct ResidentPlan {
    const void *kernel;
    size_t lds_bytes;
    unsigned grid;
    int threads, tile_h, g;
};
static ResidentPlan resident_plan(const CsrDev &M, int storage, int restart, int max_it) {
    ResidentPlan pl{};
    const int64_t n = M.nrow;
    pl.g = red_grid(n);
    pl.grid = (unsigned)(pl.g >= 64 ? (pl.g + 7) / 8 * 8 : pl.g);
    static const bool tile_on = !(getenv("MGCR_RESIDENT_TILE") && atoi(getenv("MGCR_RESIDENT_TILE")) == 0);
    const bool tile = tile_on && csr_stencil_active(M) && !M.sten_rare && sten_slots(M) == 7 && M.sten_near_f == 0x3eu && M.sten_halo_f > 0 &&
                      M.sten_halo_f <= RES_TILE_HALO;
    pl.tile_h = tile ? M.sten_halo_f : 0;
    pl.lds_bytes = tile ? sizeof(cplx) * (size_t)(RED_THREADS + 2 * pl.tile_h) : csr_stencil_active(M) ? 0 : row_mat_lds_bytes(M);
    const int R = storage <= 5 && (restart == 5 || max_it < restart) ? 5 : 10;
    static const int rpt_env = getenv("MGCR_RESIDENT_RPT") ? atoi(getenv("MGCR_RESIDENT_RPT")) : 0;
    int rpt = rpt_env == 1 || rpt_env == 2 || rpt_env == 4 ? rpt_env : RES_RPT_DEFAULT;
#define RES_K(MODE, NS, RR, RPT) ((const void *)gcr_resident_kernel<MODE, NS, RR, RPT>)
#ifdef MGCR_RES_ALL_RPT   /* experiments: one and four rows per thread as well (build with EXTRA=-DMGCR_RES_ALL_RPT) */
#define RES_K_R(MODE, NS, RR) (rpt == 1 ? RES_K(MODE, NS, RR, 1) : rpt == 2 ? RES_K(MODE, NS, RR, 2) : RES_K(MODE, NS, RR, 4))
#else
    rpt = 2;
#define RES_K_R(MODE, NS, RR) RES_K(MODE, NS, RR, 2)
#endif
#define RES_K_M(MODE, NS) (R == 5 ? RES_K_R(MODE, NS, 5) : RES_K_R(MODE, NS, 10))
    if (tile) pl.kernel = RES_K_M(6, 7);
    else if (csr_stencil_active(M)) pl.kernel = sten_slots(M) == 7 ? RES_K_M(3, 7) : RES_K_M(3, 9);
    else if (M.pat_mode == 1) pl.kernel = RES_K_M(1, 0);
    else if (M.pat_mode == 2) pl.kernel = RES_K_M(2, 0);
    else pl.kernel = RES_K_M(0, 0);
#undef RES_K_M
#undef RES_K_R
#undef RES_K
    pl.threads = RED_THREADS / rpt;
    return pl;
}

bool gcr_resident_eligible(const Op *A, const mgcr_gcr_param &p, int storage, int restart, int64_t n, bool lean, bool nested_handoff) {
    if (!resident_enabled() || !lean || nested_handoff) return false;
    if (p.use_x0 || p.flexible || p.left_precond || p.right_precond || p.profile_spmv) return false;
    if (comm_live_count() > 0) return false;   // several processes may share this GPU: a launch that needs the whole chip could wait on another one
    const Op *b0 = A->kind == OP_DIRAC ? A->base : A;
    if (!b0 || b0->kind != OP_CSR || b0->dist || A->comm) return false;
    const CsrDev &M = b0->csr;
    // one thread walks a whole row, gather after gather: short rows only.  (Long rows with few of them — the reference's 3072 x 39
    // sample, stored with 8 lanes per row — were tried through a one-thread replay of spmv.hip's lane sums: 32 us per iteration
    // against 20 for the multi-kernel path, whose SpMV spreads a row over 8 lanes.)
    if (!csr_fusable(M, nullptr) || M.nrow != n || M.W > 16) return false;
    if (csr_stencil_active(M) && (M.sten_rare || (sten_slots(M) != 7 && sten_slots(M) != 9))) return false;
    const int max_it = p.max_iter > 0 ? p.max_iter : 1;
    const bool never_closes = max_it < restart;
    if (storage > 10) return false;
    if (!never_closes && !(restart == storage && (restart == 5 || restart == 10))) return false;
    // every step consumes three exchange generations: the counter is 32 bits wide (exchange_take_generations)
    if (max_it > (1 << 28)) return false;
    if (exchange_shared_init() != MGCR_OK) return false;
    const ResidentPlan pl = resident_plan(M, storage, restart, max_it);
    int cus = exchange_shared().cus;
    if (cus > 256) cus = 256;   // one workgroup per CU by design (two rows per thread, the Krylov vectors in registers)
    if ((int64_t)pl.g * RED_THREADS < n || (int)pl.grid > cus) return false;
    // the workgroups wait for each other: the whole grid has to be on the chip at once — asked of the runtime for the
    // instantiation that would run (gcr_stepbuild.hip launch_is_coresident)
    return launch_is_coresident(pl.kernel, pl.threads, pl.lds_bytes, (int)pl.grid);
}

int gcr_resident_run(Op *A, const mgcr_gcr_param &p, int storage, int restart, const cplx *rhs, cplx *x, bool from_zero, bool alpha_only_last,
                     DevState *st, double *hist, int hist_cap, cplx *ring, SkipRef outer) {
    MGCR_TRY(exchange_shared_init());
    ExchangeShared &sh = exchange_shared();
    const Op *b0 = A->kind == OP_DIRAC ? A->base : A;
    const CsrDev &M = b0->csr;
    const int64_t n = M.nrow;
    ResidentArgs a;
    a.m = row_mat(M, A->kind == OP_DIRAC, A->k);
    a.rhs = rhs; a.x = x; a.n = n;
    const int g = red_grid(n);
    a.nlogical = g;
    a.rm = make_row_map(n, g, M.reach);
    a.st = st; a.inherit = outer.p; a.inherit_it = outer.it; a.tol2 = p.tol * p.tol;
    a.max_it = p.max_iter > 0 ? p.max_iter : 1;
    a.storage = storage;
    a.from_zero = from_zero ? 1 : 0;
    a.alpha_only_last = alpha_only_last ? 1 : 0;
    const ResidentPlan pl = resident_plan(M, storage, restart, a.max_it);
    a.tile_h = pl.tile_h;
    a.spin_limit = getenv("MGCR_TEST_RESIDENT_SPIN_LIMIT") ? atoi(getenv("MGCR_TEST_RESIDENT_SPIN_LIMIT")) : RES_SPIN_LIMIT;
    a.test_stall = getenv("MGCR_TEST_RESIDENT_STALL") ? atoi(getenv("MGCR_TEST_RESIDENT_STALL")) : 0;
    {   // how far a row's gathers go: exactly for the stencil view, CsrDev::reach otherwise (0 = unknown: every workgroup)
        int64_t reach = 0;
        if (csr_stencil_active(M)) for (int c = 0; c < M.sten_ns; c++) reach = std::max<int64_t>(reach, std::llabs((long long)M.sten_off[c]));
        else reach = M.reach > 0 ? M.reach : n;
        a.nbr = (int)std::min<int64_t>((reach + RED_THREADS - 1) / RED_THREADS, RES_BLK);
    }
    a.hist = hist; a.hist_cap = hist_cap;
    a.ring = ring;
    a.slots = sh.slots; a.abort_dev = sh.abort_dev; a.abort_host = sh.abort_host;
    static const bool timing = getenv("MGCR_RES_TIMING") && atoi(getenv("MGCR_RES_TIMING")) != 0;
    static unsigned long long *dbg = nullptr;
    if (timing && !dbg) { MGCR_HIP(hipMalloc((void **)&dbg, 16 * sizeof(unsigned long long))); }
    a.dbg = timing ? dbg : nullptr;
    a.gen0 = exchange_take_generations(3u * (unsigned)a.max_it + 4u);   // (max_it <= 2^28: gcr_resident_eligible)
    void *kargs[1] = {(void *)&a};
    MGCR_HIP(hipLaunchKernel(pl.kernel, dim3(pl.grid), dim3(pl.threads), kargs, pl.lds_bytes, ctx().stream));
    MGCR_HIP(hipGetLastError());
    g_resident_solves++;
    if (timing) {   // development aid: where the steps of this solve spent their time (workgroup 0)
        unsigned long long h[16];
        MGCR_HIP(hipStreamSynchronize(ctx().stream));
        MGCR_HIP(hipMemcpy(h, dbg, sizeof(h), hipMemcpyDeviceToHost));
        const char *names[8] = {"xr + publish |r|^2", "collect |r|^2", "bookkeeping + gather + apply", "dots + publish", "collect beta numerators",
                                "beta, table, build + publish", "collect <r,Ap>, <Ap,Ap>", "between steps"};
        fprintf(stderr, "resident solve, n = %lld, max_it %d:", (long long)n, a.max_it);
        for (int q = 0; q < 8; q++) fprintf(stderr, " [%s] %.1f + %.1f us", names[q], (double)h[q] * 0.01, (double)h[q + 8] * 0.01);
        fprintf(stderr, " (totals over the steps at cycle positions 0-4 + 5-9)\n");
    }
    return MGCR_OK;
}

// ------------------------------------------------------------------------------------------------
// Self-test of what the one-launch paths rely on (mgcr_selftest_coherence): rows written by one workgroup with
// `buffer_store ... sc1` (+ s_waitcnt) are seen by workgroups of OTHER XCDs through `buffer_load ... sc1` after nothing
// but this file's fence-free exchange in between.  That is hardware behaviour the HIP memory model does not promise, so it
// is checked on the chip and by every GPU test run, with the library's own primitives (st_coh / ld_coh, res_contrib /
// res_publish / res_collect) and the bounded polls of the solvers: `steps` times, every thread adds (1, -1) to the entry
// `shift` rows away in the vector the previous step wrote (a far gather: other workgroups, other XCDs), ping-pong between two
// vectors; the host knows what every entry must be after `steps` steps.  coherent = 0 runs the same steps with ordinary
// loads and stores (which the L2s of the 8 XCDs do not keep coherent inside a launch): the control that shows the test can fail.
// ------------------------------------------------------------------------------------------------
struct CohArgs {
    cplx *a, *b;
    int n, shift, steps, coherent, nblk;
    v4i *slots;
    unsigned gen0;
    unsigned *abort_dev;
    int *abort_host;
    int spin_limit;
};
__global__ void __launch_bounds__(RED_THREADS) coherence_selftest_kernel(CohArgs a) {
    __shared__ double lds_pw[RES_NV * 17], lds_ws[RES_NV * RES_GRP];
    __shared__ int gave_up;
    ResSync sy;
    sy.slots = res_rsrc(a.slots, (unsigned)RES_SLOT_BYTES);
    sy.gen = a.gen0;
    sy.nblk = a.nblk;
    sy.lb = (int)blockIdx.x;
    sy.abort_dev = a.abort_dev;
    sy.spin_limit = a.spin_limit;
    sy.pw = lds_pw;
    sy.ws = lds_ws;
    sy.gave_up = &gave_up;
    if (threadIdx.x == 0) gave_up = 0;
    __syncthreads();
    const int i = (int)blockIdx.x * RED_THREADS + (int)threadIdx.x;
    const __amdgpu_buffer_rsrc_t ra = res_rsrc(a.a, (unsigned)a.n * 16u), rb = res_rsrc(a.b, (unsigned)a.n * 16u);
    for (int r = 0; r < a.steps; r++) {
        const int j = (int)(((int64_t)i + a.shift) % a.n);
        if (i < a.n) {
            cplx v;
            if (a.coherent) v = ld_coh((r & 1) ? rb : ra, j, 0);
            else v = ((r & 1) ? a.b : a.a)[j];
            v.x += 1.; v.y -= 1.;
            if (a.coherent) st_coh((r & 1) ? ra : rb, i, 0, v);
            else ((r & 1) ? a.a : a.b)[i] = v;
        }
        __builtin_amdgcn_s_waitcnt(0);   // this wave's stores are acknowledged before the workgroup publishes (as in the solvers)
        double one[1] = {1.};
        res_contrib<1>(sy, 0, one, (int)threadIdx.x >> 6);
        res_publish<1>(sy, 1);
        if (!res_collect<1>(sy, 1)) {
            if (threadIdx.x == 0) {
                __hip_atomic_store(a.abort_dev, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(a.abort_host, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
            return;
        }
    }
}
// rows_wrong: entries that are not what `steps` coherent steps must leave (-1: the launch gave up / was not co-resident)
int coherence_selftest(int steps, int coherent, int64_t *rows_wrong) {
    MGCR_CHECK(steps >= 1 && steps <= 100000 && rows_wrong, MGCR_ERR_INVALID, "mgcr_selftest_coherence: 1..100000 steps");
    MGCR_TRY(exchange_shared_init());
    ExchangeShared &sh = exchange_shared();
    const int cus = sh.cus > 256 ? 256 : sh.cus;
    const int nblk = cus >= 64 ? cus / 8 * 8 : cus;   // one workgroup per CU
    *rows_wrong = -1;
    if (!launch_is_coresident((const void *)coherence_selftest_kernel, RED_THREADS, 0, nblk)) {
        set_error("mgcr_selftest_coherence: %d workgroups are not co-resident on this device", nblk);
        return MGCR_ERR_UNSUPPORTED;
    }
    const int n = nblk * RED_THREADS, shift = 77777 % n;
    cplx *a = nullptr, *b = nullptr;
    MGCR_HIP(hipMalloc((void **)&a, sizeof(cplx) * (size_t)n));
    MGCR_HIP(hipMalloc((void **)&b, sizeof(cplx) * (size_t)n));
    std::vector<cplx> h((size_t)n);
    for (int i = 0; i < n; i++) h[(size_t)i] = make_double2((double)i, -(double)i);
    MGCR_HIP(hipMemcpyAsync(a, h.data(), sizeof(cplx) * (size_t)n, hipMemcpyHostToDevice, ctx().stream));
    MGCR_HIP(hipMemsetAsync(b, 0xff, sizeof(cplx) * (size_t)n, ctx().stream));
    CohArgs ca;
    ca.a = a; ca.b = b; ca.n = n; ca.shift = shift; ca.steps = steps; ca.coherent = coherent; ca.nblk = nblk;
    ca.slots = sh.slots; ca.abort_dev = sh.abort_dev; ca.abort_host = sh.abort_host;
    ca.gen0 = exchange_take_generations((unsigned)steps + 1u);
    ca.spin_limit = getenv("MGCR_TEST_RESIDENT_SPIN_LIMIT") ? atoi(getenv("MGCR_TEST_RESIDENT_SPIN_LIMIT")) : RES_SPIN_LIMIT;
    hipLaunchKernelGGL(coherence_selftest_kernel, dim3((unsigned)nblk), dim3(RED_THREADS), 0, ctx().stream, ca);
    MGCR_HIP(hipGetLastError());
    MGCR_HIP(hipMemcpyAsync(h.data(), (steps & 1) ? b : a, sizeof(cplx) * (size_t)n, hipMemcpyDeviceToHost, ctx().stream));
    MGCR_HIP(hipStreamSynchronize(ctx().stream));
    hipFree(a); hipFree(b);
    if (*(volatile int *)sh.abort_host != 0) {   // (the self-test's own give-up: does not switch the solvers' paths off)
        *(volatile int *)sh.abort_host = 0;
        MGCR_HIP(hipMemsetAsync(sh.abort_dev, 0, sizeof(unsigned), ctx().stream));
        set_error("mgcr_selftest_coherence: a workgroup waited too long for the others");
        return MGCR_ERR_HIP;
    }
    int64_t bad = 0;
    for (int i = 0; i < n; i++) {
        const double want = (double)(((int64_t)i + (int64_t)steps * shift) % n) + steps;
        if (h[(size_t)i].x != want || h[(size_t)i].y != -(want - steps) - steps) bad++;
    }
    *rows_wrong = bad;
    return MGCR_OK;
}

}  // namespace mgcr
