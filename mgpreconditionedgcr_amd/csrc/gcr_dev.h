// Device-side state of a GCR solve shared by gcr.hip (driver + BLAS-1 step kernels) and
// gcr_fused.hip (step kernels that embed the operator apply).
#pragma once
#include "internal.h"
#include "reduce.h"

namespace mgcr {

constexpr int ND = 8;    // directions per multidot / classic build launch
constexpr int FND = 10;   // directions whose dot products the fused apply kernels take (restart 10 — the reference's usual setting — needs no second pass)
constexpr int LND = 16;  // slots a lean restart cycle can have (restart <= 16)
#ifndef MGCR_NT_SLOTS
#define MGCR_NT_SLOTS 1
#endif
constexpr bool NTS = MGCR_NT_SLOTS != 0;  // non-temporal access to the old direction slots

struct DevState {
    // Iteration at which the solve ended (INT_MAX while running).  Every kernel of the solve gets the
    // iteration number `it` it belongs to and returns at once when stop_at < it.  The bookkeeping of
    // step k (inside build_kernel) writes stop_at = k, which is >= the `it` of every kernel of step
    // k: no kernel ever acts on a value written by a concurrently running workgroup of itself.
    int stop_at;
    // Iteration numbers reach the kernels as base + it: `it` is a launch argument counted from the last
    // advance_kernel, `base` lives here.  Eager launches never advance (base = 0, it = global count);
    // a captured restart cycle (hipGraph) is replayed with it = 1..R and followed by base += R.
    int base;
    int iter;      // global_count
    int npend;     // x updates deferred so far in this restart cycle (see xr_update_kernel)
    double bnorm2; // |b|^2
    double rr;     // |r|^2 of the last finished step
    double tol2;
};

// Where a finished solve left its recurrence residual, by the number of steps it actually ran (a solve that
// converged early skipped the later steps' kernels): r[k] = residual after step k, k = 1..LND.  Consumers read
// r[st->iter] (mg.hip: the residual a V-cycle restricts after its pre-smoother, without recomputing b - A x).
// A solve asked to (gcr_set_defer_residual) does not even form the residual of its LAST possible step: if st->iter ==
// last_it the consumer computes it on the fly as r_prev - alpha * ap (alpha left on the device by alpha_only_kernel).
struct ResidualSel {
    const cplx *r[LND + 1];
    const DevState *st;
    const cplx *r_prev, *ap, *alpha;
    int last_it;   // 0: no deferred step
};

// x updates a finished solve left pending on request (gcr_set_keep_pending): x = sum_{j < st->npend} coef[j] * v[j],
// from x0 = 0.  The V-cycle adds its coarse-grid correction in the same pass (mg.hip expand_add_kernel).
struct PendingX {
    const cplx *v[LND];
    const cplx *coef;
    const DevState *st;
};

struct DirPtrs {  // the classic kernels use the first ND entries (one chunk), the lean ones up to LND
    const cplx *ps[LND];
    const cplx *aps[LND];
    int slot[LND];
};

// lean restart cycles: p_k = t[k] P0 + sum_{1<=m<=k} T[k][m] D_m;  cx = coefficients of the pending x update
struct LeanCoef {
    cplx T[LND * LND];
    cplx t[LND];
    cplx cx[LND];
};


// Row -> (workgroup, thread, trip) map shared by multidot_kernel (gcr.hip) and step_apply_kernel (gcr_fused.hip), so
// that either yields the same partial sums.  Banded (grids that are a multiple of 8 workgroups): the rows are cut
// into 8 contiguous bands, one per XCD; the `per` logical workgroups of a band sweep it together, per * RED_THREADS
// rows per trip.  Every XCD then walks consecutive planes of a stencil operand over time and finds the previous
// ones in ITS L2 — with a plain grid-stride the 8 XCDs work on 8 adjacent slices per trip and each slice's
// neighbours are fetched by three L2s (measured at 256^3: 1.73 GB of HBM traffic for an 0.84 GB launch).
// Not banded (small grids): plain grid-stride.
struct RowMap {
    int64_t band;  // rows per band (multiple of RED_THREADS); 0 = not banded
    int per;       // logical workgroups per band
    int xg;        // not banded: consecutive logical workgroups per XCD (0: one contiguous run of grid / 8 per XCD and trip)
    int nb;        // banded: number of bands (logical workgroups per * nb .. grid - 1 have no rows)
    int strips;    // banded: an XCD owns a strip of ceil(per / 8) neighbouring workgroups of EVERY band (else: whole bands, nb == 8)
    int64_t plane; // != 0: ragged plane walk — planes of `plane` rows (a multiple of 64, not of 1024): a band's per = ceil(plane / 1024)
                   // workgroups tile one plane (the last tile is short: its threads beyond the plane's end have no rows) and step by `plane`
};
// Logical workgroup of physical workgroup b (the hardware deals workgroups round-robin over the 8 XCDs, b & 7) for the
// fused apply kernels (gcr_fused.hip); partial sums are indexed by the logical number, so results do not depend on it.
// xg = 0: XCD x owns the contiguous run [x * per, (x + 1) * per) of every trip.  xg = G > 0: XCDs take turns, G logical
// workgroups at a time — a row's far neighbours (rows +- n^2 of a 3-D grid = +- 16 workgroups at 128^3) then belong to
// the SAME XCD when (workgroups per plane / G) is a multiple of 8, and only the +- n rows at the ends of a run of G
// workgroups are fetched by two L2s.
// Banded in strips (the plane-walk map below): XCD x owns workgroups [x tq, (x + 1) tq) of every band, tq = ceil(per / 8) — a strip of
// neighbouring tiles of every plane, so that only the strip's two edges fetch their +- n neighbours from another L2; its grid / 8
// physical workgroups are dealt band by band, and those left over take the logical numbers behind the last band (no rows: they
// write the zero partial sums of those slots).  A bijection of [0, grid), whatever per and nb.
__device__ __forceinline__ int logical_workgroup(const RowMap &rm, int b, int grid) {
    if (grid & 7) return b;
    const int x = b & 7, q = b >> 3;
    if (rm.xg > 0 && !rm.band) return (q / rm.xg) * (8 * rm.xg) + x * rm.xg + (q % rm.xg);
    if (rm.band && rm.strips) {
        const int gx = grid >> 3, tq = (rm.per + 7) >> 3;
        const int lo = x * tq < rm.per ? x * tq : rm.per;
        const int vt = rm.per - lo < tq ? rm.per - lo : tq;
        const int c = rm.nb * vt;
        if (q < c) return (q / vt) * rm.per + lo + q % vt;
        return rm.nb * rm.per + gx * x - rm.nb * lo + (q - c);
    }
    return x * (grid >> 3) + q;
}
// `reach` = how far a row's gathers go (CsrDev::reach; 0 = unknown).  Banding pays when that is a sizeable part of
// what one XCD covers per trip of a plain grid-stride (Poisson 256^3: 65536 of 65536 rows — 1 170 against 1 130 it/s);
// when the neighbours mostly stay inside the XCD's slice anyway (128^3: 16384 of 65536) the single sweep front of
// the plain map is faster (9.8 k against 9.4 k it/s).
// Plane walk (full grids of 512 workgroups, reach a multiple of 1024 rows — the plane of a 3-D grid whose planes hold a multiple of
// 1024 sites, 32 x 1024 <= plane <= 512 x 1024): a band's `per` = reach / 1024 workgroups tile ONE plane and step from plane to
// plane, so the thread that owns row i owns rows i +- reach one trip earlier / later — what the windowed kernels' CARRY builds on
// (gcr_fused.hip: the far neighbours never leave the registers).  nb = floor(64 / ceil(per / 8)) bands keep every XCD within its 64
// workgroups; with per = 64 (256 x 256 planes) this IS the 8-band map above.
inline RowMap make_row_map(int64_t n, int g, int64_t reach) {
    RowMap m{0, 0, 0, 0, 0, 0};
    if (g >= 64 && g % 8 == 0) {
        // XCDs take turns G logical workgroups at a time, G chosen so that a row's farthest neighbours (reach rows away =
        // P workgroups) belong to the same XCD: P a multiple of 8 G.  Poisson 128^3 (P = 16, G = 2), fused apply + dots:
        // 34.8 us against 35.9 with one contiguous run per XCD and trip (G = 64), 35.0 with G = 1, 37.1 / 40.5 with G = 4 / 8.
        // MGCR_XCD_GROUP overrides (0 = contiguous run).
        static const int xg_env = getenv("MGCR_XCD_GROUP") ? atoi(getenv("MGCR_XCD_GROUP")) : -1;
        const int per = g / 8;
        if (xg_env >= 0) {
            if (xg_env == 0 || per % xg_env == 0) m.xg = xg_env;
        } else if (reach > 0 && reach % RED_THREADS == 0 && (reach / RED_THREADS) % 8 == 0) {
            int64_t a = reach / RED_THREADS / 8, b = per;   // G = gcd(P / 8, per)
            while (b) { const int64_t t = a % b; a = b; b = t; }
            m.xg = (int)a;
        }
    }
    const int64_t slice = (int64_t)g * RED_THREADS / 8;
    const bool wide = reach > 0 ? 2 * reach >= slice : n >= ((int64_t)1 << 23);
    if (g >= 64 && g % 8 == 0 && wide) {
        m.per = g / 8;
        m.nb = 8;
        static const int walk_env = getenv("MGCR_PLANE_WALK") ? atoi(getenv("MGCR_PLANE_WALK")) : 1;   // 0: off, 2: strips for 64-wide bands too
        const int64_t T = reach > 0 && reach % RED_THREADS == 0 ? reach / RED_THREADS : 0;
        if (walk_env && g == RED_MAX_BLOCKS && T >= 32 && T <= 512 && (T != 64 || walk_env == 2)) {
            const int gx = g / 8, tq = (int)((T + 7) / 8);
            m.per = (int)T;
            m.nb = gx / tq;
            m.strips = 1;
        }
        m.band = ((n + m.nb - 1) / m.nb + RED_THREADS - 1) / RED_THREADS * RED_THREADS;
        // ... and for planes that are a multiple of 64 rows only (n = 200, 264, 328 ...; a wave's 64 rows must not straddle two planes' tiles:
        // the presence words of the stencil view are per aligned wave): the same walk with a short last tile per plane, whole planes per band
        const int64_t Tr = (reach + RED_THREADS - 1) / RED_THREADS;
        if (walk_env && g == RED_MAX_BLOCKS && T == 0 && reach > 0 && reach % 64 == 0 && Tr >= 32 && Tr <= 512) {
            const int gx = g / 8, tq = (int)((Tr + 7) / 8);
            m.per = (int)Tr;
            m.nb = gx / tq;
            m.strips = 1;
            m.plane = reach;
            const int64_t nplanes = (n + reach - 1) / reach;
            m.band = (nplanes + m.nb - 1) / m.nb * reach;
        }
    }
    return m;
}
// first row, one-past-last row and step of logical workgroup lb's thread
// valid (the windowed kernels, which walk the tiles with a uniform trip count): false for a thread of a short tile that lies beyond the
// plane's end — `first` is then still the row the thread's window entry belongs to; without `valid` such a thread gets an empty range
__device__ __forceinline__ void row_range(const RowMap &m, int lb, int nlogical, int64_t n, int64_t *first, int64_t *end, int64_t *step,
                                          bool *valid = nullptr) {
    if (valid) *valid = true;
    if (m.band) {
        const int64_t xb = lb / m.per;
        const int64_t in_plane = (int64_t)(lb % m.per) * RED_THREADS + threadIdx.x;
        *first = xb * m.band + in_plane;
        const int64_t e = (xb + 1) * m.band;
        *end = e < n ? e : n;
        *step = m.plane ? m.plane : (int64_t)m.per * RED_THREADS;
        if (m.plane && in_plane >= m.plane) {
            if (valid) *valid = false;
            else *first = *end;
        }
    } else {
        *first = (int64_t)lb * RED_THREADS + threadIdx.x;
        *end = n;
        *step = (int64_t)nlogical * RED_THREADS;
    }
}

// x += alpha p_slot, recorded in terms of P0 and D_1..D_slot: thread m < LND updates coefficient m
__device__ __forceinline__ void lean_pending_update(LeanCoef *lc, int slot, cplx alpha, int m) {
    if (slot == 0) {
        lc->cx[m] = m == 0 ? alpha : make_double2(0., 0.);
    } else if (m == 0) {
        lc->cx[0] = cadd(lc->cx[0], cmul(alpha, lc->t[slot]));
    } else if (m <= slot) {
        lc->cx[m] = cadd(lc->cx[m], cmul(alpha, lc->T[slot * LND + m]));
    }
}

// gcr_resident.hip: a lean restarted solve on a small stencil-view operator in one launch
bool gcr_resident_eligible(const Op *A, const mgcr_gcr_param &p, int storage, int restart, int64_t n, bool lean, bool nested_handoff);
int gcr_resident_run(Op *A, const mgcr_gcr_param &p, int storage, int restart, const cplx *rhs, cplx *x, bool from_zero, bool alpha_only_last,
                     DevState *st, double *hist, int hist_cap, cplx *ring /* 11 n entries */, SkipRef outer);

// gcr_stepbuild.hip: apply + dot products + direction build of a lean step in one launch
bool csr_step_build_eligible(const CsrDev &A, const DistCsr *dist, int lim);
int csr_step_build(const CsrDev &A, const cplx *x, bool shift, cplx k, const cplx *const *aps, int nd, DevState *st, int it, const double *partsR,
                   int nblkR, int strideR, double *hist, int hist_cap, const cplx *den, cplx *ap_out, double *partsA, LeanCoef *lc,
                   const RowMap &rm, cplx *xr_out = nullptr, cplx *xr_den_slot = nullptr, int xr_slot = 0, double *partsR_out = nullptr,
                   const cplx *const *close_ps = nullptr, cplx *close_p_out = nullptr, cplx *close_x = nullptr);

}  // namespace mgcr
