// Device-side state of a GCR solve shared by gcr.hip (driver + BLAS-1 step kernels) and
// gcr_fused.hip (step kernels that embed the operator apply).
#pragma once
#include "internal.h"
#include "reduce.h"

namespace mgcr {

constexpr int ND = 8;    // directions per multidot / classic build launch
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


// x += alpha p_slot, recorded in terms of P0 and D_1..D_slot (one thread)
__device__ __forceinline__ void lean_pending_update(LeanCoef *lc, int slot, cplx alpha) {
    if (slot == 0) {
        lc->cx[0] = alpha;
        for (int m = 1; m < LND; m++) lc->cx[m] = make_double2(0., 0.);
    } else {
        lc->cx[0] = cadd(lc->cx[0], cmul(alpha, lc->t[slot]));
        for (int m = 1; m <= slot; m++) lc->cx[m] = cadd(lc->cx[m], cmul(alpha, lc->T[slot * LND + m]));
    }
}

}  // namespace mgcr
