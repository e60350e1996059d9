// Device-side state of a GCR solve shared by gcr.hip (driver + BLAS-1 step kernels) and
// gcr_fused.hip (step kernels that embed the operator apply).
#pragma once
#include "internal.h"
#include "reduce.h"

namespace mgcr {

constexpr int ND = 8;  // directions per multidot / build launch
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

struct DirPtrs {
    const cplx *ps[ND];
    const cplx *aps[ND];
    int slot[ND];
};

// lean restart cycles: p_k = t[k] P0 + sum_{1<=m<=k} T[k][m] D_m;  cx = coefficients of the pending x update
struct LeanCoef {
    cplx T[ND * ND];
    cplx t[ND];
    cplx cx[ND];
};


// x += alpha p_slot, recorded in terms of P0 and D_1..D_slot (one thread)
__device__ __forceinline__ void lean_pending_update(LeanCoef *lc, int slot, cplx alpha) {
    if (slot == 0) {
        lc->cx[0] = alpha;
        for (int m = 1; m < ND; m++) lc->cx[m] = make_double2(0., 0.);
    } else {
        lc->cx[0] = cadd(lc->cx[0], cmul(alpha, lc->t[slot]));
        for (int m = 1; m <= slot; m++) lc->cx[m] = cadd(lc->cx[m], cmul(alpha, lc->T[slot * ND + m]));
    }
}

}  // namespace mgcr
