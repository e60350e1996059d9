// Device-side pieces of the Sparse SpMV shared by spmv.hip (the operator apply) and gcr_fused.hip
// (the GCR step kernels that apply the operator and take the step's dot products in one pass).
#pragma once
#include <climits>

#include "internal.h"
#include "reduce.h"

namespace mgcr {

// Workgroups are dealt round-robin over the 8 XCDs (blocks b and b+8 share an L2).  Give each XCD
// one contiguous band of rows, so that the x entries a stencil-like matrix re-reads (row +-1,
// +-n, +-n^2) stay in that XCD's 4 MiB L2 instead of being fetched by all eight.
__device__ __forceinline__ int64_t xcd_tile(int64_t ntiles) {
    int64_t b = blockIdx.x;
    int64_t per = (ntiles + 7) >> 3;
    return (b & 7) * per + (b >> 3);
}

// L = 1: one thread per row, entries in CSR order (bit-identical to the reference's row sum)
// gather of x: columns >= n_own live in the halo segment xh (multi-GPU row blocks); n_own is
// INT32_MAX and xh unused otherwise
__device__ __forceinline__ cplx gather_x(const cplx *__restrict__ x, const cplx *__restrict__ xh, int32_t n_own, int32_t j) {
    return j < n_own ? x[j] : xh[j - n_own];
}

// one stored value times an x entry; REALV: the value is a real fp64
// NT: the matrix stream is read exactly once per SpMV — load it non-temporally so that it does not
// displace the vectors (x, and the solver's r / Ar / direction slots) from L2 and the Infinity Cache.
template <bool REALV, bool NT = false>
__device__ __forceinline__ cplx vmul(const void *__restrict__ val, int64_t idx, cplx xv) {
    if (REALV) {
        const double *p = reinterpret_cast<const double *>(val) + idx;
        double v = NT ? __builtin_nontemporal_load(p) : *p;
        return make_double2(v * xv.x, v * xv.y);
    }
    const cplx *p = reinterpret_cast<const cplx *>(val) + idx;
    cplx v;
    if (NT) {
        v.x = __builtin_nontemporal_load(&p->x);
        v.y = __builtin_nontemporal_load(&p->y);
    } else {
        v = *p;
    }
    return cmul(v, xv);
}
template <bool NT>
__device__ __forceinline__ int32_t ldcol(const int32_t *__restrict__ p) { return NT ? __builtin_nontemporal_load(p) : *p; }


// One-thread-per-row view of a CsrDev (L = 1, no tail) for kernels that embed the row product.
// MODE 0: ELL slab (col + val); 1: row-pattern dictionary with values (table staged in LDS by the
// kernel); 2: row-pattern dictionary for the columns, values in the slab.
struct RowMat {
    int64_t npad;
    int32_t W;
    int32_t npat;
    const int32_t *col;    // MODE 0
    const void *val;       // MODE 0, 2: slab (double or cplx)
    int realv;             // values are real fp64
    const uint16_t *pid;   // MODE 1, 2
    const int32_t *poff;
    const double *pre, *pim;  // MODE 1
    int shift;
    cplx k;
    const cplx *xh;        // halo segment of x (row block of a distributed matrix), columns >= n_own
    int32_t n_own;
    // MODE 3: stencil view (CsrDev::sten_*)
    int32_t sten_ns, sten_stride, sten_last;   // slots, presence words per wave, last column (clamp)
    uint32_t sten_rare, sten_near;
    int32_t sten_pre;      // rare-tail layout: slot 7 is summed before the common slots
    int32_t sten_halo, sten_halo_f, sten_nwaves;   // sten_planes holds sten_nwaves rows + one all-zero row
    int32_t sten_off[STEN_MAX];
    double sten_re[STEN_MAX], sten_im[STEN_MAX];
    const uint64_t *sten_planes;
};

inline RowMat row_mat(const CsrDev &A, bool shift, cplx k) {
    RowMat m;
    m.npad = A.npad; m.W = A.W; m.npat = A.npat; m.col = A.ell_col;
    m.val = A.ell_val_re ? (const void *)A.ell_val_re : (const void *)A.ell_val;
    m.realv = A.pat_mode == 1 ? (A.pat_real ? 1 : 0) : (A.ell_val_re ? 1 : 0);
    m.pid = A.pat_id; m.poff = A.pat_off; m.pre = A.pat_re; m.pim = A.pat_im;
    m.shift = shift ? 1 : 0; m.k = k;
    m.xh = nullptr; m.n_own = INT32_MAX;
    m.sten_ns = A.sten_ns; m.sten_stride = A.sten_stride; m.sten_last = (int32_t)A.ncol - 1; m.sten_rare = A.sten_rare; m.sten_pre = A.sten_pre;
    m.sten_near = A.sten_near; m.sten_halo = A.sten_halo; m.sten_halo_f = A.sten_halo_f; m.sten_nwaves = (int32_t)(A.npad / 64);
    for (int c = 0; c < STEN_MAX; c++) { m.sten_off[c] = A.sten_off[c]; m.sten_re[c] = A.sten_re[c]; m.sten_im[c] = A.sten_im[c]; }
    m.sten_planes = A.sten_planes;
    return m;
}
// slots the stencil kernels are instantiated for (0: no stencil view)
inline int sten_slots(const CsrDev &A) { return A.sten_ns == 0 ? 0 : A.sten_kernel_ns; }
inline size_t row_mat_lds_bytes(const CsrDev &A) { return A.pat_mode == 1 ? (size_t)A.npat * A.W * (A.pat_real ? 12 : 20) : 0; }

// pattern table -> LDS (MODE 1); call from every thread of the workgroup, ends with a barrier
struct PatLds {
    const int32_t *off;
    const double *re, *im;
};
__device__ __forceinline__ PatLds stage_patterns(const RowMat &m, unsigned char *smem) {
    const int32_t ne = m.npat * m.W;
    double *wre = reinterpret_cast<double *>(smem);
    double *wim = wre + (m.realv ? 0 : ne);
    int32_t *woff = reinterpret_cast<int32_t *>(wim + ne);
    for (int32_t e = threadIdx.x; e < ne; e += blockDim.x) {
        woff[e] = m.poff[e];
        wre[e] = m.pre[e];
        if (!m.realv) wim[e] = m.pim[e];
    }
    __syncthreads();
    return PatLds{woff, wre, wim};
}

// Row `row` of A times x, entries multiplied and added in storage (= CSR) order like the SpMV kernels.
// t0 = (pattern id) * W for MODE 1 / 2.  XF(j) yields x_j — a plain load, or a value recomputed on the fly.
template <int MODE, int WT, class XF>
__device__ __forceinline__ cplx row_product(const RowMat &m, int64_t row, int32_t t0, const PatLds &pl, XF xf) {
    const int32_t W = WT ? WT : m.W;
    cplx sum = make_double2(0., 0.);
    auto column = [&](int32_t c) -> int32_t {
        if (MODE == 0) return ldcol<true>(m.col + (int64_t)c * m.npad + row);
        if (MODE == 1) return (int32_t)row + pl.off[t0 + c];
        return (int32_t)row + m.poff[t0 + c];
    };
    auto term = [&](int32_t c, cplx xv) -> cplx {
        if (MODE == 1) {
            if (m.realv) {
                double v = pl.re[t0 + c];
                return make_double2(v * xv.x, v * xv.y);
            }
            return cmul(make_double2(pl.re[t0 + c], pl.im[t0 + c]), xv);
        }
        return m.realv ? vmul<true, true>(m.val, (int64_t)c * m.npad + row, xv) : vmul<false, true>(m.val, (int64_t)c * m.npad + row, xv);
    };
    if (WT) {
        int32_t j[WT ? WT : 1];
        cplx xv[WT ? WT : 1];
#pragma unroll
        for (int32_t c = 0; c < W; c++) j[c] = column(c);
#pragma unroll
        for (int32_t c = 0; c < W; c++) xv[c] = xf(j[c]);
#pragma unroll
        for (int32_t c = 0; c < W; c++) sum = cadd(sum, term(c, xv[c]));
    } else {
#pragma unroll 4
        for (int32_t c = 0; c < W; c++) sum = cadd(sum, term(c, xf(column(c))));
    }
    return sum;
}

// MODE 3 / 4.  Row `row` of A times x through the stencil view: slot c is column row + sten_off[c] (clamped into the matrix;
// a clamped load is never used), present in this row iff bit (row & 63) of the wave's presence word c is set.  Slots
// are in ascending column order = CSR order and absent ones are skipped, so the sum is the one row_product forms minus
// its "+ 0 * x" padding terms: same bits for finite x.  The x loads depend on the row number alone — the presence
// words arrive through the scalar cache meanwhile — where the dictionary kernels chain id -> table -> gather.
// All 64 lanes of a wave must hold consecutive rows starting at a multiple of 64.
// RARE (NS = 9): slots 0..6 are the common ones, slots 7 and 8 are rarely present and lie BEHIND every common slot in
// column order (the halo columns of a slab's first and last plane).  They are looked at after the common sum, and
// loaded only by the waves whose presence word is not 0 — the other waves run exactly the 7-slot kernel, registers
// included (a 9-slot kernel with all loads up front costs 3.3 us of 36 per fused apply at 128^3).
constexpr int STEN_COMMON = 7;
// The presence words never change while a kernel runs: read them through the CONSTANT address space, which makes the
// compiler use scalar loads (s_load) everywhere.  From a plain global pointer it does so only where it can prove that no
// store of the kernel precedes the load — in the fused step kernels that held for some direction counts and not for
// others (vector loads of 14 VGPRs instead).
typedef const uint64_t __attribute__((address_space(4))) *sten_planes_ptr;
__device__ __forceinline__ sten_planes_ptr sten_wave_planes(const RowMat &m, int32_t wave) {
    return (sten_planes_ptr)(uintptr_t)(m.sten_planes + (int64_t)wave * m.sten_stride);
}
// REALV: 1 real values, 0 complex, -1 decided per slot at run time (m.realv, a uniform branch)
template <int REALV>
__device__ __forceinline__ cplx sten_term(const RowMat &m, int c, cplx xv) {
    if (REALV > 0 || (REALV < 0 && m.realv)) return make_double2(m.sten_re[c] * xv.x, m.sten_re[c] * xv.y);
    return cmul(make_double2(m.sten_re[c], m.sten_im[c]), xv);
}
// Rare-tail layout with RowMat::sten_pre: slot STEN_COMMON (7) comes FIRST in its rows' storage order (spmv.hip sten_try) — its term
// opens the row sum.  plw = the wave's presence word of that slot; xf(j) fetches column j.
template <int REALV, class XF>
__device__ __forceinline__ cplx sten_pre_sum(const RowMat &m, int64_t row, uint64_t plw, int lane, XF xf) {
    cplx sum = make_double2(0., 0.);
    if (m.sten_pre && plw != 0ull) {   // wave-uniform: a wave of the block's first plane
        int32_t j = (int32_t)row + m.sten_off[STEN_COMMON];
        j = j < 0 ? 0 : j > m.sten_last ? m.sten_last : j;
        const cplx xr = xf(j);
        const bool on = (plw >> lane & 1ull) != 0ull;
        const cplx nsum = cadd(sum, sten_term<REALV>(m, STEN_COMMON, xr));
        sum.x = on ? nsum.x : sum.x;
        sum.y = on ? nsum.y : sum.y;
    }
    return sum;
}
template <int NS, bool RARE, int REALV, class XF>
__device__ __forceinline__ cplx sten_row_product_t(const RowMat &m, int64_t row, XF xf) {
    static_assert(!RARE || NS == 9, "the rare-tail layout has 7 common + 2 rare slots");
    constexpr int NC = RARE ? STEN_COMMON : NS;
    const int32_t wave = __builtin_amdgcn_readfirstlane((int32_t)(row >> 6));
    const sten_planes_ptr pp = sten_wave_planes(m, wave);
    uint64_t pl[NS];
#pragma unroll
    for (int c = 0; c < NS; c++) pl[c] = pp[c];
    cplx xv[NC];
#pragma unroll
    for (int c = 0; c < NC; c++) {
        int32_t j = (int32_t)row + m.sten_off[c];
        j = j < 0 ? 0 : j > m.sten_last ? m.sten_last : j;
        xv[c] = xf(j);
    }
    // the presence bits must not be looked at before every load is in flight (the scheduler would otherwise
    // wait for them first and serialise the two round trips)
    __builtin_amdgcn_sched_barrier(0);
    const int lane = (int)(threadIdx.x & 63);
    cplx sum = make_double2(0., 0.);
    if constexpr (RARE) sum = sten_pre_sum<REALV>(m, row, pl[NC], lane, xf);
#pragma unroll
    for (int c = 0; c < NC; c++) {
        const bool on = (pl[c] >> lane & 1ull) != 0ull;
        const cplx nsum = cadd(sum, sten_term<REALV>(m, c, xv[c]));
        sum.x = on ? nsum.x : sum.x;
        sum.y = on ? nsum.y : sum.y;
    }
    if (RARE) {
#pragma unroll
        for (int c = NC; c < NS; c++)
            if (pl[c] != 0ull && !(c == NC && m.sten_pre)) {   // wave-uniform
                int32_t j = (int32_t)row + m.sten_off[c];
                j = j < 0 ? 0 : j > m.sten_last ? m.sten_last : j;
                const cplx xr = xf(j);
                const bool on = (pl[c] >> lane & 1ull) != 0ull;
                const cplx nsum = cadd(sum, sten_term<REALV>(m, c, xr));
                sum.x = on ? nsum.x : sum.x;
                sum.y = on ? nsum.y : sum.y;
            }
    }
    return sum;
}
template <int NS, bool RARE, class XF>
__device__ __forceinline__ cplx sten_row_product(const RowMat &m, int64_t row, XF xf) {
    // Real or complex slot values: the rare-tail kernels branch once per row (two copies of the body: 37.3 against 38.6 us
    // per fused apply at 128^3); the plain ones decide per slot — with two copies of THEIR body the fused kernels spill
    // (188 B of scratch, 128 us instead of 36).
    if constexpr (RARE) return m.realv ? sten_row_product_t<NS, RARE, 1>(m, row, xf) : sten_row_product_t<NS, RARE, 0>(m, row, xf);
    else return sten_row_product_t<NS, RARE, -1>(m, row, xf);
}

// the row product of the GCR step kernels (gcr_fused.hip): MODE 0..2 as above, 3 / 4 = stencil view without / with
// rarely present slots, WT = its slot count
template <int MODE, int WT, class XF>
__device__ __forceinline__ cplx fused_row_product(const RowMat &m, int64_t row, int32_t t0, const PatLds &pl, XF xf) {
    if constexpr (MODE >= 3) return sten_row_product<WT, MODE == 4>(m, row, xf);
    else return row_product<MODE, WT>(m, row, t0, pl, xf);
}

}  // namespace mgcr
