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
};

inline RowMat row_mat(const CsrDev &A, bool shift, cplx k) {
    RowMat m;
    m.npad = A.npad; m.W = A.W; m.npat = A.npat; m.col = A.ell_col;
    m.val = A.ell_val_re ? (const void *)A.ell_val_re : (const void *)A.ell_val;
    m.realv = A.pat_mode == 1 ? (A.pat_real ? 1 : 0) : (A.ell_val_re ? 1 : 0);
    m.pid = A.pat_id; m.poff = A.pat_off; m.pre = A.pat_re; m.pim = A.pat_im;
    m.shift = shift ? 1 : 0; m.k = k;
    m.xh = nullptr; m.n_own = INT32_MAX;
    return m;
}
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

}  // namespace mgcr
