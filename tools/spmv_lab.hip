// Kernel laboratory for the row-pattern SpMV (not part of the product; built and run by hand on the GPU box):
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off tools/spmv_lab.hip -o gpurun_out/spmv_lab
//   gpurun_out/spmv_lab 128
// Times candidate kernels for y = A x on the 7-point Poisson matrix stored as a row-pattern dictionary, with cold
// caches (a 512 MiB sweep between launches) and back to back, and checks every candidate against the first one
// bit for bit.  Variants that win move into csrc/spmv.hip.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <map>
#include <vector>

typedef double2 cplx;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

constexpr int W = 7;
constexpr int NSMAX = 12;
struct Sup { int32_t off[NSMAX]; int ns; };

__device__ __forceinline__ int64_t xcd_tile(int64_t t, int64_t ntiles) {
    int64_t per = (ntiles + 7) >> 3;
    return (t & 7) * per + (t >> 3);
}

// ---- A: copy ------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_copy(int64_t n, const cplx *__restrict__ x, cplx *__restrict__ y) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) y[i] = x[i];
}

// RPT elements per thread, a workgroup's elements contiguous, all loads issued before the first store
template <int RPT>
__global__ void __launch_bounds__(256) k_copy_n(int64_t n, const cplx *__restrict__ x, cplx *__restrict__ y) {
    const int64_t base = (int64_t)blockIdx.x * 256 * RPT + threadIdx.x;
    cplx v[RPT];
#pragma unroll
    for (int h = 0; h < RPT; h++) v[h] = base + h * 256 < n ? x[base + h * 256] : make_double2(0., 0.);
#pragma unroll
    for (int h = 0; h < RPT; h++)
        if (base + h * 256 < n) y[base + h * 256] = v[h];
}

// ---- C: the product's pat_spmv_lds (table staged per workgroup, id -> LDS -> gathers) -----------
__global__ void __launch_bounds__(256) k_pat_lds(int64_t n, int64_t ntiles, int32_t npat, const uint16_t *__restrict__ pid,
                                                  const int32_t *__restrict__ poff, const double *__restrict__ pre,
                                                  const cplx *__restrict__ x, cplx *__restrict__ y) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int64_t tile = xcd_tile(blockIdx.x, ntiles);
    if (tile >= ntiles) return;
    const int32_t ne = npat * W;
    double *sre = reinterpret_cast<double *>(smem);
    int32_t *soff = reinterpret_cast<int32_t *>(sre + ne);
    const int64_t rloc = tile * 256 + threadIdx.x;
    const bool live = rloc < n;
    const int64_t row = live ? rloc : 0;
    const int32_t t0 = (int32_t)__builtin_nontemporal_load(pid + row) * W;
    for (int32_t e = threadIdx.x; e < ne; e += 256) { soff[e] = poff[e]; sre[e] = pre[e]; }
    __syncthreads();
    cplx xv[W];
#pragma unroll
    for (int c = 0; c < W; c++) xv[c] = x[(int32_t)row + soff[t0 + c]];
    cplx sum = make_double2(0., 0.);
#pragma unroll
    for (int c = 0; c < W; c++) { double v = sre[t0 + c]; sum.x += v * xv[c].x; sum.y += v * xv[c].y; }
    if (live) y[row] = sum;
}

// ---- S1: superset offsets: the id and the gathers are independent -> ONE memory round trip -------
// every pattern's offsets are a subsequence of the ascending superset `sup`; slot c of pattern p holds the value
// (sval) and a presence bit (smask); absent slots are skipped, so the row sum has the reference's order and bits
template <int NS>
__global__ void __launch_bounds__(256) k_sup(int64_t n, int64_t ntiles, int32_t npat, Sup sup, const uint16_t *__restrict__ pid,
                                              const double *__restrict__ gval, const uint32_t *__restrict__ gmask,
                                              const cplx *__restrict__ x, cplx *__restrict__ y) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int64_t tile = xcd_tile(blockIdx.x, ntiles);
    if (tile >= ntiles) return;
    double *sval = reinterpret_cast<double *>(smem);
    uint32_t *smask = reinterpret_cast<uint32_t *>(sval + npat * NS);
    const int64_t rloc = tile * 256 + threadIdx.x;
    const bool live = rloc < n;
    const int32_t row = (int32_t)(live ? rloc : 0);
    const int32_t p = (int32_t)__builtin_nontemporal_load(pid + row);
    cplx xv[NS];
#pragma unroll
    for (int c = 0; c < NS; c++) {
        int32_t j = row + sup.off[c];
        j = j < 0 ? 0 : j >= (int32_t)n ? (int32_t)n - 1 : j;
        xv[c] = x[j];
    }
    __builtin_amdgcn_sched_barrier(0);
    for (int32_t e = threadIdx.x; e < npat * NS; e += 256) sval[e] = gval[e];
    for (int32_t e = threadIdx.x; e < npat; e += 256) smask[e] = gmask[e];
    __syncthreads();
    const uint32_t m = smask[p];
    cplx sum = make_double2(0., 0.);
#pragma unroll
    for (int c = 0; c < NS; c++) {
        double v = sval[p * NS + c];
        { const bool on = m >> c & 1u; const double nx = sum.x + v * xv[c].x, ny = sum.y + v * xv[c].y; sum.x = on ? nx : sum.x; sum.y = on ? ny : sum.y; }
    }
    if (live) y[row] = sum;
}

// ---- S2: the same, persistent: table staged once per workgroup, next tile's loads issued before this tile's sum ----
template <int NS, int BLK>
__global__ void __launch_bounds__(BLK) k_sup_persist(int64_t n, int64_t ntiles, int32_t npat, Sup sup, const uint16_t *__restrict__ pid,
                                                      const double *__restrict__ gval, const uint32_t *__restrict__ gmask,
                                                      const cplx *__restrict__ x, cplx *__restrict__ y) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double *sval = reinterpret_cast<double *>(smem);
    uint32_t *smask = reinterpret_cast<uint32_t *>(sval + npat * NS);
    int64_t t = blockIdx.x;
    cplx xa[NS], xb[NS];
    int32_t pa = 0, pb = 0, ra = 0, rb = 0;
    bool la = false, lb = false;
    auto issue = [&](int64_t tt, cplx (&xv)[NS], int32_t &p, int32_t &row, bool &live) {
        const int64_t rloc = xcd_tile(tt, ntiles) * BLK + threadIdx.x;
        live = tt < ntiles && rloc < n;
        row = (int32_t)(live ? rloc : 0);
        p = (int32_t)__builtin_nontemporal_load(pid + row);
#pragma unroll
        for (int c = 0; c < NS; c++) {
            int32_t j = row + sup.off[c];
            j = j < 0 ? 0 : j >= (int32_t)n ? (int32_t)n - 1 : j;
            xv[c] = x[j];
        }
    };
    auto finish = [&](const cplx (&xv)[NS], int32_t p, int32_t row, bool live) {
        const uint32_t m = smask[p];
        cplx sum = make_double2(0., 0.);
#pragma unroll
        for (int c = 0; c < NS; c++) {
            double v = sval[p * NS + c];
            { const bool on = m >> c & 1u; const double nx = sum.x + v * xv[c].x, ny = sum.y + v * xv[c].y; sum.x = on ? nx : sum.x; sum.y = on ? ny : sum.y; }
        }
        if (live) y[row] = sum;
    };
    issue(t, xa, pa, ra, la);
    for (int32_t e = threadIdx.x; e < npat * NS; e += BLK) sval[e] = gval[e];
    for (int32_t e = threadIdx.x; e < npat; e += BLK) smask[e] = gmask[e];
    __syncthreads();
    for (;;) {
        if (t >= ntiles) break;
        issue(t + gridDim.x, xb, pb, rb, lb);
        finish(xa, pa, ra, la);
        t += gridDim.x;
        if (t >= ntiles) break;
        issue(t + gridDim.x, xa, pa, ra, la);
        finish(xb, pb, rb, lb);
        t += gridDim.x;
    }
}

// ---- S3: superset, no LDS at all: values and masks through the scalar/vector cache (table is 1.5 KB) ----
template <int NS>
__global__ void __launch_bounds__(256) k_sup_nolds(int64_t n, int64_t ntiles, int32_t npat, Sup sup, const uint16_t *__restrict__ pid,
                                                    const double *__restrict__ gval, const uint32_t *__restrict__ gmask,
                                                    const cplx *__restrict__ x, cplx *__restrict__ y) {
    const int64_t tile = xcd_tile(blockIdx.x, ntiles);
    if (tile >= ntiles) return;
    const int64_t rloc = tile * 256 + threadIdx.x;
    const bool live = rloc < n;
    const int32_t row = (int32_t)(live ? rloc : 0);
    const int32_t p = (int32_t)__builtin_nontemporal_load(pid + row);
    cplx xv[NS];
#pragma unroll
    for (int c = 0; c < NS; c++) {
        int32_t j = row + sup.off[c];
        j = j < 0 ? 0 : j >= (int32_t)n ? (int32_t)n - 1 : j;
        xv[c] = x[j];
    }
    const uint32_t m = gmask[p];
    cplx sum = make_double2(0., 0.);
#pragma unroll
    for (int c = 0; c < NS; c++) {
        double v = gval[p * NS + c];
        { const bool on = m >> c & 1u; const double nx = sum.x + v * xv[c].x, ny = sum.y + v * xv[c].y; sum.x = on ? nx : sum.x; sum.y = on ? ny : sum.y; }
    }
    if (live) y[row] = sum;
}

// ---- F: fixed 7-point stencil, no id, no table (what the gathers alone cost) -------------------
__global__ void __launch_bounds__(256) k_fixed(int64_t n, int64_t ntiles, Sup sup, const cplx *__restrict__ x, cplx *__restrict__ y) {
    const int64_t tile = xcd_tile(blockIdx.x, ntiles);
    if (tile >= ntiles) return;
    const int64_t rloc = tile * 256 + threadIdx.x;
    if (rloc >= n) return;
    const int32_t row = (int32_t)rloc;
    cplx sum = make_double2(0., 0.);
#pragma unroll
    for (int c = 0; c < 7; c++) {
        int32_t j = row + sup.off[c];
        j = j < 0 ? 0 : j >= (int32_t)n ? (int32_t)n - 1 : j;
        cplx v = x[j];
        double a = c == 3 ? 6. : -1.;
        sum.x += a * v.x; sum.y += a * v.y;
    }
    y[row] = sum;
}


// ---- F1 / F3: how the time scales with the number of loads per row (results are wrong on purpose) ----
template <int NLOAD>
__global__ void __launch_bounds__(256) k_fixed_n(int64_t n, int64_t ntiles, Sup sup, const cplx *__restrict__ x, cplx *__restrict__ y) {
    const int64_t tile = xcd_tile(blockIdx.x, ntiles);
    if (tile >= ntiles) return;
    const int64_t rloc = tile * 256 + threadIdx.x;
    if (rloc >= n) return;
    const int32_t row = (int32_t)rloc;
    // NLOAD = 1: centre only; 3: centre and +-n^2; 5: all but +-1
    constexpr int sel1[1] = {3}, sel3[3] = {0, 3, 6}, sel5[5] = {0, 1, 3, 5, 6};
    cplx sum = make_double2(0., 0.);
#pragma unroll
    for (int q = 0; q < NLOAD; q++) {
        const int c = NLOAD == 1 ? sel1[q] : NLOAD == 3 ? sel3[q] : sel5[q];
        int32_t j = row + sup.off[c];
        j = j < 0 ? 0 : j >= (int32_t)n ? (int32_t)n - 1 : j;
        cplx v = x[j];
        double a = c == 3 ? 6. : -1.;
        sum.x += a * v.x; sum.y += a * v.y;
    }
    y[row] = sum;
}

// ---- L: x staged in LDS with a halo of H entries on both sides; offsets |off| <= H are served from LDS, the rest from
// global (issued before the barrier).  NEAR = bit mask of the superset slots served from LDS. ----
template <int NS, int BLK, unsigned NEAR, bool PAT>
__global__ void __launch_bounds__(BLK) k_tile(int64_t n, int64_t ntiles, int32_t npat, Sup sup, int32_t H, const uint16_t *__restrict__ pid,
                                              const double *__restrict__ gval, const uint32_t *__restrict__ gmask,
                                              const cplx *__restrict__ x, cplx *__restrict__ y) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int64_t tile = xcd_tile(blockIdx.x, ntiles);
    if (tile >= ntiles) return;
    cplx *sx = reinterpret_cast<cplx *>(smem);                 // [BLK + 2H]
    double *sval = reinterpret_cast<double *>(sx + BLK + 2 * H);
    uint32_t *smask = reinterpret_cast<uint32_t *>(sval + (PAT ? npat * NS : 0));
    const int64_t base = tile * BLK;
    const int64_t rloc = base + threadIdx.x;
    const bool live = rloc < n;
    const int32_t row = (int32_t)(live ? rloc : n - 1);
    int32_t p = 0;
    if (PAT) p = (int32_t)__builtin_nontemporal_load(pid + row);
    cplx xv[NS];
#pragma unroll
    for (int c = 0; c < NS; c++)
        if (!(NEAR >> c & 1u)) {
            int32_t j = row + sup.off[c];
            j = j < 0 ? 0 : j >= (int32_t)n ? (int32_t)n - 1 : j;
            xv[c] = x[j];
        }
    __builtin_amdgcn_sched_barrier(0);
    {   // window [base - H, base + BLK + H)
        sx[H + threadIdx.x] = x[row];
        if ((int)threadIdx.x < 2 * H) {
            const int t = threadIdx.x;
            int64_t j = t < H ? base - H + t : base + BLK + (t - H);
            j = j < 0 ? 0 : j >= n ? n - 1 : j;
            sx[t < H ? t : BLK + t] = x[j];
        }
    }
    if (PAT) {
        for (int32_t e = threadIdx.x; e < npat * NS; e += BLK) sval[e] = gval[e];
        for (int32_t e = threadIdx.x; e < npat; e += BLK) smask[e] = gmask[e];
    }
    __syncthreads();
    const uint32_t m = PAT ? smask[p] : 0x7fu;
    cplx sum = make_double2(0., 0.);
#pragma unroll
    for (int c = 0; c < NS; c++) {
        cplx v = (NEAR >> c & 1u) ? sx[H + (int)threadIdx.x + sup.off[c]] : xv[c];
        double a = PAT ? sval[p * NS + c] : (c == 3 ? 6. : -1.);
        { const bool on = m >> c & 1u; const double nx = sum.x + a * v.x, ny = sum.y + a * v.y; sum.x = on ? nx : sum.x; sum.y = on ? ny : sum.y; }
    }
    if (live) y[row] = sum;
}


// ---- M: stencil-mask storage: superset offsets + ONE value per slot (kernel arguments) + a presence mask per row ----
struct SlotVal { double v[NSMAX]; };
template <int NS, int BLK>
__global__ void __launch_bounds__(BLK) k_mask(int64_t n, int64_t ntiles, Sup sup, SlotVal sv, const uint16_t *__restrict__ rmask,
                                              const cplx *__restrict__ x, cplx *__restrict__ y) {
    const int64_t tile = xcd_tile(blockIdx.x, ntiles);
    if (tile >= ntiles) return;
    const int64_t rloc = tile * BLK + threadIdx.x;
    if (rloc >= n) return;
    const int32_t row = (int32_t)rloc;
    const uint32_t m = __builtin_nontemporal_load(rmask + row);
    cplx xv[NS];
#pragma unroll
    for (int c = 0; c < NS; c++) {
        int32_t j = row + sup.off[c];
        j = j < 0 ? 0 : j >= (int32_t)n ? (int32_t)n - 1 : j;
        xv[c] = x[j];
    }
    __builtin_amdgcn_sched_barrier(0);
    cplx sum = make_double2(0., 0.);
#pragma unroll
    for (int c = 0; c < NS; c++)
        { const bool on = m >> c & 1u; const double nx = sum.x + sv.v[c] * xv[c].x, ny = sum.y + sv.v[c] * xv[c].y; sum.x = on ? nx : sum.x; sum.y = on ? ny : sum.y; }
    y[row] = sum;
}
// mask + LDS window for the near offsets
template <int NS, int BLK, unsigned NEAR>
__global__ void __launch_bounds__(BLK) k_mask_tile(int64_t n, int64_t ntiles, Sup sup, SlotVal sv, int32_t H, const uint16_t *__restrict__ rmask,
                                                   const cplx *__restrict__ x, cplx *__restrict__ y) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int64_t tile = xcd_tile(blockIdx.x, ntiles);
    if (tile >= ntiles) return;
    cplx *sx = reinterpret_cast<cplx *>(smem);
    const int64_t base = tile * BLK;
    const int64_t rloc = base + threadIdx.x;
    const bool live = rloc < n;
    const int32_t row = (int32_t)(live ? rloc : n - 1);
    const uint32_t m = __builtin_nontemporal_load(rmask + row);
    cplx xv[NS];
#pragma unroll
    for (int c = 0; c < NS; c++)
        if (!(NEAR >> c & 1u)) {
            int32_t j = row + sup.off[c];
            j = j < 0 ? 0 : j >= (int32_t)n ? (int32_t)n - 1 : j;
            xv[c] = x[j];
        }
    __builtin_amdgcn_sched_barrier(0);
    sx[H + threadIdx.x] = x[row];
    if ((int)threadIdx.x < 2 * H) {
        const int t = threadIdx.x;
        int64_t j = t < H ? base - H + t : base + BLK + (t - H);
        j = j < 0 ? 0 : j >= n ? n - 1 : j;
        sx[t < H ? t : BLK + t] = x[j];
    }
    __syncthreads();
    cplx sum = make_double2(0., 0.);
#pragma unroll
    for (int c = 0; c < NS; c++) {
        cplx v = (NEAR >> c & 1u) ? sx[H + (int)threadIdx.x + sup.off[c]] : xv[c];
        { const bool on = m >> c & 1u; const double nx = sum.x + sv.v[c] * v.x, ny = sum.y + sv.v[c] * v.y; sum.x = on ? nx : sum.x; sum.y = on ? ny : sum.y; }
    }
    if (live) y[row] = sum;
}


// ---- M variants: where does the mask's cost come from? ----
// MV 0: no load (m = all present; wrong on boundaries); 1: temporal u16 load; 2: bit planes, one u64 per wave and slot
// through the scalar cache
template <int NS, int BLK, int MV>
__global__ void __launch_bounds__(BLK) k_mask2(int64_t n, int64_t ntiles, Sup sup, SlotVal sv, const uint16_t *__restrict__ rmask,
                                               const uint64_t *__restrict__ planes, const cplx *__restrict__ x, cplx *__restrict__ y) {
    const int64_t tile = xcd_tile(blockIdx.x, ntiles);
    if (tile >= ntiles) return;
    const int64_t rloc = tile * BLK + threadIdx.x;
    const bool live = rloc < n;
    const int32_t row = (int32_t)(live ? rloc : n - 1);
    uint32_t m = 0x7fu;
    uint64_t pl[NS];
    if (MV == 1) m = rmask[row];
    if (MV == 2) {
        const int64_t wave = __builtin_amdgcn_readfirstlane((int32_t)(rloc >> 6));
        const uint64_t *pp = planes + wave * 8;
#pragma unroll
        for (int c = 0; c < NS; c++) pl[c] = pp[c];
    }
    cplx xv[NS];
#pragma unroll
    for (int c = 0; c < NS; c++) {
        int32_t j = row + sup.off[c];
        j = j < 0 ? 0 : j >= (int32_t)n ? (int32_t)n - 1 : j;
        xv[c] = x[j];
    }
    __builtin_amdgcn_sched_barrier(0);
    const int lane = threadIdx.x & 63;
    cplx sum = make_double2(0., 0.);
#pragma unroll
    for (int c = 0; c < NS; c++) {
        const bool on = MV == 2 ? (pl[c] >> lane & 1ull) != 0 : (m >> c & 1u) != 0;
        const double nx = sum.x + sv.v[c] * xv[c].x, ny = sum.y + sv.v[c] * xv[c].y;
        sum.x = on ? nx : sum.x; sum.y = on ? ny : sum.y;
    }
    if (live) y[row] = sum;
}


// ---- Lb: presence words (scalar loads) + LDS window for the near offsets ----
template <int NS, int BLK, unsigned NEAR>
__global__ void __launch_bounds__(BLK) k_planes_tile(int64_t n, int64_t ntiles, Sup sup, SlotVal sv, int32_t H, const uint64_t *__restrict__ planes,
                                                     const cplx *__restrict__ x, cplx *__restrict__ y) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int64_t tile = xcd_tile(blockIdx.x, ntiles);
    if (tile >= ntiles) return;
    cplx *sx = reinterpret_cast<cplx *>(smem);
    const int64_t base = tile * BLK;
    const int64_t rloc = base + threadIdx.x;
    const bool live = rloc < n;
    const int32_t row = (int32_t)(live ? rloc : n - 1);
    const int64_t wave = __builtin_amdgcn_readfirstlane((int32_t)(rloc >> 6));
    const uint64_t *pp = planes + wave * 8;
    uint64_t pl[NS];
#pragma unroll
    for (int c = 0; c < NS; c++) pl[c] = pp[c];
    cplx xv[NS];
#pragma unroll
    for (int c = 0; c < NS; c++)
        if (!(NEAR >> c & 1u)) {
            int32_t j = row + sup.off[c];
            j = j < 0 ? 0 : j >= (int32_t)n ? (int32_t)n - 1 : j;
            xv[c] = x[j];
        }
    cplx own = x[row];
    cplx halo = make_double2(0., 0.);
    int hidx = -1;
    if ((int)threadIdx.x < 2 * H) {
        const int t = threadIdx.x;
        int64_t j = t < H ? base - H + t : base + BLK + (t - H);
        j = j < 0 ? 0 : j >= n ? n - 1 : j;
        halo = x[j];
        hidx = t < H ? t : BLK + t;
    }
    __builtin_amdgcn_sched_barrier(0);
    sx[H + threadIdx.x] = own;
    if (hidx >= 0) sx[hidx] = halo;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    cplx sum = make_double2(0., 0.);
#pragma unroll
    for (int c = 0; c < NS; c++) {
        cplx v = (NEAR >> c & 1u) ? sx[H + (int)threadIdx.x + sup.off[c]] : xv[c];
        const bool on = (pl[c] >> lane & 1ull) != 0;
        const double nx = sum.x + sv.v[c] * v.x, ny = sum.y + sv.v[c] * v.y;
        sum.x = on ? nx : sum.x; sum.y = on ? ny : sum.y;
    }
    if (live) y[row] = sum;
}

// Lb with a non-temporal store of y (does y's write-back traffic push x's planes out of the 4 MB L2?)
template <int NS, int BLK, unsigned NEAR>
__global__ void __launch_bounds__(BLK) k_planes_tile_nt(int64_t n, int64_t ntiles, Sup sup, SlotVal sv, int32_t H, const uint64_t *__restrict__ planes,
                                                     const cplx *__restrict__ x, cplx *__restrict__ y) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int64_t tile = xcd_tile(blockIdx.x, ntiles);
    if (tile >= ntiles) return;
    cplx *sx = reinterpret_cast<cplx *>(smem);
    const int64_t base = tile * BLK;
    const int64_t rloc = base + threadIdx.x;
    const bool live = rloc < n;
    const int32_t row = (int32_t)(live ? rloc : n - 1);
    const int64_t wave = __builtin_amdgcn_readfirstlane((int32_t)(rloc >> 6));
    const uint64_t *pp = planes + wave * 8;
    uint64_t pl[NS];
#pragma unroll
    for (int c = 0; c < NS; c++) pl[c] = pp[c];
    cplx xv[NS];
#pragma unroll
    for (int c = 0; c < NS; c++)
        if (!(NEAR >> c & 1u)) {
            int32_t j = row + sup.off[c];
            j = j < 0 ? 0 : j >= (int32_t)n ? (int32_t)n - 1 : j;
            xv[c] = x[j];
        }
    cplx own = x[row];
    cplx halo = make_double2(0., 0.);
    int hidx = -1;
    if ((int)threadIdx.x < 2 * H) {
        const int t = threadIdx.x;
        int64_t j = t < H ? base - H + t : base + BLK + (t - H);
        j = j < 0 ? 0 : j >= n ? n - 1 : j;
        halo = x[j];
        hidx = t < H ? t : BLK + t;
    }
    __builtin_amdgcn_sched_barrier(0);
    sx[H + threadIdx.x] = own;
    if (hidx >= 0) sx[hidx] = halo;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    cplx sum = make_double2(0., 0.);
#pragma unroll
    for (int c = 0; c < NS; c++) {
        cplx v = (NEAR >> c & 1u) ? sx[H + (int)threadIdx.x + sup.off[c]] : xv[c];
        const bool on = (pl[c] >> lane & 1ull) != 0;
        const double nx = sum.x + sv.v[c] * v.x, ny = sum.y + sv.v[c] * v.y;
        sum.x = on ? nx : sum.x; sum.y = on ? ny : sum.y;
    }
    if (live) { __builtin_nontemporal_store(sum.x, &y[row].x); __builtin_nontemporal_store(sum.y, &y[row].y); }
}


// Lb with RPT rows per thread: a workgroup of BLK threads owns RPT * BLK consecutive rows (thread t: rows t, t + BLK, ...), one
// window of RPT * BLK + 2 H entries, all loads of all rows in flight before the barrier
template <int NS, int BLK, unsigned NEAR, int RPT>
__global__ void __launch_bounds__(BLK) k_planes_tile_n(int64_t n, int64_t ntiles, Sup sup, SlotVal sv, int32_t H, const uint64_t *__restrict__ planes,
                                                       const cplx *__restrict__ x, cplx *__restrict__ y) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int64_t tile = xcd_tile(blockIdx.x, ntiles);
    if (tile >= ntiles) return;
    cplx *sx = reinterpret_cast<cplx *>(smem);
    const int64_t base = tile * BLK * RPT;
    uint64_t pl[RPT][NS];
    cplx xv[RPT][NS], own[RPT];
    int32_t row[RPT];
    bool live[RPT];
#pragma unroll
    for (int h = 0; h < RPT; h++) {
        const int64_t rloc = base + h * BLK + threadIdx.x;
        live[h] = rloc < n;
        row[h] = (int32_t)(live[h] ? rloc : n - 1);
        const int64_t wave = __builtin_amdgcn_readfirstlane((int32_t)(rloc >> 6));
        const uint64_t *pp = planes + (wave < (n + 63) / 64 ? wave : 0) * 8;
#pragma unroll
        for (int c = 0; c < NS; c++) pl[h][c] = pp[c];
#pragma unroll
        for (int c = 0; c < NS; c++)
            if (!(NEAR >> c & 1u)) {
                int32_t j = row[h] + sup.off[c];
                j = j < 0 ? 0 : j >= (int32_t)n ? (int32_t)n - 1 : j;
                xv[h][c] = x[j];
            }
        own[h] = x[row[h]];
    }
    cplx halo = make_double2(0., 0.);
    int hidx = -1;
    if ((int)threadIdx.x < 2 * H) {
        const int t = threadIdx.x;
        int64_t j = t < H ? base - H + t : base + BLK * RPT + (t - H);
        j = j < 0 ? 0 : j >= n ? n - 1 : j;
        halo = x[j];
        hidx = t < H ? t : BLK * RPT + t;
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int h = 0; h < RPT; h++) sx[H + h * BLK + threadIdx.x] = own[h];
    if (hidx >= 0) sx[hidx] = halo;
    __syncthreads();
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int h = 0; h < RPT; h++) {
        cplx sum = make_double2(0., 0.);
#pragma unroll
        for (int c = 0; c < NS; c++) {
            cplx v = (NEAR >> c & 1u) ? sx[H + h * BLK + (int)threadIdx.x + sup.off[c]] : xv[h][c];
            const bool on = (pl[h][c] >> lane & 1ull) != 0;
            const double nx = sum.x + sv.v[c] * v.x, ny = sum.y + sv.v[c] * v.y;
            sum.x = on ? nx : sum.x; sum.y = on ? ny : sum.y;
        }
        if (live[h]) y[row[h]] = sum;
    }
}

__global__ void k_flush(int64_t n, const double4 *__restrict__ a, double4 *__restrict__ b) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) b[i] = a[i];
}

int main(int argc, char **argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 128;
    const int reps = argc > 2 ? atoi(argv[2]) : 30;
    const int64_t N = (int64_t)n * n * n;
    // patterns
    std::map<std::vector<int32_t>, int> dict;
    std::vector<std::vector<int32_t>> pats;   // offsets with sentinel for absent
    std::vector<uint16_t> h_pid((size_t)N);
    for (int64_t r = 0; r < N; r++) {
        int k = (int)(r % n), j = (int)(r / n % n), i = (int)(r / ((int64_t)n * n));
        std::vector<int32_t> o;
        if (i > 0) o.push_back(-n * n);
        if (j > 0) o.push_back(-n);
        if (k > 0) o.push_back(-1);
        o.push_back(0);
        if (k < n - 1) o.push_back(1);
        if (j < n - 1) o.push_back(n);
        if (i < n - 1) o.push_back(n * n);
        auto it = dict.find(o);
        if (it == dict.end()) { it = dict.emplace(o, (int)pats.size()).first; pats.push_back(o); }
        h_pid[(size_t)r] = (uint16_t)it->second;
    }
    const int npat = (int)pats.size();
    std::vector<int32_t> h_off((size_t)npat * W);
    std::vector<double> h_re((size_t)npat * W);
    for (int p = 0; p < npat; p++)
        for (int c = 0; c < W; c++) {
            const auto &o = pats[(size_t)p];
            if (c < (int)o.size()) { h_off[(size_t)p * W + c] = o[(size_t)c]; h_re[(size_t)p * W + c] = o[(size_t)c] == 0 ? 6. : -1.; }
            else { h_off[(size_t)p * W + c] = o.back(); h_re[(size_t)p * W + c] = 0.; }
        }
    Sup sup;
    std::vector<int32_t> S;
    for (auto &o : pats) for (int32_t v : o) S.push_back(v);
    std::sort(S.begin(), S.end());
    S.erase(std::unique(S.begin(), S.end()), S.end());
    sup.ns = (int)S.size();
    for (int c = 0; c < NSMAX; c++) sup.off[c] = c < sup.ns ? S[(size_t)c] : 0;
    const int NS = 7;
    if (sup.ns != NS) { printf("superset has %d entries\n", sup.ns); return 1; }
    std::vector<double> h_sval((size_t)npat * NS, 0.);
    std::vector<uint32_t> h_smask((size_t)npat, 0);
    for (int p = 0; p < npat; p++)
        for (int32_t o : pats[(size_t)p]) {
            int c = (int)(std::lower_bound(S.begin(), S.end(), o) - S.begin());
            h_sval[(size_t)p * NS + c] = o == 0 ? 6. : -1.;
            h_smask[(size_t)p] |= 1u << c;
        }
    printf("n=%d N=%lld npat=%d\n", n, (long long)N, npat);
    std::vector<cplx> h_x((size_t)N);
    uint64_t s = 12345;
    for (auto &v : h_x) { s = s * 6364136223846793005ull + 1442695040888963407ull; v.x = (double)((s >> 33) % 2000) / 1000. - 1.; s = s * 6364136223846793005ull + 1442695040888963407ull; v.y = (double)((s >> 33) % 2000) / 1000. - 1.; }
    cplx *x, *y, *y0;
    uint16_t *pid; int32_t *poff; double *pre, *sval; uint32_t *smask;
    CK(hipMalloc(&x, sizeof(cplx) * N)); CK(hipMalloc(&y, sizeof(cplx) * N)); CK(hipMalloc(&y0, sizeof(cplx) * N));
    CK(hipMalloc(&pid, 2 * N)); CK(hipMalloc(&poff, 4 * npat * W)); CK(hipMalloc(&pre, 8 * npat * W));
    CK(hipMalloc(&sval, 8 * npat * NS)); CK(hipMalloc(&smask, 4 * npat));
    CK(hipMemcpy(x, h_x.data(), sizeof(cplx) * N, hipMemcpyHostToDevice));
    CK(hipMemcpy(pid, h_pid.data(), 2 * N, hipMemcpyHostToDevice));
    CK(hipMemcpy(poff, h_off.data(), 4 * npat * W, hipMemcpyHostToDevice));
    CK(hipMemcpy(pre, h_re.data(), 8 * npat * W, hipMemcpyHostToDevice));
    CK(hipMemcpy(sval, h_sval.data(), 8 * npat * NS, hipMemcpyHostToDevice));
    CK(hipMemcpy(smask, h_smask.data(), 4 * npat, hipMemcpyHostToDevice));
    const int64_t nflush = (int64_t)512 * 1024 * 1024 / 32;
    double4 *fa, *fb;
    CK(hipMalloc(&fa, 32 * nflush)); CK(hipMalloc(&fb, 32 * nflush));
    CK(hipMemset(fa, 0, 32 * nflush));
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    int ncu = 256;
    { hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, 0)); ncu = pr.multiProcessorCount; printf("CUs=%d\n", ncu); }
    const int64_t nt256 = (N + 255) / 256;
    const unsigned g256 = (unsigned)((nt256 + 7) / 8 * 8);
    const size_t lds_c = (size_t)npat * W * 12, lds_s = (size_t)npat * NS * 8 + (size_t)npat * 4;
    struct Var { const char *name; std::function<void()> run; };
    std::vector<Var> vars;
    vars.push_back({"C  pat_spmv_lds (product)", [&] { hipLaunchKernelGGL(k_pat_lds, dim3(g256), dim3(256), lds_c, st, N, nt256, npat, pid, poff, pre, x, y); }});
    vars.push_back({"A  copy y=x", [&] { hipLaunchKernelGGL(k_copy, dim3((unsigned)nt256), dim3(256), 0, st, N, x, y); }});
    vars.push_back({"A2 copy, 2 per thread", [&] { hipLaunchKernelGGL((k_copy_n<2>), dim3((unsigned)((N + 511) / 512)), dim3(256), 0, st, N, x, y); }});
    vars.push_back({"A4 copy, 4 per thread", [&] { hipLaunchKernelGGL((k_copy_n<4>), dim3((unsigned)((N + 1023) / 1024)), dim3(256), 0, st, N, x, y); }});
    vars.push_back({"A8 copy, 8 per thread", [&] { hipLaunchKernelGGL((k_copy_n<8>), dim3((unsigned)((N + 2047) / 2048)), dim3(256), 0, st, N, x, y); }});
    vars.push_back({"F  fixed stencil (no id/table)", [&] { hipLaunchKernelGGL(k_fixed, dim3(g256), dim3(256), 0, st, N, nt256, sup, x, y); }});
    vars.push_back({"S1 superset, 256 thr, table per wg", [&] { hipLaunchKernelGGL((k_sup<7>), dim3(g256), dim3(256), lds_s, st, N, nt256, npat, sup, pid, sval, smask, x, y); }});
    vars.push_back({"S3 superset, no LDS", [&] { hipLaunchKernelGGL((k_sup_nolds<7>), dim3(g256), dim3(256), 0, st, N, nt256, npat, sup, pid, sval, smask, x, y); }});
    for (int per : {8}) {
        static char nm[8][64]; static int k = 0;
        snprintf(nm[k], 64, "S2 persistent 256 thr x %d/CU", per);
        unsigned g = (unsigned)(ncu * per);
        vars.push_back({nm[k++], [&, g] { hipLaunchKernelGGL((k_sup_persist<7, 256>), dim3(g), dim3(256), lds_s, st, N, nt256, npat, sup, pid, sval, smask, x, y); }});
    }
    for (int per : {4}) {
        static char nm[8][64]; static int k = 0;
        snprintf(nm[k], 64, "S2 persistent 512 thr x %d/CU", per);
        unsigned g = (unsigned)(ncu * per);
        const int64_t nt512 = (N + 511) / 512;
        vars.push_back({nm[k++], [&, g, nt512] { hipLaunchKernelGGL((k_sup_persist<7, 512>), dim3(g), dim3(512), lds_s, st, N, nt512, npat, sup, pid, sval, smask, x, y); }});
    }
    for (int per : {2}) {
        static char nm[8][64]; static int k = 0;
        snprintf(nm[k], 64, "S2 persistent 1024 thr x %d/CU", per);
        unsigned g = (unsigned)(ncu * per);
        const int64_t nt1k = (N + 1023) / 1024;
        vars.push_back({nm[k++], [&, g, nt1k] { hipLaunchKernelGGL((k_sup_persist<7, 1024>), dim3(g), dim3(1024), lds_s, st, N, nt1k, npat, sup, pid, sval, smask, x, y); }});
    }

    vars.push_back({"F1 fixed, 1 load (wrong result)", [&] { hipLaunchKernelGGL((k_fixed_n<1>), dim3(g256), dim3(256), 0, st, N, nt256, sup, x, y); }});
    vars.push_back({"F3 fixed, 3 loads (wrong result)", [&] { hipLaunchKernelGGL((k_fixed_n<3>), dim3(g256), dim3(256), 0, st, N, nt256, sup, x, y); }});
    vars.push_back({"F5 fixed, 5 loads (wrong result)", [&] { hipLaunchKernelGGL((k_fixed_n<5>), dim3(g256), dim3(256), 0, st, N, nt256, sup, x, y); }});
    {
        const int H = n;   // +-1 and +-n from LDS
        const int64_t nt1k = (N + 1023) / 1024, nt512 = (N + 511) / 512;
        const unsigned g1k = (unsigned)((nt1k + 7) / 8 * 8), g512 = (unsigned)((nt512 + 7) / 8 * 8);
        const size_t l1k = (size_t)(1024 + 2 * H) * 16, l512 = (size_t)(512 + 2 * H) * 16;
        if (2 * H <= 512) {
            vars.push_back({"LF tile 1024 + halo n, fixed", [&, H, nt1k, g1k, l1k] { hipLaunchKernelGGL((k_tile<7, 1024, 0x3eu, false>), dim3(g1k), dim3(1024), l1k, st, N, nt1k, npat, sup, H, pid, sval, smask, x, y); }});
            vars.push_back({"LF tile 512 + halo n, fixed", [&, H, nt512, g512, l512] { hipLaunchKernelGGL((k_tile<7, 512, 0x3eu, false>), dim3(g512), dim3(512), l512, st, N, nt512, npat, sup, H, pid, sval, smask, x, y); }});
            vars.push_back({"LP tile 1024 + halo n, patterns", [&, H, nt1k, g1k, l1k] { hipLaunchKernelGGL((k_tile<7, 1024, 0x3eu, true>), dim3(g1k), dim3(1024), l1k + lds_s, st, N, nt1k, npat, sup, H, pid, sval, smask, x, y); }});
            vars.push_back({"LP tile 512 + halo n, patterns", [&, H, nt512, g512, l512] { hipLaunchKernelGGL((k_tile<7, 512, 0x3eu, true>), dim3(g512), dim3(512), l512 + lds_s, st, N, nt512, npat, sup, H, pid, sval, smask, x, y); }});
        }
        const int H1 = 1;  // only +-1 from LDS
        vars.push_back({"LP tile 256 + halo 1, patterns", [&, H1] { hipLaunchKernelGGL((k_tile<7, 256, 0x1cu, true>), dim3(g256), dim3(256), (size_t)(256 + 2) * 16 + lds_s, st, N, nt256, npat, sup, H1, pid, sval, smask, x, y); }});
    }

    {
        std::vector<uint16_t> h_rmask((size_t)N);
        for (int64_t r = 0; r < N; r++) h_rmask[(size_t)r] = (uint16_t)h_smask[h_pid[(size_t)r]];
        static uint16_t *rmask; CK(hipMalloc(&rmask, 2 * N));
        CK(hipMemcpy(rmask, h_rmask.data(), 2 * N, hipMemcpyHostToDevice));
        static SlotVal sv; for (int c = 0; c < NSMAX; c++) sv.v[c] = c < NS ? (S[(size_t)c] == 0 ? 6. : -1.) : 0.;
        const int64_t nt1k = (N + 1023) / 1024, nt512 = (N + 511) / 512;
        const unsigned g1k = (unsigned)((nt1k + 7) / 8 * 8), g512 = (unsigned)((nt512 + 7) / 8 * 8);
        vars.push_back({"M  mask, 256 thr", [&] { hipLaunchKernelGGL((k_mask<7, 256>), dim3(g256), dim3(256), 0, st, N, nt256, sup, sv, rmask, x, y); }});
        vars.push_back({"M  mask, 512 thr", [&, nt512, g512] { hipLaunchKernelGGL((k_mask<7, 512>), dim3(g512), dim3(512), 0, st, N, nt512, sup, sv, rmask, x, y); }});
        vars.push_back({"M  mask, 1024 thr", [&, nt1k, g1k] { hipLaunchKernelGGL((k_mask<7, 1024>), dim3(g1k), dim3(1024), 0, st, N, nt1k, sup, sv, rmask, x, y); }});

        {
            const int64_t nw = (N + 63) / 64;
            std::vector<uint64_t> h_pl((size_t)nw * 8, 0);
            for (int64_t r = 0; r < N; r++)
                for (int c = 0; c < NS; c++)
                    if (h_rmask[(size_t)r] >> c & 1) h_pl[(size_t)(r >> 6) * 8 + c] |= 1ull << (r & 63);
            static uint64_t *planes; CK(hipMalloc(&planes, 8 * h_pl.size()));
            CK(hipMemcpy(planes, h_pl.data(), 8 * h_pl.size(), hipMemcpyHostToDevice));
            vars.push_back({"M0 no mask load (wrong), 256", [&] { hipLaunchKernelGGL((k_mask2<7, 256, 0>), dim3(g256), dim3(256), 0, st, N, nt256, sup, sv, rmask, planes, x, y); }});
            vars.push_back({"Mt temporal u16 mask, 256", [&] { hipLaunchKernelGGL((k_mask2<7, 256, 1>), dim3(g256), dim3(256), 0, st, N, nt256, sup, sv, rmask, planes, x, y); }});
            vars.push_back({"Mb bit planes (scalar loads), 256", [&] { hipLaunchKernelGGL((k_mask2<7, 256, 2>), dim3(g256), dim3(256), 0, st, N, nt256, sup, sv, rmask, planes, x, y); }});
            vars.push_back({"Mb bit planes (scalar loads), 512", [&, nt512, g512] { hipLaunchKernelGGL((k_mask2<7, 512, 2>), dim3(g512), dim3(512), 0, st, N, nt512, sup, sv, rmask, planes, x, y); }});
            vars.push_back({"Mb bit planes (scalar loads), 1024", [&, nt1k, g1k] { hipLaunchKernelGGL((k_mask2<7, 1024, 2>), dim3(g1k), dim3(1024), 0, st, N, nt1k, sup, sv, rmask, planes, x, y); }});
            {
                const int H = n;
                if (2 * H <= 512) {
                    vars.push_back({"Lb planes + tile 1024 halo n", [&, H, nt1k, g1k] { hipLaunchKernelGGL((k_planes_tile<7, 1024, 0x3eu>), dim3(g1k), dim3(1024), (size_t)(1024 + 2 * H) * 16, st, N, nt1k, sup, sv, H, planes, x, y); }});
                    vars.push_back({"Lb planes + tile 512 halo n", [&, H, nt512, g512] { hipLaunchKernelGGL((k_planes_tile<7, 512, 0x3eu>), dim3(g512), dim3(512), (size_t)(512 + 2 * H) * 16, st, N, nt512, sup, sv, H, planes, x, y); }});
                    {
                        const int64_t nt2 = (N + 1023) / 1024, nt4 = (N + 2047) / 2048;
                        const unsigned g2 = (unsigned)((nt2 + 7) / 8 * 8), g4 = (unsigned)((nt4 + 7) / 8 * 8);
                        vars.push_back({"Lb planes + tile, 512 thr x 2 rows", [&, H, nt2, g2] { hipLaunchKernelGGL((k_planes_tile_n<7, 512, 0x3eu, 2>), dim3(g2), dim3(512), (size_t)(1024 + 2 * H) * 16, st, N, nt2, sup, sv, H, planes, x, y); }});
                        vars.push_back({"Lb planes + tile, 256 thr x 2 rows", [&, H, nt512, g512] { hipLaunchKernelGGL((k_planes_tile_n<7, 256, 0x3eu, 2>), dim3(g512), dim3(256), (size_t)(512 + 2 * H) * 16, st, N, nt512, sup, sv, H, planes, x, y); }});
                        vars.push_back({"Lb planes + tile, 512 thr x 4 rows", [&, H, nt4, g4] { hipLaunchKernelGGL((k_planes_tile_n<7, 512, 0x3eu, 4>), dim3(g4), dim3(512), (size_t)(2048 + 2 * H) * 16, st, N, nt4, sup, sv, H, planes, x, y); }});
                        vars.push_back({"Lb planes + tile, 256 thr x 4 rows", [&, H, nt2, g2] { hipLaunchKernelGGL((k_planes_tile_n<7, 256, 0x3eu, 4>), dim3(g2), dim3(256), (size_t)(1024 + 2 * H) * 16, st, N, nt2, sup, sv, H, planes, x, y); }});
                    }
                    vars.push_back({"Lb planes + tile 512, nontemporal y", [&, H, nt512, g512] { hipLaunchKernelGGL((k_planes_tile_nt<7, 512, 0x3eu>), dim3(g512), dim3(512), (size_t)(512 + 2 * H) * 16, st, N, nt512, sup, sv, H, planes, x, y); }});
                    vars.push_back({"Lb planes + tile 1024 halo n (+-n only)", [&, H, nt1k, g1k] { hipLaunchKernelGGL((k_planes_tile<7, 1024, 0x22u>), dim3(g1k), dim3(1024), (size_t)(1024 + 2 * H) * 16, st, N, nt1k, sup, sv, H, planes, x, y); }});
                }
            }

        }
        const int H = n;
        if (2 * H <= 512) {
            vars.push_back({"LM mask + tile 1024 halo n", [&, H, nt1k, g1k] { hipLaunchKernelGGL((k_mask_tile<7, 1024, 0x3eu>), dim3(g1k), dim3(1024), (size_t)(1024 + 2 * H) * 16, st, N, nt1k, sup, sv, H, rmask, x, y); }});
            vars.push_back({"LM mask + tile 512 halo n", [&, H, nt512, g512] { hipLaunchKernelGGL((k_mask_tile<7, 512, 0x3eu>), dim3(g512), dim3(512), (size_t)(512 + 2 * H) * 16, st, N, nt512, sup, sv, H, rmask, x, y); }});
            vars.push_back({"LM mask + tile 1024 halo n, +-n only", [&, H, nt1k, g1k] { hipLaunchKernelGGL((k_mask_tile<7, 1024, 0x22u>), dim3(g1k), dim3(1024), (size_t)(1024 + 2 * H) * 16, st, N, nt1k, sup, sv, H, rmask, x, y); }});
        }
    }
    const double bytes = 2. * N + 32. * N;
    bool first = true;
    for (auto &v : vars) {
        CK(hipMemsetAsync(y, 0xff, sizeof(cplx) * N, st));
        v.run();
        CK(hipStreamSynchronize(st));
        CK(hipGetLastError());
        bool same = true;
        if (first) { CK(hipMemcpy(y0, y, sizeof(cplx) * N, hipMemcpyDeviceToDevice)); first = false; }
        else if (v.name[0] != 'A') {
            std::vector<cplx> a((size_t)N), b((size_t)N);
            CK(hipMemcpy(a.data(), y, sizeof(cplx) * N, hipMemcpyDeviceToHost));
            CK(hipMemcpy(b.data(), y0, sizeof(cplx) * N, hipMemcpyDeviceToHost));
            same = memcmp(a.data(), b.data(), sizeof(cplx) * N) == 0;
        }
        // warm: back to back
        CK(hipEventRecord(e0, st));
        for (int i = 0; i < reps; i++) v.run();
        CK(hipEventRecord(e1, st));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const double warm = ms * 1e3 / reps;
        // cold: sweep the caches between launches
        double cold = 0., cmin = 1e30;
        for (int i = 0; i < reps; i++) {
            hipLaunchKernelGGL(k_flush, dim3(2048), dim3(256), 0, st, nflush, fa, fb);
            CK(hipEventRecord(e0, st));
            v.run();
            CK(hipEventRecord(e1, st));
            CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms, e0, e1));
            cold += ms * 1e3; cmin = std::min(cmin, (double)ms * 1e3);
        }
        cold /= reps;
        printf("%-40s warm %7.2f us (%5.0f GB/s)  cold avg %7.2f min %7.2f us (%5.0f GB/s, %.2f of 8 TB/s)  %s\n", v.name, warm, bytes / warm / 1e3,
               cold, cmin, bytes / cold / 1e3, bytes / cold / 1e3 / 8000., same ? "bits==C" : "DIFFERENT");
    }
    return 0;
}
