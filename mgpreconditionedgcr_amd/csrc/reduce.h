// Deterministic wave64 / workgroup reductions and complex helpers (device side, gfx950).
#pragma once
#include "internal.h"

namespace mgcr {

__device__ __forceinline__ cplx cmul(cplx a, cplx b) {
    // (a.x + i a.y)(b.x + i b.y), the un-fused order of the reference's std::complex operator*
    return make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
__device__ __forceinline__ cplx cconj_mul(cplx a, cplx b) {
    // conj(a) * b  (src/Fields.h:222): (a.x - i a.y)(b.x + i b.y)
    return make_double2(a.x * b.x + a.y * b.y, a.x * b.y - a.y * b.x);
}
__device__ __forceinline__ cplx cadd(cplx a, cplx b) { return make_double2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ cplx csub(cplx a, cplx b) { return make_double2(a.x - b.x, a.y - b.y); }
// complex division with libgcc's __divdc3 operation order (Smith), which is what the reference's
// `alpha = r.dot(Ap) / Ap.dot(Ap)` (src/GCR.h:230,258) lowers to
__device__ __forceinline__ cplx cdiv(cplx n, cplx d) {
    double a = n.x, b = n.y, c = d.x, e = d.y;
    if (fabs(c) < fabs(e)) {
        double ratio = c / e, denom = (c * ratio) + e;
        return make_double2(((a * ratio) + b) / denom, ((b * ratio) - a) / denom);
    } else {
        double ratio = e / c, denom = (e * ratio) + c;
        return make_double2(((b * ratio) + a) / denom, (b - (a * ratio)) / denom);
    }
}

// Non-temporal access to streams that are touched once per solver iteration (old direction slots):
// they then do not displace the vectors that ARE re-used within the iteration (r, Ar, the newest Ap)
// from L2 / the 256 MiB Infinity Cache.
template <bool NT>
__device__ __forceinline__ cplx ld_stream(const cplx *p) {
    if (NT) return make_double2(__builtin_nontemporal_load(&p->x), __builtin_nontemporal_load(&p->y));
    return *p;
}
template <bool NT>
__device__ __forceinline__ void st_stream(cplx *p, cplx v) {
    if (NT) {
        __builtin_nontemporal_store(v.x, &p->x);
        __builtin_nontemporal_store(v.y, &p->y);
    } else {
        *p = v;
    }
}

// Re-materialise a wave-uniform double in scalar registers (frees 2 VGPRs per value)
__device__ __forceinline__ double to_sgpr(double v) {
    int lo = __builtin_amdgcn_readfirstlane(__double2loint(v));
    int hi = __builtin_amdgcn_readfirstlane(__double2hiint(v));
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ cplx to_sgpr(cplx v) { return make_double2(to_sgpr(v.x), to_sgpr(v.y)); }

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_down(v, off, 64);
    return v;  // valid in lane 0
}

// NV wave sums at once.  Each scalar is summed over the 64 lanes by the very tree wave_sum builds (pairs (l, l + 32),
// then (l, l + 16) of those sums, ...: same operands, same order, same bits), but the NV trees share their shuffles:
// at offset 32 the lower half-wave keeps the first half of the scalars and hands the second half to the upper
// half-wave (which keeps those and hands over the first half), at offset 16 the halves split again, ... until every
// lane carries one scalar; the remaining offsets are a plain butterfly.  NV = 16: 17 shuffles of a double instead
// of 96 (they go through the LDS permute unit, which all 16 waves of a workgroup share: the folds and block
// reductions of 10-20 scalars were a visible part of every solver kernel in the latency regime).
// On return lane `l` holds in v[0] the total of scalar wave_multi_owner<NV>(l) (several lanes hold each scalar).
template <int NV>
struct WaveMulti {
    static constexpr int NVP = NV <= 1 ? 1 : NV <= 2 ? 2 : NV <= 4 ? 4 : NV <= 8 ? 8 : NV <= 16 ? 16 : NV <= 32 ? 32 : 64;
    static_assert(NV <= 64, "at most 64 scalars per multi-sum");
};
template <int NV>
__device__ __forceinline__ int wave_multi_sum(double (&v)[NV], double &out) {
    constexpr int NVP = WaveMulti<NV>::NVP;
    const int lane = threadIdx.x & 63;
    double w[NVP];
#pragma unroll
    for (int j = 0; j < NVP; j++) w[j] = j < NV ? v[j] : 0.;
    int base = 0;
    int c = NVP;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const bool upper = (lane & off) != 0;
        if (c > 1) {
            const int h = c / 2;
#pragma unroll
            for (int j = 0; j < NVP / 2; j++) {
                if (j < h) {
                    const double lo = w[j], hi = w[j + h];
                    const double keep = upper ? hi : lo;
                    const double recv = __shfl_xor(upper ? lo : hi, off, 64);
                    w[j] = upper ? (recv + keep) : (keep + recv);   // always (lower lane's value) + (upper lane's value)
                }
            }
            if (upper) base += h;
            c = h;
        } else {
            const double recv = __shfl_xor(w[0], off, 64);
            w[0] = upper ? (recv + w[0]) : (w[0] + recv);
        }
    }
    out = w[0];
    return base;
}

// Sum NV per-thread doubles over the workgroup in a fixed order (lanes by shuffle tree, waves in
// index order) and broadcast the totals to every thread.  `lds` needs NV * 17 doubles.
template <int NV>
__device__ __forceinline__ void block_sum_bcast(double (&v)[NV], double *lds) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwave = (blockDim.x + 63) >> 6;
    constexpr int NVP = WaveMulti<NV>::NVP;
    double s;
    const int k = wave_multi_sum<NV>(v, s);
    // scalar k sits in the 64 / NVP lanes whose upper bits spell k: the one with the low bits clear stores it
    if ((lane & (64 / NVP - 1)) == 0 && k < NV) lds[k * 17 + wave] = s;
    __syncthreads();
    if (threadIdx.x < NV) {
        double t = 0.;
        for (int w = 0; w < nwave; w++) t += lds[threadIdx.x * 17 + w];
        lds[threadIdx.x * 17 + 16] = t;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < NV; j++) v[j] = lds[j * 17 + 16];
    __syncthreads();
}

// Same sums (same order, same bits) for a kernel tail that only WRITES them out: thread t < NV returns the total
// of scalar t, the other threads return 0 — no broadcast back to the workgroup (NV LDS reads per thread and two
// barriers less).  `lds` must not be reused before another barrier.
template <int NV>
__device__ __forceinline__ double block_sum_owner(double (&v)[NV], double *lds) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwave = (blockDim.x + 63) >> 6;
    constexpr int NVP = WaveMulti<NV>::NVP;
    double s;
    const int k = wave_multi_sum<NV>(v, s);
    if ((lane & (64 / NVP - 1)) == 0 && k < NV) lds[k * 17 + wave] = s;
    __syncthreads();
    double t = 0.;
    if (threadIdx.x < NV)
        for (int w = 0; w < nwave; w++) t += lds[threadIdx.x * 17 + w];
    return t;
}

// One wave folds one scalar's slab of per-workgroup partials (multi-GPU fold kernels): the 16 wave-sized pieces are
// summed by the shuffle tree and added in index order — the tree fold_partials / block_sum_bcast build over a
// 1024-thread workgroup, so the result has the bits of the in-kernel folds.  The 16 piece sums share their
// shuffles (wave_multi_sum); piece k ends up in lanes 4k..4k+3.  Returns the total in every lane.
__device__ __forceinline__ double wave_fold_slab(const double *__restrict__ src, int nblk) {
    static_assert(RED_THREADS == 1024, "16 wave-sized pieces");
    const int lane = threadIdx.x & 63;
    double v[16];
#pragma unroll
    for (int w = 0; w < 16; w++) {
        const int t = w * 64 + lane;
        v[w] = t < nblk ? src[t] : 0.;
    }
    double s;
    wave_multi_sum<16>(v, s);
    double acc = 0.;
#pragma unroll
    for (int k = 0; k < 16; k++) acc += __shfl(s, 4 * k, 64);
    return acc;
}

// Fold a partial slab parts[k * stride + blk] (nblk valid entries per scalar) to NV totals,
// identically in every workgroup that calls it (blockDim.x must be >= RED_MAX_BLOCKS).
// stride = RED_MAX_BLOCKS for a slab written by a producer kernel; stride = 1, nblk = 1 for scalars
// that were already folded and all-reduced over the ranks (multi-GPU).
template <int NV>
__device__ __forceinline__ void fold_partials(const double *__restrict__ parts, int nblk, int stride, double (&v)[NV], double *lds) {
    if (nblk == 1) {
        // already folded (multi-GPU): the tree would add zeros to the one value — x + 0.0 is x, and turns -0.0 into the
        // +0.0 the tree yields — so every thread can simply load it: no shuffles, no barriers
#pragma unroll
        for (int k = 0; k < NV; k++) v[k] = parts[(size_t)k * stride] + 0.;
        return;
    }
#pragma unroll
    for (int k = 0; k < NV; k++) v[k] = ((int)threadIdx.x < nblk) ? parts[(size_t)k * stride + threadIdx.x] : 0.;
    block_sum_bcast<NV>(v, lds);
}

}  // namespace mgcr
