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

// Lane exchange l <-> l ^ OFF inside a wave64 WITHOUT the LDS permute unit (ds_bpermute_b32, what __shfl_xor compiles to:
// ~100 cycles of latency per level of a reduction tree, and all 16 waves of a workgroup share the unit): DPP moves for
// OFF = 1, 2, 4, 8 and gfx950's v_permlane16_swap / v_permlane32_swap for 16 and 32 (tools/xor_lane_lab.hip checks them).
template <int OFF>
__device__ __forceinline__ int xor_lane_b32(int v) {
    static_assert(OFF == 1 || OFF == 2 || OFF == 4 || OFF == 8, "DPP reaches inside a row of 16 lanes");
    if constexpr (OFF == 1) return __builtin_amdgcn_mov_dpp(v, 0xB1, 0xf, 0xf, false);        // quad_perm [1,0,3,2]
    else if constexpr (OFF == 2) return __builtin_amdgcn_mov_dpp(v, 0x4E, 0xf, 0xf, false);   // quad_perm [2,3,0,1]
    else if constexpr (OFF == 4) {
        const int t = __builtin_amdgcn_mov_dpp(v, 0x141, 0xf, 0xf, false);                    // row_half_mirror: l -> l ^ 7
        return __builtin_amdgcn_mov_dpp(t, 0x1B, 0xf, 0xf, false);                            // quad_perm [3,2,1,0]: l -> l ^ 3
    } else return __builtin_amdgcn_mov_dpp(v, 0x128, 0xf, 0xf, false);                        // row_ror:8
}
// One level of a reduction tree over the lane pairs (l, l ^ OFF): every lane hands in the value `lo` it has for the scalar
// the LOWER lane of its pair keeps and `hi` for the scalar the UPPER lane keeps; it gets back, for the scalar it keeps,
// (lower lane's value) + (upper lane's value) — operands and order of wave_sum's tree.  lo == hi: a plain butterfly level.
// OFF = 32 / 16: v_permlane{32,16}_swap exchanges the upper half (odd rows) of its first operand with the lower half (even
// rows) of its second one, which IS "the lower lane keeps lo and receives the upper lane's lo, the upper lane keeps hi and
// receives the lower lane's hi": two swaps and one add, no select.
template <int OFF>
__device__ __forceinline__ double tree_level(double lo, double hi) {
    if constexpr (OFF == 32 || OFF == 16) {
        const unsigned ll = (unsigned)__double2loint(lo), lh = (unsigned)__double2hiint(lo);
        const unsigned hl = (unsigned)__double2loint(hi), hh = (unsigned)__double2hiint(hi);
        if constexpr (OFF == 32) {
            const auto a = __builtin_amdgcn_permlane32_swap(ll, hl, false, false);
            const auto b = __builtin_amdgcn_permlane32_swap(lh, hh, false, false);
            return __hiloint2double((int)b[0], (int)a[0]) + __hiloint2double((int)b[1], (int)a[1]);
        } else {
            const auto a = __builtin_amdgcn_permlane16_swap(ll, hl, false, false);
            const auto b = __builtin_amdgcn_permlane16_swap(lh, hh, false, false);
            return __hiloint2double((int)b[0], (int)a[0]) + __hiloint2double((int)b[1], (int)a[1]);
        }
    } else {
        const bool upper = ((int)threadIdx.x & OFF) != 0;
        const double keep = upper ? hi : lo, send = upper ? lo : hi;
        const double recv = __hiloint2double(xor_lane_b32<OFF>(__double2hiint(send)), xor_lane_b32<OFF>(__double2loint(send)));
        return upper ? (recv + keep) : (keep + recv);   // always (lower lane's value) + (upper lane's value)
    }
}

// pairs (l, l + 32), then (l, l + 16) of those sums, ... : valid in lane 0 (every lane whose bits below the level are clear)
__device__ __forceinline__ double wave_sum(double v) {
    v = tree_level<32>(v, v);
    v = tree_level<16>(v, v);
    v = tree_level<8>(v, v);
    v = tree_level<4>(v, v);
    v = tree_level<2>(v, v);
    v = tree_level<1>(v, v);
    return v;  // valid in lane 0
}

// NV wave sums at once.  Each scalar is summed over the 64 lanes by the very tree wave_sum builds (pairs (l, l + 32),
// then (l, l + 16) of those sums, ...: same operands, same order, same bits), but the NV trees share their exchanges:
// at offset 32 the lower half-wave keeps the first half of the scalars and hands the second half to the upper
// half-wave (which keeps those and hands over the first half), at offset 16 the halves split again, ... until every
// lane carries one scalar; the remaining offsets are a plain butterfly.  NV = 16: 17 exchanges of a double instead
// of 96.
// On return lane `l` holds in v[0] the total of scalar wave_multi_owner<NV>(l) (several lanes hold each scalar).
template <int NV>
struct WaveMulti {
    static constexpr int NVP = NV <= 1 ? 1 : NV <= 2 ? 2 : NV <= 4 ? 4 : NV <= 8 ? 8 : NV <= 16 ? 16 : NV <= 32 ? 32 : 64;
    static_assert(NV <= 64, "at most 64 scalars per multi-sum");
};
template <int C, int OFF, int NVP>
__device__ __forceinline__ void wave_multi_level(double (&w)[NVP], int &base) {
    // C scalars are still carried by every lane (C is a power of two)
    if constexpr (C > 1) {
        constexpr int h = C / 2;
#pragma unroll
        for (int j = 0; j < h; j++) w[j] = tree_level<OFF>(w[j], w[j + h]);
        if (((int)threadIdx.x & OFF) != 0) base += h;
    } else {
        w[0] = tree_level<OFF>(w[0], w[0]);
    }
    if constexpr (OFF > 1) wave_multi_level<(C > 1 ? C / 2 : 1), OFF / 2, NVP>(w, base);
}
template <int NV>
__device__ __forceinline__ int wave_multi_sum(double (&v)[NV], double &out) {
    constexpr int NVP = WaveMulti<NV>::NVP;
    double w[NVP];
#pragma unroll
    for (int j = 0; j < NVP; j++) w[j] = j < NV ? v[j] : 0.;
    int base = 0;
    wave_multi_level<NVP, 32, NVP>(w, base);
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
