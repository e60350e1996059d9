// Device-wide exchanges of a few scalars among the workgroups of ONE launch, without fences (gcr_resident.hip: the
// one-launch solver for small systems; gcr_stepbuild.hip: apply + dots + direction build as one launch for systems whose
// A r fits the chip's LDS).  See gcr_resident.hip's header for the measurements behind the design.
//   * data that crosses workgroups is written buffer_store ... sc1 and read buffer_load ... sc1 (coherent at the memory
//     side, no release / acquire fence: tools/coherent_lab.hip);
//   * a workgroup's partial sums are 16-byte {value, generation} slots; they are folded in two hops — (64 workgroups x
//     scalar) tasks dealt over waves, whose group sums every workgroup polls from one of 16 replicas — in the order and
//     with the tree of reduce.h's fold_partials: the totals have the bits of the per-workgroup partial slabs folded by
//     a consumer kernel;
//   * every poll is bounded: a workgroup that does not show up makes the others give up (abort flag), never spin.
#pragma once
#include "internal.h"
#include "reduce.h"

namespace mgcr {

constexpr int RES_NV = 24;          // scalars per exchange kind (<= 2 * 10 beta numerators, 5 at step 0)
constexpr int RES_BLK = 512;        // workgroups at most (two per CU)
constexpr int RES_GRP = RES_BLK / 64;   // groups of 64 workgroups
constexpr int RES_CHUNK = 8;        // scalars folded at a time (register footprint of the shuffle tree)
constexpr int RES_SPIN_LIMIT = 1 << 22;
constexpr int RES_RPT_DEFAULT = 2;  // rows per thread: 64^3 GCR(10) 18.7 / 16.0 / 19.8 us per iteration with 1 / 2 / 4 (MGCR_RESIDENT_RPT in -DMGCR_RES_ALL_RPT builds)
constexpr int RES_COPIES = 16;      // copies of every group sum (workgroup b reads copy b % 16)
constexpr int RES_L1_BYTES = 3 * RES_NV * RES_BLK * 16;            // {value, generation} per (kind, scalar, workgroup)
constexpr int RES_L2_BASE = RES_L1_BYTES;                          // then [copy][kind][scalar][group of 64 workgroups]
constexpr int RES_SLOT_BYTES = RES_L1_BYTES + RES_COPIES * 3 * RES_NV * RES_GRP * 16;

typedef int v4i __attribute__((ext_vector_type(4)));

__device__ __forceinline__ __amdgpu_buffer_rsrc_t res_rsrc(const void *p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, (int)bytes, 0x00020000);
}
constexpr int RES_SC1 = 16;   // cache-policy operand of the buffer intrinsics on gfx94x / gfx950: sc1
// row idx of the vector that starts soff bytes into the buffer (soff wave-uniform)
__device__ __forceinline__ cplx ld_coh(__amdgpu_buffer_rsrc_t r, int idx, int soff) {
    const v4i w = __builtin_amdgcn_raw_buffer_load_b128(r, idx * 16, soff, RES_SC1);
    return make_double2(__hiloint2double(w.y, w.x), __hiloint2double(w.w, w.z));
}
__device__ __forceinline__ void st_coh(__amdgpu_buffer_rsrc_t r, int idx, int soff, cplx v) {
    const v4i w = {__double2loint(v.x), __double2hiint(v.x), __double2loint(v.y), __double2hiint(v.y)};
    __builtin_amdgcn_raw_buffer_store_b128(w, r, idx * 16, soff, RES_SC1);
}

struct ResSync {
    __amdgpu_buffer_rsrc_t slots;
    unsigned gen;        // generation of the NEXT exchange
    int nblk, lb;
    unsigned *abort_dev;
    int spin_limit;
    double *pw;          // [RES_NV][17] this workgroup's wave sums of the exchange being posted
    double *ws;          // [RES_NV][RES_GRP]  sums over 64 workgroups each of the exchange being collected
    int *gave_up;        // LDS flag
};

// An exchange = contribute (every wave: its 64 rows' terms -> LDS), publish (one barrier; thread k adds the 16 wave
// sums of scalar k in wave order — reduce.h block_sum_owner — and stores {sum, generation} into this workgroup's
// slot), collect (every wave polls a share of the slots — 64 workgroups x a few scalars — and sums them with the
// shuffle tree of reduce.h fold_partials; one barrier), total (any thread: the <= 4 group sums in order).  Same
// operands, same order, same bits as the per-workgroup partial slabs of gcr.hip's kernels; 2 barriers per exchange.
// vw: which 64 rows of the workgroup's 1024 these lanes hold (a thread owns RPT rows, 1024 / RPT apart: wave w holds the
// row sets w, w + nwaves, ...) — the slot a 1024-thread workgroup's wave vw would write
template <int NV>
__device__ __forceinline__ void res_contrib(ResSync &s, int k0, double (&v)[NV], int vw) {
    static_assert(NV <= RES_CHUNK, "chunk");
    constexpr int NVP = WaveMulti<NV>::NVP;
    const int lane = threadIdx.x & 63;
    double t;
    const int k = wave_multi_sum<NV>(v, t);
    if ((lane & (64 / NVP - 1)) == 0 && k < NV) s.pw[(k0 + k) * 17 + vw] = t;
}
template <int NVT>
__device__ __forceinline__ void res_publish(ResSync &s, int kind) {
    __syncthreads();
    if ((int)threadIdx.x < NVT) {
        double t = 0.;
#pragma unroll
        for (int w = 0; w < RED_THREADS / 64; w++) t += s.pw[threadIdx.x * 17 + w];   // 16 sets of 64 rows, whatever the thread count
        const v4i w4 = {__double2loint(t), __double2hiint(t), (int)s.gen, 0};
        __builtin_amdgcn_raw_buffer_store_b128(w4, s.slots, ((kind * RES_NV + (int)threadIdx.x) * RES_BLK + s.lb) * 16, 0, RES_SC1);
    }
}
// Collect in two hops, so that no slot is read by more than a few workgroups at a time (256 workgroups polling the same
// 4 KB took 5-6 us per exchange: one memory channel serves it all).  Hop 1: the (group g of 64 workgroups, scalar k) sums
// are TASKS dealt over the waves of the first workgroups; a task's wave polls the 64 slots, sums them with the shuffle
// tree of reduce.h fold_partials and stores the group sum, RES_COPIES times (4.5 KB apart: other channels).  Hop 2:
// thread (k, g) of every workgroup polls its copy of group sum (k, g) into LDS.  false: somebody did not show up in
// time (abort).
// EXTRA: scalar NVT of this exchange is scalar 0 of the kind-0 exchange published one generation earlier and not collected
// then (the |r|^2 partials, whose publication only served as the neighbours' signal: res_neighbour_wait).
template <int NVT, bool EXTRA = false>
__device__ __forceinline__ bool res_collect(ResSync &s, int kind) {
    constexpr int NS = NVT + (EXTRA ? 1 : 0);
    static_assert(NS <= RES_NV, "scalars per exchange");
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ng = (s.nblk + 63) >> 6;   // groups that hold workgroups
    for (int task = wave * s.nblk + s.lb; task < ng * NS; task += ((int)blockDim.x >> 6) * s.nblk) {   // wave-uniform
        const int g = task / NS, k = task - g * NS;
        const int blk = g * 64 + lane;
        const bool extra = EXTRA && k == NVT;
        const int src = extra ? blk : (kind * RES_NV + k) * RES_BLK + blk;
        const unsigned want = extra ? s.gen - 1u : s.gen;
        double v = 0.;
        if (blk < s.nblk) {
            int spins = 0;
            for (;;) {
                const v4i w = __builtin_amdgcn_raw_buffer_load_b128(s.slots, src * 16, 0, RES_SC1);
                v = __hiloint2double(w.y, w.x);
                if ((unsigned)w.z == want) break;
                spins++;
                if (spins > s.spin_limit || ((spins & 255) == 0 && __hip_atomic_load(s.abort_dev, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u)) {
                    *s.gave_up = 1;
                    v = 0.;
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
            }
        }
        double t = wave_sum(v);
        t = __shfl(t, 0, 64);
        if (lane < RES_COPIES) {
            const v4i w4 = {__double2loint(t), __double2hiint(t), (int)s.gen, 0};
            __builtin_amdgcn_raw_buffer_store_b128(w4, s.slots, RES_L2_BASE + ((lane * 3 + kind) * RES_NV + k) * RES_GRP * 16 + g * 16, 0, RES_SC1);
        }
    }
    if ((int)threadIdx.x < RES_GRP * NS) {
        const int k = (int)threadIdx.x / RES_GRP, g = (int)threadIdx.x % RES_GRP;
        double v = 0.;
        if (g < ng) {
            const int copy = s.lb & (RES_COPIES - 1);
            int spins = 0;
            for (;;) {
                const v4i w = __builtin_amdgcn_raw_buffer_load_b128(s.slots, RES_L2_BASE + ((copy * 3 + kind) * RES_NV + k) * RES_GRP * 16 + g * 16, 0, RES_SC1);
                v = __hiloint2double(w.y, w.x);
                if ((unsigned)w.z == s.gen) break;
                spins++;
                if (spins > s.spin_limit || ((spins & 255) == 0 && __hip_atomic_load(s.abort_dev, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u)) {
                    *s.gave_up = 1;
                    v = 0.;
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
            }
        }
        s.ws[k * RES_GRP + g] = v;
    }
    __syncthreads();
    s.gen++;
    return *s.gave_up == 0;
}
// The residual hand-over needs no device-wide rendezvous: a workgroup gathers rows of the workgroups lb - nbr .. lb + nbr
// only, and each of those publishes its |r|^2 slot (kind 0, scalar 0) after its residual rows have reached memory.  Wave 0
// polls those slots (one hop), the totals are folded later (res_collect<.., true>).  A slot may already carry a later
// generation only in theory (see the file header); >= keeps the wait finite even then.
__device__ __forceinline__ bool res_neighbour_wait(ResSync &s, int nbr) {
    if (threadIdx.x < 64) {
        for (int b = s.lb - nbr + (int)threadIdx.x; b <= s.lb + nbr; b += 64) {
            if (b < 0 || b >= s.nblk || b == s.lb) continue;
            int spins = 0;
            for (;;) {
                const v4i w = __builtin_amdgcn_raw_buffer_load_b128(s.slots, b * 16, 0, RES_SC1);
                if ((int)((unsigned)w.z - s.gen) >= 0) break;
                spins++;
                if (spins > s.spin_limit || ((spins & 255) == 0 && __hip_atomic_load(s.abort_dev, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u)) {
                    *s.gave_up = 1;
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
            }
        }
    }
    __syncthreads();
    s.gen++;
    return *s.gave_up == 0;
}

// sum over all workgroups of scalar k of the exchange collected last (reduce.h block_sum_bcast: waves in index order from 0.;
// the waves past the fourth hold no workgroup and would add + 0.0)
__device__ __forceinline__ double res_total(const ResSync &s, int k) {
    double t = 0.;
#pragma unroll
    for (int g = 0; g < RES_GRP; g++) t += s.ws[k * RES_GRP + g];
    return t;
}


// the process-wide slots, generation counter and abort flags (gcr_resident.hip)
struct ExchangeShared {
    v4i *slots = nullptr;
    unsigned *abort_dev = nullptr;
    int *abort_host = nullptr;   // host-mapped
    unsigned gen = 1;
    int cus = 0;
};
int exchange_shared_init();
ExchangeShared &exchange_shared();
unsigned exchange_take_generations(unsigned need);   // first of `need` fresh generations

}  // namespace mgcr
