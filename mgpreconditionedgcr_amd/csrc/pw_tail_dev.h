// Multi-GPU: the fold of a producer kernel's per-workgroup partials AND their sum over the ranks, done by the LAST workgroup of
// the producing launch itself — instead of a fold + exchange launch of its own behind it (comm.hip fold_pw_kernel, whose body
// this is).  Every workgroup writes its partials as always, makes them visible (release fence), takes a ticket; the one that
// draws the last ticket folds each scalar's slab with reduce.h's wave_fold_slab (the tree of fold_partials: the bits of the
// single-GPU path), stores the sum into every peer's mailbox, polls its own mailbox — bounded — for theirs and adds the values
// in rank order (identical bits on every rank).  The consumers read the global scalars as they do after fold_pw_kernel.
// An iteration of a distributed solve is then xr_update | halo | apply + dots (+ fold + exchange) | build (+ fold + exchange):
// two launches less on the critical path.  One release fence per workgroup is the price (profiles/r02_barrier_lab.txt: ~3 us
// per kernel at 512 workgroups, against a launch + the gap between two dependent kernels).
#pragma once
#include "internal.h"
#include "reduce.h"

namespace mgcr {

constexpr int PW_MAX_SCALARS = 64;

struct PwTail {
    uint64_t *mb[PW_MAX_RANKS];   // rank r's mailbox as mapped into this process (own one for r == rank)
    int rank, nranks;
    uint32_t seq;                 // this exchange's sequence number (comm.hip pw_next_seq)
    int *err;                     // pinned host word: a wait timed out
    long long timeout;            // wall_clock64 ticks
    unsigned *ticket;             // device counter, 0 between launches
    const double *pa;             // slab A: na scalars [k][RED_MAX_BLOCKS] ...
    int na;
    const double *pb;             // ... slab B: nb scalars
    int nb;
    double *out;                  // na + nb sums over all workgroups and ranks
    int nblk;                     // partials per scalar
};

// sum over the ranks of this wave's `acc` (scalar k of exchange t.seq): every lane < nranks talks to one peer.  The wire format and
// the time-out handling are fold_pw_kernel's (comm.hip: value and sequence number in the same 8-byte store, two words per double).
__device__ __forceinline__ double pw_exchange_scalar(uint64_t *const *mb, int rank, int nranks, uint32_t seq, int *err, long long timeout, int k,
                                                     double acc) {
    const int lane = threadIdx.x & 63;
    acc = __shfl(acc, 0, 64);
    const size_t slot = (size_t)(seq & 1u) * PW_MAX_RANKS;
    double val = acc;
    if (lane < nranks && lane != rank) {
        const unsigned long long bits = (unsigned long long)__double_as_longlong(acc);
        uint64_t *dst = mb[lane] + ((slot + (size_t)rank) * PW_MAX_SCALARS + (size_t)k) * 2;
        __hip_atomic_store(dst, (bits & 0xffffffffull) | ((unsigned long long)seq << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(dst + 1, (bits >> 32) | ((unsigned long long)seq << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        uint64_t *src = mb[rank] + ((slot + (size_t)lane) * PW_MAX_SCALARS + (size_t)k) * 2;
        unsigned long long w0 = 0, w1 = 0;
        const long long t0 = wall_clock64();
        bool ok = false;
        if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) == 0) {
            for (;;) {
                w0 = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                w1 = __hip_atomic_load(src + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                if ((uint32_t)(w0 >> 32) == seq && (uint32_t)(w1 >> 32) == seq) { ok = true; break; }
                if (wall_clock64() - t0 > timeout) break;
                __builtin_amdgcn_s_sleep(2);
            }
        }
        if (ok) {
            val = __longlong_as_double((long long)((w0 & 0xffffffffull) | (w1 << 32)));
        } else {  // a peer never arrived: flag it (the host turns it into MGCR_ERR_COMM) and poison the result
            __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            val = __longlong_as_double(0x7ff8000000000000LL);
        }
    }
    double tot = 0.;
    for (int r = 0; r < nranks; r++) tot += __shfl(val, r, 64);  // rank order: the same bits on every rank
    return tot;
}

// Called by every thread of every workgroup that has written its partials (nblocks of them: the workgroups that reach this
// point).  Returns at once in all workgroups but the one that arrives last, which folds and exchanges all na + nb scalars.
__device__ __forceinline__ void pw_tail(const PwTail &t, int nblocks) {
    __shared__ int pw_is_last;
    __threadfence();   // this thread's partials are visible device-wide before the ticket is taken
    __syncthreads();
    if (threadIdx.x == 0)
        pw_is_last = __hip_atomic_fetch_add(t.ticket, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)nblocks - 1u;
    __syncthreads();
    if (!pw_is_last) return;
    __threadfence();   // (acquire side: the other workgroups' partials)
    const int wave = (int)threadIdx.x >> 6, nw = (int)blockDim.x >> 6, lane = threadIdx.x & 63;
    for (int k = wave; k < t.na + t.nb; k += nw) {   // one wave per scalar, like fold_pw_kernel's one-wave workgroups
        const double *src = k < t.na ? t.pa + (size_t)k * RED_MAX_BLOCKS : t.pb + (size_t)(k - t.na) * RED_MAX_BLOCKS;
        const double acc = wave_fold_slab(src, t.nblk);
        const double tot = pw_exchange_scalar(t.mb, t.rank, t.nranks, t.seq, t.err, t.timeout, k, acc);
        if (lane == 0) t.out[k] = tot;
    }
    if (threadIdx.x == 0) __hip_atomic_store(t.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // for the next launch (stream order)
}

// comm.hip: fills everything but pa / na / pb / nb / out / nblk and takes the exchange's sequence number; false: this communicator's
// scalars do not travel by peer writes (or the tail is switched off): the caller keeps the separate fold + exchange launch
bool comm_pw_tail_begin(Comm *c, PwTail *t);
int64_t comm_pw_tail_count();
bool set_pw_tail_enabled(bool on);

}  // namespace mgcr
