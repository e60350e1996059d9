// How long does a device-wide barrier take on MI355X?  (decides whether a persistent multi-iteration solver for the
// latency regime can beat one kernel launch per phase: a dependent launch costs ~4.5 us on this part)
//   hipcc -O3 --offload-arch=gfx950 tools/barrier_lab.hip -o tools/build/barrier_lab && tools/build/barrier_lab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// sense-reversing counter barrier in device memory; one thread per block arrives, the block then syncs
__device__ __forceinline__ void grid_barrier(unsigned *count, unsigned *gen, unsigned nblocks) {
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        const unsigned g = __hip_atomic_load(gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (__hip_atomic_fetch_add(count, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == nblocks - 1) {
            __hip_atomic_store(count, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_add(gen, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            while (__hip_atomic_load(gen, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) == g) __builtin_amdgcn_s_sleep(1);
        }
        __threadfence();
    }
    __syncthreads();
}

// two-level: blocks of one XCD (blockIdx & 7) meet on their own counter, the 8 XCD leaders on a global one
__device__ __forceinline__ void grid_barrier2(unsigned *xc /*[8][16]*/, unsigned *xg, unsigned *count, unsigned *gen, unsigned nblocks) {
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        const unsigned x = blockIdx.x & 7u, per = nblocks >> 3;
        const unsigned g = __hip_atomic_load(xg + x * 16, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (__hip_atomic_fetch_add(xc + x * 16, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == per - 1) {
            __hip_atomic_store(xc + x * 16, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned gg = __hip_atomic_load(gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (__hip_atomic_fetch_add(count, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == 7u) {
                __hip_atomic_store(count, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_fetch_add(gen, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                while (__hip_atomic_load(gen, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) == gg) __builtin_amdgcn_s_sleep(1);
            }
            __hip_atomic_fetch_add(xg + x * 16, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            while (__hip_atomic_load(xg + x * 16, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) == g) __builtin_amdgcn_s_sleep(1);
        }
        __threadfence();
    }
    __syncthreads();
}


// the same flat counter with relaxed atomics and no fences: the cost of the rendezvous alone (no cache maintenance)
__device__ __forceinline__ void grid_barrier_relaxed(unsigned *count, unsigned *gen, unsigned nblocks) {
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned g = __hip_atomic_load(gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (__hip_atomic_fetch_add(count, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == nblocks - 1) {
            __hip_atomic_store(count, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_add(gen, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            while (__hip_atomic_load(gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == g) __builtin_amdgcn_s_sleep(1);
        }
    }
    __syncthreads();
}

// no read-modify-write at all: block b stores its generation into arrive[b * 16] (one line each); the threads of
// block 0 each poll one of them, meet, and each stores go[t * 16]; block b polls go[b * 16].  FENCE: thread 0 of every
// block releases before arriving and acquires after leaving (what a solver phase needs for ordinary loads / stores)
template <bool FENCE>
__device__ __forceinline__ void grid_barrier_flags(unsigned *arrive, unsigned *go, unsigned nblocks, unsigned gen) {
    __syncthreads();
    if (threadIdx.x == 0) {
        if (FENCE) __atomic_thread_fence(__ATOMIC_RELEASE);   // agent scope by default in HIP device code
        __hip_atomic_store(arrive + blockIdx.x * 16, gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (blockIdx.x == 0) {
        for (unsigned t = threadIdx.x; t < nblocks; t += blockDim.x)
            while (__hip_atomic_load(arrive + t * 16, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != gen) __builtin_amdgcn_s_sleep(1);
        __syncthreads();
        for (unsigned t = threadIdx.x; t < nblocks; t += blockDim.x)
            __hip_atomic_store(go + t * 16, gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (threadIdx.x == 0) {
        while (__hip_atomic_load(go + blockIdx.x * 16, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != gen) __builtin_amdgcn_s_sleep(1);
        if (FENCE) __atomic_thread_fence(__ATOMIC_ACQUIRE);
    }
    __syncthreads();
}

template <int KIND>
__global__ void __launch_bounds__(1024) k_bar(unsigned *ctr, int reps, double *sink, const double *x, int64_t n, unsigned gen0) {
    double acc = 0.;
    for (int r = 0; r < reps; r++) {
        // a little work between barriers, like a solver phase: one element per thread
        const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x + (int64_t)r * 977) % n;
        acc += x[i];
        if (KIND == 1) grid_barrier(ctr, ctr + 16, gridDim.x);
        if (KIND == 2) grid_barrier2(ctr + 64, ctr + 64 + 128, ctr, ctr + 16, gridDim.x);
        if (KIND == 3) grid_barrier_relaxed(ctr, ctr + 16, gridDim.x);
        if (KIND == 4) grid_barrier_flags<false>(ctr + 1024, ctr + 1024 + 16 * 1024, gridDim.x, gen0 + (unsigned)r);
        if (KIND == 5) grid_barrier_flags<true>(ctr + 1024, ctr + 1024 + 16 * 1024, gridDim.x, gen0 + (unsigned)r);
    }
    if (acc == 12345.678) sink[0] = acc;
}

int main() {
    unsigned *ctr; double *sink, *x;
    const int64_t n = 262144;
    const size_t ctr_bytes = sizeof(unsigned) * (1024 + 2 * 16 * 1024);
    CK(hipMalloc(&ctr, ctr_bytes)); CK(hipMemset(ctr, 0, ctr_bytes));
    CK(hipMalloc(&sink, 8)); CK(hipMalloc(&x, 8 * n)); CK(hipMemset(x, 0, 8 * n));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    typedef void (*kern_t)(unsigned *, int, double *, const double *, int64_t, unsigned);
    const kern_t kern[6] = {k_bar<0>, k_bar<1>, k_bar<2>, k_bar<3>, k_bar<4>, k_bar<5>};
    unsigned gen = 1;   // the flag barriers compare generations: never reuse one
    for (int blocks : {64, 128, 256, 512}) {
        for (int threads : {256, 1024}) {
            if (blocks * threads > 512 * 1024) continue;
            const int reps = 2000;
            float ms[6];
            for (int kind = 0; kind < 6; kind++) {
                hipLaunchKernelGGL(kern[kind], dim3(blocks), dim3(threads), 0, 0, ctr, 10, sink, x, n, gen);
                gen += 10;
                CK(hipDeviceSynchronize());
                CK(hipEventRecord(e0, 0));
                hipLaunchKernelGGL(kern[kind], dim3(blocks), dim3(threads), 0, 0, ctr, reps, sink, x, n, gen);
                gen += reps;
                CK(hipEventRecord(e1, 0));
                CK(hipEventSynchronize(e1));
                CK(hipEventElapsedTime(&ms[kind], e0, e1));
            }
            printf("%d blocks x %d threads: loop without barrier %.2f us/rep; barriers: flat counter %.2f us, two-level (per XCD) %.2f us, "
                   "flat counter relaxed / no fences %.2f us, per-block flags relaxed / no fences %.2f us, per-block flags + release / acquire fences %.2f us\n",
                   blocks, threads, ms[0] * 1e3 / reps, (ms[1] - ms[0]) * 1e3 / reps, (ms[2] - ms[0]) * 1e3 / reps,
                   (ms[3] - ms[0]) * 1e3 / reps, (ms[4] - ms[0]) * 1e3 / reps, (ms[5] - ms[0]) * 1e3 / reps);
            fflush(stdout);
        }
    }
    // for comparison: dependent empty kernels
    {
        const int reps = 2000;
        hipLaunchKernelGGL(k_bar<0>, dim3(256), dim3(1024), 0, 0, ctr, 1, sink, x, n, 0u);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0, 0));
        for (int r = 0; r < reps; r++) hipLaunchKernelGGL(k_bar<0>, dim3(256), dim3(1024), 0, 0, ctr, 1, sink, x, n, 0u);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("256 x 1024 kernels launched back to back: %.2f us per kernel\n", ms * 1e3 / reps);
    }
    return 0;
}
