// What does it cost to keep the vectors of a persistent (many-phases-in-one-launch) solver coherent across the 8 XCDs
// WITHOUT release / acquire fences at every device-wide barrier?  (barrier_lab: the rendezvous alone takes 1.7 us with
// per-block flags; the fences add 3.3 us at 256 workgroups, which makes a phase boundary as dear as a kernel boundary.)
// Three ways to read / write a 262 144-row complex vector set (64^3: the coarsest level of the 256^3 hierarchy):
//   plain  ordinary loads / stores on hipMalloc memory (not coherent across XCDs between barriers: timing reference)
//   uc     ordinary loads / stores on hipExtMallocWithFlags(hipDeviceMallocUncached) memory
//   sc1    global_load/store_dwordx4 ... sc1 (the agent-scope cache policy of a relaxed atomic) on hipMalloc memory
// Phases, each followed by one relaxed flag barrier:  A  y = a x + z;   G  y = 7-point gather of x;   M  y = sum of 8 vectors
//   hipcc -O3 --offload-arch=gfx950 tools/coherent_lab.hip -o tools/build/coherent_lab && tools/build/coherent_lab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

typedef double2 cplx;
typedef double v2d __attribute__((ext_vector_type(2)));   // what the inline assembly sees: 4 consecutive VGPRs

__device__ __forceinline__ void grid_barrier_flags(unsigned *arrive, unsigned *go, unsigned nblocks, unsigned gen) {
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(arrive + blockIdx.x * 16, gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (blockIdx.x == 0) {
        for (unsigned t = threadIdx.x; t < nblocks; t += blockDim.x)
            while (__hip_atomic_load(arrive + t * 16, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != gen) __builtin_amdgcn_s_sleep(1);
        __syncthreads();
        for (unsigned t = threadIdx.x; t < nblocks; t += blockDim.x)
            __hip_atomic_store(go + t * 16, gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (threadIdx.x == 0)
        while (__hip_atomic_load(go + blockIdx.x * 16, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != gen) __builtin_amdgcn_s_sleep(1);
    __syncthreads();
}

template <bool SC1>
__device__ __forceinline__ void load8(cplx (&v)[8], const cplx *const (&p)[8]) {
    if (SC1) {
        v2d w[8];
        asm volatile(
            "global_load_dwordx4 %0, %8, off sc1\n global_load_dwordx4 %1, %9, off sc1\n global_load_dwordx4 %2, %10, off sc1\n"
            "global_load_dwordx4 %3, %11, off sc1\n global_load_dwordx4 %4, %12, off sc1\n global_load_dwordx4 %5, %13, off sc1\n"
            "global_load_dwordx4 %6, %14, off sc1\n global_load_dwordx4 %7, %15, off sc1\n s_waitcnt vmcnt(0)"
            : "=&v"(w[0]), "=&v"(w[1]), "=&v"(w[2]), "=&v"(w[3]), "=&v"(w[4]), "=&v"(w[5]), "=&v"(w[6]), "=&v"(w[7])
            : "v"(p[0]), "v"(p[1]), "v"(p[2]), "v"(p[3]), "v"(p[4]), "v"(p[5]), "v"(p[6]), "v"(p[7]) : "memory");
#pragma unroll
        for (int k = 0; k < 8; k++) v[k] = make_double2(w[k].x, w[k].y);
    } else {
#pragma unroll
        for (int k = 0; k < 8; k++) v[k] = *p[k];
    }
}

template <bool SC1>
__device__ __forceinline__ void store1(cplx *p, cplx v) {
    if (SC1) {
        v2d w = {v.x, v.y};
        asm volatile("global_store_dwordx4 %0, %1, off sc1\n s_waitcnt vmcnt(0)" :: "v"(p), "v"(w) : "memory");
    } else *p = v;
}

// PHASE 0: A (2 loads used), 1: G (7 gathers), 2: M (8 streams)
template <int PHASE, bool SC1>
__global__ void __launch_bounds__(1024) k_phase(cplx *vec, int64_t n, int nvec, int reps, unsigned *ctr, unsigned gen0, int barrier) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t n1 = 64, n2 = 64 * 64;
    for (int r = 0; r < reps; r++) {
        const cplx *src = vec + (int64_t)(r % nvec) * n;
        cplx *dst = vec + (int64_t)((r + 9) % nvec) * n;
        const cplx *p[8];
        if (PHASE == 1) {
            const int64_t off[8] = {0, 1, -1, n1, -n1, n2, -n2, 0};
#pragma unroll
            for (int k = 0; k < 8; k++) { int64_t j = i + off[k]; j = j < 0 ? j + n : j >= n ? j - n : j; p[k] = src + j; }
        } else {
#pragma unroll
            for (int k = 0; k < 8; k++) p[k] = vec + (int64_t)((r + k) % nvec) * n + i;
            if (PHASE == 0) {
#pragma unroll
                for (int k = 2; k < 8; k++) p[k] = p[k & 1];   // 2 distinct streams; the repeats hit in the first level
            }
        }
        cplx v[8];
        load8<SC1>(v, p);
        cplx s = make_double2(0., 0.);
#pragma unroll
        for (int k = 0; k < 8; k++) { s.x += v[k].x * 0.125; s.y += v[k].y * 0.125; }
        store1<SC1>(dst + i, s);
        if (barrier) grid_barrier_flags(ctr, ctr + 16 * 1024, gridDim.x, gen0 + (unsigned)r);
    }
}


// ---- correctness: does data written with sc1 stores in one phase arrive at sc1 loads of OTHER workgroups (other XCDs)
// in the next phase, with only the relaxed flag barrier and a wave-level s_waitcnt in between?  Each step shifts a
// vector by S rows and adds 1: after R steps v[i] must be init[(i + R S) % n] + R.  KIND 0: plain loads / stores
// (expected to fail now and then: stale lines in L1 / L2), 1: buffer_load / buffer_store ... sc1 (compiler-managed waits)
typedef int v4i __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc_of(const void *p) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, 0x7fffffff, 0x00020000);
}
template <int KIND>
__global__ void __launch_bounds__(1024) k_shift(cplx *a, cplx *b, int64_t n, int64_t S, int reps, unsigned *ctr, unsigned gen0) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (int r = 0; r < reps; r++) {
        const cplx *src = (r & 1) ? b : a;
        cplx *dst = (r & 1) ? a : b;
        const int64_t j = (i + S) % n;
        cplx v;
        if (KIND == 0) {
            v = src[j];
            v.x += 1.; v.y -= 1.;
            dst[i] = v;
        } else {
            v4i w = __builtin_amdgcn_raw_buffer_load_b128(rsrc_of(src), (int)(j * 16), 0, 16);
            union { v4i w; cplx c; } u; u.w = w;
            u.c.x += 1.; u.c.y -= 1.;
            __builtin_amdgcn_raw_buffer_store_b128(u.w, rsrc_of(dst), (int)(i * 16), 0, 16);
        }
        __builtin_amdgcn_s_waitcnt(0);   // this wave's stores are acknowledged before the workgroup arrives
        grid_barrier_flags(ctr, ctr + 16 * 1024, gridDim.x, gen0 + (unsigned)r);
    }
}

int main() {
    const int64_t n = 262144;
    const int nvec = 22;   // 88 MB: GCR(10)'s p, Ap, r, x
    unsigned *ctr;
    const size_t ctr_bytes = sizeof(unsigned) * 2 * 16 * 1024;
    CK(hipMalloc(&ctr, ctr_bytes)); CK(hipMemset(ctr, 0, ctr_bytes));
    cplx *plain, *uc;
    CK(hipMalloc(&plain, sizeof(cplx) * n * nvec)); CK(hipMemset(plain, 0, sizeof(cplx) * n * nvec));
    CK(hipExtMallocWithFlags((void **)&uc, sizeof(cplx) * n * nvec, hipDeviceMallocUncached)); CK(hipMemset(uc, 0, sizeof(cplx) * n * nvec));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    typedef void (*kern_t)(cplx *, int64_t, int, int, unsigned *, unsigned, int);
    const kern_t kern[3][2] = {{k_phase<0, false>, k_phase<0, true>}, {k_phase<1, false>, k_phase<1, true>}, {k_phase<2, false>, k_phase<2, true>}};
    const char *pname[3] = {"A  y = a x + z (3 V = 12.6 MB)", "G  7-point gather (2 V = 8.4 MB)", "M  8 streams + 1 (9 V = 37.7 MB)"};
    unsigned gen = 1;
    const int reps = 1000;
    for (int ph = 0; ph < 3; ph++) {
        for (int kind = 0; kind < 3; kind++) {   // plain, uc, sc1
            cplx *mem = kind == 1 ? uc : plain;
            kern_t k = kern[ph][kind == 2 ? 1 : 0];
            float ms[2];
            for (int barrier = 1; barrier >= 0; barrier--) {
                hipLaunchKernelGGL(k, dim3(256), dim3(1024), 0, 0, mem, n, nvec, 10, ctr, gen, barrier); gen += 10;
                CK(hipDeviceSynchronize());
                CK(hipEventRecord(e0, 0));
                hipLaunchKernelGGL(k, dim3(256), dim3(1024), 0, 0, mem, n, nvec, reps, ctr, gen, barrier); gen += reps;
                CK(hipEventRecord(e1, 0));
                CK(hipEventSynchronize(e1));
                CK(hipEventElapsedTime(&ms[barrier], e0, e1));
            }
            printf("%-36s %-5s: %.2f us per phase with the barrier, %.2f us free-running\n", pname[ph],
                   kind == 0 ? "plain" : kind == 1 ? "uc" : "sc1", ms[1] * 1e3 / reps, ms[0] * 1e3 / reps);
            fflush(stdout);
        }
    }
    // the same phases as separate launches (plain memory): what the solver pays today
    for (int ph = 0; ph < 3; ph++) {
        kern_t k = kern[ph][0];
        hipLaunchKernelGGL(k, dim3(256), dim3(1024), 0, 0, plain, n, nvec, 1, ctr, 0u, 0);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0, 0));
        for (int r = 0; r < reps; r++) hipLaunchKernelGGL(k, dim3(256), dim3(1024), 0, 0, plain + (int64_t)(r % 4) * n, n, nvec - 4, 1, ctr, 0u, 0);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-36s one launch per phase: %.2f us\n", pname[ph], ms * 1e3 / reps);
    }
    // coherence check
    {
        cplx *a, *b;
        CK(hipMalloc(&a, sizeof(cplx) * n)); CK(hipMalloc(&b, sizeof(cplx) * n));
        cplx *h = (cplx *)malloc(sizeof(cplx) * n);
        const int64_t S = 77777;
        for (int kind = 0; kind < 2; kind++) {
            for (int trial = 0; trial < 3; trial++) {
                const int R = 1000 + trial;
                for (int64_t i = 0; i < n; i++) { h[i].x = (double)i; h[i].y = -(double)i; }
                CK(hipMemcpy(a, h, sizeof(cplx) * n, hipMemcpyHostToDevice));
                CK(hipMemset(b, 0xff, sizeof(cplx) * n));
                if (kind == 0) hipLaunchKernelGGL(k_shift<0>, dim3(256), dim3(1024), 0, 0, a, b, n, S, R, ctr, gen);
                else hipLaunchKernelGGL(k_shift<1>, dim3(256), dim3(1024), 0, 0, a, b, n, S, R, ctr, gen);
                gen += R;
                CK(hipDeviceSynchronize());
                CK(hipMemcpy(h, (R & 1) ? b : a, sizeof(cplx) * n, hipMemcpyDeviceToHost));
                int64_t bad = 0;
                for (int64_t i = 0; i < n; i++) {
                    const double want = (double)((i + (int64_t)R * S) % n) + R;
                    if (h[i].x != want || h[i].y != -(double)((i + (int64_t)R * S) % n) - R) bad++;
                }
                printf("coherence check, %s, %d steps: %lld of %lld rows wrong\n", kind == 0 ? "plain loads / stores" : "buffer_load / store sc1", R, (long long)bad, (long long)n);
            }
        }
    }
    return 0;
}
