// What does a random 16-byte gather of x cost, and does a cache-policy bit change it?  (The irregular-matrix SpMV, bench.py
// irregular_spmv, turned out to be bound by its gathers: PMC shows one L2 request per gathered entry.)
// y[i] = sum_{c < W} x[col[c * N + i]]  — the ELL kernel without its value stream — with the gather issued as
//   aux 0   buffer_load_dwordx4                 aux 1  ... sc0        aux 2  ... nt         aux 3  ... sc0 nt
//   aux 16  ... sc1                             aux 17 ... sc0 sc1    aux 18 ... sc1 nt     aux 19 ... sc0 sc1 nt
// for columns within +-window of the row.  Also: the same with the window of x staged in LDS (windows up to 4096).
//   hipcc -O3 --offload-arch=gfx950 tools/gather_lab.hip -o tools/build/gather_lab && tools/build/gather_lab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <random>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef double2 cplx;
typedef int v4i __attribute__((ext_vector_type(4)));

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(const void *p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, (int)bytes, 0x00020000);
}
template <int AUX, int W>
__global__ void __launch_bounds__(256) k_gather(const int *__restrict__ col, const cplx *__restrict__ x, cplx *__restrict__ y, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const __amdgpu_buffer_rsrc_t r = rsrc(x, (unsigned)n * 16u);
    int j[W];
#pragma unroll
    for (int c = 0; c < W; c++) j[c] = __builtin_nontemporal_load(col + (size_t)c * n + i);
    double sx = 0., sy = 0.;
#pragma unroll
    for (int c = 0; c < W; c++) {
        const v4i w = __builtin_amdgcn_raw_buffer_load_b128(r, j[c] * 16, 0, AUX);
        sx += __hiloint2double(w.y, w.x);
        sy += __hiloint2double(w.w, w.z);
    }
    y[i] = make_double2(sx, sy);
}
// x window in LDS: workgroup of 1024 rows, window [r0 - H, r0 + 1024 + H)
template <int W, int H>
__global__ void __launch_bounds__(1024) k_gather_lds(const int *__restrict__ col, const cplx *__restrict__ x, cplx *__restrict__ y, int n) {
    extern __shared__ cplx win[];
    const int r0 = blockIdx.x * 1024;
    for (int t = threadIdx.x; t < 1024 + 2 * H; t += 1024) {
        int g = r0 - H + t;
        g = g < 0 ? 0 : g >= n ? n - 1 : g;
        win[t] = x[g];
    }
    __syncthreads();
    const int i = r0 + threadIdx.x;
    if (i >= n) return;
    int j[W];
#pragma unroll
    for (int c = 0; c < W; c++) j[c] = __builtin_nontemporal_load(col + (size_t)c * n + i);
    double sx = 0., sy = 0.;
#pragma unroll
    for (int c = 0; c < W; c++) {
        const cplx v = win[j[c] - r0 + H];
        sx += v.x; sy += v.y;
    }
    y[i] = make_double2(sx, sy);
}

template <int AUX>
static float run(const int *col, const cplx *x, cplx *y, int n, cplx *sweep_a, cplx *sweep_b, size_t sweep_n) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e9f;
    for (int rep = 0; rep < 5; rep++) {
        CK(hipMemcpyAsync(sweep_b, sweep_a, sweep_n, hipMemcpyDeviceToDevice, 0));   // cold caches
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL((k_gather<AUX, 9>), dim3((n + 255) / 256), dim3(256), 0, 0, col, x, y, n);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    return best;
}

int main() {
    const int n = 8 * 1024 * 1024, W = 9;
    cplx *x, *y, *sa, *sb;
    int *col;
    const size_t sweep_n = (size_t)256 << 20;
    CK(hipMalloc(&x, sizeof(cplx) * n)); CK(hipMalloc(&y, sizeof(cplx) * n)); CK(hipMalloc(&col, sizeof(int) * (size_t)n * W));
    CK(hipMalloc(&sa, sweep_n)); CK(hipMalloc(&sb, sweep_n));
    CK(hipMemset(x, 0, sizeof(cplx) * n)); CK(hipMemset(sa, 0, sweep_n));
    std::vector<int> h((size_t)n * W);
    for (int window : {256, 4096, 131072}) {
        std::mt19937 rng(7);
        for (int c = 0; c < W; c++)
            for (int i = 0; i < n; i++) {
                long j = (long)i + (long)(rng() % (2u * window + 1u)) - window;
                h[(size_t)c * n + i] = (int)(j < 0 ? 0 : j >= n ? n - 1 : j);
            }
        CK(hipMemcpy(col, h.data(), sizeof(int) * h.size(), hipMemcpyHostToDevice));
        const double gathers = (double)n * W;
        printf("window +-%d: %d gathers of 16 B per row, %.1f M gathers, index stream %.0f MB, y %.0f MB\n", window, W, gathers / 1e6, 4. * gathers / 1e6, 16. * n / 1e6);
#define R(A) { float ms = run<A>(col, x, y, n, sa, sb, sweep_n); printf("  aux %2d: %.3f ms = %.0f G gathers/s\n", A, ms, gathers / ms / 1e6); }
        R(0) R(1) R(2) R(3) R(16) R(17) R(18) R(19)
        if (window <= 4096) {
            hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            float best = 1e9f;
            for (int rep = 0; rep < 5; rep++) {
                CK(hipMemcpyAsync(sb, sa, sweep_n, hipMemcpyDeviceToDevice, 0));
                CK(hipEventRecord(e0, 0));
                if (window == 256) hipLaunchKernelGGL((k_gather_lds<9, 256>), dim3((n + 1023) / 1024), dim3(1024), sizeof(cplx) * (1024 + 512), 0, col, x, y, n);
                else {
                    CK(hipFuncSetAttribute((const void *)k_gather_lds<9, 4096>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(sizeof(cplx) * (1024 + 8192))));
                    hipLaunchKernelGGL((k_gather_lds<9, 4096>), dim3((n + 1023) / 1024), dim3(1024), sizeof(cplx) * (1024 + 8192), 0, col, x, y, n);
                }
                CK(hipEventRecord(e1, 0));
                CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                if (ms < best) best = ms;
            }
            printf("  LDS window (1024 rows per workgroup, +-%d staged): %.3f ms = %.0f G gathers/s\n", window, best, gathers / best / 1e6);
        }
        fflush(stdout);
    }
    return 0;
}
