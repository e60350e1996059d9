// Stand-alone Field kernels (src/Fields.h): copy / fill / axpy-like / dot.  HBM-bound streaming
// kernels: 16 B per lane per access (one complex fp64 = one dwordx4), grid-stride over at most
// RED_MAX_BLOCKS x RED_THREADS threads so that every CU holds its full 32 waves.
#include "internal.h"
#include "reduce.h"

namespace mgcr {

int red_grid(int64_t n) {
    int64_t g = (n + RED_THREADS - 1) / RED_THREADS;
    if (g < 1) g = 1;
    if (g > RED_MAX_BLOCKS) g = RED_MAX_BLOCKS;
    return (int)g;
}

#define GRID_STRIDE(i, n) \
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (n); i += (int64_t)gridDim.x * blockDim.x)

__global__ void __launch_bounds__(RED_THREADS) copy_kernel(cplx *__restrict__ dst, const cplx *__restrict__ src, int64_t n) {
    GRID_STRIDE(i, n) dst[i] = src[i];
}
__global__ void __launch_bounds__(RED_THREADS) set_kernel(cplx *__restrict__ dst, cplx c, int64_t n) {
    GRID_STRIDE(i, n) dst[i] = c;
}
// same, as part of an operator apply enqueued by a solver: no-ops once that solve is over (SkipRef)
__global__ void __launch_bounds__(RED_THREADS) copy_skip_kernel(cplx *__restrict__ dst, const cplx *__restrict__ src, int64_t n,
                                                                const int *__restrict__ skip, int skip_it) {
    if (skip && skip[0] < skip[1] + skip_it) return;
    GRID_STRIDE(i, n) dst[i] = src[i];
}
__global__ void __launch_bounds__(RED_THREADS) set_skip_kernel(cplx *__restrict__ dst, cplx c, int64_t n,
                                                               const int *__restrict__ skip, int skip_it) {
    if (skip && skip[0] < skip[1] + skip_it) return;
    GRID_STRIDE(i, n) dst[i] = c;
}
__device__ __forceinline__ uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
// values on the 0.001 grid of Field::init_rand (src/Fields.h:133), from a repo-owned generator
__global__ void __launch_bounds__(RED_THREADS) fill_rhs_kernel(cplx *__restrict__ dst, int64_t n, uint64_t seed, int64_t off) {
    GRID_STRIDE(i, n) {
        uint64_t g = (uint64_t)(i + off);
        uint64_t a = splitmix64(seed * 0x100000001B3ull + 2 * g);
        uint64_t b = splitmix64(seed * 0x100000001B3ull + 2 * g + 1);
        dst[i] = make_double2((double)(a % 2000) / 1000. - 1., (double)(b % 2000) / 1000. - 1.);
    }
}
// out = a + alpha*b; out may alias a or b (each element is read before it is written by the same lane)
__global__ void __launch_bounds__(RED_THREADS) add_scaled_kernel(cplx *out, const cplx *a, cplx alpha, const cplx *b, int64_t n) {
    GRID_STRIDE(i, n) out[i] = cadd(a[i], cmul(alpha, b[i]));
}
__global__ void __launch_bounds__(RED_THREADS) scale_kernel(cplx *v, cplx alpha, int64_t n) {
    GRID_STRIDE(i, n) v[i] = cmul(alpha, v[i]);
}

// Field::gamma5 (src/Fields.h:310-339): out[index with spinor 0<->2, 1<->3] = in[index]; the spinor dimension has 4
// entries and `inner` = the product of the dimensions after it (row-major index algebra of src/Mesh.h:156-165)
__global__ void __launch_bounds__(RED_THREADS) gamma5_kernel(cplx *__restrict__ out, const cplx *__restrict__ in, int64_t n, int64_t inner) {
    GRID_STRIDE(i, n) {
        const int64_t s = (i / inner) & 3;  // spinor index of element i
        out[i + ((s ^ 2) - s) * inner] = in[i];
    }
}

// partial sums of conj(a_i) b_i  ->  parts[0][blk] (re), parts[1][blk] (im)
__global__ void __launch_bounds__(RED_THREADS) dot_partials_kernel(const cplx *__restrict__ a, const cplx *__restrict__ b,
                                                                  int64_t n, double *__restrict__ parts) {
    __shared__ double lds[2 * 17];
    double v[2] = {0., 0.};
    GRID_STRIDE(i, n) {
        cplx t = cconj_mul(a[i], b[i]);
        v[0] += t.x;
        v[1] += t.y;
    }
    block_sum_bcast<2>(v, lds);
    if (threadIdx.x == 0) {
        parts[blockIdx.x] = v[0];
        parts[RED_MAX_BLOCKS + blockIdx.x] = v[1];
    }
}

// Folds partial slabs to scalars, one WAVE per scalar (all scalars of up to two slabs in one launch): wave w walks
// the 16 wave-sized pieces of its scalar's slab and adds the pieces' shuffle-tree sums in index order — the very
// tree fold_partials / block_sum_bcast build over a 1024-thread workgroup, so the result has the same bits as the
// in-kernel folds of the single-GPU path.
__global__ void __launch_bounds__(64) fold_kernel(const double *__restrict__ pa, int na, double *__restrict__ outa,
                                                  const double *__restrict__ pb, int nb, double *__restrict__ outb, int nblk) {
    const int k = blockIdx.x, lane = threadIdx.x;
    const double *src = k < na ? pa + (size_t)k * RED_MAX_BLOCKS : pb + (size_t)(k - na) * RED_MAX_BLOCKS;
    const double acc = wave_fold_slab(src, nblk);
    if (lane == 0) {
        if (k < na) outa[k] = acc;
        else outb[k - na] = acc;
    }
}

#define LAUNCH(kernel, grid, ...)                                                     \
    do {                                                                              \
        hipLaunchKernelGGL(kernel, dim3(grid), dim3(RED_THREADS), 0, ctx().stream, __VA_ARGS__); \
        MGCR_HIP(hipGetLastError());                                                  \
    } while (0)

int k_copy(cplx *dst, const cplx *src, int64_t n) {
    if (n == 0 || dst == src) return MGCR_OK;
    LAUNCH(copy_kernel, red_grid(n), dst, src, n);
    return MGCR_OK;
}
int k_zero(cplx *dst, int64_t n) { return k_set_constant(dst, make_double2(0., 0.), n); }
// Inside an operator apply (GCR / MG used as preconditioner): the output vector may be live storage of
// the calling solver (a direction slot), so even the initialisation must respect its stop predicate.
int k_copy_apply(cplx *dst, const cplx *src, int64_t n) {
    if (n == 0 || dst == src) return MGCR_OK;
    SkipRef sk = get_apply_skip();
    LAUNCH(copy_skip_kernel, red_grid(n), dst, src, n, sk.p, sk.it);
    return MGCR_OK;
}
int k_zero_apply(cplx *dst, int64_t n) {
    if (n == 0) return MGCR_OK;
    SkipRef sk = get_apply_skip();
    LAUNCH(set_skip_kernel, red_grid(n), dst, make_double2(0., 0.), n, sk.p, sk.it);
    return MGCR_OK;
}
int k_set_constant(cplx *dst, cplx c, int64_t n) {
    if (n == 0) return MGCR_OK;
    LAUNCH(set_kernel, red_grid(n), dst, c, n);
    return MGCR_OK;
}
int k_fill_rhs(cplx *dst, int64_t n, uint64_t seed, int64_t offset) {
    if (n == 0) return MGCR_OK;
    LAUNCH(fill_rhs_kernel, red_grid(n), dst, n, seed, offset);
    return MGCR_OK;
}
int k_add_scaled(cplx *out, const cplx *a, cplx alpha, const cplx *b, int64_t n) {
    if (n == 0) return MGCR_OK;
    LAUNCH(add_scaled_kernel, red_grid(n), out, a, alpha, b, n);
    return MGCR_OK;
}
int k_gamma5(cplx *out, const cplx *in, int64_t n, int64_t inner) {
    if (n == 0) return MGCR_OK;
    LAUNCH(gamma5_kernel, red_grid(n), out, in, n, inner);
    return MGCR_OK;
}
int k_scale(cplx *v, cplx alpha, int64_t n) {
    if (n == 0) return MGCR_OK;
    LAUNCH(scale_kernel, red_grid(n), v, alpha, n);
    return MGCR_OK;
}
int k_dot_partials(const cplx *a, const cplx *b, int64_t n, double *parts, int *nblk) {
    int g = red_grid(n);
    LAUNCH(dot_partials_kernel, g, a, b, n, parts);
    *nblk = g;
    return MGCR_OK;
}
int k_fold2(const double *pa, int na, double *outa, const double *pb, int nb, double *outb, int nblk) {
    if (na + nb <= 0) return MGCR_OK;
    hipLaunchKernelGGL(fold_kernel, dim3(na + nb), dim3(64), 0, ctx().stream, pa, na, outa, pb, nb, outb, nblk);
    MGCR_HIP(hipGetLastError());
    return MGCR_OK;
}
int k_fold(const double *parts, int nblk, int nscal, double *out_dev) {
    return k_fold2(parts, nscal, out_dev, nullptr, 0, nullptr, nblk);
}

}  // namespace mgcr
