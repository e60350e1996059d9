// Lane exchange l <-> l ^ OFF of a wave64 without the LDS permute unit (ds_bpermute: ~100 cycles of latency per step of a
// reduction tree): DPP moves for OFF = 1, 2, 4, 8 and gfx950's v_permlane16_swap / v_permlane32_swap for 16, 32.  Checks the
// primitives reduce.h builds its trees from.   hipcc -O3 --offload-arch=gfx950 tools/xor_lane_lab.hip -o tools/build/xor_lane_lab
#include <hip/hip_runtime.h>
template <int OFF>
__device__ __forceinline__ int xor_lane_b32(int v) {
    if constexpr (OFF == 1) return __builtin_amdgcn_mov_dpp(v, 0xB1, 0xf, 0xf, false);        // quad_perm [1,0,3,2]
    else if constexpr (OFF == 2) return __builtin_amdgcn_mov_dpp(v, 0x4E, 0xf, 0xf, false);   // quad_perm [2,3,0,1]
    else if constexpr (OFF == 4) {
        const int t = __builtin_amdgcn_mov_dpp(v, 0x141, 0xf, 0xf, false);                    // row_half_mirror: l -> l ^ 7
        return __builtin_amdgcn_mov_dpp(t, 0x1B, 0xf, 0xf, false);                            // quad_perm [3,2,1,0]: l -> l ^ 3
    } else if constexpr (OFF == 8) return __builtin_amdgcn_mov_dpp(v, 0x128, 0xf, 0xf, false); // row_ror:8
    else if constexpr (OFF == 16) {
        auto r = __builtin_amdgcn_permlane16_swap((unsigned)v, (unsigned)v, false, false);
        const bool odd_row = (threadIdx.x >> 4) & 1;
        return odd_row ? (int)r[0] : (int)r[1];
    } else {
        auto r = __builtin_amdgcn_permlane32_swap((unsigned)v, (unsigned)v, false, false);
        const bool upper = (threadIdx.x >> 5) & 1;
        return upper ? (int)r[0] : (int)r[1];
    }
}
template <int OFF>
__device__ __forceinline__ double xor_lane(double v) {
    return __hiloint2double(xor_lane_b32<OFF>(__double2hiint(v)), xor_lane_b32<OFF>(__double2loint(v)));
}
__global__ void k(const double *x, double *y, int *z) {
    int i = threadIdx.x;
    double v = x[i];
    y[i] = xor_lane<1>(v); y[64 + i] = xor_lane<2>(v); y[128 + i] = xor_lane<4>(v); y[192 + i] = xor_lane<8>(v);
    y[256 + i] = xor_lane<16>(v); y[320 + i] = xor_lane<32>(v);
}
int main() {
    double h[64], *dx, *dy, out[384]; int *dz;
    for (int i = 0; i < 64; i++) h[i] = i + 0.5;
    hipMalloc(&dx, 512); hipMalloc(&dy, 384 * 8); hipMalloc(&dz, 4);
    hipMemcpy(dx, h, 512, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dx, dy, dz);
    hipMemcpy(out, dy, 384 * 8, hipMemcpyDeviceToHost);
    int bad = 0; const int offs[6] = {1, 2, 4, 8, 16, 32};
    for (int o = 0; o < 6; o++) for (int i = 0; i < 64; i++) if (out[o * 64 + i] != (i ^ offs[o]) + 0.5) bad++;
    printf("xor-lane exchange by DPP / permlane swap: %d mismatches\n", bad);
    return bad != 0;
}
