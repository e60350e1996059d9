// step_apply_xr_tile_kernel: part of gcr_fused.hip (included there, inside namespace mgcr, after DotVecs) — kept in a file of its own
// so that tools/xr_tile_regs.sh can compile just this kernel when its register budget is being worked on.
#pragma once
#ifndef XR_TILE_APC_NDT
#define XR_TILE_APC_NDT 5    // up to this many: the newest direction's A p stays in registers for its dot product (APC)
#endif
#ifndef XR_TILE_OCC8_NDT
#define XR_TILE_OCC8_NDT 2   // up to this many direction streams: 64 registers, two workgroups per CU (more spill there — 3, 13 registers at 3, 4 streams, whose scratch traffic PMC shows as 1.2-1.3x the model bytes; one workgroup per CU with 88 registers is faster: 288 against 293 us per iteration at 256^3)
#endif
// Bandwidth regime with the LDS window AND a row map whose step is the reach of the far slots (step_apply_tile_kernel's CARRY: a
// 256 x 256 x Z grid): the residual update INSIDE the windowed apply.  What made that lose at 128^3 — r' = r - alpha Ap formed again
// for each of a row's 7 gathered entries, 14 requests per row — is here 2 requests per row (+ the halo's): the five near slots come
// from the window, which holds r' already, and the two far ones are the r' this thread formed one trip ago and the one it forms now
// for its next trip.  The launch reads r and Ap once (Ap is also the newest of the direction streams of the dots: carried along in
// registers when APC, else requested again — an L2 hit), writes r' and A r': against xr_update_kernel (read r, Ap, write r') followed by
// step_apply_tile_kernel (read r', Ap_j, write A r') two vector reads and one launch less per iteration.  Same expression for r'
// wherever it is formed (the bits xr_update_kernel writes), same row arithmetic and per-thread accumulation order as
// step_apply_tile_kernel; |r'|^2 is summed over the apply's (banded) row map where xr_update_kernel sums over the plain grid-stride one:
// the one difference in bits (the oracle's device model knows it: orc_set_device_xr_banded).  Lean cycles only (r_out != r_in), single GPU.
template <int NS, bool RARE, int NDT, bool APC>
__global__ void __launch_bounds__(RED_THREADS, (NDT <= XR_TILE_OCC8_NDT ? 8 : 4)) step_apply_xr_tile_kernel(RowMat m, const cplx *__restrict__ r_in, const cplx *__restrict__ ap,
                                                                                           cplx *__restrict__ r_out, cplx *__restrict__ y, DotVecs d,
                                                                                           int64_t n, int nlogical, RowMap rm, double *__restrict__ parts,
                                                                                           double *__restrict__ partsR, DevState *__restrict__ st, int it,
                                                                                           const double *__restrict__ partsA, int nblkA, int strideA,
                                                                                           cplx *__restrict__ den_slot, int slot, LeanCoef *__restrict__ lc) {
    __shared__ double lds[(2 * NDT + 1) * 17 > 4 * 17 ? (2 * NDT + 1) * 17 : 4 * 17];
    extern __shared__ __attribute__((aligned(16))) unsigned char step_smem[];
    constexpr unsigned NEAR = 0x3eu;
    constexpr int NC = STEN_COMMON;
    static_assert(NS == 7 || (RARE && NS == 9), "7 common slots, optionally 2 rare ones behind them");
    if (st->stop_at < st->base + it) return;
    const int lb = logical_workgroup(rm, (int)blockIdx.x, (int)gridDim.x);
    int64_t i, end, stride;
    bool row_ok;   // (false: a thread of a plane's short last tile beyond the plane's end — it fills its window entry, nothing else)
    row_range(rm, lb < nlogical ? lb : 0, nlogical, n, &i, &end, &stride, &row_ok);
    const int32_t H = m.sten_halo_f;
    auto clampj = [&](int64_t j) -> int32_t { return (int32_t)(j < 0 ? 0 : j > m.sten_last ? m.sten_last : j); };
    // the first trip's operands are requested before alpha is folded from the partials (xr_update_kernel's prologue)
    const int32_t jp = clampj(i + m.sten_off[0]), jc = clampj(i);
    const cplx p_r = r_in[jp], p_a = ap[jp], c_r = r_in[jc], c_a = ap[jc];
    double sa[4];
    fold_partials<4>(partsA, nblkA, strideA, sa, lds);
    const cplx num = make_double2(sa[0], sa[1]), den = make_double2(sa[2], sa[3]);
    const cplx alpha = to_sgpr(cdiv(num, den));
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        *den_slot = den;
        st->npend = slot + 1;
    }
    if (blockIdx.x == 0 && (int)threadIdx.x < LND) lean_pending_update(lc, slot, alpha, (int)threadIdx.x);
    if (lb >= nlogical) return;
    cplx *win = reinterpret_cast<cplx *>(step_smem);   // 2 x [H + RED_THREADS + H]
    const int wlen = RED_THREADS + 2 * H;
    const int lane = (int)(threadIdx.x & 63);
    auto upd = [&](cplx rv, cplx av) -> cplx { return csub(rv, cmul(alpha, av)); };   // xr_update_kernel's expression
    double v[2 * NDT + 1];
#pragma unroll
    for (int j = 0; j < 2 * NDT + 1; j++) v[j] = 0.;
    cplx prev = upd(p_r, p_a), cur = upd(c_r, c_a);   // r' of rows i - step and i
    cplx cur_a = c_a;                                 // APC: Ap of row i
    // The window of a trip is written ONE TRIP AHEAD (the first one here): a trip requests the next trip's own entry — its + step
    // neighbour — and the next trip's halo together, so a halo line is asked for while the workgroup that owns it asks for it too
    // (one fetch from memory for both; requested a trip later it had left the L2 again: PMC 1.15-1.23x the model's bytes, now 1.0x)
    const int hidx = (int)threadIdx.x < 2 * H ? ((int)threadIdx.x < H ? (int)threadIdx.x : RED_THREADS + (int)threadIdx.x) : -1;
    auto halo_row = [&](int64_t base) -> int32_t {
        const int t = (int)threadIdx.x;
        return clampj(t < H ? base - H + t : base + RED_THREADS + (t - H));
    };
    {
        win[H + threadIdx.x] = cur;
        if (hidx >= 0) {
            const int32_t jh = halo_row(i - threadIdx.x);
            win[hidx] = upd(r_in[jh], ap[jh]);
        }
    }
    int buf = 0;
    for (int64_t base = i - threadIdx.x; base < end; base += stride, i += stride, buf ^= 1) {   // uniform trip count per workgroup
        const bool live = row_ok && i < end;
        const int32_t wave = __builtin_amdgcn_readfirstlane((int32_t)(i >> 6));
        const sten_planes_ptr pp = sten_wave_planes(m, wave < m.sten_nwaves ? wave : m.sten_nwaves);
        uint64_t pl[NS];
#pragma unroll
        for (int c = 0; c < NS; c++) pl[c] = pp[c];
        const int32_t jn = clampj(i + stride);   // the + step neighbour now, this thread's own row next trip
        const cplx n_r = r_in[jn], n_a = ap[jn];
        cplx hr = make_double2(0., 0.), ha = hr;
        if (hidx >= 0) {
            const int32_t jh = halo_row(base + stride);
            hr = r_in[jh];
            ha = ap[jh];
        }
        __builtin_amdgcn_sched_barrier(0);   // everything above is in flight before anything is waited for
        if (live) {
            r_out[i] = cur;
            v[2 * NDT] += cur.x * cur.x + cur.y * cur.y;
        }
        const cplx *sx = win + buf * wlen;
        __syncthreads();   // this trip's window (written during the previous trip) is complete; the other buffer is free
        const cplx next = upd(n_r, n_a);
        {
            cplx *sn = win + (buf ^ 1) * wlen;
            sn[H + threadIdx.x] = next;
            if (hidx >= 0) sn[hidx] = upd(hr, ha);
        }
        cplx sum = make_double2(0., 0.);
#pragma unroll
        for (int c = 0; c < NC; c++) {
            const cplx xv = (NEAR >> c & 1u) ? sx[H + (int)threadIdx.x + m.sten_off[c]] : (c == 0 ? prev : next);
            const bool on = (pl[c] >> lane & 1ull) != 0ull;
            const cplx nsum = cadd(sum, sten_term<1>(m, c, xv));
            sum.x = on ? nsum.x : sum.x;
            sum.y = on ? nsum.y : sum.y;
        }
        if (RARE) {
#pragma unroll
            for (int c = NC; c < NS; c++)
                if (pl[c] != 0ull) {   // wave-uniform: a wave of a boundary plane
                    const int32_t jr = clampj(i + m.sten_off[c]);
                    const cplx xr = upd(r_in[jr], ap[jr]);
                    const bool on = (pl[c] >> lane & 1ull) != 0ull;
                    const cplx nsum = cadd(sum, sten_term<1>(m, c, xr));
                    sum.x = on ? nsum.x : sum.x;
                    sum.y = on ? nsum.y : sum.y;
                }
        }
        const cplx yi = m.shift ? csub(cur, cmul(m.k, sum)) : sum;
        if (live) y[i] = yi;
        __builtin_amdgcn_sched_barrier(0);
        cplx b[NDT];
#pragma unroll
        for (int j = 0; j < NDT; j++) {
            if (APC && j == NDT - 1) b[j] = cur_a;   // the newest direction's A p: came with the residual update's operands
            else b[j] = live ? ld_stream<true>(d.v[j] + i) : make_double2(0., 0.);
        }
        if (live) {
#pragma unroll
            for (int j = 0; j < NDT; j++) {
                cplx t = cconj_mul(yi, b[j]);
                v[2 * j] += t.x;
                v[2 * j + 1] += t.y;
            }
        }
        prev = cur;
        cur = next;
        if (APC) cur_a = n_a;
    }
    const double mine = block_sum_owner<2 * NDT + 1>(v, lds);
    if (threadIdx.x < 2 * NDT) parts[(size_t)threadIdx.x * RED_MAX_BLOCKS + lb] = mine;
    if (threadIdx.x == 2 * NDT) partsR[lb] = mine;
}

