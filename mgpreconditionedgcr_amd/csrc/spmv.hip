// Sparse operators on the device: Sparse<long> (CSR -> ELL slab + CSR tail) and
// HierarchicalSparse<long,int> (block-CSR of dense blocks).
//
//   reference: Sparse::operator()            src/Operator.h:330-346
//              DiracOp::operator()           src/Operator.h:569-575   (fused epilogue y = x - k*sum)
//              HierarchicalSparse::operator()  src/HierarchicalSparse.h:101-161
//              Dense::operator()             src/Operator.h:159-173
//
// All of it is HBM-bound (8 flop per 20 stored bytes): the kernels are built around coalesced
// 16-B-per-lane streams of the matrix, L2/MALL-served gathers of x, and wave64 shuffle / LDS
// reductions.  No MFMA.
#include <algorithm>
#include <climits>

#include "internal.h"
#include "reduce.h"
#include "spmv_dev.h"

namespace mgcr {

// ------------------------------------------------------------------------------------------------
// set-up: raw CSR (int64, as the reference stores it) -> ELL + tail, on the device
// ------------------------------------------------------------------------------------------------

// one thread per (row, lane): copies the row's first W entries into the slab, pads the rest with
// (last valid column, 0) so that padding never touches an x entry the row does not already read
__global__ void ell_fill_kernel(int64_t nrow, int64_t ncol, const int64_t *__restrict__ rowptr,
                                const int64_t *__restrict__ col, const cplx *__restrict__ val, int32_t W, int32_t L,
                                int32_t nchunk, int64_t npad, cplx *__restrict__ ell_val, int32_t *__restrict__ ell_col) {
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t row = t / L;
    int32_t l = (int32_t)(t % L);
    if (row >= npad) return;
    int64_t beg = 0, len = 0;
    if (row < nrow) {
        beg = rowptr[row];
        len = rowptr[row + 1] - beg;
    }
    int64_t take = len < W ? len : W;
    int32_t padcol = 0;
    if (take > 0) padcol = (int32_t)col[beg + take - 1];
    else if (row < ncol) padcol = (int32_t)row;
    for (int32_t c = 0; c < nchunk; c++) {
        int32_t w = c * L + l;
        int64_t dst = ((int64_t)c * npad + row) * L + l;
        if (w < take) {
            ell_val[dst] = val[beg + w];
            ell_col[dst] = (int32_t)col[beg + w];
        } else {
            ell_val[dst] = make_double2(0., 0.);
            ell_col[dst] = padcol;
        }
    }
}

__global__ void tail_fill_kernel(int64_t n_tail_rows, const int32_t *__restrict__ tail_rows,
                                 const int32_t *__restrict__ tail_ptr, const int64_t *__restrict__ rowptr,
                                 const int64_t *__restrict__ col, const cplx *__restrict__ val, int32_t W,
                                 int32_t *__restrict__ tail_col, cplx *__restrict__ tail_val) {
    int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    int lane = threadIdx.x & 63;
    if (wave >= n_tail_rows) return;
    int64_t row = tail_rows[wave];
    int64_t src = rowptr[row] + W;
    int32_t dst = tail_ptr[wave], cnt = tail_ptr[wave + 1] - dst;
    for (int32_t i = lane; i < cnt; i += 64) {
        tail_col[dst + i] = (int32_t)col[src + i];
        tail_val[dst + i] = val[src + i];
    }
}

// does any value have a non-zero imaginary part?
__global__ void imag_check_kernel(int64_t nnz, const cplx *__restrict__ val, int *__restrict__ has_imag) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nnz && val[i].y != 0.) *has_imag = 1;
}
__global__ void slab_real_kernel(int64_t n, const cplx *__restrict__ in, double *__restrict__ out) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = in[i].x;
}

// column range check of the uploaded CSR (bad indices would fault inside the SpMV gather)
__global__ void col_check_kernel(int64_t nnz, const int64_t *__restrict__ col, int64_t ncol, int *__restrict__ bad) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nnz && (col[i] < 0 || col[i] >= ncol)) *bad = 1;
}

template <typename T>
static int dev_upload(T **d, const T *h, size_t count) {
    *d = nullptr;
    if (count == 0) return MGCR_OK;
    hipError_t e = hipMalloc((void **)d, sizeof(T) * count);
    if (e != hipSuccess) {
        set_error("hipMalloc of %zu bytes failed: %s", sizeof(T) * count, hipGetErrorString(e));
        return MGCR_ERR_ALLOC;
    }
    if (h) MGCR_HIP(hipMemcpyAsync(*d, h, sizeof(T) * count, hipMemcpyHostToDevice, ctx().stream));
    return MGCR_OK;
}

void csr_free(CsrDev *c) {
    hipFree(c->ell_val); hipFree(c->ell_val_re); hipFree(c->ell_col);
    hipFree(c->pat_id); hipFree(c->pat_off); hipFree(c->pat_re); hipFree(c->pat_im); hipFree(c->sten_planes);
    hipFree(c->tail_rows); hipFree(c->tail_ptr); hipFree(c->tail_col); hipFree(c->tail_val);
    hipFree(c->tail_chunk); hipFree(c->tail_long); hipFree(c->win_tile_tail); hipFree(c->win_row_tail);
    *c = CsrDev();
}

// Picks the ELL width that minimises the bytes one SpMV streams, from the row-length histogram.
static int32_t choose_width(const std::vector<int64_t> &hist, int64_t nrow, int32_t maxlen) {
    // rows_ge[w] = #rows with len >= w ;  tail_nnz(W) = sum_{w > W} rows_ge[w]
    std::vector<int64_t> rows_ge(maxlen + 2, 0);
    for (int32_t w = maxlen; w >= 0; w--) rows_ge[w] = rows_ge[w + 1] + hist[w];
    std::vector<int64_t> tail(maxlen + 2, 0);
    for (int32_t w = maxlen - 1; w >= 0; w--) tail[w] = tail[w + 1] + rows_ge[w + 1];
    int32_t best = maxlen;
    double best_cost = 1e300;
    for (int32_t W = 0; W <= maxlen; W++) {
        // 20 B per stored entry; a tail entry also costs an uncoalesced row visit (~x2) and each
        // tail row a read-modify-write of y plus bookkeeping (~64 B)
        double cost = 20. * (double)W * (double)nrow + 40. * (double)tail[W] + 64. * (double)rows_ge[W + 1];
        if (cost < best_cost) { best_cost = cost; best = W; }
    }
    return best;
}

static int32_t choose_lanes(int64_t nrow, int32_t W) {
    // one thread per row (entries summed in CSR order, like the reference) whenever that alone
    // fills the machine; otherwise split rows over 2..16 lanes while padding stays below 10 %
    if (W <= 8 || nrow >= (int64_t)1 << 18) return 1;
    int32_t best = 1;
    for (int32_t L = 2; L <= 16; L *= 2) {
        int32_t padded = (W + L - 1) / L * L;
        if ((padded - W) * 10 > W) continue;
        best = L;
        if (nrow * L >= (int64_t)1 << 17) break;
    }
    return best;
}

// ------------------------------------------------------------------------------------------------
// Row-pattern dictionary.  Operators that come from a lattice or a grid repeat a handful of row
// patterns: the tuple (column - row, value) per stored entry is the same for every interior row and
// for every row of a given boundary class (7-point Poisson: 27 patterns for any grid size; Galerkin
// coarse operators of it likewise).  Such a matrix is stored as one 2-byte pattern id per row plus the
// pattern table, which the SpMV reads through L1/L2 — the 12-20 B per stored entry the ELL slab costs
// shrink to 2 B per ROW.  When the values differ from row to row but the sparsity pattern repeats
// (lattice-QCD hopping terms), only the column indices go into the table and the values stay in the slab.
// Entries are multiplied and added in the same order as the ELL kernels do, so y has the same bits.
// The dictionary is found on the device: a hash set of 64-bit row hashes (open addressing, atomicCAS),
// then every row is compared entry by entry with its pattern's first row, so a hash collision can only
// cost the compression (fallback to the plain slab), never correctness.
// ------------------------------------------------------------------------------------------------
constexpr int PAT_TABLE_BITS = 14;  // 16384 slots
constexpr int PAT_MAX = 4096;       // patterns: table stays L2-resident (<= 4096 * W * 20 B)
constexpr int64_t PAT_MIN_ROWS = 1 << 15;

__device__ __forceinline__ uint64_t pat_mix(uint64_t h, uint64_t v) {
    h = (h ^ v) * 0xff51afd7ed558ccdull;
    return h ^ (h >> 29);
}

template <bool VALS>
__device__ __forceinline__ uint64_t pat_row_hash(int64_t i, int64_t npad, int32_t W, const int32_t *__restrict__ col,
                                                 const cplx *__restrict__ val) {
    uint64_t h = 0x9e3779b97f4a7c15ull;
    for (int32_t w = 0; w < W; w++) {
        int64_t idx = (int64_t)w * npad + i;
        h = pat_mix(h, (uint64_t)(uint32_t)(col[idx] - (int32_t)i));
        if (VALS) {
            h = pat_mix(h, (uint64_t)__double_as_longlong(val[idx].x));
            h = pat_mix(h, (uint64_t)__double_as_longlong(val[idx].y));
        }
    }
    return h | 1ull;  // 0 marks an empty slot
}

__device__ __forceinline__ int pat_find(uint64_t h, const unsigned long long *keys) {
    const int mask = (1 << PAT_TABLE_BITS) - 1;
    int s = (int)(h >> 20) & mask;
    for (int probe = 0; probe <= mask; probe++) {
        if (keys[s] == h) return s;
        s = (s + 1) & mask;
    }
    return -1;
}

template <bool VALS>
__global__ void pat_insert_kernel(int64_t nrow, int64_t npad, int32_t W, const int32_t *__restrict__ col,
                                  const cplx *__restrict__ val, unsigned long long *keys, int *rep, int *count,
                                  volatile int *overflow) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nrow || *overflow) return;
    const uint64_t h = pat_row_hash<VALS>(i, npad, W, col, val);
    const int mask = (1 << PAT_TABLE_BITS) - 1;
    int s = (int)(h >> 20) & mask;
    for (int probe = 0; probe <= mask; probe++) {
        unsigned long long k = *(volatile unsigned long long *)(keys + s);
        if (k == 0ull) {
            k = atomicCAS(keys + s, 0ull, (unsigned long long)h);
            if (k == 0ull) {
                if (atomicAdd(count, 1) + 1 > PAT_MAX) *overflow = 1;
                k = h;
            }
        }
        if (k == h) {
            if ((int)i < *(volatile int *)(rep + s)) atomicMin(rep + s, (int)i);
            return;
        }
        if (*overflow) return;
        s = (s + 1) & mask;
    }
    *overflow = 1;
}

template <bool VALS>
__global__ void pat_assign_kernel(int64_t nrow, int64_t npad, int32_t W, const int32_t *__restrict__ col,
                                  const cplx *__restrict__ val, const unsigned long long *__restrict__ keys,
                                  const int *__restrict__ rep, const int *__restrict__ slot_id, uint16_t *__restrict__ pid,
                                  int *__restrict__ mismatch) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npad) return;
    if (i >= nrow) { pid[i] = 0; return; }
    const uint64_t h = pat_row_hash<VALS>(i, npad, W, col, val);
    int s = pat_find(h, keys);
    if (s < 0) { *mismatch = 1; pid[i] = 0; return; }
    const int64_t r = rep[s];
    bool same = true;
    for (int32_t w = 0; w < W; w++) {
        int64_t a = (int64_t)w * npad + i, b = (int64_t)w * npad + r;
        same = same && (col[a] - (int32_t)i) == (col[b] - (int32_t)r);
        if (VALS)
            same = same && __double_as_longlong(val[a].x) == __double_as_longlong(val[b].x) &&
                   __double_as_longlong(val[a].y) == __double_as_longlong(val[b].y);
    }
    if (!same) *mismatch = 1;
    pid[i] = (uint16_t)slot_id[s];
}

__global__ void pat_fill_kernel(int32_t npat, int32_t W, int64_t npad, const int *__restrict__ rep_row,
                                const int32_t *__restrict__ col, const cplx *__restrict__ val, int32_t *__restrict__ off,
                                double *__restrict__ re, double *__restrict__ im) {
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= npat * W) return;
    int id = t / W, w = t - id * W;
    int64_t r = rep_row[id];
    int64_t idx = (int64_t)w * npad + r;
    off[t] = col[idx] - (int32_t)r;
    if (re) { re[t] = val[idx].x; im[t] = val[idx].y; }
}

static int g_patterns = -1;
static bool patterns_enabled() {
    if (g_patterns < 0) g_patterns = !(getenv("MGCR_PATTERNS") && atoi(getenv("MGCR_PATTERNS")) == 0);
    return g_patterns != 0;
}
bool set_patterns_enabled(bool on) {
    bool prev = patterns_enabled();
    g_patterns = on ? 1 : 0;
    return prev;
}

// tries to build the dictionary for the (L = 1) slab of A; leaves A.pat_mode = 0 when it does not pay
template <bool VALS>
static int pat_try(CsrDev &A, bool *ok) {
    Context &c = ctx();
    *ok = false;
    const int T = 1 << PAT_TABLE_BITS;
    unsigned long long *d_keys = nullptr;
    int *d_rep = nullptr, *d_small = nullptr, *d_slot_id = nullptr, *d_rep_row = nullptr;
    std::vector<unsigned long long> keys((size_t)T);
    std::vector<int> rep((size_t)T), slot_id((size_t)T, 0), small(3, 0);
    int rc = MGCR_OK;
    auto done = [&](int r) {
        hipFree(d_keys); hipFree(d_rep); hipFree(d_small); hipFree(d_slot_id); hipFree(d_rep_row);
        return r;
    };
    MGCR_HIP(hipMalloc((void **)&d_keys, sizeof(unsigned long long) * T));
    if (hipMalloc((void **)&d_rep, sizeof(int) * T) != hipSuccess || hipMalloc((void **)&d_small, sizeof(int) * 3) != hipSuccess ||
        hipMalloc((void **)&d_slot_id, sizeof(int) * T) != hipSuccess)
        return done(MGCR_OK);  // no memory for the attempt: keep the plain slab
    hipMemsetAsync(d_keys, 0, sizeof(unsigned long long) * T, c.stream);
    hipMemsetAsync(d_rep, 0x7f, sizeof(int) * T, c.stream);
    hipMemsetAsync(d_small, 0, sizeof(int) * 3, c.stream);
    const unsigned grid = (unsigned)((A.nrow + 255) / 256), gridp = (unsigned)((A.npad + 255) / 256);
    hipLaunchKernelGGL((pat_insert_kernel<VALS>), dim3(grid), dim3(256), 0, c.stream, A.nrow, A.npad, A.W, (const int32_t *)A.ell_col,
                       (const cplx *)A.ell_val, d_keys, d_rep, d_small, d_small + 1);
    hipMemcpyAsync(small.data(), d_small, sizeof(int) * 3, hipMemcpyDeviceToHost, c.stream);
    MGCR_HIP(hipStreamSynchronize(c.stream));
    const int npat = small[0];
    // worth it only when the table is far smaller than the matrix
    if (small[1] || npat < 1 || npat > PAT_MAX || (int64_t)npat * 64 > A.nrow) return done(MGCR_OK);
    MGCR_HIP(hipMemcpy(keys.data(), d_keys, sizeof(unsigned long long) * T, hipMemcpyDeviceToHost));
    MGCR_HIP(hipMemcpy(rep.data(), d_rep, sizeof(int) * T, hipMemcpyDeviceToHost));
    // ids in the order of each pattern's first row: deterministic whatever order the inserts raced in
    std::vector<std::pair<int, int>> order;  // (first row, slot)
    for (int s = 0; s < T; s++)
        if (keys[(size_t)s]) order.emplace_back(rep[(size_t)s], s);
    std::sort(order.begin(), order.end());
    if ((int)order.size() != npat) return done(MGCR_OK);
    std::vector<int> rep_row((size_t)npat);
    for (int id = 0; id < npat; id++) { slot_id[(size_t)order[(size_t)id].second] = id; rep_row[(size_t)id] = order[(size_t)id].first; }
    if (hipMalloc((void **)&d_rep_row, sizeof(int) * npat) != hipSuccess) return done(MGCR_OK);
    MGCR_HIP(hipMemcpy(d_slot_id, slot_id.data(), sizeof(int) * T, hipMemcpyHostToDevice));
    MGCR_HIP(hipMemcpy(d_rep_row, rep_row.data(), sizeof(int) * npat, hipMemcpyHostToDevice));
    uint16_t *pid = nullptr;
    int32_t *off = nullptr;
    double *re = nullptr, *im = nullptr;
    bool alloc_ok = hipMalloc((void **)&pid, sizeof(uint16_t) * (size_t)A.npad) == hipSuccess &&
                    hipMalloc((void **)&off, sizeof(int32_t) * (size_t)npat * A.W) == hipSuccess;
    if (alloc_ok && VALS)
        alloc_ok = hipMalloc((void **)&re, sizeof(double) * (size_t)npat * A.W) == hipSuccess &&
                   hipMalloc((void **)&im, sizeof(double) * (size_t)npat * A.W) == hipSuccess;
    if (alloc_ok) {
        hipLaunchKernelGGL((pat_assign_kernel<VALS>), dim3(gridp), dim3(256), 0, c.stream, A.nrow, A.npad, A.W,
                           (const int32_t *)A.ell_col, (const cplx *)A.ell_val, (const unsigned long long *)d_keys, (const int *)d_rep,
                           (const int *)d_slot_id, pid, d_small + 2);
        hipLaunchKernelGGL(pat_fill_kernel, dim3((unsigned)((npat * A.W + 255) / 256)), dim3(256), 0, c.stream, npat, A.W, A.npad,
                           (const int *)d_rep_row, (const int32_t *)A.ell_col, (const cplx *)A.ell_val, off, re, im);
        hipMemcpyAsync(small.data(), d_small, sizeof(int) * 3, hipMemcpyDeviceToHost, c.stream);
        if (hipStreamSynchronize(c.stream) != hipSuccess || hipGetLastError() != hipSuccess) rc = MGCR_ERR_HIP;
    }
    if (!alloc_ok || rc != MGCR_OK || small[2]) {  // small[2]: two different rows shared a hash
        hipFree(pid); hipFree(off); hipFree(re); hipFree(im);
        if (rc != MGCR_OK) set_error("pattern dictionary kernels failed");
        return done(rc);
    }
    {   // how far the gathers of a row reach (decides the row -> workgroup map of the GCR step kernels, gcr_dev.h)
        std::vector<int32_t> h_off((size_t)npat * A.W);
        if (hipMemcpy(h_off.data(), off, sizeof(int32_t) * h_off.size(), hipMemcpyDeviceToHost) == hipSuccess)
            for (int32_t o : h_off) A.reach = std::max<int64_t>(A.reach, o < 0 ? -(int64_t)o : (int64_t)o);
    }
    A.pat_mode = VALS ? 1 : 2;
    A.npat = npat;
    A.pat_id = pid; A.pat_off = off; A.pat_re = re; A.pat_im = im;
    *ok = true;
    return done(MGCR_OK);
}


// ------------------------------------------------------------------------------------------------
// Stencil view of a mode-1 dictionary (CsrDev::sten_*; kernels: sten_spmv below, MODE 3 of the fused GCR step
// kernels).  The dictionary kernels are bound by a dependent chain per row — id -> table -> gathers, two memory round
// trips — not by bandwidth.  When all patterns are sub-stencils of one small stencil (their column offsets are
// subsequences of one ascending list of at most STEN_MAX offsets) and a slot's value is the same in every pattern that
// has it — the 7-point Poisson matrix, its Galerkin coarse operators, any constant-coefficient stencil with
// truncated boundaries — a row needs only to know WHICH slots it has: one bit per row and slot, stored as one 64-bit
// word per wave of 64 rows and slot and read through the scalar cache.  The x loads then depend on the row number
// alone (coalesced, wave-uniform offsets) and are in flight while the presence words arrive.  Measured on MI355X
// (tools/spmv_lab.hip, Poisson): 13.1 against 16.8 us at 128^3 back to back, 134 against 168 us at 256^3.
// Entries whose stored value is exactly 0 (the slab's padding) are treated as absent: they only ever add +-0.
// ------------------------------------------------------------------------------------------------
// (each wave walks several waves' worth of rows and keeps the per-slot row counts in lane 0's registers: one atomic per slot and
// WAVE OF THE GRID at the end.  One atomic per slot and 64 rows — 2.3 M of them on 16 addresses at 256^3 — took 25 ms.)
__global__ void __launch_bounds__(256) sten_planes_kernel(int64_t nrow, int64_t nwaves, int32_t ns, int32_t stride,
                                                          const uint16_t *__restrict__ pid, const uint16_t *__restrict__ pmask,
                                                          uint64_t *__restrict__ planes, unsigned long long *__restrict__ counts) {
    const int lane = threadIdx.x & 63;
    const int64_t gw = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 6, tw = (int64_t)gridDim.x * 4;
    unsigned long long cnt[16];
#pragma unroll
    for (int c = 0; c < 16; c++) cnt[c] = 0ull;
    for (int64_t wave = gw; wave < nwaves; wave += tw) {
        const int64_t row = wave * 64 + lane;
        const uint32_t m = row < nrow ? pmask[pid[row]] : 0u;
#pragma unroll
        for (int32_t c = 0; c < 16; c++) {
            if (c < stride) {
                const unsigned long long b = __ballot(c < ns && (m >> c & 1u));
                if (lane == 0) planes[wave * stride + c] = b;
                cnt[c] += (unsigned long long)__popcll(b);
            }
        }
    }
    if (lane == 0) {
#pragma unroll
        for (int c = 0; c < 16; c++)
            if (cnt[c]) atomicAdd(counts + c, cnt[c]);
    }
}

static int g_stencil = -1;
static bool stencil_enabled() {
    if (g_stencil < 0) g_stencil = !(getenv("MGCR_STENCIL") && atoi(getenv("MGCR_STENCIL")) == 0);
    return g_stencil != 0;
}
bool set_stencil_enabled(bool on) {
    bool prev = stencil_enabled();
    g_stencil = on ? 1 : 0;
    return prev;
}

bool csr_stencil_active(const CsrDev &A) { return A.sten_ns > 0 && stencil_enabled(); }

// builds the stencil view of A (pat_mode 1) when the dictionary has that shape; leaves A.sten_ns = 0 otherwise
static int sten_try(CsrDev &A) {
    Context &c = ctx();
    A.sten_ns = 0;
    if (A.pat_mode != 1 || A.npat < 1 || !stencil_enabled()) return MGCR_OK;
    const size_t ne = (size_t)A.npat * A.W;
    std::vector<int32_t> off(ne);
    std::vector<double> re(ne), im(ne);
    MGCR_HIP(hipMemcpy(off.data(), A.pat_off, sizeof(int32_t) * ne, hipMemcpyDeviceToHost));
    MGCR_HIP(hipMemcpy(re.data(), A.pat_re, sizeof(double) * ne, hipMemcpyDeviceToHost));
    MGCR_HIP(hipMemcpy(im.data(), A.pat_im, sizeof(double) * ne, hipMemcpyDeviceToHost));
    std::vector<int32_t> S;
    for (size_t e = 0; e < ne; e++)
        if (re[e] != 0. || im[e] != 0.) S.push_back(off[e]);
    std::sort(S.begin(), S.end());
    S.erase(std::unique(S.begin(), S.end()), S.end());
    const int ns = (int)S.size();
    if (ns < 1 || ns > STEN_MAX) return MGCR_OK;
    // per pattern: which of the ns slots it has (bit s = slot s of the ascending list)
    std::vector<uint16_t> pbits((size_t)A.npat, 0);
    std::vector<char> have((size_t)ns, 0);
    double sre[16] = {}, sim[16] = {};
    // The slot with the largest offset may come FIRST in the rows that have it: the halo column of a row block's first plane (rows
    // handed over in global column order — the neighbour below has the smallest global column and, as local column nloc + k, the
    // largest offset).  Such a LEADING slot is summed before the others (kernel slot 7 of the rare layout, RowMat::sten_pre), so the
    // row sum keeps its storage order.  lead_mode: 0 undecided, 1 leading, 2 in ascending position.
    int lead_mode = 0;
    for (int p = 0; p < A.npat; p++) {
        int last = -1, count = 0;
        bool lead_here = false;
        for (int32_t w = 0; w < A.W; w++) {
            const size_t e = (size_t)p * A.W + w;
            if (re[e] == 0. && im[e] == 0.) continue;
            const int s = (int)(std::lower_bound(S.begin(), S.end(), off[e]) - S.begin());
            if (count == 0 && s == ns - 1 && ns > 1) lead_here = true;      // (decided below, once the pattern is known to have more entries)
            else {
                if (s <= last) return MGCR_OK;   // a repeated or descending column: not a sub-stencil in storage order
                last = s;
                if (s == ns - 1 && ns > 1 && count > 0) {
                    if (lead_mode == 1) return MGCR_OK;
                    lead_mode = 2;
                }
            }
            count++;
            if (!have[(size_t)s]) { have[(size_t)s] = 1; sre[s] = re[e]; sim[s] = im[e]; }
            else if (memcmp(&sre[s], &re[e], sizeof(double)) || memcmp(&sim[s], &im[e], sizeof(double))) return MGCR_OK;  // value differs between patterns
            pbits[(size_t)p] |= (uint16_t)(1u << s);
        }
        if (lead_here && count > 1) {
            if (lead_mode == 2) return MGCR_OK;
            lead_mode = 1;
        }
    }
    const bool lead = lead_mode == 1;
    const int64_t nwaves = A.npad / 64;
    uint16_t *d_pmask = nullptr;
    unsigned long long *d_counts = nullptr;
    uint64_t *planes = nullptr;
    bool ok = hipMalloc((void **)&d_pmask, sizeof(uint16_t) * (size_t)A.npat) == hipSuccess &&
              hipMalloc((void **)&d_counts, sizeof(unsigned long long) * 16) == hipSuccess &&
              hipMalloc((void **)&planes, sizeof(uint64_t) * (size_t)(nwaves + 1) * 16) == hipSuccess;
    // presence words for the slot numbering `pm` (pattern -> mask), `stride` words per wave; counts[k] = rows that have slot k
    auto build = [&](const std::vector<uint16_t> &pm, int nslots, int32_t stride, std::vector<unsigned long long> &counts) -> bool {
        counts.assign(16, 0);
        bool g = hipMemcpyAsync(d_pmask, pm.data(), sizeof(uint16_t) * (size_t)A.npat, hipMemcpyHostToDevice, c.stream) == hipSuccess &&
                 hipMemsetAsync(d_counts, 0, sizeof(unsigned long long) * 16, c.stream) == hipSuccess &&
                 hipMemsetAsync(planes + (size_t)nwaves * stride, 0, sizeof(uint64_t) * stride, c.stream) == hipSuccess;
        if (g && nwaves) {
            const int64_t pg = (nwaves + 3) / 4;   // 4 waves per workgroup; at most 2048 workgroups, each wave then walks several
            hipLaunchKernelGGL(sten_planes_kernel, dim3((unsigned)(pg < 2048 ? pg : 2048)), dim3(256), 0, c.stream, A.nrow, nwaves, nslots,
                               stride, (const uint16_t *)A.pat_id, (const uint16_t *)d_pmask, planes, d_counts);
            g = hipGetLastError() == hipSuccess;
        }
        return g && hipMemcpyAsync(counts.data(), d_counts, sizeof(unsigned long long) * 16, hipMemcpyDeviceToHost, c.stream) == hipSuccess &&
               hipStreamSynchronize(c.stream) == hipSuccess;
    };
    std::vector<unsigned long long> counts;
    if (ok) ok = build(pbits, ns, 16, counts);   // first pass: how many rows have each slot
    // Layout the kernels read (spmv_dev.h sten_row_product).  Rare tail: the slots fewer than 1/16 of the rows have are the
    // LAST one or two of the list (halo columns of a slab's first / last plane: local column nloc + slot lies behind
    // every owned column) and at most 7 common ones remain: common slots -> 0..6, rare ones -> 7, 8.  Otherwise every slot
    // is treated as common, 7 or 9 of them.
    // A leading slot (above) always takes the rare layout: kernel slot 7, summed first; one rare slot behind the common ones may
    // then follow as slot 8.
    int nrare = 0;
    const int nsl = lead ? ns - 1 : ns;    // the slots in ascending position
    while (ok && nrare < (lead ? 1 : 2) && nrare < nsl - 1 && (int64_t)counts[(size_t)(nsl - 1 - nrare)] * 16 < A.nrow) nrare++;
    if (lead && nrare == 0 && nsl == STEN_COMMON + 1) nrare = 1;   // (the upper halo column of a block of few planes: not rare by count, but the ninth slot)
    bool tail = (nrare > 0 || lead) && nsl - nrare <= STEN_COMMON;
    for (int s = 0; ok && tail && !lead && s < nsl - nrare; s++)
        if ((int64_t)counts[(size_t)s] * 16 < A.nrow) tail = false;   // a rare slot among the common ones: no special treatment
    if (lead && !tail) {   // more than 7 slots besides the leading one: no view (the dictionary kernels keep the storage order)
        hipFree(d_pmask); hipFree(d_counts); hipFree(planes);
        return MGCR_OK;
    }
    const bool force = ns <= STEN_COMMON && getenv("MGCR_TEST_FORCE_RARE") && atoi(getenv("MGCR_TEST_FORCE_RARE")) != 0;
    if (force) { tail = true; nrare = 0; }   // measurement aid: a single-GPU operator through the kernels of a distributed row block
    int slot_of[16];
    for (int s = 0; s < ns; s++) {
        if (lead && s == ns - 1) slot_of[s] = STEN_COMMON;                                           // summed first
        else if (tail && s >= nsl - nrare) slot_of[s] = STEN_COMMON + (lead ? 1 : 0) + (s - (nsl - nrare));
        else slot_of[s] = s;
    }
    const int kernel_ns = tail ? 9 : ns <= 7 ? 7 : 9;
    const int32_t stride = kernel_ns == 7 ? 8 : 16;
    if (ok) {
        std::vector<uint16_t> pm((size_t)A.npat, 0);
        for (int p = 0; p < A.npat; p++)
            for (int s = 0; s < ns; s++)
                if (pbits[(size_t)p] >> s & 1u) pm[(size_t)p] |= (uint16_t)(1u << slot_of[s]);
        ok = build(pm, kernel_ns, stride, counts);
    }
    hipFree(d_pmask); hipFree(d_counts);
    if (!ok) {  // no memory for the view: the dictionary kernels stay
        (void)hipGetLastError();
        hipFree(planes);
        return MGCR_OK;
    }
    A.sten_ns = ns;
    A.sten_kernel_ns = kernel_ns;
    A.sten_stride = stride;
    A.sten_planes = planes;
    A.sten_rare = tail ? 3u << STEN_COMMON : 0u;
    A.sten_pre = lead ? 1 : 0;
    for (int k = 0; k < 16; k++) { A.sten_off[k] = 0; A.sten_re[k] = 0.; A.sten_im[k] = 0.; }
    for (int s = 0; s < ns; s++) {
        A.sten_off[slot_of[s]] = S[(size_t)s];
        A.sten_re[slot_of[s]] = sre[s];
        A.sten_im[slot_of[s]] = sim[s];
    }
    // slots close to the diagonal (|offset| <= STEN_TILE / 2, e.g. +-1 and +-n of a 3-D grid up to n = 256) are read by
    // several rows of the same workgroup: the stand-alone kernel stages x once in an LDS window and serves them from there
    A.sten_near = 0;
    A.sten_halo = 0;
    for (int s = 0; s < ns; s++) {
        if (tail && slot_of[s] >= STEN_COMMON) continue;
        const int32_t a = S[(size_t)s] < 0 ? -S[(size_t)s] : S[(size_t)s];
        if (a <= STEN_TILE / 2) { A.sten_near |= 1u << slot_of[s]; A.sten_halo = std::max(A.sten_halo, a); }
    }
    if (A.sten_halo < 32) { A.sten_near = 0; A.sten_halo = 0; }   // only +-1-like neighbours: L1 serves those as well
    A.sten_near_f = 0;
    A.sten_halo_f = 0;
    for (int s = 0; s < ns; s++) {
        if (tail && slot_of[s] >= STEN_COMMON) continue;
        const int32_t a = S[(size_t)s] < 0 ? -S[(size_t)s] : S[(size_t)s];
        if (a <= RED_THREADS / 2) { A.sten_near_f |= 1u << slot_of[s]; A.sten_halo_f = std::max(A.sten_halo_f, a); }
    }
    if (A.sten_halo_f < 32) { A.sten_near_f = 0; A.sten_halo_f = 0; }
    if (tail) {
        // how far a row's gathers reach decides the row -> workgroup map of the GCR step kernels (gcr_dev.h): the two
        // rare slots (halo columns, "nloc rows away") concern one plane each and must not count
        A.reach = 0;
        for (int s = 0; s < ns; s++)
            if (slot_of[s] < STEN_COMMON) A.reach = std::max<int64_t>(A.reach, S[(size_t)s] < 0 ? -(int64_t)S[(size_t)s] : (int64_t)S[(size_t)s]);
    }
    return MGCR_OK;
}

// How local are the slab's columns?  (count of slots within 1024 / 4096 rows of their row)
__global__ void __launch_bounds__(256) ell_band_count_kernel(int64_t nrow, int64_t npad, int32_t W, const int32_t *__restrict__ col,
                                                             unsigned long long *__restrict__ cnt) {
    const int64_t row = (int64_t)blockIdx.x * 256 + threadIdx.x;
    unsigned long long c1 = 0, c4 = 0;
    if (row < nrow)
        for (int32_t w = 0; w < W; w++) {
            const int64_t d = (int64_t)col[(int64_t)w * npad + row] - row;
            const int64_t a = d < 0 ? -d : d;
            c1 += a <= 1024;
            c4 += a <= 4096;
        }
    // wave totals, then one atomic per wave and counter
    for (int off = 32; off >= 1; off >>= 1) {
        c1 += __shfl_down(c1, off, 64);
        c4 += __shfl_down(c4, off, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(cnt, c1);
        atomicAdd(cnt + 1, c4);
    }
}
constexpr int ELL_WIN_ROWS = 1024;   // rows per workgroup of the window kernel
static bool ell_window_enabled() {
    static const bool on = !(getenv("MGCR_ELL_WINDOW") && atoi(getenv("MGCR_ELL_WINDOW")) == 0);
    return on;
}
// Banded irregular matrices (FEM / graph matrices in a bandwidth-reducing order): the gathers of x, one L2 request of 128 B per
// 16-byte entry, are what bounds the slab kernel (profiles/r03_gather_lab.txt); with >= 90 % of the columns within H rows of the
// row the window kernel reads x[tile - H, tile + 1024 + H) once, coalesced, into LDS and gathers from there.
static int ell_window_try(CsrDev &A) {
    A.win_h = 0;
    if (!ell_window_enabled()) return MGCR_OK;
    unsigned long long *d_cnt = nullptr, h_cnt[2] = {0, 0};
    MGCR_HIP(hipMalloc((void **)&d_cnt, 2 * sizeof(unsigned long long)));
    MGCR_HIP(hipMemsetAsync(d_cnt, 0, 2 * sizeof(unsigned long long), ctx().stream));
    hipLaunchKernelGGL(ell_band_count_kernel, dim3((unsigned)((A.nrow + 255) / 256)), dim3(256), 0, ctx().stream, A.nrow, A.npad, A.W,
                       (const int32_t *)A.ell_col, d_cnt);
    MGCR_HIP(hipGetLastError());
    MGCR_HIP(hipMemcpyAsync(h_cnt, d_cnt, sizeof(h_cnt), hipMemcpyDeviceToHost, ctx().stream));
    MGCR_HIP(hipStreamSynchronize(ctx().stream));
    hipFree(d_cnt);
    const double slots = (double)A.nrow * A.W;
    if ((double)h_cnt[0] >= 0.9 * slots) A.win_h = 1024;
    else if ((double)h_cnt[1] >= 0.9 * slots) A.win_h = 4096;
    return MGCR_OK;
}

// device CSR (already resident) + host row pointers -> CsrDev
static bool real_storage_enabled() {
    static const bool on = !(getenv("MGCR_REAL_STORAGE") && atoi(getenv("MGCR_REAL_STORAGE")) == 0);
    return on;
}

static int ell_from_device_csr(int64_t nrow, int64_t ncol, const int64_t *h_rowptr, const int64_t *d_rowptr,
                               const int64_t *d_col, const cplx *d_val, CsrDev *out) {
    Context &c = ctx();
    CsrDev A;
    A.nrow = nrow; A.ncol = ncol; A.nnz = h_rowptr[nrow];
    int64_t maxlen64 = 0;
    for (int64_t r = 0; r < nrow; r++) maxlen64 = std::max(maxlen64, h_rowptr[r + 1] - h_rowptr[r]);
    MGCR_CHECK(maxlen64 < ((int64_t)1 << 30), MGCR_ERR_UNSUPPORTED, "row with %lld entries", (long long)maxlen64);
    int32_t maxlen = (int32_t)maxlen64;
    std::vector<int64_t> hist((size_t)maxlen + 2, 0);
    for (int64_t r = 0; r < nrow; r++) hist[(size_t)(h_rowptr[r + 1] - h_rowptr[r])]++;
    A.W = choose_width(hist, nrow, maxlen);
    A.L = choose_lanes(nrow, A.W);
    A.nchunk = (A.W + A.L - 1) / A.L;
    A.npad = (nrow + 63) / 64 * 64;
    // tail lists
    std::vector<int32_t> trows, tptr(1, 0);
    for (int64_t r = 0; r < nrow; r++) {
        int64_t len = h_rowptr[r + 1] - h_rowptr[r];
        if (len > A.W) {
            trows.push_back((int32_t)r);
            int64_t nxt = (int64_t)tptr.back() + (len - A.W);
            MGCR_CHECK(nxt < ((int64_t)1 << 31), MGCR_ERR_UNSUPPORTED, "CSR tail exceeds 2^31 entries");
            tptr.push_back((int32_t)nxt);
        }
    }
    A.n_tail_rows = (int64_t)trows.size();
    A.tail_nnz = tptr.back();

    size_t slab = (size_t)A.nchunk * (size_t)A.npad * (size_t)A.L;
    if (slab) {
        hipError_t e1 = hipMalloc((void **)&A.ell_val, sizeof(cplx) * slab);
        hipError_t e2 = hipMalloc((void **)&A.ell_col, sizeof(int32_t) * slab);
        if (e1 != hipSuccess || e2 != hipSuccess) {
            csr_free(&A);
            set_error("hipMalloc of the ELL slab (%zu entries) failed", slab);
            return MGCR_ERR_ALLOC;
        }
        int64_t threads = A.npad * A.L;
        hipLaunchKernelGGL(ell_fill_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, c.stream, nrow, ncol,
                           d_rowptr, d_col, d_val, A.W, A.L, A.nchunk, A.npad, A.ell_val, A.ell_col);
        MGCR_HIP(hipGetLastError());
    }
    std::vector<int4> chunks;
    std::vector<int32_t> long_rows;
    if (A.n_tail_rows) {
        // deal the tail rows to workgroups: runs of consecutive tail rows of at most TAIL_CAP entries / TAIL_THREADS rows
        int32_t i = 0;
        const int32_t nt = (int32_t)trows.size();
        while (i < nt) {
            if (tptr[(size_t)i + 1] - tptr[(size_t)i] > TAIL_CAP) { long_rows.push_back(i); i++; continue; }
            const int32_t first = i;
            const int32_t e0 = tptr[(size_t)i];
            while (i < nt && i - first < TAIL_THREADS && tptr[(size_t)i + 1] - e0 <= TAIL_CAP) i++;
            chunks.push_back(make_int4(first, i, e0, tptr[(size_t)i]));
        }
        A.n_tail_chunks = (int32_t)chunks.size();
        A.n_tail_long = (int32_t)long_rows.size();
        if (A.n_tail_chunks) MGCR_TRY(dev_upload(&A.tail_chunk, chunks.data(), chunks.size()));
        if (A.n_tail_long) MGCR_TRY(dev_upload(&A.tail_long, long_rows.data(), long_rows.size()));
    }
    if (A.n_tail_rows) {
        MGCR_TRY(dev_upload(&A.tail_rows, trows.data(), trows.size()));
        MGCR_TRY(dev_upload(&A.tail_ptr, tptr.data(), tptr.size()));
        MGCR_TRY(dev_upload<int32_t>(&A.tail_col, nullptr, (size_t)A.tail_nnz));
        MGCR_TRY(dev_upload<cplx>(&A.tail_val, nullptr, (size_t)A.tail_nnz));
        int64_t threads = A.n_tail_rows * 64;
        hipLaunchKernelGGL(tail_fill_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, c.stream,
                           A.n_tail_rows, A.tail_rows, A.tail_ptr, d_rowptr, d_col, d_val, A.W, A.tail_col, A.tail_val);
        MGCR_HIP(hipGetLastError());
    }
    MGCR_HIP(hipStreamSynchronize(c.stream));  // trows/tptr go out of scope
    int has_imag = 1;
    if (slab && A.nnz > 0) {
        int *d_flag = nullptr;
        MGCR_HIP(hipMalloc((void **)&d_flag, sizeof(int)));
        MGCR_HIP(hipMemsetAsync(d_flag, 0, sizeof(int), c.stream));
        hipLaunchKernelGGL(imag_check_kernel, dim3((unsigned)((A.nnz + 255) / 256)), dim3(256), 0, c.stream, A.nnz, d_val, d_flag);
        MGCR_HIP(hipMemcpyAsync(&has_imag, d_flag, sizeof(int), hipMemcpyDeviceToHost, c.stream));
        MGCR_HIP(hipStreamSynchronize(c.stream));
        hipFree(d_flag);
    }
    if (slab && A.L == 1 && A.W >= 1 && A.W <= 32 && A.nrow >= PAT_MIN_ROWS && patterns_enabled()) {
        bool ok = false;
        MGCR_TRY(pat_try<true>(A, &ok));
        if (ok) {  // the table holds everything: no slab
            A.pat_real = !has_imag;
            hipFree(A.ell_val); hipFree(A.ell_col);
            A.ell_val = nullptr; A.ell_col = nullptr;
            MGCR_TRY(sten_try(A));
            *out = A;
            return MGCR_OK;
        }
        MGCR_TRY(pat_try<false>(A, &ok));
        if (ok) { hipFree(A.ell_col); A.ell_col = nullptr; }
    }
    if (slab && A.L == 1 && A.pat_mode == 0 && A.nrow == A.ncol && A.nrow >= ELL_WIN_ROWS && A.W >= 2) MGCR_TRY(ell_window_try(A));
    if (A.win_h && A.n_tail_rows) {
        // per tile: its first tail row; per row: its index into the tail-row list (-1: none, or longer than a chunk)
        const int64_t ntiles = (A.nrow + ELL_WIN_ROWS - 1) / ELL_WIN_ROWS;
        std::vector<int32_t> tile_tail((size_t)ntiles + 1, (int32_t)trows.size()), row_tail((size_t)A.nrow, -1);
        for (int32_t t = (int32_t)trows.size() - 1; t >= 0; t--) {
            tile_tail[(size_t)(trows[(size_t)t] / ELL_WIN_ROWS)] = t;
            if (tptr[(size_t)t + 1] - tptr[(size_t)t] <= TAIL_CAP) row_tail[(size_t)trows[(size_t)t]] = t;
        }
        for (int64_t q = ntiles - 1; q >= 0; q--)   // tiles without tail rows: the next tile's first
            if (tile_tail[(size_t)q] > tile_tail[(size_t)q + 1]) tile_tail[(size_t)q] = tile_tail[(size_t)q + 1];
        MGCR_TRY(dev_upload(&A.win_tile_tail, tile_tail.data(), tile_tail.size()));
        MGCR_TRY(dev_upload(&A.win_row_tail, row_tail.data(), row_tail.size()));
        MGCR_HIP(hipStreamSynchronize(c.stream));
    }
    // Real matrices (every imaginary part exactly 0, e.g. Poisson): keep the slab's values as fp64
    // reals, 12 B instead of 20 B per stored entry.  v*(c+di) with v real is (vc, vd): the same numbers
    // the complex product (vc - 0*d, vd + 0*c) gives for finite x.
    if (slab && A.nnz > 0 && real_storage_enabled()) {
        if (!has_imag) {
            hipError_t e = hipMalloc((void **)&A.ell_val_re, sizeof(double) * slab);
            if (e == hipSuccess) {
                hipLaunchKernelGGL(slab_real_kernel, dim3((unsigned)((slab + 255) / 256)), dim3(256), 0, c.stream, (int64_t)slab,
                                   (const cplx *)A.ell_val, A.ell_val_re);
                MGCR_HIP(hipStreamSynchronize(c.stream));
                hipFree(A.ell_val);
                A.ell_val = nullptr;
            }
        }
    }
    *out = A;
    return MGCR_OK;
}

// device-resident CSR (int64 indices) -> ELL + tail; h_rowptr is the host copy of the row pointers
int csr_build_from_device(int64_t nrow, int64_t ncol, const int64_t *h_rowptr, const int64_t *d_rowptr, const int64_t *d_col,
                          const cplx *d_val, CsrDev *out) {
    return ell_from_device_csr(nrow, ncol, h_rowptr, d_rowptr, d_col, d_val, out);
}

int csr_build_device(int64_t nrow, int64_t ncol, const int64_t *h_rowptr, const int64_t *h_col, const double *h_val_ri,
                     CsrDev *out) {
    Context &c = ctx();
    MGCR_CHECK(nrow >= 0 && ncol >= 0 && nrow < ((int64_t)1 << 31) && ncol < ((int64_t)1 << 31), MGCR_ERR_UNSUPPORTED,
               "matrix dimensions must fit int32 per GPU (got %lld x %lld)", (long long)nrow, (long long)ncol);
    MGCR_CHECK(h_rowptr[0] == 0, MGCR_ERR_INVALID, "rowptr[0] must be 0");
    for (int64_t r = 0; r < nrow; r++)
        MGCR_CHECK(h_rowptr[r + 1] >= h_rowptr[r], MGCR_ERR_INVALID, "rowptr is not non-decreasing at row %lld", (long long)r);
    int64_t nnz = h_rowptr[nrow];
    int64_t *d_rowptr = nullptr, *d_col = nullptr;
    cplx *d_val = nullptr;
    int *d_bad = nullptr;
    int rc = dev_upload(&d_rowptr, h_rowptr, (size_t)nrow + 1);
    if (rc == MGCR_OK) rc = dev_upload(&d_col, h_col, (size_t)nnz);
    if (rc == MGCR_OK) rc = dev_upload(&d_val, (const cplx *)h_val_ri, (size_t)nnz);
    int bad = 0;
    if (rc == MGCR_OK && nnz > 0) {
        rc = dev_upload(&d_bad, &bad, 1);
        if (rc == MGCR_OK) {
            hipLaunchKernelGGL(col_check_kernel, dim3((unsigned)((nnz + 255) / 256)), dim3(256), 0, c.stream, nnz, d_col, ncol, d_bad);
            hipMemcpyAsync(&bad, d_bad, sizeof(int), hipMemcpyDeviceToHost, c.stream);
            hipStreamSynchronize(c.stream);
            if (bad) { set_error("mgcr_csr_create: a column index is outside [0, ncol)"); rc = MGCR_ERR_INVALID; }
        }
    }
    if (rc == MGCR_OK) rc = ell_from_device_csr(nrow, ncol, h_rowptr, d_rowptr, d_col, d_val, out);
    hipStreamSynchronize(c.stream);
    hipFree(d_rowptr); hipFree(d_col); hipFree(d_val); hipFree(d_bad);
    return rc;
}

// ------------------------------------------------------------------------------------------------
// SpMV
// ------------------------------------------------------------------------------------------------
// has the solve this apply belongs to stopped on the device?  (skip = {stop_at, base}, gcr.hip DevState; null: stand-alone apply.)
// Read through the constant address space: scalar loads wherever the call stands.
__device__ __forceinline__ bool stop_flag(const int *skip, int skip_it) {
    typedef const int __attribute__((address_space(4))) *stop_ptr;
    const stop_ptr sk = (stop_ptr)(uintptr_t)skip;
    return skip && sk[0] < sk[1] + skip_it;
}

template <int WT, bool SHIFT, bool XCD, bool REALV, bool NT>
__global__ void __launch_bounds__(256) ell_spmv_rowthread(int64_t row_begin, int64_t row_count, int64_t npad, int32_t Wrt,
                                                          int64_t ntiles, const void *__restrict__ val,
                                                          const int32_t *__restrict__ col, const cplx *__restrict__ x,
                                                          const cplx *__restrict__ xh, int32_t n_own,
                                                          cplx *__restrict__ y, cplx k, const cplx *__restrict__ w, const int *__restrict__ skip, int skip_it) {
    // One scalar batch for every argument (pinned by the empty asm) instead of one dependent fetch per early exit: a wave
    // lives a few microseconds and each dependent scalar round trip in front of its first load costs ~0.2 us of that.  The
    // solver's stop flag (skip = {stop_at, base}: see gcr.hip DevState) is looked at while the loads fly and only gates the store.
    asm volatile("" ::"s"(row_begin), "s"(row_count), "s"(npad), "s"(Wrt), "s"(ntiles), "s"(val), "s"(col), "s"(x), "s"(xh), "s"(n_own), "s"(y),
                 "s"(w), "s"(skip), "s"(skip_it));
    int64_t tile = XCD ? xcd_tile(ntiles) : (int64_t)blockIdx.x;
    if (tile >= ntiles) return;
    int64_t rloc = tile * 256 + threadIdx.x;
    if (rloc >= row_count) return;
    int64_t row = row_begin + rloc;
    const int32_t W = WT ? WT : Wrt;
    cplx sum = make_double2(0., 0.);
    bool stopped = false;
    if (WT) {
        int32_t j[WT ? WT : 1];
        cplx xv[WT ? WT : 1];
#pragma unroll
        for (int32_t c = 0; c < W; c++) j[c] = ldcol<NT>(col + (int64_t)c * npad + row);
        stopped = stop_flag(skip, skip_it);
#pragma unroll
        for (int32_t c = 0; c < W; c++) xv[c] = gather_x(x, xh, n_own, j[c]);
#pragma unroll
        for (int32_t c = 0; c < W; c++) sum = cadd(sum, vmul<REALV, NT>(val, (int64_t)c * npad + row, xv[c]));
    } else {
        stopped = stop_flag(skip, skip_it);
#pragma unroll 4
        for (int32_t c = 0; c < W; c++) {
            int32_t j = ldcol<NT>(col + (int64_t)c * npad + row);
            sum = cadd(sum, vmul<REALV, NT>(val, (int64_t)c * npad + row, gather_x(x, xh, n_own, j)));
        }
    }
    if (!stopped) y[row] = SHIFT ? csub((w ? w : x)[row], cmul(k, sum)) : sum;
}

// ELL slab, one thread per row, x window in LDS (banded irregular matrices: CsrDev::win_h).  A workgroup owns 1024 consecutive
// rows; x[r0 - H, r0 + 1024 + H) is read once, coalesced, into LDS and every column inside it is served from there — the few
// outside take the global gather.  Same products, added in the same (CSR) order as ell_spmv_rowthread: same bits.
// TAIL: the CSR tail of the tile's rows in the SAME launch.  The tail entries of consecutive rows are one contiguous piece of the
// tail arrays: the workgroup streams it 1024 entries at a time (coalesced), gathers x from the window it already holds, stages the
// products in LDS, and every thread adds the products of ITS row in CSR order onto a tail sum; y = (ELL sum) + (tail sum) is
// written once — csr_tail_chunk_kernel's arithmetic and order (same bits) without the second launch, the second gather of x from
// memory and the read-modify-write of y.  Rows whose tail is longer than a chunk (TAIL_CAP) are left to csr_tail_kernel.
constexpr int WIN_TAIL_PER = 2;                           // tail entries per thread and trip
constexpr int WIN_TAIL_CH = WIN_TAIL_PER * ELL_WIN_ROWS;   // tail entries staged per trip (32 KB of products)
template <bool SHIFT, bool REALV, int H, bool TAIL>
__global__ void __launch_bounds__(ELL_WIN_ROWS) ell_spmv_window(int64_t nrow, int64_t npad, int32_t W, int64_t ntiles, const void *__restrict__ val,
                                                                const int32_t *__restrict__ col, const cplx *__restrict__ x, cplx *__restrict__ y,
                                                                cplx k, const cplx *__restrict__ w, const int *__restrict__ skip, int skip_it,
                                                                const int32_t *__restrict__ tile_tail, const int32_t *__restrict__ row_tail,
                                                                const int32_t *__restrict__ tail_ptr, const int32_t *__restrict__ tail_col,
                                                                const cplx *__restrict__ tail_val) {
    extern __shared__ __attribute__((aligned(16))) unsigned char win_smem[];
    cplx *win = reinterpret_cast<cplx *>(win_smem);
    constexpr int WLEN = ELL_WIN_ROWS + 2 * H;
    cplx *prod = win + WLEN;   // [WIN_TAIL_CH] (TAIL only)
    const int64_t tile = ntiles >= 64 ? xcd_tile(ntiles) : (int64_t)blockIdx.x;
    if (tile >= ntiles) return;
    const int64_t r0 = tile * ELL_WIN_ROWS, base = r0 - H;
    const int64_t row = r0 + threadIdx.x;
    const bool live = row < nrow;
    // the row's first columns are requested together with the window (they do not depend on it)
    int32_t j0 = 0, j1 = 0;
    if (live) {
        j0 = ldcol<true>(col + row);
        if (W > 1) j1 = ldcol<true>(col + npad + row);
    }
    int32_t t0 = 0, t1 = 0, ti = -1;
    if (TAIL) {
        t0 = tile_tail[tile];
        t1 = tile_tail[tile + 1];
        if (live) ti = row_tail[row];
    }
    for (int t = threadIdx.x; t < WLEN; t += ELL_WIN_ROWS) {
        int64_t g = base + t;
        g = g < 0 ? 0 : g >= nrow ? nrow - 1 : g;
        win[t] = x[g];
    }
    const bool stopped = stop_flag(skip, skip_it);
    __syncthreads();
    if (stopped) return;   // (uniform)
    auto xat = [&](int32_t j) -> cplx {
        const int64_t off = (int64_t)j - base;
        return (off >= 0 && off < WLEN) ? win[off] : x[j];
    };
    cplx sum = make_double2(0., 0.);
    if (live) {
        sum = cadd(sum, vmul<REALV, true>(val, row, xat(j0)));
        if (W > 1) sum = cadd(sum, vmul<REALV, true>(val, npad + row, xat(j1)));
#pragma unroll 4
        for (int32_t c = 2; c < W; c++) {
            const int32_t j = ldcol<true>(col + (int64_t)c * npad + row);
            sum = cadd(sum, vmul<REALV, true>(val, (int64_t)c * npad + row, xat(j)));
        }
    }
    if (!TAIL) {
        if (live) y[row] = SHIFT ? csub((w ? w : x)[row], cmul(k, sum)) : sum;
        return;
    }
    const int32_t e0 = tail_ptr[t0], e1 = tail_ptr[t1];
    int32_t rb = 0, re = 0;
    if (ti >= 0) { rb = tail_ptr[ti]; re = tail_ptr[ti + 1]; }
    cplx tsum = make_double2(0., 0.);
    // the next trip's columns and values are requested before this trip's products are formed and summed: their latency hides
    // behind the LDS phase (two workgroups per CU do not hide it by themselves)
    int32_t jn[WIN_TAIL_PER];
    cplx vn[WIN_TAIL_PER];
    auto fetch = [&](int32_t cb) {
#pragma unroll
        for (int q = 0; q < WIN_TAIL_PER; q++) {
            const int32_t e = cb + q * ELL_WIN_ROWS + (int32_t)threadIdx.x;
            jn[q] = -1;
            vn[q] = make_double2(0., 0.);
            if (e < e1) {
                jn[q] = tail_col[e];
                vn[q] = make_double2(__builtin_nontemporal_load(&tail_val[e].x), __builtin_nontemporal_load(&tail_val[e].y));
            }
        }
    };
    if (e0 < e1) fetch(e0);
    for (int32_t cb = e0; cb < e1; cb += WIN_TAIL_CH) {   // uniform trip count
        int32_t jc[WIN_TAIL_PER];
        cplx vc[WIN_TAIL_PER];
#pragma unroll
        for (int q = 0; q < WIN_TAIL_PER; q++) { jc[q] = jn[q]; vc[q] = vn[q]; }
        if (cb + WIN_TAIL_CH < e1) fetch(cb + WIN_TAIL_CH);
#pragma unroll
        for (int q = 0; q < WIN_TAIL_PER; q++)
            if (jc[q] >= 0) prod[q * ELL_WIN_ROWS + threadIdx.x] = cmul(vc[q], xat(jc[q]));
        __syncthreads();
        const int32_t ib = (rb > cb ? rb : cb) - cb, ie = (re < cb + WIN_TAIL_CH ? re : cb + WIN_TAIL_CH) - cb;
        {   // this row's products of the trip, in CSR order (four independent LDS reads in flight)
            int32_t i = ib;
            for (; i + 4 <= ie; i += 4) {
                const cplx p0 = prod[i], p1 = prod[i + 1], p2 = prod[i + 2], p3 = prod[i + 3];
                tsum = cadd(cadd(cadd(cadd(tsum, p0), p1), p2), p3);
            }
            for (; i < ie; i++) tsum = cadd(tsum, prod[i]);
        }
        __syncthreads();
    }
    if (live) {
        if (SHIFT) {
            cplx v = csub((w ? w : x)[row], cmul(k, sum));
            if (ti >= 0) v = csub(v, cmul(k, tsum));
            y[row] = v;
        } else {
            y[row] = ti >= 0 ? cadd(sum, tsum) : sum;
        }
    }
}

// Row-pattern dictionary SpMV (L = 1): one thread per row; the row's 2-byte id selects the table row
// holding its W column offsets (and, MODE 1, its W values).  Interior rows of a wave share one id, so
// the table loads are single-line broadcasts out of L1.  Same multiply/add order as ell_spmv_rowthread.
template <int WT, bool SHIFT, bool XCD, int MODE, bool REALV>
__global__ void __launch_bounds__(256) pat_spmv_rowthread(int64_t row_begin, int64_t row_count, int64_t npad, int32_t Wrt,
                                                          int64_t ntiles, const uint16_t *__restrict__ pid,
                                                          const int32_t *__restrict__ poff, const double *__restrict__ pre,
                                                          const double *__restrict__ pim, const void *__restrict__ val,
                                                          const cplx *__restrict__ x, const cplx *__restrict__ xh, int32_t n_own,
                                                          cplx *__restrict__ y, cplx k, const cplx *__restrict__ w, const int *__restrict__ skip, int skip_it) {
    if (skip && skip[0] < skip[1] + skip_it) return;  // {stop_at, base}: see gcr.hip DevState
    int64_t tile = XCD ? xcd_tile(ntiles) : (int64_t)blockIdx.x;
    if (tile >= ntiles) return;
    int64_t rloc = tile * 256 + threadIdx.x;
    if (rloc >= row_count) return;
    const int64_t row = row_begin + rloc;
    const int32_t W = WT ? WT : Wrt;
    const int32_t t0 = (int32_t)__builtin_nontemporal_load(pid + row) * W;
    cplx sum = make_double2(0., 0.);
    auto term = [&](int32_t c, cplx xv) -> cplx {
        if (MODE == 1) {
            if (REALV) {
                double v = pre[t0 + c];
                return make_double2(v * xv.x, v * xv.y);
            }
            return cmul(make_double2(pre[t0 + c], pim[t0 + c]), xv);
        }
        return vmul<REALV, true>(val, (int64_t)c * npad + row, xv);
    };
    if (WT) {
        cplx xv[WT ? WT : 1];
#pragma unroll
        for (int32_t c = 0; c < W; c++) xv[c] = gather_x(x, xh, n_own, (int32_t)row + poff[t0 + c]);
#pragma unroll
        for (int32_t c = 0; c < W; c++) sum = cadd(sum, term(c, xv[c]));
    } else {
#pragma unroll 4
        for (int32_t c = 0; c < W; c++) sum = cadd(sum, term(c, gather_x(x, xh, n_own, (int32_t)row + poff[t0 + c])));
    }
    y[row] = SHIFT ? csub((w ? w : x)[row], cmul(k, sum)) : sum;
}

// Same, with the pattern table staged in LDS (one dependent memory round trip less per row: id -> LDS ->
// gathers; 17.6 against 20.7 us at Poisson 128^3).  MODE 1 only.  R rows per thread with all id loads and
// gathers in flight together was measured too (R = 2, 4; 256 / 512 threads): no faster than R = 1, warm or
// cold; nor was staging the workgroup's own 256..1024 entries of x in LDS to serve the +-1 / +-n columns
// (26 against 25 us cold) — PMC: TA busy 61 %, L2 hit rate 0.61, 90 % of wave cycles waiting; the kernel
// moves 71 MB in 17 (warm) .. 25 us (cold caches) where a 67 MB copy takes 11.4 us.
template <int WT, bool SHIFT, bool REALV, int R, int BLK>
__global__ void __launch_bounds__(BLK) pat_spmv_lds(int64_t row_begin, int64_t row_count, int32_t Wrt, int64_t ntiles, int xcd,
                                                    int32_t npat, const uint16_t *__restrict__ pid,
                                                    const int32_t *__restrict__ poff, const double *__restrict__ pre,
                                                    const double *__restrict__ pim, const cplx *__restrict__ x,
                                                    const cplx *__restrict__ xh, int32_t n_own, cplx *__restrict__ y, cplx k,
                                                    const cplx *__restrict__ w, const int *__restrict__ skip, int skip_it) {
    extern __shared__ __attribute__((aligned(16))) unsigned char pat_smem[];
    if (skip && skip[0] < skip[1] + skip_it) return;  // {stop_at, base}: see gcr.hip DevState
    const int64_t tile = xcd ? xcd_tile(ntiles) : (int64_t)blockIdx.x;
    if (tile >= ntiles) return;
    const int32_t W = WT ? WT : Wrt;
    const int32_t ne = npat * W;
    double *sre = reinterpret_cast<double *>(pat_smem);
    double *sim = sre + (REALV ? 0 : ne);
    int32_t *soff = reinterpret_cast<int32_t *>(sim + ne);
    int64_t row[R];
    bool live[R];
    int32_t t0[R];
#pragma unroll
    for (int r = 0; r < R; r++) {
        const int64_t rloc = tile * (BLK * R) + r * BLK + threadIdx.x;
        live[r] = rloc < row_count;
        row[r] = row_begin + (live[r] ? rloc : 0);
        t0[r] = (int32_t)__builtin_nontemporal_load(pid + row[r]) * W;
    }
    for (int32_t e = threadIdx.x; e < ne; e += BLK) {
        soff[e] = poff[e];
        sre[e] = pre[e];
        if (!REALV) sim[e] = pim[e];
    }
    __syncthreads();
    if (WT) {
        cplx xv[R][WT ? WT : 1];
#pragma unroll
        for (int r = 0; r < R; r++)
#pragma unroll
            for (int32_t c = 0; c < W; c++) xv[r][c] = gather_x(x, xh, n_own, (int32_t)row[r] + soff[t0[r] + c]);
#pragma unroll
        for (int r = 0; r < R; r++) {
            cplx sum = make_double2(0., 0.);
#pragma unroll
            for (int32_t c = 0; c < W; c++) {
                cplx t;
                if (REALV) { double v = sre[t0[r] + c]; t = make_double2(v * xv[r][c].x, v * xv[r][c].y); }
                else t = cmul(make_double2(sre[t0[r] + c], sim[t0[r] + c]), xv[r][c]);
                sum = cadd(sum, t);
            }
            if (live[r]) y[row[r]] = SHIFT ? csub((w ? w : x)[row[r]], cmul(k, sum)) : sum;
        }
    } else {
#pragma unroll
        for (int r = 0; r < R; r++) {
            cplx sum = make_double2(0., 0.);
#pragma unroll 4
            for (int32_t c = 0; c < W; c++) {
                cplx xv = gather_x(x, xh, n_own, (int32_t)row[r] + soff[t0[r] + c]);
                cplx t;
                if (REALV) { double v = sre[t0[r] + c]; t = make_double2(v * xv.x, v * xv.y); }
                else t = cmul(make_double2(sre[t0[r] + c], sim[t0[r] + c]), xv);
                sum = cadd(sum, t);
            }
            if (live[r]) y[row[r]] = SHIFT ? csub((w ? w : x)[row[r]], cmul(k, sum)) : sum;
        }
    }
}

// Stencil view SpMV (MODE 3, spmv_dev.h sten_row_product): BLK consecutive rows per workgroup, waves aligned to
// multiples of 64 rows (`first` = row_begin rounded down), one memory round trip per row.
template <int NS, bool RARE, bool SHIFT, int BLK>
__global__ void __launch_bounds__(BLK) sten_spmv(RowMat m, int64_t row_begin, int64_t row_end, int64_t first, int64_t ntiles, int xcd,
                                                 const cplx *__restrict__ x, cplx *__restrict__ y, const cplx *__restrict__ w,
                                                 const int *__restrict__ skip, int skip_it) {
    if (skip && skip[0] < skip[1] + skip_it) return;  // {stop_at, base}: see gcr.hip DevState
    const int64_t tile = xcd ? xcd_tile(ntiles) : (int64_t)blockIdx.x;
    if (tile >= ntiles) return;
    const int64_t rloc = first + tile * BLK + threadIdx.x;
    if ((rloc | 63) < row_begin || (rloc & ~(int64_t)63) >= row_end) return;   // the whole wave lies outside
    const bool live = rloc >= row_begin && rloc < row_end;
    const cplx sum = sten_row_product<NS, RARE>(m, rloc, [&](int32_t j) -> cplx { return gather_x(x, m.xh, m.n_own, j); });
    if (live) y[rloc] = SHIFT ? csub((w ? w : x)[rloc], cmul(m.k, sum)) : sum;
}

// The same with an LDS window: the workgroup's BLK entries of x plus sten_halo entries on either side are staged once
// (one coalesced load per thread, the halo by the first 2 * sten_halo threads) and the slots close to the diagonal
// (sten_near: +-1, +-n of a grid) are read from there; the far slots (+-n^2) are requested before the barrier.  A
// 7-point row then costs ~3.5 loads through L1 / L2 instead of 7 — tools/spmv_lab.hip on MI355X, Poisson 256^3: 130
// against 151 us with cold caches (the +-n neighbours, which the neighbouring workgroups fetch at the same time, are
// the expensive ones), 128^3: 20.4 against 21.9 us.  Same slot order, same selects: same bits.  NEAR is a template
// parameter (the kernel exists for the mask of a 3-D stencil, slots 1..5 of 7): with a run-time mask the compiler keeps
// the gathered values in scratch memory and waits for every load in turn — 4x slower than no window at all.
template <int NS, bool RARE, bool SHIFT, int BLK, unsigned NEAR, bool DMA>
__global__ void __launch_bounds__(BLK) sten_spmv_tile(RowMat m, int64_t row_begin, int64_t row_end, int64_t first, int64_t ntiles, int xcd,
                                                      const cplx *__restrict__ x, cplx *__restrict__ y, const cplx *__restrict__ w,
                                                      const int *__restrict__ skip, int skip_it) {
    extern __shared__ __attribute__((aligned(16))) unsigned char sten_smem[];
    constexpr int NC = RARE ? STEN_COMMON : NS;   // rare-tail layout: slots NC.. are looked at after the common sum
    // A wave of this kernel lives ~3 us and every DEPENDENT scalar fetch in front of its gathers costs ~0.2 us of that (measured:
    // one kernel argument fetched late = +6 % kernel time).  So: every argument the gathers need is fetched in ONE batch (the
    // empty asm pins them), the gathers go out, and only then come the scalar loads from memory — presence words, the solver's
    // stop flag (skip = {stop_at, base}, gcr.hip DevState), whose answers nobody needs before the window is filled.
    const int32_t H = m.sten_halo, last = m.sten_last, n_own = m.n_own, nwaves = m.sten_nwaves, pstride = m.sten_stride;
    const cplx *const xh = m.xh;
    const uint64_t *const planes = m.sten_planes;
    int32_t off[NC];
#pragma unroll
    for (int c = 0; c < NC; c++) off[c] = m.sten_off[c];
    asm volatile("" ::"s"(H), "s"(last), "s"(n_own), "s"(nwaves), "s"(pstride), "s"(xh), "s"(planes), "s"(x), "s"(first), "s"(ntiles), "s"(xcd),
                 "s"(row_begin), "s"(row_end), "s"(skip), "s"(skip_it), "s"(y));
    const int realv = m.realv;
    double re[NC];
#pragma unroll
    for (int c = 0; c < NC; c++) {
        re[c] = m.sten_re[c];
        asm volatile("" ::"s"(off[c]), "s"(re[c]));
    }
    asm volatile("" ::"s"(realv));
    const int64_t tile = xcd ? xcd_tile(ntiles) : (int64_t)blockIdx.x;
    if (tile >= ntiles) return;
    cplx *sx = reinterpret_cast<cplx *>(sten_smem);   // [H + BLK + H], entry e = column base - H + e
    const int64_t base = first + tile * BLK;
    const int64_t rloc = base + threadIdx.x;
    const bool live = rloc >= row_begin && rloc < row_end;
    auto clampj = [&](int64_t j) -> int32_t { return (int32_t)(j < 0 ? 0 : j > last ? last : j); };
    cplx xv[NC];
#pragma unroll
    for (int c = 0; c < NC; c++) {
        xv[c] = make_double2(0., 0.);
        if (!(NEAR >> c & 1u)) xv[c] = gather_x(x, xh, n_own, clampj(rloc + off[c]));
    }
    cplx own = make_double2(0., 0.), halo = make_double2(0., 0.);
    int hidx = -1;
    if constexpr (DMA) {
        // the window is filled by the memory system itself (gfx950 global_load_lds_dwordx4: 16 bytes per lane, a wave's 64 entries
        // land contiguously at the LDS address in M0): no registers, no ds_write between the data's arrival and the barrier —
        // 14.6-15.0 -> 14.3 us back to back at 128^3, 127-128 -> 122-123 us at 256^3 (cold: 22.0-22.6 -> 21.4, 131-132 -> 125-127).
        // (H is a multiple of 64 here: a wave of halo threads lies entirely left or entirely right of the tile.  The same fill in
        // the solver's windowed step kernels, gcr_fused.hip, LOSES 2 %: they need the row's own entry in registers afterwards
        // and the per-lane addresses cost what the data registers saved — measured, not kept.)
        const int wv = (int)(threadIdx.x >> 6) * 64;
        const int32_t jo = clampj(rloc);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(jo < n_own ? x + jo : xh + (jo - n_own)),
                                         (__attribute__((address_space(3))) void *)(sx + H + wv), 16, 0, 0);
        if ((int)threadIdx.x < 2 * H) {
            const int t = (int)threadIdx.x;
            const int32_t jh = clampj(t < H ? base - H + t : base + BLK + (t - H));
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(jh < n_own ? x + jh : xh + (jh - n_own)),
                                             (__attribute__((address_space(3))) void *)(sx + (t < H ? 0 : BLK) + wv), 16, 0, 0);
        }
    } else {
        own = gather_x(x, xh, n_own, clampj(rloc));
        if ((int)threadIdx.x < 2 * H) {
            const int t = (int)threadIdx.x;
            halo = gather_x(x, xh, n_own, clampj(t < H ? base - H + t : base + BLK + (t - H)));
            hidx = t < H ? t : BLK + t;
        }
    }
    __builtin_amdgcn_sched_barrier(0);   // every gather is in flight before anything else is asked for
    // presence words of this wave (rows beyond the padded end of the matrix have none: the planes array ends with a zero row)
    int32_t wave = __builtin_amdgcn_readfirstlane((int32_t)(rloc >> 6));
    wave = wave < nwaves ? wave : nwaves;
    const sten_planes_ptr pp = (sten_planes_ptr)(uintptr_t)(planes + (int64_t)wave * pstride);
    uint64_t pl[NS];
#pragma unroll
    for (int c = 0; c < NS; c++) pl[c] = pp[c];
    const bool stopped = stop_flag(skip, skip_it);
    __builtin_amdgcn_sched_barrier(0);   // every load is in flight before the first one is waited for
    if constexpr (DMA) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's part of the window has landed
    } else {
        sx[H + threadIdx.x] = own;
        if (hidx >= 0) sx[hidx] = halo;
    }
    __syncthreads();
    const int lane = (int)(threadIdx.x & 63);
    cplx sum = make_double2(0., 0.);
    if constexpr (RARE) sum = sten_pre_sum<-1>(m, rloc, pl[NC], lane, [&](int32_t j) -> cplx { return gather_x(x, xh, n_own, j); });
    // (the real / complex decision once, not per slot: per slot it put a scalar fetch and a branch between every two terms)
    if (realv) {
#pragma unroll
        for (int c = 0; c < NC; c++) {
            // a window entry outside the matrix is a clamped copy: only read by rows whose presence bit for the slot is clear
            const cplx v = (NEAR >> c & 1u) ? sx[H + (int)threadIdx.x + off[c]] : xv[c];
            const bool on = (pl[c] >> lane & 1ull) != 0ull;
            const cplx nsum = cadd(sum, make_double2(re[c] * v.x, re[c] * v.y));
            sum.x = on ? nsum.x : sum.x;
            sum.y = on ? nsum.y : sum.y;
        }
    } else {
#pragma unroll
        for (int c = 0; c < NC; c++) {
            const cplx v = (NEAR >> c & 1u) ? sx[H + (int)threadIdx.x + off[c]] : xv[c];
            const bool on = (pl[c] >> lane & 1ull) != 0ull;
            const cplx nsum = cadd(sum, cmul(make_double2(re[c], m.sten_im[c]), v));
            sum.x = on ? nsum.x : sum.x;
            sum.y = on ? nsum.y : sum.y;
        }
    }
    if (RARE) {
#pragma unroll
        for (int c = NC; c < NS; c++)
            if (pl[c] != 0ull && !(c == NC && m.sten_pre)) {   // wave-uniform: a wave of a boundary plane
                const cplx xr = gather_x(x, xh, n_own, clampj(rloc + m.sten_off[c]));
                const bool on = (pl[c] >> lane & 1ull) != 0ull;
                const cplx t = m.realv ? make_double2(m.sten_re[c] * xr.x, m.sten_re[c] * xr.y) : cmul(make_double2(m.sten_re[c], m.sten_im[c]), xr);
                const cplx nsum = cadd(sum, t);
                sum.x = on ? nsum.x : sum.x;
                sum.y = on ? nsum.y : sum.y;
            }
    }
    if (live && !stopped) y[rloc] = SHIFT ? csub((w ? w : x)[rloc], cmul(m.k, sum)) : sum;
}

// L in {2,4,8,16}: L consecutive lanes share a row; per chunk the (row, lane) pairs are contiguous
template <int L, bool SHIFT, bool REALV>
__global__ void __launch_bounds__(256) ell_spmv_lanes(int64_t row_begin, int64_t row_count, int64_t npad, int32_t nchunk,
                                                      const void *__restrict__ val, const int32_t *__restrict__ col,
                                                      const cplx *__restrict__ x, const cplx *__restrict__ xh, int32_t n_own,
                                                      cplx *__restrict__ y, cplx k, const cplx *__restrict__ w, const int *__restrict__ skip, int skip_it) {
    if (skip && skip[0] < skip[1] + skip_it) return;  // {stop_at, base}: see gcr.hip DevState
    int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    int64_t rloc = t / L;
    int64_t row = row_begin + rloc;
    int64_t nrow = row_begin + row_count;
    int l = (int)(t % L);
    cplx sum = make_double2(0., 0.);
    if (row < nrow) {
#pragma unroll 4
        for (int32_t c = 0; c < nchunk; c++) {
            int64_t idx = ((int64_t)c * npad + row) * L + l;
            sum = cadd(sum, vmul<REALV>(val, idx, gather_x(x, xh, n_own, col[idx])));
        }
    }
#pragma unroll
    for (int off = L / 2; off >= 1; off >>= 1) {
        sum.x += __shfl_down(sum.x, off, L);
        sum.y += __shfl_down(sum.y, off, L);
    }
    if (row < nrow && l == 0) y[row] = SHIFT ? csub((w ? w : x)[row], cmul(k, sum)) : sum;
}

// CSR tail, rows longer than a chunk on their own (> TAIL_CAP entries): one wave per row, lanes stride the entries, wave64 tree
template <bool SHIFT>
__global__ void __launch_bounds__(256) csr_tail_kernel(int64_t n_long, const int32_t *__restrict__ tail_long, const int32_t *__restrict__ tail_rows,
                                                       const int32_t *__restrict__ tail_ptr,
                                                       const int32_t *__restrict__ tail_col,
                                                       const cplx *__restrict__ tail_val, const cplx *__restrict__ x,
                                                       const cplx *__restrict__ xh, int32_t n_own,
                                                       cplx *__restrict__ y, cplx k, const cplx *__restrict__ w, const int *__restrict__ skip, int skip_it) {
    if (skip && skip[0] < skip[1] + skip_it) return;  // {stop_at, base}: see gcr.hip DevState
    int64_t wave = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 6;
    int lane = threadIdx.x & 63;
    if (wave >= n_long) return;
    const int32_t t = tail_long[wave];
    int32_t beg = tail_ptr[t], end = tail_ptr[t + 1];
    cplx sum = make_double2(0., 0.);
    for (int32_t i = beg + lane; i < end; i += 64) sum = cadd(sum, cmul(tail_val[i], gather_x(x, xh, n_own, tail_col[i])));
    sum.x = wave_sum(sum.x);
    sum.y = wave_sum(sum.y);
    if (lane == 0) {
        int32_t row = tail_rows[t];
        y[row] = SHIFT ? csub(y[row], cmul(k, sum)) : cadd(y[row], sum);
    }
}

// CSR tail, everything else: one workgroup per CHUNK — a run of consecutive tail rows with at most TAIL_CAP entries and
// TAIL_THREADS rows (dealt at build time, ell_from_device_csr).  The chunk's entries are one contiguous piece of the tail
// arrays: all threads stream it with coalesced loads (4 entries per thread in flight: columns, values, the gathers of x),
// the products val * x are staged in LDS, and thread t then adds row t's products in CSR order — the reference's order
// (src/Operator.h:338-341) — onto the row's ELL sum.  Against one wave per row (rows of 1..55 entries: 1.7 M waves that each
// fetch half-used lines and spend their life in three dependent memory round trips) this streams whole lines once and keeps
// 32 waves per CU busy: 8 M-row skewed matrix (bench.py irregular_spmv), tail part: 0.69 ms -> see DESIGN.md.
template <bool SHIFT>
__global__ void __launch_bounds__(TAIL_THREADS) csr_tail_chunk_kernel(const int4 *__restrict__ chunks,
                                                                      const int32_t *__restrict__ tail_rows, const int32_t *__restrict__ tail_ptr,
                                                                      const int32_t *__restrict__ tail_col, const cplx *__restrict__ tail_val,
                                                                      const cplx *__restrict__ x, const cplx *__restrict__ xh, int32_t n_own,
                                                                      cplx *__restrict__ y, cplx k, const int *__restrict__ skip, int skip_it) {
    __shared__ cplx prod[TAIL_CAP];
    if (skip && skip[0] < skip[1] + skip_it) return;
    const int t = threadIdx.x;
    const int4 ch = chunks[blockIdx.x];   // one record: the entry range does not wait for a second, dependent fetch
    const int32_t r0 = ch.x, r1 = ch.y, e0 = ch.z, e1 = ch.w;
    // this thread's row (phase 2): requested now, used after the barrier
    int32_t rb = 0, re = 0, row = 0;
    if (r0 + t < r1) { rb = tail_ptr[r0 + t]; re = tail_ptr[r0 + t + 1]; row = tail_rows[r0 + t]; }
    constexpr int PER = TAIL_CAP / TAIL_THREADS;
    int32_t j[PER];
    cplx v[PER], xv[PER];
#pragma unroll
    for (int q = 0; q < PER; q++) {
        const int32_t e = e0 + q * TAIL_THREADS + t;
        j[q] = e < e1 ? tail_col[e] : -1;
    }
#pragma unroll
    for (int q = 0; q < PER; q++) {
        const int32_t e = e0 + q * TAIL_THREADS + t;
        v[q] = e < e1 ? make_double2(__builtin_nontemporal_load(&tail_val[e].x), __builtin_nontemporal_load(&tail_val[e].y)) : make_double2(0., 0.);
        xv[q] = j[q] >= 0 ? gather_x(x, xh, n_own, j[q]) : make_double2(0., 0.);
    }
#pragma unroll
    for (int q = 0; q < PER; q++) prod[q * TAIL_THREADS + t] = cmul(v[q], xv[q]);
    __syncthreads();
    if (r0 + t < r1) {
        const cplx y0 = y[row];     // (requested before the LDS walk)
        cplx sum = make_double2(0., 0.);
        int32_t i = rb - e0;
        const int32_t ie = re - e0;
        for (; i + 4 <= ie; i += 4) {   // four independent LDS reads in flight, added in CSR order
            const cplx p0 = prod[i], p1 = prod[i + 1], p2 = prod[i + 2], p3 = prod[i + 3];
            sum = cadd(cadd(cadd(cadd(sum, p0), p1), p2), p3);
        }
        for (; i < ie; i++) sum = cadd(sum, prod[i]);
        y[row] = SHIFT ? csub(y0, cmul(k, sum)) : cadd(y0, sum);
    }
}

// measurement aid (bench.py: the ELL part and the CSR tail of a hybrid matrix timed separately): 0 = the whole apply,
// 1 = only the ELL slab's kernel, 2 = only the tail kernel (which then adds to whatever y holds).  mgcr_set_option("spmv_part").
static int g_spmv_part = 0;
int set_spmv_part(int part) {
    const int prev = g_spmv_part;
    g_spmv_part = part < 0 || part > 2 ? 0 : part;
    return prev;
}

// does the window kernel of A also multiply the chunk-sized tails (one launch for slab + tail)?  Not while bench.py times the
// two parts apart (spmv_part), not for the row block of a distributed matrix (halo columns live outside x)
static bool window_fuses_tail(const CsrDev &A) {
    static const bool on = !(getenv("MGCR_ELL_WINDOW_TAIL") && atoi(getenv("MGCR_ELL_WINDOW_TAIL")) == 0);
    // (H = 4096: the window alone takes 144 KB — with the products' 32 KB there is no room, and with 16 KB it measured slower than two launches)
    return on && A.win_h == 1024 && A.win_tile_tail && A.win_row_tail && g_spmv_part == 0;
}

static SkipRef g_skip;  // consulted by apply kernels (set by the GCR driver around its operator applies)
void set_apply_skip(SkipRef s) { g_skip = s; }
SkipRef get_apply_skip() { return g_skip; }

// rows [row_begin, row_begin + row_count) of the ELL part
template <bool SHIFT>
static int ell_rows(const CsrDev &A, int64_t row_begin, int64_t row_count, const cplx *x, const cplx *xh, int32_t n_own, cplx *y, cplx k,
                    const cplx *w) {
    Context &c = ctx();
    if (row_count <= 0) return MGCR_OK;
    if (csr_stencil_active(A)) {
        static const bool tile_on = !(getenv("MGCR_STENCIL_TILE") && atoi(getenv("MGCR_STENCIL_TILE")) == 0);
        // window variants: 512 rows + halo <= 256 (grids up to n = 256), or 1024 rows + halo <= 512 when that catches
        // near slots the smaller one cannot (+-n of planes up to 512 wide)
        const bool big = tile_on && A.sten_near_f == 0x3eu && A.sten_halo_f > 0 && (A.sten_near != 0x3eu || A.sten_halo == 0);
        const bool small = tile_on && !big && A.sten_near == 0x3eu && A.sten_halo > 0;
        const int BLKr = big ? RED_THREADS : STEN_TILE;
        const int64_t first = row_begin & ~(int64_t)63;
        const int64_t ntiles = (row_begin + row_count - first + BLKr - 1) / BLKr;
        const bool xcd = ntiles >= 64;
        const unsigned grid = (unsigned)(xcd ? ((ntiles + 7) / 8) * 8 : ntiles);
        RowMat m = row_mat(A, SHIFT, k);
        m.xh = xh; m.n_own = n_own;
#define SL(NS, RARE)                                                                                                      \
    hipLaunchKernelGGL((sten_spmv<NS, RARE, SHIFT, STEN_TILE>), dim3(grid), dim3(STEN_TILE), 0, c.stream, m, row_begin, row_begin + row_count, \
                       first, ntiles, xcd ? 1 : 0, x, y, w, g_skip.p, g_skip.it)
        static const bool dma_on = !(getenv("MGCR_STENCIL_DMA") && atoi(getenv("MGCR_STENCIL_DMA")) == 0);
#define SLT(NS, RARE, BLK, HH)                                                                                            \
    do {                                                                                                                  \
        m.sten_halo = (HH);                                                                                               \
        if (dma_on && (HH) % 64 == 0)                                                                                     \
            hipLaunchKernelGGL((sten_spmv_tile<NS, RARE, SHIFT, BLK, 0x3eu, true>), dim3(grid), dim3(BLK), (size_t)(BLK + 2 * (HH)) * sizeof(cplx), \
                               c.stream, m, row_begin, row_begin + row_count, first, ntiles, xcd ? 1 : 0, x, y, w, g_skip.p, g_skip.it); \
        else                                                                                                              \
            hipLaunchKernelGGL((sten_spmv_tile<NS, RARE, SHIFT, BLK, 0x3eu, false>), dim3(grid), dim3(BLK), (size_t)(BLK + 2 * (HH)) * sizeof(cplx), \
                               c.stream, m, row_begin, row_begin + row_count, first, ntiles, xcd ? 1 : 0, x, y, w, g_skip.p, g_skip.it); \
    } while (0)
        if (big) {
            if (A.sten_rare) SLT(9, true, RED_THREADS, A.sten_halo_f);
            else if (sten_slots(A) == 7) SLT(7, false, RED_THREADS, A.sten_halo_f);
            else SLT(9, false, RED_THREADS, A.sten_halo_f);
        } else if (small) {
            if (A.sten_rare) SLT(9, true, STEN_TILE, A.sten_halo);
            else if (sten_slots(A) == 7) SLT(7, false, STEN_TILE, A.sten_halo);
            else SLT(9, false, STEN_TILE, A.sten_halo);
        } else if (A.sten_rare) SL(9, true);
        else if (sten_slots(A) == 7) SL(7, false);
        else SL(9, false);
#undef SLT
#undef SL
        MGCR_HIP(hipGetLastError());
        return MGCR_OK;
    }
    if (A.pat_mode == 1 && (int64_t)A.npat * A.W * 20 <= 48 * 1024) {  // pattern table fits LDS
        const int64_t ntiles = (row_count + 255) / 256;
        const bool xcd = ntiles >= 64;
        const unsigned grid = (unsigned)(xcd ? ((ntiles + 7) / 8) * 8 : ntiles);
        const size_t lds = (size_t)A.npat * A.W * (A.pat_real ? 12 : 20);
#define PL(WT, RV)                                                                                                            \
    hipLaunchKernelGGL((pat_spmv_lds<WT, SHIFT, RV, 1, 256>), dim3(grid), dim3(256), lds, c.stream, row_begin, row_count, A.W, \
                       ntiles, xcd ? 1 : 0, A.npat, (const uint16_t *)A.pat_id, (const int32_t *)A.pat_off,                   \
                       (const double *)A.pat_re, (const double *)A.pat_im, x, xh, n_own, y, k, w, g_skip.p, g_skip.it)
#define PL_V(WT) do { if (A.pat_real) PL(WT, true); else PL(WT, false); } while (0)
        if (A.W == 7) PL_V(7); else PL_V(0);
#undef PL_V
#undef PL
        MGCR_HIP(hipGetLastError());
        return MGCR_OK;
    }
    if (A.pat_mode) {
        int64_t ntiles = (row_count + 255) / 256;
        bool xcd = ntiles >= 64;
        unsigned grid = (unsigned)(xcd ? ((ntiles + 7) / 8) * 8 : ntiles);
        const void *vals = A.ell_val_re ? (const void *)A.ell_val_re : (const void *)A.ell_val;
        const bool realv = A.pat_mode == 1 ? A.pat_real : A.ell_val_re != nullptr;
#define PT(WT, X, M, RV)                                                                                                       \
    hipLaunchKernelGGL((pat_spmv_rowthread<WT, SHIFT, X, M, RV>), dim3(grid), dim3(256), 0, c.stream, row_begin, row_count,    \
                       A.npad, A.W, ntiles, (const uint16_t *)A.pat_id, (const int32_t *)A.pat_off, (const double *)A.pat_re, \
                       (const double *)A.pat_im, vals, x, xh, n_own, y, k, w, g_skip.p, g_skip.it)
#define PT_RV(WT, X, M) do { if (realv) PT(WT, X, M, true); else PT(WT, X, M, false); } while (0)
#define PT_X(WT, M) do { if (xcd) PT_RV(WT, true, M); else PT_RV(WT, false, M); } while (0)
        if (A.pat_mode == 1) { if (A.W == 7) PT_X(7, 1); else PT_X(0, 1); }
        else { if (A.W == 7) PT_X(7, 2); else PT_X(0, 2); }
#undef PT_X
#undef PT_RV
#undef PT
        MGCR_HIP(hipGetLastError());
        return MGCR_OK;
    }
    if (A.L == 1 && A.win_h && !xh && row_begin == 0 && row_count == A.nrow && !A.pat_mode) {
        const int64_t ntiles = (A.nrow + ELL_WIN_ROWS - 1) / ELL_WIN_ROWS;
        const unsigned grid = (unsigned)(ntiles >= 64 ? ((ntiles + 7) / 8) * 8 : ntiles);
        const void *vals = A.ell_val_re ? (const void *)A.ell_val_re : (const void *)A.ell_val;
        const bool tail = window_fuses_tail(A);
        const size_t lds = sizeof(cplx) * (size_t)(ELL_WIN_ROWS + 2 * A.win_h + (tail ? WIN_TAIL_CH : 0));
#define WK(RV, HH, TL)                                                                                                                \
    do {                                                                                                                              \
        static bool big_lds = false;                                                                                                  \
        if (!big_lds) {                                                                                                               \
            MGCR_HIP(hipFuncSetAttribute((const void *)ell_spmv_window<SHIFT, RV, HH, TL>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); \
            big_lds = true;                                                                                                           \
        }                                                                                                                             \
        hipLaunchKernelGGL((ell_spmv_window<SHIFT, RV, HH, TL>), dim3(grid), dim3(ELL_WIN_ROWS), lds, c.stream, A.nrow, A.npad, A.W, ntiles, vals, \
                           (const int32_t *)A.ell_col, x, y, k, w, g_skip.p, g_skip.it, (const int32_t *)A.win_tile_tail,              \
                           (const int32_t *)A.win_row_tail, (const int32_t *)A.tail_ptr, (const int32_t *)A.tail_col, (const cplx *)A.tail_val); \
    } while (0)
#define WK_T(RV, HH) do { if (tail) WK(RV, HH, true); else WK(RV, HH, false); } while (0)
        if (A.win_h == 1024) { if (A.ell_val_re) WK_T(true, 1024); else WK_T(false, 1024); }
        else { if (A.ell_val_re) WK_T(true, 4096); else WK_T(false, 4096); }
#undef WK_T
#undef WK
        MGCR_HIP(hipGetLastError());
        return MGCR_OK;
    }
    if (A.L == 1) {
        int64_t ntiles = (row_count + 255) / 256;
        bool xcd = ntiles >= 64;
        unsigned grid = (unsigned)(xcd ? ((ntiles + 7) / 8) * 8 : ntiles);
        const void *vals = A.ell_val_re ? (const void *)A.ell_val_re : (const void *)A.ell_val;
        // on by default (+3 % GCR iterations/s at 128^3, measured); MGCR_SPMV_NT=0 turns it off
        static const bool nt = !(getenv("MGCR_SPMV_NT") && atoi(getenv("MGCR_SPMV_NT")) == 0);
#define RT(WT, X, RV)                                                                                                            \
    do {                                                                                                                         \
        if (nt)                                                                                                                  \
            hipLaunchKernelGGL((ell_spmv_rowthread<WT, SHIFT, X, RV, true>), dim3(grid), dim3(256), 0, c.stream, row_begin,      \
                               row_count, A.npad, A.W, ntiles, vals, A.ell_col, x, xh, n_own, y, k, w, g_skip.p, g_skip.it);        \
        else                                                                                                                     \
            hipLaunchKernelGGL((ell_spmv_rowthread<WT, SHIFT, X, RV, false>), dim3(grid), dim3(256), 0, c.stream, row_begin,     \
                               row_count, A.npad, A.W, ntiles, vals, A.ell_col, x, xh, n_own, y, k, w, g_skip.p, g_skip.it);        \
    } while (0)
        if (A.ell_val_re) {
            if (A.W == 7) { if (xcd) RT(7, true, true); else RT(7, false, true); }
            else { if (xcd) RT(0, true, true); else RT(0, false, true); }
        } else {
            if (A.W == 7) { if (xcd) RT(7, true, false); else RT(7, false, false); }
            else { if (xcd) RT(0, true, false); else RT(0, false, false); }
        }
#undef RT
    } else {
        int64_t threads = row_count * A.L;
        unsigned grid = (unsigned)((threads + 255) / 256);
        const void *vals = A.ell_val_re ? (const void *)A.ell_val_re : (const void *)A.ell_val;
#define LN(LL)                                                                                                                      \
    do {                                                                                                                            \
        if (A.ell_val_re)                                                                                                           \
            hipLaunchKernelGGL((ell_spmv_lanes<LL, SHIFT, true>), dim3(grid), dim3(256), 0, c.stream, row_begin, row_count, A.npad, \
                               A.nchunk, vals, A.ell_col, x, xh, n_own, y, k, w, g_skip.p, g_skip.it);                                 \
        else                                                                                                                        \
            hipLaunchKernelGGL((ell_spmv_lanes<LL, SHIFT, false>), dim3(grid), dim3(256), 0, c.stream, row_begin, row_count, A.npad, \
                               A.nchunk, vals, A.ell_col, x, xh, n_own, y, k, w, g_skip.p, g_skip.it);                                 \
    } while (0)
        switch (A.L) {
            case 2: LN(2); break;
            case 4: LN(4); break;
            case 8: LN(8); break;
            default: LN(16); break;
        }
#undef LN
    }
    MGCR_HIP(hipGetLastError());
    return MGCR_OK;
}

template <bool SHIFT>
static int csr_apply_t(const CsrDev &A, const cplx *x, cplx *y, cplx k, DistCsr *dist, const cplx *w) {
    Context &c = ctx();
    if (A.nrow == 0) return MGCR_OK;
    const cplx *xh = nullptr;
    int32_t n_own = INT32_MAX;
    if (dist) {
        // halo exchange on the communication stream, overlapped with the rows that need no halo
        int64_t ib = 0, ie = 0;
        dist_info(dist, &xh, &ib, &ie);
        n_own = (int32_t)A.nrow;
        MGCR_TRY(dist_halo_begin(dist, x, ie > ib && g_spmv_part == 0));
        xh = dist_halo_ptr(dist);
        MGCR_TRY(ell_rows<SHIFT>(A, ib, ie - ib, x, xh, n_own, y, k, w));
        MGCR_TRY(dist_halo_end(dist));
        MGCR_TRY(ell_rows<SHIFT>(A, 0, ib, x, xh, n_own, y, k, w));
        MGCR_TRY(ell_rows<SHIFT>(A, ie, A.nrow - ie, x, xh, n_own, y, k, w));
    } else if (g_spmv_part != 2) {
        // 256 x 256 x Z grids: the carried-window form (gcr_fused.hip) — every entry of x requested once, no far gathers
        if (g_spmv_part == 0 && csr_stencil_active(A) && csr_apply_carry(A, x, y, SHIFT, k, w)) return MGCR_OK;
        MGCR_TRY(ell_rows<SHIFT>(A, 0, A.nrow, x, xh, n_own, y, k, w));
    }
    if (A.n_tail_rows && g_spmv_part != 1) {
        if (A.n_tail_chunks && !(window_fuses_tail(A) && !dist)) {
            hipLaunchKernelGGL((csr_tail_chunk_kernel<SHIFT>), dim3((unsigned)A.n_tail_chunks), dim3(TAIL_THREADS), 0, c.stream, (const int4 *)A.tail_chunk,
                               A.tail_rows, A.tail_ptr, A.tail_col, A.tail_val, x, xh, n_own, y, k, g_skip.p, g_skip.it);
            MGCR_HIP(hipGetLastError());
        }
        if (A.n_tail_long) {
            int64_t threads = (int64_t)A.n_tail_long * 64;
            hipLaunchKernelGGL((csr_tail_kernel<SHIFT>), dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, c.stream, (int64_t)A.n_tail_long,
                               A.tail_long, A.tail_rows, A.tail_ptr, A.tail_col, A.tail_val, x, xh, n_own, y, k, w, g_skip.p, g_skip.it);
            MGCR_HIP(hipGetLastError());
        }
    }
    return MGCR_OK;
}

int csr_apply(const CsrDev &A, const cplx *x, cplx *y, bool shift, cplx k, DistCsr *dist, const cplx *w) {
    MGCR_CHECK(x != y, MGCR_ERR_INVALID, "SpMV cannot run in place");
    MGCR_CHECK(!w || (shift && w != y), MGCR_ERR_INVALID, "csr_apply: w needs the shifted form and its own storage");
    return shift ? csr_apply_t<true>(A, x, y, k, dist, w) : csr_apply_t<false>(A, x, y, k, dist, nullptr);
}

// ------------------------------------------------------------------------------------------------
// block-CSR of dense bs x bs blocks (HierarchicalSparse).  One wave per block-row.  Per block the
// wave streams the bs*bs entries with coalesced 16-B loads, stages the products m[r][c]*x[c] in
// LDS, and lanes r < bs then add their row's products in column order — the order of
// Dense::operator() (src/Operator.h:165-170) — onto the block-row accumulator
// (value += ..., src/HierarchicalSparse.h:144).
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64) bcsr_wave_kernel(int32_t nbrow, int32_t bs, const int32_t *__restrict__ browptr,
                                                       const int32_t *__restrict__ bcol, const cplx *__restrict__ blocks,
                                                       const cplx *__restrict__ x, const cplx *__restrict__ xh, int32_t nb_own,
                                                       cplx *__restrict__ y, const int *__restrict__ skip, int skip_it, const int32_t *__restrict__ order) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    cplx *prod = reinterpret_cast<cplx *>(smem_raw);  // [bs][bs+1]
    if (skip && skip[0] < skip[1] + skip_it) return;  // {stop_at, base}: see gcr.hip DevState
    const int32_t brow = order ? order[blockIdx.x] : (int32_t)blockIdx.x;   // longest block rows first (bcsr_build_device)
    const int lane = threadIdx.x;
    const int32_t bs2 = bs * bs, ld = bs + 1;
    const int32_t beg = browptr[brow], end = browptr[brow + 1];
    // rows of a block owned by this lane: r = lane, lane+64 (bs <= 128)
    cplx acc0 = make_double2(0., 0.), acc1 = make_double2(0., 0.);
    for (int32_t l = beg; l < end; l++) {
        const cplx *m = blocks + (int64_t)l * bs2;
        const int32_t bc = bcol[l];
        const cplx *xb = bc < nb_own ? x + (int64_t)bc * bs : xh + (int64_t)(bc - nb_own) * bs;
        for (int32_t e = lane; e < bs2; e += 64) {
            int32_t r = e / bs, cc = e - r * bs;
            prod[r * ld + cc] = cmul(m[e], xb[cc]);
        }
        __syncthreads();
        if (lane < bs) {
            cplx o = make_double2(0., 0.);
            for (int32_t cc = 0; cc < bs; cc++) o = cadd(o, prod[lane * ld + cc]);
            acc0 = cadd(acc0, o);
        }
        if (lane + 64 < bs) {
            cplx o = make_double2(0., 0.);
            for (int32_t cc = 0; cc < bs; cc++) o = cadd(o, prod[(lane + 64) * ld + cc]);
            acc1 = cadd(acc1, o);
        }
        __syncthreads();
    }
    if (lane < bs) y[(int64_t)brow * bs + lane] = acc0;
    if (lane + 64 < bs) y[(int64_t)brow * bs + lane + 64] = acc1;
}

// Same algorithm with the block's bs*bs entries held in registers: TT = ceil(bs*bs/64) (rounded up to
// 1, 2, 4, 8, 16) matrix loads and x gathers per lane are ISSUED TOGETHER for every block, instead of
// one dependent load per loop trip — the generic kernel above keeps a single 1-KiB load in flight per
// wave and is latency-bound.  The (row, column) of each lane's entries does not depend on the block
// and is computed once.  PREFETCH additionally keeps the NEXT block's loads in flight during the LDS
// phase; measured on MI355X (bs = 20, 3 GB of blocks) it loses to the plain form (5.08 vs 5.28 TB/s:
// 164 VGPRs cost a third of the resident waves), so it is compiled but not dispatched.
template <int TT, bool PREFETCH>
__global__ void __launch_bounds__(64) bcsr_wave_kernel_t(int32_t nbrow, int32_t bs, const int32_t *__restrict__ browptr,
                                                         const int32_t *__restrict__ bcol, const cplx *__restrict__ blocks,
                                                         const cplx *__restrict__ x, const cplx *__restrict__ xh, int32_t nb_own,
                                                         cplx *__restrict__ y, const int *__restrict__ skip, int skip_it, const int32_t *__restrict__ order) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    cplx *prod = reinterpret_cast<cplx *>(smem_raw);  // [bs][bs+1]
    if (skip && skip[0] < skip[1] + skip_it) return;  // {stop_at, base}: see gcr.hip DevState
    const int32_t brow = order ? order[blockIdx.x] : (int32_t)blockIdx.x;   // longest block rows first (bcsr_build_device)
    const int lane = threadIdx.x;
    const int32_t bs2 = bs * bs, ld = bs + 1;
    const int32_t beg = browptr[brow], end = browptr[brow + 1];
    int32_t ecol[TT], elds[TT];
    bool live[TT];
#pragma unroll
    for (int t = 0; t < TT; t++) {
        int32_t e = lane + 64 * t;
        live[t] = e < bs2;
        int32_t r = live[t] ? e / bs : 0;
        ecol[t] = live[t] ? e - r * bs : 0;
        elds[t] = r * ld + ecol[t];
    }
    cplx acc = make_double2(0., 0.);
    cplx mv[TT], xv[TT], mn[TT], xn[TT];
    auto fetch = [&](int32_t l, cplx (&mo)[TT], cplx (&xo)[TT]) {
        const cplx *m = blocks + (int64_t)l * bs2;
        const int32_t bc = bcol[l];
        const cplx *xb = bc < nb_own ? x + (int64_t)bc * bs : xh + (int64_t)(bc - nb_own) * bs;
#pragma unroll
        for (int t = 0; t < TT; t++) {
            mo[t] = live[t] ? m[lane + 64 * t] : make_double2(0., 0.);
            xo[t] = xb[ecol[t]];
        }
    };
    if (beg < end) fetch(beg, mv, xv);
    for (int32_t l = beg; l < end; l++) {
        // the next block's loads are in flight while this block goes through LDS
        if (PREFETCH && l + 1 < end) fetch(l + 1, mn, xn);
#pragma unroll
        for (int t = 0; t < TT; t++)
            if (live[t]) prod[elds[t]] = cmul(mv[t], xv[t]);
        __syncthreads();
        if (lane < bs) {
            cplx o = make_double2(0., 0.);
            for (int32_t cc = 0; cc < bs; cc++) o = cadd(o, prod[lane * ld + cc]);
            acc = cadd(acc, o);
        }
        __syncthreads();
        if (PREFETCH) {
#pragma unroll
            for (int t = 0; t < TT; t++) { mv[t] = mn[t]; xv[t] = xn[t]; }
        } else if (l + 1 < end) {
            fetch(l + 1, mv, xv);
        }
    }
    if (lane < bs) y[(int64_t)brow * bs + lane] = acc;
}

void bcsr_free(BcsrDev *b) {
    hipFree(b->browptr); hipFree(b->bcol); hipFree(b->blocks); hipFree(b->order);
    *b = BcsrDev();
}

int bcsr_build_device(int32_t nbrow, int32_t nbcol, int32_t bs, const int32_t *h_browptr, const int32_t *h_bcol,
                      const double *h_blocks, BcsrDev *out) {
    MGCR_CHECK(nbrow >= 0 && nbcol >= 0 && bs >= 1 && bs <= 128, MGCR_ERR_UNSUPPORTED,
               "block-CSR: block size must be in [1,128] (got %d)", bs);
    MGCR_CHECK(h_browptr[0] == 0, MGCR_ERR_INVALID, "browptr[0] must be 0");
    for (int32_t r = 0; r < nbrow; r++) MGCR_CHECK(h_browptr[r + 1] >= h_browptr[r], MGCR_ERR_INVALID, "browptr not monotone");
    int32_t nb = h_browptr[nbrow];
    for (int32_t i = 0; i < nb; i++) MGCR_CHECK(h_bcol[i] >= 0 && h_bcol[i] < nbcol, MGCR_ERR_INVALID, "block column out of range");
    BcsrDev B;
    B.nbrow = nbrow; B.nbcol = nbcol; B.bs = bs; B.nblocks = nb;
    MGCR_TRY(dev_upload(&B.browptr, h_browptr, (size_t)nbrow + 1));
    MGCR_TRY(dev_upload(&B.bcol, h_bcol, (size_t)nb));
    MGCR_TRY(dev_upload(&B.blocks, (const cplx *)h_blocks, (size_t)nb * bs * bs));
    // One wave per block row, rows of 5 .. 64 blocks: dealt in stored order, the rows that happen to come last decide when the
    // kernel ends (a 64-block row started near the end runs almost alone).  Longest rows first (stable: equal counts keep
    // their order) — each row is still summed by one wave in its own order, so the result does not change by a bit.
    static const bool lpt_on = !(getenv("MGCR_BCSR_ORDER") && atoi(getenv("MGCR_BCSR_ORDER")) == 0);
    if (lpt_on && nbrow >= 1024) {
        int32_t cmin = INT32_MAX, cmax = 0;
        for (int32_t r = 0; r < nbrow; r++) {
            const int32_t c = h_browptr[r + 1] - h_browptr[r];
            cmin = std::min(cmin, c); cmax = std::max(cmax, c);
        }
        if (cmax > cmin) {
            std::vector<int32_t> order((size_t)nbrow);
            for (int32_t r = 0; r < nbrow; r++) order[(size_t)r] = r;
            std::stable_sort(order.begin(), order.end(), [&](int32_t a, int32_t b) {
                return h_browptr[a + 1] - h_browptr[a] > h_browptr[b + 1] - h_browptr[b];
            });
            MGCR_TRY(dev_upload(&B.order, order.data(), (size_t)nbrow));
        }
    }
    MGCR_HIP(hipStreamSynchronize(ctx().stream));
    *out = B;
    return MGCR_OK;
}

int bcsr_apply(const BcsrDev &A, const cplx *x, cplx *y, const cplx *xh, int32_t nb_own) {
    MGCR_CHECK(x != y, MGCR_ERR_INVALID, "block SpMV cannot run in place");
    if (A.nbrow == 0) return MGCR_OK;
    size_t lds = sizeof(cplx) * (size_t)A.bs * (size_t)(A.bs + 1);
    static bool attr_set = false;
    if (!attr_set) {
        MGCR_HIP(hipFuncSetAttribute((const void *)bcsr_wave_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set = true;
    }
    MGCR_CHECK(lds <= 160 * 1024, MGCR_ERR_UNSUPPORTED, "block size %d needs more than 160 KiB of LDS", A.bs);
    const int tt = (A.bs * A.bs + 63) / 64;
#define BT(T_)                                                                                                              \
    hipLaunchKernelGGL((bcsr_wave_kernel_t<T_, false>), dim3((unsigned)A.nbrow), dim3(64), lds, ctx().stream, A.nbrow, A.bs, A.browptr, \
                       A.bcol, A.blocks, x, xh, nb_own, y, g_skip.p, g_skip.it, A.order)
    if (A.bs > 64 || tt > 16)  // rows beyond lane 63 / too many registers: generic kernel
        hipLaunchKernelGGL(bcsr_wave_kernel, dim3((unsigned)A.nbrow), dim3(64), lds, ctx().stream, A.nbrow, A.bs, A.browptr,
                           A.bcol, A.blocks, x, xh, nb_own, y, g_skip.p, g_skip.it, A.order);
    else if (tt <= 1) BT(1);
    else if (tt <= 2) BT(2);
    else if (tt <= 4) BT(4);
    else if (tt <= 8) BT(8);
    else BT(16);
#undef BT
    MGCR_HIP(hipGetLastError());
    return MGCR_OK;
}

}  // namespace mgcr
