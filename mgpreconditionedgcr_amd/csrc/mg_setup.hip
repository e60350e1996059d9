// Multigrid set-up ON THE DEVICE (the offload the reference's author
// earmarked in src/cuda.cu:11-15 and never wrote): aggregate index / member lists, block-local
// prolongator with per-aggregate Gram-Schmidt, Galerkin coarse operator, restricted near-null
// vectors for the next level — without pulling the operator back to the host.
//
// Every kernel keeps the reference's evaluation order (src/MG.h:171-198,216-274), one thread per
// aggregate / per coarse block, so the hierarchy is bit-identical to the oracle's
// (tests/test_gpu_mg.py::test_device_setup_bit_identical_to_oracle) and, through it, pinned to the
// reference's golden G9.
//
// Distributed operators (row blocks): aggregates are formed inside the local block; the aggregate
// ids and prolongator rows of the halo columns are fetched once over the SpMV's halo lists
// (set-up-time host-level exchange), the Galerkin kernels then see owned + halo columns, and the
// resulting row block of the coarse operator (global column ids) becomes a distributed Sparse (one near-null
// vector) or a distributed HierarchicalSparse (several: ne x ne blocks).
#include <algorithm>

#include "internal.h"
#include "reduce.h"

namespace mgcr {

struct MeshDesc {
    int ndim;
    int64_t dims[8];
    int32_t blocked[8];
    int64_t sub;
};

// rows of the operator the Galerkin product is taken of: ELL slab + CSR tail, or block-CSR
struct RowSrc {
    int kind;  // 0: ELL (+ tail), 1: block-CSR
    int64_t npad;
    int32_t Wp, L;
    const cplx *val;
    const double *val_re;
    const int32_t *col;
    // row-pattern dictionary (CsrDev::pat_mode): ids, offsets and (mode 1) values replace col / val
    int pat_mode;
    const uint16_t *pid;
    const int32_t *poff;
    const double *pre, *pim;
    int64_t ntail;
    const int32_t *trows, *tptr, *tcol;
    const cplx *tval;
    int32_t bs;
    const int32_t *browptr, *bcol;
    const cplx *blocks;
};

template <class F>
__device__ __forceinline__ void for_each_entry(const RowSrc &s, int64_t i, F f) {
    if (s.kind == 0) {
        for (int32_t w = 0; w < s.Wp; w++) {
            int64_t idx = ((int64_t)(w / s.L) * s.npad + i) * s.L + (w % s.L);
            if (s.pat_mode) {
                int32_t t = (int32_t)s.pid[i] * s.Wp + w;
                cplx v = s.pat_mode == 1 ? make_double2(s.pre[t], s.pim[t]) : (s.val_re ? make_double2(s.val_re[idx], 0.) : s.val[idx]);
                f(i + (int64_t)s.poff[t], v);
                continue;
            }
            cplx v = s.val_re ? make_double2(s.val_re[idx], 0.) : s.val[idx];
            f((int64_t)s.col[idx], v);
        }
        if (s.ntail) {
            int64_t lo = 0, hi = s.ntail;
            while (lo < hi) {
                int64_t mid = (lo + hi) >> 1;
                if (s.trows[mid] < i) lo = mid + 1; else hi = mid;
            }
            if (lo < s.ntail && s.trows[lo] == i)
                for (int32_t q = s.tptr[lo]; q < s.tptr[lo + 1]; q++) f((int64_t)s.tcol[q], s.tval[q]);
        }
    } else {
        int64_t br = i / s.bs;
        int32_t r = (int32_t)(i - br * s.bs);
        for (int32_t l = s.browptr[br]; l < s.browptr[br + 1]; l++)
            for (int32_t c = 0; c < s.bs; c++) f((int64_t)s.bcol[l] * s.bs + c, s.blocks[(int64_t)l * s.bs * s.bs + (int64_t)r * s.bs + c]);
    }
}

// Mesh::blocking (src/Mesh.h:236-298): aggregate of unknown i, and its position inside the aggregate
// in ascending-index order (row-major over the in-block / unblocked coordinates)
__global__ void agg_kernel(int64_t n, MeshDesc m, int64_t S, int32_t *__restrict__ agg, int32_t *__restrict__ amem) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int64_t idx[8], rem = i;
    for (int d = m.ndim - 1; d >= 0; d--) { idx[d] = rem % m.dims[d]; rem /= m.dims[d]; }
    int64_t b = 0, pos = 0;
    for (int d = 0; d < m.ndim; d++) {
        if (m.blocked[d]) {
            b = b * (m.dims[d] / m.sub) + idx[d] / m.sub;
            pos = pos * m.sub + idx[d] % m.sub;
        } else {
            pos = pos * m.dims[d] + idx[d];
        }
    }
    agg[i] = (int32_t)b;
    amem[b * S + pos] = (int32_t)i;
}

__global__ void aptr_kernel(int64_t nagg, int64_t S, int32_t *__restrict__ aptr) {
    int64_t a = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (a <= nagg) aptr[a] = (int32_t)(a * S);
}

// pv[i][k] = vecs[k][i]
__global__ void pv_init_kernel(int64_t n, int ne, const cplx *__restrict__ vecs, cplx *__restrict__ pv) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    for (int k = 0; k < ne; k++) pv[i * ne + k] = vecs[(int64_t)k * n + i];
}

// per-aggregate modified Gram-Schmidt + normalise (src/MG.h:190-198), one thread per aggregate,
// sums over the members in ascending index order
__global__ void gs_kernel(int64_t nagg, int ne, const int32_t *__restrict__ aptr, const int32_t *__restrict__ amem, cplx *pv) {
    int64_t a = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= nagg) return;
    const int32_t b = aptr[a], e = aptr[a + 1];
    for (int vec = 0; vec < ne; vec++) {
        for (int j = 0; j < vec; j++) {
            cplx h = make_double2(0., 0.);
            for (int32_t m = b; m < e; m++) h = cadd(h, cconj_mul(pv[(int64_t)amem[m] * ne + j], pv[(int64_t)amem[m] * ne + vec]));
            for (int32_t m = b; m < e; m++) {
                int64_t q = (int64_t)amem[m] * ne;
                pv[q + vec] = csub(pv[q + vec], cmul(h, pv[q + j]));
            }
        }
        double s = 0.;
        for (int32_t m = b; m < e; m++) {
            cplx t = pv[(int64_t)amem[m] * ne + vec];
            cplx u = cconj_mul(t, t);
            s += u.x;  // Re(conj(t) t) accumulated like src/Fields.h:228-235
        }
        const double inv = 1. / sqrt(s);
        for (int32_t m = b; m < e; m++) {
            int64_t q = (int64_t)amem[m] * ne + vec;
            pv[q] = make_double2(pv[q].x * inv, pv[q].y * inv);  // field[i] *= 1./norm
        }
    }
}

constexpr int MAXNB_FIRST = 64, MAXNB_LAST = 4096;  // neighbour-list capacity per aggregate: grown x4 on overflow

// neighbour aggregates of every aggregate (ascending, itself included for a shifted operator)
__global__ void gal_count_kernel(int64_t nagg, RowSrc src, int shift, const int32_t *__restrict__ agg,
                                 const int32_t *__restrict__ aptr, const int32_t *__restrict__ amem,
                                 int32_t *__restrict__ cnt, int32_t *__restrict__ nbr, int *__restrict__ overflow, int maxnb) {
    int64_t ap = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (ap >= nagg) return;
    int32_t *mine = nbr + ap * maxnb;
    int32_t nn = 0, last = -1;
    bool over = false;
    auto add = [&](int32_t a) {
        if (a == last) return;   // consecutive entries mostly fall into one aggregate (a block's columns, a stencil's neighbours): no search
        last = a;
        int32_t lo = 0;
        while (lo < nn && mine[lo] < a) lo++;
        if (lo < nn && mine[lo] == a) return;
        if (nn == maxnb) { over = true; return; }
        for (int32_t q = nn; q > lo; q--) mine[q] = mine[q - 1];
        mine[lo] = a;
        nn++;
    };
    if (shift) add((int32_t)ap);
    for (int32_t m = aptr[ap]; m < aptr[ap + 1]; m++)
        for_each_entry(src, (int64_t)amem[m], [&](int64_t j, cplx) { add(agg[j]); });
    cnt[ap] = nn;
    if (over) *overflow = 1;
}

__global__ void gal_index_kernel(int64_t nagg, const int32_t *__restrict__ browptr, const int32_t *__restrict__ nbr,
                                 int32_t *__restrict__ brow, int32_t *__restrict__ bcol, int maxnb) {
    int64_t ap = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (ap >= nagg) return;
    for (int32_t b = browptr[ap]; b < browptr[ap + 1]; b++) {
        brow[b] = (int32_t)ap;
        bcol[b] = nbr[ap * maxnb + (b - browptr[ap])];
    }
}

// one thread per coarse block (a', a):  blk[k'][k] = sum_{i in a', ascending} conj(pv[i][k']) * y_i[k],
// y_i[k] = row sum over the columns of a in storage order, then the optional shift (src/MG.h:216-274)
__global__ void gal_fill_kernel(int64_t nblk, RowSrc src, int shift, cplx kshift, int ne, const int32_t *__restrict__ agg,
                                const int32_t *__restrict__ aptr, const int32_t *__restrict__ amem, const cplx *__restrict__ pv,
                                const int32_t *__restrict__ brow, const int32_t *__restrict__ bcol, cplx *__restrict__ blocks,
                                cplx *__restrict__ tscratch) {
    int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nblk) return;
    const int32_t ap = brow[b], a = bcol[b];
    cplx *blk = blocks + b * ne * ne;
    cplx *t = tscratch + b * ne;
    for (int32_t m = aptr[ap]; m < aptr[ap + 1]; m++) {
        const int64_t i = amem[m];
        for (int k = 0; k < ne; k++) t[k] = make_double2(0., 0.);
        for_each_entry(src, i, [&](int64_t j, cplx v) {
            if (agg[j] == a)
                for (int k = 0; k < ne; k++) t[k] = cadd(t[k], cmul(v, pv[j * ne + k]));
        });
        for (int k = 0; k < ne; k++) {
            cplx y = t[k];
            if (shift) y = csub((agg[i] == a) ? pv[i * ne + k] : make_double2(0., 0.), cmul(kshift, t[k]));
            for (int kp = 0; kp < ne; kp++) blk[kp * ne + k] = cadd(blk[kp * ne + k], cconj_mul(pv[i * ne + kp], y));
        }
    }
}

__global__ void widen_kernel(int64_t n, const int32_t *__restrict__ in, int64_t *__restrict__ out) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = in[i];
}

int csr_build_from_device(int64_t nrow, int64_t ncol, const int64_t *h_rowptr, const int64_t *d_rowptr, const int64_t *d_col,
                          const cplx *d_val, CsrDev *out);
int mg_restrict_raw(int64_t nc, int ne, const int32_t *aptr, const int32_t *amem, const cplx *pv, const cplx *x, cplx *xc);

template <typename T>
static int dmalloc(T **p, size_t count) {
    *p = nullptr;
    hipError_t e = hipMalloc((void **)p, sizeof(T) * (count ? count : 1));
    if (e != hipSuccess) {
        set_error("hipMalloc of %zu bytes failed: %s", sizeof(T) * count, hipGetErrorString(e));
        return MGCR_ERR_ALLOC;
    }
    return MGCR_OK;
}

static unsigned g256(int64_t n) { return (unsigned)((n + 255) / 256); }

static RowSrc row_source(const Op *op) {
    RowSrc s;
    memset(&s, 0, sizeof(s));
    if (op->kind == OP_BCSR) {
        s.kind = 1; s.bs = op->bcsr.bs; s.browptr = op->bcsr.browptr; s.bcol = op->bcsr.bcol; s.blocks = op->bcsr.blocks;
    } else {
        const CsrDev &A = op->csr;
        s.kind = 0; s.npad = A.npad; s.Wp = A.nchunk * A.L; s.L = A.L; s.val = A.ell_val; s.val_re = A.ell_val_re; s.col = A.ell_col;
        s.pat_mode = A.pat_mode; s.pid = A.pat_id; s.poff = A.pat_off; s.pre = A.pat_re; s.pim = A.pat_im;
        s.ntail = A.n_tail_rows; s.trows = A.tail_rows; s.tptr = A.tail_ptr; s.tcol = A.tail_col; s.tval = A.tail_val;
    }
    return s;
}

// One level of the hierarchy, entirely on the device.  `A` is the level's operator (Sparse, DiracOp,
// or a coarse operator built by a previous call), d_vecs its ne near-null vectors [ne][n].
// Outputs: aggregate arrays, prolongator, the coarse operator (new Op, owned by the caller) and the
// restricted vectors for the next level (d_vecs_next [ne][nagg*ne], may be skipped with want_next = false).
int mg_level_setup_device(Op *A, int ndim, const int64_t *dims, const int32_t *blocked, int64_t sub, int ne, const cplx *d_vecs,
                          int64_t *nagg_out, int32_t **d_agg, int32_t **d_aptr, int32_t **d_amem, cplx **d_pv, Op **coarse,
                          bool want_next, cplx **d_vecs_next) {
    hipStream_t st = ctx().stream;
    const Op *base = A->kind == OP_DIRAC ? A->base : A;
    const bool shift = A->kind == OP_DIRAC;
    DistCsr *dist = base->dist;   // row block of a distributed Sparse or HierarchicalSparse
    MeshDesc m;
    m.ndim = ndim; m.sub = sub;
    int64_t n = 1, nagg = 1, S = 1;
    for (int d = 0; d < ndim; d++) {
        m.dims[d] = dims[d]; m.blocked[d] = blocked[d];
        n *= dims[d];
        if (blocked[d]) {
            if (sub <= 0 || dims[d] % sub) { set_error("Dimension not exactly divisible by block size!"); return MGCR_ERR_INVALID; }
            nagg *= dims[d] / sub;
            S *= sub;
        } else {
            S *= dims[d];
        }
    }
    MGCR_CHECK(nagg * ne < ((int64_t)1 << 31), MGCR_ERR_UNSUPPORTED, "coarse level too large");
    MGCR_TRY(dmalloc(d_agg, (size_t)n));
    MGCR_TRY(dmalloc(d_aptr, (size_t)nagg + 1));
    MGCR_TRY(dmalloc(d_amem, (size_t)n));
    MGCR_TRY(dmalloc(d_pv, (size_t)n * ne));
    hipLaunchKernelGGL(agg_kernel, dim3(g256(n)), dim3(256), 0, st, n, m, S, *d_agg, *d_amem);
    hipLaunchKernelGGL(aptr_kernel, dim3(g256(nagg + 1)), dim3(256), 0, st, nagg, S, *d_aptr);
    hipLaunchKernelGGL(pv_init_kernel, dim3(g256(n)), dim3(256), 0, st, n, ne, d_vecs, *d_pv);
    hipLaunchKernelGGL(gs_kernel, dim3(g256(nagg)), dim3(256), 0, st, nagg, ne, (const int32_t *)*d_aptr, (const int32_t *)*d_amem, *d_pv);
    MGCR_HIP(hipGetLastError());
    // columns the Galerkin product sees: the owned rows, plus (distributed) the halo slots
    const int32_t *g_agg = *d_agg;
    const cplx *g_pv = *d_pv;
    int32_t *d_agg_ext = nullptr;
    cplx *d_pv_ext = nullptr;
    int64_t agg_off = 0, nagg_glob = nagg;
    std::vector<int64_t> ext_gid;  // global id of the extended-local aggregate nagg + k
    Comm *comm = nullptr;
    if (dist) {
        int64_t nh = 0;
        int rank = 0, nranks = 1;
        dist_sizes(dist, nullptr, &nh, nullptr, nullptr, &rank, &nranks);
        comm = dist_comm(dist);
        std::vector<double> cnt((size_t)nranks, 0.);
        cnt[(size_t)rank] = (double)nagg;
        MGCR_TRY(comm_allreduce_host_pub(comm, cnt.data(), nranks));
        nagg_glob = 0;
        for (int r = 0; r < nranks; r++) { if (r == rank) agg_off = nagg_glob; nagg_glob += (int64_t)cnt[(size_t)r]; }
        std::vector<int32_t> h_agg((size_t)n);
        std::vector<double> h_pv((size_t)n * ne * 2);
        MGCR_HIP(hipMemcpyAsync(h_agg.data(), *d_agg, sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToHost, st));
        MGCR_HIP(hipMemcpyAsync(h_pv.data(), *d_pv, sizeof(cplx) * (size_t)n * ne, hipMemcpyDeviceToHost, st));
        MGCR_HIP(hipStreamSynchronize(st));
        const int w = 1 + 2 * ne;
        std::vector<double> own((size_t)n * w), halo((size_t)nh * w);
        for (int64_t i = 0; i < n; i++) {
            own[(size_t)i * w] = (double)(agg_off + h_agg[(size_t)i]);
            memcpy(&own[(size_t)i * w + 1], &h_pv[(size_t)i * ne * 2], sizeof(double) * 2 * (size_t)ne);
        }
        MGCR_TRY(dist_exchange_rows_host(dist, own.data(), w, halo.data()));
        for (int64_t h = 0; h < nh; h++) ext_gid.push_back((int64_t)halo[(size_t)h * w]);
        std::sort(ext_gid.begin(), ext_gid.end());
        ext_gid.erase(std::unique(ext_gid.begin(), ext_gid.end()), ext_gid.end());
        h_agg.resize((size_t)(n + nh));
        h_pv.resize((size_t)(n + nh) * ne * 2);
        for (int64_t h = 0; h < nh; h++) {
            int64_t gid = (int64_t)halo[(size_t)h * w];
            h_agg[(size_t)(n + h)] = (int32_t)(nagg + (std::lower_bound(ext_gid.begin(), ext_gid.end(), gid) - ext_gid.begin()));
            memcpy(&h_pv[(size_t)(n + h) * ne * 2], &halo[(size_t)h * w + 1], sizeof(double) * 2 * (size_t)ne);
        }
        MGCR_TRY(dmalloc(&d_agg_ext, (size_t)(n + nh)));
        MGCR_TRY(dmalloc(&d_pv_ext, (size_t)(n + nh) * ne));
        MGCR_HIP(hipMemcpy(d_agg_ext, h_agg.data(), sizeof(int32_t) * (size_t)(n + nh), hipMemcpyHostToDevice));
        MGCR_HIP(hipMemcpy(d_pv_ext, h_pv.data(), sizeof(cplx) * (size_t)(n + nh) * ne, hipMemcpyHostToDevice));
        g_agg = d_agg_ext;
        g_pv = d_pv_ext;
    }
    // Galerkin: symbolic
    RowSrc src = row_source(base);
    int32_t *d_cnt = nullptr, *d_nbr = nullptr;
    int *d_over = nullptr;
    MGCR_TRY(dmalloc(&d_cnt, (size_t)nagg));
    MGCR_TRY(dmalloc(&d_over, 1));
    std::vector<int32_t> cnt((size_t)nagg);
    int maxnb = MAXNB_FIRST;
    for (;;) {  // lattice operators couple an aggregate to a few dozen others; unstructured block operators to hundreds
        MGCR_TRY(dmalloc(&d_nbr, (size_t)nagg * maxnb));
        MGCR_HIP(hipMemsetAsync(d_over, 0, sizeof(int), st));
        hipLaunchKernelGGL(gal_count_kernel, dim3(g256(nagg)), dim3(256), 0, st, nagg, src, shift ? 1 : 0, g_agg,
                           (const int32_t *)*d_aptr, (const int32_t *)*d_amem, d_cnt, d_nbr, d_over, maxnb);
        MGCR_HIP(hipGetLastError());
        int over = 0;
        MGCR_HIP(hipMemcpyAsync(cnt.data(), d_cnt, sizeof(int32_t) * (size_t)nagg, hipMemcpyDeviceToHost, st));
        MGCR_HIP(hipMemcpyAsync(&over, d_over, sizeof(int), hipMemcpyDeviceToHost, st));
        MGCR_HIP(hipStreamSynchronize(st));
        if (!over) break;
        hipFree(d_nbr);
        d_nbr = nullptr;
        if (maxnb >= MAXNB_LAST) {
            hipFree(d_cnt); hipFree(d_over);
            set_error("an aggregate couples to more than %d aggregates", MAXNB_LAST);
            return MGCR_ERR_UNSUPPORTED;
        }
        maxnb *= 4;
    }
    hipFree(d_cnt); hipFree(d_over);
    std::vector<int32_t> browptr((size_t)nagg + 1, 0);
    for (int64_t a = 0; a < nagg; a++) {
        int64_t nx = (int64_t)browptr[(size_t)a] + cnt[(size_t)a];
        MGCR_CHECK(nx < ((int64_t)1 << 31), MGCR_ERR_UNSUPPORTED, "coarse operator exceeds 2^31 blocks");
        browptr[(size_t)a + 1] = (int32_t)nx;
    }
    const int64_t nblk = browptr[(size_t)nagg];
    // Galerkin: numeric
    int32_t *d_browptr = nullptr, *d_brow = nullptr, *d_bcol = nullptr;
    cplx *d_blocks = nullptr, *d_t = nullptr;
    MGCR_TRY(dmalloc(&d_browptr, (size_t)nagg + 1));
    MGCR_TRY(dmalloc(&d_brow, (size_t)nblk));
    MGCR_TRY(dmalloc(&d_bcol, (size_t)nblk));
    MGCR_TRY(dmalloc(&d_blocks, (size_t)nblk * ne * ne));
    MGCR_TRY(dmalloc(&d_t, (size_t)nblk * ne));
    MGCR_HIP(hipMemcpyAsync(d_browptr, browptr.data(), sizeof(int32_t) * ((size_t)nagg + 1), hipMemcpyHostToDevice, st));
    MGCR_HIP(hipMemsetAsync(d_blocks, 0, sizeof(cplx) * (size_t)nblk * ne * ne, st));
    hipLaunchKernelGGL(gal_index_kernel, dim3(g256(nagg)), dim3(256), 0, st, nagg, (const int32_t *)d_browptr, (const int32_t *)d_nbr, d_brow, d_bcol, maxnb);
    hipLaunchKernelGGL(gal_fill_kernel, dim3(g256(nblk)), dim3(256), 0, st, nblk, src, shift ? 1 : 0, A->k, ne, g_agg,
                       (const int32_t *)*d_aptr, (const int32_t *)*d_amem, g_pv, (const int32_t *)d_brow,
                       (const int32_t *)d_bcol, d_blocks, d_t);
    MGCR_HIP(hipGetLastError());
    MGCR_HIP(hipStreamSynchronize(st));
    hipFree(d_nbr); hipFree(d_brow); hipFree(d_t); hipFree(d_agg_ext); hipFree(d_pv_ext);
    // coarse operator
    mgcr_op_s *Ac = new mgcr_op_s();
    const int64_t nc = nagg * ne;
    Ac->dim = Ac->nrow = nc;
    int rc = MGCR_OK;
    if (dist && ne > 1) {
        // this rank's block rows [agg_off, agg_off + nagg) of the coarse operator as a distributed HierarchicalSparse
        // (block columns GLOBAL).  The partition plan is host logic: the block columns make one trip back, the blocks
        // themselves too (dist_bcsr_create uploads host arrays)
        std::vector<int32_t> h_bcol((size_t)nblk);
        std::vector<double> h_blk((size_t)nblk * ne * ne * 2);
        MGCR_HIP(hipMemcpy(h_bcol.data(), d_bcol, sizeof(int32_t) * (size_t)nblk, hipMemcpyDeviceToHost));
        MGCR_HIP(hipMemcpy(h_blk.data(), d_blocks, sizeof(cplx) * (size_t)nblk * ne * ne, hipMemcpyDeviceToHost));
        hipFree(d_browptr); hipFree(d_bcol); hipFree(d_blocks);
        std::vector<int64_t> gcol((size_t)nblk);
        for (int64_t b = 0; b < nblk; b++) {
            const int64_t ca = h_bcol[(size_t)b];
            gcol[(size_t)b] = ca < nagg ? agg_off + ca : ext_gid[(size_t)(ca - nagg)];
        }
        Ac->kind = OP_BCSR;
        rc = dist_bcsr_create(comm, nagg_glob, agg_off, (int32_t)nagg, ne, browptr.data(), gcol.data(), h_blk.data(), Ac);
    } else if (dist) {
        // ne == 1: this rank's row block [agg_off, agg_off + nagg) of the coarse operator, GLOBAL columns, as a distributed
        // Sparse (row-pattern / stencil storage then applies to it); the partition plan is host logic, so the entries make
        // one trip back
        std::vector<int32_t> h_bcol((size_t)nblk);
        std::vector<double> h_blk((size_t)nblk * 2);
        MGCR_HIP(hipMemcpy(h_bcol.data(), d_bcol, sizeof(int32_t) * (size_t)nblk, hipMemcpyDeviceToHost));
        MGCR_HIP(hipMemcpy(h_blk.data(), d_blocks, sizeof(cplx) * (size_t)nblk, hipMemcpyDeviceToHost));
        hipFree(d_browptr); hipFree(d_bcol); hipFree(d_blocks);
        std::vector<int64_t> rp((size_t)nc + 1, 0), ci((size_t)nblk);
        for (int64_t a = 0; a <= nagg; a++) rp[(size_t)a] = browptr[(size_t)a];
        for (int64_t b = 0; b < nblk; b++) {
            const int64_t ca = h_bcol[(size_t)b];
            ci[(size_t)b] = ca < nagg ? agg_off + ca : ext_gid[(size_t)(ca - nagg)];
        }
        Ac->kind = OP_CSR;
        rc = dist_csr_create(comm, nagg_glob, agg_off, nc, rp.data(), ci.data(), h_blk.data(), Ac);
    } else if (ne == 1) {
        Ac->kind = OP_CSR;
        std::vector<int64_t> rp64(browptr.begin(), browptr.end());
        int64_t *d_rp = nullptr, *d_ci = nullptr;
        rc = dmalloc(&d_rp, (size_t)nagg + 1);
        if (rc == MGCR_OK) rc = dmalloc(&d_ci, (size_t)nblk);
        if (rc == MGCR_OK) {
            hipLaunchKernelGGL(widen_kernel, dim3(g256(nagg + 1)), dim3(256), 0, st, nagg + 1, (const int32_t *)d_browptr, d_rp);
            hipLaunchKernelGGL(widen_kernel, dim3(g256(nblk)), dim3(256), 0, st, nblk, (const int32_t *)d_bcol, d_ci);
            rc = csr_build_from_device(nc, nc, rp64.data(), d_rp, d_ci, d_blocks, &Ac->csr);
        }
        hipStreamSynchronize(st);
        hipFree(d_rp); hipFree(d_ci); hipFree(d_browptr); hipFree(d_bcol); hipFree(d_blocks);
    } else {
        Ac->kind = OP_BCSR;
        Ac->bcsr.nbrow = Ac->bcsr.nbcol = (int32_t)nagg;
        Ac->bcsr.bs = ne;
        Ac->bcsr.nblocks = (int32_t)nblk;
        Ac->bcsr.browptr = d_browptr; Ac->bcsr.bcol = d_bcol; Ac->bcsr.blocks = d_blocks;  // ownership moves to the operator
    }
    if (rc != MGCR_OK) { delete Ac; return rc; }
    *coarse = Ac;
    *nagg_out = nagg;
    if (want_next) {
        MGCR_TRY(dmalloc(d_vecs_next, (size_t)ne * nc));
        for (int k = 0; k < ne; k++)
            MGCR_TRY(mg_restrict_raw(nc, ne, *d_aptr, *d_amem, *d_pv, d_vecs + (size_t)k * n, *d_vecs_next + (size_t)k * nc));
        MGCR_HIP(hipStreamSynchronize(st));
    }
    return MGCR_OK;
}

}  // namespace mgcr
