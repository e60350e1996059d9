// Aggregation multigrid: set-up and the V-cycle (src/MG.h, src/Mesh.h re-designed for MI355X).
//
//   reference                                         here
//   Mesh::blocking / block_map   src/Mesh.h:236-298    aggregate index per unknown + member lists
//   prolongator[block][k] as full-length zero-padded    block-local pv[i][k] (N x ne) — restrict and
//   Fields; restrict/expand = nblocks*ne full-length    expand each stream ne*V + V instead of
//   dots / axpys                 src/MG.h:347-403       nblocks*ne*V
//   Galerkin blocks by ne^2 full SpMVs + dots per       one pass over the matrix per aggregate pair
//   block pair                   src/MG.h:204-278
//   m_coarse = HierarchicalSparse src/MG.h:281          block-CSR (ne > 1) or ELL/CSR (ne == 1)
//   MG::solve                    src/MG.h:405-430       corrected V-cycle, see mg_cycle()
//
// The set-up (aggregates, Gram-Schmidt, Galerkin) exists twice with identical arithmetic: on the
// device (mg_setup.hip, the default for single-GPU operators) and on the host from a copy of the
// operator pulled back from HBM (this file; used for distributed operators, whose halo rows travel
// through host-level exchanges, and by the tests as the cross-check).  Everything the solve touches
// per iteration (restrict, prolong+add, residual, smoothers, coarse solve) is HIP and is enqueued
// on the library stream without host round trips.
#include <algorithm>
#include <cmath>
#include <complex>

#include "internal.h"
#include "reduce.h"

namespace mgcr {

typedef std::complex<double> hc;

// ------------------------------------------------------------------------------------------------
// pull an ELL + tail operator back to the host as CSR (padding slots become explicit zeros whose
// column is one the row already references, so they add neither couplings nor value)
// ------------------------------------------------------------------------------------------------
int csr_download_host(const CsrDev &A, HostCsr *out) {
    Context &c = ctx();
    const int64_t Wp = (int64_t)A.nchunk * A.L;
    const size_t slab = (size_t)Wp * (size_t)A.npad;
    std::vector<hc> ev(slab);
    std::vector<int32_t> ec(slab);
    std::vector<double> evr;
    if (slab && A.ell_val_re) {
        evr.resize(slab);
        MGCR_HIP(hipMemcpyAsync(evr.data(), A.ell_val_re, sizeof(double) * slab, hipMemcpyDeviceToHost, c.stream));
        MGCR_HIP(hipMemcpyAsync(ec.data(), A.ell_col, sizeof(int32_t) * slab, hipMemcpyDeviceToHost, c.stream));
    } else if (slab) {
        MGCR_HIP(hipMemcpyAsync(ev.data(), A.ell_val, sizeof(cplx) * slab, hipMemcpyDeviceToHost, c.stream));
        MGCR_HIP(hipMemcpyAsync(ec.data(), A.ell_col, sizeof(int32_t) * slab, hipMemcpyDeviceToHost, c.stream));
    }
    std::vector<int32_t> trows((size_t)A.n_tail_rows), tptr((size_t)A.n_tail_rows + 1, 0), tcol((size_t)A.tail_nnz);
    std::vector<hc> tval((size_t)A.tail_nnz);
    if (A.n_tail_rows) {
        MGCR_HIP(hipMemcpyAsync(trows.data(), A.tail_rows, sizeof(int32_t) * trows.size(), hipMemcpyDeviceToHost, c.stream));
        MGCR_HIP(hipMemcpyAsync(tptr.data(), A.tail_ptr, sizeof(int32_t) * tptr.size(), hipMemcpyDeviceToHost, c.stream));
        MGCR_HIP(hipMemcpyAsync(tcol.data(), A.tail_col, sizeof(int32_t) * tcol.size(), hipMemcpyDeviceToHost, c.stream));
        MGCR_HIP(hipMemcpyAsync(tval.data(), A.tail_val, sizeof(cplx) * tval.size(), hipMemcpyDeviceToHost, c.stream));
    }
    MGCR_HIP(hipStreamSynchronize(c.stream));
    for (size_t i = 0; i < evr.size(); i++) ev[i] = hc(evr[i], 0.);
    out->nrow = A.nrow;
    out->ncol = A.ncol;
    std::vector<int64_t> tail_of((size_t)A.nrow, -1);
    for (int64_t t = 0; t < A.n_tail_rows; t++) tail_of[(size_t)trows[(size_t)t]] = t;
    out->rowptr.assign((size_t)A.nrow + 1, 0);
    for (int64_t r = 0; r < A.nrow; r++) {
        int64_t len = Wp;
        if (tail_of[(size_t)r] >= 0) len += tptr[(size_t)tail_of[(size_t)r] + 1] - tptr[(size_t)tail_of[(size_t)r]];
        out->rowptr[(size_t)r + 1] = out->rowptr[(size_t)r] + len;
    }
    out->col.resize((size_t)out->rowptr[(size_t)A.nrow]);
    out->val_ri.resize(2 * (size_t)out->rowptr[(size_t)A.nrow]);
    hc *ov = reinterpret_cast<hc *>(out->val_ri.data());
    for (int64_t r = 0; r < A.nrow; r++) {
        int64_t p = out->rowptr[(size_t)r];
        for (int64_t w = 0; w < Wp; w++) {
            size_t idx = ((size_t)(w / A.L) * (size_t)A.npad + (size_t)r) * (size_t)A.L + (size_t)(w % A.L);
            out->col[(size_t)p] = ec[idx];
            ov[p] = ev[idx];
            p++;
        }
        if (tail_of[(size_t)r] >= 0) {
            int64_t t = tail_of[(size_t)r];
            for (int32_t q = tptr[(size_t)t]; q < tptr[(size_t)t + 1]; q++) {
                out->col[(size_t)p] = tcol[(size_t)q];
                ov[p] = tval[(size_t)q];
                p++;
            }
        }
    }
    return MGCR_OK;
}

// ------------------------------------------------------------------------------------------------
// host set-up pieces
// ------------------------------------------------------------------------------------------------

// Mesh::blocking (src/Mesh.h:236-298): block index of every unknown; block index row-major over the
// block counts of the blocked dimensions, unblocked dimensions stay inside the aggregate.
static int64_t lattice_aggregates(int ndim, const int64_t *dims, const int32_t *blocked, int64_t sub, std::vector<int32_t> &agg) {
    int64_t n = 1, nagg = 1;
    for (int d = 0; d < ndim; d++) {
        n *= dims[d];
        if (blocked[d]) {
            if (sub <= 0 || dims[d] % sub) return -1;
            nagg *= dims[d] / sub;
        }
    }
    agg.resize((size_t)n);
    std::vector<int64_t> idx((size_t)ndim);
    for (int64_t i = 0; i < n; i++) {
        int64_t rem = i;
        for (int d = ndim - 1; d >= 0; d--) { idx[(size_t)d] = rem % dims[d]; rem /= dims[d]; }
        int64_t b = 0;
        for (int d = 0; d < ndim; d++)
            if (blocked[d]) b = b * (dims[d] / sub) + idx[(size_t)d] / sub;
        agg[(size_t)i] = (int32_t)b;
    }
    return nagg;
}

static void member_lists(int64_t n, int64_t nagg, const std::vector<int32_t> &agg, std::vector<int32_t> &ptr, std::vector<int32_t> &mem) {
    ptr.assign((size_t)nagg + 1, 0);
    for (int64_t i = 0; i < n; i++) ptr[(size_t)agg[(size_t)i] + 1]++;
    for (int64_t a = 0; a < nagg; a++) ptr[(size_t)a + 1] += ptr[(size_t)a];
    std::vector<int32_t> fill(ptr.begin(), ptr.end() - 1);
    mem.resize((size_t)n);
    for (int64_t i = 0; i < n; i++) mem[(size_t)fill[(size_t)agg[(size_t)i]]++] = (int32_t)i;  // ascending inside an aggregate
}

// restrict_block + per-block modified Gram-Schmidt + normalise (src/MG.h:171-198), block-local
static void build_prolongator(int64_t n, int ne, int64_t nagg, const std::vector<int32_t> &ptr, const std::vector<int32_t> &mem,
                              const hc *vecs /*[ne][n]*/, std::vector<hc> &pv /*[n][ne]*/) {
    pv.resize((size_t)n * ne);
    for (int64_t i = 0; i < n; i++)
        for (int k = 0; k < ne; k++) pv[(size_t)i * ne + k] = vecs[(size_t)k * n + i];
    for (int64_t a = 0; a < nagg; a++) {
        const int32_t b = ptr[(size_t)a], e = ptr[(size_t)a + 1];
        for (int vec = 0; vec < ne; vec++) {
            for (int j = 0; j < vec; j++) {
                hc h(0., 0.);
                for (int32_t m = b; m < e; m++) h += std::conj(pv[(size_t)mem[(size_t)m] * ne + j]) * pv[(size_t)mem[(size_t)m] * ne + vec];
                for (int32_t m = b; m < e; m++) pv[(size_t)mem[(size_t)m] * ne + vec] -= h * pv[(size_t)mem[(size_t)m] * ne + j];
            }
            hc s(0., 0.);
            for (int32_t m = b; m < e; m++) s += std::conj(pv[(size_t)mem[(size_t)m] * ne + vec]) * pv[(size_t)mem[(size_t)m] * ne + vec];
            const double nrm = std::sqrt(s.real());
            for (int32_t m = b; m < e; m++) pv[(size_t)mem[(size_t)m] * ne + vec] *= 1. / nrm;
        }
    }
}

// (R x)[a*ne+k] on the host (only used to carry the near-null vectors to the next level)
static void host_restrict(int64_t n, int ne, int64_t nagg, const std::vector<int32_t> &agg, const std::vector<hc> &pv, const hc *x, hc *xc) {
    for (int64_t c = 0; c < nagg * ne; c++) xc[c] = hc(0., 0.);
    for (int64_t i = 0; i < n; i++)
        for (int k = 0; k < ne; k++) xc[(size_t)agg[(size_t)i] * ne + k] += std::conj(pv[(size_t)i * ne + k]) * x[i];
}

struct CoarseBlocks {
    std::vector<int32_t> browptr, bcol;
    std::vector<hc> blocks;  // [nblk][ne][ne] row-major
};

// Galerkin blocks P_{a'}^H (A P_a) for every coupled pair of aggregates (src/MG.h:216-274), with the
// reference's evaluation order: row sum over the columns of aggregate a in CSR order, optional Dirac
// shift y = x - k*sum, then the dot over the rows of a' ascending.
// Multi-GPU: rows are this rank's aggregates (nagg of them); `agg` and `pv` cover the owned rows AND the
// halo columns, halo aggregates carrying ids >= nagg (nagg_cols in total).
static void galerkin(const HostCsr &A, bool shift, hc kshift, int ne, int64_t nagg, int64_t nagg_cols, const std::vector<int32_t> &agg,
                     const std::vector<int32_t> &ptr, const std::vector<int32_t> &mem, const std::vector<hc> &pv, CoarseBlocks &out) {
    const hc *val = reinterpret_cast<const hc *>(A.val_ri.data());
    std::vector<int32_t> mark((size_t)nagg_cols, -1), nbr;
    std::vector<hc> t((size_t)ne);
    out.browptr.assign((size_t)nagg + 1, 0);
    out.bcol.clear();
    out.blocks.clear();
    for (int64_t ap = 0; ap < nagg; ap++) {
        nbr.clear();
        if (shift) { mark[(size_t)ap] = (int32_t)ap; nbr.push_back((int32_t)ap); }
        for (int32_t m = ptr[(size_t)ap]; m < ptr[(size_t)ap + 1]; m++) {
            const int64_t i = mem[(size_t)m];
            for (int64_t l = A.rowptr[(size_t)i]; l < A.rowptr[(size_t)i + 1]; l++) {
                const int32_t a = agg[(size_t)A.col[(size_t)l]];
                if (mark[(size_t)a] != (int32_t)ap) { mark[(size_t)a] = (int32_t)ap; nbr.push_back(a); }
            }
        }
        std::sort(nbr.begin(), nbr.end());
        const size_t base = out.blocks.size();
        out.blocks.resize(base + nbr.size() * (size_t)ne * ne, hc(0., 0.));
        for (size_t q = 0; q < nbr.size(); q++) {
            const int32_t a = nbr[q];
            hc *blk = out.blocks.data() + base + q * (size_t)ne * ne;
            out.bcol.push_back(a);
            for (int32_t m = ptr[(size_t)ap]; m < ptr[(size_t)ap + 1]; m++) {
                const int64_t i = mem[(size_t)m];
                for (int k = 0; k < ne; k++) t[(size_t)k] = hc(0., 0.);
                for (int64_t l = A.rowptr[(size_t)i]; l < A.rowptr[(size_t)i + 1]; l++) {
                    const int64_t j = A.col[(size_t)l];
                    if (agg[(size_t)j] == a)
                        for (int k = 0; k < ne; k++) t[(size_t)k] += val[l] * pv[(size_t)j * ne + k];
                }
                for (int k = 0; k < ne; k++) {
                    hc y = t[(size_t)k];
                    if (shift) y = ((agg[(size_t)i] == a) ? pv[(size_t)i * ne + k] : hc(0., 0.)) - kshift * t[(size_t)k];
                    for (int kp = 0; kp < ne; kp++) blk[kp * ne + k] += std::conj(pv[(size_t)i * ne + kp]) * y;
                }
            }
        }
        out.browptr[(size_t)ap + 1] = out.browptr[(size_t)ap] + (int32_t)nbr.size();
    }
}

// block-CSR -> scalar CSR (host) so that the next level's Galerkin and an ne == 1 coarse operator
// can use the ELL path
static void blocks_to_csr(const CoarseBlocks &B, int ne, int64_t nagg, HostCsr &out) {
    const int64_t nc = nagg * ne;
    out.nrow = out.ncol = nc;
    out.rowptr.assign((size_t)nc + 1, 0);
    out.col.clear();
    out.val_ri.clear();
    int64_t p = 0;
    for (int64_t a = 0; a < nagg; a++)
        for (int kp = 0; kp < ne; kp++) {
            out.rowptr[(size_t)(a * ne + kp)] = p;
            for (int32_t b = B.browptr[(size_t)a]; b < B.browptr[(size_t)a + 1]; b++)
                for (int k = 0; k < ne; k++) {
                    out.col.push_back((int64_t)B.bcol[(size_t)b] * ne + k);
                    const hc v = B.blocks[(size_t)b * ne * ne + (size_t)kp * ne + k];
                    out.val_ri.push_back(v.real());
                    out.val_ri.push_back(v.imag());
                    p++;
                }
        }
    out.rowptr[(size_t)nc] = p;
}

// ------------------------------------------------------------------------------------------------
// device kernels of the cycle
// ------------------------------------------------------------------------------------------------

// (R x)[a*ne + k] = sum over the members of a (ascending) of conj(pv[i][k]) x[i]   (src/MG.h:366-383)
// one thread per coarse unknown: aggregates are small (2^d sites x dof), neighbouring threads read
// neighbouring aggregates, and the sum runs in the reference's order (bit-identical to the oracle).
__global__ void __launch_bounds__(256) restrict_kernel(int64_t nc, int ne, const int32_t *__restrict__ aptr,
                                                       const int32_t *__restrict__ amem, const cplx *__restrict__ pv,
                                                       const cplx *__restrict__ x, cplx *__restrict__ xc,
                                                       const int *__restrict__ skip, int skip_it) {
    if (skip && skip[0] < skip[1] + skip_it) return;  // {stop_at, base}: see gcr.hip DevState
    int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (c >= nc) return;
    int64_t a = c / ne;
    int k = (int)(c - a * ne);
    cplx s = make_double2(0., 0.);
    for (int32_t m = aptr[a]; m < aptr[a + 1]; m++) {
        int32_t i = amem[m];
        s = cadd(s, cconj_mul(pv[(int64_t)i * ne + k], x[i]));
    }
    xc[c] = s;
}

// x[i] = x[i] + damp * sum_k xc[agg[i]*ne+k] * pv[i][k]      (expand, src/MG.h:347-364, + src/MG.h:426)
__global__ void __launch_bounds__(256) expand_add_kernel(int64_t n, int ne, const int32_t *__restrict__ agg,
                                                         const cplx *__restrict__ pv, const cplx *__restrict__ xc,
                                                         cplx *__restrict__ x, cplx damp, int add,
                                                         const int *__restrict__ skip, int skip_it) {
    if (skip && skip[0] < skip[1] + skip_it) return;  // {stop_at, base}: see gcr.hip DevState
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const cplx *c = xc + (int64_t)agg[i] * ne;
    cplx s = make_double2(0., 0.);
    for (int k = 0; k < ne; k++) s = cadd(s, cmul(c[k], pv[i * ne + k]));
    x[i] = add ? cadd(x[i], cmul(damp, s)) : s;
}

// r = b - r   (r holds A x on entry)
__global__ void __launch_bounds__(256) residual_kernel(int64_t n, const cplx *__restrict__ b, cplx *__restrict__ r,
                                                       const int *__restrict__ skip, int skip_it) {
    if (skip && skip[0] < skip[1] + skip_it) return;  // {stop_at, base}: see gcr.hip DevState
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) r[i] = csub(b[i], r[i]);
}

// ------------------------------------------------------------------------------------------------
// hierarchy
// ------------------------------------------------------------------------------------------------
struct MgLevel {
    int64_t n = 0, nagg = 0;
    int ne = 0;
    Op *A = nullptr;
    bool owns_A = false;
    int32_t *d_agg = nullptr, *d_aptr = nullptr, *d_amem = nullptr;
    cplx *d_pv = nullptr;
    cplx *x = nullptr, *b = nullptr, *r = nullptr;  // work vectors (x, b: levels >= 1)
    GcrState *pre = nullptr, *post = nullptr, *coarse = nullptr;
    std::vector<int32_t> h_agg;
    std::vector<hc> h_pv;
};

struct MgState {
    std::vector<MgLevel> lev;
    double damping = 1.0;
};

template <typename T>
static int up(T **d, const T *h, size_t count) {
    *d = nullptr;
    hipError_t e = hipMalloc((void **)d, sizeof(T) * (count ? count : 1));
    if (e != hipSuccess) {
        set_error("hipMalloc of %zu bytes failed: %s", sizeof(T) * count, hipGetErrorString(e));
        return MGCR_ERR_ALLOC;
    }
    if (h && count) MGCR_HIP(hipMemcpy(*d, h, sizeof(T) * count, hipMemcpyHostToDevice));
    return MGCR_OK;
}

void mg_destroy(MgState *m) {
    if (!m) return;
    if (ctx().ready) hipStreamSynchronize(ctx().stream);
    for (MgLevel &L : m->lev) {
        hipFree(L.d_agg); hipFree(L.d_aptr); hipFree(L.d_amem); hipFree(L.d_pv);
        hipFree(L.x); hipFree(L.b); hipFree(L.r);
        gcr_state_destroy(L.pre); gcr_state_destroy(L.post); gcr_state_destroy(L.coarse);
        if (L.owns_A && L.A) {
            if (L.A->kind == OP_CSR) { csr_free(&L.A->csr); dist_free(L.A->dist); }
            if (L.A->kind == OP_BCSR) bcsr_free(&L.A->bcsr);
            delete static_cast<mgcr_op_s *>(L.A);
        }
    }
    delete m;
}

static int mg_create_host(Op *A, const mgcr_mg_param *p, MgState **out) {
    MGCR_CHECK(A && (A->kind == OP_CSR || A->kind == OP_DIRAC), MGCR_ERR_UNSUPPORTED,
               "mgcr_mg_create: the fine operator must be a Sparse or a DiracOp");
    MGCR_CHECK(p->ndim >= 1 && p->ndim <= 8 && p->n_vec >= 1 && p->vecs_ri && p->n_level >= 1 && p->n_level <= 6,
               MGCR_ERR_INVALID, "mgcr_mg_create: bad parameters");
    Op *base0 = A->kind == OP_DIRAC ? A->base : A;
    const CsrDev &A0 = base0->csr;
    const bool distributed = base0->dist != nullptr;
    MGCR_CHECK(distributed || A0.nrow == A0.ncol, MGCR_ERR_INVALID, "mgcr_mg_create: operator must be square");
    int64_t n = 1;
    int nblocked = 0;
    for (int d = 0; d < p->ndim; d++) { n *= p->dims[d]; nblocked += p->blocked[d] ? 1 : 0; }
    // distributed operator: the mesh describes THIS RANK's row block (e.g. its slab of planes)
    MGCR_CHECK(n == A0.nrow, MGCR_ERR_INVALID, "mgcr_mg_create: mesh has %lld points, operator has %lld (local) rows", (long long)n, (long long)A0.nrow);
    MGCR_CHECK(nblocked >= 1 && nblocked <= 4, MGCR_ERR_INVALID, "mgcr_mg_create: 1..4 dimensions can be blocked");

    MgState *m = new MgState();
    m->damping = p->damping;
    const int nlev = p->n_level + 1;
    m->lev.resize((size_t)nlev);
    int rc = MGCR_OK;
    HostCsr hA, hNext;
    rc = csr_download_host(A0, &hA);
    bool shift = A->kind == OP_DIRAC;
    hc kshift(A->k.x, A->k.y);
    int ndim = p->ndim, ne = p->n_vec;
    std::vector<int64_t> dims(p->dims, p->dims + ndim);
    std::vector<int32_t> blocked(p->blocked, p->blocked + ndim);
    std::vector<hc> vecs((size_t)ne * n);
    memcpy((void *)vecs.data(), p->vecs_ri, sizeof(hc) * (size_t)ne * n);
    m->lev[0].A = A;
    m->lev[0].n = n;
    DistCsr *dist = base0->dist;  // distribution of the current level's operator
    for (int l = 0; rc == MGCR_OK && l + 1 < nlev; l++) {
        MgLevel &L = m->lev[(size_t)l];
        L.ne = ne;
        L.nagg = lattice_aggregates(ndim, dims.data(), blocked.data(), p->subblock_dim, L.h_agg);
        if (L.nagg < 0) {  // assertm(dim[i] % subblock_dim == 0, ...) src/Mesh.h:245
            set_error("Dimension not exactly divisible by block size! (level %d)", l);
            rc = MGCR_ERR_INVALID;
            break;
        }
        std::vector<int32_t> ptr, mem;
        member_lists(L.n, L.nagg, L.h_agg, ptr, mem);
        build_prolongator(L.n, ne, L.nagg, ptr, mem, vecs.data(), L.h_pv);
        // aggregate ids / prolongator rows of the columns the Galerkin product sees
        int64_t nh = 0, agg_off = 0, nagg_glob = L.nagg, nagg_cols = L.nagg;
        std::vector<int32_t> agg_ext;                 // owned rows, then halo slots (ids >= L.nagg)
        std::vector<hc> pv_ext;
        std::vector<int64_t> ext_gid;                 // global id of extended aggregate L.nagg + k
        const std::vector<int32_t> *aggp = &L.h_agg;
        const std::vector<hc> *pvp = &L.h_pv;
        Comm *comm = nullptr;
        if (dist) {
            int rank = 0, nranks = 1;
            dist_sizes(dist, nullptr, &nh, nullptr, nullptr, &rank, &nranks);
            comm = dist_comm(dist);
            std::vector<double> cnt((size_t)nranks, 0.);
            cnt[(size_t)rank] = (double)L.nagg;
            rc = comm_allreduce_host_pub(comm, cnt.data(), nranks);
            if (rc != MGCR_OK) break;
            nagg_glob = 0;
            for (int r = 0; r < nranks; r++) { if (r == rank) agg_off = nagg_glob; nagg_glob += (int64_t)cnt[(size_t)r]; }
            const int w = 1 + 2 * ne;
            std::vector<double> own((size_t)L.n * w), halo((size_t)nh * w);
            for (int64_t i = 0; i < L.n; i++) {
                own[(size_t)i * w] = (double)(agg_off + L.h_agg[(size_t)i]);
                memcpy(&own[(size_t)i * w + 1], &L.h_pv[(size_t)i * ne], sizeof(hc) * (size_t)ne);
            }
            rc = dist_exchange_rows_host(dist, own.data(), w, halo.data());
            if (rc != MGCR_OK) break;
            for (int64_t h = 0; h < nh; h++) ext_gid.push_back((int64_t)halo[(size_t)h * w]);
            std::sort(ext_gid.begin(), ext_gid.end());
            ext_gid.erase(std::unique(ext_gid.begin(), ext_gid.end()), ext_gid.end());
            nagg_cols = L.nagg + (int64_t)ext_gid.size();
            agg_ext = L.h_agg;
            pv_ext = L.h_pv;
            agg_ext.resize((size_t)(L.n + nh));
            pv_ext.resize((size_t)(L.n + nh) * ne);
            for (int64_t h = 0; h < nh; h++) {
                int64_t g = (int64_t)halo[(size_t)h * w];
                agg_ext[(size_t)(L.n + h)] = (int32_t)(L.nagg + (std::lower_bound(ext_gid.begin(), ext_gid.end(), g) - ext_gid.begin()));
                memcpy(&pv_ext[(size_t)(L.n + h) * ne], &halo[(size_t)h * w + 1], sizeof(hc) * (size_t)ne);
            }
            aggp = &agg_ext;
            pvp = &pv_ext;
        }
        CoarseBlocks cb;
        galerkin(hA, shift, kshift, ne, L.nagg, nagg_cols, *aggp, ptr, mem, *pvp, cb);
        rc = up(&L.d_agg, L.h_agg.data(), (size_t)L.n);
        if (rc == MGCR_OK) rc = up(&L.d_aptr, ptr.data(), ptr.size());
        if (rc == MGCR_OK) rc = up(&L.d_amem, mem.data(), mem.size());
        if (rc == MGCR_OK) rc = up(&L.d_pv, reinterpret_cast<const cplx *>(L.h_pv.data()), (size_t)L.n * ne);
        if (rc == MGCR_OK) rc = up<cplx>(&L.r, nullptr, (size_t)L.n);
        if (rc != MGCR_OK) break;
        // coarse operator on the device
        MgLevel &C = m->lev[(size_t)l + 1];
        C.n = L.nagg * ne;
        Op *Ac = new mgcr_op_s();
        Ac->dim = Ac->nrow = C.n;
        if (dist) {
            // row block [agg_off*ne, (agg_off + nagg)*ne) of the global coarse matrix, global columns
            HostCsr g;
            blocks_to_csr(cb, ne, L.nagg, g);   // columns in extended local aggregate numbering
            for (int64_t &cidx : g.col) {
                int64_t a = cidx / ne, k = cidx % ne;
                int64_t gid = a < L.nagg ? agg_off + a : ext_gid[(size_t)(a - L.nagg)];
                cidx = gid * ne + k;
            }
            Ac->kind = OP_CSR;
            rc = dist_csr_create(comm, nagg_glob * ne, agg_off * ne, C.n, g.rowptr.data(), g.col.data(), g.val_ri.data(), Ac);
            if (rc == MGCR_OK) rc = csr_download_host(Ac->csr, &hNext);  // local numbering incl. halo slots, for the next level
        } else {
            blocks_to_csr(cb, ne, L.nagg, hNext);
            if (ne == 1) {
                Ac->kind = OP_CSR;
                rc = csr_build_device(C.n, C.n, hNext.rowptr.data(), hNext.col.data(), hNext.val_ri.data(), &Ac->csr);
            } else {
                Ac->kind = OP_BCSR;
                rc = bcsr_build_device((int32_t)L.nagg, (int32_t)L.nagg, ne, cb.browptr.data(), cb.bcol.data(),
                                       reinterpret_cast<const double *>(cb.blocks.data()), &Ac->bcsr);
            }
        }
        C.A = Ac;
        C.owns_A = true;
        if (rc != MGCR_OK) break;
        rc = up<cplx>(&C.x, nullptr, (size_t)C.n);
        if (rc == MGCR_OK) rc = up<cplx>(&C.b, nullptr, (size_t)C.n);
        if (rc != MGCR_OK) break;
        // smoothers of this level
        mgcr_gcr_param sp = p->smoother;
        sp.verbose = 0; sp.left_precond = sp.right_precond = nullptr; sp.flexible = 0; sp.profile_spmv = 0;
        sp.use_x0 = 0;
        rc = gcr_state_create(L.A, &sp, 1, &L.pre);
        sp.use_x0 = 1;
        if (rc == MGCR_OK) rc = gcr_state_create(L.A, &sp, 1, &L.post);
        if (rc != MGCR_OK) break;
        // next level: lattice of aggregates x ne, near-null vectors R v
        if (l + 2 < nlev) {
            std::vector<hc> nv((size_t)ne * C.n);
            for (int k = 0; k < ne; k++) host_restrict(L.n, ne, L.nagg, L.h_agg, L.h_pv, vecs.data() + (size_t)k * L.n, nv.data() + (size_t)k * C.n);
            vecs.swap(nv);
            std::vector<int64_t> d2;
            std::vector<int32_t> b2;
            for (int d = 0; d < ndim; d++)
                if (blocked[(size_t)d]) { d2.push_back(dims[(size_t)d] / p->subblock_dim); b2.push_back(1); }
            d2.push_back(ne);
            b2.push_back(0);
            dims.swap(d2);
            blocked.swap(b2);
            ndim = (int)dims.size();
            std::swap(hA, hNext);
            shift = false;
            dist = Ac->dist;
        }
    }
    if (rc == MGCR_OK) {
        MgLevel &Z = m->lev[(size_t)nlev - 1];
        mgcr_gcr_param cp = p->coarse;
        cp.verbose = 0; cp.left_precond = cp.right_precond = nullptr; cp.flexible = 0; cp.use_x0 = 0; cp.profile_spmv = 0;
        rc = gcr_state_create(Z.A, &cp, 1, &Z.coarse);
    }
    if (rc != MGCR_OK) { mg_destroy(m); return rc; }
    *out = m;
    return MGCR_OK;
}

int mg_level_setup_device(Op *A, int ndim, const int64_t *dims, const int32_t *blocked, int64_t sub, int ne, const cplx *d_vecs,
                          int64_t *nagg_out, int32_t **d_agg, int32_t **d_aptr, int32_t **d_amem, cplx **d_pv, Op **coarse,
                          bool want_next, cplx **d_vecs_next);

// Hierarchy built on the device (mg_setup.hip): single-GPU operators.
static int mg_create_device(Op *A, const mgcr_mg_param *p, MgState **out) {
    const CsrDev &A0 = (A->kind == OP_DIRAC ? A->base : A)->csr;
    MGCR_CHECK(A0.nrow == A0.ncol, MGCR_ERR_INVALID, "mgcr_mg_create: operator must be square");
    int64_t n = 1;
    int nblocked = 0;
    for (int d = 0; d < p->ndim; d++) { n *= p->dims[d]; nblocked += p->blocked[d] ? 1 : 0; }
    MGCR_CHECK(n == A0.nrow, MGCR_ERR_INVALID, "mgcr_mg_create: mesh has %lld points, operator has %lld rows", (long long)n, (long long)A0.nrow);
    MGCR_CHECK(nblocked >= 1 && nblocked <= 4, MGCR_ERR_INVALID, "mgcr_mg_create: 1..4 dimensions can be blocked");
    MgState *m = new MgState();
    m->damping = p->damping;
    const int nlev = p->n_level + 1;
    m->lev.resize((size_t)nlev);
    int ndim = p->ndim, ne = p->n_vec;
    std::vector<int64_t> dims(p->dims, p->dims + ndim);
    std::vector<int32_t> blocked(p->blocked, p->blocked + ndim);
    m->lev[0].A = A;
    m->lev[0].n = n;
    cplx *d_vecs = nullptr;
    int rc = up(&d_vecs, reinterpret_cast<const cplx *>(p->vecs_ri), (size_t)ne * n);
    for (int l = 0; rc == MGCR_OK && l + 1 < nlev; l++) {
        MgLevel &L = m->lev[(size_t)l];
        MgLevel &C = m->lev[(size_t)l + 1];
        L.ne = ne;
        Op *Ac = nullptr;
        cplx *d_next = nullptr;
        rc = mg_level_setup_device(L.A, ndim, dims.data(), blocked.data(), p->subblock_dim, ne, d_vecs, &L.nagg, &L.d_agg, &L.d_aptr,
                                   &L.d_amem, &L.d_pv, &Ac, l + 2 < nlev, &d_next);
        if (rc != MGCR_OK) break;
        C.A = Ac;
        C.owns_A = true;
        C.n = L.nagg * ne;
        rc = up<cplx>(&L.r, nullptr, (size_t)L.n);
        if (rc == MGCR_OK) rc = up<cplx>(&C.x, nullptr, (size_t)C.n);
        if (rc == MGCR_OK) rc = up<cplx>(&C.b, nullptr, (size_t)C.n);
        if (rc != MGCR_OK) break;
        mgcr_gcr_param sp = p->smoother;
        sp.verbose = 0; sp.left_precond = sp.right_precond = nullptr; sp.flexible = 0; sp.profile_spmv = 0;
        sp.use_x0 = 0;
        rc = gcr_state_create(L.A, &sp, 1, &L.pre);
        sp.use_x0 = 1;
        if (rc == MGCR_OK) rc = gcr_state_create(L.A, &sp, 1, &L.post);
        if (rc != MGCR_OK) break;
        hipFree(d_vecs);
        d_vecs = d_next;
        std::vector<int64_t> d2;
        std::vector<int32_t> b2;
        for (int d = 0; d < ndim; d++)
            if (blocked[(size_t)d]) { d2.push_back(dims[(size_t)d] / p->subblock_dim); b2.push_back(1); }
        d2.push_back(ne);
        b2.push_back(0);
        dims.swap(d2);
        blocked.swap(b2);
        ndim = (int)dims.size();
    }
    hipFree(d_vecs);
    if (rc == MGCR_OK) {
        MgLevel &Z = m->lev[(size_t)nlev - 1];
        mgcr_gcr_param cp = p->coarse;
        cp.verbose = 0; cp.left_precond = cp.right_precond = nullptr; cp.flexible = 0; cp.use_x0 = 0; cp.profile_spmv = 0;
        rc = gcr_state_create(Z.A, &cp, 1, &Z.coarse);
    }
    if (rc != MGCR_OK) { mg_destroy(m); return rc; }
    *out = m;
    return MGCR_OK;
}

// MGCR_MG_HOST_SETUP=1 forces the host set-up (used by the tests to compare the two);
// distributed operators always take it (their halo rows travel through host-level exchanges)
int mg_create(Op *A, const mgcr_mg_param *p, MgState **out) {
    MGCR_CHECK(A && (A->kind == OP_CSR || A->kind == OP_DIRAC), MGCR_ERR_UNSUPPORTED,
               "mgcr_mg_create: the fine operator must be a Sparse or a DiracOp");
    MGCR_CHECK(p->ndim >= 1 && p->ndim <= 8 && p->n_vec >= 1 && p->vecs_ri && p->n_level >= 1 && p->n_level <= 6,
               MGCR_ERR_INVALID, "mgcr_mg_create: bad parameters");
    const bool distributed = (A->kind == OP_DIRAC ? A->base : A)->dist != nullptr;
    const bool host = getenv("MGCR_MG_HOST_SETUP") && atoi(getenv("MGCR_MG_HOST_SETUP")) != 0;
    if (distributed || host) return mg_create_host(A, p, out);
    return mg_create_device(A, p, out);
}

static unsigned g256(int64_t n) { return (unsigned)((n + 255) / 256); }

int mg_restrict_raw(int64_t nc, int ne, const int32_t *aptr, const int32_t *amem, const cplx *pv, const cplx *x, cplx *xc) {
    hipLaunchKernelGGL(restrict_kernel, dim3(g256(nc)), dim3(256), 0, ctx().stream, nc, ne, aptr, amem, pv, x, xc, (const int *)nullptr, 0);
    MGCR_HIP(hipGetLastError());
    return MGCR_OK;
}

int mg_restrict(MgState *m, int l, const cplx *x, cplx *xc) {
    MgLevel &L = m->lev[(size_t)l];
    int64_t nc = L.nagg * L.ne;
    hipLaunchKernelGGL(restrict_kernel, dim3(g256(nc)), dim3(256), 0, ctx().stream, nc, L.ne, L.d_aptr, L.d_amem, L.d_pv, x, xc,
                       get_apply_skip().p, get_apply_skip().it);
    MGCR_HIP(hipGetLastError());
    return MGCR_OK;
}

int mg_expand(MgState *m, int l, const cplx *xc, cplx *x, bool add, double damping) {
    MgLevel &L = m->lev[(size_t)l];
    hipLaunchKernelGGL(expand_add_kernel, dim3(g256(L.n)), dim3(256), 0, ctx().stream, L.n, L.ne, L.d_agg, L.d_pv, xc, x,
                       make_double2(damping, 0.), add ? 1 : 0, get_apply_skip().p, get_apply_skip().it);
    MGCR_HIP(hipGetLastError());
    return MGCR_OK;
}

// Corrected cycle (report Algorithm 2; the structure of src/MG.h:405-430 with its defects fixed,
// see DESIGN.md):  x = S(b) from x0 = 0;  r = b - A x;  b_c = R r;  x_c = cycle(l+1) | coarsest GCR
// from x0 = 0;  x += damping * P x_c;  x = S(b, x0 = x).
static int mg_cycle(MgState *m, int l, const cplx *b, cplx *x) {
    MgLevel &L = m->lev[(size_t)l];
    const int nlev = (int)m->lev.size();
    if (l == nlev - 1) {
        MGCR_TRY(k_zero(x, L.n));
        return gcr_run(L.coarse, b, x, true, nullptr, 0, nullptr, nullptr);
    }
    MGCR_TRY(k_zero(x, L.n));
    MGCR_TRY(gcr_run(L.pre, b, x, true, nullptr, 0, nullptr, nullptr));
    MGCR_TRY(op_apply_raw(L.A, x, L.r, L.n));
    hipLaunchKernelGGL(residual_kernel, dim3(g256(L.n)), dim3(256), 0, ctx().stream, L.n, b, L.r, get_apply_skip().p, get_apply_skip().it);
    MGCR_HIP(hipGetLastError());
    MgLevel &C = m->lev[(size_t)l + 1];
    MGCR_TRY(mg_restrict(m, l, L.r, C.b));
    MGCR_TRY(mg_cycle(m, l + 1, C.b, C.x));
    MGCR_TRY(mg_expand(m, l, C.x, x, true, m->damping));
    return gcr_run(L.post, b, x, true, nullptr, 0, nullptr, nullptr);
}

int mg_apply(MgState *m, const cplx *f, cplx *y) { return mg_cycle(m, 0, f, y); }

}  // namespace mgcr

using namespace mgcr;
namespace mgcr {
int mg_restrict(MgState *m, int l, const cplx *x, cplx *xc);
int mg_expand(MgState *m, int l, const cplx *xc, cplx *x, bool add, double damping);
}  // namespace mgcr

#define LOCK() std::lock_guard<std::recursive_mutex> lk__(ctx().mtx)

extern "C" {

int mgcr_mg_create(mgcr_op_t A, const mgcr_mg_param *param, mgcr_op_t *out) {
    MGCR_TRY(require_ctx());
    MGCR_CHECK(A && param && out, MGCR_ERR_INVALID, "mgcr_mg_create: null argument");
    LOCK();
    MgState *m = nullptr;
    MGCR_TRY(mg_create(A, param, &m));
    mgcr_op_s *op = new mgcr_op_s();
    op->kind = OP_MG;
    op->dim = op->nrow = A->dim;
    op->mg = m;
    *out = op;
    return MGCR_OK;
}

int mgcr_mg_level_info(mgcr_op_t mg, int32_t level, int64_t *dim, int32_t *ne, int64_t *nagg) {
    MGCR_CHECK(mg && mg->kind == OP_MG, MGCR_ERR_INVALID, "not an MG operator");
    MGCR_CHECK(level >= 0 && level < (int32_t)mg->mg->lev.size(), MGCR_ERR_INVALID, "level %d out of range", level);
    const MgLevel &L = mg->mg->lev[(size_t)level];
    if (dim) *dim = L.n;
    if (ne) *ne = L.ne;
    if (nagg) *nagg = L.nagg;
    return MGCR_OK;
}

int mgcr_mg_restrict(mgcr_op_t mg, int32_t level, mgcr_vec_t fine, mgcr_vec_t coarse) {
    MGCR_TRY(require_ctx());
    MGCR_CHECK(mg && mg->kind == OP_MG && fine && coarse, MGCR_ERR_INVALID, "mgcr_mg_restrict: bad argument");
    MGCR_CHECK(level >= 0 && level + 1 < (int32_t)mg->mg->lev.size(), MGCR_ERR_INVALID, "level %d out of range", level);
    const MgLevel &L = mg->mg->lev[(size_t)level];
    MGCR_CHECK(fine->n == L.n && coarse->n == L.nagg * L.ne, MGCR_ERR_INVALID, "Lengths of two fields do not match!");
    LOCK();
    return mg_restrict(mg->mg, level, fine->d, coarse->d);
}

int mgcr_mg_expand(mgcr_op_t mg, int32_t level, mgcr_vec_t coarse, mgcr_vec_t fine) {
    MGCR_TRY(require_ctx());
    MGCR_CHECK(mg && mg->kind == OP_MG && fine && coarse, MGCR_ERR_INVALID, "mgcr_mg_expand: bad argument");
    MGCR_CHECK(level >= 0 && level + 1 < (int32_t)mg->mg->lev.size(), MGCR_ERR_INVALID, "level %d out of range", level);
    const MgLevel &L = mg->mg->lev[(size_t)level];
    MGCR_CHECK(fine->n == L.n && coarse->n == L.nagg * L.ne, MGCR_ERR_INVALID, "Lengths of two fields do not match!");
    LOCK();
    return mg_expand(mg->mg, level, coarse->d, fine->d, false, 1.0);
}

int mgcr_mg_level_op(mgcr_op_t mg, int32_t level, mgcr_op_t *out) {
    MGCR_CHECK(mg && mg->kind == OP_MG && out, MGCR_ERR_INVALID, "mgcr_mg_level_op: bad argument");
    MGCR_CHECK(level >= 0 && level < (int32_t)mg->mg->lev.size(), MGCR_ERR_INVALID, "level %d out of range", level);
    *out = static_cast<mgcr_op_s *>(mg->mg->lev[(size_t)level].A);
    return MGCR_OK;
}

int mgcr_mg_download_prolongator(mgcr_op_t mg, int32_t level, double *pv_ri, int32_t *agg) {
    MGCR_CHECK(mg && mg->kind == OP_MG, MGCR_ERR_INVALID, "not an MG operator");
    MGCR_CHECK(level >= 0 && level + 1 < (int32_t)mg->mg->lev.size(), MGCR_ERR_INVALID, "level %d out of range", level);
    const MgLevel &L = mg->mg->lev[(size_t)level];
    LOCK();
    if (ctx().ready) hipStreamSynchronize(ctx().stream);
    if (pv_ri) MGCR_HIP(hipMemcpy(pv_ri, L.d_pv, sizeof(cplx) * (size_t)L.n * L.ne, hipMemcpyDeviceToHost));
    if (agg) MGCR_HIP(hipMemcpy(agg, L.d_agg, sizeof(int32_t) * (size_t)L.n, hipMemcpyDeviceToHost));
    return MGCR_OK;
}

}  // extern "C"
