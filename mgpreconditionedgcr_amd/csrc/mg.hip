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
// The set-up (aggregates, Gram-Schmidt, Galerkin) runs on the device (mg_setup.hip); this file holds
// the hierarchy and the cycle.  Everything the solve touches per iteration (restrict, prolong+add,
// residual, smoothers, coarse solve) is HIP and is enqueued on the library stream without host
// round trips.
#include <algorithm>
#include <cmath>
#include <complex>

#include "internal.h"
#include "reduce.h"
#include "gcr_dev.h"

namespace mgcr {

typedef std::complex<double> hc;

// ------------------------------------------------------------------------------------------------
// device kernels of the cycle
// ------------------------------------------------------------------------------------------------

// (R x)[a*ne + k] = sum over the members of a (ascending) of conj(pv[i][k]) x[i]   (src/MG.h:366-383)
// one thread per coarse unknown: aggregates are small (2^d sites x dof), neighbouring threads read
// neighbouring aggregates, and the sum runs in the reference's order (bit-identical to the oracle).
__global__ void __launch_bounds__(256) restrict_kernel(int64_t nc, int ne, const int32_t *__restrict__ aptr,
                                                       const int32_t *__restrict__ amem, const cplx *__restrict__ pv,
                                                       const cplx *__restrict__ x, cplx *__restrict__ xc,
                                                       const int *__restrict__ skip, int skip_it, ResidualSel sel, int pvu, cplx pvc) {
    // pvu: every entry of the prolongator equals pvc (one constant near-null vector, aggregates of one size: MgLevel::pv_uniform) —
    // the value is then not read from memory; same products, same bits
    if (skip && skip[0] < skip[1] + skip_it) return;  // {stop_at, base}: see gcr.hip DevState
    int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (c >= nc) return;
    const cplx *ap = nullptr;
    cplx alpha = make_double2(0., 0.);
    if (sel.st) {   // x = the residual the smoother that just ran ended with: its slot depends on how many steps it took
        int k = sel.st->iter;
        x = sel.r[k < 0 ? 0 : k > LND ? LND : k];
        if (sel.last_it && k == sel.last_it) {   // ... and its last step's residual is formed here: r_prev - alpha Ap
            x = sel.r_prev;
            ap = sel.ap;
            alpha = *sel.alpha;
        }
    }
    int64_t a = c / ne;
    int k = (int)(c - a * ne);
    cplx s = make_double2(0., 0.);
    // members in batches of 8 (one 2^3 aggregate): the 8 index loads, then the 16 value loads, are issued together
    // instead of one dependent index -> value chain per member; the sum keeps the members' order
    const int32_t beg = aptr[a], end = aptr[a + 1];
    for (int32_t m0 = beg; m0 < end; m0 += 8) {
        int32_t idx[8];
#pragma unroll
        for (int u = 0; u < 8; u++) idx[u] = m0 + u < end ? amem[m0 + u] : -1;
        cplx pvv[8], xv[8];
        // a full batch whose members come in adjacent pairs (i, i + 1) with i even — the x-direction pair of a 2^d aggregate on a
        // mesh whose fastest dimension is even — is read with one 32-byte load per pair and array: neighbouring threads then
        // cover whole cache lines with ONE instruction instead of the two halves with two (restrict at 256^3: 200 -> 186-191 us; 16-byte loads of the member list on top changed nothing)
        const bool pairs = ne == 1 && idx[7] >= 0 && ((idx[0] | idx[2] | idx[4] | idx[6]) & 1) == 0 && idx[1] == idx[0] + 1 &&
                           idx[3] == idx[2] + 1 && idx[5] == idx[4] + 1 && idx[7] == idx[6] + 1;
        if (pairs) {
#pragma unroll
            for (int u = 0; u < 8; u += 2) {
                if (pvu) { pvv[u] = pvc; pvv[u + 1] = pvc; }
                else {
                    const double4 p2 = *reinterpret_cast<const double4 *>(pv + idx[u]);
                    pvv[u] = make_double2(p2.x, p2.y); pvv[u + 1] = make_double2(p2.z, p2.w);
                }
                const double4 x2 = *reinterpret_cast<const double4 *>(x + idx[u]);
                xv[u] = make_double2(x2.x, x2.y); xv[u + 1] = make_double2(x2.z, x2.w);
                if (ap) {
                    const double4 a2 = *reinterpret_cast<const double4 *>(ap + idx[u]);
                    xv[u] = csub(xv[u], cmul(alpha, make_double2(a2.x, a2.y)));
                    xv[u + 1] = csub(xv[u + 1], cmul(alpha, make_double2(a2.z, a2.w)));
                }
            }
        } else {
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int64_t i = idx[u] >= 0 ? idx[u] : 0;
                pvv[u] = pvu ? pvc : pv[i * ne + k];
                xv[u] = x[i];
                if (ap) xv[u] = csub(xv[u], cmul(alpha, ap[i]));   // same expression as xr_update_kernel: the same bits
            }
        }
#pragma unroll
        for (int u = 0; u < 8; u++)
            if (idx[u] >= 0) s = cadd(s, cconj_mul(pvv[u], xv[u]));
    }
    xc[c] = s;
}

// x[i] = x[i] + damp * sum_k xc[agg[i]*ne+k] * pv[i][k]      (expand, src/MG.h:347-364, + src/MG.h:426)
// pend.st != nullptr: the x to add to was never written — it is the pre-smoother's pending update
// sum_j coef[j] v_j[i] from x0 = 0 (gcr.hip flush_x_kernel's sum, same order), formed here on the fly
__global__ void __launch_bounds__(256) expand_add_kernel(int64_t n, int ne, const int32_t *__restrict__ agg,
                                                         const cplx *__restrict__ pv, const cplx *__restrict__ xc,
                                                         cplx *__restrict__ x, cplx damp, int add,
                                                         const int *__restrict__ skip, int skip_it, PendingX pend, int pvu, cplx pvc) {
    if (skip && skip[0] < skip[1] + skip_it) return;  // {stop_at, base}: see gcr.hip DevState
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const cplx *c = xc + (int64_t)agg[i] * ne;
    cplx s = make_double2(0., 0.);
    if (pvu) s = cadd(s, cmul(c[0], pvc));   // (restrict_kernel: pvu)
    else
    for (int k = 0; k < ne; k++) s = cadd(s, cmul(c[k], pv[i * ne + k]));
    if (pend.st) {
        const int np = pend.st->npend;
        cplx xv = make_double2(0., 0.);
        if (np == 2) {          // a smoother's 2 sweeps (3 vectors at most): loads issued together; same sum, same order
            const cplx v0 = pend.v[0][i], v1 = pend.v[1][i];
            xv = cadd(xv, cmul(pend.coef[0], v0));
            xv = cadd(xv, cmul(pend.coef[1], v1));
        } else if (np == 3) {
            const cplx v0 = pend.v[0][i], v1 = pend.v[1][i], v2 = pend.v[2][i];
            xv = cadd(xv, cmul(pend.coef[0], v0));
            xv = cadd(xv, cmul(pend.coef[1], v1));
            xv = cadd(xv, cmul(pend.coef[2], v2));
        } else {
            for (int j = 0; j < np && j < LND; j++) xv = cadd(xv, cmul(pend.coef[j], pend.v[j][i]));
        }
        x[i] = cadd(xv, cmul(damp, s));
        return;
    }
    x[i] = add ? cadd(x[i], cmul(damp, s)) : s;
}


// flag[0] = 0 as soon as an entry of pv differs (in its bits) from pv[0]
__global__ void __launch_bounds__(256) pv_uniform_kernel(int64_t n, const cplx *__restrict__ pv, int *flag) {
    const cplx v0 = pv[0];
    bool same = true;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const cplx v = pv[i];
        same = same && __double_as_longlong(v.x) == __double_as_longlong(v0.x) && __double_as_longlong(v.y) == __double_as_longlong(v0.y);
    }
    if (!same) flag[0] = 0;
}

// ------------------------------------------------------------------------------------------------
// hierarchy
// ------------------------------------------------------------------------------------------------
static bool recurrence_residual_enabled() {
    static const bool on = !(getenv("MGCR_MG_RECURRENCE_RESIDUAL") && atoi(getenv("MGCR_MG_RECURRENCE_RESIDUAL")) == 0);
    return on;
}

struct MgLevel {
    int64_t n = 0, nagg = 0;
    int ne = 0;
    Op *A = nullptr;
    bool owns_A = false;
    int32_t *d_agg = nullptr, *d_aptr = nullptr, *d_amem = nullptr;
    cplx *d_pv = nullptr;
    bool pv_uniform = false;    // ne == 1 and every entry of d_pv has the bits of pv_value: restrict / expand do not read d_pv
    cplx pv_value = {0., 0.};
    cplx *x = nullptr, *b = nullptr, *r = nullptr;  // work vectors (x, b: levels >= 1)
    GcrState *pre = nullptr, *post = nullptr, *coarse = nullptr;
    cplx *inv = nullptr;   // coarsest level, direct solve (dense.hip): the inverse of A, n x n
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
        hipFree(L.x); hipFree(L.b); hipFree(L.r); hipFree(L.inv);
        gcr_state_destroy(L.pre); gcr_state_destroy(L.post); gcr_state_destroy(L.coarse);
        if (L.owns_A && L.A) {
            if (L.A->kind == OP_CSR) { csr_free(&L.A->csr); dist_free(L.A->dist); }
            if (L.A->kind == OP_BCSR) { bcsr_free(&L.A->bcsr); dist_free(L.A->dist); }
            delete static_cast<mgcr_op_s *>(L.A);
        }
    }
    delete m;
}

int mg_level_setup_device(Op *A, int ndim, const int64_t *dims, const int32_t *blocked, int64_t sub, int ne, const cplx *d_vecs,
                          int64_t *nagg_out, int32_t **d_agg, int32_t **d_aptr, int32_t **d_amem, cplx **d_pv, Op **coarse,
                          bool want_next, cplx **d_vecs_next);

// Hierarchy built on the device (mg_setup.hip).  For a distributed operator the mesh describes THIS
// RANK's row block (e.g. its slab of planes) and the call is collective.
static int mg_create_device(Op *A, const mgcr_mg_param *p, MgState **out) {
    const Op *base0 = A->kind == OP_DIRAC ? A->base : A;
    const bool blockop = base0->kind == OP_BCSR;  // HierarchicalSparse as the fine operator (e.g. aggregates of block rows)
    const int64_t nrow0 = blockop ? (int64_t)base0->bcsr.nbrow * base0->bcsr.bs : base0->csr.nrow;
    const int64_t ncol0 = blockop ? (int64_t)base0->bcsr.nbcol * base0->bcsr.bs : base0->csr.ncol;
    MGCR_CHECK(base0->dist || nrow0 == ncol0, MGCR_ERR_INVALID, "mgcr_mg_create: operator must be square");
    int64_t n = 1;
    int nblocked = 0;
    for (int d = 0; d < p->ndim; d++) { n *= p->dims[d]; nblocked += p->blocked[d] ? 1 : 0; }
    MGCR_CHECK(n == nrow0, MGCR_ERR_INVALID, "mgcr_mg_create: mesh has %lld points, operator has %lld rows", (long long)n, (long long)nrow0);
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
        if (ne == 1 && L.n > 0 && !(getenv("MGCR_MG_UNIFORM_PV") && atoi(getenv("MGCR_MG_UNIFORM_PV")) == 0)) {
            // one constant near-null vector over aggregates of one size: every entry of the prolongator is the same number — the cycle's
            // transfer kernels then skip that stream (a vector's worth of reads each)
            int *d_flag = nullptr, h_flag = 1;
            rc = up<int>(&d_flag, &h_flag, 1);
            if (rc != MGCR_OK) break;
            hipLaunchKernelGGL(pv_uniform_kernel, dim3(2048), dim3(256), 0, ctx().stream, L.n, (const cplx *)L.d_pv, d_flag);
            hipError_t e = hipMemcpyAsync(&h_flag, d_flag, sizeof(int), hipMemcpyDeviceToHost, ctx().stream);
            if (e == hipSuccess) e = hipMemcpyAsync(&L.pv_value, L.d_pv, sizeof(cplx), hipMemcpyDeviceToHost, ctx().stream);
            if (e == hipSuccess) e = hipStreamSynchronize(ctx().stream);
            hipFree(d_flag);
            if (e != hipSuccess) { set_error("mg set-up: prolongator scan failed"); rc = MGCR_ERR_HIP; break; }
            L.pv_uniform = h_flag != 0;
        }
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
        if (rc == MGCR_OK) gcr_set_keep_pending(L.pre, true);   // mg_cycle writes the pre-smoother's x together with + P x_c
        if (rc == MGCR_OK && recurrence_residual_enabled()) gcr_set_defer_residual(L.pre, true);   // restrict_kernel forms the last sweep's residual
        sp.use_x0 = 1;
        if (rc == MGCR_OK) rc = gcr_state_create(L.A, &sp, 1, &L.post);
        if (rc == MGCR_OK) gcr_set_discard_residual(L.post, true);   // the cycle only takes x from its post-smoother
        if (rc == MGCR_OK) gcr_set_bnorm_source(L.post, L.pre);      // ... and its |b|^2 from the pre-smoother: same b (gcr.hip bnorm_src)
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
        if (rc == MGCR_OK) gcr_set_discard_residual(Z.coarse, true);
        // opt-in: a small coarsest level is inverted once and solved by one mat-vec per cycle (dense.hip)
        const Op *zb = Z.A->kind == OP_DIRAC ? Z.A->base : Z.A;
        if (rc == MGCR_OK && p->coarse_direct_rows > 0 && Z.n <= p->coarse_direct_rows && Z.n <= DENSE_MAX_ROWS && !zb->dist)
            rc = dense_inverse_of(Z.A, Z.n, &Z.inv);
    }
    if (rc != MGCR_OK) { mg_destroy(m); return rc; }
    *out = m;
    return MGCR_OK;
}

int mg_create(Op *A, const mgcr_mg_param *p, MgState **out) {
    MGCR_CHECK(A && (A->kind == OP_CSR || A->kind == OP_DIRAC || A->kind == OP_BCSR), MGCR_ERR_UNSUPPORTED,
               "mgcr_mg_create: the fine operator must be a Sparse, a DiracOp or a HierarchicalSparse");
    MGCR_CHECK(p->ndim >= 1 && p->ndim <= 8 && p->n_vec >= 1 && p->vecs_ri && p->n_level >= 1 && p->n_level <= 6,
               MGCR_ERR_INVALID, "mgcr_mg_create: bad parameters");
    return mg_create_device(A, p, out);
}

static unsigned g256(int64_t n) { return (unsigned)((n + 255) / 256); }

int mg_restrict_raw(int64_t nc, int ne, const int32_t *aptr, const int32_t *amem, const cplx *pv, const cplx *x, cplx *xc) {
    hipLaunchKernelGGL(restrict_kernel, dim3(g256(nc)), dim3(256), 0, ctx().stream, nc, ne, aptr, amem, pv, x, xc, (const int *)nullptr, 0, ResidualSel{}, 0,
                       make_double2(0., 0.));
    MGCR_HIP(hipGetLastError());
    return MGCR_OK;
}

int mg_restrict(MgState *m, int l, const cplx *x, cplx *xc, const ResidualSel *sel) {
    MgLevel &L = m->lev[(size_t)l];
    int64_t nc = L.nagg * L.ne;
    hipLaunchKernelGGL(restrict_kernel, dim3(g256(nc)), dim3(256), 0, ctx().stream, nc, L.ne, L.d_aptr, L.d_amem, L.d_pv, x, xc,
                       get_apply_skip().p, get_apply_skip().it, sel ? *sel : ResidualSel{}, L.pv_uniform ? 1 : 0, L.pv_value);
    MGCR_HIP(hipGetLastError());
    return MGCR_OK;
}

int mg_expand(MgState *m, int l, const cplx *xc, cplx *x, bool add, double damping, const PendingX *pend) {
    MgLevel &L = m->lev[(size_t)l];
    hipLaunchKernelGGL(expand_add_kernel, dim3(g256(L.n)), dim3(256), 0, ctx().stream, L.n, L.ne, L.d_agg, L.d_pv, xc, x,
                       make_double2(damping, 0.), add ? 1 : 0, get_apply_skip().p, get_apply_skip().it, pend ? *pend : PendingX{}, L.pv_uniform ? 1 : 0,
                       L.pv_value);
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
        if (L.inv) return dense_apply(L.inv, L.n, b, x);
        return gcr_run_from_zero(L.coarse, b, x);
    }
    MGCR_TRY(gcr_run_from_zero(L.pre, b, x));
    // a lean pre-smoother leaves its x unwritten (x = sum of two or three scaled vectors it still holds): the
    // prolongation kernel below forms it on the fly while adding the coarse-grid correction — x is written once
    PendingX pend;
    bool x_pending = gcr_take_pending(L.pre, &pend);
    MgLevel &C = m->lev[(size_t)l + 1];
    // The residual to restrict is the one the pre-smoother's recurrence ended with (r_k = r_{k-1} - alpha A p_{k-1}: b - A x
    // up to rounding) — no further pass over A, x and b.  Solvers with the literal preconditioner hooks, or of more
    // than 16 sweeps, do not offer it: b - A x is then recomputed (one pass for a Sparse, b enters the SpMV epilogue).
    ResidualSel sel;
    if (recurrence_residual_enabled() && gcr_last_residual(L.pre, &sel)) {
        MGCR_TRY(mg_restrict(m, l, nullptr, C.b, &sel));
    } else {
        if (x_pending) {   // b - A x needs x after all
            MGCR_TRY(gcr_flush_pending(pend, x, L.n));
            x_pending = false;
        }
        MGCR_TRY(op_residual_raw(L.A, x, b, L.r, L.n));
        MGCR_TRY(mg_restrict(m, l, L.r, C.b, nullptr));
    }
    MGCR_TRY(mg_cycle(m, l + 1, C.b, C.x));
    MGCR_TRY(mg_expand(m, l, C.x, x, true, m->damping, x_pending ? &pend : nullptr));
    return gcr_run(L.post, b, x, true, nullptr, 0, nullptr, nullptr);
}

int mg_apply(MgState *m, const cplx *f, cplx *y) { return mg_cycle(m, 0, f, y); }

}  // namespace mgcr

using namespace mgcr;
namespace mgcr {
int mg_restrict(MgState *m, int l, const cplx *x, cplx *xc, const ResidualSel *sel);
int mg_expand(MgState *m, int l, const cplx *xc, cplx *x, bool add, double damping, const PendingX *pend);
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
    return mg_restrict(mg->mg, level, fine->d, coarse->w(), nullptr);
}

int mgcr_mg_expand(mgcr_op_t mg, int32_t level, mgcr_vec_t coarse, mgcr_vec_t fine) {
    MGCR_TRY(require_ctx());
    MGCR_CHECK(mg && mg->kind == OP_MG && fine && coarse, MGCR_ERR_INVALID, "mgcr_mg_expand: bad argument");
    MGCR_CHECK(level >= 0 && level + 1 < (int32_t)mg->mg->lev.size(), MGCR_ERR_INVALID, "level %d out of range", level);
    const MgLevel &L = mg->mg->lev[(size_t)level];
    MGCR_CHECK(fine->n == L.n && coarse->n == L.nagg * L.ne, MGCR_ERR_INVALID, "Lengths of two fields do not match!");
    LOCK();
    return mg_expand(mg->mg, level, coarse->d, fine->w(), false, 1.0, nullptr);
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
