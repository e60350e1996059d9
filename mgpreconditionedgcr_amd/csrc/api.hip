// Operator / solver part of the C ABI (include/mgcr.h).
#include <algorithm>
#include <numeric>

#include <cstring>

#include "internal.h"

using namespace mgcr;

#define LOCK() std::lock_guard<std::recursive_mutex> lk__(ctx().mtx)

extern "C" {

int mgcr_csr_create(int64_t nrow, int64_t ncol, const int64_t *rowptr, const int64_t *col, const double *val_ri,
                    mgcr_op_t *out) {
    MGCR_TRY(require_ctx());
    MGCR_CHECK(out && rowptr && (col || rowptr[nrow] == 0) && (val_ri || rowptr[nrow] == 0), MGCR_ERR_INVALID,
               "mgcr_csr_create: null argument");
    LOCK();
    mgcr_op_s *op = new mgcr_op_s();
    op->kind = OP_CSR;
    op->dim = ncol;
    op->nrow = nrow;
    int rc = csr_build_device(nrow, ncol, rowptr, col, val_ri, &op->csr);
    if (rc != MGCR_OK) { delete op; return rc; }
    *out = op;
    return MGCR_OK;
}

int mgcr_csr_replace(mgcr_op_t op, int64_t nrow, int64_t ncol, const int64_t *rowptr, const int64_t *col, const double *val_ri) {
    MGCR_TRY(require_ctx());
    MGCR_CHECK(op && rowptr && (col || rowptr[nrow] == 0) && (val_ri || rowptr[nrow] == 0), MGCR_ERR_INVALID, "mgcr_csr_replace: null argument");
    MGCR_CHECK(op->kind == OP_CSR && !op->dist, MGCR_ERR_INVALID, "mgcr_csr_replace: not a (single-GPU) Sparse");
    LOCK();
    MGCR_HIP(hipStreamSynchronize(ctx().stream));   // nothing queued may still read the old matrix
    CsrDev fresh;
    MGCR_TRY(csr_build_device(nrow, ncol, rowptr, col, val_ri, &fresh));
    csr_free(&op->csr);
    op->csr = fresh;
    op->dim = ncol;
    op->nrow = nrow;
    return MGCR_OK;
}

int mgcr_dirac_create(mgcr_op_t csr, const double k_ri[2], mgcr_op_t *out) {
    MGCR_TRY(require_ctx());
    MGCR_CHECK(csr && k_ri && out, MGCR_ERR_INVALID, "mgcr_dirac_create: null argument");
    MGCR_CHECK(csr->kind == OP_CSR, MGCR_ERR_INVALID, "mgcr_dirac_create: DiracOp wraps a Sparse (CSR) operator");
    // (the row block of a distributed Sparse has its halo columns appended: square means dim == nrow there)
    MGCR_CHECK(csr->dist ? csr->dim == csr->nrow : csr->csr.nrow == csr->csr.ncol, MGCR_ERR_INVALID, "mgcr_dirac_create: matrix must be square");
    LOCK();
    mgcr_op_s *op = new mgcr_op_s();
    op->kind = OP_DIRAC;
    op->dim = csr->dim;
    op->nrow = csr->nrow;
    op->comm = csr->comm;
    op->base = csr;
    op->k = make_double2(k_ri[0], k_ri[1]);
    *out = op;
    return MGCR_OK;
}

int mgcr_dirac_set_k(mgcr_op_t dirac, const double k_ri[2]) {
    MGCR_CHECK(dirac && k_ri && dirac->kind == OP_DIRAC, MGCR_ERR_INVALID, "mgcr_dirac_set_k: not a DiracOp");
    LOCK();
    dirac->k = make_double2(k_ri[0], k_ri[1]);
    return MGCR_OK;
}

int mgcr_bcsr_create(int32_t nbrow, int32_t nbcol, int32_t bs, const int32_t *browptr, const int32_t *bcol,
                     const double *blocks_ri, mgcr_op_t *out) {
    MGCR_TRY(require_ctx());
    MGCR_CHECK(out && browptr && bcol && blocks_ri, MGCR_ERR_INVALID, "mgcr_bcsr_create: null argument");
    LOCK();
    mgcr_op_s *op = new mgcr_op_s();
    op->kind = OP_BCSR;
    int rc = bcsr_build_device(nbrow, nbcol, bs, browptr, bcol, blocks_ri, &op->bcsr);
    if (rc != MGCR_OK) { delete op; return rc; }
    op->dim = (int64_t)nbcol * bs;  // this->dim = block_cols * sub_dim, src/HierarchicalSparse.h:62
    op->nrow = (int64_t)nbrow * bs;
    *out = op;
    return MGCR_OK;
}

// HierarchicalSparse constructor (src/HierarchicalSparse.h:58-98): sort the triplets by
// row*nbcol+col, keep duplicates.  Unlike the reference we accept empty block-rows and a first
// triplet outside row 0 (its row counter would mis-assign those), and the sort is
// stable.
int mgcr_bcsr_create_from_triplets(int32_t nbrow, int32_t nbcol, int32_t bs, int32_t ntriplets, const int32_t *rows,
                                   const int32_t *cols, const double *blocks_ri, mgcr_op_t *out) {
    MGCR_CHECK(out && rows && cols && blocks_ri && ntriplets >= 0, MGCR_ERR_INVALID,
               "mgcr_bcsr_create_from_triplets: bad argument");
    for (int32_t t = 0; t < ntriplets; t++)
        MGCR_CHECK(rows[t] >= 0 && rows[t] < nbrow && cols[t] >= 0 && cols[t] < nbcol, MGCR_ERR_INVALID,
                   "triplet %d: block index (%d,%d) out of range", t, rows[t], cols[t]);
    std::vector<int32_t> order((size_t)ntriplets);
    std::iota(order.begin(), order.end(), 0);
    std::stable_sort(order.begin(), order.end(), [&](int32_t a, int32_t b) {
        return (int64_t)rows[a] * nbcol + cols[a] < (int64_t)rows[b] * nbcol + cols[b];
    });
    std::vector<int32_t> browptr((size_t)nbrow + 1, 0), bcol((size_t)ntriplets);
    std::vector<double> blocks((size_t)ntriplets * bs * bs * 2);
    const size_t bsz = (size_t)bs * bs * 2;
    for (int32_t t = 0; t < ntriplets; t++) {
        int32_t s = order[(size_t)t];
        bcol[(size_t)t] = cols[s];
        std::copy(blocks_ri + (size_t)s * bsz, blocks_ri + (size_t)(s + 1) * bsz, blocks.begin() + (size_t)t * bsz);
        browptr[(size_t)rows[s] + 1]++;
    }
    for (int32_t r = 0; r < nbrow; r++) browptr[(size_t)r + 1] += browptr[(size_t)r];
    return mgcr_bcsr_create(nbrow, nbcol, bs, browptr.data(), bcol.data(), blocks.data(), out);
}

int mgcr_op_destroy(mgcr_op_t op) {
    if (!op) return MGCR_OK;
    LOCK();
    if (ctx().ready) hipStreamSynchronize(ctx().stream);
    switch (op->kind) {
        case OP_CSR: csr_free(&op->csr); dist_free(op->dist); break;
        case OP_BCSR: bcsr_free(&op->bcsr); dist_free(op->dist); break;
        case OP_GCR: gcr_state_destroy(op->gcr); break;
        case OP_MG: mg_destroy(op->mg); break;
        default: break;  // OP_DIRAC borrows its Sparse (src/Operator.h:117,555-560)
    }
    delete op;
    return MGCR_OK;
}

int64_t mgcr_op_dim(mgcr_op_t op) { return op ? op->dim : -1; }
int64_t mgcr_op_nrow(mgcr_op_t op) { return op ? op->nrow : -1; }
int64_t mgcr_op_nnz(mgcr_op_t op) {
    if (!op) return -1;
    switch (op->kind) {
        case OP_CSR: return op->csr.nnz;
        case OP_DIRAC: return op->base->csr.nnz;
        case OP_BCSR: return (int64_t)op->bcsr.nblocks * op->bcsr.bs * op->bcsr.bs;  // src/HierarchicalSparse.h:31
        default: return -1;
    }
}

int mgcr_op_storage_format(mgcr_op_t op, int32_t *format, int32_t *n_patterns) {
    MGCR_CHECK(op, MGCR_ERR_INVALID, "null operator");
    const Op *o = op->kind == OP_DIRAC ? op->base : op;
    MGCR_CHECK(o->kind == OP_CSR, MGCR_ERR_UNSUPPORTED, "mgcr_op_storage_format: not a Sparse");
    if (format) *format = csr_stencil_active(o->csr) ? 3 : o->csr.pat_mode;
    if (n_patterns) *n_patterns = csr_stencil_active(o->csr) ? o->csr.sten_ns : o->csr.npat;
    return MGCR_OK;
}

int mgcr_op_ell_layout(mgcr_op_t op, int32_t *ell_width, int32_t *lanes, int64_t *tail_rows, int64_t *reach, int32_t *tail_chunk_cap,
                       int32_t *x_window) {
    MGCR_CHECK(op, MGCR_ERR_INVALID, "null operator");
    const Op *o = op->kind == OP_DIRAC ? op->base : op;
    MGCR_CHECK(o->kind == OP_CSR, MGCR_ERR_UNSUPPORTED, "mgcr_op_ell_layout: not a Sparse");
    if (ell_width) *ell_width = o->csr.W;
    if (lanes) *lanes = o->csr.L;
    if (tail_rows) *tail_rows = o->csr.n_tail_rows;
    if (reach) *reach = o->csr.reach;
    if (tail_chunk_cap) *tail_chunk_cap = TAIL_CAP;
    if (x_window) *x_window = o->csr.win_h;
    return MGCR_OK;
}

int mgcr_op_xr_fuse_kind(mgcr_op_t op, int32_t *kind) {
    MGCR_CHECK(op && kind, MGCR_ERR_INVALID, "mgcr_op_xr_fuse_kind: null argument");
    const Op *o = op->kind == OP_DIRAC ? op->base : op;
    MGCR_CHECK(o->kind == OP_CSR, MGCR_ERR_UNSUPPORTED, "mgcr_op_xr_fuse_kind: not a Sparse");
    *kind = csr_fusable(o->csr, o->dist) ? csr_xr_fuse_kind(o->csr, o->dist) : 0;
    return MGCR_OK;
}

int mgcr_op_halo_kind(mgcr_op_t op, int32_t *kind) {
    MGCR_CHECK(op && kind, MGCR_ERR_INVALID, "mgcr_op_halo_kind: null argument");
    const Op *o = op->kind == OP_DIRAC ? op->base : op;
    MGCR_CHECK(o->dist, MGCR_ERR_UNSUPPORTED, "mgcr_op_halo_kind: not a distributed Sparse");
    *kind = dist_halo_kind(o->dist);
    return MGCR_OK;
}

int mgcr_set_option(const char *name, int value, int *previous) {
    MGCR_CHECK(name, MGCR_ERR_INVALID, "null option name");
    if (!strcmp(name, "spmv_part")) {   // measurement aid: 0 whole apply, 1 ELL slab only, 2 CSR tail only (spmv.hip)
        const int p = set_spmv_part(value);
        if (previous) *previous = p;
        return MGCR_OK;
    }
    bool prev;
    if (!strcmp(name, "pattern_storage")) prev = set_patterns_enabled(value != 0);
    else if (!strcmp(name, "stencil_storage")) prev = set_stencil_enabled(value != 0);
    else if (!strcmp(name, "lean_cycles")) prev = set_lean_enabled(value != 0);
    else if (!strcmp(name, "fused_apply")) prev = set_fuse_enabled(value != 0);
    else if (!strcmp(name, "graph_replay")) prev = set_graph_enabled(value != 0);
    else if (!strcmp(name, "resident_solver")) prev = set_resident_enabled(value != 0);
    else if (!strcmp(name, "step_build")) prev = set_stepbuild_enabled(value != 0);
    else if (!strcmp(name, "halo_split")) prev = set_halo_split(value != 0);
    else if (!strcmp(name, "pw_tail")) prev = set_pw_tail_enabled(value != 0);
    else { set_error("mgcr_set_option: unknown option '%s'", name); return MGCR_ERR_INVALID; }
    if (previous) *previous = prev ? 1 : 0;
    return MGCR_OK;
}

int mgcr_stat(const char *name, int64_t *value) {
    MGCR_CHECK(name && value, MGCR_ERR_INVALID, "mgcr_stat: null argument");
    if (!strcmp(name, "resident_solves")) *value = resident_solve_count();
    else if (!strcmp(name, "step_build_launches")) *value = stepbuild_launch_count();
    else if (!strcmp(name, "small_solves")) *value = gcr_small_solve_count();
    else if (!strcmp(name, "one_launch_fallbacks")) *value = gcr_fallback_count();
    else if (!strcmp(name, "halo_split_exchanges")) *value = dist_halo_split_count();
    else if (!strcmp(name, "pw_tail_folds")) *value = comm_pw_tail_count();
    else { set_error("mgcr_stat: unknown counter '%s'", name); return MGCR_ERR_INVALID; }
    return MGCR_OK;
}

int mgcr_selftest_coherence(int32_t steps, int32_t coherent, int64_t *rows_wrong) {
    MGCR_TRY(require_ctx());
    LOCK();
    return coherence_selftest(steps, coherent, rows_wrong);
}

int mgcr_op_stored_bytes(mgcr_op_t op, int64_t *matrix_bytes, int32_t *ell_width, int64_t *tail_nnz) {
    MGCR_CHECK(op, MGCR_ERR_INVALID, "null operator");
    const Op *o = op->kind == OP_DIRAC ? op->base : op;
    if (o->kind == OP_CSR) {
        const CsrDev &A = o->csr;
        int64_t slab = (int64_t)A.nchunk * A.npad * A.L;
        int64_t ell_bytes = slab * (A.ell_val_re ? 12 : 20);
        if (csr_stencil_active(A)) ell_bytes = (A.npad / 64) * A.sten_stride * 8 + A.sten_kernel_ns * 20;   // presence words + slot table
        else if (A.pat_mode == 1) ell_bytes = A.npad * 2 + (int64_t)A.npat * A.W * (A.pat_real ? 12 : 20);
        else if (A.pat_mode == 2) ell_bytes = A.npad * 2 + (int64_t)A.npat * A.W * 4 + slab * (A.ell_val_re ? 8 : 16);
        if (matrix_bytes) *matrix_bytes = ell_bytes + A.tail_nnz * 20 + A.n_tail_rows * 8 + (A.n_tail_rows ? 4 : 0);
        if (ell_width) *ell_width = A.nchunk * A.L;
        if (tail_nnz) *tail_nnz = A.tail_nnz;
        return MGCR_OK;
    }
    if (o->kind == OP_BCSR) {
        const BcsrDev &B = o->bcsr;
        if (matrix_bytes) *matrix_bytes = (int64_t)B.nblocks * (16LL * B.bs * B.bs + 4) + ((int64_t)B.nbrow + 1) * 4;
        if (ell_width) *ell_width = 0;
        if (tail_nnz) *tail_nnz = 0;
        return MGCR_OK;
    }
    set_error("mgcr_op_stored_bytes: not a matrix operator");
    return MGCR_ERR_UNSUPPORTED;
}

int mgcr_op_apply(mgcr_op_t op, mgcr_vec_t x, mgcr_vec_t y) {
    MGCR_TRY(require_ctx());
    MGCR_CHECK(op && x && y, MGCR_ERR_INVALID, "mgcr_op_apply: null argument");
    MGCR_CHECK(x->n == op->dim, MGCR_ERR_INVALID, "Sparse matrix dimension does not match Field dimension!");
    int64_t nrow = op->nrow ? op->nrow : op->dim;
    MGCR_CHECK(y->n == nrow, MGCR_ERR_INVALID, "output Field has %lld entries, operator has %lld rows", (long long)y->n, (long long)nrow);
    // no operator applies in place: the matrix kernels gather from x while they write y, and an MG / GCR operator reads its
    // input again after it has started to write the output (post-smoother, x0 handling)
    MGCR_CHECK(x->d != y->d, MGCR_ERR_INVALID, "mgcr_op_apply: input and output must be different Fields");
    LOCK();
    return op_apply_raw(op, x->d, y->w(), x->n);
}

int mgcr_gcr_solve(mgcr_op_t A, const mgcr_gcr_param *param, mgcr_vec_t rhs, mgcr_vec_t x, double *hist, int32_t hist_cap,
                   int32_t *n_iter, int32_t *converged) {
    MGCR_TRY(require_ctx());
    MGCR_CHECK(A && param && rhs && x, MGCR_ERR_INVALID, "mgcr_gcr_solve: null argument");
    // assertm(rhs.field_size() == this->dim, ...) src/GCR.h:160-161
    MGCR_CHECK(rhs->n == A->dim, MGCR_ERR_INVALID, "Field dimension does not match with Operator!");
    MGCR_CHECK(x->n == A->dim, MGCR_ERR_INVALID, "x dimension does not match with Operator!");
    MGCR_CHECK(rhs->d != x->d, MGCR_ERR_INVALID, "rhs and x must be different Fields");
    LOCK();
    GcrState *s = nullptr;
    MGCR_TRY(gcr_state_create(A, param, 1, &s));
    int it = 0, conv = 0;
    const bool xz = x->zero_known;
    int rc = gcr_run(s, rhs->d, x->w(), false, hist, hist_cap, &it, &conv, xz);
    gcr_state_destroy(s);
    if (n_iter) *n_iter = it;
    if (converged) *converged = conv;
    return rc;
}

int mgcr_gcr_create(mgcr_op_t A, const mgcr_gcr_param *param, int32_t x0_mode, mgcr_op_t *out) {
    MGCR_TRY(require_ctx());
    MGCR_CHECK(param && out, MGCR_ERR_INVALID, "mgcr_gcr_create: null argument");
    LOCK();
    mgcr_op_s *op = new mgcr_op_s();
    op->kind = OP_GCR;
    op->dim = A ? A->dim : 0;
    op->nrow = op->dim;
    int rc = gcr_state_create(A, param, x0_mode, &op->gcr);
    if (rc != MGCR_OK) { delete op; return rc; }
    *out = op;
    return MGCR_OK;
}

int mgcr_gcr_set_operator(mgcr_op_t gcr, mgcr_op_t A) {
    MGCR_CHECK(gcr && A && gcr->kind == OP_GCR, MGCR_ERR_INVALID, "mgcr_gcr_set_operator: not a GCR operator");
    LOCK();
    gcr->dim = A->dim;  // GCR::initialise src/GCR.h:31
    gcr->nrow = A->dim;
    return gcr_state_set_operator(gcr->gcr, A);
}

int mgcr_gcr_set_param(mgcr_op_t gcr, const mgcr_gcr_param *param) {
    MGCR_CHECK(gcr && param && gcr->kind == OP_GCR, MGCR_ERR_INVALID, "mgcr_gcr_set_param: not a GCR operator");
    LOCK();
    return gcr_state_set_param(gcr->gcr, param);
}

int mgcr_gcr_solve_op(mgcr_op_t gcr, mgcr_vec_t rhs, mgcr_vec_t x, double *hist, int32_t hist_cap, int32_t *n_iter,
                      int32_t *converged) {
    MGCR_TRY(require_ctx());
    MGCR_CHECK(gcr && rhs && x && gcr->kind == OP_GCR, MGCR_ERR_INVALID, "mgcr_gcr_solve_op: bad argument");
    MGCR_CHECK(rhs->n == gcr->dim, MGCR_ERR_INVALID, "Field dimension does not match with Operator!");
    MGCR_CHECK(x->n == gcr->dim, MGCR_ERR_INVALID, "x dimension does not match with Operator!");
    MGCR_CHECK(rhs->d != x->d, MGCR_ERR_INVALID, "rhs and x must be different Fields");
    LOCK();
    int it = 0, conv = 0;
    const bool xz = x->zero_known;
    int rc = gcr_run(gcr->gcr, rhs->d, x->w(), false, hist, hist_cap, &it, &conv, xz);
    if (n_iter) *n_iter = it;
    if (converged) *converged = conv;
    return rc;
}

int mgcr_gcr_set_x0(mgcr_op_t gcr, mgcr_vec_t x0) {
    MGCR_TRY(require_ctx());
    MGCR_CHECK(gcr && gcr->kind == OP_GCR, MGCR_ERR_INVALID, "mgcr_gcr_set_x0: not a GCR operator");
    LOCK();
    return gcr_state_set_x0(gcr->gcr, x0 ? x0->d : nullptr, x0 ? x0->n : 0);
}

int mgcr_set_small_solve_rows(int64_t rows) {
    MGCR_CHECK(rows >= 0, MGCR_ERR_INVALID, "rows must be >= 0");
    gcr_small_set_limit(rows);
    return MGCR_OK;
}

int mgcr_gcr_last_profile(double *phase_ms_total, int32_t *n_iter, int32_t *fused) {
    MGCR_CHECK(phase_ms_total && n_iter && fused, MGCR_ERR_INVALID, "null argument");
    int n = 0, f = 0;
    gcr_last_profile(phase_ms_total, &n, &f);
    *n_iter = n;
    *fused = f;
    return MGCR_OK;
}

int mgcr_bench_op_apply(mgcr_op_t op, mgcr_vec_t x, mgcr_vec_t y, int32_t reps, double *ms_avg) {
    MGCR_TRY(require_ctx());
    MGCR_CHECK(op && x && y && reps > 0 && ms_avg, MGCR_ERR_INVALID, "mgcr_bench_op_apply: bad argument");
    LOCK();
    Context &c = ctx();
    MGCR_TRY(mgcr_op_apply(op, x, y));  // warm-up (and argument checks)
    MGCR_HIP(hipEventRecord(c.ev0, c.stream));
    for (int i = 0; i < reps; i++) MGCR_TRY(op_apply_raw(op, x->d, y->w(), x->n));
    MGCR_HIP(hipEventRecord(c.ev1, c.stream));
    MGCR_HIP(hipEventSynchronize(c.ev1));
    float ms = 0.f;
    MGCR_HIP(hipEventElapsedTime(&ms, c.ev0, c.ev1));
    *ms_avg = (double)ms / reps;
    return MGCR_OK;
}

}  // extern "C"
