/* TEST INFRASTRUCTURE — CPU oracle (multigrid part), NOT part of the product path.
 *
 * Restates the well-defined pieces of the reference's two-level adaptive-aggregation multigrid
 * (src/MG.h, src/Mesh.h) in plain C, in the reference's operation order:
 *
 *   aggregates        Mesh::blocking + alloc_full_index           src/Mesh.h:236-324
 *   prolongator       restrict_block + per-block Gram-Schmidt     src/MG.h:171-198,385-403
 *   restrict/expand   MG::restrict / MG::expand                   src/MG.h:347-383
 *   coarse operator   Galerkin blocks P_b'^H (A P_b)              src/MG.h:204-281
 *
 * These are PINNED against the real reference by tests/test_oracle_golden.py (golden G9).
 *
 * The multigrid *application* has no reference output: MG::operator() passes its fields by
 * value and returns uninitialised memory (src/MG.h:29-30,124-129,405-430).
 * orc_mg_apply therefore implements the corrected cycle documented in DESIGN.md (report
 * Algorithm 2: pre-smooth, residual, restrict, coarse solve / recursion, prolong + add,
 * post-smooth) — "parity unpinned" for the cycle as a whole; it is what the HIP cycle is
 * checked against.  Differences to the reference's set-up, all documented in DESIGN.md:
 *   - any coupled pair of aggregates gets a Galerkin block (the reference only visits the 2*4
 *     face neighbours, writes explicit zero blocks and uses the wrong prolongator for the
 *     negative direction when a direction has >= 3 blocks, src/MG.h:260-267 / SURVEY Q8);
 *   - 1..4 blocked dimensions (the reference hard-wires 4) and any number of levels (the
 *     reference stores n_level and never reads it).
 */
#include <complex.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef double _Complex cplx;

struct orc_op;
typedef struct orc_op orc_op;
void orc_op_apply(orc_op *op, const cplx *x, cplx *y);
void orc_op_residual(orc_op *op, const cplx *x, const cplx *b, cplx *r);
int orc_device_recurrence_residual(void);
cplx *orc_take_last_residual(void);
orc_op *orc_op_csr(int64_t nrow, int64_t ncol, const int64_t *rowptr, const int64_t *col, const cplx *val);
orc_op *orc_op_bcsr_from_triplets(int32_t nbrow, int32_t nbcol, int32_t bs, int32_t nt, const int32_t *rows, const int32_t *cols, const cplx *blocks);
void orc_op_free(orc_op *op);
int64_t orc_op_dim(const orc_op *op);

typedef struct orc_gcr_param {
    int truncation, restart, max_iter;
    double tol;
    int verbose;
    orc_op *left_precond, *right_precond;
    int use_x0, flexible;
} orc_gcr_param;
int orc_gcr_solve(orc_op *A, const orc_gcr_param *gp, const cplx *rhs, cplx *x, double *hist, int hist_cap, int *converged);

/* ------------------------------------------------------------------ aggregates ----------- */

/* Aggregate (block) index of every unknown of a row-major mesh dims[ndim] whose dimensions with
 * blocked[d] != 0 are cut into blocks of edge `sub`; all other dimensions (spinor, colour, or the
 * vectors-per-aggregate index of a coarse level) stay inside the aggregate.  Block index is
 * row-major over the block counts of the blocked dimensions in mesh order (src/Mesh.h:281-291).
 * Returns the number of aggregates, or -1 if a blocked dimension is not divisible. */
int64_t orc_mg_aggregates(int ndim, const int64_t *dims, const int32_t *blocked, int64_t sub, int32_t *agg) {
    int64_t n = 1, nagg = 1;
    for (int d = 0; d < ndim; d++) {
        n *= dims[d];
        if (blocked[d]) {
            if (dims[d] % sub) return -1; /* "Dimension not exactly divisible by block size!" src/Mesh.h:245 */
            nagg *= dims[d] / sub;
        }
    }
    for (int64_t i = 0; i < n; i++) {
        int64_t rem = i, b = 0;
        /* decompose from the fastest dimension; accumulate the block index from the slowest */
        int64_t idx[16];
        for (int d = ndim - 1; d >= 0; d--) { idx[d] = rem % dims[d]; rem /= dims[d]; }
        for (int d = 0; d < ndim; d++)
            if (blocked[d]) b = b * (dims[d] / sub) + idx[d] / sub;
        agg[i] = (int32_t)b;
    }
    return nagg;
}

/* ------------------------------------------------------------------ prolongator ---------- */

/* pv[i*ne + k] <- block-local, per-aggregate orthonormalised copy of the ne vectors
 * vecs[k*n + i]  (src/MG.h:171-198): within every aggregate, modified Gram-Schmidt in vector
 * order with dot = sum conj(P_j) P_vec over the aggregate's members in ascending global index
 * (the reference sums over the full zero-padded field in index order, which visits the members
 * in that order), P_vec -= P_j * h, then field *= 1./norm. */
/* The vectors handed to orc_mg_create ARE level 0's prolongator columns (already restricted to the aggregates and orthonormal — e.g. the
 * prolongator a device set-up produced from its own near-null vectors): take them as they are instead of orthonormalising them once more
 * (which moves their last bits).  Level 0 only; read once by the next orc_mg_create. */
static int g_vectors_are_prolongator = 0;
void orc_mg_set_vectors_are_prolongator(int on) { g_vectors_are_prolongator = on; }
void orc_mg_prolongator(int64_t n, int ne, int64_t nagg, const int32_t *agg, const cplx *vecs, cplx *pv) {
    if (g_vectors_are_prolongator) {
        g_vectors_are_prolongator = 0;
        for (int64_t i = 0; i < n; i++)
            for (int k = 0; k < ne; k++) pv[i * ne + k] = vecs[(int64_t)k * n + i];
        (void)nagg; (void)agg;
        return;
    }
    /* member lists in ascending index order */
    int64_t *ptr = (int64_t *)calloc((size_t)nagg + 1, sizeof(int64_t));
    int64_t *mem = (int64_t *)malloc(sizeof(int64_t) * (size_t)n);
    for (int64_t i = 0; i < n; i++) ptr[agg[i] + 1]++;
    for (int64_t a = 0; a < nagg; a++) ptr[a + 1] += ptr[a];
    int64_t *fill = (int64_t *)malloc(sizeof(int64_t) * (size_t)nagg);
    memcpy(fill, ptr, sizeof(int64_t) * (size_t)nagg);
    for (int64_t i = 0; i < n; i++) mem[fill[agg[i]]++] = i;
    for (int64_t i = 0; i < n; i++)
        for (int k = 0; k < ne; k++) pv[i * ne + k] = vecs[(int64_t)k * n + i];
    for (int64_t a = 0; a < nagg; a++) {
        for (int vec = 0; vec < ne; vec++) {
            for (int j = 0; j < vec; j++) {
                cplx h = 0.0;
                for (int64_t m = ptr[a]; m < ptr[a + 1]; m++) h += conj(pv[mem[m] * ne + j]) * pv[mem[m] * ne + vec];
                for (int64_t m = ptr[a]; m < ptr[a + 1]; m++) pv[mem[m] * ne + vec] -= h * pv[mem[m] * ne + j];
            }
            cplx s = 0.0;
            for (int64_t m = ptr[a]; m < ptr[a + 1]; m++) s += conj(pv[mem[m] * ne + vec]) * pv[mem[m] * ne + vec];
            double nrm = sqrt(creal(s));
            for (int64_t m = ptr[a]; m < ptr[a + 1]; m++) pv[mem[m] * ne + vec] *= 1. / nrm;
        }
    }
    free(ptr); free(mem); free(fill);
}

/* src/MG.h:366-383  (R x)[a*ne + k] = P_{a,k}.dot(x) = sum_{i in a, ascending} conj(pv[i][k]) x[i] */
void orc_mg_restrict(int64_t n, int ne, int64_t nagg, const int32_t *agg, const cplx *pv, const cplx *x, cplx *xc) {
    for (int64_t c = 0; c < nagg * ne; c++) xc[c] = 0.0;
    for (int64_t i = 0; i < n; i++)
        for (int k = 0; k < ne; k++) xc[(int64_t)agg[i] * ne + k] += conj(pv[i * ne + k]) * x[i];
}

/* src/MG.h:347-364  x_fine = sum_{a,k} P_{a,k} * x_c[a*ne+k], k ascending (operator*: a * field) */
void orc_mg_expand(int64_t n, int ne, const int32_t *agg, const cplx *pv, const cplx *xc, cplx *x) {
    for (int64_t i = 0; i < n; i++) {
        cplx s = 0.0;
        for (int k = 0; k < ne; k++) s += xc[(int64_t)agg[i] * ne + k] * pv[i * ne + k];
        x[i] = s;
    }
}

/* ------------------------------------------------------------------ Galerkin ------------- */

/* Coarse operator blocks (src/MG.h:216-274): block(a', a)[k'][k] = P_{a',k'}.dot( A P_{a,k} ).
 * A P_{a,k} is evaluated like the reference's operator apply restricted to the columns of
 * aggregate a: row sum in CSR order (src/Operator.h:338-341), then the optional Dirac shift
 * y = x - k_shift * sum (src/Operator.h:573); the outer dot runs over the rows of a' ascending.
 * Two-pass protocol: with rows == NULL only the number of blocks is returned.  Blocks come out
 * sorted by (a', a). */
int64_t orc_mg_galerkin(int64_t n, const int64_t *rowptr, const int64_t *col, const cplx *val, int has_shift,
                        double k_re, double k_im, int ne, int64_t nagg, const int32_t *agg, const cplx *pv,
                        int32_t *rows, int32_t *cols, cplx *blocks) {
    cplx kshift = k_re + k_im * I;
    int64_t *ptr = (int64_t *)calloc((size_t)nagg + 1, sizeof(int64_t));
    int64_t *mem = (int64_t *)malloc(sizeof(int64_t) * (size_t)n);
    for (int64_t i = 0; i < n; i++) ptr[agg[i] + 1]++;
    for (int64_t a = 0; a < nagg; a++) ptr[a + 1] += ptr[a];
    int64_t *fill = (int64_t *)malloc(sizeof(int64_t) * (size_t)nagg);
    memcpy(fill, ptr, sizeof(int64_t) * (size_t)nagg);
    for (int64_t i = 0; i < n; i++) mem[fill[agg[i]]++] = i;
    int32_t *mark = (int32_t *)malloc(sizeof(int32_t) * (size_t)nagg); /* last a' that touched column aggregate */
    for (int64_t a = 0; a < nagg; a++) mark[a] = -1;
    int32_t *nbr = (int32_t *)malloc(sizeof(int32_t) * (size_t)nagg);
    cplx *t = (cplx *)malloc(sizeof(cplx) * (size_t)ne);
    int64_t nblk = 0;
    for (int64_t ap = 0; ap < nagg; ap++) {
        /* neighbour aggregates of a' (itself included when has_shift), ascending */
        int32_t nn = 0;
        if (has_shift) { mark[ap] = (int32_t)ap; nbr[nn++] = (int32_t)ap; }
        for (int64_t m = ptr[ap]; m < ptr[ap + 1]; m++) {
            int64_t i = mem[m];
            for (int64_t l = rowptr[i]; l < rowptr[i + 1]; l++) {
                int32_t a = agg[col[l]];
                if (mark[a] != (int32_t)ap) { mark[a] = (int32_t)ap; nbr[nn++] = a; }
            }
        }
        for (int32_t x = 1; x < nn; x++) { /* insertion sort, nn is small */
            int32_t v = nbr[x], y = x - 1;
            while (y >= 0 && nbr[y] > v) { nbr[y + 1] = nbr[y]; y--; }
            nbr[y + 1] = v;
        }
        if (rows) {
            for (int32_t q = 0; q < nn; q++) {
                int32_t a = nbr[q];
                cplx *blk = blocks + (size_t)(nblk + q) * ne * ne;
                for (int e = 0; e < ne * ne; e++) blk[e] = 0.0;
                rows[nblk + q] = (int32_t)ap;
                cols[nblk + q] = a;
                for (int64_t m = ptr[ap]; m < ptr[ap + 1]; m++) {
                    int64_t i = mem[m];
                    for (int k = 0; k < ne; k++) t[k] = 0.0;
                    for (int64_t l = rowptr[i]; l < rowptr[i + 1]; l++)
                        if (agg[col[l]] == a)
                            for (int k = 0; k < ne; k++) t[k] += val[l] * pv[col[l] * ne + k];
                    for (int k = 0; k < ne; k++) {
                        cplx y = t[k];
                        if (has_shift) y = ((agg[i] == a) ? pv[i * ne + k] : 0.0) - kshift * t[k];
                        for (int kp = 0; kp < ne; kp++) blk[kp * ne + k] += conj(pv[i * ne + kp]) * y;
                    }
                }
            }
        }
        nblk += nn;
    }
    free(ptr); free(mem); free(fill); free(mark); free(nbr); free(t);
    return nblk;
}

/* ------------------------------------------------------------------ hierarchy + cycle ----- */

#define ORC_MG_MAXLEV 8

typedef struct orc_mg {
    int nlev;                      /* number of operator levels (fine = 0 ... coarsest = nlev-1) */
    orc_op *A[ORC_MG_MAXLEV];      /* A[0] borrowed; A[l>0] owned (CSR of the expanded Galerkin blocks) */
    int64_t n[ORC_MG_MAXLEV];
    int ne[ORC_MG_MAXLEV];         /* vectors per aggregate going from level l to l+1 */
    int64_t nagg[ORC_MG_MAXLEV];
    int32_t *agg[ORC_MG_MAXLEV];
    cplx *pv[ORC_MG_MAXLEV];
    int64_t *rp[ORC_MG_MAXLEV], *ci[ORC_MG_MAXLEV];
    cplx *va[ORC_MG_MAXLEV];
    orc_gcr_param smoother, coarse;
    double damping;
} orc_mg;

/* Build the hierarchy.  Level 0: mesh dims0[ndim0] with mask blocked0, operator CSR (+ optional
 * Dirac shift), ne0 near-null vectors vecs0[k*n + i].  Every further level re-blocks the lattice
 * of aggregates (dims = block counts..., ne) by the same `sub`, with the restricted vectors R v
 * as near-null vectors.  `nlev` counts operator levels (2 = the reference's two-level method). */
orc_mg *orc_mg_create(orc_op *A0, int64_t n0, const int64_t *rowptr, const int64_t *col, const cplx *val, int has_shift,
                      double k_re, double k_im, int ndim0, const int64_t *dims0, const int32_t *blocked0, int64_t sub,
                      int ne0, const cplx *vecs0, int nlev, const orc_gcr_param *smoother, const orc_gcr_param *coarse,
                      double damping) {
    if (nlev < 2 || nlev > ORC_MG_MAXLEV) return NULL;
    orc_mg *mg = (orc_mg *)calloc(1, sizeof(orc_mg));
    mg->nlev = nlev; mg->smoother = *smoother; mg->coarse = *coarse; mg->damping = damping;
    mg->A[0] = A0; mg->n[0] = n0;
    int ndim = ndim0;
    int64_t dims[16]; int32_t blocked[16];
    for (int d = 0; d < ndim; d++) { dims[d] = dims0[d]; blocked[d] = blocked0[d]; }
    const int64_t *rp = rowptr, *ci = col; const cplx *va = val;
    int shift = has_shift;
    int ne = ne0;
    cplx *vecs = (cplx *)malloc(sizeof(cplx) * (size_t)ne * n0);
    memcpy(vecs, vecs0, sizeof(cplx) * (size_t)ne * n0);
    for (int l = 0; l + 1 < nlev; l++) {
        int64_t n = mg->n[l];
        mg->ne[l] = ne;
        mg->agg[l] = (int32_t *)malloc(sizeof(int32_t) * (size_t)n);
        int64_t nagg = orc_mg_aggregates(ndim, dims, blocked, sub, mg->agg[l]);
        if (nagg < 0) { fprintf(stderr, "orc_mg_create: dimension not divisible by block size at level %d\n", l); abort(); }
        mg->nagg[l] = nagg;
        mg->pv[l] = (cplx *)malloc(sizeof(cplx) * (size_t)n * ne);
        orc_mg_prolongator(n, ne, nagg, mg->agg[l], vecs, mg->pv[l]);
        /* Galerkin blocks -> scalar CSR of the coarse operator */
        int64_t nblk = orc_mg_galerkin(n, rp, ci, va, shift, k_re, k_im, ne, nagg, mg->agg[l], mg->pv[l], NULL, NULL, NULL);
        int32_t *br = (int32_t *)malloc(sizeof(int32_t) * (size_t)nblk), *bc = (int32_t *)malloc(sizeof(int32_t) * (size_t)nblk);
        cplx *bl = (cplx *)malloc(sizeof(cplx) * (size_t)nblk * ne * ne);
        orc_mg_galerkin(n, rp, ci, va, shift, k_re, k_im, ne, nagg, mg->agg[l], mg->pv[l], br, bc, bl);
        int64_t nc = nagg * ne;
        int64_t *crp = (int64_t *)calloc((size_t)nc + 1, sizeof(int64_t));
        int64_t *cci = (int64_t *)malloc(sizeof(int64_t) * (size_t)nblk * ne * ne);
        cplx *cva = (cplx *)malloc(sizeof(cplx) * (size_t)nblk * ne * ne);
        /* blocks are sorted by (row, col): row r = a'*ne + kp holds, block after block, ne entries */
        int64_t *brp = (int64_t *)calloc((size_t)nagg + 1, sizeof(int64_t));
        for (int64_t b = 0; b < nblk; b++) brp[br[b] + 1]++;
        for (int64_t a = 0; a < nagg; a++) brp[a + 1] += brp[a];
        int64_t p = 0;
        for (int64_t a = 0; a < nagg; a++)
            for (int kp = 0; kp < ne; kp++) {
                crp[a * ne + kp] = p;
                for (int64_t b = brp[a]; b < brp[a + 1]; b++)
                    for (int k = 0; k < ne; k++) { cci[p] = (int64_t)bc[b] * ne + k; cva[p] = bl[(size_t)b * ne * ne + kp * ne + k]; p++; }
            }
        crp[nc] = p;
        mg->rp[l + 1] = crp; mg->ci[l + 1] = cci; mg->va[l + 1] = cva;
        /* The operator itself: with several unknowns per aggregate it is the reference's HierarchicalSparse (src/MG.h:204-281) — every
         * block applied as a Dense and the block results accumulated (src/HierarchicalSparse.h:101-161), NOT a row-long sequential sum over
         * the expanded entries; the expanded CSR above only feeds the next level's Galerkin products.  One unknown per aggregate: the two
         * are the same sums, and the CSR form carries the device's row layouts (orc_op_set_layout). */
        if (ne > 1) mg->A[l + 1] = orc_op_bcsr_from_triplets((int32_t)nagg, (int32_t)nagg, ne, (int32_t)nblk, br, bc, bl);
        else mg->A[l + 1] = orc_op_csr(nc, nc, crp, cci, cva);
        free(br); free(bc); free(bl); free(brp);
        mg->n[l + 1] = nc;
        /* next level: lattice of aggregates x ne, near-null vectors = R v */
        if (l + 2 < nlev) {
            cplx *nv = (cplx *)malloc(sizeof(cplx) * (size_t)ne * nc);
            for (int k = 0; k < ne; k++) orc_mg_restrict(n, ne, nagg, mg->agg[l], mg->pv[l], vecs + (int64_t)k * n, nv + (int64_t)k * nc);
            free(vecs); vecs = nv;
            int nd2 = 0; int64_t d2[16]; int32_t b2[16];
            for (int d = 0; d < ndim; d++) if (blocked[d]) { d2[nd2] = dims[d] / sub; b2[nd2] = 1; nd2++; }
            d2[nd2] = ne; b2[nd2] = 0; nd2++;
            ndim = nd2;
            for (int d = 0; d < ndim; d++) { dims[d] = d2[d]; blocked[d] = b2[d]; }
            rp = crp; ci = cci; va = cva; shift = 0;
        }
    }
    free(vecs);
    return mg;
}

int64_t orc_mg_level_dim(const orc_mg *mg, int l) { return mg->n[l]; }
int orc_mg_level_ne(const orc_mg *mg, int l) { return mg->ne[l]; }
int64_t orc_mg_level_nagg(const orc_mg *mg, int l) { return mg->nagg[l]; }
const cplx *orc_mg_level_pv(const orc_mg *mg, int l) { return mg->pv[l]; }
const int32_t *orc_mg_level_agg(const orc_mg *mg, int l) { return mg->agg[l]; }
orc_op *orc_mg_level_op(const orc_mg *mg, int l) { return mg->A[l]; }
void orc_mg_level_restrict(const orc_mg *mg, int l, const cplx *x, cplx *xc) {
    orc_mg_restrict(mg->n[l], mg->ne[l], mg->nagg[l], mg->agg[l], mg->pv[l], x, xc);
}
void orc_mg_level_expand(const orc_mg *mg, int l, const cplx *xc, cplx *x) {
    orc_mg_expand(mg->n[l], mg->ne[l], mg->agg[l], mg->pv[l], xc, x);
}

/* Corrected cycle (report Algorithm 2; structure of src/MG.h:405-430 with its defects fixed):
 *   x = S(b)                      pre-smoothing: nu GCR iterations from x0 = 0
 *   r = b - A x ; b_c = R r       (the reference restricts rhs, :418)
 *   x_c = cycle(l+1, b_c)  or coarsest GCR solve from x0 = 0
 *   x += damping * P x_c          (the reference hard-codes 0.1, :426)
 *   x = S(b, x0 = x)              post-smoothing honours x0 (the reference overwrites x, :429) */
static void cycle(orc_mg *mg, int l, const cplx *b, cplx *x) {
    int64_t n = mg->n[l];
    if (l == mg->nlev - 1) {
        memset(x, 0, sizeof(cplx) * (size_t)n);
        orc_gcr_param p = mg->coarse;
        p.use_x0 = 0; p.verbose = 0;
        orc_gcr_solve(mg->A[l], &p, b, x, NULL, 0, NULL);
        return;
    }
    orc_gcr_param p = mg->smoother;
    p.verbose = 0;
    memset(x, 0, sizeof(cplx) * (size_t)n);
    p.use_x0 = 0;
    orc_gcr_solve(mg->A[l], &p, b, x, NULL, 0, NULL);
    cplx *r;
    if (orc_device_recurrence_residual()) {
        r = orc_take_last_residual();      /* order 3, device default: the residual the pre-smoother's recurrence ended with (csrc/mg.hip) */
    } else {
        r = (cplx *)malloc(sizeof(cplx) * (size_t)n);
        orc_op_residual(mg->A[l], x, b, r);   /* r = b - A x (order 0: apply, then subtract) */
    }
    int64_t nc = mg->n[l + 1];
    cplx *bc = (cplx *)malloc(sizeof(cplx) * (size_t)nc), *xc = (cplx *)malloc(sizeof(cplx) * (size_t)nc);
    orc_mg_level_restrict(mg, l, r, bc);
    cycle(mg, l + 1, bc, xc);
    orc_mg_level_expand(mg, l, xc, r);
    cplx damp = mg->damping;
    for (int64_t i = 0; i < n; i++) x[i] = x[i] + damp * r[i]; /* x += x_corr * damping (src/MG.h:426) */
    p.use_x0 = 1;
    orc_gcr_solve(mg->A[l], &p, b, x, NULL, 0, NULL);
    free(r); free(bc); free(xc);
}

void orc_mg_apply(orc_mg *mg, const cplx *f, cplx *y) { cycle(mg, 0, f, y); }

void orc_mg_free(orc_mg *mg) {
    if (!mg) return;
    for (int l = 0; l < mg->nlev; l++) {
        free(mg->agg[l]); free(mg->pv[l]);
        if (l > 0) { orc_op_free(mg->A[l]); free(mg->rp[l]); free(mg->ci[l]); free(mg->va[l]); }
    }
    free(mg);
}
