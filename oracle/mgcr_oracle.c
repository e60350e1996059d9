/* TEST INFRASTRUCTURE — CPU oracle, NOT part of the product path.
 *
 * Clean-room plain-C restatement of the reference's hot path (jing2li/MGPreconditionedGCR @
 * 2024_10_08), function by function, in the reference's own operation order so that results
 * agree with the real reference to rounding (most of them bit for bit).  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library — as the
 * checker, never as the thing shipped or measured.  The HIP library never links or calls it.
 *
 * Parity status: PINNED.  tests/test_oracle_golden.py checks every function below against the
 * golden vectors in tests/golden/ that were produced by the real reference compiled here
 * (oracle/ref_harness.cpp, tests/golden/make_golden.py).  The one part with no reference
 * output is the multigrid *application* (src/MG.h:124-129,405-430 returns uninitialised
 * memory): orc_mg_* restates its well-defined pieces (pinned) and a
 * corrected cycle (documented in DESIGN.md; "parity unpinned" for the cycle as a whole).
 *
 * Every function cites the reference lines it follows (paths relative to the reference root).
 * All complex data is C99 `double _Complex`, layout-compatible with std::complex<double> and
 * with interleaved (re,im) doubles.  gcc's complex multiply (__muldc3) is the same routine
 * libstdc++'s std::complex<double>::operator* lowers to, so element-wise results match g++.
 */
#include <complex.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef double _Complex cplx;

/* ------------------------------------------------------------------ Field algebra -------- */

/* Summation order of the two reductions below.  0 (default) = the reference's: sequential in
 * index order.  1 = reverse index order, 2 = pairwise tree: equally valid orders, used by the
 * tests ONLY to measure how sensitive a residual history is to re-association, which is the one
 * thing a parallel reduction cannot reproduce.  3 = the order the HIP library sums in (the model
 * below): with it a solve of the oracle and a solve on the GPU must agree BIT FOR BIT, which is
 * how tests/test_gpu_bitwise.py proves that summation order is the only difference between the
 * device path and the reference. */
static int g_sum_order = 0;
void orc_set_sum_order(int mode) { g_sum_order = mode; }

static cplx dot_pairwise(int64_t n, const cplx *a, const cplx *b) {
    if (n <= 8) {
        cplx s = 0.0;
        for (int64_t i = 0; i < n; i++) s += conj(a[i]) * b[i];
        return s;
    }
    int64_t h = n / 2;
    return dot_pairwise(h, a, b) + dot_pairwise(n - h, a + h, b + h);
}

/* ---- order 3: model of the device's two-stage reduction (mgpreconditionedgcr_amd/csrc/reduce.h,
 * blas1.hip:red_grid, gcr_dev.h:RowMap).  A reducing kernel runs `g` workgroups of 1024 threads;
 * thread t of workgroup b adds the terms of its rows (first, first + step, ...: ascending) into a
 * private accumulator that starts at 0; the 64 lanes of a wave are summed by the tree
 * (l, l+32), (l, l+16), ..., (l, l+1), always lower + upper (reduce.h:wave_sum); the 16 wave sums
 * are added in wave order onto 0 (block_sum_owner); the g workgroup partials are folded by the same
 * 1024-wide tree with zeros beyond g (fold_partials).  Real and imaginary parts are separate sums.
 * Row map: plain grid-stride (first = b*1024 + t, step = g*1024), or — only for the kernels that
 * embed the operator apply, when the operator's rows reach far (RowMap::band != 0) — eight
 * contiguous bands swept by g/8 workgroups each. */
#define DEV_THREADS 1024
#define DEV_MAX_BLOCKS 512
static int g_dev_blocks = 0;      /* 0: red_grid(n) = min(ceil(n/1024), 512); 1: the one-workgroup solver (gcr_small.hip) */
static int64_t g_dev_band = 0;    /* RowMap::band (0 = not banded) */
static int g_dev_per = 0;         /* RowMap::per */
static int g_dev_init_banded = 0; /* step 0's sums come out of the kernel that embeds the apply (gcr_fused.hip init_apply_kernel) */
static int g_dev_ell_w = -1;      /* SpMV layout: ELL width W (rows longer than W keep a CSR tail); -1: every row sequential */
static int g_dev_ell_l = 1;       /* lanes per row of the ELL part (1: sequential in CSR order) */
static int g_dev_tail_cap = 0;    /* tail entries of a row: up to this many are summed in CSR order by one thread, more by a wave (lanes + tree) */
static int64_t g_dev_plane = 0;   /* RowMap::plane: != 0, a band's workgroups tile ONE plane of this many rows (the last tile short) and step by it */
void orc_set_device_plane(int64_t plane) { g_dev_plane = plane; }
void orc_set_device_model(int blocks, int64_t band, int per, int init_banded, int ell_width, int ell_lanes, int tail_cap) {
    g_dev_blocks = blocks; g_dev_band = band; g_dev_per = per; g_dev_init_banded = init_banded;
    g_dev_ell_w = ell_width; g_dev_ell_l = ell_lanes < 1 ? 1 : ell_lanes; g_dev_tail_cap = tail_cap;
}

/* Several ranks (one process per GPU, rows dealt in contiguous blocks: csrc/comm.hip): every rank sums the terms of ITS rows as
 * above (grid of red_grid(rows of the rank) workgroups over local row numbers), the rank totals are then added in rank order
 * onto 0 (fold_pw_kernel: identical bits on every rank).  A row's entries keep their CSR order on every rank, halo columns
 * included, whatever the block's storage (measured: tests/test_gpu_dist.py compares the distributed apply bit for bit). */
#define DEV_MAX_RANKS 16
static int g_dev_nranks = 1;
static int64_t g_dev_rank_off[DEV_MAX_RANKS + 1];
void orc_set_device_ranks(int nranks, const int64_t *row_offsets) {
    g_dev_nranks = nranks < 1 ? 1 : nranks > DEV_MAX_RANKS ? DEV_MAX_RANKS : nranks;
    for (int r = 0; r <= g_dev_nranks && nranks >= 1; r++) g_dev_rank_off[r] = row_offsets ? row_offsets[r] : 0;
}

/* Lean restart cycles (csrc/gcr.hip header): in restart mode without the literal preconditioner hooks the device never forms the
 * directions p_k inside a cycle.  It keeps the cycle's first direction P0 and the vectors D_1..D_k the later directions were
 * started from (the residuals, or M r in flexible mode), the triangular table p_k = t_k P0 + sum_m T_km D_m, and the coefficients
 * cx of the pending update x += sum_k alpha_k p_k, which is applied when a cycle closes (together with the next cycle's
 * P0' = dir - sum_j beta_j p_j) or when the solve ends.  r, Ap and every scalar follow the classic recurrences — only x (and P0) are
 * associated differently.  orc_set_device_lean(1, .) makes orc_gcr_solve form x that way (order 3 only), coefficient by coefficient
 * in the kernels' order, so that x is comparable bit for bit as well.  recurrence_residual: the V-cycle restricts the residual
 * the pre-smoother's recurrence ended with instead of b - A x (csrc/mg.hip). */
static int g_dev_lean = 0, g_dev_recurrence_residual = 0;
void orc_set_device_lean(int lean, int recurrence_residual) { g_dev_lean = lean; g_dev_recurrence_residual = recurrence_residual; }
/* Where the device forms r' = r - alpha Ap INSIDE the kernel that embeds the apply (csrc/gcr_fused_xr_tile.h: lean cycles on a 3-D
 * stencil whose far neighbours are one step of the banded row map away, e.g. a 256 x 256 x Z grid), |r'|^2 is summed over THAT kernel's
 * row map — every step but the solve's last (max_iter reached: nothing is applied after it, the plain update kernel runs). */
static int g_dev_xr_banded = 0;
void orc_set_device_xr_banded(int on) { g_dev_xr_banded = on; }
int orc_device_recurrence_residual(void) { return g_sum_order == 3 && g_dev_recurrence_residual; }
static cplx *g_last_r = NULL;        /* the residual the last orc_gcr_solve ended with (malloc'ed; taken over by the caller) */
cplx *orc_take_last_residual(void) { cplx *r = g_last_r; g_last_r = NULL; return r; }
/* the device's complex multiply (reduce.h:cmul), spelled out: the coefficient arithmetic below must not depend on how the C
 * compiler lowers `*` */
static inline cplx cm(cplx a, cplx b) {
    const double ax = creal(a), ay = cimag(a), bx = creal(b), by = cimag(b);
    return (ax * bx - ay * by) + (ax * by + ay * bx) * I;
}
#define LEAN_MAX 16

static int dev_grid(int64_t n) {
    if (g_dev_blocks > 0) return g_dev_blocks;
    int64_t g = (n + DEV_THREADS - 1) / DEV_THREADS;
    if (g < 1) g = 1;
    if (g > DEV_MAX_BLOCKS) g = DEV_MAX_BLOCKS;
    return (int)g;
}
static double dev_wave_tree(double *v) { /* v[64], destroyed */
    for (int off = 32; off >= 1; off >>= 1)
        for (int l = 0; l < off; l++) v[l] = v[l] + v[l + off];
    return v[0];
}
static double dev_block_sum(double *acc) { /* acc[1024], destroyed */
    double t = 0.;
    for (int w = 0; w < DEV_THREADS / 64; w++) t += dev_wave_tree(acc + 64 * w);
    return t;
}
/* sum of term[0..n) in the device's order; banded != 0 selects the RowMap of the apply-embedding kernels */
static double dev_sum(int64_t n, const double *term, int banded) {
    const int g = dev_grid(n);
    double acc[DEV_THREADS], parts[DEV_THREADS];
    for (int b = 0; b < DEV_THREADS; b++) parts[b] = 0.;
    const int use_band = banded && g_dev_band > 0 && g_dev_per > 0 && g % 8 == 0;
    for (int b = 0; b < g; b++) {
        for (int t = 0; t < DEV_THREADS; t++) {
            int64_t first, end, step;
            if (use_band) {
                const int64_t xb = b / g_dev_per, in_plane = (int64_t)(b % g_dev_per) * DEV_THREADS + t;
                first = xb * g_dev_band + in_plane;
                end = (xb + 1) * g_dev_band; if (end > n) end = n;
                step = g_dev_plane ? g_dev_plane : (int64_t)g_dev_per * DEV_THREADS;
                if (g_dev_plane && in_plane >= g_dev_plane) first = end;   /* a short last tile's threads beyond the plane's end */
            } else {
                first = (int64_t)b * DEV_THREADS + t; end = n; step = (int64_t)g * DEV_THREADS;
            }
            double a = 0.;
            for (int64_t i = first; i < end; i += step) a += term[i];
            acc[t] = a;
        }
        parts[b] = dev_block_sum(acc);
    }
    if (g == 1) return parts[0] + 0.; /* fold_partials: one partial is loaded as it is (x + 0.0) */
    return dev_block_sum(parts);
}
static double dev_sum_ranks(int64_t n, const double *term, int banded) {
    if (g_dev_nranks <= 1) return dev_sum(n, term, banded);
    double tot = 0.;
    for (int r = 0; r < g_dev_nranks; r++) {
        const int64_t a = g_dev_rank_off[r], b = g_dev_rank_off[r + 1];
        tot += dev_sum(b - a, term + a, banded); /* (every rank deals ITS rows by the map of its block: band / per as set for a rank's row count) */
    }
    (void)n;
    return tot;
}
static int g_dev_banded_now = 0; /* set by the GCR loop around the sums the apply-embedding kernels take */
static cplx dot_device(int64_t n, const cplx *a, const cplx *b) {
    double *re = (double *)malloc(sizeof(double) * (size_t)(n ? n : 1)), *im = (double *)malloc(sizeof(double) * (size_t)(n ? n : 1));
    for (int64_t i = 0; i < n; i++) {
        const double ax = creal(a[i]), ay = cimag(a[i]), bx = creal(b[i]), by = cimag(b[i]);
        re[i] = ax * bx + ay * by;   /* conj(a) * b, un-fused (reduce.h:cconj_mul) */
        im[i] = ax * by - ay * bx;
    }
    const double sr = dev_sum_ranks(n, re, g_dev_banded_now), si = dev_sum_ranks(n, im, g_dev_banded_now);
    free(re); free(im);
    return sr + si * I;
}
static double sqnorm_device(int64_t n, const cplx *a) {
    double *t = (double *)malloc(sizeof(double) * (size_t)(n ? n : 1));
    for (int64_t i = 0; i < n; i++) { const double x = creal(a[i]), y = cimag(a[i]); t[i] = x * x + y * y; }
    const double s = dev_sum_ranks(n, t, g_dev_banded_now);
    free(t);
    return s;
}

/* src/Fields.h:216-226  dot(a,b) = sum_i conj(a_i) * b_i, sequential, index order */
void orc_dot(int64_t n, const cplx *a, const cplx *b, cplx *out) {
    cplx s = 0.0;
    if (g_sum_order == 1) for (int64_t i = n - 1; i >= 0; i--) s += conj(a[i]) * b[i];
    else if (g_sum_order == 2) s = dot_pairwise(n, a, b);
    else if (g_sum_order == 3) s = dot_device(n, a, b);
    else for (int64_t i = 0; i < n; i++) s += conj(a[i]) * b[i];
    *out = s;
}

/* src/Fields.h:228-235  squarednorm = Re sum_i conj(a_i) * a_i (complex accumulator) */
double orc_sqnorm(int64_t n, const cplx *a) {
    cplx s = 0.0;
    if (g_sum_order == 1) for (int64_t i = n - 1; i >= 0; i--) s += conj(a[i]) * a[i];
    else if (g_sum_order == 2) s = dot_pairwise(n, a, a);
    else if (g_sum_order == 3) return sqnorm_device(n, a);
    else for (int64_t i = 0; i < n; i++) s += conj(a[i]) * a[i];
    return creal(s);
}

/* src/Fields.h:192-214,245-253  out = a + b*alpha  (operator* computes `alpha * b_i`) */
void orc_add_scaled(int64_t n, const cplx *a, const cplx *b, const cplx *alpha, cplx *out) {
    for (int64_t i = 0; i < n; i++) out[i] = a[i] + (*alpha) * b[i];
}
/* out = a - b*alpha */
void orc_sub_scaled(int64_t n, const cplx *a, const cplx *b, const cplx *alpha, cplx *out) {
    for (int64_t i = 0; i < n; i++) out[i] = a[i] - (*alpha) * b[i];
}
/* src/Fields.h:237-243  normalise: field[i] *= 1./norm */
void orc_normalise(int64_t n, cplx *a) {
    double nrm = sqrt(orc_sqnorm(n, a));
    for (int64_t i = 0; i < n; i++) a[i] *= 1. / nrm;
}

/* ------------------------------------------------------------------ operators ------------ */

typedef enum { OP_CSR = 1, OP_DIRAC = 2, OP_BCSR = 3, OP_GCR = 4, OP_MG = 5 } op_kind;

struct orc_op;
typedef struct orc_gcr_param {
    /* mirror of GCR_Param (src/SolverParam.h:21-35) */
    int truncation, restart, max_iter;
    double tol;
    int verbose;
    struct orc_op *left_precond, *right_precond;
    /* extensions the reference does not have (both default 0 = reference behaviour):
     *  use_x0      r0 = b - A x0 instead of the reference's r0 = b (src/GCR.h:189)
     *  flexible    proper flexible right preconditioning: p = M r, Ap = A p, r stays the true
     *              residual — instead of the reference's literal r = M(r) (src/GCR.h:236-238) */
    int use_x0, flexible;
} orc_gcr_param;

typedef struct orc_mg orc_mg;

typedef struct orc_op {
    op_kind kind;
    int64_t dim;
    /* CSR (src/Operator.h:56-101): int64 indices as in the reference (num_type = long) */
    int64_t nrow;
    const int64_t *rowptr, *col;
    const cplx *val;
    /* Dirac = 1 - k D (src/Operator.h:104-122) */
    struct orc_op *D;
    cplx k;
    /* block-CSR of dense bs x bs blocks (src/HierarchicalSparse.h:22-48) */
    int32_t nbrow, bs;
    int32_t *browptr, *bcol;
    cplx *blocks;
    /* GCR used as an operator (src/GCR.h:62-68) */
    struct orc_op *A;
    orc_gcr_param gp;
    int x0_mode; /* 0: reference (x0 = caller-supplied random field, r0 = b) ; 1: x0 = 0 */
    /* MG */
    orc_mg *mg;
    int owns;
    /* order 3: this operator's own device layout (orc_op_set_layout; the levels of a multigrid hierarchy are stored differently) */
    int has_layout, lay_w, lay_l, lay_cap;
    /* order 3: the row map of the solver kernels that work on THIS operator (orc_op_set_rowmap: the levels of a hierarchy have
     * different sizes and reaches, hence different maps) — overrides orc_set_device_model's band / per / init_banded,
     * orc_set_device_plane and orc_set_device_xr_banded for the duration of a GCR solve on it */
    int has_rowmap, rm_per, rm_init_banded, rm_xr_banded;
    int64_t rm_band, rm_plane;
} orc_op;

void orc_op_apply(orc_op *op, const cplx *x, cplx *y);
void orc_op_residual(orc_op *op, const cplx *x, const cplx *b, cplx *r);
int64_t orc_op_nrow(const orc_op *op);

/* src/Operator.h:330-346  y_row = sum_l VAL[l] * x[COL[l]], sequential per row.
 * Order 3 with a device layout set (orc_set_device_model: W, L): the row's first min(len, W) entries are summed the way
 * csrc/spmv.hip:ell_spmv_lanes does — lane l of L adds entries l, l+L, ... in order, the L lane sums are combined by the tree
 * (l, l+L/2), ..., (l, l+1), lower + upper — and the entries beyond W the way the tail kernels do: in CSR order by one thread
 * (csr_tail_chunk_kernel; rows with at most tail_cap tail entries), or 64 lanes striding them + wave tree (csr_tail_kernel),
 * then y_row = y_ell + y_tail (DiracOp: y = (x - k y_ell) - k y_tail, see dirac_apply).  L = 1 and no tail is
 * the reference's order. */
static void csr_row_device(const orc_op *op, int64_t row, const cplx *x, cplx *ell, cplx *tail, int *has_tail) {
    const int64_t b = op->rowptr[row], e = op->rowptr[row + 1];
    const int W = op->has_layout ? op->lay_w : g_dev_ell_w, L = op->has_layout ? op->lay_l : g_dev_ell_l;
    const int tail_cap = op->has_layout ? op->lay_cap : g_dev_tail_cap;
    const int64_t ne = (e - b) < W ? (e - b) : W;
    double lr[16], li[16];
    for (int l = 0; l < L; l++) { lr[l] = 0.; li[l] = 0.; }
    for (int64_t w = 0; w < ne; w++) {
        const cplx t = op->val[b + w] * x[op->col[b + w]];
        lr[w % L] += creal(t); li[w % L] += cimag(t);
    }
    for (int off = L / 2; off >= 1; off >>= 1)
        for (int l = 0; l < off; l++) { lr[l] = lr[l] + lr[l + off]; li[l] = li[l] + li[l + off]; }
    *ell = lr[0] + li[0] * I;
    *has_tail = (e - b) > W;
    if (*has_tail && (e - b) - W <= tail_cap) { /* csr_tail_chunk_kernel: products in LDS, one thread adds them in CSR order */
        cplx t = 0.0;
        for (int64_t j = b + W; j < e; j++) t = t + op->val[j] * x[op->col[j]];
        *tail = t;
    } else if (*has_tail) {
        double tr[64], ti[64];
        for (int l = 0; l < 64; l++) { tr[l] = 0.; ti[l] = 0.; }
        for (int64_t j = b + W; j < e; j++) {
            const cplx t = op->val[j] * x[op->col[j]];
            tr[(j - b - W) % 64] += creal(t); ti[(j - b - W) % 64] += cimag(t);
        }
        const double sr = dev_wave_tree(tr), si = dev_wave_tree(ti);
        *tail = sr + si * I;
    }
}
static int csr_device_layout_of(const orc_op *op) {
    if (g_sum_order != 3) return 0;
    if (op && op->has_layout) return op->lay_w >= 0 && op->lay_l <= 16;
    return g_dev_ell_w >= 0 && g_dev_ell_l <= 16;
}
static void csr_apply(const orc_op *op, const cplx *x, cplx *y) {
    if (csr_device_layout_of(op)) {
        for (int64_t row = 0; row < op->nrow; row++) {
            cplx ell, tail = 0.0; int ht;
            csr_row_device(op, row, x, &ell, &tail, &ht);
            y[row] = ht ? ell + tail : ell;
        }
        return;
    }
    for (int64_t row = 0; row < op->nrow; row++) {
        cplx sum = 0.0;
        for (int64_t l = op->rowptr[row]; l < op->rowptr[row + 1]; l++) sum += op->val[l] * x[op->col[l]];
        y[row] = sum;
    }
}

/* src/Operator.h:569-575  f - (D f) * k */
static void dirac_apply(const orc_op *op, const cplx *x, cplx *y) {
    if (op->D->kind == OP_CSR && csr_device_layout_of(op->D)) { /* the shift sits in the SpMV epilogue of each of the two kernels */
        for (int64_t row = 0; row < op->D->nrow; row++) {
            cplx ell, tail = 0.0; int ht;
            csr_row_device(op->D, row, x, &ell, &tail, &ht);
            cplx v = x[row] - op->k * ell;
            if (ht) v = v - op->k * tail;
            y[row] = v;
        }
        return;
    }
    cplx *t = (cplx *)malloc(sizeof(cplx) * (size_t)op->dim);
    orc_op_apply(op->D, x, t);
    for (int64_t i = 0; i < op->dim; i++) y[i] = x[i] - op->k * t[i];
    free(t);
}

/* src/HierarchicalSparse.h:101-161 with Dense::operator() (src/Operator.h:159-173) as the block
 * kernel: value = 0; for l in row: value += Dense_l(x_block[col]);  Dense: out[r] = 0, then
 * out[r] = out[r] + mat[r*bs+c] * f[c] for c ascending. */
static void bcsr_apply(const orc_op *op, const cplx *x, cplx *y) {
    int bs = op->bs;
    cplx *tmp = (cplx *)malloc(sizeof(cplx) * (size_t)bs);
    for (int32_t brow = 0; brow < op->nbrow; brow++) {
        cplx *value = y + (size_t)brow * bs;
        for (int i = 0; i < bs; i++) value[i] = 0.0;
        for (int32_t l = op->browptr[brow]; l < op->browptr[brow + 1]; l++) {
            const cplx *m = op->blocks + (size_t)l * bs * bs;
            const cplx *f = x + (size_t)op->bcol[l] * bs;
            for (int r = 0; r < bs; r++) {
                cplx o = 0.0;
                for (int c = 0; c < bs; c++) o = o + m[r * bs + c] * f[c];
                tmp[r] = o;
            }
            for (int r = 0; r < bs; r++) value[r] += tmp[r]; /* Field::operator+= src/Fields.h:288-297 */
        }
    }
    free(tmp);
}

/* r = b - op(x).  Order 3, Sparse with multi-lane rows / a CSR tail: the device forms it in the SpMV's epilogue
 * (gcr.hip:op_residual_raw), i.e. (b - ell part) - tail part, not b - (ell part + tail part) */
void orc_op_residual(orc_op *op, const cplx *x, const cplx *b, cplx *r) {
    if (op->kind == OP_CSR && csr_device_layout_of(op) && g_dev_blocks != 1) {
        for (int64_t row = 0; row < op->nrow; row++) {
            cplx ell, tail = 0.0; int ht;
            csr_row_device(op, row, x, &ell, &tail, &ht);
            cplx v = b[row] - ell;
            if (ht) v = v - tail;
            r[row] = v;
        }
        return;
    }
    int64_t n = orc_op_nrow(op);
    orc_op_apply(op, x, r);
    for (int64_t i = 0; i < n; i++) r[i] = b[i] - r[i];
}
void orc_op_set_rowmap(orc_op *op, int64_t band, int per, int64_t plane, int init_banded, int xr_banded) {
    op->has_rowmap = 1; op->rm_band = band; op->rm_per = per; op->rm_plane = plane; op->rm_init_banded = init_banded; op->rm_xr_banded = xr_banded;
}
void orc_op_set_layout(orc_op *op, int ell_width, int ell_lanes, int tail_cap) {
    op->has_layout = 1; op->lay_w = ell_width; op->lay_l = ell_lanes < 1 ? 1 : ell_lanes; op->lay_cap = tail_cap;
}

orc_op *orc_op_csr(int64_t nrow, int64_t ncol, const int64_t *rowptr, const int64_t *col, const cplx *val) {
    orc_op *op = (orc_op *)calloc(1, sizeof(orc_op));
    op->kind = OP_CSR; op->dim = ncol; op->nrow = nrow;
    op->rowptr = rowptr; op->col = col; op->val = val; /* borrowed: caller keeps them alive */
    return op;
}
orc_op *orc_op_dirac(orc_op *D, double k_re, double k_im) {
    orc_op *op = (orc_op *)calloc(1, sizeof(orc_op));
    op->kind = OP_DIRAC; op->dim = D->dim; op->D = D; op->k = k_re + k_im * I;
    return op;
}

/* src/HierarchicalSparse.h:58-98  unsorted (row,col,block) triplets -> block-CSR; duplicates of a
 * (row,col) pair are KEPT (summed at apply time).  Sorted by key row*nbcol+col.  The reference
 * uses std::sort, whose order among equal keys is unspecified; we use a stable sort (equal keys
 * keep input order) — sums over duplicates may differ from the reference in the last bit. */
typedef struct { int64_t key; int32_t idx; } trip_key;
static int trip_cmp(const void *a, const void *b) {
    const trip_key *x = (const trip_key *)a, *y = (const trip_key *)b;
    if (x->key != y->key) return x->key < y->key ? -1 : 1;
    return x->idx < y->idx ? -1 : (x->idx > y->idx);
}
orc_op *orc_op_bcsr_from_triplets(int32_t nbrow, int32_t nbcol, int32_t bs, int32_t nt, const int32_t *rows,
                                  const int32_t *cols, const cplx *blocks) {
    orc_op *op = (orc_op *)calloc(1, sizeof(orc_op));
    op->kind = OP_BCSR; op->dim = (int64_t)nbcol * bs; op->nbrow = nbrow; op->bs = bs; op->owns = 1;
    op->browptr = (int32_t *)calloc((size_t)nbrow + 1, sizeof(int32_t));
    op->bcol = (int32_t *)malloc(sizeof(int32_t) * (size_t)nt);
    op->blocks = (cplx *)malloc(sizeof(cplx) * (size_t)nt * bs * bs);
    trip_key *keys = (trip_key *)malloc(sizeof(trip_key) * (size_t)nt);
    for (int32_t t = 0; t < nt; t++) { keys[t].key = (int64_t)rows[t] * nbcol + cols[t]; keys[t].idx = t; }
    qsort(keys, (size_t)nt, sizeof(trip_key), trip_cmp);
    for (int32_t t = 0; t < nt; t++) {
        int32_t s = keys[t].idx;
        op->bcol[t] = cols[s];
        memcpy(op->blocks + (size_t)t * bs * bs, blocks + (size_t)s * bs * bs, sizeof(cplx) * (size_t)bs * bs);
        op->browptr[rows[s] + 1]++;
    }
    for (int32_t r = 0; r < nbrow; r++) op->browptr[r + 1] += op->browptr[r];
    free(keys);
    return op;
}
int32_t orc_bcsr_nblocks(const orc_op *op) { return op->browptr[op->nbrow]; }
void orc_bcsr_export(const orc_op *op, int32_t *browptr, int32_t *bcol, cplx *blocks) {
    memcpy(browptr, op->browptr, sizeof(int32_t) * ((size_t)op->nbrow + 1));
    int32_t nb = op->browptr[op->nbrow];
    memcpy(bcol, op->bcol, sizeof(int32_t) * (size_t)nb);
    memcpy(blocks, op->blocks, sizeof(cplx) * (size_t)nb * op->bs * op->bs);
}

/* src/HierarchicalSparse.h:164-178  val_at(row,col): sum over duplicate blocks */
void orc_bcsr_val_at(const orc_op *op, int64_t row, int64_t col, cplx *out) {
    int32_t bs = op->bs;
    int32_t br = (int32_t)(row / bs), bc = (int32_t)(col / bs);
    int32_t ro = (int32_t)(row - (int64_t)br * bs), co = (int32_t)(col - (int64_t)bc * bs);
    cplx o = 0.0;
    for (int32_t i = op->browptr[br]; i < op->browptr[br + 1]; i++)
        if (op->bcol[i] == bc) o += op->blocks[(size_t)i * bs * bs + (size_t)ro * bs + co];
    *out = o;
}

int64_t orc_op_dim(const orc_op *op) { return op->dim; }
/* number of rows of the output (differs from dim for rectangular CSR / block-CSR) */
int64_t orc_op_nrow(const orc_op *op) {
    if (op->kind == OP_CSR) return op->nrow;
    if (op->kind == OP_BCSR) return (int64_t)op->nbrow * op->bs;
    return op->dim;
}

/* ------------------------------------------------------------------ GCR ------------------ */

static cplx *vnew(int64_t n) { return (cplx *)malloc(sizeof(cplx) * (size_t)n); }

/* src/GCR.h:158-302.  Returns the number of iterations performed (global_count).
 * hist[0] = step-0 entry, hist[k] = sqrt(|r|^2)/|b| printed at step k (src/GCR.h:214,271-272);
 * at most hist_cap entries are stored.  *converged = 1 unless global_count == max_iter
 * (src/GCR.h:294-298). */
static int gcr_solve_impl(orc_op *A, const orc_gcr_param *gp, const cplx *rhs, cplx *x, double *hist, int hist_cap, int *converged);
int orc_gcr_solve(orc_op *A, const orc_gcr_param *gp, const cplx *rhs, cplx *x, double *hist, int hist_cap,
                  int *converged) {
    const orc_op *b0 = A->kind == OP_DIRAC && A->D ? A->D : A;
    /* The row map in force for THIS solve: the operator's own (a level of a hierarchy) or, for an operator without one, the map
     * orc_set_device_model / _plane / _xr_banded set — never the one an ENCLOSING solve on another operator put in place (a coarse solve
     * inside the preconditioner of a solve on a banded level would otherwise inherit that level's bands).  Nested solves do the same. */
    static int depth = 0;
    static int64_t base_band, base_plane;
    static int base_per, base_ib, base_xb;
    if (depth == 0) { base_band = g_dev_band; base_plane = g_dev_plane; base_per = g_dev_per; base_ib = g_dev_init_banded; base_xb = g_dev_xr_banded; }
    const int64_t band = g_dev_band, plane = g_dev_plane;
    const int per = g_dev_per, ib = g_dev_init_banded, xb = g_dev_xr_banded;
    if (b0->has_rowmap) {
        g_dev_band = b0->rm_band; g_dev_per = b0->rm_per; g_dev_plane = b0->rm_plane; g_dev_init_banded = b0->rm_init_banded; g_dev_xr_banded = b0->rm_xr_banded;
    } else {
        g_dev_band = base_band; g_dev_per = base_per; g_dev_plane = base_plane; g_dev_init_banded = base_ib; g_dev_xr_banded = base_xb;
    }
    depth++;
    const int rc = gcr_solve_impl(A, gp, rhs, x, hist, hist_cap, converged);
    depth--;
    g_dev_band = band; g_dev_per = per; g_dev_plane = plane; g_dev_init_banded = ib; g_dev_xr_banded = xb;
    return rc;
}
static int gcr_solve_impl(orc_op *A, const orc_gcr_param *gp, const cplx *rhs, cplx *x, double *hist, int hist_cap,
                          int *converged) {
    int64_t n = A->dim;
    /* mode selection, src/GCR.h:171-185 */
    int truncation, restart, storage = gp->max_iter;
    if (gp->truncation != 0) { truncation = gp->truncation; storage = truncation; } else truncation = gp->max_iter;
    if (gp->restart != 0) { restart = gp->restart; storage = restart; } else restart = gp->max_iter;
    (void)truncation;
    if (storage < 1) storage = 1; /* max_iter = 0 with neither mode: the reference would new[0]; unused in practice */
    if (restart < 1) restart = 1;

    cplx *r = vnew(n), *p = vnew(n), *Ap = vnew(n), *Ar = vnew(n), *t = vnew(n);
    cplx *z = NULL;
    /* r = rhs (src/GCR.h:189) — the reference ignores x0 here */
    if (gp->use_x0) {
        orc_op_residual(A, x, rhs, r);
    } else {
        memcpy(r, rhs, sizeof(cplx) * (size_t)n);
    }
    if (gp->flexible && gp->right_precond) {
        z = vnew(n);
        orc_op_apply(gp->right_precond, r, z);
        memcpy(p, z, sizeof(cplx) * (size_t)n);
    } else {
        memcpy(p, r, sizeof(cplx) * (size_t)n); /* p = r */
    }
    orc_op_apply(A, p, Ap);                  /* Ap = A p  (src/GCR.h:191) */
    memcpy(Ar, Ap, sizeof(cplx) * (size_t)n); /* Ar = Ap */
    if (!gp->flexible) {
        /* src/GCR.h:197-204: r = Mr(r); r = Ml(r)  — AFTER p and Ap were formed */
        if (gp->right_precond) { orc_op_apply(gp->right_precond, r, t); memcpy(r, t, sizeof(cplx) * (size_t)n); }
        if (gp->left_precond) { orc_op_apply(gp->left_precond, r, t); memcpy(r, t, sizeof(cplx) * (size_t)n); }
    }
    cplx **Aps = (cplx **)calloc((size_t)storage, sizeof(cplx *));
    cplx **ps = (cplx **)calloc((size_t)storage, sizeof(cplx *));
    Aps[0] = vnew(n); ps[0] = vnew(n);
    memcpy(Aps[0], Ap, sizeof(cplx) * (size_t)n);
    memcpy(ps[0], p, sizeof(cplx) * (size_t)n);

    /* order 3, lean restart cycles (see orc_set_device_lean) */
    const int flex = gp->flexible && gp->right_precond;
    int storage_dev = storage;   /* gcr.hip gcr_prepare: a restart cycle longer than the whole solve keeps max_iter + 1 slots */
    if (gp->restart != 0 && gp->max_iter >= 1 && gp->max_iter + 1 < storage_dev) storage_dev = gp->max_iter + 1;
    const int lean = g_sum_order == 3 && g_dev_lean && g_dev_blocks != 1 && gp->restart != 0 && storage_dev <= LEAN_MAX && !gp->left_precond &&
                     (!gp->right_precond || flex);
    cplx *LP[LEAN_MAX + 1];      /* LP[0] = P0, LP[m] = D_m */
    cplx lT[LEAN_MAX][LEAN_MAX], lt[LEAN_MAX], lcx[LEAN_MAX], betas[LEAN_MAX];
    int npend = 0;
    for (int m = 0; m <= LEAN_MAX; m++) LP[m] = NULL;
    if (lean) {
        LP[0] = vnew(n);
        memcpy(LP[0], p, sizeof(cplx) * (size_t)n);
        for (int a = 0; a < LEAN_MAX; a++) { lt[a] = 0.0; lcx[a] = 0.0; for (int b2 = 0; b2 < LEAN_MAX; b2++) lT[a][b2] = 0.0; }
    }
    /* order 3: step 0's sums come out of the kernel that embeds the apply when the device fuses the start */
    /* (csrc/gcr.hip: fuse_start — no x0, no preconditioner hooks — or fuse_init — a lean solve shorter than its restart cycle, whose first
     * direction IS its start residual, x0 or not; every other start takes the plain-order kernels) */
    const int dev_lean_like = g_dev_lean && gp->restart != 0 && storage_dev <= LEAN_MAX && !gp->left_precond && !gp->right_precond;
    const int short_lean = dev_lean_like && gp->max_iter >= 1 && gp->max_iter < gp->restart;
    const int init_banded = g_dev_init_banded && !gp->right_precond && !gp->left_precond && (!gp->use_x0 || short_lean);
    g_dev_banded_now = init_banded;
    double bnorm2 = orc_sqnorm(n, rhs);
    double bnorm = sqrt(bnorm2);
    if (hist && hist_cap > 0) hist[0] = sqrt(orc_sqnorm(n, r)) / bnorm;
    if (gp->verbose) printf("Step %d residual norm = %.10e\n", 0, sqrt(orc_sqnorm(n, r)) / bnorm);
    /* order 3: the device keeps <Ap_i,Ap_i> from the pass that formed Ap_i (gcr.hip: den[slot]) instead of summing it
     * again for every beta — the same number unless the two passes deal their rows differently (banded RowMap) */
    cplx *den_cache = (cplx *)calloc((size_t)storage, sizeof(cplx));
    int cur_slot = 0, first_step = 1;

    int iter_count = 0, global_count = 0;
    cplx *Ap_corr = vnew(n), *p_corr = vnew(n);
    double rn2;
    do {
        global_count++;
        iter_count++;
        /* alpha = r.dot(Ap) / Ap.dot(Ap)   (src/GCR.h:230) */
        cplx num, den;
        g_dev_banded_now = first_step ? init_banded : 0;
        orc_dot(n, r, Ap, &num);
        orc_dot(n, Ap, Ap, &den);
        g_dev_banded_now = 0;
        first_step = 0;
        den_cache[cur_slot] = den;
        cplx alpha = num / den;
        /* x = x + p*alpha ; r = r - Ap*alpha  (src/GCR.h:232-233) */
        if (lean) {   /* gcr_dev.h lean_pending_update: x += alpha p_cur recorded in terms of P0 and D_1..D_cur */
            const int sl = cur_slot;
            if (sl == 0) { lcx[0] = alpha; for (int m = 1; m < LEAN_MAX; m++) lcx[m] = 0.0; }
            else {
                lcx[0] = lcx[0] + cm(alpha, lt[sl]);
                for (int m = 1; m <= sl; m++) lcx[m] = lcx[m] + cm(alpha, lT[sl][m]);
            }
            npend = sl + 1;
        } else
        for (int64_t i = 0; i < n; i++) x[i] = x[i] + alpha * p[i];
        for (int64_t i = 0; i < n; i++) r[i] = r[i] - alpha * Ap[i];
        const cplx *dir = r; /* the vector the new direction is built from */
        if (gp->flexible && gp->right_precond) {
            orc_op_apply(gp->right_precond, r, z);
            dir = z;
        } else if (gp->right_precond) { /* src/GCR.h:236-238 */
            orc_op_apply(gp->right_precond, r, t);
            memcpy(r, t, sizeof(cplx) * (size_t)n);
        }
        orc_op_apply(A, dir, Ar); /* src/GCR.h:242 */
        if (gp->left_precond) {   /* src/GCR.h:245-247 */
            orc_op_apply(gp->left_precond, Ar, t);
            memcpy(Ar, t, sizeof(cplx) * (size_t)n);
        }
        int lim = storage < iter_count ? storage : iter_count; /* src/GCR.h:251 */
        for (int64_t i = 0; i < n; i++) { Ap_corr[i] = 0.0; p_corr[i] = 0.0; }
        for (int i = 0; i < lim; i++) { /* src/GCR.h:257-262 */
            cplx bn, bd;
            g_dev_banded_now = 1; /* the beta numerators are summed by the apply-embedding kernel / multidot_kernel (RowMap) */
            orc_dot(n, Ar, Aps[i], &bn);
            g_dev_banded_now = 0;
            if (g_sum_order == 3) bd = den_cache[i];
            else orc_dot(n, Aps[i], Aps[i], &bd);
            cplx beta = bn / bd;
            if (lean) betas[i] = beta;
            for (int64_t j = 0; j < n; j++) p_corr[j] = p_corr[j] - beta * ps[i][j];
            for (int64_t j = 0; j < n; j++) Ap_corr[j] = Ap_corr[j] - beta * Aps[i][j];
        }
        for (int64_t i = 0; i < n; i++) p[i] = dir[i] + p_corr[i];   /* src/GCR.h:265 */
        for (int64_t i = 0; i < n; i++) Ap[i] = Ar[i] + Ap_corr[i];  /* src/GCR.h:266 */
        if (lean && iter_count % restart == 0) {
            /* the step that closes a cycle (gcr.hip build_close_kernel / close_x_kernel): cp_0 = sum_j beta_j t_j (t_0 = 1),
             * cp_m = sum_{j >= m} beta_j T_jm (T_mm = 1); x += sum_j cx_j p_j; P0' = dir - sum_j cp_j p_j  (p_0 = P0, p_m = D_m) */
            const int R = lim;
            cplx cp[LEAN_MAX];
            for (int m = 0; m < R; m++) {
                cplx a = 0.0;
                if (m == 0) for (int j = 0; j < R; j++) a = a + cm(betas[j], j == 0 ? (cplx)1.0 : lt[j]);
                else for (int j = m; j < R; j++) a = a + cm(betas[j], j == m ? (cplx)1.0 : lT[j][m]);
                cp[m] = a;
            }
            cplx *np0 = vnew(n);
            for (int64_t i = 0; i < n; i++) {
                cplx xv = x[i];
                for (int j = 0; j < R; j++) xv = xv + cm(lcx[j], LP[j][i]);
                x[i] = xv;
                cplx pc = 0.0;
                for (int j = 0; j < R; j++) pc = pc - cm(cp[j], LP[j][i]);
                np0[i] = dir[i] + pc;
            }
            free(LP[0]);
            LP[0] = np0;
            npend = 0;
        } else if (lean) {
            /* inside a cycle (gcr.hip build_lean_kernel): row k = lim of the table, and D_k = what direction k was started from */
            const int k = lim;
            if (k < LEAN_MAX) {
                cplx a = 0.0;
                for (int j = 0; j < k; j++) a = a - cm(betas[j], j == 0 ? (cplx)1.0 : lt[j]);
                lt[k] = a;
                for (int m = 1; m < k; m++) {
                    a = 0.0;
                    for (int j = m; j < k; j++) a = a - cm(betas[j], j == m ? (cplx)1.0 : lT[j][m]);
                    lT[k][m] = a;
                }
                lT[k][k] = 1.0;
                if (!LP[k]) LP[k] = vnew(n);
                memcpy(LP[k], dir, sizeof(cplx) * (size_t)n);
            }
        }
        g_dev_banded_now = lean && g_dev_xr_banded && !flex && gp->restart > 1 && global_count < gp->max_iter;   /* (csrc/gcr.hip xr_fuse: cycles of at least two steps) */
        rn2 = orc_sqnorm(n, r);
        g_dev_banded_now = 0;
        if (hist && global_count < hist_cap) hist[global_count] = sqrt(rn2) / bnorm;
        if (gp->verbose) printf("Step %d residual norm = %.10e\n", global_count, sqrt(rn2) / bnorm);
        if (iter_count % restart == 0) { /* src/GCR.h:277-283: wipe (slots are never read again before being rewritten) */
            iter_count = 0;
        }
        int slot = iter_count % storage; /* src/GCR.h:286-287 */
        if (!Aps[slot]) { Aps[slot] = vnew(n); ps[slot] = vnew(n); }
        memcpy(Aps[slot], Ap, sizeof(cplx) * (size_t)n);
        memcpy(ps[slot], p, sizeof(cplx) * (size_t)n);
        cur_slot = slot;
    } while ((rn2 / bnorm2) > gp->tol * gp->tol && global_count < gp->max_iter); /* src/GCR.h:288 */

    if (lean && npend > 0)   /* the updates still pending when the solve ends (gcr.hip flush_x_kernel) */
        for (int64_t i = 0; i < n; i++) {
            cplx xv = x[i];
            for (int j = 0; j < npend; j++) xv = xv + cm(lcx[j], LP[j][i]);
            x[i] = xv;
        }
    for (int m = 0; m <= LEAN_MAX; m++) free(LP[m]);
    free(g_last_r);
    g_last_r = vnew(n);
    memcpy(g_last_r, r, sizeof(cplx) * (size_t)n);
    if (converged) *converged = (global_count == gp->max_iter) ? 0 : 1;
    if (gp->verbose) {
        if (global_count == gp->max_iter)
            printf("GCR did not converge after %d steps! Residual norm = %.10e\n", gp->max_iter, sqrt(rn2) / bnorm);
        else
            printf("GCR converged after %d steps. Residual norm=%.10e\n", global_count, sqrt(rn2) / bnorm);
    }
    for (int i = 0; i < storage; i++) { free(Aps[i]); free(ps[i]); }
    free(Aps); free(ps); free(Ap_corr); free(p_corr); free(den_cache);
    free(r); free(p); free(Ap); free(Ar); free(t); free(z);
    return global_count;
}

/* GCR as an Operator (src/GCR.h:62-68): x = init_rand(2); solve(f, x); return x.
 * x0_mode 0 reproduces the reference when the caller supplies the same random x0 through
 * orc_op_gcr_set_x0 (libc rand() is compiler/evaluation-order dependent, SURVEY.md §0 fact 9, so
 * the oracle never calls it); x0_mode 1 starts from x0 = 0. */
orc_op *orc_op_gcr(orc_op *A, const orc_gcr_param *gp, int x0_mode) {
    orc_op *op = (orc_op *)calloc(1, sizeof(orc_op));
    op->kind = OP_GCR; op->dim = A ? A->dim : 0; op->A = A; op->gp = *gp; op->x0_mode = x0_mode;
    return op;
}
void orc_op_gcr_set_operator(orc_op *g, orc_op *A) { g->A = A; g->dim = A->dim; } /* GCR::initialise src/GCR.h:31 */
static const cplx *g_gcr_x0 = NULL;
void orc_op_gcr_set_x0(const cplx *x0) { g_gcr_x0 = x0; }

static void gcr_apply(orc_op *op, const cplx *f, cplx *y) {
    int64_t n = op->dim;
    if (op->x0_mode == 0 && g_gcr_x0) memcpy(y, g_gcr_x0, sizeof(cplx) * (size_t)n);
    else memset(y, 0, sizeof(cplx) * (size_t)n);
    orc_gcr_solve(op->A, &op->gp, f, y, NULL, 0, NULL);
}

void orc_mg_apply(orc_mg *mg, const cplx *f, cplx *y);

void orc_op_apply(orc_op *op, const cplx *x, cplx *y) {
    switch (op->kind) {
        case OP_CSR: csr_apply(op, x, y); break;
        case OP_DIRAC: dirac_apply(op, x, y); break;
        case OP_BCSR: bcsr_apply(op, x, y); break;
        case OP_GCR: gcr_apply(op, x, y); break;
        case OP_MG: orc_mg_apply(op->mg, x, y); break;
    }
}

/* multigrid cycle as an Operator (what MG::operator() was meant to be, src/MG.h:124-129) */
orc_op *orc_op_mg(orc_mg *mg, int64_t dim) {
    orc_op *op = (orc_op *)calloc(1, sizeof(orc_op));
    op->kind = OP_MG; op->dim = dim; op->mg = mg;
    return op;
}

void orc_op_free(orc_op *op) {
    if (!op) return;
    if (op->owns) { free(op->browptr); free(op->bcol); free(op->blocks); }
    free(op);
}

/* ------------------------------------------------------------------ data loader ---------- */

/* src/Parse.cpp:64-90  text-CSR reader: "nrow ncol nnz", then nrow row offsets (ROW[nrow] = nnz
 * implied, src/Operator.h:61), then nnz lines "col (re,im)".  Two-call protocol: first call with
 * NULL arrays returns the sizes. */
int orc_read_text_csr(const char *path, int64_t *nrow, int64_t *ncol, int64_t *nnz, int64_t *rowptr, int64_t *col,
                      cplx *val) {
    FILE *f = fopen(path, "r");
    if (!f) return -1;
    long long a, b, c;
    if (fscanf(f, "%lld %lld %lld", &a, &b, &c) != 3) { fclose(f); return -2; }
    *nrow = a; *ncol = b; *nnz = c;
    if (!rowptr) { fclose(f); return 0; }
    for (int64_t i = 0; i < a; i++) {
        long long v;
        if (fscanf(f, "%lld", &v) != 1) { fclose(f); return -3; }
        rowptr[i] = v;
    }
    rowptr[a] = c;
    for (int64_t i = 0; i < c; i++) {
        long long cc; double re, im;
        if (fscanf(f, "%lld (%lf,%lf)", &cc, &re, &im) != 3) { fclose(f); return -4; }
        col[i] = cc; val[i] = re + im * I;
    }
    fclose(f);
    return 0;
}

/* ------------------------------------------------------------------ generators ----------- */

static inline uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
/* repo-owned deterministic RHS on the reference's 0.001 grid (src/Fields.h:133); same recipe as
 * oracle/ref_harness.cpp:fill_rhs and the product's mgcr_fill_rhs */
void orc_fill_rhs(int64_t n, uint64_t seed, cplx *out) {
    for (int64_t i = 0; i < n; i++) {
        uint64_t a = splitmix64(seed * 0x100000001B3ull + 2 * (uint64_t)i);
        uint64_t b = splitmix64(seed * 0x100000001B3ull + 2 * (uint64_t)i + 1);
        out[i] = ((double)(a % 2000) / 1000. - 1.) + ((double)(b % 2000) / 1000. - 1.) * I;
    }
}

/* 3-D 7-point Poisson: lexicographic, diag 6, off-diag -1, Dirichlet
 * truncation; columns ascending within a row.  nnz = 7 n^3 - 6 n^2. */
void orc_poisson3d(int64_t n, int64_t *rowptr, int64_t *col, cplx *val) {
    int64_t p = 0;
    for (int64_t i = 0; i < n; i++) for (int64_t j = 0; j < n; j++) for (int64_t k = 0; k < n; k++) {
        int64_t r = (i * n + j) * n + k;
        rowptr[r] = p;
        if (i > 0)     { col[p] = r - n * n; val[p++] = -1.; }
        if (j > 0)     { col[p] = r - n;     val[p++] = -1.; }
        if (k > 0)     { col[p] = r - 1;     val[p++] = -1.; }
        col[p] = r; val[p++] = 6.;
        if (k < n - 1) { col[p] = r + 1;     val[p++] = -1.; }
        if (j < n - 1) { col[p] = r + n;     val[p++] = -1.; }
        if (i < n - 1) { col[p] = r + n * n; val[p++] = -1.; }
    }
    rowptr[n * n * n] = p;
}
