// Internal declarations of libmgcr_hip.so (gfx950 only; no compatibility layers).
#pragma once
#include <hip/hip_runtime.h>
#include <climits>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/mgcr.h"

namespace mgcr {

typedef double2 cplx;  // interleaved (re, im), 16 B: one dwordx4 per element per lane

void set_error(const char *fmt, ...);
int hip_fail(hipError_t e, const char *what, const char *file, int line);

#define MGCR_HIP(call)                                                     \
    do {                                                                   \
        hipError_t e__ = (call);                                           \
        if (e__ != hipSuccess) return mgcr::hip_fail(e__, #call, __FILE__, __LINE__); \
    } while (0)
#define MGCR_CHECK(cond, code, ...)        \
    do {                                   \
        if (!(cond)) {                     \
            mgcr::set_error(__VA_ARGS__);  \
            return (code);                 \
        }                                  \
    } while (0)
#define MGCR_TRY(call)                 \
    do {                               \
        int rc__ = (call);             \
        if (rc__ != MGCR_OK) return rc__; \
    } while (0)

// ----------------------------------------------------------------------------------------------
// Reductions.  Every reducing kernel runs a grid of at most RED_MAX_BLOCKS blocks of RED_THREADS
// threads (= 512 x 16 waves: two blocks per CU fill all 32 wave slots of the 256 CUs) and writes
// one partial per block and scalar into a [nscal][RED_MAX_BLOCKS] slab.  Whoever consumes the
// scalars (the next kernel of the iteration) folds the slab in a fixed order inside each of its
// own blocks, so that every block sees bit-identical values, nothing goes through atomics, no
// extra "finalise" launch is needed and results are reproducible run to run.
constexpr int RED_THREADS = 1024;
constexpr int RED_MAX_BLOCKS = 512;
constexpr int STEN_TILE = 512;   // rows per workgroup of the stand-alone stencil SpMV
constexpr int STEN_MAX = 9;   // slots of a stencil view (CsrDev::sten_*): kernels are instantiated for 7 and 9

struct Context {
    bool ready = false;
    int device = -1;
    int n_cu = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    // pinned host mailbox for small device->host reads (scalars, flags)
    double *h_mail = nullptr;  // 64 doubles
    std::recursive_mutex mtx;
};
Context &ctx();
int require_ctx();

struct Vec {
    int64_t n = 0;
    cplx *d = nullptr;
    bool owns = true;
    // set by mgcr_vec_zero, cleared by whatever writes the Field next (every writer takes the pointer through w()): a solve
    // that has to be repeated on another kernel path (gcr.hip gcr_run) then re-zeroes x instead of keeping a copy of it
    bool zero_known = false;
    cplx *w() { zero_known = false; return d; }
};

enum OpKind { OP_CSR = 1, OP_DIRAC = 2, OP_BCSR = 3, OP_GCR = 4, OP_MG = 5 };

// ELL slab + CSR tail (int32 indices).
//   ELL: W = nchunk*L entries per row, L lanes co-operate on a row.
//        element (row, w) lives at ((w / L) * npad + row) * L + (w % L)  -> for a fixed chunk,
//        consecutive (row, lane) pairs are consecutive in memory: coalesced 16-B val loads.
//        L = 1 is the classic column-major ELL, one thread per row, entries in CSR order (so
//        the per-row sum runs in the reference's order, src/Operator.h:338-341).
//   tail: rows longer than W keep their remaining entries in CSR (tail_rows lists them).
struct CsrDev {
    int64_t nrow = 0, ncol = 0, nnz = 0;
    int32_t W = 0, L = 1, nchunk = 0;
    int64_t npad = 0;
    cplx *ell_val = nullptr;       // complex slab, or ...
    double *ell_val_re = nullptr;  // ... real slab when every stored value has a zero imaginary part (12 B/nnz)
    int32_t *ell_col = nullptr;
    // Row-pattern dictionary (spmv.hip): rows whose (column - row, value) tuples coincide share one
    // table entry and store a 2-byte id.  pat_mode 1: offsets and values in the table (no slab at all);
    // 2: offsets only (values stay in the slab, ell_col is dropped); 0: plain ELL.
    int pat_mode = 0;
    int32_t npat = 0;
    uint16_t *pat_id = nullptr;   // [npad]
    int32_t *pat_off = nullptr;   // [npat][W]
    double *pat_re = nullptr, *pat_im = nullptr;  // [npat][W], mode 1
    bool pat_real = false;        // mode 1: every imaginary part is zero
    int64_t reach = 0;            // max |column - row| over the pattern table (0: unknown) — how far a row's gathers go
    // Stencil view of a mode-1 dictionary (spmv.hip sten_try), what the apply kernels read when it exists: the
    // ascending superset of all patterns' column offsets (sten_ns <= STEN_MAX slots), ONE value per slot, and per
    // wave of 64 rows one 64-bit presence word per slot (sten_planes[wave * sten_stride + slot], bit l = row
    // 64 * wave + l has the slot).  A row's x loads then depend on nothing but the row number.
    int32_t sten_ns = 0, sten_stride = 0;   // sten_ns: slots the operator has (reporting); the arrays below are in KERNEL layout:
    int32_t sten_kernel_ns = 0;             // 7 or 9 slots (unused ones: offset 0, value 0, no presence bits)
    int32_t sten_pre = 0;         // rare-tail layout: slot 7 is summed BEFORE the common ones (a row block's lower halo column: first in storage order)
    uint32_t sten_rare = 0;       // non-zero: rare-tail layout — slots 0..6 common, slots 7, 8 present in fewer than 1/16 of the rows and
                                  // behind every common slot in column order: loaded only by waves whose presence word is not 0
    int32_t sten_off[16] = {};
    double sten_re[16] = {}, sten_im[16] = {};
    uint64_t *sten_planes = nullptr;
    uint32_t sten_near = 0;       // slots within STEN_TILE / 2 rows of the diagonal: the stand-alone SpMV serves them from an LDS window
    int32_t sten_halo = 0;        // largest |offset| among them (0: no window)
    uint32_t sten_near_f = 0;     // the same for the fused GCR step kernels, whose window spans RED_THREADS rows: within RED_THREADS / 2
    int32_t sten_halo_f = 0;
    // Banded irregular matrices (spmv.hip ell_window_try): when at least 90 % of the slab's columns lie within win_h rows of their
    // row, the stand-alone apply stages x[r0 - win_h, r0 + 1024 + win_h) of each 1024-row tile in LDS and serves those gathers
    // from there (0: no window)
    int32_t win_h = 0;
    // ... and the tile's TAIL entries are multiplied in the same launch from the same window (spmv.hip ell_spmv_window<.., TAIL>):
    int32_t *win_tile_tail = nullptr;   // [ntiles + 1] first tail row (index into tail_rows) whose row is >= tile * 1024
    int32_t *win_row_tail = nullptr;    // [nrow] the row's index into tail_rows; -1: no tail, or a tail longer than a chunk (csr_tail_kernel's)
    int64_t n_tail_rows = 0, tail_nnz = 0;
    int32_t *tail_rows = nullptr;   // [n_tail_rows]
    int32_t *tail_ptr = nullptr;    // [n_tail_rows+1]
    int32_t *tail_col = nullptr;    // [tail_nnz]
    cplx *tail_val = nullptr;
    // how the tail kernels are dealt the tail rows (spmv.hip csr_tail_chunk_kernel): runs of consecutive tail rows with at most
    // TAIL_CAP entries and TAIL_THREADS rows together ("chunks": one workgroup each, products staged in LDS, every row then summed
    // by one thread in CSR order), and the rows that are longer than a chunk on their own (one wave each, csr_tail_kernel)
    int32_t n_tail_chunks = 0, n_tail_long = 0;
    int4 *tail_chunk = nullptr;     // [n_tail_chunks] {first tail row (index into tail_rows), one past the last, first entry, one past the last entry}
    int32_t *tail_long = nullptr;   // [n_tail_long] indices into tail_rows
};
constexpr int TAIL_CAP = 2048;      // entries per chunk (32 KB of LDS products, 8 entries per thread in flight)
constexpr int TAIL_THREADS = 256;

struct BcsrDev {
    int32_t nbrow = 0, nbcol = 0, bs = 0, nblocks = 0;
    int32_t *browptr = nullptr, *bcol = nullptr;
    cplx *blocks = nullptr;  // [nblocks][bs][bs] row-major
    int32_t *order = nullptr;  // block rows by falling block count: the order the wave kernels are dealt them (null: as stored)
};

struct GcrState;
struct MgState;
struct Comm;
struct DistCsr;

struct Op {
    OpKind kind;
    int64_t dim = 0, nrow = 0;
    CsrDev csr;          // OP_CSR
    Op *base = nullptr;  // OP_DIRAC: borrowed CSR
    cplx k = {0., 0.};   // OP_DIRAC
    BcsrDev bcsr;        // OP_BCSR
    GcrState *gcr = nullptr;  // OP_GCR
    MgState *mg = nullptr;    // OP_MG
    DistCsr *dist = nullptr;  // OP_CSR / OP_BCSR row block of a distributed matrix (comm.hip)
    Comm *comm = nullptr;     // communicator the operator's Fields are distributed over (borrowed)
};

// ---- blas1.hip -------------------------------------------------------------------------------
int red_grid(int64_t n);
int k_copy(cplx *dst, const cplx *src, int64_t n);
int k_zero(cplx *dst, int64_t n);
int k_copy_apply(cplx *dst, const cplx *src, int64_t n);  // skip-aware (operator applies)
int k_zero_apply(cplx *dst, int64_t n);
int k_set_constant(cplx *dst, cplx c, int64_t n);
int k_fill_rhs(cplx *dst, int64_t n, uint64_t seed, int64_t offset);
int k_add_scaled(cplx *out, const cplx *a, cplx alpha, const cplx *b, int64_t n);  // out = a + alpha*b
int k_scale(cplx *v, cplx alpha, int64_t n);
int k_gamma5(cplx *out, const cplx *in, int64_t n, int64_t inner);
// generic reductions into a partial slab, then fold to `out` (device, nscal cplx) by a 1-block kernel
int k_dot_partials(const cplx *a, const cplx *b, int64_t n, double *parts /*[2][RED_MAX_BLOCKS]*/, int *nblk);
int k_fold(const double *parts, int nblk, int nscal, double *out_dev);
// two slabs in one launch: na scalars of pa -> outa, nb scalars of pb -> outb
int k_fold2(const double *pa, int na, double *outa, const double *pb, int nb, double *outb, int nblk);

// ---- spmv.hip --------------------------------------------------------------------------------
int csr_build_device(int64_t nrow, int64_t ncol, const int64_t *h_rowptr, const int64_t *h_col,
                     const double *h_val_ri, CsrDev *out);
void csr_free(CsrDev *c);
bool set_patterns_enabled(bool on);
bool set_stencil_enabled(bool on);
bool csr_stencil_active(const CsrDev &A);  // the apply kernels read A through its stencil view (CsrDev::sten_*)
int set_spmv_part(int part);
bool set_lean_enabled(bool on);
bool set_fuse_enabled(bool on);
bool set_graph_enabled(bool on);
bool set_resident_enabled(bool on);   // gcr_resident.hip: whole small solves in one launch
constexpr int MGCR_INT_GAVE_UP = 1001;   // internal status: a one-launch path gave up (gcr_run repeats the solve; never leaves the library)
int resident_check(bool internal = false);   // did such a solve give up (launch not co-resident)?  Called at host synchronisation points
                                       // (internal: MGCR_INT_GAVE_UP instead of MGCR_ERR_HIP + message)
void resident_shutdown();
int64_t resident_solve_count();
bool set_stepbuild_enabled(bool on);   // gcr_stepbuild.hip: apply + dots + build of a lean step as one launch
int64_t stepbuild_launch_count();
int coherence_selftest(int steps, int coherent, int64_t *rows_wrong);   // gcr_resident.hip
bool stepbuild_is_enabled();
bool launch_is_coresident(const void *kernel, int threads, size_t dyn_lds, int grid);   // gcr_stepbuild.hip: asks the runtime
bool one_launch_paths_enabled();       // either of the two above
// y = A x   or (shift) y = w - k*(A x) with w = x unless given (w = b, k = 1: the residual b - A x in one pass);
// dist != nullptr: row block with halo exchange
int csr_apply(const CsrDev &A, const cplx *x, cplx *y, bool shift, cplx k, DistCsr *dist = nullptr, const cplx *w = nullptr);
// SpMV fused with <y, v_j> partials (gcr_fused.hip); parts laid out like gcr.hip's partsB, red_grid(nrow) partials each
bool csr_fusable(const CsrDev &A, const DistCsr *dist);
struct RowMap;  // gcr_dev.h
struct PwTail;  // pw_tail_dev.h
bool csr_step_apply_has_pw_tail(const CsrDev &A, const DistCsr *dist);
int csr_step_apply(const CsrDev &A, const cplx *x, cplx *y, bool shift, cplx k, const cplx *const *vecs, int nd, double *parts,
                   DistCsr *dist, const RowMap &rm, const PwTail *pw = nullptr);
struct DevState;
struct LeanCoef;
bool csr_xr_fusable(const CsrDev &A, const DistCsr *dist);
bool csr_apply_carry(const CsrDev &A, const cplx *x, cplx *y, bool shift, cplx k, const cplx *w);   // gcr_fused.hip
int csr_xr_fuse_kind(const CsrDev &A, const DistCsr *dist);
int csr_step_apply_xr(const CsrDev &A, const cplx *r_in, const cplx *ap, cplx *r_out, cplx *y, bool shift, cplx k, const cplx *const *vecs,
                      int nd, double *parts, double *partsR, DevState *st, int it, const double *partsA, int nblkA, int strideA,
                      cplx *den_slot, int slot, LeanCoef *lc, const RowMap &rm);
int csr_init_apply(const CsrDev &A, const cplx *r0, cplx *aps0, bool shift, cplx k, const cplx *b, double *partsA, double *partsR,
                   double *partsN, DistCsr *dist, const RowMap &rm);
int bcsr_build_device(int32_t nbrow, int32_t nbcol, int32_t bs, const int32_t *h_browptr, const int32_t *h_bcol,
                      const double *h_blocks, BcsrDev *out);
void bcsr_free(BcsrDev *b);
// xh / nb_own: row block of a distributed operator — block columns >= nb_own live in the halo segment xh
int bcsr_apply(const BcsrDev &A, const cplx *x, cplx *y, const cplx *xh = nullptr, int32_t nb_own = INT32_MAX);

// Device-side "this solve is over" predicate consulted by operator-apply kernels that run inside a
// solver iteration: p points at {stop_at, base} of the solver's DevState (gcr.hip) and the kernel
// returns at once when stop_at < base + it.
struct SkipRef {
    const int *p = nullptr;
    int it = 0;
};
void set_apply_skip(SkipRef s);
SkipRef get_apply_skip();

// ---- comm.hip --------------------------------------------------------------------------------
// overlap_interior: the caller multiplies the rows that need no halo between begin and end — a peer-write exchange is then split
// into store + publish (begin) and the wait for the neighbours (end)
int dist_halo_begin(DistCsr *d, const cplx *x, bool overlap_interior = false);
int64_t dist_halo_split_count();
bool set_halo_split(bool on);
bool set_pw_tail_enabled(bool on);     // comm.hip / pw_tail_dev.h: fold + cross-rank sum inside the producing kernel
int64_t comm_pw_tail_count();
int dist_halo_end(DistCsr *d);
bool dist_halo_overlaps();  // MGCR_HALO_OVERLAP: exchange on the communication stream, overlapped with interior rows
const cplx *dist_halo_ptr(DistCsr *d);  // halo segment of the exchange begun last (peer-write: alternates between two slots)
int dist_halo_kind(DistCsr *d);
void dist_info(DistCsr *d, const cplx **xh, int64_t *interior_begin, int64_t *interior_end);
void dist_free(DistCsr *d);
Comm *dist_comm(DistCsr *d);
void dist_sizes(DistCsr *d, int64_t *nloc, int64_t *nh, int64_t *row0, int64_t *n_global, int *rank, int *nranks);
int comm_allreduce_host_pub(Comm *c, double *buf, int64_t count);
int dist_exchange_rows_host(DistCsr *d, const double *own, int w, double *halo);
int dist_csr_create(Comm *c, int64_t n_global, int64_t row0, int64_t nloc, const int64_t *rowptr, const int64_t *col,
                    const double *val_ri, Op *op);
int dist_bcsr_create(Comm *c, int64_t nb_global, int64_t brow0, int32_t nbloc, int32_t bs, const int32_t *browptr,
                     const int64_t *bcol_global, const double *blocks_ri, Op *op);
int comm_nranks(Comm *c);
bool comm_collectives(Comm *c);
int comm_allreduce_dev(Comm *c, double *dbuf, int count);
int comm_fold_allreduce(Comm *c, const double *pa, int na, const double *pb, int nb, double *out, int nblk);
int comm_check(Comm *c);
int comm_live_count();  // communicators alive in this process
int comm_check_all();   // every live communicator; called where results are handed back to the host
constexpr int PW_MAX_RANKS = 16;  // peer-write all-reduce (comm.hip): one lane per rank

// ---- dense.hip -------------------------------------------------------------------------------
constexpr int DENSE_MAX_ROWS = 2048;   // direct coarsest solve: the inverse is n x n complex fp64 (64 MiB at the limit)
int dense_inverse_of(Op *A, int64_t n, cplx **inv_out);
int dense_apply(const cplx *inv, int64_t n, const cplx *b, cplx *x);

// ---- mg.hip ----------------------------------------------------------------------------------
int mg_create(Op *A, const mgcr_mg_param *p, MgState **out);
void mg_destroy(MgState *m);
int mg_apply(MgState *m, const cplx *f, cplx *y);

// ---- gcr.hip ---------------------------------------------------------------------------------
int op_apply_raw(Op *op, const cplx *x, cplx *y, int64_t n);
int op_residual_raw(Op *op, const cplx *x, const cplx *b, cplx *r, int64_t n);  // r = b - op(x)
int gcr_state_create(Op *A, const mgcr_gcr_param *p, int x0_mode, GcrState **out);
void gcr_state_destroy(GcrState *s);
int gcr_state_set_operator(GcrState *s, Op *A);
int gcr_state_set_x0(GcrState *s, const cplx *x0, int64_t n);
// runs the solve; `nested` = no host round trips (used when GCR is itself applied as an operator)
struct ResidualSel;  // gcr_dev.h
bool gcr_last_residual(GcrState *s, ResidualSel *out);
void gcr_set_discard_residual(GcrState *s, bool on);
void gcr_set_bnorm_source(GcrState *s, GcrState *src);   // the solve that ran on the same right-hand side just before (|b|^2 partials reused)
void gcr_set_defer_residual(GcrState *s, bool on);     // the caller forms the last step's residual on the fly (ResidualSel)
struct PendingX;  // gcr_dev.h
void gcr_set_keep_pending(GcrState *s, bool on);
bool gcr_take_pending(GcrState *s, PendingX *out);
int gcr_flush_pending(const PendingX &pd, cplx *x, int64_t n);  // nested solves whose caller never looks at the final residual
int gcr_run_from_zero(GcrState *s, const cplx *rhs, cplx *x);  // nested, x0 = 0, x's content on entry is irrelevant
int gcr_run(GcrState *s, const cplx *rhs, cplx *x, bool nested, double *hist, int hist_cap, int *n_iter,
            int *converged, bool x_known_zero = false);
int64_t gcr_fallback_count();   // solves that were repeated on the multi-kernel path after a one-launch path gave up
void gcr_state_set_use_x0(GcrState *s, bool use_x0);
int gcr_state_set_param(GcrState *s, const mgcr_gcr_param *p);
int gcr_apply_as_operator(GcrState *s, const cplx *f, cplx *y);
void gcr_last_profile(double *phase_ms_total, int *n_iter, int *fused);

// ---- gcr_small.hip ---------------------------------------------------------------------------
void gcr_small_set_limit(int64_t rows);
int64_t gcr_small_solve_count();
bool gcr_small_eligible(const Op *A, const mgcr_gcr_param &p, int storage, int64_t n);
int gcr_small_run(Op *A, const mgcr_gcr_param &p, int storage, int restart, const cplx *rhs, cplx *x, cplx *r, cplx *ar,
                  cplx *const *ps, cplx *const *aps, double *hist, int hist_cap, int *state);

}  // namespace mgcr

struct mgcr_vec_s : mgcr::Vec {};
struct mgcr_op_s : mgcr::Op {};
