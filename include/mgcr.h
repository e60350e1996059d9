/* mgcr.h — C ABI of libmgcr_hip.so: the MI355X (gfx950) implementation of the multigrid-
 * preconditioned GCR inner loop (SpMV + MG V-cycle pieces + GCR orthogonalisation).
 *
 * The reference (jing2li/MGPreconditionedGCR @ 2024_10_08) has no FFI: its seam is the C++
 * virtual `Operator<num_type>` plus `Field<num_type>` and the `*_Param` structs.  Every entry
 * point below names the reference interface it replaces (paths relative to the reference root);
 * include/mgcr/ *.h re-creates those C++ classes on top of this ABI.
 *
 * Conventions
 *  - plain C types only; complex data is interleaved (re,im) doubles ("ri"), i.e. layout-
 *    compatible with std::complex<double>[] (src/Fields.h:70, src/Operator.h:97);
 *  - every function returns an int status (MGCR_OK = 0); mgcr_last_error() gives the message of
 *    the last failure on the calling thread.  Nothing aborts (the reference asserts/exit(1)s,
 *    src/Fields.h:14,279-283 — the C++ shim turns a non-zero status back into that behaviour);
 *  - host arrays passed in are copied, never adopted; handles are opaque and destroyed explicitly;
 *  - one process drives one GPU (mgcr_init(device)); the library is thread-compatible: entry
 *    points serialise on an internal mutex, so an operator may be applied from several host
 *    threads (the reference does that inside its OpenMP set-up loops, src/MG.h:206-278);
 *  - there is NO CPU fallback: without a usable HIP device every compute entry point fails with
 *    MGCR_ERR_NO_DEVICE.
 */
#ifndef MGCR_H
#define MGCR_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MGCR_OK 0
#define MGCR_ERR_INVALID 1      /* bad argument (size mismatch, null handle, ...) */
#define MGCR_ERR_NO_DEVICE 2    /* no HIP device / mgcr_init not called */
#define MGCR_ERR_HIP 3          /* a HIP runtime call failed */
#define MGCR_ERR_ALLOC 4
#define MGCR_ERR_IO 5
#define MGCR_ERR_COMM 6
#define MGCR_ERR_UNSUPPORTED 7

typedef struct mgcr_vec_s *mgcr_vec_t; /* device-resident Field   (src/Fields.h:29-71)   */
typedef struct mgcr_op_s *mgcr_op_t;   /* device-resident Operator (src/Operator.h:16-29) */

/* ---- context ------------------------------------------------------------------------------ */
int mgcr_init(int device);
int mgcr_finalize(void);
const char *mgcr_last_error(void);
const char *mgcr_version(void);
/* name, CU count and total memory of the active device */
int mgcr_device_info(char *name, int name_cap, int *n_cu, int64_t *mem_bytes);
/* blocks until all work queued by this library has finished */
int mgcr_synchronize(void);

/* ---- Field: src/Fields.h ------------------------------------------------------------------- */
int mgcr_vec_create(int64_t n, mgcr_vec_t *out);                   /* Field(dims,ndim)   :77-80  */
int mgcr_vec_destroy(mgcr_vec_t v);                                /* ~Field             :186-190 */
int64_t mgcr_vec_size(mgcr_vec_t v);                               /* field_size         :120-123 */
int mgcr_vec_upload(mgcr_vec_t v, const double *host_ri);          /* Field(dims,ndim,init) :83-89 */
int mgcr_vec_download(mgcr_vec_t v, double *host_ri);              /* val_at             :163-168 */
int mgcr_vec_copy(mgcr_vec_t dst, mgcr_vec_t src);                 /* operator=          :256-286 */
int mgcr_vec_zero(mgcr_vec_t v);                                   /* set_zero           :137-145 */
int mgcr_vec_set_constant(mgcr_vec_t v, const double c_ri[2]);     /* set_constant       :146-151 */
/* deterministic repo-owned RHS on the grid of init_rand (:125-135) — splitmix64, not libc rand */
int mgcr_vec_fill_rhs(mgcr_vec_t v, uint64_t seed, int64_t global_offset);
int mgcr_dot(mgcr_vec_t a, mgcr_vec_t b, double out_ri[2]);        /* dot: sum conj(a)b  :216-226 */
int mgcr_norm2(mgcr_vec_t a, double *out);                         /* squarednorm        :228-235 */
/* out = a + alpha*b   (operator+ / operator- / operator* fused)                 :192-214,245-253 */
int mgcr_add_scaled(mgcr_vec_t out, mgcr_vec_t a, const double alpha_ri[2], mgcr_vec_t b);
int mgcr_axpy(const double alpha_ri[2], mgcr_vec_t x, mgcr_vec_t y); /* y += alpha*x  (+=  :288-297) */
int mgcr_scale(mgcr_vec_t v, const double alpha_ri[2]);            /* operator*          :245-253 */
int mgcr_normalise(mgcr_vec_t v);                                  /* normalise          :237-243 */
/* gamma5 :310-339  out[index with spinor 0<->2, 1<->3] = in[index]; `inner` = product of the mesh dimensions after the
 * (4-entry) spinor dimension */
int mgcr_vec_gamma5(mgcr_vec_t in, mgcr_vec_t out, int64_t inner);

/* ---- Operators: src/Operator.h, src/HierarchicalSparse.h ---------------------------------- */
/* Sparse<long> (CSR, int64 indices as the reference stores them, src/Operator.h:56-101).  The
 * device copy is an ELL slab (column-major, int32 columns) plus a CSR tail for long rows. */
int mgcr_csr_create(int64_t nrow, int64_t ncol, const int64_t *rowptr, const int64_t *col,
                    const double *val_ri, mgcr_op_t *out);
/* DiracOp = 1 - k D (src/Operator.h:104-122,569-575): a view on `csr` that applies
 * y = x - k*(D x) with the shift fused into the SpMV epilogue.  Borrows `csr` (as the reference
 * borrows its Sparse*). */
int mgcr_dirac_create(mgcr_op_t csr, const double k_ri[2], mgcr_op_t *out);
int mgcr_dirac_set_k(mgcr_op_t dirac, const double k_ri[2]);       /* set_k src/Operator.h:116 */
/* HierarchicalSparse<long,int> (block-CSR of dense bs x bs blocks, row-major inside a block,
 * src/HierarchicalSparse.h:22-48) from UNSORTED (block-row, block-col, block) triplets with the
 * constructor's semantics (:58-98): sorted by row*nbcol+col, duplicates of a pair kept and
 * summed at apply time. */
int mgcr_bcsr_create_from_triplets(int32_t nbrow, int32_t nbcol, int32_t bs, int32_t ntriplets,
                                   const int32_t *rows, const int32_t *cols, const double *blocks_ri,
                                   mgcr_op_t *out);
int mgcr_bcsr_create(int32_t nbrow, int32_t nbcol, int32_t bs, const int32_t *browptr,
                     const int32_t *bcol, const double *blocks_ri, mgcr_op_t *out);
int mgcr_op_destroy(mgcr_op_t op);
int64_t mgcr_op_dim(mgcr_op_t op);                                 /* get_dim src/Operator.h:21 */
int64_t mgcr_op_nrow(mgcr_op_t op);
int64_t mgcr_op_nnz(mgcr_op_t op);
/* y = op(x)        Operator::operator() src/Operator.h:19; Sparse :330-346; DiracOp :569-575;
 *                  HierarchicalSparse.h:101-161; GCR.h:62-68; MG.h:124-129 */
int mgcr_op_apply(mgcr_op_t op, mgcr_vec_t x, mgcr_vec_t y);
/* bytes the device layout of `op` occupies / streams per apply (for roofline accounting) */
int mgcr_op_stored_bytes(mgcr_op_t op, int64_t *matrix_bytes, int32_t *ell_width, int64_t *tail_nnz);
/* storage a Sparse ended up with: *format = 0 plain ELL slab (+ CSR tail), 1 row-pattern dictionary
 * holding column offsets and values, 2 row-pattern dictionary for the columns + value slab;
 * *n_patterns = dictionary size (0 for format 0) */
int mgcr_op_storage_format(mgcr_op_t op, int32_t *format, int32_t *n_patterns);
/* how the rows of a Sparse are dealt to threads — what decides the order in which a row's products are added, i.e. the
 * one thing in which y = A x may differ from the reference's row loop (src/Operator.h:338-341): *ell_width = W, entries
 * 0..W-1 of a row sit in the ELL slab; *lanes = L, lane l of L adds entries l, l+L, ... in order and the L sums are
 * combined by a tree (L = 1: CSR order, bit-identical to the reference); *tail_rows = rows longer than W: their remaining
 * entries are summed in CSR order by one thread (products staged in LDS) and added to the row's ELL sum — unless there are
 * more than *tail_chunk_cap of them, which one wave sums (64 lanes striding, tree); *reach = max |column - row| when the
 * operator has a row-pattern dictionary, else 0 (it decides the row -> workgroup map of the GCR kernels that embed the
 * apply); *x_window = H > 0 when the stand-alone apply stages x[tile - H, tile + 1024 + H) in LDS (banded irregular matrices:
 * >= 90 % of the slab's columns within H rows of their row; no influence on the bits).  Any pointer may be NULL.  tests/test_gpu_bitwise.py feeds these to the CPU oracle's model of the device's
 * summation order. */
int mgcr_op_ell_layout(mgcr_op_t op, int32_t *ell_width, int32_t *lanes, int64_t *tail_rows, int64_t *reach, int32_t *tail_chunk_cap,
                       int32_t *x_window);
/* Where a lean restarted GCR on this operator forms r' = r - alpha A p: *kind = 0 in a launch of its own (xr_update_kernel), 1 inside
 * the kernel that embeds the apply, latency regime (same sums, same bits), 2 inside the windowed apply of the bandwidth regime
 * (csrc/gcr_fused_xr_tile.h: |r'|^2 is then summed over that kernel's banded row map — tests/test_gpu_bitwise.py tells the oracle). */
int mgcr_op_xr_fuse_kind(mgcr_op_t op, int32_t *kind);
/* Sparse::dagger / mod_*_at (src/Operator.h:296-328,84-86) change a Sparse IN PLACE while a DiracOp, GCR or MG may hold a
 * pointer to it (src/Operator.h:117): this replaces the matrix behind an existing handle, so that every operator that
 * borrowed the handle (mgcr_dirac_create, mgcr_gcr_create) applies the new matrix.  On failure the old matrix stays.
 * Not for the row block of a distributed Sparse.  (An MG hierarchy built from the old matrix is not rebuilt — nor is the
 * reference's, src/MG.h:131-285.) */
int mgcr_csr_replace(mgcr_op_t op, int64_t nrow, int64_t ncol, const int64_t *rowptr, const int64_t *col, const double *val_ri);
/* Implementation switches (default on unless noted; each also has an environment variable read at first use).
 * They select between code paths that compute the same thing; tests use them to compare the paths.
 *   "pattern_storage" ($MGCR_PATTERNS): try the row-pattern dictionary for every Sparse of >= 2^15 rows
 *                      created while it is on;
 *   "lean_cycles"     ($MGCR_LEAN): restart-mode GCR keeps residuals instead of search directions inside
 *                      a restart cycle (same r, Ap and scalars; x differs by rounding);
 *   "fused_apply"     ($MGCR_FUSE): GCR on a Sparse / DiracOp runs the SpMV and the beta dot
 *                      products of its result as one kernel (same bits as the two kernels);
 *   "graph_replay"    ($MGCR_GRAPH, default OFF): restart cycles of systems of <= 2^18 rows are captured in a
 *                      hipGraph once and replayed (same kernels, same results; measured slower than eager
 *                      launches since the iteration shrank to 3 kernels, see gcr.hip).
 *   "resident_solver" ($MGCR_RESIDENT): a lean restarted GCR solve on a stencil-view Sparse / DiracOp of at most one row per
 *                      thread of the chip (<= 262 144 rows) runs as ONE launch with its vectors in registers
 *                      (csrc/gcr_resident.hip; same iterates, bit for bit, as the multi-kernel path).
 *   "step_build"      ($MGCR_STEPBUILD): a lean step with up to 5 stored directions on a 7-point stencil-view operator of 2^19 .. 2^21
 *                      rows runs its apply, dot products and direction build as one launch, A r staying in LDS
 *                      (csrc/gcr_stepbuild.hip; same iterates, bit for bit, as the two kernels).
 *   "halo_split"      ($MGCR_HALO_SPLIT): the stand-alone apply of a distributed Sparse whose halo travels by peer writes stores and
 *                      publishes its boundary rows, multiplies the rows that need no halo, THEN waits for the neighbours and
 *                      multiplies the boundary rows (same bits; off: the exchange completes before any row).
 *   "pw_tail"         ($MGCR_PW_TAIL): on a distributed operator whose scalars travel by peer writes, the kernel that produces a
 *                      reduction's partials (apply + dot products of a stencil row block; direction build up to 8 directions) also
 *                      folds them and sums them over the ranks in its last workgroup — no fold + exchange launch behind it (same bits).
 *   "spmv_part"       (measurement aid, default 0): 1 = a Sparse apply launches only its ELL-slab kernel, 2 = only its CSR-tail
 *                      kernel (bench.py times the two parts of the hybrid layout separately); 0 = the whole apply.
 * *previous (may be NULL) receives the old value. */
int mgcr_set_option(const char *name, int value, int *previous);
/* Counters for tests and benchmarks: "resident_solves" = GCR solves that took the one-launch path since mgcr_init,
 * "step_build_launches" = steps that ran as one apply + build launch, "small_solves" = solves that ran as one launch of one
 * workgroup (csrc/gcr_small.hip), "one_launch_fallbacks" = top-level solves that were repeated on the multi-kernel path
 * because a one-launch path gave up (foreign work on the device: its grid was not co-resident), "halo_split_exchanges" =
 * peer-write halo exchanges of distributed applies that ran split (store + publish | interior rows | wait | boundary rows),
 * "pw_tail_folds" = reductions folded and summed over the ranks inside their producing kernel. */
int mgcr_stat(const char *name, int64_t *value);

/* Self-test of the hardware behaviour the one-launch solver paths (csrc/gcr_resident.hip, gcr_stepbuild.hip) build on: inside
 * ONE launch, rows stored with `buffer_store ... sc1` by one workgroup are read correctly with `buffer_load ... sc1` by
 * workgroups of other XCDs after a fence-free {value, generation} exchange.  Runs `steps` dependent steps over one
 * workgroup per CU with the library's own primitives and bounded polls; *rows_wrong = entries that differ from what the
 * host computes (0 expected).  coherent = 0 runs the control with ordinary loads / stores (expected to FAIL: it shows the
 * test can).  tests/test_gpu_resident.py runs both on every GPU test pass. */
int mgcr_selftest_coherence(int32_t steps, int32_t coherent, int64_t *rows_wrong);

/* ---- GCR: src/GCR.h, src/SolverParam.h ---------------------------------------------------- */
typedef struct mgcr_gcr_param {
    /* GCR_Param (src/SolverParam.h:21-35) */
    int32_t truncation; /* != 0: ring of `truncation` directions, never wiped      */
    int32_t restart;    /* != 0: `restart` slots, all wiped every `restart` steps  */
    int32_t max_iter;   /* do..while: at least one iteration even for 0 (src/GCR.h:222,288) */
    double tol;         /* stop when |r|^2/|b|^2 <= tol^2 (src/GCR.h:288)          */
    int32_t verbose;    /* print "Step %d residual norm = %.10e" lines (src/GCR.h:214,271) */
    mgcr_op_t left_precond;  /* SolverParam (src/SolverParam.h:10-18); may be NULL */
    mgcr_op_t right_precond;
    /* extensions, 0 = the reference's behaviour */
    int32_t use_x0;      /* 1: r0 = b - A x0 (the reference uses r0 = b, src/GCR.h:189) */
    int32_t flexible;    /* 1: flexible right preconditioning (p = M r, true residual kept)
                            instead of the reference's literal r = M(r) (src/GCR.h:236-238) */
    int32_t check_every; /* host looks at the device-side convergence flag every this many
                            iterations (0 = library default); results do not depend on it */
    int32_t profile_spmv; /* 1: bracket the phases of every iteration with hipEvents on the library
                            stream; read the result with mgcr_gcr_last_profile (bench.py) */
} mgcr_gcr_param;

/* GCR::solve(rhs, x) (src/GCR.h:158-302).  hist[0] is the step-0 entry, hist[k] the value
 * printed at step k (sqrt(|r|^2)/|b|); at most hist_cap entries are written (hist may be NULL).
 * *n_iter = iterations performed (global_count); *converged = 0 iff n_iter == max_iter. */
int mgcr_gcr_solve(mgcr_op_t A, const mgcr_gcr_param *param, mgcr_vec_t rhs, mgcr_vec_t x,
                   double *hist, int32_t hist_cap, int32_t *n_iter, int32_t *converged);
/* Unpreconditioned solves on a single-GPU Sparse / DiracOp with at most `rows` unknowns (default
 * 1024, $MGCR_SMALL_SOLVE_ROWS; and at most 16*rows stored entries) and <= 8 stored directions run
 * as ONE launch of one workgroup (latency regime: coarsest multigrid levels); 0 disables that path.
 * Same arithmetic, dot products summed in a different fixed order. */
int mgcr_set_small_solve_rows(int64_t rows);
/* GCR as an Operator (src/GCR.h:19-50,62-68): apply(f) solves A x = f.  A may be NULL and be
 * supplied later with mgcr_gcr_set_operator (GCR(GCR_Param*) + initialise(), src/GCR.h:30-31).
 * x0_mode 0: x0 = the vector given with mgcr_gcr_set_x0 (the reference seeds x0 with
 * init_rand(2)); 1: x0 = 0.  Used as smoother / coarse solver / preconditioner, it runs without
 * host round-trips: the device-side convergence flag turns the remaining iterations into no-ops. */
int mgcr_gcr_create(mgcr_op_t A, const mgcr_gcr_param *param, int32_t x0_mode, mgcr_op_t *out);
int mgcr_gcr_set_operator(mgcr_op_t gcr, mgcr_op_t A);
/* GCR::solve(rhs, x) on a GCR object made with mgcr_gcr_create: same as mgcr_gcr_solve, but the
 * work vectors (r, Ar, the direction slots, reduction slabs) live in the object and are re-used
 * from solve to solve instead of being allocated and freed around every call.  The reference's GCR
 * reads its GCR_Param* at solve time (src/GCR.h:171-185); mgcr_gcr_set_param is that refresh. */
int mgcr_gcr_set_param(mgcr_op_t gcr, const mgcr_gcr_param *param);
int mgcr_gcr_solve_op(mgcr_op_t gcr, mgcr_vec_t rhs, mgcr_vec_t x, double *hist, int32_t hist_cap,
                      int32_t *n_iter, int32_t *converged);
int mgcr_gcr_set_x0(mgcr_op_t gcr, mgcr_vec_t x0);

/* ---- MG: src/MG.h, src/Mesh.h, src/SolverParam.h:38-59 ------------------------------------ */
typedef struct mgcr_mg_param {
    /* MG_Param::mesh + spacetime mask (src/SolverParam.h:41,47): row-major mesh of the fine
     * operator; dimensions with blocked[d] != 0 are cut into blocks of edge subblock_dim
     * (Mesh::blocking, src/Mesh.h:236-298), the others (spinor, colour) stay inside an aggregate.
     * The reference hard-wires 4 blocked dimensions; 1..4 are accepted here. */
    int32_t ndim;
    int64_t dims[8];
    int32_t blocked[8];
    int64_t subblock_dim;    /* MG_Param::subblock_dim */
    /* near-null vectors the prolongator is built from, [n_vec][N] interleaved re/im on the host.
     * (The reference computes n_eigen vectors by inverse iteration and doubles them by chirality,
     * src/MG.h:90-122,316-345; the host mirror does that and passes the 2*n_eigen vectors in.) */
    int32_t n_vec;
    const double *vecs_ri;
    int32_t n_level;         /* MG_Param::n_level = number of COARSE grids (1 = the reference's
                                two-level method; the reference stores it and never reads it) */
    mgcr_gcr_param smoother; /* GCR_Param of MG_Param::smoother_solver (max_iter = sweeps) */
    mgcr_gcr_param coarse;   /* GCR_Param of MG_Param::coarse_solver (coarsest level) */
    double damping;          /* x += damping * P x_c; the reference hard-codes 0.1 (src/MG.h:426) */
    /* extension (0 = the reference's behaviour: the coarsest system goes to `coarse`, src/MG.h:424): when the coarsest
     * level has at most this many unknowns (limit 2048) it is solved DIRECTLY — its operator is inverted once at set-up
     * (dense Gauss-Jordan with partial pivoting on the device) and every cycle applies the inverse with one mat-vec */
    int32_t coarse_direct_rows;
} mgcr_mg_param;

/* MG(Operator*, MG_Param*) + initialise (src/MG.h:131-285): aggregates, block-local prolongator
 * with per-aggregate Gram-Schmidt, Galerkin coarse operators for every level.  A must be a
 * Sparse or a DiracOp.  The result is an Operator whose apply is the corrected V-cycle described
 * in DESIGN.md (MG::operator() of the reference returns uninitialised memory, src/MG.h:124-129). */
int mgcr_mg_create(mgcr_op_t A, const mgcr_mg_param *param, mgcr_op_t *out);
/* number of operator levels, and per level: dimension, vectors per aggregate, aggregates */
int mgcr_mg_level_info(mgcr_op_t mg, int32_t level, int64_t *dim, int32_t *ne, int64_t *nagg);
/* MG::restrict / MG::expand between level and level+1 (src/MG.h:347-383) */
int mgcr_mg_restrict(mgcr_op_t mg, int32_t level, mgcr_vec_t fine, mgcr_vec_t coarse);
int mgcr_mg_expand(mgcr_op_t mg, int32_t level, mgcr_vec_t coarse, mgcr_vec_t fine);
/* borrowed handle of the level's operator (level 0 = A, level >= 1 = Galerkin m_coarse, src/MG.h:281) */
int mgcr_mg_level_op(mgcr_op_t mg, int32_t level, mgcr_op_t *out);
/* prolongator of `level` as [n][ne] block-local values plus the aggregate index of every row */
int mgcr_mg_download_prolongator(mgcr_op_t mg, int32_t level, double *pv_ri, int32_t *agg);

/* ---- multi-GPU: one process per GPU, row-block partition (new design — the reference has no
 *      distribution at all) ------------------------------------------ */
typedef struct mgcr_comm_s *mgcr_comm_t;
typedef struct mgcr_plan_s *mgcr_plan_t;
#define MGCR_RCCL_ID_BYTES 128
/* RCCL over xGMI.  Rank 0 obtains an id, the launcher broadcasts the 128 bytes (bench.py does it
 * with torch.distributed), every rank then creates its communicator.  librccl is dlopen()ed on
 * first use, so single-GPU users do not need it. */
int mgcr_rccl_unique_id(void *id128);
int mgcr_comm_create_rccl(int rank, int nranks, const void *id128, mgcr_comm_t *out);
/* Host-staged transport through caller-supplied callbacks (bring-up and tests: lets several ranks
 * share one GPU, or run the partition logic with no GPU at all).  Buffers are host memory, counts
 * are in doubles.  allreduce: in-place sum over all ranks.  exchange: post all sends / receives
 * to the listed peers and complete them. */
typedef int (*mgcr_allreduce_cb)(void *user, double *buf, int64_t count);
typedef int (*mgcr_exchange_cb)(void *user, int32_t npeers, const int32_t *peers, const double *const *send,
                                const int64_t *send_count, double *const *recv, const int64_t *recv_count);
int mgcr_comm_create_host(int rank, int nranks, mgcr_allreduce_cb allreduce, mgcr_exchange_cb exchange, void *user,
                          mgcr_comm_t *out);
int mgcr_comm_destroy(mgcr_comm_t comm);
/* in-place sum over the ranks of `count` host doubles (set-up-time scalars: global dot products of distributed
 * Fields, sizes); collective */
int mgcr_comm_allreduce_sum(mgcr_comm_t comm, double *buf, int32_t count);
/* how the per-iteration scalars are summed over the ranks: 0 host callbacks, 1 RCCL all-reduce, 2 peer-write
 * mailboxes (one kernel folds and exchanges; chosen by a self-test when the first distributed operator is created,
 * MGCR_PEER_ALLREDUCE=0 disables) */
int mgcr_comm_allreduce_kind(mgcr_comm_t comm, int32_t *kind);
/* how a distributed Sparse (mgcr_dcsr_create) exchanges its halo before an apply: 0 host callbacks, 1 RCCL send/recv
 * group, 2 peer-write (one kernel stores the boundary rows into the neighbours' receive slots and waits for theirs;
 * self-tested at creation, MGCR_PEER_HALO=0 disables) */
int mgcr_op_halo_kind(mgcr_op_t op, int32_t *kind);
/* measurement aid (bench.py, N > 1): average microseconds of one in-place device all-reduce of `count` doubles,
 * issued `reps` times back to back on the library stream (collective: every rank must call it) */
int mgcr_comm_bench_allreduce(mgcr_comm_t comm, int32_t count, int32_t reps, double *us_avg);
/* Partition plan of a row block [row0, row0 + nrow_local) of an n_global-row matrix given with
 * GLOBAL column indices: which remote x entries this rank needs (halo), from whom, what it has
 * to send to whom, and where the rows that touch no remote column sit (they can be multiplied
 * while the halo is in flight).  Pure host code: works without a GPU.  Collective over `comm`. */
int mgcr_plan_create(mgcr_comm_t comm, int64_t n_global, int64_t row0, int64_t nrow_local, const int64_t *rowptr,
                     const int64_t *col_global, mgcr_plan_t *out);
int mgcr_plan_info(mgcr_plan_t plan, int64_t *n_halo, int32_t *npeers, int64_t *interior_begin, int64_t *interior_end);
int mgcr_plan_peers(mgcr_plan_t plan, int32_t *peers, int64_t *send_counts, int64_t *recv_counts);
/* column indices in the local numbering: owned -> col - row0, halo -> nrow_local + slot */
int mgcr_plan_local_columns(mgcr_plan_t plan, int64_t *col_local);
int mgcr_plan_send_indices(mgcr_plan_t plan, int32_t peer_slot, int64_t *local_rows);
int mgcr_plan_halo_globals(mgcr_plan_t plan, int64_t *global_cols);
int mgcr_plan_destroy(mgcr_plan_t plan);
/* Row block of a distributed Sparse: Fields it applies to hold this rank's nrow_local entries;
 * apply = halo exchange (overlapped with the interior rows) + local SpMV; GCR on it all-reduces
 * its dot products (2 small all-reduces per iteration).  Collective over `comm`. */
int mgcr_dcsr_create(mgcr_comm_t comm, int64_t n_global, int64_t row0, int64_t nrow_local, const int64_t *rowptr,
                     const int64_t *col_global, const double *val_ri, mgcr_op_t *out);
/* Row block of a distributed HierarchicalSparse (reference apply: src/HierarchicalSparse.h:101-161; the reference has no
 * distribution, SURVEY.md 8(e) "Unstructured (config 5)"): block rows [brow0, brow0 + nbrow_local) of nb_global, block
 * columns GLOBAL, blocks [nblocks][bs][bs] row-major interleaved (re, im), duplicates of a (row, col) pair kept and
 * summed at apply time like the single-GPU operator.  The halo travels at block granularity (bs values per needed block
 * row of x) and the block kernel reads it in place.  Fields hold this rank's nbrow_local * bs entries.  Collective. */
int mgcr_dbcsr_create(mgcr_comm_t comm, int64_t nb_global, int64_t brow0, int32_t nbrow_local, int32_t bs, const int32_t *browptr,
                      const int64_t *bcol_global, const double *blocks_ri, mgcr_op_t *out);

/* ---- measurement helpers (bench.py) ------------------------------------------------------- */
/* runs `reps` applies back to back on the library stream, bracketed by hipEvents there;
 * returns the average milliseconds per apply */
int mgcr_bench_op_apply(mgcr_op_t op, mgcr_vec_t x, mgcr_vec_t y, int32_t reps, double *ms_avg);
/* In-loop timing of the last solve that ran with profile_spmv = 1: total milliseconds, over its
 * *n_iter iterations, of the three phases of an iteration — [0] alpha / residual update (+ right
 * preconditioner), [1] operator apply (incl. halo exchange) + beta dot products, [2] direction build —
 * from hipEvents recorded on the library stream between the phases.  *fused = 1 when phase 1 ran as
 * the single SpMV+dot kernel. */
int mgcr_gcr_last_profile(double *phase_ms_total, int32_t *n_iter, int32_t *fused);
/* opaque hipEvent-based stopwatch on the library stream */
int mgcr_timer_start(void);
int mgcr_timer_stop(double *ms);

#ifdef __cplusplus
}
#endif
#endif /* MGCR_H */
