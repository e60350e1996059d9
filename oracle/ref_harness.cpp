// TEST INFRASTRUCTURE — not part of the product path.
//
// Harness around the REAL reference (jing2li/MGPreconditionedGCR), compiled from
// the reference's own sources where they lie under /root/reference/src (see
// oracle/Makefile, target `_ref`).  It is used to
//   (1) generate the golden vectors committed under tests/golden/ (make_golden.py)
//   (2) validate the clean-room CPU restatement in oracle/mgcr_oracle.c
//   (3) serve as the `cpu_baseline.kind = "reference"` timing leg of bench.py
//       (the built binary oracle/_ref/ref_harness travels to the GPU box; the
//       reference sources do not).
// Nothing here is copied from the reference: it only #includes its headers and
// calls its public API (private members of MG are reached via the usual
// `#define private public` test trick, scoped to that one include).
//
// Include-order constraints: std headers first (the
// reference `#define`s `one` and `zero`), utils.h before Operator.h.
#include <algorithm>
#include <cassert>
#include <chrono>
#include <complex>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <random>
#include <string>
#include <vector>
#include <unistd.h>
#include <omp.h>

#include "utils.h"
#include "Mesh.h"
#include "Fields.h"
#include "Operator.h"
#include "SolverParam.h"
#include "GCR.h"
#include "HierarchicalSparse.h"
#define private public
#include "MG.h"
#undef private
#include "Parse.h"

typedef std::complex<double> cplx;
static std::string g_out;

static void dump(const std::string &name, const void *p, size_t nbytes) {
    std::string path = g_out + "/" + name + ".bin";
    FILE *f = fopen(path.c_str(), "wb");
    if (!f) { fprintf(stderr, "cannot open %s\n", path.c_str()); exit(2); }
    fwrite(p, 1, nbytes, f);
    fclose(f);
}
static void dump_field(const std::string &name, const Field<long> &f) {
    std::vector<cplx> v(f.field_size());
    for (long i = 0; i < f.field_size(); i++) v[i] = f.val_at(i);
    dump(name, v.data(), v.size() * sizeof(cplx));
}
template <typename T>
static void dump_vec(const std::string &name, const std::vector<T> &v) {
    dump(name, v.data(), v.size() * sizeof(T));
}
static std::vector<double> read_doubles(const std::string &path) {
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) { fprintf(stderr, "cannot read %s\n", path.c_str()); exit(2); }
    fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET);
    std::vector<double> v(n / 8);
    if (fread(v.data(), 8, v.size(), f) != v.size()) exit(2);
    fclose(f);
    return v;
}

// Operator wrapper that records ||f|| of every field it is applied to.  Inside
// GCR::solve (src/GCR.h:191,242) the operator is applied once at set-up (to
// p = r0) and then once per iteration to the r whose norm that iteration prints
// (src/GCR.h:271-272), so norms[1..] / ||b|| is the full-precision history.
class SpyOp : public Operator<long> {
public:
    explicit SpyOp(Operator<long> *inner) : op(inner) { this->dim = inner->get_dim(); }
    Field<long> operator()(const Field<long> &f) override {
        norms.push_back(std::sqrt(f.squarednorm()));
        return (*op)(f);
    }
    std::complex<double> val_at(long l) const override { return op->val_at(l); }
    std::complex<double> val_at(long r, long c) const override { return op->val_at(r, c); }
    std::vector<double> norms;
    Operator<long> *op;
};

static std::vector<double> history_of(const SpyOp &spy, double bnorm) {
    std::vector<double> h;
    h.push_back(spy.norms.at(0) / bnorm);  // step 0 (r0 = b, un-preconditioned)
    for (size_t i = 1; i < spy.norms.size(); i++) h.push_back(spy.norms[i] / bnorm);
    return h;
}

// deterministic repo-owned RHS generator (same integer recipe as
// oracle/oracle.py:rhs_grid and mgcr_fill_rhs): values on the 0.001 grid of
// src/Fields.h:133, from splitmix64 instead of libc rand().
static inline uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
static void fill_rhs(Field<long> &f, uint64_t seed) {
    for (long i = 0; i < f.field_size(); i++) {
        uint64_t a = splitmix64(seed * 0x100000001B3ull + 2 * (uint64_t)i);
        uint64_t b = splitmix64(seed * 0x100000001B3ull + 2 * (uint64_t)i + 1);
        f.mod_val_at(i, cplx((double)(a % 2000) / 1000. - 1., (double)(b % 2000) / 1000. - 1.));
    }
}

// 3-D 7-point Poisson, lexicographic rows, diag 6, off-diag -1, Dirichlet
// truncation, built straight into malloc'd CSR
// arrays that the adopting constructor (src/Operator.h:64) takes over.
static Sparse<long> *make_poisson(long n) {
    long N = n * n * n, nnz = 7 * N - 6 * n * n;
    long *row = (long *)malloc(sizeof(long) * (N + 1));
    long *col = (long *)malloc(sizeof(long) * nnz);
    cplx *val = (cplx *)malloc(sizeof(cplx) * nnz);
    long p = 0;
    for (long i = 0; i < n; i++) for (long j = 0; j < n; j++) for (long k = 0; k < n; k++) {
        long r = (i * n + j) * n + k;
        row[r] = p;
        if (i > 0)     { col[p] = r - n * n; val[p++] = -1.; }
        if (j > 0)     { col[p] = r - n;     val[p++] = -1.; }
        if (k > 0)     { col[p] = r - 1;     val[p++] = -1.; }
        col[p] = r; val[p++] = 6.;
        if (k < n - 1) { col[p] = r + 1;     val[p++] = -1.; }
        if (j < n - 1) { col[p] = r + n;     val[p++] = -1.; }
        if (i < n - 1) { col[p] = r + n * n; val[p++] = -1.; }
    }
    row[N] = p;
    assert(p == nnz);
    return new Sparse<long>(N, N, row, col, val);
}

static void run_gcr_case(const std::string &tag, Operator<long> *A, GCR_Param<long> &param,
                         const Field<long> &rhs, bool zero_x0) {
    SpyOp spy(A);
    GCR<long> gcr(&spy, &param);
    Field<long> x(rhs.get_mesh());
    if (zero_x0) x.set_zero(); else x.init_rand(2);
    gcr.solve(rhs, x);
    auto h = history_of(spy, rhs.norm());
    dump_vec(tag + "_hist", h);
    dump_field(tag + "_x", x);
    fprintf(stderr, "[%s] %zu history entries, last %.10e\n", tag.c_str(), h.size(), h.back());
}

static void case_sample() {
    long dims[6] = {4, 4, 4, 4, 4, 3};
    auto D = new Sparse<long>(read_data("4x4parsed.txt"));
    cplx k(0.15, 0.);
    auto Dirac = new DiracOp<long>(D, k);

    // G1: SpMV
    Field<long> x(dims, 6);
    x.init_rand(0);
    dump_field("g1_x", x);
    dump_field("g1_Dx", (*D)(x));
    dump_field("g1_dirac_x", (*Dirac)(x));

    // G2: BLAS-1
    Field<long> a(dims, 6), b(dims, 6);
    a.init_rand(2); b.init_rand(5);
    dump_field("g2_a", a); dump_field("g2_b", b);
    cplx alpha(0.3, -0.7);
    std::vector<cplx> sc;
    sc.push_back(a.dot(b));
    sc.push_back(cplx(a.squarednorm(), b.squarednorm()));
    sc.push_back(alpha);
    dump_vec("g2_scalars", sc);
    dump_field("g2_a_plus_alpha_b", a + b * alpha);
    dump_field("g2_a_minus_alpha_b", a - b * alpha);

    // G3..G6: GCR histories (rhs = init_rand(0) with g++ evaluation order)
    Field<long> rhs(dims, 6);
    rhs.init_rand(0);
    dump_field("gcr_rhs", rhs);
    { GCR_Param<long> p(0, 5, 4000, 1e-13, true, nullptr, nullptr); run_gcr_case("g3_restart5", Dirac, p, rhs, true); }
    { GCR_Param<long> p(0, 2, 4000, 1e-13, true, nullptr, nullptr); run_gcr_case("g4_restart2", Dirac, p, rhs, true); }
    // truncated mode: tol loose enough to stop BEFORE max_iter — when a truncated run reaches
    // max_iter the reference's wipe loop (src/GCR.h:277-283) runs i < restart(=max_iter) over
    // storage_size(=truncation) slots, i.e. out of bounds (observed segfault here).
    { GCR_Param<long> p(8, 0, 300, 1e-3, true, nullptr, nullptr);   run_gcr_case("g5_trunc8", Dirac, p, rhs, true); }
    { GCR_Param<long> p(0, 0, 300, 1e-13, false, nullptr, nullptr); run_gcr_case("g6_full", Dirac, p, rhs, true); }
    // G10: smoother semantics: max_iter = 0 performs exactly one iteration (do…while)
    { GCR_Param<long> p(0, 10, 0, 1e-8, false, nullptr, nullptr);   run_gcr_case("g10_maxiter0", Dirac, p, rhs, true); }
    // x0 quirk (src/GCR.h:63-68,189): r0 = b whatever x0 is
    { GCR_Param<long> p(0, 5, 20, 1e-13, false, nullptr, nullptr);  run_gcr_case("g10_x0rand", Dirac, p, rhs, false); }
    // G11: preconditioner hooks, literal semantics r = M(r) (src/GCR.h:197-204,236-247),
    // with a well-defined M: first-order Neumann polynomial 1 + kD.
    {
        auto Mr = new DiracOp<long>(D, -k);
        GCR_Param<long> p(0, 5, 20, 1e-13, true, nullptr, Mr);  // diverges (Q5): pin 20 steps only
        run_gcr_case("g11_right_neumann", Dirac, p, rhs, true);
        GCR_Param<long> q(0, 5, 60, 1e-13, true, Mr, nullptr);
        run_gcr_case("g11_left_neumann", Dirac, q, rhs, true);
        delete Mr;
    }
    // complex k
    {
        auto Dk = new DiracOp<long>(D, cplx(0.12, 0.05));
        GCR_Param<long> p(0, 5, 40, 1e-13, false, nullptr, nullptr);
        run_gcr_case("g3b_complexk", Dk, p, rhs, true);
        delete Dk;
    }
    delete Dirac;
    delete D;
}

static void case_poisson(long n, int iters, const std::string &tag, bool timing_only,
                         int trunc = 0, int restart = 5, double tol = 1e-13) {
    auto A = make_poisson(n);
    long dims[3] = {n, n, n};
    Field<long> rhs(dims, 3);
    fill_rhs(rhs, 0);
    if (!timing_only && n <= 32) dump_field(tag + "_rhs", rhs);
    GCR_Param<long> p(trunc, restart, iters, tol, false, nullptr, nullptr);
    SpyOp spy(A);
    GCR<long> gcr(&spy, &p);
    Field<long> x(dims, 3);
    x.set_zero();
    auto t0 = std::chrono::steady_clock::now();
    gcr.solve(rhs, x);
    auto t1 = std::chrono::steady_clock::now();
    double secs = std::chrono::duration<double>(t1 - t0).count();
    auto h = history_of(spy, rhs.norm());
    if (!timing_only) {
        dump_vec(tag + "_hist", h);
        if (n <= 32) dump_field(tag + "_x", x);
    }
    // SpMV alone
    auto t2 = std::chrono::steady_clock::now();
    int reps = n >= 128 ? 3 : 20;
    double sink = 0;
    for (int r = 0; r < reps; r++) { Field<long> y = (*A)(rhs); sink += y.val_at(r).real(); }
    auto t3 = std::chrono::steady_clock::now();
    double spmv = std::chrono::duration<double>(t3 - t2).count() / reps;
    printf("{\"case\":\"poisson\",\"n\":%ld,\"iters\":%d,\"gcr_seconds\":%.6f,\"it_per_s\":%.6f,"
           "\"spmv_seconds\":%.6f,\"last_rel_res\":%.10e,\"threads\":1,\"sink\":%g}\n",
           n, iters, secs, iters / secs, spmv, h.back(), sink);
    delete A;
}

// G8: HierarchicalSparse apply on random Dense<int> blocks
static void case_hsparse() {
    const int nb = 6, bs = 4;
    std::mt19937_64 rng(1234);
    std::uniform_real_distribution<double> U(-1., 1.);
    // (row, col) list: row 0 first, every block-row non-empty, one duplicate pair,
    // rows with 1..5 blocks (ctor limits: src/HierarchicalSparse.h:73-97)
    std::vector<std::pair<int, int>> rc = {
        {0, 0}, {0, 3}, {1, 1}, {2, 0}, {2, 2}, {2, 5}, {3, 3}, {3, 4}, {3, 1}, {3, 0}, {3, 5},
        {4, 4}, {4, 4} /*duplicate*/, {4, 2}, {5, 5}, {5, 0}, {1, 4}, {0, 5}, {5, 3}, {2, 1}};
    int nt = (int)rc.size();
    auto trip = new std::pair<Operator<int> *, std::pair<int, int>>[nt];
    std::vector<cplx> blocks((size_t)nt * bs * bs);
    std::vector<int> rows(nt), cols(nt);
    for (int t = 0; t < nt; t++) {
        for (int e = 0; e < bs * bs; e++) blocks[(size_t)t * bs * bs + e] = cplx(U(rng), U(rng));
        trip[t].first = new Dense<int>(&blocks[(size_t)t * bs * bs], bs);
        trip[t].second = rc[t];
        rows[t] = rc[t].first; cols[t] = rc[t].second;
    }
    dump_vec("g8_blocks", blocks); dump_vec("g8_rows", rows); dump_vec("g8_cols", cols);
    int meta[3] = {nb, bs, nt};
    dump("g8_meta", meta, sizeof(meta));
    HierarchicalSparse<long, int> H(nb, nb, trip, nt);
    long dims[1] = {nb * bs};
    Field<long> x(dims, 1);
    fill_rhs(x, 7);
    dump_field("g8_x", x);
    dump_field("g8_y", H(x));
    // val_at(row,col) sums duplicates (src/HierarchicalSparse.h:164-178)
    std::vector<cplx> dense((size_t)nb * bs * nb * bs);
    for (long r = 0; r < nb * bs; r++) for (long c = 0; c < nb * bs; c++) dense[r * nb * bs + c] = H.val_at(r, c);
    dump_vec("g8_dense", dense);
    delete[] trip;
}

// G9: MG pieces on the 4x4 sample, block 2^4, n_eigen 2
static void case_mg() {
    long dims[6] = {4, 4, 4, 4, 4, 3};
    Mesh<long> mesh(dims, 6);
    auto D = new Sparse<long>(read_data("4x4parsed.txt"));
    auto Dirac = new DiracOp<long>(D, cplx(0.1, 0.));
    GCR_Param<long> eigen(0, 10, 10, 1e-8, false, nullptr, nullptr);
    GCR_Param<long> coarse(0, 10, 1, 1e-8, false, nullptr, nullptr);
    GCR_Param<long> smooth(0, 10, 1, 1e-8, false, nullptr, nullptr);
    auto solver_coarse = new GCR<long>(&coarse);
    auto solver_smooth = new GCR<long>(&smooth);
    const int n_eigen = 2, ne = 2 * n_eigen, sub = 2;
    MG_Param<long> param(mesh, sub, n_eigen, &eigen, solver_coarse, solver_smooth, 1, nullptr, nullptr);

    // the near-null vectors the set-up starts from (Arnoldi, src/MG.h:90-122) and their
    // chirality doubling (src/MG.h:316-345) — dumped so that the restatement of the
    // deterministic part of the set-up can be pinned without re-running the iteration
    {
        auto ev = new Field<long>[n_eigen];
        Arnoldi<long> ar(&eigen, n_eigen);
        ar.solve(Dirac, ev, mesh);
        for (int i = 0; i < n_eigen; i++) dump_field("g9_eigvec" + std::to_string(i), ev[i]);
        dump_field("g9_gamma5_eigvec0", ev[0].gamma5(4));
        delete[] ev;
    }

    auto mg = new MG<long>(Dirac, &param);
    long nblocks = param.mesh.get_nblocks();
    long bsz = param.mesh.get_block_size();
    long meta[6] = {nblocks, bsz, ne, sub, 0, 0};
    dump("g9_meta", meta, sizeof(meta));
    // block map (src/Mesh.h:236-298)
    std::vector<long> bmap((size_t)nblocks * bsz);
    for (long b = 0; b < nblocks; b++) for (long o = 0; o < bsz; o++) bmap[b * bsz + o] = param.mesh.get_block_map(b)[o];
    dump_vec("g9_block_map", bmap);
    // prolongator columns as full-length fields [block][i][N]
    long N = Dirac->get_dim();
    std::vector<cplx> P((size_t)nblocks * ne * N);
    for (long b = 0; b < nblocks; b++) for (int i = 0; i < ne; i++)
        for (long j = 0; j < N; j++) P[((size_t)b * ne + i) * N + j] = mg->prolongator[b][i].val_at(j);
    dump_vec("g9_P", P);
    // restrict / expand / coarse apply
    Field<long> v(dims, 6);
    v.init_rand(42);
    dump_field("g9_v", v);
    Field<long> Rv = mg->restrict(v);
    dump_field("g9_Rv", Rv);
    Field<long> PRv = mg->expand(Rv);
    dump_field("g9_PRv", PRv);
    Field<long> AcRv = (*mg->m_coarse)(Rv);
    dump_field("g9_AcRv", AcRv);
    // coarse operator as dense (nb*ne)^2 via val_at(row,col)
    long nc = nblocks * ne;
    std::vector<cplx> Ac((size_t)nc * nc);
    for (long r = 0; r < nc; r++) for (long c = 0; c < nc; c++) Ac[r * nc + c] = mg->m_coarse->val_at(r, c);
    dump_vec("g9_Ac_dense", Ac);
    // identities of test_MG_property (src/main.cpp:899-909)
    Field<long> i1 = mg->restrict(v), i2 = mg->expand(i1), i3 = mg->restrict(i2);
    Field<long> i11 = mg->restrict(i2), i22 = mg->expand(i11);
    double idn[2] = {(i2 - i22).norm(), (i3 - i1).norm()};
    dump("g9_identities", idn, sizeof(idn));
    fprintf(stderr, "[mg] RT-Id %.3e  TRTR-TR %.3e\n", idn[0], idn[1]);
    delete mg; delete solver_coarse; delete solver_smooth; delete Dirac; delete D;
}


// ---- round 2 pins: the input builders, Arnoldi, and the well-defined part of the operator algebra -------------------

// G12: Sparse from shuffled triplets with duplicated (row, col) pairs (src/Operator.h:250-294).  The constructor
// assumes that the first sorted triplet sits in row 0 and that no row is empty; duplicates come in PAIRS only
// (std::sort is not stable: with three or more equal keys the order of the additions, hence the rounding, would be
// unspecified).
static void case_builders(const std::string &mtx_path) {
    const long rows = 9, cols = 7;
    std::mt19937_64 rng(99);
    std::uniform_real_distribution<double> U(-1., 1.);
    typedef std::pair<cplx, std::pair<long, long>> Trip;
    std::vector<Trip> t;
    for (long r = 0; r < rows; r++) {
        int len = 1 + (int)(rng() % 4);
        std::vector<long> cs;
        while ((int)cs.size() < len) { long c = (long)(rng() % cols); if (std::find(cs.begin(), cs.end(), c) == cs.end()) cs.push_back(c); }
        for (long c : cs) t.push_back(Trip(cplx(U(rng), U(rng)), {r, c}));
        if (r % 3 == 1) t.push_back(Trip(cplx(U(rng), U(rng)), {r, cs[0]}));   // one duplicated pair in this row
    }
    std::shuffle(t.begin(), t.end(), rng);
    std::vector<long> tr, tc; std::vector<cplx> tv;
    for (auto &e : t) { tr.push_back(e.second.first); tc.push_back(e.second.second); tv.push_back(e.first); }
    dump_vec("g12_trip_rows", tr); dump_vec("g12_trip_cols", tc); dump_vec("g12_trip_vals", tv);
    Sparse<long> S(rows, cols, t.data(), (long)t.size());
    long nnz = S.get_nnz();
    std::vector<long> R(rows + 1), C(nnz); std::vector<cplx> V(nnz);
    for (long r = 0; r <= rows; r++) R[r] = S.get_ROW(r);
    for (long l = 0; l < nnz; l++) { C[l] = S.get_COL(l); V[l] = S.val_at(l); }
    long meta[3] = {rows, cols, nnz};
    dump("g12_meta", meta, sizeof(meta));
    dump_vec("g12_ROW", R); dump_vec("g12_COL", C); dump_vec("g12_VAL", V);

    // G15 (algebra on the same matrix): dagger (src/Operator.h:296-328) and * scalar (:535-544)
    {
        Sparse<long> T(S);
        T.dagger();
        long tn = T.get_nnz();
        std::vector<long> TR(T.get_nrow() + 1), TC(tn); std::vector<cplx> TV(tn);
        for (long r = 0; r <= T.get_nrow(); r++) TR[r] = T.get_ROW(r);
        for (long l = 0; l < tn; l++) { TC[l] = T.get_COL(l); TV[l] = T.val_at(l); }
        long tm[3] = {T.get_nrow(), T.get_dim(), tn};
        dump("g15_dagger_meta", tm, sizeof(tm));
        dump_vec("g15_dagger_ROW", TR); dump_vec("g15_dagger_COL", TC); dump_vec("g15_dagger_VAL", TV);
        cplx a(0.3, -1.1);
        Sparse<long> M = S * a;
        std::vector<cplx> MV(nnz);
        for (long l = 0; l < nnz; l++) MV[l] = M.val_at(l);
        dump_vec("g15_scaled_VAL", MV);
        cplx av[1] = {a};
        dump("g15_scalar", av, sizeof(av));
    }
    // Dense algebra (src/Operator.h:139-190).  operator+ passes `d`, not d*d, to vec_add (:144): only the first d
    // entries (the first row) of A + B are computed, the rest of the result is uninitialised memory — the golden
    // holds the first row only.
    {
        const long d = 5;
        std::vector<cplx> A(d * d), B(d * d);
        for (auto &v : A) v = cplx(U(rng), U(rng));
        for (auto &v : B) v = cplx(U(rng), U(rng));
        dump_vec("g15_dense_A", A); dump_vec("g15_dense_B", B);
        Dense<long> DA(A.data(), d), DB(B.data(), d);
        Dense<long> P = DA * DB, H = DA.dagger(), Sum = DA + DB;
        std::vector<cplx> vp(d * d), vh(d * d), vs(d);
        for (long e = 0; e < d * d; e++) { vp[e] = P.val_at(e); vh[e] = H.val_at(e); }
        for (long e = 0; e < d; e++) vs[e] = Sum.val_at(e);
        dump_vec("g15_dense_AB", vp); dump_vec("g15_dense_Adag", vh); dump_vec("g15_dense_sum_row0", vs);
    }

    // G13: parse_data (src/Parse.cpp:9-61): MatrixMarket "row col re im" (1-based, % comments) -> the text CSR format
    // of read_data.  It writes "../../data/sample_matrix/parsed.txt" relative to the cwd: make_golden.py runs this case
    // from a scratch directory two levels below a scratch root that holds data/sample_matrix/, never inside the reference.
    if (!mtx_path.empty()) parse_data(mtx_path);
}

// G14: Arnoldi (src/MG.h:90-122) on the 4x4 sample, k = 0.1, GCR_Param(0,10,10,1e-8): the start vector init_rand(9),
// the first vector — gcr.solve(b, b) ALIASES rhs and x, i.e. x0 = b with r0 = b (src/GCR.h:189): b <- normalise(b + GCR_10(b)),
// ten times — and the second vector in its only well-defined form: the reference solves into `Field tmp(mesh)`, which is
// malloc'ed and never initialised (src/MG.h:110, src/Fields.h:97-101), so its own second vector depends on heap contents;
// here tmp is zeroed first, everything else is the reference's code.
static void case_arnoldi() {
    long dims[6] = {4, 4, 4, 4, 4, 3};
    Mesh<long> mesh(dims, 6);
    auto D = new Sparse<long>(read_data("4x4parsed.txt"));
    auto Dirac = new DiracOp<long>(D, cplx(0.1, 0.));
    GCR_Param<long> eigen(0, 10, 10, 1e-8, false, nullptr, nullptr);
    Field<long> b(mesh);
    b.init_rand(9);
    dump_field("g14_start", b);
    GCR<long> gcr(Dirac, &eigen);
    for (int i = 0; i < 10; i++) { gcr.solve(b, b); b.normalise(); }
    dump_field("g14_vec0", b);
    {   // the reference's own Arnoldi must give the same first vector, bit for bit
        auto ev = new Field<long>[1];
        Arnoldi<long> ar(&eigen, 1);
        ar.solve(Dirac, ev, mesh);
        for (long i = 0; i < b.field_size(); i++) assert(ev[0].val_at(i) == b.val_at(i));
        delete[] ev;
    }
    Field<long> tmp(mesh);
    tmp.set_zero();
    gcr.solve(b, tmp);
    cplx h = b.dot(tmp);
    tmp -= b * h;
    tmp.normalise();
    dump_field("g14_vec1_x0zero", tmp);
    delete Dirac; delete D;
}

// G16: the legacy raw-pointer dense GCR (src/GCR.h:70-156) and the utils BLAS it is written with (src/utils.cpp:8-90):
// r0 = rhs - A x0 (x0 honoured), truncated directions, stops when |r|^2 <= tol (absolute, tested BEFORE every step) or
// after max_iter steps.  It keeps no history: the golden holds the final x at full precision and the printed norms
// (`Step %d residual norm = %.10e`, captured from stdout by make_golden.py).
static void case_legacy() {
    const long d = 24;
    std::mt19937_64 rng(321);
    std::uniform_real_distribution<double> U(-1., 1.);
    std::vector<cplx> A(d * d, cplx(0., 0.)), rhs(d), x(d);
    for (long i = 0; i < d; i++) {
        A[i * d + i] = cplx(4. + 0.1 * U(rng), 0.3 * U(rng));
        if (i > 0) A[i * d + i - 1] = cplx(-1., 0.2 * U(rng));
        if (i + 1 < d) A[i * d + i + 1] = cplx(-1. + 0.1 * U(rng), 0.);
        if (i + 5 < d) A[i * d + i + 5] = cplx(0.25 * U(rng), 0.25 * U(rng));
        rhs[i] = cplx(U(rng), U(rng));
        x[i] = cplx(0.5 * U(rng), 0.5 * U(rng));
    }
    dump_vec("g16_A", A); dump_vec("g16_rhs", rhs); dump_vec("g16_x0", x);
    {
        GCR<long> g(A.data(), d);
        std::vector<cplx> xs(x);
        printf("LEGACY trunc3\n");
        g.solve(rhs.data(), xs.data(), 1e-20, 40, 3);
        dump_vec("g16_x_trunc3", xs);
    }
    {
        GCR<long> g(A.data(), d);
        std::vector<cplx> xs(x);
        printf("LEGACY trunc8_tol\n");
        g.solve(rhs.data(), xs.data(), 1e-12, 200, 8);       // stops on the tolerance
        dump_vec("g16_x_trunc8", xs);
    }
    {   // |r0|^2 below the tolerance: zero steps, x untouched
        GCR<long> g(A.data(), d);
        std::vector<cplx> xs(x);
        printf("LEGACY zero_steps\n");
        g.solve(rhs.data(), xs.data(), 1e6, 10, 2);
        dump_vec("g16_x_zero", xs);
    }
    {   // rhs = 0, x0 != 0: the absolute test keeps iterating on r0 = -A x0 and x is driven towards 0
        GCR<long> g(A.data(), d);
        std::vector<cplx> xs(x), rhs0(d, cplx(0., 0.));
        printf("LEGACY rhs0\n");
        g.solve(rhs0.data(), xs.data(), 1e-12, 30, 4);
        dump_vec("g16_x_rhs0", xs);
    }
    printf("LEGACY end\n");
    // utils BLAS on the same data
    std::vector<cplx> z(d), y(d), Ax(d), nrm(rhs), sc(4);
    cplx a(0.3, -0.7), b(-1.2, 0.4);
    vec_add(a, rhs.data(), b, x.data(), z.data(), (int)d);
    vec_amult(a, rhs.data(), y.data(), (int)d);
    sc[0] = vec_innprod(rhs.data(), x.data(), (int)d);
    sc[1] = vec_squarednorm(rhs.data(), (int)d);
    vec_normalise(nrm.data(), (int)d);
    mat_vec(A.data(), x.data(), Ax.data(), (int)d);
    sc[2] = a; sc[3] = b;
    dump_vec("g16_u_add", z); dump_vec("g16_u_amult", y); dump_vec("g16_u_scalars", sc); dump_vec("g16_u_normalised", nrm);
    dump_vec("g16_u_matvec", Ax);
}

int main(int argc, char **argv) {
    if (argc < 3) {
        fprintf(stderr, "usage: %s <outdir> sample|hsparse|mg|arnoldi|legacy|builders [file.mtx]|poisson <n> <iters> <tag> [trunc restart tol]|bench <n> <iters>\n", argv[0]);
        return 1;
    }
    char absout[4096];
    if (!realpath(argv[1], absout)) { fprintf(stderr, "bad outdir\n"); return 1; }
    g_out = absout;
    std::string c = argv[2];
    // read_data() opens "../../data/sample_matrix/<name>" relative to the cwd (src/Parse.cpp:66)
    if (c == "sample" || c == "mg" || c == "arnoldi") {
        const char *root = getenv("MGCR_REFERENCE_ROOT");
        std::string d = std::string(root ? root : "/root/reference") + "/data/sample_matrix";
        if (chdir(d.c_str()) != 0) { fprintf(stderr, "cannot chdir to %s\n", d.c_str()); return 1; }
    }
    if (c == "sample") case_sample();
    else if (c == "hsparse") case_hsparse();
    else if (c == "mg") case_mg();
    else if (c == "arnoldi") case_arnoldi();
    else if (c == "legacy") case_legacy();
    else if (c == "builders") case_builders(argc >= 4 ? argv[3] : "");
    else if (c == "poisson" && argc >= 9) case_poisson(atol(argv[3]), atoi(argv[4]), argv[5], false, atoi(argv[6]), atoi(argv[7]), atof(argv[8]));
    else if (c == "poisson" && argc >= 6) case_poisson(atol(argv[3]), atoi(argv[4]), argv[5], false);
    else if (c == "bench" && argc >= 5) case_poisson(atol(argv[3]), atoi(argv[4]), "bench", true);
    else { fprintf(stderr, "unknown case\n"); return 1; }
    return 0;
}
