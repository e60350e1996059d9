// BASELINE config 2 driven from C++ through the drop-in headers: the 3-D 7-point Poisson matrix assembled with
// the reference's own Sparse(rows, cols, nnz) + mod_ROW_at / mod_COL_at / mod_VAL_at interface
// (src/Operator.h:60-83), an unpreconditioned restarted GCR (src/GCR.h:158-302) and, optionally, the 3-level
// aggregation MG of config 3 as flexible right preconditioner.  Prints the iteration rate of the timed solve:
// the C++ host side reaches the HIP kernels through the same C ABI as bench.py does.
//
//   make -C examples
//   examples/build/poisson_gcr [n = 128] [iterations = 200] [restart = 5] [mg]
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "Fields.h"
#include "GCR.h"
#include "MG.h"
#include "Operator.h"

static Sparse<long> *poisson(long n) {
    const long N = n * n * n, nnz = 7 * N - 6 * n * n;
    auto *A = new Sparse<long>(N, N, nnz);
    long p = 0;
    for (long i = 0; i < n; i++)
        for (long j = 0; j < n; j++)
            for (long k = 0; k < n; k++) {
                const long row = (i * n + j) * n + k;
                A->mod_ROW_at(row, p);
                auto put = [&](bool in, long col, double v) { if (in) { A->mod_COL_at(p, col); A->mod_VAL_at(p, std::complex<double>(v, 0.)); p++; } };
                put(i > 0, row - n * n, -1.); put(j > 0, row - n, -1.); put(k > 0, row - 1, -1.);
                put(true, row, 6.);
                put(k < n - 1, row + 1, -1.); put(j < n - 1, row + n, -1.); put(i < n - 1, row + n * n, -1.);
            }
    A->mod_ROW_at(N, p);
    return A;
}

int main(int argc, char **argv) {
    const long n = argc > 1 ? std::atol(argv[1]) : 128;
    const int iters = argc > 2 ? std::atoi(argv[2]) : 200;
    const int restart = argc > 3 ? std::atoi(argv[3]) : 5;
    const bool use_mg = argc > 4;
    long dims[3] = {n, n, n};
    Mesh<long> mesh(dims, 3);
    Sparse<long> *A = poisson(n);

    GCR_Param<long> coarse(0, 10, 50, 1e-2, false, nullptr, nullptr), smooth(0, 10, 2, 1e-30, false, nullptr, nullptr);
    auto solver_coarse = new GCR<long>(&coarse);
    auto solver_smooth = new GCR<long>(&smooth);
    MG_Param<long> mgp(mesh, 2, 1, nullptr, solver_coarse, solver_smooth, 2, nullptr, nullptr);
    for (int d = 0; d < 6; d++) { mgp.spacetime[d] = d < 3; mgp.spinor[d] = false; }
    Field<long> ones(dims, 3);
    ones.set_constant(std::complex<double>(1., 0.));
    std::vector<Field<long>> nullvecs{ones};
    mgp.null_vectors = &nullvecs;   // piecewise-constant aggregation (config 3); nullptr: inverse iteration as in the reference
    MG<long> *mg = use_mg ? new MG<long>(A, &mgp) : nullptr;

    Field<long> rhs(dims, 3), x(dims, 3);
    rhs.init_rand(0);
    GCR_Param<long> prm(0, restart, iters, use_mg ? 1e-8 : 0., false, nullptr, mg);
    prm.flexible = use_mg;
    GCR<long> gcr(A, &prm);
    x.set_zero();
    gcr.solve(rhs, x);  // warm-up: allocates the solver's work vectors
    x.set_zero();
    const auto t0 = std::chrono::steady_clock::now();
    gcr.solve(rhs, x);
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    const int done = (int)gcr.history.size() - 1;
    Field<long> r = rhs - (*A)(x);
    std::printf("poisson %ld^3 restart %d%s: %d iterations in %.4f s = %.1f it/s, |r|/|b| = %.6e (recurrence %.6e)\n", n, restart,
                use_mg ? " + MG" : "", done, dt, done / dt, r.norm() / rhs.norm(), gcr.history.back());
    delete mg; delete solver_coarse; delete solver_smooth; delete A;
    return 0;
}
