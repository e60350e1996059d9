// The reference's one live experiment, k_critical_mg_precond() (src/main.cpp:834-875), written
// against the drop-in headers: same classes, same constructor arguments, same call sequence —
// only the matrix file (the shipped 4x4 sample instead of the absent 8x8 one) and its mesh differ,
// and the MG preconditioner the reference has commented out (:854-857) can be switched on.
//
//   make -C examples            (g++ -std=c++17 -Iinclude/mgcr ... -lmgcr_hip)
//   MGCR_SAMPLE_DIR=<dir with 4x4parsed.txt> examples/build/k_critical [k] [mg]
#include <iostream>
#include "Fields.h"
#include "GCR.h"
#include "utils.h"
#include "Parse.h"
#include "Operator.h"
#include "MG.h"

int main(int argc, char **argv) {
    long dims[6] = {4, 4, 4, 4, 4, 3};
    Mesh mesh(dims, 6);
    auto D = new Sparse(read_data("4x4parsed.txt"));

    GCR_Param<long> eigen(0, 10, 10, 1e-8, false, nullptr, nullptr);
    GCR_Param<long> coarse(0, 10, 50, 1e-2, false, nullptr, nullptr);
    GCR_Param<long> smooth(0, 10, 2, 1e-8, false, nullptr, nullptr);

    double const k = argc > 1 ? std::atof(argv[1]) : 0.15;
    bool const use_mg = argc > 2;
    auto Dirac = new DiracOp<long>(D, k);

    auto solver_coarse = new GCR(&coarse);
    auto solver_smooth = new GCR(&smooth);
    MG_Param<long> param(mesh, 2, 2, &eigen, solver_coarse, solver_smooth, 1, nullptr, nullptr);
    MG<long> *mg = use_mg ? new MG(Dirac, &param) : nullptr;

    GCR_Param<long> gcr_param_new(0, 5, 4000, 1e-13, true, nullptr, mg);
    gcr_param_new.flexible = use_mg;

    Field<long> rhs(dims, 6);
    rhs.init_rand(0);
    GCR gcr_precond(Dirac, &gcr_param_new);
    Field x = gcr_precond(rhs);

    // true residual of what GCR::operator() returns: x = x0 + A^-1 b with x0 = init_rand(2) (src/GCR.h:63-68,189)
    Field<long> x0(dims, 6);
    x0.init_rand(2);
    Field r = rhs - (*Dirac)(x - x0);
    printf("true relative residual of (x - x0): %.3e\n", r.norm() / rhs.norm());

    delete mg;
    delete Dirac;
    delete solver_coarse;
    delete solver_smooth;
    delete D;
    return 0;
}
