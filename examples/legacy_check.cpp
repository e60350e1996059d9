// The reference's legacy raw-pointer dense GCR (src/GCR.h:70-156) through the drop-in headers, on the GPU, and its
// utils BLAS (host): inputs of tests/golden/legacy_dense.npz, outputs compared by tests/test_builders.py with the
// reference's.   legacy_check <dir> [utils-only]
#include <complex>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "utils.h"
#include "GCR.h"

typedef std::complex<double> cplx;
template <typename T>
static std::vector<T> rd(const std::string &p) {
    FILE *f = std::fopen(p.c_str(), "rb");
    if (!f) { std::fprintf(stderr, "cannot read %s\n", p.c_str()); std::exit(2); }
    std::fseek(f, 0, SEEK_END);
    long n = std::ftell(f);
    std::fseek(f, 0, SEEK_SET);
    std::vector<T> v((size_t)n / sizeof(T));
    if (std::fread(v.data(), sizeof(T), v.size(), f) != v.size()) std::exit(2);
    std::fclose(f);
    return v;
}
static void wr(const std::string &p, const std::vector<cplx> &v) {
    FILE *f = std::fopen(p.c_str(), "wb");
    std::fwrite(v.data(), sizeof(cplx), v.size(), f);
    std::fclose(f);
}

int main(int argc, char **argv) {
    if (argc < 2) return 1;
    const std::string d = argv[1];
    auto A = rd<cplx>(d + "/A.bin"), rhs = rd<cplx>(d + "/rhs.bin"), x0 = rd<cplx>(d + "/x0.bin"), sc = rd<cplx>(d + "/ab.bin");
    const long n = (long)rhs.size();
    {   // utils BLAS (src/utils.cpp), host only
        std::vector<cplx> z((size_t)n), y((size_t)n), Ax((size_t)n), nrm(rhs), out(2);
        vec_add(sc[0], rhs.data(), sc[1], x0.data(), z.data(), (int)n);
        vec_amult(sc[0], rhs.data(), y.data(), (int)n);
        out[0] = vec_innprod(rhs.data(), x0.data(), (int)n);
        out[1] = vec_squarednorm(rhs.data(), (int)n);
        vec_normalise(nrm.data(), (int)n);
        mat_vec(A.data(), x0.data(), Ax.data(), (int)n);
        wr(d + "/out_u_add.bin", z); wr(d + "/out_u_amult.bin", y); wr(d + "/out_u_scalars.bin", out);
        wr(d + "/out_u_normalised.bin", nrm); wr(d + "/out_u_matvec.bin", Ax);
    }
    if (argc > 2) return 0;
    struct Case { const char *tag; double tol; int max_iter, trunc; } cases[] = {{"trunc3", 1e-20, 40, 3}, {"trunc8", 1e-12, 200, 8}, {"zero", 1e6, 10, 2}};
    for (auto &c : cases) {
        GCR<long> g(A.data(), n);
        std::vector<cplx> x(x0);
        std::printf("LEGACY %s\n", c.tag);
        g.solve(rhs.data(), x.data(), c.tol, c.max_iter, c.trunc);
        wr(d + "/out_x_" + c.tag + ".bin", x);
    }
    {   // rhs = 0, x0 != 0: the absolute test keeps iterating on r0 = -A x0 (src/GCR.h:85-103)
        GCR<long> g(A.data(), n);
        std::vector<cplx> x(x0), rhs0((size_t)n, cplx(0., 0.));
        std::printf("LEGACY rhs0\n");
        g.solve(rhs0.data(), x.data(), 1e-12, 30, 4);
        wr(d + "/out_x_rhs0.bin", x);
    }
    std::printf("LEGACY end\n");
    return 0;
}
