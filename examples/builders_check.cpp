// Host-side checks of the C++ drop-in interface (no GPU is touched: the device copy of an operator is built on first
// apply): the triplet constructor of Sparse, dagger, * scalar, the Dense algebra and parse_data, fed with the inputs
// of tests/golden/builders.npz and compared by tests/test_builders.py with the reference's outputs.
//   builders_check <dir>    reads <dir>/trip_{rows,cols,vals}.bin, meta.bin, scalar.bin, dense_{A,B}.bin, in.mtx
//                           writes <dir>/out_*.bin and ($MGCR_SAMPLE_DIR = <dir>) <dir>/parsed.txt
#include <complex>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "GCR.h"
#include "Operator.h"
#include "Parse.h"

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
template <typename T>
static void wr(const std::string &p, const std::vector<T> &v) {
    FILE *f = std::fopen(p.c_str(), "wb");
    std::fwrite(v.data(), sizeof(T), v.size(), f);
    std::fclose(f);
}
static void dump_csr(const std::string &d, const std::string &tag, const Sparse<long> &S) {
    std::vector<long> R((size_t)S.get_nrow() + 1), C((size_t)S.get_nnz()), M = {S.get_nrow(), S.get_dim(), S.get_nnz()};
    std::vector<cplx> V((size_t)S.get_nnz());
    for (long r = 0; r <= S.get_nrow(); r++) R[(size_t)r] = S.get_ROW(r);
    for (long l = 0; l < S.get_nnz(); l++) { C[(size_t)l] = S.get_COL(l); V[(size_t)l] = S.val_at(l); }
    wr(d + "/out_" + tag + "_meta.bin", M); wr(d + "/out_" + tag + "_ROW.bin", R);
    wr(d + "/out_" + tag + "_COL.bin", C); wr(d + "/out_" + tag + "_VAL.bin", V);
}

int main(int argc, char **argv) {
    if (argc < 2) return 1;
    const std::string d = argv[1];
    auto tr = rd<long>(d + "/trip_rows.bin"), tc = rd<long>(d + "/trip_cols.bin"), meta = rd<long>(d + "/meta.bin");
    auto tv = rd<cplx>(d + "/trip_vals.bin");
    std::vector<std::pair<cplx, std::pair<long, long>>> t(tr.size());
    for (size_t i = 0; i < tr.size(); i++) t[i] = {tv[i], {tr[i], tc[i]}};
    Sparse<long> S(meta[0], meta[1], t.data(), (long)t.size());       // src/Operator.h:250-294
    dump_csr(d, "csr", S);
    Sparse<long> T(S);
    T.dagger();                                                       // :296-328
    dump_csr(d, "dagger", T);
    Sparse<long> M = S * rd<cplx>(d + "/scalar.bin")[0];              // :535-544
    dump_csr(d, "scaled", M);
    auto A = rd<cplx>(d + "/dense_A.bin"), B = rd<cplx>(d + "/dense_B.bin");
    long dim = 1;
    while (dim * dim < (long)A.size()) dim++;
    Dense<long> DA(A.data(), dim), DB(B.data(), dim);
    Dense<long> P = DA * DB, H = DA.dagger(), Sm = DA + DB;          // :139-190
    std::vector<cplx> vp(A.size()), vh(A.size()), vs(A.size());
    for (long e = 0; e < dim * dim; e++) { vp[(size_t)e] = P.val_at(e); vh[(size_t)e] = H.val_at(e); vs[(size_t)e] = Sm.val_at(e); }
    wr(d + "/out_dense_AB.bin", vp); wr(d + "/out_dense_Adag.bin", vh); wr(d + "/out_dense_sum.bin", vs);
    parse_data(d + "/in.mtx");                                        // src/Parse.cpp:9-61 -> $MGCR_SAMPLE_DIR/parsed.txt
    return 0;
}
