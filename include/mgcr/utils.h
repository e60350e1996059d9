// Drop-in for the reference header of the same name (src/utils.h, src/utils.cpp:8-90): the raw-pointer complex BLAS
// helpers its legacy dense GCR (src/GCR.h:70-156) and Dense algebra are written with.  Plain host loops in the
// reference's operation order (sums in index order); pinned to the reference's output in tests/test_builders.py
// (tests/golden/legacy_dense.npz).
#pragma once
#include <cmath>
#include <complex>

#ifndef one
#define one std::complex<double>(1., 0.)
#endif
#ifndef zero
#define zero std::complex<double>(0., 0.)
#endif

// z = a x + b y
inline void vec_add(const std::complex<double> a, const std::complex<double> *x, const std::complex<double> b, const std::complex<double> *y,
                    std::complex<double> *z, int const dim) {
    for (int i = 0; i < dim; i++) z[i] = a * x[i] + b * y[i];
}
// y = a x
inline void vec_amult(const std::complex<double> a, const std::complex<double> *x, std::complex<double> *y, int const dim) {
    for (int i = 0; i < dim; i++) y[i] = a * x[i];
}
// (x, y) with the conjugate on x
inline std::complex<double> vec_innprod(const std::complex<double> *x, const std::complex<double> *y, const int dim) {
    std::complex<double> s(0., 0.);
    for (int i = 0; i < dim; i++) s += std::conj(x[i]) * y[i];
    return s;
}
inline void vec_copy(const std::complex<double> *source, std::complex<double> *destination, int count) {
    for (int i = 0; i < count; i++) destination[i] = source[i];
}
inline std::complex<double> vec_squarednorm(const std::complex<double> *x, const int dim) {
    std::complex<double> s(0., 0.);
    for (int i = 0; i < dim; i++) s += std::conj(x[i]) * x[i];
    return s;
}
inline void vec_normalise(std::complex<double> *x, const int dim) {
    const std::complex<double> f = 1. / std::sqrt(vec_squarednorm(x, dim));
    for (int i = 0; i < dim; i++) x[i] = f * x[i];
}
// y = A x, A row-major dim x dim
inline void mat_vec(const std::complex<double> *A, const std::complex<double> *x, std::complex<double> *y, const int dim) {
    for (int i = 0; i < dim; i++) {
        y[i] = std::complex<double>(0., 0.);
        for (int j = 0; j < dim; j++) y[i] += A[i * dim + j] * x[j];
    }
}
// C = A B
inline void mat_mult(const std::complex<double> *A, const std::complex<double> *B, std::complex<double> *C, const int dim) {
    for (int i = 0; i < dim; i++)
        for (int j = 0; j < dim; j++) {
            std::complex<double> s(0., 0.);
            for (int k = 0; k < dim; k++) s += A[i * dim + k] * B[k * dim + j];
            C[i * dim + j] = s;
        }
}
// B = A^+
inline void mat_dagger(const std::complex<double> *A, std::complex<double> *B, const int dim) {
    for (int i = 0; i < dim; i++)
        for (int j = 0; j < dim; j++) B[j * dim + i] = std::conj(A[i * dim + j]);
}
