// The reference ships raw-pointer BLAS helpers here (src/utils.h) that only its legacy dense GCR and
// commented tests use (out of scope). Kept as an empty include target.
#pragma once
#include <complex>
