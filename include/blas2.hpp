// blas2.hpp -- level-2 entry points (reference: include/blas2.hpp:57 extrsv, :95 exgemv).
#ifndef BLAS2_HPP_
#define BLAS2_HPP_

#include "config.h"

/** Declared for source compatibility; the triangular solve is not part of this backend (returns -1). */
int extrsv(const char uplo, const char transa, const char diag, const int n, double *a, const int lda,
           const int offseta, double *x, const int incx, const int offsetx, const int fpe,
           const bool early_exit = false);

/**
 * y := round(alpha * op(A) * x (+) beta * y), every y_i the correctly rounded exact row sum.
 * A is column-major with leading dimension lda.  fpe == 0 superaccumulators only, fpe == 1 plain
 * (non-reproducible) DGEMV, otherwise floating-point expansions as in exsum.
 */
int exgemv(const char transa, const int m, const int n, const double alpha, double *a, const int lda,
           const int offseta, double *x, const int incx, const int offsetx, const double beta, double *y,
           const int incy, const int offsety, const int fpe, const bool early_exit = false);

#endif // BLAS2_HPP_
