// blas2.hpp -- level-2 entry points (reference: include/blas2.hpp:57 extrsv, :95 exgemv).
#ifndef BLAS2_HPP_
#define BLAS2_HPP_

#include "config.h"

/**
 * Solves A*x = b or A**T*x = b for a triangular column-major A; x holds b on entry and the solution on return,
 * every x_i = fl(Round(b_i - sum_j A(i,j)*x_j) / A(i,i)) with the sum exact.  uplo 'L'/'U', transa 'N'/'T',
 * diag 'N'/'U'.  fpe == 0 superaccumulators only, fpe == 1 plain DTRSV, 2..8 expansions (early_exit buckets 4/6/8).
 * Returns 0; fpe >= 9 (the reference's iterative-refinement kernels, which it does not ship, ExTRSV.cpp:91-120)
 * prints a message, leaves x untouched and returns -1.
 */
int extrsv(const char uplo, const char transa, const char diag, const int n, double *a, const int lda,
           const int offseta, double *x, const int incx, const int offsetx, const int fpe,
           const bool early_exit = false);

/**
 * y_i := Round(sum_k op(A)_ik * fl(alpha * x_k)  (+)  beta * y_i): the sum -- including beta * y_i, exactly (beta = 1: y_i
 * itself; otherwise the error-free product) -- is exact and rounded once, correctly.  alpha is folded into x by a
 * ROUNDED multiply per element first, as the reference's kernel does (ExGEMV.Superacc.cl:238): for alpha other than
 * 0, +-1 or a power of two the result is the correctly rounded product with the rounded vector, not with alpha * x.
 * A is column-major with leading dimension lda.  fpe == 0 superaccumulators only, fpe == 1 plain
 * (non-reproducible) DGEMV, otherwise floating-point expansions as in exsum.
 */
int exgemv(const char transa, const int m, const int n, const double alpha, double *a, const int lda,
           const int offseta, double *x, const int incx, const int offsetx, const double beta, double *y,
           const int incy, const int offsety, const int fpe, const bool early_exit = false);

#endif // BLAS2_HPP_
