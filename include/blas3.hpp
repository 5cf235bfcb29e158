// blas3.hpp -- level-3 entry point (reference: include/blas3.hpp:56 exgemm).
#ifndef BLAS3_HPP_
#define BLAS3_HPP_

#include "config.h"

/**
 * C := beta*C + round(alpha * op(A) * op(B)), row-major, every dot product correctly rounded.
 * fpe < 3 superaccumulators only, otherwise floating-point expansions as in exdot.
 */
int exgemm(char transa, char transb, int m, int n, int k, double alpha, double *a, int lda, double *b,
           int ldb, double beta, double *c, int ldc, int fpe, bool early_exit = false);

#endif // BLAS3_HPP_
