// blas3.hpp -- level-3 entry point (reference: include/blas3.hpp:56 exgemm).
#ifndef BLAS3_HPP_
#define BLAS3_HPP_

#include "config.h"

/**
 * C_ij := fl(beta * C_ij) + Round(sum_l fl(alpha * op(A)_il) * op(B)_lj), row-major: the sum is exact and rounded once
 * (correctly); alpha is folded into A by a ROUNDED multiply per element first -- what the reference's GEMV kernel does
 * with x (ExGEMV.Superacc.cl:238; its GEMM kernel ignores alpha, beta, the transposes and the leading dimensions
 * altogether, ExGEMM.Superacc.cl:246-280) -- so for alpha other than 0, +-1 or a power of two the result is the
 * correctly rounded product of the rounded operand, not of alpha * A.  The update with C is an ordinary fp64
 * multiply-add (the reference: C += Round(acc), ExGEMM.Superacc.cl:280).
 * fpe < 3 superaccumulators only, otherwise floating-point expansions as in exdot (same bits either way).
 */
int exgemm(char transa, char transb, int m, int n, int k, double alpha, double *a, int lda, double *b,
           int ldb, double beta, double *c, int ldc, int fpe, bool early_exit = false);

#endif // BLAS3_HPP_
