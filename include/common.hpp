// common.hpp -- constants and input generators shared by the tests
// (reference: include/common.hpp:31-161, implemented in src/common/common.cpp).
#ifndef COMMON_H
#define COMMON_H

#include <cmath>
#include <cstdio>
#include <cstdlib>

int constexpr e_bits = 1023;        // largest binary exponent of a double
int constexpr f_bits = 1023 + 52;   // ... plus the significand width
int constexpr bin_count = 39;       // limb count of the reference's GPU accumulators (kept for source compat)

double randDoubleUniform();
double randDouble(int emin, int emax, int neg_ratio);
void init_fpuniform(const int n, double *a, int range, int emax);
void init_fpuniform_matrix(const bool iscolumnwise, const int m, const int n, double *a, const int lda,
                           const int range, const int emax);
void init_fpuniform_tr_matrix(const char uplo, const char diag, const int n, double *a, const int range,
                              const int emax);
void init_lognormal(const int n, double *a, double mean, double stddev);
void init_lognormal_matrix(const bool iscolumnwise, const int m, const int n, double *a, const int lda,
                           const double mean, const double stddev);
void init_lognormal_tr_matrix(const char uplo, const char diag, const int n, double *a, const double mean,
                              const double stddev);
void init_ill_cond(const int n, double *a, double c);
void init_naive(const int n, double *a);

#endif // COMMON_H
