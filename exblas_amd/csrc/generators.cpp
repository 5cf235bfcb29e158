// generators.cpp -- host-side input generators exported by libexblas.so for callers that use
// include/common.hpp.  Same distributions and the same glibc rand() draw order as the reference's
// src/common/common.cpp (so a test seeded the same way sees the same vectors); written from its
// description in SURVEY section 2 row 2, not copied.
#include "../../include/common.hpp"

#include <random>

namespace {
inline double unit_rand() { return static_cast<double>(rand()) / static_cast<double>(RAND_MAX); }

template <typename F>
void fill_matrix(bool colmajor, int m, int n, double *a, int lda, F &&next)
{
    const int outer = colmajor ? n : m, inner = colmajor ? m : n;
    for (int o = 0; o < outer; ++o)
        for (int i = 0; i < inner; ++i) a[static_cast<long>(o) * lda + i] = next();
}

template <typename F>
void fill_triangle(char uplo, char diag, int n, double *a, bool rowmajor_index, F &&next)
{
    // walks the triangle in the reference's order (common.cpp:47-64 / :91-111)
    auto at = [&](int i, int j) -> double & {
        return rowmajor_index ? a[static_cast<long>(i) * n + j] : a[static_cast<long>(j) * n + i];
    };
    if (uplo == 'U') {
        for (int i = n - 1; i >= 0; --i)
            for (int j = i; j < n; ++j) at(i, j) = (diag == 'U' && i == j) ? 1.0 : next();
    } else {
        for (int i = 0; i < n; ++i)
            for (int j = 0; j <= i; ++j) at(i, j) = (diag == 'U' && i == j) ? 1.0 : next();
    }
}
}  // namespace

double randDoubleUniform() { return static_cast<double>(rand() - RAND_MAX / 4) * 12345.678901234; }

double randDouble(int emin, int emax, int neg_ratio)
{
    // mantissa in [1, 2.01), then a uniform exponent, then (optionally) a sign: three rand() draws
    double mant = static_cast<double>(rand()) / static_cast<double>(RAND_MAX * .99) + 1.;
    const int e = rand() % (emax - emin) + emin;
    if (neg_ratio > 1 && rand() % neg_ratio == 0) mant = -mant;
    return ldexp(mant, e);
}

void init_fpuniform(const int n, double *a, int range, int emax)
{
    for (int i = 0; i < n; ++i) a[i] = randDouble(emax - range, emax, 1);
}

void init_fpuniform_matrix(const bool iscolumnwise, const int m, const int n, double *a, const int lda,
                           const int range, const int emax)
{
    (void)emax;  // the reference spreads the exponents over [0, range) regardless of emax
    fill_matrix(iscolumnwise, m, n, a, lda, [&] { return randDouble(0, range, 1); });
}

void init_fpuniform_tr_matrix(const char uplo, const char diag, const int n, double *a, const int range,
                              const int emax)
{
    fill_triangle(uplo, diag, n, a, false, [&] { return randDouble(emax - range, emax, 1); });
}

void init_lognormal(const int n, double *a, double mean, double stddev)
{
    std::random_device rd;
    std::default_random_engine gen(rd());
    std::lognormal_distribution<> dist(mean, stddev);
    for (int i = 0; i < n; ++i) a[i] = dist(gen);
}

void init_lognormal_matrix(const bool iscolumnwise, const int m, const int n, double *a, const int lda,
                           const double mean, const double stddev)
{
    (void)mean; (void)stddev;  // upstream writes all-ones here (common.cpp:83,:87); kept for parity
    fill_matrix(iscolumnwise, m, n, a, lda, [] { return 1.0; });
}

void init_lognormal_tr_matrix(const char uplo, const char diag, const int n, double *a, const double mean,
                              const double stddev)
{
    std::random_device rd;
    std::default_random_engine gen(rd());
    std::lognormal_distribution<> dist(mean, stddev);
    fill_triangle(uplo, diag, n, a, true, [&] { return dist(gen); });
}

void init_ill_cond(const int n, double *a, double c)
{
    // Ogita-Rump-Oishi Alg. 6.1 as upstream uses it (common.cpp:113-145): first half with random
    // exponents in [0, log2(c)/2], second half with exponents ramping 0 -> log2(c)/2
    const int half = static_cast<int>(round(n / 2));
    const double b = log2(c);
    double *e = static_cast<double *>(malloc(sizeof(double) * (n > 0 ? n : 1)));
    for (int i = 0; i < n; ++i) a[i] = 0.0;
    for (int i = 0; i < half; ++i) e[i] = round(unit_rand() * b / 2);
    if (n > 0) {
        e[0] = round(b / 2) + 1.;
        e[n - 1] = 0;
    }
    for (int i = 0; i < half; ++i) a[i] = (2. * unit_rand() - 1.) * pow(2., e[i]);
    const double step = (b / 2) / (n - half);
    for (int i = half; i < n; ++i) {
        const double u = unit_rand();
        e[i] = step * (i - half);
        a[i] = (2. * u - 1.) * pow(2., e[i]);
    }
    free(e);
}

void init_naive(const int n, double *a)
{
    for (int i = 0; i < n; ++i) a[i] = 1.1;
}
