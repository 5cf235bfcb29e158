// test_exsum_gpu.cpp -- stand-alone C++ caller of the drop-in API, shaped like the reference's
// tests/test.exsum.gpu.cpp / test.exdot.gpu.cpp (same argv, same variant list, same pass message), but
// with a BIT-EXACT pass criterion instead of the reference's 1e-16 relative error (test.exsum.gpu.cpp:43,:133).
// Links only against libexblas.so: no Python, no torch -- this is what a user of the reference would build.
//   usage: test_exsum_gpu log2(N) [range|stddev] [emax|mean] [n|i]
#include "blas1.hpp"
#include "common.hpp"

#include <cstdio>
#include <cstdlib>
#include <cstring>

#ifdef EXBLAS_VS_MPFR
// The reference's -DEXBLAS_VS_MPFR build sums with mpfr_add_d at 2098 bits (tests/test.exsum.gpu.cpp:20-38) and forms
// exact 128-bit products at 4196 bits for the dot (tests/test.exdot.gpu.cpp:24-46).  Those two loops live in
// oracle/libmpfr_oracle.so (oracle/mpfr_oracle.c), the checker this test links against.
extern "C" double mpfr_exsum(long n, const double *a, long inca, long offset);
extern "C" double mpfr_exdot(long n, const double *a, long inca, long offa, const double *b, long incb, long offb);
#endif

static bool same(double x, double y) { return std::memcmp(&x, &y, sizeof x) == 0 || (x != x && y != y); }

int main(int argc, char *argv[])
{
    int N = 1 << 20;
    bool lognormal = false;
    if (argc > 1) N = 1 << atoi(argv[1]);
    if (argc > 4 && argv[4][0] == 'n') lognormal = true;
    int range = 1, emax = 0;
    double mean = 1., stddev = 1.;
    if (lognormal) {
        stddev = strtod(argv[2], 0);
        mean = strtod(argv[3], 0);
    } else {
        if (argc > 2) range = atoi(argv[2]);
        if (argc > 3) emax = atoi(argv[3]);
    }
    double *a, *b;
    if (posix_memalign((void **)&a, 64, N * sizeof(double)) || posix_memalign((void **)&b, 64, N * sizeof(double))) return 2;
    srand(1);
    if (lognormal) {
        init_lognormal(N, a, mean, stddev);
        init_lognormal(N, b, mean, stddev);
    } else if (argc > 4 && argv[4][0] == 'i') {
        init_ill_cond(N, a, strtod(argv[2], 0));  // the reference parses this with atoi ("1e+50" -> 1), SURVEY section 4
        init_ill_cond(N, b, strtod(argv[2], 0));
    } else if (range == 1) {
        init_naive(N, a);
        init_naive(N, b);
    } else {
        init_fpuniform(N, a, range, emax);
        init_fpuniform(N, b, range, emax);
    }
    bool pass = true;
    // test.exsum.gpu.cpp:98-105
    const double s0 = exsum(N, a, 1, 0, 0);
    const int sum_fpe[] = {2, 3, 4, 8, 4, 6, 8};
    const bool sum_ee[] = {false, false, false, false, true, true, true};
    printf("  exsum with superacc = %.16g\n", s0);
    for (int v = 0; v < 7; ++v) {
        const double s = exsum(N, a, 1, 0, sum_fpe[v], sum_ee[v]);
        printf("  exsum with FPE%d%s and superacc = %.16g\n", sum_fpe[v], sum_ee[v] ? " early-exit" : "", s);
        if (!same(s, s0)) pass = false;
    }
    // test.exdot.gpu.cpp:111-117
    const double d0 = exdot(N, a, 1, 0, b, 1, 0, 0);
    const int dot_fpe[] = {3, 4, 8, 4, 6, 8};
    const bool dot_ee[] = {false, false, false, true, true, true};
    printf("  exdot with superacc = %.16g\n", d0);
    for (int v = 0; v < 6; ++v) {
        const double d = exdot(N, a, 1, 0, b, 1, 0, dot_fpe[v], dot_ee[v]);
        printf("  exdot with FPE%d%s and superacc = %.16g\n", dot_fpe[v], dot_ee[v] ? " early-exit" : "", d);
        if (!same(d, d0)) pass = false;
    }
#ifdef EXBLAS_VS_MPFR
    // every variant above equals s0 / d0 bit for bit; MPFR pins that value (the reference allows 1e-16 here,
    // test.exsum.gpu.cpp:118-133 -- the criterion below is bit equality)
    const double sm = mpfr_exsum(N, a, 1, 0), dm = mpfr_exdot(N, a, 1, 0, b, 1, 0);
    printf("  exsum with MPFR = %.16g\n  exdot with MPFR = %.16g\n", sm, dm);
    if (!same(s0, sm) || !same(d0, dm)) pass = false;
#endif
    // reproducibility under repetition (RNGExample.cpp:300-325 compares repeated calls with !=)
    for (int r = 0; r < 5; ++r)
        if (!same(exsum(N, a, 1, 0, 8, true), s0)) pass = false;
    printf(pass ? "TestPassed; ALL OK!\n" : "TestFailed!\n");
    free(a);
    free(b);
    return pass ? 0 : 1;
}
