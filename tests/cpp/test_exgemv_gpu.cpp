// test_exgemv_gpu.cpp -- C++ caller shaped like the reference's tests/test.exgemv.gpu.cpp and test.exgemm.gpu.cpp:
// same argument conventions (column-major gemv with alpha = beta = 1, :160; row-major gemm 256^3, test.exgemm.gpu.cpp:183),
// every variant the reference's tests run, compared BITWISE against the superaccumulator-only result
// (the reference accepts 1e-15 norm-wise, test.exgemv.gpu.cpp:159) and norm-wise against a plain fp64 evaluation.
//   usage: test_exgemv_gpu [m] [n]
#include "blas2.hpp"
#include "blas3.hpp"
#include "common.hpp"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#ifdef EXBLAS_VS_MPFR
// the MPFR oracles of the reference's -DEXBLAS_VS_MPFR tests (tests/test.exgemv.gpu.cpp:35-103,
// tests/test.exgemm.gpu.cpp:53-125), kept in oracle/libmpfr_oracle.so
extern "C" void mpfr_exgemv(char trans, int m, int n, double alpha, const double *a, int lda, const double *x, int incx,
                            double beta, const double *y, int incy, double *y_out);
extern "C" void mpfr_exgemm_dots(int m, int n, int k, const double *a, int lda, const double *b, int ldb, double *d_out,
                                 int ldd);
#endif

static bool same(const std::vector<double> &x, const std::vector<double> &y)
{
    return std::memcmp(x.data(), y.data(), x.size() * sizeof(double)) == 0;
}

int main(int argc, char *argv[])
{
    const int m = argc > 1 ? atoi(argv[1]) : 512, n = argc > 2 ? atoi(argv[2]) : 384;
    bool pass = true;
    srand(3);
    std::vector<double> a((size_t)m * n), x(std::max(m, n)), y0(std::max(m, n));
    init_fpuniform_matrix(true, m, n, a.data(), m, 10, 0);
    init_fpuniform(std::max(m, n), x.data(), 10, 0);
    init_fpuniform(std::max(m, n), y0.data(), 10, 0);
    for (char trans : {'N', 'T'}) {
        const int rows = trans == 'T' ? n : m, inner = trans == 'T' ? m : n;
        std::vector<double> ref(y0.begin(), y0.begin() + rows);
        exgemv(trans, m, n, 1.0, a.data(), m, 0, x.data(), 1, 0, 1.0, ref.data(), 1, 0, 0);   // superacc only
        const int fpe[] = {3, 4, 8, 4, 6, 8};
        const bool ee[] = {false, false, false, true, true, true};
        for (int v = 0; v < 6; ++v) {
            std::vector<double> y(y0.begin(), y0.begin() + rows);
            exgemv(trans, m, n, 1.0, a.data(), m, 0, x.data(), 1, 0, 1.0, y.data(), 1, 0, fpe[v], ee[v]);
            if (!same(y, ref)) { pass = false; printf("exgemv %c fpe%d%s differs\n", trans, fpe[v], ee[v] ? "ee" : ""); }
        }
#ifdef EXBLAS_VS_MPFR
        {
            std::vector<double> ym(rows);
            mpfr_exgemv(trans, m, n, 1.0, a.data(), m, x.data(), 1, 1.0, y0.data(), 1, ym.data());
            if (!same(ym, ref)) { pass = false; printf("exgemv %c differs from MPFR\n", trans); }
            else printf("  exgemv %c == MPFR bit for bit (%d outputs)\n", trans, rows);
        }
#endif
        // plain evaluation and the library's DGEMV baseline (fpe == 1) agree to rounding
        std::vector<double> yd(y0.begin(), y0.begin() + rows);
        exgemv(trans, m, n, 1.0, a.data(), m, 0, x.data(), 1, 0, 1.0, yd.data(), 1, 0, 1);
        double nrm = 0, val = 0;
        for (int i = 0; i < rows; ++i) {
            double s = y0[i];
            for (int k = 0; k < inner; ++k) s += (trans == 'T' ? a[(size_t)i * m + k] : a[(size_t)k * m + i]) * x[k];
            nrm = fmax(nrm, fmax(fabs(s - ref[i]), fabs(yd[i] - ref[i])));
            val = fmax(val, fabs(ref[i]));
        }
        printf("  exgemv %c: max |plain - exact| / max|exact| = %.3g\n", trans, nrm / val);
        if (!(nrm / val < 1e-13)) pass = false;
    }
    {
        const int g = 256;
        std::vector<double> A((size_t)g * g), B((size_t)g * g), C0((size_t)g * g), ref;
        init_fpuniform_matrix(false, g, g, A.data(), g, 10, 0);
        init_fpuniform_matrix(false, g, g, B.data(), g, 10, 0);
        init_fpuniform_matrix(false, g, g, C0.data(), g, 10, 0);
        ref = C0;
        exgemm('N', 'N', g, g, g, 1.0, A.data(), g, B.data(), g, 1.0, ref.data(), g, 0);
        const int fpe[] = {3, 4, 8, 4, 6, 8};
        const bool ee[] = {false, false, false, true, true, true};
        for (int v = 0; v < 6; ++v) {
            std::vector<double> C = C0;
            exgemm('N', 'N', g, g, g, 1.0, A.data(), g, B.data(), g, 1.0, C.data(), g, fpe[v], ee[v]);
            if (!same(C, ref)) { pass = false; printf("exgemm fpe%d%s differs\n", fpe[v], ee[v] ? "ee" : ""); }
        }
#ifdef EXBLAS_VS_MPFR
        {
            // C0 + RN(A*B): the reference kernel's plain fp64 "+=" on the correctly rounded dot (ExGEMM.Superacc.cl:280)
            std::vector<double> D((size_t)g * g);
            mpfr_exgemm_dots(g, g, g, A.data(), g, B.data(), g, D.data(), g);
            for (size_t t = 0; t < D.size(); ++t) D[t] = 1.0 * C0[t] + D[t];
            if (!same(D, ref)) { pass = false; printf("exgemm differs from MPFR\n"); }
            else printf("  exgemm 256^3 == MPFR bit for bit\n");
        }
#endif
        double nrm = 0, val = 0;
        for (int i = 0; i < g; ++i)
            for (int j = 0; j < g; ++j) {
                double s = 0;
                for (int l = 0; l < g; ++l) s += A[(size_t)i * g + l] * B[(size_t)l * g + j];
                s += C0[(size_t)i * g + j];
                nrm = fmax(nrm, fabs(s - ref[(size_t)i * g + j]));
                val = fmax(val, fabs(ref[(size_t)i * g + j]));
            }
        printf("  exgemm 256^3: max |plain - exact| / max|exact| = %.3g\n", nrm / val);
        if (!(nrm / val < 1e-13)) pass = false;
    }
    printf(pass ? "TestPassed; ALL OK!\n" : "TestFailed!\n");
    return pass ? 0 : 1;
}
