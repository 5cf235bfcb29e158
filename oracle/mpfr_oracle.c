/*
 * mpfr_oracle.c -- the MPFR oracles exactly as the reference's own tests define them.
 *
 * TEST INFRASTRUCTURE ONLY (see exblas_oracle.h).  Links the system libmpfr/libgmp (present in
 * this image; headers under /opt/conda/include).  Optional: tests skip when it is not built.
 * Citations relative to /root/reference.
 */
#include <mpfr.h>
#include <stdint.h>

/* tests/test.exsum.cpu.cpp:24-38 (== tests/test.exsum.gpu.cpp:23-38): 2098-bit accumulator */
double mpfr_exsum(long n, const double *a, long inca, long offset)
{
    mpfr_t acc;
    mpfr_init2(acc, 2098);
    mpfr_set_zero(acc, 0);
    for (long i = 0; i != n; ++i) mpfr_add_d(acc, acc, a[offset + i * inca], MPFR_RNDN);
    double r = mpfr_get_d(acc, MPFR_RNDN);
    mpfr_clear(acc);
    return r;
}

/* tests/test.exdot.gpu.cpp:24-46: exact 128-bit products, 4196-bit sum */
double mpfr_exdot(long n, const double *a, long inca, long offa, const double *b, long incb, long offb)
{
    mpfr_t sum, dot, op;
    mpfr_init2(op, 64);
    mpfr_init2(dot, 128);
    mpfr_init2(sum, 4196);
    mpfr_set_zero(dot, 0);
    mpfr_set_zero(sum, 0);
    for (long i = 0; i < n; i++) {
        mpfr_set_d(op, a[offa + i * inca], MPFR_RNDN);
        mpfr_mul_d(dot, op, b[offb + i * incb], MPFR_RNDN);
        mpfr_add(sum, sum, dot, MPFR_RNDN);
    }
    double r = mpfr_get_d(sum, MPFR_RNDN);
    mpfr_clear(op);
    mpfr_clear(dot);
    mpfr_clear(sum);
    return r;
}

/* tests/test.exgemv.gpu.cpp:35-103, column-major; y_out receives round(alpha*A*x + beta*y).
 * The products carry 192 bits here (the reference uses 128, exact only for alpha a power of
 * two -- the only case it tests, :160); the sum is widened to 4196 bits like the dot oracle. */
void mpfr_exgemv(char trans, int m, int n, double alpha, const double *a, int lda, const double *x,
                 int incx, double beta, const double *y, int incy, double *y_out)
{
    mpfr_t sum, dot;
    mpfr_init2(dot, 192);
    mpfr_init2(sum, 4196);
    int rows = (trans == 'T') ? n : m, inner = (trans == 'T') ? m : n;
    for (int i = 0; i < rows; i++) {
        mpfr_set_d(sum, 0.0, MPFR_RNDN);
        for (int j = 0; j < inner; j++) {
            double av = (trans == 'T') ? a[(long)i * lda + j] : a[(long)j * lda + i];
            mpfr_set_d(dot, av, MPFR_RNDN);
            mpfr_mul_d(dot, dot, alpha, MPFR_RNDN);
            mpfr_mul_d(dot, dot, x[(long)j * incx], MPFR_RNDN);
            mpfr_add(sum, sum, dot, MPFR_RNDN);
        }
        mpfr_set_d(dot, y[(long)i * incy], MPFR_RNDN);
        mpfr_mul_d(dot, dot, beta, MPFR_RNDN);
        mpfr_add(sum, sum, dot, MPFR_RNDN);
        y_out[i] = mpfr_get_d(sum, MPFR_RNDN);
    }
    mpfr_clear(dot);
    mpfr_clear(sum);
}

/* tests/test.exgemm.gpu.cpp:53-125, row-major: d_out[i][j] = RN(sum_l a_il*b_lj) (the
 * reference rounds the sum DOWN, :74/:82, and relies on its 1e-15 tolerance; we round to
 * nearest, which is what "correctly rounded" means and what the kernels produce). */
void mpfr_exgemm_dots(int m, int n, int k, const double *a, int lda, const double *b, int ldb,
                      double *d_out, int ldd)
{
    mpfr_t sum, dot, op1;
    mpfr_init2(op1, 64);
    mpfr_init2(dot, 192);
    mpfr_init2(sum, 4196);
    for (int i = 0; i < m; i++)
        for (int j = 0; j < n; j++) {
            mpfr_set_d(sum, 0.0, MPFR_RNDN);
            for (int l = 0; l < k; l++) {
                mpfr_set_d(op1, a[(long)i * lda + l], MPFR_RNDN);
                mpfr_mul_d(dot, op1, b[(long)l * ldb + j], MPFR_RNDN);
                mpfr_add(sum, sum, dot, MPFR_RNDN);
            }
            d_out[(long)i * ldd + j] = mpfr_get_d(sum, MPFR_RNDN);
        }
    mpfr_clear(op1);
    mpfr_clear(dot);
    mpfr_clear(sum);
}

/* tests/test.extrsv.gpu.cpp:27-66, column-major, x holds b on entry.  two_step != 0: the sum
 * (exact: 4196 bits, 128-bit products) is rounded to double and then divided in fp64, which is
 * what the reference kernels compute (ExTRSV.lnn.Superacc.cl:323-328); two_step == 0: the
 * reference test's own oracle, which divides inside MPFR and rounds once -- the two differ by
 * an ulp now and then, hence the test's 1e-13 tolerance (:141). */
void mpfr_extrsv(char uplo, char trans, char diag, int n, const double *a, int lda, double *x, int incx,
                 int two_step)
{
    mpfr_t sum, dot;
    mpfr_init2(dot, 128);
    mpfr_init2(sum, 4196);
    const int lower = (uplo == 'L'), tr = (trans == 'T'), unit = (diag == 'U');
    const int fwd = lower != tr;
    for (int s = 0; s < n; s++) {
        const int i = fwd ? s : n - 1 - s;
        const int j0 = fwd ? 0 : i + 1, j1 = fwd ? i : n;
        mpfr_set_d(sum, 0.0, MPFR_RNDN);
        for (int j = j0; j < j1; j++) {
            double av = tr ? a[(long)i * lda + j] : a[(long)j * lda + i];
            mpfr_set_d(dot, av, MPFR_RNDN);
            mpfr_mul_d(dot, dot, -x[(long)j * incx], MPFR_RNDN);
            mpfr_add(sum, sum, dot, MPFR_RNDN);
        }
        mpfr_add_d(sum, sum, x[(long)i * incx], MPFR_RNDN);
        double v;
        if (two_step) {
            v = mpfr_get_d(sum, MPFR_RNDN);
            if (!unit) v = v / a[(long)i * lda + i];
        } else {
            if (!unit) mpfr_div_d(sum, sum, a[(long)i * lda + i], MPFR_RNDN);
            v = mpfr_get_d(sum, MPFR_RNDN);
        }
        x[(long)i * incx] = v;
    }
    mpfr_clear(dot);
    mpfr_clear(sum);
    mpfr_free_cache();
}
