/*
 * exblas_oracle.h -- CPU restatement of the reference ExBLAS hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library, and only
 * as the checker.  The shipped path is exblas_amd/csrc (HIP) and fails loudly without a GPU.
 *
 * Parity status: PINNED for ExSUM (every variant is checked bit-for-bit against the compiled
 * reference arithmetic core in oracle/_ref and against MPFR, see tests/test_oracle.py and
 * tests/golden/); ExDOT/ExGEMV/ExGEMM have no runnable reference implementation on the CPU
 * (the reference only ships OpenCL kernels for them) and are pinned by MPFR exactly as the
 * reference's own tests define the oracle (tests/test.exdot.gpu.cpp:24-46,
 * tests/test.exgemv.gpu.cpp:35-103, tests/test.exgemm.gpu.cpp:53-125).
 *
 * All file:line citations are relative to /root/reference.
 */
#ifndef EXBLAS_ORACLE_H_
#define EXBLAS_ORACLE_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Geometry of the reference CPU superaccumulator (superaccumulator.hpp:118-120,
 * superaccumulator.cpp:14-22): 21 fraction words + 20 exponent words of 52 payload bits. */
#define ORC_K 12
#define ORC_DIGITS 52
#define ORC_FWORDS 21
#define ORC_EWORDS 20
#define ORC_NLIMBS 41

/* rounding modes for the final limbs -> double step */
#define ORC_ROUND_EXACT 0     /* correctly rounded (RN-even) value of the exact sum = MPFR */
#define ORC_ROUND_REFERENCE 1 /* bug-compatible restatement of Superaccumulator::Round */

typedef struct {
    int64_t acc[ORC_NLIMBS];
    int imin, imax;
    int overflowed;
} orc_superacc;

void orc_sa_init(orc_superacc *sa);
void orc_sa_accumulate(orc_superacc *sa, double x);
void orc_sa_merge(orc_superacc *sa, orc_superacc *other);
int orc_sa_normalize(orc_superacc *sa); /* returns 1 when the value is negative */
double orc_sa_round_reference(orc_superacc *sa);
double orc_sa_round_exact(orc_superacc *sa);

/* limbs (41 x int64, any carry-save state) -> double.  Normalises a private copy. */
double orc_round_limbs(const int64_t *limbs, int mode);
void orc_normalize_limbs(int64_t *limbs);

/* ExSUM: element count semantics of the GPU backend (a[offset + i*inca], i < n).
 * limbs_out (may be NULL) receives the 41 normalised limbs. Returns the rounded double. */
double orc_exsum(int n, const double *a, int inca, int offset, int fpe, int early_exit,
                 int round_mode, int64_t *limbs_out);
/* OpenMP-sliced variant of the FPE path (cpu:ExSUM.cpp:235-263); the CPU baseline "port". */
double orc_exsum_omp(int n, const double *a, int fpe, int early_exit, int nthreads,
                     int round_mode, int64_t *limbs_out);

double orc_exdot(int n, const double *a, int inca, int offseta, const double *b, int incb,
                 int offsetb, int fpe, int early_exit, int round_mode, int64_t *limbs_out);
double orc_exdot_omp(int n, const double *a, const double *b, int fpe, int early_exit,
                     int nthreads, int round_mode, int64_t *limbs_out);

int orc_exgemv(char transa, int m, int n, double alpha, const double *a, int lda, int offseta,
               const double *x, int incx, int offsetx, double beta, double *y, int incy,
               int offsety, int fpe, int early_exit, int round_mode);

int orc_exgemm(char transa, char transb, int m, int n, int k, double alpha, const double *a,
               int lda, const double *b, int ldb, double beta, double *c, int ldc, int fpe,
               int early_exit, int round_mode);
/* x := A^-1 x (or A^-T x), A triangular, column-major; returns -1 for the fpe values (>= 9) whose kernels the
 * reference does not ship */
int orc_extrsv(char uplo, char transa, char diag, int n, const double *a, int lda, int offseta,
               double *x, int incx, int offsetx, int fpe, int early_exit, int round_mode);

/* Input generators.  The *_rand ones follow src/common/common.cpp and draw from glibc
 * rand() (call srand() first for a fixed stream); the *_ctr ones are our counter-based
 * restatements (splitmix64(seed, i)) that the HIP generators reproduce bit for bit. */
void orc_init_naive(int n, double *a);
void orc_init_fpuniform_rand(int n, double *a, int range, int emax);
void orc_init_ill_cond_rand(int n, double *a, double c);
void orc_srand(unsigned seed);

void orc_gen_ctr(int kind, uint64_t seed, int64_t first, int64_t count, int64_t n_total,
                 double p0, double p1, double *out);

#ifdef __cplusplus
}
#endif
#endif
