/*
 * exblas_oracle.c -- CPU restatement of the reference ExBLAS hot path (see exblas_oracle.h).
 *
 * TEST INFRASTRUCTURE ONLY: loaded by tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg, never by the product path.  Parity status in exblas_oracle.h.
 *
 * Build: gcc -O2 -ffp-contract=off -mfma -fopenmp (see oracle/Makefile).  -ffp-contract=off is
 * mandatory: the error-free transforms below must not be fused (the reference's NVIDIA
 * -cl-fast-relaxed-math flag, ExSUM.Launcher.cpp:41, is unsafe and is not followed).
 *
 * All file:line citations are relative to /root/reference.
 */
#include "exblas_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------------------------------
 * bit helpers (mylibm.hpp:107-141)
 * ---------------------------------------------------------------------------------------- */
static inline uint64_t d2u(double x) { uint64_t u; memcpy(&u, &x, 8); return u; }
static inline double u2d(uint64_t u) { double x; memcpy(&x, &u, 8); return x; }

/* exponent(): mylibm.hpp:107-118 reads the raw biased field minus 0x3ff.  For subnormal
 * inputs that value (-1023) is not the exponent of x and the reference then mis-scales x
 * (exsum([5e-324]) = 2^-1023 in the reference).  DELIBERATE DEVIATION: we return
 * the true exponent for subnormals so that the accumulator stays exact over the whole
 * double range; for every normal input the value is identical to the reference's. */
static inline int orc_exponent(double x)
{
    int be = (int)((d2u(x) >> 52) & 0x7ff);
    if (be != 0) return be - 0x3ff;
    return ilogb(x); /* subnormal: floor(log2|x|) in [-1074,-1023] */
}

/* ------------------------------------------------------------------------------------------
 * Superaccumulator (superaccumulator.hpp:32-129, superaccumulator.cpp)
 * ---------------------------------------------------------------------------------------- */
void orc_sa_init(orc_superacc *sa)
{
    /* superaccumulator.cpp:14-22 */
    memset(sa->acc, 0, sizeof(sa->acc));
    sa->imin = 0;
    sa->imax = ORC_NLIMBS - 1;
    sa->overflowed = 0;
}

/* xadd + seto (mylibm.hpp:182-198): returns the old word, *of = signed overflow of the add. */
static inline int64_t orc_xadd(int64_t *mem, int64_t x, int *of)
{
    int64_t old = *mem, res;
    *of = __builtin_add_overflow(old, x, &res);
    *mem = res; /* two's-complement wrap, like the x86 instruction */
    return old;
}

/* AccumulateWord (superaccumulator.hpp:132-171), TSAFE == 0 */
static inline void orc_accumulate_word(orc_superacc *sa, int64_t x, int i)
{
    int64_t carry = x, carrybit;
    int overflow;
    int64_t oldword = orc_xadd(&sa->acc[i], x, &overflow);
    while (__builtin_expect(overflow, 0)) {
        /* wrapping arithmetic throughout, as on the reference's x86 target */
        carry = (int64_t)((uint64_t)oldword + (uint64_t)carry) >> ORC_DIGITS;
        int s = oldword > 0;
        carrybit = s ? ((int64_t)1 << ORC_K) : -((int64_t)1 << ORC_K);
        orc_xadd(&sa->acc[i], (int64_t)(0 - ((uint64_t)carry << ORC_DIGITS)), &overflow);
        carry += carrybit;
        ++i;
        if (i >= ORC_NLIMBS) {
            sa->overflowed = 1;
            return;
        }
        oldword = orc_xadd(&sa->acc[i], carry, &overflow);
    }
}

/* Accumulate(double) (superaccumulator.hpp:173-194) */
void orc_sa_accumulate(orc_superacc *sa, double x)
{
    if (x == 0) return;
    /* Inf/NaN: the reference's loop below never terminates on them and walks off the limb array
     * (superaccumulator.hpp:173-194 has no check).  The oracle drops the value instead of faulting; tests
     * that feed non-finite data state IEEE's expected answer themselves. */
    if (!(fabs(x) <= DBL_MAX)) return;
    int e = orc_exponent(x);
    int exp_word = e / ORC_DIGITS; /* C truncation, as the reference */
    int iup = exp_word + ORC_FWORDS;
    /* myldexp (mylibm.hpp:130-141) adds to the exponent field; ldexp is the same value for
     * every normal x and is exact for subnormal x as well (deviation noted above). */
    double xscaled = ldexp(x, -ORC_DIGITS * exp_word);
    const double deltaScale = 4503599627370496.0; /* 2^52, superaccumulator.hpp:120 */
    for (int i = iup; xscaled != 0; --i) {
        double xrounded = rint(xscaled);    /* myrint = roundsd, nearest-even */
        int64_t xint = llrint(xscaled);     /* myllrint = cvtsd2si */
        orc_accumulate_word(sa, xint, i);
        xscaled -= xrounded;
        xscaled *= deltaScale;
    }
}

/* Normalize (superaccumulator.cpp:138-162): carry-propagate from imin; every limb ends in
 * [0, 2^52) except the top one, which keeps the remaining (signed) carry. */
int orc_sa_normalize(orc_superacc *sa)
{
    if (sa->imin > sa->imax) return 0;
    int64_t carry_in = sa->acc[sa->imin] >> ORC_DIGITS;
    sa->acc[sa->imin] -= carry_in << ORC_DIGITS;
    int i;
    for (i = sa->imin + 1; i < ORC_NLIMBS; ++i) {
        sa->acc[i] += carry_in;
        int64_t carry_out = sa->acc[i] >> ORC_DIGITS;
        sa->acc[i] -= (int64_t)((uint64_t)carry_out << ORC_DIGITS);
        carry_in = carry_out;
    }
    sa->imax = i - 1;
    sa->acc[sa->imax] += (int64_t)((uint64_t)carry_in << ORC_DIGITS);
    return carry_in < 0;
}

/* Accumulate(Superaccumulator&) (superaccumulator.cpp:68-78) */
void orc_sa_merge(orc_superacc *sa, orc_superacc *other)
{
    orc_sa_normalize(sa);
    orc_sa_normalize(other);
    if (other->imin < sa->imin) sa->imin = other->imin;
    if (other->imax > sa->imax) sa->imax = other->imax;
    for (int i = sa->imin; i <= sa->imax; ++i) sa->acc[i] += other->acc[i];
}

/* OddRoundSumNonnegative (mylibm.hpp:156-171) */
static inline double orc_odd_round_sum_nonneg(double th, double tl)
{
    uint64_t u = d2u(th + tl);
    u |= (uint64_t)(tl != 0.0);
    return u2d(u);
}

/* Round (superaccumulator.cpp:80-134), restated literally INCLUDING its rounding defect
 * (sticky OR-ed into bit 0 of the low word; two's complement of the low word taken as if
 * all lower limbs were zero) -- see SURVEY 8a.  This is the "reference" rounding mode. */
double orc_sa_round_reference(orc_superacc *sa)
{
    if (sa->imin > sa->imax) return 0;
    int negative = orc_sa_normalize(sa);
    const int64_t mask = ((int64_t)1 << ORC_DIGITS) - 1;
    int i;
    /* the reference tests acc[i] before i >= imin; same result, minus the out-of-range read */
    for (i = sa->imax; i >= sa->imin && sa->acc[i] == 0; --i) { }
    if (negative) {
        for (; i >= sa->imin && (sa->acc[i] & mask) == mask; --i) { }
    }
    if (i < 0) return 0.0;
    int64_t hiword = negative ? mask - sa->acc[i] : sa->acc[i];
    double rounded = (double)hiword;
    double hi = ldexp(rounded, (i - ORC_FWORDS) * ORC_DIGITS);
    if (i == 0) return negative ? -hi : hi;
    hiword -= llrint(rounded);
    double mid = ldexp((double)hiword, (i - ORC_FWORDS) * ORC_DIGITS);
    int64_t sticky = 0;
    for (int j = sa->imin; j != i - 1; ++j)
        sticky |= negative ? ((int64_t)1 << ORC_DIGITS) - sa->acc[j] : sa->acc[j];
    int64_t loword = negative ? ((int64_t)1 << ORC_DIGITS) - sa->acc[i - 1] : sa->acc[i - 1];
    loword |= !!sticky;
    double lo = ldexp((double)loword, (i - 1 - ORC_FWORDS) * ORC_DIGITS);
    if (mid != 0) lo = orc_odd_round_sum_nonneg(mid, lo);
    hi = hi + lo;
    return negative ? -hi : hi;
}

static inline unsigned orc_getbit(const uint64_t *d, long b)
{
    if (b < 0) return 0;
    long di = b / ORC_DIGITS;
    if (di > ORC_NLIMBS - 1) di = ORC_NLIMBS - 1;
    long off = b - di * ORC_DIGITS;
    if (off > 63) return 0;
    return (unsigned)((d[di] >> off) & 1u);
}

/* Correct rounding (round-to-nearest-even) of the exact value held in the limbs.  This is
 * what the reference documents ("correctly rounded", README.md:1-19) and what its MPFR
 * oracle computes (tests/test.exsum.cpu.cpp:24-38); Round() above misses it by 1 ulp on a few
 * percent of ill-conditioned inputs. */
double orc_sa_round_exact(orc_superacc *sa)
{
    int negative = orc_sa_normalize(sa);
    /* magnitude as 41 digits of 52 bits + a wide top digit */
    uint64_t d[ORC_NLIMBS];
    const uint64_t mask = ((uint64_t)1 << ORC_DIGITS) - 1;
    if (!negative) {
        for (int i = 0; i < ORC_NLIMBS; ++i) d[i] = (uint64_t)sa->acc[i];
    } else {
        /* two's complement negate across the 52-bit digits (top digit is 64-bit signed) */
        uint64_t borrow = 1; /* -(x) = ~x + 1 digit-wise */
        for (int i = 0; i < ORC_NLIMBS - 1; ++i) {
            uint64_t v = ((~(uint64_t)sa->acc[i]) & mask) + borrow;
            d[i] = v & mask;
            borrow = v >> ORC_DIGITS;
        }
        d[ORC_NLIMBS - 1] = ~(uint64_t)sa->acc[ORC_NLIMBS - 1] + borrow;
    }
    int top = ORC_NLIMBS - 1;
    while (top >= 0 && d[top] == 0) --top;
    if (top < 0) return 0.0;
    /* bit index (from the accumulator LSB = 2^-1092) of the leading one; the top digit is a
     * full 64-bit word, all others hold 52 bits */
    int lead = 63 - __builtin_clzll(d[top]);
    long msb = (long)top * ORC_DIGITS + lead;
    uint64_t mant = 0;
    for (int k = 0; k < 53; ++k) mant = (mant << 1) | orc_getbit(d, msb - k);
    long rb = msb - 53;
    unsigned roundbit = orc_getbit(d, rb), sticky = 0;
    for (long b = rb - 1; b >= 0 && !sticky; --b) sticky |= orc_getbit(d, b);
    if (roundbit && (sticky || (mant & 1))) mant += 1;
    /* value = mant * 2^(msb - 52 - 1092); mant <= 2^53 converts exactly */
    double r = ldexp((double)mant, (int)(msb - 52 - (long)ORC_FWORDS * ORC_DIGITS));
    return negative ? -r : r;
}

void orc_normalize_limbs(int64_t *limbs)
{
    orc_superacc sa;
    orc_sa_init(&sa);
    memcpy(sa.acc, limbs, sizeof(sa.acc));
    orc_sa_normalize(&sa);
    memcpy(limbs, sa.acc, sizeof(sa.acc));
}

double orc_round_limbs(const int64_t *limbs, int mode)
{
    orc_superacc sa;
    orc_sa_init(&sa);
    memcpy(sa.acc, limbs, sizeof(sa.acc));
    return mode == ORC_ROUND_REFERENCE ? orc_sa_round_reference(&sa) : orc_sa_round_exact(&sa);
}

static double orc_finish(orc_superacc *sa, int round_mode, int64_t *limbs_out)
{
    orc_sa_normalize(sa);
    if (limbs_out) memcpy(limbs_out, sa->acc, sizeof(sa->acc));
    return round_mode == ORC_ROUND_REFERENCE ? orc_sa_round_reference(sa) : orc_sa_round_exact(sa);
}

/* ------------------------------------------------------------------------------------------
 * Floating-point expansion, 8 lanes wide like the reference's 2 x Vec4d hot loop
 * (ExSUM.FPE.hpp:316-387 Accumulate(x1,x2); cpu:ExSUM.cpp:246-249).
 * ---------------------------------------------------------------------------------------- */
#define ORC_LANES 8
#define ORC_MAXFPE 8

typedef struct {
    double a[ORC_MAXFPE][ORC_LANES]; /* most significant first (ExSUM.FPE.hpp:84) */
    int n, early_exit;
    orc_superacc *sa;
} orc_fpe;

static void orc_fpe_init(orc_fpe *f, int n, int early_exit, orc_superacc *sa)
{
    memset(f->a, 0, sizeof(f->a));
    f->n = n;
    f->early_exit = early_exit;
    f->sa = sa;
}

/* Knuth2Sum (ExSUM.FPE.hpp:97-104); the AVX2 build uses FMA2Sum (:139-146), which computes
 * the same values (fma(1,x,y) == x+y exactly rounded). */
static inline double orc_two_sum(double a, double b, double *s)
{
    double r = a + b;
    double z = r - a;
    *s = (a - (r - z)) + (b - z);
    return r;
}

/* FlushVector (ExSUM.FPE.hpp:405-417) */
static inline void orc_fpe_flush_vector(orc_fpe *f, const double *x)
{
    for (int j = 0; j < ORC_LANES; ++j) orc_sa_accumulate(f->sa, x[j]);
}

/* Accumulate(x1, x2) (ExSUM.FPE.hpp:316-387) on 8 lanes at once.  The default traits
 * (FPExpansionTraits<EX>, ExSUM.FPE.hpp:23-35) leave FlushHi/Horz2Sum/Victimcache off. */
static inline void orc_fpe_accumulate8(orc_fpe *f, const double *xin)
{
    double x[ORC_LANES];
    for (int j = 0; j < ORC_LANES; ++j) x[j] = xin[j];
    const int n = f->n;
    for (int i = 0; i < n; ++i) {
        double any = 0.0;
        for (int j = 0; j < ORC_LANES; ++j) {
            double s;
            f->a[i][j] = orc_two_sum(f->a[i][j], x[j], &s);
            x[j] = s;
        }
        if (f->early_exit && i != 0) {
            uint64_t orbits = 0;
            for (int j = 0; j < ORC_LANES; ++j) orbits |= d2u(x[j]) << 1; /* != +-0 */
            (void)any;
            if (orbits == 0) return;
        }
    }
    uint64_t orbits = 0;
    for (int j = 0; j < ORC_LANES; ++j) orbits |= d2u(x[j]) << 1;
    if (f->early_exit || orbits != 0) orc_fpe_flush_vector(f, x);
}

/* Flush (ExSUM.FPE.hpp:392-403) */
static void orc_fpe_flush(orc_fpe *f)
{
    for (int i = 0; i < f->n; ++i) {
        orc_fpe_flush_vector(f, f->a[i]);
        for (int j = 0; j < ORC_LANES; ++j) f->a[i][j] = 0;
    }
}

/* Dispatch (cpu:ExSUM.cpp:24-100 == gpu:ExSUM.cpp:64-84): returns the expansion size to use,
 * 0 for superaccumulators only, -1 for "unsupported combination -> result 0.0". */
static int orc_exsum_variant(int fpe, int early_exit)
{
    if (fpe < 2) return 0;
    if (early_exit) {
        if (fpe <= 4) return 4;
        if (fpe <= 6) return 6;
        if (fpe <= 8) return 8;
        return -1;
    }
    if (fpe <= 8) return fpe;
    return -1;
}

static void orc_exsum_range(orc_superacc *sa, const double *a, int64_t inca, int64_t first,
                            int64_t count, int nfpe, int early_exit)
{
    if (nfpe == 0) {
        /* TBBlongsum body (ExSUM.hpp:47-50) */
        for (int64_t i = 0; i < count; ++i) orc_sa_accumulate(sa, a[first + i * inca]);
        return;
    }
    orc_fpe f;
    orc_fpe_init(&f, nfpe, early_exit, sa);
    int64_t body = count & ~(int64_t)7; /* cpu:ExSUM.cpp:243-249 */
    if (inca == 1) {
        for (int64_t i = 0; i < body; i += 8) orc_fpe_accumulate8(&f, a + first + i);
    } else {
        double tmp[8];
        for (int64_t i = 0; i < body; i += 8) {
            for (int j = 0; j < 8; ++j) tmp[j] = a[first + (i + j) * inca];
            orc_fpe_accumulate8(&f, tmp);
        }
    }
    orc_fpe_flush(&f);
    /* scalar tail (cpu:ExSUM.cpp:252-259) */
    for (int64_t i = body; i < count; ++i) orc_sa_accumulate(sa, a[first + i * inca]);
}

double orc_exsum(int n, const double *a, int inca, int offset, int fpe, int early_exit,
                 int round_mode, int64_t *limbs_out)
{
    if (fpe < 0) return NAN; /* reference prints and exit(1)s (cpu:ExSUM.cpp:25-28) */
    int nfpe = orc_exsum_variant(fpe, early_exit);
    if (nfpe < 0) {
        if (limbs_out) memset(limbs_out, 0, sizeof(int64_t) * ORC_NLIMBS);
        return 0.0; /* cpu:ExSUM.cpp:99 */
    }
    orc_superacc sa;
    orc_sa_init(&sa);
    if (n > 0) orc_exsum_range(&sa, a, inca, offset, n, nfpe, early_exit);
    return orc_finish(&sa, round_mode, limbs_out);
}

/* OpenMP slice-per-thread driver (cpu:ExSUM.cpp:235-263).  The reference's log-tree merge
 * with spin flags (:179-215) is replaced by a serial merge after the parallel region --
 * integer limb addition is associative, so the limbs are identical. */
double orc_exsum_omp(int n, const double *a, int fpe, int early_exit, int nthreads,
                     int round_mode, int64_t *limbs_out)
{
    int nfpe = orc_exsum_variant(fpe, early_exit);
    if (nfpe < 0) return 0.0;
    if (nthreads < 1) nthreads = 1;
    orc_superacc *accs = (orc_superacc *)malloc(sizeof(orc_superacc) * (size_t)nthreads);
#pragma omp parallel num_threads(nthreads)
    {
        int tid = 0, tnum = 1;
#ifdef _OPENMP
        tid = omp_get_thread_num();
        tnum = omp_get_num_threads();
#endif
        if (tid < nthreads) {
            orc_sa_init(&accs[tid]);
            int64_t l = ((int64_t)tid * n / tnum) & ~(int64_t)7;
            int64_t r = (tid == tnum - 1) ? n : ((((int64_t)tid + 1) * n / tnum) & ~(int64_t)7);
            orc_exsum_range(&accs[tid], a, 1, l, r - l, nfpe, early_exit);
            orc_sa_normalize(&accs[tid]);
        }
#pragma omp barrier
#pragma omp single
        {
            for (int t = tnum; t < nthreads; ++t) orc_sa_init(&accs[t]);
        }
    }
    for (int t = 1; t < nthreads; ++t) orc_sa_merge(&accs[0], &accs[t]);
    double r = orc_finish(&accs[0], round_mode, limbs_out);
    free(accs);
    return r;
}

/* ------------------------------------------------------------------------------------------
 * ExDOT element step (ExDOT.FPE.cl:226-270; ExDOT.Superacc.cl:244-253)
 * ---------------------------------------------------------------------------------------- */
/* TwoProductFMA (ExDOT.Superacc.cl:25-29) */
static inline double orc_two_prod(double a, double b, double *d)
{
    double p = a * b;
    *d = fma(a, b, -p);
    return p;
}

typedef struct {
    double a[ORC_MAXFPE];
    int n, early_exit;
    orc_superacc *sa;
} orc_fpe1;

static void orc_fpe1_flush_all(orc_fpe1 *f)
{
    for (int i = 0; i < f->n; ++i) {
        orc_sa_accumulate(f->sa, f->a[i]);
        f->a[i] = 0.0;
    }
}

/* One lane of the GPU FPE kernels.  from = first expansion slot the value enters
 * (0 for the product, NBFPE-3 for its rounding error, ExDOT.FPE.cl:254). */
static inline void orc_fpe1_add(orc_fpe1 *f, double x, int from)
{
    for (int i = from; i < f->n; ++i) {
        double s;
        f->a[i] = orc_two_sum(f->a[i], x, &s);
        x = s;
        /* early-exit kernels leave the cascade once the residue is zero
         * (ExDOT.FPE.EX.4.cl main loop); same values either way */
        if (f->early_exit && x == 0.0) return;
    }
    if (x != 0.0) {
        /* residue survived: spill it and the whole expansion (ExDOT.FPE.cl:240-250) */
        orc_sa_accumulate(f->sa, x);
        orc_fpe1_flush_all(f);
    }
}

static int orc_exdot_variant(int fpe, int early_exit)
{
    if (fpe < 3) return 0; /* ExDOT.cpp:78 */
    if (early_exit) {
        if (fpe <= 4) return 4;
        if (fpe <= 6) return 6;
        if (fpe <= 8) return 8;
        return -1;
    }
    if (fpe <= 8) return fpe;
    return -1;
}

static void orc_exdot_range(orc_superacc *sa, const double *a, int64_t inca, int64_t fa,
                            const double *b, int64_t incb, int64_t fb, int64_t count, int nfpe,
                            int early_exit)
{
    if (nfpe == 0) {
        /* ExDOT.Superacc.cl:244-253: accumulate the product and, if non-zero, its error */
        for (int64_t i = 0; i < count; ++i) {
            double r, x = orc_two_prod(a[fa + i * inca], b[fb + i * incb], &r);
            orc_sa_accumulate(sa, x);
            if (r != 0.0) orc_sa_accumulate(sa, r);
        }
        return;
    }
    orc_fpe1 f;
    memset(&f, 0, sizeof(f));
    f.n = nfpe;
    f.early_exit = early_exit;
    f.sa = sa;
    for (int64_t i = 0; i < count; ++i) {
        double r, x = orc_two_prod(a[fa + i * inca], b[fb + i * incb], &r);
        orc_fpe1_add(&f, x, 0);
        if (r != 0.0) orc_fpe1_add(&f, r, nfpe - 3);
    }
    orc_fpe1_flush_all(&f);
}

double orc_exdot(int n, const double *a, int inca, int offseta, const double *b, int incb,
                 int offsetb, int fpe, int early_exit, int round_mode, int64_t *limbs_out)
{
    if (limbs_out) memset(limbs_out, 0, sizeof(int64_t) * ORC_NLIMBS);
    if (n <= 0) return 0.0; /* ExDOT.cpp:70-71 */
    int nfpe = orc_exdot_variant(fpe, early_exit);
    if (nfpe < 0) return 0.0;
    orc_superacc sa;
    orc_sa_init(&sa);
    orc_exdot_range(&sa, a, inca, offseta, b, incb, offsetb, n, nfpe, early_exit);
    return orc_finish(&sa, round_mode, limbs_out);
}

double orc_exdot_omp(int n, const double *a, const double *b, int fpe, int early_exit,
                     int nthreads, int round_mode, int64_t *limbs_out)
{
    int nfpe = orc_exdot_variant(fpe, early_exit);
    if (nfpe < 0 || n <= 0) return 0.0;
    if (nthreads < 1) nthreads = 1;
    orc_superacc *accs = (orc_superacc *)malloc(sizeof(orc_superacc) * (size_t)nthreads);
    for (int t = 0; t < nthreads; ++t) orc_sa_init(&accs[t]);
#pragma omp parallel for num_threads(nthreads) schedule(static, 1)
    for (int t = 0; t < nthreads; ++t) {
        int64_t l = (int64_t)t * n / nthreads, r = ((int64_t)t + 1) * n / nthreads;
        orc_exdot_range(&accs[t], a, 1, l, b, 1, l, r - l, nfpe, early_exit);
        orc_sa_normalize(&accs[t]);
    }
    for (int t = 1; t < nthreads; ++t) orc_sa_merge(&accs[0], &accs[t]);
    double r = orc_finish(&accs[0], round_mode, limbs_out);
    free(accs);
    return r;
}

/* ------------------------------------------------------------------------------------------
 * ExGEMV row step (ExGEMV.Superacc.cl:233-291 'N', :295-392 'T'), column-major A.
 *   y_i = Round( sum_k A(i,k) * fl(alpha * x_k)  (+)  beta*y_i )
 * beta == 0 ignores y, beta == 1 adds y exactly, otherwise TwoProd(beta, y) is added.
 * fpe == 1 is the plain, non-reproducible DGEMV (ExGEMV.cpp:92-94; DGEMV.cl).
 * ---------------------------------------------------------------------------------------- */
int orc_exgemv(char transa, int m, int n, double alpha, const double *a, int lda, int offseta,
               const double *x, int incx, int offsetx, double beta, double *y, int incy,
               int offsety, int fpe, int early_exit, int round_mode)
{
    int trans = (transa == 'T' || transa == 't');
    int rows = trans ? n : m;   /* length of y */
    int inner = trans ? m : n;  /* length of x */
    int nfpe;
    if (fpe == 0) nfpe = 0;
    else if (fpe == 1) nfpe = -2; /* DGEMV */
    else {
        nfpe = orc_exdot_variant(fpe < 3 ? 3 : fpe, early_exit);
        if (fpe == 2 && !early_exit) nfpe = 2; /* ExGEMV.FPE.cl accepts NBFPE=2 as given */
        if (nfpe < 0) return 0;
    }
    for (int i = 0; i < rows; ++i) {
        double *yi = &y[offsety + (int64_t)i * incy];
        if (nfpe == -2) {
            double s = 0.0;
            for (int k = 0; k < inner; ++k) {
                double av = trans ? a[offseta + (int64_t)i * lda + k] : a[offseta + i + (int64_t)lda * k];
                s += alpha * av * x[offsetx + (int64_t)k * incx];
            }
            *yi = (beta == 0.0) ? s : s + beta * (*yi);
            continue;
        }
        orc_superacc sa;
        orc_sa_init(&sa);
        orc_fpe1 f;
        memset(&f, 0, sizeof(f));
        f.n = nfpe;
        f.early_exit = early_exit;
        f.sa = &sa;
        for (int k = 0; k < inner; ++k) {
            double av = trans ? a[offseta + (int64_t)i * lda + k] : a[offseta + i + (int64_t)lda * k];
            double xv = alpha * x[offsetx + (int64_t)k * incx]; /* rounded fold of alpha */
            double r, p = orc_two_prod(av, xv, &r);
            if (nfpe == 0) {
                orc_sa_accumulate(&sa, p);
                if (r != 0.0) orc_sa_accumulate(&sa, r);
            } else {
                orc_fpe1_add(&f, p, 0);
                if (r != 0.0) orc_fpe1_add(&f, r, nfpe >= 3 ? nfpe - 3 : 0);
            }
        }
        if (nfpe > 0) orc_fpe1_flush_all(&f);
        if (beta == 1.0) {
            orc_sa_accumulate(&sa, *yi);
        } else if (beta != 0.0) {
            double r, p = orc_two_prod(beta, *yi, &r);
            orc_sa_accumulate(&sa, p);
            if (r != 0.0) orc_sa_accumulate(&sa, r);
        }
        *yi = (round_mode == ORC_ROUND_REFERENCE) ? orc_sa_round_reference(&sa)
                                                  : orc_sa_round_exact(&sa);
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * ExGEMM element step (ExGEMM.Superacc.cl:246-280): row-major, C_ij += Round(sum_l A_il*B_lj).
 * The reference kernel ignores alpha/beta/trans/ld* (square, alpha=beta=1 is the only tested
 * case, tests/test.exgemm.gpu.cpp:183-184).  We restate the general BLAS meaning so that it
 * reduces to the reference's for that case:
 *   C_ij = Round(alpha * sum) (+) beta*C_ij   with alpha == 1:  Round(sum) + beta*C_ij in fp64
 * The "+=" is a plain fp64 add in the reference (:280); kept (beta==1 -> c + round(sum)).
 * ---------------------------------------------------------------------------------------- */
int orc_exgemm(char transa, char transb, int m, int n, int k, double alpha, const double *a,
               int lda, const double *b, int ldb, double beta, double *c, int ldc, int fpe,
               int early_exit, int round_mode)
{
    int ta = (transa == 'T' || transa == 't'), tb = (transb == 'T' || transb == 't');
    int nfpe = orc_exdot_variant(fpe, early_exit); /* fpe < 3 -> superacc (ExGEMM.cpp:84) */
    if (nfpe < 0) return 0;
    for (int i = 0; i < m; ++i) {
        for (int j = 0; j < n; ++j) {
            orc_superacc sa;
            orc_sa_init(&sa);
            orc_fpe1 f;
            memset(&f, 0, sizeof(f));
            f.n = nfpe;
            f.early_exit = early_exit;
            f.sa = &sa;
            for (int l = 0; l < k; ++l) {
                double av = ta ? a[(int64_t)l * lda + i] : a[(int64_t)i * lda + l];
                double bv = tb ? b[(int64_t)j * ldb + l] : b[(int64_t)l * ldb + j];
                double r, p = orc_two_prod(alpha * av, bv, &r);
                if (nfpe == 0) {
                    orc_sa_accumulate(&sa, p);
                    if (r != 0.0) orc_sa_accumulate(&sa, r);
                } else {
                    orc_fpe1_add(&f, p, 0);
                    if (r != 0.0) orc_fpe1_add(&f, r, nfpe - 3);
                }
            }
            if (nfpe > 0) orc_fpe1_flush_all(&f);
            double s = (round_mode == ORC_ROUND_REFERENCE) ? orc_sa_round_reference(&sa)
                                                           : orc_sa_round_exact(&sa);
            double *cij = &c[(int64_t)i * ldc + j];
            *cij = (beta == 0.0) ? s : beta * (*cij) + s;
        }
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * ExTRSV (ExTRSV.cpp:70-123 dispatch; kernels ExTRSV.lnn.Superacc.cl:254-348 lower,
 * ExTRSV.unn.Superacc.cl:262-355 upper), column-major A, solves A x = b in place (x holds b):
 *     x_i = fl( Round( b_i - sum_j A(i,j) * x_j ) / A(i,i) ),   j < i (lower) or j > i (upper),
 * rows in dependency order; the sum is exact (TwoProductFMA into the superaccumulator,
 * lnn.Superacc.cl:305-311,:331-336), Round is the superaccumulator rounding, the division an
 * ordinary fp64 one (:326-328) -- so x_i carries two roundings, by design of the reference.
 * The reference kernels ignore transa and diag (isunit = 0, lnn.Superacc.cl:272; they always
 * divide by the stored diagonal, which init_*_tr_matrix sets to 1 for diag == 'U') and incx /
 * the offsets; we restate the BLAS meaning of all of them ('T': A(i,j) is read at a[i*lda+j];
 * diag == 'U': the diagonal is not read).  fpe == 1 is the plain, non-reproducible DTRSV
 * (ExTRSV.cpp:76-77).  fpe >= 9 names iterative-refinement kernel files the reference does not
 * ship (ExTRSV.cpp:91-120; no *.IR.cl / *.ExIR.*.cl under src/gpu/blas/blas2): returns -1.
 * Parity: no CPU ExTRSV exists in the reference and its OpenCL path cannot run here, so this
 * function is pinned by MPFR only (mpfr_oracle.c:mpfr_extrsv, bit for bit) and by the
 * reference test's own criterion (test.extrsv.gpu.cpp:27-92, inf-norm error <= 1e-13).
 * ---------------------------------------------------------------------------------------- */
int orc_extrsv(char uplo, char transa, char diag, int n, const double *a, int lda, int offseta,
               double *x, int incx, int offsetx, int fpe, int early_exit, int round_mode)
{
    if (fpe < 0 || fpe >= 9) return -1;
    int nfpe;
    if (fpe == 0) nfpe = 0;
    else if (fpe == 1) nfpe = -2;
    else if (early_exit) nfpe = fpe <= 4 ? 4 : (fpe <= 6 ? 6 : 8); /* ExTRSV.cpp:79-86 */
    else nfpe = fpe;                                                 /* ExTRSV.cpp:87-88, NBFPE = fpe % 10 */
    const int lower = (uplo == 'L' || uplo == 'l');
    const int trans = (transa == 'T' || transa == 't');
    const int unit = (diag == 'U' || diag == 'u');
    /* A**T of a lower matrix is upper and vice versa: the dependency order follows the logical shape */
    const int fwd = lower != trans;
    if (nfpe == 0 && !trans && n > 512) {
        /* Same arithmetic, column by column (right-looking): one superaccumulator per row, and every finished
         * x_i is applied to all remaining rows at once.  Exact accumulation does not depend on the order, so
         * the result is the row-by-row one below; this form walks A along its columns (contiguous) and lets
         * OpenMP share the rows, which is what makes n = 40000 checkable in seconds. */
        orc_superacc *sa = (orc_superacc *)malloc((size_t)n * sizeof(orc_superacc));
        if (!sa) return -2;
        for (int i = 0; i < n; ++i) orc_sa_init(&sa[i]);
        for (int s = 0; s < n; ++s) {
            const int i = fwd ? s : n - 1 - s;
            double *xi = &x[offsetx + (int64_t)i * incx];
            orc_sa_accumulate(&sa[i], *xi);
            double v = (round_mode == ORC_ROUND_REFERENCE) ? orc_sa_round_reference(&sa[i]) : orc_sa_round_exact(&sa[i]);
            if (!unit) v = v / a[offseta + (int64_t)i * lda + i];
            *xi = v;
            const int r0 = fwd ? i + 1 : 0, r1 = fwd ? n : i;
            const double *col = &a[offseta + (int64_t)lda * i];
#pragma omp parallel for schedule(static) if (r1 - r0 > 2048)
            for (int r = r0; r < r1; ++r) {
                double e, p = orc_two_prod(col[r], -v, &e);
                orc_sa_accumulate(&sa[r], p);
                if (e != 0.0) orc_sa_accumulate(&sa[r], e);
            }
        }
        free(sa);
        return 0;
    }
    for (int s = 0; s < n; ++s) {
        const int i = fwd ? s : n - 1 - s;
        const int j0 = fwd ? 0 : i + 1, j1 = fwd ? i : n;
        double *xi = &x[offsetx + (int64_t)i * incx];
        double v;
        if (nfpe == -2) {
            v = *xi;
            for (int j = j0; j < j1; ++j) {
                double av = trans ? a[offseta + (int64_t)i * lda + j] : a[offseta + i + (int64_t)lda * j];
                v -= av * x[offsetx + (int64_t)j * incx];
            }
        } else {
            orc_superacc sa;
            orc_sa_init(&sa);
            orc_fpe1 f;
            memset(&f, 0, sizeof(f));
            f.n = nfpe;
            f.early_exit = early_exit;
            f.sa = &sa;
            for (int j = j0; j < j1; ++j) {
                double av = trans ? a[offseta + (int64_t)i * lda + j] : a[offseta + i + (int64_t)lda * j];
                double r, p = orc_two_prod(av, -x[offsetx + (int64_t)j * incx], &r);
                if (nfpe == 0) {
                    orc_sa_accumulate(&sa, p);
                    if (r != 0.0) orc_sa_accumulate(&sa, r);
                } else {
                    orc_fpe1_add(&f, p, 0);
                    if (r != 0.0) orc_fpe1_add(&f, r, nfpe >= 3 ? nfpe - 3 : 0);
                }
            }
            if (nfpe > 0) orc_fpe1_flush_all(&f);
            orc_sa_accumulate(&sa, *xi);
            v = (round_mode == ORC_ROUND_REFERENCE) ? orc_sa_round_reference(&sa) : orc_sa_round_exact(&sa);
        }
        if (!unit) v = v / a[offseta + (int64_t)i * lda + i];
        *xi = v;
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * Generators
 * ---------------------------------------------------------------------------------------- */
void orc_srand(unsigned seed) { srand(seed); }

/* init_naive (common.cpp:147-150) */
void orc_init_naive(int n, double *a)
{
    for (int i = 0; i != n; ++i) a[i] = 1.1;
}

/* randDouble (common.cpp:18-28) + init_fpuniform (:30-33) */
static double orc_rand_double(int emin, int emax, int neg_ratio)
{
    double x = (double)rand() / (double)(RAND_MAX * .99) + 1.;
    int e = (rand() % (emax - emin)) + emin;
    if (neg_ratio > 1 && rand() % neg_ratio == 0) x = -x;
    return ldexp(x, e);
}

void orc_init_fpuniform_rand(int n, double *a, int range, int emax)
{
    for (int i = 0; i != n; ++i) a[i] = orc_rand_double(emax - range, emax, 1);
}

/* init_ill_cond (common.cpp:113-145) */
void orc_init_ill_cond_rand(int n, double *a, double c)
{
    int n2 = (int)round(n / 2);
    for (int i = 0; i != n; ++i) a[i] = 0.0;
    double *e = (double *)malloc((size_t)n * sizeof(double));
    double b = log2(c);
    for (int i = 0; i != n2; ++i) {
        double x = (double)rand() / (double)RAND_MAX;
        e[i] = round(x * b / 2);
    }
    e[0] = round(b / 2) + 1.;
    e[n - 1] = 0;
    for (int i = 0; i != n2; ++i) {
        double x = (double)rand() / (double)RAND_MAX;
        a[i] = (2. * x - 1.) * pow(2., e[i]);
    }
    double step = (b / 2) / (n - n2);
    for (int i = n2; i != n; ++i) {
        double x = (double)rand() / (double)RAND_MAX;
        e[i] = step * (i - n2);
        a[i] = (2. * x - 1.) * pow(2., e[i]);
    }
    free(e);
}

/* Counter-based generators (ours): element i depends only on (kind, seed, i, n_total, p0, p1)
 * and uses integer arithmetic, exact int->double conversions and multiplications by powers
 * of two only, so exblas_amd/csrc/gen.hip produces the same bits on the GPU.  They follow the
 * *shape* of the reference's distributions (common.cpp), not its rand() stream:
 *   0 naive            1.1                                            (common.cpp:147)
 *   1 fpuniform        m in [1,2) * 2^(emax-range+U[0,range)), >0      (common.cpp:18-33)
 *   2 lognormal-like   m in [1,2) * 2^(rint(z*p1/ln2)+rint(p0/ln2)), z ~ Irwin-Hall(4) normalised
 *                      (stands in for std::lognormal_distribution, common.cpp:66-73)
 *   3 ill_cond(c=p0)   (2U-1) * 2^e, first half e = U[0, b/2], e_0 = b/2+1; second half e
 *                      ramps 0 -> b/2 (integer part)                   (common.cpp:113-145)
 *   4 cancel(E=p0)     a[i+n/2] = -a[i], a[i] = (2U-1)*2^U[0,E); a[n/2-1] = 1, a[n-1] = 2^-60:
 *                      exact sum 1 + 2^-60, condition number ~ 2^E * n
 *   5 fpuniform with random sign
 */
static inline uint64_t orc_mix64(uint64_t z)
{
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
static inline uint64_t orc_rnd(uint64_t seed, uint64_t i, uint64_t k)
{
    return orc_mix64(seed * 0xD1342543DE82EF95ull + (2 * i + k + 1) * 0x9E3779B97F4A7C15ull);
}
static inline double orc_pow2(int e) /* normal range only */
{
    return u2d((uint64_t)(e + 1023) << 52);
}
static inline double orc_mant12(uint64_t r) { return u2d(0x3FF0000000000000ull | (r >> 12)); }
static inline double orc_mant_signed(uint64_t r) /* (2U-1), U 53-bit in [0,1) */
{
    int64_t k = (int64_t)(r >> 11);
    return (double)(2 * k - ((int64_t)1 << 53)) * 0x1p-53;
}
static inline uint32_t orc_uni(uint64_t r, uint32_t range)
{
    return (uint32_t)(((r >> 32) * (uint64_t)range) >> 32);
}

static double orc_gen_one(int kind, uint64_t seed, int64_t i, int64_t n, double p0, double p1)
{
    uint64_t r0 = orc_rnd(seed, (uint64_t)i, 0), r1 = orc_rnd(seed, (uint64_t)i, 1);
    switch (kind) {
    case 0:
        return 1.1;
    case 1:
    case 5: {
        int range = (int)p0, emax = (int)p1;
        int e = emax - range + (range > 0 ? (int)orc_uni(r1, (uint32_t)range) : 0);
        double v = orc_mant12(r0) * orc_pow2(e);
        if (kind == 5 && (r1 & 1)) v = -v;
        return v;
    }
    case 2: {
        /* z: sum of four 16-bit uniforms, centred; std = 65536/sqrt(3) */
        int64_t z = (int64_t)(r1 & 0xffff) + (int64_t)((r1 >> 16) & 0xffff) +
                    (int64_t)((r1 >> 32) & 0xffff) + (int64_t)((r1 >> 48) & 0xffff) - 2 * 65535;
        double scale = p1 * (1.0 / (0.6931471805599453 * 37837.22690659431));
        int e = (int)rint((double)z * scale) + (int)rint(p0 * (1.0 / 0.6931471805599453));
        if (e > 1000) e = 1000;
        if (e < -1000) e = -1000;
        return orc_mant12(r0) * orc_pow2(e);
    }
    case 3: {
        int bh = (int)rint(log2(p0) * 0.5); /* b/2 */
        int64_t n2 = n / 2;
        int e;
        if (i < n2) e = (i == 0) ? bh + 1 : (int)orc_uni(r1, (uint32_t)(bh + 1));
        else e = (n - n2 > 0) ? (int)(((i - n2) * (int64_t)bh) / (n - n2)) : 0;
        return orc_mant_signed(r0) * orc_pow2(e);
    }
    case 4: {
        int64_t h = n / 2;
        if (i >= 2 * h) return 0.0;
        if (i == h - 1) return 1.0;
        if (i == 2 * h - 1) return 0x1p-60;
        int64_t j = (i < h) ? i : i - h;
        uint64_t q0 = orc_rnd(seed, (uint64_t)j, 0), q1 = orc_rnd(seed, (uint64_t)j, 1);
        int E = (int)p0;
        double v = orc_mant_signed(q0) * orc_pow2(E > 0 ? (int)orc_uni(q1, (uint32_t)E) : 0);
        return (i < h) ? v : -v;
    }
    default:
        return 0.0;
    }
}

void orc_gen_ctr(int kind, uint64_t seed, int64_t first, int64_t count, int64_t n_total,
                 double p0, double p1, double *out)
{
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < count; ++i)
        out[i] = orc_gen_one(kind, seed, first + i, n_total, p0, p1);
}
