/*
 * ref_driver.cpp -- OUR driver around the reference's own CPU arithmetic core.
 *
 * TEST INFRASTRUCTURE ONLY (see exblas_oracle.h).  This file is compiled together with the
 * reference sources *where they lie* under /root/reference (never copied):
 *     src/cpu/blas/blas1/superaccumulator.cpp   (Superaccumulator: Accumulate/Normalize/Round)
 *     src/cpu/blas/blas1/ExSUM.FPE.hpp          (FPExpansionVect, header-only, included below)
 *     src/common/common.cpp                     (init_* input generators)
 * into oracle/_ref/libexblas_ref.so by oracle/Makefile.  The reference's own driver file
 * src/cpu/blas/blas1/ExSUM.cpp needs oneTBB headers, which this image lacks, so it is treated
 * as unbuildable; the slice-per-thread loop below is our restatement of its lines 235-263 and
 * dispatch lines 24-100.  Everything arithmetic (TwoSum cascade, digit split, overflow-driven
 * carries, Normalize, Round) is the reference's compiled code.
 */
#include <iostream> // ExSUM.FPE.hpp uses std::cout without including it
#include <cstring>
#include <vector>
#include <omp.h>

#include "superaccumulator.hpp"
#include "ExSUM.FPE.hpp"
#include "common.hpp"

namespace {

template <typename CACHE>
void fpe_slice(Superaccumulator &acc, const double *a, long l, long r /* exclusive, 8-aligned length */)
{
    CACHE cache(acc);
    for (long i = l; i + 8 <= r; i += 8)
        cache.Accumulate(Vec4d().load(a + i), Vec4d().load(a + i + 4));
    cache.Flush();
}

template <typename CACHE>
void run_fpe(std::vector<Superaccumulator> &acc, const double *a, long n, int nthreads)
{
#pragma omp parallel num_threads(nthreads)
    {
        int tid = omp_get_thread_num(), tnum = omp_get_num_threads();
        long l = ((long)tid * n / tnum) & ~7l;
        long r = (tid == tnum - 1) ? n : ((((long)tid + 1) * n / tnum) & ~7l);
        long body = l + ((r - l) & ~7l);
        fpe_slice<CACHE>(acc[tid], a, l, body);
        for (long i = body; i < r; ++i) acc[tid].Accumulate(a[i]);
        acc[tid].Normalize();
    }
}

} // namespace

extern "C" {

/* returns Superaccumulator::Round(); limbs_out (41 x int64, may be null) = normalised limbs */
double ref_exsum(long n, const double *a, int fpe, int early_exit, int nthreads, int64_t *limbs_out)
{
    if (nthreads < 1) nthreads = 1;
    std::vector<Superaccumulator> acc(nthreads);
    if (fpe < 2) {
#pragma omp parallel num_threads(nthreads)
        {
            int tid = omp_get_thread_num(), tnum = omp_get_num_threads();
            long l = (long)tid * n / tnum, r = ((long)tid + 1) * n / tnum;
            for (long i = l; i < r; ++i) acc[tid].Accumulate(a[i]);
            acc[tid].Normalize();
        }
    } else if (early_exit) {
        if (fpe <= 4) run_fpe<FPExpansionVect<Vec4d, 4, FPExpansionTraits<true> > >(acc, a, n, nthreads);
        else if (fpe <= 6) run_fpe<FPExpansionVect<Vec4d, 6, FPExpansionTraits<true> > >(acc, a, n, nthreads);
        else if (fpe <= 8) run_fpe<FPExpansionVect<Vec4d, 8, FPExpansionTraits<true> > >(acc, a, n, nthreads);
        else return 0.0;
    } else {
        switch (fpe) {
        case 2: run_fpe<FPExpansionVect<Vec4d, 2> >(acc, a, n, nthreads); break;
        case 3: run_fpe<FPExpansionVect<Vec4d, 3> >(acc, a, n, nthreads); break;
        case 4: run_fpe<FPExpansionVect<Vec4d, 4> >(acc, a, n, nthreads); break;
        case 5: run_fpe<FPExpansionVect<Vec4d, 5> >(acc, a, n, nthreads); break;
        case 6: run_fpe<FPExpansionVect<Vec4d, 6> >(acc, a, n, nthreads); break;
        case 7: run_fpe<FPExpansionVect<Vec4d, 7> >(acc, a, n, nthreads); break;
        case 8: run_fpe<FPExpansionVect<Vec4d, 8> >(acc, a, n, nthreads); break;
        default: return 0.0;
        }
    }
    for (int t = 1; t < nthreads; ++t) acc[0].Accumulate(acc[t]);
    acc[0].Normalize();
    if (limbs_out) {
        std::vector<int64_t> v = acc[0].get_accumulator();
        std::memcpy(limbs_out, v.data(), v.size() * sizeof(int64_t));
    }
    return acc[0].Round();
}

double ref_round_limbs(const int64_t *limbs)
{
    Superaccumulator acc;
    std::vector<int64_t> v(limbs, limbs + acc.get_f_words() + acc.get_e_words());
    Superaccumulator b(v);
    return b.Round();
}

int ref_nlimbs(void)
{
    Superaccumulator acc;
    return acc.get_f_words() + acc.get_e_words();
}

/* the reference's generators (src/common/common.cpp), glibc rand() stream */
void ref_srand(unsigned seed) { srand(seed); }
void ref_init_naive(int n, double *a) { init_naive(n, a); }
void ref_init_fpuniform(int n, double *a, int range, int emax) { init_fpuniform(n, a, range, emax); }
void ref_init_ill_cond(int n, double *a, double c) { init_ill_cond(n, a, c); }
void ref_init_lognormal(int n, double *a, double mean, double stddev) { init_lognormal(n, a, mean, stddev); }

} // extern "C"
