// blas1.hpp -- level-1 entry points of the MI355X backend.
// Signatures are those of the reference's include/blas1.hpp:48 (exsum) and :74 (exdot), so code
// written against the reference compiles and links against libexblas.so unchanged.
#ifndef BLAS1_HPP_
#define BLAS1_HPP_

#include "config.h"

/**
 * Exact, reproducible sum of a[offset + i*inca], i < Ng: the correctly rounded value of the exact
 * sum, independent of the GPU count and launch geometry.
 *   fpe < 2            superaccumulators only
 *   fpe in [2,8]       floating-point expansion of that size in front of the superaccumulator
 *   early_exit         expansion sizes 4 / 6 / 8 for fpe <= 4 / 6 / 8 with the early-exit test
 *   parallel           accepted and ignored (the GPU path is always parallel)
 * Unsupported (fpe, early_exit) combinations return 0.0; fpe < 0 prints and exits, as upstream.
 */
double exsum(const int Ng, double *ag, const int inca, const int offset, const int fpe,
             const bool early_exit = false, const bool parallel = true);

/**
 * Exact, reproducible dot product of a[offseta + i*inca] and b[offsetb + i*incb], i < Ng.
 *   fpe < 3            superaccumulators only; otherwise as exsum.  Ng <= 0 returns 0.0.
 */
double exdot(const int Ng, double *ag, const int inca, const int offseta, double *bg, const int incb,
             const int offsetb, const int fpe, const bool early_exit = false);

#endif // BLAS1_HPP_
