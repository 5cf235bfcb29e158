// fpe.hip.h -- register-resident floating-point expansions (FPE) shared by the blas1/2/3 kernels.
//
// Behavioural counterpart of the reference's expansion code (src/cpu/blas/blas1/ExSUM.FPE.hpp:316-417
// FPExpansionVect::Accumulate/Flush; src/gpu/blas/blas1/ExSUM.FPE.cl:230-388 and ExDOT.FPE.cl:226-270 for
// the per-work-item GPU form).  One expansion = N doubles per lane, most significant first; an element is
// pushed down the cascade with Knuth TwoSum and whatever is left after N levels goes to a "sink": an
// integer superaccumulator in LDS (blas1, gemvT, gemm) or a per-row one in global memory (gemv).
#pragma once
#include "superacc.hip.h"

namespace exb {

typedef double d2_t __attribute__((ext_vector_type(2)));

template <bool NT>
__device__ __forceinline__ d2_t ld2(const d2_t *p)
{
    if constexpr (NT) return __builtin_nontemporal_load(p);
    else return *p;
}

// biased exponent field of a double, and the guard threshold 2^1000 (see fpe_absorb_sink)
__device__ __forceinline__ unsigned expo_field(double x) { return ((unsigned)__double2hiint(x) >> 20) & 0x7ffu; }
constexpr unsigned BIG_EXPO = 1023u + 1000u;
// a product fl(a b) with an exponent field below this one (|a b| < 2^-968, zero included) may have bits below 2^-1074: its
// TwoProd error term is then not representable
constexpr unsigned LOW_EXPO = 1023u - 968u;

// "is any of these doubles non-zero" with 32-bit integer ops only (two per element instead of an fp64
// compare each): OR of the low words and the high words shifted left by one (drops the sign, so -0.0 is zero)
template <int CNT, int ZM = 0>
__device__ __forceinline__ bool any_nonzero(const double (&x)[CNT])
{
    if constexpr (ZM == 0) {
        unsigned bits = 0;
#pragma unroll
        for (int j = 0; j < CNT; ++j)
            bits |= ((unsigned)__double2hiint(x[j]) << 1) | (unsigned)__double2loint(x[j]);
        return bits != 0;
    } else {  // fp64 compares (A/B variant)
        bool nz = false;
#pragma unroll
        for (int j = 0; j < CNT; ++j) nz |= (x[j] != 0.0);
        return nz;
    }
}

// ---- sinks -----------------------------------------------------------------------------------
template <int COPIES>
struct LdsSink {
    long long *col;   // this lane's column of the wave's LDS accumulator, stride COPIES between limbs
    unsigned &flags;
    __device__ __forceinline__ void add(double x) { lds_add<COPIES>(col, x, flags); }
    __device__ __forceinline__ void note(unsigned bits) { flags |= bits; }
};

// limbs of one accumulator in global memory: acc[0..NL) + three non-finite indicators at NL..NL+2
struct GlobalSink {
    long long *acc;
    __device__ __forceinline__ void note(unsigned) {}  // rows of a GEMV have no flag channel (y is plain doubles)
    __device__ __forceinline__ void add(double x)
    {
        const unsigned long long u = (unsigned long long)__double_as_longlong(x);
        unsigned be = (unsigned)(u >> 52) & 0x7ffu;
        unsigned long long m = u & 0x000fffffffffffffull;
        if (be == 0x7ffu) {
            atomicAdd((unsigned long long *)&acc[NL + (m ? 2 : ((u >> 63) ? 1 : 0))], 1ull);
            return;
        }
        if (be) m |= 0x0010000000000000ull; else be = 1u;
        const unsigned p = be - 1u;
        const unsigned idx = p >> 5, sh = p & 31u;
        const unsigned long long lo = m << sh;
        const unsigned hi = (unsigned)((m >> 32) >> (32u - sh));
        long long c0 = (long long)(lo & 0xffffffffull), c1 = (long long)(lo >> 32), c2 = (long long)hi;
        if (u >> 63) { c0 = -c0; c1 = -c1; c2 = -c2; }
        unsigned long long *q = (unsigned long long *)(acc + idx);
        if (c0) atomicAdd(q, (unsigned long long)c0);
        if (c1) atomicAdd(q + 1, (unsigned long long)c1);
        if (c2) atomicAdd(q + 2, (unsigned long long)c2);
    }
};

// one product p + e = a * b straight into the integer accumulator.  The error term of a non-finite product is
// meaningless (fma(a, b, -inf)) and is dropped; p = +-Inf with e = fma(a, b, -p) = -+Inf (not NaN) means both operands
// were finite and the product overflowed: noted (FLAG_POVER).
template <class Sink>
__device__ __forceinline__ void sink_product(Sink &sink, double p, double e)
{
    sink.add(p);
    if (expo_field(p) != 0x7ffu) {
        if (e != 0.0) sink.add(e);
    } else if (__builtin_isinf(p) && __builtin_isinf(e)) {
        sink.note(FLAG_POVER);
    }
}

// ExDOT: the exact product domain.  Products of two non-zero operands below 2^-968 have bits below 2^-1074 that the
// accumulator cannot hold (their TwoProd error term is not representable); products of two finite operands at or above
// 2^1024 are +-Inf as doubles.  This routine is the range guard of the ExDOT kernels (it replaces fpe_guard there: the
// callers absorb with GUARD = false): one exponent minimum + maximum and ONE wave-uniform vote per tile on the hot path;
// only a tile that holds a tiny, zero, huge (>= 2^1000: the expansion could overflow) or non-finite product, or an a0
// that has reached 2^1000, looks at the operands.  Such a tile is DIVERTED as a whole (returns true: the caller skips
// the expansion for it): every ordinary product goes straight to the integer accumulator (exact); a tiny one is formed
// again at a scaled exponent -- a 2^-ea (in [1/2, 1)) times b 2^(ea + EXT_SHIFT_BITS): both factors and the product are
// normal, so TwoProd is error-free -- and added, exactly, to the LOW accumulator `lo` (global memory, the geometry of
// the main one, unit 2^-(1074 + EXT_SHIFT_BITS)); an overflowed one likewise with b 2^(ea - EXT_SHIFT_BITS) into the
// HIGH accumulator `hi` (unit 2^(-1074 + EXT_SHIFT_BITS)); the finalize kernel folds both back (superacc.hip.h:
// FLAG_PLOW_EXACT / FLAG_PHIGH_EXACT).  Products with an Inf / NaN OPERAND take the ordinary path (IEEE result).
template <int CNT, class Sink, class FA, class FB>
__device__ __forceinline__ bool prod_range_divert(double &a0, const double (&p)[CNT], const double (&e)[CNT], Sink &sink,
                                                  long long *lo, long long *hi, FA &&a_of, FB &&b_of)
{
    unsigned mn = 0x7ffu, mx = expo_field(a0);
#pragma unroll
    for (int j = 0; j < CNT; ++j) {
        const unsigned ex = expo_field(p[j]);
        mn = min(mn, ex);
        mx = max(mx, ex);
    }
    if (__builtin_expect(!__any(mn < LOW_EXPO || mx >= BIG_EXPO), 1)) return false;
    bool hz = mx >= BIG_EXPO;
#pragma unroll
    for (int j = 0; j < CNT; ++j) hz |= expo_field(p[j]) < LOW_EXPO && a_of(j) != 0.0 && b_of(j) != 0.0;
    if (!__any(hz)) return false;   // zeros only
    GlobalSink low{lo}, high{hi};
#pragma unroll   // (unrolled although cold: a rolled loop would index the callers' register arrays dynamically = scratch memory)
    for (int j = 0; j < CNT; ++j) {
        const double a = a_of(j), b = b_of(j);
        const unsigned ex = expo_field(p[j]);
        if (ex < LOW_EXPO && a != 0.0 && b != 0.0) {
            int ea;
            const double a1 = frexp(a, &ea);                       // a = a1 2^ea, 1/2 <= |a1| < 1 (subnormal a included)
            const double b1 = ldexp(b, EXT_SHIFT_BITS + ea);       // normal: a b 2^EXT_SHIFT_BITS lies in [2^-932, 2^249)
            const double P = a1 * b1, E = __builtin_fma(a1, b1, -P);
            low.add(P);
            if (E != 0.0) low.add(E);
            sink.note(FLAG_PUNDER);
        } else if (ex == 0x7ffu && expo_field(a) != 0x7ffu && expo_field(b) != 0x7ffu) {
            int ea;
            const double a1 = frexp(a, &ea);
            const double b1 = ldexp(b, ea - EXT_SHIFT_BITS);       // normal: a b 2^-EXT_SHIFT_BITS lies in [2^-192, 2^832)
            const double P = a1 * b1, E = __builtin_fma(a1, b1, -P);
            high.add(P);
            if (E != 0.0) high.add(E);
            sink.note(FLAG_POVER);
        } else {
            sink_product(sink, p[j], e[j]);
        }
    }
    if (expo_field(a0) >= BIG_EXPO) {   // (as fpe_guard: the head of the expansion never exceeds (1 + CNT) 2^1000)
        sink.add(a0);
        a0 = 0.0;
    }
    return true;
}

// ---- the cascade -----------------------------------------------------------------------------
// Range guard (one wave-uniform test per tile): TwoSum is only error-free while a + x stays finite.
// When any lane of the wave holds an element of magnitude >= 2^1000 (or Inf/NaN), or an a[0] that has reached
// 2^1000, the WHOLE tile of every lane bypasses the expansion and goes straight to the integer accumulator
// (which holds any double exactly, resp. classifies Inf/NaN), and such an a[0] is spilled -- so a[0] never
// exceeds (1 + CNT) * 2^1000 and cannot overflow.  The reference has no such guard (ExSUM.FPE.hpp:408
// "TODO ... Inf/Overflow/NaN").  `e` (optional): the TwoProd error terms that belong to x; the error term of a
// non-finite product is meaningless (fma(a,b,-inf)) and is dropped.
// Diverting the tile as a whole (instead of zeroing only the big elements in place, as this routine first did)
// leaves x / e untouched on the hot path: no merged values after the branch, i.e. none of the register copies
// the in-place form cost (17 v_mov_b64 + 8 repeated v_bfe per 8-element tile in k_gemvT).
// Returns (wave-uniform) true when the tile was diverted.
template <int CNT, class Sink>
__device__ __forceinline__ bool fpe_guard(double &a0, const double (&x)[CNT], const double *e, Sink &sink)
{
    unsigned mx = expo_field(a0);
#pragma unroll
    for (int j = 0; j < CNT; ++j) mx = max(mx, expo_field(x[j]));
    if (__builtin_expect(!__any(mx >= BIG_EXPO), 1)) return false;
#pragma unroll
    for (int j = 0; j < CNT; ++j) {
        if (e) sink_product(sink, x[j], e[j]);
        else sink.add(x[j]);
    }
    if (expo_field(a0) >= BIG_EXPO) {
        sink.add(a0);
        a0 = 0.0;
    }
    return true;
}

// Push CNT elements (a "tile") through expansion levels from..N-1.  One wave-uniform test per level per tile ends the
// cascade as soon as every residue of every lane is zero.  Skipping a level whose incoming terms are all zero is the
// IDENTITY on the expansion (TwoSum(a, 0) = (a, 0)), so the registers evolve bit for bit as in the full cascade: the
// variants WITHOUT early exit take the same shortcut (round 3; before, they ran all N levels on every tile and `fpe = 8`
// streamed at 3.9-4.5 TB/s where `fpe = 8, early_exit` reached 6.7 -- for identical registers).  `early_exit` therefore
// only selects the expansion size (4 / 6 / 8 against N = fpe), as gpu:ExSUM.cpp:72-83 does.  -DEXBLAS_FULL_CASCADE=1
// builds the library with the unconditional N-level cascade for EE = false (A/B measurements).
// Returns (wave-uniform) whether anything had to be spilled to the integer accumulator.
#ifndef EXBLAS_FULL_CASCADE
#define EXBLAS_FULL_CASCADE 0
#endif
template <bool EE>
inline constexpr bool SKIP_ZERO_LEVELS = EE || !EXBLAS_FULL_CASCADE;

template <int N, bool EE, int CNT, class Sink, int ZM = 0>
__device__ __forceinline__ bool fpe_cascade(double (&a)[N > 0 ? N : 1], double (&x)[CNT], int from, Sink &sink)
{
    bool live = true;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        if (i >= from && live) {
#pragma unroll
            for (int j = 0; j < CNT; ++j) {
                double s;
                a[i] = two_sum(a[i], x[j], s);
                x[j] = s;
            }
            if (SKIP_ZERO_LEVELS<EE> && i > from) live = __any(any_nonzero<CNT, ZM>(x));  // wave-uniform
        }
    }
    // what survived every level goes to the integer accumulator (rare: one wave-uniform test first)
    const bool spill = live && __any(any_nonzero<CNT, ZM>(x));
    if (spill) {
#pragma unroll
        for (int j = 0; j < CNT; ++j)
            if (x[j] != 0.0) sink.add(x[j]);
    }
    return spill;
}

template <int N, bool EE, int CNT, class Sink, int ZM = 0>
__device__ __forceinline__ bool fpe_absorb_sink(double (&a)[N > 0 ? N : 1], double (&x)[CNT], int from, Sink &sink)
{
    if constexpr (N == 0) {
#pragma unroll
        for (int j = 0; j < CNT; ++j) sink.add(x[j]);
        return false;
    } else {
        if (from == 0 && fpe_guard<CNT>(a[0], x, nullptr, sink)) return true;
        return fpe_cascade<N, EE, CNT, Sink, ZM>(a, x, from, sink);
    }
}

// Adaptive front-end.  When a tile had to spill (its residues outlived all N levels: the data's exponent range is wider
// than the expansion can hold), the following tiles skip the expansion and go straight to the integer accumulator --
// which on gfx950 streams at ~6 TB/s on its own -- before the expansion is tried again; every spill in a row doubles
// the length of the bypass (BYPASS_MIN .. BYPASS_MAX tiles), a tile the expansion absorbs resets it.  Wide-range inputs
// then cost what the superaccumulator-only variant costs instead of N TwoSum levels PLUS the spill per element;
// well-conditioned inputs never take the branch.  This is a property of the DATA, not of the variant
// (the reference's FPE kernels flush the whole expansion per surviving element instead, ExSUM.FPE.cl:262-292).
// The sum is exact either way, and so are the limbs the parity tests compare.
#ifndef EXBLAS_BYPASS_MIN
#define EXBLAS_BYPASS_MIN 63
#define EXBLAS_BYPASS_MAX 4095
#endif
constexpr int BYPASS_MIN = EXBLAS_BYPASS_MIN, BYPASS_MAX = EXBLAS_BYPASS_MAX;   // (macros: A/B builds, tools/ab_build.sh)

// Measured and dropped (round 3): a word shared by the launches of one reduction that the first
// spilling wave sets and that makes later waves START in bypass mode (wide-range rows 5.5-5.8 -> 4.2 TB/s); other spans
// (7..63, 15..255, 255..4095) and 4 or 16 accumulator copies per wave: within +-3 % (4 copies: -6 %).
struct Bypass {
    int left = 0;    // tiles still to go straight to the integer accumulator (wave-uniform)
    int span = BYPASS_MIN;
    __device__ __forceinline__ void spilled()
    {
        left = span;
        span = min(2 * span + 1, BYPASS_MAX);
    }
};

template <int N, bool EE, int CNT, class Sink, int ZM = 0>
__device__ __forceinline__ void fpe_absorb_adaptive(double (&a)[N > 0 ? N : 1], double (&x)[CNT], Sink &sink, Bypass &bp)
{
    if constexpr (N == 0) {
        fpe_absorb_sink<N, EE, CNT, Sink, ZM>(a, x, 0, sink);
    } else {
        if (__builtin_expect(bp.left > 0, 0)) {  // wave-uniform
            --bp.left;
#pragma unroll
            for (int j = 0; j < CNT; ++j) sink.add(x[j]);
        } else if (fpe_absorb_sink<N, EE, CNT, Sink, ZM>(a, x, 0, sink)) {
            bp.spilled();
        } else {
            bp.span = BYPASS_MIN;
        }
    }
}

// Products: p[j] + e[j] = a_j * b_j exactly (two_prod, NOT the _safe form: the guard handles overflow).
// The rounding errors enter the expansion at slot max(N-3, 0) like ExDOT.FPE.cl:254.
template <int N, bool EE, int CNT, class Sink, int ZM = 0, bool GUARD = true>
__device__ __forceinline__ bool fpe_absorb_prod(double (&a)[N > 0 ? N : 1], double (&p)[CNT], double (&e)[CNT],
                                                Sink &sink)
{
    if constexpr (N == 0) {
#pragma unroll
        for (int j = 0; j < CNT; ++j) {
            sink_product(sink, p[j], e[j]);
        }
        return false;
    } else {
        constexpr int EFROM = (N >= 3) ? N - 3 : 0;
        if constexpr (GUARD) {   // (the ExDOT kernels run prod_range_divert first: GUARD = false)
            if (fpe_guard<CNT>(a[0], p, e, sink)) return true;
        }
        const bool s1 = fpe_cascade<N, EE, CNT, Sink, ZM>(a, p, 0, sink);
        const bool s2 = fpe_cascade<N, EE, CNT, Sink, ZM>(a, e, EFROM, sink);
        return s1 || s2;
    }
}

// adaptive form for products (see fpe_absorb_adaptive)
template <int N, bool EE, int CNT, class Sink, int ZM = 0, bool GUARD = true>
__device__ __forceinline__ void fpe_absorb_prod_adaptive(double (&a)[N > 0 ? N : 1], double (&p)[CNT],
                                                         double (&e)[CNT], Sink &sink, Bypass &bp)
{
    if constexpr (N == 0) {
        fpe_absorb_prod<N, EE, CNT>(a, p, e, sink);
    } else {
        if (__builtin_expect(bp.left > 0, 0)) {
            --bp.left;
#pragma unroll
            for (int j = 0; j < CNT; ++j) {
                sink_product(sink, p[j], e[j]);
            }
        } else if (fpe_absorb_prod<N, EE, CNT, Sink, ZM, GUARD>(a, p, e, sink)) {
            bp.spilled();
        } else {
            bp.span = BYPASS_MIN;
        }
    }
}

template <int N, class Sink>
__device__ __forceinline__ void fpe_flush_sink(double (&a)[N > 0 ? N : 1], Sink &sink)
{
    if constexpr (N > 0) {
#pragma unroll
        for (int i = 0; i < N; ++i) {
            if (a[i] != 0.0) sink.add(a[i]);
            a[i] = 0.0;
        }
    }
}

}  // namespace exb
