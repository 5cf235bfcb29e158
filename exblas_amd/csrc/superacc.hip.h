// superacc.hip.h -- device-side Kulisch superaccumulator for gfx950 (CDNA4).
//
// Replaces, with a different geometry, the reference's OpenCL helpers Accumulate /
// AccumulateWord / Normalize / Round (src/gpu/blas/blas1/ExSUM.Superacc.cl:27-209, twins of
// src/cpu/blas/blas1/superaccumulator.hpp:132-194 and superaccumulator.cpp:80-162).
//
// Geometry (ours, chosen for the GPU): the exact sum is a fixed-point integer with LSB weight
// 2^-1074 (the smallest subnormal), held as NL = 68 signed 64-bit limbs spaced 32 bits apart:
//     value = 2^-1074 * sum_i limb[i] * 2^(32 i)
// A double  +-M * 2^(p-1074)  (M < 2^53, p = max(biased_exp,1)-1 in [0,2045]) is split with pure
// integer shifts into three 32-bit chunks of M << (p & 31) that are added to limbs p>>5 .. +2.
// Each add moves a limb by < 2^32, so a limb survives 2^31 adds: with the API's `int n` no carry
// handling is ever needed while streaming -- the reference's overflow-driven AccumulateWord loop
// and its K = 12 carry-save bits (renormalise every 2^11 adds) disappear.  The range covers every
// finite double incl. subnormals (the reference's 39-limb GPU geometry does not).
// After the last add one carry-propagation pass produces normalised 32-bit digits; those are
// re-cut into the reference's canonical 41 x 52-bit limbs (weight 2^(52(i-21)),
// superaccumulator.cpp:14-22) for parity checks, and rounded once.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace exb {

constexpr int NL = 68;          // limbs, 32-bit spacing
constexpr int CANON = 41;       // reference canonical limbs (21 fraction + 20 exponent words)
constexpr int CANON_DIGITS = 52;
constexpr int CANON_FWORDS = 21;

// layout of the int64 result record every finalize writes
constexpr int OUT_EXACT = 0;    // bits of the correctly rounded double
constexpr int OUT_REFMODE = 1;  // bits of the reference-compatible Round()
constexpr int OUT_FLAGS = 2;    // bit0 +inf seen, bit1 -inf seen, bit2 NaN seen
constexpr int OUT_CANON = 4;    // 41 canonical limbs
constexpr int OUT_DIGITS = 48;  // 68 normalised digits ...
constexpr int OUT_FLAGCNT = 116;  // ... followed by 3 flag indicators (+inf, -inf, NaN) and one pad word:
constexpr int SET_WORDS = 72;   // words [48,120) form one "digit set", summable across ranks as int64
constexpr int OUT_WORDS = 128;

constexpr unsigned FLAG_PINF = 1u, FLAG_NINF = 2u, FLAG_NAN = 4u;
// products (ExDOT): the double-range accumulator holds a 106-bit product only while it neither overflows nor reaches
// below 2^-1074 -- the two limits the reference's kernels share (its MPFR test oracle uses 4196 bits for that reason,
// tests/test.exdot.gpu.cpp:24-46).  PUNDER = some product of two non-zero operands is below 2^-968 in magnitude (it has
// bits below 2^-1074); POVER = a product of two FINITE operands is 2^1024 or more (it overflowed to +-Inf as a double).
constexpr unsigned FLAG_PUNDER = 8u, FLAG_POVER = 16u, FLAG_NONFINITE = 7u;
// PLOW_EXACT / PHIGH_EXACT (set by the finalize kernel together with PUNDER / POVER): the flagged products did NOT lose
// anything -- each was formed exactly at a scaled exponent and accumulated in the LOW / HIGH accumulator (the 68-limb
// geometry of the main one, unit 2^-(1074 + EXT_SHIFT_BITS) resp. 2^(-1074 + EXT_SHIFT_BITS)), which the finalize folded
// back: the high one exactly (a sum that is still beyond the double range after every cancellation yields +-Inf, as
// rounding the exact value does), the low one exactly at or above 2^-1074 and as the half / sticky bits of the rounding
// below.  The result is then the correctly rounded exact dot product -- what mpfr at 4196 bits returns.
constexpr unsigned FLAG_PLOW_EXACT = 32u, FLAG_PHIGH_EXACT = 64u;
// 1216 bits: a b 2^1216 is normal for every a b >= 2^-2148, a b 2^-1216 for every a b in [2^1024, 2^2048)
constexpr int EXT_SHIFT_DIGITS = 38, EXT_SHIFT_BITS = 32 * EXT_SHIFT_DIGITS;
constexpr int LOW_SHIFT_DIGITS = EXT_SHIFT_DIGITS, LOW_SHIFT_BITS = EXT_SHIFT_BITS;
constexpr int EXT_WORDS = 2 * SET_WORDS;   // an exported extension: [low digit set | high digit set]
// per accumulator slot: one 64-byte line for the flag word, then the low and the high accumulator (SET_WORDS int64 each:
// 68 limbs + pad)
constexpr int FLAG_BLOCK_BYTES = 64 + EXT_WORDS * 8;
__device__ __forceinline__ long long *low_acc_of(unsigned *gflags) { return (long long *)((char *)gflags + 64); }
__device__ __forceinline__ long long *high_acc_of(unsigned *gflags) { return low_acc_of(gflags) + SET_WORDS; }

// ---------------------------------------------------------------------------------------------
// error-free transforms (ExSUM.FPE.cl:27-32 KnuthTwoSum; ExDOT.Superacc.cl:25-29 TwoProductFMA)
// compiled with -ffp-contract=off: nothing here may be fused or reassociated
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ double two_sum(double a, double b, double &s)
{
    double r = a + b;
    double z = r - a;
    s = (a - (r - z)) + (b - z);
    return r;
}

__device__ __forceinline__ double two_prod(double a, double b, double &e)
{
    double p = a * b;
    e = __builtin_fma(a, b, -p);
    return p;
}

// As two_prod, but the error term is forced to 0 when the product is not finite (overflow or
// non-finite input): fma(a,b,-inf) would be an opposite infinity and turn a legitimate +-inf
// result into NaN further down.
__device__ __forceinline__ double two_prod_safe(double a, double b, double &e)
{
    double p = a * b;
    double t = __builtin_fma(a, b, -p);
    e = __builtin_isfinite(p) ? t : 0.0;
    return p;
}

// ---------------------------------------------------------------------------------------------
// LDS accumulation: one sub-accumulator column per (wave, lane % COPIES); limb-major layout
// sacc[limb * COPIES + copy] so that the bank of an access depends on the copy only
// ---------------------------------------------------------------------------------------------
template <int COPIES>
__device__ __forceinline__ void lds_add(long long *col, double x, unsigned &flags)
{
    const unsigned long long u = (unsigned long long)__double_as_longlong(x);
    unsigned be = (unsigned)(u >> 52) & 0x7ffu;
    unsigned long long m = u & 0x000fffffffffffffull;
    if (be == 0x7ffu) {  // Inf / NaN never enter the integer accumulator
        flags |= m ? FLAG_NAN : ((u >> 63) ? FLAG_NINF : FLAG_PINF);
        return;
    }
    if (be) m |= 0x0010000000000000ull; else be = 1u;  // subnormals: no hidden bit, exponent 1
    const unsigned p = be - 1u;
    const unsigned idx = p >> 5, sh = p & 31u;
    const unsigned long long lo = m << sh;                    // bits 0..63 of M << sh
    const unsigned hi = (unsigned)((m >> 32) >> (32u - sh));  // bits 64..84 (0 when sh == 0)
    long long c0 = (long long)(lo & 0xffffffffull);
    long long c1 = (long long)(lo >> 32);
    long long c2 = (long long)hi;
    if (u >> 63) { c0 = -c0; c1 = -c1; c2 = -c2; }
    unsigned long long *q = (unsigned long long *)(col + idx * COPIES);
    // no-return LDS atomics (ds_add_u64); zero chunks still issue: cheaper than diverging
    atomicAdd(q, (unsigned long long)c0);
    atomicAdd(q + COPIES, (unsigned long long)c1);
    atomicAdd(q + 2 * COPIES, (unsigned long long)c2);
}

// ---------------------------------------------------------------------------------------------
// final step, run by ONE thread on NL limbs held in LDS/local memory
// ---------------------------------------------------------------------------------------------
// carry-propagate: digits 0..NL-2 end in [0,2^32), the top limb keeps the signed remainder
__device__ inline void normalize_digits(long long *v, int nd = NL)
{
    long long carry = 0;
    for (int i = 0; i < nd - 1; ++i) {
        long long t = v[i] + carry;
        carry = t >> 32;  // arithmetic
        v[i] = t & 0xffffffffll;
    }
    v[nd - 1] += carry;
}

// main digits v[NL] (any signed words) + high digits h[NL] at EXT_SHIFT_DIGITS digits up, exactly.  Returns 0 and leaves
// the normalised sum in v when it fits the main geometry (|sum| < 2^(32 NL - 1) units, far above the double range), else
// +1 / -1 = the sign of a sum that is out of range whatever the rounding.  c: NL + EXT_SHIFT_DIGITS words of scratch.
__device__ inline int fold_high_digits(long long *v, const long long *h, long long *c)
{
    constexpr int ND = NL + EXT_SHIFT_DIGITS;
    for (int i = 0; i < ND; ++i) c[i] = (i < NL ? v[i] : 0) + (i >= EXT_SHIFT_DIGITS ? h[i - EXT_SHIFT_DIGITS] : 0);
    normalize_digits(c, ND);
    bool zeros = c[ND - 1] == 0, ones = c[ND - 1] == -1;
    for (int i = NL; i < ND - 1; ++i) {
        zeros = zeros && c[i] == 0;
        ones = ones && c[i] == 0xffffffffll;
    }
    // (a negative value whose digit NL-1 has its top bit clear does not fit either: the signed top digit must hold it)
    if (ones && !((c[NL - 1] >> 31) & 1)) ones = false;
    if (zeros && ((c[NL - 1] >> 31) & 1)) zeros = false;
    if (!zeros && !ones) return c[ND - 1] < 0 ? -1 : 1;
    for (int i = 0; i < NL - 1; ++i) v[i] = c[i];
    v[NL - 1] = zeros ? c[NL - 1] : c[NL - 1] - 0x100000000ll;
    return 0;
}

// correctly rounded (RN-even) double of the normalised digits; returns the bit pattern
__device__ inline unsigned long long round_exact_bits(const long long *v)
{
    const bool neg = v[NL - 1] < 0;
    // magnitude digits
    unsigned mag[NL];
    if (!neg) {
        for (int i = 0; i < NL; ++i) mag[i] = (unsigned)v[i];
    } else {
        unsigned long long c = 1;
        for (int i = 0; i < NL; ++i) {
            unsigned long long t = (unsigned long long)(~(unsigned)v[i]) + c;
            mag[i] = (unsigned)t;
            c = t >> 32;
        }
    }
    int t = NL - 1;
    while (t >= 0 && mag[t] == 0) --t;
    const unsigned long long sign = neg ? 0x8000000000000000ull : 0ull;
    if (t < 0) return 0ull;  // exact zero -> +0.0
    const int lz = __builtin_clz(mag[t]);
    const int msb = 32 * t + 31 - lz;
    if (msb <= 52) {
        // below 2^53 units of 2^-1074: exactly representable (subnormal or first binades)
        unsigned long long val = ((unsigned long long)(t >= 1 ? mag[1] : 0u) << 32) | mag[0];
        return sign | val;
    }
    const unsigned d1 = (t >= 1) ? mag[t - 1] : 0u, d2 = (t >= 2) ? mag[t - 2] : 0u;
    unsigned long long w = ((unsigned long long)mag[t] << 32) | d1;  // leading one at bit 63-lz
    unsigned rest;  // bits of d2 not consumed by the window
    if (lz) {
        w = (w << lz) | (unsigned long long)(d2 >> (32 - lz));
        rest = d2 << lz;
    } else {
        rest = d2;
    }
    bool sticky = (w & 0x3ffull) != 0 || rest != 0;
    for (int i = t - 3; i >= 0 && !sticky; --i) sticky = mag[i] != 0;
    unsigned long long bits = ((unsigned long long)(msb - 52) << 52) + (w >> 11);
    const bool rnd = (w >> 10) & 1ull;
    if (rnd && (sticky || (bits & 1ull))) bits += 1;       // carries into the exponent naturally
    if ((bits >> 52) >= 0x7ffull) bits = 0x7ff0000000000000ull;  // overflow -> inf
    return sign | bits;
}

// The same for value = (normalised digits) + f units, 0 <= f < 1, f known by its first bit (half) and whether anything
// follows it (sticky): the sub-LSB part of a reduction whose low accumulator held something (PLOW_EXACT).
__device__ inline unsigned long long round_exact_bits_frac(const long long *v, bool half, bool sticky)
{
    if (!half && !sticky) return round_exact_bits(v);
    const bool neg = v[NL - 1] < 0;
    unsigned mag[NL];
    if (!neg) {
        for (int i = 0; i < NL; ++i) mag[i] = (unsigned)v[i];
    } else {
        // |H + f| = (|H| - 1) + (1 - f): magnitude digits |H| - 1 = ~H (no +1), fraction 1 - f
        for (int i = 0; i < NL; ++i) mag[i] = ~(unsigned)v[i];
        const bool h = half, st = sticky;
        half = !(h && st);      // 1 - f >= 1/2  <=>  f <= 1/2
        sticky = !(h && !st);   // 1 - f != 1/2 exactly (and it is never 0)
    }
    const unsigned long long sign = neg ? 0x8000000000000000ull : 0ull;
    int t = NL - 1;
    while (t >= 0 && mag[t] == 0) --t;
    const int lz = t >= 0 ? __builtin_clz(mag[t]) : 0;
    const int msb = t >= 0 ? 32 * t + 31 - lz : -1;
    if (msb <= 52) {
        // spacing of the doubles here is one unit (2^-1074): round at the unit position
        unsigned long long val = t < 0 ? 0ull : (((unsigned long long)(t >= 1 ? mag[1] : 0u) << 32) | mag[0]);
        if (half && (sticky || (val & 1ull))) val += 1;   // 2^53 units: the bit pattern carries into the exponent naturally
        return sign | val;
    }
    const unsigned d1 = (t >= 1) ? mag[t - 1] : 0u, d2 = (t >= 2) ? mag[t - 2] : 0u;
    unsigned long long w = ((unsigned long long)mag[t] << 32) | d1;
    unsigned rest;
    if (lz) {
        w = (w << lz) | (unsigned long long)(d2 >> (32 - lz));
        rest = d2 << lz;
    } else {
        rest = d2;
    }
    bool st2 = true;   // the fraction is non-zero: everything below the rounding position is sticky
    (void)rest;
    unsigned long long bits = ((unsigned long long)(msb - 52) << 52) + (w >> 11);
    const bool rnd = (w >> 10) & 1ull;
    if (rnd && (st2 || (bits & 1ull))) bits += 1;
    if ((bits >> 52) >= 0x7ffull) bits = 0x7ff0000000000000ull;
    return sign | bits;
}

// two's-complement bit field [o, o+52) of the value held in the normalised digits
__device__ inline long long digits_field52(const long long *v, int o)
{
    // 128-bit window starting at digit q = floor(o/32)
    const int q = o >> 5, r = o & 31;  // arithmetic shift: o may be negative
    auto dig = [&](int i) -> unsigned long long {
        if (i < 0) return 0ull;
        if (i >= NL - 1) {
            // top limb is a signed 64-bit remainder: expand it into 32-bit sign-extended digits
            const long long top = v[NL - 1];
            const int k = i - (NL - 1);
            if (k == 0) return (unsigned long long)(unsigned)top;
            if (k == 1) return (unsigned long long)(unsigned)(top >> 32);
            return (top < 0) ? 0xffffffffull : 0ull;
        }
        return (unsigned long long)(unsigned)v[i];
    };
    const unsigned long long lo = dig(q) | (dig(q + 1) << 32);
    const unsigned long long hi = dig(q + 2) | (dig(q + 3) << 32);
    const unsigned long long win = r ? ((lo >> r) | (hi << (64 - r))) : lo;
    return (long long)(win & ((1ull << CANON_DIGITS) - 1));
}

// canonical 41 x 52-bit limbs of the reference (limb i weighs 2^(52(i-21)), so the LSB of limb 0
// is 2^-1092 = our LSB shifted left by 18 bits); all limbs in [0,2^52) but the signed top one
__device__ inline void digits_to_canon(const long long *v, long long *canon)
{
    for (int j = 0; j < CANON - 1; ++j) canon[j] = digits_field52(v, CANON_DIGITS * j - 18);
    // top limb: everything from bit 52*40-18 up, sign-extended
    const int o = CANON_DIGITS * (CANON - 1) - 18;
    const int q = o >> 5, r = o & 31;
    auto dig = [&](int i) -> unsigned long long {
        if (i >= NL - 1) {
            const long long top = v[NL - 1];
            const int k = i - (NL - 1);
            if (k == 0) return (unsigned long long)(unsigned)top;
            if (k == 1) return (unsigned long long)(unsigned)(top >> 32);
            return (top < 0) ? 0xffffffffull : 0ull;
        }
        return (unsigned long long)(unsigned)v[i];
    };
    const unsigned long long lo = dig(q) | (dig(q + 1) << 32);
    const unsigned long long hi = dig(q + 2) | (dig(q + 3) << 32);
    canon[CANON - 1] = (long long)(r ? ((lo >> r) | (hi << (64 - r))) : lo);
}

// Superaccumulator::Round restated for the device on canonical limbs, INCLUDING its defect
// (superaccumulator.cpp:80-134; identical code in ExSUM.Superacc.cl:100-143): the
// "reference" rounding mode.  Input limbs are already normalised.
__device__ inline double round_reference(const long long *acc)
{
    const long long mask = (1ll << CANON_DIGITS) - 1;
    const bool negative = acc[CANON - 1] < 0;
    int i;
    for (i = CANON - 1; i >= 0 && acc[i] == 0; --i) { }
    if (negative) {
        for (; i >= 0 && (acc[i] & mask) == mask; --i) { }
    }
    if (i < 0) return 0.0;
    long long hiword = negative ? mask - acc[i] : acc[i];
    double rounded = (double)hiword;
    double hi = ldexp(rounded, (i - CANON_FWORDS) * CANON_DIGITS);
    if (i == 0) return negative ? -hi : hi;
    hiword -= __double2ll_rn(rounded);
    double mid = ldexp((double)hiword, (i - CANON_FWORDS) * CANON_DIGITS);
    long long sticky = 0;
    for (int j = 0; j != i - 1; ++j) sticky |= negative ? (1ll << CANON_DIGITS) - acc[j] : acc[j];
    long long loword = negative ? (1ll << CANON_DIGITS) - acc[i - 1] : acc[i - 1];
    loword |= (sticky != 0);
    double lo = ldexp((double)loword, (i - 1 - CANON_FWORDS) * CANON_DIGITS);
    if (mid != 0) {
        // OddRoundSumNonnegative (mylibm.hpp:156-171)
        unsigned long long b = (unsigned long long)__double_as_longlong(mid + lo);
        b |= (unsigned long long)(lo != 0.0);
        lo = __longlong_as_double((long long)b);
    }
    hi = hi + lo;
    return negative ? -hi : hi;
}

// One thread: v[NL] raw limbs -> record (normalised digits, canonical limbs, both roundings)
__device__ inline void finish_record(long long *v, unsigned flags, long long *out)
{
    normalize_digits(v);
    long long canon[CANON];
    digits_to_canon(v, canon);
    unsigned long long ex = round_exact_bits(v);
    double rf = round_reference(canon);
    if (flags & FLAG_NONFINITE) {
        // IEEE semantics for non-finite inputs: NaN, or opposite infinities -> NaN; else +-inf
        const bool nan = (flags & FLAG_NAN) || ((flags & FLAG_PINF) && (flags & FLAG_NINF));
        ex = nan ? 0x7ff8000000000000ull
                 : ((flags & FLAG_NINF) ? 0xfff0000000000000ull : 0x7ff0000000000000ull);
        rf = __longlong_as_double((long long)ex);
    }
    out[OUT_EXACT] = (long long)ex;
    out[OUT_REFMODE] = __double_as_longlong(rf);
    out[OUT_FLAGS] = (long long)flags;
    out[3] = 0;
    for (int j = 0; j < CANON; ++j) out[OUT_CANON + j] = canon[j];
    for (int j = 0; j < NL; ++j) out[OUT_DIGITS + j] = v[j];
    out[OUT_FLAGCNT + 0] = (flags & FLAG_PINF) ? 1 : 0;
    out[OUT_FLAGCNT + 1] = (flags & FLAG_NINF) ? 1 : 0;
    out[OUT_FLAGCNT + 2] = (flags & FLAG_NAN) ? 1 : 0;
    out[OUT_FLAGCNT + 3] = ((flags & FLAG_PUNDER) ? 1 : 0) + ((flags & FLAG_POVER) ? 65536 : 0);  // sums over ranks stay apart
}

// ---------------------------------------------------------------------------------------------
// The final step inside ONE wavefront, registers only: lane l holds limb l (v0) and, for l < 4, limb 64+l (v1).
// Carries travel one lane per pass (DPP wave_shr:1; a wave-wide vote ends the loop), the leading / lowest non-zero
// digits come from __ballot masks, canonical limbs are cut with per-lane __shfl gathers.  No LDS, no barrier.
// All 64 lanes must call it.  Same bits as the single-thread finish_record above (the scalar ExGEMM kernel still
// uses that one; the parity tests run both against the oracle).
// ---------------------------------------------------------------------------------------------
// cross-lane moves that stay in the VALU / SALU (no LDS crossbar round trip as with __shfl):
// wave_shr1: lane l receives lane l-1's value, lane 0 receives 0 (DPP wave_shr:1, GFX9 family);
// lane_bcast: every lane receives lane `l`'s value, l wave-uniform (v_readlane)
__device__ __forceinline__ long long wave_shr1(long long v)
{
    int lo = (int)v, hi = (int)(v >> 32);
    lo = __builtin_amdgcn_update_dpp(0, lo, 0x138, 0xf, 0xf, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, 0x138, 0xf, 0xf, true);
    return (long long)(((unsigned long long)(unsigned)hi << 32) | (unsigned long long)(unsigned)lo);
}
__device__ __forceinline__ unsigned lane_bcast(unsigned v, int l)
{
    return (unsigned)__builtin_amdgcn_readlane((int)v, l);
}
__device__ __forceinline__ long long lane_bcast(long long v, int l)
{
    const unsigned lo = lane_bcast((unsigned)v, l), hi = lane_bcast((unsigned)(v >> 32), l);
    return (long long)(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ double lane_bcast(double v, int l)
{
    return __longlong_as_double(lane_bcast(__double_as_longlong(v), l));
}

// Adds the wave-uniform double x to an accumulator held across the wave (lane l: limb l in v0, lanes 0..3: limbs
// 64..67 in v1) -- the register twin of lds_add.
__device__ __forceinline__ void wave_add_double(long long &v0, long long &v1, double x, unsigned &flags)
{
    const int lane = (int)(threadIdx.x & 63u);
    const unsigned long long u = (unsigned long long)__double_as_longlong(x);
    unsigned be = (unsigned)(u >> 52) & 0x7ffu;
    unsigned long long m = u & 0x000fffffffffffffull;
    if (be == 0x7ffu) {
        flags |= m ? FLAG_NAN : ((u >> 63) ? FLAG_NINF : FLAG_PINF);
        return;
    }
    if (be) m |= 0x0010000000000000ull; else be = 1u;
    const unsigned p = be - 1u;
    const int idx = (int)(p >> 5);
    const unsigned sh = p & 31u;
    const unsigned long long lo = m << sh;
    const unsigned hi = (unsigned)((m >> 32) >> (32u - sh));
    long long c0 = (long long)(lo & 0xffffffffull), c1 = (long long)(lo >> 32), c2 = (long long)hi;
    if (u >> 63) { c0 = -c0; c1 = -c1; c2 = -c2; }
    const int k0 = lane - idx;        // limb `lane` takes chunk k0 when 0 <= k0 <= 2
    const int k1 = lane + 64 - idx;   // limb 64 + lane
    v0 += k0 == 0 ? c0 : (k0 == 1 ? c1 : (k0 == 2 ? c2 : 0ll));
    if (lane < NL - 64) v1 += k1 == 0 ? c0 : (k1 == 1 ? c1 : (k1 == 2 ? c2 : 0ll));
}

struct WaveFinish {
    unsigned long long ex;  // bits of the correctly rounded double (uniform)
    double rf;              // reference-compatible rounding (uniform)
    long long canon;        // lane j < 41: canonical limb j
    long long d0, d1;       // normalised digits: limb lane, and limb 64+lane for lane < 4
};

template <bool FULL = true>
__device__ inline WaveFinish finish_wave(long long v0, long long v1, const unsigned flags)
{
    const int lane = (int)(threadIdx.x & 63u);
    constexpr int HI = NL - 64;  // limbs kept in the second register (4); limb NL-1 is the signed top limb
    if (lane >= HI) v1 = 0;
    // ---- carry propagation: one lane per pass, confined to the limbs that hold something ----
    // Limbs 0..hi (hi = highest non-zero limb) are cut to 32 bits; limb hi+1 only collects what leaves limb hi
    // (|.| < 2^32) and is sign-extended to the top afterwards.  Without this a negative total would need one pass
    // per empty limb above it for the borrow to reach the top limb.
    const unsigned long long nzA = __ballot(v0 != 0), nzB = __ballot(v1 != 0);
    const int hi = nzB ? 64 + (63 - __builtin_clzll(nzB)) : (nzA ? 63 - __builtin_clzll(nzA) : -1);
    const bool split0 = lane <= hi;
    const bool split1 = (64 + lane <= hi) && (lane < HI - 1);  // the top limb is never cut
    for (int pass = 0; pass < 2 * NL && hi >= 0; ++pass) {
        const long long c0 = split0 ? (v0 >> 32) : 0, lo0 = split0 ? (v0 & 0xffffffffll) : v0;
        const long long c1 = split1 ? (v1 >> 32) : 0, lo1 = split1 ? (v1 & 0xffffffffll) : v1;
        long long in0 = wave_shr1(c0), in1 = wave_shr1(c1);
        const long long c63 = lane_bcast(c0, 63);
        if (lane == 0) in1 = c63;
        v0 = lo0 + in0;
        v1 = (lane < HI) ? lo1 + in1 : 0;
        const bool pending = (split0 && (v0 >> 32) != 0) || (split1 && (v1 >> 32) != 0);
        if (!__any(pending)) break;
    }
    if (hi >= 0 && hi + 1 < NL - 1) {  // sign-extend limb hi+1 through the empty limbs up to the top one
        const int e = hi + 1;
        const long long s0 = lane_bcast(v0, e & 63), s1 = lane_bcast(v1, (e - 64) & 63);
        const long long sv = e < 64 ? s0 : s1;
        const long long fill = sv < 0 ? 0xffffffffll : 0ll;
        if (lane == e) v0 = sv & 0xffffffffll;
        if (lane > e) v0 = fill;
        if (lane < HI) {
            const int i1 = 64 + lane;
            if (i1 == e) v1 = sv & 0xffffffffll;
            if (i1 > e) v1 = (i1 == NL - 1) ? (sv < 0 ? -1ll : 0ll) : fill;
        }
    }
    WaveFinish r;
    r.d0 = v0;
    r.d1 = v1;
    const long long top = lane_bcast(v1, HI - 1);
    const bool neg = top < 0;
    const unsigned d0 = (unsigned)v0, d1 = (unsigned)v1;
    // ---- magnitude digits ----
    const unsigned long long nz0 = __ballot(d0 != 0), nz1 = __ballot(lane < HI && d1 != 0);
    const int z = nz0 ? __builtin_ctzll(nz0) : (nz1 ? 64 + __builtin_ctzll(nz1) : NL);
    const int i1 = 64 + lane;
    const unsigned m0 = !neg ? d0 : (lane < z ? 0u : (lane == z ? (0u - d0) : ~d0));
    const unsigned m1 = (lane >= HI) ? 0u : (!neg ? d1 : (i1 < z ? 0u : (i1 == z ? (0u - d1) : ~d1)));
    const unsigned long long mz0 = __ballot(m0 != 0), mz1 = __ballot(m1 != 0);
    const int tp = mz1 ? 64 + (63 - __builtin_clzll(mz1)) : (mz0 ? 63 - __builtin_clzll(mz0) : -1);
    auto mag = [&](int i) -> unsigned {  // i uniform
        const unsigned a = lane_bcast(m0, i & 63), b = lane_bcast(m1, (i - 64) & 63);
        return i < 0 ? 0u : (i < 64 ? a : b);
    };
    unsigned long long ex = 0ull;
    const unsigned long long sign = neg ? 0x8000000000000000ull : 0ull;
    {
        const unsigned mt = mag(tp < 0 ? 0 : tp), ma = mag(tp - 1), mb = mag(tp - 2), q0 = mag(0), q1 = mag(1);
        if (tp >= 0) {
            const int lz = __builtin_clz(mt | 1u);  // mt != 0 here; the OR only keeps the builtin defined
            const int msb = 32 * tp + 31 - lz;
            if (msb <= 52) {
                ex = sign | (((unsigned long long)(tp >= 1 ? q1 : 0u) << 32) | q0);
            } else {
                unsigned long long w = ((unsigned long long)mt << 32) | ma;
                unsigned rest = mb;
                if (lz) {
                    w = (w << lz) | (unsigned long long)(mb >> (32 - lz));
                    rest = mb << lz;
                }
                const bool sticky = (w & 0x3ffull) != 0 || rest != 0 || z < tp - 2;
                unsigned long long bits = ((unsigned long long)(msb - 52) << 52) + (w >> 11);
                if (((w >> 10) & 1ull) && (sticky || (bits & 1ull))) bits += 1;
                if ((bits >> 52) >= 0x7ffull) bits = 0x7ff0000000000000ull;
                ex = sign | bits;
            }
        }
    }
    double rf = 0.0;
    r.canon = 0;
    if constexpr (FULL) {
    // ---- canonical limbs: lane j cuts bits [52j-18, 52j+34) of the two's-complement digit string ----
    auto digit = [&](int i) -> unsigned long long {  // i per lane; digits beyond the top limb are its sign extension
        const unsigned a = __shfl(d0, i & 63), b = __shfl(d1, (i - 64) & 63);
        const int k = i - (NL - 1);
        unsigned v = (i < 64) ? a : b;
        if (k == 0) v = (unsigned)top;
        if (k == 1) v = (unsigned)(top >> 32);
        if (k >= 2) v = neg ? 0xffffffffu : 0u;
        return i < 0 ? 0ull : (unsigned long long)v;
    };
    {
        const int j = lane < CANON ? lane : 0;
        const int o = CANON_DIGITS * j - 18;
        const int q = o >> 5, sh = o & 31;
        const unsigned long long lo = digit(q) | (digit(q + 1) << 32), hi = digit(q + 2) | (digit(q + 3) << 32);
        const unsigned long long win = sh ? ((lo >> sh) | (hi << (64 - sh))) : lo;
        r.canon = (j == CANON - 1) ? (long long)win : (long long)(win & ((1ull << CANON_DIGITS) - 1));
    }
    // ---- reference-compatible rounding (superaccumulator.cpp:80-134) ----
    {
        const long long mask = (1ll << CANON_DIGITS) - 1;
        const long long cj = r.canon;
        const bool in = lane < CANON;
        const unsigned long long lead = __ballot(in && (neg ? ((cj & mask) != mask) : (cj != 0)));
        const unsigned long long nzc = __ballot(in && cj != 0);
        const int i = lead ? 63 - __builtin_clzll(lead) : -1;
        const int cz = nzc ? __builtin_ctzll(nzc) : CANON;
        const long long ci = __shfl(cj, i < 0 ? 0 : i), cl = __shfl(cj, i < 1 ? 0 : i - 1);
        if (i >= 0) {
            long long hiword = neg ? mask - ci : ci;
            const double rounded = (double)hiword;
            double hi = ldexp(rounded, (i - CANON_FWORDS) * CANON_DIGITS);
            if (i == 0) {
                rf = neg ? -hi : hi;
            } else {
                hiword -= __double2ll_rn(rounded);
                const double mid = ldexp((double)hiword, (i - CANON_FWORDS) * CANON_DIGITS);
                const bool sticky = neg ? (i >= 2) : (cz < i - 1);
                long long loword = neg ? (1ll << CANON_DIGITS) - cl : cl;
                loword |= (long long)sticky;
                double lo = ldexp((double)loword, (i - 1 - CANON_FWORDS) * CANON_DIGITS);
                if (mid != 0) {
                    unsigned long long b = (unsigned long long)__double_as_longlong(mid + lo);
                    b |= (unsigned long long)(lo != 0.0);
                    lo = __longlong_as_double((long long)b);
                }
                hi = hi + lo;
                rf = neg ? -hi : hi;
            }
        }
    }
    }
    if (flags & FLAG_NONFINITE) {
        const bool nan = (flags & FLAG_NAN) || ((flags & FLAG_PINF) && (flags & FLAG_NINF));
        ex = nan ? 0x7ff8000000000000ull : ((flags & FLAG_NINF) ? 0xfff0000000000000ull : 0x7ff0000000000000ull);
        rf = __longlong_as_double((long long)ex);
    }
    r.ex = ex;
    r.rf = rf;
    return r;
}

// one wave writes the full record
__device__ inline void write_record_wave(const WaveFinish &r, unsigned flags, long long *out)
{
    const int lane = (int)(threadIdx.x & 63u);
    if (lane == 0) {
        out[OUT_EXACT] = (long long)r.ex;
        out[OUT_REFMODE] = __double_as_longlong(r.rf);
        out[OUT_FLAGS] = (long long)flags;
        out[3] = 0;
        out[OUT_FLAGCNT + 0] = (flags & FLAG_PINF) ? 1 : 0;
        out[OUT_FLAGCNT + 1] = (flags & FLAG_NINF) ? 1 : 0;
        out[OUT_FLAGCNT + 2] = (flags & FLAG_NAN) ? 1 : 0;
        out[OUT_FLAGCNT + 3] = ((flags & FLAG_PUNDER) ? 1 : 0) + ((flags & FLAG_POVER) ? 65536 : 0);  // sums over ranks stay apart
    }
    if (lane < CANON) out[OUT_CANON + lane] = r.canon;
    out[OUT_DIGITS + lane] = r.d0;
    if (lane < NL - 64) out[OUT_DIGITS + 64 + lane] = r.d1;
}

}  // namespace exb
