// gemm_fixed.hip.h -- pieces shared by the two int8 matrix-core ExGEMM paths (blas3_i8.hip: base-256 digit slices;
// blas3_crt.hip: residues modulo pairwise coprime 8-bit moduli): the info block, the tile-major int8 plane layout, the
// exact double -> fixed-point conversion, wide two's-complement integers with both roundings, the XCD-aware tile order.
#pragma once
#include "superacc.hip.h"
#include "exblas_internal.h"
#include "gemm_scan.hip.h"

#include <type_traits>

namespace exb {

typedef int v4i_t __attribute__((ext_vector_type(4)));
typedef int v16i_t __attribute__((ext_vector_type(16)));

constexpr int I8_T = 64;         // tile edge: rows (columns) per tile, and k bytes per chunk
constexpr int I8_TILE = I8_T * I8_T;  // bytes of one plane of one tile
constexpr int I8_SMAX = 8;       // digits per operand and pass of the generic body and of multi-block operands
constexpr int I8_XMAX = 9;       // digits per operand and pass the unrolled bodies go up to (17 int32 accumulator
                                 // tiles = 272 AGPRs per wave + 2 x 18 fragments; 2 x 18 planes x 4 KiB = 144 KiB of LDS;
                                 // 10 digits would need 304 + 160 registers: it spills)
constexpr int I8_SCAP = 16;      // digits per operand the path supports (2 x 2 passes)
constexpr int I8_KPASS = 8192;   // k per pass: 8192 * 8 pairs * 2^14 = 2^30 < 2^31
constexpr int I8_NWG = 5;        // 64-bit words of the per-entry accumulator in global memory (multi-pass)
constexpr int I8_ERANGE = 300;   // |exponent| bound of the operands: keeps every rounded result a normal double

enum { INFO_SA = 5, INFO_SB = 6, INFO_PATH = 7, INFO_BS = 8, INFO_EXACT = 9,     // extends the scan's info block
       INFO_CRT_L = 10, INFO_CRT_NA = 11, INFO_CRT_NB = 12 };                    // residue path (blas3_crt.hip)
static_assert(INFO_PATH == I8_INFO_PATH, "exblas_internal.h");
enum { PATH_SCALAR = 0, PATH_I8 = 2, PATH_CRT = 4 };

template <int B, int E, class F>
__device__ __forceinline__ void static_for_i8(F &&f)
{
    if constexpr (B < E) {
        f(std::integral_constant<int, B>{});
        static_for_i8<B + 1, E>(f);
    }
}

// byte offset of (row r, k byte kb) inside a 4 KiB tile plane: rows of 64 bytes, the four 16-byte chunks of a row
// permuted by (r >> 2) & 3 -- with that, the 16 lanes ds_read_b128 serves together (rows {0-3, 12-15, 20-27} + 32h of
// one chunk column) fall into 16 different 16-byte bank groups
__device__ __forceinline__ int tile_off(int r, int kb) { return r * I8_T + ((((kb >> 4) ^ (r >> 2)) & 3) << 4) + (kb & 15); }

// 128-bit two's-complement integer in two words
struct I128 {
    unsigned long long lo;
    long long hi;
};

// X = x / 2^u as an exact integer (the scan guarantees x is a multiple of 2^u and |X| < 2^(8s-2) <= 2^126)
__device__ __forceinline__ I128 to_fixed(double x, int u)
{
    const unsigned long long bits = (unsigned long long)__double_as_longlong(x);
    const int be = (int)((bits >> 52) & 0x7ffu);
    I128 r{0ull, 0ll};
    if (be == 0) return r;  // zero (subnormals never reach this path)
    const unsigned long long m = (bits & 0x000fffffffffffffull) | 0x0010000000000000ull;
    const int sh = be - 1023 - 52 - u;  // >= -52: the bits shifted out below are zero by construction
    if (sh >= 64) {
        r.lo = 0;
        r.hi = (long long)(m << (sh - 64));
    } else if (sh > 0) {
        r.lo = m << sh;
        r.hi = (long long)(m >> (64 - sh));
    } else {
        r.lo = m >> (-sh);
    }
    if (bits >> 63) {  // negate
        r.lo = ~r.lo + 1ull;
        r.hi = ~r.hi + (r.lo == 0 ? 1 : 0);
    }
    return r;
}

// ---------------------------------------------------------------------------------------------
// wide two's-complement integers
// ---------------------------------------------------------------------------------------------
// acc += sign_extend(T) << SH   (SH compile-time)
template <int NW, int SH>
__device__ __forceinline__ void wide_add_c(unsigned long long (&acc)[NW], long long T)
{
    constexpr int w = SH >> 6, b = SH & 63;
    const unsigned long long ext = (unsigned long long)(T >> 63);
    const unsigned long long lo = (unsigned long long)T << b;
    const unsigned long long hi = b ? (unsigned long long)(T >> (64 - b)) : ext;
    unsigned long long c = 0, cn;
#pragma unroll
    for (int i = 0; i < NW; ++i) {
        const unsigned long long v = (i < w) ? 0ull : (i == w ? lo : (i == w + 1 ? hi : ext));
        acc[i] = __builtin_addcll(acc[i], v, c, &cn);
        c = cn;
    }
}

// dst (ND words) += sign_extend(src (3 words)) << sh, sh >= 0 a runtime multiple of 8
template <int ND>
__device__ __forceinline__ void wide_add_v(unsigned long long (&dst)[ND], const unsigned long long (&src)[3], int sh)
{
    const int w = sh >> 6, b = sh & 63;
    const unsigned long long ext = (unsigned long long)((long long)src[2] >> 63);
    unsigned long long c = 0, cn;
#pragma unroll
    for (int i = 0; i < ND; ++i) {
        // word i of (src << sh): bits from src word i-w (<< b) and src word i-w-1 (>> 64-b)
        auto word = [&](int j) -> unsigned long long { return j < 0 ? 0ull : (j < 3 ? src[j] : ext); };
        const int j = i - w;
        unsigned long long v = word(j) << b;
        if (b) v |= word(j - 1) >> (64 - b);
        if (j < 0) v = 0;
        dst[i] = __builtin_addcll(dst[i], v, c, &cn);
        c = cn;
    }
}

// round the NW-word two's-complement integer times 2^unit_exp to nearest-even; the caller guarantees a normal result
template <int NW>
__device__ inline double wide_round_n(const unsigned long long (&in)[NW], int unit_exp)
{
    unsigned long long m[NW];
    const bool neg = (long long)in[NW - 1] < 0;
    {
        unsigned long long c = neg ? 1 : 0, cn;
#pragma unroll
        for (int i = 0; i < NW; ++i) {
            m[i] = neg ? __builtin_addcll(~in[i], 0ull, c, &cn) : in[i];
            c = neg ? cn : 0;
        }
    }
    int top = NW - 1;
    while (top > 0 && m[top] == 0) --top;
    if (m[top] == 0) return 0.0;
    const int lz = __builtin_clzll(m[top]);
    const int msb = 64 * top + 63 - lz;
    double r;
    if (msb <= 52) {
        r = ldexp((double)m[0], unit_exp);
    } else {
        const unsigned long long below = top > 0 ? m[top - 1] : 0ull;
        const unsigned long long w = lz ? ((m[top] << lz) | (below >> (64 - lz))) : m[top];
        bool sticky = (w & 0x3ffull) != 0 || (lz ? (below << lz) != 0 : below != 0);
        for (int i = top - 2; i >= 0; --i) sticky |= m[i] != 0;
        unsigned long long mant = w >> 11;
        if (((w >> 10) & 1ull) && (sticky || (mant & 1ull))) mant += 1;  // may reach 2^53: still exact in fp64
        r = ldexp((double)mant, msb - 52 + unit_exp);
    }
    return neg ? -r : r;
}

// The same rounding without dynamically indexed arrays (wide_round_n's m[top] puts its words in scratch memory): the
// leading word, the word below it and the sticky OR of the rest are picked by select chains over the unrolled words.
template <int NW>
__device__ __forceinline__ double wide_round_sel(const unsigned long long (&in)[NW], int unit_exp)
{
    unsigned long long m[NW];
    const bool neg = (long long)in[NW - 1] < 0;
    {
        unsigned long long c = neg ? 1 : 0, cn;
#pragma unroll
        for (int i = 0; i < NW; ++i) {
            m[i] = neg ? __builtin_addcll(~in[i], 0ull, c, &cn) : in[i];
            c = neg ? cn : 0;
        }
    }
    unsigned long long hi = 0, below = 0;
    int top = 0;
    bool f1 = false, f2 = false, rest = false;  // f1: leading word found in an earlier step; f2: at least two steps earlier
#pragma unroll
    for (int i = NW - 1; i >= 0; --i) {
        const bool nz = m[i] != 0;
        const bool take = !f1 && nz;
        hi = take ? m[i] : hi;
        below = take ? (i > 0 ? m[i - 1] : 0ull) : below;
        top = take ? i : top;
        rest = rest || (f2 && nz);
        f2 = f1;
        f1 = f1 || take;
    }
    if (hi == 0) return 0.0;
    const int lz = __builtin_clzll(hi);
    const int msb = 64 * top + 63 - lz;
    double r;
    if (msb <= 52) {
        r = ldexp((double)m[0], unit_exp);
    } else {
        const unsigned long long w = lz ? ((hi << lz) | (below >> (64 - lz))) : hi;
        const bool sticky = (w & 0x3ffull) != 0 || (lz ? (below << lz) != 0 : below != 0) || rest;
        unsigned long long mant = w >> 11;
        if (((w >> 10) & 1ull) && (sticky || (mant & 1ull))) mant += 1;  // may reach 2^53: still exact in fp64
        r = ldexp((double)mant, msb - 52 + unit_exp);
    }
    return neg ? -r : r;
}

// The reference's rounding on the same integer: cut value = W * 2^unit_exp into the canonical 41 x 52-bit limbs
// (limb j = bits [52(j-21), 52(j-20)) of the value, the top limb signed; superaccumulator.cpp:14-22) and run its
// Round() (round_reference, superacc.hip.h).  Bits of the value below 2^-1092 cannot occur on this path (|e| <= 300).
template <int NW>
__device__ inline double wide_round_reference(const unsigned long long (&in)[NW], int unit_exp)
{
    long long canon[CANON];
    auto bit_field = [&](long long o, bool top) -> long long {
        // 64-bit window of W starting at bit o (o may be negative or beyond the integer): sign-extended two's complement
        const unsigned long long ext = (unsigned long long)((long long)in[NW - 1] >> 63);
        auto word = [&](long long i) -> unsigned long long { return i < 0 ? 0ull : (i < NW ? in[i] : ext); };
        unsigned long long win;
        if (o <= -64) {
            win = 0;
        } else if (o < 0) {
            win = word(0) << (-o);  // low bits of the window lie below the integer: zeros
        } else {
            const long long q = o >> 6;
            const int r = (int)(o & 63);
            win = r ? ((word(q) >> r) | (word(q + 1) << (64 - r))) : word(q);
        }
        return top ? (long long)win : (long long)(win & ((1ull << CANON_DIGITS) - 1));
    };
    for (int j = 0; j < CANON; ++j)
        canon[j] = bit_field((long long)CANON_DIGITS * (j - CANON_FWORDS) - unit_exp, j == CANON - 1);
    return round_reference(canon);
}

// ---------------------------------------------------------------------------------------------
// the contraction
// ---------------------------------------------------------------------------------------------
// XCD-aware tile order (speed only).  Workgroups b and b + 8 share an XCD and its L2: an XCD works through
// "supertiles" of 8 x 4 tiles (its 32 CUs at a time), which need 8 A tile streams + 4 B tile streams from beyond L2
// instead of 1 + 32; all XCDs sweep the column supertiles in the same order, so a B stream fetched from HBM by one
// XCD is an Infinity-Cache hit for the other seven.
__device__ __forceinline__ void tile_of_block(int bid, int nbid, int gy, int gx, int *ty, int *tx)
{
    const int q = nbid / 8, rem = nbid % 8, xcd = bid % 8;
    const int lin = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + bid / 8;  // bijective
    constexpr int SR = 8, SC = 4;
    const int per_band = SR * gx;           // tiles in a band of SR tile rows
    const int band = lin / per_band, inb = lin % per_band;
    const int rows_here = min(SR, gy - band * SR);
    const int sc = inb / (rows_here * SC), ins = inb % (rows_here * SC);
    const int cols_here = min(SC, gx - sc * SC);
    // inside a (rows_here x cols_here) supertile: column fastest (only the last supertile of a band can be ragged, and
    // ins then already runs over rows_here * cols_here tiles)
    const int ry = ins / cols_here, cx = ins % cols_here;
    *ty = band * SR + ry;
    *tx = sc * SC + cx;
}

}  // namespace exb
