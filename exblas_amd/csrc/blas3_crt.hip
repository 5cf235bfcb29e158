// blas3_crt.hip -- ExGEMM on the int8 matrix cores through residue arithmetic (Chinese remainder theorem): the same
// exact result as blas3_i8.hip with O(bits) instead of O(bits^2) matrix-core work.
//
// blas3_i8.hip writes every entry as sa (sb) base-256 digits and contracts all sa*sb digit pairs: 64 int8 products
// per element pair for 53-bit mantissas with 10 binades of spread.  Here the fixed-point integers X_il = A'_il / 2^ua_i
// and Y_lj = B_lj / 2^ub_j (|X| < 2^na, |Y| < 2^nb, exact: gemm_scan.hip.h) are reduced modulo L pairwise coprime
// moduli p_0 = 256, 255, 253, 251, 247, ... (<= 256, so every symmetric residue is an int8):
//
//   1. residues (k_crt_residues): x mod p_t for t < L, one int8 plane per modulus, same tile-major layout as the digit
//      planes.  L is the smallest count with  p_0 ... p_{L-1} >= 4 * k * 2^na * 2^nb  > 4 |sum_l X_il Y_lj|, decided
//      on the device from the scan (k_crt_decide): 18 moduli where the digit path needs 8 x 8 = 64 products.
//   2. contract (k_gemm_crt): per modulus ONE plain int8 GEMM on v_mfma_i32_32x32x32_i8, exact in int32
//      (|residue| <= 128, k <= 8192 per launch: |sum| <= 2^27); longer k runs one launch per 8192, each adding its
//      residues to those of the earlier ones.  A workgroup owns a 256 x 256 block of one
//      modulus (each of its 4 waves 128 x 128 = 16 MFMA tiles, 256 accumulator registers), staged through registers.
//   3. reconstruct + round (k_crt_finish): per entry the L residues -> the integer in (-M/4, M/4) (moduli grouped in
//      threes: Garner inside a group in fp64 modular arithmetic, the classical CRT sum across the 24-bit
//      super-moduli in 32-bit words), then ONE rounding of value * 2^(ua_i + ub_j) with the routines of the digit path
//      (round-to-nearest-even, or the reference's Round() on the re-cut 41 limbs).
//
// Stream-ordered like the digit path: the decision lives in device memory, kernels that are not needed exit at their
// first instruction.  Inputs the scan rejects (Inf/NaN/subnormal, exponents beyond +-300, more than 126 bits per
// operand) leave PATH_SCALAR in the info block and the scalar kernel (blas3.hip) runs.
#include "gemm_fixed.hip.h"

#include <cstdint>
#include <cstring>
#include <mutex>

namespace exb {

constexpr int CRT_LMAX = 39;      // moduli available: 285 bits (126 + 126 bits of operands and k up to 2^31)
constexpr int CRT_G = 13;         // groups of three moduli
constexpr int CRT_W32 = 10;       // 32-bit words of the reconstructed integer (320 bits)
constexpr int CRT_NL24 = 12;      // 24-bit limbs (carried in doubles) that hold M_L for L <= 39 (288 >= 285 bits)
constexpr int CRT_BT = 256;       // block of C per workgroup: 256 x 256
constexpr int CRT_KPASS = 128;    // k chunks (of 64) per launch: 8192 * 2^14 = 2^27 < 2^31

struct CrtTables {
    int p[CRT_LMAX];
    float invp[CRT_LMAX];            // 1/p rounded to float
    unsigned c8[CRT_LMAX][4];        // 256^t mod p for t = 0..15, byte t%4 of word t/4
    int bits[CRT_LMAX + 1];          // floor(log2(M_L)), M_L = p_0 ... p_{L-1}
    double P[CRT_G][3], invP[CRT_G][3];             // super-modulus of group b when it holds w + 1 moduli
    double invPlow[CRT_G][3];                       // a shade below 1/P: floor(x invPlow) is floor(x / P) or one less
    // with L moduli in use (the last group then holds L - 3 (G - 1) of its three):
    double wc[CRT_LMAX + 1][CRT_G];                 // (M_L / P_b)^-1 mod P_b
    // gc[L][b][j] = wc[L][b] * e_j mod P_b, e_j the CRT basis of modulus j inside group b (1 mod p_j, 0 mod the others;
    // 0 for a modulus the last group does not use):  y_b = (sum_j r_j gc[L][b][j]) mod P_b  in one reduction
    double gc[CRT_LMAX + 1][CRT_G][3];
    double ml[CRT_LMAX + 1][CRT_G][CRT_NL24];       // M_L / P_b in 24-bit limbs, least significant first
    double Ml[CRT_LMAX + 1][CRT_NL24];              // M_L in 24-bit limbs
};

__device__ CrtTables g_crt;

// ---------------------------------------------------------------------------------------------
// host: the tables (small-number arithmetic only; checked end to end by the parity tests)
// ---------------------------------------------------------------------------------------------
namespace {

long long inv_mod(long long a, long long m)  // a^-1 mod m, gcd(a, m) = 1
{
    long long g = m, x = 0, y = 1, aa = ((a % m) + m) % m;
    while (aa) {
        const long long q = g / aa;
        long long t = g - q * aa; g = aa; aa = t;
        t = x - q * y; x = y; y = t;
    }
    return ((x % m) + m) % m;
}

long long gcd_ll(long long a, long long b) { return b ? gcd_ll(b, a % b) : a; }

struct Big {  // little-endian 32-bit words
    uint32_t w[CRT_W32 + 2] = {0};
    void mul_small(uint32_t f)
    {
        uint64_t carry = 0;
        for (auto &x : w) {
            const uint64_t t = (uint64_t)x * f + carry;
            x = (uint32_t)t;
            carry = t >> 32;
        }
    }
    uint32_t div_small(uint32_t d)  // in place; returns the remainder
    {
        uint64_t rem = 0;
        for (int i = CRT_W32 + 1; i >= 0; --i) {
            const uint64_t cur = (rem << 32) | w[i];
            w[i] = (uint32_t)(cur / d);
            rem = cur % d;
        }
        return (uint32_t)rem;
    }
    uint32_t mod_small(uint32_t d) const
    {
        uint64_t rem = 0;
        for (int i = CRT_W32 + 1; i >= 0; --i) rem = ((rem << 32) | w[i]) % d;
        return (uint32_t)rem;
    }
    double limb24(int i) const  // bits [24 i, 24 i + 24)
    {
        const int bit = 24 * i, wi = bit >> 5, sh = bit & 31;
        uint64_t v = wi < CRT_W32 + 2 ? w[wi] : 0u;
        if (wi + 1 < CRT_W32 + 2) v |= (uint64_t)w[wi + 1] << 32;
        return (double)((v >> sh) & 0xffffffu);
    }
    int bitlen() const
    {
        for (int i = CRT_W32 + 1; i >= 0; --i)
            if (w[i]) return 32 * i + 32 - __builtin_clz(w[i]);
        return 0;
    }
};

const CrtTables &crt_tables_host()
{
    static CrtTables t;
    static std::once_flag once;
    std::call_once(once, [] {
        memset(&t, 0, sizeof(t));
        // pairwise coprime moduli, largest first: 256, 255, 253, 251, 247, 241, ...
        int cnt = 0;
        for (int c = 256; c > 1 && cnt < CRT_LMAX; --c) {
            bool ok = true;
            for (int i = 0; i < cnt; ++i) ok = ok && gcd_ll(c, t.p[i]) == 1;
            if (ok) t.p[cnt++] = c;
        }
        for (int i = 0; i < CRT_LMAX; ++i) {
            const int p = t.p[i];
            t.invp[i] = (float)(1.0 / p);
            long long pw = 1;
            for (int d = 0; d < 16; ++d) {
                t.c8[i][d >> 2] |= (unsigned)(pw % p) << (8 * (d & 3));
                pw = (pw * 256) % p;
            }
        }
        for (int b = 0; b < CRT_G; ++b) {
            const long long p0 = t.p[3 * b], p1 = t.p[3 * b + 1], p2 = t.p[3 * b + 2];
            const long long Pw[3] = {p0, p0 * p1, p0 * p1 * p2};
            for (int w = 0; w < 3; ++w) {
                t.P[b][w] = (double)Pw[w];
                t.invP[b][w] = 1.0 / (double)Pw[w];
                t.invPlow[b][w] = (1.0 / (double)Pw[w]) * (1.0 - 1.0 / (double)(1ll << 30));
            }
        }
        Big mm;
        mm.w[0] = 1;
        for (int l = 0; l <= CRT_LMAX; ++l) {
            if (l > 0) mm.mul_small((uint32_t)t.p[l - 1]);
            t.bits[l] = mm.bitlen() - 1;
            for (int i = 0; i < CRT_NL24; ++i) t.Ml[l][i] = mm.limb24(i);
            const int G = (l + 2) / 3;
            for (int b = 0; b < G; ++b) {
                const int w = b == G - 1 ? l - 3 * (G - 1) : 3;
                const uint32_t Pb = (uint32_t)t.P[b][w - 1];
                Big q = mm;
                q.div_small(Pb);  // exact
                for (int i = 0; i < CRT_NL24; ++i) t.ml[l][b][i] = q.limb24(i);
                const long long wcl = inv_mod((long long)q.mod_small(Pb), (long long)Pb);
                t.wc[l][b] = (double)wcl;
                for (int j = 0; j < 3; ++j) {
                    t.gc[l][b][j] = 0.0;
                    if (j >= w) continue;
                    const long long pj = t.p[3 * b + j], rest = (long long)Pb / pj;
                    const long long ej = (rest * inv_mod(rest % pj, pj)) % (long long)Pb;   // < 2^24
                    t.gc[l][b][j] = (double)((wcl * ej) % (long long)Pb);                    // < 2^48 before the reduction
                }
            }
        }
    });
    return t;
}

}  // namespace

// ---------------------------------------------------------------------------------------------
// reconstruction core, shared by the kernel and by the host self-test (same code, same tables)
// ---------------------------------------------------------------------------------------------
// Moduli in groups of three with the 24-bit super-modulus P_b; the classical CRT formula on two levels:
//   value = sum_b y_b (M / P_b) - kappa M,   y_b = (sum_j r_j gc_j) mod P_b  (gc_j folds the basis inside the group and the
//   group's weight (M / P_b)^-1 mod P_b; the sum is an exact integer below 2^34),   kappa = round(sum_b y_b / P_b).
// y_b is left NON-canonical in [0, 2 P_b) (one multiply, one floor, one fma; no fix-ups): an extra P_b adds exactly M to
// the sum and exactly 1 to the fraction sum, so kappa absorbs it.  kappa is exact because |value| / M < 1/4 (k_crt_decide)
// while the fraction sum is good to 1e-14.  The big products live in 24-bit limbs carried in doubles -- y_b < 2^25 times a
// limb < 2^24, at most 13 groups: every limb sum is an exact integer below 2^53 -- i.e. NL fused multiply-adds per group
// where 32-bit words took a v_mad_u64_u32 carry chain (~3 instructions per word), and no carries until the end.
template <int NL>
struct CrtAcc {
    double limb[NL];
    double phi;
};

template <int NL>
__host__ __device__ __forceinline__ void crt_acc_zero(CrtAcc<NL> &a)
{
#pragma unroll
    for (int i = 0; i < NL; ++i) a.limb[i] = 0.0;
    a.phi = 0.0;
}

// one group: residues r0, r1, r2 (a modulus the last group does not use: any value, its coefficient is zero)
template <int NL>
__host__ __device__ __forceinline__ void crt_group_step(CrtAcc<NL> &a, double r0, double r1, double r2, double c0, double c1,
                                                        double c2, double P, double invP, double invPlow, const double (&ml)[NL])
{
    const double x = fma(r2, c2, fma(r1, c1, r0 * c0));   // exact, < 2^34
    const double y = fma(-floor(x * invPlow), P, x);       // x mod P, or that plus P: exact
    a.phi = fma(y, invP, a.phi);
#pragma unroll
    for (int i = 0; i < NL; ++i) a.limb[i] = fma(y, ml[i], a.limb[i]);
}

// minus kappa M, carry propagation (limbs to [0, 2^24), the top one keeps the sign), then the limbs packed into a
// 320-bit two's-complement integer (compile-time shifts)
// (W words of output: the kernels ask for just enough to hold 24 NL bits -- 3 words for 6 and 8 limbs -- so that the
// rounding routine walks 3 words instead of 5; the host self-test checks all CRT_W32 / 2)
template <int NL, int W = CRT_W32 / 2>
__host__ __device__ __forceinline__ void crt_acc_finish(CrtAcc<NL> &a, const double (&Ml)[NL], unsigned long long (&w)[W])
{
    static_assert(64 * W >= 24 * NL, "the words must hold every limb");
    const double kappa = rint(a.phi);
    double carry = 0.0;
    long long li[NL];
#pragma unroll
    for (int i = 0; i < NL; ++i) {
        const double t = fma(-kappa, Ml[i], a.limb[i]) + carry;   // integers below 2^53 in magnitude: exact
        if (i < NL - 1) {
            carry = floor(t * 0x1p-24);
            li[i] = (long long)(t - carry * 0x1p24);            // in [0, 2^24)
        } else {
            li[i] = (long long)t;                                 // signed top limb (|value| < M / 4: a small number)
        }
    }
#pragma unroll
    for (int k = 0; k < W; ++k) w[k] = 0ull;
#pragma unroll
    for (int i = 0; i < NL - 1; ++i) {                            // non-negative 24-bit fields: plain ORs
        const int bit = 24 * i, k = bit >> 6, sh = bit & 63;
        w[k] |= (unsigned long long)li[i] << sh;
        if (sh > 40 && k + 1 < W) w[k + 1] |= (unsigned long long)li[i] >> (64 - sh);
    }
    {   // the signed top limb: sign-extended add at its (compile-time) position
        constexpr int bit = 24 * (NL - 1), k0 = bit >> 6, sh = bit & 63;
        const long long T = li[NL - 1];
        const unsigned long long ext = (unsigned long long)(T >> 63);
        const unsigned long long lo = (unsigned long long)T << sh;
        const unsigned long long hi = sh ? (unsigned long long)(T >> (64 - sh)) : ext;
        unsigned long long c = 0;
#pragma unroll
        for (int k = 0; k < W; ++k) {
            const unsigned long long v = (k < k0) ? 0ull : (k == k0 ? lo : (k == k0 + 1 ? hi : ext));
            const unsigned long long s1 = w[k] + v, c1 = s1 < v ? 1ull : 0ull;
            const unsigned long long s2 = s1 + c, c2 = s2 < c ? 1ull : 0ull;
            w[k] = s2;
            c = c1 | c2;
        }
    }
}

// Host mirror of k_crt_finish's arithmetic over the same tables and the same routines (CPU-only check of the table
// generator and of the reconstruction, tests/test_abi.py): random integers |S| < M_L / 4 for every L -> residues ->
// groups -> limbs -> minus kappa M_L -> words; the 320-bit two's-complement result must be S again.  Runs every limb
// count the kernels instantiate that can hold M_L.
template <int NL>
static bool crt_selftest_one(const CrtTables &t, int L, const Big &mag, bool neg)
{
    if (24 * NL < t.bits[L] + 1) return true;   // this limb count is not used for L
    const int G = (L + 2) / 3, wlast = L - 3 * (G - 1);
    CrtAcc<NL> a;
    crt_acc_zero(a);
    for (int b = 0; b < G; ++b) {
        const int w = b == G - 1 ? wlast : 3;
        double r[3] = {0, 0, 0};
        for (int j = 0; j < 3; ++j) {
            const uint32_t p = (uint32_t)t.p[min(3 * b + j, L - 1)];   // past the end: a valid residue, coefficient zero
            uint32_t m = mag.mod_small(p);
            if (neg && m) m = p - m;
            r[j] = (double)m;
        }
        double ml[NL];
        for (int i = 0; i < NL; ++i) ml[i] = t.ml[L][b][i];
        crt_group_step<NL>(a, r[0], r[1], r[2], t.gc[L][b][0], t.gc[L][b][1], t.gc[L][b][2], t.P[b][w - 1], t.invP[b][w - 1],
                           t.invPlow[b][w - 1], ml);
    }
    double Ml[NL];
    for (int i = 0; i < NL; ++i) Ml[i] = t.Ml[L][i];
    unsigned long long got[CRT_W32 / 2];
    crt_acc_finish<NL>(a, Ml, got);
    // expected: two's complement of +-mag
    uint64_t c = neg ? 1 : 0;
    for (int k = 0; k < CRT_W32 / 2; ++k) {
        uint64_t word = (uint64_t)mag.w[2 * k] | ((uint64_t)mag.w[2 * k + 1] << 32);
        if (neg) {
            word = ~word;
            const uint64_t s = word + c;
            c = s < c ? 1 : 0;
            word = s;
        }
        if (word != got[k]) return false;
    }
    return true;
}

static int crt_selftest_host(int cases, unsigned seed)
{
    const CrtTables &t = crt_tables_host();
    unsigned long long rng = 0x9e3779b97f4a7c15ull * (seed + 1);
    auto next = [&]() {
        rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17;
        return (uint32_t)(rng >> 16);
    };
    int bad = 0;
    for (int L = 1; L <= CRT_LMAX; ++L)
        for (int cs = 0; cs < cases; ++cs) {
            // magnitude below 2^(bits[L] - 2) <= M_L / 4, random sign; a few edge magnitudes first
            const int mbits = t.bits[L] - 2;
            Big mag;
            if (mbits > 0 && cs != 0) {
                for (int i = 0; i < CRT_W32; ++i) mag.w[i] = cs == 1 ? 0xffffffffu : next();
                for (int i = 0; i < CRT_W32 + 2; ++i) {
                    const int lo = 32 * i;
                    if (lo >= mbits) mag.w[i] = 0;
                    else if (lo + 32 > mbits) mag.w[i] &= (1u << (mbits - lo)) - 1u;
                }
            }
            const bool neg = (cs & 1) && mag.bitlen() > 0;
            const bool ok = crt_selftest_one<4>(t, L, mag, neg) && crt_selftest_one<6>(t, L, mag, neg) &&
                            crt_selftest_one<8>(t, L, mag, neg) && crt_selftest_one<10>(t, L, mag, neg) &&
                            crt_selftest_one<12>(t, L, mag, neg);
            if (!ok) ++bad;
        }
    return bad;
}

// copies the tables into the current device's copy of g_crt (context creation: never inside a stream capture)
hipError_t crt_tables_upload()
{
    const CrtTables &t = crt_tables_host();
    return hipMemcpyToSymbol(HIP_SYMBOL(g_crt), &t, sizeof(t), 0, hipMemcpyHostToDevice);
}

// ---------------------------------------------------------------------------------------------
// decide
// ---------------------------------------------------------------------------------------------
__global__ void k_crt_decide(int *info, int clog2k, int lcap)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    int path = PATH_CRT;
    if (info[INFO_FLAGS]) path = PATH_SCALAR;  // Inf / NaN / subnormal input
    const bool nonzero = info[INFO_EMAX] >= info[INFO_EMIN];
    if (nonzero && (info[INFO_EMIN] < -I8_ERANGE || info[INFO_EMAX] > I8_ERANGE)) path = PATH_SCALAR;
    const int na = max(info[INFO_NEED_A], 1), nb = max(info[INFO_NEED_B], 1);
    if (na > 126 || nb > 126) path = PATH_SCALAR;
    // |sum_l X_il Y_lj| < 2^(na + nb + clog2k) =: 2^B.  M >= 2^(B+2) keeps value / M inside (-1/4, 1/4), which is what
    // lets k_crt_finish find the multiple of M to subtract from a floating-point sum of fractions (error << 1/4)
    const int need = na + nb + clog2k + 2;
    int L = 1;
    while (L <= lcap && g_crt.bits[L] < need) ++L;
    if (L > lcap) path = PATH_SCALAR;
    info[INFO_CRT_L] = L;
    info[INFO_CRT_NA] = na;
    info[INFO_CRT_NB] = nb;
    info[INFO_PATH] = path;
}

// ---------------------------------------------------------------------------------------------
// residues
// ---------------------------------------------------------------------------------------------
// One workgroup per (64-vector tile, 64-k chunk) as in the digit slicers; a thread holds 16 consecutive k of one vector
// as magnitudes + signs and walks the moduli: |X| mod p = (sum_t byte_t(|X|) * (256^t mod p)) mod p, four bytes per
// v_dot4_u32_u8.  CONTIG: element (v, l) at src[v*ld + l] (A for 'N', B for 'T'); else at src[l*ld + v].
template <bool CONTIG, int NW>
__device__ __forceinline__ void crt_residues_body(const double *__restrict__ src, long long ld, int nvec, int len, double scale,
                                                  int u, int v, int l0, int L, signed char *__restrict__ tile,
                                                  size_t plane_stride)
{
    unsigned w[NW][16];
    unsigned sgn[16];   // bits of +-2^23 with the element's sign (see the modulus loop)
    // ALL 16 loads are issued before the first use, from addresses clamped into the matrix (an element outside it is
    // zeroed afterwards), and the conversion below is branch-free: the first version loaded and converted element by
    // element under `if (in range)` -- the compiler kept the structure, i.e. 16 dependent memory round trips per thread
    // (global_load_dwordx2 + s_waitcnt vmcnt(0) sixteen times; 64 % of the wave time was spent waiting)
    double xr[16];
    const int vc = min(v, nvec - 1);
    if (CONTIG && l0 + 16 <= len) {
        const double *q = src + (long long)vc * ld + l0;   // (consecutive addresses: merged into wider loads)
#pragma unroll
        for (int e = 0; e < 16; ++e) xr[e] = q[e];
    } else {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int lc = min(l0 + e, len - 1);
            xr[e] = CONTIG ? src[(long long)vc * ld + lc] : src[(long long)lc * ld + vc];
        }
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const double x = (v < nvec && l0 + e < len) ? scale * xr[e] : 0.0;
        // |x| / 2^u as an integer of NW 32-bit words (gemm_fixed.hip.h: to_fixed, magnitude only, no branches):
        // sh = the mantissa's shift, >= -52 (the bits shifted out below are zero by construction), < 32 NW - 52
        const unsigned long long bits = (unsigned long long)__double_as_longlong(x);
        const int be = (int)((bits >> 52) & 0x7ffu);
        const unsigned long long m = be ? ((bits & 0x000fffffffffffffull) | 0x0010000000000000ull) : 0ull;
        const int sh = be - 1023 - 52 - u;
        const unsigned long long down = m >> ((-sh) & 63), up = m << (sh & 63);
        unsigned long long lo, hi = 0;
        if constexpr (NW <= 2) {
            lo = sh >= 0 ? up : down;
        } else {
            const unsigned long long carry = (sh & 63) ? m >> ((64 - sh) & 63) : 0ull;   // bits that cross into the high word
            lo = sh >= 64 ? 0ull : (sh >= 0 ? up : down);
            hi = sh >= 64 ? up : (sh > 0 ? carry : 0ull);
        }
        w[0][e] = (unsigned)lo;
        if constexpr (NW > 1) w[1][e] = (unsigned)(lo >> 32);
        if constexpr (NW > 2) w[2][e] = (unsigned)hi;
        if constexpr (NW > 3) w[3][e] = (unsigned)(hi >> 32);
        sgn[e] = ((unsigned)(bits >> 32) & 0x80000000u) | 0x4b000000u;
    }
    // The modulus loop in packed fp32 (v_pk_add_f32 / v_pk_fma_f32: two elements per instruction).  With S = 2^23 and
    // cf = +-S (the element's sign): F = bits(s | cf) = +-(S + s) exactly (s < 2^20), sf = F - cf = +-s;
    // q = (fma(sf, 1/p, 1.5 S) - 1.5 S) = the integer nearest to sf / p (|sf / p| < 2^22: the sum lies where the spacing of
    // the floats is 1, and |sf fl(1/p) - sf / p| < 2^-10); r' = fma(q, -p, sf + 1.5 S) = 1.5 S + r with the symmetric
    // residue r = sf - q p, |r| <= (p - 1) / 2 for the odd moduli, <= 128 for p = 256 -- every step exact -- and the LOW
    // BYTE of r' is r as an int8 (+128 wraps to -128 = 128 mod 256).  Per element and modulus: NW dot4 + 1 or + 2.5
    // packed instructions + 0.75 byte permutes, where the scalar form (cvt, xor, mul, rndne, fma, cvt, pack) took 6.5.
    typedef float f2_t __attribute__((ext_vector_type(2)));
    const float MAGIC = 12582912.0f;   // 1.5 * 2^23
    const f2_t mg = {MAGIC, MAGIC};
#pragma unroll 1
    for (int t = 0; t < L; ++t) {
        const float pf = (float)g_crt.p[t], invp = g_crt.invp[t];
        const f2_t invp2 = {invp, invp}, np2 = {-pf, -pf};
        unsigned c[NW];
#pragma unroll
        for (int j = 0; j < NW; ++j) c[j] = g_crt.c8[t][j];
        v4i_t pk;
#pragma unroll
        for (int e4 = 0; e4 < 4; ++e4) {
            f2_t rr[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int e = 4 * e4 + 2 * h;
                unsigned s0 = 0, s1 = 0;
#pragma unroll
                for (int j = 0; j < NW; ++j) {
                    s0 = __builtin_amdgcn_udot4(w[j][e], c[j], s0, false);
                    s1 = __builtin_amdgcn_udot4(w[j][e + 1], c[j], s1, false);
                }
                const f2_t F = {__uint_as_float(s0 | sgn[e]), __uint_as_float(s1 | sgn[e + 1])};
                const f2_t C = {__uint_as_float(sgn[e]), __uint_as_float(sgn[e + 1])};
                const f2_t sf = F - C;
                const f2_t q = __builtin_elementwise_fma(sf, invp2, mg) - mg;
                rr[h] = __builtin_elementwise_fma(q, np2, sf + mg);
            }
            const unsigned lo = __builtin_amdgcn_perm(__float_as_uint(rr[0].y), __float_as_uint(rr[0].x), 0x0c0c0400u);
            const unsigned hi = __builtin_amdgcn_perm(__float_as_uint(rr[1].y), __float_as_uint(rr[1].x), 0x04000c0cu);
            pk[e4] = (int)(lo | hi);
        }
        *(v4i_t *)(tile + (size_t)t * plane_stride) = pk;
    }
}

template <bool CONTIG>
__global__ void __launch_bounds__(256, 4) k_crt_residues(const double *__restrict__ src, long long ld, int nvec, int len,
                                                      double scale, const int *__restrict__ E,
                                                      const int *__restrict__ info, int which,
                                                      signed char *__restrict__ planes, size_t plane_stride, int vt0)
{
    if (info[INFO_PATH] != PATH_CRT) return;
    const int L = info[INFO_CRT_L], need = info[which ? INFO_CRT_NB : INFO_CRT_NA];
    // vt: tile index inside `planes` (a row chunk of A' starts at vector tile vt0; B: vt0 = 0)
    // CONTIG: a workgroup reads 512 contiguous bytes of 64 vectors and consecutive workgroups continue along them (kc
    // fastest).  Strided (vectors = columns): it reads 512 contiguous bytes of 64 ROWS l, so the workgroups that run
    // together must be neighbours in v (vt fastest: together they sweep whole rows) -- with kc fastest they touched 512
    // bytes every 64 KiB
    const int kc = CONTIG ? blockIdx.x : blockIdx.y, vt = CONTIG ? blockIdx.y : blockIdx.x, KC = CONTIG ? gridDim.x : gridDim.y;
    const int r = CONTIG ? (threadIdx.x >> 2) : (threadIdx.x & 63), seg = CONTIG ? (threadIdx.x & 3) : (threadIdx.x >> 6);
    const int v = (vt0 + vt) * I8_T + r, l0 = kc * I8_T + seg * 16;
    const int u = (v < nvec ? E[v] : 0) - need;
    signed char *tile = planes + ((size_t)vt * KC + kc) * I8_TILE + tile_off(r, seg * 16);
    const int nw = (need + 31) >> 5;  // 32-bit words of |X| in use (wave-uniform): the loops carry no branches
#define CRT_RES(NW) crt_residues_body<CONTIG, NW>(src, ld, nvec, len, scale, u, v, l0, L, tile, plane_stride)
    if (nw <= 1) CRT_RES(1);
    else if (nw == 2) CRT_RES(2);
    else if (nw == 3) CRT_RES(3);
    else CRT_RES(4);
#undef CRT_RES
}

// ---------------------------------------------------------------------------------------------
// the contraction: one int8 GEMM per modulus
// ---------------------------------------------------------------------------------------------
// Workgroup = (modulus, 256 x 256 block of C); wave (wy, wx) owns the 128 x 128 quarter = 4 x 4 MFMA tiles.  Per 64-byte
// k chunk the workgroup stages 4 + 4 tile planes of 4 KiB (32 KiB, three stages), a wave reads 2 x (4 + 4) fragments
// and issues 2 x 16 MFMAs; the fragments of the next k-step and one staging instruction are issued between the MFMAs
// of the current k-step, one barrier per chunk.
// Result: R[modulus][row / 4][col] = the four residues (rows 4g .. 4g+3, unsigned bytes) of C mod p_t.
// PA holds the residue planes of the row chunk that starts at tile row ty0 (ty_cnt tile rows), R the residues of the same
// rows of C (m4 = groups of four rows in the chunk).
__global__ void __launch_bounds__(256, 1) k_gemm_crt(int n, int row_end, int ty0, int ty_cnt, int gx, int KC,
                                                     int kc0, int kc1, const signed char *__restrict__ PA,
                                                     const signed char *__restrict__ PB,
                                                     size_t plane_a, size_t plane_b, const int *__restrict__ info,
                                                     unsigned *__restrict__ R, int m4, int mod0)
{
    __shared__ v4i_t lds[3][8 * 256];
    if (info[INFO_PATH] != PATH_CRT) return;
    const int L = info[INFO_CRT_L];
    const int by_cnt = (ty_cnt + 3) >> 2, bx_cnt = (gx + 3) >> 2, ntiles = by_cnt * bx_cnt;
    const int mod = mod0 + blockIdx.x / ntiles;
    if (mod >= L) return;
    int by, bx;
    tile_of_block(blockIdx.x % ntiles, ntiles, by_cnt, bx_cnt, &by, &bx);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wy = wave >> 1, wx = wave & 1, half = lane >> 5;
    const int p = g_crt.p[mod];
    const float invp = g_crt.invp[mod];

    const signed char *ga[4], *gb[4];  // wave-uniform tile streams; the lane offset is added at the load
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int ta = min(by * 4 + i, ty_cnt - 1), tb = min(bx * 4 + i, gx - 1);  // ragged edge: a valid tile again
        ga[i] = PA + (size_t)mod * plane_a + (size_t)ta * KC * I8_TILE;
        gb[i] = PB + (size_t)mod * plane_b + (size_t)tb * KC * I8_TILE;
    }
    const unsigned lane_off = (unsigned)tid * 16u;
    int fo[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) fo[ks] = tile_off(lane & 31, (2 * ks + half) * 16) >> 4;
    const int abase = wy * 2 * 256, bbase = (4 + wx * 2) * 256;

    v16i_t acc[16];
#pragma unroll
    for (int g = 0; g < 16; ++g)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[g][r] = 0;

    auto clampk = [&](int kc) { return kc < kc1 ? kc : kc1 - 1; };
    // Staging through registers: global_load_dwordx4 -> (two chunks later) ds_write_b128.  Measured at 8192^3, 18 moduli:
    // this form 9.2 ms; LDS-DMA with two stages 9.7, with four 9.1; the same loop without any staging (stale LDS) 7.2 and
    // with neither fragment reads nor barriers 7.05 -- the chip is power-limited here (the clock drops as the matrix
    // pipe fills), so what staging costs is its energy, not its issue slots or its latency.  Chunk c lives in LDS stage c % 3 and, before that, in register set c % 3; in iteration i (chunk
    // i): MFMAs of chunk i, fragment reads for the next k-step, ds_write of chunk i + 2 (set -> stage (i + 2) % 3, free
    // since the barrier of iteration i - 1), global loads of chunk i + 4 into the set chunk i + 1 left.  One barrier
    // per chunk, LDS traffic only (no vmcnt wait: the loads in flight belong to later chunks).
    v4i_t st[3][8];
    auto gload = [&](auto sc, auto jc, int kc) {
        constexpr int sidx = decltype(sc)::value, j = decltype(jc)::value;
        st[sidx][j] = *(const v4i_t *)((j < 4 ? ga[j & 3] : gb[j & 3]) + (size_t)clampk(kc) * I8_TILE + lane_off);
    };
    auto lwrite = [&](auto sc, auto jc) {  // set s -> stage s, lane-linear image of the 4 KiB tile plane
        constexpr int sidx = decltype(sc)::value, j = decltype(jc)::value;
        lds[sidx][j * 256 + tid] = st[sidx][j];
    };
    // unit u (32 rows) of this wave's A (B) quarter: tile plane u / 2, rows 32 (u % 2) ..
    auto fload = [&](int buf, int ks, v4i_t (&fa)[4], v4i_t (&fb)[4]) {
#pragma unroll
        for (int q = 0; q < 4; ++q) fb[q] = lds[buf][bbase + (q >> 1) * 256 + (q & 1) * 128 + fo[ks]];
#pragma unroll
        for (int q = 0; q < 4; ++q) fa[q] = lds[buf][abase + (q >> 1) * 256 + (q & 1) * 128 + fo[ks]];
    };
    // one k-step (16 MFMAs) of chunk kc, residue class r = (kc - kc0) % 3; h = 0: first k-step (the fragments of the
    // second are read meanwhile), h = 1: second (fragments of the next chunk's first).  Explicit issue order: per
    // group one fragment read, one staging instruction (a store of chunk kc + 2 or a load of chunk kc + 4, alternating)
    // and two MFMAs; a scheduling barrier pins each group.
    auto khalf = [&](auto hc, auto rc, const v4i_t (&ca)[4], const v4i_t (&cb)[4], v4i_t (&na)[4], v4i_t (&nb)[4], int kc) {
        constexpr int h = decltype(hc)::value, r = decltype(rc)::value;
        constexpr int rstage = h == 0 ? r : (r + 1) % 3, rks = h == 0 ? 1 : 0;
        constexpr int wset = (r + 2) % 3, lset = (r + 1) % 3;
        static_for_i8<0, 8>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            if constexpr (i < 4) nb[i] = lds[rstage][bbase + (i >> 1) * 256 + (i & 1) * 128 + fo[rks]];
            else na[i - 4] = lds[rstage][abase + ((i - 4) >> 1) * 256 + ((i - 4) & 1) * 128 + fo[rks]];
            if constexpr (i & 1) lwrite(std::integral_constant<int, wset>{}, std::integral_constant<int, h * 4 + i / 2>{});
            else gload(std::integral_constant<int, lset>{}, std::integral_constant<int, h * 4 + i / 2>{}, kc + 4);
            static_for_i8<2 * i, 2 * i + 2>([&](auto mc) {
                constexpr int mm = decltype(mc)::value, pu = mm >> 2, qu = mm & 3;
                acc[mm] = __builtin_amdgcn_mfma_i32_32x32x32_i8(ca[pu], cb[qu], acc[mm], 0, 0, 0);
            });
            __builtin_amdgcn_sched_barrier(0);
        });
    };

    v4i_t fa0[4], fb0[4], fa1[4], fb1[4];
    auto body = [&](auto rc, int kc) {
        khalf(std::integral_constant<int, 0>{}, rc, fa0, fb0, fa1, fb1, kc);
        // the stores of chunk kc+1 (previous iteration) and the first half of chunk kc+2 are visible after this; nobody
        // reads stage r any more
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        khalf(std::integral_constant<int, 1>{}, rc, fa1, fb1, fa0, fb0, kc);
    };
    // prologue: chunks 0 and 1 through registers into stages 0 and 1, chunks 2 and 3 into sets 2 and 0
    static_for_i8<0, 8>([&](auto jc) { gload(std::integral_constant<int, 0>{}, jc, kc0); });
    static_for_i8<0, 8>([&](auto jc) { gload(std::integral_constant<int, 1>{}, jc, kc0 + 1); });
    static_for_i8<0, 8>([&](auto jc) { lwrite(std::integral_constant<int, 0>{}, jc); });
    static_for_i8<0, 8>([&](auto jc) { lwrite(std::integral_constant<int, 1>{}, jc); });
    static_for_i8<0, 8>([&](auto jc) { gload(std::integral_constant<int, 2>{}, jc, kc0 + 2); });
    static_for_i8<0, 8>([&](auto jc) { gload(std::integral_constant<int, 0>{}, jc, kc0 + 3); });
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    fload(0, 0, fa0, fb0);
    int kc = kc0;
    for (; kc + 3 <= kc1; kc += 3) {  // branch-free: a merge inside would make the register sets wait for their loads
        body(std::integral_constant<int, 0>{}, kc);
        body(std::integral_constant<int, 1>{}, kc + 1);
        body(std::integral_constant<int, 2>{}, kc + 2);
    }
    if (kc < kc1) {
        body(std::integral_constant<int, 0>{}, kc);
        if (kc + 1 < kc1) body(std::integral_constant<int, 1>{}, kc + 1);
    }

    // epilogue: residues in [0, p), four rows per 32-bit word.  C layout of the 32x32 MFMA tile: col = lane & 31,
    // row = 8 * (r / 4) + 4 * (lane / 32) + (r % 4)
    const int row_base = (ty0 + by * 4) * I8_T + wy * 128, col_base = bx * CRT_BT + wx * 128 + (lane & 31);
#pragma unroll
    for (int pu = 0; pu < 4; ++pu)
#pragma unroll
        for (int qu = 0; qu < 4; ++qu)
#pragma unroll
            for (int a4 = 0; a4 < 4; ++a4) {
                unsigned word = 0;
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const int a = acc[pu * 4 + qu][4 * a4 + b];
                    int rr = a - __mul24((int)floorf((float)a * invp), p);  // |q| < 2^22; off by at most one either way
                    rr = rr < 0 ? rr + p : rr;
                    rr = rr >= p ? rr - p : rr;
                    word |= (unsigned)rr << (8 * b);
                }
                const int gi = row_base + pu * 32 + 8 * a4 + 4 * half, gj = col_base + qu * 32;
                if (gi < row_end && gj < n) {
                    unsigned *dst = R + ((size_t)mod * m4 + ((gi - ty0 * I8_T) >> 2)) * n + gj;
                    if (kc0 > 0) {  // a later k block (k > 8192): add to the residues of the earlier ones
                        const unsigned old = *dst;
                        unsigned sum = 0;
#pragma unroll
                        for (int b = 0; b < 4; ++b) {
                            unsigned v = ((old >> (8 * b)) & 255u) + ((word >> (8 * b)) & 255u);
                            v = v >= (unsigned)p ? v - (unsigned)p : v;
                            sum |= v << (8 * b);
                        }
                        word = sum;
                    }
                    *dst = word;
                }
            }
}

// ---------------------------------------------------------------------------------------------
// reconstruct and round
// ---------------------------------------------------------------------------------------------
// NL = 24-bit limbs the sums are carried in (a bucket of the width of M_L): the inner loops carry no branches.
// Thread = (group of 4 rows, column): the L words R[t][g][j] hold the residues of its 4 entries, loaded two groups of
// moduli ahead of their use; the table entries of a group are wave-uniform scalars, read once for the four entries.
template <int NL>
__device__ __forceinline__ void crt_finish_body(int L, int na, int nb, int g, int gj, int row1, const unsigned *__restrict__ rp,
                                                size_t stride, const int *__restrict__ EA, const int *__restrict__ EB,
                                                double beta, double *__restrict__ c, long long ldc, int round_mode)
{
    const int G = (L + 2) / 3, wlast = L - 3 * (G - 1);   // groups in use; moduli in the last one (1..3)
    auto fetch = [&](int b, unsigned (&d)[3]) {
#pragma unroll
        for (int j = 0; j < 3; ++j) d[j] = rp[(size_t)min(3 * b + j, L - 1) * stride];  // past the end: a valid word, unused
    };
    CrtAcc<NL> acc[4];
#pragma unroll
    for (int o = 0; o < 4; ++o) crt_acc_zero(acc[o]);
    auto group = [&](int b, int w, const unsigned (&cur)[3]) {
        const double c0 = g_crt.gc[L][b][0], c1 = g_crt.gc[L][b][1], c2 = g_crt.gc[L][b][2];
        const double Pb = g_crt.P[b][w - 1], iPb = g_crt.invP[b][w - 1], iPlow = g_crt.invPlow[b][w - 1];
        double ml[NL];
#pragma unroll
        for (int i = 0; i < NL; ++i) ml[i] = g_crt.ml[L][b][i];
#pragma unroll
        for (int o = 0; o < 4; ++o)
            crt_group_step<NL>(acc[o], (double)((cur[0] >> (8 * o)) & 255u), (double)((cur[1] >> (8 * o)) & 255u),
                               (double)((cur[2] >> (8 * o)) & 255u), c0, c1, c2, Pb, iPb, iPlow, ml);
    };
    unsigned cur[3], nx1[3], nx2[3];
    fetch(0, cur);
    fetch(1, nx1);
#pragma unroll 1
    for (int b = 0; b < G - 1; ++b) {
        fetch(b + 2, nx2);
        group(b, 3, cur);
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            cur[j] = nx1[j];
            nx1[j] = nx2[j];
        }
    }
    group(G - 1, wlast, cur);
    double Ml[NL];
#pragma unroll
    for (int i = 0; i < NL; ++i) Ml[i] = g_crt.Ml[L][i];
    const int ebj = EB[gj] - nb;
#pragma unroll
    for (int o = 0; o < 4; ++o) {
        const int gi = 4 * g + o;
        if (gi < row1) {
            constexpr int W = (24 * NL + 63) / 64;   // 3 words for 6 and 8 limbs, 5 for 12
            unsigned long long w5[W];
            crt_acc_finish<NL, W>(acc[o], Ml, w5);
            const int u0 = EA[gi] - na + ebj;
            const double s = round_mode ? wide_round_reference<W>(w5, u0) : wide_round_sel<W>(w5, u0);
            double *cij = c + (long long)gi * ldc + gj;
            *cij = (beta == 0.0) ? s : beta * (*cij) + s;   // (fetching EA / C ahead of the reconstruction: 83 -> 110 VGPRs, 0.82 -> 0.87 ms)
        }
    }
}

// One kernel per limb-count bucket (6: up to 144 bits -- 18 moduli; 8: 192 bits; 12: every L) and rounding mode, each
// with the register budget of its own body (one kernel holding all bodies and both roundings took 191 VGPRs and spilled
// 274 SGPRs); the launches the data does not need exit at once (an empty launch of 16384 workgroups costs ~5 us).
template <int NL, int RM>
__global__ void __launch_bounds__(256) k_crt_finish(int row0, int row1, int n, const int *__restrict__ info,
                                                    const int *__restrict__ EA, const int *__restrict__ EB, double beta,
                                                    double *__restrict__ c, long long ldc,
                                                    const unsigned *__restrict__ R, int m4, int all)
{
    // R: the residues of the row chunk that starts at row0 (m4 groups of four rows)
    if (info[INFO_PATH] != PATH_CRT) return;
    const int L = info[INFO_CRT_L], na = info[INFO_CRT_NA], nb = info[INFO_CRT_NB];
    // 24-bit limbs that hold M_L (the top limb, a double, takes the headroom of the 13-term sums)
    const int nl = (g_crt.bits[L] + 1 + 23) / 24;
    if (!all && (nl <= 6 ? 6 : (nl <= 8 ? 8 : CRT_NL24)) != NL) return;
    const long long loc = (long long)blockIdx.x * 256 + threadIdx.x;
    const int groups = (row1 - row0 + 3) >> 2;
    if (loc >= (long long)groups * n) return;
    const int gl = (int)(loc / n), g = (row0 >> 2) + gl, gj = (int)(loc % n);
    const size_t stride = (size_t)m4 * n;
    const unsigned *rp = R + (size_t)gl * n + gj;
    crt_finish_body<NL>(L, na, nb, g, gj, row1, rp, stride, EA, EB, beta, c, ldc, RM);
}

// ---------------------------------------------------------------------------------------------
// host side: a pure sequence of launches (same two-step shape as the digit path: whole operands, then rows of C)
// ---------------------------------------------------------------------------------------------
// Workspace: info | EA EB LA LB | residue planes of B (whole, lcap moduli) | residue planes of ONE row chunk of A' |
// residues of the same rows of C.  The rows of C are produced chunk by chunk (crt_chunk_rows: 2048 rows once m exceeds
// 3072), each chunk = residues of its rows of A', the contractions, the reconstruction -- so only B's planes scale with
// the whole problem: 8192^3 reserves 39 x (64 + 16 + 16) MiB = 3.7 GiB (was 39 x 192 MiB = 7.3 GiB with whole-matrix
// planes of A' and R), 16384^3 12.2 GiB (was 29 GiB), and the chunk's planes (0.3 GiB for 18 moduli) stay in the
// Infinity Cache's reach between the kernels that write and read them.
static int crt_chunk_rows(int m) { return m > 3072 ? 2048 : ((m + 255) / 256) * 256; }

hipError_t exgemm_crt_prepare(Ctx &c, char transa, char transb, int m, int n, int k, double alpha, const double *a, int lda,
                              const double *b, int ldb, double beta, double *cmat, int ldc, int round_mode,
                              hipStream_t st, I8Plan *plan)
{
    plan->ok = false;
    if (k <= 0 || m <= 0 || n <= 0) return hipSuccess;
    const int ta = (transa == 'T' || transa == 't'), tb = (transb == 'T' || transb == 't');
    const int gx = (n + I8_T - 1) / I8_T, KC = (k + I8_T - 1) / I8_T;
    int lcap = c.gemm_max_moduli > 0 ? c.gemm_max_moduli : CRT_LMAX;
    if (lcap > CRT_LMAX) lcap = CRT_LMAX;
    int clog2k = 0;
    while ((1ll << clog2k) < (long long)k) ++clog2k;
    const int chunk = crt_chunk_rows(m), ctiles = chunk / I8_T, cm4 = chunk / 4;
    const size_t plane_a = (size_t)ctiles * KC * I8_TILE, plane_b = (size_t)gx * KC * I8_TILE;
    size_t off = 0, o_info = 0, o_e = 0, o_pa = 0, o_pb = 0, o_r = 0;
    auto take = [&](size_t bytes) {
        const size_t o = off;
        off += (bytes + 255) & ~(size_t)255;
        return o;
    };
    auto layout = [&](int l) {
        off = 0;
        o_info = take(sizeof(int) * INFO_WORDS);
        o_e = take(sizeof(int) * 2 * ((size_t)m + n));
        o_pb = take(plane_b * l);
        o_pa = take(plane_a * l);
        o_r = take((size_t)l * cm4 * n * sizeof(unsigned));
    };
    // Out of memory: retry with fewer moduli (24 cover 53-bit mantissas with 30 binades of spread inside a row at
    // k = 8192, 18 what fpuniform(10) needs); data that needs more than were reserved takes the scalar kernel, decided
    // on the device as always.  c.gemm_ws_moduli records what was reserved (exblas_last_gemm_info out[4]).
    hipError_t e = hipSuccess;
    char *base = nullptr;
    const int tries[4] = {lcap, 24, 18, 12};
    for (int t = 0; t < 4 && !base; ++t) {
        if (t > 0 && tries[t] >= lcap) continue;
        if (t > 0) lcap = tries[t];
        layout(lcap);
        base = (char *)workspace(c, off, st, &e);
        if (!base && e == hipErrorStreamCaptureUnsupported) return e;  // the caller must reserve before capturing
    }
    c.gemm_ws_moduli = base ? lcap : 0;
    if (!base) return hipSuccess;  // out of memory even for 12 moduli: the caller falls back (digit path, scalar kernel)
    int *info = (int *)(base + o_info);
    int *EA = (int *)(base + o_e), *EB = EA + m, *LA = EB + n, *LB = LA + m;
    signed char *PA = (signed char *)(base + o_pa), *PB = (signed char *)(base + o_pb);

    hipLaunchKernelGGL(k_scan_init, dim3((m + n + 255) / 256), dim3(256), 0, st, m + n, EA, LA, info);
    const int ysplit = k >= 2048 ? 32 : (k >= 256 ? 8 : 1);
    if (!ta)
        hipLaunchKernelGGL(k_scan_contig, dim3(m), dim3(256), 0, st, a, (long long)lda, m, k, alpha, EA, LA, info);
    else
        hipLaunchKernelGGL(k_scan_strided, dim3((m + 255) / 256, ysplit), dim3(256), 0, st, a, (long long)lda, m, k, alpha,
                           EA, LA, info);
    if (!tb)
        hipLaunchKernelGGL(k_scan_strided, dim3((n + 255) / 256, ysplit), dim3(256), 0, st, b, (long long)ldb, n, k, 1.0,
                           EB, LB, info);
    else
        hipLaunchKernelGGL(k_scan_contig, dim3(n), dim3(256), 0, st, b, (long long)ldb, n, k, 1.0, EB, LB, info);
    hipLaunchKernelGGL(k_scan_finish, dim3((m + 255) / 256), dim3(256), 0, st, m, EA, LA, info, INFO_NEED_A);
    hipLaunchKernelGGL(k_scan_finish, dim3((n + 255) / 256), dim3(256), 0, st, n, EB, LB, info, INFO_NEED_B);
    hipLaunchKernelGGL(k_crt_decide, dim3(1), dim3(64), 0, st, info, clog2k, lcap);

    // B's residues once for all row chunks; A's per chunk (exgemm_crt_rows)
    if (!tb)
        hipLaunchKernelGGL((k_crt_residues<false>), dim3(gx, KC), dim3(256), 0, st, b, (long long)ldb, n, k, 1.0, EB, info,
                           1, PB, plane_b, 0);
    else
        hipLaunchKernelGGL((k_crt_residues<true>), dim3(KC, gx), dim3(256), 0, st, b, (long long)ldb, n, k, 1.0, EB, info,
                           1, PB, plane_b, 0);
    plan->ok = true;
    plan->crt = true;
    plan->m = m; plan->n = n; plan->KC = KC;
    plan->info = info; plan->EA = EA; plan->EB = EB; plan->PA = PA; plan->PB = PB;
    plan->R = (unsigned *)(base + o_r);
    plan->plane_a = plane_a; plan->plane_b = plane_b; plan->lcap = lcap; plan->m4 = cm4;
    plan->chunk_rows = chunk; plan->a = a; plan->lda = lda; plan->alpha = alpha; plan->ta = ta; plan->k = k;
    // tuning variants 21..25: 1, 2, 3, 6 moduli per launch / all in one; default: by the number of tiles (exgemm_crt_rows)
    plan->mods_per_launch = c.variant == 21 ? 1 : (c.variant == 22 ? 2 : (c.variant == 23 ? 3 : (c.variant == 24 ? 6 : (c.variant == 25 ? lcap : 0))));
    plan->num_cu = c.num_cu;
    plan->beta = beta; plan->c = cmat; plan->ldc = ldc; plan->round_mode = round_mode;
    c.gemm_info_dev = info;
    return hipGetLastError();
}

// rows [row0, row1) of C, row0 a multiple of 64; internally in chunks of plan.chunk_rows rows
hipError_t exgemm_crt_rows(const I8Plan &p, int row0, int row1, hipStream_t st)
{
    const int gx = (p.n + I8_T - 1) / I8_T, bx_cnt = (gx + 3) / 4;
    for (int c0 = row0; c0 < row1; c0 += p.chunk_rows) {
        const int c1 = min(row1, c0 + p.chunk_rows);
        const int ty0 = c0 / I8_T, ty_cnt = (c1 - c0 + I8_T - 1) / I8_T, by_cnt = (ty_cnt + 3) / 4;
        // residues of this chunk's rows of A' = fl(alpha * A)
        if (!p.ta)
            hipLaunchKernelGGL((k_crt_residues<true>), dim3(p.KC, ty_cnt), dim3(256), 0, st, p.a, (long long)p.lda, p.m, p.k,
                               p.alpha, p.EA, p.info, 0, p.PA, p.plane_a, ty0);
        else
            hipLaunchKernelGGL((k_crt_residues<false>), dim3(ty_cnt, p.KC), dim3(256), 0, st, p.a, (long long)p.lda, p.m,
                               p.k, p.alpha, p.EA, p.info, 0, p.PA, p.plane_a, ty0);
        // A few moduli per launch.  The workgroups of an XCD share their A / B tile streams through its L2 only while
        // they run in step; they start in step at the beginning of a launch and drift apart afterwards: about 12 rounds
        // of one workgroup per CU per launch (8192^3, whole-matrix launches: 15 GB through the fabric per call with 12
        // rounds, 30.5 GB with 72).  Launches for moduli the data does not need exit at once.
        int per = p.mods_per_launch;
        if (per <= 0) {
            const long long tiles = (long long)by_cnt * bx_cnt;
            per = (int)max(1ll, min((long long)p.lcap, (12ll * p.num_cu) / max(1ll, tiles)));
        }
        for (int kc0 = 0; kc0 < p.KC; kc0 += CRT_KPASS)
            for (int mod0 = 0; mod0 < p.lcap; mod0 += per) {
                const int nm = min(per, p.lcap - mod0);
                hipLaunchKernelGGL(k_gemm_crt, dim3((unsigned)(nm * by_cnt * bx_cnt)), dim3(256), 0, st, p.n, c1, ty0, ty_cnt,
                                   gx, p.KC, kc0, min(p.KC, kc0 + CRT_KPASS), p.PA, p.PB, p.plane_a, p.plane_b, p.info,
                                   p.R, p.m4, mod0);
            }
        const long long groups = (c1 - c0 + 3) / 4;
#define CRT_FIN(NLB, ALL)                                                                                                \
    do {                                                                                                                 \
        if (p.round_mode)                                                                                                \
            hipLaunchKernelGGL((k_crt_finish<NLB, 1>), dim3((unsigned)((groups * p.n + 255) / 256)), dim3(256), 0, st, c0, \
                               c1, p.n, p.info, p.EA, p.EB, p.beta, p.c, (long long)p.ldc, p.R, p.m4, ALL);              \
        else                                                                                                             \
            hipLaunchKernelGGL((k_crt_finish<NLB, 0>), dim3((unsigned)((groups * p.n + 255) / 256)), dim3(256), 0, st, c0, \
                               c1, p.n, p.info, p.EA, p.EB, p.beta, p.c, (long long)p.ldc, p.R, p.m4, ALL);              \
    } while (0)
        if (groups * p.n < (1 << 18)) {
            CRT_FIN(CRT_NL24, 1);  // small products are launch-bound: one kernel (12 limbs) for every width
        } else {
            CRT_FIN(6, 0);
            CRT_FIN(8, 0);
            CRT_FIN(CRT_NL24, 0);
        }
#undef CRT_FIN
    }
    return hipGetLastError();
}

}  // namespace exb

// CPU-only: 0 = the tables reconstruct `cases` random integers per modulus count (tests/test_abi.py)
extern "C" int exblas_crt_selftest(int cases, unsigned seed) { return exb::crt_selftest_host(cases, seed); }
