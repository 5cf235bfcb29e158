// blas3_mfma.hip -- ExGEMM fast path: error-free slicing + v_mfma_f64_16x16x4_f64, exact by construction.
//
// The north star asks for MFMA_F64 in the panel x panel contraction "that feeds the accumulator".  An MFMA
// rounds every fused multiply-add, so it can only be used where no rounding can happen.  We make that so
// with an Ozaki-style error-free split (Ozaki, Ogita, Oishi, Rump, "Error-free transformations of matrix
// multiplication", Numer. Algorithms 59, 2012), done with integer shifts instead of floating-point tricks:
//
//   * row i of A' = fl(alpha*A) has a scale ea_i (|A'_il| < 2^ea_i), column j of B a scale eb_j;
//   * every entry is cut into S slices of BETA = 21 bits: A'_il = sign * sum_p a_p(i,l) * 2^(ea_i - BETA*p),
//     0 <= a_p < 2^21 (p = 1..SA), likewise b_q(l,j) -- a scan kernel measures how many slices the data needs
//     (mantissa bits actually used + exponent spread inside the row/column) and the host picks SA, SB <= 4,
//     or falls back to the scalar kernel (k_gemm) when the spread is wider;
//   * for a k-block of KP = 512 the integer products a_p*b_q (< 2^42) are contracted on the matrix cores in
//     fp64; all pairs with the same d = p+q share one accumulator:  KP * min(SA,SB) * 2^42 <= 2^53, so every
//     MFMA partial sum is an integer below 2^53 and therefore exact, in any summation order;
//   * after each k-block the (SA+SB-1) accumulators per output are converted to int64 and added, shifted by
//     BETA*(SA+SB-d), into a 256-bit two's-complement fixed-point accumulator held in registers -- the Kulisch
//     accumulator of this output, shrunk to the window the scales allow (no LDS, no global spill);
//   * the epilogue rounds that integer once (round-to-nearest-even) and scales it by 2^(ea_i+eb_j-BETA*(SA+SB)).
//
// Result: bit-identical to k_gemm / the oracle / MPFR (tests/test_gpu_blas23.py), with the O(mnk) work on
// the MFMA pipe.  On gfx950 the f64 MFMA rate equals the f64 VALU FMA rate (78.6 TFLOP/s), so the gain over
// k_gemm is the ~30 VALU ops per product of the TwoProd+TwoSum chain against SA*SB MFMA-FMAs per product.
#include "superacc.hip.h"
#include "exblas_internal.h"
#include "gemm_scan.hip.h"

#include <type_traits>

namespace exb {

typedef double v4d_t __attribute__((ext_vector_type(4)));

// compile-time loop: f(std::integral_constant<int, I>) for I in [B, E)
template <int B, int E, class F>
__device__ __forceinline__ void static_for(F &&f)
{
    if constexpr (B < E) {
        f(std::integral_constant<int, B>{});
        static_for<B + 1, E>(f);
    }
}

constexpr int MF_BETA = 21;
constexpr int MF_KP = 512;    // k-block over which MFMA partial sums stay exact
constexpr int MF_KB = 16;  // workgroup tile: BM = 32*RT rows x BN = 32*CT columns (RT x CT MFMA tiles of 16x16 per wave)
constexpr int MF_THREADS = 256;

// ---------------------------------------------------------------------------------------------
// slicing: x = sum_p out[p] * 2^(ea - BETA*(p+1)) with integer digits out[p], exactly (the scan guarantees that x is
// a multiple of 2^(ea - BETA*S) and |x| < 2^ea).  Digits are taken with the classic error-free extraction
// q = (x + c) - c, c = 1.5 * 2^(52 + w): x rounded to the nearest multiple of 2^w, exact, and x - q is exact too --
// four fp64 instructions per digit instead of a 128-bit variable shift.  Round-to-nearest makes the digits
// balanced: |out[0]| <= 2^BETA, |out[p > 0]| <= 2^(BETA-1), so a group sum over KP = 512 products stays below 2^53
// (worst group: 4 pairs, each with at least one lower digit: 4 * 512 * 2^(2*BETA-1) = 2^52).
// ---------------------------------------------------------------------------------------------
template <int S>
__device__ __forceinline__ void slice(double x, int ea, double (&out)[S])
{
#pragma unroll
    for (int p = 0; p < S; ++p) {
        const int w = ea - MF_BETA * (p + 1);  // weight exponent of digit p
        if (p < S - 1) {
            const double c = ldexp(1.5, 52 + w);
            const double q = (x + c) - c;
            x -= q;
            out[p] = ldexp(q, -w);
        } else {
            out[p] = ldexp(x, -w);
        }
    }
}

// 256-bit accumulator += sign_extend(T) << SH   (SH is a compile-time constant after unrolling)
template <int SH>
__device__ __forceinline__ void wide_add(unsigned long long (&acc)[4], long long T)
{
    constexpr int w = SH >> 6, b = SH & 63;
    const unsigned long long ext = (unsigned long long)(T >> 63);
    const unsigned long long lo = (unsigned long long)T << b;
    const unsigned long long hi = b ? (unsigned long long)(T >> (64 - b)) : ext;
    unsigned long long v[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = (i < w) ? 0ull : (i == w ? lo : (i == w + 1 ? hi : ext));
    unsigned long long c = 0, cn;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        acc[i] = __builtin_addcll(acc[i], v[i], c, &cn);
        c = cn;
    }
}

// round the 256-bit two's-complement integer to nearest-even and scale by 2^unit_exp (result stays normal:
// the host only takes this path when all scales are within +-400)
__device__ inline double wide_round(const unsigned long long (&in)[4], int unit_exp)
{
    unsigned long long m[4] = {in[0], in[1], in[2], in[3]};
    const bool neg = (long long)m[3] < 0;
    if (neg) {
        unsigned long long c = 1, cn;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            m[i] = __builtin_addcll(~m[i], 0ull, c, &cn);
            c = cn;
        }
    }
    int top = 3;
    while (top > 0 && m[top] == 0) --top;
    if (m[top] == 0) return 0.0;
    const int lz = __builtin_clzll(m[top]);
    const int msb = 64 * top + 63 - lz;
    double r;
    if (msb <= 52) {
        r = ldexp((double)m[0], unit_exp);
    } else {
        const unsigned long long below = top > 0 ? m[top - 1] : 0ull;
        unsigned long long w = lz ? ((m[top] << lz) | (below >> (64 - lz))) : m[top];
        bool sticky = (w & 0x3ffull) != 0 || (lz ? (below << lz) != 0 : below != 0);
        for (int i = top - 2; i >= 0; --i) sticky |= m[i] != 0;
        unsigned long long mant = w >> 11;
        if (((w >> 10) & 1ull) && (sticky || (mant & 1ull))) mant += 1;  // may reach 2^53: still exact in fp64
        r = ldexp((double)mant, msb - 52 + unit_exp);
    }
    return neg ? -r : r;
}

// ---------------------------------------------------------------------------------------------
// the kernel: 64x64 outputs per workgroup, each wave a 32x32 quadrant = 2x2 MFMA tiles of 16x16
// ---------------------------------------------------------------------------------------------
template <int SA, int SB, int RT, int CT, int WPS>
__global__ void __launch_bounds__(MF_THREADS, WPS) k_gemm_mfma(int ta, int tb, int m, int n, int k, double alpha,
                                                             const double *__restrict__ a, long long lda,
                                                             const double *__restrict__ b, long long ldb, double beta,
                                                             double *__restrict__ c, long long ldc,
                                                             const int *__restrict__ EA, const int *__restrict__ EB)
{
    constexpr int G = SA + SB - 1;  // accumulator groups d = p+q, d-2 in [0, G)
    constexpr int D = SA + SB;
    __shared__ double As[SA][MF_KB][32 * RT + 17];  // [slice][k][row + pad]; odd pitch: 16 lanes writing 16 k's of one row hit 16 banks
    __shared__ double Bs[SB][MF_KB][32 * CT + 16];  // [slice][k][col (BN <= 64) + pad]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int BM = 32 * RT;
    constexpr int AEPT = BM * MF_KB / MF_THREADS;  // A elements per thread per k-step (4 or 2)
    constexpr int BN = 32 * CT;
    constexpr int BEPT = MF_KB * BN / MF_THREADS;  // B elements per thread per k-step (4 or 2)
    // XCD-aware tile order (speed only): workgroups b and b+8 share an XCD (and its L2), so every XCD gets a
    // contiguous run of the row-major tile sequence and re-reads its A / B panels from its own L2.
    const int gx = (n + BN - 1) / BN;
    int wgid = blockIdx.x;
    {
        const int nwg = gridDim.x, q = nwg / 8, r = nwg % 8, xcd = wgid % 8;
        wgid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + wgid / 8;  // bijective for any nwg
    }
    const int i0 = (wgid / gx) * BM, j0 = (wgid % gx) * BN;
    const int wr = (wave >> 1) * (16 * RT), wc = (wave & 1) * (16 * CT);

    // global -> register staging, AEPT elements of A and BEPT of B per thread per k-step.  Sixteen consecutive
    // lanes take sixteen consecutive k of one row of A (128 contiguous bytes of a row-major A) resp. sixteen
    // consecutive columns of one k of B (128 contiguous bytes of a row-major B); the u-th element of a thread is 16
    // rows / columns further on.  Both the global loads and the LDS writes of a 16-lane group are then conflict-free.
    const int ak = tid & 15, ar0 = tid >> 4;   // A: k = ak, rows ar0 + 16*u
    const int bk = tid >> 4, bc0 = tid & 15;   // B: k = bk, cols bc0 + 16*u
    int ea_r[AEPT], eb_c[BEPT];
#pragma unroll
    for (int u = 0; u < AEPT; ++u) ea_r[u] = (i0 + ar0 + 16 * u < m) ? EA[i0 + ar0 + 16 * u] : 0;
#pragma unroll
    for (int u = 0; u < BEPT; ++u) eb_c[u] = (j0 + bc0 + 16 * u < n) ? EB[j0 + bc0 + 16 * u] : 0;

    auto load_a = [&](int l0, double (&ra)[AEPT]) {
        const int gl = l0 + ak;
#pragma unroll
        for (int u = 0; u < AEPT; ++u) {
            const int gi = i0 + ar0 + 16 * u;
            double v = 0.0;
            if (gi < m && gl < k) v = alpha * (ta ? a[(long long)gl * lda + gi] : a[(long long)gi * lda + gl]);
            ra[u] = v;
        }
    };
    auto load_b = [&](int l0, double (&rb)[BEPT]) {
        const int gl = l0 + bk;
#pragma unroll
        for (int u = 0; u < BEPT; ++u) {
            const int gj = j0 + bc0 + 16 * u;
            double v = 0.0;
            if (gl < k && gj < n) v = tb ? b[(long long)gj * ldb + gl] : b[(long long)gl * ldb + gj];
            rb[u] = v;
        }
    };

    unsigned long long wide[RT][CT][4][4];  // [row tile][col tile][reg][limb]
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int w = 0; w < 4; ++w) wide[rt][ct][r][w] = 0ull;

    double ra[AEPT], rb[BEPT];
    load_a(0, ra);
    load_b(0, rb);
    for (int kb = 0; kb < k; kb += MF_KP) {
        v4d_t acc[G][RT][CT];
#pragma unroll
        for (int g = 0; g < G; ++g)
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) acc[g][rt][ct] = (v4d_t){0.0, 0.0, 0.0, 0.0};
        const int kend = min(k, kb + MF_KP);
        for (int l0 = kb; l0 < kend; l0 += MF_KB) {
            __syncthreads();  // previous step's fragments consumed
#pragma unroll
            for (int u = 0; u < AEPT; ++u) {
                double sa[SA];
                slice<SA>(ra[u], ea_r[u], sa);
#pragma unroll
                for (int p = 0; p < SA; ++p) As[p][ak][ar0 + 16 * u] = sa[p];
            }
#pragma unroll
            for (int u = 0; u < BEPT; ++u) {
                double sb[SB];
                slice<SB>(rb[u], eb_c[u], sb);
#pragma unroll
                for (int q = 0; q < SB; ++q) Bs[q][bk][bc0 + 16 * u] = sb[q];
            }
            // prefetch the next k-step while this one is contracted
            const int ln = l0 + MF_KB;
            if (ln < k) {
                load_a(ln, ra);
                load_b(ln, rb);
            }
            __syncthreads();
#pragma unroll
            for (int kk = 0; kk < MF_KB; kk += 4) {
                double fa[RT][SA], fb[CT][SB];
                const int kx = kk + (lane >> 4);
#pragma unroll
                for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                    for (int p = 0; p < SA; ++p) fa[rt][p] = As[p][kx][wr + rt * 16 + (lane & 15)];
#pragma unroll
                for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                    for (int q = 0; q < SB; ++q) fb[ct][q] = Bs[q][kx][wc + ct * 16 + (lane & 15)];
#pragma unroll
                for (int p = 0; p < SA; ++p)
#pragma unroll
                    for (int q = 0; q < SB; ++q)
#pragma unroll
                        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                            for (int ct = 0; ct < CT; ++ct)
                                acc[p + q][rt][ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[rt][p], fb[ct][q],
                                                                                          acc[p + q][rt][ct], 0, 0, 0);
            }
        }
        // fold the exact group sums of this k-block into the wide integers: group g holds pairs with
        // p+q = g+2 (1-based), i.e. weight 2^(-BETA*(g+2)); unit of the wide integer = 2^(-BETA*D)
        static_for<0, G>([&](auto gc) {
            constexpr int g = decltype(gc)::value;
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        wide_add<MF_BETA * (D - (g + 2))>(wide[rt][ct][r], __double2ll_rn(acc[g][rt][ct][r]));
        });
    }
    // epilogue: f64 MFMA C/D layout: col = lane & 15, row = (lane >> 4) + 4 * reg
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int gi = i0 + wr + rt * 16 + (lane >> 4) + 4 * r;
                const int gj = j0 + wc + ct * 16 + (lane & 15);
                if (gi < m && gj < n) {
                    const double s = wide_round(wide[rt][ct][r], EA[gi] + EB[gj] - MF_BETA * D);
                    double *cij = c + (long long)gi * ldc + gj;
                    *cij = (beta == 0.0) ? s : beta * (*cij) + s;
                }
            }
}

template <int SA, int SB, int RT, int CT, int WPS>
static void launch_mfma(int ta, int tb, int m, int n, int k, double alpha, const double *a, int lda, const double *b,
                        int ldb, double beta, double *c, int ldc, const int *EA, const int *EB, hipStream_t st)
{
    constexpr int BM = 32 * RT;
    constexpr int BN = 32 * CT;
    dim3 grid(((n + BN - 1) / BN) * ((m + BM - 1) / BM));
    hipLaunchKernelGGL((k_gemm_mfma<SA, SB, RT, CT, WPS>), grid, dim3(MF_THREADS), 0, st, ta, tb, m, n, k, alpha, a, (long long)lda,
                       b, (long long)ldb, beta, c, (long long)ldc, EA, EB);
}

// The fp64-slice path (exblas_set_gemm_path(3)): kept as the literal "MFMA_F64 panel contraction" and as an A/B partner
// of the int8 path (blas3_i8.hip), which is what exgemm uses by default.
// Returns true when the MFMA path ran; false -> caller uses the scalar kernel.  Synchronises the stream once
// (the slice counts are read back on the host), so this path cannot be captured into a hipGraph.
bool exgemm_try_mfma(Ctx &c, char transa, char transb, int m, int n, int k, double alpha, const double *a, int lda,
                     const double *b, int ldb, double beta, double *cmat, int ldc, hipStream_t st, hipError_t *err)
{
    *err = hipSuccess;
    c.last_gemm_slices = 0;
    if (k <= 0) return false;
    const int ta = (transa == 'T' || transa == 't'), tb = (transb == 'T' || transb == 't');
    int *buf = (int *)workspace(c, sizeof(int) * (2 * ((size_t)m + n) + INFO_WORDS), st, err);
    if (!buf) return true;
    int *info = buf, *EA = buf + INFO_WORDS, *EB = EA + m, *LA = EB + n, *LB = LA + m;
    hipLaunchKernelGGL(k_scan_init, dim3((m + n + 255) / 256), dim3(256), 0, st, m + n, EA, LA, info);  // EA|EB and LA|LB are contiguous
    const int ysplit = k >= 2048 ? 32 : (k >= 256 ? 8 : 1);
    // rows of A' (reduction over l)
    if (!ta)
        hipLaunchKernelGGL(k_scan_contig, dim3(m), dim3(256), 0, st, a, (long long)lda, m, k, alpha, EA, LA, info);
    else
        hipLaunchKernelGGL(k_scan_strided, dim3((m + 255) / 256, ysplit), dim3(256), 0, st, a, (long long)lda, m, k, alpha,
                           EA, LA, info);
    // columns of B (reduction over l)
    if (!tb)
        hipLaunchKernelGGL(k_scan_strided, dim3((n + 255) / 256, ysplit), dim3(256), 0, st, b, (long long)ldb, n, k, 1.0,
                           EB, LB, info);
    else
        hipLaunchKernelGGL(k_scan_contig, dim3(n), dim3(256), 0, st, b, (long long)ldb, n, k, 1.0, EB, LB, info);
    hipLaunchKernelGGL(k_scan_finish, dim3((m + 255) / 256), dim3(256), 0, st, m, EA, LA, info, INFO_NEED_A);
    hipLaunchKernelGGL(k_scan_finish, dim3((n + 255) / 256), dim3(256), 0, st, n, EB, LB, info, INFO_NEED_B);
    int h[INFO_WORDS];
    if ((*err = hipMemcpyAsync(h, info, sizeof(h), hipMemcpyDeviceToHost, st)) != hipSuccess) return true;
    if ((*err = hipStreamSynchronize(st)) != hipSuccess) return true;
    if (h[INFO_FLAGS]) return false;                        // Inf/NaN/subnormal input
    if (h[INFO_EMAX] < h[INFO_EMIN]) return false;          // all-zero operand: let the scalar kernel handle it
    if (h[INFO_EMIN] < -400 || h[INFO_EMAX] > 400) return false;  // keep every result in the normal range
    const int sa = (h[INFO_NEED_A] + MF_BETA - 1) / MF_BETA, sb = (h[INFO_NEED_B] + MF_BETA - 1) / MF_BETA;
    const int s = sa > sb ? sa : sb;
    if (s > 4 || s < 1) return false;
    // k-blocks: (k / KP) * G * 2^53 must fit the 256-bit accumulator: 53 + BETA*(2S-2) + log2(k/KP * G) < 255
#define MF_GO(SA_, SB_, RT, CT, WPS) \
    launch_mfma<SA_, SB_, RT, CT, WPS>(ta, tb, m, n, k, alpha, a, lda, b, ldb, beta, cmat, ldc, EA, EB, st)
    // wave tile RT x CT and waves/SIMD: chosen by A/B on MI355X (tools/bench_gemm.py); c.variant selects the others.
    // Operands that need different digit counts get their own instantiation (SA x SB products per element pair
    // instead of max^2) for the common mixed cases; the rest run with the larger count on both sides.
    const int ua = sa < 2 ? 2 : sa, ub = sb < 2 ? 2 : sb;
    if (c.variant == 0 && ua == 2 && ub == 3) {
        MF_GO(2, 3, 1, 2, 2);
    } else if (c.variant == 0 && ua == 3 && ub == 2) {
        MF_GO(3, 2, 1, 2, 2);
    } else {
        switch (s) {
        case 1: case 2:
            if (c.variant == 1) MF_GO(2, 2, 2, 2, 1); else MF_GO(2, 2, 1, 2, 2);
            break;
        case 3:
            if (c.variant == 1) MF_GO(3, 3, 2, 2, 1);
            else if (c.variant == 2) MF_GO(3, 3, 1, 1, 4);
            else if (c.variant == 3) MF_GO(3, 3, 1, 1, 3);
            else MF_GO(3, 3, 1, 2, 2);
            break;
        default:
            if (c.variant == 2) MF_GO(4, 4, 1, 1, 2); else MF_GO(4, 4, 1, 2, 1);
            break;
        }
    }
#undef MF_GO
    c.last_gemm_slices = s < 2 ? 2 : s;
    *err = hipGetLastError();
    return true;
}

}  // namespace exb
