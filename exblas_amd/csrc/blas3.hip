// blas3.hip -- ExGEMM for gfx950, row-major (the reference's layout, tests/test.exgemm.gpu.cpp:183-184).
//
// Replaces the reference's OpenCL kernel gemm (src/gpu/blas/blas3/ExGEMM.Superacc.cl:200-283, ExGEMM.FPE.cl:209-341):
//   C_ij := beta*C_ij + Round( sum_l fl(alpha*A_il) * B_lj ),   every dot product correctly rounded.
// (The reference kernel ignores alpha/beta/trans/ld* and computes C += Round(A*B) for square matrices,
// ExGEMM.Superacc.cl:197-198,:280; with alpha = beta = 1, 'N','N' this is the same value.  The final
// beta*C + s is a plain fp64 multiply-add pair, like the reference's "+=".)
//
// Kernel k_gemm (all variants): 16x16 outputs per workgroup, one output per thread like the reference, but the
// thread's private superaccumulator is a column of an LDS array [limb][thread] (bank = f(lane) only, no
// conflicts, no scratch spills -- the reference's 39 private longs per thread live in scratch memory); A and B
// stream through LDS in 16x32 / 32x16 tiles; products are made exact with TwoProductFMA and pushed, four at a
// time, through the thread's register expansion (or straight into the column for fpe < 3).
#include "superacc.hip.h"
#include "fpe.hip.h"
#include "exblas_internal.h"

namespace exb {

constexpr int GM_T = 16;   // outputs per workgroup edge
constexpr int GM_KB = 32;  // k-depth of one LDS tile
constexpr int GM_THREADS = GM_T * GM_T;

template <int N, bool EE>
__global__ void __launch_bounds__(GM_THREADS) k_gemm(int ta, int tb, int m, int n, int k, double alpha,
                                                     const double *__restrict__ a, long long lda,
                                                     const double *__restrict__ b, long long ldb, double beta,
                                                     double *__restrict__ c, long long ldc, int round_mode,
                                                     const int *__restrict__ gate)
{
    __shared__ long long acc[NL * GM_THREADS];          // 139,264 B: private column per thread
    // Predicated launch: the int8 path (blas3_i8.hip) decides ON THE DEVICE whether the data qualifies for it; this
    // kernel is enqueued behind it either way and does the work only when *gate says "scalar" (0).
    if (gate && *gate != 0) return;
    __shared__ double As[GM_T][GM_KB + 1], Bs[GM_KB][GM_T + 1];
    const int tid = threadIdx.x, tx = tid & (GM_T - 1), ty = tid >> 4;
    // The grid is capped (gemm_variant): a workgroup walks over tiles, so that a predicated launch that has nothing to
    // do costs a few thousand empty workgroups instead of one per 16 x 16 outputs (0.11 ms at 8192 x 8192).
    const int tiles_x = (n + GM_T - 1) / GM_T, tiles_y = (m + GM_T - 1) / GM_T;
    for (long long tile = blockIdx.x; tile < (long long)tiles_x * tiles_y; tile += gridDim.x) {
    const int i0 = (int)(tile / tiles_x) * GM_T, j0 = (int)(tile % tiles_x) * GM_T;
    __syncthreads();  // the previous tile's last reads of As / Bs / acc
    for (int t = tid; t < NL * GM_THREADS; t += GM_THREADS) acc[t] = 0;
    unsigned flags = 0;
    LdsSink<GM_THREADS> sink{acc + tid, flags};
    double f[N > 0 ? N : 1];
#pragma unroll
    for (int q = 0; q < (N > 0 ? N : 1); ++q) f[q] = 0.0;

    Bypass bypass;
    for (int l0 = 0; l0 < k; l0 += GM_KB) {
        __syncthreads();
        // A tile: rows i0..i0+15, depth l0..l0+31 ; B tile: depth x cols j0..j0+15
        for (int t = tid; t < GM_T * GM_KB; t += GM_THREADS) {
            const int r = t / GM_KB, l = t % GM_KB;
            const int gi = i0 + r, gl = l0 + l;
            double v = 0.0;
            if (gi < m && gl < k) v = alpha * (ta ? a[(long long)gl * lda + gi] : a[(long long)gi * lda + gl]);
            As[r][l] = v;
        }
        for (int t = tid; t < GM_KB * GM_T; t += GM_THREADS) {
            const int l = t / GM_T, cc = t % GM_T;
            const int gl = l0 + l, gj = j0 + cc;
            double v = 0.0;
            if (gl < k && gj < n) v = tb ? b[(long long)gj * ldb + gl] : b[(long long)gl * ldb + gj];
            Bs[l][cc] = v;
        }
        __syncthreads();
#pragma unroll 2
        for (int l = 0; l < GM_KB; l += 4) {
            double p[4], e[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) p[u] = two_prod(As[ty][l + u], Bs[l + u][tx], e[u]);
            fpe_absorb_prod_adaptive<N, EE, 4>(f, p, e, sink, bypass);
        }
    }
    fpe_flush_sink<N>(f, sink);
    // finish: every thread normalises and rounds its own column (private, no barrier needed)
    const int gi = i0 + ty, gj = j0 + tx;
    long long loc[NL];
    for (int q = 0; q < NL; ++q) loc[q] = acc[q * GM_THREADS + tid];
    normalize_digits(loc);
    double s;
    if (flags) {
        const bool nan = (flags & FLAG_NAN) || ((flags & FLAG_PINF) && (flags & FLAG_NINF));
        s = nan ? __longlong_as_double(0x7ff8000000000000ll)
                : __longlong_as_double((flags & FLAG_NINF) ? (long long)0xfff0000000000000ull : 0x7ff0000000000000ll);
    } else if (round_mode) {
        long long canon[CANON];
        digits_to_canon(loc, canon);
        s = round_reference(canon);
    } else {
        s = __longlong_as_double((long long)round_exact_bits(loc));
    }
    if (gi < m && gj < n) {
        double *cij = c + (long long)gi * ldc + gj;
        *cij = (beta == 0.0) ? s : beta * (*cij) + s;
    }
    }  // tile loop
}

template <int N, bool EE>
static hipError_t gemm_variant(char transa, char transb, int m, int n, int k, double alpha, const double *a, int lda,
                               const double *b, int ldb, double beta, double *c, int ldc, int round_mode,
                               hipStream_t st, const int *gate)
{
    const int ta = (transa == 'T' || transa == 't'), tb = (transb == 'T' || transb == 't');
    const long long tiles = (long long)((n + GM_T - 1) / GM_T) * ((m + GM_T - 1) / GM_T);
    dim3 grid((unsigned)(tiles < 8192 ? tiles : 8192));  // 1 workgroup per CU (139 KiB of LDS): 32 rounds of 256
    hipLaunchKernelGGL((k_gemm<N, EE>), grid, dim3(GM_THREADS), 0, st, ta, tb, m, n, k, alpha, a, (long long)lda, b,
                       (long long)ldb, beta, c, (long long)ldc, round_mode, gate);
    return hipGetLastError();
}

// one (fpe, early_exit) variant of the scalar kernel on a row block
static hipError_t gemm_scalar(char transa, char transb, int m, int n, int k, double alpha, const double *a, int lda,
                              const double *b, int ldb, double beta, double *cmat, int ldc, int fpe, int early_exit,
                              int round_mode, hipStream_t st, const int *gate)
{
#define GM_ARGS transa, transb, m, n, k, alpha, a, lda, b, ldb, beta, cmat, ldc, round_mode, st, gate
    if (fpe < 3) return gemm_variant<0, false>(GM_ARGS);
    if (early_exit) {
        if (fpe <= 4) return gemm_variant<4, true>(GM_ARGS);
        if (fpe <= 6) return gemm_variant<6, true>(GM_ARGS);
        return gemm_variant<8, true>(GM_ARGS);
    }
    switch (fpe) {
    case 3: return gemm_variant<3, false>(GM_ARGS);
    case 4: return gemm_variant<4, false>(GM_ARGS);
    case 5: return gemm_variant<5, false>(GM_ARGS);
    case 6: return gemm_variant<6, false>(GM_ARGS);
    case 7: return gemm_variant<7, false>(GM_ARGS);
    default: return gemm_variant<8, false>(GM_ARGS);  // fpe >= 8: ExGEMM.FPE.cl with NBFPE = fpe (ExGEMM.cpp:96-97), same bits
    }
#undef GM_ARGS
}

// variant selection: ExGEMM.cpp:78-99 (fpe < 3 superaccumulators only; early-exit buckets 4/6/8).
// chunks != nullptr: the rows of C are produced chunk by chunk and chunks->hook runs after each (row-sharded GEMM).
hipError_t exgemm_dispatch(Ctx &c, char transa, char transb, int m, int n, int k, double alpha, const double *a,
                           int lda, const double *b, int ldb, double beta, double *cmat, int ldc, int fpe,
                           int early_exit, int round_mode, hipStream_t st, const GemmChunks *chunks)
{
    if (m <= 0 || n <= 0) return hipSuccess;
    c.last_gemm_slices = 0;
    c.gemm_info_dev = nullptr;
    if (early_exit && fpe > 8) return hipSuccess;  // the reference's silent no-op (ExGEMM.cpp:88-99), on every path
    const bool ta = (transa == 'T' || transa == 't');
    GemmChunks whole;
    whole.n = 1;
    whole.bound[0] = 0;
    whole.bound[1] = m;
    const GemmChunks &ch = chunks ? *chunks : whole;
    // int8 slices on the matrix cores (blas3_i8.hip) for every variant and both rounding modes: the result is the
    // correctly rounded exact dot product whichever expansion size the caller names.  The scalar kernel is enqueued
    // behind it, predicated on the device-side decision.
    I8Plan plan;
    if (c.gemm_path != 1 && c.gemm_path != 3) {
        hipError_t e = exgemm_i8_prepare(c, transa, transb, m, n, k, alpha, a, lda, b, ldb, beta, cmat, ldc, round_mode,
                                         st, &plan);
        if (e != hipSuccess) return e;
    }
    const int *gate = plan.ok ? plan.info + I8_INFO_PATH : nullptr;
    for (int i = 0; i < ch.n; ++i) {
        const int row0 = ch.bound[i], row1 = ch.bound[i + 1];
        if (row1 > row0) {
            const double *ar = ta ? a + row0 : a + (size_t)row0 * lda;
            double *cr = cmat + (size_t)row0 * ldc;
            bool done = false;
            if (plan.ok) {
                hipError_t e = exgemm_i8_rows(plan, row0, row1, st);
                if (e != hipSuccess) return e;
            } else if (c.gemm_path == 3 && round_mode == 0) {
                // fp64 slices on MFMA-F64 (blas3_mfma.hip): exact-rounding mode, host-decided
                hipError_t e = hipSuccess;
                done = exgemm_try_mfma(c, transa, transb, row1 - row0, n, k, alpha, ar, lda, b, ldb, beta, cr, ldc, st, &e);
                if (done && e != hipSuccess) return e;
            }
            if (!done) {
                hipError_t e = gemm_scalar(transa, transb, row1 - row0, n, k, alpha, ar, lda, b, ldb, beta, cr, ldc, fpe,
                                           early_exit, round_mode, st, gate);
                if (e != hipSuccess) return e;
            }
        }
        if (ch.hook) {
            const int rc = ch.hook(ch.user, i, st);
            if (rc) return (hipError_t)rc;
        }
    }
    return hipSuccess;
}

}  // namespace exb
