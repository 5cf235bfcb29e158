// blas3_i8.hip -- ExGEMM on the int8 matrix cores of gfx950, exact by construction and fully stream-ordered.
//
// What it replaces: the reference kernel gemm (src/gpu/blas/blas3/ExGEMM.Superacc.cl:200-283, ExGEMM.FPE.cl:209-341)
// keeps one 39-limb superaccumulator per thread and pushes every TwoProductFMA product through it.  Same result here,
//     C_ij := beta*C_ij + Round( sum_l fl(alpha*A_il) * B_lj ),
// but the O(mnk) work runs on v_mfma_i32_32x32x32_i8 (5 Pop/s dense on MI355X, 64x the fp64 matrix rate) and only the
// O(mn) carry-propagate + round touches a long accumulator:
//
//   1. scan (gemm_scan.hip.h): per row i of A' = fl(alpha*A) a scale ea_i with |A'_il| < 2^ea_i and the lowest set bit;
//      likewise per column j of B.  k_i8_decide turns the widest span into slice counts sa, sb ON THE DEVICE.
//   2. slice ONCE (k_i8_slice_*): every entry is the integer X = x / 2^ua_i (ua_i = ea_i - 8*sa + 2), |X| < 2^(8sa-2),
//      written as sa balanced base-256 digits d_t in [-128, 127], X = sum_t d_t 256^t -- integer shifts only, no
//      rounding anywhere.  Digit t of all entries forms an int8 "plane".  Planes are stored tile-major
//      ([64-row tile][64-byte k chunk][plane][4 KiB]) so that the GEMM kernel's global->LDS traffic is linear 4 KiB
//      copies, with the 16-byte XOR swizzle the MFMA fragment reads need to be bank-conflict free already applied.
//   3. contract (k_gemm_i8): a workgroup owns a 64 x 64 block of C, each of its 4 waves a 32 x 32 tile.  All digit
//      pairs (p, q) with the same p + q share one int32 accumulator tile: |d_p d_q| <= 2^14, at most 8 pairs per
//      group, k <= 8192 per pass  =>  |sum| <= 2^30: every MFMA partial sum is an exact integer.  The slice counts are
//      read from device memory; digit pairs beyond (sa, sb) are skipped by wave-uniform scalar branches, so one
//      generic body serves every operand width up to 8 digits per pass with exactly sa*sb MFMAs per 32x32x32 block; pairs
//      of (nearly) equal width -- the usual case -- are padded to a common count and run a fully unrolled, LDS-DMA staged,
//      software-pipelined body (4 .. 9 digits per pass).
//   4. the 15 group sums of a C entry are added, shifted by 8 bits per group, into a 192-bit two's-complement integer
//      held in registers: the Kulisch accumulator of that entry, shrunk to the window the scales allow.  Operands
//      wider than 8 digits (ill-conditioned data: up to 16) or k > 8192 take several passes; each pass adds its
//      192-bit integer, shifted, into a 320-bit accumulator per entry in global memory and k_i8_finish rounds.
//   5. round ONCE: correctly (round-to-nearest-even) or, for EXBLAS_ROUND=reference, by re-cutting the integer into
//      the reference's 41 canonical limbs and running its Round() restated in superacc.hip.h -- so every (fpe,
//      early_exit) variant and both rounding modes take this path; only Inf/NaN/subnormal inputs, exponents outside
//      +-300 or spans beyond 16 digits fall back to the scalar kernel (blas3.hip), again decided on the device.
//
// Nothing synchronises with the host: kernels that turn out not to be needed (passes beyond the digit count, the
// scalar fallback) are launched anyway and exit at their first instruction, so exblas_exgemm_dev can be captured into
// a hipGraph.  Results are bit-identical to the scalar kernel, the oracle and MPFR (tests/test_gpu_blas23.py).
#include "gemm_fixed.hip.h"

namespace exb {

// ---------------------------------------------------------------------------------------------
// decide: slice counts and path, on the device
// ---------------------------------------------------------------------------------------------
__global__ void k_i8_decide(int *info, int scap)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    int path = PATH_I8;
    if (info[INFO_FLAGS]) path = PATH_SCALAR;  // Inf / NaN / subnormal input
    const bool nonzero = info[INFO_EMAX] >= info[INFO_EMIN];
    if (nonzero && (info[INFO_EMIN] < -I8_ERANGE || info[INFO_EMAX] > I8_ERANGE)) path = PATH_SCALAR;
    // |X| < 2^(8s-2) must hold with X a multiple of the lowest set bit: span + 2 bits in s digits
    int sa = (info[INFO_NEED_A] + 2 + 7) / 8, sb = (info[INFO_NEED_B] + 2 + 7) / 8;
    sa = sa < 1 ? 1 : sa;
    sb = sb < 1 ? 1 : sb;
    if (sa > scap || sb > scap) path = PATH_SCALAR;
    // Digit blocks.  A pass contracts bs x bs digit pairs with 2 bs - 1 accumulator tiles per wave (bs <= 8).
    // Operands of (nearly) equal width -- the usual case: both come from one distribution -- are padded with zero
    // digits to a common count that splits into equal blocks, so that the fully unrolled, software-pipelined pass body
    // for that block size runs (i8_pass_exact); anything else takes 8 x 8 blocks through the generic body.
    int bs = I8_SMAX, exact = 0;
    const int hi = sa > sb ? sa : sb, lo = sa > sb ? sb : sa;
    if (lo >= hi - 1 && hi >= 4) {
        const int padded = hi <= I8_XMAX ? hi : 2 * ((hi + 1) / 2);
        if (padded <= scap) {
            sa = sb = padded;
            bs = padded <= I8_XMAX ? padded : padded / 2;
            exact = 1;
        }
    }
    info[INFO_SA] = sa;
    info[INFO_SB] = sb;
    info[INFO_PATH] = path;
    info[INFO_BS] = bs;
    info[INFO_EXACT] = exact;
}

// ---------------------------------------------------------------------------------------------
// slicing
// ---------------------------------------------------------------------------------------------
// balanced base-256 digits, least significant first: d = (int8)(X & 255), X = (X - d) >> 8
template <int S>
__device__ __forceinline__ void digits(I128 X, int s, signed char (&d)[S])
{
#pragma unroll
    for (int t = 0; t < S; ++t) {
        signed char v = 0;
        if (t < s) {
            v = (signed char)(X.lo & 0xffull);
            // X -= v (sign-extended), then arithmetic shift right by 8
            const unsigned long long sub = (unsigned long long)(long long)v;
            const unsigned long long lo = X.lo - sub;
            const long long hi = X.hi - (long long)(v < 0 ? -1 : 0) - (X.lo < sub ? 1 : 0);
            X.lo = (lo >> 8) | ((unsigned long long)hi << 56);
            X.hi = hi >> 8;
        }
        d[t] = v;
    }
}

// Source vectors CONTIGUOUS along k (A for 'N', B for 'T'): element (v, l) at src[v*ld + l].
// One workgroup per (64-vector tile, 64-k chunk); thread = (vector tid/4, 16 consecutive k).
template <int S>
__global__ void __launch_bounds__(256) k_i8_slice_contig(const double *__restrict__ src, long long ld, int nvec, int len,
                                                         double scale, const int *__restrict__ E,
                                                         const int *__restrict__ info, int which,
                                                         signed char *__restrict__ planes)
{
    if (info[INFO_PATH] != PATH_I8) return;
    const int s = info[which ? INFO_SB : INFO_SA];
    const int kc = blockIdx.x, vt = blockIdx.y, KC = gridDim.x;
    const int r = threadIdx.x >> 2, seg = threadIdx.x & 3;
    const int v = vt * I8_T + r, l0 = kc * I8_T + seg * 16;
    const int u = (v < nvec ? E[v] : 0) - 8 * s + 2;
    signed char dg[16][S];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int l = l0 + e;
        const double x = (v < nvec && l < len) ? scale * src[(long long)v * ld + l] : 0.0;
        digits<S>(to_fixed(x, u), s, dg[e]);
    }
    signed char *tile = planes + ((size_t)vt * KC + kc) * (size_t)s * I8_TILE;
#pragma unroll
    for (int t = 0; t < S; ++t) {
        if (t < s) {
            union { v4i_t v; signed char b[16]; } pk;
#pragma unroll
            for (int e = 0; e < 16; ++e) pk.b[e] = dg[e][t];
            *(v4i_t *)(tile + (size_t)t * I8_TILE + tile_off(r, seg * 16)) = pk.v;
        }
    }
}

// Source vectors STRIDED along k (A for 'T', B for 'N'): element (v, l) at src[l*ld + v]; adjacent vectors are
// contiguous, so a wave reads 64 vectors x one l per load.  thread = (vector tid%64, 16 consecutive k).
template <int S>
__global__ void __launch_bounds__(256) k_i8_slice_strided(const double *__restrict__ src, long long ld, int nvec, int len,
                                                          double scale, const int *__restrict__ E,
                                                          const int *__restrict__ info, int which,
                                                          signed char *__restrict__ planes)
{
    if (info[INFO_PATH] != PATH_I8) return;
    const int s = info[which ? INFO_SB : INFO_SA];
    const int kc = blockIdx.x, vt = blockIdx.y, KC = gridDim.x;
    const int r = threadIdx.x & 63, seg = threadIdx.x >> 6;
    const int v = vt * I8_T + r, l0 = kc * I8_T + seg * 16;
    const int u = (v < nvec ? E[v] : 0) - 8 * s + 2;
    signed char dg[16][S];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int l = l0 + e;
        const double x = (v < nvec && l < len) ? scale * src[(long long)l * ld + v] : 0.0;
        digits<S>(to_fixed(x, u), s, dg[e]);
    }
    signed char *tile = planes + ((size_t)vt * KC + kc) * (size_t)s * I8_TILE;
#pragma unroll
    for (int t = 0; t < S; ++t) {
        if (t < s) {
            union { v4i_t v; signed char b[16]; } pk;
#pragma unroll
            for (int e = 0; e < 16; ++e) pk.b[e] = dg[e][t];
            *(v4i_t *)(tile + (size_t)t * I8_TILE + tile_off(r, seg * 16)) = pk.v;
        }
    }
}

// ---- pieces shared by the pass bodies ------------------------------------------------------------------------
struct PassArgs {
    int m, n, KC, kc0, kc1, ta0, tb0;
    int ty0, ty_cnt;     // this launch covers the tile rows [ty0, ty0 + ty_cnt) (row chunks of the sharded GEMM)
    const signed char *PA, *PB;
    const int *EA, *EB;
    double beta;
    double *c;
    long long ldc;
    int round_mode;
    unsigned long long *W;
    int sa_all, sb_all;  // digits of the whole operands (plane strides, units)
    bool single;         // this pass is the whole product: round and write C
};

// epilogue: G group sums -> one 192-bit integer per entry; C layout of the 32x32 MFMA:
// col = lane & 31, row = 8 * (r / 4) + 4 * (lane / 32) + (r % 4)
template <int G>
__device__ __forceinline__ void i8_epilogue(const PassArgs &a, const v16i_t (&acc)[G], int ty, int tx, int wr, int wc)
{
    const int lane = threadIdx.x & 63, half = lane >> 5;
    const int gj = tx * I8_T + wc + (lane & 31);
    const int ebj = gj < a.n ? a.EB[gj] : 0;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        unsigned long long w3[3] = {0ull, 0ull, 0ull};
        static_for_i8<0, G>([&](auto gc) {
            constexpr int g = decltype(gc)::value;
            wide_add_c<3, 8 * g>(w3, (long long)acc[g][r]);
        });
        const int gi = ty * I8_T + wr + 8 * (r >> 2) + 4 * half + (r & 3);
        if (gi < a.m && gj < a.n) {
            // unit of this pass's integer: 2^(ua + ub) * 256^(ta0 + tb0), ua = ea - 8 sa + 2
            const int u0 = a.EA[gi] - 8 * a.sa_all + 2 + ebj - 8 * a.sb_all + 2;
            if (a.single) {
                const double sres = a.round_mode ? wide_round_reference<3>(w3, u0 + 8 * (a.ta0 + a.tb0))
                                                 : wide_round_n<3>(w3, u0 + 8 * (a.ta0 + a.tb0));
                double *cij = a.c + (long long)gi * a.ldc + gj;
                *cij = (a.beta == 0.0) ? sres : a.beta * (*cij) + sres;
            } else {
                unsigned long long *wg = a.W + ((size_t)gi * a.n + gj) * I8_NWG;
                unsigned long long acc5[I8_NWG];
#pragma unroll
                for (int i = 0; i < I8_NWG; ++i) acc5[i] = wg[i];
                wide_add_v<I8_NWG>(acc5, w3, 8 * (a.ta0 + a.tb0));
#pragma unroll
                for (int i = 0; i < I8_NWG; ++i) wg[i] = acc5[i];
            }
        }
    }
}

// LDS of one workgroup: [buffer][A planes | B planes][256 x 16 B] = 128 KiB
typedef v4i_t (*LdsBuf)[2 * I8_XMAX * (I8_TILE / 16)];

// EXACT body: SA = SB = B digits in this pass, everything unrolled, software-pipelined:
//   * staging by LDS-DMA (global_load_lds_dwordx4): a wave-instruction copies 64 x 16 B = this wave's quarter of one
//     4 KiB plane straight into LDS (the LDS image is lane-linear, exactly the tile-major layout the slicer wrote,
//     swizzle included) -- no staging registers, no ds_write.  Chunk kc+2 is issued into the buffer chunk kc occupied
//     right after the barrier that retires its last fragment reads, and has a full chunk of MFMAs to land before the
//     vmcnt(0) + barrier that precedes its first read;
//   * per k chunk (64 bytes = two MFMA k-steps) a wave issues 2 x B*B MFMAs; the fragments of a k-step are read from
//     LDS into one of two register sets while the MFMAs of the previous k-step run; one barrier per chunk;
//   * explicit issue order: per group one fragment read for the NEXT k-step, (in the second k-step) one LDS-DMA piece,
//     and the next few MFMAs; a scheduling barrier pins each group (left alone, the scheduler gathers the DMAs and
//     their M0 writes at one end of the region and the matrix pipe idles while they issue).
// History at 8192^3, 8 x 8 digits: register-staged with ds_write_b128, memory instructions clustered 30.0 ms; one per
// MFMA gap 25.9-26.9 ms; LDS-DMA 25.8-26.4 ms (5 x 5 digit passes: 47.6 -> 44.8 ms).
template <int B>
__device__ __forceinline__ void i8_pass_exact(const PassArgs &a, LdsBuf lds)
{
    constexpr int G = 2 * B - 1;
    constexpr int NMEM = 2 * B;  // fragment reads per k-step = LDS-DMA pieces per chunk
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int gx = (a.n + I8_T - 1) / I8_T;
    int ty, tx;
    tile_of_block(blockIdx.x, gridDim.x, a.ty_cnt, gx, &ty, &tx);
    ty += a.ty0;
    const int wr = (wave >> 1) * 32, wc = (wave & 1) * 32;
    const signed char *pa = a.PA + ((size_t)ty * a.KC * a.sa_all + a.ta0) * I8_TILE + (size_t)tid * 16;
    const signed char *pb = a.PB + ((size_t)tx * a.KC * a.sb_all + a.tb0) * I8_TILE + (size_t)tid * 16;
    const size_t stride_a = (size_t)a.sa_all * I8_TILE, stride_b = (size_t)a.sb_all * I8_TILE;
    const int arow = wr + (lane & 31), brow = wc + (lane & 31), half = lane >> 5;
    int aoff[2], boff[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        aoff[ks] = tile_off(arow, (2 * ks + half) * 16) >> 4;
        boff[ks] = tile_off(brow, (2 * ks + half) * 16) >> 4;
    }
    v16i_t acc[G];
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[g][r] = 0;
    auto clampk = [&](int kc) { return kc < a.kc1 ? kc : a.kc1 - 1; };
    auto dma_piece = [&](auto ic, int kc, int buf) {
        constexpr int i = decltype(ic)::value;
        if constexpr (i < B)
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void *)(pa + kc * stride_a + (size_t)i * I8_TILE),
                (__attribute__((address_space(3))) void *)&lds[buf][i * 256 + wave * 64], 16, 0, 0);
        else
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void *)(pb + kc * stride_b + (size_t)(i - B) * I8_TILE),
                (__attribute__((address_space(3))) void *)&lds[buf][(I8_XMAX + i - B) * 256 + wave * 64], 16, 0, 0);
    };
    auto dma = [&](int kc, int buf) { static_for_i8<0, NMEM>([&](auto ic) { dma_piece(ic, kc, buf); }); };
    auto fload = [&](int buf, int ks, v4i_t (&fa)[B], v4i_t (&fb)[B]) {
#pragma unroll
        for (int q = 0; q < B; ++q) fb[q] = lds[buf][(I8_XMAX + q) * 256 + boff[ks]];
#pragma unroll
        for (int p = 0; p < B; ++p) fa[p] = lds[buf][p * 256 + aoff[ks]];
    };
    auto kstep = [&](const v4i_t (&ca)[B], const v4i_t (&cb)[B], v4i_t (&na)[B], v4i_t (&nb)[B], int rbuf, int rks,
                     bool with_dma, int dkc, int dbuf) {
        constexpr int PER = B * B / NMEM, EXTRA = B * B % NMEM;
        static_for_i8<0, NMEM>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            if constexpr (i < B) nb[i] = lds[rbuf][(I8_XMAX + i) * 256 + boff[rks]];
            else na[i - B] = lds[rbuf][(i - B) * 256 + aoff[rks]];
            if (with_dma) dma_piece(ic, dkc, dbuf);
            constexpr int m0 = i * PER + (i < EXTRA ? i : EXTRA), m1 = m0 + PER + (i < EXTRA ? 1 : 0);
            static_for_i8<m0, m1>([&](auto mc) {
                constexpr int mm = decltype(mc)::value, p = mm / B, q = mm % B;
                acc[p + q] = __builtin_amdgcn_mfma_i32_32x32x32_i8(ca[p], cb[q], acc[p + q], 0, 0, 0);
            });
            __builtin_amdgcn_sched_barrier(0);
        });
    };

    v4i_t fa0[B], fb0[B], fa1[B], fb1[B];
    dma(a.kc0, 0);
    __syncthreads();  // vmcnt(0) + lgkmcnt(0) + barrier
    fload(0, 0, fa0, fb0);
    dma(clampk(a.kc0 + 1), 1);
    for (int kc = a.kc0; kc < a.kc1; ++kc) {
        const int buf = (kc - a.kc0) & 1;
        kstep(fa0, fb0, fa1, fb1, buf, 1, false, 0, 0);
        __syncthreads();  // chunk kc+1 has landed in the other buffer; nobody reads this one any more
        // past the last chunk the DMA repeats the last chunk into a buffer nobody reads: the loop body stays branch-free
        kstep(fa1, fb1, fa0, fb0, buf ^ 1, 0, true, clampk(kc + 2), buf);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the clamped tail loads must not outlive the LDS allocation
    i8_epilogue<G>(a, acc, ty, tx, wr, wc);
}

// GENERIC body: up to 8 x 8 digits in this pass, counts known at run time only; digit pairs beyond (sa, sb) are skipped
// by wave-uniform scalar branches.  Serves operand pairs whose digit counts differ by more than one.
__device__ __forceinline__ void i8_pass_generic(const PassArgs &a, int sa, int sb, LdsBuf lds)
{
    constexpr int G = 2 * I8_SMAX - 1;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int gx = (a.n + I8_T - 1) / I8_T;
    int ty, tx;
    tile_of_block(blockIdx.x, gridDim.x, a.ty_cnt, gx, &ty, &tx);
    ty += a.ty0;
    const int wr = (wave >> 1) * 32, wc = (wave & 1) * 32;
    const signed char *pa = a.PA + ((size_t)ty * a.KC * a.sa_all + a.ta0) * I8_TILE + (size_t)tid * 16;
    const signed char *pb = a.PB + ((size_t)tx * a.KC * a.sb_all + a.tb0) * I8_TILE + (size_t)tid * 16;
    const size_t stride_a = (size_t)a.sa_all * I8_TILE, stride_b = (size_t)a.sb_all * I8_TILE;

    v4i_t ra[I8_SMAX], rb[I8_SMAX];
    auto gload = [&](int kc) {
#pragma unroll
        for (int p = 0; p < I8_SMAX; ++p)
            if (p < sa) ra[p] = *(const v4i_t *)(pa + kc * stride_a + (size_t)p * I8_TILE);
#pragma unroll
        for (int q = 0; q < I8_SMAX; ++q)
            if (q < sb) rb[q] = *(const v4i_t *)(pb + kc * stride_b + (size_t)q * I8_TILE);
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int p = 0; p < I8_SMAX; ++p)
            if (p < sa) lds[buf][p * 256 + tid] = ra[p];
#pragma unroll
        for (int q = 0; q < I8_SMAX; ++q)
            if (q < sb) lds[buf][(I8_XMAX + q) * 256 + tid] = rb[q];
    };
    v16i_t acc[G];
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[g][r] = 0;
    const int arow = wr + (lane & 31), brow = wc + (lane & 31), half = lane >> 5;
    int aoff[2], boff[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        aoff[ks] = tile_off(arow, (2 * ks + half) * 16) >> 4;
        boff[ks] = tile_off(brow, (2 * ks + half) * 16) >> 4;
    }
    gload(a.kc0);
    lstore(0);
    __syncthreads();
    for (int kc = a.kc0; kc < a.kc1; ++kc) {
        const int buf = (kc - a.kc0) & 1;
        if (kc + 1 < a.kc1) gload(kc + 1);  // in flight while this chunk is contracted
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            v4i_t fa[I8_SMAX], fb[I8_SMAX];
#pragma unroll
            for (int p = 0; p < I8_SMAX; ++p)
                if (p < sa) fa[p] = lds[buf][p * 256 + aoff[ks]];
#pragma unroll
            for (int q = 0; q < I8_SMAX; ++q)
                if (q < sb) fb[q] = lds[buf][(I8_XMAX + q) * 256 + boff[ks]];
#pragma unroll
            for (int p = 0; p < I8_SMAX; ++p) {
                if (p < sa) {
#pragma unroll
                    for (int q = 0; q < I8_SMAX; ++q)
                        if (q < sb) acc[p + q] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa[p], fb[q], acc[p + q], 0, 0, 0);
                }
            }
        }
        if (kc + 1 < a.kc1) lstore(buf ^ 1);  // that buffer was last read one iteration ago, before the barrier below
        __syncthreads();
    }
    i8_epilogue<G>(a, acc, ty, tx, wr, wc);
}

// One pass: C block (64 x 64 per workgroup) += digits [ta0, ta0+bs) of A  x  digits [tb0, tb0+bs) of B over k chunks
// [kc0, kc1), where (ta0, tb0) = (ia, ib) * bs and bs is the digit block size the device chose (k_i8_decide).
// allow_single: with one pass in all this launch rounds and writes C itself; otherwise it adds into W.
// Two kernels share the argument list: k_gemm_i8 holds the unrolled bodies (it runs when the decision says "exact":
// equal digit counts in every pass), k_gemm_i8g the generic body (all other operand pairs); each exits at once when
// the other one is meant.  They are separate kernels because the generic body's scalar bookkeeping spills thousands of
// SGPRs into VGPR lanes, which the 9-digit unrolled body (272 accumulator + 144 fragment registers) cannot spare.
__device__ __forceinline__ bool i8_pass_args(PassArgs &a, const int *info, int m, int n, int ty0, int ty_cnt, int KC, int kc0,
                                             int kc1, int ia, int ib, const signed char *PA, const signed char *PB,
                                             const int *EA, const int *EB, double beta, double *c, long long ldc,
                                             int round_mode, int allow_single, unsigned long long *W, int *sa, int *sb)
{
    a.sa_all = info[INFO_SA];
    a.sb_all = info[INFO_SB];
    const int bs = info[INFO_BS];
    a.ta0 = ia * bs;
    a.tb0 = ib * bs;
    if (a.ta0 >= a.sa_all || a.tb0 >= a.sb_all) return false;
    *sa = min(bs, a.sa_all - a.ta0);
    *sb = min(bs, a.sb_all - a.tb0);
    a.m = m; a.n = n; a.KC = KC; a.kc0 = kc0; a.kc1 = kc1; a.ty0 = ty0; a.ty_cnt = ty_cnt;
    a.PA = PA; a.PB = PB; a.EA = EA; a.EB = EB;
    a.beta = beta; a.c = c; a.ldc = ldc; a.round_mode = round_mode; a.W = W;
    a.single = allow_single && a.sa_all <= bs && a.sb_all <= bs;
    return true;
}

#define I8_PASS_PARAMS                                                                                                  \
    int m, int n, int ty0, int ty_cnt, int KC, int kc0, int kc1, int ia, int ib, const signed char *__restrict__ PA,     \
        const signed char *__restrict__ PB, const int *__restrict__ info, const int *__restrict__ EA,                   \
        const int *__restrict__ EB, double beta, double *__restrict__ c, long long ldc, int round_mode, int allow_single, \
        unsigned long long *__restrict__ W

__global__ void __launch_bounds__(256, 1) k_gemm_i8(I8_PASS_PARAMS)
{
    __shared__ v4i_t lds[2][2 * I8_XMAX * (I8_TILE / 16)];
    if (info[INFO_PATH] != PATH_I8 || !info[INFO_EXACT]) return;
    PassArgs a;
    int sa, sb;
    if (!i8_pass_args(a, info, m, n, ty0, ty_cnt, KC, kc0, kc1, ia, ib, PA, PB, EA, EB, beta, c, ldc, round_mode,
                      allow_single, W, &sa, &sb))
        return;
    switch (info[INFO_BS]) {  // sa == sb == bs in every pass
    case 4: i8_pass_exact<4>(a, lds); break;
    case 5: i8_pass_exact<5>(a, lds); break;
    case 6: i8_pass_exact<6>(a, lds); break;
    case 7: i8_pass_exact<7>(a, lds); break;
    case 9: i8_pass_exact<9>(a, lds); break;
    default: i8_pass_exact<8>(a, lds); break;
    }
}

__global__ void __launch_bounds__(256, 1) k_gemm_i8g(I8_PASS_PARAMS)
{
    __shared__ v4i_t lds[2][2 * I8_XMAX * (I8_TILE / 16)];
    if (info[INFO_PATH] != PATH_I8 || info[INFO_EXACT]) return;
    PassArgs a;
    int sa, sb;
    if (!i8_pass_args(a, info, m, n, ty0, ty_cnt, KC, kc0, kc1, ia, ib, PA, PB, EA, EB, beta, c, ldc, round_mode,
                      allow_single, W, &sa, &sb))
        return;
    i8_pass_generic(a, sa, sb, lds);
}
#undef I8_PASS_PARAMS

// multi-pass epilogue: round the 320-bit accumulators
__global__ void __launch_bounds__(256) k_i8_finish(int row0, int row1, int n, const int *__restrict__ info,
                                                   const int *__restrict__ EA, const int *__restrict__ EB, double beta,
                                                   double *__restrict__ c, long long ldc, int round_mode, int force_multi,
                                                   const unsigned long long *__restrict__ W)
{
    if (info[INFO_PATH] != PATH_I8) return;
    const int sa = info[INFO_SA], sb = info[INFO_SB], bs = info[INFO_BS];
    if (!force_multi && sa <= bs && sb <= bs) return;  // the single pass wrote C itself
    const long long loc = (long long)blockIdx.x * 256 + threadIdx.x;
    if (loc >= (long long)(row1 - row0) * n) return;
    const int gi = row0 + (int)(loc / n), gj = (int)(loc % n);
    const size_t idx = (size_t)gi * n + gj;
    unsigned long long w[I8_NWG];
#pragma unroll
    for (int i = 0; i < I8_NWG; ++i) w[i] = W[idx * I8_NWG + i];
    const int u0 = EA[gi] - 8 * sa + 2 + EB[gj] - 8 * sb + 2;
    const double s = round_mode ? wide_round_reference<I8_NWG>(w, u0) : wide_round_n<I8_NWG>(w, u0);
    double *cij = c + (long long)gi * ldc + gj;
    *cij = (beta == 0.0) ? s : beta * (*cij) + s;
}

__global__ void __launch_bounds__(256) k_i8_zero_w(long long words, const int *__restrict__ info, int force_multi,
                                                   unsigned long long *__restrict__ W)
{
    if (info[INFO_PATH] != PATH_I8) return;
    if (!force_multi && info[INFO_SA] <= info[INFO_BS] && info[INFO_SB] <= info[INFO_BS]) return;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < words; i += (long long)gridDim.x * 256) W[i] = 0ull;
}

// ---------------------------------------------------------------------------------------------
// host side: a pure sequence of launches
// ---------------------------------------------------------------------------------------------
// exgemm_i8_prepare: scan, decide, slice, zero the accumulators -- everything that concerns the WHOLE operands.  On
// return plan->ok says whether the int8 path was enqueued (it may still decide, on the device, to leave the work to
// the scalar kernel: the caller enqueues that one predicated on plan->info[INFO_PATH]); !ok: not attempted (k == 0,
// workspace unavailable) -> the caller runs the scalar kernel unconditionally.
// exgemm_i8_rows: the contraction passes (and the rounding of multi-pass accumulators) for the rows [row0, row1) of
// C, row0 a multiple of 64.  A row-sharded caller (comm.hip) runs it chunk by chunk and ships every finished chunk
// while the next one is computed; the operands are scanned and sliced once for all chunks.
hipError_t exgemm_i8_prepare(Ctx &c, char transa, char transb, int m, int n, int k, double alpha, const double *a, int lda,
                             const double *b, int ldb, double beta, double *cmat, int ldc, int round_mode,
                             hipStream_t st, I8Plan *plan)
{
    plan->ok = false;
    if (k <= 0 || m <= 0 || n <= 0) return hipSuccess;
    if (c.gemm_path == 4 || (c.gemm_path == 0 && m >= CRT_MIN_EDGE && n >= CRT_MIN_EDGE)) {
        hipError_t e = exgemm_crt_prepare(c, transa, transb, m, n, k, alpha, a, lda, b, ldb, beta, cmat, ldc, round_mode, st,
                                          plan);
        // not enqueued although nothing failed = no memory even for 12 moduli: try the digit-slice workspace below
        if (e != hipSuccess || plan->ok || c.gemm_path == 4) return e;
    }
    const int ta = (transa == 'T' || transa == 't'), tb = (transb == 'T' || transb == 't');
    const int gy = (m + I8_T - 1) / I8_T, gx = (n + I8_T - 1) / I8_T, KC = (k + I8_T - 1) / I8_T;
    int scap = c.gemm_max_slices > 0 ? c.gemm_max_slices : I8_SCAP;
    if (scap > I8_SCAP) scap = I8_SCAP;
    const int kpasses = (k + I8_KPASS - 1) / I8_KPASS;
    const bool maybe_multi = scap > I8_SMAX || kpasses > 1;
    // workspace: info | EA EB LA LB | planes of A | planes of B | W
    size_t off = 0;
    auto take = [&](size_t bytes) {
        const size_t o = off;
        off += (bytes + 255) & ~(size_t)255;
        return o;
    };
    const size_t o_info = take(sizeof(int) * INFO_WORDS);
    const size_t o_e = take(sizeof(int) * 2 * ((size_t)m + n));
    const size_t o_pa = take((size_t)gy * KC * scap * I8_TILE);
    const size_t o_pb = take((size_t)gx * KC * scap * I8_TILE);
    const size_t o_w = take(maybe_multi ? (size_t)m * n * I8_NWG * sizeof(unsigned long long) : 0);
    hipError_t e = hipSuccess;
    char *base = (char *)workspace(c, off, st, &e);
    if (!base) {
        if (e == hipErrorStreamCaptureUnsupported) return e;  // the caller must reserve before capturing
        return hipSuccess;                                    // out of memory: scalar kernel
    }
    int *info = (int *)(base + o_info);
    int *EA = (int *)(base + o_e), *EB = EA + m, *LA = EB + n, *LB = LA + m;
    signed char *PA = (signed char *)(base + o_pa), *PB = (signed char *)(base + o_pb);
    unsigned long long *W = (unsigned long long *)(base + o_w);

    // ---- scan (shared with the fp64-slice path) ----
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
    hipLaunchKernelGGL(k_i8_decide, dim3(1), dim3(64), 0, st, info, scap);

    // ---- slice once ----
    if (!ta)
        hipLaunchKernelGGL((k_i8_slice_contig<I8_SCAP>), dim3(KC, gy), dim3(256), 0, st, a, (long long)lda, m, k, alpha,
                           EA, info, 0, PA);
    else
        hipLaunchKernelGGL((k_i8_slice_strided<I8_SCAP>), dim3(KC, gy), dim3(256), 0, st, a, (long long)lda, m, k, alpha,
                           EA, info, 0, PA);
    if (!tb)
        hipLaunchKernelGGL((k_i8_slice_strided<I8_SCAP>), dim3(KC, gx), dim3(256), 0, st, b, (long long)ldb, n, k, 1.0,
                           EB, info, 1, PB);
    else
        hipLaunchKernelGGL((k_i8_slice_contig<I8_SCAP>), dim3(KC, gx), dim3(256), 0, st, b, (long long)ldb, n, k, 1.0,
                           EB, info, 1, PB);
    const int force_multi = kpasses > 1 ? 1 : 0;
    if (maybe_multi)
        hipLaunchKernelGGL(k_i8_zero_w, dim3(c.num_cu * 8), dim3(256), 0, st, (long long)m * n * I8_NWG, info,
                           force_multi, W);
    plan->ok = true;
    plan->m = m; plan->n = n; plan->KC = KC; plan->kpasses = kpasses;
    plan->dblocks = (scap + I8_SMAX - 1) / I8_SMAX;
    plan->force_multi = force_multi; plan->maybe_multi = maybe_multi ? 1 : 0;
    plan->info = info; plan->EA = EA; plan->EB = EB; plan->PA = PA; plan->PB = PB; plan->W = W;
    plan->beta = beta; plan->c = cmat; plan->ldc = ldc; plan->round_mode = round_mode;
    c.gemm_info_dev = info;
    return hipGetLastError();
}

// contraction passes over (digit block of A, digit block of B, k block) for the rows [row0, row1); the device skips
// what the data does not need.  With one k pass and one digit block on both sides the (0, 0) pass rounds and writes C
// itself; otherwise k_i8_finish rounds the 320-bit accumulators of these rows.
hipError_t exgemm_i8_rows(const I8Plan &p, int row0, int row1, hipStream_t st)
{
    if (row1 <= row0) return hipSuccess;
    if (p.crt) return exgemm_crt_rows(p, row0, row1, st);
    const int gx = (p.n + I8_T - 1) / I8_T;
    const int ty0 = row0 / I8_T, ty_cnt = (row1 - row0 + I8_T - 1) / I8_T;
    for (int kp = 0; kp < p.kpasses; ++kp) {
        const int kc0 = kp * (I8_KPASS / I8_T), kc1 = min(p.KC, kc0 + I8_KPASS / I8_T);
        for (int pa_ = 0; pa_ < p.dblocks; ++pa_)
            for (int pb_ = 0; pb_ < p.dblocks; ++pb_)
            {
                hipLaunchKernelGGL(k_gemm_i8, dim3(ty_cnt * gx), dim3(256), 0, st, p.m, p.n, ty0, ty_cnt, p.KC, kc0, kc1,
                                   pa_, pb_, p.PA, p.PB, p.info, p.EA, p.EB, p.beta, p.c, (long long)p.ldc, p.round_mode,
                                   p.force_multi ? 0 : 1, p.W);
                hipLaunchKernelGGL(k_gemm_i8g, dim3(ty_cnt * gx), dim3(256), 0, st, p.m, p.n, ty0, ty_cnt, p.KC, kc0, kc1,
                                   pa_, pb_, p.PA, p.PB, p.info, p.EA, p.EB, p.beta, p.c, (long long)p.ldc, p.round_mode,
                                   p.force_multi ? 0 : 1, p.W);
            }
    }
    if (p.maybe_multi)
        hipLaunchKernelGGL(k_i8_finish, dim3((unsigned)(((long long)(row1 - row0) * p.n + 255) / 256)), dim3(256), 0, st,
                           row0, row1, p.n, p.info, p.EA, p.EB, p.beta, p.c, (long long)p.ldc, p.round_mode,
                           p.force_multi, p.W);
    return hipGetLastError();
}

}  // namespace exb
