// trsv.hip -- ExTRSV: reproducible triangular solve  A x = b  /  A**T x = b  (x holds b on entry), column-major A.
//
//     x_i = fl( Round( b_i - sum_j A(i,j) * x_j ) / A(i,i) )        j before i in the substitution order
//
// with the sum exact (TwoProd into expansions + integer superaccumulator) and Round the superaccumulator rounding:
// the arithmetic of the reference kernels (src/gpu/blas/blas2/ExTRSV.lnn.Superacc.cl:254-348 lower,
// ExTRSV.unn.Superacc.cl:262-355 upper; FPE variants ExTRSV.{lnn,unn}.FPE*.cl; dispatch ExTRSV.cpp:70-123).
// Unlike the reference kernels, which ignore them (isunit = 0 at lnn.Superacc.cl:272; no transposed kernel exists),
// transa, diag, incx and lda are honoured.
//
// Shape of the computation on MI355X.  The n rounded divisions form one dependency chain -- x_i needs x_{i-1} --
// so the solve is LATENCY bound, not HBM bound: the n^2/2 matrix elements (0.56 ms of HBM time at n = 32768) are
// consumed far faster than the chain advances.  The design therefore keeps everything off the chain that can be:
//   * one workgroup per block-row of 64 rows (lane = row, 4 waves split the 64 columns of a tile); block-rows are
//     handed out by an atomic ticket in dependency order, so a workgroup only ever waits for workgroups that
//     already run (no deadlock when there are more block-rows than resident workgroups);
//   * the off-diagonal tiles of a block-row are consumed as soon as the x block they need is published, i.e. all
//     but the last one long before the chain arrives; their matrix elements are loaded BEFORE the wait;
//   * per row the running sum lives in a register expansion per (wave, lane) and spills to the row's private
//     integer accumulator in LDS (68 limbs, pitch 69 words: a wave adding to the same limb of 64 rows, the common
//     case, touches 64 different banks);
//   * the diagonal block is solved by ONE wave.  A lone wave retires one fp64 instruction per 8 shader clocks (3.35 ns,
//     tools/micro/lone_wave.hip), dependent or not, so the chain step is priced in instructions: each row's exact value is kept as a 3-level TwoSum expansion in the
//     lane's registers (plus whatever the integer accumulator still holds, with a bound on it); a new x costs every
//     later row one TwoProd and two short cascades, and at its turn a row is folded to one double whose correct
//     rounding is certified by letting the adder round (|s| +- bound must return |s|).  Only rows that fail the
//     certificate -- ties, cancellation to the noise floor, huge / subnormal / non-finite values -- go through the
//     wave-parallel integer rounding (finish_wave<false>).  About 110 instructions per row instead of 400.
// Cross-workgroup traffic: finished x values are posted to a mailbox (one double per row, preset to a reserved NaN
// pattern) with agent-scope relaxed atomic stores and fetched with agent-scope atomic loads, which bypass the
// non-coherent cache levels: the value is its own ready flag -- no fence, counter or barrier on the hand-off.
#include "exblas_internal.h"
#include "fpe.hip.h"

namespace exb {
namespace {

constexpr int TB = 64;            // rows per block-row = lanes of a wave
constexpr int TW = 4;             // waves per workgroup
constexpr int TCW = TB / TW;      // tile columns per wave
constexpr int TPITCH = NL + 1;    // int64 words per row accumulator (odd)
static_assert(TCW == 16, "the tile loop below is written for 2 x 8 columns per wave");

// one row's accumulator in LDS, limbs contiguous
struct RowSink {
    long long *row;
    unsigned *flagp;
    int *touched;  // workgroup-wide: "some integer accumulator of this block holds something"
    __device__ __forceinline__ void add(double x)
    {
        unsigned f = 0;
        lds_add<1>(row, x, f);
        if (f) atomicOr(flagp, f);  // Inf / NaN: rare
        *touched = 1;
    }
    __device__ __forceinline__ void note(unsigned) {}  // no flag channel per row of x
};

constexpr int DG_M = 3;    // expansion levels per row in the diagonal phase
constexpr int DG_PUB = 4;  // non-empty expansion levels a tile-phase wave hands over as doubles (more: through the accumulator)
constexpr int DG_K = 6;    // limbs below a row's leading limb that row_to_fpe moves into the expansion

// x enters levels FROM..DG_M-1 of g; what is left after the last level goes to the row's integer accumulator
// (exactly) and into its bound B (rounded up)
template <int FROM>
__device__ __forceinline__ void fpe_push(double (&g)[DG_M], double &B, double x, RowSink &sink)
{
#pragma unroll
    for (int k = FROM; k < DG_M; ++k) {
        double r;
        g[k] = two_sum(g[k], x, r);
        x = r;
    }
    if (__builtin_expect(x != 0.0, 0)) {
        sink.add(x);
        B += fabs(x) * 1.0000001;
    }
}

// Moves this lane's row of the LDS accumulator into the expansion g: the leading non-zero limb and the DG_K limbs
// below it (at least 193 bits), top down; lower limbs stay where they are and only enter the bound B.  Rows that
// saw Inf/NaN or hold anything at limb 62 or above (|value| >= 2^910) are left alone with B = inf, which sends
// them to the integer path.  All 64 lanes must call it.
__device__ inline void row_to_fpe(long long *row, unsigned rowflags, double (&g)[DG_M], double &B, RowSink &sink)
{
#pragma unroll
    for (int k = 0; k < DG_M; ++k) g[k] = 0.0;
    B = rowflags ? __builtin_inf() : 0.0;
    // which limbs are non-zero in ANY row (wave-uniform): the reads are independent, so they pipeline
    unsigned long long u0 = 0ull;
    unsigned u1 = 0u;
#pragma unroll
    for (int l = 0; l < NL; ++l) {
        const bool nz = __any(row[l] != 0);
        if (l < 64) u0 |= nz ? (1ull << l) : 0ull;
        else u1 |= nz ? (1u << (l - 64)) : 0u;
    }
    int top = -1;
    for (int l = NL - 1; l >= 0; --l) {
        const bool on = l < 64 ? ((u0 >> l) & 1ull) != 0 : ((u1 >> (l - 64)) & 1u) != 0;
        if (!on) continue;  // uniform
        const long long v = row[l];
        if (v == 0) continue;
        if (l >= 62 || B == __builtin_inf()) {
            B = __builtin_inf();
            continue;
        }
        if (top < 0) top = l;
        if (top - l <= DG_K) {
            row[l] = 0;
            // v = hi * 2^32 + lo, both halves exact in fp64; limb l has weight 2^(32 l - 1074)
            const double xh = ldexp((double)(int)(v >> 32), 32 * l + 32 - 1074);
            const double xl = ldexp((double)(unsigned)v, 32 * l - 1074);
            fpe_push<0>(g, B, xh, sink);
            fpe_push<0>(g, B, xl, sink);
        } else {
            B += ldexp(1.0, 32 * l + 64 - 1074);  // |v| < 2^63 at weight 2^(32 l - 1074)
        }
    }
}

// Solution values travel between workgroups through a mailbox xq[logical row], preset to a NaN pattern no result
// can carry (the writer maps that one pattern to the canonical NaN): the value is its own "ready" flag, so a
// consumer needs ONE coherent load round trip per tile and no fence, counter or workgroup barrier.
constexpr long long XQ_EMPTY = -1ll;  // 0xFFFFFFFFFFFFFFFF
__device__ __forceinline__ void post_x(double *xq, double v)
{
    long long b = __double_as_longlong(v);
    if (b == XQ_EMPTY) b = 0x7ff8000000000000ll;
    __hip_atomic_store((long long *)xq, b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// lanes with `want` set fetch *q, polling until it has been posted; wave-uniform loop
__device__ __forceinline__ double fetch_x(const double *q, bool want)
{
    long long b = 0;
    for (;;) {
        if (want) b = __hip_atomic_load((const long long *)q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (!__any(want && b == XQ_EMPTY)) break;
        __builtin_amdgcn_s_sleep(1);  // every workgroup behind the front polls the same few lines: keep it light
    }
    return __longlong_as_double(b);
}

// ---------------------------------------------------------------------------------------------
// the exact solve.  Logical indices run in substitution order; phys() maps them to storage (reversed for
// backward substitution).  Element (row r, column c) of the logical matrix is a[phys(r)*rs + phys(c)*cs].
// sync[0]: block-row ticket, sync[2]: rows rounded by the integer path (statistics).
// ---------------------------------------------------------------------------------------------
template <int N, bool EE>
__global__ void __launch_bounds__(TB *TW) k_trsv(int n, const double *__restrict__ a, long long rs, long long cs,
                                                 double *x, long long incx, int rev, int unit, int mode, int *sync,
                                                 double *xq)
{
    __shared__ long long acc[TB * TPITCH];
    __shared__ double dg[TB * TB];  // diagonal block, dg[c * TB + r], strictly-lower part
    __shared__ unsigned rflags[TB];
    __shared__ double fpub[TW][DG_PUB][TB];  // leading expansion levels of the tile phase, wave -> diagonal wave
    __shared__ int s_row, s_touched;
    const int tid = (int)threadIdx.x, lane = tid & 63, w = tid >> 6;
    if (tid == 0) {
        s_row = atomicAdd(&sync[0], 1);
        s_touched = 0;
    }
    for (int i = tid; i < TB * TPITCH; i += TB * TW) acc[i] = 0;
    if (tid < TB) rflags[tid] = 0;
    __syncthreads();
    const int R = s_row, base = R * TB;
    const int r = base + lane;
    const bool active = r < n;
    auto phys = [&](int k) -> long long { return rev ? (long long)(n - 1 - k) : (long long)k; };
    const long long rp = phys(active ? r : base);
    const double *arow = a + rp * rs;

    for (int cc = w * TCW; cc < (w + 1) * TCW; ++cc) {
        double v = 0.0;
        if (active && cc < lane) v = arow[phys(base + cc) * cs];
        dg[cc * TB + lane] = v;
    }
    double dv = 1.0, rhs = 0.0;
    if (w == 0 && active) {
        if (!unit) dv = arow[rp * cs];
        rhs = x[rp * incx];  // written before the launch, replaced by this workgroup only
    }

    double f[N > 0 ? N : 1];
#pragma unroll
    for (int i = 0; i < (N > 0 ? N : 1); ++i) f[i] = 0.0;
    RowSink sink{acc + lane * TPITCH, &rflags[lane], &s_touched};
    Bypass bypass;

    for (int C = 0; C < R; ++C) {
        const int c0 = C * TB + w * TCW;
        double av[TCW];
#pragma unroll
        for (int u = 0; u < TCW; ++u)
            av[u] = active ? __builtin_nontemporal_load(arow + phys(c0 + u) * cs) : 0.0;
        const double xl = fetch_x(xq + c0 + lane, lane < TCW);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            double p[8], e[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) p[u] = two_prod(av[h * 8 + u], -lane_bcast(xl, h * 8 + u), e[u]);
            fpe_absorb_prod_adaptive<N, EE, 8>(f, p, e, sink, bypass);
        }
    }
    if (mode != 0) {
        fpe_flush_sink<N>(f, sink);
        if (w == 0 && active) sink.add(rhs);
    } else {
        // hand the expansions over as doubles (the diagonal wave adds them to its own expansion): no detour through
        // the integer accumulator unless more than DG_PUB levels hold something
        // (the early-exit variants keep products in levels 0-1 and their error terms in levels N-3.. : the levels in
        // use, not the first DG_PUB, are the ones that travel; a fifth non-empty level goes through the accumulator)
        int slot = 0;  // wave-uniform
#pragma unroll
        for (int k = 0; k < N; ++k) {
            if (__any(f[k] != 0.0)) {
                if (slot < DG_PUB) {
                    fpub[w][slot][lane] = f[k];
                    ++slot;
                } else if (f[k] != 0.0) {
                    sink.add(f[k]);
                }
            }
        }
        for (; slot < DG_PUB; ++slot) fpub[w][slot][lane] = 0.0;
    }
    __syncthreads();
    if (w != 0) return;

    // ---- diagonal block: one wave, one row at a time ----
    const int rows = min(TB, n - base);
    double xs = 0.0;
    if (mode != 0) {
        // reference-compatible rounding: every row through the integer accumulator and the full finish_wave
        for (int i = 0; i < rows; ++i) {
            __atomic_signal_fence(__ATOMIC_SEQ_CST);  // the row reads follow this wave's LDS adds in program order
            const long long v0 = acc[i * TPITCH + lane], v1 = lane < NL - 64 ? acc[i * TPITCH + 64 + lane] : 0;
            double v = finish_wave<true>(v0, v1, rflags[i]).rf;
            if (!unit) v = v / lane_bcast(dv, i);
            if (lane == i) xs = v;
            if (lane > i && active) {
                double e;
                const double p = two_prod(dg[i * TB + lane], -v, e);
                sink.add(p);
                if (e != 0.0 && expo_field(p) != 0x7ffu) sink.add(e);
            }
        }
    } else {
        // Exact rounding, register fast path.  A lone wave retires one fp64 instruction per 8 clocks, so the
        // chain is priced in INSTRUCTIONS per row; the integer route (2 x lds_add, LDS round trip, wave-wide
        // carry resolution and rounding, ~400 instructions) is replaced by a DG_M-level expansion per lane:
        //     value of row j  =  g_j[0] + ... + g_j[DG_M-1]  +  (row j of the LDS accumulator),   |LDS part| <= B_j
        // holds exactly at every step.  row_to_fpe moves the limbs gathered by the tile phase into g (once per
        // block, all rows at a time); every x_i then costs each later row one TwoProd and two TwoSum cascades.
        // At its turn a row folds g bottom-up into s with TwoSum: total = s + (sum of the errors) + LDS part, so
        // when sum|errors| + B is below half the distance from s to its nearer neighbour, s IS the correctly
        // rounded total (one double, no tie possible).  Otherwise -- near-ties, cancellation down to the noise,
        // huge / tiny / non-finite values -- the row is flushed to LDS and rounded by the integer path.
        double g[DG_M], B = 0.0;
        if (s_touched) {  // uniform; rare for the expansion variants on data they can hold
            row_to_fpe(acc + lane * TPITCH, rflags[lane], g, B, sink);
        } else {
#pragma unroll
            for (int k = 0; k < DG_M; ++k) g[k] = 0.0;
        }
        // leading levels first: the larger terms settle in the upper levels of g
#pragma unroll
        for (int k = 0; k < DG_PUB; ++k)
#pragma unroll
            for (int ww = 0; ww < TW; ++ww) {
                const double v = fpub[ww][k][lane];
                if (__any(v != 0.0)) fpe_push<0>(g, B, v, sink);
            }
        if (active) {
            if (__builtin_expect(expo_field(rhs) >= BIG_EXPO, 0)) {  // huge or non-finite right-hand side: integer side
                sink.add(rhs);
                B = __builtin_inf();
            } else {
                fpe_push<0>(g, B, rhs, sink);
            }
        }
        double acol = dg[lane];
        for (int i = 0; i < rows; ++i) {
            // ---- row i's turn (every lane evaluates its own row; lane i's answer is taken) ----
            double s = g[DG_M - 1], err = 0.0;
#pragma unroll
            for (int k = DG_M - 2; k >= 0; --k) {
                double e;
                s = two_sum(g[k], s, e);
                err += fabs(e);
            }
            // |total - s| <= err + B =: eb (rounded up into y).  s is the correctly rounded total when y is below half
            // the distance to s's neighbour on either side -- tested by letting the adder round: |s| + y and |s| - y
            // must both give |s| back.  (y == 0 passes for any s, zero and subnormals included; B == inf, NaN never do.)
            const double as = fabs(s);
            const double y = (err + B) * 1.0000001;
            const bool ok = (as + y == as) && (as - y == as);
            double v;
            if (__builtin_expect((int)((__ballot(ok) >> i) & 1ull), 1)) {
                v = lane_bcast(s, i) + 0.0;  // + 0.0: an exact zero total is +0 like the integer path's
            } else {
                if (lane == 0) atomicAdd(&sync[2], 1);  // statistics: rows that took the integer path
                if (lane == i) {
#pragma unroll
                    for (int k = 0; k < DG_M; ++k)
                        if (g[k] != 0.0) sink.add(g[k]);
                }
                __atomic_signal_fence(__ATOMIC_SEQ_CST);
                const long long v0 = acc[i * TPITCH + lane], v1 = lane < NL - 64 ? acc[i * TPITCH + 64 + lane] : 0;
                v = __longlong_as_double((long long)finish_wave<false>(v0, v1, rflags[i]).ex);
            }
            if (!unit) v = v / lane_bcast(dv, i);
            if (lane == i) xs = v;
            // ---- rows below take their product with x_i ----
            const double ac = acol;
            if (i + 1 < rows) acol = dg[(i + 1) * TB + lane];
            if (lane > i && active) {
                double e;
                const double p = two_prod(ac, -v, e);
                if (__builtin_expect(expo_field(p) >= BIG_EXPO, 0)) {  // too big for TwoSum, Inf or NaN: integer side
                    sink.add(p);
                    if (e != 0.0 && expo_field(p) != 0x7ffu) sink.add(e);
                    B = __builtin_inf();
                } else {
                    fpe_push<0>(g, B, p, sink);
                    fpe_push<DG_M - 2>(g, B, e, sink);
                }
            }
        }
    }
    if (active) {
        post_x(xq + r, xs);
        x[rp * incx] = xs;
    }
}

// ---------------------------------------------------------------------------------------------
// fpe == 1: the plain fp64 solve (DTRSV.lnn.cl / DTRSV.unn.cl), same blocking.  Not exact, but deterministic here:
// the partial sums of the 4 waves are combined in a fixed order.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(TB *TW) k_dtrsv(int n, const double *__restrict__ a, long long rs, long long cs,
                                                  double *x, long long incx, int rev, int unit, int *sync, double *xq)
{
    __shared__ double dg[TB * TB];
    __shared__ double part[TW][TB];
    __shared__ int s_row;
    const int tid = (int)threadIdx.x, lane = tid & 63, w = tid >> 6;
    if (tid == 0) s_row = atomicAdd(&sync[0], 1);
    __syncthreads();
    const int R = s_row, base = R * TB;
    const int r = base + lane;
    const bool active = r < n;
    auto phys = [&](int k) -> long long { return rev ? (long long)(n - 1 - k) : (long long)k; };
    const long long rp = phys(active ? r : base);
    const double *arow = a + rp * rs;
    for (int cc = w * TCW; cc < (w + 1) * TCW; ++cc) {
        double v = 0.0;
        if (active && cc < lane) v = arow[phys(base + cc) * cs];
        dg[cc * TB + lane] = v;
    }
    double s = 0.0;
    for (int C = 0; C < R; ++C) {
        const int c0 = C * TB + w * TCW;
        double av[TCW];
#pragma unroll
        for (int u = 0; u < TCW; ++u)
            av[u] = active ? __builtin_nontemporal_load(arow + phys(c0 + u) * cs) : 0.0;
        const double xl = fetch_x(xq + c0 + lane, lane < TCW);
#pragma unroll
        for (int u = 0; u < TCW; ++u) s -= av[u] * lane_bcast(xl, u);
    }
    part[w][lane] = s;
    __syncthreads();
    if (w != 0) return;
    double t = active ? x[rp * incx] : 0.0;
    t = ((t + part[0][lane]) + part[1][lane]) + (part[2][lane] + part[3][lane]);
    double dv = 1.0;
    if (active && !unit) dv = arow[rp * cs];
    const int rows = min(TB, n - base);
    for (int i = 0; i < rows; ++i) {
        if (lane == i && !unit) t = t / dv;
        const double v = __shfl(t, i);
        if (lane > i && active) t -= dg[i * TB + lane] * v;
    }
    if (active) {
        post_x(xq + r, t);
        x[rp * incx] = t;
    }
}

template <int N, bool EE>
hipError_t trsv_variant(int n, const double *a, long long rs, long long cs, double *x, long long incx, int rev, int unit,
                        int mode, int *sync, double *xq, hipStream_t st)
{
    hipLaunchKernelGGL((k_trsv<N, EE>), dim3((n + TB - 1) / TB), dim3(TB * TW), 0, st, n, a, rs, cs, x, incx, rev, unit,
                       mode, sync, xq);
    return hipGetLastError();
}

}  // namespace

// variant selection: ExTRSV.cpp:70-123.  Returns hipErrorNotSupported for the iterative-refinement values of fpe
// (>= 9), whose kernel files the reference names but does not ship.
hipError_t extrsv_dispatch(Ctx &c, char uplo, char transa, char diag, int n, const double *a, int lda, double *x,
                           int incx, int fpe, int early_exit, int round_mode, hipStream_t st)
{
    if (n <= 0) return hipSuccess;
    if (fpe < 0 || fpe >= 9) return hipErrorNotSupported;
    const bool lower = (uplo == 'L' || uplo == 'l'), trans = (transa == 'T' || transa == 't');
    const int unit = (diag == 'U' || diag == 'u') ? 1 : 0;
    const int rev = (lower != trans) ? 0 : 1;  // A**T of a lower matrix is upper: backward substitution
    const long long rs = trans ? (long long)lda : 1ll, cs = trans ? 1ll : (long long)lda;
    // workspace: 16 ints (ticket, statistics) then the mailbox of n doubles
    hipError_t e;
    int *sync = (int *)workspace(c, 64 + (size_t)n * sizeof(double), st, &e);
    if (!sync) return e;
    double *xq = (double *)((char *)sync + 64);
    e = hipMemsetAsync(sync, 0, 64, st);
    if (e != hipSuccess) return e;
    if ((e = hipMemsetAsync(xq, 0xff, (size_t)n * sizeof(double), st)) != hipSuccess) return e;
#define TV_ARGS n, a, rs, cs, x, (long long)incx, rev, unit, round_mode, sync, xq, st
    if (fpe == 0) return trsv_variant<0, false>(TV_ARGS);
    if (fpe == 1) {
        hipLaunchKernelGGL(k_dtrsv, dim3((n + TB - 1) / TB), dim3(TB * TW), 0, st, n, a, rs, cs, x, (long long)incx, rev,
                           unit, sync, xq);
        return hipGetLastError();
    }
    if (early_exit) {
        if (fpe <= 4) return trsv_variant<4, true>(TV_ARGS);
        if (fpe <= 6) return trsv_variant<6, true>(TV_ARGS);
        return trsv_variant<8, true>(TV_ARGS);
    }
    switch (fpe) {
    case 2: return trsv_variant<2, false>(TV_ARGS);
    case 3: return trsv_variant<3, false>(TV_ARGS);
    case 4: return trsv_variant<4, false>(TV_ARGS);
    case 5: return trsv_variant<5, false>(TV_ARGS);
    case 6: return trsv_variant<6, false>(TV_ARGS);
    case 7: return trsv_variant<7, false>(TV_ARGS);
    default: return trsv_variant<8, false>(TV_ARGS);
    }
#undef TV_ARGS
}

}  // namespace exb
