// blas1.hip -- ExSUM / ExDOT for gfx950: streaming FPE front-end + per-wavefront LDS superaccumulators.
//
// Replaces the reference's OpenCL kernels ExSUM/ExSUMComplete (src/gpu/blas/blas1/ExSUM.Superacc.cl:211-356,
// ExSUM.FPE.cl:230-388, ExSUM.FPE.EX.{4,6,8}.cl) and ExDOT/ExDOTComplete (ExDOT.Superacc.cl:217-359,
// ExDOT.FPE.cl:201-345) -- behaviour only; the structure is ours:
//   * every lane streams 16-byte (double2) coalesced, non-temporal loads, U of them in flight per tile, and the
//     next tile is already on its way into a second register set while the current one is absorbed;
//   * a register-resident floating-point expansion of NFPE doubles absorbs the elements with Knuth
//     TwoSum; the early-exit test is one wave-uniform branch per level per *tile* (not per element); a tile
//     whose residues outlive all levels switches the wave to the direct path for the next 63 tiles
//     (fpe_absorb_adaptive), a tile with |x| >= 2^1000 / Inf / NaN takes the range guard (fpe_guard);
//   * what survives the expansion is split by integer shifts and added with ds_add_u64 to a
//     per-wavefront superaccumulator in LDS (COPIES columns per wave, limb-major, see superacc.hip.h);
//   * block epilogue: merge the columns, add the non-zero limbs to one of NGROUPS global accumulators
//     with int64 atomics (exact, order-free); the finalize kernel sums the groups, carry-propagates
//     ONCE, re-cuts into the reference's canonical limbs and rounds.  No inter-workgroup reads inside a
//     launch (the reference's ExSUMComplete races on that).
#include "superacc.hip.h"
#include "fpe.hip.h"
#include "exblas_internal.h"
#include <hip/hip_ext.h>

namespace exb {

constexpr int BLOCK = 256;
constexpr int WAVES = BLOCK / 64;

// blas1 sinks everything into the wave's LDS accumulator column
template <int N, bool EE, int COPIES, int CNT, int ZM = 0>
__device__ __forceinline__ void fpe_absorb(double (&a)[N > 0 ? N : 1], double (&x)[CNT], int from,
                                           long long *col, unsigned &flags)
{
    LdsSink<COPIES> sink{col, flags};
    fpe_absorb_sink<N, EE, CNT, LdsSink<COPIES>, ZM>(a, x, from, sink);
}

template <int N, int COPIES>
__device__ __forceinline__ void fpe_flush(double (&a)[N > 0 ? N : 1], long long *col, unsigned &flags)
{
    LdsSink<COPIES> sink{col, flags};
    fpe_flush_sink<N>(a, sink);
}

// block epilogue: columns -> one limb vector -> global group accumulator
template <int COPIES>
__device__ __forceinline__ void block_epilogue(long long *s_acc, unsigned flags, long long *gacc,
                                               unsigned *gflags, int ngroups)
{
    __shared__ unsigned s_flags;
    if (threadIdx.x == 0) s_flags = 0;
    __syncthreads();  // also orders every wave's LDS atomics before the merge
    if (flags) atomicOr(&s_flags, flags);
    long long *g = gacc + (size_t)(blockIdx.x % ngroups) * NL;
    for (int l = threadIdx.x; l < NL; l += BLOCK) {
        long long sum = 0;
#pragma unroll
        for (int w = 0; w < WAVES; ++w)
#pragma unroll
            for (int c = 0; c < COPIES; ++c) sum += s_acc[(w * NL + l) * COPIES + c];
        if (sum != 0) atomicAdd((unsigned long long *)&g[l], (unsigned long long)sum);
    }
    __syncthreads();
    if (threadIdx.x == 0 && s_flags) atomicOr(gflags, s_flags);
}

// ---------------------------------------------------------------------------------------------
// ExSUM, contiguous input
// ---------------------------------------------------------------------------------------------
template <int N, bool EE, int COPIES, int U, bool NT, bool PF, int ZM = 0>
__global__ void __launch_bounds__(BLOCK) k_exsum(const double *__restrict__ a, long long n,
                                                 long long *__restrict__ gacc,
                                                 unsigned *__restrict__ gflags, int ngroups, int chunked)
{
    __shared__ long long s_acc[WAVES * NL * COPIES];
    auto zero_lds = [&]() {   // (after the first tile's loads are in flight, see k_exdot)
        for (int i = threadIdx.x; i < WAVES * NL * COPIES; i += BLOCK) s_acc[i] = 0;
        __syncthreads();
    };
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    long long *col = s_acc + wave * NL * COPIES + (lane & (COPIES - 1));
    unsigned flags = 0;
    double fpe[N > 0 ? N : 1];
#pragma unroll
    for (int i = 0; i < (N > 0 ? N : 1); ++i) fpe[i] = 0.0;

    // 16-byte alignment: at most one scalar head element
    const long long head = (((uintptr_t)a & 8u) && n > 0) ? 1 : 0;
    const d2_t *v = (const d2_t *)(a + head);
    const long long nv = (n - head) >> 1;
    constexpr long long TILE = (long long)BLOCK * U;
    const long long ntiles = nv / TILE;
    LdsSink<COPIES> sinkm{col, flags};
    Bypass bypass;

    if constexpr (!PF) {
        zero_lds();
        for (long long t = blockIdx.x; t < ntiles; t += gridDim.x) {
            const d2_t *p = v + t * TILE + threadIdx.x;
            d2_t r[U];
#pragma unroll
            for (int u = 0; u < U; ++u) r[u] = ld2<NT>(p + u * BLOCK);
            double x[2 * U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                x[2 * u] = r[u].x;
                x[2 * u + 1] = r[u].y;
            }
            fpe_absorb_adaptive<N, EE, 2 * U, LdsSink<COPIES>, ZM>(fpe, x, sinkm, bypass);
        }
    } else {
        // register double-buffering: the next tile's loads are in flight while this one is absorbed.
        // Tile -> workgroup map: strided (tile t to workgroup t mod grid: the grid sweeps one compact window
        // of addresses) or chunked (each workgroup streams its own contiguous range).
        long long t = blockIdx.x, tstride = gridDim.x, tend = ntiles;
        if (chunked) {
            const long long per = (ntiles + gridDim.x - 1) / gridDim.x;
            t = blockIdx.x * per;
            tstride = 1;
            tend = min(ntiles, t + per);
        }
        if (t < tend) {
            // two explicit register sets, filled alternately with unconditional loads (past the end: the last tile
            // again) -- see k_exdot: no per-trip copies between the sets, both in flight
            d2_t r0[U], r1[U];
            auto fill = [&](long long tile, d2_t (&r)[U]) {
                const d2_t *p = v + (tile < tend ? tile : tend - 1) * TILE + threadIdx.x;
#pragma unroll
                for (int u = 0; u < U; ++u) r[u] = ld2<NT>(p + u * BLOCK);
            };
            auto absorb = [&](d2_t (&r)[U]) {
                double x[2 * U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    x[2 * u] = r[u].x;
                    x[2 * u + 1] = r[u].y;
                }
                fpe_absorb_adaptive<N, EE, 2 * U, LdsSink<COPIES>, ZM>(fpe, x, sinkm, bypass);
            };
            fill(t, r0);
            zero_lds();
            for (;;) {
                fill(t + tstride, r1);
                absorb(r0);
                t += tstride;
                if (t >= tend) break;
                fill(t + tstride, r0);
                absorb(r1);
                t += tstride;
                if (t >= tend) break;
            }
        } else {
            zero_lds();
        }
    }
    // remainder vectors, grid-strided one double2 at a time
    for (long long i = ntiles * TILE + (long long)blockIdx.x * BLOCK + threadIdx.x; i < nv;
         i += (long long)gridDim.x * BLOCK) {
        d2_t r = v[i];
        double x[2] = {r.x, r.y};
        fpe_absorb<N, false, COPIES, 2>(fpe, x, 0, col, flags);
    }
    // scalar head / tail straight into the superaccumulator
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        if (head) lds_add<COPIES>(col, a[0], flags);
        if ((n - head) & 1) lds_add<COPIES>(col, a[n - 1], flags);
    }
    fpe_flush<N, COPIES>(fpe, col, flags);
    block_epilogue<COPIES>(s_acc, flags, gacc, gflags, ngroups);
}

// ExSUM, strided input a[i*inca] (ExSUM.Superacc.cl:248-264 takes the same slow path)
template <int N, bool EE, int COPIES>
__global__ void __launch_bounds__(BLOCK) k_exsum_strided(const double *__restrict__ a, long long n,
                                                         long long inca, long long *__restrict__ gacc,
                                                         unsigned *__restrict__ gflags, int ngroups)
{
    __shared__ long long s_acc[WAVES * NL * COPIES];
    for (int i = threadIdx.x; i < WAVES * NL * COPIES; i += BLOCK) s_acc[i] = 0;
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    long long *col = s_acc + wave * NL * COPIES + (lane & (COPIES - 1));
    unsigned flags = 0;
    double fpe[N > 0 ? N : 1];
#pragma unroll
    for (int i = 0; i < (N > 0 ? N : 1); ++i) fpe[i] = 0.0;
    // four independent loads in flight per lane (a strided element is its own cache line: latency, not bandwidth)
    const long long T = (long long)gridDim.x * BLOCK;
    for (long long i = (long long)blockIdx.x * BLOCK + threadIdx.x; i < n; i += 4 * T) {
        double x[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) x[u] = i + u * T < n ? a[(i + u * T) * inca] : 0.0;
        fpe_absorb<N, false, COPIES, 4>(fpe, x, 0, col, flags);
    }
    fpe_flush<N, COPIES>(fpe, col, flags);
    block_epilogue<COPIES>(s_acc, flags, gacc, gflags, ngroups);
}

// ---------------------------------------------------------------------------------------------
// ExDOT: TwoProductFMA front-end (ExDOT.Superacc.cl:25-29, :244-253); the rounding error of the
// product enters the expansion at slot max(N-3,0) like ExDOT.FPE.cl:254
// ---------------------------------------------------------------------------------------------
template <int N, bool EE, int COPIES, int U, bool NT, bool PF, int WPS = 1, bool HALVES = false, int ZM = 0>
__global__ void __launch_bounds__(BLOCK, WPS) k_exdot(const double *__restrict__ a, const double *__restrict__ b,
                                                 long long n, long long *__restrict__ gacc,
                                                 unsigned *__restrict__ gflags, int ngroups)
{
    __shared__ long long s_acc[WAVES * NL * COPIES];
    // (zeroed below, AFTER the first tile's loads are in flight: a microsecond of every workgroup's life, which counts
    // for the short shards of a multi-GPU job -- 2^25 elements are 16 us of HBM time per stream)
    auto zero_lds = [&]() {
        for (int i = threadIdx.x; i < WAVES * NL * COPIES; i += BLOCK) s_acc[i] = 0;
        __syncthreads();
    };
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    long long *col = s_acc + wave * NL * COPIES + (lane & (COPIES - 1));
    unsigned flags = 0;
    LdsSink<COPIES> sink{col, flags};
    long long *const lo_acc = low_acc_of(gflags), *const hi_acc = high_acc_of(gflags);   // exact homes of the products below 2^-968 / beyond the double range (prod_range_divert)
    double fpe[N > 0 ? N : 1];
#pragma unroll
    for (int i = 0; i < (N > 0 ? N : 1); ++i) fpe[i] = 0.0;

    // vector path only when both streams are 16-byte aligned (host guarantees or falls to strided)
    const d2_t *va = (const d2_t *)a, *vb = (const d2_t *)b;
    const long long nv = n >> 1;
    Bypass bypass;
    constexpr long long TILE = (long long)BLOCK * U;
    const long long ntiles = nv / TILE;
    if constexpr (!PF) {
        zero_lds();
        for (long long t = blockIdx.x; t < ntiles; t += gridDim.x) {
            const long long base = t * TILE + threadIdx.x;
            d2_t ra[U], rb[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                ra[u] = ld2<NT>(va + base + u * BLOCK);
                rb[u] = ld2<NT>(vb + base + u * BLOCK);
            }
            double x[2 * U], e[2 * U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                x[2 * u] = two_prod(ra[u].x, rb[u].x, e[2 * u]);
                x[2 * u + 1] = two_prod(ra[u].y, rb[u].y, e[2 * u + 1]);
            }
            if (!prod_range_divert<2 * U>(fpe[0], x, e, sink, lo_acc, hi_acc, [&](int j) { return j & 1 ? ra[j >> 1].y : ra[j >> 1].x; },
                                              [&](int j) { return j & 1 ? rb[j >> 1].y : rb[j >> 1].x; }))
                fpe_absorb_prod_adaptive<N, EE, 2 * U, LdsSink<COPIES>, 0, false>(fpe, x, e, sink, bypass);
        }
    } else {
        long long t = blockIdx.x;
        d2_t ra[U], rb[U];
        if (t < ntiles) {
            const long long base = t * TILE + threadIdx.x;
#pragma unroll
            for (int u = 0; u < U; ++u) {
                ra[u] = ld2<NT>(va + base + u * BLOCK);
                rb[u] = ld2<NT>(vb + base + u * BLOCK);
            }
        }
        zero_lds();
        if constexpr (HALVES) {
            while (t < ntiles) {
                const long long tn = t + gridDim.x;
                // two half-tiles: the registers of a half are re-loaded (next tile) as soon as its products are
                // formed, so only U products + U errors are live at a time (fewer VGPRs -> more waves per SIMD)
                constexpr int H = U / 2;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    double x[2 * H], e[2 * H];
#pragma unroll
                    for (int u = 0; u < H; ++u) {
                        x[2 * u] = two_prod(ra[h * H + u].x, rb[h * H + u].x, e[2 * u]);
                        x[2 * u + 1] = two_prod(ra[h * H + u].y, rb[h * H + u].y, e[2 * u + 1]);
                    }
                    const bool diverted = prod_range_divert<2 * H>(fpe[0], x, e, sink, lo_acc, hi_acc, [&](int j) { return j & 1 ? ra[h * H + (j >> 1)].y : ra[h * H + (j >> 1)].x; },
                        [&](int j) { return j & 1 ? rb[h * H + (j >> 1)].y : rb[h * H + (j >> 1)].x; });
                    if (tn < ntiles) {
                        const long long base = tn * TILE + threadIdx.x;
#pragma unroll
                        for (int u = 0; u < H; ++u) {
                            ra[h * H + u] = ld2<NT>(va + base + (h * H + u) * BLOCK);
                            rb[h * H + u] = ld2<NT>(vb + base + (h * H + u) * BLOCK);
                        }
                    }
                    if (!diverted) fpe_absorb_prod_adaptive<N, EE, 2 * H, LdsSink<COPIES>, 0, false>(fpe, x, e, sink, bypass);
                }
                t = tn;
            }
        } else if (t < ntiles) {
            // Two explicit register sets, filled alternately; the loads are unconditional (past the end they
            // fetch the last tile again), so the compiler has no reason to copy one set into the other each
            // trip (16 v_mov_b64 per tile with a conditional reload) and both sets stay in flight.
            d2_t rc[U], rd[U];
            auto fill = [&](long long tile, d2_t (&pa)[U], d2_t (&pb)[U]) {
                const long long base = (tile < ntiles ? tile : ntiles - 1) * TILE + threadIdx.x;
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    pa[u] = ld2<NT>(va + base + u * BLOCK);
                    pb[u] = ld2<NT>(vb + base + u * BLOCK);
                }
            };
            auto absorb = [&](d2_t (&pa)[U], d2_t (&pb)[U]) {
                double x[2 * U], e[2 * U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    x[2 * u] = two_prod(pa[u].x, pb[u].x, e[2 * u]);
                    x[2 * u + 1] = two_prod(pa[u].y, pb[u].y, e[2 * u + 1]);
                }
                if (!prod_range_divert<2 * U>(fpe[0], x, e, sink, lo_acc, hi_acc, [&](int j) { return j & 1 ? pa[j >> 1].y : pa[j >> 1].x; },
                                                  [&](int j) { return j & 1 ? pb[j >> 1].y : pb[j >> 1].x; }))
                    fpe_absorb_prod_adaptive<N, EE, 2 * U, LdsSink<COPIES>, ZM, false>(fpe, x, e, sink, bypass);
            };
            for (;;) {
                fill(t + gridDim.x, rc, rd);
                absorb(ra, rb);
                t += gridDim.x;
                if (t >= ntiles) break;
                fill(t + gridDim.x, ra, rb);
                absorb(rc, rd);
                t += gridDim.x;
                if (t >= ntiles) break;
            }
        }
    }
    for (long long i = ntiles * TILE + (long long)blockIdx.x * BLOCK + threadIdx.x; i < nv;
         i += (long long)gridDim.x * BLOCK) {
        d2_t ra = va[i], rb = vb[i];
        double x[2], e[2];
        x[0] = two_prod(ra.x, rb.x, e[0]);
        x[1] = two_prod(ra.y, rb.y, e[1]);
        if (!prod_range_divert<2>(fpe[0], x, e, sink, lo_acc, hi_acc, [&](int j) { return j ? ra.y : ra.x; }, [&](int j) { return j ? rb.y : rb.x; }))
            fpe_absorb_prod<N, false, 2, LdsSink<COPIES>, 0, false>(fpe, x, e, sink);
    }
    if (blockIdx.x == 0 && threadIdx.x == 0 && (n & 1)) {
        double x1[1], e1[1];
        x1[0] = two_prod(a[n - 1], b[n - 1], e1[0]);
        // (only this lane is active: the helper's votes are votes of one)
        if (!prod_range_divert<1>(fpe[0], x1, e1, sink, lo_acc, hi_acc, [&](int) { return a[n - 1]; }, [&](int) { return b[n - 1]; }))
            sink_product(sink, x1[0], e1[0]);
    }
    fpe_flush<N, COPIES>(fpe, col, flags);
    block_epilogue<COPIES>(s_acc, flags, gacc, gflags, ngroups);
}

template <int N, bool EE, int COPIES>
__global__ void __launch_bounds__(BLOCK) k_exdot_strided(const double *__restrict__ a, long long inca,
                                                         const double *__restrict__ b, long long incb,
                                                         long long n, long long *__restrict__ gacc,
                                                         unsigned *__restrict__ gflags, int ngroups)
{
    __shared__ long long s_acc[WAVES * NL * COPIES];
    for (int i = threadIdx.x; i < WAVES * NL * COPIES; i += BLOCK) s_acc[i] = 0;
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    long long *col = s_acc + wave * NL * COPIES + (lane & (COPIES - 1));
    unsigned flags = 0;
    LdsSink<COPIES> sink{col, flags};
    long long *const lo_acc = low_acc_of(gflags), *const hi_acc = high_acc_of(gflags);
    double fpe[N > 0 ? N : 1];
#pragma unroll
    for (int i = 0; i < (N > 0 ? N : 1); ++i) fpe[i] = 0.0;
    const long long T = (long long)gridDim.x * BLOCK;
    for (long long i = (long long)blockIdx.x * BLOCK + threadIdx.x; i < n; i += 4 * T) {
        double va[4], vb[4], x[4], e[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const bool in = i + u * T < n;
            va[u] = in ? a[(i + u * T) * inca] : 0.0;
            vb[u] = in ? b[(i + u * T) * incb] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) x[u] = two_prod(va[u], vb[u], e[u]);
        if (!prod_range_divert<4>(fpe[0], x, e, sink, lo_acc, hi_acc, [&](int j) { return va[j]; }, [&](int j) { return vb[j]; }))
            fpe_absorb_prod<N, false, 4, LdsSink<COPIES>, 0, false>(fpe, x, e, sink);
    }
    fpe_flush<N, COPIES>(fpe, col, flags);
    block_epilogue<COPIES>(s_acc, flags, gacc, gflags, ngroups);
}

// ---------------------------------------------------------------------------------------------
// finalize: sum `nsets` limb vectors, ONE carry propagation, canonical limbs, rounding.
// zero_sets: the input is the context's group accumulators -> leave them zeroed for the next call.
// ---------------------------------------------------------------------------------------------
// ext_out (with gflags): EXPORT the context's low and high accumulators (ExDOT products beyond the accumulator's range,
// fpe.hip.h: prod_range_divert) as two normalised digit sets [low | high] instead of folding them (the multi-rank path
// all-reduces them beside the main digit set); ext_in: fold THESE digit sets (the all-reduced ones).
__global__ void __launch_bounds__(64) k_finalize(long long *sets, int nsets, int set_stride, unsigned *gflags,
                                                 unsigned flags_or, int zero_sets, long long *out,
                                                 const long long *ext_in, long long *ext_out)
{
    // one wavefront: lane l owns limb l and (l < 4) limb 64+l
    const int lane = threadIdx.x;
    unsigned flags = flags_or;
    if (gflags) flags |= *gflags;
    if (set_stride >= SET_WORDS) {  // record-style sets carry their own flag indicators
        for (int g = 0; g < nsets; ++g)
        {
            for (int k = 0; k < 3; ++k)
                if (sets[(size_t)g * set_stride + NL + k] != 0) flags |= (1u << k);
            const long long pw = sets[(size_t)g * set_stride + NL + 3];  // product flags: low 16 bits / the rest
            if (pw & 0xffff) flags |= FLAG_PUNDER;
            if (pw >> 16) flags |= FLAG_POVER;
        }
    }
    // all loads of a batch are issued before the first use (the words were last touched by other CUs' atomics,
    // so each load is a full memory round trip: 32 dependent ones cost ~20 us)
    // The low 32 bits and the (signed) high parts of the sets are summed separately and the high parts enter one
    // limb up, so this sum cannot overflow however many adds the sets hold (each set alone stays below 2^63 as long
    // as it received fewer than 2^31 adds: with 32 group accumulators that is 2^36 values per reduction).
    long long lo0 = 0, hi0 = 0, lo1 = 0, hi1 = 0;
    for (int g0 = 0; g0 < nsets; g0 += 16) {
        long long t0[16], t1[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const bool ok = g0 + k < nsets;
            const long long *p = sets + (size_t)(ok ? g0 + k : 0) * set_stride;
            t0[k] = ok ? __builtin_nontemporal_load(p + lane) : 0;
            t1[k] = (ok && lane < NL - 64) ? __builtin_nontemporal_load(p + 64 + lane) : 0;
        }
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            lo0 += t0[k] & 0xffffffffll;
            hi0 += t0[k] >> 32;
            if (lane == NL - 65) {  // the top limb is never split
                lo1 += t1[k];
            } else {
                lo1 += t1[k] & 0xffffffffll;
                hi1 += t1[k] >> 32;
            }
        }
    }
    long long in0 = __shfl_up(hi0, 1), in1 = __shfl_up(hi1, 1);
    const long long h63 = __shfl(hi0, 63);
    if (lane == 0) { in0 = 0; in1 = h63; }
    long long v0 = lo0 + in0, v1 = (lane < NL - 64) ? lo1 + in1 : 0;
    if (zero_sets) {
        for (int i = lane; i < nsets * set_stride; i += 64) sets[i] = 0;
        if (lane == 0 && gflags) *gflags = 0;
    }
    // ExDOT with products below 2^-968 (FLAG_PUNDER) or beyond the double range (FLAG_POVER): fold the LOW / HIGH
    // accumulator in.  The high digits are added to the main digits exactly, EXT_SHIFT_DIGITS digits up (a sum that does
    // not fit the main geometry is beyond the double range whatever follows: +-Inf); the low accumulator's digits at or
    // above 2^-1074 are added exactly, what lies below becomes the half / sticky bits of the rounding.  Rare, so one lane
    // does it with the scalar routines (carry passes over 68 / 106 digits + one rounding).
    __shared__ long long s_v[NL], s_lo[NL], s_hi[NL], s_c[NL + EXT_SHIFT_DIGITS];
    __shared__ unsigned long long s_ex;
    __shared__ int s_over;
    bool low_folded = false;
    const bool have_low = (flags & FLAG_PUNDER) && (ext_in || gflags);   // wave-uniform
    const bool have_high = (flags & FLAG_POVER) && (ext_in || gflags);
    if (ext_out) {   // export mode: all-zero sets for the accumulators nothing was added to
        if (!have_low) {
            ext_out[lane] = 0;
            if (lane < SET_WORDS - 64) ext_out[64 + lane] = 0;
        }
        if (!have_high) {
            ext_out[SET_WORDS + lane] = 0;
            if (lane < SET_WORDS - 64) ext_out[SET_WORDS + 64 + lane] = 0;
        }
    }
    if (have_low || have_high) {
        const long long *lsrc = ext_in ? ext_in : low_acc_of(gflags);
        const long long *hsrc = ext_in ? ext_in + SET_WORDS : high_acc_of(gflags);
        s_lo[lane] = have_low ? lsrc[lane] : 0;
        s_hi[lane] = have_high ? hsrc[lane] : 0;
        s_v[lane] = v0;
        if (lane < NL - 64) {
            s_lo[64 + lane] = have_low ? lsrc[64 + lane] : 0;
            s_hi[64 + lane] = have_high ? hsrc[64 + lane] : 0;
            s_v[64 + lane] = v1;
        }
        if (!ext_in && zero_sets) {
            long long *lo = low_acc_of(gflags), *hi = high_acc_of(gflags);
            if (have_low) {
                lo[lane] = 0;
                if (lane < NL - 64) lo[64 + lane] = 0;
            }
            if (have_high) {
                hi[lane] = 0;
                if (lane < NL - 64) hi[64 + lane] = 0;
            }
        }
        __syncthreads();
        if (lane == 0) {
            s_over = 0;
            if (have_low) normalize_digits(s_lo);
            if (have_high) normalize_digits(s_hi);
            if (!ext_out) {
                if (have_high) s_over = fold_high_digits(s_v, s_hi, s_c);
                if (have_low && !s_over) {
                    for (int j = LOW_SHIFT_DIGITS; j < NL; ++j) s_v[j - LOW_SHIFT_DIGITS] += s_lo[j];
                    normalize_digits(s_v);
                    const bool half = (s_lo[LOW_SHIFT_DIGITS - 1] >> 31) & 1ll;
                    bool sticky = (s_lo[LOW_SHIFT_DIGITS - 1] & 0x7fffffffll) != 0;
                    for (int j = 0; j < LOW_SHIFT_DIGITS - 1; ++j) sticky = sticky || s_lo[j] != 0;
                    s_ex = round_exact_bits_frac(s_v, half, sticky);
                }
            }
        }
        __syncthreads();
        if (ext_out) {   // digits < 2^32 (the top one signed and small): sums over 2^31 ranks cannot overflow
            if (have_low) {
                ext_out[lane] = s_lo[lane];
                if (lane < SET_WORDS - 64) ext_out[64 + lane] = lane < NL - 64 ? s_lo[64 + lane] : 0;
            }
            if (have_high) {
                ext_out[SET_WORDS + lane] = s_hi[lane];
                if (lane < SET_WORDS - 64) ext_out[SET_WORDS + 64 + lane] = lane < NL - 64 ? s_hi[64 + lane] : 0;
            }
        } else {
            const int over = s_over;   // wave-uniform
            if (over) {
                // the exact sum is beyond the double range: it rounds to +-Inf, reported like an infinity in the input
                flags |= over > 0 ? FLAG_PINF : FLAG_NINF;
            } else {
                v0 = s_v[lane];
                v1 = lane < NL - 64 ? s_v[64 + lane] : 0;
                low_folded = have_low;
            }
            if (have_low) flags |= FLAG_PLOW_EXACT;
            if (have_high) flags |= FLAG_PHIGH_EXACT;
        }
    }
    // every input word has been read (into registers) before the first output word is written: out may alias sets
    WaveFinish r = finish_wave(v0, v1, flags);
    if (low_folded && !(flags & FLAG_NONFINITE)) r.ex = s_ex;
    write_record_wave(r, flags, out);
}

// ---------------------------------------------------------------------------------------------
// Segmented ExSUM: out[s] = Round(sum values[offsets[s] .. offsets[s+1])), one wave per segment, four
// segments per workgroup.  This is the batched form of the call pattern of the reference's SpMV example, which
// runs one exsum per matrix row on a short vector of products (src/cpu/examples/spmv (Parboil)/
// StrongReproducibility/main.cpp:85): thousands of tiny reductions cost one launch instead of one each.
// ---------------------------------------------------------------------------------------------
template <int N, bool EE>
__global__ void __launch_bounds__(BLOCK) k_exsum_segmented(const double *__restrict__ values,
                                                           const long long *__restrict__ offsets, long long nseg,
                                                           int round_mode, double *__restrict__ out)
{
    __shared__ long long acc[WAVES][NL];
    __shared__ unsigned fl[WAVES];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const long long seg = (long long)blockIdx.x * WAVES + w;
    const bool valid = seg < nseg;
    for (int t = lane; t < NL; t += 64) acc[w][t] = 0;
    if (lane == 0) fl[w] = 0;
    __syncthreads();
    if (valid) {
        unsigned flags = 0;
        LdsSink<1> sink{acc[w], flags};
        double fpe[N > 0 ? N : 1];
#pragma unroll
        for (int i = 0; i < (N > 0 ? N : 1); ++i) fpe[i] = 0.0;
        const long long b = offsets[seg], e = offsets[seg + 1];
        // all lanes run the same number of iterations (the early-exit vote is wave-wide); lanes past the end add 0
        for (long long i0 = b; i0 < e; i0 += 64) {
            const long long i = i0 + lane;
            double x[1] = {i < e ? values[i] : 0.0};
            fpe_absorb_sink<N, EE, 1>(fpe, x, 0, sink);
        }
        fpe_flush_sink<N>(fpe, sink);
        if (flags) atomicOr(&fl[w], flags);
    }
    __syncthreads();
    const WaveFinish r = finish_wave(acc[w][lane], lane < NL - 64 ? acc[w][64 + lane] : 0, fl[w]);
    if (valid && lane == 0) out[seg] = round_mode ? r.rf : __longlong_as_double((long long)r.ex);
}

// ---------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------

// streaming-kernel launch: plain, or -- when the context holds events for it -- with the events attached to the
// dispatch packet itself (exblas_internal.h: launch_stop_event)
#define EXB_LAUNCH_STREAMING(c, kernel, grid, st, ...)                                                                \
    do {                                                                                                               \
        if ((c).launch_stop_event || (c).launch_start_event) {                                                         \
            hipEvent_t ev0_ = (c).launch_start_event, ev1_ = (c).launch_stop_event;                                    \
            (c).launch_start_event = (c).launch_stop_event = nullptr;                                                  \
            hipExtLaunchKernelGGL(kernel, dim3(grid), dim3(BLOCK), 0, st, ev0_, ev1_, 0, __VA_ARGS__);                 \
        } else {                                                                                                       \
            hipLaunchKernelGGL(kernel, dim3(grid), dim3(BLOCK), 0, st, __VA_ARGS__);                                   \
        }                                                                                                              \
    } while (0)

static inline int grid_for(const Ctx &c, long long work_items, long long per_block, int blocks_per_cu)
{
    long long want = (work_items + per_block - 1) / per_block;
    long long cap = (long long)c.num_cu * blocks_per_cu;
    if (want < 1) want = 1;
    if (want >= cap) return (int)max(1ll, cap + c.grid_adj);
    return (int)want;
}

template <int N, bool EE, int COPIES, int U, bool NT, bool PF, int ZM = 0>
static void run_exsum(Ctx &c, const double *a, long long n, hipStream_t st)
{
    // resident blocks per CU: the HBM-bound kernels want few fat blocks; the variants that run the full N-level cascade
    // on every element (no early exit, N >= 5: 30-48 dependent fp64 adds per element) are VALU-latency-bound and want
    // more waves per SIMD to hide it
    const int bpc = N == 0 ? c.bpc_sa : ((!SKIP_ZERO_LEVELS<EE> && N >= 5) ? c.bpc_heavy : c.bpc_sum);
    int grid = grid_for(c, n, (long long)BLOCK * 2 * U, bpc);
    // an odd number of workgroups (one resident slot left idle): the tiles a workgroup has in flight are grid x 16 KiB
    // apart, and with an even grid they compete for the same HBM channels -- 865 against 855 Gelem/s at n = 2^28
    if (grid == c.num_cu * bpc && grid > 1 && !(grid & 1)) grid -= 1;
    EXB_LAUNCH_STREAMING(c, (k_exsum<N, EE, COPIES, U, NT, PF, ZM>), grid, st, a, n, c.gacc, c.gflags, c.ngroups,
                         c.variant == 9 ? 1 : 0);
}

#ifndef EXBLAS_SUM_COPIES
#define EXBLAS_SUM_COPIES 8
#endif
#ifndef EXBLAS_DOT_COPIES
#define EXBLAS_DOT_COPIES 8
#endif
template <int N, bool EE>
static hipError_t launch_exsum(Ctx &c, const double *a, long long n, long long inca, hipStream_t st)
{
    constexpr int COPIES = (N == 0) ? 16 : EXBLAS_SUM_COPIES;
    if (inca == 1) {
        if constexpr (N == 8 && EE) {
            // tuning variants of the production kernel, selected with exblas_set_tuning() for A/B runs
            switch (c.variant) {
            case 1: run_exsum<N, EE, COPIES, 4, true, false>(c, a, n, st); break;
            case 2: run_exsum<N, EE, COPIES, 8, true, false>(c, a, n, st); break;
            case 3: run_exsum<N, EE, COPIES, 2, true, true>(c, a, n, st); break;
            case 4: run_exsum<N, EE, COPIES, 4, false, true>(c, a, n, st); break;
            case 5: run_exsum<N, EE, COPIES, 2, true, false>(c, a, n, st); break;
            case 6: run_exsum<N, EE, COPIES, 8, true, true>(c, a, n, st); break;
            case 7: run_exsum<N, EE, COPIES, 6, true, true>(c, a, n, st); break;
            case 8: run_exsum<N, EE, COPIES, 4, true, true, 1>(c, a, n, st); break;
            default: run_exsum<N, EE, COPIES, 4, true, true>(c, a, n, st); break;
            }
        } else {
            run_exsum<N, EE, COPIES, 4, true, true>(c, a, n, st);
        }
    } else {
        int grid = grid_for(c, n, BLOCK, c.blocks_per_cu);
        EXB_LAUNCH_STREAMING(c, (k_exsum_strided<N, EE, COPIES>), grid, st, a, n, inca, c.gacc, c.gflags, c.ngroups);
    }
    return hipGetLastError();
}

template <int N, bool EE, int COPIES, int U, bool NT, bool PF, int WPS = 1, bool HALVES = false, int ZM = 0>
static void run_exdot(Ctx &c, const double *a, const double *b, long long n, hipStream_t st)
{
    // Workgroups per CU: 48 at n = 2^28 (placement probe, round 2), but a workgroup should stream about five tiles or
    // more: at n = 2^25 -- a rank's shard of ONE 2^28 vector over 8 GPUs -- 48 per CU leave 1.3 tiles per workgroup
    // (some stream two, most one: 108 us with the finalize), 12 per CU 5.3 (93 us); 2^26: 179 -> 174 us with 24
    // (tools/tune_shard.py).  An explicit smaller setting (exblas_set_tuning / EXBLAS_BPC_DOT) is kept.
    const long long tiles = n / ((long long)BLOCK * 2 * U);
    const int bpc = (int)min((long long)c.bpc_dot, max(4ll, tiles / (5ll * c.num_cu)));
    int grid = grid_for(c, n, (long long)BLOCK * 2 * U, bpc);
    // An ODD number of workgroups: a workgroup's successive tiles (grid tiles = grid x 16 KiB apart) then walk through
    // the 32 KiB period with which the two streams' addresses compete for HBM channels, whatever b - a is.  With the
    // even grid (32 x 256) the step time at n = 2^28 depended on the relative placement of the two vectors, 0.625 ms
    // (b - a = 16 KiB mod 32 KiB) to 0.667 ms (0 mod 32 KiB); odd: 0.625-0.635 ms for every placement
    // (tools/dot_align.py, profiles/r02_exdot_placement.log).
    if (grid > c.num_cu && !(grid & 1)) grid += 1;
    EXB_LAUNCH_STREAMING(c, (k_exdot<N, EE, COPIES, U, NT, PF, WPS, HALVES, ZM>), grid, st, a, b, n, c.gacc, c.gflags,
                         c.ngroups);
}

template <int N, bool EE>
static hipError_t launch_exdot(Ctx &c, const double *a, long long inca, const double *b, long long incb,
                               long long n, hipStream_t st)
{
    constexpr int COPIES = (N == 0) ? 16 : EXBLAS_DOT_COPIES;
    const bool vec = inca == 1 && incb == 1 && (((uintptr_t)a | (uintptr_t)b) & 15u) == 0;
    if (vec) {
        if constexpr (N == 8 && EE) {
            switch (c.variant) {
            case 1: run_exdot<N, EE, COPIES, 2, true, false>(c, a, b, n, st); break;
            case 2: run_exdot<N, EE, COPIES, 4, true, false>(c, a, b, n, st); break;
            case 3: run_exdot<N, EE, COPIES, 1, true, true>(c, a, b, n, st); break;
            case 4: run_exdot<N, EE, COPIES, 2, false, true>(c, a, b, n, st); break;
            case 5: run_exdot<N, EE, COPIES, 1, true, false>(c, a, b, n, st); break;
            case 6: run_exdot<N, EE, COPIES, 2, true, true>(c, a, b, n, st); break;
            case 7: run_exdot<N, EE, COPIES, 3, true, true>(c, a, b, n, st); break;
            case 8: run_exdot<N, EE, COPIES, 4, true, true, 1, true>(c, a, b, n, st); break;
            case 9: run_exdot<N, EE, COPIES, 8, true, true, 1, true>(c, a, b, n, st); break;
            case 10: run_exdot<N, EE, COPIES, 4, true, true>(c, a, b, n, st); break;  // votes by integer ORs (round 1)
            // early-exit votes by one fp64 compare per residue: median 0.658 ms against 0.699 (tools/tune.py, same box)
            default: run_exdot<N, EE, COPIES, 4, true, true, 1, false, 1>(c, a, b, n, st); break;
            }
        } else {
            run_exdot<N, EE, COPIES, 4, true, true, 1, false, SKIP_ZERO_LEVELS<EE> ? 1 : 0>(c, a, b, n, st);
        }
    } else {
        int grid = grid_for(c, n, BLOCK, c.blocks_per_cu);
        EXB_LAUNCH_STREAMING(c, (k_exdot_strided<N, EE, COPIES>), grid, st, a, inca, b, incb, n, c.gacc, c.gflags,
                             c.ngroups);
    }
    return hipGetLastError();
}

// d_ext_out == nullptr: the context's low / high accumulators (ExDOT products below 2^-968 / beyond the double range) are
// folded into the record; otherwise they are exported as two digit sets (EXT_WORDS int64: [low | high]) and the record
// holds the main digits alone
hipError_t finalize_groups(Ctx &c, hipStream_t st, long long *d_out, long long *d_ext_out)
{
    hipLaunchKernelGGL(k_finalize, dim3(1), dim3(64), 0, st, c.gacc, c.ngroups, NL, c.gflags, 0u, 1, d_out,
                       (const long long *)nullptr, d_ext_out);
    return hipGetLastError();
}

// d_ext_sets: the sums of the ranks' exported [low | high] digit sets (folded when the flags say products left the range), or nullptr
hipError_t finalize_sets(const long long *d_sets, int nsets, unsigned flags_or, hipStream_t st, long long *d_out,
                         const long long *d_ext_sets)
{
    hipLaunchKernelGGL(k_finalize, dim3(1), dim3(64), 0, st, const_cast<long long *>(d_sets), nsets, SET_WORDS,
                       (unsigned *)nullptr, flags_or, 0, d_out, d_ext_sets, (long long *)nullptr);
    return hipGetLastError();
}

// variant selection: gpu:ExSUM.cpp:64-84 (fpe < 2 -> superaccumulators only; early_exit buckets 4/6/8)
hipError_t exsum_dispatch(Ctx &c, const double *a, long long n, long long inca, int fpe, int early_exit,
                          hipStream_t st, bool *supported)
{
    *supported = true;
    if (fpe < 2) return launch_exsum<0, false>(c, a, n, inca, st);
    if (early_exit) {
        if (fpe <= 4) return launch_exsum<4, true>(c, a, n, inca, st);
        if (fpe <= 6) return launch_exsum<6, true>(c, a, n, inca, st);
        if (fpe <= 8) return launch_exsum<8, true>(c, a, n, inca, st);
    } else {
        switch (fpe) {
        case 2: return launch_exsum<2, false>(c, a, n, inca, st);
        case 3: return launch_exsum<3, false>(c, a, n, inca, st);
        case 4: return launch_exsum<4, false>(c, a, n, inca, st);
        case 5: return launch_exsum<5, false>(c, a, n, inca, st);
        case 6: return launch_exsum<6, false>(c, a, n, inca, st);
        case 7: return launch_exsum<7, false>(c, a, n, inca, st);
        // fpe > 8 without early exit: the reference builds ExSUM.FPE.cl with -DNBFPE=fpe (gpu:ExSUM.cpp:80-81).  The
        // exact sum does not depend on the expansion size, so the largest instantiation returns the same bits.
        default: return launch_exsum<8, false>(c, a, n, inca, st);
        }
    }
    *supported = false;  // early_exit with fpe > 8: the reference silently returns 0.0 (gpu:ExSUM.cpp:72-83)
    return hipSuccess;
}

// same variant rules as exsum (gpu:ExSUM.cpp:64-84); an unsupported combination yields 0.0 for every segment
hipError_t exsum_segmented_dispatch(const double *values, const long long *offsets, long long nseg, int fpe,
                                    int early_exit, int round_mode, hipStream_t st, double *out)
{
    if (nseg <= 0) return hipSuccess;
    const dim3 grid((unsigned)((nseg + WAVES - 1) / WAVES)), block(BLOCK);
#define SEG_GO(N, EE) \
    hipLaunchKernelGGL((k_exsum_segmented<N, EE>), grid, block, 0, st, values, offsets, nseg, round_mode, out)
    if (fpe < 2) SEG_GO(0, false);
    else if (early_exit) {
        if (fpe <= 4) SEG_GO(4, true);
        else if (fpe <= 6) SEG_GO(6, true);
        else if (fpe <= 8) SEG_GO(8, true);
        else return hipMemsetAsync(out, 0, sizeof(double) * nseg, st);
    } else {
        switch (fpe) {
        case 2: SEG_GO(2, false); break;
        case 3: SEG_GO(3, false); break;
        case 4: SEG_GO(4, false); break;
        case 5: SEG_GO(5, false); break;
        case 6: SEG_GO(6, false); break;
        case 7: SEG_GO(7, false); break;
        default: SEG_GO(8, false); break;  // fpe >= 8 (see exsum_dispatch)
        }
    }
#undef SEG_GO
    return hipGetLastError();
}

// ExDOT.cpp:69-98 (fpe < 3 -> superaccumulators only)
hipError_t exdot_dispatch(Ctx &c, const double *a, long long inca, const double *b, long long incb, long long n,
                          int fpe, int early_exit, hipStream_t st, bool *supported)
{
    *supported = true;
    if (fpe < 3) return launch_exdot<0, false>(c, a, inca, b, incb, n, st);
    if (early_exit) {
        if (fpe <= 4) return launch_exdot<4, true>(c, a, inca, b, incb, n, st);
        if (fpe <= 6) return launch_exdot<6, true>(c, a, inca, b, incb, n, st);
        if (fpe <= 8) return launch_exdot<8, true>(c, a, inca, b, incb, n, st);
    } else {
        switch (fpe) {
        case 3: return launch_exdot<3, false>(c, a, inca, b, incb, n, st);
        case 4: return launch_exdot<4, false>(c, a, inca, b, incb, n, st);
        case 5: return launch_exdot<5, false>(c, a, inca, b, incb, n, st);
        case 6: return launch_exdot<6, false>(c, a, inca, b, incb, n, st);
        case 7: return launch_exdot<7, false>(c, a, inca, b, incb, n, st);
        default: return launch_exdot<8, false>(c, a, inca, b, incb, n, st);  // fpe >= 8 (ExDOT.cpp:93-94)
        }
    }
    *supported = false;
    return hipSuccess;
}

}  // namespace exb
