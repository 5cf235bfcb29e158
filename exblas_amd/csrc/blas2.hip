// blas2.hip -- ExGEMV for gfx950, column-major A (the reference's layout, tests/test.exgemv.gpu.cpp:160).
//
// Replaces the reference's OpenCL kernels gemv / gemvT (src/gpu/blas/blas2/ExGEMV.Superacc.cl:192-392,
// ExGEMV.FPE.cl:199/:382, ExGEMV.FPE.EX.*.cl) and dgemv/dgemvT (DGEMV.cl:13,:86) -- behaviour only:
//   y_i = Round( sum_k A(i,k) * fl(alpha * x_k)  (+)  beta * y_i ),   every y_i correctly rounded.
// The reference gives one work-item per output row a private 39-limb accumulator in *global* memory and
// needs all of x in LDS (256 KiB at n = 32768: it cannot run BASELINE config 4).  Here:
//   'N'  lanes own ROWS (two adjacent rows per lane, one 16-byte load per column, 1 KiB per wave-load);
//        each row has its expansion in registers; columns are split over blockIdx.y so that ~2K workgroups
//        stream A once; the per-(row, k-split) expansions are written out and a second kernel folds them,
//        the row's rare global-memory spill accumulator and beta*y into an LDS accumulator and rounds.
//   'T'  one workgroup per output: a column of A is a contiguous vector, so this is the ExDOT kernel
//        with the final carry-propagate + round done inside the workgroup.
#include "superacc.hip.h"
#include "fpe.hip.h"
#include "exblas_internal.h"

#include <algorithm>

// tools/ only: a device buffer of 2 x (workgroups x 4) uint64 that k_gemvN_fpe_sx fills with per-wave start / end clocks
// (100 MHz wall clock); nullptr (the default) = off.  Not part of the C ABI (not declared in include/).
extern "C" void *exblas_debug_timeline = nullptr;

namespace exb {

constexpr int GV_BLOCK = 256;
constexpr int GV_WAVES = GV_BLOCK / 64;
constexpr int GV_KC = 1024;  // columns of x staged in LDS at a time (8 KiB)
constexpr int GV_TICKET_STRIDE = 4096 + 256;  // bytes between two ticket counters of k_gemvN_fpe_sx

// ---------------------------------------------------------------------------------------------
// 'N', expansion path: two rows per lane
// ---------------------------------------------------------------------------------------------
template <int N, bool EE, bool VEC, int U = 4>
__global__ void __launch_bounds__(GV_BLOCK) k_gemvN_fpe(int m, int n, double alpha, const double *__restrict__ a,
                                                        long long lda, const double *__restrict__ x, long long incx,
                                                        int kper, double *__restrict__ part,
                                                        long long *__restrict__ ws)
{
    __shared__ double xs[GV_KC];
    const int tid = threadIdx.x;
    const long long r0 = ((long long)blockIdx.x * GV_BLOCK + tid) * 2;
    const int ks = blockIdx.y, KS = gridDim.y;
    const int k0 = ks * kper, k1 = min(n, k0 + kper);
    const bool v0 = r0 < m, v1 = r0 + 1 < m;
    double f0[N], f1[N];
#pragma unroll
    for (int i = 0; i < N; ++i) f0[i] = f1[i] = 0.0;
    GlobalSink s0{ws + (v0 ? r0 : 0) * SET_WORDS}, s1{ws + (v1 ? r0 + 1 : 0) * SET_WORDS};

    for (int kc = k0; kc < k1; kc += GV_KC) {
        const int cnt = min(GV_KC, k1 - kc);
        __syncthreads();
        for (int i = tid; i < cnt; i += GV_BLOCK) xs[i] = alpha * x[(long long)(kc + i) * incx];  // rounded fold of alpha
        __syncthreads();
        if (v0) {
            const double *col = a + r0 + lda * kc;
            int k = 0;
            for (; k + U <= cnt; k += U) {
                double ax[U], ay[U];
#pragma unroll
                for (int j = 0; j < U; ++j) {
                    if constexpr (VEC) {
                        const d2_t r = ld2<true>((const d2_t *)(col + lda * (k + j)));
                        ax[j] = r.x;
                        ay[j] = r.y;
                    } else {
                        ax[j] = col[lda * (k + j)];
                        ay[j] = v1 ? col[lda * (k + j) + 1] : 0.0;
                    }
                }
                double p[U], e[U];
#pragma unroll
                for (int j = 0; j < U; ++j) p[j] = two_prod(ax[j], xs[k + j], e[j]);
                fpe_absorb_prod<N, EE, U>(f0, p, e, s0);
#pragma unroll
                for (int j = 0; j < U; ++j) p[j] = two_prod(ay[j], xs[k + j], e[j]);
                fpe_absorb_prod<N, EE, U>(f1, p, e, s1);
            }
            for (; k < cnt; ++k) {
                const double ax = col[lda * k], ay = v1 ? col[lda * k + 1] : 0.0;
                double p[1], e[1];
                p[0] = two_prod(ax, xs[k], e[0]);
                fpe_absorb_prod<N, false, 1>(f0, p, e, s0);
                p[0] = two_prod(ay, xs[k], e[0]);
                fpe_absorb_prod<N, false, 1>(f1, p, e, s1);
            }
        }
    }
    if (v0) {
        double *o = part + ((size_t)r0 * KS + ks) * N;
#pragma unroll
        for (int i = 0; i < N; ++i) o[i] = f0[i];
    }
    if (v1) {
        double *o = part + ((size_t)(r0 + 1) * KS + ks) * N;
#pragma unroll
        for (int i = 0; i < N; ++i) o[i] = f1[i];
    }
}

// x' = fl(alpha * x), contiguous: what k_gemvN_fpe_sx reads with SCALAR loads
__global__ void __launch_bounds__(256) k_scale_x(int n, double alpha, const double *__restrict__ x, long long incx,
                                                 double *__restrict__ xa)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) xa[i] = alpha * x[(long long)i * incx];
}

// 'N', expansion path, x from SGPRs.  Lanes are ROWS, so x_k is the same for every lane of the wave: the pre-scaled
// vector is read with scalar loads (one s_load_dwordx16 per 8 columns) and enters the TwoProd as an SGPR operand -- no
// LDS staging of x, no workgroup barriers, no fragment registers for it.  The early-exit vote is one fp64 compare
// per residue (the lane masks are OR-ed by the scalar unit).  Otherwise as k_gemvN_fpe (two rows per lane).
template <int N, bool EE, int U = 8, bool TL = false>
__global__ void __launch_bounds__(GV_BLOCK, 4) k_gemvN_fpe_sx(int m, int n, const double *__restrict__ a, long long lda,
                                                           const double *__restrict__ xa, int kper,
                                                           double *__restrict__ part, long long *__restrict__ ws, int il,
                                                           int *__restrict__ tickets,
                                                           unsigned long long *__restrict__ timeline)
{
    const int tid = threadIdx.x;
    // TL: tools only (exblas_debug_timeline; a separate instantiation, the production kernels carry no trace of it):
    // start / end clock of every wave
    if (TL && (tid & 63) == 0)
        timeline[(((size_t)blockIdx.y * gridDim.x + blockIdx.x) * GV_WAVES + (tid >> 6)) * 2] = wall_clock64();
    const long long r0 = ((long long)blockIdx.x * GV_BLOCK + tid) * 2;
    const int ks = blockIdx.y, KS = gridDim.y;
    const int k0 = ks * kper, k1 = min(n, k0 + kper);
    const bool v0 = r0 < m;
    double f0[N], f1[N];
#pragma unroll
    for (int i = 0; i < N; ++i) f0[i] = f1[i] = 0.0;
    GlobalSink s0{ws + (v0 ? r0 : 0) * SET_WORDS}, s1{ws + (v0 ? r0 + 1 : 0) * SET_WORDS};
    if (v0) {
        // il = 1: the k splits take the groups of U columns round-robin (split ks: groups ks, ks + KS, ...) instead of
        // one contiguous range each (+1-2 %).  il = 2: in addition a group is every SECOND column of a block of 2U (even
        // ones, then odd ones), so the loads a wave has in flight are 2 lda apart -- the 256 KiB column stride of
        // lda = 32768 is the one that costs 9 % (tools/gemv_lda.py), 512 KiB does not.  il = 3 (production): the groups
        // of il = 2, handed out dynamically.
        static_assert(U == 8, "column groups are blocks of 16");
        const int cs = il >= 2 ? 2 : 1;                          // column step inside a group
        const int nfull = il >= 2 ? (n / (2 * U)) * 2 * U : (n / U) * U;  // columns covered by whole groups
        auto group_k = [&](int gg) { return il >= 2 ? (gg >> 1) * 2 * U + (gg & 1) : gg * U; };
        // one counter per 128 rows, GV_TICKET_STRIDE bytes apart: device-scope atomics execute at the memory side, and 256
        // counters in one 1 KiB line made every draw of the whole chip queue at ONE channel (1.74 ms against 1.46 static)
        int *ctr = tickets + (size_t)(blockIdx.x * GV_WAVES + (tid >> 6)) * (GV_TICKET_STRIDE / sizeof(int));
        auto draw = [&]() {
            int t = 0;
            if ((tid & 63) == 0) t = __hip_atomic_fetch_add(ctr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return t;   // lane 0's value; read with readfirstlane where it is used
        };
        int drawn = 0;
        auto do_group = [&](int g, bool draw_after_loads) {
            const int k = group_k(g);
            const double *col = a + r0 + lda * k;
            double ax[U], ay[U], xs[U];
#pragma unroll
            for (int j = 0; j < U; ++j) {
                const d2_t r = ld2<true>((const d2_t *)(col + lda * (cs * j)));
                ax[j] = r.x;
                ay[j] = r.y;
                xs[j] = xa[k + cs * j];  // uniform address: scalar load
            }
            if (draw_after_loads) drawn = draw();  // behind the loads in the (in-order) return queue: it cannot hold them up
            double p[U], e[U];
#pragma unroll
            for (int j = 0; j < U; ++j) p[j] = two_prod(ax[j], xs[j], e[j]);
            fpe_absorb_prod<N, EE, U, GlobalSink, 1>(f0, p, e, s0);
#pragma unroll
            for (int j = 0; j < U; ++j) p[j] = two_prod(ay[j], xs[j], e[j]);
            fpe_absorb_prod<N, EE, U, GlobalSink, 1>(f1, p, e, s1);
        };
        if (il == 3) {
            // DYNAMIC: the KS waves that own the same 128 rows (one per k split) draw column groups from a shared counter.
            // The four workgroups resident on a CU do not run at the same pace -- the instruction arbiter favours the
            // oldest: at 32768^2 with static ranges they finished at 1010 / 1081 / 1238 / 1412 us (tools/gemv_timeline.py),
            // the CU draining from four waves per SIMD to one over the last 30 % of the kernel.  Every row segment has
            // waves of all four ages, so with a shared counter they all run dry together.  The sum is exact, so which
            // wave adds which columns cannot change a bit.  A ticket is TB groups (32 KiB of matrix per wave); tickets
            // are drawn TWO ahead and the atomic is issued behind the loads of a group, so its (long, device-scope)
            // latency has a whole ticket's compute time to pass and never sits in front of a load in the return queue.
            constexpr int TB = 2;
            const int ntick = (nfull / U + TB - 1) / TB, ngroups = nfull / U;
            int t0 = __builtin_amdgcn_readfirstlane(draw());
            int t1v = draw();
            while (t0 < ntick) {
                const int g = t0 * TB;
                do_group(g, true);                       // draws the ticket after next into `drawn`
                if (g + 1 < ngroups) do_group(g + 1, false);
                t0 = __builtin_amdgcn_readfirstlane(t1v);
                t1v = drawn;
            }
        } else if (il) {
            for (int g = ks; g < nfull / U; g += KS) do_group(g, false);
        }
        // what the groups do not cover: il == 0: this split's whole range; otherwise the last columns, split 0's job
        int k = il ? nfull : k0;
        const int kend = il ? (ks == 0 ? n : 0) : k1;
        const double *col = a + r0 + lda * k;
        if (!il) {
            for (; k + U <= kend; k += U, col += lda * U) {
                double ax[U], ay[U], xs[U];
#pragma unroll
                for (int j = 0; j < U; ++j) {
                    const d2_t r = ld2<true>((const d2_t *)(col + lda * j));
                    ax[j] = r.x;
                    ay[j] = r.y;
                    xs[j] = xa[k + j];
                }
                double p[U], e[U];
#pragma unroll
                for (int j = 0; j < U; ++j) p[j] = two_prod(ax[j], xs[j], e[j]);
                fpe_absorb_prod<N, EE, U, GlobalSink, 1>(f0, p, e, s0);
#pragma unroll
                for (int j = 0; j < U; ++j) p[j] = two_prod(ay[j], xs[j], e[j]);
                fpe_absorb_prod<N, EE, U, GlobalSink, 1>(f1, p, e, s1);
            }
        }
        for (; k < kend; ++k, col += lda) {
            const d2_t r = ld2<true>((const d2_t *)col);
            const double xv = xa[k];
            double p[1], e[1];
            p[0] = two_prod(r.x, xv, e[0]);
            fpe_absorb_prod<N, false, 1>(f0, p, e, s0);
            p[0] = two_prod(r.y, xv, e[0]);
            fpe_absorb_prod<N, false, 1>(f1, p, e, s1);
        }
        double *o = part + ((size_t)r0 * KS + ks) * N;
#pragma unroll
        for (int i = 0; i < N; ++i) o[i] = f0[i];
        o = part + ((size_t)(r0 + 1) * KS + ks) * N;
#pragma unroll
        for (int i = 0; i < N; ++i) o[i] = f1[i];
    }
    if (TL && (tid & 63) == 0)
        timeline[(((size_t)blockIdx.y * gridDim.x + blockIdx.x) * GV_WAVES + (tid >> 6)) * 2 + 1] = wall_clock64();
}

// ---------------------------------------------------------------------------------------------
// 'N', superaccumulators only (fpe == 0): 64 rows per workgroup, one private LDS column per row
// (limb-major [limb][row]: the bank depends on the lane only, so the data-dependent limb index never
// conflicts); the four waves take every fourth column.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(GV_BLOCK) k_gemvN_sa(int m, int n, double alpha, const double *__restrict__ a,
                                                       long long lda, const double *__restrict__ x, long long incx,
                                                       int kper, long long *__restrict__ ws)
{
    __shared__ long long acc[NL * 64];
    __shared__ double xs[GV_KC];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < NL * 64; i += GV_BLOCK) acc[i] = 0;
    const long long row = (long long)blockIdx.x * 64 + lane;
    const bool valid = row < m;
    const int k0 = blockIdx.y * kper, k1 = min(n, k0 + kper);
    unsigned flags = 0;
    LdsSink<64> sink{acc + lane, flags};
    for (int kc = k0; kc < k1; kc += GV_KC) {
        const int cnt = min(GV_KC, k1 - kc);
        __syncthreads();
        for (int i = tid; i < cnt; i += GV_BLOCK) xs[i] = alpha * x[(long long)(kc + i) * incx];
        __syncthreads();
        if (valid) {
            const double *col = a + row + lda * kc;
            // 8 loads in flight per wave: with one (the plain loop) the kernel was bound by memory latency, 1.4 TB/s
            constexpr int UN = 8;
            for (int k = wave; k < cnt; k += GV_WAVES * UN) {
                double v[UN];
#pragma unroll
                for (int u = 0; u < UN; ++u) {
                    const int kk = k + u * GV_WAVES;
                    v[u] = kk < cnt ? __builtin_nontemporal_load(col + lda * kk) : 0.0;
                }
#pragma unroll
                for (int u = 0; u < UN; ++u) {
                    const int kk = k + u * GV_WAVES;
                    if (kk < cnt) {
                        double e, p = two_prod_safe(v[u], xs[kk], e);
                        sink.add(p);
                        if (e != 0.0) sink.add(e);
                    }
                }
            }
        }
    }
    __syncthreads();
    // spill the 64 x 68 limbs into the rows' global accumulators (other k-splits add to the same rows)
    for (int i = tid; i < NL * 64; i += GV_BLOCK) {
        const int l = i >> 6, r = i & 63;
        const long long v = acc[i];
        const long long grow = (long long)blockIdx.x * 64 + r;
        if (v != 0 && grow < m) atomicAdd((unsigned long long *)&ws[grow * SET_WORDS + l], (unsigned long long)v);
    }
    if (valid && flags) {
        for (int k = 0; k < 3; ++k)
            if (flags & (1u << k)) atomicAdd((unsigned long long *)&ws[row * SET_WORDS + NL + k], 1ull);
    }
}

// ---------------------------------------------------------------------------------------------
// 'N', second kernel: one wave per row folds the row's partial expansions, its global spill
// accumulator and beta*y (ExGEMV.Superacc.cl:258-291: beta == 0 ignores y, beta == 1 adds y
// exactly, otherwise TwoProductFMA(beta, y)) and rounds once.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(GV_BLOCK) k_gemv_finish(int rows, int nvals, const double *__restrict__ part,
                                                          const long long *__restrict__ ws, double beta,
                                                          double *__restrict__ y, long long incy, int round_mode)
{
    __shared__ long long acc[GV_WAVES][NL];
    __shared__ unsigned fl[GV_WAVES];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const long long row = (long long)blockIdx.x * GV_WAVES + w;
    const bool valid = row < rows;
    for (int t = lane; t < NL; t += 64) acc[w][t] = 0;
    if (lane == 0) fl[w] = 0;
    __syncthreads();
    if (valid) {
        unsigned flags = 0;
        LdsSink<1> sink{acc[w], flags};
        const double *pr = part + (size_t)row * nvals;
        for (int i = lane; i < nvals; i += 64) {
            const double v = pr[i];
            if (v != 0.0) sink.add(v);
        }
        if (lane == 0 && beta != 0.0) {
            const double yv = y[row * incy];
            if (beta == 1.0) {
                sink.add(yv);
            } else {
                double e, p = two_prod_safe(beta, yv, e);
                sink.add(p);
                if (e != 0.0) sink.add(e);
            }
        }
        const long long *g = ws + row * SET_WORDS;
        for (int t = lane; t < NL; t += 64) {
            const long long v = g[t];
            if (v) atomicAdd((unsigned long long *)&acc[w][t], (unsigned long long)v);
        }
        if (lane < 3 && g[NL + lane] != 0) flags |= 1u << lane;
        if (flags) atomicOr(&fl[w], flags);
    }
    __syncthreads();
    const WaveFinish r = finish_wave(acc[w][lane], lane < NL - 64 ? acc[w][64 + lane] : 0, fl[w]);
    if (valid && lane == 0) y[row * incy] = round_mode ? r.rf : __longlong_as_double((long long)r.ex);
}

// ---------------------------------------------------------------------------------------------
// 'T': y_j = Round(sum_i A(i,j) * fl(alpha*x_i) (+) beta*y_j); column j is contiguous -> ExDOT per workgroup
// ---------------------------------------------------------------------------------------------
// x: the pre-scaled contiguous vector x' = fl(alpha * x) (k_scale_x), so the kernel carries neither the multiply nor incx
template <int N, bool EE, int COPIES, int U = 2, int ZM = 0, int MINW = 1>
__global__ void __launch_bounds__(GV_BLOCK, MINW) k_gemvT(int m, const double *__restrict__ a, long long lda,
                                                    const double *__restrict__ x, double beta,
                                                    double *__restrict__ y, long long incy, int round_mode,
                                                    int stagger, long long *__restrict__ ws)
{
    __shared__ long long s_acc[GV_WAVES * NL * COPIES];
    __shared__ long long merged[NL];
    __shared__ unsigned s_flags;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < GV_WAVES * NL * COPIES; i += GV_BLOCK) s_acc[i] = 0;
    if (tid == 0) s_flags = 0;
    __syncthreads();
    const long long j = blockIdx.x;
    const double *col = a + lda * j;
    unsigned flags = 0;
    LdsSink<COPIES> sink{s_acc + wave * NL * COPIES + (lane & (COPIES - 1)), flags};
    double f[N > 0 ? N : 1];
#pragma unroll
    for (int i = 0; i < (N > 0 ? N : 1); ++i) f[i] = 0.0;

    const bool vec = ((((uintptr_t)col) | ((uintptr_t)x)) & 15u) == 0;
    long long done = 0;
    if (vec) {
        const d2_t *va = (const d2_t *)col, *vx = (const d2_t *)x;
        const long long nv = m >> 1, tile = (long long)GV_BLOCK * U, ntiles = nv / tile;
        // Every column of a power-of-two lda starts on the same HBM channel; since the sum is order-free, each
        // workgroup starts its sweep at a different tile (and wraps) so that concurrent columns spread over the
        // channels instead of marching through them in lockstep.
        const long long t0 = stagger ? (j * 37) % (ntiles > 0 ? ntiles : 1) : 0;
        // a wave owns U KiB of the tile and each of its loads covers 1 KiB of it: the U loads of a lane are 1 KiB apart,
        // i.e. ONE address register pair + immediate offsets (they were 4 KiB apart, beyond the 12-bit offset: a 64-bit
        // add per load)
        const int lane_off = (tid >> 6) * (64 * U) + (tid & 63);
        if (ntiles > 0) {
            // two register sets filled alternately with unconditional loads (see k_exdot in blas1.hip)
            d2_t ra[U], rx[U], rb[U], ry[U];
            auto fill = [&](long long tt, d2_t (&qa)[U], d2_t (&qx)[U]) {
                long long t = (tt < ntiles ? tt : ntiles - 1) + t0;
                if (t >= ntiles) t -= ntiles;
                const long long base = t * tile + lane_off;
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    qa[u] = ld2<true>(va + base + u * 64);
                    qx[u] = vx[base + u * 64];  // x is re-read by every workgroup: keep it cacheable
                }
            };
            // returns (wave-uniform) whether the tile spilled: the data's exponent range outgrows the expansion
            auto absorb = [&](d2_t (&qa)[U], d2_t (&qx)[U]) {
                double p[2 * U], e[2 * U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    p[2 * u] = two_prod(qa[u].x, qx[u].x, e[2 * u]);
                    p[2 * u + 1] = two_prod(qa[u].y, qx[u].y, e[2 * u + 1]);
                }
                return fpe_absorb_prod<N, EE, 2 * U, LdsSink<COPIES>, ZM>(f, p, e, sink);
            };
            // After a spill the REST OF THE COLUMN goes straight to the integer accumulator in a loop of its own (a
            // column is 16 tiles per lane at m = 32768: shorter than the shortest bypass span of fpe_absorb_prod_adaptive,
            // which this replaces here).  Keeping the bypass as a per-tile flag inside the one loop cost the hot path 7
            // v_mov_b64 (the expansion's registers merged after the branch) and 8 hoisted v_bfe per tile.
            long long tt = 0;
            bool direct = false;
            fill(0, ra, rx);
            for (;;) {
                fill(tt + 1, rb, ry);
                direct = absorb(ra, rx);
                if (++tt >= ntiles || direct) break;
                fill(tt + 1, ra, rx);
                direct = absorb(rb, ry);
                if (++tt >= ntiles || direct) break;
            }
            if (N > 0 && direct) {
                for (; tt < ntiles; ++tt) {
                    fill(tt, ra, rx);
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        double e0, e1;
                        const double p0 = two_prod(ra[u].x, rx[u].x, e0), p1 = two_prod(ra[u].y, rx[u].y, e1);
                        sink_product(sink, p0, e0);
                        sink_product(sink, p1, e1);
                    }
                }
            }
        }
        done = ntiles * tile * 2;
    }
    for (long long i = done + tid; i < m; i += GV_BLOCK) {
        double p[1], e[1];
        p[0] = two_prod(col[i], x[i], e[0]);
        fpe_absorb_prod<N, false, 1>(f, p, e, sink);
    }
    fpe_flush_sink<N>(f, sink);
    if (ws == nullptr && tid == 0 && beta != 0.0) {
        const double yv = y[j * incy];
        if (beta == 1.0) {
            sink.add(yv);
        } else {
            double e, p = two_prod_safe(beta, yv, e);
            sink.add(p);
            if (e != 0.0) sink.add(e);
        }
    }
    if (flags) atomicOr(&s_flags, flags);
    __syncthreads();
    for (int l = tid; l < NL; l += GV_BLOCK) {
        long long sum = 0;
#pragma unroll
        for (int w = 0; w < GV_WAVES; ++w)
#pragma unroll
            for (int c = 0; c < COPIES; ++c) sum += s_acc[(w * NL + l) * COPIES + c];
        if (ws) ws[j * SET_WORDS + l] = sum;   // deferred (A/B variant): k_gemv_finish carries, adds beta * y and rounds
        else merged[l] = sum;
    }
    if (ws) {
        // A/B variant: the column's 68 limbs + 3 non-finite indicators go to memory and ONE k_gemv_finish launch rounds
        // all n outputs (a wave per output, ~30 us at n = 32768)
        if (tid < 3) ws[j * SET_WORDS + NL + tid] = (s_flags >> tid) & 1u;
        return;
    }
    __syncthreads();
    if (wave == 0) {  // one wavefront carries, cuts and rounds (shuffles + ballots only)
        const WaveFinish r = finish_wave(merged[lane], lane < NL - 64 ? merged[64 + lane] : 0, s_flags);
        if (lane == 0) y[j * incy] = round_mode ? r.rf : __longlong_as_double((long long)r.ex);
    }
}

// ---------------------------------------------------------------------------------------------
// fpe == 1: the plain, NON-reproducible fp64 GEMV the reference ships as its baseline (DGEMV.cl)
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(GV_BLOCK) k_dgemvN(int m, int n, double alpha, const double *__restrict__ a,
                                                     long long lda, const double *__restrict__ x, long long incx,
                                                     double beta, double *__restrict__ y, long long incy)
{
    const long long row = (long long)blockIdx.x * GV_BLOCK + threadIdx.x;
    if (row >= m) return;
    double sum = 0.0;
    for (int k = 0; k < n; ++k) sum += alpha * a[row + lda * k] * x[(long long)k * incx];
    y[row * incy] = (beta == 0.0) ? sum : sum + beta * y[row * incy];
}

__global__ void __launch_bounds__(64) k_dgemvT(int m, double alpha, const double *__restrict__ a, long long lda,
                                               const double *__restrict__ x, long long incx, double beta,
                                               double *__restrict__ y, long long incy)
{
    const long long j = blockIdx.x;
    double sum = 0.0;
    for (long long i = threadIdx.x; i < m; i += 64) sum += alpha * a[lda * j + i] * x[i * incx];
    for (int off = 32; off > 0; off >>= 1) sum += __shfl_down(sum, off);
    if (threadIdx.x == 0) y[j * incy] = (beta == 0.0) ? sum : sum + beta * y[j * incy];
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
template <int N, bool EE>
static hipError_t gemvN_fpe(Ctx &c, int m, int n, double alpha, const double *a, int lda, const double *x, int incx,
                            double beta, double *y, int incy, int round_mode, hipStream_t st)
{
    const int gx = (m + 2 * GV_BLOCK - 1) / (2 * GV_BLOCK);
    const int wg_target = c.num_cu * (c.variant == 3 ? 8 : (c.variant == 4 ? 16 : (c.variant == 5 ? 32 : 4)));  // 4/CU: 1.43 ms, 8: 1.47, 16: 1.5-1.6
    int KS = (wg_target + gx - 1) / gx;
    const int max_ks = (n + 63) / 64;
    if (KS > max_ks) KS = max_ks;
    if (KS > 128) KS = 128;
    if (KS < 1) KS = 1;
    int kper = (n + KS - 1) / KS;
    kper = (kper + 7) & ~7;
    KS = (n + kper - 1) / kper;
    const size_t ws_bytes = (size_t)m * SET_WORDS * sizeof(long long);
    const size_t part_bytes = (size_t)m * KS * N * sizeof(double);
    const size_t xa_bytes = ((size_t)n * sizeof(double) + 255) & ~(size_t)255;
    const size_t tk_bytes = (size_t)gx * GV_WAVES * GV_TICKET_STRIDE;  // group tickets, one per 128 rows
    hipError_t e;
    char *base = (char *)workspace(c, ws_bytes + tk_bytes + part_bytes + xa_bytes, st, &e);
    if (!base) return e;
    long long *ws = (long long *)base;
    int *tickets = (int *)(base + ws_bytes);
    double *part = (double *)(base + ws_bytes + tk_bytes);
    double *xa = (double *)(base + ws_bytes + tk_bytes + part_bytes);
    e = hipMemsetAsync(ws, 0, ws_bytes + tk_bytes, st);  // the rows' spill accumulators and the tickets
    if (e != hipSuccess) return e;
    const bool vec = (m % 2 == 0) && (lda % 2 == 0) && (((uintptr_t)a) & 15u) == 0;
    dim3 grid(gx, KS);
    if (vec && c.variant != 1 && c.variant != 2 && c.variant != 6) {
        // production: x from SGPRs (scalar loads of the pre-scaled vector), early-exit votes by fp64 compares.  Against
        // the LDS-staged kernel below (variant 6) in one process: 1.40-1.45 ms against 1.41-1.48 at 32768^2
        hipLaunchKernelGGL(k_scale_x, dim3((n + 255) / 256), dim3(256), 0, st, n, alpha, x, (long long)incx, xa);
        const int il = c.variant == 9 ? 0 : (c.variant == 10 ? 1 : (c.variant == 11 ? 2 : 3));
        bool traced = false;
        if constexpr (N == 8 && EE) {
            if (exblas_debug_timeline) {
                traced = true;
                hipLaunchKernelGGL((k_gemvN_fpe_sx<N, EE, 8, true>), grid, dim3(GV_BLOCK), 0, st, m, n, a, (long long)lda, xa,
                                   kper, part, ws, il, tickets, (unsigned long long *)exblas_debug_timeline);
            }
        }
        if (!traced)
            hipLaunchKernelGGL((k_gemvN_fpe_sx<N, EE, 8>), grid, dim3(GV_BLOCK), 0, st, m, n, a, (long long)lda, xa, kper, part,
                               ws, il, tickets, (unsigned long long *)nullptr);
    } else if (vec && c.variant == 1)
        hipLaunchKernelGGL((k_gemvN_fpe<N, EE, true, 4>), grid, dim3(GV_BLOCK), 0, st, m, n, alpha, a, (long long)lda, x,
                           (long long)incx, kper, part, ws);
    else if (vec && N == 8 && EE && c.variant == 2)
        hipLaunchKernelGGL((k_gemvN_fpe<N, EE, true, 2>), grid, dim3(GV_BLOCK), 0, st, m, n, alpha, a, (long long)lda, x,
                           (long long)incx, kper, part, ws);
    else if (vec)  // (variant 6) x staged in LDS; 8 columns per step: 6.06 TB/s vs 5.78 with 4 (tools/tune_gemv.py, 32768^2)
        hipLaunchKernelGGL((k_gemvN_fpe<N, EE, true, 8>), grid, dim3(GV_BLOCK), 0, st, m, n, alpha, a, (long long)lda, x,
                           (long long)incx, kper, part, ws);
    else
        hipLaunchKernelGGL((k_gemvN_fpe<N, EE, false>), grid, dim3(GV_BLOCK), 0, st, m, n, alpha, a, (long long)lda, x,
                           (long long)incx, kper, part, ws);
    hipLaunchKernelGGL(k_gemv_finish, dim3((m + GV_WAVES - 1) / GV_WAVES), dim3(GV_BLOCK), 0, st, m, KS * N, part, ws,
                       beta, y, (long long)incy, round_mode);
    return hipGetLastError();
}

static hipError_t gemvN_sa(Ctx &c, int m, int n, double alpha, const double *a, int lda, const double *x, int incx,
                           double beta, double *y, int incy, int round_mode, hipStream_t st)
{
    const int gx = (m + 63) / 64;
    int KS = (c.num_cu * 4 + gx - 1) / gx;
    const int max_ks = (n + 255) / 256;
    if (KS > max_ks) KS = max_ks;
    if (KS < 1) KS = 1;
    const int kper = (n + KS - 1) / KS;
    KS = (n + kper - 1) / kper;
    const size_t ws_bytes = (size_t)m * SET_WORDS * sizeof(long long);
    hipError_t e;
    long long *ws = (long long *)workspace(c, ws_bytes, st, &e);
    if (!ws) return e;
    e = hipMemsetAsync(ws, 0, ws_bytes, st);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_gemvN_sa, dim3(gx, KS), dim3(GV_BLOCK), 0, st, m, n, alpha, a, (long long)lda, x,
                       (long long)incx, kper, ws);
    hipLaunchKernelGGL(k_gemv_finish, dim3((m + GV_WAVES - 1) / GV_WAVES), dim3(GV_BLOCK), 0, st, m, 0,
                       (const double *)nullptr, ws, beta, y, (long long)incy, round_mode);
    return hipGetLastError();
}

template <int N, bool EE>
static hipError_t gemvT(Ctx &c, int m, int n, double alpha, const double *a, int lda, const double *x, int incx,
                        double beta, double *y, int incy, int round_mode, hipStream_t st)
{
    constexpr int COPIES = (N == 0) ? 16 : 8;
    // A/B on MI355X (tools/tune_gemv.py, 32768^2): 4 loads per stream in flight + staggered sweeps 5.3 TB/s;
    // 2 loads 5.0; no stagger 4.9; a wave-per-column form (no workgroup barriers) and a persistent form were slower
    // Rounding inside the workgroup (production) or deferred to one k_gemv_finish launch over all n outputs (variant 12:
    // the column's limbs go through memory).  Measured at 32768^2: 1.616 ms against 1.649 deferred -- the ~6 us single-wave
    // carry + round chain at the end of a workgroup's 39 us life is covered by the other workgroups of the CU; the extra
    // launch is not.
    long long *ws = nullptr;
    // x' = fl(alpha * x), contiguous, once (m doubles behind the deferred-finish area of the workspace)
    const size_t ws_bytes = c.variant == 12 ? (size_t)n * SET_WORDS * sizeof(long long) : 0;
    {
        hipError_t e;
        char *base = (char *)workspace(c, ws_bytes + (size_t)m * sizeof(double), st, &e);
        if (!base) return e;
        if (ws_bytes) ws = (long long *)base;
        double *xa = (double *)(base + ws_bytes);
        hipLaunchKernelGGL(k_scale_x, dim3((m + 255) / 256), dim3(256), 0, st, m, alpha, x, (long long)incx, xa);
        x = xa;
    }
#define GVT_LAUNCH(...)                                                                                                \
    hipLaunchKernelGGL((k_gemvT<N, EE, COPIES, __VA_ARGS__>), dim3(n), dim3(GV_BLOCK), 0, st, m, a, (long long)lda, x, \
                       beta, y, (long long)incy, round_mode, c.variant == 1 ? 0 : 1, ws)
    if (c.variant == 1) GVT_LAUNCH(2);             // two loads per stream in flight, no stagger
    else if (c.variant == 13) GVT_LAUNCH(4, 1, 4); // A/B: four waves per SIMD (the compiler must fit 128 registers: it spills)
    else if (c.variant == 6) GVT_LAUNCH(4);        // A/B: early-exit votes by integer ORs of the residue words
    else GVT_LAUNCH(4, 1);                         // early-exit votes by fp64 compares: 1.58-1.61 ms against 1.64-1.66 at 32768^2
#undef GVT_LAUNCH
    if (ws)
        hipLaunchKernelGGL(k_gemv_finish, dim3((n + GV_WAVES - 1) / GV_WAVES), dim3(GV_BLOCK), 0, st, n, 0,
                           (const double *)nullptr, ws, beta, y, (long long)incy, round_mode);
    return hipGetLastError();
}

template <int N, bool EE>
static hipError_t gemv_variant(Ctx &c, bool trans, int m, int n, double alpha, const double *a, int lda,
                               const double *x, int incx, double beta, double *y, int incy, int round_mode,
                               hipStream_t st)
{
    if (trans) return gemvT<N, EE>(c, m, n, alpha, a, lda, x, incx, beta, y, incy, round_mode, st);
    if constexpr (N == 0) return gemvN_sa(c, m, n, alpha, a, lda, x, incx, beta, y, incy, round_mode, st);
    else return gemvN_fpe<N, EE>(c, m, n, alpha, a, lda, x, incx, beta, y, incy, round_mode, st);
}

// variant selection: ExGEMV.cpp:81-107 (fpe == 0 superaccumulators, fpe == 1 plain DGEMV, early-exit buckets 4/6/8)
hipError_t exgemv_dispatch(Ctx &c, char transa, int m, int n, double alpha, const double *a, int lda, const double *x,
                           int incx, double beta, double *y, int incy, int fpe, int early_exit, int round_mode,
                           hipStream_t st)
{
    if (m <= 0 || n <= 0) return hipSuccess;
    const bool t = (transa == 'T' || transa == 't');
#define GV_ARGS c, t, m, n, alpha, a, lda, x, incx, beta, y, incy, round_mode, st
    if (fpe == 0) return gemv_variant<0, false>(GV_ARGS);
    if (fpe == 1) {
        if (t)
            hipLaunchKernelGGL(k_dgemvT, dim3(n), dim3(64), 0, st, m, alpha, a, (long long)lda, x, (long long)incx, beta,
                               y, (long long)incy);
        else
            hipLaunchKernelGGL(k_dgemvN, dim3((m + GV_BLOCK - 1) / GV_BLOCK), dim3(GV_BLOCK), 0, st, m, n, alpha, a,
                               (long long)lda, x, (long long)incx, beta, y, (long long)incy);
        return hipGetLastError();
    }
    if (early_exit) {
        if (fpe <= 4) return gemv_variant<4, true>(GV_ARGS);
        if (fpe <= 6) return gemv_variant<6, true>(GV_ARGS);
        if (fpe <= 8) return gemv_variant<8, true>(GV_ARGS);
        return hipSuccess;  // early_exit with fpe > 8: y untouched, the reference's silent return (ExGEMV.cpp:96-106)
    }
    switch (fpe) {
    case 2: return gemv_variant<2, false>(GV_ARGS);
    case 3: return gemv_variant<3, false>(GV_ARGS);
    case 4: return gemv_variant<4, false>(GV_ARGS);
    case 5: return gemv_variant<5, false>(GV_ARGS);
    case 6: return gemv_variant<6, false>(GV_ARGS);
    case 7: return gemv_variant<7, false>(GV_ARGS);
    default: return gemv_variant<8, false>(GV_ARGS);  // fpe >= 8: ExGEMV.FPE.cl with NBFPE = fpe (ExGEMV.cpp:103-104), same bits
    }
#undef GV_ARGS
}

}  // namespace exb

// tools/ only: what the occupancy API says for the production 'N' kernel (blocks of 256 threads per CU)
extern "C" int exblas_debug_gemv_occupancy(void)
{
    int nb = -1;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, exb::k_gemvN_fpe_sx<8, true, 8, false>, exb::GV_BLOCK, 0) != hipSuccess)
        return -1;
    return nb;
}
