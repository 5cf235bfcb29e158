// capi.hip -- context, generators and the C ABI / C++ API of libexblas.so (see include/exblas_hip.h).
//
// The C++ functions exsum/exdot/exgemv/exgemm at the bottom carry the exact signatures of the
// reference's public headers (include/blas1.hpp:48,74; blas2.hpp:57,95; blas3.hpp:56) and replace
// src/gpu/blas/blas{1,2,3}/Ex*.cpp: same argument meaning, same variant dispatch, same error
// behaviour (print + exit on fpe < 0 or device failure, 0.0 for Ng <= 0 / unsupported variants).
#include "../../include/exblas_hip.h"
#include "../../include/blas1.hpp"
#include "../../include/blas2.hpp"
#include "../../include/blas3.hpp"
#include "exblas_internal.h"
#include "superacc.hip.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <atomic>
#include <thread>
#include <vector>

namespace exb {

static_assert(OUT_WORDS == EXBLAS_OUT_WORDS && OUT_CANON == EXBLAS_OUT_CANON && OUT_DIGITS == EXBLAS_OUT_DIGITS &&
                  NL == EXBLAS_NDIGITS && SET_WORDS == EXBLAS_SET_WORDS && CANON == EXBLAS_NCANON && OUT_EXACT == EXBLAS_OUT_EXACT &&
                  OUT_REFMODE == EXBLAS_OUT_REFMODE && OUT_FLAGS == EXBLAS_OUT_FLAGS,
              "record layout out of sync with include/exblas_hip.h");

[[noreturn]] void die(const char *what, hipError_t e, const char *file, int line)
{
    // the reference prints and exit(EXIT_FAILURE)s on any backend failure (gpu:ExSUM.cpp:111-115)
    fprintf(stderr, "exblas(hip): %s failed: %s (%s:%d)\n", what, hipGetErrorString(e), file, line);
    exit(EXIT_FAILURE);
}

static int env_int(const char *name, int dflt)
{
    const char *s = getenv(name);
    return (s && *s) ? atoi(s) : dflt;
}

std::atomic<bool> g_comm_created{false};  // set by comm.hip when a communicator of more than one rank is created

static int g_round_mode = -1;
int round_mode()
{
    if (g_round_mode < 0) {
        const char *s = getenv("EXBLAS_ROUND");
        g_round_mode = (s && (s[0] == 'r' || s[0] == 'R' || s[0] == '1')) ? 1 : 0;
    }
    return g_round_mode;
}

static constexpr int MAX_DEV = 16;
static Ctx g_ctx[MAX_LAYERS][MAX_DEV];
static std::mutex g_ctx_mu;
static int g_last_layer[MAX_DEV];  // which layer ran the most recent exgemv / exgemm / extrsv (diagnostics only)

// run f on every context of `device` that exists (tuning knobs are per device, not per layer)
template <class F>
static void for_each_layer(int device, F &&f)
{
    for (int l = 0; l < MAX_LAYERS; ++l)
        if (g_ctx[l][device].device >= 0) f(g_ctx[l][device]);
}

static int current_device()
{
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess) d = 0;
    return d < MAX_DEV ? d : 0;
}

// creates the device objects of a context (accumulators, flags, record buffers, stream) on `device`; knobs come from the
// environment, then from the device's layer-0 context when that exists (what the API has set so far)
static void init_ctx(Ctx &c, int device, int layer)
{
    c.layer = layer;
    int prev = 0;
    EXB_CHECK(hipGetDevice(&prev));
    EXB_CHECK(hipSetDevice(device));
    hipDeviceProp_t prop;
    EXB_CHECK(hipGetDeviceProperties(&prop, device));
    c.num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    c.blocks_per_cu = env_int("EXBLAS_BLOCKS_PER_CU", 8);
    c.bpc_sum = env_int("EXBLAS_BPC_SUM", 2);
    c.bpc_dot = env_int("EXBLAS_BPC_DOT", 48);
    c.bpc_sa = env_int("EXBLAS_BPC_SA", 3);
    c.bpc_heavy = env_int("EXBLAS_BPC_HEAVY", 4);
    c.ngroups = env_int("EXBLAS_NGROUPS", 32);
    c.grid_adj = env_int("EXBLAS_GRID_ADJ", 0);
    if (c.ngroups < 1) c.ngroups = 1;
    c.variant = env_int("EXBLAS_VARIANT", 0);
    c.gemm_path = env_int("EXBLAS_GEMM_PATH", 0);
    if (layer > 0 && g_ctx[0][device].device >= 0) {  // knobs set through the API so far apply to every layer
        const Ctx &z = g_ctx[0][device];
        c.blocks_per_cu = z.blocks_per_cu; c.bpc_sum = z.bpc_sum; c.bpc_dot = z.bpc_dot; c.bpc_sa = z.bpc_sa;
        c.bpc_heavy = z.bpc_heavy;
        c.grid_adj = z.grid_adj;
        c.ngroups = z.ngroups; c.variant = z.variant; c.gemm_path = z.gemm_path;
        c.gemm_max_slices = z.gemm_max_slices;
        c.gemm_max_moduli = z.gemm_max_moduli;
    }
    EXB_CHECK(crt_tables_upload());
    EXB_CHECK(hipMalloc(&c.gacc_all, 2 * sizeof(long long) * NL * c.ngroups));
    EXB_CHECK(hipMemset(c.gacc_all, 0, 2 * sizeof(long long) * NL * c.ngroups));
    // per slot: the flag word (own 64-byte line) + the low and high accumulators of ExDOT (superacc.hip.h: low_acc_of, high_acc_of)
    EXB_CHECK(hipMalloc(&c.gflags_all, 2 * FLAG_BLOCK_BYTES));
    EXB_CHECK(hipMemset(c.gflags_all, 0, 2 * FLAG_BLOCK_BYTES));
    c.gacc = c.gacc_all;
    c.gflags = c.gflags_all;
    c.slot = 0;
    EXB_CHECK(hipMalloc(&c.d_record, sizeof(long long) * OUT_WORDS));
    // portable: the record of one device's part is copied to the first device when a host call spans several
    EXB_CHECK(hipHostMalloc(&c.h_record, sizeof(long long) * OUT_WORDS, hipHostMallocPortable));
    EXB_CHECK(hipStreamCreateWithFlags(&c.stream, hipStreamNonBlocking));
    EXB_CHECK(hipDeviceSynchronize());
    EXB_CHECK(hipSetDevice(prev));
    c.device = device;
}

Ctx &ctx(int device, int layer)
{
    if (device < 0) {
        hipError_t e = hipGetDevice(&device);
        if (e != hipSuccess) {
            // no usable HIP device: the product path refuses to run (there is no CPU fallback)
            fprintf(stderr, "exblas(hip): no HIP device available: %s\n", hipGetErrorString(e));
            exit(EXIT_FAILURE);
        }
    }
    if (device >= MAX_DEV) {
        fprintf(stderr, "exblas(hip): device index %d out of range\n", device);
        exit(EXIT_FAILURE);
    }
    if (layer < 0 || layer >= MAX_LAYERS) layer = 0;
    Ctx &c = g_ctx[layer][device];
    std::lock_guard<std::mutex> lk(g_ctx_mu);
    if (c.device < 0) init_ctx(c, device, layer);
    return c;
}

Ctx &default_ctx() { return ctx(-1); }

void *stage_buf(Ctx &c, int slot, size_t bytes)
{
    if (bytes > c.stage_bytes[slot]) {
        if (c.stage[slot]) EXB_CHECK(hipFree(c.stage[slot]));
        size_t cap = bytes + (bytes >> 3) + 4096;
        EXB_CHECK(hipMalloc(&c.stage[slot], cap));
        c.stage_bytes[slot] = cap;
    }
    return c.stage[slot];
}

void *workspace(Ctx &c, size_t bytes, hipStream_t st, hipError_t *err)
{
    *err = hipSuccess;
    if (bytes > c.ws_bytes) {
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (st && hipStreamIsCapturing(st, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone) {
            // hipMalloc is illegal during capture, and the graph would bake in a pointer that the next growth moves
            *err = hipErrorStreamCaptureUnsupported;
            return nullptr;
        }
        // Allocate FIRST: only a successful growth may change the context.  On failure ws / ws_bytes / retired stay as
        // they were (the live block must never also sit in `retired`: a later exblas_release_retired_workspaces()
        // would free memory the next call launches into) and the failed hipMalloc's error is cleared, so that the
        // fallback path's hipGetLastError() does not report it.
        size_t cap = bytes + (c.ws_bytes >> 1);  // geometric growth bounds what the parked blocks can add up to
        void *p = nullptr;
        hipError_t e = hipMalloc(&p, cap);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            cap = bytes;
            e = hipMalloc(&p, cap);
        }
        if (e != hipSuccess) {
            (void)hipGetLastError();
            *err = e;
            return nullptr;
        }
        // the old block is parked, not freed: a graph captured earlier may still replay into it
        if (c.ws) c.retired.push_back(c.ws);
        c.ws = p;
        c.ws_bytes = cap;
    }
    return c.ws;
}

// ---------------------------------------------------------------------------------------------
// counter-based generators: the device twin of oracle/exblas_oracle.c:orc_gen_one (integer math,
// exact conversions and power-of-two scalings only, so the bits are identical on CPU and GPU)
// ---------------------------------------------------------------------------------------------
__host__ __device__ inline unsigned long long mix64(unsigned long long z)
{
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__device__ inline unsigned long long rnd(unsigned long long seed, unsigned long long i, unsigned long long k)
{
    return mix64(seed * 0xD1342543DE82EF95ull + (2 * i + k + 1) * 0x9E3779B97F4A7C15ull);
}
__device__ inline double pow2i(int e) { return __longlong_as_double((long long)(e + 1023) << 52); }
__device__ inline double mant12(unsigned long long r)
{
    return __longlong_as_double((long long)(0x3FF0000000000000ull | (r >> 12)));
}
__device__ inline double mant_signed(unsigned long long r)
{
    long long k = (long long)(r >> 11);
    return (double)(2 * k - (1ll << 53)) * 0x1p-53;
}
__device__ inline unsigned uni(unsigned long long r, unsigned range)
{
    return (unsigned)(((r >> 32) * (unsigned long long)range) >> 32);
}

__global__ void __launch_bounds__(256) k_gen(int kind, unsigned long long seed, long long first, long long count,
                                             long long n, int i0, int i1, double dscale, double *out)
{
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < count;
         t += (long long)gridDim.x * blockDim.x) {
        const long long i = first + t;
        const unsigned long long r0 = rnd(seed, (unsigned long long)i, 0), r1 = rnd(seed, (unsigned long long)i, 1);
        double v = 0.0;
        switch (kind) {
        case EXBLAS_GEN_NAIVE: v = 1.1; break;
        case EXBLAS_GEN_FPUNIFORM:
        case EXBLAS_GEN_FPUNIFORM_SIGNED: {
            int e = i1 - i0 + (i0 > 0 ? (int)uni(r1, (unsigned)i0) : 0);
            v = mant12(r0) * pow2i(e);
            if (kind == EXBLAS_GEN_FPUNIFORM_SIGNED && (r1 & 1)) v = -v;
            break;
        }
        case EXBLAS_GEN_LOGNORMAL: {
            long long z = (long long)(r1 & 0xffff) + (long long)((r1 >> 16) & 0xffff) +
                          (long long)((r1 >> 32) & 0xffff) + (long long)((r1 >> 48) & 0xffff) - 2 * 65535;
            int e = (int)rint((double)z * dscale) + i0;
            e = e > 1000 ? 1000 : (e < -1000 ? -1000 : e);
            v = mant12(r0) * pow2i(e);
            break;
        }
        case EXBLAS_GEN_ILLCOND: {
            const int bh = i0;
            const long long n2 = n / 2;
            int e;
            if (i < n2) e = (i == 0) ? bh + 1 : (int)uni(r1, (unsigned)(bh + 1));
            else e = (n - n2 > 0) ? (int)(((i - n2) * (long long)bh) / (n - n2)) : 0;
            v = mant_signed(r0) * pow2i(e);
            break;
        }
        case EXBLAS_GEN_CANCEL: {
            const long long h = n / 2;
            if (i >= 2 * h) v = 0.0;
            else if (i == h - 1) v = 1.0;
            else if (i == 2 * h - 1) v = 0x1p-60;
            else {
                const long long j = (i < h) ? i : i - h;
                const unsigned long long q0 = rnd(seed, (unsigned long long)j, 0),
                                         q1 = rnd(seed, (unsigned long long)j, 1);
                double w = mant_signed(q0) * pow2i(i0 > 0 ? (int)uni(q1, (unsigned)i0) : 0);
                v = (i < h) ? w : -w;
            }
            break;
        }
        default: break;
        }
        out[t] = v;
    }
}

// plain (inexact) streaming sum: read-bandwidth probe
typedef double d2_t __attribute__((ext_vector_type(2)));
__global__ void __launch_bounds__(256) k_stream_read(const double *a, long long n, double *sink)
{
    const d2_t *v = (const d2_t *)a;
    const long long nv = n >> 1;
    double s0 = 0, s1 = 0;
    constexpr int U = 4;
    const long long tile = 256ll * U, ntiles = nv / tile;
    for (long long t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const d2_t *p = v + t * tile + threadIdx.x;
        d2_t r[U];
#pragma unroll
        for (int u = 0; u < U; ++u) r[u] = __builtin_nontemporal_load(p + u * 256);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            s0 += r[u].x;
            s1 += r[u].y;
        }
    }
    if (s0 + s1 == 0x1.23456789abcdep-333) *sink = s0;  // keeps the loads alive, never true in practice
}

// plain (inexact) two-stream dot: read-bandwidth probe for the ExDOT access pattern
__global__ void __launch_bounds__(256) k_stream_read2(const double *a, const double *b, long long n, double *sink)
{
    const d2_t *va = (const d2_t *)a, *vb = (const d2_t *)b;
    const long long nv = n >> 1;
    double s0 = 0, s1 = 0;
    constexpr int U = 4;
    const long long tile = 256ll * U, ntiles = nv / tile;
    for (long long t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const long long base = t * tile + threadIdx.x;
        d2_t r[U], q[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            r[u] = __builtin_nontemporal_load(va + base + u * 256);
            q[u] = __builtin_nontemporal_load(vb + base + u * 256);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            s0 += r[u].x * q[u].x;
            s1 += r[u].y * q[u].y;
        }
    }
    if (s0 + s1 == 0x1.23456789abcdep-333) *sink = s0;
}

}  // namespace exb

using namespace exb;

// =============================================================================================
// C ABI
// =============================================================================================
extern "C" {

int exblas_hip_init(int device)
{
    ctx(device);
    return 0;
}

int exblas_hip_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

const char *exblas_hip_version(void) { return "exblas-hip 0.1 (gfx950)"; }

void exblas_set_round_mode(int mode) { g_round_mode = mode ? 1 : 0; }

int exblas_set_tuning(int blocks_per_cu, int ngroups, int variant)
{
    ctx(-1);
    for_each_layer(current_device(), [&](Ctx &c) {
        std::lock_guard<std::mutex> lk(c.mu);
        if (blocks_per_cu > 0) c.blocks_per_cu = c.bpc_sum = c.bpc_dot = c.bpc_sa = c.bpc_heavy = blocks_per_cu;
        if (ngroups > 0 && ngroups != c.ngroups) {
            EXB_CHECK(hipDeviceSynchronize());
            EXB_CHECK(hipFree(c.gacc_all));
            c.ngroups = ngroups;
            EXB_CHECK(hipMalloc(&c.gacc_all, 2 * sizeof(long long) * NL * c.ngroups));
            EXB_CHECK(hipMemset(c.gacc_all, 0, 2 * sizeof(long long) * NL * c.ngroups));
            c.gacc = c.gacc_all + (size_t)c.slot * NL * c.ngroups;
            EXB_CHECK(hipDeviceSynchronize());
        }
        if (variant >= 0) c.variant = variant;
    });
    return 0;
}
int exblas_get_round_mode(void) { return round_mode(); }

int exblas_set_accumulator_slot(int slot)
{
    Ctx &c = ctx(-1);
    std::lock_guard<std::mutex> lk(c.mu);
    if (slot < 0 || slot > 1) return (int)hipErrorInvalidValue;
    c.slot = slot;
    c.gacc = c.gacc_all + (size_t)slot * NL * c.ngroups;
    c.gflags = (unsigned *)((char *)c.gflags_all + (size_t)FLAG_BLOCK_BYTES * slot);
    return 0;
}

int exblas_set_launch_events(void *ev_start, void *ev_stop)
{
    Ctx &c = ctx(-1);
    std::lock_guard<std::mutex> lk(c.mu);
    c.launch_start_event = (hipEvent_t)ev_start;
    c.launch_stop_event = (hipEvent_t)ev_stop;
    return 0;
}

// out[0] = implementation of the most recent exgemm on this device (0 scalar kernel, 1 fp64 slices on MFMA-F64,
// 2 int8 slices on the int8 matrix cores), out[1] / out[2] = digits (slices) of A / B, out[3..7] reserved.
// The int8 path decides on the device: this call then synchronises the device and reads the decision back.
static int last_gemm_info_on(Ctx &c, int *out)
{
    std::lock_guard<std::mutex> lk(c.mu);
    for (int i = 0; i < 8; ++i) out[i] = 0;
    if (c.gemm_info_dev) {
        int h[16];
        if (hipDeviceSynchronize() != hipSuccess) return -1;
        if (hipMemcpy(h, c.gemm_info_dev, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess) return -1;
        out[0] = (h[7] == 2 || h[7] == 4) ? h[7] : 0;  // INFO_PATH
        if (out[0] == 2) {
            out[1] = h[5];           // INFO_SA
            out[2] = h[6];           // INFO_SB
        } else if (out[0] == 4) {
            out[1] = h[11];          // INFO_CRT_NA: bits of the fixed-point entries of A'
            out[2] = h[12];          // INFO_CRT_NB
            out[3] = h[10];          // INFO_CRT_L: moduli = int8 GEMMs
            out[4] = c.gemm_ws_moduli;  // moduli the workspace was reserved for (39 unless memory ran short)
        }
    } else if (c.last_gemm_slices > 0) {
        out[0] = 1;
        out[1] = out[2] = c.last_gemm_slices;
    }
    return 0;
}

int exblas_last_gemm_info(int *out) { return last_gemm_info_on(ctx(-1, g_last_layer[current_device()]), out); }

int exblas_last_gemm_slices(void)
{
    int v[8];
    if (exblas_last_gemm_info(v) != 0) return -1;
    if (v[0] == 4) return v[3];  // residue path: the number of moduli (= int8 GEMMs)
    return v[1] > v[2] ? v[1] : v[2];
}

void exblas_set_gemm_max_slices(int s)
{
    ctx(-1);
    for_each_layer(current_device(), [&](Ctx &c) {
        std::lock_guard<std::mutex> lk(c.mu);
        c.gemm_max_slices = s;
    });
}

void exblas_set_gemm_max_moduli(int l)
{
    ctx(-1);
    for_each_layer(current_device(), [&](Ctx &c) {
        std::lock_guard<std::mutex> lk(c.mu);
        c.gemm_max_moduli = l;
    });
}

void exblas_set_gemm_path(int mode)
{
    ctx(-1);
    for_each_layer(current_device(), [&](Ctx &c) {
        std::lock_guard<std::mutex> lk(c.mu);
        c.gemm_path = mode;
    });
}

// ---- implementations on an explicit context (layer 0 for the *_dev entry points, a private one for host calls) ----
static int exsum_accumulate_on(Ctx &c, const double *d_a, int64_t n, int64_t inca, int fpe, int early_exit,
                               hipStream_t st)
{
    if (fpe < 0 || n > 0x7fffffffll) return (int)hipErrorInvalidValue;
    std::lock_guard<std::mutex> lk(c.mu);
    bool ok = true;
    // unsupported (fpe, early_exit) combination: nothing is launched, the accumulators stay zero -> 0.0
    return n > 0 ? (int)exsum_dispatch(c, d_a, n, inca, fpe, early_exit, st, &ok) : 0;
}

static int exdot_accumulate_on(Ctx &c, const double *d_a, int64_t inca, const double *d_b, int64_t incb, int64_t n,
                               int fpe, int early_exit, hipStream_t st)
{
    if (fpe < 0 || n > 0x7fffffffll) return (int)hipErrorInvalidValue;
    std::lock_guard<std::mutex> lk(c.mu);
    bool ok = true;
    return n > 0 ? (int)exdot_dispatch(c, d_a, inca, d_b, incb, n, fpe, early_exit, st, &ok) : 0;
}

static int finish_on(Ctx &c, hipStream_t st, int64_t *d_out)
{
    std::lock_guard<std::mutex> lk(c.mu);
    return (int)finalize_groups(c, st, (long long *)d_out);
}

static int exgemv_on(Ctx &c, char transa, int m, int n, double alpha, const double *d_a, int lda, const double *d_x,
                     int incx, double beta, double *d_y, int incy, int fpe, int early_exit, hipStream_t st)
{
    if (fpe < 0) return (int)hipErrorInvalidValue;
    std::lock_guard<std::mutex> lk(c.mu);
    if (c.layer < MAX_LAYERS) g_last_layer[c.device] = c.layer;
    return (int)exgemv_dispatch(c, transa, m, n, alpha, d_a, lda, d_x, incx, beta, d_y, incy, fpe, early_exit,
                                round_mode(), st);
}

static int extrsv_on(Ctx &c, char uplo, char transa, char diag, int n, const double *d_a, int lda, double *d_x,
                     int incx, int fpe, int early_exit, hipStream_t st)
{
    if (fpe < 0) return (int)hipErrorInvalidValue;
    if (fpe >= 9) return EXBLAS_UNSUPPORTED;
    if (n > 0 && (lda < n || incx <= 0)) return (int)hipErrorInvalidValue;
    std::lock_guard<std::mutex> lk(c.mu);
    if (c.layer < MAX_LAYERS) g_last_layer[c.device] = c.layer;
    return (int)extrsv_dispatch(c, uplo, transa, diag, n, d_a, lda, d_x, incx, fpe, early_exit, round_mode(), st);
}

static int exgemm_on(Ctx &c, char transa, char transb, int m, int n, int k, double alpha, const double *d_a, int lda,
                     const double *d_b, int ldb, double beta, double *d_c, int ldc, int fpe, int early_exit,
                     hipStream_t st, const GemmChunks *chunks = nullptr)
{
    if (fpe < 0) return (int)hipErrorInvalidValue;
    std::lock_guard<std::mutex> lk(c.mu);
    if (c.layer < MAX_LAYERS) g_last_layer[c.device] = c.layer;
    return (int)exgemm_dispatch(c, transa, transb, m, n, k, alpha, d_a, lda, d_b, ldb, beta, d_c, ldc, fpe,
                                early_exit, round_mode(), st, chunks);
}

int exblas_exsum_accumulate_dev(const double *d_a, int64_t n, int64_t inca, int fpe, int early_exit, void *stream)
{
    return exsum_accumulate_on(ctx(-1), d_a, n, inca, fpe, early_exit, (hipStream_t)stream);
}

int exblas_exdot_accumulate_dev(const double *d_a, int64_t inca, const double *d_b, int64_t incb, int64_t n, int fpe,
                                int early_exit, void *stream)
{
    return exdot_accumulate_on(ctx(-1), d_a, inca, d_b, incb, n, fpe, early_exit, (hipStream_t)stream);
}

int exblas_finish_dev(void *stream, int64_t *d_out) { return finish_on(ctx(-1), (hipStream_t)stream, d_out); }

int exblas_exsum_dev(const double *d_a, int64_t n, int64_t inca, int fpe, int early_exit, void *stream,
                     int64_t *d_out)
{
    int rc = exblas_exsum_accumulate_dev(d_a, n, inca, fpe, early_exit, stream);
    return rc ? rc : exblas_finish_dev(stream, d_out);
}

int exblas_exdot_dev(const double *d_a, int64_t inca, const double *d_b, int64_t incb, int64_t n, int fpe,
                     int early_exit, void *stream, int64_t *d_out)
{
    int rc = exblas_exdot_accumulate_dev(d_a, inca, d_b, incb, n, fpe, early_exit, stream);
    return rc ? rc : exblas_finish_dev(stream, d_out);
}

int exblas_exsum_segmented_dev(const double *d_values, const int64_t *d_offsets, int64_t nseg, int fpe, int early_exit,
                               void *stream, double *d_out)
{
    if (fpe < 0) return (int)hipErrorInvalidValue;
    ctx(-1);
    return (int)exsum_segmented_dispatch(d_values, (const long long *)d_offsets, nseg, fpe, early_exit, round_mode(),
                                         (hipStream_t)stream, d_out);
}

int exblas_finalize_dev(const int64_t *d_digit_sets, int nsets, uint32_t flags_or, void *stream, int64_t *d_out)
{
    ctx(-1);
    return (int)finalize_sets((const long long *)d_digit_sets, nsets, flags_or, (hipStream_t)stream,
                              (long long *)d_out);
}

int exblas_exgemv_dev(char transa, int m, int n, double alpha, const double *d_a, int lda, const double *d_x,
                      int incx, double beta, double *d_y, int incy, int fpe, int early_exit, void *stream)
{
    return exgemv_on(ctx(-1), transa, m, n, alpha, d_a, lda, d_x, incx, beta, d_y, incy, fpe, early_exit,
                     (hipStream_t)stream);
}

int exblas_extrsv_dev(char uplo, char transa, char diag, int n, const double *d_a, int lda, double *d_x, int incx,
                      int fpe, int early_exit, void *stream)
{
    return extrsv_on(ctx(-1), uplo, transa, diag, n, d_a, lda, d_x, incx, fpe, early_exit, (hipStream_t)stream);
}

int exblas_extrsv_last_slow_rows(void)
{
    Ctx &c = ctx(-1, g_last_layer[current_device()]);
    std::lock_guard<std::mutex> lk(c.mu);
    int v[4] = {0, 0, 0, 0};
    if (!c.ws || c.ws_bytes < sizeof(v)) return -1;
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    if (hipMemcpy(v, c.ws, sizeof(v), hipMemcpyDeviceToHost) != hipSuccess) return -1;
    return v[2];
}

int exblas_exgemm_dev(char transa, char transb, int m, int n, int k, double alpha, const double *d_a, int lda,
                      const double *d_b, int ldb, double beta, double *d_c, int ldc, int fpe, int early_exit,
                      void *stream)
{
    return exgemm_on(ctx(-1), transa, transb, m, n, k, alpha, d_a, lda, d_b, ldb, beta, d_c, ldc, fpe, early_exit,
                     (hipStream_t)stream);
}

}  // extern "C"
int exb::exgemm_chunked_dev(char transa, char transb, int m, int n, int k, double alpha, const double *d_a, int lda,
                            const double *d_b, int ldb, double beta, double *d_c, int ldc, int fpe, int early_exit,
                            hipStream_t st, const GemmChunks *chunks)
{
    return exgemm_on(ctx(-1), transa, transb, m, n, k, alpha, d_a, lda, d_b, ldb, beta, d_c, ldc, fpe, early_exit, st,
                     chunks);
}
extern "C" {

// ---- context handles: independent accumulators, flags and workspace per caller-owned handle --------------------
}  // extern "C"
struct exblas_ctx {
    exb::Ctx c;
};
// the handle's context; the calling thread must be on the handle's device (like any stream or buffer of that device)
static Ctx *handle_ctx(exblas_ctx *h)
{
    if (!h) return &ctx(-1);
    return current_device() == h->c.device ? &h->c : nullptr;
}
extern "C" {

int exblas_ctx_create(exblas_ctx_t **out)
{
    if (!out) return (int)hipErrorInvalidValue;
    ctx(-1);  // the device's default context first: a handle inherits the knobs set through the API
    exblas_ctx *h = new exblas_ctx;
    {
        std::lock_guard<std::mutex> lk(g_ctx_mu);
        init_ctx(h->c, current_device(), MAX_LAYERS);  // layer >= 1: inherits; >= MAX_LAYERS: not one of the static ones
    }
    *out = h;
    return 0;
}

int exblas_ctx_destroy(exblas_ctx_t *h)
{
    if (!h) return 0;
    Ctx &c = h->c;
    int prev = 0;
    (void)hipGetDevice(&prev);
    hipError_t first = hipSetDevice(c.device);
    hipError_t e = hipDeviceSynchronize();  // work enqueued on the handle may still be running
    if (first == hipSuccess) first = e;
    auto rel = [&](void *p) {
        if (!p) return;
        hipError_t e2 = hipFree(p);
        if (first == hipSuccess) first = e2;
    };
    rel(c.gacc_all);
    rel(c.gflags_all);
    rel(c.d_record);
    for (void *p : c.stage) rel(p);
    for (void *p : c.retired) rel(p);
    rel(c.ws);
    if (c.h_record) (void)hipHostFree(c.h_record);
    if (c.stream) (void)hipStreamDestroy(c.stream);
    (void)hipSetDevice(prev);
    delete h;
    return (int)first;
}

#define EXB_HANDLE(h)              \
    Ctx *cp = handle_ctx(h);       \
    if (!cp) return (int)hipErrorInvalidDevice

int exblas_exsum_accumulate_ctx(exblas_ctx_t *h, const double *d_a, int64_t n, int64_t inca, int fpe, int early_exit,
                                void *stream)
{
    EXB_HANDLE(h);
    return exsum_accumulate_on(*cp, d_a, n, inca, fpe, early_exit, (hipStream_t)stream);
}

int exblas_exdot_accumulate_ctx(exblas_ctx_t *h, const double *d_a, int64_t inca, const double *d_b, int64_t incb,
                                int64_t n, int fpe, int early_exit, void *stream)
{
    EXB_HANDLE(h);
    return exdot_accumulate_on(*cp, d_a, inca, d_b, incb, n, fpe, early_exit, (hipStream_t)stream);
}

int exblas_finish_ctx(exblas_ctx_t *h, void *stream, int64_t *d_out)
{
    EXB_HANDLE(h);
    return finish_on(*cp, (hipStream_t)stream, d_out);
}

int exblas_exsum_ctx(exblas_ctx_t *h, const double *d_a, int64_t n, int64_t inca, int fpe, int early_exit, void *stream,
                     int64_t *d_out)
{
    EXB_HANDLE(h);
    int rc = exsum_accumulate_on(*cp, d_a, n, inca, fpe, early_exit, (hipStream_t)stream);
    return rc ? rc : finish_on(*cp, (hipStream_t)stream, d_out);
}

int exblas_exdot_ctx(exblas_ctx_t *h, const double *d_a, int64_t inca, const double *d_b, int64_t incb, int64_t n, int fpe,
                     int early_exit, void *stream, int64_t *d_out)
{
    EXB_HANDLE(h);
    int rc = exdot_accumulate_on(*cp, d_a, inca, d_b, incb, n, fpe, early_exit, (hipStream_t)stream);
    return rc ? rc : finish_on(*cp, (hipStream_t)stream, d_out);
}

int exblas_exgemv_ctx(exblas_ctx_t *h, char transa, int m, int n, double alpha, const double *d_a, int lda,
                      const double *d_x, int incx, double beta, double *d_y, int incy, int fpe, int early_exit,
                      void *stream)
{
    EXB_HANDLE(h);
    return exgemv_on(*cp, transa, m, n, alpha, d_a, lda, d_x, incx, beta, d_y, incy, fpe, early_exit, (hipStream_t)stream);
}

int exblas_extrsv_ctx(exblas_ctx_t *h, char uplo, char transa, char diag, int n, const double *d_a, int lda, double *d_x,
                      int incx, int fpe, int early_exit, void *stream)
{
    EXB_HANDLE(h);
    return extrsv_on(*cp, uplo, transa, diag, n, d_a, lda, d_x, incx, fpe, early_exit, (hipStream_t)stream);
}

int exblas_exgemm_ctx(exblas_ctx_t *h, char transa, char transb, int m, int n, int k, double alpha, const double *d_a,
                      int lda, const double *d_b, int ldb, double beta, double *d_c, int ldc, int fpe, int early_exit,
                      void *stream)
{
    EXB_HANDLE(h);
    return exgemm_on(*cp, transa, transb, m, n, k, alpha, d_a, lda, d_b, ldb, beta, d_c, ldc, fpe, early_exit,
                     (hipStream_t)stream);
}

int exblas_reserve_workspace_ctx(exblas_ctx_t *h, size_t bytes)
{
    EXB_HANDLE(h);
    std::lock_guard<std::mutex> lk(cp->mu);
    hipError_t e = hipSuccess;
    workspace(*cp, bytes, nullptr, &e);
    return (int)e;
}

size_t exblas_workspace_bytes_ctx(exblas_ctx_t *h)
{
    Ctx *cp = handle_ctx(h);
    if (!cp) return 0;
    std::lock_guard<std::mutex> lk(cp->mu);
    return cp->ws_bytes;
}

int exblas_last_gemm_info_ctx(exblas_ctx_t *h, int *out8)
{
    EXB_HANDLE(h);
    return last_gemm_info_on(*cp, out8);
}
#undef EXB_HANDLE

int exblas_reserve_workspace(size_t bytes)
{
    Ctx &c = ctx(-1);
    std::lock_guard<std::mutex> lk(c.mu);
    hipError_t e = hipSuccess;
    workspace(c, bytes, nullptr, &e);
    return (int)e;
}

size_t exblas_workspace_bytes(void)
{
    Ctx &c = ctx(-1);
    std::lock_guard<std::mutex> lk(c.mu);
    return c.ws_bytes;
}

int exblas_release_retired_workspaces(void)
{
    ctx(-1);
    hipError_t first = hipDeviceSynchronize();
    for_each_layer(current_device(), [&](Ctx &c) {
        std::lock_guard<std::mutex> lk(c.mu);
        for (void *p : c.retired) {
            hipError_t e = hipFree(p);
            if (first == hipSuccess) first = e;
        }
        c.retired.clear();
    });
    return (int)first;
}

int exblas_release_workspace(void)
{
    ctx(-1);
    hipError_t first = hipDeviceSynchronize();
    for_each_layer(current_device(), [&](Ctx &c) {
        std::lock_guard<std::mutex> lk(c.mu);
        for (void *p : c.retired) {
            hipError_t e = hipFree(p);
            if (first == hipSuccess) first = e;
        }
        c.retired.clear();
        if (c.ws) {
            hipError_t e = hipFree(c.ws);
            if (first == hipSuccess) first = e;
        }
        c.ws = nullptr;
        c.ws_bytes = 0;
        c.gemm_info_dev = nullptr;  // it pointed into the workspace
    });
    return (int)first;
}

int exblas_gen_dev(int kind, uint64_t seed, int64_t first, int64_t count, int64_t n_total, double p0, double p1,
                   double *d_out, void *stream)
{
    Ctx &c = ctx(-1);
    if (count <= 0) return 0;
    int i0 = 0, i1 = 0;
    double dscale = 0.0;
    switch (kind) {
    case EXBLAS_GEN_FPUNIFORM:
    case EXBLAS_GEN_FPUNIFORM_SIGNED: i0 = (int)p0; i1 = (int)p1; break;
    case EXBLAS_GEN_LOGNORMAL:
        // same expressions as orc_gen_one, evaluated on the host in IEEE double
        dscale = p1 * (1.0 / (0.6931471805599453 * 37837.22690659431));
        i0 = (int)rint(p0 * (1.0 / 0.6931471805599453));
        break;
    case EXBLAS_GEN_ILLCOND: i0 = (int)rint(log2(p0) * 0.5); break;
    case EXBLAS_GEN_CANCEL: i0 = (int)p0; break;
    default: break;
    }
    long long blocks = (count + 255) / 256;
    long long cap = (long long)c.num_cu * 16;
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL(k_gen, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, kind,
                       (unsigned long long)seed, (long long)first, (long long)count, (long long)n_total, i0, i1,
                       dscale, d_out);
    return (int)hipGetLastError();
}

int exblas_stream_read_dev(const double *d_a, int64_t n, void *stream, double *d_sink)
{
    Ctx &c = ctx(-1);
    long long blocks = (long long)c.num_cu * c.blocks_per_cu;
    hipLaunchKernelGGL(k_stream_read, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, d_a, (long long)n,
                       d_sink);
    return (int)hipGetLastError();
}

int exblas_stream_read2_dev(const double *d_a, const double *d_b, int64_t n, int blocks_per_cu, void *stream,
                            double *d_sink)
{
    Ctx &c = ctx(-1);
    long long blocks = (long long)c.num_cu * (blocks_per_cu > 0 ? blocks_per_cu : c.bpc_dot);
    hipLaunchKernelGGL(k_stream_read2, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, d_a, d_b, (long long)n,
                       d_sink);
    return (int)hipGetLastError();
}

// ---- host-pointer layer ---------------------------------------------------------------------

static void check_fpe(int fpe)
{
    if (fpe < 0) {
        // cpu:ExSUM.cpp:25-28
        fprintf(stderr, "Size of floating-point expansion should be a positive number. Preferably, it should be "
                        "in the interval [2, 8]\n");
        exit(1);
    }
}

// ---- ExSUM / ExDOT of host vectors ---------------------------------------------------------------------------
// The reference's GPU backend copies the whole vector to ONE device per call (gpu:ExSUM.cpp:126), its CPU backend
// scatters slices from rank 0 to the other ranks (cpu:ExSUM.cpp:33-63).  Here one call spreads the vector over the
// GPUs of the node: every "virtual device" (a GPU, with a private host-layer context) gets a contiguous part, streams
// it through its own PCIe link in 64 MiB chunks (copy, accumulate kernel, copy, ...: bounded staging memory, the
// kernels hide behind the copies) and normalises its sum; the parts' 576-byte digit sets are then added and rounded
// once on the first device -- so the bits do not depend on how many GPUs took part.
static std::mutex g_host_mu;           // host-pointer calls are serialised process-wide (the reference's are not re-entrant at all)
static std::vector<int> g_host_devs;   // set by exblas_set_host_devices / EXBLAS_HOST_DEVICES; empty = default policy
static bool g_host_devs_env_read = false;
constexpr long long HOST_CHUNK_ELEMS = 8ll << 20;          // 64 MiB of doubles per copy
constexpr long long HOST_SPLIT_MIN_BYTES = 256ll << 20;    // smaller inputs stay on the current device

static std::vector<int> host_devices(long long bytes)
{
    if (!g_host_devs_env_read) {
        g_host_devs_env_read = true;
        const char *e = getenv("EXBLAS_HOST_DEVICES");  // "all", "current", or a list such as "0,1,2" / "0,0"
        if (e && *e && g_host_devs.empty()) {
            int ndev = exblas_hip_device_count();
            if (!strcmp(e, "all")) {
                for (int d = 0; d < ndev && d < MAX_LAYERS - 1; ++d) g_host_devs.push_back(d);
            } else if (strcmp(e, "current")) {
                for (const char *q = e; *q;) {
                    char *end;
                    long d = strtol(q, &end, 10);
                    if (end == q) break;
                    if (d >= 0 && d < ndev && (int)g_host_devs.size() < MAX_LAYERS - 1) g_host_devs.push_back((int)d);
                    q = *end ? end + 1 : end;
                }
            } else {
                g_host_devs.push_back(-1);  // current device only
            }
        }
    }
    std::vector<int> devs = g_host_devs;
    if (devs.empty()) {
        // default: one big call uses every GPU this process can see (8 PCIe links instead of 1); small ones do not
        // pay for waking the other devices.  NOT in a one-process-per-GPU job (a launcher's rank variables in the
        // environment, or a communicator created through exblas_comm_*): every rank sees all devices there, and a rank
        // that spread its host calls would create contexts and push PCIe traffic on its peers' GPUs -- the reference's
        // GPU backend uses exactly one device per call (gpu:ExSUM.cpp:86-126).
        static const bool rank_env = [] {
            for (const char *v : {"WORLD_SIZE", "OMPI_COMM_WORLD_SIZE", "PMI_SIZE", "SLURM_NTASKS"}) {
                const char *e = getenv(v);
                if (e && atoi(e) > 1) return true;
            }
            return false;
        }();
        const int ndev = exblas_hip_device_count();
        if (bytes >= HOST_SPLIT_MIN_BYTES && ndev > 1 && !rank_env && !exb::g_comm_created.load())
            for (int d = 0; d < ndev && d < MAX_LAYERS - 1; ++d) devs.push_back(d);
        else
            devs.push_back(-1);
    }
    const int cur = current_device();
    for (int &d : devs)
        if (d < 0) d = cur;
    return devs;
}

struct HostPart {
    int dev = 0, layer = 1;
    long long i0 = 0, i1 = 0;
    int rc = 0;
    const char *what = "";
    bool want_ext = false;          // exdot over several parts: export the low / high digit sets instead of folding them
    std::vector<long long> ext;     // ... and bring them to the host (EXT_WORDS)
};

// one part of the vector(s) on one virtual device; leaves the part's record in the context's pinned h_record
static void host_reduce_part(HostPart &p, const double *a, long long inca, const double *b, long long incb, int fpe,
                             int early_exit)
{
    hipError_t e = hipSetDevice(p.dev);
    if (e != hipSuccess) { p.rc = (int)e; p.what = "hipSetDevice"; return; }
    Ctx &c = ctx(p.dev, p.layer);
    for (long long i = p.i0; i < p.i1 && !p.rc; i += HOST_CHUNK_ELEMS) {
        const long long cnt = std::min(HOST_CHUNK_ELEMS, p.i1 - i);
        const size_t spa = ((size_t)(cnt - 1) * (size_t)inca + 1) * sizeof(double);
        double *da, *db = nullptr;
        {
            std::lock_guard<std::mutex> lk(c.mu);
            da = (double *)stage_buf(c, 0, spa);
            if (b) db = (double *)stage_buf(c, 1, ((size_t)(cnt - 1) * (size_t)incb + 1) * sizeof(double));
        }
        // element count semantics of the GPU backend: a[offset + i*inca] (ExSUM.Superacc.cl:249-250); the touched
        // span of the chunk is copied like gpu:ExSUM.cpp:126 copies the whole vector
        e = hipMemcpyAsync(da, a + i * inca, spa, hipMemcpyHostToDevice, c.stream);
        if (e == hipSuccess && b)
            e = hipMemcpyAsync(db, b + i * incb, ((size_t)(cnt - 1) * (size_t)incb + 1) * sizeof(double),
                               hipMemcpyHostToDevice, c.stream);
        if (e != hipSuccess) { p.rc = (int)e; p.what = "hipMemcpyAsync"; return; }
        p.rc = b ? exdot_accumulate_on(c, da, inca, db, incb, cnt, fpe, early_exit, c.stream)
                 : exsum_accumulate_on(c, da, cnt, inca, fpe, early_exit, c.stream);
        p.what = "accumulate";
    }
    if (p.rc) return;
    long long *d_ext = nullptr;
    if (p.want_ext) {
        std::lock_guard<std::mutex> lk(c.mu);
        d_ext = (long long *)stage_buf(c, 2, sizeof(long long) * EXT_WORDS);
        p.rc = (int)finalize_groups(c, c.stream, c.d_record, d_ext);
    } else {
        p.rc = finish_on(c, c.stream, (int64_t *)c.d_record);
    }
    p.what = "finish";
    if (p.rc) return;
    e = hipMemcpyAsync(c.h_record, c.d_record, sizeof(long long) * OUT_WORDS, hipMemcpyDeviceToHost, c.stream);
    if (e == hipSuccess && d_ext) {
        p.ext.resize(EXT_WORDS);
        e = hipMemcpyAsync(p.ext.data(), d_ext, sizeof(long long) * EXT_WORDS, hipMemcpyDeviceToHost, c.stream);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(c.stream);
    if (e != hipSuccess) { p.rc = (int)e; p.what = "record copy"; }
}

static int host_reduce(long long n, const double *a, long long inca, const double *b, long long incb, int fpe,
                       int early_exit, int64_t *out_words)
{
    std::lock_guard<std::mutex> host_lock(g_host_mu);
    const int home = current_device();
    if (n < 0) n = 0;
    std::vector<int> devs = host_devices(n * (long long)sizeof(double) * (b ? 2 : 1));
    const int nv = (int)std::min<long long>((long long)devs.size(), std::max<long long>(1, n / 2));
    std::vector<HostPart> parts(nv);
    for (int v = 0; v < nv; ++v) {
        parts[v].dev = devs[v];
        parts[v].layer = 1 + v;
        parts[v].i0 = (n * v) / nv;
        parts[v].i1 = (n * (v + 1)) / nv;
        parts[v].want_ext = nv > 1 && b != nullptr;
    }
    if (nv == 1) {
        host_reduce_part(parts[0], a, inca, b, incb, fpe, early_exit);
    } else {
        std::vector<std::thread> th;
        for (int v = 0; v < nv; ++v)
            th.emplace_back([&, v] { host_reduce_part(parts[v], a, inca, b, incb, fpe, early_exit); });
        for (auto &t : th) t.join();
    }
    EXB_CHECK(hipSetDevice(home));
    for (auto &p : parts)
        if (p.rc) die(p.what, (hipError_t)p.rc, __FILE__, __LINE__);
    Ctx &c0 = ctx(parts[0].dev, parts[0].layer);
    if (nv > 1) {
        // add the parts' digit sets and round once, on the first device (exblas_finalize_dev's job in the *_dev layer)
        EXB_CHECK(hipSetDevice(c0.device));
        long long *d_sets;
        {
            std::lock_guard<std::mutex> lk(c0.mu);
            d_sets = (long long *)stage_buf(c0, 2, sizeof(long long) * (SET_WORDS * nv + EXT_WORDS));
        }
        for (int v = 0; v < nv; ++v) {
            Ctx &cv = ctx(parts[v].dev, parts[v].layer);
            EXB_CHECK(hipMemcpyAsync(d_sets + (size_t)v * SET_WORDS, cv.h_record + OUT_DIGITS,
                                     sizeof(long long) * SET_WORDS, hipMemcpyHostToDevice, c0.stream));
        }
        // exdot: the parts exported their low / high digit sets (products outside the double range, superacc.hip.h)
        // instead of folding them; their sums are folded once, like the multi-rank path does after its all-reduce
        long long *d_ext = nullptr;
        std::vector<long long> ext_sum;
        if (parts[0].want_ext) {
            ext_sum.assign(EXT_WORDS, 0);
            for (auto &p : parts)
                for (int i = 0; i < EXT_WORDS; ++i) ext_sum[i] += p.ext[i];
            d_ext = d_sets + (size_t)SET_WORDS * nv;
            EXB_CHECK(hipMemcpyAsync(d_ext, ext_sum.data(), sizeof(long long) * EXT_WORDS, hipMemcpyHostToDevice, c0.stream));
        }
        EXB_CHECK(finalize_sets(d_sets, nv, 0u, c0.stream, c0.d_record, d_ext));
        EXB_CHECK(hipMemcpyAsync(c0.h_record, c0.d_record, sizeof(long long) * OUT_WORDS, hipMemcpyDeviceToHost,
                                 c0.stream));
        EXB_CHECK(hipStreamSynchronize(c0.stream));
        EXB_CHECK(hipSetDevice(home));
    }
    memcpy(out_words, c0.h_record, sizeof(long long) * OUT_WORDS);
    return nv;
}

int exblas_set_host_devices(int count, const int *devices)
{
    std::lock_guard<std::mutex> host_lock(g_host_mu);
    const int ndev = exblas_hip_device_count();
    if (count < 0 || count > MAX_LAYERS - 1) return (int)hipErrorInvalidValue;
    for (int i = 0; i < count; ++i)
        if (devices[i] < 0 || devices[i] >= ndev) return (int)hipErrorInvalidDevice;
    g_host_devs.assign(devices, devices + count);  // count == 0: back to the default policy
    g_host_devs_env_read = true;
    return 0;
}

int exblas_exsum_record(int Ng, const double *ag, int inca, int offset, int fpe, int early_exit,
                        int64_t *out_words)
{
    check_fpe(fpe);
    host_reduce(Ng > 0 ? Ng : 0, ag + offset, inca > 0 ? inca : 1, nullptr, 1, fpe, early_exit, out_words);
    return 0;
}

int exblas_exdot_record(int Ng, const double *ag, int inca, int offseta, const double *bg, int incb, int offsetb,
                        int fpe, int early_exit, int64_t *out_words)
{
    check_fpe(fpe);
    host_reduce(Ng > 0 ? Ng : 0, ag + offseta, inca > 0 ? inca : 1, bg + offsetb, incb > 0 ? incb : 1, fpe, early_exit,
                out_words);
    return 0;
}

static double record_value(const int64_t *rec)
{
    double d;
    memcpy(&d, &rec[round_mode() ? EXBLAS_OUT_REFMODE : EXBLAS_OUT_EXACT], sizeof(d));
    return d;
}

double exblas_exsum(int Ng, const double *ag, int inca, int offset, int fpe, int early_exit)
{
    int64_t rec[EXBLAS_OUT_WORDS];
    exblas_exsum_record(Ng, ag, inca, offset, fpe, early_exit, rec);
    return record_value(rec);
}

double exblas_exdot(int Ng, const double *ag, int inca, int offseta, const double *bg, int incb, int offsetb,
                    int fpe, int early_exit)
{
    if (Ng <= 0) return 0.0;  // ExDOT.cpp:70-71
    int64_t rec[EXBLAS_OUT_WORDS];
    exblas_exdot_record(Ng, ag, inca, offseta, bg, incb, offsetb, fpe, early_exit, rec);
    return record_value(rec);
}

int exblas_exgemv(char transa, int m, int n, double alpha, const double *a, int lda, int offseta, const double *x,
                  int incx, int offsetx, double beta, double *y, int incy, int offsety, int fpe, int early_exit)
{
    check_fpe(fpe);
    if (m <= 0 || n <= 0) return 0;
    Ctx &c = ctx(-1, 1);
    std::lock_guard<std::mutex> api_lock(g_host_mu);
    const bool trans = (transa == 'T' || transa == 't');
    const int rows = trans ? n : m, inner = trans ? m : n;
    double *d_a, *d_x, *d_y;
    size_t abytes = (size_t)lda * (size_t)n * sizeof(double);  // column-major: n columns of lda
    size_t xspan = (size_t)(inner - 1) * (size_t)incx + 1, yspan = (size_t)(rows - 1) * (size_t)incy + 1;
    {
        std::lock_guard<std::mutex> lk(c.mu);
        d_a = (double *)stage_buf(c, 0, abytes);
        d_x = (double *)stage_buf(c, 1, xspan * sizeof(double));
        d_y = (double *)stage_buf(c, 2, yspan * sizeof(double));
        EXB_CHECK(hipMemcpyAsync(d_a, a + offseta, abytes - (size_t)(lda - m) * sizeof(double), hipMemcpyHostToDevice,
                                 c.stream));
        EXB_CHECK(hipMemcpyAsync(d_x, x + offsetx, xspan * sizeof(double), hipMemcpyHostToDevice, c.stream));
        EXB_CHECK(hipMemcpyAsync(d_y, y + offsety, yspan * sizeof(double), hipMemcpyHostToDevice, c.stream));
    }
    int rc = exgemv_on(c, transa, m, n, alpha, d_a, lda, d_x, incx, beta, d_y, incy, fpe, early_exit, c.stream);
    if (rc) die("exblas_exgemv", (hipError_t)rc, __FILE__, __LINE__);
    EXB_CHECK(hipMemcpyAsync(y + offsety, d_y, yspan * sizeof(double), hipMemcpyDeviceToHost, c.stream));
    EXB_CHECK(hipStreamSynchronize(c.stream));
    return 0;
}

int exblas_extrsv(char uplo, char transa, char diag, int n, const double *a, int lda, int offseta, double *x,
                  int incx, int offsetx, int fpe, int early_exit)
{
    check_fpe(fpe);
    if (fpe >= 9) {
        fprintf(stderr, "exblas(hip): extrsv fpe = %d selects an iterative-refinement kernel the reference does not "
                        "ship (ExTRSV.cpp:91-120); nothing done\n", fpe);
        return -1;
    }
    if (n <= 0) return 0;
    Ctx &c = ctx(-1, 1);
    std::lock_guard<std::mutex> api_lock(g_host_mu);
    double *d_a, *d_x;
    const size_t abytes = ((size_t)lda * (size_t)(n - 1) + (size_t)n) * sizeof(double);  // n columns of lda
    const size_t xspan = (size_t)(n - 1) * (size_t)incx + 1;
    {
        std::lock_guard<std::mutex> lk(c.mu);
        d_a = (double *)stage_buf(c, 0, abytes);
        d_x = (double *)stage_buf(c, 1, xspan * sizeof(double));
        EXB_CHECK(hipMemcpyAsync(d_a, a + offseta, abytes, hipMemcpyHostToDevice, c.stream));
        EXB_CHECK(hipMemcpyAsync(d_x, x + offsetx, xspan * sizeof(double), hipMemcpyHostToDevice, c.stream));
    }
    int rc = extrsv_on(c, uplo, transa, diag, n, d_a, lda, d_x, incx, fpe, early_exit, c.stream);
    if (rc) die("exblas_extrsv", (hipError_t)rc, __FILE__, __LINE__);
    EXB_CHECK(hipMemcpyAsync(x + offsetx, d_x, xspan * sizeof(double), hipMemcpyDeviceToHost, c.stream));
    EXB_CHECK(hipStreamSynchronize(c.stream));
    return 0;
}

int exblas_exgemm(char transa, char transb, int m, int n, int k, double alpha, const double *a, int lda,
                  const double *b, int ldb, double beta, double *cm, int ldc, int fpe, int early_exit)
{
    check_fpe(fpe);
    if (m <= 0 || n <= 0) return 0;
    Ctx &c = ctx(-1, 1);
    std::lock_guard<std::mutex> api_lock(g_host_mu);
    const bool ta = (transa == 'T' || transa == 't'), tb = (transb == 'T' || transb == 't');
    // row-major storage (ExGEMM.Superacc.cl:254-255): A is m x k (k x m when transposed), etc.
    size_t abytes = (size_t)(ta ? k : m) * (size_t)lda * sizeof(double);
    size_t bbytes = (size_t)(tb ? n : k) * (size_t)ldb * sizeof(double);
    size_t cbytes = (size_t)m * (size_t)ldc * sizeof(double);
    double *d_a, *d_b, *d_c;
    {
        std::lock_guard<std::mutex> lk(c.mu);
        d_a = (double *)stage_buf(c, 0, abytes);
        d_b = (double *)stage_buf(c, 1, bbytes);
        d_c = (double *)stage_buf(c, 2, cbytes);
        EXB_CHECK(hipMemcpyAsync(d_a, a, abytes, hipMemcpyHostToDevice, c.stream));
        EXB_CHECK(hipMemcpyAsync(d_b, b, bbytes, hipMemcpyHostToDevice, c.stream));
        EXB_CHECK(hipMemcpyAsync(d_c, cm, cbytes, hipMemcpyHostToDevice, c.stream));
    }
    int rc = exgemm_on(c, transa, transb, m, n, k, alpha, d_a, lda, d_b, ldb, beta, d_c, ldc, fpe, early_exit, c.stream);
    if (rc) die("exblas_exgemm", (hipError_t)rc, __FILE__, __LINE__);
    EXB_CHECK(hipMemcpyAsync(cm, d_c, cbytes, hipMemcpyDeviceToHost, c.stream));
    EXB_CHECK(hipStreamSynchronize(c.stream));
    return 0;
}

}  // extern "C"

// =============================================================================================
// C++ API with the reference's signatures (global namespace, C++ linkage)
// =============================================================================================
double exsum(const int Ng, double *ag, const int inca, const int offset, const int fpe, const bool early_exit,
             const bool parallel)
{
    (void)parallel;  // "Does not affect GPU implementation since it is always parallel" (gpu:ExSUM.cpp:61)
    return exblas_exsum(Ng, ag, inca, offset, fpe, early_exit ? 1 : 0);
}

double exdot(const int Ng, double *ag, const int inca, const int offseta, double *bg, const int incb,
             const int offsetb, const int fpe, const bool early_exit)
{
    return exblas_exdot(Ng, ag, inca, offseta, bg, incb, offsetb, fpe, early_exit ? 1 : 0);
}

int exgemv(const char transa, const int m, const int n, const double alpha, double *a, const int lda,
           const int offseta, double *x, const int incx, const int offsetx, const double beta, double *y,
           const int incy, const int offsety, const int fpe, const bool early_exit)
{
    return exblas_exgemv(transa, m, n, alpha, a, lda, offseta, x, incx, offsetx, beta, y, incy, offsety, fpe,
                         early_exit ? 1 : 0);
}

int extrsv(const char uplo, const char transa, const char diag, const int n, double *a, const int lda,
           const int offseta, double *x, const int incx, const int offsetx, const int fpe, const bool early_exit)
{
    return exblas_extrsv(uplo, transa, diag, n, a, lda, offseta, x, incx, offsetx, fpe, early_exit ? 1 : 0);
}

int exgemm(char transa, char transb, int m, int n, int k, double alpha, double *a, int lda, double *b, int ldb,
           double beta, double *c, int ldc, int fpe, bool early_exit)
{
    return exblas_exgemm(transa, transb, m, n, k, alpha, a, lda, b, ldb, beta, c, ldc, fpe, early_exit ? 1 : 0);
}
