// exblas_internal.h -- per-device context and cross-file declarations of libexblas.so
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <mutex>
#include <vector>

namespace exb {

// Lazily created, one per device.  Replaces the file-static kernel/buffer globals of the reference
// launchers (ExSUM.Launcher.cpp:16-36), which make the reference GPU library non re-entrant, and the
// per-call OpenCL context + JIT (gpu:ExSUM.cpp:86-209).
struct Ctx {
    int device = -1;
    int num_cu = 256;
    int blocks_per_cu = 8;   // EXBLAS_BLOCKS_PER_CU: generic cap of resident blocks per CU
    // Measured on MI355X (tools/tune.py, n = 2^28): the one-stream ExSUM kernel is fastest with FEW fat
    // blocks (2 per CU: 7.15 TB/s vs 6.46 at 8), the two-stream ExDOT kernel with many (16-32 per CU).
    int bpc_sum = 2, bpc_dot = 48;   // ExDOT: 48 (an odd grid of 12289 workgroups) edges out 32 by ~1 %, 16 and 96 lose 2-3 %
    int bpc_sa = 3;          // superaccumulator-only ExSUM (LDS-atomic bound; 3/CU: 6.65 TB/s, 2/CU: 6.0)
    int bpc_heavy = 4;       // ExSUM variants without early exit, N >= 5, in -DEXBLAS_FULL_CASCADE=1 builds only (VALU-latency-bound)
    int ngroups = 32;        // EXBLAS_NGROUPS: global group accumulators the blocks add into
    int grid_adj = 0;        // EXBLAS_GRID_ADJ: workgroups added to the grid of the streaming ExSUM / ExDOT kernels
    int variant = 0;         // tuning variant of the production kernels (exblas_set_tuning)
    // when set, the NEXT streaming ExSUM / ExDOT launch carries this event as the completion signal of its own dispatch
    // packet (hipExtLaunchKernelGGL) and clears the field: no separate event packet follows the kernel in the queue
    // (comm.hip: pipelined_step)
    hipEvent_t launch_stop_event = nullptr;
    hipEvent_t launch_start_event = nullptr;
    // which ExGEMM implementation the last call used.  The int8 path decides on the device: gemm_info_dev then points
    // at its info block (read lazily, with a synchronisation, by exblas_last_gemm_info); otherwise the host knows.
    int last_gemm_slices = 0;  // host-decided paths: 0 scalar kernel; 2..4: fp64-slice MFMA path with that many slices
    const int *gemm_info_dev = nullptr;
    int gemm_path = 0;       // 0: int8 matrix cores when the data qualifies (decided on the device) -- residues modulo
                             // 8-bit moduli for min(m, n) >= 192, base-256 digit slices below; 2: digit slices always;
                             // 4: residues always; 1: scalar kernel only; 3: fp64 slices on MFMA-F64 (host-decided)
    int gemm_max_slices = 0; // 0 = default (16): digits per operand the int8 path reserves workspace for
    int gemm_max_moduli = 0; // 0 = default (39): moduli the residue path reserves workspace for
    int gemm_ws_moduli = 0;  // moduli the last residue-path call actually reserved for (fewer after an out-of-memory retry)
    long long *gacc = nullptr;   // ACTIVE accumulator set: [ngroups][NL] int64, zero between calls
    unsigned *gflags = nullptr;  // non-finite input flags of the active set, zero between calls
    // two sets, so that the finalize of step i (side stream) can overlap the streaming kernel of step i+1
    // (exblas_set_accumulator_slot); single-stream callers never leave slot 0
    long long *gacc_all = nullptr;
    unsigned *gflags_all = nullptr;
    int slot = 0;
    // host-pointer API staging
    hipStream_t stream = nullptr;
    void *stage[3] = {nullptr, nullptr, nullptr};
    size_t stage_bytes[3] = {0, 0, 0};
    long long *d_record = nullptr;  // OUT_WORDS
    long long *h_record = nullptr;  // pinned
    // blas2/blas3 workspaces.  Growth never frees the old block (a hipGraph captured earlier may still replay into
    // it): it is parked in `retired` until exblas_release_retired_workspaces(); growth while the stream is being
    // captured is refused (hipErrorStreamCaptureUnsupported) -- reserve first (exblas_reserve_workspace).
    void *ws = nullptr;
    size_t ws_bytes = 0;
    std::vector<void *> retired;
    int layer = 0;      // 0: the *_dev layer's context, >= 1: a private context of the host-pointer layer
    std::mutex mu;      // guards launches that touch the context workspace
};

// layer 0: context of the device-pointer (*_dev) entry points; layers 1..: the host-pointer API's own contexts (own
// accumulators, flags, workspace, stream; one per "virtual device" a host call spreads its data over), so that a
// host call can never touch state an in-flight *_dev call uses
constexpr int MAX_LAYERS = 9;
Ctx &ctx(int device, int layer = 0);
void *stage_buf(Ctx &c, int slot, size_t bytes);
// nullptr + *err set when the block would have to grow while `st` is being captured into a graph
void *workspace(Ctx &c, size_t bytes, hipStream_t st, hipError_t *err);
[[noreturn]] void die(const char *what, hipError_t e, const char *file, int line);

#define EXB_CHECK(expr)                                          \
    do {                                                         \
        hipError_t _e = (expr);                                  \
        if (_e != hipSuccess) exb::die(#expr, _e, __FILE__, __LINE__); \
    } while (0)

struct GemmChunks;

// blas1.hip
hipError_t exsum_dispatch(Ctx &c, const double *a, long long n, long long inca, int fpe, int early_exit,
                          hipStream_t st, bool *supported);
hipError_t exdot_dispatch(Ctx &c, const double *a, long long inca, const double *b, long long incb, long long n,
                          int fpe, int early_exit, hipStream_t st, bool *supported);
hipError_t exsum_segmented_dispatch(const double *values, const long long *offsets, long long nseg, int fpe,
                                    int early_exit, int round_mode, hipStream_t st, double *out);
hipError_t finalize_groups(Ctx &c, hipStream_t st, long long *d_out, long long *d_ext_out = nullptr);
hipError_t finalize_sets(const long long *d_sets, int nsets, unsigned flags_or, hipStream_t st, long long *d_out,
                         const long long *d_ext_sets = nullptr);
// the *_dev layer's context of the current device (comm.hip finishes its accumulators with exported low / high sets)
Ctx &default_ctx();

// blas2.hip / blas3.hip
hipError_t exgemv_dispatch(Ctx &c, char transa, int m, int n, double alpha, const double *a, int lda,
                           const double *x, int incx, double beta, double *y, int incy, int fpe, int early_exit,
                           int round_mode, hipStream_t st);
hipError_t exgemm_dispatch(Ctx &c, char transa, char transb, int m, int n, int k, double alpha, const double *a,
                           int lda, const double *b, int ldb, double beta, double *cmat, int ldc, int fpe,
                           int early_exit, int round_mode, hipStream_t st, const GemmChunks *chunks = nullptr);

// trsv.hip
hipError_t extrsv_dispatch(Ctx &c, char uplo, char transa, char diag, int n, const double *a, int lda, double *x,
                           int incx, int fpe, int early_exit, int round_mode, hipStream_t st);

// blas3_i8.hip: the int8 ExGEMM path in two steps (whole operands, then rows of C)
struct I8Plan {
    bool ok = false;
    int m = 0, n = 0, KC = 0, kpasses = 0, dblocks = 0, force_multi = 0, maybe_multi = 0;
    int *info = nullptr, *EA = nullptr, *EB = nullptr;
    signed char *PA = nullptr, *PB = nullptr;
    unsigned long long *W = nullptr;
    // residue path (blas3_crt.hip): PA / PB hold one plane per modulus, R the residues of C
    bool crt = false;
    unsigned *R = nullptr;
    size_t plane_a = 0, plane_b = 0;
    int lcap = 0, m4 = 0, mods_per_launch = 0, num_cu = 256;
    // residue path: A' is reduced row chunk by row chunk (PA / R hold one chunk)
    int chunk_rows = 0, lda = 0, ta = 0, k = 0;
    const double *a = nullptr;
    double alpha = 1.0;
    double beta = 0.0;
    double *c = nullptr;
    int ldc = 0, round_mode = 0;
};
constexpr int I8_INFO_PATH = 7;  // info word holding the device-side decision: 0 scalar kernel, 2 int8 path
hipError_t exgemm_i8_prepare(Ctx &c, char transa, char transb, int m, int n, int k, double alpha, const double *a, int lda,
                             const double *b, int ldb, double beta, double *cmat, int ldc, int round_mode,
                             hipStream_t st, I8Plan *plan);
hipError_t exgemm_i8_rows(const I8Plan &plan, int row0, int row1, hipStream_t st);
// blas3_crt.hip: same two steps on residues modulo 8-bit moduli (info word 7 then holds 4)
hipError_t exgemm_crt_prepare(Ctx &c, char transa, char transb, int m, int n, int k, double alpha, const double *a, int lda,
                              const double *b, int ldb, double beta, double *cmat, int ldc, int round_mode,
                              hipStream_t st, I8Plan *plan);
hipError_t exgemm_crt_rows(const I8Plan &plan, int row0, int row1, hipStream_t st);
hipError_t crt_tables_upload();  // into the current device's tables; context creation only
constexpr int CRT_MIN_EDGE = 192;  // gemm_path 0: the residue path serves products with min(m, n) >= this

// Row chunks of one exgemm call: the rows [bound[i], bound[i+1]) of C are finished (on the stream) when hook(user, i)
// is called; bound[0] = 0, bound[n] = m, inner bounds multiples of 64.  Used by the row-sharded GEMM (comm.hip) to
// ship finished rows while the next chunk is computed.
struct GemmChunks {
    int n = 1;
    int bound[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    int (*hook)(void *user, int chunk, hipStream_t st) = nullptr;
    void *user = nullptr;
};
int exgemm_chunked_dev(char transa, char transb, int m, int n, int k, double alpha, const double *d_a, int lda,
                       const double *d_b, int ldb, double beta, double *d_c, int ldc, int fpe, int early_exit,
                       hipStream_t st, const GemmChunks *chunks);
bool exgemm_try_mfma(Ctx &c, char transa, char transb, int m, int n, int k, double alpha, const double *a, int lda,
                     const double *b, int ldb, double beta, double *cmat, int ldc, hipStream_t st, hipError_t *err);

int round_mode();

}  // namespace exb
