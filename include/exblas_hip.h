/*
 * exblas_hip.h -- C ABI of libexblas.so, the MI355X (gfx950) replacement for the reference's
 * OpenCL backend (src/gpu of nikolovjovan/exblas).  Plain pointers and sizes only.
 *
 * Two layers:
 *  (1) host-pointer entry points with exactly the argument meaning of the reference's public C++
 *      API (include/blas1.hpp:48,74; blas2.hpp:57,95; blas3.hpp:56) -- what its tests and examples call;
 *      the C++ overloads themselves (exsum/exdot/extrsv/exgemv/exgemm, global namespace) are declared in
 *      include/blas1.hpp, blas2.hpp, blas3.hpp of this repository and exported by the same library.
 *  (2) device-pointer, stream-ordered entry points underneath: what the reference's launcher layer
 *      (extern "C" initEx* / Ex* / closeEx* on cl_mem, src/gpu/blas/blas1/ExSUM.Launcher.hpp,
 *      ExDOT.Launcher.hpp, blas2/ExGEMV.Launcher.hpp, blas3/ExGEMM.Launcher.hpp) is to OpenCL.
 *      These keep data resident in HBM and are what bench.py and the multi-GPU path drive.
 *
 * Error behaviour mirrors the reference: no error channel in the BLAS-style calls;
 * fpe < 0 or a HIP failure prints to stderr and exit(EXIT_FAILURE)s (cpu:ExSUM.cpp:25-28,
 * gpu:ExSUM.cpp:111-115); Ng <= 0 returns 0.0 (ExDOT.cpp:70-71).  The *_dev functions return a
 * hipError_t-compatible int instead of exiting (0 = success).
 *
 * Concurrency: the host-pointer layer may be called from any number of threads (calls on one device are
 * serialised; the reference's GPU library is not re-entrant at all, ExSUM.Launcher.cpp:16-36).  It owns a PRIVATE
 * context per device -- its own group accumulators, flags, workspace, staging buffers and stream -- so a host call
 * never touches state an in-flight *_dev call uses, and vice versa: the two layers may be mixed freely.
 * The *_dev layer keeps ONE set of group accumulators and ONE workspace per device (two accumulator slots, see
 * exblas_set_accumulator_slot): its calls may come from any thread, but the work they enqueue must be ordered on
 * the device -- one stream, or streams chained by events -- exactly like kernels sharing a scratch buffer.
 * Callers with several independent streams create one CONTEXT HANDLE per stream (exblas_ctx_create) and use the *_ctx
 * entry points: every handle owns its accumulators, flags and workspace, so work enqueued through different handles
 * may run concurrently in any order.
 * Workspace and hipGraphs: exgemv / exgemm / extrsv use a context workspace whose size depends on the problem.  It
 * only grows; a block it outgrows is parked (not freed) so that graphs captured earlier stay valid until
 * exblas_release_retired_workspaces().  Growth needs hipMalloc, which is illegal while a stream is being captured:
 * a call that would have to grow during capture fails with hipErrorStreamCaptureUnsupported (900) -- make the same
 * call once before capturing, or exblas_reserve_workspace().  Every *_dev entry point is a pure sequence of
 * stream-ordered launches: no host synchronisation, no device-to-host read, capturable.
 *
 * Rounding: EXBLAS_ROUND=exact (default; correctly rounded = the MPFR oracle of
 * tests/test.exsum.gpu.cpp:23-38) or EXBLAS_ROUND=reference (bug-compatible with
 * Superaccumulator::Round, superaccumulator.cpp:80-134).  The limbs are identical in both.
 */
#ifndef EXBLAS_HIP_H_
#define EXBLAS_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* result record written by every *_dev reduction: EXBLAS_OUT_WORDS int64 words */
#define EXBLAS_OUT_WORDS 128
#define EXBLAS_OUT_EXACT 0    /* bit pattern of the correctly rounded double */
#define EXBLAS_OUT_REFMODE 1  /* bit pattern of the reference-compatible rounding */
#define EXBLAS_OUT_FLAGS 2    /* bit0 +inf, bit1 -inf, bit2 NaN seen in the input.  ExDOT also:
                               * bit3 (PRODUCT_UNDERFLOW) = a product of two non-zero operands was below 2^-968, i.e. had bits
                               *   below 2^-1074, the last place of the double-range accumulator;
                               * bit4 (PRODUCT_OVERFLOW) = a product of two FINITE operands was 2^1024 or more (+-Inf as a double);
                               * bit5 (PRODUCT_LOW_EXACT, with bit3) / bit6 (PRODUCT_HIGH_EXACT, with bit4) = those products lost
                               *   nothing: each was formed again at a scaled exponent (error-free) and summed in a second (LOW)
                               *   resp. third (HIGH) accumulator that the finalize kernel folded back -- the high one exactly,
                               *   1216 bits up; the low one exactly where it reaches 2^-1074 and as the half / sticky bits of
                               *   the rounding below -- so the result is the correctly rounded EXACT dot product: finite where
                               *   overflowing products cancel (IEEE arithmetic and the reference's kernels give Inf - Inf =
                               *   NaN there), +-Inf where the exact sum is beyond the double range (if it is also beyond the
                               *   record's digits, 2^1101, bit0 / bit1 is set as for an infinity in the input and the digit
                               *   fields are meaningless).  The library's multi-rank calls all-reduce the low and high digit
                               *   sets beside the main one (a host-pointer exdot spread over several devices adds them on the
                               *   host), so this holds for every rank / device count.  Bit3 / bit4 WITHOUT bit5 / bit6 only
                               *   arises from exblas_finalize_dev on user-held digit sets, which has no such sets, and means:
                               *   the correctly rounded sum of the sets' values, each truncated at 2^-1074 resp. saturated.
                               * Whatever the bits (3..6), a result of a single call or of the library's multi-rank calls is
                               * the MPFR-4196 value of tests/test.exdot.gpu.cpp:24-46.  The reference's kernels have both
                               * limits, silently. */
#define EXBLAS_FLAG_PRODUCT_UNDERFLOW 8
#define EXBLAS_FLAG_PRODUCT_OVERFLOW 16
#define EXBLAS_FLAG_PRODUCT_LOW_EXACT 32
#define EXBLAS_FLAG_PRODUCT_HIGH_EXACT 64
#define EXBLAS_OUT_CANON 4    /* 41 canonical limbs (52-bit, reference geometry) */
#define EXBLAS_OUT_DIGITS 48  /* 68 normalised 32-bit digits, then 3 flag indicators + 1 pad word: */
#define EXBLAS_SET_WORDS 72   /* words [48,120) = one "digit set", the int64-sum all-reduce payload */
#define EXBLAS_NDIGITS 68
#define EXBLAS_NCANON 41

/* generator kinds for exblas_gen_dev (restating the distributions of src/common/common.cpp) */
#define EXBLAS_GEN_NAIVE 0
#define EXBLAS_GEN_FPUNIFORM 1
#define EXBLAS_GEN_LOGNORMAL 2
#define EXBLAS_GEN_ILLCOND 3
#define EXBLAS_GEN_CANCEL 4
#define EXBLAS_GEN_FPUNIFORM_SIGNED 5

/* ---- context ------------------------------------------------------------------------------ */
/* Lazily creates the per-device context (workspace, CU count).  device < 0: current device.
 * Replaces the per-call OpenCL platform/context/queue/JIT of gpu:ExSUM.cpp:86-209. */
int exblas_hip_init(int device);
int exblas_hip_device_count(void);
const char *exblas_hip_version(void);
/* Launch-geometry knobs for A/B measurements (<= 0 / < 0 leave a value unchanged): resident blocks per CU
 * of the streaming kernels, number of global group accumulators, kernel variant (0 = production). */
int exblas_set_tuning(int blocks_per_cu, int ngroups, int variant);
/* ExGEMM implementation.  0 (default): error-free integer arithmetic on the int8 matrix cores
 * (v_mfma_i32_32x32x32_i8) for every (fpe, early_exit) variant and both rounding modes whenever the data qualifies --
 * decided on the device, the scalar kernel runs otherwise.  Products with min(m, n) >= 192 use residues modulo
 * pairwise coprime 8-bit moduli (one int8 GEMM per modulus, Chinese-remainder reconstruction: blas3_crt.hip), smaller
 * ones base-256 digit slices (all digit pairs: blas3_i8.hip); 4 = residues always, 2 = digit slices always;
 * 1 = scalar kernel only (TwoProd + expansions + one superaccumulator per output, the reference's own scheme);
 * 3 = error-free 21-bit slices on MFMA-F64 (v_mfma_f64_16x16x4_f64; host-decided: synchronises the stream,
 * exact-rounding mode only).  Same bits on every path. */
void exblas_set_gemm_path(int mode);
/* digits per operand the digit-slice path may use (workspace: that many bytes per matrix entry); 0 = default 16 */
void exblas_set_gemm_max_slices(int s);
/* moduli the residue path may use (workspace: that many bytes per entry of B and of a 2048-row chunk of A and C);
 * 0 = default 39 = every input the path accepts (126 bits per operand).  When the reservation fails the call retries with
 * 24, 18 and 12 moduli (data that needs more then takes the scalar kernel), then with the digit-slice workspace. */
void exblas_set_gemm_max_moduli(int l);
/* Which implementation the last exgemm on this device used: out[0] = 0 scalar kernel / 1 fp64 slices / 2 int8 digit
 * slices / 4 int8 residues.  Slices: out[1], out[2] = slices of A, B (out[1]*out[2] matrix multiply-adds per element
 * pair).  Residues: out[1], out[2] = bits of the fixed-point entries of A, B; out[3] = moduli (= matrix multiply-adds
 * per element pair); out[4] = moduli the workspace was reserved for (39, fewer after an out-of-memory retry).  Synchronises the device when the decision was taken there.  exblas_last_gemm_slices() =
 * max(out[1], out[2]) for the slice paths, out[3] for residues, 0 for the scalar kernel. */
int exblas_last_gemm_info(int *out8);
/* CPU-only self-test of the residue path's constant tables and reconstruction formulas: `cases` random integers per
 * modulus count through a host mirror of the device arithmetic; returns the number of failures (0 = pass). */
int exblas_crt_selftest(int cases, unsigned seed);
int exblas_last_gemm_slices(void);
/* Makes the *_dev layer's workspace at least `bytes` large (see "Workspace and hipGraphs" above).  A request that cannot
 * be met returns the hipError_t and leaves the current workspace (and the parked blocks) exactly as they were. */
int exblas_reserve_workspace(size_t bytes);
/* Bytes the *_dev layer's workspace holds right now on the current device (the parked blocks not included).  Footprint of
 * the residue path of exgemm: 39 bytes per entry of B, of a 2048-row chunk of A and of a 2048-row chunk of C --
 * 8192^3: 3.7 GiB, 16384^3: 12.2 GiB; the digit-slice path: 16 bytes per entry of A and B (+ 40 per entry of C when more
 * than one pass can be needed). */
size_t exblas_workspace_bytes(void);
/* Frees the workspace blocks that later, larger calls replaced.  Synchronises the device; only call it when no graph
 * captured before the growth will be replayed again. */
int exblas_release_retired_workspaces(void);
/* Frees the current workspace as well (a large exgemm leaves gigabytes reserved: 39 bytes per entry of A, B and
 * 4-row group of C for the residue path).  Synchronises the device; graphs captured so far must not be replayed. */
int exblas_release_workspace(void);
/* 0 = exact, 1 = reference; overrides EXBLAS_ROUND for the host-pointer API */
void exblas_set_round_mode(int mode);
int exblas_get_round_mode(void);

/* ---- (2) device-pointer layer -------------------------------------------------------------- */
/* ExSUM over d_a[i*inca], i < n (offset already applied to the pointer).  Replaces
 * initExSUM/ExSUM/closeExSUM (ExSUM.Launcher.hpp) = kernels ExSUM + ExSUMComplete
 * (ExSUM.Superacc.cl:211-356, ExSUM.FPE.cl:230-388).  fpe/early_exit select the variant exactly
 * as gpu:ExSUM.cpp:64-84.  d_out: EXBLAS_OUT_WORDS int64 in device memory. */
int exblas_exsum_dev(const double *d_a, int64_t n, int64_t inca, int fpe, int early_exit,
                     void *stream, int64_t *d_out);
/* ExDOT.  Replaces initExDOT/ExDOT/closeExDOT (ExDOT.Launcher.hpp) = kernels ExDOT +
 * ExDOTComplete (ExDOT.Superacc.cl:217-359, ExDOT.FPE.cl:201-345); variants as ExDOT.cpp:69-98. */
int exblas_exdot_dev(const double *d_a, int64_t inca, const double *d_b, int64_t incb, int64_t n,
                     int fpe, int early_exit, void *stream, int64_t *d_out);
/* Segmented (batched) ExSUM: d_out[s] = correctly rounded sum of d_values[d_offsets[s] .. d_offsets[s+1]) for
 * s < nseg, all in one launch (one wavefront per segment).  The batched form of what the reference's examples do
 * with one exsum() call per CSR row (src/cpu/examples/spmv (Parboil)/StrongReproducibility/main.cpp:85).
 * Rounding mode as set by exblas_set_round_mode / EXBLAS_ROUND. */
int exblas_exsum_segmented_dev(const double *d_values, const int64_t *d_offsets, int64_t nseg, int fpe,
                               int early_exit, void *stream, double *d_out);
/* The two phases of the calls above, separately: *_accumulate_dev launches only the streaming kernel
 * and adds its result into the context's (zero-initialised) accumulators, so several arrays can be
 * folded into ONE exact sum; exblas_finish_dev carry-propagates, rounds, writes the record and
 * leaves the accumulators zero again.  bench.py brackets the streaming kernel with events this way.
 * Capacity: one reduction (everything between two exblas_finish_dev calls) may hold up to 2^36 values (each of the
 * 32 group accumulators takes 2^31 adds; the finalize sums them carry-safely); a single call takes n < 2^31 like
 * the reference API's `int Ng` -- larger n is rejected with an error. */
int exblas_exsum_accumulate_dev(const double *d_a, int64_t n, int64_t inca, int fpe, int early_exit,
                                void *stream);
int exblas_exdot_accumulate_dev(const double *d_a, int64_t inca, const double *d_b, int64_t incb,
                                int64_t n, int fpe, int early_exit, void *stream);
int exblas_finish_dev(void *stream, int64_t *d_out);
/* The context owns TWO accumulator sets.  *_accumulate_dev and exblas_finish_dev act on the selected one
 * (0 by default).  A caller that pipelines independent reductions alternates the slot per reduction and runs
 * exblas_finish_dev on a second stream (ordered by events), so that the finalize of reduction i overlaps the
 * streaming kernel of reduction i+1 (bench.py does).  Returns 0, or an error for a slot other than 0 / 1. */
int exblas_set_accumulator_slot(int slot);
/* Timing without packets of its own: the NEXT exblas_exsum_accumulate_dev / exblas_exdot_accumulate_dev on the current
 * device's default context attaches these hipEvent_t (either may be NULL) to the dispatch packet of its streaming kernel
 * (hipExtLaunchKernelGGL): they then carry the kernel's own start / stop timestamps -- what rocprofv3 reports for that
 * dispatch -- and no hipEventRecord barrier sits in the queue around the kernel.  Consumed by that one launch; a call that
 * launches nothing leaves them unrecorded and cleared.  bench.py's kernel_ms. */
int exblas_set_launch_events(void *ev_start, void *ev_stop);
/* Sum `nsets` digit sets (EXBLAS_SET_WORDS int64 each = words [48,120) of a record, e.g. the
 * all-reduced payloads of several GPUs), carry-propagate once and round: the "single global
 * carry-propagated normalise".  d_out may alias d_digit_sets - EXBLAS_OUT_DIGITS (in-place).
 * Plays the role of MPI_Reduce + Round in cpu:ExSUM.cpp:142-156. flags_or: OR of the ranks' flags. */
int exblas_finalize_dev(const int64_t *d_digit_sets, int nsets, uint32_t flags_or, void *stream,
                        int64_t *d_out);
/* ExGEMV on device pointers, column-major A (ExGEMV.Launcher.hpp; kernels gemv/gemvT,
 * ExGEMV.Superacc.cl:192-392).  y is updated in place. */
int exblas_exgemv_dev(char transa, int m, int n, double alpha, const double *d_a, int lda,
                      const double *d_x, int incx, double beta, double *d_y, int incy, int fpe,
                      int early_exit, void *stream);
/* ExTRSV on device pointers, column-major A (ExTRSV.Launcher.hpp; kernels trsv_init/trsv,
 * ExTRSV.lnn.Superacc.cl:241-348, ExTRSV.unn.Superacc.cl:249-355).  d_x holds b on entry and the solution on
 * return: x_i = fl(Round(b_i - sum_j A(i,j) x_j) / A(i,i)), the sum exact.  uplo 'L'/'U', transa 'N'/'T',
 * diag 'N'/'U' (unit diagonal: not read).  fpe as ExTRSV.cpp:70-123: 0 superaccumulators, 1 plain DTRSV,
 * 2..8 expansions (early_exit: buckets 4/6/8); returns EXBLAS_UNSUPPORTED (-1) for fpe >= 9, the
 * iterative-refinement variants whose kernel files the reference does not ship.  Other non-zero: a hipError_t. */
#define EXBLAS_UNSUPPORTED (-1)
int exblas_extrsv_dev(char uplo, char transa, char diag, int n, const double *d_a, int lda, double *d_x,
                      int incx, int fpe, int early_exit, void *stream);
/* Diagnostics: how many rows of the most recent exact ExTRSV on this device were rounded by the integer
 * (superaccumulator) path instead of the register expansion -- near-ties, heavy cancellation, huge/tiny/non-finite
 * values.  Synchronises the device; valid until the next exgemv/exgemm/extrsv call; -1 when unknown. */
int exblas_extrsv_last_slow_rows(void);
/* ExGEMM on device pointers, row-major (ExGEMM.Launcher.hpp; kernel gemm, ExGEMM.Superacc.cl:200-283). */
int exblas_exgemm_dev(char transa, char transb, int m, int n, int k, double alpha,
                      const double *d_a, int lda, const double *d_b, int ldb, double beta,
                      double *d_c, int ldc, int fpe, int early_exit, void *stream);
/* Deterministic counter-based input generators, bit-identical to oracle/exblas_oracle.c:orc_gen_ctr;
 * element index range [first, first+count) of a vector of n_total elements. */
int exblas_gen_dev(int kind, uint64_t seed, int64_t first, int64_t count, int64_t n_total,
                   double p0, double p1, double *d_out, void *stream);
/* Plain streaming read (sum of doubles, not exact): measures the box's achievable read bandwidth,
 * the second roofline denominator of BASELINE.md section 3. */
int exblas_stream_read_dev(const double *d_a, int64_t n, void *stream, double *d_sink);
/* the same for the two-stream (ExDOT) access pattern: plain fp64 dot, blocks_per_cu <= 0 uses the ExDOT geometry */
int exblas_stream_read2_dev(const double *d_a, const double *d_b, int64_t n, int blocks_per_cu, void *stream,
                            double *d_sink);

/* ---- (2a) context handles ---------------------------------------------------------------------- */
/* The reference launchers keep their kernels and buffers in file-static globals (src/gpu/blas/blas1/ExSUM.Launcher.cpp:
 * 16-36): one call at a time per process.  The *_dev layer above relaxes that to "one ordered sequence per device"; a
 * handle relaxes it completely: exblas_ctx_create (on the current device) allocates private group accumulators (both
 * slots), flag words and -- on first use -- a private workspace; the *_ctx functions are the *_dev functions on that
 * state (handle NULL = the device's default context = exactly the *_dev call).  Calls on ONE handle must still be
 * ordered on the device; calls on different handles need no ordering at all.  A handle belongs to the device it was
 * created on: using it while another device is current returns hipErrorInvalidDevice (101).  exblas_ctx_destroy
 * synchronises the device and frees everything the handle owns (graphs captured through it must not be replayed
 * afterwards).  Tuning knobs (exblas_set_tuning, exblas_set_gemm_path, ...) are inherited at creation. */
typedef struct exblas_ctx exblas_ctx_t;
int exblas_ctx_create(exblas_ctx_t **ctx);
int exblas_ctx_destroy(exblas_ctx_t *ctx);
int exblas_exsum_ctx(exblas_ctx_t *ctx, const double *d_a, int64_t n, int64_t inca, int fpe, int early_exit,
                     void *stream, int64_t *d_out);
int exblas_exdot_ctx(exblas_ctx_t *ctx, const double *d_a, int64_t inca, const double *d_b, int64_t incb, int64_t n,
                     int fpe, int early_exit, void *stream, int64_t *d_out);
int exblas_exsum_accumulate_ctx(exblas_ctx_t *ctx, const double *d_a, int64_t n, int64_t inca, int fpe, int early_exit,
                                void *stream);
int exblas_exdot_accumulate_ctx(exblas_ctx_t *ctx, const double *d_a, int64_t inca, const double *d_b, int64_t incb,
                                int64_t n, int fpe, int early_exit, void *stream);
int exblas_finish_ctx(exblas_ctx_t *ctx, void *stream, int64_t *d_out);
int exblas_exgemv_ctx(exblas_ctx_t *ctx, char transa, int m, int n, double alpha, const double *d_a, int lda,
                      const double *d_x, int incx, double beta, double *d_y, int incy, int fpe, int early_exit,
                      void *stream);
int exblas_extrsv_ctx(exblas_ctx_t *ctx, char uplo, char transa, char diag, int n, const double *d_a, int lda,
                      double *d_x, int incx, int fpe, int early_exit, void *stream);
int exblas_exgemm_ctx(exblas_ctx_t *ctx, char transa, char transb, int m, int n, int k, double alpha,
                      const double *d_a, int lda, const double *d_b, int ldb, double beta, double *d_c, int ldc,
                      int fpe, int early_exit, void *stream);
int exblas_reserve_workspace_ctx(exblas_ctx_t *ctx, size_t bytes);
size_t exblas_workspace_bytes_ctx(exblas_ctx_t *ctx);
int exblas_last_gemm_info_ctx(exblas_ctx_t *ctx, int *out8);

/* ---- (2b) multi-GPU: one process per GPU ----------------------------------------------------- */
/* The reference reduces across processes inside the library call: local reduction, MPI_Reduce(MPI_LONG, MPI_SUM) of
 * the normalised limbs, Round on the root (src/cpu/blas/blas1/ExSUM.cpp:142-152, :266-273; scatter :33-63).  Here:
 * local reduction on each GPU, ONE int64-sum all-reduce launch over the 576-byte main digit set and the two extension sets
 * of the same size (low / high digits = ExDOT products below 2^-968 / beyond the double range, all zero otherwise), the
 * same carry-propagation +
 * rounding kernel on every rank.  Integer addition is order-free, so the result is bit-identical for any number of
 * ranks and any shard boundaries.  ExGEMV / ExGEMM shard the outputs (no reduction collective): x resp. B is
 * replicated by one broadcast, y resp. C completed by an all-gather that overlaps the remaining compute.
 *
 * A communicator wraps a transport: RCCL (resolved at run time with dlopen("librccl.so.1"); collectives are
 * enqueued on the caller's stream, nothing synchronises) or three host callbacks (what an MPI program passes to keep
 * MPI_Allreduce / MPI_Bcast / MPI_Allgatherv as the transport, and what lets several ranks share one GPU in tests;
 * this transport synchronises the stream around each callback).  All functions return 0, a hipError_t value, or
 * EXBLAS_COMM_ERROR for a transport failure (message on stderr). */
typedef struct exblas_comm exblas_comm_t;
#define EXBLAS_UNIQUE_ID_BYTES 128
#define EXBLAS_COMM_ERROR (-2)
/* rank 0: make an id (ncclGetUniqueId) and hand the 128 bytes to every rank by any means (MPI_Bcast, a file, a store) */
int exblas_comm_unique_id(void *id128);
/* every rank, current device = its GPU: ncclCommInitRank */
int exblas_comm_init_rccl(exblas_comm_t **comm, int nranks, int rank, const void *id128);
/* wrap an ncclComm_t the application already has (not destroyed by exblas_comm_destroy) */
int exblas_comm_adopt_rccl(exblas_comm_t **comm, void *nccl_comm, int nranks, int rank);
/* host transport; the callbacks operate in place on host memory and return 0 on success:
 *   allreduce: buf[0..count) := element-wise int64 sum over all ranks
 *   bcast:     the root's `bytes` bytes of buf reach every rank
 *   allgatherv: rank r owns bytes [off[r], off[r+1]) of buf (off has nranks+1 entries, off[0] == 0); afterwards every
 *               rank holds all off[nranks] bytes */
typedef int (*exblas_host_allreduce_i64_fn)(void *user, int64_t *buf, int64_t count);
typedef int (*exblas_host_bcast_fn)(void *user, void *buf, int64_t bytes, int root);
typedef int (*exblas_host_allgatherv_fn)(void *user, void *buf, const int64_t *off);
int exblas_comm_init_host(exblas_comm_t **comm, int nranks, int rank, exblas_host_allreduce_i64_fn allreduce,
                          exblas_host_bcast_fn bcast, exblas_host_allgatherv_fn allgatherv, void *user);
int exblas_comm_destroy(exblas_comm_t *comm);
int exblas_comm_rank(const exblas_comm_t *comm);
int exblas_comm_size(const exblas_comm_t *comm);
/* [first, last) of rank's contiguous shard of n items; boundaries are even, so fp64 shards stay 16-byte aligned */
void exblas_shard_range(int64_t n, int rank, int nranks, int64_t *first, int64_t *last);
/* ExSUM / ExDOT of the concatenation of every rank's local array(s): d_out (EXBLAS_OUT_WORDS int64, device) holds
 * the SAME record on every rank.  = exblas_ex*_accumulate_dev + exblas_allreduce_finish_dev. */
int exblas_exsum_allreduce_dev(exblas_comm_t *comm, const double *d_a_local, int64_t n_local, int64_t inca, int fpe,
                               int early_exit, void *stream, int64_t *d_out);
int exblas_exdot_allreduce_dev(exblas_comm_t *comm, const double *d_a_local, int64_t inca, const double *d_b_local,
                               int64_t incb, int64_t n_local, int fpe, int early_exit, void *stream, int64_t *d_out);
/* The second half alone: normalise the selected accumulator slot (exblas_finish_dev), all-reduce the digit set,
 * carry-propagate + round.  A caller that pipelines reductions runs this on a second stream while the next streaming
 * kernel fills the other slot (bench.py). */
int exblas_allreduce_finish_dev(exblas_comm_t *comm, void *stream, int64_t *d_out);
/* Throughput form for a caller that runs reduction after reduction (bench.py's step loop): ONE call per reduction.  The
 * streaming kernel runs on `stream` into the accumulator slot the communicator alternates (0, 1, 0, ...); the second half
 * -- normalise, all-reduce, carry-propagate + round into d_out -- is enqueued on the communicator's own side stream behind
 * an event, so it overlaps the NEXT call's streaming kernel (a slot is reused only after its second half has left it
 * zero: the call waits for that -- on the HOST, for at most EXBLAS_PIPE_SPIN_US microseconds (default 500; the host
 * then runs at most two reductions ahead of the device), after that by making `stream` wait; a wait packet in front of
 * the streaming kernel would expose its launch latency on every call).  d_out of call i is complete once work ordered after
 * exblas_pipeline_drain_dev(comm, stream) runs (which also re-selects slot 0); use distinct d_out buffers for reductions
 * in flight (two).  ev_kernel_start / ev_kernel_end: optional hipEvent_t that bracket the streaming kernel (attached to
 * its dispatch packet: they carry the kernel's own start / stop timestamps; bench.py times the kernel with them), NULL
 * otherwise.  Uses the device's default context: do not interleave with
 * exblas_*_accumulate_dev / exblas_finish_dev on it before draining. */
int exblas_exsum_allreduce_pipelined_dev(exblas_comm_t *comm, const double *d_a_local, int64_t n_local, int64_t inca,
                                         int fpe, int early_exit, void *stream, int64_t *d_out, void *ev_kernel_start,
                                         void *ev_kernel_end);
int exblas_exdot_allreduce_pipelined_dev(exblas_comm_t *comm, const double *d_a_local, int64_t inca,
                                         const double *d_b_local, int64_t incb, int64_t n_local, int fpe, int early_exit,
                                         void *stream, int64_t *d_out, void *ev_kernel_start, void *ev_kernel_end);
int exblas_pipeline_drain_dev(exblas_comm_t *comm, void *stream);
/* Row-sharded ExGEMV.  transa 'N': rank r owns rows [first, last) = exblas_shard_range(m, r, size) of A and y;
 * d_a_local is that row block (column-major, leading dimension lda >= last - first).  transa 'T': rank r owns the
 * OUTPUTS [first, last) of n, i.e. columns first..last-1 of A; d_a_local points at column `first`.  d_x is the full
 * vector on every rank; x_root >= 0 broadcasts it from that rank first, x_root < 0 says it is already replicated.
 * d_y is the full vector on every rank: on entry a rank's own part holds its input (beta != 0); gather != 0: on return
 * every rank holds all of y (in-place all-gather); gather == 0: only the rank's own part is written and NO collective
 * is issued after the product (y stays sharded: what a caller that keeps iterating on row blocks wants).
 * early_exit with fpe > 8 is the reference's silent no-op: nothing is computed or communicated, on any rank. */
int exblas_exgemv_sharded_dev(exblas_comm_t *comm, char transa, int m, int n, double alpha, const double *d_a_local,
                              int lda, double *d_x, int incx, int x_root, double beta, double *d_y, int incy, int gather,
                              int fpe, int early_exit, void *stream);
/* Row-sharded ExGEMM (row-major): rank r owns rows [first, last) = exblas_shard_range(m, r, size) of op(A) and C;
 * d_a_local points at the rank's first row of op(A) (for transa 'T': column `first` of the stored k x m matrix).
 * d_b: all of op(B)'s storage on every rank, broadcast from b_root first when b_root >= 0.  d_c: the full m x ldc
 * matrix on every rank; a rank's own rows hold its input (beta != 0).  gather != 0: on return every rank holds all of C
 * (with the RCCL transport the all-gather of a finished row chunk overlaps the computation of the next one);
 * gather == 0: only the rank's own rows are written, no collective follows the product -- with b_root < 0 the call
 * then issues no collective at all (rows of C are independent units; the reference has no distributed GEMM to imitate,
 * and an m x n all-gather costs more than the product itself from 4 GPUs up). */
int exblas_exgemm_sharded_dev(exblas_comm_t *comm, char transa, char transb, int m, int n, int k, double alpha,
                              const double *d_a_local, int lda, double *d_b, int ldb, int b_root, double beta,
                              double *d_c, int ldc, int gather, int fpe, int early_exit, void *stream);

/* ---- (1) host-pointer layer (reference semantics; copies H2D per call like gpu:ExSUM.cpp:126) -- */
/* exsum / exdot of host vectors spread one call over several GPUs of the node (each streams its contiguous part in
 * 64 MiB chunks through its own PCIe link; the 576-byte digit sets of the parts are added and rounded once): the
 * reference's rank-0 scatter (cpu:ExSUM.cpp:33-63) inside one process.  Default: every visible device for inputs of
 * 256 MiB or more in a stand-alone process; always the current device only in a one-process-per-GPU job (a launcher's
 * WORLD_SIZE / OMPI_COMM_WORLD_SIZE / PMI_SIZE / SLURM_NTASKS > 1 in the environment, or a communicator of
 * more than one rank created through exblas_comm_*), where every rank sees its peers' devices;
 * `EXBLAS_HOST_DEVICES=all|current|0,1,...` or this call override it
 * (count == 0 restores the default; a device may be listed twice -- two independent parts on one GPU).  The result
 * does not depend on the choice.  exgemv / exgemm / extrsv always use the current device. */
int exblas_set_host_devices(int count, const int *devices);
double exblas_exsum(int Ng, const double *ag, int inca, int offset, int fpe, int early_exit);
double exblas_exdot(int Ng, const double *ag, int inca, int offseta, const double *bg, int incb,
                    int offsetb, int fpe, int early_exit);
int exblas_exgemv(char transa, int m, int n, double alpha, const double *a, int lda, int offseta,
                  const double *x, int incx, int offsetx, double beta, double *y, int incy,
                  int offsety, int fpe, int early_exit);
/* blas2.hpp:57 extrsv; returns 0, or -1 (message on stderr, x untouched) for fpe >= 9 */
int exblas_extrsv(char uplo, char transa, char diag, int n, const double *a, int lda, int offseta, double *x,
                  int incx, int offsetx, int fpe, int early_exit);
int exblas_exgemm(char transa, char transb, int m, int n, int k, double alpha, const double *a,
                  int lda, const double *b, int ldb, double beta, double *c, int ldc, int fpe,
                  int early_exit);
/* as exblas_exsum / exblas_exdot, additionally returning the full record (limbs, both roundings) */
int exblas_exsum_record(int Ng, const double *ag, int inca, int offset, int fpe, int early_exit,
                        int64_t *out_words);
int exblas_exdot_record(int Ng, const double *ag, int inca, int offseta, const double *bg, int incb,
                        int offsetb, int fpe, int early_exit, int64_t *out_words);

#ifdef __cplusplus
}
#endif
#endif /* EXBLAS_HIP_H_ */
