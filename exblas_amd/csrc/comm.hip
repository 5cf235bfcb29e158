// comm.hip -- the multi-GPU path of libexblas.so: one process per GPU, exact results independent of the GPU count.
//
// What the reference does inside its library call (src/cpu/blas/blas1/ExSUM.cpp): rank 0 scatters slices (:33-63),
// every rank reduces its slice to a normalised superaccumulator, MPI_Reduce(MPI_LONG, MPI_SUM) adds the limbs
// (:142-152, :266-273) and the root rounds.  Here the same three steps run on the GPUs:
//   * every rank reduces its shard with the streaming kernels and normalises (k_finalize) -> 72 int64 words
//     (68 digits < 2^32 + 3 non-finite indicators) that never leave HBM;
//   * ONE int64-sum all-reduce launch over those 576 bytes and the 2 x 576 bytes of the LOW and HIGH digit sets (ExDOT
//     products below 2^-968 / beyond the double range; all zero otherwise) (RCCL over xGMI: a group of two
//     ncclAllReduce(ncclInt64, ncclSum) on the caller's stream).  Integer addition is associative and commutative, so ring/tree order, GPU count and shard boundaries
//     cannot change a bit; digits < 2^32 leave room for 2^31 ranks;
//   * every rank runs the same carry-propagation + rounding kernel on the summed digits.
// ExGEMV / ExGEMM shard the OUTPUT (rows of A and y resp. C): no reduction collective at all, only data movement --
// x resp. B replicated by one broadcast, y resp. C completed by an all-gather that overlaps the remaining compute.
//
// Transports.  RCCL is resolved at run time (dlopen of librccl.so.1, i.e. the copy already mapped when the process
// also uses PyTorch, else /opt/rocm's), so libexblas.so itself links only against the HIP runtime and single-GPU
// users never load a collective library.  The second transport is a set of host callbacks (in-place sum / broadcast /
// all-gather on host memory): what an MPI program passes to keep the reference's MPI_Reduce transport, and what the
// tests use to run several ranks on one GPU (RCCL refuses two ranks on one device).
#include "../../include/exblas_hip.h"
#include "exblas_internal.h"
#include "superacc.hip.h"

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <chrono>
#include <vector>

namespace exb {

extern std::atomic<bool> g_comm_created;  // capi.hip: a process that owns a communicator is one rank of a multi-GPU job

struct RcclApi {
    void *handle = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclBroadcast) Broadcast = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    bool ok = false;
};

static RcclApi &rccl()
{
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, [] {
        // the soname is the same for PyTorch's bundled copy and /opt/rocm's: an already loaded one is reused
        const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char *nm : names) {
            api.handle = dlopen(nm, RTLD_NOW | RTLD_GLOBAL);
            if (api.handle) break;
        }
        if (!api.handle) return;
#define EXB_SYM(field, name) api.field = (decltype(api.field))dlsym(api.handle, name)
        EXB_SYM(GetUniqueId, "ncclGetUniqueId");
        EXB_SYM(CommInitRank, "ncclCommInitRank");
        EXB_SYM(CommDestroy, "ncclCommDestroy");
        EXB_SYM(AllReduce, "ncclAllReduce");
        EXB_SYM(Broadcast, "ncclBroadcast");
        EXB_SYM(AllGather, "ncclAllGather");
        EXB_SYM(GroupStart, "ncclGroupStart");
        EXB_SYM(GroupEnd, "ncclGroupEnd");
        EXB_SYM(GetErrorString, "ncclGetErrorString");
#undef EXB_SYM
        api.ok = api.GetUniqueId && api.CommInitRank && api.CommDestroy && api.AllReduce && api.Broadcast &&
                 api.AllGather && api.GroupStart && api.GroupEnd;
    });
    return api;
}

// error space of the comm layer: hipError_t values as everywhere, or EXBLAS_COMM_ERROR for transport failures
static int nccl_rc(ncclResult_t r, const char *what)
{
    if (r == ncclSuccess) return 0;
    RcclApi &a = rccl();
    fprintf(stderr, "exblas(hip): %s failed: %s\n", what, a.GetErrorString ? a.GetErrorString(r) : "rccl error");
    return EXBLAS_COMM_ERROR;
}

}  // namespace exb

using namespace exb;

struct exblas_comm {
    int nranks = 1, rank = 0;
    int kind = 0;  // 0: RCCL, 1: host callbacks
    int device = 0;
    ncclComm_t nccl = nullptr;
    bool owned = false;
    exblas_host_allreduce_i64_fn h_allreduce = nullptr;
    exblas_host_bcast_fn h_bcast = nullptr;
    exblas_host_allgatherv_fn h_allgatherv = nullptr;
    void *user = nullptr;
    // overlap machinery of the sharded GEMM: a side stream for the all-gather pieces + events
    hipStream_t side = nullptr;
    hipEvent_t ev_chunk = nullptr, ev_done = nullptr;
    // pipelined reductions (exblas_ex*_allreduce_pipelined_dev): per accumulator slot, "streaming kernel enqueued" and
    // "slot's accumulators zero again"
    hipEvent_t ev_acc[2] = {nullptr, nullptr}, ev_zero[2] = {nullptr, nullptr};
    bool zero_pending[2] = {false, false};
    int pipe_slot = 0;
    // the low and high digit sets of a reduction in flight (ExDOT products below 2^-968 / beyond the double range), one
    // pair per accumulator slot: all-reduced beside the main digit set so that the result stays bit-identical for every
    // rank count in those corners too
    long long *xext = nullptr;   // two slots x [low | high] exported digit sets (superacc.hip.h: EXT_WORDS)
    // host transport bounce buffer (pinned)
    void *bounce = nullptr;
    size_t bounce_bytes = 0;
    std::mutex mu;
};

namespace exb {

// EXBLAS_COMM_FORCE=1: issue the broadcasts / all-gathers of the sharded calls even in a one-rank communicator
// (they are in-place no-ops there) -- lets a one-GPU box drive every RCCL call the multi-GPU path makes
static bool comm_force()
{
    static const bool f = [] {
        const char *s = getenv("EXBLAS_COMM_FORCE");
        return s && *s && *s != '0';
    }();
    return f;
}

static int comm_side(exblas_comm *cm)
{
    if (!cm->side) {
        hipError_t e = hipStreamCreateWithFlags(&cm->side, hipStreamNonBlocking);
        if (e != hipSuccess) return (int)e;
        if ((e = hipEventCreateWithFlags(&cm->ev_chunk, hipEventDisableTiming)) != hipSuccess) return (int)e;
        if ((e = hipEventCreateWithFlags(&cm->ev_done, hipEventDisableTiming)) != hipSuccess) return (int)e;
    }
    return 0;
}

static int comm_xext(exblas_comm *cm)
{
    if (cm->xext) return 0;
    hipError_t e = hipMalloc(&cm->xext, 2 * EXT_WORDS * sizeof(long long));
    if (e == hipSuccess) e = hipMemset(cm->xext, 0, 2 * EXT_WORDS * sizeof(long long));
    return (int)e;
}

static int comm_pipe(exblas_comm *cm)
{
    int rc = comm_side(cm);
    if (rc) return rc;
    for (int s = 0; s < 2; ++s)
        if (!cm->ev_acc[s]) {
            hipError_t e = hipEventCreateWithFlags(&cm->ev_acc[s], hipEventDisableTiming);
            if (e == hipSuccess) e = hipEventCreateWithFlags(&cm->ev_zero[s], hipEventDisableTiming);
            if (e != hipSuccess) return (int)e;
        }
    return 0;
}

static int bounce_buf(exblas_comm *cm, size_t bytes, void **out)
{
    if (bytes > cm->bounce_bytes) {
        if (cm->bounce) (void)hipHostFree(cm->bounce);
        cm->bounce = nullptr;
        cm->bounce_bytes = 0;
        hipError_t e = hipHostMalloc(&cm->bounce, bytes);
        if (e != hipSuccess) return (int)e;
        cm->bounce_bytes = bytes;
    }
    *out = cm->bounce;
    return 0;
}

// host transport helper: run `f(host_ptr)` on a host copy of [d_buf, d_buf + bytes) and copy the result back.
// Synchronises the stream (the callbacks are host code); refuses to do so inside a graph capture.
template <class F>
static int via_host(exblas_comm *cm, void *d_buf, size_t bytes, hipStream_t st, F &&f)
{
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (st && hipStreamIsCapturing(st, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone)
        return (int)hipErrorStreamCaptureUnsupported;
    void *h = nullptr;
    int rc = bounce_buf(cm, bytes, &h);
    if (rc) return rc;
    hipError_t e = hipMemcpyAsync(h, d_buf, bytes, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) return (int)e;
    rc = f(h);
    if (rc) return EXBLAS_COMM_ERROR;
    e = hipMemcpyAsync(d_buf, h, bytes, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);  // the bounce buffer is reused by the next call
    return (int)e;
}

// ---- the three collectives the path needs, on device memory, ordered on `st` ---------------------------------
static int comm_allreduce_i64(exblas_comm *cm, long long *d_buf, size_t count, hipStream_t st)
{
    if (cm->nranks == 1 && cm->kind == 1 && !cm->h_allreduce) return 0;
    if (cm->kind == 0)
        return nccl_rc(rccl().AllReduce(d_buf, d_buf, count, ncclInt64, ncclSum, cm->nccl, st), "ncclAllReduce");
    return via_host(cm, d_buf, count * sizeof(long long), st,
                    [&](void *h) { return cm->h_allreduce(cm->user, (int64_t *)h, (int64_t)count); });
}

// the main digit set and the [low | high] extension sets of one reduction: two buffers, ONE launch (group call) on the
// RCCL transport
static int comm_allreduce_sets(exblas_comm *cm, long long *d_main, long long *d_ext, hipStream_t st)
{
    if (cm->nranks == 1 && cm->kind == 1 && !cm->h_allreduce) return 0;
    if (cm->kind == 0) {
        RcclApi &a = rccl();
        int rc = nccl_rc(a.GroupStart(), "ncclGroupStart");
        if (!rc) rc = nccl_rc(a.AllReduce(d_main, d_main, SET_WORDS, ncclInt64, ncclSum, cm->nccl, st), "ncclAllReduce");
        if (!rc) rc = nccl_rc(a.AllReduce(d_ext, d_ext, EXT_WORDS, ncclInt64, ncclSum, cm->nccl, st), "ncclAllReduce(ext)");
        const int rc2 = nccl_rc(a.GroupEnd(), "ncclGroupEnd");
        return rc ? rc : rc2;
    }
    int rc = comm_allreduce_i64(cm, d_main, SET_WORDS, st);
    return rc ? rc : comm_allreduce_i64(cm, d_ext, EXT_WORDS, st);
}

static int comm_bcast(exblas_comm *cm, void *d_buf, size_t bytes, int root, hipStream_t st)
{
    if (bytes == 0) return 0;
    if (cm->kind == 0)
        return nccl_rc(rccl().Broadcast(d_buf, d_buf, bytes, ncclChar, root, cm->nccl, st), "ncclBroadcast");
    return via_host(cm, d_buf, bytes, st, [&](void *h) { return cm->h_bcast(cm->user, h, (int64_t)bytes, root); });
}

// in place: rank r owns bytes [off[r], off[r+1]) of d_buf; afterwards every rank holds all of [off[0], off[nranks])
static int comm_allgatherv(exblas_comm *cm, void *d_buf, const int64_t *off, hipStream_t st)
{
    const int R = cm->nranks;
    if (off[R] == off[0]) return 0;
    if (cm->kind == 0) {
        RcclApi &a = rccl();
        bool equal = true;
        for (int r = 1; r < R; ++r) equal &= (off[r + 1] - off[r]) == (off[1] - off[0]);
        if (equal) {
            char *base = (char *)d_buf + off[0];
            return nccl_rc(a.AllGather(base + (size_t)cm->rank * (off[1] - off[0]), base, (size_t)(off[1] - off[0]),
                                       ncclChar, cm->nccl, st), "ncclAllGather");
        }
        // ragged pieces: one broadcast per owner, fused into a single group launch
        int rc = nccl_rc(a.GroupStart(), "ncclGroupStart");
        for (int r = 0; r < R && !rc; ++r) {
            const size_t len = (size_t)(off[r + 1] - off[r]);
            if (len == 0) continue;
            char *p = (char *)d_buf + off[r];
            rc = nccl_rc(a.Broadcast(p, p, len, ncclChar, r, cm->nccl, st), "ncclBroadcast(piece)");
        }
        const int rc2 = nccl_rc(a.GroupEnd(), "ncclGroupEnd");
        return rc ? rc : rc2;
    }
    char *base = (char *)d_buf + off[0];
    std::vector<int64_t> rel(R + 1);
    for (int r = 0; r <= R; ++r) rel[r] = off[r] - off[0];
    return via_host(cm, base, (size_t)rel[R], st,
                    [&](void *h) { return cm->h_allgatherv(cm->user, h, rel.data()); });
}

static void shard(long long n, int rank, int nranks, long long *first, long long *last)
{
    // even boundaries: every shard of a contiguous fp64 array stays 16-byte aligned (the vector kernels' fast path)
    auto cut = [&](int r) -> long long {
        if (r >= nranks) return n;
        const long long c = (long long)(((__int128)n * r) / nranks);
        return c & ~1ll;
    };
    *first = cut(rank);
    *last = cut(rank + 1);
}

}  // namespace exb

extern "C" {

int exblas_comm_unique_id(void *id128)
{
    RcclApi &a = rccl();
    if (!a.ok) return EXBLAS_COMM_ERROR;
    ncclUniqueId id;
    int rc = nccl_rc(a.GetUniqueId(&id), "ncclGetUniqueId");
    if (rc) return rc;
    static_assert(sizeof(id) == EXBLAS_UNIQUE_ID_BYTES, "ncclUniqueId size");
    memcpy(id128, &id, sizeof(id));
    return 0;
}

int exblas_comm_init_rccl(exblas_comm_t **comm, int nranks, int rank, const void *id128)
{
    RcclApi &a = rccl();
    if (!a.ok) {
        fprintf(stderr, "exblas(hip): librccl.so.1 could not be loaded: %s\n", dlerror());
        return EXBLAS_COMM_ERROR;
    }
    if (!comm || nranks < 1 || rank < 0 || rank >= nranks) return (int)hipErrorInvalidValue;
    ctx(-1);
    ncclUniqueId id;
    memcpy(&id, id128, sizeof(id));
    ncclComm_t nc = nullptr;
    int rc = nccl_rc(a.CommInitRank(&nc, nranks, id, rank), "ncclCommInitRank");
    if (rc) return rc;
    exblas_comm *cm = new exblas_comm;
    cm->nranks = nranks;
    cm->rank = rank;
    cm->kind = 0;
    cm->nccl = nc;
    cm->owned = true;
    (void)hipGetDevice(&cm->device);
    *comm = cm;
    if (nranks > 1) g_comm_created.store(true);
    return 0;
}

int exblas_comm_adopt_rccl(exblas_comm_t **comm, void *nccl_comm, int nranks, int rank)
{
    RcclApi &a = rccl();
    if (!a.ok) return EXBLAS_COMM_ERROR;
    if (!comm || !nccl_comm || nranks < 1 || rank < 0 || rank >= nranks) return (int)hipErrorInvalidValue;
    ctx(-1);
    exblas_comm *cm = new exblas_comm;
    cm->nranks = nranks;
    cm->rank = rank;
    cm->kind = 0;
    cm->nccl = (ncclComm_t)nccl_comm;
    cm->owned = false;
    (void)hipGetDevice(&cm->device);
    *comm = cm;
    if (nranks > 1) g_comm_created.store(true);
    return 0;
}

int exblas_comm_init_host(exblas_comm_t **comm, int nranks, int rank, exblas_host_allreduce_i64_fn allreduce,
                          exblas_host_bcast_fn bcast, exblas_host_allgatherv_fn allgatherv, void *user)
{
    if (!comm || nranks < 1 || rank < 0 || rank >= nranks) return (int)hipErrorInvalidValue;
    if (nranks > 1 && (!allreduce || !bcast || !allgatherv)) return (int)hipErrorInvalidValue;
    exblas_comm *cm = new exblas_comm;
    cm->nranks = nranks;
    cm->rank = rank;
    cm->kind = 1;
    cm->h_allreduce = allreduce;
    cm->h_bcast = bcast;
    cm->h_allgatherv = allgatherv;
    cm->user = user;
    *comm = cm;
    if (nranks > 1) g_comm_created.store(true);
    return 0;
}

int exblas_comm_destroy(exblas_comm_t *cm)
{
    if (!cm) return 0;
    if (cm->side) {
        (void)hipStreamSynchronize(cm->side);
        (void)hipStreamDestroy(cm->side);
        (void)hipEventDestroy(cm->ev_chunk);
        (void)hipEventDestroy(cm->ev_done);
        for (int s = 0; s < 2; ++s) {
            if (cm->ev_acc[s]) (void)hipEventDestroy(cm->ev_acc[s]);
            if (cm->ev_zero[s]) (void)hipEventDestroy(cm->ev_zero[s]);
        }
    }
    if (cm->bounce) (void)hipHostFree(cm->bounce);
    if (cm->xext) (void)hipFree(cm->xext);
    int rc = 0;
    if (cm->kind == 0 && cm->owned && cm->nccl) rc = nccl_rc(rccl().CommDestroy(cm->nccl), "ncclCommDestroy");
    delete cm;
    return rc;
}

int exblas_comm_rank(const exblas_comm_t *cm) { return cm ? cm->rank : 0; }
int exblas_comm_size(const exblas_comm_t *cm) { return cm ? cm->nranks : 1; }

void exblas_shard_range(int64_t n, int rank, int nranks, int64_t *first, int64_t *last)
{
    long long f, l;
    shard(n, rank, nranks, &f, &l);
    *first = f;
    *last = l;
}

// finish the context's accumulators (k_finalize), all-reduce the digit set, carry-propagate + round again
int exblas_allreduce_finish_dev(exblas_comm_t *cm, void *stream, int64_t *d_out)
{
    if (!cm) return (int)hipErrorInvalidValue;
    hipStream_t st = (hipStream_t)stream;
    std::lock_guard<std::mutex> lk(cm->mu);
    int rc = comm_xext(cm);
    if (rc) return rc;
    Ctx &c = default_ctx();
    long long *ext = cm->xext + (size_t)c.slot * EXT_WORDS;
    {
        std::lock_guard<std::mutex> lc(c.mu);
        rc = (int)finalize_groups(c, st, (long long *)d_out, ext);   // main digits in the record, low / high digits exported
    }
    if (rc) return rc;
    rc = comm_allreduce_sets(cm, (long long *)d_out + OUT_DIGITS, ext, st);
    if (rc) return rc;
    // in place: k_finalize reads every input word before it writes the first output word
    return (int)finalize_sets((const long long *)d_out + OUT_DIGITS, 1, 0u, st, (long long *)d_out, ext);
}

// ---- pipelined form: the second half of reduction i runs on the communicator's side stream beside the streaming
// kernel of reduction i + 1 (two accumulator slots, alternating) -- one library call per reduction
static int pipelined_step(exblas_comm_t *cm, const double *d_a, int64_t inca, const double *d_b, int64_t incb, int64_t n,
                          int fpe, int early_exit, hipStream_t st, int64_t *d_out, hipEvent_t t0, hipEvent_t t1)
{
    if (!cm) return (int)hipErrorInvalidValue;
    std::lock_guard<std::mutex> lk(cm->mu);
    int rc = comm_pipe(cm);
    if (rc) return rc;
    const int slot = cm->pipe_slot;
    cm->pipe_slot ^= 1;
    hipError_t e = hipSuccess;
    if (cm->zero_pending[slot]) {   // the slot's previous reduction (two calls ago) must have left it zero
        // normally long done: then no wait packet goes into the stream (one packet between two streaming kernels costs
        // ~3 us); the query is not legal while the stream is being captured -- there the dependency is always recorded
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        const bool capturing = hipStreamIsCapturing(st, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone;
        bool ready = false;
        if (!capturing) {
            // A host that submits faster than the GPU executes is always a few reductions ahead, so the event is "not
            // ready" at submission time although it will be long before the kernel could start -- and the wait packet
            // that would follow is a barrier in front of the streaming kernel (launch latency exposed, ~8 us per step
            // on a 2^25-element shard).  The host waits instead, briefly and bounded: it then runs at most two
            // reductions ahead of the device, which is all a two-slot pipeline can use.  EXBLAS_PIPE_SPIN_US=0: never.
            static const long spin_us = [] { const char *v = getenv("EXBLAS_PIPE_SPIN_US"); return v ? atol(v) : 500l; }();
            const auto t_start = std::chrono::steady_clock::now();
            for (;;) {
                if (hipEventQuery(cm->ev_zero[slot]) == hipSuccess) { ready = true; break; }
                if (std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t_start).count() >= spin_us) break;
            }
            (void)hipGetLastError();    // hipEventQuery's hipErrorNotReady is not an error
        }
        if (!ready) e = hipStreamWaitEvent(st, cm->ev_zero[slot], 0);
    }
    if (e != hipSuccess) return (int)e;
    if ((rc = exblas_set_accumulator_slot(slot)) != 0) return rc;
    // The "kernel done" event the side stream waits for rides on the streaming kernel's own dispatch packet
    // (hipExtLaunchKernelGGL, exblas_internal.h: launch_stop_event).  As a separate hipEventRecord it was a barrier packet
    // between two streaming kernels, and the queue then exposed the launch latency of every kernel: in the kernel trace
    // of a 2^25-element shard the next k_exsum started 13 us after the previous one ended, together with the side
    // stream's first k_finalize (period 54 us around a 40.5 us kernel).  Not under stream capture (plain records there);
    // EXBLAS_PIPE_EXT_EVENTS=0 restores the separate records (A/B).
    static const bool ext_events = [] { const char *v = getenv("EXBLAS_PIPE_EXT_EVENTS"); return !(v && v[0] == '0'); }();
    hipStreamCaptureStatus cs2 = hipStreamCaptureStatusNone;
    const bool plain = !ext_events || (hipStreamIsCapturing(st, &cs2) == hipSuccess && cs2 != hipStreamCaptureStatusNone);
    Ctx &cd = default_ctx();
    if (plain) {
        if (t0 && (e = hipEventRecord(t0, st)) != hipSuccess) return (int)e;
    } else {
        cd.launch_start_event = t0;
        cd.launch_stop_event = t1 ? t1 : cm->ev_acc[slot];   // (a sampled step: the caller's pair brackets the kernel)
    }
    rc = d_b ? exblas_exdot_accumulate_dev(d_a, inca, d_b, incb, n, fpe, early_exit, st)
             : exblas_exsum_accumulate_dev(d_a, n, inca, fpe, early_exit, st);
    bool acc_recorded = false;
    if (!plain) {
        if (cd.launch_stop_event) {   // no kernel was launched (nothing to add, or a combination the reference ignores)
            cd.launch_start_event = cd.launch_stop_event = nullptr;
            if (!rc && t0) e = hipEventRecord(t0, st);
            if (!rc && e == hipSuccess && t1) e = hipEventRecord(t1, st);
        } else {
            acc_recorded = !t1;
        }
    } else if (!rc && t1) {
        e = hipEventRecord(t1, st);
    }
    if (rc) return rc;
    if (e != hipSuccess) return (int)e;
    if (!acc_recorded) e = hipEventRecord(cm->ev_acc[slot], st);
    if (e == hipSuccess) e = hipStreamWaitEvent(cm->side, cm->ev_acc[slot], 0);
    if (e != hipSuccess) return (int)e;
    // normalise the slot (leaves it zero), all-reduce main + low / high digit sets, carry-propagate + round: on the side stream
    if ((rc = comm_xext(cm)) != 0) return rc;
    long long *ext = cm->xext + (size_t)slot * EXT_WORDS;
    {
        Ctx &c = default_ctx();
        std::lock_guard<std::mutex> lc(c.mu);
        rc = (int)finalize_groups(c, cm->side, (long long *)d_out, ext);
    }
    if (rc) return rc;
    // the slot's accumulators are zero again from here on (the exported sets of the slot are only touched by this side
    // stream, in order): the event marks THIS point, not the end of the chain
    e = hipEventRecord(cm->ev_zero[slot], cm->side);
    cm->zero_pending[slot] = e == hipSuccess;
    if (e != hipSuccess) return (int)e;
    if ((rc = comm_allreduce_sets(cm, (long long *)d_out + OUT_DIGITS, ext, cm->side)) != 0) return rc;
    return (int)finalize_sets((const long long *)d_out + OUT_DIGITS, 1, 0u, cm->side, (long long *)d_out, ext);
}

int exblas_exsum_allreduce_pipelined_dev(exblas_comm_t *cm, const double *d_a_local, int64_t n_local, int64_t inca, int fpe,
                                         int early_exit, void *stream, int64_t *d_out, void *ev_kernel_start,
                                         void *ev_kernel_end)
{
    return pipelined_step(cm, d_a_local, inca, nullptr, 1, n_local, fpe, early_exit, (hipStream_t)stream, d_out,
                          (hipEvent_t)ev_kernel_start, (hipEvent_t)ev_kernel_end);
}

int exblas_exdot_allreduce_pipelined_dev(exblas_comm_t *cm, const double *d_a_local, int64_t inca, const double *d_b_local,
                                         int64_t incb, int64_t n_local, int fpe, int early_exit, void *stream,
                                         int64_t *d_out, void *ev_kernel_start, void *ev_kernel_end)
{
    return pipelined_step(cm, d_a_local, inca, d_b_local, incb, n_local, fpe, early_exit, (hipStream_t)stream, d_out,
                          (hipEvent_t)ev_kernel_start, (hipEvent_t)ev_kernel_end);
}

int exblas_pipeline_drain_dev(exblas_comm_t *cm, void *stream)
{
    if (!cm) return (int)hipErrorInvalidValue;
    std::lock_guard<std::mutex> lk(cm->mu);
    hipError_t e = hipSuccess;
    if (cm->zero_pending[0] || cm->zero_pending[1]) {
        // join everything the side stream holds (ev_zero marks the middle of a chain -- the slot zero again --, not its end)
        int rc0 = comm_side(cm);
        if (rc0) return rc0;
        e = hipEventRecord(cm->ev_done, cm->side);
        if (e == hipSuccess) e = hipStreamWaitEvent((hipStream_t)stream, cm->ev_done, 0);
        cm->zero_pending[0] = cm->zero_pending[1] = false;
    }
    cm->pipe_slot = 0;
    const int rc = exblas_set_accumulator_slot(0);
    return e != hipSuccess ? (int)e : rc;
}

int exblas_exsum_allreduce_dev(exblas_comm_t *cm, const double *d_a_local, int64_t n_local, int64_t inca, int fpe,
                               int early_exit, void *stream, int64_t *d_out)
{
    int rc = exblas_exsum_accumulate_dev(d_a_local, n_local, inca, fpe, early_exit, stream);
    return rc ? rc : exblas_allreduce_finish_dev(cm, stream, d_out);
}

int exblas_exdot_allreduce_dev(exblas_comm_t *cm, const double *d_a_local, int64_t inca, const double *d_b_local,
                               int64_t incb, int64_t n_local, int fpe, int early_exit, void *stream, int64_t *d_out)
{
    int rc = exblas_exdot_accumulate_dev(d_a_local, inca, d_b_local, incb, n_local, fpe, early_exit, stream);
    return rc ? rc : exblas_allreduce_finish_dev(cm, stream, d_out);
}

int exblas_exgemv_sharded_dev(exblas_comm_t *cm, char transa, int m, int n, double alpha, const double *d_a_local,
                              int lda, double *d_x, int incx, int x_root, double beta, double *d_y, int incy, int gather,
                              int fpe, int early_exit, void *stream)
{
    if (!cm || m < 0 || n < 0 || incx <= 0 || incy <= 0) return (int)hipErrorInvalidValue;
    if (m == 0 || n == 0) return 0;
    // the reference's silent no-op (ExGEMV.cpp: early_exit with fpe > 8 matches no kernel): decided here, the same way on
    // every rank, BEFORE any collective is posted
    if (early_exit && fpe > 8) return 0;
    hipStream_t st = (hipStream_t)stream;
    const bool t = (transa == 'T' || transa == 't');
    const int inner = t ? m : n, outs = t ? n : m;  // length of x, length of y
    std::lock_guard<std::mutex> lk(cm->mu);
    int rc = 0;
    const bool multi = cm->nranks > 1 || (cm->kind == 0 && comm_force());
    if (x_root >= 0 && multi)
        rc = comm_bcast(cm, d_x, ((size_t)(inner - 1) * incx + 1) * sizeof(double), x_root, st);
    if (rc) return rc;
    long long o0, o1;
    shard(outs, cm->rank, cm->nranks, &o0, &o1);
    const int loc = (int)(o1 - o0);
    if (loc > 0) {
        // 'N': this rank's rows of A and y; 'T': this rank's columns of A (= outputs), all of x
        rc = t ? exblas_exgemv_dev('T', m, loc, alpha, d_a_local, lda, d_x, incx, beta, d_y + o0 * incy, incy, fpe,
                                   early_exit, stream)
               : exblas_exgemv_dev('N', loc, n, alpha, d_a_local, lda, d_x, incx, beta, d_y + o0 * incy, incy, fpe,
                                   early_exit, stream);
        if (rc) return rc;
    }
    if (!multi || !gather) return 0;
    // rank r owns the elements [cut(r), cut(r+1)) of y, i.e. the bytes from cut(r)*incy on; the last piece ends with
    // the last element of y
    std::vector<int64_t> off(cm->nranks + 1);
    const int64_t span = ((int64_t)(outs - 1) * incy + 1) * (int64_t)sizeof(double);
    for (int r = 0; r < cm->nranks; ++r) {
        long long a0, a1;
        shard(outs, r, cm->nranks, &a0, &a1);
        const int64_t b = a0 * incy * (int64_t)sizeof(double);
        off[r] = b < span ? b : span;
    }
    off[cm->nranks] = span;
    return comm_allgatherv(cm, d_y, off.data(), st);
}

int exblas_exgemm_sharded_dev(exblas_comm_t *cm, char transa, char transb, int m, int n, int k, double alpha,
                              const double *d_a_local, int lda, double *d_b, int ldb, int b_root, double beta,
                              double *d_c, int ldc, int gather, int fpe, int early_exit, void *stream)
{
    if (!cm || m < 0 || n < 0 || k < 0 || ldc < n) return (int)hipErrorInvalidValue;
    if (m == 0 || n == 0) return 0;
    // the reference's silent no-op (ExGEMM.cpp:88-99) must be taken by every rank before any collective: a rank with
    // rows would otherwise skip the chunk hooks that a rank without rows still posts
    if (early_exit && fpe > 8) return 0;
    hipStream_t st = (hipStream_t)stream;
    const bool tb = (transb == 'T' || transb == 't');
    std::lock_guard<std::mutex> lk(cm->mu);
    int rc = 0;
    const bool multi = cm->nranks > 1 || (cm->kind == 0 && comm_force());
    if (b_root >= 0 && multi)
        rc = comm_bcast(cm, d_b, (size_t)(tb ? n : k) * (size_t)ldb * sizeof(double), b_root, st);
    if (rc) return rc;
    const int R = cm->nranks;
    long long r0, r1;
    shard(m, cm->rank, R, &r0, &r1);
    if (!gather) {
        // C stays sharded: the rank's rows only, no collective at all (rows are independent units)
        if (r1 <= r0) return 0;
        return exgemm_chunked_dev(transa, transb, (int)(r1 - r0), n, k, alpha, d_a_local, lda, d_b, ldb, beta,
                                  d_c + (size_t)r0 * ldc, ldc, fpe, early_exit, st, nullptr);
    }
    // The operands are scanned and sliced ONCE; the local rows of C are then produced in NCH chunks (boundaries at
    // multiples of 256 local rows, the block height of the int8 kernels), and the all-gather of chunk c -- one piece per
    // rank -- runs on a side stream while chunk c+1 is computed.  Chunk boundaries are the same function of (m, R) on
    // every rank, so each rank knows every other rank's pieces.
    const bool overlap = multi && cm->kind == 0;
    const long long maxloc = (m + R - 1) / R + 2;
    const int NCH = overlap ? (maxloc >= 2048 ? 4 : (maxloc >= 512 ? 2 : 1)) : 1;
    if (overlap && (rc = comm_side(cm)) != 0) return rc;
    auto chunk_rows = [&](int r, int c, long long *lo, long long *hi) {
        long long a0, a1;
        shard(m, r, R, &a0, &a1);
        const long long len = a1 - a0;
        auto cut = [&](int i) -> long long {
            if (i >= NCH) return len;
            const long long v = ((len * i / NCH) + 255) / 256 * 256;
            return v < len ? v : len;
        };
        *lo = a0 + cut(c);
        *hi = a0 + cut(c + 1);
    };
    struct Ship {
        exblas_comm *cm;
        int NCH, R, m, ldc;
        bool multi, overlap;
        double *d_c;
        decltype(chunk_rows) *rows;
    } ship{cm, NCH, R, m, ldc, multi, overlap, d_c, &chunk_rows};
    auto hook = [](void *u, int c, hipStream_t stc) -> int {
        Ship &s = *(Ship *)u;
        exblas_comm *cm = s.cm;
        if (!s.multi) return 0;
        hipStream_t cs = stc;
        if (s.overlap) {
            hipError_t e = hipEventRecord(cm->ev_chunk, stc);
            if (e == hipSuccess) e = hipStreamWaitEvent(cm->side, cm->ev_chunk, 0);
            if (e != hipSuccess) return (int)e;
            cs = cm->side;
        }
        // pieces of this chunk: rank r owns rows [lo_r, hi_r); they are not adjacent in memory, so the in-place
        // all-gather is a group of per-owner broadcasts
        if (cm->kind == 0) {
            RcclApi &a = rccl();
            int rc = nccl_rc(a.GroupStart(), "ncclGroupStart");
            for (int r = 0; r < s.R && !rc; ++r) {
                long long pl, ph;
                (*s.rows)(r, c, &pl, &ph);
                if (ph <= pl) continue;
                double *p = s.d_c + (size_t)pl * s.ldc;
                rc = nccl_rc(a.Broadcast(p, p, (size_t)(ph - pl) * s.ldc * sizeof(double), ncclChar, r, cm->nccl, cs),
                             "ncclBroadcast(C rows)");
            }
            const int rc2 = nccl_rc(a.GroupEnd(), "ncclGroupEnd");
            return rc ? rc : rc2;
        }
        std::vector<int64_t> off(s.R + 1);  // NCH == 1 here: the pieces are the ranks' whole row blocks, adjacent
        for (int r = 0; r < s.R; ++r) {
            long long pl, ph;
            (*s.rows)(r, c, &pl, &ph);
            off[r] = pl * (int64_t)s.ldc * (int64_t)sizeof(double);
            off[r + 1] = ph * (int64_t)s.ldc * (int64_t)sizeof(double);
        }
        return comm_allgatherv(cm, s.d_c, off.data(), cs);
    };
    GemmChunks ch;
    ch.n = NCH;
    for (int c = 0; c <= NCH; ++c) {
        long long lo, hi;
        chunk_rows(cm->rank, c < NCH ? c : NCH - 1, &lo, &hi);
        ch.bound[c] = (int)((c < NCH ? lo : hi) - r0);
    }
    ch.hook = hook;
    ch.user = &ship;
    if (r1 > r0) {
        rc = exgemm_chunked_dev(transa, transb, (int)(r1 - r0), n, k, alpha, d_a_local, lda, d_b, ldb, beta,
                                d_c + (size_t)r0 * ldc, ldc, fpe, early_exit, st, &ch);
        if (rc) return rc;
    } else {
        // a rank without rows still takes part in every chunk's collective
        for (int c = 0; c < NCH; ++c)
            if ((rc = hook(&ship, c, st)) != 0) return rc;
    }
    if (overlap) {
        hipError_t e = hipEventRecord(cm->ev_done, cm->side);
        if (e == hipSuccess) e = hipStreamWaitEvent(st, cm->ev_done, 0);
        if (e != hipSuccess) return (int)e;
    }
    return 0;
}

}  // extern "C"
