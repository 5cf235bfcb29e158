#!/usr/bin/env python3
"""bench.py -- ExSUM / ExDOT throughput on MI355X with inputs resident in HBM.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

With --gpus N > 1 and no WORLD_SIZE in the environment the script launches its own N ranks (a child
`python -m torch.distributed.run ...` started before this process touches a GPU), relays rank 0's JSON line and exits
with the child's status; it refuses (exit 2) when the box has fewer than N devices.  It never degrades to fewer ranks.

A "step" is one pass of the hot path over one of the synthetic vectors: the streaming kernel, the finalize
kernel and -- for N > 1 -- the int64 all-reduce of the three 576-byte digit sets (main + low + high) plus the re-finalize, all issued by
libexblas.so (exblas_exsum_accumulate_dev + exblas_allreduce_finish_dev: ncclAllReduce on a HIP stream, no
torch.distributed in the step).  --rotate (default 4) DISTINCT vectors are cycled through, so no step can be served by
the 256 MiB Infinity Cache from the previous one.  The same-buffer variant (every step re-reads one vector) is measured
beside it and reported under roofline.same_buffer.

Scaling (BASELINE config 3; reference: ONE vector scattered over the ranks, limbs reduced, root rounds --
src/cpu/blas/blas1/ExSUM.cpp:33-63,142-152):
  * N == 1: the vector has n = 2^log2n elements.
  * N > 1, --scaling strong (default): the SAME vectors of n_total = 2^log2n elements, rank r holding the contiguous
    shard exblas_shard_range(n_total, r, N) of each; `value` = n_total*K / max-over-ranks time; `result_bits` (the 8
    bytes of the rounded result) and `limbs_crc` are therefore comparable across N, and rank 0 re-computes the full
    vector on the CPU (`bit_exact_vs_cpu`).  The weak-scaling figure (2^log2n elements PER GPU) is measured beside it
    and reported under "weak".
  * --scaling weak makes the weak figure the headline (n_total = N * 2^log2n).

Prints ONE JSON line on rank 0 (see README / task contract), with `roofline` for the dominant kernel (k_exsum resp.
k_exdot: algorithmic 8 resp. 16 B/element over the HIP-event time of that kernel alone) and `cpu_baseline` (N == 1
only: the reference's own compiled FPE+superaccumulator core from oracle/_ref when it can be loaded, else our C port,
on the host cores).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
F64_MFMA_PEAK_TF = 78.6   # AMD MI355X datasheet fp64 matrix = fp64 vector peak (the guide lists no f64 row)
I8_MFMA_PEAK_TOPS = 5000.0  # MI355X_MICROARCH.md: I8 MFMA = 2x the BF16 rate per clock, BF16 ~2.5 PF dense


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--prewarm-ms", type=float, default=400.0,
                    help="untimed clock ramp-up before the W warmup steps: the same kernel run back to back "
                         "(the chip needs a few hundred ms of load to reach its sustained clocks)")
    ap.add_argument("--op", default="exsum", choices=["exsum", "exdot"])
    ap.add_argument("--log2n", type=int, default=28, help="elements per vector = 2^log2n (per GPU for weak scaling)")
    ap.add_argument("--scaling", default="strong", choices=["strong", "weak"],
                    help="N > 1: strong = ONE vector of 2^log2n elements partitioned n/N (BASELINE config 3, bits "
                         "comparable across N); weak = 2^log2n elements per GPU")
    ap.add_argument("--no-weak-leg", action="store_true", help="N > 1, strong: skip the secondary weak-scaling figure")
    ap.add_argument("--rotate", type=int, default=4, help="distinct input vectors the steps cycle through")
    ap.add_argument("--kind", default="ill_cond")
    ap.add_argument("--p0", type=float, default=1e32)
    ap.add_argument("--p1", type=float, default=0.0)
    ap.add_argument("--fpe", type=int, default=8)
    ap.add_argument("--no-early-exit", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the secondary line items (ExDOT, BLAS2/3, host API)")
    ap.add_argument("--skip-blas23", action="store_true", help="skip the ExGEMV / ExGEMM / ExTRSV line items")
    ap.add_argument("--blas23-timeout", type=float, default=240.0,
                    help="N > 1: seconds the BLAS2/3 items (which post collectives) may take before rank 0 prints the line "
                         "without them and the ranks exit; 0 = no limit")
    ap.add_argument("--no-host-api", action="store_true",
                    help="skip the host-pointer exsum() line item (its 64 MiB chunk launches of k_exsum would dilute the "
                         "per-kernel averages of a rocprofv3 --stats run)")
    ap.add_argument("--traffic-json", default=os.path.join(ROOT, "profiles", "hbm_traffic.json"),
                    help="per-launch HBM bytes from a separate rocprofv3 --pmc pass, if present")
    return ap.parse_args()


def timed_steps(ex, torch, dist, comm, op, buffers, fpe, ee, steps, warmup, world, rec, prewarm_ms=0.0):
    """Returns (wall seconds for `steps` steps, mean ms of the streaming kernel alone).

    With a communicator (N > 1, or the one-rank rehearsal): two-deep pipeline on two HIP streams inside ONE library call
    per step -- the streaming kernel of step i fills accumulator slot i mod 2 on the main stream; the step's second half
    (normalise, all-reduce, carry-propagate + round into a record) runs on the communicator's side stream beside the
    streaming kernel of step i+1.  Without one (N == 1): streaming kernel + finalize back to back on one stream.  Either way every step is carried to its
    final rounded result inside the timed region (drain() before the closing synchronize)."""
    ring = [rec] + [ex.new_record_buffer() for _ in range(3)]
    state = {"i": 0, "last": rec}
    nbuf = len(buffers)

    def one_step_single(e0=None, e1=None):
        # N == 1: nothing to overlap but the 5 us finalize kernel, which costs less than the event packets of the
        # two-stream pipeline (measured: 0.3147 ms per step on one stream against 0.3205 on two)
        i = state["i"]
        state["i"] += 1
        r = ring[i % len(ring)]
        if e0 is not None:
            # the pair rides on the streaming kernel's own dispatch packet (its start / stop timestamps: what rocprofv3
            # reports for the dispatch) instead of two hipEventRecord barriers around it
            ex.set_launch_events(e0, e1)
        b = buffers[i % nbuf]
        if op == "exsum":
            ex.exsum_accumulate_dev(b[0], fpe, ee)
        else:
            ex.exdot_accumulate_dev(b[0], b[1], fpe, ee)
        ex.finish_dev(out=r)
        state["last"] = r

    def one_step(e0=None, e1=None):
        if comm is None:
            return one_step_single(e0, e1)
        # ONE library call per step (exblas_ex*_allreduce_pipelined_dev): streaming kernel on this stream into the slot
        # the communicator alternates; normalise + all-reduce + round on its side stream beside the next step's kernel.
        # (The same pipeline driven from Python -- slot select, two event records, a stream switch, three launches --
        # cost ~10 us of host time per step more than the 46 us kernel of a 2^25-element shard.)
        i = state["i"]
        state["i"] += 1
        r = ring[i % len(ring)]
        b = buffers[i % nbuf]
        if op == "exsum":
            ex.exsum_allreduce_pipelined(comm, b[0], fpe, ee, out=r, ev_start=e0, ev_end=e1)
        else:
            ex.exdot_allreduce_pipelined(comm, b[0], b[1], fpe, ee, out=r, ev_start=e0, ev_end=e1)
        state["last"] = r

    def drain():
        if comm is not None:
            ex.pipeline_drain(comm)

    # clock ramp-up: untimed steps for about prewarm_ms.  With several ranks the number of steps (= collectives) must
    # be the same everywhere, so the ranks vote after every batch whether to go on.
    t_pre = time.perf_counter()
    while prewarm_ms > 0:
        go = (time.perf_counter() - t_pre) * 1e3 < prewarm_ms
        if world > 1:
            flag = torch.tensor([1 if go else 0], dtype=torch.int32, device="cuda")
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            go = bool(flag.item())
        if not go:
            break
        for _ in range(20):
            one_step()
        drain()
        torch.cuda.synchronize()
    for _ in range(warmup):
        one_step()
    drain()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    for a_, b_ in ev:        # a torch event gets its handle at its first record(): the library records them by handle
        a_.record()
        b_.record()
    stride = nbuf + 1 if nbuf > 1 else 4
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        # every event is one more packet in the queue between two streaming kernels (~3 us each): the kernel time is
        # sampled on every `stride`-th step instead of bracketing all of them; the stride is coprime with the number of
        # rotating buffers, so the samples visit every buffer (with a stride of 4 over 4 buffers the sampled kernel
        # was always the one on buffer 0 -- for ExDOT that pair of vectors happened to stream 6 % faster than the rest)
        if i % stride and steps >= 2 * stride:
            one_step()
        else:
            one_step(*ev[i])
    drain()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    timed = [p for i, p in enumerate(ev) if not (i % stride and steps >= 2 * stride)]
    samples = sorted(a.elapsed_time(b) for a, b in timed)
    kms = sum(samples) / max(len(samples), 1)
    # min and median of the sampled launches beside the mean (the reference times its kernel as the minimum of 20 runs,
    # gpu:ExSUM.cpp:149-185; SURVEY 8d asks for min and median): local to this rank, informational
    timed_steps.last_samples = {"kernel_ms_min": samples[0] if samples else None,
                                "kernel_ms_median": samples[len(samples) // 2] if samples else None,
                                "kernel_samples": len(samples)}
    if state["last"] is not rec:
        rec.copy_(state["last"])
    return dt, kms


def host_cores():
    """CPU threads this process may actually run: the affinity mask, capped by the cgroup CPU quota."""
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            cores = max(1, min(cores, int(-(-int(quota) // int(period)))))
    except Exception:  # noqa: BLE001  (cgroup v1 or no cgroup: keep the affinity count)
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p_ = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                cores = max(1, min(cores, -(-q // p_)))
        except Exception:  # noqa: BLE001
            pass
    return cores


def cpu_baseline(op, host_arrays, fpe, ee, rec_gpu):
    """Time the CPU path on this box's host cores, on the same vector(s); returns (JSON object, parity object).

    The cgroup of a GPU box grants fewer CPUs (cpu.max) than the affinity mask shows, but short runs may burst
    beyond the quota; the thread count is therefore swept (quota, 2x, 4x, ... up to the physical cores) and the
    FASTEST configuration is reported, with its thread count -- the most favourable number for the CPU -- next to
    the figure at the quota."""
    import numpy as np
    from oracle import pyoracle as O
    quota = host_cores()
    affinity = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else quota
    n = host_arrays[0].size
    use_ref = op == "exsum" and O.ref() is not None
    counts = sorted({max(1, min(affinity, c)) for c in (quota, 2 * quota, 4 * quota, affinity // 2)})

    def run(nt):
        if use_ref:
            return O.ref_exsum(host_arrays[0], fpe, ee, nthreads=nt, limbs=True)
        if op == "exsum":
            return O.exsum_omp(host_arrays[0], fpe, ee, nt, limbs=True)
        return O.exdot_omp(host_arrays[0], host_arrays[1], fpe, ee, nt, limbs=True)

    best, best_nt, res, tried = None, quota, None, {}
    t_start = time.perf_counter()
    for nt in counts:
        for _ in range(3):
            if time.perf_counter() - t_start > 25.0 and best is not None:
                break
            t0 = time.perf_counter()
            res = run(nt)
            dt = time.perf_counter() - t0
            tried[nt] = min(dt, tried.get(nt, dt))
            if best is None or dt < best:
                best, best_nt = dt, nt
    cpu_value, limbs = res
    limbs = np.asarray(limbs)
    limbs_equal = bool((limbs == np.asarray(rec_gpu.canon)).all())
    # the CPU core returns the reference's own Round() when it is the compiled reference, the correctly rounded value
    # when it is our port; the GPU record carries both roundings of the same limbs
    same = lambda a, b: np.float64(a).view(np.int64) == np.float64(b).view(np.int64)  # noqa: E731
    parity = {
        "limbs_equal": limbs_equal,
        "double_equal_exact_rounding": bool(same(O.round_limbs(limbs, O.ROUND_EXACT), rec_gpu.exact)),
        "double_equal_reference_rounding": bool(same(O.round_limbs(limbs, O.ROUND_REFERENCE), rec_gpu.refmode)),
        "cpu_returned_equals_gpu": bool(same(cpu_value, rec_gpu.refmode if use_ref else rec_gpu.exact)),
        "semantics": "limbs_equal: the 41 canonical 52-bit limbs of the GPU result == the limbs of the CPU core on the "
                     "same vector (rounding-independent).  The GPU record holds two doubles cut from those limbs: "
                     "`exact` (default API result, round-to-nearest-even = MPFR) and `refmode` (EXBLAS_ROUND=reference, "
                     "bug-compatible with Superaccumulator::Round); each is compared with the oracle's rounding of the "
                     "CPU limbs, and the double the CPU core itself returned with the matching one.",
    }
    impl = ("reference FPExpansionVect+Superaccumulator (oracle/_ref, -O1 -mavx2 -mfma) under our OpenMP slice driver"
            if use_ref else "oracle/exblas_oracle.c OpenMP port")
    sweep = ", ".join(f"{k} thr: {n / v / 1e9:.2f}" for k, v in sorted(tried.items()))
    return {
        "value": n / best / 1e9, "unit": "Gelem/s", "cores": best_nt,
        "kind": "reference" if use_ref else "port",
        "sample": f"full workload, n={n}, {impl}, fpe={fpe} early_exit={ee}; cgroup cpu quota {quota}, affinity "
                  f"{affinity}; best of 3 per thread count, Gelem/s by threads: {sweep}",
        "seconds": best,
        "value_at_cgroup_quota": n / tried[min(tried, key=lambda k: abs(k - quota))] / 1e9,
        "cgroup_quota_cores": quota,
    }, parity


def load_traffic(path, kernel):
    """Per-launch HBM bytes of `kernel` from the separate rocprofv3 --pmc passes (profiles/hbm_traffic.json)."""
    try:
        return json.load(open(path)).get(kernel)
    except Exception:  # noqa: BLE001
        return None


def probe_read(ex, torch, buffers, two_stream, n):
    """plain (inexact) streaming read over the same buffers, rotating: the box's achievable read bandwidth"""
    import ctypes as C
    lib = ex.load_library()
    sink = torch.zeros(1, dtype=torch.float64, device="cuda")
    stp = C.c_void_p(torch.cuda.current_stream().cuda_stream)

    def go(b):
        if two_stream:
            lib.exblas_stream_read2_dev(C.c_void_p(b[0].data_ptr()), C.c_void_p(b[1].data_ptr()), n, 0, stp,
                                        C.c_void_p(sink.data_ptr()))
        else:
            lib.exblas_stream_read_dev(C.c_void_p(b[0].data_ptr()), n, stp, C.c_void_p(sink.data_ptr()))

    out = []
    for bufs in (buffers, buffers[:1]):
        for i in range(4):
            go(bufs[i % len(bufs)])
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        reps = 12
        for i in range(reps):
            go(bufs[i % len(bufs)])
        e1.record()
        torch.cuda.synchronize()
        out.append(reps * n * (16 if two_stream else 8) / (e0.elapsed_time(e1) * 1e-3) / 1e9)
    return out  # [rotating, same buffer]


def bench_blas23(ex, torch, comm, world, rank):
    """ExGEMV m=n=32768 'N' column-major per GPU (config 4; replicas for N > 1) and ExGEMM n=8192 row-sharded
    over the ranks (config 5: each rank owns 8192/N rows of A and C, B replicated; alpha = beta = 1 as in the
    reference test, tests/test.exgemm.gpu.cpp:183-184).  N > 1 reports two timings: `ms` = the row-sharded product with
    C left sharded (B already on every rank: no collective at all) and `ms_gathered` = B broadcast from rank 0 + the
    product + in-place all-gather of C inside exblas_exgemm_sharded_dev."""
    def timeit(fn, reps, prewarm_ms=250.0):
        # the same policy as the headline's --prewarm-ms: the chip needs a few hundred ms of load to reach its sustained
        # clocks, and each item here starts after an idle gap (allocation, generation, host work) -- measured with
        # tools/gemv_ctx.py: the first 11 calls after such a gap run 1.52 ms, every later batch 1.34-1.39
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        while (time.perf_counter() - t0) * 1e3 < prewarm_ms:
            for _ in range(5):
                fn()
            torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps

    lib = ex.load_library()
    m = n = 32768
    a = ex.gen_dev("fpuniform", m * n, 11, 10.0, 0.0)
    x = ex.gen_dev("fpuniform", n, 12, 10.0, 0.0)
    y = ex.gen_dev("fpuniform", m, 13, 10.0, 0.0)
    gv = {"workload": "ExGEMV m=n=32768 fp64 column-major alpha=beta=1, fpe=8 early_exit, per GPU",
          "bytes": 8.0 * (m * n + n + 2 * m)}
    gv["ms"] = timeit(lambda: ex.exgemv_dev("N", m, n, 1.0, a, m, x, 1.0, y, 8, True), 10)
    gv["ms_T"] = timeit(lambda: ex.exgemv_dev("T", m, n, 1.0, a, m, x, 1.0, y, 8, True), 10)
    gv["ms_superacc_only"] = timeit(lambda: ex.exgemv_dev("N", m, n, 1.0, a, m, x, 1.0, y, 0, False), 3, 50.0)
    # ExTRSV on the same matrix storage (lower triangle of a, made diagonally dominant): latency-bound, replicas
    # for N > 1 (the substitution does not shard)
    d = ex.gen_dev("fpuniform", n, 16, 1.0, 17.0)        # diagonal entries in [2^16, 2^17): dominates 32767 entries < 1
    a.view(n, n).diagonal().copy_(d)
    b = ex.gen_dev("fpuniform_signed", n, 17, 10.0, 0.0)
    xs = torch.empty_like(b)

    def solve():
        xs.copy_(b)
        ex.extrsv_dev("L", "N", "N", n, a, n, xs, 8, True)
    ms = timeit(solve, 3, 0.0)
    tv = {"workload": "ExTRSV 'L','N','N' n=32768 fp64 column-major, fpe=8 early_exit, per GPU", "ms": ms,
          "us_per_row": ms * 1e3 / n, "n2_per_s_G": n * float(n) / (ms * 1e-3) / 1e9,
          "bound": "latency: one dependency chain of n rounded divisions (DESIGN.md 5a)",
          "rows_on_integer_path": lib.exblas_extrsv_last_slow_rows(),
          "finite": bool(torch.isfinite(xs).all())}
    del a, d, b, xs
    N = 8192
    r0, r1 = ex.row_block(N, rank, world)
    A = ex.gen_dev("fpuniform", N * N, 14, 10.0, 0.0, first=r0 * N, count=(r1 - r0) * N)
    B = ex.gen_dev("fpuniform", N * N, 15, 10.0, 0.0)
    C0 = ex.gen_dev("fpuniform", N * N, 18, 10.0, 0.0)
    C = torch.empty_like(C0)

    def gemm(gather):
        if comm is not None:
            ex.exgemm_sharded(comm, N, N, N, 1.0, A, B, 1.0, C, 8, True, b_root=0 if gather else -1, gather=gather)
        else:
            ex.exgemm_dev("N", "N", N, N, N, 1.0, A, N, B, N, 1.0, C, N, 8, True)

    def timed_gemm(gather, reps=3):
        # beta = 1 updates C in place: every timed call starts from the same C0 (the copy is outside the events)
        tot = 0.0
        for _ in range(12):                      # ~150 ms of untimed load first (see timeit)
            gemm(gather)
        for it in range(reps + 1):
            C.copy_(C0)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            gemm(gather)
            e1.record()
            torch.cuda.synchronize()
            if it:
                tot += e0.elapsed_time(e1)
        return tot / reps
    ms = timed_gemm(False)
    gm = {"workload": f"ExGEMM n=8192 fp64 row-major alpha=beta=1 (tests/test.exgemm.gpu.cpp:183-184), rows of A and C "
                      f"sharded over {world} GPU(s), B replicated" + (", C left sharded" if comm is not None else ""),
          "ms": ms, "flop_2mnk": 2.0 * N * N * N}
    gm.update(gemm_path_info(lib))
    # checksum of the rank's rows of C (position-weighted, exact in int64 modular arithmetic): comparable across N
    own = C.view(torch.int64)[r0 * N:r1 * N]
    idx = torch.arange(r0 * N, r1 * N, device="cuda", dtype=torch.int64)
    gm["c_checksum_local"] = int(((own ^ (idx * -7046029254386353131)) * 1099511628211 + idx).sum().item())
    if comm is not None:
        gm["ms_gathered"] = timed_gemm(True)
        gm["gathered_note"] = "B broadcast from rank 0 + product + in-place all-gather of C inside the call"
    return {"exgemv": gv, "exgemm": gm, "extrsv": tv}


GEMM_KERNEL = {"mfma_i8_crt": "k_gemm_crt", "mfma_i8": "k_gemm_i8", "mfma_f64": "k_gemm_mfma", "scalar": "k_gemm"}


def gemm_path_info(lib):
    """which ExGEMM implementation ran and how many slice products it issued per element pair"""
    info = {"slices": lib.exblas_last_gemm_slices()}
    if hasattr(lib, "exblas_last_gemm_info"):
        import ctypes as C
        v = (C.c_int * 8)()
        lib.exblas_last_gemm_info(v)
        if v[0] == 4:  # residues modulo 8-bit moduli: one int8 GEMM per modulus
            info.update({"path": "mfma_i8_crt", "bits_a": v[1], "bits_b": v[2], "moduli": v[3],
                         "products_per_pair": v[3]})
        else:
            info.update({"path": {0: "scalar", 1: "mfma_f64", 2: "mfma_i8"}.get(v[0], str(v[0])), "slices_a": v[1],
                         "slices_b": v[2], "products_per_pair": v[1] * v[2]})
    return info


def bench_host_api(ex, torch, x_dev, fpe, ee):
    """The reference-style host-pointer call (what a user of the reference links against): exsum(n, host array),
    H2D transfer included, beside the box's pinned H2D copy rate."""
    import numpy as np
    n = x_dev.numel()
    host = x_dev.cpu().numpy()
    pinned = torch.empty(n, dtype=torch.float64).pin_memory()
    pinned.copy_(torch.from_numpy(host))
    dst = torch.empty_like(x_dev)
    dst.copy_(pinned, non_blocking=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        dst.copy_(pinned, non_blocking=True)
    torch.cuda.synchronize()
    pcie = 3 * n * 8 / (time.perf_counter() - t0) / 1e9
    del pinned, dst
    ex.exsum(1 << 16, host, 1, 0, fpe, ee)
    best, val = None, None
    for _ in range(3):
        t0 = time.perf_counter()
        val = ex.exsum(n, host, 1, 0, fpe, ee)
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    rec = ex.read_record(ex.exsum_dev(x_dev, fpe, ee))
    return {"call": f"exsum(n=2^{n.bit_length() - 1}, host pointer, pageable memory)", "seconds": best,
            "value": n / best / 1e9, "unit": "Gelem/s", "GBs": n * 8 / best / 1e9,
            "pinned_h2d_probe_GBs": pcie, "frac_of_pinned_h2d": n * 8 / best / 1e9 / pcie,
            "same_bits_as_dev_path": bool(np.float64(val).view(np.int64) == np.float64(rec.value()).view(np.int64))}


def launch_ranks(args):
    """--gpus N > 1 without a launcher: start N fresh ranks as a CHILD process tree.  This process has not touched a
    GPU (device_count() does not initialise the runtime) and never execs; it relays rank 0's JSON line."""
    import socket
    import subprocess
    import torch
    backend = os.environ.get("EXBLAS_BENCH_BACKEND", "nccl")
    have = torch.cuda.device_count()
    if backend == "nccl" and have < args.gpus:
        print(f"[bench] --gpus {args.gpus} but only {have} HIP device(s) are visible: refusing to run with fewer ranks "
              "(EXBLAS_BENCH_BACKEND=gloo rehearses N ranks on fewer devices through the host transport)",
              file=sys.stderr, flush=True)
        sys.exit(2)
    if have < 1:
        print("[bench] no HIP device visible; there is no CPU fallback", file=sys.stderr, flush=True)
        sys.exit(2)
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True, env=env)
    line = None
    for ln in proc.stdout:
        if ln.startswith("{") and '"metric"' in ln:
            line = ln.strip()
        else:
            sys.stderr.write(ln)
    rc = proc.wait()
    if rc != 0 or line is None:
        print(f"[bench] the {args.gpus}-rank run failed (exit {rc}, JSON line {'present' if line else 'missing'})",
              file=sys.stderr, flush=True)
        sys.exit(rc or 1)
    print(line, flush=True)
    sys.exit(0)


class stdout_to_stderr:
    """fd-level redirection: RCCL prints a version banner on stdout when its first communicator comes up (torch's and
    ours); the contract is ONE JSON line on stdout"""

    def __enter__(self):
        sys.stdout.flush()
        self.saved = os.dup(1)
        os.dup2(2, 1)
        return self

    def __exit__(self, *exc):
        sys.stdout.flush()
        os.dup2(self.saved, 1)
        os.close(self.saved)
        return False


def record_ids(rec):
    """the 8 bytes of the rounded result (both roundings) and a CRC of the 41 canonical limbs: equal across GPU counts"""
    import struct
    import zlib
    import numpy as np
    return {"result_bits": "0x%016x" % struct.unpack("<Q", struct.pack("<d", rec.exact))[0],
            "result_bits_reference_rounding": "0x%016x" % struct.unpack("<Q", struct.pack("<d", rec.refmode))[0],
            "limbs_crc": "0x%08x" % (zlib.crc32(np.ascontiguousarray(rec.canon, dtype="<i8").tobytes()) & 0xffffffff)}


def cpu_parity_only(op, host_arrays, fpe, ee, rec_gpu):
    """ONE run of the CPU core on the full vector(s) (all host threads): the bit check of an N > 1 run and of the ExDOT
    leg; not a timing."""
    import numpy as np
    from oracle import pyoracle as O
    nt = max(1, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else host_cores())
    use_ref = op == "exsum" and O.ref() is not None
    t0 = time.perf_counter()
    if use_ref:
        val, limbs = O.ref_exsum(host_arrays[0], fpe, ee, nthreads=nt, limbs=True)
    elif op == "exsum":
        val, limbs = O.exsum_omp(host_arrays[0], fpe, ee, nt, limbs=True)
    else:
        val, limbs = O.exdot_omp(host_arrays[0], host_arrays[1], fpe, ee, nt, limbs=True)
    dt = time.perf_counter() - t0
    limbs = np.asarray(limbs)
    same = lambda a, b: np.float64(a).view(np.int64) == np.float64(b).view(np.int64)  # noqa: E731
    d = {"limbs_equal": bool((limbs == np.asarray(rec_gpu.canon)).all()),
         "double_equal_exact_rounding": bool(same(O.round_limbs(limbs, O.ROUND_EXACT), rec_gpu.exact)),
         "double_equal_reference_rounding": bool(same(O.round_limbs(limbs, O.ROUND_REFERENCE), rec_gpu.refmode)),
         "cpu_core": ("oracle/_ref (reference FPExpansionVect+Superaccumulator, compiled)" if use_ref
                      else "oracle/exblas_oracle.c OpenMP port"),
         "cpu_threads": nt, "cpu_seconds": dt, "n": int(host_arrays[0].size)}
    return bool(d["limbs_equal"] and d["double_equal_exact_rounding"] and d["double_equal_reference_rounding"]), d


def main():
    args = parse()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        launch_ranks(args)          # does not return
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        # never run a different number of ranks than the line will claim
        print(f"[bench] --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks", file=sys.stderr, flush=True)
        sys.exit(2)

    import torch
    import exblas_amd as ex

    dist = None
    comm = None
    transport = "none (single GPU)"
    force_dist = os.environ.get("EXBLAS_BENCH_FORCE_DIST") == "1"  # rehearse the RCCL path with a 1-rank communicator
    with stdout_to_stderr():
        if world > 1:
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            # torch.distributed is the CONTROL plane only (barriers, the max-over-ranks of the timings, handing out the
            # RCCL unique id); the data-path collectives are issued by libexblas.so on its own RCCL communicator.
            # EXBLAS_BENCH_BACKEND=gloo rehearses the N > 1 control flow on a box with fewer GPUs than ranks (ranks then
            # share the devices round-robin and the library uses its host-callback transport).
            backend = os.environ.get("EXBLAS_BENCH_BACKEND", "nccl")
            if backend == "nccl":
                if torch.cuda.device_count() < world:
                    print(f"[bench] rank {rank}: {world} ranks but {torch.cuda.device_count()} devices", file=sys.stderr,
                          flush=True)
                    sys.exit(2)
                torch.cuda.set_device(local_rank)
                dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
                transport = "native RCCL int64 all-reduce (ncclAllReduce issued by libexblas.so)"
            else:
                torch.cuda.set_device(local_rank % torch.cuda.device_count())
                dist.init_process_group(backend)
            try:
                comm = ex.Comm.from_torch()
            except Exception as e:  # noqa: BLE001
                print(f"[bench] rank {rank}: native communicator could not be created: {e}", file=sys.stderr, flush=True)
            if backend == "nccl":
                # every rank must take the same branch: if the RCCL communicator inside libexblas.so failed anywhere, all
                # ranks fall back to the library's host-callback transport over a gloo group (slower, same bits) and the
                # JSON line says so -- a number with a caveat beats no number
                okt = torch.tensor([1 if comm is not None else 0], dtype=torch.int32, device="cuda")
                dist.all_reduce(okt, op=dist.ReduceOp.MIN)
                if int(okt.item()) == 0:
                    if comm is not None:
                        comm.destroy()
                    comm = ex.Comm.from_torch(dist.new_group(backend="gloo"), transport="host")
                    transport = "host-callback transport over a gloo group (the native RCCL communicator could not be created)"
            elif comm is None:
                raise SystemExit("no communicator")
            else:
                transport = "host-callback transport (gloo rehearsal, ranks may share a GPU)"
        else:
            torch.cuda.set_device(0)
            if force_dist:
                comm = ex.Comm.rccl(ex.Comm.unique_id(), 0, 1)
                transport = "native RCCL, one-rank communicator (rehearsal)"
        if comm is not None:
            # the first collective brings the transports up (and RCCL's banner out) while stdout is still redirected
            ex.read_record(ex.exsum_allreduce(comm, torch.zeros(2, dtype=torch.float64, device="cuda"), 8, True))
            if dist is not None:
                dist.barrier()
    lib = ex.load_library()
    lib.exblas_hip_init(-1)
    n_ranks_seen = lib.exblas_comm_size(comm.handle) if comm is not None else 1
    if world > 1 and n_ranks_seen != world:
        print(f"[bench] communicator spans {n_ranks_seen} ranks, expected {world}", file=sys.stderr, flush=True)
        sys.exit(2)

    n = 1 << args.log2n
    ee = not args.no_early_exit
    nrot = max(1, args.rotate)
    strong = world > 1 and args.scaling == "strong"

    def make_buffers(op, mode):
        """mode 'strong': the rank's shard of vectors of n elements in all; 'weak': n elements per rank of vectors of
        n*world.  Seeds 1, 3, 5, ... (and 2, 4, ... for ExDOT's second operand) per rotating vector."""
        if mode == "strong":
            n_total = n
            first, last = ex.shard_range(n_total, rank, world)
        else:
            n_total = n * world
            first, last = rank * n, (rank + 1) * n
        out = []
        for j in range(nrot):
            t = [ex.gen_dev(args.kind, n_total, 1 + 2 * j, args.p0, args.p1, first=first, count=last - first,
                            n_total=n_total)]
            if op == "exdot":
                t.append(ex.gen_dev(args.kind, n_total, 2 + 2 * j, args.p0, args.p1, first=first, count=last - first,
                                    n_total=n_total))
            out.append(t)
        return out, n_total, last - first

    def full_host_vectors(op, n_total):
        """rank 0: vector 0 (seed 1, and 2) in full on the host -- generated on the GPU, where the generator is
        bit-identical to the oracle's (tests/), then copied"""
        out = []
        for seed in ((1, 2) if op == "exdot" else (1,)):
            t = ex.gen_dev(args.kind, n_total, seed, args.p0, args.p1)
            out.append(t.cpu().numpy())
            del t
        torch.cuda.empty_cache()
        return out

    def reduce_max(*vals):
        if world == 1:
            return vals
        t = torch.tensor(vals, dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return tuple(float(v) for v in t)

    def run_op(op, mode, same_buffer=True, probes=True):
        """rotating (headline) and same-buffer timings + probes of one op; returns the JSON pieces"""
        bufs, n_total, n_local = make_buffers(op, mode)
        rec = ex.new_record_buffer()
        dt, kms = timed_steps(ex, torch, dist, comm, op, bufs, args.fpe, ee, args.steps, args.warmup, world, rec,
                              args.prewarm_ms)
        dt, kms = reduce_max(dt, kms)
        bpe = 8 if op == "exsum" else 16
        ach = n_local * bpe / (kms * 1e-3) / 1e9
        roof = {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                "traffic": load_traffic(args.traffic_json, f"k_{op}") if n_local == (1 << 28) else None,
                "kernel": f"k_{op}", "kernel_ms": kms, "elements_per_launch": n_local, "rotating_buffers": nrot}
        roof.update(getattr(timed_steps, "last_samples", {}))
        if world > 1:
            roof["note"] = ("per GPU: the algorithmic bytes of the rank's shard / the slowest rank's mean kernel time; "
                            "traffic counters are collected at N = 1 only")
        result = None
        if same_buffer:
            rec1 = ex.new_record_buffer()
            dt1, kms1 = timed_steps(ex, torch, dist, comm, op, bufs[:1], args.fpe, ee, args.steps, args.warmup, world,
                                    rec1, 0.0)
            dt1, kms1 = reduce_max(dt1, kms1)
            ach1 = n_local * bpe / (kms1 * 1e-3) / 1e9
            roof["same_buffer"] = {"achieved": ach1, "frac": ach1 / HBM_PEAK_GBS, "kernel_ms": kms1,
                                   "ms_per_step": dt1 / args.steps * 1e3,
                                   "note": "every step re-reads ONE vector: up to 256 MiB of it can come from the "
                                           "Infinity Cache"}
            # the record of the LAST rotating step belongs to buffer (steps-1) mod rotate; buffer 0's result is rec1
            result = ex.read_record(rec1)
        if probes:
            probe_rot, probe_same = probe_read(ex, torch, bufs, op == "exdot", n_local)
            roof["measured_read_probe_GBs"] = probe_rot
            roof["frac_of_probe"] = ach / probe_rot
            roof["probe_note"] = ("the probe is timed over 12 back-to-back launches (the ~1 % of inter-kernel gaps included), "
                                  "kernel_ms is the dispatch's own duration: a ratio within 1 % of 1 means equal")
            if same_buffer:
                roof["same_buffer"]["measured_read_probe_GBs"] = probe_same
        return bufs, dt, roof, result, n_total

    def leg(op, want_cpu_timing):
        """the headline measurement of one op in the selected scaling mode (+ the weak figure beside a strong one,
        + the CPU bit check); returns (JSON pieces, device vector 0 for the host-API item)"""
        mode = "strong" if strong else "weak"
        bufs, dt, roof, result, n_total = run_op(op, mode)
        o = {"value": n_total * args.steps / dt / 1e9, "unit": "Gelem/s", "ms_per_step": dt / args.steps * 1e3,
             "scaling": "strong" if strong else "weak", "n_total": n_total, "roofline": roof,
             "result": result.exact}
        o.update(record_ids(result))
        keep = bufs[0][0] if world == 1 else None
        host0 = None
        if world == 1 and not args.no_cpu_baseline:
            host0 = [t.cpu().numpy() for t in bufs[0]]
        del bufs
        torch.cuda.empty_cache()
        if strong and not args.no_weak_leg:
            wb, wdt, wroof, _, wn = run_op(op, "weak", same_buffer=False, probes=False)
            o["weak"] = {"value": wn * args.steps / wdt / 1e9, "unit": "Gelem/s", "ms_per_step": wdt / args.steps * 1e3,
                         "scaling": "weak", "elements_per_gpu": n, "n_total": wn, "kernel_ms": wroof["kernel_ms"],
                         "frac_hbm_peak_per_gpu": wroof["frac"]}
            del wb
            torch.cuda.empty_cache()
        if not args.no_cpu_baseline and rank == 0:
            if world > 1:
                host0 = full_host_vectors(op, n_total)
            if want_cpu_timing and world == 1:
                base, parity = cpu_baseline(op, host0, args.fpe, ee, result)
                o["cpu_baseline"] = base
                o["bit_exact_vs_cpu"] = bool(parity["limbs_equal"] and parity["double_equal_exact_rounding"] and
                                             parity["double_equal_reference_rounding"])
                o["bit_exact_detail"] = parity
            else:
                o["bit_exact_vs_cpu"], o["bit_exact_detail"] = cpu_parity_only(op, host0, args.fpe, ee, result)
        host0 = None
        if world > 1:
            dist.barrier()      # the other ranks wait for rank 0's CPU run
        return o, keep

    head, keep = leg(args.op, True)

    def emit(secondary, blas23, host_api, extra=None):
        """rank 0: assemble and print THE JSON line from what has been measured"""
        if rank == 0:
            opname = f"Ex{args.op[2:].upper()}"
            if world == 1:
                shape = f"n=2^{args.log2n}"
                par = "single"
            elif strong:
                shape = f"ONE vector of n=2^{args.log2n} partitioned n/{world} (exblas_shard_range)"
                par = f"shard{world} of one vector, three 576-byte digit sets (main + low + high) all-reduced per step"
            else:
                shape = f"n=2^{args.log2n} per GPU ({world}*2^{args.log2n} in all)"
                par = f"shard{world}, 2^{args.log2n} elements per GPU, three 576-byte digit sets (main + low + high) all-reduced per step"
            out = {
                "metric": f"{opname} fp64 Gelem/s at n=2^{args.log2n} (bit-exact vs CPU superaccumulator/MPFR)",
                "value": head["value"],
                "unit": "Gelem/s",
                "n_gpus": n_ranks_seen,
                "steps": args.steps,
                "warmup": args.warmup,
                "ms_per_step": head["ms_per_step"],
                "higher_is_better": True,
                "scaling": head["scaling"],
                "vs_baseline": None,
                "dtype": "f64",
                "data": f"synthetic ({args.kind} p0={args.p0:g} p1={args.p1:g}, counter-based generator, {nrot} distinct "
                        f"vectors, seeds 1,3,5,...)",
                "transport": transport,
                "config": {"workload": f"{opname} {shape} fp64 {args.kind}(c={args.p0:g}), fpe={args.fpe} early_exit={ee}, "
                                       f"{world}xMI355X, inputs resident in HBM, steps rotate over {nrot} distinct vectors",
                           "n_total": head["n_total"], "fpe": args.fpe, "early_exit": ee, "parallelism": par},
            }
            for k in ("roofline", "result", "result_bits", "result_bits_reference_rounding", "limbs_crc", "weak",
                      "cpu_baseline", "bit_exact_vs_cpu", "bit_exact_detail"):
                if k in head:
                    out[k] = head[k]
            if "cpu_baseline" in head:
                out["cpu_baseline_at_cgroup_quota_Gelems"] = head["cpu_baseline"]["value_at_cgroup_quota"]
            if secondary:
                out["exdot"] = secondary
            if blas23:
                gv, gm = blas23["exgemv"], blas23["exgemm"]
                gv["GBs"] = gv["bytes"] / (gv["ms"] * 1e-3) / 1e9
                gv["frac_hbm_peak"] = gv["GBs"] / HBM_PEAK_GBS
                gv["frac_hbm_peak_T"] = gv["bytes"] / (gv["ms_T"] * 1e-3) / 1e9 / HBM_PEAK_GBS
                gv["frac_hbm_peak_superacc_only"] = gv["bytes"] / (gv["ms_superacc_only"] * 1e-3) / 1e9 / HBM_PEAK_GBS
                # roofline.frac = what the matrix pipe that does the work is asked to do / its peak: every element pair
                # costs products_per_pair multiply-adds of the MFMA type the path uses (int8 residues or digits: 5 Pop/s
                # dense; fp64 slices: 78.6 TFLOP/s).  frac_2mnk (algorithmic 2mnk flop / the fp64 matrix
                # peak) is kept beside it under its own name: it can exceed 1 because no fp64 unit does the work.
                t2 = gm["flop_2mnk"] / (gm["ms"] * 1e-3) / 1e12
                prods = gm.get("products_per_pair") or gm["slices"] ** 2
                issued = t2 * prods
                i8 = str(gm.get("path", "")).startswith("mfma_i8")
                peak_issue = (I8_MFMA_PEAK_TOPS if i8 else F64_MFMA_PEAK_TF) * world
                gm["TFLOPs_2mnk"] = t2
                gm["roofline"] = {"bound": "mfma", "achieved": issued, "peak": peak_issue,
                                  "unit": "Top/s (int8 multiply-add = 2 op)" if i8 else "TFLOP/s",
                                  "frac": issued / peak_issue,
                                  "frac_2mnk": t2 / (F64_MFMA_PEAK_TF * world),
                                  "frac_2mnk_note": "algorithmic 2mnk flop / time / fp64 matrix peak: NOT a utilisation",
                                  "traffic": load_traffic(args.traffic_json, GEMM_KERNEL.get(gm.get("path"), "k_gemm")),
                                  "kernel": GEMM_KERNEL.get(gm.get("path"), "k_gemm"),
                                  "note": "whole call (scans, residues, contraction, reconstruction); the contraction "
                                          "kernel alone: profiles/ kernel stats"}
                out["exgemv"] = gv
                out["exgemm"] = gm
                out["extrsv"] = blas23["extrsv"]
            if host_api:
                out["host_api"] = host_api
            if extra:
                out.update(extra)
            print(json.dumps(out), flush=True)

    secondary = None
    blas23 = None
    host_api = None
    if not args.no_secondary and args.op == "exsum":
        # ExDOT on the same shape (BASELINE config 3), reported beside the headline number
        secondary, _ = leg("exdot", False)
        secondary["metric"] = "ExDOT fp64 Gelem/s"
        # BASELINE configs 4 and 5 on the same box (kernel-chain time by HIP events; parity is covered by tests/)
        if not args.skip_blas23:
            # N > 1: the BLAS2/3 items include collectives that have never met real links (B broadcast, chunked all-gather
            # of C on a side stream).  Should they hang, the headline must not hang with them: after --blas23-timeout
            # seconds rank 0 prints the line with what it has and every rank leaves.
            dog = None
            if world > 1 and args.blas23_timeout > 0:
                import threading

                def bail():
                    sys.stderr.write(f"[bench] rank {rank}: BLAS2/3 items did not finish in {args.blas23_timeout} s\n")
                    sys.stderr.flush()
                    if rank == 0:
                        emit(secondary, None, None, {"blas23_timed_out_after_s": args.blas23_timeout})
                    os._exit(0)
                dog = threading.Timer(args.blas23_timeout, bail)
                dog.daemon = True
                dog.start()
            blas23 = bench_blas23(ex, torch, comm, world, rank)
            if dog is not None:
                dog.cancel()
        if blas23 and world > 1:
            gm = blas23["exgemm"]
            gvm, gmm, gmg = reduce_max(blas23["exgemv"]["ms"], gm["ms"], gm.get("ms_gathered", 0.0))
            blas23["exgemv"]["ms"], gm["ms"], gm["ms_gathered"] = gvm, gmm, gmg
            ck = torch.tensor([gm["c_checksum_local"]], dtype=torch.int64, device="cuda")
            dist.all_reduce(ck, op=dist.ReduceOp.SUM)
            gm["c_checksum"] = "0x%016x" % (int(ck.item()) & 0xffffffffffffffff)
        elif blas23:
            blas23["exgemm"]["c_checksum"] = "0x%016x" % (blas23["exgemm"]["c_checksum_local"] & 0xffffffffffffffff)
        if blas23:
            del blas23["exgemm"]["c_checksum_local"]
        if world == 1 and not args.no_host_api:
            host_api = bench_host_api(ex, torch, keep, args.fpe, ee)
    del keep

    emit(secondary, blas23, host_api)
    if comm is not None:
        comm.destroy()
    if dist is not None and dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
