#!/usr/bin/env python3
"""bench.py -- ExSUM / ExDOT throughput on MI355X with inputs resident in HBM.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over one of the rank's synthetic vectors: the streaming kernel, the finalize
kernel and -- for N > 1 -- the 576-byte int64 all-reduce of the digit set plus the re-finalize, all issued by
libexblas.so (exblas_exsum_accumulate_dev + exblas_allreduce_finish_dev: ncclAllReduce on a HIP stream, no
torch.distributed in the step).  Weak scaling: every rank holds --rotate (default 4) DISTINCT vectors of n elements
(default 2^28 = 2 GiB each) and step i reads vector i mod rotate, so no step can be served by the 256 MiB Infinity
Cache from the previous one; `value` = N*n*K / max-over-ranks time.  The same-buffer variant (every step re-reads one
vector) is measured beside it and reported under roofline.same_buffer.

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
    ap.add_argument("--log2n", type=int, default=28, help="elements per GPU and vector = 2^log2n")
    ap.add_argument("--rotate", type=int, default=4, help="distinct input vectors the steps cycle through")
    ap.add_argument("--kind", default="ill_cond")
    ap.add_argument("--p0", type=float, default=1e32)
    ap.add_argument("--p1", type=float, default=0.0)
    ap.add_argument("--fpe", type=int, default=8)
    ap.add_argument("--no-early-exit", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the secondary line items (ExDOT, BLAS2/3, host API)")
    ap.add_argument("--no-host-api", action="store_true",
                    help="skip the host-pointer exsum() line item (its 64 MiB chunk launches of k_exsum would dilute the "
                         "per-kernel averages of a rocprofv3 --stats run)")
    ap.add_argument("--traffic-json", default=os.path.join(ROOT, "profiles", "hbm_traffic.json"),
                    help="per-launch HBM bytes from a separate rocprofv3 --pmc pass, if present")
    return ap.parse_args()


def timed_steps(ex, torch, dist, comm, op, buffers, fpe, ee, steps, warmup, world, rec, prewarm_ms=0.0):
    """Returns (wall seconds for `steps` steps, mean ms of the streaming kernel alone).

    With a communicator (N > 1, or the one-rank rehearsal): two-deep pipeline on two HIP streams -- the streaming kernel
    of step i fills accumulator slot i mod 2 on the main stream; the step's second half (normalise, all-reduce,
    carry-propagate + round into a record) runs on a side stream beside the streaming kernel of step i+1.  Without
    one (N == 1): streaming kernel + finalize back to back on one stream.  Either way every step is carried to its
    final rounded result inside the timed region (drain() before the closing synchronize)."""
    main = torch.cuda.current_stream()
    side = torch.cuda.Stream()
    ring = [rec] + [ex.new_record_buffer() for _ in range(3)]
    ev_acc = [torch.cuda.Event(), torch.cuda.Event()]        # streaming kernel of the slot enqueued
    ev_done = [None, None]                                   # the slot's accumulators are zero again
    state = {"i": 0, "last": rec}
    nbuf = len(buffers)

    def one_step_single(e0=None, e1=None):
        # N == 1: nothing to overlap but the 5 us finalize kernel, which costs less than the event packets of the
        # two-stream pipeline (measured: 0.3147 ms per step on one stream against 0.3205 on two)
        i = state["i"]
        state["i"] += 1
        r = ring[i % len(ring)]
        if e0 is not None:
            e0.record()
        b = buffers[i % nbuf]
        if op == "exsum":
            ex.exsum_accumulate_dev(b[0], fpe, ee)
        else:
            ex.exdot_accumulate_dev(b[0], b[1], fpe, ee)
        if e1 is not None:
            e1.record()
        ex.finish_dev(out=r)
        state["last"] = r

    def one_step(e0=None, e1=None):
        if comm is None:
            return one_step_single(e0, e1)
        i = state["i"]
        state["i"] += 1
        slot = i & 1
        r = ring[i % len(ring)]
        if ev_done[slot] is not None:
            main.wait_event(ev_done[slot])
        ex.set_accumulator_slot(slot)
        if e0 is not None:
            e0.record()
        b = buffers[i % nbuf]
        if op == "exsum":
            ex.exsum_accumulate_dev(b[0], fpe, ee)
        else:
            ex.exdot_accumulate_dev(b[0], b[1], fpe, ee)
        if e1 is not None:
            e1.record()
        ev_acc[slot].record(main)
        with torch.cuda.stream(side):
            side.wait_event(ev_acc[slot])
            if comm is not None:
                ex.allreduce_finish(comm, out=r)
            else:
                ex.finish_dev(out=r)
            if ev_done[slot] is None:
                ev_done[slot] = torch.cuda.Event()
            ev_done[slot].record(side)
        state["last"] = r

    def drain():
        main.wait_stream(side)

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
    kms = sum(a.elapsed_time(b) for a, b in timed) / max(len(timed), 1)
    ex.set_accumulator_slot(0)
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
    over the ranks (config 5: each rank owns 8192/N rows of A and C; B broadcast and C all-gathered by
    exblas_exgemm_sharded_dev when N > 1)."""
    def timeit(fn, reps):
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
    gv["ms_superacc_only"] = timeit(lambda: ex.exgemv_dev("N", m, n, 1.0, a, m, x, 1.0, y, 0, False), 3)
    # ExTRSV on the same matrix storage (lower triangle of a, made diagonally dominant): latency-bound, replicas
    # for N > 1 (the substitution does not shard)
    d = ex.gen_dev("fpuniform", n, 16, 1.0, 17.0)        # diagonal entries in [2^16, 2^17): dominates 32767 entries < 1
    a.view(n, n).diagonal().copy_(d)
    b = ex.gen_dev("fpuniform_signed", n, 17, 10.0, 0.0)
    xs = torch.empty_like(b)

    def solve():
        xs.copy_(b)
        ex.extrsv_dev("L", "N", "N", n, a, n, xs, 8, True)
    ms = timeit(solve, 3)
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
    C = torch.zeros(N * N, dtype=torch.float64, device="cuda")

    def gemm():
        if comm is not None:
            ex.exgemm_sharded(comm, N, N, N, 1.0, A, B, 0.0, C, 8, True, b_root=0)
        else:
            ex.exgemm_dev("N", "N", N, N, N, 1.0, A, N, B, N, 0.0, C, N, 8, True)
    ms = timeit(gemm, 3)
    gm = {"workload": f"ExGEMM n=8192 fp64 row-major alpha=1 beta=0, rows sharded over {world} GPU(s)"
                      + (", B broadcast + C all-gather inside the call" if comm is not None else ""),
          "ms": ms, "flop_2mnk": 2.0 * N * N * N}
    gm.update(gemm_path_info(lib))
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


def main():
    args = parse()
    import torch
    import exblas_amd as ex

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    comm = None
    transport = "native RCCL int64 all-reduce (ncclAllReduce issued by libexblas.so)"
    force_dist = os.environ.get("EXBLAS_BENCH_FORCE_DIST") == "1"  # rehearse the RCCL path with a 1-rank communicator
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # torch.distributed is the CONTROL plane only (barriers, the max-over-ranks of the timings, handing out the
        # RCCL unique id); the data-path collectives are issued by libexblas.so on its own RCCL communicator.
        # EXBLAS_BENCH_BACKEND=gloo rehearses the N > 1 control flow on a box with fewer GPUs than ranks (ranks then
        # share the devices round-robin and the library uses its host-callback transport).
        backend = os.environ.get("EXBLAS_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
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
            transport = "host-callback transport (gloo rehearsal)"
    else:
        torch.cuda.set_device(0)
        if force_dist:
            comm = ex.Comm.rccl(ex.Comm.unique_id(), 0, 1)
    assert world == args.gpus or world == 1, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    ex.load_library().exblas_hip_init(-1)

    n = 1 << args.log2n
    n_total = n * world
    ee = not args.no_early_exit
    first = rank * n
    nrot = max(1, args.rotate)

    def make_buffers(op):
        out = []
        for j in range(nrot):
            t = [ex.gen_dev(args.kind, n, 1 + 2 * j, args.p0, args.p1, first=first, count=n, n_total=n_total)]
            if op == "exdot":
                t.append(ex.gen_dev(args.kind, n, 2 + 2 * j, args.p0, args.p1, first=first, count=n, n_total=n_total))
            out.append(t)
        return out

    def reduce_max(*vals):
        if world == 1:
            return vals
        t = torch.tensor(vals, dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return tuple(float(v) for v in t)

    def run_op(op):
        """rotating (headline) and same-buffer timings + probes of one op; returns the JSON pieces"""
        bufs = make_buffers(op)
        rec = ex.new_record_buffer()
        dt, kms = timed_steps(ex, torch, dist, comm, op, bufs, args.fpe, ee, args.steps, args.warmup, world, rec,
                              args.prewarm_ms)
        dt, kms = reduce_max(dt, kms)
        rec1 = ex.new_record_buffer()
        dt1, kms1 = timed_steps(ex, torch, dist, comm, op, bufs[:1], args.fpe, ee, args.steps, args.warmup, world,
                                rec1, 0.0)
        dt1, kms1 = reduce_max(dt1, kms1)
        probe_rot, probe_same = probe_read(ex, torch, bufs, op == "exdot", n)
        bpe = 8 if op == "exsum" else 16
        ach, ach1 = n * bpe / (kms * 1e-3) / 1e9, n * bpe / (kms1 * 1e-3) / 1e9
        roof = {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                "traffic": load_traffic(args.traffic_json, f"k_{op}"), "kernel": f"k_{op}", "kernel_ms": kms,
                "rotating_buffers": nrot, "measured_read_probe_GBs": probe_rot, "frac_of_probe": ach / probe_rot,
                "same_buffer": {"achieved": ach1, "frac": ach1 / HBM_PEAK_GBS, "kernel_ms": kms1,
                                "ms_per_step": dt1 / args.steps * 1e3, "measured_read_probe_GBs": probe_same,
                                "note": "every step re-reads ONE 2^log2n-element vector: up to 256 MiB of it can "
                                        "come from the Infinity Cache"}}
        # the record of the LAST step belongs to buffer (steps-1) mod rotate; buffer 0's result comes from rec1
        return bufs, dt, roof, ex.read_record(rec1)

    bufs, dt, roof, result = run_op(args.op)
    x0 = bufs[0]

    secondary = None
    blas23 = None
    host_api = None
    if not args.no_secondary and args.op == "exsum":
        host0 = [t.cpu().numpy() for t in x0] if (world == 1 and not args.no_cpu_baseline) else None
        keep = x0[0]
        del bufs
        torch.cuda.empty_cache()
        # ExDOT on the same shape (BASELINE config 2), reported beside the headline number
        dbufs, ddt, droof, dres = run_op("exdot")
        secondary = {"metric": "ExDOT fp64 Gelem/s", "value": n_total * args.steps / ddt / 1e9, "unit": "Gelem/s",
                     "ms_per_step": ddt / args.steps * 1e3, "roofline": droof, "result": dres.exact}
        del dbufs
        torch.cuda.empty_cache()
        # BASELINE configs 4 and 5 on the same box (kernel-chain time by HIP events; parity is covered by tests/)
        blas23 = bench_blas23(ex, torch, comm, world, rank)
        if world > 1:
            gvm, gmm = reduce_max(blas23["exgemv"]["ms"], blas23["exgemm"]["ms"])
            blas23["exgemv"]["ms"], blas23["exgemm"]["ms"] = gvm, gmm
        if world == 1 and not args.no_host_api:
            host_api = bench_host_api(ex, torch, keep, args.fpe, ee)
    else:
        host0 = [t.cpu().numpy() for t in x0] if (world == 1 and not args.no_cpu_baseline) else None

    if rank == 0:
        out = {
            "metric": f"Ex{args.op[2:].upper()} fp64 Gelem/s at n=2^{args.log2n} per GPU (bit-exact vs CPU superaccumulator/MPFR)",
            "value": n_total * args.steps / dt / 1e9,
            "unit": "Gelem/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": f"synthetic ({args.kind} p0={args.p0:g} p1={args.p1:g}, counter-based generator, {nrot} distinct "
                    f"vectors per GPU, seeds 1,3,5,...)",
            "config": {"workload": f"Ex{args.op[2:].upper()} n=2^{args.log2n} fp64 {args.kind}"
                                   f"(c={args.p0:g}) per GPU, fpe={args.fpe} early_exit={ee}, "
                                   f"{world}xMI355X, inputs resident in HBM, steps rotate over {nrot} distinct vectors",
                       "elements_per_gpu": n, "fpe": args.fpe, "early_exit": ee,
                       "parallelism": (f"shard{world}, 576-byte digit set per step, {transport}"
                                       if world > 1 else "single")},
            "roofline": roof,
            "result": result.exact,
        }
        if secondary:
            out["exdot"] = secondary
        if blas23:
            gv, gm = blas23["exgemv"], blas23["exgemm"]
            gv["GBs"] = gv["bytes"] / (gv["ms"] * 1e-3) / 1e9
            gv["frac_hbm_peak"] = gv["GBs"] / HBM_PEAK_GBS
            gv["frac_hbm_peak_T"] = gv["bytes"] / (gv["ms_T"] * 1e-3) / 1e9 / HBM_PEAK_GBS
            gv["frac_hbm_peak_superacc_only"] = gv["bytes"] / (gv["ms_superacc_only"] * 1e-3) / 1e9 / HBM_PEAK_GBS
            # algorithmic work of ExGEMM = 2*m*n*k flop; the roofline it is priced against is the fp64
            # matrix peak.  mfma_util = issued MFMA work / the peak of the MFMA type the path actually uses: every
            # element pair costs slices_a * slices_b MFMA multiply-adds (fp64 slices of 21 bits or int8 slices of 8).
            t2 = gm["flop_2mnk"] / (gm["ms"] * 1e-3) / 1e12
            prods = gm.get("products_per_pair") or gm["slices"] ** 2
            issued = t2 * prods
            peak_issue = I8_MFMA_PEAK_TOPS if str(gm.get("path", "")).startswith("mfma_i8") else F64_MFMA_PEAK_TF
            gm["TFLOPs_2mnk"] = t2
            gm["roofline"] = {"bound": "mfma", "achieved": t2, "peak": F64_MFMA_PEAK_TF * world, "unit": "TFLOP/s",
                              "frac": t2 / (F64_MFMA_PEAK_TF * world), "frac_2mnk": t2 / (F64_MFMA_PEAK_TF * world),
                              "mfma_issued_Tops": issued, "mfma_issue_peak_Tops": peak_issue * world,
                              "mfma_util": issued / (peak_issue * world),
                              "traffic": load_traffic(args.traffic_json, GEMM_KERNEL.get(gm.get("path"), "k_gemm")),
                              "kernel": GEMM_KERNEL.get(gm.get("path"), "k_gemm"),
                              "note": "frac = frac_2mnk: algorithmic 2mnk flop / time / fp64 matrix peak (SURVEY 8(d)); it "
                                      "can exceed 1 because the exact product runs on the int8 matrix cores (residues "
                                      "modulo 8-bit moduli: products_per_pair int8 GEMMs); mfma_util = issued int8 "
                                      "multiply-adds / the int8 peak"}
            out["exgemv"] = gv
            out["exgemm"] = gm
            out["extrsv"] = blas23["extrsv"]
        if host_api:
            out["host_api"] = host_api
        if host0 is not None:
            base, parity = cpu_baseline(args.op, host0, args.fpe, ee, result)
            out["cpu_baseline"] = base
            out["cpu_baseline_at_cgroup_quota_Gelems"] = base["value_at_cgroup_quota"]
            out["bit_exact_vs_cpu"] = bool(parity["limbs_equal"] and parity["double_equal_exact_rounding"] and
                                           parity["double_equal_reference_rounding"])
            out["bit_exact_detail"] = parity
        print(json.dumps(out), flush=True)
    if comm is not None:
        comm.destroy()
    if dist is not None and dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
