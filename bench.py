#!/usr/bin/env python3
"""bench.py -- ExSUM / ExDOT throughput on MI355X with inputs resident in HBM.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over the rank's synthetic vector(s): the streaming kernel, the
finalize kernel and -- for N > 1 -- the 576-byte int64 all-reduce of the digit set (RCCL) plus the
re-finalize.  Weak scaling: every rank holds n elements (default 2^28) of a global vector of N*n
elements generated in place by the counter-based generator, so `value` = N*n*K / max-over-ranks time.

Prints ONE JSON line on rank 0 (see README / task contract), with `roofline` for the dominant
kernel (k_exsum resp. k_exdot: algorithmic 8 resp. 16 B/element over the HIP-event time of that
kernel alone) and `cpu_baseline` (N == 1 only: the reference's own compiled FPE+superaccumulator
core from oracle/_ref when it can be loaded, else our C port, on the host cores).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--prewarm-ms", type=float, default=400.0,
                    help="untimed clock ramp-up before the W warmup steps: the same kernel run back to back "
                         "(the chip needs a few hundred ms of load to reach its sustained clocks)")
    ap.add_argument("--op", default="exsum", choices=["exsum", "exdot"])
    ap.add_argument("--log2n", type=int, default=28, help="elements per GPU = 2^log2n")
    ap.add_argument("--kind", default="ill_cond")
    ap.add_argument("--p0", type=float, default=1e32)
    ap.add_argument("--p1", type=float, default=0.0)
    ap.add_argument("--fpe", type=int, default=8)
    ap.add_argument("--no-early-exit", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the secondary op (ExDOT) line items")
    ap.add_argument("--traffic-json", default=os.path.join(ROOT, "profiles", "hbm_traffic.json"),
                    help="per-launch HBM bytes from a separate rocprofv3 --pmc pass, if present")
    return ap.parse_args()


def timed_steps(ex, torch, dist, op, tensors, fpe, ee, steps, warmup, world, rec, prewarm_ms=0.0):
    """Returns (wall seconds for `steps` steps, mean ms of the streaming kernel alone)."""
    use_dist = world > 1 or (dist is not None and dist.is_initialized())
    # N > 1: the 576-byte all-reduce of step i overlaps the streaming kernel of step i+1 (records live in a small
    # ring; a step's second finalize is issued once its all-reduce has landed).  Every step is still carried to
    # its final rounded result inside the timed region (drain() before the closing synchronize).
    ring = [rec] + [ex.new_record_buffer() for _ in range(3)] if use_dist else [rec]
    pending = []
    state = {"i": 0, "last": rec}

    side = torch.cuda.Stream() if use_dist else None  # second finalize runs beside the next streaming kernel

    slot_free = {}  # record buffer -> event recorded on the side stream once its second finalize has been issued

    def retire(limit):
        while len(pending) > limit:
            work, r = pending.pop(0)
            with torch.cuda.stream(side):
                work.wait()
                ex.finalize_dev(r[ex.OUT_DIGITS:ex.OUT_DIGITS + ex.SET_WORDS], out=r)
                ev_done = torch.cuda.Event()
                ev_done.record()
            slot_free[r.data_ptr()] = ev_done
            state["last"] = r

    def one_step(e0=None, e1=None):
        r = ring[state["i"] % len(ring)]
        state["i"] += 1
        ev_free = slot_free.pop(r.data_ptr(), None)
        if ev_free is not None and not ev_free.query():
            # the buffer's previous occupant must be fully retired; it was issued four steps ago, so this is
            # practically always true already and the queue is spared a wait packet
            torch.cuda.current_stream().wait_event(ev_free)
        if e0 is not None:
            e0.record()
        if op == "exsum":
            ex.exsum_accumulate_dev(tensors[0], fpe, ee)
        else:
            ex.exdot_accumulate_dev(tensors[0], tensors[1], fpe, ee)
        if e1 is not None:
            e1.record()
        ex.finish_dev(out=r)
        if use_dist:
            work = dist.all_reduce(r[ex.OUT_DIGITS:ex.OUT_DIGITS + ex.SET_WORDS], op=dist.ReduceOp.SUM, async_op=True)
            pending.append((work, r))
            retire(2)
        else:
            state["last"] = r

    def drain():
        retire(0)
        if side is not None:
            torch.cuda.current_stream().wait_stream(side)

    # clock ramp-up: untimed steps for about prewarm_ms.  With several ranks the number of steps (= collectives) must
    # be the same everywhere, so the ranks vote after every batch whether to go on.
    t_pre = time.perf_counter()
    while prewarm_ms > 0:
        go = (time.perf_counter() - t_pre) * 1e3 < prewarm_ms
        if use_dist:
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
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        # every event is one more packet in the queue between two streaming kernels (~3 us each): the kernel time is
        # sampled on every fourth step instead of bracketing all of them
        if i % 4 and steps >= 8:
            one_step()
        else:
            one_step(*ev[i])
    drain()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    timed = [p for i, p in enumerate(ev) if not (i % 4 and steps >= 8)]
    kms = sum(a.elapsed_time(b) for a, b in timed) / max(len(timed), 1)
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


def cpu_baseline(op, host_arrays, fpe, ee, limbs_gpu):
    """Time the CPU path on this box's host cores, on the same vector(s); returns the JSON object.

    The cgroup of a GPU box grants fewer CPUs (cpu.max) than the affinity mask shows, but short runs may burst
    beyond the quota; the thread count is therefore swept (quota, 2x, 4x, ... up to the physical cores) and the
    FASTEST configuration is reported, with its thread count -- the most favourable number for the CPU."""
    import numpy as np
    from oracle import pyoracle as O
    quota = host_cores()
    affinity = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else quota
    n = host_arrays[0].size
    use_ref = op == "exsum" and O.ref() is not None
    counts = sorted({max(1, min(affinity, c)) for c in (quota, 2 * quota, 4 * quota, affinity // 2)})

    def run(nt):
        if use_ref:
            return O.ref_exsum(host_arrays[0], fpe, ee, nthreads=nt, limbs=True)[1]
        if op == "exsum":
            return O.exsum_omp(host_arrays[0], fpe, ee, nt, limbs=True)[1]
        return O.exdot_omp(host_arrays[0], host_arrays[1], fpe, ee, nt, limbs=True)[1]

    best, best_nt, limbs, tried = None, quota, None, {}
    t_start = time.perf_counter()
    for nt in counts:
        for _ in range(3):
            if time.perf_counter() - t_start > 25.0 and best is not None:
                break
            t0 = time.perf_counter()
            limbs = run(nt)
            dt = time.perf_counter() - t0
            tried[nt] = min(dt, tried.get(nt, dt))
            if best is None or dt < best:
                best, best_nt = dt, nt
    ok = bool((np.asarray(limbs) == np.asarray(limbs_gpu)).all())
    impl = ("reference FPExpansionVect+Superaccumulator (oracle/_ref, -O1 -mavx2 -mfma) under our OpenMP slice driver"
            if use_ref else "oracle/exblas_oracle.c OpenMP port")
    sweep = ", ".join(f"{k} thr: {n / v / 1e9:.2f}" for k, v in sorted(tried.items()))
    return {
        "value": n / best / 1e9, "unit": "Gelem/s", "cores": best_nt,
        "kind": "reference" if use_ref else "port",
        "sample": f"full workload, n={n}, {impl}, fpe={fpe} early_exit={ee}; cgroup cpu quota {quota}, affinity "
                  f"{affinity}; best of 3 per thread count, Gelem/s by threads: {sweep}",
        "seconds": best,
    }, ok


def load_traffic(path, kernel):
    """Per-launch HBM bytes of `kernel` from the separate rocprofv3 --pmc passes (profiles/hbm_traffic.json)."""
    try:
        return json.load(open(path)).get(kernel)
    except Exception:  # noqa: BLE001
        return None


def bench_blas23(ex, torch, world, rank):
    """ExGEMV m=n=32768 'N' column-major per GPU (config 4; replicas for N > 1) and ExGEMM n=8192 row-sharded
    over the ranks (config 5: each rank owns 8192/N rows of A and C, B replicated, no collective)."""
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

    m = n = 32768
    a = ex.gen_dev("fpuniform", m * n, 11, 10.0, 0.0)
    x = ex.gen_dev("fpuniform", n, 12, 10.0, 0.0)
    y = ex.gen_dev("fpuniform", m, 13, 10.0, 0.0)
    ms = timeit(lambda: ex.exgemv_dev("N", m, n, 1.0, a, m, x, 1.0, y, 8, True), 10)
    gv = {"workload": "ExGEMV 'N' m=n=32768 fp64 column-major alpha=beta=1, fpe=8 early_exit, per GPU", "ms": ms,
          "bytes": 8.0 * (m * n + n + 2 * m)}
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
          "rows_on_integer_path": ex.load_library().exblas_extrsv_last_slow_rows(),
          "finite": bool(torch.isfinite(xs).all())}
    del a, d, b, xs
    N = 8192
    r0, r1 = ex.row_block(N, rank, world)
    A = ex.gen_dev("fpuniform", N * N, 14, 10.0, 0.0, first=r0 * N, count=(r1 - r0) * N)
    B = ex.gen_dev("fpuniform", N * N, 15, 10.0, 0.0)
    C = torch.zeros((r1 - r0) * N, dtype=torch.float64, device="cuda")
    ms = timeit(lambda: ex.exgemm_rows(r1 - r0, N, N, 1.0, A, B, 0.0, C, 8, True), 2)
    gm = {"workload": f"ExGEMM n=8192 fp64 row-major alpha=1, rows sharded over {world} GPU(s), "
                      "MFMA-F64 slice path", "ms": ms, "flop_total": 2.0 * N * N * N,
          "slices": ex.load_library().exblas_last_gemm_slices()}
    return {"exgemv": gv, "exgemm": gm, "extrsv": tv}


def main():
    args = parse()
    import torch
    import exblas_amd as ex

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    force_dist = os.environ.get("EXBLAS_BENCH_FORCE_DIST") == "1"  # rehearse the RCCL path with a 1-rank group
    if world > 1 or force_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if force_dist and world == 1:  # stand-alone rehearsal without a launcher: a group of one
            os.environ.setdefault("MASTER_PORT", "29517")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        # EXBLAS_BENCH_BACKEND=gloo rehearses the N > 1 control flow on a box with fewer GPUs than ranks (ranks then
        # share the devices round-robin); the measured configuration is always nccl (= RCCL), one GPU per rank
        backend = os.environ.get("EXBLAS_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            torch.cuda.set_device(local_rank % torch.cuda.device_count())
            dist.init_process_group(backend)
    else:
        torch.cuda.set_device(0)
    assert world == args.gpus or world == 1, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    ex.load_library().exblas_hip_init(-1)

    n = 1 << args.log2n
    n_total = n * world
    ee = not args.no_early_exit
    first = rank * n
    x = ex.gen_dev(args.kind, n, 1, args.p0, args.p1, first=first, count=n, n_total=n_total)
    tensors = [x]
    if args.op == "exdot":
        tensors.append(ex.gen_dev(args.kind, n, 2, args.p0, args.p1, first=first, count=n, n_total=n_total))
    rec = ex.new_record_buffer()
    bytes_per_elem = 8 if args.op == "exsum" else 16

    dt, kms = timed_steps(ex, torch, dist, args.op, tensors, args.fpe, ee, args.steps, args.warmup, world, rec,
                          args.prewarm_ms)
    if world > 1:
        t = torch.tensor([dt, kms], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt, kms = float(t[0]), float(t[1])
    result = ex.read_record(rec)

    # read-bandwidth probe of the box (plain streaming sum, same launch geometry)
    sink = torch.zeros(1, dtype=torch.float64, device="cuda")
    for _ in range(3):
        ex.stream_read_dev(x, sink)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        ex.stream_read_dev(x, sink)
    e1.record()
    torch.cuda.synchronize()
    probe_gbs = 10 * n * 8 / (e0.elapsed_time(e1) * 1e-3) / 1e9

    secondary = None
    if not args.no_secondary and args.op == "exsum":
        # ExDOT on the same shape (BASELINE config 2), reported beside the headline number
        y = ex.gen_dev(args.kind, n, 2, args.p0, args.p1, first=first, count=n, n_total=n_total)
        rec2 = ex.new_record_buffer()
        ddt, dkms = timed_steps(ex, torch, dist, "exdot", [x, y], args.fpe, ee, args.steps, args.warmup, world, rec2,
                                args.prewarm_ms)
        if world > 1:
            t = torch.tensor([ddt, dkms], dtype=torch.float64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            ddt, dkms = float(t[0]), float(t[1])
        dach = n * 16 / (dkms * 1e-3) / 1e9
        # plain (inexact) fp64 dot with the same two-stream access pattern: the box's ceiling for this kernel
        import ctypes as C
        lib = ex.load_library()
        stp = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        args2 = (C.c_void_p(x.data_ptr()), C.c_void_p(y.data_ptr()), n, 0, stp, C.c_void_p(sink.data_ptr()))
        for _ in range(3):
            lib.exblas_stream_read2_dev(*args2)
        p0_, p1_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        p0_.record()
        for _ in range(10):
            lib.exblas_stream_read2_dev(*args2)
        p1_.record()
        torch.cuda.synchronize()
        probe2_gbs = 10 * n * 16 / (p0_.elapsed_time(p1_) * 1e-3) / 1e9
        secondary = {"metric": "ExDOT fp64 Gelem/s", "value": n_total * args.steps / ddt / 1e9, "unit": "Gelem/s",
                     "ms_per_step": ddt / args.steps * 1e3,
                     "roofline": {"bound": "hbm", "achieved": dach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                  "frac": dach / HBM_PEAK_GBS, "traffic": load_traffic(args.traffic_json, "k_exdot"),
                                  "kernel": "k_exdot", "measured_read_probe_GBs": probe2_gbs,
                                  "frac_of_probe": dach / probe2_gbs,
                                  "kernel_ms": dkms},
                     "result": ex.read_record(rec2).exact}
        del y

    # BASELINE configs 4 and 5 on the same box (kernel-chain time by HIP events; parity is covered by tests/)
    blas23 = None
    if not args.no_secondary and args.op == "exsum":
        blas23 = bench_blas23(ex, torch, world, rank)
        if world > 1:
            t = torch.tensor([blas23["exgemv"]["ms"], blas23["exgemm"]["ms"]], dtype=torch.float64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            blas23["exgemv"]["ms"], blas23["exgemm"]["ms"] = float(t[0]), float(t[1])

    out = None
    if rank == 0:
        achieved = n * bytes_per_elem / (kms * 1e-3) / 1e9
        traffic = load_traffic(args.traffic_json, f"k_{args.op}")
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
            "data": f"synthetic ({args.kind} p0={args.p0:g} p1={args.p1:g}, counter-based generator, seed 1)",
            "config": {"workload": f"Ex{args.op[2:].upper()} n=2^{args.log2n} fp64 {args.kind}"
                                   f"(c={args.p0:g}) per GPU, fpe={args.fpe} early_exit={ee}, "
                                   f"{world}xMI355X, inputs resident in HBM",
                       "elements_per_gpu": n, "fpe": args.fpe, "early_exit": ee,
                       "parallelism": f"shard{world}" if world > 1 else "single"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": f"k_{args.op}", "kernel_ms": kms,
                         "measured_read_probe_GBs": probe_gbs, "frac_of_probe": achieved / probe_gbs},
            "result": result.exact,
        }
        if secondary:
            out["exdot"] = secondary
        if blas23:
            gv, gm = blas23["exgemv"], blas23["exgemm"]
            gv["GBs"] = gv["bytes"] * world / (gv["ms"] * 1e-3) / 1e9
            gv["frac_hbm_peak"] = gv["bytes"] / (gv["ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS
            gm["TFLOPs_2mnk"] = gm["flop_total"] / (gm["ms"] * 1e-3) / 1e12
            # every product costs slices^2 MFMA-FMAs; f64 MFMA peak 78.6 TFLOP/s per GPU (AMD MI355X datasheet;
            # equals the fp64 vector rate on CDNA4 -- not listed in MI355X_MICROARCH.md)
            mf = gm["flop_total"] * gm["slices"] ** 2 / (gm["ms"] * 1e-3) / 1e12
            gm["roofline"] = {"bound": "mfma", "achieved": mf, "peak": 78.6 * world, "unit": "TFLOP/s",
                              "frac": mf / (78.6 * world), "traffic": None, "kernel": "k_gemm_mfma"}
            out["exgemv"] = gv
            out["exgemm"] = gm
            out["extrsv"] = blas23["extrsv"]
        if world == 1 and not args.no_cpu_baseline:
            host = [t.cpu().numpy() for t in tensors]
            base, ok = cpu_baseline(args.op, host, args.fpe, ee, result.canon)
            out["cpu_baseline"] = base
            out["bit_exact_vs_cpu"] = ok
        print(json.dumps(out), flush=True)
    if dist is not None and dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
