#!/usr/bin/env python3
"""Per-rank legs of the multi-GPU calls on ONE GPU: what a rank of a G-GPU job computes (its share of the rows), and what
the collective calls cost to issue through a one-rank RCCL communicator (EXBLAS_COMM_FORCE=1 makes the sharded calls post
their broadcasts / all-gathers although there is one rank: launch + in-place self-copy, NO link traffic).  Feeds the
time model of DESIGN.md section 7; the link terms of that model are specification figures, not measured here.
usage: EXBLAS_COMM_FORCE=1 python tools/comm_legs.py"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("EXBLAS_COMM_FORCE", "1")
import torch
import exblas_amd as ex

lib = ex.load_library()
lib.exblas_hip_init(-1)
comm = ex.Comm.rccl(ex.Comm.unique_id(), 0, 1)


def timeit(fn, reps=5, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


out = {}
N = 8192
B = ex.gen_dev("fpuniform", N * N, 15, 10.0, 0.0)
for G in (1, 2, 4, 8):
    rows = N // G
    A = ex.gen_dev("fpuniform", rows * N, 14, 10.0, 0.0)
    C = ex.gen_dev("fpuniform", rows * N, 18, 10.0, 0.0)
    for _ in range(10):
        ex.exgemm_dev("N", "N", rows, N, N, 1.0, A, N, B, N, 1.0, C, N, 8, True)
    t = timeit(lambda: ex.exgemm_dev("N", "N", rows, N, N, 1.0, A, N, B, N, 1.0, C, N, 8, True))
    out[f"exgemm_rank_share_G{G}_rows{rows}_ms"] = t
    print(f"ExGEMM 8192^3, a rank's share at G={G}: {rows} rows of A and C, all of B: {t:.3f} ms", flush=True)
    del A, C
# the collective legs of the gathered form, issued on one rank (no link traffic): B broadcast 512 MiB + C all-gather 512 MiB
A = ex.gen_dev("fpuniform", N * N, 14, 10.0, 0.0)
C = ex.gen_dev("fpuniform", N * N, 18, 10.0, 0.0)
t0 = timeit(lambda: ex.exgemm_sharded(comm, N, N, N, 1.0, A, B, 1.0, C, 8, True, b_root=-1, gather=False), 3)
t1 = timeit(lambda: ex.exgemm_sharded(comm, N, N, N, 1.0, A, B, 1.0, C, 8, True, b_root=0, gather=True), 3)
out["exgemm_1rank_nogather_ms"], out["exgemm_1rank_bcast_gather_forced_ms"] = t0, t1
print(f"one-rank communicator, 8192^3: C left sharded {t0:.3f} ms; B broadcast + C all-gather posted (self) {t1:.3f} ms", flush=True)
del A, C, B
# ExSUM / ExDOT: a rank's shard of ONE 2^28 vector at G ranks, step = kernel + finalize + 576-byte all-reduce + finalize
for G in (1, 2, 4, 8):
    n = (1 << 28) // G
    xs = [ex.gen_dev("ill_cond", n, 1 + i, 1e32) for i in range(4)]
    ys = [ex.gen_dev("ill_cond", n, 11 + i, 1e32) for i in range(4)]
    rec = ex.new_record_buffer()
    for name, fn_plain, fn_comm in (
            ("exsum", lambda i: ex.exsum_dev(xs[i % 4], 8, True, out=rec), lambda i: ex.exsum_allreduce(comm, xs[i % 4], 8, True, out=rec)),
            ("exdot", lambda i: ex.exdot_dev(xs[i % 4], ys[i % 4], 8, True, out=rec), lambda i: ex.exdot_allreduce(comm, xs[i % 4], ys[i % 4], 8, True, out=rec))):
        for tag, fn in (("plain", fn_plain), ("allreduce_1rank", fn_comm)):
            for i in range(300):
                fn(i)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(200):
                fn(i)
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / 200 * 1e3
            out[f"{name}_shard_G{G}_{tag}_us"] = us
            print(f"{name} n=2^28/{G} per rank, {tag}: {us:.1f} us per call (unpipelined, one stream)", flush=True)
    del xs, ys
print(json.dumps(out))
comm.destroy()
