#!/usr/bin/env python3
"""Does the allocation history of the process change ExGEMV's time?  (bench.py measures it after ExSUM / ExDOT have
allocated and freed 16 GiB; tools/tune_gemv.py in a fresh process.)  usage: python tools/gemv_ctx.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import exblas_amd as ex
m = n = 32768
ex.load_library().exblas_hip_init(-1)

def timeit(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

def run(tag):
    a = ex.gen_dev("fpuniform", m * n, 11, 10.0, 0.0)
    x = ex.gen_dev("fpuniform", n, 12, 10.0, 0.0)
    y = ex.gen_dev("fpuniform", m, 13, 10.0, 0.0)
    for beta in (1.0, 0.0):
        tn = timeit(lambda: ex.exgemv_dev("N", m, n, 1.0, a, m, x, beta, y, 8, True))
        tt = timeit(lambda: ex.exgemv_dev("T", m, n, 1.0, a, m, x, beta, y, 8, True))
        print(f"{tag}: beta={beta} N {tn:.3f} ms  T {tt:.3f} ms  (a at {a.data_ptr():#x})", flush=True)

run("fresh process")
junk = [ex.gen_dev("ill_cond", 1 << 28, 1 + i, 1e32) for i in range(8)]
rec = ex.new_record_buffer()
for i in range(200):
    ex.exdot_dev(junk[i % 4], junk[4 + i % 4], 8, True, out=rec)
torch.cuda.synchronize()
del junk
torch.cuda.empty_cache()
run("after 16 GiB of ExDOT buffers + 200 ExDOT calls")
for i in range(3):
    run(f"again {i}")
