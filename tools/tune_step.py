#!/usr/bin/env python3
"""Per-step overhead (step time minus back-to-back kernel time) for ExSUM / ExDOT vs launch geometry."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import exblas_amd as ex
n = 1 << 28
lib = ex.load_library()
x = ex.gen_dev("ill_cond", n, 1, 1e32); y = ex.gen_dev("ill_cond", n, 2, 1e32)
rec = ex.new_record_buffer()
def t(fn, reps=30):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for op in ("exsum", "exdot"):
    for bpc in ((2, 4, 8) if op == "exsum" else (8, 16, 32)):
        for ng in (8, 32):
            lib.exblas_set_tuning(bpc, ng, 0)
            if op == "exsum":
                k = min(t(lambda: ex.exsum_accumulate_dev(x, 8, True)) for _ in range(3)); ex.finish_dev(out=rec)
                s = min(t(lambda: ex.exsum_dev(x, 8, True, out=rec)) for _ in range(3))
            else:
                k = min(t(lambda: ex.exdot_accumulate_dev(x, y, 8, True)) for _ in range(3)); ex.finish_dev(out=rec)
                s = min(t(lambda: ex.exdot_dev(x, y, 8, True, out=rec)) for _ in range(3))
            print(f"{op} bpc{bpc:2d} ngroups{ng:2d}: kernel-only loop {k:.4f} ms, step loop {s:.4f} ms, overhead {1e3*(s-k):.1f} us", flush=True)
