#!/usr/bin/env python3
"""Step time (streaming kernel + finalize) vs the number of global group accumulators."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import exblas_amd as ex
n = 1 << 28
lib = ex.load_library()
x = ex.gen_dev("ill_cond", n, 1, 1e32)
rec = ex.new_record_buffer()
groups = [4, 8, 16, 32, 64]
times = {g: [] for g in groups}
for r in range(8):
    for g in groups:
        lib.exblas_set_tuning(-1, g, -1)
        for _ in range(3):
            ex.exsum_dev(x, 8, True, out=rec)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            ex.exsum_dev(x, 8, True, out=rec)
        e1.record(); torch.cuda.synchronize()
        if r:
            times[g].append(e0.elapsed_time(e1) / 20)
for g in groups:
    print(f"ngroups {g:3d}: step {statistics.median(times[g]):.4f} ms (min {min(times[g]):.4f})")
lib.exblas_set_tuning(-1, 32, -1)
