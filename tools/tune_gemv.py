#!/usr/bin/env python3
"""A/B ExGEMV 'N'/'T' variants in one process (interleaved).  usage: python tools/tune_gemv.py [log2n] [variants]"""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import exblas_amd as ex
lg = int(sys.argv[1]) if len(sys.argv) > 1 else 15
variants = [int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "0,1,2,3,4,5").split(",")]
bpcs = [int(v) for v in (sys.argv[3] if len(sys.argv) > 3 else "8").split(",")]
m = n = 1 << lg
lib = ex.load_library()
a = ex.gen_dev("fpuniform", m * n, 1, 10.0, 0.0)
x = ex.gen_dev("fpuniform", n, 2, 10.0, 0.0)
y = ex.gen_dev("fpuniform", m, 3, 10.0, 0.0)
bytes_alg = 8.0 * (m * n + n + 2 * m)
for trans in ("N", "T"):
    cfgs = [(v, b) for v in variants for b in bpcs]
    times = {c: [] for c in cfgs}
    ref = None
    for r in range(6):
        for c in cfgs:
            v = c
            lib.exblas_set_tuning(c[1], -1, c[0])
            yy = y.clone()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                ex.exgemv_dev(trans, m, n, 1.0, a, m, x, 0.0, yy, 8, True)
            e1.record(); torch.cuda.synchronize()
            if ref is None:
                ref = yy.clone()
            assert torch.equal(ref.view(torch.int64), yy.view(torch.int64)), v
            if r:
                times[v].append(e0.elapsed_time(e1) / 3)
    for v in cfgs:
        med = statistics.median(times[v])
        print(f"gemv {trans} v{v}: {med:.3f} ms  {bytes_alg/med/1e6:.0f} GB/s", flush=True)
lib.exblas_set_tuning(8, -1, 0)
