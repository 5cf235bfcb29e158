#!/usr/bin/env python3
"""A/B of exblas_set_tuning variants for ExGEMM in one process, interleaved: python tools/ab_gemm_variant.py n variants [kind p0 p1]
(int8 path: variant 0 = LDS-DMA staged pass body, 1 = register-staged)"""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import exblas_amd as ex
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
variants = [int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "0,1").split(",")]
kind = sys.argv[3] if len(sys.argv) > 3 else "fpuniform"
p0 = float(sys.argv[4]) if len(sys.argv) > 4 else 10.0
p1 = float(sys.argv[5]) if len(sys.argv) > 5 else 0.0
lib = ex.load_library()
A = ex.gen_dev(kind, n * n, 4, p0, p1)
B = ex.gen_dev(kind, n * n, 5, p0, p1)
times = {v: [] for v in variants}
ref = None
for r in range(5):
    for v in variants:
        lib.exblas_set_tuning(-1, -1, v)
        C = torch.zeros(n * n, dtype=torch.float64, device="cuda")
        ex.exgemm_dev("N", "N", n, n, n, 1.0, A, n, B, n, 0.0, C, n, 8, True)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            ex.exgemm_dev("N", "N", n, n, n, 1.0, A, n, B, n, 0.0, C, n, 8, True)
        e1.record(); torch.cuda.synchronize()
        if ref is None:
            ref = C.clone()
        assert torch.equal(ref.view(torch.int64), C.view(torch.int64)), v
        if r:
            times[v].append(e0.elapsed_time(e1) / 3)
for v in variants:
    print(f"gemm n={n} {kind} variant {v}: median {statistics.median(times[v]):.3f} ms  min {min(times[v]):.3f} ms", flush=True)
lib.exblas_set_tuning(-1, -1, 0)
