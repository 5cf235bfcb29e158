#!/usr/bin/env python3
"""ExGEMM timing: scalar kernel vs MFMA-F64 slice path (and its tuning variants).
usage: python tools/bench_gemm.py n [rows] [variants e.g. 0,1]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import exblas_amd as ex
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
rows = int(sys.argv[2]) if len(sys.argv) > 2 else n
variants = [int(v) for v in (sys.argv[3] if len(sys.argv) > 3 else "0").split(",")]
lib = ex.load_library()
A = ex.gen_dev("fpuniform", rows * n, 4, 10.0, 0.0)
B = ex.gen_dev("fpuniform", n * n, 5, 10.0, 0.0)
res = {}
cfgs = [(0, v) for v in variants] + ([(1, 0)] if n <= 2048 else [])
for rnd in range(3):
    for path, var in cfgs:
        lib.exblas_set_gemm_path(path)
        lib.exblas_set_tuning(-1, -1, var)
        C = torch.zeros(rows * n, dtype=torch.float64, device="cuda")
        reps = 2 if path == 0 else 1
        ex.exgemm_dev("N", "N", rows, n, n, 1.0, A, n, B, n, 0.0, C, n, 8, True)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            ex.exgemm_dev("N", "N", rows, n, n, 1.0, A, n, B, n, 0.0, C, n, 8, True)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        fl = 2.0 * rows * n * n
        s = lib.exblas_last_gemm_slices()
        key = f"path{path}_v{var}"
        res.setdefault(key, []).append(ms)
        print(key, f"{ms:.2f} ms, {fl/ms/1e9:.2f} TFLOP/s (2mnk), slices={s}, mfma {fl*s*s/ms/1e9 if s else 0:.1f} TF, checksum {float(C.sum()):.6e}", flush=True)
lib.exblas_set_tuning(-1, -1, 0); lib.exblas_set_gemm_path(0)
print(json.dumps({k: min(v) for k, v in res.items()}))
