#!/usr/bin/env python3
"""ExGEMM timing by path: automatic (0: residues for min(m, n) >= 192), int8 residues (4), int8 digit slices (2),
fp64 slices on MFMA-F64 (3), scalar kernel (1).
usage: python tools/bench_gemm.py n [rows] [paths e.g. 0,3,1] [kind p0 p1]"""
import ctypes as C
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import exblas_amd as ex
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
rows = int(sys.argv[2]) if len(sys.argv) > 2 else n
paths = [int(v) for v in (sys.argv[3] if len(sys.argv) > 3 else "0,3").split(",")]
kind = sys.argv[4] if len(sys.argv) > 4 else "fpuniform"
p0 = float(sys.argv[5]) if len(sys.argv) > 5 else 10.0
p1 = float(sys.argv[6]) if len(sys.argv) > 6 else 0.0
rounds = int(sys.argv[7]) if len(sys.argv) > 7 else 3
lib = ex.load_library()
A = ex.gen_dev(kind, rows * n, 4, p0, p1)
B = ex.gen_dev(kind, n * n, 5, p0, p1)
res = {}
for rnd in range(rounds):
    for path in paths:
        lib.exblas_set_gemm_path(path)
        Cm = torch.zeros(rows * n, dtype=torch.float64, device="cuda")
        reps = 1 if path == 1 else 3
        ex.exgemm_dev("N", "N", rows, n, n, 1.0, A, n, B, n, 0.0, Cm, n, 8, True)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            ex.exgemm_dev("N", "N", rows, n, n, 1.0, A, n, B, n, 0.0, Cm, n, 8, True)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        fl = 2.0 * rows * n * n
        v = (C.c_int * 8)()
        lib.exblas_last_gemm_info(v)
        key = f"path{path}"
        res.setdefault(key, []).append(ms)
        prods = v[3] if v[0] == 4 else v[1] * v[2]
        what = f"bits={v[1]}+{v[2]} moduli={v[3]}" if v[0] == 4 else f"slices={v[1]}x{v[2]}"
        print(key, f"{ms:.3f} ms, {fl/ms/1e9:.2f} TFLOP/s (2mnk), impl={v[0]} {what}, "
                   f"issued {fl*prods/ms/1e9:.1f} Top/s, checksum {float(Cm.sum()):.6e}", flush=True)
lib.exblas_set_gemm_path(0)
print(json.dumps({k: min(v) for k, v in res.items()}))
