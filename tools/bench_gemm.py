#!/usr/bin/env python3
"""ExGEMM timing: scalar kernel vs MFMA-F64 slice path.  usage: python tools/bench_gemm.py n [rows]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import exblas_amd as ex
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
rows = int(sys.argv[2]) if len(sys.argv) > 2 else n
lib = ex.load_library()
A = ex.gen_dev("fpuniform", rows * n, 4, 10.0, 0.0)
B = ex.gen_dev("fpuniform", n * n, 5, 10.0, 0.0)
res = {}
for path, fpe, ee, reps in ((0, 8, True, 3), (1, 8, True, 1)):
    if path == 1 and n > 4096:
        continue
    lib.exblas_set_gemm_path(path)
    C = torch.zeros(rows * n, dtype=torch.float64, device="cuda")
    ex.exgemm_dev("N", "N", rows, n, n, 1.0, A, n, B, n, 1.0, C, n, fpe, ee)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        ex.exgemm_dev("N", "N", rows, n, n, 1.0, A, n, B, n, 0.0, C, n, fpe, ee)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    fl = 2.0 * rows * n * n
    s = lib.exblas_last_gemm_slices()
    res[f"path{path}"] = {"ms": ms, "TFLOPs_2mnk": fl / ms / 1e9, "slices": s,
                          "mfma_TFLOPs": (fl * s * s / ms / 1e9) if s else None, "checksum": float(C.sum())}
    print(path, f"{ms:.2f} ms, {fl/ms/1e9:.2f} TFLOP/s (2mnk), slices={s}", flush=True)
print(json.dumps(res))
