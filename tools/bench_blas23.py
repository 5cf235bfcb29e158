#!/usr/bin/env python3
"""Times ExGEMV (BASELINE config 4: m=n=32768, 'N', column-major) and ExGEMM (config 5 shape per GPU) with
inputs resident in HBM.  usage: python tools/bench_blas23.py [gemv_log2n] [gemm_n] [gemm_rows]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import exblas_amd as ex

lg = int(sys.argv[1]) if len(sys.argv) > 1 else 15
gn = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
grows = int(sys.argv[3]) if len(sys.argv) > 3 else gn
out = {}


def timeit(fn, reps):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


m = n = 1 << lg
a = ex.gen_dev("fpuniform", m * n, 1, 10.0, 0.0)
x = ex.gen_dev("fpuniform", n, 2, 10.0, 0.0)
y = ex.gen_dev("fpuniform", m, 3, 10.0, 0.0)
bytes_alg = 8.0 * (m * n + n + 2 * m)
for trans in ("N", "T"):
    for fpe, ee in ((8, True), (4, True), (4, False), (0, False), (1, False)):
        if fpe == 0 and lg > 13 and trans == "N":
            reps = 1
        else:
            reps = 5
        ms = timeit(lambda: ex.exgemv_dev(trans, m, n, 1.0, a, m, x, 1.0, y.clone(), fpe, ee), reps)
        out[f"exgemv_{trans}_m{m}_fpe{fpe}{'ee' if ee else ''}"] = {"ms": ms, "GBs": bytes_alg / ms / 1e6,
                                                                    "frac_hbm_peak": bytes_alg / ms / 1e6 / 8000}
        print(trans, fpe, ee, f"{ms:.3f} ms  {bytes_alg / ms / 1e6:.0f} GB/s", flush=True)
del a
A = ex.gen_dev("fpuniform", grows * gn, 4, 10.0, 0.0)
B = ex.gen_dev("fpuniform", gn * gn, 5, 10.0, 0.0)
C = torch.zeros(grows * gn, dtype=torch.float64, device="cuda")
for fpe, ee in ((8, True), (4, True), (3, False), (0, False)):
    ms = timeit(lambda: ex.exgemm_dev("N", "N", grows, gn, gn, 1.0, A, gn, B, gn, 1.0, C, gn, fpe, ee), 1)
    fl = 2.0 * grows * gn * gn
    out[f"exgemm_{grows}x{gn}x{gn}_fpe{fpe}{'ee' if ee else ''}"] = {"ms": ms, "GFLOPs": fl / ms / 1e6}
    print("gemm", fpe, ee, f"{ms:.2f} ms  {fl / ms / 1e6:.1f} GFLOP/s (2mnk)", flush=True)
print(json.dumps(out))
