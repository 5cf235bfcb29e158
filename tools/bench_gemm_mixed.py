"""ExGEMM 4096^3 with operands that need 2 or 3 digits each (integers below 2^15 vs full mantissas): python tools/bench_gemm_mixed.py"""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import exblas_amd as ex
lib = ex.load_library(); lib.exblas_hip_init(-1)
n = 4096
g = torch.Generator(device="cuda").manual_seed(1)
small = torch.randint(-30000, 30001, (n * n,), device="cuda", generator=g).double()
full = ex.gen_dev("fpuniform", n * n, 5, 10.0, 0.0)
C = torch.zeros(n * n, dtype=torch.float64, device="cuda")
def t(a, b, label):
    for _ in range(2): ex.exgemm_dev("N", "N", n, n, n, 1.0, a, n, b, n, 0.0, C, n, 8, True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3): ex.exgemm_dev("N", "N", n, n, n, 1.0, a, n, b, n, 0.0, C, n, 8, True)
    e1.record(); torch.cuda.synchronize()
    print(label, f"{e0.elapsed_time(e1)/3:.2f} ms slices {lib.exblas_last_gemm_slices()}", flush=True)
t(full, full, "3x3"); t(small, full, "2x3"); t(full, small, "3x2"); t(small, small, "2x2")
