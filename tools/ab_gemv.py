"""gemv N/T 32768^2 timing for A/B runs (EXBLAS_AMD_LIB selects the build): python tools/ab_gemv.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import exblas_amd as ex
ex.load_library().exblas_hip_init(-1)
m = k = 32768
a = ex.gen_dev("fpuniform", m * k, 1, 10.0, 0.0)
x = ex.gen_dev("fpuniform", k, 2, 10.0, 0.0)
y = ex.gen_dev("fpuniform", m, 3, 10.0, 0.0)
out = []
for trans in ("N", "T"):
    yy = y.clone()
    for _ in range(20):
        ex.exgemv_dev(trans, m, k, 1.0, a, m, x, 0.0, yy, 8, True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(40):
        ex.exgemv_dev(trans, m, k, 1.0, a, m, x, 0.0, yy, 8, True)
    e1.record(); torch.cuda.synchronize()
    out.append(f"{trans} {e0.elapsed_time(e1) / 40:.4f} ms sum {ex.read_record(ex.exsum_dev(yy)).exact.hex()}")
print(" | ".join(out))
