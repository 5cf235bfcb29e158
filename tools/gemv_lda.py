"""ExGEMV 32768^2 against the leading dimension: python tools/gemv_lda.py  (is the power-of-two column stride a problem?)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import exblas_amd as ex
m = n = 32768
ex.load_library().exblas_hip_init(-1)
x = ex.gen_dev("fpuniform", n, 12, 10.0, 0.0)
y = ex.gen_dev("fpuniform", m, 13, 10.0, 0.0)
for pad in (0, 16, 512, 32768):
    lda = m + pad
    a = ex.gen_dev("fpuniform", lda * n, 11, 10.0, 0.0)
    for tr in ("N", "T"):
        for _ in range(3):
            ex.exgemv_dev(tr, m, n, 1.0, a, lda, x, 1.0, y, 8, True)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            ex.exgemv_dev(tr, m, n, 1.0, a, lda, x, 1.0, y, 8, True)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        print(f"lda = m + {pad:5d}  '{tr}': {ms:.4f} ms, {8.0 * m * n / ms / 1e6:.0f} GB/s", flush=True)
    del a
