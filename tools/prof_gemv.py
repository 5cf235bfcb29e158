import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import exblas_amd as ex
m = n = 1 << 15
a = ex.gen_dev("fpuniform", m * n, 1, 10.0, 0.0)
x = ex.gen_dev("fpuniform", n, 2, 10.0, 0.0)
y = ex.gen_dev("fpuniform", m, 3, 10.0, 0.0)
for trans in ("N", "T"):
    for _ in range(6):
        ex.exgemv_dev(trans, m, n, 1.0, a, m, x, 1.0, y, 8, True)
torch.cuda.synchronize()
