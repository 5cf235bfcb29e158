"""ExGEMM 8192^3 for every transpose combination, beta = 0 and beta = 1: python tools/bench_gemm_trans.py [n]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import exblas_amd as ex
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
ex.load_library().exblas_hip_init(-1)
A = ex.gen_dev("fpuniform", n * n, 4, 10.0, 0.0)
B = ex.gen_dev("fpuniform", n * n, 5, 10.0, 0.0)
for ta in "NT":
    for tb in "NT":
        for beta in (0.0, 1.0):
            C = torch.ones(n * n, dtype=torch.float64, device="cuda")
            ex.exgemm_dev(ta, tb, n, n, n, 1.0, A, n, B, n, beta, C, n, 8, True)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                ex.exgemm_dev(ta, tb, n, n, n, 1.0, A, n, B, n, beta, C, n, 8, True)
            e1.record(); torch.cuda.synchronize()
            print(f"{ta}{tb} beta={beta:g}: {e0.elapsed_time(e1) / 3:.3f} ms", flush=True)
