"""Strided ExSUM / ExDOT (inca = 2, 3; the reference's slow path): python tools/bench_strided.py [log2n]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import exblas_amd as ex
log2n = int(sys.argv[1]) if len(sys.argv) > 1 else 26
n = 1 << log2n
ex.load_library().exblas_hip_init(-1)
for inc in (2, 3, 8):
    a = ex.gen_dev("ill_cond", n * inc, 1, 1e32, 0)
    b = ex.gen_dev("ill_cond", n * inc, 2, 1e32, 0)
    for name, fn in (("exsum", lambda: ex.exsum_dev(a, 8, True, n=n, inca=inc)),
                     ("exdot", lambda: ex.exdot_dev(a, b, 8, True, n=n, incx=inc, incy=inc))):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            fn()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        print(f"{name} inc={inc}: {ms:.3f} ms, {n / ms / 1e6:.1f} Gelem/s", flush=True)
