#!/usr/bin/env python3
"""Plain-read ceilings of the box for the one-stream (ExSUM) and two-stream (ExDOT) access patterns."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import exblas_amd as ex
n = 1 << 28
lib = ex.load_library()
x = ex.gen_dev("ill_cond", n, 1, 1e32); y = ex.gen_dev("ill_cond", n, 2, 1e32)
sink = torch.zeros(1, dtype=torch.float64, device="cuda")
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for bpc in (2, 4, 8, 16, 32):
    for _ in range(3):
        lib.exblas_stream_read2_dev(C.c_void_p(x.data_ptr()), C.c_void_p(y.data_ptr()), n, bpc, st, C.c_void_p(sink.data_ptr()))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        lib.exblas_stream_read2_dev(C.c_void_p(x.data_ptr()), C.c_void_p(y.data_ptr()), n, bpc, st, C.c_void_p(sink.data_ptr()))
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print(f"plain 2-stream dot bpc{bpc:2d}: {ms:.4f} ms  {n*16/ms/1e6:.0f} GB/s")
