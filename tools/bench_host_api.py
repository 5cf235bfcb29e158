#!/usr/bin/env python3
"""PCIe-inclusive rate of the reference-style host-pointer API (H2D copy per call, like gpu:ExSUM.cpp:126)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import exblas_amd as ex
from oracle import pyoracle as O
n = 1 << (int(sys.argv[1]) if len(sys.argv) > 1 else 28)
a = O.gen("ill_cond", n, 1, 1e32)
ex.exsum(1024, a, 1, 0, 8, True)
for _ in range(3):
    t0 = time.perf_counter(); r = ex.exsum(n, a, 1, 0, 8, True); dt = time.perf_counter() - t0
    print(f"exsum host API n={n}: {dt*1e3:.1f} ms  {n/dt/1e9:.2f} Gelem/s  ({n*8/dt/1e9:.1f} GB/s over PCIe, pageable memory)  result {r!r}")
