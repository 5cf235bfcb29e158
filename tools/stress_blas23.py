"""Randomised soak of ExGEMV / ExGEMM against the oracle: python tools/stress_blas23.py [iterations] [seed].
Random shapes, transposes, alpha/beta, leading dimensions, strides and variants; bits must equal the oracle's."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa: F401
import exblas_amd as ex
from oracle import pyoracle as o

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
lib = ex.load_library()
lib.exblas_hip_init(-1)
gv_variants = [(0, False), (2, False), (3, False), (4, False), (6, False), (8, False), (4, True), (6, True), (8, True)]
gm_variants = [(0, False), (3, False), (4, False), (8, False), (4, True), (6, True), (8, True)]
kinds = [("fpuniform", 10, 0), ("fpuniform_signed", 40, 20), ("fpuniform_signed", 200, 100), ("lognormal", 0.0, 2.0),
         ("ill_cond", 1e16, 0)]
scalars = [0.0, 1.0, -1.0, 2.5, -0.3, 1e-3]
bits = lambda v: np.ascontiguousarray(v, dtype=np.float64).view(np.int64)  # noqa: E731
t0 = time.time()
bad = 0
fast = 0
for it in range(iters):
    kind, p0, p1 = kinds[int(rng.integers(0, len(kinds)))]
    seed = int(rng.integers(1, 1 << 30))
    if it % 3 != 2:
        m, n = int(rng.integers(1, 900)), int(rng.integers(1, 900))
        trans = str(rng.choice(["N", "T"]))
        lda = m + int(rng.integers(0, 4))
        incx, incy = int(rng.choice([1, 1, 2])), int(rng.choice([1, 1, 3]))
        offa, offx, offy = int(rng.integers(0, 3)), int(rng.integers(0, 3)), int(rng.integers(0, 3))
        alpha, beta = float(rng.choice(scalars[1:])), float(rng.choice(scalars))
        rows, inner = (n, m) if trans == "T" else (m, n)
        a = o.gen(kind, lda * n + offa, seed, p0, p1)
        x = o.gen(kind, (inner - 1) * incx + 1 + offx, seed + 1, p0, p1)
        y0 = o.gen(kind, (rows - 1) * incy + 1 + offy, seed + 2, p0, p1)
        fpe, ee = gv_variants[int(rng.integers(0, len(gv_variants)))]
        want = o.exgemv(trans, m, n, alpha, a, lda, x, beta, y0, 0, incx=incx, incy=incy, offa=offa, offx=offx, offy=offy)
        y = y0.copy()
        ex.exgemv(trans, m, n, alpha, a, lda, offa, x, incx, offx, beta, y, incy, offy, fpe, ee)
        ok = (bits(y) == bits(want)).all()
        desc = f"gemv {trans} m={m} n={n} lda={lda} inc={incx},{incy} off={offa},{offx},{offy} a={alpha} b={beta}"
    else:
        m, n, k = int(rng.integers(1, 200)), int(rng.integers(1, 200)), int(rng.integers(1, 400))
        ta, tb = str(rng.choice(["N", "T"])), str(rng.choice(["N", "T"]))
        lda = (m if ta == "T" else k) + int(rng.integers(0, 3))
        ldb = (k if tb == "T" else n) + int(rng.integers(0, 3))
        ldc = n + int(rng.integers(0, 3))
        alpha, beta = float(rng.choice(scalars[1:])), float(rng.choice(scalars))
        a = o.gen(kind, (k if ta == "T" else m) * lda, seed, p0, p1)
        b = o.gen(kind, (n if tb == "T" else k) * ldb, seed + 1, p0, p1)
        c0 = o.gen(kind, m * ldc, seed + 2, p0, p1)
        fpe, ee = gm_variants[int(rng.integers(0, len(gm_variants)))]
        want = o.exgemm(ta, tb, m, n, k, alpha, a, lda, b, ldb, beta, c0, ldc, 0)
        c = c0.copy()
        ex.exgemm(ta, tb, m, n, k, alpha, a, lda, b, ldb, beta, c, ldc, fpe, ee)
        fast += lib.exblas_last_gemm_slices() >= 2
        # the padding columns of C (ldc > n) must stay untouched, the rest must match
        ok = (bits(c) == bits(want)).all()
        desc = f"gemm {ta}{tb} m={m} n={n} k={k} ld={lda},{ldb},{ldc} a={alpha} b={beta} slices={lib.exblas_last_gemm_slices()}"
    if not ok:
        bad += 1
        print(f"MISMATCH it={it} {desc} kind={kind} fpe={fpe} ee={ee}", flush=True)
    if it % 40 == 0:
        print(f"it {it}: {desc} kind={kind} fpe={fpe}{'ee' if ee else ''} [{time.time() - t0:.0f} s]", flush=True)
print(f"done: {iters} cases, {bad} mismatches, {fast} gemm cases on the MFMA path, {time.time() - t0:.0f} s")
sys.exit(1 if bad else 0)
