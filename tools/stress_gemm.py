"""Randomised soak of ExGEMM's int8 paths against the oracle: python tools/stress_gemm.py [iterations] [seed].
EXBLAS_GEMM_PATH=4 in the environment forces the residue path (blas3_crt.hip) at every shape, =2 the digit slices.
Random shapes (ragged tiles, k across the 8192-per-pass boundary), transposes, leading dimensions, alpha/beta, operand
families chosen independently for A and B (so every digit count 1..16 and every pairing occurs: unrolled bodies,
generic body, multi-pass, scalar fallback), both rounding modes, now and then a non-finite or subnormal entry.
Bits must equal the oracle's."""
import ctypes as C
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
families = [("fpuniform", 1, 0), ("fpuniform", 10, 0), ("fpuniform", 17, 0), ("fpuniform_signed", 24, 12),
            ("fpuniform_signed", 40, 20), ("fpuniform_signed", 60, 30), ("fpuniform_signed", 100, 50),
            ("fpuniform_signed", 200, 100), ("lognormal", 0.0, 2.0), ("lognormal", 0.0, 8.0), ("ill_cond", 1e16, 0),
            ("ill_cond", 1e32, 0), ("naive", 0, 0), ("ints", 0, 0)]
scalars = [0.0, 1.0, -1.0, 2.5, -0.3, 1e-3, 3.0]
variants = [(0, False), (3, False), (4, False), (8, False), (4, True), (6, True), (8, True)]
bits = lambda v: np.ascontiguousarray(v, dtype=np.float64).view(np.int64)  # noqa: E731


def gen(fam, count, seed):
    kind, p0, p1 = fam
    if kind == "ints":
        return rng.integers(-5000, 5001, count).astype(np.float64) * 2.0 ** float(rng.integers(-4, 5))
    return o.gen(kind, count, seed, p0, p1)


t0 = time.time()
bad = 0
paths = {}
for it in range(iters):
    big_k = rng.random() < 0.08
    m, n = int(rng.integers(1, 330)), int(rng.integers(1, 330))
    k = int(rng.integers(8100, 9300)) if big_k else int(rng.integers(1, 700))
    if big_k:
        m, n = min(m, 70), min(n, 70)
    ta, tb = str(rng.choice(["N", "T"])), str(rng.choice(["N", "T"]))
    lda = (m if ta == "T" else k) + int(rng.integers(0, 3))
    ldb = (k if tb == "T" else n) + int(rng.integers(0, 3))
    ldc = n + int(rng.integers(0, 3))
    alpha, beta = float(rng.choice(scalars[1:])), float(rng.choice(scalars))
    fa, fb = families[int(rng.integers(0, len(families)))], families[int(rng.integers(0, len(families)))]
    seed = int(rng.integers(1, 1 << 30))
    a = gen(fa, (k if ta == "T" else m) * lda, seed)
    b = gen(fb, (n if tb == "T" else k) * ldb, seed + 1)
    c0 = gen(("fpuniform_signed", 20, 10), m * ldc, seed + 2)
    special = ""
    r = rng.random()
    if r < 0.04:
        a[int(rng.integers(0, a.size))] = 5e-324 * float(rng.integers(1, 1000))
        special = " +subnormal"
    elif r < 0.07:
        b[int(rng.integers(0, b.size))] = 2.0 ** 600
        special = " +huge"
    mode = int(rng.integers(0, 2))
    fpe, ee = variants[int(rng.integers(0, len(variants)))]
    want = o.exgemm(ta, tb, m, n, k, alpha, a, lda, b, ldb, beta, c0, ldc, 0, mode=mode)
    lib.exblas_set_round_mode(mode)
    c = c0.copy()
    ex.exgemm(ta, tb, m, n, k, alpha, a, lda, b, ldb, beta, c, ldc, fpe, ee)
    v = (C.c_int * 8)()
    lib.exblas_last_gemm_info(v)
    key = (v[0], v[1], v[2])
    paths[key] = paths.get(key, 0) + 1
    ok = (bits(c) == bits(want)).all()
    desc = (f"gemm {ta}{tb} m={m} n={n} k={k} ld={lda},{ldb},{ldc} a={alpha} b={beta} A={fa} B={fb}{special} mode={mode} "
            f"fpe={fpe}{'ee' if ee else ''} path={key}")
    if not ok:
        bad += 1
        print(f"MISMATCH it={it} {desc} ({int((bits(c) != bits(want)).sum())} entries)", flush=True)
    if it % 50 == 0:
        print(f"it {it}: {desc} [{time.time() - t0:.0f} s]", flush=True)
lib.exblas_set_round_mode(0)
print("paths (impl, digits or bits of A, of B): count ->", dict(sorted(paths.items())))
print(f"done: {iters} cases, {bad} mismatches, {time.time() - t0:.0f} s")
sys.exit(1 if bad else 0)
