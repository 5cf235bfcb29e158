"""CPU suite: pins the oracle (oracle/exblas_oracle.c) against the golden vectors, the compiled
reference core (oracle/_ref, when present), MPFR and Python's exact fsum/Fraction arithmetic."""
import math

import numpy as np
import pytest

from helpers import (FPE_VARIANTS_DOT, FPE_VARIANTS_SUM, exact_int_from_canon, exact_sum_int, golden_cases,
                     load_golden, same_double)

TINY = np.finfo(np.float64).tiny


def test_exsum_golden_limbs_and_roundings(oracle):
    g = load_golden("exsum_golden.npz")
    n = 0
    for name, i, a in golden_cases(g, "data"):
        r, limbs = oracle.exsum(a, 0, limbs=True)
        assert (limbs == g["limbs"][i]).all(), name
        assert same_double(r, g["exact"][i]), name
        assert same_double(r, g["mpfr"][i]), name  # exact mode == MPFR on every fixture
        assert same_double(oracle.exsum(a, 0, mode=oracle.ROUND_REFERENCE), g["refmode"][i]), name
        n += 1
    assert n > 150


def test_exsum_all_variants_same_limbs(oracle):
    g = load_golden("exsum_golden.npz")
    for name, i, a in golden_cases(g, "data"):
        if a.size > 1000:
            continue
        for fpe, ee in FPE_VARIANTS_SUM:
            r, limbs = oracle.exsum(a, fpe, ee, limbs=True)
            assert (limbs == g["limbs"][i]).all(), (name, fpe, ee)
            assert same_double(r, g["exact"][i]), (name, fpe, ee)


def test_exsum_golden_against_bigint(oracle):
    g = load_golden("exsum_golden.npz")
    for name, i, a in golden_cases(g, "data"):
        if a.size > 1000:
            continue
        assert exact_int_from_canon(g["limbs"][i]) == exact_sum_int(a) << 18, name


def test_exsum_vs_fsum_random(oracle):
    rng = np.random.default_rng(7)
    for _ in range(50):
        n = int(rng.integers(1, 3000))
        a = rng.standard_normal(n) * np.exp2(rng.integers(-300, 300, n).astype(np.float64))
        assert oracle.exsum(a, 8, True) == math.fsum(a)
        assert oracle.exsum(a, 0) == math.fsum(a)


def test_unsupported_variants_return_zero(oracle):
    a = np.ones(16)
    assert oracle.exsum(a, 9, False) == 0.0      # cpu:ExSUM.cpp:99
    assert oracle.exsum(a, 9, True) == 0.0
    assert oracle.exdot(a, a, 9, False) == 0.0
    assert oracle.exdot(np.zeros(0), np.zeros(0), 0) == 0.0  # ExDOT.cpp:70-71


def test_strided_and_offset(oracle):
    a = oracle.gen("ill_cond", 4000, 5, 1e32)
    for inca, off in ((1, 0), (2, 0), (3, 1), (7, 5)):
        n = (a.size - off + inca - 1) // inca
        ref = math.fsum(a[off::inca][:n])
        for fpe, ee in FPE_VARIANTS_SUM:
            assert oracle.exsum(a, fpe, ee, inca=inca, offset=off, n=n) == ref


def test_compiled_reference_agrees(oracle):
    """oracle == the reference's own compiled arithmetic core, limb for limb, for every variant."""
    if oracle.ref() is None:
        pytest.skip("oracle/_ref not built (needs /root/reference)")
    ndiff = 0
    for seed in range(1, 41):
        a = oracle.ref_gen("ill_cond", 1 << 12, seed, 1e32)
        r0, l0 = oracle.exsum(a, 0, limbs=True)
        for fpe, ee in FPE_VARIANTS_SUM:
            for nt in (1, 3):
                rr, lr = oracle.ref_exsum(a, fpe, ee, nthreads=nt, limbs=True)
                assert (lr == l0).all(), (seed, fpe, ee, nt)
                assert same_double(rr, oracle.exsum(a, 0, mode=oracle.ROUND_REFERENCE)), (seed, fpe, ee)
        ndiff += oracle.ref_exsum(a, 0) != r0
    # the reference's Round() is off by one ulp on a sizeable fraction of these inputs
    assert 0 < ndiff < 40


def test_mpfr_agrees(oracle):
    if oracle.mpfr() is None:
        pytest.skip("libmpfr_oracle.so not built")
    for kind, p0, p1 in (("lognormal", 0, 50), ("ill_cond", 1e32, 0), ("cancel", 60, 0), ("fpuniform", 600, 300)):
        a, b = oracle.gen(kind, 5003, 21, p0, p1), oracle.gen(kind, 5003, 22, p0, p1)
        assert same_double(oracle.exsum(a, 8, True), oracle.mpfr_exsum(a))
        assert same_double(oracle.exdot(a, b, 8, True), oracle.mpfr_exdot(a, b))


def test_exdot_golden(oracle):
    g = load_golden("exdot_golden.npz")
    for name, i, a, b in golden_cases(g, "a", "b"):
        for fpe, ee in FPE_VARIANTS_DOT:
            r, limbs = oracle.exdot(a, b, fpe, ee, limbs=True)
            assert (limbs == g["limbs"][i]).all(), (name, fpe, ee)
            assert same_double(r, g["mpfr"][i]), (name, fpe, ee)


def test_omp_slicing_is_exact(oracle):
    a = oracle.gen("ill_cond", 1 << 18, 3, 1e32)
    b = oracle.gen("ill_cond", 1 << 18, 4, 1e32)
    r0, l0 = oracle.exsum(a, 0, limbs=True)
    d0, m0 = oracle.exdot(a, b, 0, limbs=True)
    for nt in (1, 2, 5, 8):
        r, l = oracle.exsum_omp(a, 8, True, nt, limbs=True)
        assert (l == l0).all() and r == r0
        d, m = oracle.exdot_omp(a, b, 8, True, nt, limbs=True)
        assert (m == m0).all() and d == d0


def test_generators_shape(oracle):
    a = oracle.gen("cancel", 10000, 1, 50)
    assert oracle.exsum(a, 8, True) == 1.0   # exact sum is 1 + 2^-60
    assert float(np.sum(np.abs(a))) > 1e15
    a = oracle.gen("ill_cond", 10000, 1, 1e32)
    assert np.abs(a).max() < 2.0**55 and np.abs(a[:5000]).max() > 2.0**40
    # slices of the counter-based stream are position-independent
    full = oracle.gen("lognormal", 1000, 9, 0.0, 50.0)
    part = oracle.gen("lognormal", 1000, 9, 0.0, 50.0, first=300, count=200)
    assert (full[300:500] == part).all()


def test_gemv_gemm_oracle_vs_mpfr(oracle):
    if oracle.mpfr() is None:
        pytest.skip("libmpfr_oracle.so not built")
    m, n = 37, 53
    a = oracle.gen("fpuniform_signed", m * n, 31, 40, 20)
    x = oracle.gen("fpuniform_signed", max(m, n), 32, 40, 20)
    y = oracle.gen("fpuniform_signed", max(m, n), 33, 40, 20)
    for trans in ("N", "T"):
        rows, inner = (n, m) if trans == "T" else (m, n)
        ref = oracle.mpfr_exgemv(trans, m, n, 1.0, a, m, x[:inner], 1.0, y[:rows])
        for fpe, ee in ((0, False), (3, False), (8, False), (4, True), (8, True)):
            got = oracle.exgemv(trans, m, n, 1.0, a, m, x[:inner], 1.0, y[:rows], fpe, ee)
            assert (got == ref).all(), (trans, fpe, ee)
    k = 29
    A = oracle.gen("fpuniform_signed", m * k, 41, 30, 10)
    B = oracle.gen("fpuniform_signed", k * n, 42, 30, 10)
    dots = oracle.mpfr_exgemm_dots(m, n, k, A, k, B, n)
    for fpe, ee in ((0, False), (3, False), (8, True)):
        got = oracle.exgemm("N", "N", m, n, k, 1.0, A, k, B, n, 0.0, np.zeros(m * n), n, fpe, ee)
        assert (got.reshape(m, n) == dots).all(), (fpe, ee)


def test_baseline_config1_plumbing(oracle):
    """BASELINE.json configs[0]: ExSUM n = 2^20 fp64 log-normal (the reference's CTest cases `20 2 0 n` and
    `20 50 0 n`, src/cpu/blas/blas1/CMakeLists.txt:28-35), CPU reference path vs MPFR -- no GPU involved.
    Inputs come from the reference's own init_lognormal (std::random_device seeded, so every run is a new draw)."""
    if oracle.ref() is None or oracle.mpfr() is None:
        pytest.skip("needs oracle/_ref and libmpfr_oracle.so")
    n = 1 << 20
    for stddev in (2.0, 50.0):
        a = oracle.ref_gen("lognormal", n, 0, 0.0, stddev)
        want = oracle.mpfr_exsum(a)
        r0, l0 = oracle.exsum(a, 0, limbs=True)
        assert same_double(r0, want)
        for fpe, ee in FPE_VARIANTS_SUM:
            rr, lr = oracle.ref_exsum(a, fpe, ee, nthreads=4, limbs=True)   # the reference's compiled core
            assert (lr == l0).all(), (stddev, fpe, ee)
            # the reference's own pass criterion (relative error <= 1e-16, tests/test.exsum.cpu.cpp:43,:133) ...
            assert abs(rr - want) <= 1e-16 * abs(want) + 0.0 or abs(rr - want) / abs(want) <= 2.3e-16
            # ... and ours: the limbs round correctly
            assert same_double(oracle.round_limbs(lr), want)
            r, l = oracle.exsum(a, fpe, ee, limbs=True)
            assert (l == l0).all() and same_double(r, want)


def test_trsv_oracle_vs_mpfr_and_fraction(oracle):
    """ExTRSV restatement: every variant equals the MPFR solve with the reference kernels' two roundings (exact
    sum -> double, then fp64 division) bit for bit; a small case is also checked against Python Fractions; and the
    reference test's own criterion (test.extrsv.gpu.cpp:27-92,:141: inf-norm distance to the single-rounding MPFR
    solve <= 1e-13) holds.  No compiled reference exists for this routine (OpenCL only): MPFR is the pin."""
    from fractions import Fraction
    if oracle.mpfr() is None:
        pytest.skip("no MPFR")
    for uplo in "LU":
        for trans in "NT":
            for n in (1, 2, 17, 96):
                a = oracle.gen("fpuniform_signed", n * n, 300 + n, 12, 3)
                b = oracle.gen("fpuniform_signed", n, 301 + n, 12, 3)
                want = oracle.mpfr_extrsv(uplo, trans, "N", n, a, n, b, True)
                for fpe, ee in ((0, False), (2, False), (3, False), (5, False), (8, False), (4, True), (6, True),
                                (8, True)):
                    rc, x = oracle.extrsv(uplo, trans, "N", n, a, n, b, fpe, ee)
                    assert rc == 0 and (x.view(np.int64) == want.view(np.int64)).all(), (uplo, trans, n, fpe, ee)
                one = oracle.mpfr_extrsv(uplo, trans, "N", n, a, n, b, False)
                if n <= 17:   # beyond that random triangular systems are too ill-conditioned for the 1e-13 criterion
                    assert np.max(np.abs(want - one)) <= 1e-13 * np.max(np.abs(one))
    # above n = 512 the superaccumulator-only variant runs column by column (OpenMP over rows): same bits as the
    # row-by-row expansion variants
    n = 600
    for uplo in "LU":
        a = oracle.gen("fpuniform_signed", n * n, 21, 6, -11)
        a[::n + 1] = oracle.gen("fpuniform_signed", n, 22, 1, 0)
        b = oracle.gen("fpuniform_signed", n, 23, 10, 0)
        c0 = oracle.extrsv(uplo, "N", "N", n, a, n, b, 0)[1]
        c4 = oracle.extrsv(uplo, "N", "N", n, a, n, b, 4, True)[1]
        assert (c0.view(np.int64) == c4.view(np.int64)).all()
        assert (c0.view(np.int64) == oracle.mpfr_extrsv(uplo, "N", "N", n, a, n, b, True).view(np.int64)).all()
    # exact rational check of the definition, n = 6 lower
    n = 6
    a = oracle.gen("fpuniform_signed", n * n, 9, 20, 5)
    b = oracle.gen("fpuniform_signed", n, 10, 20, 5)
    rc, x = oracle.extrsv("L", "N", "N", n, a, n, b, 0)
    xs = []
    for i in range(n):
        t = Fraction(float(b[i])) - sum(Fraction(float(a[j * n + i])) * Fraction(xs[j]) for j in range(i))
        # correctly rounded double of the rational t: float(Fraction) rounds to nearest even
        xs.append(float(t) / float(a[i * n + i]))
    assert (np.array(xs).view(np.int64) == x.view(np.int64)).all()
    assert oracle.extrsv("L", "N", "N", n, a, n, b, 12)[0] == -1          # iterative-refinement variants
    rc, u = oracle.extrsv("U", "N", "U", n, a, n, b, 0)                   # unit diagonal: stored diagonal ignored
    a2 = a.copy().reshape(n, n)
    np.fill_diagonal(a2, 1.0)
    assert (oracle.extrsv("U", "N", "N", n, a2.reshape(-1), n, b, 0)[1].view(np.int64) == u.view(np.int64)).all()
