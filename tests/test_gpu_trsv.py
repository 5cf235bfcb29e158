"""GPU parity tests for ExTRSV against the oracle (bit-exact), MPFR and the reference test's own criterion."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TRSV_VARIANTS = [(0, False), (2, False), (3, False), (4, False), (5, False), (6, False), (7, False), (8, False),
                 (4, True), (6, True), (8, True)]


@pytest.fixture(scope="module")
def ex():
    import torch
    import exblas_amd
    assert torch.cuda.is_available()
    exblas_amd.load_library().exblas_hip_init(-1)
    return exblas_amd


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.int64)


def tri_system(oracle, uplo, n, seed, lda=None, dominant=False, unit=False, rng=(10, 0)):
    """Column-major triangular matrix (flat, n columns of lda) + right-hand side.  The triangle that must not be read
    holds NaN (the reference's generators leave it uninitialised, common.cpp:48-64).  dominant: off-diagonal
    magnitudes below 1/n, diagonal in [0.5, 1) -- keeps |x| bounded for large n; otherwise entries of either sign
    spread over 2^rng[0] binades, which makes |x| grow to 1e100 and beyond within a few hundred rows."""
    lda = lda or n
    if dominant:
        k = int(np.ceil(np.log2(max(n, 2)))) + 1
        off = oracle.gen("fpuniform_signed", n * n, seed, 6, -k).reshape(n, n)
        dia = oracle.gen("fpuniform_signed", n, seed + 1, 1, 0)
    else:
        off = oracle.gen("fpuniform_signed", n * n, seed, rng[0], rng[1]).reshape(n, n)
        dia = oracle.gen("fpuniform_signed", n, seed + 1, rng[0], rng[1])
    m = off.T.copy()                     # logical m[i][j]
    keep = np.tril(np.ones((n, n), bool), -1) if uplo == "L" else np.triu(np.ones((n, n), bool), 1)
    m[~keep] = np.nan
    np.fill_diagonal(m, np.nan if unit else dia)
    a = np.full((n, lda), np.nan)        # a[col][row]
    a[:, :n] = m.T
    b = oracle.gen("fpuniform_signed", n, seed + 2, rng[0], rng[1])
    return a.reshape(-1), b


@pytest.mark.parametrize("uplo", ["L", "U"])
@pytest.mark.parametrize("trans", ["N", "T"])
@pytest.mark.parametrize("n", [1, 5, 63, 64, 65, 200, 333])
def test_extrsv_vs_oracle(ex, oracle, uplo, trans, n):
    """the reference's CTest shape (U N N 256, blas2/CMakeLists.txt:73-80) widened to L/U x N/T and ragged sizes; every
    (fpe, early_exit) variant must give the oracle's bits, and the oracle MPFR's"""
    a, b = tri_system(oracle, uplo, n, 100 + n)
    rc, want = oracle.extrsv(uplo, trans, "N", n, a, n, b, 0)
    assert rc == 0
    if oracle.mpfr() is not None and n <= 200:
        assert (_bits(oracle.mpfr_extrsv(uplo, trans, "N", n, a, n, b, True)) == _bits(want)).all()
    for fpe, ee in TRSV_VARIANTS:
        x = b.copy()
        assert ex.extrsv(uplo, trans, "N", n, a, n, 0, x, 1, 0, fpe, ee) == 0
        assert (_bits(x) == _bits(want)).all(), (uplo, trans, n, fpe, ee, np.nonzero(_bits(x) != _bits(want))[0][:5])


def test_extrsv_unit_lda_incx_offsets(ex, oracle):
    n, lda = 150, 157
    for uplo in "LU":
        for trans in "NT":
            for diag in "NU":
                a, b = tri_system(oracle, uplo, n, 7, lda=lda, unit=(diag == "U"))
                a = np.concatenate([np.full(4, np.nan), a])
                xb = np.full(3 * n + 2, np.nan)
                xb[2::3] = b
                rc, want = oracle.extrsv(uplo, trans, diag, n, a, lda, xb, 0, incx=3, offa=4, offx=2)
                assert rc == 0 and np.isfinite(want[2::3]).all()
                for fpe, ee in ((0, False), (3, False), (8, True)):
                    x = xb.copy()
                    assert ex.extrsv(uplo, trans, diag, n, a, lda, 4, x, 3, 2, fpe, ee) == 0
                    assert (_bits(x[2::3]) == _bits(want[2::3])).all(), (uplo, trans, diag, fpe, ee)
                    assert np.isnan(x[0::3]).all() and np.isnan(x[1::3]).all()   # the gaps are not written


def test_extrsv_reference_criterion_and_dtrsv(ex, oracle):
    """test.extrsv.gpu.cpp:27-92,:141: inf-norm error against the MPFR solve (division inside MPFR, one rounding)
    must stay below 1e-13 for every exact variant; the plain DTRSV (fpe = 1) is only printed there."""
    if oracle.mpfr() is None:
        pytest.skip("no MPFR")
    n = 256
    for uplo in "LU":
        a, b = tri_system(oracle, uplo, n, 31, dominant=True)
        ref = oracle.mpfr_extrsv(uplo, "N", "N", n, a, n, b, False)
        for fpe, ee in ((0, False), (3, False), (4, False), (8, False), (4, True), (6, True), (8, True)):
            x = b.copy()
            ex.extrsv(uplo, "N", "N", n, a, n, 0, x, 1, 0, fpe, ee)
            assert np.max(np.abs(x - ref)) / np.max(np.abs(ref)) <= 1e-13
        x = b.copy()
        ex.extrsv(uplo, "N", "N", n, a, n, 0, x, 1, 0, 1)
        assert np.max(np.abs(x - ref)) / np.max(np.abs(ref)) <= 1e-10   # plain fp64: tolerance, not parity


def test_extrsv_large_dominant(ex, oracle):
    """many block-rows (more than one per CU wave of workgroups), all four orientations"""
    n = 3000
    for uplo, trans in (("L", "N"), ("U", "N"), ("L", "T"), ("U", "T")):
        a, b = tri_system(oracle, uplo, n, 57, dominant=True)
        rc, want = oracle.extrsv(uplo, trans, "N", n, a, n, b, 0)
        for fpe, ee in ((0, False), (4, False), (8, True)):
            x = b.copy()
            ex.extrsv(uplo, trans, "N", n, a, n, 0, x, 1, 0, fpe, ee)
            assert (_bits(x) == _bits(want)).all(), (uplo, trans, fpe, ee)


def test_extrsv_reference_rounding_mode(ex, oracle):
    lib = ex.load_library()
    n = 130
    a, b = tri_system(oracle, "L", n, 77)
    rc, want = oracle.extrsv("L", "N", "N", n, a, n, b, 0, mode=oracle.ROUND_REFERENCE)
    lib.exblas_set_round_mode(1)
    try:
        for fpe, ee in ((0, False), (8, True)):
            x = b.copy()
            ex.extrsv("L", "N", "N", n, a, n, 0, x, 1, 0, fpe, ee)
            assert (_bits(x) == _bits(want)).all()
    finally:
        lib.exblas_set_round_mode(0)


def test_extrsv_unsupported_and_nonfinite(ex, oracle):
    n = 70
    a, b = tri_system(oracle, "L", n, 5)
    x = b.copy()
    assert ex.extrsv("L", "N", "N", n, a, n, 0, x, 1, 0, 10) == -1      # iterative-refinement variants: not shipped
    assert (_bits(x) == _bits(b)).all()
    assert ex.extrsv("L", "N", "N", 0, a, n, 0, x, 1, 0, 0) == 0        # n = 0: nothing to do
    # a zero on the diagonal: IEEE division, then Inf/NaN travel down the substitution
    m = a.reshape(n, n).copy()
    m[3, 3] = 0.0
    x = b.copy()
    ex.extrsv("L", "N", "N", n, m.reshape(-1), n, 0, x, 1, 0, 0)
    rc, want = oracle.extrsv("L", "N", "N", n, a, n, b, 0)
    assert (_bits(x[:3]) == _bits(want[:3])).all() and np.isinf(x[3]) and not np.isfinite(x[4:]).any()


def test_extrsv_device_pointer_api(ex, oracle):
    import torch
    n = 500
    a, b = tri_system(oracle, "U", n, 91, dominant=True)
    rc, want = oracle.extrsv("U", "N", "N", n, a, n, b, 0)
    da = torch.from_numpy(a).cuda()
    for fpe, ee in ((0, False), (6, True)):
        dx = torch.from_numpy(b.copy()).cuda()
        assert ex.extrsv_dev("U", "N", "N", n, da, n, dx, fpe, ee) == 0
        torch.cuda.synchronize()
        assert (_bits(dx.cpu().numpy()) == _bits(want)).all()
    dx = torch.from_numpy(b.copy()).cuda()
    assert ex.extrsv_dev("U", "N", "N", n, da, n, dx, 12) == -1


def test_extrsv_ties_take_the_integer_path(ex, oracle):
    """rows whose exact value is a rounding tie, or within 2^-106 of one: the register fast path cannot decide them
    and must hand them to the superaccumulator; bits still equal the oracle's"""
    lib = ex.load_library()
    n = 96
    m = np.zeros((n, n))                      # logical lower-triangular, unit-free: a_ii = 1
    np.fill_diagonal(m, 1.0)
    b = np.ones(n)
    m[1:, 0] = -2.0 ** -53                    # T_i = 1 + 2^-53 * x_0: a tie, rounds to even (1.0)
    m[2::3, 1] = -2.0 ** -106                 # ... pushed just above the tie: 1 + 2^-52
    m[3::3, 1] = 2.0 ** -106                  # ... just below: 1.0
    m[40:, 0] = 2.0 ** -54                    # T = 1 - 2^-54: tie below a power of two, rounds to 1.0
    a = np.ascontiguousarray(m.T).reshape(-1)  # column-major
    rc, want = oracle.extrsv("L", "N", "N", n, a, n, b, 0)
    if oracle.mpfr() is not None:
        assert (_bits(oracle.mpfr_extrsv("L", "N", "N", n, a, n, b, True)) == _bits(want)).all()
    assert len(set(want.tolist())) >= 2
    for fpe, ee in ((0, False), (4, False), (8, True)):
        x = b.copy()
        ex.extrsv("L", "N", "N", n, a, n, 0, x, 1, 0, fpe, ee)
        assert (_bits(x) == _bits(want)).all(), (fpe, ee, np.nonzero(x != want)[0][:8])
        assert lib.exblas_extrsv_last_slow_rows() >= n // 2


def test_extrsv_scaled_rows_and_columns(ex, oracle):
    """power-of-two row scalings up to 2^+-300 and column scalings up to 2^+-40: every magnitude of numerator,
    denominator and quotient the division shortcut (and its fallback to '/') can meet; the well-conditioned core
    keeps almost every row on the register fast path"""
    lib = ex.load_library()
    n = 2048
    rng = np.random.default_rng(12)
    for uplo in "LU":
        a, b = tri_system(oracle, uplo, n, 41, dominant=True)
        m = a.reshape(n, n).T.copy()          # logical
        rs = 2.0 ** rng.integers(-300, 301, n)
        cs = 2.0 ** rng.integers(-40, 41, n)
        rs[::97] = 2.0 ** rng.integers(-500, 501, rs[::97].size)   # some rows beyond the shortcut's exponent window
        m = m * rs[:, None] * cs[None, :]
        bb = b * rs
        aa = np.ascontiguousarray(m.T).reshape(-1)
        rc, want = oracle.extrsv(uplo, "N", "N", n, aa, n, bb, 0)
        assert np.isfinite(want).all()
        for fpe, ee in ((0, False), (6, True)):
            x = bb.copy()
            ex.extrsv(uplo, "N", "N", n, aa, n, 0, x, 1, 0, fpe, ee)
            assert (_bits(x) == _bits(want)).all(), (uplo, fpe, ee, np.nonzero(_bits(x) != _bits(want))[0][:8])
            assert lib.exblas_extrsv_last_slow_rows() < n // 4


def test_standalone_cpp_extrsv_caller(ex):
    """reference-style C++ program (the flow of tests/test.extrsv.gpu.cpp, its CTest arguments
    blas2/CMakeLists.txt:73-80) linked only against libexblas.so"""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.run(["make", "-C", os.path.join(root, "tests", "cpp")], check=True, capture_output=True)
    for argv in (["U", "N", "N", "256"], ["U", "N", "N", "256", "10", "0"], ["L", "N", "N", "256", "10", "0"],
                 ["L", "T", "U", "300", "6", "0"]):
        r = subprocess.run([os.path.join(root, "tests", "cpp", "test_extrsv_gpu"), *argv], capture_output=True,
                           text=True, timeout=600)
        assert r.returncode == 0 and "TestPassed; ALL OK!" in r.stdout, (argv, r.stdout[-2000:], r.stderr[-2000:])


def test_extrsv_full_size_more_blocks_than_resident(ex, oracle):
    """n = 40000: 625 block-rows, more than the 512 workgroups the chip holds at once, so the ticket order of
    the block-rows (a workgroup only waits for workgroups that already run) is what keeps the solve alive; and the
    BASELINE-scale parity check of the whole chain: every component equal to the oracle's (a sequential CPU solve of
    8e8 exact multiply-adds, ~15 s)."""
    n = 40000
    k = int(np.ceil(np.log2(n))) + 1
    a = oracle.gen("fpuniform_signed", n * n, 123, 6, -k)       # column-major; the upper triangle is never read
    a[::n + 1] = oracle.gen("fpuniform_signed", n, 124, 1, 0)   # diagonal in +-[0.5, 1): dominant
    b = oracle.gen("fpuniform_signed", n, 125, 10, 0)
    x = b.copy()
    assert ex.extrsv("L", "N", "N", n, a, n, 0, x, 1, 0, 8, True) == 0
    slow = ex.load_library().exblas_extrsv_last_slow_rows()
    assert 0 <= slow < n // 100
    rc, want = oracle.extrsv("L", "N", "N", n, a, n, b, 0)
    assert rc == 0 and np.isfinite(want).all()
    assert (_bits(x) == _bits(want)).all(), np.nonzero(_bits(x) != _bits(want))[0][:8]


def test_extrsv_randomized_soak(ex):
    """tools/stress_trsv.py: 120 random (n, uplo, trans, diag, lda, incx, variant) cases, well-conditioned and wild
    (overflowing) systems, each run three times: bits equal to the oracle and equal run to run"""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "stress_trsv.py"), "120", "3"], capture_output=True,
                       text=True, timeout=900)
    assert r.returncode == 0 and "0 mismatches" in r.stdout, (r.stdout[-3000:], r.stderr[-2000:])


def test_extrsv_and_gemv_in_a_hip_graph(ex, oracle):
    """exgemv_dev and extrsv_dev are stream-ordered launches (plus two memset nodes for the solve): captured once,
    replayed three times -- solve L x = A v + y repeatedly, bits equal to the oracle's every time"""
    import torch
    n = 700
    a, b = tri_system(oracle, "L", n, 333, dominant=True)
    a = np.nan_to_num(a, nan=0.0)            # exgemv reads the whole square: the upper triangle must be finite
    v = oracle.gen("fpuniform_signed", n, 334, 10, 0)
    da, dv = torch.from_numpy(a).cuda(), torch.from_numpy(v).cuda()
    dy0 = torch.from_numpy(b).cuda()
    dx = dy0.clone()
    ex.exgemv_dev("N", n, n, 1.0, da, n, dv, 1.0, dx, 8, True)     # lazy allocations happen outside the capture
    ex.extrsv_dev("L", "N", "N", n, da, n, dx, 8, True)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s):
            dx.copy_(dy0)
            ex.exgemv_dev("N", n, n, 1.0, da, n, dv, 1.0, dx, 8, True)
            ex.extrsv_dev("L", "N", "N", n, da, n, dx, 8, True)
    want = oracle.extrsv("L", "N", "N", n, a, n, oracle.exgemv("N", n, n, 1.0, a, n, v, 1.0, b, 0), 0)[1]
    for _ in range(3):
        dx.zero_()
        g.replay()
        torch.cuda.synchronize()
        assert (_bits(dx.cpu().numpy()) == _bits(want)).all()
