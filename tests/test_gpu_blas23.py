"""GPU parity tests for ExGEMV / ExGEMM against the oracle (bit-exact) and MPFR."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GEMV_VARIANTS = [(0, False), (2, False), (3, False), (4, False), (8, False), (4, True), (6, True), (8, True)]
GEMM_VARIANTS = [(0, False), (3, False), (4, False), (8, False), (4, True), (6, True), (8, True)]


@pytest.fixture(scope="module")
def ex():
    import torch
    import exblas_amd
    assert torch.cuda.is_available()
    exblas_amd.load_library().exblas_hip_init(-1)
    return exblas_amd


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.int64)


@pytest.mark.parametrize("trans", ["N", "T"])
@pytest.mark.parametrize("m,n", [(256, 256), (384, 200), (130, 517), (1, 9), (7, 1), (1030, 770)])
def test_exgemv_vs_oracle(ex, oracle, trans, m, n):
    """the reference's CTest matrix: trans N/T x (m=n, m<n, m>n), column-major, alpha=beta=1 (blas2/CMakeLists.txt:11-64)"""
    for kind, p0, p1 in (("fpuniform", 10, 0), ("fpuniform_signed", 60, 30), ("lognormal", 0.0, 2.0)):
        lda = m + (3 if (m % 2) else 2)
        a = oracle.gen(kind, lda * n, 51, p0, p1)
        rows, inner = (n, m) if trans == "T" else (m, n)
        x = oracle.gen(kind, inner, 52, p0, p1)
        y0 = oracle.gen(kind, rows, 53, p0, p1)
        want = oracle.exgemv(trans, m, n, 1.0, a, lda, x, 1.0, y0, 0)
        if oracle.mpfr() is not None and m * n < 100000:
            a_dense = a.reshape(n, lda)[:, :m].copy().reshape(-1)
            assert (_bits(oracle.mpfr_exgemv(trans, m, n, 1.0, a_dense, m, x, 1.0, y0)) == _bits(want)).all()
        for fpe, ee in GEMV_VARIANTS:
            y = y0.copy()
            ex.exgemv(trans, m, n, 1.0, a, lda, 0, x, 1, 0, 1.0, y, 1, 0, fpe, ee)
            assert (_bits(y) == _bits(want)).all(), (trans, m, n, kind, fpe, ee, np.nonzero(y != want)[0][:5])


@pytest.mark.parametrize("trans", ["N", "T"])
def test_exgemv_long_vectors_wide_range(ex, oracle, trans):
    """Columns / rows long enough for the tiled vector loops (m >= 2048 for 'T': several 2048-element tiles + a ragged
    tail; n >= 16 column groups drawn dynamically for 'N') on data whose exponent range outgrows the small expansions:
    tiles spill, 'T' sends the rest of the column through its direct loop (straight to the integer accumulator), the
    range guard diverts whole tiles (|x| >= 2^1000 entries).  Every variant, the oracle's bits."""
    m, n = (3 * 2048 + 77, 40) if trans == "T" else (520, 3 * 128 + 5)
    lda = m + 2
    rows, inner = (n, m) if trans == "T" else (m, n)
    for kind, p0, p1 in (("fpuniform_signed", 600, 300), ("lognormal", 0.0, 50.0), ("fpuniform", 10, 0)):
        a = oracle.gen(kind, lda * n, 151, p0, p1)
        x = oracle.gen("fpuniform_signed", inner, 152, 20, 10)
        y0 = oracle.gen("fpuniform_signed", rows, 153, 20, 10)
        if kind == "fpuniform":                       # a few entries the guard must divert (products stay finite)
            a[5 * lda + 7] = 2.0 ** 1010
            a[9 * lda + (m - 3)] = -(2.0 ** 1005)
            x[:] = np.where(np.abs(x) > 1.0, 1.0, x)
        want = oracle.exgemv(trans, m, n, 1.0, a, lda, x, 1.0, y0, 0)
        for fpe, ee in GEMV_VARIANTS:
            y = y0.copy()
            ex.exgemv(trans, m, n, 1.0, a, lda, 0, x, 1, 0, 1.0, y, 1, 0, fpe, ee)
            assert (_bits(y) == _bits(want)).all(), (trans, kind, fpe, ee, np.nonzero(y != want)[0][:5])


def test_exgemv_alpha_beta_strides_offsets(ex, oracle):
    m, n, lda = 300, 210, 301
    a = oracle.gen("fpuniform_signed", lda * n + 5, 61, 40, 20)
    for trans in ("N", "T"):
        rows, inner = (n, m) if trans == "T" else (m, n)
        x = oracle.gen("fpuniform_signed", 2 * inner + 3, 62, 40, 20)
        y0 = oracle.gen("fpuniform_signed", 3 * rows + 2, 63, 40, 20)
        for alpha, beta in ((1.0, 0.0), (2.5, 1.0), (-0.3, 0.7)):
            want = oracle.exgemv(trans, m, n, alpha, a, lda, x, beta, y0, 0, incx=2, incy=3, offa=5, offx=3, offy=2)
            for fpe, ee in ((0, False), (4, False), (8, True)):
                y = y0.copy()
                ex.exgemv(trans, m, n, alpha, a, lda, 5, x, 2, 3, beta, y, 3, 2, fpe, ee)
                assert (_bits(y) == _bits(want)).all(), (trans, alpha, beta, fpe, ee)


def test_dgemv_baseline_close(ex, oracle):
    m, n = 200, 300
    a = oracle.gen("fpuniform", m * n, 71, 10, 0)
    x = oracle.gen("fpuniform", max(m, n), 72, 10, 0)
    for trans in ("N", "T"):
        rows, inner = (n, m) if trans == "T" else (m, n)
        y = np.ones(rows)
        ex.exgemv(trans, m, n, 1.0, a, m, 0, x[:inner], 1, 0, 1.0, y, 1, 0, 1)   # fpe == 1: plain DGEMV
        want = oracle.exgemv(trans, m, n, 1.0, a, m, x[:inner], 1.0, np.ones(rows), 0)
        assert np.max(np.abs(y - want) / np.abs(want)) < 1e-13   # tolerance 1e-13: not reproducible by design


@pytest.mark.parametrize("m,n,k", [(256, 256, 256), (33, 47, 129), (16, 16, 5), (100, 20, 300)])
def test_exgemm_vs_oracle(ex, oracle, m, n, k):
    """reference CTest: 256^3 row-major alpha=beta=1 (blas3/CMakeLists.txt:11-18); plus ragged shapes"""
    for kind, p0, p1 in (("fpuniform", 10, 0), ("fpuniform_signed", 40, 20)):
        a = oracle.gen(kind, m * k, 81, p0, p1)
        b = oracle.gen(kind, k * n, 82, p0, p1)
        c0 = oracle.gen(kind, m * n, 83, p0, p1)
        want = oracle.exgemm("N", "N", m, n, k, 1.0, a, k, b, n, 1.0, c0, n, 0)
        if oracle.mpfr() is not None and m * n * k < 3000000:
            dots = oracle.mpfr_exgemm_dots(m, n, k, a, k, b, n).reshape(-1)
            assert (_bits(c0 + dots) == _bits(want)).all()
        variants = GEMM_VARIANTS if m * n * k <= 256 ** 3 // 4 else [(0, False), (4, False), (8, True)]
        for fpe, ee in variants:
            c = c0.copy()
            ex.exgemm("N", "N", m, n, k, 1.0, a, k, b, n, 1.0, c, n, fpe, ee)
            assert (_bits(c) == _bits(want)).all(), (m, n, k, kind, fpe, ee)


def test_exgemm_trans_alpha_beta(ex, oracle):
    m, n, k = 40, 56, 72
    for ta, tb in (("N", "T"), ("T", "N"), ("T", "T")):
        lda = (m if ta == "T" else k) + 1
        ldb = (k if tb == "T" else n) + 2
        a = oracle.gen("fpuniform_signed", (k if ta == "T" else m) * lda, 91, 30, 10)
        b = oracle.gen("fpuniform_signed", (n if tb == "T" else k) * ldb, 92, 30, 10)
        c0 = oracle.gen("fpuniform_signed", m * (n + 3), 93, 30, 10)
        for alpha, beta in ((1.0, 1.0), (0.5, 0.0), (-1.25, 2.0)):
            want = oracle.exgemm(ta, tb, m, n, k, alpha, a, lda, b, ldb, beta, c0, n + 3, 4, False)
            c = c0.copy()
            ex.exgemm(ta, tb, m, n, k, alpha, a, lda, b, ldb, beta, c, n + 3, 4, False)
            assert (_bits(c) == _bits(want)).all(), (ta, tb, alpha, beta)


def gemm_info(lib):
    """(path, slices of A, slices of B) of the last exgemm: path 0 scalar kernel, 1 fp64 slices, 2 int8 digit slices;
    4 int8 residues: (4, bits of A, bits of B, moduli, moduli the workspace was reserved for)"""
    import ctypes as C
    v = (C.c_int * 8)()
    assert lib.exblas_last_gemm_info(v) == 0
    return (v[0], v[1], v[2], v[3], v[4]) if v[0] == 4 else (v[0], v[1], v[2])


# cumulative bits of the residue path's moduli 256, 255, 253, 251, 247, 241, ... (floor(log2(product of the first L)))
def _crt_moduli_needed(bits_a, bits_b, k):
    from math import gcd
    need = bits_a + bits_b + max(0, (k - 1).bit_length()) + 2
    ps, prod, c = [], 1, 256
    while prod.bit_length() - 1 < need:
        if all(gcd(c, q) == 1 for q in ps):
            ps.append(c)
            prod *= c
        c -= 1
    return len(ps)


@pytest.mark.parametrize("m,n,k", [(64, 64, 512), (130, 75, 1100), (16, 200, 33), (200, 260, 150), (257, 300, 70)])
def test_exgemm_residue_path_is_exact(ex, oracle, m, n, k):
    """The residue path (blas3_crt.hip: one int8 GEMM per 8-bit modulus + Chinese-remainder reconstruction), forced
    (mode 4) on shapes below its automatic threshold and automatic (mode 0) from min(m, n) >= 192: the oracle's bits
    for every data family, both rounding modes, transposes, leading dimensions, alpha / beta -- with exactly the number
    of moduli the operand widths call for."""
    lib = ex.load_library()
    rng = np.random.default_rng(6)
    cases = {
        "fpuniform_r10": (oracle.gen("fpuniform", m * k, 81, 10, 0), oracle.gen("fpuniform", k * n, 82, 10, 0)),
        "fpuniform_r17": (oracle.gen("fpuniform", m * k, 91, 17, 0), oracle.gen("fpuniform", k * n, 92, 17, 0)),
        "naive": (oracle.gen("naive", m * k, 1), oracle.gen("naive", k * n, 1)),
        "small_ints": (rng.integers(-1000, 1000, m * k).astype(np.float64), rng.integers(-1000, 1000, k * n).astype(np.float64)),
        "ill_cond_1e32": (oracle.gen("ill_cond", m * k, 83, 1e32), oracle.gen("ill_cond", k * n, 84, 1e32)),
        "wide_r60": (oracle.gen("fpuniform_signed", m * k, 85, 60, 30), oracle.gen("fpuniform_signed", k * n, 86, 60, 30)),
        "ones": (np.ones(m * k), -np.ones(k * n)),
    }
    auto = min(m, n) >= 192
    try:
        for name, (a, b) in cases.items():
            c0 = oracle.gen("fpuniform_signed", m * n, 87, 10, 5)
            for mode in (oracle.ROUND_EXACT, oracle.ROUND_REFERENCE):
                want = oracle.exgemm("N", "N", m, n, k, 1.0, a, k, b, n, 1.0, c0, n, 0, mode=mode)
                lib.exblas_set_round_mode(mode)
                for path in ((0, 4) if auto else (4,)):
                    lib.exblas_set_gemm_path(path)
                    c = c0.copy()
                    ex.exgemm("N", "N", m, n, k, 1.0, a, k, b, n, 1.0, c, n, 8, True)
                    info = gemm_info(lib)
                    assert info[0] == 4, (name, path, info)
                    assert info[3] == _crt_moduli_needed(info[1], info[2], k), (name, info)
                    assert (_bits(c) == _bits(want)).all(), (name, path, mode, info, int((c != want).sum()))
        lib.exblas_set_gemm_path(4)
        for ta, tb in (("N", "T"), ("T", "N"), ("T", "T")):
            lda = (m if ta == "T" else k) + 1
            ldb = (k if tb == "T" else n) + 2
            a = oracle.gen("fpuniform_signed", (k if ta == "T" else m) * lda, 91, 8, 4)
            b = oracle.gen("fpuniform_signed", (n if tb == "T" else k) * ldb, 92, 8, 4)
            c0 = oracle.gen("fpuniform_signed", m * (n + 3), 93, 8, 4)
            for alpha, beta in ((1.0, 1.0), (0.5, 0.0), (-1.25, 2.0)):
                for mode in (oracle.ROUND_EXACT, oracle.ROUND_REFERENCE):
                    want = oracle.exgemm(ta, tb, m, n, k, alpha, a, lda, b, ldb, beta, c0, n + 3, 4, False, mode=mode)
                    lib.exblas_set_round_mode(mode)
                    c = c0.copy()
                    ex.exgemm(ta, tb, m, n, k, alpha, a, lda, b, ldb, beta, c, n + 3, 4, False)
                    assert gemm_info(lib)[0] == 4
                    assert (_bits(c) == _bits(want)).all(), (ta, tb, alpha, beta, mode)
    finally:
        lib.exblas_set_gemm_path(0)
        lib.exblas_set_round_mode(0)


@pytest.mark.parametrize("ta,tb,m,n,k", [("N", "N", 1000, 777, 2500), ("T", "N", 515, 1300, 900), ("N", "T", 700, 260, 20000)])
def test_exgemm_residue_vs_digits_midsize(ex, oracle, ta, tb, m, n, k):
    """Ragged mid-size products (several 256 x 256 workgroup tiles with clamped edges, k across the 8192-per-launch
    boundary): the residue path, the digit-slice path and the scalar kernel produce the same bits, and a block of rows
    equals the oracle."""
    import torch
    lib = ex.load_library()
    lda = (m if ta == "T" else k) + 3
    ldb = (k if tb == "T" else n) + 1
    A = ex.gen_dev("fpuniform_signed", (k if ta == "T" else m) * lda, 71, 12, 6)
    B = ex.gen_dev("lognormal", (n if tb == "T" else k) * ldb, 72, 0.0, 2.0)
    C0 = ex.gen_dev("fpuniform_signed", m * n, 73, 10, 5)
    outs = {}
    try:
        for path in (4, 2, 1):
            lib.exblas_set_gemm_path(path)
            C = C0.clone()
            ex.exgemm_dev(ta, tb, m, n, k, -0.75, A, lda, B, ldb, 2.0, C, n, 8, True)
            torch.cuda.synchronize()
            assert gemm_info(lib)[0] == {4: 4, 2: 2, 1: 0}[path]
            outs[path] = C
    finally:
        lib.exblas_set_gemm_path(0)
    assert torch.equal(outs[4].view(torch.int64), outs[2].view(torch.int64)), "residues != digit slices"
    assert torch.equal(outs[4].view(torch.int64), outs[1].view(torch.int64)), "residues != scalar kernel"
    rows = 24                                    # the oracle on the first rows (full k)
    ha, hb, hc0 = A.cpu().numpy(), B.cpu().numpy(), C0.cpu().numpy()
    if ta == "T":
        a_blk = np.ascontiguousarray(ha.reshape(k, lda)[:, :rows]).reshape(-1)
        want = oracle.exgemm(ta, tb, rows, n, k, -0.75, a_blk, rows, hb, ldb, 2.0, hc0[:rows * n].copy(), n, 0)
    else:
        want = oracle.exgemm(ta, tb, rows, n, k, -0.75, ha[:rows * lda], lda, hb, ldb, 2.0, hc0[:rows * n].copy(), n, 0)
    assert (_bits(outs[4].cpu().numpy()[:rows * n]) == _bits(want)).all()


@pytest.mark.parametrize("ta", ["N", "T"])
def test_exgemm_residue_row_chunks(ex, oracle, ta):
    """m > 3072: the residue path reduces A' and reconstructs C in row chunks of 2048 rows that share one chunk-sized
    workspace (planes of A', residues of C); the last chunk is ragged (not a multiple of 256, 64 or 4 rows).  Every
    entry equals the digit-slice path, rows from every chunk equal the oracle, beta = 1 reads the right rows of C."""
    import torch
    lib = ex.load_library()
    m, n, k = 2 * 2048 + 203, 260, 150
    lda = (m if ta == "T" else k) + 1
    A = ex.gen_dev("fpuniform_signed", (k if ta == "T" else m) * lda, 171, 12, 6)
    B = ex.gen_dev("fpuniform_signed", k * n, 172, 10, 5)
    C0 = ex.gen_dev("fpuniform_signed", m * n, 173, 10, 5)
    outs = {}
    try:
        for path in (4, 2):
            lib.exblas_set_gemm_path(path)
            C = C0.clone()
            ex.exgemm_dev(ta, "N", m, n, k, 1.0, A, lda, B, n, 1.0, C, n, 8, True)
            torch.cuda.synchronize()
            assert gemm_info(lib)[0] == path
            outs[path] = C
    finally:
        lib.exblas_set_gemm_path(0)
    assert torch.equal(outs[4].view(torch.int64), outs[2].view(torch.int64))
    ha, hb, hc0, got = A.cpu().numpy(), B.cpu().numpy(), C0.cpu().numpy(), outs[4].cpu().numpy()
    for r0 in (0, 2040, 4090, m - 9):            # rows around every chunk boundary and the ragged tail
        rows = min(9, m - r0)
        if ta == "T":
            a_blk = np.ascontiguousarray(ha.reshape(k, lda)[:, r0:r0 + rows]).reshape(-1)
            want = oracle.exgemm("T", "N", rows, n, k, 1.0, a_blk, rows, hb, n, 1.0, hc0[r0 * n:(r0 + rows) * n].copy(), n, 0)
        else:
            want = oracle.exgemm("N", "N", rows, n, k, 1.0, ha[r0 * lda:(r0 + rows) * lda], lda, hb, n, 1.0,
                                 hc0[r0 * n:(r0 + rows) * n].copy(), n, 0)
        assert (_bits(got[r0 * n:(r0 + rows) * n]) == _bits(want)).all(), (ta, r0)


def test_exgemm_residue_chunks_tiles_and_k_passes_together(ex, oracle):
    """Several row chunks (m > 3072) x several 256-column workgroup tiles x two contraction launches per modulus
    (k > 8192: residues of the second k block added modulo p to those of the first) in ONE call, ragged in every
    dimension, alpha / beta other than 0 / 1: residues == digit slices everywhere, rows at the chunk boundaries == oracle."""
    import torch
    lib = ex.load_library()
    m, n, k = 2 * 2048 + 130, 3 * 256 + 37, 8192 + 801
    A = ex.gen_dev("fpuniform_signed", m * k, 181, 12, 6)
    B = ex.gen_dev("fpuniform_signed", k * n, 182, 10, 5)
    C0 = ex.gen_dev("fpuniform_signed", m * n, 183, 10, 5)
    outs = {}
    try:
        for path in (4, 2):
            lib.exblas_set_gemm_path(path)
            C = C0.clone()
            ex.exgemm_dev("N", "N", m, n, k, -0.5, A, k, B, n, 2.0, C, n, 8, True)
            torch.cuda.synchronize()
            assert gemm_info(lib)[0] == path
            outs[path] = C
    finally:
        lib.exblas_set_gemm_path(0)
    assert torch.equal(outs[4].view(torch.int64), outs[2].view(torch.int64))
    ha, hb, hc0, got = A.cpu().numpy(), B.cpu().numpy(), C0.cpu().numpy(), outs[4].cpu().numpy()
    for r0 in (2045, 4094, m - 5):
        rows = min(5, m - r0)
        want = oracle.exgemm("N", "N", rows, n, k, -0.5, ha[r0 * k:(r0 + rows) * k], k, hb, n, 2.0,
                             hc0[r0 * n:(r0 + rows) * n].copy(), n, 0)
        assert (_bits(got[r0 * n:(r0 + rows) * n]) == _bits(want)).all(), r0


def test_workspace_failed_growth_leaves_context_intact(ex, oracle):
    """A reservation that cannot be met must not touch the live workspace: afterwards it is neither parked (a later
    exblas_release_retired_workspaces() would free memory the next call launches into) nor resized, the failed
    hipMalloc leaves no error behind, and calls that fit the old block keep working."""
    import torch
    lib = ex.load_library()
    m, n = 700, 300
    a, x, y0 = oracle.gen("fpuniform_signed", m * n, 51, 10, 5), oracle.gen("fpuniform", n, 52, 10, 0), np.zeros(m)
    want = oracle.exgemv("N", m, n, 1.0, a, m, x, 0.0, y0, 0)
    A, X = torch.from_numpy(a).cuda(), torch.from_numpy(x).cuda()
    Y = torch.zeros(m, dtype=torch.float64, device="cuda")
    ex.exgemv_dev("N", m, n, 1.0, A, m, X, 0.0, Y, 0, False)       # makes sure a workspace exists
    torch.cuda.synchronize()
    before = lib.exblas_workspace_bytes()
    assert before > 0
    rc = lib.exblas_reserve_workspace(1 << 50)                        # a petabyte: must fail
    assert rc != 0
    assert lib.exblas_workspace_bytes() == before
    assert lib.exblas_release_retired_workspaces() == 0               # must not free the live block
    for fpe, ee in ((0, False), (8, True)):
        Y.zero_()
        ex.exgemv_dev("N", m, n, 1.0, A, m, X, 0.0, Y, fpe, ee)
        torch.cuda.synchronize()
        assert (_bits(Y.cpu().numpy()) == _bits(want)).all()
    assert lib.exblas_release_workspace() == 0                        # and no double free
    Y.zero_()
    ex.exgemv_dev("N", m, n, 1.0, A, m, X, 0.0, Y, 8, True)
    torch.cuda.synchronize()
    assert (_bits(Y.cpu().numpy()) == _bits(want)).all()


def test_exgemm_residue_path_long_k_and_capacity(ex, oracle):
    """k > 8192 (one contraction launch per 8192, residues added modulo p), k = 1, and the capacity knob: with fewer
    moduli reserved than the data needs the scalar kernel does the work -- same bits."""
    lib = ex.load_library()
    try:
        lib.exblas_set_gemm_path(4)
        for (m, n, k, kind, p0, p1) in ((33, 65, 9000, "fpuniform_signed", 10, 0), (70, 40, 16500, "ill_cond", 1e16, 0.0),
                                        (50, 60, 1, "fpuniform", 10, 0), (260, 200, 8193, "fpuniform", 3, 0)):
            a = oracle.gen(kind, m * k, 31, p0, p1)
            b = oracle.gen(kind, k * n, 32, p0, p1)
            c0 = oracle.gen("fpuniform_signed", m * n, 33, 10, 5)
            for mode in (oracle.ROUND_EXACT, oracle.ROUND_REFERENCE):
                want = oracle.exgemm("N", "N", m, n, k, 1.0, a, k, b, n, 1.0, c0, n, 0, mode=mode)
                lib.exblas_set_round_mode(mode)
                c = c0.copy()
                ex.exgemm("N", "N", m, n, k, 1.0, a, k, b, n, 1.0, c, n, 8, True)
                info = gemm_info(lib)
                assert info[0] == 4, info
                assert (_bits(c) == _bits(want)).all(), (m, n, k, kind, mode, info, int((c != want).sum()))
        lib.exblas_set_round_mode(0)
        lib.exblas_set_gemm_max_moduli(12)
        m, n, k = 48, 48, 200
        a, b = oracle.gen("ill_cond", m * k, 41, 1e32), oracle.gen("ill_cond", k * n, 42, 1e32)
        want = oracle.exgemm("N", "N", m, n, k, 1.0, a, k, b, n, 0.0, np.zeros(m * n), n, 0)
        c = np.zeros(m * n)
        ex.exgemm("N", "N", m, n, k, 1.0, a, k, b, n, 0.0, c, n, 8, True)
        assert gemm_info(lib)[0] == 0 and (_bits(c) == _bits(want)).all()
        a, b = oracle.gen("naive", m * k, 43), oracle.gen("naive", k * n, 44)      # 53 + 53 + 8 + 2 bits: 15 moduli
        want = oracle.exgemm("N", "N", m, n, k, 1.0, a, k, b, n, 0.0, np.zeros(m * n), n, 0)
        ex.exgemm("N", "N", m, n, k, 1.0, a, k, b, n, 0.0, c, n, 8, True)
        assert gemm_info(lib)[0] == 0 and (_bits(c) == _bits(want)).all()
        lib.exblas_set_gemm_max_moduli(16)
        ex.exgemm("N", "N", m, n, k, 1.0, a, k, b, n, 0.0, c, n, 8, True)
        assert gemm_info(lib)[0] == 4 and (_bits(c) == _bits(want)).all()
    finally:
        lib.exblas_set_gemm_max_moduli(0)
        lib.exblas_set_gemm_path(0)
        lib.exblas_set_round_mode(0)


@pytest.mark.parametrize("m,n,k", [(64, 64, 512), (130, 75, 1100), (16, 200, 33)])
def test_exgemm_slice_paths_are_exact(ex, oracle, m, n, k):
    """The int8-slice path (blas3_i8.hip, default), the fp64-slice path (blas3_mfma.hip, mode 3) and the scalar
    kernel (mode 1) against the oracle, bit for bit, with the digit counts each data family must take."""
    lib = ex.load_library()
    rng = np.random.default_rng(5)
    # name: (A, B, 21-bit fp64 slices [0 = refused], int8 digits of A, of B)
    cases = {
        "fpuniform_r10": (oracle.gen("fpuniform", m * k, 81, 10, 0), oracle.gen("fpuniform", k * n, 82, 10, 0), 3, 8, 8),
        "fpuniform_r17": (oracle.gen("fpuniform", m * k, 91, 17, 0), oracle.gen("fpuniform", k * n, 92, 17, 0), 4, 9, 9),
        "signed_r20": (oracle.gen("fpuniform_signed", m * k, 83, 20, 10), oracle.gen("fpuniform_signed", k * n, 84, 20, 10), 4, 10, 10),
        "naive": (oracle.gen("naive", m * k, 1), oracle.gen("naive", k * n, 1), 3, 7, 7),
        "small_ints": (rng.integers(-1000, 1000, m * k).astype(np.float64), rng.integers(-1000, 1000, k * n).astype(np.float64), 2, 2, 2),
        "wide_r60": (oracle.gen("fpuniform_signed", m * k, 85, 60, 30), oracle.gen("fpuniform_signed", k * n, 86, 60, 30), 0, 15, 15),
    }
    try:
        for name, (a, b, f64_slices, da, db) in cases.items():
            c0 = oracle.gen("fpuniform_signed", m * n, 87, 10, 5)
            want = oracle.exgemm("N", "N", m, n, k, 1.0, a, k, b, n, 1.0, c0, n, 0)
            for path, fpe, ee in ((0, 8, True), (0, 0, False), (3, 8, True), (1, 8, True)):
                lib.exblas_set_gemm_path(path)
                c = c0.copy()
                ex.exgemm("N", "N", m, n, k, 1.0, a, k, b, n, 1.0, c, n, fpe, ee)
                info = gemm_info(lib)
                assert (_bits(c) == _bits(want)).all(), (name, path, fpe, info, int((c != want).sum()))
                if path == 0:
                    # (nearly) equal operand widths are padded to a common, evenly splitting digit count
                    assert info[0] == 2 and max(da, db) - 1 <= max(info[1], info[2]) <= max(da, db) + 1, \
                        (name, info, da, db)
                elif path == 3:
                    assert info == ((1, f64_slices, f64_slices) if f64_slices else (0, 0, 0)), (name, info)
                else:
                    assert info == (0, 0, 0)
        # transposes / alpha / beta / leading dimensions through the int8 path, both rounding modes
        lib.exblas_set_gemm_path(0)
        for ta, tb in (("N", "T"), ("T", "N"), ("T", "T")):
            lda = (m if ta == "T" else k) + 1
            ldb = (k if tb == "T" else n) + 2
            a = oracle.gen("fpuniform_signed", (k if ta == "T" else m) * lda, 91, 8, 4)
            b = oracle.gen("fpuniform_signed", (n if tb == "T" else k) * ldb, 92, 8, 4)
            c0 = oracle.gen("fpuniform_signed", m * (n + 3), 93, 8, 4)
            for alpha, beta in ((1.0, 1.0), (0.5, 0.0), (-1.25, 2.0)):
                for mode in (oracle.ROUND_EXACT, oracle.ROUND_REFERENCE):
                    want = oracle.exgemm(ta, tb, m, n, k, alpha, a, lda, b, ldb, beta, c0, n + 3, 4, False, mode=mode)
                    lib.exblas_set_round_mode(mode)
                    c = c0.copy()
                    ex.exgemm(ta, tb, m, n, k, alpha, a, lda, b, ldb, beta, c, n + 3, 4, False)
                    assert gemm_info(lib)[0] == 2
                    assert (_bits(c) == _bits(want)).all(), (ta, tb, alpha, beta, mode)
    finally:
        lib.exblas_set_gemm_path(0)
        lib.exblas_set_round_mode(0)


def test_exgemm_i8_multi_pass(ex, oracle):
    """More than 8 digits per operand (2 x 2 digit-block passes into the 320-bit accumulators) and k > 8192 (k passes):
    ill-conditioned operands (init_ill_cond, c = 1e32: 14 digits), negative-heavy reference rounding, odd shapes."""
    lib = ex.load_library()
    try:
        for (m, n, k, kind, p0, p1) in ((70, 66, 300, "ill_cond", 1e32, 0.0), (33, 65, 9000, "fpuniform_signed", 10, 0),
                                        (40, 40, 8300, "ill_cond", 1e16, 0.0), (129, 64, 64, "fpuniform_signed", 50, 25)):
            a = oracle.gen(kind, m * k, 31, p0, p1)
            b = oracle.gen(kind, k * n, 32, p0, p1)
            c0 = oracle.gen("fpuniform_signed", m * n, 33, 10, 5)
            for mode in (oracle.ROUND_EXACT, oracle.ROUND_REFERENCE):
                want = oracle.exgemm("N", "N", m, n, k, 1.0, a, k, b, n, 1.0, c0, n, 0, mode=mode)
                lib.exblas_set_round_mode(mode)
                c = c0.copy()
                ex.exgemm("N", "N", m, n, k, 1.0, a, k, b, n, 1.0, c, n, 8, True)
                info = gemm_info(lib)
                assert info[0] == 2 and (max(info[1], info[2]) > 8 or k > 8192), (kind, info)
                assert (_bits(c) == _bits(want)).all(), (m, n, k, kind, mode, info, int((c != want).sum()))
        # capacity knob: with at most 8 digits reserved, 14-digit data must take the scalar kernel -- same bits
        lib.exblas_set_round_mode(0)
        lib.exblas_set_gemm_max_slices(8)
        m, n, k = 48, 48, 200
        a, b = oracle.gen("ill_cond", m * k, 41, 1e32), oracle.gen("ill_cond", k * n, 42, 1e32)
        want = oracle.exgemm("N", "N", m, n, k, 1.0, a, k, b, n, 0.0, np.zeros(m * n), n, 0)
        c = np.zeros(m * n)
        ex.exgemm("N", "N", m, n, k, 1.0, a, k, b, n, 0.0, c, n, 8, True)
        assert gemm_info(lib)[0] == 0 and (_bits(c) == _bits(want)).all()
    finally:
        lib.exblas_set_gemm_max_slices(0)
        lib.exblas_set_round_mode(0)


def test_exgemm_is_stream_ordered_and_capturable(ex, oracle):
    """exblas_exgemm_dev is a pure sequence of launches (slice counts and the path are decided on the device): it can
    be captured into a hipGraph and replayed on new data -- including data that takes a different path on replay."""
    import torch
    lib = ex.load_library()
    m, n, k = 192, 160, 320
    A = ex.gen_dev("fpuniform", m * k, 51, 10, 0)
    B = ex.gen_dev("fpuniform", k * n, 52, 10, 0)
    C = torch.zeros(m * n, dtype=torch.float64, device="cuda")
    ex.exgemm_dev("N", "N", m, n, k, 1.0, A, k, B, n, 0.0, C, n, 8, True)    # sizes the workspace outside the capture
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s):
            ex.exgemm_dev("N", "N", m, n, k, 1.0, A, k, B, n, 0.0, C, n, 8, True)
    datasets = [("fpuniform", 10, 0), ("ill_cond", 1e32, 0), ("naive", 0, 0), ("lognormal", 0.0, 40.0)]
    for kind, p0, p1 in datasets:                # int8 x 8 digits, int8 x 14 digits (multi-pass), 7 digits, scalar kernel
        A.copy_(ex.gen_dev(kind, m * k, 61, p0, p1))
        B.copy_(ex.gen_dev(kind, k * n, 62, p0, p1))
        C.fill_(-1.0)
        g.replay()
        torch.cuda.synchronize()
        want = oracle.exgemm("N", "N", m, n, k, 1.0, A.cpu().numpy(), k, B.cpu().numpy(), n, 0.0, np.zeros(m * n), n, 0)
        assert (_bits(C.cpu().numpy()) == _bits(want)).all(), kind
    assert gemm_info(lib)[0] == 0                # the last data set (log-normal, sigma 40) took the scalar kernel
    # the same through the residue path (shapes from 192 x 192 up take it by default)
    m2, n2, k2 = 256, 200, 320
    A2 = ex.gen_dev("fpuniform", m2 * k2, 51, 10, 0)
    B2 = ex.gen_dev("fpuniform", k2 * n2, 52, 10, 0)
    C2 = torch.zeros(m2 * n2, dtype=torch.float64, device="cuda")
    ex.exgemm_dev("N", "N", m2, n2, k2, 1.0, A2, k2, B2, n2, 0.0, C2, n2, 8, True)
    torch.cuda.synchronize()
    assert gemm_info(lib)[0] == 4
    g3 = torch.cuda.CUDAGraph()
    with torch.cuda.stream(s):
        with torch.cuda.graph(g3, stream=s):
            ex.exgemm_dev("N", "N", m2, n2, k2, 1.0, A2, k2, B2, n2, 0.0, C2, n2, 8, True)
    for kind, p0, p1 in datasets:                # 18 moduli, 32, 15, scalar kernel
        A2.copy_(ex.gen_dev(kind, m2 * k2, 61, p0, p1))
        B2.copy_(ex.gen_dev(kind, k2 * n2, 62, p0, p1))
        C2.fill_(-1.0)
        g3.replay()
        torch.cuda.synchronize()
        want = oracle.exgemm("N", "N", m2, n2, k2, 1.0, A2.cpu().numpy(), k2, B2.cpu().numpy(), n2, 0.0,
                             np.zeros(m2 * n2), n2, 0)
        assert (_bits(C2.cpu().numpy()) == _bits(want)).all(), kind
    # growing the workspace during a capture is refused with an error instead of reallocating under the graph
    big = 1024
    Ab = ex.gen_dev("fpuniform", big * big, 53, 10, 0)
    Cb = torch.zeros(big * big, dtype=torch.float64, device="cuda")
    del g, g3                                    # the graphs captured above die with the workspace they point into
    assert lib.exblas_release_workspace() == 0   # whatever earlier tests left reserved: the next call has to allocate
    g2 = torch.cuda.CUDAGraph()
    with pytest.raises(RuntimeError):
        with torch.cuda.stream(s):
            with torch.cuda.graph(g2, stream=s):
                ex.exgemm_dev("N", "N", big, big, big, 1.0, Ab, big, Ab, big, 0.0, Cb, big, 8, True)
    torch.cuda.synchronize()


def test_standalone_cpp_gemv_gemm_caller(ex):
    """reference-style C++ program for exgemv / exgemm, linked only against libexblas.so"""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.run(["make", "-C", os.path.join(root, "tests", "cpp")], check=True, capture_output=True)
    for argv in (["512", "384"], ["200", "700"]):
        r = subprocess.run([os.path.join(root, "tests", "cpp", "test_exgemv_gpu"), *argv], capture_output=True,
                           text=True, timeout=600)
        assert r.returncode == 0 and "TestPassed; ALL OK!" in r.stdout, (argv, r.stdout[-2000:], r.stderr[-2000:])
    # the -DEXBLAS_VS_MPFR build (tests/test.exgemv.gpu.cpp:35-103, test.exgemm.gpu.cpp:53-125): bit-equal to MPFR
    exe = os.path.join(root, "tests", "cpp", "test_exgemv_gpu_mpfr")
    if os.path.exists(exe):
        r = subprocess.run([exe, "300", "260"], capture_output=True, text=True, timeout=900)
        assert r.returncode == 0 and "TestPassed; ALL OK!" in r.stdout and "exgemm 256^3 == MPFR" in r.stdout, \
            (r.stdout[-2000:], r.stderr[-2000:])


def test_full_size_configs_stripes(ex, oracle):
    """BASELINE configs 4 and 5 at full size: ExGEMV m=n=32768 (column-major, alpha=beta=1) and ExGEMM n=8192
    (row-major).  ExGEMV: EVERY output of one 256-row stripe ('N') and of one 256-column stripe ('T') against the
    oracle.  ExGEMM: every output of one 256-row stripe (2M outputs) produced by the fast path equals the stripe
    produced by the independent scalar kernel (TwoProd + expansions + per-thread superaccumulator), and a 256 x 48
    block of that stripe equals the oracle."""
    import torch
    lib = ex.load_library()
    m = n = 32768
    a = ex.gen_dev("fpuniform", m * n, 11, 10.0, 0.0)
    x = ex.gen_dev("fpuniform", n, 12, 10.0, 0.0)
    y0 = ex.gen_dev("fpuniform", m, 13, 10.0, 0.0)
    y = y0.clone()
    ex.exgemv_dev("N", m, n, 1.0, a, m, x, 1.0, y, 8, True)
    yt = y0.clone()
    ex.exgemv_dev("T", m, n, 1.0, a, m, x, 1.0, yt, 4, True)
    hx, hy0, hy, hyt = x.cpu().numpy(), y0.cpu().numpy(), y.cpu().numpy(), yt.cpu().numpy()
    A = a.view(n, m)                      # column-major: A[k, i] = A(i, k)
    i0 = 20224                            # stripe of rows / columns i0 .. i0+255
    rows = A[:, i0:i0 + 256].contiguous().cpu().numpy().reshape(-1)     # 256 x n block, column-major, lda = 256
    want = oracle.exgemv("N", 256, n, 1.0, rows, 256, hx, 1.0, hy0[i0:i0 + 256].copy(), 0)
    assert (_bits(hy[i0:i0 + 256]) == _bits(want)).all(), "gemv N stripe"
    cols = A[i0:i0 + 256, :].contiguous().cpu().numpy().reshape(-1)     # m x 256 block, column-major, lda = m
    want_t = oracle.exgemv("T", m, 256, 1.0, cols, m, hx, 1.0, hy0[i0:i0 + 256].copy(), 0)
    assert (_bits(hyt[i0:i0 + 256]) == _bits(want_t)).all(), "gemv T stripe"
    del a, A
    N = 8192
    Am = ex.gen_dev("fpuniform", N * N, 14, 10.0, 0.0)
    Bm = ex.gen_dev("fpuniform", N * N, 15, 10.0, 0.0)
    # alpha = beta = 1 as in the reference test (tests/test.exgemm.gpu.cpp:183-184): C += Round(A B)
    C0 = ex.gen_dev("fpuniform_signed", N * N, 18, 10.0, 5.0)
    C = C0.clone()
    assert lib.exblas_release_workspace() == 0             # measure what THIS call reserves
    ex.exgemm_dev("N", "N", N, N, N, 1.0, Am, N, Bm, N, 1.0, C, N, 8, True)
    assert lib.exblas_last_gemm_slices() >= 2              # the fast (MFMA) path ran
    info = gemm_info(lib)
    assert info[0] == 4 and info[4] == 39, info            # residues, reserved for every input the path accepts
    # footprint: 39 bytes per entry of B, of a 2048-row chunk of A and of C (was 39 x 3 x 64 MiB = 7.3 GiB)
    assert lib.exblas_workspace_bytes() <= 4 * 10**9, lib.exblas_workspace_bytes()
    r0 = 5120                                              # a stripe inside the third of the four row chunks
    stripe = C.view(N, N)[r0:r0 + 256].clone()
    Cs = C0[r0 * N:(r0 + 256) * N].clone()
    lib.exblas_set_gemm_path(1)                            # scalar kernel only
    try:
        ex.exgemm_dev("N", "N", 256, N, N, 1.0, Am[r0 * N:], N, Bm, N, 1.0, Cs, N, 8, True)
        assert lib.exblas_last_gemm_slices() == 0
    finally:
        lib.exblas_set_gemm_path(0)
    assert torch.equal(stripe.view(-1).view(torch.int64), Cs.view(torch.int64)), "gemm stripe: fast path != scalar kernel"
    j0 = 4000
    hc0 = C0.view(N, N)
    for rb in (r0, 2040, N - 256):                         # blocks across a chunk boundary and at the end
        blk = C.view(N, N)[rb:rb + 256, j0:j0 + 48].contiguous().cpu().numpy().reshape(-1)
        Ab = Am.view(N, N)[rb:rb + 256].contiguous().cpu().numpy().reshape(-1)
        Bb = Bm.view(N, N)[:, j0:j0 + 48].contiguous().cpu().numpy().reshape(-1)
        c0b = hc0[rb:rb + 256, j0:j0 + 48].contiguous().cpu().numpy().reshape(-1)
        wantc = oracle.exgemm("N", "N", 256, 48, N, 1.0, Ab, N, Bb, 48, 1.0, c0b, 48, 0)
        assert (_bits(blk) == _bits(wantc)).all(), ("gemm block vs oracle", rb)


def same_bits(x, y):
    return np.float64(x).view(np.int64) == np.float64(y).view(np.int64)


@pytest.mark.parametrize("path", [0, 4])
def test_exgemm_mfma_fallbacks(ex, oracle, path):
    """Inputs the int8 paths (digit slices: mode 0 at this size; residues: mode 4) must refuse ON THE DEVICE (the predicated scalar kernel then does the work):
    subnormals, Inf/NaN, huge / tiny exponents; plus all-zero operands, rows/columns of zeros and signed zeros inside
    the fast path.  Always the oracle's bits."""
    lib = ex.load_library()
    m, n, k = 48, 40, 96
    base_a = oracle.gen("fpuniform_signed", m * k, 71, 8, 4)
    base_b = oracle.gen("fpuniform_signed", k * n, 72, 8, 4)
    c0 = np.zeros(m * n)

    def run(a, b, want_fast):
        # the oracle (like the reference) has no defined behaviour on Inf/NaN: take its exact result on the finite
        # part and IEEE's answer (numpy's matmul class: +-Inf or NaN) for the entries a non-finite value reaches
        fa, fb = np.where(np.isfinite(a), a, 0.0), np.where(np.isfinite(b), b, 0.0)
        want = oracle.exgemm("N", "N", m, n, k, 1.0, fa, k, fb, n, 0.0, c0, n, 0)
        with np.errstate(all="ignore"):
            ieee = (a.reshape(m, k) @ b.reshape(k, n)).reshape(-1)
        if not (np.isfinite(a).all() and np.isfinite(b).all()):
            want = np.where(np.isfinite(ieee), want, ieee)
        c = c0.copy()
        lib.exblas_set_gemm_path(path)
        try:
            ex.exgemm("N", "N", m, n, k, 1.0, a, k, b, n, 0.0, c, n, 8, True)
            used = gemm_info(lib)
        finally:
            lib.exblas_set_gemm_path(0)
        assert want_fast is None or (used[0] == (4 if path == 4 else 2)) == want_fast, used
        ok = (_bits(c) == _bits(want)) | (np.isnan(c) & np.isnan(want))
        assert ok.all(), np.nonzero(~ok)[0][:5]

    a = base_a.copy(); a[5] = 5e-324
    run(a, base_b, False)                                  # subnormal entry
    a = base_a.copy(); a[7] = np.inf
    run(a, base_b, False)                                  # Inf
    b = base_b.copy(); b[3] = np.nan
    run(base_a, b, False)                                  # NaN
    run(base_a * 2.0**500, base_b, False)                  # exponents outside +-300
    run(base_a * 2.0**-305, base_b, False)
    run(base_a * 2.0**280, base_b * 2.0**280, True)        # inside: products up to 2^570, still exact and normal
    run(base_a, np.zeros(k * n), True)                     # all-zero operand: one all-zero digit plane, exact zeros
    run(np.zeros(m * k), np.zeros(k * n), True)            # nothing but zeros
    a = base_a.copy().reshape(m, k); a[3, :] = 0.0; a[10, ::2] = -0.0
    b = base_b.copy().reshape(k, n); b[:, 7] = 0.0
    run(a.reshape(-1), b.reshape(-1), True)                # zero row / zero column / signed zeros: fast path


def test_gemv_gemm_randomized_soak(ex):
    """tools/stress_blas23.py: 180 random shapes / transposes / alpha, beta / leading dimensions / strides / offsets /
    variants / data families for ExGEMV and ExGEMM (scalar and MFMA paths): bits equal to the oracle"""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "stress_blas23.py"), "180", "11"],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "0 mismatches" in r.stdout, (r.stdout[-3000:], r.stderr[-2000:])


def test_exgemm_mixed_digit_counts(ex, oracle):
    """operands that need different numbers of digits (A: 16-bit integers times powers of two, B: full 53-bit
    mantissas over 2^10) cost sa*sb matrix multiply-adds per element pair, not max^2: same bits as the oracle"""
    lib = ex.load_library()
    rng = np.random.default_rng(5)
    m, n, k = 96, 130, 700
    small = rng.integers(-30000, 30001, size=m * k).astype(np.float64) * 2.0 ** rng.integers(-3, 4, size=m * k)
    full = oracle.gen("fpuniform_signed", k * n, 77, 10, 0)
    c0 = oracle.gen("fpuniform_signed", m * n, 78, 10, 0)
    try:
        for a, b, mm, nn, swap in ((small, full, m, n, False), (full[:n * k], small[:k * m], n, m, True)):
            want = oracle.exgemm("N", "N", mm, nn, k, 1.0, a, k, b, nn, 1.0, c0[:mm * nn], nn, 0)
            for path in (0, 3):
                lib.exblas_set_gemm_path(path)
                c = c0[:mm * nn].copy()
                ex.exgemm("N", "N", mm, nn, k, 1.0, a, k, b, nn, 1.0, c, nn, 8, True)
                info = gemm_info(lib)
                if path == 0:
                    da, db = (info[2], info[1]) if swap else (info[1], info[2])
                    assert info[0] == 2 and da <= 4 and db == 8, info
                else:
                    assert info == (1, 3, 3), info
                assert (_bits(c) == _bits(want)).all(), (path, swap)
    finally:
        lib.exblas_set_gemm_path(0)


def test_expansion_sizes_above_8_gemv_gemm(ex, oracle):
    """fpe > 8: without early exit the same exact result (ExGEMV.cpp:103-104, ExGEMM.cpp:96-97); with early exit the
    reference falls through to `return` and leaves y / C untouched -- for every data set, whichever ExGEMM path
    (MFMA slices or scalar kernel) the data would have qualified for."""
    m, n, k = 70, 52, 45
    a = oracle.gen("fpuniform", m * n, 91, 10, 0)
    x = oracle.gen("fpuniform", n, 92, 10, 0)
    y0 = oracle.gen("fpuniform", m, 93, 10, 0)
    want = oracle.exgemv("N", m, n, 1.0, a, m, x, 1.0, y0, 0)
    A = oracle.gen("fpuniform", m * k, 94, 10, 0)
    B = oracle.gen("fpuniform", k * n, 95, 10, 0)
    C0 = oracle.gen("fpuniform", m * n, 96, 10, 0)
    wantc = oracle.exgemm("N", "N", m, n, k, 1.0, A, k, B, n, 1.0, C0, n, 0)
    for fpe in (9, 12):
        y = y0.copy()
        ex.exgemv("N", m, n, 1.0, a, m, 0, x, 1, 0, 1.0, y, 1, 0, fpe, False)
        assert (_bits(y) == _bits(want)).all()
        y = y0.copy()
        ex.exgemv("N", m, n, 1.0, a, m, 0, x, 1, 0, 1.0, y, 1, 0, fpe, True)
        assert (_bits(y) == _bits(y0)).all()
        c = C0.copy()
        ex.exgemm("N", "N", m, n, k, 1.0, A, k, B, n, 1.0, c, n, fpe, False)
        assert (_bits(c) == _bits(wantc)).all()
        c = C0.copy()
        ex.exgemm("N", "N", m, n, k, 1.0, A, k, B, n, 1.0, c, n, fpe, True)
        assert (_bits(c) == _bits(C0)).all()


def test_exgemm_randomized_soak(ex):
    """tools/stress_gemm.py: 120 random ExGEMM cases -- shapes with ragged tiles and k across the 8192-per-pass
    boundary, transposes, leading dimensions, alpha/beta, independently chosen operand families (every digit count
    1..16 and pairing: unrolled bodies, generic body, multi-pass, scalar fallback), both rounding modes, occasional
    subnormal / huge entries: bits equal to the oracle (a 700-case run of the same tool: 0 mismatches)"""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "stress_gemm.py"), "120", "7"],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "0 mismatches" in r.stdout, (r.stdout[-3000:], r.stderr[-2000:])
    # the same cases with the residue path forced at every shape (by default it serves min(m, n) >= 192 only)
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "stress_gemm.py"), "120", "8"],
                       capture_output=True, text=True, timeout=900, env=dict(os.environ, EXBLAS_GEMM_PATH="4"))
    assert r.returncode == 0 and "0 mismatches" in r.stdout, (r.stdout[-3000:], r.stderr[-2000:])
