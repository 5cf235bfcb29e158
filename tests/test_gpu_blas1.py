"""GPU parity tests for ExSUM / ExDOT: the HIP path (through the C ABI) against the oracle.

Bit-exact bar: canonical limbs equal, both roundings equal, for every (fpe, early_exit) variant."""
import numpy as np
import pytest

from helpers import (FPE_VARIANTS_DOT, FPE_VARIANTS_SUM, exact_int_from_canon, exact_int_from_digits, golden_cases,
                     load_golden, same_double)

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ex():
    import torch
    import exblas_amd
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    exblas_amd.load_library().exblas_hip_init(-1)
    return exblas_amd


def _check_record(rec, limbs, exact, refmode, what):
    assert (rec.canon == limbs).all(), (what, "canonical limbs differ")
    assert same_double(rec.exact, exact), (what, rec.exact, exact)
    assert same_double(rec.refmode, refmode), (what, rec.refmode, refmode)
    # the all-reduce payload holds the same integer (canonical LSB is 2^-1092 = 2^-1074 >> 18)
    assert exact_int_from_digits(rec.digits) << 18 == exact_int_from_canon(rec.canon), what


def test_exsum_golden_all_variants(ex):
    g = load_golden("exsum_golden.npz")
    for name, i, a in golden_cases(g, "data"):
        for fpe, ee in FPE_VARIANTS_SUM:
            rec = ex.exsum_record(a.size, a if a.size else np.zeros(1), 1, 0, fpe, ee)
            _check_record(rec, g["limbs"][i], g["mpfr"][i], g["refmode"][i], (name, fpe, ee))


def test_exdot_golden_all_variants(ex):
    g = load_golden("exdot_golden.npz")
    for name, i, a, b in golden_cases(g, "a", "b"):
        for fpe, ee in FPE_VARIANTS_DOT:
            rec = ex.exdot_record(a.size, a, 1, 0, b, 1, 0, fpe, ee)
            assert (rec.canon == g["limbs"][i]).all(), (name, fpe, ee)
            assert same_double(rec.exact, g["mpfr"][i]), (name, fpe, ee)


@pytest.mark.parametrize("kind,p0,p1", [("naive", 0, 0), ("fpuniform", 10, 0), ("lognormal", 0.0, 2.0),
                                         ("lognormal", 0.0, 50.0), ("ill_cond", 1e32, 0.0), ("cancel", 50.0, 0.0),
                                         ("fpuniform_signed", 1000.0, 500.0)])
def test_exsum_vs_oracle_mid_sizes(ex, oracle, kind, p0, p1):
    import torch
    for n in (100003, 1 << 20, (1 << 21) + 5):
        x = ex.gen_dev(kind, n, 7, p0, p1)
        host = oracle.gen(kind, n, 7, p0, p1)
        assert (x.cpu().numpy().view(np.int64) == host.view(np.int64)).all(), "GPU generator differs from oracle's"
        r0, l0 = oracle.exsum_omp(host, 8, True, 8, limbs=True)
        rref = oracle.round_limbs(l0, oracle.ROUND_REFERENCE)
        for fpe, ee in FPE_VARIANTS_SUM:
            rec = ex.read_record(ex.exsum_dev(x, fpe, ee))
            _check_record(rec, l0, r0, rref, (kind, n, fpe, ee))
        torch.cuda.synchronize()


@pytest.mark.parametrize("kind,p0,p1", [("naive", 0, 0), ("lognormal", 0.0, 2.0), ("ill_cond", 1e32, 0.0),
                                         ("fpuniform_signed", 400.0, 200.0)])
def test_exdot_vs_oracle_mid_sizes(ex, oracle, kind, p0, p1):
    for n in (100003, 1 << 20):
        x, y = ex.gen_dev(kind, n, 8, p0, p1), ex.gen_dev(kind, n, 9, p0, p1)
        ha, hb = oracle.gen(kind, n, 8, p0, p1), oracle.gen(kind, n, 9, p0, p1)
        r0, l0 = oracle.exdot_omp(ha, hb, 8, True, 8, limbs=True)
        for fpe, ee in FPE_VARIANTS_DOT:
            rec = ex.read_record(ex.exdot_dev(x, y, fpe, ee))
            assert (rec.canon == l0).all(), (kind, n, fpe, ee)
            assert same_double(rec.exact, r0), (kind, n, fpe, ee)


def test_strided_offset_misaligned(ex, oracle):
    a = oracle.gen("ill_cond", 50001, 5, 1e32)
    b = oracle.gen("lognormal", 50001, 6, 0.0, 2.0)
    for inca, off in ((1, 0), (1, 1), (1, 3), (2, 0), (3, 1), (7, 5)):
        n = (a.size - off + inca - 1) // inca
        r0, l0 = oracle.exsum(a, 0, inca=inca, offset=off, n=n, limbs=True)
        d0, m0 = oracle.exdot(a, b, 0, inca=inca, offa=off, incb=inca, offb=off, n=n, limbs=True)
        for fpe, ee in ((0, False), (4, False), (8, True)):
            rec = ex.exsum_record(n, a, inca, off, fpe, ee)
            assert (rec.canon == l0).all() and same_double(rec.exact, r0), (inca, off, fpe, ee)
            rec = ex.exdot_record(n, a, inca, off, b, inca, off, fpe, ee)
            assert (rec.canon == m0).all() and same_double(rec.exact, d0), (inca, off, fpe, ee)


def test_reference_api_semantics(ex, oracle):
    a = oracle.gen("lognormal", 4097, 2, 0.0, 2.0)
    want = oracle.exsum(a, 0)
    assert ex.exsum(a.size, a, 1, 0, 0) == want
    assert ex.exsum(a.size, a, 1, 0, 8, True, True) == want
    assert ex.exsum(a.size, a, 1, 0, 9, True) == 0.0      # early_exit with fpe > 8 -> 0.0 (gpu:ExSUM.cpp:72-83)
    assert ex.exsum(a.size, a, 1, 0, 9) == want           # no early exit: ExSUM.FPE.cl with NBFPE = 9, same value (:80-81)
    assert ex.exsum(0, a, 1, 0, 4) == 0.0
    assert ex.exdot(0, a, 1, 0, a, 1, 0, 4) == 0.0        # ExDOT.cpp:70-71
    assert ex.exdot(a.size, a, 1, 0, a, 1, 0, 5) == oracle.exdot(a, a, 0)
    lib = ex.load_library()
    lib.exblas_set_round_mode(1)
    try:
        neg = np.array([-1.5])
        assert ex.exsum(1, neg, 1, 0, 0) == oracle.exsum(neg, 0, mode=oracle.ROUND_REFERENCE) != -1.5
    finally:
        lib.exblas_set_round_mode(0)
    assert ex.exsum(1, np.array([-1.5]), 1, 0, 0) == -1.5


def test_nonfinite_policy(ex):
    inf, nan = np.inf, np.nan
    base = np.ones(5000)
    for fpe, ee in ((0, False), (4, False), (8, True)):
        for vals, want in (([inf], inf), ([-inf], -inf), ([inf, -inf], nan), ([nan], nan), ([inf, inf], inf)):
            a = base.copy()
            a[100:100 + len(vals)] = vals
            got = ex.exsum(a.size, a, 1, 0, fpe, ee)
            assert (np.isnan(want) and np.isnan(got)) or got == want, (fpe, ee, vals, got)
        a = np.full(4096, np.finfo(np.float64).max)
        b = np.full(4096, 2.0)
        assert ex.exdot(a.size, a, 1, 0, b, 1, 0, fpe if fpe else 0, ee) == inf


def test_full_size_properties(ex, oracle):
    """n = 2^28 (BASELINE config): size-independent properties + one full oracle comparison."""
    import torch
    n = 1 << 28
    x = ex.gen_dev("cancel", n, 1, 50.0)
    rec = ex.read_record(ex.exsum_dev(x, 8, True))
    assert rec.exact == 1.0                       # exact sum is 1 + 2^-60 by construction
    assert exact_int_from_digits(rec.digits) == (1 << 1074) + (1 << 1014)
    del x
    x = ex.gen_dev("ill_cond", n, 1, 1e32)
    recs = [ex.read_record(ex.exsum_dev(x, fpe, ee)) for fpe, ee in ((8, True), (4, True), (0, False), (3, False))]
    for r in recs[1:]:
        assert (r.canon == recs[0].canon).all() and same_double(r.exact, recs[0].exact)
    # linearity: sum(x) == sum(x[:k]) + sum(x[k:]) as exact integers, for an odd split
    k = 123456789
    p1 = ex.read_record(ex.exsum_dev(x[:k], 8, True))
    p2 = ex.read_record(ex.exsum_dev(x[k:], 6, True))
    assert exact_int_from_digits(p1.digits) + exact_int_from_digits(p2.digits) == exact_int_from_digits(recs[0].digits)
    # the whole vector against the CPU oracle (OpenMP, FPE8-EE)
    host = x.cpu().numpy()
    r0, l0 = oracle.exsum_omp(host, 8, True, 16, limbs=True)
    assert (recs[0].canon == l0).all() and same_double(recs[0].exact, r0)
    # dot of the vector with a second one, against the oracle
    y = ex.gen_dev("ill_cond", n, 2, 1e32)
    d = ex.read_record(ex.exdot_dev(x, y, 8, True))
    d2 = ex.read_record(ex.exdot_dev(x, y, 0, False))
    assert (d.canon == d2.canon).all()
    hy = y.cpu().numpy()
    r1, l1 = oracle.exdot_omp(host, hy, 8, True, 16, limbs=True)
    assert (d.canon == l1).all() and same_double(d.exact, r1)
    torch.cuda.synchronize()


def test_standalone_cpp_caller(ex):
    """The reference-style C++ test program, linked only against libexblas.so (no Python in the process)."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "tests", "cpp", "test_exsum_gpu")
    subprocess.run(["make", "-C", os.path.join(root, "tests", "cpp")], check=True, capture_output=True)
    for argv in (["20"], ["20", "50", "0"], ["20", "1e32", "0", "i"], ["18", "2", "0", "n"]):
        r = subprocess.run([exe, *argv], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0 and "TestPassed; ALL OK!" in r.stdout, (argv, r.stdout[-2000:], r.stderr[-2000:])
    # the -DEXBLAS_VS_MPFR build of the same caller (tests/test.exsum.gpu.cpp:20-38,:118-133): every variant == MPFR
    if os.path.exists(exe + "_mpfr"):
        for argv in (["18"], ["18", "50", "0"], ["18", "1e32", "0", "i"], ["16", "50", "0", "n"]):
            r = subprocess.run([exe + "_mpfr", *argv], capture_output=True, text=True, timeout=600)
            assert r.returncode == 0 and "TestPassed; ALL OK!" in r.stdout and "exsum with MPFR" in r.stdout, \
                (argv, r.stdout[-2000:], r.stderr[-2000:])


def test_concurrent_cpp_callers(ex):
    """pthread-style concurrent use of the host-pointer API (the reference's RNGExample pattern)."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.run(["make", "-C", os.path.join(root, "tests", "cpp")], check=True, capture_output=True)
    r = subprocess.run([os.path.join(root, "tests", "cpp", "test_threads")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "TestPassed; ALL OK!" in r.stdout, (r.stdout[-2000:], r.stderr[-2000:])


def test_exsum_segmented(ex, oracle):
    """Batched exsum over CSR-style segments (empty, 1-element, ragged, longer than a wave) vs the oracle per segment."""
    import torch
    rng = np.random.default_rng(3)
    lens = np.concatenate([[0, 1, 2, 63, 64, 65, 0, 1000, 5000], rng.integers(0, 200, 500)])
    offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    for kind, p0, p1 in (("ill_cond", 1e32, 0.0), ("lognormal", 0.0, 50.0), ("fpuniform_signed", 1500.0, 700.0)):
        vals = oracle.gen(kind, int(offs[-1]), 17, p0, p1)
        want = np.array([oracle.exsum(vals[offs[i]:offs[i + 1]], 0) if lens[i] else 0.0 for i in range(len(lens))])
        dv, do = torch.from_numpy(vals).cuda(), torch.from_numpy(offs).cuda()
        for fpe, ee in FPE_VARIANTS_SUM:
            got = ex.exsum_segmented_dev(dv, do, fpe, ee).cpu().numpy()
            assert (got.view(np.int64) == want.view(np.int64)).all(), (kind, fpe, ee, np.nonzero(got != want)[0][:5])


def test_max_size_int_api_limit(ex):
    """n close to the API's `int` limit (2^31 - 2 elements, 16 GiB): 64-bit indexing, limb headroom (each limb
    takes < 2^31 adds of < 2^32), known exact answers by construction."""
    import torch
    from fractions import Fraction
    n = (1 << 31) - 2
    x = ex.gen_dev("cancel", n, 5, 40.0)
    for fpe, ee in ((8, True), (0, False)):
        rec = ex.read_record(ex.exsum_dev(x, fpe, ee))
        assert rec.exact == 1.0 and exact_int_from_digits(rec.digits) == (1 << 1074) + (1 << 1014), (fpe, ee)
    del x
    x = ex.gen_dev("naive", n, 1)
    want = Fraction(1.1) * n                       # exact value of n copies of the double nearest 1.1
    rec = ex.read_record(ex.exsum_dev(x, 4, True))
    assert Fraction(exact_int_from_digits(rec.digits), 1 << 1074) == want
    rec0 = ex.read_record(ex.exsum_dev(x, 0, False))  # every element adds to the same three limbs: worst case headroom
    assert (rec0.canon == rec.canon).all() and rec0.exact == rec.exact == float(want)
    # three arrays folded into ONE reduction: 3 * (2^31 - 2) values, more than one int64 limb could take if the
    # finalize summed the group accumulators naively (each add moves a limb by up to 2^32 - 1)
    for _ in range(3):
        ex.exsum_accumulate_dev(x, 0, False)
    rec3 = ex.read_record(ex.finish_dev())
    assert Fraction(exact_int_from_digits(rec3.digits), 1 << 1074) == 3 * want
    torch.cuda.synchronize()


def test_hip_graph_capture(ex, oracle):
    """The device-pointer calls are pure stream-ordered launches: capture 8 ExSUM + ExDOT steps in a graph, replay."""
    import torch
    n = (1 << 22) + 2
    x, y = ex.gen_dev("ill_cond", n, 1, 1e32), ex.gen_dev("lognormal", n, 2, 0.0, 2.0)
    recs = [ex.new_record_buffer() for _ in range(16)]
    ex.exsum_dev(x, 8, True, out=recs[0])      # context creation / lazy init outside the capture
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s):
            for i in range(8):
                ex.exsum_dev(x, 8, True, out=recs[2 * i])
                ex.exdot_dev(x, y, 8, True, out=recs[2 * i + 1])
    for r in recs:
        r.zero_()
    g.replay()
    torch.cuda.synchronize()
    hx, hy = oracle.gen("ill_cond", n, 1, 1e32), oracle.gen("lognormal", n, 2, 0.0, 2.0)
    r0, l0 = oracle.exsum_omp(hx, 8, True, 8, limbs=True)
    d0, m0 = oracle.exdot_omp(hx, hy, 8, True, 8, limbs=True)
    for i in range(8):
        a, b = ex.read_record(recs[2 * i]), ex.read_record(recs[2 * i + 1])
        assert (a.canon == l0).all() and a.exact == r0 and (b.canon == m0).all() and b.exact == d0


def test_randomized_small_cases(ex, oracle):
    """300 random (n, offset, stride, distribution, variant) cases incl. zeros, signed zeros, subnormals, huge and
    tiny magnitudes and exact cancellations -- every case bit-exact against the oracle (limbs and double)."""
    rng = np.random.default_rng(20261004)
    specials = np.array([0.0, -0.0, 5e-324, -5e-324, 2.2250738585072014e-308, 1.7976931348623157e308,
                         -1.7976931348623157e308, 1.0, -1.0, 2.0**-1000, 2.0**1000, 2.0**52, 2.0**53 - 1])
    for case in range(300):
        n = int(rng.integers(0, 3000))
        inca = int(rng.choice([1, 1, 1, 2, 3, 5]))
        off = int(rng.integers(0, 4))
        total = off + max(n - 1, 0) * inca + 1 + int(rng.integers(0, 3))
        spread = int(rng.choice([1, 20, 200, 900]))
        a = rng.standard_normal(total) * np.exp2(rng.integers(-spread, spread, total).astype(np.float64))
        k = int(rng.integers(0, max(total // 4, 1)))
        if k:
            a[rng.integers(0, total, k)] = rng.choice(specials, k)
        if total > 8 and rng.random() < 0.5:          # plant exact cancellations
            half = total // 2
            a[half:2 * half] = -a[:half][rng.permutation(half)]
        fpe, ee = FPE_VARIANTS_SUM[int(rng.integers(0, len(FPE_VARIANTS_SUM)))]
        want, limbs = oracle.exsum(a, 0, inca=inca, offset=off, n=n, limbs=True)
        rec = ex.exsum_record(n, a, inca, off, fpe, ee)
        assert (rec.canon == limbs).all() and same_double(rec.exact, want), (case, n, inca, off, fpe, ee)
        if spread <= 200:                             # dot: keep products inside the exact domain
            b = rng.standard_normal(total) * np.exp2(rng.integers(-spread, spread, total).astype(np.float64))
            a2 = np.where(np.abs(a) > 1e250, 1.0, a)
            a2 = np.where((np.abs(a2) < 1e-250) & (a2 != 0), 1.0, a2)
            fd, ed = FPE_VARIANTS_DOT[int(rng.integers(0, len(FPE_VARIANTS_DOT)))]
            wd, ld = oracle.exdot(a2, b, 0, inca=inca, offa=off, incb=inca, offb=off, n=n, limbs=True)
            rd = ex.exdot_record(n, a2, inca, off, b, inca, off, fd, ed)
            assert (rd.canon == ld).all() and same_double(rd.exact, wd), (case, n, inca, off, fd, ed)


def test_error_behaviour_matches_reference(ex):
    """fpe < 0: the host-pointer API prints and exit(1)s like cpu:ExSUM.cpp:25-28; the *_dev layer returns an error."""
    import subprocess
    import sys
    import torch
    x = torch.ones(16, dtype=torch.float64, device="cuda")
    with pytest.raises(RuntimeError):
        ex.exsum_dev(x, fpe=-1)
    with pytest.raises(RuntimeError):
        ex.exdot_dev(x, x, fpe=-3)
    code = ("import numpy as np, exblas_amd as ex\n"
            "ex.exsum(4, np.ones(4), 1, 0, -1)\n"
            "print('not reached')\n")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300,
                       cwd=__import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
    assert r.returncode == 1 and "not reached" not in r.stdout
    assert "Size of floating-point expansion should be a positive number" in r.stderr


def test_host_and_dev_layers_do_not_share_state(ex, oracle):
    """A host-pointer call while a *_dev reduction is still in flight (no synchronisation in between): the host
    layer has its own accumulators / workspace / stream, so neither result is disturbed."""
    n = 1 << 22
    x = ex.gen_dev("ill_cond", n, 5, 1e32)
    h = oracle.gen("lognormal", 300001, 6, 0.0, 2.0)
    want_dev = ex.read_record(ex.exsum_dev(x, 8, True)).exact
    want_host = oracle.exsum(h, 0)
    m, k = 96, 80
    A = oracle.gen("fpuniform_signed", m * k, 7, 40, 20)
    xv = oracle.gen("fpuniform_signed", k, 8, 40, 20)
    want_y = oracle.exgemv("N", m, k, 1.0, A, m, xv, 0.0, np.zeros(m), 0)
    for it in range(12):
        rec = ex.exsum_dev(x, 8, True)                       # enqueued, not waited for
        assert ex.exsum(h.size, h, 1, 0, 8, True) == want_host, it
        y = np.zeros(m)
        ex.exgemv("N", m, k, 1.0, A, m, 0, xv, 1, 0, 0.0, y, 1, 0, 8, True)
        assert (y.view(np.int64) == want_y.view(np.int64)).all(), it
        assert ex.read_record(rec).exact == want_dev, it


def test_expansion_sizes_above_8(ex, oracle):
    """Without early exit the reference builds its FPE kernel with NBFPE = fpe for any fpe (gpu:ExSUM.cpp:80-81,
    ExDOT.cpp:93-94): the result is the same exact value.  With early exit and fpe > 8 it silently returns 0.0."""
    a = oracle.gen("ill_cond", 20000, 9, 1e32)
    b = oracle.gen("lognormal", 20000, 10, 0.0, 2.0)
    s, d = oracle.exsum(a, 0), oracle.exdot(a, b, 0)
    for fpe in (9, 10, 16):
        assert ex.exsum(a.size, a, 1, 0, fpe, False) == s
        assert ex.exdot(a.size, a, 1, 0, b, 1, 0, fpe, False) == d
        assert ex.exsum(a.size, a, 1, 0, fpe, True) == 0.0
        assert ex.exdot(a.size, a, 1, 0, b, 1, 0, fpe, True) == 0.0


def test_host_api_spreads_over_virtual_devices(ex, oracle):
    """The host-pointer exsum / exdot split one call over several "devices" (here: the one GPU listed several times,
    each listing an independent part with its own context, stream and accumulators), stream every part in 64 MiB
    chunks and add the parts' digit sets on the first device: records identical to the single-device *_dev path, for
    contiguous, strided and offset inputs, whatever the device list."""
    import ctypes as C
    import torch
    lib = ex.load_library()
    n = (20 << 20) + 3                                     # > 2 chunks of 8M elements per part even when split in two
    a = oracle.gen("ill_cond", n, 21, 1e32)
    b = oracle.gen("lognormal", n, 22, 0.0, 2.0)
    da, db = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
    want_s = ex.read_record(ex.exsum_dev(da, 8, True))
    want_d = ex.read_record(ex.exdot_dev(da, db, 8, True))
    ns = 6_000_001
    want_ss = ex.read_record(ex.exsum_dev(da[5:], 8, True, inca=3, n=ns))
    want_ds = ex.read_record(ex.exdot_dev(da[5:], db[2:], 8, True, incx=3, incy=2, n=ns))
    try:
        for devs in ([0], [0, 0], [0, 0, 0], [0] * 8):
            arr = (C.c_int * len(devs))(*devs)
            assert lib.exblas_set_host_devices(len(devs), arr) == 0
            for got, want in ((ex.exsum_record(n, a, 1, 0, 8, True), want_s),
                              (ex.exdot_record(n, a, 1, 0, b, 1, 0, 8, True), want_d),
                              (ex.exsum_record(ns, a, 3, 5, 8, True), want_ss),
                              (ex.exdot_record(ns, a, 3, 5, b, 2, 2, 8, True), want_ds)):
                assert got.exact == want.exact and got.refmode == want.refmode and (got.canon == want.canon).all(), devs
            assert ex.exsum(1, a, 1, 0, 8, True) == a[0]                 # fewer elements than parts
            assert ex.exsum(3, a, 1, 0, 0) == oracle.exsum(a[:3], 0)
            bad = a[:1000].copy()
            bad[17] = np.inf
            assert ex.exsum(1000, bad, 1, 0, 8, True) == np.inf          # the non-finite indicators travel with the digit sets
        # products outside the double range, cancelling only ACROSS the parts: the parts export their low / high digit
        # sets and the fold happens once -- the MPFR value and flag bits 3..6 whatever the device list
        m = 40000
        xo = oracle.gen("fpuniform_signed", m, 31, 40, 20)
        yo = oracle.gen("fpuniform_signed", m, 32, 40, 20)
        xo[:m // 4] *= 2.0 ** 560
        yo[:m // 4] *= 2.0 ** 540
        xo[3 * m // 4:] = -xo[:m // 4]
        yo[3 * m // 4:] = yo[:m // 4]
        xo[m // 4:m // 2] *= 2.0 ** -520
        yo[m // 4:m // 2] *= 2.0 ** -500
        want_o = oracle.mpfr_exdot(xo, yo)
        assert np.isfinite(want_o)
        for devs in ([0], [0, 0], [0, 0, 0], [0] * 8):
            assert lib.exblas_set_host_devices(len(devs), (C.c_int * len(devs))(*devs)) == 0
            got = ex.exdot_record(m, xo, 1, 0, yo, 1, 0, 8, True)
            assert got.exact == want_o and got.flags == 8 | 16 | 32 | 64, (devs, got.exact, got.flags)
        assert lib.exblas_set_host_devices(1, (C.c_int * 1)(7)) != 0     # no such device on this box
    finally:
        lib.exblas_set_host_devices(0, None)


def test_blas1_randomized_soak(ex):
    """tools/stress_blas1.py: 150 random ExSUM / ExDOT cases -- lengths from 1 to 12M (below and above the capped, odd
    grids; ragged tails), strides, misaligned offsets, every (fpe, early_exit) variant, independently chosen operand
    families, occasional Inf / NaN / huge / subnormal entries: limbs and rounded value equal to the oracle's (a 2000-case
    run of the same tool: profiles/r02_stress_blas1_2000.log)"""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "stress_blas1.py"), "150", "5"],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "0 mismatches" in r.stdout, (r.stdout[-3000:], r.stderr[-2000:])


FLAG_PUNDER, FLAG_POVER, FLAG_PLOW_EXACT, FLAG_PHIGH_EXACT = 8, 16, 32, 64


def test_exdot_product_domain(ex, oracle):
    """A 106-bit product fits the double-range accumulator only while it neither overflows nor reaches below 2^-1074 -- the
    limit of the reference's kernels (its test oracle sums exact products in 4196 bits for that reason,
    tests/test.exdot.gpu.cpp:24-46).  Here the LOW side is closed: products below 2^-968 are formed again at a scaled
    exponent (error-free) and accumulated in a second accumulator that the finalize folds back, so the result is the
    MPFR-4196 value; the record says so (EXBLAS_OUT_FLAGS bit 3 + bit 5).  The HIGH side likewise (bit 4 + bit 6): products
    of finite operands at or beyond 2^1024 go, scaled down, to a third accumulator.  No bit set = the
    exact domain of round 2: limbs equal to the oracle's, result == MPFR.  Families: products straddling 2^-968, products
    that underflow to zero, zeros (0 * x must not raise anything), overflow; every variant; vector / strided / odd-tail."""
    import torch
    rng = np.random.default_rng(11)
    mp = oracle.mpfr()
    assert mp is not None, "this test needs the MPFR oracle (oracle/libmpfr_oracle.so)"

    def run(a, b, fpe, ee, inca=1):
        nn = (a.size + inca - 1) // inca
        rec = ex.read_record(ex.exdot_dev(torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda(), fpe, ee, incx=inca,
                                          incy=inca, n=nn))
        want = oracle.mpfr_exdot(a, b, inca=inca, incb=inca, n=nn)
        assert same_double(rec.exact, want), (fpe, ee, rec.flags, rec.exact, want)
        if rec.flags == 0:     # inside the exact domain the oracle restates the same arithmetic: limbs equal
            r, limbs = oracle.exdot(a, b, fpe, ee, inca=inca, incb=inca, limbs=True)
            assert (rec.canon == limbs).all() and same_double(rec.exact, r)
        return rec

    n = 20000 + 3                                     # odd: the scalar tail runs too
    mant = lambda k: rng.uniform(1.0, 2.0, k) * rng.choice([-1.0, 1.0], k)  # noqa: E731
    big = np.ldexp(mant(n), rng.integers(-20, 20, n))
    for fpe, ee in FPE_VARIANTS_DOT:
        # (1) safely inside: every product at or above 2^-960 -> no flag
        a = np.ldexp(mant(n), rng.integers(-480, -450, n))
        b = np.ldexp(mant(n), rng.integers(-480, -440, n))
        assert run(a, b, fpe, ee).flags == 0
        # (2) ONE product below 2^-968 among ordinary ones
        a2, b2 = big.copy(), big[::-1].copy()
        j = int(rng.integers(0, n))
        a2[j], b2[j] = np.ldexp(1.5, -500), np.ldexp(1.25, -480)          # 2^-980: error term below 2^-1074
        assert run(a2, b2, fpe, ee).flags == FLAG_PUNDER | FLAG_PLOW_EXACT
        # (3) a product that underflows to zero entirely
        a2[j], b2[j] = np.ldexp(1.0, -600), np.ldexp(1.0, -600)
        assert run(a2, b2, fpe, ee).flags == FLAG_PUNDER | FLAG_PLOW_EXACT
        # (4) zeros are not underflow: 0 * x, x * 0, 0 * 0, subnormal * 0
        a3, b3 = big.copy(), big[::-1].copy()
        a3[::7] = 0.0
        b3[::5] = 0.0
        a3[3], b3[3] = 5e-324, 0.0
        assert run(a3, b3, fpe, ee).flags == 0
        # (5) finite operands, overflowing product -> bits 4 + 6 and the correctly rounded exact value: here 2^1100 + ...,
        # beyond the double range, i.e. +-Inf (MPFR agrees); see test_exdot_overflowing_products_are_summed_exactly
        a4, b4 = big.copy(), big[::-1].copy()
        a4[j], b4[j] = np.ldexp(1.0, 600), np.ldexp(1.0, 500)
        rec = run(a4, b4, fpe, ee)
        assert (rec.flags & ~3) == (FLAG_POVER | FLAG_PHIGH_EXACT) and rec.exact == np.inf, (fpe, ee, rec.flags, rec.exact)
        a4[j] = -a4[j]
        rec = run(a4, b4, fpe, ee)
        assert (rec.flags & ~3) == (FLAG_POVER | FLAG_PHIGH_EXACT) and rec.exact == -np.inf
        # a true Inf operand is NOT a product overflow
        a4[j], b4[j] = np.inf, 2.0
        rec = ex.read_record(ex.exdot_dev(torch.from_numpy(a4).cuda(), torch.from_numpy(b4).cuda(), fpe, ee))
        assert rec.flags == 1 and rec.exact == np.inf
    # strided path and the exact boundary: 2^-968 itself is inside (error term >= 2^-1074 representable)
    a = np.zeros(64)
    b = np.zeros(64)
    a[0], b[0] = np.ldexp(1.0 + 2.0 ** -52, -484), np.ldexp(1.0 + 2.0 ** -52, -484)   # product 2^-968 (1 + 2^-51 + 2^-104)
    a[2], b[2] = 3.0, 5.0
    assert run(a, b, 8, True, inca=2).flags == 0
    a[0] = np.ldexp(1.0 + 2.0 ** -52, -485)                                            # one binade lower
    assert run(a, b, 8, True, inca=2).flags == FLAG_PUNDER | FLAG_PLOW_EXACT


def test_exdot_underflowing_products_are_summed_exactly(ex, oracle):
    """The low accumulator at work: MANY products below 2^-968 whose sub-2^-1074 parts add up to whole units and to
    exact ties, totals that are tiny (subnormal results, where the fraction decides the last bit), negative totals
    (|H| - 1 + (1 - f)), cancellation that leaves only the low parts, and subnormal operands.  Always the MPFR-4196
    value, for every variant and on the vector, strided and tail paths."""
    import torch
    rng = np.random.default_rng(12)
    assert oracle.mpfr() is not None

    def check(a, b, what, inca=1):
        nn = (a.size + inca - 1) // inca
        want = oracle.mpfr_exdot(a, b, inca=inca, incb=inca, n=nn)
        for fpe, ee in FPE_VARIANTS_DOT:
            rec = ex.read_record(ex.exdot_dev(torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda(), fpe, ee, incx=inca,
                                              incy=inca, n=nn))
            assert same_double(rec.exact, want), (what, fpe, ee, rec.flags, rec.exact.hex(), float(want).hex())
            assert rec.flags & FLAG_PLOW_EXACT, (what, rec.flags)

    def ld(m, e):
        return np.ldexp(np.asarray(m, dtype=np.float64), np.asarray(e))

    # (a) pure tiny sums: results are subnormal or just above; 2^-1075 terms pair up into whole units and ties
    for n in (1, 2, 3, 4, 5, 1000, 4097, 33333):
        a = ld(np.ones(n), np.full(n, -537))
        b = ld(np.ones(n), np.full(n, -538))                   # every product = 2^-1075 = half a unit
        check(a, b, f"halves n={n}")
        check(-a, b, f"negative halves n={n}")
        b2 = b.copy()
        b2[0] *= 1.5                                           # one product 0.75 unit: breaks the ties
        check(a, b2, f"halves + 0.75 n={n}")
    # (b) random tiny products (exponents -2148 .. -960), random signs, mixed with subnormal operands
    for n, inca in ((5000, 1), (5001, 1), (777, 3), (20011, 2)):
        m = n * inca
        ea = rng.integers(-1074, -400, m)
        eb = np.clip(rng.integers(-1500, -960, m) - ea, -1074, 100)
        a = ld(rng.uniform(1, 2, m) * rng.choice([-1.0, 1.0], m), ea)
        b = ld(rng.uniform(1, 2, m), eb)
        check(a, b, f"random tiny n={n} inc={inca}", inca)
    # (c) an ordinary-sized sum sitting exactly on a rounding tie, decided by the low parts
    base = np.array([1.0, 2.0 ** -53]), np.array([1.0, 1.0])           # 1 + 2^-53: a tie between 1 and 1 + 2^-52
    for sg, e1, e2 in ((1.0, -540, -540), (-1.0, -540, -540), (1.0, -550, -550), (-1.0, -1000, -1000), (1.0, -1074, -1074)):
        a = np.concatenate([base[0], [sg * 2.0 ** e1]])           # the third product is +-2^(e1 + e2): far below 2^-1074
        b = np.concatenate([base[1], [2.0 ** e2]])
        check(a, b, f"tie {sg:+.0f} 2^{e1 + e2}")
        check(-a, b, f"-(tie {sg:+.0f} 2^{e1 + e2})")
    # (d) cancellation: big terms cancel exactly, only tiny products remain
    n = 3000
    x = ld(rng.uniform(1, 2, n), rng.integers(-10, 10, n))
    a = np.concatenate([x, -x, ld(rng.uniform(1, 2, 500), np.full(500, -540))])
    b = np.concatenate([x[::-1], x[::-1], ld(rng.uniform(1, 2, 500) * rng.choice([-1.0, 1.0], 500), rng.integers(-560, -520, 500))])
    check(a, b, "cancellation")


def test_exdot_overflowing_products_are_summed_exactly(ex, oracle):
    """The high accumulator at work: products of FINITE operands at or beyond 2^1024 (+-Inf as doubles) are formed again
    at a scaled exponent and summed exactly; the finalize adds them back 1216 bits up.  Where they cancel, the result is
    the finite MPFR-4196 value (the reference's kernels, and IEEE arithmetic, give Inf - Inf = NaN); where they do not,
    the exact sum is beyond the double range and rounds to +-Inf, as MPFR's does.  Every variant, vector / strided /
    tail paths, together with underflowing products, and next to true Inf / NaN operands (which keep IEEE semantics)."""
    import torch
    rng = np.random.default_rng(13)
    assert oracle.mpfr() is not None
    EX = FLAG_POVER | FLAG_PHIGH_EXACT

    def check(a, b, what, inca=1, flags=EX, variants=FPE_VARIANTS_DOT):
        nn = (a.size + inca - 1) // inca
        want = oracle.mpfr_exdot(a, b, inca=inca, incb=inca, n=nn)
        for fpe, ee in variants:
            rec = ex.read_record(ex.exdot_dev(torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda(), fpe, ee, incx=inca,
                                              incy=inca, n=nn))
            assert same_double(rec.exact, want), (what, fpe, ee, rec.flags, rec.exact.hex(), float(want).hex())
            if np.isfinite(want):
                assert rec.flags == flags, (what, fpe, ee, rec.flags)
            else:     # a sum beyond 2^1101 cannot be held by the record's digits: reported like an infinity in the input
                assert (rec.flags & ~3) == flags, (what, fpe, ee, rec.flags)
        return want

    def ld(m, e):
        return np.ldexp(np.asarray(m, dtype=np.float64), np.asarray(e))

    # (a) exact cancellation of two overflowing products leaves the ordinary part
    assert check(np.array([2.0 ** 600, -2.0 ** 600, 3.0]), np.array([2.0 ** 500, 2.0 ** 500, 5.0]), "cancel") == 15.0
    # (b) near-cancellation: the difference of two products near 2^1030 is a finite double (2^978 scale)
    x, y = 1.0 + 2.0 ** -30, 1.0 + 2.0 ** -22
    w = check(np.array([x * 2.0 ** 520, -x * 2.0 ** 520, 1.0]), np.array([y * 2.0 ** 510, y * (1 + 2.0 ** -52) * 2.0 ** 510, 1.0]), "near")
    assert np.isfinite(w) and w < -2.0 ** 970
    # (c) sums that stay beyond the range: inside the accumulator's headroom (2^1024 .. 2^1101) and beyond it, both signs
    for e1, e2 in ((512, 512), (530, 540), (600, 500), (1023, 1023), (1000, 100)):
        for sg in (1.0, -1.0):
            w = check(np.array([sg * 2.0 ** e1, 7.0]), np.array([2.0 ** e2, 9.0]), f"over {sg} {e1}+{e2}")
            assert w == sg * np.inf
    # the boundary: 2^1024 exactly overflows as a double (diverted), (2 - 2^-52) 2^1023 does not (ordinary path, exact)
    check(np.array([2.0 ** 512, -2.0 ** 512]), np.array([2.0 ** 512, 2.0 ** 512 * (1 - 2.0 ** -53)]), "2^1024 - (2^1024 - 2^971)")
    a = np.array([np.ldexp(2.0 - 2.0 ** -52, 511), -np.ldexp(2.0 - 2.0 ** -52, 511)])
    b = np.array([np.ldexp(1.0, 512), np.ldexp(1.0, 512)])
    assert check(a, b, "largest finite products", flags=0) == 0.0
    # (d) many overflowing products with random signs that cancel in pairs, among ordinary and underflowing ones
    for n, inca in ((4000, 1), (4001, 1), (999, 3), (20011, 2)):
        m = n * inca
        ea = rng.integers(400, 1023, m)
        eb = np.clip(rng.integers(1024, 2040, m) - ea, -1000, 1023)
        a = ld(rng.uniform(1, 2, m) * rng.choice([-1.0, 1.0], m), ea)
        b = ld(rng.uniform(1, 2, m), eb)
        half = (m // (2 * inca)) * inca
        a[half:2 * half] = -a[:half]                # every product of the first half has its negative in the second
        b[half:2 * half] = b[:half]
        k = min(50, m - 2 * half)
        if k > 0:
            a[2 * half:2 * half + k] = rng.uniform(-4, 4, k)
            b[2 * half:2 * half + k] = rng.uniform(-4, 4, k)
        w = check(a, b, f"pairs n={n} inc={inca}", inca)
        assert np.isfinite(w)
        # ... the same with tiny products in the mix: all four product flags
        a2, b2 = a.copy(), b.copy()
        a2[::11 * inca] = ld(rng.uniform(1, 2, a2[::11 * inca].size), -600)
        b2[::11 * inca] = ld(rng.uniform(1, 2, a2[::11 * inca].size), -520)
        a2[half:2 * half] = -a2[:half]
        b2[half:2 * half] = b2[:half]
        check(a2, b2, f"pairs + tiny n={n} inc={inca}", inca, flags=EX | FLAG_PUNDER | FLAG_PLOW_EXACT)
    # (e) partial cancellation: the surviving sum is a product-sized number that is still a double
    n = 3000
    xs = ld(rng.uniform(1, 2, n), rng.integers(500, 520, n))
    ys = ld(rng.uniform(1, 2, n), rng.integers(520, 540, n))
    a = np.concatenate([xs, -xs, [2.0 ** 500, 1.5]])
    b = np.concatenate([ys, ys, [2.0 ** 500, 2.5]])           # + 2^1000 + 3.75
    w = check(a, b, "partial")
    assert w == 2.0 ** 1000
    # (f) true Inf / NaN operands keep IEEE semantics next to overflowing products
    a = np.array([2.0 ** 600, -2.0 ** 600, np.inf, 1.0])
    b = np.array([2.0 ** 500, 2.0 ** 500, 2.0, 1.0])
    for fpe, ee in FPE_VARIANTS_DOT:
        rec = ex.read_record(ex.exdot_dev(torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda(), fpe, ee))
        assert rec.exact == np.inf and rec.flags == (EX | 1)
    a[2] = np.nan
    rec = ex.read_record(ex.exdot_dev(torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda(), 8, True))
    assert np.isnan(rec.exact) and rec.flags == (EX | 4)
    # (g) a long vector: the accumulate / finish split, several launches into one reduction
    n = (1 << 20) + 5
    a = ld(rng.uniform(1, 2, n) * rng.choice([-1.0, 1.0], n), rng.integers(505, 525, n))
    b = ld(rng.uniform(1, 2, n), rng.integers(505, 525, n))
    a[n // 2:2 * (n // 2)] = -a[:n // 2]
    b[n // 2:2 * (n // 2)] = b[:n // 2]
    a[-1], b[-1] = 1.25, 3.0                                  # (n is odd: the unpaired last element)
    want = oracle.mpfr_exdot(a, b)
    ta, tb = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
    third = n // 3
    for lo, hi in ((0, third), (third, 2 * third), (2 * third, n)):
        ex.exdot_accumulate_dev(ta[lo:hi], tb[lo:hi], 8, True)
    rec = ex.read_record(ex.finish_dev())
    assert same_double(rec.exact, want) and rec.flags == EX and np.isfinite(want)
    three, five = torch.tensor([3.0], dtype=torch.float64, device="cuda"), torch.tensor([5.0], dtype=torch.float64, device="cuda")
    rec = ex.read_record(ex.exdot_dev(three, five, 8, True))                      # (the accumulators were left clean)
    assert rec.exact == 15.0 and rec.flags == 0


def test_exdot_flags_survive_the_digit_set(ex):
    """The product flags travel with the 576-byte digit set (its pad word).  exblas_finalize_dev on user-held digit sets
    has no low sets to fold: a result with bit 3 but not bit 5 is the correctly rounded sum of the sets' values, each
    truncated at 2^-1074 (the library's own multi-rank calls all-reduce the low digit sets too and stay exact:
    tests/test_gpu_multirank.py)."""
    import torch
    a = torch.tensor([2.0 ** -500, 1.0], dtype=torch.float64, device="cuda")
    b = torch.tensor([2.0 ** -490, 3.0], dtype=torch.float64, device="cuda")
    r1 = ex.exdot_dev(a, b, 8, True)
    r2 = ex.exdot_dev(b[1:], b[1:], 8, True)
    assert ex.read_record(r1).flags == FLAG_PUNDER | FLAG_PLOW_EXACT and ex.read_record(r2).flags == 0
    sets = torch.stack([r1[ex.OUT_DIGITS:ex.OUT_DIGITS + ex.SET_WORDS], r2[ex.OUT_DIGITS:ex.OUT_DIGITS + ex.SET_WORDS]])
    tot = ex.read_record(ex.finalize_dev(sets.contiguous()))
    assert tot.flags == FLAG_PUNDER
    assert tot.exact == 2.0 ** -990 + 3.0 + 9.0
