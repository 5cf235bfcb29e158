"""ctypes front-end of the CPU checkers in oracle/.

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg -- never by exblas_amd (the product).  See oracle/exblas_oracle.h.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
NLIMBS = 41
ROUND_EXACT, ROUND_REFERENCE = 0, 1
KINDS = {"naive": 0, "fpuniform": 1, "lognormal": 2, "ill_cond": 3, "cancel": 4, "fpuniform_signed": 5}

_dp = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
_lp = np.ctypeslib.ndpointer(dtype=np.int64, flags="C_CONTIGUOUS")


def build(force=False):
    """(Re)build the checkers with oracle/Makefile (gcc; the _ref target needs /root/reference)."""
    if force or not os.path.exists(os.path.join(HERE, "liboracle.so")):
        subprocess.run(["make", "-C", HERE, "all"], check=True, stdout=subprocess.DEVNULL)


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(os.path.join(HERE, "liboracle.so"))
        L.orc_exsum.restype = C.c_double
        L.orc_exsum.argtypes = [C.c_int, _dp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.orc_exsum_omp.restype = C.c_double
        L.orc_exsum_omp.argtypes = [C.c_int, _dp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.orc_exdot.restype = C.c_double
        L.orc_exdot.argtypes = [C.c_int, _dp, C.c_int, C.c_int, _dp, C.c_int, C.c_int, C.c_int, C.c_int,
                                C.c_int, C.c_void_p]
        L.orc_exdot_omp.restype = C.c_double
        L.orc_exdot_omp.argtypes = [C.c_int, _dp, _dp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.orc_exgemv.restype = C.c_int
        L.orc_exgemv.argtypes = [C.c_char, C.c_int, C.c_int, C.c_double, _dp, C.c_int, C.c_int, _dp, C.c_int,
                                 C.c_int, C.c_double, _dp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
        L.orc_exgemm.restype = C.c_int
        L.orc_extrsv.restype = C.c_int
        L.orc_extrsv.argtypes = [C.c_char, C.c_char, C.c_char, C.c_int, _dp, C.c_int, C.c_int, _dp, C.c_int, C.c_int,
                                 C.c_int, C.c_int, C.c_int]
        L.orc_exgemm.argtypes = [C.c_char, C.c_char, C.c_int, C.c_int, C.c_int, C.c_double, _dp, C.c_int, _dp,
                                 C.c_int, C.c_double, _dp, C.c_int, C.c_int, C.c_int, C.c_int]
        L.orc_round_limbs.restype = C.c_double
        L.orc_round_limbs.argtypes = [_lp, C.c_int]
        L.orc_normalize_limbs.argtypes = [_lp]
        L.orc_gen_ctr.argtypes = [C.c_int, C.c_uint64, C.c_int64, C.c_int64, C.c_int64, C.c_double, C.c_double, _dp]
        L.orc_srand.argtypes = [C.c_uint]
        L.orc_init_naive.argtypes = [C.c_int, _dp]
        L.orc_init_fpuniform_rand.argtypes = [C.c_int, _dp, C.c_int, C.c_int]
        L.orc_init_ill_cond_rand.argtypes = [C.c_int, _dp, C.c_double]
        _lib = L
    return _lib


def _limbs_arg(want):
    if not want:
        return None, None
    buf = np.zeros(NLIMBS, dtype=np.int64)
    return buf, buf.ctypes.data_as(C.c_void_p)


def exsum(a, fpe=0, early_exit=False, inca=1, offset=0, n=None, mode=ROUND_EXACT, limbs=False):
    a = np.ascontiguousarray(a, dtype=np.float64)
    if n is None:
        n = 0 if a.size == 0 else (a.size - offset + inca - 1) // inca
    buf, ptr = _limbs_arg(limbs)
    if a.size == 0:
        a = np.zeros(1)
    r = lib().orc_exsum(n, a, inca, offset, fpe, int(early_exit), mode, ptr)
    return (r, buf) if limbs else r


def exsum_omp(a, fpe=8, early_exit=True, nthreads=1, mode=ROUND_EXACT, limbs=False):
    a = np.ascontiguousarray(a, dtype=np.float64)
    buf, ptr = _limbs_arg(limbs)
    r = lib().orc_exsum_omp(a.size, a, fpe, int(early_exit), nthreads, mode, ptr)
    return (r, buf) if limbs else r


def exdot(a, b, fpe=0, early_exit=False, inca=1, offa=0, incb=1, offb=0, n=None, mode=ROUND_EXACT, limbs=False):
    a = np.ascontiguousarray(a, dtype=np.float64)
    b = np.ascontiguousarray(b, dtype=np.float64)
    if n is None:
        n = 0 if a.size == 0 else (a.size - offa + inca - 1) // inca
    buf, ptr = _limbs_arg(limbs)
    if a.size == 0:
        a = np.zeros(1)
        b = np.zeros(1)
    r = lib().orc_exdot(n, a, inca, offa, b, incb, offb, fpe, int(early_exit), mode, ptr)
    return (r, buf) if limbs else r


def exdot_omp(a, b, fpe=8, early_exit=True, nthreads=1, mode=ROUND_EXACT, limbs=False):
    a = np.ascontiguousarray(a, dtype=np.float64)
    b = np.ascontiguousarray(b, dtype=np.float64)
    buf, ptr = _limbs_arg(limbs)
    r = lib().orc_exdot_omp(a.size, a, b, fpe, int(early_exit), nthreads, mode, ptr)
    return (r, buf) if limbs else r


def exgemv(trans, m, n, alpha, a, lda, x, beta, y, fpe=0, early_exit=False, incx=1, incy=1, offa=0, offx=0,
           offy=0, mode=ROUND_EXACT):
    y = np.array(y, dtype=np.float64, copy=True)
    lib().orc_exgemv(trans.encode(), m, n, alpha, np.ascontiguousarray(a, dtype=np.float64), lda, offa,
                     np.ascontiguousarray(x, dtype=np.float64), incx, offx, beta, y, incy, offy, fpe,
                     int(early_exit), mode)
    return y


def exgemm(transa, transb, m, n, k, alpha, a, lda, b, ldb, beta, c, ldc, fpe=0, early_exit=False,
           mode=ROUND_EXACT):
    c = np.array(c, dtype=np.float64, copy=True)
    lib().orc_exgemm(transa.encode(), transb.encode(), m, n, k, alpha, np.ascontiguousarray(a, dtype=np.float64),
                     lda, np.ascontiguousarray(b, dtype=np.float64), ldb, beta, c, ldc, fpe, int(early_exit), mode)
    return c


def extrsv(uplo, trans, diag, n, a, lda, x, fpe=0, early_exit=False, incx=1, offa=0, offx=0, mode=ROUND_EXACT):
    """returns (rc, solution); x is copied"""
    x = np.array(x, dtype=np.float64, copy=True)
    rc = lib().orc_extrsv(uplo.encode(), trans.encode(), diag.encode(), n, np.ascontiguousarray(a, dtype=np.float64),
                          lda, offa, x, incx, offx, fpe, int(early_exit), mode)
    return rc, x


def round_limbs(limbs, mode=ROUND_EXACT):
    return lib().orc_round_limbs(np.ascontiguousarray(limbs, dtype=np.int64), mode)


def normalize_limbs(limbs):
    out = np.array(limbs, dtype=np.int64, copy=True)
    lib().orc_normalize_limbs(out)
    return out


def gen(kind, n, seed=1, p0=0.0, p1=0.0, first=0, count=None, n_total=None):
    """Counter-based generator (bit-identical to exblas_amd's HIP generator)."""
    k = KINDS[kind] if isinstance(kind, str) else int(kind)
    if count is None:
        count = n
    if n_total is None:
        n_total = n
    out = np.empty(count, dtype=np.float64)
    if count:
        lib().orc_gen_ctr(k, seed, first, count, n_total, p0, p1, out)
    return out


def gen_rand(kind, n, seed=1, p0=0.0, p1=0.0):
    """Restatement of the reference generators on glibc rand() (src/common/common.cpp)."""
    out = np.empty(max(n, 1), dtype=np.float64)
    L = lib()
    L.orc_srand(seed)
    if kind == "naive":
        L.orc_init_naive(n, out)
    elif kind == "fpuniform":
        L.orc_init_fpuniform_rand(n, out, int(p0), int(p1))
    elif kind == "ill_cond":
        L.orc_init_ill_cond_rand(n, out, float(p0))
    else:
        raise ValueError(kind)
    return out[:n]


# ---------------------------------------------------------------------------------------------
# the compiled reference core (oracle/_ref) and MPFR -- optional
# ---------------------------------------------------------------------------------------------
_ref = None
_mpfr = None


def ref():
    """oracle/_ref/libexblas_ref.so (reference sources + our driver) or None."""
    global _ref
    if _ref is None:
        p = os.path.join(HERE, "_ref", "libexblas_ref.so")
        if not os.path.exists(p):
            return None
        try:
            with open("/proc/cpuinfo") as f:
                flags = f.read()
            if " avx2" not in flags or " fma" not in flags:
                return None
        except OSError:
            pass
        L = C.CDLL(p)
        L.ref_exsum.restype = C.c_double
        L.ref_exsum.argtypes = [C.c_long, _dp, C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.ref_round_limbs.restype = C.c_double
        L.ref_round_limbs.argtypes = [_lp]
        L.ref_srand.argtypes = [C.c_uint]
        L.ref_init_naive.argtypes = [C.c_int, _dp]
        L.ref_init_fpuniform.argtypes = [C.c_int, _dp, C.c_int, C.c_int]
        L.ref_init_ill_cond.argtypes = [C.c_int, _dp, C.c_double]
        L.ref_init_lognormal.argtypes = [C.c_int, _dp, C.c_double, C.c_double]
        _ref = L
    return _ref


def ref_exsum(a, fpe=0, early_exit=False, nthreads=1, limbs=False):
    a = np.ascontiguousarray(a, dtype=np.float64)
    buf, ptr = _limbs_arg(limbs)
    arr = a if a.size else np.zeros(1)
    r = ref().ref_exsum(a.size, arr, fpe, int(early_exit), nthreads, ptr)
    return (r, buf) if limbs else r


def ref_gen(kind, n, seed=1, p0=0.0, p1=0.0):
    out = np.empty(max(n, 1), dtype=np.float64)
    L = ref()
    L.ref_srand(seed)
    if kind == "naive":
        L.ref_init_naive(n, out)
    elif kind == "fpuniform":
        L.ref_init_fpuniform(n, out, int(p0), int(p1))
    elif kind == "ill_cond":
        L.ref_init_ill_cond(n, out, float(p0))
    elif kind == "lognormal":
        L.ref_init_lognormal(n, out, float(p0), float(p1))
    else:
        raise ValueError(kind)
    return out[:n]


def mpfr():
    global _mpfr
    if _mpfr is None:
        p = os.path.join(HERE, "libmpfr_oracle.so")
        if not os.path.exists(p):
            return None
        try:
            L = C.CDLL(p)
        except OSError:
            return None
        L.mpfr_exsum.restype = C.c_double
        L.mpfr_exsum.argtypes = [C.c_long, _dp, C.c_long, C.c_long]
        L.mpfr_exdot.restype = C.c_double
        L.mpfr_exdot.argtypes = [C.c_long, _dp, C.c_long, C.c_long, _dp, C.c_long, C.c_long]
        L.mpfr_exgemv.argtypes = [C.c_char, C.c_int, C.c_int, C.c_double, _dp, C.c_int, _dp, C.c_int, C.c_double,
                                  _dp, C.c_int, _dp]
        L.mpfr_exgemm_dots.argtypes = [C.c_int, C.c_int, C.c_int, _dp, C.c_int, _dp, C.c_int, _dp, C.c_int]
        L.mpfr_extrsv.argtypes = [C.c_char, C.c_char, C.c_char, C.c_int, _dp, C.c_int, _dp, C.c_int, C.c_int]
        _mpfr = L
    return _mpfr


def mpfr_exsum(a, inca=1, offset=0, n=None):
    a = np.ascontiguousarray(a, dtype=np.float64)
    if n is None:
        n = 0 if a.size == 0 else (a.size - offset + inca - 1) // inca
    return mpfr().mpfr_exsum(n, a if a.size else np.zeros(1), inca, offset)


def mpfr_exdot(a, b, inca=1, offa=0, incb=1, offb=0, n=None):
    a = np.ascontiguousarray(a, dtype=np.float64)
    b = np.ascontiguousarray(b, dtype=np.float64)
    if n is None:
        n = 0 if a.size == 0 else (a.size - offa + inca - 1) // inca
    return mpfr().mpfr_exdot(n, a if a.size else np.zeros(1), inca, offa, b if b.size else np.zeros(1), incb, offb)


def mpfr_exgemv(trans, m, n, alpha, a, lda, x, beta, y):
    rows = n if trans == "T" else m
    out = np.empty(rows, dtype=np.float64)
    mpfr().mpfr_exgemv(trans.encode(), m, n, alpha, np.ascontiguousarray(a, dtype=np.float64), lda,
                       np.ascontiguousarray(x, dtype=np.float64), 1, beta,
                       np.ascontiguousarray(y, dtype=np.float64), 1, out)
    return out


def mpfr_exgemm_dots(m, n, k, a, lda, b, ldb):
    out = np.empty((m, n), dtype=np.float64)
    mpfr().mpfr_exgemm_dots(m, n, k, np.ascontiguousarray(a, dtype=np.float64), lda,
                            np.ascontiguousarray(b, dtype=np.float64), ldb, out, n)
    return out


def mpfr_extrsv(uplo, trans, diag, n, a, lda, b, two_step=True):
    x = np.array(b, dtype=np.float64, copy=True)
    mpfr().mpfr_extrsv(uplo.encode(), trans.encode(), diag.encode(), n, np.ascontiguousarray(a, dtype=np.float64), lda,
                       x, 1, int(two_step))
    return x
