"""The reference's OWN GPU test programs, run unmodified against libexblas.so on the GPU.

oracle/_ref/reftest_<op>[_mpfr] are tests/test.ex{sum,dot,gemv,gemm,trsv}.gpu.cpp of nikolovjovan/exblas compiled where
they lie (oracle/Makefile: reftests) against this repository's include/ and linked against exblas_amd/lib/libexblas.so --
nothing of the reference is in the repository; the binaries are built in the container that has /root/reference and
travel to the GPU box like our own .so files.  The command lines are the ones the reference registers with CTest
(src/gpu/blas/blas1/CMakeLists.txt:9-30, blas2/CMakeLists.txt:12-80, blas3/CMakeLists.txt:11-18); the pass criterion is
the reference's own: the program prints "TestPassed; ALL OK" (variants agree; with -DEXBLAS_VS_MPFR: every variant equals
the MPFR oracle the test file defines, test.exsum.gpu.cpp:23-38, test.exdot.gpu.cpp:24-46, test.exgemv.gpu.cpp:35-103,
test.exgemm.gpu.cpp:53-125, test.extrsv.gpu.cpp:27-66)."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFDIR = os.path.join(ROOT, "oracle", "_ref")

# (program, args) exactly as registered with CTest by the reference
CTEST = [
    ("exsum", ["24"]), ("exsum", ["24", "2", "0", "n"]), ("exsum", ["24", "50", "0", "n"]), ("exsum", ["24", "1e+50", "0", "i"]),
    ("exdot", ["24"]), ("exdot", ["24", "2", "0", "n"]), ("exdot", ["24", "50", "0", "n"]), ("exdot", ["24", "1e+50", "0", "i"]),
    ("exgemm", ["256", "256", "256"]), ("exgemm", ["256", "256", "256", "2", "0", "n"]),
    ("exgemm", ["256", "256", "256", "50", "0", "n"]), ("exgemm", ["256", "256", "256", "1e+50", "0", "i"]),
    ("extrsv", ["U", "N", "N", "256"]), ("extrsv", ["U", "N", "N", "256", "50", "0", "n"]),
    ("extrsv", ["U", "N", "N", "256", "10", "0", "y"]), ("extrsv", ["U", "N", "N", "256", "1e+50", "0", "i"]),
]
for _t in ("N", "T"):
    for _m, _n in (("512", "512"), ("512", "1024"), ("1024", "512")):
        CTEST += [("exgemv", [_t, _m, _n]), ("exgemv", [_t, _m, _n, "50", "0", "n"]), ("exgemv", [_t, _m, _n, "10", "0", "y"]),
                  ("exgemv", [_t, _m, _n, "1e+50", "0", "i"])]


def _run(exe, args):
    r = subprocess.run([exe] + args, capture_output=True, text=True, timeout=600, cwd=REFDIR)
    return r.returncode, r.stdout, r.stderr


@pytest.mark.parametrize("mpfr", [False, True], ids=["variants-agree", "vs-mpfr"])
def test_reference_ctest_suite_passes_against_libexblas(mpfr):
    missing = [op for op in ("exsum", "exdot", "exgemv", "exgemm", "extrsv")
               if not os.path.exists(os.path.join(REFDIR, f"reftest_{op}" + ("_mpfr" if mpfr else "")))]
    if missing:
        pytest.skip(f"oracle/_ref/reftest_* not built for {missing} (needs /root/reference at build time)")
    failed = []
    for op, args in CTEST:
        exe = os.path.join(REFDIR, f"reftest_{op}" + ("_mpfr" if mpfr else ""))
        rc, out, err = _run(exe, args)
        if rc != 0 or "TestPassed; ALL OK" not in out:
            failed.append((op, args, rc, out[-600:], err[-300:]))
    assert not failed, failed
