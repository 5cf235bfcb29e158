"""CPU suite: the C-ABI library builds for gfx950, loads, and exports every symbol the headers declare."""
import ctypes
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    import exblas_amd
    return exblas_amd.load_library()


def _declared_c_symbols():
    txt = open(os.path.join(ROOT, "include", "exblas_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(exblas_[a-z0-9_]+)\s*\(", txt)))


def test_c_abi_symbols_exported(lib):
    import exblas_amd
    declared = _declared_c_symbols()
    assert len(declared) >= 18
    assert sorted(exblas_amd.C_ABI_SYMBOLS) == declared
    for name in declared:
        assert hasattr(lib, name), name


def test_cxx_api_symbols_exported():
    """exsum/exdot/exgemv/extrsv/exgemm with the reference's C++ signatures (blas1.hpp:48,74; blas2.hpp:57,95; blas3.hpp:56)."""
    import exblas_amd
    out = subprocess.run(["nm", "-D", "--defined-only", "-C", exblas_amd.LIB_PATH], capture_output=True, text=True,
                         check=True).stdout
    for sig in ("exsum(int, double*, int, int, int, bool, bool)",
                "exdot(int, double*, int, int, double*, int, int, int, bool)",
                "exgemv(char, int, int, double, double*, int, int, double*, int, int, double, double*, int, int, int, bool)",
                "extrsv(char, char, char, int, double*, int, int, double*, int, int, int, bool)",
                "exgemm(char, char, int, int, int, double, double*, int, double*, int, double, double*, int, int, bool)",
                "init_ill_cond(int, double*, double)", "init_naive(int, double*)",
                "init_fpuniform(int, double*, int, int)", "init_lognormal(int, double*, double, double)"):
        assert sig in out, sig


def test_library_contains_gfx950_code_object():
    import exblas_amd
    data = open(exblas_amd.LIB_PATH, "rb").read()
    assert b"gfx950" in data


def test_no_gpu_means_loud_failure(lib):
    """Without a HIP device the product refuses to compute (no CPU fallback)."""
    import torch
    import exblas_amd
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    assert lib.exblas_hip_device_count() == 0
    with pytest.raises(RuntimeError):
        exblas_amd.exsum(4, [1.0, 2.0, 3.0, 4.0], 1, 0, 0)


def test_product_does_not_import_oracle():
    """The shipped package may mention the oracle in comments, but never import, include, link or dlopen it."""
    bad = re.compile(r"^\s*(from|import)\s+.*oracle|#\s*include\s*[\"<].*oracle|CDLL\(.*oracle|dlopen\(.*oracle", re.M)
    for dirpath, _, files in os.walk(os.path.join(ROOT, "exblas_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                assert not bad.search(txt), (f, "uses oracle code")


def test_residue_gemm_tables_selftest(lib):
    """The residue ExGEMM path (blas3_crt.hip) reconstructs an integer from its residues modulo up to 39 coprime 8-bit
    moduli with constant tables generated on the host.  exblas_crt_selftest runs a host mirror of the device
    arithmetic (Garner inside groups of three, classical CRT across the 24-bit super-moduli, fp64 fraction sum for the
    multiple of M) over those tables: 200 random integers |S| < M_L / 4 per modulus count L = 1..39, no GPU involved.
    The GPU parity tests (tests/test_gpu_blas23.py::test_exgemm_residue_path_*) check the kernels end to end."""
    assert lib.exblas_crt_selftest(200, 1) == 0
    assert lib.exblas_crt_selftest(50, 12345) == 0
