"""bench.py's launch contract (BASELINE config 3): `--gpus N` starts exactly N ranks or fails loudly, and the strong-
scaling mode -- ONE vector partitioned n/N as the reference scatters it (src/cpu/blas/blas1/ExSUM.cpp:33-63) with the
limbs reduced across ranks (:142-152) -- yields the same 8 result bytes and the same limbs for every N."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(args, env_extra=None, timeout=900):
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "EXBLAS_BENCH_BACKEND"):
        env.pop(k, None)
    env.update(env_extra or {})
    return subprocess.run([sys.executable, BENCH] + args, capture_output=True, text=True, timeout=timeout, env=env,
                          cwd=ROOT)


def _have_gpu():
    import torch
    return torch.cuda.device_count() > 0


def test_gpus_n_without_devices_fails_loudly():
    """`python bench.py --gpus 2` on a box with fewer than 2 devices exits non-zero with a message instead of running one
    rank and printing n_gpus = 1 (here: no device at all, or the single GPU of a test box)."""
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("box has two devices")
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "0"])
    assert r.returncode == 2, (r.returncode, r.stderr[-500:])
    assert "refusing to run with fewer ranks" in r.stderr
    assert '"metric"' not in r.stdout


def test_world_size_mismatch_fails_loudly():
    """a launcher that started a different number of ranks than --gpus claims: refused before any GPU work"""
    r = _run(["--gpus", "4", "--steps", "1", "--warmup", "0"], {"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode == 2 and "WORLD_SIZE=2" in r.stderr
    r = _run(["--gpus", "0"])
    assert r.returncode != 0


@pytest.mark.gpu
def test_strong_scaling_bits_identical_across_rank_counts():
    """N = 1, 2, 3 ranks (2 and 3 share the one GPU through the gloo rehearsal transport): identical result_bits and
    limbs_crc for ExSUM and ExDOT, each checked against the CPU core on the FULL vector by rank 0, n_gpus = the ranks
    the communicator saw."""
    common = ["--log2n", "21", "--steps", "4", "--warmup", "1", "--prewarm-ms", "0", "--rotate", "2", "--no-host-api",
              "--skip-blas23"]
    lines = {}
    for n in (1, 2, 3):
        r = _run(["--gpus", str(n)] + common, {"EXBLAS_BENCH_BACKEND": "gloo"} if n > 1 else None)
        assert r.returncode == 0, (n, r.stdout[-1500:], r.stderr[-3000:])
        js = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        assert len(js) == 1, r.stdout[-1500:]
        lines[n] = json.loads(js[0])
    one = lines[1]
    assert one["bit_exact_vs_cpu"] is True and one["exdot"]["bit_exact_vs_cpu"] is True
    assert one["cpu_baseline"]["kind"] in ("reference", "port")
    for n in (2, 3):
        ln = lines[n]
        assert ln["n_gpus"] == n and ln["scaling"] == "strong" and ln["config"]["n_total"] == 1 << 21
        assert "gloo" in ln["transport"]
        assert "cpu_baseline" not in ln                      # a reported baseline at N = 1 only
        assert ln["weak"]["n_total"] == n << 21 and ln["weak"]["scaling"] == "weak"
        for a, b in ((one, ln), (one["exdot"], ln["exdot"])):
            assert b["bit_exact_vs_cpu"] is True, (n, b.get("bit_exact_detail"))
            for key in ("result_bits", "result_bits_reference_rounding", "limbs_crc"):
                assert a[key] == b[key], (n, key, a[key], b[key])


@pytest.mark.gpu
def test_blas23_watchdog_keeps_the_headline():
    """N > 1: the BLAS2/3 items post collectives; if they do not finish in --blas23-timeout seconds rank 0 still prints the
    line (headline + ExDOT, bit checks included) and every rank leaves.  Forced here with a timeout no ExGEMV can meet."""
    r = _run(["--gpus", "2", "--log2n", "21", "--steps", "4", "--warmup", "1", "--prewarm-ms", "0", "--rotate", "2",
              "--no-host-api", "--blas23-timeout", "0.05"], {"EXBLAS_BENCH_BACKEND": "gloo"})
    js = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(js) == 1, (r.returncode, r.stdout[-1500:], r.stderr[-3000:])
    ln = json.loads(js[0])
    assert ln["n_gpus"] == 2 and ln["bit_exact_vs_cpu"] is True and ln["exdot"]["bit_exact_vs_cpu"] is True
    assert ln["blas23_timed_out_after_s"] == 0.05 and "exgemm" not in ln
