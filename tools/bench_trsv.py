"""ExTRSV timing on one GPU: python tools/bench_trsv.py [n ...].  Device-resident A and x, HIP events.
Reports ms per solve, the per-row chain latency (ms / n) and the reference's own figure of merit n^2 / t
(ExTRSV.cpp:253-255 prints it as "GFLOPS")."""
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import exblas_amd as ex


def system(n, uplo):
    g = torch.Generator(device="cuda").manual_seed(3)
    k = int(np.ceil(np.log2(n))) + 1
    a = (torch.rand(n, n, device="cuda", dtype=torch.float64, generator=g) - 0.5) * 2.0 ** (-k)
    d = (torch.rand(n, device="cuda", dtype=torch.float64, generator=g) * 0.5 + 0.5)
    a = torch.tril(a, -1) if uplo == "U" else torch.triu(a, 1)   # a[col][row]: transposed view of the logical matrix
    a = a + torch.diag(d)
    b = torch.rand(n, device="cuda", dtype=torch.float64, generator=g) - 0.5
    return a.contiguous(), b


def main():
    quick = "--sweep" in sys.argv
    sizes = [int(s) for s in sys.argv[1:] if not s.startswith("--")] or [4096, 32768]
    lib = ex.load_library()
    lib.exblas_hip_init(-1)
    if quick:   # latency model: t = t_launch + rows * t_row + blocks * t_block
        for n in (8, 16, 32, 48, 64, 128, 256, 512, 1024, 2048, 4096, 8192):
            a, b = system(n, "L")
            ts = []
            for it in range(6):
                x = b.clone()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                ex.extrsv_dev("L", "N", "N", n, a, n, x, 4, False)
                e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1))
            print(f"sweep n={n}: {min(ts[1:]) * 1e3:.1f} us  slow rows {lib.exblas_extrsv_last_slow_rows()}", flush=True)
        return
    for n in sizes:
        for uplo, trans in (("L", "N"), ("U", "N"), ("L", "T")):
            a, b = system(n, uplo)
            res = {}
            for fpe, ee in ((0, False), (4, False), (8, True), (1, False)):
                xs = []
                ts = []
                for it in range(4):
                    x = b.clone()
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    ex.extrsv_dev(uplo, trans, "N", n, a, n, x, fpe, ee)
                    e1.record()
                    torch.cuda.synchronize()
                    ts.append(e0.elapsed_time(e1))
                    xs.append(x)
                assert all(torch.equal(xs[0].view(torch.int64), y.view(torch.int64)) for y in xs[1:])
                res[(fpe, ee)] = xs[0]
                t = min(ts[1:])
                print(f"n={n} {uplo}{trans} fpe={fpe}{'ee' if ee else ''}: {t:.3f} ms  {t * 1e3 / n:.3f} us/row  "
                      f"{n * n / t * 1e-6:.2f} 'GFLOPS'  finite={bool(torch.isfinite(xs[0]).all())}", flush=True)
            same = all(torch.equal(res[(0, False)].view(torch.int64), res[k].view(torch.int64))
                       for k in ((4, False), (8, True)))
            # residual of the exact solve, fp64 evaluation
            m = a.t() if trans == "N" else a
            m = torch.tril(m) if (uplo == "L") != (trans == "T") else torch.triu(m)
            rres = (m @ res[(0, False)] - b).abs().max().item()
            print(f"   variants bit-identical: {same}; max residual {rres:.2e}", flush=True)


if __name__ == "__main__":
    main()
