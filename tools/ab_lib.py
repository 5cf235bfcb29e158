"""A/B of two builds of the library: python tools/ab_lib.py <lib.so|default>.  Times gemv N/T 32768^2, exdot and exsum 2^28 (kernel
chain by events), and checks the results' bits against each other across libs via the printed hex."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import exblas_amd as ex
if len(sys.argv) > 1 and sys.argv[1] != "default":
    os.environ["EXBLAS_AMD_LIB"] = sys.argv[1]   # honoured by exblas_amd.load_library()
ex.load_library().exblas_hip_init(-1)
print("lib:", ex.LIB_PATH, flush=True)


def timeit(fn, reps):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


n = 1 << 28
for kind, p0, p1 in (("ill_cond", 1e32, 0.0), ("fpuniform_signed", 40.0, 20.0)):
    v = ex.gen_dev(kind, n, 1, p0, p1)
    w = ex.gen_dev(kind, n, 2, p0, p1)
    rec = ex.new_record_buffer()
    for rep in range(2):
        ms = timeit(lambda: ex.exdot_dev(v, w, 8, True, out=rec), 50)
        print(f"exdot {kind}: {ms:.4f} ms  {n * 16 / ms / 1e6:.0f} GB/s  result {ex.read_record(rec).exact.hex()}", flush=True)
    ms = timeit(lambda: ex.exdot_dev(v, w, 4, False, out=rec), 50)
    print(f"exdot {kind} fpe4: {ms:.4f} ms  {n * 16 / ms / 1e6:.0f} GB/s  result {ex.read_record(rec).exact.hex()}", flush=True)
    del v, w
m = k = 32768
a = ex.gen_dev("fpuniform", m * k, 1, 10.0, 0.0)
x = ex.gen_dev("fpuniform", k, 2, 10.0, 0.0)
y = ex.gen_dev("fpuniform", m, 3, 10.0, 0.0)
for rep in range(2):
    for trans in ("N", "T"):
        for fpe, ee in ((8, True), (4, False)):
            yy = y.clone()
            ms = timeit(lambda: ex.exgemv_dev(trans, m, k, 1.0, a, m, x, 0.0, yy, fpe, ee), 10)
            print(f"gemv {trans} fpe{fpe}{'ee' if ee else ''}: {ms:.3f} ms  checksum {ex.read_record(ex.exsum_dev(yy)).exact.hex()}", flush=True)
