"""Launch geometry for a rank's shard at G = 8 (2^25 elements of ONE 2^28 vector): time of streaming kernel + finalize
against the workgroups per CU, ExSUM and ExDOT.  python tools/tune_shard.py [log2n]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import exblas_amd as ex
lib = ex.load_library()
lib.exblas_hip_init(-1)
lg = int(sys.argv[1]) if len(sys.argv) > 1 else 25
n = 1 << lg
xs = [ex.gen_dev("ill_cond", n, 1 + i, 1e32) for i in range(8)]
ys = [ex.gen_dev("ill_cond", n, 11 + i, 1e32) for i in range(8)]
rec = ex.new_record_buffer()


def t(fn, reps=400):
    for _ in range(50):
        fn(0)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(reps):
        fn(i)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for bpc in (0, 1, 2, 3, 4, 6, 8, 12, 16, 24, 48):
    if bpc:
        lib.exblas_set_tuning(bpc, -1, -1)
    ts = t(lambda i: ex.exsum_dev(xs[i % 8], 8, True, out=rec))
    td = t(lambda i: ex.exdot_dev(xs[i % 8], ys[i % 8], 8, True, out=rec))
    print(f"blocks/CU {bpc or 'default'}: exsum {ts:.1f} us ({n * 8 / ts / 1e6:.2f} TB/s)   exdot {td:.1f} us ({n * 16 / td / 1e6:.2f} TB/s)", flush=True)
