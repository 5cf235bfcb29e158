#!/usr/bin/env python3
"""A/B the kernel variants / launch geometry in ONE process, interleaved rounds (guide rule 24).

usage: python tools/tune.py [exsum|exdot] [log2n]
Prints per configuration the median and min kernel time (HIP events around the streaming kernel only)
and the achieved algorithmic bandwidth.  Every configuration must produce identical limbs.
"""
import sys, os, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import exblas_amd as ex

op = sys.argv[1] if len(sys.argv) > 1 else "exsum"
log2n = int(sys.argv[2]) if len(sys.argv) > 2 else 28
variants = [int(v) for v in os.environ.get("TUNE_VARIANTS", "0,1,2,3,4,5").split(",")]
bpcs = [int(v) for v in os.environ.get("TUNE_BPC", "4,8,16").split(",")]
rounds = int(os.environ.get("TUNE_ROUNDS", "7"))
fpe = int(os.environ.get("TUNE_FPE", "8"))
ee = os.environ.get("TUNE_EE", "1") == "1"
kind = os.environ.get("TUNE_KIND", "ill_cond")
p0 = float(os.environ.get("TUNE_P0", "1e32"))
p1 = float(os.environ.get("TUNE_P1", "0"))
n = 1 << log2n
lib = ex.load_library()
x = ex.gen_dev(kind, n, 1, p0, p1)
y = ex.gen_dev(kind, n, 2, p0, p1) if op == "exdot" else None
rec = ex.new_record_buffer()
bpe = 8 if op == "exsum" else 16
cfgs = [(v, b) for v in variants for b in bpcs]
times = {c: [] for c in cfgs}
canon0 = None
for r in range(rounds + 1):
    for c in cfgs:
        lib.exblas_set_tuning(c[1], -1, c[0])
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 3
        e0.record()
        for _ in range(reps):
            if op == "exsum":
                ex.exsum_accumulate_dev(x, fpe, ee)
            else:
                ex.exdot_accumulate_dev(x, y, fpe, ee)
        e1.record()
        ex.finish_dev(out=rec)
        torch.cuda.synchronize()
        if r == 0:
            got = ex.read_record(rec)
            # three accumulations of the same vector: limbs are 3x the single sum -> compare across configs
            if canon0 is None:
                canon0 = got.canon
            assert (got.canon == canon0).all(), c
        else:
            times[c].append(e0.elapsed_time(e1) / reps)
print(f"{op} n=2^{log2n}: variant,blocks_per_cu -> median ms, min ms, GB/s(median), GB/s(best)")
for c in cfgs:
    med, mn = statistics.median(times[c]), min(times[c])
    print(f"  v{c[0]} bpc{c[1]:2d}: {med:.4f} {mn:.4f}  {n*bpe/med/1e6:8.1f} {n*bpe/mn/1e6:8.1f}")
