#!/usr/bin/env python3
"""Throughput of every (fpe, early_exit) variant of ExSUM / ExDOT on several input distributions.
usage: python tools/bench_variants.py [log2n] [wide]   -> markdown table on stdout ("wide": only the rows that spill;
EXBLAS_AMD_LIB=<path> selects an A/B build of the library, tools/ab_build.sh)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import exblas_amd as ex
lg = int(sys.argv[1]) if len(sys.argv) > 1 else 27
n = 1 << lg
data = [("naive", 0.0, 0.0), ("ill_cond", 1e32, 0.0), ("lognormal", 0.0, 2.0), ("lognormal", 0.0, 50.0), ("fpuniform_signed", 1800.0, 900.0)]
if len(sys.argv) > 2 and sys.argv[2] == "wide":
    data = [d for d in data if d[2] == 50.0 or d[1] in (1e32, 1800.0)]
sum_var = [(0, False), (2, False), (3, False), (4, False), (8, False), (4, True), (6, True), (8, True)]
dot_var = [(0, False), (3, False), (4, False), (8, False), (4, True), (6, True), (8, True)]


def t(fn, reps=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


rec = ex.new_record_buffer()
print(f"n = 2^{lg}; GB/s of algorithmic bytes (8 B/elem ExSUM, 16 B/elem ExDOT), kernel + finalize\n")
for op, variants, bpe in (("exsum", sum_var, 8), ("exdot", dot_var, 16)):
    print(f"| {op} data \\ variant | " + " | ".join(f"fpe{f}{'ee' if e else ''}" for f, e in variants) + " |")
    print("|---|" + "---|" * len(variants))
    for kind, p0, p1 in data:
        if op == "exdot" and kind == "fpuniform_signed":
            continue  # products over/underflow: outside ExDOT's exact domain
        x = ex.gen_dev(kind, n, 1, p0, p1)
        y = ex.gen_dev(kind, n, 2, p0, p1) if op == "exdot" else None
        row, ref = [], None
        for f, e in variants:
            if op == "exsum":
                ms = t(lambda: ex.exsum_dev(x, f, e, out=rec))
            else:
                ms = t(lambda: ex.exdot_dev(x, y, f, e, out=rec))
            r = ex.read_record(rec)
            if ref is None:
                ref = r
            assert (r.canon == ref.canon).all() and r.exact == ref.exact, (op, kind, f, e)
            row.append(f"{n * bpe / ms / 1e6:.0f}")
        print(f"| {kind}({p0:g},{p1:g}) | " + " | ".join(row) + " |")
        del x, y
    print()
