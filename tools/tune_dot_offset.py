#!/usr/bin/env python3
"""Does the relative placement of the two ExDOT streams matter (DRAM channel/bank aliasing)?"""
import sys, os, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import exblas_amd as ex
n = 1 << 28
lib = ex.load_library()
x = ex.gen_dev("ill_cond", n, 1, 1e32)
big = torch.empty(n + (1 << 22), dtype=torch.float64, device="cuda")
offs = [0, 2, 32, 64, 512, 514, 4096 + 32, 65536 + 32, (1 << 20) + 32, (1 << 21) + 512 + 32]
print("x ptr %x big ptr %x delta %d" % (x.data_ptr(), big.data_ptr(), big.data_ptr() - x.data_ptr()))
times = {o: [] for o in offs}
rec = ex.new_record_buffer()
for r in range(6):
    for o in offs:
        y = big[o:o + n]
        if r == 0:
            ex.gen_dev("ill_cond", n, 2, 1e32, out=y)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            ex.exdot_accumulate_dev(x, y, 8, True)
        e1.record()
        ex.finish_dev(out=rec)
        torch.cuda.synchronize()
        if r:
            times[o].append(e0.elapsed_time(e1) / 3)
for o in offs:
    med = statistics.median(times[o])
    print(f"offset {o:8d} elements: {med:.4f} ms  {n*16/med/1e6:.0f} GB/s")
