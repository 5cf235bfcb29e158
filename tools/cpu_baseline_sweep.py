import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import pyoracle as O
print("affinity", len(os.sched_getaffinity(0)), "cpu_count", os.cpu_count())
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
    try: print(f, open(f).read().strip())
    except Exception as e: print(f, "n/a")
os.system("lscpu | grep -E 'Model name|^CPU\\(s\\)|Thread|Core|Socket' | head -6")
n = 1 << 28
a = O.gen("ill_cond", n, 1, 1e32)
for nt in (8, 16, 32, 64, 128, 256):
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter(); O.ref_exsum(a, 8, True, nthreads=nt); best = min(best, time.perf_counter() - t0)
    print(f"ref FPE8-EE threads={nt:3d}: {best*1e3:.1f} ms  {n/best/1e9:.2f} Gelem/s", flush=True)
