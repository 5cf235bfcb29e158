"""ExDOT rate against the relative placement of the two vectors: python tools/dot_align.py [log2n]
Both vectors live in one allocation; b starts `pad` bytes after the end of a.  Rotates over 3 such pairs so the
Infinity Cache cannot serve a step."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import exblas_amd as ex
log2n = int(sys.argv[1]) if len(sys.argv) > 1 else 28
n = 1 << log2n
ex.load_library().exblas_hip_init(-1)
src_a = ex.gen_dev("ill_cond", n, 1, 1e32, 0)
src_b = ex.gen_dev("ill_cond", n, 2, 1e32, 0)
pads = [int(v) for v in sys.argv[2].split(',')] if len(sys.argv) > 2 else [0, 256, 1024, 4096, 16384, 65536, 1 << 20, (1 << 20) + 4096, 3 << 19]
for pad in pads:
    pairs = []
    for j in range(3):
        blob = torch.empty(2 * n + (4 << 20) // 8, dtype=torch.float64, device="cuda")
        a = blob[:n]
        b = blob[n + pad // 8: 2 * n + pad // 8]
        a.copy_(src_a); b.copy_(src_b)
        pairs.append((a, b))
    for _ in range(6):
        for a, b in pairs:
            ex.exdot_accumulate_dev(a, b, 8, True); ex.finish_dev()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 30
    e0.record()
    for i in range(reps):
        a, b = pairs[i % 3]
        ex.exdot_accumulate_dev(a, b, 8, True); ex.finish_dev()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print(f"pad {pad:8d} B: {ms:.4f} ms per step, {16.0 * n / ms / 1e6:.0f} GB/s", flush=True)
    del pairs, blob
