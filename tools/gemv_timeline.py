#!/usr/bin/env python3
"""Per-wave start / end clocks of k_gemvN_fpe_sx at 32768^2 (tools only: libexblas's exblas_debug_timeline hook).
usage: python tools/gemv_timeline.py [lda_extra] [variant]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import exblas_amd as ex
extra = int(sys.argv[1]) if len(sys.argv) > 1 else 0
variant = int(sys.argv[2]) if len(sys.argv) > 2 else 0
lib = ex.load_library()
lib.exblas_hip_init(-1)
lib.exblas_set_tuning(-1, -1, variant)
m = n = 32768
lda = m + extra
a = ex.gen_dev("fpuniform", lda * n, 1, 10.0, 0.0)
x = ex.gen_dev("fpuniform", n, 2, 10.0, 0.0)
y = ex.gen_dev("fpuniform", m, 3, 10.0, 0.0)
for _ in range(10):
    ex.exgemv_dev("N", m, n, 1.0, a, lda, x, 0.0, y, 8, True)
torch.cuda.synchronize()
gx, KS, W = 64, 128, 4          # upper bounds for the buffer
buf = torch.zeros(gx * KS * W * 2, dtype=torch.int64, device="cuda")
hook = C.c_void_p.in_dll(lib, "exblas_debug_timeline")
hook.value = buf.data_ptr()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
ex.exgemv_dev("N", m, n, 1.0, a, lda, x, 0.0, y, 8, True)
e1.record()
torch.cuda.synchronize()
hook.value = None
t = buf.cpu().numpy().reshape(-1, 2)
t = t[t[:, 0] != 0]
nw = t.shape[0]
t0 = t[:, 0].min()
st, en = (t[:, 0] - t0) / 100.0, (t[:, 1] - t0) / 100.0    # us (100 MHz)
print(f"lda=m+{extra} variant={variant}: call {e0.elapsed_time(e1)*1e3:.0f} us; {nw} waves; kernel span {en.max():.0f} us")
print("start  us: min %.0f p50 %.0f p99 %.0f max %.0f" % (st.min(), np.percentile(st, 50), np.percentile(st, 99), st.max()))
print("end    us: min %.0f p10 %.0f p50 %.0f p90 %.0f max %.0f" % (en.min(), np.percentile(en, 10), np.percentile(en, 50), np.percentile(en, 90), en.max()))
print("occupancy API: %d blocks/CU; waves started within 20 us: %d, within 200 us: %d" % (
    lib.exblas_debug_gemv_occupancy(), int((st < 20).sum()), int((st < 200).sum())))
life = en - st
print("life   us: min %.0f p50 %.0f max %.0f; mean life / span = %.3f" % (life.min(), np.percentile(life, 50), life.max(), life.mean() / en.max()))
# by row block (bx) and by k split: the order in the buffer is [ks][bx][wave]
KSn = nw // (gx * W)
e3 = en.reshape(KSn, gx, W)
print("end by bx (mean over ks, waves):", np.round(e3.mean(axis=(0, 2))[::4]).astype(int).tolist())
print("end by ks (mean over bx, waves):", np.round(e3.mean(axis=(1, 2))).astype(int).tolist())
