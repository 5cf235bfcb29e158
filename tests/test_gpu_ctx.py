"""Context handles (exblas_ctx_create / *_ctx): private accumulators, flags and workspace per handle.

The reference's GPU launchers keep kernels and buffers in file-static globals (src/gpu/blas/blas1/ExSUM.Launcher.cpp:16-36)
-- one call at a time per process.  The *_dev layer shares one accumulator set and one workspace per device (work must be
ordered on the device); handles remove that constraint.  Parity is unchanged: every result here is compared bit for bit
with the default-context call, which tests/test_gpu_blas1.py and test_gpu_blas23.py pin to the oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ex():
    import exblas_amd
    return exblas_amd


def _bits(t):
    return t.cpu().numpy().view(np.int64)


def test_interleaved_accumulations_merge_on_the_default_context_and_not_on_handles(ex):
    """The documented behaviour of the shared accumulators, and its cure: accumulate(x1), accumulate(y), accumulate(x2),
    finish on the default context is ONE exact sum over x1, y, x2 (several arrays folded into one reduction -- by
    design); on two handles the same interleaving yields sum(x1, x2) and sum(y), each equal to its stand-alone call."""
    import torch
    n = (1 << 20) + 5
    x1 = ex.gen_dev("ill_cond", n, 1, 1e32)
    x2 = ex.gen_dev("lognormal", n, 2, 0.0, 2.0)
    y = ex.gen_dev("fpuniform_signed", n, 3, 40, 20)
    # default context: everything between two finishes is one reduction
    ex.exsum_accumulate_dev(x1)
    ex.exsum_accumulate_dev(y)
    ex.exsum_accumulate_dev(x2)
    merged = ex.read_record(ex.finish_dev())
    whole = ex.read_record(ex.exsum_dev(torch.cat([x1, y, x2])))
    assert merged.exact == whole.exact and (merged.canon == whole.canon).all()
    # handles: independent
    ca, cb = ex.Context(), ex.Context()
    ca.exsum_accumulate(x1)
    cb.exsum_accumulate(y)
    ca.exsum_accumulate(x2)
    ra, rb = ex.read_record(ca.finish()), ex.read_record(cb.finish())
    wa = ex.read_record(ex.exsum_dev(torch.cat([x1, x2])))
    wb = ex.read_record(ex.exsum_dev(y))
    assert ra.exact == wa.exact and (ra.canon == wa.canon).all()
    assert rb.exact == wb.exact and (rb.canon == wb.canon).all()
    # the handles left the default context's accumulators untouched (still zero between calls)
    again = ex.read_record(ex.exsum_dev(y))
    assert again.exact == wb.exact and (again.canon == wb.canon).all()
    ca.destroy()
    cb.destroy()


def test_two_streams_two_handles_run_concurrently(ex):
    """ExSUM / ExDOT on one stream and ExGEMM / ExGEMV / ExTRSV on another, each through its own handle, enqueued
    back to back with no ordering between the streams: every result equals the serial default-context result."""
    import torch
    lib = ex.load_library()
    nsum = (1 << 24) + 7
    xs = [ex.gen_dev("ill_cond", nsum, 11 + i, 1e32) for i in range(3)]
    ys = [ex.gen_dev("lognormal", nsum, 21 + i, 0.0, 2.0) for i in range(3)]
    m, n, k = 1000, 700, 900
    A = ex.gen_dev("fpuniform_signed", m * k, 31, 12, 6)
    B = ex.gen_dev("fpuniform_signed", k * n, 32, 12, 6)
    C0 = ex.gen_dev("fpuniform_signed", m * n, 33, 10, 5)
    gm, gn = 3000, 2000
    GA = ex.gen_dev("fpuniform_signed", gm * gn, 34, 20, 10)
    gx = ex.gen_dev("fpuniform_signed", gn, 35, 20, 10)
    gy0 = ex.gen_dev("fpuniform_signed", gm, 36, 20, 10)
    tn = 700
    TA = ex.gen_dev("fpuniform", tn * tn, 37, 1.0, 0.0)
    TA.view(tn, tn).diagonal().copy_(ex.gen_dev("fpuniform", tn, 38, 1.0, 12.0))
    tb = ex.gen_dev("fpuniform_signed", tn, 39, 10, 0)
    # serial reference results on the default context
    want_sum = [ex.read_record(ex.exsum_dev(x)) for x in xs]
    want_dot = [ex.read_record(ex.exdot_dev(x, y)) for x, y in zip(xs, ys)]
    want_c = C0.clone()
    ex.exgemm_dev("N", "N", m, n, k, 1.0, A, k, B, n, 1.0, want_c, n, 8, True)
    want_y = gy0.clone()
    ex.exgemv_dev("N", gm, gn, 1.0, GA, gm, gx, 1.0, want_y, 8, True)
    want_t = tb.clone()
    ex.extrsv_dev("L", "N", "N", tn, TA, tn, want_t, 8, True)
    torch.cuda.synchronize()

    c1, c2 = ex.Context(), ex.Context()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    rounds = 6
    recs_sum = [[ex.new_record_buffer() for _ in xs] for _ in range(rounds)]
    recs_dot = [[ex.new_record_buffer() for _ in xs] for _ in range(rounds)]
    outs_c = [C0.clone() for _ in range(rounds)]
    outs_y = [gy0.clone() for _ in range(rounds)]
    outs_t = [tb.clone() for _ in range(rounds)]
    torch.cuda.synchronize()
    for r in range(rounds):
        with torch.cuda.stream(s1):
            for i, (x, y) in enumerate(zip(xs, ys)):
                c1.exsum(x, 8, True, out=recs_sum[r][i])
                c1.exdot(x, y, 8, True, out=recs_dot[r][i])
        with torch.cuda.stream(s2):
            c2.exgemm("N", "N", m, n, k, 1.0, A, k, B, n, 1.0, outs_c[r], n, 8, True)
            c2.exgemv("N", gm, gn, 1.0, GA, gm, gx, 1.0, outs_y[r], 8, True)
            c2.extrsv("L", "N", "N", tn, TA, tn, outs_t[r], 8, True)
    torch.cuda.synchronize()
    for r in range(rounds):
        for i in range(len(xs)):
            for got, want in ((ex.read_record(recs_sum[r][i]), want_sum[i]), (ex.read_record(recs_dot[r][i]), want_dot[i])):
                assert got.exact == want.exact and got.refmode == want.refmode and (got.canon == want.canon).all(), (r, i)
        assert (_bits(outs_c[r]) == _bits(want_c)).all(), r
        assert (_bits(outs_y[r]) == _bits(want_y)).all(), r
        assert (_bits(outs_t[r]) == _bits(want_t)).all(), r
    # each handle grew its OWN workspace; the default context's is what the serial calls left
    assert c2.workspace_bytes() > 0 and c1.workspace_bytes() == 0
    import ctypes as C
    info = (C.c_int * 8)()
    c2.exgemm("N", "N", m, n, k, 1.0, A, k, B, n, 1.0, outs_c[0], n, 8, True)   # (the info block lives in the workspace: valid
    assert lib.exblas_last_gemm_info_ctx(c2.handle, info) == 0 and info[0] == 4  #  until the handle's next gemv / trsv)
    c1.destroy()
    c2.destroy()


def test_handle_lifecycle(ex):
    """create / use / destroy repeatedly; NULL handle = the default context; knobs are inherited at creation"""
    import ctypes as C
    import torch
    lib = ex.load_library()
    x = ex.gen_dev("ill_cond", 100003, 5, 1e32)
    want = ex.read_record(ex.exsum_dev(x, 4, False))
    for _ in range(5):
        c = ex.Context()
        got = ex.read_record(c.exsum(x, 4, False))
        assert got.exact == want.exact and (got.canon == want.canon).all()
        c.destroy()
        c.destroy()                                    # idempotent
    rec = ex.new_record_buffer()
    rc = lib.exblas_exsum_ctx(None, C.c_void_p(x.data_ptr()), x.numel(), 1, 4, 0,
                              C.c_void_p(torch.cuda.current_stream().cuda_stream), C.c_void_p(rec.data_ptr()))
    assert rc == 0
    got = ex.read_record(rec)
    assert got.exact == want.exact and (got.canon == want.canon).all()
    # gemm path knob set before creation is what the handle uses
    m = n = k = 96
    A, B = ex.gen_dev("fpuniform", m * k, 6, 10, 0), ex.gen_dev("fpuniform", k * n, 7, 10, 0)
    try:
        lib.exblas_set_gemm_path(1)                    # scalar kernel only
        c = ex.Context()
        Cm = torch.zeros(m * n, dtype=torch.float64, device="cuda")
        c.exgemm("N", "N", m, n, k, 1.0, A, k, B, n, 0.0, Cm, n, 8, True)
        info = (C.c_int * 8)()
        assert lib.exblas_last_gemm_info_ctx(c.handle, info) == 0 and info[0] == 0
        c.destroy()
    finally:
        lib.exblas_set_gemm_path(0)


def test_launch_events_bracket_the_streaming_kernel(ex):
    """exblas_set_launch_events: the next accumulate call attaches the events to its kernel's dispatch packet -- they
    complete with the kernel, measure a plausible duration, are consumed by ONE launch, and the result is unchanged."""
    import torch
    n = (1 << 24) + 3
    x = ex.gen_dev("ill_cond", n, 3, 1e32)
    y = ex.gen_dev("lognormal", n, 4, 0.0, 2.0)
    want_s = ex.read_record(ex.exsum_dev(x, 8, True))
    want_d = ex.read_record(ex.exdot_dev(x, y, 8, True))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); e1.record()                       # (a torch event gets its handle at its first record())
    torch.cuda.synchronize()
    for op in ("sum", "dot"):
        ex.set_launch_events(e0, e1)
        if op == "sum":
            ex.exsum_accumulate_dev(x, 8, True)
        else:
            ex.exdot_accumulate_dev(x, y, 8, True)
        got = ex.read_record(ex.finish_dev())
        torch.cuda.synchronize()
        want = want_s if op == "sum" else want_d
        assert got.exact == want.exact and (got.canon == want.canon).all()
        ms = e0.elapsed_time(e1)
        bytes_ = n * (8 if op == "sum" else 16)
        assert 0.0 < ms < 5.0 and bytes_ / (ms * 1e-3) < 8.5e12, (op, ms)      # a kernel time, below the HBM peak
        # consumed: a second launch without events does not move them
        t_before = e0.elapsed_time(e1)
        ex.exsum_accumulate_dev(x, 8, True)
        ex.finish_dev()
        torch.cuda.synchronize()
        assert e0.elapsed_time(e1) == t_before
