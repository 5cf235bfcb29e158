"""GPU tests of the native multi-GPU path (csrc/comm.hip) with several ranks SHARING the one GPU of the test box.

Every rank calls the library's own entry points -- exblas_exsum_allreduce_dev, exblas_exdot_allreduce_dev,
exblas_exgemv_sharded_dev, exblas_exgemm_sharded_dev -- with the HIP kernels doing the work; the collectives go through
the library's host-callback transport over gloo (RCCL refuses two ranks on one device).  The results must be
bit-identical to the single-rank results for 2 and 3 ranks, odd row splits included.  The RCCL transport itself is
exercised with a one-rank communicator (test_rccl_transport_one_rank) and by bench.py --gpus N on a multi-GPU node."""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


M, N, K = 301, 222, 259     # gemm / gemv shapes: odd, so the even-cut shards are unequal
NSUM = (1 << 22) + 10


def _inputs(ex):
    """the full (unsharded) operands, generated on the GPU by the counter-based generators"""
    x = ex.gen_dev("ill_cond", NSUM, 1, 1e32)
    y = ex.gen_dev("lognormal", NSUM, 2, 0.0, 2.0)
    a_cm = ex.gen_dev("fpuniform_signed", M * N, 81, 40, 20)      # column-major M x N (gemv)
    xv = ex.gen_dev("fpuniform_signed", max(M, N), 82, 40, 20)
    yv = ex.gen_dev("fpuniform_signed", max(M, N), 83, 40, 20)
    A = ex.gen_dev("fpuniform", M * K, 84, 10, 0)                  # row-major M x K
    B = ex.gen_dev("fpuniform", K * N, 85, 10, 0)
    C0 = ex.gen_dev("fpuniform", M * N, 86, 10, 0)
    return x, y, a_cm, xv, yv, A, B, C0


def _run_all(ex, torch, comm, rank, world):
    """what one rank does; world == 1 with comm None computes the single-rank reference through the plain *_dev calls"""
    x, y, a_cm, xv, yv, A, B, C0 = _inputs(ex)
    out = {}
    first, last = ex.shard_range(NSUM, rank, world)
    for fpe, ee in ((8, True), (0, False)):
        if comm is None:
            rs, rd = ex.exsum_dev(x, fpe, ee), ex.exdot_dev(x, y, fpe, ee)
        else:
            rs = ex.exsum_allreduce(comm, x[first:last], fpe, ee)
            rd = ex.exdot_allreduce(comm, x[first:last], y[first:last], fpe, ee)
        for name, r in (("sum", rs), ("dot", rd)):
            rec = ex.read_record(r)
            out[f"{name}{fpe}"] = (rec.exact, rec.refmode, rec.canon.tolist())
    # products below 2^-968 (their low parts live in the LOW accumulator): every rank exports its low digit set, both sets
    # are all-reduced, the fold happens once -- the same 8 bytes and flag bits 3 + 5 for every rank count
    nu = 200000 + 6
    xu = ex.gen_dev("fpuniform_signed", nu, 91, 40, 20)
    yu = ex.gen_dev("fpuniform_signed", nu, 92, 40, 20)
    xu[: nu // 2] *= 2.0 ** -520
    yu[: nu // 2] *= 2.0 ** -500
    u0, u1 = ex.shard_range(nu, rank, world)
    for fpe, ee in ((8, True), (0, False), (4, False)):
        r = ex.exdot_dev(xu, yu, fpe, ee) if comm is None else ex.exdot_allreduce(comm, xu[u0:u1], yu[u0:u1], fpe, ee)
        rec = ex.read_record(r)
        assert rec.flags == 8 | 32, rec.flags
        out[f"dot_under{fpe}"] = (rec.exact, rec.flags)
    # products beyond the double range (HIGH accumulator), spread over the shards so that they cancel only ACROSS ranks
    # (a rank alone would overflow), mixed with underflowing ones: the same 8 finite bytes, flag bits 3..6, for every rank count
    xo, yo = xu.clone(), yu.clone()
    q = (nu - nu // 2 - 6) // 2              # (the last six elements stay ordinary)
    xo[nu // 2: nu // 2 + q] *= 2.0 ** 560
    yo[nu // 2: nu // 2 + q] *= 2.0 ** 540
    xo[nu // 2 + q: nu // 2 + 2 * q] = -xo[nu // 2: nu // 2 + q]
    yo[nu // 2 + q: nu // 2 + 2 * q] = yo[nu // 2: nu // 2 + q]
    for fpe, ee in ((8, True), (0, False), (4, False)):
        r = ex.exdot_dev(xo, yo, fpe, ee) if comm is None else ex.exdot_allreduce(comm, xo[u0:u1], yo[u0:u1], fpe, ee)
        rec = ex.read_record(r)
        assert rec.flags == 8 | 16 | 32 | 64 and np.isfinite(rec.exact), (rec.flags, rec.exact)
        out[f"dot_over{fpe}"] = (rec.exact, rec.flags)
    if comm is None:
        from oracle import pyoracle as O
        if O.mpfr() is not None:
            assert out["dot_under8"][0] == O.mpfr_exdot(xu.cpu().numpy(), yu.cpu().numpy())
            assert out["dot_over8"][0] == O.mpfr_exdot(xo.cpu().numpy(), yo.cpu().numpy())
    r0, r1 = ex.row_block(M, rank, world)
    c0, c1 = ex.row_block(N, rank, world)
    for fpe, ee in ((8, True), (0, False), (4, False)):
        # gemv 'N': rows r0..r1 of A (column-major: a strided slice -> made contiguous with lda = r1 - r0) and of y
        yN = torch.zeros(M, dtype=torch.float64, device="cuda")
        yN[r0:r1] = yv[:M][r0:r1]
        xN = xv[:N].clone() if rank == 0 else torch.zeros(N, dtype=torch.float64, device="cuda")
        if comm is None:
            ex.exgemv_dev("N", M, N, 1.5, a_cm, M, xN, 1.0, yN, fpe, ee)
        else:
            a_loc = a_cm.view(N, M)[:, r0:r1].contiguous()
            ex.exgemv_sharded(comm, "N", M, N, 1.5, a_loc, max(r1 - r0, 1), xN, 1.0, yN, fpe, ee)
            assert (xN == xv[:N]).all()                      # x was broadcast from rank 0
        out[f"gemvN{fpe}"] = yN.cpu().numpy().view(np.int64).tolist()
        # gemv 'T': outputs c0..c1 = columns of A
        yT = torch.zeros(N, dtype=torch.float64, device="cuda")
        yT[c0:c1] = yv[:N][c0:c1]
        xT = xv[:M].clone() if rank == 0 else torch.zeros(M, dtype=torch.float64, device="cuda")
        if comm is None:
            ex.exgemv_dev("T", M, N, 1.0, a_cm, M, xT, 1.0, yT, fpe, ee)
        else:
            ex.exgemv_sharded(comm, "T", M, N, 1.0, a_cm[c0 * M:], M, xT, 1.0, yT, fpe, ee)
        out[f"gemvT{fpe}"] = yT.cpu().numpy().view(np.int64).tolist()
    for fpe, ee in ((8, True), (0, False)):
        Cm = torch.zeros(M * N, dtype=torch.float64, device="cuda")
        Cm[r0 * N:r1 * N] = C0[r0 * N:r1 * N]
        Bm = B.clone() if rank == 0 else torch.zeros(K * N, dtype=torch.float64, device="cuda")
        if comm is None:
            ex.exgemm_dev("N", "N", M, N, K, 1.0, A, K, Bm, N, 1.0, Cm, N, fpe, ee)
        else:
            ex.exgemm_sharded(comm, M, N, K, 1.0, A[r0 * K:], Bm, 1.0, Cm, fpe, ee)
            assert (Bm == B).all()
        out[f"gemm{fpe}"] = Cm.cpu().numpy().view(np.int64).tolist()
    # gather = 0: y / C stay sharded -- only the rank's own block is written and no collective follows the product
    # (B / x already replicated: b_root = x_root = -1, so the call posts no collective at all).  The untouched part keeps
    # its sentinel; the own block equals the single-rank result.
    sentinel = -12345.0
    yS = torch.full((M,), sentinel, dtype=torch.float64, device="cuda")
    yS[r0:r1] = yv[:M][r0:r1]
    Cs = torch.full((M * N,), sentinel, dtype=torch.float64, device="cuda")
    Cs[r0 * N:r1 * N] = C0[r0 * N:r1 * N]
    if comm is None:
        yS[:] = yv[:M]
        Cs[:] = C0
        ex.exgemv_dev("N", M, N, 1.5, a_cm, M, xv[:N].clone(), 1.0, yS, 8, True)
        ex.exgemm_dev("N", "N", M, N, K, 1.0, A, K, B, N, 1.0, Cs, N, 8, True)
        out["sharded_nogather"] = (yS.cpu().numpy().view(np.int64).tolist(), Cs.cpu().numpy().view(np.int64).tolist())
    else:
        a_loc = a_cm.view(N, M)[:, r0:r1].contiguous()
        ex.exgemv_sharded(comm, "N", M, N, 1.5, a_loc, max(r1 - r0, 1), xv[:N].clone(), 1.0, yS, 8, True, x_root=-1,
                          gather=False)
        ex.exgemm_sharded(comm, M, N, K, 1.0, A[r0 * K:], B, 1.0, Cs, 8, True, b_root=-1, gather=False)
        yh, ch = yS.cpu().numpy(), Cs.cpu().numpy()
        mask_y = np.ones(M, bool); mask_y[r0:r1] = False
        mask_c = np.ones(M * N, bool); mask_c[r0 * N:r1 * N] = False
        assert (yh[mask_y] == sentinel).all() and (ch[mask_c] == sentinel).all()
        out["sharded_nogather"] = (r0, r1, yh[r0:r1].view(np.int64).tolist(), ch[r0 * N:r1 * N].view(np.int64).tolist())
    # the reference's silent no-op (early_exit with fpe > 8) in a communicator where some ranks own no rows (m = 2 over
    # 3 ranks): every rank must return without posting a collective (a mismatch would hang the transport)
    if comm is not None:
        m2 = 2
        q0, q1 = ex.row_block(m2, rank, world)
        C2 = torch.full((m2 * N,), 7.0, dtype=torch.float64, device="cuda")
        ex.exgemm_sharded(comm, m2, N, K, 1.0, A[q0 * K:], B.clone(), 1.0, C2, 9, True)
        y2 = torch.full((m2,), 7.0, dtype=torch.float64, device="cuda")
        ex.exgemv_sharded(comm, "N", m2, N, 1.0, a_cm, max(q1 - q0, 1), xv[:N].clone(), 1.0, y2, 9, True)
        assert (C2 == 7.0).all() and (y2 == 7.0).all()
        # ... and the same shapes with a variant that does compute: ranks without rows still take part in every chunk
        C3 = torch.zeros(m2 * N, dtype=torch.float64, device="cuda")
        ex.exgemm_sharded(comm, m2, N, K, 1.0, A[q0 * K:], B.clone(), 0.0, C3, 8, True)
        want = torch.zeros(m2 * N, dtype=torch.float64, device="cuda")
        ex.exgemm_dev("N", "N", m2, N, K, 1.0, A, K, B, N, 0.0, want, N, 8, True)
        assert torch.equal(C3.view(torch.int64), want.view(torch.int64))
    torch.cuda.synchronize()
    return out


def _worker(rank, world, port, q):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import exblas_amd as ex
        torch.cuda.set_device(0)
        comm = ex.Comm.from_torch()          # gloo group -> the library's host-callback transport
        assert comm.size == world and comm.rank == rank
        q.put((rank, _run_all(ex, torch, comm, rank, world)))
        comm.destroy()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_shared_gpu_ranks_bit_identical(world):
    import torch
    import torch.multiprocessing as mp
    import exblas_amd as ex
    one = _run_all(ex, torch, None, 0, 1)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=400) for _ in range(world)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    y_one, c_one = one["sharded_nogather"]
    for rank, out in res:
        assert out.keys() == one.keys()
        for key in one:
            if key == "sharded_nogather":
                r0, r1, yb, cb = out[key]
                assert yb == y_one[r0:r1] and cb == c_one[r0 * N:r1 * N], (world, rank, key)
            else:
                assert out[key] == one[key], (world, rank, key)


def test_rccl_transport_one_rank():
    """The RCCL transport end to end (ncclGetUniqueId, ncclCommInitRank, ncclAllReduce / ncclBroadcast /
    ncclAllGather enqueued by libexblas.so on the caller's stream) with a communicator of one rank: same bits as the
    plain calls, and the whole sequence is capturable into a hipGraph (no host synchronisation anywhere)."""
    import torch
    import exblas_amd as ex
    os.environ["EXBLAS_COMM_FORCE"] = "1"     # issue the broadcasts / all-gathers although there is one rank
    comm = ex.Comm.rccl(ex.Comm.unique_id(), 0, 1)
    one = _run_all(ex, torch, None, 0, 1)
    got = _run_all(ex, torch, comm, 0, 1)
    for key in one:
        if key == "sharded_nogather":
            assert list(got[key][2:]) == list(one[key]), key
        else:
            assert got[key] == one[key], key
    # row chunks: 2200 local rows are produced in 4 chunks (cuts at multiples of 64 rows), each shipped by its own group
    # of broadcasts on the side stream while the next chunk is computed; operands scanned and sliced once
    m, n, k = 2200, 96, 130
    A = ex.gen_dev("fpuniform_signed", m * k, 87, 20, 10)
    B = ex.gen_dev("fpuniform_signed", k * n, 88, 20, 10)
    C0 = ex.gen_dev("fpuniform_signed", m * n, 89, 20, 10)
    for fpe, ee in ((8, True), (0, False)):
        want_c = C0.clone()
        ex.exgemm_dev("N", "N", m, n, k, 1.0, A, k, B, n, 1.0, want_c, n, fpe, ee)
        got_c = C0.clone()
        ex.exgemm_sharded(comm, m, n, k, 1.0, A, B, 1.0, got_c, fpe, ee)
        torch.cuda.synchronize()
        assert torch.equal(want_c.view(torch.int64), got_c.view(torch.int64)), (fpe, ee)
    # pipelined form: one call per reduction, the second half on the communicator's side stream; seven reductions in
    # flight two at a time (slots alternate, each reused three times), every record equal to the plain call's
    vecs = [ex.gen_dev("ill_cond", (1 << 20) + 17 * i, 30 + i, 1e32) for i in range(7)]
    ws = [ex.gen_dev("lognormal", (1 << 20) + 17 * i, 40 + i, 0.0, 2.0) for i in range(7)]
    want_s = [ex.read_record(ex.exsum_dev(v, 8, True)) for v in vecs]
    want_d = [ex.read_record(ex.exdot_dev(v, w, 6, True)) for v, w in zip(vecs, ws)]
    recs_s = [ex.new_record_buffer() for _ in vecs]
    recs_d = [ex.new_record_buffer() for _ in vecs]
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    t1.record()
    for i, v in enumerate(vecs):
        ex.exsum_allreduce_pipelined(comm, v, 8, True, out=recs_s[i], ev_start=t0 if i == 3 else None,
                                     ev_end=t1 if i == 3 else None)
    for i, (v, w) in enumerate(zip(vecs, ws)):
        ex.exdot_allreduce_pipelined(comm, v, w, 6, True, out=recs_d[i])
    ex.pipeline_drain(comm)
    torch.cuda.synchronize()
    assert t0.elapsed_time(t1) > 0.0
    for i in range(len(vecs)):
        for got, want in ((ex.read_record(recs_s[i]), want_s[i]), (ex.read_record(recs_d[i]), want_d[i])):
            assert got.exact == want.exact and got.refmode == want.refmode and (got.canon == want.canon).all(), i
    again = ex.read_record(ex.exsum_dev(vecs[0], 8, True))       # slot 0 is selected again, accumulators are zero
    assert again.exact == want_s[0].exact and (again.canon == want_s[0].canon).all()
    # graph capture of exsum + all-reduce + finalize
    x = ex.gen_dev("ill_cond", 1 << 20, 3, 1e32)
    rec = ex.new_record_buffer()
    ex.exsum_allreduce(comm, x, 8, True, out=rec)       # allocations and RCCL's lazy set-up happen outside the capture
    want = ex.read_record(rec)
    s = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    rec2 = ex.new_record_buffer()
    torch.cuda.synchronize()
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s):
            ex.exsum_allreduce(comm, x, 8, True, out=rec2)
    rec2.zero_()
    g.replay()
    torch.cuda.synchronize()
    r2 = ex.read_record(rec2)
    assert r2.exact == want.exact and (r2.canon == want.canon).all()
    comm.destroy()


def test_standalone_cpp_comm_caller():
    """tests/cpp/test_comm.cpp: the multi-GPU entry points called from C++ (no Python, no torch in the process) over
    both transports, bit-identical to the plain single-GPU calls."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.run(["make", "-C", os.path.join(root, "tests", "cpp")], check=True, capture_output=True)
    r = subprocess.run([os.path.join(root, "tests", "cpp", "test_comm")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "TestPassed; ALL OK!" in r.stdout, (r.stdout[-2000:], r.stderr[-2000:])
