"""CPU suite, world_size 2..4 over gloo: the N > 1 path of exblas_amd.dist.

What runs here is the product's own sharding + all-reduce code (shard_range, allreduce_record) on CPU
tensors; the per-rank digit sets are produced by the test (from the oracle's canonical limbs, converted
with Python integers) because the HIP kernels need a GPU.  The check: after the int64-sum all-reduce every
rank holds digits whose exact integer value equals the exact sum of the WHOLE vector, for any world size
and any shard boundary -- i.e. the result cannot depend on the GPU count.
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from helpers import digits_from_int, exact_int_from_canon, exact_int_from_digits


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, kind, p0, n, op, q):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    sys.path.insert(0, os.path.join(root, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import exblas_amd as ex
        from oracle import pyoracle as O
        first, last = ex.shard_range(n, rank, world)
        assert first % 2 == 0 and (last % 2 == 0 or last == n)
        a = O.gen(kind, n, 1, p0, 0.0, first=first, count=last - first, n_total=n)
        if op == "exsum":
            _, limbs = O.exsum(a, 8, True, limbs=True)
        else:
            b = O.gen(kind, n, 2, p0, 0.0, first=first, count=last - first, n_total=n)
            _, limbs = O.exdot(a, b, 8, True, limbs=True)
        v = exact_int_from_canon(limbs)
        assert v % (1 << 18) == 0
        rec = torch.zeros(ex.OUT_WORDS, dtype=torch.int64)
        rec[ex.OUT_DIGITS:ex.OUT_DIGITS + ex.NDIGITS] = torch.from_numpy(digits_from_int(v >> 18))
        rec[ex.OUT_DIGITS + ex.NDIGITS + 2] = 1 if rank == world - 1 else 0   # pretend the last rank saw a NaN
        ex.allreduce_record(rec)
        digits = rec[ex.OUT_DIGITS:ex.OUT_DIGITS + ex.NDIGITS].numpy()
        assert digits.max() < world * (1 << 32)
        q.put((rank, exact_int_from_digits(digits), int(rec[ex.OUT_DIGITS + ex.NDIGITS + 2])))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("op,kind,p0", [("exsum", "ill_cond", 1e32), ("exsum", "cancel", 50.0), ("exdot", "ill_cond", 1e32)])
def test_allreduce_is_shard_invariant(oracle, world, op, kind, p0):
    n = 40001
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, kind, p0, n, op, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    got = sorted(q.get(timeout=10) for _ in range(world))
    a = oracle.gen(kind, n, 1, p0, 0.0)
    if op == "exsum":
        _, limbs = oracle.exsum(a, 0, limbs=True)
    else:
        _, limbs = oracle.exdot(a, oracle.gen(kind, n, 2, p0, 0.0), 0, limbs=True)
    want = exact_int_from_canon(limbs) >> 18
    for rank, val, nanflag in got:
        assert val == want, (rank, world)
        assert nanflag == 1


def test_shard_range_covers_everything():
    """Python shard_range == the library's exblas_shard_range (what exblas_ex*_sharded_dev cut by)."""
    import ctypes as C
    import exblas_amd as ex
    lib = ex.load_library()
    for n in (0, 1, 2, 7, 1000, 8191, (1 << 28) + 3, (1 << 31) - 2):
        for world in (1, 2, 3, 4, 8):
            cuts = [ex.shard_range(n, r, world) for r in range(world)]
            assert cuts[0][0] == 0 and cuts[-1][1] == n
            for (a0, a1), (b0, b1) in zip(cuts, cuts[1:]):
                assert a1 == b0 and a0 <= a1
            assert all(c[0] % 2 == 0 for c in cuts)
            for r in range(world):
                f, l = C.c_int64(), C.c_int64()
                lib.exblas_shard_range(n, r, world, C.byref(f), C.byref(l))
                assert (f.value, l.value) == cuts[r]


def _shard_worker(rank, world, port, m, n, k, q):
    """One rank of the row-sharded ExGEMV / ExGEMM data path, with the oracle standing in for the HIP kernels:
    x / B broadcast from rank 0, own rows computed, y / C all-gathered through the SAME transport callables
    (exblas_amd.dist.torch_host_transport) the library's host-callback communicator is given on a GPU box."""
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import exblas_amd as ex
        from exblas_amd.dist import torch_host_transport, Comm
        from oracle import pyoracle as O
        allreduce, bcast, allgatherv = torch_host_transport()
        comm = Comm.host(rank, world, allreduce, bcast, allgatherv)   # the C object is created without a GPU
        assert ex.load_library().exblas_comm_size(comm.handle) == world
        assert ex.load_library().exblas_comm_rank(comm.handle) == rank
        # --- gemv 'N': rows sharded
        r0, r1 = ex.row_block(m, rank, world)
        a = O.gen("fpuniform_signed", m * n, 71, 40, 20)          # column-major m x n, lda = m
        x = O.gen("fpuniform_signed", n, 72, 40, 20) if rank == 0 else np.zeros(n)
        y = np.zeros(m)
        y[r0:r1] = O.gen("fpuniform_signed", m, 73, 40, 20)[r0:r1]
        bcast(x.view(np.uint8), 0)
        a_loc = np.ascontiguousarray(a.reshape(n, m)[:, r0:r1]).reshape(-1)      # (r1-r0) x n, lda = r1-r0
        if r1 > r0:
            y[r0:r1] = O.exgemv("N", r1 - r0, n, 1.0, a_loc, r1 - r0, x, 1.0, y[r0:r1].copy(), 0)
        offs = [ex.row_block(m, r, world)[0] * 8 for r in range(world)] + [m * 8]
        allgatherv(y.view(np.uint8), offs)
        # --- gemv 'T': outputs (columns) sharded
        c0, c1 = ex.row_block(n, rank, world)
        xt = O.gen("fpuniform_signed", m, 74, 40, 20) if rank == 0 else np.zeros(m)
        bcast(xt.view(np.uint8), 0)
        yt = np.zeros(n)
        if c1 > c0:
            yt[c0:c1] = O.exgemv("T", m, c1 - c0, 1.0, a[c0 * m:c1 * m], m, xt, 0.0, np.zeros(c1 - c0), 0)
        allgatherv(yt.view(np.uint8), [ex.row_block(n, r, world)[0] * 8 for r in range(world)] + [n * 8])
        # --- gemm: rows of A and C sharded, B broadcast
        A = O.gen("fpuniform", m * k, 75, 10, 0)
        B = O.gen("fpuniform", k * n, 76, 10, 0) if rank == 0 else np.zeros(k * n)
        bcast(B.view(np.uint8), 0)
        Cm = np.zeros(m * n)
        if r1 > r0:
            Cm[r0 * n:r1 * n] = O.exgemm("N", "N", r1 - r0, n, k, 1.0, A[r0 * k:r1 * k], k, B, n, 0.0,
                                         np.zeros((r1 - r0) * n), n, 0)
        allgatherv(Cm.view(np.uint8), [ex.row_block(m, r, world)[0] * n * 8 for r in range(world)] + [m * n * 8])
        # --- the 72-word digit-set all-reduce
        d = np.arange(72, dtype=np.int64) * (rank + 1)
        allreduce(d)
        q.put((rank, y.view(np.int64).tolist(), yt.view(np.int64).tolist(), Cm.view(np.int64).tolist(), d.tolist()))
        comm.destroy()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,m", [(2, 37), (3, 37), (3, 64), (4, 5)])
def test_sharded_gemv_gemm_data_path(oracle, world, m):
    """Gathered y / C of the row-sharded path == the single-rank result, bit for bit, for odd row splits (including
    ranks that own no rows at all: world 4, m 5)."""
    n, k = 29, 33
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_shard_worker, args=(r, world, port, m, n, k, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    a = oracle.gen("fpuniform_signed", m * n, 71, 40, 20)
    x = oracle.gen("fpuniform_signed", n, 72, 40, 20)
    y0 = oracle.gen("fpuniform_signed", m, 73, 40, 20)
    want_y = oracle.exgemv("N", m, n, 1.0, a, m, x, 1.0, y0, 0).view(np.int64).tolist()
    xt = oracle.gen("fpuniform_signed", m, 74, 40, 20)
    want_yt = oracle.exgemv("T", m, n, 1.0, a, m, xt, 0.0, np.zeros(n), 0).view(np.int64).tolist()
    A = oracle.gen("fpuniform", m * k, 75, 10, 0)
    B = oracle.gen("fpuniform", k * n, 76, 10, 0)
    want_c = oracle.exgemm("N", "N", m, n, k, 1.0, A, k, B, n, 0.0, np.zeros(m * n), n, 0).view(np.int64).tolist()
    want_d = (np.arange(72, dtype=np.int64) * sum(range(1, world + 1))).tolist()
    for rank, y, yt, c, d in got:
        assert y == want_y, rank
        assert yt == want_yt, rank
        assert c == want_c, rank
        assert d == want_d, rank
