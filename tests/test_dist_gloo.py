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
    import exblas_amd as ex
    for n in (0, 1, 2, 7, 1000, (1 << 28) + 3):
        for world in (1, 2, 3, 4, 8):
            cuts = [ex.shard_range(n, r, world) for r in range(world)]
            assert cuts[0][0] == 0 and cuts[-1][1] == n
            for (a0, a1), (b0, b1) in zip(cuts, cuts[1:]):
                assert a1 == b0 and a0 <= a1
            assert all(c[0] % 2 == 0 for c in cuts)
