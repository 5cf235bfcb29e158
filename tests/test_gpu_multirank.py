"""GPU test of the N > 1 path with several ranks SHARING the one GPU of the test box (gloo transport, HIP
kernels): every rank reduces its shard with the HIP kernels, the digit sets are all-reduced, every rank runs the
second finalize -- and the result must be bit-identical to the single-rank result, for 2 and 3 ranks, ExSUM and ExDOT.
(The RCCL transport itself needs one GPU per rank; bench.py rehearses that call sequence with a 1-rank nccl group.)"""
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


def _worker(rank, world, port, n, q):
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
        first, last = ex.shard_range(n, rank, world)
        x = ex.gen_dev("ill_cond", n, 1, 1e32, first=first, count=last - first, n_total=n)
        y = ex.gen_dev("lognormal", n, 2, 0.0, 2.0, first=first, count=last - first, n_total=n)
        out = []
        for op in ("exsum", "exdot"):
            rec = ex.exsum_dev(x, 8, True) if op == "exsum" else ex.exdot_dev(x, y, 8, True)
            host = rec.cpu()                                  # gloo moves host memory
            ex.allreduce_record(host)
            rec.copy_(host)
            ex.finalize_dev(rec[ex.OUT_DIGITS:ex.OUT_DIGITS + ex.SET_WORDS], out=rec)
            r = ex.read_record(rec)
            out.append((r.exact, r.canon.tolist()))
        q.put((rank, out))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_shared_gpu_ranks_bit_identical(world):
    import torch
    import torch.multiprocessing as mp
    import exblas_amd as ex
    n = (1 << 22) + 10
    x = ex.gen_dev("ill_cond", n, 1, 1e32)
    y = ex.gen_dev("lognormal", n, 2, 0.0, 2.0)
    one = [ex.read_record(ex.exsum_dev(x, 8, True)), ex.read_record(ex.exdot_dev(x, y, 8, True))]
    torch.cuda.synchronize()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, out in res:
        for k in range(2):
            assert out[k][0] == one[k].exact and (np.array(out[k][1]) == one[k].canon).all(), (rank, k)
