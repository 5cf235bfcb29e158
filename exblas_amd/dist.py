"""Multi-GPU path: thin Python mirror of the native communicator layer of libexblas.so (csrc/comm.hip).

The reference's distributed path is one ``MPI_Reduce`` of the 41 normalised limbs (MPI_LONG, MPI_SUM) followed by
``Round`` on the root, inside the library call (src/cpu/blas/blas1/ExSUM.cpp:142-152, :266-273; scatter :33-63).
Here every rank (one process per GPU) reduces its shard to a normalised digit set that stays in HBM (72 int64 =
576 B; plus the low and high digit sets of the same size, all zero unless ExDOT products left the double range), ``ncclAllReduce(ncclInt64, ncclSum)`` -- called from C++ on the caller's stream, RCCL over xGMI -- adds them,
and every rank runs the same carry-propagation + rounding kernel.  ExGEMV / ExGEMM shard the outputs: x resp. B is
broadcast, y resp. C all-gathered (``exblas_exgemv_sharded_dev`` / ``exblas_exgemm_sharded_dev``).

All collectives happen in ``libexblas.so``.  ``torch.distributed`` is used here for ONE thing: handing the 128-byte
RCCL unique id from rank 0 to the other ranks when the communicator is created (any out-of-band channel would do;
a C++ program would use MPI_Bcast or a file).  For process groups that are not RCCL-backed (``gloo``: several ranks on
one GPU, CPU-side rehearsals) ``Comm.from_torch`` plugs torch.distributed collectives into the library's
host-callback transport instead -- the same C entry points run either way.
"""
import ctypes as C

import numpy as np


def shard_range(n, rank, world):
    """[first, last) of rank's contiguous shard; boundaries are even so every shard stays 16-byte aligned.
    Same function as exblas_shard_range (csrc/comm.hip); pure Python so it works without the library."""
    def cut(r):
        c = (n * r) // world
        return n if r >= world else (c & ~1)
    return cut(rank), cut(rank + 1)


def row_block(m, rank, world):
    """[first, last) rows (ExGEMV 'N', ExGEMM) or outputs (ExGEMV 'T') owned by `rank`."""
    return shard_range(m, rank, world)


class Comm:
    """Owner of an ``exblas_comm_t *``.  Create with ``Comm.from_torch(group)``, ``Comm.rccl(...)`` or ``Comm.host(...)``."""

    def __init__(self, handle, rank, size, keepalive=()):
        self.handle = handle
        self.rank = rank
        self.size = size
        self._keepalive = keepalive  # ctypes callbacks must outlive the communicator

    # -- constructors ---------------------------------------------------------------------------
    @staticmethod
    def unique_id():
        from . import load_library, _check, UNIQUE_ID_BYTES
        buf = (C.c_ubyte * UNIQUE_ID_BYTES)()
        _check(load_library().exblas_comm_unique_id(C.cast(buf, C.c_void_p)), "comm_unique_id")
        return bytes(buf)

    @staticmethod
    def rccl(uid, rank, size):
        """ncclCommInitRank on the current device with a unique id made by ``Comm.unique_id()`` on one rank."""
        from . import load_library, _check, _require_gpu
        _require_gpu()
        h = C.c_void_p()
        buf = (C.c_ubyte * len(uid)).from_buffer_copy(uid)
        _check(load_library().exblas_comm_init_rccl(C.byref(h), size, rank, C.cast(buf, C.c_void_p)), "comm_init_rccl")
        return Comm(h, rank, size)

    @staticmethod
    def host(rank, size, allreduce, bcast, allgatherv):
        """Host-callback transport.  The three callables get numpy views of the library's host buffer:
        allreduce(int64 array) in-place sum; bcast(uint8 array, root); allgatherv(uint8 array, offsets list)."""
        from . import load_library, _check, HOST_ALLREDUCE_FN, HOST_BCAST_FN, HOST_ALLGATHERV_FN

        def view(ptr, nbytes, dtype):
            return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_ubyte)), shape=(nbytes,)).view(dtype)

        def _ar(_user, buf, count):
            try:
                allreduce(view(buf, count * 8, np.int64))
                return 0
            except Exception:  # noqa: BLE001  (an exception must not unwind through C)
                import traceback
                traceback.print_exc()
                return 1

        def _bc(_user, buf, nbytes, root):
            try:
                bcast(view(buf, nbytes, np.uint8), root)
                return 0
            except Exception:  # noqa: BLE001
                import traceback
                traceback.print_exc()
                return 1

        def _ag(_user, buf, off):
            try:
                offs = [int(off[i]) for i in range(size + 1)]
                allgatherv(view(buf, offs[-1], np.uint8), offs)
                return 0
            except Exception:  # noqa: BLE001
                import traceback
                traceback.print_exc()
                return 1

        cbs = (HOST_ALLREDUCE_FN(_ar), HOST_BCAST_FN(_bc), HOST_ALLGATHERV_FN(_ag))
        h = C.c_void_p()
        _check(load_library().exblas_comm_init_host(C.byref(h), size, rank, cbs[0], cbs[1], cbs[2], None),
               "comm_init_host")
        return Comm(h, rank, size, keepalive=cbs)

    @staticmethod
    def from_torch(group=None, transport=None):
        """Communicator spanning a torch.distributed process group.  transport 'rccl' (default for an nccl group):
        a fresh RCCL communicator inside libexblas.so, bootstrapped with one broadcast of the unique id;
        'host' (default otherwise): the library's host-callback transport driven by the group's own collectives."""
        import torch.distributed as dist
        rank, size = dist.get_rank(group), dist.get_world_size(group)
        if transport is None:
            transport = "rccl" if dist.get_backend(group) == "nccl" else "host"
        if transport == "rccl":
            box = [Comm.unique_id() if rank == 0 else None]
            dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
            return Comm.rccl(box[0], rank, size)

        return Comm.host(rank, size, *torch_host_transport(group))

    def destroy(self):
        if self.handle is not None:
            from . import load_library
            load_library().exblas_comm_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.destroy()
        except Exception:  # noqa: BLE001
            pass


def torch_host_transport(group=None):
    """(allreduce, bcast, allgatherv) callables for ``Comm.host`` that move host memory with torch.distributed
    (any backend that handles CPU tensors, i.e. gloo).  They act in place on numpy arrays."""
    import torch
    import torch.distributed as dist
    size = dist.get_world_size(group)

    def g(r):
        return dist.get_global_rank(group, r) if group is not None else r

    def allreduce(a):
        dist.all_reduce(torch.from_numpy(a), op=dist.ReduceOp.SUM, group=group)

    def bcast(a, root):
        dist.broadcast(torch.from_numpy(a), src=g(root), group=group)

    def allgatherv(a, offs):
        t = torch.from_numpy(a)
        for r in range(size):
            if offs[r + 1] > offs[r]:
                dist.broadcast(t[offs[r]:offs[r + 1]], src=g(r), group=group)

    return allreduce, bcast, allgatherv


def _args(torch):
    from . import _stream_ptr
    return _stream_ptr(torch)


def exsum_allreduce(comm, x_local, fpe=8, early_exit=True, out=None, inca=1, n=None):
    """Exact sum of the concatenation of every rank's ``x_local`` (CUDA float64); record tensor, same on every rank."""
    from . import load_library, _check, _require_gpu, new_record_buffer
    torch = _require_gpu()
    if n is None:
        n = (x_local.numel() + inca - 1) // inca
    if out is None:
        out = new_record_buffer()
    _check(load_library().exblas_exsum_allreduce_dev(comm.handle, C.c_void_p(x_local.data_ptr()), n, inca, fpe,
                                                     int(early_exit), _args(torch), C.c_void_p(out.data_ptr())),
           "exsum_allreduce_dev")
    return out


def exdot_allreduce(comm, x_local, y_local, fpe=8, early_exit=True, out=None):
    from . import load_library, _check, _require_gpu, new_record_buffer
    torch = _require_gpu()
    if out is None:
        out = new_record_buffer()
    _check(load_library().exblas_exdot_allreduce_dev(comm.handle, C.c_void_p(x_local.data_ptr()), 1,
                                                     C.c_void_p(y_local.data_ptr()), 1, x_local.numel(), fpe,
                                                     int(early_exit), _args(torch), C.c_void_p(out.data_ptr())),
           "exdot_allreduce_dev")
    return out


def _evh(ev):
    return C.c_void_p(ev.cuda_event) if ev is not None else None


def exsum_allreduce_pipelined(comm, x_local, fpe=8, early_exit=True, out=None, ev_start=None, ev_end=None):
    """One call per reduction; the second half overlaps the next call's streaming kernel (see
    ``exblas_exsum_allreduce_pipelined_dev``).  ev_start / ev_end: torch.cuda.Event (created with enable_timing=True and
    already recorded once, so that they have a handle) recorded around the streaming kernel."""
    from . import load_library, _check, _require_gpu, new_record_buffer
    torch = _require_gpu()
    if out is None:
        out = new_record_buffer()
    _check(load_library().exblas_exsum_allreduce_pipelined_dev(comm.handle, C.c_void_p(x_local.data_ptr()),
                                                               x_local.numel(), 1, fpe, int(early_exit), _args(torch),
                                                               C.c_void_p(out.data_ptr()), _evh(ev_start), _evh(ev_end)),
           "exsum_allreduce_pipelined_dev")
    return out


def exdot_allreduce_pipelined(comm, x_local, y_local, fpe=8, early_exit=True, out=None, ev_start=None, ev_end=None):
    from . import load_library, _check, _require_gpu, new_record_buffer
    torch = _require_gpu()
    if out is None:
        out = new_record_buffer()
    _check(load_library().exblas_exdot_allreduce_pipelined_dev(comm.handle, C.c_void_p(x_local.data_ptr()), 1,
                                                               C.c_void_p(y_local.data_ptr()), 1, x_local.numel(), fpe,
                                                               int(early_exit), _args(torch), C.c_void_p(out.data_ptr()),
                                                               _evh(ev_start), _evh(ev_end)),
           "exdot_allreduce_pipelined_dev")
    return out


def pipeline_drain(comm):
    """The current stream waits for the pipelined second halves in flight; slot 0 selected again."""
    from . import load_library, _check, _require_gpu
    torch = _require_gpu()
    _check(load_library().exblas_pipeline_drain_dev(comm.handle, _args(torch)), "pipeline_drain_dev")


def allreduce_finish(comm, out=None):
    """Second half of the calls above on the current stream: normalise the selected accumulator slot, all-reduce the
    digit set, carry-propagate + round (see ``exblas_allreduce_finish_dev``)."""
    from . import load_library, _check, _require_gpu, new_record_buffer
    torch = _require_gpu()
    if out is None:
        out = new_record_buffer()
    _check(load_library().exblas_allreduce_finish_dev(comm.handle, _args(torch), C.c_void_p(out.data_ptr())),
           "allreduce_finish_dev")
    return out


def exgemv_sharded(comm, trans, m, n, alpha, a_local, lda, x, beta, y, fpe=8, early_exit=True, x_root=0, incx=1,
                   incy=1, gather=True):
    """y := alpha*op(A)*x + beta*y with the outputs sharded over the ranks; see ``exblas_exgemv_sharded_dev``.
    gather=False leaves y sharded (only the rank's own part is written, no collective after the product)."""
    from . import load_library, _check, _require_gpu
    torch = _require_gpu()
    _check(load_library().exblas_exgemv_sharded_dev(comm.handle, trans.encode(), m, n, alpha,
                                                    C.c_void_p(a_local.data_ptr()), lda, C.c_void_p(x.data_ptr()),
                                                    incx, x_root, beta, C.c_void_p(y.data_ptr()), incy, int(gather),
                                                    fpe, int(early_exit), _args(torch)), "exgemv_sharded_dev")
    return y


def exgemm_sharded(comm, m, n, k, alpha, a_local, b, beta, c, fpe=8, early_exit=True, b_root=0, transa="N",
                   transb="N", lda=None, ldb=None, ldc=None, gather=True):
    """C := alpha*op(A)*op(B) + beta*C (row-major) with the rows sharded over the ranks; see
    ``exblas_exgemm_sharded_dev``.  gather=False leaves C sharded (with b_root < 0: no collective at all)."""
    from . import load_library, _check, _require_gpu
    torch = _require_gpu()
    lda = lda if lda is not None else (k if transa in "Nn" else m)
    ldb = ldb if ldb is not None else (n if transb in "Nn" else k)
    ldc = ldc if ldc is not None else n
    _check(load_library().exblas_exgemm_sharded_dev(comm.handle, transa.encode(), transb.encode(), m, n, k, alpha,
                                                    C.c_void_p(a_local.data_ptr()), lda, C.c_void_p(b.data_ptr()),
                                                    ldb, b_root, beta, C.c_void_p(c.data_ptr()), ldc, int(gather),
                                                    fpe, int(early_exit), _args(torch)), "exgemm_sharded_dev")
    return c


def allreduce_record(rec, group=None, force=False):
    """In-place int64 SUM all-reduce of the digit set (words 48..119) of a record tensor with torch.distributed.
    Kept for host-side rehearsals of the arithmetic (tests/test_dist_gloo.py); the product path is ``Comm``."""
    import torch.distributed as dist
    from . import OUT_DIGITS, SET_WORDS
    payload = rec[OUT_DIGITS:OUT_DIGITS + SET_WORDS]
    if dist.is_available() and dist.is_initialized() and (force or dist.get_world_size(group) > 1):
        dist.all_reduce(payload, op=dist.ReduceOp.SUM, group=group)
    return rec
