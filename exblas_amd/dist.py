"""Multi-GPU ExSUM / ExDOT: contiguous shards + one int64-sum all-reduce of the digit set.

The reference's distributed path is one ``MPI_Reduce`` of the 41 normalised limbs (MPI_LONG, MPI_SUM)
followed by ``Round`` on the root (src/cpu/blas/blas1/ExSUM.cpp:142-152, :266-273).  Here every rank
(one process per GPU) reduces its shard to a normalised digit set that stays in HBM (72 int64 = 576 B:
68 digits < 2^32 plus three non-finite indicators), ``torch.distributed.all_reduce(SUM)`` -- RCCL over
xGMI with the ``nccl`` backend -- adds them, and every rank runs the same carry-propagation + rounding
kernel.  Integer addition is associative and commutative, so the result is bit-identical for any GPU
count, ring/tree order or shard boundary; digits < 2^32 leave 31 bits of headroom per limb.
"""
from . import OUT_DIGITS, SET_WORDS  # noqa: F401  (re-exported constants)


def shard_range(n, rank, world):
    """[first, last) of rank's contiguous shard; boundaries are even so every shard stays 16-byte aligned."""
    def cut(r):
        c = (n * r) // world
        return n if r == world else (c & ~1)
    return cut(rank), cut(rank + 1)


def allreduce_record(rec, group=None, force=False):
    """In-place int64 SUM all-reduce of the digit set (words 48..119) of a record tensor.

    Works on any device/back-end pair torch.distributed supports (nccl=RCCL on GPUs, gloo on CPU)."""
    import torch.distributed as dist
    payload = rec[OUT_DIGITS:OUT_DIGITS + SET_WORDS]
    if dist.is_available() and dist.is_initialized() and (force or dist.get_world_size(group) > 1):
        dist.all_reduce(payload, op=dist.ReduceOp.SUM, group=group)
    return rec


def _finish(rec, group):
    from . import finalize_dev
    allreduce_record(rec, group)
    # carry-propagate + round the summed digits; in place (the kernel reads everything before it writes)
    return finalize_dev(rec[OUT_DIGITS:OUT_DIGITS + SET_WORDS], out=rec)


def exsum_allreduce(x_local, fpe=8, early_exit=True, group=None, out=None):
    """Exact sum of the concatenation of every rank's ``x_local`` (CUDA float64); record tensor on device."""
    from . import exsum_dev
    rec = exsum_dev(x_local, fpe=fpe, early_exit=early_exit, out=out)
    return _finish(rec, group)


def exdot_allreduce(x_local, y_local, fpe=8, early_exit=True, group=None, out=None):
    from . import exdot_dev
    rec = exdot_dev(x_local, y_local, fpe=fpe, early_exit=early_exit, out=out)
    return _finish(rec, group)


# ---------------------------------------------------------------------------------------------
# ExGEMV / ExGEMM: outputs are independent, so the path shards by ROWS with no data-path collective
# (SURVEY 8e): every rank owns a contiguous block of rows of A (and y resp. C) and the whole of x resp. B.
# ---------------------------------------------------------------------------------------------
def row_block(m, rank, world):
    """[first, last) rows of this rank; even boundaries keep the 16-byte alignment of column-major A blocks."""
    return shard_range(m, rank, world)


def exgemv_rows(trans, m_local, n, alpha, a_local, lda_local, x, beta, y_local, fpe=8, early_exit=True):
    """y_local := alpha*op(A_local)*x + beta*y_local for this rank's row block ('N': rows of A; column-major
    A_local with leading dimension lda_local).  For trans == 'T' the reduction runs over the sharded dimension,
    so callers shard the OUTPUT instead (columns of A) -- also independent, also no collective."""
    from . import exgemv_dev
    return exgemv_dev(trans, m_local, n, alpha, a_local, lda_local, x, beta, y_local, fpe, early_exit)


def exgemm_rows(m_local, n, k, alpha, a_local, b, beta, c_local, fpe=8, early_exit=True):
    """C_local := alpha*A_local*B + beta*C_local (row-major) for this rank's rows of A and C; B is replicated."""
    from . import exgemm_dev
    return exgemm_dev("N", "N", m_local, n, k, alpha, a_local, k, b, n, beta, c_local, n, fpe, early_exit)
