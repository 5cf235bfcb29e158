"""exblas_amd -- MI355X-native (gfx950, HIP) ExBLAS hot path: exact, reproducible sum / dot / gemv / gemm.

This package is the host-side mirror of the reference's public API (include/blas1.hpp:48,74,
blas2.hpp:95, blas3.hpp:56 of nikolovjovan/exblas) on top of the C ABI of ``lib/libexblas.so``
(``include/exblas_hip.h``).  PyTorch is used only as plumbing (device memory, streams,
``torch.distributed``); every reduction runs in the hand-written HIP kernels under ``csrc/``.

There is NO CPU fallback: importing works without a GPU (so the ABI can be inspected), but any
compute call without a HIP device fails loudly.
"""
import ctypes as C
import os

import numpy as np

from . import build as _build

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_PKG, "lib", "libexblas.so")

OUT_WORDS, OUT_EXACT, OUT_REFMODE, OUT_FLAGS, OUT_CANON, OUT_DIGITS = 128, 0, 1, 2, 4, 48
NDIGITS, NCANON, SET_WORDS = 68, 41, 72
GEN_KINDS = {"naive": 0, "fpuniform": 1, "lognormal": 2, "ill_cond": 3, "cancel": 4, "fpuniform_signed": 5}

# every symbol include/exblas_hip.h declares (checked by tests/test_abi.py)
C_ABI_SYMBOLS = [
    "exblas_hip_init", "exblas_hip_device_count", "exblas_hip_version", "exblas_set_round_mode",
    "exblas_get_round_mode", "exblas_exsum_dev", "exblas_exdot_dev", "exblas_finalize_dev", "exblas_exgemv_dev",
    "exblas_exgemm_dev", "exblas_gen_dev", "exblas_stream_read_dev", "exblas_exsum", "exblas_exdot",
    "exblas_exgemv", "exblas_exgemm", "exblas_exsum_record", "exblas_exdot_record",
    "exblas_exsum_accumulate_dev", "exblas_exdot_accumulate_dev", "exblas_finish_dev", "exblas_set_tuning",
    "exblas_set_gemm_path", "exblas_last_gemm_slices", "exblas_exsum_segmented_dev",
    "exblas_set_accumulator_slot", "exblas_set_launch_events", "exblas_stream_read2_dev", "exblas_extrsv_dev", "exblas_extrsv",
    "exblas_extrsv_last_slow_rows", "exblas_reserve_workspace", "exblas_release_retired_workspaces", "exblas_release_workspace",
    "exblas_comm_unique_id", "exblas_comm_init_rccl", "exblas_comm_adopt_rccl", "exblas_comm_init_host",
    "exblas_comm_destroy", "exblas_comm_rank", "exblas_comm_size", "exblas_shard_range",
    "exblas_exsum_allreduce_dev", "exblas_exdot_allreduce_dev", "exblas_allreduce_finish_dev",
    "exblas_exgemv_sharded_dev", "exblas_exgemm_sharded_dev", "exblas_last_gemm_info", "exblas_set_gemm_max_slices",
    "exblas_set_gemm_max_moduli", "exblas_crt_selftest",
    "exblas_set_host_devices", "exblas_workspace_bytes",
    "exblas_exsum_allreduce_pipelined_dev", "exblas_exdot_allreduce_pipelined_dev", "exblas_pipeline_drain_dev",
    "exblas_ctx_create", "exblas_ctx_destroy", "exblas_exsum_ctx", "exblas_exdot_ctx", "exblas_exsum_accumulate_ctx",
    "exblas_exdot_accumulate_ctx", "exblas_finish_ctx", "exblas_exgemv_ctx", "exblas_extrsv_ctx", "exblas_exgemm_ctx",
    "exblas_reserve_workspace_ctx", "exblas_workspace_bytes_ctx", "exblas_last_gemm_info_ctx",
]

# host-transport callback types of include/exblas_hip.h
HOST_ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_int64), C.c_int64)
HOST_BCAST_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_int)
HOST_ALLGATHERV_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_int64))
UNIQUE_ID_BYTES = 128
COMM_ERROR = -2

_lib = None


def load_library():
    """Load (building first if sources are newer) lib/libexblas.so.  Raises if it cannot be had."""
    global _lib, LIB_PATH
    if _lib is not None:
        return _lib
    alt = os.environ.get("EXBLAS_AMD_LIB")  # A/B of an alternative build (tools/): load exactly this file
    if alt:
        LIB_PATH = os.path.abspath(alt)
        _build.stale = lambda: False
    if not os.path.exists(LIB_PATH) or _build.stale():
        have_hipcc = bool(_build.hipcc()) and os.path.exists(_build.hipcc())
        if have_hipcc:
            # a failed rebuild is fatal: a library older than its sources must never be loaded in its place
            try:
                _build.build()
            except Exception as exc:  # noqa: BLE001
                raise ImportError(f"exblas_amd: building {LIB_PATH} failed ({exc}); refusing to load a stale "
                                  "library") from exc
        elif os.path.exists(LIB_PATH):
            raise ImportError(f"exblas_amd: {LIB_PATH} is older than its sources and hipcc is not available to "
                              "rebuild it")
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"exblas_amd: {LIB_PATH} is missing and hipcc is not available; "
                          "there is no CPU fallback")
    # PyTorch wheels bundle their own libamdhip64/libhsa-runtime64.  Two HIP runtimes in one process do
    # not share devices, streams or pointers (and the second one to initialise finds no device), so when
    # torch is importable it must be loaded FIRST: libexblas.so's NEEDED libamdhip64.so.7 then resolves to
    # the runtime torch already mapped.  Stand-alone C/C++ users simply get /opt/rocm's runtime.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(LIB_PATH)
    vp, i64, i32, dbl = C.c_void_p, C.c_int64, C.c_int, C.c_double
    L.exblas_hip_init.argtypes = [i32]
    L.exblas_hip_version.restype = C.c_char_p
    L.exblas_set_round_mode.argtypes = [i32]
    L.exblas_set_tuning.argtypes = [i32, i32, i32]
    L.exblas_set_accumulator_slot.argtypes = [i32]
    L.exblas_set_launch_events.argtypes = [vp, vp]
    L.exblas_set_gemm_path.argtypes = [i32]
    L.exblas_set_gemm_path.restype = None
    L.exblas_exsum_dev.argtypes = [vp, i64, i64, i32, i32, vp, vp]
    L.exblas_exdot_dev.argtypes = [vp, i64, vp, i64, i64, i32, i32, vp, vp]
    L.exblas_finalize_dev.argtypes = [vp, i32, C.c_uint32, vp, vp]
    L.exblas_exsum_accumulate_dev.argtypes = [vp, i64, i64, i32, i32, vp]
    L.exblas_exdot_accumulate_dev.argtypes = [vp, i64, vp, i64, i64, i32, i32, vp]
    L.exblas_finish_dev.argtypes = [vp, vp]
    L.exblas_exsum_segmented_dev.argtypes = [vp, vp, i64, i32, i32, vp, vp]
    L.exblas_exgemv_dev.argtypes = [C.c_char, i32, i32, dbl, vp, i32, vp, i32, dbl, vp, i32, i32, i32, vp]
    L.exblas_exgemm_dev.argtypes = [C.c_char, C.c_char, i32, i32, i32, dbl, vp, i32, vp, i32, dbl, vp, i32, i32,
                                    i32, vp]
    L.exblas_extrsv_dev.argtypes = [C.c_char, C.c_char, C.c_char, i32, vp, i32, vp, i32, i32, i32, vp]
    L.exblas_extrsv.argtypes = [C.c_char, C.c_char, C.c_char, i32, vp, i32, i32, vp, i32, i32, i32, i32]
    L.exblas_gen_dev.argtypes = [i32, C.c_uint64, i64, i64, i64, dbl, dbl, vp, vp]
    L.exblas_stream_read_dev.argtypes = [vp, i64, vp, vp]
    L.exblas_stream_read2_dev.argtypes = [vp, vp, i64, i32, vp, vp]
    L.exblas_exsum.restype = dbl
    L.exblas_exsum.argtypes = [i32, vp, i32, i32, i32, i32]
    L.exblas_exdot.restype = dbl
    L.exblas_exdot.argtypes = [i32, vp, i32, i32, vp, i32, i32, i32, i32]
    L.exblas_exgemv.argtypes = [C.c_char, i32, i32, dbl, vp, i32, i32, vp, i32, i32, dbl, vp, i32, i32, i32, i32]
    L.exblas_exgemm.argtypes = [C.c_char, C.c_char, i32, i32, i32, dbl, vp, i32, vp, i32, dbl, vp, i32, i32, i32]
    L.exblas_reserve_workspace.argtypes = [C.c_size_t]
    L.exblas_workspace_bytes.restype = C.c_size_t
    L.exblas_ctx_create.argtypes = [C.POINTER(vp)]
    L.exblas_ctx_destroy.argtypes = [vp]
    L.exblas_exsum_ctx.argtypes = [vp] + L.exblas_exsum_dev.argtypes
    L.exblas_exdot_ctx.argtypes = [vp] + L.exblas_exdot_dev.argtypes
    L.exblas_exsum_accumulate_ctx.argtypes = [vp] + L.exblas_exsum_accumulate_dev.argtypes
    L.exblas_exdot_accumulate_ctx.argtypes = [vp] + L.exblas_exdot_accumulate_dev.argtypes
    L.exblas_finish_ctx.argtypes = [vp] + L.exblas_finish_dev.argtypes
    L.exblas_exgemv_ctx.argtypes = [vp] + L.exblas_exgemv_dev.argtypes
    L.exblas_extrsv_ctx.argtypes = [vp] + L.exblas_extrsv_dev.argtypes
    L.exblas_exgemm_ctx.argtypes = [vp] + L.exblas_exgemm_dev.argtypes
    L.exblas_reserve_workspace_ctx.argtypes = [vp, C.c_size_t]
    L.exblas_workspace_bytes_ctx.argtypes = [vp]
    L.exblas_workspace_bytes_ctx.restype = C.c_size_t
    L.exblas_last_gemm_info_ctx.argtypes = [vp, C.POINTER(C.c_int)]
    L.exblas_set_host_devices.argtypes = [i32, C.POINTER(C.c_int)]
    L.exblas_last_gemm_info.argtypes = [C.POINTER(C.c_int)]
    L.exblas_set_gemm_max_slices.argtypes = [i32]
    L.exblas_set_gemm_max_slices.restype = None
    L.exblas_set_gemm_max_moduli.argtypes = [i32]
    L.exblas_set_gemm_max_moduli.restype = None
    L.exblas_crt_selftest.argtypes = [i32, C.c_uint]
    L.exblas_crt_selftest.restype = i32
    L.exblas_comm_unique_id.argtypes = [vp]
    L.exblas_comm_init_rccl.argtypes = [C.POINTER(vp), i32, i32, vp]
    L.exblas_comm_adopt_rccl.argtypes = [C.POINTER(vp), vp, i32, i32]
    L.exblas_comm_init_host.argtypes = [C.POINTER(vp), i32, i32, HOST_ALLREDUCE_FN, HOST_BCAST_FN, HOST_ALLGATHERV_FN,
                                        vp]
    L.exblas_comm_destroy.argtypes = [vp]
    L.exblas_comm_rank.argtypes = [vp]
    L.exblas_comm_size.argtypes = [vp]
    L.exblas_shard_range.argtypes = [i64, i32, i32, C.POINTER(i64), C.POINTER(i64)]
    L.exblas_shard_range.restype = None
    L.exblas_exsum_allreduce_dev.argtypes = [vp, vp, i64, i64, i32, i32, vp, vp]
    L.exblas_exdot_allreduce_dev.argtypes = [vp, vp, i64, vp, i64, i64, i32, i32, vp, vp]
    L.exblas_allreduce_finish_dev.argtypes = [vp, vp, vp]
    L.exblas_exsum_allreduce_pipelined_dev.argtypes = [vp, vp, i64, i64, i32, i32, vp, vp, vp, vp]
    L.exblas_exdot_allreduce_pipelined_dev.argtypes = [vp, vp, i64, vp, i64, i64, i32, i32, vp, vp, vp, vp]
    L.exblas_pipeline_drain_dev.argtypes = [vp, vp]
    L.exblas_exgemv_sharded_dev.argtypes = [vp, C.c_char, i32, i32, dbl, vp, i32, vp, i32, i32, dbl, vp, i32, i32, i32,
                                            i32, vp]
    L.exblas_exgemm_sharded_dev.argtypes = [vp, C.c_char, C.c_char, i32, i32, i32, dbl, vp, i32, vp, i32, i32, dbl,
                                            vp, i32, i32, i32, i32, vp]
    L.exblas_exsum_record.argtypes = [i32, vp, i32, i32, i32, i32, vp]
    L.exblas_exdot_record.argtypes = [i32, vp, i32, i32, vp, i32, i32, i32, i32, vp]
    _lib = L
    return L


def _torch():
    import torch
    return torch


def _require_gpu():
    torch = _torch()
    if not torch.cuda.is_available():
        raise RuntimeError("exblas_amd: no HIP device visible; the MI355X path has no CPU fallback")
    return torch


def _check(rc, what):
    if rc != 0:
        raise RuntimeError(f"exblas_amd: {what} failed with HIP error {rc}")


def _stream_ptr(torch):
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class Record:
    """Decoded result record of one reduction (see include/exblas_hip.h)."""

    def __init__(self, words):
        w = np.asarray(words, dtype=np.int64)
        self.words = w
        self.exact = float(w[OUT_EXACT:OUT_EXACT + 1].view(np.float64)[0])
        self.refmode = float(w[OUT_REFMODE:OUT_REFMODE + 1].view(np.float64)[0])
        self.flags = int(w[OUT_FLAGS])
        self.canon = w[OUT_CANON:OUT_CANON + NCANON].copy()
        self.digits = w[OUT_DIGITS:OUT_DIGITS + NDIGITS].copy()

    def value(self, mode=None):
        if mode is None:
            mode = load_library().exblas_get_round_mode()
        return self.refmode if mode else self.exact


# ---------------------------------------------------------------------------------------------
# device-resident API (torch CUDA tensors): what bench.py and the multi-GPU path use
# ---------------------------------------------------------------------------------------------
def new_record_buffer():
    torch = _require_gpu()
    return torch.zeros(OUT_WORDS, dtype=torch.int64, device="cuda")


def exsum_dev(x, fpe=8, early_exit=True, inca=1, n=None, out=None):
    """ExSUM of a CUDA float64 tensor, stream-ordered; returns the int64 record tensor (on device)."""
    torch = _require_gpu()
    assert x.is_cuda and x.dtype == torch.float64
    if n is None:
        n = (x.numel() + inca - 1) // inca
    if out is None:
        out = new_record_buffer()
    _check(load_library().exblas_exsum_dev(C.c_void_p(x.data_ptr()), n, inca, fpe, int(early_exit),
                                           _stream_ptr(torch), C.c_void_p(out.data_ptr())), "exsum_dev")
    return out


def exdot_dev(x, y, fpe=8, early_exit=True, incx=1, incy=1, n=None, out=None):
    torch = _require_gpu()
    assert x.is_cuda and y.is_cuda and x.dtype == torch.float64 and y.dtype == torch.float64
    if n is None:
        n = (x.numel() + incx - 1) // incx
    if out is None:
        out = new_record_buffer()
    _check(load_library().exblas_exdot_dev(C.c_void_p(x.data_ptr()), incx, C.c_void_p(y.data_ptr()), incy, n, fpe,
                                           int(early_exit), _stream_ptr(torch), C.c_void_p(out.data_ptr())),
           "exdot_dev")
    return out


def exsum_accumulate_dev(x, fpe=8, early_exit=True, inca=1, n=None):
    """Phase 1 only: stream x into the context accumulators (several calls fold into one exact sum)."""
    torch = _require_gpu()
    if n is None:
        n = (x.numel() + inca - 1) // inca
    _check(load_library().exblas_exsum_accumulate_dev(C.c_void_p(x.data_ptr()), n, inca, fpe, int(early_exit),
                                                      _stream_ptr(torch)), "exsum_accumulate_dev")


def exdot_accumulate_dev(x, y, fpe=8, early_exit=True, incx=1, incy=1, n=None):
    torch = _require_gpu()
    if n is None:
        n = (x.numel() + incx - 1) // incx
    _check(load_library().exblas_exdot_accumulate_dev(C.c_void_p(x.data_ptr()), incx, C.c_void_p(y.data_ptr()), incy,
                                                      n, fpe, int(early_exit), _stream_ptr(torch)),
           "exdot_accumulate_dev")


def set_launch_events(ev_start, ev_stop):
    """The next exsum / exdot accumulate call attaches these torch events (already recorded once, so that they own a
    handle; either may be None) to its streaming kernel's dispatch packet: kernel start / stop timestamps, no packets."""
    h = lambda e: C.c_void_p(e.cuda_event) if e is not None else None  # noqa: E731
    _check(load_library().exblas_set_launch_events(h(ev_start), h(ev_stop)), "set_launch_events")


def set_accumulator_slot(slot):
    """Select which of the context's two accumulator sets the next accumulate/finish calls use (pipelining)."""
    _check(load_library().exblas_set_accumulator_slot(int(slot)), "set_accumulator_slot")


def finish_dev(out=None):
    """Phase 2: carry-propagate + round the context accumulators into a record; zeroes them."""
    torch = _require_gpu()
    if out is None:
        out = new_record_buffer()
    _check(load_library().exblas_finish_dev(_stream_ptr(torch), C.c_void_p(out.data_ptr())), "finish_dev")
    return out


def exsum_segmented_dev(values, offsets, fpe=8, early_exit=True, out=None):
    """out[s] = exact sum of values[offsets[s]:offsets[s+1]] (CUDA float64 / int64 tensors), one launch."""
    torch = _require_gpu()
    assert values.is_cuda and values.dtype == torch.float64 and offsets.is_cuda and offsets.dtype == torch.int64
    nseg = offsets.numel() - 1
    if out is None:
        out = torch.empty(max(nseg, 0), dtype=torch.float64, device="cuda")
    _check(load_library().exblas_exsum_segmented_dev(C.c_void_p(values.data_ptr()), C.c_void_p(offsets.data_ptr()),
                                                     nseg, fpe, int(early_exit), _stream_ptr(torch),
                                                     C.c_void_p(out.data_ptr())), "exsum_segmented_dev")
    return out


def finalize_dev(digit_sets, flags_or=0, out=None):
    """Sum [nsets, 72] int64 digit sets (record words 48..119), carry-propagate once, round."""
    torch = _require_gpu()
    assert digit_sets.is_cuda and digit_sets.dtype == torch.int64 and digit_sets.is_contiguous()
    nsets = digit_sets.numel() // SET_WORDS
    if out is None:
        out = new_record_buffer()
    _check(load_library().exblas_finalize_dev(C.c_void_p(digit_sets.data_ptr()), nsets, flags_or, _stream_ptr(torch),
                                              C.c_void_p(out.data_ptr())), "finalize_dev")
    return out


def exgemv_dev(trans, m, n, alpha, a, lda, x, beta, y, fpe=0, early_exit=False, incx=1, incy=1):
    torch = _require_gpu()
    _check(load_library().exblas_exgemv_dev(trans.encode(), m, n, alpha, C.c_void_p(a.data_ptr()), lda,
                                            C.c_void_p(x.data_ptr()), incx, beta, C.c_void_p(y.data_ptr()), incy,
                                            fpe, int(early_exit), _stream_ptr(torch)), "exgemv_dev")
    return y


def extrsv_dev(uplo, trans, diag, n, a, lda, x, fpe=0, early_exit=False, incx=1):
    """x := A^-1 x (or A^-T x) in place on device tensors; returns 0, or -1 for the unsupported fpe >= 9."""
    torch = _require_gpu()
    rc = load_library().exblas_extrsv_dev(uplo.encode(), trans.encode(), diag.encode(), n, C.c_void_p(a.data_ptr()),
                                          lda, C.c_void_p(x.data_ptr()), incx, fpe, int(early_exit),
                                          _stream_ptr(torch))
    if rc != -1:
        _check(rc, "extrsv_dev")
    return rc


def exgemm_dev(transa, transb, m, n, k, alpha, a, lda, b, ldb, beta, c, ldc, fpe=0, early_exit=False):
    torch = _require_gpu()
    _check(load_library().exblas_exgemm_dev(transa.encode(), transb.encode(), m, n, k, alpha,
                                            C.c_void_p(a.data_ptr()), lda, C.c_void_p(b.data_ptr()), ldb, beta,
                                            C.c_void_p(c.data_ptr()), ldc, fpe, int(early_exit), _stream_ptr(torch)),
           "exgemm_dev")
    return c


class Context:
    """Owner of an ``exblas_ctx_t *``: private accumulators, flags and workspace on the current device, so that work
    enqueued through different contexts (on different streams) needs no ordering.  Methods mirror the ``*_dev``
    functions; tensors are CUDA float64 / int64 on the context's device, calls go to the CURRENT torch stream."""

    def __init__(self):
        _require_gpu()
        h = C.c_void_p()
        _check(load_library().exblas_ctx_create(C.byref(h)), "ctx_create")
        self.handle = h

    def destroy(self):
        if self.handle is not None:
            load_library().exblas_ctx_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.destroy()
        except Exception:  # noqa: BLE001
            pass

    def exsum(self, x, fpe=8, early_exit=True, inca=1, n=None, out=None):
        torch = _require_gpu()
        if n is None:
            n = (x.numel() + inca - 1) // inca
        if out is None:
            out = new_record_buffer()
        _check(load_library().exblas_exsum_ctx(self.handle, C.c_void_p(x.data_ptr()), n, inca, fpe, int(early_exit),
                                               _stream_ptr(torch), C.c_void_p(out.data_ptr())), "exsum_ctx")
        return out

    def exdot(self, x, y, fpe=8, early_exit=True, out=None):
        torch = _require_gpu()
        if out is None:
            out = new_record_buffer()
        _check(load_library().exblas_exdot_ctx(self.handle, C.c_void_p(x.data_ptr()), 1, C.c_void_p(y.data_ptr()), 1,
                                               x.numel(), fpe, int(early_exit), _stream_ptr(torch),
                                               C.c_void_p(out.data_ptr())), "exdot_ctx")
        return out

    def exsum_accumulate(self, x, fpe=8, early_exit=True):
        torch = _require_gpu()
        _check(load_library().exblas_exsum_accumulate_ctx(self.handle, C.c_void_p(x.data_ptr()), x.numel(), 1, fpe,
                                                          int(early_exit), _stream_ptr(torch)), "exsum_accumulate_ctx")

    def finish(self, out=None):
        torch = _require_gpu()
        if out is None:
            out = new_record_buffer()
        _check(load_library().exblas_finish_ctx(self.handle, _stream_ptr(torch), C.c_void_p(out.data_ptr())),
               "finish_ctx")
        return out

    def exgemv(self, trans, m, n, alpha, a, lda, x, beta, y, fpe=0, early_exit=False, incx=1, incy=1):
        torch = _require_gpu()
        _check(load_library().exblas_exgemv_ctx(self.handle, trans.encode(), m, n, alpha, C.c_void_p(a.data_ptr()), lda,
                                                C.c_void_p(x.data_ptr()), incx, beta, C.c_void_p(y.data_ptr()), incy,
                                                fpe, int(early_exit), _stream_ptr(torch)), "exgemv_ctx")
        return y

    def exgemm(self, transa, transb, m, n, k, alpha, a, lda, b, ldb, beta, c, ldc, fpe=0, early_exit=False):
        torch = _require_gpu()
        _check(load_library().exblas_exgemm_ctx(self.handle, transa.encode(), transb.encode(), m, n, k, alpha,
                                                C.c_void_p(a.data_ptr()), lda, C.c_void_p(b.data_ptr()), ldb, beta,
                                                C.c_void_p(c.data_ptr()), ldc, fpe, int(early_exit),
                                                _stream_ptr(torch)), "exgemm_ctx")
        return c

    def extrsv(self, uplo, trans, diag, n, a, lda, x, fpe=0, early_exit=False, incx=1):
        torch = _require_gpu()
        rc = load_library().exblas_extrsv_ctx(self.handle, uplo.encode(), trans.encode(), diag.encode(), n,
                                              C.c_void_p(a.data_ptr()), lda, C.c_void_p(x.data_ptr()), incx, fpe,
                                              int(early_exit), _stream_ptr(torch))
        if rc != -1:
            _check(rc, "extrsv_ctx")
        return rc

    def workspace_bytes(self):
        return load_library().exblas_workspace_bytes_ctx(self.handle)


def gen_dev(kind, n, seed=1, p0=0.0, p1=0.0, first=0, count=None, n_total=None, out=None):
    """Counter-based generator on the GPU; bit-identical to oracle.pyoracle.gen()."""
    torch = _require_gpu()
    k = GEN_KINDS[kind] if isinstance(kind, str) else int(kind)
    if count is None:
        count = n
    if n_total is None:
        n_total = n
    if out is None:
        out = torch.empty(count, dtype=torch.float64, device="cuda")
    _check(load_library().exblas_gen_dev(k, seed, first, count, n_total, p0, p1, C.c_void_p(out.data_ptr()),
                                         _stream_ptr(torch)), "gen_dev")
    return out


def stream_read_dev(x, sink=None):
    torch = _require_gpu()
    if sink is None:
        sink = torch.zeros(1, dtype=torch.float64, device="cuda")
    _check(load_library().exblas_stream_read_dev(C.c_void_p(x.data_ptr()), x.numel(), _stream_ptr(torch),
                                                 C.c_void_p(sink.data_ptr())), "stream_read_dev")
    return sink


def read_record(rec_tensor):
    """Synchronising D2H read of a record tensor."""
    return Record(rec_tensor.cpu().numpy())


# ---------------------------------------------------------------------------------------------
# reference-style API (host arrays in, doubles out) -- same argument order as the C++ headers
# ---------------------------------------------------------------------------------------------
def _host(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def exsum(Ng, ag, inca, offset, fpe, early_exit=False, parallel=True):
    """double exsum(Ng, ag, inca, offset, fpe, early_exit, parallel) -- include/blas1.hpp:48."""
    _require_gpu()
    a = _host(ag)
    return load_library().exblas_exsum(int(Ng), C.c_void_p(a.ctypes.data), int(inca), int(offset), int(fpe),
                                       int(bool(early_exit)))


def exdot(Ng, ag, inca, offseta, bg, incb, offsetb, fpe, early_exit=False):
    """double exdot(...) -- include/blas1.hpp:74."""
    _require_gpu()
    a, b = _host(ag), _host(bg)
    return load_library().exblas_exdot(int(Ng), C.c_void_p(a.ctypes.data), int(inca), int(offseta),
                                       C.c_void_p(b.ctypes.data), int(incb), int(offsetb), int(fpe),
                                       int(bool(early_exit)))


def exsum_record(Ng, ag, inca, offset, fpe, early_exit=False):
    _require_gpu()
    a = _host(ag)
    out = np.zeros(OUT_WORDS, dtype=np.int64)
    load_library().exblas_exsum_record(int(Ng), C.c_void_p(a.ctypes.data), int(inca), int(offset), int(fpe),
                                       int(bool(early_exit)), C.c_void_p(out.ctypes.data))
    return Record(out)


def exdot_record(Ng, ag, inca, offseta, bg, incb, offsetb, fpe, early_exit=False):
    _require_gpu()
    a, b = _host(ag), _host(bg)
    out = np.zeros(OUT_WORDS, dtype=np.int64)
    load_library().exblas_exdot_record(int(Ng), C.c_void_p(a.ctypes.data), int(inca), int(offseta),
                                       C.c_void_p(b.ctypes.data), int(incb), int(offsetb), int(fpe),
                                       int(bool(early_exit)), C.c_void_p(out.ctypes.data))
    return Record(out)


def exgemv(transa, m, n, alpha, a, lda, offseta, x, incx, offsetx, beta, y, incy, offsety, fpe, early_exit=False):
    """int exgemv(...) -- include/blas2.hpp:95; y (numpy float64) is updated in place."""
    _require_gpu()
    a_, x_ = _host(a), _host(x)
    assert isinstance(y, np.ndarray) and y.dtype == np.float64 and y.flags.c_contiguous
    return load_library().exblas_exgemv(transa.encode(), m, n, alpha, C.c_void_p(a_.ctypes.data), lda, offseta,
                                        C.c_void_p(x_.ctypes.data), incx, offsetx, beta, C.c_void_p(y.ctypes.data),
                                        incy, offsety, fpe, int(bool(early_exit)))


def extrsv(uplo, transa, diag, n, a, lda, offseta, x, incx, offsetx, fpe, early_exit=False):
    """int extrsv(...) -- include/blas2.hpp:57; x (numpy float64) holds b on entry and the solution on return."""
    _require_gpu()
    a_ = _host(a)
    assert isinstance(x, np.ndarray) and x.dtype == np.float64 and x.flags.c_contiguous
    return load_library().exblas_extrsv(uplo.encode(), transa.encode(), diag.encode(), n, C.c_void_p(a_.ctypes.data),
                                        lda, offseta, C.c_void_p(x.ctypes.data), incx, offsetx, fpe,
                                        int(bool(early_exit)))


def exgemm(transa, transb, m, n, k, alpha, a, lda, b, ldb, beta, c, ldc, fpe, early_exit=False):
    """int exgemm(...) -- include/blas3.hpp:56; c (numpy float64) is updated in place."""
    _require_gpu()
    a_, b_ = _host(a), _host(b)
    assert isinstance(c, np.ndarray) and c.dtype == np.float64 and c.flags.c_contiguous
    return load_library().exblas_exgemm(transa.encode(), transb.encode(), m, n, k, alpha, C.c_void_p(a_.ctypes.data),
                                        lda, C.c_void_p(b_.ctypes.data), ldb, beta, C.c_void_p(c.ctypes.data), ldc,
                                        fpe, int(bool(early_exit)))


from .dist import (Comm, exsum_allreduce, exdot_allreduce, allreduce_finish, allreduce_record,  # noqa: E402,F401
                   shard_range, row_block, exgemv_sharded, exgemm_sharded, exsum_allreduce_pipelined,
                   exdot_allreduce_pipelined, pipeline_drain)
