"""ctypes binding of libsemigcn_hip.so (include/semigcn.h) for PyTorch-ROCm tensors.

This is the only seam between the Python host code and the HIP kernels.  There
is no CPU implementation behind it: if the library is missing, cannot be loaded,
or a tensor is not on a HIP device, the call raises -- it never falls back.
"""
from __future__ import annotations

import ctypes
import functools
from array import array as _array
import os
from ctypes import POINTER, byref, c_char_p, c_float, c_int, c_int32, c_int64, c_void_p
from typing import Optional

import torch  # imported first so that libamdhip64.so.7 resolves to the copy torch already loaded

_LIB_NAME = "libsemigcn_hip.so"
_LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), _LIB_NAME)

SG_F32, SG_BF16 = 0, 1
_DTYPES = {torch.float32: SG_F32, torch.bfloat16: SG_BF16}


class SemigcnLibraryError(RuntimeError):
    """libsemigcn_hip.so is absent / unloadable, or a call into it failed."""


class sg_graph_info(ctypes.Structure):
    _fields_ = [("V_dst", c_int64), ("V_src", c_int64), ("nnz", c_int64),
                ("symmetric", c_int32), ("max_degree", c_int32)]


class sg_block(ctypes.Structure):
    """Mirror of ``struct sg_block`` (include/semigcn.h): one [ChebConv -> pool? -> BatchNorm1d -> LeakyReLU] block."""
    _fields_ = [("graph", c_void_p), ("pool", c_void_p),
                ("pool_mode", c_int32), ("dtype", c_int32), ("K", c_int32), ("order", c_int32), ("training", c_int32),
                ("refresh_weights", c_int32), ("need_dx", c_int32), ("reserved_", c_int32),
                ("V", c_int64), ("V_out", c_int64), ("Cin", c_int64), ("Cout", c_int64),
                ("momentum", c_float), ("eps", c_float), ("slope", c_float), ("reserved2_", c_float),
                ("W", c_void_p * 3), ("bias", c_void_p), ("gamma", c_void_p), ("beta", c_void_p),
                ("running_mean", c_void_p), ("running_var", c_void_p), ("batches_tracked", c_void_p),
                ("wpack", c_void_p), ("wpack_t", c_void_p), ("wpack32", c_void_p), ("wpack32_t", c_void_p), ("bias_k", c_void_p),
                ("wsplit", c_void_p), ("wsplit_t", c_void_p),
                ("X", c_void_p), ("ldx", c_int64), ("T", c_void_p), ("ldt", c_int64), ("H", c_void_p), ("stats", c_void_p),
                ("Y", c_void_p), ("ldy", c_int64),
                ("dY", c_void_p), ("lddy", c_int64), ("dX", c_void_p), ("lddx", c_int64), ("dW", c_void_p), ("dvec", c_void_p),
                ("acc_W", c_void_p * 3), ("acc_bias", c_void_p), ("acc_gamma", c_void_p), ("acc_beta", c_void_p),
                ("ws", c_void_p), ("ws_bytes", c_int64),
                ("phase", c_int32), ("world", c_int32), ("graph_wide", c_void_p), ("V_ext", c_int64), ("ldh", c_int64),
                ("local", c_void_p), ("stats_rows", c_void_p), ("gathered", c_void_p), ("gathered_ready", c_int32),
                ("reserved3_", c_int32), ("count", c_void_p), ("send_index", c_void_p), ("n_send", c_int64), ("send", c_void_p),
                ("recv", c_void_p), ("G", c_void_p)]


PHASE_CONV, PHASE_BN, PHASE_BWD_REDUCE, PHASE_BWD_A, PHASE_BWD_B = 1, 2, 4, 8, 16


class sg_part_step(ctypes.Structure):
    """Mirror of ``struct sg_part_step`` (include/semigcn.h): one entry of a rank's schedule for ``sg_part_run``."""
    _fields_ = [("kind", c_int32), ("reserved_", c_int32), ("blocks", POINTER(sg_block)), ("n", c_int64),
                ("send", c_void_p), ("recv", c_void_p)]


STEP_BLOCKS, STEP_EXCHANGE, STEP_ALL_REDUCE, STEP_ALL_GATHER = 0, 1, 2, 3


class sg_trace_record(ctypes.Structure):
    _fields_ = [("kind", c_int32), ("dtype", c_int32), ("engine", c_int32), ("reserved_", c_int32),
                ("a", c_int64), ("b", c_int64), ("c", c_int64), ("ms", c_float), ("reserved2_", c_float)]


# name -> (restype, argtypes); mirrors include/semigcn.h one to one
_SIGNATURES = {
    "sg_last_error": (c_char_p, []),
    "sg_abi_version": (c_int, []),
    "sg_device_count": (c_int, []),
    "sg_graph_create": (c_int, [c_void_p, c_int64, c_int64, c_void_p, POINTER(c_void_p)]),
    "sg_graph_create_rect": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int64, c_void_p, c_void_p,
                                     POINTER(c_void_p)]),
    "sg_graph_create_rows": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_void_p,
                                     POINTER(c_void_p)]),
    "sg_graph_destroy": (c_int, [c_void_p]),
    "sg_graph_query": (c_int, [c_void_p, POINTER(sg_graph_info)]),
    "sg_graph_is_reordered": (c_int, [c_void_p]),
    "sg_graph_export": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "sg_spmm": (c_int, [c_void_p, c_int, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64,
                        c_void_p, c_int64, c_int64, c_int, c_float, c_float, c_float, c_void_p]),
    "sg_graph_prepare": (c_int, [c_void_p, c_int64, c_int, c_void_p]),
    "sg_pool_create": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int64, c_void_p, POINTER(c_void_p)]),
    "sg_pool_destroy": (c_int, [c_void_p]),
    "sg_pool_mean": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_int64, c_int, c_void_p]),
    "sg_pool_mean_bwd": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_int64, c_int, c_void_p]),
    "sg_unpool": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_int64, c_int, c_void_p]),
    "sg_unpool_bwd": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_int64, c_int, c_void_p]),
    "sg_tuning_set": (c_int, [c_int, c_int]),
    "sg_mesh_loss_blocks": (c_int64, [c_int64, c_int64]),
    "sg_mesh_loss_fwd": (c_int, [c_void_p] * 6 + [c_int64, c_int64, c_void_p, c_void_p]),
    "sg_mesh_loss_bwd": (c_int, [c_void_p] * 7 + [c_int64, c_int64, c_int64, c_void_p, c_void_p]),
    "sg_mesh_loss_bwd_det": (c_int, [c_void_p] * 7 + [c_int64, c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_void_p]),
    "sg_col_blocks": (c_int64, [c_int64]),
    "sg_col_moments": (c_int, [c_void_p, c_int64, c_int64, c_int64, c_int, c_void_p, c_int64, c_void_p]),
    "sg_bn_merge": (c_int, [c_void_p, c_int64, c_int64, c_int64, c_void_p, c_void_p]),
    "sg_bn_stats_finalize": (c_int, [c_void_p, c_int64, c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_float,
                                     c_float, c_void_p, c_void_p, c_void_p]),
    "sg_bn_stats_finalize_tiles": (c_int, [c_void_p, c_int64, c_int64, c_int64, c_int64, c_void_p, c_void_p, c_void_p,
                                           c_void_p, c_float, c_float, c_void_p, c_void_p, c_void_p]),
    "sg_gemm_nt_takes_big_tile": (c_int, [c_int64, c_int64, c_int64, c_int64, c_int64, c_int64]),
    "sg_gemm_tile_rows": (c_int64, [c_int64]),
    "sg_gemm_row_tiles": (c_int64, [c_int64, c_int64]),
    "sg_gemm_nt": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_int64, c_int64, c_int64,
                           c_int, c_void_p, c_void_p]),
    "sg_gemm_nt_f32_supported": (c_int, [c_int64, c_int64, c_int64, c_int64, c_int64]),
    "sg_gemm_nt_f32_pays": (c_int, [c_int64, c_int64, c_int64]),
    "sg_gemm_nt_f32_variant": (c_int, [c_int64]),
    "sg_gemm_nt_f32_workspace": (c_int64, [c_int64, c_int64]),
    "sg_gemm_nt_f32": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_int64, c_int64, c_int64,
                               c_int64, c_void_p, c_int64, c_void_p]),
    "sg_gemm_tn_f32_supported": (c_int, [c_int64, c_int64, c_int64, c_int64, c_int64]),
    "sg_gemm_tn_f32_workspace": (c_int64, [c_int64, c_int64, c_int64]),
    "sg_gemm_tn_f32": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_int64, c_int64, c_int64, c_void_p, c_int64, c_void_p,
                               c_int64, c_void_p]),
    "sg_gemm_tn_slabs": (c_int64, [c_int64, c_int64, c_int64]),
    "sg_gemm_tn_takes_big_tile": (c_int, [c_int64, c_int64, c_int64, c_int64, c_int64]),
    "sg_gemm_tn": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_int64, c_int64, c_int64, c_int, c_void_p, c_void_p, c_int64,
                           c_void_p]),
    "sg_thin_supported": (c_int, [c_int64, c_int64]),
    "sg_thin_tn_blocks": (c_int64, [c_int64]),
    "sg_thin_nt": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_int64, c_int64, c_int64, c_int,
                           c_void_p]),
    "sg_thin_tn": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_int64, c_int64, c_int64, c_int, c_void_p, c_void_p, c_int64,
                           c_void_p]),
    "sg_bn_bwd_coeffs": (c_int, [c_void_p, c_int64, c_int64, ctypes.c_double, c_void_p, c_void_p, c_void_p, c_void_p,
                                 c_void_p, c_void_p, c_void_p]),
    "sg_bn_merge_tiles": (c_int, [c_void_p, c_int64, c_int64, c_int64, c_int64, c_void_p, c_void_p, c_void_p]),
    "sg_multi_add": (c_int, [c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "sg_input_bounds": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "sg_input_prep_bwd_routed": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                         c_void_p, c_int64, c_int, c_void_p]),
    "sg_mesh_loss_finalize": (c_int, [c_void_p, c_int64, c_float, c_float, c_float, c_float, c_void_p, c_void_p]),
    "sg_input_prep_blocks": (c_int64, [c_int64]),
    "sg_input_prep": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int, c_void_p]),
    "sg_input_prep_bwd": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                  c_void_p, c_void_p, c_int64, c_int, c_void_p]),
    "sg_bn_finalize_ranks": (c_int, [c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_float,
                                     c_void_p, c_void_p, c_void_p, c_void_p]),
    "sg_bn_finalize": (c_int, [c_void_p, ctypes.c_double, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_float,
                               c_float, c_void_p, c_void_p]),
    "sg_scale_shift_act": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_float, c_void_p, c_int64, c_int64,
                                   c_int64, c_int, c_void_p]),
    "sg_bn_act_bwd_reduce": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p,
                                     c_float, c_void_p, c_int64, c_int64, c_int64, c_int, c_void_p]),
    "sg_bn_act_bwd_apply": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p,
                                    c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_int64, c_int64, c_int64, c_int,
                                    c_void_p]),
    "sg_col_apply_blocks": (c_int64, [c_int64, c_int64, c_int]),
    "sg_bn_act_bwd_apply_colsum": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p,
                                           c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_int64, c_int64, c_int64, c_int,
                                           c_void_p, c_void_p, c_void_p]),
    "sg_gather_rows": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_int64, c_int,
                               c_void_p]),
    "sg_block_sizeof": (c_int64, []),
    "sg_block_workspace": (c_int64, [POINTER(sg_block), c_int]),
    "sg_block_planar": (c_int, [POINTER(sg_block)]),
    "sg_block_forward": (c_int, [POINTER(sg_block), c_void_p]),
    "sg_block_backward": (c_int, [POINTER(sg_block), c_void_p]),
    "sg_block_run": (c_int, [POINTER(sg_block), c_int64, c_void_p]),
    "sg_comm_available": (c_int, []),
    "sg_comm_unique_id": (c_int, [c_void_p]),
    "sg_comm_create": (c_int, [c_void_p, c_int, c_int, POINTER(c_int64), POINTER(c_int64), POINTER(c_void_p)]),
    "sg_comm_share": (c_int, [c_void_p, POINTER(c_int64), POINTER(c_int64), POINTER(c_void_p)]),
    "sg_comm_destroy": (c_int, [c_void_p]),
    "sg_halo_exchange": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    "sg_comm_all_reduce_f32": (c_int, [c_void_p, c_void_p, c_int64, c_void_p]),
    "sg_comm_all_gather": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    "sg_part_step_sizeof": (c_int64, []),
    "sg_part_run": (c_int, [c_void_p, POINTER(sg_part_step), c_int64, c_void_p]),
    "sg_part_failed_step": (c_int64, []),
    "sg_comm_test_stub": (c_int, [c_int]),
    "sg_comm_test_fail_send": (c_int, [c_int]),
    "sg_comm_test_log": (c_int64, [POINTER(c_int64), c_int64, POINTER(c_int64)]),
    "sg_block_chain_forward": (c_int, [POINTER(sg_block), c_int64, c_void_p]),
    "sg_block_chain_backward": (c_int, [POINTER(sg_block), c_int64, c_void_p]),
    "sg_trace_begin": (c_int, [c_int64, c_int]),
    "sg_trace_read": (c_int64, [POINTER(sg_trace_record), c_int64]),
    "sg_trace_end": (c_int, []),
    "sg_mesh_edges": (c_int, [c_void_p, c_int64, c_int64, c_void_p, c_void_p, POINTER(c_int64), POINTER(c_int),
                              c_void_p]),
    "sg_mask_dilate": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    "sg_face_mask": (c_int, [c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_int64, c_void_p]),
}

_lib = None


class LaunchTimer:
    """Optional per-launch timing of the aggregation kernel with HIP events recorded on the
    stream the kernel is launched on (torch's current stream).  bench.py installs one over
    its timed region; ``results()`` must be called after a device synchronize."""

    def __init__(self):
        self.records = []  # (key, start_event, end_event)

    def results(self):
        out = {}
        for key, a, b in self.records:
            out.setdefault(key, []).append(a.elapsed_time(b))  # milliseconds
        return out


_timer: Optional[LaunchTimer] = None


def set_launch_timer(timer: Optional[LaunchTimer]) -> None:
    global _timer
    _timer = timer


def library_path() -> str:
    return _LIB_PATH


def load():
    """Load the shared library (idempotent). Raises SemigcnLibraryError when it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.isfile(_LIB_PATH):
        raise SemigcnLibraryError(
            f"{_LIB_PATH} not found: build it with `make -C semigcn_amd/csrc` "
            "(or `python -c 'import __graft_entry__ as g; g.build()'`). "
            "semigcn_amd has no CPU or PyTorch fallback for its HIP kernels.")
    try:
        lib = ctypes.CDLL(_LIB_PATH, mode=ctypes.RTLD_LOCAL)
    except OSError as e:  # pragma: no cover - depends on the host
        raise SemigcnLibraryError(f"cannot load {_LIB_PATH}: {e}") from e
    for name, (res, args) in _SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise SemigcnLibraryError(f"{_LIB_PATH} does not export {name}") from e
        fn.restype, fn.argtypes = res, args
    if lib.sg_abi_version() != 1:
        raise SemigcnLibraryError(f"ABI version mismatch: library reports {lib.sg_abi_version()}")
    if lib.sg_block_sizeof() != ctypes.sizeof(sg_block):
        raise SemigcnLibraryError(f"struct sg_block: the library's has {lib.sg_block_sizeof()} bytes, the binding's "
                                  f"{ctypes.sizeof(sg_block)} -- rebuild the library (make -C semigcn_amd/csrc)")
    if lib.sg_part_step_sizeof() != ctypes.sizeof(sg_part_step):
        raise SemigcnLibraryError(f"struct sg_part_step: the library's has {lib.sg_part_step_sizeof()} bytes, the binding's "
                                  f"{ctypes.sizeof(sg_part_step)} -- rebuild the library (make -C semigcn_amd/csrc)")
    _lib = lib
    if os.environ.get("SEMIGCN_F32_ENGINE"):               # A/B and bisecting runs: see SG_TUNE_F32_ENGINE in include/semigcn.h
        lib.sg_tuning_set(8, int(os.environ["SEMIGCN_F32_ENGINE"]))
    if os.environ.get("SEMIGCN_BN_ROWS"):                  # A/B runs: SG_TUNE_BN_ROWS
        lib.sg_tuning_set(9, int(os.environ["SEMIGCN_BN_ROWS"]))
    return lib


def _check(rc: int, what: str):
    if rc != 0:
        msg = load().sg_last_error()
        raise SemigcnLibraryError(f"{what} failed (code {rc}): {msg.decode() if msg else '?'}")


def _require_device(t: torch.Tensor, name: str):
    if not t.is_cuda:
        raise SemigcnLibraryError(
            f"{name} is on {t.device}: the semigcn_amd kernels run on a HIP device only "
            "(there is no CPU path; move the model and its inputs to cuda)")


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_current_device = getattr(torch._C, "_cuda_getDevice", None) or torch.cuda.current_device


def _stream(t: torch.Tensor) -> int:
    """hipStream_t of torch's CURRENT stream on the tensor's device (looked up on every launch: a
    caller may have switched streams; the raw accessor costs ~0.3 us against ~5 us for the Stream object)."""
    if _raw_stream is not None:
        idx = t.device.index
        return _raw_stream(_current_device() if idx is None else idx)
    return torch.cuda.current_stream(t.device).cuda_stream


class _NoGuard:
    __slots__ = ()

    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False


_NO_GUARD = _NoGuard()


def _on_device(device: torch.device):
    """``torch.cuda.device(dev)`` only when ``dev`` is not already current (the context manager costs
    ~10 us of host time per call, which matters at ~500 launches per training iteration)."""
    idx = device.index
    return _NO_GUARD if (idx is None or idx == _current_device()) else torch.cuda.device(device)


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    """Device address for a ``c_void_p`` argument (ctypes takes the int as it is; None = NULL)."""
    return None if t is None else t.data_ptr()


def _rows2d(t: torch.Tensor, name: str) -> int:
    """Row stride (elements) of a 2-D tensor whose rows are contiguous."""
    if t.dim() != 2 or (t.shape[1] > 1 and t.stride(1) != 1):
        raise SemigcnLibraryError(f"{name}: need a 2-D tensor with unit column stride, got "
                                  f"shape {tuple(t.shape)} strides {t.stride()}")
    return t.stride(0) if t.shape[0] > 1 else max(t.stride(0), t.shape[1])


def dtype_code(t: torch.Tensor) -> int:
    try:
        return _DTYPES[t.dtype]
    except KeyError:
        raise SemigcnLibraryError(f"unsupported dtype {t.dtype} (float32 and bfloat16 only)") from None


class GraphHandle:
    """Owns one sg_graph (CSR of the scaled Laplacian of one edge_index)."""

    def __init__(self, handle: int, device: torch.device):
        self._h = c_void_p(handle)
        self.device = device
        info = sg_graph_info()
        _check(load().sg_graph_query(self._h, byref(info)), "sg_graph_query")
        self.num_rows, self.num_cols, self.nnz = info.V_dst, info.V_src, info.nnz
        self.symmetric, self.max_degree = bool(info.symmetric), info.max_degree
        self.rows_processed = self.num_rows      # rows one sg_spmm launch computes (a row-subset handle: fewer than num_rows)
        self.reordered = load().sg_graph_is_reordered(self._h) == 1     # rows processed in a graph-derived locality order

    @classmethod
    def from_edge_index(cls, edge_index: torch.Tensor, num_vertices: int) -> "GraphHandle":
        _require_device(edge_index, "edge_index")
        if edge_index.dtype != torch.int64 or edge_index.dim() != 2 or edge_index.shape[0] != 2:
            raise SemigcnLibraryError("edge_index must be int64 [2, E]")
        ei = edge_index.contiguous()
        out = c_void_p()
        with _on_device(ei.device):
            _check(load().sg_graph_create(_ptr(ei), ei.shape[1], int(num_vertices), _stream(ei), byref(out)),
                   "sg_graph_create")
        return cls(out.value, ei.device)

    @classmethod
    def from_partition(cls, dst: torch.Tensor, src: torch.Tensor, n_owned: int, n_ext: int,
                       dis_ext: torch.Tensor) -> "GraphHandle":
        for t, n in ((dst, "dst"), (src, "src"), (dis_ext, "dis_ext")):
            _require_device(t, n)
        dst, src = dst.contiguous(), src.contiguous()
        dis_ext = dis_ext.contiguous().float()
        assert dst.dtype == torch.int64 and src.dtype == torch.int64 and dis_ext.numel() == n_ext
        out = c_void_p()
        with _on_device(dst.device):
            _check(load().sg_graph_create_rect(_ptr(dst), _ptr(src), dst.numel(), int(n_owned), int(n_ext),
                                               _ptr(dis_ext), _stream(dst), byref(out)),
                   "sg_graph_create_rect")
        return cls(out.value, dst.device)

    @classmethod
    def from_rows(cls, dst_pos: torch.Tensor, src: torch.Tensor, row_id: torch.Tensor, out_rows: int, n_ext: int,
                  dis_rows: torch.Tensor, dis_ext: torch.Tensor) -> "GraphHandle":
        """Row subset of a partition operator: processed row p (``dst_pos`` values) writes row ``row_id[p]`` of a Y with
        ``out_rows`` rows; ``src`` indexes the ``n_ext`` rows of X."""
        for t, n in ((dst_pos, "dst_pos"), (src, "src"), (row_id, "row_id"), (dis_rows, "dis_rows"), (dis_ext, "dis_ext")):
            _require_device(t, n)
        dst_pos, src = dst_pos.contiguous().long(), src.contiguous().long()
        row_id = row_id.contiguous().to(torch.int32)
        dis_rows, dis_ext = dis_rows.contiguous().float(), dis_ext.contiguous().float()
        assert dis_rows.numel() == row_id.numel() and dis_ext.numel() == n_ext
        out = c_void_p()
        with _on_device(src.device):
            _check(load().sg_graph_create_rows(_ptr(dst_pos), _ptr(src), dst_pos.numel(), row_id.numel(), int(n_ext),
                                               _ptr(row_id), _ptr(dis_rows), _ptr(dis_ext), _stream(src), byref(out)),
                   "sg_graph_create_rows")
        h = cls(out.value, src.device)
        h.num_rows = int(out_rows)          # rows of the Y it writes into (it touches only the rows row_id names)
        return h                            # (rows_processed stays the subset's size)

    def arrays(self):
        """(rowptr int32 [rows+1], colidx int32 [nnz], dis float32 [cols]) as torch tensors."""
        rp = torch.empty(self.num_rows + 1, dtype=torch.int32, device=self.device)
        ci = torch.empty(self.nnz, dtype=torch.int32, device=self.device)
        ds = torch.empty(self.num_cols, dtype=torch.float32, device=self.device)
        with _on_device(self.device):
            _check(load().sg_graph_export(self._h, _ptr(rp), _ptr(ci), _ptr(ds), _stream(rp)), "sg_graph_export")
        return rp, ci, ds

    def prepare(self, C: int, dtype: torch.dtype) -> None:
        """Build the tile records an aggregation of C channels of ``dtype`` would otherwise build at its first call
        (``sg_graph_prepare``: that call allocates and synchronises); afterwards every ``spmm`` is asynchronous."""
        with _on_device(self.device):
            _check(load().sg_graph_prepare(self._h, int(C), _DTYPES[dtype], torch.cuda.current_stream(self.device).cuda_stream),
                   "sg_graph_prepare")

    def spmm(self, X: torch.Tensor, Y: torch.Tensor, *, alpha: float = 1.0, X0: Optional[torch.Tensor] = None,
             beta: float = 0.0, X1: Optional[torch.Tensor] = None, gamma: float = 0.0,
             transpose: bool = False) -> torch.Tensor:
        """Y = alpha * op(L^) X + beta * X0 + gamma * X1 (in place into Y)."""
        # Fast path (~50 calls per training iteration, host-bound on small meshes): the common, well-formed call is
        # recognised with a handful of attribute reads and handed to the library with plain ints; anything else --
        # including every malformed call -- goes through _spmm_checked, which raises the precise error.
        if _timer is None and X.is_cuda and X.dim() == 2 and Y.dim() == 2 and X.device.index == _current_device():
            dt = X.dtype
            code = _DTYPES.get(dt)
            C = X.shape[1]
            n_in, n_out = (self.num_rows, self.num_cols) if transpose else (self.num_cols, self.num_rows)
            sx, sy = X.stride(), Y.stride()
            ok = (code is not None and X.shape[0] == n_in and Y.dtype == dt and Y.device == X.device and Y.shape[0] == n_out
                  and Y.shape[1] == C and (C <= 1 or (sx[1] == 1 and sy[1] == 1)))
            p0 = l0 = p1 = l1 = 0
            if ok and X0 is not None:
                s0 = X0.stride()
                ok = (X0.dtype == dt and X0.device == X.device and X0.dim() == 2 and X0.shape[0] == n_out and X0.shape[1] == C
                      and (C <= 1 or s0[1] == 1))
                p0, l0 = X0.data_ptr(), (s0[0] if n_out > 1 else max(s0[0], C))
            if ok and X1 is not None:
                s1 = X1.stride()
                ok = (X1.dtype == dt and X1.device == X.device and X1.dim() == 2 and X1.shape[0] == n_out and X1.shape[1] == C
                      and (C <= 1 or s1[1] == 1))
                p1, l1 = X1.data_ptr(), (s1[0] if n_out > 1 else max(s1[0], C))
            if ok and _raw_stream is not None:
                rc = _lib.sg_spmm(self._h, 1 if transpose else 0, X.data_ptr(), sx[0] if n_in > 1 else max(sx[0], C),
                                  p0, l0, p1, l1, Y.data_ptr(), sy[0] if n_out > 1 else max(sy[0], C), C, code,
                                  alpha, beta, gamma, _raw_stream(X.device.index))
                if rc:
                    _check(rc, "sg_spmm")
                return Y
        return self._spmm_checked(X, Y, alpha, X0, beta, X1, gamma, transpose)

    def _spmm_checked(self, X, Y, alpha, X0, beta, X1, gamma, transpose) -> torch.Tensor:
        for t, n in ((X, "X"), (Y, "Y"), (X0, "X0"), (X1, "X1")):
            if t is not None:
                _require_device(t, n)
                if t.dtype != X.dtype:
                    raise SemigcnLibraryError(f"{n} dtype {t.dtype} != X dtype {X.dtype}")
        C = X.shape[1]
        n_in, n_out = (self.num_rows, self.num_cols) if transpose else (self.num_cols, self.num_rows)
        if X.shape[0] != n_in or Y.shape != (n_out, C):
            raise SemigcnLibraryError(f"spmm shape mismatch: X {tuple(X.shape)} Y {tuple(Y.shape)} "
                                      f"operator {n_out}x{n_in}")
        for t, n in ((X0, "X0"), (X1, "X1")):
            if t is not None and t.shape != (n_out, C):
                raise SemigcnLibraryError(f"{n} shape {tuple(t.shape)} != {(n_out, C)}")
        timer = _timer
        with _on_device(X.device):
            if timer is not None:
                ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                ev0.record()
            _check(load().sg_spmm(self._h, int(transpose), _ptr(X), _rows2d(X, "X"),
                                  _ptr(X0), 0 if X0 is None else _rows2d(X0, "X0"),
                                  _ptr(X1), 0 if X1 is None else _rows2d(X1, "X1"),
                                  _ptr(Y), _rows2d(Y, "Y"), C, dtype_code(X),
                                  float(alpha), float(beta), float(gamma), _stream(X)), "sg_spmm")
            if timer is not None:
                ev1.record()
                n_epi = int(X0 is not None) + int(X1 is not None)
                timer.records.append(((C, str(X.dtype).replace("torch.", ""), n_epi, self.rows_processed), ev0, ev1))
        return Y

    def dilate_bits(self, bits: torch.Tensor) -> torch.Tensor:
        """One ring of dilation of bit-packed vertex masks: int64 [V, W] -> int64 [V, W]."""
        _require_device(bits, "bits")
        if bits.dtype != torch.int64 or bits.dim() != 2 or bits.shape[0] != self.num_rows:
            raise SemigcnLibraryError(f"bits must be int64 [{self.num_rows}, W], got {bits.dtype} {tuple(bits.shape)}")
        bits = bits.contiguous()
        out = torch.empty_like(bits)
        with _on_device(bits.device):
            _check(load().sg_mask_dilate(self._h, _ptr(bits), _ptr(out), bits.shape[1], _stream(bits)),
                   "sg_mask_dilate")
        return out

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            load().sg_graph_destroy(self._h)
            self._h = c_void_p(0)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class PoolHandle:
    """Owns one sg_pool (cluster map of one pool_hash)."""

    def __init__(self, fine: torch.Tensor, coarse: torch.Tensor, n_fine: int, n_coarse: int):
        _require_device(fine, "fine")
        _require_device(coarse, "coarse")
        fine, coarse = fine.contiguous().long(), coarse.contiguous().long()
        self.device, self.n_fine, self.n_coarse = fine.device, int(n_fine), int(n_coarse)
        out = c_void_p()
        with _on_device(fine.device):
            _check(load().sg_pool_create(_ptr(fine), _ptr(coarse), fine.numel(), self.n_fine, self.n_coarse,
                                         _stream(fine), byref(out)), "sg_pool_create")
        self._h = out

    def _run(self, fn_name: str, X: torch.Tensor, n_in: int, n_out: int) -> torch.Tensor:
        _require_device(X, "X")
        if X.dim() != 2 or X.shape[0] != n_in:
            raise SemigcnLibraryError(f"{fn_name}: expected [{n_in}, C], got {tuple(X.shape)}")
        if X.stride(1) != 1 and X.shape[1] > 1:
            X = X.contiguous()
        Y = torch.empty((n_out, X.shape[1]), dtype=X.dtype, device=X.device)
        with _on_device(X.device):
            _check(getattr(load(), fn_name)(self._h, _ptr(X), _rows2d(X, "X"), _ptr(Y), _rows2d(Y, "Y"),
                                            X.shape[1], dtype_code(X), _stream(X)), fn_name)
        return Y

    def pool_mean(self, X):
        return self._run("sg_pool_mean", X, self.n_fine, self.n_coarse)

    def pool_mean_bwd(self, dY):
        return self._run("sg_pool_mean_bwd", dY, self.n_coarse, self.n_fine)

    def unpool(self, X):
        return self._run("sg_unpool", X, self.n_coarse, self.n_fine)

    def unpool_bwd(self, dY):
        return self._run("sg_unpool_bwd", dY, self.n_fine, self.n_coarse)

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            load().sg_pool_destroy(self._h)
            self._h = c_void_p(0)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


TUNE_CHUNK_ROWS, TUNE_FLAGS, TUNE_UNROLL, TUNE_SLAB, TUNE_TILED_MIN_ROW_BYTES, TUNE_GEMM_TILE, TUNE_GRAPH_REORDER = 0, 1, 2, 3, 4, 5, 6
TUNE_BLOCK_PLANES = 7
TUNE_F32_ENGINE = 8
TUNE_BN_ROWS = 9


#: bumped by every tuning_set: a BlockPlan's packed weight copies (the split-bf16 images in particular: their tile shape follows
#: SG_TUNE_F32_ENGINE) are rebuilt at the first call after a knob changed
tuning_generation = [0]


def tuning_set(knob: int, value: int) -> None:
    """Launch tuning of the aggregation kernel (benchmarking aid; results do not depend on it)."""
    _check(load().sg_tuning_set(int(knob), int(value)), "sg_tuning_set")
    _sizes.cache_clear()          # the GEMM tile knob changes sg_gemm_tile_rows / sg_gemm_row_tiles
    tuning_generation[0] += 1     # packed weight images follow the knobs (functional.BlockPlan.stale re-packs after a change)


@functools.lru_cache(maxsize=512)
def _sizes(fn: str, *args: int) -> int:
    """The library's buffer-sizing rules (sg_col_blocks, sg_gemm_tn_slabs, ...): pure functions of their integer
    arguments (and of the tuning knobs: cleared by tuning_set), asked ~100 times per training iteration with the same
    few arguments -- answered from a cache instead of a foreign call each."""
    return int(getattr(load(), fn)(*args))


def gather_rows(rows: torch.Tensor, X: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out[i] = X[rows[i]]; rows int32 on the device (halo packing)."""
    _require_device(X, "X")
    _require_device(rows, "rows")
    if rows.dtype != torch.int32:
        raise SemigcnLibraryError("rows must be int32")
    rows = rows.contiguous()
    if out is None:
        out = torch.empty((rows.numel(), X.shape[1]), dtype=X.dtype, device=X.device)
    with _on_device(X.device):
        _check(load().sg_gather_rows(_ptr(rows), rows.numel(), _ptr(X), _rows2d(X, "X"), _ptr(out),
                                     _rows2d(out, "out"), X.shape[1], dtype_code(X), _stream(X)),
               "sg_gather_rows")
    return out


# ---- BatchNorm(+LeakyReLU) over the vertex axis -------------------------------------------------
def col_blocks(num_rows: int) -> int:
    return _sizes("sg_col_blocks", int(num_rows))


def _f32vec(t: torch.Tensor, n: int, name: str) -> torch.Tensor:
    if t.dtype != torch.float32 or t.numel() != n or not t.is_contiguous():
        raise SemigcnLibraryError(f"{name} must be a contiguous float32 vector of {n} elements")
    _require_device(t, name)
    return t


def _counter(t: Optional[torch.Tensor], like: torch.Tensor) -> Optional[int]:
    """Device pointer of nn.BatchNorm1d's ``num_batches_tracked`` (an int64 scalar) -- incremented by the finalize launch."""
    if t is None:
        return None
    if t.dtype != torch.int64 or t.numel() != 1 or t.device != like.device:
        raise SemigcnLibraryError("num_batches_tracked must be an int64 scalar on the device of the statistics")
    return t.data_ptr()


def col_moments(X: torch.Tensor) -> torch.Tensor:
    """partial[b] = (mean, sum of squared deviations) per channel over row block b; float32 [nb, 2, C]."""
    _require_device(X, "X")
    V, C = X.shape
    nb = col_blocks(V)
    part = torch.empty((nb, 2, C), dtype=torch.float32, device=X.device)
    with _on_device(X.device):
        _check(load().sg_col_moments(_ptr(X), _rows2d(X, "X"), V, C, dtype_code(X), _ptr(part), nb, _stream(X)),
               "sg_col_moments")
    return part


def bn_merge(partial: torch.Tensor, num_rows: int, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """(mean, M2) [2, C] over all rows from the per-block partials of col_moments (``out``: 2C contiguous floats)."""
    nb, _, C = partial.shape
    stats = torch.empty((2, C), dtype=torch.float32, device=partial.device) if out is None else out
    with _on_device(partial.device):
        _check(load().sg_bn_merge(_ptr(partial), nb, int(num_rows), C, _ptr(stats), _stream(partial)), "sg_bn_merge")
    return stats


def bn_local_stats(partial: torch.Tensor, rows_per_tile: int, num_rows: int, out: torch.Tensor) -> torch.Tensor:
    """``out`` (2C + 1 contiguous floats) = (mean[C], M2[C], row count) of this device's rows from per-tile moments -- the
    blocks of col_moments (``rows_per_tile = 0``) or the tiles sg_gemm_nt emits: the row a rank contributes to the
    all-gather of a vertex-partitioned BatchNorm, in one launch."""
    nb, _, C = partial.shape
    if out.dtype != torch.float32 or out.numel() != 2 * C + 1 or not out.is_contiguous():
        raise SemigcnLibraryError(f"bn_local_stats: out must hold 2C + 1 = {2 * C + 1} contiguous floats")
    rpt = int(rows_per_tile) if rows_per_tile else (int(num_rows) + nb - 1) // nb
    base = out.data_ptr()
    with _on_device(partial.device):
        _check(load().sg_bn_merge_tiles(_ptr(partial), nb, rpt, int(num_rows), C, base, base + 8 * C, _stream(partial)),
               "sg_bn_merge_tiles")
    return out


def bn_finalize(stats: torch.Tensor, count: float, gamma: torch.Tensor, beta: torch.Tensor,
                running_mean: Optional[torch.Tensor], running_var: Optional[torch.Tensor], momentum: float,
                eps: float) -> torch.Tensor:
    """[4, C] = (mean, invstd, scale, shift); updates the running statistics in place when given."""
    C = stats.shape[1]
    out = torch.empty((4, C), dtype=torch.float32, device=stats.device)
    for t, n in ((running_mean, "running_mean"), (running_var, "running_var")):
        if t is not None:
            _f32vec(t, C, n)
    with _on_device(stats.device):
        _check(load().sg_bn_finalize(_ptr(stats), float(count), C, _ptr(_f32vec(gamma, C, "weight")),
                                     _ptr(_f32vec(beta, C, "bias")), _ptr(running_mean), _ptr(running_var),
                                     float(momentum), float(eps), _ptr(out), _stream(stats)), "sg_bn_finalize")
    return out


def bn_finalize_ranks(all_stats: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor,
                      running_mean: Optional[torch.Tensor], running_var: Optional[torch.Tensor], momentum: float,
                      eps: float, batches_tracked: Optional[torch.Tensor] = None):
    """[world, 2C+1] gathered rows of (mean, M2, count) -> ([4, C] mean/invstd/scale/shift of the whole mesh,
    [1] total row count), everything on the device."""
    world, w = all_stats.shape
    C = (w - 1) // 2
    out = torch.empty((4, C), dtype=torch.float32, device=all_stats.device)
    n = torch.empty((1,), dtype=torch.float32, device=all_stats.device)
    with _on_device(all_stats.device):
        _check(load().sg_bn_finalize_ranks(_ptr(all_stats), world, C, _ptr(_f32vec(gamma, C, "weight")),
                                           _ptr(_f32vec(beta, C, "bias")), _ptr(running_mean), _ptr(running_var),
                                           float(momentum), float(eps), _ptr(out), _ptr(n),
                                           _counter(batches_tracked, all_stats), _stream(all_stats)),
               "sg_bn_finalize_ranks")
    return out, n


def bn_stats_finalize(partial: torch.Tensor, num_rows: int, gamma: torch.Tensor, beta: torch.Tensor,
                      running_mean: Optional[torch.Tensor], running_var: Optional[torch.Tensor], momentum: float,
                      eps: float, batches_tracked: Optional[torch.Tensor] = None) -> torch.Tensor:
    """bn_merge + bn_finalize in one launch (statistics over this device's rows only): [4, C].  ``batches_tracked``:
    the module's num_batches_tracked, += 1 by the same launch."""
    nb, _, C = partial.shape
    out = torch.empty((4, C), dtype=torch.float32, device=partial.device)
    for t, n in ((running_mean, "running_mean"), (running_var, "running_var")):
        if t is not None:
            _f32vec(t, C, n)
    with _on_device(partial.device):
        _check(load().sg_bn_stats_finalize(_ptr(partial), nb, int(num_rows), C, _ptr(_f32vec(gamma, C, "weight")),
                                           _ptr(_f32vec(beta, C, "bias")), _ptr(running_mean), _ptr(running_var),
                                           float(momentum), float(eps), _ptr(out), _counter(batches_tracked, partial),
                                           _stream(partial)), "sg_bn_stats_finalize")
    return out


def bn_stats_finalize_tiles(partial: torch.Tensor, rows_per_tile: int, num_rows: int, gamma: torch.Tensor,
                            beta: torch.Tensor, running_mean: Optional[torch.Tensor], running_var: Optional[torch.Tensor],
                            momentum: float, eps: float, batches_tracked: Optional[torch.Tensor] = None) -> torch.Tensor:
    """bn_stats_finalize for the per-tile moments the MFMA GEMM emits (uniform tiles of ``rows_per_tile`` rows)."""
    nb, _, C = partial.shape
    out = torch.empty((4, C), dtype=torch.float32, device=partial.device)
    for t, n in ((running_mean, "running_mean"), (running_var, "running_var")):
        if t is not None:
            _f32vec(t, C, n)
    with _on_device(partial.device):
        _check(load().sg_bn_stats_finalize_tiles(_ptr(partial), nb, int(rows_per_tile), int(num_rows), C,
                                                 _ptr(_f32vec(gamma, C, "weight")), _ptr(_f32vec(beta, C, "bias")),
                                                 _ptr(running_mean), _ptr(running_var), float(momentum), float(eps),
                                                 _ptr(out), _counter(batches_tracked, partial), _stream(partial)),
               "sg_bn_stats_finalize_tiles")
    return out


# ---- dense feature x weight product on the matrix cores ------------------------------------------
def gemm_tile_rows(N: int) -> int:
    """Rows per output tile (and per BatchNorm-moments record) of an N-column sg_gemm_nt product."""
    return _sizes("sg_gemm_tile_rows", int(N))


def gemm_nt_supported(A: torch.Tensor, B: torch.Tensor, ldc: int) -> bool:
    """Shapes sg_gemm_nt takes: bf16, unit column strides, K / N / row strides multiples of 8, 16-byte aligned."""
    if A.dtype != torch.bfloat16 or B.dtype != torch.bfloat16 or not A.is_cuda or A.dim() != 2 or B.dim() != 2:
        return False
    if A.shape[1] != B.shape[1] or A.stride(1) != 1 or B.stride(1) != 1 or A.shape[0] == 0:
        return False
    K, N = A.shape[1], B.shape[0]
    return (K % 8 == 0 and N % 8 == 0 and A.stride(0) % 8 == 0 and B.stride(0) % 8 == 0 and ldc % 8 == 0
            and A.data_ptr() % 16 == 0 and B.data_ptr() % 16 == 0)


def gemm_nt_takes_big_tile(M: int, N: int, K: int, lda: int, ldb: int, ldc: int) -> bool:
    """True when a moments-free sg_gemm_nt call of this shape is served by the persistent 256 x 256 kernel
    (csrc/gemm_mfma256.hip: the compute-bound products) rather than by the 128-row-tile kernel."""
    return bool(_sizes("sg_gemm_nt_takes_big_tile", int(M), int(N), int(K), int(lda), int(ldb), int(ldc)))


def gemm_nt(A: torch.Tensor, B: torch.Tensor, bias: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None,
            moments: bool = False):
    """``out = A @ B.T (+ bias)`` on the MFMA kernel (csrc/gemm_mfma.hip): A [M, K] and B [N, K] bf16 with unit column
    stride, bias fp32 [N], out bf16 [M, N] (any row stride that is a multiple of 8).  ``moments=True`` also returns
    the float32 [ceil(M / gemm_tile_rows(N)), 2, N] per-tile column (mean, M2) of the rounded result."""
    _require_device(A, "A")
    _require_device(B, "B")
    M, K = A.shape
    N = B.shape[0]
    if out is None:
        out = torch.empty((M, N), dtype=torch.bfloat16, device=A.device)
    if out.shape != (M, N) or out.dtype != torch.bfloat16 or B.shape[1] != K:
        raise SemigcnLibraryError(f"gemm_nt shape mismatch: A {tuple(A.shape)} B {tuple(B.shape)} out {tuple(out.shape)}")
    if bias is not None:
        _f32vec(bias, N, "bias")
    mom = None
    if moments:
        mom = torch.empty((_sizes("sg_gemm_row_tiles", M, N), 2, N), dtype=torch.float32, device=A.device)
    with _on_device(A.device):
        _check(load().sg_gemm_nt(_ptr(A), _rows2d(A, "A"), _ptr(B), _rows2d(B, "B"), _ptr(bias), _ptr(out),
                                 _rows2d(out, "out"), M, N, K, SG_BF16, _ptr(mom), _stream(A)), "sg_gemm_nt")
    return (out, mom) if moments else out


THIN_ELEMS = 256      # sg_thin_nt / sg_thin_tn: weight matrices of at most 256 entries (see thin_shape)


def thin_shape(N: int, K: int) -> bool:
    """sg_thin_supported: K <= 8 with N <= 32, or N, K <= 16, or K <= 32 with N <= 8."""
    return bool(_sizes("sg_thin_supported", int(N), int(K)))


def thin_supported(A: torch.Tensor, N: int, K: int) -> bool:
    """Shapes the thin products take: a device tensor [V, *] with unit column stride, fp32 or bf16, a thin weight shape."""
    return (A.is_cuda and A.dim() == 2 and A.stride(1) == 1 and A.dtype in (torch.float32, torch.bfloat16)
            and N >= 1 and K >= 1 and thin_shape(N, K) and A.shape[0] > 0)


def thin_nt(X: torch.Tensor, W: torch.Tensor, bias: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``out = X @ W.T (+ bias)`` for a tiny weight matrix W [N, K] (N, K <= 16; fp32, any row stride): X [V, K] and out
    [V, N] fp32 or bf16 (same dtype, unit column stride), fp32 accumulation (csrc/thin_gemm.hip)."""
    _require_device(X, "X")
    V, K = X.shape
    N = W.shape[0]
    W = W if (W.dtype == torch.float32 and W.stride(1) == 1) else W.float().contiguous()
    if out is None:
        out = torch.empty((V, N), dtype=X.dtype, device=X.device)
    if out.shape != (V, N) or out.dtype != X.dtype or W.shape[1] != K or out.stride(1) != 1 or X.stride(1) != 1:
        raise SemigcnLibraryError(f"thin_nt shape mismatch: X {tuple(X.shape)} W {tuple(W.shape)} out {tuple(out.shape)}")
    if bias is not None:
        _f32vec(bias, N, "bias")
    with _on_device(X.device):
        _check(load().sg_thin_nt(_ptr(X), _rows2d(X, "X"), _ptr(W), _rows2d(W, "W"), _ptr(bias), _ptr(out), _rows2d(out, "out"),
                                 V, N, K, dtype_code(X), _stream(X)), "sg_thin_nt")
    return out


def thin_tn(A: torch.Tensor, B: torch.Tensor) -> torch.Tensor:
    """``A.T @ B`` in float32 for A [V, N], B [V, K] (N, K <= 16; both fp32 or both bf16, unit column stride): the
    weight gradient of a tiny layer, per-block partial sums added in block order (deterministic)."""
    _require_device(A, "A")
    _require_device(B, "B")
    V, N = A.shape
    K = B.shape[1]
    if B.shape[0] != V or A.dtype != B.dtype or A.stride(1) != 1 or B.stride(1) != 1:
        raise SemigcnLibraryError(f"thin_tn shape mismatch: A {tuple(A.shape)} B {tuple(B.shape)}")
    out = torch.empty((N, K), dtype=torch.float32, device=A.device)
    ws = torch.empty((_sizes("sg_thin_tn_blocks", V), THIN_ELEMS), dtype=torch.float32, device=A.device)
    with _on_device(A.device):
        _check(load().sg_thin_tn(_ptr(A), _rows2d(A, "A"), _ptr(B), _rows2d(B, "B"), V, N, K, dtype_code(A), _ptr(ws), _ptr(out),
                                 K, _stream(A)), "sg_thin_tn")
    return out


def gemm_tn_supported(A: torch.Tensor, B: torch.Tensor) -> bool:
    if A.dtype != torch.bfloat16 or B.dtype != torch.bfloat16 or not A.is_cuda or A.dim() != 2 or B.dim() != 2:
        return False
    if A.shape[0] != B.shape[0] or A.shape[0] == 0 or A.stride(1) != 1 or B.stride(1) != 1:
        return False
    return (A.shape[1] % 8 == 0 and B.shape[1] % 8 == 0 and A.stride(0) % 8 == 0 and B.stride(0) % 8 == 0
            and A.data_ptr() % 16 == 0 and B.data_ptr() % 16 == 0)


def gemm_tn_takes_big_tile(M: int, N: int, Kp: int, lda: int, ldb: int) -> bool:
    """True when sg_gemm_tn serves this weight gradient with the persistent 256 x 256 ring (csrc/gemm_mfma256.hip)."""
    return bool(_sizes("sg_gemm_tn_takes_big_tile", int(M), int(N), int(Kp), int(lda), int(ldb)))


def gemm_tn(A: torch.Tensor, B: torch.Tensor) -> torch.Tensor:
    """``A.T @ B`` in float32 for bf16 A [M, N], B [M, Kp] (unit column stride) on the MFMA kernel: the weight gradient
    ``dOut^T [Tx0|Tx1|Tx2]``; deterministic (slab partials summed in order)."""
    _require_device(A, "A")
    _require_device(B, "B")
    M, N = A.shape
    Kp = B.shape[1]
    if B.shape[0] != M:
        raise SemigcnLibraryError(f"gemm_tn shape mismatch: A {tuple(A.shape)} B {tuple(B.shape)}")
    out = torch.empty((N, Kp), dtype=torch.float32, device=A.device)
    ws = torch.empty((_sizes("sg_gemm_tn_slabs", M, N, Kp), N, Kp), dtype=torch.float32, device=A.device)
    with _on_device(A.device):
        _check(load().sg_gemm_tn(_ptr(A), _rows2d(A, "A"), _ptr(B), _rows2d(B, "B"), M, N, Kp, SG_BF16, _ptr(ws), _ptr(out),
                                 Kp, _stream(A)), "sg_gemm_tn")
    return out


def gemm_nt_f32_supported(A: torch.Tensor, N: int, ldc: Optional[int] = None) -> bool:
    """Shapes sg_gemm_nt_f32 takes (float32 on the bf16 matrix cores, exact three-way split): see include/semigcn.h."""
    if A.dtype != torch.float32 or not A.is_cuda or A.dim() != 2 or A.stride(1) != 1 or A.data_ptr() % 16:
        return False
    M, K = A.shape
    return bool(_sizes("sg_gemm_nt_f32_supported", int(M), int(N), int(K), int(A.stride(0)), int(ldc if ldc is not None else N)))


def gemm_nt_f32_workspace(N: int, K: int) -> int:
    """Bytes of the split-bf16 image of an [N, K] weight matrix (``sg_gemm_nt_f32_workspace``; 0 for small-weight shapes)."""
    return _sizes("sg_gemm_nt_f32_workspace", int(N), int(K))


def gemm_nt_f32_variant(M: int) -> int:
    """Tile variant (= layout of the split weight image) that serves a float32 product of M rows (``sg_gemm_nt_f32_variant``)."""
    return _sizes("sg_gemm_nt_f32_variant", int(M))


def gemm_nt_f32_pays(M: int, N: int, K: int) -> bool:
    """The library's rule for "own float32 kernels or the BLAS library" (``sg_gemm_nt_f32_pays``; follows SG_TUNE_F32_ENGINE)."""
    return bool(load().sg_gemm_nt_f32_pays(int(M), int(N), int(K)))


def gemm_nt_f32(A: torch.Tensor, W: torch.Tensor, bias: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None,
                w_is_kn: bool = False) -> torch.Tensor:
    """``A @ W.T (+ bias)`` for float32 A [M, K] and W [N, K] -- or ``A @ W`` for W [K, N] with ``w_is_kn`` -- on the bf16
    matrix cores at float32-equivalent error (csrc/gemm_split.hip)."""
    _require_device(A, "A")
    _require_device(W, "W")
    M, K = A.shape
    N = W.shape[1] if w_is_kn else W.shape[0]
    if (W.shape[0] if w_is_kn else W.shape[1]) != K or A.dtype != torch.float32 or W.dtype != torch.float32:
        raise SemigcnLibraryError(f"gemm_nt_f32 shape / dtype mismatch: A {tuple(A.shape)} W {tuple(W.shape)}")
    if out is None:
        out = torch.empty((M, N), dtype=torch.float32, device=A.device)
    rs, cs = (W.stride(1), W.stride(0)) if w_is_kn else (W.stride(0), W.stride(1))
    nbytes = _sizes("sg_gemm_nt_f32_workspace", N, K)
    ws = torch.empty((nbytes,), dtype=torch.uint8, device=A.device)
    with _on_device(A.device):
        _check(load().sg_gemm_nt_f32(_ptr(A), _rows2d(A, "A"), _ptr(W), rs, cs, _ptr(bias), _ptr(out), _rows2d(out, "out"),
                                     M, N, K, _ptr(ws), nbytes, _stream(A)), "sg_gemm_nt_f32")
    return out


def gemm_tn_f32_supported(A: torch.Tensor, B: torch.Tensor) -> bool:
    if A.dtype != torch.float32 or B.dtype != torch.float32 or not A.is_cuda or A.dim() != 2 or B.dim() != 2:
        return False
    if A.shape[0] != B.shape[0] or A.stride(1) != 1 or B.stride(1) != 1 or A.data_ptr() % 16 or B.data_ptr() % 16:
        return False
    return bool(_sizes("sg_gemm_tn_f32_supported", int(A.shape[0]), int(A.shape[1]), int(B.shape[1]), int(A.stride(0)),
                       int(B.stride(0))))


def gemm_tn_f32(A: torch.Tensor, B: torch.Tensor) -> torch.Tensor:
    """``A.T @ B`` for float32 A [M, N], B [M, Kp] on the bf16 matrix cores at float32-equivalent error; deterministic."""
    _require_device(A, "A")
    _require_device(B, "B")
    M, N = A.shape
    Kp = B.shape[1]
    if B.shape[0] != M:
        raise SemigcnLibraryError(f"gemm_tn_f32 shape mismatch: A {tuple(A.shape)} B {tuple(B.shape)}")
    out = torch.empty((N, Kp), dtype=torch.float32, device=A.device)
    nbytes = _sizes("sg_gemm_tn_f32_workspace", M, N, Kp)
    ws = torch.empty((nbytes,), dtype=torch.uint8, device=A.device)
    with _on_device(A.device):
        _check(load().sg_gemm_tn_f32(_ptr(A), _rows2d(A, "A"), _ptr(B), _rows2d(B, "B"), M, N, Kp, _ptr(ws), nbytes, _ptr(out),
                                     Kp, _stream(A)), "sg_gemm_tn_f32")
    return out


def bn_bwd_coeffs(partial: torch.Tensor, count, gamma: torch.Tensor, invstd: torch.Tensor,
                  acc_dweight: Optional[torch.Tensor] = None, acc_dbias: Optional[torch.Tensor] = None) -> torch.Tensor:
    """[5, C] = (sum dz, sum dz*xhat, c1, c2, k) from the partials of bn_act_bwd_reduce.  ``acc_dweight`` / ``acc_dbias``:
    fp32 [C] gradient accumulators (the parameters' .grad), += sum dz*xhat / += sum dz by the same launch.  ``count``:
    the row count N, a number -- or a float32 device scalar (a partition's mesh-wide count, bn_finalize_ranks' second result)."""
    nb, _, C = partial.shape
    count_dev = None
    if isinstance(count, torch.Tensor):
        if count.dtype != torch.float32 or count.numel() != 1 or count.device != partial.device:
            raise SemigcnLibraryError("bn_bwd_coeffs: a device count must be one float32 on the device of the partials")
        count_dev, count = count, 0.0
    out = torch.empty((5, C), dtype=torch.float32, device=partial.device)
    for t, n in ((acc_dweight, "acc_dweight"), (acc_dbias, "acc_dbias")):
        if t is not None:
            _f32vec(t, C, n)
    with _on_device(partial.device):
        _check(load().sg_bn_bwd_coeffs(_ptr(partial), nb, C, float(count), _ptr(_f32vec(gamma, C, "weight")),
                                       _ptr(_f32vec(invstd, C, "invstd")), _ptr(out), _ptr(acc_dweight), _ptr(acc_dbias),
                                       _ptr(count_dev), _stream(partial)), "sg_bn_bwd_coeffs")
    return out


MULTI_ADD_MAX = 8


def multi_add(srcs, dsts) -> None:
    """``dsts[i] += srcs[i]`` for small fp32 matrices / vectors in one launch per 8 (sg_multi_add): ``dsts`` contiguous,
    ``srcs`` of the same shapes with unit inner stride (column or row blocks of a wider matrix are fine)."""
    n_all = len(srcs)
    if n_all != len(dsts):
        raise SemigcnLibraryError("multi_add: one destination per source")
    if n_all == 0:
        return
    f32 = torch.float32
    sp, dp, ld, rows, cols = [], [], [], [], []
    dev = dsts[0].device
    for s, d in zip(srcs, dsts):
        nd = s.dim()
        if s.dtype != f32 or d.dtype != f32 or s.shape != d.shape or s.device != dev or d.device != dev \
                or not d.is_cuda or not d.is_contiguous() or nd not in (1, 2):
            raise SemigcnLibraryError(f"multi_add: need float32 pairs of one shape (1-D or 2-D) on one HIP device with a "
                                      f"contiguous destination, got {s.dtype} {tuple(s.shape)} -> {d.dtype} {tuple(d.shape)}")
        st = s.stride()
        if nd == 1:
            if st[0] != 1 and s.shape[0] > 1:
                s = s.contiguous()
            r, c, l = 1, s.shape[0], s.shape[0]
        else:
            r, c = s.shape
            if st[1] != 1 and c > 1:
                s = s.contiguous()
                st = s.stride()
            l = st[0] if r > 1 else max(st[0], c)
        sp.append(s.data_ptr()); dp.append(d.data_ptr()); ld.append(l); rows.append(r); cols.append(c)
    lib = load()
    stream = _stream(dsts[0])
    with _on_device(dev):
        for at in range(0, n_all, MULTI_ADD_MAX):
            n = min(MULTI_ADD_MAX, n_all - at)
            # one small host array of 5 n 64-bit words: [src pointers | row strides | rows | cols | dst pointers]
            pack = _array("q", sp[at:at + n] + ld[at:at + n] + rows[at:at + n] + cols[at:at + n] + dp[at:at + n])
            base = pack.buffer_info()[0]
            rc = lib.sg_multi_add(n, base, base + 8 * n, base + 16 * n, base + 24 * n, base + 32 * n, stream)
            if rc:
                _check(rc, "sg_multi_add")


def scale_shift_act(X: torch.Tensor, scale: torch.Tensor, shift: torch.Tensor, slope: float,
                    out: Optional[torch.Tensor] = None) -> torch.Tensor:
    _require_device(X, "X")
    V, C = X.shape
    if out is None:
        out = torch.empty((V, C), dtype=X.dtype, device=X.device)
    with _on_device(X.device):
        _check(load().sg_scale_shift_act(_ptr(X), _rows2d(X, "X"), _ptr(_f32vec(scale, C, "scale")),
                                         _ptr(_f32vec(shift, C, "shift")), float(slope), _ptr(out), _rows2d(out, "out"),
                                         V, C, dtype_code(X), _stream(X)), "sg_scale_shift_act")
    return out


def bn_act_bwd_reduce(dA, H, scale, shift, mean, invstd, slope: float) -> torch.Tensor:
    _require_device(dA, "dA")
    V, C = H.shape
    nb = col_blocks(V)
    part = torch.empty((nb, 2, C), dtype=torch.float32, device=H.device)
    with _on_device(H.device):
        _check(load().sg_bn_act_bwd_reduce(_ptr(dA), _rows2d(dA, "dA"), _ptr(H), _rows2d(H, "H"),
                                           _ptr(_f32vec(scale, C, "scale")), _ptr(_f32vec(shift, C, "shift")),
                                           _ptr(_f32vec(mean, C, "mean")), _ptr(_f32vec(invstd, C, "invstd")),
                                           float(slope), _ptr(part), nb, V, C, dtype_code(H), _stream(H)),
               "sg_bn_act_bwd_reduce")
    return part


def bn_act_bwd_apply(dA, H, scale, shift, mean, invstd, k, c1, c2, slope: float,
                     out: Optional[torch.Tensor] = None) -> torch.Tensor:
    _require_device(dA, "dA")
    V, C = H.shape
    dH = torch.empty((V, C), dtype=H.dtype, device=H.device) if out is None else out
    with _on_device(H.device):
        _check(load().sg_bn_act_bwd_apply(_ptr(dA), _rows2d(dA, "dA"), _ptr(H), _rows2d(H, "H"),
                                          _ptr(_f32vec(scale, C, "scale")), _ptr(_f32vec(shift, C, "shift")),
                                          _ptr(_f32vec(mean, C, "mean")), _ptr(_f32vec(invstd, C, "invstd")),
                                          _ptr(_f32vec(k, C, "k")), _ptr(_f32vec(c1, C, "c1")), _ptr(_f32vec(c2, C, "c2")),
                                          float(slope), _ptr(dH), _rows2d(dH, "dH"), V, C, dtype_code(H), _stream(H)),
               "sg_bn_act_bwd_apply")
    return dH


def bn_act_bwd_apply_colsum(dA, H, scale, shift, mean, invstd, k, c1, c2, slope: float,
                            out: Optional[torch.Tensor] = None):
    """bn_act_bwd_apply that also returns the fp32 column sums of the dH it wrote (the bias gradient of the ChebConv in
    front of the BatchNorm), or ``(dH, None)`` when the shape is not served by the row-owning kernel."""
    _require_device(dA, "dA")
    V, C = H.shape
    nb = _sizes("sg_col_apply_blocks", V, C, dtype_code(H))
    dH = torch.empty((V, C), dtype=H.dtype, device=H.device) if out is None else out
    strides_ok = all(_rows2d(t, "operand") % 4 == 0 for t in (dA, H, dH))
    if nb == 0 or not strides_ok:
        return bn_act_bwd_apply(dA, H, scale, shift, mean, invstd, k, c1, c2, slope, out=dH), None
    part = torch.empty((nb, C), dtype=torch.float32, device=H.device)
    sums = torch.empty((C,), dtype=torch.float32, device=H.device)
    with _on_device(H.device):
        _check(load().sg_bn_act_bwd_apply_colsum(_ptr(dA), _rows2d(dA, "dA"), _ptr(H), _rows2d(H, "H"),
                                                 _ptr(_f32vec(scale, C, "scale")), _ptr(_f32vec(shift, C, "shift")),
                                                 _ptr(_f32vec(mean, C, "mean")), _ptr(_f32vec(invstd, C, "invstd")),
                                                 _ptr(_f32vec(k, C, "k")), _ptr(_f32vec(c1, C, "c1")), _ptr(_f32vec(c2, C, "c2")),
                                                 float(slope), _ptr(dH), _rows2d(dH, "dH"), V, C, dtype_code(H), _ptr(part),
                                                 _ptr(sums), _stream(H)), "sg_bn_act_bwd_apply_colsum")
    return dH, sums


# ---- the network's input step ------------------------------------------------------------------
def _prep_args(z1, dm, perm, lo, hi):
    _require_device(z1, "z1")
    V = z1.shape[0]
    if z1.dtype != torch.float32 or z1.dim() != 2 or z1.shape[1] != 3 or not z1.is_contiguous():
        raise SemigcnLibraryError(f"z1 must be contiguous float32 [V, 3], got {z1.dtype} {tuple(z1.shape)}")
    if dm is not None and (dm.dtype != torch.float32 or dm.numel() != V or not dm.is_contiguous() or dm.device != z1.device):
        raise SemigcnLibraryError("dm must be V contiguous float32 values on the device of z1")
    if perm is not None and (perm.dtype != torch.int64 or perm.numel() != V or not perm.is_contiguous()
                             or perm.device != z1.device):
        raise SemigcnLibraryError("order / rank must be V contiguous int64 values on the device of z1")
    for t, n in ((lo, "lo"), (hi, "hi")):
        if t.dtype != torch.float32 or t.numel() != 3 or not t.is_contiguous() or t.device != z1.device:
            raise SemigcnLibraryError(f"{n} must be 3 contiguous float32 values on the device of z1")
    return V


def input_prep(z1: torch.Tensor, dm: Optional[torch.Tensor], order: Optional[torch.Tensor], lo: torch.Tensor,
               hi: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
    """[V, 4] network input in processing order and feature dtype (sg_input_prep): ``(dm (z1 - mid) / extent, dm)``."""
    V = _prep_args(z1, dm, order, lo, hi)
    X = torch.empty((V, 4), dtype=dtype, device=z1.device)
    with _on_device(z1.device):
        _check(load().sg_input_prep(_ptr(z1), _ptr(dm), _ptr(order), _ptr(lo), _ptr(hi), _ptr(X), 4, V, dtype_code(X),
                                    _stream(z1)), "sg_input_prep")
    return X


def input_prep_bwd(gX: torch.Tensor, z1: torch.Tensor, dm: Optional[torch.Tensor], rank: Optional[torch.Tensor],
                   lo: torch.Tensor, hi: torch.Tensor, need_dz1: bool = True):
    """(dz1 [V, 3] or None, d_lo [3], d_hi [3]) from gX = dL/dX [V, >= 3 columns] (sg_input_prep_bwd)."""
    V = _prep_args(z1, dm, rank, lo, hi)
    if gX.shape[0] != V or gX.device != z1.device:
        raise SemigcnLibraryError("gX must have one row per vertex on the device of z1")
    ldg = _rows2d(gX, "gX")
    dz1 = torch.empty((V, 3), dtype=torch.float32, device=z1.device) if need_dz1 else None
    part = torch.empty((_sizes("sg_input_prep_blocks", V), 4), dtype=torch.float32, device=z1.device)
    dlh = torch.empty((2, 3), dtype=torch.float32, device=z1.device)
    with _on_device(z1.device):
        _check(load().sg_input_prep_bwd(_ptr(gX), ldg, _ptr(z1), _ptr(dm), _ptr(rank), _ptr(lo), _ptr(hi), _ptr(dz1),
                                        _ptr(part), dlh.data_ptr(), dlh.data_ptr() + 12, V, dtype_code(gX), _stream(z1)),
               "sg_input_prep_bwd")
    return dz1, dlh[0], dlh[1]


def input_bounds(z1: torch.Tensor):
    """(bounds float32 [6] = lo[3] | hi[3] of z1 over its rows, arg int64 [6] = the vertex each bound came from)."""
    _require_device(z1, "z1")
    V = z1.shape[0]
    if z1.dtype != torch.float32 or z1.dim() != 2 or z1.shape[1] != 3 or not z1.is_contiguous() or V == 0:
        raise SemigcnLibraryError(f"z1 must be contiguous float32 [V > 0, 3], got {z1.dtype} {tuple(z1.shape)}")
    nb = _sizes("sg_input_prep_blocks", V)
    pv = torch.empty((nb + 1, 6), dtype=torch.float32, device=z1.device)       # (last row: the result)
    pi = torch.empty((nb + 1, 6), dtype=torch.int64, device=z1.device)
    with _on_device(z1.device):
        _check(load().sg_input_bounds(_ptr(z1), V, _ptr(pv), _ptr(pi), pv.data_ptr() + 24 * nb, pi.data_ptr() + 48 * nb, _stream(z1)),
               "sg_input_bounds")
    return pv[nb], pi[nb]


def input_prep_bwd_routed(gX: torch.Tensor, z1: torch.Tensor, dm: Optional[torch.Tensor], rank: Optional[torch.Tensor],
                          bounds: torch.Tensor, arg: torch.Tensor) -> torch.Tensor:
    """dz1 [V, 3] from gX = dL/dX, the gradients of the bounds already added at the vertices the bounds came from."""
    V = _prep_args(z1, dm, rank, bounds[:3], bounds[3:])
    if gX.shape[0] != V or gX.device != z1.device or arg.dtype != torch.int64 or arg.numel() != 6 or not arg.is_contiguous():
        raise SemigcnLibraryError("input_prep_bwd_routed: gX must have one row per vertex, arg six int64 values")
    dz1 = torch.empty((V, 3), dtype=torch.float32, device=z1.device)
    part = torch.empty((_sizes("sg_input_prep_blocks", V) + 2, 4), dtype=torch.float32, device=z1.device)
    with _on_device(z1.device):
        _check(load().sg_input_prep_bwd_routed(_ptr(gX), _rows2d(gX, "gX"), _ptr(z1), _ptr(dm), _ptr(rank), _ptr(bounds), _ptr(arg),
                                               _ptr(dz1), _ptr(part), part.data_ptr() + 16 * (part.shape[0] - 2), V, dtype_code(gX),
                                               _stream(z1)), "sg_input_prep_bwd_routed")
    return dz1


# ---- fused loss step ---------------------------------------------------------------------------
def mesh_loss_finalize(partial: torch.Tensor, n_v: float, n_f: float, w_pos: float, k1: float) -> torch.Tensor:
    """float32 [3] = (w_pos sqrt(S_p / n_v + 1e-6) + k1 S_n / n_f, d loss / d S_p, d loss / d S_n) from mesh_loss_fwd's partials."""
    out = torch.empty((3,), dtype=torch.float32, device=partial.device)
    with _on_device(partial.device):
        _check(load().sg_mesh_loss_finalize(_ptr(partial), partial.shape[0], float(n_v), float(n_f), float(w_pos), float(k1), _ptr(out),
                                            _stream(partial)), "sg_mesh_loss_finalize")
    return out


def _f32c(t: torch.Tensor, name: str) -> torch.Tensor:
    _require_device(t, name)
    if t.dtype != torch.float32 or not t.is_contiguous():
        raise SemigcnLibraryError(f"{name} must be contiguous float32")
    return t


def mesh_loss_fwd(pos, faces, target_pos, v_keep, target_fn, f_keep) -> torch.Tensor:
    """float32 [nb, 2] block partials of (sum keep_v |p-t|^2, sum keep_f |n-n_t|_1)."""
    V, F = target_pos.shape[0], faces.shape[0]
    if faces.dtype != torch.int64 or not faces.is_contiguous():
        raise SemigcnLibraryError("faces must be contiguous int64 [F, 3]")
    nb = _sizes("sg_mesh_loss_blocks", V, F)
    part = torch.empty((nb, 2), dtype=torch.float32, device=pos.device)
    with _on_device(pos.device):
        _check(load().sg_mesh_loss_fwd(_ptr(_f32c(pos, "pos")), _ptr(faces), _ptr(_f32c(target_pos, "target_pos")),
                                       _ptr(_f32c(v_keep, "v_keep")), _ptr(_f32c(target_fn, "target_fn")),
                                       _ptr(_f32c(f_keep, "f_keep")), V, F, _ptr(part), _stream(pos)), "sg_mesh_loss_fwd")
    return part


def mesh_loss_bwd(pos, faces, target_pos, v_keep, target_fn, f_keep, g: torch.Tensor) -> torch.Tensor:
    V, F = target_pos.shape[0], faces.shape[0]
    grad = torch.empty_like(pos)
    with _on_device(pos.device):
        _check(load().sg_mesh_loss_bwd(_ptr(_f32c(pos, "pos")), _ptr(faces), _ptr(target_pos), _ptr(v_keep), _ptr(target_fn),
                                       _ptr(f_keep), _ptr(_f32c(g, "g")), V, pos.shape[0], F, _ptr(grad), _stream(pos)),
               "sg_mesh_loss_bwd")
    return grad


def mesh_loss_bwd_det(pos, faces, target_pos, v_keep, target_fn, f_keep, g: torch.Tensor,
                      incidence: "PoolHandle") -> torch.Tensor:
    """mesh_loss_bwd without atomics: per-corner gradients + a CSR sum over ``incidence`` (see face_incidence)."""
    V, F = target_pos.shape[0], faces.shape[0]
    grad = torch.empty_like(pos)
    corner = torch.empty((3 * F, 3), dtype=torch.float32, device=pos.device)
    with _on_device(pos.device):
        _check(load().sg_mesh_loss_bwd_det(_ptr(_f32c(pos, "pos")), _ptr(faces), _ptr(target_pos), _ptr(v_keep),
                                           _ptr(target_fn), _ptr(f_keep), _ptr(_f32c(g, "g")), V, pos.shape[0], F,
                                           incidence._h, _ptr(corner), _ptr(grad), _stream(pos)), "sg_mesh_loss_bwd_det")
    return grad


def face_incidence(faces: torch.Tensor, num_rows: int) -> "PoolHandle":
    """vertex -> incident face corners (corner id = 3 f + i) as an sg_pool, for mesh_loss_bwd_det."""
    F = faces.shape[0]
    corners = torch.arange(3 * F, device=faces.device, dtype=torch.int64)
    return PoolHandle(corners, faces.reshape(-1), 3 * F, int(num_rows))


def mesh_edges(faces: torch.Tensor, num_vertices: int, with_f2f: bool = True):
    """(edges int64 [n,2] in first-meeting order, f2f int64 [F,3] or None, manifold: bool)."""
    _require_device(faces, "faces")
    if faces.dtype != torch.int64 or faces.dim() != 2 or faces.shape[1] != 3:
        raise SemigcnLibraryError(f"faces must be int64 [F,3], got {faces.dtype} {tuple(faces.shape)}")
    faces = faces.contiguous()
    F = faces.shape[0]
    edges = torch.empty((3 * F, 2), dtype=torch.int64, device=faces.device)
    f2f = torch.empty((F, 3), dtype=torch.int64, device=faces.device) if with_f2f else None
    n, manifold = c_int64(0), c_int(1)
    with _on_device(faces.device):
        _check(load().sg_mesh_edges(_ptr(faces), F, int(num_vertices), _ptr(edges), _ptr(f2f), byref(n),
                                    byref(manifold), _stream(faces)), "sg_mesh_edges")
    return edges[: n.value].clone(), f2f, bool(manifold.value)


def face_mask_bits(faces: torch.Tensor, vbits: torch.Tensor) -> torch.Tensor:
    """fbits[f] = AND of the three vertices' words; vbits int64 [V, W] -> int64 [F, W]."""
    _require_device(faces, "faces")
    _require_device(vbits, "vbits")
    if faces.dtype != torch.int64 or vbits.dtype != torch.int64 or vbits.dim() != 2:
        raise SemigcnLibraryError("face_mask_bits: faces int64 [F,3] and vbits int64 [V,W] expected")
    faces, vbits = faces.contiguous(), vbits.contiguous()
    out = torch.empty((faces.shape[0], vbits.shape[1]), dtype=torch.int64, device=faces.device)
    with _on_device(faces.device):
        _check(load().sg_face_mask(_ptr(faces), faces.shape[0], vbits.shape[0], _ptr(vbits), _ptr(out),
                                   vbits.shape[1], _stream(faces)), "sg_face_mask")
    return out


# ---- one [ChebConv -> pool? -> BatchNorm -> activation] block per foreign call ---------------------------------------
def block_planar(blk: sg_block) -> bool:
    """Does the library take this block's T as K planes [K][V][Cin] (ldt = Cin)?  (sg_block_planar: narrow bf16 layers)"""
    r = int(load().sg_block_planar(byref(blk)))
    if r < 0:
        _check(r, "sg_block_planar")
    return r == 1


def block_workspace(blk: sg_block, backward: bool) -> int:
    """Bytes of scratch sg_block_forward / sg_block_backward need for this block (depends on its shape fields only)."""
    n = load().sg_block_workspace(byref(blk), int(backward))       # 0 forward, 1 backward, 2: a partition block (all phases)
    if n < 0:
        _check(int(n), "sg_block_workspace")
    return int(n)


#: host seconds spent inside the two chain entry points since import (tools/host_profile.py reads them): [forward, backward]
chain_host_seconds = [0.0, 0.0]
_PROFILE_CHAINS = os.environ.get("SEMIGCN_PROFILE_CHAINS") == "1"


def block_chain_forward(blks, n: int, stream: int, device: Optional[torch.device] = None) -> None:
    """``blks``: a ctypes array of sg_block (n of them are run, first to last).  ``device``: the device of the blocks' tensors
    -- made current for the call when it is not (the kernels, hipMemcpy2DAsync and the BLAS handle below the ABI all go by
    the CURRENT device; ``stream`` must be a stream of it)."""
    with (_NO_GUARD if device is None else _on_device(device)):
        if _PROFILE_CHAINS:
            import time
            t0 = time.perf_counter()
            rc = _lib.sg_block_chain_forward(blks, n, stream)
            chain_host_seconds[0] += time.perf_counter() - t0
        else:
            rc = _lib.sg_block_chain_forward(blks, n, stream)
    if rc:
        _check(rc, "sg_block_chain_forward")


def block_run(blks, n: int, stream: int, device: Optional[torch.device] = None) -> None:
    """The phases (``blks[i].phase``) of n partition blocks in array order: what a rank does between two collectives."""
    with (_NO_GUARD if device is None else _on_device(device)):
        rc = _lib.sg_block_run(blks, n, stream)
    if rc:
        _check(rc, "sg_block_run")


class Comm:
    """One RCCL communicator of the library's own (``sg_comm``, csrc/comm.hip): the rank's collectives enqueued below the C
    ABI, on the stream of the kernels around them.  ``Comm.unique_id()`` on rank 0 -> the 128 bytes every rank passes to the
    constructor (a collective call: all ranks, each with its device current)."""

    def __init__(self, unique_id: bytes, rank: int, world: int, send_rows, recv_rows, device: torch.device):
        lib = load()
        if len(unique_id) != 128:
            raise SemigcnLibraryError(f"Comm: the unique id has {len(unique_id)} bytes, not 128")
        self.rank, self.world, self.device = rank, world, torch.device(device)
        srows = (c_int64 * world)(*[int(v) for v in send_rows])
        rrows = (c_int64 * world)(*[int(v) for v in recv_rows])
        buf = ctypes.create_string_buffer(bytes(unique_id), 128)
        h = c_void_p()
        with _on_device(self.device):
            _check(lib.sg_comm_create(buf, rank, world, srows, rrows, ctypes.byref(h)), "sg_comm_create")
        self._h = h

    @staticmethod
    def available() -> bool:
        return bool(load().sg_comm_available())

    @staticmethod
    def unique_id() -> bytes:
        buf = ctypes.create_string_buffer(128)
        _check(load().sg_comm_unique_id(buf), "sg_comm_unique_id")
        return buf.raw

    def halo_exchange(self, recv: torch.Tensor, send: torch.Tensor) -> None:
        """``send`` [sum send_rows, C] packed by peer -> ``recv`` [sum recv_rows, C]; on the current stream of ``send``."""
        _require_device(send, "send")
        _require_device(recv, "recv")
        if not (send.is_contiguous() and recv.is_contiguous()) or send.dtype != recv.dtype or send.shape[1:] != recv.shape[1:]:
            raise SemigcnLibraryError("Comm.halo_exchange: contiguous [rows, C] buffers of one dtype and width")
        rb = send.element_size()
        for d in send.shape[1:]:
            rb *= int(d)
        with _on_device(self.device):
            _check(_lib.sg_halo_exchange(self._h, send.data_ptr(), recv.data_ptr(), rb, _stream(send)), "sg_halo_exchange")

    def all_reduce_(self, t: torch.Tensor) -> None:
        _require_device(t, "t")
        if t.dtype != torch.float32 or not t.is_contiguous():
            raise SemigcnLibraryError("Comm.all_reduce_: a contiguous float32 tensor")
        with _on_device(self.device):
            _check(_lib.sg_comm_all_reduce_f32(self._h, t.data_ptr(), t.numel(), _stream(t)), "sg_comm_all_reduce_f32")

    def all_gather(self, out: torch.Tensor, inp: torch.Tensor) -> None:
        _require_device(inp, "inp")
        _require_device(out, "out")
        nb = inp.numel() * inp.element_size()
        if not (inp.is_contiguous() and out.is_contiguous()) or out.numel() * out.element_size() != nb * self.world:
            raise SemigcnLibraryError("Comm.all_gather: contiguous buffers, out = world x inp")
        with _on_device(self.device):
            _check(_lib.sg_comm_all_gather(self._h, inp.data_ptr(), out.data_ptr(), nb, _stream(inp)), "sg_comm_all_gather")

    def share(self, send_rows, recv_rows) -> "Comm":
        """A second handle on the SAME RCCL communicator with other per-peer row counts (``sg_comm_share``): one exchange
        layout per MGCN level without one ncclCommInitRank per level.  Not a collective call."""
        other = object.__new__(Comm)
        other.rank, other.world, other.device = self.rank, self.world, self.device
        srows = (c_int64 * self.world)(*[int(v) for v in send_rows])
        rrows = (c_int64 * self.world)(*[int(v) for v in recv_rows])
        h = c_void_p()
        _check(load().sg_comm_share(self._h, srows, rrows, ctypes.byref(h)), "sg_comm_share")
        other._h = h
        return other

    def close(self) -> None:
        h, self._h = self._h, None
        if h and _lib is not None:
            _lib.sg_comm_destroy(h)


class PartRunError(SemigcnLibraryError):
    """sg_part_run stopped at ``step`` with the steps before it already enqueued: this rank's collective sequence is out of
    step with its peers' -- the job must be aborted (a retry on another communicator would meet peers that are somewhere else
    in the schedule)."""

    def __init__(self, msg: str, step: int):
        super().__init__(msg)
        self.step = step


def part_run(comm: Optional["Comm"], steps, n: int, stream: int, device: Optional[torch.device] = None) -> None:
    """A rank's schedule -- runs of partition-block phases with the collectives between them -- in ONE foreign call."""
    with (_NO_GUARD if device is None else _on_device(device)):
        rc = _lib.sg_part_run(comm._h if comm is not None else None, steps, n, stream)
    if rc:
        step = int(_lib.sg_part_failed_step())
        try:
            _check(rc, "sg_part_run")
        except SemigcnLibraryError as e:
            raise PartRunError(str(e), step) from None


def block_chain_backward(blks, n: int, stream: int, device: Optional[torch.device] = None) -> None:
    with (_NO_GUARD if device is None else _on_device(device)):
        if _PROFILE_CHAINS:
            import time
            t0 = time.perf_counter()
            rc = _lib.sg_block_chain_backward(blks, n, stream)
            chain_host_seconds[1] += time.perf_counter() - t0
        else:
            rc = _lib.sg_block_chain_backward(blks, n, stream)
    if rc:
        _check(rc, "sg_block_chain_backward")


class LaunchTrace:
    """Per-launch HIP-event timing INSIDE the library (sg_trace_*): the aggregations and dense products a block call
    launches, which no Python-side timer can bracket any more.  ``with LaunchTrace(capacity) as t: ...`` then, after a
    device synchronize, ``t.records()`` -> list of dicts (kind "agg" / "nt" / "tn" / "pool", dtype, engine, a, b, c, ms)."""
    KINDS = {0: "agg", 1: "nt", 2: "tn", 3: "pool"}
    ENGINES = {0: "agg", 1: "mfma", 2: "thin", 3: "blas", 4: "split"}

    def __init__(self, capacity: int = 1 << 16, kinds=("agg", "nt", "tn")):
        self.capacity = int(capacity)
        self.mask = sum(1 << k for k, name in self.KINDS.items() if name in kinds)
        self._out = None

    def __enter__(self):
        _check(load().sg_trace_begin(self.capacity, self.mask), "sg_trace_begin")
        return self

    def stop(self):
        """Read the records (call after torch.cuda.synchronize()) and release the events."""
        if self._out is None:
            buf = (sg_trace_record * self.capacity)()
            n = load().sg_trace_read(buf, self.capacity)
            if n < 0:
                _check(int(n), "sg_trace_read")
            self._out = [{"kind": self.KINDS.get(r.kind, r.kind), "dtype": "bfloat16" if r.dtype == SG_BF16 else "float32",
                          "engine": self.ENGINES.get(r.engine, r.engine), "a": r.a, "b": r.b, "c": r.c, "ms": r.ms}
                         for r in buf[:n]]
            load().sg_trace_end()
        return self._out

    def records(self):
        return self.stop()

    def __exit__(self, *exc):
        if exc[0] is not None:
            load().sg_trace_end()
        return False
