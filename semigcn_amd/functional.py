"""Autograd wrappers around the HIP kernels (forward AND backward call sg_spmm / sg_pool_*).

``cheb_conv`` is the whole of ChebConv.forward [3P torch_geometric 2.2.0] (call
sites util/networks.py:42,49; util/meshnet.py:40-240):

    Tx0 = x, Tx1 = L^ x, Txk = 2 L^ Tx(k-1) - Tx(k-2),  out = sum_k Txk Wk^T + b

The K terms are written side by side into one [V, K*Cin] buffer by the
aggregation kernel (strided output, fused ``2 L^ X - X0`` epilogue), so the K
bias-free ``linear`` calls of the reference collapse into ONE [V,K*Cin] x
[K*Cin,Cout] GEMM (hipBLASLt through torch.addmm; MFMA work stays in the GEMM).
Backward: dT = dOut Wcat (one GEMM), dWcat = dOut^T T (one GEMM), then the
recurrence is unwound with K-1 fused aggregations (L^ is symmetric for a mesh;
otherwise the transposed CSR kept by sg_graph_create is used).
"""
from __future__ import annotations

import contextlib
import os
import weakref
from typing import Optional, Sequence

import torch

from . import capi
from .graph import MeshGraph


def _mm_f32_out(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """a @ b with an fp32 result also for bf16 operands."""
    if a.dtype == torch.float32:
        return a @ b
    try:
        return torch.mm(a, b, out_dtype=torch.float32)
    except (TypeError, RuntimeError):
        return (a @ b).float()


def _bmm_f32_out(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    if a.dtype == torch.float32:
        return torch.bmm(a, b)
    try:
        return torch.bmm(a, b, out_dtype=torch.float32)
    except (TypeError, RuntimeError):
        return torch.bmm(a, b).float()


#: optional per-call timing of the dense products (capi.LaunchTimer: HIP events on the launching stream); bench.py installs
#: one over its timed region.  Keys: (kind, M, N, K, dtype name, engine) with kind "nt" (features x weights: forward and
#: input gradient) or "tn" (weight gradient), engine "mfma" (csrc/gemm_mfma.hip) or "blas" (hipBLASLt through torch)
_gemm_timer = None


def set_gemm_timer(timer) -> None:
    global _gemm_timer
    _gemm_timer = timer


class _timed:
    """``with _timed(key):`` -- records an event pair around the product when a timer is installed, nothing otherwise."""
    __slots__ = ("key", "ev")

    def __init__(self, key):
        self.key, self.ev = key, None

    def __enter__(self):
        if _gemm_timer is not None:
            self.ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            self.ev[0].record()
        return self

    def __exit__(self, *exc):
        if self.ev is not None:
            self.ev[1].record()
            _gemm_timer.records.append((self.key, self.ev[0], self.ev[1]))
        return False


def _tn_own(dout: torch.Tensor, T: torch.Tensor) -> bool:
    """The weight gradient runs on the library's own MFMA kernels: the 128 x 128 kernel for small outputs, the persistent
    256 x 256 ring for the wide layers (where it takes the shape); everything else on the BLAS library."""
    if not (USE_MFMA_GEMM and dout.is_cuda and dout.dtype == torch.bfloat16 and capi.gemm_tn_supported(dout, T)):
        return False
    if dout.shape[1] * T.shape[1] <= MFMA_MAX_WEIGHT_ELEMS:
        return True
    return USE_MFMA_BIG_TILE and capi.gemm_tn_takes_big_tile(dout.shape[0], dout.shape[1], T.shape[1], dout.stride(0), T.stride(0))


def _tn_thin(dout: torch.Tensor, T: torch.Tensor) -> bool:
    return (dout.is_cuda and dout.dim() == 2 and T.dim() == 2 and dout.dtype == T.dtype and T.stride(1) == 1
            and dout.shape[0] == T.shape[0] and _thin_ok(dout, dout.shape[1], T.shape[1]))


#: float32 features: the dense products on the split-bf16 MFMA kernels (csrc/gemm_split.hip) where they take the shape;
#: SEMIGCN_F32_BLAS=1 leaves them with the BLAS library (A/B switch; the block path has SG_TUNE_F32_ENGINE)
USE_SPLIT_F32 = os.environ.get("SEMIGCN_F32_BLAS") != "1"


#: float32 blocks keep the split-bf16 images of their weights between weight updates (sg_block::wsplit / wsplit_t); False: every
#: product splits its weights into scratch, as before round 6 (A/B and test switch)
USE_SPLIT_IMAGES = os.environ.get("SEMIGCN_NO_SPLIT_IMAGES") != "1"


def _tn_split(dout: torch.Tensor, T: torch.Tensor) -> bool:
    return (USE_SPLIT_F32 and dout.is_cuda and dout.dtype == torch.float32 and T.dtype == torch.float32
            and not _tn_thin(dout, T) and capi.gemm_tn_f32_supported(dout, T))


def _nt_split(a: torch.Tensor, n: int, out: Optional[torch.Tensor]) -> bool:
    if not (USE_SPLIT_F32 and a.is_cuda and a.dtype == torch.float32 and a.dim() == 2):
        return False
    if out is not None and (out.dtype != torch.float32 or out.stride(1) != 1 or out.data_ptr() % 16):
        return False
    # (own kernels or the BLAS library: the library's rule, asked where it lives -- sg_gemm_nt_f32_pays; since round 6 products
    #  with few rows run on 128 x 128 tiles and the answer is yes at every size)
    if not capi.gemm_nt_f32_pays(a.shape[0], n, a.shape[1]):
        return False
    return capi.gemm_nt_f32_supported(a, n, None if out is None else out.stride(0))


def weight_grad(dout: torch.Tensor, T: torch.Tensor) -> torch.Tensor:
    if _gemm_timer is None or not dout.is_cuda:
        return _weight_grad(dout, T)
    own = _tn_own(dout, T)
    engine = "mfma" if own else ("thin" if _tn_thin(dout, T) else ("split" if _tn_split(dout, T) else "blas"))
    with _timed(("tn", dout.shape[0], dout.shape[1], T.shape[1], str(dout.dtype).replace("torch.", ""), engine)):
        return _weight_grad(dout, T)


def _weight_grad(dout: torch.Tensor, T: torch.Tensor) -> torch.Tensor:
    """dW = dOut^T T in fp32: a [Cout x V] x [V x K*Cin] product whose reduction runs over ALL
    vertices and whose output is tiny.  hipBLASLt's own choice for that shape leaves most CUs idle
    (measured at V = 1 M: 5.5 ms fp32 / 2.3 ms bf16 for 256x768); cutting V into S slabs, one
    batched GEMM over the slabs and a sum over S runs 2-10x faster (2.7 / 0.45 ms) and is
    deterministic (tools/gemm_bench.py).  bf16 features take the library's own kernel (csrc/gemm_mfma.hip, gemm_tn)."""
    V = dout.shape[0]
    if _tn_own(dout, T):
        return capi.gemm_tn(dout, T)       # own MFMA kernels (transposing LDS reads, slab partials summed in order)
    if _tn_thin(dout, T):
        return capi.thin_tn(dout, T)       # a tiny weight matrix: per-block partial sums, added in block order
    if _tn_split(dout, T):
        return capi.gemm_tn_f32(dout, T)   # float32 on the bf16 matrix cores (exact three-way split), slab partials in order
    S = min(128 if dout.dtype == torch.float32 else 64, V // 4096)
    if S <= 1 or not (dout.is_contiguous() and T.is_contiguous()):
        return _mm_f32_out(dout.t(), T)
    Vs = (V // S) * S
    a = dout[:Vs].view(S, Vs // S, dout.shape[1]).transpose(1, 2)
    b = T[:Vs].view(S, Vs // S, T.shape[1])
    out = _bmm_f32_out(a, b).sum(0)
    if Vs < V:
        out = out + _mm_f32_out(dout[Vs:].t(), T[Vs:])
    return out


#: wide [rows, K*C] buffers handed out by ``bn_act(..., widen=K)`` / the fused BatchNorm backward (``grad_widen``),
#: keyed by the address of their storage.  Values are weak: an entry lives exactly as long as the buffer (which
#: the [:V, :C] view handed to the caller keeps alive), so a recycled address can never match a dead entry.
_wide_buffers: "weakref.WeakValueDictionary[int, torch.Tensor]" = weakref.WeakValueDictionary()


def _new_wide(rows: int, V: int, C: int, K: int, dtype, device) -> torch.Tensor:
    """View [:V, :C] of a fresh, registered [rows, K*C] buffer: block 0 for the caller, blocks 1..K-1 reserved for
    the ChebConv that adopts it (``_adopt_wide``)."""
    base = torch.empty((rows, K * C), dtype=dtype, device=device)
    _wide_buffers[base.untyped_storage().data_ptr()] = base
    return base[:V, :C]


def _adopt_wide(x: torch.Tensor, K: int, rows: Optional[int] = None):
    """If ``x`` is the block-0 view of a [rows, K*C] buffer that ``_new_wide`` handed out (what
    ``bn_act(..., widen=K, rows=...)`` returns; rows = V by default, V + halo rows on a partition), hand back that
    buffer so Tx1..Tx(K-1) are written next to it without a copy.  Only REGISTERED buffers qualify: a tensor that
    merely has the same strides (a column slice of a caller's tensor, the gradient view ``torch.cat`` hands to a
    backward) owns its neighbouring columns and is copied instead."""
    V, C = x.shape
    rows = V if rows is None else rows
    if x.stride(1) != 1 or x.stride(0) != K * C or x.storage_offset() != 0:
        return None
    base = _wide_buffers.get(x.untyped_storage().data_ptr())
    if base is None or base.shape != (rows, K * C) or base.dtype != x.dtype or V > rows:
        return None
    return base


#: [K, rows, C] buffers of K dense planes (``_new_planes``): what a NARROW layer that aggregates first keeps its Tx_k in
#: (include/semigcn.h, sg_block_planar): plane 0 is handed to the caller as an ordinary contiguous [V, C] tensor.
_plane_buffers: "weakref.WeakValueDictionary[int, torch.Tensor]" = weakref.WeakValueDictionary()
#: C (channels per plane) the library takes as planes -- the static part of sg_block_planar's rule; a consumer that does not
#: qualify after all (its products on another engine) reads plane 0 as the plain tensor it is
PLANE_CHANNELS = (8, 16, 32, 64)
USE_PLANES = True


def _planes_wanted(C: int, K: int, dtype) -> bool:
    return USE_PLANES and K > 1 and dtype == torch.bfloat16 and C in PLANE_CHANNELS


def _new_planes(V: int, C: int, K: int, dtype, device) -> torch.Tensor:
    """Plane 0 (a contiguous [V, C] tensor) of a fresh, registered [K, V, C] buffer; planes 1..K-1 are reserved for the
    block that adopts it (``_adopt_planes``)."""
    base = torch.empty((K, V, C), dtype=dtype, device=device)
    _plane_buffers[base.untyped_storage().data_ptr()] = base
    return base[0]


def _adopt_planes(x: torch.Tensor, K: int):
    """The registered [K, V, C] buffer whose plane 0 ``x`` is, or None."""
    if x.dim() != 2 or not x.is_contiguous() or x.storage_offset() != 0:
        return None
    base = _plane_buffers.get(x.untyped_storage().data_ptr())
    if base is None or base.shape != (K, x.shape[0], x.shape[1]) or base.dtype != x.dtype:
        return None
    return base


#: column sums that the kernel which WROTE a tensor has already taken (the fused BatchNorm backward leaves the sums of
#: its dH: the bias gradient of the ChebConv in front of it): id(tensor) -> (weakref to it, its version, fp32 [C] sums).
#: An entry is honoured only for the very same tensor object, unmodified since; at most a handful are kept.
_known_column_sums: dict = {}


def _remember_column_sums(t: torch.Tensor, sums: torch.Tensor) -> None:
    if len(_known_column_sums) > 8:
        _known_column_sums.clear()
    _known_column_sums[id(t)] = (weakref.ref(t), t._version, sums)


def column_sums(x: torch.Tensor) -> torch.Tensor:
    """fp32 column sums of a [V, C] device tensor in ONE streaming pass (the block-moments kernel of
    csrc/bn_act.hip + its merge: mean * V); ATen's column reduction needs ~3x the time at V = 1 M.  Sums the producing
    kernel already took (``_remember_column_sums``) are handed back without touching x."""
    ent = _known_column_sums.pop(id(x), None)
    if ent is not None and ent[0]() is x and ent[1] == x._version:
        return ent[2]
    if not x.is_cuda or x.shape[0] < 4096:
        return x.sum(0, dtype=torch.float32)
    if x.stride(1) != 1 and x.shape[1] > 1:
        x = x.contiguous()
    return capi.bn_merge(capi.col_moments(x), x.shape[0])[0] * float(x.shape[0])


#: > 0 inside ``sink_param_grads()``
_sink_depth = 0

#: False (or SEMIGCN_NO_GRAD_SINKS=1): ``sink_param_grads()`` does nothing -- every gradient goes through autograd (A/B switch)
SINK_PARAM_GRADS = os.environ.get("SEMIGCN_NO_GRAD_SINKS") != "1"


@contextlib.contextmanager
def sink_param_grads():
    """Inside this context (the trainers wrap ``loss.backward()`` in it) every layer of this package ADDS its parameter
    gradients into the parameters' existing fp32 ``.grad`` accumulators itself -- one ``sg_multi_add`` launch per ChebConv,
    none per BatchNorm (its coefficient kernel does it) -- and hands autograd ``None`` for them, instead of one
    AccumulateGrad add (or copy) launch per parameter: ~85 launches less per SGCN iteration, same sums in the same order.
    Parameters whose ``.grad`` is missing (or not a contiguous fp32 tensor) get their gradient through autograd as usual.
    Not for ``torch.autograd.grad`` / parameter hooks: those never see a gradient that was sunk."""
    global _sink_depth
    if not SINK_PARAM_GRADS:
        yield
        return
    _sink_depth += 1
    try:
        yield
    finally:
        _sink_depth -= 1


def _sink(params, grads) -> bool:
    """Add ``grads[i]`` (fp32, or None) into ``params[i].grad`` in one launch if sinking is on and every accumulator
    involved exists and is a contiguous fp32 tensor; else do nothing and return False (the caller returns the gradients)."""
    if not _sink_depth:
        return False
    srcs, dsts = [], []
    for p, g in zip(params, grads):
        if g is None:
            continue
        acc = None if p is None else p.grad
        if acc is None or acc.dtype != torch.float32 or g.dtype != torch.float32 or not acc.is_contiguous() \
                or acc.device != g.device or acc.shape != g.shape:
            return False
        srcs.append(g)
        dsts.append(acc)
    if srcs:
        capi.multi_add(srcs, dsts)
    return True


class _LinearFn(torch.autograd.Function):
    """y = x W^T + b for the [V, C] vertex features with fp32 parameters; the weight gradient (a
    reduction over all V) uses the slab-batched product and the streaming column sum."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        ctx.params = (weight, bias)
        if x.is_cuda and x.dim() == 2 and _thin_ok(x, weight.shape[0], weight.shape[1]) and x.dtype == weight.dtype:
            # (the thin kernels take fp32 parameters: a module whose parameters were moved to bf16 gets them widened here,
            # as _dense_nt does)
            bias32 = None if bias is None else (bias if bias.dtype == torch.float32 else bias.float()).contiguous()
            return capi.thin_nt(x, weight, bias32)
        return torch.addmm(bias, x, weight.t()) if bias is not None else x @ weight.t()

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        dy = dy.contiguous()
        dx = dense_nn(dy, weight) if ctx.needs_input_grad[0] else None
        dw = weight_grad(dy, x.contiguous()).to(weight.dtype) if ctx.needs_input_grad[1] else None
        db = column_sums(dy).to(weight.dtype) if (ctx.has_bias and ctx.needs_input_grad[2]) else None
        if dy.is_cuda and _sink(ctx.params, (dw, db)):
            return dx, None, None
        return dx, dw, db


def linear_vertices(x: torch.Tensor, weight: torch.Tensor, bias) -> torch.Tensor:
    return _LinearFn.apply(x, weight, bias)


class WeightCache:
    """The concatenated (and, for bf16 features, cast) copy of a layer's K weight matrices, rebuilt only when a
    parameter changed (version counter / storage): saves a cat and a cast per call -- ~40 tiny kernels per SGCN
    iteration between optimiser steps.  The cached tensors are never modified in place, so the copy an earlier
    forward saved for its backward stays valid."""

    def __init__(self):
        self._key = None
        self._val = None

    def get(self, kind: str, dtype: torch.dtype, weights, bias, build):
        if weights[0].is_cuda and torch.cuda.is_current_stream_capturing():
            return build()     # inside a hipGraph capture the cat/cast must be part of the graph (replays see new weights)
        key = (kind, dtype, tuple((w.data_ptr(), w._version) for w in weights),
               None if bias is None else (bias.data_ptr(), bias._version))
        if key != self._key:
            self._key, self._val = key, build()
        return self._val


def _wcat(weights, dtype):
    w = weights[0] if len(weights) == 1 else torch.cat(list(weights), dim=1)      # [Cout, K*C]
    return w.to(dtype)


#: bf16 features: the dense products run on the library's own MFMA kernel (csrc/gemm_mfma.hip) wherever its shape
#: rules allow; False sends every product back to the BLAS library (hipBLASLt through torch), e.g. for A/B timing
USE_MFMA_GEMM = True

#: products with a tiny weight matrix (<= 16 x 16: the 4 -> 16 input layer with its K = 12 columns, the 16 -> 3 output
#: layer, their autograd) on the library's thin-product kernels (csrc/thin_gemm.hip); False leaves them with the BLAS library
USE_THIN_GEMM = os.environ.get("SEMIGCN_NO_THIN_GEMM") != "1"


def _thin_ok(a: torch.Tensor, N: int, K: int, out: Optional[torch.Tensor] = None) -> bool:
    return (USE_THIN_GEMM and capi.thin_supported(a, N, K)
            and (out is None or (out.dtype == a.dtype and out.dim() == 2 and out.stride(1) == 1 and out.data_ptr() != a.data_ptr())))


#: The 128-row-tile kernel streams A and C at 45-75 % of the HBM rate, which is what bounds the products with small
#: and medium weight matrices; the few compute-bound ones (K*N > ~100 K: the 256/512-channel layers, where a
#: 256 x 256-tile pipeline reaches 43 % of the MFMA peak against this kernel's 31 %) stay with the BLAS library.
#: Measured per product at V = 1 M in profiles/r02_mfma_gemm_bench.json (tools/mfma_gemm_bench.py).
MFMA_MAX_WEIGHT_ELEMS = 100_000


#: the compute-bound products (K * N above MFMA_MAX_WEIGHT_ELEMS) on the library's persistent 256 x 256-tile kernel
#: (csrc/gemm_mfma256.hip) where it takes the shape; False leaves them with the BLAS library (A/B switch)
USE_MFMA_BIG_TILE = os.environ.get("SEMIGCN_NO_BIG_TILE_GEMM") != "1"


def _mfma_big(a: torch.Tensor, b: torch.Tensor, ldc: int) -> bool:
    return USE_MFMA_BIG_TILE and capi.gemm_nt_takes_big_tile(a.shape[0], b.shape[0], a.shape[1], a.stride(0), b.stride(0), ldc)


def _mfma_ok(a: torch.Tensor, b: torch.Tensor, ldc: int) -> bool:
    if not (USE_MFMA_GEMM and a.is_cuda and a.dtype == torch.bfloat16 and capi.gemm_nt_supported(a, b, ldc)):
        return False
    return b.shape[0] * b.shape[1] <= MFMA_MAX_WEIGHT_ELEMS or _mfma_big(a, b, ldc)


def _wcat_pair(weights, dtype):
    """(Wcat [Cout, K*C], its transposed copy [K*C, Cout] for the input-gradient product on the MFMA kernels -- only
    built for bf16 features, where they serve)."""
    w = _wcat(weights, dtype)
    return w, (w.t().contiguous() if dtype == torch.bfloat16 and w.is_cuda and USE_MFMA_GEMM else None)


def dense_nt(a: torch.Tensor, b: torch.Tensor, bias: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None,
             moments: Optional[dict] = None) -> torch.Tensor:
    """``_dense_nt`` (below), timed per call when a timer is installed (``set_gemm_timer``)."""
    if _gemm_timer is None or not a.is_cuda:
        return _dense_nt(a, b, bias, out, moments)
    ldc = b.shape[0] if out is None else out.stride(0)
    own = _mfma_ok(a, b, ldc) and (out is None or (out.stride(1) == 1 and out.data_ptr() % 16 == 0))
    engine = "mfma" if own else ("thin" if _thin_ok(a, b.shape[0], a.shape[1], out) else
                                 ("split" if _nt_split(a, b.shape[0], out) and b.dtype == torch.float32 else "blas"))
    with _timed(("nt", a.shape[0], b.shape[0], a.shape[1], str(a.dtype).replace("torch.", ""), engine)):
        return _dense_nt(a, b, bias, out, moments)


def _dense_nt(a: torch.Tensor, b: torch.Tensor, bias: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None,
              moments: Optional[dict] = None) -> torch.Tensor:
    """``a @ b.T (+ bias)`` for the [V, C] vertex features: the MFMA kernel for bf16 operands it accepts, else the BLAS
    library.  ``bias`` is the fp32 parameter.  ``moments`` (a dict): on the MFMA path it receives ``"tiles"`` = the
    per-row-tile column (mean, M2) of the result and ``"rows"`` = rows per tile, for the BatchNorm that follows."""
    ldc = b.shape[0] if out is None else out.stride(0)
    if _mfma_ok(a, b, ldc) and (out is None or (out.stride(1) == 1 and out.data_ptr() % 16 == 0)):
        bias32 = None if bias is None else (bias if bias.dtype == torch.float32 else bias.float()).contiguous()
        if moments is not None and not _mfma_big(a, b, ldc):      # (the 256 x 256 kernel leaves no tile moments)
            res, mom = capi.gemm_nt(a, b, bias32, out=out, moments=True)
            moments["tiles"], moments["rows"] = mom, capi.gemm_tile_rows(b.shape[0])
            return res
        return capi.gemm_nt(a, b, bias32, out=out)
    if _thin_ok(a, b.shape[0], a.shape[1], out):        # a tiny weight matrix (K = 12 is no MFMA step): one thread per row
        bias32 = None if bias is None else (bias if bias.dtype == torch.float32 else bias.float()).contiguous()
        return capi.thin_nt(a, b, bias32, out=out)
    if b.dtype == torch.float32 and b.dim() == 2 and _nt_split(a, b.shape[0], out):
        return capi.gemm_nt_f32(a, b, None if bias is None else bias.float().contiguous(), out=out)
    if bias is not None:
        return torch.addmm(bias.to(a.dtype), a, b.t()) if out is None else torch.addmm(bias.to(a.dtype), a, b.t(), out=out)
    return a @ b.t() if out is None else torch.mm(a, b.t(), out=out)


def dense_nn(a: torch.Tensor, w: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``a @ w`` on the BLAS library (the input-gradient product where no transposed weight copy is kept: fp32 features);
    a tiny ``w`` (<= 16 x 16) on the thin-product kernel."""
    thin = a.is_cuda and a.dim() == 2 and w.dim() == 2 and _thin_ok(a, w.shape[1], a.shape[1], out)
    split = (not thin) and w.dim() == 2 and w.dtype == torch.float32 and _nt_split(a, w.shape[1], out)

    def run():
        if thin:
            return capi.thin_nt(a, w.t(), None, out=out)
        if split:
            return capi.gemm_nt_f32(a, w, None, out=out, w_is_kn=True)
        return a @ w if out is None else torch.mm(a, w, out=out)

    if _gemm_timer is None or not a.is_cuda:
        return run()
    with _timed(("nt", a.shape[0], w.shape[1], a.shape[1], str(a.dtype).replace("torch.", ""),
                 "thin" if thin else ("split" if split else "blas"))):
        return run()


class _ChebConvFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, graph: MeshGraph, cache: Optional[WeightCache], moments: Optional[dict], x: torch.Tensor,
                bias: Optional[torch.Tensor], *weights: torch.Tensor):
        K = len(weights)
        V, C = x.shape
        wcat, wcat_t = (_wcat_pair(weights, x.dtype) if cache is None
                        else cache.get("cat", x.dtype, weights, None, lambda: _wcat_pair(weights, x.dtype)))
        if K == 1:
            T = x.contiguous()
        else:
            T = _adopt_wide(x, K)
            fresh = T is None
            if fresh:
                T = torch.empty((V, K * C), dtype=x.dtype, device=x.device)
            blk = [T[:, k * C:(k + 1) * C] for k in range(K)]
            if fresh:
                blk[0].copy_(x)
            graph.aggregate(blk[0], blk[1], alpha=1.0)
            for k in range(2, K):
                graph.aggregate(blk[k - 1], blk[k], alpha=2.0, X0=blk[k - 2], beta=-1.0)
        out = dense_nt(T, wcat, bias, moments=moments)
        ctx.graph, ctx.K, ctx.C = graph, K, C
        ctx.has_bias = bias is not None
        ctx.param_dtype = weights[0].dtype
        ctx.wcat_t = wcat_t
        ctx.params = (bias, *weights)
        ctx.save_for_backward(T, wcat)
        return out

    @staticmethod
    def backward(ctx, dout: torch.Tensor):
        T, wcat = ctx.saved_tensors
        graph, K, C = ctx.graph, ctx.K, ctx.C
        need_x, need_b = ctx.needs_input_grad[3], ctx.needs_input_grad[4]
        db = column_sums(dout).to(ctx.param_dtype) if (ctx.has_bias and need_b) else None   # (may be known from the producer)
        dout = dout.contiguous()
        need_w = any(ctx.needs_input_grad[5:])
        dws = [None] * K
        if need_w:
            dwcat = weight_grad(dout, T).to(ctx.param_dtype)  # [Cout, K*C], reduced over V in fp32
            dws = [dwcat[:, k * C:(k + 1) * C] for k in range(K)]
        dx = None
        if need_x:
            # [V, K*C]; block k = dL/dTx_k before the recurrence is unwound
            dT = dense_nt(dout, ctx.wcat_t) if ctx.wcat_t is not None else dense_nn(dout, wcat)
            if K == 1:
                dx = dT
            else:
                g = [dT[:, k * C:(k + 1) * C] for k in range(K)]
                tr = not graph.symmetric
                # g_k += 2 L^T g_(k+1) - g_(k+2), highest k first, in place (Y may alias X0)
                for k in range(K - 2, 0, -1):
                    x1 = g[k + 2] if k + 2 <= K - 1 else None
                    graph.aggregate(g[k + 1], g[k], alpha=2.0, X0=g[k], beta=1.0, X1=x1, gamma=-1.0,
                                    transpose=tr)
                dx = torch.empty((T.shape[0], C), dtype=dout.dtype, device=dout.device)
                x1 = g[2] if K >= 3 else None
                graph.aggregate(g[1], dx, alpha=1.0, X0=g[0], beta=1.0, X1=x1, gamma=-1.0, transpose=tr)
        if dout.is_cuda and _sink(ctx.params, (db, *dws)):
            return (None, None, None, dx) + (None,) * (K + 1)
        return (None, None, None, dx, db, *dws)


def _wstack_set(weights, bias, dtype):
    """(Wstack [K*Cout, Cin] in the feature dtype, the fp32 bias padded to K*Cout (it rides in on Z_0), the transposed
    copy [Cin, K*Cout] for the input-gradient product on the MFMA kernel or None)."""
    K, Co = len(weights), weights[0].shape[0]
    ws = torch.cat(list(weights), dim=0).to(dtype)
    bk = None
    if bias is not None:
        bk = torch.cat([bias.float(), bias.new_zeros((K - 1) * Co, dtype=torch.float32)])
    wt = ws.t().contiguous() if dtype == torch.bfloat16 and ws.is_cuda and USE_MFMA_GEMM else None
    return ws, bk, wt


class _ChebConvPostFn(torch.autograd.Function):
    """The same operator with the aggregation AFTER the GEMM, for layers that narrow (Cout < Cin):
    L^ is linear, so  sum_k T_k(L^) x W_k^T = sum_k T_k(L^) Z_k  with  Z = x [W_0|..|W_(K-1)]^T,
    evaluated by Clenshaw's recurrence  b_k = Z_k + 2 L^ b_(k+1) - b_(k+2),  out = Z_0 + L^ b_1 - b_2.
    The GEMM has the same FLOPs, but every aggregation is Cout wide instead of Cin wide, and only x
    is kept for backward (not the [V, K*Cin] buffer).  Backward: G = [T_0|T_1|..](L^T) dOut is the
    forward Chebyshev recurrence applied to dOut, then dx = G Wstack, dWstack = G^T x."""

    @staticmethod
    def forward(ctx, graph: MeshGraph, cache: Optional[WeightCache], x, bias, *weights):
        K = len(weights)
        Co = weights[0].shape[0]

        def build():
            return _wstack_set(weights, bias, x.dtype)
        wstack, bias_k, wstack_t = build() if cache is None else cache.get("stack", x.dtype, weights, bias, build)
        x = x if x.stride(1) == 1 else x.contiguous()
        # the bias rides in on Z_0 (coefficient +1 in the recurrence): free in the GEMM epilogue
        Z = dense_nt(x, wstack, bias_k)                              # [V, K*Cout]
        z = [Z[:, k * Co:(k + 1) * Co] for k in range(K)]
        # Clenshaw, in place in Z: after step k, z[k] holds b_k
        for k in range(K - 2, 0, -1):
            x1 = z[k + 2] if k + 2 <= K - 1 else None
            graph.aggregate(z[k + 1], z[k], alpha=2.0, X0=z[k], beta=1.0, X1=x1, gamma=-1.0)
        out = torch.empty((x.shape[0], Co), dtype=x.dtype, device=x.device)
        graph.aggregate(z[1], out, alpha=1.0, X0=z[0], beta=1.0, X1=z[2] if K >= 3 else None, gamma=-1.0)
        ctx.graph, ctx.K, ctx.Co = graph, K, Co
        ctx.has_bias, ctx.param_dtype = bias is not None, weights[0].dtype
        ctx.wstack_t = wstack_t
        ctx.params = (bias, *weights)
        ctx.save_for_backward(x, wstack)
        return out

    @staticmethod
    def backward(ctx, dout):
        x, wstack = ctx.saved_tensors
        graph, K, Co = ctx.graph, ctx.K, ctx.Co
        V = dout.shape[0]
        tr = not graph.symmetric
        db = column_sums(dout).to(ctx.param_dtype) if (ctx.has_bias and ctx.needs_input_grad[3]) else None
        G = _adopt_wide(dout, K)          # the fused BatchNorm backward writes dout into block 0 of a [V, K*Co] buffer
        if G is None:
            dout = dout.contiguous()
            G = torch.empty((V, K * Co), dtype=dout.dtype, device=dout.device)
            G[:, :Co].copy_(dout)
        g = [G[:, k * Co:(k + 1) * Co] for k in range(K)]
        dout = g[0]
        graph.aggregate(g[0], g[1], alpha=1.0, transpose=tr)
        for k in range(2, K):
            graph.aggregate(g[k - 1], g[k], alpha=2.0, X0=g[k - 2], beta=-1.0, transpose=tr)
        dx = None
        if ctx.needs_input_grad[2]:
            dx = dense_nt(G, ctx.wstack_t) if ctx.wstack_t is not None else dense_nn(G, wstack)
        dws = [None] * K
        if any(ctx.needs_input_grad[4:]):
            dwstack = weight_grad(G, x.contiguous()).to(ctx.param_dtype)   # [K*Cout, Cin]
            dws = [dwstack[k * Co:(k + 1) * Co] for k in range(K)]
        if dout.is_cuda and _sink(ctx.params, (db, *dws)):
            return (None, None, dx) + (None,) * (K + 1)
        return (None, None, dx, db, *dws)


#: layers with Cout < Cin aggregate after the GEMM (see _ChebConvPostFn); set False to force the
#: reference's evaluation order everywhere
AGGREGATE_AFTER_GEMM_WHEN_NARROWING = True


def cheb_conv(graph: MeshGraph, x: torch.Tensor, weights: Sequence[torch.Tensor],
              bias: Optional[torch.Tensor] = None, cache: Optional[WeightCache] = None,
              moments: Optional[dict] = None) -> torch.Tensor:
    """ChebConv forward on a prepared graph; ``weights[k]`` is ``lins[k].weight`` [Cout, Cin].  ``cache``: the
    calling layer's WeightCache (optional).  ``moments``: a dict that receives the output's per-row-tile column
    moments when the layer's last step is the MFMA product (see dense_nt) -- the BatchNorm behind it then skips its
    own moments pass; it stays empty otherwise."""
    if x.dim() != 2:
        raise ValueError(f"x must be [V, C], got {tuple(x.shape)}")
    if getattr(graph, "sg_partitioned", False):
        from .dist import dist_cheb_conv
        return dist_cheb_conv(graph, x, weights, bias, cache, moments)
    if x.shape[0] != graph.num_vertices:
        raise ValueError(f"x has {x.shape[0]} rows but the graph has {graph.num_vertices} vertices")
    if AGGREGATE_AFTER_GEMM_WHEN_NARROWING and len(weights) >= 2 and weights[0].shape[0] < weights[0].shape[1]:
        return _ChebConvPostFn.apply(graph, cache, x, bias, *weights)
    return _ChebConvFn.apply(graph, cache, moments, x, bias, *weights)


class _LaplacianFn(torch.autograd.Function):
    """y = alpha * L^ x (+ beta * x0): the bare `propagate` step, differentiable."""

    @staticmethod
    def forward(ctx, graph: MeshGraph, x, alpha: float, x0, beta: float):
        x = x.contiguous() if x.stride(-1) != 1 else x
        y = torch.empty((graph.handle.num_rows, x.shape[1]), dtype=x.dtype, device=x.device)
        graph.aggregate(x, y, alpha=alpha, X0=x0, beta=beta)
        ctx.graph, ctx.alpha, ctx.beta = graph, alpha, beta
        return y

    @staticmethod
    def backward(ctx, dy):
        g = ctx.graph
        dy = dy.contiguous()
        dx = None
        if ctx.needs_input_grad[1]:
            dx = torch.empty((g.handle.num_cols, dy.shape[1]), dtype=dy.dtype, device=dy.device)
            g.aggregate(dy, dx, alpha=ctx.alpha, transpose=not g.symmetric)
        dx0 = ctx.beta * dy if ctx.needs_input_grad[3] else None
        return None, dx, None, dx0, None


def laplacian_apply(graph: MeshGraph, x, alpha: float = 1.0, x0=None, beta: float = 0.0):
    return _LaplacianFn.apply(graph, x, alpha, x0, beta)


class _PoolFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pool: capi.PoolHandle, x, mode: str):
        ctx.pool, ctx.mode = pool, mode
        return pool.pool_mean(x) if mode == "pool" else pool.unpool(x)

    @staticmethod
    def backward(ctx, dy):
        dy = dy.contiguous()
        dx = ctx.pool.pool_mean_bwd(dy) if ctx.mode == "pool" else ctx.pool.unpool_bwd(dy)
        return None, dx, None


def mesh_pool(pool: capi.PoolHandle, x: torch.Tensor) -> torch.Tensor:
    """MeshPool.forward (util/meshnet.py:14-17): cluster mean."""
    return _PoolFn.apply(pool, x, "pool")


def mesh_unpool(pool: capi.PoolHandle, x: torch.Tensor) -> torch.Tensor:
    """MeshUnpool.forward (util/meshnet.py:25-27): gather by pool_hash."""
    return _PoolFn.apply(pool, x, "unpool")


# --------------------------------------------------------------------------------------------
# BatchNorm over the vertex axis + LeakyReLU, fused (csrc/bn_act.hip)
# --------------------------------------------------------------------------------------------
class _BNActFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, running_mean, running_var, training, momentum, eps, slope, group, widen,
                grad_widen=1, rows=0, tile_moments=None, batches_tracked=None):
        """``batches_tracked``: the module's num_batches_tracked when this call is to count (training with running
        statistics) -- incremented by the statistics kernel itself."""
        V, C = x.shape
        if x.stride(1) != 1 and C > 1:
            x = x.contiguous()
        dev = x.device
        w32 = weight if weight.dtype == torch.float32 else weight.float()
        b32 = bias if bias.dtype == torch.float32 else bias.float()
        if training and V > 1 and not ctx_group_active(group):
            if tile_moments is not None:       # left behind by the MFMA product that wrote x: no pass over x needed
                fin = capi.bn_stats_finalize_tiles(tile_moments["tiles"], tile_moments["rows"], V, w32, b32, running_mean,
                                                   running_var, momentum, eps, batches_tracked)
            else:
                fin = capi.bn_stats_finalize(capi.col_moments(x), V, w32, b32, running_mean, running_var, momentum, eps,
                                             batches_tracked)
            mean, invstd, scale, shift = fin[0], fin[1], fin[2], fin[3]
            ctx.N = float(V)
        elif training and ctx_group_active(group):
            # vertex partition: (mean, M2, count) of this rank's rows -> all-gather -> merged and finalised in one
            # kernel; the total row count stays on the device (no host round trip per BatchNorm)
            from . import dist as _d
            import torch.distributed as tdist
            world = tdist.get_world_size(group)
            local = torch.empty((1, 2 * C + 1), dtype=torch.float32, device=dev)
            if tile_moments is not None:       # left behind by the MFMA product that wrote x
                capi.bn_local_stats(tile_moments["tiles"], tile_moments["rows"], V, local)
            else:
                capi.bn_local_stats(capi.col_moments(x), 0, V, local)
            allst = torch.empty((world, 2 * C + 1), dtype=torch.float32, device=dev)
            _d._all_gather_rows(allst, local, group)
            fin, n_dev = capi.bn_finalize_ranks(allst, w32, b32, running_mean, running_var, momentum, eps, batches_tracked)
            mean, invstd, scale, shift = fin[0], fin[1], fin[2], fin[3]
            ctx.N = n_dev
        elif training:      # a single row: nothing to merge
            if batches_tracked is not None:
                batches_tracked.add_(1)
            stats = capi.bn_merge(capi.col_moments(x), V)
            fin = capi.bn_finalize(stats, float(V), w32, b32, running_mean, running_var, momentum, eps)
            mean, invstd, scale, shift = fin[0], fin[1], fin[2], fin[3]
            ctx.N = float(V)
        else:
            mean = running_mean.float()
            invstd = torch.rsqrt(running_var.float() + eps)
            scale = (w32 * invstd).contiguous()
            shift = (b32 - mean * scale).contiguous()
            ctx.N = float(V)
        if widen > 1:     # rows > V: the consumer is a partitioned conv whose buffer also holds the halo rows
            y = _new_wide(max(rows, V), V, C, widen, x.dtype, dev)
        else:
            y = torch.empty((V, C), dtype=x.dtype, device=dev)
        capi.scale_shift_act(x, scale, shift, slope, out=y)
        for obs in bn_act_observers:
            obs(y)
        ctx.save_for_backward(x, scale, shift, mean, invstd, w32)
        ctx.training, ctx.slope, ctx.group, ctx.param_dtype = training, slope, group, weight.dtype
        ctx.grad_widen = grad_widen
        ctx.params = (weight, bias)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, scale, shift, mean, invstd, w32 = ctx.saved_tensors
        if dy.stride(1) != 1 and dy.shape[1] > 1:
            dy = dy.contiguous()
        part = capi.bn_act_bwd_reduce(dy, x, scale, shift, mean, invstd, ctx.slope)
        nothing = (None,) * 12
        if ctx.training and not ctx_group_active(ctx.group):       # one launch: block sums, /N and gamma*invstd
            weight, bias = ctx.params
            # sink_param_grads(): the same launch adds the two parameter gradients into their .grad accumulators
            sunk = bool(_sink_depth) and ctx.needs_input_grad[1] and ctx.needs_input_grad[2] and all(
                p.grad is not None and p.grad.dtype == torch.float32 and p.grad.is_contiguous() and p.grad.device == x.device
                for p in (weight, bias))
            co = capi.bn_bwd_coeffs(part, ctx.N, w32, invstd, weight.grad if sunk else None, bias.grad if sunk else None)
            out = None
            if ctx.grad_widen > 1:
                out = _new_wide(x.shape[0], x.shape[0], x.shape[1], ctx.grad_widen, x.dtype, x.device)
            # dx is the output gradient of the ChebConv in front: its bias gradient = the column sums of dx, taken here
            if FUSE_BIAS_GRAD:
                dx, sums = capi.bn_act_bwd_apply_colsum(dy, x, scale, shift, mean, invstd, co[4], co[2], co[3], ctx.slope,
                                                        out=out)
                if sums is not None:
                    _remember_column_sums(dx, sums)
            else:
                dx = capi.bn_act_bwd_apply(dy, x, scale, shift, mean, invstd, co[4], co[2], co[3], ctx.slope, out=out)
            if sunk:
                return (dx, None, None) + nothing
            return (dx, co[1].to(ctx.param_dtype), co[0].to(ctx.param_dtype)) + nothing
        if ctx.training:
            # vertex partition: this rank's sums (one launch; sunk into the gradient accumulators when those are on) ->
            # all-reduce -> c1, c2, k of the whole mesh from the reduced sums and the device-resident row count
            from . import dist as _d
            import torch.distributed as tdist
            weight, bias = ctx.params
            sunk = bool(_sink_depth) and ctx.needs_input_grad[1] and ctx.needs_input_grad[2] and all(
                p.grad is not None and p.grad.dtype == torch.float32 and p.grad.is_contiguous() and p.grad.device == x.device
                for p in (weight, bias))
            loc = capi.bn_bwd_coeffs(part, 1.0, w32, invstd, weight.grad if sunk else None, bias.grad if sunk else None)
            dgamma = dbeta = None
            if not sunk:
                dbeta, dgamma = loc[0].to(ctx.param_dtype).clone(), loc[1].to(ctx.param_dtype).clone()
            s = loc[:2]                                     # [2, C] contiguous: reduced in place
            _d._all_reduce(s, tdist.ReduceOp.SUM, ctx.group)
            co = capi.bn_bwd_coeffs(s.view(1, 2, -1), ctx.N, w32, invstd)
            k, c1, c2 = co[4], co[2], co[3]
        else:
            s = part.sum(0)                                                            # [2, C]: sum dz, sum dz*xhat
            dbeta, dgamma = s[0].to(ctx.param_dtype), s[1].to(ctx.param_dtype)
            sunk = False
            c1 = torch.zeros_like(scale)
            c2 = c1
            k = scale
        out = None
        if ctx.grad_widen > 1:     # born as block 0 of the conv's [V, K*C] gradient buffer (see _ChebConvPostFn.backward)
            out = _new_wide(x.shape[0], x.shape[0], x.shape[1], ctx.grad_widen, x.dtype, x.device)
        if FUSE_BIAS_GRAD:
            dx, sums = capi.bn_act_bwd_apply_colsum(dy, x, scale, shift, mean, invstd, k, c1, c2, ctx.slope, out=out)
            if sums is not None:
                _remember_column_sums(dx, sums)
        else:
            dx = capi.bn_act_bwd_apply(dy, x, scale, shift, mean, invstd, k, c1, c2, ctx.slope, out=out)
        if sunk:
            return (dx, None, None) + nothing
        if dy.is_cuda and _sink(ctx.params, (dgamma if ctx.needs_input_grad[1] else None,
                                             dbeta if ctx.needs_input_grad[2] else None)):
            return (dx, None, None) + nothing
        return (dx, dgamma, dbeta) + nothing


#: the fused BatchNorm backward also leaves the column sums of its dH (= the bias gradient of the ChebConv in front);
#: False: that layer sums the columns of dH in a pass of its own (A/B switch)
FUSE_BIAS_GRAD = os.environ.get("SEMIGCN_NO_FUSED_BIAS_GRAD") != "1"

#: callables invoked with every fused BN+activation output (sign(y) == sign of the BatchNorm output);
#: an observability hook, e.g. for recording activation patterns -- empty in normal operation
bn_act_observers: list = []


def ctx_group_active(group) -> bool:
    """``group`` is False for a plain (single-device) BatchNorm, else a process group / None = WORLD."""
    if group is False:
        return False
    import torch.distributed as tdist
    if not tdist.is_initialized():
        return False
    from .dist import _solo
    return not _solo(tdist.get_world_size(group))


def bn_act(x: torch.Tensor, bn: torch.nn.BatchNorm1d, slope: float, widen: int = 1,
           grad_widen: int = 1, rows: int = 0, tile_moments: Optional[dict] = None) -> torch.Tensor:
    """``leaky_relu(bn(x), slope)`` with nn.BatchNorm1d semantics (batch statistics and running-stat
    updates in training mode, running statistics in eval mode) in two HIP passes.  ``widen = K``
    returns a view of the first C columns of a fresh [V, K*C] buffer, which the next ChebConv
    adopts as its [Tx0|Tx1|..] buffer instead of copying.  ``tile_moments``: {"tiles": [nt, 2, C] per-row-tile
    (mean, M2) of x, "rows": rows per tile} when the kernel that produced x already took them (dense_nt)."""
    training = bn.training or not bn.track_running_stats
    momentum = 0.0
    counter = None
    if bn.training and bn.track_running_stats:
        if bn.momentum is not None:
            momentum = bn.momentum
            if bn.num_batches_tracked is not None and bn.num_batches_tracked.device == x.device:
                counter = bn.num_batches_tracked       # += 1 inside the statistics kernel: no launch of its own
            elif bn.num_batches_tracked is not None:
                bn.num_batches_tracked.add_(1)
        else:      # cumulative moving average: the count is needed on the host
            if bn.num_batches_tracked is not None:
                bn.num_batches_tracked.add_(1)
            momentum = 1.0 / float(bn.num_batches_tracked)
    group = getattr(bn, "group", False) if getattr(bn, "sg_mesh_wide", False) else False
    rm = bn.running_mean if bn.track_running_stats else None
    rv = bn.running_var if bn.track_running_stats else None
    return _BNActFn.apply(x, bn.weight, bn.bias, rm, rv, training, momentum, bn.eps, float(slope), group, int(widen),
                          int(grad_widen), int(rows), tile_moments, counter)


# --------------------------------------------------------------------------------------------
# the network's input / output steps (csrc/input_prep.hip)
# --------------------------------------------------------------------------------------------
class _InputPrepFn(torch.autograd.Function):
    """``cat([dm (z1 - mid) / extent, dm], 1)[order].to(dtype)`` of SingleScaleGCN.forward (util/networks.py:67-79) in one
    launch; backward in two (dz1 in caller order + the gradients of the bounds ``lo`` / ``hi``, which autograd then routes
    to the arg-extreme vertices through the min / max that produced them)."""

    @staticmethod
    def forward(ctx, z1, lo, hi, dm, order, rank, dtype):
        z1c = z1.contiguous()
        dmc = None if dm is None else dm.reshape(-1).contiguous()
        if lo is None:
            # the bounds are z1's own (one device): two launches (sg_input_bounds), the vertices they came from kept for the
            # backward pass, which routes the bounds' gradients there itself
            bounds, arg = capi.input_bounds(z1c)
            ctx.save_for_backward(z1c, bounds, arg)
            ctx.dm, ctx.rank, ctx.routed = dmc, rank, True
            return capi.input_prep(z1c, dmc, order, bounds[:3], bounds[3:], dtype)
        lo_c, hi_c = lo.reshape(-1).contiguous(), hi.reshape(-1).contiguous()
        ctx.save_for_backward(z1c, lo_c, hi_c)
        ctx.dm, ctx.rank, ctx.lo_shape, ctx.routed = dmc, rank, lo.shape, False
        return capi.input_prep(z1c, dmc, order, lo_c, hi_c, dtype)

    @staticmethod
    def backward(ctx, gX):
        if gX.stride(1) != 1:
            gX = gX.contiguous()
        if ctx.routed:
            z1, bounds, arg = ctx.saved_tensors
            dz1 = capi.input_prep_bwd_routed(gX, z1, ctx.dm, ctx.rank, bounds, arg) if ctx.needs_input_grad[0] else None
            return dz1, None, None, None, None, None, None
        z1, lo, hi = ctx.saved_tensors
        dz1, d_lo, d_hi = capi.input_prep_bwd(gX, z1, ctx.dm, ctx.rank, lo, hi, need_dz1=ctx.needs_input_grad[0])
        return (dz1, d_lo.view(ctx.lo_shape) if ctx.needs_input_grad[1] else None,
                d_hi.view(ctx.lo_shape) if ctx.needs_input_grad[2] else None, None, None, None, None)


def input_prep(z1: torch.Tensor, lo: Optional[torch.Tensor], hi: Optional[torch.Tensor], dm: Optional[torch.Tensor],
               order: Optional[torch.Tensor], rank: Optional[torch.Tensor], dtype: torch.dtype) -> torch.Tensor:
    """The [V, 4] network input in processing order (``order`` / ``rank``: the inverse permutations, or None) and
    feature dtype, from z1 [V, 3] fp32, the bounds ``lo`` / ``hi`` [1, 3] (None: z1's own column minima / maxima, taken by the
    library) and the mask ``dm`` [V, 1] (or None)."""
    return _InputPrepFn.apply(z1, lo, hi, dm, order, rank, dtype)


class _OutputFn(torch.autograd.Function):
    """``base + x[rank]`` (rows back in caller order); backward is the gather by the inverse permutation instead of the
    zero-fill + atomic index_add that autograd derives for index_select."""

    @staticmethod
    def forward(ctx, x, base, rank, order):
        ctx.order = order
        return base + x.index_select(0, rank)

    @staticmethod
    def backward(ctx, g):
        gx = g.index_select(0, ctx.order) if ctx.needs_input_grad[0] else None
        return gx, (g if ctx.needs_input_grad[1] else None), None, None


def output_in_caller_order(x: torch.Tensor, base: torch.Tensor, rank: torch.Tensor, order: torch.Tensor) -> torch.Tensor:
    return _OutputFn.apply(x, base, rank, order)


# --------------------------------------------------------------------------------------------
# fused loss step (csrc/mesh_loss.hip)
# --------------------------------------------------------------------------------------------
_incidence_cache: dict = {}


def _face_incidence(faces: torch.Tensor, num_rows: int) -> capi.PoolHandle:
    """One vertex -> face-corner incidence handle per (faces tensor, row count); a reference to the tensor is
    held so that its id is not recycled."""
    key = (id(faces), num_rows)
    ent = _incidence_cache.get(key)
    if ent is not None and ent[0] is faces and ent[1] == faces._version:
        return ent[2]
    if len(_incidence_cache) > 8:
        _incidence_cache.clear()
    h = capi.face_incidence(faces, num_rows)
    _incidence_cache[key] = (faces, faces._version, h)
    return h


#: the normal term's gradient through per-corner buffers and a fixed-order CSR sum (bit-reproducible, and faster
#: than 9 float atomics per face); False restores the atomic kernel
DETERMINISTIC_LOSS_BACKWARD = True


class _MeshLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pos, faces, target_pos, v_keep, target_fn, f_keep):
        pos = pos.contiguous()
        sums = capi.mesh_loss_fwd(pos, faces, target_pos, v_keep, target_fn, f_keep).double().sum(0).float()
        ctx.save_for_backward(pos, faces, target_pos, v_keep, target_fn, f_keep)
        return sums

    @staticmethod
    def backward(ctx, g):
        pos, faces, target_pos, v_keep, target_fn, f_keep = ctx.saved_tensors
        if DETERMINISTIC_LOSS_BACKWARD:
            grad = capi.mesh_loss_bwd_det(pos, faces, target_pos, v_keep, target_fn, f_keep, g.float().contiguous(),
                                          _face_incidence(faces, pos.shape[0]))
        else:
            grad = capi.mesh_loss_bwd(pos, faces, target_pos, v_keep, target_fn, f_keep, g.float().contiguous())
        return grad, None, None, None, None, None


class _MeshLossScalarFn(torch.autograd.Function):
    """``w_pos * sqrt(S_p / n_v + 1e-6) + k1 * S_n / n_f`` as ONE differentiable scalar (forward: the sums kernel + a one-block
    finalize; backward: the finalize's two derivatives scaled by the incoming gradient feed the gradient kernels)."""

    @staticmethod
    def forward(ctx, pos, faces, target_pos, v_keep, target_fn, f_keep, n_v, n_f, w_pos, k1):
        pos = pos.contiguous()
        out = capi.mesh_loss_finalize(capi.mesh_loss_fwd(pos, faces, target_pos, v_keep, target_fn, f_keep), n_v, n_f, w_pos, k1)
        ctx.save_for_backward(pos, faces, target_pos, v_keep, target_fn, f_keep, out)
        return out[0]

    @staticmethod
    def backward(ctx, g):
        pos, faces, target_pos, v_keep, target_fn, f_keep, out = ctx.saved_tensors
        gs = out[1:] * g
        if DETERMINISTIC_LOSS_BACKWARD and faces.shape[0] > 0:
            grad = capi.mesh_loss_bwd_det(pos, faces, target_pos, v_keep, target_fn, f_keep, gs, _face_incidence(faces, pos.shape[0]))
        else:      # (no faces: nothing is accumulated with atomics)
            grad = capi.mesh_loss_bwd(pos, faces, target_pos, v_keep, target_fn, f_keep, gs)
        return (grad,) + (None,) * 9


_no_faces: dict = {}


def mesh_loss(pos: torch.Tensor, faces: Optional[torch.Tensor], target_pos: torch.Tensor, v_keep: torch.Tensor,
              target_fn: Optional[torch.Tensor], f_keep: Optional[torch.Tensor], n_v: float, n_f: float, w_pos: float = 1.0,
              k1: float = 0.0) -> torch.Tensor:
    """The loss of sgcn.py:130-138 -- ``mask_pos_rec_loss + k1 * mask_norm_rec_loss`` on the positions the network produced --
    as one scalar; with ``faces=None`` one resolution's weighted position term of mgcn.py:138-143 (``w_pos`` = its weight)."""
    if faces is None:
        ent = _no_faces.get(pos.device)
        if ent is None:
            ent = _no_faces[pos.device] = (torch.zeros((0, 3), dtype=torch.int64, device=pos.device),
                                           torch.zeros((0, 3), dtype=torch.float32, device=pos.device),
                                           torch.zeros((0,), dtype=torch.float32, device=pos.device))
        faces, target_fn, f_keep = ent
        n_f, k1 = 0.0, 0.0
    return _MeshLossScalarFn.apply(pos, faces, target_pos, v_keep.reshape(-1), target_fn, f_keep.reshape(-1), float(n_v), float(n_f),
                                   float(w_pos), float(k1))


def mesh_loss_sums(pos: torch.Tensor, faces: torch.Tensor, target_pos: torch.Tensor, v_keep: torch.Tensor,
                   target_fn: torch.Tensor, f_keep: torch.Tensor) -> torch.Tensor:
    """[S_p, S_n] = [sum_kept |pos - target|^2, sum_kept_faces |n(pos) - n_target|_1], differentiable in
    ``pos`` ([V_ext, 3]; rows beyond ``target_pos.shape[0]`` are halo rows that only faces read)."""
    return _MeshLossFn.apply(pos, faces, target_pos, v_keep.reshape(-1), target_fn, f_keep.reshape(-1))


# --------------------------------------------------------------------------------------------
# [ChebConv -> (MeshPool | MeshUnpool)? -> BatchNorm1d -> LeakyReLU] blocks below the C ABI
# (sg_block_chain_forward / sg_block_chain_backward, csrc/block.hip): the unit SingleScaleGCN.forward loops over
# (util/networks.py:83-101) and DownConv / UpConv / the MGCN heads are made of (util/meshnet.py:39-62,105-128,
# 223-245).  The library launches the kernel chain of a whole RUN of consecutive blocks; this side keeps one
# autograd node, three allocations and one ctypes call per run and direction.
# --------------------------------------------------------------------------------------------
#: False (or SEMIGCN_NO_BLOCK_CALLS=1): nn.Sequential runs every module on its own, as before (A/B switch; the two paths
#: launch the same kernels with the same arguments)
USE_BLOCK_CALLS = os.environ.get("SEMIGCN_NO_BLOCK_CALLS") != "1"

#: False (or SEMIGCN_NO_BLOCK_CHAINS=1): every block is a call (and an autograd node) of its own
USE_BLOCK_CHAINS = os.environ.get("SEMIGCN_NO_BLOCK_CHAINS") != "1"

#: blocks run since import (forward, backward) and foreign calls that ran them (forward, backward)
block_calls = [0, 0]
chain_calls = [0, 0]


#: runs of blocks share one call (and keep their activations in one arena until backward) up to this many vertices: the
#: meshes where an iteration is bound by host work; above it every block is a call of its own, freed on its own
CHAIN_MAX_ROWS = 1 << 19


def blocks_enabled() -> bool:
    """The process-wide part of the test "may this block go through sg_block_*": the A/B switches are at their defaults and
    no Python-side launch timer is installed (those bracket single launches)."""
    return (USE_BLOCK_CALLS and USE_MFMA_GEMM and USE_THIN_GEMM and USE_MFMA_BIG_TILE and FUSE_BIAS_GRAD
            and _gemm_timer is None and capi._timer is None)


def chaining_allowed() -> bool:
    """Runs of consecutive blocks may share one call unless something wants to see every block's output."""
    return USE_BLOCK_CHAINS and not bn_act_observers


def _f32_dev(t, dev) -> bool:
    return t is not None and t.dtype == torch.float32 and t.device == dev and t.is_contiguous()


class BlockPlan:
    """What stays the same between calls of one block: its modules, the evaluation order, references to its parameter
    tensors and the packed weight copies per feature dtype."""

    def __init__(self, conv, bn, slope: float, pool=None):
        self.conv, self.bn, self.pool, self.slope = conv, bn, pool, float(slope)
        self.K, self.Cin, self.Cout = conv.K, conv.in_channels, conv.out_channels
        self.order = 1 if (AGGREGATE_AFTER_GEMM_WHEN_NARROWING and self.K >= 2 and self.Cout < self.Cin) else 0
        self.pool_mode = 0 if pool is None else int(pool.sg_pool_mode)
        self._packs: dict = {}         # dtype -> [version sum at the last packing, wpack, wpack_t, wpack32, wpack32_t, bias_k]
        self._static: dict = {}        # (dtype, device) -> bool, for the parameter tensors of `_mark`
        self._mark = None
        self.fingerprint()
        conv.__dict__.setdefault("_block_plans_of", []).append(weakref.ref(self))    # (ChebConv.invalidate_weight_cache finds us)

    def fingerprint(self):
        """Identity and address of every tensor the descriptors point at -- the K weights, the conv bias, the BatchNorm's
        weight / bias / running statistics / batch counter -- and the BatchNorm's momentum and eps: when they are what they
        were, the cached references (and every address derived from them) are current.  (``.to()``, ``load_state_dict`` and
        optimiser steps keep the objects; replacing ANY of them by hand, or changing ``bn.momentum`` / ``bn.eps``, moves the
        mark and the descriptors are bound again -- as the per-module path, which re-reads the module on every call.)"""
        conv, bn = self.conv, self.bn
        bp, bb = bn._parameters, bn._buffers
        g, rm = bp["weight"], bb["running_mean"]
        ws = [lin._parameters["weight"] for lin in conv.lins]
        # (identity of every tensor; the address of three of them -- `.to()` and `.data = ..` move all of a module's storage
        #  together, so the first weight, the BatchNorm weight and the running mean stand for the rest: this runs three to four
        #  times per block and iteration on a host-bound path)
        mark = (*map(id, ws), id(conv._parameters.get("bias")), id(g), id(bp["bias"]), id(rm), id(bb["running_var"]),
                id(bb.get("num_batches_tracked")), ws[0].data_ptr(), g.data_ptr(), None if rm is None else rm.data_ptr(),
                bn.momentum, bn.eps)
        if mark != self._mark:
            self.weights = [lin.weight for lin in conv.lins]
            self.cbias, self.gamma, self.beta = conv.bias, bn.weight, bn.bias
            self.rm, self.rv, self.nbt = bn.running_mean, bn.running_var, bn.num_batches_tracked
            self.param_tuple = (self.cbias, *self.weights, self.gamma, self.beta)
            self._static.clear()
            self._mark = mark
        return mark

    def init_descriptor(self, blk) -> None:
        blk.K, blk.order, blk.Cin, blk.Cout, blk.slope, blk.pool_mode = self.K, self.order, self.Cin, self.Cout, self.slope, self.pool_mode

    def usable(self, x: torch.Tensor) -> bool:
        return x.is_cuda and x.dim() == 2 and self.usable_for(x.dtype, x.device, x.shape[0], x.shape[1])

    def rows_out(self, rows: int) -> int:
        """Rows of the block's output for ``rows`` input rows (a pool between conv and BatchNorm changes them)."""
        if self.pool is None:
            return rows
        h = self.pool._pool()
        return h.n_coarse if self.pool_mode == 1 else h.n_fine

    def usable_for(self, dtype, dev, rows: int, cin: int) -> bool:
        """[rows, cin] device features of a dtype and width the block kernels take, fp32 parameters on that device, a
        BatchNorm with running statistics and a momentum (the reference's) whose statistics are this device's own."""
        if cin != self.Cin or rows < 2:
            return False
        self.fingerprint()
        bn = self.bn
        key = (dtype, dev, bn.momentum is None)
        ok = self._static.get(key)
        if ok is None:
            vec = 4 if dtype == torch.float32 else (8 if dtype == torch.bfloat16 else 0)
            ok = bool(vec) and self.Cout % vec == 0 and self.Cout // vec <= 256 and self.K <= 3 \
                and bn.affine and bn.track_running_stats and bn.momentum is not None and bn.num_features == self.Cout \
                and all(_f32_dev(w, dev) for w in self.weights) and _f32_dev(self.gamma, dev) \
                and _f32_dev(self.beta, dev) and _f32_dev(self.rm, dev) and _f32_dev(self.rv, dev) \
                and (self.cbias is None or _f32_dev(self.cbias, dev))
            self._static[key] = ok
        if not ok:
            return False
        if getattr(bn, "sg_mesh_wide", False) and ctx_group_active(getattr(bn, "group", False)):
            return False
        if self.pool is not None:
            if self.pool._dist is not None:
                return False
            h = self.pool._pool()
            if (h.n_fine if self.pool_mode == 1 else h.n_coarse) != rows or self.rows_out(rows) < 2:
                return False
        return True

    def bind(self, blk, dtype: torch.dtype, dev) -> None:
        """Parameter addresses and the packed-weight buffers for ``dtype`` into the descriptor."""
        for k, w in enumerate(self.weights):
            blk.W[k] = w.data_ptr()
        blk.bias = None if self.cbias is None else self.cbias.data_ptr()
        blk.gamma, blk.beta = self.gamma.data_ptr(), self.beta.data_ptr()
        blk.running_mean, blk.running_var = self.rm.data_ptr(), self.rv.data_ptr()
        nbt = self.nbt
        blk.batches_tracked = nbt.data_ptr() if (nbt is not None and nbt.device == dev) else None
        blk.momentum, blk.eps = float(self.bn.momentum), float(self.bn.eps)
        ent = self._packs.get(dtype)
        if ent is None or ent[1].device != dev:
            n = self.K * self.Cin * self.Cout
            thin = self.order == 0 and capi.thin_shape(self.Cout, self.K * self.Cin)
            ent = [None, torch.empty(n, dtype=dtype, device=dev),
                   torch.empty(n, dtype=dtype, device=dev) if dtype == torch.bfloat16 else None,
                   torch.empty(n, dtype=torch.float32, device=dev) if thin else None,
                   torch.empty(n, dtype=torch.float32, device=dev) if thin else None,
                   torch.empty(self.K * self.Cout, dtype=torch.float32, device=dev) if (self.order == 1 and self.cbias is not None) else None]
            # float32 features: the split-bf16 images of the weights for the forward / input-gradient products, rebuilt by the library
            # only when the weights changed (sg_block::wsplit, wsplit_t) instead of inside each product.  One pair of images per TILE
            # VARIANT (the layout follows the row count's size class, sg_gemm_nt_f32_variant): a module applied to a 5 K and a 50 K
            # mesh in one autograd graph must not have one mesh's forward overwrite what the other's backward still reads
            ent += [{}, {}, {}]          # [6], [7]: variant -> image; [8]: variant -> staleness key of that pair
            self._packs[dtype] = ent
        blk.wpack = ent[1].data_ptr()
        blk.wpack_t = None if ent[2] is None else ent[2].data_ptr()
        blk.wpack32 = None if ent[3] is None else ent[3].data_ptr()
        blk.wpack32_t = None if ent[4] is None else ent[4].data_ptr()
        blk.bias_k = None if ent[5] is None else ent[5].data_ptr()
        blk.wsplit = blk.wsplit_t = None
        if dtype == torch.float32 and USE_SPLIT_IMAGES:
            var = capi.gemm_nt_f32_variant(int(blk.V))
            if var not in ent[8]:
                na, ka = (self.Cout, self.K * self.Cin) if self.order == 0 else (self.K * self.Cout, self.Cin)
                for slot, (nn_, kk_) in ((6, (na, ka)), (7, (ka, na))):
                    nb = capi.gemm_nt_f32_workspace(nn_, kk_) if (nn_ >= 64 and kk_ >= 64 and kk_ % 32 == 0 and nn_ % 4 == 0) else 0
                    ent[slot][var] = torch.empty(nb, dtype=torch.uint8, device=dev) if nb > 0 else None
                ent[8][var] = None
            a, b = ent[6][var], ent[7][var]
            blk.wsplit = None if a is None else a.data_ptr()
            blk.wsplit_t = None if b is None else b.data_ptr()

    def stale(self, dtype: torch.dtype, capturing: bool, rows: int = 0) -> int:
        """1 when the packed copies for ``dtype`` are older than the parameters (version counters: optimiser steps and
        every autograd-visible in-place update bump them) -- the library then rebuilds them in the same call; inside a
        hipGraph capture always (the packing is part of the graph: replays see new weights)."""
        v = 0 if self.cbias is None else self.cbias._version
        for w in self.weights:
            v += w._version
        key = (self._mark, v, capi.tuning_generation[0])
        ent = self._packs[dtype]
        var = capi.gemm_nt_f32_variant(rows) if (rows and dtype == torch.float32 and ent[8]) else None
        if ent[0] != key or capturing or (var is not None and ent[8].get(var, key) != key):
            ent[0] = key
            if var is not None:
                ent[8][var] = key         # (the refresh rebuilds the images of THIS descriptor's variant only)
            return 1
        return 0

    def invalidate(self) -> None:
        """Forget the packed copies' state (after a write through ``.data`` that the version counters do not see)."""
        for ent in self._packs.values():
            ent[0] = None
            for var in ent[8]:
                ent[8][var] = None


def _up256(n: int) -> int:
    return (n + 255) & ~255


class _Layout:
    """Where everything of one run of blocks lives for one input shape: offsets into the activation arena (kept until
    backward), the gradient buffer, and the scratch sizes."""
    __slots__ = ("rows", "off_in", "off_H", "off_stats", "arena_bytes", "ws_fwd", "ws_bwd", "dx_bytes", "off_dW", "off_dvec",
                 "grad_floats", "planar", "ldt")


class BlockChain:
    """A run of consecutive blocks served by ONE foreign call per direction: the descriptor arrays, the cached layouts, and
    what the descriptors were last filled with (a steady-state training loop gets the same buffers from the caching
    allocator iteration after iteration: then a call fills in nothing)."""

    def __init__(self, plans):
        self.plans = list(plans)
        n = len(self.plans)
        self.fwd, self.bwd = (capi.sg_block * n)(), (capi.sg_block * n)()
        for i, p in enumerate(self.plans):
            p.init_descriptor(self.fwd[i])
            p.init_descriptor(self.bwd[i])
        self._layouts: dict = {}
        self._planar: dict = {}
        self._sig = [None, None]        # static part last written into fwd / bwd
        self._dyn = [None, None]        # buffer addresses last written
        self._train = [None, None]
        self._acc = None                # gradient-accumulator addresses last written into bwd

    def takes_planes(self, i: int, dtype: torch.dtype, V: int, Vo: int, graph_h, pool_h) -> bool:
        """Does block i keep its [Tx0 | Tx1 | ..] as K planes at this size (sg_block_planar)?"""
        p = self.plans[i]
        if not (p.order == 0 and _planes_wanted(p.Cin, p.K, dtype)):
            return False
        key = (i, dtype, V, Vo)
        ans = self._planar.get(key)
        if ans is None:
            probe = capi.sg_block()
            p.init_descriptor(probe)
            probe.graph, probe.pool = graph_h, pool_h
            probe.dtype, probe.V, probe.V_out, probe.training = capi._DTYPES[dtype], V, Vo, 1
            ans = self._planar[key] = capi.block_planar(probe)
        return ans

    def layout(self, dtype: torch.dtype, rows, in_place: bool, handles, planes0: bool = False) -> _Layout:
        key = (dtype, rows, in_place, planes0, USE_PLANES)
        lay = self._layouts.get(key)
        if lay is not None:
            return lay
        e = 4 if dtype == torch.float32 else 2
        lay = _Layout()
        lay.rows = rows
        # which blocks keep their recurrence buffer T as planes [K][V][Cin] (ldt = Cin) instead of [V, K*Cin]
        lay.planar = [(planes0 if (i == 0 and in_place) else self.takes_planes(i, dtype, V, Vo, *handles[i]))
                      for i, (V, Vo) in enumerate(rows)]
        lay.ldt = [(p.Cin if pl else p.K * p.Cin) for p, pl in zip(self.plans, lay.planar)]
        lay.off_in, lay.off_H, lay.off_stats, lay.off_dW, lay.off_dvec = [], [], [], [], []
        at = gat = 0
        ws_f = ws_b = dxb = 0
        probe = capi.sg_block()
        for i, (p, (V, Vo)) in enumerate(zip(self.plans, rows)):
            # the block's input buffer: [V, K*Cin] when it aggregates first (block 0 may find its input already inside such
            # a buffer: in_place), the plain [V, Cin] input otherwise (block 0 reads the caller's tensor)
            if (i == 0 and (in_place or p.order == 1)):
                lay.off_in.append(-1)
            else:
                lay.off_in.append(at)
                at += _up256(V * (p.K * p.Cin if p.order == 0 else p.Cin) * e)
            lay.off_H.append(at)
            at += _up256(Vo * p.Cout * e)
            lay.off_stats.append(at)
            at += _up256(4 * p.Cout * 4)
            lay.off_dW.append(gat)
            gat += p.Cout * p.K * p.Cin
            lay.off_dvec.append(gat)
            gat += 6 * p.Cout
            p.init_descriptor(probe)
            probe.graph, probe.pool = handles[i]
            probe.dtype, probe.V, probe.V_out, probe.training = capi._DTYPES[dtype], V, Vo, 1
            ws_f = max(ws_f, capi.block_workspace(probe, False))
            ws_b = max(ws_b, capi.block_workspace(probe, True))
            dxb = max(dxb, _up256(V * p.Cin * e))
        lay.arena_bytes, lay.ws_fwd, lay.ws_bwd, lay.dx_bytes, lay.grad_floats = max(at, 256), max(ws_f, 256), max(ws_b, 256), dxb, gat
        if len(self._layouts) > 16:
            self._layouts.clear()
        self._layouts[key] = lay
        return lay

    def fill_static(self, which: int, lay: _Layout, dtype, dev, graphs, pools) -> bool:
        """Write what does not depend on this call's buffers into the descriptors of direction ``which`` (0 forward, 1
        backward) unless it is what they hold already; True when something was written."""
        marks = tuple(p._mark for p in self.plans)      # (refreshed by cheb_chain on the way here, in this same call)
        sig = (lay, dtype, graphs, tuple(pools), marks)
        old = self._sig[which]
        if old is not None and old[0] is lay and old[1] == dtype and old[2] == graphs and old[3] == sig[3] and old[4] == marks:
            return False
        blks = self.bwd if which else self.fwd
        code = capi._DTYPES[dtype]
        for i, p in enumerate(self.plans):
            blk = blks[i]
            blk.graph = graphs[i].handle._h
            blk.pool = None if pools[i] is None else pools[i]._h
            blk.dtype, blk.V, blk.V_out = code, lay.rows[i][0], lay.rows[i][1]
            p.bind(blk, dtype, dev)
            blk.ws_bytes = lay.ws_bwd if which else lay.ws_fwd
        self._sig[which] = sig
        self._dyn[which] = None
        self._train[which] = None
        return True


_chains: dict = {}


def chain_for(plans) -> BlockChain:
    key = tuple(id(p) for p in plans)
    ent = _chains.get(key)
    if ent is None or any(a is not b for a, b in zip(ent.plans, plans)):
        if len(_chains) > 256:
            _chains.clear()
        ent = _chains[key] = BlockChain(plans)
    return ent


def _grad_acc(p, dev):
    """Address of the parameter's .grad when the call may add into it itself (sink_param_grads() is on and the accumulator
    is a contiguous fp32 tensor of the parameter's shape on the device), else None."""
    g = None if p is None else p.grad
    if g is not None and g.dtype == torch.float32 and g.device == dev and g.shape == p.shape and g.is_contiguous():
        return g.data_ptr()
    return None


class _ChainFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, chain: BlockChain, graphs, x: torch.Tensor, widen: int, *params):
        """``params``: per block (conv bias or None, W_0 .. W_(K-1), BatchNorm weight, BatchNorm bias) -- the tensors autograd
        routes the parameter gradients to; the kernels read them through the descriptors."""
        plans, blks = chain.plans, chain.fwd
        n, dev, dtype = len(plans), x.device, x.dtype
        rows, pools, V = [], [], x.shape[0]
        for p in plans:
            pool = None if p.pool is None else p.pool._pool()
            Vo = V if pool is None else (pool.n_coarse if p.pool_mode == 1 else pool.n_fine)
            rows.append((V, Vo))
            pools.append(pool)
            V = Vo
        rows = tuple(rows)
        p0 = plans[0]
        base = None                       # block 0's [V, K*Cin] (or [K, V, Cin]) buffer when its input was born inside one
        planes0 = False
        if p0.order == 0:
            base = _adopt_wide(x, p0.K) if p0.K > 1 else (x if x.is_contiguous() else None)
            if base is None and p0.K > 1:
                pb = _adopt_planes(x, p0.K)
                if pb is not None and chain.takes_planes(0, dtype, rows[0][0], rows[0][1], graphs[0].handle._h,
                                                         None if pools[0] is None else pools[0]._h):
                    base, planes0 = pb, True
        if base is None and x.stride(1) != 1:
            x = x.contiguous()
        lay = chain._layouts.get((dtype, rows, base is not None, planes0, USE_PLANES))
        if lay is None:
            lay = chain.layout(dtype, rows, base is not None,
                               [(g.handle._h, None if pl is None else pl._h) for g, pl in zip(graphs, pools)], planes0)
        arena = torch.empty(lay.arena_bytes, dtype=torch.uint8, device=dev)
        ws = torch.empty(lay.ws_fwd, dtype=torch.uint8, device=dev)
        pl = plans[-1]
        Vo = rows[-1][1]
        if widen > 1 and _planes_wanted(pl.Cout, widen, dtype):
            y = _new_planes(Vo, pl.Cout, widen, dtype, dev)          # (a consumer that does not take planes reads plane 0 as is)
        elif widen > 1:
            y = _new_wide(Vo, Vo, pl.Cout, widen, dtype, dev)
        else:
            y = torch.empty((Vo, pl.Cout), dtype=dtype, device=dev)
        chain.fill_static(0, lay, dtype, dev, graphs, pools)
        training = tuple(1 if p.bn.training else 0 for p in plans)
        if training != chain._train[0]:
            for i in range(n):
                blks[i].training = training[i]
            chain._train[0] = training
        capturing = torch.cuda.is_current_stream_capturing()
        for i, p in enumerate(plans):
            blks[i].refresh_weights = p.stale(dtype, capturing, int(blks[i].V))
        a0, w0 = arena.data_ptr(), ws.data_ptr()
        dyn = (a0, w0, y.data_ptr(), y.stride(0), x.data_ptr(), x.stride(0), None if base is None else base.data_ptr())
        if dyn != chain._dyn[0]:
            for i, p in enumerate(plans):
                blk = blks[i]
                if i == 0:
                    if base is not None:
                        blk.T = blk.X = dyn[6]
                        blk.ldt = blk.ldx = lay.ldt[0]
                    else:
                        blk.X, blk.ldx = dyn[4], dyn[5]
                        blk.T, blk.ldt = (a0 + lay.off_in[0], lay.ldt[0]) if p.order == 0 else (None, 0)
                elif p.order == 0:          # the block in front wrote its output into the first columns (or plane) of T
                    blk.T = blk.X = a0 + lay.off_in[i]
                    blk.ldt = blk.ldx = lay.ldt[i]
                else:
                    blk.X, blk.ldx, blk.T, blk.ldt = a0 + lay.off_in[i], p.Cin, None, 0
                blk.H, blk.stats = a0 + lay.off_H[i], a0 + lay.off_stats[i]
                if i == n - 1:
                    blk.Y, blk.ldy = dyn[2], dyn[3]
                else:
                    q = plans[i + 1]
                    blk.Y, blk.ldy = a0 + lay.off_in[i + 1], (lay.ldt[i + 1] if q.order == 0 else q.Cin)
                blk.ws = w0
            chain._dyn[0] = dyn
        capi.block_chain_forward(blks, n, capi._stream(x), x.device)
        del ws
        block_calls[0] += n
        chain_calls[0] += 1
        if n == 1:
            for obs in bn_act_observers:
                obs(y)
        ctx.chain, ctx.graphs, ctx.pools, ctx.lay, ctx.training = chain, graphs, pools, lay, training
        ctx.ldx0 = x.stride(0)
        ctx.params = params
        ctx.save_for_backward(base if base is not None else x, arena)
        return y

    @staticmethod
    def backward(ctx, dy: torch.Tensor):
        chain, lay = ctx.chain, ctx.lay
        plans, blks = chain.plans, chain.bwd
        x0, arena = ctx.saved_tensors        # x0: block 0's [V, K*Cin] buffer when it was adopted, else the chain's input
        n, dev, dtype = len(plans), arena.device, dy.dtype
        if dy.stride(1) != 1 or dy.stride(0) % 8 or dy.data_ptr() % 16:
            dy = dy.contiguous()
        need_dx = ctx.needs_input_grad[2]
        V0, p0 = lay.rows[0][0], plans[0]
        dx = torch.empty((V0, p0.Cin), dtype=dtype, device=dev) if need_dx else None
        out = torch.empty(max(lay.grad_floats, 1), dtype=torch.float32, device=dev)
        ws = torch.empty(lay.ws_bwd + 2 * lay.dx_bytes + 256, dtype=torch.uint8, device=dev)
        chain.fill_static(1, lay, dtype, dev, ctx.graphs, ctx.pools)
        if ctx.training != chain._train[1]:
            for i in range(n):
                blks[i].training = ctx.training[i]
                blks[i].refresh_weights = 0        # (the forward call brought the packed copies up to date)
            chain._train[1] = ctx.training
        a0, w0, o0 = arena.data_ptr(), ws.data_ptr(), out.data_ptr()
        dyn = (a0, w0, o0, dy.data_ptr(), dy.stride(0), None if dx is None else dx.data_ptr(), x0.data_ptr(), ctx.ldx0, need_dx)
        if dyn != chain._dyn[1]:
            pp = (w0 + lay.ws_bwd, w0 + lay.ws_bwd + lay.dx_bytes)       # input gradients between the blocks: two buffers in turn
            in_place = lay.off_in[0] < 0 and p0.order == 0
            for i, p in enumerate(plans):
                blk = blks[i]
                if p.order == 0:
                    blk.T, blk.ldt, blk.X, blk.ldx = (dyn[6] if (i == 0 and in_place) else a0 + lay.off_in[i]), lay.ldt[i], None, 0
                elif i == 0:
                    blk.T, blk.ldt, blk.X, blk.ldx = None, 0, dyn[6], dyn[7]
                else:
                    blk.T, blk.ldt, blk.X, blk.ldx = None, 0, a0 + lay.off_in[i], p.Cin
                blk.H, blk.stats = a0 + lay.off_H[i], a0 + lay.off_stats[i]
                if i == n - 1:
                    blk.dY, blk.lddy = dyn[3], dyn[4]
                else:
                    blk.dY, blk.lddy = pp[(i + 1) & 1], p.Cout
                if i == 0:
                    blk.need_dx = 1 if need_dx else 0
                    blk.dX, blk.lddx = (dyn[5], p.Cin) if need_dx else (None, 0)
                else:
                    blk.need_dx, blk.dX, blk.lddx = 1, pp[i & 1], p.Cin
                blk.dW, blk.dvec = o0 + 4 * lay.off_dW[i], o0 + 4 * lay.off_dvec[i]
                blk.ws = w0
            chain._dyn[1] = dyn
        # sink_param_grads(): the call adds the gradients into the parameters' .grad accumulators itself
        sunk, accs = [], []
        sinking = bool(_sink_depth)
        for p in plans:
            if sinking:
                aw = [_grad_acc(w, dev) for w in p.weights]
                sw = all(a is not None for a in aw)
                ab = _grad_acc(p.cbias, dev)
                ag, at = _grad_acc(p.gamma, dev), _grad_acc(p.beta, dev)
                sb = ag is not None and at is not None
                accs.append((tuple(aw) if sw else None, ab, (ag, at) if sb else None))
            else:
                sw, ab, sb = False, None, False
                accs.append((None, None, None))
            sunk.append((sw, ab is not None, sb))
        if accs != chain._acc:
            for i, p in enumerate(plans):
                blk = blks[i]
                aw, ab, gb = accs[i]
                for k in range(3):
                    blk.acc_W[k] = aw[k] if (aw is not None and k < p.K) else None
                blk.acc_bias = ab
                blk.acc_gamma, blk.acc_beta = gb if gb is not None else (None, None)
            chain._acc = accs
        capi.block_chain_backward(blks, n, capi._stream(dy), dy.device)
        del ws
        block_calls[1] += n
        chain_calls[1] += 1
        grads = []
        for i, p in enumerate(plans):
            K, Cin, Cout = p.K, p.Cin, p.Cout
            sw, sbias, sbn = sunk[i]
            if sw and sbn and (sbias or p.cbias is None):
                grads.extend([None] * (3 + K))
                continue
            dvec = out[lay.off_dvec[i]:lay.off_dvec[i] + 6 * Cout].view(6, Cout)
            grads.append(None if (p.cbias is None or sbias) else dvec[5].to(p.cbias.dtype))
            if sw:
                grads.extend([None] * K)
            else:
                dW = out[lay.off_dW[i]:lay.off_dW[i] + Cout * K * Cin]
                if p.order == 0:
                    dWm = dW.view(Cout, K * Cin)
                    grads.extend(dWm[:, k * Cin:(k + 1) * Cin] for k in range(K))
                else:
                    dWm = dW.view(K * Cout, Cin)
                    grads.extend(dWm[k * Cout:(k + 1) * Cout] for k in range(K))
            grads.extend((None, None) if sbn else (dvec[1].to(p.gamma.dtype), dvec[0].to(p.gamma.dtype)))
        return (None, None, dx, None, *grads)


def cheb_chain(plans, graphs, x: torch.Tensor, widen: int = 1) -> torch.Tensor:
    """``act(bn(pool?(conv(.))))`` of a run of blocks in one foreign call per direction; ``graphs[i]``: the MeshGraph block
    i's ChebConv runs on; ``widen``: as in ``bn_act``, for the last block's output."""
    if x.shape[0] != graphs[0].num_vertices:
        raise ValueError(f"x has {x.shape[0]} rows but the graph has {graphs[0].num_vertices} vertices")
    if len(plans) > 1 and (x.shape[0] > CHAIN_MAX_ROWS or not chaining_allowed()):
        for i, (p, g) in enumerate(zip(plans, graphs)):      # one call per block: activations are freed block by block
            nxt = plans[i + 1] if i + 1 < len(plans) else None
            x = cheb_chain([p], [g], x, widen if nxt is None else (nxt.K if nxt.order == 0 else 1))
        return x
    params = []
    for p in plans:
        p.fingerprint()
        params.extend(p.param_tuple)
    return _ChainFn.apply(chain_for(plans), tuple(graphs), x, int(widen), *params)


def cheb_block(plan: BlockPlan, graph: MeshGraph, x: torch.Tensor, widen: int = 1) -> torch.Tensor:
    return cheb_chain([plan], [graph], x, widen)
