"""bf16-STORAGE oracle of the SGCN (BASELINE configs[3]: "bf16 features").

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

The reference computes in fp32 throughout (util/networks.py:63-103).  BASELINE configs[3] stores the per-vertex
features between kernels in bf16 while every sum is taken in fp32 and parameters, loss and output stay fp32.  "Is that
path right?" needs a checker that rounds where the path STORES and nowhere else -- then the two may differ only by
the order of fp32 additions, i.e. by the rare bf16 rounding that falls the other way, not by "a few per cent of
bf16 noise".  This module is the reference's composition (oracle.models.SGCNOracle: same modules, same state-dict
keys) with ``x.bfloat16().float()`` at exactly these points:

  forward   the [V,4] network input; every Chebyshev term Tx1, Tx2 (util/networks.py:42 -> [3P] ChebConv.forward);
            the bf16 copy of the weights the products read; the conv output; the BatchNorm+LeakyReLU output
            (util/networks.py:43-45); the skip Linear's output (util/networks.py:96-99).  The last Linear(16, 3)
            and ``x_pos + x`` stay fp32.
  backward  every gradient ROW that is stored: dL/d(conv output) (what the BatchNorm backward writes), the K blocks
            dL/dTx_k = dOut W_k, the recurrence's in-place updates of those blocks, dL/dx.  Parameter gradients are
            fp32 sums over all vertices of products of stored (bf16) rows.

Layers that narrow (Cout < Cin) may be evaluated product first, aggregation after (Clenshaw's recurrence on
Z_k = x W_k^T: the same polynomial in L^); with bf16 storage the two orders differ at bf16
level, so the order is part of what is stored and ``post_when_narrowing`` selects it.

All tensors here are fp32 tensors whose VALUES are bf16-representable where the path stores bf16; every product and
sum runs in fp32 on them, as the fp32-accumulating kernels do.
"""
from __future__ import annotations

from typing import Callable, Optional, Sequence

import numpy as np
import torch
import torch.nn as nn

from .models import SGCN_WIDTHS, normalise_input
from .pyg_restatement import ChebConv, Sequential


def rb(t: torch.Tensor) -> torch.Tensor:
    """Round to bf16 storage (round-to-nearest-even, what the kernels' stores do) and widen again."""
    return t.bfloat16().float()


class _RoundST(torch.autograd.Function):
    """Storage point: value rounded in forward; the gradient arriving at it is a stored row too."""

    @staticmethod
    def forward(ctx, x):
        return rb(x)

    @staticmethod
    def backward(ctx, g):
        return rb(g)


def round_st(x: torch.Tensor) -> torch.Tensor:
    return _RoundST.apply(x)


class _ChebConvBf16Fn(torch.autograd.Function):
    """ChebConv(K) on bf16-stored rows with fp32 accumulation; ``prop(t)`` applies L^ = -D^-1/2 A D^-1/2 with the
    oracle's gather -> multiply -> scatter_add (symmetric for a mesh, so the backward uses it as its own transpose)."""

    @staticmethod
    def forward(ctx, x, bias, post: bool, bias_to_bf16: bool, prop: Callable, *weights):
        K = len(weights)
        wb = [rb(w) for w in weights]
        b = None if bias is None else (rb(bias) if bias_to_bf16 else bias)
        if not post:
            T = [x]
            if K > 1:
                T.append(rb(prop(T[0])))
            for k in range(2, K):
                T.append(rb(2.0 * prop(T[k - 1]) - T[k - 2]))
            acc = T[0] @ wb[0].t()
            for k in range(1, K):
                acc = acc + T[k] @ wb[k].t()
            out = rb(acc if b is None else acc + b)
            ctx.save_for_backward(*T, *wb)
        else:
            Z = [rb(x @ wb[k].t() + (b if (k == 0 and b is not None) else 0.0)) for k in range(K)]
            # Clenshaw, highest term first; z[k] becomes b_k in place
            for k in range(K - 2, 0, -1):
                Z[k] = rb(Z[k] + 2.0 * prop(Z[k + 1]) - (Z[k + 2] if k + 2 <= K - 1 else 0.0))
            out = Z[0]
            if K > 1:
                out = rb(Z[0] + prop(Z[1]) - (Z[2] if K >= 3 else 0.0))
            ctx.save_for_backward(x, *wb)
        ctx.K, ctx.post, ctx.prop, ctx.has_bias = K, post, prop, bias is not None
        return out

    @staticmethod
    def backward(ctx, dout):
        K, prop = ctx.K, ctx.prop
        dout = rb(dout)
        db = dout.sum(0) if ctx.has_bias else None
        if not ctx.post:
            T, wb = ctx.saved_tensors[:K], ctx.saved_tensors[K:]
            dws = [dout.t() @ T[k] for k in range(K)]
            g = [rb(dout @ wb[k]) for k in range(K)]
            if K == 1:
                dx = g[0]
            else:
                for k in range(K - 2, 0, -1):
                    g[k] = rb(g[k] + 2.0 * prop(g[k + 1]) - (g[k + 2] if k + 2 <= K - 1 else 0.0))
                dx = rb(g[0] + prop(g[1]) - (g[2] if K >= 3 else 0.0))
        else:
            x, wb = ctx.saved_tensors[0], ctx.saved_tensors[1:]
            G = [dout]
            if K > 1:
                G.append(rb(prop(G[0])))
            for k in range(2, K):
                G.append(rb(2.0 * prop(G[k - 1]) - G[k - 2]))
            acc = G[0] @ wb[0]
            for k in range(1, K):
                acc = acc + G[k] @ wb[k]
            dx = rb(acc)
            dws = [G[k].t() @ x for k in range(K)]
        return (dx, db, None, None, None, *dws)


class ChebConvBf16(ChebConv):
    """oracle.pyg_restatement.ChebConv (same parameters, same keys) evaluated with bf16 storage."""

    #: evaluate Cout < Cin layers product first (what the path under test does unless told otherwise)
    post_when_narrowing = True
    #: the conv bias is rounded to bf16 before it is added (True where the product is served by the BLAS library,
    #: which takes the bias in the operand type; the path's own MFMA product adds the fp32 parameter)
    bias_to_bf16 = False

    def forward(self, x, edge_index):
        ei, norm = self.norm(edge_index, x.size(0), x.dtype)

        def prop(t):
            return self.propagate(ei, t, norm)
        K = len(self.lins)
        post = bool(self.post_when_narrowing and K >= 2 and self.out_channels < self.in_channels)
        return _ChebConvBf16Fn.apply(x, self.bias, post, bool(self.bias_to_bf16), prop, *[l.weight for l in self.lins])


class _StoredAct(nn.Module):
    """LeakyReLU whose output is a stored row."""

    def __init__(self, act: nn.Module):
        super().__init__()
        self.act = act

    def forward(self, x):
        return round_st(self.act(x))


class SGCNOracleBf16(nn.Module):
    """oracle.models.SGCNOracle (util/networks.py:9-103) with bf16 feature storage.  ``act`` may be replaced by
    any module with LeakyReLU's call signature (the tests inject one that applies a prescribed sign pattern).
    ``bias_bf16_layers``: indices of blocks whose conv bias enters the product rounded to bf16."""

    def __init__(self, skip: bool = False, post_when_narrowing: bool = True, act: Optional[nn.Module] = None,
                 bias_bf16_layers: Sequence[int] = ()):
        super().__init__()
        h = SGCN_WIDTHS
        self.skip = skip
        act = nn.LeakyReLU() if act is None else act
        self.act = _StoredAct(act)
        blocks = []
        for i in range(13):
            conv = ChebConvBf16(h[i], h[i + 1], K=3)
            conv.post_when_narrowing = post_when_narrowing
            conv.bias_to_bf16 = i in set(bias_bf16_layers)
            mods = [(conv, "x, edge_index -> x"), nn.BatchNorm1d(h[i + 1]), self.act]
            if i == 12:
                mods.append((nn.Linear(h[13], h[14]), "x -> x"))
            blocks.append(Sequential("x, edge_index", mods))
        self.blocks = nn.ModuleList(blocks)
        self.skip_blocks = nn.ModuleList([nn.Linear(2 * h[j + 1], h[j + 1]) for j in range(6)])

    def forward(self, z1, x_pos, edge_index, dm=None):
        if isinstance(dm, np.ndarray):
            dm = torch.from_numpy(dm)
        elif not isinstance(dm, torch.Tensor):
            dm = torch.ones(z1.shape[0], 1)
        x = round_st(normalise_input(z1, dm.to(z1.dtype)))
        kept = []
        for i, blk in enumerate(self.blocks):
            if i >= 8 and self.skip:
                j = 13 - i
                # (the two halves of the concatenated gradient are stored rows as well)
                x = round_st(self.skip_blocks[j](torch.cat([round_st(kept[j]), round_st(x)], dim=1)))
            x = blk(x, edge_index)
            if i <= 5:
                kept.append(x)
        return x_pos + x


class MGCNOracleBf16(nn.Module):
    """oracle.models.MGCNOracle (util/meshnet.py:31-160,212-248,278-318) with bf16 feature storage on every level, as
    ``semigcn_amd.meshnet.MGCN.set_feature_dtype(torch.bfloat16)`` stores them: the [V,4] input, every Chebyshev term, conv
    output and BatchNorm+LeakyReLU output (ChebConvBf16 / _StoredAct, as in SGCNOracleBf16), and in addition the pooled /
    unpooled rows between a conv and its BatchNorm (util/meshnet.py:44-47,106-109), the Dropout outputs (:62,128) and the
    skip Linears' outputs.  The three heads' and the decoder's Linear(., 3) and ``smposs + x`` stay fp32.  Same modules and
    state-dict keys as MGCNOracle.  ``act``: factory of the activation modules (the tests inject a prescribed pattern);
    ``bias_bf16_convs``: names of ChebConvs whose bias enters the product rounded to bf16 (products served by the BLAS
    library, which takes the bias in the operand type)."""

    def __new__(cls, edge_inds, pool_hashes, smposs, K: int = 3, skip: bool = False, drop=(0.0, 0.2, 0.2),
                post_when_narrowing: bool = True, act=None, bias_bf16_convs: Sequence[str] = ()):
        from .models import MGCNOracle

        def conv(cin, cout, K=3):
            c = ChebConvBf16(cin, cout, K=K)
            c.post_when_narrowing = post_when_narrowing
            return c
        inner = (lambda: nn.LeakyReLU()) if act is None else act
        net = MGCNOracle(edge_inds, pool_hashes, smposs, K=K, skip=skip, drop=drop, conv=conv,
                         act=lambda: _StoredAct(inner()), store=round_st)
        mods = dict(net.named_modules())
        for name in bias_bf16_convs:
            mods[name].bias_to_bf16 = True
        return net
