"""Operator tier of the drop-in boundary: what the reference imports from
``torch_geometric.nn`` (util/networks.py:4, util/meshnet.py:6) -- ``ChebConv``,
``GCNConv`` (name only) and ``Sequential`` -- with the same constructor
arguments, parameter names/shapes (``lins.{k}.weight`` [Cout, Cin], ``bias``
[Cout]), child naming (``module_{i}``) and call signatures, running on the HIP
kernels.  ``semigcn_amd.compat.install()`` makes ``import torch_geometric``
resolve here so the reference's scripts run unmodified.
"""
from __future__ import annotations

import math
from typing import List, Optional, Sequence, Tuple, Union

import torch
import torch.nn as nn
from torch import Tensor

from . import functional as F_sg
from .graph import MeshGraph, graph_for


class _GlorotLinear(nn.Module):
    """Bias-free weight holder equal to torch_geometric's ``Linear(in, out, bias=False,
    weight_initializer='glorot')``: ``weight`` [out, in] ~ U(-a, a), a = sqrt(6/(in+out))."""

    def __init__(self, in_channels: int, out_channels: int):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.weight = nn.Parameter(torch.empty(out_channels, in_channels))
        self.reset_parameters()

    def reset_parameters(self):
        bound = math.sqrt(6.0 / (self.in_channels + self.out_channels))
        nn.init.uniform_(self.weight, -bound, bound)

    def forward(self, x: Tensor) -> Tensor:
        return x @ self.weight.t()

    def extra_repr(self):
        return f"{self.in_channels}, {self.out_channels}, bias=False"


class ChebConv(nn.Module):
    """Chebyshev spectral graph convolution, ``ChebConv(in, out, K, normalization='sym',
    bias=True)``; ``forward(x[V,Cin], edge_index[2,E]) -> [V,Cout]``.

    The scaled Laplacian of ``edge_index`` is prepared once (graph cache) instead
    of once per call; ``forward`` also accepts an already prepared ``MeshGraph``
    in place of ``edge_index``."""

    def __init__(self, in_channels: int, out_channels: int, K: int, normalization: Optional[str] = "sym",
                 bias: bool = True, **kwargs):
        super().__init__()
        if K <= 0:
            raise ValueError("K must be positive")
        if normalization != "sym":
            raise NotImplementedError("only normalization='sym' (the reference's setting) is implemented")
        self.in_channels, self.out_channels, self.K = in_channels, out_channels, K
        self.normalization = normalization
        self.lins = nn.ModuleList([_GlorotLinear(in_channels, out_channels) for _ in range(K)])
        if bias:
            self.bias = nn.Parameter(torch.empty(out_channels))
        else:
            self.register_parameter("bias", None)
        self.reset_parameters()

    def reset_parameters(self):
        for lin in self.lins:
            lin.reset_parameters()
        if self.bias is not None:
            nn.init.zeros_(self.bias)

    def invalidate_weight_cache(self) -> None:
        """Drop the cached concatenated / cast copy of ``lins[k].weight``.  The cache follows the parameters' version
        counters, which optimiser steps and every autograd-visible in-place update bump; writes through ``.data``
        (``p.data.copy_(ema)``, manual clipping) do NOT bump them -- call this after such a write."""
        self.__dict__.pop("_weight_cache", None)
        for ref in self.__dict__.get("_block_plans_of", ()):       # the packed copies of the block calls (functional.BlockPlan)
            plan = ref()
            if plan is not None:
                plan.invalidate()

    def _load_from_state_dict(self, *args, **kwargs):
        self.invalidate_weight_cache()
        return super()._load_from_state_dict(*args, **kwargs)

    def __getstate__(self):
        """Pickling / ``copy.deepcopy`` / ``torch.save(model)``: the caches of the block calls (weak references to plans,
        packed weight copies on the device) belong to THIS object and are rebuilt on the copy's first call."""
        state = dict(self.__dict__)
        state.pop("_weight_cache", None)
        state.pop("_block_plans_of", None)
        return state

    def _apply(self, fn, *args, **kwargs):
        self.invalidate_weight_cache()
        return super()._apply(fn, *args, **kwargs)

    def input_buffer_blocks(self) -> int:
        """K when this layer evaluates [Tx0|..|Tx(K-1)] next to its input (so a producer may write the
        input straight into the first block of a [V, K*Cin] buffer), 1 when it aggregates after the GEMM."""
        post = F_sg.AGGREGATE_AFTER_GEMM_WHEN_NARROWING and self.K >= 2 and self.out_channels < self.in_channels
        return 1 if (post or self.K == 1) else self.K

    def grad_buffer_blocks(self) -> int:
        """K when the backward pass runs the Chebyshev recurrence on the OUTPUT gradient (aggregate-after-GEMM
        layers): the producer of that gradient may then write it straight into block 0 of a [V, K*Cout] buffer."""
        post = F_sg.AGGREGATE_AFTER_GEMM_WHEN_NARROWING and self.K >= 2 and self.out_channels < self.in_channels
        return self.K if post else 1

    def forward(self, x: Tensor, edge_index: Union[Tensor, MeshGraph], edge_weight=None, batch=None,
                lambda_max=None, moments: Optional[dict] = None) -> Tensor:
        """``moments`` (not part of the torch_geometric signature; used by ``Sequential``): a dict that receives the
        output's per-row-tile column moments when the MFMA product emits them (functional.dense_nt)."""
        if edge_weight is not None or batch is not None or lambda_max is not None:
            raise NotImplementedError("edge_weight / batch / lambda_max are not used by the reference "
                                      "and are not implemented")
        prepared = isinstance(edge_index, MeshGraph) or getattr(edge_index, "sg_partitioned", False)
        graph = edge_index if prepared else graph_for(edge_index, x.shape[0])
        cache = self.__dict__.get("_weight_cache")
        if cache is None:
            cache = self.__dict__["_weight_cache"] = F_sg.WeightCache()
        return F_sg.cheb_conv(graph, x, [lin.weight for lin in self.lins], self.bias, cache, moments)

    def __repr__(self):
        return (f"{self.__class__.__name__}({self.in_channels}, {self.out_channels}, K={self.K}, "
                f"normalization={self.normalization})")


class GCNConv(nn.Module):
    """Importable by name only: every ``conv == "gcnconv"`` branch of the reference is
    dead code (util/networks.py:13,21-37; util/meshnet.py:36,64-90,131-158)."""

    def __init__(self, *args, **kwargs):
        super().__init__()
        raise NotImplementedError("GCNConv is never instantiated by SeMIGCN (conv is hard-wired to "
                                  "'chebconv'); only ChebConv is implemented")


def _signature(desc: str) -> Tuple[List[str], List[str]]:
    if "->" not in desc:
        raise ValueError(f"bad Sequential signature {desc!r}")
    lhs, rhs = desc.split("->")
    return [a.strip() for a in lhs.split(",") if a.strip()], [a.strip() for a in rhs.split(",") if a.strip()]


class Sequential(nn.Module):
    """``Sequential(input_args, [(module, "x, edge_index -> x"), module, ...])``.

    Children are registered as ``module_0 .. module_{n-1}`` (so reference
    checkpoints' keys such as ``blocks.3.module_0.lins.1.weight`` load); an entry
    without a signature string maps the previous entry's outputs to themselves."""

    def __init__(self, input_args: str, modules: Sequence[Union[nn.Module, Tuple[nn.Module, str]]]):
        super().__init__()
        self._args = [a.strip() for a in input_args.split(",") if a.strip()]
        self._plan: List[Tuple[str, List[str], List[str]]] = []
        last_out: Optional[List[str]] = None
        for i, entry in enumerate(modules):
            if isinstance(entry, (tuple, list)):
                module, desc = entry
                ins, outs = _signature(desc)
            else:
                module = entry
                if last_out is None:
                    raise ValueError("the first entry of Sequential needs an explicit signature")
                ins, outs = list(last_out), list(last_out)
            name = f"module_{i}"
            if isinstance(module, nn.Module):
                self.add_module(name, module)
            else:
                object.__setattr__(self, name, module)
            self._plan.append((name, ins, outs))
            last_out = outs

    #: set by a model when this Sequential's output feeds a ChebConv(K) directly (see functional.bn_act)
    out_widen = 1

    def _fusable_at(self, i: int):
        """(slope, widen, grad_widen) when entries i, i+1 are BatchNorm1d -> LeakyReLU/ReLU on one variable."""
        if i + 1 >= len(self._plan):
            return None
        (n0, in0, out0), (n1, in1, out1) = self._plan[i], self._plan[i + 1]
        bn, act = getattr(self, n0), getattr(self, n1)
        if not (isinstance(bn, nn.BatchNorm1d) and bn.affine and len(in0) == 1 and in0 == out0 == in1 == out1):
            return None
        if isinstance(act, nn.LeakyReLU):
            slope = act.negative_slope
        elif isinstance(act, nn.ReLU):
            slope = 0.0
        else:
            return None
        widen = 1
        if i + 2 < len(self._plan):
            nxt = getattr(self, self._plan[i + 2][0])
            if isinstance(nxt, ChebConv) and nxt.in_channels == bn.num_features \
                    and self._plan[i + 2][1][:1] == out1:
                widen = nxt.input_buffer_blocks()
        else:
            widen = self.out_widen
        grad_widen = 1       # the conv in front takes its output gradient as block 0 of a [V, K*Cout] buffer?
        if i >= 1:
            prv = getattr(self, self._plan[i - 1][0])
            if isinstance(prv, ChebConv) and prv.out_channels == bn.num_features and self._plan[i - 1][2] == in0:
                grad_widen = prv.grad_buffer_blocks()
        return slope, widen, grad_widen

    def __getstate__(self):
        """(see ChebConv.__getstate__: block plans and the leading-block look-up are per-object caches)"""
        state = dict(self.__dict__)
        state.pop("_block_plans", None)
        state.pop("_lead", None)
        return state

    def _block_at(self, i: int):
        """(BlockPlan, index of its activation entry, widen) when entries i.. read ChebConv [-> MeshPool | MeshUnpool] ->
        BatchNorm1d -> (Leaky)ReLU on ONE variable -- a block the library runs in one call per direction
        (functional.cheb_block); None otherwise.  Looked up once per entry (and per evaluation-order switch)."""
        key = (i, F_sg.AGGREGATE_AFTER_GEMM_WHEN_NARROWING, self.out_widen)
        plans = self.__dict__.setdefault("_block_plans", {})
        if key in plans:
            return plans[key]
        found = None
        name, ins, outs = self._plan[i]
        conv = getattr(self, name)
        if isinstance(conv, ChebConv) and len(ins) == 2 and len(outs) == 1 and ins[0] == outs[0] and conv.K <= 3:
            var, j, pool = outs, i + 1, None
            if j < len(self._plan):
                nxt = getattr(self, self._plan[j][0])
                if getattr(nxt, "sg_pool_mode", 0) and self._plan[j][1] == var and self._plan[j][2] == var:
                    pool, j = nxt, j + 1
            fused = self._fusable_at(j) if j + 1 < len(self._plan) else None
            if fused is not None and self._plan[j][1] == var:
                bn = getattr(self, self._plan[j][0])
                if bn.num_features == conv.out_channels:
                    found = (F_sg.BlockPlan(conv, bn, fused[0], pool), j + 1, fused[1])
        plans[key] = found
        return found

    def _leading_blocks(self):
        """(plans, index of the first entry behind them, widen of the last one) when the entries from 0 on are >= 1
        consecutive blocks on the Sequential's first argument with its second argument as the graph; None otherwise.
        Used by the models to run blocks of SEVERAL Sequentials in one call (functional.cheb_chain)."""
        key = (F_sg.AGGREGATE_AFTER_GEMM_WHEN_NARROWING, self.out_widen)
        hit = self.__dict__.get("_lead")
        if hit is not None and hit[0] == key:
            return hit[1]
        lead = self._find_leading_blocks()
        self.__dict__["_lead"] = (key, lead)
        return lead

    def _find_leading_blocks(self):
        if len(self._args) != 2:
            return None
        xvar, gvar = self._args
        plans, i, widen = [], 0, 1
        while i < len(self._plan):
            _, ins, outs = self._plan[i]
            blk = self._block_at(i) if (ins == [xvar, gvar] and outs == [xvar]) else None
            if blk is None:
                break
            plans.append(blk[0])
            i, widen = blk[1] + 1, blk[2]
        return (plans, i, widen) if plans else None

    def forward(self, *args, _start: int = 0, **kwargs):
        """``_start`` (not part of the torch_geometric signature): run the entries from that index on -- the models use it
        after they have run this Sequential's leading blocks as part of a longer chain."""
        scope = dict(zip(self._args, args))
        scope.update(kwargs)
        result = args[0] if (_start and args) else None
        pending = None             # (conv output, its tile moments) from the ChebConv just run, for the BatchNorm behind it
        i, n = _start, len(self._plan)
        blocks_on = F_sg.blocks_enabled()
        while i < n:
            name, ins, outs = self._plan[i]
            vals = [scope[a] for a in ins]
            on_dev = len(vals) >= 1 and torch.is_tensor(vals[0]) and vals[0].is_cuda and vals[0].dim() == 2
            if on_dev and len(vals) == 2 and blocks_on and getattr(vals[1], "phases", False) and self.training:
                # one rank of a vertex partition (dist.partition_mgcn): the run of plain blocks that starts here, phase by
                # phase below the C ABI with the rank's collectives between the phases (dist.part_blocks)
                plans, j = [], i
                while j < n and self._plan[j][1] == ins and self._plan[j][2] == outs:
                    nxt = self._block_at(j)
                    if nxt is None or nxt[0].pool is not None:
                        break
                    plans.append(nxt[0])
                    j = nxt[1] + 1
                if plans:
                    from .dist import part_blocks
                    y = part_blocks(plans, vals[1], vals[0])
                    if y is not None:
                        scope[outs[0]] = result = y
                        i = j
                        continue
            if on_dev and len(vals) == 2 and blocks_on:
                blk = self._block_at(i)          # conv [-> pool] -> BatchNorm -> activation below the C ABI ...
                if blk is not None and blk[0].usable(vals[0]):
                    g = vals[1]
                    if not isinstance(g, MeshGraph):
                        g = None if getattr(g, "sg_partitioned", False) or not torch.is_tensor(g) else graph_for(g, vals[0].shape[0])
                    if g is not None:
                        plans, last = [blk[0]], blk
                        while F_sg.chaining_allowed():      # ... together with the blocks that follow it on the same graph
                            j = last[1] + 1
                            nxt = self._block_at(j) if (j < n and self._plan[j][1] == ins and self._plan[j][2] == outs) else None
                            if nxt is None or last[0].pool is not None \
                                    or not nxt[0].usable_for(vals[0].dtype, vals[0].device, vals[0].shape[0], last[0].Cout):
                                break
                            plans.append(nxt[0])
                            last = nxt
                        scope[outs[0]] = result = F_sg.cheb_chain(plans, [g] * len(plans), vals[0], last[2])
                        i = last[1] + 1
                        continue
            fused = self._fusable_at(i) if (len(vals) == 1 and on_dev) else None
            mod = getattr(self, name)
            if fused is not None:      # BatchNorm1d + (Leaky)ReLU in two HIP passes instead of five ATen ones
                # (a partitioned conv keeps [owned | halo] rows in its buffer: the widened output gets the halo rows too)
                part = next((v for v in scope.values() if getattr(v, "sg_partitioned", False)), None)
                tiles = pending[1] if (pending is not None and pending[0] is vals[0]) else None
                result = F_sg.bn_act(vals[0], mod, fused[0], fused[1],
                                     1 if part is not None else fused[2], 0 if part is None else part.n_ext,
                                     tile_moments=tiles)
                i += 1
            elif isinstance(mod, ChebConv) and on_dev and mod.training and self._fusable_at(i + 1) is not None \
                    and self._plan[i + 1][1] == outs:
                mom = {}              # the MFMA product leaves the BatchNorm's block moments behind (when it serves)
                result = mod(*vals, moments=mom)
                pending = (result, mom) if "tiles" in mom else None
            else:
                result = mod(*vals)
            if len(outs) == 1:
                scope[outs[0]] = result
            else:
                scope.update(zip(outs, result))
            i += 1
        return result

    def __len__(self):
        return len(self._plan)

    def __getitem__(self, i: int):
        return getattr(self, self._plan[i][0])


def _as_graph(g, rows: int):
    """The prepared MeshGraph for a Sequential's graph argument (a MeshGraph, or an edge_index tensor); None for a vertex
    partition (its convs exchange halos between the kernels) or a graph of another size."""
    if not isinstance(g, MeshGraph):
        if getattr(g, "sg_partitioned", False) or not torch.is_tensor(g):
            return None
        g = graph_for(g, rows)
    return g if g.num_vertices == rows else None


def run_sequentials(pairs, x):
    """``seq_n(.. seq_2(seq_1(x, g_1), g_2) .., g_n)`` for Sequentials that take (x, graph) -- the composition of
    DownConv.forward / UpConv.forward (util/meshnet.py:92-95,157-160) and of SingleScaleGCN's block loop
    (util/networks.py:83-101).  The blocks the Sequentials START with are run as ONE chain across them
    (functional.cheb_chain: one foreign call and one autograd node per direction) as far as whole Sequentials consist of
    blocks; the entries behind the last block of the chain, and everything else, run module by module as written."""
    if torch.is_tensor(x) and x.is_cuda and x.dim() == 2 and len(pairs) > 1 and F_sg.blocks_enabled() and F_sg.chaining_allowed() \
            and x.shape[0] <= F_sg.CHAIN_MAX_ROWS:
        plans, graphs, rows, cin, end, widen = [], [], x.shape[0], x.shape[1], None, 1
        for k, (seq, g) in enumerate(pairs):
            lead = seq._leading_blocks() if isinstance(seq, Sequential) else None
            gg = _as_graph(g, rows) if lead is not None else None
            if gg is None:
                break
            r, c = rows, cin
            for p in lead[0]:
                if not p.usable_for(x.dtype, x.device, r, c):
                    r = -1
                    break
                r, c = p.rows_out(r), p.Cout
            if r < 0:
                break
            plans.extend(lead[0])
            graphs.extend([gg] * len(lead[0]))
            rows, cin, end, widen = r, c, (k, lead[1]), lead[2]
            if lead[1] < len(seq):      # modules behind this Sequential's blocks: the chain ends here
                break
        if end is not None and len(plans) > 1:
            x = F_sg.cheb_chain(plans, graphs, x, widen)
            seq, g = pairs[end[0]]
            if end[1] < len(seq):
                x = seq(x, g, _start=end[1])
            pairs = pairs[end[0] + 1:]
    for seq, g in pairs:
        x = seq(x, g)
    return x
