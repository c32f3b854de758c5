"""Model tier of the drop-in boundary: ``SingleScaleGCN`` with the reference's
constructor, ``forward(data, dm=None)`` contract, parameter creation order and
state-dict keys (/root/reference/util/networks.py:8-103), running on the HIP
aggregation kernels through :mod:`semigcn_amd.nn`.

Differences that are deliberate (none changes results beyond fp32 summation order):
  * ``data.x_pos`` / ``data.edge_index`` are uploaded once per source tensor instead of
    on every forward (util/networks.py:65), and the edge_index is turned into a CSR once;
  * ``reorder=True`` (default): vertices are processed in Morton order of ``data.x_pos`` so
    that the aggregation's gathers hit L2 whatever order the scan arrived in; inputs are
    permuted at entry and the [V,3] result is returned in the caller's vertex order;
  * ``set_feature_dtype(torch.bfloat16)`` stores the per-vertex features between layers in
    bf16 (fp32 accumulation, fp32 parameters, fp32 output) -- BASELINE config c4;
  * on a CPU device the first aggregation raises (capi) -- there is no CPU path.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Tuple

import numpy as np
import torch
import torch.nn as nn

from . import reorder as _reorder
from .graph import MeshGraph, graph_for
from .nn import ChebConv, Sequential, run_sequentials

# util/networks.py:15 -- input (xyz displacement + mask) ... output xyz offset
CHANNELS: Tuple[int, ...] = (4, 16, 32, 64, 128, 256, 256, 512, 256, 256, 128, 64, 32, 16, 3)
N_BLOCKS = 13
N_ENCODER = 6      # blocks 0..5 feed the skip connections
FIRST_DECODER = 8  # blocks 8..12 may consume them


class _DeviceCache:
    """Device copies of per-mesh constant tensors, keyed by the identity and version of
    the source tensor (a reference to the source is held so ids are not recycled)."""

    def __init__(self):
        self._items: Dict[int, Tuple[torch.Tensor, int, torch.Tensor]] = {}

    def get(self, src: torch.Tensor, device: torch.device) -> torch.Tensor:
        if src.device == device:
            return src
        ent = self._items.get(id(src))
        if ent is not None and ent[0] is src and ent[1] == src._version and ent[2].device == device:
            return ent[2]
        dev = src.to(device)
        if len(self._items) > 32:
            self._items.clear()
        self._items[id(src)] = (src, src._version, dev)
        return dev


def _column_min_max(z1: torch.Tensor, block: int = 4096):
    """min / max over dim 0 of [V, 3] as [1, 3] tensors, with the values and gradient routing of
    ``torch.min(z1, dim=0)`` (one arg-extreme per column).  ATen's strided column reduction takes 1.3 ms at
    V = 1 M and a 3-row reduction of the transposed copy 0.25 ms per call (three workgroups); two stages over
    ``block``-wide pieces of the transposed copy keep the whole chip busy (~0.03 ms)."""
    zt = z1.t().contiguous()                       # [3, V]
    V = zt.shape[1]
    n = (V // block) * block
    if n == 0:
        return torch.min(zt, dim=1)[0].view(1, -1), torch.max(zt, dim=1)[0].view(1, -1)
    body = zt[:, :n].view(zt.shape[0], -1, block)
    lo, hi = torch.min(body, dim=2)[0], torch.max(body, dim=2)[0]
    if n < V:
        lo = torch.cat([lo, torch.min(zt[:, n:], dim=1, keepdim=True)[0]], dim=1)
        hi = torch.cat([hi, torch.max(zt[:, n:], dim=1, keepdim=True)[0]], dim=1)
    return torch.min(lo, dim=1)[0].view(1, -1), torch.max(hi, dim=1)[0].view(1, -1)


def prepare_input(z1: torch.Tensor, dm: torch.Tensor, lo: Optional[torch.Tensor] = None,
                  hi: Optional[torch.Tensor] = None) -> torch.Tensor:
    """util/networks.py:67-79 -- bounding-box normalisation with ONE scalar scale and a
    per-axis centre, masking of all three coordinates, mask appended as 4th channel.
    ``lo`` / ``hi`` [1,3]: mesh-wide bounds when z1 is only one rank's share."""
    if lo is None:
        lo, hi = _column_min_max(z1)
    extent = torch.max(hi - lo)
    centred = (z1 - (lo + hi) * 0.5) / extent
    return torch.cat([dm * centred, dm], dim=1)


class _Fp32Linear(nn.Linear):
    """nn.Linear whose parameters stay fp32: a reduced-precision input is widened first.  On the
    device its weight gradient (a sum over all V vertices) takes the slab-batched product."""

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        x = x.to(self.weight.dtype)
        if x.is_cuda and x.dim() == 2:
            from .functional import linear_vertices
            return linear_vertices(x, self.weight, self.bias)
        return super().forward(x)


class SingleScaleGCN(nn.Module):
    def __init__(self, device, activation: str = "lrelu", skip: bool = False, reorder: bool = True):
        super().__init__()
        self.device = torch.device(device)
        self.skip = skip
        self.reorder = reorder
        self.feature_dtype = torch.float32
        act = {"relu": nn.ReLU(), "lrelu": nn.LeakyReLU()}[activation]  # one shared instance (:17-18)
        h = CHANNELS
        blocks: List[nn.Module] = []
        for i in range(N_BLOCKS):
            layers = [(ChebConv(h[i], h[i + 1], K=3), "x, edge_index -> x"), nn.BatchNorm1d(h[i + 1]), act]
            if i == N_BLOCKS - 1:
                layers.append((_Fp32Linear(h[i + 1], h[i + 2]), "x -> x"))
            blocks.append(Sequential("x, edge_index", layers))
        # block i's activation output IS block i+1's Tx0: let it be born inside that layer's [V, 3C] buffer
        last_direct = N_BLOCKS - 1 if not skip else FIRST_DECODER - 1     # with skip, decoder inputs come from a Linear
        for i in range(last_direct):
            blocks[i].out_widen = blocks[i + 1].module_0.input_buffer_blocks()
        self.blocks = nn.ModuleList(blocks)
        # created unconditionally, after the blocks (RNG order), used only when skip=True (:58-61)
        self.skip_blocks = nn.ModuleList([_Fp32Linear(2 * h[j + 1], h[j + 1]) for j in range(N_ENCODER)])
        self._consts = _DeviceCache()
        self._orders: Dict[int, tuple] = {}

    def set_feature_dtype(self, dtype: torch.dtype) -> "SingleScaleGCN":
        if dtype not in (torch.float32, torch.bfloat16):
            raise ValueError("feature dtype must be float32 or bfloat16")
        self.feature_dtype = dtype
        return self

    # -- helpers ---------------------------------------------------------------------------
    def _mask(self, dm, n: int, dtype) -> torch.Tensor:
        if isinstance(dm, np.ndarray):
            dm = torch.from_numpy(dm)
        elif not isinstance(dm, torch.Tensor):
            return torch.ones((n, 1), dtype=dtype, device=self.device)
        return dm.to(self.device)

    def _layout(self, data):
        """(graph, order, rank) for this mesh: the CSR of the (re-numbered) edge_index and the
        Morton permutation, computed once per (edge_index, x_pos) pair."""
        prebuilt = getattr(data, "graph", None)
        if prebuilt is not None:           # a prepared (possibly partitioned) graph: nothing to derive
            return prebuilt, None, None
        ei = self._consts.get(data.edge_index, self.device)
        if not self.reorder:
            return graph_for(ei, data.z1.shape[0]), None, None
        key = (id(data.edge_index), id(data.x_pos))
        ent = self._orders.get(key)
        if (ent is not None and ent[0] is data.edge_index and ent[1] is data.x_pos
                and ent[2] == (data.edge_index._version, data.x_pos._version)):
            return ent[3], ent[4], ent[5]
        x_pos = self._consts.get(data.x_pos, self.device)
        order, rank = _reorder.morton_order(x_pos)
        graph = MeshGraph.from_edge_index(_reorder.permute_edge_index(ei, rank), data.z1.shape[0])
        if len(self._orders) > 8:
            self._orders.clear()
        self._orders[key] = (data.edge_index, data.x_pos, (data.edge_index._version, data.x_pos._version),
                             graph, order, rank)
        return graph, order, rank

    def graph(self, data) -> MeshGraph:
        return self._layout(data)[0]

    # -- forward ---------------------------------------------------------------------------
    def forward(self, data, dm=None):
        z1 = data.z1.to(self.device)
        x_pos = self._consts.get(data.x_pos, self.device)
        graph, order, rank = self._layout(data)
        lo = hi = None
        if getattr(graph, "sg_partitioned", False):
            from .dist import dist_min_max
            lo, hi = dist_min_max(z1, graph.group)
        mask = self._mask(dm, z1.shape[0], z1.dtype)
        if z1.is_cuda and z1.dtype == torch.float32 and mask.dtype == torch.float32 and mask.numel() == z1.shape[0]:
            # one launch: normalise, mask, append the mask, rows into processing order, feature dtype (csrc/input_prep.hip)
            from .functional import input_prep
            x = input_prep(z1, lo, hi, mask, order, rank, self.feature_dtype)    # (lo is None: the library takes the bounds too)
        else:
            x = prepare_input(z1, mask, lo, hi)
            if order is not None:
                x = x.index_select(0, order)
            x = x.to(self.feature_dtype)

        halo_in = getattr(data, "halo_inputs", None)
        if not self.skip and halo_in is not None and getattr(graph, "sg_partitioned", False) and x.is_cuda and self.training:
            # one rank of a vertex partition: the blocks phase by phase below the C ABI, the collectives between the calls;
            # the halo rows of the network input are prepared here from the halo copies of z1 / dm (no exchange for block 0)
            from .dist import part_chain
            from .functional import input_prep
            with torch.no_grad():
                x_halo = input_prep(halo_in[0], lo.detach(), hi.detach(), halo_in[1], None, None, self.feature_dtype)
            res = part_chain(self.blocks, graph, x, x_halo)
            if res is not None:
                x = self.blocks[-1](res[0], graph, _start=res[1])
                return x_pos + x
        if not self.skip:
            # the loop over the 13 blocks (:83-101); on the device their whole kernel chain is ONE call below the C ABI
            x = run_sequentials([(block, graph) for block in self.blocks], x)
        else:
            enc: List[torch.Tensor] = []
            for i, block in enumerate(self.blocks):
                if i >= FIRST_DECODER:
                    j = N_BLOCKS - i  # 5, 4, 3, 2, 1 -- skip_blocks[0] is never used (:96-99)
                    x = self.skip_blocks[j](torch.cat([enc[j], x], dim=1)).to(self.feature_dtype)
                x = block(x, graph)
                if i < N_ENCODER:
                    enc.append(x)
        if rank is not None:
            if x.is_cuda:
                from .functional import output_in_caller_order
                return output_in_caller_order(x, x_pos, rank, order)
            x = x.index_select(0, rank)
        return x_pos + x
