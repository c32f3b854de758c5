"""Model tier of the drop-in boundary: ``SingleScaleGCN`` with the reference's
constructor, ``forward(data, dm=None)`` contract, parameter creation order and
state-dict keys (/root/reference/util/networks.py:8-103), running on the HIP
aggregation kernels through :mod:`semigcn_amd.nn`.

Differences that are deliberate (none changes results):
  * ``data.x_pos`` / ``data.edge_index`` are uploaded once per source tensor instead of
    on every forward (util/networks.py:65), and the edge_index is turned into a CSR once;
  * on a CPU device the first aggregation raises (capi) -- there is no CPU path.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Tuple

import numpy as np
import torch
import torch.nn as nn

from .graph import MeshGraph, graph_for
from .nn import ChebConv, Sequential

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


def prepare_input(z1: torch.Tensor, dm: torch.Tensor) -> torch.Tensor:
    """util/networks.py:67-79 -- bounding-box normalisation with ONE scalar scale and a
    per-axis centre, masking of all three coordinates, mask appended as 4th channel."""
    lo = torch.min(z1, dim=0, keepdim=True)[0]
    hi = torch.max(z1, dim=0, keepdim=True)[0]
    extent = torch.max(hi - lo)
    centred = (z1 - (lo + hi) * 0.5) / extent
    return torch.cat([dm * centred, dm], dim=1)


class SingleScaleGCN(nn.Module):
    def __init__(self, device, activation: str = "lrelu", skip: bool = False):
        super().__init__()
        self.device = torch.device(device)
        self.skip = skip
        act = {"relu": nn.ReLU(), "lrelu": nn.LeakyReLU()}[activation]  # one shared instance (:17-18)
        h = CHANNELS
        blocks: List[nn.Module] = []
        for i in range(N_BLOCKS):
            layers = [(ChebConv(h[i], h[i + 1], K=3), "x, edge_index -> x"), nn.BatchNorm1d(h[i + 1]), act]
            if i == N_BLOCKS - 1:
                layers.append((nn.Linear(h[i + 1], h[i + 2]), "x -> x"))
            blocks.append(Sequential("x, edge_index", layers))
        self.blocks = nn.ModuleList(blocks)
        # created unconditionally, after the blocks (RNG order), used only when skip=True (:58-61)
        self.skip_blocks = nn.ModuleList([nn.Linear(2 * h[j + 1], h[j + 1]) for j in range(N_ENCODER)])
        self._consts = _DeviceCache()

    # -- helpers ---------------------------------------------------------------------------
    def _mask(self, dm, n: int, dtype) -> torch.Tensor:
        if isinstance(dm, np.ndarray):
            dm = torch.from_numpy(dm)
        elif not isinstance(dm, torch.Tensor):
            return torch.ones((n, 1), dtype=dtype, device=self.device)
        return dm.to(self.device)

    def graph(self, data) -> MeshGraph:
        ei = self._consts.get(data.edge_index, self.device)
        return graph_for(ei, data.z1.shape[0])

    # -- forward ---------------------------------------------------------------------------
    def forward(self, data, dm=None):
        z1 = data.z1.to(self.device)
        x_pos = self._consts.get(data.x_pos, self.device)
        graph = self.graph(data)
        x = prepare_input(z1, self._mask(dm, z1.shape[0], z1.dtype))

        enc: List[torch.Tensor] = []
        for i, block in enumerate(self.blocks):
            if self.skip and i >= FIRST_DECODER:
                j = N_BLOCKS - i  # 5, 4, 3, 2, 1 -- skip_blocks[0] is never used (:96-99)
                x = self.skip_blocks[j](torch.cat([enc[j], x], dim=1))
            x = block(x, graph)
            if i < N_ENCODER:
                enc.append(x)
        return x_pos + x
