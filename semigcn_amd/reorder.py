"""Locality ordering of mesh vertices for the aggregation kernel.

A scan arrives in arbitrary vertex order; the aggregation then gathers 6 random rows
per vertex and nothing is re-used from cache (about 6x the compulsory HBM traffic).
Sorting the vertices along a Morton (Z-order) curve through their 3-D positions makes
consecutive rows a compact patch of the surface, so a wavefront chunk's neighbours are
mostly rows that it or its neighbouring chunks already pulled into L2.  The model tier
applies the permutation once per mesh at its entry and un-permutes its [V,3] output, so
callers still see the reference's vertex order (util/networks.py:63-103 contract).
"""
from __future__ import annotations

import torch


def _spread3(x: torch.Tensor) -> torch.Tensor:
    """Spread the low 21 bits of int64 x so that two zero bits separate consecutive bits."""
    x = x & 0x1FFFFF
    x = (x | (x << 32)) & 0x1F00000000FFFF
    x = (x | (x << 16)) & 0x1F0000FF0000FF
    x = (x | (x << 8)) & 0x100F00F00F00F00F
    x = (x | (x << 4)) & 0x10C30C30C30C30C3
    x = (x | (x << 2)) & 0x1249249249249249
    return x


def morton_codes(pos: torch.Tensor, bits: int = 21) -> torch.Tensor:
    """63-bit Morton code of [V,3] positions quantised to a 2^bits grid on their bounding box."""
    p = pos.detach().to(torch.float64)
    lo = p.min(dim=0, keepdim=True)[0]
    extent = (p.max(dim=0, keepdim=True)[0] - lo).max().clamp(min=1e-30)
    q = ((p - lo) / extent * (2 ** bits - 1)).round().to(torch.int64).clamp_(0, 2 ** bits - 1)
    return _spread3(q[:, 0]) | (_spread3(q[:, 1]) << 1) | (_spread3(q[:, 2]) << 2)


def morton_order(pos: torch.Tensor):
    """(order, rank): ``order[k]`` = old index of the vertex placed k-th; ``rank[i]`` = new
    index of old vertex i (``rank = order^-1``)."""
    order = torch.argsort(morton_codes(pos), stable=True)
    rank = torch.empty_like(order)
    rank[order] = torch.arange(order.numel(), device=order.device)
    return order, rank


def permute_edge_index(edge_index: torch.Tensor, rank: torch.Tensor) -> torch.Tensor:
    """Relabel the endpoints of [2,E] edge_index with the new vertex numbers."""
    return rank[edge_index]
