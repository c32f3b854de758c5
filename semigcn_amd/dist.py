"""Vertex-partitioned SGCN over the GPUs of one node.

New design -- the reference is single-process, single-device and has no counterpart
(no torch.distributed / NCCL call anywhere in /root/reference).  The contract is
"N-GPU result == 1-GPU result" to fp32 summation order.

  * The mesh is renumbered along a Morton curve (semigcn_amd.reorder) and cut into
    ``world`` contiguous blocks of vertices: each rank OWNS the rows of every [V, C]
    tensor for its block, and the CSR rows of L^ for them with the column indices
    remapped to ``[owned | halo]`` (sg_graph_create_rect).
  * Before every aggregation the 1-ring halo rows are exchanged: boundary rows are packed by
    a HIP gather kernel (sg_gather_rows), sent with ONE all_to_all_single (RCCL over xGMI:
    grouped point-to-point sends to the few neighbouring ranks), and land at the tail of the
    ``[owned | halo]`` feature buffer the kernel reads.  Backward is owner-computes on
    exchanged GRADIENT rows (the global L^ is symmetric), so no reverse scatter is needed.
  * BatchNorm statistics couple all vertices (util/networks.py:43): per-rank (count, mean,
    M2) are all-gathered and merged with Chan's formula (forward), per-channel
    (sum dy, sum dy*xhat) are all-reduced (backward).
  * The bounding-box normalisation (util/networks.py:67-70) all-reduces min / max; the
    losses all-reduce their masked sums; face normals read halo positions through a
    differentiable exchange (every vertex of a face that touches an owned vertex is in that
    vertex's 1-ring, i.e. already in the halo plan).
  * Parameter gradients are partial sums over the owned vertices: one flat all-reduce per
    optimiser step (7 MB for SGCN), after the ``accumulate`` backward passes.
"""
from __future__ import annotations

import ctypes
import os
from dataclasses import dataclass
from typing import List, Optional, Sequence

import numpy as np
import torch
import torch.distributed as dist
import torch.nn as nn

from . import capi
from . import reorder as _reorder


# --------------------------------------------------------------------------------------
# collectives.  RCCL ("nccl") takes device tensors directly.  Under gloo -- the CPU tests, and the
# single-GPU self-test where several ranks share one device -- device tensors are staged through
# host memory, because gloo moves host buffers only.
# --------------------------------------------------------------------------------------
#: collectives issued by this rank since import, by kind (bench.py reports the per-iteration count)
collective_counts = {"all_reduce": 0, "all_gather": 0, "all_to_all": 0}

#: SEMIGCN_DIST_FORCE_COLLECTIVES=1: a ONE-rank group issues every collective as well (normally skipped: with one rank
#: they are the identity).  A self-test switch: on a box with a single GPU this is the only way to drive the RCCL code
#: path -- communicator set-up, the asynchronous all-to-all and its stream wait, the statistics all-gather, the
#: all-reduces -- through the real library (tests/test_gpu_scale.py).
FORCE_COLLECTIVES = os.environ.get("SEMIGCN_DIST_FORCE_COLLECTIVES") == "1"


def _solo(world: int) -> bool:
    """True when collectives may be skipped: a single rank (and the self-test switch is off)."""
    return world == 1 and not FORCE_COLLECTIVES


_backend_of: dict = {}


def _pg(group):
    return group if group is not None else dist.distributed_c10d._get_default_group()


def _backend(group) -> str:
    """Backend name of ``group`` (None = WORLD), looked up once per group object."""
    g = _pg(group)
    ent = _backend_of.get(id(g))
    if ent is None or ent[0] is not g:
        if len(_backend_of) > 16:
            _backend_of.clear()
        ent = (g, dist.get_backend(group))
        _backend_of[id(g)] = ent
    return ent[1]


def _staged(t: torch.Tensor, group) -> bool:
    return t.is_cuda and _backend(group) == "gloo"


# The collectives go to the ProcessGroup object directly (``allreduce`` / ``_allgather_base`` / ``alltoall_base``: what
# torch.distributed's module-level functions call after their argument checks, logging wrapper and group lookups --
# ~10-15 us of host time each, 57 times per partitioned iteration on a rank that is launch-bound).  DIRECT_PG = False
# (or an older torch without these methods) uses the module-level functions.
DIRECT_PG = os.environ.get("SEMIGCN_DIST_PUBLIC_API") != "1"
_opts: dict = {}


def _pg_all_reduce(t: torch.Tensor, op, group) -> None:
    if DIRECT_PG:
        try:
            o = _opts.get(("ar", op))
            if o is None:
                o = dist.AllreduceOptions()
                o.reduceOp = op
                _opts[("ar", op)] = o
            work = _pg(group).allreduce([t], o)
        except (AttributeError, TypeError):
            work = dist.all_reduce(t, op=op, group=group, async_op=True)
        if work is not None:
            work.wait()            # device tensors: the CURRENT stream waits for the collective, the host does not block
        return
    dist.all_reduce(t, op=op, group=group)


def _pg_all_gather(out: torch.Tensor, inp: torch.Tensor, group) -> None:
    if DIRECT_PG:
        try:
            work = _pg(group)._allgather_base(out, inp)
        except (AttributeError, TypeError):
            work = dist.all_gather_into_tensor(out, inp, group=group, async_op=True)
        if work is not None:
            work.wait()
        return
    dist.all_gather_into_tensor(out, inp, group=group)


def _pg_all_to_all(recv: torch.Tensor, send: torch.Tensor, recv_splits, send_splits, group):
    """Starts the exchange and returns the Work handle (``.wait()``: the current stream waits)."""
    if DIRECT_PG:
        try:
            o = _opts.get("a2a")
            if o is None:
                o = _opts["a2a"] = dist.AllToAllOptions()
            return _pg(group).alltoall_base(recv, send, recv_splits, send_splits, o)
        except (AttributeError, TypeError):
            pass
    return dist.all_to_all_single(recv, send, recv_splits, send_splits, group=group, async_op=True)


# (Rounds 2-4 could record a partitioned iteration as hipGraph segments with these calls between them, or as ONE graph with the
# RCCL calls captured inside; both modes were retired in round 5 -- the eager phase path is faster than either on the rank
# proxy and is the one path a first multi-GPU lease has to debug.  DESIGN.md section 8.)
def _all_reduce(t: torch.Tensor, op, group) -> None:
    def run():
        collective_counts["all_reduce"] += 1
        if _staged(t, group):
            h = t.cpu()
            _pg_all_reduce(h, op, group)
            t.copy_(h)
        else:
            _pg_all_reduce(t, op, group)
    run()


def _all_gather_rows(out: torch.Tensor, inp: torch.Tensor, group) -> None:
    def run():
        collective_counts["all_gather"] += 1
        if _staged(inp, group):
            ho, hi = out.cpu(), inp.cpu()
            _pg_all_gather(ho, hi, group)
            out.copy_(ho)
        else:
            _pg_all_gather(out, inp, group)
    run()


def _all_to_all_rows(recv: torch.Tensor, send: torch.Tensor, recv_splits, send_splits, group) -> None:
    recv_splits, send_splits = list(recv_splits), list(send_splits)

    def run():
        collective_counts["all_to_all"] += 1
        if _staged(send, group):
            hr, hs = recv.cpu(), send.cpu()
            work = _pg_all_to_all(hr, hs, recv_splits, send_splits, group)
            if work is not None:
                work.wait()
            recv.copy_(hr)
        else:
            work = _pg_all_to_all(recv, send, recv_splits, send_splits, group)
            if work is not None:
                work.wait()
    run()


# --------------------------------------------------------------------------------------
# partition plan
# --------------------------------------------------------------------------------------
def block_bounds(num_vertices: int, world: int) -> List[int]:
    """Balanced contiguous blocks: bounds[r] .. bounds[r+1] is rank r's range."""
    base, rem = divmod(num_vertices, world)
    b = [0]
    for r in range(world):
        b.append(b[-1] + base + (1 if r < rem else 0))
    return b


class _RowExchange:
    """Exchange of rows of a block-partitioned [V, C] tensor along a fixed plan.  Needs: ``world, group,
    n_own, n_halo, n_send, send_rows`` (int32 local row ids, grouped by destination rank, ascending),
    ``send_splits, recv_splits`` (rows per peer)."""

    def _a2a(self, recv: torch.Tensor, send: torch.Tensor, recv_splits, send_splits):
        if _solo(self.world):
            return
        _all_to_all_rows(recv, send, recv_splits, send_splits, self.group)

    def exchange(self, blk_ext: torch.Tensor) -> None:
        """Fill rows [n_own:] of ``blk_ext`` ([n_ext, C], unit column stride, any row stride) with
        the owners' copies of those rows; rows [:n_own] must be final."""
        self.exchange_end(self.exchange_begin(blk_ext))

    def exchange_begin(self, blk_ext: torch.Tensor):
        """Start the exchange and return a token for ``exchange_end``.  Over RCCL the all-to-all is issued asynchronously
        (it runs on the communicator's stream behind the pack kernel), so kernels launched before ``exchange_end`` --
        the aggregation of the rows that read no halo row -- overlap with it; under gloo (host-staged) it completes here."""
        if _solo(self.world):
            return None                # (with world > 1 every rank takes part, even with an empty halo: it is a collective)
        own = blk_ext[:self.n_own]
        send = capi.gather_rows(self.send_rows, own)
        recv = torch.empty((self.n_halo, blk_ext.shape[1]), dtype=blk_ext.dtype, device=blk_ext.device)
        work = None
        if send.is_cuda and _backend(self.group) != "gloo":
            work = {}                  # filled by the start action, drained by the wait action

            def start():
                collective_counts["all_to_all"] += 1
                work["w"] = _pg_all_to_all(recv, send, self.recv_splits, self.send_splits, self.group)
                _c10d_in_flight[0] += 1          # the one torch.distributed collective left un-awaited across calls
            start()
        else:
            self._a2a(recv, send, self.recv_splits, self.send_splits)
        return work, recv, send, blk_ext

    def exchange_end(self, token) -> None:
        if token is None:
            return
        work, recv, _send, blk_ext = token
        if work is not None:
            def wait():
                w = work.pop("w", None)
                if w is not None:
                    w.wait()           # the CURRENT stream waits for the collective; the host does not block
                    _c10d_in_flight[0] -= 1
            wait()
        blk_ext[self.n_own:].copy_(recv)

    def exchange_reverse_add(self, grad_halo: torch.Tensor, grad_own: torch.Tensor) -> None:
        """Adjoint of ``exchange``: halo-row gradients travel back to their owners and are added."""
        if _solo(self.world):
            return
        recv = torch.empty((self.n_send, grad_halo.shape[1]), dtype=grad_halo.dtype, device=grad_halo.device)
        self._a2a(recv, grad_halo.contiguous(), self.send_splits, self.recv_splits)
        if recv.is_cuda and self.n_send > 0:
            # a boundary row can sit in the halo of SEVERAL peers (3+ ranks): its returning contributions are summed in a
            # fixed order by the segment kernel (sg_unpool_bwd over the send list), not by index_add_'s atomics -- the
            # partitioned iteration stays bit-reproducible run to run
            rev = getattr(self, "_reverse_sum", None)
            if rev is None:
                rev = self._reverse_sum = capi.PoolHandle(torch.arange(self.n_send, device=recv.device),
                                                          self.send_rows.long(), self.n_send, self.n_own)
            grad_own.add_(rev.unpool_bwd(recv))
        else:
            grad_own.index_add_(0, self.send_rows.long(), recv)


class HaloPlan(_RowExchange):
    """Generic plan: "rank ``consumer[i]`` needs global row ``rows[i]``" for a tensor cut into the
    contiguous blocks ``bounds``.  Every rank builds it from the same global lists, so what to send
    is known without a handshake.  Pairs whose consumer owns the row are ignored."""

    def __init__(self, consumer: torch.Tensor, rows: torch.Tensor, bounds: Sequence[int], rank: int, world: int,
                 group=None):
        dev = rows.device
        self.group, self.rank, self.world = group, rank, world
        b = torch.tensor(list(bounds), device=dev, dtype=torch.long)
        n_rows = int(bounds[-1])
        self.start, self.end = int(bounds[rank]), int(bounds[rank + 1])
        self.n_own = self.end - self.start
        owner = torch.bucketize(rows, b[1:], right=True)
        remote = owner != consumer
        pair = torch.unique(consumer[remote] * n_rows + rows[remote])          # sorted by (consumer, row)
        cons, row = pair // n_rows, pair % n_rows
        mine = cons == rank
        self.halo_ids = row[mine]                                              # ascending = grouped by owner
        self.n_halo = int(self.halo_ids.numel())
        self.n_ext = self.n_own + self.n_halo
        self.recv_splits = torch.bincount(torch.bucketize(self.halo_ids, b[1:], right=True), minlength=world).tolist()
        out = (row >= self.start) & (row < self.end)
        self.send_splits = torch.bincount(cons[out], minlength=world).tolist()
        self.send_rows = (row[out] - self.start).to(torch.int32)
        self.n_send = int(self.send_rows.numel())

    def extended_index(self, rows: torch.Tensor) -> torch.Tensor:
        """Position of global ``rows`` (owned or in the halo) inside this rank's ``[owned | halo]`` buffer."""
        own = (rows >= self.start) & (rows < self.end)
        return torch.where(own, rows - self.start, self.n_own + torch.searchsorted(self.halo_ids, rows))


class DistMeshGraph(_RowExchange):
    """One rank's share of the scaled Laplacian plus its halo-exchange plan.

    The halo is TWO rings deep: ``[owned | halo]`` holds the owned rows, their remote neighbours (ring 1) and those
    neighbours' remote neighbours (ring 2), halo rows in ascending global id (= grouped by owner, so a peer's rows
    land contiguously).  ONE exchange then serves both aggregations of a K = 3 ChebConv: ``handle_wide`` applies
    L^ on the owned AND ring-1 rows (Tx1 recomputed redundantly on ring 1 -- a few per cent of the rows), ``handle``
    on the owned rows only (Tx2).  Ring-2 rows carry no edges in ``handle_wide`` (their output rows are not used)."""
    sg_partitioned = True
    #: True: a Sequential that meets this graph runs its runs of plain blocks phase by phase below the C ABI (part_blocks;
    #: set by partition_mgcn -- the SGCN's trainer calls part_chain itself)
    phases = False

    def __init__(self, edge_index: torch.Tensor, num_vertices: int, rank: int, world: int,
                 group=None, bounds: Optional[Sequence[int]] = None):
        """``edge_index`` [2,E] int64 on the compute device, ALREADY in processing order
        (callers renumber with reorder.morton_order first); every rank passes the full graph."""
        dev = edge_index.device
        self.group, self.rank, self.world = group, rank, world
        self.num_vertices_global = V = int(num_vertices)
        bounds = list(bounds) if bounds is not None else block_bounds(num_vertices, world)
        self.bounds = bounds
        start, end = bounds[rank], bounds[rank + 1]
        self.start, self.end, self.n_own = start, end, end - start

        ei = edge_index[:, edge_index[0] != edge_index[1]]
        src, dst = ei[0], ei[1]
        fwd = torch.sort(dst * num_vertices + src)[0]
        bwd = torch.sort(src * num_vertices + dst)[0]
        if not torch.equal(fwd, bwd):
            raise ValueError("vertex partitioning needs a symmetric edge_index (an undirected mesh graph)")
        deg = torch.bincount(src, minlength=num_vertices).float()
        dis = torch.where(deg > 0, deg.rsqrt(), torch.zeros_like(deg))

        def rings_of(lo: int, hi: int):
            """(ring 1, ring 2) of the block [lo, hi) as boolean vertex masks: a row gathers from the SOURCES of the
            edges that end in it."""
            own = torch.zeros(V, dtype=torch.bool, device=dev)
            own[lo:hi] = True
            r1 = torch.zeros(V, dtype=torch.bool, device=dev)
            r1[src[own[dst]]] = True
            r1 &= ~own
            r2 = torch.zeros(V, dtype=torch.bool, device=dev)
            r2[src[r1[dst]]] = True
            r2 &= ~(own | r1)
            return own, r1, r2

        own, r1, r2 = rings_of(start, end)
        halo = torch.nonzero(r1 | r2).flatten()                  # ascending global ids = grouped by owner
        self.halo_ids = halo
        self.n_halo = int(halo.numel())
        self.n_halo1 = int(r1.sum())
        self.n_ext = self.n_own + self.n_halo
        ext_of = torch.full((V,), -1, dtype=torch.long, device=dev)
        ext_of[start:end] = torch.arange(self.n_own, device=dev)
        ext_of[halo] = self.n_own + torch.arange(self.n_halo, device=dev)
        self._ext_of = ext_of
        dis_ext = torch.cat([dis[start:end], dis[halo]])
        e_own = own[dst]                                          # edges that end in an owned row
        self.handle = capi.GraphHandle.from_partition(dst[e_own] - start, ext_of[src[e_own]], self.n_own, self.n_ext, dis_ext)
        e_wide = e_own | r1[dst]                                  # ... or in a ring-1 row (complete rows: global degrees)
        self.handle_wide = capi.GraphHandle.from_partition(ext_of[dst[e_wide]], ext_of[src[e_wide]], self.n_ext, self.n_ext,
                                                           dis_ext)
        # the same wide operator cut in two row sets, for overlap with the exchange: INTERIOR = owned rows none of whose
        # neighbours is a halo row (they can be aggregated while the halo is in flight), REST = the owned boundary rows
        # and ALL ring-1 rows (a ring-1 row may read owned rows only, but its own epilogue operand / output row is a
        # halo row that the exchange has yet to fill)
        reads_halo = torch.zeros(self.n_ext, dtype=torch.bool, device=dev)
        reads_halo[ext_of[dst[e_wide & ~own[src]]]] = True
        is_row = torch.zeros(self.n_ext, dtype=torch.bool, device=dev)
        is_row[:self.n_own] = True
        is_row[ext_of[torch.nonzero(r1).flatten()]] = True
        # (kept for FoldedLayout: the same two operators on the folded row numbering)
        self._fold_src = (dst, src, e_own, e_wide, dis, halo, [None] * world)
        self._split = []
        dst_w, src_w = ext_of[dst[e_wide]], ext_of[src[e_wide]]
        owned_row = torch.zeros(self.n_ext, dtype=torch.bool, device=dev)
        owned_row[:self.n_own] = True
        interior = owned_row & ~reads_halo
        for rows_mask in (interior, is_row & ~interior):
            rows = torch.nonzero(rows_mask).flatten()
            pos = torch.full((self.n_ext,), -1, dtype=torch.long, device=dev)
            pos[rows] = torch.arange(rows.numel(), device=dev)
            sel = rows_mask[dst_w]
            self._split.append(capi.GraphHandle.from_rows(pos[dst_w[sel]], src_w[sel], rows, self.n_ext, self.n_ext,
                                                          dis_ext[rows], dis_ext))
        self.n_interior = int(interior.sum())                    # rows aggregated while the exchange is in flight

        # what I receive from each peer: my halo ids that it owns (contiguous runs of `halo`)
        b = torch.tensor(bounds, device=dev, dtype=torch.long)
        self.recv_splits = torch.bincount(torch.bucketize(halo, b[1:], right=True), minlength=world).tolist()
        # what I send to peer q: my rows inside q's two rings, ascending -- q's receive order
        send_rows, send_splits = [], []
        for q in range(world):
            if q == rank or bounds[q + 1] == bounds[q]:
                send_splits.append(0)
                continue
            _, q1, q2 = rings_of(bounds[q], bounds[q + 1])
            rows = torch.nonzero((q1 | q2)[start:end]).flatten()
            self._fold_src[6][q] = rows
            send_rows.append(rows)
            send_splits.append(int(rows.numel()))
        self.send_splits = send_splits
        self.send_rows = (torch.cat(send_rows) if send_rows else torch.zeros(0, dtype=torch.long, device=dev)).to(torch.int32)
        self.n_send = int(self.send_rows.numel())
        self.device = dev
        self.symmetric = True

    def extended_index(self, rows: torch.Tensor) -> torch.Tensor:
        """Position of global ``rows`` (owned or in the halo) inside this rank's ``[owned | halo]`` buffer."""
        return self._ext_of[rows]

    def folded(self) -> "FoldedLayout":
        """The layout of the phase-by-phase block path (``part_chain``): ``[owned | per peer: its halo rows, PAD_ROWS pad
        rows]`` -- the pad rows carry the peer's BatchNorm statistics, so an exchange lands in place, statistics included."""
        f = self.__dict__.get("_folded")
        if f is None:
            f = self.__dict__["_folded"] = FoldedLayout(self)
        return f

    # MeshGraph-compatible surface -----------------------------------------------------
    @property
    def num_vertices(self) -> int:
        return self.n_own

    def aggregate(self, X_ext: torch.Tensor, Y_own: torch.Tensor, **kw):
        """L^ on the owned rows: X on ``[owned | halo]`` (ring 1 valid), Y on the owned rows."""
        return self.handle.spmm(X_ext, Y_own, **kw)

    def aggregate_wide(self, X_ext: torch.Tensor, Y_ext: torch.Tensor, **kw):
        """L^ on the owned and ring-1 rows: X on ``[owned | halo]`` (both rings valid), Y on all ``n_ext`` rows (its
        ring-2 rows receive only the epilogue terms and are not to be used)."""
        return self.handle_wide.spmm(X_ext, Y_ext, **kw)

    #: aggregate the interior rows while the halo exchange is in flight (False: one launch after the exchange)
    overlap = True

    def exchange_and_aggregate_wide(self, X_blk: torch.Tensor, X_ext: torch.Tensor, Y_ext: torch.Tensor, **kw):
        """``exchange(X_blk)`` + ``aggregate_wide(X_ext, Y_ext)`` with the interior rows computed during the exchange.
        ``X_blk``: the column block(s) to exchange (its owned rows final); ``X_ext``: the block the operator reads
        (``X_blk`` or a column slice of it).  Epilogue operands in ``kw`` are read on owned / ring-1 rows only."""
        if not self.overlap or _solo(self.world):
            self.exchange(X_blk)
            return self.aggregate_wide(X_ext, Y_ext, **kw)
        token = self.exchange_begin(X_blk)
        self._split[0].spmm(X_ext, Y_ext, **kw)        # rows that read owned rows only
        self.exchange_end(token)
        return self._split[1].spmm(X_ext, Y_ext, **kw)  # boundary rows and ring-1 rows


class _HaloExtend(torch.autograd.Function):
    """x_own [n,C] -> [x_own ; halo rows] [n_ext, C], differentiable."""

    @staticmethod
    def forward(ctx, g: _RowExchange, x: torch.Tensor):
        ext = torch.empty((g.n_ext, x.shape[1]), dtype=x.dtype, device=x.device)
        ext[:g.n_own].copy_(x)
        g.exchange(ext)
        ctx.g = g
        return ext

    @staticmethod
    def backward(ctx, grad_ext):
        g = ctx.g
        grad_own = grad_ext[:g.n_own].clone()
        g.exchange_reverse_add(grad_ext[g.n_own:], grad_own)
        return None, grad_own


def halo_extend(g: _RowExchange, x: torch.Tensor) -> torch.Tensor:
    return _HaloExtend.apply(g, x)


class _AllReduceSum(torch.autograd.Function):
    """y = sum over ranks of x; every rank continues with the same replicated computation, so
    the gradient of the replicated result w.r.t. the local term is the incoming gradient."""

    @staticmethod
    def forward(ctx, x, group):
        y = x.clone()
        _all_reduce(y, dist.ReduceOp.SUM, group)
        return y

    @staticmethod
    def backward(ctx, grad):
        return grad, None


def all_reduce_sum(x: torch.Tensor, group=None) -> torch.Tensor:
    if not dist.is_initialized() or _solo(dist.get_world_size(group)):
        return x
    return _AllReduceSum.apply(x, group)


PAD_ROWS = 5        # csrc/block.hip kPadRows: (2 C + 1) floats fit into 5 rows of C bf16 / fp32 values

#: The phase path's collectives below the C ABI (csrc/comm.hip: the library's own RCCL communicator, a rank's forward /
#: backward pass over its blocks as ONE foreign call each).  SEMIGCN_DIST_NATIVE=0 keeps them on torch.distributed -- what
#: gloo groups always use, and what bench.py's supervisor falls back to.
NATIVE_COLLECTIVES = os.environ.get("SEMIGCN_DIST_NATIVE", "1") != "0"
_comm_seq = [0]
#: passes (forward, backward) this rank ran as one sg_part_run call
native_runs = [0, 0]


#: SEMIGCN_DIST_KNOWN_ANSWER_TIMEOUT (seconds, 0 = off): bench.py's supervisor sets it for its workers.  The known-answer
#: collectives below are the FIRST time csrc/comm.hip talks to a real peer; if they hang (rather than disagree), the worker
#: would sit in a device synchronise until the supervisor's attempt limit (300 s on the 4 M mesh).  With the switch on, a timer
#: thread ends THIS process after that many seconds (exit code 86, a line on stderr, a mark in SEMIGCN_BENCH_MARK): the
#: supervisor sees a failed attempt at once and starts a fresh worker with the collectives on torch.distributed.  Nothing is
#: exec'ed; a library user who never sets the variable never has a process ended under them.
KNOWN_ANSWER_TIMEOUT_S = float(os.environ.get("SEMIGCN_DIST_KNOWN_ANSWER_TIMEOUT", "0") or 0)
EXIT_NATIVE_HANG = 86


class _HangGuard:
    def __init__(self, what: str, seconds: float = None):
        self.what, self.seconds, self.timer = what, KNOWN_ANSWER_TIMEOUT_S if seconds is None else seconds, None

    def _fire(self):
        import sys
        msg = (f"semigcn_amd.dist: {self.what} did not finish within {self.seconds:.0f} s -- the library's own communicator "
               f"hangs against its peers; ending this worker (exit {EXIT_NATIVE_HANG}) so that the supervisor retries with "
               "SEMIGCN_DIST_NATIVE=0")
        try:
            print(msg, file=sys.stderr, flush=True)
            mark = os.environ.get("SEMIGCN_BENCH_MARK")
            if mark:
                with open(os.path.join(mark, f"native_comm_hang_rank{os.environ.get('RANK', '0')}"), "w") as f:
                    f.write(msg)
        finally:
            os._exit(EXIT_NATIVE_HANG)

    def __enter__(self):
        if self.seconds > 0:
            import threading
            self.timer = threading.Timer(self.seconds, self._fire)
            self.timer.daemon = True
            self.timer.start()
        return self

    def __exit__(self, *exc):
        if self.timer is not None:
            self.timer.cancel()
        return False


class _SharedComm:
    """A handle from `Comm.share` that keeps its base alive and closes with it."""

    def __init__(self, comm, base):
        self._c, self._base = comm, base
        self._h = comm._h

    def __getattr__(self, name):
        return getattr(self._c, name)


_base_comms: dict = {}     # id(process group) -> (group, capi.Comm or False)


def _agree(ok: bool, group, dev) -> bool:
    """True when EVERY rank of the group says ok (an all-reduce on the torch process group: the fallback path itself)."""
    flag = torch.tensor([0.0 if ok else 1.0], device=dev)
    _pg_all_reduce(flag, dist.ReduceOp.SUM, group)
    return float(flag.item()) == 0.0


def _base_comm(lay):
    """The process group's ONE library communicator, created (collectively) by the first layout that asks, or None.

    Nothing below may strand a peer inside ncclCommInitRank or a native collective, so every stage is agreed on through the
    TORCH process group before the next one starts:
      1. every rank loads librccl and obtains the 128-byte id (rank 0 draws it, the store carries it)     -> agree
      2. every rank creates its communicator (ncclCommInitRank: collective, all ranks are known to arrive) -> agree
      3. known answers, bit for bit against torch.distributed's own collectives on the same buffers: ONE exchange with this
         layout's per-peer rows (row r of what rank p sends carries (p, r): a wrong peer offset shows), ONE all-reduce, ONE
         all-gather -- the three collectives sg_part_run issues                                             -> agree
    On any disagreement every rank drops its communicator and the phase path keeps its collectives on torch.distributed."""
    pg = _pg(lay.group)
    ent = _base_comms.get(id(pg))
    if ent is not None and ent[0] is pg:
        return ent[1] or None
    if len(_base_comms) > 16:
        _base_comms.clear()
    _base_comms[id(pg)] = (pg, False)
    dev = lay.graph.device
    import warnings

    def give_up(why: str):
        warnings.warn(f"semigcn_amd.dist: {why}; the phase path keeps its collectives on torch.distributed")
        return None
    # -- 1. library + id
    uid, err = None, None
    try:
        if not capi.Comm.available():
            raise RuntimeError("librccl is not loadable from the library")
        store = dist.distributed_c10d._get_default_store()
        ranks = dist.get_process_group_ranks(pg)
        _comm_seq[0] += 1
        key = "semigcn/sg_comm/%s/%d" % ("-".join(map(str, ranks)), _comm_seq[0])
        if lay.rank == 0:
            uid = capi.Comm.unique_id()
            store.set(key, uid)
        else:
            uid = bytes(store.get(key))
    except Exception as e:                                  # noqa: BLE001 -- whatever went wrong, the peers must hear of it
        err = f"{type(e).__name__}: {e}"
    if not _agree(err is None, lay.group, dev):
        return give_up("a rank could not load RCCL or obtain the communicator id" + (f" (here: {err})" if err else ""))
    # -- 2. the communicator
    comm = None
    try:
        comm = capi.Comm(uid, lay.rank, lay.world, lay.send_splits, lay.recv_splits, dev)
    except Exception as e:                                  # noqa: BLE001
        err = f"{type(e).__name__}: {e}"
    if not _agree(comm is not None, lay.group, dev):
        if comm is not None:
            comm.close()
        return give_up("ncclCommInitRank failed on a rank" + (f" (here: {err})" if err else ""))
    # -- 3. known answers (the two-communicator rule below holds: every torch collective is awaited before a native one starts).
    #       torch.distributed's side runs FIRST and outside the hang guard: its first all-to-all sets up the process group's own
    #       point-to-point channels (seconds on eight ranks) and must not be mistaken for a hang of the library's communicator
    ok = True
    try:
        send = torch.empty((lay.n_send, 4), dtype=torch.float32, device=dev)
        send[:, 0] = float(lay.rank)
        send[:, 1] = torch.arange(lay.n_send, device=dev, dtype=torch.float32)
        send[:, 2:] = 0.5
        got = torch.full((lay.n_ext - lay.n_own, 4), -1.0, device=dev)
        want = torch.full_like(got, -2.0)
        red = torch.arange(7, device=dev, dtype=torch.float32) * float(lay.rank + 1)
        red_want = red.clone()
        gin = torch.arange(5, device=dev, dtype=torch.float32) + 10.0 * lay.rank
        gout = torch.full((lay.world * 5,), -1.0, device=dev)
        gwant = torch.full_like(gout, -2.0)
        _all_to_all_rows(want, send, lay.recv_splits, lay.send_splits, lay.group)
        collective_counts["all_to_all"] -= 1
        _pg_all_reduce(red_want, dist.ReduceOp.SUM, lay.group)
        _pg_all_gather(gwant, gin, lay.group)
        torch.cuda.synchronize(dev)
        with _HangGuard("the known-answer collectives of the library's own communicator"):
            comm.halo_exchange(got, send)
            comm.all_reduce_(red)
            comm.all_gather(gout, gin)
            torch.cuda.synchronize(dev)
        ok = bool(torch.equal(got, want)) and bool(torch.equal(red, red_want)) and bool(torch.equal(gout, gwant))
    except Exception as e:                                  # noqa: BLE001
        ok, err = False, f"{type(e).__name__}: {e}"
    if not _agree(ok, lay.group, dev):
        comm.close()
        return give_up("the library's own collectives disagreed with torch.distributed's on the known-answer buffers"
                       + (f" (here: {err})" if err else ""))
    comm.layout = lay
    _base_comms[id(pg)] = (pg, comm)
    return comm


def _assert_c10d_drained() -> None:
    """THE TWO-COMMUNICATOR RULE.  A rank holds two RCCL communicators on one device: torch.distributed's (input bounds, loss
    sums, the flat gradient all-reduce: 4 collectives per iteration) and the library's (the 40 of the phase path).  Two
    communicators must never have work interleaved on the device in different orders on different ranks, so: every
    torch.distributed collective this module starts is AWAITED ON THE COMPUTE STREAM (`work.wait()`) before the call that
    started it returns or, for the one asynchronous exchange (`_Exchange`), before any native collective is enqueued --
    and the native ones are enqueued on that same stream.  `_c10d_in_flight` counts the asynchronous ones; sg_part_run must
    find it at zero.  Anyone adding `async_op=True` elsewhere has to count it here too."""
    if _c10d_in_flight[0] != 0:
        raise RuntimeError(f"semigcn_amd.dist: {_c10d_in_flight[0]} torch.distributed collective(s) still un-awaited when a "
                           "native collective is about to be enqueued (two RCCL communicators would interleave)")


_c10d_in_flight = [0]


class FoldedLayout:
    """One rank's operators and exchange plan on the folded row numbering ``[owned | per peer q != rank, ascending: q's rows
    of the two-ring halo (ascending global id), PAD_ROWS pad rows]``.  The pad rows of a peer's segment carry that peer's
    BatchNorm statistics of the exchanged tensor: the statistics all-gather rides in the halo exchange (57 -> 44 collectives
    per iteration), and a receive lands in place -- rows ``[n_own:]`` of the ``[n_ext, C]`` buffer ARE the receive buffer."""

    def __init__(self, g: "DistMeshGraph"):
        dst, src, e_own, e_wide, dis, halo, send_per_peer = g._fold_src
        dev, W, rank, n = g.device, g.world, g.rank, g.n_own
        self.graph, self.world, self.rank, self.group, self.n_own = g, W, rank, g.group, n
        self._comm = None
        self._send_rows = self._send_pad = None
        V = g.num_vertices_global
        ext_of = torch.full((V,), -1, dtype=torch.long, device=dev)
        ext_of[g.start:g.end] = torch.arange(n, device=dev)
        b = torch.tensor(g.bounds, device=dev, dtype=torch.long)
        owner = torch.bucketize(halo, b[1:], right=True)
        at, ext_src, stats_rows = n, [torch.arange(g.start, g.end, device=dev)], []
        send_index, self.send_splits, self.recv_splits = [], [], []
        pad = torch.full((PAD_ROWS,), -1, dtype=torch.long, device=dev)
        for q in range(W):
            if q == rank:
                stats_rows.append(-1)
                self.send_splits.append(0)
                self.recv_splits.append(0)
                continue
            mine = halo[owner == q]
            ext_of[mine] = at + torch.arange(mine.numel(), device=dev)
            ext_src += [mine, pad]
            stats_rows.append(at + int(mine.numel()))
            at += int(mine.numel()) + PAD_ROWS
            self.recv_splits.append(int(mine.numel()) + PAD_ROWS)
            rows = send_per_peer[q] if send_per_peer[q] is not None else torch.zeros(0, dtype=torch.long, device=dev)
            send_index += [rows, -1 - torch.arange(PAD_ROWS, device=dev)]
            self.send_splits.append(int(rows.numel()) + PAD_ROWS)
        self.n_ext = at
        self.ext_src = torch.cat(ext_src)                          # global (processing-order) vertex per row, -1: pad row
        self.send_index = (torch.cat(send_index) if send_index else torch.zeros(0, dtype=torch.long, device=dev)).to(torch.int32)
        self.n_send = int(self.send_index.numel())
        self.stats_rows = torch.tensor(stats_rows, dtype=torch.int64, device=dev)
        dis_ext = torch.where(self.ext_src >= 0, dis[self.ext_src.clamp(min=0)], torch.zeros((), device=dev))
        self.handle = capi.GraphHandle.from_partition(dst[e_own] - g.start, ext_of[src[e_own]], n, self.n_ext, dis_ext)
        self.handle_wide = capi.GraphHandle.from_partition(ext_of[dst[e_wide]], ext_of[src[e_wide]], self.n_ext, self.n_ext, dis_ext)

    def native_comm(self):
        """The library's own communicator for this layout (capi.Comm) -- a collective call at FIRST use, which PartChain and
        part_blocks make at partition set-up, not inside a forward pass.  ONE RCCL communicator per process group
        (`_base_comm`): the first layout of a group creates it, every later one (the other levels of an MGCN) takes a second
        handle on it with its own per-peer row counts (`sg_comm_share`: no ncclCommInitRank, no buffers).  Returns None when the
        phase path keeps its collectives on torch.distributed."""
        if self._comm is not None:
            return self._comm or None
        self._comm = False
        if not NATIVE_COLLECTIVES or _solo(self.world) or _backend(self.group) != "nccl":
            return None
        base = _base_comm(self)
        if base is None:
            return None
        self._comm = base if base.layout is self else _SharedComm(base.share(self.send_splits, self.recv_splits), base)
        return self._comm

    def halo_of(self, full: torch.Tensor) -> torch.Tensor:
        """Rows ``[n_own:]`` of the folded buffer of a mesh-wide [V, C] tensor in processing order (zeros in the pad rows)."""
        idx = self.ext_src[self.n_own:]
        out = full.index_select(0, idx.clamp(min=0))
        out[idx < 0] = 0
        return out

    def exchange(self, recv: torch.Tensor, send: torch.Tensor) -> None:
        """One all-to-all: ``send`` [n_send, C] (peer segments with their pad rows) -> ``recv`` = rows [n_own:] of a buffer."""
        if _solo(self.world):
            return
        comm = self.native_comm() if (send.is_cuda and send.is_contiguous() and recv.is_contiguous()) else None
        if comm is not None:
            collective_counts["all_to_all"] += 1
            comm.halo_exchange(recv, send)
            return
        _all_to_all_rows(recv, send, self.recv_splits, self.send_splits, self.group)

    def halo_rows(self, x_own: torch.Tensor) -> torch.Tensor:
        """Rows ``[n_own:]`` of the folded buffer of a tensor this rank holds the owned rows of: ONE exchange (the rows the
        peers need, packed with zero pad rows).  No gradient flows through it (see part_blocks)."""
        with torch.no_grad():
            if self._send_rows is None:
                idx = self.send_index.long()
                self._send_rows, self._send_pad = idx.clamp(min=0), (idx < 0)
            send = x_own.detach().index_select(0, self._send_rows) if self.n_send else x_own.new_zeros((0, x_own.shape[1]))
            if self.n_send:
                send[self._send_pad] = 0
            recv = x_own.new_zeros((self.n_ext - self.n_own, x_own.shape[1]))
            self.exchange(recv, send)
        return recv


class PartChain:
    """The 13 [ChebConv -> BatchNorm -> LeakyReLU] blocks of SingleScaleGCN (util/networks.py:83-101) on one rank of a vertex
    partition, run phase by phase BELOW the C ABI (sg_block_run, csrc/block.hip) with the rank's collectives between the
    calls: per block forward ONE all-to-all (rows of the conv output + this rank's BatchNorm statistics in the pad rows;
    the block behind applies BatchNorm + activation to owned and received rows alike), backward one all-reduce (the two
    BatchNorm sums) and one all-to-all (gradient rows).  One autograd node for the whole run.

    A rank is bound by its host, so nothing is set up twice: the chain keeps SETS of buffers with their descriptor arrays
    filled in (``_PartBuffers``); a forward takes a free set (a training loop has one: taken by the forward pass, handed
    back at the end of the backward pass) and writes into its descriptors only what changed since that set's last use."""

    def __init__(self, plans, layout: FoldedLayout):
        from . import functional as F_sg
        self.plans, self.lay = list(plans), layout
        if dist.is_initialized():
            layout.native_comm()        # (a collective call: at set-up, where every rank builds its chains in the same order)
        self._ws = {}
        self._free = {}                 # dtype -> [_PartBuffers, ..] not in use
        self._F = F_sg

    def workspace(self, dtype) -> int:
        code = capi._DTYPES[dtype]
        wsb = self._ws.get(code)
        if wsb is None:
            lay = self.lay
            probe = capi.sg_block()
            wsb = 0
            for p in self.plans:
                p.init_descriptor(probe)
                probe.graph, probe.graph_wide = lay.handle._h, lay.handle_wide._h
                probe.dtype, probe.V, probe.V_out, probe.V_ext, probe.world, probe.ldh = code, lay.n_own, lay.n_own, lay.n_ext, lay.world, p.Cout
                wsb = max(wsb, capi.block_workspace(probe, 2))
            self._ws[code] = wsb
        return wsb

    def take(self, dtype, dev) -> "_PartBuffers":
        free = self._free.setdefault(dtype, [])
        return free.pop() if free else _PartBuffers(self, dtype, dev)

    def give_back(self, b: "_PartBuffers") -> None:
        free = self._free.setdefault(b.dtype, [])
        if len(free) < 2:
            free.append(b)


def _up256(n: int) -> int:
    return (n + 255) & ~255


class _PartBuffers:
    """One set of everything a forward + backward pass of a PartChain touches: the activation arena, the gradient arena
    (allocated at the first backward pass), the scratch, the tensor views the collectives take, and the descriptor arrays
    -- one array per foreign call between two collectives -- with every field that does not change from call to call
    already in place."""

    def __init__(self, pc: PartChain, dtype, dev):
        self.pc, self.dtype, self.dev = pc, dtype, dev
        plans, lay = pc.plans, pc.lay
        n = len(plans)
        V, Ve, W = lay.n_own, lay.n_ext, lay.world
        e = 4 if dtype == torch.float32 else 2
        mk = lambda k: (capi.sg_block * k)()
        # descriptor arrays, one per call between two collectives
        self.f_seg = [mk(1)] + [mk(2) for _ in range(n - 1)]        # [conv 0] ; [bn i-1, conv i]
        self.f_tail = mk(1)                                          # [bn n-1]
        self.b_head = mk(1)                                          # [reduce n-1]
        self.b_a = [mk(1) for _ in range(n)]                         # [bn-apply + conv first half] of block i
        self.b_b = [mk(1)] + [mk(2) for _ in range(n - 1)]           # [conv second half i, reduce i-1]
        # ---- forward arena: per block its input [Ve, K*Cin | Cin], conv output H [Ve, Cout], the small fp32 vectors, send rows
        at, offs = 0, []
        for p in plans:
            o_in = at
            at += _up256(Ve * (p.K * p.Cin if p.order == 0 else p.Cin) * e)
            o_h = at
            at += _up256(Ve * p.Cout * e)
            o_small = at
            at += _up256((4 * p.Cout + (2 * p.Cout + 1) * (W + 1) + 1) * 4)
            o_send = at
            at += _up256(lay.n_send * p.Cout * e)
            offs.append((o_in, o_h, o_small, o_send))
        self.fwd = torch.empty(max(at, 256), dtype=torch.uint8, device=dev)
        self.ws = torch.empty(max(pc.workspace(dtype), 256), dtype=torch.uint8, device=dev)

        def view(arena, off, rows, cols, dt):
            es = 4 if dt == torch.float32 else 2
            return arena[off:off + rows * cols * es].view(dt).view(rows, cols)
        self.inp = [view(self.fwd, o[0], Ve, (p.K * p.Cin if p.order == 0 else p.Cin), dtype) for p, o in zip(plans, offs)]
        self.H = [view(self.fwd, o[1], Ve, p.Cout, dtype) for p, o in zip(plans, offs)]
        self.small = [self.fwd[o[2]:o[2] + (4 * p.Cout + (2 * p.Cout + 1) * (W + 1) + 1) * 4].view(torch.float32) for p, o in zip(plans, offs)]
        self.send = [view(self.fwd, o[3], lay.n_send, p.Cout, dtype) for p, o in zip(plans, offs)]
        self.recv = [h[V:] for h in self.H]                         # rows [V:] of H ARE the receive buffer of the exchange
        self.in_own, self.in_halo = self.inp[0][:V, :plans[0].Cin], self.inp[0][V:, :plans[0].Cin]
        C = plans[-1].Cout
        self.last_local = self.small[-1][4 * C:4 * C + 2 * C + 1].view(1, -1)
        self.last_gathered = self.small[-1][4 * C + 2 * C + 1:4 * C + (2 * C + 1) * (W + 1)].view(W, 2 * C + 1)
        self.bwd = None
        self.marks = None               # fingerprints of the plans the descriptors were bound to
        self.training = None
        self.accs = None
        self.y_ptr = self.dy_ptr = None
        self.steps = [None, None]
        self._fill_forward()

    def schedule(self, backward: bool):
        """The steps of one pass for ``capi.part_run``: the descriptor arrays of this set with the collectives between
        them, in the order _PartChainFn issues them one by one on the torch.distributed path."""
        if self.steps[backward] is not None:
            return self.steps[backward]
        plans, n = self.pc.plans, len(self.pc.plans)
        e = 4 if self.dtype == torch.float32 else 2
        ent = []
        if not backward:
            for i, p in enumerate(plans):
                ent.append((capi.STEP_BLOCKS, self.f_seg[i], 1 if i == 0 else 2, None, None))
                if i + 1 < n:
                    ent.append((capi.STEP_EXCHANGE, None, p.Cout * e, self.send[i], self.recv[i]))
            C = plans[-1].Cout
            ent.append((capi.STEP_ALL_GATHER, None, (2 * C + 1) * 4, self.last_local, self.last_gathered))
            ent.append((capi.STEP_BLOCKS, self.f_tail, 1, None, None))
        else:
            ent.append((capi.STEP_BLOCKS, self.b_head, 1, None, None))
            for i in range(n - 1, -1, -1):
                p = plans[i]
                width = 2 * p.Cin if p.order == 0 else p.Cout
                ent.append((capi.STEP_ALL_REDUCE, None, 2 * p.Cout, None, self.sums[i]))
                ent.append((capi.STEP_BLOCKS, self.b_a[i], 1, None, None))
                ent.append((capi.STEP_EXCHANGE, None, width * e, self.gsend[i], self.grecv[i]))
                ent.append((capi.STEP_BLOCKS, self.b_b[i], 1 if i == 0 else 2, None, None))
        arr = (capi.sg_part_step * len(ent))()
        for st, (kind, blocks, cnt, send, recv) in zip(arr, ent):
            st.kind, st.n = kind, cnt
            if blocks is not None:
                st.blocks = ctypes.cast(blocks, ctypes.POINTER(capi.sg_block))
            st.send = send.data_ptr() if send is not None else None
            st.recv = recv.data_ptr() if recv is not None else None
        self.steps[backward] = arr
        return arr

    def descriptors(self, i):
        """Every descriptor that describes block i (they all get the same static fields)."""
        n = len(self.pc.plans)
        out = [self.f_seg[i][0 if i == 0 else 1], self.b_a[i][0], self.b_b[i][0]]
        out.append(self.f_seg[i + 1][0] if i + 1 < n else self.f_tail[0])
        out.append(self.b_b[i + 1][1] if i + 1 < n else self.b_head[0])
        return out

    def _fill_forward(self) -> None:
        pc, lay, dtype = self.pc, self.pc.lay, self.dtype
        plans, n = pc.plans, len(pc.plans)
        code = capi._DTYPES[dtype]
        V, Ve, W = lay.n_own, lay.n_ext, lay.world
        wsb = pc.workspace(dtype)
        for i, p in enumerate(plans):
            C = p.Cout
            s0 = self.small[i].data_ptr()
            stats, local, gathered, count = s0, s0 + 16 * C, s0 + 16 * C + 4 * (2 * C + 1), s0 + 16 * C + 4 * (2 * C + 1) * (W + 1)
            inp = self.inp[i].data_ptr()
            for blk in self.descriptors(i):
                p.init_descriptor(blk)
                blk.graph, blk.graph_wide = lay.handle._h, lay.handle_wide._h
                blk.dtype, blk.V, blk.V_ext, blk.world = code, V, Ve, W
                blk.V_out = Ve if i + 1 < n else V
                if p.order == 0:
                    blk.T, blk.ldt, blk.X, blk.ldx = inp, p.K * p.Cin, inp, p.K * p.Cin
                else:
                    blk.T, blk.ldt, blk.X, blk.ldx = None, 0, inp, p.Cin
                blk.H, blk.ldh, blk.stats, blk.local, blk.gathered, blk.count = self.H[i].data_ptr(), C, stats, local, gathered, count
                blk.gathered_ready = 0
                blk.stats_rows = lay.stats_rows.data_ptr()
                blk.send_index, blk.n_send, blk.send = lay.send_index.data_ptr(), (lay.n_send if i + 1 < n else 0), self.send[i].data_ptr()
                if i + 1 < n:
                    q = plans[i + 1]
                    blk.Y, blk.ldy = self.inp[i + 1].data_ptr(), (q.K * q.Cin if q.order == 0 else q.Cin)
                blk.ws, blk.ws_bytes = self.ws.data_ptr(), wsb
        # the phase of every descriptor is a property of its place in the arrays
        self.f_seg[0][0].phase = capi.PHASE_CONV
        for i in range(1, n):
            self.f_seg[i][0].phase, self.f_seg[i][1].phase = capi.PHASE_BN, capi.PHASE_CONV
        self.f_tail[0].phase, self.f_tail[0].gathered_ready = capi.PHASE_BN, 1
        self.f_tail[0].ldy = plans[-1].Cout

    def bind(self) -> None:
        """Parameter addresses and packed-weight buffers, again whenever a plan's fingerprint moved."""
        marks = tuple(p.fingerprint() for p in self.pc.plans)
        if marks == self.marks:
            return
        for i, p in enumerate(self.pc.plans):
            for blk in self.descriptors(i):
                p.bind(blk, self.dtype, self.dev)
        self.marks = marks

    def set_training(self, training) -> None:
        if training == self.training:
            return
        for i, t in enumerate(training):
            for blk in self.descriptors(i):
                blk.training = t
        self.training = training

    def backward_buffers(self) -> None:
        """The gradient arena and the backward halves of the descriptors, at the first backward pass of this set."""
        if self.bwd is not None:
            return
        pc, lay, dtype, dev = self.pc, self.pc.lay, self.dtype, self.dev
        plans, n = pc.plans, len(pc.plans)
        V, Ve = lay.n_own, lay.n_ext
        e = 4 if dtype == torch.float32 else 2
        at, offs = 0, []
        for p in plans:
            width = 2 * p.Cin if p.order == 0 else p.Cout
            o = []
            for nbytes in (Ve * p.K * (p.Cin if p.order == 0 else p.Cout) * e, 6 * p.Cout * 4, p.Cout * p.K * p.Cin * 4,
                           lay.n_send * width * e, (Ve - V) * width * e, V * p.Cin * e):
                o.append(at)
                at += _up256(nbytes)
            offs.append(o)
        self.bwd = torch.empty(max(at, 256), dtype=torch.uint8, device=dev)

        def view(off, rows, cols, dt):
            es = 4 if dt == torch.float32 else 2
            return self.bwd[off:off + rows * cols * es].view(dt).view(rows, cols)
        self.G, self.dvec, self.dW, self.gsend, self.grecv, self.dx = [], [], [], [], [], []
        for p, o in zip(plans, offs):
            width = 2 * p.Cin if p.order == 0 else p.Cout
            self.G.append(view(o[0], Ve, p.K * (p.Cin if p.order == 0 else p.Cout), dtype))
            self.dvec.append(view(o[1], 6, p.Cout, torch.float32))
            self.dW.append(self.bwd[o[2]:o[2] + p.Cout * p.K * p.Cin * 4].view(torch.float32))
            self.gsend.append(view(o[3], lay.n_send, width, dtype))
            self.grecv.append(view(o[4], Ve - V, width, dtype))
            self.dx.append(view(o[5], V, p.Cin, dtype))
        self.sums = [d.view(-1)[:2 * p.Cout] for d, p in zip(self.dvec, plans)]       # what the all-reduce of a block carries
        for i, p in enumerate(plans):
            d = self.descriptors(i)
            for blk in (d[1], d[2], d[4]):          # the descriptors of the backward calls: BWD_A, BWD_B, BWD_REDUCE of block i
                if i + 1 < n:
                    blk.dY, blk.lddy = self.dx[i + 1].data_ptr(), p.Cout
                else:
                    blk.lddy = p.Cout
                blk.dX, blk.lddx, blk.need_dx = self.dx[i].data_ptr(), p.Cin, 1
                blk.dW, blk.dvec, blk.G = self.dW[i].data_ptr(), self.dvec[i].data_ptr(), self.G[i].data_ptr()
                # (send / recv / n_send serve the gradient-row exchange here, the rows of H in the forward descriptors)
                blk.send, blk.recv, blk.n_send = self.gsend[i].data_ptr(), self.grecv[i].data_ptr(), lay.n_send
        self.b_head[0].phase = capi.PHASE_BWD_REDUCE
        for i in range(n):
            self.b_a[i][0].phase = capi.PHASE_BWD_A
            self.b_b[i][0].phase = capi.PHASE_BWD_B
            if i > 0:
                self.b_b[i][1].phase = capi.PHASE_BWD_REDUCE


class _PartChainFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pc: PartChain, x_own: torch.Tensor, x_halo: torch.Tensor, *params):
        F_sg, lay, plans = pc._F, pc.lay, pc.plans
        n, dev, dtype = len(plans), x_own.device, x_own.dtype
        V, Ve, W = lay.n_own, lay.n_ext, lay.world
        stream = capi._stream(x_own)
        b = pc.take(dtype, dev)
        b.bind()
        training = tuple(1 if p.bn.training else 0 for p in plans)
        b.set_training(training)
        # the network input on all rows: owned rows from the autograd side, halo rows computed from the halo copies of z1 / dm
        b.in_own.copy_(x_own)
        if Ve > V:
            b.in_halo.copy_(x_halo)
        for i, p in enumerate(plans):
            d0 = b.f_seg[i][0 if i == 0 else 1]
            d0.refresh_weights = p.stale(dtype, False, int(d0.V))
        y = torch.empty((V, plans[-1].Cout), dtype=dtype, device=dev)
        b.f_tail[0].Y = y.data_ptr()
        comm = lay.native_comm() if all(training) else None
        if comm is not None:
            # ONE foreign call: the 14 runs of phases with the 12 exchanges and the statistics all-gather enqueued between them
            steps = b.schedule(False)
            _assert_c10d_drained()
            capi.part_run(comm, steps, len(steps), stream, dev)
            collective_counts["all_to_all"] += n - 1
            collective_counts["all_gather"] += 1
            native_runs[0] += 1
        else:
            for i in range(n):
                capi.block_run(b.f_seg[i], 1 if i == 0 else 2, stream, dev)
                if i + 1 < n:
                    lay.exchange(b.recv[i], b.send[i])            # rows of H_i + this rank's statistics -> the peers, in place
            # the last BatchNorm has no exchange behind it: a plain all-gather of the statistics
            if training[-1]:
                if _solo(W):
                    b.last_gathered.copy_(b.last_local)
                else:
                    _all_gather_rows(b.last_gathered, b.last_local, lay.group)
            capi.block_run(b.f_tail, 1, stream, dev)
        F_sg.block_calls[0] += n
        ctx.pc, ctx.params = pc, params
        if any(ctx.needs_input_grad):
            ctx.b = b                  # the set stays with this pass until its backward pass has run
        else:
            ctx.b = None
            pc.give_back(b)
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, dy: torch.Tensor):
        pc, b = ctx.pc, ctx.b
        if b is None:
            raise RuntimeError("part_chain: a second backward pass through the same forward (its buffers went back to the chain)")
        F_sg, lay, plans = pc._F, pc.lay, pc.plans
        n, dev = len(plans), dy.device
        W = lay.world
        stream = capi._stream(dy)
        dy = dy.contiguous()
        b.backward_buffers()
        sinking = bool(F_sg._sink_depth)
        sunk, accs = [], []
        for p in plans:
            aw = tuple(F_sg._grad_acc(w, dev) if sinking else None for w in p.weights)
            sw = all(a is not None for a in aw)
            ab = F_sg._grad_acc(p.cbias, dev) if sinking else None
            ag, at = (F_sg._grad_acc(p.gamma, dev), F_sg._grad_acc(p.beta, dev)) if sinking else (None, None)
            sb = ag is not None and at is not None
            sunk.append((sw, ab is not None, sb))
            accs.append((aw if sw else None, ab, (ag, at) if sb else None))
        if accs != b.accs:
            for i, p in enumerate(plans):
                aw, ab, gb = accs[i]
                for blk in b.descriptors(i):
                    for k in range(3):
                        blk.acc_W[k] = aw[k] if (aw is not None and k < p.K) else None
                    blk.acc_bias = ab
                    blk.acc_gamma, blk.acc_beta = gb if gb is not None else (None, None)
            b.accs = accs
        if dy.data_ptr() != b.dy_ptr:
            for blk in b.descriptors(n - 1):
                blk.dY = dy.data_ptr()
            b.dy_ptr = dy.data_ptr()
        # BatchNorm n-1: this rank's sums -> all-reduce; then per block: [dH, dW, gradient blocks] -> all-to-all ->
        # [recurrence unwound -> dX ; sums of the BatchNorm in front] -> all-reduce
        bn_local = [None] * n
        # (BatchNorm gradients that autograd wants as tensors are this rank's PARTIAL sums, read between a block's reduce
        #  phase and its all-reduce: such a pass takes the call-by-call path)
        comm = lay.native_comm() if all(sk[2] for sk in sunk) else None
        if comm is not None:
            steps = b.schedule(True)
            _assert_c10d_drained()
            capi.part_run(comm, steps, len(steps), stream, dev)
            collective_counts["all_reduce"] += n
            collective_counts["all_to_all"] += n
            native_runs[1] += 1
        else:
            capi.block_run(b.b_head, 1, stream, dev)
            for i in range(n - 1, -1, -1):
                if not sunk[i][2]:
                    bn_local[i] = b.dvec[i][:2].clone()     # this rank's partial (sum dz, sum dz xhat): the gradients autograd gets
                if not _solo(W):
                    _all_reduce(b.sums[i], dist.ReduceOp.SUM, lay.group)
                capi.block_run(b.b_a[i], 1, stream, dev)
                lay.exchange(b.grecv[i], b.gsend[i])
                capi.block_run(b.b_b[i], 1 if i == 0 else 2, stream, dev)
        F_sg.block_calls[1] += n
        grads_out = []
        for i, p in enumerate(plans):
            K, Cin, Cout = p.K, p.Cin, p.Cout
            sw, sbias, sbn = sunk[i]
            grads_out.append(None if (p.cbias is None or sbias) else b.dvec[i][5].clone().to(p.cbias.dtype))
            if sw:
                grads_out.extend([None] * K)
            elif p.order == 0:
                dWm = b.dW[i].view(Cout, K * Cin).clone()
                grads_out.extend(dWm[:, k * Cin:(k + 1) * Cin] for k in range(K))
            else:
                dWm = b.dW[i].view(K * Cout, Cin).clone()
                grads_out.extend(dWm[k * Cout:(k + 1) * Cout] for k in range(K))
            grads_out.extend((None, None) if sbn else (bn_local[i][1].to(p.gamma.dtype), bn_local[i][0].to(p.gamma.dtype)))
        dx0 = b.dx[0].clone()                                    # (the set goes back to the chain: its buffers are reused)
        ctx.b = None
        pc.give_back(b)
        return (None, dx0, None, *grads_out)


def prepare_native_comm(graphs) -> None:
    """At partition SET-UP (never inside a forward pass, where a failure would burn an attempt's time limit mid-iteration):
    create the process group's library communicator and every level's handle on it, in level order -- collective calls that all
    ranks make alike.  A no-op on CPU tensors, gloo groups and with SEMIGCN_DIST_NATIVE=0."""
    if not (dist.is_initialized() and NATIVE_COLLECTIVES):
        return
    for g in graphs:
        if g.device.type != "cuda" or _solo(g.world) or _backend(g.group) != "nccl":
            return
        if min(g.bounds[q + 1] - g.bounds[q] for q in range(len(g.bounds) - 1)) < 2:
            continue                    # (part_blocks refuses this level on every rank)
        g.folded().native_comm()


def part_chain(sequentials, graph: "DistMeshGraph", x_own: torch.Tensor, x_halo: torch.Tensor):
    """Blocks that START every Sequential of ``sequentials`` (each [ChebConv, BatchNorm1d, LeakyReLU (, ...)] one or more
    times) on this rank's rows, phase by phase below the C ABI; returns (activation of the last block on the owned rows, index
    of the first entry of the last Sequential that was NOT run), or None when the run cannot take this path (then the caller's
    per-module path does)."""
    from . import functional as F_sg
    if not (x_own.is_cuda and F_sg.blocks_enabled() and dist.is_initialized()):
        return None
    plans = []
    sequentials = list(sequentials)
    tail = 0
    for k, seq in enumerate(sequentials):
        lead = seq._leading_blocks() if hasattr(seq, "_leading_blocks") else None
        if lead is None or not lead[0]:
            return None
        if k + 1 < len(sequentials) and lead[1] != len(seq):
            return None              # modules behind the blocks of an INNER Sequential (a Dropout, a pooled conv) would be skipped
        plans.extend(lead[0])
        tail = lead[1]
    y = part_blocks(plans, graph, x_own, x_halo)
    return None if y is None else (y, tail)


def part_blocks(plans, graph: "DistMeshGraph", x_own: torch.Tensor, x_halo: Optional[torch.Tensor] = None):
    """A run of consecutive [ChebConv -> BatchNorm1d -> LeakyReLU] blocks (functional.BlockPlan, no pool between conv and
    BatchNorm) on one partitioned graph, phase by phase below the C ABI (``PartChain``); ``x_halo``: the halo rows of the input
    in the folded numbering, fetched with ONE exchange when the caller has none (forward-only copies: the backward pass is
    owner-computes on exchanged gradient rows).  None when the run cannot take this path."""
    from . import functional as F_sg
    if not (x_own.is_cuda and F_sg.blocks_enabled() and dist.is_initialized()) or not plans:
        return None
    # every refusal below is decided from what ALL ranks know alike (dtype, the plans, the partition's block bounds): a rank that
    # left this path alone would issue other collectives than its peers.  The bounds are global knowledge: a level at which SOME
    # rank owns fewer than two rows (the coarsest level of a small MGCN on many ranks) is refused by every rank, before the halo
    # exchange below issues a collective
    if x_own.dtype not in (torch.float32, torch.bfloat16):
        return None
    bounds = graph.bounds
    if min(bounds[q + 1] - bounds[q] for q in range(len(bounds) - 1)) < 2:
        return None
    cin = x_own.shape[1]
    for p in plans:
        p.fingerprint()
        bn = p.bn
        vec = 4 if x_own.dtype == torch.float32 else 8
        if p.K != 3 or p.pool is not None or p.Cin != cin or p.Cout % vec or not bn.training or not getattr(bn, "sg_mesh_wide", False) \
                or not (bn.affine and bn.track_running_stats and bn.momentum is not None) \
                or not all(F_sg._f32_dev(w, x_own.device) for w in p.weights) or p.Cout * (2 if vec == 8 else 4) * PAD_ROWS < (2 * p.Cout + 1) * 4:
            return None
        cin = p.Cout
    lay = graph.folded()
    if x_own.shape[0] != lay.n_own:
        return None
    if x_halo is None:
        x_halo = lay.halo_rows(x_own)
    key = tuple(id(p) for p in plans)
    pc = graph.__dict__.setdefault("_part_chains", {}).get(key)
    if pc is None:
        pc = graph.__dict__["_part_chains"][key] = PartChain(plans, lay)
    params = []
    for p in plans:
        params.extend(p.param_tuple)
    return _PartChainFn.apply(pc, x_own, x_halo, *params)


# --------------------------------------------------------------------------------------
# ChebConv on a partition
# --------------------------------------------------------------------------------------
class _DistChebConvFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, g: DistMeshGraph, cache, moments, x, bias, *weights):
        K, n = len(weights), g.n_own
        C = x.shape[1]
        from .functional import _wcat_pair, dense_nt
        wcat, wcat_t = (_wcat_pair(weights, x.dtype) if cache is None
                        else cache.get("cat", x.dtype, weights, None, lambda: _wcat_pair(weights, x.dtype)))
        from .functional import _adopt_wide
        T = _adopt_wide(x, K, g.n_ext) if K > 1 else None     # the fused BatchNorm in front may have written x in place
        fresh = T is None
        if fresh:
            T = torch.empty((g.n_ext if K > 1 else n, K * C), dtype=x.dtype, device=x.device)
        blk = [T[:, k * C:(k + 1) * C] for k in range(K)]
        if fresh:
            blk[0][:n].copy_(x)
        if K == 3:       # ONE exchange (two rings of x): Tx1 on owned + ring-1 rows, Tx2 on the owned rows
            g.exchange_and_aggregate_wide(blk[0], blk[0], blk[1], alpha=1.0)
            g.aggregate(blk[1], blk[2][:n], alpha=2.0, X0=blk[0][:n], beta=-1.0)
        else:
            if K > 1:
                g.exchange(blk[0])
                g.aggregate(blk[0], blk[1][:n], alpha=1.0)
            for k in range(2, K):
                g.exchange(blk[k - 1])
                g.aggregate(blk[k - 1], blk[k][:n], alpha=2.0, X0=blk[k - 2][:n], beta=-1.0)
        out = dense_nt(T[:n], wcat, bias, moments=moments)    # (+ this rank's per-tile BatchNorm moments)
        ctx.g, ctx.K, ctx.C = g, K, C
        ctx.has_bias, ctx.param_dtype = bias is not None, weights[0].dtype
        ctx.wcat_t = wcat_t
        ctx.params = (bias, *weights)
        ctx.save_for_backward(T, wcat)
        return out

    @staticmethod
    def backward(ctx, dout):
        from .functional import _sink, column_sums, dense_nn, dense_nt, weight_grad
        T, wcat = ctx.saved_tensors
        g, K, C, n = ctx.g, ctx.K, ctx.C, ctx.g.n_own
        dout = dout.contiguous()
        need_x, need_b = ctx.needs_input_grad[3], ctx.needs_input_grad[4]
        dws = [None] * K
        if any(ctx.needs_input_grad[5:]):
            dwcat = weight_grad(dout, T[:n]).to(ctx.param_dtype)       # partial: summed over ranks later
            dws = [dwcat[:, k * C:(k + 1) * C] for k in range(K)]
        db = column_sums(dout).to(ctx.param_dtype) if (ctx.has_bias and need_b) else None
        dx = None
        if need_x:
            dT = torch.empty((g.n_ext if K > 1 else n, K * C), dtype=dout.dtype, device=dout.device)
            if ctx.wcat_t is not None:
                dense_nt(dout, ctx.wcat_t, out=dT[:n])
            else:
                dense_nn(dout, wcat, out=dT[:n])
            if K == 1:
                dx = dT
            else:
                gk = [dT[:, k * C:(k + 1) * C] for k in range(K)]
                dx = torch.empty((n, C), dtype=dout.dtype, device=dout.device)
                if K == 3:   # ONE exchange of the [g1 | g2] column blocks on both rings; g1 += 2 L g2 on owned + ring 1
                    g.exchange_and_aggregate_wide(dT[:, C:3 * C], gk[2], gk[1], alpha=2.0, X0=gk[1], beta=1.0)
                    g.aggregate(gk[1], dx, alpha=1.0, X0=gk[0][:n], beta=1.0, X1=gk[2][:n], gamma=-1.0)
                else:
                    for k in range(K - 2, 0, -1):
                        g.exchange(gk[k + 1])
                        x1 = gk[k + 2][:n] if k + 2 <= K - 1 else None
                        g.aggregate(gk[k + 1], gk[k][:n], alpha=2.0, X0=gk[k][:n], beta=1.0, X1=x1, gamma=-1.0)
                    g.exchange(gk[1])
                    x1 = gk[2][:n] if K >= 3 else None
                    g.aggregate(gk[1], dx, alpha=1.0, X0=gk[0][:n], beta=1.0, X1=x1, gamma=-1.0)
        if dout.is_cuda and _sink(ctx.params, (db, *dws)):      # this rank's partial sums, straight into the .grad accumulators
            return (None, None, None, dx) + (None,) * (K + 1)
        return (None, None, None, dx, db, *dws)


class _DistChebConvPostFn(torch.autograd.Function):
    """functional._ChebConvPostFn (GEMM first, Clenshaw aggregation after; for Cout < Cin) on a partition:
    the halo rows exchanged are Cout wide instead of Cin wide."""

    @staticmethod
    def forward(ctx, g: DistMeshGraph, cache, x, bias, *weights):
        K, n, Co = len(weights), g.n_own, weights[0].shape[0]
        from .functional import _wstack_set, dense_nt

        def build():      # the bias rides in on Z_0 (see functional._ChebConvPostFn)
            return _wstack_set(weights, bias, x.dtype)
        wstack, bias_k, wstack_t = build() if cache is None else cache.get("stack", x.dtype, weights, bias, build)
        x = x if x.stride(1) == 1 else x.contiguous()
        Z = torch.empty((g.n_ext, K * Co), dtype=x.dtype, device=x.device)
        dense_nt(x, wstack, bias_k, out=Z[:n])
        z = [Z[:, k * Co:(k + 1) * Co] for k in range(K)]
        out = torch.empty((n, Co), dtype=x.dtype, device=x.device)
        if K == 3:       # ONE exchange of [Z1 | Z2] on both rings; b1 = Z1 + 2 L Z2 on owned + ring 1
            g.exchange_and_aggregate_wide(Z[:, Co:3 * Co], z[2], z[1], alpha=2.0, X0=z[1], beta=1.0)
            g.aggregate(z[1], out, alpha=1.0, X0=z[0][:n], beta=1.0, X1=z[2][:n], gamma=-1.0)
        else:
            for k in range(K - 2, 0, -1):
                g.exchange(z[k + 1])
                x1 = z[k + 2][:n] if k + 2 <= K - 1 else None
                g.aggregate(z[k + 1], z[k][:n], alpha=2.0, X0=z[k][:n], beta=1.0, X1=x1, gamma=-1.0)
            g.exchange(z[1])
            g.aggregate(z[1], out, alpha=1.0, X0=z[0][:n], beta=1.0, X1=z[2][:n] if K >= 3 else None, gamma=-1.0)
        ctx.g, ctx.K, ctx.Co = g, K, Co
        ctx.has_bias, ctx.param_dtype = bias is not None, weights[0].dtype
        ctx.wstack_t = wstack_t
        ctx.params = (bias, *weights)
        ctx.save_for_backward(x, wstack)
        return out

    @staticmethod
    def backward(ctx, dout):
        from .functional import _sink, column_sums, dense_nn, dense_nt, weight_grad
        x, wstack = ctx.saved_tensors
        g, K, Co, n = ctx.g, ctx.K, ctx.Co, ctx.g.n_own
        dout = dout.contiguous()
        G = torch.empty((g.n_ext, K * Co), dtype=dout.dtype, device=dout.device)
        gk = [G[:, k * Co:(k + 1) * Co] for k in range(K)]
        gk[0][:n].copy_(dout)
        if K == 3:       # one exchange (two rings of dOut), as in _DistChebConvFn.forward
            g.exchange_and_aggregate_wide(gk[0], gk[0], gk[1], alpha=1.0)
            g.aggregate(gk[1], gk[2][:n], alpha=2.0, X0=gk[0][:n], beta=-1.0)
        else:
            g.exchange(gk[0])
            g.aggregate(gk[0], gk[1][:n], alpha=1.0)
            for k in range(2, K):
                g.exchange(gk[k - 1])
                g.aggregate(gk[k - 1], gk[k][:n], alpha=2.0, X0=gk[k - 2][:n], beta=-1.0)
        own = G[:n]
        dx = None
        if ctx.needs_input_grad[2]:
            dx = dense_nt(own, ctx.wstack_t) if ctx.wstack_t is not None else dense_nn(own, wstack)
        dws = [None] * K
        if any(ctx.needs_input_grad[4:]):
            dwstack = weight_grad(own, x.contiguous()).to(ctx.param_dtype)
            dws = [dwstack[k * Co:(k + 1) * Co] for k in range(K)]
        db = column_sums(dout).to(ctx.param_dtype) if (ctx.has_bias and ctx.needs_input_grad[3]) else None
        if dout.is_cuda and _sink(ctx.params, (db, *dws)):
            return (None, None, dx) + (None,) * (K + 1)
        return (None, None, dx, db, *dws)


def dist_cheb_conv(g: DistMeshGraph, x, weights, bias=None, cache=None, moments=None):
    """``moments``: as in functional.cheb_conv -- filled with this rank's per-row-tile moments when the layer's last step
    is the MFMA product; the mesh-wide BatchNorm behind it merges them with the other ranks' (functional._BNActFn)."""
    from . import functional as F_sg
    if x.shape[0] != g.n_own:
        raise ValueError(f"x has {x.shape[0]} rows but this rank owns {g.n_own} vertices")
    if F_sg.AGGREGATE_AFTER_GEMM_WHEN_NARROWING and len(weights) >= 2 and weights[0].shape[0] < weights[0].shape[1]:
        return _DistChebConvPostFn.apply(g, cache, x, bias, *weights)
    return _DistChebConvFn.apply(g, cache, moments, x, bias, *weights)


# --------------------------------------------------------------------------------------
# BatchNorm over ALL vertices of the mesh
# --------------------------------------------------------------------------------------
class _SyncBNFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, running_mean, running_var, eps, momentum, group):
        xf = x.float()
        n = x.shape[0]
        C = x.shape[1]
        var_r, mean_r = torch.var_mean(xf, dim=0, unbiased=False)
        world = dist.get_world_size(group)
        local = torch.cat([mean_r, var_r * n, xf.new_tensor([float(n)])])
        allst = torch.empty((world, 2 * C + 1), dtype=torch.float32, device=x.device)
        _all_gather_rows(allst, local.unsqueeze(0).contiguous(), group)
        cnt = allst[:, 2 * C:]                                   # [world, 1]
        N = cnt.sum()
        mean = (allst[:, :C] * cnt).sum(0) / N
        m2 = (allst[:, C:2 * C] + cnt * (allst[:, :C] - mean) ** 2).sum(0)   # Chan et al. merge
        var = m2 / N
        invstd = torch.rsqrt(var + eps)
        if running_mean is not None:
            with torch.no_grad():
                running_mean.mul_(1 - momentum).add_(mean.to(running_mean.dtype), alpha=momentum)
                running_var.mul_(1 - momentum).add_((m2 / (N - 1)).to(running_var.dtype), alpha=momentum)
        xhat = (xf - mean) * invstd
        y = xhat * weight.float() + bias.float()
        ctx.save_for_backward(xhat, invstd, weight)
        ctx.group, ctx.N = group, N
        return y.to(x.dtype)

    @staticmethod
    def backward(ctx, dy):
        xhat, invstd, weight = ctx.saved_tensors
        dyf = dy.float()
        s1 = dyf.sum(0)
        s2 = (dyf * xhat).sum(0)
        dw, db = s2.clone(), s1.clone()                          # partial sums (reduced with the other grads)
        red = torch.stack([s1, s2])
        _all_reduce(red, dist.ReduceOp.SUM, ctx.group)
        N = ctx.N
        dx = (weight.float() * invstd) * (dyf - red[0] / N - xhat * (red[1] / N))
        return dx.to(dy.dtype), dw.to(weight.dtype), db.to(weight.dtype), None, None, None, None, None


class DistBatchNorm1d(nn.BatchNorm1d):
    """nn.BatchNorm1d whose training statistics span all ranks' vertices (same parameters,
    buffers and state-dict keys)."""
    group = None
    sg_mesh_wide = True      # functional.bn_act merges the moments across ranks for these modules

    def forward(self, x):
        if not self.training or not dist.is_initialized() or _solo(dist.get_world_size(self.group)):
            return super().forward(x)
        if self.num_batches_tracked is not None:
            self.num_batches_tracked.add_(1)
        momentum = self.momentum if self.momentum is not None else 1.0 / float(self.num_batches_tracked)
        return _SyncBNFn.apply(x, self.weight, self.bias, self.running_mean if self.track_running_stats else None,
                               self.running_var if self.track_running_stats else None, self.eps, momentum, self.group)


def convert_batchnorm(model: nn.Module, group=None) -> nn.Module:
    """Switch every BatchNorm1d of ``model`` to mesh-wide statistics, in place."""
    for mod in model.modules():
        if type(mod) is nn.BatchNorm1d:
            mod.__class__ = DistBatchNorm1d
            mod.group = group
    return model


def all_reduce_gradients(params, group=None, flat: Optional[torch.Tensor] = None) -> None:
    """Sum the per-rank partial parameter gradients: one flat bucket, one all-reduce.  ``flat``: the buffer all the
    ``.grad`` tensors are views of (train.GradBuffer) -- reduced in place, no gather / scatter copies."""
    grads = [p.grad for p in params if p.grad is not None]
    if not grads or not dist.is_initialized() or _solo(dist.get_world_size(group)):
        return
    if flat is not None:
        _all_reduce(flat, dist.ReduceOp.SUM, group)
        return
    flat = torch.cat([g.reshape(-1) for g in grads])
    _all_reduce(flat, dist.ReduceOp.SUM, group)
    off = 0
    for g in grads:
        g.copy_(flat[off:off + g.numel()].view_as(g))
        off += g.numel()


class _GlobalMinMax(torch.autograd.Function):
    """(lo_local, hi_local) [1, n] -> mesh-wide (lo, hi): ONE all-reduce (max of [-lo | hi]).  Every rank goes on with
    the replicated bounds, so d loss / d bound is the SUM over ranks of the local terms (one all-reduce in backward); it
    is then handed to ONE rank whose local extreme IS the global one -- the lowest such rank when several ranks hold
    the same extreme value (grid-like meshes do) -- where ``_column_min_max``'s own backward routes it to the arg-extreme
    vertex: one vertex per bound, as torch.min/max(z1, dim=0) does on a single device.  The tie is broken inside the
    backward's all-reduce: every holder adds 2^rank to a per-bound word, the sum is the set of holders (exact in fp32 up
    to 24 ranks), its lowest bit the winner."""

    @staticmethod
    def forward(ctx, lo_l, hi_l, group):
        n = lo_l.shape[1]
        both = torch.cat([-lo_l, hi_l], dim=1).contiguous()
        _all_reduce(both, dist.ReduceOp.MAX, group)
        lo_g, hi_g = -both[:, :n], both[:, n:].clone()
        ctx.group = group
        ctx.save_for_backward(torch.cat([lo_l == lo_g, hi_l == hi_g], dim=1))
        return lo_g, hi_g

    @staticmethod
    def backward(ctx, g_lo, g_hi):
        (own,) = ctx.saved_tensors
        n = g_lo.shape[1]
        rank = dist.get_rank(ctx.group)
        if dist.get_world_size(ctx.group) > 24:
            raise RuntimeError("the holder word of _GlobalMinMax is exact up to 24 ranks")
        g = torch.cat([g_lo, g_hi, own.to(g_lo.dtype) * float(1 << rank)], dim=1).contiguous()
        _all_reduce(g, dist.ReduceOp.SUM, ctx.group)
        holders = g[:, 2 * n:].to(torch.int64)
        mine = ((holders & -holders) == (1 << rank)).to(g.dtype)          # lowest set bit = the lowest holding rank
        return g[:, :n] * mine[:, :n], g[:, n:2 * n] * mine[:, n:], None


def dist_min_max(z1: torch.Tensor, group=None):
    """Per-axis min / max of the whole mesh's z1 ([1, 3] each) from this rank's rows, differentiable like the
    single-device ``torch.min/max(z1, dim=0)`` (see _GlobalMinMax): dz1 of the partitioned run equals the 1-GPU run's."""
    from .networks import _column_min_max
    lo_l, hi_l = _column_min_max(z1)
    if not dist.is_initialized() or _solo(dist.get_world_size(group)):
        return lo_l, hi_l
    return _GlobalMinMax.apply(lo_l, hi_l, group)


# --------------------------------------------------------------------------------------
# per-rank training state
# --------------------------------------------------------------------------------------
@dataclass
class PartitionedMesh:
    """One rank's slice of the training constants (all in processing order, owned rows only
    unless noted)."""
    graph: DistMeshGraph
    z1: torch.Tensor            # [n,3] requires_grad
    x_pos: torch.Tensor         # [n,3]
    faces_ext: torch.Tensor     # [F_own,3] indices into [owned | halo]
    target_pos: torch.Tensor    # [n,3]
    target_fn: torch.Tensor     # [F_own,3]
    v_keep: torch.Tensor        # [n,1]
    f_keep: torch.Tensor        # [F_own,1]
    dummy_masks: torch.Tensor   # [n,M]
    n_v_keep: float             # GLOBAL counts
    n_f_keep: float
    edge_index = None           # SingleScaleGCN reads .graph instead
    # the halo rows' copies of the network's inputs in the folded layout (graph.folded()): with them the first block needs no
    # exchange -- every rank prepares the input rows of its halo itself.  Set by partition_mesh; halo_inputs per iteration.
    z1_halo: Optional[torch.Tensor] = None          # [n_ext - n, 3] (constant: the gradient of z1 is owner-computes)
    dm_halo: Optional[torch.Tensor] = None          # [n_ext - n, M]  v_keep * dummy mask of the halo vertices
    halo_inputs: Optional[tuple] = None             # (z1_halo, this iteration's mask column [n_ext - n, 1])


def partition_mesh(mesh, rank: int, world: int, device, group=None, n_masks: int = 5, seed: int = 317, log=None) -> PartitionedMesh:
    """Cut a synth.SynthMesh (every rank builds the same one) into this rank's share."""
    from . import synth, train
    log = log or (lambda msg: None)
    V = mesh.num_vertices
    pos_all = torch.from_numpy(mesh.x_pos).to(device)
    order, rank_of = _reorder.morton_order(pos_all)
    ei = _reorder.permute_edge_index(torch.from_numpy(mesh.edge_index).to(device), rank_of)
    log("edges in Morton order")
    g = DistMeshGraph(ei, V, rank, world, group)
    log(f"partition plan: {g.n_own} owned + {g.n_halo} halo rows")
    own = order[g.start:g.end]                                   # old ids of my vertices, in my order
    faces = rank_of[torch.from_numpy(mesh.faces).to(device)]     # new ids
    f_mine = (faces[:, 0] >= g.start) & (faces[:, 0] < g.end)
    fo = faces[f_mine]
    faces_ext = g.extended_index(fo)
    vs_all = torch.from_numpy(mesh.vs.astype(np.float32)).to(device)
    tfn_all = train.face_normals(vs_all, torch.from_numpy(mesh.faces).to(device))
    v_keep_all = torch.from_numpy(mesh.v_mask.astype(np.float32)).to(device)
    fa = torch.from_numpy(mesh.faces).to(device)
    f_keep_all = v_keep_all[fa[:, 0]] * v_keep_all[fa[:, 1]] * v_keep_all[fa[:, 2]]
    log("targets and keep masks")
    dm_all = torch.from_numpy(synth.make_dummy_masks(mesh.edge_index, V, dm_size=n_masks, k=4, p=0.014, seed=seed)).to(device)
    log("dummy masks")
    z1_all = torch.from_numpy(mesh.z1).to(device)
    z1 = z1_all[own].clone().requires_grad_(True)
    part = PartitionedMesh(g, z1, pos_all[own].clone(), faces_ext, vs_all[own].clone(), tfn_all[f_mine].clone(),
                           v_keep_all[own].view(-1, 1).clone(), f_keep_all[f_mine].view(-1, 1).clone(),
                           dm_all[own].clone(), float(v_keep_all.sum()), float(f_keep_all.sum()))
    if z1_all.is_cuda:      # the phase-by-phase block path (part_chain) prepares the halo rows of the input itself
        lay = g.folded()
        log("folded layout")
        part.z1_halo = lay.halo_of(z1_all[order]).contiguous()
        part.dm_halo = lay.halo_of((v_keep_all.view(-1, 1) * dm_all)[order]).contiguous()
    return part


class DistSGCNTrainer:
    """SGCNTrainer (semigcn_amd.train, the loop of sgcn.py:118-147) on a vertex partition."""

    def __init__(self, model: nn.Module, part: PartitionedMesh, group=None, lr: float = 0.01, k1: float = 4.0,
                 accumulate: int = 5, phases: bool = True):
        """``phases`` (the default): the 13 blocks run phase by phase below the C ABI with the BatchNorm statistics riding
        in the halo exchange (part_chain: ~50 foreign calls and 44 collectives per iteration).  ``phases=False``: every
        module on its own, the exchange inside each convolution and an all-gather per BatchNorm (57 collectives): the
        supervisor's last-resort fallback (bench.py) and the path MGCN still takes.  Both are eager."""
        self.model, self.part, self.group, self.k1, self.accumulate = model, part, group, k1, accumulate
        self.phases = bool(phases)
        convert_batchnorm(model, group)
        self.params = [p for p in model.parameters()]
        self.opt = torch.optim.Adam(self.params, lr=lr)
        self.sched = torch.optim.lr_scheduler.StepLR(self.opt, step_size=50, gamma=0.5)
        self.iteration = 0
        self.loss_sum = torch.zeros((), device=part.z1.device)
        from .train import GradBuffer
        self.grads = GradBuffer(self.params)
        if self.phases:
            prepare_native_comm([part.graph])

    def _forward_backward(self, dm: torch.Tensor) -> torch.Tensor:
        from .functional import sink_param_grads
        if not self.model.training:
            self.model.train()
        loss = self.loss(self.model(self.part, dm))
        with sink_param_grads():
            loss.backward()
        return loss.detach()

    def loss(self, pos_own: torch.Tensor) -> torch.Tensor:
        from . import train
        p = self.part
        if pos_own.is_cuda and pos_own.dtype == torch.float32:     # fused HIP kernels; halo rows via the exchange
            from .functional import mesh_loss_sums
            s = mesh_loss_sums(halo_extend(p.graph, pos_own), p.faces_ext, p.target_pos, p.v_keep, p.target_fn, p.f_keep)
            s = all_reduce_sum(s, self.group)
            return torch.sqrt(s[0] / p.n_v_keep + 1.0e-6) + self.k1 * (s[1] / p.n_f_keep)
        d = (p.target_pos - pos_own) * p.v_keep
        lp = torch.sqrt(all_reduce_sum((d * d).sum(), self.group) / p.n_v_keep + 1.0e-6)
        pos_ext = halo_extend(p.graph, pos_own)
        fn = train.face_normals(pos_ext, p.faces_ext)
        ln = all_reduce_sum(((fn - p.target_fn).abs() * p.f_keep).sum(), self.group) / p.n_f_keep
        return lp + self.k1 * ln

    def iteration_step(self, mask_index: Optional[int] = None) -> torch.Tensor:
        p = self.part
        k = self.iteration % p.dummy_masks.shape[1] if mask_index is None else mask_index
        dm = p.v_keep * p.dummy_masks[:, k:k + 1]
        if p.dm_halo is not None and self.phases:
            p.halo_inputs = (p.z1_halo, p.dm_halo[:, k:k + 1].contiguous())
        else:
            p.halo_inputs = None
        loss = self._forward_backward(dm)
        self.loss_sum += loss
        self.iteration += 1
        if self.iteration % self.accumulate == 0:
            self.grads.attach()
            all_reduce_gradients(self.params, self.group, flat=self.grads.whole())
            self.opt.step()
            self.grads.zero()
        return loss


@dataclass
class _Job:
    trainer: DistSGCNTrainer
    V_total: int
    E_total: int
    workload: str


def build_partitioned_job(nu: int, nv: int, world: int, rank: int, device, permute: bool = False,
                          dtype=torch.float32, group=None, mesh=None, phases: bool = True, log=None) -> _Job:
    """bench.py's N > 1 leg: the SAME nu x nv mesh as the 1-GPU run, cut into ``world`` blocks
    (strong scaling)."""
    from . import synth
    from .networks import SingleScaleGCN
    log = log or (lambda msg: None)
    if mesh is None:
        mesh = synth.torus_mesh(nu, nv, permute=permute)
    log("partitioning")
    part = partition_mesh(mesh, rank, world, device, group, log=log)
    log("partition done")
    torch.manual_seed(314)
    model = SingleScaleGCN(device).to(device)
    if dtype != torch.float32:
        model.set_feature_dtype(dtype)
    trainer = DistSGCNTrainer(model, part, group, phases=phases)
    halo = torch.tensor([part.graph.n_halo], device=device)
    if world > 1:
        _all_reduce(halo, dist.ReduceOp.MAX, group)
    workload = (f"SGCN train iteration on a closed torus mesh {nu}x{nv} (V={mesh.num_vertices} E={mesh.num_edges}), "
                f"{'fp32' if dtype == torch.float32 else 'bf16'} features, Morton-ordered and vertex-partitioned into "
                f"{world} blocks (<= {int(halo)} halo rows per rank), "
                + ("blocks run phase by phase below the C ABI, BatchNorm statistics carried by the halo exchange, "
                   if trainer.phases else "halo exchange + an all-gather per BatchNorm, ")
                + "gradient all-reduce over RCCL")
    return _Job(trainer, mesh.num_vertices, mesh.num_edges, workload)


# --------------------------------------------------------------------------------------
# MGCN on a partition (SURVEY.md section 8(e) "MGCN", 8(f)-4): every level is cut into blocks;
# pool / unpool read the few members / parents that live on another rank through a row exchange
# --------------------------------------------------------------------------------------
class DistPool:
    """MeshPool / MeshUnpool between two partitioned levels.  ``fine``/``coarse``: the pool_hash pairs
    in PROCESSING numbering (global).  Owner computes: the owner of a coarse vertex averages its
    members, fetching those owned elsewhere (``fine_plan``); the owner of a fine vertex copies its
    parent, fetching it when it lives elsewhere (``coarse_plan``).  Backward passes are the autograd
    transposes: local segment kernels + the reverse row exchange."""

    def __init__(self, fine: torch.Tensor, coarse: torch.Tensor, bounds_f: Sequence[int], bounds_c: Sequence[int],
                 rank: int, world: int, group=None):
        dev = fine.device
        bf = torch.tensor(list(bounds_f), device=dev, dtype=torch.long)
        bc = torch.tensor(list(bounds_c), device=dev, dtype=torch.long)
        owner_f = torch.bucketize(fine, bf[1:], right=True)
        owner_c = torch.bucketize(coarse, bc[1:], right=True)
        self.fine_plan = HaloPlan(owner_c, fine, bounds_f, rank, world, group)      # coarse owners need fine rows
        self.coarse_plan = HaloPlan(owner_f, coarse, bounds_c, rank, world, group)  # fine owners need coarse rows
        mine_c = owner_c == rank
        self.pool_handle = capi.PoolHandle(self.fine_plan.extended_index(fine[mine_c]),
                                           coarse[mine_c] - int(bounds_c[rank]),
                                           self.fine_plan.n_ext, self.coarse_plan.n_own)
        mine_f = owner_f == rank
        self.unpool_handle = capi.PoolHandle(fine[mine_f] - int(bounds_f[rank]),
                                             self.coarse_plan.extended_index(coarse[mine_f]),
                                             self.fine_plan.n_own, self.coarse_plan.n_ext)

    def pool(self, x_own_fine: torch.Tensor) -> torch.Tensor:
        from .functional import mesh_pool
        return mesh_pool(self.pool_handle, halo_extend(self.fine_plan, x_own_fine))

    def unpool(self, x_own_coarse: torch.Tensor) -> torch.Tensor:
        from .functional import mesh_unpool
        return mesh_unpool(self.unpool_handle, halo_extend(self.coarse_plan, x_own_coarse))


@dataclass
class MGCNPartition:
    """This rank's share of an MGCN hierarchy.  ``own_ids[l]``: caller-numbering ids of the level-l
    vertices this rank owns, in processing order (what ``MGCN.forward`` returns rows for)."""
    graphs: List[DistMeshGraph]
    pools: List[DistPool]
    bounds: List[List[int]]
    own_ids: List[torch.Tensor]
    rank_of: List[torch.Tensor]        # per level: processing position of every caller-numbered vertex
    smposs_own: List[torch.Tensor]
    rank: int
    world: int
    group: object = None


def partition_mgcn(model: nn.Module, rank: int, world: int, group=None, phases: bool = True) -> MGCNPartition:
    """Switch ``model`` (semigcn_amd.meshnet.MGCN, already on its device) to run on rank ``rank``'s share
    of every level, in place.  Level 0 is cut into balanced blocks along the Morton curve of its smooth
    positions; a coarse vertex goes to the owner of its first member (in processing order) and the
    coarse level is numbered by that member, so blocks stay contiguous and local at every level.
    Parameters stay replicated; BatchNorm statistics become mesh-wide.  ``phases`` (default): the runs of plain [ChebConv ->
    BatchNorm -> LeakyReLU] blocks of every stage -- 4 of the 5 blocks of a DownConv, 4 of 5 of an UpConv, the block of each
    head: 27 of the 33 -- go phase by phase below the C ABI (part_blocks) in train mode, one exchange for the run's input and
    one per block inside it with the BatchNorm statistics in its pad rows; the blocks with a pool between conv and BatchNorm
    (the rows change owner there: DistPool) stay module by module."""
    dev = model.smposs_list[0].device
    n_levels = len(model.smposs_list)
    order, rank_of = _reorder.morton_order(model.smposs_list[0])
    orders, ranks = [order], [rank_of]
    bounds = [block_bounds(model.smposs_list[0].shape[0], world)]
    pairs = []
    for l in range(n_levels - 1):
        ph = torch.as_tensor(np.asarray(model._pool_pairs[l]), dtype=torch.long, device=dev)
        fine_new = ranks[l][ph[:, 0]]
        n_c = model.smposs_list[l + 1].shape[0]
        first = torch.full((n_c,), int(bounds[l][-1]), dtype=torch.long, device=dev)
        first.scatter_reduce_(0, ph[:, 1], fine_new, "amin")
        if bool((first == bounds[l][-1]).any()):
            raise ValueError(f"level {l + 1} has a vertex without members in pool_hash")
        order_c = torch.argsort(first, stable=True)
        rank_c = torch.empty_like(order_c)
        rank_c[order_c] = torch.arange(n_c, device=dev)
        b = torch.searchsorted(first[order_c], torch.tensor(bounds[l], device=dev, dtype=torch.long)).tolist()
        b[0], b[-1] = 0, n_c
        orders.append(order_c), ranks.append(rank_c), bounds.append(b)
        pairs.append((fine_new, rank_c[ph[:, 1]]))
    graphs = [DistMeshGraph(_reorder.permute_edge_index(model.edge_inds[l].to(dev), ranks[l]),
                            model.smposs_list[l].shape[0], rank, world, group, bounds=bounds[l])
              for l in range(n_levels)]
    pools = [DistPool(f, c, bounds[l], bounds[l + 1], rank, world, group) for l, (f, c) in enumerate(pairs)]
    for g in graphs:
        g.phases = bool(phases)
    if phases:
        prepare_native_comm(graphs)
    own_ids = [orders[l][bounds[l][rank]:bounds[l][rank + 1]] for l in range(n_levels)]
    part = MGCNPartition(graphs, pools, bounds, own_ids, ranks,
                         [model.smposs_list[l].index_select(0, own_ids[l]) for l in range(n_levels)], rank, world, group)
    for stage, (lf, lc) in ((model.encoder1, (0, 1)), (model.encoder2, (1, 2)), (model.encoder3, (2, 3))):
        stage._graphs = (graphs[lf], graphs[lc])
        stage.model1.module_4._dist = pools[lf]
    for stage, (lc, lf) in ((model.decoder3, (3, 2)), (model.decoder2, (2, 1)), (model.decoder1[0], (1, 0))):
        stage._graphs = (graphs[lc], graphs[lf])
        stage.model1.module_1._dist = pools[lf]
    model._orders = None
    model._part = part
    convert_batchnorm(model, group)
    return part


def gather_level(x_own: torch.Tensor, part: MGCNPartition, level: int) -> torch.Tensor:
    """All ranks' rows of one level, back in the caller's numbering (inference / tests)."""
    b = part.bounds[level]
    n_max = max(b[r + 1] - b[r] for r in range(part.world))
    pad = x_own.new_zeros((n_max, x_own.shape[1]))
    pad[:x_own.shape[0]] = x_own.detach()
    ids = torch.full((n_max,), -1, dtype=torch.long, device=x_own.device)
    ids[:x_own.shape[0]] = part.own_ids[level]
    if _solo(part.world):
        allx, alli = pad, ids
    else:
        allx = pad.new_empty((part.world * n_max, x_own.shape[1]))
        alli = ids.new_empty((part.world * n_max,))
        _all_gather_rows(allx, pad, part.group)
        _all_gather_rows(alli.view(-1, 1), ids.view(-1, 1), part.group)
    keep = alli >= 0
    out = x_own.new_empty((b[-1], x_own.shape[1]))
    out[alli[keep]] = allx[keep]
    return out


class DistMGCNTrainer:
    """MGCNTrainer (semigcn_amd.train, the loop of mgcn.py:121-160) on a vertex partition: the masked
    sums of every level are all-reduced; the normal term reads halo positions of the finest level."""

    def __init__(self, model: nn.Module, part: MGCNPartition, batch, lr: float = 0.01, k1: float = 4.0,
                 accumulate: int = 5, weights=(0.35, 0.3, 0.2, 0.15)):
        self.model, self.part, self.batch, self.k1, self.accumulate, self.weights = model, part, batch, k1, accumulate, weights
        self.group = part.group
        dev = batch.target_pos.device
        self.params = [p for p in model.parameters()]
        self.opt = torch.optim.Adam(self.params, lr=lr)
        self.iteration = 0
        self.loss_sum = torch.zeros((), device=dev)
        self.targets = [t.index_select(0, ids) for t, ids in zip(model.poss_list, part.own_ids)]
        keeps = [m.to(dev) for m in model.v_masks_list]
        self.counts = [float(k.sum()) for k in keeps]
        self.keeps = [k.index_select(0, ids) for k, ids in zip(keeps, part.own_ids)]
        # faces of the finest level whose first corner this rank owns, corners as [owned | halo] positions
        g0 = part.graphs[0]
        faces = part.rank_of[0][batch.faces]
        mine = (faces[:, 0] >= g0.start) & (faces[:, 0] < g0.end)
        fo = faces[mine]
        self.faces_ext = g0.extended_index(fo)
        self.target_fn = batch.target_fn[mine]
        self.f_keep = batch.f_keep[mine]
        self.n_f_keep = float(batch.f_keep.sum())
        from .train import GradBuffer
        self.grads = GradBuffer(self.params)

    def loss(self, poss) -> torch.Tensor:
        from . import train
        total, first = 0.0, 0
        pos_ext = halo_extend(self.part.graphs[0], poss[0])
        if poss[0].is_cuda and poss[0].dtype == torch.float32:       # finest level: fused HIP kernels, as DistSGCNTrainer
            from .functional import mesh_loss_sums
            s0 = all_reduce_sum(mesh_loss_sums(pos_ext, self.faces_ext, self.targets[0], self.keeps[0], self.target_fn,
                                               self.f_keep), self.group)
            total = self.weights[0] * torch.sqrt(s0[0] / self.counts[0] + 1.0e-6) + self.k1 * (s0[1] / self.n_f_keep)
            first = 1
        else:
            fn = train.face_normals(pos_ext, self.faces_ext)
            total = self.k1 * all_reduce_sum(((fn - self.target_fn).abs() * self.f_keep).sum(), self.group) / self.n_f_keep
        for w, p, t, keep, n in list(zip(self.weights, poss, self.targets, self.keeps, self.counts))[first:]:
            d = (t - p) * keep
            total = total + w * torch.sqrt(all_reduce_sum((d * d).sum(), self.group) / n + 1.0e-6)
        return total

    def iteration_step(self, mask_index: Optional[int] = None) -> torch.Tensor:
        b = self.batch
        k = self.iteration % b.dummy_masks.shape[1] if mask_index is None else mask_index
        if not self.model.training:
            self.model.train()
        poss = self.model(b.data, b.v_keep * b.dummy_masks[:, k:k + 1])
        loss = self.loss(poss)
        from .functional import sink_param_grads
        with sink_param_grads():
            loss.backward()
        self.loss_sum += loss.detach()
        self.iteration += 1
        if self.iteration % self.accumulate == 0:
            self.grads.attach()
            all_reduce_gradients(self.params, self.group, flat=self.grads.whole())
            self.opt.step()
            self.grads.zero()
        return loss
