"""A training iteration recorded as a TAPE of hipGraph segments and eager actions, and replayed from it.

One hipGraph cannot hold a partitioned iteration: its 57 collectives (halo all-to-alls, BatchNorm all-gathers /
all-reduces, loss and bounding-box all-reduces) go through torch.distributed, the asynchronous ones return handles that are
waited on later, and under gloo they are staged through host memory.  But everything BETWEEN two collectives is a fixed
chain of kernels on fixed buffers.  So the iteration is recorded once, the normal code running as it is (modules, autograd,
the autograd Functions of functional.py / dist.py): kernels issued between two eager actions are captured into one
``torch.cuda.CUDAGraph`` each (all segments share one memory pool, so every buffer keeps its address), every eager action
(``segments.eager(fn)``: a collective, a wait) ends the running segment, is executed, is put on the tape as the callable
itself, and opens the next segment.  A replay walks the tape: ``graph.replay()`` / ``fn()`` in the recorded order -- ~60
graph launches and 57 collectives of host work instead of ~450 kernel launches with their Python and autograd glue, on a rank
that is host-bound.  The recording pass itself executes no captured kernel (its collectives move
uninitialised buffers); the tape is replayed once right after it to produce that iteration's result.

Rules for code that runs while recording: no host synchronisation and no value-dependent control flow outside ``eager``
actions; an eager action may only touch tensors that stay alive with the tape (it is the SAME callable, called again).
"""
from __future__ import annotations

from typing import Callable, List, Optional, Tuple

import torch

#: the recorder of the iteration being recorded right now (set by SegmentRecorder.record), else None
_active: Optional["SegmentRecorder"] = None


def recording() -> bool:
    return _active is not None


def eager(fn: Callable[[], object]):
    """Run ``fn()`` now; while an iteration is being recorded also cut the segment around it and put ``fn`` on the tape
    (its return value is that of this first call: later replays call ``fn`` for its effects only)."""
    rec = _active
    if rec is None or rec.in_eager:
        return fn()
    return rec.eager(fn)


class SegmentRecorder:
    def __init__(self, stream: torch.cuda.Stream):
        self.stream = stream
        self.pool = torch.cuda.graph_pool_handle()
        self.tape: List[Tuple[str, object]] = []
        self.graph: Optional[torch.cuda.CUDAGraph] = None
        self.in_eager = False

    def _begin(self) -> None:
        g = torch.cuda.CUDAGraph()
        g.capture_begin(pool=self.pool, capture_error_mode="relaxed")
        self.graph = g

    def _end(self) -> None:
        self.graph.capture_end()
        self.tape.append(("graph", self.graph))
        self.graph = None

    def eager(self, fn):
        self._end()
        self.in_eager = True
        try:
            out = fn()
        finally:
            self.in_eager = False
        self.tape.append(("call", fn))
        self._begin()
        return out

    def record(self, body: Callable[[], object]):
        """Run ``body()`` on ``self.stream`` with every kernel captured into segments; returns what it returns."""
        global _active
        if _active is not None:
            raise RuntimeError("an iteration is already being recorded")
        torch.cuda.synchronize()
        with torch.cuda.stream(self.stream):
            _active = self
            self._begin()
            try:
                out = body()
            except BaseException:
                try:
                    if self.graph is not None:
                        self.graph.capture_end()
                finally:
                    _active = None
                    self.graph = None
                raise
            self._end()
            _active = None
        return out

    def replay(self) -> None:
        with torch.cuda.stream(self.stream):
            for kind, obj in self.tape:
                if kind == "graph":
                    obj.replay()
                else:
                    obj()

    def counts(self) -> Tuple[int, int]:
        return (sum(1 for k, _ in self.tape if k == "graph"), sum(1 for k, _ in self.tape if k == "call"))
