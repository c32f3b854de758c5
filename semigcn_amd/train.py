"""Training-step harness with the shape of the reference's inner loop
(/root/reference/sgcn.py:118-147): for each optimiser step, ``batch`` masked
forward + loss + backward passes are accumulated, then Adam steps once.

The losses run right after the network each iteration:
  face normals          util/models.py:121-126 (compute_fn)
  masked position RMSE  util/loss.py:14-34     (mask_pos_rec_loss, ltype='rmse')
  masked normal L1      util/loss.py:78-107    (mask_norm_rec_loss, ltype='l1mae')
  bilateral normal term util/loss.py:196-253   (fn_bnf_detach_loss; the -CAD option, off by default)
On the device in fp32 the first three are one fused HIP forward and two backward kernels
(functional.mesh_loss_sums, csrc/mesh_loss.hip); the plain-torch versions below serve other
dtypes, MGCN's coarse levels and the tests.  Unlike the reference no ``.item()`` is taken
inside the loop (sgcn.py:140-144 forces 3-4 device syncs per iteration); loss values are
accumulated on the device.
"""
from __future__ import annotations

import os
from dataclasses import dataclass
from typing import Optional

import torch


def face_normals(vs: torch.Tensor, faces: torch.Tensor) -> torch.Tensor:
    a, b, c = vs[faces[:, 0]], vs[faces[:, 1]], vs[faces[:, 2]]
    n = torch.linalg.cross(b - a, c - a, dim=1)
    return n / torch.sqrt((n * n).sum(dim=1, keepdim=True))


def masked_position_rmse(pred: torch.Tensor, real: torch.Tensor, keep: torch.Tensor, count: int) -> torch.Tensor:
    """sqrt(mean over kept vertices of |real - pred|^2 + 1e-6); ``keep`` float [V,1], ``count`` = #kept."""
    d = (real - pred) * keep
    return torch.sqrt((d * d).sum() / count + 1.0e-6)


def masked_normal_l1(pred: torch.Tensor, real: torch.Tensor, keep: torch.Tensor, count: int) -> torch.Tensor:
    return ((pred - real).abs() * keep).sum() / count


def bilateral_normal_loss(pos: torch.Tensor, fn: torch.Tensor, faces: torch.Tensor, f2f: torch.Tensor,
                          loop: int = 5, sigma_s: float = 0.3):
    """``Loss.fn_bnf_detach_loss`` (util/loss.py:196-253, ltype='l1mae'; the ``-CAD`` term of
    sgcn.py:133-135 / mgcn.py:146-148): ``loop`` rounds of bilateral filtering of the face normals
    over the face 1-ring ``f2f`` ([F,3], -1 = no neighbour; semigcn_amd.meshprep.MeshTopology.f2f),
    weights exp(-|dc|^2 / 2 sigma_c^2) * exp(-|dn|^2 / 2 sigma_s^2) * area, computed on DETACHED
    geometry; returns (mean_f |filtered_f - fn_f|_1, filtered normals).  Gradients reach ``fn`` only.
    Plain torch ops (a dozen small face-sized kernels per round): an optional term, not on the
    default path.  A -1 in ``f2f`` indexes the last face like the reference's numpy-style wrap, and
    is then weighted by zero area."""
    p = pos.detach()
    a, b, c = p[faces[:, 0]], p[faces[:, 1]], p[faces[:, 2]]
    fc = (a + b + c) / 3.0
    fa = 0.5 * torch.sqrt((torch.linalg.cross(b - a, c - a, dim=1) ** 2).sum(1) + 1.0e-12)
    has = (f2f != -1).to(fa.dtype)
    neig_fa = fa[f2f] * has
    fc_dist = ((fc[f2f] - fc.unsqueeze(1)) ** 2).sum(2)
    sigma_c = torch.sqrt(fc_dist + 1.0e-12).sum() / fc_dist.numel()
    wc = torch.exp(-fc_dist / (2 * sigma_c ** 2))
    new_fn = fn
    for _ in range(loop):
        neig_fn = new_fn[f2f]
        ws = torch.exp(-((neig_fn - new_fn.unsqueeze(1)) ** 2).sum(2) / (2 * sigma_s ** 2))
        new_fn = ((wc * ws * neig_fa).unsqueeze(2) * neig_fn).sum(1)
        new_fn = (new_fn / (torch.sqrt((new_fn ** 2).sum(1, keepdim=True) + 1.0e-12) + 1.0e-12)).detach()
    return (new_fn - fn).abs().sum(1).sum() / fn.shape[0], new_fn


@dataclass
class MeshBatch:
    """Per-mesh constants of the training loop, resident on the device."""
    data: object                 # .z1 .x_pos .edge_index (device tensors)
    faces: torch.Tensor          # [F,3] int64
    target_pos: torch.Tensor     # [V,3]  ini_mesh.vs
    target_fn: torch.Tensor      # [F,3]  ini_mesh face normals
    v_keep: torch.Tensor         # [V,1] float, 1 = original vertex (v_mask)
    f_keep: torch.Tensor         # [F,1] float
    dummy_masks: torch.Tensor    # [V, n_masks] float (vmask_dummy)
    n_v_keep: int = 0
    n_f_keep: int = 0
    f2f: Optional[torch.Tensor] = None   # [F,3] int64 face 1-ring, -1 padded; only for the -CAD term (k2 > 0)

    def __post_init__(self):
        self.n_v_keep = int(self.v_keep.sum().item())
        self.n_f_keep = int(self.f_keep.sum().item())


#: ROCm 7.2's graph "packet capture" fast path replays stale packets after an eager kernel has run on an idle
#: GPU (wrong gradients; bisected in tools/graph_replay_check.py).  With this runtime flag off, replays are exact.
GRAPH_ENV = ("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")


#: value of GRAPH_ENV[0] at the moment the HIP runtime was initialised in this process (the runtime reads the flag
#: once, then): recorded by a callback queued on torch's lazy CUDA initialisation when this module is imported
#: before the first GPU call, else the value found at import time (the best that can still be known).
_flag_at_gpu_init = {"value": None, "known": False}


def _record_flag_at_init():
    _flag_at_gpu_init["value"] = os.environ.get(GRAPH_ENV[0])
    _flag_at_gpu_init["known"] = True


if torch.cuda.is_initialized():
    _record_flag_at_init()
else:
    try:
        torch.cuda._lazy_call(_record_flag_at_init)
    except Exception:      # no lazy-init hook on this build: fall back to the value seen at first use
        pass


def graphs_usable() -> bool:
    """True only if DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 was in the environment when the HIP runtime initialised (setting
    it afterwards has no effect on the runtime and would let a corrupting replay through).  BEST EFFORT: the value is
    sampled at torch's lazy CUDA initialisation (or at import, if torch was initialised already); the HIP runtime itself
    can have been brought up earlier (a profiler's preloaded library, ``torch.cuda.device_count()`` on some builds) and
    would then not have seen a flag set in between.  That is why the replaying trainers do not rely on this guard alone:
    the first replay is compared with an eager pass over the same inputs before it is trusted
    (``replay_matches_eager``)."""
    if not _flag_at_gpu_init["known"]:
        if torch.cuda.is_initialized():
            _record_flag_at_init()
        else:
            return os.environ.get(GRAPH_ENV[0]) == GRAPH_ENV[1]     # the runtime has yet to read it
    return _flag_at_gpu_init["value"] == GRAPH_ENV[1] and os.environ.get(GRAPH_ENV[0]) == GRAPH_ENV[1]


class _PrivateData:
    """The caller's ``data`` with ``z1`` replaced by a leaf of the trainer's own (same values, same
    requires_grad): autograd keeps a leaf's AccumulateGrad node on the stream of its first use, and one
    that has lived on another stream (an eager run on the default stream, say) would fork the capture."""

    def __init__(self, data):
        for name in ("x_pos", "edge_index", "graph"):
            if hasattr(data, name):
                setattr(self, name, getattr(data, name))
        self.z1 = data.z1.detach().clone().requires_grad_(data.z1.requires_grad)


class _GraphedIteration:
    """One training iteration (forward + loss + backward) captured into a hipGraph (``torch.cuda.CUDAGraph``) and
    replayed: one host call instead of ~650 (SGCN) / ~1300 (MGCN) kernel launches.  Pays off where the iteration is
    launch-bound -- meshes of <= ~200 K vertices and MGCN's small coarse levels; at V = 1 M the GPU is busy anyway.

    Contract of ``body(mask)``: reads only tensors that keep their storage between calls (model parameters and
    buffers, the trainer's constants, the static ``mask`` handed in), returns the detached scalar loss.  Parameter
    gradients accumulate into ``.grad`` tensors that exist before the capture, so the optimiser must zero them in
    place (``set_to_none=False``).  Needs ``DEBUG_CLR_GRAPH_PACKET_CAPTURE=0`` in the environment BEFORE the first
    HIP call of the process (see GRAPH_ENV); refuses to run otherwise."""

    WARMUP = 3          # eager iterations first: allocator, graph/CSR caches, hipBLASLt heuristics

    def __init__(self, params, mask_like: torch.Tensor, body):
        if not graphs_usable():
            raise RuntimeError(f"hipGraph replay needs {GRAPH_ENV[0]}={GRAPH_ENV[1]} in the environment BEFORE the process first touches "
                               "the GPU: with ROCm 7.2's default graph packet capture a replay that follows an eager "
                               "kernel on an idle GPU returns wrong gradients (tools/graph_replay_check.py)")
        self.params, self.body = [p for p in params], body
        self.mask = torch.zeros_like(mask_like)
        self.graph = None
        self.loss = None
        self.calls = 0
        # ONE side stream for the warm-ups and the capture: autograd's AccumulateGrad nodes run on the stream they
        # were first used on, and a node left over from another stream would fork the capture
        self.stream = torch.cuda.Stream(mask_like.device)

    def __call__(self, mask: torch.Tensor) -> torch.Tensor:
        self.mask.copy_(mask)
        if self.graph is not None:
            self.graph.replay()
            return self.loss.clone()       # a fresh tensor per call, as in eager mode (the static one is overwritten by the next replay)
        self.calls += 1
        if self.calls <= self.WARMUP:
            self.stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self.stream):
                loss = self.body(self.mask)
            torch.cuda.current_stream().wait_stream(self.stream)
            return loss
        for p in self.params:                      # AccumulateGrad must add in place inside the graph
            if p.requires_grad and p.grad is None:
                p.grad = torch.zeros_like(p)
        import gc
        gc.collect()                               # a dead CUDAGraph collected DURING a capture aborts the process
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=self.stream):
            self.loss = self.body(self.mask)
        self.graph = graph                         # the capture itself executed nothing: replay it for this call
        graph.replay()
        return self.loss.clone()


def replay_matches_eager(trainer, mask_index: int = 0, rtol: float = 1e-6, grad_rtol: float = 1e-4) -> bool:
    """Trust a replaying trainer only after ONE replayed iteration has reproduced an eager one: both run on the same mask
    with the same parameters (no optimiser step lies between them; BatchNorm normalises with batch statistics in training
    mode, so the running averages moving in between do not matter), and their loss values -- on a partition the mesh-wide,
    all-reduced value, hence the same number on every rank -- must agree to ``rtol`` (the replay tests find them
    bit-identical), AND the gradient increment each of the two iterations adds to every parameter's ``.grad`` and the
    BatchNorm running statistics they leave must agree to ``grad_rtol``: the corrupted replays this guards against
    (tools/graph_replay_check.py) keep the forward pass intact and return wrong gradients.  Costs two iterations.  Not for
    models with active Dropout (MGCN: two passes draw different masks) -- bench.py does not replay those by default.
    Returns False (and leaves the trainer on its EAGER path) otherwise.  On a partition the gradient verdict is local to a
    rank; the callers all-reduce it where ranks must agree (bench.py)."""
    rep = getattr(trainer, "_graphed", None)
    if rep is None:
        return True
    attr = "_graphed"
    while getattr(rep, "graph", None) is None:
        trainer.iteration_step(mask_index)              # still warming up / recording
    if trainer.accumulate < 3:
        # the pair of iterations below must not straddle an optimiser step (it would change the parameters between them),
        # which takes two steps in a row without one: impossible here -- stay eager rather than trust an unchecked replay
        import sys
        print("semigcn_amd: accumulate < 3 leaves no room for the replay == eager check; hipGraph replay switched off for this "
              "trainer", file=sys.stderr, flush=True)
        setattr(trainer, attr, None)
        return False
    for _ in range(trainer.accumulate):                 # keep the pair clear of the optimiser step
        if (trainer.iteration + 1) % trainer.accumulate and (trainer.iteration + 2) % trainer.accumulate:
            break
        trainer.iteration_step(mask_index)
    # what a bad replay corrupts is the BACKWARD pass (tools/graph_replay_check.py: gradients off by rel-L2 ~20 with an intact
    # forward): compare the gradient INCREMENT of the two iterations and the BatchNorm running statistics they leave,
    # not the loss alone
    params = [p for p in getattr(trainer, "params", None) or trainer.model.parameters() if p.requires_grad and p.grad is not None]
    bns = [m for m in trainer.model.modules() if isinstance(m, torch.nn.BatchNorm1d) and m.running_mean is not None]

    def snap():
        return ([p.grad.detach().clone() for p in params],
                [(m.running_mean.detach().clone(), m.running_var.detach().clone()) for m in bns])

    def restore(state):
        with torch.no_grad():
            for m, (rm, rv) in zip(bns, state[1]):
                m.running_mean.copy_(rm)
                m.running_var.copy_(rv)
    g0 = snap()
    setattr(trainer, attr, None)
    try:
        eager = float(trainer.iteration_step(mask_index))
    finally:
        setattr(trainer, attr, rep)
    g1 = snap()
    restore(g0)                                         # both iterations start from the same running statistics
    replayed = float(trainer.iteration_step(mask_index))
    g2 = snap()
    ok = abs(replayed - eager) <= rtol * max(abs(eager), 1e-30)
    worst = 0.0
    if ok:
        for a, b, c in zip(g0[0], g1[0], g2[0]):
            de, dr = (b - a).double(), (c - b).double()
            scale = float(de.norm())
            if scale > 0:
                worst = max(worst, float((dr - de).norm()) / scale)
        for (_, _), (rm1, rv1), (rm2, rv2) in zip(g0[1], g1[1], g2[1]):
            for x1, x2 in ((rm1, rm2), (rv1, rv2)):
                scale = float(x1.double().norm())
                if scale > 0:
                    worst = max(worst, float((x2.double() - x1.double()).norm()) / scale)
        ok = worst <= grad_rtol
    try:            # on a partition every rank must reach the SAME verdict (a rank that went eager alone would stop pairing
        import torch.distributed as tdist      # its collectives with the replaying ranks')
        if tdist.is_available() and tdist.is_initialized() and tdist.get_world_size(getattr(trainer, "group", None)) > 1:
            from . import dist as _d
            flag = torch.tensor([1.0 if ok else 0.0], device=params[0].device if params else "cpu")
            _d._all_reduce(flag, tdist.ReduceOp.MIN, getattr(trainer, "group", None))
            ok = bool(float(flag) > 0.5)
    except ImportError:
        pass
    if not ok:
        import sys
        print(f"semigcn_amd: replayed iteration differs from the eager one (loss {replayed!r} vs {eager!r}, worst relative "
              f"difference of a gradient increment / running statistic {worst:.3g}); hipGraph replay switched off for this "
              "trainer", file=sys.stderr, flush=True)
        setattr(trainer, attr, None)
    return ok


class GradBuffer:
    """Every fp32 parameter gradient of a model as a view of ONE flat buffer: ``p.grad`` exists from the start (the layers
    add into it in place under ``functional.sink_param_grads``; autograd's own AccumulateGrad adds in place too), zeroing
    all of them is one fill instead of one per parameter, and a partitioned trainer all-reduces the buffer itself."""

    def __init__(self, params):
        self.params = [p for p in params if p.requires_grad]
        main = [p for p in self.params if p.dtype == torch.float32]
        self.flat = None
        self.views = {}
        if main and all(p.device == main[0].device for p in main):
            self.flat = torch.zeros(sum(p.numel() for p in main), dtype=torch.float32, device=main[0].device)
            off = 0
            for p in main:
                self.views[id(p)] = self.flat[off:off + p.numel()].view_as(p)
                off += p.numel()
        # a NEW buffer starts from zero: whatever a backward pass made before the trainer existed left in ``.grad`` (a
        # parity check, a smoke run) must not reach the first optimiser step -- the reference zeroes the gradients at the
        # head of every accumulation cycle (sgcn.py:121)
        self.attach(keep=False)

    def attach(self, keep: bool = True) -> None:
        """(Re)install the views as ``.grad`` -- also after something set a gradient to None or replaced it in the middle
        of an accumulation cycle (``keep``: what that gradient holds is carried over into the view)."""
        for p in self.params:
            v = self.views.get(id(p))
            if v is None:
                if p.grad is None:
                    p.grad = torch.zeros_like(p)
                elif not keep:
                    p.grad.zero_()
            elif p.grad is not v:
                if keep and p.grad is not None and p.grad.data_ptr() != v.data_ptr():
                    v.copy_(p.grad)
                p.grad = v

    def whole(self):
        """The flat buffer if EVERY gradient is a view of it (then one collective on it covers them all), else None."""
        return self.flat if len(self.views) == len(self.params) else None

    def zero(self) -> None:
        self.attach()
        if self.flat is not None:
            self.flat.zero_()
        for p in self.params:
            if id(p) not in self.views:
                p.grad.zero_()


class _Epochs:
    """Epoch-level semantics of the reference's loop (sgcn.py:110-148, mgcn.py:112-160) on top of ``iteration_step``: every
    epoch visits all ``n_data`` dummy masks in a fresh ``torch.randperm`` order, ``accumulate`` of them per optimiser step
    (``batch_index = torch.randperm(n_data).reshape(-1, args.batch)``), and ends with ``scheduler.step()`` (StepLR(50,
    0.5)).  ``iteration_step()`` without an index keeps cycling through the masks in storage order (benchmarking); a
    training run calls ``train_epoch()``."""
    epoch = 0

    def train_epoch(self, order=None):
        """One epoch; returns the mean loss over its iterations as a device scalar (no host synchronisation).  ``order``:
        the mask order to use instead of a fresh ``torch.randperm(n_data)`` (tests)."""
        n = self.mesh.dummy_masks.shape[1]
        if n % self.accumulate:
            raise ValueError(f"{n} dummy masks cannot be cut into batches of {self.accumulate} (sgcn.py:112 reshapes the "
                             "permutation to [-1, batch])")
        if self.iteration % self.accumulate:
            raise RuntimeError("train_epoch() in the middle of an accumulation cycle")
        order = torch.randperm(n).tolist() if order is None else [int(k) for k in order]
        total = torch.zeros((), device=self.loss_sum.device)
        for k in order:
            total = total + self.iteration_step(k)
        self.epoch_end()
        return total / n

    def epoch_end(self) -> None:
        """``scheduler.step()`` (sgcn.py:148): the learning rate halves every 50 epochs."""
        self.sched.step()
        self.epoch += 1


class SGCNTrainer(_Epochs):
    """optimizer = Adam(lr), StepLR(50, 0.5) as sgcn.py:79-80; k1 = 4 (sgcn.py:47)."""

    def __init__(self, model: torch.nn.Module, batch: MeshBatch, lr: float = 0.01, k1: float = 4.0,
                 accumulate: int = 5, k2: float = 0.0, capture: bool = False):
        """``k2 > 0`` adds the bilateral normal term of the reference's ``-CAD`` runs (sgcn.py:133-135, its
        default weight is 4.0); it needs ``batch.f2f``.  ``capture=True`` replays the iteration from a hipGraph
        after three eager ones (see _GraphedIteration)."""
        self.model, self.mesh, self.k1, self.accumulate, self.k2 = model, batch, k1, accumulate, k2
        if k2 > 0 and batch.f2f is None:
            raise ValueError("k2 > 0 (the -CAD bilateral normal term) needs MeshBatch.f2f")
        self.opt = torch.optim.Adam(model.parameters(), lr=lr)
        self.sched = torch.optim.lr_scheduler.StepLR(self.opt, step_size=50, gamma=0.5)
        self.iteration = 0
        self.loss_sum = torch.zeros((), device=batch.target_pos.device)
        self.grads = GradBuffer(model.parameters())
        self._graphed = _GraphedIteration(model.parameters(), batch.v_keep, self._forward_backward) if capture else None
        self._data = _PrivateData(batch.data) if capture else batch.data

    def _forward_backward(self, dm: torch.Tensor) -> torch.Tensor:
        from .functional import sink_param_grads
        if not self.model.training:      # the recursive mode switch costs ~0.5 ms of host time per call
            self.model.train()
        loss = self.loss(self.model(self._data, dm))
        with sink_param_grads():         # the layers add their parameter gradients into the GradBuffer views themselves
            loss.backward()
        return loss.detach()

    def loss(self, pos: torch.Tensor) -> torch.Tensor:
        b = self.mesh
        if pos.is_cuda and pos.dtype == torch.float32:     # fused HIP kernels (csrc/mesh_loss.hip): one scalar
            from .functional import mesh_loss
            loss = mesh_loss(pos, b.faces, b.target_pos, b.v_keep, b.target_fn, b.f_keep, b.n_v_keep, b.n_f_keep, 1.0, self.k1)
            if self.k2 > 0:
                loss = loss + self.k2 * bilateral_normal_loss(pos, face_normals(pos, b.faces), b.faces, b.f2f)[0]
            return loss
        fn = face_normals(pos, b.faces)
        loss = masked_position_rmse(pos, b.target_pos, b.v_keep, b.n_v_keep) \
            + self.k1 * masked_normal_l1(fn, b.target_fn, b.f_keep, b.n_f_keep)
        if self.k2 > 0:
            loss = loss + self.k2 * bilateral_normal_loss(pos, fn, b.faces, b.f2f)[0]
        return loss

    def iteration_step(self, mask_index: Optional[int] = None) -> torch.Tensor:
        """One forward + loss + backward for one dummy mask (sgcn.py:123-144); every
        ``accumulate``-th call also applies Adam (sgcn.py:146)."""
        b = self.mesh
        k = self.iteration % b.dummy_masks.shape[1] if mask_index is None else mask_index
        dm = b.v_keep * b.dummy_masks[:, k:k + 1]
        loss = self._graphed(dm) if self._graphed is not None else self._forward_backward(dm)
        self.loss_sum += loss
        self.iteration += 1
        if self.iteration % self.accumulate == 0:
            self.opt.step()
            self.grads.zero()
        return loss


class MGCNTrainer(_Epochs):
    """The loop of /root/reference/mgcn.py:121-160: multi-resolution weighted position RMSE
    (weights 0.35/0.3/0.2/0.15, mgcn.py:82,138-143) + k1 x normal L1 on the finest level; Adam + StepLR(50, 0.5)
    (mgcn.py:72-73)."""

    def __init__(self, model: torch.nn.Module, batch: MeshBatch, lr: float = 0.01, k1: float = 4.0,
                 accumulate: int = 5, weights=(0.35, 0.3, 0.2, 0.15), k2: float = 0.0, capture: bool = False):
        self.model, self.mesh, self.k1, self.accumulate, self.weights, self.k2 = model, batch, k1, accumulate, weights, k2
        if k2 > 0 and batch.f2f is None:
            raise ValueError("k2 > 0 (the -CAD bilateral normal term, mgcn.py:146-148) needs MeshBatch.f2f")
        self.opt = torch.optim.Adam(model.parameters(), lr=lr)
        self.sched = torch.optim.lr_scheduler.StepLR(self.opt, step_size=50, gamma=0.5)
        self.iteration = 0
        self.loss_sum = torch.zeros((), device=batch.target_pos.device)
        self.keeps = [m.to(batch.target_pos.device) for m in model.v_masks_list]
        self.counts = [float(k.sum()) for k in self.keeps]
        self.grads = GradBuffer(model.parameters())
        self._graphed = _GraphedIteration(model.parameters(), batch.v_keep, self._forward_backward) if capture else None
        self._data = _PrivateData(batch.data) if capture else batch.data

    def iteration_step(self, mask_index: Optional[int] = None) -> torch.Tensor:
        b = self.mesh
        k = self.iteration % b.dummy_masks.shape[1] if mask_index is None else mask_index
        dm = b.v_keep * b.dummy_masks[:, k:k + 1]
        loss = self._graphed(dm) if self._graphed is not None else self._forward_backward(dm)
        self.loss_sum += loss
        self.iteration += 1
        if self.iteration % self.accumulate == 0:
            self.opt.step()
            self.grads.zero()
        return loss

    def _forward_backward(self, dm: torch.Tensor) -> torch.Tensor:
        b = self.mesh
        if not self.model.training:      # the recursive mode switch costs ~0.5 ms of host time per call
            self.model.train()
        # mgcn.py passes a Tensor mask, which MGCN.forward replaces by ones (util/meshnet.py:287-290)
        poss = self.model(self._data, dm)
        if poss[0].is_cuda and poss[0].dtype == torch.float32:
            # finest level: position and normal terms from the fused HIP kernels (csrc/mesh_loss.hip), as in SGCNTrainer
            from .functional import mesh_loss
            loss = mesh_loss(poss[0], b.faces, self.model.poss_list[0], self.keeps[0], b.target_fn, b.f_keep, self.counts[0],
                             b.n_f_keep, self.weights[0], self.k1)
            fn = None
            for w, p, t, keep, n in list(zip(self.weights, poss, self.model.poss_list, self.keeps, self.counts))[1:]:
                loss = loss + mesh_loss(p, None, t, keep, None, None, n, 0.0, w)     # the coarser resolutions: position term only
        else:
            fn = face_normals(poss[0], b.faces)
            loss = self.k1 * masked_normal_l1(fn, b.target_fn, b.f_keep, b.n_f_keep)
            for w, p, t, keep, n in zip(self.weights, poss, self.model.poss_list, self.keeps, self.counts):
                loss = loss + w * masked_position_rmse(p, t, keep, n)
        if self.k2 > 0:
            fn = face_normals(poss[0], b.faces) if fn is None else fn
            loss = loss + self.k2 * bilateral_normal_loss(poss[0], fn, b.faces, b.f2f)[0]
        from .functional import sink_param_grads
        with sink_param_grads():
            loss.backward()
        return loss.detach()

