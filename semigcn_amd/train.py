"""Training-step harness with the shape of the reference's inner loop
(/root/reference/sgcn.py:118-147): for each optimiser step, ``batch`` masked
forward + loss + backward passes are accumulated, then Adam steps once.

The losses run right after the network each iteration; they are plain
torch-ROCm ops here (SURVEY.md section 8(f)-1 ranks fusing them as "next"):
  face normals          util/models.py:121-126 (compute_fn)
  masked position RMSE  util/loss.py:14-34     (mask_pos_rec_loss, ltype='rmse')
  masked normal L1      util/loss.py:78-107    (mask_norm_rec_loss, ltype='l1mae')
Unlike the reference no ``.item()`` is taken inside the loop (sgcn.py:140-144 forces
3-4 device syncs per iteration); loss values are accumulated on the device.
"""
from __future__ import annotations

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

    def __post_init__(self):
        self.n_v_keep = int(self.v_keep.sum().item())
        self.n_f_keep = int(self.f_keep.sum().item())


class SGCNTrainer:
    """optimizer = Adam(lr), StepLR(50, 0.5) as sgcn.py:79-80; k1 = 4 (sgcn.py:47)."""

    def __init__(self, model: torch.nn.Module, batch: MeshBatch, lr: float = 0.01, k1: float = 4.0,
                 accumulate: int = 5):
        self.model, self.mesh, self.k1, self.accumulate = model, batch, k1, accumulate
        self.opt = torch.optim.Adam(model.parameters(), lr=lr)
        self.sched = torch.optim.lr_scheduler.StepLR(self.opt, step_size=50, gamma=0.5)
        self.iteration = 0
        self.loss_sum = torch.zeros((), device=batch.target_pos.device)
        self.opt.zero_grad(set_to_none=True)

    def loss(self, pos: torch.Tensor) -> torch.Tensor:
        b = self.mesh
        if pos.is_cuda and pos.dtype == torch.float32:     # fused HIP kernels (csrc/mesh_loss.hip)
            from .functional import mesh_loss_sums
            s = mesh_loss_sums(pos, b.faces, b.target_pos, b.v_keep, b.target_fn, b.f_keep)
            return torch.sqrt(s[0] / b.n_v_keep + 1.0e-6) + self.k1 * (s[1] / b.n_f_keep)
        lp = masked_position_rmse(pos, b.target_pos, b.v_keep, b.n_v_keep)
        ln = masked_normal_l1(face_normals(pos, b.faces), b.target_fn, b.f_keep, b.n_f_keep)
        return lp + self.k1 * ln

    def iteration_step(self, mask_index: Optional[int] = None) -> torch.Tensor:
        """One forward + loss + backward for one dummy mask (sgcn.py:123-144); every
        ``accumulate``-th call also applies Adam (sgcn.py:146)."""
        b = self.mesh
        k = self.iteration % b.dummy_masks.shape[1] if mask_index is None else mask_index
        dm = b.v_keep * b.dummy_masks[:, k:k + 1]
        if not self.model.training:      # the recursive mode switch costs ~0.5 ms of host time per call
            self.model.train()
        pos = self.model(b.data, dm)
        loss = self.loss(pos)
        loss.backward()
        self.loss_sum += loss.detach()
        self.iteration += 1
        if self.iteration % self.accumulate == 0:
            self.opt.step()
            self.opt.zero_grad(set_to_none=True)
        return loss


class MGCNTrainer:
    """The loop of /root/reference/mgcn.py:121-160: multi-resolution weighted position RMSE
    (weights 0.35/0.3/0.2/0.15, mgcn.py:82,138-143) + k1 x normal L1 on the finest level."""

    def __init__(self, model: torch.nn.Module, batch: MeshBatch, lr: float = 0.01, k1: float = 4.0,
                 accumulate: int = 5, weights=(0.35, 0.3, 0.2, 0.15)):
        self.model, self.mesh, self.k1, self.accumulate, self.weights = model, batch, k1, accumulate, weights
        self.opt = torch.optim.Adam(model.parameters(), lr=lr)
        self.iteration = 0
        self.loss_sum = torch.zeros((), device=batch.target_pos.device)
        self.keeps = [m.to(batch.target_pos.device) for m in model.v_masks_list]
        self.counts = [float(k.sum()) for k in self.keeps]
        self.opt.zero_grad(set_to_none=True)

    def iteration_step(self, mask_index: Optional[int] = None) -> torch.Tensor:
        b = self.mesh
        k = self.iteration % b.dummy_masks.shape[1] if mask_index is None else mask_index
        if not self.model.training:      # the recursive mode switch costs ~0.5 ms of host time per call
            self.model.train()
        # mgcn.py passes a Tensor mask, which MGCN.forward replaces by ones (util/meshnet.py:287-290)
        poss = self.model(b.data, b.v_keep * b.dummy_masks[:, k:k + 1])
        loss = sum(w * masked_position_rmse(p, t, keep, n)
                   for w, p, t, keep, n in zip(self.weights, poss, self.model.poss_list, self.keeps, self.counts))
        loss = loss + self.k1 * masked_normal_l1(face_normals(poss[0], b.faces), b.target_fn, b.f_keep, b.n_f_keep)
        loss.backward()
        self.loss_sum += loss.detach()
        self.iteration += 1
        if self.iteration % self.accumulate == 0:
            self.opt.step()
            self.opt.zero_grad(set_to_none=True)
        return loss

