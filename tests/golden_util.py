"""Shared by oracle/make_golden.py (writer) and the tests (readers): deterministic
parameter fill that does not depend on torch's RNG stream (numpy RandomState is
frozen by policy), so golden files hold inputs/outputs but not megabytes of weights.
"""
from __future__ import annotations

import math
import os
import zlib

import numpy as np
import torch

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _seed_for(name: str, seed: int) -> int:
    return (zlib.crc32(name.encode()) + 1000003 * seed) % (2 ** 31 - 1)


def fill_state(module: torch.nn.Module, seed: int) -> None:
    """Overwrite every floating parameter/buffer of ``module`` in place, keyed by its
    state-dict name: 2-D weights glorot-uniform, BatchNorm scale/var in [0.5, 1.5], every
    other vector in [-0.1, 0.1].  Sparse and integer buffers are left alone."""
    with torch.no_grad():
        for name, t in module.state_dict().items():
            if t.is_sparse or not t.is_floating_point():
                continue
            rs = np.random.RandomState(_seed_for(name, seed))
            if t.dim() == 2:
                b = math.sqrt(6.0 / (t.shape[0] + t.shape[1]))
                v = rs.uniform(-b, b, size=tuple(t.shape))
            elif name.endswith("running_var") or (name.endswith("weight") and t.dim() == 1):
                v = rs.uniform(0.5, 1.5, size=tuple(t.shape))
            else:
                v = rs.uniform(-0.1, 0.1, size=tuple(t.shape))
            t.copy_(torch.from_numpy(v.astype(np.float32)).to(t.device))


def probe(name: str, shape, seed: int = 7) -> np.ndarray:
    """Fixed random direction used to compress a big gradient into one number."""
    rs = np.random.RandomState(_seed_for("probe/" + name, seed))
    return rs.standard_normal(size=tuple(shape)).astype(np.float32)


def grad_summary(named_grads, full_below: int = 3000):
    """{name: full grad} for small tensors, {name+'#dot', name+'#norm'} for big ones."""
    out = {}
    for name, g in named_grads:
        g = g.detach().cpu().float().numpy()
        if g.size <= full_below:
            out[name] = g
        else:
            out[name + "#dot"] = np.float64((g.astype(np.float64) * probe(name, g.shape)).sum())
            out[name + "#norm"] = np.float64(np.sqrt((g.astype(np.float64) ** 2).sum()))
    return out


def check_grad_summary(named_grads, golden, rtol: float, where: str = "", floor_frac: float = 1e-2):
    """Assert gradients match a summary written by ``grad_summary``.  Relative to the
    tensor's own scale in L2: |a-b|_2 <= rtol * max(|b|_2, floor * sqrt(n)) with floor =
    floor_frac * the largest gradient entry of the whole model.  (L2, not max: a single
    LeakyReLU sign flip caused by 1-ulp noise moves ONE channel's gradient by ~1/V.)  The floor matters for the
    ChebConv biases in front of a BatchNorm, whose true gradient is exactly zero (BN removes
    constant shifts) so that what autograd returns there is rounding noise."""
    named_grads = list(named_grads)
    abs_floor = floor_frac * max(float(np.abs(v).max()) for k, v in golden.items() if not k.endswith("#dot"))
    for name, g in named_grads:
        g = g.detach().cpu().float().numpy()
        if name in golden:
            ref = golden[name].astype(np.float64)
            scale = max(float(np.sqrt((ref ** 2).sum())), abs_floor * np.sqrt(ref.size))
            err = float(np.sqrt(((g - ref) ** 2).sum())) / scale
            assert err <= rtol, f"{where}{name}: rel L2 err {err:.3e} > {rtol:.1e}"
        else:
            dot = float((g.astype(np.float64) * probe(name, g.shape)).sum())
            norm = float(np.sqrt((g.astype(np.float64) ** 2).sum()))
            rn = float(golden[name + "#norm"])
            # |<g - ref, p>| ~ |g - ref| for a unit-variance probe
            assert abs(dot - float(golden[name + "#dot"])) <= rtol * max(rn, abs_floor) * 8, \
                f"{where}{name}: probe mismatch {dot} vs {float(golden[name + '#dot'])}"
            assert abs(norm - rn) <= rtol * max(rn, abs_floor) * 8, f"{where}{name}: norm mismatch"


def rel_l2(a, b) -> float:
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.sqrt(((a - b) ** 2).sum()) / max(np.sqrt((b ** 2).sum()), 1e-30))


def load(name: str):
    return np.load(os.path.join(GOLDEN_DIR, name), allow_pickle=False)


class ActivationMasks:
    """Records the sign pattern every LeakyReLU/ReLU sees during forward passes.

    A one-ulp difference in a BatchNorm output that happens to sit within ~1e-6 of zero flips
    that element's LeakyReLU slope (1 -> 0.01) in the backward pass; ONE such flip moves every
    upstream gradient by ~|g_i|/|g| ~ 0.5 % on the small fixture meshes.  Gradient parity is
    therefore asserted tightly only between runs with identical activation patterns, and with
    a per-flip allowance otherwise (see grad_tolerance)."""

    def __init__(self, model: torch.nn.Module, fused: bool = False):
        """``fused=True`` additionally listens to semigcn_amd's fused BatchNorm+activation op (which
        bypasses the activation modules); only ONE such recorder may be open at a time."""
        self.masks = []
        self._hooks = []
        self._observer = None
        seen = set()
        for mod in model.modules():
            if isinstance(mod, (torch.nn.LeakyReLU, torch.nn.ReLU)) and id(mod) not in seen:
                seen.add(id(mod))
                self._hooks.append(mod.register_forward_hook(self._record))
        if fused:
            from semigcn_amd import functional as F_sg
            self._observer = lambda y: self.masks.append((y.detach() > 0).cpu())   # sign(act(z)) == sign(z)
            F_sg.bn_act_observers.append(self._observer)

    def _record(self, mod, inputs, output):
        self.masks.append((inputs[0].detach() > 0).cpu())

    def close(self):
        for h in self._hooks:
            h.remove()
        if self._observer is not None:
            from semigcn_amd import functional as F_sg
            F_sg.bn_act_observers.remove(self._observer)
            self._observer = None

    def flips_against(self, other: "ActivationMasks", row_maps=None) -> int:
        """Number of elements whose sign differs; ``row_maps[i]`` (optional, LongTensor) re-orders
        the rows of this recorder's i-th mask into the other's vertex order."""
        assert len(self.masks) == len(other.masks), (len(self.masks), len(other.masks))
        n = 0
        for i, (a, b) in enumerate(zip(self.masks, other.masks)):
            if row_maps is not None and row_maps[i] is not None:
                a = a.index_select(0, row_maps[i])
            n += int((a != b).sum())
        return n


def grad_tolerance(flips: int, tight: float, per_flip: float = 2e-2) -> float:
    return tight if flips == 0 else tight + per_flip * flips


class PrescribedLeakyReLU(torch.nn.Module):
    """LeakyReLU that applies a PRESCRIBED sign pattern: call i uses ``masks[i]`` (bool, the input's shape) instead of
    ``z > 0``.  Gives two implementations of one network the same piecewise-linear branch, so that their gradients can
    be compared with a flat bound: where a BatchNorm output sits within rounding of zero, one implementation's
    ``z > 0`` and the other's differ, and that one slope (1 vs 0.01) moves every upstream gradient.  The forward value
    changes only where the prescribed sign differs from the module's own, by 0.99 |z| with |z| at rounding level; the
    module counts those elements (``flips``) and keeps their largest |z| / rms(z) (``max_flip_z``) so a test can
    assert that nothing but rounding-level kink crossings was overridden."""

    def __init__(self, masks=None, negative_slope: float = 0.01):
        super().__init__()
        self.negative_slope = negative_slope
        self.masks = masks
        self.reset()

    def reset(self, masks=None):
        if masks is not None:
            self.masks = masks
        self.calls, self.flips, self.elements, self.max_flip_z = 0, 0, 0, 0.0

    def forward(self, z):
        m = self.masks[self.calls % len(self.masks)]
        self.calls += 1
        with torch.no_grad():
            own = z > 0
            diff = own != m
            n = int(diff.sum())
            self.elements += z.numel()
            if n:
                self.flips += n
                rms = float(z.pow(2).mean().sqrt())
                self.max_flip_z = max(self.max_flip_z, float(z[diff].abs().max()) / max(rms, 1e-30))
        return torch.where(m, z, self.negative_slope * z)


class FusedActivationMasks:
    """The sign pattern of every fused BatchNorm+activation output of ``semigcn_amd`` during the forward passes inside the
    ``with`` block, rows brought into the caller's vertex order by ``rank`` (the model's processing-order map, or None);
    kept on the device until ``cpu()``."""

    def __init__(self, rank=None):
        self.rank = rank
        self.masks = []

    def __enter__(self):
        from semigcn_amd import functional as F_sg
        self._obs = lambda y: self.masks.append((y.detach() > 0) if self.rank is None
                                                else (y.detach() > 0).index_select(0, self.rank))
        F_sg.bn_act_observers.append(self._obs)
        return self

    def __exit__(self, *exc):
        from semigcn_amd import functional as F_sg
        F_sg.bn_act_observers.remove(self._obs)
        return False

    def cpu(self):
        return [m.cpu() for m in self.masks]


def synchronised_trajectory(trainer, net, ora, oracle_iteration, masks_dev, n_steps: int = 2, accumulate: int = 5,
                            lr: float = 0.01, loss_tol: float = 1e-5, noise: float = 5e-2, step_tol=(0.02, 0.1), off_frac: float = 1e-3):
    """The training loop of sgcn.py:118-147 on both sides, RE-SYNCHRONISED after every optimiser step.

    Why not simply compare two free-running loss curves: Adam's first step is ``lr * sign(g)`` for every parameter, so
    an entry whose accumulated gradient is rounding noise (every ChebConv bias in front of a BatchNorm -- true gradient
    exactly zero --, and a few dozen weight entries per layer) moves by +lr on one side and -lr on the other.  The
    reference's own arithmetic does this to itself: the oracle run with 1 and with 8 threads agrees to 1e-6 on the first
    five losses and to 1e-3 .. 9e-3 on the next five (measured on the 240- and the 5 000-vertex meshes).  So:
      * every iteration's loss is compared at ``loss_tol`` (both sides hold the same parameters when it runs);
      * at every optimiser step the two updated parameter sets are compared entry by entry wherever the oracle's
        accumulated gradient is above its noise floor (``noise`` x the tensor's rms gradient: the fp32 gradients of this
        network carry ~2e-3 of relative noise at 5 000 vertices, so single entries a few sigma out still differ): same
        step to ``step_tol[step]`` x lr for all but ``off_frac`` of those entries (the second step divides by a running
        variance that carries the first step's rounding) --
        elsewhere the step may go either way but never exceeds Adam's bound;
      * then the oracle's parameters are copied into the model under test and both continue.
    ``trainer``: train.SGCNTrainer on ``net``; ``oracle_iteration(k)`` runs forward + loss + backward of mask k on
    ``ora`` and returns the loss value; ``masks_dev``: the trainer's [V, n] dummy masks.  Returns the per-iteration
    relative loss errors and, per optimiser step, the largest above-noise parameter deviation (value, parameter name)."""
    opt = torch.optim.Adam(ora.parameters(), lr=lr)
    errs, k = [], 0
    step_dev = [(0.0, None)] * n_steps
    n_above, n_off = [0] * n_steps, [0] * n_steps
    for step in range(n_steps):
        trainer.grads.zero()
        opt.zero_grad()
        ora.train()
        for _ in range(accumulate):
            dm = trainer.mesh.v_keep * masks_dev[:, k:k + 1]
            lh = float(trainer._forward_backward(dm))
            lo = oracle_iteration(k)
            errs.append(abs(lh - lo) / abs(lo))
            assert errs[-1] < loss_tol, (step, k, lh, lo, errs)
            k += 1
        trainer.opt.step()
        opt.step()
        po = dict(ora.named_parameters())
        for name, p in net.named_parameters():
            g = po[name].grad
            if g is None:
                continue
            d = (p.detach().cpu() - po[name].detach()).abs()
            assert float(d.max()) <= 2.0 * lr * (step + 1) + 1e-6, name
            above = g.abs() > noise * g.pow(2).mean().sqrt()
            if name.endswith("module_0.bias"):
                continue            # in front of a BatchNorm: the true gradient is exactly zero, the step pure noise
            if bool(above.any()):
                tol = step_tol[min(step, len(step_tol) - 1)] * lr
                n_above[step] += int(above.sum())
                n_off[step] += int((d[above] > tol).sum())
                worst = float(d[above].max())
                if worst > step_dev[step][0]:
                    step_dev[step] = (worst, name)
        with torch.no_grad():
            for name, p in net.named_parameters():
                p.copy_(po[name].detach().to(p.device))
            sd = ora.state_dict()
            for name, b in net.named_buffers():
                if name in sd and "running" in name:
                    b.copy_(sd[name].to(b.device))
    for step in range(n_steps):          # an entry a few sigma out in the fp32 gradient noise may still step the other way
        assert n_off[step] <= off_frac * max(n_above[step], 1), (step, n_off[step], n_above[step], step_dev[step])
    return errs, [(v, name, n_off[i], n_above[i]) for i, (v, name) in enumerate(step_dev)]
