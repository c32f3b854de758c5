"""CPU restatement of the reference's model composition on the hot path.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  These follow the reference's
own module code -- they exist because /root/reference does not travel to the
GPU box, where the ``-m gpu`` tests and ``smoke()`` still need a checker.  They
are pinned here (tests/test_oracle.py) against golden vectors produced by the
reference's own classes through oracle/ref_shim.py, and against the reference
classes directly when /root/reference is present.

  SGCNOracle      util/networks.py:9-103   (SingleScaleGCN)
  pool_mean       util/meshnet.py:9-17     (MeshPool.forward: P.x / rowsum(P))
  unpool_gather   util/meshnet.py:20-27    (MeshUnpool.forward: U.x)
  MGCNOracle      util/meshnet.py:31-160,212-248,278-318  (DownConv/UpConv/MGCN.forward)
  compute_fn / mask_pos_rec_loss / mask_norm_rec_loss
                  util/models.py:121-126, util/loss.py:14-34,78-107
"""
from __future__ import annotations

from typing import Optional, Sequence

import numpy as np
import torch
import torch.nn as nn

from .pyg_restatement import ChebConv, Sequential

SGCN_WIDTHS = (4, 16, 32, 64, 128, 256, 256, 512, 256, 256, 128, 64, 32, 16, 3)  # util/networks.py:15


def normalise_input(z1: torch.Tensor, dm: torch.Tensor) -> torch.Tensor:
    """util/networks.py:67-79: centre per axis, ONE scalar scale, mask, append mask."""
    lo = z1.min(dim=0, keepdim=True)[0]
    hi = z1.max(dim=0, keepdim=True)[0]
    scale = (hi - lo).max()
    z = (z1 - (lo + hi) * 0.5) / scale
    z = dm * z
    return torch.cat([z, dm], dim=1)


class SGCNOracle(nn.Module):
    """13 x [ChebConv(K=3), BatchNorm1d, LeakyReLU]; last block + Linear(16,3);
    six skip Linears always constructed (util/networks.py:40-61)."""

    def __init__(self, skip: bool = False, act: Optional[nn.Module] = None):
        """``act``: stands in for the ONE shared ``nn.LeakyReLU()`` instance of util/networks.py:17-18 (the tests inject
        a module that applies a prescribed sign pattern, tests/golden_util.py::PrescribedLeakyReLU)."""
        super().__init__()
        h = SGCN_WIDTHS
        self.skip = skip
        act = nn.LeakyReLU() if act is None else act
        blocks = []
        for i in range(13):
            mods = [(ChebConv(h[i], h[i + 1], K=3), "x, edge_index -> x"), nn.BatchNorm1d(h[i + 1]), act]
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
        x = normalise_input(z1, dm.to(z1.dtype))
        kept = []
        for i, blk in enumerate(self.blocks):
            if i >= 8 and self.skip:  # decoder: util/networks.py:95-99
                j = 13 - i
                x = self.skip_blocks[j](torch.cat([kept[j], x], dim=1))
            x = blk(x, edge_index)
            if i <= 5:
                kept.append(x)
        return x_pos + x


class SGCNComposition(nn.Module):
    """util/networks.py:9-103 restated over an INJECTED operator namespace: ``ops.ChebConv`` / ``ops.Sequential`` stand
    where the reference has ``from torch_geometric.nn import ChebConv, Sequential`` (util/networks.py:4).  With
    ``ops = oracle.pyg_restatement`` this is SGCNOracle; with ``ops = semigcn_amd.nn`` it is what the reference's own
    class becomes after ``compat.install()`` -- the zero-line integration -- including the reference's habit of moving
    ``z1``, ``x_pos``, ``edge_index`` and ``dm`` to the device on EVERY forward (:65,77).  Needed because the reference
    itself does not travel to the GPU box."""

    def __init__(self, device, ops, skip: bool = False):
        super().__init__()
        self.device, self.skip = device, skip
        h = SGCN_WIDTHS
        act = nn.LeakyReLU()
        blocks = []
        for i in range(13):
            mods = [(ops.ChebConv(h[i], h[i + 1], K=3), "x, edge_index -> x"), nn.BatchNorm1d(h[i + 1]), act]
            if i == 12:
                mods.append((nn.Linear(h[13], h[14]), "x -> x"))
            blocks.append(ops.Sequential("x, edge_index", mods))
        self.blocks = nn.ModuleList(blocks)
        self.skip_blocks = nn.ModuleList([nn.Linear(2 * h[j + 1], h[j + 1]) for j in range(6)])

    def forward(self, data, dm=None):
        z1, x_pos, edge_index = data.z1.to(self.device), data.x_pos.to(self.device), data.edge_index.to(self.device)
        if type(dm) == np.ndarray:
            dm = torch.from_numpy(dm)
        elif type(dm) != torch.Tensor:
            dm = torch.ones([z1.shape[0], 1])
        x = normalise_input(z1, dm.to(self.device))
        kept = []
        for i, blk in enumerate(self.blocks):
            if i >= 8 and self.skip:
                x = self.skip_blocks[13 - i](torch.cat([kept[13 - i], x], dim=1))
            x = blk(x, edge_index)
            if i <= 5:
                kept.append(x)
        return x_pos + x


def pool_mean(pool_hash: np.ndarray, x: torch.Tensor, n_coarse: Optional[int] = None) -> torch.Tensor:
    """out[s] = mean_{o: hash[o]=s} x[o]   (util/meshnet.py:14-17, builder :331-335)."""
    fine = torch.as_tensor(pool_hash[:, 0]).long()
    coarse = torch.as_tensor(pool_hash[:, 1]).long()
    n = int(coarse.max()) + 1 if n_coarse is None else n_coarse
    acc = x.new_zeros(n, x.shape[1]).index_add_(0, coarse, x[fine])
    cnt = x.new_zeros(n, 1).index_add_(0, coarse, x.new_ones(fine.shape[0], 1))
    return acc / cnt


def unpool_gather(pool_hash: np.ndarray, x: torch.Tensor, n_fine: Optional[int] = None) -> torch.Tensor:
    """out[o] = x[hash[o]]   (util/meshnet.py:25-27, builder :337-341)."""
    fine = torch.as_tensor(pool_hash[:, 0]).long()
    coarse = torch.as_tensor(pool_hash[:, 1]).long()
    n = int(fine.max()) + 1 if n_fine is None else n_fine
    out = x.new_zeros(n, x.shape[1])
    return out.index_add_(0, fine, x[coarse])


def _ident(x):
    return x


class _Pool(nn.Module):
    def __init__(self, h, store=_ident):
        super().__init__()
        self.h, self.store = h, store

    def forward(self, x):
        return self.store(pool_mean(self.h, x))


class _Unpool(nn.Module):
    def __init__(self, h, store=_ident):
        super().__init__()
        self.h, self.store = h, store

    def forward(self, x):
        return self.store(unpool_gather(self.h, x))


class Drop(nn.Module):
    """``nn.Dropout(p)`` (util/meshnet.py:62,128) whose Bernoulli draws can be PRESCRIBED: with ``masks`` set (a list of
    0/1 tensors of the input's shape, one per call), call i keeps exactly ``masks[i]`` and scales by 1 / (1 - p), as
    ``nn.Dropout`` does with its own draw -- so that two implementations of the network can be run on the same draws.
    ``store``: applied to the result (the bf16-storage oracle rounds there)."""

    def __init__(self, p: float, store=_ident):
        super().__init__()
        self.p, self.store = float(p), store
        self.drop = nn.Dropout(p)
        self.masks, self.calls = None, 0

    def forward(self, x):
        if self.masks is None or not self.training or self.p == 0.0:
            return self.store(self.drop(x))
        m = self.masks[self.calls % len(self.masks)].to(x.dtype)
        self.calls += 1
        return self.store(x * (m * (1.0 / (1.0 - self.p))))


def _cbl(cin, cout, K, conv=ChebConv, act=None):
    return [(conv(cin, cout, K=K), "x, edge_index -> x"),
            (nn.BatchNorm1d(cout), "x -> x"), ((nn.LeakyReLU() if act is None else act()), "x -> x")]


class DownOracle(nn.Module):
    """util/meshnet.py:31-95: the pool sits between the 2nd ChebConv and its BN."""

    def __init__(self, cin, cout, ei1, ei2, pool_hash, K=3, drop=0.0, conv=ChebConv, act=None, store=_ident):
        super().__init__()
        self.ei1, self.ei2 = ei1, ei2
        a = (lambda: nn.LeakyReLU()) if act is None else act
        self.model1 = Sequential("x, edge_index", _cbl(cin, cout, K, conv, a) + [
            (conv(cout, cout, K=K), "x, edge_index -> x"), (_Pool(pool_hash, store), "x -> x"),
            (nn.BatchNorm1d(cout), "x -> x"), (a(), "x -> x")])
        self.model2 = Sequential("x, edge_index", _cbl(cout, cout, K, conv, a) + _cbl(cout, cout, K, conv, a)
                                 + _cbl(cout, cout, K, conv, a) + [(Drop(drop, store), "x -> x")])

    def forward(self, x):
        return self.model2(self.model1(x, self.ei1), self.ei2)


class UpOracle(nn.Module):
    """util/meshnet.py:98-160: the unpool sits between the 1st ChebConv and its BN."""

    def __init__(self, cin, cout, ei1, ei2, pool_hash, K=3, drop=0.0, conv=ChebConv, act=None, store=_ident):
        super().__init__()
        self.ei1, self.ei2 = ei1, ei2
        a = (lambda: nn.LeakyReLU()) if act is None else act
        self.model1 = Sequential("x, edge_index", [
            (conv(cin, cout, K=K), "x, edge_index -> x"), (_Unpool(pool_hash, store), "x -> x"),
            (nn.BatchNorm1d(cout), "x -> x"), (a(), "x -> x")])
        self.model2 = Sequential("x, edge_index", _cbl(cout, cout, K, conv, a) + _cbl(cout, cout, K, conv, a)
                                 + _cbl(cout, cout, K, conv, a) + _cbl(cout, cout, K, conv, a)
                                 + [(Drop(drop, store), "x -> x")])

    def forward(self, x):
        return self.model2(self.model1(x, self.ei1), self.ei2)


class MGCNOracle(nn.Module):
    """MGCN over a PRECOMPUTED 3-level hierarchy (the reference builds it in
    ``__init__`` with its QEM simplifier, util/meshnet.py:182-201 -- out of scope).
    ``edge_inds``: 4 edge_index tensors (fine..coarse); ``pool_hashes``: 3 arrays of
    (fine_i, coarse_i) rows; ``smposs``: 4 smooth-position tensors.
    ``conv`` / ``act`` / ``store``: the ChebConv class, a factory for the activation modules and what is applied to every
    stored row that is not a conv or activation output (pool, unpool, dropout, network input, skip Linear) -- the
    bf16-storage variant (oracle/bf16.py::MGCNOracleBf16) passes its rounding versions; the defaults are the reference's."""

    def __init__(self, edge_inds: Sequence[torch.Tensor], pool_hashes: Sequence[np.ndarray],
                 smposs: Sequence[torch.Tensor], K: int = 3, skip: bool = False,
                 drop=(0.0, 0.2, 0.2), conv=ChebConv, act=None, store=_ident):
        super().__init__()
        e, p = list(edge_inds), list(pool_hashes)
        self.skip, self.edge_inds, self.smposs_list, self.store = skip, e, list(smposs), store
        kw = dict(conv=conv, act=act, store=store)
        self.encoder1 = DownOracle(4, 32, e[0], e[1], p[0], K, drop[0], **kw)
        self.encoder2 = DownOracle(32, 128, e[1], e[2], p[1], K, drop[1], **kw)
        self.encoder3 = DownOracle(128, 256, e[2], e[3], p[2], K, drop[2], **kw)
        self.decoder3 = UpOracle(256, 128, e[3], e[2], p[2], K, drop[2], **kw)
        self.decoder2 = UpOracle(128, 32, e[2], e[1], p[1], K, drop[1], **kw)
        self.decoder1 = nn.Sequential(UpOracle(32, 16, e[1], e[0], p[0], K, drop[0], **kw), nn.Linear(16, 3))
        a = (lambda: nn.LeakyReLU()) if act is None else act

        def head(c):
            return Sequential("x, edge_index", _cbl(c, 32, K, conv, a) + [(nn.Linear(32, 3), "x -> x")])

        self.mcnn3, self.mcnn2, self.mcnn1 = head(256), head(128), head(32)
        self.skip2, self.skip1 = nn.Linear(256, 128), nn.Linear(64, 32)

    def forward(self, z1, dm=None):
        # util/meshnet.py:287-290: anything that is not an ndarray becomes all-ones
        dm = torch.from_numpy(dm) if isinstance(dm, np.ndarray) else torch.ones(z1.shape[0], 1)
        st = self.store
        x = st(normalise_input(z1, dm.to(z1.dtype)))
        r1 = self.encoder1(x)
        r2 = self.encoder2(r1)
        r3 = self.encoder3(r2)
        o3 = self.mcnn3(r3, self.edge_inds[3])
        d2 = self.decoder3(r3)
        if self.skip:
            d2 = st(self.skip2(torch.cat([st(d2), st(r2)], dim=1)))
        o2 = self.mcnn2(d2, self.edge_inds[2])
        d1 = self.decoder2(d2)
        if self.skip:
            d1 = st(self.skip1(torch.cat([st(d1), st(r1)], dim=1)))
        o1 = self.mcnn1(d1, self.edge_inds[1])
        o0 = self.decoder1(d1)
        s = self.smposs_list
        return s[0] + o0, s[1] + o1, s[2] + o2, s[3] + o3

    def dropouts(self):
        """The six Drop modules in execution order (encoder1..3, decoder3..1): a test prescribes their draws."""
        return [self.encoder1.model2.module_9, self.encoder2.model2.module_9, self.encoder3.model2.module_9,
                self.decoder3.model2.module_12, self.decoder2.model2.module_12, self.decoder1[0].model2.module_12]


# ---- per-iteration geometry + losses (plain torch) ----
def compute_fn(vs: torch.Tensor, faces) -> torch.Tensor:
    """Unit face normals (util/models.py:121-126)."""
    f = torch.as_tensor(faces).long()
    n = torch.cross(vs[f[:, 1]] - vs[f[:, 0]], vs[f[:, 2]] - vs[f[:, 0]], dim=1)
    return n / torch.sqrt((n * n).sum(1, keepdim=True))


def mask_pos_rec_loss(pred: torch.Tensor, real: torch.Tensor, mask) -> torch.Tensor:
    """Masked RMSE (util/loss.py:14-34, ltype='rmse')."""
    m = torch.as_tensor(mask, dtype=torch.bool)
    d = (real[m] - pred[m]) ** 2
    return torch.sqrt(d.sum(1).sum() / d.shape[0] + 1.0e-6)


def mask_norm_rec_loss(pred: torch.Tensor, real: torch.Tensor, mask) -> torch.Tensor:
    """Masked L1 mean (util/loss.py:78-107, ltype='l1mae')."""
    m = torch.as_tensor(mask, dtype=torch.bool)
    d = (pred[m] - real[m]).abs().sum(1)
    return d.sum() / d.shape[0]


def sgcn_training_loop(net: nn.Module, z1, x_pos, edge_index, faces, target_pos, target_fn, v_mask, f_mask,
                       dummy_masks: torch.Tensor, mask_order: Sequence[int], batch: int = 5, lr: float = 0.01,
                       k1: float = 4.0, on_iteration=None):
    """The inner loop of sgcn.py:118-147 on the oracle: for every group of ``batch`` masks -- ``zero_grad``, then per
    mask ``dm = v_mask * dummy`` (:126-128), forward, ``loss_p + k1 * loss_n`` (:130-131,138), ``backward`` -- then ONE
    ``Adam.step`` (:146); ``Adam(lr=pos_lr)`` and ``k1 = 4`` are the script's defaults (:79, :47).  ``mask_order`` stands
    in for the script's ``torch.randperm`` (:119).  ``net(z1, x_pos, edge_index, dm)`` is an oracle model in train
    mode.  Returns the list of per-iteration loss values (Python floats, the ``loss.item()`` of :144).
    ``on_iteration(i)`` (optional) is called before iteration i's forward."""
    opt = torch.optim.Adam(net.parameters(), lr=lr)
    vm = torch.as_tensor(v_mask)
    rm = vm.reshape(-1, 1).float()
    losses = []
    order = list(mask_order)
    for start in range(0, len(order), batch):
        net.train()
        opt.zero_grad()
        for i, k in enumerate(order[start:start + batch]):
            if on_iteration is not None:
                on_iteration(start + i)
            dm = rm * dummy_masks[:, k].reshape(-1, 1)
            pos = net(z1, x_pos, edge_index, dm)
            loss = mask_pos_rec_loss(pos, target_pos, vm) + k1 * mask_norm_rec_loss(compute_fn(pos, faces), target_fn, f_mask)
            loss.backward()
            losses.append(float(loss.item()))
        opt.step()
    return losses
