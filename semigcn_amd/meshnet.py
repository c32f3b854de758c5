"""Model tier, multi-resolution: ``MeshPool`` / ``MeshUnpool`` / ``DownConv`` / ``UpConv`` /
``MGCN`` with the reference's constructors, attributes, forward contract and state-dict keys
(/root/reference/util/meshnet.py), on the HIP kernels.

What is different on purpose:
  * ``MeshPool.forward`` is a segment mean with the cluster sizes counted ONCE at build time;
    the reference densifies the [V_coarse x V_fine] matrix on every call
    (util/meshnet.py:15 -- 6 GB at 50 K vertices);
  * ``MGCN.__init__`` has no file-system side effects unless ``save_pooled=True``
    (the reference always writes pooled/*.obj / *.ply, util/meshnet.py:181,201,270);
  * ``MGCN.from_hierarchy`` builds the network from precomputed ``pool_hash`` pairs and
    coarse ``edge_index`` tensors; the reference can only get them from its own Python QEM
    simplifier (``Mesh.simplification``, util/mesh.py:394-482 -- out of scope, minutes at
    50 K vertices).  ``MGCN(device, smo_mesh, ini_mesh, v_mask)`` still calls
    ``mesh.simplification(target_v)`` on whatever mesh objects it is given, exactly like the
    reference, so the reference's ``Mesh`` class keeps working with it.
"""
from __future__ import annotations

import copy
import os
from typing import Optional, Sequence

import numpy as np
import torch
import torch.nn as nn

from . import capi
from . import functional as F_sg
from . import reorder as _reorder
from .graph import MeshGraph
from .networks import _Fp32Linear, prepare_input
from .nn import ChebConv, Sequential, run_sequentials

POOL_LEVELS = 3          # util/meshnet.py:169
POOL_RATIO = 0.6         # util/meshnet.py:171
LEVEL_WEIGHTS = (0.35, 0.3, 0.2, 0.15)  # mgcn.py:82 (loss weights per resolution)


def _pairs_from_sparse(mat: torch.Tensor):
    """(row_index, col_index, shape) of a 0/1 sparse COO matrix as the reference builds them
    (util/meshnet.py:331-341); duplicates are kept, values must be ones."""
    if not mat.is_sparse:
        raise TypeError("expected a sparse COO tensor (pool_hash_to_mask / unpool_hash_to_mask output)")
    idx = mat._indices()
    val = mat._values()
    if val.numel() and not bool(torch.all(val == 1)):
        raise NotImplementedError("pool/unpool matrices with values other than 1 are not implemented")
    return idx[0], idx[1], tuple(mat.shape)


def pool_hash_to_mask(pool_hash) -> torch.Tensor:
    """[V_coarse, V_fine] 0/1 sparse matrix from (fine_i, coarse_i) rows (util/meshnet.py:331-335)."""
    ph = torch.as_tensor(np.asarray(pool_hash), dtype=torch.long)
    ind = torch.stack([ph[:, 1], ph[:, 0]], dim=0)
    return torch.sparse_coo_tensor(ind, torch.ones(ind.shape[1]), size=(int(ph[:, 1].max()) + 1, int(ph[:, 0].max()) + 1))


def unpool_hash_to_mask(pool_hash) -> torch.Tensor:
    """[V_fine, V_coarse] 0/1 sparse matrix (util/meshnet.py:337-341)."""
    ph = torch.as_tensor(np.asarray(pool_hash), dtype=torch.long)
    return torch.sparse_coo_tensor(ph.T.contiguous(), torch.ones(ph.shape[0]),
                                   size=(int(ph[:, 0].max()) + 1, int(ph[:, 1].max()) + 1))


class _HashOp(nn.Module):
    """Shared plumbing: keeps the reference's sparse buffer (state-dict key) and a device handle."""
    _buffer_name = ""
    _transposed = False   # True: buffer is [V_fine, V_coarse]

    def __init__(self, mat: torch.Tensor):
        super().__init__()
        self.register_buffer(self._buffer_name, mat)
        self._handle = None
        # processing-order relabelling of the two vertex sets (set by MGCN when it reorders); the buffer
        # itself stays in the caller's numbering, so state dicts remain interchangeable with the reference
        self._rank_fine: Optional[torch.Tensor] = None
        self._rank_coarse: Optional[torch.Tensor] = None
        self._dist = None      # dist.DistPool when the model runs on a vertex partition

    def _pool(self) -> capi.PoolHandle:
        mat = getattr(self, self._buffer_name)
        h = self._handle
        if h is None or h.device != mat.device:
            r, c, shape = _pairs_from_sparse(mat)
            fine, coarse = (r, c) if self._transposed else (c, r)
            n_fine, n_coarse = (shape[0], shape[1]) if self._transposed else (shape[1], shape[0])
            if self._rank_fine is not None:
                fine = self._rank_fine.to(fine.device)[fine]
            if self._rank_coarse is not None:
                coarse = self._rank_coarse.to(coarse.device)[coarse]
            h = capi.PoolHandle(fine, coarse, n_fine, n_coarse)
            self._handle = h
        return h

    def _apply(self, fn, *args, **kwargs):   # .to(device) moves the buffer; drop the stale handle
        self._handle = None
        return super()._apply(fn, *args, **kwargs)

    def _load_from_state_dict(self, *args, **kwargs):   # a loaded pool_hash invalidates the handle as well
        self._handle = None
        return super()._load_from_state_dict(*args, **kwargs)


class MeshPool(_HashOp):
    """out[s] = mean over the fine vertices o with pool_hash[o] = s (util/meshnet.py:9-17)."""
    _buffer_name = "pool_hash"
    sg_pool_mode = 1          # (nn.Sequential folds it into the block call of the ChebConv in front: sg_block.pool_mode)

    def forward(self, input: torch.Tensor) -> torch.Tensor:
        if self._dist is not None:
            return self._dist.pool(input)
        return F_sg.mesh_pool(self._pool(), input)


class MeshUnpool(_HashOp):
    """out[o] = input[pool_hash[o]] (util/meshnet.py:20-27)."""
    _buffer_name = "unpool_hash"
    _transposed = True
    sg_pool_mode = 2

    def forward(self, input: torch.Tensor) -> torch.Tensor:
        if self._dist is not None:
            return self._dist.unpool(input)
        return F_sg.mesh_unpool(self._pool(), input)


def _conv_bn_act(cin: int, cout: int, K: int):
    return [(ChebConv(cin, cout, K=K), "x, edge_index -> x"), (nn.BatchNorm1d(cout), "x -> x"),
            (nn.LeakyReLU(), "x -> x")]


class DownConv(nn.Module):
    """Encoder stage (util/meshnet.py:31-95): conv-BN-act, conv, POOL, BN-act on the fine/coarse
    boundary (the pool sits between the 2nd conv and its BatchNorm), then 3 x conv-BN-act on the
    coarse level and a Dropout."""

    def __init__(self, in_channels, out_channels, edge_index1, edge_index2, pool_hash, K=3, drop_rate=0.0):
        super().__init__()
        self.edge_index1, self.edge_index2 = edge_index1, edge_index2   # plain attributes, as the reference
        c = out_channels
        self.model1 = Sequential("x, edge_index", _conv_bn_act(in_channels, c, K) + [
            (ChebConv(c, c, K=K), "x, edge_index -> x"), (MeshPool(pool_hash), "x -> x"),
            (nn.BatchNorm1d(c), "x -> x"), (nn.LeakyReLU(), "x -> x")])
        self.model2 = Sequential("x, edge_index", _conv_bn_act(c, c, K) + _conv_bn_act(c, c, K)
                                 + _conv_bn_act(c, c, K) + [(nn.Dropout(drop_rate), "x -> x")])

    def forward(self, input):
        g1, g2 = getattr(self, "_graphs", (self.edge_index1, self.edge_index2))
        return run_sequentials([(self.model1, g1), (self.model2, g2)], input)      # = model2(model1(input, g1), g2)


class UpConv(nn.Module):
    """Decoder stage (util/meshnet.py:98-160): conv, UNPOOL, BN-act, then 4 x conv-BN-act on the
    fine level and a Dropout."""

    def __init__(self, in_channels, out_channels, edge_index1, edge_index2, unpool_hash, K=3, drop_rate=0.0):
        super().__init__()
        self.edge_index1, self.edge_index2 = edge_index1, edge_index2
        c = out_channels
        self.model1 = Sequential("x, edge_index", [
            (ChebConv(in_channels, c, K=K), "x, edge_index -> x"), (MeshUnpool(unpool_hash), "x -> x"),
            (nn.BatchNorm1d(c), "x -> x"), (nn.LeakyReLU(), "x -> x")])
        self.model2 = Sequential("x, edge_index", _conv_bn_act(c, c, K) + _conv_bn_act(c, c, K)
                                 + _conv_bn_act(c, c, K) + _conv_bn_act(c, c, K)
                                 + [(nn.Dropout(drop_rate), "x -> x")])

    def forward(self, input):
        g1, g2 = getattr(self, "_graphs", (self.edge_index1, self.edge_index2))
        return run_sequentials([(self.model1, g1), (self.model2, g2)], input)      # = model2(model1(input, g1), g2)


def _head(cin: int, K: int) -> Sequential:
    return Sequential("x, edge_index", _conv_bn_act(cin, 32, K) + [(_Fp32Linear(32, 3), "x -> x")])


def _as_f32(a) -> torch.Tensor:
    """Mesh attribute (numpy array in the reference's Mesh, device tensor in meshprep.DeviceMesh) -> float32 tensor."""
    return a.float() if torch.is_tensor(a) else torch.from_numpy(np.asarray(a)).float()


class MGCN(nn.Module):
    def __init__(self, device, smo_mesh, ini_mesh, v_mask, K=3, skip=False, save_pooled: bool = False,
                 reorder: bool = True):
        super().__init__()
        self.reorder = reorder
        device = torch.device(device)
        nv = len(smo_mesh.vs)
        self.nvs = [int(nv * (POOL_RATIO ** i)) for i in range(1, POOL_LEVELS + 1)]
        meshes, pool_hashes = [smo_mesh], []
        if save_pooled:
            os.makedirs("{}/pooled".format(os.path.dirname(smo_mesh.path)), exist_ok=True)
        for i in range(POOL_LEVELS):
            s_mesh = meshes[i].simplification(target_v=self.nvs[i])   # the caller's Mesh class (util/mesh.py:394)
            meshes.append(s_mesh)
            pool_hashes.append(np.asarray(s_mesh.pool_hash, dtype=np.int64))
        edge_inds = [ini_mesh.edge_index] + [mm.edge_index for mm in meshes[1:]]
        smposs = [_as_f32(mm.vs) for mm in meshes]
        faces = [getattr(ini_mesh, "faces", None)] + [getattr(mm, "faces", None) for mm in meshes[1:]]
        self._build(device, edge_inds, pool_hashes, smposs, _as_f32(ini_mesh.vs),
                    v_mask, faces, K, skip)
        self.meshes = meshes
        if save_pooled:
            self._save_pooled(smo_mesh)

    @classmethod
    def from_hierarchy(cls, device, edge_inds: Sequence[torch.Tensor], pool_hashes: Sequence[np.ndarray],
                       smposs: Sequence[torch.Tensor], ini_pos: Optional[torch.Tensor] = None,
                       v_mask: Optional[torch.Tensor] = None, faces: Optional[Sequence] = None,
                       K: int = 3, skip: bool = False, reorder: bool = True) -> "MGCN":
        """Build from a precomputed hierarchy: ``edge_inds`` 4 x [2,E_l] (fine..coarse),
        ``pool_hashes`` 3 x int64 [V_l, 2] rows (fine_i, coarse_i), ``smposs`` 4 x [V_l, 3]."""
        self = cls.__new__(cls)
        nn.Module.__init__(self)
        self.reorder = reorder
        self.nvs = [int(len(smposs[0]) * (POOL_RATIO ** i)) for i in range(1, POOL_LEVELS + 1)]
        self.meshes = None
        if ini_pos is None:
            ini_pos = smposs[0]
        if v_mask is None:
            v_mask = torch.ones(len(smposs[0]), dtype=torch.bool)
        self._build(torch.device(device), list(edge_inds), [np.asarray(p, dtype=np.int64) for p in pool_hashes],
                    [torch.as_tensor(s).float() for s in smposs], torch.as_tensor(ini_pos).float(), v_mask,
                    list(faces) if faces is not None else [None] * 4, K, skip)
        return self

    # ------------------------------------------------------------------------------------
    def _build(self, device, edge_inds, pool_hashes, smposs, ini_pos, v_mask, faces, K, skip):
        self.device, self.skip = device, skip
        e = [torch.as_tensor(ei).long().to(device) for ei in edge_inds]
        self.edge_inds = e
        self.p_hashes = [pool_hash_to_mask(ph).to(device) for ph in pool_hashes]
        self.up_hashes = [unpool_hash_to_mask(ph).to(device) for ph in pool_hashes]
        self._pool_pairs = pool_hashes

        # masks per level (util/meshnet.py:177-199): a coarse vertex is kept iff all its members are
        vm = torch.as_tensor(v_mask).reshape(-1, 1).float().cpu()
        self.v_masks_list = [vm]
        for ph in pool_hashes:
            fine, coarse = torch.from_numpy(ph[:, 0]), torch.from_numpy(ph[:, 1])
            n_c = int(coarse.max()) + 1
            holes = torch.zeros(n_c, 1).index_add_(0, coarse, 1.0 - self.v_masks_list[-1][fine])
            self.v_masks_list.append((holes == 0).float())
        self.v_masks = torch.cat(self.v_masks_list, dim=0)
        self.f_masks_list = []
        for l, f in enumerate(faces):
            if f is None:
                self.f_masks_list.append(None)
            else:
                f = (f if torch.is_tensor(f) else torch.as_tensor(np.asarray(f))).long().cpu()
                self.f_masks_list.append((self.v_masks_list[l][:, 0][f] > 0).all(dim=1))
        self.f_masks = (torch.cat(self.f_masks_list, dim=0) if all(m is not None for m in self.f_masks_list) else None)

        p, u = self.p_hashes, self.up_hashes
        self.encoder1 = DownConv(4, 32, e[0], e[1], p[0], K=K, drop_rate=0.0)
        self.encoder2 = DownConv(32, 128, e[1], e[2], p[1], K=K, drop_rate=0.2)
        self.encoder3 = DownConv(128, 256, e[2], e[3], p[2], K=K, drop_rate=0.2)
        self.decoder3 = UpConv(256, 128, e[3], e[2], u[2], K=K, drop_rate=0.2)
        self.decoder2 = UpConv(128, 32, e[2], e[1], u[1], K=K, drop_rate=0.2)
        self.decoder1 = nn.Sequential(UpConv(32, 16, e[1], e[0], u[0], K=K, drop_rate=0.0), _Fp32Linear(16, 3))
        self.mcnn3, self.mcnn2, self.mcnn1 = _head(256, K), _head(128, K), _head(32, K)
        self.skip2, self.skip1 = _Fp32Linear(256, 128), _Fp32Linear(64, 32)     # (nn.Linear: fp32 parameters, any input dtype)
        self.feature_dtype = torch.float32

        # target / smooth positions per level (util/meshnet.py:251-276)
        self.smposs_list = [s.to(device) for s in smposs]
        self.smposs = torch.cat(self.smposs_list, dim=0)
        pos = ini_pos.to(device)
        self.poss_list = [pos]
        if device.type == "cuda":
            for l in range(POOL_LEVELS):
                fine = torch.from_numpy(pool_hashes[l][:, 0]).to(device)
                coarse = torch.from_numpy(pool_hashes[l][:, 1]).to(device)
                h = capi.PoolHandle(fine, coarse, self.poss_list[-1].shape[0], self.smposs_list[l + 1].shape[0])
                self.poss_list.append(h.pool_mean(self.poss_list[-1].contiguous()))
        else:  # construction on a CPU device is allowed (the reference builds there); forward is not
            for l in range(POOL_LEVELS):
                fine, coarse = torch.from_numpy(pool_hashes[l][:, 0]), torch.from_numpy(pool_hashes[l][:, 1])
                n_c = self.smposs_list[l + 1].shape[0]
                acc = torch.zeros(n_c, 3).index_add_(0, coarse, self.poss_list[-1][fine])
                cnt = torch.zeros(n_c, 1).index_add_(0, coarse, torch.ones(fine.numel(), 1))
                self.poss_list.append(acc / cnt)
        self.poss = torch.cat(self.poss_list, dim=0)
        self._orders = None
        if self.reorder and device.type == "cuda":
            self._prepare_processing_order()

    def _prepare_processing_order(self):
        """Every level is processed in Morton order of its smooth positions (like SingleScaleGCN): the
        per-level graphs are built on relabelled edges, pool/unpool handles relabel both vertex sets,
        the input is permuted once and the four outputs are returned in the caller's order."""
        orders = [_reorder.morton_order(p) for p in self.smposs_list]           # (order, rank) per level
        graphs = [MeshGraph.from_edge_index(_reorder.permute_edge_index(e, r), p.shape[0])
                  for e, (_, r), p in zip(self.edge_inds, orders, self.smposs_list)]
        for stage, (lf, lc) in ((self.encoder1, (0, 1)), (self.encoder2, (1, 2)), (self.encoder3, (2, 3))):
            stage._graphs = (graphs[lf], graphs[lc])
            stage.model1.module_4._rank_fine, stage.model1.module_4._rank_coarse = orders[lf][1], orders[lc][1]
        for stage, (lc, lf) in ((self.decoder3, (3, 2)), (self.decoder2, (2, 1)), (self.decoder1[0], (1, 0))):
            stage._graphs = (graphs[lc], graphs[lf])
            stage.model1.module_1._rank_fine, stage.model1.module_1._rank_coarse = orders[lf][1], orders[lc][1]
        self._orders, self._graphs = orders, graphs

    def _save_pooled(self, smo_mesh):
        root = os.path.dirname(smo_mesh.path)
        for l, s_mesh in enumerate(self.meshes[1:]):
            s_mesh.vc = np.ones([len(s_mesh.vs), 3]) * self.v_masks_list[l + 1].numpy()
            s_mesh.save("{}/pooled/{}_vs.obj".format(root, len(s_mesh.vs)), color=True)
            simp = copy.deepcopy(s_mesh)
            simp.vs = self.poss_list[l + 1].detach().cpu().numpy().copy()
            color = np.ones([len(simp.faces), 3]) * np.array([1.0, 0.0, 1.0])
            if self.f_masks_list[l + 1] is not None:
                color[self.f_masks_list[l + 1].numpy()] = np.array([0.332, 0.664, 1.0])
            simp.save_as_ply("{}/pooled/ini_{}_vs.ply".format(root, len(simp.vs)), color)

    def set_feature_dtype(self, dtype: torch.dtype) -> "MGCN":
        """Store the per-vertex features between kernels in ``dtype`` (float32 or bfloat16) on every level, as
        SingleScaleGCN.set_feature_dtype does: fp32 accumulation, fp32 parameters, fp32 heads and outputs."""
        if dtype not in (torch.float32, torch.bfloat16):
            raise ValueError("feature dtype must be float32 or bfloat16")
        self.feature_dtype = dtype
        return self

    # ------------------------------------------------------------------------------------
    def forward(self, data, dm=None):
        z1 = data.z1.to(self.device)
        # quirk kept (util/meshnet.py:287-290): anything but an ndarray becomes an all-ones mask,
        # so the Tensor that mgcn.py:128-134 passes is ignored
        if isinstance(dm, np.ndarray):
            mask = torch.from_numpy(dm).to(self.device)
        else:
            mask = torch.ones((z1.shape[0], 1), dtype=z1.dtype, device=self.device)
        heads = self.edge_inds
        part = getattr(self, "_part", None)
        fd = getattr(self, "feature_dtype", torch.float32)
        fused_io = part is None and z1.is_cuda and z1.dtype == torch.float32 and mask.numel() == z1.shape[0]
        if fused_io:
            # one launch for the bounds' normalisation, the mask, the processing order and the feature dtype (as SingleScaleGCN)
            order, rank = self._orders[0] if self._orders is not None else (None, None)
            x = F_sg.input_prep(z1, None, None, mask.to(z1.dtype), order, rank, fd)
            if self._orders is not None:
                heads = self._graphs
        else:
            x = prepare_input(z1, mask.to(z1.dtype))
            if part is not None:
                # vertex-partitioned (dist.partition_mgcn): the input is the whole mesh (replicated, cheap: [V,4]);
                # every level continues with this rank's rows only and the outputs are this rank's rows
                x = x.index_select(0, part.own_ids[0])
                heads = part.graphs
            elif self._orders is not None:
                x = x.index_select(0, self._orders[0][0])
                heads = self._graphs
            x = x.to(fd)

        res1_enc = self.encoder1(x)
        res2_enc = self.encoder2(res1_enc)
        res3_bot = self.encoder3(res2_enc)
        out3 = self.mcnn3(res3_bot, heads[3])

        res2_dec = self.decoder3(res3_bot)
        if self.skip:
            res2_dec = self.skip2(torch.cat([res2_dec, res2_enc], dim=1)).to(fd)
        out2 = self.mcnn2(res2_dec, heads[2])

        res1_dec = self.decoder2(res2_dec)
        if self.skip:
            res1_dec = self.skip1(torch.cat([res1_dec, res1_enc], dim=1)).to(fd)
        out1 = self.mcnn1(res1_dec, heads[1])

        out0 = self.decoder1(res1_dec)
        outs = [out0, out1, out2, out3]
        if part is not None:
            return tuple(s_own + o for s_own, o in zip(part.smposs_own, outs))
        s = self.smposs_list
        if self._orders is not None:
            if fused_io:      # base + rows back in caller order, one autograd node each (its backward is a gather, no atomics)
                return tuple(F_sg.output_in_caller_order(o, sm, rank, order) for o, sm, (order, rank) in zip(outs, s, self._orders))
            outs = [o.index_select(0, rank) for o, (_, rank) in zip(outs, self._orders)]
        return (s[0] + outs[0], s[1] + outs[1], s[2] + outs[2], s[3] + outs[3])

    # helpers the reference exposes (util/meshnet.py:320-341)
    def pool(self, dx, pool_hash):
        return MeshPool(pool_hash.to(dx.device))(dx)

    def unpool(self, dx, unpool_hash):
        return MeshUnpool(unpool_hash.to(dx.device))(dx)

    pool_hash_to_mask = staticmethod(pool_hash_to_mask)
    unpool_hash_to_mask = staticmethod(unpool_hash_to_mask)
