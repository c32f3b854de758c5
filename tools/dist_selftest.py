#!/usr/bin/env python3
"""Self-test of the vertex-partitioned path with the REAL HIP kernels on ONE GPU:

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 \\
        --master-port 29555 tools/dist_selftest.py

All ranks use cuda:0 and talk over gloo (RCCL refuses two ranks on one device; the library
stages device tensors through the host under gloo); SEMIGCN_SELFTEST_BACKEND=nccl runs one rank
per GPU over RCCL instead.  tests/test_gpu_scale.py runs it with 2 and 4 ranks under ``-m gpu``.  Checks the partitioned forward, loss and
reduced parameter gradients against the plain single-GPU model on the whole mesh."""
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    backend = os.environ.get("SEMIGCN_SELFTEST_BACKEND", "gloo")
    if backend == "nccl":          # one GPU per rank, RCCL over xGMI (needs >= world devices)
        dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", rank)))
        torch.cuda.set_device(dev)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    else:                          # every rank on cuda:0, collectives staged through the host
        dist.init_process_group("gloo", rank=rank, world_size=world)
        dev = torch.device("cuda:0")
        torch.cuda.set_device(dev)
    import golden_util as GU
    from semigcn_amd import dist as sgdist, reorder, synth, train
    from semigcn_amd.networks import SingleScaleGCN
    import bench

    nu, nv = map(int, os.environ.get("SEMIGCN_SELFTEST_MESH", "96x64").split("x"))
    mesh = synth.torus_mesh(nu, nv, permute=True)
    part = sgdist.partition_mesh(mesh, rank, world, dev, n_masks=2)
    model = SingleScaleGCN(dev)
    GU.fill_state(model, seed=77)
    model.to(dev)
    # SEMIGCN_SELFTEST_DTYPE=bf16: bf16 feature storage on both sides (what bench.py runs); the comparison is then between two
    # bf16 evaluations that differ in the order of their fp32 additions (tile moments per rank, halo rows computed twice)
    bf16 = os.environ.get("SEMIGCN_SELFTEST_DTYPE") == "bf16"
    if bf16:
        model.set_feature_dtype(torch.bfloat16)
    tr = sgdist.DistSGCNTrainer(model, part, accumulate=1000)
    model.train()
    dm = part.v_keep * part.dummy_masks[:, :1]
    # SEMIGCN_SELFTEST_PATH: "modules" = every module on its own with the exchange inside each conv and an all-gather per
    # BatchNorm (57 collectives per iteration); "phases" (default) = the blocks phase by phase below the C ABI with the
    # statistics riding in the halo exchange (dist.part_chain: 44); "phases-sunk": the same with the parameter gradients
    # added into the .grad accumulators by the library (what the trainers do)
    path = os.environ.get("SEMIGCN_SELFTEST_PATH", "phases")
    part.halo_inputs = None if path == "modules" else (part.z1_halo, part.dm_halo[:, :1].contiguous())
    from semigcn_amd import functional as F_sg
    c0, b0 = dict(sgdist.collective_counts), list(F_sg.block_calls)
    pos = model(part, dm)
    loss = tr.loss(pos)
    if path == "phases-sunk":
        tr.grads.zero()
        with F_sg.sink_param_grads():
            loss.backward()
    else:
        loss.backward()
    n_coll = sum(sgdist.collective_counts.values()) - sum(c0.values())
    blocks = [F_sg.block_calls[0] - b0[0], F_sg.block_calls[1] - b0[1]]
    if path != "modules":
        assert blocks == [13, 13], blocks
        assert n_coll == 44 or not (world > 1 or sgdist.FORCE_COLLECTIVES), n_coll
    elif world > 1 or sgdist.FORCE_COLLECTIVES:
        assert n_coll == 57, n_coll
    sgdist.all_reduce_gradients(tr.params)
    # what this rank computed, bit for bit (two runs of one path on the same inputs -- e.g. the collectives below the C ABI
    # against the same collectives through torch.distributed -- must print the same line), and how the passes were issued
    import hashlib
    h = hashlib.sha256()
    h.update(pos.detach().float().cpu().numpy().tobytes())
    h.update(loss.detach().float().cpu().numpy().tobytes())
    for q in model.parameters():
        if q.grad is not None:
            h.update(q.grad.detach().float().cpu().numpy().tobytes())
    print(f"[rank {rank}/{world}] digest {h.hexdigest()[:24]} native_part_runs={sgdist.native_runs}", flush=True)
    if path != "modules":
        # no buffer of the phase path keeps statistics bytes (or anything else that is not a finite feature value) in the pad
        # rows of the peers' segments: SG_PHASE_BN clears them in H once it has read them
        n_checked = 0
        for pc in part.graph.__dict__.get("_part_chains", {}).values():
            for bufs in pc._free.values():
                for b in bufs:
                    for t in b.inp + b.H:
                        assert bool(torch.isfinite(t.float()).all()), "a non-finite value in a phase-path buffer"
                        n_checked += 1
        assert n_checked >= 26 or world == 1, n_checked

    if os.environ.get("SEMIGCN_SELFTEST_CROSS") == "1" and path == "phases":
        # the SAME partitioned model once more on the per-module path: the two differ in where BatchNorm's moments are merged
        # and in which launch computes a halo row, not in the arithmetic -- with bf16 storage this pins the phase path far
        # tighter than the comparison with the single-device model can (see the bf16 bounds below)
        import copy
        twin = copy.deepcopy(model)
        for p in twin.parameters():
            p.grad = None
        GU.fill_state(twin, seed=77)
        twin.to(dev)
        if bf16:
            twin.set_feature_dtype(torch.bfloat16)
        twin.train()
        part.halo_inputs = None
        tw = sgdist.DistSGCNTrainer(twin, part, accumulate=1000, phases=False)
        pos2 = twin(part, dm)
        loss2 = tw.loss(pos2)
        loss2.backward()
        sgdist.all_reduce_gradients(tw.params)
        d_pos = float((pos.detach() - pos2.detach()).norm() / pos2.detach().norm())
        d_off = float((pos.detach() - pos2.detach()).norm() / (pos2.detach() - part.x_pos).norm())
        d_loss = abs(float(loss) - float(loss2)) / abs(float(loss2))
        gmax2 = max(float(q.grad.abs().max()) for q in twin.parameters() if q.grad is not None)
        d_grad = max(float((p.grad - q.grad).norm()) / max(float(q.grad.norm()), 1e-3 * gmax2 * q.grad.numel() ** 0.5)
                     for p, q in zip(model.parameters(), twin.parameters()) if q.grad is not None)
        print(f"[rank {rank}/{world}] phases vs per-module path on the same partition: positions {d_pos:.2e} (offsets {d_off:.2e})  "
              f"loss {d_loss:.2e}  worst param-grad {d_grad:.2e}", flush=True)
        if not bf16:          # (measured: 5e-8 / 7e-8 / 7e-4; bf16 storage: 1.2e-3 / 2.6e-4 / 1.3 -- see the bf16 bounds below)
            assert d_pos < 1e-6 and d_loss < 1e-6 and d_grad < 5e-3, (d_pos, d_loss, d_grad)
        part.halo_inputs = (part.z1_halo, part.dm_halo[:, :1].contiguous())

    # single-GPU reference on the whole mesh (plain BatchNorm, no partition)
    ref = SingleScaleGCN(dev)
    GU.fill_state(ref, seed=77)
    ref.to(dev).train()
    if bf16:
        ref.set_feature_dtype(torch.bfloat16)
    batch = bench.build_mesh_batch(mesh, dev, n_masks=2)
    rt = train.SGCNTrainer(ref, batch, accumulate=1000)
    dm_full = batch.v_keep * batch.dummy_masks[:, :1]
    rpos = ref(batch.data, dm_full)
    rloss = rt.loss(rpos)
    rloss.backward()

    order = reorder.morton_order(torch.from_numpy(mesh.x_pos).to(dev))[0]
    g = part.graph
    mine = rpos.detach()[order[g.start:g.end]]
    e_pos = float((pos.detach() - mine).norm() / mine.norm())
    e_loss = abs(float(loss) - float(rloss)) / abs(float(rloss))
    worst = 0.0
    gmax = max(float(p.grad.abs().max()) for p in ref.parameters() if p.grad is not None)
    for (n, p), (_, q) in zip(model.named_parameters(), ref.named_parameters()):
        if q.grad is None:
            continue
        scale = max(float(q.grad.norm()), 1e-3 * gmax * q.grad.numel() ** 0.5)
        worst = max(worst, float((p.grad - q.grad).norm()) / scale)
    if path == "phases" and os.environ.get("SEMIGCN_SELFTEST_PREFIX", "1") == "1":
        block_prefix_check(model, ref, part, batch, rank, world, dev, bf16)
    print(f"[rank {rank}/{world}] path={path} collectives={n_coll} own={g.n_own} halo={g.n_halo} send={g.n_send}  pos rel-L2 {e_pos:.2e}  "
          f"loss rel {e_loss:.2e}  worst param-grad rel-L2 {worst:.2e}", flush=True)
    if bf16:
        # measured (2 and 4 ranks, 96 x 64 torus; the per-module path reads the same): positions 1.9e-3 (the offsets are ~1e-2 of
        # the positions), loss 2e-4 .. 4e-4, worst parameter gradient 1.6 -- two bf16 evaluations that differ only in the order
        # of their fp32 additions do not stay together through 13 BatchNorm layers; what this run pins
        # is that nothing GROSS is wrong with bf16 rows on a partition (statistics in bf16 pad rows, halo rows, packing)
        assert e_pos < 4e-3 and e_loss < 5e-3 and worst < 3.0, (e_pos, e_loss, worst)
    else:
        assert e_pos < 1e-5 and e_loss < 2e-6 and worst < 3e-2
    if os.environ.get("SEMIGCN_SELFTEST_SKIP_MGCN") != "1":
        mgcn_selftest(rank, world, dev, phases=(path != "modules"))
    dist.barrier()
    if rank == 0:
        import json
        print("collectives " + json.dumps({"backend": dist.get_backend(), "world_size": world, **sgdist.collective_counts}))
        print("dist_selftest OK")
    dist.destroy_process_group()


def block_prefix_check(model, ref, part, batch, rank, world, dev, bf16):
    """The first THREE blocks (4 -> 16 -> 32 -> 64: thin products, planes, both kinds of engines) through ``part_chain`` on
    this rank's rows against the same three blocks of the single-device model on the whole mesh, same input and same
    upstream gradient: three BatchNorm layers deep a bf16 run has not yet diverged, so this pins the
    phase path with bf16 rows -- statistics in bf16 pad rows, 2-byte halo and gradient rows -- to a few bf16 roundings."""
    import torch.distributed as dist
    from semigcn_amd import dist as sgdist
    from semigcn_amd.nn import run_sequentials
    dtype = torch.bfloat16 if bf16 else torch.float32
    g = part.graph
    graph, order, rnk = ref._layout(batch.data)
    V = graph.num_vertices
    gen = torch.Generator(device=dev).manual_seed(12)
    x_all = torch.randn(V, 4, device=dev, generator=gen).to(dtype)              # processing order
    r_all = torch.randn(V, 64, device=dev, generator=gen)
    lay = g.folded()
    for m in (model, ref):
        for p in m.parameters():
            p.grad = None
    x_own = x_all[g.start:g.end].clone().requires_grad_(True)
    res = sgdist.part_chain(list(model.blocks[:3]), g, x_own, lay.halo_of(x_all).contiguous())
    assert res is not None and res[1] == 3, "part_chain did not take the three blocks"
    y = res[0]
    (y.float() * r_all[g.start:g.end]).sum().backward()
    sgdist.all_reduce_gradients([p for blk in model.blocks[:3] for p in blk.parameters()])
    xf = x_all.clone().requires_grad_(True)
    yf = run_sequentials([(blk, graph) for blk in ref.blocks[:3]], xf)
    (yf.float() * r_all).sum().backward()
    e_y = float((y.detach().float() - yf.detach().float()[g.start:g.end]).norm() / yf.detach().float()[g.start:g.end].norm())
    e_dx = float((x_own.grad.float() - xf.grad.float()[g.start:g.end]).norm() / xf.grad.float()[g.start:g.end].norm())
    e_w = 0.0
    for bp, br in zip(model.blocks[:3], ref.blocks[:3]):
        for (n, p), (_, q) in zip(bp.named_parameters(), br.named_parameters()):
            if q.grad is None or n.endswith("module_0.bias"):          # (a conv bias in front of a BatchNorm: true gradient zero)
                continue
            e_w = max(e_w, float((p.grad - q.grad).norm() / q.grad.norm()))
    print(f"[rank {rank}/{world}] three blocks, {'bf16' if bf16 else 'fp32'} rows: output {e_y:.2e}  input gradient {e_dx:.2e}  "
          f"parameter gradients {e_w:.2e}", flush=True)
    # measured: bf16 (2 / 4 ranks, 6 K vertices) 1.0-1.3e-3 / 1.2-1.9e-2 / 2.2e-2; fp32 4.4e-7 / 2.6e-7 / 2.2e-6 there and
    # 4.3e-7 / 7e-6 .. 3e-3 / 2.1e-3 on the 200 K-vertex mesh with 8 ranks: the gradients are kink-limited (an output within
    # 1e-7 of zero takes the other LeakyReLU slope), the forward output is not
    tol = (2.5e-3, 5e-2, 5e-2) if bf16 else (2e-6, 1e-2, 1e-2)
    assert e_y < tol[0] and e_dx < tol[1] and e_w < tol[2], (e_y, e_dx, e_w)
    for m in (model, ref):
        for p in m.parameters():
            p.grad = None


def mgcn_selftest(rank, world, dev, phases=True):
    """Partitioned MGCN (pool / unpool across the cut) against the plain single-GPU MGCN, real kernels.  ``phases``: the runs of
    plain blocks of every stage phase by phase below the C ABI (dist.part_blocks: 27 of the 33 blocks), the pooled blocks module
    by module; False: every module on its own (rounds 1-4)."""
    import numpy as np
    import golden_util as GU
    from semigcn_amd import dist as sgdist, meshprep, synth, train
    from semigcn_amd.meshnet import MGCN
    import bench

    mesh = synth.torus_mesh(96, 64, permute=True)
    batch = bench.build_mesh_batch(mesh, dev, n_masks=2)

    def build():
        smo = meshprep.DeviceMesh(mesh.x_pos, mesh.faces, dev)
        ini = meshprep.DeviceMesh(mesh.vs.astype(np.float32), mesh.faces, dev)
        net = MGCN(dev, smo, ini, torch.from_numpy(mesh.v_mask))
        GU.fill_state(net, seed=5)
        for mod in net.modules():
            if isinstance(mod, torch.nn.Dropout):
                mod.p = 0.0
        return net.to(dev)

    ref = build()
    rt = train.MGCNTrainer(ref, batch, accumulate=1000)
    ref.train()
    rposs = ref(batch.data, None)
    rloss = sum(w * train.masked_position_rmse(p, t, k, n)
                for w, p, t, k, n in zip(rt.weights, rposs, ref.poss_list, rt.keeps, rt.counts)) \
        + rt.k1 * train.masked_normal_l1(train.face_normals(rposs[0], batch.faces), batch.target_fn, batch.f_keep,
                                         batch.n_f_keep)
    rloss.backward()

    net = build()
    part = sgdist.partition_mgcn(net, rank, world, phases=phases)
    tr = sgdist.DistMGCNTrainer(net, part, batch, accumulate=1000)
    net.train()
    from semigcn_amd import functional as F_sg
    c0, b0 = dict(sgdist.collective_counts), list(F_sg.block_calls)
    poss = net(batch.data, None)
    loss = tr.loss(poss)
    if phases:
        tr.grads.zero()
        with F_sg.sink_param_grads():       # (what DistMGCNTrainer.iteration_step does: the backward pass of a run is one call too)
            loss.backward()
    else:
        loss.backward()
    n_coll = sum(sgdist.collective_counts.values()) - sum(c0.values())
    blocks = [F_sg.block_calls[0] - b0[0], F_sg.block_calls[1] - b0[1]]
    if phases and dev.type == "cuda":
        assert blocks == [27, 27], blocks       # 3 x 4 (encoders), 3 x 4 (decoders), the conv block of each of the 3 heads
    sgdist.all_reduce_gradients(tr.params)
    e_pos = max(float((p.detach() - r.detach()[ids]).norm() / r.detach()[ids].norm())
                for p, r, ids in zip(poss, rposs, part.own_ids))
    e_loss = abs(float(loss) - float(rloss)) / abs(float(rloss))
    gmax = max(float(p.grad.abs().max()) for p in ref.parameters() if p.grad is not None)
    worst = 0.0
    for (n, p), (_, q) in zip(net.named_parameters(), ref.named_parameters()):
        if q.grad is None:
            continue
        scale = max(float(q.grad.norm()), 1e-3 * gmax * q.grad.numel() ** 0.5)
        worst = max(worst, float((p.grad - q.grad).norm()) / scale)
    halos = [(g.n_own, g.n_halo) for g in part.graphs]
    print(f"[rank {rank}/{world}] MGCN phases={phases} blocks below the C ABI {blocks} collectives {n_coll}", flush=True)
    print(f"[rank {rank}/{world}] MGCN levels (own, halo) {halos} pool halos "
          f"{[(q.fine_plan.n_halo, q.coarse_plan.n_halo) for q in part.pools]}  pos rel-L2 {e_pos:.2e}  "
          f"loss rel {e_loss:.2e}  worst param-grad rel-L2 {worst:.2e}", flush=True)
    # (reduced gradients through 33 BatchNorm + LeakyReLU layers with dropout-free kinks: one sign that differs between the
    #  partitioned and the single-device run moves a small layer's gradient by a few per cent; the hierarchy's clusters of
    #  up to five make the coarse levels smaller than round 2's pairs did)
    assert e_pos < 1e-5 and e_loss < 5e-6 and worst < 8e-2


if __name__ == "__main__":
    main()
