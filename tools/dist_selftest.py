#!/usr/bin/env python3
"""Self-test of the vertex-partitioned path with the REAL HIP kernels on ONE GPU:

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 \\
        --master-port 29555 tools/dist_selftest.py

Both ranks use cuda:0 and talk over gloo (RCCL refuses two ranks on one device; the library
stages device tensors through the host under gloo).  Checks the partitioned forward, loss and
reduced parameter gradients against the plain single-GPU model on the whole mesh."""
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    import golden_util as GU
    from semigcn_amd import dist as sgdist, reorder, synth, train
    from semigcn_amd.networks import SingleScaleGCN
    import bench

    mesh = synth.torus_mesh(96, 64, permute=True)
    part = sgdist.partition_mesh(mesh, rank, world, dev, n_masks=2)
    model = SingleScaleGCN(dev)
    GU.fill_state(model, seed=77)
    model.to(dev)
    tr = sgdist.DistSGCNTrainer(model, part, accumulate=1000)
    model.train()
    dm = part.v_keep * part.dummy_masks[:, :1]
    pos = model(part, dm)
    loss = tr.loss(pos)
    loss.backward()
    sgdist.all_reduce_gradients(tr.params)

    # single-GPU reference on the whole mesh (plain BatchNorm, no partition)
    ref = SingleScaleGCN(dev)
    GU.fill_state(ref, seed=77)
    ref.to(dev).train()
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
    print(f"[rank {rank}/{world}] own={g.n_own} halo={g.n_halo} send={g.n_send}  pos rel-L2 {e_pos:.2e}  "
          f"loss rel {e_loss:.2e}  worst param-grad rel-L2 {worst:.2e}", flush=True)
    assert e_pos < 2e-5 and e_loss < 2e-5 and worst < 3e-2
    dist.barrier()
    if rank == 0:
        print("dist_selftest OK")
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
