"""Run by tests/test_gpu_scale.py in SUBPROCESS ranks (torch.distributed.run) whose environment carries
DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 from the start: the partitioned SGCN iteration recorded as hipGraph segments with the
collectives between them (semigcn_amd/segments.py) must give, over nine iterations (three eager warm-ups, the recording,
five replays; one Adam step inside), bit for bit the losses, parameters and BatchNorm statistics of nine eager iterations
of the same partitioned trainer.  SEMIGCN_SELFTEST_BACKEND=nccl: one rank per GPU over RCCL (a single rank then needs
SEMIGCN_DIST_FORCE_COLLECTIVES=1 to issue its collectives); gloo: every rank on cuda:0, collectives staged through the host."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    backend = os.environ.get("SEMIGCN_SELFTEST_BACKEND", "gloo")
    if backend == "nccl":
        dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", rank)))
        torch.cuda.set_device(dev)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
        dev = torch.device("cuda:0")
        torch.cuda.set_device(dev)
    import golden_util as GU
    from semigcn_amd import dist as sgdist, synth
    from semigcn_amd.networks import SingleScaleGCN

    mesh = synth.torus_mesh(96, 64, permute=True)
    dtype = torch.bfloat16 if os.environ.get("SEMIGCN_SELFTEST_DTYPE") == "bf16" else torch.float32

    def run(capture):
        part = sgdist.partition_mesh(mesh, rank, world, dev, n_masks=3)
        model = SingleScaleGCN(dev)
        GU.fill_state(model, seed=91)
        model.to(dev)
        if dtype != torch.float32:
            model.set_feature_dtype(dtype)
        tr = sgdist.DistSGCNTrainer(model, part, capture=capture, phases=False)     # segments replay the per-module path
        losses = []
        for _ in range(9):
            _ = part.v_keep * 2.0                  # an unrelated eager kernel between iterations ...
            losses.append(float(tr.iteration_step()))
            torch.cuda.synchronize()               # ... and an idle GPU before the next replay
        if capture is True:
            assert tr._segmented.rec is not None
            print(f"[rank {rank}] tape: %d graph segments, %d eager actions" % tr._segmented.rec.counts(), flush=True)
        return losses, {k: v.clone() for k, v in model.state_dict().items()}

    mode = os.environ.get("SEMIGCN_SELFTEST_CAPTURE", "segments")
    le, se = run(False)
    c0 = dict(sgdist.collective_counts)
    lg, sg = run(True if mode == "segments" else mode)
    c1 = dict(sgdist.collective_counts)
    assert lg == le, (lg, le)
    for k in se:
        assert torch.equal(se[k], sg[k]), k
    # the replays issued the same collectives as the eager iterations
    per_run = {k: c1[k] - c0[k] for k in c0}
    if mode == "segments":
        assert all(per_run[k] >= 9 * n for k, n in (("all_to_all", 28), ("all_gather", 13))), per_run
    dist.barrier()
    if rank == 0:
        print("SEGMENT_REPLAY_OK", backend, world, str(dtype), le[-1], per_run)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
