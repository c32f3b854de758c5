#!/usr/bin/env python3
"""Host side of SGCN training iterations on a small mesh (the CPU-launch-bound regime a rank of an 8-way partition
of the 1 M mesh is in): wall time per iteration, time until the last launch is enqueued, and a cProfile with the
backward pass pulled onto the calling thread (autograd multithreading off) so that its Python frames are visible.

    python tools/host_profile.py [250x200] [bf16|fp32] [N lines] [partitioned]

``partitioned``: ONE rank runs the partitioned code path with every collective issued through RCCL (dist.FORCE_COLLECTIVES):
the host work of a rank of an N-rank job.
"""
import cProfile, os, pstats, sys, io, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
MESH = sys.argv[1] if len(sys.argv) > 1 else "250x200"
DT = sys.argv[2] if len(sys.argv) > 2 else "bf16"
NL = int(sys.argv[3]) if len(sys.argv) > 3 else 45
PART = len(sys.argv) > 4 and sys.argv[4] == "partitioned"
sys.argv = ["bench.py", "--mesh", MESH]
import bench
from semigcn_amd import synth, train
from semigcn_amd.networks import SingleScaleGCN
dev = torch.device("cuda:0")
mesh = synth.torus_mesh(*map(int, MESH.split("x")))
if PART:
    import torch.distributed as dist
    from semigcn_amd import dist as sgdist
    sgdist.FORCE_COLLECTIVES = True
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", str(bench.free_port()))
    dist.init_process_group("nccl", device_id=dev, rank=0, world_size=1)
    nu, nv = map(int, MESH.split("x"))
    tr = sgdist.build_partitioned_job(nu, nv, 1, 0, dev, dtype=torch.bfloat16 if DT == "bf16" else torch.float32, mesh=mesh).trainer
else:
    batch = bench.build_mesh_batch(mesh, dev, 5)
    model = SingleScaleGCN(dev).to(dev)
    if DT == "bf16":
        model.set_feature_dtype(torch.bfloat16)
    tr = train.SGCNTrainer(model, batch)


def run(n):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): tr.iteration_step()
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    return (t1 - t0) / n * 1e3, (t2 - t0) / n * 1e3


from semigcn_amd import functional as F_sg
for _ in range(10): tr.iteration_step()
for rnd in range(3):          # A/B in one process, interleaved: parameter gradients sunk by the kernels vs through autograd
    for on in (True, False):
        F_sg.SINK_PARAM_GRADS = on
        for _ in range(3): tr.iteration_step()
        print(f"{MESH} {DT}: grad sinks {'on ' if on else 'off'}: enqueue %.2f ms  wall %.2f ms per iteration" % run(40))
F_sg.SINK_PARAM_GRADS = True
print(f"{MESH} {DT}: enqueue %.2f ms  wall %.2f ms per iteration (multithreaded autograd)" % run(40))
with torch.autograd.set_multithreading_enabled(False):
    for _ in range(5): tr.iteration_step()
    print(f"{MESH} {DT}: enqueue %.2f ms  wall %.2f ms per iteration (backward on the calling thread)" % run(40))
    pr = cProfile.Profile(); pr.enable()
    for _ in range(20): tr.iteration_step()
    torch.cuda.synchronize(); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(NL); print(s.getvalue()[:12000])
