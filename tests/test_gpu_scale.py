"""``-m gpu`` tests at the benchmark sizes and of the N > 1 path with the real kernels.

  * V = 1 M (BASELINE config c4): SGCN outputs on sampled rows against the ORACLE evaluated on the
    rows' dependency neighbourhood (fp32, 1e-5), the bf16-feature run against the fp32 one;
  * V = 4 M (config c5's mesh, 2000 x 2000) on one GPU: aggregation properties and one training
    iteration in fp32 and with bf16 features;
  * the vertex-partitioned path (sg_graph_create_rect, sg_bn_finalize_ranks, DistPool, halo exchange)
    with 2 and 4 ranks sharing cuda:0 over gloo == the single-rank model on the device;
  * ``bench.py --gpus N`` starting its own ranks.
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

import golden_util as GU
from oracle import models as OM          # checker only
from semigcn_amd import synth, train
from semigcn_amd.graph import MeshGraph
from semigcn_amd.networks import SingleScaleGCN

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _child_env(**extra):
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["OMP_NUM_THREADS"] = "4"
    env.update(extra)
    return env


# --------------------------------------------------------------------------------------
# V = 1 M: model outputs on sampled rows vs the oracle on their dependency neighbourhood
# --------------------------------------------------------------------------------------
RADIUS = 34          # grid half-width of a patch
VALID = 6            # rows within this grid distance of the patch centre are exact (see below)


def _patch(m, cu, cv):
    """Sub-mesh of the nu x nv torus grid around grid position (cu, cv), away from the periodic seam: vertex ids
    (global), induced directed edges relabelled to patch ids, and the patch ids of the rows whose 13-layer K=3
    dependency cone lies inside the patch.  One ChebConv reads 2 rings, so 13 layers read 26 rings; an edge weight
    -dis[i]*dis[j] needs the FULL degree of j, i.e. j's neighbours: rows within RADIUS - 28 of the centre are exact
    (a flipped quad diagonal still joins grid neighbours, so one ring <= one grid step)."""
    nu, nv = m.nu, m.nv
    assert RADIUS <= cu < nu - RADIUS and RADIUS <= cv < nv - RADIUS and RADIUS - 28 >= VALID
    uu, vv = np.meshgrid(np.arange(cu - RADIUS, cu + RADIUS + 1), np.arange(cv - RADIUS, cv + RADIUS + 1), indexing="ij")
    ids = (uu * nv + vv).ravel()
    local = np.full(m.num_vertices, -1, np.int64)
    local[ids] = np.arange(ids.size)
    ei = m.edge_index
    keep = (local[ei[0]] >= 0) & (local[ei[1]] >= 0)
    ei_loc = np.stack([local[ei[0][keep]], local[ei[1][keep]]])
    centre = (np.abs(uu - cu) <= VALID) & (np.abs(vv - cv) <= VALID)
    return ids, ei_loc, np.nonzero(centre.ravel())[0]


def test_full_size_sgcn_rows_vs_oracle_and_bf16_step():
    """SGCN at V = 1 M in eval mode (BatchNorm on running statistics that one full-size training pass has moved):
    338 output rows against the oracle run on two 69 x 69 patches, 1e-5; then the same forward with bf16 feature
    storage against the fp32 one (13 layers x ~5 roundings to 8 bits, BatchNorm-amplified: rel-L2 of the offsets
    < 0.2, the bound of test_sgcn_bf16_features_close_to_fp32) and a bf16 training step."""
    m = synth.torus_mesh(1000, 1000)
    V = m.num_vertices
    net = SingleScaleGCN(DEV)
    GU.fill_state(net, seed=1234)
    net.to(DEV)
    # the model returns x_pos + offset in fp32; with |x_pos| ~ 500 the sum carries 3e-5 of rounding, which would hide
    # the network's own error: shrink the positions (same Morton order) so that out - x_pos IS the offset to ~1e-8
    xp_small = (m.x_pos * np.float32(1e-3)).astype(np.float32)

    class D:
        z1 = torch.from_numpy(m.z1).to(DEV).requires_grad_(True)
        x_pos = torch.from_numpy(xp_small).to(DEV)
        edge_index = torch.from_numpy(m.edge_index).to(DEV)
    dm_np = synth.make_dummy_masks(m.edge_index, V, 1, k=4, p=0.014, seed=317)
    dm = torch.from_numpy(dm_np).to(DEV)
    net.train()
    net(D, dm)                                         # moves every running_mean / running_var off its initial value
    net.eval()
    with torch.no_grad():
        out = net(D, dm)
    assert out.shape == (V, 3) and bool(torch.isfinite(out).all())

    ora = OM.SGCNOracle()
    ora.load_state_dict({k: v.cpu() for k, v in net.state_dict().items()})
    ora.eval()
    lo, hi = m.z1.min(0), m.z1.max(0)
    checked, worst = 0, 0.0
    for cu, cv in ((300, 300), (700, 640)):
        ids, ei_loc, rows = _patch(m, cu, cv)
        n = ids.size
        # two isolated extra vertices carry the mesh-wide bounding box into the oracle's own normalisation
        z1 = np.concatenate([m.z1[ids], lo[None], hi[None]]).astype(np.float32)
        xp = np.concatenate([xp_small[ids], np.zeros((2, 3), np.float32)])
        dmp = np.concatenate([dm_np[ids], np.ones((2, 1), np.float32)])
        with torch.no_grad():
            ref = ora(torch.from_numpy(z1), torch.from_numpy(xp), torch.from_numpy(ei_loc), torch.from_numpy(dmp))[:n]
        got = out[torch.from_numpy(ids[rows]).to(DEV)].cpu()
        off_ref = ref[rows] - torch.from_numpy(xp_small[ids[rows]])         # compare the network's offsets, not x_pos + offset
        off_got = got - torch.from_numpy(xp_small[ids[rows]])
        err = float((off_got - off_ref).abs().max() / off_ref.abs().max())
        worst = max(worst, err)
        assert err < 1e-5, (cu, cv, err)
        assert GU.rel_l2(off_got, off_ref) < 1e-5
        checked += rows.size
    assert checked >= 256
    print(f"V=1M eval rows vs oracle patches: {checked} rows, worst max-rel error {worst:.2e}")

    # bf16 feature storage at full size: forward against fp32, then one training step
    net.set_feature_dtype(torch.bfloat16)
    with torch.no_grad():
        out16 = net(D, dm)
    e16 = GU.rel_l2((out16 - D.x_pos).cpu(), (out - D.x_pos).cpu())
    assert out16.dtype == torch.float32 and e16 < 0.2, e16
    net.train()
    pos = net(D, dm)
    (pos - D.x_pos).square().mean().backward()
    assert all(p.grad is not None and p.grad.dtype == torch.float32 and bool(torch.isfinite(p.grad).all())
               for nme, p in net.named_parameters() if not nme.startswith("skip_blocks"))
    assert bool(torch.isfinite(D.z1.grad).all())
    print(f"V=1M bf16-feature forward vs fp32: rel-L2 of the offsets {e16:.3e}")


# --------------------------------------------------------------------------------------
# config c5's mesh (2000 x 2000, V = 4 M, E = 24 M) on ONE GPU
# --------------------------------------------------------------------------------------
def test_c5_mesh_aggregation_properties_and_training_step():
    m = synth.torus_mesh(2000, 2000, masks=False)
    V = m.num_vertices
    ei = torch.from_numpy(m.edge_index).to(DEV)
    g = MeshGraph.from_edge_index(ei, V)
    assert V == 4_000_000 and ei.shape[1] == 24_000_000 and g.symmetric
    deg = torch.bincount(ei[0], minlength=V).float()
    for C, dtype in ((64, torch.float32), (256, torch.bfloat16)):
        f32 = dtype == torch.float32
        gen = torch.Generator(device=DEV).manual_seed(C)
        x = torch.randn(V, C, device=DEV, generator=gen).to(dtype)
        y = torch.randn(V, C, device=DEV, generator=gen).to(dtype)
        Lx, Ly = g.aggregate(x, torch.empty_like(x)), g.aggregate(y, torch.empty_like(y))
        s = deg.sqrt().view(-1, 1).expand(V, C).contiguous().to(dtype)              # eigenvector: L^ D^1/2 1 = -D^1/2 1
        Ls = g.aggregate(s, torch.empty_like(s))
        assert float((Ls.float() + s.float()).abs().max() / s.float().abs().max()) < (2e-6 if f32 else 2e-2)
        a = float((y.double() * Lx.double()).sum())                                   # symmetry <y, Lx> = <Ly, x>
        b = float((Ly.double() * x.double()).sum())
        assert abs(a - b) <= (1e-7 if f32 else 1e-4) * float(x.double().norm() * y.double().norm())
        fused = g.aggregate(x, torch.empty_like(x), alpha=2.0, X0=y, beta=-1.0)       # fused epilogue == separate ops
        sep = 2 * Lx.float() - y.float()
        assert float((fused.float() - sep).abs().max() / sep.abs().max()) < (1e-5 if f32 else 3e-2)
        rp, ci, dis = g.handle.arrays()                                               # spot rows against a direct gather
        for rr in torch.randint(0, V, (32,), generator=torch.Generator().manual_seed(2)).tolist():
            nb = ci[int(rp[rr]):int(rp[rr + 1])].long()
            want = -(dis[rr] * (dis[nb].view(-1, 1) * x[nb].float()).sum(0))
            assert float((Lx[rr].float() - want).abs().max()) <= (4e-5 if f32 else 3e-2) * max(float(want.abs().max()), 1e-3)
        del x, y, Lx, Ly, s, Ls, fused, sep
    torch.cuda.empty_cache()

    faces = torch.from_numpy(m.faces).to(DEV)
    target = torch.from_numpy(m.vs.astype(np.float32)).to(DEV)
    v_keep = torch.ones(V, 1, device=DEV)

    class D:
        z1 = torch.from_numpy(m.z1).to(DEV).requires_grad_(True)
        x_pos = torch.from_numpy(m.x_pos).to(DEV)
        edge_index = ei
    batch = train.MeshBatch(D, faces, target, train.face_normals(target, faces), v_keep,
                            torch.ones(faces.shape[0], 1, device=DEV), torch.ones(V, 1, device=DEV))
    losses = {}
    for dtype in (torch.float32, torch.bfloat16):
        net = SingleScaleGCN(DEV)
        GU.fill_state(net, seed=99)
        net.to(DEV)
        net.set_feature_dtype(dtype)
        tr = train.SGCNTrainer(net, batch)
        loss = tr.iteration_step()
        torch.cuda.synchronize()
        losses[dtype] = float(loss)
        assert np.isfinite(losses[dtype])
        assert all(bool(torch.isfinite(p.grad).all()) for p in net.parameters() if p.grad is not None)
        del net, tr
        torch.cuda.empty_cache()
    # same weights, same inputs: the bf16-feature loss sits within the bf16 forward error of the fp32 one
    assert abs(losses[torch.bfloat16] - losses[torch.float32]) < 0.1 * abs(losses[torch.float32]), losses


# --------------------------------------------------------------------------------------
# N > 1 with the real kernels: ranks share cuda:0, collectives over gloo (host-staged)
# --------------------------------------------------------------------------------------
def _run_selftest(world, backend, **env):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(29640 + world), os.path.join(ROOT, "tools", "dist_selftest.py")]
    return subprocess.run(cmd, env=_child_env(SEMIGCN_SELFTEST_BACKEND=backend, **env), capture_output=True, text=True,
                          timeout=900)


@pytest.mark.parametrize("world", [2, 4])
def test_partitioned_sgcn_and_mgcn_equal_single_rank_on_device(world):
    """tools/dist_selftest.py: partitioned SGCN (the 13 blocks phase by phase below the C ABI, sg_block_run, BatchNorm
    statistics in the pad rows of the halo exchange: 44 collectives, asserted inside the ranks) and MGCN
    (sg_graph_create_rect, sg_gather_rows, sg_bn_finalize_ranks, DistPool) == the plain single-device models: positions
    <= 1e-5, loss <= 2e-6 (asserted inside the ranks); and the phase path against the per-module path on the SAME partition:
    positions and loss <= 1e-6, reduced gradients <= 5e-3 (same arithmetic, other merge points)."""
    r = _run_selftest(world, "gloo", SEMIGCN_SELFTEST_CROSS="1")
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "dist_selftest OK" in r.stdout and "path=phases collectives=44" in r.stdout
    assert "phases vs per-module path on the same partition" in r.stdout
    # the MGCN: the runs of plain blocks of every stage phase by phase below the C ABI (27 of its 33 blocks: round 5)
    assert r.stdout.count("MGCN phases=True blocks below the C ABI [27, 27]") == world, r.stdout[-2000:]
    print(r.stdout[-1500:])


@pytest.mark.parametrize("path,n_coll", [("modules", 57), ("phases-sunk", 44)])
def test_partitioned_sgcn_other_paths_two_ranks(path, n_coll):
    """The same check on the per-module path of rounds 1-3 (halo exchange inside each convolution, an all-gather per
    BatchNorm: 57 collectives; bench.py's fallback) and on the phase path with the parameter gradients added into the
    .grad accumulators by the library (what the trainers run)."""
    r = _run_selftest(2, "gloo", SEMIGCN_SELFTEST_PATH=path, SEMIGCN_SELFTEST_SKIP_MGCN="1")
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "dist_selftest OK" in r.stdout and f"path={path} collectives={n_coll}" in r.stdout


def test_partitioned_bf16_features_two_ranks():
    """bf16 feature storage on a partition (what `bench.py --gpus N` runs): the statistics travel as fp32 words inside bf16
    pad rows (5 rows of C bf16 values hold 2 C + 1 floats), halo rows and gradient rows are packed 2 bytes per value.
    Against the single-device bf16 model: positions and loss close (asserted inside the ranks); gradients only loosely --
    two bf16 evaluations with different summation orders diverge."""
    r = _run_selftest(2, "gloo", SEMIGCN_SELFTEST_DTYPE="bf16", SEMIGCN_SELFTEST_SKIP_MGCN="1")
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "dist_selftest OK" in r.stdout and "path=phases collectives=44" in r.stdout


def test_eight_ranks_on_a_200k_vertex_mesh_equal_single_rank():
    """The 8-way partition the scaling run uses (Morton blocks, two-ring halos, 7 peers per rank), on a 500x400 mesh with
    the eight ranks sharing the one GPU over gloo: partitioned SGCN forward / loss / reduced gradients == the single-device
    model (asserted inside the ranks: positions <= 1e-5, loss <= 2e-6)."""
    r = _run_selftest(8, "gloo", SEMIGCN_SELFTEST_MESH="500x400", SEMIGCN_SELFTEST_SKIP_MGCN="1")
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "dist_selftest OK" in r.stdout and "path=phases collectives=44" in r.stdout
    assert r.stdout.count("pos rel-L2") == 8           # (occurrences, not lines: the ranks' prints can share a line)
    print(r.stdout[-1800:])


@pytest.mark.parametrize("path", ["phases", "modules"])
def test_one_rank_over_rccl_with_every_collective_issued(path):
    """The RCCL code path on a one-GPU box: ONE rank, backend "nccl", and SEMIGCN_DIST_FORCE_COLLECTIVES=1 so that the
    rank issues every collective of the partitioned iteration through the real library (communicator set-up, asynchronous
    all-to-all + stream wait, statistics all-gather, min/max / loss / gradient all-reduces) although each is the identity
    for a single rank; results must still equal the plain single-device model (asserted inside the rank).  Both paths:
    the blocks phase by phase (27 all-to-all with the statistics inside, 1 all-gather, 16 all-reduces = 44) and the
    per-module one (28 all-to-all, 13 all-gathers, 16 all-reduces = 57)."""
    r = _run_selftest(1, "nccl", SEMIGCN_DIST_FORCE_COLLECTIVES="1", SEMIGCN_SELFTEST_PATH=path, SEMIGCN_SELFTEST_PREFIX="0",
                      SEMIGCN_SELFTEST_SKIP_MGCN="1" if path == "phases" else "0")       # (the MGCN rides with "modules";
                                                                                         #  no prefix check: exact counts below)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "dist_selftest OK" in r.stdout
    counts = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("collectives ")][-1].split(" ", 1)[1])
    assert counts["backend"] == "nccl"
    if path == "phases":           # (the gradient all-reduce after the backward pass is counted too)
        assert (counts["all_to_all"], counts["all_gather"], counts["all_reduce"]) == (27, 1, 17), counts
    else:
        assert counts["all_to_all"] >= 28 and counts["all_gather"] >= 13 and counts["all_reduce"] >= 10
    print(r.stdout[-800:])


def test_one_rank_phase_path_collectives_below_the_c_abi_equal_the_torch_distributed_ones():
    """csrc/comm.hip on a one-GPU box: ONE rank over RCCL with every collective issued.  With the parameter gradients sunk
    into the .grad accumulators (what the trainers run) the forward AND the backward pass over the 13 blocks are ONE
    sg_part_run call each -- the library's own communicator, the 12 + 13 exchanges, 13 all-reduces and the statistics
    all-gather enqueued between the kernels -- and the rank's positions, loss and gradients are bit for bit those of the
    same run with the collectives issued one by one through torch.distributed (SEMIGCN_DIST_NATIVE=0); the collective
    count stays 44 + the gradient all-reduce."""
    out = {}
    for native in ("1", "0"):
        r = _run_selftest(1, "nccl", SEMIGCN_DIST_FORCE_COLLECTIVES="1", SEMIGCN_SELFTEST_PATH="phases-sunk",
                          SEMIGCN_SELFTEST_SKIP_MGCN="1", SEMIGCN_DIST_NATIVE=native)
        assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
        assert "dist_selftest OK" in r.stdout and "path=phases-sunk collectives=44" in r.stdout
        line = [ln for ln in r.stdout.splitlines() if " digest " in ln][-1]
        out[native] = (line.split(" digest ")[1].split()[0], line.split("native_part_runs=")[1].strip())
        counts = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("collectives ")][-1].split(" ", 1)[1])
        assert (counts["all_to_all"], counts["all_gather"], counts["all_reduce"]) == (27, 1, 17), counts
    assert out["1"][1] == "[1, 1]" and out["0"][1] == "[0, 0]", out
    assert out["1"][0] == out["0"][0], out


def test_one_rank_mgcn_phase_runs_over_rccl_and_per_module_mgcn_two_ranks():
    """The partitioned MGCN with its runs of plain blocks on dist.part_blocks (27 of 33 blocks; the pooled blocks module by
    module): one rank over RCCL with every collective issued -- the runs' forward and backward passes are sg_part_run calls
    with the library's own communicator, the input of every run fetched by one sg_halo_exchange -- against the single-device
    MGCN (asserted inside the rank: positions 1e-5, loss 5e-6); and the per-module MGCN of rounds 1-4 on two ranks (the
    supervisor's fallback for it)."""
    r = _run_selftest(1, "nccl", SEMIGCN_DIST_FORCE_COLLECTIVES="1", SEMIGCN_SELFTEST_PATH="phases-sunk", SEMIGCN_SELFTEST_PREFIX="0")
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "dist_selftest OK" in r.stdout and "MGCN phases=True blocks below the C ABI [27, 27]" in r.stdout
    r = _run_selftest(2, "gloo", SEMIGCN_SELFTEST_PATH="modules")
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "dist_selftest OK" in r.stdout and r.stdout.count("MGCN phases=False") == 2


def test_halo_exchange_entry_points_on_a_one_rank_communicator():
    """sg_comm_* / sg_halo_exchange / sg_part_run directly: a one-rank RCCL communicator whose rank sends rows to ITSELF (a
    send and a receive to one's own rank inside a group is a copy) drives the grouped send / receive path, the per-peer
    offsets, the in-place all-reduce and the all-gather through the real library."""
    import ctypes
    from semigcn_amd import capi
    dev = torch.device("cuda:0")
    assert capi.Comm.available()
    comm = capi.Comm(capi.Comm.unique_id(), 0, 1, [1000], [1000], dev)
    for dt, C in ((torch.float32, 64), (torch.bfloat16, 256)):
        send = torch.randn((1000, C), device=dev).to(dt)
        recv = torch.zeros_like(send)
        comm.halo_exchange(recv, send)
        assert torch.equal(recv, send)
    v = torch.arange(128, device=dev, dtype=torch.float32)
    w = v.clone()
    comm.all_reduce_(w)
    assert torch.equal(w, v)
    out = torch.zeros((1, 33), device=dev)
    comm.all_gather(out, v[:33].contiguous())
    assert torch.equal(out[0], v[:33])
    # the same three as one schedule
    send = torch.randn((1000, 32), device=dev)
    recv = torch.zeros_like(send)
    w.mul_(2.0)
    steps = (capi.sg_part_step * 3)()
    steps[0].kind, steps[0].n, steps[0].send, steps[0].recv = capi.STEP_EXCHANGE, 128, send.data_ptr(), recv.data_ptr()
    steps[1].kind, steps[1].n, steps[1].recv = capi.STEP_ALL_REDUCE, 128, w.data_ptr()
    steps[2].kind, steps[2].n, steps[2].send, steps[2].recv = capi.STEP_ALL_GATHER, 33 * 4, v.data_ptr(), out.data_ptr()
    out.zero_()
    capi.part_run(comm, steps, 3, torch.cuda.current_stream(dev).cuda_stream, dev)
    torch.cuda.synchronize()
    assert torch.equal(recv, send) and torch.equal(w, 2.0 * v) and torch.equal(out[0], v[:33])
    # a collective step without a communicator, an unknown step kind: refused, nothing enqueued
    with pytest.raises(capi.SemigcnLibraryError, match="no communicator"):
        capi.part_run(None, steps, 3, 0, dev)
    steps[0].kind = 9
    with pytest.raises(capi.SemigcnLibraryError, match="kind 9"):
        capi.part_run(comm, steps, 1, 0, dev)
    comm.close()


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs >= 2 GPUs for RCCL")
def test_partitioned_ranks_over_rccl():
    r = _run_selftest(2, "nccl")
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]


def test_bench_gpus_2_starts_its_own_ranks():
    """`python bench.py --gpus 2` without a launcher: refuses when fewer devices are visible; with the share-one-GPU
    self-test switch it starts two ranks itself and rank 0 prints ONE line with n_gpus = 2."""
    base = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
            "--mesh", "96x64", "--no-cpu-baseline", "--dtype", "fp32"]
    if torch.cuda.device_count() < 2:
        r = subprocess.run(base, env=_child_env(), capture_output=True, text=True, timeout=600)
        assert r.returncode != 0 and "refusing" in r.stderr and not r.stdout.strip()
    r = subprocess.run(base, env=_child_env(SEMIGCN_BENCH_SHARE_GPU="1"), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = r.stdout.splitlines()
    assert len(lines) == 1 and lines[0].startswith("{"), r.stdout[-2000:]      # ONE JSON line, nothing else on stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 2 and line["value"] > 0
    assert line["distributed"]["world_size"] == 2 and line["distributed"]["collectives_per_iteration"] > 0
    # the default: the blocks phase by phase below the C ABI, 44 collectives + the gradient all-reduce every 5th iteration
    d = line["distributed"]
    assert d["per_module_path"] is False and 44 <= d["collectives_per_iteration"] <= 45
    assert d["block_calls_per_iteration"] == [13.0, 13.0]        # measured: the 13 blocks ran below the C ABI in both directions
    assert "process group: ready" in d["startup_marks_s"] and "model built; warm-up" in d["startup_marks_s"]
    assert line["roofline"] is not None
    # a partitioned rank is always eager: its hipGraph replay modes of rounds 2-4 are gone
    r = subprocess.run(base + ["--graph"], env=_child_env(SEMIGCN_BENCH_SHARE_GPU="1"), capture_output=True, text=True, timeout=900)
    assert r.returncode != 0 and "one GPU only" in r.stderr
    # the cross-N parity check the lines carry: the loss of the very first iteration is a function of the workload alone -- the
    # two-rank line and the one-GPU line of the same mesh agree on it (fp32 features: to float32 summation order)
    solo = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--mesh", "96x64",
                           "--no-cpu-baseline", "--dtype", "fp32", "--single-dtype", "--no-second-order", "--no-distributed-estimate"],
                          env=_child_env(), capture_output=True, text=True, timeout=600)
    assert solo.returncode == 0, solo.stderr[-2000:]
    l1 = json.loads(solo.stdout.splitlines()[-1])["first_iteration_loss"]
    l2 = line["first_iteration_loss"]
    assert l1 is not None and l2 is not None and abs(l1 - l2) <= 2e-6 * abs(l1), (l1, l2)


def test_bench_supervisor_retries_same_path_then_per_module_after_stalls():
    """an N > 1 run whose attempts hang (injected: the last rank sleeps after its warm-up) is killed at the
    attempt's wall-clock limit and re-run from FRESH worker processes -- first on the SAME phase path, only then on the
    per-module path (--no-phases); the line says which attempt and path produced it, and the supervisor prints the stalled
    worker's start-up marks and its faulthandler dump.  A run that stalls in every attempt exits non-zero with the reasons."""
    base = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "5",
            "--mesh", "96x64", "--no-cpu-baseline", "--dtype", "bf16"]
    env = _child_env(SEMIGCN_BENCH_SHARE_GPU="1", SEMIGCN_BENCH_ATTEMPT_TIMEOUT="22")
    r = subprocess.run(base + ["--stall-after-warmup", "600", "--stall-attempts", "2"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])["distributed"]
    assert d["attempt"] == 3 and "within 22 s" in d["first_attempt_failure"] and "attempt 2" in d["first_attempt_failure"]
    assert d["per_module_path"] is True and 57 <= d["collectives_per_iteration"] <= 58 and d["block_calls_per_iteration"] == [0.0, 0.0]
    assert "starting a fresh worker on the SAME phase path" in r.stderr and "starting a fresh worker on the per-module path" in r.stderr
    # the post-mortem of the stalled worker: its marks up to the stall, and where its threads were shortly before the limit
    assert "marks of attempt 1" in r.stderr and "first warm-up iteration done" in r.stderr
    assert "traceback of attempt 1" in r.stderr and "in timed_run" in r.stderr
    env["SEMIGCN_BENCH_ATTEMPT_TIMEOUT"] = "9"
    r = subprocess.run(base + ["--stall-after-warmup", "-600"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode != 0 and "all 3 attempts failed" in r.stderr and not [l for l in r.stdout.splitlines() if l.startswith("{")]
