#!/usr/bin/env python3
"""Benchmark of the hot path: SGCN training iterations (forward + loss + backward, Adam every
5th iteration -- the loop of /root/reference/sgcn.py:118-147) on a synthetic closed manifold
mesh of V = 1 M vertices / E = 6 M directed edges (BASELINE.json's metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Prints ONE JSON line (rank 0).  `value` = training iterations per second of the whole job,
inputs resident in HBM; `roofline` = the dominant edge-aggregation kernel's ALGORITHMIC
bytes per launch / its mean launch duration measured with HIP events over the timed region;
`cpu_baseline` = the oracle's PyG-equivalent ATen path on this box's host cores on a bounded
sample (a reported baseline, not the target).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
# hipGraph replays (--graph, one GPU) are exact only with the runtime's graph "packet capture"
# off, and the runtime reads the flag once, at its initialisation: it has to be in the environment before the first HIP call
# (semigcn_amd.train.GRAPH_ENV).  It changes nothing for eager execution.
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")

import numpy as np
import torch

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (6.29 TB/s measured copy)
AGG_PER_ITER = 52              # SGCN: 26 aggregations forward + 26 backward


_T0 = time.perf_counter()


def log(msg: str):
    if os.environ.get("SEMIGCN_BENCH_MARK"):             # a supervised worker: every rank keeps its marks (see log_all)
        log_all(msg)
        if os.environ.get("SEMIGCN_BENCH_VERBOSE") == "1":
            return
    if int(os.environ.get("RANK", "0")) == 0:
        print(f"[bench +{time.perf_counter() - _T0:7.1f}s] {msg}", file=sys.stderr, flush=True)


_MARKS = []


def log_all(msg: str):
    """Start-up marks of EVERY rank: where an N-rank job spends its time before the first step.  Kept in memory (rank 0's go
    into the JSON line), appended to a per-rank file in the supervisor's marker directory as they happen (what a stalled
    attempt leaves behind for the post-mortem, next to its faulthandler dump), printed with SEMIGCN_BENCH_VERBOSE=1."""
    t = time.perf_counter() - _T0
    _MARKS.append((round(t, 2), msg))
    d = os.environ.get("SEMIGCN_BENCH_MARK")
    if d:
        try:
            with open(os.path.join(d, f"marks_attempt{os.environ.get('SEMIGCN_BENCH_ATTEMPT', '1')}_rank{os.environ.get('RANK', '0')}.txt"), "a") as f:
                f.write(f"{t:8.2f} {msg}\n")
        except OSError:
            pass
    if os.environ.get("SEMIGCN_BENCH_VERBOSE") == "1":
        print(f"[bench rank {os.environ.get('RANK', '0')} +{t:7.1f}s] {msg}", file=sys.stderr, flush=True)


def startup_marks():
    return {m: t for t, m in _MARKS}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--mesh", default="1000x1000", help="torus nu x nv of the whole job (the same mesh at every N: strong scaling)")
    ap.add_argument("--dtype", default="bf16", choices=["fp32", "bf16"],
                    help="feature storage of the measured workload: bf16 = what BASELINE configs[3] names (bf16 features, "
                         "fp32 accumulate, fp32 parameters and output); fp32 = the reference's own precision "
                         "(parity-checked to 1e-5).  At N=1 the other one is measured too and reported as "
                         "`fp32_features` / `bf16_features`")
    ap.add_argument("--single-dtype", action="store_true", help="skip the second run at the other precision")
    ap.add_argument("--permute", action="store_true", help="random vertex order (raw-scan like)")
    ap.add_argument("--partitioned", action="store_true",
                    help="diagnostic, --gpus 1 only: run the partitioned code path on ONE rank with every collective issued "
                         "through RCCL (what a rank of an N-rank job does; not a scaling point)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-distributed-estimate", action="store_true",
                    help="skip the `distributed_estimate` of the default N=1 run (two short child runs on the 1/8 mesh)")
    ap.add_argument("--cpu-sample", default="250x200", help="torus for the CPU baseline sample")
    ap.add_argument("--cpu-full", dest="cpu_full", action="store_true", default=True,
                    help="CPU baseline (default): time ONE oracle iteration at the full mesh size as SURVEY 8(d) / BASELINE.md "
                         "section 3 ask (~2 min and ~50 GB of host memory at V = 1 M; skipped with a note when the host has less "
                         "than 80 GB available), the bounded sample kept beside it")
    ap.add_argument("--no-cpu-full", dest="cpu_full", action="store_false", help="bounded CPU sample only (scaled linearly in V)")
    ap.add_argument("--mesh-recipe", default="survey", choices=["survey", "diagonal"],
                    help="survey (default): SURVEY 8(d)'s irregularity -- 0.15 * E_und random valid edge flips, valence 4..9; "
                         "diagonal: independent quad-diagonal flips p = 0.45, valence 4..8 (rounds 1-2)")
    ap.add_argument("--no-second-order", action="store_true",
                    help="skip the extra run in the other vertex order (SURVEY 8(d): report grid AND random order)")
    ap.add_argument("--worker", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--stall-after-warmup", type=float, default=0.0, help=argparse.SUPPRESS)
    ap.add_argument("--stall-attempts", type=int, default=1, help=argparse.SUPPRESS)
    ap.add_argument("--no-launch-timer", action="store_true")
    ap.add_argument("--graph", action="store_true", default=None,
                    help="ONE GPU only: replay each iteration from a hipGraph (pays off on launch-bound meshes <= ~200 K vertices "
                         "and MGCN); needs --warmup >= 4.  A partitioned rank is always eager (its two replay modes of rounds "
                         "2-4 were retired: DESIGN.md section 8)")
    ap.add_argument("--no-graph", dest="graph", action="store_false", help="eager execution (the default)")
    ap.add_argument("--no-planes", action="store_true",
                    help="A/B switch: the narrow bf16 layers keep their recurrence buffers as column blocks of one [V, K*C] buffer "
                         "instead of K dense [V, C] planes (include/semigcn.h, sg_block_planar)")
    ap.add_argument("--no-phases", action="store_true",
                    help="partitioned SGCN: every module on its own (halo exchange inside each convolution, an all-gather per "
                         "BatchNorm: 57 collectives) instead of the blocks run phase by phase below the C ABI with the "
                         "statistics riding in the halo exchange (dist.part_chain: 44 collectives; the default)")
    ap.add_argument("--model", default="sgcn", choices=["sgcn", "mgcn"],
                    help="mgcn: BASELINE config c3 (3 pool levels, hierarchy from meshprep.DeviceMesh); not the headline metric")
    return ap.parse_args()


def algorithmic_bytes(V: int, E: int, C: int, elem: int, n_epilogue: int) -> float:
    """read X once + write Y (+ one read per epilogue operand), int32 column
    index per edge, row pointers, deg^-1/2."""
    return (2 + n_epilogue) * V * C * elem + 4.0 * E + 4.0 * (V + 1) + 4.0 * V


def aggregation_kernel_name(C: int, esize: int, n_epi: int) -> str:
    """Which kernel sg_spmm dispatches to for this shape on a mesh graph (csrc/spmm.hip, launch_typed_one)."""
    if esize == 2 and C in (128, 256):        # pipelined LDS-tile kernel (graphs that carry tile records: every mesh graph here)
        return "spmm_ring"
    if esize == 4 and (C == 128 or (C == 256 and n_epi >= 1)):      # the same pipeline on float32 rows (256 channels: two half-row launches)
        return "spmm_ring_f32" if C == 128 else "spmm_ring_f32 x 2 (128-channel halves)"
    row_bytes = C * esize
    shared = esize == 4 and row_bytes >= 1024 and (row_bytes >= 2048 or n_epi == 0) and 16 < C // (16 // esize) <= 128
    return "spmm_shared" if shared else "spmm_rows"


def pmc_traffic(C: int, dtype_name: str, n_epi: int, esize: int):
    """HBM-side bytes per launch of the aggregation kernel variant that serves (C, dtype, n_epi), from
    the committed rocprofv3 PMC summary of this same workload (tools/pmc_summary.py; counters cannot be
    read from inside the process).  None when no summary matches."""
    vec = 16 // esize
    nvec = C // vec
    if C % vec:
        return None
    lanes = 1
    while lanes < min(nvec, 64):
        lanes *= 2
    per_lane = 1 if nvec <= 64 else (2 if nvec <= 128 else 4)
    import glob
    cands = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_pmc_traffic_{'fp32' if esize == 4 else 'bf16'}.json")))
    kernel = aggregation_kernel_name(C, esize, n_epi)
    try:
        path = cands[-1]                   # the latest round's table
        pmc_traffic.source = os.path.relpath(path, ROOT)
        table = json.load(open(path))
        pmc_traffic.commit = table.get("commit")
        # the table describes the kernel it was collected from: stale the moment csrc/spmm.hip changes
        import hashlib
        now = hashlib.sha256(open(os.path.join(ROOT, "semigcn_amd", "csrc", "spmm.hip"), "rb").read()).hexdigest()[:16]
        pmc_traffic.stale = table.get("spmm_hip_sha16") != now
        if pmc_traffic.stale:
            return None
        for k in table["kernels"]:
            if (k.get("kernel", "spmm_rows") == kernel and k["dtype"] == dtype_name and k["lanes_per_row"] == lanes
                    and k["vectors_per_lane"] == per_lane and k["epilogue_operands"] == n_epi):
                return k["traffic_bytes_per_launch"]
    except (OSError, KeyError, ValueError, IndexError):
        pass
    return None


def build_mesh_batch(mesh, device, n_masks: int):
    from semigcn_amd import synth, train
    V = mesh.num_vertices
    faces = torch.from_numpy(mesh.faces).to(device)
    target = torch.from_numpy(mesh.vs.astype(np.float32)).to(device)
    v_keep = torch.from_numpy(mesh.v_mask.astype(np.float32)).view(-1, 1).to(device)
    f_keep = (v_keep[faces[:, 0]] * v_keep[faces[:, 1]] * v_keep[faces[:, 2]])
    dm = torch.from_numpy(synth.make_dummy_masks(mesh.edge_index, V, dm_size=n_masks, k=4, p=0.014, seed=317)).to(device)

    class Data:
        z1 = torch.from_numpy(mesh.z1).to(device).requires_grad_(True)   # util/datamaker.py:71
        x_pos = torch.from_numpy(mesh.x_pos).to(device)
        edge_index = torch.from_numpy(mesh.edge_index).to(device)

    return train.MeshBatch(Data, faces, target, train.face_normals(target, faces), v_keep, f_keep, dm)


def make_mesh(nu: int, nv: int, recipe: str, permute: bool = False):
    from semigcn_amd import synth
    return synth.torus_mesh(nu, nv, permute=permute, edge_flips=0.15 if recipe == "survey" else None)


def cpu_baseline(sample: str, full_V: int, budget_s: float = 25.0, full: bool = False, recipe: str = "survey"):
    """Oracle (PyG-equivalent ATen ops on CPU: index_select -> multiply -> scatter_add_, normalisation per call, three
    linears, BatchNorm1d, LeakyReLU -- BASELINE.md section 3) SGCN iteration on the host cores of this box.
    Default: a BOUNDED sample (a smaller torus whose iterations fit `budget_s` seconds), scaled linearly in V (the
    path is O(V) at fixed valence).  `full=True` (--cpu-full): ONE timed iteration at the full V as BASELINE.md
    section 3 specifies (~2 min and ~50 GB of host memory at V = 1 M), the scaled figure kept beside it.
    Threads: the fastest count of a short probe over 8, 16, 32, .. os.cpu_count() threads (ATen's scatter/index
    kernels stop scaling early); the probe timings are reported in `sample`."""
    from oracle import models as OM            # cpu_baseline leg: the only oracle use in bench.py
    from semigcn_amd import synth
    ncpu = os.cpu_count() or 1
    torch.manual_seed(314)
    net = OM.SGCNOracle().train()

    def make(nu, nv):
        m = make_mesh(nu, nv, recipe)
        z1 = torch.from_numpy(m.z1).requires_grad_(True)
        x_pos, ei = torch.from_numpy(m.x_pos), torch.from_numpy(m.edge_index)
        faces = torch.from_numpy(m.faces)
        tgt = torch.from_numpy(m.vs.astype(np.float32))
        tfn = OM.compute_fn(tgt, faces)
        f_mask = m.v_mask[m.faces].all(1)
        dm = torch.from_numpy(synth.make_dummy_masks(m.edge_index, m.num_vertices, 1)).float()

        def it():
            net.zero_grad(set_to_none=True)
            pos = net(z1, x_pos, ei, dm)
            loss = OM.mask_pos_rec_loss(pos, tgt, m.v_mask) + 4.0 * OM.mask_norm_rec_loss(OM.compute_fn(pos, faces), tfn, f_mask)
            loss.backward()
        return m, it

    def timed(it, n=1):
        t0 = time.perf_counter()
        for _ in range(n):
            it()
        return (time.perf_counter() - t0) / n

    t_start = time.perf_counter()
    # thread-count probe on a 1.2 K-vertex mesh, counts in increasing order, stopping as soon as more threads are slower
    # (measured on the 2 x 128-thread GPU host: 0.69 s at 32 threads against 163 s at 256 for the 5 K-vertex probe --
    # ATen's index/scatter kernels drown in fork-join overhead; an unbounded all-core probe would cost minutes)
    m, it = make(40, 30)
    probes = {}
    for c in sorted({c for c in (8, 16, 32, 64, 128, ncpu) if c <= ncpu}):
        torch.set_num_threads(c)
        it()
        probes[c] = timed(it, 2)
        if probes[c] > 1.3 * min(probes.values()) or time.perf_counter() - t_start > 10.0:
            break
    cores = min(probes, key=probes.get)
    torch.set_num_threads(cores)
    probe_note = ", ".join(f"{probes[c] * 1e3:.0f} ms on {c} threads" for c in sorted(probes))
    m, it = make(100, 50)
    it()
    probes5k = timed(it, 1)
    nu, nv = map(int, sample.split("x"))
    per_vertex = probes5k / m.num_vertices
    while nu * nv * per_vertex * 3 > budget_s and nu * nv > 2 * m.num_vertices:
        nu, nv = max(nu * 3 // 4, 16), max(nv * 3 // 4, 16)
    if nu * nv > m.num_vertices:
        m, it = make(nu, nv)
        it()                                    # warm-up
    n, t0 = 0, time.perf_counter()
    while n < 2 or (time.perf_counter() - t_start < budget_s and n < 8):
        it()
        n += 1
    dt = (time.perf_counter() - t0) / n
    scaled = (1.0 / dt) * (m.num_vertices / full_V)
    out = {"value": scaled, "unit": "iter/s", "cores": cores, "kind": "port",
           "sample": f"{n} timed SGCN iterations (fwd+loss+bwd, fp32) of the oracle on a {m.nu}x{m.nv} torus "
                     f"(V={m.num_vertices}, E={m.num_edges}), {dt:.2f} s each on {cores} of {ncpu} logical CPUs "
                     f"(thread probe, 1.2K vertices, stopped once more threads ran slower: {probe_note}), scaled linearly in V to V={full_V}",
           "edges_aggregated_per_s": AGG_PER_ITER * m.num_edges / dt}
    if full:
        import psutil
        avail = psutil.virtual_memory().available
        if avail < 80e9 * (full_V / 1e6):
            out["sample"] += (f"; the full-size iteration was SKIPPED: {avail / 1e9:.0f} GB of host memory available, "
                              f"~{50 * full_V / 1e6:.0f} GB needed")
            full = False
    if full:
        import math
        side = int(round(math.sqrt(full_V)))
        mf, itf = make(side, full_V // side)
        dtf = timed(itf, 1)
        out["scaled_from_sample"] = scaled
        out["value"] = 1.0 / dtf
        out["edges_aggregated_per_s"] = AGG_PER_ITER * mf.num_edges / dtf
        out["sample"] = (f"ONE timed SGCN iteration (fwd+loss+bwd, fp32, no warm-up) of the oracle at the full size "
                         f"V={mf.num_vertices} E={mf.num_edges}: {dtf:.1f} s on {cores} of {ncpu} logical CPUs; "
                         f"beside it: " + out["sample"])
    return out


#: the partitioned code path is in use (N > 1 ranks, or --partitioned on one rank)
DIST_ON = False


def build_trainer(args, dtype, device, world, rank, mesh):
    """(trainer, workload string, edge-aggregations per iteration) for one feature dtype."""
    from semigcn_amd import synth, train
    from semigcn_amd.networks import SingleScaleGCN
    nu, nv = map(int, args.mesh.split("x"))
    if DIST_ON and args.model == "sgcn":
        from semigcn_amd import dist as sgdist
        job = sgdist.build_partitioned_job(nu, nv, world, rank, device, permute=args.permute, dtype=dtype, mesh=mesh, log=log_all,
                                           phases=not args.no_phases)
        return job.trainer, job.workload, AGG_PER_ITER * mesh.num_edges
    batch = build_mesh_batch(mesh, device, n_masks=5)
    torch.manual_seed(314)                               # sgcn.py:19-25,76
    if args.model == "mgcn":
        from semigcn_amd.meshnet import MGCN
        from semigcn_amd import meshprep
        smo = meshprep.DeviceMesh(mesh.x_pos, mesh.faces, device)       # hierarchy built on the device
        ini = meshprep.DeviceMesh(mesh.vs.astype(np.float32), mesh.faces, device)
        model = MGCN(device, smo, ini, torch.from_numpy(mesh.v_mask)).to(device)   # the reference's signature
        if dtype != torch.float32:
            model.set_feature_dtype(dtype)
        eis = model.edge_inds
        if DIST_ON:                                      # every level cut into `world` blocks (dist.partition_mgcn)
            from semigcn_amd import dist as sgdist
            trainer = sgdist.DistMGCNTrainer(model, sgdist.partition_mgcn(model, rank, world, phases=not args.no_phases), batch)
            trainer.phases = not args.no_phases          # (27 of the 33 blocks phase by phase below the C ABI; the pooled ones per module)
        else:
            trainer = train.MGCNTrainer(model, batch, capture=args.graph)
        agg_edges = 2 * sum(n * e.shape[1] for n, e in zip((6, 11, 11, 5), eis))
        trainer.levels = {int(p.shape[0]): int(e.shape[1]) for p, e in zip(model.poss_list, eis)}
    else:
        model = SingleScaleGCN(device).to(device)
        if dtype != torch.float32:
            model.set_feature_dtype(dtype)
        trainer = train.SGCNTrainer(model, batch, capture=args.graph)
        agg_edges = AGG_PER_ITER * mesh.num_edges
    workload = (f"{args.model.upper()} train iteration ({13 if args.model == 'sgcn' else 33} ChebConv K=3 + BN + LeakyReLU, "
                f"fwd+loss+bwd, Adam every 5th) on a closed torus mesh {nu}x{nv}: V={mesh.num_vertices} "
                f"E={mesh.num_edges} directed, {'random' if args.permute else 'grid'} vertex order, "
                f"{'fp32' if dtype == torch.float32 else 'bf16'} features"
                + (", iteration replayed from a hipGraph" if args.graph else "")
                + (f", every level vertex-partitioned into {world} blocks" if DIST_ON else ""))
    return trainer, workload, agg_edges


class TraceTimer:
    """The library's own per-launch event trace (sg_trace_*, include/semigcn.h) in the shape of capi.LaunchTimer: since a
    whole run of [ChebConv -> BatchNorm -> activation] blocks is ONE foreign call, no Python-side timer can bracket the
    aggregation / product launches any more; the library records the event pairs on the launching stream itself."""

    def __init__(self, kinds):
        from semigcn_amd import capi
        self.trace = capi.LaunchTrace(1 << 18, kinds)
        self.trace.__enter__()
        self.records = None

    def stop(self):
        """After a device synchronize: read the records and release the events."""
        self.records = self.trace.records()
        return self

    def results(self):
        out = {}
        for r in self.records:
            if r["ms"] < 0:
                continue
            if r["kind"] == "agg":
                key = (r["a"], r["dtype"], r["b"], r["c"])
            elif r["kind"] == "pool":
                key = ("pool", r["a"], r["dtype"], r["b"], r["c"])
            else:
                key = (r["kind"], r["a"], r["b"], r["c"], r["dtype"], r["engine"])
            out.setdefault(key, []).append(r["ms"])
        return out


def timed_run(trainer, args, device, world, with_timer: bool):
    """W untimed + exactly K timed iterations between barrier+synchronize brackets; max over ranks."""
    from semigcn_amd import capi

    def sync():
        torch.cuda.synchronize(device)
        if DIST_ON:
            import torch.distributed as dist
            dist.barrier()
            torch.cuda.synchronize(device)

    for i in range(args.warmup):
        loss = trainer.iteration_step()
        torch.cuda.synchronize(device)
        if i == 0:
            # the loss of the very first iteration (initial weights, mask 0, before any optimiser step) is a function of the
            # workload alone: the lines of an N = 1 and an N = 8 run of the same mesh must agree on it to the precision of the
            # features (~1e-3 relative with bf16 storage, ~1e-6 with fp32) -- a parity check across real devices for free
            timed_run.first_loss = float(loss.detach()) if torch.is_tensor(loss) else None
        log(f"warm-up iteration {i} done")
        if i == 0:
            log_all("first warm-up iteration done")
    timed_run.replay_check = None
    if getattr(trainer, "_graphed", None) is not None:
        # a replaying trainer is trusted only after ONE replayed iteration has reproduced an eager one on the same inputs
        # (train.replay_matches_eager; MGCN draws dropout masks, so its two passes cannot be compared)
        if args.model == "sgcn":
            from semigcn_amd import train as sgtrain
            ok = sgtrain.replay_matches_eager(trainer)
            torch.cuda.synchronize(device)
            timed_run.replay_check = "replayed loss == eager loss" if ok else "MISMATCH: the run continues eagerly"
            log(f"hipGraph replay check: {timed_run.replay_check}")
    # (--stall-after-warmup, the supervisor's self-test: > 0 stalls the first --stall-attempts attempts, < 0 every attempt)
    stall = args.stall_after_warmup
    if stall and (stall < 0 or int(os.environ.get("SEMIGCN_BENCH_ATTEMPT", "1")) <= args.stall_attempts) \
            and int(os.environ.get("RANK", "0")) == world - 1:
        log(f"TEST HOOK: the last rank stalls for {abs(stall):.0f} s")
        time.sleep(abs(stall))
    replays = False
    from semigcn_amd import functional as F_sg
    # the blocks run below the C ABI (functional.cheb_chain; dist.part_chain on a partition), timed by the library's own
    # trace; the per-module partitioned path issues every launch from Python and keeps the Python-side timers
    traced = with_timer and not replays and F_sg.blocks_enabled() and (not DIST_ON or getattr(trainer, "phases", False))
    timer = capi.LaunchTimer() if (with_timer and not replays and not traced) else None
    sync()
    capi.set_launch_timer(timer)
    # SGCN: the aggregation kernel's event pairs are recorded INSIDE the timed region (the contract; ~0.7 % of the 1 M iteration).
    # MGCN has 66 aggregations + 12 pool passes per iteration on meshes down to 10 K vertices, where two event records per launch
    # double the iteration: its launches are timed in a pass of their own right after the timed region (`measured_over` says so)
    # The same goes for the reference's own mesh sizes (5 K - 50 K vertices: an event pair per launch costs a 50 K iteration 10 %):
    # in the region from 200 K vertices on, where the pairs cost under 1 %
    nu_, nv_ = map(int, args.mesh.split("x"))
    in_region = args.model == "sgcn" and nu_ * nv_ >= 200_000
    if traced and in_region:
        timer = TraceTimer(("agg", "pool"))
    if DIST_ON:
        from semigcn_amd import dist as sgdist
        c0 = dict(sgdist.collective_counts)
    bc0 = list(F_sg.block_calls)
    nr0 = list(sgdist.native_runs) if DIST_ON else [0, 0]
    t0 = time.perf_counter()
    for _ in range(args.steps):
        trainer.iteration_step()
    sync()
    dt = time.perf_counter() - t0
    # blocks that really ran below the C ABI (forward, backward) per iteration: what `per_module_path` claims, measured
    timed_run.block_calls = [round((F_sg.block_calls[i] - bc0[i]) / args.steps, 2) for i in range(2)]
    # passes (forward, backward) per iteration that ran as ONE sg_part_run call with the collectives enqueued by the library
    timed_run.native_runs = [round((sgdist.native_runs[i] - nr0[i]) / args.steps, 2) for i in range(2)] if DIST_ON else None
    capi.set_launch_timer(None)
    if DIST_ON:      # (counted over the timed region only: the launch-timing pass below runs more iterations)
        timed_run.collectives = {k: (v - c0[k]) / args.steps for k, v in sgdist.collective_counts.items()}
    timed_run.timer_dt, timed_run.timer_steps = dt, args.steps
    timed_run.timer_where = "the timed region"
    if traced and in_region:
        timer.stop()
    elif traced:
        extra = max(1, min(args.steps, 5))
        timer = TraceTimer(("agg", "pool"))
        t1 = time.perf_counter()
        for _ in range(extra):
            trainer.iteration_step()
        sync()
        timed_run.timer_dt, timed_run.timer_steps = time.perf_counter() - t1, extra
        timed_run.timer_where = f"{extra} iterations after the timed region (HIP events on the launching stream)"
        timer.stop()
    timed_run.gemm_timer, timed_run.gemm_steps = None, 0
    if with_timer:
        from semigcn_amd import functional as F_sg
        # the dense products get their own short pass AFTER the timed region (an event pair per product inside it would
        # cost the headline ~0.7 %; the aggregation kernel's pairs stay inside it on one GPU, as the contract asks)
        timed_run.gemm_steps = max(1, min(args.steps, 5))
        if traced:
            timed_run.gemm_timer = TraceTimer(("nt", "tn"))
        else:
            timed_run.gemm_timer = capi.LaunchTimer()
            F_sg.set_gemm_timer(timed_run.gemm_timer)
        t1 = time.perf_counter()
        for _ in range(timed_run.gemm_steps):
            trainer.iteration_step()
        sync()
        timed_run.gemm_dt = time.perf_counter() - t1
        if traced:
            timed_run.gemm_timer.stop()
        F_sg.set_gemm_timer(None)
    if DIST_ON:
        import torch.distributed as dist
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        if dist.get_backend() == "gloo":
            t = t.cpu()
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return dt, timer


def summarize(dt, timer, args, dtype, mesh, world, agg_edges, trainer):
    """value / ms_per_step / per-kernel roofline numbers of one timed run."""
    V_total, E_total = mesh.num_vertices, mesh.num_edges
    value = args.steps / dt
    elem = 4 if dtype == torch.float32 else 2
    V_local, E_local = V_total // world, E_total // world
    nu, nv = map(int, args.mesh.split("x"))
    kernels, roof, pools = [], None, []
    # rows of a launch -> directed edges of the graph it ran on (SGCN: the one mesh; MGCN: one entry per level)
    levels = getattr(trainer, "levels", None) or {V_total: E_total}
    if timer is not None:
        results = timer.results()
        for key in [k for k in results if k[0] == "pool"]:
            _, C, dt_name, rows_w, rows_r = key
            times = results.pop(key)
            es = 4 if dt_name == "float32" else 2
            # SURVEY 8(d) per-unit figure for MeshPool / MeshUnpool (util/meshnet.py:9-27): both feature tensors once + one index
            # word per row on either side
            B = (rows_w + rows_r) * C * es + 4 * rows_w + 4 * rows_r
            mean_ms = float(np.mean(times))
            pools.append({"C": C, "dtype": dt_name, "rows_written": rows_w, "rows_read": rows_r, "launches": len(times),
                          "mean_ms": round(mean_ms, 4), "total_ms": round(float(np.sum(times)), 3),
                          "algorithmic_MB": round(B / 1e6, 3), "achieved_GBs": round(B / mean_ms / 1e6, 1),
                          "hbm_frac": round(B / mean_ms / 1e6 / HBM_PEAK_GBS, 4)})
        pools.sort(key=lambda k: -k["total_ms"])
        # launches that compute a rank's whole block (N > 1 also has row-subset launches for the overlap with the halo
        # exchange: interior rows / boundary + ring-1 rows; they are left out of the per-kernel roofline)
        full_rows = max(r for (_, _, _, r) in results) if results else 0
        if DIST_ON and results:
            g = getattr(getattr(trainer, "part", None), "graph", None)
            if g is not None:
                full_rows = g.n_own
            else:
                from collections import Counter
                full_rows = Counter(r for (_, _, _, r), t in results.items() for _ in t).most_common(1)[0][0]
        merged = {}
        for (C, dt_name, n_epi, rows), times in results.items():
            if DIST_ON:
                if rows == full_rows:
                    merged.setdefault((C, dt_name, n_epi, rows), []).extend(times)
            elif rows in levels:
                merged.setdefault((C, dt_name, n_epi, rows), []).extend(times)
        if DIST_ON:
            V_local, E_local = full_rows, int(E_total * full_rows / max(V_total, 1))
            levels = {full_rows: E_local}
        for (C, dt_name, n_epi, rows), times in sorted(merged.items(), key=lambda kv: -sum(kv[1])):
            mean_ms = float(np.mean(times))
            B = algorithmic_bytes(rows, levels[rows], C, 4 if dt_name == "float32" else 2, n_epi)
            kernels.append({"C": C, "dtype": dt_name, "epilogue_operands": n_epi,
                            **({"V": rows, "E": levels[rows]} if len(levels) > 1 else {}), "launches": len(times),
                            "mean_ms": round(mean_ms, 4), "total_ms": round(float(np.sum(times)), 3),
                            "algorithmic_MB": round(B / 1e6, 2), "achieved_GBs": round(B / mean_ms / 1e6, 1)})
        total_B = sum(k["algorithmic_MB"] * k["launches"] for k in kernels)
        total_t = sum(k["total_ms"] for k in kernels)
        dom = kernels[0]
        traffic = pmc_traffic(dom["C"], dom["dtype"], dom["epilogue_operands"], elem) if (
            not DIST_ON and (nu, nv) == (1000, 1000) and not args.permute and args.model == "sgcn") else None
        roof = {"bound": "hbm", "kernel": f"sg::{aggregation_kernel_name(dom['C'], elem, dom['epilogue_operands'])} C={dom['C']} "
                                          f"{dom['dtype']} (+{dom['epilogue_operands']} epilogue operands)",
                "achieved": dom["achieved_GBs"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(dom["achieved_GBs"] / HBM_PEAK_GBS, 4), "traffic": traffic,
                "traffic_note": "HBM-side bytes per launch of this kernel variant from the committed rocprofv3 PMC table of this "
                                "same workload (2*FETCH_SIZE+WRITE_SIZE KiB, separate passes; counters cannot be read from "
                                f"inside the process): {getattr(pmc_traffic, 'source', None)}, collected at commit "
                                f"{getattr(pmc_traffic, 'commit', None)}"
                                + ("; NULL because csrc/spmm.hip has changed since that table was collected"
                                   if getattr(pmc_traffic, "stale", False) else "")
                                + f"; algorithmic bytes = {round(dom['algorithmic_MB'] * 1e6)}",
                "all_aggregations_GBs": round(total_B / total_t, 1),
                "all_aggregations_frac": round(total_B / total_t / HBM_PEAK_GBS, 4),
                "aggregation_share_of_step": round(total_t / (getattr(timed_run, "timer_dt", dt) * 1e3), 4),
                "measured_over": getattr(timed_run, "timer_where", "the timed region")}
        if len(levels) > 1:
            roof["kernel"] += f" on the level with V={dom['V']} E={dom['E']}"
    dense = dense_products(getattr(timed_run, "gemm_timer", None), getattr(timed_run, "gemm_steps", 0),
                           getattr(timed_run, "gemm_dt", 0.0))
    return {"value": value, "ms_per_step": dt / args.steps * 1e3, "dense_products": dense,
            "dtype": "f32" if dtype == torch.float32 else "bf16 storage / f32 accumulate",
            "edges_aggregated_per_s": agg_edges * value, "optimizer_steps_per_s": value / 5.0,
            "mean_loss": float(trainer.loss_sum.item()) / max(trainer.iteration, 1),
            "first_iteration_loss": getattr(timed_run, "first_loss", None),
            "roofline": roof, "aggregation_kernels": kernels, "pool_kernels": pools or None}


MFMA_PEAK_TFLOPS = {"bfloat16": 2500.0, "float32": 157.3}     # MI355X_MICROARCH.md, dense peaks (bf16 MFMA; f32-input MFMA)


def _pipe(engine: str, dt_name: str):
    """(peak TFLOP/s of the pipe the launch runs on, instructions' flops issued per algorithmic flop, name of the pipe).
    Engine "split" computes a float32 product from SIX bf16 piece products on the bf16 matrix pipe (csrc/gemm_split.hip): it
    is priced against the 2.5 PF bf16 peak with the six products counted, never against the 157 TF float32-MFMA peak."""
    if engine == "split":
        return MFMA_PEAK_TFLOPS["bfloat16"], 6.0, "bf16 MFMA, six piece products per float32 product"
    if dt_name == "float32":
        return MFMA_PEAK_TFLOPS["float32"], 1.0, "float32 MFMA / vector ALUs"
    return MFMA_PEAK_TFLOPS["bfloat16"], 1.0, "bf16 MFMA"


def dense_products(timer, steps: int, dt: float):
    """The dense per-vertex feature x weight products (ChebConv's `lins`, util/networks.py:42,49, and their autograd) of
    the timed region, each launch timed with HIP events on its stream: per shape the MFMA roofline fraction
    (issued flops against the dense peak of the pipe the engine runs on: `_pipe`) next to the HBM one (operand + result
    bytes, the weight matrix L2-resident), and the totals -- which engine ran it: "mfma" = csrc/gemm_mfma*.hip,
    "split" = csrc/gemm_split.hip, "thin" = vector-ALU kernels, "blas" = hipBLASLt.  `frac` is the share of the products' time
    their issued flops would take at the peak of their own pipe (sum of issued / peak over the launches, divided by the time):
    it cannot exceed 1.  Measured over `steps` extra iterations right after the timed region (see timed_run)."""
    if timer is None or not timer.records or steps < 1:
        return None
    shapes, by_engine = [], {}
    tot_ms = tot_flops = tot_peak_ms = 0.0
    for (kind, M, N, K, dt_name, engine), times in sorted(timer.results().items(), key=lambda kv: -sum(kv[1])):
        elem = 4 if dt_name == "float32" else 2
        mean_ms = float(np.mean(times))
        flops = 2.0 * M * N * K
        peak, issued_per_flop, pipe = _pipe(engine, dt_name)
        # "nt": A [M,K] read, C [M,N] written;  "tn" (weight gradient): both [M,N] and [M,K] operands read, result tiny
        byts = (M * K + M * N) * elem
        row = {"kind": kind, "M": M, "N": N, "K": K, "dtype": dt_name, "engine": engine, "launches": len(times),
               "mean_ms": round(mean_ms, 4), "TFLOPs": round(flops / mean_ms / 1e9, 1),
               "mfma_frac": round(flops * issued_per_flop / mean_ms / 1e9 / peak, 4),
               "hbm_frac": round(byts / mean_ms / 1e6 / HBM_PEAK_GBS, 4)}
        if issued_per_flop != 1.0:
            row["issued_TFLOPs"] = round(flops * issued_per_flop / mean_ms / 1e9, 1)
        shapes.append(row)
        e = by_engine.setdefault(engine, {"ms_per_iteration": 0.0, "TFLOP_per_iteration": 0.0, "pipe": pipe, "peak": peak,
                                          "issued_per_flop": issued_per_flop})
        e["ms_per_iteration"] += sum(times) / steps
        e["TFLOP_per_iteration"] += flops * len(times) / steps / 1e12
        tot_ms += sum(times)
        tot_flops += flops * len(times)
        tot_peak_ms += flops * issued_per_flop * len(times) / peak / 1e9
    for e in by_engine.values():
        e["TFLOPs"] = round(e["TFLOP_per_iteration"] / e["ms_per_iteration"] * 1e3, 1)
        e["issued_TFLOPs"] = round(e["TFLOPs"] * e["issued_per_flop"], 1)
        e["frac_of_pipe_peak"] = round(e["issued_TFLOPs"] / e["peak"], 4)
        e["ms_per_iteration"], e["TFLOP_per_iteration"] = round(e["ms_per_iteration"], 3), round(e["TFLOP_per_iteration"], 3)
    dom = max(by_engine.values(), key=lambda e: e["ms_per_iteration"])
    return {"bound": "mfma", "peak": dom["peak"], "pipe": dom["pipe"], "unit": "TFLOP/s",
            "achieved": round(tot_peak_ms / tot_ms * dom["peak"], 1), "frac": round(tot_peak_ms / tot_ms, 4),
            "algorithmic_TFLOPs": round(tot_flops / tot_ms / 1e9, 1),
            "achieved_note": "issued flops per second in units of the dominant engine's pipe (`pipe`): the time-weighted mean of "
                             "every launch's fraction of its OWN pipe's peak, times `peak`; `algorithmic_TFLOPs` = 2 M N K per "
                             "launch over the same time (for engine `split` the float32-equivalent rate)",
            "ms_per_iteration": round(tot_ms / steps, 3), "share_of_step": round(tot_ms / (dt * 1e3), 4),
            "measured_over": f"{steps} iterations after the timed region (HIP events on the launching stream)",
            "by_engine": by_engine, "shapes": shapes[:12] + [s for s in shapes[12:] if s["engine"] not in ("mfma", "split")]}


def distributed_estimate(args, ms_per_step_n1: float):
    """What this code predicts for the 8-GPU strong-scaling point of the SAME mesh, measured on this one GPU (no box with more
    than one GPU has been available to this build; the driver's SCALE run has a number to be held against): one rank of an
    8-rank job owns V / 8 vertices, so a mesh of that size is run (a) on the partitioned code path with every collective
    issued through RCCL on a one-rank communicator (what a rank's host and GPU do, minus the wire) and (b) unpartitioned
    (what the GPU needs for a block of that size).  Child processes of this one -- nothing is exec'ed over a process that
    holds the GPU."""
    import math
    import subprocess
    nu, nv = map(int, args.mesh.split("x"))
    side = max(8, int(round(math.sqrt(nu * nv / 8.0))))
    base = [sys.executable, os.path.abspath(__file__), "--mesh", f"{side}x{side}", "--dtype", args.dtype, "--single-dtype",
            "--no-second-order", "--no-cpu-baseline", "--no-launch-timer", "--no-distributed-estimate", "--steps", "40", "--warmup", "8"]
    out = {"ranks": 8, "rank_mesh": f"{side}x{side}", "rank_vertices": side * side}

    def run(extra):
        try:
            r = subprocess.run(base + extra, capture_output=True, text=True, timeout=240)
            line = [l for l in r.stdout.splitlines() if l.startswith("{")]
            return json.loads(line[-1]) if line else {"error": (r.stderr or "")[-300:]}
        except Exception as e:          # the estimate must never cost the bench line
            return {"error": f"{type(e).__name__}: {e}"}
    # (a rank is bound by its host: the figure moves with whatever else the box's cores are doing, so the rank's run is
    # repeated and the faster of the two stands -- both are reported)
    parts = [run(["--partitioned"]) for _ in range(2)]
    solo = run(["--no-graph"])
    good = [p for p in parts if "ms_per_step" in p]
    part = min(good, key=lambda p: p["ms_per_step"]) if good else parts[0]
    if "ms_per_step" in part:
        out["rank_ms_per_step"] = round(part["ms_per_step"], 3)
        out["rank_ms_per_step_runs"] = [round(p["ms_per_step"], 3) for p in good]
        d = part.get("distributed") or {}
        out["collectives_per_iteration"] = d.get("collectives_per_iteration")
        out["rank_path"] = "per-module" if d.get("per_module_path") else "blocks phase by phase below the C ABI (eager)"
    else:
        out["rank_error"] = part.get("error")
    if "ms_per_step" in solo:
        out["block_ms_per_step_unpartitioned"] = round(solo["ms_per_step"], 3)
    else:
        out["block_error"] = solo.get("error")
    if "rank_ms_per_step" in out:
        out["predicted_speedup_8_gpus_before_wire_time"] = round(ms_per_step_n1 / out["rank_ms_per_step"], 2)
        out["note"] = (f"upper bound: the {out.get('collectives_per_iteration')} collectives per iteration cross no link here "
                       "(one-rank communicator); halo rows per rank and their bytes are in DESIGN.md section 5")
    return out


def free_port() -> int:
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def attempt_timeout_s(mesh: str) -> float:
    """Wall-clock limit of ONE attempt of an N-rank job, from its mesh size.  Measured with eight ranks sharing one GPU and one
    host (profiles/r04_shared_gpu_ranks.jsonl: every worker generates the mesh and its partition plan on the host, all at
    once): 2000 x 2000 (c5, V = 4 M) reaches its first warm-up iteration after 50 s and finishes after 57 s; 1000 x 1000 after
    7 s / 14 s.  The limit is ~5 x that: 120 s + 45 s per million vertices (c5: 300 s, the 1 M mesh: 165 s)."""
    env = os.environ.get("SEMIGCN_BENCH_ATTEMPT_TIMEOUT")
    if env:
        return float(env)
    try:
        nu, nv = map(int, mesh.split("x"))
    except ValueError:
        return 300.0
    return 120.0 + 45.0 * (nu * nv / 1.0e6)


ATTEMPT_TIMEOUT_S = 300.0          # set from the mesh in main() / spawn_ranks() (attempt_timeout_s)


def spawn_ranks(args) -> int:
    """`python bench.py --gpus N` without a launcher: start the N ranks ourselves.  Runs BEFORE this process has
    made any HIP call (torch.cuda.device_count() does not initialise the GPU on this image); the ranks are child
    processes of `python -m torch.distributed.run`, nothing is exec'ed over a process that touched the GPU.  Every rank
    supervises its own worker (``supervise_rank``: wall-clock limit per attempt, fresh retries: the phase path once more, then
    the per-module path); this launcher adds the outer limit -- three attempts plus slack -- after which the whole process group is killed and the
    reason reported."""
    import signal
    import subprocess
    global ATTEMPT_TIMEOUT_S
    ATTEMPT_TIMEOUT_S = attempt_timeout_s(args.mesh)
    shared = os.environ.get("SEMIGCN_BENCH_SHARE_GPU") == "1"
    n_dev = torch.cuda.device_count()
    if n_dev < args.gpus and not shared:
        print(f"bench.py: --gpus {args.gpus} but only {n_dev} HIP device(s) are visible; refusing to fall back to fewer "
              "ranks (SEMIGCN_BENCH_SHARE_GPU=1 runs the N-rank code path on one device over gloo as a self-test)",
              file=sys.stderr)
        return 2
    port = free_port()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "8")
    limit = 3 * ATTEMPT_TIMEOUT_S + 180
    log(f"--gpus {args.gpus} without a launcher: starting {args.gpus} ranks with torch.distributed.run on port {port} "
        f"(outer limit {limit:.0f} s)")
    proc = subprocess.Popen(cmd, env=env, start_new_session=True)
    try:
        return proc.wait(timeout=limit)
    except subprocess.TimeoutExpired:
        print(f"bench.py: the {args.gpus}-rank job did not finish within {limit:.0f} s (three attempts of "
              f"{ATTEMPT_TIMEOUT_S:.0f} s each); killing its process group", file=sys.stderr, flush=True)
        try:
            os.killpg(proc.pid, signal.SIGKILL)
        except ProcessLookupError:
            pass
        proc.wait()
        return 124


def _terminate(proc) -> None:
    import subprocess
    if proc.poll() is None:
        proc.terminate()
        try:
            proc.wait(timeout=15)
        except subprocess.TimeoutExpired:
            proc.kill()
            proc.wait()


def supervise_rank(args) -> int:
    """One rank of an N > 1 job as its launcher (the driver's `torch.distributed.run`, or ``spawn_ranks``) started it.  This
    process never touches the GPU: it runs the actual work in a CHILD (`bench.py ... --worker`, same rank environment) and
    waits for it with a wall-clock limit.  A partitioned SGCN rank runs its blocks phase by phase below the C ABI
    (dist.part_chain), a path exercised between real devices only by the driver's own SCALE run.  If ANY rank's worker fails
    or overruns the limit, every supervisor kills its worker (they agree through marker files in a directory named after the
    job's rendezvous port -- one node) and starts a FRESH one, on a fresh store prefix:
      attempt 2: the SAME phase path again (a start-up hiccup must not turn the first real SCALE number into a measurement of
                 the slow path), its collectives issued one by one through torch.distributed instead of by the library's own
                 communicator (SEMIGCN_DIST_NATIVE=0: csrc/comm.hip has met real peers only in the driver's SCALE run);
      attempt 3: the per-module path of rounds 1-3 (`--no-phases`: every collective issued from Python, 57 per iteration).
    A third failure exits non-zero with the reasons.  Every worker leaves its start-up marks and, if it is still alive shortly
    before the limit, a faulthandler dump of all its threads in the marker directory; the supervisor prints both for its rank
    when an attempt fails.  Nothing is ever exec'ed over a process that has initialised the GPU."""
    import glob
    import subprocess
    import tempfile
    global ATTEMPT_TIMEOUT_S
    ATTEMPT_TIMEOUT_S = attempt_timeout_s(args.mesh)
    rank = int(os.environ.get("RANK", "0"))
    port = os.environ.get("MASTER_PORT", "0")
    mark = os.path.join(tempfile.gettempdir(), f"semigcn_bench_{os.getuid()}_{port}_{os.environ.get('TORCHELASTIC_RUN_ID', 'x')}")
    os.makedirs(mark, exist_ok=True)
    if rank == 0:
        for f in glob.glob(os.path.join(mark, "*")):
            os.remove(f)
    reasons = []
    n_attempts = 3
    for attempt in range(1, n_attempts + 1):
        env = dict(os.environ)
        env["SEMIGCN_BENCH_ATTEMPT"], env["SEMIGCN_BENCH_MARK"] = str(attempt), mark
        # the library's own communicator meets real peers for the first time in attempt 1: if its known-answer collectives hang
        # (rather than disagree) the worker ends itself after 45 s (dist._HangGuard, exit code 86) instead of sitting out the
        # attempt's limit, and attempt 2 runs with the collectives on torch.distributed.  The guard covers the three native
        # collectives and a device synchronise only: torch.distributed's reference collectives run before it.
        env.setdefault("SEMIGCN_DIST_KNOWN_ANSWER_TIMEOUT", "45")
        if attempt > 1:
            env["SEMIGCN_BENCH_FIRST_FAILURE"] = " | ".join(reasons)[:600]
            # a store prefix / port of its own: the earlier attempts' keys (and, without an agent store, their listening socket)
            # must not be met again
            env["TORCHELASTIC_RESTART_COUNT"] = str(int(env.get("TORCHELASTIC_RESTART_COUNT", "0")) + attempt - 1)
            if env.get("TORCHELASTIC_USE_AGENT_STORE") != "True":
                env["MASTER_PORT"] = str(int(port) + attempt - 1)
            env["SEMIGCN_DIST_NATIVE"] = "0"
        per_module = attempt == n_attempts
        cmd = [sys.executable, os.path.abspath(__file__)] + sys.argv[1:] + ["--worker"] + (["--no-phases"] if per_module else [])
        path = "per-module path" if (per_module or args.no_phases) else "blocks phase by phase"
        t0 = time.perf_counter()
        proc = subprocess.Popen(cmd, env=env)
        why = None
        while True:
            try:
                rc = proc.wait(timeout=1.0)
                if rc != 0 and not os.path.exists(os.path.join(mark, f"done{attempt}")):
                    why = f"rank {rank}'s worker exited with code {rc} in attempt {attempt} ({path})"
                break
            except subprocess.TimeoutExpired:
                pass
            if os.path.exists(os.path.join(mark, f"failed{attempt}")):
                why = open(os.path.join(mark, f"failed{attempt}")).read() or "another rank failed"
                _terminate(proc)
                break
            if time.perf_counter() - t0 > ATTEMPT_TIMEOUT_S:
                why = f"rank {rank}'s worker did not finish attempt {attempt} within {ATTEMPT_TIMEOUT_S:.0f} s ({path})"
                _terminate(proc)
                break
        if why is None:
            return 0
        try:                                   # first writer wins; the others read it
            with open(os.path.join(mark, f"failed{attempt}"), "x") as f:
                f.write(why)
        except FileExistsError:
            pass
        reasons.append(why)
        nxt = ("" if attempt == n_attempts else
               "; starting a fresh worker on " + ("the SAME phase path (collectives through torch.distributed)" if attempt + 1 < n_attempts else "the per-module path"))
        print(f"bench.py supervisor (rank {rank}): {why}{nxt}", file=sys.stderr, flush=True)
        for kind in ("marks", "traceback"):    # the post-mortem of THIS rank's worker
            fn = os.path.join(mark, f"{kind}_attempt{attempt}_rank{rank}.txt")
            if os.path.exists(fn) and os.path.getsize(fn):
                print(f"bench.py supervisor (rank {rank}): {kind} of attempt {attempt}:\n" + open(fn).read()[-3000:], file=sys.stderr, flush=True)
        # every supervisor must have seen the marker and killed its worker before the fresh set meets
        time.sleep(3.0)
    print(f"bench.py: all {n_attempts} attempts failed on rank {rank}: {' | '.join(reasons)}", file=sys.stderr, flush=True)
    return 3


#: file descriptor of the process's real standard output (see main)
REAL_STDOUT = 1


def main():
    global REAL_STDOUT
    args = parse()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(spawn_ranks(args))
    if args.gpus != int(os.environ.get("WORLD_SIZE", "1")):
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={os.environ.get('WORLD_SIZE', '1')}: one rank per GPU, launched as "
                         f"`python -m torch.distributed.run --nproc-per-node {args.gpus} bench.py --gpus {args.gpus} ...`")
    if args.gpus > 1 and not args.worker:
        raise SystemExit(supervise_rank(args))
    # ONE JSON line on standard output, nothing else: RCCL prints a version banner and gloo its connection notes to fd 1 from
    # C code, so fd 1 is pointed at standard error for the run and the line is written to the saved descriptor at the end
    sys.stdout.flush()
    REAL_STDOUT = os.dup(1)
    os.dup2(2, 1)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: one rank per GPU, launched as "
                         f"`python -m torch.distributed.run --nproc-per-node {args.gpus} bench.py --gpus {args.gpus} ...`")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (there is no CPU path)")
    # self-test mode for boxes with ONE GPU: every rank uses cuda:0 and talks over gloo (host-staged);
    # it exercises this N > 1 code path, its numbers mean nothing
    shared = os.environ.get("SEMIGCN_BENCH_SHARE_GPU") == "1"
    if shared:
        local = 0
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    global DIST_ON
    DIST_ON = world > 1 or args.partitioned
    if args.partitioned and world == 1:
        # diagnostic: ONE rank runs the partitioned code path -- its own [owned | halo] operators, mesh-wide BatchNorm,
        # and every collective issued through RCCL although each is the identity (what a rank of an N-rank job does on
        # the host, measurable on a one-GPU box); its number is not a scaling point
        from semigcn_amd import dist as sgdist
        sgdist.FORCE_COLLECTIVES = True
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(free_port()))
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if os.environ.get("SEMIGCN_BENCH_MARK"):
        # a supervised worker: if it is still running shortly before its supervisor's limit, every thread's Python stack goes
        # into the marker directory -- what a stalled start-up was doing (the 1-in-8 stall of round 4 left nothing behind)
        import faulthandler
        try:
            tb = open(os.path.join(os.environ["SEMIGCN_BENCH_MARK"], f"traceback_attempt{os.environ.get('SEMIGCN_BENCH_ATTEMPT', '1')}"
                                                                      f"_rank{rank}.txt"), "w")
            faulthandler.enable(file=tb, all_threads=True)
            faulthandler.dump_traceback_later(max(10.0, attempt_timeout_s(args.mesh) - 8.0), repeat=False, file=tb, exit=False)
        except OSError:
            pass
    log_all("worker started")
    if DIST_ON:
        import torch.distributed as dist
        store = None
        attempt = int(os.environ.get("SEMIGCN_BENCH_ATTEMPT", "1"))
        if attempt > 1:
            # a supervisor's second set of workers (supervise_rank): the launcher's store still holds the first set's
            # rendezvous keys (this torch adds no per-attempt prefix), so the second set meets under a prefix of its own
            import datetime
            agent = os.environ.get("TORCHELASTIC_USE_AGENT_STORE") == "True"
            base = dist.TCPStore(os.environ["MASTER_ADDR"], int(os.environ["MASTER_PORT"]), world,
                                 is_master=(rank == 0 and not agent), timeout=datetime.timedelta(seconds=300),
                                 wait_for_workers=False)
            store = dist.PrefixStore(f"semigcn/attempt{attempt}", base)
        log_all("process group: init")
        if shared:
            dist.init_process_group("gloo", store=store, rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", store=store, device_id=device, rank=rank, world_size=world)
        log_all("process group: ready")

    from semigcn_amd import capi, synth
    capi.load()
    if args.no_planes:
        from semigcn_amd import functional as F_sg
        F_sg.USE_PLANES = False
        capi.tuning_set(capi.TUNE_BLOCK_PLANES, 0)
    log("library loaded")
    nu, nv = map(int, args.mesh.split("x"))
    mesh = make_mesh(nu, nv, args.mesh_recipe, permute=args.permute)
    log(f"mesh generated V={mesh.num_vertices} E={mesh.num_edges}")
    dtypes = {"fp32": torch.float32, "bf16": torch.bfloat16}
    if args.graph is None:
        # default: EAGER everywhere.  A partitioned rank is ALWAYS eager: the blocks run phase by phase below the C ABI
        # (dist.part_chain, 44 collectives); its hipGraph replay modes of rounds 2-4 were retired.
        args.graph = False
        # one GPU: EAGER at every size.  The reference's own mesh sizes (c1: 5 K, c2 / c3: 50 K vertices) used to be bound by
        # ~300 launches with their Python glue and were replayed from a hipGraph by default (rounds 2-3); with runs of
        # blocks below the C ABI (sg_block_chain_*) the eager iteration is within ~1.2x of the replayed one, needs no
        # environment flag, and MGCN's dropout draws stay fresh.  --graph still replays (SGCN checked against an eager pass).
    if args.graph and DIST_ON:
        raise SystemExit("--graph: one GPU only (a partitioned rank runs eagerly; its replay modes were retired in round 5)")
    if args.graph and args.warmup < 4:
        raise SystemExit("--graph: --warmup must be >= 4 (3 eager iterations + the capture)")
    # byte accounting assumes the finest mesh only; inside a whole-iteration hipGraph launches cannot be timed
    with_timer = not args.no_launch_timer and not args.graph and (args.model == "sgcn" or not DIST_ON)

    trainer, workload, agg_edges = build_trainer(args, dtypes[args.dtype], device, world, rank, mesh)
    log("model built; warm-up")
    dt, timer = timed_run(trainer, args, device, world, with_timer)
    main_res = summarize(dt, timer, args, dtypes[args.dtype], mesh, world, agg_edges, trainer)
    log(f"timed region done: {main_res['ms_per_step']:.2f} ms/iteration ({args.dtype})")

    other = None
    if not DIST_ON and args.model == "sgcn" and not args.single_dtype:
        # the same workload at the other feature precision, for the record (never `value`)
        del trainer
        torch.cuda.empty_cache()
        od = "fp32" if args.dtype == "bf16" else "bf16"
        tr2, _, agg2 = build_trainer(args, dtypes[od], device, world, rank, mesh)
        dt2, timer2 = timed_run(tr2, args, device, world, with_timer)
        other = summarize(dt2, timer2, args, dtypes[od], mesh, world, agg2, tr2)
        # (the reference's own precision keeps its per-shape aggregation table: the narrow float32 shapes have a fraction in the
        #  line too; the per-shape product table stays with the headline precision only)
        other.pop("pool_kernels", None)
        if other.get("dense_products"):
            other["dense_products"].pop("shapes")
        log(f"second precision done: {other['ms_per_step']:.2f} ms/iteration ({od})")
        del tr2
        torch.cuda.empty_cache()

    order2 = None
    if not DIST_ON and args.model == "sgcn" and not args.no_second_order and not args.single_dtype:
        # report BOTH vertex orders -- the same mesh renumbered at random (a raw scan) or, with --permute, in grid
        # order; the model's Morton processing order (networks.py) is what makes the two agree
        a2 = argparse.Namespace(**vars(args))
        a2.permute = not args.permute
        mesh2 = make_mesh(nu, nv, args.mesh_recipe, permute=a2.permute)
        tr3, _, agg3 = build_trainer(a2, dtypes[args.dtype], device, world, rank, mesh2)
        dt3, timer3 = timed_run(tr3, a2, device, world, with_timer)
        r3 = summarize(dt3, timer3, a2, dtypes[args.dtype], mesh2, world, agg3, tr3)
        order2 = {"vertex_order": "random" if a2.permute else "grid", "value": r3["value"], "ms_per_step": r3["ms_per_step"],
                  "edges_aggregated_per_s": r3["edges_aggregated_per_s"], "dtype": r3["dtype"],
                  "dominant_aggregation": None if r3["roofline"] is None else
                  {k: r3["roofline"][k] for k in ("kernel", "achieved", "frac", "all_aggregations_frac")}}
        log(f"other vertex order done: {r3['ms_per_step']:.2f} ms/iteration ({order2['vertex_order']})")
        del tr3, mesh2
        torch.cuda.empty_cache()

    if rank == 0:
        line = {
            "metric": "GCN train iters/sec + edges-aggregated/sec, 1M-vert mesh",
            "value": main_res["value"], "unit": "iter/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": main_res["ms_per_step"], "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": main_res["dtype"], "data": "synthetic",
            "config": {"workload": workload, "V": mesh.num_vertices, "E": mesh.num_edges,
                       "baseline_config": (("BASELINE configs[3] (SGCN, 1M-vertex mesh, bf16 features: bf16 storage of the "
                                            "per-vertex features, fp32 accumulation, fp32 parameters / loss / output); the "
                                            "same workload at the reference's fp32 precision is in `fp32_features`")
                                           if args.dtype == "bf16" else
                                           ("BASELINE configs[3]'s mesh (SGCN, 1M vertices) at the reference's fp32 "
                                            "precision; the bf16-feature variant named there is in `bf16_features`")) if (
                           args.model == "sgcn" and (nu, nv) == (1000, 1000)) else None,
                       "mesh_recipe": ("SURVEY 8(d): closed torus grid, 0.15 * E_und random valid edge flips (valence 4..9), "
                                       if args.mesh_recipe == "survey" else
                                       "closed torus grid, quad diagonals flipped independently with p=0.45 (valence 4..8), ")
                                      + "jitter N(0,0.05^2), x_pos = positions - z1, seeds 314-317 (semigcn_amd/synth.py)",
                       "vertex_order": "random" if args.permute else "grid",
                       "aggregations_per_iteration": AGG_PER_ITER if args.model == "sgcn" else 66},
            "edges_aggregated_per_s": main_res["edges_aggregated_per_s"],
            "optimizer_steps_per_s": main_res["optimizer_steps_per_s"], "mean_loss": main_res["mean_loss"],
            "first_iteration_loss": main_res.get("first_iteration_loss"),
            "roofline": main_res["roofline"], "aggregation_kernels": main_res["aggregation_kernels"],
            "dense_products": main_res["dense_products"],
        }
        if main_res.get("pool_kernels"):
            line["pool_kernels"] = main_res["pool_kernels"]
        if other is not None:
            line["fp32_features" if other["dtype"] == "f32" else "bf16_features"] = other
        if order2 is not None:
            line["other_vertex_order"] = order2
        if getattr(timed_run, "replay_check", None):
            line["hip_graph_replay_check"] = timed_run.replay_check
        if DIST_ON:
            import torch.distributed as dist
            g = trainer.part.graph if hasattr(trainer, "part") and hasattr(trainer.part, "graph") else None
            coll = getattr(timed_run, "collectives", {})
            line["distributed"] = {"world_size": dist.get_world_size(), "backend": dist.get_backend(),
                                   "devices_visible": torch.cuda.device_count(), "ranks_share_one_gpu": shared,
                                   "single_rank_diagnostic": bool(args.partitioned and world == 1),
                                   "attempt": int(os.environ.get("SEMIGCN_BENCH_ATTEMPT", "1")),
                                   "first_attempt_failure": os.environ.get("SEMIGCN_BENCH_FIRST_FAILURE"),
                                   "per_module_path": not getattr(trainer, "phases", False),
                                   "block_calls_per_iteration": getattr(timed_run, "block_calls", None),
                                   "part_run_calls_per_iteration": getattr(timed_run, "native_runs", None),
                                   "startup_marks_s": startup_marks(),
                                   "collectives_per_iteration": round(sum(coll.values()), 1), "collectives_by_kind": coll,
                                   "rank0_owned_rows": None if g is None else g.n_own,
                                   "rank0_halo_rows": None if g is None else g.n_halo}
        if (not DIST_ON and world == 1 and args.model == "sgcn" and not args.no_distributed_estimate and not args.no_cpu_baseline
                and mesh.num_vertices >= 500_000):
            log("distributed estimate: two short child runs on the 1/8 mesh")
            line["distributed_estimate"] = distributed_estimate(args, main_res["ms_per_step"])
        line["cpu_baseline"] = cpu_baseline(args.cpu_sample, mesh.num_vertices, full=args.cpu_full, recipe=args.mesh_recipe) if (
            not DIST_ON and not args.no_cpu_baseline) else None
        sys.stdout.flush()
        os.write(REAL_STDOUT, (json.dumps(line) + "\n").encode())
        if os.environ.get("SEMIGCN_BENCH_MARK"):          # the line is out: what follows cannot fail the attempt any more
            open(os.path.join(os.environ["SEMIGCN_BENCH_MARK"], f"done{os.environ.get('SEMIGCN_BENCH_ATTEMPT', '1')}"), "w").close()
    if DIST_ON:
        import torch.distributed as dist
        try:
            dist.barrier()
            dist.destroy_process_group()
        except Exception as e:                             # tear-down trouble after the result is out is not a failed run
            log(f"process-group tear-down: {type(e).__name__}: {e}")


if __name__ == "__main__":
    main()
