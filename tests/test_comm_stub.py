"""The library's own collectives (csrc/comm.hip: sg_halo_exchange, sg_comm_all_reduce_f32, sg_comm_all_gather, sg_part_run)
with MORE THAN ONE RANK, on a box without a second GPU: `sg_comm_test_stub(1)` replaces RCCL by an in-process stand-in
(include/semigcn.h), every "rank" is a communicator of this process and its buffers are host memory.  What a one-rank
communicator on one GPU cannot show -- a wrong peer offset in exchange()'s walk over the per-peer row counts -- shows here:
the rows every rank receives are compared with what torch.distributed's `all_to_all_single` moves for the same split lists
(dist.FoldedLayout.send_splits / recv_splits), and the logged (pointer, bytes, peer) of every ncclSend / ncclRecv with the
prefix sums of those lists.  SURVEY section 8(b) `sg_halo_exchange`, section 8(e).  No GPU, no RCCL call."""
import ctypes
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from semigcn_amd import capi  # noqa: E402
from test_dist import _folded_layouts  # noqa: E402

c_int64 = ctypes.c_int64


@pytest.fixture
def stub():
    lib = capi.load()
    assert lib.sg_comm_test_stub(1) == 0
    comms = []
    try:
        yield lib, comms
    finally:
        for h in comms:
            lib.sg_comm_destroy(h)
        lib.sg_comm_test_stub(0)


def _make_comms(lib, comms, world, send_rows, recv_rows):
    uid = ctypes.create_string_buffer(128)
    assert lib.sg_comm_unique_id(uid) == 0
    out = []
    for r in range(world):
        h = ctypes.c_void_p()
        rc = lib.sg_comm_create(uid, r, world, (c_int64 * world)(*send_rows[r]), (c_int64 * world)(*recv_rows[r]), ctypes.byref(h))
        assert rc == 0, lib.sg_last_error()
        comms.append(h)
        out.append(h)
    return out


def _log(lib):
    n = int(lib.sg_comm_test_log(None, 0, None))
    buf = (c_int64 * (5 * max(n, 1)))()
    state = (c_int64 * 3)()
    lib.sg_comm_test_log(buf, n, state)
    recs = np.frombuffer(buf, dtype=np.int64)[:5 * n].reshape(n, 5).copy()
    return recs, [int(v) for v in state]


def _all_to_all_single(sends, send_splits, recv_splits):
    """What torch.distributed.all_to_all_single(out_r, in_r, recv_splits[r], send_splits[r]) leaves in out_r, for every r."""
    world = len(sends)
    outs = []
    for r in range(world):
        parts = []
        for q in range(world):
            at = sum(send_splits[q][:r])
            parts.append(sends[q][at:at + send_splits[q][r]])
            assert send_splits[q][r] == recv_splits[r][q]
        outs.append(np.concatenate(parts) if parts else np.zeros((0,) + sends[r].shape[1:], sends[r].dtype))
    return outs


@pytest.mark.parametrize("world", [2, 3, 8])
@pytest.mark.parametrize("C,dtype", [(6, np.float32), (16, np.uint16)])
def test_halo_exchange_moves_what_all_to_all_single_moves(stub, world, C, dtype):
    lib, comms = stub
    sgdist, ei, V, gs, lays = _folded_layouts(world, 48, 32)
    send_rows = [lay.send_splits for lay in lays]
    recv_rows = [lay.recv_splits for lay in lays]
    hs = _make_comms(lib, comms, world, send_rows, recv_rows)
    rng = np.random.default_rng(world * 100 + C)
    # row i of rank r's send buffer carries (r, i): any misplaced byte shows
    sends, recvs = [], []
    for r, lay in enumerate(lays):
        s = rng.integers(1, 60000, size=(lay.n_send, C)).astype(dtype)
        s[:, 0] = r
        s[:, 1] = np.arange(lay.n_send) % 60000
        sends.append(np.ascontiguousarray(s))
        recvs.append(np.full((lay.n_ext - lay.n_own, C), 7, dtype=dtype))
    row_bytes = C * np.dtype(dtype).itemsize
    for r in range(world):                                # every rank posts its exchange, one after the other
        rc = lib.sg_halo_exchange(hs[r], sends[r].ctypes.data, recvs[r].ctypes.data, row_bytes, None)
        assert rc == 0, lib.sg_last_error()
    want = _all_to_all_single(sends, send_rows, recv_rows)
    for r in range(world):
        assert np.array_equal(recvs[r], want[r]), f"rank {r} of {world}"
        # and they are the rows the layout says they are: segment of peer q = q's rows send_index, then its pad rows
        lay = lays[r]
        at = 0
        for q in range(world):
            n = lay.recv_splits[q]
            if n:
                assert (recvs[r][at:at + n, 0] == q).all()
                at_send = sum(lays[q].send_splits[:r])
                assert np.array_equal(recvs[r][at:at + n, 1], (np.arange(at_send, at_send + n) % 60000).astype(dtype))
            at += n
    recs, state = _log(lib)
    assert state == [0, 0, 0], state                      # nothing unmatched, no size mismatch, every group closed
    # the log: per rank ONE group with one send and one receive per peer with rows to move, at the prefix-sum offsets
    for r in range(world):
        mine = recs[(recs[:, 1] == r) & (recs[:, 0] <= 1)]
        exp = []
        so = ro = 0
        for q in range(world):
            sb, rb = send_rows[r][q] * row_bytes, recv_rows[r][q] * row_bytes
            if sb:
                exp.append((0, r, q, sends[r].ctypes.data + so, sb))
            if rb:
                exp.append((1, r, q, recvs[r].ctypes.data + ro, rb))
            so += sb
            ro += rb
        assert [tuple(int(v) for v in row) for row in mine] == exp
    assert int((recs[:, 0] == 4).sum()) == world and int((recs[:, 0] == 5).sum()) == world


def test_shared_communicator_carries_other_row_counts(stub):
    """sg_comm_share: a second layout (an MGCN level) on the SAME communicator -- no second ncclCommInitRank."""
    lib, comms = stub
    world = 3
    _, _, _, _, lays_a = _folded_layouts(world, 48, 32)
    _, _, _, _, lays_b = _folded_layouts(world, 24, 16)
    hs = _make_comms(lib, comms, world, [l.send_splits for l in lays_a], [l.recv_splits for l in lays_a])
    shared = []
    for r in range(world):
        h = ctypes.c_void_p()
        rc = lib.sg_comm_share(hs[r], (c_int64 * world)(*lays_b[r].send_splits), (c_int64 * world)(*lays_b[r].recv_splits), ctypes.byref(h))
        assert rc == 0, lib.sg_last_error()
        shared.append(h)
    for lays, handles in ((lays_b, shared), (lays_a, hs), (lays_b, shared)):
        sends = [np.full((l.n_send, 4), r + 1, dtype=np.float32) * np.arange(1, l.n_send + 1, dtype=np.float32)[:, None]
                 for r, l in enumerate(lays)]
        recvs = [np.zeros((l.n_ext - l.n_own, 4), dtype=np.float32) for l in lays]
        for r in range(world):
            assert lib.sg_halo_exchange(handles[r], sends[r].ctypes.data, recvs[r].ctypes.data, 16, None) == 0
        want = _all_to_all_single(sends, [l.send_splits for l in lays], [l.recv_splits for l in lays])
        for r in range(world):
            assert np.array_equal(recvs[r], want[r])
    # the base handles may go first: the communicator lives as long as its last user
    for h in hs:
        lib.sg_comm_destroy(h)
        comms.remove(h)
    l = lays_b
    sends = [np.full((l[r].n_send, 2), r, dtype=np.float32) for r in range(world)]
    recvs = [np.zeros((l[r].n_ext - l[r].n_own, 2), dtype=np.float32) for r in range(world)]
    for r in range(world):
        assert lib.sg_halo_exchange(shared[r], sends[r].ctypes.data, recvs[r].ctypes.data, 8, None) == 0
    for r in range(world):
        assert np.array_equal(recvs[r], _all_to_all_single(sends, [x.send_splits for x in l], [x.recv_splits for x in l])[r])
    comms.extend(shared)
    assert _log(lib)[1] == [0, 0, 0]


@pytest.mark.parametrize("world", [2, 8])
def test_part_run_walks_a_schedule_of_collectives(stub, world):
    """sg_part_run with exchange / all-reduce / all-gather steps (no kernel steps: those need a device): every rank's schedule
    runs to the end, the reductions see every rank, the gathers land rank-major."""
    lib, comms = stub
    _, _, _, _, lays = _folded_layouts(world, 48, 32)
    hs = _make_comms(lib, comms, world, [l.send_splits for l in lays], [l.recv_splits for l in lays])
    C = 8
    sends = [np.full((l.n_send, C), r + 1, dtype=np.float32) for r, l in enumerate(lays)]
    recvs = [np.zeros((l.n_ext - l.n_own, C), dtype=np.float32) for l in lays]
    red = [np.arange(5, dtype=np.float32) * (r + 1) for r in range(world)]
    gin = [np.full(3, r, dtype=np.int64) for r in range(world)]
    gout = [np.full(3 * world, -1, dtype=np.int64) for _ in range(world)]
    keep = []
    for r in range(world):
        steps = (capi.sg_part_step * 4)()
        steps[0].kind, steps[0].n, steps[0].send, steps[0].recv = 1, C * 4, sends[r].ctypes.data, recvs[r].ctypes.data
        steps[1].kind, steps[1].n, steps[1].recv = 2, 5, red[r].ctypes.data
        steps[2].kind, steps[2].n, steps[2].send, steps[2].recv = 3, 24, gin[r].ctypes.data, gout[r].ctypes.data
        steps[3].kind, steps[3].n, steps[3].send, steps[3].recv = 1, C * 4, sends[r].ctypes.data, recvs[r].ctypes.data
        keep.append(steps)
        assert lib.sg_part_run(hs[r], steps, 4, None) == 0, lib.sg_last_error()
        assert lib.sg_part_failed_step() == -1
    want = _all_to_all_single(sends, [l.send_splits for l in lays], [l.recv_splits for l in lays])
    tot = world * (world + 1) / 2
    for r in range(world):
        assert np.array_equal(recvs[r], want[r])
        assert np.array_equal(red[r], np.arange(5, dtype=np.float32) * tot)
        assert np.array_equal(gout[r], np.repeat(np.arange(world), 3))
    assert _log(lib)[1] == [0, 0, 0]


def test_a_failed_send_closes_its_group_and_names_the_step(stub):
    """an error between ncclGroupStart and ncclGroupEnd must not leave the thread's group open (later RCCL calls,
    torch.distributed's fallback included, would be queued into it and hang); sg_part_run says which step failed."""
    lib, comms = stub
    world = 3
    _, _, _, _, lays = _folded_layouts(world, 24, 16)
    hs = _make_comms(lib, comms, world, [l.send_splits for l in lays], [l.recv_splits for l in lays])
    lay = lays[0]
    send = np.zeros((lay.n_send, 4), dtype=np.float32)
    recv = np.zeros((lay.n_ext - lay.n_own, 4), dtype=np.float32)
    red = np.ones(4, dtype=np.float32)
    steps = (capi.sg_part_step * 3)()
    steps[0].kind, steps[0].n, steps[0].recv = 2, 4, red.ctypes.data
    steps[1].kind, steps[1].n, steps[1].send, steps[1].recv = 1, 16, send.ctypes.data, recv.ctypes.data
    steps[2].kind, steps[2].n, steps[2].recv = 2, 4, red.ctypes.data
    lib.sg_comm_test_fail_send(1)                         # the second send of the exchange fails
    rc = lib.sg_part_run(hs[0], steps, 3, None)
    assert rc != 0
    msg = lib.sg_last_error().decode()
    assert "step 1 of 3" in msg and "group was closed" in msg, msg
    assert lib.sg_part_failed_step() == 1
    recs, state = _log(lib)
    assert state[2] == 0, "the group was left open"
    assert int((recs[:, 0] == 4).sum()) == int((recs[:, 0] == 5).sum()) == 1
    with pytest.raises(capi.PartRunError) as e:
        lib.sg_comm_test_fail_send(0)
        capi.part_run(type("H", (), {"_h": hs[0]})(), steps, 3, None)
    assert e.value.step == 1


def test_hang_guard_ends_the_worker_with_its_own_exit_code(tmp_path):
    """a known-answer exchange that HANGS must fail the attempt fast: with
    SEMIGCN_DIST_KNOWN_ANSWER_TIMEOUT set (bench.py's supervisor sets 45 s for its workers) a timer thread ends the process
    with exit code 86 and leaves a mark for the supervisor; without the variable nothing is ever ended."""
    import subprocess
    code = ("import os, sys, time; sys.path.insert(0, %r); from semigcn_amd import dist as d\n"
            "with d._HangGuard('the test section'):\n    time.sleep(float(sys.argv[1]))\nprint('survived')\n" % ROOT)
    env = dict(os.environ, SEMIGCN_DIST_KNOWN_ANSWER_TIMEOUT="1", SEMIGCN_BENCH_MARK=str(tmp_path), RANK="3")
    r = subprocess.run([sys.executable, "-c", code, "20"], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 86 and "did not finish within 1 s" in r.stderr and "survived" not in r.stdout
    assert (tmp_path / "native_comm_hang_rank3").exists()
    r = subprocess.run([sys.executable, "-c", code, "0.1"], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "survived" in r.stdout                      # a section that finishes in time cancels the timer
    env.pop("SEMIGCN_DIST_KNOWN_ANSWER_TIMEOUT")
    r = subprocess.run([sys.executable, "-c", code, "1.5"], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "survived" in r.stdout                      # off by default


def test_two_communicator_rule_is_asserted():
    """a torch.distributed collective left un-awaited when sg_part_run is about to enqueue native ones
    would interleave two RCCL communicators on one device: the phase path refuses."""
    from semigcn_amd import dist as d
    d._assert_c10d_drained()
    d._c10d_in_flight[0] += 1
    try:
        with pytest.raises(RuntimeError, match="un-awaited"):
            d._assert_c10d_drained()
    finally:
        d._c10d_in_flight[0] -= 1
    d._assert_c10d_drained()
