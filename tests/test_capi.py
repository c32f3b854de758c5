"""The C-ABI library loads without a GPU and exports exactly what include/semigcn.h
declares; the product has no CPU path and no route into oracle/."""
import ctypes
import os
import re

import pytest
import torch

from semigcn_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    text = open(os.path.join(ROOT, "include", "semigcn.h")).read()
    return re.findall(r"^SG_API\s+[\w\s\*]+?\b(sg_\w+)\s*\(", text, flags=re.M)


def test_header_declares_expected_entry_points():
    syms = header_symbols()
    assert len(syms) == len(set(syms)) and len(syms) >= 16
    for must in ("sg_graph_create", "sg_spmm", "sg_pool_mean", "sg_unpool", "sg_gather_rows", "sg_last_error"):
        assert must in syms


def test_library_exports_every_header_symbol():
    assert os.path.isfile(capi.library_path()), "build with: make -C semigcn_amd/csrc"
    lib = ctypes.CDLL(capi.library_path())
    for name in header_symbols():
        assert hasattr(lib, name), f"{name} declared in semigcn.h but not exported"


def test_ctypes_signatures_cover_header():
    assert sorted(capi._SIGNATURES) == sorted(header_symbols())
    lib = capi.load()
    assert lib.sg_abi_version() == 1
    assert isinstance(lib.sg_last_error(), bytes)


def test_argument_validation_without_gpu():
    lib = capi.load()
    out = ctypes.c_void_p()
    assert lib.sg_graph_create(None, -1, 4, None, ctypes.byref(out)) == -1
    assert b"negative" in lib.sg_last_error()
    assert lib.sg_graph_create(None, 5, 4, None, ctypes.byref(out)) == -1
    assert lib.sg_spmm(None, 0, None, 0, None, 0, None, 0, None, 0, 4, 0, 1.0, 0.0, 0.0, None) == -1
    assert b"null graph" in lib.sg_last_error()
    assert lib.sg_graph_destroy(None) == 0 and lib.sg_pool_destroy(None) == 0
    # newer entry points reject bad arguments before touching a device as well
    n, man = ctypes.c_int64(0), ctypes.c_int(0)
    assert lib.sg_mesh_edges(None, 5, 10, None, None, ctypes.byref(n), ctypes.byref(man), None) == -1
    assert b"null pointer" in lib.sg_last_error()
    assert lib.sg_mask_dilate(None, None, None, 1, None) == -1
    assert lib.sg_face_mask(None, 3, 4, None, None, 1, None) == -1
    g2 = (ctypes.c_float * 2)(1.0, 1.0)
    assert lib.sg_mesh_loss_bwd_det(None, None, None, None, None, None, g2, 0, 0, 0, None, None, None, None) == -1
    assert b"incidence" in lib.sg_last_error()
    assert lib.sg_bn_bwd_coeffs(None, 0, 4, 1.0, None, None, None, None, None, None, None) == -1
    assert lib.sg_bn_merge_tiles(None, 1, 1, 0, 4, None, None, None) == -1
    assert lib.sg_bn_stats_finalize(None, 1, 1, 4, None, None, None, None, 0.1, 1e-5, None, None, None) == -1
    assert lib.sg_multi_add(9, None, None, None, None, None, None) == -1
    assert b"at most 8" in lib.sg_last_error()
    assert lib.sg_multi_add(0, None, None, None, None, None, None) == 0
    assert lib.sg_input_prep(None, None, None, None, None, None, 4, 5, 0, None) == -1
    assert lib.sg_input_prep_bwd(None, 4, None, None, None, None, None, None, None, None, None, 5, 0, None) == -1
    assert lib.sg_input_prep_blocks(0) == 0 and lib.sg_input_prep_blocks(1025) == 2


def test_block_and_trace_entry_points_without_gpu():
    """sg_block_* / sg_trace_* (include/semigcn.h; the per-block orchestration of util/networks.py:83-101 below the C ABI):
    the ctypes mirror of the descriptor has the library's size, and every entry point rejects a bad descriptor before it
    touches a device."""
    lib = capi.load()
    assert ctypes.sizeof(capi.sg_block) == lib.sg_block_sizeof()
    blk = capi.sg_block()
    for fn in (lib.sg_block_forward, lib.sg_block_backward):
        assert fn(None, None) == -1 and b"null block" in lib.sg_last_error()
        assert fn(ctypes.byref(blk), None) == -1 and b"null graph" in lib.sg_last_error()
    assert lib.sg_block_workspace(None, 0) == -1 and lib.sg_block_workspace(ctypes.byref(blk), 1) == -1
    assert lib.sg_block_planar(None) == -1 and b"null block" in lib.sg_last_error()
    assert lib.sg_block_planar(ctypes.byref(blk)) == -1 and b"null graph" in lib.sg_last_error()
    for fn in (lib.sg_block_chain_forward, lib.sg_block_chain_backward, lib.sg_block_run):
        assert fn(None, 0, None) == 0
        assert fn(None, 2, None) == -1 and b"bad argument" in lib.sg_last_error()
        assert fn(None, -1, None) == -1
    blks = (capi.sg_block * 2)()
    assert lib.sg_block_chain_forward(blks, 2, None) == -1 and b"null graph" in lib.sg_last_error()
    assert lib.sg_block_run(blks, 2, None) == -1 and b"no phase" in lib.sg_last_error()
    # the launch trace: switched on and off without a device, nothing recorded
    assert lib.sg_trace_begin(-1, 7) == -1 and b"capacity" in lib.sg_last_error()
    assert lib.sg_trace_begin(0, 7) == 0 and lib.sg_trace_read(None, 0) == 0 and lib.sg_trace_end() == 0
    assert lib.sg_trace_read(None, 4) == -1
    # which weight shapes the thin-product kernels take (<= 256 entries, <= 32 columns / rows)
    assert lib.sg_thin_supported(16, 12) == 1 and lib.sg_thin_supported(3, 16) == 1 and lib.sg_thin_supported(32, 48) == 0
    assert lib.sg_tuning_set(capi.TUNE_BLOCK_PLANES, 1) == 0


def test_comm_entry_points_without_gpu():
    """sg_comm_* / sg_halo_exchange / sg_part_run (csrc/comm.hip): the schedule entry's ctypes mirror has
    the library's size, and every entry point rejects a bad argument before RCCL or a device is touched."""
    lib = capi.load()
    assert ctypes.sizeof(capi.sg_part_step) == lib.sg_part_step_sizeof()
    assert lib.sg_comm_unique_id(None) == -1 and b"null buffer" in lib.sg_last_error()
    out = ctypes.c_void_p()
    rows = (ctypes.c_int64 * 2)(0, 3)
    idb = ctypes.create_string_buffer(128)
    assert lib.sg_comm_create(None, 0, 2, rows, rows, ctypes.byref(out)) == -1 and b"no id" in lib.sg_last_error()
    assert lib.sg_comm_create(idb, 2, 2, rows, rows, ctypes.byref(out)) == -1 and b"bad rank" in lib.sg_last_error()
    assert lib.sg_comm_create(idb, 0, 2, None, None, ctypes.byref(out)) == -1 and b"row counts" in lib.sg_last_error()
    neg = (ctypes.c_int64 * 2)(0, -1)
    assert lib.sg_comm_create(idb, 0, 2, neg, rows, ctypes.byref(out)) == -1 and b"peer 1" in lib.sg_last_error()
    assert out.value is None
    assert lib.sg_comm_destroy(None) == 0
    assert lib.sg_halo_exchange(None, None, None, 8, None) == -1 and b"null communicator" in lib.sg_last_error()
    assert lib.sg_comm_all_reduce_f32(None, None, 4, None) == -1
    assert lib.sg_comm_all_gather(None, None, None, 4, None) == -1
    assert lib.sg_part_run(None, None, 0, None) == 0
    assert lib.sg_part_run(None, None, 2, None) == -1 and b"bad argument" in lib.sg_last_error()
    steps = (capi.sg_part_step * 2)()
    steps[0].kind = capi.STEP_BLOCKS                      # an empty run of blocks: nothing to do
    steps[1].kind = capi.STEP_EXCHANGE
    assert lib.sg_part_run(None, steps, 1, None) == 0
    assert lib.sg_part_run(None, steps, 2, None) == -1 and b"no communicator" in lib.sg_last_error()
    steps[0].kind = 7
    assert lib.sg_part_run(None, steps, 1, None) == -1 and b"kind 7" in lib.sg_last_error()
    assert lib.sg_comm_available() in (0, 1)


def test_no_cpu_fallback():
    ei = torch.tensor([[0, 1], [1, 0]])
    with pytest.raises(capi.SemigcnLibraryError, match="HIP device only"):
        capi.GraphHandle.from_edge_index(ei, 2)
    from semigcn_amd.nn import ChebConv
    with pytest.raises(capi.SemigcnLibraryError):
        ChebConv(4, 8, K=3)(torch.randn(2, 4), ei)
    from semigcn_amd.networks import SingleScaleGCN

    class D:
        z1, x_pos, edge_index = torch.randn(2, 3), torch.randn(2, 3), ei
    with pytest.raises(capi.SemigcnLibraryError):
        SingleScaleGCN("cpu")(D)


def test_missing_library_fails_loudly(monkeypatch):
    monkeypatch.setattr(capi, "_lib", None)
    monkeypatch.setattr(capi, "_LIB_PATH", "/nonexistent/libsemigcn_hip.so")
    with pytest.raises(capi.SemigcnLibraryError, match="no CPU or PyTorch fallback"):
        capi.load()


def test_product_never_touches_oracle():
    pkg = os.path.join(ROOT, "semigcn_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f
                # citations of /root/reference in docstrings are fine; reading it at run time is not
                assert not re.search(r"(sys\.path[^\n]*reference|open\([^\n]*reference|REFERENCE_ROOT)", src), f


def test_ring_kernel_isa_keeps_only_the_hand_written_vector_memory_waits(tmp_path):
    """csrc/spmm.hip::spmm_ring lives on the LDS-DMA of three tiles staying in flight across the barrier: every global
    read is an LDS-DMA and the only waits on the vector-memory queue are the hand-written ones.  hipcc once put an
    `s_waitcnt vmcnt(0)` in front of an LDS read it could not tell from a pending DMA (a load folded into a reference
    argument had lost its alias tag), which drains the ring and costs the kernel its lead silently.  Compile the file to
    ISA and count: per ring kernel 4 full drains in the producers (prologue, first iteration, the jump table's default, the
    end) + 3 in the consumers' rarely taken global-gather fallback, whose 3 loads are the only plain global loads."""
    import re
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = tmp_path / "spmm.s"
    subprocess.run([hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", f"-I{root}/include", "-DNDEBUG", "-S", "--cuda-device-only",
                    "-o", str(out), os.path.join(root, "semigcn_amd", "csrc", "spmm.hip")], check=True, capture_output=True, timeout=900)
    asm = out.read_text()
    kernels = list(re.finditer(r"^(_ZN2sg12_GLOBAL__N_19spmm_ring\w+):", asm, flags=re.M))
    assert len(kernels) >= 9
    for m in kernels:
        body = asm[m.end():asm.index(".Lfunc_end", m.end())]
        drains = len(re.findall(r"s_waitcnt vmcnt\(0\)", body))
        plain = len(re.findall(r"global_load_dword", body))
        assert drains <= 7 and plain == 3 and "global_load_lds_dwordx4" in body, (m.group(1), drains, plain)
