"""Multi-GPU layer on the GPU box (one rank: the box has one GPU): the C-ABI shard / gather over RCCL
(include/xsi_hip.h "multi-GPU") on GPU-encoded blocks, and bench.py's multi-rank code path with --force-dist."""
import ctypes
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from xsqueezeit_amd import binding, synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(autouse=True)
def _torch_first():
    import gpu_util as G
    G.ctx()


def test_c_abi_gather_world_one_on_gpu_encoded_blocks():
    """xsi_hip_comm_create (ncclCommInitRank, world 1) + xsi_hip_gather_block_streams on what the GPU encoder wrote:
    the writer rank's image must be the encoder's region, its offsets the encoder's file offsets - 256; a writer
    buffer that is too small must fail with XSI_ERR_CAPACITY before any byte moves."""
    import gpu_util as G
    torch = G.torch_mod()
    L = binding.lib()
    n_haps, n_lines, bl = 5008, 3000, 512
    bits = synth.synth_bits(9, 0, n_lines, n_haps)
    stride = synth.row_stride_bytes(n_haps)
    packed = synth.pack_rows(bits, stride)
    p = G.params(n_haps // 2, bl, 5)
    region, offsets, res = G.encode_packed(packed, n_haps, p)
    n_blocks = len(offsets)
    d_region = G.dev_u8(np.frombuffer(region, dtype=np.uint8))
    d_offs = torch.from_numpy((offsets.astype(np.int64) - 256)).cuda()
    idb = (ctypes.c_uint8 * 128)()
    binding.check(L.xsi_hip_comm_unique_id(idb))
    comm = ctypes.c_void_p()
    binding.check(L.xsi_hip_comm_create(ctypes.byref(comm), G.ctx().handle, 1, 0, idb))
    try:
        assert L.xsi_hip_comm_world(comm) == 1 and L.xsi_hip_comm_rank(comm) == 0
        d_all = G.dev_empty(len(region) + 64)
        d_oall = torch.zeros(n_blocks, dtype=torch.int64, device="cuda")
        per_b = (ctypes.c_uint64 * 1)()
        per_n = (ctypes.c_uint64 * 1)()
        binding.check(L.xsi_hip_gather_block_streams(comm, d_region.data_ptr(), len(region), d_offs.data_ptr(), n_blocks, 0,
                                                     d_all.data_ptr(), d_all.numel(), d_oall.data_ptr(), n_blocks, per_b, per_n))
        binding.check(L.xsi_hip_comm_wait(comm, 1))
        assert per_b[0] == len(region) and per_n[0] == n_blocks
        assert d_all[:len(region)].cpu().numpy().tobytes() == region
        assert np.array_equal(d_oall.cpu().numpy().astype(np.uint64) + 256, offsets)
        rc = L.xsi_hip_gather_block_streams(comm, d_region.data_ptr(), len(region), d_offs.data_ptr(), n_blocks, 0,
                                            d_all.data_ptr(), len(region) - 1, d_oall.data_ptr(), n_blocks, per_b, per_n)
        assert rc == binding.XSI_ERR_CAPACITY
        # the file a writer rank assembles from the gathered image equals the oracle's
        names = ["S%d" % i for i in range(n_haps // 2)]
        got = G.assemble_file(d_all[:len(region)].cpu().numpy().tobytes(), d_oall.cpu().numpy().astype(np.uint64) + 256, p,
                              n_lines, n_lines, names)
        assert got == G.oracle_file_from_bits(bits, p, names)
    finally:
        L.xsi_hip_comm_destroy(comm)
    lo, hi = ctypes.c_uint64(), ctypes.c_uint64()
    L.xsi_hip_shard_blocks(1221, 8, 3, ctypes.byref(lo), ctypes.byref(hi))
    assert (lo.value, hi.value) == (458, 611)


def test_point_to_point_branch_with_self_send_and_rounds(monkeypatch):
    """The grouped ncclSend / ncclRecv branch of the gather has no second GPU to run on here, so the writer rank sends its
    own part to ITSELF (XSI_DIST_SELF_SEND=1: ncclSend + ncclRecv posted inside the group instead of the device copy).
    Two "ranks' worth" of regions: the GPU-encoded blocks region is cut at a block boundary and gathered in two rounds
    (xsi_hip_gather_block_streams_round) into one buffer - second round behind the first, its offsets (relative to its
    own part, as a rank's are) rebased by the bytes in front of it.  The assembled file must equal the oracle's.
    Then a receive that is refused (XSI_DIST_TEST_BAD_RECV=1: the first post of the group fails): the call reports the error,
    the group is closed, xsi_hip_comm_wait does not hang, and the next gather on the same communicator works."""
    import gpu_util as G
    torch = G.torch_mod()
    L = binding.lib()
    n_haps, n_lines, bl = 5008, 4000, 512
    bits = synth.synth_bits(12, 0, n_lines, n_haps)
    stride = synth.row_stride_bytes(n_haps)
    packed = synth.pack_rows(bits, stride)
    p = G.params(n_haps // 2, bl, 5)
    region, offsets, res = G.encode_packed(packed, n_haps, p)
    n_blocks = len(offsets)
    k = 3  # "rank 0": blocks [0, 3), "rank 1": blocks [3, n)
    cut = int(offsets[k]) - 256
    rel = offsets.astype(np.int64) - 256
    parts = [(region[:cut], rel[:k]), (region[cut:], rel[k:] - cut)]
    idb = (ctypes.c_uint8 * 128)()
    binding.check(L.xsi_hip_comm_unique_id(idb))
    comm = ctypes.c_void_p()
    binding.check(L.xsi_hip_comm_create(ctypes.byref(comm), G.ctx().handle, 1, 0, idb))
    monkeypatch.setenv("XSI_DIST_SELF_SEND", "1")
    try:
        d_all = G.dev_empty(len(region) + 64)
        d_all.fill_(0xEE)
        d_oall = torch.full((n_blocks,), -1, dtype=torch.int64, device="cuda")
        per_b = (ctypes.c_uint64 * 1)()
        per_n = (ctypes.c_uint64 * 1)()
        base_b = base_n = 0
        keep = []
        for reg, offs in parts:
            d_reg = G.dev_u8(np.frombuffer(reg, dtype=np.uint8))
            d_off = torch.from_numpy(np.ascontiguousarray(offs)).cuda()
            keep.append((d_reg, d_off))
            binding.check(L.xsi_hip_gather_block_streams_round(comm, d_reg.data_ptr(), len(reg), d_off.data_ptr(), len(offs), 0,
                                                              d_all.data_ptr(), len(region), d_oall.data_ptr(), n_blocks,
                                                              base_b, base_n, per_b, per_n))
            binding.check(L.xsi_hip_comm_wait(comm, 1))
            assert per_b[0] == len(reg) and per_n[0] == len(offs)
            base_b += len(reg)
            base_n += len(offs)
        assert d_all[:len(region)].cpu().numpy().tobytes() == region
        assert int(d_all[len(region)].item()) == 0xEE  # nothing behind the gathered bytes
        assert np.array_equal(d_oall.cpu().numpy().astype(np.uint64) + 256, offsets)
        names = ["S%d" % i for i in range(n_haps // 2)]
        got = G.assemble_file(d_all[:len(region)].cpu().numpy().tobytes(), d_oall.cpu().numpy().astype(np.uint64) + 256, p,
                              n_lines, n_lines, names)
        assert got == G.oracle_file_from_bits(bits, p, names)
        # a round that does not fit behind its base is refused before anything moves
        d_reg, d_off = keep[1]
        rc = L.xsi_hip_gather_block_streams_round(comm, d_reg.data_ptr(), len(parts[1][0]), d_off.data_ptr(), len(parts[1][1]), 0,
                                                  d_all.data_ptr(), len(region), d_oall.data_ptr(), n_blocks, cut + 1, k, per_b, per_n)
        assert rc == binding.XSI_ERR_CAPACITY
        binding.check(L.xsi_hip_comm_wait(comm, 1))
        # a refused receive: error out, group closed, communicator still good
        monkeypatch.setenv("XSI_DIST_TEST_BAD_RECV", "1")
        d_reg, d_off = keep[0]
        rc = L.xsi_hip_gather_block_streams(comm, d_reg.data_ptr(), len(parts[0][0]), d_off.data_ptr(), k, 0,
                                            d_all.data_ptr(), len(region), d_oall.data_ptr(), n_blocks, per_b, per_n)
        assert rc == binding.XSI_ERR_HIP, rc
        assert b"ncclRecv" in L.xsi_hip_last_error()
        binding.check(L.xsi_hip_comm_wait(comm, 1))
        monkeypatch.delenv("XSI_DIST_TEST_BAD_RECV")
        d_all.fill_(0)
        binding.check(L.xsi_hip_gather_block_streams(comm, d_reg.data_ptr(), len(parts[0][0]), d_off.data_ptr(), k, 0,
                                                     d_all.data_ptr(), len(region), d_oall.data_ptr(), n_blocks, per_b, per_n))
        binding.check(L.xsi_hip_comm_wait(comm, 1))
        assert d_all[:cut].cpu().numpy().tobytes() == region[:cut]
    finally:
        L.xsi_hip_comm_destroy(comm)


@pytest.mark.parametrize("torch_gather", [False, True])
def test_bench_multi_rank_path_with_one_rank(torch_gather):
    """bench.py --force-dist: process group (nccl), the C-ABI gather inside the timed region and alone, the
    per-rank report - on blocks the GPU encoded in this run.  Second case: the fallback bench.py takes when the
    library's communicator cannot be created (gather through torch.distributed)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["MASTER_PORT"] = "29578" if torch_gather else "29577"
    if torch_gather:
        env["XSI_BENCH_TORCH_GATHER"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--force-dist", "--config", "1",
                        "--sites", "40000", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["roundtrip_equal"] and out["n_gpus"] == 1
    assert out["gather_ms"] > 0 and len(out["ms_per_step_per_rank"]) == 1
    g = out["gathered"]
    assert g["ranks"] == 1 and g["own_part_equals_encode_output"] and g["offsets_ascending"]
    assert g["bytes"] == out["config"]["xsi_bytes_this_gpu"] and g["blocks"] == out["config"]["blocks_this_gpu"]
    assert ("torch.distributed" in g["via"]) == torch_gather


def test_bench_attaches_the_north_star_job_on_the_multi_rank_path():
    """At N > 1 `python bench.py` runs BASELINE configs[3] sharded over the same ranks after the weak-scaling line and
    attaches it as other_configs.  Rehearsed here with one rank (--force-dist) and a fraction of the
    sites: one process group for both runs, a second RCCL communicator, gather_ms, CPU sample and byte comparison of
    the first blocks in the sub-run too."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["MASTER_PORT"] = "29579"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--force-dist", "--sites-fraction", "0.007",
                        "--cpu-sample-cells", "6e8", "--steps", "1", "--warmup", "1"],
                       capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["roundtrip_equal"] and out["scaling"] == "weak" and out["gather_ms"] > 0
    assert out["cpu_baseline"] and out["bit_exact_vs_oracle"]
    (name, sub), = out["other_configs"].items()
    assert "configs[3]" in name and "north_star" in name and "error" not in sub
    assert sub["scaling"] == "strong" and sub["roundtrip_equal"] and sub["gather_ms"] > 0
    assert sub["config"]["haps"] == 500000 and sub["gathered"]["own_part_equals_encode_output"]
    assert sub["config"]["memory_plan_bytes"]["fits"]
    # ... and a second time as flat scalars at the END of the line: a record that keeps only the tail of
    # stdout, or only the top-level scalars of the line, still carries the job's result
    ns = out["north_star"]
    assert list(out)[-len(ns) - 1] == "north_star" and lines[0].rstrip().endswith("}")
    assert ns["value"] == sub["value"] and ns["ms_per_step"] == sub["ms_per_step"] and ns["gather_ms"] == sub["gather_ms"]
    assert ns["n_gpus"] == 1 and ns["scaling"] == "strong" and ns["roundtrip_equal"] is True and ns["error"] is None
    assert ns["frac"] == sub["roofline"]["frac"] and ns["bit_exact_vs_oracle"] == sub["bit_exact_vs_oracle"]
    assert ns["max_rank_ms"] >= ns["min_rank_ms"] > 0
    for k, v in ns.items():
        assert out["north_star_" + k] == v and not isinstance(v, (dict, list))
    tail = lines[0][-1500:]  # what a cut record keeps
    assert '"north_star_value"' in tail and '"north_star_gather_ms"' in tail and '"north_star_roundtrip_equal"' in tail
