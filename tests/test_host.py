"""CPU-only tests: the C-ABI library loads and exports every symbol the header declares, host
helpers agree with the oracle, the synthetic generator is frozen, and the multi-rank gather /
assemble path (gloo, world_size 2) reproduces the single-process file."""
import ctypes
import hashlib
import json
import os
import re
import socket
import struct

import numpy as np
import pytest

from xsqueezeit_amd import binding, dist as xdist, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "xsi_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(xsi_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    assert declared == set(binding.SYMBOLS), declared ^ set(binding.SYMBOLS)
    L = binding.lib()
    for s in sorted(declared):
        assert hasattr(L, s), s
    assert L.xsi_hip_abi_version() == 1


def test_no_gpu_means_error_not_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(binding.XsiError) as e:
        binding.Context(0)
    assert e.value.code == binding.XSI_ERR_HIP
    assert "no CPU fallback" in str(e.value)


def test_make_header_matches_oracle():
    from oracle import oracle
    n = 30
    rng = np.random.default_rng(1)
    lines = []
    for _ in range(70):
        al = (rng.random(2 * n) < 0.2).astype(np.int32)
        gt = (al + 1) << 1
        gt[1::2] |= 1
        lines.append((gt.astype(np.int32), 2))
    ref = oracle.encode_file(lines, n, block_len=32, mac_thr=3, default_phased=1)
    io, so = struct.unpack_from("<QQ", ref, 72)
    hf = binding.HeaderFields(n, 2, 32, 3, 1, 0, 70, 70, io, so)
    h = (ctypes.c_uint8 * 256)()
    binding.check(binding.lib().xsi_hip_make_header(ctypes.byref(hf), h))
    assert bytes(h) == ref[:256]


def test_fill_loop_helpers_match_the_reference_rules(golden_dir):
    """The parameter derivation of the compressor's fill loop, in the library instead of test code:
    MAC threshold (gt_compressor_new.hpp:96-99), default phase (xcf.cpp:811-836) and the BM values of
    the variant BCF (xcf.cpp:685-703), against the oracle's restatement and the micro fixtures."""
    from oracle import oracle
    from xsqueezeit_amd import vcf_lite
    from test_oracle import _random_lines, ANCHORS
    L = binding.lib()
    for n, pl, maf in ((2504, 2, 0.001), (10, 2, 0.002), (64976 // 2, 2, 0.001), (250000, 2, 0.001), (7, 1, 0.3), (5, 2, 0.0)):
        assert L.xsi_mac_threshold(n, pl, maf) == int(float(n * pl) * maf)
    rng = np.random.default_rng(3)
    cases = []
    for kw in (dict(), dict(phase=True), dict(missing=True, eov=True, phase=True, multi=True)):
        cases.append((_random_lines(rng, 37, 9, **kw), 37))
    unphased = [((np.full(20, 2, np.int32)), 2)] * 3
    cases.append((unphased, 10))
    hap = [(np.full(10, 2, np.int32), 2)] + unphased
    cases.append((hap, 10))
    for name in sorted(ANCHORS):
        samples, recs = vcf_lite.read_vcf(os.path.join(golden_dir, name + ".vcf"))
        cases.append(([(r["gt"], r["n_allele"]) for r in recs], len(samples)))
    for lines, n in cases:
        rows = [np.ascontiguousarray(gt, dtype=np.int32) for gt, _ in lines[:3]]
        ptrs = (ctypes.c_void_p * len(rows))(*[r.ctypes.data for r in rows])
        ngt = (ctypes.c_uint32 * len(rows))(*[r.size for r in rows])
        assert L.xsi_default_phased(ptrs, ngt, len(rows), n) == oracle.default_phased_of(lines, n)
    # BM: block advances every block_len BCF lines, the offset counts binary lines inside the block
    st = binding.BmState()
    L.xsi_bm_init(ctypes.byref(st))
    nal = [2, 3, 2, 2, 4, 2, 2, 2, 3, 2, 2]
    block = off = 0
    for i, na in enumerate(nal):
        if i and i % 4 == 0:
            block, off = block + 1, 0
        assert L.xsi_bm_next(ctypes.byref(st), 4, na) == (block << 15) | off
        off += na - 1
    # 32768 binary lines in one block cannot be addressed: the reference throws (xcf.cpp:692-695)
    L.xsi_bm_init(ctypes.byref(st))
    for i in range(8192):
        assert L.xsi_bm_next(ctypes.byref(st), 8192, 5) == 4 * i
    L.xsi_bm_init(ctypes.byref(st))
    for i in range(8191):
        assert L.xsi_bm_next(ctypes.byref(st), 8192, 6) == 5 * i if 5 * i < 32768 else True
        if 5 * (i + 1) >= 32768:
            break
    assert L.xsi_bm_next(ctypes.byref(st), 8192, 6) == binding.XSI_ERR_FORMAT
    assert b"cannot be represented" in L.xsi_hip_last_error()


def test_encode_bound_covers_oracle_output():
    from oracle import oracle
    for n_haps, n_lines, bl, thr, seed in ((200, 300, 64, 0, 1), (5008, 600, 256, 5, 2), (5008, 300, 64, 2000, 3)):
        bits = synth.synth_bits(seed, 0, n_lines, n_haps)
        w = oracle.Writer(n_haps // 2, bl, thr, 1)
        w.append_rows(synth.bits_to_gt(bits, 1), 2)
        data = w.finalize(2)
        io = struct.unpack_from("<Q", data, 72)[0]
        p = binding.EncodeParams(n_haps // 2, bl, thr, 1, 0, 0)
        assert binding.lib().xsi_hip_encode_bound(ctypes.byref(p), n_lines, n_lines) >= io - 256


def test_synth_generator_is_frozen():
    man = json.load(open(os.path.join(ROOT, "tests", "golden", "synth_manifest.json")))
    for name in ("h20_l64", "h200_l700", "h5008_l3000"):
        m = man[name]
        bits = synth.synth_bits(m["seed"], m["first_line"], m["n_lines"], m["n_haps"])
        assert hashlib.sha256(bits.tobytes()).hexdigest() == m["bits_sha256"]


def test_oracle_matches_synth_manifest():
    from oracle import oracle
    man = json.load(open(os.path.join(ROOT, "tests", "golden", "synth_manifest.json")))
    for name in ("h20_l64", "h200_l700", "h5008_l3000"):
        m = man[name]
        bits = synth.synth_bits(m["seed"], m["first_line"], m["n_lines"], m["n_haps"])
        w = oracle.Writer(m["n_haps"] // 2, m["block_len"], m["mac_thr"], 1)
        w.append_rows(synth.bits_to_gt(bits, 1), 2)
        data = w.finalize(2)
        assert len(data) == m["size"] and hashlib.sha256(data).hexdigest() == m["sha256"]


def test_pack_unpack_rows():
    rng = np.random.default_rng(0)
    b = (rng.random((7, 5008)) < 0.3).astype(np.uint8)
    st = synth.row_stride_bytes(5008)
    assert st == 640 and st % 128 == 0
    p = synth.pack_rows(b, st)
    assert np.array_equal(synth.unpack_rows(p, 5008), b)
    assert p[:, 626:].sum() == 0


def test_shard_blocks_partition():
    for nb in (1, 7, 123, 1221):
        for w in (1, 2, 3, 8):
            seen = []
            for r in range(w):
                lo, hi = xdist.shard_blocks(nb, w, r)
                seen.extend(range(lo, hi))
            assert seen == list(range(nb))


def test_queries_are_routed_to_the_rank_that_owns_their_block():
    L = binding.lib()
    for nb in (1, 7, 123, 611, 1221):
        for w in (1, 2, 3, 8):
            owner = np.empty(nb, dtype=np.int64)
            for r in range(w):
                lo, hi = xdist.shard_blocks(nb, w, r)
                owner[lo:hi] = r
            bm = (np.arange(nb, dtype=np.int64) << 15) | 17
            assert np.array_equal(xdist.route_queries(bm, nb, w), owner)
            assert [L.xsi_hip_shard_of_block(nb, w, b) for b in range(nb)] == owner.tolist()
    assert L.xsi_hip_shard_of_block(5, 2, 5) == -1


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _rank_main(rank, world, port, n_haps, n_lines, bl, thr, q, path):
    import torch
    import torch.distributed as tdist
    from oracle import oracle
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    tdist.init_process_group("gloo", rank=rank, world_size=world)
    n_blocks = (n_lines + bl - 1) // bl
    lo, hi = xdist.shard_blocks(n_blocks, world, rank)
    l0, l1 = lo * bl, min(hi * bl, n_lines)
    # this rank's blocks, produced by the oracle standing in for the GPU encoder
    bits = synth.synth_bits(5, l0, l1 - l0, n_haps)
    w = oracle.Writer(n_haps // 2, bl, thr, 1)
    w.append_rows(synth.bits_to_gt(bits, 1), 2)
    data = w.finalize(2)
    io, so = struct.unpack_from("<QQ", data, 72)
    offs = np.frombuffer(data, dtype="<u8", count=(so - io) // 8, offset=io).astype(np.int64) - 256
    # blocks region without the final pad-to-8 (each block already padded to 4)
    region = np.frombuffer(data, dtype=np.uint8, count=io - 256, offset=256)
    # strip the region's pad to 8: block sizes are multiples of 4, so at most 4 zero bytes were added
    n_real = int(offs[-1]) + _block_len(data, 256 + int(offs[-1]))
    region = region[:n_real]
    got = xdist.gather_block_streams(torch.from_numpy(region.copy()), torch.from_numpy(offs.copy()), tdist)
    # the alternative exchange: sizes only, every rank writes its own byte range of the file
    my_start, total = xdist.exchange_region_offsets(len(region), tdist)
    if rank == 0:
        with open(path, "wb") as f:
            f.truncate(256 + total)
    tdist.barrier()
    xdist.write_own_range(path, region.tobytes(), my_start)
    tdist.barrier()
    if rank == 0:
        with open(path, "rb") as f:
            f.seek(256)
            pwritten = f.read()
        q.put((got[0].numpy().tobytes(), got[1].numpy().tolist(), pwritten))
    tdist.barrier()
    tdist.destroy_process_group()


def _block_len(data, off):
    """Length of the block at file offset `off`, padded to 4 (walk its GT dictionary)."""
    n = struct.unpack_from("<I", data, off + 4)[0]
    gt = off + [struct.unpack_from("<II", data, off + 8 + 8 * i) for i in range(n)][0][1]
    nk = struct.unpack_from("<I", data, gt + 4)[0]
    d = dict(struct.unpack_from("<II", data, gt + 8 + 8 * i) for i in range(nk))
    # bi-allelic, no side channels: the sparse matrix is last; walk it
    p = gt + d[0x21]
    n_bin = d[1]
    is_wah = _wah_bits(data, gt + d[0x10], n_bin)
    aet = 2 if struct.unpack_from("<Q", data, 112)[0] <= 65535 else 4
    for k in range(n_bin):
        if not is_wah[k]:
            num = struct.unpack_from("<H" if aet == 2 else "<I", data, p)[0] & (0x7FFF if aet == 2 else 0x7FFFFFFF)
            p += (1 + num) * aet
    return ((p - off) + 3) // 4 * 4


def _wah_bits(data, off, n):
    from oracle import oracle
    words = np.frombuffer(data, dtype="<u2", count=min((len(data) - off) // 2, n // 15 + 2), offset=off)
    return oracle.wah_extract(words, n)[0]


def test_two_rank_gather_reproduces_single_process_file(tmp_path):
    import torch.multiprocessing as mp
    from oracle import oracle
    n_haps, n_lines, bl, thr = 200, 1000, 128, 1
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    path = str(tmp_path / "shared.xsi")
    procs = [ctx.Process(target=_rank_main, args=(r, 2, port, n_haps, n_lines, bl, thr, q, path)) for r in range(2)]
    for p in procs:
        p.start()
    region, offs, pwritten = q.get(timeout=120)
    assert pwritten == region  # every rank wrote its own range: same bytes as the gathered stream
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    bits = synth.synth_bits(5, 0, n_lines, n_haps)
    names = ["S%d" % i for i in range(n_haps // 2)]
    w = oracle.Writer(n_haps // 2, bl, thr, 1, False, names)
    w.append_rows(synth.bits_to_gt(bits, 1), 2)
    ref = w.finalize(2)

    def hdr(io, so):
        hf = binding.HeaderFields(n_haps // 2, 2, bl, thr, 1, 0, n_lines, n_lines, io, so)
        h = (ctypes.c_uint8 * 256)()
        binding.check(binding.lib().xsi_hip_make_header(ctypes.byref(hf), h))
        return bytes(h)

    got = xdist.assemble_file(region, offs, hdr, names)
    assert got == ref


def _fake_block(b):
    """Stand-in for an encoded block: a length (multiple of 4, as interfaces.hpp:254-263 pads them) and bytes that
    depend on the block's number only."""
    n = 4 * (2 + (b * 7) % 9)
    return bytes(((b * 131 + i * 17) & 0xFF) for i in range(n))


def _fake_rank_main(rank, world, port, n_blocks, rounds, q):
    import torch
    import torch.distributed as tdist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    tdist.init_process_group("gloo", rank=rank, world_size=world)
    region_all, offs_all, shards = bytearray(), [], []
    for rd in range(rounds):  # a job gathered piece by piece: round rd covers a contiguous range of the file's blocks
        r_lo, r_hi = rd * n_blocks // rounds, (rd + 1) * n_blocks // rounds
        lo, hi = xdist.shard_blocks(r_hi - r_lo, world, rank)
        shards.append(hi - lo)
        parts = [_fake_block(r_lo + b) for b in range(lo, hi)]
        offs = np.cumsum([0] + [len(x) for x in parts[:-1]], dtype=np.int64) if parts else np.zeros(0, dtype=np.int64)
        region = np.frombuffer(b"".join(parts), dtype=np.uint8).copy()
        got = xdist.gather_block_streams(torch.from_numpy(region), torch.from_numpy(offs), tdist)
        if rank == 0:
            base = len(region_all)  # what the writer rank already holds (xsi_hip_gather_block_streams_round's region_base)
            region_all += got[0].numpy().tobytes()
            offs_all += [int(o) + base for o in got[1].numpy().tolist()]
    every = [None] * world
    tdist.all_gather_object(every, shards)
    if rank == 0:
        q.put((bytes(region_all), offs_all, every))
    tdist.barrier()
    tdist.destroy_process_group()


@pytest.mark.parametrize("world,n_blocks,rounds", [
    (4, 1221, 1),   # BASELINE configs[3]: 1221 blocks of 8192 lines
    (8, 1221, 1),   # ... over the 8 GPUs of a node: shards of 152 and 153 blocks
    (8, 1221, 3),   # gathered in three rounds (a shard encoded and sent piece by piece)
    (4, 3, 1),      # fewer blocks than ranks: a rank with nothing to send
])
def test_gather_at_world_4_and_8(world, n_blocks, rounds):
    """The multi-rank path at the world sizes of the driver's scaling run (gloo on the CPU): block b goes to
    rank floor(b G / B), every rank sends exactly its bytes and offsets, the writer rank's concatenation is the
    single-process stream with ascending offsets."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_fake_rank_main, args=(r, world, port, n_blocks, rounds, q)) for r in range(world)]
    for p in procs:
        p.start()
    region, offs, shards = q.get(timeout=240)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    blocks = [_fake_block(b) for b in range(n_blocks)]
    assert region == b"".join(blocks)
    assert offs == np.cumsum([0] + [len(x) for x in blocks[:-1]]).tolist()
    per_rank = [sum(sh) for sh in shards]
    assert sum(per_rank) == n_blocks
    if rounds == 1:
        assert max(per_rank) - min(per_rank) <= 1
        if n_blocks == 1221 and world == 8:
            assert sorted(set(per_rank)) == [152, 153]
        if n_blocks < world:
            assert 0 in per_rank
        # the C ABI's shard arithmetic is the same function
        L = binding.lib()
        for r in range(world):
            lo, hi = ctypes.c_uint64(), ctypes.c_uint64()
            L.xsi_hip_shard_blocks(n_blocks, world, r, ctypes.byref(lo), ctypes.byref(hi))
            assert (lo.value, hi.value) == xdist.shard_blocks(n_blocks, world, r)
            for b in (lo.value, hi.value - 1):
                if lo.value < hi.value:
                    assert L.xsi_hip_shard_of_block(n_blocks, world, b) == r


def test_bench_dry_launch_plans_the_north_star_job():
    """`bench.py --gpus 8 --config 3 --dry-launch`: the eight ranks of the north_star job (500 000 haplotypes x 10 M sites,
    1221 blocks) report their shards and their HBM plans without a GPU: input + decoded output + file image + (on the
    writer rank) the gathered streams leave room for the library's workspace on every rank of a 288 GB card."""
    import json
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--config", "3", "--dry-launch"],
                       capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 8 and out["ranks_reported"] == list(range(8))
    assert sum(out["blocks_per_rank"]) == 1221 and sorted(set(out["blocks_per_rank"])) == [152, 153]
    assert sum(out["sites_per_rank"]) == 10_000_000
    assert all(out["fits_per_rank"])
    plan = out["memory_plan_rank0"]
    assert plan["gathered"] > 0 and plan["held"] == max(out["held_bytes_per_rank"])  # the writer rank holds the most
    assert plan["held"] + plan["workspace_floor"] <= out["hbm_bytes"]
    assert plan["left_for_workspace"] >= 60 * 10**9  # the shard's 47 GB of permuted rows in ONE encode launch, as on a rank that gathers nothing


def test_header_is_plain_c():
    """include/xsi_hip.h is the drop-in boundary: it must compile as C99 on its own (no C++ / HIP types)."""
    import shutil
    import subprocess
    gcc = shutil.which("gcc")
    if not gcc:
        pytest.skip("no gcc")
    hdr = os.path.join(ROOT, "include", "xsi_hip.h")
    r = subprocess.run([gcc, "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only", "-x", "c", hdr],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_htslib_shim_compiles_against_declaration_only_prototypes():
    """csrc/xsi_htslib_shim.cpp (c_xcf_* of the reference's c_api.h, the -c and -x fill loops) needs htslib, which this
    image lacks, so the default library carries its stub.  Its real body is at least COMPILED here: -fsyntax-only with
    -DXSI_HAVE_HTSLIB against tests/cxx/htslib_decls/htslib/*.h, hand-written prototypes of the ~35 htslib names it
    uses (test infrastructure: no bodies, never linked).  It has never run; configs[0] stays untested."""
    import shutil
    import subprocess
    gxx = shutil.which("g++")
    if not gxx:
        pytest.skip("no g++")
    src = os.path.join(ROOT, "xsqueezeit_amd", "csrc", "xsi_htslib_shim.cpp")
    r = subprocess.run([gxx, "-std=c++17", "-Wall", "-Wextra", "-Werror", "-fsyntax-only", "-DXSI_HAVE_HTSLIB",
                        "-I", os.path.join(ROOT, "tests", "cxx", "htslib_decls"), src], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    # both fill loops and every c_xcf_* entry point of c_api.h:38-93 are defined in the htslib build ...
    text = open(src).read()
    for name in ("c_xcf_new", "c_xcf_add_readers", "c_xcf_update_readers", "c_xcf_sample_name", "c_xcf_nsamples",
                 "__c__xcf__get__genotypes__void", "c_xcf_delete", "xsi_compress_bcf", "xsi_decompress_bcf"):
        assert name + "(" in text, name
    # ... and the default library says it has no htslib and refuses the loops instead of pretending
    L = binding.lib()
    assert L.xsi_htslib_shim_available() == 0
    L.xsi_compress_bcf.restype = ctypes.c_int
    L.xsi_compress_bcf.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_double, ctypes.c_uint32, ctypes.c_uint32]
    L.xsi_decompress_bcf.restype = ctypes.c_int
    L.xsi_decompress_bcf.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_void_p]
    assert L.xsi_compress_bcf(b"in.bcf", b"out.xsi", 0.001, 8192, 0) == binding.XSI_ERR_UNSUPPORTED
    assert L.xsi_decompress_bcf(b"in.xsi", b"out.bcf", None) == binding.XSI_ERR_UNSUPPORTED


def test_file_num_samples_without_a_device(tmp_path):
    """c_xcf_nsamples-style query: header-only, works on a box without a GPU."""
    from oracle import oracle
    rng = np.random.default_rng(3)
    n = 23
    lines = [(((rng.random(2 * n) < 0.3).astype(np.int32) + 1) << 1 | np.tile([0, 1], n), 2) for _ in range(40)]
    data = oracle.encode_file(lines, n, block_len=16, mac_thr=1, default_phased=1)
    path = tmp_path / "s.xsi"
    path.write_bytes(data)
    assert binding.lib().xsi_file_num_samples(str(path).encode()) == n
    (tmp_path / "bad.xsi").write_bytes(b"\0" * 300)
    assert binding.lib().xsi_file_num_samples(str(tmp_path / "bad.xsi").encode()) == -4
    assert binding.lib().xsi_file_num_samples(str(tmp_path / "missing.xsi").encode()) == -6


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher around it must start two ranks itself (children of
    torch.distributed.run, started before the parent touches HIP) and print ONE JSON line from rank 0.
    --dry-launch keeps the ranks off the GPU: rendezvous over gloo, every rank reports in."""
    import json
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-launch"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["ranks_reported"] == [0, 1]
    assert len(set(out["pids"])) == 2 and os.getpid() not in out["pids"]


@pytest.mark.parametrize("isa", ["0", "1"])
def test_append_packer_other_instruction_sets(isa):
    """The same checks with the scalar and the AVX2 form forced (the library picks the widest form once per process)."""
    import subprocess
    import sys
    env = dict(os.environ, XSI_PACK_ISA=isa)
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", os.path.join(ROOT, "tests", "test_host.py"), "-k",
                        "test_append_packs_simple_rows_to_bits"], capture_output=True, text=True, env=env, cwd=ROOT, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]


def test_append_packs_simple_rows_to_bits():
    """The writer's pack-on-append (host only, AVX-512 / AVX2 / scalar by the CPU): a row of alleles 0 / 1 with the
    default phase on the second values becomes its ALT-carrier bit row; anything else is left to the int32 path."""
    L = binding.lib()
    rng = np.random.default_rng(8)
    for n in list(range(2, 80, 2)) + [5008, 64976 * 2 // 2]:
        for dp in (0, 1):
            al = (rng.random(n) < 0.3).astype(np.int32)
            gt = ((al + 1) << 1).astype(np.int32)
            gt[1::2] |= dp
            gt[0::2] |= rng.integers(0, 2, size=len(gt[0::2])).astype(np.int32)  # the first value's phase bit is not stored
            out = np.full((n + 7) // 8 + 8, 0xEE, dtype=np.uint8)
            assert L.xsi_debug_pack_bit_row(gt.ctypes.data, n, dp, out.ctypes.data) == 1
            want = np.packbits(al.astype(np.uint8), bitorder="little")
            assert np.array_equal(out[:len(want)], want), (n, dp)
            assert (out[len(want):] == 0xEE).all()  # nothing written beyond ceil(n / 8)
            for pos, bad in ((n - 1, ((al[n - 1] + 1) << 1) | (1 - dp)),   # second value with the other phase
                             (n // 2, 6), (n // 2, 0), (0, -2147483647), (n - 2, 1), (n - 1, -2147483648)):
                g2 = gt.copy()
                g2[pos] = bad
                assert L.xsi_debug_pack_bit_row(g2.ctypes.data, n, dp, out.ctypes.data) == 0, (n, dp, pos, bad)


@pytest.mark.parametrize("isa", [None, "0", "1"])
def test_host_packer_under_sanitizers(isa):
    """`make -C xsqueezeit_amd/csrc asan-host`: the writer's packer (csrc/xsi_pack.cpp) built with AddressSanitizer and
    UndefinedBehaviorSanitizer against exact-size heap buffers, every row length from 1 to 1100 and the bench sizes, in
    every instruction-set form this CPU has (the reference's own sanitizer build: /root/reference/Makefile:7-10)."""
    import subprocess
    csrc = os.path.join(ROOT, "xsqueezeit_amd", "csrc")
    subprocess.check_call(["make", "-s", "-C", csrc, "asan-host"])
    env = dict(os.environ, XSI_ENABLE_TUNING_ENV="1", ASAN_OPTIONS="detect_leaks=1:abort_on_error=0")
    env.pop("XSI_PACK_ISA", None)
    env.pop("LD_PRELOAD", None)
    if isa is not None:
        env["XSI_PACK_ISA"] = isa
    r = subprocess.run([os.path.join(ROOT, "tests", "cxx", "_build", "pack_asan")], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0 and "pack_asan ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


def test_generated_wah_classification_is_in_sync():
    """csrc/xsi_wah_classify.inc (one hand-scheduled asm statement) is what tools/gen_wah_classify.py writes."""
    import subprocess
    import sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_wah_classify.py"), "--check"])
    assert r.returncode == 0


def test_north_star_summary_is_flat_scalars():
    """The N > 1 line repeats the north_star job as flat scalars: the summary is made of the sub-run's
    own fields, every value a scalar, an error text carried when the sub-run failed."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("xsi_bench", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    o3 = {"metric": "GT cells/sec", "value": 1.0e13, "unit": "GT cells/s", "ms_per_step": 500.0, "gather_ms": 12.5,
          "ms_per_step_per_rank": [480.0, 500.0, 470.0, 490.0], "roundtrip_equal": True, "bit_exact_vs_oracle": True,
          "config": {"workload": "BASELINE configs[3]", "haps": 500000},
          "roofline": {"frac": 0.05, "chain_encode_ms": 240.0, "chain_decode_ms": 139.0, "stage_ms": {"x": 1}}}
    ns = bench.north_star_summary(o3, 4)
    assert ns["value"] == 1.0e13 and ns["ms_per_step"] == 500.0 and ns["frac"] == 0.05 and ns["gather_ms"] == 12.5
    assert ns["n_gpus"] == 4 and ns["scaling"] == "strong" and ns["max_rank_ms"] == 500.0 and ns["min_rank_ms"] == 470.0
    assert ns["roundtrip_equal"] is True and ns["bit_exact_vs_oracle"] is True and ns["error"] is None
    assert all(not isinstance(v, (dict, list)) for v in ns.values())
    bad = bench.north_star_summary({"error": "RuntimeError: x"}, 8)
    assert bad["error"] == "RuntimeError: x" and bad["value"] is None and bad["max_rank_ms"] is None and bad["n_gpus"] == 8
