"""GPU parity tests at depth: one full-depth block per chain kernel against the oracle (the short
blocks of test_gpu_packed.py never reach a deep PBWT order, long look-ahead histories or the later
batches of a block-batched call), the u32 accessor under random access with an evicting cache, and a
version-4 file image (u32 index)."""
import ctypes
import struct

import numpy as np
import pytest

from xsqueezeit_amd import binding, synth
from test_oracle import _random_lines

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _torch_first():
    """torch must initialise HIP before libxsi_hip.so is loaded: the wheel carries its own HIP runtime,
    and with the system one mapped first no device is found."""
    import gpu_util as G
    G.ctx()


def _device_synth(n_haps, n_lines, seed):
    """Synthetic matrix made on the device (the numpy mirror is checked elsewhere and is slow at this
    size); returns (bits01 [n_lines, n_haps] uint8, packed rows, stride)."""
    import gpu_util as G
    stride = synth.row_stride_bytes(n_haps)
    d = G.dev_empty(n_lines * stride)
    binding.check(binding.lib().xsi_hip_synth_packed(G.ctx().handle, seed, 0, n_lines, n_haps, d.data_ptr(), stride))
    G.torch_mod().cuda.synchronize()
    packed = d.cpu().numpy().reshape(n_lines, stride)
    return synth.unpack_rows(packed, n_haps), packed, stride


@pytest.mark.parametrize("n_haps,n_lines,thr,kernels", [
    (64976, 8192, 64, ("k_chain_rank_enc", "k_chain_decode_rank_big")),   # BASELINE configs[2] shape, one whole block
    (200000, 2048, 200, ("k_chain_rank_enc_multi", "k_chain_decode_rank_big")),   # configs[4] haplotype count: 4 workgroups per block
    (500000, 2048, 500, ("k_chain_rank_enc_multi", "k_chain_decode_rank_big")),   # configs[3] haplotype count: 8 workgroups per block
    (200000, 1024, 200, ("k_chain_stream", "k_chain_decode_rank_big")),           # the position-major streaming chain (forced)
    (600000, 512, 600, ("k_chain_stream", "k_chain_decode_rank_big")),            # above the multi-workgroup kernel's range
])
def test_full_depth_block_against_oracle(n_haps, n_lines, thr, kernels, monkeypatch):
    import gpu_util as G
    L = binding.lib()
    if kernels[0] == "k_chain_stream" and n_haps <= 524288:
        monkeypatch.setenv("XSI_NO_RANKENC_MULTI", "1")
    assert L.xsi_hip_chain_kernel(n_haps, 1, 0).decode() == kernels[0]
    assert L.xsi_hip_chain_kernel(n_haps, 1, 1).decode() == kernels[1]
    bits, packed, stride = _device_synth(n_haps, n_lines, 43)
    p = G.params(n_haps // 2, n_lines, thr)
    ref = G.oracle_file_from_bits(bits, p)
    region, offsets, res = G.encode_packed(packed, n_haps, p)
    assert res.n_wah_lines > n_lines // 3  # a deep chain, not a block of sparse lines
    names = ["S%d" % i for i in range(n_haps // 2)]
    got = G.assemble_file(region, offsets, p, n_lines, n_lines, names)
    io = struct.unpack_from("<Q", ref, 72)[0]
    if got[:io] != ref[:io]:
        first = next(i for i in range(min(len(got), len(ref))) if got[i] != ref[i])
        raise AssertionError("blocks region differs at file offset %d (sizes %d vs %d)" % (first, len(got), len(ref)))
    assert got == ref
    out, counts = G.decode_packed(got, n_haps, stride)
    assert np.array_equal(out, packed)
    assert np.array_equal(counts, bits.sum(1, dtype=np.int64).astype(np.int32))


@pytest.mark.parametrize("n_haps,n_blocks,block_len,thr,force", [
    (64976, 3, 700, 64, dict(XSI_RANK_WG_MIN_BLOCKS="1")),   # one-workgroup-per-block decode chain at depth (>= 192 blocks by default), 12 line ranges
    (64976, 3, 700, 64, dict(XSI_RANK_WG_MIN_BLOCKS="1", XSI_DEC_PHASES="1")),   # the same in one range
    (40000, 4, 777, 40, dict(XSI_RANK_WG_MIN_BLOCKS="1", XSI_DEC_PHASES="7")),   # ranges that do not divide the blocks' lines
    (50002, 3, 500, 50, dict(XSI_RANK_WG_MIN_BLOCKS="1", XSI_DEC_PHASES="3")),   # 56 chunks per wave: the instantiation whose gathers are pinned in front of the updates
    (5008, 6, 900, 5, dict(XSI_DEC_PHASES_SMALL="1", XSI_DEC_PHASES="5", XSI_RANKENC_MIN_N="2")),   # the batch-staged decode chain in ranges
    (64976, 1, 8192, 64, dict(XSI_RANK_WG_MIN_BLOCKS="1")),   # the kernel instantiation bench.py's default line decodes with (<64>, ramp + equal ranges, two-part boundary scan), one whole block deep
    (64976, 1, 8192, 64, dict(XSI_RANK_WG_MIN_BLOCKS="1", XSI_DEC_PHASES="1")),   # and in one range
    # round 6: the position-major decode chain (k_chain_decode_pos) in place of the one-workgroup rank chain
    (64976, 3, 700, 64, dict(XSI_RANK_WG_MIN_BLOCKS="1", XSI_POS_DECODE="1")),                      # a partial last chunk (16 of 64 positions), 12 line ranges
    (64976, 1, 8192, 64, dict(XSI_RANK_WG_MIN_BLOCKS="1", XSI_POS_DECODE="1", XSI_DEC_PHASES="1")),   # one whole block deep, one range
    (64976, 1, 8192, 64, dict(XSI_RANK_WG_MIN_BLOCKS="1", XSI_POS_DECODE="1")),                      # ... in the bench's ranges (array parked between launches)
    (40000, 4, 777, 40, dict(XSI_RANK_WG_MIN_BLOCKS="1", XSI_POS_DECODE="1", XSI_DEC_PHASES="7")),    # 625 chunks (odd), 40 per wave, tail chunk in the middle of a wave's slots
    (40960, 3, 500, 40, dict(XSI_RANK_WG_MIN_BLOCKS="1", XSI_POS_DECODE="1", XSI_DEC_PHASES="3")),    # N a multiple of 64: no partial chunk
    (50002, 3, 500, 50, dict(XSI_RANK_WG_MIN_BLOCKS="1", XSI_POS_DECODE="1", XSI_DEC_PHASES="3")),    # 56 chunks per wave
    (16450, 5, 600, 16, dict(XSI_RANK_WG_MIN_BLOCKS="1", XSI_POS_DECODE="1")),                       # the smallest size the family takes: 24 chunks per wave
    (24576, 3, 900, 24, dict()),                          # rank-tracking encode below 64 chunks per wave
    (12300, 2, 1200, 12, dict(XSI_RANKENC_MIN_N="2")),
])
def test_chain_kernel_variants_at_depth(n_haps, n_blocks, block_len, thr, force, monkeypatch):
    """The kernels the size rules pick for other block counts, forced through the environment (the
    library reads these two variables on every call)."""
    import gpu_util as G
    L = binding.lib()
    for k, v in force.items():
        monkeypatch.setenv(k, v)
    if "XSI_RANK_WG_MIN_BLOCKS" in force:
        assert L.xsi_hip_chain_kernel(n_haps, n_blocks, 1).decode() == ("k_chain_decode_pos" if "XSI_POS_DECODE" in force else "k_chain_decode_rank_wg")
    if "XSI_POS_DECODE" not in force:
        assert L.xsi_hip_chain_kernel(n_haps, n_blocks, 0).decode() == "k_chain_rank_enc"
    n_lines = n_blocks * block_len
    bits, packed, stride = _device_synth(n_haps, n_lines, 7)
    p = G.params(n_haps // 2, block_len, thr)
    ref = G.oracle_file_from_bits(bits, p)
    region, offsets, res = G.encode_packed(packed, n_haps, p)
    got = G.assemble_file(region, offsets, p, n_lines, n_lines, ["S%d" % i for i in range(n_haps // 2)])
    assert got == ref
    out, _ = G.decode_packed(got, n_haps, stride)
    assert np.array_equal(out, packed)


@pytest.mark.parametrize("n_haps,block_len,n_blocks", [
    (140000, 150, 2),   # 3 workgroups per block (an odd count: the last 16-byte units of a bitmap have no second half)
    (330000, 120, 2),   # 6
    (500000, 100, 2),   # 8
])
@pytest.mark.parametrize("list_thr", ["0", "1000000", None])
def test_long_row_encode_exchange_forms(n_haps, block_len, n_blocks, list_thr, monkeypatch):
    """k_chain_rank_enc_multi chooses per line between rank lists with one hand-off (sparse lines; a row with more ones
    than zeros as the list of its zeros) and bitmaps with two (dense lines).  Forced to either form for every line, and
    with the default threshold, the bytes are the oracle's (wah.hpp:506-578, internal_gt_record.hpp:32-49)."""
    import gpu_util as G
    L = binding.lib()
    if list_thr is not None:
        monkeypatch.setenv("XSI_MULTI_LIST_THR", list_thr)
    assert L.xsi_hip_chain_kernel(n_haps, n_blocks, 0).decode() == "k_chain_rank_enc_multi"
    n_lines = block_len * n_blocks
    bits, packed, stride = _device_synth(n_haps, n_lines, 11)
    # a dense row (more ones than zeros, few zeros) and an all-but-empty WAH line among them
    bits[5] = 1
    bits[5, ::997] = 0
    bits[7] = 0
    bits[7, 3::n_haps // 700] = 1
    packed = synth.pack_rows(bits, stride)
    p = G.params(n_haps // 2, block_len, n_haps // 1000)
    ref = G.oracle_file_from_bits(bits, p)
    before = L.xsi_hip_ctx_chain_fallbacks(G.ctx().handle)
    region, offsets, res = G.encode_packed(packed, n_haps, p)
    assert L.xsi_hip_ctx_chain_fallbacks(G.ctx().handle) == before
    got = G.assemble_file(region, offsets, p, n_lines, n_lines, ["S%d" % i for i in range(n_haps // 2)])
    assert got == ref


def test_long_row_encode_with_more_groups_than_a_round_of_eight_member_blocks():
    """140 000 haplotypes are 3 workgroups per block: 80 groups of a 256-CU chip, 90 blocks on them - the balanced line
    schedule with several cuts, parking slots and flags indexed by (group, member) beyond the 32 groups of the 500 000-
    haplotype shape.  Bytes == the oracle's, and the long-row kernel itself ran (no fallback to the streaming chain)."""
    import gpu_util as G
    L = binding.lib()
    n_haps, block_len, n_blocks = 140000, 36, 90
    n_lines = block_len * n_blocks
    bits, packed, stride = _device_synth(n_haps, n_lines, 21)
    p = G.params(n_haps // 2, block_len, 0)  # MAC threshold 0: every line with a carrier is a WAH line
    ref = G.oracle_file_from_bits(bits, p)
    before = L.xsi_hip_ctx_chain_fallbacks(G.ctx().handle)
    region, offsets, res = G.encode_packed(packed, n_haps, p)
    assert L.xsi_hip_ctx_chain_fallbacks(G.ctx().handle) == before
    assert res.n_wah_lines > n_lines // 2
    got = G.assemble_file(region, offsets, p, n_lines, n_lines, ["S%d" % i for i in range(n_haps // 2)])
    assert got == ref
    out, _ = G.decode_packed(got, n_haps, stride)
    assert np.array_equal(out, packed)


def test_multi_workgroup_chain_falls_back_instead_of_hanging(monkeypatch):
    """The encode chain over several workgroups per block exchanges rank lists and row slices once per line and
    needs every workgroup of a block resident at once.  If a workgroup never shows up (here: member 1 of every
    group leaves at once) the waits of the others run out (bounded by the 100 MHz clock, not by a poll count), the
    abort flag ends every workgroup of the launch, and the call runs the batch again with the one-workgroup-per-block
    streaming chain: same bytes as the oracle, and the context counts the fallback."""
    import gpu_util as G
    L = binding.lib()
    monkeypatch.setenv("XSI_MULTI_TEST_DESERT", "1")
    monkeypatch.setenv("XSI_MULTI_TIMEOUT_MS", "200")
    n_haps, n_lines = 140000, 24
    bits, packed, stride = _device_synth(n_haps, n_lines, 3)
    p = G.params(n_haps // 2, 8, 140)
    before = L.xsi_hip_ctx_chain_fallbacks(G.ctx().handle)
    region, offsets, res = G.encode_packed(packed, n_haps, p)
    assert L.xsi_hip_ctx_chain_fallbacks(G.ctx().handle) == before + 1
    assert res.n_blocks == 3
    names = ["S%d" % i for i in range(n_haps // 2)]
    got = G.assemble_file(region, offsets, p, n_lines, n_lines, names)
    assert got == G.oracle_file_from_bits(bits, p)
    # and without the desertion the same call takes the multi-workgroup chain, no fallback
    monkeypatch.delenv("XSI_MULTI_TEST_DESERT")
    region2, offsets2, res2 = G.encode_packed(packed, n_haps, p)
    assert L.xsi_hip_ctx_chain_fallbacks(G.ctx().handle) == before + 1
    assert bytes(region2) == bytes(region)


def test_phased_decode_with_a_short_last_block(monkeypatch):
    """The decode runs every block's WAH lines in ranges (expansion of the next range underneath the chain of the
    current one).  A block with fewer WAH lines than ranges has empty ranges: its first lines must still start
    from the identity and its ranks must survive the launches in which it has nothing to do."""
    import gpu_util as G
    monkeypatch.setenv("XSI_RANK_WG_MIN_BLOCKS", "1")
    monkeypatch.setenv("XSI_DEC_PHASES", "7")
    n_haps, block_len, thr = 40000, 777, 40
    n_lines = 3 * block_len + 4   # the fourth block: 4 lines
    bits, packed, stride = _device_synth(n_haps, n_lines, 11)
    p = G.params(n_haps // 2, block_len, thr)
    ref = G.oracle_file_from_bits(bits, p)
    region, offsets, res = G.encode_packed(packed, n_haps, p)
    got = G.assemble_file(region, offsets, p, n_lines, n_lines, ["S%d" % i for i in range(n_haps // 2)])
    assert got == ref
    out, _ = G.decode_packed(got, n_haps, stride)
    assert np.array_equal(out, packed)


def test_block_batched_calls_equal_single_call():
    """A workspace budget smaller than the job: encode and decode run as several batches of whole
    blocks inside one call and must produce the bytes / rows of the unbatched call."""
    import gpu_util as G
    L = binding.lib()
    n_haps, block_len, n_blocks = 5008, 256, 11
    n_lines = n_blocks * block_len - 37  # ragged last block
    bits, packed, stride = _device_synth(n_haps, n_lines, 11)
    p = G.params(n_haps // 2, block_len, 5)
    region1, offs1, res1 = G.encode_packed(packed, n_haps, p)
    file1 = G.assemble_file(region1, offs1, p, n_lines, n_lines, ["S%d" % i for i in range(n_haps // 2)])
    h = G.ctx().handle
    try:
        # room for about three blocks of per-line workspace: 4 batches of 3, 3, 3, 2 blocks
        per_block = (8 * ((n_haps + 63) // 64) + 2 * ((n_haps + 14) // 15 + 4) + 40) * block_len
        binding.check(L.xsi_hip_ctx_set_workspace_budget(h, 3 * per_block + 1000))
        region2, offs2, res2 = G.encode_packed(packed, n_haps, p)
        assert res2.n_blocks == n_blocks and res2.n_wah_lines == res1.n_wah_lines
        assert region2 == region1
        assert np.array_equal(offs2, offs1)
        out, counts = G.decode_packed(file1, n_haps, stride)
        assert np.array_equal(out, packed)
        assert np.array_equal(counts, bits.sum(1).astype(np.int32))
        # a budget below one block still makes progress, one block at a time
        binding.check(L.xsi_hip_ctx_set_workspace_budget(h, 1000))
        region3, offs3, _ = G.encode_packed(packed, n_haps, p)
        assert region3 == region1 and np.array_equal(offs3, offs1)
        out, _ = G.decode_packed(file1, n_haps, stride, first_block=2, n_blocks=5, max_rows=5 * block_len)
        assert np.array_equal(out, packed[2 * block_len:7 * block_len])
    finally:
        binding.check(L.xsi_hip_ctx_set_workspace_budget(h, 0))


def test_batched_decode_keeps_the_workspace_within_the_budget():
    """the batch-cutting pre-pass of xsi_hip_decode_packed must not size the expanded-row buffer by the
    WHOLE block range.  A fresh context (its row buffer has never grown) decodes 11 blocks under a budget of about
    three: the rows come back right and the context holds no more than the budget plus its small per-line arrays."""
    import ctypes
    import gpu_util as G
    import torch
    L = binding.lib()
    n_haps, block_len, n_blocks = 40000, 256, 11
    n_lines = n_blocks * block_len
    bits, packed, stride = _device_synth(n_haps, n_lines, 19)
    p = G.params(n_haps // 2, block_len, 40)
    region, offs, res = G.encode_packed(packed, n_haps, p)
    image = G.assemble_file(region, offs, p, n_lines, n_lines, ["S%d" % i for i in range(n_haps // 2)])
    per_line = 16 * ((n_haps + 63) // 64) + 64
    whole = int(res.n_wah_lines) * per_line
    budget = 3 * block_len * per_line
    assert whole > 2 * budget  # the unbatched row buffer would be more than twice the budget
    G.ctx()
    stream = torch.cuda.current_stream()
    c2 = binding.Context(0, stream.cuda_stream)
    try:
        binding.check(L.xsi_hip_ctx_set_workspace_budget(c2.handle, budget))
        d_file = G.dev_u8(np.frombuffer(image, dtype=np.uint8))
        d_out = G.dev_empty(n_lines * stride)
        rows = ctypes.c_uint64(0)
        binding.check(L.xsi_hip_decode_packed(c2.handle, d_file.data_ptr(), len(image), 0, n_blocks, d_out.data_ptr(),
                                              stride, n_lines, ctypes.byref(rows), None))
        c2.synchronize()
        assert rows.value == n_lines
        assert np.array_equal(d_out.cpu().numpy().reshape(n_lines, stride), packed)
        held = c2.workspace_bytes()
        # budget + the buffer's own growth margin (1/8) + per-line arrays, tiles and the ranks parked between launches
        assert held < budget * 1.125 + (4 << 20), (held, budget, whole)
    finally:
        c2.close()


def test_accessor_u32_random_access_evicting_cache(tmp_path):
    """BASELINE configs[4] in small: > 131 072 haplotypes (u32 A_T in header and blocks), multi-allelic
    lines, end-of-vector ("male") samples, missing values and fully haploid lines, random BM positions
    through the accessor with a cache that holds two blocks, against the oracle's reader."""
    import gpu_util as G
    from oracle import oracle
    L = binding.lib()
    rng = np.random.default_rng(20260)
    n, block_len = 66000, 6
    lines = []
    for b in range(5):
        for i in range(block_len):
            if b % 2 == 1 and i % 3 == 1:
                # fully haploid lines only in blocks without multi-allelic lines: the reference writes
                # KEY_LINE_HAPLOID per BCF line and reads it per binary line
                al = (rng.random(n) < 0.2).astype(np.int32)
                lines.append((((al + 1) << 1).astype(np.int32), 2))
            else:
                lines.extend(_random_lines(rng, n, 1, multi=(b % 2 == 0 and i % 2 == 0), eov=(i % 2 == 0),
                                           missing=(i % 5 == 0)))
    dp = oracle.default_phased_of(lines, n)
    ref = oracle.encode_file(lines, n, block_len=block_len, mac_thr=n // 500, default_phased=dp)
    assert ref[14] == 4  # aet_bytes
    path = tmp_path / "u32.xsi"
    path.write_bytes(ref)
    a = ctypes.c_void_p()
    binding.check(L.xsi_accessor_open(ctypes.byref(a), G.ctx().handle, str(path).encode()))
    buf = np.zeros(2 * n, dtype=np.int32)
    assert L.xsi_accessor_fill_genotype_array(a, buf.ctypes.data, buf.size, lines[0][1], 0) > 0
    nb, by = ctypes.c_uint64(0), ctypes.c_uint64(0)
    binding.check(L.xsi_accessor_cache_stats(a, ctypes.byref(nb), ctypes.byref(by), None, None))
    binding.check(L.xsi_accessor_set_cache_bytes(a, 2 * by.value + 4096))
    bms = []
    block = off = 0
    for i, (_, na) in enumerate(lines):
        if i and i % block_len == 0:
            block += 1
            off = 0
        bms.append((block << 15) | off)
        off += na - 1
    order = [int(x) for x in rng.integers(0, len(lines), 60)]
    rd = oracle.Reader(ref)
    expect = {i: rd.fill_genotype_array(lines[i][1], bms[i]) for i in sorted(set(order))}
    cnt = np.zeros(8, dtype=np.uint64)
    for k, i in enumerate(order):
        egt, ecnt = expect[i]
        r = L.xsi_accessor_fill_genotype_array(a, buf.ctypes.data, buf.size, lines[i][1], bms[i])
        assert r == len(egt), "line %d: %s" % (i, L.xsi_hip_last_error())
        assert np.array_equal(buf[:r], egt), "line %d (query %d)" % (i, k)
        assert np.array_equal(buf[:r], lines[i][0]), "line %d vs source" % i
        binding.check(L.xsi_accessor_allele_counts(a, cnt.ctypes.data, lines[i][1]))
        assert np.array_equal(cnt[:lines[i][1]], ecnt)
    hits, misses = ctypes.c_uint64(0), ctypes.c_uint64(0)
    binding.check(L.xsi_accessor_cache_stats(a, ctypes.byref(nb), None, ctypes.byref(hits), ctypes.byref(misses)))
    assert nb.value <= 2 and misses.value > (len(lines) + block_len - 1) // block_len
    L.xsi_accessor_close(a)


def _as_version4(v5):
    """The same file as a version-4 image: u32 block index (accessor_internals_new.hpp:849-869)."""
    io, so = struct.unpack_from("<QQ", v5, 72)
    idx = np.frombuffer(v5, dtype="<u8", count=(so - io) // 8, offset=io)
    assert int(idx.max()) < 2 ** 32
    body = bytearray(v5[:io]) + idx.astype("<u4").tobytes()
    new_so = len(body)
    body += v5[so:]
    struct.pack_into("<I", body, 8, 4)
    struct.pack_into("<Q", body, 80, new_so)
    return bytes(body)


def test_version4_image_decodes(tmp_path):
    """v4 files carry a u32 index; everything else on this path is the v5 layout."""
    import gpu_util as G
    from oracle import oracle
    L = binding.lib()
    n_haps, n_lines, block_len = 1000, 700, 128
    bits = synth.synth_bits(3, 0, n_lines, n_haps)
    stride = synth.row_stride_bytes(n_haps)
    packed = synth.pack_rows(bits, stride)
    p = G.params(n_haps // 2, block_len, 1)
    v4 = _as_version4(G.oracle_file_from_bits(bits, p))
    assert struct.unpack_from("<I", v4, 8)[0] == 4
    out, counts = G.decode_packed(v4, n_haps, stride, n_blocks=(n_lines + block_len - 1) // block_len, max_rows=n_lines)
    assert np.array_equal(out, packed)
    out2, _ = G.decode_packed(v4, n_haps, stride, first_block=3, n_blocks=2, max_rows=2 * block_len)
    assert np.array_equal(out2, packed[3 * block_len:5 * block_len])
    # and through the file-level accessor
    path = tmp_path / "v4.xsi"
    path.write_bytes(v4)
    a = ctypes.c_void_p()
    binding.check(L.xsi_accessor_open(ctypes.byref(a), G.ctx().handle, str(path).encode()))
    gt_all = synth.bits_to_gt(bits, 1)
    buf = np.zeros(n_haps, dtype=np.int32)
    for line in (0, 127, 128, 400, 699, 5):
        bm = ((line // block_len) << 15) | (line % block_len)
        assert L.xsi_accessor_fill_genotype_array(a, buf.ctypes.data, buf.size, 2, bm) == n_haps
        assert np.array_equal(buf, gt_all[line])
    L.xsi_accessor_close(a)


def test_config4_shape_against_oracle(tmp_path):
    """BASELINE configs[4] at its shape: 200 000 haplotypes (u32 A_T), 2048-line blocks, 10 % tri-allelic sites,
    5 % "male" samples with an end-of-vector second value (bench.py --config 4's content, synth.config4_rows_device).
    GPU encode == oracle bytes; xsi_hip_decode_gt rows == source; random accessor queries and their allele counts ==
    the oracle's reader (Accessor::fill_genotype_array with its seek replay, accessor_internals_new.hpp:154-384)."""
    import gpu_util as G
    from oracle import oracle
    torch = G.torch_mod()
    L = binding.lib()
    n_haps, block_len, n_lines, thr = 200000, 2048, 4096, 200
    n = n_haps // 2
    rows_d, nal = synth.config4_rows_device(L, G.ctx(), torch, torch.device("cuda"), 45, 0, n_lines, n_haps, 1)
    rows = rows_d.cpu().numpy()
    assert (nal == 3).sum() == 410 and (rows[:, 2 * 7 + 1] == synth.INT32_VECTOR_END).all()
    lines = [(rows[i], int(nal[i])) for i in range(n_lines)]
    ref = oracle.encode_file(lines, n, block_len=block_len, mac_thr=thr, default_phased=1)
    assert ref[14] == 4  # u32 A_T
    p = G.params(n, block_len, thr, 1)
    n_bin = int((nal.astype(np.int64) - 1).sum())
    cap = int(L.xsi_hip_encode_gt_bound(ctypes.byref(p), n_lines, n_bin))
    d_out = G.dev_empty(cap)
    d_off = torch.zeros(2, dtype=torch.int64, device="cuda")
    res = binding.EncodeResult()
    ngt = np.full(n_lines, n_haps, dtype=np.uint32)
    binding.check(L.xsi_hip_encode_gt(G.ctx().handle, ctypes.byref(p), rows_d.data_ptr(), n_haps, n_lines, ngt.ctypes.data,
                                      nal.ctypes.data, d_out.data_ptr(), cap, d_off.data_ptr(), ctypes.byref(res)))
    region = d_out[:res.blocks_bytes].cpu().numpy().tobytes()
    del d_out
    got = G.assemble_file(region, d_off.cpu().numpy().astype(np.uint64), p, n_lines, n_bin, ["S%d" % i for i in range(n)])
    io = struct.unpack_from("<Q", ref, 72)[0]
    assert got[256:io] == ref[256:io], "blocks region differs from the oracle's"
    assert got == ref
    # whole-file decode on the device against the source rows (compared on the device)
    d_file = G.dev_u8(np.frombuffer(got, dtype=np.uint8))
    d_dec = torch.zeros((n_lines, n_haps), dtype=torch.int32, device="cuda")
    d_cnt = torch.zeros((n_lines, 3), dtype=torch.int64, device="cuda")
    lngt = np.zeros(n_lines, dtype=np.uint32)
    binding.check(L.xsi_hip_decode_gt(G.ctx().handle, d_file.data_ptr(), len(got), 0, 2, nal.ctypes.data, n_lines,
                                      d_dec.data_ptr(), n_haps, lngt.ctypes.data, d_cnt.data_ptr(), 3))
    assert (lngt == n_haps).all()
    assert bool(torch.equal(d_dec, rows_d))
    del d_dec
    # random access through the accessor against the oracle's reader
    path = tmp_path / "c4.xsi"
    path.write_bytes(got)
    a = ctypes.c_void_p()
    binding.check(L.xsi_accessor_open(ctypes.byref(a), G.ctx().handle, str(path).encode()))
    bm = synth.bm_positions(nal, block_len)
    rng = np.random.default_rng(45)
    rd = oracle.Reader(ref)
    buf = np.zeros(n_haps, dtype=np.int32)
    pbuf = ctypes.c_void_p(buf.ctypes.data)
    nout = ctypes.c_int(0)
    cnt = np.zeros(4, dtype=np.uint64)
    picks = [int(x) for x in rng.integers(0, n_lines, 40)] + [3, 13, 2047, 2048, 2061, n_lines - 1]
    for i in picks:
        egt, ecnt = rd.fill_genotype_array(int(nal[i]), int(bm[i]))
        r = L.xsi_accessor_get_genotypes(a, int(nal[i]), int(bm[i]), ctypes.byref(pbuf), ctypes.byref(nout))
        assert r == n_haps and nout.value == n_haps, "line %d: %s" % (i, L.xsi_hip_last_error())
        assert np.array_equal(buf, egt), "line %d vs oracle reader" % i
        assert np.array_equal(buf, rows[i]), "line %d vs source" % i
        binding.check(L.xsi_accessor_allele_counts(a, cnt.ctypes.data, int(nal[i])))
        assert np.array_equal(cnt[:int(nal[i])], ecnt)
    # first touches stopped at the requested line (prefix decode) and later queries further in continued from there
    pd, ext = ctypes.c_uint64(0), ctypes.c_uint64(0)
    binding.check(L.xsi_accessor_prefix_stats(a, ctypes.byref(pd), ctypes.byref(ext)))
    assert pd.value >= 1 and ext.value >= 1, (pd.value, ext.value)
    # a REGISTERED page-locked array (xsi_accessor_alloc_array + xsi_accessor_register_array): the compose kernel stores
    # single lines into it: the same line twice, a sequential run (which goes back through the window), another
    # (unregistered, pageable) array in between, then the first again; a too-small array and pageable memory are refused
    small = np.zeros(n_haps - 2, dtype=np.int32)
    assert L.xsi_accessor_register_array(a, small.ctypes.data, small.size) == binding.XSI_ERR_CAPACITY
    assert L.xsi_accessor_register_array(a, buf.ctypes.data, buf.size) == binding.XSI_ERR_ARG  # pageable: not taken
    assert b"pageable" in L.xsi_hip_last_error()
    pbuf = G.accessor_array(a, (n_haps,))
    binding.check(L.xsi_accessor_register_array(a, pbuf.ctypes.data, pbuf.size))
    buf2 = np.zeros(n_haps, dtype=np.int32)
    seq = [77, 77, 78, 79, 80, 81, 77, 3000, 3000]
    for k, i in enumerate(seq):
        dst = buf2 if k == 4 else pbuf
        dst[:] = -5
        assert L.xsi_accessor_fill_genotype_array(a, dst.ctypes.data, dst.size, int(nal[i]), int(bm[i])) == n_haps
        assert np.array_equal(dst, rows[i]), "step %d line %d" % (k, i)
    binding.check(L.xsi_accessor_unregister_array(a))
    G.accessor_array_free(a, pbuf)
    # batched queries at this shape: 48 random lines of all blocks in one call, into a registered 2-D array
    bq = np.asarray([int(x) for x in rng.integers(0, n_lines, 48)], dtype=np.int64)
    b_na = np.ascontiguousarray(nal[bq], dtype=np.uint32)
    b_bm = np.ascontiguousarray(bm[bq], dtype=np.uint64)
    brow = G.accessor_array(a, (len(bq), n_haps), -3)
    binding.check(L.xsi_accessor_register_array(a, brow.ctypes.data, brow.size))
    tot = L.xsi_accessor_get_genotypes_batch(a, len(bq), b_na.ctypes.data, b_bm.ctypes.data, brow.ctypes.data, n_haps, None)
    assert tot == len(bq) * n_haps, L.xsi_hip_last_error()
    for k, i in enumerate(bq):
        assert np.array_equal(brow[k], rows[int(i)]), "batched query %d (line %d)" % (k, int(i))
    binding.check(L.xsi_accessor_unregister_array(a))
    buf[:] = -5   # pageable memory: the ordinary path
    assert L.xsi_accessor_fill_genotype_array(a, buf.ctypes.data, buf.size, int(nal[77]), int(bm[77])) == n_haps
    assert np.array_equal(buf, rows[77])
    L.xsi_accessor_close(a)  # frees brow (an array of xsi_accessor_alloc_array the caller did not free)


@pytest.mark.parametrize("seed", [1, 2, 3, 4, 5, 6])
def test_phased_decode_on_random_shapes(seed, monkeypatch):
    """The phased decode (line ranges with a ramp in front, the boundary scan in two parts, ranks parked between the
    chain launches) on random geometries: blocks of unequal length (the last one short, down to a handful of lines),
    range counts that do not divide anything, rows from 16 384 to 65 536 haplotypes: rows == input, file == oracle."""
    import gpu_util as G
    rng = np.random.default_rng(1000 + seed)
    n_haps = int(rng.integers(8192, 32768)) * 2
    block_len = int(rng.integers(700, 1500))
    n_blocks = int(rng.integers(2, 5))
    n_lines = (n_blocks - 1) * block_len + int(rng.integers(3, block_len))
    phases = int(rng.integers(2, 33))
    thr = int(rng.integers(0, 40))
    monkeypatch.setenv("XSI_RANK_WG_MIN_BLOCKS", "1")
    monkeypatch.setenv("XSI_DEC_PHASES", str(phases))
    bits, packed, stride = _device_synth(n_haps, n_lines, 100 + seed)
    p = G.params(n_haps // 2, block_len, thr)
    ref = G.oracle_file_from_bits(bits, p)
    region, offsets, res = G.encode_packed(packed, n_haps, p)
    assert res.n_wah_lines >= 256 * n_blocks  # what decode_planes asks for before it cuts the lines into ranges
    got = G.assemble_file(region, offsets, p, n_lines, n_lines, ["S%d" % i for i in range(n_haps // 2)])
    assert got == ref, (n_haps, block_len, n_lines, phases)
    out, counts = G.decode_packed(got, n_haps, stride)
    assert np.array_equal(out, packed), (n_haps, block_len, n_lines, phases)
    assert np.array_equal(counts, bits.sum(1).astype(np.int32))
