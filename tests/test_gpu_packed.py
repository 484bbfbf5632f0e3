"""GPU parity tests of the packed (bi-allelic, diploid, fully called) path, through the C ABI.

Each stage is checked against an independent reference so a mismatch is localised in one run:
synthetic generator vs numpy mirror, PBWT chain vs numpy, encoded .xsi bytes vs the CPU oracle
(bit-exact), decode round trip vs the input.
"""
import ctypes
import hashlib
import struct

import numpy as np
import pytest

from xsqueezeit_amd import binding, synth

pytestmark = pytest.mark.gpu


def _mk(n_haps, n_lines, seed):
    bits = synth.synth_bits(seed, 0, n_lines, n_haps)
    stride = synth.row_stride_bytes(n_haps)
    return bits, synth.pack_rows(bits, stride), stride


def test_synth_matches_numpy():
    import gpu_util as G
    torch = G.torch_mod()
    for n_haps, n_lines, first in ((5008, 300, 0), (20, 50, 7), (70001, 20, 12345)):
        stride = synth.row_stride_bytes(n_haps)
        d = G.dev_empty(n_lines * stride)
        binding.check(binding.lib().xsi_hip_synth_packed(G.ctx().handle, 42, first, n_lines, n_haps, d.data_ptr(),
                                                         stride))
        torch.cuda.synchronize()
        got = d.cpu().numpy().reshape(n_lines, stride)
        exp = synth.pack_rows(synth.synth_bits(42, first, n_lines, n_haps), stride)
        assert np.array_equal(got, exp)


@pytest.mark.parametrize("n_haps,n_lines,block_len,thr", [
    (20, 64, 16, 0),
    (200, 300, 128, 0),
    (1000, 300, 100, 1),
    (5008, 600, 256, 5),
    (16390, 80, 64, 16),
    (64976, 24, 16, 64),
    (65534, 12, 8, 65),       # 32767 samples: last size below the A_T mismatch window
    (12000, 260, 130, 12),    # 12 chunks per wave: segment pre-pass with two nibble words per thread
    (16384, 200, 100, 16),    # 16 chunks per wave, exact capacity
    (65410, 120, 60, 65),     # 64 chunks per wave with padding members, long history
    (131074, 200, 100, 131),
    (200000, 64, 64, 200),
])
def test_chain_matches_numpy(n_haps, n_lines, block_len, thr):
    import gpu_util as G
    torch = G.torch_mod()
    bits, packed, stride = _mk(n_haps, n_lines, 100 + n_haps)
    p = G.params(n_haps // 2, block_len, thr)
    ys = (n_haps + 63) // 64 * 8
    d_bits = G.dev_u8(packed)
    d_y = G.dev_empty(n_lines * ys)
    d_y.zero_()
    d_kind = G.dev_empty(n_lines + 64)
    nw = ctypes.c_uint64(0)
    binding.check(binding.lib().xsi_hip_debug_chain_encode(G.ctx().handle, ctypes.byref(p), d_bits.data_ptr(), n_lines,
                                                           stride, d_y.data_ptr(), ys, d_kind.data_ptr(),
                                                           ctypes.byref(nw)))
    ref = G.numpy_chain_yrows(bits, block_len, thr)
    assert nw.value == len(ref)
    kinds = d_kind.cpu().numpy()[:n_lines]
    exp_wah = np.zeros(n_lines, dtype=bool)
    for l, _ in ref:
        exp_wah[l] = True
    assert np.array_equal((kinds & 1).astype(bool), exp_wah)
    y = d_y.cpu().numpy().reshape(n_lines, ys)
    for j, (l, yb) in enumerate(ref):
        got = np.unpackbits(y[j], bitorder="little")[:n_haps]
        assert np.array_equal(got, yb), "WAH line %d (BCF line %d) permuted bits differ" % (j, l)


@pytest.mark.parametrize("n_haps,n_lines,block_len,thr", [
    (20, 64, 16, 0),
    (200, 700, 128, 0),
    (1000, 300, 100, 1),
    (5008, 20000, 8192, 5),
    (5008, 700, 64, 0),
    (16390, 200, 64, 16),
    (64976, 40, 16, 64),
    (65534, 12, 8, 65),      # 32767 samples: largest count below the A_T mismatch window
    (131074, 12, 8, 131),    # 65537 samples: u32 A_T everywhere, global-memory chain
    (200000, 10, 8, 200),
    (10000, 520, 260, 10),    # 10 chunks per wave, chain cut into line segments
    (140000, 260, 128, 140),  # streaming encode chain + LDS-staged long-row decode chain, long blocks
    (530000, 24, 12, 530),    # > 16384 rank-select pairs per row: deepest prefetch variant; above the multi-workgroup encode
    (140000, 390, 3, 140),    # 130 blocks over 80 groups of 3 workgroups: the groups' persistent walk over the blocks
])
def test_encode_bit_exact_and_roundtrip(n_haps, n_lines, block_len, thr):
    import gpu_util as G
    bits, packed, stride = _mk(n_haps, n_lines, 7 + n_haps)
    p = G.params(n_haps // 2, block_len, thr)
    names = ["S%d" % i for i in range(n_haps // 2)]
    ref = G.oracle_file_from_bits(bits, p, names)
    region, offsets, res = G.encode_packed(packed, n_haps, p)
    got = G.assemble_file(region, offsets, p, n_lines, n_lines, names)
    io = struct.unpack_from("<Q", ref, 72)[0]
    # localise: header, index, then blocks
    assert got[:256] == ref[:256]
    assert got[io:] == ref[io:], "index / sample names differ"
    if got != ref:
        first = next(i for i in range(min(len(got), len(ref))) if got[i] != ref[i])
        raise AssertionError("blocks region differs at file offset %d (sizes %d vs %d)" % (first, len(got), len(ref)))
    assert hashlib.sha256(got).hexdigest() == hashlib.sha256(ref).hexdigest()
    # decode what we encoded
    out, counts = G.decode_packed(got, n_haps, stride)
    assert out.shape[0] == n_lines
    assert np.array_equal(out, packed)
    assert np.array_equal(counts, bits.sum(1).astype(np.int32))


def test_decode_reference_written_file():
    """Decode a file produced by the oracle (i.e. what the reference writes), not by us."""
    import gpu_util as G
    n_haps, n_lines = 5008, 3000
    bits, packed, stride = _mk(n_haps, n_lines, 99)
    p = G.params(n_haps // 2, 1024, 5)
    ref = G.oracle_file_from_bits(bits, p)
    out, counts = G.decode_packed(ref, n_haps, stride)
    assert np.array_equal(out, packed)
    # a sub-range of blocks
    out2, _ = G.decode_packed(ref, n_haps, stride, first_block=1, n_blocks=1, max_rows=1024)
    assert np.array_equal(out2, packed[1024:2048])


def test_edge_rows_all_zero_all_one_saturated_runs():
    """Monomorphic lines, all-ALT lines, and runs past the 16383-group fill saturation
    (needs > 245745 haplotypes)."""
    import gpu_util as G
    n_haps, n_lines = 300000, 12
    bits = np.zeros((n_lines, n_haps), dtype=np.uint8)
    bits[1] = 1
    bits[2, : n_haps // 2] = 1            # sorts the haplotypes: second half zeros first
    bits[3, -1] = 1
    bits[4, 1::2] = 1
    bits[5, : n_haps // 2] = 1
    bits[6, 5] = 1
    bits[7] = 1
    bits[7, 7] = 0
    bits[8, 1000:260000] = 1
    rng = np.random.default_rng(5)
    bits[9] = rng.random(n_haps) < 0.5
    bits[10] = rng.random(n_haps) < 0.01
    bits[11] = bits[9]
    stride = synth.row_stride_bytes(n_haps)
    packed = synth.pack_rows(bits, stride)
    p = G.params(n_haps // 2, 8, 300)
    ref = G.oracle_file_from_bits(bits, p)
    region, offsets, res = G.encode_packed(packed, n_haps, p)
    names = ["S%d" % i for i in range(n_haps // 2)]
    got = G.assemble_file(region, offsets, p, n_lines, n_lines, names)
    assert got == ref
    out, counts = G.decode_packed(got, n_haps, stride)
    assert np.array_equal(out, packed)


def test_capacity_error_is_reported():
    import gpu_util as G
    torch = G.torch_mod()
    bits, packed, stride = _mk(5008, 300, 3)
    p = G.params(2504, 128, 5)
    d_bits = G.dev_u8(packed)
    d_out = G.dev_empty(1024)
    d_off = torch.zeros(3, dtype=torch.int64, device="cuda")
    res = binding.EncodeResult()
    rc = binding.lib().xsi_hip_encode_packed(G.ctx().handle, ctypes.byref(p), d_bits.data_ptr(), 300, stride,
                                             d_out.data_ptr(), 1024, d_off.data_ptr(), ctypes.byref(res))
    assert rc == binding.XSI_ERR_CAPACITY


@pytest.mark.parametrize("n_haps,n_lines,block_len,thr", [
    (65534, 20, 8, 65),     # largest count with u16 A_T in header AND blocks (32767 samples)
    (131072, 10, 8, 131),   # 65536 samples: first size with u32 A_T everywhere
    (524288, 16, 8, 524),   # largest size of the multi-workgroup encode chain: 8 workgroups, 16384-entry table
    (64, 200, 50, 0),       # exactly one wave chunk
    (66, 200, 50, 0),       # one sample past a chunk boundary
    (4096, 300, 128, 4),    # exact multiple of the chain capacity (no padding members)
])
def test_size_boundaries(n_haps, n_lines, block_len, thr):
    import gpu_util as G
    bits, packed, stride = _mk(n_haps, n_lines, 1000 + n_haps)
    p = G.params(n_haps // 2, block_len, thr)
    names = ["S%d" % i for i in range(n_haps // 2)]
    ref = G.oracle_file_from_bits(bits, p, names)
    region, offsets, res = G.encode_packed(packed, n_haps, p)
    got = G.assemble_file(region, offsets, p, n_lines, n_lines, names)
    assert got == ref
    out, counts = G.decode_packed(got, n_haps, stride)
    assert np.array_equal(out, packed)


@pytest.mark.parametrize("n_samples", [32768, 35001, 40000, 65535])
def test_at_mismatch_window_is_refused(n_samples, tmp_path):
    """32768 <= n_samples <= 65535: the reference writes 16-bit block data (with a prefix array that
    wraps modulo 65536, gt_block.hpp:171,179) under a header that says 32-bit A_T and cannot decode
    it; every encode entry point returns XSI_ERR_UNSUPPORTED instead of writing such a file."""
    import gpu_util as G
    torch = G.torch_mod()
    L = binding.lib()
    n_haps = 2 * n_samples
    stride = synth.row_stride_bytes(n_haps)
    p = G.params(n_samples, 8, 10)
    d_bits = G.dev_empty(4 * stride)
    d_bits.zero_()
    d_out = G.dev_empty(1 << 20)
    d_off = torch.zeros(1, dtype=torch.int64, device="cuda")
    res = binding.EncodeResult()
    rc = L.xsi_hip_encode_packed(G.ctx().handle, ctypes.byref(p), d_bits.data_ptr(), 4, stride, d_out.data_ptr(),
                                 1 << 20, d_off.data_ptr(), ctypes.byref(res))
    assert rc == binding.XSI_ERR_UNSUPPORTED
    assert b"mismatch window" in L.xsi_hip_last_error()
    d_gt = torch.full((2, n_haps), 2, dtype=torch.int32, device="cuda")
    ngt = np.full(2, n_haps, dtype=np.uint32)
    nal = np.full(2, 2, dtype=np.uint32)
    rc = L.xsi_hip_encode_gt(G.ctx().handle, ctypes.byref(p), d_gt.data_ptr(), n_haps, 2, ngt.ctypes.data,
                             nal.ctypes.data, d_out.data_ptr(), 1 << 20, d_off.data_ptr(), ctypes.byref(res))
    assert rc == binding.XSI_ERR_UNSUPPORTED
    w = ctypes.c_void_p()
    rc = L.xsi_writer_open(ctypes.byref(w), G.ctx().handle, str(tmp_path / "w.xsi").encode(), ctypes.byref(p), None)
    assert rc == binding.XSI_ERR_UNSUPPORTED and not w.value
    from oracle import oracle
    with pytest.raises(ValueError):
        oracle.Writer(n_samples)


def test_all_sparse_and_all_wah_blocks():
    """Blocks whose lines are all sparse (no PBWT work at all, chain kernels exit early) and
    blocks whose lines are all WAH; plus a ragged last block of a single line."""
    import gpu_util as G
    n_haps = 5008
    rng = np.random.default_rng(9)
    rows = []
    for i in range(64):   # block 0: monomorphic / singletons only -> all sparse
        r = np.zeros(n_haps, np.uint8)
        if i % 3:
            r[rng.integers(0, n_haps, size=i % 4)] = 1
        rows.append(r)
    for i in range(64):   # block 1: common variants -> all WAH
        rows.append((rng.random(n_haps) < 0.3 + 0.005 * i).astype(np.uint8))
    for i in range(64):   # block 2: all-ALT and near-all-ALT lines (negated sparse)
        r = np.ones(n_haps, np.uint8)
        r[rng.integers(0, n_haps, size=i % 5)] = 0
        rows.append(r)
    rows.append((rng.random(n_haps) < 0.5).astype(np.uint8))  # block 3: one line
    bits = np.stack(rows)
    stride = synth.row_stride_bytes(n_haps)
    packed = synth.pack_rows(bits, stride)
    p = G.params(n_haps // 2, 64, 5)
    names = ["S%d" % i for i in range(n_haps // 2)]
    ref = G.oracle_file_from_bits(bits, p, names)
    region, offsets, res = G.encode_packed(packed, n_haps, p)
    assert res.n_blocks == 4
    got = G.assemble_file(region, offsets, p, len(rows), len(rows), names)
    assert got == ref
    out, counts = G.decode_packed(got, n_haps, stride)
    assert np.array_equal(out, packed)
    assert np.array_equal(counts, bits.sum(1).astype(np.int32))


def test_argument_errors():
    import gpu_util as G
    torch = G.torch_mod()
    L = binding.lib()
    bits, packed, stride = _mk(200, 10, 1)
    d_bits = G.dev_u8(packed)
    d_out = G.dev_empty(1 << 16)
    d_off = torch.zeros(4, dtype=torch.int64, device="cuda")
    res = binding.EncodeResult()
    p = G.params(100, 8, 0)
    h = G.ctx().handle
    # zero lines, bad stride, block_len beyond the BM offset range
    assert L.xsi_hip_encode_packed(h, ctypes.byref(p), d_bits.data_ptr(), 0, stride, d_out.data_ptr(), 1 << 16,
                                   d_off.data_ptr(), ctypes.byref(res)) == binding.XSI_ERR_ARG
    assert L.xsi_hip_encode_packed(h, ctypes.byref(p), d_bits.data_ptr(), 10, 12, d_out.data_ptr(), 1 << 16,
                                   d_off.data_ptr(), ctypes.byref(res)) == binding.XSI_ERR_ARG
    pbad = G.params(100, 40000, 0)
    assert L.xsi_hip_encode_packed(h, ctypes.byref(pbad), d_bits.data_ptr(), 10, stride, d_out.data_ptr(), 1 << 16,
                                   d_off.data_ptr(), ctypes.byref(res)) == binding.XSI_ERR_ARG
    # decode: garbage image, truncated image, block range beyond the file
    junk = G.dev_u8(np.zeros(4096, np.uint8))
    rows = ctypes.c_uint64(0)
    assert L.xsi_hip_decode_packed(h, junk.data_ptr(), 4096, 0, 1, d_out.data_ptr(), stride, 10, ctypes.byref(rows),
                                   None) == binding.XSI_ERR_FORMAT
    assert L.xsi_hip_decode_packed(h, junk.data_ptr(), 100, 0, 1, d_out.data_ptr(), stride, 10, ctypes.byref(rows),
                                   None) == binding.XSI_ERR_FORMAT
    ref = G.oracle_file_from_bits(bits, p)
    d_file = G.dev_u8(np.frombuffer(ref, np.uint8))
    assert L.xsi_hip_decode_packed(h, d_file.data_ptr(), len(ref), 1, 5, d_out.data_ptr(), stride, 100,
                                   ctypes.byref(rows), None) == binding.XSI_ERR_ARG
    # row capacity too small
    assert L.xsi_hip_decode_packed(h, d_file.data_ptr(), len(ref), 0, 2, d_out.data_ptr(), stride, 3,
                                   ctypes.byref(rows), None) == binding.XSI_ERR_CAPACITY


@pytest.mark.gpu
def test_corrupt_images_are_rejected_not_dereferenced():
    """A damaged index or dictionary must come back as XSI_ERR_FORMAT: the parser checks every offset
    against the image before anything is read through it."""
    import gpu_util as G
    n_haps, n_lines = 512, 300
    bits, packed, stride = _mk(n_haps, n_lines, 77)
    p = G.params(n_haps // 2, 100, 1)
    good = bytearray(G.oracle_file_from_bits(bits, p))
    io = struct.unpack_from("<Q", good, 72)[0]
    blk0 = struct.unpack_from("<Q", good, io)[0]
    n_outer = struct.unpack_from("<I", good, blk0 + 4)[0]
    gt_rel_at = None
    for i in range(n_outer):
        if struct.unpack_from("<I", good, blk0 + 8 + 8 * i)[0] == 256:
            gt_rel_at = blk0 + 12 + 8 * i
    gt0 = blk0 + struct.unpack_from("<I", good, gt_rel_at)[0]
    n_keys = struct.unpack_from("<I", good, gt0 + 4)[0]
    key_at = {struct.unpack_from("<I", good, gt0 + 8 + 8 * i)[0]: gt0 + 12 + 8 * i for i in range(n_keys)}

    def decode(img):
        return G.decode_packed(bytes(img), n_haps, stride)

    out, _ = decode(good)
    assert np.array_equal(out, packed)
    cases = {
        "index entry beyond the image": (io, struct.pack("<Q", len(good) + 4096)),
        "misaligned index entry": (io, struct.pack("<Q", blk0 + 2)),
        "GT block offset beyond the image": (gt_rel_at, struct.pack("<I", 0x7FFFFFF0)),
        "sparse matrix offset beyond the image": (key_at[0x21], struct.pack("<I", 0x7FFFFF00)),
        "WAH matrix offset after the sparse matrix": (key_at[0x20], struct.pack("<I", struct.unpack_from("<I", good, key_at[0x21])[0] + 64)),
        "more binary lines than a block can hold": (key_at[0x1], struct.pack("<I", 40000)),
    }
    for name, (at, val) in cases.items():
        bad = bytearray(good)
        bad[at:at + len(val)] = val
        with pytest.raises(binding.XsiError) as ei:
            decode(bad)
        assert ei.value.code == -4, "%s: %s" % (name, ei.value)  # XSI_ERR_FORMAT


@pytest.mark.parametrize("n_haps", [12320, 15360, 16384, 30000, 65534, 131072, 140000])
def test_wah_unit_encoder_on_adversarial_rows(n_haps):
    """The unit WAH16 encoder (32 groups = 480 bits per lane, heads by bit logic, fill counts from the distance to
    the next head) on rows built to sit on its seams: runs that start or end exactly at group (15), word (32) and
    unit (480) boundaries, runs one bit short or long of them, alternating fills, literals at the ends of a unit,
    rows that end inside a group.  Every line is a block of its own, so the permuted row IS the input row."""
    import gpu_util as G
    rng = np.random.default_rng(n_haps)
    rows = []
    def row_from_runs(runs):  # [(value, length)] repeated to fill the row
        out = np.zeros(n_haps, dtype=np.uint8)
        pos, k = 0, 0
        while pos < n_haps:
            v, ln = runs[k % len(runs)]
            out[pos:pos + ln] = v
            pos += ln
            k += 1
        return out
    for a in (15, 30, 32, 480, 465, 495, 479, 481, 960, 7, 16):
        for b in (15, 1, 480, 14, 16, 31):
            rows.append(row_from_runs([(1, a), (0, b)]))
            rows.append(row_from_runs([(0, a), (1, b)]))
    for start in (0, 14, 15, 479, 480, 481, 959, n_haps - 16, n_haps - 15, n_haps - 1):
        r = np.zeros(n_haps, dtype=np.uint8)
        r[start:start + 1] = 1                      # a lone literal at a seam
        rows.append(r)
        r = np.ones(n_haps, dtype=np.uint8)
        r[start:start + 1] = 0
        r[n_haps // 2] = 0                          # keep the minor count above 0 on both sides
        rows.append(r)
    r = np.zeros(n_haps, dtype=np.uint8); r[:n_haps // 3] = 1; rows.append(r)      # two long runs
    r = np.zeros(n_haps, dtype=np.uint8); r[1::2] = 1; rows.append(r)              # all literals
    r = (rng.random(n_haps) < 0.5).astype(np.uint8); rows.append(r)
    r = np.zeros(n_haps, dtype=np.uint8); r[480 * 3:480 * 5] = 1; rows.append(r)   # a run of whole units
    bits = np.stack(rows)
    stride = synth.row_stride_bytes(n_haps)
    packed = synth.pack_rows(bits, stride)
    p = G.params(n_haps // 2, 1, 0)   # block_len 1, MAC threshold 0: every polymorphic line is a WAH line
    names = ["S%d" % i for i in range(n_haps // 2)]
    ref = G.oracle_file_from_bits(bits, p, names)
    region, offsets, res = G.encode_packed(packed, n_haps, p)
    assert res.n_wah_lines == len(rows)
    got = G.assemble_file(region, offsets, p, len(rows), len(rows), names)
    if got != ref:
        first = next(i for i in range(min(len(got), len(ref))) if got[i] != ref[i])
        raise AssertionError("file differs at offset %d (sizes %d vs %d)" % (first, len(got), len(ref)))
    out, _ = G.decode_packed(got, n_haps, stride)
    assert np.array_equal(out, packed)


@pytest.mark.parametrize("n_haps", [32, 482, 962, 1442, 5008, 7682, 12224])
def test_small_row_unit_encoder_on_adversarial_rows(n_haps):
    """Rows short enough for several lines per wave (k_wah_units_small: lane = (line, unit), heads counted per line by a
    segmented scan, units 1 .. 26 per line here): runs on the group / word / unit seams, lone literals at the first
    and last positions of a line and of a unit, neighbouring lines with very different head counts (the per-line
    offsets come from the lane before the line), lines that are not WAH lines in between (they are skipped: the
    lines of a wave are then not consecutive binary lines)."""
    import gpu_util as G
    rng = np.random.default_rng(n_haps)
    rows = []

    def row_from_runs(runs):
        out = np.zeros(n_haps, dtype=np.uint8)
        pos, k = 0, 0
        while pos < n_haps:
            v, ln = runs[k % len(runs)]
            out[pos:pos + ln] = v
            pos += ln
            k += 1
        return out
    for a in (15, 30, 32, 480, 465, 479, 481, 7):
        for b in (15, 1, 480, 14, 31):
            rows.append(row_from_runs([(1, a), (0, b)]))
            rows.append(row_from_runs([(0, a), (1, b)]))
    for start in (0, 14, 15, 479, 480, 481, n_haps - 16, n_haps - 15, n_haps - 1):
        if 0 <= start < n_haps:
            r = np.zeros(n_haps, dtype=np.uint8)
            r[start] = 1
            rows.append(r)                          # (a singleton: sparse at threshold 0? no - minor count 1 > 0: WAH)
            r = np.ones(n_haps, dtype=np.uint8)
            r[start] = 0
            rows.append(r)
    rows.append(np.zeros(n_haps, dtype=np.uint8))   # monomorphic: not a WAH line, sits between WAH lines
    r = np.zeros(n_haps, dtype=np.uint8); r[1::2] = 1; rows.append(r)              # all literals next to ...
    r = np.zeros(n_haps, dtype=np.uint8); r[:n_haps // 2] = 1; rows.append(r)      # ... two words
    rows.append(np.ones(n_haps, dtype=np.uint8))    # monomorphic again
    for _ in range(12):
        rows.append((rng.random(n_haps) < rng.choice([0.01, 0.3, 0.5, 0.97])).astype(np.uint8))
    order = rng.permutation(len(rows))
    bits = np.stack([rows[i] for i in order])
    stride = synth.row_stride_bytes(n_haps)
    packed = synth.pack_rows(bits, stride)
    for block_len in (1, 7):
        p = G.params(n_haps // 2, block_len, 0)
        names = ["S%d" % i for i in range(n_haps // 2)]
        ref = G.oracle_file_from_bits(bits, p, names)
        region, offsets, res = G.encode_packed(packed, n_haps, p)
        got = G.assemble_file(region, offsets, p, len(rows), len(rows), names)
        if got != ref:
            first = next(i for i in range(min(len(got), len(ref))) if got[i] != ref[i])
            raise AssertionError("block_len %d: file differs at offset %d (sizes %d vs %d)" % (block_len, first, len(got), len(ref)))
        out, _ = G.decode_packed(got, n_haps, stride)
        assert np.array_equal(out, packed)


def test_sparse_lists_on_the_main_stream(monkeypatch):
    """The sparse lists are normally written into a scratch underneath the chain (side stream) and moved into place;
    when that scratch would be too large - or with XSI_NO_SPARSE_OVERLAP - they are written in place after the layout.
    Both orders must give the same file."""
    import gpu_util as G
    n_haps, n_lines, block_len, thr = 5008, 900, 256, 40
    bits, packed, stride = _mk(n_haps, n_lines, 4242)
    p = G.params(n_haps // 2, block_len, thr)
    names = ["S%d" % i for i in range(n_haps // 2)]
    ref = G.oracle_file_from_bits(bits, p, names)
    for env in (None, "1"):
        if env:
            monkeypatch.setenv("XSI_NO_SPARSE_OVERLAP", env)
        region, offsets, res = G.encode_packed(packed, n_haps, p)
        assert res.n_wah_lines < n_lines            # there are sparse lines
        got = G.assemble_file(region, offsets, p, n_lines, n_lines, names)
        assert got == ref, "sparse overlap %s" % ("off" if env else "on")


@pytest.mark.parametrize("n_haps,n_lines,block_len,thr", [(5008, 600, 256, 5), (64976, 40, 16, 64), (200000, 48, 24, 200)])
def test_counts_from_the_producer(n_haps, n_lines, block_len, thr, monkeypatch):
    """xsi_hip_encode_packed_counted: with the rows' ALT counts supplied (xsi_hip_count_packed_rows here, the packer's
    popcounts in the writer) the bytes are those of xsi_hip_encode_packed; XSI_CHECK_ROW_COUNTS=1 recounts and
    refuses counts that do not belong to the rows."""
    import gpu_util as G
    torch = G.torch_mod()
    L = binding.lib()
    bits, packed, stride = _mk(n_haps, n_lines, 900 + n_haps)
    p = G.params(n_haps // 2, block_len, thr)
    ref_region, ref_off, ref_res = G.encode_packed(packed, n_haps, p)
    d_bits = G.dev_u8(packed)
    d_cnt = torch.zeros(n_lines, dtype=torch.int32, device="cuda")
    binding.check(L.xsi_hip_count_packed_rows(G.ctx().handle, d_bits.data_ptr(), n_lines, stride, n_haps, d_cnt.data_ptr()))
    torch.cuda.synchronize()
    assert np.array_equal(d_cnt.cpu().numpy(), bits.sum(axis=1).astype(np.int32))
    cap = int(L.xsi_hip_encode_bound(ctypes.byref(p), n_lines, n_lines))
    d_out = G.dev_empty(cap)
    n_blocks = (n_lines + block_len - 1) // block_len
    d_off = torch.zeros(n_blocks, dtype=torch.int64, device="cuda")
    res = binding.EncodeResult()

    def run(cnt):
        return L.xsi_hip_encode_packed_counted(G.ctx().handle, ctypes.byref(p), d_bits.data_ptr(), n_lines, stride,
                                               cnt.data_ptr(), d_out.data_ptr(), cap, d_off.data_ptr(), ctypes.byref(res))

    monkeypatch.setenv("XSI_CHECK_ROW_COUNTS", "1")
    binding.check(run(d_cnt))
    assert d_out[:res.blocks_bytes].cpu().numpy().tobytes() == ref_region
    assert np.array_equal(d_off.cpu().numpy().astype(np.uint64), ref_off)
    assert res.n_wah_lines == ref_res.n_wah_lines
    monkeypatch.delenv("XSI_CHECK_ROW_COUNTS")
    d_out.zero_()
    binding.check(run(d_cnt))
    assert d_out[:res.blocks_bytes].cpu().numpy().tobytes() == ref_region
    wrong = d_cnt.clone()
    wrong[n_lines // 2] += 1
    monkeypatch.setenv("XSI_CHECK_ROW_COUNTS", "1")
    assert run(wrong) == binding.XSI_ERR_ARG
    assert b"row counts" in L.xsi_hip_last_error()


@pytest.mark.parametrize("n_haps,n_lines,block_len", [(64976, 24, 12), (20000, 40, 20), (131074, 12, 6), (300000, 8, 4)])
def test_incompressible_lines_and_words_kept_in_the_row(n_haps, n_lines, block_len):
    """The WAH sizing pass leaves a line's words in the line's own permuted row when they fit (the writing pass then
    only moves them); a line of (almost) only literal groups takes more bytes than its row (16/15) and is encoded from
    the row again.  Half the lines here are coin flips (every group a literal), the others synthetic; bytes against
    the oracle."""
    import gpu_util as G
    rng = np.random.default_rng(n_haps)
    bits = synth.synth_bits(5, 0, n_lines, n_haps)
    for l in range(0, n_lines, 2):
        bits[l] = rng.integers(0, 2, n_haps, dtype=np.uint8)
    bits[3] = 0
    bits[3, ::15] = 1  # every group a literal with one bit: WAH lines of 2 bytes per 15 bits exactly
    stride = synth.row_stride_bytes(n_haps)
    packed = synth.pack_rows(bits, stride)
    p = G.params(n_haps // 2, block_len, n_haps // 1000)
    names = ["S%d" % i for i in range(n_haps // 2)]
    ref = G.oracle_file_from_bits(bits, p, names)
    region, offsets, res = G.encode_packed(packed, n_haps, p)
    got = G.assemble_file(region, offsets, p, n_lines, n_lines, names)
    assert got == ref
    out, _ = G.decode_packed(got, n_haps, stride)
    assert np.array_equal(out, packed)


@pytest.mark.parametrize("n_haps", [131074, 140002, 262144])
def test_long_row_expansion_on_adversarial_rows(n_haps):
    """Rows above 16 KiB are expanded by toggles (k_wah_expand_wide_t: every WAH16 word flips the bits where the row
    changes, the row is the running XOR): runs that start and end on 15-bit group, 32-bit word and 64-bit chunk
    boundaries, fills next to literals, literal after literal (lines of several 4096-word rounds), all ones, a single
    bit at either end - decoded rows against the source.  MAC threshold 0 and one line per block keep every line a WAH line in identity order, so
    the WAH words are exactly the rows' runs."""
    import gpu_util as G
    rng = np.random.default_rng(n_haps + 7)
    rows = []
    def row():
        rows.append(np.zeros(n_haps, dtype=np.uint8))
        return rows[-1]
    row()[:] = 1                                            # one ones-fill (+ a literal for the tail)
    r = row(); r[0] = 1                                     # a literal, then a zero fill
    r = row(); r[-1] = 1
    r = row(); r[15 * 7:15 * 9000] = 1                      # a fill from group to group
    r = row(); r[32 * 100:32 * 3000] = 1                    # ... from word boundary to word boundary
    r = row(); r[64 * 11 + 63:64 * 1500 + 1] = 1            # ... across chunk boundaries
    r = row(); r[14::15] = 1                                # every group a literal with its top bit: toggles that cancel
    r = row(); r[::15] = 1
    r = row(); r[:] = 1; r[15 * 400 + 7] = 0                # fill, literal, fill
    r = row(); r[:] = 1; r[::4097] = 0
    r = row(); r[:] = rng.integers(0, 2, n_haps)            # literal after literal: ceil(n / 15) words, several rounds
    r = row(); r[:] = (rng.random(n_haps) < 0.001)
    r = row()                                               # runs of random lengths
    pos, v = 0, 0
    while pos < n_haps:
        ln = int(rng.integers(1, 700))
        r[pos:pos + ln] = v
        pos, v = pos + ln, v ^ 1
    r = row(); r[n_haps - 15 * 3:] = 1                      # ones up to the very end
    bits = np.stack(rows)
    stride = synth.row_stride_bytes(n_haps)
    packed = synth.pack_rows(bits, stride)
    p = G.params(n_haps // 2, 1, 0)                         # every line its own block: identity order, thr 0
    names = ["S%d" % i for i in range(n_haps // 2)]
    # rows with no minor allele (all ones / all zeros) are sparse lines whatever the threshold: they ride along
    ref = G.oracle_file_from_bits(bits, p, names)
    out, counts = G.decode_packed(ref, n_haps, stride)
    assert np.array_equal(out, packed)
    assert np.array_equal(counts, bits.sum(1).astype(np.int32))
