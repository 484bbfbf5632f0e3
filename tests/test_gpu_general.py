"""GPU parity tests of the general genotype path (int32 rows: multi-allelic, missing,
end-of-vector, non-default phase, haploid lines) and of the file-level writer / accessor."""
import ctypes
import hashlib
import os
import struct

import numpy as np
import pytest

from xsqueezeit_amd import binding, vcf_lite
from test_oracle import ANCHORS, _random_lines

pytestmark = pytest.mark.gpu


def _ploidy(lines, n_samples):
    return max(len(gt) // n_samples for gt, _ in lines)


@pytest.mark.parametrize("name", sorted(ANCHORS))
def test_micro_fixture_through_gpu(name, golden_dir):
    """The reference's own fixtures: GPU-encoded bytes hit the reference's recorded SHA-256, and
    GPU decode returns the VCF's genotypes (the reference's test criterion, verify_v4.sh)."""
    import gpu_util as G
    from oracle import oracle
    samples, recs = vcf_lite.read_vcf(os.path.join(golden_dir, name + ".vcf"))
    lines = [(r["gt"], r["n_allele"]) for r in recs]
    n = len(samples)
    first_ploidy = len(lines[0][0]) // n
    p = G.params(n, 8192, int(n * first_ploidy * 0.002), oracle.default_phased_of(lines, n))
    region, offsets, res = G.encode_gt(lines, n, p)
    got = G.assemble_file(region, offsets, p, len(lines), G.num_variants(lines), samples, _ploidy(lines, n))
    size, sha = ANCHORS[name]
    assert len(got) == size
    assert hashlib.sha256(got).hexdigest() == sha
    assert res.max_ploidy == _ploidy(lines, n)
    rows, counts = G.decode_gt(got, [na for _, na in lines])
    ref = oracle.decode_file(got, [na for _, na in lines])
    for i, (r, (gt, _)) in enumerate(zip(rows, lines)):
        assert np.array_equal(r, gt), "line %d" % i
        assert np.array_equal(counts[i][:lines[i][1]], ref[i][1])


@pytest.mark.parametrize("maf,block_len", [(0.002, 8192), (0.0, 8192), (0.002, 4), (0.0, 4)])
def test_region_target_fixture_through_gpu(maf, block_len, golden_dir, tmp_path):
    """The reference's binary fixture test_region_target.bcf (test/cukinia_v4.conf:19; 6 records x 3202 samples of
    1000 Genomes data = 6404 haplotypes, re-expressed GT-only by tests/golden/make_region_target_fixture.py) through
    every GPU entry point.  --maf 0.002 is what verify_v4.sh passes (MAC threshold 12: two WAH lines, four sparse);
    --maf 0 makes all six WAH lines; block length 4 gives two blocks.
      * xsi_hip_encode_gt and xsi_hip_encode_packed (the lines are bi-allelic, fully called, default-phased): file ==
        the oracle's file, byte for byte;
      * xsi_writer (append / finalize) with the fixture's sample names: file == the oracle's;
      * xsi_hip_decode_gt == the VCF's genotypes (the reference's own pass criterion) and the oracle's allele counts;
      * xsi_hip_decode_packed == the ALT bit rows;
      * the -t chr17:117980-117999 subset (records by POS) through xsi_accessor_get_genotypes == the oracle's reader."""
    import gpu_util as G
    from oracle import oracle
    from xsqueezeit_amd import synth
    L = binding.lib()
    samples, recs = vcf_lite.read_vcf(os.path.join(golden_dir, "region_target.vcf"))
    lines = [(r["gt"], r["n_allele"]) for r in recs]
    n = len(samples)
    assert n == 3202 and len(lines) == 6
    thr = int(float(2 * n) * maf)
    dp = oracle.default_phased_of(lines, n)
    ref = oracle.encode_file(lines, n, maf=maf, block_len=block_len, sample_names=samples)
    p = G.params(n, block_len, thr, dp)
    nal = [na for _, na in lines]
    # general entry point
    region, offsets, res = G.encode_gt(lines, n, p)
    got = G.assemble_file(region, offsets, p, len(lines), G.num_variants(lines), samples, 2)
    assert got == ref
    assert res.n_wah_lines == (2 if maf else 6)
    # packed entry point on the ALT bit rows
    bits = np.stack([(((g >> 1) - 1) == 1).astype(np.uint8) for g, _ in lines])
    stride = synth.row_stride_bytes(2 * n)
    packed = synth.pack_rows(bits, stride)
    region2, offsets2, _ = G.encode_packed(packed, 2 * n, p)
    assert region2 == region and np.array_equal(offsets2, offsets)
    # file writer
    path = str(tmp_path / "rt.xsi").encode()
    w = ctypes.c_void_p()
    arr = (ctypes.c_char_p * n)(*[x.encode() for x in samples])
    binding.check(L.xsi_writer_open(ctypes.byref(w), G.ctx().handle, path, ctypes.byref(p), arr))
    for gt, na in lines:
        gt = np.ascontiguousarray(gt, dtype=np.int32)
        binding.check(L.xsi_writer_append(w, gt.ctypes.data, gt.size, na))
    binding.check(L.xsi_writer_finalize(w, 0))
    L.xsi_writer_close(w)
    assert open(path, "rb").read() == ref
    # decode
    rows, counts = G.decode_gt(ref, nal)
    oref = oracle.decode_file(ref, nal, block_len=block_len)
    for i in range(len(lines)):
        assert np.array_equal(rows[i], lines[i][0]), "record %d" % i
        assert np.array_equal(counts[i][:2], oref[i][1][:2])
    out, cnt = G.decode_packed(ref, 2 * n, stride)
    assert np.array_equal(out, packed)
    assert np.array_equal(cnt, bits.sum(1).astype(np.int32))
    # -t chr17:117980-117999 through the accessor, out of order
    a = ctypes.c_void_p()
    binding.check(L.xsi_accessor_open(ctypes.byref(a), G.ctx().handle, path))
    assert L.xsi_accessor_num_samples(a) == n and L.xsi_accessor_sample_name(a, 0) == samples[0].encode()
    rd = oracle.Reader(ref)
    picked = [i for i, r in enumerate(recs) if r["chrom"] == "chr17" and 117980 <= r["pos"] <= 117999]
    assert picked == [1, 2, 3, 4, 5]
    pp = ctypes.c_void_p()
    ngt_arr = ctypes.c_int(0)
    cbuf = np.zeros(2, dtype=np.uint64)
    for i in picked + [5, 2, 4]:
        bm = ((i // block_len) << 15) | (i % block_len)
        r = L.xsi_accessor_get_genotypes(a, 2, bm, ctypes.byref(pp), ctypes.byref(ngt_arr))
        assert r == 2 * n and ngt_arr.value == 2 * n, L.xsi_hip_last_error()
        gotrow = np.ctypeslib.as_array(ctypes.cast(pp, ctypes.POINTER(ctypes.c_int32)), shape=(r,))
        egt, ecnt = rd.fill_genotype_array(2, bm)
        assert np.array_equal(gotrow, egt) and np.array_equal(gotrow, recs[i]["gt"]), "target record %d" % i
        binding.check(L.xsi_accessor_allele_counts(a, cbuf.ctypes.data, 2))
        assert np.array_equal(cbuf, ecnt[:2])
    ctypes.CDLL(None).free(pp)
    L.xsi_accessor_close(a)


@pytest.mark.parametrize("n_samples,n_lines,block_len,maf,kw", [
    (50, 300, 64, 0.01, {}),
    (50, 200, 64, 0.01, dict(multi=True)),
    (37, 150, 32, 0.01, dict(missing=True, eov=True, phase=True, multi=True)),
    (333, 500, 128, 0.02, dict(missing=True, eov=True, phase=True, multi=True)),
    (2504, 300, 100, 0.001, dict(missing=True, phase=True)),
    (2504, 200, 64, 0.001, dict(multi=True, eov=True)),
    (32767, 12, 8, 0.001, dict(multi=True)),        # largest u16 A_T size (65534 haplotypes)
    (70000, 10, 4, 0.001, dict(multi=True, missing=True, eov=True, phase=True)),  # u32, global-memory chain
])
def test_general_encode_bit_exact_and_decode(n_samples, n_lines, block_len, maf, kw):
    import gpu_util as G
    from oracle import oracle
    rng = np.random.default_rng(n_samples * 7 + n_lines)
    lines = _random_lines(rng, n_samples, n_lines, **kw)
    dp = oracle.default_phased_of(lines, n_samples)
    p = G.params(n_samples, block_len, int(2 * n_samples * maf), dp)
    ref = oracle.encode_file(lines, n_samples, block_len=block_len, mac_thr=p.mac_threshold, default_phased=dp)
    region, offsets, res = G.encode_gt(lines, n_samples, p)
    names = ["S%d" % i for i in range(n_samples)]
    got = G.assemble_file(region, offsets, p, n_lines, G.num_variants(lines), names, 2)
    if got != ref:
        first = next(i for i in range(min(len(got), len(ref))) if got[i] != ref[i])
        raise AssertionError("file differs at offset %d (sizes %d vs %d)" % (first, len(got), len(ref)))
    nal = [na for _, na in lines]
    rows, counts = G.decode_gt(got, nal)
    oref = oracle.decode_file(ref, nal, block_len=block_len)
    for i in range(n_lines):
        assert np.array_equal(rows[i], oref[i][0]), "line %d" % i
        assert np.array_equal(rows[i], lines[i][0]), "line %d vs source" % i
        assert np.array_equal(counts[i][:nal[i]], oref[i][1]), "allele counts line %d" % i


@pytest.mark.gpu
@pytest.mark.parametrize("n_samples,n_lines,kw", [
    (2, 40, dict(missing=True, eov=True, phase=True)),             # 4 values: one quad
    (126, 60, dict(missing=True, eov=True, phase=True)),           # 252 values: less than a segment
    (1026, 50, dict(missing=True, eov=True, phase=True, multi=True)),   # 2052 values: one super-group and a quad
    (2504, 40, dict(missing=True, eov=True, phase=True, multi=True)),   # 5008: every wave's only super-group is partial
    (16390, 12, dict(missing=True, phase=True)),                   # 32780 values: four super-groups per wave, ragged end
    (70000, 6, dict(missing=True, eov=True, phase=True, multi=True)),
])
def test_quad_unpack_equals_ballot_unpack(n_samples, n_lines, kw, monkeypatch):
    """Bi-allelic lines whose rows are 16-byte aligned take the lane-local unpack (unpack_quads, xsi_gt.hip); the file it
    leads to is the oracle's, and byte-equal to what the ballot form (XSI_GT_NO_QUADS=1) writes for the same rows."""
    import gpu_util as G
    from oracle import oracle
    rng = np.random.default_rng(n_samples * 13 + n_lines)
    lines = _random_lines(rng, n_samples, n_lines, **kw)
    dp = oracle.default_phased_of(lines, n_samples)
    p = G.params(n_samples, 16, max(1, n_samples // 50), dp)
    ref = oracle.encode_file(lines, n_samples, block_len=16, mac_thr=p.mac_threshold, default_phased=dp)
    names = ["S%d" % i for i in range(n_samples)]
    files = []
    for no_quads in ("", "1"):
        if no_quads:
            monkeypatch.setenv("XSI_GT_NO_QUADS", "1")
        region, offsets, res = G.encode_gt(lines, n_samples, p)
        files.append(G.assemble_file(region, offsets, p, n_lines, G.num_variants(lines), names, 2))
    assert files[0] == files[1], "the two unpack forms disagree"
    assert files[0] == ref


@pytest.mark.parametrize("n,n_lines,block_len", [(60, 240, 64), (1600, 160, 64), (2504, 96, 32), (70000, 24, 8),
                                                 (12000, 40, 8),    # rank tracking + LDS kernel for the haploid blocks
                                                 (20000, 24, 8)])   # rank tracking + global-memory kernel for them
def test_haploid_and_mixed_lines(n, n_lines, block_len):
    """Fully haploid lines interleaved with diploid ones (mixed-ploidy chrX-like input), through the
    256-thread, the 1024-thread and the global-memory chain kernels."""
    import gpu_util as G
    from oracle import oracle
    rng = np.random.default_rng(11 + n)
    lines = []
    for i in range(n_lines):
        if i % 5 == 3:
            al = (rng.random(n) < 0.3).astype(np.int32)
            lines.append((((al + 1) << 1).astype(np.int32), 2))
        else:
            lines.extend(_random_lines(rng, n, 1, eov=(i % 7 == 0)))
    dp = oracle.default_phased_of(lines, n)
    thr = n // 500
    p = G.params(n, block_len, thr, dp)
    ref = oracle.encode_file(lines, n, block_len=block_len, mac_thr=thr, default_phased=dp)
    region, offsets, res = G.encode_gt(lines, n, p)
    got = G.assemble_file(region, offsets, p, len(lines), G.num_variants(lines), ["S%d" % i for i in range(n)], 2)
    assert got == ref
    nal = [na for _, na in lines]
    rows, _ = G.decode_gt(got, nal)
    oref = oracle.decode_file(ref, nal, block_len=block_len)
    for i in range(len(lines)):
        assert np.array_equal(rows[i], oref[i][0]), "line %d" % i


def test_wah_encode_missing_strategy():
    import gpu_util as G
    from oracle import oracle
    rng = np.random.default_rng(7)
    n = 90
    lines = _random_lines(rng, n, 200, missing=True, eov=True)
    dp = oracle.default_phased_of(lines, n)
    p = G.params(n, 50, 1, dp, wah_encode_missing=1)
    ref = oracle.encode_file(lines, n, block_len=50, mac_thr=1, default_phased=dp, wah_encode_missing=True)
    region, offsets, _ = G.encode_gt(lines, n, p)
    got = G.assemble_file(region, offsets, p, len(lines), G.num_variants(lines), ["S%d" % i for i in range(n)], 2)
    assert got == ref
    rows, _ = G.decode_gt(got, [na for _, na in lines])
    for r, (gt, _) in zip(rows, lines):
        assert np.array_equal(r, gt)


def test_allele_above_the_line_alleles_is_an_error():
    import gpu_util as G
    gt = np.full(20, 2, dtype=np.int32)
    gt[3] = (5 + 1) << 1  # allele 5 on a bi-allelic line
    p = G.params(10, 16, 0, 0)
    with pytest.raises(binding.XsiError):
        G.encode_gt([(gt, 2)], 10, p)


@pytest.mark.parametrize("kw", [{}, dict(multi=True, missing=True, eov=True, phase=True)])
def test_writer_and_accessor_files(tmp_path, kw):
    """File level: xsi_writer (XsiFactoryExt::append/finalize_file) output equals the oracle's file;
    xsi_accessor (Accessor::fill_genotype_array) returns the oracle reader's rows and counts,
    sequentially and with random jumps."""
    import gpu_util as G
    from oracle import oracle
    L = binding.lib()
    rng = np.random.default_rng(21)
    n, n_lines, block_len = 120, 700, 256
    lines = _random_lines(rng, n, n_lines, **kw)
    dp = oracle.default_phased_of(lines, n)
    names = ["sample_%d" % i for i in range(n)]
    mac = 2
    ref = oracle.encode_file(lines, n, block_len=block_len, mac_thr=mac, default_phased=dp, sample_names=names)
    path = str(tmp_path / "out.xsi").encode()
    p = G.params(n, block_len, mac, dp)
    w = ctypes.c_void_p()
    arr = (ctypes.c_char_p * n)(*[s.encode() for s in names])
    binding.check(L.xsi_writer_open(ctypes.byref(w), G.ctx().handle, path, ctypes.byref(p), arr))
    for gt, na in lines:
        gt = np.ascontiguousarray(gt, dtype=np.int32)
        binding.check(L.xsi_writer_append(w, gt.ctypes.data, gt.size, na))
    binding.check(L.xsi_writer_finalize(w, 0))
    L.xsi_writer_close(w)
    got = open(path, "rb").read()
    assert got == ref

    a = ctypes.c_void_p()
    binding.check(L.xsi_accessor_open(ctypes.byref(a), G.ctx().handle, path))
    assert L.xsi_accessor_hap_samples(a) == 2 * n
    assert L.xsi_accessor_num_samples(a) == n
    assert L.xsi_accessor_sample_name(a, 5) == b"sample_5"
    rd = oracle.Reader(ref)
    bms = []
    block = off = 0
    for i, (_, na) in enumerate(lines):
        if i and i % block_len == 0:
            block += 1
            off = 0
        bms.append((block << 15) | off)
        off += na - 1
    buf = np.zeros(2 * n, dtype=np.int32)
    cnt = np.zeros(8, dtype=np.uint64)
    order = list(range(n_lines)) + [int(x) for x in rng.integers(0, n_lines, 40)]
    for i in order:
        na = lines[i][1]
        r = L.xsi_accessor_fill_genotype_array(a, buf.ctypes.data, buf.size, na, bms[i])
        assert r == len(lines[i][0]), L.xsi_hip_last_error()
        exp, ecnt = rd.fill_genotype_array(na, bms[i])
        assert np.array_equal(buf[:r], exp), "line %d" % i
        binding.check(L.xsi_accessor_allele_counts(a, cnt.ctypes.data, na))
        assert np.array_equal(cnt[:na], ecnt)
    # the same lines through ONE batched call: random order, several blocks, rows with a stride wider than a line;
    # first into ordinary memory (device window + one copy per chunk), then into a registered array (the kernels store
    # into it), then with a cache that holds one block (every switch of block decodes again)
    q = np.asarray(order[::-1] + [5, 5, 699], dtype=np.int64)
    q_na = np.asarray([lines[i][1] for i in q], dtype=np.uint32)
    q_bm = np.asarray([bms[i] for i in q], dtype=np.uint64)
    stride = 2 * n + 3
    for mode in ("plain", "registered", "one-block cache"):
        rows2 = np.full((len(q), stride), -9, dtype=np.int32)
        ngt2 = np.zeros(len(q), dtype=np.uint32)
        if mode == "registered":  # page-locked memory of the accessor's (pageable memory is not taken)
            assert L.xsi_accessor_register_array(a, rows2.ctypes.data, rows2.size) == binding.XSI_ERR_ARG
            rows2 = G.accessor_array(a, (len(q), stride), -9)
            binding.check(L.xsi_accessor_register_array(a, rows2.ctypes.data, rows2.size))
        if mode == "one-block cache":
            binding.check(L.xsi_accessor_set_cache_bytes(a, 1))
        tot = L.xsi_accessor_get_genotypes_batch(a, len(q), q_na.ctypes.data, q_bm.ctypes.data, rows2.ctypes.data, stride,
                                                 ngt2.ctypes.data)
        assert tot == int(sum(len(lines[i][0]) for i in q)), L.xsi_hip_last_error()
        for k, i in enumerate(q):
            assert ngt2[k] == len(lines[i][0])
            assert np.array_equal(rows2[k, :ngt2[k]], lines[i][0]), "%s: query %d (line %d)" % (mode, k, i)
        assert np.all(rows2[:, 2 * n:] == -9)
        if mode == "registered":
            binding.check(L.xsi_accessor_unregister_array(a))
            G.accessor_array_free(a, rows2)
            assert L.xsi_accessor_free_array(a, rows2.ctypes.data) == binding.XSI_ERR_ARG  # not twice
            rows2 = None
    rows2 = np.full((len(q), stride), -9, dtype=np.int32)
    assert L.xsi_accessor_get_genotypes_batch(a, 1, q_na.ctypes.data, q_bm.ctypes.data, rows2.ctypes.data, 2 * n - 1,
                                              None) == binding.XSI_ERR_CAPACITY
    # get_genotypes allocates like Accessor::get_genotypes (accessor.hpp:58-67)
    pp = ctypes.c_void_p()
    ngt_arr = ctypes.c_int(0)
    r = L.xsi_accessor_get_genotypes(a, lines[0][1], bms[0], ctypes.byref(pp), ctypes.byref(ngt_arr))
    assert r == len(lines[0][0]) and ngt_arr.value == 2 * n and pp.value
    got0 = np.ctypeslib.as_array(ctypes.cast(pp, ctypes.POINTER(ctypes.c_int32)), shape=(r,)).copy()
    assert np.array_equal(got0, lines[0][0])
    ctypes.CDLL(None).free(pp)
    L.xsi_accessor_close(a)


def test_allele_counts_without_expansion(tmp_path):
    """fill_allele_counts path (no genotype expansion, no PBWT chain): device API and accessor mirror
    against the oracle's fill_allele_counts, including its quirk of not subtracting missing / EOV."""
    import gpu_util as G
    from oracle import oracle
    torch = G.torch_mod()
    L = binding.lib()
    rng = np.random.default_rng(33)
    n, n_lines, block_len = 150, 500, 128
    lines = _random_lines(rng, n, n_lines, multi=True, missing=True, eov=True)
    dp = oracle.default_phased_of(lines, n)
    ref = oracle.encode_file(lines, n, block_len=block_len, mac_thr=3, default_phased=dp)
    n_bin = G.num_variants(lines)
    d_file = G.dev_u8(np.frombuffer(ref, dtype=np.uint8))
    d_ones = torch.zeros(n_bin, dtype=torch.int32, device="cuda")
    d_kind = torch.zeros(n_bin, dtype=torch.uint8, device="cuda")
    nb = ctypes.c_uint64(0)
    n_blocks = (n_lines + block_len - 1) // block_len
    binding.check(L.xsi_hip_decode_counts(G.ctx().handle, d_file.data_ptr(), len(ref), 0, n_blocks, d_ones.data_ptr(),
                                          d_kind.data_ptr(), n_bin, ctypes.byref(nb)))
    assert nb.value == n_bin
    ones = d_ones.cpu().numpy()
    rd = oracle.Reader(ref)
    path = tmp_path / "f.xsi"
    path.write_bytes(ref)
    a = ctypes.c_void_p()
    binding.check(L.xsi_accessor_open(ctypes.byref(a), G.ctx().handle, str(path).encode()))
    cnt = np.zeros(8, dtype=np.uint64)
    block = off = g = 0
    for i, (gt, na) in enumerate(lines):
        if i and i % block_len == 0:
            block += 1
            off = 0
        bm = (block << 15) | off
        exp = rd.fill_allele_counts(na, bm)
        assert np.array_equal(ones[g:g + na - 1], exp[1:na].astype(np.int32)), "line %d" % i
        binding.check(L.xsi_accessor_fill_allele_counts(a, na, bm))
        binding.check(L.xsi_accessor_allele_counts(a, cnt.ctypes.data, na))
        assert np.array_equal(cnt[:na], exp), "accessor line %d" % i
        off += na - 1
        g += na - 1
    # interleave with genotype fills on the same accessor (state switches between the two views)
    buf = np.zeros(2 * n, dtype=np.int32)
    r = L.xsi_accessor_fill_genotype_array(a, buf.ctypes.data, buf.size, lines[3][1], 0 << 15 | sum(x[1] - 1 for x in lines[:3]))
    assert r == 2 * n and np.array_equal(buf, lines[3][0])
    binding.check(L.xsi_accessor_fill_allele_counts(a, lines[0][1], 0))
    L.xsi_accessor_close(a)


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["readahead", "no_readahead", "reparse", "small_cache"])
def test_sequential_scan_readahead_in_the_accessor(tmp_path, monkeypatch, mode):
    """Round 6.  A scan that asks for every line in file order: after a few consecutive queries the
    accessor finishes a prefix-decoded block in ONE continuation (from the plan it kept: no second parse, no second
    boundary scan), decodes cold blocks whole, and has block b + 1 decoded by a second thread while b is served
    (accessor_internals_new.hpp:154-196 replays; loading_time/gt_loader_new.hpp:112-172 is the sequential consumer).
    Every row and allele count equals the source / the oracle's reader in every mode: the policy on, off
    (XSI_ACCESSOR_NO_READAHEAD: rounds 4 - 5), continuations that parse again (XSI_ACCESSOR_REPARSE), and a cache too
    small for two blocks (no read-ahead may throw out the block being served).  A jump away in the middle of the scan
    and back must not disturb it."""
    import gpu_util as G
    from oracle import oracle
    L = binding.lib()
    for k in ("XSI_ACCESSOR_NO_READAHEAD", "XSI_ACCESSOR_REPARSE"):
        monkeypatch.delenv(k, raising=False)
    if mode == "no_readahead":
        monkeypatch.setenv("XSI_ACCESSOR_NO_READAHEAD", "1")
    if mode == "reparse":
        monkeypatch.setenv("XSI_ACCESSOR_REPARSE", "1")
    rng = np.random.default_rng(6100)
    n, block_len, n_blocks = 25000, 150, 5
    n_lines = block_len * n_blocks - 20
    lines = _random_lines(rng, n, n_lines, multi=True, missing=True, eov=True)
    dp = oracle.default_phased_of(lines, n)
    names = ["s%d" % i for i in range(n)]
    plain = oracle.encode_file(lines, n, block_len=block_len, mac_thr=50, default_phased=dp, sample_names=names)
    path = tmp_path / "seq.xsi"
    path.write_bytes(plain)
    rd = oracle.Reader(plain)
    bms, block, off = [], 0, 0
    for i, (_, na) in enumerate(lines):
        if i and i % block_len == 0:
            block, off = block + 1, 0
        bms.append((block << 15) | off)
        off += na - 1
    a = ctypes.c_void_p()
    binding.check(L.xsi_accessor_open(ctypes.byref(a), G.ctx().handle, str(path).encode()))
    if mode == "small_cache":
        # room for one decoded block and a half
        buf0 = np.zeros(2 * n, dtype=np.int32)
        assert L.xsi_accessor_fill_genotype_array(a, buf0.ctypes.data, buf0.size, lines[block_len][1], bms[block_len]) > 0
        nb, by = ctypes.c_uint64(0), ctypes.c_uint64(0)
        binding.check(L.xsi_accessor_cache_stats(a, ctypes.byref(nb), ctypes.byref(by), None, None))
        assert nb.value == 1
        binding.check(L.xsi_accessor_set_cache_bytes(a, 0))
        binding.check(L.xsi_accessor_set_cache_bytes(a, int(by.value * 1.5)))
    buf = np.zeros(2 * n, dtype=np.int32)
    cnt = np.zeros(8, dtype=np.uint64)

    def check(i, full=False):
        na = lines[i][1]
        r = L.xsi_accessor_fill_genotype_array(a, buf.ctypes.data, buf.size, na, bms[i])
        assert r == len(lines[i][0]), L.xsi_hip_last_error()
        assert np.array_equal(buf[:r], lines[i][0]), "line %d vs source" % i
        if full:
            egt, ecnt = rd.fill_genotype_array(na, bms[i])
            assert np.array_equal(buf[:r], egt), "line %d vs oracle reader" % i
            binding.check(L.xsi_accessor_allele_counts(a, cnt.ctypes.data, na))
            assert np.array_equal(cnt[:na], ecnt), "allele counts of line %d" % i

    for i in range(n_lines):
        check(i, full=(i % 23 == 0))
        if i == 2 * block_len + 40:  # a jump into a block far away and back: the scan's count starts again
            check(n_lines - 3, full=True)
            check(5, full=True)
    st, hits = ctypes.c_uint64(0), ctypes.c_uint64(0)
    binding.check(L.xsi_accessor_readahead_stats(a, ctypes.byref(st), ctypes.byref(hits)))
    pd, ext = ctypes.c_uint64(0), ctypes.c_uint64(0)
    binding.check(L.xsi_accessor_prefix_stats(a, ctypes.byref(pd), ctypes.byref(ext)))
    if mode in ("readahead", "reparse"):
        assert st.value >= 2 and hits.value >= 2, (st.value, hits.value)
        assert ext.value <= 4, "a prefix-decoded block of a sequential scan is finished in one continuation (%d)" % ext.value
    elif mode == "no_readahead":
        assert st.value == 0 and hits.value == 0
        assert ext.value >= 5  # the growing continuations of rounds 4 - 5
    else:
        assert st.value == 0  # the cache cannot hold two blocks: nothing is read ahead
    # a second scan, everything in HBM or evicted: same rows
    for i in range(0, n_lines, 7):
        check(i)
    L.xsi_accessor_close(a)



@pytest.mark.parametrize("zstd", [False, True])
def test_prefix_decode_in_the_accessor(tmp_path, zstd):
    """Round 4: a cold query decodes its block only up to the requested line (the chain over the WAH lines in front of
    it, the sparse lists and side matrices up to it) and later queries further in continue from the parked state -
    the reference's seek replays the same prefix on the host (accessor_internals_new.hpp:154-196).  25 000 samples
    (50 000 haplotypes: the ranged chain's domain), multi-allelic + missing + end-of-vector + phase lines, plain and
    zstd file (a continuation inflates the block again); queries: late in a block first?  No - early first, then
    further in (continuation), then backwards (already decoded), then the other block, then a batch across both;
    rows and allele counts against the oracle's reader and the source."""
    import gpu_util as G
    from oracle import oracle
    L = binding.lib()
    if zstd:
        try:
            ctypes.CDLL("libzstd.so.1")
        except OSError:
            pytest.skip("no libzstd.so.1 on this box")
    rng = np.random.default_rng(4100 + int(zstd))
    n, n_lines, block_len = 25000, 400, 200
    lines = _random_lines(rng, n, n_lines, multi=True, missing=True, eov=True, phase=True)
    dp = oracle.default_phased_of(lines, n)
    names = ["s%d" % i for i in range(n)]
    path = str(tmp_path / "p.xsi").encode()
    p = G.params(n, block_len, 50, dp)
    p.zstd_level = 3 if zstd else 0
    w = ctypes.c_void_p()
    arr = (ctypes.c_char_p * n)(*[x.encode() for x in names])
    binding.check(L.xsi_writer_open(ctypes.byref(w), G.ctx().handle, path, ctypes.byref(p), arr))
    for gt, na in lines:
        gt = np.ascontiguousarray(gt, dtype=np.int32)
        binding.check(L.xsi_writer_append(w, gt.ctypes.data, gt.size, na))
    binding.check(L.xsi_writer_finalize(w, 0))
    L.xsi_writer_close(w)
    plain = oracle.encode_file(lines, n, block_len=block_len, mac_thr=50, default_phased=dp, sample_names=names)
    if not zstd:
        assert open(path, "rb").read() == plain
    rd = oracle.Reader(plain)
    bms, block, off = [], 0, 0
    for i, (_, na) in enumerate(lines):
        if i and i % block_len == 0:
            block, off = block + 1, 0
        bms.append((block << 15) | off)
        off += na - 1
    a = ctypes.c_void_p()
    binding.check(L.xsi_accessor_open(ctypes.byref(a), G.ctx().handle, path))
    buf = np.zeros(2 * n, dtype=np.int32)
    cnt = np.zeros(8, dtype=np.uint64)
    pd, ext = ctypes.c_uint64(0), ctypes.c_uint64(0)

    def check(i):
        na = lines[i][1]
        r = L.xsi_accessor_fill_genotype_array(a, buf.ctypes.data, buf.size, na, bms[i])
        assert r == len(lines[i][0]), L.xsi_hip_last_error()
        egt, ecnt = rd.fill_genotype_array(na, bms[i])
        assert np.array_equal(buf[:r], lines[i][0]), "line %d vs source" % i
        assert np.array_equal(buf[:r], egt), "line %d vs oracle reader" % i
        binding.check(L.xsi_accessor_allele_counts(a, cnt.ctypes.data, na))
        assert np.array_equal(cnt[:na], ecnt), "allele counts of line %d" % i

    check(7)                                   # cold block 0: prefix up to line 7
    binding.check(L.xsi_accessor_prefix_stats(a, ctypes.byref(pd), ctypes.byref(ext)))
    assert pd.value == 1 and ext.value == 0
    check(8)                                   # inside the rounded-up prefix or a first continuation
    check(60)                                  # continuation
    check(3)                                   # backwards: already there
    check(150)                                 # continuation
    check(199)                                 # to the block's end
    binding.check(L.xsi_accessor_prefix_stats(a, ctypes.byref(pd), ctypes.byref(ext)))
    assert ext.value >= 2
    check(399)                                 # cold block 1, last line: (nearly) the whole block
    check(200)
    for i in [int(x) for x in rng.integers(0, n_lines, 12)]:
        check(i)
    # a fresh accessor, one batch over both cold blocks, unordered
    L.xsi_accessor_close(a)
    a = ctypes.c_void_p()
    binding.check(L.xsi_accessor_open(ctypes.byref(a), G.ctx().handle, path))
    q = np.asarray([250, 12, 13, 390, 101, 12, 201], dtype=np.int64)
    q_na = np.asarray([lines[i][1] for i in q], dtype=np.uint32)
    q_bm = np.asarray([bms[i] for i in q], dtype=np.uint64)
    rows2 = np.full((len(q), 2 * n), -9, dtype=np.int32)
    ngt2 = np.zeros(len(q), dtype=np.uint32)
    tot = L.xsi_accessor_get_genotypes_batch(a, len(q), q_na.ctypes.data, q_bm.ctypes.data, rows2.ctypes.data, 2 * n, ngt2.ctypes.data)
    assert tot == int(sum(len(lines[i][0]) for i in q)), L.xsi_hip_last_error()
    for k, i in enumerate(q):
        assert np.array_equal(rows2[k, :ngt2[k]], lines[i][0]), "batched query %d (line %d)" % (k, i)
    binding.check(L.xsi_accessor_prefix_stats(a, ctypes.byref(pd), ctypes.byref(ext)))
    assert pd.value >= 1
    # a cache that cannot hold a block: the prefix is decoded, cannot be stored, and the block is finished in the
    # context's workspace (every switch of block decodes again)
    binding.check(L.xsi_accessor_set_cache_bytes(a, 1))
    for i in (10, 210, 11, 399, 0):
        check(i)
    L.xsi_accessor_close(a)


def test_zstd_layer_roundtrip(tmp_path):
    """--zstd files (BlockWithZstdCompressor, interfaces.hpp:288-315): u64 sizes + one zstd frame per
    block.  The inflated blocks equal the plain file's blocks byte for byte, and the accessor reads
    the compressed file back.  (Frame bytes themselves depend on the libzstd version, as they do for
    the reference, so they are not pinned.)"""
    import struct
    import gpu_util as G
    from oracle import oracle
    L = binding.lib()
    try:
        Z = ctypes.CDLL("libzstd.so.1")
    except OSError:
        pytest.skip("no libzstd.so.1 on this box")
    Z.ZSTD_decompress.restype = ctypes.c_size_t
    Z.ZSTD_decompress.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t]
    rng = np.random.default_rng(77)
    n, n_lines, block_len = 100, 600, 256
    lines = _random_lines(rng, n, n_lines, multi=True, missing=True)
    dp = oracle.default_phased_of(lines, n)
    names = ["s%d" % i for i in range(n)]
    plain = oracle.encode_file(lines, n, block_len=block_len, mac_thr=2, default_phased=dp, sample_names=names)
    path = str(tmp_path / "z.xsi").encode()
    p = G.params(n, block_len, 2, dp)
    p.zstd_level = 7
    w = ctypes.c_void_p()
    arr = (ctypes.c_char_p * n)(*[s.encode() for s in names])
    binding.check(L.xsi_writer_open(ctypes.byref(w), G.ctx().handle, path, ctypes.byref(p), arr))
    for gt, na in lines:
        gt = np.ascontiguousarray(gt, dtype=np.int32)
        binding.check(L.xsi_writer_append(w, gt.ctypes.data, gt.size, na))
    binding.check(L.xsi_writer_finalize(w, 0))
    L.xsi_writer_close(w)
    z = open(path, "rb").read()
    assert z[17] & 4, "zstd bit of the header"
    assert len(z) < len(plain)
    # inflate every block and compare with the plain file's block
    zio, zso = struct.unpack_from("<QQ", z, 72)
    pio, pso = struct.unpack_from("<QQ", plain, 72)
    zoffs = np.frombuffer(z, "<u8", (zso - zio) // 8, zio)
    poffs = list(np.frombuffer(plain, "<u8", (pso - pio) // 8, pio)) + [pio]
    assert len(zoffs) == len(poffs) - 1
    for b, off in enumerate(zoffs):
        csize, usize = struct.unpack_from("<QQ", z, int(off))
        out = ctypes.create_string_buffer(usize)
        src = z[int(off) + 16:int(off) + 16 + csize]
        r = Z.ZSTD_decompress(out, usize, src, csize)
        assert r == usize
        assert out.raw == plain[int(poffs[b]):int(poffs[b]) + usize], "block %d" % b
        assert int(poffs[b + 1]) - int(poffs[b]) - usize < 8
    # read the compressed file back
    a = ctypes.c_void_p()
    binding.check(L.xsi_accessor_open(ctypes.byref(a), G.ctx().handle, path))
    buf = np.zeros(2 * n, dtype=np.int32)
    block = off = 0
    for i, (gt, na) in enumerate(lines):
        if i and i % block_len == 0:
            block += 1
            off = 0
        r = L.xsi_accessor_fill_genotype_array(a, buf.ctypes.data, buf.size, na, (block << 15) | off)
        assert r == len(gt) and np.array_equal(buf[:r], gt), "line %d" % i
        off += na - 1
    binding.check(L.xsi_accessor_fill_allele_counts(a, lines[0][1], 0))
    L.xsi_accessor_close(a)


@pytest.mark.gpu
def test_zstd_writer_pool_equals_one_thread_writer_and_oracle_blocks(tmp_path, monkeypatch):
    """--zstd at block-parallel speed: batches of several blocks, every block compressed by one of a pool
    of host threads, frames written in block order.  ZSTD_compress is one-shot per block and deterministic, so the file
    must be, byte for byte, (a) the file of the one-thread / one-block-per-batch writer of rounds 1 - 5 and (b) the
    oracle's plain blocks, each cut at its unpadded length and wrapped by this box's libzstd at the same level
    (compress_and_write, interfaces.hpp:291-314)."""
    import struct
    import gpu_util as G
    from oracle import oracle
    L = binding.lib()
    try:
        Z = ctypes.CDLL("libzstd.so.1")
    except OSError:
        pytest.skip("no libzstd.so.1 on this box")
    Z.ZSTD_compress.restype = ctypes.c_size_t
    Z.ZSTD_compress.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    rng = np.random.default_rng(4242)
    n, block_len, n_blocks = 400, 48, 37
    n_lines = block_len * n_blocks - 11  # a short last block
    lines = _random_lines(rng, n, n_lines, multi=True, missing=True, eov=True)
    dp = oracle.default_phased_of(lines, n)
    names = ["s%d" % i for i in range(n)]
    plain = oracle.encode_file(lines, n, block_len=block_len, mac_thr=3, default_phased=dp, sample_names=names)

    def write(tag, env):
        for k in ("XSI_WRITER_ZSTD_THREADS", "XSI_WRITER_BATCH_BLOCKS"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        path = str(tmp_path / ("z_%s.xsi" % tag)).encode()
        p = G.params(n, block_len, 3, dp)
        p.zstd_level = 7
        w = ctypes.c_void_p()
        arr = (ctypes.c_char_p * n)(*[s.encode() for s in names])
        binding.check(L.xsi_writer_open(ctypes.byref(w), G.ctx().handle, path, ctypes.byref(p), arr))
        for gt, na in lines:
            gt = np.ascontiguousarray(gt, dtype=np.int32)
            binding.check(L.xsi_writer_append(w, gt.ctypes.data, gt.size, na))
        binding.check(L.xsi_writer_finalize(w, 0))
        L.xsi_writer_close(w)
        return open(path, "rb").read()

    pooled = write("pool", {})                                                            # default pool, default batches
    one = write("one", {"XSI_WRITER_ZSTD_THREADS": "1", "XSI_WRITER_BATCH_BLOCKS": "1"})  # rounds 1 - 5
    odd = write("odd", {"XSI_WRITER_ZSTD_THREADS": "3", "XSI_WRITER_BATCH_BLOCKS": "7"})  # batches that do not divide the file
    assert pooled == one, "pooled writer differs from the one-thread writer"
    assert odd == one
    # (b) the oracle's blocks wrapped by libzstd
    zio, zso = struct.unpack_from("<QQ", pooled, 72)
    pio, pso = struct.unpack_from("<QQ", plain, 72)
    zoffs = np.frombuffer(pooled, "<u8", (zso - zio) // 8, zio)
    poffs = list(np.frombuffer(plain, "<u8", (pso - pio) // 8, pio)) + [pio]
    assert len(zoffs) == n_blocks == len(poffs) - 1
    pos = 256
    for b in range(n_blocks):
        assert int(zoffs[b]) == pos, "block %d starts where the one before ended (padded to 4)" % b
        csize, usize = struct.unpack_from("<QQ", pooled, pos)
        padded = int(poffs[b + 1]) - int(poffs[b]) if b + 1 < n_blocks else None
        if padded is not None:
            assert 0 <= padded - usize < 4 and plain[int(poffs[b]) + usize:int(poffs[b]) + padded] == b"\0" * (padded - usize)
        raw = plain[int(poffs[b]):int(poffs[b]) + usize]
        dst = ctypes.create_string_buffer(2 * usize + 64)
        cs = Z.ZSTD_compress(dst, len(dst), raw, usize, 7)
        assert cs == csize and dst.raw[:cs] == pooled[pos + 16:pos + 16 + csize], "frame of block %d" % b
        pos += 16 + csize
        pos += (-pos) % 4
    # and it reads back
    a = ctypes.c_void_p()
    path = str(tmp_path / "z_pool.xsi").encode()
    binding.check(L.xsi_accessor_open(ctypes.byref(a), G.ctx().handle, path))
    buf = np.zeros(2 * n, dtype=np.int32)
    block = off = 0
    for i, (gt, na) in enumerate(lines):
        if i and i % block_len == 0:
            block += 1
            off = 0
        if i % 7 == 0:
            r = L.xsi_accessor_fill_genotype_array(a, buf.ctypes.data, buf.size, na, (block << 15) | off)
            assert r == len(gt) and np.array_equal(buf[:r], gt), "line %d" % i
        off += na - 1
    L.xsi_accessor_close(a)



@pytest.mark.gpu
@pytest.mark.parametrize("budget", [None, "two_blocks", 0])
def test_accessor_random_access_block_cache(tmp_path, budget):
    """Config-5 style use: uniformly random BM positions over many blocks (mixed ploidy, multi-allelic,
    EOV) through fill_genotype_array, interleaved with fill_allele_counts.  Decoded blocks stay
    resident in HBM; results must not depend on the cache budget (default / two blocks with
    evictions / none = decode in the shared workspace)."""
    import gpu_util as G
    from oracle import oracle
    L = binding.lib()
    rng = np.random.default_rng(5150)
    n, n_lines, block_len = 90, 1300, 100
    lines = _random_lines(rng, n, n_lines, multi=True, missing=True, eov=True, phase=True)
    dp = oracle.default_phased_of(lines, n)
    ref = oracle.encode_file(lines, n, block_len=block_len, mac_thr=2, default_phased=dp)
    path = tmp_path / "ra.xsi"
    path.write_bytes(ref)
    a = ctypes.c_void_p()
    binding.check(L.xsi_accessor_open(ctypes.byref(a), G.ctx().handle, str(path).encode()))
    u64 = ctypes.c_uint64
    if budget == "two_blocks":
        # decode one block, read its footprint, allow two of them
        buf0 = np.zeros(2 * n, dtype=np.int32)
        assert L.xsi_accessor_fill_genotype_array(a, buf0.ctypes.data, buf0.size, lines[0][1], 0) > 0
        nb, by = u64(0), u64(0)
        binding.check(L.xsi_accessor_cache_stats(a, ctypes.byref(nb), ctypes.byref(by), None, None))
        assert nb.value == 1 and by.value > 0
        binding.check(L.xsi_accessor_set_cache_bytes(a, 2 * by.value + 1024))
    elif budget == 0:
        binding.check(L.xsi_accessor_set_cache_bytes(a, 0))
    bms = []
    block = off = 0
    for i, (_, na) in enumerate(lines):
        if i and i % block_len == 0:
            block += 1
            off = 0
        bms.append((block << 15) | off)
        off += na - 1
    # one reader per block on the oracle side: it replays forward only, so visit in sorted order there
    order = [int(x) for x in rng.integers(0, n_lines, 400)]
    rd = oracle.Reader(ref)
    expect = {}
    for i in sorted(set(order)):
        expect[i] = rd.fill_genotype_array(lines[i][1], bms[i])
    buf = np.zeros(2 * n, dtype=np.int32)
    cnt = np.zeros(8, dtype=np.uint64)
    for k, i in enumerate(order):
        na = lines[i][1]
        egt, ecnt = expect[i]
        r = L.xsi_accessor_fill_genotype_array(a, buf.ctypes.data, buf.size, na, bms[i])
        assert r == len(egt), "line %d: %s" % (i, binding.lib().xsi_hip_last_error())
        assert np.array_equal(buf[:r], egt), "line %d (query %d)" % (i, k)
        binding.check(L.xsi_accessor_allele_counts(a, cnt.ctypes.data, na))
        assert np.array_equal(cnt[:na], ecnt)
        if k % 7 == 0:  # counts-only view of another block in between
            j = order[(k * 13 + 5) % len(order)]
            binding.check(L.xsi_accessor_fill_allele_counts(a, lines[j][1], bms[j]))
    nb, by, hits, misses = u64(0), u64(0), u64(0), u64(0)
    binding.check(L.xsi_accessor_cache_stats(a, ctypes.byref(nb), ctypes.byref(by), ctypes.byref(hits), ctypes.byref(misses)))
    n_blocks = (n_lines + block_len - 1) // block_len
    if budget is None:
        assert nb.value == n_blocks and misses.value == n_blocks  # every block decoded exactly once
    elif budget == "two_blocks":
        assert nb.value <= 2 and misses.value > n_blocks
    else:
        assert nb.value == 0
    L.xsi_accessor_close(a)


@pytest.mark.gpu
def test_phenotype_dot_products_on_decoded_planes():
    """Sxy per ALT allele = sum of the phenotype over the carrying haplotypes
    (dot_prod/dot_prod.hpp), float64.  Expected values from the oracle-decoded genotypes; the GPU sums in
    a different (fixed) order than the reference's PBWT-order walk, hence a relative tolerance of 1e-12
    on sums of O(1) values."""
    import gpu_util as G
    from oracle import oracle
    torch = G.torch_mod()
    L = binding.lib()
    rng = np.random.default_rng(808)
    n, n_lines, block_len, K = 700, 500, 128, 3
    lines = _random_lines(rng, n, n_lines, phase=True)           # bi-allelic, fully called, mixed phase
    for i in range(0, n_lines, 9):                                # some lines with the ALT allele in the majority
        gt = lines[i][0]
        flip = ((gt >> 1) - 1) ^ 1
        lines[i] = ((((flip + 1) << 1) | (gt & 1)).astype(np.int32), 2)
    dp = oracle.default_phased_of(lines, n)
    ref = oracle.encode_file(lines, n, block_len=block_len, mac_thr=14, default_phased=dp)
    y = rng.normal(0.0, 10.0, size=(n, K))
    d_file = G.dev_u8(np.frombuffer(ref, dtype=np.uint8))
    d_y = torch.from_numpy(y).cuda()
    d_out = torch.zeros((n_lines, K), dtype=torch.float64, device="cuda")
    nb = ctypes.c_uint64(0)
    n_blocks = (n_lines + block_len - 1) // block_len
    binding.check(L.xsi_hip_decode_dot(G.ctx().handle, d_file.data_ptr(), len(ref), 0, n_blocks, d_y.data_ptr(), K,
                                       d_out.data_ptr(), n_lines, ctypes.byref(nb)))
    assert nb.value == n_lines
    got = d_out.cpu().numpy()
    dec = oracle.decode_file(ref, [2] * n_lines, block_len=block_len)
    exp = np.stack([((((g >> 1) - 1) == 1).astype(np.float64)[:, None] * np.repeat(y, 2, axis=0)).sum(0) for g, _ in dec])
    assert np.allclose(got, exp, rtol=1e-12, atol=1e-9)
    # many phenotypes: float64 matrix cores (16 per pass; 21 = one full and one partial group)
    K2 = 21
    y2m = rng.normal(0.0, 10.0, size=(n, K2))
    d_y2 = torch.from_numpy(y2m).cuda()
    d_out2 = torch.zeros((n_lines, K2), dtype=torch.float64, device="cuda")
    binding.check(L.xsi_hip_decode_dot(G.ctx().handle, d_file.data_ptr(), len(ref), 0, n_blocks, d_y2.data_ptr(), K2,
                                       d_out2.data_ptr(), n_lines, None))
    alt = np.stack([(((g >> 1) - 1) == 1).astype(np.float64) for g, _ in dec])
    exp2 = alt @ np.repeat(y2m, 2, axis=0)
    assert np.allclose(d_out2.cpu().numpy(), exp2, rtol=1e-12, atol=1e-9)
    # files the planes alone cannot answer are refused, not answered approximately
    lines2 = _random_lines(rng, 40, 60, multi=True, missing=True)
    ref2 = oracle.encode_file(lines2, 40, block_len=32, mac_thr=1, default_phased=oracle.default_phased_of(lines2, 40))
    d_file2 = G.dev_u8(np.frombuffer(ref2, dtype=np.uint8))
    y2 = torch.zeros((40, 1), dtype=torch.float64, device="cuda")
    o2 = torch.zeros((400, 1), dtype=torch.float64, device="cuda")
    rc = L.xsi_hip_decode_dot(G.ctx().handle, d_file2.data_ptr(), len(ref2), 0, 2, y2.data_ptr(), 1, o2.data_ptr(), 400, None)
    assert rc == -5  # XSI_ERR_UNSUPPORTED


def _expected_dot(dec, nal, y, n):
    """numpy contraction of oracle-decoded rows: one output row per (BCF line, ALT allele)."""
    out = []
    for (g, _), na in zip(dec, nal):
        ploidy = len(g) // n
        al = (g.astype(np.int64) >> 1) - 1          # missing -> -1, end-of-vector -> negative
        ys = np.repeat(y, ploidy, axis=0)
        for k in range(1, na):
            out.append(((al == k).astype(np.float64)[:, None] * ys).sum(0))
    return np.stack(out)


@pytest.mark.gpu
def test_phenotype_dot_products_on_composed_genotypes():
    """SURVEY §8f-4 on the blocks the plane path refuses: multi-allelic lines (with a majority ALT allele, i.e.
    negated sparse lines), missing and end-of-vector entries, non-default phase, and fully haploid lines in
    blocks of their own.  Expected: numpy contraction of the oracle-decoded rows (rtol 1e-12: fixed but
    different summation order).  Also: identical bits to xsi_hip_decode_dot where both apply, and the same
    result when the workspace budget forces one block per range."""
    import gpu_util as G
    from oracle import oracle
    torch = G.torch_mod()
    L = binding.lib()
    rng = np.random.default_rng(909)
    n, block_len, K = 421, 48, 5
    lines = []
    for b in range(5):
        for i in range(block_len if b < 4 else 17):
            if b == 2:     # a block of fully haploid and plain diploid lines (no multi-allelic ones beside them)
                if i % 3 == 0:
                    al = (rng.random(n) < 0.4).astype(np.int32)
                    lines.append((((al + 1) << 1).astype(np.int32), 2))
                else:
                    lines.extend(_random_lines(rng, n, 1, missing=(i % 4 == 1)))
            else:
                lines.extend(_random_lines(rng, n, 1, multi=True, missing=(i % 3 == 0), eov=(i % 5 == 0), phase=True))
                if i % 7 == 0:   # make ALT 1 the majority allele: its line is stored as a negated sparse line
                    gt, na = lines[-1]
                    keep = (gt == oracle.INT32_VECTOR_END) | ((gt >> 1) == 0)
                    al = (gt >> 1) - 1
                    sw = np.where(al == 0, 1, np.where(al == 1, 0, al))
                    g2 = (((sw + 1) << 1) | (gt & 1)).astype(np.int32)
                    lines[-1] = (np.where(keep, gt, g2).astype(np.int32), na)
    nal = [na for _, na in lines]
    dp = oracle.default_phased_of(lines, n)
    ref = oracle.encode_file(lines, n, block_len=block_len, mac_thr=9, default_phased=dp)
    dec = oracle.decode_file(ref, nal, block_len=block_len)
    y = rng.normal(0.0, 10.0, size=(n, K))
    exp = _expected_dot(dec, nal, y, n)
    n_bin = exp.shape[0]
    d_file = G.dev_u8(np.frombuffer(ref, dtype=np.uint8))
    d_y = torch.from_numpy(y).cuda()
    h_nal = np.asarray(nal, dtype=np.uint32)

    def run(n_pheno_y, first=0, nb_blocks=5, nal_slice=h_nal):
        d_out = torch.full((n_bin + 3, n_pheno_y.shape[1]), np.nan, dtype=torch.float64, device="cuda")
        nb = ctypes.c_uint64(0)
        binding.check(L.xsi_hip_decode_dot_gt(G.ctx().handle, d_file.data_ptr(), len(ref), first, nb_blocks,
                                              nal_slice.ctypes.data, len(nal_slice), n_pheno_y.data_ptr(),
                                              n_pheno_y.shape[1], d_out.data_ptr(), n_bin + 3, ctypes.byref(nb)))
        return d_out.cpu().numpy(), nb.value

    got, nb = run(d_y)
    assert nb == n_bin
    assert np.allclose(got[:n_bin], exp, rtol=1e-12, atol=1e-9)
    assert np.isnan(got[n_bin:]).all()
    # a sub-range of blocks (the haploid block and the one after it)
    lo, hi = 2 * block_len, 4 * block_len
    bin_lo = int(sum(a - 1 for a in nal[:lo])); bin_hi = int(sum(a - 1 for a in nal[:hi]))
    sub, nb = run(d_y, first=2, nb_blocks=2, nal_slice=np.ascontiguousarray(h_nal[lo:hi]))
    assert nb == bin_hi - bin_lo
    assert np.array_equal(sub[:nb], got[bin_lo:bin_hi])
    # one block per range (tiny workspace budget): same bits
    binding.check(L.xsi_hip_ctx_set_workspace_budget(G.ctx().handle, 1 << 16))
    try:
        small, _ = run(d_y)
    finally:
        binding.check(L.xsi_hip_ctx_set_workspace_budget(G.ctx().handle, 0))
    assert np.array_equal(small[:n_bin], got[:n_bin])
    # wrong allele numbers are refused
    bad = h_nal.copy(); bad[3] += 1
    d_o = torch.zeros((n_bin + 8, K), dtype=torch.float64, device="cuda")
    rc = L.xsi_hip_decode_dot_gt(G.ctx().handle, d_file.data_ptr(), len(ref), 0, 5, bad.ctypes.data, len(bad),
                                 d_y.data_ptr(), K, d_o.data_ptr(), n_bin + 8, None)
    assert rc == -1
    # on a file both entry points cover the two agree bit for bit
    lines3 = _random_lines(rng, n, 100, phase=True)
    ref3 = oracle.encode_file(lines3, n, block_len=block_len, mac_thr=9, default_phased=oracle.default_phased_of(lines3, n))
    d_file3 = G.dev_u8(np.frombuffer(ref3, dtype=np.uint8))
    y3 = torch.from_numpy(rng.normal(0.0, 10.0, size=(n, 3))).cuda()
    o_a = torch.zeros((100, 3), dtype=torch.float64, device="cuda")
    o_b = torch.zeros((100, 3), dtype=torch.float64, device="cuda")
    nal3 = np.full(100, 2, dtype=np.uint32)
    binding.check(L.xsi_hip_decode_dot(G.ctx().handle, d_file3.data_ptr(), len(ref3), 0, 3, y3.data_ptr(), 3,
                                       o_a.data_ptr(), 100, None))
    binding.check(L.xsi_hip_decode_dot_gt(G.ctx().handle, d_file3.data_ptr(), len(ref3), 0, 3, nal3.ctypes.data, 100,
                                          y3.data_ptr(), 3, o_b.data_ptr(), 100, None))
    assert torch.equal(o_a, o_b)


def test_accessor_internal_access(tmp_path):
    """get_internal_access (accessor_internals_new.hpp:444-471): per binary line of a record WAH-or-sparse and
    where its data sits, plus the PBWT arrangement `a` in force at the record's last line.  Checked by
    recomputation: WAH-or-sparse from the allele counts and the MAC threshold (gt_block.hpp:299-326), `a` by
    numpy stable partitions over the earlier WAH lines of the block (internal_gt_record.hpp:32-59), and the
    data itself: the WAH words at the returned offset expand (oracle wah_extract) to the line's bits gathered
    through `a` (wah.hpp:530-537), sparse lists hold the carriers (the REF haplotypes when negated)."""
    import gpu_util as G
    from oracle import oracle
    L = binding.lib()
    rng = np.random.default_rng(4242)
    n, block_len, thr = 300, 60, 12
    N = 2 * n
    lines = []
    for i in range(3 * block_len - 7):
        lines.extend(_random_lines(rng, n, 1, multi=(i % 3 == 0), missing=(i % 5 == 0), phase=True))
        if i % 11 == 0:   # ALT 1 the majority allele: a negated sparse line or a dense WAH line
            gt, na = lines[-1]
            al = (gt >> 1) - 1
            sw = np.where(al == 0, 1, np.where(al == 1, 0, al))
            keep = (gt >> 1) == 0
            lines[-1] = (np.where(keep, gt, ((sw + 1) << 1) | (gt & 1)).astype(np.int32), na)
    dp = oracle.default_phased_of(lines, n)
    ref = oracle.encode_file(lines, n, block_len=block_len, mac_thr=thr, default_phased=dp)
    path = tmp_path / "ia.xsi"
    path.write_bytes(ref)
    a = ctypes.c_void_p()
    binding.check(L.xsi_accessor_open(ctypes.byref(a), G.ctx().handle, str(path).encode()))
    # per BCF line: BM, and per binary line: allele bits, WAH or sparse
    bms, first_bin = [], []
    block = off = 0
    for i, (_, na) in enumerate(lines):
        if i and i % block_len == 0:
            block, off = block + 1, 0
        bms.append((block << 15) | off)
        off += na - 1
    def bits_of(i, k):
        gt = lines[i][0]
        return (((gt >> 1) - 1) == k).astype(np.uint8)
    def is_wah(x):
        c = int(x.sum())
        return min(c, N - c) > thr
    checked_wah = checked_sparse = checked_neg = 0
    for i in [int(v) for v in rng.integers(0, len(lines), 30)] + [0, block_len, len(lines) - 1]:
        gt, na = lines[i]
        info = binding.InternalAccess()
        sp = np.zeros(na - 1, dtype=np.uint8)
        offs = np.zeros(na - 1, dtype=np.uint64)
        arr = np.zeros(N, dtype=np.uint32)
        binding.check(L.xsi_accessor_get_internal_access(a, na, bms[i], ctypes.byref(info), sp.ctypes.data, offs.ctypes.data,
                                                         arr.ctypes.data))
        assert (info.n_alleles, info.sparse_bytes, info.wah_bytes, info.a_bytes, info.n_a) == (na, 2, 2, 4, N)
        img = np.frombuffer((ctypes.c_uint8 * info.image_len).from_address(info.image), dtype=np.uint8)
        # arrangement at the record's last binary line: stable partitions over the block's earlier WAH lines
        exp_a = np.arange(N, dtype=np.uint32)
        b0 = (i // block_len) * block_len
        for j in range(b0, i + 1):
            for k in range(1, lines[j][1]):
                if j == i and k == na - 1:
                    break
                x = bits_of(j, k)
                if is_wah(x):
                    y = x[exp_a]
                    exp_a = np.concatenate([exp_a[y == 0], exp_a[y == 1]])
        assert np.array_equal(arr, exp_a), "line %d" % i
        for k in range(1, na):
            x = bits_of(i, k)
            assert sp[k - 1] == (0 if is_wah(x) else 1), "line %d allele %d" % (i, k)
            o = int(offs[k - 1])
            if not sp[k - 1]:
                if k == na - 1:   # `a` is the arrangement of this line
                    words = img[o:o + 2 * (N // 15 + 2)].view(np.uint16)
                    y, _, _ = oracle.wah_extract(words, N)
                    assert np.array_equal(y, x[arr]), "line %d allele %d" % (i, k)
                    checked_wah += 1
            else:
                num = int(img[o:o + 2].view(np.uint16)[0])
                neg = bool(num & 0x8000)
                num &= 0x7FFF
                listed = img[o + 2:o + 2 + 2 * num].view(np.uint16).astype(np.int64)
                want = np.flatnonzero((((gt >> 1) - 1) == 0) if neg else x)
                assert np.array_equal(listed, want), "line %d allele %d" % (i, k)
                checked_sparse += 1
                checked_neg += neg
                if k == 1:
                    assert info.default_allele == (1 if neg else 0)
    assert checked_wah >= 5 and checked_sparse >= 5
    # the accessor still serves genotypes afterwards, by copy and by view (a pointer into its pinned window)
    buf = np.zeros(N, dtype=np.int32)
    assert L.xsi_accessor_fill_genotype_array(a, buf.ctypes.data, N, lines[7][1], bms[7]) == N
    assert np.array_equal(buf, lines[7][0])
    for i in (0, 7, block_len + 3, len(lines) - 1):
        ptr = ctypes.c_void_p()
        n = L.xsi_accessor_genotypes_view(a, lines[i][1], bms[i], ctypes.byref(ptr))
        assert n == N and ptr.value
        view = np.frombuffer((ctypes.c_int32 * n).from_address(ptr.value), dtype=np.int32)
        assert np.array_equal(view, lines[i][0]), "line %d" % i
    L.xsi_accessor_close(a)


def test_accessor_sample_subset(tmp_path):
    """fill_selected_genotypes == the reference's fill_selected_genotypes (gt_decompressor_new.hpp:209-238):
    the listed samples' values in list order, 1 or 2 per sample by the line's ploidy, AN and the AC of
    every ALT allele over the selection."""
    import gpu_util as G
    from oracle import oracle
    L = binding.lib()
    rng = np.random.default_rng(77)
    n, block_len = 333, 40
    lines = []
    for b in range(3):
        for i in range(block_len):
            if b == 1 and i % 4 == 2:  # fully haploid lines, kept apart from the multi-allelic ones
                al = (rng.random(n) < 0.3).astype(np.int32)
                lines.append((((al + 1) << 1).astype(np.int32), 2))
            else:
                lines.extend(_random_lines(rng, n, 1, multi=(b != 1), missing=True, eov=(i % 2 == 0), phase=True))
    dp = oracle.default_phased_of(lines, n)
    ref = oracle.encode_file(lines, n, block_len=block_len, mac_thr=1, default_phased=dp)
    path = tmp_path / "sub.xsi"
    path.write_bytes(ref)
    a = ctypes.c_void_p()
    binding.check(L.xsi_accessor_open(ctypes.byref(a), G.ctx().handle, str(path).encode()))
    buf = np.zeros(2 * n, dtype=np.int32)
    ac = np.zeros(8, dtype=np.int32)
    assert L.xsi_accessor_fill_selected_genotypes(a, buf.ctypes.data, buf.size, 2, 0, None) == binding.XSI_ERR_ARG
    bad = np.array([n], dtype=np.uint32)
    assert L.xsi_accessor_set_sample_subset(a, bad.ctypes.data, 1) == binding.XSI_ERR_ARG
    bms = []
    block = off = 0
    for i, (_, na) in enumerate(lines):
        if i and i % block_len == 0:
            block, off = block + 1, 0
        bms.append((block << 15) | off)
        off += na - 1
    for sel in (np.array([5, 0, 332, 5, 17], dtype=np.uint32), rng.permutation(n)[:120].astype(np.uint32)):
        binding.check(L.xsi_accessor_set_sample_subset(a, sel.ctypes.data, len(sel)))
        for i in [int(x) for x in rng.integers(0, len(lines), 40)]:
            gt, na = lines[i]
            ploidy = len(gt) // n
            exp = gt.reshape(n, ploidy)[sel].reshape(-1)
            an = L.xsi_accessor_fill_selected_genotypes(a, buf.ctypes.data, buf.size, na, bms[i], ac.ctypes.data)
            assert an == len(sel) * ploidy, L.xsi_hip_last_error()
            assert np.array_equal(buf[:an], exp), "line %d" % i
            alleles = (exp >> 1) - 1
            for k in range(1, na):
                assert ac[k - 1] == int(np.sum(alleles == k)), "AC of allele %d on line %d" % (k, i)
        # the unselected path still works next to it
        full = L.xsi_accessor_fill_genotype_array(a, buf.ctypes.data, buf.size, lines[3][1], bms[3])
        assert full == len(lines[3][0]) and np.array_equal(buf[:full], lines[3][0])
    binding.check(L.xsi_accessor_set_sample_subset(a, None, 0))
    assert L.xsi_accessor_fill_selected_genotypes(a, buf.ctypes.data, buf.size, 2, 0, None) == binding.XSI_ERR_ARG
    L.xsi_accessor_close(a)


def test_reencode_on_device(tmp_path):
    """xsi_hip_reencode = the -Ox path (gt_decompressor_new.hpp:241-273): decode a file and encode it again
    with other parameters, optionally for a selection of samples, genotypes staying in HBM.  The result must
    be the file the oracle writes from the same (selected) lines with the new parameters."""
    import gpu_util as G
    from oracle import oracle
    L = binding.lib()
    torch = G.torch_mod()
    rng = np.random.default_rng(4242)
    n, block_len = 260, 50
    lines = []
    for b in range(4):
        for i in range(block_len):
            if b == 2 and i % 5 == 1:
                al = (rng.random(n) < 0.25).astype(np.int32)
                lines.append((((al + 1) << 1).astype(np.int32), 2))
            else:
                lines.extend(_random_lines(rng, n, 1, multi=(b != 2), missing=(i % 3 == 0), eov=(i % 4 == 0), phase=True))
    dp = oracle.default_phased_of(lines, n)
    src = oracle.encode_file(lines, n, block_len=block_len, mac_thr=1, default_phased=dp)
    d_file = G.dev_u8(np.frombuffer(src, np.uint8))
    nal = np.asarray([na for _, na in lines], dtype=np.uint32)
    n_bin = int((nal - 1).sum())
    for sel in (None, np.sort(rng.permutation(n)[:97]).astype(np.uint32)):
        n_new = n if sel is None else len(sel)
        # the haploid lines sit in block 2 of the source; with 80-line blocks they share blocks with
        # multi-allelic lines, so the new files are compared as bytes, not decoded
        p = G.params(n_new, 80, 3, dp)
        cap = int(L.xsi_hip_encode_gt_bound(ctypes.byref(p), len(lines), n_bin))
        d_out = G.dev_empty(cap)
        n_blocks = (len(lines) + 79) // 80
        d_off = torch.zeros(n_blocks, dtype=torch.int64, device="cuda")
        res = binding.EncodeResult()
        binding.check(L.xsi_hip_reencode(G.ctx().handle, d_file.data_ptr(), len(src), nal.ctypes.data, len(lines),
                                         ctypes.byref(p), sel.ctypes.data if sel is not None else None, n_new,
                                         d_out.data_ptr(), cap, d_off.data_ptr(), ctypes.byref(res)))
        region = d_out[:res.blocks_bytes].cpu().numpy().tobytes()
        names = ["S%d" % i for i in range(n_new)]
        got = G.assemble_file(region, d_off.cpu().numpy().astype(np.uint64), p, len(lines), G.num_variants(lines), names, 2)
        if sel is None:
            new_lines = lines
        else:
            new_lines = [(gt.reshape(n, len(gt) // n)[sel].reshape(-1), na) for gt, na in lines]
        ref = oracle.encode_file(new_lines, n_new, block_len=80, mac_thr=3, default_phased=dp, sample_names=names)
        assert got == ref
        assert L.xsi_hip_ctx_reencode_ranges(G.ctx().handle) == 1
        # the same job under a workspace budget that cannot hold the file's rows - it is walked in
        # ranges of whole source blocks (50 lines) that do not line up with the new blocks (80 or 30 lines): the leftover
        # of a range is carried into the next.  1000 bytes: the smallest staging buffer that makes progress (one new
        # block - 1 + one source block of lines): 2 ranges for 80-line blocks, 4 for 30-line blocks.
        per_line = 6 * 2 * n + (8 * n_new if sel is not None else 0)
        for budget, new_bl, min_ranges in ((1000, 80, 2), (1000, 30, 4), (170 * per_line, 80, 2)):
            p2 = G.params(n_new, new_bl, 3, dp)
            nb2 = (len(lines) + new_bl - 1) // new_bl
            d_off2 = torch.zeros(nb2, dtype=torch.int64, device="cuda")
            cap2 = int(L.xsi_hip_encode_gt_bound(ctypes.byref(p2), len(lines), n_bin))
            d_out2 = G.dev_empty(cap2)
            try:
                binding.check(L.xsi_hip_ctx_set_workspace_budget(G.ctx().handle, budget))
                res2 = binding.EncodeResult()
                binding.check(L.xsi_hip_reencode(G.ctx().handle, d_file.data_ptr(), len(src), nal.ctypes.data, len(lines),
                                                 ctypes.byref(p2), sel.ctypes.data if sel is not None else None, n_new,
                                                 d_out2.data_ptr(), cap2, d_off2.data_ptr(), ctypes.byref(res2)))
            finally:
                binding.check(L.xsi_hip_ctx_set_workspace_budget(G.ctx().handle, 0))
            assert L.xsi_hip_ctx_reencode_ranges(G.ctx().handle) >= min_ranges
            assert res2.n_blocks == nb2 and res2.n_binary_lines == res.n_binary_lines and res2.max_ploidy == res.max_ploidy
            region2 = d_out2[:res2.blocks_bytes].cpu().numpy().tobytes()
            got2 = G.assemble_file(region2, d_off2.cpu().numpy().astype(np.uint64), p2, len(lines), G.num_variants(lines), names, 2)
            ref2 = ref if new_bl == 80 else oracle.encode_file(new_lines, n_new, block_len=new_bl, mac_thr=3, default_phased=dp,
                                                               sample_names=names)
            assert got2 == ref2, "budget %d, %d-line blocks" % (budget, new_bl)
    bad = G.params(n + 1, 80, 3, dp)
    assert L.xsi_hip_reencode(G.ctx().handle, d_file.data_ptr(), len(src), nal.ctypes.data, len(lines), ctypes.byref(bad),
                              None, 0, d_out.data_ptr(), cap, d_off.data_ptr(), ctypes.byref(res)) == binding.XSI_ERR_ARG


@pytest.mark.gpu
@pytest.mark.parametrize("bad", [(3 + 1) << 1, ((2 + 1) << 1) | 1, -1, -2, -3, -4, -2147483646])
def test_unknown_allele_is_an_error(bad):
    """An allele number outside [0, n_allele) ends the encode with "Unknown allele error !" (scan_genotypes,
    gt_block.hpp:226-268): too large, and the negative int32 values that are neither missing (0, 1, INT_MIN)
    nor end-of-vector (INT_MIN + 1).  The same rows without the bad value encode, and missing / end-of-vector
    values right next to it are not mistaken for alleles."""
    import gpu_util as G
    torch = G.torch_mod()
    L = binding.lib()
    n, n_lines = 70, 6          # 140 values: the bad one sits in the ragged last chunk of 64
    N = 2 * n
    rng = np.random.default_rng(77)
    m = (((rng.integers(0, 2, size=(n_lines, N)) + 1) << 1) | (np.arange(N) & 1)).astype(np.int32)
    m[1, 5] = 0                          # missing
    m[1, 6] = np.int32(-2147483648)      # missing (bcf_int32_missing)
    m[2, 7] = np.int32(-2147483647)      # end of vector
    ngt = np.full(n_lines, N, dtype=np.uint32)
    nal = np.full(n_lines, 2, dtype=np.uint32)
    p = G.params(n, 8192, 1)
    cap = int(L.xsi_hip_encode_gt_bound(ctypes.byref(p), n_lines, n_lines))
    d_out = G.dev_empty(cap)
    d_off = torch.zeros(1, dtype=torch.int64, device="cuda")
    res = binding.EncodeResult()

    def run(mat):
        d_gt = torch.from_numpy(mat).cuda()
        return L.xsi_hip_encode_gt(G.ctx().handle, ctypes.byref(p), d_gt.data_ptr(), N, n_lines, ngt.ctypes.data,
                                   nal.ctypes.data, d_out.data_ptr(), cap, d_off.data_ptr(), ctypes.byref(res))

    assert run(m) == 0
    for pos in (3, 131):                 # first chunk; last (partial) chunk
        mb = m.copy()
        mb[4, pos] = np.int32(bad)
        rc = run(mb)
        assert rc == binding.XSI_ERR_ARG, (bad, pos, rc)
        assert b"Unknown allele" in L.xsi_hip_last_error()
    assert run(m) == 0                   # the context stays usable


@pytest.mark.parametrize("every,use_row_buffer", [(0, False), (37, False), (1, False), (37, True), (0, True)])
def test_writer_packs_on_append_and_mixes_line_kinds(tmp_path, every, use_row_buffer, monkeypatch):
    """xsi_writer_append packs a simple line (alleles 0 / 1, fully called, default phase) to bits in the caller's
    thread and ships only those; other lines travel as int32.  A batch without such lines takes xsi_hip_encode_packed,
    a mixed one has its packed lines expanded on the device and takes xsi_hip_encode_gt.  Whatever the mix
    (`every`: 0 = none general, k = every k-th line general; through append or through the row buffer), the file
    is the oracle's, and the same as with packing switched off."""
    import gpu_util as G
    from oracle import oracle
    L = binding.lib()
    rng = np.random.default_rng(5)
    n, n_lines, block_len = 333, 900, 256   # 666 values: a ragged tail for every vector width
    simple = _random_lines(rng, n, n_lines)
    general = _random_lines(rng, n, n_lines, multi=True, missing=True, eov=True, phase=True)
    lines = [general[i] if every and i % every == every // 2 else simple[i] for i in range(n_lines)]
    dp = 1
    names = ["s%d" % i for i in range(n)]
    ref = oracle.encode_file(lines, n, block_len=block_len, mac_thr=3, default_phased=dp, sample_names=names)
    p = G.params(n, block_len, 3, dp)
    arr = (ctypes.c_char_p * n)(*[s.encode() for s in names])

    def write(path):
        w = ctypes.c_void_p()
        binding.check(L.xsi_writer_open(ctypes.byref(w), G.ctx().handle, path, ctypes.byref(p), arr))
        L.xsi_writer_row_buffer.restype = ctypes.POINTER(ctypes.c_int32)
        for gt, na in lines:
            gt = np.ascontiguousarray(gt, dtype=np.int32)
            if use_row_buffer:
                dst = L.xsi_writer_row_buffer(w)
                assert dst
                ctypes.memmove(dst, gt.ctypes.data, gt.nbytes)
                binding.check(L.xsi_writer_commit_row(w, gt.size, na))
            else:
                binding.check(L.xsi_writer_append(w, gt.ctypes.data, gt.size, na))
        binding.check(L.xsi_writer_finalize(w, 0))
        L.xsi_writer_close(w)
        return open(path, "rb").read()

    got = write(str(tmp_path / "a.xsi").encode())
    assert got == ref


def test_accessor_internal_access_with_haploid_lines(tmp_path):
    """get_internal_access in blocks that hold fully haploid lines (one value per sample): such a line partitions the
    diploid arrangement by its sample's bit (pbwt_sort1, internal_gt_record.hpp:50-59) and its WAH words are the
    sample bits gathered through the even members of `a`, halved (haploid_rearrangement_from_diploid,
    interfaces.hpp:318-333).  Checked by recomputation in numpy, as test_accessor_internal_access."""
    import gpu_util as G
    from oracle import oracle
    L = binding.lib()
    rng = np.random.default_rng(777)
    n, block_len, thr = 400, 50, 6
    N = 2 * n
    lines = []
    for i in range(2 * block_len + 9):
        if i % 4 == 2:
            al = (rng.random(n) < float(rng.random()) * 0.6).astype(np.int32)
            lines.append((((al + 1) << 1).astype(np.int32), 2))   # haploid: n values
        else:
            lines.extend(_random_lines(rng, n, 1))
    dp = oracle.default_phased_of(lines, n)
    ref = oracle.encode_file(lines, n, block_len=block_len, mac_thr=thr, default_phased=dp)
    path = tmp_path / "iah.xsi"
    path.write_bytes(ref)
    a = ctypes.c_void_p()
    binding.check(L.xsi_accessor_open(ctypes.byref(a), G.ctx().handle, str(path).encode()))

    def bits_of(i):
        return (((lines[i][0] >> 1) - 1) == 1).astype(np.uint8)

    def is_wah(x):
        c = int(x.sum())
        return min(c, x.size - c) > thr

    checked_hap = checked_dip = 0
    for i in list(range(0, len(lines), 7)) + [2, 6, block_len + 2, len(lines) - 1]:
        info = binding.InternalAccess()
        sp = np.zeros(1, dtype=np.uint8)
        offs = np.zeros(1, dtype=np.uint64)
        arr = np.zeros(N, dtype=np.uint32)
        bm = ((i // block_len) << 15) | (i % block_len)
        binding.check(L.xsi_accessor_get_internal_access(a, 2, bm, ctypes.byref(info), sp.ctypes.data, offs.ctypes.data,
                                                         arr.ctypes.data))
        exp_a = np.arange(N, dtype=np.uint32)
        for j in range((i // block_len) * block_len, i):
            x = bits_of(j)
            if is_wah(x):
                key = x[exp_a // 2] if x.size == n else x[exp_a]
                exp_a = np.concatenate([exp_a[key == 0], exp_a[key == 1]])
        assert np.array_equal(arr, exp_a), "line %d" % i
        x = bits_of(i)
        assert sp[0] == (0 if is_wah(x) else 1), "line %d" % i
        if not sp[0]:
            img = np.frombuffer((ctypes.c_uint8 * info.image_len).from_address(info.image), dtype=np.uint8)
            o = int(offs[0])
            words = img[o:o + 2 * (N // 15 + 2)].view(np.uint16)
            if x.size == n:
                a1 = arr[arr % 2 == 0] // 2
                y, _, _ = oracle.wah_extract(words, n)
                assert np.array_equal(y, x[a1]), "haploid line %d" % i
                checked_hap += 1
            else:
                y, _, _ = oracle.wah_extract(words, N)
                assert np.array_equal(y, x[arr]), "line %d" % i
                checked_dip += 1
    assert checked_hap >= 3 and checked_dip >= 3
    L.xsi_accessor_close(a)


def test_accessor_fills_a_caller_pinned_array(tmp_path):
    """xsi_accessor_register_array with an array the CALLER has page-locked (torch pinned memory): the accessor uses it
    as it is (the kernel stores into it) and leaves it alone at close."""
    import gpu_util as G
    from oracle import oracle
    torch = G.torch_mod()
    L = binding.lib()
    rng = np.random.default_rng(99)
    n = 12000
    lines = _random_lines(rng, n, 12, multi=True, missing=True)
    dp = oracle.default_phased_of(lines, n)
    ref = oracle.encode_file(lines, n, block_len=8, mac_thr=12, default_phased=dp)
    path = tmp_path / "pinned.xsi"
    path.write_bytes(ref)
    a = ctypes.c_void_p()
    binding.check(L.xsi_accessor_open(ctypes.byref(a), G.ctx().handle, str(path).encode()))
    t = torch.empty(2 * n, dtype=torch.int32, pin_memory=True)
    buf = t.numpy()
    binding.check(L.xsi_accessor_register_array(a, buf.ctypes.data, buf.size))
    bms, block, off = [], 0, 0
    for i, (_, na) in enumerate(lines):
        if i and i % 8 == 0:
            block, off = block + 1, 0
        bms.append((block << 15) | off)
        off += na - 1
    for i in (3, 3, 9, 0, 11, 5, 5):
        buf[:] = -7
        assert L.xsi_accessor_fill_genotype_array(a, buf.ctypes.data, buf.size, lines[i][1], bms[i]) == 2 * n
        assert np.array_equal(buf, lines[i][0]), "line %d" % i
    L.xsi_accessor_close(a)
    t.fill_(5)   # still the caller's pinned tensor
    d = t.cuda(non_blocking=True)
    torch.cuda.synchronize()
    assert int(d.sum().item()) == 10 * n


@pytest.mark.parametrize("no_zerocopy", [False, True])
def test_haploid_file_never_overruns_the_callers_array(tmp_path, monkeypatch, no_zerocopy):
    """a max_ploidy = 1 file has hap_samples = num_samples, but composed rows are 2 * num_samples wide.  The
    array get_genotypes mallocs (hap_samples values) must never be written past its end - not by the kernel, not by
    the copy engine - and registering it for the direct path is refused.  20 000 samples: 80 KB lines, with guard
    values behind the array; once through the kernel-store path, once with XSI_ACCESSOR_NO_ZEROCOPY=1."""
    import gpu_util as G
    from oracle import oracle
    L = binding.lib()
    if no_zerocopy:
        monkeypatch.setenv("XSI_ACCESSOR_NO_ZEROCOPY", "1")
    rng = np.random.default_rng(5150)
    n = 20000
    lines = []
    for i in range(6):
        k = int(rng.integers(30, n // 2))
        g = np.zeros(n, dtype=np.int32)
        g[rng.choice(n, k, replace=False)] = 1
        lines.append((((g + 1) << 1).astype(np.int32), 2))   # haploid line: n values, unphased
    ref = oracle.encode_file(lines, n, block_len=4, mac_thr=20, default_phased=0)
    assert struct.unpack_from("<Q", ref, 32)[0] == n and struct.unpack_from("<Q", ref, 112)[0] == n
    path = tmp_path / "hap.xsi"
    path.write_bytes(ref)
    a = ctypes.c_void_p()
    binding.check(L.xsi_accessor_open(ctypes.byref(a), G.ctx().handle, str(path).encode()))
    assert L.xsi_accessor_hap_samples(a) == n
    guard = 4096
    store = np.full(n + guard, -99, dtype=np.int32)
    buf = store[:n]
    # an array of hap_samples values cannot take a composed row: the direct path refuses it
    assert L.xsi_accessor_register_array(a, buf.ctypes.data, n) == binding.XSI_ERR_CAPACITY
    pbuf = ctypes.c_void_p(buf.ctypes.data)
    nout = ctypes.c_int(0)
    for i in (1, 1, 1, 4, 5, 0, 0):
        store[:] = -99
        bm = ((i // 4) << 15) | (i % 4)
        r = L.xsi_accessor_get_genotypes(a, 2, bm, ctypes.byref(pbuf), ctypes.byref(nout))
        assert r == n and nout.value == n, L.xsi_hip_last_error()
        assert np.array_equal(buf, lines[i][0]), "line %d" % i
        assert np.all(store[n:] == -99), "values written behind the caller's array (line %d)" % i
    # a registered array that is wide enough takes the direct path and gets the same values
    wide = G.accessor_array(a, (2 * n + guard,), -99)
    binding.check(L.xsi_accessor_register_array(a, wide.ctypes.data, 2 * n))
    for i in (2, 2, 3):
        wide[:] = -99
        assert L.xsi_accessor_fill_genotype_array(a, wide.ctypes.data, 2 * n, 2, ((i // 4) << 15) | (i % 4)) == n
        assert np.array_equal(wide[:n], lines[i][0])
        assert np.all(wide[2 * n:] == -99)
    L.xsi_accessor_close(a)


@pytest.mark.parametrize("n,n_lines,block_len,kw", [
    (37, 150, 32, dict(missing=True, eov=True, phase=True, multi=True)),
    (600, 90, 40, dict(missing=True, eov=True)),
    (2504, 60, 16, dict(missing=True, multi=True)),
    (70000, 6, 3, dict(missing=True, eov=True, multi=True)),   # u32 A_T, rows above 16 KiB
])
def test_decode_pbwt_weirdness_files(n, n_lines, block_len, kw, tmp_path):
    """Files whose missing / end-of-vector lines are WAH lines permuted by a PBWT-sorted a_weirdness (WS_PBWT_WAH, the
    version-4 default; gt_block.hpp:340-395, accessor_internals_new.hpp:300-340, 503-533), written by the oracle (the
    reference's CLI cannot select the strategy any more; no fixture of the reference's exists for it, so this pins the
    GPU decode to the CPU restatement and to the source rows only).  Whole-file decode and random accessor queries."""
    import gpu_util as G
    from oracle import oracle
    L = binding.lib()
    rng = np.random.default_rng(31 * n + n_lines)
    lines = _random_lines(rng, n, n_lines, **kw)
    dp = oracle.default_phased_of(lines, n)
    ref = oracle.encode_file(lines, n, block_len=block_len, mac_thr=max(1, n // 500), default_phased=dp, wah_encode_missing=2)
    nal = [na for _, na in lines]
    rows, counts = G.decode_gt(ref, nal)
    oref = oracle.decode_file(ref, nal, block_len=block_len)
    for i in range(len(lines)):
        assert np.array_equal(rows[i], lines[i][0]), "line %d vs source" % i
        assert np.array_equal(rows[i], oref[i][0]), "line %d vs oracle" % i
        assert np.array_equal(counts[i][:nal[i]], oref[i][1][:nal[i]]), "line %d counts" % i
    path = tmp_path / "pw.xsi"
    path.write_bytes(ref)
    a = ctypes.c_void_p()
    binding.check(L.xsi_accessor_open(ctypes.byref(a), G.ctx().handle, str(path).encode()))
    buf = np.zeros(2 * n, dtype=np.int32)
    bms, block, off = [], 0, 0
    for i, (_, na) in enumerate(lines):
        if i and i % block_len == 0:
            block, off = block + 1, 0
        bms.append((block << 15) | off)
        off += na - 1
    for i in [int(v) for v in rng.integers(0, len(lines), 12)]:
        assert L.xsi_accessor_fill_genotype_array(a, buf.ctypes.data, buf.size, nal[i], bms[i]) == 2 * n
        assert np.array_equal(buf, lines[i][0]), "accessor line %d" % i
    L.xsi_accessor_close(a)
