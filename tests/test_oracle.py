"""Pins the CPU oracle (oracle/xsi_oracle.c) against everything the reference offers for this
path: the WAH16 known answers, the .xsi size + SHA-256 anchors SURVEY.md §8c recorded
for the 7 micro VCF fixtures (second-hand: from the reference's headers behind a stand-in vcf.h, in a
harness that was not kept - see ANCHORS), the worked example (§9.4b), decode round trips (the
reference's own test criterion, test/scripts/verify_v4.sh) on the micro VCFs and on the reference's
binary fixture test_region_target.bcf (6 records x 6404 haplotypes, tests/golden/region_target.vcf).
"""
import hashlib
import os
import struct

import numpy as np
import pytest

from oracle import oracle
from xsqueezeit_amd import vcf_lite

# Size + SHA-256 of the .xsi the reference's hot-path headers produced for each micro VCF, as SURVEY.md section 8c
# recorded them.  SECOND-HAND: the surveyor's harness drove those headers through a stand-in for htslib's vcf.h and was
# not persisted; nothing in this repo can rebuild it (the reference needs htslib, absent from the image).  The
# reference's own pass criterion for these fixtures is decode equality (verify_v4.sh), which the tests check as well.
ANCHORS = {
    "micro_eov": (536, "922f1dcb3e706ffc41feef268bfe1e3d0b6ae6077d547b92a6ab1e29a0e7bc36"),
    "micro_haploid": (480, "718837a6a150f62ece14873d6353bc841d27a05dc69b574acd1773d7def1a1d8"),
    "micro_missing": (544, "851e1a74ab564ef5572eaab3db8c158b7cc48238d9cde920f79a7ea18aa2c367"),
    "micro_missing_non_uniform_phasing": (552, "347043911a1e5c242cc0cc855ff0090e1dad5895fbdc3f4cb01269dbdfc127d4"),
    "micro_missing_non_uniform_phasing_ploidy": (608, "574735585d0f3f179043e6775e57bd0f40995836ae9ccf0b26086571b9dfb1b2"),
    "micro_mixed_ploidy": (592, "5ce2f0aeab000f81ff244942c2c227da91c2fbcfd8ce0d482b34b3deb57ec42b"),
    "micro_non_uniform_phase": (528, "3ab2f55f12aea0bcae7ccb96b1c04941e9dceed2e1dba20a437ccd688aac94d0"),
}


def hexwords(w):
    return " ".join("%04x" % int(x) for x in w)


# ---- WAH16 known answers (reference wah_encode2<uint16_t>) ----
def test_wah_known_answers():
    assert hexwords(oracle.wah_encode_bits(np.zeros(5008, np.uint8))) == "814e"
    assert hexwords(oracle.wah_encode_bits(np.zeros(300000, np.uint8))) == "bfff 8e21"
    assert hexwords(oracle.wah_encode_bits(np.ones(300000, np.uint8))) == "ffff ce21"
    b = np.zeros(500000, np.uint8)
    b[-1] = 1
    assert hexwords(oracle.wah_encode_bits(b)) == "bfff bfff 8237 0010"
    b = np.ones(31, np.uint8)
    b[15] = 0
    assert hexwords(oracle.wah_encode_bits(b)) == "c001 7ffe 0001"


@pytest.mark.parametrize("n", [1, 14, 15, 16, 29, 30, 31, 5008, 65535, 250000])
def test_wah_roundtrip_random(n):
    rng = np.random.default_rng(n)
    for p in (0.0, 0.001, 0.3, 0.97, 1.0):
        bits = (rng.random(n) < p).astype(np.uint8)
        # add long runs
        if n > 100:
            bits[n // 4:n // 2] = bits[0]
        w = oracle.wah_encode_bits(bits)
        back, used, ones = oracle.wah_extract(w, n)
        assert used == len(w)
        assert np.array_equal(back, bits)


def test_wah_saturation_boundaries():
    # a run of exactly 16383 / 16384 / 32767 groups (wah.hpp:396-401)
    for groups, expect in ((16383, "bfff"), (16384, "bfff 8001"), (32766, "bfff bfff"), (32767, "bfff bfff 8001")):
        assert hexwords(oracle.wah_encode_bits(np.zeros(groups * 15, np.uint8))) == expect


# ---- reference fixtures: size + sha256 anchors and decode round trip ----
@pytest.mark.parametrize("name", sorted(ANCHORS))
def test_micro_fixture_anchor(name, golden_dir):
    samples, recs = vcf_lite.read_vcf(os.path.join(golden_dir, name + ".vcf"))
    lines = [(r["gt"], r["n_allele"]) for r in recs]
    data = oracle.encode_file(lines, len(samples), maf=0.002, sample_names=samples)
    size, sha = ANCHORS[name]
    assert len(data) == size
    assert hashlib.sha256(data).hexdigest() == sha
    dec = oracle.decode_file(data, [r["n_allele"] for r in recs])
    for (gt, counts), r in zip(dec, recs):
        assert np.array_equal(gt, r["gt"])


def region_target_records(golden_dir):
    samples, recs = vcf_lite.read_vcf(os.path.join(golden_dir, "region_target.vcf"))
    assert len(samples) == 3202 and len(recs) == 6
    return samples, recs


@pytest.mark.parametrize("maf,block_len", [(0.002, 8192), (0.0, 8192), (0.002, 4), (0.0, 4)])
def test_region_target_fixture(maf, block_len, golden_dir):
    """test/cukinia_v4.conf:19: verify_v4.sh compresses test_region_target.bcf with --maf 0.002 (MAC threshold
    (size_t)(3202 * 2 * 0.002) = 12: the records with 48 and 37 ALT alleles are WAH lines, the other four sparse) and
    extracts with -t chr17:117980-117999; pass = the extracted genotypes equal bcftools' view of the input.  Here:
    oracle encode -> whole-file decode == the fixture's rows, and the -t subset (records by POS, the first one at
    117959 is outside) fetched by BM position through the reader == those rows.  --maf 0 makes every line a WAH line;
    block length 4 cuts the 6 records into two blocks (second block: the PBWT order restarts)."""
    samples, recs = region_target_records(golden_dir)
    lines = [(r["gt"], r["n_allele"]) for r in recs]
    n = len(samples)
    assert all(len(g) == 2 * n for g, _ in lines)
    assert oracle.default_phased_of(lines, n) == 1
    data = oracle.encode_file(lines, n, maf=maf, block_len=block_len, sample_names=samples)
    thr = int(float(2 * n) * maf)
    assert thr == (12 if maf else 0)
    assert struct.unpack_from("<I", data, 96)[0] == thr and data[14] == 2           # u16 A_T at 6404 haplotypes
    assert struct.unpack_from("<Q", data, 32)[0] == 2 * n and struct.unpack_from("<Q", data, 40)[0] == 6
    nal = [r["n_allele"] for r in recs]
    dec = oracle.decode_file(data, nal, block_len=block_len)
    for i, ((gt, counts), r) in enumerate(zip(dec, recs)):
        assert np.array_equal(gt, r["gt"]), "record %d" % i
        alt = int((((r["gt"] >> 1) - 1) == 1).sum())
        assert list(counts[:2]) == [2 * n - alt, alt]
    # -t chr17:117980-117999
    rd = oracle.Reader(data)
    picked = [i for i, r in enumerate(recs) if r["chrom"] == "chr17" and 117980 <= r["pos"] <= 117999]
    assert picked == [1, 2, 3, 4, 5]
    for i in picked:
        bm = ((i // block_len) << 15) | (i % block_len)
        gt, _ = rd.fill_genotype_array(2, bm)
        assert np.array_equal(gt, recs[i]["gt"]), "target record %d" % i
    # WAH / sparse split of the first block as the threshold dictates
    gt0 = 256 + 16
    nd = struct.unpack_from("<I", data, gt0 + 4)[0]
    d = dict(struct.unpack_from("<II", data, gt0 + 8 + 8 * k) for k in range(nd))
    sel = int(np.frombuffer(data, dtype="<u2", count=1, offset=gt0 + d[0x10])[0])
    n_first = min(block_len, 6)
    expect = 0
    for i in range(n_first):
        alt = int((((recs[i]["gt"] >> 1) - 1) == 1).sum())
        if min(alt, 2 * n - alt) > thr:
            expect |= 1 << i
    assert sel == expect


def test_worked_example_micro_missing(golden_dir):
    """SURVEY.md §9.4b, hand-verified against the reference output."""
    samples, recs = vcf_lite.read_vcf(os.path.join(golden_dir, "micro_missing.vcf"))
    data = oracle.encode_file([(r["gt"], r["n_allele"]) for r in recs], len(samples), maf=0.002,
                              sample_names=samples)
    assert len(data) == 544
    hdr = data[:256]
    assert struct.unpack_from("<I", hdr, 8)[0] == 5
    assert hdr[12] == 2 and hdr[14] == 2 and hdr[15] == 2 and hdr[16] == 4 and hdr[17] == 1
    assert struct.unpack_from("<Q", hdr, 72)[0] == 456  # index
    assert struct.unpack_from("<Q", hdr, 80)[0] == 464  # samples
    assert struct.unpack_from("<Q", data, 456)[0] == 256
    gt0 = 256 + 16
    n = struct.unpack_from("<I", data, gt0 + 4)[0]
    d = dict(struct.unpack_from("<II", data, gt0 + 8 + 8 * i) for i in range(n))
    assert d == {0: 12, 1: 12, 2: 2, 3: 1, 4: 2, 0x10: 104, 0x11: 104, 0x20: 106, 0x21: 146, 0x16: 150,
                 0x36: 152, 0x26: 0xFFFFFFFF}
    words = np.frombuffer(data, dtype="<u2", count=(146 - 104) // 2, offset=gt0 + 104)
    assert hexwords(words[:1]) == "0fcf"
    assert hexwords(words[1:]) == ("1014 0001 5430 000a 0200 8001 0400 8001 0100 0002 8001 0010 0001 8001 "
                                   "8001 0002 8001 0004 0001 8001")
    sp = np.frombuffer(data, dtype="<u2", count=2, offset=gt0 + 146)
    assert list(sp) == [0, 0]
    assert hexwords(np.frombuffer(data, dtype="<u2", count=1, offset=gt0 + 150)) == "0505"
    ms = np.frombuffer(data, dtype="<u2", count=15, offset=gt0 + 152)
    assert list(ms) == [1, 6, 1, 13, 6, 2, 3, 4, 5, 6, 7, 3, 12, 13, 15]


def test_dictionary_orders_match_this_libstdcxx(tmp_path):
    """The oracle hard-codes the 16 key orders of SURVEY.md §9.4.  Re-derive them here from the
    real std::unordered_map of this toolchain with the reference's insertion sequence
    (gt_block.hpp:464-510) so a libstdc++ change is noticed."""
    import subprocess
    src = tmp_path / "order.cpp"
    src.write_text(r'''
#include <unordered_map>
#include <cstdint>
#include <cstdio>
int main() {
  for (int idx = 0; idx < 16; ++idx) {
    bool m = idx & 1, e = idx & 2, p = idx & 4, h = idx & 8;
    std::unordered_map<uint32_t, uint32_t> d;
    d[0]=1; d[1]=1; d[2]=1; d[3]=1; d[4]=1;
    d[0x10]=1; d[0x11]=1; d[0x20]=1; d[0x21]=1;
    if (m) { d[0x16]=1; d[0x26]=1; d[0x36]=1; }
    if (e) { d[0x18]=1; d[0x28]=1; d[0x38]=1; }
    if (p) { d[0x17]=1; d[0x27]=1; }
    if (h) { d[0x12]=1; }
    for (auto& kv : d) std::printf("%02x ", kv.first);
    std::printf("\n");
  }
}''')
    exe = tmp_path / "order"
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-o", str(exe), str(src)])
    got = subprocess.check_output([str(exe)]).decode().strip().split("\n")
    expect = [
        "21 20 11 04 10 03 02 01 00",
        "26 16 21 20 11 04 10 03 36 02 01 00",
        "18 21 20 38 11 04 10 03 02 28 01 00",
        "38 28 00 01 02 36 10 11 03 20 04 21 16 26 18",
        "17 21 20 11 04 10 03 02 01 27 00",
        "27 00 01 02 36 10 11 03 20 04 21 16 26 17",
        "27 00 01 28 02 10 11 38 03 20 04 21 18 17",
        "27 17 38 28 00 01 02 36 10 11 03 20 04 21 16 26 18",
        "12 21 20 11 04 10 03 02 01 00",
        "12 26 16 21 20 11 04 10 03 36 02 01 00",
        "12 18 21 20 38 11 04 10 03 02 28 01 00",
        "12 38 28 00 01 02 36 10 11 03 20 04 21 16 26 18",
        "12 17 21 20 11 04 10 03 02 01 27 00",
        "12 27 00 01 02 36 10 11 03 20 04 21 16 26 17",
        "12 27 00 01 28 02 10 11 38 03 20 04 21 18 17",
        "12 27 17 38 28 00 01 02 36 10 11 03 20 04 21 16 26 18",
    ]
    assert [g.strip() for g in got] == expect


# ---- synthetic round trips through the oracle (multi-block, multi-allelic, u32 A_T) ----
def _random_lines(rng, n_samples, n_lines, multi=False, missing=False, eov=False, phase=False):
    lines = []
    for _ in range(n_lines):
        n_allele = 2 + (int(rng.integers(0, 3)) if multi else 0)
        p = float(rng.random()) ** 4 * 0.5
        al = (rng.random(2 * n_samples) < p).astype(np.int32)
        if n_allele > 2:
            extra = rng.random(2 * n_samples) < 0.02
            al[extra] = rng.integers(1, n_allele, size=int(extra.sum()))
        gt = ((al + 1) << 1)
        gt[1::2] |= 1
        if phase:
            flip = rng.random(n_samples) < 0.05
            gt[1::2][flip] ^= 1
        if missing:
            ms = rng.random(2 * n_samples) < 0.01
            gt[ms] = gt[ms] & 1
        if eov:
            ev = rng.random(n_samples) < 0.05
            gt[1::2][ev] = oracle.INT32_VECTOR_END
        lines.append((gt.astype(np.int32), n_allele))
    return lines


@pytest.mark.parametrize("n_samples,n_lines,block_len,kw", [
    (50, 300, 64, {}),
    (50, 200, 64, dict(multi=True)),
    (37, 150, 32, dict(missing=True, eov=True, phase=True, multi=True)),
    (2504, 40, 16, {}),
    (32767, 6, 4, {}),       # largest size with u16 A_T in header and blocks
    (70000, 4, 4, {}),       # u32 A_T everywhere
])
def test_oracle_roundtrip_synthetic(n_samples, n_lines, block_len, kw):
    rng = np.random.default_rng(n_samples * 1000 + n_lines)
    lines = _random_lines(rng, n_samples, n_lines, **kw)
    data = oracle.encode_file(lines, n_samples, maf=0.01, block_len=block_len)
    dec = oracle.decode_file(data, [n for _, n in lines], block_len=block_len)
    for (gt, counts), (src, n_allele) in zip(dec, lines):
        assert np.array_equal(gt, src)
        alleles = (src >> 1) - 1
        for k in range(1, n_allele):
            assert counts[k] == int(np.sum((alleles == k) & (src != oracle.INT32_VECTOR_END)))


@pytest.mark.parametrize("n_samples", [32768, 40000, 65535])
def test_oracle_refuses_at_mismatch_window(n_samples):
    """The reference's uint16 prefix array wraps in this window (gt_block.hpp:171,179); the
    restatement keeps uint32 and therefore refuses the window instead of claiming compatibility."""
    with pytest.raises(ValueError):
        oracle.Writer(n_samples)


def test_oracle_wah_encode_missing_strategy():
    rng = np.random.default_rng(7)
    lines = _random_lines(rng, 30, 120, missing=True, eov=True)
    data = oracle.encode_file(lines, 30, maf=0.01, block_len=50, wah_encode_missing=True)
    dec = oracle.decode_file(data, [n for _, n in lines], block_len=50)
    for (gt, _), (src, _) in zip(dec, lines):
        assert np.array_equal(gt, src)


@pytest.mark.parametrize("n_samples,n_lines,block_len,kw", [
    (37, 150, 32, dict(missing=True, eov=True, phase=True, multi=True)),
    (50, 200, 64, dict(missing=True)),
    (2504, 40, 16, dict(missing=True, eov=True)),
    (70000, 4, 4, dict(missing=True, eov=True, multi=True)),
])
def test_oracle_roundtrip_pbwt_weirdness(n_samples, n_lines, block_len, kw):
    """WS_PBWT_WAH (the version-4 default, gt_block.hpp:340-395; accessor_internals_new.hpp:300-340, 503-533): missing
    and end-of-vector lines stored as WAH lines permuted by a_weirdness, which is partitioned by "missing or end of
    vector" after every such line.  No fixture of the reference's exists for this strategy (its CLI cannot select
    it any more): the restatement is checked against itself and against the source rows only - parity unpinned."""
    rng = np.random.default_rng(n_samples * 7 + n_lines)
    lines = _random_lines(rng, n_samples, n_lines, **kw)
    data = oracle.encode_file(lines, n_samples, maf=0.01, block_len=block_len, wah_encode_missing=2)
    plain = oracle.encode_file(lines, n_samples, maf=0.01, block_len=block_len, wah_encode_missing=True)
    assert data != plain   # same sections, other words once a_weirdness has left the identity
    dec = oracle.decode_file(data, [n for _, n in lines], block_len=block_len)
    for (gt, counts), (src, n_allele) in zip(dec, lines):
        assert np.array_equal(gt, src)
