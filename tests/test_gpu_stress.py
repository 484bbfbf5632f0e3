"""Random-shape parity under the driver's eyes: a fixed-seed subset of tests/stress_parity.py as `-m gpu`
tests - every case draws its shape (haplotypes, lines, block length, MAC threshold, densities, side channels) from its own
seeded generator, encodes through the C ABI, compares the bytes with the oracle's file and the decode with the source -
plus the two shapes the random sweep leaves out on purpose:
  * multi-allelic AND fully haploid lines in one block (the reference writes KEY_LINE_HAPLOID per BCF line,
    gt_block.hpp:219-224, and reads it per binary line, accessor_internals_new.hpp:116,165,204: SURVEY 9.6.2), pinned
    bug-compatibly: GPU bytes == oracle bytes, GPU decode == the oracle READER's rows (not the input);
  * wah_encode_missing = 1 (WS_WAH) above 131 072 haplotypes (u32 A_T in header and blocks).
"""
import os
import sys
import tempfile

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from oracle import oracle  # noqa: E402
from test_oracle import _random_lines  # noqa: E402

pytestmark = pytest.mark.gpu

SEED = 20261005


def _cases():
    out = []
    for i in range(14):
        out.append(("packed", i, False))
    for i in range(10):
        out.append(("general", i, False))
    for i in range(8):
        out.append(("file", i, False))
    for i in range(10):
        out.append(("packed", i, True))
    return out


@pytest.fixture(scope="module")
def stress_tmp():
    with tempfile.TemporaryDirectory(prefix="xsi_stress_") as d:
        yield d


@pytest.mark.parametrize("kind,idx,long_rows", _cases(),
                         ids=["%s%s-%02d" % (k, "-long-rows" if lr else "", i) for k, i, lr in _cases()])
def test_random_shape_parity(kind, idx, long_rows, stress_tmp):
    """One random case: GPU .xsi bytes == oracle bytes and GPU decode == source (packed rows + ALT counts, int32 rows, or
    random-order accessor reads of a written file).  The long-row cases run 140 000 .. 524 288 haplotypes with a density of
    its own per line: both exchange forms of k_chain_rank_enc_multi, 3 to 8 workgroups per block, and the long-row decode."""
    import stress_parity
    rng = np.random.default_rng([SEED, {"packed": 1, "general": 2, "file": 3}[kind], int(long_rows), idx])
    ok, ok2, line = stress_parity.run_case(kind, rng, idx, cells=16_000_000 if not long_rows else 30_000_000,
                                           max_lines=300 if not long_rows else 120, long_rows=long_rows, tmpdir=stress_tmp)
    print(line)
    assert ok, "bytes differ from the oracle's: " + line
    assert ok2, "decode differs from the source: " + line


def _haploid_line(rng, n, p):
    al = (rng.random(n) < p).astype(np.int32)
    return (((al + 1) << 1).astype(np.int32), 2)


def _multiallelic_line(rng, n, n_allele, p_alt, p_extra):
    """Diploid phased line with alleles 0 .. n_allele - 1: ALT 1 at frequency p_alt, the further ALTs at p_extra each."""
    al = (rng.random(2 * n) < p_alt).astype(np.int32)
    for k in range(2, n_allele):
        al[rng.random(2 * n) < p_extra] = k
    gt = (al + 1) << 1
    gt[1::2] |= 1
    return (gt.astype(np.int32), n_allele)


@pytest.mark.parametrize("order", ["haploid_first", "multiallelic_first"])
@pytest.mark.parametrize("n", [300, 2504, 70000])
def test_haploid_flags_misaligned_by_multiallelic_lines(order, n):
    """SURVEY 9.6.2, kept bug-compatibly.  The block has 12 BCF lines = 13 or 14 binary lines (one 15-bit group of flag
    bits either way: reading the vector per binary line stays inside what was written).
    haploid_first: the fully haploid line stands in front of every multi-allelic one, so BCF and binary index agree at
      its flag: the block decodes back to its input.
    multiallelic_first: a tri-allelic line stands directly in front, the flag written at the haploid line's BCF index is read at
      that BINARY index - the tri-allelic line's second ALT is taken for the haploid line and the haploid line for a
      diploid one.  Both are kept sparse here (few carriers, MAC threshold above them), so the misreading stays inside the
      two lines (sparse lines carry their own counts and never touch the prefix array).  The decode is then NOT the input:
      it must be what the oracle's reader - the restatement of the reference's - makes of the same bytes."""
    import gpu_util as G
    rng = np.random.default_rng(900 + n + len(order))
    thr = max(3, n // 50)
    rare = 0.3 * thr / (2 * n)          # expected carriers well under the MAC threshold: a sparse line
    common = [_multiallelic_line(rng, n, 2, 0.2 + 0.05 * i, 0.0) for i in range(9)]  # WAH lines: the chain runs
    tri = _multiallelic_line(rng, n, 3, 0.3, rare)       # ALT 1 a WAH line, ALT 2 sparse
    tri2 = _multiallelic_line(rng, n, 3 if n != 2504 else 2, 0.1, rare)
    hap = _haploid_line(rng, n, rare * 2)                 # sparse fully haploid line
    if order == "haploid_first":
        lines = common[:2] + [hap] + common[2:4] + [tri] + common[4:7] + [tri2] + common[7:]
    else:
        # the tri-allelic line DIRECTLY in front: flag bit (BCF index of the haploid line) = binary index of its second ALT
        lines = common[:2] + [tri, hap] + common[2:7] + [tri2] + common[7:]
    assert len(lines) == 12
    n_bin = sum(na - 1 for _, na in lines)
    assert (len(lines) + 14) // 15 == (n_bin + 14) // 15 == 1
    dp = oracle.default_phased_of(lines, n)
    ref = oracle.encode_file(lines, n, block_len=12, mac_thr=thr, default_phased=dp)
    p = G.params(n, 12, thr, dp)
    region, offsets, res = G.encode_gt(lines, n, p)
    names = ["S%d" % i for i in range(n)]
    got = G.assemble_file(region, offsets, p, len(lines), G.num_variants(lines), names, 2)
    assert got == ref, "GPU bytes differ from the oracle's"
    nal = [na for _, na in lines]
    want = oracle.decode_file(ref, nal, block_len=12)
    rows, counts = G.decode_gt(got, nal)
    for i, (gt_o, cnt_o) in enumerate(want):
        assert len(rows[i]) == len(gt_o), "line %d: %d values, the oracle's reader has %d" % (i, len(rows[i]), len(gt_o))
        assert np.array_equal(rows[i], gt_o), "line %d differs from the oracle reader's row" % i
        assert [int(x) for x in counts[i][:nal[i]]] == [int(x) for x in cnt_o], "allele counts of line %d" % i
    same_as_input = all(len(rows[i]) == len(lines[i][0]) and np.array_equal(rows[i], lines[i][0]) for i in range(len(lines)))
    if order == "haploid_first":
        assert same_as_input
    else:
        assert not same_as_input  # the reference's own misalignment, reproduced


@pytest.mark.parametrize("n,block_len", [(66000, 5), (70001, 8192)])
def test_wah_encode_missing_above_131072_haplotypes(n, block_len):
    """--wah-encode-missing (WS_WAH, gt_block.hpp:340-371,574-584) with u32 A_T: missing and end-of-vector lines as
    unpermuted WAH16 lines of 2 n > 131 072 bits next to u32 sparse lists; bytes == oracle, decode == source."""
    import gpu_util as G
    rng = np.random.default_rng(n)
    lines = _random_lines(rng, n, 14, multi=True, missing=True, eov=True, phase=True)
    dp = oracle.default_phased_of(lines, n)
    thr = n // 400
    ref = oracle.encode_file(lines, n, block_len=block_len, mac_thr=thr, default_phased=dp, wah_encode_missing=True)
    assert ref[14] == 4  # aet_bytes
    p = G.params(n, block_len, thr, dp, wah_encode_missing=1)
    region, offsets, res = G.encode_gt(lines, n, p)
    names = ["S%d" % i for i in range(n)]
    got = G.assemble_file(region, offsets, p, len(lines), G.num_variants(lines), names, 2)
    assert got == ref
    nal = [na for _, na in lines]
    rows, counts = G.decode_gt(got, nal)
    for i, (src, na) in enumerate(lines):
        assert np.array_equal(rows[i][:len(src)], src), i
        alleles = (src >> 1) - 1
        for k in range(1, na):
            assert counts[i][k] == int(np.sum((alleles == k) & (src != oracle.INT32_VECTOR_END)))
