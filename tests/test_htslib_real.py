"""SURVEY 8f-1 / BASELINE configs[0] against the REAL htslib: `make -C xsqueezeit_amd/csrc htslib` (run by
__graft_entry__.build() wherever <htslib/vcf.h> and libhts exist) puts xsqueezeit_amd/libxsi_hip_htslib.so next to the
default library, and this file drives the reference's own end-to-end recipe (test/scripts/verify_v4.sh:98-129, run by
test/cukinia_v4.conf:19 as `verify_v4.sh --maf 0.002 -t chr17:117980-117999`) on the reference's own binary fixture,
committed as DATA: tests/golden/region_target.bcf (+ .csi; 6 records x 3202 samples of phased 1000 Genomes genotypes).

  -c   xsi_compress_bcf: the .xsi must be the oracle's file byte for byte (the oracle encodes the rows of
       tests/golden/region_target.vcf, the GT-only text copy of the same records made by make_region_target_fixture.py)
  -x   xsi_decompress_bcf -Ov: every record's genotype columns must be the fixture's
  -t   the same with the target region of the reference's test: exactly the records inside it

This image has no htslib, and neither has the GPU box the driver uses: every test here SKIPS there.  They exist so that
the first box that has it turns row f1 green without new code.  What runs everywhere is
tests/test_shim_mock.py (the same shim body against a stand-in for htslib over VCF text)."""
import ctypes
import os

import pytest

from xsqueezeit_amd import vcf_lite

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "xsqueezeit_amd", "libxsi_hip_htslib.so")

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(not os.path.exists(LIB), reason="no libxsi_hip_htslib.so: this box has no htslib (make -C xsqueezeit_amd/csrc htslib)")]


class Opt(ctypes.Structure):  # xsi_decompress_options, include/xsi_hip.h
    _fields_ = [("regions", ctypes.c_char_p), ("regions_is_file", ctypes.c_int), ("targets", ctypes.c_char_p),
                ("samples", ctypes.c_char_p), ("output_type", ctypes.c_char), ("fast_pipe", ctypes.c_int),
                ("no_header", ctypes.c_int), ("maf", ctypes.c_double), ("zstd_level", ctypes.c_uint32)]


@pytest.fixture(scope="module")
def hts():
    L = ctypes.CDLL(LIB)
    L.xsi_compress_bcf.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_double, ctypes.c_uint32, ctypes.c_uint32]
    L.xsi_decompress_bcf.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.POINTER(Opt)]
    L.c_xcf_nsamples.argtypes = [ctypes.c_char_p]
    L.xsi_hip_last_error.restype = ctypes.c_char_p
    assert L.xsi_htslib_shim_available() == 1
    return L


def _records(path):
    """(pos, genotype columns) of every record of a VCF text file."""
    out = []
    with open(path) as f:
        for l in f:
            if l.startswith("#"):
                continue
            t = l.rstrip("\n").split("\t")
            out.append((int(t[1]), [g.split(":")[0] for g in t[9:]]))
    return out


@pytest.mark.parametrize("block_len", [8192, 4])
def test_compress_decompress_region_target_bcf(hts, block_len, golden_dir, tmp_path):
    from oracle import oracle
    bcf = os.path.join(golden_dir, "region_target.bcf")
    want_samples, want = vcf_lite.read_vcf(os.path.join(golden_dir, "region_target.vcf"))
    assert hts.c_xcf_nsamples(bcf.encode()) == len(want_samples)
    xsi = str(tmp_path / "rt.xsi")
    assert hts.xsi_compress_bcf(bcf.encode(), xsi.encode(), 0.002, block_len, 0) == 0, hts.xsi_hip_last_error()
    ref = oracle.encode_file([(r["gt"], r["n_allele"]) for r in want], len(want_samples), maf=0.002, block_len=block_len,
                             sample_names=want_samples)
    assert open(xsi, "rb").read() == ref
    assert os.path.exists(xsi + "_var.bcf") and os.path.exists(xsi + "_var.bcf.csi")
    # -x -Ov: the genotypes of every record
    out = str(tmp_path / "rt.out.vcf")
    o = Opt(output_type=b"v")
    assert hts.xsi_decompress_bcf(xsi.encode(), out.encode(), ctypes.byref(o)) == 0, hts.xsi_hip_last_error()
    fixture = _records(os.path.join(golden_dir, "region_target.vcf"))
    assert _records(out) == fixture
    # -t chr17:117980-117999 (test/cukinia_v4.conf:19): the first record, at 117959, lies outside
    o = Opt(output_type=b"v", targets=b"chr17:117980-117999")
    assert hts.xsi_decompress_bcf(xsi.encode(), out.encode(), ctypes.byref(o)) == 0, hts.xsi_hip_last_error()
    inside = [r for r in fixture if 117980 <= r[0] <= 117999]
    assert 0 < len(inside) < len(fixture)
    assert _records(out) == inside
    # -r needs the index -c made
    o = Opt(output_type=b"v", regions=b"chr17:117980-117999")
    assert hts.xsi_decompress_bcf(xsi.encode(), out.encode(), ctypes.byref(o)) == 0, hts.xsi_hip_last_error()
    assert _records(out) == inside


def test_compress_with_zstd_and_reencode(hts, golden_dir, tmp_path):
    """--zstd on write (interfaces.hpp:288-315) and -Ox (gt_decompressor_new.hpp:241-273): the new pair decompresses to the same records."""
    bcf = os.path.join(golden_dir, "region_target.bcf")
    fixture = _records(os.path.join(golden_dir, "region_target.vcf"))
    xsi = str(tmp_path / "z.xsi")
    assert hts.xsi_compress_bcf(bcf.encode(), xsi.encode(), 0.002, 4, 7) == 0, hts.xsi_hip_last_error()
    out = str(tmp_path / "z.out.vcf")
    o = Opt(output_type=b"v")
    assert hts.xsi_decompress_bcf(xsi.encode(), out.encode(), ctypes.byref(o)) == 0, hts.xsi_hip_last_error()
    assert _records(out) == fixture
    x2 = str(tmp_path / "re.xsi")
    o = Opt(output_type=b"x", maf=0.01)
    assert hts.xsi_decompress_bcf(xsi.encode(), x2.encode(), ctypes.byref(o)) == 0, hts.xsi_hip_last_error()
    o = Opt(output_type=b"v")
    assert hts.xsi_decompress_bcf(x2.encode(), out.encode(), ctypes.byref(o)) == 0, hts.xsi_hip_last_error()
    assert _records(out) == fixture
