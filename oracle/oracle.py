"""ctypes wrapper around oracle/liboracle.so — CPU ORACLE, test infrastructure only.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The product package (xsqueezeit_amd) never does.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

INT32_MISSING = -(2 ** 31)
INT32_VECTOR_END = -(2 ** 31) + 1


def build():
    """Compile liboracle.so from xsi_oracle.c (gcc, seconds)."""
    subprocess.check_call(["make", "-s", "-C", _HERE, "liboracle.so"])


def lib():
    global _LIB
    if _LIB is not None:
        return _LIB
    path = os.path.join(_HERE, "liboracle.so")
    if not os.path.exists(path):
        build()
    L = ctypes.CDLL(path)
    c = ctypes
    L.xo_wah_encode_bits.restype = c.c_size_t
    L.xo_wah_encode_bits.argtypes = [c.c_void_p, c.c_size_t, c.c_void_p]
    L.xo_wah_extract.restype = c.c_size_t
    L.xo_wah_extract.argtypes = [c.c_void_p, c.c_size_t, c.c_void_p, c.POINTER(c.c_size_t)]
    L.xo_pbwt_sort.restype = None
    L.xo_pbwt_sort.argtypes = [c.c_void_p, c.c_void_p, c.c_size_t, c.c_void_p, c.c_int32, c.c_uint32]
    L.xo_writer_new.restype = c.c_void_p
    L.xo_writer_new.argtypes = [c.c_uint32, c.c_uint32, c.c_uint32, c.c_int32, c.c_int, c.POINTER(c.c_char_p)]
    L.xo_writer_append.restype = c.c_int
    L.xo_writer_append.argtypes = [c.c_void_p, c.c_void_p, c.c_int32, c.c_int32]
    L.xo_writer_append_rows.restype = c.c_int
    L.xo_writer_append_rows.argtypes = [c.c_void_p, c.c_void_p, c.c_size_t, c.c_size_t, c.c_int32, c.c_int32]
    L.xo_reader_fill_rows.restype = c.c_int64
    L.xo_reader_fill_rows.argtypes = [c.c_void_p, c.c_void_p, c.c_size_t, c.c_uint64, c.c_size_t, c.c_uint32]
    L.xo_writer_finalize.restype = c.c_int
    L.xo_writer_finalize.argtypes = [c.c_void_p, c.c_uint32, c.POINTER(c.c_void_p), c.POINTER(c.c_size_t)]
    L.xo_writer_free.restype = None
    L.xo_writer_free.argtypes = [c.c_void_p]
    L.xo_free.restype = None
    L.xo_free.argtypes = [c.c_void_p]
    L.xo_reader_open.restype = c.c_void_p
    L.xo_reader_open.argtypes = [c.c_void_p, c.c_size_t]
    L.xo_reader_fill_genotype_array.restype = c.c_int64
    L.xo_reader_fill_genotype_array.argtypes = [c.c_void_p, c.c_void_p, c.c_size_t, c.c_uint32, c.c_uint64]
    L.xo_reader_fill_allele_counts.restype = c.c_int
    L.xo_reader_fill_allele_counts.argtypes = [c.c_void_p, c.c_uint32, c.c_uint64]
    L.xo_reader_allele_counts.restype = c.POINTER(c.c_uint64)
    L.xo_reader_allele_counts.argtypes = [c.c_void_p]
    L.xo_reader_hap_samples.restype = c.c_uint64
    L.xo_reader_hap_samples.argtypes = [c.c_void_p]
    L.xo_reader_num_samples.restype = c.c_uint64
    L.xo_reader_num_samples.argtypes = [c.c_void_p]
    L.xo_reader_close.restype = None
    L.xo_reader_close.argtypes = [c.c_void_p]
    _LIB = L
    return L


def wah_encode_bits(bits01):
    bits01 = np.ascontiguousarray(bits01, dtype=np.uint8)
    out = np.empty(bits01.size // 15 + 2, dtype=np.uint16)
    n = lib().xo_wah_encode_bits(bits01.ctypes.data, bits01.size, out.ctypes.data)
    return out[:n].copy()


def wah_extract(words, n):
    words = np.ascontiguousarray(words, dtype=np.uint16)
    bits = np.zeros(n + 15, dtype=np.uint8)
    ones = ctypes.c_size_t(0)
    used = lib().xo_wah_extract(words.ctypes.data, n, bits.ctypes.data, ctypes.byref(ones))
    return bits[:n].copy(), int(used), int(ones.value)


def default_phased_of(lines, n_samples, limit=3):
    """seek_default_phased (xcf.cpp:811-836) on (gt, n_allele) rows."""
    counts = [0, 0]
    for gt, _ in lines[:limit]:
        ploidy = len(gt) // n_samples
        if ploidy == 1:
            return 0
        for i in range(n_samples):
            counts[int(gt[i * ploidy + 1]) & 1] += 1
    return 0 if counts[0] > counts[1] else 1


class Writer:
    """XsiFactoryExt-shaped: append(gt_row, n_allele) ... finalize(max_ploidy) -> bytes."""

    def __init__(self, n_samples, block_len=8192, mac_thr=0, default_phased=1, wah_encode_missing=False,
                 sample_names=None):
        if 32768 <= n_samples <= 65535:
            raise ValueError("%d samples: the reference's A_T mismatch window (uint16 prefix array wraps, "
                             "gt_block.hpp:171,179); neither the oracle nor the product encodes it" % n_samples)
        names = sample_names or ["S%d" % i for i in range(n_samples)]
        assert len(names) == n_samples
        arr = (ctypes.c_char_p * max(n_samples, 1))(*[s.encode() for s in names])
        self._h = lib().xo_writer_new(n_samples, block_len, mac_thr, default_phased,
                                      (2 if wah_encode_missing == 2 else (1 if wah_encode_missing else 0)), arr)
        if not self._h:
            raise MemoryError("xo_writer_new")

    def append(self, gt, n_allele=2):
        gt = np.ascontiguousarray(gt, dtype=np.int32)
        rc = lib().xo_writer_append(self._h, gt.ctypes.data, gt.size, n_allele)
        if rc:
            raise ValueError("xo_writer_append rc=%d" % rc)

    def append_rows(self, gt2d, n_allele=2):
        gt2d = np.ascontiguousarray(gt2d, dtype=np.int32)
        rc = lib().xo_writer_append_rows(self._h, gt2d.ctypes.data, gt2d.shape[0], gt2d.shape[1], gt2d.shape[1],
                                         n_allele)
        if rc:
            raise ValueError("xo_writer_append_rows rc=%d" % rc)

    def finalize(self, max_ploidy=2):
        p = ctypes.c_void_p()
        n = ctypes.c_size_t()
        rc = lib().xo_writer_finalize(self._h, max_ploidy, ctypes.byref(p), ctypes.byref(n))
        if rc:
            raise RuntimeError("xo_writer_finalize rc=%d" % rc)
        data = ctypes.string_at(p, n.value)
        lib().xo_free(p)
        return data

    def __del__(self):
        if getattr(self, "_h", None):
            lib().xo_writer_free(self._h)
            self._h = None


def encode_file(lines, n_samples, maf=0.001, block_len=8192, sample_names=None, wah_encode_missing=False,
                default_phased=None, mac_thr=None):
    """GtCompressorStream flow (gt_compressor_new.hpp:84-142) on a list of (gt_row, n_allele)."""
    first_ploidy = len(lines[0][0]) // n_samples
    if mac_thr is None:
        mac_thr = int(float(n_samples * first_ploidy) * maf)
    if default_phased is None:
        default_phased = default_phased_of(lines, n_samples)
    w = Writer(n_samples, block_len, mac_thr, default_phased, wah_encode_missing, sample_names)
    ploidy = first_ploidy
    for gt, n_allele in lines:
        ploidy = max(ploidy, len(gt) // n_samples)
        w.append(gt, n_allele)
    return w.finalize(ploidy)


class Reader:
    """AccessorInternalsNewTemplate-shaped reader over an in-memory .xsi image."""

    def __init__(self, data):
        self._buf = np.frombuffer(bytes(data), dtype=np.uint8).copy()
        self._h = lib().xo_reader_open(self._buf.ctypes.data, self._buf.size)
        if not self._h:
            raise ValueError("xo_reader_open failed (magic/version/zstd)")
        self.hap_samples = int(lib().xo_reader_hap_samples(self._h))
        self.num_samples = int(lib().xo_reader_num_samples(self._h))
        self._n = max(self.hap_samples, 2 * self.num_samples)

    def fill_genotype_array(self, n_alleles, bm):
        gt = np.empty(self._n, dtype=np.int32)
        n = lib().xo_reader_fill_genotype_array(self._h, gt.ctypes.data, gt.size, n_alleles, bm)
        if n < 0:
            raise ValueError("fill_genotype_array rc=%d" % n)
        counts = np.ctypeslib.as_array(lib().xo_reader_allele_counts(self._h), shape=(n_alleles,)).copy()
        return gt[:n].copy(), counts

    def fill_rows(self, first_line, n_rows, block_len, out=None):
        """n_rows consecutive bi-allelic diploid lines -> int32 [n_rows, N]."""
        if out is None:
            out = np.empty((n_rows, self._n), dtype=np.int32)
        n = lib().xo_reader_fill_rows(self._h, out.ctypes.data, out.shape[1], first_line, n_rows, block_len)
        if n < 0:
            raise ValueError("fill_rows rc=%d" % n)
        return out

    def fill_allele_counts(self, n_alleles, bm):
        rc = lib().xo_reader_fill_allele_counts(self._h, n_alleles, bm)
        if rc:
            raise ValueError("fill_allele_counts rc=%d" % rc)
        return np.ctypeslib.as_array(lib().xo_reader_allele_counts(self._h), shape=(n_alleles,)).copy()

    def __del__(self):
        if getattr(self, "_h", None):
            lib().xo_reader_close(self._h)
            self._h = None


def decode_file(data, n_alleles_per_line, block_len=8192):
    """Walk every BCF line in order (the -x loop, gt_decompressor_new.hpp:157-206) given the
    per-line allele numbers; BM = block<<15 | binary offset (xcf.cpp:685-703)."""
    r = Reader(data)
    out = []
    block = 0
    offset = 0
    for i, n_allele in enumerate(n_alleles_per_line):
        if i and i % block_len == 0:
            block += 1
            offset = 0
        gt, counts = r.fill_genotype_array(n_allele, (block << 15) | offset)
        out.append((gt, counts))
        offset += n_allele - 1
    return out
