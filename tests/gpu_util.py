"""Helpers for the -m gpu parity tests: device buffers via torch, calls through the C ABI."""
import ctypes
import struct

import numpy as np

from xsqueezeit_amd import binding, synth


def torch_mod():
    import torch
    return torch


_CTX = None
_STREAM = None


def ctx():
    """One context for the whole test session, sharing ONE explicit stream with torch: torch's
    default stream has handle 0, which the C ABI reads as "create your own stream", and two
    unordered streams would race (torch fills / copies vs our kernels)."""
    global _CTX, _STREAM
    if _CTX is None:
        torch = torch_mod()
        assert torch.cuda.is_available(), "GPU tests need a GPU"
        torch.cuda.init()
        _STREAM = torch.cuda.Stream()
        torch.cuda.set_stream(_STREAM)
        assert _STREAM.cuda_stream != 0
        _CTX = binding.Context(0, _STREAM.cuda_stream)
    return _CTX


def dev_u8(arr):
    torch = torch_mod()
    ctx()
    return torch.from_numpy(np.ascontiguousarray(arr).view(np.uint8).reshape(-1)).cuda()


def accessor_array(a, shape, fill=0):
    """An int32 numpy array over page-locked memory from xsi_accessor_alloc_array (what xsi_accessor_register_array
    takes); free it with accessor_array_free before the accessor is closed, or leave it to xsi_accessor_close."""
    L = binding.lib()
    n = int(np.prod(shape))
    p = ctypes.c_void_p()
    binding.check(L.xsi_accessor_alloc_array(a, n, ctypes.byref(p)))
    arr = np.ctypeslib.as_array(ctypes.cast(p, ctypes.POINTER(ctypes.c_int32)), shape=(n,)).reshape(shape)
    arr[...] = fill
    return arr


def accessor_array_free(a, arr):
    binding.check(binding.lib().xsi_accessor_free_array(a, arr.ctypes.data))


def dev_empty(nbytes):
    torch = torch_mod()
    ctx()
    return torch.empty(int(nbytes), dtype=torch.uint8, device="cuda")


def params(n_samples, block_len=8192, mac_thr=0, default_phased=1, wah_encode_missing=0):
    return binding.EncodeParams(n_samples, block_len, mac_thr, default_phased, wah_encode_missing, 0)  # zstd_level=0


def encode_packed(packed, n_haps, p):
    """packed: uint8 [n_lines, stride].  Returns (blocks_region bytes, block_offsets, result)."""
    torch = torch_mod()
    L = binding.lib()
    n_lines, stride = packed.shape
    d_bits = dev_u8(packed)
    cap = int(L.xsi_hip_encode_bound(ctypes.byref(p), n_lines, n_lines))
    d_out = dev_empty(cap)
    n_blocks = (n_lines + p.block_len - 1) // p.block_len
    d_off = torch.zeros(n_blocks, dtype=torch.int64, device="cuda")
    res = binding.EncodeResult()
    binding.check(L.xsi_hip_encode_packed(ctx().handle, ctypes.byref(p), d_bits.data_ptr(), n_lines, stride,
                                          d_out.data_ptr(), cap, d_off.data_ptr(), ctypes.byref(res)))
    region = d_out[:res.blocks_bytes].cpu().numpy().tobytes()
    return region, d_off.cpu().numpy().astype(np.uint64), res


def assemble_file(region, offsets, p, n_lines, num_variants, sample_names, max_ploidy=2):
    """Host-side rest of XsiFactoryExt::finalize_file (xsi_factory.hpp:558-605) around a GPU-made
    blocks region: pad to 8, u64 index, sample names, header."""
    body = bytearray(region)
    while (256 + len(body)) % 8:
        body.append(0)
    indices_offset = 256 + len(body)
    body += np.asarray(offsets, dtype="<u8").tobytes()
    samples_offset = 256 + len(body)
    for s in sample_names:
        body += s.encode() + b"\0"
    hf = binding.HeaderFields(p.n_samples, max_ploidy, p.block_len, p.mac_threshold, p.default_phased, 0,
                              num_variants, n_lines, indices_offset, samples_offset)
    hdr = (ctypes.c_uint8 * 256)()
    binding.check(binding.lib().xsi_hip_make_header(ctypes.byref(hf), hdr))
    return bytes(hdr) + bytes(body)


def decode_packed(file_bytes, n_haps, stride, first_block=0, n_blocks=None, max_rows=None):
    torch = torch_mod()
    L = binding.lib()
    d_file = dev_u8(np.frombuffer(file_bytes, dtype=np.uint8))
    if n_blocks is None:
        io, so = struct.unpack_from("<QQ", file_bytes, 72)
        n_blocks = (so - io) // 8 - first_block
    if max_rows is None:
        max_rows = struct.unpack_from("<Q", file_bytes, 40)[0]
    d_out = dev_empty(max_rows * stride)
    d_cnt = torch.zeros(max_rows, dtype=torch.int32, device="cuda")
    rows = ctypes.c_uint64(0)
    binding.check(L.xsi_hip_decode_packed(ctx().handle, d_file.data_ptr(), len(file_bytes), first_block, n_blocks,
                                          d_out.data_ptr(), stride, max_rows, ctypes.byref(rows), d_cnt.data_ptr()))
    n = rows.value
    out = d_out[:n * stride].cpu().numpy().reshape(n, stride)
    return out, d_cnt[:n].cpu().numpy()


def oracle_file_from_bits(bits01, p, sample_names=None):
    from oracle import oracle
    n_samples = bits01.shape[1] // 2
    w = oracle.Writer(n_samples, p.block_len, p.mac_threshold, p.default_phased, bool(p.wah_encode_missing),
                      sample_names)
    gt = synth.bits_to_gt(bits01, p.default_phased)
    for r in gt:
        w.append(r, 2)
    return w.finalize(2)


def numpy_chain_yrows(bits01, block_len, mac_thr):
    """Reference PBWT chain in numpy: list of (line, permuted bits) for the WAH lines."""
    n_lines, N = bits01.shape
    out = []
    a = None
    for l in range(n_lines):
        if l % block_len == 0:
            a = np.arange(N)
        x = bits01[l]
        c = int(x.sum())
        if min(c, N - c) > mac_thr:
            y = x[a]
            out.append((l, y.copy()))
            a = np.concatenate([a[y == 0], a[y == 1]])
    return out


# ---------------- general (int32 genotype rows) path ----------------
def rows_matrix(lines, n_samples):
    """list of (gt, n_allele) -> (int32 [n_lines, 2*n_samples] padded, ngt, n_allele)."""
    N = 2 * n_samples
    m = np.zeros((len(lines), N), dtype=np.int32)
    ngt = np.zeros(len(lines), dtype=np.uint32)
    nal = np.zeros(len(lines), dtype=np.uint32)
    for i, (gt, na) in enumerate(lines):
        m[i, :len(gt)] = gt
        ngt[i] = len(gt)
        nal[i] = na
    return m, ngt, nal


def encode_gt(lines, n_samples, p):
    torch = torch_mod()
    L = binding.lib()
    m, ngt, nal = rows_matrix(lines, n_samples)
    n_lines = len(lines)
    n_bin = int((nal - 1).sum())
    d_gt = torch.from_numpy(m).cuda()
    cap = int(L.xsi_hip_encode_gt_bound(ctypes.byref(p), n_lines, n_bin))
    d_out = dev_empty(cap)
    n_blocks = (n_lines + p.block_len - 1) // p.block_len
    d_off = torch.zeros(n_blocks, dtype=torch.int64, device="cuda")
    res = binding.EncodeResult()
    binding.check(L.xsi_hip_encode_gt(ctx().handle, ctypes.byref(p), d_gt.data_ptr(), m.shape[1], n_lines,
                                      ngt.ctypes.data, nal.ctypes.data, d_out.data_ptr(), cap, d_off.data_ptr(),
                                      ctypes.byref(res)))
    region = d_out[:res.blocks_bytes].cpu().numpy().tobytes()
    return region, d_off.cpu().numpy().astype(np.uint64), res


def decode_gt(file_bytes, n_alleles, max_alleles=None):
    """Decode every block of the image; returns (rows list, allele counts [n_lines, max_alleles])."""
    torch = torch_mod()
    L = binding.lib()
    d_file = dev_u8(np.frombuffer(file_bytes, dtype=np.uint8))
    io, so = struct.unpack_from("<QQ", file_bytes, 72)
    n_blocks = (so - io) // 8
    num_samples = struct.unpack_from("<Q", file_bytes, 112)[0]
    N = 2 * num_samples
    nal = np.asarray(n_alleles, dtype=np.uint32)
    n_lines = len(nal)
    if max_alleles is None:
        max_alleles = int(nal.max())
    d_out = torch.zeros((n_lines, N), dtype=torch.int32, device="cuda")
    d_cnt = torch.zeros((n_lines, max_alleles), dtype=torch.int64, device="cuda")
    ngt = np.zeros(n_lines, dtype=np.uint32)
    binding.check(L.xsi_hip_decode_gt(ctx().handle, d_file.data_ptr(), len(file_bytes), 0, n_blocks, nal.ctypes.data,
                                      n_lines, d_out.data_ptr(), N, ngt.ctypes.data, d_cnt.data_ptr(), max_alleles))
    out = d_out.cpu().numpy()
    return [out[i, :ngt[i]] for i in range(n_lines)], d_cnt.cpu().numpy()


def num_variants(lines):
    return sum(na - 1 for _, na in lines)
