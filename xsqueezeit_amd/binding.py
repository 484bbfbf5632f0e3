"""ctypes binding of libxsi_hip.so (include/xsi_hip.h).  Raises if the library is missing."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("XSI_LIB_PATH") or os.path.join(_HERE, "libxsi_hip.so")  # (XSI_LIB_PATH: A/B runs of two builds on one box)

XSI_OK = 0
XSI_ERR_ARG = -1
XSI_ERR_HIP = -2
XSI_ERR_CAPACITY = -3
XSI_ERR_FORMAT = -4
XSI_ERR_UNSUPPORTED = -5
XSI_ERR_IO = -6


class XsiError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("xsi_hip error %d: %s" % (code, msg))
        self.code = code


class EncodeParams(ctypes.Structure):
    _fields_ = [("n_samples", ctypes.c_uint32), ("block_len", ctypes.c_uint32), ("mac_threshold", ctypes.c_uint32),
                ("default_phased", ctypes.c_int32), ("wah_encode_missing", ctypes.c_uint32),
                ("zstd_level", ctypes.c_uint32)]


class EncodeResult(ctypes.Structure):
    _fields_ = [("n_blocks", ctypes.c_uint64), ("blocks_bytes", ctypes.c_uint64),
                ("n_binary_lines", ctypes.c_uint64), ("n_wah_lines", ctypes.c_uint64),
                ("max_ploidy", ctypes.c_uint32), ("last_block_bytes", ctypes.c_uint32)]


class HeaderFields(ctypes.Structure):
    _fields_ = [("n_samples", ctypes.c_uint32), ("max_ploidy", ctypes.c_uint32), ("block_len", ctypes.c_uint32),
                ("mac_threshold", ctypes.c_uint32), ("default_phased", ctypes.c_int32), ("zstd", ctypes.c_uint32),
                ("num_variants", ctypes.c_uint64), ("xcf_entries", ctypes.c_uint64),
                ("indices_offset", ctypes.c_uint64), ("samples_offset", ctypes.c_uint64)]


# every symbol include/xsi_hip.h declares; tests check the library exports all of them
SYMBOLS = [
    "xsi_hip_abi_version", "xsi_hip_last_error", "xsi_hip_ctx_create", "xsi_hip_ctx_destroy",
    "xsi_hip_ctx_synchronize", "xsi_hip_ctx_workspace_bytes", "xsi_hip_ctx_set_timing",
    "xsi_hip_ctx_get_timing", "xsi_hip_stage_name", "xsi_hip_encode_bound", "xsi_hip_encode_packed",
    "xsi_hip_encode_gt", "xsi_hip_encode_gt_bound", "xsi_hip_make_header", "xsi_hip_decode_packed", "xsi_hip_decode_gt", "xsi_hip_decode_counts", "xsi_accessor_fill_allele_counts",
    "xsi_hip_synth_packed", "xsi_hip_debug_chain_encode", "xsi_writer_open", "xsi_writer_append", "xsi_writer_row_buffer", "xsi_writer_commit_row",
    "xsi_writer_finalize", "xsi_writer_close", "xsi_accessor_open", "xsi_accessor_fill_genotype_array",
    "xsi_accessor_get_genotypes", "xsi_accessor_genotypes_view", "xsi_accessor_allele_counts", "xsi_accessor_get_internal_access", "xsi_accessor_hap_samples",
    "xsi_accessor_num_samples", "xsi_accessor_sample_name", "xsi_accessor_close",
    "xsi_accessor_set_cache_bytes", "xsi_accessor_cache_stats", "xsi_file_num_samples", "xsi_hip_decode_dot", "xsi_hip_decode_dot_gt",
    "xsi_hip_ctx_set_workspace_budget", "xsi_hip_chain_kernel", "xsi_mac_threshold", "xsi_default_phased",
    "xsi_bm_init", "xsi_bm_next", "xsi_accessor_set_sample_subset", "xsi_accessor_fill_selected_genotypes",
    "xsi_hip_reencode", "xsi_hip_ctx_chain_fallbacks", "xsi_hip_ctx_set_block_sizes_out", "xsi_accessor_readahead_stats",
    "xsi_hip_shard_blocks", "xsi_hip_shard_of_block", "xsi_hip_comm_unique_id", "xsi_hip_comm_create", "xsi_hip_comm_destroy",
    "xsi_hip_comm_world", "xsi_hip_comm_rank", "xsi_hip_gather_block_streams", "xsi_hip_comm_wait", "xsi_htslib_shim_available", "xsi_debug_pack_bit_row",
    "xsi_hip_encode_packed_counted", "xsi_hip_count_packed_rows",
    "xsi_accessor_register_array", "xsi_accessor_unregister_array", "xsi_accessor_alloc_array", "xsi_accessor_free_array", "xsi_hip_gather_block_streams_round",
    "xsi_compress_bcf", "xsi_decompress_bcf", "xsi_hip_ctx_reencode_ranges",
    "xsi_accessor_get_genotypes_batch", "xsi_accessor_prefix_stats",
]


class InternalAccess(ctypes.Structure):
    _fields_ = [("position", ctypes.c_uint64), ("n_alleles", ctypes.c_uint32), ("sparse_bytes", ctypes.c_uint32),
                ("wah_bytes", ctypes.c_uint32), ("a_bytes", ctypes.c_uint32), ("default_allele", ctypes.c_int32),
                ("n_a", ctypes.c_uint32), ("image", ctypes.c_void_p), ("image_len", ctypes.c_uint64)]


class BmState(ctypes.Structure):
    _fields_ = [("line", ctypes.c_uint64), ("block", ctypes.c_uint64), ("offset", ctypes.c_uint64)]


_LIB = None


def lib():
    """Load libxsi_hip.so.  No fallback: a missing library is an error."""
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(LIB_PATH):
        raise ImportError("%s is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(hipcc --offload-arch=gfx950); there is no CPU fallback" % LIB_PATH)
    # The torch wheel carries its own HIP runtime.  With torch in the process (it provides the device
    # buffers and streams of this harness) that runtime has to be mapped before libxsi_hip.so pulls in the
    # system one, or no device is found.  A C / C++ caller has no torch and nothing to order.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = ctypes.CDLL(LIB_PATH)
    c = ctypes
    vp, u64, u32, i32 = c.c_void_p, c.c_uint64, c.c_uint32, c.c_int32
    L.xsi_hip_abi_version.restype = c.c_int
    L.xsi_hip_last_error.restype = c.c_char_p
    L.xsi_hip_ctx_create.restype = c.c_int
    L.xsi_hip_ctx_create.argtypes = [c.POINTER(vp), c.c_int, vp]
    L.xsi_hip_ctx_destroy.restype = None
    L.xsi_hip_ctx_destroy.argtypes = [vp]
    L.xsi_hip_ctx_synchronize.restype = c.c_int
    L.xsi_hip_ctx_synchronize.argtypes = [vp]
    L.xsi_hip_ctx_workspace_bytes.restype = u64
    L.xsi_hip_ctx_workspace_bytes.argtypes = [vp]
    L.xsi_hip_shard_blocks.restype = None
    L.xsi_hip_shard_blocks.argtypes = [u64, c.c_int, c.c_int, c.POINTER(u64), c.POINTER(u64)]
    L.xsi_hip_shard_of_block.restype = c.c_int
    L.xsi_hip_shard_of_block.argtypes = [u64, c.c_int, u64]
    L.xsi_hip_comm_unique_id.restype = c.c_int
    L.xsi_hip_comm_unique_id.argtypes = [vp]
    L.xsi_hip_comm_create.restype = c.c_int
    L.xsi_hip_comm_create.argtypes = [c.POINTER(vp), vp, c.c_int, c.c_int, vp]
    L.xsi_hip_comm_destroy.restype = None
    L.xsi_hip_comm_destroy.argtypes = [vp]
    L.xsi_hip_comm_world.restype = c.c_int
    L.xsi_hip_comm_world.argtypes = [vp]
    L.xsi_hip_comm_rank.restype = c.c_int
    L.xsi_hip_comm_rank.argtypes = [vp]
    L.xsi_debug_pack_bit_row.restype = c.c_int
    L.xsi_debug_pack_bit_row.argtypes = [vp, u32, i32, vp]
    L.xsi_hip_comm_wait.restype = c.c_int
    L.xsi_hip_comm_wait.argtypes = [vp, c.c_int]
    L.xsi_hip_gather_block_streams.restype = c.c_int
    L.xsi_hip_gather_block_streams.argtypes = [vp, vp, u64, vp, u64, c.c_int, vp, u64, vp, u64, vp, vp]
    L.xsi_hip_gather_block_streams_round.restype = c.c_int
    L.xsi_hip_gather_block_streams_round.argtypes = [vp, vp, u64, vp, u64, c.c_int, vp, u64, vp, u64, u64, u64, vp, vp]
    L.xsi_hip_ctx_reencode_ranges.restype = u32
    L.xsi_hip_ctx_reencode_ranges.argtypes = [vp]
    L.xsi_hip_ctx_chain_fallbacks.restype = u64
    L.xsi_hip_ctx_chain_fallbacks.argtypes = [vp]
    L.xsi_hip_ctx_set_block_sizes_out.restype = ctypes.c_int
    L.xsi_hip_ctx_set_block_sizes_out.argtypes = [vp, vp, u64]
    L.xsi_hip_ctx_set_timing.restype = c.c_int
    L.xsi_hip_ctx_set_timing.argtypes = [vp, c.c_int]
    L.xsi_hip_ctx_get_timing.restype = c.c_int
    L.xsi_hip_ctx_get_timing.argtypes = [vp, c.POINTER(c.c_double), c.POINTER(u64), c.c_int]
    L.xsi_hip_stage_name.restype = c.c_char_p
    L.xsi_hip_stage_name.argtypes = [c.c_int]
    L.xsi_hip_ctx_set_workspace_budget.restype = c.c_int
    L.xsi_hip_ctx_set_workspace_budget.argtypes = [vp, u64]
    L.xsi_hip_chain_kernel.restype = c.c_char_p
    L.xsi_hip_chain_kernel.argtypes = [u32, u64, c.c_int]
    L.xsi_hip_reencode.restype = c.c_int
    L.xsi_hip_reencode.argtypes = [vp, vp, u64, vp, u64, c.POINTER(EncodeParams), vp, u32, vp, u64, vp,
                                   c.POINTER(EncodeResult)]
    L.xsi_mac_threshold.restype = u32
    L.xsi_mac_threshold.argtypes = [u32, u32, c.c_double]
    L.xsi_default_phased.restype = i32
    L.xsi_default_phased.argtypes = [c.POINTER(vp), c.POINTER(u32), u32, u32]
    L.xsi_bm_init.restype = None
    L.xsi_bm_init.argtypes = [c.POINTER(BmState)]
    L.xsi_bm_next.restype = c.c_int64
    L.xsi_bm_next.argtypes = [c.POINTER(BmState), u32, u32]
    L.xsi_accessor_set_sample_subset.restype = c.c_int
    L.xsi_accessor_set_sample_subset.argtypes = [vp, vp, u32]
    L.xsi_accessor_fill_selected_genotypes.restype = c.c_int64
    L.xsi_accessor_fill_selected_genotypes.argtypes = [vp, vp, u64, u32, u64, vp]
    L.xsi_hip_encode_bound.restype = u64
    L.xsi_hip_encode_bound.argtypes = [c.POINTER(EncodeParams), u64, u64]
    L.xsi_hip_encode_gt_bound.restype = u64
    L.xsi_hip_encode_gt_bound.argtypes = [c.POINTER(EncodeParams), u64, u64]
    L.xsi_hip_encode_packed.restype = c.c_int
    L.xsi_hip_encode_packed.argtypes = [vp, c.POINTER(EncodeParams), vp, u64, u32, vp, u64, vp,
                                        c.POINTER(EncodeResult)]
    L.xsi_hip_encode_packed_counted.restype = c.c_int
    L.xsi_hip_encode_packed_counted.argtypes = [vp, c.POINTER(EncodeParams), vp, u64, u32, vp, vp, u64, vp,
                                                c.POINTER(EncodeResult)]
    L.xsi_hip_count_packed_rows.restype = c.c_int
    L.xsi_hip_count_packed_rows.argtypes = [vp, vp, u64, u32, u32, vp]
    L.xsi_hip_encode_gt.restype = c.c_int
    L.xsi_hip_encode_gt.argtypes = [vp, c.POINTER(EncodeParams), vp, u64, u64, vp, vp, vp, u64, vp,
                                    c.POINTER(EncodeResult)]
    L.xsi_hip_make_header.restype = c.c_int
    L.xsi_hip_make_header.argtypes = [c.POINTER(HeaderFields), vp]
    L.xsi_hip_decode_packed.restype = c.c_int
    L.xsi_hip_decode_packed.argtypes = [vp, vp, u64, u64, u64, vp, u32, u64, c.POINTER(u64), vp]
    L.xsi_hip_decode_gt.restype = c.c_int
    L.xsi_hip_decode_gt.argtypes = [vp, vp, u64, u64, u64, vp, u64, vp, u64, vp, vp, u32]
    L.xsi_hip_decode_counts.restype = c.c_int
    L.xsi_hip_decode_counts.argtypes = [vp, vp, u64, u64, u64, vp, vp, u64, c.POINTER(u64)]
    L.xsi_accessor_fill_allele_counts.restype = c.c_int
    L.xsi_accessor_fill_allele_counts.argtypes = [vp, u32, u64]
    L.xsi_hip_synth_packed.restype = c.c_int
    L.xsi_hip_synth_packed.argtypes = [vp, u64, u64, u64, u32, vp, u32]
    L.xsi_hip_debug_chain_encode.restype = c.c_int
    L.xsi_hip_debug_chain_encode.argtypes = [vp, c.POINTER(EncodeParams), vp, u64, u32, vp, u32, vp,
                                             c.POINTER(u64)]
    L.xsi_writer_open.restype = c.c_int
    L.xsi_writer_open.argtypes = [c.POINTER(vp), vp, c.c_char_p, c.POINTER(EncodeParams), c.POINTER(c.c_char_p)]
    L.xsi_writer_append.restype = c.c_int
    L.xsi_writer_append.argtypes = [vp, vp, u32, u32]
    L.xsi_writer_row_buffer.restype = c.c_void_p
    L.xsi_writer_row_buffer.argtypes = [vp]
    L.xsi_writer_commit_row.restype = c.c_int
    L.xsi_writer_commit_row.argtypes = [vp, u32, u32]
    L.xsi_writer_finalize.restype = c.c_int
    L.xsi_writer_finalize.argtypes = [vp, u32]
    L.xsi_writer_close.restype = None
    L.xsi_writer_close.argtypes = [vp]
    L.xsi_accessor_open.restype = c.c_int
    L.xsi_accessor_open.argtypes = [c.POINTER(vp), vp, c.c_char_p]
    L.xsi_accessor_fill_genotype_array.restype = c.c_int64
    L.xsi_accessor_fill_genotype_array.argtypes = [vp, vp, u64, u32, u64]
    L.xsi_accessor_register_array.restype = c.c_int
    L.xsi_accessor_register_array.argtypes = [vp, vp, u64]
    L.xsi_accessor_unregister_array.restype = c.c_int
    L.xsi_accessor_unregister_array.argtypes = [vp]
    L.xsi_accessor_alloc_array.restype = c.c_int
    L.xsi_accessor_alloc_array.argtypes = [vp, u64, c.POINTER(vp)]
    L.xsi_accessor_free_array.restype = c.c_int
    L.xsi_accessor_free_array.argtypes = [vp, vp]
    L.xsi_accessor_get_genotypes.restype = c.c_int64
    L.xsi_accessor_get_genotypes.argtypes = [vp, u32, u64, c.POINTER(vp), c.POINTER(c.c_int)]
    L.xsi_accessor_get_genotypes_batch.restype = c.c_int64
    L.xsi_accessor_get_genotypes_batch.argtypes = [vp, u64, vp, vp, vp, u64, vp]
    L.xsi_accessor_genotypes_view.restype = c.c_int64
    L.xsi_accessor_genotypes_view.argtypes = [vp, u32, u64, vp]
    L.xsi_accessor_get_internal_access.restype = c.c_int
    L.xsi_accessor_get_internal_access.argtypes = [vp, u32, u64, c.POINTER(InternalAccess), vp, vp, vp]
    L.xsi_accessor_allele_counts.restype = c.c_int
    L.xsi_accessor_allele_counts.argtypes = [vp, vp, u32]
    L.xsi_accessor_hap_samples.restype = u64
    L.xsi_accessor_hap_samples.argtypes = [vp]
    L.xsi_accessor_num_samples.restype = u64
    L.xsi_accessor_num_samples.argtypes = [vp]
    L.xsi_accessor_sample_name.restype = c.c_char_p
    L.xsi_accessor_sample_name.argtypes = [vp, u64]
    L.xsi_accessor_set_cache_bytes.restype = c.c_int
    L.xsi_accessor_set_cache_bytes.argtypes = [vp, u64]
    L.xsi_accessor_prefix_stats.restype = c.c_int
    L.xsi_accessor_prefix_stats.argtypes = [vp, c.POINTER(u64), c.POINTER(u64)]
    L.xsi_accessor_readahead_stats.restype = c.c_int
    L.xsi_accessor_readahead_stats.argtypes = [vp, c.POINTER(u64), c.POINTER(u64)]
    L.xsi_accessor_cache_stats.restype = c.c_int
    L.xsi_accessor_cache_stats.argtypes = [vp, c.POINTER(u64), c.POINTER(u64), c.POINTER(u64), c.POINTER(u64)]
    L.xsi_hip_decode_dot.restype = c.c_int
    L.xsi_hip_decode_dot.argtypes = [vp, vp, u64, u64, u64, vp, u32, vp, u64, c.POINTER(u64)]
    L.xsi_hip_decode_dot_gt.restype = c.c_int
    L.xsi_hip_decode_dot_gt.argtypes = [vp, vp, u64, u64, u64, vp, u64, vp, u32, vp, u64, c.POINTER(u64)]
    L.xsi_file_num_samples.restype = c.c_int64
    L.xsi_file_num_samples.argtypes = [c.c_char_p]
    L.xsi_accessor_close.restype = None
    L.xsi_accessor_close.argtypes = [vp]
    _LIB = L
    return L


def check(rc):
    if rc < 0:
        raise XsiError(rc, lib().xsi_hip_last_error().decode(errors="replace"))
    return rc


class Context:
    """One per process / GPU.  `stream` is a raw hipStream_t (e.g. torch.cuda.current_stream().cuda_stream)."""

    def __init__(self, device=0, stream=None):
        h = ctypes.c_void_p()
        check(lib().xsi_hip_ctx_create(ctypes.byref(h), device, ctypes.c_void_p(stream) if stream else None))
        self.handle = h

    def synchronize(self):
        check(lib().xsi_hip_ctx_synchronize(self.handle))

    def set_timing(self, on=True):
        check(lib().xsi_hip_ctx_set_timing(self.handle, 1 if on else 0))

    def timing(self):
        """{stage name: (total ms, launches)} since set_timing(True)."""
        n = 32
        ms = (ctypes.c_double * n)()
        cnt = (ctypes.c_uint64 * n)()
        k = check(lib().xsi_hip_ctx_get_timing(self.handle, ms, cnt, n))
        return {lib().xsi_hip_stage_name(i).decode(): (ms[i], int(cnt[i])) for i in range(k)}

    def workspace_bytes(self):
        return int(lib().xsi_hip_ctx_workspace_bytes(self.handle))

    def close(self):
        if self.handle:
            lib().xsi_hip_ctx_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
