/*
 * xsi_hip.h — C ABI of the MI355X-native xSqueezeIt genotype-block codec (libxsi_hip.so).
 *
 * This is the drop-in boundary for the reference's per-block hot path.  Plain C, caller-owned
 * buffers, int return codes (0 = ok, <0 = error; text via xsi_hip_last_error()), no exceptions
 * and no torch / C++ types in any signature.  Pointers named d_* are DEVICE pointers (HBM),
 * pointers named h_* are HOST pointers.  All device work is enqueued on the context's HIP
 * stream; calls that return host-visible results synchronise that stream themselves.
 *
 * What each group replaces in the reference (paths relative to the reference tree):
 *   - block encode   : GtBlock::encode_line + write_to_stream (include/gt_block.hpp:279-406,
 *                      185-204, 512-647), wah_encode2_with_size (include/wah.hpp:441-578),
 *                      pbwt_sort (include/internal_gt_record.hpp:32-59), Sparse/SparseGtLine
 *                      (include/block.hpp:54-99), IBinaryBlock::write_to_file
 *                      (include/interfaces.hpp:176-268) and the per-block part of
 *                      XsiFactoryExt::append/finalize_file (include/xsi_factory.hpp:513-606).
 *   - block decode   : DecompressPointerGTBlock (include/accessor_internals_new.hpp:49-717) and
 *                      AccessorInternalsNewTemplate::fill_genotype_array / fill_allele_counts
 *                      (include/accessor_internals_new.hpp:719-906).
 *   - writer/accessor: XsiFactoryInterface::append/finalize_file (include/xsi_factory.hpp:38-46)
 *                      and Accessor::fill_genotype_array / get_genotypes (include/accessor.hpp:48-67),
 *                      i.e. what c_xcf_get_genotypes (include/c_api.h:82-83) lands on.
 * The reference-side bindings are shown in INTEGRATION.md.
 */
#ifndef XSI_HIP_H
#define XSI_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define XSI_HIP_ABI_VERSION 1

/* error codes */
#define XSI_OK 0
#define XSI_ERR_ARG (-1)       /* bad argument */
#define XSI_ERR_HIP (-2)       /* HIP runtime error (no device, launch failure, ...) */
#define XSI_ERR_CAPACITY (-3)  /* caller-provided output buffer too small */
#define XSI_ERR_FORMAT (-4)    /* bad magic / version / corrupt block */
#define XSI_ERR_UNSUPPORTED (-5)
#define XSI_ERR_IO (-6)

typedef struct xsi_hip_ctx xsi_hip_ctx;

int xsi_hip_abi_version(void);
/* Thread-local text of the last error returned on this thread. */
const char* xsi_hip_last_error(void);

/* Create a context on `device` using `stream` (a hipStream_t passed as void*, NULL = the
 * context creates its own).  Fails with XSI_ERR_HIP when no GPU is present: there is no
 * CPU fallback anywhere in this library. */
int xsi_hip_ctx_create(xsi_hip_ctx** ctx, int device, void* stream);
void xsi_hip_ctx_destroy(xsi_hip_ctx* ctx);
int xsi_hip_ctx_synchronize(xsi_hip_ctx* ctx);
/* Bytes of device workspace currently held by the context. */
uint64_t xsi_hip_ctx_workspace_bytes(const xsi_hip_ctx* ctx);

/* Encode batches this context has run with the one-workgroup-per-block streaming chain although the chain that
 * spreads a block over several workgroups (65 536 < haplotypes <= 524 288) was eligible: its launch was aborted
 * (its workgroups must all be resident at once, which another process, stream or CU mask can prevent) and the
 * batch was run a second time, or it could not be launched at all (a device too small for one group).
 * The bytes written are the same either way (pbwt_sort, include/internal_gt_record.hpp:32-59). */
uint64_t xsi_hip_ctx_chain_fallbacks(const xsi_hip_ctx* ctx);
/* Optional side output of the encode entry points (xsi_hip_encode_packed[_counted], xsi_hip_encode_gt, xsi_hip_reencode):
 * d_sizes[b] (in HBM, stream-ordered like the call's other outputs) = bytes of block b of the call BEFORE its pad to 4 -
 * the block as the reference streams it into its zstd layer (compress_and_write, interfaces.hpp:291-314), which
 * xsi_encode_result.last_block_bytes gives for the last block only.  With it a caller (xsi_writer_* does) can wrap every
 * block of a multi-block call in its own zstd frame.  capacity_blocks: blocks d_sizes can hold - a call with more fails with
 * XSI_ERR_CAPACITY before it writes anything; d_sizes == NULL switches the output off.  The setting stays until changed. */
int xsi_hip_ctx_set_block_sizes_out(xsi_hip_ctx* ctx, uint32_t* d_sizes, uint64_t capacity_blocks);
uint32_t xsi_hip_ctx_reencode_ranges(const xsi_hip_ctx* ctx);

/* Bytes of per-line device workspace one block-level call may hold.  A job that needs more (e.g. 153 blocks
 * of 500 000 haplotypes) is run as several batches of whole blocks inside the call; the bytes written are
 * those of a single pass, since blocks are independent.  0 (default) = half of the HBM that is free at the call plus what the context already holds and will reuse (never
 * more than the one buffer that takes the per-line rows can get); the environment variable XSI_WS_BUDGET_MB overrides the default -
 * like every XSI_* variable named in this header it is read ONLY when the process opted in with XSI_ENABLE_TUNING_ENV=1
 * (csrc/xsi_common.hpp); without that gate a stray XSI_* variable in a caller's environment changes nothing. */
int xsi_hip_ctx_set_workspace_budget(xsi_hip_ctx* ctx, uint64_t bytes);

/* Per-stage device timing with HIP events recorded on the context's stream (used by bench.py for
 * the roofline of the dominant kernel).  get_timing fills h_ms[i] / h_launches[i] for stage i and
 * returns the number of stages; xsi_hip_stage_name(i) names them. */
int xsi_hip_ctx_set_timing(xsi_hip_ctx* ctx, int on);
int xsi_hip_ctx_get_timing(xsi_hip_ctx* ctx, double* h_ms, uint64_t* h_launches, int n);
const char* xsi_hip_stage_name(int i);
/* Name of the kernel that runs the PBWT chain (the dominant kernel) for a batch of n_blocks blocks of
 * n_haps haplotypes without fully haploid lines: what bench.py names in its roofline object. */
const char* xsi_hip_chain_kernel(uint32_t n_haps, uint64_t n_blocks, int decode);

/* Parameters of one encode job (the arguments of XsiFactoryExt's constructor that reach
 * GtBlock: include/xsi_factory.hpp:439-449, include/gt_block.hpp:159-181). */
typedef struct xsi_encode_params {
    uint32_t n_samples;          /* samples; haplotype columns = 2*n_samples (diploid lines) */
    uint32_t block_len;          /* BCF lines per block (--variant-block-length, 8192) */
    uint32_t mac_threshold;      /* (size_t)(n_samples*ploidy*MAF), gt_compressor_new.hpp:98-99 */
    int32_t default_phased;      /* 0/1, xcf.cpp:811-836 */
    uint32_t wah_encode_missing; /* 0 = WS_SPARSE (default), 1 = WS_WAH (--wah-encode-missing) */
    uint32_t zstd_level;         /* file writer only: 0 = no zstd layer, else --zstd with this --zl level
                                    (reference default 7, include/xsqueezeit.hpp); block calls ignore it */
} xsi_encode_params;

/* Result of an encode call (host struct, filled after the stream is synchronised). */
typedef struct xsi_encode_result {
    uint64_t n_blocks;      /* blocks written */
    uint64_t blocks_bytes;  /* bytes of the blocks region (each block padded to 4, region not yet padded to 8) */
    uint64_t n_binary_lines;
    uint64_t n_wah_lines;
    uint32_t max_ploidy;    /* max line ploidy seen (for finalize_file(max_ploidy)) */
    uint32_t last_block_bytes; /* bytes of the last block before its pad to 4 (what the zstd layer compresses) */
} xsi_encode_result;

/* Upper bound of the blocks region for a job, for sizing d_out. */
uint64_t xsi_hip_encode_bound(const xsi_encode_params* p, uint64_t n_bcf_lines, uint64_t n_binary_lines);

/*
 * Encode bi-allelic, diploid, fully called genotypes from a bit-packed site-major matrix
 * resident in HBM: row l = BCF line l, bit h (LSB-first within little-endian 32-bit words) =
 * 1 iff haplotype h carries ALT.  row_stride_bytes must be a multiple of 8; bits >= 2*n_samples
 * in a row must be zero.  Phase: every second allele carries p->default_phased.
 *
 * Writes the blocks region of the .xsi file (what lies between byte 256 and indices_offset,
 * before the pad to 8) to d_out[0..blocks_bytes) and the file offset of every block
 * (256 + offset in d_out) to d_block_offsets[0..n_blocks).  Byte-identical to what the
 * reference writes for the same lines.
 */
int xsi_hip_encode_packed(xsi_hip_ctx* ctx, const xsi_encode_params* p, const void* d_bits,
                          uint64_t n_lines, uint32_t row_stride_bytes, void* d_out, uint64_t out_capacity,
                          uint64_t* d_block_offsets, xsi_encode_result* h_result);

/*
 * The same with the ALT count of every row supplied by whoever produced the rows (d_row_counts[l] = number of set
 * bits of row l, in HBM): the pass over the matrix that only counts (GtBlock::scan_genotypes' allele histogram,
 * gt_block.hpp:207-269; 16.4 GB of reads at 64 976 haplotypes x 2 M sites) is then not made again.  The counts decide
 * WAH against sparse and the sparse side (gt_block.hpp:298-327): wrong counts give a wrong file, not an error -
 * XSI_CHECK_ROW_COUNTS=1 in the environment (effective only under XSI_ENABLE_TUNING_ENV=1) recounts and returns XSI_ERR_ARG on a difference.  d_row_counts == NULL is
 * xsi_hip_encode_packed.  xsi_hip_count_packed_rows is that counting pass by itself (stream-ordered on the context),
 * for a producer that has no cheaper way; xsi_writer_* counts while it packs (one popcount per mask).
 */
int xsi_hip_encode_packed_counted(xsi_hip_ctx* ctx, const xsi_encode_params* p, const void* d_bits,
                                  uint64_t n_lines, uint32_t row_stride_bytes, const uint32_t* d_row_counts,
                                  void* d_out, uint64_t out_capacity, uint64_t* d_block_offsets,
                                  xsi_encode_result* h_result);
int xsi_hip_count_packed_rows(xsi_hip_ctx* ctx, const void* d_bits, uint64_t n_lines, uint32_t row_stride_bytes,
                              uint32_t n_haps, uint32_t* d_row_counts);

/*
 * General encode from htslib-encoded int32 genotype rows resident in HBM (what
 * bcf_get_genotypes hands GtBlock::encode_line): row l at d_gt + l*gt_stride (int32 units),
 * h_ngt[l] values used (n_samples or 2*n_samples), h_n_allele[l] alleles (>= 2).
 * Handles multi-allelic lines, missing, end-of-vector, non-default phase and haploid lines.
 */
int xsi_hip_encode_gt(xsi_hip_ctx* ctx, const xsi_encode_params* p, const int32_t* d_gt, uint64_t gt_stride,
                      uint64_t n_lines, const uint32_t* h_ngt, const uint32_t* h_n_allele, void* d_out,
                      uint64_t out_capacity, uint64_t* d_block_offsets, xsi_encode_result* h_result);

/*
 * Decode a whole file image and encode it again with other parameters (block length, MAC threshold, default
 * phase, missing-data strategy) and, optionally, a selection of its samples, without the genotypes leaving the
 * device: the "-Ox" path of NewDecompressor (include/gt_decompressor_new.hpp:241-273, fill_selected_genotypes
 * :209-238 + XsiFactoryInterface::append).  h_n_allele[l] for every BCF line as for xsi_hip_decode_gt;
 * h_sample_idx (n_sel indices into the file's samples, NULL = all); p_new->n_samples must be the number of
 * samples written.  Output as xsi_hip_encode_gt.  The file is walked in ranges of whole source blocks sized by the
 * context's workspace budget (xsi_hip_ctx_set_workspace_budget; about 6 bytes per value staged), so its int32 rows
 * never have to fit HBM at once; source blocks and new blocks need not line up (the leftover of a range is carried
 * into the next).  xsi_hip_ctx_reencode_ranges says how many ranges the last call took.
 */
int xsi_hip_reencode(xsi_hip_ctx* ctx, const void* d_file, uint64_t file_len, const uint32_t* h_n_allele, uint64_t n_lines,
                     const xsi_encode_params* p_new, const uint32_t* h_sample_idx, uint32_t n_sel, void* d_out,
                     uint64_t out_capacity, uint64_t* d_block_offsets, xsi_encode_result* h_result);

/* Upper bound of the blocks region for xsi_hip_encode_gt (adds the side channels). */
uint64_t xsi_hip_encode_gt_bound(const xsi_encode_params* p, uint64_t n_bcf_lines, uint64_t n_binary_lines);

/* Fill a complete 256-byte .xsi v5 header (include/compression.hpp:40-104 as written by
 * xsi_factory.hpp:468-500, 543-605).  Host-only helper, no device work. */
typedef struct xsi_header_fields {
    uint32_t n_samples;
    uint32_t max_ploidy;
    uint32_t block_len;
    uint32_t mac_threshold;
    int32_t default_phased;
    uint32_t zstd;
    uint64_t num_variants;   /* sum over lines of (n_allele-1) */
    uint64_t xcf_entries;    /* BCF lines */
    uint64_t indices_offset;
    uint64_t samples_offset;
} xsi_header_fields;
int xsi_hip_make_header(const xsi_header_fields* f, uint8_t h_header[256]);

/*
 * Decode.  d_file is a complete .xsi image in HBM (header, blocks, index).  Blocks
 * [first_block, first_block+n_blocks) are decoded.
 *
 * xsi_hip_decode_packed: every BCF line of those blocks must be bi-allelic and diploid without
 * side channels; output row r (r counts binary lines from the first decoded block) is the
 * natural-order haplotype bit row, same layout as xsi_hip_encode_packed's input.
 * h_counts (optional, may be NULL): ALT allele count per row.
 */
int xsi_hip_decode_packed(xsi_hip_ctx* ctx, const void* d_file, uint64_t file_len, uint64_t first_block,
                          uint64_t n_blocks, void* d_bits_out, uint32_t row_stride_bytes, uint64_t out_rows_capacity,
                          uint64_t* h_rows_written, uint32_t* d_counts);

/*
 * xsi_hip_decode_gt: general decode to htslib int32 rows, what Accessor::fill_genotype_array
 * writes (include/accessor_internals_new.hpp:198-384).  h_n_allele[l] for every BCF line l of
 * the decoded blocks (the reference takes it from the variant BCF record, accessor.hpp:66).
 * Row l at d_gt_out + l*gt_stride; h_line_ngt[l] (optional) receives the value count of the
 * line (N_HAPS or N_SAMPLES); d_allele_counts (optional) receives max_alleles counts per line.
 */
int xsi_hip_decode_gt(xsi_hip_ctx* ctx, const void* d_file, uint64_t file_len, uint64_t first_block,
                      uint64_t n_blocks, const uint32_t* h_n_allele, uint64_t n_lines, int32_t* d_gt_out,
                      uint64_t gt_stride, uint32_t* h_line_ngt, uint64_t* d_allele_counts, uint32_t max_alleles);

/*
 * Allele counts without expanding genotypes — the GPU form of AccessorInternals::fill_allele_counts
 * (include/accessor_internals_new.hpp:407-438): d_ones[r] = ALT count of binary line r of the decoded
 * blocks (WAH lines by popcount of their words, sparse lines from their count field), d_kind[r]
 * (optional) = bit0 WAH line, bit1 negated sparse, bit2 fully haploid line.  No PBWT chain is run.
 */
int xsi_hip_decode_counts(xsi_hip_ctx* ctx, const void* d_file, uint64_t file_len, uint64_t first_block,
                          uint64_t n_blocks, uint32_t* d_ones, uint8_t* d_kind, uint64_t capacity,
                          uint64_t* h_n_bin);

/*
 * Phenotype dot products on the decoded blocks — the GPU form of the reference's compute-on-compressed
 * consumer (Accessor::get_internal_access, include/accessor.hpp:69-75, + dot_prod/dot_prod.hpp:122-245):
 * d_out[r * n_pheno + k] = sum, over the haplotypes that carry the ALT allele of binary line r, of
 * d_pheno[sample * n_pheno + k] (sample = haplotype / 2; = haplotype on fully haploid lines).  float64,
 * fixed summation order; agrees with the reference's sum (taken in PBWT order) up to rounding.
 * Works on the bit planes alone, so it covers bi-allelic, fully called lines (diploid or fully haploid);
 * blocks with multi-allelic lines or missing / end-of-vector entries return XSI_ERR_UNSUPPORTED (use
 * xsi_hip_decode_dot_gt).
 */
int xsi_hip_decode_dot(xsi_hip_ctx* ctx, const void* d_file, uint64_t file_len, uint64_t first_block,
                       uint64_t n_blocks, const double* d_pheno, uint32_t n_pheno, double* d_out,
                       uint64_t capacity_lines, uint64_t* h_n_bin);

/*
 * The same products for ANY block: multi-allelic lines, missing and end-of-vector entries, mixed ploidy.
 * The genotypes are composed in HBM (as xsi_hip_decode_gt does, never leaving the device) and contracted
 * there: d_out[r * n_pheno + k] = sum of d_pheno[sample * n_pheno + k] over the haplotypes whose allele is
 * the ALT allele of binary line r (binary lines in file order: ALT 1 .. n_allele-1 of each BCF line); missing
 * and end-of-vector values carry no allele.  h_n_allele[n_lines] is the allele number of every BCF line of
 * the requested blocks, as for xsi_hip_decode_gt (it lives in the variant BCF).  Blocks are walked in ranges
 * whose int32 rows fit the workspace budget.  Same summation order as xsi_hip_decode_dot: on lines both
 * entry points cover the results are bit-identical.
 */
int xsi_hip_decode_dot_gt(xsi_hip_ctx* ctx, const void* d_file, uint64_t file_len, uint64_t first_block,
                          uint64_t n_blocks, const uint32_t* h_n_allele, uint64_t n_lines, const double* d_pheno,
                          uint32_t n_pheno, double* d_out, uint64_t capacity_lines, uint64_t* h_n_bin);

/* Deterministic synthetic haplotype matrix: writes n_lines packed rows starting
 * at site index first_line.  Generator defined in DESIGN.md; mirrored in numpy for the tests. */
int xsi_hip_synth_packed(xsi_hip_ctx* ctx, uint64_t seed, uint64_t first_line, uint64_t n_lines, uint32_t n_haps,
                         void* d_bits, uint32_t row_stride_bytes);

/* ---- per-stage entry points used by the parity tests to localise a mismatch ---- */
/* PBWT chain only: permuted bit rows y (one per WAH line, ordered) for the given packed input. */
int xsi_hip_debug_chain_encode(xsi_hip_ctx* ctx, const xsi_encode_params* p, const void* d_bits, uint64_t n_lines,
                               uint32_t row_stride_bytes, void* d_yrows, uint32_t y_stride_bytes,
                               uint32_t* d_line_kind, uint64_t* h_n_wah);

/* What xsi_writer_append does to a row in the caller's thread: a bi-allelic, fully called diploid row whose second
 * values carry default_phased becomes one bit per haplotype (h_bits[0 .. ceil(n / 8)), bit h LSB-first = haplotype h
 * carries ALT) and 1 is returned; 0 = the row needs the int32 form (another allele, missing, end-of-vector, a
 * second value with the other phase).  Host only. */
int xsi_debug_pack_bit_row(const int32_t* h_gt, uint32_t n, int32_t default_phased, uint8_t* h_bits);

/* ---- host-only helpers of the fill loops either side of the block path (no device work) ---- */
/* MINOR_ALLELE_COUNT_THRESHOLD = (size_t)((double)(n_samples * PLOIDY) * MAF), include/gt_compressor_new.hpp:96-99;
 * PLOIDY = ploidy of the first record.  This is xsi_encode_params.mac_threshold. */
uint32_t xsi_mac_threshold(uint32_t n_samples, uint32_t ploidy, double maf);
/* seek_default_phased (xcf.cpp:811-836) on the first records of the input (the reference passes 3): rows in
 * htslib encoding, h_ngt[r] values each.  Returns 0/1 = xsi_encode_params.default_phased, <0 on error. */
int32_t xsi_default_phased(const int32_t* const* h_gt_rows, const uint32_t* h_ngt, uint32_t n_rows, uint32_t n_samples);
/* The BM value of every record of the variant-only BCF (replace_samples_by_pos_in_binary_matrix,
 * xcf.cpp:641-714, BM at :685-703): block << 15 | offset, the block advancing every block_len BCF lines and
 * the offset counting binary lines (n_allele - 1 per record) inside the block.  Call xsi_bm_next once per
 * record, in file order; it returns the record's BM (>= 0 as the reference's int32) or XSI_ERR_FORMAT when
 * the offset no longer fits 15 bits ("Offset cannot be represented on 15 bits !", :692-695).
 * The inverse, Accessor::position_from_bm_entry (accessor.hpp:37-46), is the value itself. */
typedef struct xsi_bm_state { uint64_t line, block, offset; } xsi_bm_state;
void xsi_bm_init(xsi_bm_state* st);
int64_t xsi_bm_next(xsi_bm_state* st, uint32_t block_len, uint32_t n_allele);

/* ---- multi-GPU: block sharding and the gather of the compressed block streams over RCCL ----
 * Every 8192-line block is independent (fresh GtBlock with a = iota, include/gt_block.hpp:179-180,
 * include/xsi_factory.hpp:536-537; the decoder resets `a` per block, include/accessor_internals_new.hpp:144), so
 * rank r of G runs the block calls above on its own contiguous block range and no collective touches the data
 * path.  The one exchange step feeds XsiFactoryExt::finalize_file (include/xsi_factory.hpp:543-605) on the writer
 * rank: every rank's blocks region in rank (= file) order plus the offset of every block.  One process per GPU;
 * the communicator is RCCL's (librccl.so.1 is bound at run time, a single-GPU user does not need it). */
typedef struct xsi_hip_comm xsi_hip_comm;
#define XSI_HIP_COMM_ID_BYTES 128
/* Contiguous block range [*lo, *hi) of `rank`: block b goes to rank floor(b * world / n_blocks), so the gathered
 * streams concatenate in file order. */
void xsi_hip_shard_blocks(uint64_t n_blocks, int world, int rank, uint64_t* lo, uint64_t* hi);
/* The rank that owns `block` under that partition (decode side: every rank opens the file, or just its block range,
 * read-only, and a query for BM position p is served by the owner of block p >> 15; no exchange).  -1 on bad arguments. */
int xsi_hip_shard_of_block(uint64_t n_blocks, int world, uint64_t block);
/* ncclGetUniqueId: made by one rank, handed to the others by whatever the host program has (MPI, a file, a socket). */
int xsi_hip_comm_unique_id(uint8_t id[XSI_HIP_COMM_ID_BYTES]);
/* ncclCommInitRank on the context's device; collective over all `world` ranks. */
int xsi_hip_comm_create(xsi_hip_comm** comm, xsi_hip_ctx* ctx, int world, int rank, const uint8_t id[XSI_HIP_COMM_ID_BYTES]);
void xsi_hip_comm_destroy(xsi_hip_comm* comm);
int xsi_hip_comm_world(const xsi_hip_comm* comm);
int xsi_hip_comm_rank(const xsi_hip_comm* comm);
/*
 * Collective.  Every rank passes its blocks region (d_region[0..nbytes), as xsi_hip_encode_* wrote it) and the
 * offsets of its blocks RELATIVE TO ITS REGION (d_offsets[0..n_blocks); xsi_hip_encode_* returns file offsets of
 * a single-rank file: subtract 256).  On rank `dst`, d_region_all receives the regions back to back in rank
 * order and d_offsets_all the offsets of all blocks relative to the start of the concatenated region (file offset
 * = 256 + that); the other ranks pass NULL / 0 for them.  h_bytes_per_rank / h_blocks_per_rank (optional, `world`
 * entries) are filled on every rank.  Sizes travel by ncclAllGather, the bytes by grouped ncclSend / ncclRecv of
 * exactly each rank's size.  The exchange runs on the communicator's own stream, ordered behind everything the
 * context's stream held at the call (the encode), so work enqueued on the context afterwards (the decode of the
 * same blocks, the next batch's H2D copies) overlaps with it; the call returns once everything is enqueued (the
 * size exchange synchronises once).  d_region / d_offsets must stay untouched, and the outputs unread, until
 * xsi_hip_comm_wait.  XSI_ERR_CAPACITY is returned on EVERY rank, before any byte moves, when the writer rank's
 * buffers are too small.
 */
int xsi_hip_gather_block_streams(xsi_hip_comm* comm, const void* d_region, uint64_t nbytes, const uint64_t* d_offsets,
                                 uint64_t n_blocks, int dst, void* d_region_all, uint64_t region_capacity,
                                 uint64_t* d_offsets_all, uint64_t offsets_capacity, uint64_t* h_bytes_per_rank,
                                 uint64_t* h_blocks_per_rank);
/* The same for a job whose ranks encode their shards in several ROUNDS (a shard that does not fit HBM at once: round k
 * of every rank is gathered while round k + 1 is encoded).  The writer rank appends: this round's regions land at
 * d_region_all + region_base, its offsets at d_offsets_all + blocks_base, rebased by region_base as well, so that
 * after the last round d_offsets_all holds the offsets of all blocks relative to the start of d_region_all.  With
 * rounds the file order is round-major; a job that wants rank-major order gives every rank its own base.
 * region_capacity / offsets_capacity stay the sizes of the whole buffers.  A call that fails (a refused post, a
 * capacity error) leaves no RCCL group open and the communicator usable, and xsi_hip_comm_wait behind it returns
 * without waiting for anything.  xsi_hip_gather_block_streams is round 0 with both bases 0. */
int xsi_hip_gather_block_streams_round(xsi_hip_comm* comm, const void* d_region, uint64_t nbytes, const uint64_t* d_offsets,
                                       uint64_t n_blocks, int dst, void* d_region_all, uint64_t region_capacity,
                                       uint64_t* d_offsets_all, uint64_t offsets_capacity, uint64_t region_base,
                                       uint64_t blocks_base, uint64_t* h_bytes_per_rank, uint64_t* h_blocks_per_rank);

/* Wait for the exchange started last: host != 0 blocks the calling thread, host == 0 makes the context's stream
 * wait (work enqueued on it afterwards sees the gathered bytes). */
int xsi_hip_comm_wait(xsi_hip_comm* comm, int host);

/* ---- host-side writer / accessor (file level), mirroring XsiFactoryInterface and Accessor ---- */
typedef struct xsi_writer xsi_writer;
typedef struct xsi_accessor xsi_accessor;

/* XsiFactoryExt(filename, block_len, mac_thr, default_phased, sample_list, zstd=false)
 * (include/xsi_factory.hpp:439-511).  sample_names: n_samples C strings. */
int xsi_writer_open(xsi_writer** w, xsi_hip_ctx* ctx, const char* path, const xsi_encode_params* p,
                    const char* const* sample_names);
/* XsiFactoryInterface::append(bcf_fri): one BCF line, host int32 row (bcf_fri.gt_arr), ngt values,
 * n_allele = bcf_fri.line->n_allele.  Lines are batched per block and encoded on the GPU.  A bi-allelic, fully
 * called diploid line with default phase is packed to one bit per haplotype right here (4 N bytes read, N / 8
 * written and shipped); any other line is copied as int32.  The file does not depend on which way a line went. */
int xsi_writer_append(xsi_writer* w, const int32_t* h_gt, uint32_t ngt, uint32_t n_allele);
/* The same without the copy: xsi_writer_row_buffer returns the next row's slot in the writer's pinned staging
 * (room for 2 * n_samples int32 values; NULL on error) for the caller to fill - e.g. as the destination array of
 * bcf_get_genotypes, which GtCompressorStream otherwise fills and hands to append - and xsi_writer_commit_row
 * appends it.  One buffer is outstanding at a time.  The per-line memcpy of xsi_writer_append is what bounds the
 * file-level write rate on one host core. */
int32_t* xsi_writer_row_buffer(xsi_writer* w);
int xsi_writer_commit_row(xsi_writer* w, uint32_t ngt, uint32_t n_allele);
/* XsiFactoryInterface::finalize_file(max_ploidy); max_ploidy = 0 -> use the maximum seen. */
int xsi_writer_finalize(xsi_writer* w, uint32_t max_ploidy);
void xsi_writer_close(xsi_writer* w);

/* Accessor(filename).get_number_of_samples() without a device: what c_xcf_nsamples (c_api.cpp:60-65)
 * asks of an .xsi file.  Reads the 256-byte header only.  Returns the count or <0. */
int64_t xsi_file_num_samples(const char* path);
/* Accessor(filename) (accessor.cpp:26-82). */
int xsi_accessor_open(xsi_accessor** a, xsi_hip_ctx* ctx, const char* path);
/* Accessor::fill_genotype_array(gt_arr, gt_arr_size, n_alleles, position) (accessor.hpp:48-50):
 * position = BM value (block<<15 | binary-line offset).  Returns the number of values written
 * (N_HAPS or N_SAMPLES) or <0.
 * Any array works: the line is composed on the device, lands in the accessor's pinned window and is copied out
 * behind the capacity check (XSI_ERR_CAPACITY when gt_size is smaller than the line).  Nothing is page-locked
 * behind the caller's back; the fast path is the opt-in below. */
int64_t xsi_accessor_fill_genotype_array(xsi_accessor* a, int32_t* h_gt, uint64_t gt_size, uint32_t n_alleles,
                                         uint64_t position);
/* Opt-in fast path for a caller that reuses ONE destination array (what an htslib caller's gt_arr is): single-line
 * fills into exactly the registered pointer are stored there by the compose kernel itself (posted PCIe writes: one
 * launch, one completion, no window copy; 33 instead of 73 us per line at 200 000 haplotypes), and rows inside it
 * serve xsi_accessor_get_genotypes_batch the same way.
 * The array must be PAGE-LOCKED memory: take it from xsi_accessor_alloc_array (hipHostMalloc; freed by
 * xsi_accessor_free_array or, at the latest, by xsi_accessor_close), or hand in an allocation you page-locked yourself
 * (hipHostMalloc, a framework's pinned allocator).  Pageable memory is refused with XSI_ERR_ARG: the accessor never
 * page-locks caller memory behind an unregister of its own - on this runtime the hipHostUnregister of a range that is
 * not page-aligned also revokes the device's access to neighbouring pages that the runtime keeps pinned for other host
 * allocations, and an unrelated asynchronous copy faults later (round 4: an intermittent "Memory access fault by GPU").
 * n_values must be at least 2 * num_samples - the width of a composed row whatever a line's ploidy; a smaller array
 * is refused (XSI_ERR_CAPACITY) and keeps working through the ordinary path.
 * LIFETIME: the array must stay allocated until xsi_accessor_unregister_array, the next xsi_accessor_register_array,
 * xsi_accessor_free_array of it, or xsi_accessor_close returns.  One registered array per accessor.
 * XSI_ACCESSOR_NO_REGISTER=1 makes the registration a no-op (measurement); XSI_ACCESSOR_NO_ZEROCOPY=1 keeps it but
 * fills the array with the copy engine (both effective only under XSI_ENABLE_TUNING_ENV=1). */
int xsi_accessor_alloc_array(xsi_accessor* a, uint64_t n_values, int32_t** h_gt);
int xsi_accessor_free_array(xsi_accessor* a, int32_t* h_gt);
int xsi_accessor_register_array(xsi_accessor* a, int32_t* h_gt, uint64_t n_values);
int xsi_accessor_unregister_array(xsi_accessor* a);
/* Accessor::get_genotypes without the htslib record: mallocs *h_gt when NULL (hap_samples ints),
 * sets *ngt_arr = hap_samples (accessor.hpp:58-67). */
int64_t xsi_accessor_get_genotypes(xsi_accessor* a, uint32_t n_alleles, uint64_t position, void** h_gt,
                                   int* ngt_arr);
/* n lines in one call (no reference counterpart: Accessor::get_genotypes is one line per call, and every call here is
 * one kernel launch and one completion, 33 us at 200 000 haplotypes): query i = (n_alleles[i], positions[i]), any
 * order, any blocks; its values land in h_rows + i * row_stride (row_stride >= 2 * num_samples values), their number in
 * h_ngt[i] (optional).  The queries are grouped by block, each block touched costs one compose launch, a chunk of up
 * to 4096 lines one completion.  Rows inside the array registered with xsi_accessor_register_array are stored there by
 * the kernels themselves; other memory is filled through the accessor's device window, one copy per chunk.  Returns the
 * total number of values or <0.  Allele counts are not collected (xsi_accessor_allele_counts keeps the last single
 * fill's). */
int64_t xsi_accessor_get_genotypes_batch(xsi_accessor* a, uint64_t n, const uint32_t* n_alleles, const uint64_t* positions,
                                         int32_t* h_rows, uint64_t row_stride, uint32_t* h_ngt);
/* The line's values without the copy into a caller array: *h_gt points into the accessor's pinned host window
 * (valid until the next call on this accessor); returns the number of values.  Counts as after a fill. */
int64_t xsi_accessor_genotypes_view(xsi_accessor* a, uint32_t n_alleles, uint64_t position, const int32_t** h_gt);
/* Decoded blocks stay resident in HBM (LRU) so that backward and random seeks do not replay a block
 * prefix the way accessor_internals_new.hpp:154-196 does.  Budget in bytes (default: half of the free
 * HBM at open, at most 64 GiB; XSI_ACCESSOR_CACHE_MB overrides it under XSI_ENABLE_TUNING_ENV=1); 0 keeps only the current block. */
int xsi_accessor_set_cache_bytes(xsi_accessor* a, uint64_t bytes);
/* Any of the outputs may be NULL. */
int xsi_accessor_cache_stats(const xsi_accessor* a, uint64_t* blocks, uint64_t* bytes, uint64_t* hits,
                             uint64_t* misses);
/* The first touch of a block decodes only what the query needs: its sparse lines and side channels, and the PBWT chain
 * over the WAH lines IN FRONT of the requested line - what the reference's seek replays on the host
 * (accessor_internals_new.hpp:154-196) - leaving the chain's ranks parked in HBM; a later query further into the block
 * continues from there (in steps that grow by half of what is decoded, so a scan through a cold block is a dozen
 * continuations).  prefix_decodes: first touches that stopped before the block's end; extensions: continuations since
 * open.  XSI_ACCESSOR_FULL_DECODE=1 (under XSI_ENABLE_TUNING_ENV=1) decodes whole blocks on first touch (measurement, as
 * before round 4). */
int xsi_accessor_prefix_stats(const xsi_accessor* a, uint64_t* prefix_decodes, uint64_t* extensions);
/* Sequential scans (the reference's published benchmark loads every line in order, loading_time/gt_loader_new.hpp:112-172):
 * once a handful of single-line queries have followed one another line by line, a prefix-decoded block is finished in one
 * continuation, a cold block is decoded whole, and block b + 1 is decoded by a second thread (own stream and workspace)
 * while block b is served, provided the cache can hold both.  started: read-aheads begun; hits: blocks whose first touch
 * found them in HBM because of one.  Plain files only (a zstd block is inflated on the host first).
 * XSI_ACCESSOR_NO_READAHEAD=1 (under XSI_ENABLE_TUNING_ENV=1) switches the whole policy off (measurement). */
int xsi_accessor_readahead_stats(const xsi_accessor* a, uint64_t* started, uint64_t* hits);
/* Sample selection on decode (NewDecompressor::fill_selected_genotypes, include/gt_decompressor_new.hpp:209-238):
 * after set_sample_subset(idx, n) (indices into the file's sample list, any order, repeats allowed; n = 0
 * clears), fill_selected_genotypes composes the line on the device, gathers the listed samples there and
 * copies only those values to h_gt (1 or 2 per sample by the line's ploidy).  Returns AN = n * line ploidy;
 * h_ac (optional, n_alleles - 1 ints) receives the selection's count of every ALT allele, the AC the
 * reference rewrites (:251-254). */
int xsi_accessor_set_sample_subset(xsi_accessor* a, const uint32_t* sample_idx, uint32_t n);
int64_t xsi_accessor_fill_selected_genotypes(xsi_accessor* a, int32_t* h_gt, uint64_t gt_size, uint32_t n_alleles,
                                             uint64_t position, int32_t* h_ac);
/* Accessor::fill_allele_counts(n_alleles, position) (accessor.hpp:52-54): counts only, no genotype
 * expansion; like the reference, counts[0] = line values - sum of ALT counts (missing / end-of-vector
 * are not subtracted, accessor_internals_new.hpp:437). */
int xsi_accessor_fill_allele_counts(xsi_accessor* a, uint32_t n_alleles, uint64_t position);
/* Accessor::get_allele_counts after a fill (accessor.hpp:56). */
int xsi_accessor_allele_counts(xsi_accessor* a, uint64_t* h_counts, uint32_t n_alleles);
/*
 * Accessor::get_internal_access (include/accessor.hpp:69-75 -> AccessorInternalsNewTemplate::get_internal_access,
 * include/accessor_internals_new.hpp:444-471; class InternalGtAccess, include/accessor_internals.hpp:374-397):
 * the compressed form of a record for callers that compute on it (dot_prod/dot_prod.hpp).  For each of the
 * n_alleles - 1 binary lines of the record at `position` (BM value): h_sparse[i] = 1 sparse list / 0 WAH16
 * words, h_offsets[i] = byte offset of that data inside info->image, a host copy of the file (of the inflated
 * block for zstd files) valid until the next call on this accessor.  A sparse list is {count | MSB, indices} of
 * sparse_bytes each; WAH words are wah_bytes each.  h_a (optional, hap_samples entries, 4 bytes each whatever
 * the file's A_T: a_bytes reports 4) receives the PBWT arrangement in force at the LAST of those lines, as the
 * reference's `a` pointer shows it after its seeks: a[i] = haplotype at position i of the permuted rows.  It is
 * rebuilt on the device from the block's decoded lines (the decode kernels track ranks, never `a`), one stable
 * partition per earlier WAH line of the block: a replay, as in the reference (a fully haploid line partitions the
 * haplotypes by their sample's bit, internal_gt_record.hpp:50-59; its words are the sample bits gathered through the
 * even members of `a`, halved).  default_allele as in the reference: 1 when the first line is a negated sparse line.
 */
typedef struct xsi_internal_access {
    uint64_t position;
    uint32_t n_alleles;
    uint32_t sparse_bytes, wah_bytes, a_bytes;
    int32_t default_allele;
    uint32_t n_a;       /* entries of a = hap_samples */
    const void* image;  /* base of the offsets */
    uint64_t image_len;
} xsi_internal_access;
int xsi_accessor_get_internal_access(xsi_accessor* a, uint32_t n_alleles, uint64_t position, xsi_internal_access* info,
                                     uint8_t* h_sparse, uint64_t* h_offsets, uint32_t* h_a);
uint64_t xsi_accessor_hap_samples(const xsi_accessor* a);
uint64_t xsi_accessor_num_samples(const xsi_accessor* a);
/* Accessor::get_sample_list()[i] */
const char* xsi_accessor_sample_name(const xsi_accessor* a, uint64_t i);
void xsi_accessor_close(xsi_accessor* a);

/* ---- htslib-facing layer (csrc/xsi_htslib_shim.cpp) ----
 * 1 when the library was built with htslib (make HTSLIB=1) and therefore also exports the reference's own C API over
 * this one - c_xcf_new / c_xcf_add_readers / c_xcf_update_readers / c_xcf_sample_name / c_xcf_nsamples /
 * __c__xcf__get__genotypes__void / c_xcf_delete (include/c_api.h:38-93; declared by that header, they take htslib
 * types) - and gives the two fill loops below a body; 0 otherwise: the loops then return XSI_ERR_UNSUPPORTED. */
int xsi_htslib_shim_available(void);
/* xsqueezeit -c -f in_bcf -o out_xsi [--maf] [--variant-block-length] [--zstd --zl]: writes out_xsi and
 * out_xsi + "_var.bcf" (the variant-only BCF whose one pseudo sample carries FORMAT/BM = block << 15 | binary-line
 * offset; replace_samples_by_pos_in_binary_matrix, xcf.cpp:641-714), the genotypes through xsi_writer_append
 * (bcf_traversal.cpp:3-16, gt_compressor_new.hpp:84-142).  zstd_level 0 = no zstd layer. */
int xsi_compress_bcf(const char* in_bcf, const char* out_xsi, double maf, uint32_t block_len, uint32_t zstd_level);
/* xsqueezeit -x -f in_xsi -o out_path with the CLI's selection and output flags (xsqueezeit.hpp:36-93): */
typedef struct xsi_decompress_options {
    const char* regions;  /* -r "chr:from-to[,...]" or, with regions_is_file, -R file; NULL = none (needs the index) */
    int regions_is_file;
    const char* targets;  /* -t; used when no region is given; NULL = none */
    const char* samples;  /* -s "A,B" (that order) or "^A,B" (all but); NULL = all.  AC / AN are recomputed */
    char output_type;     /* -O: 'b' BCF (0 = default), 'u' uncompressed BCF, 'z' vcf.gz, 'v' VCF, 'x' a new .xsi */
    int fast_pipe;        /* -p: out_path "-" is written as uncompressed BCF */
    int no_header;        /* -H, VCF outputs only */
    double maf;           /* 'x' only: --maf of the new file */
    uint32_t zstd_level;  /* 'x' only: 0 keeps the input file's setting */
} xsi_decompress_options;
/* NewDecompressor::decompress (gt_decompressor_new.hpp:57-90, 113-206, 241-320, 432-543): walks in_xsi + "_var.bcf",
 * fetches every record's genotypes by its BM value (xsi_accessor_fill_genotype_array / _fill_selected_genotypes),
 * puts them back into the record and writes it; 'x' re-encodes into out_path (+ "_var.bcf" with the new BM values).
 * opt may be NULL (everything, as BCF). */
int xsi_decompress_bcf(const char* in_xsi, const char* out_path, const xsi_decompress_options* opt);

#ifdef __cplusplus
}
#endif
#endif /* XSI_HIP_H */
