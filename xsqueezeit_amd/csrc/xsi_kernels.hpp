// xsi_kernels.hpp — launchers of the gfx950 kernels (implemented in xsi_kernels.hip).
// All launchers enqueue on `s` and return the hipError_t of the launch.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "xsi_common.hpp"

namespace xsi {

// ---- per-line metadata of an encode batch (device arrays, indexed by binary line) ----
struct EncLines {
    const uint32_t* planes;     // bit row of binary line l at planes + l*plane_stride_w (natural hap order)
    uint32_t plane_stride_w;
    uint32_t n_bin;
    uint32_t N;                 // 2 * n_samples
    uint32_t aet;               // A_T bytes of the block encoder (2 or 4)
    uint32_t thr;               // MAC threshold
    // general path only (nullptr on the packed fast path)
    const uint32_t* bin_nbits;  // GT values of the parent BCF line (N or N/2)
    const uint32_t* bin_parent; // parent BCF line
    const uint32_t* ref_planes; // per BCF line: allele == 0 plane (stride = plane_stride_w)
    const uint32_t* ref_cnt;    // per BCF line: count of allele == 0
    // outputs of the classification
    uint32_t* cnt;              // ones per binary line
    uint8_t* kind;              // KIND_* bits
    uint32_t* line_block;       // block of each binary line
    uint32_t* wah_rank;         // rank among the block's WAH lines
    uint32_t* sparse_off;       // byte offset inside the block's sparse matrix
    uint32_t* wah_lines;        // batch-wide list of WAH binary lines, in order
    uint32_t* wah_len;          // words per WAH line (by batch-wide rank)
    uint32_t* wah_off;          // word offset inside the block's WAH matrix (by batch-wide rank)
    uint64_t* yrows;            // permuted bit rows, one per WAH line (by batch-wide rank)
    uint32_t y_stride64;
    uint32_t y_rows;            // rows to provide for: the batch's WAH lines when known beforehand, else 0 (= n_bin)
    // chain over several workgroups per block (N > 65536), nullptr = not available to this call:
    // chain_sync  [0] abort flag, [16 + g] arrivals of group g, [CHAIN_SYNC_WORDS ..] list counts
    // chain_lists per-wave rank lists, CHAIN_LIST_WORDS words
    // chain_slices the members' finished table slices (owner-computes exchange), CHAIN_SLICE_BYTES
    // no_multi    this call must not use that chain (it is being run again after an aborted launch)
    uint32_t* chain_sync;
    uint32_t* chain_lists;
    void* chain_slices;
    uint32_t* chain_bmps;       // the members' private bitmaps of a row (bitmap exchange), CHAIN_BMP_BYTES
    uint32_t* chain_items;      // the launch's schedule: CHAIN_ITEM_BEGIN_WORDS of per-group begins, then 16-byte items (k_multi_schedule)
    uint32_t* chain_park;       // ranks handed from a block's head part to its tail part, CHAIN_PARK_BYTES
    uint32_t no_multi;
    uint16_t* wah_scratch;      // [rank][wah_scratch_stride] WAH16 words of each line (encoded once, copied to place)
    uint32_t wah_scratch_stride;
    uint32_t wah_inplace;       // the sizing pass leaves a line's words in the line's own y row (k_wah_units / _wide MODE 2)
    uint32_t* flagbits;         // [n_blocks][FV_COUNT][MAX_BIN_PER_BLOCK/32] packed flag vectors
    uint16_t* flagwah;          // [n_blocks][FV_COUNT][FLAG_WORDS_MAX] encoded flag vectors
};

// side channels of the general path (per BCF line; all nullptr on the packed fast path)
struct EncSide {
    const uint32_t* miss_planes;  // [n_bcf][plane_stride_w] missing positions (MissingPred, gt_block.hpp:76-81)
    const uint32_t* eov_planes;   // end-of-vector positions (EndOfVectorPred, gt_block.hpp:82-87)
    const uint32_t* phase_planes; // odd index && phase bit != default (NonDefaultPhasingPred, gt_block.hpp:93-100)
    const uint32_t* bcf_nbits;    // GT values per BCF line (n_samples or 2*n_samples)
    const uint32_t* bcf_flags;    // bit0 missing, bit1 eov, bit2 phase, bit3 haploid
    const uint32_t* bcf_first_bin;// first binary line of the BCF line (batch-wide)
    const uint32_t* miss_cnt;
    const uint32_t* eov_cnt;
    uint32_t* miss_size;          // bytes of the line's entry in the missing matrix (0 if none)
    uint32_t* eov_size;
    uint32_t* phase_len;          // words of the line's phase WAH line (0 if none)
    uint32_t* miss_off;           // byte offset inside the block's missing matrix
    uint32_t* eov_off;
    uint32_t* phase_off;          // word offset inside the block's phase matrix
    uint32_t n_bcf;
    uint32_t plane_stride_w;
    uint32_t aet;
    uint32_t strategy;            // WS_SPARSE or WS_WAH
};

// WAH lines (minor allele count above the threshold) of every block_len lines, from the per-line counts
hipError_t launch_wah_lines_per_block(hipStream_t s, const uint32_t* cnt, uint64_t n_lines, uint32_t block_len, uint32_t nbits,
                                      uint32_t thr, uint32_t* out);
hipError_t launch_compare_u32(hipStream_t s, const uint32_t* a, const uint32_t* b, uint64_t n, uint32_t* n_diff);
hipError_t launch_count_rows(hipStream_t s, const uint32_t* planes, uint32_t stride_w, uint32_t nbits,
                             uint32_t n_rows, uint32_t* cnt);
hipError_t launch_classify(hipStream_t s, EncBlock* blocks, uint32_t n_blocks, const EncLines& L);
hipError_t launch_scan_blocks_wah(hipStream_t s, EncBlock* blocks, uint32_t n_blocks, uint32_t* totals);
hipError_t launch_build_wah_list(hipStream_t s, const EncBlock* blocks, uint32_t n_blocks, const EncLines& L);
hipError_t launch_chain_encode(hipStream_t s, const EncBlock* blocks, uint32_t n_blocks, const EncLines& L,
                               uint32_t* scratch_a /*global-memory variant only*/, bool any_haploid,
                               bool* multi_refused = nullptr /* out: the long-row kernel was eligible but could not be launched */);
hipError_t launch_wah_sizes(hipStream_t s, const EncLines& L, const uint32_t* d_total_wah, uint32_t max_wah);
// rows short enough for the unit encoder (no WAH scratch is needed then)
bool wah_units_ok(uint32_t y_stride64);
// either form of the unit encoder applies (one wave per 4 lines up to 8 KiB, one workgroup per line above)
bool wah_units_any(uint32_t y_stride64);
hipError_t launch_block_layout(hipStream_t s, EncBlock* blocks, uint32_t n_blocks, const EncLines& L,
                               const EncSide& S, int32_t default_phased);
hipError_t launch_scan_blocks_out(hipStream_t s, EncBlock* blocks, uint32_t n_blocks, uint64_t capacity,
                                  uint64_t* d_block_offsets, uint64_t* d_result /*[5]*/, uint64_t file_base, uint32_t* d_block_sizes = nullptr);
hipError_t launch_write_headers(hipStream_t s, const EncBlock* blocks, uint32_t n_blocks, const EncLines& L,
                                int32_t default_phased, uint32_t strategy, uint8_t* out, const uint64_t* d_result);
hipError_t launch_wah_write(hipStream_t s, const EncBlock* blocks, const EncLines& L, uint32_t max_wah,
                            uint8_t* out, const uint64_t* d_result);
hipError_t launch_sparse_write(hipStream_t s, const EncBlock* blocks, const EncLines& L, uint8_t* out,
                               const uint64_t* d_result, uint8_t* scratch /*nullable*/, uint64_t scratch_stride);
hipError_t launch_sparse_copy(hipStream_t s, const EncBlock* blocks, uint32_t n_blocks, uint8_t* out,
                              const uint64_t* d_result, const uint8_t* scratch, uint64_t scratch_stride);

// ---- decode ----
constexpr uint32_t WAH_BND_TILE_WORDS = 8192u;  // words per tile of the WAH line-boundary scan (k_wah_tile_sums / k_wah_boundaries)

struct DecLines {
    uint32_t N;            // 2 * n_samples
    uint32_t n_samples;
    uint32_t aet;          // header.aet_bytes
    uint32_t max_bin;      // capacity of the per-line arrays
    uint8_t* kind;         // per binary line (batch-wide index)
    uint32_t* line_block;
    uint32_t* rank;        // rank among the block's WAH (or sparse) lines
    uint32_t* wah_start;   // [wah rank] word offset of the line inside the block's WAH matrix
    uint32_t* sparse_start;// [sparse rank] byte offset inside the block's sparse matrix
    uint32_t* wah_lines;   // [wah rank] binary line
    uint32_t* sparse_lines;
    uint2* yp;             // [wah rank][yp_stride] permuted rows as {32 bits, ones before these bits} pairs
    uint32_t yp_stride;    // pairs per row (>= ceil(N/32), even)
    // compact form of the same buffer (one-workgroup-per-block decode chain, rows of at most 65 536 bits): rows
    // of y_stride64 64-bit chunks, then - behind all yp_rows of them - 16-bit "ones before the chunk": 10 bytes
    // per 64 positions instead of 16; the chain kernel forms the pairs while it stages a row in LDS
    uint32_t yp_compact;
    uint32_t yp_rev;            // long rows (k_wah_expand_wide_t -> k_chain_decode_rank_big): pairs {row bits bit-reversed, MINUS the
                                // ones up to the END of the word}: the bit and the ones before a position from ONE shift (round 5)
    uint32_t yp_rows;
    uint32_t* wah_z;       // [wah rank] zeros of the line (line bits - ones)
    uint32_t y_stride64;   // ceil(N/64): 64-bit words of a plain bit row
    uint32_t* ones;        // per binary line: allele count (accessor "ones")
    uint32_t* wah_cumg;    // [wah rank] cumulative 15-bit groups before the line (mixed-ploidy blocks)
    uint32_t* tile_sum;    // [block][max_tiles] 15-bit groups per tile of WAH_BND_TILE_WORDS words of the WAH matrix
    uint64_t* tile_base;   // [block][max_tiles] groups before the tile
    uint32_t max_tiles;
    uint64_t file_len;     // bytes of the file image (bounds every read)
    // ranged sparse lists (the accessor's prefix decode, one block): the sparse lines of rank [sp_lo, sp_hi) only; the
    // walk's cursor behind rank sp_hi - 1 is left in sp_state[0] and picked up from there when sp_lo > 0.
    // sp_state == nullptr: every sparse line (everything else).
    uint32_t sp_lo, sp_hi;
    uint64_t* sp_state;
};

hipError_t launch_parse_blocks(hipStream_t s, const uint8_t* file, uint64_t file_len, uint64_t indices_offset,
                               uint32_t version, uint64_t first_block, uint32_t n_blocks, DecBlock* blocks,
                               uint32_t* d_totals /*[4]: n_bin, n_wah, n_sparse, error*/);
hipError_t launch_decode_flags(hipStream_t s, const uint8_t* file, DecBlock* blocks, uint32_t n_blocks,
                               const DecLines& L);
hipError_t launch_scan_dec_blocks(hipStream_t s, DecBlock* blocks, uint32_t n_blocks, uint32_t* d_totals);
hipError_t launch_scan_dec_blocks2(hipStream_t s, DecBlock* blocks, uint32_t n_blocks, uint32_t* d_totals);
hipError_t launch_dec_line_lists(hipStream_t s, const DecBlock* blocks, uint32_t n_blocks, const DecLines& L);
hipError_t launch_wah_boundaries(hipStream_t s, const uint8_t* file, const DecBlock* blocks, uint32_t n_blocks,
                                 const DecLines& L);
// The same in two parts for the phased decode: part 1 = tile sums, their scan and the starts of every block's first
// n_wah x num / ranges lines (the tiles in front of that many lines' groups), part 2 = the other tiles (any stream,
// behind part 1).  Blocks without fully haploid lines only.
hipError_t launch_wah_boundaries_part(hipStream_t s, const uint8_t* file, const DecBlock* blocks, uint32_t n_blocks,
                                      const DecLines& L, uint32_t ranges, uint32_t num, int part);
hipError_t launch_sparse_walk(hipStream_t s, const uint8_t* file, const DecBlock* blocks, uint32_t n_blocks,
                              const DecLines& L);
hipError_t launch_wah_expand(hipStream_t s, const uint8_t* file, const DecBlock* blocks, const DecLines& L,
                             uint32_t max_wah, const uint32_t* d_totals);
// one range of every block's WAH lines: ph_start[b] first line (batch-wide rank), ph_cnt[b] lines,
// ph_gpre[b] groups of wah_expand_lines_per_group(L) lines before block b (ph_gpre[n_blocks] = n_groups)
uint32_t wah_expand_lines_per_group(const DecLines& L);
hipError_t launch_wah_expand_phase(hipStream_t s, const uint8_t* file, const DecBlock* blocks, const DecLines& L,
                                   const uint32_t* d_totals, const uint32_t* ph_start, const uint32_t* ph_cnt,
                                   const uint32_t* ph_gpre, uint32_t n_blocks, uint32_t n_groups);
// the one-workgroup-per-block decode chain over one range of lines; ranks parked in `state` between the launches
bool rank_decode_phased_ok(uint32_t N, uint32_t yp_stride, uint32_t n_blocks);
bool rank_decode_takes_compact(uint32_t N, uint32_t yp_stride, uint32_t n_blocks);
bool rank_decode_takes_reversed(uint32_t N, uint32_t yp_stride, uint32_t n_blocks);  // (the chain side; the expansion side: wah_expand_wide)
bool wah_expand_wide(const DecLines& L);
uint64_t rank_decode_state_words(uint32_t N, uint32_t n_blocks);
hipError_t launch_rank_decode_phase(hipStream_t s, const DecBlock* blocks, uint32_t n_blocks, const DecLines& L,
                                    uint32_t* out_rows, uint32_t out_stride_w, const uint32_t* ph_start,
                                    const uint32_t* ph_cnt, uint32_t* state, uint32_t active_blocks = 0);
uint32_t rank_decode_big_wgs_per_block(uint32_t N, uint32_t yp_stride, uint32_t active = 0);
uint32_t rank_decode_cus();  // CUs of the current device
// element-major decode chain (xsi_rank.hip): all blocks without fully haploid lines
hipError_t launch_rank_decode(hipStream_t s, const DecBlock* blocks, uint32_t n_blocks, const DecLines& L,
                              uint32_t* out_rows, uint32_t out_stride_w);
hipError_t launch_chain_decode(hipStream_t s, const DecBlock* blocks, uint32_t n_blocks, const DecLines& L,
                               uint32_t* out_rows, uint32_t out_stride_w, uint32_t* scratch_a, bool any_haploid);
hipError_t launch_sparse_fill(hipStream_t s, const uint8_t* file, const DecBlock* blocks, const DecLines& L,
                              uint32_t max_sparse, const uint32_t* d_totals, uint32_t* out_rows,
                              uint32_t out_stride_w, int apply_negation);

hipError_t launch_line_counts(hipStream_t s, const uint8_t* file, const DecBlock* blocks, const DecLines& L,
                              uint32_t max_wah, uint32_t max_sparse, const uint32_t* d_totals);

// name of the element-major decode kernel launch_rank_decode picks (xsi_rank.hip)
const char* rank_decode_kernel_name(uint32_t N, uint32_t yp_stride, uint32_t n_blocks);

// element-major encode chain (xsi_rankenc.hip): blocks without fully haploid lines, N <= 65536
bool chain_rank_enc_supported(uint32_t N);
hipError_t launch_rank_encode(hipStream_t s, const EncBlock* blocks, uint32_t n_blocks, const EncLines& L);
// 65 536 < N <= 524 288: several workgroups per block (xsi_rankenc.hip)
constexpr uint32_t CHAIN_SYNC_WORDS = 16u + 256u;
constexpr uint32_t CHAIN_MAX_WGS = 256u;  // workgroups of the launch (one per CU)
constexpr uint64_t CHAIN_LIST_WORDS = (uint64_t)CHAIN_MAX_WGS * 2u * 16u * 4096u;
constexpr uint32_t CHAIN_LISTFLAG_WORDS = CHAIN_MAX_WGS * 2u * 16u * 2u;  // 8 bytes per wave and parity
constexpr uint32_t CHAIN_SLICEFLAG_WORDS = CHAIN_MAX_WGS * 2u * 16u * 2u;  // 8 bytes per wave and parity
constexpr uint32_t CHAIN_XCC_WORDS = (CHAIN_MAX_WGS / 2u) * 8u;
constexpr uint32_t CHAIN_SYNC_TOTAL_WORDS = CHAIN_SYNC_WORDS + CHAIN_LISTFLAG_WORDS + CHAIN_SLICEFLAG_WORDS + CHAIN_XCC_WORDS;
constexpr uint64_t CHAIN_SLICE_BYTES = (uint64_t)CHAIN_MAX_WGS * 2u * 16384u;
constexpr uint32_t CHAIN_ITEM_BEGIN_WORDS = 132u;  // the schedule's per-group begins (up to 128 groups + 1), then its items
constexpr uint64_t CHAIN_PARK_BYTES = (uint64_t)CHAIN_MAX_WGS * 64u * 1024u * 4u;  // per workgroup 64 x 1024 ranks
constexpr uint64_t CHAIN_BMP_BYTES = (uint64_t)CHAIN_MAX_WGS * 8u * 8192u;  // per workgroup a bitmap of up to 8 slices
bool chain_rank_enc_multi_supported(const EncLines& L);
// *refused = true (and hipSuccess, nothing enqueued): the device or CU mask cannot hold one group of workgroups, or the grid
// would exceed the exchange buffers - the caller takes k_chain_stream.  Every other error is a real one and is returned.
hipError_t launch_rank_encode_multi(hipStream_t s, const EncBlock* blocks, uint32_t n_blocks, const EncLines& L, bool* refused);

// ---- synthetic data ----
hipError_t launch_synth_packed(hipStream_t s, uint64_t seed, uint64_t first_line, uint64_t n_lines, uint32_t n_haps,
                               uint32_t* bits, uint32_t stride_w);

// chain kernel geometry (exposed for DESIGN.md / tuning)
struct ChainGeom {
    int threads, chunks;       // T, E
    uint32_t batch;            // columns prefetched per batch
    uint32_t lds_bytes;
    bool in_lds;
};
ChainGeom chain_geometry(uint32_t N, bool decode);
// name of the kernel that runs the PBWT chain of the blocks without fully haploid lines
const char* chain_kernel_name(uint32_t N, uint32_t n_blocks, bool decode);

}  // namespace xsi
