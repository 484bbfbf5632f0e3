// xsi_ctx.hpp — context, workspace and internal host-side declarations of libxsi_hip.so.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <map>
#include <string>
#include <vector>

#include "xsi_kernels.hpp"

enum {
    XSI_ST_COUNT = 0,      // row popcount
    XSI_ST_CLASSIFY,       // classify + scans + WAH line list
    XSI_ST_CHAIN_ENC,      // PBWT chain (encode)
    XSI_ST_WAH_SIZE,       // WAH16 sizing pass
    XSI_ST_LAYOUT,         // block layout + offsets
    XSI_ST_WRITE,          // dictionary + WAH + sparse writers
    XSI_ST_DEC_PARSE,      // dictionaries, flag vectors, line lists
    XSI_ST_DEC_BOUND,      // WAH line boundaries
    XSI_ST_DEC_EXPAND,     // WAH16 expand
    XSI_ST_CHAIN_DEC,      // PBWT chain (decode)
    XSI_ST_DEC_SPARSE,     // sparse walk + fill
    XSI_ST_GT_UNPACK,      // int32 rows -> planes (+ side sizing)
    XSI_ST_GT_COMPOSE,     // planes -> int32 rows (+ side planes)
    XSI_STAGE_COUNT
};

struct xsi_hip_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool owns_stream = false;
    // side stream for work that is independent of the serial chain (sparse lines): forked from and
    // joined back into `stream` with the two events, so callers still see one in-order stream
    hipStream_t side = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    // second side stream + events of the phased decode: the WAH expansion of the next range of lines runs
    // underneath the chain of the current one (created on first use)
    hipStream_t side2 = nullptr;
    bool low_priority = false;       // the context's own streams were made at the device's lowest priority (ctx_make_low_priority)
    std::vector<hipEvent_t> ev_phase;
    struct Buf {
        void* p = nullptr;
        size_t cap = 0;
    };
    std::map<std::string, Buf> bufs;  // named device workspace, grown on demand, reused across calls
    // Bytes of per-line workspace one encode / decode call may hold (0 = half of free + held HBM at the call, see ws_budget_now).
    // A job that needs more runs as several batches of whole blocks (blocks are independent).
    uint64_t ws_budget = 0;
    void* pinned = nullptr;           // pinned host staging
    size_t pinned_cap = 0;
    // optional per-stage timing with HIP events recorded on `stream` (bench.py roofline leg)
    bool timing = false;
    std::vector<hipEvent_t> ev_pool;
    std::vector<int> ev_stage;        // stage that STARTS at event i (-1 = end marker)
    size_t ev_used = 0;
    double stage_ms[XSI_STAGE_COUNT] = {0};
    uint64_t stage_n[XSI_STAGE_COUNT] = {0};
    uint64_t chain_fallbacks = 0;     // encode batches run again with the streaming chain after an aborted launch
    uint32_t reencode_ranges = 0;     // block ranges the last xsi_hip_reencode walked the file in
    // optional side output of the encode entry points (xsi_hip_ctx_set_block_sizes_out): bytes of every block of the call
    // before its pad to 4; `pos` = blocks of the running call already encoded (a call may run as several batches)
    uint32_t* block_sizes_out = nullptr;
    uint64_t block_sizes_cap = 0, block_sizes_pos = 0;
};

struct xsi_encode_params;
struct xsi_encode_result;

namespace xsi {
// Waits for the stream on EVERY exit of the scope it stands in until release() is called: for the stretches of a call in
// which asynchronous copies still read host memory the call does not own beyond its return (the caller's arrays, local
// vectors), or kernels still store into the caller's registered array.  An early error return there would otherwise hand
// that memory back while the GPU is using it.
struct StreamSyncGuard {
    hipStream_t s;
    bool armed = true;
    explicit StreamSyncGuard(hipStream_t stream) : s(stream) {}
    StreamSyncGuard(const StreamSyncGuard&) = delete;
    StreamSyncGuard& operator=(const StreamSyncGuard&) = delete;
    void release() { armed = false; }
    ~StreamSyncGuard() {
        if (armed) (void)hipStreamSynchronize(s);
    }
};
}  // namespace xsi

namespace xsi {

int set_error(int code, const char* fmt, ...);
// mark the start of `stage` on the stream (no-op unless timing is on); stage -1 ends the last one
void stage_mark(xsi_hip_ctx* c, int stage);
// after a stream sync: fold the recorded events into stage_ms / stage_n
void stage_collect(xsi_hip_ctx* c);
int ws_ensure(xsi_hip_ctx* c, const char* name, size_t bytes, void** out);
// background work (the accessor's read-ahead): the context's own stream and side streams at the lowest priority of the device,
// so that a foreground context's kernels are dispatched first.  Only for a context that owns its stream and has not run yet.
int ctx_make_low_priority(xsi_hip_ctx* c);
int pinned_ensure(xsi_hip_ctx* c, size_t bytes, void** out);

struct DecodePlan {
    uint32_t n_blocks = 0, version = 0;
    uint32_t n_bin = 0, n_bcf = 0, n_wah = 0, n_sparse = 0;
    uint64_t hap_samples = 0;
    bool has_side = false;
    DecBlock* d_blocks = nullptr;
    uint32_t* d_totals = nullptr;
    std::vector<DecBlock> blocks_h;
    DecLines L{};
    std::vector<uint32_t> phase_tab;  // host copy of the phase table (outlives its async upload)
};

struct DecodedPlanes {
    uint32_t* planes = nullptr;  // [n_bin][stride_w]; sparse lines hold their listed positions
    uint32_t stride_w = 0;
    bool has_side = false;
    uint8_t* side = nullptr;     // per binary line: bit0 missing, bit1 eov, bit2 phase
    uint32_t *miss_planes = nullptr, *eov_planes = nullptr, *phase_planes = nullptr;
    uint32_t *n_miss = nullptr, *n_eov = nullptr;
};

// part != nullptr: one block, only the WAH lines [wah_lo, wah_hi) of it (block-relative rank order), the chain's ranks
// parked in / resumed from d_state (4 * rank_decode_state_words(N, 1) bytes); first: everything that does not depend on
// the chain (sparse lines, side channels) is decoded as well, into the workspace; !first: more WAH lines into out->planes
struct PartialDecode {
    uint32_t wah_lo, wah_hi;
    uint32_t* d_state;
    bool first;
    // the binary lines this call makes valid: [bin_lo, bin_hi) (everything in front of the WAH line of rank wah_hi); the
    // side matrices (missing / end of vector / phase) of exactly these lines are walked and expanded, the walk's cursors
    // carried in d_walk[0..3)
    uint32_t bin_lo, bin_hi;
    uint64_t* d_walk;  // [0..3) side-matrix cursors, [3] the sparse matrix cursor
    // the plan still holds the WAH line starts of the block (kept by the accessor's cache entry since the first decode):
    // the boundary scan is not run again
    bool skip_boundaries = false;
};
int decode_all_planes(xsi_hip_ctx* ctx, const void* d_file, DecodePlan& P, DecodedPlanes* out, const PartialDecode* part = nullptr);
int dot_planes(xsi_hip_ctx* ctx, const DecodePlan& P, const uint32_t* planes, uint32_t stride_w, const double* d_y,
               uint32_t n_pheno, double* d_out);
int compose_lines(xsi_hip_ctx* ctx, const DecodePlan& P, const DecodedPlanes& D, const uint32_t* d_first_bin,
                  const uint32_t* d_n_allele, uint32_t n_out, int32_t* d_gt_out, uint64_t gt_stride,
                  uint32_t* d_line_ngt, uint64_t* d_allele_counts, uint32_t max_alleles,
                  const uint32_t* d_out_index = nullptr);
int select_samples(xsi_hip_ctx* ctx, const int32_t* d_rows, uint64_t row_stride, const uint32_t* d_line_ngt,
                   uint32_t n_lines, uint32_t n_samples, const uint32_t* d_sel, uint32_t n_sel, int32_t* d_out,
                   uint64_t out_stride, uint32_t* d_ac, uint32_t n_alt);
// counts_only: parse the dictionaries and flag vectors (blocks_h[b].n_wah, n_bin, ...) without allocating the expanded
// rows, the boundary tiles or anything else sized by the range's WAH lines (the batch-cutting pre-pass of a decode)
int decode_prepare(xsi_hip_ctx* ctx, const void* d_file, uint64_t file_len, uint64_t first_block, uint64_t n_blocks,
                   DecodePlan* P, bool counts_only = false);
int decode_counts_only(xsi_hip_ctx* ctx, const void* d_file, DecodePlan& P);
int decode_planes(xsi_hip_ctx* ctx, const void* d_file, DecodePlan& P, uint32_t* out, uint32_t stride_w,
                  int apply_negation);
bool decode_partial_supported(const DecodePlan& P);
int decode_planes_partial(xsi_hip_ctx* ctx, const void* d_file, DecodePlan& P, uint32_t* out, uint32_t stride_w,
                          uint32_t wah_lo, uint32_t wah_hi, uint32_t* d_state, uint32_t sp_lo, uint32_t sp_hi, uint64_t* d_sp_state,
                          bool skip_boundaries = false);
// (re)assigns the context-owned scratch of a plan; restore_totals: also the device totals, from the plan's host fields
int decode_plan_scratch(xsi_hip_ctx* ctx, DecodePlan* P, bool restore_totals = false);
// region_offset: bytes of the blocks region that earlier batches of the same job already wrote before d_out
int encode_run(xsi_hip_ctx* ctx, const xsi_encode_params* p, EncLines L, EncSide S, std::vector<EncBlock>& blocks_h,
               void* d_out, uint64_t out_capacity, uint64_t* d_block_offsets, xsi_encode_result* h_result,
               uint64_t region_offset = 0, bool use_wah_scratch = true);
// xsi_pack.cpp: one htslib row -> one bit per haplotype when the bit form can hold it (see there); out gets ceil(n / 8) bytes
bool pack_bit_row(const int32_t* gt, uint32_t n, int dp, uint8_t* out);
// the writer's packed lines back to htslib int32 rows on the device: row l of d_rows (stride N values) is rewritten
// from bit row l where d_fast[l] != 0; second values carry default_phased
int expand_bit_rows(xsi_hip_ctx* ctx, const uint8_t* d_bits, uint32_t bit_stride, const uint8_t* d_fast, int32_t* d_rows,
                    uint64_t N, uint64_t n_lines, int32_t default_phased);
// bytes of per-line workspace budget in force for a call made now
uint64_t ws_budget_now(const xsi_hip_ctx* ctx, const char* prefix = nullptr);
int encode_side_write(xsi_hip_ctx* ctx, const EncBlock* d_blocks, uint32_t n_blocks, const EncLines& L,
                      const EncSide& S, uint8_t* out, const uint64_t* d_result);

}  // namespace xsi
