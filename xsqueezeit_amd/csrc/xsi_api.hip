// xsi_api.hip — C ABI of libxsi_hip.so (include/xsi_hip.h): context, workspace, and the
// launch sequences of the encode / decode pipelines.  No CPU fallback: every entry point that
// does codec work needs a context, and a context needs a GPU.
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../../include/xsi_hip.h"
#include "xsi_ctx.hpp"
#include "xsi_kernels.hpp"

using namespace xsi;

namespace xsi {
static thread_local std::string g_err;
int set_error(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}
}  // namespace xsi

#define HIP_TRY(expr)                                                                                        \
    do {                                                                                                     \
        hipError_t _e = (expr);                                                                              \
        if (_e != hipSuccess) return set_error(XSI_ERR_HIP, "%s: %s (%s:%d)", #expr, hipGetErrorString(_e), \
                                               __FILE__, __LINE__);                                          \
    } while (0)

namespace xsi {
int ctx_make_low_priority(xsi_hip_ctx* c) {
    if (!c || !c->owns_stream) return set_error(XSI_ERR_ARG, "ctx_make_low_priority: a context with a stream of its own");
    int least = 0, greatest = 0;
    HIP_TRY(hipDeviceGetStreamPriorityRange(&least, &greatest));
    hipStream_t a = nullptr, b = nullptr;
    HIP_TRY(hipStreamCreateWithPriority(&a, hipStreamNonBlocking, least));
    hipError_t e = hipStreamCreateWithPriority(&b, hipStreamNonBlocking, least);
    if (e != hipSuccess) {
        (void)hipStreamDestroy(a);
        return set_error(XSI_ERR_HIP, "hipStreamCreateWithPriority: %s", hipGetErrorString(e));
    }
    (void)hipStreamDestroy(c->stream);
    if (c->side) (void)hipStreamDestroy(c->side);
    c->stream = a;
    c->side = b;
    c->low_priority = true;
    return XSI_OK;
}
}  // namespace xsi

extern "C" {

int xsi_hip_abi_version(void) { return XSI_HIP_ABI_VERSION; }
const char* xsi_hip_last_error(void) { return g_err.c_str(); }

int xsi_hip_ctx_create(xsi_hip_ctx** out, int device, void* stream) {
    if (!out) return set_error(XSI_ERR_ARG, "ctx_create: null out pointer");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return set_error(XSI_ERR_HIP, "no HIP device available (%s); this library has no CPU fallback",
                         e == hipSuccess ? "device count 0" : hipGetErrorString(e));
    if (device < 0 || device >= n) return set_error(XSI_ERR_ARG, "device %d out of range (0..%d)", device, n - 1);
    HIP_TRY(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return set_error(XSI_ERR_HIP, "device %d is %s; the kernels are built for gfx950 only", device, prop.gcnArchName);
    xsi_hip_ctx* c = new xsi_hip_ctx();
    c->device = device;
    if (stream) {
        c->stream = reinterpret_cast<hipStream_t>(stream);
        c->owns_stream = false;
    } else {
        hipError_t se = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
        if (se != hipSuccess) {
            delete c;
            return set_error(XSI_ERR_HIP, "hipStreamCreate: %s", hipGetErrorString(se));
        }
        c->owns_stream = true;
    }
    if (hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming) != hipSuccess) {
        xsi_hip_ctx_destroy(c);
        return set_error(XSI_ERR_HIP, "side stream / events could not be created");
    }
    *out = c;
    return XSI_OK;
}

void xsi_hip_ctx_destroy(xsi_hip_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    if (c->side) {
        (void)hipStreamSynchronize(c->side);
        (void)hipStreamDestroy(c->side);
    }
    if (c->side2) {
        (void)hipStreamSynchronize(c->side2);
        (void)hipStreamDestroy(c->side2);
    }
    for (auto e : c->ev_phase) (void)hipEventDestroy(e);
    if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
    if (c->ev_join) (void)hipEventDestroy(c->ev_join);
    for (auto& kv : c->bufs)
        if (kv.second.p) (void)hipFree(kv.second.p);
    if (c->pinned) (void)hipHostFree(c->pinned);
    for (auto e : c->ev_pool) (void)hipEventDestroy(e);
    if (c->owns_stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

int xsi_hip_ctx_synchronize(xsi_hip_ctx* c) {
    if (!c) return set_error(XSI_ERR_ARG, "null context");
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (c->side) HIP_TRY(hipStreamSynchronize(c->side));  // normally already joined; covers error exits
    if (c->side2) HIP_TRY(hipStreamSynchronize(c->side2));
    return XSI_OK;
}

uint64_t xsi_hip_ctx_workspace_bytes(const xsi_hip_ctx* c) {
    uint64_t t = 0;
    if (c)
        for (auto& kv : c->bufs) t += kv.second.cap;
    return t;
}

uint64_t xsi_hip_ctx_chain_fallbacks(const xsi_hip_ctx* c) { return c ? c->chain_fallbacks : 0; }
int xsi_hip_ctx_set_block_sizes_out(xsi_hip_ctx* c, uint32_t* d_sizes, uint64_t capacity_blocks) {
    if (!c || (d_sizes && !capacity_blocks)) return set_error(XSI_ERR_ARG, "set_block_sizes_out: null context or zero capacity");
    c->block_sizes_out = d_sizes;
    c->block_sizes_cap = d_sizes ? capacity_blocks : 0;
    c->block_sizes_pos = 0;
    return XSI_OK;
}

int xsi_hip_ctx_set_workspace_budget(xsi_hip_ctx* c, uint64_t bytes) {
    if (!c) return set_error(XSI_ERR_ARG, "null context");
    c->ws_budget = bytes;
    return XSI_OK;
}

int xsi_hip_ctx_set_timing(xsi_hip_ctx* c, int on) {
    if (!c) return set_error(XSI_ERR_ARG, "null context");
    c->timing = on != 0;
    c->ev_used = 0;
    for (int i = 0; i < XSI_STAGE_COUNT; ++i) {
        c->stage_ms[i] = 0;
        c->stage_n[i] = 0;
    }
    return XSI_OK;
}

int xsi_hip_ctx_get_timing(xsi_hip_ctx* c, double* h_ms, uint64_t* h_launches, int n) {
    if (!c || !h_ms || !h_launches) return set_error(XSI_ERR_ARG, "get_timing: null argument");
    for (int i = 0; i < n; ++i) {
        h_ms[i] = i < XSI_STAGE_COUNT ? c->stage_ms[i] : 0.0;
        h_launches[i] = i < XSI_STAGE_COUNT ? c->stage_n[i] : 0;
    }
    return XSI_STAGE_COUNT;
}

const char* xsi_hip_chain_kernel(uint32_t n_haps, uint64_t n_blocks, int decode) {
    return chain_kernel_name(n_haps, (uint32_t)(n_blocks > 0xFFFFFFFFull ? 0xFFFFFFFFull : n_blocks), decode != 0);
}

const char* xsi_hip_stage_name(int i) {
    static const char* names[XSI_STAGE_COUNT] = {"count_rows", "classify_scan", "chain_encode", "wah_size", "block_layout",
                                                 "write_blocks", "dec_parse_flags", "dec_wah_boundaries", "dec_wah_expand",
                                                 "chain_decode", "dec_sparse", "gt_unpack", "gt_compose"};
    return (i >= 0 && i < XSI_STAGE_COUNT) ? names[i] : "";
}

uint64_t xsi_hip_encode_bound(const xsi_encode_params* p, uint64_t n_bcf_lines, uint64_t n_binary_lines) {
    if (!p || !p->block_len) return 0;
    const uint64_t N = 2ull * p->n_samples;
    const uint64_t aet = p->n_samples <= 65535u ? 2 : 4;
    const uint64_t n_blocks = (n_bcf_lines + p->block_len - 1) / p->block_len;
    const uint64_t G = (N + 14) / 15;
    // per binary line: a WAH line of at most G words, or a sparse list of at most 1 + thr (+1) entries
    uint64_t per_line = G * 2;
    const uint64_t sp = (2ull + p->mac_threshold) * aet;
    if (sp > per_line) per_line = sp;
    // per block: outer dict 16, GT dict 8 + 18*8, five flag vectors, pad
    const uint64_t per_block = 16 + 8 + 18 * 8 + 5ull * FLAG_WORDS_MAX * 2 + 4;
    return n_blocks * per_block + n_binary_lines * per_line;
}

int xsi_hip_make_header(const xsi_header_fields* f, uint8_t h[256]) {
    if (!f || !h) return set_error(XSI_ERR_ARG, "make_header: null argument");
    memset(h, 0, 256);
    auto put = [&](size_t off, uint64_t v, int bytes) {
        for (int i = 0; i < bytes; ++i) h[off + i] = (uint8_t)(v >> (8 * i));
    };
    // header_t, compression.hpp:40-104, as filled by xsi_factory.hpp:468-500, 543-605
    put(0, 0xaabbccddu, 4);
    put(4, 0xfeed1767u, 4);
    put(8, 5, 4);
    h[12] = (uint8_t)f->max_ploidy;
    h[13] = 4;  // ind_bytes = sizeof(uint32_t) (stale: the v5 index is u64)
    h[14] = ((uint64_t)f->n_samples * 2 <= 65535u) ? 2 : 4;  // gt_compressor_new.hpp:177-187
    h[15] = 2;
    h[16] = (uint8_t)((f->default_phased ? 1 : 0) << 2);
    h[17] = (uint8_t)(1u | (f->zstd ? 4u : 0u));
    put(32, (uint64_t)f->n_samples * f->max_ploidy, 8);
    put(40, f->num_variants, 8);
    put(48, 0, 4);
    put(52, 1, 4);
    put(56, f->block_len, 4);
    put(60, f->block_len ? (uint32_t)((f->xcf_entries + f->block_len - 1) / f->block_len) : 0, 4);
    put(64, 256, 8);
    put(72, f->indices_offset, 8);
    put(80, f->samples_offset, 8);
    put(88, 0xFFFFFFFFu, 4);
    put(92, 0xFFFFFFFFu, 4);
    put(96, f->mac_threshold, 4);
    put(100, f->xcf_entries, 8);
    put(108, 0, 4);
    put(112, f->n_samples, 8);
    put(252, 0xfeed1767u, 4);
    return XSI_OK;
}

}  // extern "C"

namespace xsi {

uint64_t ws_budget_now(const xsi_hip_ctx* c, const char* grown) {
    if (c->ws_budget) return c->ws_budget;
    if (const char* e = tuning_env("XSI_WS_BUDGET_MB")) return (uint64_t)strtoull(e, nullptr, 10) << 20;
    // half of what is free now plus what this context already holds and will reuse - but never more than the buffer
    // that takes the per-line rows (`grown`: it is freed before it is allocated again) can actually get
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) return 8ull << 30;
    uint64_t held = 0, own = 0;
    for (auto& kv : c->bufs) {
        held += kv.second.cap;
        if (grown && kv.first == grown) own = kv.second.cap;
    }
    uint64_t b = (uint64_t)((free_b + held) * 0.5);
    if (grown) {
        const uint64_t reach = (uint64_t)((free_b + own) * 0.85);
        if (b > reach) b = reach;
    }
    return b;
}

void stage_mark(xsi_hip_ctx* c, int stage) {
    if (!c->timing) return;
    if (c->ev_used == c->ev_pool.size()) {
        hipEvent_t e;
        if (hipEventCreate(&e) != hipSuccess) return;
        c->ev_pool.push_back(e);
        c->ev_stage.push_back(-1);
    }
    c->ev_stage[c->ev_used] = stage;
    (void)hipEventRecord(c->ev_pool[c->ev_used], c->stream);
    c->ev_used++;
}

void stage_collect(xsi_hip_ctx* c) {
    if (!c->timing) return;
    for (size_t i = 0; i + 1 < c->ev_used; ++i) {
        const int st = c->ev_stage[i];
        if (st < 0 || st >= XSI_STAGE_COUNT) continue;
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, c->ev_pool[i], c->ev_pool[i + 1]) == hipSuccess) {
            c->stage_ms[st] += ms;
            c->stage_n[st] += 1;
        }
    }
    c->ev_used = 0;
}

int ws_ensure(xsi_hip_ctx* c, const char* name, size_t bytes, void** out) {
    auto& b = c->bufs[name];
    if (b.cap < bytes) {
        // growing a buffer that queued work may still read: drain the stream first
        hipError_t e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess) return set_error(XSI_ERR_HIP, "hipStreamSynchronize: %s", hipGetErrorString(e));
        if (b.p) (void)hipFree(b.p);
        b.p = nullptr;
        b.cap = 0;
        const size_t want = bytes + (bytes / 8 < (256u << 20) ? bytes / 8 : (size_t)(256u << 20)) + 256;  // room to grow, at most 256 MiB of it
        e = hipMalloc(&b.p, want);
        if (e != hipSuccess) return set_error(XSI_ERR_HIP, "hipMalloc(%zu) for %s: %s", want, name, hipGetErrorString(e));
        b.cap = want;
    }
    *out = b.p;
    return XSI_OK;
}

int pinned_ensure(xsi_hip_ctx* c, size_t bytes, void** out) {
    if (c->pinned_cap < bytes) {
        hipError_t e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess) return set_error(XSI_ERR_HIP, "hipStreamSynchronize: %s", hipGetErrorString(e));
        if (c->pinned) (void)hipHostFree(c->pinned);
        c->pinned = nullptr;
        c->pinned_cap = 0;
        e = hipHostMalloc(&c->pinned, bytes + 4096, hipHostMallocDefault);
        if (e != hipSuccess) return set_error(XSI_ERR_HIP, "hipHostMalloc(%zu): %s", bytes, hipGetErrorString(e));
        c->pinned_cap = bytes + 4096;
    }
    *out = c->pinned;
    return XSI_OK;
}

#define WS(ptr, name, bytes)                                                        \
    do {                                                                            \
        void* _p;                                                                   \
        int _rc = ws_ensure(ctx, name, (bytes), &_p);                               \
        if (_rc) return _rc;                                                        \
        ptr = reinterpret_cast<decltype(ptr)>(_p);                                  \
    } while (0)

// Shared tail of both encode entry points: everything after the per-line bit planes, counts
// and kinds exist.  Lines/side describe the batch; blocks_h holds the host-filled part.
// 8-byte words per permuted row
static uint32_t y_stride64_for(uint32_t N) {
    const uint32_t w = (N + 63u) / 64u;
    return N > 65536u ? (w + 1u) & ~1u : w;  // rows of whole 16-byte units: the chain over several workgroups stores them so
}

int encode_run(xsi_hip_ctx* ctx, const xsi_encode_params* p, EncLines L, EncSide S, std::vector<EncBlock>& blocks_h,
               void* d_out, uint64_t out_capacity, uint64_t* d_block_offsets, xsi_encode_result* h_result,
               uint64_t region_offset, bool use_wah_scratch) {
    hipStream_t s = ctx->stream;
    const uint32_t n_blocks = (uint32_t)blocks_h.size();
    const uint32_t n_bin = L.n_bin;
    const uint32_t N = L.N;
    const EncLines L_in = L;  // as the caller made it: what a second run after an aborted chain launch starts from

    EncBlock* d_blocks;
    WS(d_blocks, "enc.blocks", sizeof(EncBlock) * (size_t)n_blocks);
    WS(L.line_block, "enc.line_block", 4ull * n_bin);
    WS(L.wah_rank, "enc.wah_rank", 4ull * n_bin);
    WS(L.sparse_off, "enc.sparse_off", 4ull * n_bin);
    WS(L.wah_lines, "enc.wah_lines", 4ull * n_bin);
    WS(L.wah_len, "enc.wah_len", 4ull * n_bin);
    WS(L.wah_off, "enc.wah_off", 4ull * n_bin);
    L.y_stride64 = y_stride64_for(N);
    const size_t y_rows = L.y_rows ? L.y_rows : n_bin;  // one per WAH line: exact when the caller counted them
    WS(L.yrows, "ws.rows", 8ull * L.y_stride64 * y_rows);  // one buffer with the decode's expanded rows: a context runs one call at a time
    if (N > 65536u && N <= 524288u) {  // the chain over several workgroups per block
        WS(L.chain_sync, "enc.chain_sync", 4ull * CHAIN_SYNC_TOTAL_WORDS);
        WS(L.chain_lists, "enc.chain_lists", 4ull * CHAIN_LIST_WORDS);
        WS(L.chain_slices, "enc.chain_slices", CHAIN_SLICE_BYTES);
        WS(L.chain_bmps, "enc.chain_bmps", CHAIN_BMP_BYTES);
        WS(L.chain_items, "enc.chain_items", 4ull * CHAIN_ITEM_BEGIN_WORDS + 16ull * ((size_t)n_blocks + CHAIN_MAX_WGS / 2u + 2u));
        WS(L.chain_park, "enc.chain_park", CHAIN_PARK_BYTES);
    }
    // WAH16 words per line, worst case ceil(N/15) (+1 for the saturation split): encode once, then copy
    L.wah_scratch_stride = ((N + 14u) / 15u + 3u) & ~1u;  // even: rows stay 4-byte aligned for k_wah_write
    L.wah_scratch = nullptr;  // without it the lines are sized first and encoded again straight into place
    if (use_wah_scratch && !wah_units_any(L.y_stride64)) WS(L.wah_scratch, "enc.wah_scratch", 2ull * L.wah_scratch_stride * y_rows);
    // the unit encoders read a row into LDS whole: the sizing pass can leave the words in the row's place (read per call: testing)
    L.wah_inplace = wah_units_any(L.y_stride64) ? 1u : 0u;
    WS(L.flagbits, "enc.flagbits", 4ull * (MAX_BIN_PER_BLOCK / 32) * FV_COUNT * (size_t)n_blocks);
    WS(L.flagwah, "enc.flagwah", 2ull * FLAG_WORDS_MAX * FV_COUNT * (size_t)n_blocks);
    uint32_t* d_totals;
    WS(d_totals, "enc.totals", 64);
    uint64_t* d_result;
    WS(d_result, "enc.result", 64);
    uint32_t* scratch_a = nullptr;
    const ChainGeom g = chain_geometry(N, false);
    if (!g.in_lds) WS(scratch_a, "chain.a", 4ull * 2ull * (((size_t)N + 63u) & ~(size_t)63u) * n_blocks);

    HIP_TRY(hipMemcpyAsync(d_blocks, blocks_h.data(), sizeof(EncBlock) * (size_t)n_blocks, hipMemcpyHostToDevice, s));
    stage_mark(ctx, XSI_ST_CLASSIFY);
    HIP_TRY(launch_classify(s, d_blocks, n_blocks, L));
    HIP_TRY(launch_scan_blocks_wah(s, d_blocks, n_blocks, d_totals));
    HIP_TRY(launch_build_wah_list(s, d_blocks, n_blocks, L));
    // Sparse lists depend on the classification only (their sizes are known from the counts), not on
    // the chain: emit them into a scratch on the side stream underneath the chain, one fixed-size
    // region per block, and move them into place once the block layout exists.
    uint8_t* sp_scratch = nullptr;
    uint64_t sp_stride = 0;
    {
        uint32_t max_bin = 0;
        for (auto& b : blocks_h) max_bin = b.n_bin > max_bin ? b.n_bin : max_bin;
        const uint64_t per_line = (1ull + (p->mac_threshold < N / 2u ? p->mac_threshold : N / 2u)) * L.aet;
        sp_stride = (max_bin * per_line + 255u) & ~255ull;
        // (worst-case regions: 2.5 GB for the 153 blocks of a configs[3] shard, where the sparse lists take 19 ms if they wait for the chain)
        if (sp_stride * n_blocks <= (4ull << 30) && !tuning_env("XSI_NO_SPARSE_OVERLAP")) WS(sp_scratch, "enc.sparse_scratch", sp_stride * n_blocks);
    }
    // The chain over several workgroups per block fills every CU's registers with workgroups that wait for one
    // another: sparse work started first holds CUs back from it for as long as it runs (+18 ms of chain for 19 ms of
    // sparse lists at the configs[3] shard, and a group that stays incomplete long enough aborts the launch).  There
    // the sparse lists start behind the chain and the WAH sizing pass.
    const bool sparse_behind_chain = chain_rank_enc_multi_supported(L);
    auto fork_sparse = [&]() -> int {
        HIP_TRY(hipEventRecord(ctx->ev_fork, s));
        HIP_TRY(hipStreamWaitEvent(ctx->side, ctx->ev_fork, 0));
        HIP_TRY(launch_sparse_write(ctx->side, d_blocks, L, nullptr, nullptr, sp_scratch, sp_stride));
        HIP_TRY(hipEventRecord(ctx->ev_join, ctx->side));
        return XSI_OK;
    };
    if (sp_scratch && !sparse_behind_chain)
        if (int rc = fork_sparse()) return rc;
    stage_mark(ctx, XSI_ST_CHAIN_ENC);
    bool multi_refused = false;
    HIP_TRY(launch_chain_encode(s, d_blocks, n_blocks, L, scratch_a, L.bin_nbits != nullptr, &multi_refused));
    if (multi_refused) ctx->chain_fallbacks++;  // (the long-row kernel was eligible but could not be launched: the streaming chain ran)
    stage_mark(ctx, XSI_ST_WAH_SIZE);
    HIP_TRY(launch_wah_sizes(s, L, d_totals, n_bin));
    // (behind the sizing pass too: its 1024-thread workgroups take every wave slot, the lists would only queue up
    // with it; they run beside the layout and the writing of the WAH words instead)
    if (sp_scratch && sparse_behind_chain)
        if (int rc = fork_sparse()) return rc;
    stage_mark(ctx, XSI_ST_LAYOUT);
    HIP_TRY(launch_block_layout(s, d_blocks, n_blocks, L, S, p->default_phased));
    uint32_t* d_block_sizes = nullptr;
    if (ctx->block_sizes_out) {
        if (ctx->block_sizes_pos + n_blocks > ctx->block_sizes_cap)
            return set_error(XSI_ERR_CAPACITY, "encode: the block-sizes side output holds %llu blocks, the call has more",
                             (unsigned long long)ctx->block_sizes_cap);
        d_block_sizes = ctx->block_sizes_out + ctx->block_sizes_pos;
    }
    HIP_TRY(launch_scan_blocks_out(s, d_blocks, n_blocks, out_capacity, d_block_offsets, d_result, 256u + region_offset, d_block_sizes));
    const uint32_t strategy = p->wah_encode_missing ? WS_WAH : WS_SPARSE;
    stage_mark(ctx, XSI_ST_WRITE);
    HIP_TRY(launch_write_headers(s, d_blocks, n_blocks, L, p->default_phased, strategy, (uint8_t*)d_out, d_result));
    HIP_TRY(launch_wah_write(s, d_blocks, L, n_bin, (uint8_t*)d_out, d_result));
    if (sp_scratch) {
        HIP_TRY(hipStreamWaitEvent(s, ctx->ev_join, 0));
        HIP_TRY(launch_sparse_copy(s, d_blocks, n_blocks, (uint8_t*)d_out, d_result, sp_scratch, sp_stride));
    } else {
        HIP_TRY(launch_sparse_write(s, d_blocks, L, (uint8_t*)d_out, d_result, nullptr, 0));
    }
    if (S.bcf_flags) {
        int rc = encode_side_write(ctx, d_blocks, n_blocks, L, S, (uint8_t*)d_out, d_result);
        if (rc) return rc;
    }
    stage_mark(ctx, -1);
    uint64_t res[5];
    uint32_t chain_abort = 0;
    HIP_TRY(hipMemcpyAsync(res, d_result, sizeof(res), hipMemcpyDeviceToHost, s));
    if (chain_rank_enc_multi_supported(L)) HIP_TRY(hipMemcpyAsync(&chain_abort, L.chain_sync, sizeof(chain_abort), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    stage_collect(ctx);
    if (chain_abort) {
        // The workgroups of a block did not all become resident (another process or stream holds CUs) and a wait
        // ran out: every workgroup has left, the output is incomplete.  Run the batch again, in this call, with the
        // one-workgroup-per-block streaming chain, which needs no co-residency.
        if (L_in.no_multi) return set_error(XSI_ERR_HIP, "encode: chain launch aborted");
        EncLines L2 = L_in;
        L2.no_multi = 1u;
        ctx->chain_fallbacks++;
        return encode_run(ctx, p, L2, S, blocks_h, d_out, out_capacity, d_block_offsets, h_result, region_offset, use_wah_scratch);
    }
    if (res[3]) return set_error(XSI_ERR_CAPACITY, "encode: output needs %llu bytes, capacity is %llu",
                                 (unsigned long long)res[0], (unsigned long long)out_capacity);
    if (ctx->block_sizes_out) ctx->block_sizes_pos += n_blocks;  // (the next batch of the same call goes behind these)
    if (h_result) {
        h_result->n_blocks = n_blocks;
        h_result->blocks_bytes = res[0];
        h_result->n_binary_lines = n_bin;
        h_result->n_wah_lines = res[2];
        h_result->max_ploidy = 2;
        h_result->last_block_bytes = (uint32_t)res[4];
    }
    return XSI_OK;
}

}  // namespace xsi

extern "C" {

int xsi_hip_encode_packed(xsi_hip_ctx* ctx, const xsi_encode_params* p, const void* d_bits, uint64_t n_lines,
                          uint32_t row_stride_bytes, void* d_out, uint64_t out_capacity, uint64_t* d_block_offsets,
                          xsi_encode_result* h_result) {
    return xsi_hip_encode_packed_counted(ctx, p, d_bits, n_lines, row_stride_bytes, nullptr, d_out, out_capacity, d_block_offsets,
                                         h_result);
}

int xsi_hip_count_packed_rows(xsi_hip_ctx* ctx, const void* d_bits, uint64_t n_lines, uint32_t row_stride_bytes, uint32_t n_haps,
                              uint32_t* d_row_counts) {
    if (!ctx || !d_bits || !d_row_counts) return set_error(XSI_ERR_ARG, "count_packed_rows: null argument");
    if (row_stride_bytes % 8u || (uint64_t)row_stride_bytes * 8u < n_haps || n_lines > 0x7FFFFFFFull)
        return set_error(XSI_ERR_ARG, "count_packed_rows: bad stride or line count");
    if (!n_lines) return XSI_OK;
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(launch_count_rows(ctx->stream, reinterpret_cast<const uint32_t*>(d_bits), row_stride_bytes / 4u, n_haps, (uint32_t)n_lines,
                              d_row_counts));
    return XSI_OK;
}

int xsi_hip_encode_packed_counted(xsi_hip_ctx* ctx, const xsi_encode_params* p, const void* d_bits, uint64_t n_lines,
                                  uint32_t row_stride_bytes, const uint32_t* d_row_counts, void* d_out, uint64_t out_capacity,
                                  uint64_t* d_block_offsets, xsi_encode_result* h_result) {
    if (!ctx || !p || !d_bits || !d_out) return set_error(XSI_ERR_ARG, "encode_packed: null argument");
    ctx->block_sizes_pos = 0;
    if (!p->n_samples || !p->block_len) return set_error(XSI_ERR_ARG, "encode_packed: n_samples and block_len must be > 0");
    if (p->block_len > MAX_BIN_PER_BLOCK)
        return set_error(XSI_ERR_ARG, "block_len %u exceeds the BM offset range (%u binary lines per block)", p->block_len,
                         MAX_BIN_PER_BLOCK);
    if (at_mismatch_window(p->n_samples))
        return set_error(XSI_ERR_UNSUPPORTED, "%s: %u samples fall in the reference's A_T mismatch window (32768..65535: 16-bit "
                         "block data under a 32-bit header, prefix array wraps modulo 65536); it cannot be encoded decodably", "encode_packed",
                         p->n_samples);
    const uint64_t N64 = 2ull * p->n_samples;
    if (row_stride_bytes % 8u || (uint64_t)row_stride_bytes * 8u < N64)
        return set_error(XSI_ERR_ARG, "row_stride_bytes %u must be a multiple of 8 and hold %llu bits", row_stride_bytes,
                         (unsigned long long)N64);
    if (n_lines == 0 || n_lines > 0x7FFFFFFFull) return set_error(XSI_ERR_ARG, "n_lines %llu out of range", (unsigned long long)n_lines);
    HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    const uint32_t N = (uint32_t)N64;
    const uint64_t n_blocks_all = (n_lines + p->block_len - 1) / p->block_len;
    // Ones of every line first: they decide which lines are WAH lines, i.e. how many permuted rows y each block
    // needs (the bulk of the workspace: 62.5 KB per WAH line at 500 000 haplotypes).
    uint32_t* cnt_all;
    uint8_t* kind_all;
    uint32_t* d_wah_per_block;
    WS(cnt_all, "enc.cnt", 4ull * n_lines);
    WS(kind_all, "enc.kind", (size_t)n_lines);
    WS(d_wah_per_block, "enc.wah_per_block", 4ull * n_blocks_all);
    HIP_TRY(hipMemsetAsync(kind_all, 0, n_lines, s));
    stage_mark(ctx, XSI_ST_COUNT);
    if (d_row_counts) {
        // the producer of the rows counted them (the writer's packer does, one popcount per mask): the pass over the
        // matrix that GtBlock::scan_genotypes stands for (gt_block.hpp:207-269) has been made already
        if (tuning_env("XSI_CHECK_ROW_COUNTS")) {
            HIP_TRY(launch_count_rows(s, reinterpret_cast<const uint32_t*>(d_bits), row_stride_bytes / 4u, N, (uint32_t)n_lines, cnt_all));
            uint32_t* d_bad;
            WS(d_bad, "enc.cnt_check", 4);
            HIP_TRY(hipMemsetAsync(d_bad, 0, 4, s));
            HIP_TRY(launch_compare_u32(s, cnt_all, d_row_counts, n_lines, d_bad));
            uint32_t bad = 0;
            HIP_TRY(hipMemcpyAsync(&bad, d_bad, 4, hipMemcpyDeviceToHost, s));
            HIP_TRY(hipStreamSynchronize(s));
            if (bad) return set_error(XSI_ERR_ARG, "encode_packed_counted: %u of the supplied row counts differ from the rows", bad);
        }
        HIP_TRY(hipMemcpyAsync(cnt_all, d_row_counts, 4ull * n_lines, hipMemcpyDeviceToDevice, s));
    } else {
        HIP_TRY(launch_count_rows(s, reinterpret_cast<const uint32_t*>(d_bits), row_stride_bytes / 4u, N, (uint32_t)n_lines, cnt_all));
    }
    HIP_TRY(launch_wah_lines_per_block(s, cnt_all, n_lines, p->block_len, N, p->mac_threshold, d_wah_per_block));
    stage_mark(ctx, -1);
    std::vector<uint32_t> wah_per_block((size_t)n_blocks_all);
    HIP_TRY(hipMemcpyAsync(wah_per_block.data(), d_wah_per_block, 4ull * n_blocks_all, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    // Per-line workspace: the permuted row y of a WAH line and (rows above 8 KiB only) its WAH16 words, encoded
    // once and copied into place.  A job whose workspace exceeds the budget drops the WAH scratch first (lines are
    // then sized and encoded twice), then runs as batches of whole blocks: blocks are independent
    // (gt_block.hpp:179-180, xsi_factory.hpp:527-539), so the bytes are those of a single call.
    const bool units = wah_units_any(y_stride64_for(N));
    const uint64_t y_line = 8ull * y_stride64_for(N), scratch_line = units ? 0 : 2ull * (((N + 14u) / 15u + 3u) & ~1u), misc_line = 40;
    const uint64_t budget = ws_budget_now(ctx, "ws.rows");
    uint64_t wah_all = 0;
    for (uint32_t c : wah_per_block) wah_all += c;
    const bool use_scratch = (y_line + scratch_line) * wah_all + misc_line * n_lines <= budget;
    const uint64_t row_bytes = y_line + (use_scratch ? scratch_line : 0);
    auto block_need = [&](uint64_t b) {
        const uint64_t lines = (b + 1) * p->block_len <= n_lines ? p->block_len : n_lines - b * p->block_len;
        return row_bytes * wah_per_block[b] + misc_line * lines;
    };
    // batches of about equal need, none above the budget (a single block may exceed it: it runs alone)
    uint64_t need_all = 0;
    for (uint64_t b = 0; b < n_blocks_all; ++b) need_all += block_need(b);
    uint64_t n_batches = (need_all + budget - 1) / (budget ? budget : 1);
    if (n_batches < 1) n_batches = 1;
    const uint64_t target = (need_all + n_batches - 1) / n_batches;
    xsi_encode_result total{};
    uint64_t region_off = 0;
    for (uint64_t b0 = 0; b0 < n_blocks_all;) {
        uint64_t nb = 0, acc = 0, rows = 0;
        while (b0 + nb < n_blocks_all) {
            const uint64_t nd = block_need(b0 + nb);
            if (nb && (acc + nd > budget || acc >= target)) break;
            acc += nd;
            rows += wah_per_block[b0 + nb];
            ++nb;
        }
        const uint64_t l0 = b0 * p->block_len;
        const uint64_t nl = (l0 + nb * p->block_len <= n_lines) ? nb * p->block_len : n_lines - l0;
        const uint32_t n_bin = (uint32_t)nl;
        std::vector<EncBlock> blocks((size_t)nb);
        for (uint32_t b = 0; b < (uint32_t)nb; ++b) {
            memset(&blocks[b], 0, sizeof(EncBlock));
            blocks[b].first_bcf = blocks[b].first_bin = b * p->block_len;
            const uint64_t left = nl - (uint64_t)b * p->block_len;
            blocks[b].n_bcf = blocks[b].n_bin = (uint32_t)(left < p->block_len ? left : p->block_len);
        }
        EncLines L{};
        L.planes = reinterpret_cast<const uint32_t*>(reinterpret_cast<const uint8_t*>(d_bits) + l0 * row_stride_bytes);
        L.plane_stride_w = row_stride_bytes / 4u;
        L.n_bin = n_bin;
        L.N = N;
        L.aet = p->n_samples <= 65535u ? 2u : 4u;  // xsi_factory.hpp:424-427
        L.thr = p->mac_threshold;
        L.cnt = cnt_all + l0;
        L.kind = kind_all + l0;
        L.y_rows = rows ? (uint32_t)rows : 1u;
        EncSide S{};
        xsi_encode_result r{};
        if (region_off > out_capacity) return set_error(XSI_ERR_CAPACITY, "encode: output capacity exhausted");
        int rc = encode_run(ctx, p, L, S, blocks, reinterpret_cast<uint8_t*>(d_out) + region_off, out_capacity - region_off,
                            d_block_offsets ? d_block_offsets + b0 : nullptr, &r, region_off, use_scratch);
        if (rc) return rc;
        region_off += r.blocks_bytes;
        total.n_blocks += r.n_blocks;
        total.n_binary_lines += r.n_binary_lines;
        total.n_wah_lines += r.n_wah_lines;
        total.max_ploidy = r.max_ploidy;
        total.last_block_bytes = r.last_block_bytes;
        b0 += nb;
    }
    total.blocks_bytes = region_off;
    if (h_result) *h_result = total;
    return XSI_OK;
}

int xsi_hip_debug_chain_encode(xsi_hip_ctx* ctx, const xsi_encode_params* p, const void* d_bits, uint64_t n_lines,
                               uint32_t row_stride_bytes, void* d_yrows, uint32_t y_stride_bytes, uint32_t* d_line_kind,
                               uint64_t* h_n_wah) {
    if (!ctx || !p || !d_bits || !d_yrows) return set_error(XSI_ERR_ARG, "debug_chain_encode: null argument");
    if (at_mismatch_window(p->n_samples))
        return set_error(XSI_ERR_UNSUPPORTED, "%s: %u samples fall in the reference's A_T mismatch window (32768..65535: 16-bit "
                         "block data under a 32-bit header, prefix array wraps modulo 65536); it cannot be encoded decodably", "debug_chain_encode",
                         p->n_samples);
    const uint32_t N = 2u * p->n_samples;
    if (y_stride_bytes % 8u || y_stride_bytes * 8ull < (((uint64_t)N + 63u) & ~63ull))
        return set_error(XSI_ERR_ARG, "y_stride_bytes too small");
    HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    const uint32_t n_bin = (uint32_t)n_lines;
    const uint32_t n_blocks = (uint32_t)((n_lines + p->block_len - 1) / p->block_len);
    std::vector<EncBlock> blocks(n_blocks);
    for (uint32_t b = 0; b < n_blocks; ++b) {
        memset(&blocks[b], 0, sizeof(EncBlock));
        blocks[b].first_bcf = blocks[b].first_bin = b * p->block_len;
        const uint64_t left = n_lines - (uint64_t)b * p->block_len;
        blocks[b].n_bcf = blocks[b].n_bin = (uint32_t)(left < p->block_len ? left : p->block_len);
    }
    EncLines L{};
    L.planes = reinterpret_cast<const uint32_t*>(d_bits);
    L.plane_stride_w = row_stride_bytes / 4u;
    L.n_bin = n_bin;
    L.N = N;
    L.aet = p->n_samples <= 65535u ? 2u : 4u;
    L.thr = p->mac_threshold;
    EncBlock* d_blocks;
    WS(d_blocks, "enc.blocks", sizeof(EncBlock) * (size_t)n_blocks);
    WS(L.cnt, "enc.cnt", 4ull * n_bin);
    WS(L.kind, "enc.kind", (size_t)n_bin);
    WS(L.line_block, "enc.line_block", 4ull * n_bin);
    WS(L.wah_rank, "enc.wah_rank", 4ull * n_bin);
    WS(L.sparse_off, "enc.sparse_off", 4ull * n_bin);
    WS(L.wah_lines, "enc.wah_lines", 4ull * n_bin);
    WS(L.flagbits, "enc.flagbits", 4ull * (MAX_BIN_PER_BLOCK / 32) * FV_COUNT * (size_t)n_blocks);
    uint32_t* d_totals;
    WS(d_totals, "enc.totals", 64);
    L.yrows = reinterpret_cast<uint64_t*>(d_yrows);
    L.y_stride64 = y_stride_bytes / 8u;
    uint32_t* scratch_a = nullptr;
    if (!chain_geometry(N, false).in_lds) WS(scratch_a, "chain.a", 4ull * 2ull * (((size_t)N + 63u) & ~(size_t)63u) * n_blocks);
    HIP_TRY(hipMemsetAsync(L.kind, 0, n_bin, s));
    HIP_TRY(hipMemcpyAsync(d_blocks, blocks.data(), sizeof(EncBlock) * (size_t)n_blocks, hipMemcpyHostToDevice, s));
    HIP_TRY(launch_count_rows(s, L.planes, L.plane_stride_w, N, n_bin, L.cnt));
    HIP_TRY(launch_classify(s, d_blocks, n_blocks, L));
    HIP_TRY(launch_scan_blocks_wah(s, d_blocks, n_blocks, d_totals));
    HIP_TRY(launch_build_wah_list(s, d_blocks, n_blocks, L));
    HIP_TRY(launch_chain_encode(s, d_blocks, n_blocks, L, scratch_a, L.bin_nbits != nullptr));
    uint32_t tot[2] = {0, 0};
    HIP_TRY(hipMemcpyAsync(tot, d_totals, 4, hipMemcpyDeviceToHost, s));
    if (d_line_kind) {
        // widen kinds to uint32 on the host side of the test: copy raw bytes
        HIP_TRY(hipMemcpyAsync(d_line_kind, L.kind, n_bin, hipMemcpyDeviceToDevice, s));
    }
    HIP_TRY(hipStreamSynchronize(s));
    if (h_n_wah) *h_n_wah = tot[0];
    return XSI_OK;
}

int xsi_hip_decode_packed(xsi_hip_ctx* ctx, const void* d_file, uint64_t file_len, uint64_t first_block,
                          uint64_t n_blocks64, void* d_bits_out, uint32_t row_stride_bytes, uint64_t out_rows_capacity,
                          uint64_t* h_rows_written, uint32_t* d_counts) {
    if (!ctx || !d_file || !d_bits_out) return set_error(XSI_ERR_ARG, "decode_packed: null argument");
    if (file_len < 256) return set_error(XSI_ERR_FORMAT, "file image shorter than the 256-byte header");
    HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    // Per-line workspace of the decode is the expanded row of every WAH line ({bits, prefix} pairs, N/4
    // bytes); a block range that needs more than the budget is decoded as batches of whole blocks.
    std::vector<uint64_t> cuts;  // batch k = blocks [cuts[k], cuts[k + 1])
    cuts.push_back(0);
    {
        uint8_t h[256];
        HIP_TRY(hipMemcpyAsync(h, d_file, 256, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        auto get = [&](size_t off, int bytes) {
            uint64_t v = 0;
            for (int i = 0; i < bytes; ++i) v |= (uint64_t)h[off + i] << (8 * i);
            return v;
        };
        const uint64_t ns = get(112, 8), N64 = ns ? ns * 2 : get(32, 8), bl = get(56, 4) ? get(56, 4) : 8192;
        const uint64_t per_line = 16ull * ((N64 + 63u) / 64u) + 64u;
        const uint64_t budget = ws_budget_now(ctx, "ws.rows");
        if (n_blocks64 * per_line * bl > budget && n_blocks64 > 1) {
            // The worst case (every line a WAH line) does not fit: take the blocks' actual WAH line counts (one parse
            // of the dictionaries and flag vectors, no expansion) and cut batches that fit.
            // Long rows are decoded by several workgroups per block (8 at 500 000 haplotypes), one per CU: batches of
            // whole rounds of the chip (multiples of 256 / 8 blocks) leave no half-empty last round, the remainder
            // comes last.
            std::vector<uint32_t> n_wah(n_blocks64);
            {
                DecodePlan plan;
                int rc = decode_prepare(ctx, d_file, file_len, first_block, n_blocks64, &plan, /*counts_only=*/true);
                if (rc) return rc;
                for (uint64_t b = 0; b < n_blocks64; ++b) n_wah[b] = plan.blocks_h[b].n_wah;
            }
            const uint32_t yps = (uint32_t)((((N64 + 31u) / 32u) + 1u) & ~1ull);
            // (the geometry of a launch that fills the chip - the same selection the launcher makes - on this device's CUs)
            const uint64_t cus = rank_decode_cus();
            const uint64_t wgs = rank_decode_big_wgs_per_block((uint32_t)N64, yps, (uint32_t)(n_blocks64 < cus ? n_blocks64 : cus));
            const uint64_t quantum = wgs > 1 && cus / wgs > 1u ? cus / wgs : 1u;
            uint64_t b0 = 0;
            while (b0 < n_blocks64) {
                uint64_t need = 0, nb = 0, best = 0;
                while (b0 + nb < n_blocks64 && need + (uint64_t)n_wah[b0 + nb] * per_line <= budget) {
                    need += (uint64_t)n_wah[b0 + nb] * per_line;
                    ++nb;
                    if (nb % quantum == 0) best = nb;
                }
                if (b0 + nb == n_blocks64 || best == 0) best = nb ? nb : 1;  // the tail, or fewer blocks fit than a round holds
                // a remainder of less than one round would run at a fraction of the chip (128 + 25 of 153 blocks: the 25 on
                // 200 of 256 CUs, or on smaller workgroups that stage the row twice as often): a round less here, a round
                // more there (96 + 57)
                const uint64_t rest = n_blocks64 - (b0 + best);
                if (rest && rest < quantum && best % quantum == 0 && best >= 2u * quantum) best -= quantum;
                b0 += best;
                cuts.push_back(b0);
            }
        } else {
            cuts.push_back(n_blocks64);
        }
    }
    uint64_t rows_done = 0;
    for (size_t k = 0; k + 1 < cuts.size() || k == 0; ++k) {
        const uint64_t b0 = cuts[k];
        const uint64_t nb = (k + 1 < cuts.size() ? cuts[k + 1] : n_blocks64) - b0;
        DecodePlan P;
        int rc = decode_prepare(ctx, d_file, file_len, first_block + b0, nb, &P);
        if (rc) return rc;
        if (row_stride_bytes % 8u || (uint64_t)row_stride_bytes * 8u < P.L.N)
            return set_error(XSI_ERR_ARG, "row_stride_bytes %u must be a multiple of 8 and hold %u bits", row_stride_bytes, P.L.N);
        if (rows_done + P.n_bin > out_rows_capacity)
            return set_error(XSI_ERR_CAPACITY, "decode_packed: %llu rows, capacity %llu", (unsigned long long)(rows_done + P.n_bin),
                             (unsigned long long)out_rows_capacity);
        if (P.has_side)
            return set_error(XSI_ERR_UNSUPPORTED, "decode_packed: blocks carry missing / end-of-vector / phase / haploid data; use xsi_hip_decode_gt");
        if (P.n_bin != P.n_bcf)
            return set_error(XSI_ERR_UNSUPPORTED, "decode_packed: multi-allelic lines present; use xsi_hip_decode_gt");
        uint32_t* out = reinterpret_cast<uint32_t*>(reinterpret_cast<uint8_t*>(d_bits_out) + rows_done * row_stride_bytes);
        const uint32_t stride_w = row_stride_bytes / 4u;
        rc = decode_planes(ctx, d_file, P, out, stride_w, /*apply_negation=*/1);
        if (rc) return rc;
        if (d_counts) HIP_TRY(hipMemcpyAsync(d_counts + rows_done, P.L.ones, 4ull * P.n_bin, hipMemcpyDeviceToDevice, s));
        HIP_TRY(hipStreamSynchronize(s));
        stage_collect(ctx);
        rows_done += P.n_bin;
        if (n_blocks64 == 0) break;
    }
    if (h_rows_written) *h_rows_written = rows_done;
    return XSI_OK;
}

int xsi_hip_decode_counts(xsi_hip_ctx* ctx, const void* d_file, uint64_t file_len, uint64_t first_block,
                           uint64_t n_blocks64, uint32_t* d_ones, uint8_t* d_kind, uint64_t capacity, uint64_t* h_n_bin) {
    if (!ctx || !d_file || !d_ones) return set_error(XSI_ERR_ARG, "decode_counts: null argument");
    if (file_len < 256) return set_error(XSI_ERR_FORMAT, "file image shorter than the 256-byte header");
    HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    DecodePlan P;
    int rc = decode_prepare(ctx, d_file, file_len, first_block, n_blocks64, &P);
    if (rc) return rc;
    if (P.n_bin > capacity) return set_error(XSI_ERR_CAPACITY, "decode_counts: %u binary lines, capacity %llu", P.n_bin,
                                             (unsigned long long)capacity);
    rc = decode_counts_only(ctx, d_file, P);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(d_ones, P.L.ones, 4ull * P.n_bin, hipMemcpyDeviceToDevice, s));
    if (d_kind) HIP_TRY(hipMemcpyAsync(d_kind, P.L.kind, P.n_bin, hipMemcpyDeviceToDevice, s));
    HIP_TRY(hipStreamSynchronize(s));
    stage_collect(ctx);
    if (h_n_bin) *h_n_bin = P.n_bin;
    return XSI_OK;
}

int xsi_hip_decode_dot(xsi_hip_ctx* ctx, const void* d_file, uint64_t file_len, uint64_t first_block,
                       uint64_t n_blocks64, const double* d_pheno, uint32_t n_pheno, double* d_out, uint64_t capacity,
                       uint64_t* h_n_bin) {
    if (!ctx || !d_file || !d_pheno || !d_out) return set_error(XSI_ERR_ARG, "decode_dot: null argument");
    if (!n_pheno) return set_error(XSI_ERR_ARG, "decode_dot: n_pheno must be > 0");
    if (file_len < 256) return set_error(XSI_ERR_FORMAT, "file image shorter than the 256-byte header");
    HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    DecodePlan P;
    int rc = decode_prepare(ctx, d_file, file_len, first_block, n_blocks64, &P);
    if (rc) return rc;
    if (P.n_bin > capacity) return set_error(XSI_ERR_CAPACITY, "decode_dot: %u binary lines, capacity %llu", P.n_bin,
                                             (unsigned long long)capacity);
    // Exact for what the planes can say by themselves: bi-allelic, fully called lines.  A negated sparse
    // line lists its REF haplotypes, so with other ALT alleles or missing / end-of-vector entries on the
    // line the complement is not the carrier set; such blocks need the composed genotypes (decode_gt).
    for (auto& b : P.blocks_h)
        if (b.n_bin != b.n_bcf || b.off_line_missing != VAL_UNDEFINED || b.off_line_eov != VAL_UNDEFINED)
            return set_error(XSI_ERR_UNSUPPORTED, "decode_dot: multi-allelic lines or missing / end-of-vector entries in the "
                             "requested blocks; compose the genotypes with xsi_hip_decode_gt instead");
    // ALT-carrier bit planes of every binary line (negated sparse lines flipped back), then the product
    const uint32_t stride_w = P.L.y_stride64 * 2u;
    uint32_t* planes;
    WS(planes, "dot.planes", 4ull * stride_w * (size_t)(P.n_bin ? P.n_bin : 1));
    rc = decode_planes(ctx, d_file, P, planes, stride_w, /*apply_negation=*/1);
    if (rc) return rc;
    rc = dot_planes(ctx, P, planes, stride_w, d_pheno, n_pheno, d_out);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(s));
    stage_collect(ctx);
    if (h_n_bin) *h_n_bin = P.n_bin;
    return XSI_OK;
}

int xsi_hip_synth_packed(xsi_hip_ctx* ctx, uint64_t seed, uint64_t first_line, uint64_t n_lines, uint32_t n_haps,
                         void* d_bits, uint32_t row_stride_bytes) {
    if (!ctx || !d_bits) return set_error(XSI_ERR_ARG, "synth_packed: null argument");
    if (n_haps < 2 || row_stride_bytes % 8u || (uint64_t)row_stride_bytes * 8u < n_haps)
        return set_error(XSI_ERR_ARG, "synth_packed: bad n_haps / row_stride_bytes");
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(launch_synth_packed(ctx->stream, seed, first_line, n_lines, n_haps, reinterpret_cast<uint32_t*>(d_bits),
                                row_stride_bytes / 4u));
    return XSI_OK;
}

}  // extern "C"

namespace xsi {

// Parse header + block dictionaries, size and fill the per-line arrays.  Synchronises twice
// (header read-back, totals read-back): decode needs the line counts to size its workspace.
int decode_prepare(xsi_hip_ctx* ctx, const void* d_file, uint64_t file_len, uint64_t first_block, uint64_t n_blocks64,
                   DecodePlan* P, bool counts_only) {
    hipStream_t s = ctx->stream;
    uint8_t h[256];
    HIP_TRY(hipMemcpyAsync(h, d_file, 256, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    auto get = [&](size_t off, int bytes) {
        uint64_t v = 0;
        for (int i = 0; i < bytes; ++i) v |= (uint64_t)h[off + i] << (8 * i);
        return v;
    };
    if (get(0, 4) != 0xaabbccddu) return set_error(XSI_ERR_FORMAT, "Bad endianness");
    if (get(4, 4) != 0xfeed1767u || get(252, 4) != 0xfeed1767u) return set_error(XSI_ERR_FORMAT, "Bad magic");
    const uint32_t version = (uint32_t)get(8, 4);
    if (version != 4 && version != 5) return set_error(XSI_ERR_FORMAT, "Bad version");
    if (h[12] == 0) return set_error(XSI_ERR_FORMAT, "PLOIDY ERROR");
    if (h[17] & 4u) return set_error(XSI_ERR_UNSUPPORTED, "zstd-compressed blocks must be inflated on the host before decode");
    const uint32_t aet = h[14];
    if (aet != 2 && aet != 4) return set_error(XSI_ERR_FORMAT, "Unsupported A_T");
    const uint64_t hap_samples = get(32, 8), num_samples = get(112, 8);
    const uint64_t indices_offset = get(72, 8), samples_offset = get(80, 8);
    const uint64_t total_blocks = (samples_offset - indices_offset) / (version >= 5 ? 8 : 4);
    if (indices_offset > file_len || samples_offset > file_len || samples_offset < indices_offset)
        return set_error(XSI_ERR_FORMAT, "index outside the file image");
    if (first_block + n_blocks64 > total_blocks || n_blocks64 == 0 || n_blocks64 > 0xFFFFFFull)
        return set_error(XSI_ERR_ARG, "block range [%llu, +%llu) outside the %llu blocks of the file",
                         (unsigned long long)first_block, (unsigned long long)n_blocks64, (unsigned long long)total_blocks);
    const uint32_t n_blocks = (uint32_t)n_blocks64;
    // accessor_internals_new.hpp:53: N_HAPS = N_SAMPLES ? N_SAMPLES*2 : hap_samples
    const uint64_t N64 = num_samples ? num_samples * 2 : hap_samples;
    if (N64 < 2 || N64 > 0x7FFFFFFFull) return set_error(XSI_ERR_FORMAT, "haplotype count %llu out of range", (unsigned long long)N64);

    P->n_blocks = n_blocks;
    P->version = version;
    P->hap_samples = hap_samples;
    DecLines& L = P->L;
    memset(&L, 0, sizeof(L));
    L.N = (uint32_t)N64;
    L.n_samples = (uint32_t)(N64 / 2);
    L.aet = aet;
    L.file_len = file_len;
    WS(P->d_blocks, "dec.blocks", sizeof(DecBlock) * (size_t)n_blocks);
    WS(P->d_totals, "dec.totals", 64);
    stage_mark(ctx, XSI_ST_DEC_PARSE);
    HIP_TRY(launch_parse_blocks(s, (const uint8_t*)d_file, file_len, indices_offset, version, first_block, n_blocks,
                                P->d_blocks, P->d_totals));
    HIP_TRY(launch_scan_dec_blocks(s, P->d_blocks, n_blocks, P->d_totals));
    uint32_t tot[8];
    stage_mark(ctx, -1);
    HIP_TRY(hipMemcpyAsync(tot, P->d_totals, 32, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    stage_collect(ctx);
    if (tot[3]) return set_error(XSI_ERR_FORMAT, "corrupt block dictionary in the requested range");
    P->n_bin = tot[0];
    P->n_bcf = tot[4];
    const uint32_t n_bin = P->n_bin;
    L.max_bin = n_bin;
    WS(L.kind, "dec.kind", (size_t)n_bin + 64);
    WS(L.line_block, "dec.line_block", 4ull * n_bin + 64);
    WS(L.rank, "dec.rank", 4ull * n_bin + 64);
    WS(L.wah_start, "dec.wah_start", 4ull * n_bin + 64);
    WS(L.sparse_start, "dec.sparse_start", 4ull * n_bin + 64);
    WS(L.wah_lines, "dec.wah_lines", 4ull * n_bin + 64);
    WS(L.sparse_lines, "dec.sparse_lines", 4ull * n_bin + 64);
    WS(L.ones, "dec.ones", 4ull * n_bin + 64);
    WS(L.wah_cumg, "dec.wah_cumg", 4ull * n_bin + 64);
    stage_mark(ctx, XSI_ST_DEC_PARSE);
    HIP_TRY(launch_decode_flags(s, (const uint8_t*)d_file, P->d_blocks, n_blocks, L));
    HIP_TRY(launch_scan_dec_blocks2(s, P->d_blocks, n_blocks, P->d_totals));
    HIP_TRY(launch_dec_line_lists(s, P->d_blocks, n_blocks, L));
    // host copy of the block descriptors: side-channel presence decides the path
    P->blocks_h.resize(n_blocks);
    HIP_TRY(hipMemcpyAsync(P->blocks_h.data(), P->d_blocks, sizeof(DecBlock) * (size_t)n_blocks, hipMemcpyDeviceToHost, s));
    stage_mark(ctx, -1);
    HIP_TRY(hipMemcpyAsync(tot, P->d_totals, 32, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    stage_collect(ctx);
    P->n_wah = tot[1];
    P->n_sparse = tot[2];
    P->has_side = false;
    for (auto& b : P->blocks_h)
        if (b.off_line_missing != VAL_UNDEFINED || b.off_line_eov != VAL_UNDEFINED || b.off_line_phase != VAL_UNDEFINED ||
            b.off_line_haploid != VAL_UNDEFINED)
            P->has_side = true;
    L.y_stride64 = (L.N + 63u) / 64u;
    L.yp_stride = L.y_stride64 * 2u;
    if (counts_only) return XSI_OK;  // the caller only wants blocks_h: nothing sized by the WAH lines is allocated
    return decode_plan_scratch(ctx, P);
}

// The part of a plan that is scratch of the context (sized by the plan's WAH lines, overwritten by the next decode on the
// context): expanded rows, zeros per WAH line, the boundary scan's tiles - and, for a plan whose parsed part lives somewhere
// else (a cache entry of the accessor: the continuation of a prefix decode), the totals the kernels read.
int decode_plan_scratch(xsi_hip_ctx* ctx, DecodePlan* P, bool restore_totals) {
    DecLines& L = P->L;
    const uint32_t n_blocks = P->n_blocks;
    // shared with the encode's permuted rows; 16 KiB of slack: the long-row chain reads a row in whole 1024-unit pieces
    WS(L.yp, "ws.rows", 8ull * L.yp_stride * (size_t)(P->n_wah ? P->n_wah : 1) + 16384ull);
    WS(L.wah_z, "dec.wah_z", 4ull * (P->n_wah ? P->n_wah : 1) + 64);
    {
        // tiles of the boundary scan (2048 WAH words each)
        uint32_t max_words = 0;
        for (auto& b : P->blocks_h)
            if (!b.error && b.wah_words > max_words) max_words = b.wah_words;
        L.max_tiles = (max_words + WAH_BND_TILE_WORDS - 1u) / WAH_BND_TILE_WORDS;
        const size_t cells = (size_t)(L.max_tiles ? L.max_tiles : 1) * n_blocks;
        WS(L.tile_sum, "dec.tile_sum", 4ull * cells);
        WS(L.tile_base, "dec.tile_base", 8ull * cells);
    }
    if (restore_totals) {
        // [0] binary lines, [1] WAH lines, [2] sparse lines, [3] error, [4] BCF lines (k_scan_dec_blocks)
        WS(P->d_totals, "dec.totals", 64);
        uint32_t h[16] = {0};
        h[0] = P->n_bin;
        h[1] = P->n_wah;
        h[2] = P->n_sparse;
        h[4] = P->n_bcf;
        HIP_TRY(hipMemcpyAsync(P->d_totals, h, 64, hipMemcpyHostToDevice, ctx->stream));
        HIP_TRY(hipStreamSynchronize(ctx->stream));  // (h is a local)
    }
    return XSI_OK;
}

// Line boundaries + popcounts only: ones[] per binary line, no rows, no chain.
int decode_counts_only(xsi_hip_ctx* ctx, const void* d_file, DecodePlan& P) {
    hipStream_t s = ctx->stream;
    const uint8_t* f = (const uint8_t*)d_file;
    stage_mark(ctx, XSI_ST_DEC_BOUND);
    HIP_TRY(launch_wah_boundaries(s, f, P.d_blocks, P.n_blocks, P.L));
    HIP_TRY(launch_sparse_walk(s, f, P.d_blocks, P.n_blocks, P.L));
    stage_mark(ctx, XSI_ST_DEC_EXPAND);
    HIP_TRY(launch_line_counts(s, f, P.d_blocks, P.L, P.n_wah, P.n_sparse, P.d_totals));
    stage_mark(ctx, -1);
    return XSI_OK;
}

// The WAH lines [lo[b], hi[b]) of every block b (block-relative ranks; nullptr: all of them) in K ranges: the expansion
// of range p + 1 (side stream) runs underneath the chain of range p, which parks its ranks in `d_state` between the
// launches - and behind the last one, when hi[b] is not the block's last WAH line, so that a later call continues
// from there (the accessor's prefix decode).  The boundaries must have been computed on the context's stream.
// `split_boundaries`: only the starts of every block's first range have been computed (launch_wah_boundaries_part 1);
// the rest of the boundary scan goes to the side stream behind the first range's expansion, underneath the first
// chain launch (whole-block decodes only: lo == hi == nullptr).
static int run_wah_phases(xsi_hip_ctx* ctx, const uint8_t* f, DecodePlan& P, uint32_t* out, uint32_t stride_w,
                          const uint32_t* lo, const uint32_t* hi, uint32_t K, uint32_t* d_state, bool split_boundaries = false) {
    hipStream_t s = ctx->stream;
    DecLines& L = P.L;
    const uint32_t nb = P.n_blocks, lpg = wah_expand_lines_per_group(L);
    // Range p of a block with n lines = lines [n cuts[p] / den, n cuts[p + 1] / den).  K equal ranges - or, with the
    // split boundary scan, a ramp in front of them: 1/4, 1/2 and one whole range's worth of lines, so that what shows in
    // front of the first chain launch (the scan's first part and the first expansion) covers n / 4K lines, not n / K
    // (configs[2]: 64.7 -> 64.4 ms per step).
    std::vector<uint32_t> cuts;
    uint32_t den = K;
    if (split_boundaries && K >= 2u) {
        den = 4u * K;
        cuts = {0u, 1u, 3u};
        for (uint32_t c = 7u; c < den; c += 4u) cuts.push_back(c);
        cuts.push_back(den);
    } else {
        for (uint32_t p = 0; p <= K; ++p) cuts.push_back(p);
    }
    K = (uint32_t)cuts.size() - 1u;
    // per phase and block: first WAH line (batch-wide rank), lines, and the running number of 4-line groups.
    // (Launch p takes range p of EVERY block.  Round 5 tried launches that take whole rounds' worth of blocks only - a
    // rotating window over a 57-block batch on 32 slots, 43 full launches instead of 24 of 1.78 rounds: 162 against 157 ms
    // for the decode of the configs[3] shard.  The partly filled round is where the expansion of the next range runs.)
    P.phase_tab.assign((size_t)K * (3u * nb + 1u), 0u);
    for (uint32_t p = 0; p < K; ++p) {
        uint32_t* start = P.phase_tab.data() + (size_t)p * (3u * nb + 1u);
        uint32_t *cnt = start + nb, *gpre = cnt + nb;
        uint32_t g = 0;
        for (uint32_t b = 0; b < nb; ++b) {
            const DecBlock& D = P.blocks_h[b];
            const uint32_t nw = D.error ? 0u : D.n_wah;
            const uint32_t b_lo = lo ? (lo[b] < nw ? lo[b] : nw) : 0u, b_hi = hi ? (hi[b] < nw ? hi[b] : nw) : nw;
            const uint32_t n = b_hi > b_lo ? b_hi - b_lo : 0u;
            const uint32_t plo = (uint32_t)((uint64_t)n * cuts[p] / den), phi = (uint32_t)((uint64_t)n * cuts[p + 1u] / den);
            start[b] = D.wah_first + b_lo + plo;
            cnt[b] = phi - plo;
            gpre[b] = g;
            g += (phi - plo + lpg - 1u) / lpg;
        }
        gpre[nb] = g;
    }
    uint32_t* d_tab;
    WS(d_tab, "dec.phase_tab", 4ull * P.phase_tab.size());
    HIP_TRY(hipMemcpyAsync(d_tab, P.phase_tab.data(), 4ull * P.phase_tab.size(), hipMemcpyHostToDevice, s));
    if (!ctx->side2) {
        int least = 0, greatest = 0;
        if (ctx->low_priority) HIP_TRY(hipDeviceGetStreamPriorityRange(&least, &greatest));
        HIP_TRY(ctx->low_priority ? hipStreamCreateWithPriority(&ctx->side2, hipStreamNonBlocking, least)
                                  : hipStreamCreateWithFlags(&ctx->side2, hipStreamNonBlocking));
    }
    while (ctx->ev_phase.size() < (size_t)K + 1u) {
        hipEvent_t e;
        HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        ctx->ev_phase.push_back(e);
    }
    HIP_TRY(hipEventRecord(ctx->ev_phase[K], s));  // boundaries done, table uploaded
    HIP_TRY(hipStreamWaitEvent(ctx->side2, ctx->ev_phase[K], 0));
    for (uint32_t p = 0; p < K; ++p) {
        const uint32_t* tab = d_tab + (size_t)p * (3u * nb + 1u);
        const uint32_t groups = P.phase_tab[(size_t)p * (3u * nb + 1u) + 3u * nb];
        HIP_TRY(launch_wah_expand_phase(ctx->side2, f, P.d_blocks, L, P.d_totals, tab, tab + nb, tab + 2u * nb, nb, groups));
        HIP_TRY(hipEventRecord(ctx->ev_phase[p], ctx->side2));
        // (the scan's first part covered the ramp: see decode_planes)
        if (split_boundaries && p == (K > 3u ? 2u : 0u)) HIP_TRY(launch_wah_boundaries_part(ctx->side2, f, P.d_blocks, nb, L, den, cuts[p + 1u], 2));
    }
    stage_mark(ctx, XSI_ST_DEC_EXPAND);  // what shows of the expansion: the wait for its first range
    // (measurement: XSI_DEC_PHASES_SERIAL=1 lets every range expand before the first chain launch, which leaves
    // the chain's launches by themselves: their time minus the unphased chain's is the cost of cutting it up)
    HIP_TRY(hipStreamWaitEvent(s, ctx->ev_phase[tuning_env("XSI_DEC_PHASES_SERIAL") ? K - 1u : 0u], 0));
    stage_mark(ctx, XSI_ST_CHAIN_DEC);
    for (uint32_t p = 0; p < K; ++p) {
        const uint32_t* tab = d_tab + (size_t)p * (3u * nb + 1u);
        if (p) HIP_TRY(hipStreamWaitEvent(s, ctx->ev_phase[p], 0));
        HIP_TRY(launch_rank_decode_phase(s, P.d_blocks, nb, L, out, stride_w, tab, tab + nb, d_state));
    }
    return XSI_OK;
}

bool decode_partial_supported(const DecodePlan& P) {
    if (P.n_blocks != 1u) return false;
    for (auto& b : P.blocks_h) {
        if (b.error || b.off_line_haploid != VAL_UNDEFINED) return false;
        // version-4 weirdness lines are permuted by a chain of their own that is replayed whole (k_dec_side_unpermute)
        if ((b.off_line_missing != VAL_UNDEFINED || b.off_line_eov != VAL_UNDEFINED) && b.strategy == WS_PBWT_WAH) return false;
    }
    return rank_decode_phased_ok(P.L.N, P.L.yp_stride, 1u);
}

// One block, part of its WAH lines: [wah_lo, wah_hi) in block-relative rank order, continuing the chain from the ranks
// parked in d_state (4 * rank_decode_state_words(N, 1) bytes, owned by the caller) when wah_lo > 0; and its sparse lines
// of rank [sp_lo, sp_hi) (they do not depend on the chain, but their lists are a pointer chase whose cursor is carried
// in d_sp_state[0]).  The accessor's prefix decode: a cold query at line o of
// a block replays the o lines in front of it like the reference's seek (accessor_internals_new.hpp:154-196), not all
// 8192, and a later query further into the block continues where this one stopped.
int decode_planes_partial(xsi_hip_ctx* ctx, const void* d_file, DecodePlan& P, uint32_t* out, uint32_t stride_w,
                          uint32_t wah_lo, uint32_t wah_hi, uint32_t* d_state, uint32_t sp_lo, uint32_t sp_hi, uint64_t* d_sp_state,
                          bool skip_boundaries) {
    hipStream_t s = ctx->stream;
    const uint8_t* f = (const uint8_t*)d_file;
    DecLines& L = P.L;
    if (!decode_partial_supported(P)) return set_error(XSI_ERR_UNSUPPORTED, "decode_planes_partial: not a block the ranged chain takes");
    const bool sparse = sp_hi > sp_lo;
    if (sparse) {  // the sparse lines in front of the range's end: their lists are a pointer chase of their own
        L.sp_lo = sp_lo;
        L.sp_hi = sp_hi < P.n_sparse ? sp_hi : P.n_sparse;
        L.sp_state = d_sp_state;
        HIP_TRY(hipEventRecord(ctx->ev_fork, s));
        HIP_TRY(hipStreamWaitEvent(ctx->side, ctx->ev_fork, 0));
        HIP_TRY(launch_sparse_walk(ctx->side, f, P.d_blocks, P.n_blocks, L));
        if (L.sp_hi > L.sp_lo)
            HIP_TRY(launch_sparse_fill(ctx->side, f, P.d_blocks, L, L.sp_hi - L.sp_lo, P.d_totals, out, stride_w, /*apply_negation=*/0));
        HIP_TRY(hipEventRecord(ctx->ev_join, ctx->side));
    }
    L.yp_rows = P.n_wah ? P.n_wah : 1u;
    L.yp_compact = rank_decode_takes_compact(L.N, L.yp_stride, P.n_blocks) ? 1u : 0u;
    L.yp_rev = (wah_expand_wide(L) && rank_decode_takes_reversed(L.N, L.yp_stride, P.n_blocks)) ? 1u : 0u;
    stage_mark(ctx, XSI_ST_DEC_BOUND);
    // (a continuation whose plan kept the line starts of the first decode does not scan the block's WAH words again)
    if (!skip_boundaries) HIP_TRY(launch_wah_boundaries(s, f, P.d_blocks, P.n_blocks, L));
    if (wah_hi > wah_lo) {
        const uint32_t n = wah_hi - wah_lo;
        const uint32_t K = n >= 2048u ? 8u : (n >= 512u ? 4u : (n >= 128u ? 2u : 1u));
        int rc = run_wah_phases(ctx, f, P, out, stride_w, &wah_lo, &wah_hi, K, d_state);
        if (rc) return rc;
    }
    if (sparse) {
        stage_mark(ctx, XSI_ST_DEC_SPARSE);
        HIP_TRY(hipStreamWaitEvent(s, ctx->ev_join, 0));
    }
    stage_mark(ctx, -1);
    return XSI_OK;
}

// WAH boundaries -> expand -> chain; sparse walk -> fill.  Output: one natural-order bit row
// per binary line at out + l*stride_w.
int decode_planes(xsi_hip_ctx* ctx, const void* d_file, DecodePlan& P, uint32_t* out, uint32_t stride_w, int apply_negation) {
    hipStream_t s = ctx->stream;
    const uint8_t* f = (const uint8_t*)d_file;
    DecLines& L = P.L;
    uint32_t* scratch_a = nullptr;
    bool any_haploid = false;
    for (auto& b : P.blocks_h)
        if (b.off_line_haploid != VAL_UNDEFINED) any_haploid = true;
    if (any_haploid && !chain_geometry(L.N, true).in_lds)
        WS(scratch_a, "chain.a", 4ull * 2ull * (((size_t)L.N + 63u) & ~(size_t)63u) * P.n_blocks);
    // Sparse lines never touch the PBWT order and write their own output rows: their pointer walk
    // and fill run on the side stream, underneath the WAH boundary scan / expansion / chain (which
    // leave most wave slots of every CU free), and are joined back before the call returns.
    HIP_TRY(hipEventRecord(ctx->ev_fork, s));
    HIP_TRY(hipStreamWaitEvent(ctx->side, ctx->ev_fork, 0));
    HIP_TRY(launch_sparse_walk(ctx->side, f, P.d_blocks, P.n_blocks, L));
    HIP_TRY(launch_sparse_fill(ctx->side, f, P.d_blocks, L, P.n_sparse, P.d_totals, out, stride_w, apply_negation));
    HIP_TRY(hipEventRecord(ctx->ev_join, ctx->side));
    // rows for the one-workgroup-per-block chain in the compact form (10 instead of 16 bytes per 64 positions)
    L.yp_rows = P.n_wah ? P.n_wah : 1u;
    L.yp_compact = (!any_haploid && rank_decode_takes_compact(L.N, L.yp_stride, P.n_blocks)) ? 1u : 0u;
    L.yp_rev = (!any_haploid && wah_expand_wide(L) && rank_decode_takes_reversed(L.N, L.yp_stride, P.n_blocks)) ? 1u : 0u;
    stage_mark(ctx, XSI_ST_DEC_BOUND);
    // Phased: the chain needs line j of every block at its step j, so the WAH lines of every block are cut into K
    // ranges; the expansion of range p+1 (side stream) runs underneath the chain of range p, which parks its
    // ranks in HBM between the launches (4 N bytes per block: 64 MB at 245 blocks of 64 976 haplotypes).  A whole second expansion
    // next to the chain costs the chain 1 ms of its 23.4 (measured): the expansion is almost free this way.
    // Cutting the chain into launches costs nothing (24 launches behind a finished expansion: 23.5 ms, as one launch);
    // the expansion beside it costs the chain 3.5 ms and hides 5.2 of its own 5.9.
    const uint32_t n_phases = [] {  // read per call (tests switch it); ms per step at configs[2]: 1: 65.5, 12: 64.1, 24: 63.8, 32: 63.75
        const char* e = tuning_env("XSI_DEC_PHASES");
        const int v = e ? atoi(e) : 24;
        return (uint32_t)(v < 1 ? 1 : (v > 32 ? 32 : v));
    }();
    if (n_phases > 1u && !any_haploid && P.n_wah >= 256u * P.n_blocks && rank_decode_phased_ok(L.N, L.yp_stride, P.n_blocks)) {
        // The boundary scan is phased too: only the first range's line starts are found in front of the chain, the rest
        // of the scan runs underneath the first chain launch: configs[2] 65.6 -> 64.7 ms per step (boundaries 1.16 ->
        // 0.51 ms visible, the first range's expansion 0.68 -> 0.48).  Short rows only: a long-row chain fills the CUs
        // (section 5.6 of DESIGN.md), the second part just takes its turn there (configs[3] shard 493.2 against 494.0 ms).
        // (XSI_DEC_BOUNDARIES_WHOLE=1: all of it in front, as before - A/B runs)
        const bool split = tuning_env("XSI_DEC_BOUNDARIES_WHOLE") == nullptr && L.y_stride64 * 8u <= 16384u;
        if (split)
            HIP_TRY(launch_wah_boundaries_part(s, f, P.d_blocks, P.n_blocks, L, 4u * n_phases, 7u, 1));  // run_wah_phases' first three ranges
        else
            HIP_TRY(launch_wah_boundaries(s, f, P.d_blocks, P.n_blocks, L));
        uint32_t* d_state;
        WS(d_state, "dec.rank_state", 4ull * rank_decode_state_words(L.N, P.n_blocks));
        int prc = run_wah_phases(ctx, f, P, out, stride_w, nullptr, nullptr, n_phases, d_state, split);
        if (prc) return prc;
    } else {
        HIP_TRY(launch_wah_boundaries(s, f, P.d_blocks, P.n_blocks, L));
        stage_mark(ctx, XSI_ST_DEC_EXPAND);
        HIP_TRY(launch_wah_expand(s, f, P.d_blocks, L, P.n_wah, P.d_totals));
        stage_mark(ctx, XSI_ST_CHAIN_DEC);
        HIP_TRY(launch_chain_decode(s, P.d_blocks, P.n_blocks, L, out, stride_w, scratch_a, any_haploid));
    }
    stage_mark(ctx, XSI_ST_DEC_SPARSE);  // what is left of the sparse work after the chain
    HIP_TRY(hipStreamWaitEvent(s, ctx->ev_join, 0));
    stage_mark(ctx, -1);
    return XSI_OK;
}

}  // namespace xsi
