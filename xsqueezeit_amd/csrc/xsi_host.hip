// xsi_host.hip — file-level writer and accessor: the host-side mirror of the reference's
// XsiFactoryExt (include/xsi_factory.hpp:435-639) and Accessor (include/accessor.hpp:31-124,
// accessor.cpp:26-88) on top of the GPU block codec.
//
// The reference encodes and decodes one line per call on the CPU.  A GPU cannot work a line at
// a time, so both classes batch per block behind the same per-line interface:
//   writer  : append() stages int32 rows, ships them to HBM in chunks, and when block_len lines
//             are in, encodes the whole block on the GPU and appends the bytes to the file;
//   accessor: the first touch of a block decodes ALL its binary lines to bit planes on the GPU;
//             int32 rows are then composed on the GPU per window of lines (bi-allelic blocks) or
//             per requested line (blocks with multi-allelic lines, whose grouping the accessor only
//             learns from each call's n_alleles, accessor.hpp:48-50) and served from pinned memory.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <list>
#include <mutex>
#include <thread>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/xsi_hip.h"
#include "xsi_ctx.hpp"
#include "xsi_device.hpp"

using namespace xsi;

#define HIP_TRY(expr)                                                                                        \
    do {                                                                                                     \
        hipError_t _e = (expr);                                                                              \
        if (_e != hipSuccess) return set_error(XSI_ERR_HIP, "%s: %s (%s:%d)", #expr, hipGetErrorString(_e), \
                                               __FILE__, __LINE__);                                          \
    } while (0)

// ------------------------------------------------------------------------------------------
// optional zstd layer (BlockWithZstdCompressor, interfaces.hpp:288-315; set_block_ptr,
// accessor_internals_new.hpp:857-886).  Host-side, like the reference.  libzstd is bound at run
// time (no zstd headers in the build image); without it --zstd files are refused.
// ------------------------------------------------------------------------------------------
#include <dlfcn.h>
#include <condition_variable>
#include <deque>
#include <exception>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>
namespace {
struct ZstdApi {
    size_t (*compress)(void*, size_t, const void*, size_t, int) = nullptr;
    size_t (*decompress)(void*, size_t, const void*, size_t) = nullptr;
    // (optional) an explicit compression context, reused by a pool thread: ZSTD_compressCCtx is ZSTD_compress with the
    // context handed in - the same frame bytes - without the allocation and release of its tables on every block
    void* (*create_cctx)() = nullptr;
    size_t (*free_cctx)(void*) = nullptr;
    size_t (*compress_cctx)(void*, void*, size_t, const void*, size_t, int) = nullptr;
    unsigned (*is_error)(size_t) = nullptr;
    const char* (*error_name)(size_t) = nullptr;
    bool ok = false;
};
const ZstdApi& zstd_api() {
    static ZstdApi z = [] {
        ZstdApi a;
        void* h = dlopen("libzstd.so.1", RTLD_NOW | RTLD_LOCAL);
        if (!h) h = dlopen("libzstd.so", RTLD_NOW | RTLD_LOCAL);
        if (!h) return a;
        a.compress = reinterpret_cast<decltype(a.compress)>(dlsym(h, "ZSTD_compress"));
        a.decompress = reinterpret_cast<decltype(a.decompress)>(dlsym(h, "ZSTD_decompress"));
        a.is_error = reinterpret_cast<decltype(a.is_error)>(dlsym(h, "ZSTD_isError"));
        a.error_name = reinterpret_cast<decltype(a.error_name)>(dlsym(h, "ZSTD_getErrorName"));
        a.create_cctx = reinterpret_cast<decltype(a.create_cctx)>(dlsym(h, "ZSTD_createCCtx"));
        a.free_cctx = reinterpret_cast<decltype(a.free_cctx)>(dlsym(h, "ZSTD_freeCCtx"));
        a.compress_cctx = reinterpret_cast<decltype(a.compress_cctx)>(dlsym(h, "ZSTD_compressCCtx"));
        if (!a.create_cctx || !a.free_cctx || !a.compress_cctx) a.create_cctx = nullptr;
        a.ok = a.compress && a.decompress && a.is_error;
        return a;
    }();
    return z;
}

// The zstd layer at block-parallel speed.  ZSTD_compress is one-shot per block and deterministic, every call
// with a context of its own: blocks are compressed side by side on a small pool while the GPU encodes the next batch and the
// caller fills the one after, and the frames reach the file in block order - byte for byte the file of the one-thread
// writer (and of the reference, for the libzstd in use).
struct ZstdJob {
    std::vector<uint8_t> raw;    // the block as streamed, before its pad
    std::vector<uint8_t> frame;  // compressed
    size_t csize = 0;
    uint64_t usize = 0;          // raw.size() (raw is released once compressed)
    bool done = false, failed = false;
    std::string err;
};
struct ZstdPool {
    std::vector<std::thread> threads;
    std::mutex m;
    std::condition_variable cv_work, cv_done;
    std::deque<std::shared_ptr<ZstdJob>> todo;   // not yet taken by a thread
    std::deque<std::shared_ptr<ZstdJob>> order;  // in file order, not yet written
    bool stop = false;
    int level = 0;
    bool prof = false;
    // Buffers go round: a block's bytes and its frame are megabytes each, and a fresh vector per job is an mmap, a page
    // fault per 4 KiB and a munmap - with its TLB shoot-down to every core the process runs on - sixteen threads at a time
    // (measured: a 5 MB block took 75 ms alone and 190 - 300 ms next to seven others).
    std::vector<std::vector<uint8_t>> spare;
    std::vector<uint8_t> take_buf() {
        std::lock_guard<std::mutex> lk(m);
        if (spare.empty()) return {};
        std::vector<uint8_t> v = std::move(spare.back());
        spare.pop_back();
        return v;
    }
    void give_buf(std::vector<uint8_t>&& v) {
        if (!v.capacity()) return;
        std::lock_guard<std::mutex> lk(m);
        if (spare.size() < 128u) spare.push_back(std::move(v));
    }
    void run() {
        const ZstdApi& z = zstd_api();
        std::vector<uint8_t> scratch;
        void* cctx = z.create_cctx ? z.create_cctx() : nullptr;
        struct Free {
            const ZstdApi& z;
            void*& c;
            ~Free() {
                if (c) z.free_cctx(c);
            }
        } free_cctx{z, cctx};
        for (;;) {
            std::shared_ptr<ZstdJob> j;
            {
                std::unique_lock<std::mutex> lk(m);
                cv_work.wait(lk, [&] { return stop || !todo.empty(); });
                if (todo.empty()) return;  // (stop, nothing left)
                j = todo.front();
                todo.pop_front();
            }
            bool failed = false;
            std::string err;
            size_t cs = 0;
            const auto t0 = std::chrono::steady_clock::now();
            try {
                // the destination is this thread's own, kept across jobs (a fresh 2 x block-size vector per job was a page
                // fault per 4 KiB in sixteen threads at once); only the frame's bytes are copied into the job
                const size_t bound = j->raw.size() + j->raw.size() / 128u + 1024u;  // (above ZSTD_compressBound: size + size / 256 + 64 KiB-block margins)
                if (scratch.size() < bound) scratch.resize(bound + bound / 4u);
                cs = cctx ? z.compress_cctx(cctx, scratch.data(), scratch.size(), j->raw.data(), j->raw.size(), level)
                          : z.compress(scratch.data(), scratch.size(), j->raw.data(), j->raw.size(), level);
                if (z.is_error(cs)) {
                    failed = true;
                    err = z.error_name ? z.error_name(cs) : "zstd";
                } else {
                    j->frame = take_buf();
                    j->frame.assign(scratch.data(), scratch.data() + cs);
                }
                give_buf(std::move(j->raw));
                j->raw = std::vector<uint8_t>();
            } catch (const std::exception& e) {
                failed = true;
                err = e.what();
            }
            if (prof)
                fprintf(stderr, "[xsi writer prof] zstd job: %llu -> %zu bytes in %.1f ms\n", (unsigned long long)j->usize, cs,
                        std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
            {
                std::lock_guard<std::mutex> lk(m);
                j->csize = cs;
                j->failed = failed;
                j->err = err;
                j->done = true;
            }
            cv_done.notify_all();
        }
    }
    void start(unsigned n, int lvl) {
        level = lvl;
        for (unsigned i = 0; i < n; ++i) threads.emplace_back([this] { run(); });
    }
    void submit(std::shared_ptr<ZstdJob> j) {
        {
            std::lock_guard<std::mutex> lk(m);
            todo.push_back(j);
            order.push_back(j);
        }
        cv_work.notify_one();
    }
    // the oldest job once it is done (waits for it when `wait`), removed from the order; nullptr: none / not yet
    std::shared_ptr<ZstdJob> pop_done(bool wait) {
        std::unique_lock<std::mutex> lk(m);
        if (order.empty()) return nullptr;
        if (wait) cv_done.wait(lk, [&] { return order.front()->done; });
        if (!order.front()->done) return nullptr;
        std::shared_ptr<ZstdJob> j = order.front();
        order.pop_front();
        return j;
    }
    size_t pending() {
        std::lock_guard<std::mutex> lk(m);
        return order.size();
    }
    ~ZstdPool() {
        {
            std::lock_guard<std::mutex> lk(m);
            stop = true;
            todo.clear();
        }
        cv_work.notify_all();
        for (auto& t : threads) t.join();
    }
};
}  // namespace

// ------------------------------------------------------------------------------------------
// writer
//
// XsiFactoryExt::append flushes a block every block_len BCF lines (xsi_factory.hpp:527-539) and
// encodes one line per call on the CPU.  Here lines are staged in pinned chunks, shipped to one of
// two device batches on a copy stream, and a batch of up to K whole blocks is encoded by ONE
// xsi_hip_encode_gt call on a worker thread (so the PBWT chain runs one workgroup per block of the
// batch, not one workgroup in total) while append() already fills the other batch.  The bytes
// written are those of block-by-block flushing: blocks are independent.
// ------------------------------------------------------------------------------------------
#include <thread>

struct xsi_writer {
    xsi_hip_ctx* ctx = nullptr;
    xsi_encode_params p{};
    FILE* f = nullptr;
    std::vector<std::string> names;
    uint64_t N = 0;
    // two device batches of up to batch_blocks blocks of int32 rows
    uint32_t batch_blocks = 1;
    // blocks the batch being filled may take: batch_blocks, except that under --zstd the first batches are short (2, 4, 8, ...)
    // so that the compression pool - whose last block ends the file's tail - gets its first blocks early
    uint32_t batch_limit = 1;
    int32_t* d_rows[2] = {nullptr, nullptr};
    std::vector<uint32_t> ngt[2], n_allele[2];
    int cur = 0;                    // batch being filled
    uint64_t lines_in_batch = 0, lines_on_device = 0;
    hipStream_t copy_stream = nullptr;
    hipEvent_t batch_copied = nullptr;
    // Bit rows: a bi-allelic, fully called diploid line whose second values carry the default phase is one bit per
    // haplotype (what xsi_hip_encode_packed takes).  append packs such a line in the caller's thread (reads 4 N
    // bytes, writes N / 8 - cheaper than copying the int32 row) and only the bit row crosses PCIe; any other line
    // is staged as int32.  A batch without such "general" lines is encoded by xsi_hip_encode_packed; otherwise
    // the packed lines are expanded to int32 rows on the device and the batch goes through xsi_hip_encode_gt.
    // Same bytes either way (both entry points are byte-equal to the reference on such lines).
    uint32_t bit_stride = 0;        // bytes of a bit row (a multiple of 128)
    uint8_t* d_bits[2] = {nullptr, nullptr};
    uint8_t* d_fast[2] = {nullptr, nullptr};  // per line: 1 = bit row, 0 = int32 row
    std::vector<uint8_t> fast[2];
    std::vector<uint32_t> ones[2];  // set bits of every bit row, counted when it is packed (xsi_hip_encode_packed_counted)
    uint32_t* d_ones[2] = {nullptr, nullptr};
    uint64_t general_in_batch = 0;
    uint8_t* h_bits_chunks[2] = {nullptr, nullptr};
    uint8_t* h_bits_chunk = nullptr;
    uint32_t chunk_general = 0;     // int32 rows in the chunk being filled (0: its int32 half is not shipped)
    // pinned staging chunks: one fills while the other is on its way to HBM
    int32_t* h_chunk = nullptr;
    int32_t* h_chunks[2] = {nullptr, nullptr};
    hipEvent_t chunk_done[2] = {nullptr, nullptr};
    int cur_chunk = 0;
    uint32_t chunk_rows = 0, chunk_fill = 0;
    // worker that encodes and writes the batch that is not being filled
    std::thread worker;
    bool worker_active = false;
    int worker_rc = XSI_OK;
    std::string worker_err;
    uint8_t* d_out = nullptr;
    uint64_t out_cap = 0;
    uint64_t* d_offs = nullptr;
    uint64_t offs_cap = 0;
    std::vector<uint8_t> h_out;
    std::vector<uint64_t> h_offs;
    std::vector<uint64_t> indices;
    uint64_t entry_counter = 0, variant_counter = 0;
    uint32_t max_ploidy_seen = 0;
    uint64_t file_pos = 0;
    // --zstd: blocks of a batch are compressed on a pool and written in order (ZstdPool above); d_sizes takes every block's
    // length before its pad (xsi_hip_ctx_set_block_sizes_out)
    ZstdPool* zpool = nullptr;
    uint32_t* d_sizes = nullptr;
    std::vector<uint32_t> h_sizes;
    size_t zstd_max_pending = 0;
};


static int writer_ship_chunk(xsi_writer* w) {
    if (!w->chunk_fill) return XSI_OK;
    // the bit rows always go (N / 8 bytes a line); the int32 half only when the chunk holds a line that needs it
    HIP_TRY(hipMemcpyAsync(w->d_bits[w->cur] + (size_t)w->lines_on_device * w->bit_stride, w->h_bits_chunk,
                           (size_t)w->chunk_fill * w->bit_stride, hipMemcpyHostToDevice, w->copy_stream));
    if (w->chunk_general)
        HIP_TRY(hipMemcpyAsync(w->d_rows[w->cur] + (size_t)w->lines_on_device * w->N, w->h_chunk,
                               (size_t)w->chunk_fill * w->N * sizeof(int32_t), hipMemcpyHostToDevice, w->copy_stream));
    HIP_TRY(hipEventRecord(w->chunk_done[w->cur_chunk], w->copy_stream));
    w->lines_on_device += w->chunk_fill;
    w->chunk_fill = 0;
    w->chunk_general = 0;
    // keep filling the other chunk while this one is copied; wait only if that one is still in flight
    w->cur_chunk ^= 1;
    w->h_chunk = w->h_chunks[w->cur_chunk];
    w->h_bits_chunk = w->h_bits_chunks[w->cur_chunk];
    HIP_TRY(hipEventSynchronize(w->chunk_done[w->cur_chunk]));
    return XSI_OK;
}

// Frames of finished blocks -> file, in block order.  all: every block submitted so far (finalize); else only while the
// queue is longer than what keeps the pool busy (or the front happens to be ready).
static int writer_drain_zstd(xsi_writer* w, bool all) {
    if (!w->zpool) return XSI_OK;
    for (;;) {
        const bool must = all || w->zpool->pending() > w->zstd_max_pending;
        std::shared_ptr<ZstdJob> j = w->zpool->pop_done(must);
        if (!j) return XSI_OK;
        if (j->failed) return set_error(XSI_ERR_IO, "Failed to compress block: %s", j->err.c_str());
        w->indices.push_back(w->file_pos);
        const uint64_t c64 = j->csize, u64 = j->usize;
        if (fwrite(&c64, 8, 1, w->f) != 1 || fwrite(&u64, 8, 1, w->f) != 1 || fwrite(j->frame.data(), 1, j->csize, w->f) != j->csize)
            return set_error(XSI_ERR_IO, "short write");
        w->file_pos += 16 + j->csize;
        while (w->file_pos % 4) {
            if (fputc(0, w->f) == EOF) return set_error(XSI_ERR_IO, "short write");
            w->file_pos++;
        }
        w->zpool->give_buf(std::move(j->frame));
    }
}

// Worker: encode batch `b` (its rows are on the device once batch_copied has fired) and append the
// blocks to the file.  Runs on its own thread; errors are parked in the writer.
static int writer_encode_batch(xsi_writer* w, int b, uint64_t n_lines) {
    HIP_TRY(hipSetDevice(w->ctx->device));
    HIP_TRY(hipStreamWaitEvent(w->ctx->stream, w->batch_copied, 0));
    uint64_t n_bin = 0;
    for (uint32_t a : w->n_allele[b]) n_bin += a - 1;
    const uint64_t n_blocks = (n_lines + w->p.block_len - 1) / w->p.block_len;
    const uint64_t need = xsi_hip_encode_gt_bound(&w->p, n_lines, n_bin);
    if (need > w->out_cap) {
        if (w->d_out) (void)hipFree(w->d_out);
        w->d_out = nullptr;
        HIP_TRY(hipMalloc((void**)&w->d_out, need));
        w->out_cap = need;
    }
    if (n_blocks > w->offs_cap) {
        if (w->d_offs) (void)hipFree(w->d_offs);
        w->d_offs = nullptr;
        HIP_TRY(hipMalloc((void**)&w->d_offs, 8ull * n_blocks));
        w->offs_cap = n_blocks;
    }
    xsi_encode_result res{};
    int rc;
    uint64_t n_general = 0;
    for (uint8_t f : w->fast[b]) n_general += f ? 0u : 1u;
    // (the context is the caller's: the side output is on for this call only)
    struct SizesOut {
        xsi_hip_ctx* c;
        SizesOut(xsi_hip_ctx* ctx, uint32_t* d, uint64_t cap) : c(d ? ctx : nullptr) {
            if (c) (void)xsi_hip_ctx_set_block_sizes_out(c, d, cap);
        }
        ~SizesOut() {
            if (c) (void)xsi_hip_ctx_set_block_sizes_out(c, nullptr, 0);
        }
    } sizes_out(w->ctx, w->p.zstd_level ? w->d_sizes : nullptr, w->batch_blocks);
    if (n_general == 0) {
        HIP_TRY(hipMemcpyAsync(w->d_ones[b], w->ones[b].data(), 4ull * n_lines, hipMemcpyHostToDevice, w->ctx->stream));
        rc = xsi_hip_encode_packed_counted(w->ctx, &w->p, w->d_bits[b], n_lines, w->bit_stride, w->d_ones[b], w->d_out, w->out_cap,
                                           w->d_offs, &res);
    } else {
        if (n_general != n_lines) {  // the packed lines become int32 rows on the device, next to the shipped ones
            HIP_TRY(hipMemcpyAsync(w->d_fast[b], w->fast[b].data(), n_lines, hipMemcpyHostToDevice, w->ctx->stream));
            rc = xsi::expand_bit_rows(w->ctx, w->d_bits[b], w->bit_stride, w->d_fast[b], w->d_rows[b], w->N, n_lines, w->p.default_phased);
            if (rc) return rc;
        }
        rc = xsi_hip_encode_gt(w->ctx, &w->p, w->d_rows[b], w->N, n_lines, w->ngt[b].data(), w->n_allele[b].data(), w->d_out,
                               w->out_cap, w->d_offs, &res);
    }
    if (rc) return rc;
    if (w->zpool && w->zpool->prof) {
        (void)hipStreamSynchronize(w->ctx->stream);
        fprintf(stderr, "[xsi writer prof] batch of %llu blocks encoded (%llu bytes), %zu frames pending\n", (unsigned long long)n_blocks,
                (unsigned long long)res.blocks_bytes, w->zpool->pending());
    }
    w->h_out.resize(res.blocks_bytes);
    w->h_offs.resize(n_blocks + 1);
    HIP_TRY(hipMemcpy(w->h_out.data(), w->d_out, res.blocks_bytes, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(w->h_offs.data(), w->d_offs, 8ull * n_blocks, hipMemcpyDeviceToHost));
    w->h_offs[n_blocks] = 256 + res.blocks_bytes;
    if (!w->p.zstd_level) {
        for (uint64_t k = 0; k < n_blocks; ++k) w->indices.push_back(w->file_pos + (w->h_offs[k] - 256));  // xsi_factory.hpp:533
        if (fwrite(w->h_out.data(), 1, w->h_out.size(), w->f) != w->h_out.size()) return set_error(XSI_ERR_IO, "short write");
        w->file_pos += w->h_out.size();
        return XSI_OK;
    }
    // compress_and_write, interfaces.hpp:291-314: u64 compressed size, u64 original size, frame; pad to 4.
    // The reference compresses the block as streamed, before its pad: every block's own length comes from the encode
    // call's side output (xsi_hip_ctx_set_block_sizes_out).  The blocks go to the pool; frames that are ready are written,
    // in order, here and at finalize - this thread returns as soon as the queue is short enough, so that the GPU encodes
    // the next batch while the pool is still busy with this one.
    HIP_TRY(hipMemcpy(w->h_sizes.data(), w->d_sizes, 4ull * n_blocks, hipMemcpyDeviceToHost));
    for (uint64_t k = 0; k < n_blocks; ++k) {
        const uint64_t off = w->h_offs[k] - 256, usize = w->h_sizes[k];
        if (off + usize > w->h_out.size() || usize < 16) return set_error(XSI_ERR_FORMAT, "writer: block %llu of the batch has no sane length", (unsigned long long)k);
        auto j = std::make_shared<ZstdJob>();
        j->raw = w->zpool->take_buf();
        j->raw.assign(w->h_out.data() + off, w->h_out.data() + off + usize);
        j->usize = usize;
        w->zpool->submit(j);
    }
    return writer_drain_zstd(w, /*all=*/false);
}

static int writer_join(xsi_writer* w) {
    if (w->worker_active) {
        w->worker.join();
        w->worker_active = false;
    }
    if (w->worker_rc) return set_error(w->worker_rc, "%s", w->worker_err.c_str());
    return XSI_OK;
}

// hand the batch being filled to the worker and start filling the other one
static int writer_flush_batch(xsi_writer* w) {
    if (!w->lines_in_batch) return writer_join(w);
    int rc = writer_ship_chunk(w);
    if (rc) return rc;
    rc = writer_join(w);  // the previous batch must be done: its device rows and the output buffers are reused
    if (rc) return rc;
    HIP_TRY(hipEventRecord(w->batch_copied, w->copy_stream));
    const int b = w->cur;
    const uint64_t n_lines = w->lines_in_batch;
    w->worker_active = true;
    w->worker = std::thread([w, b, n_lines] {
        // nothing may leave this thread as an exception (std::terminate): an allocation failure is an error code
        int r;
        try {
            r = writer_encode_batch(w, b, n_lines);
        } catch (const std::exception& e) {
            r = set_error(XSI_ERR_IO, "writer: %s while encoding a batch", e.what());
        }
        if (r) {
            w->worker_rc = r;
            w->worker_err = xsi_hip_last_error();
        }
    });
    w->cur ^= 1;
    if (w->batch_limit < w->batch_blocks) w->batch_limit = w->batch_limit * 2u < w->batch_blocks ? w->batch_limit * 2u : w->batch_blocks;
    w->lines_in_batch = w->lines_on_device = 0;
    w->ngt[w->cur].clear();
    w->n_allele[w->cur].clear();
    w->fast[w->cur].clear();
    w->ones[w->cur].clear();
    return XSI_OK;
}

static void writer_free(xsi_writer* w) {
    if (w->worker_active) w->worker.join();
    if (w->f) fclose(w->f);
    for (int i = 0; i < 2; ++i) {
        if (w->d_rows[i]) (void)hipFree(w->d_rows[i]);
        if (w->d_bits[i]) (void)hipFree(w->d_bits[i]);
        if (w->d_fast[i]) (void)hipFree(w->d_fast[i]);
        if (w->d_ones[i]) (void)hipFree(w->d_ones[i]);
        if (w->h_bits_chunks[i]) (void)hipHostFree(w->h_bits_chunks[i]);
        if (w->h_chunks[i]) (void)hipHostFree(w->h_chunks[i]);
        if (w->chunk_done[i]) (void)hipEventDestroy(w->chunk_done[i]);
    }
    delete w->zpool;  // (joins its threads; jobs not yet taken are dropped)
    if (w->d_sizes) (void)hipFree(w->d_sizes);
    if (w->d_out) (void)hipFree(w->d_out);
    if (w->d_offs) (void)hipFree(w->d_offs);
    if (w->batch_copied) (void)hipEventDestroy(w->batch_copied);
    if (w->copy_stream) (void)hipStreamDestroy(w->copy_stream);
    delete w;
}

extern "C" {

uint64_t xsi_hip_encode_gt_bound(const xsi_encode_params* p, uint64_t n_bcf_lines, uint64_t n_binary_lines) {
    if (!p) return 0;
    const uint64_t N = 2ull * p->n_samples;
    const uint64_t aet = p->n_samples <= 65535u ? 2 : 4;
    const uint64_t G = (N + 14) / 15;
    // per BCF line: missing + end-of-vector entries share at most N positions (2 counts), or two WAH
    // lines with --wah-encode-missing; plus one phase WAH line
    uint64_t side = 2 * aet + N * aet;
    if (p->wah_encode_missing && 2 * G * 2 > side) side = 2 * G * 2;
    side += G * 2;
    return xsi_hip_encode_bound(p, n_bcf_lines, n_binary_lines) + n_bcf_lines * side;
}

int xsi_writer_open(xsi_writer** out, xsi_hip_ctx* ctx, const char* path, const xsi_encode_params* p,
                    const char* const* sample_names) {
    if (!out || !ctx || !path || !p) return set_error(XSI_ERR_ARG, "writer_open: null argument");
    if (!p->n_samples || !p->block_len || p->block_len > MAX_BIN_PER_BLOCK)
        return set_error(XSI_ERR_ARG, "writer_open: bad n_samples / block_len");
    if (at_mismatch_window(p->n_samples))
        return set_error(XSI_ERR_UNSUPPORTED, "%s: %u samples fall in the reference's A_T mismatch window (32768..65535: 16-bit "
                         "block data under a 32-bit header, prefix array wraps modulo 65536); it cannot be encoded decodably", "writer_open",
                         p->n_samples);
    if (p->zstd_level && !zstd_api().ok) return set_error(XSI_ERR_UNSUPPORTED, "--zstd requested but libzstd.so.1 could not be loaded");
    HIP_TRY(hipSetDevice(ctx->device));
    xsi_writer* w = new xsi_writer();
    w->ctx = ctx;
    w->p = *p;
    w->N = 2ull * p->n_samples;
    for (uint32_t i = 0; i < p->n_samples; ++i) w->names.emplace_back(sample_names ? sample_names[i] : "");
    w->f = fopen(path, "wb");
    if (!w->f) {
        delete w;
        return set_error(XSI_ERR_IO, "Failed to open file %s", path);
    }
    // provisional header, rewritten by finalize (xsi_factory.hpp:468-511)
    uint8_t zero[256] = {0};
    if (fwrite(zero, 1, 256, w->f) != 256) {
        writer_free(w);
        return set_error(XSI_ERR_IO, "short write");
    }
    w->file_pos = 256;
    const size_t row_bytes = (size_t)w->N * sizeof(int32_t);
    const size_t block_bytes = row_bytes * p->block_len;
    // Blocks per batch.  The device is two orders of magnitude faster than one host thread can feed it, so a batch
    // is not sized to fill the GPU but to keep append() from ever waiting: while the worker encodes batch k (a
    // latency of about 20 ms + 0.5 ms per 1000 haplotypes, set by the serial PBWT chain of a block, whatever the
    // number of blocks) the caller must be busy filling batch k + 1, at ~10 G cells/s of packing: twice that
    // latency's worth of blocks, at most 64 and at most what a quarter of the free HBM holds twice.  Small batches
    // also start the overlap early (a file shorter than one batch is encoded only at finalize).
    // XSI_WRITER_BATCH_BLOCKS overrides.
    {
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) free_b = 0;
        uint64_t fit = free_b / 4 / (block_bytes ? block_bytes : 1) / 2;
        const double gpu_ms = 20.0 + 0.5 * (double)w->N / 1000.0;
        const double host_ms_per_block = (double)p->block_len * (double)w->N / 10e9 * 1e3;
        uint64_t k = (uint64_t)(2.0 * gpu_ms / (host_ms_per_block > 1e-3 ? host_ms_per_block : 1e-3)) + 1u;
        if (k > 64) k = 64;
        if (k > fit) k = fit;
        if (const char* e = tuning_env("XSI_WRITER_BATCH_BLOCKS")) k = strtoull(e, nullptr, 10);
        // (--zstd keeps these batches: a batch costs the chain's latency whatever its size - batches of four and of six
        // blocks measured 1.28 and 1.06 G cells/s at 5008 haplotypes x 100 000 lines where eleven give 2.7 -, and the pool takes
        // a batch's blocks side by side)
        if (k < 1) k = 1;
        w->batch_blocks = (uint32_t)k;
    }
    if (p->zstd_level) {
        // pool: the host's cores less the caller's and the worker's, at most 16 (XSI_WRITER_ZSTD_THREADS overrides, 1 = the
        // one-thread writer of rounds 1 - 5 for A/B runs); the queue may hold two batches' blocks before the worker waits
        unsigned hc = std::thread::hardware_concurrency();
        unsigned nt = hc > 3 ? hc - 2 : 2;
        if (nt > 16) nt = 16;
        if (const char* e = tuning_env("XSI_WRITER_ZSTD_THREADS")) nt = (unsigned)atoi(e);
        if (nt < 1) nt = 1;
        try {  // (threads that cannot be started: an error code, not an exception through the C ABI)
            w->zpool = new ZstdPool();
            w->zpool->prof = tuning_env("XSI_WRITER_PROF") != nullptr;
            w->zpool->start(nt, (int)p->zstd_level);
        } catch (const std::exception& ex) {
            if (!w->zpool || w->zpool->threads.empty()) {
                writer_free(w);
                return set_error(XSI_ERR_IO, "writer_open: the compression pool could not be started: %s", ex.what());
            }
            // (some threads run: the pool works with those)
        }
        w->zstd_max_pending = nt > 1 ? (size_t)2 * w->batch_blocks : 0;
        w->h_sizes.resize(w->batch_blocks);
    }
    w->batch_limit = (p->zstd_level && w->batch_blocks > 2 && !tuning_env("XSI_WRITER_BATCH_BLOCKS")) ? 2u : w->batch_blocks;
    hipError_t e = hipSuccess;
    w->bit_stride = (uint32_t)(((w->N + 1023u) / 1024u) * 128u);
    for (int i = 0; i < 2 && e == hipSuccess; ++i) e = hipMalloc((void**)&w->d_rows[i], block_bytes * w->batch_blocks);
    for (int i = 0; i < 2 && e == hipSuccess; ++i) e = hipMalloc((void**)&w->d_bits[i], (size_t)w->bit_stride * p->block_len * w->batch_blocks);
    for (int i = 0; i < 2 && e == hipSuccess; ++i) e = hipMalloc((void**)&w->d_fast[i], (size_t)p->block_len * w->batch_blocks);
    for (int i = 0; i < 2 && e == hipSuccess; ++i) e = hipMalloc((void**)&w->d_ones[i], 4ull * p->block_len * w->batch_blocks);
    if (e == hipSuccess && p->zstd_level) e = hipMalloc((void**)&w->d_sizes, 4ull * w->batch_blocks);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&w->copy_stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&w->batch_copied, hipEventDisableTiming);
    size_t chunk_bytes = 64ull << 20;
    w->chunk_rows = (uint32_t)(chunk_bytes / row_bytes);
    if (w->chunk_rows < 1) w->chunk_rows = 1;
    if (w->chunk_rows > p->block_len) w->chunk_rows = p->block_len;
    for (int i = 0; i < 2 && e == hipSuccess; ++i) {
        e = hipHostMalloc((void**)&w->h_chunks[i], row_bytes * w->chunk_rows, hipHostMallocDefault);
        if (e == hipSuccess) e = hipHostMalloc((void**)&w->h_bits_chunks[i], (size_t)w->bit_stride * w->chunk_rows, hipHostMallocDefault);
        if (e == hipSuccess) memset(w->h_bits_chunks[i], 0, (size_t)w->bit_stride * w->chunk_rows);  // the pad of every row stays zero
        if (e == hipSuccess) e = hipEventCreateWithFlags(&w->chunk_done[i], hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventRecord(w->chunk_done[i], w->copy_stream);  // "free" from the start
    }
    w->h_chunk = w->h_chunks[0];
    w->h_bits_chunk = w->h_bits_chunks[0];
    if (e != hipSuccess) {
        writer_free(w);
        return set_error(XSI_ERR_HIP, "writer buffers: %s", hipGetErrorString(e));
    }
    *out = w;
    return XSI_OK;
}

int32_t* xsi_writer_row_buffer(xsi_writer* w) {
    if (!w || !w->f) {
        set_error(XSI_ERR_ARG, "writer_row_buffer: null / closed writer");
        return nullptr;
    }
    // check_flush_block, xsi_factory.hpp:527-539, K blocks at a time
    if (w->lines_in_batch == (uint64_t)w->p.block_len * w->batch_limit) {
        if (writer_flush_batch(w)) return nullptr;
    }
    return w->h_chunk + (size_t)w->chunk_fill * w->N;
}

static int writer_commit(xsi_writer* w, uint32_t ngt, uint32_t n_allele, bool line_is_packed) {
    if (!w || !w->f) return set_error(XSI_ERR_ARG, "writer_commit_row: null / closed writer");
    if (ngt != w->p.n_samples && ngt != 2u * w->p.n_samples)
        return set_error(XSI_ERR_ARG, "PLOIDY ERROR: %u values for %u samples", ngt, w->p.n_samples);
    if (n_allele < 2) return set_error(XSI_ERR_UNSUPPORTED, "lines without an ALT allele are rejected (see xsi_hip_encode_gt)");
    if (w->lines_in_batch == (uint64_t)w->p.block_len * w->batch_limit)
        return set_error(XSI_ERR_ARG, "writer_commit_row without xsi_writer_row_buffer");
    if (!line_is_packed) {
        // the caller filled the int32 slot (bcf_get_genotypes' destination): pack from it when the line allows, so
        // that a chunk of such lines still ships bit rows only
        line_is_packed = ngt == 2u * w->p.n_samples && n_allele == 2 &&
                         xsi::pack_bit_row(w->h_chunk + (size_t)w->chunk_fill * w->N, ngt, w->p.default_phased,
                                      w->h_bits_chunk + (size_t)w->chunk_fill * w->bit_stride);
    }
    if (!line_is_packed) {
        w->chunk_general++;
        w->general_in_batch++;
    }
    w->fast[w->cur].push_back(line_is_packed ? 1 : 0);
    {  // the line's ALT count, from the bit row just packed (N / 64 popcounts against the 4 N bytes the packer read)
        uint32_t c = 0;
        if (line_is_packed) {
            const uint64_t* q = reinterpret_cast<const uint64_t*>(w->h_bits_chunk + (size_t)w->chunk_fill * w->bit_stride);
            for (uint32_t i = 0, n64 = (ngt + 63u) / 64u; i < n64; ++i) c += (uint32_t)__builtin_popcountll(q[i]);
        }
        w->ones[w->cur].push_back(c);
    }
    w->chunk_fill++;
    w->lines_in_batch++;
    w->ngt[w->cur].push_back(ngt);
    w->n_allele[w->cur].push_back(n_allele);
    const uint32_t pl = ngt / w->p.n_samples;
    if (pl > w->max_ploidy_seen) w->max_ploidy_seen = pl;
    w->variant_counter += n_allele - 1;
    w->entry_counter++;
    if (w->chunk_fill == w->chunk_rows) return writer_ship_chunk(w);
    return XSI_OK;
}

int xsi_writer_commit_row(xsi_writer* w, uint32_t ngt, uint32_t n_allele) { return writer_commit(w, ngt, n_allele, false); }

int xsi_debug_pack_bit_row(const int32_t* h_gt, uint32_t n, int32_t default_phased, uint8_t* h_bits) {
    if (!h_gt || !h_bits) return set_error(XSI_ERR_ARG, "debug_pack_bit_row: null argument");
    return xsi::pack_bit_row(h_gt, n, default_phased, h_bits) ? 1 : 0;
}

int xsi_writer_append(xsi_writer* w, const int32_t* h_gt, uint32_t ngt, uint32_t n_allele) {
    if (!w || !h_gt) return set_error(XSI_ERR_ARG, "writer_append: null argument");
    if (ngt != w->p.n_samples && ngt != 2u * w->p.n_samples)
        return set_error(XSI_ERR_ARG, "PLOIDY ERROR: %u values for %u samples", ngt, w->p.n_samples);
    if (n_allele < 2) return set_error(XSI_ERR_UNSUPPORTED, "lines without an ALT allele are rejected (see xsi_hip_encode_gt)");
    int32_t* dst = xsi_writer_row_buffer(w);
    if (!dst) return XSI_ERR_HIP;  // the flush's own message stands
    // one pass over the caller's row: 4 N bytes read, N / 8 written; only a line the bit form cannot hold is copied
    const bool packed = ngt == 2u * w->p.n_samples && n_allele == 2 &&
                        xsi::pack_bit_row(h_gt, ngt, w->p.default_phased, w->h_bits_chunk + (size_t)w->chunk_fill * w->bit_stride);
    if (!packed) memcpy(dst, h_gt, (size_t)ngt * sizeof(int32_t));
    return writer_commit(w, ngt, n_allele, packed);
}

int xsi_writer_finalize(xsi_writer* w, uint32_t max_ploidy) {
    if (!w || !w->f) return set_error(XSI_ERR_ARG, "writer_finalize: null / closed writer");
    int rc = writer_flush_batch(w);
    if (rc) return rc;
    rc = writer_join(w);
    if (rc) return rc;
    rc = writer_drain_zstd(w, /*all=*/true);  // (--zstd: the frames still with the pool)
    if (rc) return rc;
    // xsi_factory.hpp:558-605
    while (w->file_pos % 8) {
        if (fputc(0, w->f) == EOF) return set_error(XSI_ERR_IO, "short write");
        w->file_pos++;
    }
    xsi_header_fields hf{};
    hf.n_samples = w->p.n_samples;
    hf.max_ploidy = max_ploidy ? max_ploidy : w->max_ploidy_seen;
    hf.block_len = w->p.block_len;
    hf.mac_threshold = w->p.mac_threshold;
    hf.default_phased = w->p.default_phased;
    hf.zstd = w->p.zstd_level ? 1 : 0;
    hf.num_variants = w->variant_counter;
    hf.xcf_entries = w->entry_counter;
    hf.indices_offset = w->file_pos;
    if (!w->indices.empty() && fwrite(w->indices.data(), 8, w->indices.size(), w->f) != w->indices.size())
        return set_error(XSI_ERR_IO, "short write");
    w->file_pos += 8 * w->indices.size();
    hf.samples_offset = w->file_pos;
    for (auto& s : w->names)
        if (fwrite(s.c_str(), 1, s.size() + 1, w->f) != s.size() + 1) return set_error(XSI_ERR_IO, "short write");
    uint8_t h[256];
    xsi_hip_make_header(&hf, h);
    if (fflush(w->f) != 0 || fseek(w->f, 0, SEEK_SET) != 0) return set_error(XSI_ERR_IO, "header rewrite failed");
    if (fwrite(h, 1, 256, w->f) != 256) return set_error(XSI_ERR_IO, "short write");
    const int crc = fclose(w->f);
    w->f = nullptr;
    if (crc != 0) return set_error(XSI_ERR_IO, "close failed");
    return XSI_OK;
}

void xsi_writer_close(xsi_writer* w) {
    if (!w) return;
    writer_free(w);
}

}  // extern "C"

// ------------------------------------------------------------------------------------------
// accessor
// ------------------------------------------------------------------------------------------
struct xsi_accessor {
    xsi_hip_ctx* ctx = nullptr;  // private context (own stream + workspace) on the caller's device
    std::vector<uint8_t> file;
    uint8_t* d_file = nullptr;
    uint32_t version = 0, aet = 0, ploidy = 0;
    uint64_t hap_samples = 0, num_samples = 0, n_blocks = 0;
    std::vector<std::string> names;
    // current block: a view (P, D) of either a cache entry or, when the cache cannot take it, the
    // context workspace
    int64_t cur_block = -1;
    bool cur_in_workspace = false;
    DecodePlan P;
    DecodedPlanes D;
    bool biallelic = false;
    // decoded blocks kept in HBM (LRU): the reference replays a block's prefix on every backward or
    // random seek (accessor_internals_new.hpp:154-196); here a block is decoded once and stays
    // resident, so a revisit costs one compose kernel
    struct CachedBlock {
        uint64_t block = 0, last_use = 0;
        size_t bytes = 0;
        uint8_t* mem = nullptr;
        DecodePlan P;
        DecodedPlanes D;
        bool biallelic = false;
        // Prefix decode: only the first wah_done WAH lines of the block have been through the chain (sparse lines and
        // side channels are complete); the chain's ranks behind line wah_done are parked in d_state, and a query
        // further into the block continues from there (accessor_ensure_lines).
        bool partial = false;
        uint32_t wah_done = 0;
        uint32_t* d_state = nullptr;           // inside mem: the chain's parked ranks, then 64 bytes of side-matrix cursors
        uint64_t* d_walk = nullptr;
        std::vector<uint32_t> wah_before;      // [n_bin + 1]: WAH lines among the binary lines in front of line i
        // the parsed part of the plan (line lists, ranks, WAH line starts: P.L.* point into mem) is kept with a prefix-decoded
        // block, so that a continuation neither parses the dictionaries nor scans the WAH words again
        bool has_plan = false;
    };
    // (a list: entries keep their address while the read-ahead thread inserts others)
    std::list<CachedBlock> cache;
    // the cache entry of cur_block (nullptr: none / the block lives in the context workspace): entries keep their address and
    // the read-ahead thread never evicts this one, so the per-query paths read it without the lock (the thread holds the lock
    // across a hipMalloc of hundreds of megabytes: milliseconds)
    CachedBlock* cur_entry = nullptr;
    std::mutex cache_m;  // guards cache, cache_bytes, tick: the read-ahead thread inserts what it has decoded
    uint64_t prefix_decodes = 0, prefix_extensions = 0;
    size_t cache_bytes = 0, cache_budget = 0;
    uint64_t tick = 0, cache_hits = 0, cache_misses = 0;
    // Sequential scans (the reference's only published benchmark loads every line in order, loading_time/gt_loader_new.hpp:
    // 112-172): queries that follow one another line by line are counted; from SEQ_MIN of them on, a prefix-decoded block
    // is finished in one continuation instead of a dozen growing ones, and block b + 1 is decoded by a second thread with
    // a context (stream + workspace) of its own while block b is being served, so that its first touch finds it in HBM.
    int64_t seq_block = -1;
    uint32_t seq_next = 0, seq_run = 0;
    std::thread pf_thread;
    bool pf_started = false;       // pf_thread is joinable
    int64_t pf_block = -1;         // the block it decodes / decoded last
    xsi_hip_ctx* pf_ctx = nullptr;
    uint64_t readahead_started = 0, readahead_hits = 0;
    // zstd files: every block is inflated on the host into a one-block image (header + block + index)
    bool zstd = false;
    std::vector<uint8_t> mini;
    uint8_t* d_mini = nullptr;
    size_t d_mini_cap = 0;
    // counts-only view of a block (fill_allele_counts never expands genotypes)
    int64_t cnt_block = -1;
    std::vector<uint32_t> cnt_ones;
    std::vector<uint8_t> cnt_kind;
    // composed window of a bi-allelic block, or the single last composed line
    int32_t* d_rows = nullptr;
    int32_t* h_rows = nullptr;  // pinned
    // A page-locked destination array (xsi_accessor_alloc_array, or memory the caller page-locked by an allocation of
    // its own) registered with xsi_accessor_register_array: a single-line request is stored by the compose kernel
    // straight into it, without the stop in the pinned window (at 200 000 haplotypes that memcpy is a third of a warm
    // random query).  The accessor never page-locks caller memory itself (see xsi_accessor_register_array).
    void* reg_dst = nullptr;        // the array registered with xsi_accessor_register_array
    size_t reg_bytes = 0;
    int32_t* reg_dev = nullptr;     // the device's address of that array (the compose kernel may store into it)
    std::vector<std::pair<void*, size_t>> owned_arrays;  // xsi_accessor_alloc_array (pointer, bytes): freed at close at the latest
    uint64_t n_full = 0;            // values of a composed row: 2 * num_samples (hap_samples of a v4 file without the field)
    int32_t* direct_dst = nullptr;  // set for the duration of one call: where a single composed line should land
    bool direct_done = false;       // this call's line went there (not into h_rows)
    bool win_in_rows = true;        // the window's lines are in h_rows (false after a direct single-line copy)
    uint64_t* h_counts = nullptr;  // pinned [win][2] or [1][max]
    uint32_t* h_meta = nullptr;    // pinned
    uint32_t* h_bmeta = nullptr;   // pinned, batched queries: [4][bmeta_cap] first binary line, alleles, output row, values
    uint32_t bmeta_cap = 0;
    uint32_t win_rows = 0, win_first = 0, win_n = 0, win_target = 1;
    int64_t win_block = -1;
    uint32_t counts_cap = 0;
    std::vector<uint64_t> last_counts;
    // sample selection (NewDecompressor --samples, gt_decompressor_new.hpp:209-238)
    uint32_t n_sel = 0;
    uint32_t* d_sel = nullptr;
    int32_t* d_sel_row = nullptr;
    int32_t* h_sel_row = nullptr;  // pinned, 2 * n_sel values
    uint32_t* h_sel_ac = nullptr;  // pinned, 32 counters (device-written)
};

// Device image + block number to hand the decoder for file block `block` (set_block_ptr,
// accessor_internals_new.hpp:845-893).  Plain files: the whole file already sits in HBM.  zstd files:
// inflate the block on the host (ZSTD_decompress) and upload a one-block image.
static int accessor_block_image(xsi_accessor* a, uint64_t block, const uint8_t** d_img, uint64_t* len, uint64_t* blk) {
    if (block >= a->n_blocks) return set_error(XSI_ERR_ARG, "block %llu beyond the %llu blocks of the file",
                                               (unsigned long long)block, (unsigned long long)a->n_blocks);
    if (!a->zstd) {
        *d_img = a->d_file;
        *len = a->file.size();
        *blk = block;
        return XSI_OK;
    }
    const uint8_t* h = a->file.data();
    auto get = [&](size_t off, int bytes) {
        uint64_t v = 0;
        for (int i = 0; i < bytes; ++i) v |= (uint64_t)h[off + i] << (8 * i);
        return v;
    };
    const uint64_t io = get(72, 8);
    const uint64_t off = a->version >= 5 ? get(io + block * 8, 8) : get(io + block * 4, 4);
    const uint64_t sz = a->version >= 5 ? 8 : 4;
    // untrusted file values: every comparison is written so that it cannot wrap
    const uint64_t fsz = a->file.size();
    if (off > fsz || fsz - off < 2 * sz) return set_error(XSI_ERR_FORMAT, "block offset outside the file");
    const uint64_t csize = get(off, (int)sz), usize = get(off + sz, (int)sz);
    if (csize > fsz - off - 2 * sz || usize > (1ull << 34)) return set_error(XSI_ERR_FORMAT, "corrupt zstd block header");
    const ZstdApi& z = zstd_api();
    if (!z.ok) return set_error(XSI_ERR_UNSUPPORTED, "zstd-compressed file but libzstd.so.1 could not be loaded");
    size_t body = (size_t)usize;
    while ((256 + body) % 8) body++;
    try {  // usize comes from the file: an absurd value must end as an error, not as std::bad_alloc through extern "C"
        a->mini.assign(256 + body + 8, 0);
    } catch (const std::exception&) {
        return set_error(XSI_ERR_FORMAT, "zstd block header declares %llu bytes: cannot allocate", (unsigned long long)usize);
    }
    memcpy(a->mini.data(), h, 256);
    a->mini[17] &= (uint8_t)~4u;  // the image handed to the GPU is not compressed
    const size_t r = z.decompress(a->mini.data() + 256, (size_t)usize, h + off + 2 * sz, (size_t)csize);
    if (z.is_error(r) || r != usize) return set_error(XSI_ERR_FORMAT, "Failed to decompress block");
    auto put = [&](size_t o, uint64_t v, int bytes) {
        for (int i = 0; i < bytes; ++i) a->mini[o + i] = (uint8_t)(v >> (8 * i));
    };
    put(8, 5, 4);  // the one-block image always carries a u64 index
    put(72, 256 + body, 8);
    put(80, 256 + body + 8, 8);
    put(256 + body, 256, 8);
    if (a->mini.size() > a->d_mini_cap) {
        if (a->d_mini) (void)hipFree(a->d_mini);
        a->d_mini = nullptr;
        a->d_mini_cap = a->mini.size() + a->mini.size() / 4;
        HIP_TRY(hipMalloc((void**)&a->d_mini, a->d_mini_cap));
    }
    HIP_TRY(hipMemcpy(a->d_mini, a->mini.data(), a->mini.size(), hipMemcpyHostToDevice));
    *d_img = a->d_mini;
    *len = a->mini.size();
    *blk = 0;
    return XSI_OK;
}

using CacheIt = std::list<xsi_accessor::CachedBlock>::iterator;

static void accessor_evict(xsi_accessor* a, CacheIt it) {  // (cache_m held)
    (void)hipFree(it->mem);
    a->cache_bytes -= it->bytes;
    a->cache.erase(it);
}

// least recently used entry other than block `protect` (< 0: any); end() when there is none  (cache_m held)
static CacheIt accessor_lru(xsi_accessor* a, int64_t protect) {
    CacheIt lru = a->cache.end();
    for (CacheIt it = a->cache.begin(); it != a->cache.end(); ++it) {
        if (protect >= 0 && it->block == (uint64_t)protect) continue;
        if (lru == a->cache.end() || it->last_use < lru->last_use) lru = it;
    }
    return lru;
}

static xsi_accessor::CachedBlock* accessor_find(xsi_accessor* a, uint64_t block) {  // (cache_m held)
    for (auto& e : a->cache)
        if (e.block == block) return &e;
    return nullptr;
}

// Copy the decoded state of a block that sits in a context's workspace into a private allocation: `e` receives plan and
// planes repointed at it.  Returns false when the budget / HBM cannot take it.
struct PartialInfo {
    uint32_t wah_done = 0;
    const uint32_t* ws_state = nullptr;  // the chain's parked ranks in the context workspace
    size_t state_bytes = 0;
    std::vector<uint32_t>* wah_before = nullptr;
};

// protect_current: the block the main thread is serving must stay (the read-ahead thread is making room: cur_block is read
// under the lock, where the main thread sets it on a cache hit)
static bool accessor_cache_build(xsi_accessor* a, xsi_hip_ctx* ctx, const DecodePlan& P, const DecodedPlanes& D, bool biallelic,
                                 uint64_t block, const PartialInfo* part, bool protect_current, xsi_accessor::CachedBlock* out) {
    const size_t n_bin = P.n_bin ? P.n_bin : 1;
    const size_t plane_b = 4ull * D.stride_w * n_bin;
    auto al = [](size_t v) { return (v + 255u) & ~(size_t)255u; };
    const bool side = D.has_side;
    const size_t need = al(plane_b) * (side ? 4u : 1u) + al(n_bin) * (side ? 2u : 1u) + al(4 * n_bin + 64) * 4u +
                        al(sizeof(DecBlock)) + (part ? al(part->state_bytes) + 6u * al(4 * n_bin + 64) : 0u);
    uint8_t* mem = nullptr;
    {
        std::lock_guard<std::mutex> lk(a->cache_m);
        const int64_t protect = protect_current && !a->cur_in_workspace ? a->cur_block : -1;
        if (need > a->cache_budget) return false;
        while (a->cache_bytes + need > a->cache_budget) {
            CacheIt lru = accessor_lru(a, protect);
            if (lru == a->cache.end()) return false;
            accessor_evict(a, lru);
        }
        while (hipMalloc((void**)&mem, need) != hipSuccess) {
            (void)hipGetLastError();
            CacheIt lru = accessor_lru(a, protect);
            if (lru == a->cache.end()) return false;
            accessor_evict(a, lru);
        }
        a->cache_bytes += need;  // reserved; given back below if the copy fails
    }
    hipStream_t s = ctx->stream;
    size_t off = 0;
    bool ok = true;
    auto take = [&](const void* src, size_t bytes) -> void* {
        void* dst = mem + off;
        if (src && bytes && hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, s) != hipSuccess) ok = false;
        off += al(bytes);
        return dst;
    };
    xsi_accessor::CachedBlock& e = *out;
    e.P = P;
    e.D = D;
    e.D.planes = (uint32_t*)take(D.planes, plane_b);
    e.P.L.kind = (uint8_t*)take(P.L.kind, n_bin);
    e.P.L.ones = (uint32_t*)take(P.L.ones, 4 * n_bin);
    e.P.L.line_block = (uint32_t*)take(P.L.line_block, 4 * n_bin);
    e.D.n_miss = (uint32_t*)take(D.n_miss, 4 * n_bin + 64);
    e.D.n_eov = (uint32_t*)take(D.n_eov, 4 * n_bin + 64);
    e.P.d_blocks = (DecBlock*)take(P.d_blocks, sizeof(DecBlock));
    if (side) {
        e.D.side = (uint8_t*)take(D.side, n_bin);
        e.D.miss_planes = (uint32_t*)take(D.miss_planes, plane_b);
        e.D.eov_planes = (uint32_t*)take(D.eov_planes, plane_b);
        e.D.phase_planes = (uint32_t*)take(D.phase_planes, plane_b);
    }
    if (part) {
        e.partial = true;
        e.wah_done = part->wah_done;
        e.d_state = (uint32_t*)take(part->ws_state, part->state_bytes);
        e.d_walk = reinterpret_cast<uint64_t*>(reinterpret_cast<uint8_t*>(e.d_state) + part->state_bytes - 64u);
        e.wah_before = *part->wah_before;
        // the parsed plan: what a continuation reads instead of parsing and scanning the block again
        e.P.L.rank = (uint32_t*)take(P.L.rank, 4 * n_bin + 64);
        e.P.L.wah_start = (decltype(e.P.L.wah_start))take(P.L.wah_start, 4 * n_bin + 64);
        e.P.L.sparse_start = (decltype(e.P.L.sparse_start))take(P.L.sparse_start, 4 * n_bin + 64);
        e.P.L.wah_lines = (uint32_t*)take(P.L.wah_lines, 4 * n_bin + 64);
        e.P.L.sparse_lines = (uint32_t*)take(P.L.sparse_lines, 4 * n_bin + 64);
        e.P.L.wah_cumg = (decltype(e.P.L.wah_cumg))take(P.L.wah_cumg, 4 * n_bin + 64);
        e.has_plan = true;
    }
    if (!ok || hipStreamSynchronize(s) != hipSuccess || off > need) {
        (void)hipGetLastError();
        (void)hipFree(mem);
        std::lock_guard<std::mutex> lk(a->cache_m);
        a->cache_bytes -= need;
        return false;
    }
    e.block = block;
    e.bytes = need;
    e.mem = mem;
    e.biallelic = biallelic;
    return true;
}

// The block the main thread has just decoded in its context's workspace -> cache; (P, D) repointed at the entry.
static bool accessor_cache_store(xsi_accessor* a, uint64_t block, const PartialInfo* part = nullptr) {
    xsi_accessor::CachedBlock e;
    if (!accessor_cache_build(a, a->ctx, a->P, a->D, a->biallelic, block, part, false, &e)) return false;
    std::lock_guard<std::mutex> lk(a->cache_m);
    e.last_use = ++a->tick;
    a->P = e.P;
    a->D = e.D;
    a->cache.push_back(std::move(e));
    a->cur_entry = &a->cache.back();  // (the callers make this block the current one)
    return true;
}

// WAH lines to have decoded so that every binary line below `need_bin` is valid (0 or beyond the block: all), growing
// geometrically from what is there so that a scan through a cold block is a handful of extensions, not one per line
static uint32_t prefix_target(const std::vector<uint32_t>& wah_before, uint32_t n_wah, uint32_t wah_done, uint32_t need_bin) {
    const uint32_t n_bin = (uint32_t)wah_before.size() - 1u;
    const uint32_t k_need = (need_bin == 0u || need_bin >= n_bin) ? n_wah : wah_before[need_bin];
    if (k_need <= wah_done) return wah_done;
    uint32_t k = wah_done + (wah_done / 2u > 32u ? wah_done / 2u : 32u);
    if (k < k_need) k = k_need;
    if (k > n_wah || (uint64_t)k * 10u >= (uint64_t)n_wah * 9u) k = n_wah;  // the last tenth is not worth another launch sequence
    return k;
}

// every binary line in front of the WAH line of rank k is valid once k WAH lines have been through the chain
static uint32_t bin_valid_of(const std::vector<uint32_t>& wah_before, uint32_t n_wah, uint32_t k) {
    const uint32_t n_bin = (uint32_t)wah_before.size() - 1u;
    if (k >= n_wah) return n_bin;
    // smallest i with wah_before[i + 1] > k: the id of the WAH line of rank k
    return (uint32_t)(std::upper_bound(wah_before.begin() + 1, wah_before.end(), k) - (wah_before.begin() + 1));
}

// four or more single-line queries have followed one another line by line (accessor_line_view counts them)
static bool accessor_sequential(const xsi_accessor* a) { return a->seq_run >= 4u && !tuning_env("XSI_ACCESSOR_NO_READAHEAD"); }

// The current block is a prefix-decoded cache entry and the caller is about to read binary lines below need_bin (0: all):
// run the chain on from where it stopped (the reference's seek does the same replay, one line at a time, on the host:
// accessor_internals_new.hpp:154-196).
static int accessor_ensure_lines(xsi_accessor* a, uint32_t need_bin) {
    if (a->cur_block < 0 || a->cur_in_workspace) return XSI_OK;
    xsi_accessor::CachedBlock* e = a->cur_entry;
    if (!e || e->block != (uint64_t)a->cur_block || !e->partial) return XSI_OK;
    const uint32_t n_wah = e->P.n_wah;
    // a sequential scan will read the whole block: one continuation to its end instead of a dozen growing ones
    const uint32_t target = accessor_sequential(a) && prefix_target(e->wah_before, n_wah, e->wah_done, need_bin) > e->wah_done
                                ? n_wah
                                : prefix_target(e->wah_before, n_wah, e->wah_done, need_bin);
    if (target <= e->wah_done) return XSI_OK;
    const uint8_t* img;
    uint64_t len, blk;
    int rc = accessor_block_image(a, (uint64_t)a->cur_block, &img, &len, &blk);
    if (rc) return rc;
    DecodePlan Pt;
    const bool keep_plan = e->has_plan && !tuning_env("XSI_ACCESSOR_REPARSE");
    if (keep_plan) {
        // the parsed plan of the first decode (dictionary, flag vectors, line lists, WAH line starts) is the entry's; only
        // the scratch (expanded rows, tiles, totals) is the context's
        Pt = e->P;
        rc = decode_plan_scratch(a->ctx, &Pt, /*restore_totals=*/true);
        if (rc) return rc;
    } else {
        rc = decode_prepare(a->ctx, img, len, blk, 1, &Pt);
        if (rc) return rc;
        Pt.L.ones = e->P.L.ones;  // the expansion writes the new lines' counts next to the ones already there
    }
    PartialDecode pd{e->wah_done, target, e->d_state, false, bin_valid_of(e->wah_before, n_wah, e->wah_done),
                     bin_valid_of(e->wah_before, n_wah, target), e->d_walk, keep_plan};
    DecodedPlanes Dv = e->D;
    rc = decode_all_planes(a->ctx, img, Pt, &Dv, &pd);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(a->ctx->stream));
    e->wah_done = target;
    if (target == n_wah) e->partial = false;
    ++a->prefix_extensions;
    return XSI_OK;
}

// ---- read-ahead: block b + 1 of a sequential scan, decoded whole by a thread of its own into the cache
static void accessor_readahead_join(xsi_accessor* a) {
    if (a->pf_started) {
        a->pf_thread.join();
        a->pf_started = false;
    }
}

static void accessor_readahead_body(xsi_accessor* a, uint64_t block) {
    // (errors end the attempt silently: the main thread decodes the block itself when it gets there)
    if (hipSetDevice(a->ctx->device) != hipSuccess) return;
    DecodePlan P;
    DecodedPlanes D;
    if (decode_prepare(a->pf_ctx, a->d_file, a->file.size(), block, 1, &P)) return;
    if (decode_all_planes(a->pf_ctx, a->d_file, P, &D)) return;
    xsi_accessor::CachedBlock e;
    if (!accessor_cache_build(a, a->pf_ctx, P, D, P.n_bin == P.n_bcf, block, nullptr, true, &e)) return;
    std::lock_guard<std::mutex> lk(a->cache_m);
    if (accessor_find(a, block)) {  // (the main thread got there first)
        (void)hipFree(e.mem);
        a->cache_bytes -= e.bytes;
        return;
    }
    e.last_use = ++a->tick;
    a->cache.push_back(std::move(e));
}

// called behind a served query of a sequential scan in block `cur`
static void accessor_readahead_maybe(xsi_accessor* a, uint64_t cur) {
    const uint64_t next = cur + 1u;
    if (next >= a->n_blocks || a->zstd || a->pf_block == (int64_t)next || !accessor_sequential(a)) return;
    {
        std::lock_guard<std::mutex> lk(a->cache_m);
        if (accessor_find(a, next)) return;
        // room for the block being served and the next one, or the read-ahead would throw out what is being read
        xsi_accessor::CachedBlock* c = accessor_find(a, cur);
        if (!c || 2u * c->bytes + (c->bytes >> 2) > a->cache_budget) return;
    }
    accessor_readahead_join(a);
    if (!a->pf_ctx) {
        if (xsi_hip_ctx_create(&a->pf_ctx, a->ctx->device, nullptr) != XSI_OK) {
            a->pf_ctx = nullptr;
            a->pf_block = (int64_t)next;  // (not tried again for this block)
            return;
        }
        // its kernels go to the CUs the foreground leaves idle: the compose kernel of a query is dispatched first (without
        // this a query that falls into the read-ahead's expansion or fill waits milliseconds for its turn)
        (void)xsi::ctx_make_low_priority(a->pf_ctx);
    }
    a->pf_block = (int64_t)next;
    try {  // (a thread that cannot be started is a read-ahead that does not happen, not an exception through the C ABI)
        a->pf_thread = std::thread([a, next] {
            try {
                accessor_readahead_body(a, next);
            } catch (...) {
            }
        });
    } catch (...) {
        return;
    }
    ++a->readahead_started;
    a->pf_started = true;
}

// need_bin: the binary lines below it are what the caller reads first (0: the whole block); a cold block is decoded up
// to there (prefix decode) when the ranged chain takes it and the cache can hold it
static int accessor_load_block(xsi_accessor* a, uint64_t block, uint32_t need_bin = 0) {
    if (a->pf_started && a->pf_block == (int64_t)block) accessor_readahead_join(a);  // the read-ahead is at this very block: take its result
    {
        std::unique_lock<std::mutex> lk(a->cache_m);
        if (xsi_accessor::CachedBlock* e = accessor_find(a, block)) {
            e->last_use = ++a->tick;
            a->P = e->P;
            a->D = e->D;
            a->biallelic = e->biallelic;
            a->cur_block = (int64_t)block;
            a->cur_entry = e;
            a->cur_in_workspace = false;
            a->win_n = 0;
            ++a->cache_hits;
            if (a->pf_block == (int64_t)block) ++a->readahead_hits;
            const bool partial = e->partial;
            lk.unlock();
            return partial ? accessor_ensure_lines(a, need_bin) : XSI_OK;
        }
    }
    // a decode in this thread reuses the context workspace the current block may live in, and evicts: the read-ahead thread
    // must not be making room at the same time with a stale idea of what is being served
    accessor_readahead_join(a);
    ++a->cache_misses;
    // XSI_ACCESSOR_PROF=1: wall clock of the pieces of a first touch (synchronising between them) on stderr
    const bool prof = tuning_env("XSI_ACCESSOR_PROF") != nullptr;
    auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double t_prof = prof ? now() : 0.0;
    auto lap = [&](const char* what) {
        if (!prof) return;
        (void)hipStreamSynchronize(a->ctx->stream);
        const double t = now();
        fprintf(stderr, "[xsi accessor prof] block %llu %-28s %8.3f ms\n", (unsigned long long)block, what, t - t_prof);
        stage_collect(a->ctx);
        for (int i = 0; i < XSI_STAGE_COUNT; ++i)
            if (a->ctx->stage_n[i]) {
                fprintf(stderr, "[xsi accessor prof]     stage %-20s %8.3f ms (%llu)\n", xsi_hip_stage_name(i), a->ctx->stage_ms[i],
                        (unsigned long long)a->ctx->stage_n[i]);
                a->ctx->stage_ms[i] = 0;
                a->ctx->stage_n[i] = 0;
            }
        t_prof = now();
    };
    if (prof) a->ctx->timing = true;
    const uint8_t* img;
    uint64_t len, blk;
    int rc = accessor_block_image(a, block, &img, &len, &blk);
    if (rc) return rc;
    a->cur_block = -1;
    a->cur_entry = nullptr;
    rc = decode_prepare(a->ctx, img, len, blk, 1, &a->P);
    if (rc) return rc;
    lap("decode_prepare");
    a->cnt_block = -1;
    a->win_n = 0;
    const bool want_prefix = need_bin && need_bin < a->P.n_bin && a->P.n_wah >= 64u && decode_partial_supported(a->P) &&
                             !tuning_env("XSI_ACCESSOR_FULL_DECODE") && !accessor_sequential(a);  // (a scan reads every line anyway)
    if (want_prefix) {
        hipStream_t s = a->ctx->stream;
        std::vector<uint8_t> kind(a->P.n_bin);
        HIP_TRY(hipMemcpyAsync(kind.data(), a->P.L.kind, a->P.n_bin, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        std::vector<uint32_t> wah_before(a->P.n_bin + 1u);
        for (uint32_t i = 0; i < a->P.n_bin; ++i) wah_before[i + 1u] = wah_before[i] + ((kind[i] & KIND_WAH) ? 1u : 0u);
        const uint32_t target = prefix_target(wah_before, a->P.n_wah, 0u, need_bin);
        if (target < a->P.n_wah) {
            const size_t state_bytes = 4ull * (size_t)rank_decode_state_words(a->P.L.N, 1u) + 64u;  // + the side matrices' cursors
            void* st;
            rc = ws_ensure(a->ctx, "acc.rank_state", state_bytes, &st);
            if (rc) return rc;
            uint64_t* const d_walk = reinterpret_cast<uint64_t*>(static_cast<uint8_t*>(st) + state_bytes - 64u);
            PartialDecode pd{0u, target, static_cast<uint32_t*>(st), true, 0u, bin_valid_of(wah_before, a->P.n_wah, target), d_walk};
            rc = decode_all_planes(a->ctx, img, a->P, &a->D, &pd);
            if (rc) return rc;
            lap("prefix decode (planes + side)");
            a->biallelic = a->P.n_bin == a->P.n_bcf;
            PartialInfo pi;
            pi.wah_done = target;
            pi.ws_state = static_cast<const uint32_t*>(st);
            pi.state_bytes = state_bytes;
            pi.wah_before = &wah_before;
            if (accessor_cache_store(a, block, &pi)) {
                lap("cache store (malloc + copy)");
                a->cur_in_workspace = false;
                a->cur_block = (int64_t)block;
                ++a->prefix_decodes;
                return XSI_OK;
            }
            // the cache cannot hold the block: finish it in the workspace (a->P still describes this decode)
            PartialDecode rest{target, a->P.n_wah, static_cast<uint32_t*>(st), false, bin_valid_of(wah_before, a->P.n_wah, target),
                               a->P.n_bin, d_walk};
            rc = decode_all_planes(a->ctx, img, a->P, &a->D, &rest);
            if (rc) return rc;
            a->cur_in_workspace = true;
            a->cur_block = (int64_t)block;
            return XSI_OK;
        }
    }
    rc = decode_all_planes(a->ctx, img, a->P, &a->D);
    if (rc) return rc;
    lap("full decode (planes + side)");
    a->biallelic = a->P.n_bin == a->P.n_bcf;
    a->cur_in_workspace = !accessor_cache_store(a, block);
    lap("cache store (malloc + copy)");
    a->cur_block = (int64_t)block;
    return XSI_OK;
}

// compose `n` lines starting at binary line `first` (bi-allelic window) or one line with n_alleles
static int accessor_compose(xsi_accessor* a, uint32_t first, uint32_t n, uint32_t n_alleles) {
    hipStream_t s = a->ctx->stream;
    const uint32_t N = a->P.L.N;
    const uint32_t max_al = n_alleles;
    if ((uint64_t)n * max_al > a->counts_cap) {
        if (a->h_counts) (void)hipHostFree(a->h_counts);
        a->h_counts = nullptr;
        a->counts_cap = (uint32_t)((uint64_t)n * max_al + 64);
        HIP_TRY(hipHostMalloc((void**)&a->h_counts, 8ull * a->counts_cap, hipHostMallocDefault));
    }
    uint32_t* fb = a->h_meta;
    uint32_t* na = a->h_meta + a->win_rows;
    for (uint32_t i = 0; i < n; ++i) {
        fb[i] = first + i * (n_alleles - 1);
        na[i] = n_alleles;
    }
    // line metadata, per-line value counts and allele counts live in pinned host memory that the
    // kernel reads / writes directly (a few hundred bytes over PCIe): the only copy per call is the rows
    // A single line for a page-locked caller array: the compose kernel stores it there itself (the array's device
    // address; posted writes over PCIe) - one submission and one completion less than kernel + copy.
    const bool zero_copy = tuning_env("XSI_ACCESSOR_NO_ZEROCOPY") == nullptr;  // read per call: tests switch it
    const bool direct = n == 1u && a->direct_dst;  // only ever set for a registered array of at least N values
    const bool stores_direct = direct && zero_copy && a->reg_dev && a->direct_dst == a->reg_dst;
    int rc = compose_lines(a->ctx, a->P, a->D, a->h_meta, a->h_meta + a->win_rows, n, stores_direct ? a->reg_dev : a->d_rows, N,
                           a->h_meta + 2ull * a->win_rows, a->h_counts, max_al);
    if (rc) return rc;
    int32_t* const dst = direct ? a->direct_dst : a->h_rows;
    a->direct_done = direct;  // tells the caller where the line went
    a->win_in_rows = !a->direct_done;
    if (!stores_direct) HIP_TRY(hipMemcpyAsync(dst, a->d_rows, (size_t)n * N * sizeof(int32_t), hipMemcpyDeviceToHost, s));
    // latency path of a random-access query: poll instead of sleeping on the completion interrupt
    hipError_t q;
    while ((q = hipStreamQuery(s)) == hipErrorNotReady) {
    }
    HIP_TRY(q);
    return XSI_OK;
}

extern "C" {

uint32_t xsi_mac_threshold(uint32_t n_samples, uint32_t ploidy, double maf) {
    // gt_compressor_new.hpp:98-99: N_HAPS = n_samples * PLOIDY; (size_t)((double)N_HAPS * MAF)
    const double v = (double)((uint64_t)n_samples * ploidy) * maf;
    return v <= 0.0 ? 0u : (v >= 4294967295.0 ? 0xFFFFFFFFu : (uint32_t)v);
}

int32_t xsi_default_phased(const int32_t* const* h_gt_rows, const uint32_t* h_ngt, uint32_t n_rows, uint32_t n_samples) {
    // seek_default_phased, xcf.cpp:811-836: over the first records (the reference looks at 3), count the
    // phase bit of every sample's SECOND allele; a haploid record decides "unphased" at once; ties are phased
    if (!h_gt_rows || !h_ngt || !n_samples) return set_error(XSI_ERR_ARG, "default_phased: null argument");
    uint64_t counts[2] = {0, 0};
    for (uint32_t r = 0; r < n_rows; ++r) {
        const uint32_t ploidy = h_ngt[r] / n_samples;
        if (ploidy == 1) return 0;
        if (!h_gt_rows[r] || ploidy < 1) return set_error(XSI_ERR_ARG, "default_phased: bad row %u", r);
        for (uint32_t i = 0; i < n_samples; ++i) counts[h_gt_rows[r][(size_t)i * ploidy + 1] & 1]++;  // bcf_gt_is_phased
    }
    return counts[0] > counts[1] ? 0 : 1;
}

void xsi_bm_init(xsi_bm_state* st) {
    if (st) st->line = st->block = st->offset = 0;
}

int64_t xsi_bm_next(xsi_bm_state* st, uint32_t block_len, uint32_t n_allele) {
    // replace_samples_by_pos_in_binary_matrix, xcf.cpp:685-703 (new_version: blocks count BCF lines)
    if (!st || !block_len) return set_error(XSI_ERR_ARG, "bm_next: null state / zero block length");
    if (st->line && (st->line % block_len) == 0) {
        st->block++;
        st->offset = 0;
    }
    if (st->offset >> BM_BLOCK_BITS)
        return set_error(XSI_ERR_FORMAT, "Offset cannot be represented on %u bits !", BM_BLOCK_BITS);
    const int64_t bm = (int64_t)(int32_t)((uint32_t)st->block << BM_BLOCK_BITS | (uint32_t)st->offset);
    if (n_allele) st->offset += n_allele - 1;
    st->line++;
    return bm;
}

int64_t xsi_file_num_samples(const char* path) {
    if (!path) return set_error(XSI_ERR_ARG, "file_num_samples: null path");
    FILE* f = fopen(path, "rb");
    if (!f) return set_error(XSI_ERR_IO, "Failed to open file %s", path);
    uint8_t h[256];
    const size_t got = fread(h, 1, sizeof(h), f);
    fclose(f);
    auto get = [&](size_t off, int bytes) {
        uint64_t v = 0;
        for (int i = 0; i < bytes; ++i) v |= (uint64_t)h[off + i] << (8 * i);
        return v;
    };
    if (got != sizeof(h) || get(4, 4) != 0xfeed1767u || get(252, 4) != 0xfeed1767u) return set_error(XSI_ERR_FORMAT, "Bad magic");
    const uint64_t version = get(8, 4);
    if (version != 4 && version != 5) return set_error(XSI_ERR_FORMAT, "Bad version");
    if (h[12] == 0) return set_error(XSI_ERR_FORMAT, "PLOIDY ERROR");
    return (int64_t)(get(32, 8) / h[12]);  // hap_samples / ploidy names are stored (accessor.cpp:53-60)
}

int xsi_accessor_open(xsi_accessor** out, xsi_hip_ctx* ctx, const char* path) {
    if (!out || !ctx || !path) return set_error(XSI_ERR_ARG, "accessor_open: null argument");
    FILE* f = fopen(path, "rb");
    if (!f) return set_error(XSI_ERR_IO, "Failed to open file %s", path);
    fseek(f, 0, SEEK_END);
    const long sz = ftell(f);
    fseek(f, 0, SEEK_SET);
    if (sz < 256) {
        fclose(f);
        return set_error(XSI_ERR_FORMAT, "Bad magic");
    }
    xsi_accessor* a = new (std::nothrow) xsi_accessor();
    if (a) {
        try {
            a->file.resize((size_t)sz);
        } catch (const std::exception&) {
            delete a;
            a = nullptr;
        }
    }
    if (!a) {
        fclose(f);
        return set_error(XSI_ERR_IO, "accessor_open: cannot hold %ld bytes of %s in memory", sz, path);
    }
    if (fread(a->file.data(), 1, (size_t)sz, f) != (size_t)sz) {
        fclose(f);
        delete a;
        return set_error(XSI_ERR_IO, "short read of %s", path);
    }
    fclose(f);
    const uint8_t* h = a->file.data();
    auto get = [&](size_t off, int bytes) {
        uint64_t v = 0;
        for (int i = 0; i < bytes; ++i) v |= (uint64_t)h[off + i] << (8 * i);
        return v;
    };
    // accessor.cpp:36-51
    if (get(4, 4) != 0xfeed1767u || get(252, 4) != 0xfeed1767u) {
        delete a;
        return set_error(XSI_ERR_FORMAT, "Bad magic");
    }
    a->version = (uint32_t)get(8, 4);
    if (a->version != 4 && a->version != 5) {
        delete a;
        return set_error(XSI_ERR_FORMAT, a->version == 2 || a->version == 3 ? "Unsupported version" : "Bad version");
    }
    a->ploidy = h[12];
    a->aet = h[14];
    a->hap_samples = get(32, 8);
    a->num_samples = get(112, 8);
    if (a->aet != 2 && a->aet != 4) {
        delete a;
        return set_error(XSI_ERR_FORMAT, "Unsupported A_T");
    }
    if (a->ploidy == 0) {
        delete a;
        return set_error(XSI_ERR_FORMAT, "PLOIDY ERROR");
    }
    a->zstd = (h[17] & 4u) != 0;
    if (a->zstd && !zstd_api().ok) {
        delete a;
        return set_error(XSI_ERR_UNSUPPORTED, "zstd-compressed file but libzstd.so.1 could not be loaded");
    }
    const uint64_t io = get(72, 8), so = get(80, 8);
    if (io > (uint64_t)sz || so > (uint64_t)sz || so < io) {
        delete a;
        return set_error(XSI_ERR_FORMAT, "index outside the file");
    }
    a->n_blocks = (so - io) / (a->version >= 5 ? 8 : 4);
    // sample list, accessor.cpp:53-60
    {
        const uint64_t want = a->hap_samples / a->ploidy;
        size_t pos = (size_t)so;
        while (pos < (size_t)sz && a->names.size() < want) {
            const void* z = memchr(h + pos, 0, (size_t)sz - pos);
            if (!z) break;
            a->names.emplace_back((const char*)h + pos);
            pos = (const uint8_t*)z - h + 1;
        }
    }
    int rc = xsi_hip_ctx_create(&a->ctx, ctx->device, nullptr);
    if (rc) {
        delete a;
        return rc;
    }
    hipError_t e = hipSuccess;
    if (!a->zstd) {
        e = hipMalloc((void**)&a->d_file, (size_t)sz);
        if (e == hipSuccess) e = hipMemcpy(a->d_file, h, (size_t)sz, hipMemcpyHostToDevice);
    }
    const uint64_t N = a->num_samples ? a->num_samples * 2 : a->hap_samples;
    a->n_full = N;
    // window of composed rows: <= 64 MiB of int32
    uint64_t win = N ? (64ull << 20) / (N * 4) : 1;
    if (win < 1) win = 1;
    if (win > MAX_BIN_PER_BLOCK) win = MAX_BIN_PER_BLOCK;
    a->win_rows = (uint32_t)win;
    if (e == hipSuccess) e = hipMalloc((void**)&a->d_rows, (size_t)win * N * 4);
    if (e == hipSuccess) e = hipHostMalloc((void**)&a->h_rows, (size_t)win * N * 4, hipHostMallocDefault);
    if (e == hipSuccess) e = hipHostMalloc((void**)&a->h_meta, 12ull * win, hipHostMallocDefault);
    if (e != hipSuccess) {
        xsi_accessor_close(a);
        return set_error(XSI_ERR_HIP, "accessor buffers: %s", hipGetErrorString(e));
    }
    {
        // decoded-block cache: half of the free HBM, at most 64 GiB, unless XSI_ACCESSOR_CACHE_MB says otherwise
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) free_b = 0;
        a->cache_budget = free_b / 2;
        if (a->cache_budget > (64ull << 30)) a->cache_budget = 64ull << 30;
        if (const char* e = tuning_env("XSI_ACCESSOR_CACHE_MB")) a->cache_budget = (size_t)strtoull(e, nullptr, 10) << 20;
    }
    *out = a;
    return XSI_OK;
}

int xsi_accessor_set_cache_bytes(xsi_accessor* a, uint64_t bytes) {
    if (!a) return set_error(XSI_ERR_ARG, "set_cache_bytes: null accessor");
    accessor_readahead_join(a);
    std::lock_guard<std::mutex> lk(a->cache_m);
    a->cache_budget = (size_t)bytes;
    while (a->cache_bytes > a->cache_budget && !a->cache.empty()) {
        CacheIt lru = accessor_lru(a, -1);
        if (a->cur_block >= 0 && !a->cur_in_workspace && lru->block == (uint64_t)a->cur_block) {
            a->cur_block = -1;
            a->cur_entry = nullptr;
        }
        accessor_evict(a, lru);
    }
    a->pf_block = -1;  // (what was read ahead may be gone)
    return XSI_OK;
}

int xsi_accessor_cache_stats(const xsi_accessor* a, uint64_t* blocks, uint64_t* bytes, uint64_t* hits, uint64_t* misses) {
    if (!a) return set_error(XSI_ERR_ARG, "cache_stats: null accessor");
    std::lock_guard<std::mutex> lk(const_cast<xsi_accessor*>(a)->cache_m);
    if (blocks) *blocks = a->cache.size();
    if (bytes) *bytes = a->cache_bytes;
    if (hits) *hits = a->cache_hits;
    if (misses) *misses = a->cache_misses;
    return XSI_OK;
}

int xsi_accessor_readahead_stats(const xsi_accessor* a, uint64_t* started, uint64_t* hits) {
    if (!a) return set_error(XSI_ERR_ARG, "readahead_stats: null accessor");
    if (started) *started = a->readahead_started;
    if (hits) *hits = a->readahead_hits;
    return XSI_OK;
}

int xsi_accessor_prefix_stats(const xsi_accessor* a, uint64_t* prefix_decodes, uint64_t* extensions) {
    if (!a) return set_error(XSI_ERR_ARG, "prefix_stats: null accessor");
    if (prefix_decodes) *prefix_decodes = a->prefix_decodes;
    if (extensions) *extensions = a->prefix_extensions;
    return XSI_OK;
}

// the line's values in the accessor's pinned window (valid until the next call on the accessor) and their number
static int64_t accessor_line_view(xsi_accessor* a, uint32_t n_alleles, uint64_t position, const int32_t** view);

int64_t xsi_accessor_fill_genotype_array(xsi_accessor* a, int32_t* h_gt, uint64_t gt_size, uint32_t n_alleles,
                                         uint64_t position) {
    if (!a || !h_gt) return set_error(XSI_ERR_ARG, "fill_genotype_array: null argument");
    a->direct_dst = nullptr;
    a->direct_done = false;
    // The composed row is n_full = 2 * num_samples values wide whatever the line's ploidy, so the device may only
    // write into the caller's array when it holds that many (a haploid file's hap_samples is half of it) and the
    // caller has registered it (xsi_accessor_register_array); every other array is filled by the memcpy below,
    // behind the capacity check.
    if (a->reg_dst && h_gt == a->reg_dst && gt_size >= a->n_full && a->reg_bytes >= a->n_full * sizeof(int32_t)) a->direct_dst = h_gt;
    const int32_t* view = nullptr;
    const int64_t ngt = accessor_line_view(a, n_alleles, position, &view);
    if (ngt < 0) return ngt;
    if (gt_size < (uint64_t)ngt) return set_error(XSI_ERR_CAPACITY, "gt array holds %llu values, line has %lld", (unsigned long long)gt_size, (long long)ngt);
    if (view != h_gt) memcpy(h_gt, view, (size_t)ngt * sizeof(int32_t));
    return ngt;
}

static void accessor_drop_registration(xsi_accessor* a) {
    a->reg_dst = nullptr;
    a->reg_dev = nullptr;
    a->reg_bytes = 0;
}

// Is [p, p + bytes) page-locked memory the device can address (hipHostMalloc / the caller's own hipHostRegister)?
// Every page is asked about, not only the two ends: two pinned allocations with pageable or unmapped memory between
// them would pass an end-point test, and the compose kernels store straight through the mapping.  Done once
// per registration (a 200 MB array: 50 000 queries, tens of milliseconds).
static bool page_locked_range(const void* p, size_t bytes) {
    hipPointerAttribute_t at;
    const uintptr_t b = reinterpret_cast<uintptr_t>(p), e = b + (bytes ? bytes - 1 : 0);
    for (uintptr_t q = b;; q = (q | 4095u) + 1u) {
        if (q > e) q = e;
        if (hipPointerGetAttributes(&at, reinterpret_cast<const void*>(q)) != hipSuccess) {
            (void)hipGetLastError();
            return false;
        }
        if (at.type != hipMemoryTypeHost) return false;
        if (q == e) break;
    }
    return true;
}

int xsi_accessor_alloc_array(xsi_accessor* a, uint64_t n_values, int32_t** h_gt) {
    if (!a || !h_gt) return set_error(XSI_ERR_ARG, "alloc_array: null argument");
    if (n_values < a->n_full)
        return set_error(XSI_ERR_CAPACITY, "alloc_array: %llu values asked for, a composed row has %llu",
                         (unsigned long long)n_values, (unsigned long long)a->n_full);
    void* p = nullptr;
    HIP_TRY(hipHostMalloc(&p, (size_t)n_values * sizeof(int32_t), hipHostMallocDefault));
    a->owned_arrays.push_back({p, (size_t)n_values * sizeof(int32_t)});
    *h_gt = static_cast<int32_t*>(p);
    return XSI_OK;
}

int xsi_accessor_free_array(xsi_accessor* a, int32_t* h_gt) {
    if (!a || !h_gt) return set_error(XSI_ERR_ARG, "free_array: null argument");
    auto it = std::find_if(a->owned_arrays.begin(), a->owned_arrays.end(), [&](const std::pair<void*, size_t>& e) { return e.first == h_gt; });
    if (it == a->owned_arrays.end()) return set_error(XSI_ERR_ARG, "free_array: not an array of xsi_accessor_alloc_array");
    if (a->ctx) HIP_TRY(hipStreamSynchronize(a->ctx->stream));
    const uint8_t* b = reinterpret_cast<const uint8_t*>(h_gt);
    const uint8_t* r = reinterpret_cast<const uint8_t*>(a->reg_dst);
    if (a->reg_dst && r >= b && r < b + it->second) accessor_drop_registration(a);  // the registered array lies in this allocation: gone with it
    a->owned_arrays.erase(it);
    HIP_TRY(hipHostFree(h_gt));
    return XSI_OK;
}

int xsi_accessor_register_array(xsi_accessor* a, int32_t* h_gt, uint64_t n_values) {
    if (!a || !h_gt) return set_error(XSI_ERR_ARG, "register_array: null argument");
    if (n_values < a->n_full)
        return set_error(XSI_ERR_CAPACITY, "register_array: the array holds %llu values, a composed row has %llu",
                         (unsigned long long)n_values, (unsigned long long)a->n_full);
    if (a->ctx) HIP_TRY(hipStreamSynchronize(a->ctx->stream));
    accessor_drop_registration(a);
    if (tuning_env("XSI_ACCESSOR_NO_REGISTER")) return XSI_OK;  // measurement: every line through the pinned window + memcpy
    const size_t bytes = (size_t)n_values * sizeof(int32_t);  // the whole array: a batch fills many rows of it
    // Only memory that IS page-locked is taken - xsi_accessor_alloc_array's, or an allocation the caller page-locked
    // itself.  The accessor does not hipHostRegister pageable caller memory any more: on this runtime the unregister
    // of a range that is not page-aligned takes the device's access to the pages next to it away as well, pages the
    // runtime itself may hold pinned for some other host allocation (its cache of pins made for asynchronous copies from
    // and to pageable memory) - the next such copy then faults on the device, in a call that has nothing to do with
    // this accessor (found as an intermittent "Memory access fault by GPU" one page behind a formerly registered array).
    if (!page_locked_range(h_gt, bytes))
        return set_error(XSI_ERR_ARG, "register_array: the array is pageable memory; take it from xsi_accessor_alloc_array "
                                      "(or page-lock the whole allocation yourself) - see include/xsi_hip.h");
    a->reg_dst = h_gt;
    a->reg_bytes = bytes;
    void* dev = nullptr;
    if (hipHostGetDevicePointer(&dev, h_gt, 0) == hipSuccess) a->reg_dev = static_cast<int32_t*>(dev);
    else (void)hipGetLastError();  // the copy engine still fills the page-locked array directly
    return XSI_OK;
}

int xsi_accessor_unregister_array(xsi_accessor* a) {
    if (!a) return set_error(XSI_ERR_ARG, "unregister_array: null accessor");
    if (a->ctx) HIP_TRY(hipStreamSynchronize(a->ctx->stream));
    accessor_drop_registration(a);
    return XSI_OK;
}

int64_t xsi_accessor_genotypes_view(xsi_accessor* a, uint32_t n_alleles, uint64_t position, const int32_t** h_gt) {
    if (!a || !h_gt) return set_error(XSI_ERR_ARG, "genotypes_view: null argument");
    a->direct_dst = nullptr;
    a->direct_done = false;
    return accessor_line_view(a, n_alleles, position, h_gt);
}

static int64_t accessor_line_view(xsi_accessor* a, uint32_t n_alleles, uint64_t position, const int32_t** view) {
    if (n_alleles < 2) return set_error(XSI_ERR_ARG, "fill_genotype_array: n_alleles < 2");
    // AccessorInternalsNewTemplate::seek, accessor_internals_new.hpp:722-738
    const uint64_t block = (position & 0xFFFFFFFFull) >> BM_BLOCK_BITS;
    const uint32_t offset = (uint32_t)(position & ((1u << BM_BLOCK_BITS) - 1u));
    // does this query follow the one before, line by line (the next binary line of the same block, or the first line of the
    // next block)?  Counted before the block is loaded: the count decides how it is decoded.
    if ((a->seq_block == (int64_t)block && offset == a->seq_next) || (a->seq_block + 1 == (int64_t)block && offset == 0u && a->seq_block >= 0)) {
        if (a->seq_run < 0x7FFFFFFFu) ++a->seq_run;
    } else {
        a->seq_run = 0;
    }
    a->seq_block = (int64_t)block;
    a->seq_next = offset + (n_alleles - 1u);
    if (a->cur_block < 0 || (uint64_t)a->cur_block != block) {
        int rc = accessor_load_block(a, block, offset + (n_alleles - 1u));
        if (rc) return rc;
    }
    if (!a->cur_in_workspace) accessor_readahead_maybe(a, block);  // (a no-op unless the scan is sequential and block + 1 is cold)
    if (offset + (n_alleles - 1) > a->P.n_bin)
        return set_error(XSI_ERR_ARG, "position offset %u (+%u alleles) beyond the %u binary lines of block %llu", offset,
                         n_alleles - 1, a->P.n_bin, (unsigned long long)block);
    const uint32_t N = a->P.L.N;
    uint32_t row = 0;
    {   // a prefix-decoded block: the chain runs on to the lines this call reads (no-op for a complete block)
        int rc = accessor_ensure_lines(a, offset + (n_alleles - 1u));
        if (rc) return rc;
    }
    if (a->biallelic && n_alleles == 2) {
        if (!(a->win_n && a->win_in_rows && offset >= a->win_first && offset < a->win_first + a->win_n)) {
            // window length follows the access pattern: a request that continues the previous window
            // doubles it (sequential scan -> few large compose + copy steps), a jump resets it to one
            // line (random access -> no composing / copying of lines nobody asked for)
            const bool sequential = a->win_n && a->win_block == (int64_t)block && offset == a->win_first + a->win_n;
            a->win_target = sequential ? (a->win_target * 2u > a->win_rows ? a->win_rows : a->win_target * 2u) : 1u;
            a->win_block = (int64_t)block;
            uint32_t n = a->P.n_bin - offset;
            if (n > a->win_target) n = a->win_target;
            int rc = accessor_ensure_lines(a, offset + n);
            if (rc) return rc;
            rc = accessor_compose(a, offset, n, 2);
            if (rc) return rc;
            a->win_first = offset;
            a->win_n = n;
        }
        row = offset - a->win_first;
    } else {
        int rc = accessor_compose(a, offset, 1, n_alleles);
        if (rc) return rc;
        a->win_n = 0;
        row = 0;
    }
    const uint32_t ngt = a->h_meta[2ull * a->win_rows + row];
    *view = a->direct_done ? a->direct_dst : a->h_rows + (size_t)row * N;
    a->last_counts.assign(a->h_counts + (size_t)row * n_alleles, a->h_counts + (size_t)(row + 1) * n_alleles);
    return (int64_t)ngt;
}

int64_t xsi_accessor_get_genotypes(xsi_accessor* a, uint32_t n_alleles, uint64_t position, void** h_gt, int* ngt_arr) {
    if (!a || !h_gt || !ngt_arr) return set_error(XSI_ERR_ARG, "get_genotypes: null argument");
    const uint64_t ngt = a->hap_samples;  // accessor.hpp:59
    if (!*h_gt) {
        *h_gt = malloc(sizeof(int) * (ngt ? ngt : 1));
        if (!*h_gt) return set_error(XSI_ERR_ARG, "malloc failed");
    }
    *ngt_arr = (int)ngt;
    return xsi_accessor_fill_genotype_array(a, (int32_t*)*h_gt, ngt, n_alleles, position);
}

int64_t xsi_accessor_get_genotypes_batch(xsi_accessor* a, uint64_t n, const uint32_t* n_alleles, const uint64_t* positions,
                                         int32_t* h_rows, uint64_t row_stride, uint32_t* h_ngt) {
    if (!a || !n_alleles || !positions || !h_rows) return set_error(XSI_ERR_ARG, "get_genotypes_batch: null argument");
    if (row_stride < a->n_full)
        return set_error(XSI_ERR_CAPACITY, "get_genotypes_batch: row_stride %llu < %llu values of a composed row",
                         (unsigned long long)row_stride, (unsigned long long)a->n_full);
    if (!n) return 0;
    hipStream_t s = a->ctx->stream;
    const uint64_t N = a->n_full;
    a->direct_dst = nullptr;
    a->direct_done = false;
    a->win_n = 0;  // the single-line window does not survive a batch
    // Destination: rows inside the registered array are stored there by the compose kernels themselves (posted PCIe
    // writes, one completion per chunk); any other memory goes through the device window and one copy per chunk.
    const bool zero_copy = tuning_env("XSI_ACCESSOR_NO_ZEROCOPY") == nullptr;
    const uint8_t* rb = reinterpret_cast<const uint8_t*>(a->reg_dst);
    const uint8_t* hb = reinterpret_cast<const uint8_t*>(h_rows);
    const bool direct = zero_copy && a->reg_dst && a->reg_dev && hb >= rb &&
                        hb + ((n - 1u) * row_stride + N) * sizeof(int32_t) <= rb + a->reg_bytes;
    const uint32_t chunk_cap = direct ? 4096u : a->win_rows;
    if (a->bmeta_cap < chunk_cap) {
        if (a->h_bmeta) (void)hipHostFree(a->h_bmeta);
        a->h_bmeta = nullptr;
        a->bmeta_cap = 0;
        HIP_TRY(hipHostMalloc((void**)&a->h_bmeta, 16ull * chunk_cap, hipHostMallocDefault));
        a->bmeta_cap = chunk_cap;
    }
    uint32_t* fb = a->h_bmeta;
    uint32_t* na = fb + a->bmeta_cap;
    uint32_t* oi = na + a->bmeta_cap;
    uint32_t* ng = oi + a->bmeta_cap;
    std::vector<uint32_t> order;
    int64_t total = 0;
    // an error inside a chunk leaves compose kernels of its earlier groups in flight, storing into the caller's array
    // (direct path): they are waited for before the call returns, whichever exit it takes - the `return rc` paths and the
    // HIP_TRY ones alike: the guard synchronises the stream on every exit but the last
    StreamSyncGuard in_flight(s);
    auto bail = [&](int rc) { return rc; };
    for (uint64_t c0 = 0; c0 < n; c0 += chunk_cap) {
        const uint32_t m = (uint32_t)(n - c0 < chunk_cap ? n - c0 : chunk_cap);
        // queries of a chunk grouped by block (stable): one compose launch per block touched
        order.resize(m);
        for (uint32_t i = 0; i < m; ++i) order[i] = i;
        std::stable_sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) {
            return ((positions[c0 + x] & 0xFFFFFFFFull) >> BM_BLOCK_BITS) < ((positions[c0 + y] & 0xFFFFFFFFull) >> BM_BLOCK_BITS);
        });
        int32_t* const d_dst = direct ? a->reg_dev + ((h_rows - static_cast<int32_t*>(a->reg_dst)) + (ptrdiff_t)(c0 * row_stride)) : a->d_rows;
        const uint64_t d_stride = direct ? row_stride : N;
        for (uint32_t g0 = 0; g0 < m;) {
            const uint64_t block = (positions[c0 + order[g0]] & 0xFFFFFFFFull) >> BM_BLOCK_BITS;
            uint32_t g1 = g0;
            while (g1 < m && ((positions[c0 + order[g1]] & 0xFFFFFFFFull) >> BM_BLOCK_BITS) == block) ++g1;
            uint32_t need = 0;  // binary lines of this block the group reads: what a cold block is decoded up to
            for (uint32_t g = g0; g < g1; ++g) {
                const uint64_t q = c0 + order[g];
                const uint32_t end = (uint32_t)(positions[q] & ((1u << BM_BLOCK_BITS) - 1u)) + (n_alleles[q] > 1u ? n_alleles[q] - 1u : 1u);
                if (end > need) need = end;
            }
            if (a->cur_block < 0 || (uint64_t)a->cur_block != block) {
                // a block that lives in the context workspace (cache too small) is overwritten by the next decode:
                // the composes that read it must have finished
                if (a->cur_in_workspace) HIP_TRY(hipStreamSynchronize(s));
                int rc = accessor_load_block(a, block, need);
                if (rc) return bail(rc);
            }
            for (uint32_t g = g0; g < g1; ++g) {
                const uint64_t q = c0 + order[g];
                const uint32_t offset = (uint32_t)(positions[q] & ((1u << BM_BLOCK_BITS) - 1u));
                if (n_alleles[q] < 2u) return bail(set_error(XSI_ERR_ARG, "get_genotypes_batch: query %llu: n_alleles < 2", (unsigned long long)q));
                if (offset + (n_alleles[q] - 1u) > a->P.n_bin)
                    return bail(set_error(XSI_ERR_ARG, "get_genotypes_batch: query %llu: offset %u (+%u alleles) beyond the %u binary lines of block %llu",
                                          (unsigned long long)q, offset, n_alleles[q] - 1u, a->P.n_bin, (unsigned long long)block));
                fb[g] = offset;
                na[g] = n_alleles[q];
                oi[g] = order[g];
            }
            int rc = accessor_ensure_lines(a, need);
            if (rc) return bail(rc);
            rc = compose_lines(a->ctx, a->P, a->D, fb + g0, na + g0, g1 - g0, d_dst, d_stride, ng, nullptr, 0, oi + g0);
            if (rc) return bail(rc);
            g0 = g1;
        }
        if (!direct) {
            if (row_stride == N)
                HIP_TRY(hipMemcpyAsync(h_rows + c0 * row_stride, a->d_rows, (size_t)m * N * sizeof(int32_t), hipMemcpyDeviceToHost, s));
            else
                HIP_TRY(hipMemcpy2DAsync(h_rows + c0 * row_stride, row_stride * sizeof(int32_t), a->d_rows, N * sizeof(int32_t),
                                         N * sizeof(int32_t), m, hipMemcpyDeviceToHost, s));
        }
        // one completion per chunk: the pinned metadata is reused by the next chunk, and the rows must have landed (a chunk
        // is milliseconds of work: a blocking wait, not the single query's poll loop, which holds a core)
        HIP_TRY(hipStreamSynchronize(s));
        for (uint32_t i = 0; i < m; ++i) {
            if (h_ngt) h_ngt[c0 + i] = ng[i];
            total += ng[i];
        }
    }
    in_flight.release();  // the last chunk's synchronisation stands above
    return total;
}

int xsi_accessor_set_sample_subset(xsi_accessor* a, const uint32_t* sample_idx, uint32_t n) {
    if (!a || (n && !sample_idx)) return set_error(XSI_ERR_ARG, "set_sample_subset: null argument");
    const uint64_t ns = a->num_samples ? a->num_samples : a->hap_samples / (a->ploidy ? a->ploidy : 1);
    for (uint32_t i = 0; i < n; ++i)
        if (sample_idx[i] >= ns) return set_error(XSI_ERR_ARG, "set_sample_subset: sample %u of %llu", sample_idx[i], (unsigned long long)ns);
    HIP_TRY(hipStreamSynchronize(a->ctx->stream));
    if (a->d_sel) (void)hipFree(a->d_sel);
    if (a->d_sel_row) (void)hipFree(a->d_sel_row);
    if (a->h_sel_row) (void)hipHostFree(a->h_sel_row);
    a->d_sel = nullptr;
    a->d_sel_row = nullptr;
    a->h_sel_row = nullptr;
    a->n_sel = n;
    if (!n) return XSI_OK;
    HIP_TRY(hipMalloc((void**)&a->d_sel, 4ull * n));
    HIP_TRY(hipMalloc((void**)&a->d_sel_row, 8ull * n));
    HIP_TRY(hipHostMalloc((void**)&a->h_sel_row, 8ull * n, hipHostMallocDefault));
    if (!a->h_sel_ac) HIP_TRY(hipHostMalloc((void**)&a->h_sel_ac, 4ull * 32, hipHostMallocDefault));
    HIP_TRY(hipMemcpy(a->d_sel, sample_idx, 4ull * n, hipMemcpyHostToDevice));
    return XSI_OK;
}

int64_t xsi_accessor_fill_selected_genotypes(xsi_accessor* a, int32_t* h_gt, uint64_t gt_size, uint32_t n_alleles,
                                             uint64_t position, int32_t* h_ac) {
    if (!a || !h_gt) return set_error(XSI_ERR_ARG, "fill_selected_genotypes: null argument");
    if (!a->n_sel) return set_error(XSI_ERR_ARG, "fill_selected_genotypes: no sample subset set");
    if (n_alleles < 2 || n_alleles > 33) return set_error(XSI_ERR_ARG, "fill_selected_genotypes: n_alleles %u not in 2..33", n_alleles);
    const uint64_t block = (position & 0xFFFFFFFFull) >> BM_BLOCK_BITS;
    const uint32_t offset = (uint32_t)(position & ((1u << BM_BLOCK_BITS) - 1u));
    if (a->cur_block < 0 || (uint64_t)a->cur_block != block) {
        int rc = accessor_load_block(a, block, offset + (n_alleles - 1u));
        if (rc) return rc;
    }
    if (offset + (n_alleles - 1) > a->P.n_bin)
        return set_error(XSI_ERR_ARG, "position offset %u (+%u alleles) beyond the %u binary lines of block %llu", offset,
                         n_alleles - 1, a->P.n_bin, (unsigned long long)block);
    {
        int rc = accessor_ensure_lines(a, offset + (n_alleles - 1u));
        if (rc) return rc;
    }
    hipStream_t s = a->ctx->stream;
    // compose the full line on the device (no copy to the host), gather the selected samples there
    if ((uint64_t)n_alleles > a->counts_cap) {
        if (a->h_counts) (void)hipHostFree(a->h_counts);
        a->h_counts = nullptr;
        a->counts_cap = n_alleles + 64;
        HIP_TRY(hipHostMalloc((void**)&a->h_counts, 8ull * a->counts_cap, hipHostMallocDefault));
    }
    a->h_meta[0] = offset;
    a->h_meta[a->win_rows] = n_alleles;
    a->win_n = 0;
    const uint32_t N = a->P.L.N;
    int rc = compose_lines(a->ctx, a->P, a->D, a->h_meta, a->h_meta + a->win_rows, 1, a->d_rows, N,
                           a->h_meta + 2ull * a->win_rows, a->h_counts, n_alleles);
    if (rc) return rc;
    rc = select_samples(a->ctx, a->d_rows, N, a->h_meta + 2ull * a->win_rows, 1, a->P.L.n_samples, a->d_sel, a->n_sel,
                        a->d_sel_row, 2ull * a->n_sel, a->h_sel_ac, n_alleles - 1);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(a->h_sel_row, a->d_sel_row, 8ull * a->n_sel, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    const uint32_t ploidy = a->h_meta[2ull * a->win_rows] / a->P.L.n_samples;
    const uint64_t an = (uint64_t)a->n_sel * ploidy;
    if (gt_size < an) return set_error(XSI_ERR_CAPACITY, "gt array holds %llu values, selection has %llu", (unsigned long long)gt_size,
                                       (unsigned long long)an);
    memcpy(h_gt, a->h_sel_row, an * sizeof(int32_t));
    if (h_ac)
        for (uint32_t k = 0; k + 1 < n_alleles; ++k) h_ac[k] = (int32_t)a->h_sel_ac[k];
    return (int64_t)an;
}

int xsi_accessor_fill_allele_counts(xsi_accessor* a, uint32_t n_alleles, uint64_t position) {
    if (!a) return set_error(XSI_ERR_ARG, "fill_allele_counts: null accessor");
    if (n_alleles < 2) return set_error(XSI_ERR_ARG, "fill_allele_counts: n_alleles < 2");
    const uint64_t block = (position & 0xFFFFFFFFull) >> BM_BLOCK_BITS;
    const uint32_t offset = (uint32_t)(position & ((1u << BM_BLOCK_BITS) - 1u));
    if (a->cnt_block < 0 || (uint64_t)a->cnt_block != block) {
        DecodePlan P;
        const uint8_t* img;
        uint64_t len, blk;
        int rc = accessor_block_image(a, block, &img, &len, &blk);
        if (rc) return rc;
        if (a->cur_in_workspace) {
            // decode_prepare reuses the context workspace the current block's planes live in: drop that
            // view before anything is overwritten, so that an error below cannot leave it half valid
            a->cur_block = -1;
            a->win_n = 0;
        }
        a->cnt_block = -1;
        rc = decode_prepare(a->ctx, img, len, blk, 1, &P);
        if (rc) return rc;
        rc = decode_counts_only(a->ctx, img, P);
        if (rc) return rc;
        a->cnt_ones.resize(P.n_bin);
        a->cnt_kind.resize(P.n_bin);
        HIP_TRY(hipMemcpyAsync(a->cnt_ones.data(), P.L.ones, 4ull * P.n_bin, hipMemcpyDeviceToHost, a->ctx->stream));
        HIP_TRY(hipMemcpyAsync(a->cnt_kind.data(), P.L.kind, P.n_bin, hipMemcpyDeviceToHost, a->ctx->stream));
        HIP_TRY(hipStreamSynchronize(a->ctx->stream));
        a->cnt_block = (int64_t)block;
    }
    if ((size_t)offset + (n_alleles - 1) > a->cnt_ones.size())
        return set_error(XSI_ERR_ARG, "position offset %u (+%u alleles) beyond the block", offset, n_alleles - 1);
    const uint64_t N = a->num_samples ? a->num_samples * 2 : a->hap_samples;
    const uint64_t nl = (a->cnt_kind[offset] & KIND_HAPLOID) ? N / 2 : N;
    a->last_counts.assign(n_alleles, 0);
    uint64_t total = 0;
    for (uint32_t k = 1; k < n_alleles; ++k) {
        a->last_counts[k] = a->cnt_ones[offset + k - 1];
        total += a->last_counts[k];
    }
    a->last_counts[0] = nl - total;  // sic: missing / end-of-vector not subtracted (:437)
    return XSI_OK;
}

}  // extern "C"

// The PBWT arrangement after the WAH lines [0, n_before) of a decoded block: a_{k+1} = zeros of line k in a_k's
// order, then its ones (pbwt_sort, internal_gt_record.hpp:32-59).  One workgroup; thread t owns positions
// [t K, t K + K): it counts the zeros among them, a workgroup scan places its segment, and it writes its members
// to their new positions.  This is the replay the reference's seek does on the host; nothing on the hot path
// holds `a` (the chains track ranks), so it is built only when get_internal_access asks for it.
__global__ void __launch_bounds__(1024) k_arrangement_at(const uint32_t* __restrict__ planes, uint32_t stride_w,
                                                         const uint8_t* __restrict__ kind, uint32_t n_before, uint32_t N,
                                                         uint32_t* a0, uint32_t* a1, uint32_t* which) {
    __shared__ uint64_t scan_lds[17];
    const uint32_t tid = threadIdx.x;
    const uint32_t K = (N + 1023u) / 1024u;
    const uint32_t lo = tid * K < N ? tid * K : N, hi = lo + K < N ? lo + K : N;
    for (uint32_t i = lo; i < hi; ++i) a0[i] = i;
    __syncthreads();
    uint32_t *cur = a0, *nxt = a1;
    for (uint32_t l = 0; l < n_before; ++l) {
        if (!(kind[l] & KIND_WAH)) continue;  // sparse lines never touch a (gt_block.hpp:299-326)
        const uint32_t* row = planes + (size_t)l * stride_w;
        // a fully haploid line has one bit per SAMPLE: haplotype h goes by its sample's bit (pbwt_sort1,
        // internal_gt_record.hpp:50-59; PBWTSorter::bool_pbwt_sort_two, gt_block.hpp:137-151)
        const uint32_t hs = (kind[l] & KIND_HAPLOID) ? 1u : 0u;
        uint32_t zc = 0;
        for (uint32_t i = lo; i < hi; ++i) {
            const uint32_t h = cur[i] >> hs;
            zc += 1u - ((row[h >> 5] >> (h & 31u)) & 1u);
        }
        uint64_t Z;
        const uint32_t zbase = (uint32_t)block_scan_excl64(zc, scan_lds, &Z);
        uint32_t zpos = zbase, opos = (uint32_t)Z + (lo - zbase);
        for (uint32_t i = lo; i < hi; ++i) {
            const uint32_t h = cur[i], k = h >> hs;
            if ((row[k >> 5] >> (k & 31u)) & 1u) nxt[opos++] = h;
            else nxt[zpos++] = h;
        }
        __threadfence_block();
        __syncthreads();
        uint32_t* t = cur;
        cur = nxt;
        nxt = t;
    }
    if (tid == 0) *which = cur == a0 ? 0u : 1u;
}

extern "C" {

int xsi_accessor_get_internal_access(xsi_accessor* a, uint32_t n_alleles, uint64_t position, xsi_internal_access* info,
                                     uint8_t* h_sparse, uint64_t* h_offsets, uint32_t* h_a) {
    if (!a || !info) return set_error(XSI_ERR_ARG, "get_internal_access: null argument");
    if (n_alleles >= 2 && (!h_sparse || !h_offsets)) return set_error(XSI_ERR_ARG, "get_internal_access: null output array");
    const uint64_t block = (position & 0xFFFFFFFFull) >> BM_BLOCK_BITS;
    const uint32_t offset = (uint32_t)(position & ((1u << BM_BLOCK_BITS) - 1u));
    memset(info, 0, sizeof(*info));
    info->position = position;
    info->n_alleles = n_alleles;
    info->sparse_bytes = a->aet;
    info->wah_bytes = 2;
    info->a_bytes = 4;
    info->n_a = (uint32_t)a->hap_samples;
    if (n_alleles < 2) return XSI_OK;  // the reference returns the bare header for n_alleles == 0
    hipStream_t s = a->ctx->stream;
    // the arrangement first: it needs the decoded lines of the block (cache or workspace) up to the record's last line
    if (a->cur_block < 0 || (uint64_t)a->cur_block != block) {
        int rc = accessor_load_block(a, block, offset + (n_alleles - 1u));
        if (rc) return rc;
    }
    {
        int rc = accessor_ensure_lines(a, offset + (n_alleles - 1u));
        if (rc) return rc;
    }
    const uint32_t n_lines = n_alleles - 1u;
    if (offset + n_lines > a->P.n_bin)
        return set_error(XSI_ERR_ARG, "position offset %u (+%u alleles) beyond the %u binary lines of block %llu", offset, n_lines,
                         a->P.n_bin, (unsigned long long)block);
    if (a->P.blocks_h.empty()) return set_error(XSI_ERR_FORMAT, "get_internal_access: block %llu not decoded", (unsigned long long)block);
    if (h_a) {
        const uint32_t N = a->P.L.N;
        uint32_t *d_a0, *d_a1, *d_which;
        xsi_hip_ctx* ctx = a->ctx;
        void* wp;
        int wrc = ws_ensure(ctx, "acc.arr0", 4ull * N, &wp);
        if (wrc) return wrc;
        d_a0 = (uint32_t*)wp;
        if ((wrc = ws_ensure(ctx, "acc.arr1", 4ull * N, &wp))) return wrc;
        d_a1 = (uint32_t*)wp;
        if ((wrc = ws_ensure(ctx, "acc.arr_which", 64, &wp))) return wrc;
        d_which = (uint32_t*)wp;
        k_arrangement_at<<<dim3(1), dim3(1024), 0, s>>>(a->D.planes, a->D.stride_w, a->P.L.kind, offset + n_lines - 1u, N, d_a0, d_a1,
                                                     d_which);
        HIP_TRY(hipGetLastError());
        uint32_t which = 0;
        HIP_TRY(hipMemcpyAsync(&which, d_which, 4, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        HIP_TRY(hipMemcpyAsync(h_a, which ? d_a1 : d_a0, 4ull * N, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
    }
    // where the lines' words / lists sit: parse the block again (line kinds, ranks, starts); this uses the workspace
    const uint8_t* img;
    uint64_t len, blk;
    if (a->cur_in_workspace) a->cur_block = -1;  // the view of the current block is about to be overwritten
    a->cnt_block = -1;
    a->win_n = 0;
    int rc = accessor_block_image(a, block, &img, &len, &blk);
    if (rc) return rc;
    DecodePlan P;
    rc = decode_prepare(a->ctx, img, len, blk, 1, &P);
    if (rc) return rc;
    rc = decode_counts_only(a->ctx, img, P);
    if (rc) return rc;
    std::vector<uint8_t> kind(n_lines);
    std::vector<uint32_t> rank(n_lines), wah_start(P.n_wah ? P.n_wah : 1), sparse_start(P.n_sparse ? P.n_sparse : 1);
    HIP_TRY(hipMemcpyAsync(kind.data(), P.L.kind + offset, n_lines, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(rank.data(), P.L.rank + offset, 4ull * n_lines, hipMemcpyDeviceToHost, s));
    if (P.n_wah) HIP_TRY(hipMemcpyAsync(wah_start.data(), P.L.wah_start, 4ull * P.n_wah, hipMemcpyDeviceToHost, s));
    if (P.n_sparse) HIP_TRY(hipMemcpyAsync(sparse_start.data(), P.L.sparse_start, 4ull * P.n_sparse, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    stage_collect(a->ctx);
    const DecBlock& D = P.blocks_h[0];
    const std::vector<uint8_t>& himg = a->zstd ? a->mini : a->file;
    info->image = himg.data();
    info->image_len = himg.size();
    for (uint32_t i = 0; i < n_lines; ++i) {
        const bool is_wah = (kind[i] & KIND_WAH) != 0u;
        h_sparse[i] = is_wah ? 0 : 1;
        if (is_wah) {
            if (rank[i] >= P.n_wah) return set_error(XSI_ERR_FORMAT, "get_internal_access: WAH rank out of range");
            h_offsets[i] = D.gt_off + D.off_wah + 2ull * wah_start[D.wah_first + rank[i]];
        } else {
            if (rank[i] >= P.n_sparse) return set_error(XSI_ERR_FORMAT, "get_internal_access: sparse rank out of range");
            h_offsets[i] = D.gt_off + D.off_sparse + sparse_start[D.sparse_first + rank[i]];
        }
        if (h_offsets[i] + a->aet > himg.size()) return set_error(XSI_ERR_FORMAT, "get_internal_access: line data outside the image");
    }
    if (h_sparse[0]) {  // REF listed (MSB of the count set): ALT 1 is the default allele (accessor_internals_new.hpp:458-460)
        uint64_t num = 0;
        for (uint32_t b = 0; b < a->aet; ++b) num |= (uint64_t)himg[h_offsets[0] + b] << (8 * b);
        info->default_allele = (num >> (8 * a->aet - 1)) & 1ull ? 1 : 0;
    }
    return XSI_OK;
}

int xsi_accessor_allele_counts(xsi_accessor* a, uint64_t* h_counts, uint32_t n_alleles) {
    if (!a || !h_counts) return set_error(XSI_ERR_ARG, "allele_counts: null argument");
    if (a->last_counts.size() < n_alleles) return set_error(XSI_ERR_ARG, "no fill with %u alleles precedes this call", n_alleles);
    memcpy(h_counts, a->last_counts.data(), 8ull * n_alleles);
    return XSI_OK;
}

uint64_t xsi_accessor_hap_samples(const xsi_accessor* a) { return a ? a->hap_samples : 0; }
uint64_t xsi_accessor_num_samples(const xsi_accessor* a) { return a ? a->names.size() : 0; }
const char* xsi_accessor_sample_name(const xsi_accessor* a, uint64_t i) {
    return (a && i < a->names.size()) ? a->names[i].c_str() : nullptr;
}

void xsi_accessor_close(xsi_accessor* a) {
    if (!a) return;
    accessor_readahead_join(a);
    if (a->pf_ctx) xsi_hip_ctx_destroy(a->pf_ctx);
    if (a->ctx) (void)hipStreamSynchronize(a->ctx->stream);
    accessor_drop_registration(a);
    for (auto& e : a->owned_arrays) (void)hipHostFree(e.first);  // (arrays of xsi_accessor_alloc_array the caller did not free)
    for (auto& e : a->cache) (void)hipFree(e.mem);
    if (a->d_file) (void)hipFree(a->d_file);
    if (a->d_mini) (void)hipFree(a->d_mini);
    if (a->d_rows) (void)hipFree(a->d_rows);
    if (a->h_rows) (void)hipHostFree(a->h_rows);
    if (a->h_counts) (void)hipHostFree(a->h_counts);
    if (a->h_meta) (void)hipHostFree(a->h_meta);
    if (a->h_bmeta) (void)hipHostFree(a->h_bmeta);
    if (a->d_sel) (void)hipFree(a->d_sel);
    if (a->d_sel_row) (void)hipFree(a->d_sel_row);
    if (a->h_sel_row) (void)hipHostFree(a->h_sel_row);
    if (a->h_sel_ac) (void)hipHostFree(a->h_sel_ac);
    if (a->ctx) xsi_hip_ctx_destroy(a->ctx);
    delete a;
}

}  // extern "C"
