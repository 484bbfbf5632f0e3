// xsi_dist.hip — multi-GPU at the C ABI: block sharding and the gather of the compressed block streams to the
// writer rank over RCCL (xGMI inside a node).
//
// Blocks are independent (a fresh GtBlock with a = iota per block, gt_block.hpp:179-180, xsi_factory.hpp:536-537;
// the decoder resets `a` per block, accessor_internals_new.hpp:144), so the path shards with no data-path
// collective: rank r of G owns a contiguous block range and runs xsi_hip_encode_* / xsi_hip_decode_* on it.  The
// one exchange step is what XsiFactoryExt::finalize_file (xsi_factory.hpp:543-605) needs on the writer rank: every
// rank's blocks region, in rank (= file) order, and the offset of every block.  Sizes travel first (ncclAllGather
// of four u64 per rank), then exactly each rank's bytes (grouped ncclSend / ncclRecv), on the communicator's own
// stream, ordered behind what the context's stream held at the call: the caller's next work overlaps with it.
//
// librccl is bound at run time (dlopen of librccl.so.1): a single-GPU user needs no RCCL, and a process that has
// already loaded its own RCCL (PyTorch) gets that same library by its soname.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/xsi_hip.h"
#include "xsi_ctx.hpp"

using namespace xsi;

namespace {
struct RcclApi {
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;  // (required: the recovery of a partly posted group needs it)
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    bool ok = false;
};
const RcclApi& rccl() {
    static RcclApi r = [] {
        RcclApi a;
        void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
        if (!h) return a;
#define XSI_SYM(field, name) a.field = reinterpret_cast<decltype(a.field)>(dlsym(h, name))
        XSI_SYM(GetUniqueId, "ncclGetUniqueId");
        XSI_SYM(CommInitRank, "ncclCommInitRank");
        XSI_SYM(CommDestroy, "ncclCommDestroy");
        XSI_SYM(CommAbort, "ncclCommAbort");
        XSI_SYM(AllGather, "ncclAllGather");
        XSI_SYM(Send, "ncclSend");
        XSI_SYM(Recv, "ncclRecv");
        XSI_SYM(GroupStart, "ncclGroupStart");
        XSI_SYM(GroupEnd, "ncclGroupEnd");
        XSI_SYM(GetErrorString, "ncclGetErrorString");
#undef XSI_SYM
        a.ok = a.GetUniqueId && a.CommInitRank && a.CommDestroy && a.CommAbort && a.AllGather && a.Send && a.Recv && a.GroupStart &&
               a.GroupEnd && a.GetErrorString;
        return a;
    }();
    return r;
}

#define NCCL_TRY(expr)                                                                                           \
    do {                                                                                                         \
        ncclResult_t _r = (expr);                                                                                \
        if (_r != ncclSuccess) return set_error(XSI_ERR_HIP, "%s: %s", #expr, rccl().GetErrorString(_r));        \
    } while (0)
#define HIP_TRY(expr)                                                                                             \
    do {                                                                                                          \
        hipError_t _e = (expr);                                                                                   \
        if (_e != hipSuccess) return set_error(XSI_ERR_HIP, "%s: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, \
                                               __LINE__);                                                         \
    } while (0)

// offsets of a rank's blocks, relative to its own region -> relative to the concatenated region
__global__ void k_rebase_offsets(uint64_t* offs, uint64_t n, uint64_t base) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) offs[i] += base;
}
}  // namespace

struct xsi_hip_comm {
    ncclComm_t comm = nullptr;
    int world = 1, rank = 0;
    xsi_hip_ctx* ctx = nullptr;
    hipStream_t cs = nullptr;    // the exchange runs here, ordered behind the context's stream by ev_in, so that
    hipEvent_t ev_in = nullptr, ev_done = nullptr;  // work the caller enqueues next (the decode) overlaps with it
    bool broken = false;         // a group was launched with only part of its posts (see gather_block_streams_round): aborted, unusable
    bool abort_failed = false;   // ... and ncclCommAbort itself failed: the operations it left on `cs` never complete
    uint64_t* d_meta = nullptr;  // [world + 1][4]: every rank's {bytes, blocks, region capacity, offsets capacity}; the last is this rank's (send buffer)
};

extern "C" {

void xsi_hip_shard_blocks(uint64_t n_blocks, int world, int rank, uint64_t* lo, uint64_t* hi) {
    if (world < 1) world = 1;
    if (rank < 0) rank = 0;
    if (lo) *lo = ((uint64_t)rank * n_blocks + (uint64_t)world - 1u) / (uint64_t)world;
    if (hi) *hi = ((uint64_t)(rank + 1) * n_blocks + (uint64_t)world - 1u) / (uint64_t)world;
}

int xsi_hip_shard_of_block(uint64_t n_blocks, int world, uint64_t block) {
    if (world < 1 || block >= n_blocks) return -1;
    return (int)(block * (uint64_t)world / n_blocks);  // the inverse of xsi_hip_shard_blocks: lo(r) = ceil(r B / G) <= b < lo(r + 1)
}

int xsi_hip_comm_unique_id(uint8_t id[XSI_HIP_COMM_ID_BYTES]) {
    if (!id) return set_error(XSI_ERR_ARG, "comm_unique_id: null id");
    if (!rccl().ok) return set_error(XSI_ERR_UNSUPPORTED, "librccl.so.1 could not be loaded");
    static_assert(sizeof(ncclUniqueId) == XSI_HIP_COMM_ID_BYTES, "ncclUniqueId is 128 bytes");
    ncclUniqueId u;
    NCCL_TRY(rccl().GetUniqueId(&u));
    memcpy(id, &u, sizeof(u));
    return XSI_OK;
}

int xsi_hip_comm_create(xsi_hip_comm** out, xsi_hip_ctx* ctx, int world, int rank, const uint8_t id[XSI_HIP_COMM_ID_BYTES]) {
    if (!out || !ctx || !id || world < 1 || rank < 0 || rank >= world) return set_error(XSI_ERR_ARG, "comm_create: bad argument");
    if (!rccl().ok) return set_error(XSI_ERR_UNSUPPORTED, "librccl.so.1 could not be loaded");
    HIP_TRY(hipSetDevice(ctx->device));
    ncclUniqueId u;
    memcpy(&u, id, sizeof(u));
    xsi_hip_comm* c = new (std::nothrow) xsi_hip_comm();
    if (!c) return set_error(XSI_ERR_ARG, "comm_create: out of memory");
    ncclResult_t r = rccl().CommInitRank(&c->comm, world, u, rank);
    if (r != ncclSuccess) {
        delete c;
        return set_error(XSI_ERR_HIP, "ncclCommInitRank: %s", rccl().GetErrorString(r));
    }
    c->world = world;
    c->rank = rank;
    c->ctx = ctx;
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&c->d_meta), 32ull * ((size_t)world + 1u));
    if (e != hipSuccess) {
        rccl().CommDestroy(c->comm);
        delete c;
        return set_error(XSI_ERR_HIP, "hipMalloc: %s", hipGetErrorString(e));
    }
    if (hipStreamCreateWithFlags(&c->cs, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_in, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_done, hipEventDisableTiming) != hipSuccess) {
        xsi_hip_comm_destroy(c);
        return set_error(XSI_ERR_HIP, "comm_create: stream / events could not be created");
    }
    *out = c;
    return XSI_OK;
}

void xsi_hip_comm_destroy(xsi_hip_comm* c) {
    if (!c) return;
    if (c->cs && !c->abort_failed) (void)hipStreamSynchronize(c->cs);  // (operations of a communicator that could not be aborted never finish)
    if (c->ev_in) (void)hipEventDestroy(c->ev_in);
    if (c->ev_done) (void)hipEventDestroy(c->ev_done);
    if (c->cs && !c->abort_failed) (void)hipStreamDestroy(c->cs);  // (destroying a stream waits for its work: leaked in that case)
    if (c->d_meta) (void)hipFree(c->d_meta);
    if (c->comm && rccl().ok) rccl().CommDestroy(c->comm);
    delete c;
}

int xsi_hip_comm_world(const xsi_hip_comm* c) { return c ? c->world : 0; }
int xsi_hip_comm_rank(const xsi_hip_comm* c) { return c ? c->rank : -1; }

int xsi_hip_gather_block_streams(xsi_hip_comm* c, const void* d_region, uint64_t nbytes, const uint64_t* d_offsets,
                                 uint64_t n_blocks, int dst, void* d_region_all, uint64_t region_capacity,
                                 uint64_t* d_offsets_all, uint64_t offsets_capacity, uint64_t* h_bytes_per_rank,
                                 uint64_t* h_blocks_per_rank) {
    return xsi_hip_gather_block_streams_round(c, d_region, nbytes, d_offsets, n_blocks, dst, d_region_all, region_capacity,
                                              d_offsets_all, offsets_capacity, 0, 0, h_bytes_per_rank, h_blocks_per_rank);
}

int xsi_hip_gather_block_streams_round(xsi_hip_comm* c, const void* d_region, uint64_t nbytes, const uint64_t* d_offsets,
                                       uint64_t n_blocks, int dst, void* d_region_all, uint64_t region_capacity,
                                       uint64_t* d_offsets_all, uint64_t offsets_capacity, uint64_t region_base,
                                       uint64_t blocks_base, uint64_t* h_bytes_per_rank, uint64_t* h_blocks_per_rank) {
    if (!c || dst < 0 || dst >= c->world) return set_error(XSI_ERR_ARG, "gather_block_streams: bad communicator / dst");
    if (c->broken) return set_error(XSI_ERR_HIP, "gather_block_streams: this communicator was aborted by an earlier failed exchange");
    if ((nbytes && !d_region) || (n_blocks && !d_offsets)) return set_error(XSI_ERR_ARG, "gather_block_streams: null input");
    const RcclApi& R = rccl();
    hipStream_t s = c->cs;
    const int W = c->world, me = c->rank;
    HIP_TRY(hipSetDevice(c->ctx->device));
    HIP_TRY(hipEventRecord(c->ev_in, c->ctx->stream));  // everything the caller has enqueued so far (the encode) comes first
    HIP_TRY(hipStreamWaitEvent(s, c->ev_in, 0));
    // Whatever happens below, ev_done is recorded on the exchange stream before this call returns, so that a later
    // xsi_hip_comm_wait never waits on the event of an EARLIER exchange.
    struct DoneGuard {
        xsi_hip_comm* c;
        ~DoneGuard() { (void)hipEventRecord(c->ev_done, c->cs); }
    } done_guard{c};
    // 1. sizes: every rank's bytes and blocks, and (from the writer rank) the room left in its output buffers behind the
    //    bases of this round, so that all ranks reach the same verdict before anybody sends
    if (!d_region_all) region_capacity = 0;  // a writer rank without buffers fails the capacity check on every rank
    if (!d_offsets_all) offsets_capacity = 0;
    const uint64_t mine[4] = {nbytes, n_blocks, region_capacity > region_base ? region_capacity - region_base : 0,
                              offsets_capacity > blocks_base ? offsets_capacity - blocks_base : 0};
    HIP_TRY(hipMemcpyAsync(c->d_meta + 4u * (size_t)W, mine, 32, hipMemcpyHostToDevice, s));
    NCCL_TRY(R.AllGather(c->d_meta + 4u * (size_t)W, c->d_meta, 4, ncclUint64, c->comm, s));
    std::vector<uint64_t> meta4(4u * (size_t)W);
    HIP_TRY(hipMemcpyAsync(meta4.data(), c->d_meta, 32ull * (size_t)W, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    std::vector<uint64_t> meta(2u * (size_t)W);
    uint64_t tot_b = 0, tot_n = 0;
    for (int r = 0; r < W; ++r) {
        meta[2u * r] = meta4[4u * r];
        meta[2u * r + 1u] = meta4[4u * r + 1u];
        if (h_bytes_per_rank) h_bytes_per_rank[r] = meta[2u * r];
        if (h_blocks_per_rank) h_blocks_per_rank[r] = meta[2u * r + 1u];
        tot_b += meta[2u * r];
        tot_n += meta[2u * r + 1u];
    }
    if (tot_b > meta4[4u * dst + 2u] || tot_n > meta4[4u * dst + 3u])
        return set_error(XSI_ERR_CAPACITY, "gather_block_streams: %llu bytes / %llu blocks, the writer rank has room for %llu / %llu",
                         (unsigned long long)tot_b, (unsigned long long)tot_n, (unsigned long long)meta4[4u * dst + 2u],
                         (unsigned long long)meta4[4u * dst + 3u]);
    // 2. exactly each rank's bytes and offsets to the writer rank (no padding to the longest region).  A call that
    //    fails inside the group must not return with the group open: every post goes through `post`, which keeps the
    //    first error and skips the rest, and ncclGroupEnd is issued whatever happened.
    // Test hooks (tests/test_gpu_dist.py): XSI_DIST_SELF_SEND=1 sends the writer rank's own part to itself with
    // ncclSend + ncclRecv inside the group instead of copying it, so that the point-to-point branch runs on a one-GPU
    // box; XSI_DIST_TEST_BAD_RECV=1 makes the FIRST receive of the group fail with ncclInvalidArgument as a refused post
    // would (reported here: nothing invalid is handed to RCCL and nothing has been queued yet, so no peer is left
    // waiting for a message that never comes).
    const bool self_send = test_hook_env("XSI_DIST_SELF_SEND") != nullptr;
    const bool bad_recv = test_hook_env("XSI_DIST_TEST_BAD_RECV") != nullptr;
    uint8_t* const reg_all = static_cast<uint8_t*>(d_region_all) + region_base;
    uint64_t* const off_all = d_offsets_all ? d_offsets_all + blocks_base : nullptr;
    ncclResult_t first = ncclSuccess;
    const char* what = "";
    int posted = 0;
    auto post = [&](const char* name, auto&& call) {
        if (first != ncclSuccess) return;
        const ncclResult_t r = call();
        if (r != ncclSuccess) {
            first = r;
            what = name;
        } else {
            ++posted;
        }
    };
    NCCL_TRY(R.GroupStart());
    if (me == dst) {
        uint64_t bb = 0, bn = 0;
        for (int r = 0; r < W; ++r) {
            const uint64_t b = meta[2u * r], n = meta[2u * r + 1u];
            if (r != me || self_send) {
                if (b) post("ncclRecv(region)", [&] { return bad_recv ? ncclInvalidArgument : R.Recv(reg_all + bb, b, ncclUint8, r, c->comm, s); });
                if (n) post("ncclRecv(offsets)", [&] { return R.Recv(off_all + bn, n, ncclUint64, r, c->comm, s); });
            }
            bb += b;
            bn += n;
        }
    }
    if (me != dst || self_send) {
        if (nbytes) post("ncclSend(region)", [&] { return R.Send(d_region, nbytes, ncclUint8, dst, c->comm, s); });
        if (n_blocks) post("ncclSend(offsets)", [&] { return R.Send(d_offsets, n_blocks, ncclUint64, dst, c->comm, s); });
    }
    const ncclResult_t ge = R.GroupEnd();  // always: a group left open poisons the communicator's next call
    if (first != ncclSuccess) {
        // Posts that went through before the refused one were launched by ncclGroupEnd: their peers have matching
        // operations outstanding and this rank's share of the group is incomplete.  Nothing sound can follow on this
        // communicator: it is aborted, so that the operations already launched do not wait for ever, and every
        // later call on it fails at once.  (A first post that is refused leaves nothing behind: the communicator stays usable.)
        if (posted > 0) {
            // (ncclCommAbort is part of rccl().ok: a library without it never gets this far)
            const ncclResult_t ar = R.CommAbort(c->comm);
            c->comm = nullptr;
            c->broken = true;
            // an abort that itself failed leaves the launched operations queued on c->cs: comm_destroy must not wait for them
            c->abort_failed = ar != ncclSuccess;
        }
        return set_error(XSI_ERR_HIP, "gather_block_streams: %s: %s%s", what, R.GetErrorString(first),
                         posted > 0 ? " (communicator aborted: part of the group had been posted)" : "");
    }
    if (ge != ncclSuccess) return set_error(XSI_ERR_HIP, "gather_block_streams: ncclGroupEnd: %s", R.GetErrorString(ge));
    if (me == dst) {
        uint64_t bb = 0, bn = 0;
        for (int r = 0; r < W; ++r) {
            const uint64_t b = meta[2u * r], n = meta[2u * r + 1u];
            if (r == me && !self_send) {
                if (b) HIP_TRY(hipMemcpyAsync(reg_all + bb, d_region, b, hipMemcpyDeviceToDevice, s));
                if (n) HIP_TRY(hipMemcpyAsync(off_all + bn, d_offsets, 8ull * n, hipMemcpyDeviceToDevice, s));
            }
            if (n && (bb + region_base)) {
                k_rebase_offsets<<<dim3((unsigned)((n + 255u) / 256u)), dim3(256), 0, s>>>(off_all + bn, n, bb + region_base);
                HIP_TRY(hipGetLastError());
            }
            bb += b;
            bn += n;
        }
    }
    return XSI_OK;  // done_guard records ev_done behind everything enqueued above
}

int xsi_hip_comm_wait(xsi_hip_comm* c, int host) {
    if (!c) return set_error(XSI_ERR_ARG, "comm_wait: null communicator");
    if (host)
        HIP_TRY(hipEventSynchronize(c->ev_done));
    else
        HIP_TRY(hipStreamWaitEvent(c->ctx->stream, c->ev_done, 0));
    return XSI_OK;
}

}  // extern "C"
