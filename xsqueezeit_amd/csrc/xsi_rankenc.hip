// xsi_rankenc.hip — element-major ("rank tracking") PBWT encode chains for gfx950: k_chain_rank_enc (N <= 65536, one
// workgroup per block, described here) and k_chain_rank_enc_multi (N <= 524288, several workgroups per block, below).
//
// Reference behaviour (per block, for every WAH line k in order; sparse lines never touch `a`,
// gt_block.hpp:299-326):
//   y_k[i]  = x_k[a_k[i]]                                         (wah.hpp:530-537, gather through a)
//   a_{k+1} = [a_k[i] : y_k[i]=0] ++ [a_k[i] : y_k[i]=1]            (internal_gt_record.hpp:32-59)
// with a_0 the identity (gt_block.hpp:179).  The position-major kernels move `a` around: per member
// and line a gather of its key bit, a ballot, two lane-prefix counts and a scatter, ~18-24 vector
// instructions per 64 members — and on this chip a wave64 vector instruction takes the SIMD for four
// cycles, so the instruction count is the bound, not LDS or HBM.
//
// This kernel never materialises `a`.  Let r_k(h) be the position of haplotype h in a_k (the inverse
// permutation, r_0(h) = h).  Then
//   y_k[r_k(h)] = x_k(h)                                            -> only the ONES of x_k are written
//   r_{k+1}(h)  = x_k(h) ? Z_k + ones_k(r_k(h)) : r_k(h) - ones_k(r_k(h))
// where ones_k(r) = set bits of y_k before position r and Z_k = zeros of y_k (a stable partition moves
// a zero at r to "zeros before r" and a one to Z + "ones before r").  A lane owns haplotype
// 64*chunk + lane, so the key bits of a chunk are one 64-bit word of the INPUT row: they arrive in
// SGPRs by scalar loads and act as the select mask directly — no gather of key bits, no ballot, no
// v_mbcnt, no v_readlane / v_writelane.  Per line:
//   M  every lane advances its E ranks: one ds_read_b64 gather of {32 row bits, ones before them},
//      v_bfm/v_and/v_bcnt, subtract, add, select (8 vector instructions per 64 haplotypes); and with
//      the new rank it deposits its bit of the NEXT line into that line's row (LDS atomic OR, chunks
//      whose 64 input bits are all zero are skipped: most chunks of most lines);
//   A  (barrier) the finished row y_{k+1}: two words per thread -> popcounts, wave scan, stored to HBM;
//   B  (barrier) cross-wave prefix -> the rank-select table {bits, ones before} of line k+1, row cleared;
//      (barrier).
// Bits at or beyond N never enter a row; lanes beyond N idle on rank 0.
#include "xsi_kernels.hpp"

#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include <vector>

#include "xsi_device.hpp"

namespace xsi {

struct RankEncArgs {
    const uint32_t* wah_lines;  // [rank] binary line
    const uint32_t* src;        // planes by binary line (natural haplotype order)
    uint32_t src_stride_w;
    uint32_t* dst;              // permuted rows y by rank
    uint32_t dst_stride_w;
    uint32_t N;
};

using LdsU32 = __attribute__((address_space(3))) uint32_t;
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
using LdsU2 = __attribute__((address_space(3))) u32x2;
using ConstU32 = __attribute__((address_space(4))) const uint32_t;

// Uniform (scalar) loads: casting to the constant address space makes the compiler use s_load for
// addresses it knows to be wave-uniform.  The input rows and the line list are never written by this
// kernel, so the scalar cache cannot go stale.
__device__ __forceinline__ const ConstU32* as_const(const uint32_t* p) {
    return reinterpret_cast<const ConstU32*>(reinterpret_cast<uintptr_t>(p));
}

typedef uint32_t v4u __attribute__((ext_vector_type(4)));
typedef uint32_t v8u __attribute__((ext_vector_type(8)));
typedef uint32_t v16u __attribute__((ext_vector_type(16)));
// s_buffer_load_dwordx8: clang has no builtin for it; the LLVM intrinsic is bound by name
extern "C" __device__ v8u __xsi_s_buffer_load_v8(v4u rsrc, uint32_t byte_offset, uint32_t cache_policy)
    __asm("llvm.amdgcn.s.buffer.load.v8i32");
extern "C" __device__ v16u __xsi_s_buffer_load_v16(v4u rsrc, uint32_t byte_offset, uint32_t cache_policy)
    __asm("llvm.amdgcn.s.buffer.load.v16i32");
template <int G>
__device__ __forceinline__ void sbuf_load_chunks(v4u rsrc, uint32_t chunk, uint64_t (&out)[G]) {
    static_assert(G == 4 || G == 8, "s_buffer_load_dwordx8 / x16");
    if constexpr (G == 4) {
        const v8u v = __xsi_s_buffer_load_v8(rsrc, chunk * 8u, 0u);
#pragma unroll
        for (int e = 0; e < G; ++e) out[e] = ((uint64_t)v[2 * e + 1] << 32) | v[2 * e];
    } else {
        const v16u v = __xsi_s_buffer_load_v16(rsrc, chunk * 8u, 0u);
#pragma unroll
        for (int e = 0; e < G; ++e) out[e] = ((uint64_t)v[2 * e + 1] << 32) | v[2 * e];
    }
}

template <int E, int RG>
__global__ void __launch_bounds__(1024) k_chain_rank_enc(const EncBlock* __restrict__ eblocks, RankEncArgs A) {
    constexpr uint32_t T = 1024, W = 16;
    constexpr uint32_t NA = W * E * 64u;  // haplotype capacity
    constexpr uint32_t CW = NA / 32u;     // words of a row over NA positions (= 2 per active thread)
    constexpr int G = RG;                 // chunks per group: their gathers are all in flight together
    static_assert(E % G == 0 && E >= 4 && E <= 64, "E in steps of the gather group");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // [0, 8 CW)          rank-select table of the current line: CW pairs {32 row bits, ones before them}
    // [16384, +4 CW)     row under construction (next line), zero between lines
    // [16384 + 4 CW, ..) per-wave ones of the row
    // Both arrays start on a power of two above their size, so "index bits | base" forms an address.
    constexpr uint32_t YOFF = 16384u;
    static_assert(8u * CW <= YOFF, "table of at most 2048 pairs");
    uint2* table = reinterpret_cast<uint2*>(smem);
    uint32_t* ybits = reinterpret_cast<uint32_t*>(smem + YOFF);
    uint32_t* wtot = ybits + CW;
    const uint32_t tab_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
    uint32_t y_lds = tab_lds + YOFF;
    asm volatile("v_mov_b32 %0, %1" : "=v"(y_lds) : "s"(y_lds));  // per-lane: v_and_or_b32 takes one scalar/literal
    const uint32_t N = A.N;
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const uint32_t w = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6));
    const EncBlock& B = eblocks[blockIdx.x];
    if (B.has_haploid) return;  // k_chain_lds takes the blocks with fully haploid lines
    const uint32_t wah_first = B.wah_first, n_wah = B.n_wah;
    if (n_wah == 0) return;

    const uint32_t c0 = w * E;  // my first chunk
    uint32_t r[E];
    static_for<0, E>([&](auto ec) {
        constexpr int e = decltype(ec)::value;
        const uint32_t h = (c0 + (uint32_t)e) * 64u + lane;
        r[e] = h < N ? h : 0u;
    });
    for (uint32_t i = tid; i < CW; i += T) ybits[i] = 0;

    const ConstU32* lines = as_const(A.wah_lines) + wah_first;
    const uint32_t row_bytes = ((N + 63u) / 64u) * 8u;  // bytes of an input row that hold haplotypes
    // Key bits of G consecutive chunks of a line: one s_buffer_load_dwordx8 / x16 through a buffer descriptor
    // whose range is the row, so chunks past the end of the row read as zero (hardware range check) and
    // every wave runs the same straight-line code.
    auto row_rsrc = [&](uint32_t line) -> v4u {
        const uint64_t base = reinterpret_cast<uint64_t>(A.src + (size_t)line * A.src_stride_w);
        v4u d;
        d[0] = (uint32_t)base;
        d[1] = (uint32_t)(base >> 32) & 0xFFFFu;  // stride 0: raw buffer, offsets and range in bytes
        d[2] = row_bytes;
        d[3] = 0x00020000u;                       // 32-bit data format (the descriptor word CDNA raw buffers use)
        return d;
    };
    auto xgroup = [&](v4u rsrc, uint32_t cb, uint64_t (&out)[G]) { sbuf_load_chunks<G>(rsrc, cb, out); };
    // deposit my bit of a line into its row at my current rank (ones only)
    auto deposit = [&](uint64_t xm, uint32_t rr) {
        if (xm) {
            if (__builtin_amdgcn_inverse_ballot_w64(xm)) {
                LdsU32* p = reinterpret_cast<LdsU32*>((uintptr_t)(((rr >> 3) & 0x1FFCu) | y_lds));
                __hip_atomic_fetch_or(p, 1u << (rr & 31u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
    };
    // phases A + B for the row in ybits -> table, y row `rank` to HBM; returns the row's zeros
    auto finish_row = [&](uint32_t rank) -> uint32_t {
        lds_barrier();  // every deposit has landed
        uint32_t w0 = 0, w1 = 0;
        const bool act = tid < CW / 2u;
        if (act) {
            const uint2 v = *reinterpret_cast<const uint2*>(ybits + 2u * tid);
            w0 = v.x;
            w1 = v.y;
        }
        const uint32_t c = (uint32_t)__popc(w0) + (uint32_t)__popc(w1);
        const uint32_t inc = wave_scan_incl_dpp(c);
        if (lane == 63u) wtot[w] = inc;
        if (tid < A.dst_stride_w / 2u)
        {
            typedef uint32_t v2u_nt __attribute__((ext_vector_type(2)));
            v2u_nt vv = {w0, w1};
            __builtin_nontemporal_store(vv, reinterpret_cast<v2u_nt*>(A.dst + (size_t)rank * A.dst_stride_w) + tid);
        }
        lds_barrier();
        uint32_t sc = row16_scan_incl(lane < W ? wtot[lane] : 0u);
        const uint32_t ones = (uint32_t)__builtin_amdgcn_readlane((int)sc, W - 1);
        const uint32_t base = w ? (uint32_t)__builtin_amdgcn_readlane((int)sc, (int)w - 1) : 0u;
        if (act) {
            const uint32_t pre0 = base + inc - c;
            *reinterpret_cast<uint4*>(table + 2u * tid) = make_uint4(w0, pre0, w1, pre0 + (uint32_t)__popc(w0));
            *reinterpret_cast<uint2*>(ybits + 2u * tid) = make_uint2(0u, 0u);
        }
        lds_barrier();
        return N - ones;
    };

    // line 0: ranks are the identity, so its row is the input row itself
    {
        const v4u rs0 = row_rsrc(lines[0]);
        lds_barrier();  // row cleared
        static_for<0, E / G>([&](auto gc) {
            constexpr int g0 = decltype(gc)::value * G;
            uint64_t x0[G];
            xgroup(rs0, c0 + (uint32_t)g0, x0);
            static_for<0, G>([&](auto ec) {
                constexpr int e = decltype(ec)::value;
                deposit(x0[e], r[g0 + e]);
            });
            __builtin_amdgcn_sched_barrier(0);
        });
    }
    uint32_t Z = finish_row(wah_first);

    // The key bits reach the waves through scalar loads straight from the input matrix, which nothing has
    // touched before: left alone every group of every line waits out a full HBM round trip (the groups are
    // kept apart, see below, so nothing else covers it).  One coalesced vector load per thread pulls the row
    // of line j+2 into L2 a whole line ahead; the scalar loads then hit L2 (37.2 -> 30.1 ms at 64 976 x 2 M).
    auto prefetch_row = [&](uint32_t line) -> uint2 {
        const uint2* rowp = reinterpret_cast<const uint2*>(A.src + (size_t)line * A.src_stride_w);
        return tid * 8u < row_bytes ? rowp[tid] : make_uint2(0u, 0u);
    };
    for (uint32_t j = 0; j < n_wah; ++j) {
        const bool more = j + 1u < n_wah;
        const uint2 pf = prefetch_row(lines[j + 2u < n_wah ? j + 2u : j]);
        // after the last line the deposits (of the same line again) go into a row nobody reads
        const v4u rsc = row_rsrc(lines[j]);
        const v4u rsn = row_rsrc(lines[more ? j + 1u : j]);
        static_for<0, E / G>([&](auto gc) {
            constexpr int g0 = decltype(gc)::value * G;
            uint64_t xc[G], xn[G];
            u32x2 pr[G];
            xgroup(rsc, c0 + (uint32_t)g0, xc);
            xgroup(rsn, c0 + (uint32_t)g0, xn);
            static_for<0, G>([&](auto ec) {
                constexpr int e = decltype(ec)::value;
                pr[e] = *reinterpret_cast<const LdsU2*>((uintptr_t)(((r[g0 + e] >> 2) & 0x3FF8u) | tab_lds));
            });
            static_for<0, G>([&](auto ec) {
                constexpr int e = decltype(ec)::value;
                const uint32_t rr = r[g0 + e];
                // ones before my position: table prefix + set bits below me in my word; v_bfe_u32 reads
                // only the low 5 bits of its width operand, so rr itself serves as "rr & 31"
                const uint32_t ob = (uint32_t)__popc(__builtin_amdgcn_ubfe(pr[e][0], 0u, rr)) + pr[e][1];
                const uint32_t rn = __builtin_amdgcn_inverse_ballot_w64(xc[e]) ? Z + ob : rr - ob;
                r[g0 + e] = rn;
                deposit(xn[e], rn);
            });
            // keep the groups apart: hoisting every group's scalar loads to the top of the line would
            // need 4 E SGPRs and spill them lane by lane
            __builtin_amdgcn_sched_barrier(0);
        });
        asm volatile("" ::"v"(pf.x), "v"(pf.y));  // the prefetch has landed (nothing reads the registers)
        if (more) Z = finish_row(wah_first + j + 1u);
    }
}

static const int k_rankenc_E[] = {8, 16, 24, 32, 40, 48, 56, 64};

static int rankenc_e_for(uint32_t N) {
    for (int e : k_rankenc_E)
        if ((uint32_t)e * 1024u >= N) return e;
    return 0;
}

bool chain_rank_enc_supported(uint32_t N) { return N >= 2u && N <= 65536u; }

hipError_t launch_rank_encode(hipStream_t s, const EncBlock* blocks, uint32_t n_blocks, const EncLines& L) {
    if (!n_blocks) return hipSuccess;
    RankEncArgs A{};
    A.wah_lines = L.wah_lines;
    A.src = L.planes;
    A.src_stride_w = L.plane_stride_w;
    A.dst = reinterpret_cast<uint32_t*>(L.yrows);
    A.dst_stride_w = L.y_stride64 * 2u;
    A.N = L.N;
    const int e = rankenc_e_for(L.N);
    if (!e) return hipErrorInvalidValue;
    const uint32_t CW = 16u * (uint32_t)e * 64u / 32u;
    const uint32_t lds = 16384u + 4u * CW + 64u;
#define XSI_RE_CASE(EE)                                                                               \
    if (e == EE) {                                                                                    \
        auto kern = &k_chain_rank_enc<EE, 8>;  /* (groups of four gathers measured the same: not instantiated) */ \
        hipError_t err = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),                     \
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);   \
        if (err != hipSuccess) return err;                                                            \
        kern<<<dim3(n_blocks), dim3(1024), lds, s>>>(blocks, A);                                      \
        return hipGetLastError();                                                                     \
    }
    XSI_RE_CASE(8)
    XSI_RE_CASE(16)
    XSI_RE_CASE(24)
    XSI_RE_CASE(32)
    XSI_RE_CASE(40)
    XSI_RE_CASE(48)
    XSI_RE_CASE(56)
    XSI_RE_CASE(64)
#undef XSI_RE_CASE
    return hipErrorInvalidValue;
}


// ------------------------------------------------------------------------------------------
// 65 536 < N <= 524 288 haplotypes: the same chain with S = ceil(N / 65536) workgroups per block: k_chain_rank_enc_multi.
//
// The position-major kernel for these sizes (k_chain_stream) moves the prefix array through HBM: 8 N bytes of scattered
// traffic per line and block, one CU per block (696 ms for 153 blocks of 500 000 haplotypes, at the HBM limit of that
// access pattern).  Element-major, a workgroup ("member") keeps 65 536 ranks in registers and holds the rank-select table
// of ALL N positions in its LDS (8 bytes per 32 positions: 128 KiB at 524 288); what the S members of a block exchange
// per line is where the ONES of the next line go, and member m OWNS the row positions [65536 m, 65536 m + 65536).
// How the ones travel is chosen PER LINE by the line's minor allele count (known from the classification).  What a line
// costs (phase clocks, XSI_MULTI_PROF): the gathers + rank updates 3.2 - 4.2 us whatever the line; a hand-off through L2 -
// store acknowledgements, flag, poll, loads - about 3 us; appending to and applying rank lists grows with the ones of
// the row.  Most WAH lines are sparse (allele counts are octave-uniform: at 500 000 haplotypes two of three WAH lines have
// fewer than 49 152 minor alleles), so:
//   minor <= thr   ONE hand-off.  A lane whose next-line bit is set appends its new rank to its wave's list during the
//                  gathers (through a 128-entry buffer in LDS, so that the list leaves as whole 256-byte stores; a row
//                  with more ones than zeros travels as the list of its ZEROS); every wave flags its own list (8 bytes:
//                  length + sequence number) as soon as its stores have drained and polls the flags of list w of every
//                  member; EVERY member applies ALL lists to a bitmap of the whole row in its own LDS (LDS atomic OR) and
//                  scans that row itself into its table: no slices, no second hand-off, no table copy;
//   minor >  thr   TWO hand-offs whose cost does not depend on the ones: a member deposits the ones of its haplotypes
//                  into a private bitmap of the whole row in its LDS (ds_or_b32, as the one-workgroup kernel does), stores
//                  it (N / 8 bytes) through the XCD's L2, and the owner of a slice ORs the S members' pieces of it (8
//                  bytes per thread and member, no filter, no per-entry work), scans it and publishes finished
//                  {bits, ones before} table entries plus the slice's ones in a flag; every member copies the S slices
//                  into its table, adding each slice's base.
// The bitmap of a row (N / 8 bytes) cannot stand next to the table (S x 16 KiB of the 160 KiB) while that is being
// gathered from: it takes the place of the table's first half once every wave is through its gathers (barrier, clear,
// barrier), so the deposits of a bitmap line are a pass of their own behind the gathers; the appends of a list line need
// no such room and stay fused with the gathers.
// A member can run at most one line ahead of another (to finish line j + 1 it needs every member's lists or bitmap of
// that line, which a member publishes behind its gathers of line j): lists and their flags are double-buffered by the
// parity of the line's sequence number.  Bitmaps and slices need no second buffer: a member passes the slice poll of a
// line only when every member has flagged its slice, i.e. has finished reading the bitmaps, and passes the bitmap poll
// of a later line only when every member has flagged its bitmap, i.e. has finished copying the slices.  Sequence
// numbers grow by one per line and never repeat inside a launch.
// Hand-offs go through the L2 of ONE XCD: the members of a group are dispatched 8 workgroups apart, i.e. to the same XCD
// (HW_REG_XCC_ID == blockIdx.x & 7 for every workgroup of a launch, tools/microbench4.hip), and a handshake at the start of
// the launch verifies it (members on different XCDs abort the launch: the host runs the batch with k_chain_stream).  So
// the bytes are stored PLAIN (write-through L1, the line stays in the XCD's L2) and loaded with sc1 loads (past L1, served
// by that L2): 128 KiB written by one workgroup and read by 7 others takes 1.36 us this way against 4.06 us with sc1
// stores, which drop the line from L2 (profiles/r03_microbench4.txt).  A storing wave waits for its stores (vmcnt(0)) in
// front of the flag (or of the workgroup barrier behind which the flag is stored); a polling wave loads the bytes only
// after its own poll has matched.
// Every workgroup of the grid must be resident for the exchanges to complete: the grid is at most one workgroup per CU
// and groups walk the blocks persistently.  A wait that does not complete within `timeout_ticks` of the 100 MHz clock
// raises the abort flag; every poll loop looks at it, the waves leave (a barrier only waits for the waves that are left)
// and the host runs the batch again with k_chain_stream (xsi_api.hip, encode_run).
// History (configs[3] shard, encode chain ms; docs/EXPERIMENTS.md has the measurements): every member applying every
// list + scanning the whole row + one counter meeting per line (round 2) 430; owner-computes lists -> slices -> table
// copy for every line (rounds 3 - 4) 363 -> 297; bitmaps for every line 326; lists with one hand-off for every line 311;
// the per-line choice 265.
// ------------------------------------------------------------------------------------------
constexpr uint32_t MULTI_LIST_CAP = 4096u;  // ranks per wave and line: every one of its 64 x 64 haplotypes
struct MultiItem {
    uint32_t block, first, count;  // WAH lines [first, first + count) of the block (counted among its WAH lines)
    uint32_t flags;                // MULTI_ITEM_LOAD: the ranks in front of `first` are parked in slot flags >> 8; _STORE: park mine there
};
constexpr uint32_t MULTI_ITEM_LOAD = 1u, MULTI_ITEM_STORE = 2u;

struct RankEncMultiArgs {
    const uint32_t* wah_lines;
    const uint32_t* src;
    uint32_t src_stride_w;
    const uint32_t* cnt;    // ones of every binary line (the classification's counts)
    uint32_t* dst;          // permuted rows y by rank
    uint32_t dst_stride_w;  // words per row
    uint32_t N;
    uint32_t n_blocks;
    uint32_t S;             // workgroups per block
    uint32_t gpx;           // groups per XCD slot: the grid is 8 * gpx * S workgroups
    uint32_t thr;           // lines with at most this many minor alleles travel as lists
    uint32_t* sync;         // [0] abort, [2] profile records written
    uint32_t* list_flags;   // [group][parity][member][32 words]: per wave {length, seq}
    uint32_t* lists;        // [group][parity][S * 16 waves][MULTI_LIST_CAP] ranks, whole 64-entry stores; the length a multiple of 4 (padding: copies of the last entry)
    uint64_t* flags;        // [group][16]: [0, 8) bitmap flags (seq), [8, 16) slice flags (seq << 32 | ones of the slice)
    uint32_t* bmps;         // [group][member][S * 2048] words: the members' private bitmaps of the row
    v4u* slices;            // [group][member][1024]: two table entries each
    uint32_t test_desert;   // testing only: member 1 of every group leaves at once
    uint32_t prof;          // XSI_MULTI_PROF = 1 + workgroup + (wave << 16): that wave records (tag << 56 | 100 MHz clock)
    uint32_t* xcc_ids;      // [group][8]: 1 + XCC_ID of each member (handshake at the start of the launch)
    uint64_t timeout_ticks;
    uint64_t* prof_buf;
    uint32_t prof_cap;
    // the launch's work, cut into per-group item lists by k_multi_schedule (below)
    const uint32_t* item_begin;  // [n_groups + 1]
    const MultiItem* items;
    uint32_t* park;              // [slot][member][64][1024]: the ranks a block's head part hands to its tail part
    uint32_t* park_flags;        // [slot][S]: 1 = that member's ranks are parked
};

// ---- the schedule.  A launch has B blocks for G groups; block b is a serial chain of n_wah(b) lines.  Walked block by
// block (b += G), the groups that get one block fewer idle for a whole block at the end: 153 blocks on 32 groups take five
// rounds for 4.78 rounds of work.  Here every group gets the same number of LINES (McNaughton's wrap-around rule): the
// blocks are laid end to end and cut every T = total / G lines; a block that straddles a cut is run in two parts by two
// groups - its HEAD lines open the timeline of group g + 1, its TAIL lines close the timeline of group g (so the head is
// long finished when the tail begins: n_wah(b) <= T), and the ranks travel through memory in between (`park`: 4 N bytes,
// once per cut, device-scope release / acquire: the two groups sit on different XCDs as a rule).
__global__ void k_multi_schedule(const EncBlock* __restrict__ eblocks, uint32_t n_blocks, uint32_t n_groups, uint32_t* item_begin,
                                 MultiItem* items, uint32_t round_robin) {
    if (threadIdx.x || blockIdx.x) return;
    if (round_robin) {  // (A/B runs: whole blocks dealt out b, b + G, b + 2 G, ... as rounds 2 - 4 did)
        uint32_t n = 0;
        for (uint32_t g = 0; g < n_groups; ++g) {
            item_begin[g] = n;
            for (uint32_t b = g; b < n_blocks; b += n_groups)
                if (!eblocks[b].has_haploid && eblocks[b].n_wah) items[n++] = MultiItem{b, 0u, eblocks[b].n_wah, 0u};
        }
        item_begin[n_groups] = n;
        return;
    }
    constexpr uint32_t MIN_PART = 96u;  // no part shorter than this (its fixed costs: a park, a wait, a first row)
    uint64_t total = 0;
    uint32_t live = 0, longest = 0;
    for (uint32_t b = 0; b < n_blocks; ++b)
        if (!eblocks[b].has_haploid && eblocks[b].n_wah) {
            total += eblocks[b].n_wah;
            longest = eblocks[b].n_wah > longest ? eblocks[b].n_wah : longest;
            ++live;
        }
    uint64_t T = (total + n_groups - 1u) / n_groups;
    const bool whole_only = live <= n_groups || T < longest;  // (fewer blocks than groups: nothing to balance, and a chain longer than T cannot wrap)
    uint32_t g = 0, n = 0;
    uint64_t load = 0;
    item_begin[0] = 0;
    // a head part is written at the front of the NEXT group's list: kept aside until that list is opened
    MultiItem pending{};
    bool have_pending = false;
    auto open_next = [&]() {
        item_begin[++g] = n;
        load = 0;
        if (have_pending) {
            items[n++] = pending;
            load = pending.count;
            have_pending = false;
        }
    };
    for (uint32_t b = 0; b < n_blocks; ++b) {
        const uint32_t wl = (eblocks[b].has_haploid || !eblocks[b].n_wah) ? 0u : eblocks[b].n_wah;
        if (!wl) continue;
        if (whole_only) {
            if (load && load + wl > T && g + 1u < n_groups) open_next();
            items[n++] = MultiItem{b, 0u, wl, 0u};
            load += wl;
            continue;
        }
        for (;;) {
            const uint64_t room = T > load ? T - load : 0u;
            if (wl <= room || g + 1u >= n_groups) {  // fits (the last group takes what is left)
                items[n++] = MultiItem{b, 0u, wl, 0u};
                load += wl;
                break;
            }
            if (room < MIN_PART) {  // this group is full
                open_next();
                continue;
            }
            if (wl - room < MIN_PART) {  // a sliver would be left: the whole block here
                items[n++] = MultiItem{b, 0u, wl, 0u};
                load += wl;
                break;
            }
            // tail lines [wl - room, wl) close this group's list, head lines [0, wl - room) open the next one's
            const uint32_t cut = wl - (uint32_t)room;
            items[n++] = MultiItem{b, cut, (uint32_t)room, MULTI_ITEM_LOAD | (g << 8)};
            pending = MultiItem{b, 0u, cut, MULTI_ITEM_STORE | (g << 8)};
            have_pending = true;
            open_next();
            break;
        }
    }
    while (g < n_groups) open_next();
}

template <bool PROF>
__global__ void __launch_bounds__(1024) k_chain_rank_enc_multi(const EncBlock* __restrict__ eblocks, RankEncMultiArgs A) {
    constexpr uint32_t T = 1024, W = 16;
    constexpr int E = 64, G = 8, SMAX = 8;
    constexpr uint32_t SL_WORDS = 2048u;  // row words of a slice: two per thread
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t S = A.S;
    const uint32_t tab_bytes = S * SL_WORDS * 8u;
    uint32_t* wtot = reinterpret_cast<uint32_t*>(smem + tab_bytes);  // [0,64) totals for the scans, [64,80) list lengths of my waves
    uint32_t* ring_all = wtot + 128;                                 // [16 waves][128] ranks on their way to the lists
    const uint32_t tab_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
    if (tab_lds != 0u) __builtin_trap();  // table and bitmap start at LDS address 0 (gathers and deposits rely on it)
    const uint32_t N = A.N;
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const uint32_t w = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6));
    const uint32_t xcd = blockIdx.x & 7u, q = blockIdx.x >> 3;
    const uint32_t group = xcd * A.gpx + q / S, member = q % S;
    const uint32_t n_lists = S * W;
    if (A.test_desert && member == 1u) return;
    const uint32_t bmp_units = S * (SL_WORDS / 4u);  // 16-byte units of a bitmap

    const uint32_t row_bytes = ((N + 63u) / 64u) * 8u;  // bytes of an input row that hold haplotypes
    auto in_rsrc = [&](uint32_t line) -> v4u {
        const uint64_t base = reinterpret_cast<uint64_t>(A.src + (size_t)line * A.src_stride_w);
        v4u d;
        d[0] = (uint32_t)base;
        d[1] = (uint32_t)(base >> 32) & 0xFFFFu;
        d[2] = row_bytes;
        d[3] = 0x00020000u;
        return d;
    };
    auto group_rsrc = [&](const void* base, uint32_t bytes) -> __amdgpu_buffer_rsrc_t {
        const uint64_t v = reinterpret_cast<uint64_t>(base);
        const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
        const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32));
        return __builtin_amdgcn_make_buffer_rsrc((void*)(((uint64_t)hi << 32) | lo), 0, (int)bytes, 0x00020000);
    };
    const __amdgpu_buffer_rsrc_t rs_lflags = group_rsrc(A.list_flags + (size_t)group * 2u * S * 32u, 2u * S * 128u);
    const __amdgpu_buffer_rsrc_t rs_bmps = group_rsrc(A.bmps + (size_t)group * S * S * SL_WORDS, S * S * SL_WORDS * 4u);
    const __amdgpu_buffer_rsrc_t rs_slices = group_rsrc(A.slices + (size_t)group * S * 1024u, S * 16384u);
    uint32_t* const glists = A.lists + (size_t)group * 2u * n_lists * MULTI_LIST_CAP;
    uint64_t* const gbflags = A.flags + (size_t)group * 16u;
    uint64_t* const gsflags = gbflags + 8u;
    uint64_t t_start = 0;
    auto give_up = [&](uint32_t& spins) -> bool {  // every 32 polls: has the launch been aborted / has this wait run out?
        if ((++spins & 31u) != 0u) return false;
        if (__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(A.sync, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) return true;
        const uint64_t now = wall_clock64();
        if (!t_start) t_start = now;
        if (now - t_start > A.timeout_ticks) {
            __hip_atomic_store(A.sync, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return true;
        }
        return false;
    };
    // phase records of one wave (every other wave stores into one dummy slot behind the buffer: no branch)
    const bool profiling = PROF && blockIdx.x == (A.prof & 0xFFFFu) - 1u && w == (A.prof >> 16);
    uint32_t pidx = profiling ? 0u : A.prof_cap;
    const uint32_t pstep = profiling ? 1u : 0u;
    auto prof = [&](uint32_t i) {
        if constexpr (PROF) {
            const uint64_t now = wall_clock64();
            uint64_t* dst = A.prof_buf + (pidx < A.prof_cap ? pidx : A.prof_cap);
            if (lane == 0) *dst = ((uint64_t)i << 56) | (now & 0xFFFFFFFFFFFFFFull);
            pidx += pstep;
        }
    };
    using LdsV4 = __attribute__((address_space(3))) v4u;
    // every wave is done with the table: the bitmap takes its place
    auto clear_bitmap_begin = [&]() {  // (the caller places the barrier behind the stores)
        lds_barrier();
        uint32_t tid_here = tid, z0, z1, z2, z3;
        asm volatile("" : "+v"(tid_here));
        // the zeros are made HERE: as a constant the compiler kept one zero quad for the whole kernel - in scratch, and
        // loaded it back in front of every one of these stores, a round trip to memory each (four per line)
        asm volatile("v_mov_b32 %0, 0\n\tv_mov_b32 %1, 0\n\tv_mov_b32 %2, 0\n\tv_mov_b32 %3, 0" : "=v"(z0), "=v"(z1), "=v"(z2), "=v"(z3));
#pragma unroll
        for (int k = 0; k < SMAX / 2; ++k) {
            const uint32_t unit = (uint32_t)k * T + tid_here;
            if (unit < bmp_units) *reinterpret_cast<LdsV4*>((uintptr_t)(unit * 16u)) = v4u{z0, z1, z2, z3};
        }
    };
    auto clear_bitmap = [&]() {
        clear_bitmap_begin();
        lds_barrier();
    };
    auto deposit = [&](uint64_t xm, uint32_t rr) {
        if (xm) {
            if (__builtin_amdgcn_inverse_ballot_w64(xm)) {
                LdsU32* p = reinterpret_cast<LdsU32*>((uintptr_t)((rr >> 3) & 0x1FFFCu));
                __hip_atomic_fetch_or(p, 1u << (rr & 31u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
    };

    // ---- ONE hand-off: the row `rank` whose ones (zeros, if `dense`) the waves have published as lists under `seq`
    auto exchange_lists = [&](uint32_t rank, uint32_t seq, bool dense, uint32_t& Zout) -> bool {
        const uint32_t par = (uint32_t)__builtin_amdgcn_readfirstlane((int)(seq & 1u));
        uint32_t cnt_l = 0;
        {
            const uint32_t my_len = (uint32_t)__builtin_amdgcn_readfirstlane((int)wtot[64u + w]);
            if (lane == 0u) {
                u32x2 fl;
                fl[0] = my_len;
                fl[1] = seq;
                __builtin_amdgcn_raw_buffer_store_b64(fl, rs_lflags, w * 8u, (par * S + member) * 128u, 0);
            }
            // My workgroup's part of the hand-off first: its waves are done with the table (barrier), the bitmap that takes
            // the table's place is cleared - while the other members are still publishing (the wait below used to stand
            // in front of this; the clear's second barrier stands behind the first pieces' loads now).
            clear_bitmap_begin();
            const uint32_t ml = lane < S ? lane : 0u;
            uint32_t spins = 0;
            t_start = 0;
            for (;;) {
                const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(rs_lflags, ml * 128u + w * 8u, par * S * 128u, 16);
                if (__builtin_amdgcn_ballot_w64(v[1] != seq) == 0ull) {
                    cnt_l = lane < S ? v[0] : 0u;
                    break;
                }
                if (give_up(spins)) return false;
                __builtin_amdgcn_s_sleep(0);
            }
        }
        prof(2);  // wait for the lists
        // the first pieces travel while the bitmap is cleared
        const uint32_t* lst = glists + (size_t)(par * n_lists + w) * MULTI_LIST_CAP;  // + k * 16 lists: member k's
        uint32_t cnt[SMAX], longest = 0;
#pragma unroll
        for (int k = 0; k < SMAX; ++k) {
            cnt[k] = (uint32_t)__builtin_amdgcn_readlane((int)cnt_l, k);
            longest = cnt[k] > longest ? cnt[k] : longest;
        }
        v4u rk[SMAX];
        auto load_piece = [&](int k, uint32_t i0) {
            // beyond a list the range check returns 0 (no memory request), masked below
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
                (void*)(lst + (size_t)k * 16u * MULTI_LIST_CAP), 0, (int)(cnt[k] * 4u), 0x00020000);
            rk[k] = __builtin_amdgcn_raw_buffer_load_b128(rs, (i0 + lane * 4u) * 4u, 0, 16);
        };
#pragma unroll
        for (int k = 0; k < SMAX; ++k) {
            load_piece(k, 0u);
            __builtin_amdgcn_sched_barrier(0);
        }
        lds_barrier();  // every wave's part of the clear has landed
        prof(8);  // bitmap cleared
        // ---- every list, every rank
        for (uint32_t i0 = 0; i0 < longest; i0 += 256u) {
#pragma unroll
            for (int k = 0; k < SMAX; ++k) {
                if (i0 + lane * 4u < cnt[k]) {  // (a list's length is a multiple of four: a lane's four entries are all there or all beyond)
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const uint32_t e = rk[k][u];
                        LdsU32* p = reinterpret_cast<LdsU32*>((uintptr_t)((e >> 3) & 0x1FFFCu));
                        __hip_atomic_fetch_or(p, 1u << (e & 31u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                }
                load_piece(k, i0 + 256u);
            }
        }
        lds_barrier();  // the row is complete
        prof(9);  // lists applied
        // ---- the whole row -> table: thread t holds the 16-byte units t, t + 1024, ... (four words each)
        uint32_t tid_here = tid;
        asm volatile("" : "+v"(tid_here));
        v4u wv[SMAX / 2];
        uint32_t cu[SMAX / 2], inc[SMAX / 2];
#pragma unroll
        for (int i = 0; i < SMAX / 2; ++i) {
            const uint32_t unit = (uint32_t)i * T + tid_here;
            v4u v = v4u{0u, 0u, 0u, 0u};
            if (unit < bmp_units) v = *reinterpret_cast<const LdsV4*>((uintptr_t)(unit * 16u));
            if (dense) {  // the lists named the row's zeros: complement, positions at or beyond N stay zero
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const uint32_t p0 = (unit * 4u + (uint32_t)u) * 32u;
                    const uint32_t m = p0 + 32u <= N ? ~0u : (p0 >= N ? 0u : (1u << (N - p0)) - 1u);
                    v[u] = ~v[u] & m;
                }
            }
            wv[i] = v;
            cu[i] = (uint32_t)__popc(v[0]) + (uint32_t)__popc(v[1]) + (uint32_t)__popc(v[2]) + (uint32_t)__popc(v[3]);
            inc[i] = wave_scan_incl_dpp(cu[i]);
            if (lane == 63u) wtot[(uint32_t)i * W + w] = inc[i];
        }
        lds_barrier();  // every word of the row is in registers: the table may take its place; the wave totals are in LDS
        prof(10);  // row read + wave scans
        const uint32_t sc = wave_scan_incl_dpp(wtot[lane]);
        const uint32_t ones = (uint32_t)__builtin_amdgcn_readlane((int)sc, 63);
#pragma unroll
        for (int i = 0; i < SMAX / 2; ++i) {
            const uint32_t unit = (uint32_t)i * T + tid_here;
            const uint32_t seg = (uint32_t)i * W + w;
            const uint32_t base = seg ? (uint32_t)__builtin_amdgcn_readlane((int)sc, (int)seg - 1) : 0u;
            const uint32_t p0 = base + inc[i] - cu[i];
            const uint32_t p1 = p0 + (uint32_t)__popc(wv[i][0]);
            const uint32_t p2 = p1 + (uint32_t)__popc(wv[i][1]);
            const uint32_t p3 = p2 + (uint32_t)__popc(wv[i][2]);
            if (unit < bmp_units) {
                *reinterpret_cast<LdsV4*>((uintptr_t)(unit * 32u)) = v4u{wv[i][0], p0, wv[i][1], p1};
                *reinterpret_cast<LdsV4*>((uintptr_t)(unit * 32u + 16u)) = v4u{wv[i][2], p2, wv[i][3], p3};
                // my slice of the row for the WAH pass (rows are whole 16-byte units)
                if ((unit >> 9) == member && unit * 4u < A.dst_stride_w)
                    __builtin_nontemporal_store(wv[i], reinterpret_cast<v4u*>(A.dst + (size_t)rank * A.dst_stride_w + unit * 4u));
            }
        }
        Zout = N - ones;
        lds_barrier();
        prof(3);  // clear + apply + scan + table
        return true;
    };

    // ---- TWO hand-offs: the row `rank` whose ones the waves have just deposited into the bitmap
    auto exchange_bmp = [&](uint32_t rank, uint32_t seq, uint32_t& Zout) -> bool {
        uint32_t tid_here = tid;
        asm volatile("" : "+v"(tid_here));  // addresses are formed here, not kept (spilled) across the lines
        lds_barrier();  // every deposit has landed
        prof(11);  // clear + deposits
        {
            v4u b[SMAX / 2];
#pragma unroll
            for (int k = 0; k < SMAX / 2; ++k) {
                const uint32_t unit = (uint32_t)k * T + tid_here;
                b[k] = *reinterpret_cast<const LdsV4*>((uintptr_t)((unit < bmp_units ? unit : 0u) * 16u));
            }
#pragma unroll
            for (int k = 0; k < SMAX / 2; ++k) {
                const uint32_t unit = (uint32_t)k * T + tid_here;
                if (unit < bmp_units) __builtin_amdgcn_raw_buffer_store_b128(b[k], rs_bmps, unit * 16u, member * S * 8192u, 0);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // my part of the bitmap has left
        __syncthreads();
        if (tid == 0)
            *reinterpret_cast<volatile __attribute__((address_space(1))) uint64_t*>(
                (__attribute__((address_space(1))) uint64_t*)(gbflags + member)) = (uint64_t)seq;
        prof(4);  // clear + deposits + bitmap stored, flagged
        {
            const uint64_t* fp = gbflags + (lane < S ? lane : 0u);
            uint32_t spins = 0;
            t_start = 0;
            for (;;) {
                const uint64_t v = __hip_atomic_load(fp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (__builtin_amdgcn_ballot_w64((uint32_t)v != seq) == 0ull) break;
                if (give_up(spins)) return false;
                __builtin_amdgcn_s_sleep(0);
            }
        }
        prof(5);  // wait for the bitmaps
        uint32_t ones_slice;
        {
            u32x2 pc[SMAX];
#pragma unroll
            for (int k = 0; k < SMAX; ++k) {
                pc[k] = u32x2{0u, 0u};
                if ((uint32_t)k < S) pc[k] = __builtin_amdgcn_raw_buffer_load_b64(rs_bmps, member * 8192u + tid_here * 8u, (uint32_t)k * S * 8192u, 16);
            }
            __builtin_amdgcn_sched_barrier(0);  // all S loads in flight before the first is consumed
            uint2 v = make_uint2(0u, 0u);
#pragma unroll
            for (int k = 0; k < SMAX; ++k) {
                v.x |= pc[k][0];
                v.y |= pc[k][1];
            }
            const uint32_t c = (uint32_t)__popc(v.x) + (uint32_t)__popc(v.y);
            const uint32_t inc = wave_scan_incl_dpp(c);
            if (lane == 63u) wtot[w] = inc;
            lds_barrier();
            const uint32_t sc = row16_scan_incl(lane < W ? wtot[lane] : 0u);
            ones_slice = (uint32_t)__builtin_amdgcn_readlane((int)sc, W - 1);
            const uint32_t base = w ? (uint32_t)__builtin_amdgcn_readlane((int)sc, (int)w - 1) : 0u;
            const uint32_t pre0 = base + inc - c;
            v4u ent;
            ent[0] = v.x;
            ent[1] = pre0;
            ent[2] = v.y;
            ent[3] = pre0 + (uint32_t)__popc(v.x);
            __builtin_amdgcn_raw_buffer_store_b128(ent, rs_slices, tid_here * 16u, member * 16384u, 0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // my entries have left
            __syncthreads();
            if (tid == 0)
                *reinterpret_cast<volatile __attribute__((address_space(1))) uint64_t*>(
                    (__attribute__((address_space(1))) uint64_t*)(gsflags + member)) = ((uint64_t)seq << 32) | ones_slice;
            // my part of the row for the WAH pass: behind the flag, nobody in the chain waits for this store
            const uint32_t roww = member * SL_WORDS + 2u * tid_here;  // rows are whole 16-byte units
            if (roww < A.dst_stride_w) {
                typedef uint32_t v2u_nt __attribute__((ext_vector_type(2)));
                v2u_nt vv = {v.x, v.y};
                __builtin_nontemporal_store(vv, reinterpret_cast<v2u_nt*>(A.dst + (size_t)rank * A.dst_stride_w + roww));
            }
        }
        prof(6);  // slice ORed, scanned, stored, flagged
        uint32_t tot_l;
        {
            const uint64_t* fp = gsflags + (lane < S ? lane : 0u);
            uint32_t spins = 0;
            t_start = 0;
            for (;;) {
                const uint64_t v = __hip_atomic_load(fp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (__builtin_amdgcn_ballot_w64((uint32_t)(v >> 32) != seq) == 0ull) {
                    tot_l = lane < S ? (uint32_t)v : 0u;
                    break;
                }
                if (give_up(spins)) return false;
                __builtin_amdgcn_s_sleep(0);
            }
        }
        prof(5);  // wait for the slices
        const uint32_t incl = wave_scan_incl_dpp(tot_l);
        {
            v4u t[SMAX];
#pragma unroll
            for (int k = 0; k < SMAX; ++k) {
                t[k] = v4u{0u, 0u, 0u, 0u};
                if ((uint32_t)k < S) t[k] = __builtin_amdgcn_raw_buffer_load_b128(rs_slices, tid_here * 16u, (uint32_t)k * 16384u, 16);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = 0; k < SMAX; ++k)
                if ((uint32_t)k < S) {
                    const uint32_t b = k ? (uint32_t)__builtin_amdgcn_readlane((int)incl, k - 1) : 0u;
                    v4u e = t[k];
                    e[1] += b;
                    e[3] += b;
                    *reinterpret_cast<LdsV4*>((uintptr_t)((uint32_t)k * 16384u + tid_here * 16u)) = e;
                }
        }
        Zout = N - (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        lds_barrier();
        prof(6);  // table copied
        return true;
    };

    for (uint32_t i = tid; i < 128u; i += T) wtot[i] = 0;
    __syncthreads();
    // One-time handshake: the members of a group must share an XCD (see the header comment).
    {
        uint32_t xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        xcc = (xcc & 15u) + 1u;
        uint32_t* ids = A.xcc_ids + group * 8u;
        if (tid == 0) __hip_atomic_store(ids + member, xcc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        uint32_t spins = 0;
        t_start = 0;
        for (;;) {
            const uint32_t v = __hip_atomic_load(ids + (lane < S ? lane : member), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (__builtin_amdgcn_ballot_w64(v == 0u && lane != member) == 0ull) {
                if (__builtin_amdgcn_ballot_w64(v != xcc && lane != member) != 0ull) {  // spread over several XCDs: not this kernel's case
                    __hip_atomic_store(A.sync, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    return;
                }
                break;
            }
            if (give_up(spins)) return;
            __builtin_amdgcn_s_sleep(4);
        }
    }
    prof(7);
    uint32_t seq = 0;
    const uint32_t c0 = member * 1024u + w * (uint32_t)E;  // my first chunk of the row
    const uint32_t full_chunks = N >> 6, rem_bits = N & 63u;
    const uint32_t it_end = as_const(A.item_begin)[group + 1u];
    for (uint32_t it = as_const(A.item_begin)[group]; it < it_end; ++it) {
        // (blocks with fully haploid lines are not in the lists: the position-major kernels take them)
        const ConstU32* item = as_const(reinterpret_cast<const uint32_t*>(A.items + it));
        const uint32_t blk = item[0], part_first = item[1], n_wah = item[2], iflags = item[3];
        const EncBlock& B = eblocks[blk];
        const uint32_t wah_first = B.wah_first + part_first;
        const ConstU32* lines = as_const(A.wah_lines) + wah_first;
        uint32_t lane_here = lane;
        asm volatile("" : "+v"(lane_here));  // formed in this item's code, not kept across the items
        // (pointers and thread offsets of the parking are formed where they are used: kept across the line loop they cost it registers)
        auto park_of = [&](uint32_t fl) -> uint32_t* {
            uint32_t t = threadIdx.x;
            asm volatile("" : "+v"(t));
            return A.park + ((size_t)(fl >> 8) * S + member) * ((size_t)E * T) + t;
        };
        auto park_flag_of = [&](uint32_t fl) -> uint32_t* { return A.park_flags + (fl >> 8) * S + member; };  // (groups x S <= 256 flags)
        uint32_t r[E];
        if (iflags & MULTI_ITEM_LOAD) {
            uint32_t* const park_flag = park_flag_of(iflags);
            // the ranks behind the block's head lines, parked by the same member of another group (another XCD as a rule)
            uint32_t spins = 0;
            t_start = 0;
            while (__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(park_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == 0) {
                if (give_up(spins)) return;
                __builtin_amdgcn_s_sleep(8);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            const uint32_t* const park = park_of(iflags);
            static_for<0, E>([&](auto ec) {
                constexpr int e = decltype(ec)::value;
                r[e] = __builtin_nontemporal_load(park + (size_t)e * T);
            });
        } else {
            static_for<0, E>([&](auto ec) {
                constexpr int e = decltype(ec)::value;
                const uint32_t h = (c0 + (uint32_t)e) * 64u + lane_here;
                r[e] = h < N ? h : 0u;
            });
        }
        // ---- lists: ranks on their way to my wave's list pass through a LINEAR 128-entry buffer in LDS, so that they
        // leave as whole 256-byte stores (a store of one to three lanes per chunk is one fabric write per lane): the ranks
        // not yet stored sit at its front (fewer than 64), a chunk's new ones go behind them at `wpos` - a scalar LDS byte
        // address, so a lane's slot is v_mbcnt x 2 + ONE v_lshl_add - and a flush stores the first 64 and moves the rest down
        uint32_t n_out = 0;
        uint32_t* my_list = nullptr;
        uint32_t* ring = ring_all + w * 128u;
        const uint32_t ring_lds = (uint32_t)__builtin_amdgcn_readfirstlane(
            (int)(uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)reinterpret_cast<unsigned char*>(ring));
        uint32_t wpos = ring_lds;
        auto open_list = [&](uint32_t sq) {
            wpos = ring_lds;
            n_out = 0;
            my_list = glists + (size_t)((sq & 1u) * n_lists + member * W + w) * MULTI_LIST_CAP;
        };
        auto flush64 = [&]() {
            my_list[n_out + lane] = ring[lane];
            n_out += 64u;
            const uint32_t rest = (wpos - ring_lds - 256u) >> 2;  // entries behind the 64 that leave
            const uint32_t v = ring[64u + lane];
            if (lane < rest) ring[lane] = v;
            wpos -= 256u;
        };
        using LdsRing = __attribute__((address_space(3))) uint32_t;
        auto append = [&](uint64_t xm, uint32_t rr) {
            if (xm) {
                if (__builtin_amdgcn_inverse_ballot_w64(xm)) {
                    const uint32_t slot = __builtin_amdgcn_mbcnt_hi((uint32_t)(xm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)xm, 0u));
                    *reinterpret_cast<LdsRing*>((uintptr_t)(wpos + (slot << 2))) = rr;
                }
                wpos += 4u * (uint32_t)__popcll(xm);
                if (wpos >= ring_lds + 256u) flush64();
            }
        };
        // the rest of the list, padded to a whole 64-entry store with entries that mean nothing (~0); its length goes to
        // LDS for the flag; the wave waits for its stores
        auto publish = [&]() {
            const uint32_t left = (wpos - ring_lds) >> 2;
            if (left) {
                // The rest leaves as one more 64-entry store, but the list's LENGTH only grows to the next multiple of four
                // (readers take four entries per lane), the one to three entries of padding being copies of the last real
                // one: setting a bit twice changes nothing, so the readers need no test for padding (a third of the apply
                // loop's instructions).  (Padding the whole store with copies put up to 63 atomics on one LDS word: +40 ms.)
                const uint32_t v = ring[lane];
                const uint32_t last = (uint32_t)__builtin_amdgcn_readlane((int)v, (int)(left - 1u));  // (every lane enabled here)
                my_list[n_out + lane] = lane < left ? v : last;
                n_out += (left + 3u) & ~3u;
            }
            if (lane == 0) wtot[64u + w] = n_out;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        };
        auto complement = [&](uint64_t x, uint32_t chunk) -> uint64_t {
            uint32_t fc = full_chunks;
            asm volatile("" : "+s"(fc));  // the mask is formed where it is used (hoisted for all 64 chunks it spills 128 SGPRs)
            const uint64_t vm = chunk < fc ? ~0ull : (chunk == fc && rem_bits ? (1ull << rem_bits) - 1ull : 0ull);
            return ~x & vm;
        };
        // how the row of a line travels: 0 = bitmap, 1 = list of its ones, 2 = list of its zeros
        auto kind_of = [&](uint32_t line) -> uint32_t {
            const uint32_t c = as_const(A.cnt)[line];
            const bool dense = c * 2u > N;
            return (dense ? N - c : c) <= A.thr ? (dense ? 2u : 1u) : 0u;
        };
        // The deposit pass has nothing to hide a scalar load behind, so the key bits of group g + 1 travel while group g is
        // deposited.  Scalar loads return out of order - a wait for one is a wait for all - so the wait for group g stands
        // in FRONT of the request for g + 1; both are written out, because the compiler places a load next to its first use
        // and cannot be told that a wait has already happened.
        auto deposit_pass = [&](v4u rsn) {
            auto request = [&](uint32_t chunk) -> v16u {
                v16u v;
                asm volatile("s_buffer_load_dwordx16 %0, %1, %2" : "=&s"(v) : "s"(rsn), "s"(chunk * 8u));
                return v;
            };
            v16u cur = request(c0);
            static_for<0, E / G>([&](auto gc) {
                constexpr int g0 = decltype(gc)::value * G;
                asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(cur));
                v16u nxt = cur;
                if constexpr (g0 + G < E) nxt = request(c0 + (uint32_t)(g0 + G));
                static_for<0, G>([&](auto ec) {
                    constexpr int e = decltype(ec)::value;
                    deposit(((uint64_t)cur[2 * e + 1] << 32) | cur[2 * e], r[g0 + e]);
                });
                cur = nxt;
                __builtin_amdgcn_sched_barrier(0);
            });
        };
        // line 0: ranks are the identity, its row is the input row itself (bits at or beyond N are zero)
        uint32_t Z = 0;
        uint32_t kind = kind_of(lines[0]);
        ++seq;
        if (kind) {
            const v4u rs0 = in_rsrc(lines[0]);
            open_list(seq);
            static_for<0, E / G>([&](auto gc) {
                constexpr int g0 = decltype(gc)::value * G;
                uint64_t x0[G];
                sbuf_load_chunks<G>(rs0, c0 + (uint32_t)g0, x0);
                if (kind == 2u) {
                    static_for<0, G>([&](auto ec) {
                        constexpr int e = decltype(ec)::value;
                        x0[e] = complement(x0[e], c0 + (uint32_t)(g0 + e));
                    });
                }
                static_for<0, G>([&](auto ec) {
                    constexpr int e = decltype(ec)::value;
                    append(x0[e], r[g0 + e]);
                });
                __builtin_amdgcn_sched_barrier(0);
            });
            publish();
            if (!exchange_lists(wah_first, seq, kind == 2u, Z)) return;
        } else {
            clear_bitmap();
            deposit_pass(in_rsrc(lines[0]));
            if (!exchange_bmp(wah_first, seq, Z)) return;
        }
        auto prefetch_row = [&](uint32_t line) -> uint2 {  // my workgroup's 8 KiB of the input row, into L2
            const uint2* rowp = reinterpret_cast<const uint2*>(A.src + (size_t)line * A.src_stride_w) + member * 1024u;
            return (member * 1024u + tid) * 8u < row_bytes ? rowp[tid] : make_uint2(0u, 0u);
        };
        for (uint32_t j = 0; j < n_wah; ++j) {
            const bool more = j + 1u < n_wah;
            const uint2 pf = prefetch_row(lines[j + 2u < n_wah ? j + 2u : j]);
            const v4u rsc = in_rsrc(lines[j]);
            kind = more ? kind_of(lines[j + 1u]) : 0u;
            // ---- gathers + rank updates; when the next line travels as lists its flagged haplotypes are appended to my
            // list on the way (ONE loop for both kinds of line: an empty range reads as zeros, no appends - two copies of
            // the loop under a branch had their common head hoisted above it and 83 registers spilled)
            v4u rsn = in_rsrc(lines[more ? j + 1u : j]);
            if (!kind) rsn[2] = 0;
            open_list(seq + 1u);
            static_for<0, E / G>([&](auto gc) {
                constexpr int g0 = decltype(gc)::value * G;
                uint64_t xc[G], xn[G];
                u32x2 pr[G];
                sbuf_load_chunks<G>(rsc, c0 + (uint32_t)g0, xc);
                sbuf_load_chunks<G>(rsn, c0 + (uint32_t)g0, xn);
                static_for<0, G>([&](auto ec) {
                    constexpr int e = decltype(ec)::value;
                    pr[e] = *reinterpret_cast<const LdsU2*>((uintptr_t)((r[g0 + e] >> 2) & 0x1FFF8u));  // the table starts at LDS address 0
                });
                if (kind == 2u) {
                    static_for<0, G>([&](auto ec) {
                        constexpr int e = decltype(ec)::value;
                        xn[e] = complement(xn[e], c0 + (uint32_t)(g0 + e));
                    });
                }
                static_for<0, G>([&](auto ec) {
                    constexpr int e = decltype(ec)::value;
                    const uint32_t rr = r[g0 + e];
                    const uint32_t ob = (uint32_t)__popc(__builtin_amdgcn_ubfe(pr[e][0], 0u, rr)) + pr[e][1];
                    const uint32_t rn = __builtin_amdgcn_inverse_ballot_w64(xc[e]) ? Z + ob : rr - ob;
                    r[g0 + e] = rn;
                    append(xn[e], rn);
                });
                __builtin_amdgcn_sched_barrier(0);
            });
            asm volatile("" ::"v"(pf.x), "v"(pf.y));
            if (kind) {
                publish();
                prof(0);  // gathers + appends + publish
                ++seq;
                if (!exchange_lists(wah_first + j + 1u, seq, kind == 2u, Z)) return;
            } else {
                prof(1);  // gathers alone
                if (more) {
                    clear_bitmap();
                    deposit_pass(in_rsrc(lines[j + 1u]));
                    ++seq;
                    if (!exchange_bmp(wah_first + j + 1u, seq, Z)) return;
                }
            }
        }
        const uint32_t iflags_end = as_const(reinterpret_cast<const uint32_t*>(A.items + it))[3];  // (read again: not kept across the lines)
        if (iflags_end & MULTI_ITEM_STORE) {  // the block goes on in another group's list
            uint32_t* const park = park_of(iflags_end);
            static_for<0, E>([&](auto ec) {
                constexpr int e = decltype(ec)::value;
                park[(size_t)e * T] = r[e];
            });
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");  // (with the wait for my stores: visible beyond this XCD's L2)
            __syncthreads();
            if (threadIdx.x == 0) __hip_atomic_store(park_flag_of(iflags_end), 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
        // (the next item's first row: a list pass touches neither table nor bitmap, clear_bitmap starts with a barrier)
    }
    if constexpr (PROF) {
        if (profiling && lane == 0) A.sync[2] = pidx;
    }
}

bool chain_rank_enc_multi_supported(const EncLines& L) {
    const bool off = tuning_env("XSI_NO_RANKENC_MULTI") != nullptr;  // read per call (tests force the other kernel)
    return !off && !L.no_multi && L.chain_sync && L.chain_lists && L.chain_slices && L.chain_bmps && L.chain_items && L.chain_park && L.N > 65536u && L.N <= 524288u &&
           (L.y_stride64 % 2u) == 0u;
}

static hipError_t launch_rank_encode_multi_grid(hipStream_t s, const EncBlock* blocks, uint32_t n_blocks, const EncLines& L, int cus, bool* refused) {
    // `refused` is raised by the checks that stand in FRONT of the first enqueue only: an error of
    // hipFuncSetAttribute or of the launch itself is a bug or a broken device and travels up as what it is
    *refused = false;
    RankEncMultiArgs A{};
    A.wah_lines = L.wah_lines;
    A.src = L.planes;
    A.src_stride_w = L.plane_stride_w;
    A.cnt = L.cnt;
    A.dst = reinterpret_cast<uint32_t*>(L.yrows);
    A.dst_stride_w = L.y_stride64 * 2u;
    A.N = L.N;
    A.n_blocks = n_blocks;
    A.S = (L.N + 65535u) / 65536u;
    A.gpx = (uint32_t)cus / 8u / A.S;
    if (A.gpx < 1u) {
        *refused = true;
        return hipSuccess;
    }
    while (A.gpx > 1u && 8u * (A.gpx - 1u) >= n_blocks) --A.gpx;  // no more groups than blocks need
    if (8u * A.gpx * A.S > CHAIN_MAX_WGS || 8u * A.gpx + 1u > CHAIN_ITEM_BEGIN_WORDS || !L.chain_items || !L.chain_park) {
        *refused = true;
        return hipSuccess;
    }
    static_assert(CHAIN_SLICEFLAG_WORDS >= (CHAIN_MAX_WGS / 2u) * 32u, "16 flags of 8 bytes per group");
    const char* thr = tuning_env("XSI_MULTI_LIST_THR");
    A.thr = thr ? (uint32_t)atoi(thr) : 49152u;
    A.sync = L.chain_sync;
    A.list_flags = L.chain_sync + CHAIN_SYNC_WORDS;
    A.flags = reinterpret_cast<uint64_t*>(L.chain_sync + CHAIN_SYNC_WORDS + CHAIN_LISTFLAG_WORDS);
    A.lists = L.chain_lists;
    A.bmps = L.chain_bmps;
    A.slices = reinterpret_cast<v4u*>(L.chain_slices);
    A.test_desert = test_hook_env("XSI_MULTI_TEST_DESERT") ? 1u : 0u;
    A.prof = tuning_env("XSI_MULTI_PROF") ? (uint32_t)atoi(tuning_env("XSI_MULTI_PROF")) : 0u;
    A.xcc_ids = L.chain_sync + CHAIN_SYNC_WORDS + CHAIN_LISTFLAG_WORDS + CHAIN_SLICEFLAG_WORDS;
    const char* tmo = tuning_env("XSI_MULTI_TIMEOUT_MS");
    A.timeout_ticks = 100000ull * (uint64_t)(tmo && atoi(tmo) > 0 ? atoi(tmo) : 2000);
    hipError_t e = hipMemsetAsync(L.chain_sync, 0, 4ull * CHAIN_SYNC_TOTAL_WORDS, s);
    if (e != hipSuccess) return e;
    const uint32_t n_groups = 8u * A.gpx;
    static_assert(CHAIN_SYNC_WORDS >= 16u + CHAIN_MAX_WGS, "a parking flag per workgroup of the launch (group x member)");
    static_assert(CHAIN_ITEM_BEGIN_WORDS % 4u == 0u && CHAIN_ITEM_BEGIN_WORDS >= CHAIN_MAX_WGS / 2u + 1u,
                  "per-group begins (at most 128 groups: two workgroups per block), the items behind them 16-byte aligned");
    A.item_begin = L.chain_items;
    A.items = reinterpret_cast<const MultiItem*>(L.chain_items + CHAIN_ITEM_BEGIN_WORDS);
    A.park = L.chain_park;
    A.park_flags = L.chain_sync + 16u;
    k_multi_schedule<<<dim3(1), dim3(64), 0, s>>>(blocks, n_blocks, n_groups, L.chain_items, reinterpret_cast<MultiItem*>(L.chain_items + CHAIN_ITEM_BEGIN_WORDS),
                                                  tuning_env("XSI_MULTI_ROUND_ROBIN") ? 1u : 0u);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    static uint64_t* prof_buf = nullptr;  // (profiling runs only; never freed)
    constexpr uint32_t PROF_CAP = 1u << 20;
    if (A.prof && !prof_buf && hipMalloc(&prof_buf, 8ull * PROF_CAP + 8) != hipSuccess) return hipErrorOutOfMemory;
    A.prof_buf = prof_buf;
    A.prof_cap = PROF_CAP;
    const uint32_t lds = A.S * 16384u + 512u + 16u * 128u * 4u;
    auto kern = A.prof ? &k_chain_rank_enc_multi<true> : &k_chain_rank_enc_multi<false>;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    kern<<<dim3(8u * A.gpx * A.S), dim3(1024), lds, s>>>(blocks, A);
    e = hipGetLastError();
    if (A.prof && e == hipSuccess) {  // phase report of the chosen wave: per tag the time that ended with it
        e = hipStreamSynchronize(s);
        uint32_t n = 0;
        if (e == hipSuccess) e = hipMemcpy(&n, L.chain_sync + 2, 4, hipMemcpyDeviceToHost);
        if (n > PROF_CAP) n = PROF_CAP;
        std::vector<uint64_t> rec(n);
        if (e == hipSuccess && n) e = hipMemcpy(rec.data(), prof_buf, 8ull * n, hipMemcpyDeviceToHost);
        static const char* nm[12] = {"gathers+appends+publish", "gathers alone", "barrier + clear issued + wait lists", "table entries written",
                                     "bitmap stored + flagged", "waits (bitmap form)", "slice / table copy", "start",
                                     "clear landed (barrier)", "lists applied", "row read + wave scans", "clear + deposits"};
        uint64_t sum[12] = {0}, cntv[12] = {0};
        for (uint32_t i = 1; i < n; ++i) {
            const uint32_t tag = (uint32_t)(rec[i] >> 56) % 12u;
            sum[tag] += (rec[i] - rec[i - 1]) & 0xFFFFFFFFFFFFFFull;
            cntv[tag]++;
        }
        for (int t = 0; t < 12; ++t)
            fprintf(stderr, "[xsi multi prof] %-28s %9.3f ms  (%llu records, %.2f us each)\n", nm[t], sum[t] * 1e-5,
                    (unsigned long long)cntv[t], cntv[t] ? sum[t] * 1e-2 / cntv[t] : 0.0);
    }
    return e;
}

hipError_t launch_rank_encode_multi(hipStream_t s, const EncBlock* blocks, uint32_t n_blocks, const EncLines& L, bool* refused) {
    *refused = false;
    if (!n_blocks) return hipSuccess;
    int dev = 0, cus = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    if (e != hipSuccess) return e;
    return launch_rank_encode_multi_grid(s, blocks, n_blocks, L, cus, refused);
}

}  // namespace xsi
