// xsi_rank.hip — element-major ("rank tracking") PBWT decode chain for gfx950.
// Split from xsi_kernels.hip so the two big kernel families compile in parallel.
#include "xsi_kernels.hpp"

#include <cstdio>

#include <cstdlib>
#include <type_traits>

#include "xsi_device.hpp"

namespace xsi {

static uint32_t next_pow2_log2(uint32_t v) {
    uint32_t l = 0;
    while ((1u << l) < v) ++l;
    return l;
}

// ------------------------------------------------------------------------------------------
// PBWT chain, decode side, element-major ("rank tracking").
//
// The position-major kernel above moves the prefix array `a` around and needs two workgroup
// barriers per line.  Decode does not need `a` at all.  Let r_k(h) be the position of haplotype
// h in a_k (the inverse permutation, r_0(h) = h).  Then, with y_k the stored permuted row,
//     x_k(h)     = y_k[r_k(h)]                                   (accessor_internals_new.hpp:228-230)
//     r_{k+1}(h) = x_k(h) ? Z_k + ones_k(r_k(h)) : r_k(h) - ones_k(r_k(h))      (gt_block.hpp:124-136)
// where ones_k(r) = number of set bits of y_k before position r and Z_k = zeros of y_k: the stable
// partition moves a zero at position r to (zeros before r) and a one to Z + (ones before r).
// Every haplotype's rank evolves on its own from read-only data (y_k plus a per-32-bit prefix
// popcount, both produced by k_wah_expand), so there is NO communication between threads: no
// barrier per line, no scatter, no atomics, and the haplotypes of one block can be split over
// several workgroups to fill all 256 CUs even when there are fewer blocks than CUs.
// 64 consecutive haplotypes live in one wave chunk, so their decoded bits are one ballot = one
// 64-bit word of the natural-order output row.
// Blocks that contain fully haploid lines keep the position-major kernel (a haploid line orders
// y by the even members of `a`, which needs the permutation itself).
// ------------------------------------------------------------------------------------------
typedef uint32_t rank_u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t rank_v4u __attribute__((ext_vector_type(4)));
typedef uint32_t rank_v16u __attribute__((ext_vector_type(16)));

struct RankArgs {
    const DecBlock* blocks;
    const uint32_t* wah_lines;  // [rank] binary line
    const uint2* yp;            // [rank][yp_stride] {bits, ones before}
    uint32_t yp_stride;
    const uint32_t* wah_z;      // [rank] zeros of the line
    uint32_t* out;              // output rows by binary line
    uint32_t out_stride_w;
    uint32_t N;
    uint32_t batch;             // lines staged per LDS batch
    uint32_t log2_cwp;
    // phased decode: this launch runs ph_cnt[b] lines of block b from batch-wide rank ph_start[b]; ranks come from /
    // go to `state` unless the range holds the block's first / last line.  nullptr: all lines.
    const uint32_t* ph_start;
    const uint32_t* ph_cnt;
    uint32_t* state;            // [block][chunk of the wave][1024 threads]
    // compact rank-select rows (DecLines::yp_compact): 64-bit chunks + 16-bit ones-before, nullptr = pairs in yp
    const uint2* yc;            // [rank][yc_stride] chunks
    const uint16_t* ypre;       // [rank][yc_stride]
    uint32_t yc_stride;
    uint32_t big_splits, big_n_blocks;  // k_chain_decode_rank_big: workgroups per block, blocks of the launch
    uint32_t big_prof;
    uint32_t yp_rev;            // rows in the reversed form (DecLines::yp_rev)
};

constexpr int RANK_RP = 8;  // {bits, prefix} pairs a thread carries while a batch is in flight

template <int T, int E, bool STAGE>
__global__ void __launch_bounds__(T) k_chain_decode_rank(RankArgs A) {
    constexpr int W = T / 64;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const DecBlock& D = A.blocks[blockIdx.x];
    if (D.error || D.n_wah == 0 || D.off_line_haploid != VAL_UNDEFINED) return;
    const uint32_t N = A.N;
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const uint32_t w = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6));
    const uint32_t cg0 = (blockIdx.y * W + w) * E;  // first chunk of my wave
    const uint32_t wah_first = A.ph_start ? A.ph_start[blockIdx.x] : D.wah_first;
    const uint32_t n_wah = A.ph_start ? A.ph_cnt[blockIdx.x] : D.n_wah;
    if (n_wah == 0) return;  // no line of this block in this range: its ranks stay parked
    const uint32_t CWP = A.yp_stride;

    uint32_t r[E];
    uint32_t* park = A.state + (((size_t)blockIdx.x * gridDim.y + blockIdx.y) * (uint32_t)E) * T + tid;  // chunk e: park[e * T]
    // the block's first line is in this range: identity; else the ranks parked by the launch of the range before
    if (!A.ph_start || wah_first == D.wah_first) {
        static_for<0, E>([&](auto ecn) {
            constexpr int e = decltype(ecn)::value;
            r[e] = (cg0 + (uint32_t)e) * 64u + lane;
            if (r[e] >= N) r[e] = 0;  // haplotypes beyond N idle on position 0; their output is masked
        });
    } else {
        static_for<0, E>([&](auto ecn) {
            constexpr int e = decltype(ecn)::value;
            r[e] = park[(size_t)e * T];
        });
    }
    const bool park_after = A.ph_start && wah_first + n_wah != D.wah_first + D.n_wah;
    // lane e (< E) stores chunk cg0+e's word; valid-bit mask of that chunk for the row tail
    uint32_t vm_lo = 0, vm_hi = 0;
    {
        const uint64_t base = (uint64_t)(cg0 + lane) * 64u;
        if (lane < (uint32_t)E && base < N) {
            const uint32_t nv = (N - base >= 64u) ? 64u : (uint32_t)(N - base);
            const uint64_t vm = nv == 64u ? ~0ull : ((1ull << nv) - 1ull);
            vm_lo = (uint32_t)vm;
            vm_hi = (uint32_t)(vm >> 32);
        }
    }
    // rows are padded (e.g. to 128 bytes): the words past the last chunk must read as zero.  The wave
    // that owns the last chunk also stores the (all-zero: their valid mask is 0) padding chunks behind
    // it with the same 8-byte-per-lane store; only when the padding does not fit its 64 lanes does the
    // first workgroup of the block zero it with a separate loop.
    const uint32_t nch = (N + 63u) / 64u;
    const uint32_t row_words = nch * 2u;
    const uint32_t row_chunks = A.out_stride_w / 2u;  // 8-byte words per output row
    const bool owns_tail = cg0 < nch && nch <= cg0 + (uint32_t)E;
    const uint32_t tail_cg0 = (nch - 1u) / (uint32_t)E * (uint32_t)E;  // first chunk of the wave that owns the tail
    const bool tail_fits = (A.out_stride_w & 1u) == 0u && row_chunks - tail_cg0 <= 64u;
    const bool store_lane = (lane < (uint32_t)E && cg0 + lane < nch) ||
                            (owns_tail && tail_fits && cg0 + lane >= nch && cg0 + lane < row_chunks);
    const bool pad_writer = blockIdx.y == 0 && A.out_stride_w > row_words && !tail_fits;

    if (!STAGE) {
        // rows too long for LDS (N > 65536): rank-select straight from L2
        for (uint32_t j = 0; j < n_wah; ++j) {
            const uint32_t rank = wah_first + j;
            const uint2* row = A.yp + (size_t)rank * CWP;
            const uint32_t Z = A.wah_z[rank];
            const uint32_t line = A.wah_lines[rank];
            uint32_t mine_lo = 0, mine_hi = 0;
            static_for<0, E>([&](auto ecn) {
                constexpr int e = decltype(ecn)::value;
                const uint2 pr = row[r[e] >> 5];
                const uint32_t bit = __builtin_amdgcn_ubfe(pr.x, r[e], 1u);
                const uint32_t ob = pr.y + (uint32_t)__popc(pr.x & ((1u << (r[e] & 31u)) - 1u));
                r[e] = bit ? Z + ob : r[e] - ob;
                const uint64_t m = __ballot(bit != 0u);
                mine_lo = write_lane(mine_lo, (uint32_t)m, (uint32_t)e);
                mine_hi = write_lane(mine_hi, (uint32_t)(m >> 32), (uint32_t)e);
            });
            if (store_lane) {
                uint2* orow = reinterpret_cast<uint2*>(A.out + (size_t)line * A.out_stride_w);
                orow[cg0 + lane] = make_uint2(mine_lo & vm_lo, mine_hi & vm_hi);
            }
            if (pad_writer)
                for (uint32_t i = row_words + tid; i < A.out_stride_w; i += T) A.out[(size_t)line * A.out_stride_w + i] = 0;
        }
        if (park_after) {
            static_for<0, E>([&](auto ecn) {
                constexpr int e = decltype(ecn)::value;
                park[(size_t)e * T] = r[e];
            });
        }
        return;
    }

    // Staged path.  Loads and stores share one in-order counter (vmcnt), so a wait for prefetched rows that
    // comes behind output stores also waits for those stores to complete.  Hence: a line's output words go to
    // LDS, a batch's outputs are flushed right after the barrier that ends it, and the loads of the batch after
    // next follow the flush - when they are waited for, a whole batch later, the flush has long completed.
    // All loads are unconditional (clamped addresses): under a branch the compiler drains vmcnt at the loop head.
    constexpr uint32_t CH = (uint32_t)W * E;                             // chunks of this workgroup
    uint2* stage = reinterpret_cast<uint2*>(smem);                       // 2 x B x CWP pairs
    const uint32_t B = A.batch;
    uint32_t* meta = reinterpret_cast<uint32_t*>(stage + 2u * B * CWP);  // 3 x B x {line, Z}
    uint2* obuf = reinterpret_cast<uint2*>(meta + 6u * B);               // 2 x B x CH output words
    const uint32_t cwp_mask = (1u << A.log2_cwp) - 1u;
    const uint32_t n_batches = (n_wah + B - 1u) / B;
    uint2 R[RANK_RP];
    uint2 Rm = make_uint2(0, 0);
    auto load_batch = [&](uint32_t bt) {
        {
            const uint32_t j = bt * B + (tid < B ? tid : 0u);
            const uint32_t rank = wah_first + (j < n_wah ? j : n_wah - 1u);
            Rm = make_uint2(A.wah_lines[rank], A.wah_z[rank]);
        }
#pragma unroll
        for (int q = 0; q < RANK_RP; ++q) {
            const uint32_t idx = (uint32_t)q * T + tid;
            const uint32_t jj = idx >> A.log2_cwp, wi = idx & cwp_mask;
            const uint32_t j = bt * B + jj;
            const bool ok = jj < B && j < n_wah && wi < CWP;
            R[q] = A.yp[(size_t)wah_first * CWP + (ok ? j * CWP + wi : 0u)];
        }
    };
    auto store_batch = [&](uint32_t bt) {
        const uint32_t buf = bt & 1u, mb = bt % 3u;
        // every loaded register is consumed on every path: what a skipped branch leaves "maybe pending" turns into
        // vmcnt bounds at later writes of these registers, inside the line loop
        asm volatile("" ::"v"(Rm.x), "v"(Rm.y));
#pragma unroll
        for (int q = 0; q < RANK_RP; ++q) asm volatile("" ::"v"(R[q].x), "v"(R[q].y));
        if (tid < B) {
            meta[(mb * B + tid) * 2u] = Rm.x;
            meta[(mb * B + tid) * 2u + 1u] = Rm.y;
        }
#pragma unroll
        for (int q = 0; q < RANK_RP; ++q) {
            const uint32_t idx = (uint32_t)q * T + tid;
            const uint32_t jj = idx >> A.log2_cwp, wi = idx & cwp_mask;
            if (jj < B && wi < CWP) stage[(buf * B + jj) * CWP + wi] = R[q];
        }
    };
    // ranks restored from `park` are waited for here: left pending into the loop, every use of r[] in it carries a
    // vmcnt bound that also covers the flush stores and prefetch loads of the batch
#pragma unroll
    for (int e = 0; e < E; ++e) asm volatile("" : "+v"(r[e]));
    const uint32_t wg_c0 = blockIdx.y * CH;  // first chunk of this workgroup
    const bool last_split = blockIdx.y + 1u == gridDim.y;
    auto flush = [&](uint32_t bt, uint32_t jn) {
        const uint32_t buf = bt & 1u, mb = bt % 3u;
        for (uint32_t idx = tid; idx < jn * CH; idx += T) {
            const uint32_t jj = idx / CH, c = idx - jj * CH;
            if (wg_c0 + c < row_chunks) {
                const uint32_t line = meta[(mb * B + jj) * 2u];
                reinterpret_cast<uint2*>(A.out + (size_t)line * A.out_stride_w)[wg_c0 + c] = obuf[(buf * B + jj) * CH + c];
            }
        }
        // rows padded beyond the chunks any workgroup holds: zeroed by the last split (rare geometry)
        if (last_split && row_chunks > gridDim.y * CH)
            for (uint32_t jj = 0; jj < jn; ++jj) {
                const uint32_t line = meta[(mb * B + jj) * 2u];
                for (uint32_t i = 2u * gridDim.y * CH + tid; i < A.out_stride_w; i += T) A.out[(size_t)line * A.out_stride_w + i] = 0;
            }
    };
    load_batch(0);
    store_batch(0);
    __syncthreads();
    for (uint32_t bt = 0; bt < n_batches; ++bt) {
        const uint32_t bt_n = bt + 1u < n_batches ? bt + 1u : bt;  // the last batch fetches itself again
        load_batch(bt_n);
        const uint32_t jn = (n_wah - bt * B) < B ? (n_wah - bt * B) : B;
        const uint32_t buf = bt & 1u, mb = bt % 3u;
        uint32_t Zv = meta[mb * B * 2u + 1u];  // zeros of the next line, read one line ahead
        for (uint32_t jj = 0; jj < jn; ++jj) {
            const uint2* row = stage + (buf * B + jj) * CWP;
            const uint32_t Z = (uint32_t)__builtin_amdgcn_readfirstlane((int)Zv);
            Zv = meta[(mb * B + (jj + 1u < jn ? jj + 1u : jj)) * 2u + 1u];
            uint2 pr[E];
            static_for<0, E>([&](auto ecn) {
                constexpr int e = decltype(ecn)::value;
                pr[e] = row[r[e] >> 5];
            });
            uint32_t mine_lo = 0, mine_hi = 0;
            static_for<0, E>([&](auto ecn) {
                constexpr int e = decltype(ecn)::value;
                const uint32_t bit = __builtin_amdgcn_ubfe(pr[e].x, r[e], 1u);
                const uint32_t ob = pr[e].y + (uint32_t)__popc(pr[e].x & ((1u << (r[e] & 31u)) - 1u));
                r[e] = bit ? Z + ob : r[e] - ob;
                const uint64_t m = __ballot(bit != 0u);
                mine_lo = write_lane(mine_lo, (uint32_t)m, (uint32_t)e);
                mine_hi = write_lane(mine_hi, (uint32_t)(m >> 32), (uint32_t)e);
            });
            if (lane < (uint32_t)E) obuf[(buf * B + jj) * CH + w * E + lane] = make_uint2(mine_lo & vm_lo, mine_hi & vm_hi);
        }
        store_batch(bt + 1u);  // (after the last batch: a copy nobody reads)
        __syncthreads();
        flush(bt, jn);
    }
    if (park_after) {
        static_for<0, E>([&](auto ecn) {
            constexpr int e = decltype(ecn)::value;
            park[(size_t)e * T] = r[e];
        });
    }
}

// Long rows (N > 65536, up to 1024*20*32 haplotypes): one {bits, prefix} row at a time in LDS
// (up to 160 KB), the next row prefetched into registers (RP pairs per thread) while the current
// one is used.  A gather from LDS serves 64 lanes in a few cycles; the same gather from L2 is one
// request per lane (~1 lane per clock per CU), which is what bounded the unstaged path.  Each
// workgroup covers 16*E chunks of haplotypes; the splits of one block are dispatched 8 workgroup ids apart, i.e. to
// the same XCD, and share the row through that XCD's L2 (see the kernel's first lines).
__device__ unsigned long long g_big_prof[4];  // XSI_BIG_PROF: 100 MHz ticks of workgroup 0, wave 0: main, barrier, row to LDS + stores + barrier

// REV: the rows come as {bits reversed within the word, -(ones up to the word's end)} (DecLines::yp_rev): shifted left by
// my position the word has MY bit on top and the positions behind me below it, so one v_lshlrev serves both the bit
// (sign) and the count (v_bcnt with the negated prefix gives -(ones before me)): 10 instead of 11 vector instructions a chunk.
template <int E, int RP, bool PROF = false, bool REV = false>
__global__ void __launch_bounds__(1024) k_chain_decode_rank_big(RankArgs A) {
    // G: gathers in flight per wave (4 where 64 ranks and 32 prefetch registers leave no room for 8 pairs)
    constexpr int T = 1024, W = 16, G = (E == 64 && RP >= 16) ? 4 : 8;
    static_assert(E % G == 0, "E must be a multiple of the gather group");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // Workgroups are dealt to the 8 XCDs round-robin by their linear id, and every XCD has its own L2.  The splits of
    // a block all stage the same 125 KB row per line: spread over the XCDs (a (splits, blocks) grid does that) each
    // L2 fetches the row from memory for itself; taken 8 ids apart they share one L2 and the row is fetched once.
    const uint32_t xcd = blockIdx.x & 7u, slot = blockIdx.x >> 3;
    const uint32_t blk_split = slot % A.big_splits, blk_y = (slot / A.big_splits) * 8u + xcd;
    if (blk_y >= A.big_n_blocks) return;
    const DecBlock& D = A.blocks[blk_y];
    if (D.error || D.n_wah == 0 || D.off_line_haploid != VAL_UNDEFINED) return;
    const uint32_t N = A.N;
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const uint32_t w = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6));
    const uint32_t cg0 = (blk_split * W + w) * E;  // first chunk of my wave
    const uint32_t wah_first = A.ph_start ? A.ph_start[blk_y] : D.wah_first;
    const uint32_t n_wah = A.ph_start ? A.ph_cnt[blk_y] : D.n_wah;
    if (n_wah == 0) return;  // no line of this block in this range: its ranks stay parked
    const uint32_t CWP = A.yp_stride;
    uint2* row = reinterpret_cast<uint2*>(smem);
    const uint32_t tab_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
    if (tab_lds != 0u) __builtin_trap();  // (the gathers address the row from 0)
    using LdsPairBig = __attribute__((address_space(3))) rank_u32x2;

    uint32_t r[E];
    uint32_t* park = A.state + (((size_t)blk_y * A.big_splits + blk_split) * (uint32_t)E) * T + tid;  // chunk e: park[e * T]
    // the block's first line is in this range: identity; else the ranks parked by the launch of the range before
    // (a short block has empty ranges: "first" and "last" are the block's own, not the launch's)
    if (!A.ph_start || wah_first == D.wah_first) {
        static_for<0, E>([&](auto ecn) {
            constexpr int e = decltype(ecn)::value;
            r[e] = (cg0 + (uint32_t)e) * 64u + lane;
            if (r[e] >= N) r[e] = 0;
        });
    } else {
        static_for<0, E>([&](auto ecn) {
            constexpr int e = decltype(ecn)::value;
            r[e] = park[(size_t)e * T];
        });
    }
    uint32_t vm_lo = 0, vm_hi = 0;
    {
        const uint64_t base = (uint64_t)(cg0 + lane) * 64u;
        if (lane < (uint32_t)E && base < N) {
            const uint32_t nv = (N - base >= 64u) ? 64u : (uint32_t)(N - base);
            const uint64_t vm = nv == 64u ? ~0ull : ((1ull << nv) - 1ull);
            vm_lo = (uint32_t)vm;
            vm_hi = (uint32_t)(vm >> 32);
        }
    }
    // padding chunks behind the last real one: stored by the wave that owns the tail (see above)
    const uint32_t nch = (N + 63u) / 64u;
    const uint32_t row_words = nch * 2u;
    const uint32_t row_chunks = A.out_stride_w / 2u;
    const bool owns_tail = cg0 < nch && nch <= cg0 + (uint32_t)E;
    const uint32_t tail_cg0 = (nch - 1u) / (uint32_t)E * (uint32_t)E;
    const bool tail_fits = (A.out_stride_w & 1u) == 0u && row_chunks - tail_cg0 <= 64u;
    const bool store_lane = (lane < (uint32_t)E && cg0 + lane < nch) ||
                            (owns_tail && tail_fits && cg0 + lane >= nch && cg0 + lane < row_chunks);
    const bool pad_writer = blk_split == 0 && A.out_stride_w > row_words && !tail_fits;

    // the row travels as 16-byte pieces (two pairs): 8-byte accesses reach 0.54-0.70 of the 16-byte rate on this chip
    // (MI355X_MICROARCH.md); rows are whole 16-byte units (yp_stride is even).  Piece q of a row = units q T .. q T + T - 1,
    // one per thread; only the last piece of a row is partial.  No per-piece offset registers, no exec masks (8 + 5 VGPRs
    // and 8 saved exec masks before): a piece's base is scalar arithmetic over ONE lane offset, the threads beyond the
    // row's end read on into what follows it (the next row; the buffer ends with 16 KiB of slack, decode_prepare) and
    // store that behind the row's end in LDS (the launch sizes LDS in whole pieces); an instantiation with more pieces
    // than the row has fetches the last one again (unconditional loads stay in flight across the line).
    static_assert(RP % 2 == 0, "pairs are staged two at a time");
    typedef uint32_t row_u32x4 __attribute__((ext_vector_type(4)));
    row_u32x4 R[RP / 2];
    const uint32_t CWP2 = CWP / 2u;
    const uint32_t q_last = (CWP2 - 1u) / T;  // uniform: the partial (or last whole) piece
    const uint32_t voff_tid = tid * 16u;
    auto load_row = [&](uint32_t j) {
        const unsigned char* src = reinterpret_cast<const unsigned char*>(A.yp + (size_t)(wah_first + j) * CWP);
#pragma unroll
        for (int q = 0; q < RP / 2; ++q) {
            const unsigned char* sq = src + (size_t)((uint32_t)q < q_last ? (uint32_t)q : q_last) * (T * 16u);  // scalar
            R[q] = *reinterpret_cast<const row_u32x4*>(sq + voff_tid);  // scalar base + 32-bit lane offset
        }
    };
    auto store_row = [&]() {
#pragma unroll
        for (int q = 0; q < RP / 2; ++q)
            if ((uint32_t)q <= q_last) reinterpret_cast<row_u32x4*>(row)[(uint32_t)q * T + tid] = R[q];  // uniform condition
    };
    load_row(0);
    __builtin_amdgcn_sched_barrier(0);  // (all pieces requested before the first is parked)
    store_row();
    uint32_t line = A.wah_lines[wah_first], Z = A.wah_z[wah_first];
    __syncthreads();
    const bool prof_on = PROF && A.big_prof && blockIdx.x == 0 && w == 0u;  // (compiled out of the production instantiations)
    uint64_t t_prof = prof_on ? wall_clock64() : 0;
    auto prof = [&](int i) {
        if constexpr (!PROF) return;
        if (prof_on) {
            const uint64_t now = wall_clock64();
            if (lane == 0) atomicAdd(&g_big_prof[i], (unsigned long long)(now - t_prof));
            t_prof = now;
        }
    };
    for (uint32_t j = 0; j < n_wah; ++j) {
        // unconditional prefetch (the last line fetches its own row again), see k_chain_decode_rank_wg
        const uint32_t jn = j + 1u < n_wah ? j + 1u : j;
        load_row(jn);
        const uint32_t line_n = A.wah_lines[wah_first + jn];
        const uint32_t Z_n = A.wah_z[wah_first + jn];
        uint32_t mine_lo = 0, mine_hi = 0;
        const uint32_t Zs = (uint32_t)__builtin_amdgcn_readfirstlane((int)Z);  // wave-uniform: keep it scalar
        static_for<0, E / G>([&](auto gcn) {
            constexpr int g0 = decltype(gcn)::value * G;
            // pins this group's gathers behind the previous group's rank updates (see k_chain_decode_rank_wg):
            // left alone the compiler issues the gathers of all groups first and spills.  The row sits at LDS address 0
            // (the kernel has no other LDS; checked at its start), so the entry's offset IS the address: no add per gather.
            asm volatile("" ::: "memory");  // (no load crosses it; asm volatile statements keep their order)
            rank_u32x2 pr[G];
            static_for<0, G>([&](auto ecn) {
                constexpr int e = decltype(ecn)::value;
                pr[e] = *reinterpret_cast<const LdsPairBig*>((uintptr_t)((r[g0 + e] >> 2) & 0x3FFF8u));
            });
            // all G gathers are issued before the first update: left to itself the scheduler interleaves them one by one
            // (ds_read, s_waitcnt lgkmcnt(0), update, ds_read, ...) in some instantiations - <64, 16> among them - and
            // every chunk then waits out a whole LDS round trip
            __builtin_amdgcn_sched_barrier(0);
            static_for<0, G>([&](auto ecn) {
                constexpr int e = decltype(ecn)::value;
                const uint32_t rr = r[g0 + e];
                uint64_t m;
                if constexpr (REV) {
                    const uint32_t sh = pr[e][0] << (rr & 31u);            // my bit on top, the positions behind me below it
                    const uint32_t nob = (uint32_t)__popc(sh) + pr[e][1];  // -(ones before me)
                    m = __ballot((int32_t)sh < 0);
                    r[g0 + e] = __builtin_amdgcn_inverse_ballot_w64(m) ? Zs - nob : rr + nob;
                } else {
                    const uint32_t bit = __builtin_amdgcn_ubfe(pr[e][0], rr, 1u);
                    const uint32_t ob = (uint32_t)__popc(__builtin_amdgcn_ubfe(pr[e][0], 0u, rr)) + pr[e][1];
                    m = __ballot(bit != 0u);
                    r[g0 + e] = __builtin_amdgcn_inverse_ballot_w64(m) ? Zs + ob : rr - ob;
                }
                mine_lo = write_lane(mine_lo, (uint32_t)m, (uint32_t)(g0 + e));
                mine_hi = write_lane(mine_hi, (uint32_t)(m >> 32), (uint32_t)(g0 + e));
            });
#pragma unroll
            for (int e = 0; e < G; ++e) asm volatile("" : "+v"(r[g0 + e]));  // this group's updates end here
        });
        // The single row buffer is rewritten between two barriers; the output stores come after that, so that the
        // wait for the prefetched row does not also wait for them (loads and stores return in order on one counter).
        const size_t orow_w = (size_t)line * A.out_stride_w;
        prof(0);
        __syncthreads();  // everyone is done with the row
        prof(1);
        store_row();
        line = (uint32_t)__builtin_amdgcn_readfirstlane((int)line_n);
        Z = (uint32_t)__builtin_amdgcn_readfirstlane((int)Z_n);
        asm volatile("" : "+s"(line), "+s"(Z)::"memory");
        if (store_lane) {
            rank_u32x2 ov = {mine_lo & vm_lo, mine_hi & vm_hi};
            __builtin_nontemporal_store(ov, reinterpret_cast<rank_u32x2*>(A.out + orow_w) + cg0 + lane);
        }
        if (pad_writer)
            for (uint32_t i = row_words + tid; i < A.out_stride_w; i += T) A.out[orow_w + i] = 0;
        __syncthreads();
        prof(2);
    }
    if (A.ph_start && wah_first + n_wah != D.wah_first + D.n_wah) {
        static_for<0, E>([&](auto ecn) {
            constexpr int e = decltype(ecn)::value;
            park[(size_t)e * T] = r[e];
        });
    }
}

// One workgroup per block, all of a block's haplotypes in its registers (N <= 65536: 64 chunks per wave at
// most): the geometry for batches with at least as many blocks as CUs, where splitting a block over
// workgroups only multiplies the row staging.  The {bits, prefix} row of line j+1 is fetched into two
// registers per thread while line j runs and parked in the other half of a double buffer: one barrier per
// line.  Per 64 haplotypes: one ds_read_b64 gather, 11 vector instructions, no scalar work.
template <int E, bool COMPACT>
__global__ void __launch_bounds__(1024) k_chain_decode_rank_wg(RankArgs A) {
    constexpr uint32_t T = 1024, W = 16;
    constexpr int G = 8;
    static_assert(E % G == 0 && E <= 64, "groups of 8 chunks");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const DecBlock& D = A.blocks[blockIdx.x];
    if (D.error || D.n_wah == 0 || D.off_line_haploid != VAL_UNDEFINED) return;
    const uint32_t N = A.N;
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const uint32_t w = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6));
    const uint32_t cg0 = w * E;
    const uint32_t wah_first = A.ph_start ? A.ph_start[blockIdx.x] : D.wah_first;
    const uint32_t n_wah = A.ph_start ? A.ph_cnt[blockIdx.x] : D.n_wah;
    if (n_wah == 0) return;  // no line of this block in this range: its ranks stay parked
    const uint32_t CWP = A.yp_stride;  // pairs per row, <= 2048
    constexpr uint32_t SLOT = 16384u;  // bytes of one staged row
    uint2* stage = reinterpret_cast<uint2*>(smem);
    const uint32_t tab_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;

    uint32_t r[E];
    uint32_t* park = A.state + ((size_t)blockIdx.x * (uint32_t)E) * T + tid;  // chunk e of my wave: park[e * T]
    // the block's first line is in this range: identity; else the ranks parked by the launch of the range before
    // (a short block has empty ranges: "first" and "last" are the block's own, not the launch's)
    if (!A.ph_start || wah_first == D.wah_first) {
        static_for<0, E>([&](auto ecn) {
            constexpr int e = decltype(ecn)::value;
            r[e] = (cg0 + (uint32_t)e) * 64u + lane;
            if (r[e] >= N) r[e] = 0;  // haplotypes beyond N idle on position 0; their output is masked
        });
    } else {
        static_for<0, E>([&](auto ecn) {
            constexpr int e = decltype(ecn)::value;
            r[e] = park[(size_t)e * T];
        });
    }
    uint32_t vm_lo = 0, vm_hi = 0;  // lane e (< E) stores chunk cg0+e's word: valid bits of that chunk
    {
        const uint64_t base = (uint64_t)(cg0 + lane) * 64u;
        if (lane < (uint32_t)E && base < N) {
            const uint32_t nv = (N - base >= 64u) ? 64u : (uint32_t)(N - base);
            const uint64_t vm = nv == 64u ? ~0ull : ((1ull << nv) - 1ull);
            vm_lo = (uint32_t)vm;
            vm_hi = (uint32_t)(vm >> 32);
        }
    }
    const uint32_t row_chunks = A.out_stride_w / 2u;  // 8-byte words per output row (padding included)
    const bool store_lane = lane < (uint32_t)E && cg0 + lane < row_chunks && (A.out_stride_w & 1u) == 0u;
    const bool odd_tail = (A.out_stride_w & 1u) != 0u;  // rows of an odd number of words: rare, word stores

    // pairs form: R[0], R[1] = pairs tid and 1024 + tid; compact form: R[0] = chunk tid (64 row bits), R[1].x = ones before it
    uint2 R[2];
    auto load_row = [&](uint32_t j) {
        if constexpr (COMPACT) {
            // threads beyond the row read chunk 0 and store nothing: an unconditional load stays in flight across
            // the line (with a default value for them the compiler waits for the load where it is issued)
            const uint2* src = A.yc + (size_t)(wah_first + j) * A.yc_stride;       // uniform: scalar address arithmetic
            const uint16_t* srp = A.ypre + (size_t)(wah_first + j) * A.yc_stride;
            const uint32_t t = tid < A.yc_stride ? tid : 0u;
            R[0] = src[t];
            R[1] = make_uint2((uint32_t)srp[t], 0u);
        } else {
            const uint2* src = A.yp + (size_t)(wah_first + j) * CWP;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const uint32_t idx = (uint32_t)q * T + tid;
                R[q] = src[idx < CWP ? idx : 0u];  // unconditional (see above); store_row keeps to the row
            }
        }
    };
    auto store_row = [&](uint32_t buf) {
        if constexpr (COMPACT) {  // pairs 2 tid, 2 tid + 1 of the staged row (16 bytes per lane: no bank conflicts)
            if (2u * tid < CWP) {
                // the pairs are formed HERE, so they are formed in the reversed form (see k_chain_decode_rank_big<.., REV>):
                // {bits reversed within the word, -(ones up to the word's end)}: one shift in the main loop serves the
                // bit and the count, 10 instead of 11 vector instructions per chunk for five more per thread and line
                const uint32_t e0 = R[1].x + (uint32_t)__popc(R[0].x), e1 = e0 + (uint32_t)__popc(R[0].y);
                *reinterpret_cast<uint4*>(stage + buf * (SLOT / 8u) + 2u * tid) =
                    make_uint4(__brev(R[0].x), 0u - e0, __brev(R[0].y), 0u - e1);
            }
        } else {
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const uint32_t idx = (uint32_t)q * T + tid;
                if (idx < CWP) stage[buf * (SLOT / 8u) + idx] = R[q];
            }
        }
    };
    load_row(0);
    store_row(0);
    uint32_t line = A.wah_lines[wah_first], Z = A.wah_z[wah_first];
    __syncthreads();
    using LdsPair = __attribute__((address_space(3))) rank_u32x2;
    if (tab_lds != 0u) __builtin_trap();  // (the gathers address the staged rows from LDS address 0)
    // One line.  The half of the double buffer it reads is a compile-time constant (the line loop below runs two lines per
    // turn), so the half's base rides in the gather's offset field: no add per gather (25.4 -> 25.2 ms at configs[2]).
    auto line_step = [&](uint32_t j, auto half_c) {
        constexpr uint32_t HALF = decltype(half_c)::value;
        // The prefetch is unconditional (the last line fetches its own row again): with it under a branch the
        // compiler cannot tell at the loop head whether the loads have been waited for and drains vmcnt there,
        // which also waits for the output stores of the line before.
        const uint32_t jn = j + 1u < n_wah ? j + 1u : j;
        load_row(jn);
        const uint32_t line_n = A.wah_lines[wah_first + jn];
        const uint32_t Z_n = A.wah_z[wah_first + jn];
        const uint32_t Zs = (uint32_t)__builtin_amdgcn_readfirstlane((int)Z);  // wave-uniform: keep it scalar
        uint32_t mine_lo = 0, mine_hi = 0;
        static_for<0, E / G>([&](auto gcn) {
            constexpr int g0 = decltype(gcn)::value * G;
            // The groups are straight-line code without a branch between them: left alone the compiler issues
            // the gathers of ALL groups first (2 E registers of pairs, E ballot masks in SGPRs) and spills.
            // An asm statement no load may cross pins each group's gathers behind the previous group's rank
            // updates (asm volatile statements keep their order).
            asm volatile("" ::: "memory");
            rank_u32x2 pr[G];
            static_for<0, G>([&](auto ecn) {
                constexpr int e = decltype(ecn)::value;
                pr[e] = *reinterpret_cast<const LdsPair*>((uintptr_t)(((r[g0 + e] >> 2) & 0x3FF8u) | (HALF * SLOT)));
            });
            // <56>: all eight gathers issued before the first update.  That instantiation came out with ONE in flight -
            // ds_read, s_waitcnt lgkmcnt(0), update, ds_read, ... (tools/isa_scan.py) -: 25.7 -> 23.2 ms at 50 000
            // haplotypes x 2 M sites, 25.5 -> 24.1 at 57 000.  Measured for every E: <48>, serialised as well, lost
            // 0.5 ms with the barrier (45 000 haplotypes: 19.8 -> 20.4) and <64>, which has three in flight without
            // it, 0.7 (27.2 -> 28.0): with fewer chunks per wave the interleaved form hides more than it costs.
            if constexpr (E == 56) __builtin_amdgcn_sched_barrier(0);
            static_for<0, G>([&](auto ecn) {
                constexpr int e = decltype(ecn)::value;
                const uint32_t rr = r[g0 + e];
                uint64_t m;
                if constexpr (COMPACT) {  // reversed pairs (store_row)
                    const uint32_t sh = pr[e][0] << (rr & 31u);            // my bit on top, the positions behind me below it
                    const uint32_t nob = (uint32_t)__popc(sh) + pr[e][1];  // -(ones before me)
                    m = __ballot((int32_t)sh < 0);
                    r[g0 + e] = __builtin_amdgcn_inverse_ballot_w64(m) ? Zs - nob : rr + nob;
                } else {
                    const uint32_t bit = __builtin_amdgcn_ubfe(pr[e][0], rr, 1u);
                    const uint32_t ob = (uint32_t)__popc(__builtin_amdgcn_ubfe(pr[e][0], 0u, rr)) + pr[e][1];
                    m = __ballot(bit != 0u);
                    r[g0 + e] = __builtin_amdgcn_inverse_ballot_w64(m) ? Zs + ob : rr - ob;
                }
                mine_lo = write_lane(mine_lo, (uint32_t)m, (uint32_t)(g0 + e));
                mine_hi = write_lane(mine_hi, (uint32_t)(m >> 32), (uint32_t)(g0 + e));
            });
#pragma unroll
            for (int e = 0; e < G; ++e) asm volatile("" : "+v"(r[g0 + e]));  // this group's updates end here
        });
        // Park the prefetched row BEFORE this line's output stores: loads and stores share vmcnt and return in order,
        // so a wait for the row placed behind the stores would also wait for them to complete, every line.
        uint32_t* orow = A.out + (size_t)line * A.out_stride_w;
        store_row(1u - HALF);
        line = (uint32_t)__builtin_amdgcn_readfirstlane((int)line_n);  // the two small loads are consumed here too
        Z = (uint32_t)__builtin_amdgcn_readfirstlane((int)Z_n);
        asm volatile("" : "+s"(line), "+s"(Z)::"memory");
        if (store_lane) {  // nobody on the device reads the row again before the launch ends: keep it out of the L2's way
            rank_u32x2 ov = {mine_lo & vm_lo, mine_hi & vm_hi};
            __builtin_nontemporal_store(ov, reinterpret_cast<rank_u32x2*>(orow) + cg0 + lane);
        }
        if (odd_tail && lane < (uint32_t)E) {
            const uint32_t wi = 2u * (cg0 + lane);
            if (wi < A.out_stride_w) orow[wi] = mine_lo & vm_lo;
            if (wi + 1u < A.out_stride_w) orow[wi + 1u] = mine_hi & vm_hi;
        }
        // words of the output row beyond the chunks this workgroup holds (rows padded past 16*E chunks)
        for (uint32_t i = 2u * W * E + tid; i < A.out_stride_w; i += T) orow[i] = 0;
        __syncthreads();  // the next row is staged; everyone is done with this one
    };
    uint32_t j = 0;
    for (; j + 1u < n_wah; j += 2u) {
        line_step(j, std::integral_constant<uint32_t, 0u>{});
        line_step(j + 1u, std::integral_constant<uint32_t, 1u>{});
    }
    if (j < n_wah) line_step(j, std::integral_constant<uint32_t, 0u>{});
    if (A.ph_start && wah_first + n_wah != D.wah_first + D.n_wah) {
        static_for<0, E>([&](auto ecn) {
            constexpr int e = decltype(ecn)::value;
            park[(size_t)e * T] = r[e];
        });
    }
}

// ------------------------------------------------------------------------------------------
// Position-major decode chain for N <= 65 536 (round 6): k_chain_decode_pos.
//
// The element-major kernels walk every haplotype through every WAH line: one random LDS gather and ten vector
// instructions per 64 haplotypes, whatever the line looks like.  Decode knows the PERMUTED row y_k - it is indexed by
// POSITION - so a kernel that owns positions gets a chunk's 64 row bits and the ones in front of it as scalars (compact
// rows: DecLines::yp_compact), and on PBWT-ordered rows most chunks are uniform (at 64 976 haplotypes 66 % of the chunks of
// a WAH line are all zero, 11 % all one, 23 % mixed).  The stable partition a_{k+1} = [a_k[i] : y = 0] ++ [a_k[i] : y = 1]
// (gt_block.hpp:124-136; accessor_internals_new.hpp:548-589) moves a uniform chunk as ONE linear run: 64 consecutive
// 16-bit entries to a destination that is a scalar -
//     zeros of chunk c  ->  position 64 c - ones_before(c)          ones of chunk c  ->  Z + ones_before(c)
// - one v_add (lane offset + scalar destination) and one conflict-free ds_write_b16; only a mixed chunk computes per-lane
// destinations (two v_mbcnt + a select).  `a` lives in registers position-major (lane l of chunk c holds a[64 c + l]): a line
// scatters the registers into a 16-bit copy of the array in LDS, a barrier, and every lane reads its positions back
// (linear ds_read_u16 with the chunk in the instruction's offset field: no address arithmetic).  The decoded row is
// x_k[a_k[i]] = y_k[i]: only the ONES are deposited, with LDS atomic ORs into a row bitmap that leaves with the line.
// Two barriers a line (the scatter must land before the read-back, the read-back before the next line's scatter).
// ------------------------------------------------------------------------------------------
extern "C" __device__ rank_v4u __xsi_s_buffer_load_v4_r(rank_v4u rsrc, uint32_t byte_offset, uint32_t cache_policy)
    __asm("llvm.amdgcn.s.buffer.load.v4i32");
extern "C" __device__ rank_v16u __xsi_s_buffer_load_v16_r(rank_v4u rsrc, uint32_t byte_offset, uint32_t cache_policy)
    __asm("llvm.amdgcn.s.buffer.load.v16i32");

// Eight chunks of a line of k_chain_decode_pos, written out.  The compiler's structurizer turns the three-way uniform branch of
// a chunk into flag registers and second branches (8 scalar instructions for a chunk of zeros; the scalar unit issues one
// instruction per SIMD every fourth clock, like the vector unit, so at 64 chunks x 4 waves they set the pace).  Here a chunk of
// zeros is s_cmp, a branch not taken, v_add, ds_write_b16, s_addk, and falls into the next chunk's compare; the other two
// cases stand behind the eight fast paths and jump back.
//   y[k]    the chunk's 64 row bits            z2 / o2   byte addresses, in the scattered array, of where my wave's next
//   lane2   2 x lane                                     zero / next one goes: moved on by what each chunk held
//   a[k]    the lane's entry of the prefix array (a haplotype): stored at its new position; deposited into the row
//           bitmap at LDS address 0 when its bit is set
#define XSI_POS_FAST(K)                                  \
    "s_cmp_lg_u64 %[y" #K "], 0\n\t"                     \
    "s_cbranch_scc1 1" #K "f\n\t"                        \
    "v_add_u32_e32 %[t0], %[z2], %[l2]\n\t"              \
    "ds_write_b16 %[t0], %[a" #K "]\n\t"                 \
    "s_addk_i32 %[z2], 0x80\n"                           \
    "2" #K ":\n\t"
#define XSI_POS_SLOW(K)                                  \
    "1" #K ":\n\t"                                       \
    "s_bcnt1_i32_b64 %[sp], %[y" #K "]\n\t"              \
    "s_lshl_b32 %[sp], %[sp], 1\n\t"                     \
    "s_cmp_eq_u64 %[y" #K "], -1\n\t"                    \
    "s_cbranch_scc1 3" #K "f\n\t"                        \
    "s_mov_b64 vcc, %[y" #K "]\n\t"                      \
    "v_mbcnt_lo_u32_b32 %[t0], vcc_lo, 0\n\t"           \
    "v_mbcnt_hi_u32_b32 %[t0], vcc_hi, %[t0]\n\t"       \
    "v_lshlrev_b32_e32 %[t0], 1, %[t0]\n\t"             \
    "v_add_u32_e32 %[t1], %[z2], %[l2]\n\t"             \
    "v_sub_u32_e32 %[t1], %[t1], %[t0]\n\t"             \
    "v_add_u32_e32 %[t0], %[o2], %[t0]\n\t"             \
    "v_cndmask_b32_e32 %[t0], %[t1], %[t0], vcc\n\t"    \
    "ds_write_b16 %[t0], %[a" #K "]\n\t"                 \
    "s_branch 4" #K "f\n"                                \
    "3" #K ":\n\t"                                       \
    "v_add_u32_e32 %[t0], %[o2], %[l2]\n\t"             \
    "ds_write_b16 %[t0], %[a" #K "]\n"                    \
    "4" #K ":\n\t"                                       \
    "s_and_saveexec_b64 %[sx], %[y" #K "]\n\t"           \
    "v_lshrrev_b32_e32 %[t0], 3, %[a" #K "]\n\t"         \
    "v_and_b32_e32 %[t0], 0x1ffc, %[t0]\n\t"            \
    "v_lshlrev_b32_e64 %[t1], %[a" #K "], 1\n\t"         \
    "ds_or_b32 %[t0], %[t1]\n\t"                        \
    "s_mov_b64 exec, %[sx]\n\t"                         \
    "s_add_i32 %[o2], %[o2], %[sp]\n\t"                 \
    "s_sub_i32 %[z2], %[z2], %[sp]\n\t"                 \
    "s_addk_i32 %[z2], 0x80\n\t"                        \
    "s_branch 2" #K "b\n"
__device__ __forceinline__ void pos_group8(const uint64_t (&y)[8], uint32_t& z2, uint32_t& o2, uint32_t lane2, uint32_t a0,
                                           uint32_t a1, uint32_t a2, uint32_t a3, uint32_t a4, uint32_t a5, uint32_t a6, uint32_t a7) {
    uint32_t t0, t1, sp;
    uint64_t sx;
    const uint64_t y0 = y[0], y1 = y[1], y2 = y[2], y3 = y[3], y4 = y[4], y5 = y[5], y6 = y[6], y7 = y[7];
    asm volatile(
        XSI_POS_FAST(0) XSI_POS_FAST(1) XSI_POS_FAST(2) XSI_POS_FAST(3) XSI_POS_FAST(4) XSI_POS_FAST(5) XSI_POS_FAST(6) XSI_POS_FAST(7)
        "s_branch 99f\n"
        XSI_POS_SLOW(0) XSI_POS_SLOW(1) XSI_POS_SLOW(2) XSI_POS_SLOW(3) XSI_POS_SLOW(4) XSI_POS_SLOW(5) XSI_POS_SLOW(6) XSI_POS_SLOW(7)
        "99:"
        : [z2] "+s"(z2), [o2] "+s"(o2), [t0] "=&v"(t0), [t1] "=&v"(t1), [sp] "=&s"(sp), [sx] "=&s"(sx)
        : [y0] "s"(y0), [y1] "s"(y1), [y2] "s"(y2), [y3] "s"(y3), [y4] "s"(y4), [y5] "s"(y5), [y6] "s"(y6), [y7] "s"(y7),
          [l2] "v"(lane2), [a0] "v"(a0), [a1] "v"(a1), [a2] "v"(a2), [a3] "v"(a3), [a4] "v"(a4), [a5] "v"(a5), [a6] "v"(a6), [a7] "v"(a7)
        : "vcc", "scc", "memory");
}
#undef XSI_POS_FAST
#undef XSI_POS_SLOW

template <int E>
__global__ void __launch_bounds__(1024) k_chain_decode_pos(RankArgs A) {
    constexpr uint32_t T = 1024, W = 16;
    constexpr int G = 8;
    static_assert(E % G == 0 && E <= 64, "groups of 8 chunks");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const DecBlock& D = A.blocks[blockIdx.x];
    if (D.error || D.n_wah == 0 || D.off_line_haploid != VAL_UNDEFINED) return;
    const uint32_t N = A.N;
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const uint32_t w = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6));
    const uint32_t cg0 = w * E;  // my wave's first chunk
    const uint32_t wah_first = A.ph_start ? A.ph_start[blockIdx.x] : D.wah_first;
    const uint32_t n_wah = A.ph_start ? A.ph_cnt[blockIdx.x] : D.n_wah;
    if (n_wah == 0) return;  // no line of this block in this range: the array stays parked
    // LDS: [0, ROW_BYTES) the row bitmap of the line (so that a haplotype's word address is two instructions);
    //      [ROW_BYTES, +2 * 16 E 64) the scattered array (16-bit entries by position)
    constexpr uint32_t ROW_WORDS = W * E * 2u;  // bits of 16 E chunks
    constexpr uint32_t ROW_BYTES = ROW_WORDS * 4u;
    uint32_t* orow = reinterpret_cast<uint32_t*>(smem);
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
    if (lds0 != 0u) __builtin_trap();  // (addresses below are formed from positions alone)
    using LdsU16 = __attribute__((address_space(3))) uint16_t;

    uint32_t a[E];
    uint32_t* park = A.state + ((size_t)blockIdx.x * (uint32_t)E) * T + tid;  // chunk e of my wave: park[e * T]
    if (!A.ph_start || wah_first == D.wah_first) {
        static_for<0, E>([&](auto ecn) {
            constexpr int e = decltype(ecn)::value;
            a[e] = (cg0 + (uint32_t)e) * 64u + lane;  // identity (gt_block.hpp:179); positions beyond N never move
        });
    } else {
        static_for<0, E>([&](auto ecn) {
            constexpr int e = decltype(ecn)::value;
            a[e] = park[(size_t)e * T];
        });
    }
#pragma unroll
    for (int e = 0; e < E; ++e) asm volatile("" : "+v"(a[e]));  // (the parked array has arrived: no pending load rides into the loop)
    for (uint32_t i = tid; i < ROW_WORDS; i += T) orow[i] = 0;
    const uint32_t nch = (N + 63u) / 64u;
    const uint32_t tail_bits = N & 63u;                          // valid positions of the last chunk (0: it is full)
    const uint32_t lane2 = lane * 2u;
    uint32_t vm_lo = 0, vm_hi = 0;  // thread t stores the row's words 2 t, 2 t + 1 = chunk t: its bits that exist
    if ((uint64_t)tid * 64u < N) {
        const uint64_t left = N - (uint64_t)tid * 64u;
        const uint64_t vmk = left >= 64u ? ~0ull : (1ull << left) - 1ull;
        vm_lo = (uint32_t)vmk;
        vm_hi = (uint32_t)(vmk >> 32);
    }
    const uint32_t rb_base = ROW_BYTES + (cg0 * 64u + lane) * 2u;  // my lane's entry of chunk 0 of my wave
    const uint32_t row_bytes = A.yc_stride * 8u;
    // my wave's chunks that exist (whole groups beyond the row are skipped by one scalar branch)
    const uint32_t my_chunks = cg0 >= nch ? 0u : (nch - cg0 < (uint32_t)E ? nch - cg0 : (uint32_t)E);
    // the slot, among my wave's chunks, of the first chunk that holds positions at or beyond N: the row's last chunk when it
    // is partial, else the chunk behind it (0xFFFFFF00: none of mine; 0 also when every chunk of mine lies beyond)
    const uint32_t first_pad_chunk = tail_bits ? nch - 1u : nch;
    const uint32_t pad_slot0 = first_pad_chunk >= cg0 + (uint32_t)E ? 0xFFFFFF00u : (first_pad_chunk > cg0 ? first_pad_chunk - cg0 : 0u);
    const uint64_t tail_pad = tail_bits ? ~((1ull << tail_bits) - 1ull) : ~0ull;
    auto rsrc_of = [&](const void* base, uint32_t bytes) -> rank_v4u {
        const uint64_t b = reinterpret_cast<uint64_t>(base);
        rank_v4u d;
        d[0] = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)b);
        d[1] = (uint32_t)__builtin_amdgcn_readfirstlane((int)((uint32_t)(b >> 32) & 0xFFFFu));
        d[2] = bytes;          // beyond the row: zeros
        d[3] = 0x00020000u;
        return d;
    };
    __syncthreads();
    // Nothing a line needs may be asked of memory when the line starts (all sixteen waves would wait out the round trip, every
    // line): its number, its zeros, my wave's first prefix and my wave's first eight chunks are requested a line ahead.
    const uint32_t pre_idx = cg0 < nch ? cg0 : 0u;
    uint32_t line_v = A.wah_lines[wah_first], z_v = A.wah_z[wah_first], ob_v = (uint32_t)A.ypre[(size_t)wah_first * A.yc_stride + pre_idx];
    auto request = [&](const rank_v4u& rs, uint32_t chunk, rank_v16u& y) {
        asm volatile("s_buffer_load_dwordx16 %0, %1, %2" : "=&s"(y) : "s"(rs), "s"(chunk * 8u));
    };
    rank_v16u yv;
    request(rsrc_of(A.yc + (size_t)wah_first * A.yc_stride, row_bytes), cg0, yv);
    for (uint32_t j = 0; j < n_wah; ++j) {
        const uint32_t rank = wah_first + j;
        const uint32_t line = (uint32_t)__builtin_amdgcn_readfirstlane((int)line_v);
        const uint32_t Z2 = ROW_BYTES + 2u * (uint32_t)__builtin_amdgcn_readfirstlane((int)z_v);
        const uint32_t ob0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)ob_v);
        const rank_v4u rs_y = rsrc_of(A.yc + (size_t)rank * A.yc_stride, row_bytes);
        const uint32_t rank_n = wah_first + (j + 1u < n_wah ? j + 1u : j);  // (the last line asks for itself again: unconditional loads)
        line_v = A.wah_lines[rank_n];
        z_v = A.wah_z[rank_n];
        ob_v = (uint32_t)A.ypre[(size_t)rank_n * A.yc_stride + pre_idx];
        // The key bits reach the waves through scalar loads of rows nothing has touched before: left alone every group of
        // every line waits out a round trip to memory.  One coalesced vector load per thread pulls the row of line j + 2
        // into L2 two lines ahead (as k_chain_rank_enc does).
        uint2 pf;
        {
            const uint32_t jp = j + 2u < n_wah ? j + 2u : j;
            const uint2* rowp = A.yc + (size_t)(wah_first + jp) * A.yc_stride;
            pf = rowp[tid < A.yc_stride ? tid : 0u];
        }
        // The key bits of group g + 1 travel while group g is scattered (the first group: since the end of the line before).
        // Written out as asm: left to the compiler every group's loads are hoisted to the top of the line (8 x 16 SGPRs:
        // spilled lane by lane), and scalar loads return out of order, so the wait for group g stands in front of the
        // request for g + 1.
        // formed inside the line (hoisted out of it, the per-chunk scalars of all 64 chunks live in SGPRs the kernel does
        // not have: they were spilled lane by lane and read back with a v_readlane each)
        uint32_t cg0_l = cg0;
        asm volatile("" : "+s"(cg0_l));
        // Byte addresses, in the scattered array, of where the next zero and the next one of my wave's positions go: the
        // prefix of my first chunk starts them, every chunk moves them on by what it held (a scalar instruction a chunk,
        // where the per-chunk prefixes cost four: the scalar unit issues one instruction per SIMD every fourth clock, like
        // the vector unit, and a chunk of zeros is otherwise three scalar and one vector instruction)
        uint32_t z2 = ROW_BYTES + (cg0_l * 64u - (cg0_l < nch ? ob0 : 0u)) * 2u, o2 = Z2 + ob0 * 2u;
        static_for<0, E / G>([&](auto gcn) {
            constexpr int g0 = decltype(gcn)::value * G;
            asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(yv));
            rank_v16u yn = yv;
            if constexpr (g0 + G < E)
                request(rs_y, cg0_l + (uint32_t)(g0 + G), yn);
            else
                request(rsrc_of(A.yc + (size_t)rank_n * A.yc_stride, row_bytes), cg0_l, yn);  // the next line's first group, across the barriers
            if ((uint32_t)g0 < my_chunks) {
                uint64_t yq[G];
                static_for<0, G>([&](auto ecn) {
                    constexpr int e = decltype(ecn)::value;
                    yq[e] = ((uint64_t)yv[2 * e + 1] << 32) | yv[2 * e];
                });
                if (__builtin_expect(pad_slot0 < (uint32_t)(g0 + G), 0)) {
                    // Positions at or beyond N - the end of the row's last chunk when N is not a multiple of 64, and the
                    // chunks behind it that this group still covers - count as ONES: the partition then keeps them where
                    // they are, behind the N real entries ([zeros][ones][pads]), with no case of their own; what they
                    // deposit into the row lies beyond bit N and is masked when the row leaves.  (One group of one wave; in
                    // asm so that the selects stay in SGPRs: the results of asm statements count as divergent.)
                    static_for<0, G>([&](auto ecn) {
                        constexpr int e = decltype(ecn)::value;
                        uint64_t t, y = yq[e];
                        const uint32_t ps = (uint32_t)__builtin_amdgcn_readfirstlane((int)pad_slot0);  // (an SGPR operand below)
                        const uint64_t tp = tail_pad;
                        asm volatile("s_cmp_eq_u32 %[ps], %[c]\n\t"
                                     "s_cselect_b64 %[t], %[tp], 0\n\t"
                                     "s_cmp_lt_u32 %[ps], %[c]\n\t"
                                     "s_cselect_b64 %[t], -1, %[t]\n\t"
                                     "s_or_b64 %[y], %[y], %[t]"
                                     : [y] "+s"(y), [t] "=&s"(t)
                                     : [ps] "s"(ps), [c] "i"(g0 + e), [tp] "s"(tp)
                                     : "scc");
                        yq[e] = y;
                    });
                }
                pos_group8(yq, z2, o2, lane2, a[g0], a[g0 + 1], a[g0 + 2], a[g0 + 3], a[g0 + 4], a[g0 + 5], a[g0 + 6], a[g0 + 7]);
            }
            yv = yn;
        });
        asm volatile("" ::"v"(pf.x), "v"(pf.y));  // the prefetch has landed (nothing reads the registers)
        lds_barrier();  // the scattered array and the row are complete
        static_for<0, E>([&](auto ecn) {
            constexpr int e = decltype(ecn)::value;
            a[e] = (uint32_t)*reinterpret_cast<const LdsU16*>((uintptr_t)(rb_base + (uint32_t)e * 128u));
        });
        {
            uint32_t* orow_g = A.out + (size_t)line * A.out_stride_w;
            uint2 v = make_uint2(0u, 0u);
            if (2u * tid < ROW_WORDS) {
                v = *reinterpret_cast<const uint2*>(orow + 2u * tid);
                *reinterpret_cast<uint2*>(orow + 2u * tid) = make_uint2(0u, 0u);  // ready for the next line (behind the barrier below)
                v.x &= vm_lo;  // (the pads' deposits)
                v.y &= vm_hi;
            }
            if (2u * tid + 1u < A.out_stride_w) {
                rank_u32x2 ov = {v.x, v.y};
                __builtin_nontemporal_store(ov, reinterpret_cast<rank_u32x2*>(orow_g) + tid);
            } else if (2u * tid < A.out_stride_w) {
                orow_g[2u * tid] = v.x;
            }
            for (uint32_t i = 2u * T + tid; i < A.out_stride_w; i += T) orow_g[i] = 0;  // rows padded past 2048 words
        }
        lds_barrier();  // everyone has read its positions back: the next line may scatter
    }
    if (A.ph_start && wah_first + n_wah != D.wah_first + D.n_wah) {
        static_for<0, E>([&](auto ecn) {
            constexpr int e = decltype(ecn)::value;
            park[(size_t)e * T] = a[e];
        });
    }
}

struct RankGeom {
    int T, E;
    uint32_t splits, batch, lds_bytes, log2_cwp;
    bool stage;
};

static RankGeom rank_geometry(uint32_t N, uint32_t yp_stride, uint32_t n_blocks) {
    RankGeom g{};
    const uint32_t nch = (N + 63u) / 64u;
    g.stage = N <= 65536u;
    // Every workgroup stages the whole rank-select row of each line, so splitting a block's
    // haplotypes over S workgroups multiplies that L2 traffic by S.  Prefer the largest workgroup
    // that still gives about one workgroup per CU, and at most 8 chunks per wave.
    static const int env_e = [] {
        const char* e = tuning_env("XSI_DEC_E");
        const int v = e ? atoi(e) : 0;
        return (v >= 1 && v <= 8) ? v : 0;
    }();
    static const int env_t = [] {
        const char* e = tuning_env("XSI_DEC_T");
        const int v = e ? atoi(e) : 0;
        return (v == 256 || v == 512 || v == 1024) ? v : 0;
    }();
    // about one workgroup per CU (measured best at the bench size: T=512, E=5, 2 splits: 2.3 ms
    // against 2.9 ms for 5 splits of T=256 and 3.4 ms for a single 1024-thread workgroup per block)
    int T, E;
    {
        const uint32_t s_target = n_blocks >= 256u ? 1u : (256u + n_blocks / 2u) / (n_blocks ? n_blocks : 1u);
        const uint32_t per_wg = (nch + s_target - 1u) / s_target;  // chunks one workgroup should cover
        // (16 waves of 3 chunks beat 8 waves of 5 at 5008 haplotypes, 2.14 against 2.33 ms: half the staging per thread)
        T = per_wg <= 8u ? 256 : (per_wg <= 32u ? 512 : 1024);
        const uint32_t waves = (uint32_t)T / 64u;
        uint32_t e = (per_wg + waves - 1u) / waves;
        if (e < 1u) e = 1u;
        if (e > 8u) e = 8u;
        E = (int)e;
    }
    if (env_t) T = env_t;
    if (env_e) E = env_e;
    g.T = T;
    g.E = E;
    const uint32_t per_wg = (uint32_t)(T / 64) * (uint32_t)E;
    g.splits = (nch + per_wg - 1u) / per_wg;
    g.log2_cwp = next_pow2_log2(yp_stride);
    if (g.stage) {
        uint32_t B = (uint32_t)(RANK_RP * g.T) >> g.log2_cwp;
        // A batch costs a barrier, a flush and a wait for its rows whatever its size (configs[1], 1024 threads: 8 lines
        // per batch 2.58 ms, 16: 2.15, 24: 2.06, 32: 2.05), so batches are as long as the registers
        // (RANK_RP pairs per thread) and LDS allow: 64 KB when several workgroups share a CU, 128 KB when one has it.
        static const uint32_t bcap = [] { const char* e = tuning_env("XSI_DEC_BCAP"); const int v = e ? atoi(e) : 32; return (uint32_t)(v < 1 ? 1 : (v > 64 ? 64 : v)); }();
        if (B > bcap) B = bcap;
        if (B < 1u) B = 1u;
        auto need = [&](uint32_t b) { return 2u * b * yp_stride * 8u + 3u * b * 8u + 2u * b * per_wg * 8u + 64u; };
        const uint32_t lds_limit = ((uint64_t)n_blocks * g.splits <= 256u ? 128u : 64u) * 1024u;
        while (B > 1u && need(B) > lds_limit) B -= (B > 16u ? 4u : B / 2u);
        g.batch = B;
        g.lds_bytes = need(B);
        if ((uint32_t)(RANK_RP * g.T) < yp_stride) g.stage = false;  // one row does not fit a register batch
    }
    if (!g.stage) {
        g.batch = 1;
        g.lds_bytes = 0;
    }
    return g;
}

template <bool STAGE>
static hipError_t launch_rank(hipStream_t s, const RankGeom& g, uint32_t n_blocks, RankArgs A) {
#define XSI_RANK_CASE(TT, EE)                                                                                 \
    if (g.T == TT && g.E == EE) {                                                                             \
        if (g.lds_bytes) {                                                                                    \
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_chain_decode_rank<TT, EE, STAGE>), \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)g.lds_bytes); \
            if (e != hipSuccess) return e;                                                                    \
        }                                                                                                     \
        k_chain_decode_rank<TT, EE, STAGE><<<dim3(n_blocks, g.splits), dim3(TT), g.lds_bytes, s>>>(A);        \
        return hipGetLastError();                                                                             \
    }
#define XSI_RANK_CASES(TT) \
    XSI_RANK_CASE(TT, 1) XSI_RANK_CASE(TT, 2) XSI_RANK_CASE(TT, 3) XSI_RANK_CASE(TT, 4) \
    XSI_RANK_CASE(TT, 5) XSI_RANK_CASE(TT, 6) XSI_RANK_CASE(TT, 7) XSI_RANK_CASE(TT, 8)
    XSI_RANK_CASES(256)
    XSI_RANK_CASES(512)
    XSI_RANK_CASES(1024)
#undef XSI_RANK_CASES
#undef XSI_RANK_CASE
    return hipErrorInvalidValue;
}


// CUs of the current device (one chain workgroup per CU); 256 when the attribute cannot be read
uint32_t rank_decode_cus() {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus > 0)
        return (uint32_t)cus;
    return 256u;
}

static uint32_t rank_big_rp(uint32_t yp_stride) { return yp_stride <= 8u * 1024u ? 8u : (yp_stride <= 16u * 1024u ? 16u : 20u); }

// Chunks per wave of the long-row kernel for a launch with `active` blocks on a chip of `cus` CUs - ONE selection, used by
// the launcher and by rank_decode_big_wgs_per_block (the batch quantum of xsi_hip_decode_packed), so that the quantum and
// the geometry actually launched agree.  Every workgroup of a block stages the whole rank-select row of each
// line: the fewer workgroups per block (the more chunks per wave) the less of that, as long as the launch still fills the
// chip.  Cost = rounds of the chip the launch takes x what a round costs: a workgroup with half the chunks per wave does a
// little more than half the work (it stages the same row: 0.59 of the time at 500 000 haplotypes,
// profiles/r05_config3_kernel_stats.csv) - 25 blocks: one round of <64> at 25 / 32 of the chip (1.0) beats two of <32>
// (1.18); 12 blocks: <32> in one round.
static uint32_t rank_big_pick_e(uint32_t N, uint32_t yp_stride, uint32_t active, uint32_t cus) {
    const uint32_t nch = (N + 63u) / 64u;
    const uint32_t RP = rank_big_rp(yp_stride);
    const uint32_t e_max = RP == 20u ? 16u : 64u;  // <64, 16> fits since the row travels as 16-byte pieces (it spilled 60 VGPRs)
    if (const char* ev = tuning_env("XSI_DEC_BIG_E")) {
        const uint32_t v = (uint32_t)atoi(ev);
        if ((v == 8u || v == 16u || v == 32u || v == 64u) && v <= e_max) return v;
    }
    if (!cus) cus = 256u;
    if (!active) active = cus;  // "enough blocks to fill the chip"
    uint32_t E = 8;
    double best_cost = 1e30;
    for (uint32_t e : {64u, 32u, 16u, 8u}) {
        if (e > e_max) continue;
        const uint64_t splits = (nch + 16u * e - 1u) / (16u * e);
        const double rounds = (double)(((uint64_t)active * splits + cus - 1u) / cus);
        const double cost = rounds * (e == 64u ? 1.0 : e == 32u ? 0.59 : e == 16u ? 0.36 : 0.23);
        if (cost < best_cost - 1e-9) {
            best_cost = cost;
            E = e;
        }
    }
    return E;
}

// `active`: blocks that have lines in this launch (a batch's launches all pass the same figure - the parked ranks' layout
// depends on the geometry chosen from it)
static hipError_t launch_rank_big(hipStream_t s, uint32_t n_blocks, RankArgs A, uint32_t active = 0) {
    if (!active || active > n_blocks) active = n_blocks;
    const uint32_t nch = (A.N + 63u) / 64u;
    const uint32_t lds = ((A.yp_stride / 2u + 1023u) / 1024u) * 16384u;  // whole 1024-unit pieces (see the kernel's store_row)
    auto splits_of = [&](uint32_t e) { return (nch + 16u * e - 1u) / (16u * e); };
    const uint32_t RP = rank_big_rp(A.yp_stride);
    const uint32_t E = rank_big_pick_e(A.N, A.yp_stride, active, rank_decode_cus());
#define XSI_BIG_CASE(EE, RR)                                                                                 \
    if (E == EE && RP == RR && A.yp_rev) {                                                                   \
        auto kern = &k_chain_decode_rank_big<EE, RR, false, true>;                                           \
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        if (e != hipSuccess) return e;                                                                       \
        A.big_splits = splits_of(EE);                                                                        \
        A.big_n_blocks = n_blocks;                                                                           \
        kern<<<dim3(splits_of(EE) * ((n_blocks + 7u) & ~7u)), dim3(1024), lds, s>>>(A);                      \
        return hipGetLastError();                                                                            \
    }                                                                                                        \
    if (E == EE && RP == RR) {                                                                               \
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_chain_decode_rank_big<EE, RR>),  \
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);            \
        if (e != hipSuccess) return e;                                                                       \
        A.big_splits = splits_of(EE);                                                                        \
        A.big_n_blocks = n_blocks;                                                                           \
        A.big_prof = tuning_env("XSI_BIG_PROF") ? 1u : 0u;                                                       \
        if (A.big_prof) {                                                                                    \
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_chain_decode_rank_big<EE, RR, true>),   \
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                   \
            if (e != hipSuccess) return e;                                                                   \
            k_chain_decode_rank_big<EE, RR, true><<<dim3(splits_of(EE) * ((n_blocks + 7u) & ~7u)), dim3(1024), lds, s>>>(A); \
        } else                                                                                               \
            k_chain_decode_rank_big<EE, RR><<<dim3(splits_of(EE) * ((n_blocks + 7u) & ~7u)), dim3(1024), lds, s>>>(A); \
        if (A.big_prof) {                                                                                    \
            unsigned long long pr[4] = {0, 0, 0, 0};                                                         \
            (void)hipStreamSynchronize(s);                                                                   \
            (void)hipMemcpyFromSymbol(pr, HIP_SYMBOL(g_big_prof), sizeof(pr));                               \
            fprintf(stderr, "[xsi big prof] E=%d cumulative: main %.2f ms, barrier %.2f ms, row to LDS + stores + barrier %.2f ms\n", EE, \
                    pr[0] * 1e-5, pr[1] * 1e-5, pr[2] * 1e-5);                                                \
        }                                                                                                    \
        return hipGetLastError();                                                                            \
    }
    XSI_BIG_CASE(8, 8)
    XSI_BIG_CASE(16, 8)
    XSI_BIG_CASE(32, 8)
    XSI_BIG_CASE(64, 8)
    XSI_BIG_CASE(8, 16)
    XSI_BIG_CASE(16, 16)
    XSI_BIG_CASE(32, 16)
    XSI_BIG_CASE(64, 16)
    XSI_BIG_CASE(8, 20)
    XSI_BIG_CASE(16, 20)
#undef XSI_BIG_CASE
    return hipErrorInvalidValue;
}

// workgroups per block of the long-row kernel for a launch with `active` blocks (0: enough blocks to fill the chip): what a
// caller that cuts a job into batches of blocks rounds the batches to, so that the last round of a launch is a full one
uint32_t rank_decode_big_wgs_per_block(uint32_t N, uint32_t yp_stride, uint32_t active) {
    if (N <= 65536u) return 1u;
    const uint32_t nch = (N + 63u) / 64u;
    const uint32_t e = rank_big_pick_e(N, yp_stride, active, rank_decode_cus());
    return (nch + 16u * e - 1u) / (16u * e);
}

// one workgroup per block: batches with about as many blocks as CUs, rows that fit a 16 KiB LDS slot
static bool use_rank_wg(uint32_t N, uint32_t yp_stride, uint32_t n_blocks) {
    const char* ev = tuning_env("XSI_RANK_WG_MIN_BLOCKS");  // read per call (tests switch kernels in one process)
    const uint32_t min_blocks = ev ? (uint32_t)atoi(ev) : 192u;
    return yp_stride <= 2048u && N >= 16384u && n_blocks >= min_blocks;
}

const char* rank_decode_kernel_name(uint32_t N, uint32_t yp_stride, uint32_t n_blocks) {
    if (use_rank_wg(N, yp_stride, n_blocks))
        return (N <= 65536u && tuning_env("XSI_POS_DECODE") && !tuning_env("XSI_NO_COMPACT_YP")) ? "k_chain_decode_pos" : "k_chain_decode_rank_wg";
    return N >= 49152u ? "k_chain_decode_rank_big" : "k_chain_decode_rank";
}

static hipError_t launch_pos(hipStream_t s, uint32_t n_blocks, const RankArgs& R, uint32_t e) {
    const uint32_t lds = 16u * e * 64u * 2u + 16u * e * 2u * 4u;
#define XSI_POS_CASE(EE)                                                                                   \
    if (e == EE) {                                                                                          \
        auto kern = &k_chain_decode_pos<EE>;                                                                \
        hipError_t err = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),                           \
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);         \
        if (err != hipSuccess) return err;                                                                  \
        kern<<<dim3(n_blocks), dim3(1024), lds, s>>>(R);                                                    \
        return hipGetLastError();                                                                           \
    }
    XSI_POS_CASE(16)
    XSI_POS_CASE(24)
    XSI_POS_CASE(32)
    XSI_POS_CASE(40)
    XSI_POS_CASE(48)
    XSI_POS_CASE(56)
    XSI_POS_CASE(64)
#undef XSI_POS_CASE
    return hipErrorInvalidValue;
}

// position-major chain (k_chain_decode_pos): compact rows (8-byte chunks: every row base is dword-aligned for the scalar loads)
static bool use_pos_decode(const RankArgs& R) {
    return R.yc && R.ypre && R.N <= 65536u && tuning_env("XSI_POS_DECODE") != nullptr;
}

static hipError_t launch_rank_wg(hipStream_t s, uint32_t n_blocks, const RankArgs& R) {
    const uint32_t nch = (R.N + 63u) / 64u;
    const uint32_t e = ((nch + 15u) / 16u + 7u) / 8u * 8u;  // chunks per wave, multiple of 8
    if (use_pos_decode(R)) return launch_pos(s, n_blocks, R, e);
    const uint32_t lds = 2u * 16384u;
#define XSI_WG_CASE(EE)                                                                                    \
    if (e == EE) {                                                                                          \
        auto kern = R.yc ? &k_chain_decode_rank_wg<EE, true> : &k_chain_decode_rank_wg<EE, false>;          \
        hipError_t err = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),                           \
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);         \
        if (err != hipSuccess) return err;                                                                  \
        kern<<<dim3(n_blocks), dim3(1024), lds, s>>>(R);                                                    \
        return hipGetLastError();                                                                           \
    }
    XSI_WG_CASE(16)
    XSI_WG_CASE(24)
    XSI_WG_CASE(32)
    XSI_WG_CASE(40)
    XSI_WG_CASE(48)
    XSI_WG_CASE(56)
    XSI_WG_CASE(64)
#undef XSI_WG_CASE
    return hipErrorInvalidValue;
}

// which family launch_rank_decode picks: 1 one workgroup per block, 2 long rows (haplotype splits, row staged whole),
// 0 the batch-staged kernels.  The first two can run a block's lines in ranges (ranks parked between launches).
static int rank_decode_family(uint32_t N, uint32_t yp_stride, uint32_t n_blocks) {
    if (use_rank_wg(N, yp_stride, n_blocks)) return 1;
    static const uint32_t big_min = [] {
        const char* e = tuning_env("XSI_BIG_RANK_MIN_N");
        return e ? (uint32_t)atoi(e) : 49152u;  // measured: 11.3 ms against 14.2 ms at 64 976 hap x 64 blocks, slower at 40 000
    }();
    const bool stage = N <= 65536u;
    if ((!stage || N >= big_min) && yp_stride <= 1024u * 20u && yp_stride * 8u <= 160u * 1024u && !tuning_env("XSI_NO_BIG_RANK")) return 2;
    return 0;
}

// Every element-major decode kernel can park its ranks, but the small-N kernels do not gain: at 5008 haplotypes x 123
// blocks the twelve ranges cost the chain 0.6 ms for 0.6 ms of expansion hidden (8.76 against 8.67 ms per step).
bool rank_decode_phased_ok(uint32_t N, uint32_t yp_stride, uint32_t n_blocks) {
    return rank_decode_family(N, yp_stride, n_blocks) != 0 || tuning_env("XSI_DEC_PHASES_SMALL") != nullptr;
}

bool rank_decode_takes_reversed(uint32_t N, uint32_t yp_stride, uint32_t n_blocks) {
    return rank_decode_family(N, yp_stride, n_blocks) == 2 && !tuning_env("XSI_NO_REVERSED_YP");
}

bool rank_decode_takes_compact(uint32_t N, uint32_t yp_stride, uint32_t n_blocks) {
    return N <= 65536u && rank_decode_family(N, yp_stride, n_blocks) == 1 && !tuning_env("XSI_NO_COMPACT_YP");
}

static void rank_args_rows(RankArgs& R, const DecLines& L) {
    R.yp = L.yp;
    R.yp_stride = L.yp_stride;
    R.yc = nullptr;
    R.ypre = nullptr;
    R.yc_stride = 0;
    R.yp_rev = L.yp_rev;
    if (L.yp_compact) {
        R.yc = reinterpret_cast<const uint2*>(L.yp);
        R.ypre = reinterpret_cast<const uint16_t*>(reinterpret_cast<const uint8_t*>(L.yp) + 8ull * L.y_stride64 * L.yp_rows);
        R.yc_stride = L.y_stride64;
    }
}

uint64_t rank_decode_state_words(uint32_t N, uint32_t n_blocks) {  // ranks of every workgroup of a block: N rounded up to
    return (uint64_t)n_blocks * ((((uint64_t)N + 65535u) / 65536u) * 65536u + 16384u);  // its workgroups' capacity, with room
}

hipError_t launch_rank_decode_phase(hipStream_t s, const DecBlock* blocks, uint32_t n_blocks, const DecLines& L,
                                    uint32_t* out_rows, uint32_t out_stride_w, const uint32_t* ph_start,
                                    const uint32_t* ph_cnt, uint32_t* state, uint32_t active_blocks) {
    if (!n_blocks) return hipSuccess;
    RankArgs R{};
    R.blocks = blocks;
    R.wah_lines = L.wah_lines;
    rank_args_rows(R, L);
    R.wah_z = L.wah_z;
    R.out = out_rows;
    R.out_stride_w = out_stride_w;
    R.N = L.N;
    R.ph_start = ph_start;
    R.ph_cnt = ph_cnt;
    R.state = state;
    const int fam = rank_decode_family(L.N, L.yp_stride, n_blocks);
    if (fam == 1) return launch_rank_wg(s, n_blocks, R);
    if (fam == 2) return launch_rank_big(s, n_blocks, R, active_blocks);
    const RankGeom g = rank_geometry(L.N, L.yp_stride, n_blocks);
    R.batch = g.batch;
    R.log2_cwp = g.log2_cwp;
    return g.stage ? launch_rank<true>(s, g, n_blocks, R) : launch_rank<false>(s, g, n_blocks, R);
}

hipError_t launch_rank_decode(hipStream_t s, const DecBlock* blocks, uint32_t n_blocks, const DecLines& L,
                              uint32_t* out_rows, uint32_t out_stride_w) {
    if (!n_blocks) return hipSuccess;
    RankArgs R{};
    R.blocks = blocks;
    R.wah_lines = L.wah_lines;
    rank_args_rows(R, L);
    R.wah_z = L.wah_z;
    R.out = out_rows;
    R.out_stride_w = out_stride_w;
    R.N = L.N;
    RankGeom g = rank_geometry(L.N, L.yp_stride, n_blocks);
    if (const char* e = tuning_env("XSI_DEC_B")) { uint32_t b = (uint32_t)atoi(e); if (b >= 1 && b < g.batch) g.batch = b; }
    R.batch = g.batch;
    R.log2_cwp = g.log2_cwp;
    const int fam = rank_decode_family(L.N, L.yp_stride, n_blocks);
    if (fam == 1) return launch_rank_wg(s, n_blocks, R);
    if (fam == 2) return launch_rank_big(s, n_blocks, R);
    return g.stage ? launch_rank<true>(s, g, n_blocks, R) : launch_rank<false>(s, g, n_blocks, R);
}

}  // namespace xsi
