// xsi_kernels.hip — gfx950 (MI355X, CDNA4) kernels of the xSqueezeIt genotype-block codec.
//
// Encode: count -> classify (WAH vs sparse, per-block scans) -> PBWT chain -> WAH16 sizing / writing
// (unit encoder: several lines per wave for rows of at most 32 units, one wave per four lines up to 8 KiB rows,
// one workgroup per line above; the serial one-wave encoder only where a caller asks for no scratch copy of short
// rows) -> sparse lists (one wave per line, on the side stream) -> block layout + dictionary.  The chain is the dominant kernel.  The element-major forms live in xsi_rankenc.hip
// (k_chain_rank_enc up to 65 536 haplotypes, k_chain_rank_enc_multi up to 524 288); here are the position-major ones:
//   k_chain_lds     prefix array `a` in LDS, bit columns staged through LDS, wave ballot + mbcnt for the
//                   stable partition; with fewer blocks than CUs a block is cut into line segments and a
//                   later segment gets its starting order from an LSD radix pre-pass (chain_prepass);
//   k_chain_stream  above 524 288 haplotypes: `a` ping-pongs in HBM/L2 and is read once per line, the per-segment
//                   zero counts are accumulated one line ahead;
//   k_chain_global  two-pass fallback (blocks with fully haploid lines at large N; decode of those).
// Decode mirrors it: parse -> flags -> WAH line boundaries (tiled scan) -> expansion to rank-select
// rows -> element-major chain (xsi_rank.hip), range of lines by range of lines with the expansion of the next
// range underneath (decode_planes, xsi_api.hip) -> sparse fill (side stream).
// Reference behaviour restated per kernel (paths relative to the reference tree).
#include "xsi_kernels.hpp"

#include <cstdlib>
#include <type_traits>

#include "xsi_device.hpp"

namespace xsi {

// ------------------------------------------------------------------------------------------
// dictionary key orders (libstdc++ unordered_map iteration order of GtBlock::fill_dictionary,
// gt_block.hpp:464-510).  index = missing | eov<<1 | phase<<2 | haploid<<3
// ------------------------------------------------------------------------------------------
__constant__ uint8_t c_dict_order[16][19] = {
    {9, 0x21, 0x20, 0x11, 0x04, 0x10, 0x03, 0x02, 0x01, 0x00},
    {12, 0x26, 0x16, 0x21, 0x20, 0x11, 0x04, 0x10, 0x03, 0x36, 0x02, 0x01, 0x00},
    {12, 0x18, 0x21, 0x20, 0x38, 0x11, 0x04, 0x10, 0x03, 0x02, 0x28, 0x01, 0x00},
    {15, 0x38, 0x28, 0x00, 0x01, 0x02, 0x36, 0x10, 0x11, 0x03, 0x20, 0x04, 0x21, 0x16, 0x26, 0x18},
    {11, 0x17, 0x21, 0x20, 0x11, 0x04, 0x10, 0x03, 0x02, 0x01, 0x27, 0x00},
    {14, 0x27, 0x00, 0x01, 0x02, 0x36, 0x10, 0x11, 0x03, 0x20, 0x04, 0x21, 0x16, 0x26, 0x17},
    {14, 0x27, 0x00, 0x01, 0x28, 0x02, 0x10, 0x11, 0x38, 0x03, 0x20, 0x04, 0x21, 0x18, 0x17},
    {17, 0x27, 0x17, 0x38, 0x28, 0x00, 0x01, 0x02, 0x36, 0x10, 0x11, 0x03, 0x20, 0x04, 0x21, 0x16, 0x26, 0x18},
    {10, 0x12, 0x21, 0x20, 0x11, 0x04, 0x10, 0x03, 0x02, 0x01, 0x00},
    {13, 0x12, 0x26, 0x16, 0x21, 0x20, 0x11, 0x04, 0x10, 0x03, 0x36, 0x02, 0x01, 0x00},
    {13, 0x12, 0x18, 0x21, 0x20, 0x38, 0x11, 0x04, 0x10, 0x03, 0x02, 0x28, 0x01, 0x00},
    {16, 0x12, 0x38, 0x28, 0x00, 0x01, 0x02, 0x36, 0x10, 0x11, 0x03, 0x20, 0x04, 0x21, 0x16, 0x26, 0x18},
    {12, 0x12, 0x17, 0x21, 0x20, 0x11, 0x04, 0x10, 0x03, 0x02, 0x01, 0x27, 0x00},
    {15, 0x12, 0x27, 0x00, 0x01, 0x02, 0x36, 0x10, 0x11, 0x03, 0x20, 0x04, 0x21, 0x16, 0x26, 0x17},
    {15, 0x12, 0x27, 0x00, 0x01, 0x28, 0x02, 0x10, 0x11, 0x38, 0x03, 0x20, 0x04, 0x21, 0x18, 0x17},
    {18, 0x12, 0x27, 0x17, 0x38, 0x28, 0x00, 0x01, 0x02, 0x36, 0x10, 0x11, 0x03, 0x20, 0x04, 0x21, 0x16, 0x26, 0x18},
};

__device__ __forceinline__ uint32_t nbits_of(const EncLines& L, uint32_t l) {
    return L.bin_nbits ? L.bin_nbits[l] : L.N;
}

// ------------------------------------------------------------------------------------------
// count: ones per bit row (allele_counts[alt] of scan_genotypes, gt_block.hpp:226-268, for
// fully called lines).  One wave per row, coalesced 256-byte reads.
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_count_rows(const uint32_t* __restrict__ planes, uint32_t stride_w,
                                                    uint32_t nbits, uint32_t n_rows, uint32_t* __restrict__ cnt) {
    const uint32_t row = blockIdx.x * 4u + (threadIdx.x >> 6);
    if (row >= n_rows) return;
    const uint32_t lane = lane_id();
    const uint32_t nw = (nbits + 31u) >> 5;
    const uint32_t* r = planes + (size_t)row * stride_w;
    uint32_t c = 0;
    for (uint32_t w = lane; w < nw; w += 64u) {
        uint32_t v = r[w];
        if (w == nw - 1u && (nbits & 31u)) v &= (1u << (nbits & 31u)) - 1u;
        c += (uint32_t)__popc(v);
    }
    c = wave_sum(c);
    if (lane == 0) cnt[row] = c;
}

// Same for rows whose stride is a multiple of 16 bytes (the packed input: rows padded to 128 B): a
// wave streams 8 consecutive rows as one run of 16-byte loads, all issued before the first popcount,
// and adds into per-row LDS counters.  4 waves x 8 rows per workgroup.
__global__ void __launch_bounds__(256) k_count_rows_v4(const uint4* __restrict__ planes, uint32_t stride_q,
                                                       uint32_t nbits, uint32_t n_rows, uint32_t* __restrict__ cnt) {
    constexpr uint32_t RPW = 8, MAXQ = 8;  // rows per wave; 16-byte loads per lane (RPW*stride_q <= 64*MAXQ)
    __shared__ uint32_t s_cnt[4 * RPW];
    const uint32_t lane = lane_id(), w = threadIdx.x >> 6;
    const uint32_t row0 = (blockIdx.x * 4u + w) * RPW;
    if (threadIdx.x < 4u * RPW) s_cnt[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t rows = row0 < n_rows ? (n_rows - row0 < RPW ? n_rows - row0 : RPW) : 0u;
    const uint32_t total_q = rows * stride_q;
    const uint4* base = planes + (size_t)row0 * stride_q;
    const uint32_t nw = (nbits + 31u) >> 5;
    uint4 v[MAXQ];
#pragma unroll
    for (uint32_t i = 0; i < MAXQ; ++i) {
        const uint32_t q = i * 64u + lane;
        v[i] = q < total_q ? base[q] : make_uint4(0, 0, 0, 0);
    }
#pragma unroll
    for (uint32_t i = 0; i < MAXQ; ++i) {
        const uint32_t q = i * 64u + lane;
        if (q < total_q) {
            const uint32_t r = q / stride_q, w0 = (q - r * stride_q) * 4u;  // row in my group, first word of the quad
            const uint32_t x[4] = {v[i].x, v[i].y, v[i].z, v[i].w};
            uint32_t c = 0;
#pragma unroll
            for (uint32_t j = 0; j < 4; ++j) {
                const uint32_t wi = w0 + j;
                uint32_t t = wi < nw ? x[j] : 0u;
                if (wi == nw - 1u && (nbits & 31u)) t &= (1u << (nbits & 31u)) - 1u;
                c += (uint32_t)__popc(t);
            }
            if (c) atomicAdd(&s_cnt[w * RPW + r], c);
        }
    }
    __syncthreads();
    if (threadIdx.x < 4u * RPW) {
        const uint32_t row = blockIdx.x * 4u * RPW + threadIdx.x;
        if (row < n_rows) cnt[row] = s_cnt[threadIdx.x];
    }
}

// Long rows (stride a multiple of 16 bytes, above 1 KiB): one wave per row, 16-byte loads, eight in flight per
// lane before the first popcount (the dword loop above reaches 5.4 TB/s at configs[2]; this is the pass that reads
// the whole input once, so it is priced against the streaming rate of the chip).
__global__ void __launch_bounds__(256) k_count_rows_wide(const uint4* __restrict__ planes, uint32_t stride_q,
                                                         uint32_t nbits, uint32_t n_rows, uint32_t* __restrict__ cnt) {
    constexpr uint32_t MAXQ = 8;
    const uint32_t row = blockIdx.x * 4u + (threadIdx.x >> 6);
    if (row >= n_rows) return;
    const uint32_t lane = lane_id();
    const uint32_t nw = (nbits + 31u) >> 5, nq = (nw + 3u) >> 2;
    const uint4* base = planes + (size_t)row * stride_q;
    const uint32_t last_mask = (nbits & 31u) ? (1u << (nbits & 31u)) - 1u : ~0u;
    uint32_t c = 0;
    for (uint32_t q0 = 0; q0 < nq; q0 += 64u * MAXQ) {
        uint4 v[MAXQ];
#pragma unroll
        for (uint32_t i = 0; i < MAXQ; ++i) {
            const uint32_t q = q0 + i * 64u + lane;
            typedef uint32_t cnt_u32x4 __attribute__((ext_vector_type(4)));
            const cnt_u32x4 t = __builtin_nontemporal_load(reinterpret_cast<const cnt_u32x4*>(base + (q < nq ? q : nq - 1u)));
            v[i] = make_uint4(t[0], t[1], t[2], t[3]);  // unconditional: all in flight together
        }
#pragma unroll
        for (uint32_t i = 0; i < MAXQ; ++i) {
            const uint32_t q = q0 + i * 64u + lane;
            const uint32_t x[4] = {v[i].x, v[i].y, v[i].z, v[i].w};
#pragma unroll
            for (uint32_t j = 0; j < 4; ++j) {
                const uint32_t wi = q * 4u + j;
                uint32_t t = (q < nq && wi < nw) ? x[j] : 0u;
                if (wi == nw - 1u) t &= last_mask;
                c += (uint32_t)__popc(t);
            }
        }
    }
    c = wave_sum(c);
    if (lane == 0) cnt[row] = c;
}

__global__ void __launch_bounds__(256) k_wah_lines_per_block(const uint32_t* __restrict__ cnt, uint64_t n_lines,
                                                             uint32_t block_len, uint32_t nbits, uint32_t thr,
                                                             uint32_t* __restrict__ out) {
    __shared__ uint32_t part[4];
    const uint64_t l0 = (uint64_t)blockIdx.x * block_len;
    uint32_t n = 0;
    for (uint32_t i = threadIdx.x; i < block_len && l0 + i < n_lines; i += 256u) {
        const uint32_t c = cnt[l0 + i];
        const uint32_t minor = c < nbits - c ? c : nbits - c;
        n += minor > thr ? 1u : 0u;  // the rule of k_classify on fully called diploid lines
    }
    const uint32_t inc = wave_scan_incl_dpp(n);
    if (lane_id() == 63u) part[threadIdx.x >> 6] = inc;
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = part[0] + part[1] + part[2] + part[3];
}

hipError_t launch_wah_lines_per_block(hipStream_t s, const uint32_t* cnt, uint64_t n_lines, uint32_t block_len, uint32_t nbits,
                                      uint32_t thr, uint32_t* out) {
    const uint64_t n_blocks = (n_lines + block_len - 1) / block_len;
    if (!n_blocks) return hipSuccess;
    k_wah_lines_per_block<<<dim3((uint32_t)n_blocks), dim3(256), 0, s>>>(cnt, n_lines, block_len, nbits, thr, out);
    return hipGetLastError();
}

__global__ void __launch_bounds__(256) k_compare_u32(const uint32_t* __restrict__ a, const uint32_t* __restrict__ b, uint64_t n,
                                                     uint32_t* __restrict__ n_diff) {
    const uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    const uint64_t m = __ballot(i < n && a[i] != b[i]);
    if (m && lane_id() == 0) atomicAdd(n_diff, (uint32_t)__popcll(m));
}

hipError_t launch_compare_u32(hipStream_t s, const uint32_t* a, const uint32_t* b, uint64_t n, uint32_t* n_diff) {
    if (!n) return hipSuccess;
    k_compare_u32<<<dim3((uint32_t)((n + 255u) / 256u)), dim3(256), 0, s>>>(a, b, n, n_diff);
    return hipGetLastError();
}

hipError_t launch_count_rows(hipStream_t s, const uint32_t* planes, uint32_t stride_w, uint32_t nbits,
                             uint32_t n_rows, uint32_t* cnt) {
    if (!n_rows) return hipSuccess;
    if ((stride_w & 3u) == 0 && (reinterpret_cast<uintptr_t>(planes) & 15u) == 0 && stride_w / 4u * 8u <= 64u * 8u) {
        k_count_rows_v4<<<dim3((n_rows + 31u) / 32u), dim3(256), 0, s>>>(reinterpret_cast<const uint4*>(planes),
                                                                          stride_w / 4u, nbits, n_rows, cnt);
        return hipGetLastError();
    }
    if ((stride_w & 3u) == 0 && (reinterpret_cast<uintptr_t>(planes) & 15u) == 0 && !tuning_env("XSI_COUNT_DWORD")) {
        k_count_rows_wide<<<dim3((n_rows + 3u) / 4u), dim3(256), 0, s>>>(reinterpret_cast<const uint4*>(planes), stride_w / 4u,
                                                                        nbits, n_rows, cnt);
        return hipGetLastError();
    }
    k_count_rows<<<dim3((n_rows + 3u) / 4u), dim3(256), 0, s>>>(planes, stride_w, nbits, n_rows, cnt);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// classify: per binary line WAH-vs-sparse decision (gt_block.hpp:298-327):
//   minor = min(cnt, ngt - cnt); WAH iff minor > MAC threshold; sparse lists the ALT positions
//   when cnt == minor, else the REF positions with the MSB of the count set.
// One workgroup per block; per-block exclusive scans give each WAH line its rank and each
// sparse line its byte offset.  Also packs the is-WAH flag vector (KEY_LINE_SORT/SELECT).
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(1024) k_classify(EncBlock* __restrict__ blocks, EncLines L) {
    __shared__ uint64_t s_scan[20];
    const uint32_t b = blockIdx.x;
    EncBlock& B = blocks[b];
    const uint32_t first = B.first_bin, n = B.n_bin;
    uint32_t wah_base = 0, sp_base = 0;
    int any_hap = 0;
    uint32_t* fb = L.flagbits + ((size_t)b * FV_COUNT + FV_IS_WAH) * (MAX_BIN_PER_BLOCK / 32);
    for (uint32_t c0 = 0; c0 < n; c0 += blockDim.x) {
        const uint32_t i = c0 + threadIdx.x;
        const bool valid = i < n;
        const uint32_t l = first + i;
        uint32_t is_wah = 0, sp_bytes = 0;
        if (valid) {
            const uint32_t ngt = nbits_of(L, l);
            const uint32_t c = L.cnt[l];
            const uint32_t minor = c < ngt - c ? c : ngt - c;
            uint32_t k = L.kind[l] & KIND_HAPLOID;
            if (minor > L.thr) {
                is_wah = 1;
                k |= KIND_WAH;
            } else {
                uint32_t listed = c;
                if (c != minor) {
                    k |= KIND_NEGATED;
                    listed = L.ref_cnt ? L.ref_cnt[L.bin_parent[l]] : ngt - c;
                }
                sp_bytes = (1u + listed) * L.aet;
            }
            L.kind[l] = (uint8_t)k;
            L.line_block[l] = b;
            if (k & KIND_HAPLOID) any_hap = 1;
        }
        const uint64_t W = __ballot(is_wah);
        if (lane_id() == 0 && c0 + (threadIdx.x & ~63u) < n) {
            const uint32_t wi = (c0 + threadIdx.x) >> 5;
            fb[wi] = (uint32_t)W;
            fb[wi + 1] = (uint32_t)(W >> 32);
        }
        uint64_t tot;
        const uint64_t ex = block_scan_excl64(((uint64_t)is_wah << 40) | sp_bytes, s_scan, &tot);
        if (valid) {
            L.wah_rank[l] = wah_base + (uint32_t)(ex >> 40);
            L.sparse_off[l] = sp_base + (uint32_t)(ex & 0xFFFFFFFFFFull);
        }
        wah_base += (uint32_t)(tot >> 40);
        sp_base += (uint32_t)(tot & 0xFFFFFFFFFFull);
    }
    any_hap = __syncthreads_or(any_hap);
    if (threadIdx.x == 0) {
        B.n_wah = wah_base;
        B.sparse_bytes = sp_base;
        B.has_haploid = any_hap ? 1u : 0u;  // chain kernel selection; block_layout recomputes it from the BCF flags
    }
}

hipError_t launch_classify(hipStream_t s, EncBlock* blocks, uint32_t n_blocks, const EncLines& L) {
    if (!n_blocks) return hipSuccess;
    k_classify<<<dim3(n_blocks), dim3(1024), 0, s>>>(blocks, L);
    return hipGetLastError();
}

// exclusive scan of n_wah over the blocks of the batch (single workgroup; batches are <= a few
// thousand blocks).  totals[0] = WAH lines in the batch.
__global__ void __launch_bounds__(1024) k_scan_blocks_wah(EncBlock* __restrict__ blocks, uint32_t n_blocks,
                                                          uint32_t* __restrict__ totals) {
    __shared__ uint64_t s_scan[20];
    uint32_t base = 0;
    for (uint32_t c0 = 0; c0 < n_blocks; c0 += blockDim.x) {
        const uint32_t i = c0 + threadIdx.x;
        const uint32_t v = i < n_blocks ? blocks[i].n_wah : 0u;
        uint64_t tot;
        const uint64_t ex = block_scan_excl64(v, s_scan, &tot);
        if (i < n_blocks) blocks[i].wah_first = base + (uint32_t)ex;
        base += (uint32_t)tot;
    }
    if (threadIdx.x == 0) totals[0] = base;
}

hipError_t launch_scan_blocks_wah(hipStream_t s, EncBlock* blocks, uint32_t n_blocks, uint32_t* totals) {
    k_scan_blocks_wah<<<dim3(1), dim3(1024), 0, s>>>(blocks, n_blocks, totals);
    return hipGetLastError();
}

__global__ void __launch_bounds__(256) k_build_wah_list(const EncBlock* __restrict__ blocks, EncLines L) {
    const EncBlock& B = blocks[blockIdx.x];
    for (uint32_t i = threadIdx.x; i < B.n_bin; i += blockDim.x) {
        const uint32_t l = B.first_bin + i;
        if (L.kind[l] & KIND_WAH) L.wah_lines[B.wah_first + L.wah_rank[l]] = l;
    }
}

hipError_t launch_build_wah_list(hipStream_t s, const EncBlock* blocks, uint32_t n_blocks, const EncLines& L) {
    if (!n_blocks) return hipSuccess;
    k_build_wah_list<<<dim3(n_blocks), dim3(256), 0, s>>>(blocks, L);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// PBWT chain.
//
// Per block, for every WAH line k in order (sparse lines never touch `a`, gt_block.hpp:299-326):
//   encode: y_k[i] = x_k[a_k[i]]                      (wah.hpp:530-537 gather through a)
//           a_{k+1} = [a_k[i] : y_k[i]=0] ++ [a_k[i] : y_k[i]=1]   (internal_gt_record.hpp:32-59)
//   decode: x_k[a_k[i]] = y_k[i]                      (accessor_internals_new.hpp:228-230)
//           same partition by y_k                     (gt_block.hpp:124-136)
// `a` restarts at identity for every block (gt_block.hpp:179; accessor_internals_new.hpp:144).
//
// MI355X mapping: one workgroup (T threads = T/64 waves) owns a block.  `a` lives in LDS as
// uint16 (N <= 65536), wave w owns positions [w*E*64, (w+1)*E*64), chunk e of a wave is 64
// consecutive positions, one per lane.  For each chunk the 64 key bits are one ballot; a lane's
// destination is (zeros before my wave) + (zeros in my earlier chunks) + mbcnt(zero mask) for a
// 0 and the mirrored expression offset by the line's total zeros for a 1, so the partition is
// stable by construction.  Every wave reads its positions into registers before the barrier
// that publishes the per-wave zero counts, so the scatter can go back into the same LDS array.
// Bit columns are prefetched in batches (global -> registers while the previous batch is
// processed -> LDS) so the serial chain never waits on HBM.
// Fully haploid lines (general path) use the slow sub-path chain_step_haploid.
// ------------------------------------------------------------------------------------------
struct ChainArgs {
    const uint32_t* wah_lines;   // [rank] binary line
    const uint8_t* kind;         // per binary line
    const uint32_t* src;         // encode: planes (by binary line); decode: yrows as uint32 (by rank)
    uint32_t src_stride_w;
    uint32_t src_elem_shift;     // 0: src rows are plain words; 1: {word, prefix} pairs (decode)
    uint32_t* dst;               // encode: yrows as uint32 (by rank); decode: output rows (by binary line)
    uint32_t dst_stride_w;
    uint32_t N;
    uint32_t cw;                 // words per column in LDS (even)
    uint32_t log2_cwp;           // log2 of next pow2 >= cw
    uint32_t batch;              // columns per prefetch batch
    uint32_t out_row_base;       // decode: first output row of the batch (binary line numbering offset)
    uint32_t only_haploid_blocks;// decode: skip blocks the element-major kernel already handled
    uint32_t plain_done;         // encode: blocks without fully haploid lines are already encoded (rank tracking)
    uint32_t segments;           // encode, LDS kernel: workgroups per block (line segments), >= 1
    uint32_t seg_q16[5];         // cumulative segment boundaries as fractions of n_wah (Q16), [0] = 0
};

constexpr uint32_t PRE_ID_CAP = 8192;  // line ids of a segment's prefix held in LDS

constexpr int CHAIN_RMAX = 4;

// Per-lane key bits of up to 64 chunks, kept as two 32-bit halves so every access is one VALU op.
template <int E>
struct KeyBits {
    uint32_t lo = 0, hi = 0;
    template <int e>
    __device__ __forceinline__ void set(uint32_t bit01) {
        if (e < 32)
            lo |= bit01 << (e & 31);
        else
            hi |= bit01 << (e & 31);
    }
    template <int e>
    __device__ __forceinline__ uint32_t get() const {
        return ((e < 32 ? lo : hi) >> (e & 31)) & 1u;
    }
};

// Stable partition of `a` by the per-lane key bits; zc = zeros of my wave.  Shared by the fast
// and the haploid paths.  `nvalid_total` = N for the haploid path (pads excluded), na otherwise.
template <int T, int E, bool PADS_ARE_ONES, typename AT>
__device__ __forceinline__ void chain_scatter(AT* a, uint32_t* wcnt, const uint32_t (&av)[E],
                                              const KeyBits<E>& keys, uint32_t zc, uint32_t w, uint32_t lane,
                                              uint32_t N, uint32_t na) {
    constexpr int W = T / 64;
    if (lane == 0) wcnt[w] = zc;
    lds_barrier();
    uint32_t sc = row16_scan_incl(lane < (uint32_t)W ? wcnt[lane] : 0u);
    const uint32_t tz = (uint32_t)__builtin_amdgcn_readlane((int)sc, W - 1);
    uint32_t zb = w ? (uint32_t)__builtin_amdgcn_readlane((int)sc, (int)w - 1) : 0u;
    uint32_t before = w * E * 64u;
    if (!PADS_ARE_ONES && before > N) before = N;
    uint32_t ob = tz + before - zb;  // destination of my wave's first one
    static_for<0, E>([&](auto ec) {
        constexpr int e = decltype(ec)::value;
        const uint32_t base = (w * E + (uint32_t)e) * 64u;
        if (base < (PADS_ARE_ONES ? na : N)) {
            const uint32_t bit = keys.template get<e>();
            uint64_t om = __ballot(bit);
            uint32_t nv = 64u;
            if (!PADS_ARE_ONES) {
                nv = (N - base >= 64u) ? 64u : (N - base);
                if (nv < 64u) om &= (1ull << nv) - 1ull;
            }
            const uint64_t zm = PADS_ARE_ONES ? ~om : (~om & (nv < 64u ? (1ull << nv) - 1ull : ~0ull));
            const uint32_t zpre = mbcnt64(zm);
            const uint32_t opre = PADS_ARE_ONES ? lane - zpre : mbcnt64(om);
            const uint32_t dest = bit ? ob + opre : zb + zpre;
            if (PADS_ARE_ONES || lane < nv) a[dest] = (AT)av[e];
            const uint32_t nz = (uint32_t)__popcll(zm);
            zb += nz;
            ob += (PADS_ARE_ONES ? 64u : nv) - nz;
        }
    });
    lds_barrier();
}

// Fully haploid line (ngt == n_samples): gt_block.hpp:304-309 + pbwt_sort1
// (internal_gt_record.hpp:55-58); accessor_internals_new.hpp:222-226, 548-571.  y (n_samples
// bits) is ordered by a1 = even members of a, halved (interfaces.hpp:318-333); the partition key
// of a[i] is the bit of sample a[i]/2.  Rare, kept out of line so it costs the fast path nothing.
// LDS arrays are passed as byte offsets into the dynamic LDS segment: handing LDS pointers to an
// out-of-line function makes hipcc (ROCm 7.2) cast them to flat and trip over its own null check.
template <int T, int E, bool DECODE, typename AT>
__device__ __attribute__((noinline)) void chain_step_haploid(uint32_t a_off, uint32_t c_off, uint32_t xrow_off,
                                                             uint32_t wcnt_off, uint32_t N, uint32_t na, uint32_t cw,
                                                             uint32_t* orow, uint32_t orow_words) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    AT* a = reinterpret_cast<AT*>(smem + a_off);
    const uint32_t* c = reinterpret_cast<const uint32_t*>(smem + c_off);
    uint32_t* xrow = reinterpret_cast<uint32_t*>(smem + xrow_off);
    uint32_t* wcnt = reinterpret_cast<uint32_t*>(smem + wcnt_off);
    constexpr int W = T / 64;
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const uint32_t w = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6));
    uint32_t* ev = wcnt + W;
    uint32_t av[E];
    KeyBits<E> evens, keys;
    uint32_t ec = 0;
    static_for<0, E>([&](auto ecn) {
        constexpr int e = decltype(ecn)::value;
        const uint32_t base = (w * E + (uint32_t)e) * 64u;
        av[e] = 1;
        if (base < N) {
            const uint32_t idx = base + lane;
            const uint32_t v = idx < N ? (uint32_t)a[idx] : 1u;
            av[e] = v;
            const uint32_t even = (idx < N && !(v & 1u)) ? 1u : 0u;
            evens.template set<e>(even);
            ec += (uint32_t)__popcll(__ballot(even));
        }
    });
    if (lane == 0) ev[w] = ec;
    lds_barrier();
    uint32_t eb = 0;
    for (uint32_t i = 0; i < w; ++i) eb += ev[i];
    if (DECODE) {
        // x[a1[p]] = y[p]
        uint32_t ebw = eb;
        static_for<0, E>([&](auto ecn) {
            constexpr int e = decltype(ecn)::value;
            const uint32_t base = (w * E + (uint32_t)e) * 64u;
            if (base < N) {
                const uint32_t even = evens.template get<e>();
                const uint64_t evm = __ballot(even);
                const uint32_t p = ebw + mbcnt64(evm);
                if (even && ((c[p >> 5] >> (p & 31u)) & 1u)) {
                    const uint32_t sidx = av[e] >> 1;
                    atomicOr(&xrow[sidx >> 5], 1u << (sidx & 31u));
                }
                ebw += (uint32_t)__popcll(evm);
            }
        });
        lds_barrier();
    }
    const uint32_t* keycol = DECODE ? xrow : c;
    uint32_t zc = 0;
    static_for<0, E>([&](auto ecn) {
        constexpr int e = decltype(ecn)::value;
        const uint32_t base = (w * E + (uint32_t)e) * 64u;
        if (base < N) {
            const bool valid = base + lane < N;
            const uint32_t sidx = av[e] >> 1;
            const uint32_t bit = valid ? ((keycol[sidx >> 5] >> (sidx & 31u)) & 1u) : 0u;
            keys.template set<e>(bit);
            const uint32_t nv = (N - base >= 64u) ? 64u : (N - base);
            zc += nv - (uint32_t)__popcll(__ballot(bit));
        }
    });
    if (DECODE) {
        lds_barrier();  // every wave has read its keys from xrow
    } else {
        // y[p] = key of the p-th even member
        uint32_t ebw = eb;
        static_for<0, E>([&](auto ecn) {
            constexpr int e = decltype(ecn)::value;
            const uint32_t base = (w * E + (uint32_t)e) * 64u;
            if (base < N) {
                const uint32_t even = evens.template get<e>();
                const uint64_t evm = __ballot(even);
                const uint32_t p = ebw + mbcnt64(evm);
                if (even && keys.template get<e>()) atomicOr(&xrow[p >> 5], 1u << (p & 31u));
                ebw += (uint32_t)__popcll(evm);
            }
        });
        lds_barrier();
    }
    // publish xrow (decode: x by sample; encode: y in a1 order), n_samples bits
    const uint32_t nbits = N >> 1;
    for (uint32_t i = tid; i < orow_words; i += T) {
        uint32_t v = 0;
        if (i < cw) {
            v = xrow[i];
            xrow[i] = 0;
            const uint32_t b0 = i * 32u;
            if (b0 + 32u > nbits) v &= (b0 >= nbits) ? 0u : ((1u << (nbits - b0)) - 1u);
        }
        orow[i] = v;
    }
    chain_scatter<T, E, false, AT>(a, wcnt, av, keys, zc, w, lane, N, na);
}

// Segment pre-pass.  With fewer blocks than CUs a block's chain is cut into line segments, one
// workgroup each.  A segment that starts at WAH line s0 needs a_{s0}; that order is the stable sort
// of the haplotypes by their bits on lines [0, s0) (last line most significant), i.e. an LSD radix
// sort with the lines as digits.  No permuted row has to be produced for those lines (the previous
// segment's workgroup does that), so 4 lines are taken per pass: one pass = read `a`, read a 4-bit
// key per member, 16-bin stable counting sort, scatter; 2 barriers per 4 lines instead of 8.
//   keys    : nib[] holds one nibble per haplotype (bit k = line 4p+k), built from the 4 bit rows
//             (prefetched one pass ahead); padding members get 0xF and stay last, as in the chain;
//   in-wave : peer mask of a lane = lanes of its chunk with the same key (4 ballots), rank = mbcnt;
//             hist[key][wave] doubles as the running count of earlier chunks;
//   x-wave  : every wave scans the 16 x W counters (bin-major) itself, lanes fetch their bin base
//             with one ds_bpermute.
// The result equals the chain's `a` after s0 lines exactly (both are stable).
template <int T, int E, typename AT>
__device__ __forceinline__ void chain_prepass(AT* a, uint32_t* pre_ids, uint32_t* nib, uint32_t* hist,
                                              const ChainArgs& A, uint32_t wah_first, uint32_t s0, uint32_t tid,
                                              uint32_t lane, uint32_t w) {
    constexpr int W = T / 64;
    constexpr uint32_t NA = (uint32_t)T * E, NW = NA / 8u;  // nibble words: 8 haplotypes each
    constexpr uint32_t KW = (NW + (uint32_t)T - 1u) / (uint32_t)T;  // nibble words per thread (1 for E <= 8)
    static_assert(W == 16 || W == 4, "scan layouts below");
    const uint32_t N = A.N;
    const uint32_t src_words = (N + 31u) >> 5;
    for (uint32_t i = tid; i < s0; i += T) pre_ids[i] = A.wah_lines[wah_first + i];
    for (uint32_t i = tid; i < 2u * 16u * W; i += T) hist[i] = 0;
    lds_barrier();
    // my nibble word q covers haplotypes 8*(q*T + tid) ..+7: byte (tid & 3) of row word (q*T + tid) / 4
    uint32_t R4[KW][4];
    auto load_rows = [&](uint32_t p) {
#pragma unroll
        for (uint32_t q = 0; q < KW; ++q) {
            const uint32_t nwi = q * (uint32_t)T + tid, rw = nwi >> 2;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint32_t line = pre_ids[4u * p + (uint32_t)k];
                R4[q][k] = (nwi < NW && rw < src_words) ? A.src[(size_t)line * A.src_stride_w + rw] : 0u;
            }
        }
    };
    auto build_nib = [&]() {
#pragma unroll
        for (uint32_t q = 0; q < KW; ++q) {
            const uint32_t nwi = q * (uint32_t)T + tid, rw = nwi >> 2, rb = (nwi & 3u) * 8u;
            if (nwi < NW) {
                uint32_t pad_or = 0;  // bits at or beyond N read as 1
                if (rw * 32u + 32u > N) pad_or = (rw * 32u >= N) ? 0xFFFFFFFFu : (0xFFFFFFFFu << (N - rw * 32u));
                uint32_t out = 0;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    uint32_t x = ((R4[q][k] | pad_or) >> rb) & 0xFFu;  // 8 haplotypes of line 4p+k
                    x = (x | (x << 12)) & 0x000F000Fu;
                    x = (x | (x << 6)) & 0x03030303u;
                    x = (x | (x << 3)) & 0x11111111u;                  // bit i -> bit 4i
                    out |= x << k;
                }
                nib[nwi] = out;
            }
        }
    };
    load_rows(0);
    build_nib();
    lds_barrier();
    const uint32_t n_pass = s0 >> 2;
    AT* aw = a + w * (E * 64u) + lane;
    // LDS integer addresses with the arrays' own addresses folded into wave-uniform constants
    using LdsU32 = __attribute__((address_space(3))) uint32_t;
    using LdsAT = __attribute__((address_space(3))) AT;
    constexpr uint32_t ASH = sizeof(AT) == 4 ? 2u : 1u;
    constexpr uint32_t LOGW = (W == 16) ? 4u : 2u;
    const uint32_t a_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)a;
    const uint32_t nib_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)nib;
    const uint32_t hist_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)hist;
    for (uint32_t p = 0; p < n_pass; ++p) {
        uint32_t* h_cur = hist + (p & 1u) * 16u * W;
        uint32_t* h_nxt = hist + ((p + 1u) & 1u) * 16u * W;
        const uint32_t hcur_w = hist_lds + ((p & 1u) * 16u * W + w) * 4u;  // &h_cur[0][w]
        const bool more = p + 1u < n_pass;
        if (more) load_rows(p + 1u);
        uint32_t av[E], key[E], rk[E];
        static_for<0, E>([&](auto ecn) {
            constexpr int e = decltype(ecn)::value;
            av[e] = (uint32_t)aw[e * 64];
        });
        static_for<0, E>([&](auto ecn) {
            constexpr int e = decltype(ecn)::value;
            const uint32_t v = av[e];
            const uint32_t word = *reinterpret_cast<LdsU32*>((uintptr_t)(nib_lds + ((v >> 1) & ~3u)));
            key[e] = __builtin_amdgcn_ubfe(word, (v << 2) & 28u, 4u);
        });
        static_for<0, E>([&](auto ecn) {
            constexpr int e = decltype(ecn)::value;
            uint32_t pm_lo = 0xFFFFFFFFu, pm_hi = 0xFFFFFFFFu;  // lanes of this chunk with my key
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint32_t nb = (uint32_t)__builtin_amdgcn_sbfe((int)key[e], k, 1);  // 0 or ~0
                const uint64_t Bk = __ballot(nb != 0u);
                pm_lo &= ~((uint32_t)Bk ^ nb);
                pm_hi &= ~((uint32_t)(Bk >> 32) ^ nb);
            }
            const uint32_t before = __builtin_amdgcn_mbcnt_hi(pm_hi, __builtin_amdgcn_mbcnt_lo(pm_lo, 0u));
            const uint32_t cnt = (uint32_t)__popc(pm_lo) + (uint32_t)__popc(pm_hi);
            LdsU32* hp = reinterpret_cast<LdsU32*>((uintptr_t)(hcur_w + (key[e] << (2u + LOGW))));
            const uint32_t old = *hp;  // same-key members in my wave's earlier chunks
            if (before == 0u) *hp = old + cnt;
            rk[e] = (old + before) << ASH;  // byte offset inside my bin
        });
        lds_barrier();  // counters complete; every wave has read `a` and nib
        // exclusive prefix over the 16 x W counters in bin-major order (every wave for itself)
        uint32_t exv;
        if constexpr (W == 16) {
            const uint4 c = reinterpret_cast<const uint4*>(h_cur)[lane];  // counters 4*lane .. 4*lane+3
            const uint32_t tot = c.x + c.y + c.z + c.w;
            const uint32_t ex0 = wave_scan_incl_dpp(tot) - tot;
            const uint32_t comp = w & 3u;  // counter (key, w) sits in lane key*4 + w/4, component w%4
            exv = ex0 + (comp > 0u ? c.x : 0u) + (comp > 1u ? c.y : 0u) + (comp > 2u ? c.z : 0u);
        } else {
            const uint32_t c = h_cur[lane];
            exv = wave_scan_incl_dpp(c) - c;
        }
        exv = (exv << ASH) + a_lds;  // LDS byte address of the bin's first slot
        const uint32_t src_base = (W == 16) ? (w >> 2) << 2 : w << 2;  // bpermute byte index of lane (key*4 + ...)
        static_for<0, E>([&](auto ecn) {
            constexpr int e = decltype(ecn)::value;
            const uint32_t base = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((key[e] << 4) + src_base), (int)exv);
            *reinterpret_cast<LdsAT*>((uintptr_t)(base + rk[e])) = (AT)av[e];
        });
        if (lane < 16u) h_nxt[lane * W + w] = 0;
        if (more) build_nib();
        lds_barrier();  // scatter done, next keys in place
    }
}

// AT = element type of `a` in LDS: uint32_t while the array fits (N <= 32768: full-rate 32-bit
// LDS writes in the scatter), uint16_t beyond (adjacent lanes then share a dword, measured ~2x the
// scatter cost, but 65536 members still fit one CU's LDS).
template <int T, int E, bool DECODE, typename AT>
__global__ void __launch_bounds__(T) k_chain_lds(const EncBlock* __restrict__ eblocks,
                                                 const DecBlock* __restrict__ dblocks, ChainArgs A) {
    constexpr int W = T / 64;
    constexpr uint32_t NA = (uint32_t)T * E;  // capacity: N real members + (NA - N) padding members
    constexpr uint32_t CW = NA / 32u;         // words per staged column
    static_assert(E <= 64 && W <= 16, "per-lane bitfields are 64 bits wide; counts scanned in one DPP row");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t N = A.N;
    AT* a = reinterpret_cast<AT*>(smem);
    uint32_t* col = reinterpret_cast<uint32_t*>(smem + (size_t)NA * sizeof(AT));
    uint32_t* xrow = col + 2u * A.batch * CW;  // CW words (decode scatter target / haploid scratch)
    uint32_t* wcnt = xrow + CW;                // 2*W words
    uint32_t* linfo = wcnt + 2 * W;            // 3 x 16: binary line | haploid<<31, two batches ahead
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    // the wave index is uniform: telling the compiler so turns per-wave arithmetic into SALU
    const uint32_t w = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6));

    constexpr bool CAN_SPLIT = !DECODE && E <= 16;  // segment pre-pass (3 registers per chunk: E <= 16)
    uint32_t wah_first, n_wah;
    uint32_t seg_start = 0;
    if (DECODE) {
        if (dblocks[blockIdx.x].error) return;
        if (A.only_haploid_blocks && dblocks[blockIdx.x].off_line_haploid == VAL_UNDEFINED) return;
        wah_first = dblocks[blockIdx.x].wah_first;
        n_wah = dblocks[blockIdx.x].n_wah;
    } else {
        const uint32_t S = CAN_SPLIT ? A.segments : 1u;
        const uint32_t blk = blockIdx.x / S, seg = blockIdx.x - blk * S;
        if (A.only_haploid_blocks && !eblocks[blk].has_haploid) return;
        wah_first = eblocks[blk].wah_first;
        n_wah = eblocks[blk].n_wah;
        if (S > 1u) {
            // blocks with fully haploid lines, very short or very long chains stay in one piece
            if (eblocks[blk].has_haploid || n_wah < 64u || n_wah > PRE_ID_CAP) {
                if (seg) return;
            } else {
                seg_start = (uint32_t)(((uint64_t)n_wah * A.seg_q16[seg]) >> 16) & ~3u;
                const uint32_t seg_end =
                    seg + 1u == S ? n_wah : ((uint32_t)(((uint64_t)n_wah * A.seg_q16[seg + 1u]) >> 16) & ~3u);
                if (seg_start >= seg_end) return;
                n_wah = seg_end;  // shifted below, once the pre-pass has used the prefix
            }
        }
    }
    if (n_wah == 0) return;

    for (uint32_t i = tid; i < NA; i += T) a[i] = (AT)i;
    for (uint32_t i = tid; i < CW; i += T) xrow[i] = 0;
    if constexpr (CAN_SPLIT) {
        if (seg_start) {
            uint32_t* pre_ids = linfo + 48;
            uint32_t* nib = pre_ids + PRE_ID_CAP;
            uint32_t* hist = nib + NA / 8u;
            lds_barrier();
            chain_prepass<T, E, AT>(a, pre_ids, nib, hist, A, wah_first, seg_start, tid, lane, w);
            wah_first += seg_start;
            n_wah -= seg_start;
        }
    }

    const uint32_t B = A.batch;
    const uint32_t cwp_mask = (1u << A.log2_cwp) - 1u;
    const uint32_t src_words = (N + 31u) >> 5;
    const uint32_t n_batches = (n_wah + B - 1u) / B;
    uint32_t R[CHAIN_RMAX];
    uint32_t Rinfo = 0;

    // Nothing in the serial chain may wait on HBM.  Line ids (+ haploid flag) are fetched two
    // batches ahead into an LDS ring; with them the bit columns are fetched one batch ahead into
    // registers while the current batch is processed, then parked in LDS.
    // line id, then its kind byte: two dependent loads, taken one batch apart (ids three batches ahead, kinds two) so
    // that neither is waited for where it is issued.  Both are unconditional (clamped index, dummy address without
    // a kind array): a load under a branch makes the compiler drain vmcnt at the loop head.
    uint32_t Rline = 0, Rkind = 0;
    // (explicitly a global pointer: the select of two pointers is "generic" to the compiler, and a flat load counts in
    //  lgkmcnt as well - the first LDS wait of the batch would wait for it)
    using GlobU8 = const __attribute__((address_space(1))) uint8_t;
    GlobU8* kind_p = A.kind ? (GlobU8*)A.kind : (GlobU8*)A.wah_lines;
    const uint32_t kind_mask = A.kind ? KIND_HAPLOID : 0u;
    auto load_line = [&](uint32_t bt) {
        const uint32_t j = bt * B + (tid < B ? tid : 0u);
        Rline = A.wah_lines[wah_first + (j < n_wah ? j : n_wah - 1u)];
    };
    auto load_kind = [&]() { Rkind = kind_p[A.kind ? Rline : 0u]; };  // of the ids in Rline
    auto store_info = [&](uint32_t bt) {  // ids in Rline, kinds in Rkind: both of batch bt
        Rinfo = (bt * B + tid < n_wah) ? (Rline | ((Rkind & kind_mask) ? 0x80000000u : 0u)) : 0u;
        if (tid < B) linfo[(bt % 3u) * 16u + tid] = Rinfo;
    };
    // unconditional loads (word 0 of the source where there is nothing to fetch; store_cols zeroes those): see above
    auto load_cols = [&](uint32_t bt) {
#pragma unroll
        for (int r = 0; r < CHAIN_RMAX; ++r) {
            const uint32_t idx = (uint32_t)r * T + tid;
            const uint32_t jj = idx >> A.log2_cwp, wi = idx & cwp_mask;
            const uint32_t j = bt * B + jj;
            const bool ok = jj < B && j < n_wah && wi < src_words;
            const size_t row = DECODE ? (size_t)(wah_first + j) : (size_t)(linfo[(bt % 3u) * 16u + (ok ? jj : 0u)] & 0x7FFFFFFFu);
            R[r] = A.src[ok ? row * A.src_stride_w + ((size_t)wi << A.src_elem_shift) : (size_t)0];
        }
    };
    auto store_cols = [&](uint32_t bt) {
        const uint32_t buf = bt & 1u;
#pragma unroll
        for (int r = 0; r < CHAIN_RMAX; ++r) asm volatile("" ::"v"(R[r]));  // waited for on every path
#pragma unroll
        for (int r = 0; r < CHAIN_RMAX; ++r) {
            const uint32_t idx = (uint32_t)r * T + tid;
            const uint32_t jj = idx >> A.log2_cwp, wi = idx & cwp_mask;
            if (jj < B && wi < CW) {
                // positions [N, NA) of `a` hold padding members whose key is always 1, so they stay
                // behind every real member (stable partition) and never need a validity test
                uint32_t v = (bt * B + jj < n_wah && wi < src_words) ? R[r] : 0u;
                const uint32_t b0 = wi * 32u;
                if (b0 + 32u > N) v |= (b0 >= N) ? 0xFFFFFFFFu : (0xFFFFFFFFu << (N - b0));
                col[(buf * B + jj) * CW + wi] = v;
            }
        }
    };

    load_line(0);
    load_kind();
    store_info(0);
    load_line(1);
    load_kind();
    store_info(1);
    load_line(2);
    lds_barrier();
    load_cols(0);
    store_cols(0);
    lds_barrier();

    AT* aw = a + w * (E * 64u) + lane;  // my element of chunk e is aw[e*64]
    using LdsAT = __attribute__((address_space(3))) AT;
    const uint32_t a_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
    for (uint32_t bt = 0; bt < n_batches; ++bt) {
        load_kind();  // of batch bt + 2, whose ids arrived a batch ago
        const uint32_t Rline_cur = Rline;
        load_line(bt + 3u);
        const uint32_t Rline_nxt = Rline;
        Rline = Rline_cur;
        load_cols(bt + 1u);  // (beyond the last batch: nothing valid, zeros are parked)
        const uint32_t jn = (n_wah - bt * B) < B ? (n_wah - bt * B) : B;
        uint32_t info_v = linfo[(bt % 3u) * 16u];  // line id of the next step, read one step ahead
        for (uint32_t jj = 0; jj < jn; ++jj) {
            const uint32_t rank = wah_first + bt * B + jj;
            const uint32_t* c = col + ((bt & 1u) * B + jj) * CW;
            const uint32_t info = (uint32_t)__builtin_amdgcn_readfirstlane((int)info_v);
            if (jj + 1u < jn) info_v = linfo[(bt % 3u) * 16u + jj + 1u];
            const uint32_t line = info & 0x7FFFFFFFu;
            if (info >> 31) {
                uint32_t* orow = DECODE ? A.dst + (size_t)(line - A.out_row_base) * A.dst_stride_w
                                        : A.dst + (size_t)rank * A.dst_stride_w;
                chain_step_haploid<T, E, DECODE, AT>(0u, (uint32_t)((const unsigned char*)c - smem),
                                                 (uint32_t)((unsigned char*)xrow - smem),
                                                 (uint32_t)((unsigned char*)wcnt - smem), N, NA, CW, orow,
                                                 A.dst_stride_w);
                continue;
            }
            // ---- pass 1: keys of my E chunks.  For E <= 8 the E ballot masks stay in SGPRs for the
            //      scatter pass; for larger E they would spill, so a per-lane bitfield carries the keys.
            constexpr bool MASKS_IN_SGPR = (E <= 8);
            uint32_t av[E];
            uint64_t ms[MASKS_IN_SGPR ? E : 1];
            KeyBits<E> keys;
            uint32_t ones = 0;
            uint32_t mine_lo = 0, mine_hi = 0;  // encode: lane e collects chunk e's 64 permuted bits
            static_for<0, E>([&](auto ecn) {
                constexpr int e = decltype(ecn)::value;
                av[e] = (uint32_t)aw[e * 64];
            });
            static_for<0, E>([&](auto ecn) {
                constexpr int e = decltype(ecn)::value;
                const uint32_t v = av[e];
                uint64_t m;
                if (DECODE) {
                    // y is already in permuted order: the chunk's 64 key bits are one 64-bit word
                    const uint32_t cgw = (w * E + (uint32_t)e) * 2u;
                    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)c[cgw]);
                    const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)c[cgw + 1u]);
                    m = ((uint64_t)hi << 32) | lo;
                    const uint32_t bit = __builtin_amdgcn_ubfe(lane < 32u ? lo : hi, lane, 1u);
                    if (bit) atomicOr(&xrow[v >> 5], 1u << (v & 31u));
                    if (!MASKS_IN_SGPR) keys.template set<e>(bit);
                } else {
                    // v_bfe_u32 uses only the low 5 bits of its offset operand: no explicit v & 31
                    const uint32_t bit = __builtin_amdgcn_ubfe(c[v >> 5], v, 1u);
                    m = __ballot(bit != 0u);
                    mine_lo = write_lane(mine_lo, (uint32_t)m, (uint32_t)e);
                    mine_hi = write_lane(mine_hi, (uint32_t)(m >> 32), (uint32_t)e);
                    if (!MASKS_IN_SGPR) keys.template set<e>(bit);
                }
                if constexpr (MASKS_IN_SGPR) ms[e] = m;
                ones += (uint32_t)__popcll(m);
            });
            if (!DECODE) {
                // 8*E contiguous bytes per wave; bits at or beyond N are padding (ignored downstream)
                const uint32_t cg = w * E + lane;
                if (lane < (uint32_t)E && cg < A.dst_stride_w / 2u) {
                    uint2* yr = reinterpret_cast<uint2*>(A.dst + (size_t)rank * A.dst_stride_w);
                    yr[cg] = make_uint2(mine_lo, mine_hi);
                }
            }
            if (lane == 0) wcnt[w] = E * 64u - ones;
            lds_barrier();
            if (DECODE) {
                uint32_t* orow = A.dst + (size_t)(line - A.out_row_base) * A.dst_stride_w;
                for (uint32_t i = tid; i < A.dst_stride_w; i += T) {
                    uint32_t v = 0;
                    if (i < CW) {
                        v = xrow[i];
                        xrow[i] = 0;
                        const uint32_t b0 = i * 32u;
                        if (b0 + 32u > N) v &= (b0 >= N) ? 0u : ((1u << (N - b0)) - 1u);
                    }
                    orow[i] = v;
                }
            }
            // ---- pass 2: stable scatter.  Destinations are formed directly as LDS byte addresses.
            uint32_t sc = row16_scan_incl(lane < (uint32_t)W ? wcnt[lane] : 0u);
            const uint32_t tz = (uint32_t)__builtin_amdgcn_readlane((int)sc, W - 1);
            constexpr uint32_t AS = sizeof(AT), ASH = sizeof(AT) == 4 ? 2u : 1u;
            // LDS byte addresses, with the array's own LDS address folded into the wave-uniform bases
            // (one VALU op per chunk less than indexing through the `smem` symbol)
            uint32_t zb2 = (w ? (uint32_t)__builtin_amdgcn_readlane((int)sc, (int)w - 1) : 0u) * AS + a_lds;
            uint32_t ob2 = (tz + w * (E * 64u)) * AS - zb2 + 2u * a_lds;  // my wave's first one
            static_for<0, E>([&](auto ecn) {
                constexpr int e = decltype(ecn)::value;
                uint64_t om;
                if constexpr (MASKS_IN_SGPR)
                    om = ms[e];
                else
                    om = __ballot(keys.template get<e>() != 0u);
                const uint32_t zpre = mbcnt64(~om);
                const uint32_t d0 = zb2 + (zpre << ASH);
                const uint32_t d1 = ob2 + ((lane - zpre) << ASH);
                // the ballot mask itself is the select condition (v_cndmask with an SGPR mask)
                const uint32_t addr = __builtin_amdgcn_inverse_ballot_w64(om) ? d1 : d0;
                *reinterpret_cast<LdsAT*>((uintptr_t)addr) = (AT)av[e];
                const uint32_t no2 = (uint32_t)__popcll(om) * AS;
                zb2 += 64u * AS - no2;
                ob2 += no2;
            });
            lds_barrier();
        }
        store_cols(bt + 1u);
        store_info(bt + 2u);
        Rline = Rline_nxt;
        lds_barrier();
    }
}

// Global-memory variant for N > 65536 (uint32 prefix array, ping-pong buffers in HBM/L2, bit
// column still in LDS).  Same algorithm, two passes per line because a wave's positions no
// longer fit in registers.  One workgroup per block.  Correctness path for UKB-scale N; the
// multi-CU cooperative version is future work.
template <bool DECODE>
__global__ void __launch_bounds__(1024) k_chain_global(const EncBlock* __restrict__ eblocks,
                                                       const DecBlock* __restrict__ dblocks, ChainArgs A,
                                                       uint32_t* __restrict__ scratch_a) {
    constexpr uint32_t T = 1024, W = 16;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t N = A.N, cw = A.cw;
    uint32_t* col = reinterpret_cast<uint32_t*>(smem);  // cw words: key column (encode: x, decode: y)
    uint32_t* xrow = col + cw;                          // cw words
    uint32_t* wcnt = xrow + cw;                         // 2*W
    const uint32_t tid = threadIdx.x, lane = tid & 63u, w = tid >> 6;
    uint32_t wah_first, n_wah;
    if (DECODE) {
        if (dblocks[blockIdx.x].error) return;
        if (A.only_haploid_blocks && dblocks[blockIdx.x].off_line_haploid == VAL_UNDEFINED) return;
        wah_first = dblocks[blockIdx.x].wah_first;
        n_wah = dblocks[blockIdx.x].n_wah;
    } else {
        if (A.only_haploid_blocks && !eblocks[blockIdx.x].has_haploid) return;
        wah_first = eblocks[blockIdx.x].wah_first;
        n_wah = eblocks[blockIdx.x].n_wah;
    }
    if (n_wah == 0) return;
    const size_t na = ((size_t)N + 63u) & ~(size_t)63u;
    uint32_t* a0 = scratch_a + (size_t)blockIdx.x * 2u * na;
    uint32_t* a1 = a0 + na;
    for (uint32_t i = tid; i < N; i += T) a0[i] = i;
    for (uint32_t i = tid; i < cw; i += T) xrow[i] = 0;
    const uint32_t src_words = (N + 31u) >> 5;
    const uint32_t nchunks = (N + 63u) >> 6;
    const uint32_t cpw = (nchunks + W - 1u) / W;  // chunks per wave
    __threadfence_block();
    __syncthreads();
    for (uint32_t j = 0; j < n_wah; ++j) {
        const uint32_t rank = wah_first + j;
        const uint32_t line = A.wah_lines[rank];
        const bool hap = A.kind && (A.kind[line] & KIND_HAPLOID);
        const uint32_t* srow = A.src + (DECODE ? (size_t)rank : (size_t)line) * A.src_stride_w;
        for (uint32_t i = tid; i < cw; i += T) col[i] = i < src_words ? srow[(size_t)i << A.src_elem_shift] : 0u;
        __syncthreads();
        const uint32_t* ain = (j & 1u) ? a1 : a0;
        uint32_t* aout = (j & 1u) ? a0 : a1;
        const uint32_t c_lo = w * cpw, c_hi = (c_lo + cpw < nchunks) ? c_lo + cpw : nchunks;
        if (hap) {
            // haploid line: build the sample-indexed key column first (decode) / y bits (encode)
            uint32_t* ev = wcnt + W;
            uint32_t ec = 0;
            for (uint32_t cg = c_lo; cg < c_hi; ++cg) {
                const uint32_t idx = cg * 64u + lane;
                const uint32_t v = idx < N ? ain[idx] : 1u;
                ec += (uint32_t)__popcll(__ballot(!(v & 1u)));
            }
            if (lane == 0) ev[w] = ec;
            __syncthreads();
            uint32_t eb = 0;
            for (uint32_t i = 0; i < w; ++i) eb += ev[i];
            for (uint32_t cg = c_lo; cg < c_hi; ++cg) {
                const uint32_t idx = cg * 64u + lane;
                const uint32_t v = idx < N ? ain[idx] : 1u;
                const bool even = !(v & 1u);
                const uint64_t evm = __ballot(even);
                const uint32_t p = eb + mbcnt64(evm);
                if (even) {
                    if (DECODE) {
                        if ((col[p >> 5] >> (p & 31u)) & 1u) atomicOr(&xrow[(v >> 1) >> 5], 1u << ((v >> 1) & 31u));
                    } else {
                        if ((col[(v >> 1) >> 5] >> ((v >> 1) & 31u)) & 1u) atomicOr(&xrow[p >> 5], 1u << (p & 31u));
                    }
                }
                eb += (uint32_t)__popcll(evm);
            }
            __syncthreads();
        }
        const uint32_t* keycol = (hap && DECODE) ? xrow : col;
        // pass 1: zeros per wave
        uint32_t zc = 0;
        for (uint32_t cg = c_lo; cg < c_hi; ++cg) {
            const uint32_t base = cg * 64u, idx = base + lane;
            const bool valid = idx < N;
            uint32_t bit;
            if (DECODE && !hap) {
                bit = valid ? ((keycol[idx >> 5] >> (idx & 31u)) & 1u) : 0u;
            } else {
                const uint32_t v = valid ? ain[idx] : 0u;
                const uint32_t k = hap ? (v >> 1) : v;
                bit = valid ? ((keycol[k >> 5] >> (k & 31u)) & 1u) : 0u;
            }
            const uint64_t vm = (N - base >= 64u) ? ~0ull : ((1ull << (N - base)) - 1ull);
            zc += (uint32_t)__popcll(~__ballot(bit) & vm);
        }
        if (lane == 0) wcnt[w] = zc;
        __syncthreads();
        uint32_t zb = 0, tz = 0;
        for (uint32_t i = 0; i < W; ++i) {
            const uint32_t cz = wcnt[i];
            tz += cz;
            if (i < w) zb += cz;
        }
        const uint32_t before = (c_lo * 64u < N) ? c_lo * 64u : N;
        uint32_t ob = before - zb;
        // pass 2: scatter (+ y emission / x scatter)
        uint64_t* yr = reinterpret_cast<uint64_t*>(A.dst + (size_t)rank * A.dst_stride_w);
        for (uint32_t cg = c_lo; cg < c_hi; ++cg) {
            const uint32_t base = cg * 64u, idx = base + lane;
            const bool valid = idx < N;
            const uint32_t v = valid ? ain[idx] : 0u;
            uint32_t bit;
            if (DECODE && !hap) {
                bit = valid ? ((keycol[idx >> 5] >> (idx & 31u)) & 1u) : 0u;
                if (bit) atomicOr(&xrow[v >> 5], 1u << (v & 31u));
            } else {
                const uint32_t k = hap ? (v >> 1) : v;
                bit = valid ? ((keycol[k >> 5] >> (k & 31u)) & 1u) : 0u;
            }
            const uint64_t vm = (N - base >= 64u) ? ~0ull : ((1ull << (N - base)) - 1ull);
            const uint64_t om = __ballot(bit), zm = ~om & vm;
            if (!DECODE && !hap && lane == 0) yr[cg] = om;
            const uint32_t dest = bit ? tz + ob + mbcnt64(om) : zb + mbcnt64(zm);
            if (valid) aout[dest] = v;
            zb += (uint32_t)__popcll(zm);
            ob += (uint32_t)__popcll(om);
        }
        __threadfence_block();
        __syncthreads();
        if (DECODE || hap) {
            uint32_t* orow = DECODE ? A.dst + (size_t)(line - A.out_row_base) * A.dst_stride_w
                                    : A.dst + (size_t)rank * A.dst_stride_w;
            for (uint32_t i = tid; i < A.dst_stride_w; i += T) {
                uint32_t v = 0;
                if (i < cw) {
                    v = xrow[i];
                    xrow[i] = 0;
                }
                orow[i] = v;
            }
            __syncthreads();
        }
    }
}

// Streaming variant of the encode chain for N > 65536 (blocks without fully haploid lines).
// Still one workgroup per block with `a` ping-ponged in HBM/L2, but `a` is read ONCE per line:
//   * wave w owns the positions of segment w (segments are 2^seg_shift positions);
//   * the per-segment zero counts a line needs before it can scatter are accumulated one line
//     ahead: while line t moves member v to position d of a_t, the wave also looks up v's bit on
//     line t+1 and adds its zero to the counter of d's segment (both bit rows sit in LDS);
//   * U chunks of 64 positions are loaded before the first is used, so a wave keeps U global
//     loads in flight instead of one.
// Per line: one pass over `a` (4 B read + 4 B written per member), two barriers.
constexpr int STREAM_U = 8;
constexpr int STREAM_ROW_REGS = 20;  // row words per thread: N <= 1024*20*32

// AT: element type of `a` in HBM/L2: uint16_t while N <= 65536 (the array traffic, 2 x sizeof(AT) per
// member per line, is what bounds this kernel once a few hundred blocks are in flight), else uint32_t.
template <typename AT>
__global__ void __launch_bounds__(1024) k_chain_stream(const EncBlock* __restrict__ eblocks, ChainArgs A,
                                                       uint32_t* __restrict__ scratch_a, uint32_t SS) {
    constexpr uint32_t T = 1024, W = 16, U = STREAM_U;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t N = A.N, cw = A.cw;
    uint32_t* rows = reinterpret_cast<uint32_t*>(smem);  // [2][cw]
    uint32_t* zc = rows + 2u * cw;                       // [3][W] zero counts per segment, rotating
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const uint32_t w = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6));
    if (eblocks[blockIdx.x].has_haploid) return;  // k_chain_global handles those blocks
    const uint32_t wah_first = eblocks[blockIdx.x].wah_first;
    const uint32_t n_wah = eblocks[blockIdx.x].n_wah;
    if (n_wah == 0) return;
    const size_t na = ((size_t)N + 63u) & ~(size_t)63u;
    AT* a0 = reinterpret_cast<AT*>(scratch_a + (size_t)blockIdx.x * 2u * na);  // region sized for uint32
    AT* a1 = a0 + na;
    const uint32_t src_words = (N + 31u) >> 5;
    const uint32_t tail_mask = (N & 31u) ? ((1u << (N & 31u)) - 1u) : ~0u;
    auto load_word = [&](const uint32_t* srow, uint32_t i) -> uint32_t {
        const uint32_t v = srow[i < src_words ? i : 0u];
        return i >= src_words ? 0u : (i == src_words - 1u ? (v & tail_mask) : v);
    };
    for (uint32_t i = tid; i < N; i += T) a0[i] = (AT)i;
    {
        const uint32_t* r0 = A.src + (size_t)A.wah_lines[wah_first] * A.src_stride_w;
        for (uint32_t i = tid; i < cw; i += T) rows[i] = load_word(r0, i);
        if (n_wah > 1) {
            const uint32_t* r1 = A.src + (size_t)A.wah_lines[wah_first + 1u] * A.src_stride_w;
            for (uint32_t i = tid; i < cw; i += T) rows[cw + i] = load_word(r1, i);
        }
    }
    if (tid < 3u * W) zc[tid] = 0;
    // wave w owns positions [w*SS, (w+1)*SS) (SS is a multiple of 64)
    const uint32_t p_lo = w * SS < N ? w * SS : N;
    const uint32_t p_hi = (w + 1u) * SS < N ? (w + 1u) * SS : N;
    const uint32_t n_full = (p_hi - p_lo) / (64u * U);  // groups of U full chunks
    __threadfence_block();
    __syncthreads();
    {
        // a_0 is the identity: zeros of the first line per segment straight from its row
        uint32_t ones = 0;
        const uint32_t w_hi = p_hi > p_lo ? (p_hi + 31u) >> 5 : 0u;  // empty segment: nothing to count
        for (uint32_t i = (p_lo >> 5) + lane; i < w_hi; i += 64u) ones += (uint32_t)__popc(rows[i]);
        ones = wave_sum(ones);
        if (lane == 0) zc[w] = (p_hi - p_lo) - ones;
    }
    for (uint32_t j = 0; j < n_wah; ++j) {
        const uint32_t rank = wah_first + j;
        const unsigned char* rcur = reinterpret_cast<const unsigned char*>(rows + (j & 1u) * cw);
        // last line: no look-ahead needed; point it at the current row so the code stays branch-free
        const unsigned char* rnxt =
            reinterpret_cast<const unsigned char*>(rows + ((j + 1u < n_wah ? j + 1u : j) & 1u) * cw);
        const bool has_next2 = j + 2u < n_wah;
        uint32_t* zcur = zc + (j % 3u) * W;
        uint32_t* znxt = zc + ((j + 1u) % 3u) * W;
        uint32_t* zclr = zc + ((j + 2u) % 3u) * W;
        __syncthreads();  // rows, zcur and a_in complete
        // row of line j+2 -> registers now, -> LDS after this line (its buffer is still in use)
        uint32_t pre[STREAM_ROW_REGS];
        if (has_next2) {
            const uint32_t* r2 = A.src + (size_t)A.wah_lines[rank + 2u] * A.src_stride_w;
#pragma unroll
            for (int k = 0; k < STREAM_ROW_REGS; ++k) pre[k] = load_word(r2, (uint32_t)k * T + tid);
        }
        if (tid < W) zclr[tid] = 0;
        uint32_t zb = 0, tz = 0;
        {
            const uint32_t c = lane < W ? zcur[lane] : 0u;
            const uint32_t sc = row16_scan_incl(c);
            tz = (uint32_t)__builtin_amdgcn_readlane((int)sc, W - 1);
            zb = w ? (uint32_t)__builtin_amdgcn_readlane((int)sc, (int)w - 1) : 0u;
        }
        uint32_t ob = tz + (p_lo - zb);  // destination of my segment's first one
        const AT* ain = (j & 1u) ? a1 : a0;
        AT* aout = (j & 1u) ? a0 : a1;
        uint2* yr = reinterpret_cast<uint2*>(A.dst + (size_t)rank * A.dst_stride_w);
        // look-ahead accumulators: zeros of line j+1 among the members I move, per destination segment
        uint32_t zseg = zb / SS, zacc = 0, zbound = (zseg + 1u) * SS;
        uint32_t oseg = ob / SS, oacc = 0, obound = (oseg + 1u) * SS;

        // one chunk of 64 positions starting at cb whose members are vv; returns the chunk's y bits
        auto step = [&](auto full_tag, uint32_t cb, uint32_t vv) -> uint64_t {
            constexpr bool FULL = decltype(full_tag)::value;
            const bool valid = FULL || (cb + lane < p_hi);  // valid lanes are a prefix of the wave
            const uint32_t woff = (vv >> 3) & ~3u;          // byte offset of the word holding bit vv
            uint32_t bit = __builtin_amdgcn_ubfe(*reinterpret_cast<const uint32_t*>(rcur + woff), vv, 1u);
            uint32_t nbit = __builtin_amdgcn_ubfe(*reinterpret_cast<const uint32_t*>(rnxt + woff), vv, 1u);
            if (!FULL && !valid) {
                bit = 0;
                nbit = 1;
            }
            const uint64_t vm = FULL ? ~0ull : __ballot(valid);
            const uint64_t om = __ballot(bit != 0u), zm = ~om & vm;
            const uint64_t nz = __ballot(nbit == 0u);  // members whose bit on the next line is 0
            const uint32_t zpre = mbcnt64(zm);
            const uint32_t dest = bit ? ob + (lane - zpre) : zb + zpre;
            if (FULL || valid) aout[dest] = (AT)vv;
            const uint32_t nzc = (uint32_t)__popcll(zm), noc = (uint32_t)__popcll(om);
            if (zb + nzc > zbound) {  // the zeros of this chunk cross into the next segment (rare)
                const uint64_t hi = __ballot(valid && !bit && dest >= zbound);
                zacc += (uint32_t)__popcll(nz & zm & ~hi);
                if (lane == 0 && zacc) atomicAdd(&znxt[zseg], zacc);
                ++zseg;
                zbound += SS;
                zacc = (uint32_t)__popcll(nz & hi);
            } else {
                zacc += (uint32_t)__popcll(nz & zm);
            }
            if (ob + noc > obound) {
                const uint64_t hi = __ballot(valid && bit && dest >= obound);
                oacc += (uint32_t)__popcll(nz & om & ~hi);
                if (lane == 0 && oacc) atomicAdd(&znxt[oseg], oacc);
                ++oseg;
                obound += SS;
                oacc = (uint32_t)__popcll(nz & hi);
            } else {
                oacc += (uint32_t)__popcll(nz & om);
            }
            zb += nzc;
            ob += noc;
            return om;
        };

        uint32_t base = p_lo;
        uint32_t vn[U];
        if (n_full) {
#pragma unroll
            for (uint32_t u = 0; u < U; ++u) vn[u] = (uint32_t)ain[base + u * 64u + lane];
        }
        for (uint32_t g = 0; g < n_full; ++g, base += 64u * U) {
            uint32_t v[U];
#pragma unroll
            for (uint32_t u = 0; u < U; ++u) v[u] = vn[u];
            if (g + 1u < n_full) {
#pragma unroll
                for (uint32_t u = 0; u < U; ++u) vn[u] = (uint32_t)ain[base + (U + u) * 64u + lane];
            }
            uint32_t mine_lo = 0, mine_hi = 0;
#pragma unroll
            for (uint32_t u = 0; u < U; ++u) {
                const uint64_t om = step(std::true_type{}, base + u * 64u, v[u]);
                mine_lo = write_lane(mine_lo, (uint32_t)om, u);
                mine_hi = write_lane(mine_hi, (uint32_t)(om >> 32), u);
            }
            if (lane < U) yr[(base >> 6) + lane] = make_uint2(mine_lo, mine_hi);
        }
        for (; base < p_hi; base += 64u) {
            const uint32_t idx = base + lane;
            const uint32_t vv = idx < p_hi ? (uint32_t)ain[idx] : 0u;
            const uint64_t om = step(std::false_type{}, base, vv);
            if (lane == 0) yr[base >> 6] = make_uint2((uint32_t)om, (uint32_t)(om >> 32));
        }
        if (lane == 0) {
            if (zacc) atomicAdd(&znxt[zseg], zacc);
            if (oacc) atomicAdd(&znxt[oseg], oacc);
        }
        __threadfence_block();
        __syncthreads();  // everyone is done with rcur
        if (has_next2) {
            uint32_t* rdst = rows + (j & 1u) * cw;
#pragma unroll
            for (int k = 0; k < STREAM_ROW_REGS; ++k) {
                const uint32_t i = (uint32_t)k * T + tid;
                if (i < cw) rdst[i] = pre[k];
            }
        }
    }
}

static uint32_t next_pow2_log2(uint32_t v) {
    uint32_t l = 0;
    while ((1u << l) < v) ++l;
    return l;
}

// E values the LDS chain kernel is instantiated for (T = 1024); N <= 1024*E
static const int k_chain_E[] = {1, 2, 3, 4, 5, 6, 7, 8, 10, 12, 16, 20, 24, 32};

ChainGeom chain_geometry(uint32_t N, bool decode) {
    ChainGeom g{};
    // The position-major kernels (needed for small N and for blocks with fully haploid lines; everything else
    // is rank tracking, xsi_rankenc.hip / xsi_rank.hip): prefix array in LDS up to 32768 haplotypes (32 chunks
    // per wave: the larger instantiations spilled), beyond that in HBM/L2 (k_chain_stream / k_chain_global).
    (void)decode;
    g.in_lds = N <= 32768u;
    if (!g.in_lds) {
        const uint32_t cw = (((N + 31u) >> 5) + 1u) & ~1u;
        g.threads = 1024;
        g.chunks = 0;
        g.batch = 1;
        g.lds_bytes = (2u * cw + 64u) * 4u;
        return g;
    }
    int T, E;
    if (N <= 256) {
        T = 256;
        E = 1;
    } else if (N <= 1024) {
        T = 256;
        E = 4;
    } else {
        T = 1024;
        E = 64;
        for (int e : k_chain_E)
            if ((uint32_t)(T * e) >= N) {
                E = e;
                break;
            }
    }
    const uint32_t NA = (uint32_t)T * E;
    const uint32_t cw = NA / 32u;
    const uint32_t cwp = 1u << next_pow2_log2(cw);
    uint32_t B = (uint32_t)(CHAIN_RMAX * T) / cwp;
    if (B > 16u) B = 16u;
    if (B < 1u) B = 1u;
    const uint32_t lds_max = 160u * 1024u;
    const uint32_t asz = NA <= 32768u ? 4u : 2u;  // element size of `a` (see k_chain_lds)
    auto need = [&](uint32_t b) { return NA * asz + (2u * b * cw + cw + 2u * 16u + 48u) * 4u; };
    while (B > 1u && need(B) > lds_max) B >>= 1;
    g.threads = T;
    g.chunks = E;
    g.batch = B;
    g.lds_bytes = need(B);
    return g;
}

template <bool DECODE>
static hipError_t launch_chain(hipStream_t s, const EncBlock* eb, const DecBlock* db, uint32_t n_blocks,
                               ChainArgs A, uint32_t* scratch_a) {
    if (!n_blocks) return hipSuccess;
    const ChainGeom g = chain_geometry(A.N, DECODE);
    A.cw = g.in_lds ? (uint32_t)(g.threads * g.chunks) / 32u : ((((A.N + 31u) >> 5) + 1u) & ~1u);
    A.log2_cwp = next_pow2_log2(A.cw);
    A.batch = g.batch;
    if (!g.in_lds) {
        if (!scratch_a) return hipErrorInvalidValue;
        // encode: A.only_haploid_blocks arrives as "some block has fully haploid lines"
        const bool any_haploid = A.only_haploid_blocks != 0;
        if (!DECODE) A.only_haploid_blocks = A.plain_done;  // rank tracking done: only the haploid blocks are left
        if (!DECODE && !A.plain_done && A.cw <= 1024u * STREAM_ROW_REGS && !tuning_env("XSI_NO_STREAM_CHAIN")) {
            // streaming kernel for the blocks without fully haploid lines, the two-pass kernel for the rest
            const uint32_t seg = (((A.N + 15u) / 16u) + 63u) & ~63u;  // positions per wave
            const uint32_t lds = (2u * A.cw + 3u * 16u) * 4u;
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_chain_stream<uint32_t>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return e;
            k_chain_stream<uint32_t><<<dim3(n_blocks), dim3(1024), lds, s>>>(eb, A, scratch_a, seg);
            e = hipGetLastError();
            if (e != hipSuccess || !any_haploid) return e;
            A.only_haploid_blocks = 1;
        }
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_chain_global<DECODE>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)g.lds_bytes);
        if (e != hipSuccess) return e;
        k_chain_global<DECODE><<<dim3(n_blocks), dim3(1024), g.lds_bytes, s>>>(eb, db, A, scratch_a);
        return hipGetLastError();
    }
    // Encode with fewer blocks than CUs: cut every block's chain into line segments (chain_prepass).
    // A later segment first pays rho x (lines before it) for the radix pre-pass, so segments shrink:
    // len_s = len_0 - rho * start_s, all segments finishing together.
    uint32_t segs = 1;
    uint32_t lds_bytes = g.lds_bytes;
    A.segments = 1;
    if (!DECODE && g.chunks <= 16 && !A.only_haploid_blocks) {
        static const int env_s = [] {
            const char* e = tuning_env("XSI_CHAIN_SEGMENTS");
            return e ? atoi(e) : 0;
        }();
        static const double rho = [] {
            const char* e = tuning_env("XSI_CHAIN_RHO");
            return e ? atof(e) : 0.55;  // measured best at the bench size (0.25 ... 0.65 swept)
        }();
        uint32_t S = 256u / n_blocks;
        if (S > 4u) S = 4u;
        if (env_s >= 1 && env_s <= 4) S = (uint32_t)env_s;
        if (S > 1u) {
            double lo = 0.0, hi = 1.0, bound[5] = {0, 0, 0, 0, 0};
            for (int it = 0; it < 50; ++it) {
                const double l0 = 0.5 * (lo + hi);
                double start = 0.0;
                for (uint32_t k = 0; k < S; ++k) {
                    double len = l0 - rho * start;
                    if (len < 0.0) len = 0.0;
                    start += len;
                    bound[k + 1] = start;
                }
                if (start < 1.0) lo = l0; else hi = l0;
            }
            for (uint32_t k = 0; k <= 4; ++k) {
                double b = k < S ? bound[k] : 1.0;
                if (b > 1.0) b = 1.0;
                A.seg_q16[k] = (uint32_t)(b * 65536.0);
            }
            A.seg_q16[0] = 0;
            segs = S;
            A.segments = S;
            lds_bytes += (PRE_ID_CAP + (uint32_t)(g.threads * g.chunks) / 8u + 2u * 16u * 16u) * 4u;
        }
    }
#define XSI_CHAIN_CASE(TT, EE)                                                                              \
    if (g.threads == TT && g.chunks == EE) {                                                                \
        using AT = std::conditional_t<((TT) * (EE) <= 32768), uint32_t, uint16_t>;                          \
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_chain_lds<TT, EE, DECODE, AT>), \
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);     \
        if (e != hipSuccess) return e;                                                                      \
        k_chain_lds<TT, EE, DECODE, AT><<<dim3(n_blocks * segs), dim3(TT), lds_bytes, s>>>(eb, db, A);       \
        return hipGetLastError();                                                                           \
    }
    XSI_CHAIN_CASE(256, 1)
    XSI_CHAIN_CASE(256, 4)
    XSI_CHAIN_CASE(1024, 2)
    XSI_CHAIN_CASE(1024, 3)
    XSI_CHAIN_CASE(1024, 4)
    XSI_CHAIN_CASE(1024, 5)
    XSI_CHAIN_CASE(1024, 6)
    XSI_CHAIN_CASE(1024, 7)
    XSI_CHAIN_CASE(1024, 8)
    XSI_CHAIN_CASE(1024, 10)
    XSI_CHAIN_CASE(1024, 12)
    XSI_CHAIN_CASE(1024, 16)
    XSI_CHAIN_CASE(1024, 20)
    XSI_CHAIN_CASE(1024, 24)
    XSI_CHAIN_CASE(1024, 32)
#undef XSI_CHAIN_CASE
    return hipErrorInvalidValue;
}

// Which kernel takes the blocks without fully haploid lines (measured on MI355X, profiles/r02_chain_sweep.txt):
//   encode  rank tracking (xsi_rankenc.hip) from ~20k haplotypes on, and from ~12k when there are enough
//           blocks that k_chain_lds could not cut them into line segments anyway; k_chain_lds below that;
//   decode  always the element-major kernels of xsi_rank.hip (launch_rank_decode picks the geometry).
// XSI_RANKENC_MIN_N overrides the size rule (testing: force the kernel for every N); read per call.
static bool use_rank_encode(uint32_t N, uint32_t n_blocks) {
    const char* ev = tuning_env("XSI_RANKENC_MIN_N");
    const int env = ev ? atoi(ev) : -1;
    if (!chain_rank_enc_supported(N)) return false;
    if (env >= 0) return N >= (uint32_t)env;
    return N >= 20480u || (N >= 12288u && n_blocks >= 192u);
}

const char* chain_kernel_name(uint32_t N, uint32_t n_blocks, bool decode) {
    if (decode) return rank_decode_kernel_name(N, ((N + 63u) / 64u) * 2u, n_blocks);
    if (use_rank_encode(N, n_blocks)) return "k_chain_rank_enc";
    if (N > 65536u && N <= 524288u && !tuning_env("XSI_NO_RANKENC_MULTI")) return "k_chain_rank_enc_multi";
    return N <= 65536u ? "k_chain_lds" : "k_chain_stream";
}

hipError_t launch_chain_encode(hipStream_t s, const EncBlock* blocks, uint32_t n_blocks, const EncLines& L,
                               uint32_t* scratch_a, bool any_haploid, bool* multi_refused) {
    bool rank_done = false;
    if (multi_refused) *multi_refused = false;
    if (chain_rank_enc_multi_supported(L)) {
        bool refused = false;
        hipError_t e = launch_rank_encode_multi(s, blocks, n_blocks, L, &refused);
        if (refused) {
            // the device (or the CU mask of this process) cannot hold a whole group of workgroups:
            // k_chain_stream below takes the blocks (the caller counts it as a fallback).  Only the launcher's own
            // pre-launch checks say so; a hipErrorInvalidValue from the runtime is an error like any other
            if (multi_refused) *multi_refused = true;
        } else {
            if (e != hipSuccess || !any_haploid) return e;
            rank_done = true;  // k_chain_global below only takes the blocks with fully haploid lines
        }
    } else if (use_rank_encode(L.N, n_blocks)) {
        hipError_t e = launch_rank_encode(s, blocks, n_blocks, L);
        if (e != hipSuccess || !any_haploid) return e;
        rank_done = true;  // k_chain_lds below only takes the blocks with fully haploid lines
    }
    ChainArgs A{};
    // LDS kernel: handles haploid lines itself.  N > 65536: the streaming kernel takes the blocks
    // without haploid lines, k_chain_global the others (only_haploid_blocks makes it skip the rest).
    A.only_haploid_blocks = (rank_done || (any_haploid && !chain_geometry(L.N, false).in_lds)) ? 1u : 0u;
    A.plain_done = rank_done ? 1u : 0u;
    A.wah_lines = L.wah_lines;
    A.kind = L.kind;
    A.src = L.planes;
    A.src_stride_w = L.plane_stride_w;
    A.dst = reinterpret_cast<uint32_t*>(L.yrows);
    A.dst_stride_w = L.y_stride64 * 2u;
    A.N = L.N;
    A.out_row_base = 0;
    return launch_chain<false>(s, blocks, nullptr, n_blocks, A, scratch_a);
}

hipError_t launch_chain_decode(hipStream_t s, const DecBlock* blocks, uint32_t n_blocks, const DecLines& L,
                               uint32_t* out_rows, uint32_t out_stride_w, uint32_t* scratch_a, bool any_haploid) {
    if (!n_blocks) return hipSuccess;
    // blocks without fully haploid lines: element-major kernels (xsi_rank.hip)
    hipError_t e = launch_rank_decode(s, blocks, n_blocks, L, out_rows, out_stride_w);
    if (e != hipSuccess || !any_haploid) return e;
    // position-major kernel: the blocks with fully haploid lines (it skips the others)
    ChainArgs A{};
    A.wah_lines = L.wah_lines;
    A.kind = L.kind;
    A.src = reinterpret_cast<const uint32_t*>(L.yp);
    A.src_stride_w = L.yp_stride * 2u;
    A.src_elem_shift = 1;
    A.dst = out_rows;
    A.dst_stride_w = out_stride_w;
    A.N = L.N;
    A.out_row_base = 0;
    A.only_haploid_blocks = 1;
    return launch_chain<true>(s, nullptr, blocks, n_blocks, A, scratch_a);
}

// ------------------------------------------------------------------------------------------
// WAH16 sizing and writing of the permuted rows (wah_encode2_with_size, wah.hpp:506-578).
// One wave per WAH line; the sizing pass and the writing pass run the same encoder.
// ------------------------------------------------------------------------------------------
constexpr uint32_t WAH_LINES_PER_WAVE = 4;  // amortises the dependent metadata loads of a line
// the words of a line (n of them) may be kept in its own row of y_stride64 8-byte pairs: 4 bytes to spare, because
// k_wah_write's re-pairing copy reads one 4-byte word past the last
__device__ __forceinline__ bool wah_fits_row(uint32_t n, uint32_t y_stride64) { return 2u * n + 4u <= 8u * y_stride64; }

__global__ void __launch_bounds__(256) k_wah_sizes(EncLines L, const uint32_t* __restrict__ d_total_wah) {
    const uint32_t lane = lane_id();
    const uint32_t j0 = (blockIdx.x * 4u + (threadIdx.x >> 6)) * WAH_LINES_PER_WAVE;
    const uint32_t total = d_total_wah[0];
    if (j0 >= total) return;
    // lane k fetches the metadata of line j0 + k; the lines are then encoded one after the other
    uint32_t m_nbits = 0;
    if (lane < WAH_LINES_PER_WAVE && j0 + lane < total) m_nbits = nbits_of(L, L.wah_lines[j0 + lane]);
    for (uint32_t k = 0; k < WAH_LINES_PER_WAVE && j0 + k < total; ++k) {
        const uint32_t j = j0 + k;
        const uint32_t nbits = (uint32_t)__builtin_amdgcn_readlane((int)m_nbits, (int)k);
        const uint32_t* row = reinterpret_cast<const uint32_t*>(L.yrows + (size_t)j * L.y_stride64);
        uint32_t n;
        if (L.wah_scratch)
            n = wave_wah_encode_row<true>(row, nbits, L.wah_scratch + (size_t)j * L.wah_scratch_stride);
        else
            n = wave_wah_encode_row<false>(row, nbits, nullptr);
        if (lane == 0) L.wah_len[j] = n;
    }
}

// Short rows (at most 32 units of 480 bits: up to 15 360 haplotypes): the unit encoder of xsi_device.hpp with SEVERAL
// lines per wave.  A line has upl = ceil(groups / 32) units; a wave takes 64 / upl lines at once, lane = (line, unit).
// The rows sit back to back in LDS, upl * 15 words each, so lane L reads from word 15 L: no bank conflicts.  Heads are
// counted per line with a segmented wave scan; with a scratch row per line the words are emitted at once (the serial
// encoder this replaces walks a line 64 groups at a time with a carried run state, about 150 instructions per step of a
// dependent chain: 0.61 ms at 5008 haplotypes x 1 M sites).
constexpr uint32_t WAH_SMALL_ROW_WORDS = 64u * 15u + 16u;  // rows of one wave + room for the one-word over-read of a literal
template <bool WRITE>
__global__ void __launch_bounds__(256) k_wah_units_small(EncLines L, const uint32_t* __restrict__ d_total_wah, uint32_t upl) {
    __shared__ uint32_t s_rows[4][WAH_SMALL_ROW_WORDS];
    __shared__ uint32_t s_fh[4][64];
    const uint32_t lane = lane_id(), wv = threadIdx.x >> 6;
    const uint32_t lpw = 64u / upl;  // lines per wave
    const uint32_t j0 = (blockIdx.x * 4u + wv) * lpw;
    const uint32_t total = d_total_wah[0];
    if (j0 >= total) return;
    using LdsRowW = __attribute__((address_space(3))) uint32_t;
    LdsRowW* rows = reinterpret_cast<LdsRowW*>((__attribute__((address_space(3))) unsigned char*)&s_rows[wv][0]);
    LdsRowW* fh = reinterpret_cast<LdsRowW*>((__attribute__((address_space(3))) unsigned char*)&s_fh[wv][0]);
    const uint32_t li = lane / upl, u = lane - li * upl;  // my line of the wave, my unit of the line
    const bool mine = li < lpw && j0 + li < total;
    const uint32_t j = mine ? j0 + li : j0;
    const uint32_t nbits = mine ? nbits_of(L, L.wah_lines[j]) : 0u;
    // stage: unit `u` of line `li` = 15 words; bits at or beyond nbits read as zero (wah.hpp:547-565)
    {
        const uint32_t* src = reinterpret_cast<const uint32_t*>(L.yrows + (size_t)j * L.y_stride64);
        const uint32_t nw = (nbits + 31u) >> 5;
        uint32_t w[15];
#pragma unroll
        for (uint32_t i = 0; i < 15u; ++i) {
            const uint32_t wi = u * 15u + i;
            w[i] = src[(mine && wi < nw) ? wi : 0u];  // unconditional loads, all in flight
        }
#pragma unroll
        for (uint32_t i = 0; i < 15u; ++i) {
            const uint32_t wi = u * 15u + i;
            uint32_t v = (mine && wi < nw) ? w[i] : 0u;
            if (wi + 1u == nw && (nbits & 31u)) v &= (1u << (nbits & 31u)) - 1u;
            rows[lane * 15u + i] = v;
        }
        if (lane < 16u) rows[64u * 15u + lane] = 0;
    }
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");  // other lanes' LDS stores are read below (one wave: its LDS operations stay in order)
    const uint32_t G = (nbits + WAH_BITS - 1u) / WAH_BITS;
    WahUnit m{0u, 0u, 0u};
    LdsCU32* row = reinterpret_cast<LdsCU32*>(rows + li * upl * 15u);  // my line's row
    if (mine) wah_unit_classify(row, u, G, m, u);
    const uint32_t cnt = (uint32_t)__popc(m.H);
    const uint32_t inc = wave_scan_incl_dpp(cnt);
    // heads before my line = inclusive count of the last lane of the line before
    const uint32_t first_lane = li * upl, last_lane = first_lane + upl - 1u;
    // (the cross-lane read outside any condition: under a lane-dependent branch the lanes it reads from may be switched off)
    const uint32_t inc_prev = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((first_lane ? first_lane - 1u : 0u) << 2), (int)inc);
    const uint32_t before_line = first_lane ? inc_prev : 0u;
    const uint32_t line_total = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((last_lane < 64u ? last_lane : 63u) << 2), (int)inc) - before_line;
    if (mine && u == 0u) L.wah_len[j] = line_total;
    if (!WRITE) return;
    // emission into the line's scratch row: every unit its own heads (few units per line: the trip count is small)
    fh[lane] = u * 32u + (uint32_t)__builtin_ctz(m.H | 0x80000000u);  // first head of my unit (group index in the line)
    const uint64_t B = __ballot(m.H != 0u);
    asm volatile("" ::: "memory");
    const uint64_t above = (lane == 63u) ? 0ull : (~0ull << (lane + 1u));
    const uint64_t upto = last_lane >= 63u ? ~0ull : ((1ull << (last_lane + 1u)) - 1ull);
    const uint64_t later = B & above & upto;  // units of MY line behind me that have a head
    const uint32_t fh_later = (uint32_t)fh[later ? (uint32_t)__builtin_ctzll(later) : 0u];
    const uint32_t nh = later ? fh_later : G;
    __attribute__((address_space(1))) uint16_t* dst =
        (__attribute__((address_space(1))) uint16_t*)(L.wah_scratch + (size_t)j * L.wah_scratch_stride);
    uint32_t idx = inc - cnt - before_line;
    uint32_t Hr = mine ? m.H : 0u;
    const uint32_t gb = u * 32u;
    while (__any(Hr != 0u)) {
        if (Hr) {
            const uint32_t k = (uint32_t)__builtin_ctz(Hr);
            Hr &= Hr - 1u;
            const uint32_t g = gb + k;
            const uint32_t nxt = Hr ? gb + (uint32_t)__builtin_ctz(Hr) : nh;
            const uint32_t o = g * WAH_BITS;
            const uint32_t lit = __builtin_amdgcn_alignbit(row[(o >> 5) + 1u], row[o >> 5], o & 31u) & 0x7FFFu;
            const uint32_t fill = 0x8000u | (((m.O >> k) & 1u) << 14) | (nxt - g);
            dst[idx++] = (uint16_t)(((m.F >> k) & 1u) ? fill : lit);
        }
    }
}

// rows the small-row kernel takes: at least two lines per wave, and none of the longer-row kernels applies
static uint32_t wah_units_small_upl(const EncLines& L) {
    static const bool off = tuning_env("XSI_WAH_NO_SMALL") != nullptr;
    const uint32_t G = (L.N + WAH_BITS - 1u) / WAH_BITS, upl = (G + 31u) / 32u;
    return (!off && upl >= 1u && upl <= 32u && !wah_units_any(L.y_stride64)) ? upl : 0u;
}

// Rows of at most 8 KiB (N <= 65 536): each wave stages its line in LDS with 8-byte loads, all in flight at
// once, fetches the next line into registers while it works on the current one, and runs the unit encoder of
// xsi_device.hpp on it.  The sizing pass counts heads; the writing pass (after the layout is known) classifies
// again and stores the words straight into the file image: no scratch copy of the words (10 GB at 64 976 x 2 M).
constexpr int WAH_STAGE_Q = 16;  // 16 x 64 lanes x 8 bytes = 8 KiB
constexpr uint32_t WAH_UNIT_SCRATCH_WORDS = 64u * (uint32_t)WAH_UNIT_ROUNDS + 32u;  // fh, 128-byte strip (behind a wave's row)
// LDS words of a wave's row: every unit's 15 words (+ the word behind a literal), at least the row itself
static uint32_t wah_units_row_words(uint32_t y_stride64) {
    const uint32_t units = ((y_stride64 * 64u + WAH_BITS - 1u) / WAH_BITS + 31u) / 32u;
    const uint32_t need = units * 15u + 1u > 2u * y_stride64 ? units * 15u + 1u : 2u * y_stride64;
    const uint32_t w = (need + 3u) & ~3u;
    return w < WAH_UNIT_ROW_WORDS ? w : WAH_UNIT_ROW_WORDS;
}
// MODE 0: sizing pass; 1: writing pass (classifies again, words straight into the file image); 2: sizing pass that also
// leaves the line's words IN THE LINE'S OWN ROW (the row is in LDS by then, and a line whose words would not fit its
// row - less than one line in 16 literal groups short of incompressible - is left alone): k_wah_write then moves the
// words into place and the rows are read once, not twice (10 GB at 64 976 haplotypes x 2 M sites).
template <int MODE>
__global__ void __launch_bounds__(256) k_wah_units(const EncBlock* __restrict__ blocks, EncLines L,
                                                   const uint32_t* __restrict__ d_total_wah, uint32_t max_wah,
                                                   uint8_t* __restrict__ out, const uint64_t* __restrict__ d_result,
                                                   uint32_t row_words) {
    // row_words: LDS words of a wave's row (wah_units_row_words: what the rows of this launch need, not the 2880 of the
    // longest row the kernel takes - 36 instead of 49 KB per workgroup at 64 976 haplotypes: four waves per SIMD, not three)
    constexpr bool WRITE_PASS = MODE == 1;
    extern __shared__ __attribute__((aligned(16))) unsigned char wah_smem[];
    if (WRITE_PASS && d_result[3]) return;  // capacity error: nothing may be written
    const uint32_t lane = lane_id();
    const uint32_t wv = threadIdx.x >> 6;
    const uint32_t j0 = (blockIdx.x * 4u + wv) * WAH_LINES_PER_WAVE;
    uint32_t total;
    if (WRITE_PASS)
        total = max_wah < (uint32_t)d_result[2] ? max_wah : (uint32_t)d_result[2];
    else
        total = d_total_wah[0];
    if (j0 >= total) return;
    const uint32_t np = L.y_stride64;  // 8-byte pairs per row, <= 64 WAH_STAGE_Q
    typedef uint32_t wah_u32x2 __attribute__((ext_vector_type(2)));
    using LdsU2 = __attribute__((address_space(3))) wah_u32x2;
    LdsU32W* lrow_w = reinterpret_cast<LdsU32W*>((__attribute__((address_space(3))) unsigned char*)wah_smem) +
                      (size_t)wv * (row_words + WAH_UNIT_SCRATCH_WORDS);
    LdsU2* lrow2 = reinterpret_cast<LdsU2*>(lrow_w);
    LdsCU32* lrow = reinterpret_cast<LdsCU32*>(lrow_w);
    LdsU32W* fh = lrow_w + row_words;
    for (uint32_t i = 2u * np + lane; i < row_words; i += 64u) lrow_w[i] = 0;  // beyond the row: zeros
    uint64_t m_dst = 0;
    uint32_t m_nbits = 0;
    if (lane < WAH_LINES_PER_WAVE && j0 + lane < total) {
        const uint32_t l = L.wah_lines[j0 + lane];
        m_nbits = nbits_of(L, l);
        if (WRITE_PASS) {
            const EncBlock& B = blocks[L.line_block[l]];
            m_dst = reinterpret_cast<uint64_t>(out + B.out_off + 16u + B.off_wah) + 2ull * L.wah_off[j0 + lane];
        }
    }
    uint2 R[WAH_STAGE_Q];
    auto fetch = [&](uint32_t j) {
        const uint2* src = reinterpret_cast<const uint2*>(L.yrows + (size_t)j * L.y_stride64);
#pragma unroll
        for (int q = 0; q < WAH_STAGE_Q; ++q) {
            const uint32_t idx = (uint32_t)q * 64u + lane;
            R[q] = src[idx < np ? idx : 0u];  // unconditional: with a default value the compiler waits for the load at once
        }
    };
    auto park = [&]() {
#pragma unroll
        for (int q = 0; q < WAH_STAGE_Q; ++q) {
            const uint32_t idx = (uint32_t)q * 64u + lane;
            if (idx < np) lrow2[idx] = wah_u32x2{R[q].x, R[q].y};
        }
    };
    fetch(j0);
    for (uint32_t k = 0; k < WAH_LINES_PER_WAVE && j0 + k < total; ++k) {
        const uint32_t j = j0 + k;
        park();
        if (k + 1u < WAH_LINES_PER_WAVE && j + 1u < total) fetch(j + 1u);
        const uint32_t nbits = (uint32_t)__builtin_amdgcn_readlane((int)m_nbits, (int)k);
        // bits at or beyond nbits read as zero (the reference pads the last group with zeros, wah.hpp:547-565)
        if (lane == 0 && (nbits & 31u)) lrow_w[nbits >> 5] &= (1u << (nbits & 31u)) - 1u;
        for (uint32_t i = ((nbits + 31u) >> 5) + lane; i < 2u * np; i += 64u) lrow_w[i] = 0;
        const uint32_t G = (nbits + WAH_BITS - 1u) / WAH_BITS;
        WahUnit m[WAH_UNIT_ROUNDS];
        const uint32_t n = wah_units_classify_line(lrow, G, m);
        if (WRITE_PASS) {
            const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)m_dst, (int)k);
            const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(m_dst >> 32), (int)k);
            wah_units_emit_line(lrow, fh, G, m, reinterpret_cast<uint16_t*>(((uint64_t)hi << 32) | lo));
        } else {
            if (lane == 0) L.wah_len[j] = n;
            if (MODE == 2 && wah_fits_row(n, np)) {
                const uint64_t ja = (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)j) * np;
                wah_units_emit_line(lrow, fh, G, m, reinterpret_cast<uint16_t*>(L.yrows + ja));
            }
        }
    }
}

// Rows above 8 KiB: the unit encoder with one 1024-thread workgroup per line (the serial encoder walks a line
// of 500 000 bits in 521 chunks: 28 + 38 ms for the two passes of the configs[3] shard).  Thread t owns unit
// r * 1024 + t in round r (at most two rounds: 65 536 groups); heads as in wah_unit_classify; the first head
// after a unit comes from the waves' "unit has a head" ballots kept in LDS.  Lines this long can hold a run
// that outgrows a fill word (16 383 groups): only the LAST head of a unit can start one, it then takes
// ceil(len / 16383) words (counts of 16383, then the rest - as the serial encoder does it).
constexpr int WAH_WIDE_ROUNDS = 2;
constexpr uint32_t WAH_WIDE_UNITS = 1024u * (uint32_t)WAH_WIDE_ROUNDS;
template <int MODE>  // as k_wah_units
__global__ void __launch_bounds__(1024) k_wah_units_wide(const EncBlock* __restrict__ blocks, EncLines L,
                                                         const uint32_t* __restrict__ d_total_wah, uint32_t max_wah,
                                                         uint8_t* __restrict__ out, const uint64_t* __restrict__ d_result,
                                                         uint32_t row_words_lds) {
    constexpr bool WRITE_PASS = MODE == 1;
    extern __shared__ __attribute__((aligned(16))) unsigned char wah_smem[];
    if (WRITE_PASS && d_result[3]) return;  // capacity error: nothing may be written
    const uint32_t j = blockIdx.x;
    const uint32_t total = WRITE_PASS ? (max_wah < (uint32_t)d_result[2] ? max_wah : (uint32_t)d_result[2]) : d_total_wah[0];
    if (j >= total) return;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, w = tid >> 6;
    LdsU32W* lrow_w = reinterpret_cast<LdsU32W*>((__attribute__((address_space(3))) unsigned char*)wah_smem);
    LdsCU32* lrow = reinterpret_cast<LdsCU32*>(lrow_w);
    LdsU32W* fh = lrow_w + row_words_lds;           // [WAH_WIDE_UNITS] first head of the unit (group index)
    LdsU32W* hb = fh + WAH_WIDE_UNITS;              // [2 * 16 * WAH_WIDE_ROUNDS] ballots "unit has a head", lo / hi words
    LdsU32W* sc = hb + 2u * 16u * WAH_WIDE_ROUNDS;  // [16 * WAH_WIDE_ROUNDS + 2] wave totals of the word counts
    const uint32_t l = L.wah_lines[j];
    const uint32_t nbits = nbits_of(L, l);
    const uint32_t rw = L.y_stride64 * 2u;
    {  // stage the row (16-byte loads; rows are 16-byte multiples: y_stride64 is even above 8 KiB), zeros behind it
        const uint4* src = reinterpret_cast<const uint4*>(L.yrows + (size_t)j * L.y_stride64);
        typedef uint32_t wah_u32x4 __attribute__((ext_vector_type(4)));
        using LdsU4 = __attribute__((address_space(3))) wah_u32x4;
        LdsU4* l4 = reinterpret_cast<LdsU4*>(lrow_w);
        // every piece of the row requested before the first is parked (as a loop this was load, wait, LDS write per
        // round: up to five HBM round trips in a row per line); rows of at most 5 x 1024 16-byte units (655 360 bits)
        constexpr int PIECES = (int)(WAH_WIDE_UNITS * 32u * WAH_BITS / 128u / 1024u) + 1;  // 8
        uint4 pc[PIECES];
        const uint32_t n4 = rw / 4u;
#pragma unroll
        for (int q = 0; q < PIECES; ++q) {
            const uint32_t i = (uint32_t)q * 1024u + tid;
            pc[q] = src[i < n4 ? i : 0u];  // unconditional (clamped): stays in flight
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < PIECES; ++q) {
            const uint32_t i = (uint32_t)q * 1024u + tid;
            if (i < n4) l4[i] = wah_u32x4{pc[q].x, pc[q].y, pc[q].z, pc[q].w};
        }
        for (uint32_t i = rw + tid; i < row_words_lds; i += 1024u) lrow_w[i] = 0;
    }
    __syncthreads();
    // bits at or beyond nbits read as zero (the reference pads the last group with zeros, wah.hpp:547-565)
    if (tid == 0 && (nbits & 31u)) lrow_w[nbits >> 5] &= (1u << (nbits & 31u)) - 1u;
    for (uint32_t i = ((nbits + 31u) >> 5) + tid; i < rw; i += 1024u) lrow_w[i] = 0;
    __syncthreads();
    const uint32_t G = (nbits + WAH_BITS - 1u) / WAH_BITS;
    const uint32_t units = (G + 31u) >> 5;
    WahUnit m[WAH_WIDE_ROUNDS];
#pragma unroll
    for (int r = 0; r < WAH_WIDE_ROUNDS; ++r) {
        const uint32_t u = (uint32_t)r * 1024u + tid;
        m[r] = WahUnit{0u, 0u, 0u};
        if (u < units) wah_unit_classify(lrow, u, G, m[r], u);
        const uint64_t B = __ballot(m[r].H != 0u);
        if (lane == 0) {
            hb[2u * ((uint32_t)r * 16u + w)] = (uint32_t)B;
            hb[2u * ((uint32_t)r * 16u + w) + 1u] = (uint32_t)(B >> 32);
        }
        fh[u] = u * 32u + (uint32_t)__builtin_ctz(m[r].H | 0x80000000u);
    }
    __syncthreads();
    // per unit: first head after it, words it emits (one per head; the last head's run may need more), their prefix
    uint32_t nh[WAH_WIDE_ROUNDS], nw[WAH_WIDE_ROUNDS], base[WAH_WIDE_ROUNDS];
#pragma unroll
    for (int r = 0; r < WAH_WIDE_ROUNDS; ++r) {
        const uint32_t slot = (uint32_t)r * 16u + w;  // my wave's ballot
        uint32_t tu = ~0u;
        {
            uint64_t mine = ((uint64_t)hb[2u * slot + 1u] << 32) | hb[2u * slot];
            mine = lane == 63u ? 0ull : (mine & (~0ull << (lane + 1u)));
            if (mine) tu = slot * 64u + (uint32_t)__builtin_ctzll(mine);
            for (uint32_t q = slot + 1u; tu == ~0u && q < 16u * (uint32_t)WAH_WIDE_ROUNDS; ++q) {
                const uint64_t b = ((uint64_t)hb[2u * q + 1u] << 32) | hb[2u * q];
                if (b) tu = q * 64u + (uint32_t)__builtin_ctzll(b);
            }
        }
        nh[r] = tu != ~0u ? fh[tu] : G;
        uint32_t n = (uint32_t)__popc(m[r].H);
        if (m[r].H) {
            const uint32_t k = 31u - (uint32_t)__builtin_clz(m[r].H);  // the unit's last head
            if ((m[r].F >> k) & 1u) {
                const uint32_t len = nh[r] - (((uint32_t)r * 1024u + tid) * 32u + k);
                n += (len + WAH_MAXC - 1u) / WAH_MAXC - 1u;
            }
        }
        nw[r] = n;
        const uint32_t inc = wave_scan_incl_dpp(n);
        base[r] = inc - n;
        if (lane == 63u) sc[slot] = inc;
    }
    __syncthreads();
    if (w == 0) {  // 32 wave totals, in unit order
        const uint32_t v = lane < 16u * (uint32_t)WAH_WIDE_ROUNDS ? sc[lane] : 0u;
        const uint32_t inc = wave_scan_incl_dpp(v);
        if (lane < 16u * (uint32_t)WAH_WIDE_ROUNDS) sc[lane] = inc - v;
        if (lane == 63u) sc[16u * (uint32_t)WAH_WIDE_ROUNDS] = inc;
    }
    __syncthreads();
    uint16_t* dst;
    if (!WRITE_PASS) {
        const uint32_t n_words = sc[16u * (uint32_t)WAH_WIDE_ROUNDS];
        if (tid == 0) L.wah_len[j] = n_words;
        if (MODE != 2 || !wah_fits_row(n_words, L.y_stride64)) return;
        dst = reinterpret_cast<uint16_t*>(L.yrows + (size_t)j * L.y_stride64);  // the row is in LDS: its words take its place
    } else {
        const EncBlock& Bk = blocks[L.line_block[l]];
        dst = reinterpret_cast<uint16_t*>(out + Bk.out_off + 16u + Bk.off_wah) + L.wah_off[j];
    }
#pragma unroll
    for (int r = 0; r < WAH_WIDE_ROUNDS; ++r) {
        const uint32_t gb = ((uint32_t)r * 1024u + tid) * 32u;
        uint32_t idx = sc[(uint32_t)r * 16u + w] + base[r];
        uint32_t Hr = m[r].H;
        while (__any(Hr != 0u)) {
            if (Hr) {
                const uint32_t k = (uint32_t)__builtin_ctz(Hr);
                Hr &= Hr - 1u;
                const uint32_t g = gb + k;
                const uint32_t nxt = Hr ? gb + (uint32_t)__builtin_ctz(Hr) : nh[r];
                if ((m[r].F >> k) & 1u) {
                    const uint32_t tag = 0x8000u | (((m[r].O >> k) & 1u) << 14);
                    uint32_t len = nxt - g;
                    while (len > WAH_MAXC) {  // a run that outgrows the counter: only ever the unit's last head
                        dst[idx++] = (uint16_t)(tag | WAH_MAXC);
                        len -= WAH_MAXC;
                    }
                    dst[idx++] = (uint16_t)(tag | len);
                } else {
                    const uint32_t o = g * WAH_BITS;
                    dst[idx++] = (uint16_t)(__builtin_amdgcn_alignbit(lrow[(o >> 5) + 1u], lrow[o >> 5], o & 31u) & 0x7FFFu);
                }
            }
        }
    }
}

static bool wah_units_wide_ok(uint32_t y_stride64) {
    const bool off = tuning_env("XSI_WAH_NO_UNITS") != nullptr;
    // rows of whole 16-byte units (y_stride64 even) of at most 65 536 groups
    return !off && y_stride64 > 64u * (uint32_t)WAH_STAGE_Q && (y_stride64 % 2u) == 0u &&
           (uint64_t)y_stride64 * 64u <= (uint64_t)WAH_WIDE_UNITS * 32u * WAH_BITS;
}

template <int MODE>
static hipError_t launch_wah_units_wide(hipStream_t s, const EncBlock* blocks, const EncLines& L, const uint32_t* d_total_wah,
                                        uint32_t max_wah, uint8_t* out, const uint64_t* d_result) {
    const uint32_t rw = L.y_stride64 * 2u;
    const uint32_t units = ((rw * 32u + WAH_BITS - 1u) / WAH_BITS + 31u) / 32u;
    const uint32_t row_words = ((units * 15u + 1u) > rw ? (units * 15u + 1u) : rw) + 3u & ~3u;  // every unit's 15 words (+ the literal read)
    const uint32_t lds = 4u * (row_words + WAH_WIDE_UNITS + 2u * 16u * WAH_WIDE_ROUNDS + 16u * WAH_WIDE_ROUNDS + 2u);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_wah_units_wide<MODE>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    k_wah_units_wide<MODE><<<dim3(max_wah), dim3(1024), lds, s>>>(blocks, L, d_total_wah, max_wah, out, d_result, row_words);
    return hipGetLastError();
}

bool wah_units_ok(uint32_t y_stride64) {
    static const bool off = tuning_env("XSI_WAH_NO_UNITS") != nullptr;
    // below ~12 000 haplotypes a line has fewer than 26 units for 64 lanes: the serial encoder with its scratch
    // copy is faster there (5008 hap x 1 M: sizing + writing 0.85 ms against 1.81 ms)
    return !off && y_stride64 >= 192u && y_stride64 <= 64u * (uint32_t)WAH_STAGE_Q;
}

bool wah_units_any(uint32_t y_stride64) { return wah_units_ok(y_stride64) || wah_units_wide_ok(y_stride64); }

hipError_t launch_wah_sizes(hipStream_t s, const EncLines& L, const uint32_t* d_total_wah, uint32_t max_wah) {
    if (!max_wah) return hipSuccess;
    const uint32_t per_wg = 4u * WAH_LINES_PER_WAVE;
    // the unit encoders size a line AND leave its words in the line's own row (MODE 2, L.wah_inplace; the size-only /
    // write-only modes 0 and 1 of round 2 - two classifications per line - are no longer instantiated)
    if (wah_units_wide_ok(L.y_stride64)) return launch_wah_units_wide<2>(s, nullptr, L, d_total_wah, max_wah, nullptr, nullptr);
    if (wah_units_ok(L.y_stride64)) {
        const uint32_t row_words = wah_units_row_words(L.y_stride64);
        const uint32_t lds = 4u * 4u * (row_words + WAH_UNIT_SCRATCH_WORDS);
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_wah_units<2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        const dim3 grid((max_wah + per_wg - 1u) / per_wg);
        k_wah_units<2><<<grid, dim3(256), lds, s>>>(nullptr, L, d_total_wah, max_wah, nullptr, nullptr, row_words);
        return hipGetLastError();
    }
    if (const uint32_t upl = wah_units_small_upl(L)) {
        const uint32_t lines_per_wg = 4u * (64u / upl);
        if (L.wah_scratch)
            k_wah_units_small<true><<<dim3((max_wah + lines_per_wg - 1u) / lines_per_wg), dim3(256), 0, s>>>(L, d_total_wah, upl);
        else
            k_wah_units_small<false><<<dim3((max_wah + lines_per_wg - 1u) / lines_per_wg), dim3(256), 0, s>>>(L, d_total_wah, upl);
        return hipGetLastError();
    }
    k_wah_sizes<<<dim3((max_wah + per_wg - 1u) / per_wg), dim3(256), 0, s>>>(L, d_total_wah);
    return hipGetLastError();
}

__global__ void __launch_bounds__(256) k_wah_write(const EncBlock* __restrict__ blocks, EncLines L, uint32_t max_wah,
                                                   uint8_t* __restrict__ out, const uint64_t* __restrict__ d_result) {
    if (d_result[3]) return;  // capacity error: nothing may be written
    const uint32_t lane = lane_id();
    const uint32_t j0 = (blockIdx.x * 4u + (threadIdx.x >> 6)) * WAH_LINES_PER_WAVE;
    const uint32_t total = max_wah < (uint32_t)d_result[2] ? max_wah : (uint32_t)d_result[2];
    if (j0 >= total) return;
    // lane k fetches the metadata of line j0 + k (dependent loads, paid once per wave)
    uint64_t m_dst = 0;
    uint32_t m_n = 0, m_nbits = 0;
    if (lane < WAH_LINES_PER_WAVE && j0 + lane < total) {
        const uint32_t l = L.wah_lines[j0 + lane];
        const EncBlock& B = blocks[L.line_block[l]];
        m_dst = reinterpret_cast<uint64_t>(out + B.out_off + 16u + B.off_wah) + 2ull * L.wah_off[j0 + lane];
        m_n = L.wah_len[j0 + lane];
        m_nbits = nbits_of(L, l);
    }
    for (uint32_t k = 0; k < WAH_LINES_PER_WAVE && j0 + k < total; ++k) {
        const uint32_t j = j0 + k;
        const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)m_dst, (int)k);
        const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(m_dst >> 32), (int)k);
        uint16_t* dst = reinterpret_cast<uint16_t*>(((uint64_t)hi << 32) | lo);
        const uint32_t n = (uint32_t)__builtin_amdgcn_readlane((int)m_n, (int)k);
        // no words kept by the sizing pass (none at all, or this line's did not fit its row): encode from the row
        if (L.wah_inplace ? !wah_fits_row(n, L.y_stride64) : !L.wah_scratch) {
            const uint32_t* row = reinterpret_cast<const uint32_t*>(L.yrows + (size_t)j * L.y_stride64);
            (void)wave_wah_encode_row<true>(row, (uint32_t)__builtin_amdgcn_readlane((int)m_nbits, (int)k), dst);
            continue;
        }
        // the sizing pass already produced the words: move them to their final place, 4 bytes per
        // store (the destination is only 2-byte aligned: an odd start takes one word by itself, the
        // source words are then re-paired)
        const uint16_t* src = L.wah_inplace ? reinterpret_cast<const uint16_t*>(L.yrows + (size_t)j * L.y_stride64)
                                            : L.wah_scratch + (size_t)j * L.wah_scratch_stride;  // 4-byte aligned rows
        const uint32_t head = (uint32_t)((reinterpret_cast<uint64_t>(dst) >> 1) & 1ull);
        if (n <= head) {
            if (head && n && lane == 0) ((__attribute__((address_space(1))) uint16_t*)dst)[0] = ((const __attribute__((address_space(1))) uint16_t*)src)[0];
            continue;
        }
        const uint32_t m = n - head;  // words from src + head to the 4-byte aligned dst + head
        // Everything here is global memory, said so: dst comes out of two v_readlane halves, which makes it "generic" to
        // the compiler, and a FLAT store (the single words at either end were) left outstanding forces every later
        // wait to be a wait for all loads.  16 bytes per lane and step: four loads in flight instead of one.
        using G32 = __attribute__((address_space(1))) uint32_t;
        using G16 = __attribute__((address_space(1))) uint16_t;
        typedef uint32_t quad_a4 __attribute__((ext_vector_type(4), aligned(4)));
        using GQuad = __attribute__((address_space(1))) quad_a4;
        const G32* s32 = (const G32*)src;
        G32* d32 = (G32*)(dst + head);
        if (head && n && lane == 0) ((G16*)dst)[0] = ((const G16*)src)[0];
        const uint32_t nd = m / 2u;
        for (uint32_t q = lane; q < nd / 4u; q += 64u) {
            const uint32_t b = 4u * q;
            const uint32_t a0 = s32[b], a1 = s32[b + 1u], a2 = s32[b + 2u], a3 = s32[b + 3u];
            quad_a4 o;
            if (head) {
                const uint32_t a4 = s32[b + 4u];
                o = quad_a4{(a0 >> 16) | (a1 << 16), (a1 >> 16) | (a2 << 16), (a2 >> 16) | (a3 << 16), (a3 >> 16) | (a4 << 16)};
            } else {
                o = quad_a4{a0, a1, a2, a3};
            }
            *(GQuad*)(d32 + b) = o;
        }
        for (uint32_t i = (nd & ~3u) + lane; i < nd; i += 64u) d32[i] = head ? (s32[i] >> 16) | (s32[i + 1u] << 16) : s32[i];
        if ((m & 1u) && lane == 0) ((G16*)dst)[n - 1u] = ((const G16*)src)[n - 1u];
    }
}

hipError_t launch_wah_write(hipStream_t s, const EncBlock* blocks, const EncLines& L, uint32_t max_wah,
                            uint8_t* out, const uint64_t* d_result) {
    if (!max_wah) return hipSuccess;
    const uint32_t per_wg = 4u * WAH_LINES_PER_WAVE;
    k_wah_write<<<dim3((max_wah + per_wg - 1u) / per_wg), dim3(256), 0, s>>>(blocks, L, max_wah, out, d_result);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// sparse lines: ascending list of set positions (Sparse<T,Pred> ctor + SparseGtLine::
// write_to_stream, block.hpp:54-99).  One wave per sparse line: each lane popcounts one 32-bit
// word, a wave scan gives its slot, the lane then peels its set bits.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void store_at(uint8_t* p, uint32_t v, uint32_t aet) {
    // A_T values may sit on 2-byte boundaries even when A_T is 4 bytes
    uint16_t* q = reinterpret_cast<uint16_t*>(p);
    q[0] = (uint16_t)v;
    if (aet == 4u) q[1] = (uint16_t)(v >> 16);
}

__device__ __forceinline__ uint32_t wave_sparse_emit(const uint32_t* __restrict__ row, uint32_t nbits, bool invert,
                                                     uint32_t msb_flag, uint32_t aet, uint8_t* __restrict__ dst) {
    const uint32_t lane = lane_id();
    const uint32_t nw = (nbits + 31u) >> 5;
    uint32_t base = 0;
    for (uint32_t w0 = 0; w0 < nw; w0 += 64u) {
        const uint32_t w = w0 + lane;
        uint32_t v = 0;
        if (w < nw) {
            v = row[w];
            if (invert) v = ~v;
            if (w == nw - 1u && (nbits & 31u)) v &= (1u << (nbits & 31u)) - 1u;
        }
        const uint32_t c = (uint32_t)__popc(v);
        const uint32_t inc = wave_scan_incl(c);
        uint32_t pos = base + inc - c;
        while (v) {
            const uint32_t bpos = (uint32_t)__ffs((int)v) - 1u;
            v &= v - 1u;
            store_at(dst + (size_t)(1u + pos) * aet, w * 32u + bpos, aet);
            ++pos;
        }
        base += __shfl(inc, 63, 64);
    }
    if (lane == 0) {
        uint32_t head = base;
        if (msb_flag) head |= (aet == 2u) ? 0x8000u : 0x80000000u;
        store_at(dst, head, aet);
    }
    return base;
}

// Long rows (16-byte aligned, whole 16-byte units): four words per lane and step, the next step's load in flight
// while the current one is scanned and peeled.  One word per lane leaves a 62.5 KB row of 500 000 haplotypes as 245
// dependent 256-byte loads per wave: 18.8 ms for the sparse lines of a configs[3] shard, 1.7 TB/s.
__device__ __forceinline__ uint32_t wave_sparse_emit_wide(const uint32_t* __restrict__ row, uint32_t nbits, bool invert,
                                                          uint32_t msb_flag, uint32_t aet, uint8_t* __restrict__ dst) {
    const uint32_t lane = lane_id();
    const uint32_t nw = (nbits + 31u) >> 5, nq = (nw + 3u) >> 2;
    const uint4* row4 = reinterpret_cast<const uint4*>(row);
    const uint32_t last_mask = (nbits & 31u) ? (1u << (nbits & 31u)) - 1u : ~0u;
    uint32_t base = 0;
    // two loads ahead: with one, a wave moved 1 KiB per memory round trip and the kernel sat at 4.1 TB/s at 500 000
    // haplotypes (8192 waves x 1 KiB / 2 us), latency-bound
    uint4 nxt = row4[lane < nq ? lane : 0u];
    uint4 nxt2 = row4[64u + lane < nq ? 64u + lane : 0u];
    for (uint32_t q0 = 0; q0 < nq; q0 += 64u) {
        const uint4 cur = nxt;
        nxt = nxt2;
        const uint32_t qn = q0 + 128u + lane;
        nxt2 = row4[qn < nq ? qn : 0u];  // unconditional: the loads stay in flight over the work below
        const uint32_t q = q0 + lane;
        uint32_t v[4] = {cur.x, cur.y, cur.z, cur.w};
        uint32_t c = 0;
#pragma unroll
        for (uint32_t i = 0; i < 4u; ++i) {
            const uint32_t w = q * 4u + i;
            uint32_t x = invert ? ~v[i] : v[i];
            if (q >= nq || w >= nw) x = 0;
            if (w == nw - 1u) x &= last_mask;
            v[i] = x;
            c += (uint32_t)__popc(x);
        }
        if (__builtin_amdgcn_ballot_w64(c != 0u) == 0ull) continue;  // a sparse line is mostly this
        const uint32_t inc = wave_scan_incl(c);
        uint32_t pos = base + inc - c;
#pragma unroll
        for (uint32_t i = 0; i < 4u; ++i) {
            uint32_t x = v[i];
            while (x) {
                const uint32_t bpos = (uint32_t)__ffs((int)x) - 1u;
                x &= x - 1u;
                store_at(dst + (size_t)(1u + pos) * aet, (q * 4u + i) * 32u + bpos, aet);
                ++pos;
            }
        }
        base += __shfl(inc, 63, 64);
    }
    if (lane == 0) {
        uint32_t head = base;
        if (msb_flag) head |= (aet == 2u) ? 0x8000u : 0x80000000u;
        store_at(dst, head, aet);
    }
    return base;
}

// scratch != nullptr: the lists go to scratch + block * scratch_stride + sparse_off (run before the
// block layout exists, underneath the chain); k_sparse_copy then moves each block's region into place.
__global__ void __launch_bounds__(256) k_sparse_write(const EncBlock* __restrict__ blocks, EncLines L,
                                                      uint8_t* __restrict__ out, const uint64_t* __restrict__ d_result,
                                                      uint8_t* __restrict__ scratch, uint64_t scratch_stride) {
    if (!scratch && d_result[3]) return;
    const uint32_t l = blockIdx.x * 4u + (threadIdx.x >> 6);
    if (l >= L.n_bin) return;
    const uint32_t k = L.kind[l];
    if (k & KIND_WAH) return;
    const uint32_t blk = L.line_block[l];
    const uint32_t nbits = nbits_of(L, l);
    const bool neg = (k & KIND_NEGATED) != 0u;
    const uint32_t* row;
    bool invert = false;
    if (neg && L.ref_planes) {
        row = L.ref_planes + (size_t)L.bin_parent[l] * L.plane_stride_w;  // positions with allele 0
    } else {
        row = L.planes + (size_t)l * L.plane_stride_w;
        invert = neg;  // fully called bi-allelic line: REF positions = complement of ALT positions
    }
    uint8_t* dst = scratch ? scratch + (size_t)blk * scratch_stride + L.sparse_off[l]
                           : out + blocks[blk].out_off + 16u + blocks[blk].off_sparse + L.sparse_off[l];
    // rows of whole 16-byte units on 16-byte addresses (the stride in words is a multiple of 4): the wide form
    if (nbits >= 32768u && (L.plane_stride_w & 3u) == 0u && (reinterpret_cast<uintptr_t>(row) & 15u) == 0u)
        (void)wave_sparse_emit_wide(row, nbits, invert, neg ? 1u : 0u, L.aet, dst);
    else
        (void)wave_sparse_emit(row, nbits, invert, neg ? 1u : 0u, L.aet, dst);
}

hipError_t launch_sparse_write(hipStream_t s, const EncBlock* blocks, const EncLines& L, uint8_t* out,
                               const uint64_t* d_result, uint8_t* scratch, uint64_t scratch_stride) {
    if (!L.n_bin) return hipSuccess;
    k_sparse_write<<<dim3((L.n_bin + 3u) / 4u), dim3(256), 0, s>>>(blocks, L, out, d_result, scratch, scratch_stride);
    return hipGetLastError();
}

// each block's sparse matrix from the scratch into its place in the output (2-byte granular)
__global__ void __launch_bounds__(256) k_sparse_copy(const EncBlock* __restrict__ blocks, uint8_t* __restrict__ out,
                                                     const uint64_t* __restrict__ d_result,
                                                     const uint8_t* __restrict__ scratch, uint64_t scratch_stride) {
    if (d_result[3]) return;
    const EncBlock& B = blocks[blockIdx.x];
    const uint16_t* src = reinterpret_cast<const uint16_t*>(scratch + (size_t)blockIdx.x * scratch_stride);
    uint16_t* dst = reinterpret_cast<uint16_t*>(out + B.out_off + 16u + B.off_sparse);
    const uint32_t n = B.sparse_bytes / 2u;
    for (uint32_t i = blockIdx.y * blockDim.x + threadIdx.x; i < n; i += blockDim.x * gridDim.y) dst[i] = src[i];
}

hipError_t launch_sparse_copy(hipStream_t s, const EncBlock* blocks, uint32_t n_blocks, uint8_t* out,
                              const uint64_t* d_result, const uint8_t* scratch, uint64_t scratch_stride) {
    if (!n_blocks) return hipSuccess;
    k_sparse_copy<<<dim3(n_blocks, 8), dim3(256), 0, s>>>(blocks, out, d_result, scratch, scratch_stride);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// block layout: one wave per block.  Exclusive scan of the WAH line lengths, WAH16 encoding
// of the flag vectors (write_boolean_vector_as_wah, gt_block.hpp:675-679) into scratch, and
// the section offsets in the order GtBlock::write_writables emits them (gt_block.hpp:512-647).
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64) k_block_layout(EncBlock* __restrict__ blocks, EncLines L, EncSide S,
                                                     int32_t default_phased) {
    const uint32_t b = blockIdx.x;
    EncBlock& B = blocks[b];
    const uint32_t lane = lane_id();
    // scan of WAH lengths
    uint32_t base = 0;
    for (uint32_t j0 = 0; j0 < B.n_wah; j0 += 64u) {
        const uint32_t j = j0 + lane;
        const uint32_t v = j < B.n_wah ? L.wah_len[B.wah_first + j] : 0u;
        const uint32_t inc = wave_scan_incl(v);
        if (j < B.n_wah) L.wah_off[B.wah_first + j] = base + inc - v;
        base += __shfl(inc, 63, 64);
    }
    const uint32_t wah_words = base;
    // side channels (general path): per-block scans over the BCF lines
    uint32_t miss_bytes = 0, eov_bytes = 0, phase_words = 0;
    uint32_t any_m = 0, any_e = 0, any_p = 0, any_h = 0, maxp = 1;
    if (S.bcf_flags) {
        for (uint32_t i0 = 0; i0 < B.n_bcf; i0 += 64u) {
            const uint32_t i = i0 + lane;
            const bool valid = i < B.n_bcf;
            const uint32_t li = B.first_bcf + i;
            const uint32_t f = valid ? S.bcf_flags[li] : 0u;
            uint32_t mb = 0, ebb = 0, pw = 0;
            if (valid) {
                if (f & 1u) mb = S.miss_size[li];
                if (f & 2u) ebb = S.eov_size[li];
                if (f & 4u) pw = S.phase_len[li];
            }
            const uint32_t im = wave_scan_incl(mb), ie = wave_scan_incl(ebb), ip = wave_scan_incl(pw);
            if (valid) {
                S.miss_off[li] = miss_bytes + im - mb;
                S.eov_off[li] = eov_bytes + ie - ebb;
                S.phase_off[li] = phase_words + ip - pw;
            }
            miss_bytes += __shfl(im, 63, 64);
            eov_bytes += __shfl(ie, 63, 64);
            phase_words += __shfl(ip, 63, 64);
            any_m |= __any(f & 1u) ? 1u : 0u;
            any_e |= __any(f & 2u) ? 1u : 0u;
            any_p |= __any(f & 4u) ? 1u : 0u;
            any_h |= __any(f & 8u) ? 1u : 0u;
            if (__any(valid && !(f & 8u))) maxp = 2;
        }
    } else {
        maxp = 2;
    }
    // flag vectors
    uint32_t flen[FV_COUNT] = {0, 0, 0, 0, 0};
    const uint32_t present[FV_COUNT] = {1u, any_m, any_e, any_p, any_h};
    for (uint32_t v = 0; v < FV_COUNT; ++v) {
        if (!present[v]) continue;
        const uint32_t* bits = L.flagbits + ((size_t)b * FV_COUNT + v) * (MAX_BIN_PER_BLOCK / 32);
        uint16_t* dst = L.flagwah + ((size_t)b * FV_COUNT + v) * FLAG_WORDS_MAX;
        // KEY_LINE_HAPLOID carries one bit per BCF line, the others one per binary line
        const uint32_t nb = (v == FV_HAPLOID) ? B.n_bcf : B.n_bin;
        flen[v] = wave_wah_encode_row<true>(bits, nb, dst);
    }
    if (lane == 0) {
        B.wah_words = wah_words;
        B.has_missing = any_m;
        B.has_eov = any_e;
        B.has_phase = any_p;
        B.has_haploid = any_h;
        B.max_ploidy = maxp;
        B.miss_bytes = miss_bytes;
        B.eov_bytes = eov_bytes;
        B.phase_words = phase_words;
        for (uint32_t v = 0; v < FV_COUNT; ++v) B.flag_len[v] = flen[v];
        B.dict_idx = any_m | (any_e << 1) | (any_p << 2) | (any_h << 3);
        B.n_keys = c_dict_order[B.dict_idx][0];
        uint32_t off = 8u + 8u * B.n_keys;
        B.off_flag[FV_IS_WAH] = off;
        off += 2u * flen[FV_IS_WAH];
        B.off_wah = off;
        off += 2u * wah_words;
        B.off_sparse = off;
        off += B.sparse_bytes;
        B.off_flag[FV_MISSING] = off;
        B.off_miss = off;
        if (any_m) {
            off += 2u * flen[FV_MISSING];
            B.off_miss = off;
            off += miss_bytes;
        }
        B.off_flag[FV_EOV] = off;
        B.off_eov = off;
        if (any_e) {
            off += 2u * flen[FV_EOV];
            B.off_eov = off;
            off += eov_bytes;
        }
        B.off_flag[FV_PHASE] = off;
        B.off_phase = off;
        if (any_p) {
            off += 2u * flen[FV_PHASE];
            B.off_phase = off;
            off += 2u * phase_words;
        }
        B.off_flag[FV_HAPLOID] = off;
        if (any_h) off += 2u * flen[FV_HAPLOID];
        B.gt_bytes = off;
        B.block_bytes = (16u + off + 3u) & ~3u;
    }
    (void)default_phased;
}

hipError_t launch_block_layout(hipStream_t s, EncBlock* blocks, uint32_t n_blocks, const EncLines& L,
                               const EncSide& S, int32_t default_phased) {
    if (!n_blocks) return hipSuccess;
    k_block_layout<<<dim3(n_blocks), dim3(64), 0, s>>>(blocks, L, S, default_phased);
    return hipGetLastError();
}

// block offsets in the blocks region (xsi_factory.hpp:533: indices.push_back(s.tellp())).
// d_result: [0] total bytes, [1] n_blocks, [2] total WAH lines, [3] error (1 = capacity)
__global__ void __launch_bounds__(1024) k_scan_blocks_out(EncBlock* __restrict__ blocks, uint32_t n_blocks,
                                                          uint64_t capacity, uint64_t* __restrict__ d_block_offsets,
                                                          uint64_t* __restrict__ d_result, uint64_t file_base,
                                                          uint32_t* __restrict__ d_block_sizes) {
    __shared__ uint64_t s_scan[20];
    uint64_t base = 0;
    uint32_t wah = 0;
    for (uint32_t c0 = 0; c0 < n_blocks; c0 += blockDim.x) {
        const uint32_t i = c0 + threadIdx.x;
        const uint64_t v = i < n_blocks ? blocks[i].block_bytes : 0u;
        uint64_t tot;
        const uint64_t ex = block_scan_excl64(v, s_scan, &tot);
        if (i < n_blocks) {
            blocks[i].out_off = base + ex;
            if (d_block_offsets) d_block_offsets[i] = file_base + base + ex;  // 256 + bytes of earlier blocks
            if (d_block_sizes) d_block_sizes[i] = 16u + blocks[i].gt_bytes;   // the block as streamed, before its pad (what the zstd layer compresses)
        }
        base += tot;
    }
    if (threadIdx.x == 0) {
        for (uint32_t i = 0; i < n_blocks; ++i) wah += blocks[i].n_wah;
        d_result[0] = base;
        d_result[1] = n_blocks;
        d_result[2] = wah;
        d_result[3] = base > capacity ? 1u : 0u;
        d_result[4] = n_blocks ? 16u + blocks[n_blocks - 1u].gt_bytes : 0u;
    }
}

hipError_t launch_scan_blocks_out(hipStream_t s, EncBlock* blocks, uint32_t n_blocks, uint64_t capacity,
                                  uint64_t* d_block_offsets, uint64_t* d_result, uint64_t file_base, uint32_t* d_block_sizes) {
    k_scan_blocks_out<<<dim3(1), dim3(1024), 0, s>>>(blocks, n_blocks, capacity, d_block_offsets, d_result, file_base, d_block_sizes);
    return hipGetLastError();
}

// outer dictionary, GT dictionary and flag vectors of every block (interfaces.hpp:176-239;
// gt_block.hpp:185-204, 464-510), plus the zero pad to 4 bytes (interfaces.hpp:254-263).
__global__ void __launch_bounds__(256) k_write_headers(const EncBlock* __restrict__ blocks, EncLines L,
                                                       int32_t default_phased, uint32_t strategy,
                                                       uint8_t* __restrict__ out, const uint64_t* __restrict__ d_result) {
    if (d_result[3]) return;
    const uint32_t b = blockIdx.x;
    const EncBlock& B = blocks[b];
    uint8_t* blk = out + B.out_off;
    uint32_t* o32 = reinterpret_cast<uint32_t*>(blk);
    if (threadIdx.x == 0) {
        o32[0] = 0xFFFFFFFFu;
        o32[1] = 1u;
        o32[2] = KEY_GT_ENTRY;
        o32[3] = 16u;
        o32[4] = 0xFFFFFFFFu;
        o32[5] = B.n_keys;
        for (uint32_t i = 0; i < B.n_keys; ++i) {
            const uint32_t key = c_dict_order[B.dict_idx][1 + i];
            uint32_t val = VAL_UNDEFINED;
            switch (key) {
                case KEY_BCF_LINES: val = B.n_bcf; break;
                case KEY_BINARY_LINES: val = B.n_bin; break;
                case KEY_MAX_LINE_PLOIDY: val = B.max_ploidy; break;
                case KEY_DEFAULT_PHASING: val = (uint32_t)default_phased; break;
                case KEY_WEIRDNESS_STRATEGY: val = strategy; break;
                case KEY_LINE_SORT:
                case KEY_LINE_SELECT: val = B.off_flag[FV_IS_WAH]; break;
                case KEY_MATRIX_WAH: val = B.off_wah; break;
                case KEY_MATRIX_SPARSE: val = B.off_sparse; break;
                case KEY_LINE_MISSING: val = B.off_flag[FV_MISSING]; break;
                case KEY_MATRIX_MISSING: val = (strategy == WS_SPARSE) ? VAL_UNDEFINED : B.off_miss; break;
                case KEY_MATRIX_MISSING_SPARSE: val = (strategy == WS_SPARSE) ? B.off_miss : VAL_UNDEFINED; break;
                case KEY_LINE_END_OF_VECTORS: val = B.off_flag[FV_EOV]; break;
                case KEY_MATRIX_END_OF_VECTORS: val = (strategy == WS_SPARSE) ? VAL_UNDEFINED : B.off_eov; break;
                case KEY_MATRIX_END_OF_VECTORS_SPARSE: val = (strategy == WS_SPARSE) ? B.off_eov : VAL_UNDEFINED; break;
                case KEY_LINE_NON_UNIFORM_PHASING: val = B.off_flag[FV_PHASE]; break;
                case KEY_MATRIX_NON_UNIFORM_PHASING: val = B.off_phase; break;
                case KEY_LINE_HAPLOID: val = B.off_flag[FV_HAPLOID]; break;
                default: break;
            }
            o32[6 + 2 * i] = key;
            o32[7 + 2 * i] = val;
        }
        // pad bytes
        for (uint32_t p = 16u + B.gt_bytes; p < B.block_bytes; ++p) blk[p] = 0;
    }
    const uint32_t present[FV_COUNT] = {1u, B.has_missing, B.has_eov, B.has_phase, B.has_haploid};
    for (uint32_t v = 0; v < FV_COUNT; ++v) {
        if (!present[v]) continue;
        const uint16_t* src = L.flagwah + ((size_t)b * FV_COUNT + v) * FLAG_WORDS_MAX;
        uint16_t* dst = reinterpret_cast<uint16_t*>(blk + 16u + B.off_flag[v]);
        for (uint32_t i = threadIdx.x; i < B.flag_len[v]; i += blockDim.x) dst[i] = src[i];
    }
}

hipError_t launch_write_headers(hipStream_t s, const EncBlock* blocks, uint32_t n_blocks, const EncLines& L,
                                int32_t default_phased, uint32_t strategy, uint8_t* out, const uint64_t* d_result) {
    if (!n_blocks) return hipSuccess;
    k_write_headers<<<dim3(n_blocks), dim3(256), 0, s>>>(blocks, L, default_phased, strategy, out, d_result);
    return hipGetLastError();
}

// ==========================================================================================
// Decode
// ==========================================================================================
__device__ __forceinline__ uint32_t rd32(const uint8_t* p) {  // 4-byte aligned by format
    return *reinterpret_cast<const uint32_t*>(p);
}
__device__ __forceinline__ uint32_t rd_at(const uint8_t* p, uint32_t aet) {  // 2-byte aligned A_T
    const uint16_t* q = reinterpret_cast<const uint16_t*>(p);
    uint32_t v = q[0];
    if (aet == 4u) v |= (uint32_t)q[1] << 16;
    return v;
}

// parse the outer and GT dictionaries of each block (set_block_ptr + DecompressPointerGTBlock
// ctor, accessor_internals_new.hpp:830-893, 52-148).  One thread per block.
__global__ void __launch_bounds__(256) k_parse_blocks(const uint8_t* __restrict__ file, uint64_t file_len,
                                                      uint64_t indices_offset, uint32_t version, uint64_t first_block,
                                                      uint32_t n_blocks, DecBlock* __restrict__ blocks) {
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= n_blocks) return;
    DecBlock D{};
    const uint64_t bi = first_block + b;
    uint64_t off;
    if (version >= 5) {
        const uint32_t* ip = reinterpret_cast<const uint32_t*>(file + indices_offset + bi * 8u);
        off = (uint64_t)ip[0] | ((uint64_t)ip[1] << 32);
    } else {
        off = rd32(file + indices_offset + bi * 4u);
    }
    D.file_off = off;
    D.error = 0;
    // every read below stays inside the image: a corrupt index or dictionary becomes an error
    // code (XSI_ERR_FORMAT on the host), never an out-of-bounds access
    if (off + 16u > file_len || (off & 3u)) {
        D.error = 1;
        blocks[b] = D;
        return;
    }
    const uint8_t* blk = file + off;
    uint32_t n_outer = rd32(blk + 4);
    if (n_outer > 64u) n_outer = 64u;
    if (off + 8u + 8ull * n_outer > file_len) {
        D.error = 1;
        blocks[b] = D;
        return;
    }
    uint32_t gt_rel = VAL_UNDEFINED;
    for (uint32_t i = 0; i < n_outer; ++i)
        if (rd32(blk + 8 + 8 * i) == KEY_GT_ENTRY) gt_rel = rd32(blk + 12 + 8 * i);
    if (gt_rel == VAL_UNDEFINED || (gt_rel & 3u) || off + gt_rel + 8u > file_len) {
        D.error = 2;
        blocks[b] = D;
        return;
    }
    D.gt_off = off + gt_rel;
    const uint8_t* gt = file + D.gt_off;
    uint32_t n = rd32(gt + 4);
    if (n > 64u) n = 64u;
    if (D.gt_off + 8u + 8ull * n > file_len) {
        D.error = 2;
        blocks[b] = D;
        return;
    }
    // the dictionary is walked once per key (the last entry of a key wins, as an array indexed by key would have it):
    // a private array indexed at run time would live in scratch memory
    auto val = [&](uint32_t key) {
        uint32_t v = VAL_UNDEFINED;
        for (uint32_t i = 0; i < n; ++i)
            if (rd32(gt + 8 + 8 * i) == key) v = rd32(gt + 12 + 8 * i);
        return v;
    };
    D.n_bcf = val(KEY_BCF_LINES);
    D.n_bin = val(KEY_BINARY_LINES);
    const uint32_t v_ploidy = val(KEY_MAX_LINE_PLOIDY), v_strategy = val(KEY_WEIRDNESS_STRATEGY);
    D.max_ploidy = v_ploidy == VAL_UNDEFINED ? 2u : v_ploidy;
    D.default_phasing = val(KEY_DEFAULT_PHASING) == 1u ? 1u : 0u;  // accessor_internals_new.hpp:77-81
    D.strategy = v_strategy == VAL_UNDEFINED ? WS_PBWT_WAH : v_strategy;
    D.off_select = val(KEY_LINE_SELECT);
    D.off_wah = val(KEY_MATRIX_WAH);
    D.off_sparse = val(KEY_MATRIX_SPARSE);
    D.off_line_missing = val(KEY_LINE_MISSING);
    D.off_miss_wah = val(KEY_MATRIX_MISSING);
    D.off_miss_sparse = val(KEY_MATRIX_MISSING_SPARSE);
    D.off_line_eov = val(KEY_LINE_END_OF_VECTORS);
    D.off_eov_wah = val(KEY_MATRIX_END_OF_VECTORS);
    D.off_eov_sparse = val(KEY_MATRIX_END_OF_VECTORS_SPARSE);
    D.off_line_phase = val(KEY_LINE_NON_UNIFORM_PHASING);
    D.off_phase = val(KEY_MATRIX_NON_UNIFORM_PHASING);
    D.off_line_haploid = val(KEY_LINE_HAPLOID);
    if (D.n_bcf == VAL_UNDEFINED || D.n_bin == VAL_UNDEFINED || D.n_bin > MAX_BIN_PER_BLOCK ||
        D.off_select == VAL_UNDEFINED || D.off_wah == VAL_UNDEFINED || D.off_sparse == VAL_UNDEFINED ||
        D.off_sparse < D.off_wah)
        D.error = 3;
    // every section must start inside the image (the WAH matrix must also end inside it)
    const uint32_t sections[] = {D.off_select,       D.off_wah,      D.off_sparse,      D.off_line_missing, D.off_miss_wah,
                                 D.off_miss_sparse,  D.off_line_eov, D.off_eov_wah,     D.off_eov_sparse,   D.off_line_phase,
                                 D.off_phase,        D.off_line_haploid};
    for (uint32_t x : sections)
        if (x != VAL_UNDEFINED && D.gt_off + (uint64_t)x > file_len) D.error = 3;
    D.wah_words = D.error ? 0u : (D.off_sparse - D.off_wah) / 2u;
    blocks[b] = D;
}

hipError_t launch_parse_blocks(hipStream_t s, const uint8_t* file, uint64_t file_len, uint64_t indices_offset,
                               uint32_t version, uint64_t first_block, uint32_t n_blocks, DecBlock* blocks,
                               uint32_t* d_totals) {
    (void)d_totals;
    if (!n_blocks) return hipSuccess;
    k_parse_blocks<<<dim3((n_blocks + 255u) / 256u), dim3(256), 0, s>>>(file, file_len, indices_offset, version,
                                                                        first_block, n_blocks, blocks);
    return hipGetLastError();
}

// batch-wide exclusive scans over the parsed blocks: first_bin / first_bcf.
// d_totals: [0] binary lines, [1] WAH lines, [2] sparse lines, [3] error, [4] BCF lines
__global__ void __launch_bounds__(1024) k_scan_dec_blocks(DecBlock* __restrict__ blocks, uint32_t n_blocks,
                                                          uint32_t* __restrict__ d_totals, int phase) {
    __shared__ uint64_t s_scan[20];
    uint32_t b0 = 0, b1 = 0, err = 0;
    for (uint32_t c0 = 0; c0 < n_blocks; c0 += blockDim.x) {
        const uint32_t i = c0 + threadIdx.x;
        uint64_t v = 0;
        if (i < n_blocks) {
            if (phase == 0)
                v = ((uint64_t)blocks[i].n_bin << 32) | blocks[i].n_bcf;
            else
                v = ((uint64_t)blocks[i].n_wah << 32) | blocks[i].n_sparse;
            if (blocks[i].error) err = 1;
        }
        uint64_t tot;
        const uint64_t ex = block_scan_excl64(v, s_scan, &tot);
        if (i < n_blocks) {
            if (phase == 0) {
                blocks[i].first_bin = b0 + (uint32_t)(ex >> 32);
                blocks[i].first_bcf = b1 + (uint32_t)ex;
            } else {
                blocks[i].wah_first = b0 + (uint32_t)(ex >> 32);
                blocks[i].sparse_first = b1 + (uint32_t)ex;
            }
        }
        b0 += (uint32_t)(tot >> 32);
        b1 += (uint32_t)tot;
    }
    err = __syncthreads_or((int)err) ? 1u : 0u;
    if (threadIdx.x == 0) {
        if (phase == 0) {
            d_totals[0] = b0;
            d_totals[4] = b1;
            d_totals[3] = err;
        } else {
            d_totals[1] = b0;
            d_totals[2] = b1;
            if (err) d_totals[3] = 1;
        }
    }
}

hipError_t launch_scan_dec_blocks(hipStream_t s, DecBlock* blocks, uint32_t n_blocks, uint32_t* d_totals) {
    k_scan_dec_blocks<<<dim3(1), dim3(1024), 0, s>>>(blocks, n_blocks, d_totals, 0);
    return hipGetLastError();
}

// flag vectors -> per-line kinds and ranks (fill_bool_vector_from_1d_dict_key,
// accessor_internals_new.hpp:591-604; sorting lines == WAH lines for this writer).  One
// workgroup of 64 per block; the vector is expanded into LDS.
__global__ void __launch_bounds__(64) k_decode_flags(const uint8_t* __restrict__ file, DecBlock* __restrict__ blocks,
                                                     DecLines L) {
    __shared__ uint32_t s_row[MAX_BIN_PER_BLOCK / 32 + 2];
    __shared__ uint32_t s_hap[MAX_BIN_PER_BLOCK / 32 + 2];
    DecBlock& D = blocks[blockIdx.x];
    if (D.error) return;
    const uint32_t lane = lane_id();
    const uint32_t nb = D.n_bin;
    const uint32_t nw = (nb + 31u) >> 5;
    for (uint32_t i = lane; i < nw + 1u; i += 64u) {
        s_row[i] = 0;
        s_hap[i] = 0;
    }
    __syncthreads();
    uint32_t ones;
    const uint8_t* gt = file + D.gt_off;
    auto words_left = [&](uint32_t rel) -> uint32_t {
        const uint64_t at = D.gt_off + rel;
        if (at >= L.file_len) return 0u;
        const uint64_t left = (L.file_len - at) / 2u;
        return left < FLAG_WORDS_MAX ? (uint32_t)left : FLAG_WORDS_MAX;
    };
    (void)wave_wah_expand_row(reinterpret_cast<const uint16_t*>(gt + D.off_select), words_left(D.off_select), nb, s_row, &ones);
    if (D.off_line_haploid != VAL_UNDEFINED) {
        // read per binary line although written per BCF line (kept)
        uint32_t o2;
        (void)wave_wah_expand_row(reinterpret_cast<const uint16_t*>(gt + D.off_line_haploid),
                                  words_left(D.off_line_haploid), nb, s_hap, &o2);
    }
    __syncthreads();
    uint32_t wbase = 0, sbase = 0;
    for (uint32_t i0 = 0; i0 < nb; i0 += 64u) {
        const uint32_t i = i0 + lane;
        const bool valid = i < nb;
        const uint32_t isw = valid ? ((s_row[i >> 5] >> (i & 31u)) & 1u) : 0u;
        const uint32_t hap = valid ? ((s_hap[i >> 5] >> (i & 31u)) & 1u) : 0u;
        const uint64_t Wm = __ballot(isw), Vm = __ballot(valid);
        const uint32_t wr = wbase + mbcnt64(Wm);
        const uint32_t sr = sbase + mbcnt64(Vm & ~Wm);
        if (valid) {
            const uint32_t l = D.first_bin + i;
            L.kind[l] = (uint8_t)((isw ? KIND_WAH : 0u) | (hap ? KIND_HAPLOID : 0u));
            L.line_block[l] = blockIdx.x;
            L.rank[l] = isw ? wr : sr;
        }
        wbase += (uint32_t)__popcll(Wm);
        sbase += (uint32_t)__popcll(Vm & ~Wm);
    }
    if (lane == 0) {
        D.n_wah = wbase;
        D.n_sparse = sbase;
    }
}

hipError_t launch_decode_flags(hipStream_t s, const uint8_t* file, DecBlock* blocks, uint32_t n_blocks,
                               const DecLines& L) {
    if (!n_blocks) return hipSuccess;
    k_decode_flags<<<dim3(n_blocks), dim3(64), 0, s>>>(file, blocks, L);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    return hipSuccess;
}

hipError_t launch_scan_dec_blocks2(hipStream_t s, DecBlock* blocks, uint32_t n_blocks, uint32_t* d_totals) {
    k_scan_dec_blocks<<<dim3(1), dim3(1024), 0, s>>>(blocks, n_blocks, d_totals, 1);
    return hipGetLastError();
}

__global__ void __launch_bounds__(256) k_dec_line_lists(const DecBlock* __restrict__ blocks, DecLines L) {
    const DecBlock& D = blocks[blockIdx.x];
    if (D.error) return;
    for (uint32_t i = threadIdx.x; i < D.n_bin; i += blockDim.x) {
        const uint32_t l = D.first_bin + i;
        if (L.kind[l] & KIND_WAH)
            L.wah_lines[D.wah_first + L.rank[l]] = l;
        else
            L.sparse_lines[D.sparse_first + L.rank[l]] = l;
    }
}

hipError_t launch_dec_line_lists(hipStream_t s, const DecBlock* blocks, uint32_t n_blocks, const DecLines& L) {
    if (!n_blocks) return hipSuccess;
    k_dec_line_lists<<<dim3(n_blocks), dim3(256), 0, s>>>(blocks, L);
    return hipGetLastError();
}

// WAH line boundaries.  Lines carry no length: a line ends when its
// ceil(n/15) groups are covered, and this writer never lets a fill run cross a line.  So the
// running group count over the whole matrix hits every line's cumulative group offset exactly
// at that line's first word.  Three small kernels so that every block's matrix is scanned by many
// workgroups: (1) groups per tile of 2048 words, (2) per block exclusive scan of the tile sums
// (+ the cumulative line offsets of mixed-ploidy blocks), (3) per tile: scan inside the tile and
// report the words at which a line starts.
// A thread takes BND_Q consecutive 8-word groups (64 bytes); smaller tiles (one group per thread: 343 000 workgroups
// at configs[2]) were bound by workgroup dispatch and by the scan's barrier, not by the 1.4 GB they read.
constexpr uint32_t BND_T = 256, BND_K = 8, BND_Q = 4, BND_TILE = BND_T * BND_K * BND_Q;
static_assert(BND_TILE == WAH_BND_TILE_WORDS, "the host sizes tile_sum / tile_base with WAH_BND_TILE_WORDS");

__device__ __forceinline__ uint32_t wah_groups_of(uint32_t word) { return (word & 0x8000u) ? (word & WAH_MAXC) : 1u; }

// The BND_K = 8 consecutive WAH16 words of a thread: one 16-byte load (the matrix is only 2-byte aligned; gfx950
// global loads take that) instead of eight 2-byte loads; the ragged end of the matrix word by word.
struct __attribute__((packed, aligned(2))) WahWords8 {
    uint32_t v[4];
};
__device__ __forceinline__ void load_words8(const uint16_t* __restrict__ wm, uint32_t w0, uint32_t nwords, uint32_t (&wd)[8]) {
    if (w0 + 8u <= nwords) {
        const WahWords8 p = *reinterpret_cast<const WahWords8*>(wm + w0);
#pragma unroll
        for (int k = 0; k < 8; ++k) wd[k] = (p.v[k >> 1] >> (16 * (k & 1))) & 0xFFFFu;
    } else {
#pragma unroll
        for (uint32_t k = 0; k < 8u; ++k) wd[k] = w0 + k < nwords ? (uint32_t)wm[w0 + k] : 0x10000u;  // marker: no word
    }
}

__global__ void __launch_bounds__(BND_T) k_wah_tile_sums(const uint8_t* __restrict__ file, const DecBlock* __restrict__ blocks,
                                                          DecLines L) {
    __shared__ uint32_t s_part[BND_T / 64];
    const DecBlock& D = blocks[blockIdx.y];
    const uint32_t c0 = blockIdx.x * BND_TILE;
    if (D.error || D.n_wah == 0 || c0 >= D.wah_words) return;
    const uint16_t* wm = reinterpret_cast<const uint16_t*>(file + D.gt_off + D.off_wah);
    const uint32_t w0 = c0 + threadIdx.x * BND_K * BND_Q;
    static_assert(BND_K == 8, "load_words8");
    uint32_t wd[BND_Q][8];
#pragma unroll
    for (uint32_t q = 0; q < BND_Q; ++q) load_words8(wm, w0 + q * BND_K, D.wah_words, wd[q]);
    uint32_t sum = 0;
#pragma unroll
    for (uint32_t q = 0; q < BND_Q; ++q)
#pragma unroll
        for (uint32_t k = 0; k < BND_K; ++k)
            if (wd[q][k] < 0x10000u) sum += wah_groups_of(wd[q][k]);
    sum = wave_sum(sum);
    if (lane_id() == 0) s_part[threadIdx.x >> 6] = sum;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t t = 0;
        for (uint32_t i = 0; i < BND_T / 64; ++i) t += s_part[i];
        L.tile_sum[(size_t)blockIdx.y * L.max_tiles + blockIdx.x] = t;
    }
}

__global__ void __launch_bounds__(1024) k_wah_tile_scan(const DecBlock* __restrict__ blocks, DecLines L) {
    __shared__ uint64_t s_scan[20];
    const DecBlock& D = blocks[blockIdx.x];
    if (D.error || D.n_wah == 0) return;
    const uint32_t ntiles = (D.wah_words + BND_TILE - 1u) / BND_TILE;
    uint64_t base = 0;
    for (uint32_t t0 = 0; t0 < ntiles; t0 += blockDim.x) {
        const uint32_t t = t0 + threadIdx.x;
        const uint64_t v = t < ntiles ? L.tile_sum[(size_t)blockIdx.x * L.max_tiles + t] : 0u;
        uint64_t tot;
        const uint64_t ex = block_scan_excl64(v, s_scan, &tot);
        if (t < ntiles) L.tile_base[(size_t)blockIdx.x * L.max_tiles + t] = base + ex;
        base += tot;
    }
    if (D.off_line_haploid != VAL_UNDEFINED) {
        // cumulative groups before each WAH line (haploid lines have n_samples bits)
        const uint32_t Gd = (L.N + WAH_BITS - 1u) / WAH_BITS;
        const uint32_t Gh = (L.n_samples + WAH_BITS - 1u) / WAH_BITS;
        uint32_t cb = 0;
        for (uint32_t j0 = 0; j0 < D.n_wah; j0 += blockDim.x) {
            const uint32_t j = j0 + threadIdx.x;
            uint32_t gl = 0;
            if (j < D.n_wah) gl = (L.kind[L.wah_lines[D.wah_first + j]] & KIND_HAPLOID) ? Gh : Gd;
            uint64_t tot;
            const uint64_t ex = block_scan_excl64(gl, s_scan, &tot);
            if (j < D.n_wah) L.wah_cumg[D.wah_first + j] = cb + (uint32_t)ex;
            cb += (uint32_t)tot;
        }
    }
}

// ranges == 0: every tile.  Else (phased decode, no fully haploid lines): part 1 = the tiles that hold the starts of the
// block's first n_wah num / ranges lines - those in front of that many lines' groups -, part 2 = the others.
__global__ void __launch_bounds__(BND_T) k_wah_boundaries(const uint8_t* __restrict__ file,
                                                          const DecBlock* __restrict__ blocks, DecLines L, uint32_t ranges,
                                                          uint32_t num, int part) {
    __shared__ uint64_t s_scan[20];
    const DecBlock& D = blocks[blockIdx.y];
    const uint32_t c0 = blockIdx.x * BND_TILE;
    if (D.error || D.n_wah == 0 || c0 >= D.wah_words) return;
    const uint32_t Gd = (L.N + WAH_BITS - 1u) / WAH_BITS;
    const uint64_t tile_base = L.tile_base[(size_t)blockIdx.y * L.max_tiles + blockIdx.x];
    if (ranges) {
        const bool first = tile_base < (uint64_t)(uint32_t)((uint64_t)D.n_wah * num / ranges) * Gd;
        if (first != (part == 1)) return;
    }
    const uint16_t* wm = reinterpret_cast<const uint16_t*>(file + D.gt_off + D.off_wah);
    const uint32_t nwords = D.wah_words;
    constexpr uint32_t K = BND_K * BND_Q;
    const bool mixed = D.off_line_haploid != VAL_UNDEFINED;
    const uint32_t w0 = c0 + threadIdx.x * K;
    uint32_t g[K];
    uint32_t sum = 0;
    {
        uint32_t wd[BND_Q][8];
#pragma unroll
        for (uint32_t q = 0; q < BND_Q; ++q) load_words8(wm, w0 + q * BND_K, nwords, wd[q]);
#pragma unroll
        for (uint32_t k = 0; k < K; ++k) {
            const uint32_t word = wd[k / BND_K][k % BND_K];
            const uint32_t ng = word < 0x10000u ? wah_groups_of(word) : 0u;
            g[k] = ng;
            sum += ng;
        }
    }
    uint64_t tot;
    uint64_t ex = tile_base + block_scan_excl64(sum, s_scan, &tot);
    // uniform lines: line index and offset inside the line by ONE division per thread, then
    // carried along word by word (a word never spans two lines, so the offset wraps exactly)
    uint64_t jline = 0;
    uint32_t rem = 0;
    if (!mixed && sum) {
        if ((ex >> 32) == 0) {
            jline = (uint32_t)ex / Gd;
            rem = (uint32_t)ex - (uint32_t)jline * Gd;
        } else {
            jline = ex / Gd;
            rem = (uint32_t)(ex - jline * Gd);
        }
    }
#pragma unroll
    for (uint32_t k = 0; k < K; ++k) {
        const uint32_t wi = w0 + k;
        if (wi < nwords && g[k]) {
            if (!mixed) {
                if (rem == 0 && jline < D.n_wah) L.wah_start[D.wah_first + (uint32_t)jline] = wi;
                rem += g[k];
                while (rem >= Gd) {
                    rem -= Gd;
                    ++jline;
                }
            } else {
                // first line whose cumulative offset is >= ex
                uint32_t lo = 0, hi = D.n_wah;
                while (lo < hi) {
                    const uint32_t mid = (lo + hi) >> 1;
                    if ((uint64_t)L.wah_cumg[D.wah_first + mid] < ex)
                        lo = mid + 1u;
                    else
                        hi = mid;
                }
                if (lo < D.n_wah && (uint64_t)L.wah_cumg[D.wah_first + lo] == ex) L.wah_start[D.wah_first + lo] = wi;
            }
        }
        ex += g[k];
    }
}

hipError_t launch_wah_boundaries(hipStream_t s, const uint8_t* file, const DecBlock* blocks, uint32_t n_blocks,
                                 const DecLines& L) {
    if (!n_blocks || !L.max_tiles) return hipSuccess;
    k_wah_tile_sums<<<dim3(L.max_tiles, n_blocks), dim3(BND_T), 0, s>>>(file, blocks, L);
    k_wah_tile_scan<<<dim3(n_blocks), dim3(1024), 0, s>>>(blocks, L);
    k_wah_boundaries<<<dim3(L.max_tiles, n_blocks), dim3(BND_T), 0, s>>>(file, blocks, L, 0u, 0u, 0);
    return hipGetLastError();
}

hipError_t launch_wah_boundaries_part(hipStream_t s, const uint8_t* file, const DecBlock* blocks, uint32_t n_blocks,
                                      const DecLines& L, uint32_t ranges, uint32_t num, int part) {
    if (!n_blocks || !L.max_tiles || !ranges) return hipSuccess;
    if (part == 1) {
        k_wah_tile_sums<<<dim3(L.max_tiles, n_blocks), dim3(BND_T), 0, s>>>(file, blocks, L);
        k_wah_tile_scan<<<dim3(n_blocks), dim3(1024), 0, s>>>(blocks, L);
    }
    k_wah_boundaries<<<dim3(L.max_tiles, n_blocks), dim3(BND_T), 0, s>>>(file, blocks, L, ranges, num, part);
    return hipGetLastError();
}

// sparse line starts: each list is `count, idx[count]` with no index, so the starts are a
// pointer chase (sparse_advance_pointer, accessor_internals_new.hpp:639-653).  One wave per
// block stages the matrix through LDS in 16 KiB tiles and lane 0 hops inside the tile.
__global__ void __launch_bounds__(64) k_sparse_walk(const uint8_t* __restrict__ file, const DecBlock* __restrict__ blocks,
                                                    DecLines L) {
    constexpr uint32_t TILE = 8192;  // uint16 units
    __shared__ uint16_t s_tile[TILE];
    const DecBlock& D = blocks[blockIdx.x];
    if (D.error || D.n_sparse == 0) return;
    const uint16_t* sm = reinterpret_cast<const uint16_t*>(file + D.gt_off + D.off_sparse);
    const uint32_t units = L.aet / 2u;  // uint16 units per A_T
    const uint32_t lane = lane_id();
    uint64_t pos = 0;  // in uint16 units
    uint32_t k = 0;
    uint32_t k_end = D.n_sparse;
    if (L.sp_state) {  // ranged walk (one block)
        k = L.sp_lo < D.n_sparse ? L.sp_lo : D.n_sparse;
        k_end = L.sp_hi < D.n_sparse ? L.sp_hi : D.n_sparse;
        if (k) pos = L.sp_state[0];
    }
    const uint64_t sm_at = D.gt_off + D.off_sparse;
    const uint64_t sm_units = sm_at < L.file_len ? (L.file_len - sm_at) / 2u : 0u;  // never read past the image
    while (k < k_end) {
        const uint64_t t0 = pos;
        for (uint32_t i = lane; i < TILE; i += 64u) s_tile[i] = (t0 + i < sm_units) ? sm[t0 + i] : (uint16_t)0;
        __syncthreads();
        if (lane == 0) {
            while (k < k_end && pos + units <= t0 + TILE) {
                uint32_t num = s_tile[pos - t0];
                if (units == 2u) num |= (uint32_t)s_tile[pos - t0 + 1] << 16;
                num &= (units == 2u) ? 0x7FFFFFFFu : 0x7FFFu;
                L.sparse_start[D.sparse_first + k] = (uint32_t)(pos * 2u);
                pos += (uint64_t)(1u + num) * units;
                ++k;
            }
            s_tile[0] = 0;
        }
        pos = __shfl(pos, 0, 64);
        k = __shfl(k, 0, 64);
        __syncthreads();
    }
    if (L.sp_state && lane == 0) L.sp_state[0] = pos;
}

hipError_t launch_sparse_walk(hipStream_t s, const uint8_t* file, const DecBlock* blocks, uint32_t n_blocks,
                              const DecLines& L) {
    if (!n_blocks) return hipSuccess;
    k_sparse_walk<<<dim3(n_blocks), dim3(64), 0, s>>>(file, blocks, L);
    return hipGetLastError();
}

// expand every WAH line into its permuted bit row (wah2_extract_count_ones, wah.hpp:232-235).
// One wave per line, row built in LDS then streamed out.
__global__ void __launch_bounds__(64) k_wah_expand(const uint8_t* __restrict__ file, const DecBlock* __restrict__ blocks,
                                                   DecLines L, const uint32_t* __restrict__ d_totals,
                                                   const uint32_t* __restrict__ ph_start, const uint32_t* __restrict__ ph_cnt,
                                                   const uint32_t* __restrict__ ph_gpre, uint32_t n_blocks) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t* row = reinterpret_cast<uint32_t*>(smem);
    uint32_t j0 = blockIdx.x * WAH_LINES_PER_WAVE;
    uint32_t total = d_totals[1];
    if (ph_start) {
        // one range of every block's lines (phased decode): group g of the launch -> block b with
        // ph_gpre[b] <= g < ph_gpre[b + 1] (blocks without lines in the range repeat their successor's value)
        const uint32_t g = blockIdx.x;
        uint32_t lo = 0, hi = n_blocks;
        while (hi - lo > 1u) {
            const uint32_t mid = (lo + hi) >> 1;
            if (ph_gpre[mid] <= g) lo = mid;
            else hi = mid;
        }
        const uint32_t first = (g - ph_gpre[lo]) * WAH_LINES_PER_WAVE;
        if (first >= ph_cnt[lo]) return;
        j0 = ph_start[lo] + first;
        const uint32_t end = ph_start[lo] + ph_cnt[lo];
        total = end < total ? end : total;
    }
    if (j0 >= total || d_totals[3]) return;
    const uint32_t lane = lane_id();
    const uint32_t rw = L.y_stride64 * 2u;
    for (uint32_t i = lane; i < rw; i += 64u) row[i] = 0;
    // lane k fetches the metadata of line j0 + k (a chain of four dependent loads, paid once per wave)
    uint32_t m_l = 0, m_nbits = 0, m_maxw = 0;
    uint64_t m_src = 0;
    if (lane < WAH_LINES_PER_WAVE && j0 + lane < total) {
        m_l = L.wah_lines[j0 + lane];
        const DecBlock& D = blocks[L.line_block[m_l]];
        const uint32_t start = L.wah_start[j0 + lane];
        m_nbits = (L.kind[m_l] & KIND_HAPLOID) ? L.n_samples : L.N;
        m_maxw = D.wah_words - start;
        m_src = reinterpret_cast<uint64_t>(file + D.gt_off + D.off_wah) + 2ull * start;
    }
    auto src_of = [&](uint32_t k) {
        const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)m_src, (int)k);
        const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(m_src >> 32), (int)k);
        return reinterpret_cast<const uint16_t*>(((uint64_t)hi << 32) | lo);
    };
    const uint16_t* src = src_of(0);
    uint32_t maxw = (uint32_t)__builtin_amdgcn_readlane((int)m_maxw, 0);
    using GlobU16 = const __attribute__((address_space(1))) uint16_t;  // see wave_wah_expand_row
    uint32_t pre = lane < maxw ? (uint32_t)((GlobU16*)src)[lane] : 0u;
    __syncthreads();
    for (uint32_t k = 0; k < WAH_LINES_PER_WAVE && j0 + k < total; ++k) {
        const uint32_t j = j0 + k;
        const uint32_t l = (uint32_t)__builtin_amdgcn_readlane((int)m_l, (int)k);
        const uint32_t nbits = (uint32_t)__builtin_amdgcn_readlane((int)m_nbits, (int)k);
        // first words of the next line while this one is expanded
        const uint16_t* src_n = src;
        uint32_t maxw_n = 0, pre_n = 0;
        if (k + 1u < WAH_LINES_PER_WAVE && j + 1u < total) {
            src_n = src_of(k + 1u);
            maxw_n = (uint32_t)__builtin_amdgcn_readlane((int)m_maxw, (int)(k + 1u));
            pre_n = (uint32_t)((GlobU16*)src_n)[lane < maxw_n ? lane : 0u];  // unconditional: in flight across this line's expansion
        }
        uint32_t ones;
        (void)wave_wah_expand_row(src, maxw, nbits, row, &ones, pre);
        __syncthreads();
        uint32_t base = 0;
        if (L.yp_compact) {
            // compact form: 64-bit chunks of the row + 16-bit "ones before the chunk" (see DecLines)
            uint2* dc = reinterpret_cast<uint2*>(L.yp) + (size_t)j * L.y_stride64;
            uint16_t* dp = reinterpret_cast<uint16_t*>(reinterpret_cast<uint8_t*>(L.yp) + 8ull * L.y_stride64 * L.yp_rows) +
                           (size_t)j * L.y_stride64;
            // two chunks per lane and step: one 16-byte LDS read / write and one 16-byte store of the chunks, one
            // 4-byte store of their two prefixes (8-byte accesses reach 0.54-0.70 of the 16-byte rate on this chip);
            // rows of an odd number of chunks end with a single one
            const uint32_t pairs = L.y_stride64 >> 1;
            const bool odd = (L.y_stride64 & 1u) != 0u;
            const bool al16 = ((reinterpret_cast<uintptr_t>(dc) | reinterpret_cast<uintptr_t>(dp)) & 15u) == 0u || !odd;
            if (al16 && (reinterpret_cast<uintptr_t>(dc) & 15u) == 0u && (reinterpret_cast<uintptr_t>(dp) & 3u) == 0u) {
                for (uint32_t i0 = 0; i0 < pairs + (odd ? 1u : 0u); i0 += 64u) {
                    const uint32_t i = i0 + lane;
                    uint4 v = make_uint4(0u, 0u, 0u, 0u);
                    if (i < pairs) {
                        uint4* rp = reinterpret_cast<uint4*>(row) + i;
                        v = *rp;
                        *rp = make_uint4(0u, 0u, 0u, 0u);  // ready for the next line
                    } else if (odd && i == pairs) {
                        uint2* rp = reinterpret_cast<uint2*>(row) + 2u * i;
                        const uint2 t = *rp;
                        *rp = make_uint2(0u, 0u);
                        v.x = t.x;
                        v.y = t.y;
                    }
                    const uint32_t c0 = (uint32_t)__popc(v.x) + (uint32_t)__popc(v.y);
                    const uint32_t c = c0 + (uint32_t)__popc(v.z) + (uint32_t)__popc(v.w);
                    const uint32_t inc = wave_scan_incl_dpp(c);
                    const uint32_t p0 = base + inc - c;
                    if (i < pairs) {  // (read once by the chain's staging, a launch later: kept out of the L2's way)
                        typedef uint32_t yc_u32x4 __attribute__((ext_vector_type(4)));
                        __builtin_nontemporal_store(yc_u32x4{v.x, v.y, v.z, v.w}, reinterpret_cast<yc_u32x4*>(dc) + i);
                        __builtin_nontemporal_store((p0 & 0xFFFFu) | ((p0 + c0) << 16), reinterpret_cast<uint32_t*>(dp) + i);
                    } else if (odd && i == pairs) {
                        dc[2u * i] = make_uint2(v.x, v.y);
                        dp[2u * i] = (uint16_t)p0;
                    }
                    base += (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
                }
            } else {
            for (uint32_t i0 = 0; i0 < L.y_stride64; i0 += 64u) {
                const uint32_t i = i0 + lane;
                uint2 v = make_uint2(0u, 0u);
                if (i < L.y_stride64) {
                    uint2* rp = reinterpret_cast<uint2*>(row) + i;  // one 8-byte LDS read and write per chunk
                    v = *rp;
                    *rp = make_uint2(0u, 0u);  // ready for the next line
                }
                const uint32_t c = (uint32_t)__popc(v.x) + (uint32_t)__popc(v.y);
                const uint32_t inc = wave_scan_incl_dpp(c);
                if (i < L.y_stride64) {
                    dc[i] = v;
                    dp[i] = (uint16_t)(base + inc - c);
                }
                base += (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
            }
            }
            if (lane == 0) {
                L.ones[l] = ones;
                L.wah_z[j] = nbits - base;
            }
            __syncthreads();
            src = src_n;
            maxw = maxw_n;
            pre = pre_n;
            continue;
        }
        // {32 row bits, ones before them}: the decode chain's rank-select table for this line
        uint2* dst = L.yp + (size_t)j * L.yp_stride;
        for (uint32_t i0 = 0; i0 < L.yp_stride; i0 += 64u) {
            const uint32_t i = i0 + lane;
            uint32_t v = 0;
            if (i < rw) {
                v = row[i];
                row[i] = 0;  // ready for the next line
            }
            const uint32_t c = (uint32_t)__popc(v);
            const uint32_t inc = wave_scan_incl_dpp(c);
            if (i < L.yp_stride) dst[i] = make_uint2(v, base + inc - c);
            base += (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
        }
        if (lane == 0) {
            L.ones[l] = ones;
            L.wah_z[j] = nbits - base;
        }
        __syncthreads();
        src = src_n;
        maxw = maxw_n;
        pre = pre_n;
    }
}

constexpr int WAH_WIDE_STRIPES = 20;  // 20 x 1024 words: rows up to 655 360 bits
// Rows above 16 KiB by TOGGLES (round 4; see "expansion by TOGGLES" in xsi_device.hpp for the idea): one 1024-thread
// workgroup walks WAH_WIDE_LPG consecutive lines.  (The one-wave-per-line kernel above keeps the row in LDS, so at 500 000
// haplotypes - 62.5 KB - only two of its one-wave workgroups fit a CU: 46 ms per launch; round 3's workgroup-per-line
// kernel that PAINTED every fill into the row took 37.9 ms and is gone: docs/EXPERIMENTS.md.)  Per line: (A) a thread takes four consecutive words of a round of
// 4096 (one 8-byte load; a line of 500 000 bits has some 2500 words, so one round as a rule), the groups in front of
// each word come from one workgroup scan, every word deposits its toggles with at most two LDS atomic XORs - no loop
// over the fills, which is what a painting kernel spends its time in; (B) thread t turns words 2 t, 2 t + 1 of
// every 2048-word stripe into the line's bits (prefix_xor32, a ballot's parity from lane to lane) and counts them AS IF
// its wave started outside a run; a wave that starts inside one has exactly the complement (64 - c ones per lane), so
// one pass of wave 0 over the (stripe, wave) parities and totals settles both the carries and the "ones before" of
// every wave, and the pairs leave as 16-byte stores.  Five barriers a line (a dozen before), the next line's metadata
// and first words already in flight.
constexpr uint32_t WAH_WIDE_LPG = 4;
typedef uint32_t u32_align2 __attribute__((aligned(2)));
__device__ unsigned long long g_wide_prof[8];  // XSI_WIDE_PROF: 100 MHz ticks of wave 0 of workgroup 0 per phase
template <bool PROF>
__global__ void __launch_bounds__(1024) k_wah_expand_wide_t(const uint8_t* __restrict__ file, const DecBlock* __restrict__ blocks,
                                                            DecLines L, const uint32_t* __restrict__ d_totals,
                                                            const uint32_t* __restrict__ ph_start, const uint32_t* __restrict__ ph_cnt,
                                                            const uint32_t* __restrict__ ph_gpre, uint32_t n_blocks) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int STR2 = WAH_WIDE_STRIPES / 2;
    uint32_t* row = reinterpret_cast<uint32_t*>(smem);
    const uint32_t rw = L.y_stride64 * 2u;
    uint32_t* a_tot = row + rw;           // [16] groups per wave of a round of words
    uint32_t* a_ones = a_tot + 16;        // [16] ones per wave (counted like the reference)
    uint32_t* b_info = a_ones + 16;       // [STR2 * 16] per (stripe, wave): ones if the wave starts outside a run | parity << 31
    uint32_t* b_base = b_info + STR2 * 16;  // [STR2 * 16 + 1] ones before the wave | "starts inside a run" << 31; [last] the row's ones
    uint32_t j0 = blockIdx.x * WAH_WIDE_LPG;
    uint32_t total = d_totals[1];
    if (ph_start) {  // one range of every block's lines (phased decode): see k_wah_expand
        const uint32_t g = blockIdx.x;
        uint32_t lo = 0, hi = n_blocks;
        while (hi - lo > 1u) {
            const uint32_t mid = (lo + hi) >> 1;
            if (ph_gpre[mid] <= g) lo = mid;
            else hi = mid;
        }
        const uint32_t first = (g - ph_gpre[lo]) * WAH_WIDE_LPG;
        if (first >= ph_cnt[lo]) return;
        j0 = ph_start[lo] + first;
        const uint32_t end = ph_start[lo] + ph_cnt[lo];
        total = end < total ? end : total;
    }
    if (j0 >= total || d_totals[3]) return;
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const uint32_t w = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6));
    uint64_t t_prof = PROF ? wall_clock64() : 0;
    auto prof = [&](int i) {
        if constexpr (!PROF) return;
        if (blockIdx.x == gridDim.x / 2u && w == 0u) {
            const uint64_t now = wall_clock64();
            if (lane == 0) atomicAdd(&g_wide_prof[i], (unsigned long long)(now - t_prof));
            t_prof = now;
        }
    };
    // lane k of every wave fetches the metadata of line j0 + k (a chain of four dependent loads, paid once per workgroup)
    uint32_t m_l = 0, m_nbits = 0, m_maxw = 0, m_safe = 0;
    uint64_t m_src = 0;
    if (lane < WAH_WIDE_LPG && j0 + lane < total) {
        m_l = L.wah_lines[j0 + lane];
        const DecBlock& D = blocks[L.line_block[m_l]];
        const uint32_t start = L.wah_start[j0 + lane];
        m_nbits = (L.kind[m_l] & KIND_HAPLOID) ? L.n_samples : L.N;
        m_maxw = D.wah_words - start;
        m_src = reinterpret_cast<uint64_t>(file + D.gt_off + D.off_wah) + 2ull * start;
        const uint64_t at = D.gt_off + D.off_wah + 2ull * start;
        const uint64_t sw = at < L.file_len ? (L.file_len - at) / 2u : 0u;
        m_safe = sw < 0xFFFFFFFFull ? (uint32_t)sw : 0xFFFFFFFFu;
        if (sw < 4u) {  // fewer than 8 bytes between the line's start and the image's end (a truncated or hostile image): the
            m_src = reinterpret_cast<uint64_t>(file);  // fallback 8-byte load of load4 would run past the end - it reads the
            m_safe = 0;                                 // header instead, and the line has no words
        }
    }
    using GlobU16W = const __attribute__((address_space(1))) uint16_t;
    using GlobU32A2 = const __attribute__((address_space(1))) u32_align2;
    auto src_of = [&](uint32_t k) -> GlobU16W* {
        const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)m_src, (int)k);
        const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(m_src >> 32), (int)k);
        return (GlobU16W*)(((uint64_t)hi << 32) | lo);  // global, not flat (see wave_wah_expand_row)
    };
    // The four words 4 t .. 4 t + 3 of the round that starts at word `wbase`, as ONE unconditional 8-byte load that
    // stays in flight until unpack4 (a load under a branch, or selected against a default, is waited for where it is
    // issued).  Words behind a line's last one are never used (their groups lie beyond the line), so the load may run
    // past the block's WAH matrix - but not past the image: `safe` = words from the line's first to the image's end
    // (a thread whose four words would cross that end reads the line's first words again and reports "no word").
    auto load4 = [&](GlobU16W* src, uint32_t wbase, uint32_t safe, uint32_t (&raw)[2]) {
        const uint32_t w4 = wbase + 4u * tid;
        const GlobU32A2* q = (const GlobU32A2*)(src + (w4 + 4u <= safe ? w4 : 0u));  // 2-byte aligned: gfx950 global loads take that
        raw[0] = q[0];
        raw[1] = q[1];
    };
    auto unpack4 = [&](const uint32_t (&raw)[2], uint32_t wbase, uint32_t max_words, uint32_t safe, uint32_t (&wd)[4]) {
        const uint32_t w4 = wbase + 4u * tid;
        const bool ok = w4 + 4u <= safe;
        wd[0] = ok && w4 < max_words ? raw[0] & 0xFFFFu : 0x10000u;  // 0x10000 = no word
        wd[1] = ok && w4 + 1u < max_words ? raw[0] >> 16 : 0x10000u;
        wd[2] = ok && w4 + 2u < max_words ? raw[1] & 0xFFFFu : 0x10000u;
        wd[3] = ok && w4 + 3u < max_words ? raw[1] >> 16 : 0x10000u;
    };
    for (uint32_t i = tid; i < rw; i += 1024u) row[i] = 0;
    GlobU16W* src = src_of(0);
    uint32_t maxw = (uint32_t)__builtin_amdgcn_readlane((int)m_maxw, 0);
    uint32_t safe = (uint32_t)__builtin_amdgcn_readlane((int)m_safe, 0);
    if (maxw > safe) maxw = safe;  // (a block that claims words beyond the image: parse_blocks flags it; never read them)
    uint32_t raw[2];
    load4(src, 0u, safe, raw);
    __syncthreads();
    prof(0);  // metadata, first words, row cleared
    const uint32_t stripes = (L.yp_stride + 2047u) / 2048u;  // <= STR2
    for (uint32_t k = 0; k < WAH_WIDE_LPG && j0 + k < total; ++k) {
        const uint32_t j = j0 + k;
        const uint32_t l = (uint32_t)__builtin_amdgcn_readlane((int)m_l, (int)k);
        const uint32_t nbits = (uint32_t)__builtin_amdgcn_readlane((int)m_nbits, (int)k);
        const uint32_t G = (nbits + WAH_BITS - 1u) / WAH_BITS;
        // ---- A: words -> toggles
        uint32_t gbase = 0, cnt1 = 0;
        for (uint32_t wbase = 0; wbase < maxw && gbase < G; wbase += 4096u) {
            // the next round's words travel while this one is scanned (the first lines of a block, hardly sorted yet,
            // are literal after literal: up to nine rounds); unconditional - the last round fetches what follows the line
            uint32_t wd[4], rawn[2];
            unpack4(raw, wbase, maxw, safe, wd);
            load4(src, wbase + 4096u, safe, rawn);
            uint32_t ng[4], tsum = 0;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                ng[q] = wd[q] < 0x10000u ? wah_groups_of(wd[q]) : 0u;
                tsum += ng[q];
            }
            const uint32_t inc = wave_scan_incl_dpp(tsum);
            if (lane == 63u) a_tot[w] = inc;
            __syncthreads();
            const uint32_t sc = row16_scan_incl(lane < 16u ? a_tot[lane] : 0u);
            const uint32_t wave_base = w ? (uint32_t)__builtin_amdgcn_readlane((int)sc, (int)w - 1) : 0u;
            const uint32_t round_groups = (uint32_t)__builtin_amdgcn_readlane((int)sc, 15);
            uint32_t sg = gbase + wave_base + inc - tsum;  // first group covered by my first word
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                cnt1 += wah_word_toggles(wd[q] & 0xFFFFu, sg, ng[q], ng[q] != 0u && sg < G, row, rw);
                sg += ng[q];
            }
            gbase += round_groups;
            raw[0] = rawn[0];
            raw[1] = rawn[1];
            __syncthreads();  // a_tot is reused by the next round; the toggles are in the row
        }
        {
            const uint32_t wsum = wave_sum(cnt1);
            if (lane == 0) a_ones[w] = wsum;
        }
        prof(1);  // A: scan + toggles
        // the next line's metadata and first words travel while this line's row is formed
        if (k + 1u < WAH_WIDE_LPG && j + 1u < total) {
            src = src_of(k + 1u);
            maxw = (uint32_t)__builtin_amdgcn_readlane((int)m_maxw, (int)(k + 1u));
            safe = (uint32_t)__builtin_amdgcn_readlane((int)m_safe, (int)(k + 1u));
            if (maxw > safe) maxw = safe;
            load4(src, 0u, safe, raw);
        }
        // ---- B1: toggles -> bits as if every wave started outside a run, back into the row; ones and parity per wave
        for (uint32_t i = 0; i < stripes; ++i) {
            const uint32_t idx = i * 2048u + 2u * tid;
            const bool in = idx < rw;  // rw is even; the in-range lanes of a wave are its first ones
            uint2 t = make_uint2(0u, 0u);
            if (in) t = *reinterpret_cast<const uint2*>(row + idx);
            const uint32_t p0 = prefix_xor32(t.x);
            const uint32_t p1 = prefix_xor32(t.y) ^ (uint32_t)((int32_t)p0 >> 31);
            const uint64_t M = __ballot((int32_t)p1 < 0);
            const uint32_t before = __builtin_amdgcn_mbcnt_hi((uint32_t)(M >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)M, 0u));
            const uint32_t flip = 0u - (before & 1u);
            const uint32_t a0 = in ? p0 ^ flip : 0u, a1 = in ? p1 ^ flip : 0u;
            if (in) *reinterpret_cast<uint2*>(row + idx) = make_uint2(a0, a1);
            const uint32_t csum = wave_sum((uint32_t)__popc(a0) + (uint32_t)__popc(a1));
            if (lane == 0u) b_info[i * 16u + w] = csum | (((uint32_t)__popcll(M) & 1u) << 31);
        }
        prof(2);  // B1
        __syncthreads();
        prof(3);  // barrier behind B1
        if (w == 0) {
            // entries in row order: idx = stripe * 16 + wave.  carry into an entry = parity of the toggles in front of it;
            // its ones = S0 outside a run, 32 * (its words inside the row) - S0 inside one
            constexpr int PER = (STR2 * 16 + 63) / 64;
            uint32_t tot[PER], par[PER];
            uint32_t lane_par = 0;
#pragma unroll
            for (int q = 0; q < PER; ++q) {
                const uint32_t idx = lane * (uint32_t)PER + (uint32_t)q;
                const uint32_t v = idx < stripes * 16u ? b_info[idx] : 0u;
                tot[q] = v & 0x7FFFFFFFu;
                par[q] = v >> 31;
                lane_par ^= par[q];
            }
            const uint64_t M = __ballot(lane_par != 0u);
            uint32_t cin = __builtin_amdgcn_mbcnt_hi((uint32_t)(M >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)M, 0u)) & 1u;
            uint32_t sum = 0, inside[PER];
#pragma unroll
            for (int q = 0; q < PER; ++q) {
                const uint32_t idx = lane * (uint32_t)PER + (uint32_t)q;
                inside[q] = cin;
                if (cin) {
                    const uint32_t first = (idx >> 4) * 2048u + (idx & 15u) * 128u;
                    const uint32_t words = first >= rw ? 0u : (rw - first < 128u ? rw - first : 128u);
                    tot[q] = words * 32u - tot[q];
                }
                sum += tot[q];
                cin ^= par[q];
            }
            uint32_t run = wave_scan_incl_dpp(sum) - sum;
#pragma unroll
            for (int q = 0; q < PER; ++q) {
                const uint32_t idx = lane * (uint32_t)PER + (uint32_t)q;
                if (idx < stripes * 16u) b_base[idx] = run | (inside[q] << 31);
                run += tot[q];
            }
            if (lane == 63u) {
                b_base[STR2 * 16] = run;
                L.wah_z[j] = nbits - run;
                uint32_t ones = 0;
                for (int i = 0; i < 16; ++i) ones += a_ones[i];
                L.ones[l] = ones;
            }
        }
        prof(4);  // wave 0's pass
        __syncthreads();
        prof(5);  // barrier behind it
        // ---- B2: the waves' carries and bases are known: final words, "ones before", 16-byte stores; the row is left zero
        uint4* dst = reinterpret_cast<uint4*>(L.yp + (size_t)j * L.yp_stride);
        const uint32_t row_ones = b_base[STR2 * 16];
        for (uint32_t i = 0; i < stripes; ++i) {
            const uint32_t idx = i * 2048u + 2u * tid;
            const bool in = idx < rw;
            uint2 a = make_uint2(0u, 0u);
            if (in) {
                a = *reinterpret_cast<const uint2*>(row + idx);
                *reinterpret_cast<uint2*>(row + idx) = make_uint2(0u, 0u);  // ready for the next line
            }
            const uint32_t info = b_base[i * 16u + w];
            const uint32_t flip = in ? 0u - (info >> 31) : 0u;
            const uint32_t f0 = a.x ^ flip, f1 = a.y ^ flip;
            const uint32_t c = (uint32_t)__popc(f0) + (uint32_t)__popc(f1);
            const uint32_t inc = wave_scan_incl_dpp(c);
            const uint32_t pre0 = in ? (info & 0x7FFFFFFFu) + inc - c : row_ones;
            if (idx < L.yp_stride) {
                typedef uint32_t yp_u32x4 __attribute__((ext_vector_type(4)));
                yp_u32x4 o;
                if (L.yp_rev)  // for k_chain_decode_rank_big<.., REV>: bits reversed, minus the ones up to the word's end
                    o = yp_u32x4{__brev(f0), 0u - (pre0 + (uint32_t)__popc(f0)), __brev(f1), 0u - (pre0 + c)};
                else
                    o = yp_u32x4{f0, pre0, f1, pre0 + (uint32_t)__popc(f0)};
                __builtin_nontemporal_store(o, reinterpret_cast<yp_u32x4*>(dst) + (idx >> 1));
            }
        }
        prof(6);  // B2
        // (b_info / b_base / a_ones of the next line are written behind at least one more barrier)
    }
}

static bool wah_expand_is_wide(const DecLines& L) {
    return L.y_stride64 * 8u > 16384u && L.yp_stride <= 1024u * (uint32_t)WAH_WIDE_STRIPES && !tuning_env("XSI_NO_WIDE_EXPAND");
}
bool wah_expand_wide(const DecLines& L) { return wah_expand_is_wide(L); }

uint32_t wah_expand_lines_per_group(const DecLines& L) { return wah_expand_is_wide(L) ? WAH_WIDE_LPG : WAH_LINES_PER_WAVE; }

// ph_start == nullptr: all WAH lines of the batch (n_groups ignored); else one range of every block's lines
static hipError_t launch_wah_expand_any(hipStream_t s, const uint8_t* file, const DecBlock* blocks, const DecLines& L,
                                        uint32_t max_wah, const uint32_t* d_totals, const uint32_t* ph_start,
                                        const uint32_t* ph_cnt, const uint32_t* ph_gpre, uint32_t n_blocks, uint32_t n_groups) {
    const uint32_t lds = L.y_stride64 * 8u;
    if (wah_expand_is_wide(L)) {
        const uint32_t lds_w = lds + 4u * (32u + 2u * (uint32_t)(WAH_WIDE_STRIPES / 2) * 16u + 1u);
        const bool prof = tuning_env("XSI_WIDE_PROF") != nullptr;
        auto kern = prof ? &k_wah_expand_wide_t<true> : &k_wah_expand_wide_t<false>;
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_w);
        if (e != hipSuccess) return e;
        const uint32_t grid = ph_start ? n_groups : (max_wah + WAH_WIDE_LPG - 1u) / WAH_WIDE_LPG;
        kern<<<dim3(grid), dim3(1024), lds_w, s>>>(file, blocks, L, d_totals, ph_start, ph_cnt, ph_gpre, n_blocks);
        if (prof) {
            unsigned long long pr[8] = {};
            (void)hipStreamSynchronize(s);
            (void)hipMemcpyFromSymbol(pr, HIP_SYMBOL(g_wide_prof), sizeof(pr));
            fprintf(stderr, "[xsi wide prof] cumulative us of wave 0 of the middle workgroup: start %.1f, A %.1f, B1 %.1f, barrier %.1f, wave-0 pass %.1f, barrier %.1f, B2 %.1f\n",
                    pr[0] * 1e-2, pr[1] * 1e-2, pr[2] * 1e-2, pr[3] * 1e-2, pr[4] * 1e-2, pr[5] * 1e-2, pr[6] * 1e-2);
        }
        return hipGetLastError();
    }
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_wah_expand),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    const uint32_t grid = ph_start ? n_groups : (max_wah + WAH_LINES_PER_WAVE - 1u) / WAH_LINES_PER_WAVE;
    k_wah_expand<<<dim3(grid), dim3(64), lds, s>>>(file, blocks, L, d_totals, ph_start, ph_cnt, ph_gpre, n_blocks);
    return hipGetLastError();
}

hipError_t launch_wah_expand(hipStream_t s, const uint8_t* file, const DecBlock* blocks, const DecLines& L,
                             uint32_t max_wah, const uint32_t* d_totals) {
    if (!max_wah) return hipSuccess;
    return launch_wah_expand_any(s, file, blocks, L, max_wah, d_totals, nullptr, nullptr, nullptr, 0u, 0u);
}

hipError_t launch_wah_expand_phase(hipStream_t s, const uint8_t* file, const DecBlock* blocks, const DecLines& L,
                                   const uint32_t* d_totals, const uint32_t* ph_start, const uint32_t* ph_cnt,
                                   const uint32_t* ph_gpre, uint32_t n_blocks, uint32_t n_groups) {
    if (!n_groups) return hipSuccess;
    return launch_wah_expand_any(s, file, blocks, L, 0u, d_totals, ph_start, ph_cnt, ph_gpre, n_blocks, n_groups);
}

// allele counts without expansion (fill_allele_counts_advance, accessor_internals_new.hpp:407-438):
// WAH lines by popcount over their words (wah2_advance_pointer_count_ones, wah.hpp:125-150),
// sparse lines from their count field (sparse_advance_pointer, :639-653).
__global__ void __launch_bounds__(256) k_wah_count(const uint8_t* __restrict__ file, const DecBlock* __restrict__ blocks,
                                                   DecLines L, const uint32_t* __restrict__ d_totals) {
    const uint32_t j = blockIdx.x * 4u + (threadIdx.x >> 6);
    if (j >= d_totals[1] || d_totals[3]) return;
    const uint32_t l = L.wah_lines[j];
    const DecBlock& D = blocks[L.line_block[l]];
    const uint32_t nbits = (L.kind[l] & KIND_HAPLOID) ? L.n_samples : L.N;
    const uint32_t start = L.wah_start[j];
    const uint16_t* src = reinterpret_cast<const uint16_t*>(file + D.gt_off + D.off_wah) + start;
    uint32_t ones;
    (void)wave_wah_expand_row(src, D.wah_words - start, nbits, nullptr, &ones);
    if (lane_id() == 0) L.ones[l] = ones;
}

__global__ void __launch_bounds__(256) k_sparse_count(const uint8_t* __restrict__ file,
                                                      const DecBlock* __restrict__ blocks, DecLines L,
                                                      const uint32_t* __restrict__ d_totals) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= d_totals[2] || d_totals[3]) return;
    const uint32_t l = L.sparse_lines[k];
    const DecBlock& D = blocks[L.line_block[l]];
    const uint32_t nbits = (L.kind[l] & KIND_HAPLOID) ? L.n_samples : L.N;
    const uint64_t at = D.gt_off + D.off_sparse + L.sparse_start[k];
    uint32_t num = 0;
    if (at + L.aet <= L.file_len) num = rd_at(file + at, L.aet);
    const uint32_t msb = (L.aet == 2u) ? 0x8000u : 0x80000000u;
    const bool neg = (num & msb) != 0u;
    num &= ~msb;
    L.ones[l] = neg ? nbits - num : num;
    if (neg) L.kind[l] |= KIND_NEGATED;
}

hipError_t launch_line_counts(hipStream_t s, const uint8_t* file, const DecBlock* blocks, const DecLines& L,
                              uint32_t max_wah, uint32_t max_sparse, const uint32_t* d_totals) {
    if (max_wah) {
        k_wah_count<<<dim3((max_wah + 3u) / 4u), dim3(256), 0, s>>>(file, blocks, L, d_totals);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    if (max_sparse) {
        k_sparse_count<<<dim3((max_sparse + 255u) / 256u), dim3(256), 0, s>>>(file, blocks, L, d_totals);
        return hipGetLastError();
    }
    return hipSuccess;
}

// sparse lines -> bit rows (sparse_extract + the fill loops of fill_genotype_array_advance,
// accessor_internals_new.hpp:208-219, 619-637).  apply_negation: write the ALT bit row of a
// negated bi-allelic line (complement of the listed REF positions); otherwise the raw listed
// positions are written and KIND_NEGATED is left for the composer.
template <int T>  // threads per line: one wave for short rows; 1024 for long ones, whose LDS copy leaves room for two
                   // workgroups per CU only (one wave each: 100 ms per launch at 500 000 haplotypes)
__global__ void __launch_bounds__(T) k_sparse_fill(const uint8_t* __restrict__ file, const DecBlock* __restrict__ blocks,
                                                    DecLines L, const uint32_t* __restrict__ d_totals,
                                                    uint32_t* __restrict__ out_rows, uint32_t out_stride_w,
                                                    int apply_negation) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t* row = reinterpret_cast<uint32_t*>(smem);
    const uint32_t k = blockIdx.x + (L.sp_state ? L.sp_lo : 0u);
    if (k >= d_totals[2] || d_totals[3]) return;
    const uint32_t l = L.sparse_lines[k];
    const DecBlock& D = blocks[L.line_block[l]];
    const uint32_t nbits = (L.kind[l] & KIND_HAPLOID) ? L.n_samples : L.N;
    const uint32_t nw = (nbits + 31u) >> 5;
    const uint32_t lane = threadIdx.x;
    for (uint32_t i = lane; i < out_stride_w; i += (uint32_t)T) row[i] = 0;
    __syncthreads();
    const uint64_t at = D.gt_off + D.off_sparse + L.sparse_start[k];
    const uint8_t* p = file + at;
    const bool in_file = at <= L.file_len && L.file_len - at >= L.aet;  // corrupt image: never read past it
    uint32_t num = in_file ? rd_at(p, L.aet) : 0u;
    const uint32_t msb = (L.aet == 2u) ? 0x8000u : 0x80000000u;
    const bool neg = (num & msb) != 0u;
    num &= ~msb;
    {
        const uint64_t room = in_file ? (L.file_len - at) / L.aet - 1u : 0u;
        if (num > room) num = (uint32_t)room;
    }
    for (uint32_t i = lane; i < num; i += (uint32_t)T) {
        const uint32_t idx = rd_at(p + (size_t)(1u + i) * L.aet, L.aet);
        // Raw lists (no complement here: the composer reads them) keep EVERY listed position of the row: the reference
        // applies a sparse list entry by entry whatever the line's haploid flag (accessor_internals_new.hpp:208-256), and
        // that flag can belong to another line (KEY_LINE_HAPLOID is written per BCF line and read per binary line, SURVEY
        // 9.6.2) - a diploid line's second ALT read as "haploid" still lists haplotypes up to N.
        if (idx < (apply_negation ? nbits : L.N)) atomicOr(&row[idx >> 5], 1u << (idx & 31u));
    }
    __syncthreads();
    uint32_t* dst = out_rows + (size_t)l * out_stride_w;
    for (uint32_t i = lane; i < out_stride_w; i += (uint32_t)T) {
        uint32_t v = row[i];
        if (neg && apply_negation) {
            v = (i < nw) ? ~v : 0u;
            if (i == nw - 1u && (nbits & 31u)) v &= (1u << (nbits & 31u)) - 1u;
        }
        dst[i] = v;
    }
    if (lane == 0) {
        L.ones[l] = neg ? nbits - num : num;  // sparse_extract: ones = negated ? N - num : num
        if (neg) L.kind[l] |= KIND_NEGATED;
    }
}

// Long rows (above 16 KiB): no copy of the row in LDS.  The workgroup stores the row's background (zeros, or ones up to
// nbits for a negated line that is to be complemented) with 16-byte stores, and behind a barrier the thread that holds
// the FIRST list entry of a 32-bit word stores that word whole (the list is ascending, block.hpp:59-65: the entries of
// a word are neighbours).  The LDS form held two 1024-thread workgroups per CU: 38.6 ms per launch at 500 000
// haplotypes, 0.4 TB/s of row stores.  Rows are whole 16-byte units on 16-byte addresses (checked by the launcher).
__global__ void __launch_bounds__(256) k_sparse_fill_direct(const uint8_t* __restrict__ file, const DecBlock* __restrict__ blocks,
                                                            DecLines L, const uint32_t* __restrict__ d_totals,
                                                            uint32_t* __restrict__ out_rows, uint32_t out_stride_w,
                                                            int apply_negation) {
    const uint32_t k = blockIdx.x + (L.sp_state ? L.sp_lo : 0u);
    if (k >= d_totals[2] || d_totals[3]) return;
    const uint32_t l = L.sparse_lines[k];
    const DecBlock& D = blocks[L.line_block[l]];
    const uint32_t nbits = (L.kind[l] & KIND_HAPLOID) ? L.n_samples : L.N;
    const uint32_t nw = (nbits + 31u) >> 5;
    const uint32_t tid = threadIdx.x;
    const uint64_t at = D.gt_off + D.off_sparse + L.sparse_start[k];
    const uint8_t* p = file + at;
    const bool in_file = at <= L.file_len && L.file_len - at >= L.aet;  // corrupt image: never read past it
    uint32_t num = in_file ? rd_at(p, L.aet) : 0u;
    const uint32_t msb = (L.aet == 2u) ? 0x8000u : 0x80000000u;
    const bool neg = (num & msb) != 0u;
    num &= ~msb;
    {
        const uint64_t room = in_file ? (L.file_len - at) / L.aet - 1u : 0u;
        if (num > room) num = (uint32_t)room;
    }
    const bool ones_bg = neg && apply_negation;
    const uint32_t last_mask = (nbits & 31u) ? (1u << (nbits & 31u)) - 1u : ~0u;
    auto background = [&](uint32_t w) -> uint32_t { return !ones_bg || w >= nw ? 0u : (w == nw - 1u ? last_mask : ~0u); };
    uint32_t* dst = out_rows + (size_t)l * out_stride_w;
    {
        uint4* dst4 = reinterpret_cast<uint4*>(dst);
        for (uint32_t q = tid; q < out_stride_w / 4u; q += 256u)
            dst4[q] = make_uint4(background(4u * q), background(4u * q + 1u), background(4u * q + 2u), background(4u * q + 3u));
    }
    __syncthreads();  // the background stores have been waited for: a word stored below lands behind them
    for (uint32_t i = tid; i < num; i += 256u) {
        const uint32_t idx = rd_at(p + (size_t)(1u + i) * L.aet, L.aet);
        const uint32_t w = idx >> 5;
        if (i && (rd_at(p + (size_t)i * L.aet, L.aet) >> 5) == w) continue;  // not the first entry of its word
        // raw lists keep every listed position of the row (see k_sparse_fill); only a complemented row ends at nbits
        const uint32_t lim = apply_negation ? nbits : L.N;
        const uint32_t nwl = (lim + 31u) >> 5;
        if (w >= nwl) continue;  // corrupt image: positions beyond the row are dropped
        uint32_t v = 1u << (idx & 31u);
        for (uint32_t j = i + 1u; j < num; ++j) {
            const uint32_t nx = rd_at(p + (size_t)(1u + j) * L.aet, L.aet);
            if ((nx >> 5) != w) break;
            v |= 1u << (nx & 31u);
        }
        if (w == nwl - 1u && (lim & 31u)) v &= (1u << (lim & 31u)) - 1u;
        dst[w] = ones_bg ? (background(w) & ~v) : v;
    }
    if (tid == 0) {
        L.ones[l] = neg ? nbits - num : num;  // sparse_extract: ones = negated ? N - num : num
        if (neg) L.kind[l] |= KIND_NEGATED;
    }
}

hipError_t launch_sparse_fill(hipStream_t s, const uint8_t* file, const DecBlock* blocks, const DecLines& L,
                              uint32_t max_sparse, const uint32_t* d_totals, uint32_t* out_rows,
                              uint32_t out_stride_w, int apply_negation) {
    if (!max_sparse) return hipSuccess;
    const uint32_t lds = out_stride_w * 4u;
    if (lds > 16384u && (out_stride_w & 3u) == 0u && (reinterpret_cast<uintptr_t>(out_rows) & 15u) == 0u &&
        !tuning_env("XSI_SPARSE_FILL_LDS")) {
        k_sparse_fill_direct<<<dim3(max_sparse), dim3(256), 0, s>>>(file, blocks, L, d_totals, out_rows, out_stride_w, apply_negation);
        return hipGetLastError();
    }
    if (lds > 16384u) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_sparse_fill<1024>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        k_sparse_fill<1024><<<dim3(max_sparse), dim3(1024), lds, s>>>(file, blocks, L, d_totals, out_rows, out_stride_w,
                                                                     apply_negation);
        return hipGetLastError();
    }
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_sparse_fill<64>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    k_sparse_fill<64><<<dim3(max_sparse), dim3(64), lds, s>>>(file, blocks, L, d_totals, out_rows, out_stride_w,
                                                          apply_negation);
    return hipGetLastError();
}

// ==========================================================================================
// Synthetic haplotype matrix (definition in DESIGN.md §"Synthetic workload"; numpy mirror in
// xsqueezeit_amd/synth.py).  Integer-only, splitmix64-based, so host and device agree exactly.
// ==========================================================================================
__host__ __device__ __forceinline__ uint64_t mix64(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

constexpr uint32_t SYN_FOUNDERS = 256;
constexpr uint32_t SYN_SEG = 4096;
constexpr uint32_t SYN_MUT = 1u << 20;  // 2^32 / 4096

__global__ void __launch_bounds__(256) k_synth_packed(uint64_t seed, uint64_t first_line, uint64_t n_lines,
                                                      uint32_t n_haps, uint32_t* __restrict__ bits, uint32_t stride_w) {
    const uint64_t row = blockIdx.y;
    const uint32_t w = blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= n_lines || w >= stride_w) return;
    const uint32_t nw = (n_haps + 31u) >> 5;
    uint32_t out = 0;
    if (w < nw) {
        const uint64_t site = first_line + row;
        const uint64_t hs = mix64(seed * 0x9E3779B97F4A7C15ull + site + 1ull);
        uint32_t nb = 0;
        while ((1ull << nb) < (uint64_t)n_haps) ++nb;  // bit length of n_haps-1 (n_haps >= 2)
        if (nb == 0) nb = 1;
        const uint32_t e = (uint32_t)(hs & 0xFFFFull) % nb;
        uint64_t k = (1ull << e) + ((hs >> 16) & ((1ull << e) - 1ull));
        if (k > (uint64_t)n_haps - 1ull) k = (uint64_t)n_haps - 1ull;
        const uint32_t t32 = (uint32_t)((k << 32) / (uint64_t)n_haps);
        const bool rare = k * SYN_FOUNDERS < 4ull * n_haps;
        for (uint32_t b = 0; b < 32u; ++b) {
            const uint64_t h = (uint64_t)w * 32u + b;
            if (h >= n_haps) break;
            uint32_t bit;
            if (rare) {
                const uint64_t r = mix64(hs ^ (h * 0xD1B54A32D192ED03ull + 1ull));
                bit = (uint32_t)(r >> 32) < t32;
            } else {
                const uint64_t off = mix64(seed ^ (h * 0xD1B54A32D192ED03ull + 7ull)) % SYN_SEG;
                const uint64_t seg = (site + off) / SYN_SEG;
                const uint64_t g = mix64(seed + h * 0x9E3779B97F4A7C15ull + seg * 0xC2B2AE3D27D4EB4Full) % SYN_FOUNDERS;
                const uint64_t r = mix64(hs ^ (g * 0xD1B54A32D192ED03ull + 0x51EDull));
                bit = (uint32_t)(r >> 32) < t32;
                const uint64_t r2 = mix64(hs ^ (h * 0xD1B54A32D192ED03ull + 0xABCDull));
                if ((uint32_t)(r2 >> 32) < SYN_MUT) bit ^= 1u;
            }
            out |= bit << b;
        }
    }
    bits[row * stride_w + w] = out;
}

hipError_t launch_synth_packed(hipStream_t s, uint64_t seed, uint64_t first_line, uint64_t n_lines, uint32_t n_haps,
                               uint32_t* bits, uint32_t stride_w) {
    if (!n_lines) return hipSuccess;
    // grid.y is limited to 65535: loop in slabs
    const uint64_t slab = 65535;
    for (uint64_t r0 = 0; r0 < n_lines; r0 += slab) {
        const uint64_t n = (n_lines - r0 < slab) ? n_lines - r0 : slab;
        k_synth_packed<<<dim3((stride_w + 255u) / 256u, (uint32_t)n), dim3(256), 0, s>>>(
            seed, first_line + r0, n, n_haps, bits + r0 * stride_w, stride_w);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

}  // namespace xsi
