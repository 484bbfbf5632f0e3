// xsi_gt.hip — general genotype path: htslib int32 rows <-> bit planes, plus the missing /
// end-of-vector / phase side channels.  The heavy per-line work (PBWT chain, WAH16, sparse
// lists) is shared with the packed path; this file only adds the conversion kernels around it.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/xsi_hip.h"
#include "xsi_ctx.hpp"
#include "xsi_device.hpp"
#include "xsi_kernels.hpp"

using namespace xsi;

#define HIP_TRY(expr)                                                                                        \
    do {                                                                                                     \
        hipError_t _e = (expr);                                                                              \
        if (_e != hipSuccess) return set_error(XSI_ERR_HIP, "%s: %s (%s:%d)", #expr, hipGetErrorString(_e), \
                                               __FILE__, __LINE__);                                          \
    } while (0)
#define WS(ptr, name, bytes)                          \
    do {                                              \
        void* _p;                                     \
        int _rc = ws_ensure(ctx, name, (bytes), &_p); \
        if (_rc) return _rc;                          \
        ptr = reinterpret_cast<decltype(ptr)>(_p);    \
    } while (0)

namespace xsi {

constexpr int32_t GT_INT32_MISSING = (int32_t)0x80000000;
constexpr int32_t GT_VECTOR_END = (int32_t)0x80000001;

// ------------------------------------------------------------------------------------------
// unpack: one BCF line of int32 genotypes -> bit planes and counts (scan_genotypes,
// gt_block.hpp:207-269, and the predicates of gt_block.hpp:76-100 / wah.hpp:431-435).
// One workgroup of 4 waves per BCF line; every wave turns 64 consecutive values into one
// ballot per plane, so reads are 256-byte coalesced and planes are written 8 bytes at a time.
//   alt planes  : allele == k              (k = 1 .. n_allele-1), one per binary line
//   ref plane   : allele == 0              (listed by negated sparse lines)
//   miss / eov  : MissingPred / EndOfVectorPred
//   phase       : odd index && phase bit != default (ploidy-2 lines only)
// ------------------------------------------------------------------------------------------
struct UnpackArgs {
    const int32_t* gt;
    uint64_t gt_stride;
    const uint32_t* bcf_nbits;
    const uint32_t* bcf_n_allele;
    const uint32_t* bcf_first_bin;
    uint32_t n_samples;
    int32_t default_phased;
    uint32_t stride_w;
    uint32_t* planes;      // [n_bin]
    uint32_t* ref_planes;  // [n_bcf]
    uint32_t* miss_planes;
    uint32_t* eov_planes;
    uint32_t* phase_planes;
    uint32_t* cnt;         // [n_bin]
    uint32_t* ref_cnt;     // [n_bcf]
    uint32_t* miss_cnt;
    uint32_t* eov_cnt;
    uint32_t* bcf_flags;
    uint8_t* kind;         // [n_bin] receives KIND_HAPLOID
    uint32_t* d_error;     // set to 1 on an allele outside [0, n_allele)
    uint32_t quad_ok;      // rows start on 16 bytes and are whole 16-byte units apart: bi-allelic lines take unpack_quads
};

// acc = 2 acc + (x == K): a compare and an add-with-carry, the condition never leaves the condition code
template <uint32_t K>
__device__ __forceinline__ void acc_eq(uint32_t& acc, uint32_t x) {
    asm("v_cmp_eq_u32 vcc, %2, %1\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(acc) : "v"(x), "i"(K) : "vcc");
}

// Bi-allelic lines, lane-local: a lane takes FOUR CONSECUTIVE values per load (16 bytes per lane, 1 KiB per wave and
// instruction, coalesced) and collects each plane's bits of its own values in a register - no ballot, no v_writelane
// (the ballot form below spends 40 vector instructions per 64 values and is bound by them, 3.6 TB/s; this one 12).
// A super-group is 8 segments of 256 values: afterwards a lane holds, per plane, the nibbles of its 4 values of every
// segment; an 8 x 8 nibble transpose inside each group of 8 lanes (three butterfly steps) turns that into whole 32-bit
// words of the plane rows: lane 8 g + i ends up with word g of segment i.
// A value is valid when exactly one of {ref, alt, missing, end of vector} claims it: no per-value range check.
struct QuadCounts {
    uint32_t ref, miss, eov, alt, any_phase, bad;
};
__device__ __forceinline__ void unpack_quads(const UnpackArgs& U, const int32_t* row, uint32_t ngt, bool diploid,
                                             uint32_t first_group, uint32_t nchunks, uint32_t lane, uint32_t* p_ref, uint32_t* p_miss,
                                             uint32_t* p_eov, uint32_t* p_ph, uint32_t* p_alt, QuadCounts& C) {
    const uint32_t i8 = lane & 7u;
    // butterfly constants of my lane: which nibbles I keep (K) and how far the partner's word is rotated right
    const uint32_t K1 = (i8 & 1u) ? 0xF0F0F0F0u : 0x0F0F0F0Fu, R1 = (i8 & 1u) ? 4u : 28u;
    const uint32_t K2 = (i8 & 2u) ? 0xFF00FF00u : 0x00FF00FFu, R2 = (i8 & 2u) ? 8u : 24u;
    const uint32_t K4 = (i8 & 4u) ? 0xFFFF0000u : 0x0000FFFFu, R4 = (i8 & 4u) ? 16u : 16u;
    auto transpose = [&](uint32_t x) -> uint32_t {
        uint32_t p = (uint32_t)__builtin_amdgcn_ds_swizzle((int)x, 0x041F);  // lane ^ 1
        x = (x & K1) | (__builtin_amdgcn_alignbit(p, p, R1) & ~K1);
        p = (uint32_t)__builtin_amdgcn_ds_swizzle((int)x, 0x081F);           // lane ^ 2
        x = (x & K2) | (__builtin_amdgcn_alignbit(p, p, R2) & ~K2);
        p = (uint32_t)__builtin_amdgcn_ds_swizzle((int)x, 0x101F);           // lane ^ 4
        x = (x & K4) | (__builtin_amdgcn_alignbit(p, p, R4) & ~K4);
        return x;
    };
    typedef int32_t i32x4 __attribute__((ext_vector_type(4)));
    const uint32_t limit = ngt;
    const uint32_t ph_flip = U.default_phased ? 0x55555555u : 0u;    // (odd values: bits 2 and 0 of a nibble collected top-first)
    // the transpose leaves word 8 i8 + g in lane 8 g + i8; one lane permutation more and lane L holds word L of the
    // super-group: whole 256-byte stores (as 64 scattered dwords the five plane stores doubled the kernel's requests)
    const uint32_t from_lane = ((lane & 7u) << 3 | (lane >> 3)) << 2;  // ds_bpermute address of the lane whose word I store
    i32x4 q[8];
    auto load_segment = [&](uint32_t V0, uint32_t sg) -> i32x4 {
        const uint32_t first = V0 + 256u * sg + 4u * lane;
        // (unconditional: beyond the line the quad at its start is read again and masked out below)
        return *reinterpret_cast<const i32x4*>(row + (first < ngt ? first : 0u));
    };
    // The four waves of the workgroup take the row's super-groups in turn: the workgroup reads its row as ONE forward
    // stream (a contiguous quarter of the row per wave made four streams a row, thousands on the chip: 3.9 TB/s).
    // Eight loads stay in flight: a segment's registers are refilled with the same segment of my next super-group as
    // soon as its values have been collected.
#pragma unroll
    for (uint32_t sg = 0; sg < 8u; ++sg) q[sg] = load_segment(first_group * 2048u, sg);
    for (uint32_t V0 = first_group * 2048u; V0 < nchunks * 64u; V0 += 4u * 2048u) {
        uint32_t a_ref = 0, a_alt = 0, a_eov = 0, a_m0 = 0, a_m1 = 0, a_ph = 0;
#pragma unroll
        for (uint32_t sg = 0; sg < 8u; ++sg) {
            if (V0 + 256u * sg >= limit) {  // wave-uniform: nothing of this segment is in the line
                a_ref <<= 4; a_alt <<= 4; a_eov <<= 4; a_m0 <<= 4; a_m1 <<= 4; a_ph <<= 4;
            } else {
#pragma unroll
                for (uint32_t j = 0; j < 4u; ++j) {
                    const uint32_t v = (uint32_t)q[sg][j];
                    const uint32_t t = (uint32_t)((int32_t)v >> 1);
                    acc_eq<1u>(a_ref, t);
                    acc_eq<2u>(a_alt, t);
                    acc_eq<0u>(a_m0, t);
                    acc_eq<0x80000000u>(a_m1, v);
                    acc_eq<0x80000001u>(a_eov, v);
                }
                a_ph = (a_ph << 4) | (((uint32_t)q[sg][1] & 1u) << 2) | ((uint32_t)q[sg][3] & 1u);
            }
            q[sg] = load_segment(V0 + 4u * 2048u, sg);
        }
        // bit 4 s + j of a collected word = value j of my quad in segment s (collected top-first: reverse); what lies
        // beyond the line or beyond my wave's values is masked
        uint32_t inm = ~0u;
        if (V0 + 2048u > limit) {  // wave-uniform
            inm = 0;
#pragma unroll
            for (uint32_t sg = 0; sg < 8u; ++sg) {
                const uint32_t first = V0 + 256u * sg + 4u * lane;
                const uint32_t nv = limit > first ? (limit - first < 4u ? limit - first : 4u) : 0u;
                inm |= ((1u << nv) - 1u) << (4u * sg);
            }
        }
        const uint32_t w_ref = __brev(a_ref) & inm, w_alt = __brev(a_alt) & inm, w_eov = __brev(a_eov) & inm;
        const uint32_t w_miss = __brev(a_m0 | a_m1) & inm;
        const uint32_t w_ph = diploid ? (__brev(a_ph ^ ph_flip) & inm) : 0u;
        C.bad |= (w_ref | w_alt | w_eov | w_miss) ^ inm;
        C.ref += (uint32_t)__popc(w_ref);
        C.alt += (uint32_t)__popc(w_alt);
        C.eov += (uint32_t)__popc(w_eov);
        C.miss += (uint32_t)__popc(w_miss);
        C.any_phase |= w_ph;
        const uint32_t word = V0 / 32u + lane;
        auto place = [&](uint32_t x) { return (uint32_t)__builtin_amdgcn_ds_bpermute((int)from_lane, (int)transpose(x)); };
        const uint32_t t_ref = place(w_ref), t_alt = place(w_alt), t_eov = place(w_eov);
        const uint32_t t_miss = place(w_miss), t_ph = place(w_ph);
        if (word < nchunks * 2u) {
            p_ref[word] = t_ref;
            p_alt[word] = t_alt;
            p_eov[word] = t_eov;
            p_miss[word] = t_miss;
            p_ph[word] = t_ph;
        }
    }
}

__global__ void __launch_bounds__(256) k_unpack_gt(UnpackArgs U) {
    __shared__ uint32_t s_cnt[4];  // ref, missing, eov, phase-any
    __shared__ uint32_t s_alt[64];
    const uint32_t l = blockIdx.x;
    const uint32_t ngt = U.bcf_nbits[l];
    const uint32_t n_allele = U.bcf_n_allele[l];
    const uint32_t b0 = U.bcf_first_bin[l];
    const bool diploid = ngt == 2u * U.n_samples;
    const uint32_t lane = lane_id();
    const uint32_t w = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));  // wave index, uniform
    if (threadIdx.x < 4) s_cnt[threadIdx.x] = 0;
    if (threadIdx.x < 64) s_alt[threadIdx.x] = 0;
    __syncthreads();
    const int32_t* row = U.gt + (size_t)l * U.gt_stride;
    const uint32_t nchunks = U.stride_w / 2u;  // 64-bit words per plane row (all written, pad = 0)
    uint32_t c_ref = 0, c_miss = 0, c_eov = 0, any_phase = 0, c_alt1 = 0, maxc = 0;
    // Each wave owns a contiguous quarter of the row.  The 64-bit ballot words of up to 64 chunks are
    // parked one per lane (v_writelane) and leave as one coalesced store per plane; the int32 loads
    // of four chunks are issued before the first one is used.
    const uint32_t cpw = (nchunks + 3u) / 4u;
    const uint32_t c_begin = w * cpw < nchunks ? w * cpw : nchunks;
    const uint32_t c_end = (w + 1u) * cpw < nchunks ? (w + 1u) * cpw : nchunks;
    uint64_t* p_ref = reinterpret_cast<uint64_t*>(U.ref_planes + (size_t)l * U.stride_w);
    uint64_t* p_miss = reinterpret_cast<uint64_t*>(U.miss_planes + (size_t)l * U.stride_w);
    uint64_t* p_eov = reinterpret_cast<uint64_t*>(U.eov_planes + (size_t)l * U.stride_w);
    uint64_t* p_ph = reinterpret_cast<uint64_t*>(U.phase_planes + (size_t)l * U.stride_w);
    uint64_t* p_alt = reinterpret_cast<uint64_t*>(U.planes + (size_t)b0 * U.stride_w);  // first ALT plane
    const bool quads = U.quad_ok && n_allele == 2u;  // uniform over the workgroup
    if (quads) {
        QuadCounts C{0, 0, 0, 0, 0, 0};
        unpack_quads(U, row, ngt, diploid, /*first super-group*/ w, /*chunks of the row*/ nchunks, lane, reinterpret_cast<uint32_t*>(p_ref), reinterpret_cast<uint32_t*>(p_miss),
                     reinterpret_cast<uint32_t*>(p_eov), reinterpret_cast<uint32_t*>(p_ph), reinterpret_cast<uint32_t*>(p_alt), C);
        c_ref = wave_sum(C.ref);
        c_miss = wave_sum(C.miss);
        c_eov = wave_sum(C.eov);
        c_alt1 = wave_sum(C.alt);
        any_phase = __builtin_amdgcn_ballot_w64(C.any_phase != 0u) ? 1u : 0u;
        if (C.bad) *U.d_error = 1;  // "Unknown allele error !": a value no plane claims
    } else
    for (uint32_t g0 = c_begin; g0 < c_end; g0 += 64u) {
        const uint32_t gn = c_end - g0 < 64u ? c_end - g0 : 64u;  // chunks in this group
        uint32_t a_ref[2] = {0, 0}, a_miss[2] = {0, 0}, a_eov[2] = {0, 0}, a_ph[2] = {0, 0}, a_alt[2] = {0, 0};
        // The int32 loads of the NEXT four chunks are in flight while the current four are worked on.  They are
        // unconditional (clamped index; `in` below discards the value): with a default value under a condition the
        // compiler waits for each load where it is issued.
        int32_t vv[4], vn[4];
        auto issue = [&](uint32_t q0, int32_t (&dst)[4]) {
#pragma unroll
            for (uint32_t u = 0; u < 4u; ++u) {
                const uint32_t i = (g0 + q0 + u) * 64u + lane;
                dst[u] = row[(q0 + u < gn && i < ngt) ? i : 0u];
            }
        };
        issue(0u, vv);
        for (uint32_t q0 = 0; q0 < gn; q0 += 4u) {
            issue(q0 + 4u, vn);
#pragma unroll
            for (uint32_t u = 0; u < 4u; ++u) {
                if (q0 + u >= gn) break;  // wave-uniform
                const uint32_t cgi = g0 + q0 + u;
                const uint32_t i = cgi * 64u + lane;
                const bool in = i < ngt;
                const int32_t v = vv[u];
                // One code per value - 0 missing, 1 end of vector, allele + 2 when called, ~0 beyond the line - so
                // that every plane is a ballot of ONE vector compare (a ballot of a combination of conditions is
                // lowered through a select and a second compare).
                const uint32_t t = (uint32_t)(v >> 1);
                uint32_t code = (t < 0x7FFFFFFFu ? t : 0x7FFFFFFFu) + 1u;  // saturating: a negative allele number cannot wrap into 0 / 1
                code = (t == 0u || v == GT_INT32_MISSING) ? 0u : code;
                code = (v == GT_VECTOR_END) ? 1u : code;  // (its t is 0xC0000000: not "missing")
                code = in ? code : ~0u;
                const bool called = code >= 2u && in;
                const int32_t allele = (int32_t)code - 2;
                maxc = in ? (code > maxc ? code : maxc) : maxc;  // "Unknown allele error !": an allele outside [0, n_allele)
                const uint32_t phx = (in && diploid && (i & 1u)) ? (((uint32_t)v & 1u) ^ (uint32_t)U.default_phased) : 0u;
                const uint64_t m_ref = __builtin_amdgcn_ballot_w64(code == 2u);
                const uint64_t m_miss = __builtin_amdgcn_ballot_w64(code == 0u);
                const uint64_t m_eov = __builtin_amdgcn_ballot_w64(code == 1u);
                const uint64_t m_ph = __builtin_amdgcn_ballot_w64(phx != 0u);
                const uint32_t slot = q0 + u;
                a_ref[0] = write_lane(a_ref[0], (uint32_t)m_ref, slot);
                a_ref[1] = write_lane(a_ref[1], (uint32_t)(m_ref >> 32), slot);
                a_miss[0] = write_lane(a_miss[0], (uint32_t)m_miss, slot);
                a_miss[1] = write_lane(a_miss[1], (uint32_t)(m_miss >> 32), slot);
                a_eov[0] = write_lane(a_eov[0], (uint32_t)m_eov, slot);
                a_eov[1] = write_lane(a_eov[1], (uint32_t)(m_eov >> 32), slot);
                a_ph[0] = write_lane(a_ph[0], (uint32_t)m_ph, slot);
                a_ph[1] = write_lane(a_ph[1], (uint32_t)(m_ph >> 32), slot);
                c_ref += (uint32_t)__popcll(m_ref);
                c_miss += (uint32_t)__popcll(m_miss);
                c_eov += (uint32_t)__popcll(m_eov);
                any_phase |= m_ph ? 1u : 0u;
                // first ALT allele through the lane-parked path, further ALT alleles directly
                {
                    const uint64_t m = __builtin_amdgcn_ballot_w64(code == 3u);
                    a_alt[0] = write_lane(a_alt[0], (uint32_t)m, slot);
                    a_alt[1] = write_lane(a_alt[1], (uint32_t)(m >> 32), slot);
                    c_alt1 += (uint32_t)__popcll(m);
                }
                for (uint32_t k = 2; k < n_allele; ++k) {
                    const uint64_t m = __builtin_amdgcn_ballot_w64(called && allele == (int32_t)k);
                    if (lane == 0) {
                        reinterpret_cast<uint64_t*>(U.planes + (size_t)(b0 + k - 1u) * U.stride_w)[cgi] = m;
                        if (m) atomicAdd(&s_alt[(k - 1u) & 63u], (uint32_t)__popcll(m));
                    }
                }
            }
#pragma unroll
            for (uint32_t u = 0; u < 4u; ++u) vv[u] = vn[u];
        }
        if (lane < gn) {
            p_ref[g0 + lane] = ((uint64_t)a_ref[1] << 32) | a_ref[0];
            p_miss[g0 + lane] = ((uint64_t)a_miss[1] << 32) | a_miss[0];
            p_eov[g0 + lane] = ((uint64_t)a_eov[1] << 32) | a_eov[0];
            p_ph[g0 + lane] = ((uint64_t)a_ph[1] << 32) | a_ph[0];
            p_alt[g0 + lane] = ((uint64_t)a_alt[1] << 32) | a_alt[0];
        }
    }
    if (maxc >= n_allele + 2u) *U.d_error = 1;  // (a negative allele number shows as a huge code)
    if (lane == 0) {
        if (c_alt1) atomicAdd(&s_alt[0], c_alt1);
        atomicAdd(&s_cnt[0], c_ref);
        atomicAdd(&s_cnt[1], c_miss);
        atomicAdd(&s_cnt[2], c_eov);
        atomicOr(&s_cnt[3], any_phase);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        U.ref_cnt[l] = s_cnt[0];
        U.miss_cnt[l] = s_cnt[1];
        U.eov_cnt[l] = s_cnt[2];
        U.bcf_flags[l] = (s_cnt[1] ? 1u : 0u) | (s_cnt[2] ? 2u : 0u) | (s_cnt[3] ? 4u : 0u) | (diploid ? 0u : 8u);
    }
    for (uint32_t k = 1 + threadIdx.x; k < n_allele; k += blockDim.x) {
        U.kind[b0 + k - 1u] = diploid ? 0 : (uint8_t)KIND_HAPLOID;
    }
    // alt counts: up to 64 alleles share the LDS counters; beyond that recount from the planes
    for (uint32_t k = 1 + w; k < n_allele; k += 4u) {
        uint32_t c;
        if (n_allele <= 65u) {
            c = s_alt[k - 1u];
        } else {
            const uint32_t* pr = U.planes + (size_t)(b0 + k - 1u) * U.stride_w;
            c = 0;
            for (uint32_t i = lane; i < U.stride_w; i += 64u) c += (uint32_t)__popc(pr[i]);
            c = wave_sum(c);
        }
        if (lane == 0) U.cnt[b0 + k - 1u] = c;
    }
}

// flag vectors of the side channels: one bit per binary line, set on the first binary line of
// a flagged BCF line (reindex_binary_vector_from_bcf_to_binary_lines, gt_block.hpp:650-666);
// the haploid vector has one bit per BCF line (gt_block.hpp:219-224).
__global__ void __launch_bounds__(256) k_side_flagbits(const EncBlock* __restrict__ blocks, EncSide S,
                                                       uint32_t* __restrict__ flagbits) {
    const uint32_t b = blockIdx.x;
    const EncBlock& B = blocks[b];
    constexpr uint32_t FW = MAX_BIN_PER_BLOCK / 32;
    uint32_t* base = flagbits + (size_t)b * FV_COUNT * FW;
    for (uint32_t i = threadIdx.x; i < 4u * FW; i += blockDim.x) base[FW + i] = 0;  // FV_MISSING..FV_HAPLOID
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < B.n_bcf; i += blockDim.x) {
        const uint32_t li = B.first_bcf + i;
        const uint32_t f = S.bcf_flags[li];
        const uint32_t rel = S.bcf_first_bin[li] - B.first_bin;
        if (f & 1u) atomicOr(&base[FV_MISSING * FW + (rel >> 5)], 1u << (rel & 31u));
        if (f & 2u) atomicOr(&base[FV_EOV * FW + (rel >> 5)], 1u << (rel & 31u));
        if (f & 4u) atomicOr(&base[FV_PHASE * FW + (rel >> 5)], 1u << (rel & 31u));
        if (f & 8u) atomicOr(&base[FV_HAPLOID * FW + (i >> 5)], 1u << (i & 31u));
    }
}

// sizes of every flagged line's side-channel entries.  One wave per BCF line.
__global__ void __launch_bounds__(256) k_side_sizes(EncSide S) {
    const uint32_t l = blockIdx.x * 4u + (threadIdx.x >> 6);
    if (l >= S.n_bcf) return;
    const uint32_t f = S.bcf_flags[l];
    const uint32_t nbits = S.bcf_nbits[l];
    uint32_t ms = 0, es = 0, pl = 0;
    if (f & 1u) {
        if (S.strategy == WS_SPARSE)
            ms = (1u + S.miss_cnt[l]) * S.aet;
        else
            ms = 2u * wave_wah_encode_row<false>(S.miss_planes + (size_t)l * S.plane_stride_w, nbits, nullptr);
    }
    if (f & 2u) {
        if (S.strategy == WS_SPARSE)
            es = (1u + S.eov_cnt[l]) * S.aet;
        else
            es = 2u * wave_wah_encode_row<false>(S.eov_planes + (size_t)l * S.plane_stride_w, nbits, nullptr);
    }
    if (f & 4u) pl = wave_wah_encode_row<false>(S.phase_planes + (size_t)l * S.plane_stride_w, nbits, nullptr);
    if (lane_id() == 0) {
        S.miss_size[l] = ms;
        S.eov_size[l] = es;
        S.phase_len[l] = pl;
    }
}

__device__ __forceinline__ void store_at16(uint8_t* p, uint32_t v, uint32_t aet) {
    uint16_t* q = reinterpret_cast<uint16_t*>(p);
    q[0] = (uint16_t)v;
    if (aet == 4u) q[1] = (uint16_t)(v >> 16);
}

// Sparse<A_T, Pred> list of a plane (block.hpp:54-76), no MSB flag.
__device__ __forceinline__ void wave_list_emit(const uint32_t* __restrict__ row, uint32_t nbits, uint32_t aet,
                                               uint8_t* __restrict__ dst) {
    const uint32_t lane = lane_id();
    const uint32_t nw = (nbits + 31u) >> 5;
    uint32_t base = 0;
    for (uint32_t w0 = 0; w0 < nw; w0 += 64u) {
        const uint32_t w = w0 + lane;
        uint32_t v = 0;
        if (w < nw) {
            v = row[w];
            if (w == nw - 1u && (nbits & 31u)) v &= (1u << (nbits & 31u)) - 1u;
        }
        const uint32_t c = (uint32_t)__popc(v);
        const uint32_t inc = wave_scan_incl(c);
        uint32_t pos = base + inc - c;
        while (v) {
            const uint32_t bpos = (uint32_t)__ffs((int)v) - 1u;
            v &= v - 1u;
            store_at16(dst + (size_t)(1u + pos) * aet, w * 32u + bpos, aet);
            ++pos;
        }
        base += __shfl(inc, 63, 64);
    }
    if (lane == 0) store_at16(dst, base, aet);
}

// write the side-channel matrices (gt_block.hpp:565-629).  One wave per BCF line.
__global__ void __launch_bounds__(256) k_side_write(const EncBlock* __restrict__ blocks,
                                                    const uint32_t* __restrict__ line_block, EncSide S,
                                                    uint8_t* __restrict__ out, const uint64_t* __restrict__ d_result) {
    if (d_result[3]) return;
    const uint32_t l = blockIdx.x * 4u + (threadIdx.x >> 6);
    if (l >= S.n_bcf) return;
    const uint32_t f = S.bcf_flags[l];
    if (!(f & 7u)) return;
    const EncBlock& B = blocks[line_block[S.bcf_first_bin[l]]];
    const uint32_t nbits = S.bcf_nbits[l];
    uint8_t* gt = out + B.out_off + 16u;
    if (f & 1u) {
        uint8_t* dst = gt + B.off_miss + S.miss_off[l];
        const uint32_t* row = S.miss_planes + (size_t)l * S.plane_stride_w;
        if (S.strategy == WS_SPARSE)
            wave_list_emit(row, nbits, S.aet, dst);
        else
            (void)wave_wah_encode_row<true>(row, nbits, reinterpret_cast<uint16_t*>(dst));
    }
    if (f & 2u) {
        uint8_t* dst = gt + B.off_eov + S.eov_off[l];
        const uint32_t* row = S.eov_planes + (size_t)l * S.plane_stride_w;
        if (S.strategy == WS_SPARSE)
            wave_list_emit(row, nbits, S.aet, dst);
        else
            (void)wave_wah_encode_row<true>(row, nbits, reinterpret_cast<uint16_t*>(dst));
    }
    if (f & 4u) {
        uint16_t* dst = reinterpret_cast<uint16_t*>(gt + B.off_phase) + S.phase_off[l];
        (void)wave_wah_encode_row<true>(S.phase_planes + (size_t)l * S.plane_stride_w, nbits, dst);
    }
}

int encode_side_write(xsi_hip_ctx* ctx, const EncBlock* d_blocks, uint32_t n_blocks, const EncLines& L,
                      const EncSide& S, uint8_t* out, const uint64_t* d_result) {
    (void)n_blocks;
    if (!S.n_bcf) return XSI_OK;
    k_side_write<<<dim3((S.n_bcf + 3u) / 4u), dim3(256), 0, ctx->stream>>>(d_blocks, L.line_block, S, out, d_result);
    HIP_TRY(hipGetLastError());
    return XSI_OK;
}

// ------------------------------------------------------------------------------------------
// decode side: flag vectors -> per-binary-line bits; pointer walk over the side matrices;
// planes of the flagged lines; int32 composer.
// ------------------------------------------------------------------------------------------
struct DecSide {
    uint8_t* side;          // per binary line: bit0 missing, bit1 eov, bit2 phase (set on the BCF line's first binary line)
    uint32_t* miss_start;   // per binary line: byte offset (rel. GT block) of the line's missing entry
    uint32_t* eov_start;
    uint32_t* phase_start;
    uint32_t* miss_planes;  // per binary line (rows of flagged first binary lines are filled)
    uint32_t* eov_planes;
    uint32_t* phase_planes;
    uint32_t* n_miss;       // per binary line
    uint32_t* n_eov;
    uint32_t stride_w;
    // ranged walk (the accessor's prefix decode, one block): binary lines [bin_lo, bin_hi) of the block; the cursors behind
    // line bin_hi - 1 are left in walk_state[0..3) and picked up from there when bin_lo > 0.  walk_state == nullptr: all lines.
    uint32_t bin_lo, bin_hi;
    uint64_t* walk_state;
};

__global__ void __launch_bounds__(64) k_dec_side_flags(const uint8_t* __restrict__ file,
                                                       const DecBlock* __restrict__ blocks, DecLines L, DecSide S) {
    __shared__ uint32_t s_row[3][MAX_BIN_PER_BLOCK / 32 + 2];
    const DecBlock& D = blocks[blockIdx.x];
    if (D.error) return;
    const uint32_t lane = lane_id();
    const uint32_t nb = D.n_bin, nw = (nb + 31u) >> 5;
    for (uint32_t v = 0; v < 3; ++v)
        for (uint32_t i = lane; i < nw + 1u; i += 64u) s_row[v][i] = 0;
    __syncthreads();
    const uint32_t offs[3] = {D.off_line_missing, D.off_line_eov, D.off_line_phase};
    for (uint32_t v = 0; v < 3; ++v) {
        if (offs[v] == VAL_UNDEFINED) continue;
        const uint64_t at = D.gt_off + offs[v];
        uint32_t maxw = 0;
        if (at < L.file_len) {
            const uint64_t left = (L.file_len - at) / 2u;
            maxw = left < FLAG_WORDS_MAX ? (uint32_t)left : FLAG_WORDS_MAX;
        }
        uint32_t ones;
        (void)wave_wah_expand_row(reinterpret_cast<const uint16_t*>(file + at), maxw, nb, s_row[v], &ones);
    }
    __syncthreads();
    for (uint32_t i = lane; i < nb; i += 64u) {
        uint32_t f = 0;
        for (uint32_t v = 0; v < 3; ++v) f |= ((s_row[v][i >> 5] >> (i & 31u)) & 1u) << v;
        S.side[D.first_bin + i] = (uint8_t)f;
    }
}

// Sequential cursor walk of the side matrices (weirdness_advance / phase_advance,
// accessor_internals_new.hpp:478-546): entry starts per flagged binary line.  Lane 0 of one wave
// per block; side channels are rare so no tiling.
__global__ void __launch_bounds__(64) k_dec_side_walk(const uint8_t* __restrict__ file,
                                                      const DecBlock* __restrict__ blocks, DecLines L, DecSide S) {
    const DecBlock& D = blocks[blockIdx.x];
    if (D.error || lane_id() != 0) return;
    const uint8_t* gt = file + D.gt_off;
    const uint64_t room = L.file_len - D.gt_off;
    const uint32_t msb = (L.aet == 2u) ? 0x8000u : 0x80000000u;
    auto rd = [&](uint64_t off) -> uint32_t {
        if (off + L.aet > room) return 0u;
        const uint16_t* q = reinterpret_cast<const uint16_t*>(gt + off);
        uint32_t v = q[0];
        if (L.aet == 4u) v |= (uint32_t)q[1] << 16;
        return v;
    };
    auto wah_skip = [&](uint64_t off, uint32_t nbits) -> uint64_t {  // wah2_advance_pointer, wah.hpp:158-174
        uint32_t pos = 0;
        while (pos < nbits && off + 2u <= room) {
            const uint32_t word = *reinterpret_cast<const uint16_t*>(gt + off);
            pos += (word & 0x8000u) ? (word & WAH_MAXC) * WAH_BITS : WAH_BITS;
            off += 2u;
        }
        return off;
    };
    uint64_t pm = (D.strategy == WS_SPARSE) ? D.off_miss_sparse : D.off_miss_wah;
    uint64_t pe = (D.strategy == WS_SPARSE) ? D.off_eov_sparse : D.off_eov_wah;
    uint64_t pp = D.off_phase;
    uint32_t i_lo = 0, i_hi = D.n_bin;
    if (S.walk_state) {
        i_lo = S.bin_lo < D.n_bin ? S.bin_lo : D.n_bin;
        i_hi = S.bin_hi < D.n_bin ? S.bin_hi : D.n_bin;
        if (i_lo) {
            pm = S.walk_state[0];
            pe = S.walk_state[1];
            pp = S.walk_state[2];
        }
    }
    for (uint32_t i = i_lo; i < i_hi; ++i) {
        const uint32_t l = D.first_bin + i;
        const uint32_t f = S.side[l];
        const uint32_t nbits = (L.kind[l] & KIND_HAPLOID) ? L.n_samples : L.N;
        if ((f & 1u) && D.off_line_missing != VAL_UNDEFINED && pm != VAL_UNDEFINED) {
            S.miss_start[l] = (uint32_t)pm;
            if (D.strategy == WS_SPARSE)
                pm += (uint64_t)(1u + (rd(pm) & ~msb)) * L.aet;
            else
                pm = wah_skip(pm, nbits);
        }
        if ((f & 2u) && D.off_line_eov != VAL_UNDEFINED && pe != VAL_UNDEFINED) {
            S.eov_start[l] = (uint32_t)pe;
            if (D.strategy == WS_SPARSE)
                pe += (uint64_t)(1u + (rd(pe) & ~msb)) * L.aet;
            else
                pe = wah_skip(pe, nbits);
        }
        if ((f & 4u) && pp != VAL_UNDEFINED) {
            S.phase_start[l] = (uint32_t)pp;
            pp = wah_skip(pp, nbits);
        }
    }
    if (S.walk_state) {
        S.walk_state[0] = pm;
        S.walk_state[1] = pe;
        S.walk_state[2] = pp;
    }
}

// planes of one flagged line's side channels.  One wave per binary line, row built in LDS.
__global__ void __launch_bounds__(64) k_dec_side_planes(const uint8_t* __restrict__ file,
                                                        const DecBlock* __restrict__ blocks, DecLines L, DecSide S,
                                                        uint32_t n_bin) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t* row = reinterpret_cast<uint32_t*>(smem);
    const uint32_t l = blockIdx.x + (S.walk_state ? S.bin_lo : 0u);  // ranged: the grid covers [bin_lo, bin_hi) of one block
    if (l >= n_bin) return;
    const uint32_t f = S.side[l];
    const uint32_t lane = lane_id();
    if (lane == 0) {
        S.n_miss[l] = 0;
        S.n_eov[l] = 0;
    }
    if (!(f & 7u)) return;
    const DecBlock& D = blocks[L.line_block[l]];
    const uint8_t* gt = file + D.gt_off;
    const uint64_t room = L.file_len - D.gt_off;
    const uint32_t nbits = (L.kind[l] & KIND_HAPLOID) ? L.n_samples : L.N;
    const uint32_t msb = (L.aet == 2u) ? 0x8000u : 0x80000000u;
    for (uint32_t ch = 0; ch < 3; ++ch) {
        if (!(f & (1u << ch))) continue;
        uint32_t* dst = (ch == 0 ? S.miss_planes : ch == 1 ? S.eov_planes : S.phase_planes) + (size_t)l * S.stride_w;
        const uint32_t start = ch == 0 ? S.miss_start[l] : ch == 1 ? S.eov_start[l] : S.phase_start[l];
        for (uint32_t i = lane; i < S.stride_w; i += 64u) row[i] = 0;
        __syncthreads();
        uint32_t count = 0;
        const bool as_list = (ch < 2u) && D.strategy == WS_SPARSE;
        if (as_list) {
            const uint16_t* q = reinterpret_cast<const uint16_t*>(gt + start);
            uint32_t num = q[0];
            if (L.aet == 4u) num |= (uint32_t)q[1] << 16;
            num &= ~msb;
            const uint64_t fit = (start + (uint64_t)L.aet <= room) ? (room - start) / L.aet - 1u : 0u;
            if (num > fit) num = (uint32_t)fit;
            for (uint32_t i = lane; i < num; i += 64u) {
                const uint16_t* e = reinterpret_cast<const uint16_t*>(gt + start + (size_t)(1u + i) * L.aet);
                uint32_t idx = e[0];
                if (L.aet == 4u) idx |= (uint32_t)e[1] << 16;
                if (idx < nbits) atomicOr(&row[idx >> 5], 1u << (idx & 31u));
            }
            count = num;
        } else {
            const uint64_t left = (room - start) / 2u;
            (void)wave_wah_expand_row(reinterpret_cast<const uint16_t*>(gt + start),
                                      left > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)left, nbits, row, &count);
        }
        __syncthreads();
        for (uint32_t i = lane; i < S.stride_w; i += 64u) dst[i] = row[i];
        if (lane == 0) {
            if (ch == 0) S.n_miss[l] = count;
            if (ch == 1) S.n_eov[l] = count;
        }
        __syncthreads();
    }
}

// WS_PBWT_WAH (the version-4 default; gt_block.hpp:340-395 writes it, accessor_internals_new.hpp:300-340 and 503-533 read
// it): the missing / end-of-vector lines of a block are stored permuted by a_weirdness, which starts as the identity
// and is partitioned by "missing or end of vector" after every such line.  k_dec_side_planes has expanded the lines
// as stored; this kernel replays the block's weird lines in order: plane[a_w[i]] = stored[i], then the stable
// partition of a_w (bool_pbwt_sort / bool_pbwt_sort_two, gt_block.hpp:124-151).  One workgroup per block, thread t
// owns positions [t K, t K + K) as k_arrangement_at does; weird lines are rare, nothing here is on the hot path.
// Blocks with fully haploid lines are refused by the host (the reference encodes those lines through a1 but reads
// them through a_weird: not decodable as written).
__global__ void __launch_bounds__(1024) k_dec_side_unpermute(const DecBlock* __restrict__ blocks, DecLines L, DecSide S,
                                                             uint32_t* __restrict__ a_scratch, uint32_t* __restrict__ tmp) {
    __shared__ uint64_t scan_lds[17];
    const DecBlock& D = blocks[blockIdx.x];
    if (D.error || D.strategy != WS_PBWT_WAH) return;
    if (D.off_line_missing == VAL_UNDEFINED && D.off_line_eov == VAL_UNDEFINED) return;
    const uint32_t N = L.N, tid = threadIdx.x;
    const uint32_t K = (N + 1023u) / 1024u;
    const uint32_t lo = tid * K < N ? tid * K : N, hi = lo + K < N ? lo + K : N;
    uint32_t* cur = a_scratch + (size_t)blockIdx.x * 2u * N;
    uint32_t* nxt = cur + N;
    uint32_t* tm = tmp + (size_t)blockIdx.x * 2u * S.stride_w;
    uint32_t* te = tm + S.stride_w;
    for (uint32_t i = lo; i < hi; ++i) cur[i] = i;
    __syncthreads();
    for (uint32_t k = 0; k < D.n_bin; ++k) {
        const uint32_t l = D.first_bin + k;
        const uint32_t f = S.side[l] & 3u;
        if (!f) continue;
        uint32_t* ym = S.miss_planes + (size_t)l * S.stride_w;
        uint32_t* ye = S.eov_planes + (size_t)l * S.stride_w;
        for (uint32_t i = tid; i < 2u * S.stride_w; i += 1024u) tm[i] = 0;
        __threadfence_block();
        __syncthreads();
        uint32_t zc = 0;
        for (uint32_t i = lo; i < hi; ++i) {
            const uint32_t h = cur[i];
            const uint32_t bm = (f & 1u) ? (ym[i >> 5] >> (i & 31u)) & 1u : 0u;
            const uint32_t be = (f & 2u) ? (ye[i >> 5] >> (i & 31u)) & 1u : 0u;
            if (bm) atomicOr(&tm[h >> 5], 1u << (h & 31u));
            if (be) atomicOr(&te[h >> 5], 1u << (h & 31u));
            zc += 1u - (bm | be);
        }
        uint64_t Z;
        const uint32_t zbase = (uint32_t)block_scan_excl64(zc, scan_lds, &Z);
        uint32_t zpos = zbase, opos = (uint32_t)Z + (lo - zbase);
        for (uint32_t i = lo; i < hi; ++i) {
            const uint32_t h = cur[i];
            const uint32_t bm = (f & 1u) ? (ym[i >> 5] >> (i & 31u)) & 1u : 0u;
            const uint32_t be = (f & 2u) ? (ye[i >> 5] >> (i & 31u)) & 1u : 0u;
            if (bm | be) nxt[opos++] = h;
            else nxt[zpos++] = h;
        }
        __threadfence_block();
        __syncthreads();  // every stored bit has been read, every natural-order bit is in tm / te
        // (the bits were set by atomics, which work in L2: read them past this CU's L1, where the lines may still sit
        // as the copy loop of an earlier weird line read them)
        if (f & 1u)
            for (uint32_t i = tid; i < S.stride_w; i += 1024u) ym[i] = __hip_atomic_load(&tm[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (f & 2u)
            for (uint32_t i = tid; i < S.stride_w; i += 1024u) ye[i] = __hip_atomic_load(&te[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __threadfence_block();
        __syncthreads();
        uint32_t* t = cur;
        cur = nxt;
        nxt = t;
    }
}

// int32 rows as Accessor::fill_genotype_array writes them (fill_genotype_array_advance,
// accessor_internals_new.hpp:198-384), bug-compatible: the per-haplotype sequence of overwrites
// is replayed exactly (REF / first ALT, extra ALTs incl. the negated-sparse overwrite/restore of
// :243-256 and the haploid "allele 1" of :269, then missing, end-of-vector, phase toggle).
// One workgroup per BCF line, one haplotype per thread per iteration, 4-byte coalesced stores.
struct ComposeArgs {
    const uint32_t* planes;   // [n_bin] decoded planes (sparse: listed positions, raw)
    uint32_t stride_w;
    const uint8_t* kind;      // per binary line
    const uint8_t* side;      // per binary line (nullptr: no side channels)
    const uint32_t* ones;     // per binary line
    DecSide S;
    const uint32_t* bcf_first_bin;
    const uint32_t* bcf_n_allele;
    const uint32_t* line_block;
    const DecBlock* blocks;
    uint32_t N, n_samples;
    int32_t* out;
    uint64_t out_stride;
    uint32_t* line_ngt;       // per BCF line
    uint64_t* allele_counts;  // [n_bcf][max_alleles] or nullptr
    uint32_t max_alleles;
    const uint32_t* out_index;  // nullptr: line li of the launch is output row li; else row out_index[li] (batched queries)
};

__global__ void __launch_bounds__(256) k_compose_gt(ComposeArgs C) {
    const uint32_t li = blockIdx.x;
    const uint32_t start = C.bcf_first_bin[li];
    const uint32_t n_allele = C.bcf_n_allele[li];
    const uint32_t k0 = C.kind[start];
    const uint32_t Nl = (k0 & KIND_HAPLOID) ? C.n_samples : C.N;
    const int32_t DP = (int32_t)C.blocks[C.line_block[start]].default_phasing;
    const uint32_t f = C.side ? C.side[start] : 0u;
    const uint32_t* mp = C.S.miss_planes ? C.S.miss_planes + (size_t)start * C.S.stride_w : nullptr;
    const uint32_t* ep = C.S.eov_planes ? C.S.eov_planes + (size_t)start * C.S.stride_w : nullptr;
    const uint32_t* pp = C.S.phase_planes ? C.S.phase_planes + (size_t)start * C.S.stride_w : nullptr;
    const uint32_t oi = C.out_index ? C.out_index[li] : li;  // where this line's row, value count and allele counts go
    int32_t* orow = C.out + (size_t)oi * C.out_stride;
    // value of haplotype i of this line
    // w0 = word i / 32 of the line's first plane (the callers fetch it ahead of time)
    auto value_of = [&](uint32_t i, uint32_t w0) -> int32_t {
        const int32_t ph = (int32_t)(i & 1u) & DP;
        const uint32_t wi = i >> 5, bi = i & 31u;
        int32_t gt;
        {
            const uint32_t bit = (w0 >> bi) & 1u;
            if (!(k0 & KIND_WAH)) {
                const bool neg = (k0 & KIND_NEGATED) != 0u;
                const int32_t allele = neg ? (bit ? 0 : 1) : (bit ? 1 : 0);
                gt = ((allele + 1) << 1) | ph;
            } else if (k0 & KIND_HAPLOID) {
                gt = ((int32_t)bit + 1) << 1;  // haploids carry no phase bit (:225)
            } else {
                gt = (((int32_t)bit + 1) << 1) | ph;
            }
        }
        for (uint32_t alt = 2; alt < n_allele; ++alt) {
            const uint32_t pos = start + alt - 1u;
            const uint32_t k = C.kind[pos];
            const uint32_t bit = (C.planes[(size_t)pos * C.stride_w + wi] >> bi) & 1u;
            if (!(k & KIND_WAH)) {
                if (k & KIND_NEGATED) {
                    if (((gt >> 1) - 1) == 0) gt = (((int32_t)alt + 1) << 1) | ph;
                    if (bit && ((gt >> 1) - 1) == (int32_t)alt) gt = ((0 + 1) << 1) | ph;
                } else if (bit) {
                    gt = (((int32_t)alt + 1) << 1) | ph;
                }
            } else if (bit) {
                if (k & KIND_HAPLOID)
                    gt = (1 + 1) << 1;  // sic: bcf_gt_unphased(y[i]) with y[i] == 1 (:269)
                else
                    gt = (((int32_t)alt + 1) << 1) | ph;
            }
        }
        if ((f & 1u) && mp && ((mp[wi] >> bi) & 1u)) gt = 0 | ph;                       // bcf_gt_missing | phase
        if ((f & 2u) && ep && ((ep[wi] >> bi) & 1u)) gt = GT_VECTOR_END;
        if ((f & 4u) && pp && ((pp[wi] >> bi) & 1u) && gt != GT_VECTOR_END) gt ^= (int32_t)(i & 1u);
        return gt;
    };
    // grid.y workgroups share one line (few lines of many haplotypes: random access); four values per
    // thread leave as one 16-byte store when the row base allows it
    const uint32_t* p0 = C.planes + (size_t)start * C.stride_w;
    if (((C.out_stride & 3u) | (reinterpret_cast<uintptr_t>(C.out) & 15u)) == 0) {
        // the plane words of a thread's next KU quads are fetched together (unconditionally: clamped index), then the
        // 16-byte stores follow one another; written once and not read back here, so the stores are non-temporal
        constexpr uint32_t KU = 8;
        typedef int32_t gt_i32x4 __attribute__((ext_vector_type(4)));
        const uint32_t Nq = Nl / 4u;
        gt_i32x4* orow4 = reinterpret_cast<gt_i32x4*>(orow);
        const uint32_t qs = blockDim.x * gridDim.y;
        for (uint32_t q0 = blockIdx.y * blockDim.x + threadIdx.x; q0 < Nq; q0 += qs * KU) {
            uint32_t pw[KU];
#pragma unroll
            for (uint32_t k = 0; k < KU; ++k) {
                const uint32_t q = q0 + k * qs;
                pw[k] = p0[(q < Nq ? q : 0u) >> 3];
            }
#pragma unroll
            for (uint32_t k = 0; k < KU; ++k) {
                const uint32_t q = q0 + k * qs;
                if (q < Nq) {
                    const gt_i32x4 v = {value_of(4u * q, pw[k]), value_of(4u * q + 1u, pw[k]), value_of(4u * q + 2u, pw[k]),
                                        value_of(4u * q + 3u, pw[k])};
                    __builtin_nontemporal_store(v, orow4 + q);
                }
            }
        }
        if (blockIdx.y == 0 && threadIdx.x < (Nl & 3u)) {
            const uint32_t i = Nq * 4u + threadIdx.x;
            orow[i] = value_of(i, p0[i >> 5]);
        }
    } else {
        for (uint32_t i = blockIdx.y * blockDim.x + threadIdx.x; i < Nl; i += blockDim.x * gridDim.y) orow[i] = value_of(i, p0[i >> 5]);
    }
    if (threadIdx.x == 0 && blockIdx.y == 0) {
        C.line_ngt[oi] = Nl;
        if (C.allele_counts) {
            uint64_t total = 0;
            for (uint32_t alt = 1; alt < n_allele && alt < C.max_alleles; ++alt) {
                const uint64_t o = C.ones[start + alt - 1u];
                C.allele_counts[(size_t)oi * C.max_alleles + alt] = o;
                total += o;
            }
            const uint64_t nm = (f & 1u) ? C.S.n_miss[start] : 0u, ne = (f & 2u) ? C.S.n_eov[start] : 0u;
            C.allele_counts[(size_t)oi * C.max_alleles] = (uint64_t)Nl - (total + nm + ne);
        }
    }
}

}  // namespace xsi

namespace xsi {

// ------------------------------------------------------------------------------------------
// Phenotype dot products (dot_prod/dot_prod.hpp:122-245): Sxy[line][k] = sum over the haplotypes
// that carry the line's ALT allele of y[sample of the haplotype][k].  The reference walks the WAH
// words / sparse list of the line through the current `a`; here the decoded bit planes are in HBM
// already, so this is a bit-matrix x dense-matrix product streamed at one bit per cell.  One wave
// per binary line, float64 accumulation, fixed summation order (lane partial sums, then a wave
// tree), so results are reproducible run to run; they differ from the reference's `a`-order sum
// only in rounding.
// ------------------------------------------------------------------------------------------
template <int KB>
__global__ void __launch_bounds__(256) k_dot_planes(const uint32_t* __restrict__ planes, uint32_t stride_w,
                                                    const uint8_t* __restrict__ kind, uint32_t n_lines, uint32_t N,
                                                    uint32_t n_samples, const double* __restrict__ y, uint32_t n_pheno,
                                                    uint32_t k0, double* __restrict__ out) {
    // phenotypes k0 .. k0+KB-1 of every line; a lane takes one SAMPLE per step: its dosage (0, 1, 2 ALT
    // copies; 0 / 1 on fully haploid lines) is formed once and reused for the KB phenotypes
    const uint32_t l = blockIdx.x * 4u + (threadIdx.x >> 6);
    if (l >= n_lines) return;
    const uint32_t lane = lane_id();
    const bool haploid = (kind[l] & KIND_HAPLOID) != 0u;
    const uint32_t* row = planes + (size_t)l * stride_w;
    const uint32_t nw = stride_w;
    double acc[KB];
#pragma unroll
    for (int k = 0; k < KB; ++k) acc[k] = 0.0;
    for (uint32_t s0 = 0; s0 < n_samples; s0 += 64u) {
        const uint32_t smp = s0 + lane;
        uint32_t dosage = 0;
        if (smp < n_samples) {
            if (haploid) {
                dosage = (row[smp >> 5] >> (smp & 31u)) & 1u;
            } else {
                const uint32_t h = 2u * smp;  // even: both haplotypes of the sample sit in one word
                const uint32_t two = (h >> 5) < nw ? (row[h >> 5] >> (h & 31u)) & 3u : 0u;
                dosage = (two & 1u) + (two >> 1);
            }
        }
        if (dosage) {
            const double d = (double)dosage;
            const double* yp = y + (size_t)smp * n_pheno + k0;
#pragma unroll
            for (int k = 0; k < KB; ++k)
                if (k0 + (uint32_t)k < n_pheno) acc[k] += d * yp[k];
        }
    }
    (void)N;
#pragma unroll
    for (int k = 0; k < KB; ++k) {
        double a = acc[k];
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) a += __shfl_xor(a, d, 64);
        if (lane == 0 && k0 + (uint32_t)k < n_pheno) out[(size_t)l * n_pheno + k0 + (uint32_t)k] = a;
    }
}

// Phenotype dot products on COMPOSED genotypes: what the bit planes alone cannot say (which ALT allele a
// haplotype of a multi-allelic line carries once a negated sparse line, missing or end-of-vector entries are
// involved) the int32 rows can.  One wave per binary line (parent BCF row, ALT index); a lane takes one
// sample per step and counts its copies of that allele (missing and end-of-vector values never match).
// float64, same fixed summation order as k_dot_planes, so both paths give identical sums on lines both cover.
template <int KB>
__global__ void __launch_bounds__(256) k_dot_gt(const int32_t* __restrict__ rows, uint64_t stride,
                                                const uint32_t* __restrict__ line_ngt,
                                                const uint32_t* __restrict__ bin_parent,
                                                const uint32_t* __restrict__ bin_alt, uint32_t n_bin, uint32_t n_samples,
                                                const double* __restrict__ y, uint32_t n_pheno, uint32_t k0,
                                                double* __restrict__ out) {
    const uint32_t l = blockIdx.x * 4u + (threadIdx.x >> 6);
    if (l >= n_bin) return;
    const uint32_t lane = lane_id();
    const uint32_t parent = bin_parent[l];
    const int32_t alt = (int32_t)bin_alt[l];
    const bool haploid = line_ngt[parent] == n_samples;
    const int32_t* row = rows + (size_t)parent * stride;
    double acc[KB];
#pragma unroll
    for (int k = 0; k < KB; ++k) acc[k] = 0.0;
    for (uint32_t s0 = 0; s0 < n_samples; s0 += 64u) {
        const uint32_t smp = s0 + lane;
        uint32_t dosage = 0;
        if (smp < n_samples) {
            if (haploid) {
                dosage = ((row[smp] >> 1) - 1) == alt;
            } else {
                const int32_t a0 = row[2u * smp], a1 = row[2u * smp + 1u];
                dosage = (uint32_t)(((a0 >> 1) - 1) == alt) + (uint32_t)(((a1 >> 1) - 1) == alt);
            }
        }
        if (dosage) {
            const double d = (double)dosage;
            const double* yp = y + (size_t)smp * n_pheno + k0;
#pragma unroll
            for (int k = 0; k < KB; ++k)
                if (k0 + (uint32_t)k < n_pheno) acc[k] += d * yp[k];
        }
    }
#pragma unroll
    for (int k = 0; k < KB; ++k) {
        double a = acc[k];
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) a += __shfl_xor(a, d, 64);
        if (lane == 0 && k0 + (uint32_t)k < n_pheno) out[(size_t)l * n_pheno + k0 + (uint32_t)k] = a;
    }
}

// Many phenotypes: the product is a real dense contraction, [lines x N] 0/1 times [N x K] float64, and
// goes to the float64 matrix cores.  v_mfma_f64_16x16x4_f64: lane l holds A[row l&15][k l>>4] and
// B[k l>>4][col l&15]; result register i of lane l is C[row (l>>4) + 4i][col l&15].  Rows = lines,
// k = 4 consecutive haplotypes, columns = 16 phenotypes.  A wave owns 64 lines (4 tiles) so that one
// B fragment (read from L2) feeds four MFMAs; products are exact (0/1 times y), accumulation is
// float64.  Diploid lines only (haplotype h -> sample h/2); blocks with fully haploid lines take
// the scalar kernel.
typedef double xsi_d4 __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(256) k_dot_mfma(const uint32_t* __restrict__ planes, uint32_t stride_w, uint32_t n_lines,
                                                  uint32_t N, const double* __restrict__ y, uint32_t n_pheno, uint32_t k0,
                                                  double* __restrict__ out) {
    const uint32_t lane = lane_id();
    const uint32_t w = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t line0 = (blockIdx.x * 4u + w) * 64u;
    if (line0 >= n_lines) return;
    const uint32_t r = lane & 15u, kk = lane >> 4, c = lane & 15u;
    const bool cvalid = k0 + c < n_pheno;
    const uint32_t* rowp[4];
    bool lvalid[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const uint32_t line = line0 + 16u * (uint32_t)t + r;
        lvalid[t] = line < n_lines;
        rowp[t] = planes + (size_t)(lvalid[t] ? line : 0u) * stride_w;
    }
    xsi_d4 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = (xsi_d4){0.0, 0.0, 0.0, 0.0};
    const double* ycol = y + k0 + c;
    for (uint32_t h32 = 0; h32 < N; h32 += 32u) {
        // one 32-haplotype word per line and tile, eight B values per lane, then 8 x 4 MFMAs
        uint32_t word[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) word[t] = lvalid[t] ? rowp[t][h32 >> 5] : 0u;
        double b[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const uint32_t hap = h32 + 4u * (uint32_t)q + kk;
            b[q] = (cvalid && hap < N) ? ycol[(size_t)(hap >> 1) * n_pheno] : 0.0;
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const uint32_t sh = 4u * (uint32_t)q + kk;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const double a = ((word[t] >> sh) & 1u) ? 1.0 : 0.0;
                acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b[q], acc[t], 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint32_t line = line0 + 16u * (uint32_t)t + kk + 4u * (uint32_t)i;
            if (line < n_lines && cvalid) out[(size_t)line * n_pheno + k0 + c] = acc[t][i];
        }
}

int dot_planes(xsi_hip_ctx* ctx, const DecodePlan& P, const uint32_t* planes, uint32_t stride_w, const double* d_y,
               uint32_t n_pheno, double* d_out) {
    if (!P.n_bin) return XSI_OK;
    bool any_haploid = false;
    for (auto& b : P.blocks_h)
        if (b.off_line_haploid != VAL_UNDEFINED) any_haploid = true;
    if (n_pheno >= 8u && !any_haploid && !tuning_env("XSI_DOT_SCALAR")) {
        for (uint32_t k0 = 0; k0 < n_pheno; k0 += 16u)
            k_dot_mfma<<<dim3((P.n_bin + 255u) / 256u), dim3(256), 0, ctx->stream>>>(planes, stride_w, P.n_bin, P.L.N, d_y,
                                                                                      n_pheno, k0, d_out);
        HIP_TRY(hipGetLastError());
        return XSI_OK;
    }
    const dim3 grid((P.n_bin + 3u) / 4u), block(256);
    uint32_t k0 = 0;
    for (; k0 + 4u <= n_pheno; k0 += 4u)
        k_dot_planes<4><<<grid, block, 0, ctx->stream>>>(planes, stride_w, P.L.kind, P.n_bin, P.L.N, P.L.n_samples, d_y,
                                                         n_pheno, k0, d_out);
    for (; k0 < n_pheno; ++k0)
        k_dot_planes<1><<<grid, block, 0, ctx->stream>>>(planes, stride_w, P.L.kind, P.n_bin, P.L.N, P.L.n_samples, d_y,
                                                         n_pheno, k0, d_out);
    HIP_TRY(hipGetLastError());
    return XSI_OK;
}

// Planes of every binary line of the parsed blocks, plus the side-channel planes when present.
int decode_all_planes(xsi_hip_ctx* ctx, const void* d_file, DecodePlan& P, DecodedPlanes* out, const PartialDecode* part) {
    hipStream_t s = ctx->stream;
    const uint32_t n_bin = P.n_bin;
    const uint32_t stride_w = P.L.y_stride64 * 2u;
    const bool cont = part && !part->first;  // continuation of a prefix decode: more lines into what the caller keeps
    if (cont) {
        if (out->stride_w != stride_w || !out->planes) return set_error(XSI_ERR_ARG, "decode_all_planes: continuation without planes");
    } else {
        out->stride_w = stride_w;
        WS(out->planes, "gt.planes", 4ull * stride_w * (size_t)(n_bin ? n_bin : 1));
    }
    // binary lines [bin_lo, bin_hi) are WAH lines [wah_lo, wah_hi) and sparse lines [bin_lo - wah_lo, bin_hi - wah_hi)
    int rc = part ? decode_planes_partial(ctx, d_file, P, out->planes, stride_w, part->wah_lo, part->wah_hi, part->d_state,
                                          part->bin_lo - part->wah_lo, part->bin_hi - part->wah_hi, part->d_walk + 3,
                                          part->skip_boundaries)
                  : decode_planes(ctx, d_file, P, out->planes, stride_w, /*apply_negation=*/0);
    P.L.sp_state = nullptr;  // the plan outlives this call (cache entries keep a copy): no dangling range in it
    P.L.sp_lo = P.L.sp_hi = 0u;
    if (rc) return rc;
    bool side = false, pbwt_weird = false;
    for (auto& b : P.blocks_h) {
        if (b.off_line_missing != VAL_UNDEFINED || b.off_line_eov != VAL_UNDEFINED || b.off_line_phase != VAL_UNDEFINED) side = true;
        if ((b.off_line_missing != VAL_UNDEFINED || b.off_line_eov != VAL_UNDEFINED) && b.strategy == WS_PBWT_WAH) {
            if (b.off_line_haploid != VAL_UNDEFINED)
                return set_error(XSI_ERR_UNSUPPORTED, "weirdness strategy WS_PBWT_WAH in a block with fully haploid lines (the reference "
                                 "writes those lines through a1 and reads them through a_weird: not decodable as written)");
            pbwt_weird = true;
        }
    }
    if (part && pbwt_weird) return set_error(XSI_ERR_UNSUPPORTED, "decode_all_planes: ranged decode of a WS_PBWT_WAH block");
    if (!cont) {
        out->has_side = side;
        WS(out->n_miss, "gt.n_miss", 4ull * n_bin + 64);
        WS(out->n_eov, "gt.n_eov", 4ull * n_bin + 64);
    }
    if (side) {
        DecSide S{};
        S.stride_w = stride_w;
        WS(S.miss_start, "gt.miss_start", 4ull * n_bin + 64);
        WS(S.eov_start, "gt.eov_start", 4ull * n_bin + 64);
        WS(S.phase_start, "gt.phase_start", 4ull * n_bin + 64);
        if (cont) {
            S.side = out->side;
            S.miss_planes = out->miss_planes;
            S.eov_planes = out->eov_planes;
            S.phase_planes = out->phase_planes;
        } else {
            WS(S.side, "gt.side", (size_t)n_bin + 64);
            WS(S.miss_planes, "gt.dmiss_planes", 4ull * stride_w * (size_t)n_bin);
            WS(S.eov_planes, "gt.deov_planes", 4ull * stride_w * (size_t)n_bin);
            WS(S.phase_planes, "gt.dphase_planes", 4ull * stride_w * (size_t)n_bin);
        }
        S.n_miss = out->n_miss;
        S.n_eov = out->n_eov;
        uint32_t lines = n_bin;
        if (part) {  // the side matrices of the lines this range makes valid, cursors carried in the state
            S.bin_lo = part->bin_lo;
            S.bin_hi = part->bin_hi < n_bin ? part->bin_hi : n_bin;
            S.walk_state = part->d_walk;
            lines = S.bin_hi > S.bin_lo ? S.bin_hi - S.bin_lo : 0u;
        }
        if (!cont) {
            k_dec_side_flags<<<dim3(P.n_blocks), dim3(64), 0, s>>>((const uint8_t*)d_file, P.d_blocks, P.L, S);
            HIP_TRY(hipGetLastError());
        }
        if (lines) {
            k_dec_side_walk<<<dim3(P.n_blocks), dim3(64), 0, s>>>((const uint8_t*)d_file, P.d_blocks, P.L, S);
            HIP_TRY(hipGetLastError());
            const uint32_t lds = stride_w * 4u;
            HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_dec_side_planes),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            k_dec_side_planes<<<dim3(lines), dim3(64), lds, s>>>((const uint8_t*)d_file, P.d_blocks, P.L, S, n_bin);
            HIP_TRY(hipGetLastError());
        }
        if (pbwt_weird) {  // version-4 files: the missing / end-of-vector lines back into natural order
            uint32_t *d_aw, *d_tmp;
            WS(d_aw, "gt.weird_a", 8ull * P.L.N * (size_t)P.n_blocks);
            WS(d_tmp, "gt.weird_tmp", 8ull * stride_w * (size_t)P.n_blocks);
            k_dec_side_unpermute<<<dim3(P.n_blocks), dim3(1024), 0, s>>>(P.d_blocks, P.L, S, d_aw, d_tmp);
            HIP_TRY(hipGetLastError());
        }
        if (!cont) {
            out->side = S.side;
            out->miss_planes = S.miss_planes;
            out->eov_planes = S.eov_planes;
            out->phase_planes = S.phase_planes;
        }
    } else if (!cont) {
        HIP_TRY(hipMemsetAsync(out->n_miss, 0, 4ull * n_bin + 64, s));
        HIP_TRY(hipMemsetAsync(out->n_eov, 0, 4ull * n_bin + 64, s));
        out->side = nullptr;
        out->miss_planes = out->eov_planes = out->phase_planes = nullptr;
    }
    return XSI_OK;
}

// int32 rows for n_out BCF lines given (first binary line, n_allele) per line (device arrays).
int compose_lines(xsi_hip_ctx* ctx, const DecodePlan& P, const DecodedPlanes& D, const uint32_t* d_first_bin,
                  const uint32_t* d_n_allele, uint32_t n_out, int32_t* d_gt_out, uint64_t gt_stride,
                  uint32_t* d_line_ngt, uint64_t* d_allele_counts, uint32_t max_alleles, const uint32_t* d_out_index) {
    if (!n_out) return XSI_OK;
    ComposeArgs C{};
    C.planes = D.planes;
    C.stride_w = D.stride_w;
    C.kind = P.L.kind;
    C.side = D.side;
    C.ones = P.L.ones;
    C.S.stride_w = D.stride_w;
    C.S.miss_planes = D.miss_planes;
    C.S.eov_planes = D.eov_planes;
    C.S.phase_planes = D.phase_planes;
    C.S.n_miss = D.n_miss;
    C.S.n_eov = D.n_eov;
    C.bcf_first_bin = d_first_bin;
    C.bcf_n_allele = d_n_allele;
    C.line_block = P.L.line_block;
    C.blocks = P.d_blocks;
    C.N = P.L.N;
    C.n_samples = P.L.n_samples;
    C.out = d_gt_out;
    C.out_stride = gt_stride;
    C.line_ngt = d_line_ngt;
    C.allele_counts = d_allele_counts;
    C.max_alleles = max_alleles;
    C.out_index = d_out_index;
    stage_mark(ctx, XSI_ST_GT_COMPOSE);
    uint32_t splits = (P.L.N + 2047u) / 2048u;  // >= 8 values per thread
    if (splits > 2048u / n_out) splits = 2048u / n_out;
    if (splits < 1u) splits = 1u;
    k_compose_gt<<<dim3(n_out, splits), dim3(256), 0, ctx->stream>>>(C);
    HIP_TRY(hipGetLastError());
    return XSI_OK;
}

// Sample subsetting of composed lines (NewDecompressor::fill_selected_genotypes,
// gt_decompressor_new.hpp:209-238): out row r = the listed samples' values of composed row r, in list
// order, 1 or 2 values per sample by the line's ploidy (line_ngt / n_samples); ac[r][k-1] = selected values
// whose allele is k (the AC the reference recomputes for bcftools-style "-s").  One workgroup per line.
__global__ void __launch_bounds__(256) k_select_samples(const int32_t* __restrict__ rows, uint64_t row_stride,
                                                        const uint32_t* __restrict__ line_ngt, uint32_t n_samples,
                                                        const uint32_t* __restrict__ sel, uint32_t n_sel,
                                                        int32_t* __restrict__ out, uint64_t out_stride,
                                                        uint32_t* __restrict__ ac, uint32_t n_alt) {
    const uint32_t r = blockIdx.x;
    const uint32_t ploidy = line_ngt[r] / n_samples;  // 1 or 2
    const int32_t* in = rows + (size_t)r * row_stride;
    int32_t* o = out + (size_t)r * out_stride;
    __shared__ uint32_t s_ac[32];
    if (threadIdx.x < 32u) s_ac[threadIdx.x] = 0;
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < n_sel * ploidy; i += blockDim.x) {
        const uint32_t smp = sel[i / ploidy], k = i - (i / ploidy) * ploidy;
        const int32_t v = in[(size_t)smp * ploidy + k];
        o[i] = v;
        const int32_t allele = (v >> 1) - 1;  // bcf_gt_allele
        if (allele >= 1 && (uint32_t)allele <= n_alt && n_alt <= 32u) atomicAdd(&s_ac[allele - 1], 1u);
    }
    __syncthreads();
    if (ac && threadIdx.x < n_alt && n_alt <= 32u) ac[(size_t)r * n_alt + threadIdx.x] = s_ac[threadIdx.x];
}

int select_samples(xsi_hip_ctx* ctx, const int32_t* d_rows, uint64_t row_stride, const uint32_t* d_line_ngt,
                   uint32_t n_lines, uint32_t n_samples, const uint32_t* d_sel, uint32_t n_sel, int32_t* d_out,
                   uint64_t out_stride, uint32_t* d_ac, uint32_t n_alt) {
    if (!n_lines) return XSI_OK;
    k_select_samples<<<dim3(n_lines), dim3(256), 0, ctx->stream>>>(d_rows, row_stride, d_line_ngt, n_samples, d_sel, n_sel,
                                                                  d_out, out_stride, d_ac, n_alt);
    HIP_TRY(hipGetLastError());
    return XSI_OK;
}

}  // namespace xsi

namespace xsi {
// bit row -> (allele + 1) << 1 | phase, four values per thread and step (one 16-byte store)
__global__ void __launch_bounds__(256) k_expand_bit_rows(const uint8_t* __restrict__ bits, uint32_t bit_stride, const uint8_t* __restrict__ fast,
                                                         int32_t* __restrict__ rows, uint64_t N, int32_t dp) {
    const uint64_t l = blockIdx.x;
    if (!fast[l]) return;
    const uint8_t* b = bits + l * bit_stride;
    int32_t* r = rows + l * N;
    for (uint64_t h = (uint64_t)threadIdx.x * 4u; h < N; h += 1024u) {
        const uint32_t nib = (uint32_t)(b[h >> 3] >> (h & 7u)) & 15u;
        int32_t v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = (int32_t)((((nib >> k) & 1u) + 1u) << 1) | ((k & 1) ? dp : 0);
        if (h + 4u <= N && ((reinterpret_cast<uintptr_t>(r + h) & 15u) == 0u)) {
            *reinterpret_cast<int4*>(r + h) = make_int4(v[0], v[1], v[2], v[3]);
        } else {
            for (int k = 0; k < 4 && h + (uint64_t)k < N; ++k) r[h + k] = v[k];
        }
    }
}

int expand_bit_rows(xsi_hip_ctx* ctx, const uint8_t* d_bits, uint32_t bit_stride, const uint8_t* d_fast, int32_t* d_rows,
                    uint64_t N, uint64_t n_lines, int32_t default_phased) {
    if (!n_lines) return XSI_OK;
    k_expand_bit_rows<<<dim3((unsigned)n_lines), dim3(256), 0, ctx->stream>>>(d_bits, bit_stride, d_fast, d_rows, N, default_phased ? 1 : 0);
    HIP_TRY(hipGetLastError());
    return XSI_OK;
}
}  // namespace xsi

extern "C" {

// region_offset: bytes of the blocks region that earlier ranges of the same job already wrote in front of d_out (block
// offsets continue from there; xsi_hip_reencode walks a file in ranges of whole blocks)
static int encode_gt_impl(xsi_hip_ctx* ctx, const xsi_encode_params* p, const int32_t* d_gt, uint64_t gt_stride,
                          uint64_t n_lines, const uint32_t* h_ngt, const uint32_t* h_n_allele, void* d_out,
                          uint64_t out_capacity, uint64_t* d_block_offsets, xsi_encode_result* h_result, uint64_t region_offset);

int xsi_hip_encode_gt(xsi_hip_ctx* ctx, const xsi_encode_params* p, const int32_t* d_gt, uint64_t gt_stride,
                      uint64_t n_lines, const uint32_t* h_ngt, const uint32_t* h_n_allele, void* d_out,
                      uint64_t out_capacity, uint64_t* d_block_offsets, xsi_encode_result* h_result) {
    return encode_gt_impl(ctx, p, d_gt, gt_stride, n_lines, h_ngt, h_n_allele, d_out, out_capacity, d_block_offsets, h_result, 0);
}

static int encode_gt_impl(xsi_hip_ctx* ctx, const xsi_encode_params* p, const int32_t* d_gt, uint64_t gt_stride,
                          uint64_t n_lines, const uint32_t* h_ngt, const uint32_t* h_n_allele, void* d_out,
                          uint64_t out_capacity, uint64_t* d_block_offsets, xsi_encode_result* h_result, uint64_t region_offset) {
    if (!ctx || !p || !d_gt || !h_ngt || !h_n_allele || !d_out) return set_error(XSI_ERR_ARG, "encode_gt: null argument");
    if (!region_offset) ctx->block_sizes_pos = 0;  // (xsi_hip_reencode's later ranges continue the same call)
    if (!p->n_samples || !p->block_len) return set_error(XSI_ERR_ARG, "encode_gt: n_samples and block_len must be > 0");
    if (at_mismatch_window(p->n_samples))
        return set_error(XSI_ERR_UNSUPPORTED, "%s: %u samples fall in the reference's A_T mismatch window (32768..65535: 16-bit "
                         "block data under a 32-bit header, prefix array wraps modulo 65536); it cannot be encoded decodably", "encode_gt",
                         p->n_samples);
    if (n_lines == 0 || n_lines > 0x7FFFFFFFull) return set_error(XSI_ERR_ARG, "n_lines out of range");
    const uint32_t N = 2u * p->n_samples;
    if (gt_stride < N) return set_error(XSI_ERR_ARG, "gt_stride %llu < 2*n_samples", (unsigned long long)gt_stride);
    HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    const uint32_t n_bcf = (uint32_t)n_lines;
    const uint32_t n_blocks = (uint32_t)((n_lines + p->block_len - 1) / p->block_len);
    // host-side line bookkeeping
    // (two plain passes over the caller's arrays: this loop runs in front of every launch of the call, and at a
    //  million lines a division and three push_backs per line cost as much as the kernels behind it)
    std::vector<uint32_t> first_bin(n_bcf);
    std::vector<EncBlock> blocks(n_blocks);
    uint64_t n_bin64 = 0;
    uint32_t max_ploidy = 0;
    for (uint32_t l = 0; l < n_bcf; ++l) {
        if (h_n_allele[l] < 2)
            return set_error(XSI_ERR_UNSUPPORTED, "line %u has %u alleles: lines without an ALT allele corrupt the reference's "
                             "flag reindexing (gt_block.hpp:650-666) and are rejected", l, h_n_allele[l]);
        if (h_ngt[l] != p->n_samples && h_ngt[l] != N) return set_error(XSI_ERR_ARG, "PLOIDY ERROR: line %u has %u values", l, h_ngt[l]);
        const uint32_t pl = h_ngt[l] == N ? 2u : 1u;
        if (pl > max_ploidy) max_ploidy = pl;
        first_bin[l] = (uint32_t)n_bin64;
        n_bin64 += h_n_allele[l] - 1;
        if (n_bin64 > 0x7FFFFFFFull) return set_error(XSI_ERR_ARG, "too many binary lines in one call");
    }
    const uint32_t n_bin = (uint32_t)n_bin64;
    for (uint32_t b = 0; b < n_blocks; ++b) {
        memset(&blocks[b], 0, sizeof(EncBlock));
        const uint32_t f = b * p->block_len;
        const uint32_t e = (uint32_t)(((uint64_t)f + p->block_len < n_lines) ? f + p->block_len : n_lines);
        blocks[b].first_bcf = f;
        blocks[b].n_bcf = e - f;
        blocks[b].first_bin = first_bin[f];
        blocks[b].n_bin = (e < n_bcf ? first_bin[e] : n_bin) - first_bin[f];
        if (blocks[b].n_bin > MAX_BIN_PER_BLOCK)
            return set_error(XSI_ERR_UNSUPPORTED, "Variant BCF generation error, BM bits: block %u has %u binary lines", b,
                             blocks[b].n_bin);
    }
    const uint32_t stride_w = ((N + 63u) / 64u) * 2u;
    // device metadata
    uint32_t *d_nbits, *d_nallele, *d_first_bin, *d_parent, *d_bin_nbits;
    WS(d_nbits, "gt.bcf_nbits", 4ull * n_bcf);
    WS(d_nallele, "gt.bcf_nallele", 4ull * n_bcf);
    WS(d_first_bin, "gt.bcf_first_bin", 4ull * n_bcf);
    WS(d_parent, "gt.bin_parent", 4ull * n_bin);
    WS(d_bin_nbits, "gt.bin_nbits", 4ull * n_bin);
    // what the unpack kernel reads goes first; the per-binary-line arrays (read by the classification behind it) are
    // formed and sent while it runs; one synchronisation behind them.
    // From here to the synchronisation behind the second pair of copies, asynchronous copies read the caller's arrays and
    // this frame's vectors: every exit in between (a workspace allocation that fails, a launch error) waits for the stream
    // first (the guard's destructor), or the DMA would read memory that has been handed back.
    std::vector<uint32_t> parent, nbits_bin;  // (declared in front of the guard: destroyed behind its wait)
    StreamSyncGuard copies_in_flight(s);
    HIP_TRY(hipMemcpyAsync(d_nbits, h_ngt, 4ull * n_bcf, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(d_nallele, h_n_allele, 4ull * n_bcf, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(d_first_bin, first_bin.data(), 4ull * n_bcf, hipMemcpyHostToDevice, s));

    UnpackArgs U{};
    U.gt = d_gt;
    U.gt_stride = gt_stride;
    U.bcf_nbits = d_nbits;
    U.bcf_n_allele = d_nallele;
    U.bcf_first_bin = d_first_bin;
    U.n_samples = p->n_samples;
    U.default_phased = p->default_phased;
    U.stride_w = stride_w;
    WS(U.planes, "gt.planes", 4ull * stride_w * (size_t)n_bin);
    WS(U.ref_planes, "gt.ref_planes", 4ull * stride_w * (size_t)n_bcf);
    WS(U.miss_planes, "gt.miss_planes", 4ull * stride_w * (size_t)n_bcf);
    WS(U.eov_planes, "gt.eov_planes", 4ull * stride_w * (size_t)n_bcf);
    WS(U.phase_planes, "gt.phase_planes", 4ull * stride_w * (size_t)n_bcf);
    WS(U.cnt, "enc.cnt", 4ull * n_bin);
    WS(U.ref_cnt, "gt.ref_cnt", 4ull * n_bcf);
    WS(U.miss_cnt, "gt.miss_cnt", 4ull * n_bcf);
    WS(U.eov_cnt, "gt.eov_cnt", 4ull * n_bcf);
    WS(U.bcf_flags, "gt.bcf_flags", 4ull * n_bcf);
    WS(U.kind, "enc.kind", (size_t)n_bin);
    WS(U.d_error, "gt.error", 64);
    HIP_TRY(hipMemsetAsync(U.d_error, 0, 4, s));
    U.quad_ok = ((reinterpret_cast<uintptr_t>(d_gt) & 15u) == 0u && (gt_stride & 3u) == 0u && !tuning_env("XSI_GT_NO_QUADS")) ? 1u : 0u;
    stage_mark(ctx, XSI_ST_GT_UNPACK);
    k_unpack_gt<<<dim3(n_bcf), dim3(256), 0, s>>>(U);
    HIP_TRY(hipGetLastError());
    parent.resize(n_bin);
    nbits_bin.resize(n_bin);
    for (uint32_t l = 0; l < n_bcf; ++l) {
        const uint32_t b = first_bin[l], na = h_n_allele[l] - 1u, ng = h_ngt[l];
        for (uint32_t k = 0; k < na; ++k) {
            parent[b + k] = l;
            nbits_bin[b + k] = ng;
        }
    }
    HIP_TRY(hipMemcpyAsync(d_parent, parent.data(), 4ull * n_bin, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(d_bin_nbits, nbits_bin.data(), 4ull * n_bin, hipMemcpyHostToDevice, s));
    HIP_TRY(hipStreamSynchronize(s));  // every copy has left the host vectors (an error return below may destroy them early)
    copies_in_flight.release();

    EncLines L{};
    L.planes = U.planes;
    L.plane_stride_w = stride_w;
    L.n_bin = n_bin;
    L.N = N;
    L.aet = p->n_samples <= 65535u ? 2u : 4u;
    L.thr = p->mac_threshold;
    L.bin_nbits = d_bin_nbits;
    L.bin_parent = d_parent;
    L.ref_planes = U.ref_planes;
    L.ref_cnt = U.ref_cnt;
    L.cnt = U.cnt;
    L.kind = U.kind;

    EncSide S{};
    S.miss_planes = U.miss_planes;
    S.eov_planes = U.eov_planes;
    S.phase_planes = U.phase_planes;
    S.bcf_nbits = d_nbits;
    S.bcf_flags = U.bcf_flags;
    S.bcf_first_bin = d_first_bin;
    S.miss_cnt = U.miss_cnt;
    S.eov_cnt = U.eov_cnt;
    S.n_bcf = n_bcf;
    S.plane_stride_w = stride_w;
    S.aet = L.aet;
    S.strategy = p->wah_encode_missing ? WS_WAH : WS_SPARSE;
    WS(S.miss_size, "gt.miss_size", 4ull * n_bcf);
    WS(S.eov_size, "gt.eov_size", 4ull * n_bcf);
    WS(S.phase_len, "gt.phase_len", 4ull * n_bcf);
    WS(S.miss_off, "gt.miss_off", 4ull * n_bcf);
    WS(S.eov_off, "gt.eov_off", 4ull * n_bcf);
    WS(S.phase_off, "gt.phase_off", 4ull * n_bcf);
    k_side_sizes<<<dim3((n_bcf + 3u) / 4u), dim3(256), 0, s>>>(S);
    HIP_TRY(hipGetLastError());
    // flag vectors need the block table on the device: encode_run uploads it, so stage a copy here
    EncBlock* d_blocks;
    WS(d_blocks, "enc.blocks", sizeof(EncBlock) * (size_t)n_blocks);
    uint32_t* flagbits;
    WS(flagbits, "enc.flagbits", 4ull * (MAX_BIN_PER_BLOCK / 32) * FV_COUNT * (size_t)n_blocks);
    HIP_TRY(hipMemcpyAsync(d_blocks, blocks.data(), sizeof(EncBlock) * (size_t)n_blocks, hipMemcpyHostToDevice, s));
    k_side_flagbits<<<dim3(n_blocks), dim3(256), 0, s>>>(d_blocks, S, flagbits);
    HIP_TRY(hipGetLastError());

    int rc = encode_run(ctx, p, L, S, blocks, d_out, out_capacity, d_block_offsets, h_result, region_offset);
    if (rc) return rc;
    uint32_t err = 0;
    HIP_TRY(hipMemcpyAsync(&err, U.d_error, 4, hipMemcpyDeviceToHost, s));  // on the context's stream: a copy on the
    HIP_TRY(hipStreamSynchronize(s));                                       // null stream drags every other stream in
    if (err) return set_error(XSI_ERR_ARG, "Unknown allele error !");
    if (h_result) h_result->max_ploidy = max_ploidy;
    return XSI_OK;
}

int xsi_hip_decode_gt(xsi_hip_ctx* ctx, const void* d_file, uint64_t file_len, uint64_t first_block,
                      uint64_t n_blocks64, const uint32_t* h_n_allele, uint64_t n_lines, int32_t* d_gt_out,
                      uint64_t gt_stride, uint32_t* h_line_ngt, uint64_t* d_allele_counts, uint32_t max_alleles) {
    if (!ctx || !d_file || !h_n_allele || !d_gt_out) return set_error(XSI_ERR_ARG, "decode_gt: null argument");
    if (file_len < 256) return set_error(XSI_ERR_FORMAT, "file image shorter than the 256-byte header");
    HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    DecodePlan P;
    int rc = decode_prepare(ctx, d_file, file_len, first_block, n_blocks64, &P);
    if (rc) return rc;
    if (P.n_bcf != n_lines)
        return set_error(XSI_ERR_ARG, "decode_gt: blocks hold %u BCF lines, caller passed %llu", P.n_bcf, (unsigned long long)n_lines);
    if (gt_stride < P.L.N) return set_error(XSI_ERR_ARG, "gt_stride %llu < %u haplotypes", (unsigned long long)gt_stride, P.L.N);
    const uint32_t n_bcf = P.n_bcf;
    // BCF line -> first binary line; must agree with every block's dictionary
    std::vector<uint32_t> first_bin(n_bcf);
    {
        uint32_t l = 0;
        for (uint32_t b = 0; b < P.n_blocks; ++b) {
            uint32_t acc = P.blocks_h[b].first_bin;
            for (uint32_t i = 0; i < P.blocks_h[b].n_bcf; ++i, ++l) {
                if (h_n_allele[l] < 2) return set_error(XSI_ERR_ARG, "decode_gt: line %u has fewer than 2 alleles", l);
                first_bin[l] = acc;
                acc += h_n_allele[l] - 1;
            }
            if (acc != P.blocks_h[b].first_bin + P.blocks_h[b].n_bin)
                return set_error(XSI_ERR_ARG, "decode_gt: allele numbers of block %u add up to %u binary lines, the block has %u", b,
                                 acc - P.blocks_h[b].first_bin, P.blocks_h[b].n_bin);
        }
    }
    uint32_t *d_first_bin, *d_nallele, *d_line_ngt;
    WS(d_first_bin, "gt.bcf_first_bin", 4ull * n_bcf);
    WS(d_nallele, "gt.bcf_nallele", 4ull * n_bcf);
    WS(d_line_ngt, "gt.line_ngt", 4ull * n_bcf);
    HIP_TRY(hipMemcpyAsync(d_first_bin, first_bin.data(), 4ull * n_bcf, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(d_nallele, h_n_allele, 4ull * n_bcf, hipMemcpyHostToDevice, s));
    // first_bin is a local vector and h_n_allele the caller's: the copies must have left host memory
    // before any error return below can unwind this frame
    HIP_TRY(hipStreamSynchronize(s));
    DecodedPlanes DPn;
    rc = decode_all_planes(ctx, d_file, P, &DPn);
    if (rc) return rc;
    rc = compose_lines(ctx, P, DPn, d_first_bin, d_nallele, n_bcf, d_gt_out, gt_stride, d_line_ngt, d_allele_counts,
                       max_alleles);
    if (rc) return rc;
    stage_mark(ctx, -1);
    if (h_line_ngt) HIP_TRY(hipMemcpyAsync(h_line_ngt, d_line_ngt, 4ull * n_bcf, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    stage_collect(ctx);
    return XSI_OK;
}


// One range of blocks through the composed rows (see k_dot_gt).  h_n_allele covers exactly these blocks.
static int dot_gt_range(xsi_hip_ctx* ctx, const void* d_file, uint64_t file_len, uint64_t first_block, uint64_t n_blocks,
                        const uint32_t* h_n_allele, uint64_t n_lines, const double* d_pheno, uint32_t n_pheno, double* d_out,
                        uint64_t* n_bin_out) {
    hipStream_t s = ctx->stream;
    DecodePlan P;
    int rc = decode_prepare(ctx, d_file, file_len, first_block, n_blocks, &P);
    if (rc) return rc;
    if (P.n_bcf != n_lines)
        return set_error(XSI_ERR_ARG, "decode_dot_gt: blocks hold %u BCF lines, caller passed %llu", P.n_bcf, (unsigned long long)n_lines);
    const uint32_t n_bcf = P.n_bcf, N = P.L.N;
    std::vector<uint32_t> first_bin(n_bcf), parent(P.n_bin), alt(P.n_bin);
    {
        uint32_t l = 0;
        for (uint32_t b = 0; b < P.n_blocks; ++b) {
            uint32_t acc = P.blocks_h[b].first_bin;
            const uint32_t end = acc + P.blocks_h[b].n_bin;
            for (uint32_t i = 0; i < P.blocks_h[b].n_bcf; ++i, ++l) {
                if (h_n_allele[l] < 2) return set_error(XSI_ERR_ARG, "decode_dot_gt: line %u has fewer than 2 alleles", l);
                first_bin[l] = acc;
                for (uint32_t k = 1; k < h_n_allele[l] && acc < end; ++k, ++acc) {
                    parent[acc] = l;
                    alt[acc] = k;
                }
                if (acc - first_bin[l] != h_n_allele[l] - 1) acc = end + 1;  // overran the block: caught below
            }
            if (acc != end)
                return set_error(XSI_ERR_ARG, "decode_dot_gt: allele numbers of block %u do not add up to its %u binary lines", b,
                                 P.blocks_h[b].n_bin);
        }
    }
    uint32_t *d_first_bin, *d_nallele, *d_line_ngt, *d_parent, *d_alt;
    int32_t* d_rows;
    WS(d_first_bin, "gt.bcf_first_bin", 4ull * n_bcf);
    WS(d_nallele, "gt.bcf_nallele", 4ull * n_bcf);
    WS(d_line_ngt, "gt.line_ngt", 4ull * n_bcf);
    WS(d_parent, "dot.bin_parent", 4ull * P.n_bin + 4);
    WS(d_alt, "dot.bin_alt", 4ull * P.n_bin + 4);
    WS(d_rows, "dot.rows", 4ull * N * (size_t)(n_bcf ? n_bcf : 1));
    HIP_TRY(hipMemcpyAsync(d_first_bin, first_bin.data(), 4ull * n_bcf, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(d_nallele, h_n_allele, 4ull * n_bcf, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(d_parent, parent.data(), 4ull * P.n_bin, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(d_alt, alt.data(), 4ull * P.n_bin, hipMemcpyHostToDevice, s));
    HIP_TRY(hipStreamSynchronize(s));  // local vectors: the copies must have left host memory
    DecodedPlanes DPn;
    rc = decode_all_planes(ctx, d_file, P, &DPn);
    if (rc) return rc;
    rc = compose_lines(ctx, P, DPn, d_first_bin, d_nallele, n_bcf, d_rows, N, d_line_ngt, nullptr, 0);
    if (rc) return rc;
    if (P.n_bin) {
        const dim3 grid((P.n_bin + 3u) / 4u), block(256);
        uint32_t k0 = 0;
        for (; k0 + 4u <= n_pheno; k0 += 4u)
            k_dot_gt<4><<<grid, block, 0, s>>>(d_rows, N, d_line_ngt, d_parent, d_alt, P.n_bin, P.L.n_samples, d_pheno, n_pheno, k0, d_out);
        for (; k0 < n_pheno; ++k0)
            k_dot_gt<1><<<grid, block, 0, s>>>(d_rows, N, d_line_ngt, d_parent, d_alt, P.n_bin, P.L.n_samples, d_pheno, n_pheno, k0, d_out);
        HIP_TRY(hipGetLastError());
    }
    HIP_TRY(hipStreamSynchronize(s));
    *n_bin_out = P.n_bin;
    return XSI_OK;
}

int xsi_hip_decode_dot_gt(xsi_hip_ctx* ctx, const void* d_file, uint64_t file_len, uint64_t first_block, uint64_t n_blocks,
                          const uint32_t* h_n_allele, uint64_t n_lines, const double* d_pheno, uint32_t n_pheno,
                          double* d_out, uint64_t capacity, uint64_t* h_n_bin) {
    if (!ctx || !d_file || !h_n_allele || !d_pheno || !d_out) return set_error(XSI_ERR_ARG, "decode_dot_gt: null argument");
    if (!n_pheno) return set_error(XSI_ERR_ARG, "decode_dot_gt: n_pheno must be > 0");
    if (file_len < 256) return set_error(XSI_ERR_FORMAT, "file image shorter than the 256-byte header");
    HIP_TRY(hipSetDevice(ctx->device));
    uint64_t n_bin_total = 0;
    for (uint64_t l = 0; l < n_lines; ++l) {
        if (h_n_allele[l] < 2) return set_error(XSI_ERR_ARG, "decode_dot_gt: line %llu has fewer than 2 alleles", (unsigned long long)l);
        n_bin_total += h_n_allele[l] - 1;
    }
    if (n_bin_total > capacity)
        return set_error(XSI_ERR_CAPACITY, "decode_dot_gt: %llu binary lines, capacity %llu", (unsigned long long)n_bin_total,
                         (unsigned long long)capacity);
    // the int32 rows of a range of blocks live in the workspace: walk the blocks in ranges that fit its budget
    DecodePlan P;
    int rc = decode_prepare(ctx, d_file, file_len, first_block, n_blocks, &P);
    if (rc) return rc;
    if (P.n_bcf != n_lines)
        return set_error(XSI_ERR_ARG, "decode_dot_gt: blocks hold %u BCF lines, caller passed %llu", P.n_bcf, (unsigned long long)n_lines);
    std::vector<uint32_t> block_bcf(P.n_blocks);
    for (uint32_t b = 0; b < P.n_blocks; ++b) block_bcf[b] = P.blocks_h[b].n_bcf;
    const uint64_t row_bytes = 4ull * P.L.N + 3ull * 4ull * P.L.y_stride64 * 2ull;  // int32 row + up to 3 planes per line
    const uint64_t budget_lines = std::max<uint64_t>(1, ws_budget_now(ctx) / 2 / row_bytes);
    uint64_t done_bin = 0, line0 = 0;
    for (uint32_t b0 = 0; b0 < (uint32_t)block_bcf.size();) {
        uint32_t b1 = b0;
        uint64_t lines = 0;
        while (b1 < block_bcf.size() && (b1 == b0 || lines + block_bcf[b1] <= budget_lines)) lines += block_bcf[b1++];
        uint64_t nb = 0;
        rc = dot_gt_range(ctx, d_file, file_len, first_block + b0, b1 - b0, h_n_allele + line0, lines, d_pheno, n_pheno,
                          d_out + done_bin * n_pheno, &nb);
        if (rc) return rc;
        done_bin += nb;
        line0 += lines;
        b0 = b1;
    }
    if (h_n_bin) *h_n_bin = done_bin;
    stage_collect(ctx);
    return XSI_OK;
}


int xsi_hip_reencode(xsi_hip_ctx* ctx, const void* d_file, uint64_t file_len, const uint32_t* h_n_allele, uint64_t n_lines,
                     const xsi_encode_params* p_new, const uint32_t* h_sample_idx, uint32_t n_sel, void* d_out,
                     uint64_t out_capacity, uint64_t* d_block_offsets, xsi_encode_result* h_result) {
    if (!ctx || !d_file || !h_n_allele || !p_new || !d_out) return set_error(XSI_ERR_ARG, "reencode: null argument");
    if (file_len < 256) return set_error(XSI_ERR_FORMAT, "file image shorter than the 256-byte header");
    HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    uint8_t h[256];
    HIP_TRY(hipMemcpyAsync(h, d_file, 256, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    auto get = [&](size_t off, int bytes) {
        uint64_t v = 0;
        for (int i = 0; i < bytes; ++i) v |= (uint64_t)h[off + i] << (8 * i);
        return v;
    };
    const uint64_t io = get(72, 8), so = get(80, 8);
    const uint64_t version = get(8, 4);
    if (so < io) return set_error(XSI_ERR_FORMAT, "index outside the file image");
    const uint64_t n_blocks = (so - io) / (version >= 5 ? 8 : 4);
    const uint64_t ns = get(112, 8), N = ns ? ns * 2 : get(32, 8);
    const uint32_t n_src = (uint32_t)(N / 2);
    const uint32_t n_dst = h_sample_idx ? n_sel : n_src;
    if (p_new->n_samples != n_dst)
        return set_error(XSI_ERR_ARG, "reencode: params say %u samples, the %s has %u", p_new->n_samples,
                         h_sample_idx ? "selection" : "file", n_dst);
    if (h_sample_idx)
        for (uint32_t i = 0; i < n_sel; ++i)
            if (h_sample_idx[i] >= n_src) return set_error(XSI_ERR_ARG, "reencode: sample %u of %u", h_sample_idx[i], n_src);
    // Decode to int32 rows in HBM, (optionally) gather the selected samples there, encode again - in RANGES of whole
    // source blocks sized by the workspace budget, so a file of any size goes through (a 64 976 x 2 M file has 520 GB of
    // rows).  The source's blocks and the new blocks need not line up: the rows of a range are staged behind what the
    // previous range left over, whole new blocks are encoded from the front of the staging buffer (their bytes and
    // offsets continue the region: blocks are independent, gt_block.hpp:179-180), the remainder (< one new block)
    // moves to the front.  gt_decompressor_new.hpp:241-273 does the same one line at a time.
    DecodePlan plan;
    int rc = decode_prepare(ctx, d_file, file_len, 0, n_blocks, &plan, /*counts_only=*/true);
    if (rc) return rc;
    if (plan.n_bcf != n_lines)
        return set_error(XSI_ERR_ARG, "reencode: the file holds %u BCF lines, caller passed %llu", plan.n_bcf, (unsigned long long)n_lines);
    std::vector<uint32_t> src_lines(n_blocks);
    uint32_t max_src = 1;
    for (uint64_t b = 0; b < n_blocks; ++b) {
        src_lines[b] = plan.blocks_h[b].n_bcf;
        if (src_lines[b] > max_src) max_src = src_lines[b];
    }
    const uint64_t bl_new = p_new->block_len;
    if (!bl_new) return set_error(XSI_ERR_ARG, "reencode: block_len must be > 0");
    // per staged line: its int32 row, the selected copy, and about half a row of bit planes on either side of the codec
    const uint64_t per_line = 4ull * N + (h_sample_idx ? 8ull * n_sel : 0ull) + 2ull * N;
    uint64_t cap_lines = ws_budget_now(ctx) / per_line;
    const uint64_t min_lines = (bl_new - 1u) + max_src;  // a leftover plus one source block: always makes progress
    if (cap_lines < min_lines) cap_lines = min_lines;
    if (cap_lines > n_lines) cap_lines = n_lines;
    int32_t* d_rows;
    WS(d_rows, "reenc.rows", 4ull * N * (size_t)cap_lines);
    uint32_t *d_sel = nullptr, *d_ngt = nullptr;
    int32_t* d_sub = nullptr;
    if (h_sample_idx) {
        WS(d_sel, "reenc.sel", 4ull * n_sel);
        WS(d_ngt, "reenc.ngt", 4ull * cap_lines);
        WS(d_sub, "reenc.sub", 8ull * n_sel * (size_t)cap_lines);
        HIP_TRY(hipMemcpyAsync(d_sel, h_sample_idx, 4ull * n_sel, hipMemcpyHostToDevice, s));
        HIP_TRY(hipStreamSynchronize(s));
    }
    std::vector<uint32_t> ngt(cap_lines), ngt_enc(cap_lines);
    xsi_encode_result total{};
    uint64_t staged = 0;      // lines in d_rows
    uint64_t line_in = 0;     // source lines decoded so far
    uint64_t line_out = 0;    // lines encoded so far
    uint64_t region_off = 0, blocks_out = 0, src_b = 0;
    uint32_t ranges = 0;
    while (line_out < n_lines) {
        // fill: as many whole source blocks as fit behind the leftover
        uint64_t take_b = 0, take_l = 0;
        while (src_b + take_b < n_blocks && staged + take_l + src_lines[src_b + take_b] <= cap_lines) take_l += src_lines[src_b + take_b++];
        if (take_b) {
            rc = xsi_hip_decode_gt(ctx, d_file, file_len, src_b, take_b, h_n_allele + line_in, take_l, d_rows + staged * N, N,
                                   ngt.data() + staged, nullptr, 0);
            if (rc) return rc;
            staged += take_l;
            line_in += take_l;
            src_b += take_b;
        }
        // encode: whole new blocks from the front (everything once the source is exhausted)
        const uint64_t m = src_b == n_blocks ? staged : staged / bl_new * bl_new;
        if (!m) return set_error(XSI_ERR_CAPACITY, "reencode: the staging buffer of %llu lines cannot hold one new block of %llu",
                                 (unsigned long long)cap_lines, (unsigned long long)bl_new);
        const int32_t* d_enc = d_rows;
        uint64_t stride = N;
        for (uint64_t l = 0; l < m; ++l) ngt_enc[l] = ngt[l];
        if (h_sample_idx) {
            HIP_TRY(hipMemcpyAsync(d_ngt, ngt.data(), 4ull * m, hipMemcpyHostToDevice, s));
            rc = select_samples(ctx, d_rows, N, d_ngt, (uint32_t)m, n_src, d_sel, n_sel, d_sub, 2ull * n_sel, nullptr, 0);
            if (rc) return rc;
            HIP_TRY(hipStreamSynchronize(s));
            for (uint64_t l = 0; l < m; ++l) ngt_enc[l] = ngt[l] / n_src * n_sel;
            d_enc = d_sub;
            stride = 2ull * n_sel;
        }
        if (region_off > out_capacity) return set_error(XSI_ERR_CAPACITY, "reencode: output capacity exhausted");
        xsi_encode_result r{};
        rc = encode_gt_impl(ctx, p_new, d_enc, stride, m, ngt_enc.data(), h_n_allele + line_out, static_cast<uint8_t*>(d_out) + region_off,
                            out_capacity - region_off, d_block_offsets ? d_block_offsets + blocks_out : nullptr, &r, region_off);
        if (rc) return rc;
        region_off += r.blocks_bytes;
        blocks_out += r.n_blocks;
        total.n_blocks += r.n_blocks;
        total.n_binary_lines += r.n_binary_lines;
        total.n_wah_lines += r.n_wah_lines;
        if (r.max_ploidy > total.max_ploidy) total.max_ploidy = r.max_ploidy;
        total.last_block_bytes = r.last_block_bytes;
        line_out += m;
        // the leftover (< one new block, so it cannot overlap its destination) moves to the front
        const uint64_t rem = staged - m;
        if (rem) {
            HIP_TRY(hipMemcpyAsync(d_rows, d_rows + m * N, 4ull * N * rem, hipMemcpyDeviceToDevice, s));
            HIP_TRY(hipStreamSynchronize(s));
            for (uint64_t l = 0; l < rem; ++l) ngt[l] = ngt[m + l];
        }
        staged = rem;
        ++ranges;
    }
    total.blocks_bytes = region_off;
    ctx->reencode_ranges = ranges;
    if (h_result) *h_result = total;
    return XSI_OK;
}

uint32_t xsi_hip_ctx_reencode_ranges(const xsi_hip_ctx* ctx) { return ctx ? ctx->reencode_ranges : 0u; }

}  // extern "C"
