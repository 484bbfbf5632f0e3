// xsi_device.hpp — wave64 device helpers shared by the gfx950 kernels.
//
// Everything here is written for CDNA4's 64-wide wavefronts: ballots are 64-bit, lane prefix
// counts use v_mbcnt, cross-lane moves use ds_bpermute/DPP through __shfl*.  No MFMA: the whole
// path is integer bit manipulation bound by HBM / LDS.
#pragma once

#include <hip/hip_runtime.h>

#include <type_traits>
#include <stdint.h>

namespace xsi {

constexpr uint32_t WAH_BITS = 15;
constexpr uint32_t WAH_MAXC = 0x3FFF;  // 16383 groups per fill word (wah.hpp:383)

// compile-time loop
template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

// v_writelane_b32: put a wave-uniform value into one lane of a VGPR.  clang has no builtin for it;
// binding the LLVM intrinsic by name keeps the compiler in charge of the SGPR hazards (an inline
// asm version of this produced wrong rows).
extern "C" __device__ uint32_t __xsi_writelane_u32(uint32_t value, uint32_t lane, uint32_t old)
    __asm("llvm.amdgcn.writelane.i32");
__device__ __forceinline__ uint32_t write_lane(uint32_t old, uint32_t uniform_value, uint32_t lane) {
    return __xsi_writelane_u32(uniform_value, lane, old);
}

__device__ __forceinline__ uint32_t lane_id() { return threadIdx.x & 63u; }

// Workgroup barrier for LDS hand-offs that involve 16-bit stores or LDS atomics.  hipcc (ROCm 7.2) emits
// no s_waitcnt lgkmcnt before the s_barrier of __syncthreads(): it assumes the LDS executes the
// operations of all waves of a workgroup in one total order.  On MI355X that did not hold for
// ds_write_b16 followed, behind the barrier, by another wave's ds_read of the same dword: about one
// run in three of a 3000-line decode read a stale member (tools history: r02, xsi_pair.hip).  Waiting
// for this wave's own LDS operations to complete before it arrives at the barrier removes the window.
__device__ __forceinline__ void lds_barrier() {
    __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0); vmcnt / expcnt untouched
    __syncthreads();
}

// popcount(mask & lanes below me)
__device__ __forceinline__ uint32_t mbcnt64(uint64_t m) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}

// Inclusive wave scan (sum) of a 32-bit value over 64 lanes.
__device__ __forceinline__ uint32_t wave_scan_incl(uint32_t v) {
    const uint32_t lane = lane_id();
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t o = __shfl_up(v, d, 64);
        if (lane >= (uint32_t)d) v += o;
    }
    return v;
}
// 16-lane inclusive prefix sum with DPP row shifts (v_add_u32_dpp row_shr:1/2/4/8, bound_ctrl:0).
__device__ __forceinline__ uint32_t row16_scan_incl(uint32_t v) {
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, true);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, true);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, true);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, true);
    return v;
}

// 64-lane inclusive prefix sum: 16-lane rows with row_shr, then row_bcast:15 / row_bcast:31
// (gfx9 DPP) carry the row totals forward.
__device__ __forceinline__ uint32_t wave_scan_incl_dpp(uint32_t v) {
    v = row16_scan_incl(v);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, false);  // row_bcast:15 -> rows 1,3
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, false);  // row_bcast:31 -> rows 2,3
    return v;
}

// 64-lane inclusive prefix maximum (unsigned; lanes outside a row read 0, the identity)
__device__ __forceinline__ uint32_t wave_scan_max_incl_dpp(uint32_t v) {
    auto mx = [](uint32_t a, int b) { return a > (uint32_t)b ? a : (uint32_t)b; };
    v = mx(v, __builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, true));
    v = mx(v, __builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, true));
    v = mx(v, __builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, true));
    v = mx(v, __builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, true));
    v = mx(v, __builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, false));  // row_bcast:15 -> rows 1,3
    v = mx(v, __builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, false));  // row_bcast:31 -> rows 2,3
    return v;
}

__device__ __forceinline__ uint64_t wave_scan_incl64(uint64_t v) {
    const uint32_t lane = lane_id();
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint64_t o = __shfl_up(v, d, 64);
        if (lane >= (uint32_t)d) v += o;
    }
    return v;
}
__device__ __forceinline__ uint32_t wave_sum(uint32_t v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}

// Workgroup exclusive scan of one uint64 per thread.  `lds` needs (blockDim/64 + 1) uint64.
// Returns the exclusive prefix; *total receives the workgroup sum.  Contains two barriers.
__device__ __forceinline__ uint64_t block_scan_excl64(uint64_t v, uint64_t* lds, uint64_t* total) {
    const uint32_t lane = lane_id();
    const uint32_t w = threadIdx.x >> 6;
    const uint32_t nw = (blockDim.x + 63u) >> 6;
    uint64_t inc = wave_scan_incl64(v);
    if (lane == 63) lds[w] = inc;
    __syncthreads();
    uint64_t base = 0, tot = 0;
    for (uint32_t i = 0; i < nw; ++i) {
        uint64_t c = lds[i];
        if (i < w) base += c;
        tot += c;
    }
    __syncthreads();
    *total = tot;
    return base + inc - v;
}

// 15-bit group g of a packed bit row (LSB-first in little-endian 32-bit words); bits at or
// beyond nbits read as zero (the reference pads the last group with zeros, wah.hpp:547-565).
__device__ __forceinline__ uint32_t load_group15(const uint32_t* __restrict__ row, uint32_t g, uint32_t nbits) {
    // branch-free: word indices are clamped into the row, bits at or beyond nbits are masked off
    // (g may lie beyond the last group: the result is then 0)
    const uint32_t last = ((nbits + 31u) >> 5) - 1u;  // nbits >= 1
    const uint32_t o = g * WAH_BITS;
    const uint32_t wi = (o >> 5) < last ? (o >> 5) : last;
    const uint32_t wi1 = wi < last ? wi + 1u : last;
    const uint32_t v = __builtin_amdgcn_alignbit(row[wi1], row[wi], o & 31u);  // (hi:lo) >> (o & 31)
    const uint32_t rem = o < nbits ? nbits - o : 0u;
    const uint32_t keep = rem < WAH_BITS ? rem : WAH_BITS;
    return v & ((1u << keep) - 1u);
}

// State carried between 64-group chunks of one WAH16 line.
struct WahCarry {
    uint32_t type;  // 0 zeros run open, 1 ones run open, 2/3 nothing open
    uint32_t len;   // groups in the open run
    uint32_t out;   // words emitted so far
};

// Encode one chunk of up to 64 groups held one-per-lane.  `val` = 15-bit group value, `valid`
// = lane holds a group, `last_chunk` = no group follows this chunk.  Reproduces
// process_wah_word + the trailing flush (wah.hpp:376-429, 567-573): literals are copied, runs of
// all-zero / all-one groups collapse into fill words of at most 16383 groups, a saturated run
// emits 0xBFFF / 0xFFFF and restarts.  Emits at the END of each run, so every word's position
// is the exclusive prefix of the emit counts.
// `next_val` = value of the group that follows this chunk (ignored when last_chunk): a run that
// fills the chunk up to its last lane ends there iff the next group differs.
template <bool WRITE>
__device__ __forceinline__ void wah_encode_chunk(uint32_t val, bool valid, bool last_chunk, uint32_t next_val,
                                                 WahCarry& c, uint16_t* __restrict__ dst) {
    const uint32_t lane = lane_id();
    const uint32_t t = (val == 0u) ? 0u : ((val == 0x7FFFu) ? 1u : 2u);
    // type of the previous lane: DPP wave_shr:1, lane 0 takes the carry
    const uint32_t prev_t = (uint32_t)__builtin_amdgcn_update_dpp((int)c.type, (int)t, 0x138, 0xF, 0xF, false);
    const bool head = valid && (t == 2u || t != prev_t);
    const uint64_t H = __ballot(head);
    const uint64_t V = __ballot(valid);
    const uint32_t nvalid = (uint32_t)__popcll(V);
    // run length ending at this lane (only meaningful for t < 2)
    const uint64_t le = (lane == 63) ? ~0ull : ((2ull << lane) - 1ull);
    const uint64_t hb = H & le;
    uint32_t len;
    if (hb) {
        len = lane - (63u - (uint32_t)__clzll((long long)hb)) + 1u;
    } else {
        len = c.len + lane + 1u;
    }
    // is this lane the last group of its run?
    bool is_end;
    if (lane + 1u < nvalid) {
        is_end = (H >> (lane + 1u)) & 1ull;
    } else {
        // last valid lane: the run stays open only if the next chunk continues it
        const uint32_t nt = (next_val == 0u) ? 0u : ((next_val == 0x7FFFu) ? 1u : 2u);
        is_end = last_chunk || nt == 2u || nt != t;
    }
    // a run longer than 16383 groups (needs > 245 745 bits) emits several fill words: rare, and
    // the only case that needs a division and a full scan
    uint32_t emit = (valid && (t == 2u || is_end)) ? 1u : 0u;
    uint32_t pos, total;
    if (!__any(emit && t < 2u && len > WAH_MAXC)) {
        const uint64_t E = __ballot(emit == 1u);
        pos = mbcnt64(E);
        total = (uint32_t)__popcll(E);
    } else {
        if (emit && t < 2u) emit = (len + WAH_MAXC - 1u) / WAH_MAXC;
        const uint32_t inc = wave_scan_incl_dpp(emit);
        pos = inc - emit;
        total = (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
    }
    if (WRITE && emit) {
        uint16_t* o = dst + c.out + pos;
        if (t == 2u) {
            o[0] = (uint16_t)val;
        } else {
            const uint32_t tag = 0x8000u | (t << 14);
            for (uint32_t k = 0; k + 1u < emit; ++k) o[k] = (uint16_t)(tag | WAH_MAXC);
            o[emit - 1u] = (uint16_t)(tag | (len - WAH_MAXC * (emit - 1u)));
        }
    }
    // carry = state after the last valid lane
    const uint32_t last = nvalid ? nvalid - 1u : 0u;
    const uint32_t lt = (uint32_t)__builtin_amdgcn_readlane((int)t, (int)last);
    const uint32_t ll = (uint32_t)__builtin_amdgcn_readlane((int)len, (int)last);
    if (nvalid) {
        c.type = lt;
        c.len = (lt < 2u) ? ll : 0u;
    }
    c.out += total;
}

// Encode a whole packed bit row with one wave.  Returns the number of WAH16 words.  The groups of
// the next two chunks are loaded before the current chunk is encoded, so the serial run-merging
// logic never waits on memory.
template <bool WRITE>
__device__ __forceinline__ uint32_t wave_wah_encode_row(const uint32_t* __restrict__ row, uint32_t nbits,
                                                        uint16_t* __restrict__ dst) {
    const uint32_t lane = lane_id();
    const uint32_t G = (nbits + WAH_BITS - 1u) / WAH_BITS;
    WahCarry c{3u, 0u, 0u};
    uint32_t cur = load_group15(row, lane, nbits);
    uint32_t n1 = load_group15(row, lane + 64u, nbits);
    for (uint32_t g0 = 0; g0 < G; g0 += 64u) {
        const uint32_t n2 = load_group15(row, g0 + 128u + lane, nbits);
        const bool last = g0 + 64u >= G;
        const uint32_t next_val = (uint32_t)__builtin_amdgcn_readfirstlane((int)n1);  // group g0+64 (0 past the end)
        wah_encode_chunk<WRITE>(cur, g0 + lane < G, last, next_val, c, dst);
        cur = n1;
        n1 = n2;
    }
    return c.out;
}

// ------------------------------------------------------------------------------------------
// WAH16 of a row staged in LDS, lines of at most 16383 groups (no run can outgrow a fill word), without a
// serial pass over the line.  The line is cut into UNITS of 32 groups = 480 bits = 15 words; a lane owns unit
// r*64 + lane in round r.  Inside a unit everything is per-lane bit logic on 32-bit masks: Z / O = groups
// that are all zeros / all ones, and a group is a HEAD (starts a word) when it is a literal or its predecessor
// (the last group of the unit before, read directly from the row) is not a fill of its type.  The number of
// words of a line is the number of heads; a fill word's count is the distance to the next head.  Units whose
// lanes sit 15 words apart read LDS without bank conflicts.
// ------------------------------------------------------------------------------------------
using LdsCU32 = const __attribute__((address_space(3))) uint32_t;
using LdsU32W = __attribute__((address_space(3))) uint32_t;
constexpr int WAH_UNIT_ROUNDS = 3;  // 3 x 64 units x 32 groups = 6144 groups >= ceil(65 536 / 15)
constexpr uint32_t WAH_UNIT_ROW_WORDS = 64u * 15u * (uint32_t)WAH_UNIT_ROUNDS;
// (hand-scheduled wave64 assembly whose SGPR write-to-read spacing is gfx950's; the hazard recognizer does not look inside
// inline asm, so building it for any other target is refused rather than left to chance - ADVICE r4)
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "xsi_wah_classify.inc is scheduled by hand for gfx950: regenerate it (tools/gen_wah_classify.py) for another target"
#endif
#include "xsi_wah_classify.inc"
struct WahUnit {
    uint32_t H, F, O;  // heads, fill groups, all-ones groups of my unit (bit k = group k of the unit)
};
// `ua`: the unit whose words are READ (callers whose lanes may hold a unit behind the line's last one pass the last
// one here: such a lane's masks come out zero whatever it reads, but it must not read behind the row's LDS)
__device__ __forceinline__ void wah_unit_classify(LdsCU32* row, uint32_t u, uint32_t G, WahUnit& m, uint32_t ua) {
    const uint32_t gb = u * 32u;
    uint32_t w[15];
#pragma unroll
    for (int i = 0; i < 15; ++i) w[i] = row[ua * 15u + (uint32_t)i];
    const uint32_t prev = row[ua ? ua * 15u - 1u : 0u] >> 17;  // the group before the unit (u > 0)
    // bit k: group k is not all zeros / is all ones.  Five vector instructions a group, hand-scheduled
    // (xsi_wah_classify.inc, generated by tools/gen_wah_classify.py; the compiler's form of "extract, compare twice, set
    // bit k" came to about 7.5 + two s_nop: k_wah_units is bound by vector issue and a third of it was this)
    uint32_t nz, on;
    wah_unit_flags(w, nz, on);
    const uint32_t nv = G > gb ? G - gb : 0u;
    const uint32_t Vm = nv >= 32u ? ~0u : ((1u << nv) - 1u);
    const uint32_t Z = ~nz & Vm, O = on & Vm;
    const uint32_t pz = (u != 0u && prev == 0u) ? 1u : 0u, po = (u != 0u && prev == 0x7FFFu) ? 1u : 0u;
    const uint32_t Zs = Z & ((Z << 1) | pz), Os = O & ((O << 1) | po);
    m.H = Vm & ~(Zs | Os);
    m.F = Z | O;
    m.O = O;
}
// Masks of every unit of the line (one call per wave); returns the number of WAH16 words of the line.
__device__ __forceinline__ uint32_t wah_units_classify_line(LdsCU32* row, uint32_t G, WahUnit (&m)[WAH_UNIT_ROUNDS]) {
    const uint32_t lane = lane_id();
    const uint32_t units = (G + 31u) >> 5, rounds = (units + 63u) >> 6;
    uint32_t cnt = 0;
#pragma unroll
    for (int r = 0; r < WAH_UNIT_ROUNDS; ++r) {
        m[r] = WahUnit{0u, 0u, 0u};
        if ((uint32_t)r < rounds) {  // wave-uniform
            const uint32_t u = (uint32_t)r * 64u + lane;
            wah_unit_classify(row, u, G, m[r], u < units ? u : units - 1u);
            cnt += (uint32_t)__popc(m[r].H);
        }
    }
    const uint32_t inc = wave_scan_incl_dpp(cnt);
    return (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
}
// position of the k-th (0-based) set bit of x, k < popcount(x)
__device__ __forceinline__ uint32_t select_bit32(uint32_t x, uint32_t k) {
    uint32_t pos = 0, t;
    t = (uint32_t)__popc(x & 0xFFFFu);
    if (k >= t) { k -= t; pos = 16u; x >>= 16; }
    t = (uint32_t)__popc(x & 0xFFu);
    if (k >= t) { k -= t; pos += 8u; x >>= 8; }
    t = (uint32_t)__popc(x & 0xFu);
    if (k >= t) { k -= t; pos += 4u; x >>= 4; }
    t = (uint32_t)__popc(x & 0x3u);
    if (k >= t) { k -= t; pos += 2u; x >>= 2; }
    if (k >= (x & 1u)) pos += 1u;
    return pos;
}
// Emit the words of the line to dst (2-byte aligned).  `fh` = LDS scratch of 64 * WAH_UNIT_ROUNDS + 32 words.
// WORD-major: lane i of a step forms word i of the round.  The heads are spread very unevenly over the units (2.4 a
// unit on average at configs[2], ~30 in the busiest of a round), so a loop in which every unit emits its own heads runs
// as long as the busiest unit at a few percent lane use.  A word's unit: the units whose first word falls into the
// step's 64 words leave their lane number at that word's slot of a 64-byte LDS strip, a prefix maximum over the lanes
// spreads it (carry: the unit of the step before); the unit's masks come over ds_bpermute, the head inside the unit
// by select_bit32; the 64 words of a step leave in one 128-byte store.
__device__ __forceinline__ void wah_units_emit_line(LdsCU32* row, LdsU32W* fh, uint32_t G, const WahUnit (&m)[WAH_UNIT_ROUNDS],
                                                    uint16_t* __restrict__ dst) {
    const uint32_t lane = lane_id();
    const uint32_t rounds = (((G + 31u) >> 5) + 63u) >> 6;
    uint64_t B[WAH_UNIT_ROUNDS];  // units with at least one head, per round
#pragma unroll
    for (int r = 0; r < WAH_UNIT_ROUNDS; ++r) {
        B[r] = __ballot(m[r].H != 0u);
        fh[(uint32_t)r * 64u + lane] = ((uint32_t)r * 64u + lane) * 32u + (uint32_t)__builtin_ctz(m[r].H | 0x80000000u);
    }
    const uint64_t above = (lane == 63u) ? 0ull : (~0ull << (lane + 1u));
    // dst is rebuilt from two v_readlane halves by the caller: say that it is global memory, or the stores are flat
    // stores, which count in lgkmcnt too and are then waited for by every LDS wait that follows
    __attribute__((address_space(1))) uint16_t* gdst = (__attribute__((address_space(1))) uint16_t*)dst;
    using LdsU8W = __attribute__((address_space(3))) uint8_t;
    LdsU8W* mk = reinterpret_cast<LdsU8W*>(fh + 64u * (uint32_t)WAH_UNIT_ROUNDS);
    uint32_t round_off = 0;
#pragma unroll
    for (int r = 0; r < WAH_UNIT_ROUNDS; ++r) {
        if ((uint32_t)r >= rounds) break;  // wave-uniform
        // first head after my unit: a later lane of this round, else the first unit with a head of a later round
        uint32_t later = ~0u;  // wave-uniform
#pragma unroll
        for (int q = WAH_UNIT_ROUNDS - 1; q > r; --q)
            if (B[q]) later = (uint32_t)q * 64u + (uint32_t)__builtin_ctzll(B[q]);
        const uint64_t mine = B[r] & above;
        const uint32_t tu = mine ? (uint32_t)r * 64u + (uint32_t)__builtin_ctzll(mine) : later;
        const uint32_t nh = tu != ~0u ? fh[tu] : G;
        const uint32_t cnt = (uint32_t)__popc(m[r].H);
        const uint32_t inc = wave_scan_incl_dpp(cnt);
        const uint32_t excl = inc - cnt;
        const uint32_t Wr = (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);  // words of this round
        uint32_t carry = 0;  // unit lane + 1 of the last word of the step before
        // Two steps of 64 words per iteration (one strip of 128 markers): the second step's chain of dependent LDS
        // operations - markers, prefix maximum, three ds_bpermute, the head's select, the literal's two words - runs in
        // the gaps of the first one's; only the prefix maxima are chained (the first step's last lane seeds the second).
        for (uint32_t t0 = 0; t0 < Wr; t0 += 128u) {
            mk[lane] = 0;
            mk[64u + lane] = 0;
            if (cnt && excl - t0 < 128u) mk[excl - t0] = (uint8_t)(lane + 1u);  // (excl < t0 wraps to a large number)
            asm volatile("" ::: "memory");  // other lanes' stores: no forwarding of my own zero to the loads below
            uint32_t u2[2];
            u2[0] = wave_scan_max_incl_dpp((uint32_t)mk[lane]);  // one wave: its LDS operations stay in order
            u2[1] = wave_scan_max_incl_dpp((uint32_t)mk[64u + lane]);
            u2[0] = u2[0] > carry ? u2[0] : carry;
            carry = (uint32_t)__builtin_amdgcn_readlane((int)u2[0], 63);
            u2[1] = u2[1] > carry ? u2[1] : carry;
            carry = (uint32_t)__builtin_amdgcn_readlane((int)u2[1], 63);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const uint32_t ul = u2[h] - 1u;  // u >= 1: word 0 of a round is the first word of a unit
                const uint32_t t = t0 + 64u * (uint32_t)h + lane;
                const int ua = (int)(ul << 2);
                const uint32_t Hs = (uint32_t)__builtin_amdgcn_ds_bpermute(ua, (int)m[r].H);
                const uint32_t ex = (uint32_t)__builtin_amdgcn_ds_bpermute(ua, (int)excl);
                const uint32_t nhs = (uint32_t)__builtin_amdgcn_ds_bpermute(ua, (int)nh);
                const uint32_t kk = t - ex;  // lanes beyond the round: anything, nothing is stored
                const uint32_t k = select_bit32(Hs, kk < 32u ? kk : 31u) & 31u;
                const uint32_t g = ((uint32_t)r * 64u + ul) * 32u + k;
                const uint32_t rest = (Hs >> k) >> 1;
                const uint32_t nxt = rest ? g + 1u + (uint32_t)__builtin_ctz(rest) : nhs;
                const uint32_t o = g * WAH_BITS;  // (a lane beyond the round holds some head of the round's last unit)
                const uint32_t lit = __builtin_amdgcn_alignbit(row[(o >> 5) + 1u], row[o >> 5], o & 31u) & 0x7FFFu;
                // a head is a fill exactly when its own group is all zeros or all ones: read off the literal the lane
                // fetches anyway (the unit's F and O masks came over two more ds_bpermute before)
                const bool all1 = lit == 0x7FFFu;
                const uint32_t fill = (all1 ? 0xC000u : 0x8000u) | (nxt - g);
                if (t < Wr) gdst[round_off + t] = (uint16_t)((lit == 0u || all1) ? fill : lit);
            }
        }
        round_off += Wr;
    }
}

// Expand one WAH16 line into a zeroed packed row held in LDS (wah2_extract_template,
// wah.hpp:177-223).  One wave.  `src` is 2-byte aligned; at most `max_words` words may be
// read.  Returns the words consumed; *ones = set bits counted like the reference (fills count
// whole groups).  The caller must barrier before reading `row`.  row == nullptr: count only
// (wah2_advance_pointer_count_ones, wah.hpp:125-150).
// `pre0` = src[lane] (0 where lane >= max_words), loaded by the caller ahead of time; the words of
// the next iteration are fetched before the current ones are used (an iteration that does not end
// the line consumes exactly 64 words, so the prefetch is never wasted).
__device__ __forceinline__ uint32_t wave_wah_expand_row(const uint16_t* __restrict__ src, uint32_t max_words,
                                                        uint32_t nbits, uint32_t* row /*LDS*/, uint32_t* ones,
                                                        uint32_t pre0) {
    const uint32_t lane = lane_id();
    const uint32_t G = (nbits + WAH_BITS - 1u) / WAH_BITS;
    const uint32_t row_bits = ((nbits + 31u) >> 5) << 5;
    uint32_t gbase = 0, wbase = 0, cnt1 = 0;
    uint32_t cur = pre0;
    // The words are in global memory; said so explicitly, because a pointer rebuilt from two v_readlane halves is
    // "generic" to the compiler and its loads become flat_load (counted in vmcnt AND lgkmcnt: every wait for one
    // also drains the LDS queue).  The look-ahead load is unconditional (clamped index, `have` discards the value):
    // under a condition with a default it is waited for where it is issued.
    const __attribute__((address_space(1))) uint16_t* gsrc = (const __attribute__((address_space(1))) uint16_t*)src;
    while (gbase < G && wbase < max_words) {
        const uint32_t wi = wbase + lane;
        const bool have = wi < max_words;
        const uint32_t nxt = (uint32_t)gsrc[wi + 64u < max_words ? wi + 64u : 0u];
        const uint32_t word = have ? cur : 0u;
        const bool fill = (word & 0x8000u) != 0u;
        const uint32_t ng = have ? (fill ? (word & WAH_MAXC) : 1u) : 0u;
        const uint32_t inc = wave_scan_incl_dpp(ng);
        const uint32_t s = gbase + inc - ng;  // first group covered by this word
        const bool active = have && s < G;
        if (active) {
            if (!fill) {
                const uint32_t o = s * WAH_BITS;
                const uint32_t v = word & 0x7FFFu;
                cnt1 += (uint32_t)__popc(v);
                if (row && v && o < row_bits) {
                    atomicOr(&row[o >> 5], v << (o & 31u));
                    if ((o & 31u) > 17u && (o >> 5) + 1u < (row_bits >> 5)) atomicOr(&row[(o >> 5) + 1u], v >> (32u - (o & 31u)));
                }
            } else if (word & 0x4000u) {
                cnt1 += ng * WAH_BITS;
            }
        }
        // ones-fills: the whole wave paints each run
        uint64_t F = row ? __ballot(active && fill && (word & 0x4000u) && ng) : 0ull;
        while (F) {
            const int f = __ffsll((long long)F) - 1;
            F &= F - 1ull;
            const uint32_t fs = (uint32_t)__builtin_amdgcn_readlane((int)s, f);
            const uint32_t fn = (uint32_t)__builtin_amdgcn_readlane((int)ng, f);
            uint32_t b0 = fs * WAH_BITS, b1 = b0 + fn * WAH_BITS;
            if (b1 > row_bits) b1 = row_bits;
            if (b0 >= b1) continue;
            const uint32_t w0 = b0 >> 5, w1 = (b1 - 1u) >> 5;
            for (uint32_t w = w0 + lane; w <= w1; w += 64u) {
                uint32_t m = 0xFFFFFFFFu;
                if (w == w0) m &= 0xFFFFFFFFu << (b0 & 31u);
                if (w == w1 && (b1 & 31u)) m &= (1u << (b1 & 31u)) - 1u;
                atomicOr(&row[w], m);
            }
        }
        const uint64_t A = __ballot(active);
        const uint32_t used = (uint32_t)__popcll(A);
        // groups covered by the consumed words
        const uint32_t covered = (uint32_t)__builtin_amdgcn_readlane((int)inc, (int)(used ? used - 1u : 0u));
        gbase += used ? covered : 0u;
        wbase += used;
        if (used < 64u) break;
        cur = nxt;
    }
    *ones = wave_sum(cnt1);
    return wbase;
}

// ---- expansion by TOGGLES (round 4) -------------------------------------------------------------------------------
// A WAH16 line is runs; painting every ones-fill into the row (wave_wah_expand_row: a serial loop over the fills of
// each 64 words, two v_readlane and a store loop per fill) is what bounded the expansion.  Here a word only marks
// where the row CHANGES: bit p of the toggle row = line bit p XOR line bit p - 1 (bit -1 = 0).  The map is linear
// over GF(2), so every word deposits its own contribution with LDS atomic XORs and contributions of neighbouring
// words cancel where they should: a ones-fill of groups [s, s + n) flips bits 15 s and 15 (s + n); a literal v of
// group s flips the 16 bits (v ^ v << 1) << 15 s; a zero fill flips nothing.  The line is then the running XOR of
// the toggle row: inside a 32-bit word five shift-xor steps (prefix_xor32), across words a carry = the word's top
// bit, across lanes the parity of a ballot.  No loop over fills, no cross-lane reads, cost independent of the runs.
// Used by k_wah_expand_wide_t (rows above 16 KiB: 37.9 -> 29.7 ms for the WAH lines of a configs[3] shard).  The
// one-wave kernel for short rows keeps painting: its toggle twin ran 7.05 against 8.30 ms by itself at configs[2] but
// beside the decode chain - where that expansion runs - it cost the chain 0.7 ms MORE (27.6 against 26.9): it trades
// the painter's scalar instructions, which are free next to a chain that is bound by vector issue, for vector ones.

__device__ __forceinline__ uint32_t prefix_xor32(uint32_t x) {  // bit i of the result = x[0] ^ ... ^ x[i]
    x ^= x << 1;
    x ^= x << 2;
    x ^= x << 4;
    x ^= x << 8;
    x ^= x << 16;
    return x;
}

// One word's toggles into the zeroed row `trow` of `rw` words.  `s` = first group the word covers, `ng` its groups
// (1 for a literal); inactive lanes pass active = false.  Returns the word's ones counted like the reference
// (wah2_extract_count_ones, wah.hpp:232-235: fills count whole groups).
__device__ __forceinline__ uint32_t wah_word_toggles(uint32_t word, uint32_t s, uint32_t ng, bool active, uint32_t* trow /*LDS*/,
                                                     uint32_t rw) {
    const bool fill = (word & 0x8000u) != 0u;
    const bool ones_fill = fill && (word & 0x4000u) != 0u;
    const uint32_t v = word & 0x7FFFu;
    const uint32_t b0 = s * WAH_BITS;
    // literal: 16 toggle bits from bit b0 on (they reach into the next 32-bit word when b0 % 32 > 16);
    // ones-fill: bit b0 and bit b0 + 15 ng; zero fill, inactive lane: nothing
    const uint64_t lit = (uint64_t)(v ^ (v << 1)) << (b0 & 31u);
    const uint32_t b1 = b0 + ng * WAH_BITS;
    uint32_t va = fill ? (ones_fill ? 1u << (b0 & 31u) : 0u) : (uint32_t)lit;
    uint32_t vb = fill ? (ones_fill ? 1u << (b1 & 31u) : 0u) : (uint32_t)(lit >> 32);
    const uint32_t wa = b0 >> 5, wb = fill ? b1 >> 5 : wa + 1u;
    if (!active) va = vb = 0u;
    if (va && wa < rw) atomicXor(&trow[wa], va);
    if (vb && wb < rw) atomicXor(&trow[wb], vb);
    return active ? (fill ? (ones_fill ? ng * WAH_BITS : 0u) : (uint32_t)__popc(v)) : 0u;
}

// Same, fetching the first words itself.
__device__ __forceinline__ uint32_t wave_wah_expand_row(const uint16_t* __restrict__ src, uint32_t max_words,
                                                        uint32_t nbits, uint32_t* row /*LDS*/, uint32_t* ones) {
    const uint32_t lane = lane_id();
    return wave_wah_expand_row(src, max_words, nbits, row, ones, lane < max_words ? (uint32_t)src[lane] : 0u);
}

}  // namespace xsi
