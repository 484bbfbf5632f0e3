// xsi_pair.hip — position-major PBWT decode chain with the prefix array packed two members per LDS
// dword (gfx950).  Blocks without fully haploid lines, N <= 65536 haplotypes, one workgroup per block:
// the kernel for many large blocks (>= ~192 blocks of >= ~40k haplotypes), where every CU has a block
// of its own; with fewer blocks the element-major kernels of xsi_rank.hip, which split a block's
// haplotypes over workgroups, fill the chip better.
//
// What the chain computes (reference, per block, for every WAH line k in order):
//   x_k[a_k[i]] = y_k[i]                                            (accessor_internals_new.hpp:228-230)
//   a_{k+1} = [a_k[i] : y_k[i]=0] ++ [a_k[i] : y_k[i]=1]             (gt_block.hpp:124-136)
// `a` restarts at identity for every block (accessor_internals_new.hpp:144).
//
// Layout.  The prefix array holds NA = 2H members (N real ones + padding members whose key is always 1,
// so they stay behind every real member and no step needs a validity test).  LDS dword q (0 <= q < H)
// holds position q in its low half and position q + H in its high half:
//   * a lane that reads dword (chunk*64 + lane) gets one member of a low chunk and one of a high chunk
//     with one conflict-free 4-byte read, already packed two to a VGPR: 65536 members in 32 registers;
//   * consecutive positions live in consecutive dwords, so a chunk's 16-bit scatter stores
//     (ds_write_b16 / ds_write_b16_d16_hi to runs of consecutive positions) hit different banks.
// Wave w owns dwords [w*EP*64, (w+1)*EP*64): EP low chunks and EP high chunks.  The chunk masks ARE the
// stored row y (loaded one line ahead, lane e < EP carries the masks of the wave's e-th chunks): no
// gather of key bits.  Per line:
//   pass 1  every lane reads its EP dwords; zeros per range from the masks;
//   barrier, 32-counter DPP scan (16 waves x {low, high} range) -> zeros before every range;
//   pass 2  stable scatter back into the same array (everything was read before the barrier).  Chunks
//           whose mask is all zero (about two thirds of them: PBWT clusters the ones) shift as one run:
//           4 vector instructions instead of 11.  x[a[i]] is set for the minority side of the line only
//           (LDS atomic OR) and the row is inverted on output when that side was the zeros;
//   barrier; the finished row goes to HBM at the top of the next line.
// Every barrier is lds_barrier(): 16-bit LDS stores must have completed before another wave reads them.
#include "xsi_kernels.hpp"

#include <cstdlib>

#include "xsi_device.hpp"

namespace xsi {

struct PairArgs {
    const uint32_t* wah_lines;  // [rank] binary line
    const uint32_t* src;        // permuted rows (by rank)
    uint32_t src_stride_w;
    uint32_t src_elem_shift;    // 1 when the rows are {word, prefix} pairs
    uint32_t* dst;              // output rows (by binary line)
    uint32_t dst_stride_w;
    uint32_t N;
};

using LdsU16 = __attribute__((address_space(3))) uint16_t;
using LdsU32 = __attribute__((address_space(3))) uint32_t;

template <int EP>
__global__ void __launch_bounds__(1024) k_chain_pair_dec(const DecBlock* __restrict__ dblocks, PairArgs A) {
    constexpr uint32_t T = 1024, W = 16;
    constexpr uint32_t H = W * EP * 64u;  // positions per half = dwords of the array
    constexpr uint32_t NA = 2u * H;       // capacity
    constexpr uint32_t CW = NA / 32u;     // words of a bit row over NA positions
    static_assert(EP >= 1 && EP <= 32, "16-bit members: NA <= 65536");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // the decoded row sits at LDS address 0 (one 8 KiB slot), so the address of the word that holds bit
    // `member` is (member >> 3) & 0x1FFC with nothing to add
    constexpr uint32_t SLOT = 8192u;
    static_assert(CW * 4u <= SLOT, "a bit row over NA <= 65536 positions fits the slot");
    uint32_t* xrow = reinterpret_cast<uint32_t*>(smem);
    uint32_t* D = reinterpret_cast<uint32_t*>(smem + SLOT);  // H dwords
    uint32_t* wcnt = D + H;                                  // 32 zero counts
    const uint32_t N = A.N;
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const uint32_t w = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6));

    const DecBlock& B = dblocks[blockIdx.x];
    if (B.error || B.off_line_haploid != VAL_UNDEFINED) return;  // k_chain_lds takes the blocks with haploid lines
    const uint32_t wah_first = B.wah_first, n_wah = B.n_wah;
    if (n_wah == 0) return;

    for (uint32_t i = tid; i < H; i += T) D[i] = i | ((i + H) << 16);
    for (uint32_t i = tid; i < CW; i += T) xrow[i] = 0;

    const uint32_t x_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
    const uint32_t a_lds = x_lds + SLOT;
    const uint32_t src_words = (N + 31u) >> 5;
    const uint32_t cgL = w * EP + lane;  // chunk whose mask lane `lane` (< EP) carries: low range
    const uint32_t cgH = H / 64u + cgL;  // ... and high range

    // masks of my chunks for one line; positions at or beyond N (padding members) read as ones
    uint32_t yl_lo = 0, yl_hi = 0, yh_lo = 0, yh_hi = 0;
    auto load_masks = [&](uint32_t rank) {
        const uint32_t* row = A.src + (size_t)rank * A.src_stride_w;
        auto word = [&](uint32_t wi) -> uint32_t {
            uint32_t v = wi < src_words ? row[(size_t)wi << A.src_elem_shift] : 0u;
            const uint32_t b0 = wi * 32u;
            if (b0 + 32u > N) v |= (b0 >= N) ? 0xFFFFFFFFu : (0xFFFFFFFFu << (N - b0));
            return v;
        };
        yl_lo = yl_hi = yh_lo = yh_hi = 0xFFFFFFFFu;
        if (lane < (uint32_t)EP) {
            yl_lo = word(2u * cgL);
            yl_hi = word(2u * cgL + 1u);
            yh_lo = word(2u * cgH);
            yh_hi = word(2u * cgH + 1u);
        }
    };
    // the decoded row of a line from LDS to its output row (bits >= N cleared), LDS row zeroed again
    auto flush_xrow = [&](uint32_t line, uint32_t inv) {
        uint32_t* orow = A.dst + (size_t)line * A.dst_stride_w;
        for (uint32_t i = tid; i < A.dst_stride_w; i += T) {
            uint32_t v = 0;
            if (i < CW) {
                v = xrow[i];
                xrow[i] = 0;
                if (inv) v = ~v;
                const uint32_t b0 = i * 32u;
                if (b0 + 32u > N) v &= (b0 >= N) ? 0u : ((1u << (N - b0)) - 1u);
            }
            orow[i] = v;
        }
    };

    uint32_t id_cur = A.wah_lines[wah_first];
    uint32_t id_n1 = n_wah > 1u ? A.wah_lines[wah_first + 1u] : 0u;
    load_masks(wah_first);
    const LdsU32* Dw = reinterpret_cast<const LdsU32*>((uintptr_t)(a_lds + (w * EP * 64u + lane) * 4u));
    uint32_t prev_line = 0, prev_inv = 0;

    for (uint32_t j = 0; j < n_wah; ++j) {
        lds_barrier();  // A: the previous scatter and the previous row are complete
        if (j) flush_xrow(prev_line, prev_inv);
        const uint32_t id_n2 = j + 2u < n_wah ? A.wah_lines[wah_first + j + 2u] : 0u;

        // ---- pass 1
        uint32_t v[EP];
        static_for<0, EP>([&](auto ec) {
            constexpr int e = decltype(ec)::value;
            v[e] = Dw[e * 64];
        });
        uint32_t ml_lo = yl_lo, ml_hi = yl_hi, mh_lo = yh_lo, mh_hi = yh_hi;  // lane e: my e-th low / high chunk
        uint32_t c = 0;
        if (lane < (uint32_t)EP)
            c = ((uint32_t)__popc(ml_lo) + (uint32_t)__popc(ml_hi)) |
                (((uint32_t)__popc(mh_lo) + (uint32_t)__popc(mh_hi)) << 16);
        c = wave_scan_incl_dpp(c);
        c = (uint32_t)__builtin_amdgcn_readlane((int)c, 63);
        if (j + 1u < n_wah) load_masks(wah_first + j + 1u);
        if (lane == 0) {
            wcnt[w] = EP * 64u - (c & 0xFFFFu);  // zeros of my low range
            wcnt[W + w] = EP * 64u - (c >> 16);  // ... and of my high range
        }
        lds_barrier();  // B: every wave has read its members; zero counts published

        // zeros before each of the 32 ranges, in position order (16 low ranges, then 16 high ranges)
        uint32_t sc = row16_scan_incl(lane < 2u * W ? wcnt[lane] : 0u);
        sc += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)sc, 0x142, 0xA, 0xF, false);  // row_bcast:15
        const uint32_t tz = (uint32_t)__builtin_amdgcn_readlane((int)sc, 2 * W - 1);
        const uint32_t zbL = w ? (uint32_t)__builtin_amdgcn_readlane((int)sc, (int)w - 1) : 0u;
        const uint32_t zbH = (uint32_t)__builtin_amdgcn_readlane((int)sc, (int)(W + w) - 1);
        // fewer zeros than ones: scatter the zeros into x and invert the row on output.  readfirstlane:
        // tz is wave-uniform, and saying so keeps every use of `inv` on the scalar unit
        const uint32_t inv = (uint32_t)__builtin_amdgcn_readfirstlane((int)(2u * tz < N ? 1u : 0u));

        // ---- pass 2: stable scatter (zeros keep their order in front, ones behind).  Branch-free per
        // chunk and light on the scalar unit: position p of a member, then its LDS byte address 4p (low
        // half of dword p) or 4(p-H)+2.  The masks come back out of the per-lane registers; the asm keeps
        // the compiler from holding all 2*EP of them in SGPRs instead (it would spill them lane by lane).
        asm volatile("" : "+v"(ml_lo), "+v"(ml_hi), "+v"(mh_lo), "+v"(mh_hi));
        uint32_t k_lo = a_lds, k_hi = a_lds - (4u * H - 2u);  // address bias of a position below / at or above H
        asm volatile("" : "+v"(k_lo), "+v"(k_hi));
        auto scatter = [&](uint64_t om, uint32_t val16, uint32_t& zb, uint32_t& ob) {
            const uint32_t no = (uint32_t)__popcll(om);
            const uint32_t opre = mbcnt64(om);
            const uint32_t p0 = zb + (lane - opre);
            const uint32_t p1 = ob + opre;
            const uint32_t p = __builtin_amdgcn_inverse_ballot_w64(om) ? p1 : p0;
            const uint32_t addr = (p << 2) + (p >= H ? k_hi : k_lo);  // v_cmp, v_cndmask, v_lshl_add
            *reinterpret_cast<LdsU16*>((uintptr_t)addr) = (uint16_t)val16;
            zb += 64u - no;
            ob += no;
        };
        auto shift_run = [&](uint32_t val16, uint32_t& zb) {  // a chunk of 64 zeros
            const uint32_t p = zb + lane;
            const uint32_t addr = (p << 2) + (p >= H ? k_hi : k_lo);
            *reinterpret_cast<LdsU16*>((uintptr_t)addr) = (uint16_t)val16;
            zb += 64u;
        };
        auto xset = [&](uint64_t om, uint32_t member) {  // address and shift take bits 15..0 of `member`
            const uint64_t sm = inv ? ~om : om;
            if (sm) {
                if (__builtin_amdgcn_inverse_ballot_w64(sm)) {
                    LdsU32* p = reinterpret_cast<LdsU32*>((uintptr_t)(((member >> 3) & 0x1FFCu) | x_lds));
                    __hip_atomic_fetch_or(p, 1u << (member & 31u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            }
        };
        // which of my chunks have any one: a ballot over the lanes that carry the masks
        uint32_t nzL = 0xFFFFFFFFu, nzH = 0xFFFFFFFFu;
        if (!inv) {
            nzL = (uint32_t)__ballot((ml_lo | ml_hi) != 0u);
            nzH = (uint32_t)__ballot((mh_lo | mh_hi) != 0u);
        }
        {
            uint32_t zb = zbL, ob = tz + (w * EP * 64u - zbL);
            static_for<0, EP>([&](auto ec) {
                constexpr int e = decltype(ec)::value;
                if (!((nzL >> e) & 1u)) {
                    shift_run(v[e], zb);
                } else {
                    const uint64_t om = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)ml_hi, e) << 32) |
                                        (uint32_t)__builtin_amdgcn_readlane((int)ml_lo, e);
                    xset(om, v[e]);
                    scatter(om, v[e], zb, ob);
                }
            });
        }
        {
            uint32_t zb = zbH, ob = tz + (H + w * EP * 64u - zbH);
            static_for<0, EP>([&](auto ec) {
                constexpr int e = decltype(ec)::value;
                if (!((nzH >> e) & 1u)) {
                    shift_run(v[e] >> 16, zb);
                } else {
                    const uint64_t om = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)mh_hi, e) << 32) |
                                        (uint32_t)__builtin_amdgcn_readlane((int)mh_lo, e);
                    xset(om, v[e] >> 16);
                    scatter(om, v[e] >> 16, zb, ob);
                }
            });
        }
        prev_line = id_cur;
        prev_inv = inv;
        id_cur = id_n1;
        id_n1 = id_n2;
    }
    lds_barrier();
    flush_xrow(prev_line, prev_inv);
}

static const int k_pair_EP[] = {8, 12, 16, 20, 24, 28, 32};

static int pair_ep_for(uint32_t N) {
    for (int ep : k_pair_EP)
        if ((uint32_t)ep * 2048u >= N) return ep;
    return 0;
}

bool chain_pair_supported(uint32_t N) { return N >= 2u && N <= 65536u && pair_ep_for(N) != 0; }

hipError_t launch_pair_decode(hipStream_t s, const DecBlock* blocks, uint32_t n_blocks, const DecLines& L,
                              uint32_t* out_rows, uint32_t out_stride_w) {
    if (!n_blocks) return hipSuccess;
    PairArgs A{};
    A.wah_lines = L.wah_lines;
    A.src = reinterpret_cast<const uint32_t*>(L.yp);
    A.src_stride_w = L.yp_stride * 2u;
    A.src_elem_shift = 1;
    A.dst = out_rows;
    A.dst_stride_w = out_stride_w;
    A.N = L.N;
    const int ep = pair_ep_for(A.N);
    if (!ep) return hipErrorInvalidValue;
    const uint32_t H = 16u * (uint32_t)ep * 64u;
    const uint32_t lds = 8192u + 4u * H + 32u * 4u;
#define XSI_PAIR_CASE(EE)                                                                                   \
    if (ep == EE) {                                                                                         \
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_chain_pair_dec<EE>),            \
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);           \
        if (e != hipSuccess) return e;                                                                      \
        k_chain_pair_dec<EE><<<dim3(n_blocks), dim3(1024), lds, s>>>(blocks, A);                            \
        return hipGetLastError();                                                                           \
    }
    XSI_PAIR_CASE(8)
    XSI_PAIR_CASE(12)
    XSI_PAIR_CASE(16)
    XSI_PAIR_CASE(20)
    XSI_PAIR_CASE(24)
    XSI_PAIR_CASE(28)
    XSI_PAIR_CASE(32)
#undef XSI_PAIR_CASE
    return hipErrorInvalidValue;
}

}  // namespace xsi
