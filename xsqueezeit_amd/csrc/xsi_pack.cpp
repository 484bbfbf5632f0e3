// xsi_pack.cpp — host-only: an htslib int32 genotype row to a bit row, in the caller's thread (the writer's
// append).  Plain C++ (x86-64 with run-time dispatch to AVX-512 / AVX2), kept out of the .hip sources because those
// are compiled a second time for the device, where the CPU feature builtins do not exist.
#include <cstdint>
#include <cstdlib>

#include "xsi_common.hpp"

namespace xsi {
bool pack_bit_row(const int32_t* gt, uint32_t n, int dp, uint8_t* out);  // declared for the library in xsi_ctx.hpp

// One htslib row -> one bit per haplotype (bit h, LSB first = haplotype h carries ALT), if every value is an allele
// 0 or 1 ((allele + 1) << 1 | phase, so 2..5) and every second value carries the default phase; false otherwise (the
// row then goes the int32 way; `out` may hold garbage).  The first value's phase bit is not stored by the format
// (NonDefaultPhasingPred looks at odd positions only, gt_block.hpp:93-100) and is ignored here as there.
// out[0 .. ceil(n / 8)) is written; the rest of the bit row stays zero.
#include <immintrin.h>
__attribute__((target("avx512f,avx512bw"))) static bool pack_bit_row_avx512(const int32_t* gt, uint32_t n, int dp, uint8_t* out) {
    const __m512i two = _mm512_set1_epi32(2), four = _mm512_set1_epi32(4), one = _mm512_set1_epi32(1);
    const __mmask16 odd = 0xAAAA, want = dp ? 0xAAAA : 0;
    __mmask16 bad = 0;
    uint32_t i = 0;
    uint16_t* o16 = reinterpret_cast<uint16_t*>(out);
    for (; i + 16 <= n; i += 16) {
        // the row comes from the caller's memory, often straight from DRAM: the hardware prefetcher restarts at every
        // 4 KiB page, so the line 4 KiB ahead is asked for here (a prefetch past the row's end is harmless)
        // (distances of 1 to 16 KiB and the non-temporal hint measure the same, profiles/r03_host_boundary.txt)
        _mm_prefetch(reinterpret_cast<const char*>(gt + i) + 4096, _MM_HINT_T0);
        const __m512i v = _mm512_loadu_si512(gt + i);
        bad |= _mm512_cmp_epu32_mask(_mm512_sub_epi32(v, two), four, _MM_CMPINT_NLT);           // not in 2..5
        bad |= (__mmask16)((_mm512_test_epi32_mask(v, one) ^ want) & odd);                         // second value's phase
        o16[i >> 4] = (uint16_t)_mm512_test_epi32_mask(v, four);
    }
    if (i < n) {
        const __mmask16 live = (__mmask16)((1u << (n - i)) - 1u);
        const __m512i v = _mm512_mask_loadu_epi32(two, live, gt + i);
        bad |= _mm512_cmp_epu32_mask(_mm512_sub_epi32(v, two), four, _MM_CMPINT_NLT);
        bad |= (__mmask16)((_mm512_test_epi32_mask(v, one) ^ want) & odd & live);
        const uint16_t m = (uint16_t)(_mm512_test_epi32_mask(v, four) & live);
        out[i >> 3] = (uint8_t)m;
        if (n - i > 8) out[(i >> 3) + 1] = (uint8_t)(m >> 8);
    }
    return bad == 0;
}
__attribute__((target("avx2"))) static bool pack_bit_row_avx2(const int32_t* gt, uint32_t n, int dp, uint8_t* out) {
    const __m256i two = _mm256_set1_epi32(2), three = _mm256_set1_epi32(3);
    const __m256i oddphase = _mm256_setr_epi32(0, 1, 0, 1, 0, 1, 0, 1), want = dp ? oddphase : _mm256_setzero_si256();
    __m256i bad = _mm256_setzero_si256();
    uint32_t i = 0;
    for (; i + 8 <= n; i += 8) {
        if ((i & 8u) == 0u) _mm_prefetch(reinterpret_cast<const char*>(gt + i) + 4096, _MM_HINT_T0);
        const __m256i v = _mm256_loadu_si256(reinterpret_cast<const __m256i*>(gt + i));
        const __m256i t = _mm256_sub_epi32(v, two);                                   // valid: 0..3 (unsigned)
        bad = _mm256_or_si256(bad, _mm256_andnot_si256(three, t));                   // any bit above the low two
        bad = _mm256_or_si256(bad, _mm256_xor_si256(_mm256_and_si256(v, oddphase), want));
        out[i >> 3] = (uint8_t)_mm256_movemask_ps(_mm256_castsi256_ps(_mm256_slli_epi32(v, 29)));  // bit 2 -> sign
    }
    bool ok = _mm256_testz_si256(bad, bad) != 0;
    if (i < n) {
        uint8_t m = 0;
        for (uint32_t k = i; k < n; ++k) {
            const uint32_t v = (uint32_t)gt[k];
            ok = ok && v - 2u < 4u && (!(k & 1u) || (int)(v & 1u) == (dp ? 1 : 0));
            m |= (uint8_t)(((v >> 2) & 1u) << (k - i));
        }
        out[i >> 3] = m;
    }
    return ok;
}
static bool pack_bit_row_scalar(const int32_t* gt, uint32_t n, int dp, uint8_t* out) {
    bool ok = true;
    for (uint32_t i = 0; i < n; i += 8) {
        uint8_t m = 0;
        for (uint32_t k = i; k < n && k < i + 8; ++k) {
            const uint32_t v = (uint32_t)gt[k];
            ok = ok && v - 2u < 4u && (!(k & 1u) || (int)(v & 1u) == (dp ? 1 : 0));
            m |= (uint8_t)(((v >> 2) & 1u) << (k - i));
        }
        out[i >> 3] = m;
    }
    return ok;
}
bool pack_bit_row(const int32_t* gt, uint32_t n, int dp, uint8_t* out) {
    static const int isa = [] {
        if (tuning_env("XSI_WRITER_NO_PACK")) return -1;  // testing: every line the int32 way
        __builtin_cpu_init();
        if (const char* e = tuning_env("XSI_PACK_ISA")) {  // testing: 0 scalar, 1 AVX2 (the tests run every form the CPU has)
            const int want = atoi(e);
            if (want == 0 || (want == 1 && __builtin_cpu_supports("avx2"))) return want;
        }
        if (__builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512bw")) return 2;
        return __builtin_cpu_supports("avx2") ? 1 : 0;
    }();
    if (isa < 0) return false;
    return isa == 2 ? pack_bit_row_avx512(gt, n, dp, out) : isa == 1 ? pack_bit_row_avx2(gt, n, dp, out) : pack_bit_row_scalar(gt, n, dp, out);
}

}  // namespace xsi
