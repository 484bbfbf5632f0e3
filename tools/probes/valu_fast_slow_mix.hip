// probe 4: do "fast" (add/logic/lshr) and "slow" (everything else) VALU forms overlap? 16 waves per CU
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define KD(NAME, BODY, REPT)                                                                                 \
    __global__ void __launch_bounds__(1024) NAME(uint32_t* out, uint32_t n, uint32_t sv) {                    \
        uint32_t a0 = threadIdx.x, a1 = a0 * 3u, a2 = a0 * 5u, a3 = a0 * 7u, a4 = a0 + 9u, a5 = a0 + 11u, a6 = a0 ^ 13u, a7 = a0 + 17u; \
        uint32_t b = threadIdx.x * 2654435761u, c = blockIdx.x + 12345u;                                     \
        for (uint32_t i = 0; i < n; ++i) {                                                                   \
            asm volatile(".rept " #REPT "\n\t" BODY "\n\t.endr"                                              \
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)    \
                         : "v"(b), "v"(c), "s"(sv) : "vcc");                                                 \
        }                                                                                                    \
        out[blockIdx.x * 1024 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;                        \
    }
#define S(x) "v_bfi_b32 " #x ", %8, %9, " #x "\n\t"
#define F(x) "v_add_u32 " #x ", " #x ", %8\n\t"
#define F2(x) "v_xor_b32 " #x ", " #x ", %9\n\t"
KD(k_s4, S(%0) S(%1) S(%2) S(%3) "s_nop 0", 32)                                    // 4 slow
KD(k_f4, F(%4) F(%5) F(%6) F(%7) "s_nop 0", 32)                                    // 4 fast
KD(k_s4f4, S(%0) F(%4) S(%1) F(%5) S(%2) F(%6) S(%3) F(%7) "s_nop 0", 32)          // 4 slow + 4 fast, interleaved
KD(k_s4f8, S(%0) F(%4) F2(%5) S(%1) F(%6) F2(%7) S(%2) F(%4) F2(%5) S(%3) F(%6) F2(%7) "s_nop 0", 32)  // 4 slow + 8 fast
KD(k_s2f8, S(%0) F(%4) F2(%5) F(%6) F2(%7) S(%1) F(%4) F2(%5) F(%6) F2(%7) "s_nop 0", 32)              // 2 slow + 8 fast
KD(k_s4f4b, S(%0) S(%1) S(%2) S(%3) F(%4) F(%5) F(%6) F(%7) "s_nop 0", 32)         // 4 slow then 4 fast (grouped)
int main() {
    uint32_t* out;
    (void)hipMalloc(&out, 256 * 1024 * 4);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    const uint32_t n = 2000;
#define RUN(NAME)                                                                                            \
    for (int rep = 0; rep < 2; ++rep) {                                                                     \
        (void)hipEventRecord(e0);                                                                           \
        NAME<<<256, 1024>>>(out, n, 77);                                                                    \
        (void)hipEventRecord(e1);                                                                           \
        (void)hipEventSynchronize(e1);                                                                      \
        float ms;                                                                                           \
        (void)hipEventElapsedTime(&ms, e0, e1);                                                             \
        if (rep) printf("%-10s %.3f ms -> %.1f cycles (2.4 GHz) per body per SIMD\n", #NAME, ms, ms * 1e6 / (n * 32.0 * 4) * 2.4); \
    }
    RUN(k_s4) RUN(k_f4) RUN(k_s4f4) RUN(k_s4f8) RUN(k_s2f8) RUN(k_s4f4b)
    return 0;
}
