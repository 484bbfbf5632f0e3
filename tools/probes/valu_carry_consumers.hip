// probe 5: a compare and the add-with-carry that consumes it: adjacent, apart, and through SGPR pairs
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
                         : "v"(b), "v"(c), "s"(sv) : "vcc", "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "s28", "s29"); \
        }                                                                                                    \
        out[blockIdx.x * 1024 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;                        \
    }
// five (compare, addc) pairs, adjacent, all through vcc  (10 instructions)
KD(k_adjacent, "v_cmp_eq_u32 vcc, 1, %8\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc\n\tv_cmp_eq_u32 vcc, 2, %8\n\tv_addc_co_u32 %1, vcc, %1, %1, vcc\n\tv_cmp_eq_u32 vcc, 0, %8\n\tv_addc_co_u32 %2, vcc, %2, %2, vcc\n\tv_cmp_eq_u32 vcc, 3, %9\n\tv_addc_co_u32 %3, vcc, %3, %3, vcc\n\tv_cmp_eq_u32 vcc, 4, %9\n\tv_addc_co_u32 %4, vcc, %4, %4, vcc", 16)
// five compares into five SGPR pairs, then five addc reading them (carry out to vcc, unused)
KD(k_grouped, "v_cmp_eq_u32 s[20:21], 1, %8\n\tv_cmp_eq_u32 s[22:23], 2, %8\n\tv_cmp_eq_u32 s[24:25], 0, %8\n\tv_cmp_eq_u32 s[26:27], 3, %9\n\tv_cmp_eq_u32 s[28:29], 4, %9\n\tv_addc_co_u32 %0, vcc, %0, %0, s[20:21]\n\tv_addc_co_u32 %1, vcc, %1, %1, s[22:23]\n\tv_addc_co_u32 %2, vcc, %2, %2, s[24:25]\n\tv_addc_co_u32 %3, vcc, %3, %3, s[26:27]\n\tv_addc_co_u32 %4, vcc, %4, %4, s[28:29]", 16)
// compare, cndmask 0/1, lshl_or  (15 instructions)
KD(k_select, "v_cmp_eq_u32 vcc, 1, %8\n\tv_cndmask_b32 %5, 0, 1, vcc\n\tv_lshl_or_b32 %0, %0, 1, %5\n\tv_cmp_eq_u32 vcc, 2, %8\n\tv_cndmask_b32 %6, 0, 1, vcc\n\tv_lshl_or_b32 %1, %1, 1, %6\n\tv_cmp_eq_u32 vcc, 0, %8\n\tv_cndmask_b32 %5, 0, 1, vcc\n\tv_lshl_or_b32 %2, %2, 1, %5\n\tv_cmp_eq_u32 vcc, 3, %9\n\tv_cndmask_b32 %6, 0, 1, vcc\n\tv_lshl_or_b32 %3, %3, 1, %6\n\tv_cmp_eq_u32 vcc, 4, %9\n\tv_cndmask_b32 %5, 0, 1, vcc\n\tv_lshl_or_b32 %4, %4, 1, %5", 16)
// arithmetic form: x = t ^ K; y = x - 1 (top bit set iff x == 0 for x < 2^31); acc = alignbit(acc, y, 31)   (15 instructions, 10 of them full rate)
KD(k_arith, "v_xor_b32 %5, 1, %8\n\tv_add_u32 %5, -1, %5\n\tv_alignbit_b32 %0, %0, %5, 31\n\tv_xor_b32 %6, 2, %8\n\tv_add_u32 %6, -1, %6\n\tv_alignbit_b32 %1, %1, %6, 31\n\tv_add_u32 %5, -1, %8\n\tv_alignbit_b32 %2, %2, %5, 31\n\tv_xor_b32 %6, 3, %9\n\tv_add_u32 %6, -1, %6\n\tv_alignbit_b32 %3, %3, %6, 31\n\tv_xor_b32 %5, 4, %9\n\tv_add_u32 %5, -1, %5\n\tv_alignbit_b32 %4, %4, %5, 31", 16)
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
        if (rep) printf("%-12s %.3f ms -> %.1f clocks (2.4 GHz) per five conditions collected, per SIMD of 4 waves\n", #NAME, ms, ms * 1e6 / (n * 16.0 * 4) * 2.4); \
    }
    RUN(k_adjacent) RUN(k_grouped) RUN(k_select) RUN(k_arith)
    return 0;
}
