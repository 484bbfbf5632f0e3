// probe 2: dependent chains and the lane-major block, 16 waves per CU
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
// dependent chains: one accumulator
KD(k_dep_add, "v_add_u32 %0, %0, %8", 128)
KD(k_dep_and, "v_and_b32 %0, 0x3fff8, %0", 128)
KD(k_dep_bfi, "v_bfi_b32 %0, %8, %9, %0", 128)
KD(k_dep_xad, "v_xad_u32 %0, %0, %8, %9", 128)
KD(k_dep_lshl, "v_lshlrev_b32 %0, %0, %8", 128)
// the lane-major block on r=%0 acc=%1, temps %2..%5, pair = %8,%9 (6 instructions) + address pair (2): 16 blocks
KD(k_lm_block, "v_lshlrev_b32 %2, %0, %8\n\tv_bcnt_u32_b32 %3, %2, %9\n\tv_ashrrev_i32 %4, 31, %2\n\tv_bfi_b32 %5, %4, %10, %0\n\tv_xad_u32 %0, %3, %4, %5\n\tv_alignbit_b32 %1, %1, %2, 31\n\tv_lshrrev_b32 %6, 5, %0\n\tv_lshl_add_u32 %6, %6, 3, %10", 16)
// two blocks interleaved (independent r: %0 and %7 ... shares temps differently)
KD(k_old_block, "v_lshrrev_b32 %6, 2, %0\n\tv_and_b32 %6, 0x3fff8, %6\n\tv_add_u32 %6, %10, %6\n\tv_lshlrev_b32 %2, %0, %8\n\tv_bcnt_u32_b32 %3, %2, %9\n\tv_cmp_gt_i32 vcc, 0, %2\n\tv_sub_u32 %4, %10, %3\n\tv_add_u32 %0, %3, %0\n\tv_cndmask_b32 %0, %0, %4, vcc\n\tv_writelane_b32 %1, vcc_lo, 3\n\tv_writelane_b32 %5, vcc_hi, 3", 16)
KD(k_cnd, "v_cmp_gt_i32 vcc, 0, %8\n\tv_cndmask_b32 %0, %0, %8, vcc\n\tv_cndmask_b32 %1, %1, %8, vcc\n\tv_cndmask_b32 %2, %2, %8, vcc\n\tv_cndmask_b32 %3, %3, %8, vcc", 32)
int main() {
    uint32_t* out;
    hipMalloc(&out, 256 * 1024 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const uint32_t n = 2000;
#define RUN(NAME, INSTS)                                                                                     \
    for (int rep = 0; rep < 2; ++rep) {                                                                     \
        hipEventRecord(e0);                                                                                 \
        NAME<<<256, 1024>>>(out, n, 77);                                                                    \
        hipEventRecord(e1);                                                                                 \
        hipEventSynchronize(e1);                                                                            \
        float ms;                                                                                           \
        hipEventElapsedTime(&ms, e0, e1);                                                                   \
        if (rep) printf("%-12s %.3f ms -> %.2f cycles (2.4 GHz) per wave-instruction, %d instructions per loop\n", #NAME, ms, ms * 1e6 / (n * (double)INSTS * 4) * 2.4, INSTS); \
    }
    RUN(k_dep_add, 128) RUN(k_dep_and, 128) RUN(k_dep_bfi, 128) RUN(k_dep_xad, 128) RUN(k_dep_lshl, 128)
    RUN(k_lm_block, 128) RUN(k_old_block, 176) RUN(k_cnd, 160)
    return 0;
}
