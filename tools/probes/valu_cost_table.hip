// probe 3: issue cost table of VALU forms, 16 waves per CU, 8 independent accumulators
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void __launch_bounds__(1024) k_add_vv(uint32_t* out, uint32_t n, uint32_t sv) {
    uint32_t a0 = threadIdx.x, a1 = a0 * 3u, a2 = a0 * 5u, a3 = a0 * 7u, a4 = a0 + 9u, a5 = a0 + 11u, a6 = a0 ^ 13u, a7 = a0 + 17u;
    uint32_t b = threadIdx.x * 2654435761u, c = blockIdx.x + 12345u;
    for (uint32_t i = 0; i < n; ++i)
        asm volatile("s_mov_b64 s[22:23], 0x55\n\ts_mov_b32 s24, 5\n\ts_nop 4\n\t.rept 16\n\tv_add_u32 %0, %0, %8\n\tv_add_u32 %1, %1, %8\n\tv_add_u32 %2, %2, %8\n\tv_add_u32 %3, %3, %8\n\tv_add_u32 %4, %4, %8\n\tv_add_u32 %5, %5, %8\n\tv_add_u32 %6, %6, %8\n\tv_add_u32 %7, %7, %8\n\t.endr"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                     : "v"(b), "v"(c), "s"(sv) : "vcc", "s20", "s21", "s22", "s23", "s24", "v40", "v41");
    out[blockIdx.x * 1024 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}
__global__ void __launch_bounds__(1024) k_add_sv(uint32_t* out, uint32_t n, uint32_t sv) {
    uint32_t a0 = threadIdx.x, a1 = a0 * 3u, a2 = a0 * 5u, a3 = a0 * 7u, a4 = a0 + 9u, a5 = a0 + 11u, a6 = a0 ^ 13u, a7 = a0 + 17u;
    uint32_t b = threadIdx.x * 2654435761u, c = blockIdx.x + 12345u;
    for (uint32_t i = 0; i < n; ++i)
        asm volatile("s_mov_b64 s[22:23], 0x55\n\ts_mov_b32 s24, 5\n\ts_nop 4\n\t.rept 16\n\tv_add_u32 %0, %10, %0\n\tv_add_u32 %1, %10, %1\n\tv_add_u32 %2, %10, %2\n\tv_add_u32 %3, %10, %3\n\tv_add_u32 %4, %10, %4\n\tv_add_u32 %5, %10, %5\n\tv_add_u32 %6, %10, %6\n\tv_add_u32 %7, %10, %7\n\t.endr"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                     : "v"(b), "v"(c), "s"(sv) : "vcc", "s20", "s21", "s22", "s23", "s24", "v40", "v41");
    out[blockIdx.x * 1024 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}
__global__ void __launch_bounds__(1024) k_add_iv(uint32_t* out, uint32_t n, uint32_t sv) {
    uint32_t a0 = threadIdx.x, a1 = a0 * 3u, a2 = a0 * 5u, a3 = a0 * 7u, a4 = a0 + 9u, a5 = a0 + 11u, a6 = a0 ^ 13u, a7 = a0 + 17u;
    uint32_t b = threadIdx.x * 2654435761u, c = blockIdx.x + 12345u;
    for (uint32_t i = 0; i < n; ++i)
        asm volatile("s_mov_b64 s[22:23], 0x55\n\ts_mov_b32 s24, 5\n\ts_nop 4\n\t.rept 16\n\tv_add_u32 %0, 17, %0\n\tv_add_u32 %1, 17, %1\n\tv_add_u32 %2, 17, %2\n\tv_add_u32 %3, 17, %3\n\tv_add_u32 %4, 17, %4\n\tv_add_u32 %5, 17, %5\n\tv_add_u32 %6, 17, %6\n\tv_add_u32 %7, 17, %7\n\t.endr"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                     : "v"(b), "v"(c), "s"(sv) : "vcc", "s20", "s21", "s22", "s23", "s24", "v40", "v41");
    out[blockIdx.x * 1024 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}
__global__ void __launch_bounds__(1024) k_add_lv(uint32_t* out, uint32_t n, uint32_t sv) {
    uint32_t a0 = threadIdx.x, a1 = a0 * 3u, a2 = a0 * 5u, a3 = a0 * 7u, a4 = a0 + 9u, a5 = a0 + 11u, a6 = a0 ^ 13u, a7 = a0 + 17u;
    uint32_t b = threadIdx.x * 2654435761u, c = blockIdx.x + 12345u;
    for (uint32_t i = 0; i < n; ++i)
        asm volatile("s_mov_b64 s[22:23], 0x55\n\ts_mov_b32 s24, 5\n\ts_nop 4\n\t.rept 16\n\tv_add_u32 %0, 0x12345, %0\n\tv_add_u32 %1, 0x12345, %1\n\tv_add_u32 %2, 0x12345, %2\n\tv_add_u32 %3, 0x12345, %3\n\tv_add_u32 %4, 0x12345, %4\n\tv_add_u32 %5, 0x12345, %5\n\tv_add_u32 %6, 0x12345, %6\n\tv_add_u32 %7, 0x12345, %7\n\t.endr"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                     : "v"(b), "v"(c), "s"(sv) : "vcc", "s20", "s21", "s22", "s23", "s24", "v40", "v41");
    out[blockIdx.x * 1024 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}
__global__ void __launch_bounds__(1024) k_sub_vv(uint32_t* out, uint32_t n, uint32_t sv) {
    uint32_t a0 = threadIdx.x, a1 = a0 * 3u, a2 = a0 * 5u, a3 = a0 * 7u, a4 = a0 + 9u, a5 = a0 + 11u, a6 = a0 ^ 13u, a7 = a0 + 17u;
    uint32_t b = threadIdx.x * 2654435761u, c = blockIdx.x + 12345u;
    for (uint32_t i = 0; i < n; ++i)
        asm volatile("s_mov_b64 s[22:23], 0x55\n\ts_mov_b32 s24, 5\n\ts_nop 4\n\t.rept 16\n\tv_sub_u32 %0, %0, %8\n\tv_sub_u32 %1, %1, %8\n\tv_sub_u32 %2, %2, %8\n\tv_sub_u32 %3, %3, %8\n\tv_sub_u32 %4, %4, %8\n\tv_sub_u32 %5, %5, %8\n\tv_sub_u32 %6, %6, %8\n\tv_sub_u32 %7, %7, %8\n\t.endr"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                     : "v"(b), "v"(c), "s"(sv) : "vcc", "s20", "s21", "s22", "s23", "s24", "v40", "v41");
    out[blockIdx.x * 1024 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}
__global__ void __launch_bounds__(1024) k_sub_sv(uint32_t* out, uint32_t n, uint32_t sv) {
    uint32_t a0 = threadIdx.x, a1 = a0 * 3u, a2 = a0 * 5u, a3 = a0 * 7u, a4 = a0 + 9u, a5 = a0 + 11u, a6 = a0 ^ 13u, a7 = a0 + 17u;
    uint32_t b = threadIdx.x * 2654435761u, c = blockIdx.x + 12345u;
    for (uint32_t i = 0; i < n; ++i)
        asm volatile("s_mov_b64 s[22:23], 0x55\n\ts_mov_b32 s24, 5\n\ts_nop 4\n\t.rept 16\n\tv_sub_u32 %0, %10, %0\n\tv_sub_u32 %1, %10, %1\n\tv_sub_u32 %2, %10, %2\n\tv_sub_u32 %3, %10, %3\n\tv_sub_u32 %4, %10, %4\n\tv_sub_u32 %5, %10, %5\n\tv_sub_u32 %6, %10, %6\n\tv_sub_u32 %7, %10, %7\n\t.endr"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                     : "v"(b), "v"(c), "s"(sv) : "vcc", "s20", "s21", "s22", "s23", "s24", "v40", "v41");
    out[blockIdx.x * 1024 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}
__global__ void __launch_bounds__(1024) k_subrev_sv(uint32_t* out, uint32_t n, uint32_t sv) {
    uint32_t a0 = threadIdx.x, a1 = a0 * 3u, a2 = a0 * 5u, a3 = a0 * 7u, a4 = a0 + 9u, a5 = a0 + 11u, a6 = a0 ^ 13u, a7 = a0 + 17u;
    uint32_t b = threadIdx.x * 2654435761u, c = blockIdx.x + 12345u;
    for (uint32_t i = 0; i < n; ++i)
        asm volatile("s_mov_b64 s[22:23], 0x55\n\ts_mov_b32 s24, 5\n\ts_nop 4\n\t.rept 16\n\tv_subrev_u32 %0, %10, %0\n\tv_subrev_u32 %1, %10, %1\n\tv_subrev_u32 %2, %10, %2\n\tv_subrev_u32 %3, %10, %3\n\tv_subrev_u32 %4, %10, %4\n\tv_subrev_u32 %5, %10, %5\n\tv_subrev_u32 %6, %10, %6\n\tv_subrev_u32 %7, %10, %7\n\t.endr"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                     : "v"(b), "v"(c), "s"(sv) : "vcc", "s20", "s21", "s22", "s23", "s24", "v40", "v41");
    out[blockIdx.x * 1024 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}
__global__ void __launch_bounds__(1024) k_and_vv(uint32_t* out, uint32_t n, uint32_t sv) {
    uint32_t a0 = threadIdx.x, a1 = a0 * 3u, a2 = a0 * 5u, a3 = a0 * 7u, a4 = a0 + 9u, a5 = a0 + 11u, a6 = a0 ^ 13u, a7 = a0 + 17u;
    uint32_t b = threadIdx.x * 2654435761u, c = blockIdx.x + 12345u;
    for (uint32_t i = 0; i < n; ++i)
        asm volatile("s_mov_b64 s[22:23], 0x55\n\ts_mov_b32 s24, 5\n\ts_nop 4\n\t.rept 16\n\tv_and_b32 %0, %0, %8\n\tv_and_b32 %1, %1, %8\n\tv_and_b32 %2, %2, %8\n\tv_and_b32 %3, %3, %8\n\tv_and_b32 %4, %4, %8\n\tv_and_b32 %5, %5, %8\n\tv_and_b32 %6, %6, %8\n\tv_and_b32 %7, %7, %8\n\t.endr"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                     : "v"(b), "v"(c), "s"(sv) : "vcc", "s20", "s21", "s22", "s23", "s24", "v40", "v41");
    out[blockIdx.x * 1024 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}
__global__ void __launch_bounds__(1024) k_and_sv(uint32_t* out, uint32_t n, uint32_t sv) {
    uint32_t a0 = threadIdx.x, a1 = a0 * 3u, a2 = a0 * 5u, a3 = a0 * 7u, a4 = a0 + 9u, a5 = a0 + 11u, a6 = a0 ^ 13u, a7 = a0 + 17u;
    uint32_t b = threadIdx.x * 2654435761u, c = blockIdx.x + 12345u;
    for (uint32_t i = 0; i < n; ++i)
        asm volatile("s_mov_b64 s[22:23], 0x55\n\ts_mov_b32 s24, 5\n\ts_nop 4\n\t.rept 16\n\tv_and_b32 %0, %10, %0\n\tv_and_b32 %1, %10, %1\n\tv_and_b32 %2, %10, %2\n\tv_and_b32 %3, %10, %3\n\tv_and_b32 %4, %10, %4\n\tv_and_b32 %5, %10, %5\n\tv_and_b32 %6, %10, %6\n\tv_and_b32 %7, %10, %7\n\t.endr"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                     : "v"(b), "v"(c), "s"(sv) : "vcc", "s20", "s21", "s22", "s23", "s24", "v40", "v41");
    out[blockIdx.x * 1024 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}
__global__ void __launch_bounds__(1024) k_and_lv(uint32_t* out, uint32_t n, uint32_t sv) {
    uint32_t a0 = threadIdx.x, a1 = a0 * 3u, a2 = a0 * 5u, a3 = a0 * 7u, a4 = a0 + 9u, a5 = a0 + 11u, a6 = a0 ^ 13u, a7 = a0 + 17u;
    uint32_t b = threadIdx.x * 2654435761u, c = blockIdx.x + 12345u;
    for (uint32_t i = 0; i < n; ++i)
        asm volatile("s_mov_b64 s[22:23], 0x55\n\ts_mov_b32 s24, 5\n\ts_nop 4\n\t.rept 16\n\tv_and_b32 %0, 0x3fff8, %0\n\tv_and_b32 %1, 0x3fff8, %1\n\tv_and_b32 %2, 0x3fff8, %2\n\tv_and_b32 %3, 0x3fff8, %3\n\tv_and_b32 %4, 0x3fff8, %4\n\tv_and_b32 %5, 0x3fff8, %5\n\tv_and_b32 %6, 0x3fff8, %6\n\tv_and_b32 %7, 0x3fff8, %7\n\t.endr"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                     : "v"(b), "v"(c), "s"(sv) : "vcc", "s20", "s21", "s22", "s23", "s24", "v40", "v41");
    out[blockIdx.x * 1024 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}
__global__ void __launch_bounds__(1024) k_or_vv(uint32_t* out, uint32_t n, uint32_t sv) {
    uint32_t a0 = threadIdx.x, a1 = a0 * 3u, a2 = a0 * 5u, a3 = a0 * 7u, a4 = a0 + 9u, a5 = a0 + 11u, a6 = a0 ^ 13u, a7 = a0 + 17u;
    uint32_t b = threadIdx.x * 2654435761u, c = blockIdx.x + 12345u;
    for (uint32_t i = 0; i < n; ++i)
        asm volatile("s_mov_b64 s[22:23], 0x55\n\ts_mov_b32 s24, 5\n\ts_nop 4\n\t.rept 16\n\tv_or_b32 %0, %0, %8\n\tv_or_b32 %1, %1, %8\n\tv_or_b32 %2, %2, %8\n\tv_or_b32 %3, %3, %8\n\tv_or_b32 %4, %4, %8\n\tv_or_b32 %5, %5, %8\n\tv_or_b32 %6, %6, %8\n\tv_or_b32 %7, %7, %8\n\t.endr"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                     : "v"(b), "v"(c), "s"(sv) : "vcc", "s20", "s21", "s22", "s23", "s24", "v40", "v41");
    out[blockIdx.x * 1024 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}
__global__ void __launch_bounds__(1024) k_xor_vv(uint32_t* out, uint32_t n, uint32_t sv) {
    uint32_t a0 = threadIdx.x, a1 = a0 * 3u, a2 = a0 * 5u, a3 = a0 * 7u, a4 = a0 + 9u, a5 = a0 + 11u, a6 = a0 ^ 13u, a7 = a0 + 17u;
    uint32_t b = threadIdx.x * 2654435761u, c = blockIdx.x + 12345u;
    for (uint32_t i = 0; i < n; ++i)
        asm volatile("s_mov_b64 s[22:23], 0x55\n\ts_mov_b32 s24, 5\n\ts_nop 4\n\t.rept 16\n\tv_xor_b32 %0, %0, %8\n\tv_xor_b32 %1, %1, %8\n\tv_xor_b32 %2, %2, %8\n\tv_xor_b32 %3, %3, %8\n\tv_xor_b32 %4, %4, %8\n\tv_xor_b32 %5, %5, %8\n\tv_xor_b32 %6, %6, %8\n\tv_xor_b32 %7, %7, %8\n\t.endr"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                     : "v"(b), "v"(c), "s"(sv) : "vcc", "s20", "s21", "s22", "s23", "s24", "v40", "v41");
    out[blockIdx.x * 1024 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}
__global__ void __launch_bounds__(1024) k_lshl_iv(uint32_t* out, uint32_t n, uint32_t sv) {
    uint32_t a0 = threadIdx.x, a1 = a0 * 3u, a2 = a0 * 5u, a3 = a0 * 7u, a4 = a0 + 9u, a5 = a0 + 11u, a6 = a0 ^ 13u, a7 = a0 + 17u;
    uint32_t b = threadIdx.x * 2654435761u, c = blockIdx.x + 12345u;
    for (uint32_t i = 0; i < n; ++i)
        asm volatile("s_mov_b64 s[22:23], 0x55\n\ts_mov_b32 s24, 5\n\ts_nop 4\n\t.rept 16\n\tv_lshlrev_b32 %0, 3, %0\n\tv_lshlrev_b32 %1, 3, %1\n\tv_lshlrev_b32 %2, 3, %2\n\tv_lshlrev_b32 %3, 3, %3\n\tv_lshlrev_b32 %4, 3, %4\n\tv_lshlrev_b32 %5, 3, %5\n\tv_lshlrev_b32 %6, 3, %6\n\tv_lshlrev_b32 %7, 3, %7\n\t.endr"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                     : "v"(b), "v"(c), "s"(sv) : "vcc", "s20", "s21", "s22", "s23", "s24", "v40", "v41");
    out[blockIdx.x * 1024 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}
__global__ void __launch_bounds__(1024) k_lshl_vv(uint32_t* out, uint32_t n, uint32_t sv) {
    uint32_t a0 = threadIdx.x, a1 = a0 * 3u, a2 = a0 * 5u, a3 = a0 * 7u, a4 = a0 + 9u, a5 = a0 + 11u, a6 = a0 ^ 13u, a7 = a0 + 17u;
    uint32_t b = threadIdx.x * 2654435761u, c = blockIdx.x + 12345u;
    for (uint32_t i = 0; i < n; ++i)
        asm volatile("s_mov_b64 s[22:23], 0x55\n\ts_mov_b32 s24, 5\n\ts_nop 4\n\t.rept 16\n\tv_lshlrev_b32 %0, %8, %0\n\tv_lshlrev_b32 %1, %8, %1\n\tv_lshlrev_b32 %2, %8, %2\n\tv_lshlrev_b32 %3, %8, %3\n\tv_lshlrev_b32 %4, %8, %4\n\tv_lshlrev_b32 %5, %8, %5\n\tv_lshlrev_b32 %6, %8, %6\n\tv_lshlrev_b32 %7, %8, %7\n\t.endr"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                     : "v"(b), "v"(c), "s"(sv) : "vcc", "s20", "s21", "s22", "s23", "s24", "v40", "v41");
    out[blockIdx.x * 1024 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}
__global__ void __launch_bounds__(1024) k_lshl_v_x(uint32_t* out, uint32_t n, uint32_t sv) {
    uint32_t a0 = threadIdx.x, a1 = a0 * 3u, a2 = a0 * 5u, a3 = a0 * 7u, a4 = a0 + 9u, a5 = a0 + 11u, a6 = a0 ^ 13u, a7 = a0 + 17u;
    uint32_t b = threadIdx.x * 2654435761u, c = blockIdx.x + 12345u;
    for (uint32_t i = 0; i < n; ++i)
        asm volatile("s_mov_b64 s[22:23], 0x55\n\ts_mov_b32 s24, 5\n\ts_nop 4\n\t.rept 16\n\tv_lshlrev_b32 %0, %0, %8\n\tv_lshlrev_b32 %1, %1, %8\n\tv_lshlrev_b32 %2, %2, %8\n\tv_lshlrev_b32 %3, %3, %8\n\tv_lshlrev_b32 %4, %4, %8\n\tv_lshlrev_b32 %5, %5, %8\n\tv_lshlrev_b32 %6, %6, %8\n\tv_lshlrev_b32 %7, %7, %8\n\t.endr"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                     : "v"(b), "v"(c), "s"(sv) : "vcc", "s20", "s21", "s22", "s23", "s24", "v40", "v41");
    out[blockIdx.x * 1024 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}
__global__ void __launch_bounds__(1024) k_lshr_iv(uint32_t* out, uint32_t n, uint32_t sv) {
    uint32_t a0 = threadIdx.x, a1 = a0 * 3u, a2 = a0 * 5u, a3 = a0 * 7u, a4 = a0 + 9u, a5 = a0 + 11u, a6 = a0 ^ 13u, a7 = a0 + 17u;
    uint32_t b = threadIdx.x * 2654435761u, c = blockIdx.x + 12345u;
    for (uint32_t i = 0; i < n; ++i)
        asm volatile("s_mov_b64 s[22:23], 0x55\n\ts_mov_b32 s24, 5\n\ts_nop 4\n\t.rept 16\n\tv_lshrrev_b32 %0, 5, %0\n\tv_lshrrev_b32 %1, 5, %1\n\tv_lshrrev_b32 %2, 5, %2\n\tv_lshrrev_b32 %3, 5, %3\n\tv_lshrrev_b32 %4, 5, %4\n\tv_lshrrev_b32 %5, 5, %5\n\tv_lshrrev_b32 %6, 5, %6\n\tv_lshrrev_b32 %7, 5, %7\n\t.endr"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                     : "v"(b), "v"(c), "s"(sv) : "vcc", "s20", "s21", "s22", "s23", "s24", "v40", "v41");
    out[blockIdx.x * 1024 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}
__global__ void __launch_bounds__(1024) k_lshr_vv(uint32_t* out, uint32_t n, uint32_t sv) {
    uint32_t a0 = threadIdx.x, a1 = a0 * 3u, a2 = a0 * 5u, a3 = a0 * 7u, a4 = a0 + 9u, a5 = a0 + 11u, a6 = a0 ^ 13u, a7 = a0 + 17u;
    uint32_t b = threadIdx.x * 2654435761u, c = blockIdx.x + 12345u;
    for (uint32_t i = 0; i < n; ++i)
        asm volatile("s_mov_b64 s[22:23], 0x55\n\ts_mov_b32 s24, 5\n\ts_nop 4\n\t.rept 16\n\tv_lshrrev_b32 %0, %8, %0\n\tv_lshrrev_b32 %1, %8, %1\n\tv_lshrrev_b32 %2, %8, %2\n\tv_lshrrev_b32 %3, %8, %3\n\tv_lshrrev_b32 %4, %8, %4\n\tv_lshrrev_b32 %5, %8, %5\n\tv_lshrrev_b32 %6, %8, %6\n\tv_lshrrev_b32 %7, %8, %7\n\t.endr"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                     : "v"(b), "v"(c), "s"(sv) : "vcc", "s20", "s21", "s22", "s23", "s24", "v40", "v41");
    out[blockIdx.x * 1024 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}
__global__ void __launch_bounds__(1024) k_ashr_iv(uint32_t* out, uint32_t n, uint32_t sv) {
    uint32_t a0 = threadIdx.x, a1 = a0 * 3u, a2 = a0 * 5u, a3 = a0 * 7u, a4 = a0 + 9u, a5 = a0 + 11u, a6 = a0 ^ 13u, a7 = a0 + 17u;
    uint32_t b = threadIdx.x * 2654435761u, c = blockIdx.x + 12345u;
    for (uint32_t i = 0; i < n; ++i)
        asm volatile("s_mov_b64 s[22:23], 0x55\n\ts_mov_b32 s24, 5\n\ts_nop 4\n\t.rept 16\n\tv_ashrrev_i32 %0, 31, %0\n\tv_ashrrev_i32 %1, 31, %1\n\tv_ashrrev_i32 %2, 31, %2\n\tv_ashrrev_i32 %3, 31, %3\n\tv_ashrrev_i32 %4, 31, %4\n\tv_ashrrev_i32 %5, 31, %5\n\tv_ashrrev_i32 %6, 31, %6\n\tv_ashrrev_i32 %7, 31, %7\n\t.endr"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                     : "v"(b), "v"(c), "s"(sv) : "vcc", "s20", "s21", "s22", "s23", "s24", "v40", "v41");
    out[blockIdx.x * 1024 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}
__global__ void __launch_bounds__(1024) k_mov_v(uint32_t* out, uint32_t n, uint32_t sv) {
    uint32_t a0 = threadIdx.x, a1 = a0 * 3u, a2 = a0 * 5u, a3 = a0 * 7u, a4 = a0 + 9u, a5 = a0 + 11u, a6 = a0 ^ 13u, a7 = a0 + 17u;
    uint32_t b = threadIdx.x * 2654435761u, c = blockIdx.x + 12345u;
    for (uint32_t i = 0; i < n; ++i)
        asm volatile("s_mov_b64 s[22:23], 0x55\n\ts_mov_b32 s24, 5\n\ts_nop 4\n\t.rept 16\n\tv_mov_b32 %0, %8\n\tv_mov_b32 %1, %8\n\tv_mov_b32 %2, %8\n\tv_mov_b32 %3, %8\n\tv_mov_b32 %4, %8\n\tv_mov_b32 %5, %8\n\tv_mov_b32 %6, %8\n\tv_mov_b32 %7, %8\n\t.endr"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                     : "v"(b), "v"(c), "s"(sv) : "vcc", "s20", "s21", "s22", "s23", "s24", "v40", "v41");
    out[blockIdx.x * 1024 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}
__global__ void __launch_bounds__(1024) k_mov_s(uint32_t* out, uint32_t n, uint32_t sv) {
    uint32_t a0 = threadIdx.x, a1 = a0 * 3u, a2 = a0 * 5u, a3 = a0 * 7u, a4 = a0 + 9u, a5 = a0 + 11u, a6 = a0 ^ 13u, a7 = a0 + 17u;
    uint32_t b = threadIdx.x * 2654435761u, c = blockIdx.x + 12345u;
    for (uint32_t i = 0; i < n; ++i)
        asm volatile("s_mov_b64 s[22:23], 0x55\n\ts_mov_b32 s24, 5\n\ts_nop 4\n\t.rept 16\n\tv_mov_b32 %0, %10\n\tv_mov_b32 %1, %10\n\tv_mov_b32 %2, %10\n\tv_mov_b32 %3, %10\n\tv_mov_b32 %4, %10\n\tv_mov_b32 %5, %10\n\tv_mov_b32 %6, %10\n\tv_mov_b32 %7, %10\n\t.endr"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                     : "v"(b), "v"(c), "s"(sv) : "vcc", "s20", "s21", "s22", "s23", "s24", "v40", "v41");
    out[blockIdx.x * 1024 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}
__global__ void __launch_bounds__(1024) k_not_v(uint32_t* out, uint32_t n, uint32_t sv) {
    uint32_t a0 = threadIdx.x, a1 = a0 * 3u, a2 = a0 * 5u, a3 = a0 * 7u, a4 = a0 + 9u, a5 = a0 + 11u, a6 = a0 ^ 13u, a7 = a0 + 17u;
    uint32_t b = threadIdx.x * 2654435761u, c = blockIdx.x + 12345u;
    for (uint32_t i = 0; i < n; ++i)
        asm volatile("s_mov_b64 s[22:23], 0x55\n\ts_mov_b32 s24, 5\n\ts_nop 4\n\t.rept 16\n\tv_not_b32 %0, %0\n\tv_not_b32 %1, %1\n\tv_not_b32 %2, %2\n\tv_not_b32 %3, %3\n\tv_not_b32 %4, %4\n\tv_not_b32 %5, %5\n\tv_not_b32 %6, %6\n\tv_not_b32 %7, %7\n\t.endr"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                     : "v"(b), "v"(c), "s"(sv) : "vcc", "s20", "s21", "s22", "s23", "s24", "v40", "v41");
    out[blockIdx.x * 1024 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}
__global__ void __launch_bounds__(1024) k_bfrev(uint32_t* out, uint32_t n, uint32_t sv) {
    uint32_t a0 = threadIdx.x, a1 = a0 * 3u, a2 = a0 * 5u, a3 = a0 * 7u, a4 = a0 + 9u, a5 = a0 + 11u, a6 = a0 ^ 13u, a7 = a0 + 17u;
    uint32_t b = threadIdx.x * 2654435761u, c = blockIdx.x + 12345u;
    for (uint32_t i = 0; i < n; ++i)
        asm volatile("s_mov_b64 s[22:23], 0x55\n\ts_mov_b32 s24, 5\n\ts_nop 4\n\t.rept 16\n\tv_bfrev_b32 %0, %0\n\tv_bfrev_b32 %1, %1\n\tv_bfrev_b32 %2, %2\n\tv_bfrev_b32 %3, %3\n\tv_bfrev_b32 %4, %4\n\tv_bfrev_b32 %5, %5\n\tv_bfrev_b32 %6, %6\n\tv_bfrev_b32 %7, %7\n\t.endr"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                     : "v"(b), "v"(c), "s"(sv) : "vcc", "s20", "s21", "s22", "s23", "s24", "v40", "v41");
    out[blockIdx.x * 1024 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}
__global__ void __launch_bounds__(1024) k_bcnt_v0(uint32_t* out, uint32_t n, uint32_t sv) {
    uint32_t a0 = threadIdx.x, a1 = a0 * 3u, a2 = a0 * 5u, a3 = a0 * 7u, a4 = a0 + 9u, a5 = a0 + 11u, a6 = a0 ^ 13u, a7 = a0 + 17u;
    uint32_t b = threadIdx.x * 2654435761u, c = blockIdx.x + 12345u;
    for (uint32_t i = 0; i < n; ++i)
        asm volatile("s_mov_b64 s[22:23], 0x55\n\ts_mov_b32 s24, 5\n\ts_nop 4\n\t.rept 16\n\tv_bcnt_u32_b32 %0, %0, 0\n\tv_bcnt_u32_b32 %1, %1, 0\n\tv_bcnt_u32_b32 %2, %2, 0\n\tv_bcnt_u32_b32 %3, %3, 0\n\tv_bcnt_u32_b32 %4, %4, 0\n\tv_bcnt_u32_b32 %5, %5, 0\n\tv_bcnt_u32_b32 %6, %6, 0\n\tv_bcnt_u32_b32 %7, %7, 0\n\t.endr"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                     : "v"(b), "v"(c), "s"(sv) : "vcc", "s20", "s21", "s22", "s23", "s24", "v40", "v41");
    out[blockIdx.x * 1024 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}
__global__ void __launch_bounds__(1024) k_bcnt_vv(uint32_t* out, uint32_t n, uint32_t sv) {
    uint32_t a0 = threadIdx.x, a1 = a0 * 3u, a2 = a0 * 5u, a3 = a0 * 7u, a4 = a0 + 9u, a5 = a0 + 11u, a6 = a0 ^ 13u, a7 = a0 + 17u;
    uint32_t b = threadIdx.x * 2654435761u, c = blockIdx.x + 12345u;
    for (uint32_t i = 0; i < n; ++i)
        asm volatile("s_mov_b64 s[22:23], 0x55\n\ts_mov_b32 s24, 5\n\ts_nop 4\n\t.rept 16\n\tv_bcnt_u32_b32 %0, %0, %8\n\tv_bcnt_u32_b32 %1, %1, %8\n\tv_bcnt_u32_b32 %2, %2, %8\n\tv_bcnt_u32_b32 %3, %3, %8\n\tv_bcnt_u32_b32 %4, %4, %8\n\tv_bcnt_u32_b32 %5, %5, %8\n\tv_bcnt_u32_b32 %6, %6, %8\n\tv_bcnt_u32_b32 %7, %7, %8\n\t.endr"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                     : "v"(b), "v"(c), "s"(sv) : "vcc", "s20", "s21", "s22", "s23", "s24", "v40", "v41");
    out[blockIdx.x * 1024 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}
__global__ void __launch_bounds__(1024) k_bfe_u(uint32_t* out, uint32_t n, uint32_t sv) {
    uint32_t a0 = threadIdx.x, a1 = a0 * 3u, a2 = a0 * 5u, a3 = a0 * 7u, a4 = a0 + 9u, a5 = a0 + 11u, a6 = a0 ^ 13u, a7 = a0 + 17u;
    uint32_t b = threadIdx.x * 2654435761u, c = blockIdx.x + 12345u;
    for (uint32_t i = 0; i < n; ++i)
        asm volatile("s_mov_b64 s[22:23], 0x55\n\ts_mov_b32 s24, 5\n\ts_nop 4\n\t.rept 16\n\tv_bfe_u32 %0, %0, 5, 14\n\tv_bfe_u32 %1, %1, 5, 14\n\tv_bfe_u32 %2, %2, 5, 14\n\tv_bfe_u32 %3, %3, 5, 14\n\tv_bfe_u32 %4, %4, 5, 14\n\tv_bfe_u32 %5, %5, 5, 14\n\tv_bfe_u32 %6, %6, 5, 14\n\tv_bfe_u32 %7, %7, 5, 14\n\t.endr"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                     : "v"(b), "v"(c), "s"(sv) : "vcc", "s20", "s21", "s22", "s23", "s24", "v40", "v41");
    out[blockIdx.x * 1024 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}
__global__ void __launch_bounds__(1024) k_bfe_i(uint32_t* out, uint32_t n, uint32_t sv) {
    uint32_t a0 = threadIdx.x, a1 = a0 * 3u, a2 = a0 * 5u, a3 = a0 * 7u, a4 = a0 + 9u, a5 = a0 + 11u, a6 = a0 ^ 13u, a7 = a0 + 17u;
    uint32_t b = threadIdx.x * 2654435761u, c = blockIdx.x + 12345u;
    for (uint32_t i = 0; i < n; ++i)
        asm volatile("s_mov_b64 s[22:23], 0x55\n\ts_mov_b32 s24, 5\n\ts_nop 4\n\t.rept 16\n\tv_bfe_i32 %0, %0, 7, 1\n\tv_bfe_i32 %1, %1, 7, 1\n\tv_bfe_i32 %2, %2, 7, 1\n\tv_bfe_i32 %3, %3, 7, 1\n\tv_bfe_i32 %4, %4, 7, 1\n\tv_bfe_i32 %5, %5, 7, 1\n\tv_bfe_i32 %6, %6, 7, 1\n\tv_bfe_i32 %7, %7, 7, 1\n\t.endr"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                     : "v"(b), "v"(c), "s"(sv) : "vcc", "s20", "s21", "s22", "s23", "s24", "v40", "v41");
    out[blockIdx.x * 1024 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}
__global__ void __launch_bounds__(1024) k_min_vv(uint32_t* out, uint32_t n, uint32_t sv) {
    uint32_t a0 = threadIdx.x, a1 = a0 * 3u, a2 = a0 * 5u, a3 = a0 * 7u, a4 = a0 + 9u, a5 = a0 + 11u, a6 = a0 ^ 13u, a7 = a0 + 17u;
    uint32_t b = threadIdx.x * 2654435761u, c = blockIdx.x + 12345u;
    for (uint32_t i = 0; i < n; ++i)
        asm volatile("s_mov_b64 s[22:23], 0x55\n\ts_mov_b32 s24, 5\n\ts_nop 4\n\t.rept 16\n\tv_min_u32 %0, %0, %8\n\tv_min_u32 %1, %1, %8\n\tv_min_u32 %2, %2, %8\n\tv_min_u32 %3, %3, %8\n\tv_min_u32 %4, %4, %8\n\tv_min_u32 %5, %5, %8\n\tv_min_u32 %6, %6, %8\n\tv_min_u32 %7, %7, %8\n\t.endr"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                     : "v"(b), "v"(c), "s"(sv) : "vcc", "s20", "s21", "s22", "s23", "s24", "v40", "v41");
    out[blockIdx.x * 1024 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}
__global__ void __launch_bounds__(1024) k_max_sv(uint32_t* out, uint32_t n, uint32_t sv) {
    uint32_t a0 = threadIdx.x, a1 = a0 * 3u, a2 = a0 * 5u, a3 = a0 * 7u, a4 = a0 + 9u, a5 = a0 + 11u, a6 = a0 ^ 13u, a7 = a0 + 17u;
    uint32_t b = threadIdx.x * 2654435761u, c = blockIdx.x + 12345u;
    for (uint32_t i = 0; i < n; ++i)
        asm volatile("s_mov_b64 s[22:23], 0x55\n\ts_mov_b32 s24, 5\n\ts_nop 4\n\t.rept 16\n\tv_max_u32 %0, %10, %0\n\tv_max_u32 %1, %10, %1\n\tv_max_u32 %2, %10, %2\n\tv_max_u32 %3, %10, %3\n\tv_max_u32 %4, %10, %4\n\tv_max_u32 %5, %10, %5\n\tv_max_u32 %6, %10, %6\n\tv_max_u32 %7, %10, %7\n\t.endr"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                     : "v"(b), "v"(c), "s"(sv) : "vcc", "s20", "s21", "s22", "s23", "s24", "v40", "v41");
    out[blockIdx.x * 1024 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}
__global__ void __launch_bounds__(1024) k_mul24(uint32_t* out, uint32_t n, uint32_t sv) {
    uint32_t a0 = threadIdx.x, a1 = a0 * 3u, a2 = a0 * 5u, a3 = a0 * 7u, a4 = a0 + 9u, a5 = a0 + 11u, a6 = a0 ^ 13u, a7 = a0 + 17u;
    uint32_t b = threadIdx.x * 2654435761u, c = blockIdx.x + 12345u;
    for (uint32_t i = 0; i < n; ++i)
        asm volatile("s_mov_b64 s[22:23], 0x55\n\ts_mov_b32 s24, 5\n\ts_nop 4\n\t.rept 16\n\tv_mul_u32_u24 %0, %0, %8\n\tv_mul_u32_u24 %1, %1, %8\n\tv_mul_u32_u24 %2, %2, %8\n\tv_mul_u32_u24 %3, %3, %8\n\tv_mul_u32_u24 %4, %4, %8\n\tv_mul_u32_u24 %5, %5, %8\n\tv_mul_u32_u24 %6, %6, %8\n\tv_mul_u32_u24 %7, %7, %8\n\t.endr"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                     : "v"(b), "v"(c), "s"(sv) : "vcc", "s20", "s21", "s22", "s23", "s24", "v40", "v41");
    out[blockIdx.x * 1024 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}
__global__ void __launch_bounds__(1024) k_mad24(uint32_t* out, uint32_t n, uint32_t sv) {
    uint32_t a0 = threadIdx.x, a1 = a0 * 3u, a2 = a0 * 5u, a3 = a0 * 7u, a4 = a0 + 9u, a5 = a0 + 11u, a6 = a0 ^ 13u, a7 = a0 + 17u;
    uint32_t b = threadIdx.x * 2654435761u, c = blockIdx.x + 12345u;
    for (uint32_t i = 0; i < n; ++i)
        asm volatile("s_mov_b64 s[22:23], 0x55\n\ts_mov_b32 s24, 5\n\ts_nop 4\n\t.rept 16\n\tv_mad_u32_u24 %0, %0, %8, %9\n\tv_mad_u32_u24 %1, %1, %8, %9\n\tv_mad_u32_u24 %2, %2, %8, %9\n\tv_mad_u32_u24 %3, %3, %8, %9\n\tv_mad_u32_u24 %4, %4, %8, %9\n\tv_mad_u32_u24 %5, %5, %8, %9\n\tv_mad_u32_u24 %6, %6, %8, %9\n\tv_mad_u32_u24 %7, %7, %8, %9\n\t.endr"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                     : "v"(b), "v"(c), "s"(sv) : "vcc", "s20", "s21", "s22", "s23", "s24", "v40", "v41");
    out[blockIdx.x * 1024 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}
__global__ void __launch_bounds__(1024) k_lshl_or(uint32_t* out, uint32_t n, uint32_t sv) {
    uint32_t a0 = threadIdx.x, a1 = a0 * 3u, a2 = a0 * 5u, a3 = a0 * 7u, a4 = a0 + 9u, a5 = a0 + 11u, a6 = a0 ^ 13u, a7 = a0 + 17u;
    uint32_t b = threadIdx.x * 2654435761u, c = blockIdx.x + 12345u;
    for (uint32_t i = 0; i < n; ++i)
        asm volatile("s_mov_b64 s[22:23], 0x55\n\ts_mov_b32 s24, 5\n\ts_nop 4\n\t.rept 16\n\tv_lshl_or_b32 %0, %0, 1, %8\n\tv_lshl_or_b32 %1, %1, 1, %8\n\tv_lshl_or_b32 %2, %2, 1, %8\n\tv_lshl_or_b32 %3, %3, 1, %8\n\tv_lshl_or_b32 %4, %4, 1, %8\n\tv_lshl_or_b32 %5, %5, 1, %8\n\tv_lshl_or_b32 %6, %6, 1, %8\n\tv_lshl_or_b32 %7, %7, 1, %8\n\t.endr"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                     : "v"(b), "v"(c), "s"(sv) : "vcc", "s20", "s21", "s22", "s23", "s24", "v40", "v41");
    out[blockIdx.x * 1024 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}
__global__ void __launch_bounds__(1024) k_and_or(uint32_t* out, uint32_t n, uint32_t sv) {
    uint32_t a0 = threadIdx.x, a1 = a0 * 3u, a2 = a0 * 5u, a3 = a0 * 7u, a4 = a0 + 9u, a5 = a0 + 11u, a6 = a0 ^ 13u, a7 = a0 + 17u;
    uint32_t b = threadIdx.x * 2654435761u, c = blockIdx.x + 12345u;
    for (uint32_t i = 0; i < n; ++i)
        asm volatile("s_mov_b64 s[22:23], 0x55\n\ts_mov_b32 s24, 5\n\ts_nop 4\n\t.rept 16\n\tv_and_or_b32 %0, %0, %8, %9\n\tv_and_or_b32 %1, %1, %8, %9\n\tv_and_or_b32 %2, %2, %8, %9\n\tv_and_or_b32 %3, %3, %8, %9\n\tv_and_or_b32 %4, %4, %8, %9\n\tv_and_or_b32 %5, %5, %8, %9\n\tv_and_or_b32 %6, %6, %8, %9\n\tv_and_or_b32 %7, %7, %8, %9\n\t.endr"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                     : "v"(b), "v"(c), "s"(sv) : "vcc", "s20", "s21", "s22", "s23", "s24", "v40", "v41");
    out[blockIdx.x * 1024 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}
__global__ void __launch_bounds__(1024) k_or3(uint32_t* out, uint32_t n, uint32_t sv) {
    uint32_t a0 = threadIdx.x, a1 = a0 * 3u, a2 = a0 * 5u, a3 = a0 * 7u, a4 = a0 + 9u, a5 = a0 + 11u, a6 = a0 ^ 13u, a7 = a0 + 17u;
    uint32_t b = threadIdx.x * 2654435761u, c = blockIdx.x + 12345u;
    for (uint32_t i = 0; i < n; ++i)
        asm volatile("s_mov_b64 s[22:23], 0x55\n\ts_mov_b32 s24, 5\n\ts_nop 4\n\t.rept 16\n\tv_or3_b32 %0, %0, %8, %9\n\tv_or3_b32 %1, %1, %8, %9\n\tv_or3_b32 %2, %2, %8, %9\n\tv_or3_b32 %3, %3, %8, %9\n\tv_or3_b32 %4, %4, %8, %9\n\tv_or3_b32 %5, %5, %8, %9\n\tv_or3_b32 %6, %6, %8, %9\n\tv_or3_b32 %7, %7, %8, %9\n\t.endr"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                     : "v"(b), "v"(c), "s"(sv) : "vcc", "s20", "s21", "s22", "s23", "s24", "v40", "v41");
    out[blockIdx.x * 1024 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}
__global__ void __launch_bounds__(1024) k_add3(uint32_t* out, uint32_t n, uint32_t sv) {
    uint32_t a0 = threadIdx.x, a1 = a0 * 3u, a2 = a0 * 5u, a3 = a0 * 7u, a4 = a0 + 9u, a5 = a0 + 11u, a6 = a0 ^ 13u, a7 = a0 + 17u;
    uint32_t b = threadIdx.x * 2654435761u, c = blockIdx.x + 12345u;
    for (uint32_t i = 0; i < n; ++i)
        asm volatile("s_mov_b64 s[22:23], 0x55\n\ts_mov_b32 s24, 5\n\ts_nop 4\n\t.rept 16\n\tv_add3_u32 %0, %0, %8, %9\n\tv_add3_u32 %1, %1, %8, %9\n\tv_add3_u32 %2, %2, %8, %9\n\tv_add3_u32 %3, %3, %8, %9\n\tv_add3_u32 %4, %4, %8, %9\n\tv_add3_u32 %5, %5, %8, %9\n\tv_add3_u32 %6, %6, %8, %9\n\tv_add3_u32 %7, %7, %8, %9\n\t.endr"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                     : "v"(b), "v"(c), "s"(sv) : "vcc", "s20", "s21", "s22", "s23", "s24", "v40", "v41");
    out[blockIdx.x * 1024 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}
__global__ void __launch_bounds__(1024) k_add_lshl(uint32_t* out, uint32_t n, uint32_t sv) {
    uint32_t a0 = threadIdx.x, a1 = a0 * 3u, a2 = a0 * 5u, a3 = a0 * 7u, a4 = a0 + 9u, a5 = a0 + 11u, a6 = a0 ^ 13u, a7 = a0 + 17u;
    uint32_t b = threadIdx.x * 2654435761u, c = blockIdx.x + 12345u;
    for (uint32_t i = 0; i < n; ++i)
        asm volatile("s_mov_b64 s[22:23], 0x55\n\ts_mov_b32 s24, 5\n\ts_nop 4\n\t.rept 16\n\tv_add_lshl_u32 %0, %0, %8, 3\n\tv_add_lshl_u32 %1, %1, %8, 3\n\tv_add_lshl_u32 %2, %2, %8, 3\n\tv_add_lshl_u32 %3, %3, %8, 3\n\tv_add_lshl_u32 %4, %4, %8, 3\n\tv_add_lshl_u32 %5, %5, %8, 3\n\tv_add_lshl_u32 %6, %6, %8, 3\n\tv_add_lshl_u32 %7, %7, %8, 3\n\t.endr"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                     : "v"(b), "v"(c), "s"(sv) : "vcc", "s20", "s21", "s22", "s23", "s24", "v40", "v41");
    out[blockIdx.x * 1024 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}
__global__ void __launch_bounds__(1024) k_lshl_add(uint32_t* out, uint32_t n, uint32_t sv) {
    uint32_t a0 = threadIdx.x, a1 = a0 * 3u, a2 = a0 * 5u, a3 = a0 * 7u, a4 = a0 + 9u, a5 = a0 + 11u, a6 = a0 ^ 13u, a7 = a0 + 17u;
    uint32_t b = threadIdx.x * 2654435761u, c = blockIdx.x + 12345u;
    for (uint32_t i = 0; i < n; ++i)
        asm volatile("s_mov_b64 s[22:23], 0x55\n\ts_mov_b32 s24, 5\n\ts_nop 4\n\t.rept 16\n\tv_lshl_add_u32 %0, %0, 3, %8\n\tv_lshl_add_u32 %1, %1, 3, %8\n\tv_lshl_add_u32 %2, %2, 3, %8\n\tv_lshl_add_u32 %3, %3, 3, %8\n\tv_lshl_add_u32 %4, %4, 3, %8\n\tv_lshl_add_u32 %5, %5, 3, %8\n\tv_lshl_add_u32 %6, %6, 3, %8\n\tv_lshl_add_u32 %7, %7, 3, %8\n\t.endr"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                     : "v"(b), "v"(c), "s"(sv) : "vcc", "s20", "s21", "s22", "s23", "s24", "v40", "v41");
    out[blockIdx.x * 1024 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}
__global__ void __launch_bounds__(1024) k_xad(uint32_t* out, uint32_t n, uint32_t sv) {
    uint32_t a0 = threadIdx.x, a1 = a0 * 3u, a2 = a0 * 5u, a3 = a0 * 7u, a4 = a0 + 9u, a5 = a0 + 11u, a6 = a0 ^ 13u, a7 = a0 + 17u;
    uint32_t b = threadIdx.x * 2654435761u, c = blockIdx.x + 12345u;
    for (uint32_t i = 0; i < n; ++i)
        asm volatile("s_mov_b64 s[22:23], 0x55\n\ts_mov_b32 s24, 5\n\ts_nop 4\n\t.rept 16\n\tv_xad_u32 %0, %0, %8, %9\n\tv_xad_u32 %1, %1, %8, %9\n\tv_xad_u32 %2, %2, %8, %9\n\tv_xad_u32 %3, %3, %8, %9\n\tv_xad_u32 %4, %4, %8, %9\n\tv_xad_u32 %5, %5, %8, %9\n\tv_xad_u32 %6, %6, %8, %9\n\tv_xad_u32 %7, %7, %8, %9\n\t.endr"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                     : "v"(b), "v"(c), "s"(sv) : "vcc", "s20", "s21", "s22", "s23", "s24", "v40", "v41");
    out[blockIdx.x * 1024 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}
__global__ void __launch_bounds__(1024) k_bfi(uint32_t* out, uint32_t n, uint32_t sv) {
    uint32_t a0 = threadIdx.x, a1 = a0 * 3u, a2 = a0 * 5u, a3 = a0 * 7u, a4 = a0 + 9u, a5 = a0 + 11u, a6 = a0 ^ 13u, a7 = a0 + 17u;
    uint32_t b = threadIdx.x * 2654435761u, c = blockIdx.x + 12345u;
    for (uint32_t i = 0; i < n; ++i)
        asm volatile("s_mov_b64 s[22:23], 0x55\n\ts_mov_b32 s24, 5\n\ts_nop 4\n\t.rept 16\n\tv_bfi_b32 %0, %8, %9, %0\n\tv_bfi_b32 %1, %8, %9, %1\n\tv_bfi_b32 %2, %8, %9, %2\n\tv_bfi_b32 %3, %8, %9, %3\n\tv_bfi_b32 %4, %8, %9, %4\n\tv_bfi_b32 %5, %8, %9, %5\n\tv_bfi_b32 %6, %8, %9, %6\n\tv_bfi_b32 %7, %8, %9, %7\n\t.endr"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                     : "v"(b), "v"(c), "s"(sv) : "vcc", "s20", "s21", "s22", "s23", "s24", "v40", "v41");
    out[blockIdx.x * 1024 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}
__global__ void __launch_bounds__(1024) k_alignbit(uint32_t* out, uint32_t n, uint32_t sv) {
    uint32_t a0 = threadIdx.x, a1 = a0 * 3u, a2 = a0 * 5u, a3 = a0 * 7u, a4 = a0 + 9u, a5 = a0 + 11u, a6 = a0 ^ 13u, a7 = a0 + 17u;
    uint32_t b = threadIdx.x * 2654435761u, c = blockIdx.x + 12345u;
    for (uint32_t i = 0; i < n; ++i)
        asm volatile("s_mov_b64 s[22:23], 0x55\n\ts_mov_b32 s24, 5\n\ts_nop 4\n\t.rept 16\n\tv_alignbit_b32 %0, %0, %8, 31\n\tv_alignbit_b32 %1, %1, %8, 31\n\tv_alignbit_b32 %2, %2, %8, 31\n\tv_alignbit_b32 %3, %3, %8, 31\n\tv_alignbit_b32 %4, %4, %8, 31\n\tv_alignbit_b32 %5, %5, %8, 31\n\tv_alignbit_b32 %6, %6, %8, 31\n\tv_alignbit_b32 %7, %7, %8, 31\n\t.endr"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                     : "v"(b), "v"(c), "s"(sv) : "vcc", "s20", "s21", "s22", "s23", "s24", "v40", "v41");
    out[blockIdx.x * 1024 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}
__global__ void __launch_bounds__(1024) k_perm(uint32_t* out, uint32_t n, uint32_t sv) {
    uint32_t a0 = threadIdx.x, a1 = a0 * 3u, a2 = a0 * 5u, a3 = a0 * 7u, a4 = a0 + 9u, a5 = a0 + 11u, a6 = a0 ^ 13u, a7 = a0 + 17u;
    uint32_t b = threadIdx.x * 2654435761u, c = blockIdx.x + 12345u;
    for (uint32_t i = 0; i < n; ++i)
        asm volatile("s_mov_b64 s[22:23], 0x55\n\ts_mov_b32 s24, 5\n\ts_nop 4\n\t.rept 16\n\tv_perm_b32 %0, %0, %8, %9\n\tv_perm_b32 %1, %1, %8, %9\n\tv_perm_b32 %2, %2, %8, %9\n\tv_perm_b32 %3, %3, %8, %9\n\tv_perm_b32 %4, %4, %8, %9\n\tv_perm_b32 %5, %5, %8, %9\n\tv_perm_b32 %6, %6, %8, %9\n\tv_perm_b32 %7, %7, %8, %9\n\t.endr"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                     : "v"(b), "v"(c), "s"(sv) : "vcc", "s20", "s21", "s22", "s23", "s24", "v40", "v41");
    out[blockIdx.x * 1024 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}
__global__ void __launch_bounds__(1024) k_mbcnt_lo(uint32_t* out, uint32_t n, uint32_t sv) {
    uint32_t a0 = threadIdx.x, a1 = a0 * 3u, a2 = a0 * 5u, a3 = a0 * 7u, a4 = a0 + 9u, a5 = a0 + 11u, a6 = a0 ^ 13u, a7 = a0 + 17u;
    uint32_t b = threadIdx.x * 2654435761u, c = blockIdx.x + 12345u;
    for (uint32_t i = 0; i < n; ++i)
        asm volatile("s_mov_b64 s[22:23], 0x55\n\ts_mov_b32 s24, 5\n\ts_nop 4\n\t.rept 16\n\tv_mbcnt_lo_u32_b32 %0, %8, %0\n\tv_mbcnt_lo_u32_b32 %1, %8, %1\n\tv_mbcnt_lo_u32_b32 %2, %8, %2\n\tv_mbcnt_lo_u32_b32 %3, %8, %3\n\tv_mbcnt_lo_u32_b32 %4, %8, %4\n\tv_mbcnt_lo_u32_b32 %5, %8, %5\n\tv_mbcnt_lo_u32_b32 %6, %8, %6\n\tv_mbcnt_lo_u32_b32 %7, %8, %7\n\t.endr"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                     : "v"(b), "v"(c), "s"(sv) : "vcc", "s20", "s21", "s22", "s23", "s24", "v40", "v41");
    out[blockIdx.x * 1024 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}
__global__ void __launch_bounds__(1024) k_cndmask_s(uint32_t* out, uint32_t n, uint32_t sv) {
    uint32_t a0 = threadIdx.x, a1 = a0 * 3u, a2 = a0 * 5u, a3 = a0 * 7u, a4 = a0 + 9u, a5 = a0 + 11u, a6 = a0 ^ 13u, a7 = a0 + 17u;
    uint32_t b = threadIdx.x * 2654435761u, c = blockIdx.x + 12345u;
    for (uint32_t i = 0; i < n; ++i)
        asm volatile("s_mov_b64 s[22:23], 0x55\n\ts_mov_b32 s24, 5\n\ts_nop 4\n\t.rept 16\n\tv_cndmask_b32 %0, %0, %8, s[22:23]\n\tv_cndmask_b32 %1, %1, %8, s[22:23]\n\tv_cndmask_b32 %2, %2, %8, s[22:23]\n\tv_cndmask_b32 %3, %3, %8, s[22:23]\n\tv_cndmask_b32 %4, %4, %8, s[22:23]\n\tv_cndmask_b32 %5, %5, %8, s[22:23]\n\tv_cndmask_b32 %6, %6, %8, s[22:23]\n\tv_cndmask_b32 %7, %7, %8, s[22:23]\n\t.endr"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                     : "v"(b), "v"(c), "s"(sv) : "vcc", "s20", "s21", "s22", "s23", "s24", "v40", "v41");
    out[blockIdx.x * 1024 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}
__global__ void __launch_bounds__(1024) k_cmp_vcc(uint32_t* out, uint32_t n, uint32_t sv) {
    uint32_t a0 = threadIdx.x, a1 = a0 * 3u, a2 = a0 * 5u, a3 = a0 * 7u, a4 = a0 + 9u, a5 = a0 + 11u, a6 = a0 ^ 13u, a7 = a0 + 17u;
    uint32_t b = threadIdx.x * 2654435761u, c = blockIdx.x + 12345u;
    for (uint32_t i = 0; i < n; ++i)
        asm volatile("s_mov_b64 s[22:23], 0x55\n\ts_mov_b32 s24, 5\n\ts_nop 4\n\t.rept 16\n\tv_cmp_gt_i32 vcc, 0, %0\n\tv_cmp_gt_i32 vcc, 0, %1\n\tv_cmp_gt_i32 vcc, 0, %2\n\tv_cmp_gt_i32 vcc, 0, %3\n\tv_cmp_gt_i32 vcc, 0, %4\n\tv_cmp_gt_i32 vcc, 0, %5\n\tv_cmp_gt_i32 vcc, 0, %6\n\tv_cmp_gt_i32 vcc, 0, %7\n\t.endr"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                     : "v"(b), "v"(c), "s"(sv) : "vcc", "s20", "s21", "s22", "s23", "s24", "v40", "v41");
    out[blockIdx.x * 1024 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}
__global__ void __launch_bounds__(1024) k_cmp_s(uint32_t* out, uint32_t n, uint32_t sv) {
    uint32_t a0 = threadIdx.x, a1 = a0 * 3u, a2 = a0 * 5u, a3 = a0 * 7u, a4 = a0 + 9u, a5 = a0 + 11u, a6 = a0 ^ 13u, a7 = a0 + 17u;
    uint32_t b = threadIdx.x * 2654435761u, c = blockIdx.x + 12345u;
    for (uint32_t i = 0; i < n; ++i)
        asm volatile("s_mov_b64 s[22:23], 0x55\n\ts_mov_b32 s24, 5\n\ts_nop 4\n\t.rept 16\n\tv_cmp_gt_i32 s[20:21], 0, %0\n\tv_cmp_gt_i32 s[20:21], 0, %1\n\tv_cmp_gt_i32 s[20:21], 0, %2\n\tv_cmp_gt_i32 s[20:21], 0, %3\n\tv_cmp_gt_i32 s[20:21], 0, %4\n\tv_cmp_gt_i32 s[20:21], 0, %5\n\tv_cmp_gt_i32 s[20:21], 0, %6\n\tv_cmp_gt_i32 s[20:21], 0, %7\n\t.endr"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                     : "v"(b), "v"(c), "s"(sv) : "vcc", "s20", "s21", "s22", "s23", "s24", "v40", "v41");
    out[blockIdx.x * 1024 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}
__global__ void __launch_bounds__(1024) k_cmp_lt_u_vv(uint32_t* out, uint32_t n, uint32_t sv) {
    uint32_t a0 = threadIdx.x, a1 = a0 * 3u, a2 = a0 * 5u, a3 = a0 * 7u, a4 = a0 + 9u, a5 = a0 + 11u, a6 = a0 ^ 13u, a7 = a0 + 17u;
    uint32_t b = threadIdx.x * 2654435761u, c = blockIdx.x + 12345u;
    for (uint32_t i = 0; i < n; ++i)
        asm volatile("s_mov_b64 s[22:23], 0x55\n\ts_mov_b32 s24, 5\n\ts_nop 4\n\t.rept 16\n\tv_cmp_lt_u32 vcc, %0, %8\n\tv_cmp_lt_u32 vcc, %1, %8\n\tv_cmp_lt_u32 vcc, %2, %8\n\tv_cmp_lt_u32 vcc, %3, %8\n\tv_cmp_lt_u32 vcc, %4, %8\n\tv_cmp_lt_u32 vcc, %5, %8\n\tv_cmp_lt_u32 vcc, %6, %8\n\tv_cmp_lt_u32 vcc, %7, %8\n\t.endr"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                     : "v"(b), "v"(c), "s"(sv) : "vcc", "s20", "s21", "s22", "s23", "s24", "v40", "v41");
    out[blockIdx.x * 1024 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}
__global__ void __launch_bounds__(1024) k_addc(uint32_t* out, uint32_t n, uint32_t sv) {
    uint32_t a0 = threadIdx.x, a1 = a0 * 3u, a2 = a0 * 5u, a3 = a0 * 7u, a4 = a0 + 9u, a5 = a0 + 11u, a6 = a0 ^ 13u, a7 = a0 + 17u;
    uint32_t b = threadIdx.x * 2654435761u, c = blockIdx.x + 12345u;
    for (uint32_t i = 0; i < n; ++i)
        asm volatile("s_mov_b64 s[22:23], 0x55\n\ts_mov_b32 s24, 5\n\ts_nop 4\n\t.rept 16\n\tv_addc_co_u32 %0, vcc, %0, %8, vcc\n\tv_addc_co_u32 %1, vcc, %1, %8, vcc\n\tv_addc_co_u32 %2, vcc, %2, %8, vcc\n\tv_addc_co_u32 %3, vcc, %3, %8, vcc\n\tv_addc_co_u32 %4, vcc, %4, %8, vcc\n\tv_addc_co_u32 %5, vcc, %5, %8, vcc\n\tv_addc_co_u32 %6, vcc, %6, %8, vcc\n\tv_addc_co_u32 %7, vcc, %7, %8, vcc\n\t.endr"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                     : "v"(b), "v"(c), "s"(sv) : "vcc", "s20", "s21", "s22", "s23", "s24", "v40", "v41");
    out[blockIdx.x * 1024 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}
__global__ void __launch_bounds__(1024) k_add_co(uint32_t* out, uint32_t n, uint32_t sv) {
    uint32_t a0 = threadIdx.x, a1 = a0 * 3u, a2 = a0 * 5u, a3 = a0 * 7u, a4 = a0 + 9u, a5 = a0 + 11u, a6 = a0 ^ 13u, a7 = a0 + 17u;
    uint32_t b = threadIdx.x * 2654435761u, c = blockIdx.x + 12345u;
    for (uint32_t i = 0; i < n; ++i)
        asm volatile("s_mov_b64 s[22:23], 0x55\n\ts_mov_b32 s24, 5\n\ts_nop 4\n\t.rept 16\n\tv_add_co_u32 %0, vcc, %0, %8\n\tv_add_co_u32 %1, vcc, %1, %8\n\tv_add_co_u32 %2, vcc, %2, %8\n\tv_add_co_u32 %3, vcc, %3, %8\n\tv_add_co_u32 %4, vcc, %4, %8\n\tv_add_co_u32 %5, vcc, %5, %8\n\tv_add_co_u32 %6, vcc, %6, %8\n\tv_add_co_u32 %7, vcc, %7, %8\n\t.endr"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                     : "v"(b), "v"(c), "s"(sv) : "vcc", "s20", "s21", "s22", "s23", "s24", "v40", "v41");
    out[blockIdx.x * 1024 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}
__global__ void __launch_bounds__(1024) k_pk_add16(uint32_t* out, uint32_t n, uint32_t sv) {
    uint32_t a0 = threadIdx.x, a1 = a0 * 3u, a2 = a0 * 5u, a3 = a0 * 7u, a4 = a0 + 9u, a5 = a0 + 11u, a6 = a0 ^ 13u, a7 = a0 + 17u;
    uint32_t b = threadIdx.x * 2654435761u, c = blockIdx.x + 12345u;
    for (uint32_t i = 0; i < n; ++i)
        asm volatile("s_mov_b64 s[22:23], 0x55\n\ts_mov_b32 s24, 5\n\ts_nop 4\n\t.rept 16\n\tv_pk_add_u16 %0, %0, %8\n\tv_pk_add_u16 %1, %1, %8\n\tv_pk_add_u16 %2, %2, %8\n\tv_pk_add_u16 %3, %3, %8\n\tv_pk_add_u16 %4, %4, %8\n\tv_pk_add_u16 %5, %5, %8\n\tv_pk_add_u16 %6, %6, %8\n\tv_pk_add_u16 %7, %7, %8\n\t.endr"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                     : "v"(b), "v"(c), "s"(sv) : "vcc", "s20", "s21", "s22", "s23", "s24", "v40", "v41");
    out[blockIdx.x * 1024 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}
__global__ void __launch_bounds__(1024) k_pk_lshl16(uint32_t* out, uint32_t n, uint32_t sv) {
    uint32_t a0 = threadIdx.x, a1 = a0 * 3u, a2 = a0 * 5u, a3 = a0 * 7u, a4 = a0 + 9u, a5 = a0 + 11u, a6 = a0 ^ 13u, a7 = a0 + 17u;
    uint32_t b = threadIdx.x * 2654435761u, c = blockIdx.x + 12345u;
    for (uint32_t i = 0; i < n; ++i)
        asm volatile("s_mov_b64 s[22:23], 0x55\n\ts_mov_b32 s24, 5\n\ts_nop 4\n\t.rept 16\n\tv_pk_lshlrev_b16 %0, %8, %0\n\tv_pk_lshlrev_b16 %1, %8, %1\n\tv_pk_lshlrev_b16 %2, %8, %2\n\tv_pk_lshlrev_b16 %3, %8, %3\n\tv_pk_lshlrev_b16 %4, %8, %4\n\tv_pk_lshlrev_b16 %5, %8, %5\n\tv_pk_lshlrev_b16 %6, %8, %6\n\tv_pk_lshlrev_b16 %7, %8, %7\n\t.endr"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                     : "v"(b), "v"(c), "s"(sv) : "vcc", "s20", "s21", "s22", "s23", "s24", "v40", "v41");
    out[blockIdx.x * 1024 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}
__global__ void __launch_bounds__(1024) k_pk_sub16(uint32_t* out, uint32_t n, uint32_t sv) {
    uint32_t a0 = threadIdx.x, a1 = a0 * 3u, a2 = a0 * 5u, a3 = a0 * 7u, a4 = a0 + 9u, a5 = a0 + 11u, a6 = a0 ^ 13u, a7 = a0 + 17u;
    uint32_t b = threadIdx.x * 2654435761u, c = blockIdx.x + 12345u;
    for (uint32_t i = 0; i < n; ++i)
        asm volatile("s_mov_b64 s[22:23], 0x55\n\ts_mov_b32 s24, 5\n\ts_nop 4\n\t.rept 16\n\tv_pk_sub_u16 %0, %0, %8\n\tv_pk_sub_u16 %1, %1, %8\n\tv_pk_sub_u16 %2, %2, %8\n\tv_pk_sub_u16 %3, %3, %8\n\tv_pk_sub_u16 %4, %4, %8\n\tv_pk_sub_u16 %5, %5, %8\n\tv_pk_sub_u16 %6, %6, %8\n\tv_pk_sub_u16 %7, %7, %8\n\t.endr"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                     : "v"(b), "v"(c), "s"(sv) : "vcc", "s20", "s21", "s22", "s23", "s24", "v40", "v41");
    out[blockIdx.x * 1024 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}
__global__ void __launch_bounds__(1024) k_add_dpp(uint32_t* out, uint32_t n, uint32_t sv) {
    uint32_t a0 = threadIdx.x, a1 = a0 * 3u, a2 = a0 * 5u, a3 = a0 * 7u, a4 = a0 + 9u, a5 = a0 + 11u, a6 = a0 ^ 13u, a7 = a0 + 17u;
    uint32_t b = threadIdx.x * 2654435761u, c = blockIdx.x + 12345u;
    for (uint32_t i = 0; i < n; ++i)
        asm volatile("s_mov_b64 s[22:23], 0x55\n\ts_mov_b32 s24, 5\n\ts_nop 4\n\t.rept 16\n\tv_add_u32_dpp %0, %8, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\tv_add_u32_dpp %1, %8, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\tv_add_u32_dpp %2, %8, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\tv_add_u32_dpp %3, %8, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\tv_add_u32_dpp %4, %8, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\tv_add_u32_dpp %5, %8, %5 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\tv_add_u32_dpp %6, %8, %6 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\tv_add_u32_dpp %7, %8, %7 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t.endr"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                     : "v"(b), "v"(c), "s"(sv) : "vcc", "s20", "s21", "s22", "s23", "s24", "v40", "v41");
    out[blockIdx.x * 1024 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}
__global__ void __launch_bounds__(1024) k_mov_dpp(uint32_t* out, uint32_t n, uint32_t sv) {
    uint32_t a0 = threadIdx.x, a1 = a0 * 3u, a2 = a0 * 5u, a3 = a0 * 7u, a4 = a0 + 9u, a5 = a0 + 11u, a6 = a0 ^ 13u, a7 = a0 + 17u;
    uint32_t b = threadIdx.x * 2654435761u, c = blockIdx.x + 12345u;
    for (uint32_t i = 0; i < n; ++i)
        asm volatile("s_mov_b64 s[22:23], 0x55\n\ts_mov_b32 s24, 5\n\ts_nop 4\n\t.rept 16\n\tv_mov_b32_dpp %0, %8 row_shr:1 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %1, %8 row_shr:1 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %2, %8 row_shr:1 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %3, %8 row_shr:1 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %4, %8 row_shr:1 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %5, %8 row_shr:1 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %6, %8 row_shr:1 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %7, %8 row_shr:1 row_mask:0xf bank_mask:0xf\n\t.endr"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                     : "v"(b), "v"(c), "s"(sv) : "vcc", "s20", "s21", "s22", "s23", "s24", "v40", "v41");
    out[blockIdx.x * 1024 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}
__global__ void __launch_bounds__(1024) k_add_sdwa(uint32_t* out, uint32_t n, uint32_t sv) {
    uint32_t a0 = threadIdx.x, a1 = a0 * 3u, a2 = a0 * 5u, a3 = a0 * 7u, a4 = a0 + 9u, a5 = a0 + 11u, a6 = a0 ^ 13u, a7 = a0 + 17u;
    uint32_t b = threadIdx.x * 2654435761u, c = blockIdx.x + 12345u;
    for (uint32_t i = 0; i < n; ++i)
        asm volatile("s_mov_b64 s[22:23], 0x55\n\ts_mov_b32 s24, 5\n\ts_nop 4\n\t.rept 16\n\tv_add_u32_sdwa %0, %0, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:WORD_1\n\tv_add_u32_sdwa %1, %1, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:WORD_1\n\tv_add_u32_sdwa %2, %2, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:WORD_1\n\tv_add_u32_sdwa %3, %3, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:WORD_1\n\tv_add_u32_sdwa %4, %4, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:WORD_1\n\tv_add_u32_sdwa %5, %5, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:WORD_1\n\tv_add_u32_sdwa %6, %6, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:WORD_1\n\tv_add_u32_sdwa %7, %7, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:WORD_1\n\t.endr"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                     : "v"(b), "v"(c), "s"(sv) : "vcc", "s20", "s21", "s22", "s23", "s24", "v40", "v41");
    out[blockIdx.x * 1024 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}
__global__ void __launch_bounds__(1024) k_readlane(uint32_t* out, uint32_t n, uint32_t sv) {
    uint32_t a0 = threadIdx.x, a1 = a0 * 3u, a2 = a0 * 5u, a3 = a0 * 7u, a4 = a0 + 9u, a5 = a0 + 11u, a6 = a0 ^ 13u, a7 = a0 + 17u;
    uint32_t b = threadIdx.x * 2654435761u, c = blockIdx.x + 12345u;
    for (uint32_t i = 0; i < n; ++i)
        asm volatile("s_mov_b64 s[22:23], 0x55\n\ts_mov_b32 s24, 5\n\ts_nop 4\n\t.rept 16\n\tv_readlane_b32 s20, %0, 5\n\tv_readlane_b32 s20, %1, 5\n\tv_readlane_b32 s20, %2, 5\n\tv_readlane_b32 s20, %3, 5\n\tv_readlane_b32 s20, %4, 5\n\tv_readlane_b32 s20, %5, 5\n\tv_readlane_b32 s20, %6, 5\n\tv_readlane_b32 s20, %7, 5\n\t.endr"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                     : "v"(b), "v"(c), "s"(sv) : "vcc", "s20", "s21", "s22", "s23", "s24", "v40", "v41");
    out[blockIdx.x * 1024 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}
__global__ void __launch_bounds__(1024) k_writelane(uint32_t* out, uint32_t n, uint32_t sv) {
    uint32_t a0 = threadIdx.x, a1 = a0 * 3u, a2 = a0 * 5u, a3 = a0 * 7u, a4 = a0 + 9u, a5 = a0 + 11u, a6 = a0 ^ 13u, a7 = a0 + 17u;
    uint32_t b = threadIdx.x * 2654435761u, c = blockIdx.x + 12345u;
    for (uint32_t i = 0; i < n; ++i)
        asm volatile("s_mov_b64 s[22:23], 0x55\n\ts_mov_b32 s24, 5\n\ts_nop 4\n\t.rept 16\n\tv_writelane_b32 %0, s24, 3\n\tv_writelane_b32 %1, s24, 3\n\tv_writelane_b32 %2, s24, 3\n\tv_writelane_b32 %3, s24, 3\n\tv_writelane_b32 %4, s24, 3\n\tv_writelane_b32 %5, s24, 3\n\tv_writelane_b32 %6, s24, 3\n\tv_writelane_b32 %7, s24, 3\n\t.endr"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                     : "v"(b), "v"(c), "s"(sv) : "vcc", "s20", "s21", "s22", "s23", "s24", "v40", "v41");
    out[blockIdx.x * 1024 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}
__global__ void __launch_bounds__(1024) k_readfirst(uint32_t* out, uint32_t n, uint32_t sv) {
    uint32_t a0 = threadIdx.x, a1 = a0 * 3u, a2 = a0 * 5u, a3 = a0 * 7u, a4 = a0 + 9u, a5 = a0 + 11u, a6 = a0 ^ 13u, a7 = a0 + 17u;
    uint32_t b = threadIdx.x * 2654435761u, c = blockIdx.x + 12345u;
    for (uint32_t i = 0; i < n; ++i)
        asm volatile("s_mov_b64 s[22:23], 0x55\n\ts_mov_b32 s24, 5\n\ts_nop 4\n\t.rept 16\n\tv_readfirstlane_b32 s20, %0\n\tv_readfirstlane_b32 s20, %1\n\tv_readfirstlane_b32 s20, %2\n\tv_readfirstlane_b32 s20, %3\n\tv_readfirstlane_b32 s20, %4\n\tv_readfirstlane_b32 s20, %5\n\tv_readfirstlane_b32 s20, %6\n\tv_readfirstlane_b32 s20, %7\n\t.endr"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                     : "v"(b), "v"(c), "s"(sv) : "vcc", "s20", "s21", "s22", "s23", "s24", "v40", "v41");
    out[blockIdx.x * 1024 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}
__global__ void __launch_bounds__(1024) k_lshl64(uint32_t* out, uint32_t n, uint32_t sv) {
    uint32_t a0 = threadIdx.x, a1 = a0 * 3u, a2 = a0 * 5u, a3 = a0 * 7u, a4 = a0 + 9u, a5 = a0 + 11u, a6 = a0 ^ 13u, a7 = a0 + 17u;
    uint32_t b = threadIdx.x * 2654435761u, c = blockIdx.x + 12345u;
    for (uint32_t i = 0; i < n; ++i)
        asm volatile("s_mov_b64 s[22:23], 0x55\n\ts_mov_b32 s24, 5\n\ts_nop 4\n\t.rept 16\n\tv_lshlrev_b64 v[40:41], 3, v[40:41]\n\tv_lshlrev_b64 v[40:41], 3, v[40:41]\n\tv_lshlrev_b64 v[40:41], 3, v[40:41]\n\tv_lshlrev_b64 v[40:41], 3, v[40:41]\n\tv_lshlrev_b64 v[40:41], 3, v[40:41]\n\tv_lshlrev_b64 v[40:41], 3, v[40:41]\n\tv_lshlrev_b64 v[40:41], 3, v[40:41]\n\tv_lshlrev_b64 v[40:41], 3, v[40:41]\n\t.endr"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                     : "v"(b), "v"(c), "s"(sv) : "vcc", "s20", "s21", "s22", "s23", "s24", "v40", "v41");
    out[blockIdx.x * 1024 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}
__global__ void __launch_bounds__(1024) k_cvt(uint32_t* out, uint32_t n, uint32_t sv) {
    uint32_t a0 = threadIdx.x, a1 = a0 * 3u, a2 = a0 * 5u, a3 = a0 * 7u, a4 = a0 + 9u, a5 = a0 + 11u, a6 = a0 ^ 13u, a7 = a0 + 17u;
    uint32_t b = threadIdx.x * 2654435761u, c = blockIdx.x + 12345u;
    for (uint32_t i = 0; i < n; ++i)
        asm volatile("s_mov_b64 s[22:23], 0x55\n\ts_mov_b32 s24, 5\n\ts_nop 4\n\t.rept 16\n\tv_cvt_f32_u32 %0, %0\n\tv_cvt_f32_u32 %1, %1\n\tv_cvt_f32_u32 %2, %2\n\tv_cvt_f32_u32 %3, %3\n\tv_cvt_f32_u32 %4, %4\n\tv_cvt_f32_u32 %5, %5\n\tv_cvt_f32_u32 %6, %6\n\tv_cvt_f32_u32 %7, %7\n\t.endr"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                     : "v"(b), "v"(c), "s"(sv) : "vcc", "s20", "s21", "s22", "s23", "s24", "v40", "v41");
    out[blockIdx.x * 1024 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}
__global__ void __launch_bounds__(1024) k_fma(uint32_t* out, uint32_t n, uint32_t sv) {
    uint32_t a0 = threadIdx.x, a1 = a0 * 3u, a2 = a0 * 5u, a3 = a0 * 7u, a4 = a0 + 9u, a5 = a0 + 11u, a6 = a0 ^ 13u, a7 = a0 + 17u;
    uint32_t b = threadIdx.x * 2654435761u, c = blockIdx.x + 12345u;
    for (uint32_t i = 0; i < n; ++i)
        asm volatile("s_mov_b64 s[22:23], 0x55\n\ts_mov_b32 s24, 5\n\ts_nop 4\n\t.rept 16\n\tv_fma_f32 %0, %0, %8, %9\n\tv_fma_f32 %1, %1, %8, %9\n\tv_fma_f32 %2, %2, %8, %9\n\tv_fma_f32 %3, %3, %8, %9\n\tv_fma_f32 %4, %4, %8, %9\n\tv_fma_f32 %5, %5, %8, %9\n\tv_fma_f32 %6, %6, %8, %9\n\tv_fma_f32 %7, %7, %8, %9\n\t.endr"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                     : "v"(b), "v"(c), "s"(sv) : "vcc", "s20", "s21", "s22", "s23", "s24", "v40", "v41");
    out[blockIdx.x * 1024 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}
__global__ void __launch_bounds__(1024) k_mul_f(uint32_t* out, uint32_t n, uint32_t sv) {
    uint32_t a0 = threadIdx.x, a1 = a0 * 3u, a2 = a0 * 5u, a3 = a0 * 7u, a4 = a0 + 9u, a5 = a0 + 11u, a6 = a0 ^ 13u, a7 = a0 + 17u;
    uint32_t b = threadIdx.x * 2654435761u, c = blockIdx.x + 12345u;
    for (uint32_t i = 0; i < n; ++i)
        asm volatile("s_mov_b64 s[22:23], 0x55\n\ts_mov_b32 s24, 5\n\ts_nop 4\n\t.rept 16\n\tv_mul_f32 %0, %0, %8\n\tv_mul_f32 %1, %1, %8\n\tv_mul_f32 %2, %2, %8\n\tv_mul_f32 %3, %3, %8\n\tv_mul_f32 %4, %4, %8\n\tv_mul_f32 %5, %5, %8\n\tv_mul_f32 %6, %6, %8\n\tv_mul_f32 %7, %7, %8\n\t.endr"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                     : "v"(b), "v"(c), "s"(sv) : "vcc", "s20", "s21", "s22", "s23", "s24", "v40", "v41");
    out[blockIdx.x * 1024 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}
__global__ void __launch_bounds__(1024) k_mul_lo(uint32_t* out, uint32_t n, uint32_t sv) {
    uint32_t a0 = threadIdx.x, a1 = a0 * 3u, a2 = a0 * 5u, a3 = a0 * 7u, a4 = a0 + 9u, a5 = a0 + 11u, a6 = a0 ^ 13u, a7 = a0 + 17u;
    uint32_t b = threadIdx.x * 2654435761u, c = blockIdx.x + 12345u;
    for (uint32_t i = 0; i < n; ++i)
        asm volatile("s_mov_b64 s[22:23], 0x55\n\ts_mov_b32 s24, 5\n\ts_nop 4\n\t.rept 16\n\tv_mul_lo_u32 %0, %0, %8\n\tv_mul_lo_u32 %1, %1, %8\n\tv_mul_lo_u32 %2, %2, %8\n\tv_mul_lo_u32 %3, %3, %8\n\tv_mul_lo_u32 %4, %4, %8\n\tv_mul_lo_u32 %5, %5, %8\n\tv_mul_lo_u32 %6, %6, %8\n\tv_mul_lo_u32 %7, %7, %8\n\t.endr"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                     : "v"(b), "v"(c), "s"(sv) : "vcc", "s20", "s21", "s22", "s23", "s24", "v40", "v41");
    out[blockIdx.x * 1024 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}
int main() {
    uint32_t* out;
    (void)hipMalloc(&out, 256 * 1024 * 4);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    const uint32_t n = 1000;
    float base = 0;
#define RUN(NAME)                                                                                           \
    for (int rep = 0; rep < 2; ++rep) {                                                                     \
        (void)hipEventRecord(e0);                                                                           \
        k_##NAME<<<256, 1024>>>(out, n, 77);                                                                \
        (void)hipEventRecord(e1);                                                                           \
        (void)hipEventSynchronize(e1);                                                                      \
        float ms;                                                                                           \
        (void)hipEventElapsedTime(&ms, e0, e1);                                                             \
        if (rep) { if (base == 0) base = ms; printf("%-12s %.3f ms  %.2f x add_vv\n", #NAME, ms, ms / base); } \
    }
    RUN(add_vv)
    RUN(add_sv)
    RUN(add_iv)
    RUN(add_lv)
    RUN(sub_vv)
    RUN(sub_sv)
    RUN(subrev_sv)
    RUN(and_vv)
    RUN(and_sv)
    RUN(and_lv)
    RUN(or_vv)
    RUN(xor_vv)
    RUN(lshl_iv)
    RUN(lshl_vv)
    RUN(lshl_v_x)
    RUN(lshr_iv)
    RUN(lshr_vv)
    RUN(ashr_iv)
    RUN(mov_v)
    RUN(mov_s)
    RUN(not_v)
    RUN(bfrev)
    RUN(bcnt_v0)
    RUN(bcnt_vv)
    RUN(bfe_u)
    RUN(bfe_i)
    RUN(min_vv)
    RUN(max_sv)
    RUN(mul24)
    RUN(mad24)
    RUN(lshl_or)
    RUN(and_or)
    RUN(or3)
    RUN(add3)
    RUN(add_lshl)
    RUN(lshl_add)
    RUN(xad)
    RUN(bfi)
    RUN(alignbit)
    RUN(perm)
    RUN(mbcnt_lo)
    RUN(cndmask_s)
    RUN(cmp_vcc)
    RUN(cmp_s)
    RUN(cmp_lt_u_vv)
    RUN(addc)
    RUN(add_co)
    RUN(pk_add16)
    RUN(pk_lshl16)
    RUN(pk_sub16)
    RUN(add_dpp)
    RUN(mov_dpp)
    RUN(add_sdwa)
    RUN(readlane)
    RUN(writelane)
    RUN(readfirst)
    RUN(lshl64)
    RUN(cvt)
    RUN(fma)
    RUN(mul_f)
    RUN(mul_lo)
    return 0;
}
