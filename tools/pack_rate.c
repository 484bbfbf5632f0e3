/* pack_rate.c — rate of the writer's pack-on-append alone (xsi_debug_pack_bit_row over int32 rows in DRAM), one thread.
 * usage: pack_rate <n_haps> <n_lines>   Build: gcc -O2 -I include tools/pack_rate.c -L xsqueezeit_amd -lxsi_hip */
#define _POSIX_C_SOURCE 199309L
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include "xsi_hip.h"
static double now_s(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }
int main(int argc, char** argv) {
    const uint32_t n = argc > 1 ? (uint32_t)atoi(argv[1]) : 5008;
    const uint64_t lines = argc > 2 ? (uint64_t)atoll(argv[2]) : 100000;
    int32_t* rows = (int32_t*)malloc(lines * n * 4);
    uint8_t* out = (uint8_t*)malloc(n / 8 + 64);
    for (uint64_t i = 0; i < lines * n; ++i) rows[i] = 2 | ((i * 2654435761u >> 13) & 2) | (i & 1);
    for (int rep = 0; rep < 2; ++rep) {
        double t0 = now_s();
        int ok = 1;
        for (uint64_t l = 0; l < lines; ++l) ok &= xsi_debug_pack_bit_row(rows + l * n, n, 1, out);
        double t = now_s() - t0;
        /* plain read of the same bytes */
        double t1 = now_s();
        uint64_t acc = 0;
        for (uint64_t i = 0; i < lines * n; i += 2) acc += *(const uint64_t*)(rows + i);
        double tr = now_s() - t1;
        printf("haps=%u lines=%llu pack %.2f G cells/s (%.1f GB/s, ok=%d)  plain 8-byte read loop %.1f GB/s (%llu)\n", n,
               (unsigned long long)lines, lines * (double)n / t * 1e-9, lines * (double)n * 4 / t * 1e-9, ok,
               lines * (double)n * 4 / tr * 1e-9, (unsigned long long)(acc & 1));
    }
    return 0;
}
