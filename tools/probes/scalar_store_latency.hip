// probe: latency of one scalar store (issue -> lgkmcnt 0), single wave and loaded
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void __launch_bounds__(1024) k(unsigned long long* out, unsigned long long* clk, int per_wait) {
    const uint32_t w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    unsigned long long* row = out + ((size_t)blockIdx.x * 16u + w) * 64u;
    unsigned long long m = 0x1234567ull + w;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < 256; ++i) {
        for (int k2 = 0; k2 < per_wait; ++k2)
            asm volatile("s_store_dwordx2 %0, %1, 0x0" ::"s"(m), "s"(row + (k2 & 63)));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        m += 3;
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    asm volatile("s_dcache_wb" ::: "memory");
    if ((threadIdx.x & 63) == 0) clk[blockIdx.x * 16 + w] = t1 - t0;
}
int main() {
    unsigned long long *a, *c;
    hipMalloc(&a, 256 * 16 * 64 * 8);
    hipMalloc(&c, 256 * 16 * 8);
    unsigned long long h[256 * 16];
    for (int per_wait : {1, 8}) {
        for (int blocks : {1, 256}) {
            for (int threads : {64, 1024}) {
                k<<<blocks, threads>>>(a, c, per_wait);
                hipDeviceSynchronize();
                hipMemcpy(h, c, sizeof(h), hipMemcpyDeviceToHost);
                printf("stores per wait %d, %3d workgroups x %4d threads: %.0f shader clocks per wait (wave 0 of wg 0)\n", per_wait, blocks, threads, h[0] / 256.0);
            }
        }
    }
    return 0;
}
