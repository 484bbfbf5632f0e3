// probe: do scalar stores work on gfx950, and what do they cost next to v_writelane + vector store?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

// every wave writes LINES rows of 64 words (u64): word e of line j = ballot(hash(lane, e, j) & 1)
__device__ __forceinline__ uint32_t hsh(uint32_t a) { a ^= a >> 16; a *= 0x7feb352du; a ^= a >> 15; a *= 0x846ca68bu; a ^= a >> 16; return a; }

extern "C" __device__ uint32_t wl(uint32_t, uint32_t, uint32_t) __asm("llvm.amdgcn.writelane.i32");
template <int MODE>
__global__ void __launch_bounds__(1024) k(unsigned long long* out, uint32_t lines, uint32_t seed) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t gw = blockIdx.x * 16u + w;
    uint32_t st[64];
#pragma unroll
    for (int e = 0; e < 64; ++e) st[e] = hsh(seed + gw * 4096u + e * 64u + lane);
    for (uint32_t j = 0; j < lines; ++j) {
        unsigned long long* row = out + ((size_t)j * gridDim.x * 16u + gw) * 64u;
        uint32_t lo = 0, hi = 0;
#pragma unroll
        for (int e = 0; e < 64; ++e) {
            st[e] = st[e] * 1664525u + 1013904223u;
            const unsigned long long m = __ballot((int32_t)st[e] < 0);
            if constexpr (MODE == 0) {
                lo = wl((uint32_t)m, e, lo);
                hi = wl((uint32_t)(m >> 32), e, hi);
            } else {
                asm volatile("s_store_dwordx2 %0, %1, %2" ::"s"(m), "s"(row), "n"(e * 8) : "memory");
            }
        }
        if constexpr (MODE == 0) {
            reinterpret_cast<uint2*>(row)[lane] = make_uint2(lo, hi);
        } else {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    }
    if constexpr (MODE == 1) asm volatile("s_dcache_wb" ::: "memory");
}

int main() {
    const uint32_t grid = 256, lines = 512;
    const size_t words = (size_t)lines * grid * 16 * 64;
    unsigned long long *a, *b;
    CK(hipMalloc(&a, words * 8));
    CK(hipMalloc(&b, words * 8));
    CK(hipMemset(a, 0, words * 8));
    CK(hipMemset(b, 0xff, words * 8));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    float ms[2] = {0, 0};
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0));
        k<0><<<grid, 1024>>>(a, lines, 7);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms[0], e0, e1));
        CK(hipEventRecord(e0));
        k<1><<<grid, 1024>>>(b, lines, 7);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms[1], e0, e1));
        printf("rep %d: writelane+vector store %.3f ms, scalar stores %.3f ms\n", rep, ms[0], ms[1]);
    }
    CK(hipDeviceSynchronize());
    std::vector<unsigned long long> ha(words), hb(words);
    CK(hipMemcpy(ha.data(), a, words * 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hb.data(), b, words * 8, hipMemcpyDeviceToHost));
    size_t bad = 0;
    for (size_t i = 0; i < words; ++i) bad += ha[i] != hb[i];
    printf("mismatching words: %zu of %zu\n", bad, words);
    return bad != 0;
}
