// microbench3.hip — LDS pipe THROUGHPUT on MI355X for the access patterns of the element-major chains:
// independent (not address-dependent) gathers from a 16 KiB table, 16 in flight per wave, 16 waves per CU,
// one workgroup per CU.  Random = a fixed pseudo-random address per (thread, slot); linear = lane-consecutive.
// Reported: LDS cycles per wave-instruction per CU at the nominal 2.4 GHz.
// Build: hipcc --offload-arch=gfx950 -O3 tools/microbench3.hip -o tools/microbench3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

// MODE 0 b32 random, 1 b64 random, 2 b32 linear, 3 b64 linear, 4 b128 random, 5 ds_or_b32 random (no return),
// 6 b64 random with the low/high halves of the wave on disjoint halves of the banks (address bit 7 = lane bit 5),
// 7 u16 random
template <int MODE>
__global__ void __launch_bounds__(1024) k_lds(int iters, uint32_t table_bytes, uint32_t* out) {
    extern __shared__ unsigned char smem[];
    uint32_t* s = reinterpret_cast<uint32_t*>(smem);
    for (uint32_t i = threadIdx.x; i < table_bytes / 4u; i += 1024u) s[i] = i * 2654435761u;
    __syncthreads();
    const uint32_t base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t ad[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        uint32_t h = (threadIdx.x * 16u + (uint32_t)k + 1u) * 2654435761u;
        h ^= h >> 15;
        h *= 2246822519u;
        h ^= h >> 13;
        uint32_t a;
        if (MODE == 2) a = (lane * 4u + (uint32_t)k * 256u) % table_bytes;
        else if (MODE == 3) a = (lane * 8u + (uint32_t)k * 512u) % table_bytes;
        else if (MODE == 4) a = (h % (table_bytes / 16u)) * 16u;
        else if (MODE == 1) a = (h % (table_bytes / 8u)) * 8u;
        else if (MODE == 6) a = ((h % (table_bytes / 8u)) * 8u & ~128u) | ((lane >> 5) << 7);
        else if (MODE == 7) a = (h % (table_bytes / 2u)) * 2u;
        else a = (h % (table_bytes / 4u)) * 4u;
        ad[k] = base + a;
    }
    uint32_t acc = 0;
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0 || MODE == 2) {
            uint32_t x[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) asm volatile("ds_read_b32 %0, %1" : "=v"(x[k]) : "v"(ad[k]));
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int k = 0; k < 16; ++k) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(acc) : "v"(x[k]));
        } else if (MODE == 7) {
            uint32_t x[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) asm volatile("ds_read_u16 %0, %1" : "=v"(x[k]) : "v"(ad[k]));
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int k = 0; k < 16; ++k) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(acc) : "v"(x[k]));
        } else if (MODE == 1 || MODE == 3 || MODE == 6) {
            uint64_t x[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) asm volatile("ds_read_b64 %0, %1" : "=v"(x[k]) : "v"(ad[k]));
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int k = 0; k < 16; ++k) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(acc) : "v"((uint32_t)x[k]));
        } else if (MODE == 4) {
            typedef uint32_t u4 __attribute__((ext_vector_type(4)));
            u4 x[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) asm volatile("ds_read_b128 %0, %1" : "=v"(x[k]) : "v"(ad[k]));
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int k = 0; k < 8; ++k) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(acc) : "v"(x[k][0]));
#pragma unroll
            for (int k = 0; k < 8; ++k) asm volatile("ds_read_b128 %0, %1" : "=v"(x[k]) : "v"(ad[k + 8]));
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int k = 0; k < 8; ++k) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(acc) : "v"(x[k][0]));
        } else {
#pragma unroll
            for (int k = 0; k < 16; ++k) asm volatile("ds_or_b32 %0, %1" ::"v"(ad[k]), "v"(lane) : "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    }
    if (acc == 0x12345678u) out[0] = acc;
}

template <typename F>
float time_it(F f) {
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    f();
    hipDeviceSynchronize();
    hipEventRecord(a);
    f();
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    return ms;
}

int main() {
    uint32_t* d;
    CK(hipMalloc(&d, 1024));
    const int iters = 4000, grid = 256;
    const double ghz = 2.4;
#define RUN(name, MODE, bytes)                                                                                         \
    {                                                                                                                  \
        hipFuncSetAttribute(reinterpret_cast<const void*>(&k_lds<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes); \
        float ms = time_it([&] { k_lds<MODE><<<grid, 1024, bytes>>>(iters, bytes, d); });                             \
        double ns = ms * 1e6 / iters / 16 / 16;                                                                        \
        printf("%-44s table %6d B %8.3f ms  %6.2f ns = %5.2f LDS cyc per wave-instr per CU\n", name, bytes, ms, ns, ns * ghz); \
    }
    RUN("ds_read_b32 linear", 2, 16384)
    RUN("ds_read_b64 linear", 3, 16384)
    RUN("ds_read_b32 random", 0, 16384)
    RUN("ds_read_u16 random", 7, 16384)
    RUN("ds_read_b64 random", 1, 16384)
    RUN("ds_read_b64 random, half-waves on own banks", 6, 16384)
    RUN("ds_read_b128 random", 4, 16384)
    RUN("ds_or_b32 random (no return)", 5, 16384)
    RUN("ds_read_b32 random", 0, 65536)
    RUN("ds_read_b64 random", 1, 65536)
    return 0;
}
