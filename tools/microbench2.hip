// microbench2.hip — issue-rate calibration on MI355X for the chain kernels' instruction mix:
// cycles per wave64 VALU instruction (a dependent chain of v_add_u32) with 1..4 waves per SIMD, 16-bit vs
// 32-bit LDS stores, random LDS gathers (with the index arithmetic around them).  One workgroup per CU.  Times are per wave-instruction.
// Build: hipcc --offload-arch=gfx950 -O3 tools/microbench2.hip -o tools/microbench2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

using LdsU16 = __attribute__((address_space(3))) uint16_t;
using LdsU32 = __attribute__((address_space(3))) uint32_t;

template <int T>
__global__ void __launch_bounds__(T) k_valu_add(int iters, uint32_t* out) {
    uint32_t a = threadIdx.x, b = a + 1, c = a + 2, d = a + 3;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            asm volatile("v_add_u32 %0, %0, %1\n v_add_u32 %1, %1, %2\n v_add_u32 %2, %2, %3\n v_add_u32 %3, %3, %0"
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
        }
    }
    if (a + b + c + d == 0x12345678u) out[0] = a;
}
// LDS stores: MODE 0 = ds_write_b32 conflict-free, 1 = ds_write_b16 to one half of consecutive dwords
// (half-split layout), 2 = ds_write_b16 to consecutive halfwords (adjacent lanes share a dword)
template <int T, int MODE>
__global__ void __launch_bounds__(T) k_lds_store(int iters, uint32_t* out) {
    extern __shared__ unsigned char smem[];
    const uint32_t base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
    const uint32_t lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
    uint32_t addr = base + w * 4096u + (MODE == 2 ? lane * 2u : lane * 4u);
    uint32_t v = threadIdx.x;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const uint32_t ad = addr + (uint32_t)k * 256u;
            if (MODE == 0) *reinterpret_cast<LdsU32*>((uintptr_t)ad) = v;
            else *reinterpret_cast<LdsU16*>((uintptr_t)ad) = (uint16_t)v;
        }
        v += 1;
    }
    __syncthreads();
    if (smem[threadIdx.x] == 0x77 && iters < 0) out[0] = 1;
}
// random gathers: MODE 0 = ds_read_b32, 1 = ds_read_b64 from a table of `words` dwords
template <int T, int MODE>
__global__ void __launch_bounds__(T) k_lds_gather(int iters, uint32_t words, uint32_t* out) {
    extern __shared__ unsigned char smem[];
    uint32_t* s = reinterpret_cast<uint32_t*>(smem);
    for (uint32_t i = threadIdx.x; i < words; i += T) s[i] = i * 2654435761u;
    __syncthreads();
    uint32_t r[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) r[k] = (threadIdx.x * 7919u + k * 104729u) * 2654435761u;
    uint32_t acc = 0;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (MODE == 0) {
                const uint32_t x = s[(r[k] >> 7) % words];
                acc += x;
                r[k] = r[k] * 1664525u + x;
            } else {
                const uint2 x = *reinterpret_cast<const uint2*>(s + (((r[k] >> 7) % (words / 2u)) * 2u));
                acc += x.x ^ x.y;
                r[k] = r[k] * 1664525u + x.x;
            }
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
    const int iters = 20000, grid = 256;
    const double ghz = 2.4;
#define RUNV(name, kern, TT, per_iter)                                                                    \
    {                                                                                                     \
        float ms = time_it([&] { kern<TT><<<grid, TT>>>(iters, d); });                                   \
        double ns = ms * 1e6 / iters / (per_iter);                                                        \
        printf("%-18s T=%4d  %8.3f ms  %6.2f ns per wave-instr per wave = %5.2f cyc@2.4GHz; per SIMD (x waves/SIMD %d): %5.2f cyc\n", \
               name, TT, ms, ns, ns * ghz, (TT / 64 + 3) / 4, ns * ghz / ((TT / 64 + 3) / 4));           \
    }
    RUNV("valu add x4", k_valu_add, 64, 64) RUNV("valu add x4", k_valu_add, 256, 64) RUNV("valu add x4", k_valu_add, 512, 64) RUNV("valu add x4", k_valu_add, 1024, 64)
#define RUNL(name, kern, TT, MODE, per_iter, lds)                                                         \
    {                                                                                                     \
        hipFuncSetAttribute(reinterpret_cast<const void*>(&kern<TT, MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, lds); \
        float ms = time_it([&] { kern<TT, MODE><<<grid, TT, lds>>>(iters, d); });                        \
        double ns = ms * 1e6 / iters / (per_iter) / (TT / 64);                                            \
        printf("%-26s T=%4d  %8.3f ms  %6.2f ns per wave-instr per CU = %5.2f LDS cyc@2.4GHz\n", name, TT, ms, ns, ns * ghz); \
    }
    RUNL("ds_write_b32", k_lds_store, 1024, 0, 16, 65536) RUNL("ds_write_b16 half-split", k_lds_store, 1024, 1, 16, 65536)
    RUNL("ds_write_b16 adjacent", k_lds_store, 1024, 2, 16, 65536)
    RUNL("ds_write_b32", k_lds_store, 256, 0, 16, 65536) RUNL("ds_write_b16 half-split", k_lds_store, 256, 1, 16, 65536)
#define RUNG(name, TT, MODE, words)                                                                       \
    {                                                                                                     \
        hipFuncSetAttribute(reinterpret_cast<const void*>(&k_lds_gather<TT, MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, (words) * 4); \
        float ms = time_it([&] { k_lds_gather<TT, MODE><<<grid, TT, (words) * 4>>>(iters / 4, words, d); }); \
        double ns = ms * 1e6 / (iters / 4) / 8 / (TT / 64);                                               \
        printf("%-26s T=%4d words %6d %8.3f ms  %6.2f ns per wave-instr per CU = %5.2f cyc@2.4GHz (incl. ~4 VALU)\n", name, TT, words, ms, ns, ns * ghz); \
    }
    RUNG("gather ds_read_b32", 1024, 0, 2048) RUNG("gather ds_read_b32", 1024, 0, 160)
    RUNG("gather ds_read_b64", 1024, 1, 4096) RUNG("gather ds_read_b64", 1024, 1, 320)
    return 0;
}
