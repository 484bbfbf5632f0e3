// microbench4.hip — what the exchange of k_chain_rank_enc_multi can be built from on MI355X:
//  (1) which XCD a workgroup lands on (HW_REG_XCC_ID against blockIdx.x & 7);
//  (2) rate of non-returning global atomic ORs into a 64 KiB region that 8 workgroups share, executed in the
//      XCD's L2 (no scope bits) or at agent scope (sc1), with 4 / 9 / 16 / 64 active lanes per instruction;
//  (3) a 64 KiB region written by one workgroup and read by the 7 others: plain stores + sc1 loads (same-XCD L2)
//      against sc1 stores + sc1 loads (what agent scope costs).
// Build: hipcc --offload-arch=gfx950 -O3 tools/microbench4.hip -o tools/microbench4
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void k_xcc(uint32_t* out) {
    uint32_t id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
    if (threadIdx.x == 0) out[blockIdx.x] = id;
}

// SC1: 0 = L2 atomic (no scope bits), 1 = sc1
template <int SC1>
__global__ void __launch_bounds__(1024) k_atom(int iters, uint32_t active, uint32_t* regions) {
    const uint32_t xcd = blockIdx.x & 7u, q = blockIdx.x >> 3, group = xcd * 4u + q / 8u;
    uint32_t* reg = regions + (size_t)group * 16384u;  // 64 KiB per group
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t h = (blockIdx.x * 1024u + threadIdx.x + 1u) * 2654435761u;
    // `active` lanes of every wave, spread over the wave
    const bool on = (lane * active) / 64u != ((lane + 1u) * active) / 64u;
    if (!on) return;
    for (int i = 0; i < iters; ++i) {
        h = h * 1664525u + 1013904223u;
        const uint32_t off = ((h >> 8) & 16383u) * 4u;
        const uint32_t bit = 1u << (h & 31u);
        if (SC1)
            asm volatile("global_atomic_or %0, %1, %2 sc1" ::"v"(off), "v"(bit), "s"(reg) : "memory");
        else
            asm volatile("global_atomic_or %0, %1, %2" ::"v"(off), "v"(bit), "s"(reg) : "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// member 0 of every group writes 64 KiB per iteration (16 B per thread x 4), the others read it; no ordering
// between them (bandwidth only).  MODE 0: plain stores, sc1 loads; 1: sc1 stores, sc1 loads
typedef uint32_t v4u __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ void __launch_bounds__(1024) k_share(int iters, v4u* regions, uint32_t* out) {
    const uint32_t xcd = blockIdx.x & 7u, q = blockIdx.x >> 3, group = xcd * 4u + q / 8u, member = q % 8u;
    v4u* reg = regions + (size_t)group * 8192u;  // 128 KiB per group
    v4u acc = {0, 0, 0, 0};
    for (int i = 0; i < iters; ++i) {
        if (member == 0) {
            v4u v = {(uint32_t)i, threadIdx.x, 0, 0};
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                v4u* p = reg + k * 1024 + threadIdx.x;
                if (MODE == 0)
                    asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(p), "v"(v) : "memory");
                else
                    asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            v4u x[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const v4u* p = reg + k * 1024 + threadIdx.x;
                asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(x[k]) : "v"(p) : "memory");
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int k = 0; k < 8; ++k) acc += x[k];
        }
    }
    if (acc[0] == 0x12345678u) out[0] = acc[1];
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
    CK(hipMalloc(&d, 32ull * 131072ull));
    CK(hipMemset(d, 0, 32ull * 131072ull));
    uint32_t* dx;
    CK(hipMalloc(&dx, 4096));
    k_xcc<<<256, 64>>>(dx);
    std::vector<uint32_t> hx(256);
    CK(hipMemcpy(hx.data(), dx, 1024, hipMemcpyDeviceToHost));
    int same = 0;
    for (int i = 0; i < 256; ++i) same += (hx[i] & 15u) == (uint32_t)(i & 7);
    printf("XCC_ID == blockIdx.x & 7 for %d of 256 workgroups (first 16 ids:", same);
    for (int i = 0; i < 16; ++i) printf(" %u", hx[i] & 15u);
    printf(")\n");
    const int iters = 2000;
    for (uint32_t active : {4u, 9u, 16u, 64u}) {
        float a = time_it([&] { k_atom<0><<<256, 1024>>>(iters, active, d); });
        float b = time_it([&] { k_atom<1><<<256, 1024>>>(iters, active, d); });
        const double n = 256.0 * 16 * active * iters;
        printf("global_atomic_or, %2u lanes per instruction: L2 scope %7.3f ms = %6.1f G atomics/s (%.1f ns per wave-instr per CU); sc1 %7.3f ms = %6.1f G/s\n",
               active, a, n / a * 1e-6, a * 1e6 / iters / 16, b, n / b * 1e-6);
    }
    {
        const int it = 200;
        float a = time_it([&] { k_share<0><<<256, 1024>>>(it, reinterpret_cast<v4u*>(d), dx); });
        float b = time_it([&] { k_share<1><<<256, 1024>>>(it, reinterpret_cast<v4u*>(d), dx); });
        printf("128 KiB written by one workgroup, read by 7: plain stores + sc1 loads %7.3f ms = %5.2f us per round; sc1 stores + sc1 loads %7.3f ms = %5.2f us per round\n",
               a, a * 1e3 / it, b, b * 1e3 / it);
    }
    return 0;
}
