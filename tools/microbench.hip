// microbench.hip — cost of the primitives the PBWT chain kernel is made of, on one workgroup per
// CU.  Build: hipcc --offload-arch=gfx950 -O3 tools/microbench.hip -o tools/microbench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int T>
__global__ void __launch_bounds__(T) k_barrier(int iters, uint32_t* out) {
    uint32_t acc = threadIdx.x;
    for (int i = 0; i < iters; ++i) {
        __syncthreads();
        acc += i;
    }
    if (acc == 0xFFFFFFFFu) out[0] = acc;
}

template <int T>
__global__ void __launch_bounds__(T) k_lds_chain(int iters, uint32_t* out) {
    __shared__ uint32_t s[4096];
    for (int i = threadIdx.x; i < 4096; i += T) s[i] = (i * 2654435761u) & 4095u;
    __syncthreads();
    uint32_t p = threadIdx.x;
    for (int i = 0; i < iters; ++i) p = s[p];  // dependent LDS read
    if (p == 0xFFFFFFFFu) out[0] = p;
}

template <int T>
__global__ void __launch_bounds__(T) k_lds_atomic(int iters, uint32_t* out) {
    __shared__ uint32_t s[4096];
    for (int i = threadIdx.x; i < 4096; i += T) s[i] = 0;
    __syncthreads();
    uint32_t p = threadIdx.x * 7u;
    for (int i = 0; i < iters; ++i) {
        atomicOr(&s[(p >> 5) & 4095u], 1u << (p & 31u));
        p = p * 1664525u + 1013904223u;
    }
    __syncthreads();
    if (s[threadIdx.x] == 0x12345u) out[0] = 1;
}

template <int T>
__global__ void __launch_bounds__(T) k_barrier_lds(int iters, uint32_t* out) {
    // barrier + one LDS write + one dependent LDS read per iteration (the skeleton of a chain step)
    __shared__ uint32_t s[2048];
    uint32_t p = threadIdx.x;
    for (int i = 0; i < iters; ++i) {
        s[p & 2047u] = p;
        __syncthreads();
        p = s[(p * 13u + i) & 2047u] + 1u;
    }
    if (p == 0xFFFFFFFFu) out[0] = p;
}

__device__ __forceinline__ uint32_t scan_dpp(uint32_t v) {
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, true);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, true);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, true);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, true);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, true);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, true);
    return v;
}
template <int T>
__global__ void __launch_bounds__(T) k_scan(int iters, uint32_t* out) {
    uint32_t v = threadIdx.x;
    for (int i = 0; i < iters; ++i) v = scan_dpp(v & 3u) + i;
    if (v == 0xFFFFFFFFu) out[0] = v;
}

template <int T>
__global__ void __launch_bounds__(T) k_valu(int iters, uint32_t* out) {
    uint32_t v = threadIdx.x, u = blockIdx.x;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            v = v * 3u + u;
            u = u ^ (v >> 3);
        }
    }
    if (v + u == 0xFFFFFFFFu) out[0] = v;
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
    const int iters = 20000, grid = 123;
#define RUN(name, kern, TT, per)                                                                        \
    {                                                                                                   \
        float ms = time_it([&] { kern<TT><<<grid, TT>>>(iters, d); });                                 \
        printf("%-14s T=%4d  %8.3f ms  -> %7.1f ns per %s\n", name, TT, ms, ms * 1e6 / iters, per);    \
    }
    RUN("barrier", k_barrier, 256, "barrier")
    RUN("barrier", k_barrier, 512, "barrier")
    RUN("barrier", k_barrier, 1024, "barrier")
    RUN("lds_chain", k_lds_chain, 64, "dependent read")
    RUN("lds_chain", k_lds_chain, 256, "dependent read")
    RUN("lds_chain", k_lds_chain, 1024, "dependent read")
    RUN("lds_atomic", k_lds_atomic, 256, "atomicOr")
    RUN("lds_atomic", k_lds_atomic, 1024, "atomicOr")
    RUN("barrier+lds", k_barrier_lds, 256, "write+barrier+read")
    RUN("barrier+lds", k_barrier_lds, 512, "write+barrier+read")
    RUN("barrier+lds", k_barrier_lds, 1024, "write+barrier+read")
    RUN("dpp_scan", k_scan, 64, "64-lane scan")
    RUN("dpp_scan", k_scan, 1024, "64-lane scan")
    RUN("valu32", k_valu, 64, "32 dependent-ish VALU")
    RUN("valu32", k_valu, 256, "32 dependent-ish VALU")
    RUN("valu32", k_valu, 1024, "32 dependent-ish VALU")
    return 0;
}
