// microbench.hip -- gfx950 rate probes that size the Bloom kernels' design (run via gpurun).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o /tmp/mb tools/microbench.hip && /tmp/mb
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>
#include "../new_bloom_filter_repo_amd/csrc/rbf_device.h"
using namespace rbf;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__device__ __forceinline__ uint64_t lcg(uint64_t x) { return x * 6364136223846793005ULL + 1442695040888963407ULL; }
__device__ __forceinline__ uint32_t mix32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }

__global__ void k_hash3(uint64_t n, uint64_t *sink)
{
    uint64_t acc = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        DecKey k = make_key((uint32_t)i);
        acc ^= xxh64_key(k, 0x12345678) ^ xxh64_key(k, 0x87654321) ^ xxh64_key(k, 999);
    }
    if (acc == 0x1234) sink[0] = acc;
}

__global__ void k_key_only(uint64_t n, uint64_t *sink)
{
    uint64_t acc = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        DecKey k = make_key((uint32_t)i);
        acc ^= k.lo + k.hi + k.len;
    }
    if (acc == 0x1234) sink[0] = acc;
}

template <int REP>
__global__ void k_mul64(uint64_t *sink, uint64_t seed)
{
    uint64_t a = seed + threadIdx.x, b = seed * 3 + blockIdx.x, c = a ^ b, d = a + b;
#pragma unroll 1
    for (int r = 0; r < REP; ++r) {
        a = a * P1 + 1; b = b * P2 + 1; c = c * P3 + 1; d = d * P5 + 1;
        a = a * P2 + 1; b = b * P3 + 1; c = c * P5 + 1; d = d * P1 + 1;
    }
    if ((a ^ b ^ c ^ d) == 0x1234) sink[0] = a;
}

template <int REP>
__global__ void k_mulhi32(uint32_t *sink, uint32_t seed)
{
    uint32_t a = seed + threadIdx.x, b = seed * 3 + blockIdx.x, c = a ^ b, d = a + b;
#pragma unroll 1
    for (int r = 0; r < REP; ++r) {
        a = __umulhi(a, 0x9E3779B1u) + 1; b = __umulhi(b, 0x85EBCA77u) + 1; c = __umulhi(c, 0xC2B2AE3Du) + 1; d = __umulhi(d, 0x27D4EB2Fu) + 1;
        a = a * 0x9E3779B1u + 1; b = b * 0x85EBCA77u + 1; c = c * 0xC2B2AE3Du + 1; d = d * 0x27D4EB2Fu + 1;
    }
    if ((a ^ b ^ c ^ d) == 0x1234) sink[0] = a;
}

template <int REP>
__global__ void k_mod2(uint64_t *sink, uint32_t m, uint64_t M, uint64_t seed)
{
    uint64_t h = seed + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t acc = 0;
#pragma unroll 1
    for (int r = 0; r < REP; ++r) {
        h = lcg(h);
        acc += mod_m(h, m, M);
        acc += mod_m(h ^ 0x5555555555555555ULL, m + 7, M);
    }
    if (acc == 0x1234567) sink[0] = acc;
}

template <int REP>
__global__ void k_gload(const uint32_t *buf, uint32_t words, uint32_t *sink)
{
    uint32_t x = (blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u + 12345u, acc = 0;
#pragma unroll 4
    for (int r = 0; r < REP; ++r) {
        x = mix32(x + r);
        acc += buf[x % words];
    }
    if (acc == 0x1234567) sink[0] = acc;
}

template <int REP>
__global__ void k_lds(uint32_t *sink, uint32_t words)
{
    extern __shared__ uint32_t lds[];
    for (uint32_t i = threadIdx.x; i < words; i += blockDim.x) lds[i] = i * 7;
    __syncthreads();
    uint32_t x = (blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u + 12345u, acc = 0;
#pragma unroll 4
    for (int r = 0; r < REP; ++r) {
        x = mix32(x + r);
        acc += lds[x % words];
    }
    if (acc == 0x1234567) sink[0] = acc;
}

template <int REP>
__global__ void k_lds_atomic(uint32_t *sink, uint32_t words)
{
    extern __shared__ uint32_t lds[];
    for (uint32_t i = threadIdx.x; i < words; i += blockDim.x) lds[i] = 0;
    __syncthreads();
    uint32_t x = (blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u + 12345u;
#pragma unroll 4
    for (int r = 0; r < REP; ++r) {
        x = mix32(x + r);
        atomicOr(&lds[x % words], 1u << (x >> 27));
    }
    __syncthreads();
    if (lds[threadIdx.x % words] == 0x1234567) sink[0] = 1;
}

template <int REP>
__global__ void k_atomic(uint32_t *buf, uint32_t words)
{
    uint32_t x = (blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u + 12345u;
#pragma unroll 4
    for (int r = 0; r < REP; ++r) {
        x = mix32(x + r);
        atomicOr(&buf[x % words], 1u << (x >> 27));
    }
}

template <typename F>
static float time_ms(F launch, int reps = 5)
{
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    launch();
    hipDeviceSynchronize();
    hipEventRecord(a);
    for (int r = 0; r < reps; ++r) launch();
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    return ms / reps;
}

int main()
{
    uint64_t *sink; CK(hipMalloc(&sink, 64));
    uint32_t *buf; const size_t BUF = 256u << 20; CK(hipMalloc(&buf, BUF)); CK(hipMemset(buf, 0x11, BUF));
    const int G = 256 * 8, B = 256;           // 2048 blocks of 256 threads
    const double threads = (double)G * B;

    { uint64_t n = 1ull << 26; float ms = time_ms([&] { hipLaunchKernelGGL(k_hash3, dim3(G), dim3(B), 0, 0, n, sink); });
      printf("hash3 (3 x XXH64 of str(i), i<2^26): %.3f ms -> %.1f Gkeys/s\n", ms, n / ms / 1e6); }
    { uint64_t n = 1ull << 26; float ms = time_ms([&] { hipLaunchKernelGGL(k_key_only, dim3(G), dim3(B), 0, 0, n, sink); });
      printf("decimal key only: %.3f ms -> %.1f Gkeys/s\n", ms, n / ms / 1e6); }
    { const int REP = 512; float ms = time_ms([&] { hipLaunchKernelGGL(k_mul64<REP>, dim3(G), dim3(B), 0, 0, sink, 77ull); });
      double ops = threads * REP * 8; printf("mul64 (a*C+1): %.3f ms -> %.1f Gmul64/s (%.2f cyc/wave-instr @2.4GHz/SIMD)\n", ms, ops / ms / 1e6,
             1024.0 * 2.4e9 / (ops / 64 / (ms * 1e-3))); }
    { const int REP = 512; float ms = time_ms([&] { hipLaunchKernelGGL(k_mulhi32<REP>, dim3(G), dim3(B), 0, 0, (uint32_t *)sink, 77u); });
      double ops = threads * REP * 8; printf("mulhi32/mullo32 mix: %.3f ms -> %.1f Gop/s (%.2f cyc/wave-instr)\n", ms, ops / ms / 1e6,
             1024.0 * 2.4e9 / (ops / 64 / (ms * 1e-3))); }
    { const int REP = 256; uint32_t m = 611158; uint64_t M = (uint64_t)((((unsigned __int128)1) << 64) / m);
      float ms = time_ms([&] { hipLaunchKernelGGL(k_mod2<REP>, dim3(G), dim3(B), 0, 0, sink, m, M, 99ull); });
      double ops = threads * REP * 2; printf("mod_m (exact Barrett): %.3f ms -> %.1f Gmod/s (%.1f cyc/wave-mod)\n", ms, ops / ms / 1e6,
             1024.0 * 2.4e9 / (ops / 64 / (ms * 1e-3))); }
    for (uint32_t kb : {76u, 2216u, 65536u, 262144u}) {
        const int REP = 64; uint32_t words = kb * 256;
        float ms = time_ms([&] { hipLaunchKernelGGL(k_gload<REP>, dim3(G), dim3(B), 0, 0, buf, words, (uint32_t *)sink); });
        double ops = threads * REP; printf("random global dword loads, %u KB region: %.3f ms -> %.1f Gloads/s\n", kb, ms, ops / ms / 1e6);
    }
    for (uint32_t kb : {76u, 2216u, 65536u}) {
        const int REP = 64; uint32_t words = kb * 256;
        float ms = time_ms([&] { hipLaunchKernelGGL(k_atomic<REP>, dim3(G), dim3(B), 0, 0, buf, words); });
        double ops = threads * REP; printf("random global atomicOr (no return), %u KB region: %.3f ms -> %.1f Gatomics/s\n", kb, ms, ops / ms / 1e6);
    }
    { const int REP = 256; uint32_t words = 76 * 256; CK(hipFuncSetAttribute((const void *)k_lds<REP>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      float ms = time_ms([&] { hipLaunchKernelGGL(k_lds<REP>, dim3(512), dim3(512), words * 4, 0, (uint32_t *)sink, words); });
      double ops = 512.0 * 512 * REP; printf("random LDS dword reads (76 KB, 2 WG/CU x 512 thr): %.3f ms -> %.1f Gloads/s\n", ms, ops / ms / 1e6); }
    { const int REP = 256; uint32_t words = 76 * 256; CK(hipFuncSetAttribute((const void *)k_lds_atomic<REP>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      float ms = time_ms([&] { hipLaunchKernelGGL(k_lds_atomic<REP>, dim3(512), dim3(512), words * 4, 0, (uint32_t *)sink, words); });
      double ops = 512.0 * 512 * REP; printf("random LDS atomicOr (76 KB): %.3f ms -> %.1f Gatomics/s\n", ms, ops / ms / 1e6); }
    return 0;
}
