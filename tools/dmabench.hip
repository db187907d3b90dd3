// dmabench.hip -- how fast can a workgroup stage a 76 KB filter from L2 into LDS on gfx950?  (profiling aid)
// Variants: LDS-DMA (global_load_lds_dwordx4 / dword, the query kernels' dma_filter) and plain global loads into VGPRs
// followed by ds_write; 16 waves per CU each issuing their share, timed with the shader clock inside the kernel:
// cycles a wave spends ISSUING its loads, and cycles until its data has landed (s_waitcnt vmcnt(0)).
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include "../new_bloom_filter_repo_amd/csrc/rbf_kernels_lds.h"
using namespace rbf;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

// KIND 0: dma16 per piece (M0 saved / restored around every load, as dma_filter does)
// KIND 1: M0 set once per load, no restore (s_mov m0 + load)
// KIND 2: plain global_load_dwordx4 into VGPRs, then ds_write_b128 after the wait
// KIND 3: LDS-DMA of single dwords (global_load_lds_dword), 4x the instructions
template <int KIND>
__global__ __launch_bounds__(1024) void k_stage(const uint32_t *__restrict__ src, uint32_t words, uint64_t *out, int reps, uint32_t active_waves)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6, nwaves = active_waves;
    uint64_t t_issue = 0, t_done = 0;
    uint32_t sink = 0;
    for (int r = 0; r < reps; ++r) {
        __syncthreads();
        const uint64_t t0 = __builtin_readcyclecounter();
        if (wave < nwaves) {
            const uint32_t base = __builtin_amdgcn_readfirstlane(lds_addr_of(lds));
            const uint32_t npieces = words >> 2, nchunks = (npieces + 63u) >> 6;
            if (KIND == 0) {
                for (uint32_t c = wave; c < nchunks; c += nwaves) {
                    const uint32_t piece = (c << 6) + lane;
                    if (piece < npieces) dma16(src + (piece << 2), __builtin_amdgcn_readfirstlane(base + (c << 10)));
                }
            } else if (KIND == 1) {
                for (uint32_t c = wave; c < nchunks; c += nwaves) {
                    const uint32_t piece = (c << 6) + lane;
                    const uint32_t *g = src + (piece << 2);
                    if (piece < npieces) asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(g), "s"(__builtin_amdgcn_readfirstlane(base + (c << 10))) : "memory", "m0");
                }
            } else if (KIND == 2) {
                uint4 v[5];
                uint32_t cnt = 0;
#pragma unroll
                for (int j = 0; j < 5; ++j) {
                    const uint32_t c = wave + j * nwaves;
                    const uint32_t piece = (c << 6) + lane;
                    if (c < nchunks && piece < npieces) v[j] = reinterpret_cast<const uint4 *>(src)[piece];
                    else v[j] = make_uint4(0, 0, 0, 0);
                    ++cnt;
                }
                const uint64_t t1 = __builtin_readcyclecounter();
                t_issue += t1 - t0;
#pragma unroll
                for (int j = 0; j < 5; ++j) {
                    const uint32_t c = wave + j * nwaves;
                    const uint32_t piece = (c << 6) + lane;
                    if (c < nchunks && piece < npieces) reinterpret_cast<uint4 *>(lds)[piece] = v[j];
                }
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                t_done += __builtin_readcyclecounter() - t0;
                sink += cnt;
                continue;
            } else {
                const uint32_t nch = (words + 63u) >> 6;
                for (uint32_t c = wave; c < nch; c += nwaves) {
                    const uint32_t w = (c << 6) + lane;
                    if (w < words) dma4(src + w, __builtin_amdgcn_readfirstlane(base + (c << 8)));
                }
            }
            const uint64_t t1 = __builtin_readcyclecounter();
            dma_wait_all();
            const uint64_t t2 = __builtin_readcyclecounter();
            t_issue += t1 - t0; t_done += t2 - t0;
        }
    }
    __syncthreads();
    if (lane == 0 && wave < nwaves) { out[(blockIdx.x * 16 + wave) * 2] = t_issue / reps; out[(blockIdx.x * 16 + wave) * 2 + 1] = t_done / reps; }
    if (sink == 0x12345 && lds[threadIdx.x] == 77) out[0] = sink;
}

// Do LDS reads (the query kernel's probes) stall while a filter is being staged?  Waves [0, dma_waves) stage `words` dwords
// into the upper half of LDS over and over (MODE 1: LDS-DMA x4; MODE 2: global_load_dwordx4 -> ds_write_b128; MODE 0:
// idle), the other waves run a fixed number of random ds_read_b32 probes of the lower half; reported: cycles the probing
// waves needed.
template <int MODE>
__global__ __launch_bounds__(1024) void k_contend(const uint32_t *__restrict__ src, uint32_t words, uint64_t *out, int probes, uint32_t dma_waves)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    __shared__ uint32_t done;
    if (threadIdx.x == 0) done = 0;
    for (uint32_t i = threadIdx.x; i < 20480; i += 1024) lds[i] = i * 2654435761u;
    __syncthreads();
    uint32_t *stage = lds + 20480;
    if (wave < dma_waves) {
        if (MODE == 0) return;
        const uint32_t base = __builtin_amdgcn_readfirstlane(lds_addr_of(stage));
        const uint32_t npieces = words >> 2, nchunks = (npieces + 63u) >> 6;
        uint32_t rounds = 0;
        while (__hip_atomic_load(&done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < 16u - dma_waves && rounds < 10000u) {
            if (MODE == 1) {
                for (uint32_t c = wave; c < nchunks; c += dma_waves) {
                    const uint32_t piece = (c << 6) + lane;
                    if (piece < npieces) dma16(src + (piece << 2), __builtin_amdgcn_readfirstlane(base + (c << 10)));
                }
                dma_wait_all();
            } else {
                for (uint32_t c = wave; c < nchunks; c += dma_waves) {
                    const uint32_t piece = (c << 6) + lane;
                    if (piece < npieces) reinterpret_cast<uint4 *>(stage)[piece] = reinterpret_cast<const uint4 *>(src)[piece];
                }
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            }
            ++rounds;
        }
        if (lane == 0) out[(blockIdx.x * 16 + wave) * 2] = rounds;
        return;
    }
    uint32_t x = threadIdx.x * 2654435761u + blockIdx.x, acc = 0;
    const uint64_t t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int r = 0; r < probes; ++r) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            x = x * 1664525u + 1013904223u;
            acc += lds[(x >> 12) % 19100u];
        }
    }
    const uint64_t t1 = __builtin_readcyclecounter();
    if (acc == 0x12345u) lds[0] = acc;
    if (lane == 0) { out[(blockIdx.x * 16 + wave) * 2 + 1] = t1 - t0; __hip_atomic_fetch_add(&done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
}

template <int MODE>
static void contend(const char *what, const uint32_t *src, uint32_t words, uint64_t *out, uint32_t dma_waves, int cus)
{
    auto kern = k_contend<MODE>;
    CK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024));
    CK(hipMemset(out, 0, (size_t)cus * 32 * 8));
    hipLaunchKernelGGL(kern, dim3(cus), dim3(1024), 159 * 1024, 0, src, words, out, 400, dma_waves);
    CK(hipDeviceSynchronize());
    std::vector<uint64_t> h((size_t)cus * 32);
    CK(hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost));
    std::vector<uint64_t> t; uint64_t rounds = 0;
    for (int b = 0; b < cus; ++b) for (uint32_t w = 0; w < 16; ++w) { if (w >= dma_waves) t.push_back(h[(b * 16 + w) * 2 + 1]); else rounds += h[(b * 16 + w) * 2]; }
    std::sort(t.begin(), t.end());
    printf("%-52s %u staging waves: probing waves took median %7llu max %7llu cycles for 3200 probes each; %.1f stagings per CU meanwhile\n", what, dma_waves,
           (unsigned long long)t[t.size() / 2], (unsigned long long)t.back(), (double)rounds / cus / (dma_waves ? dma_waves : 1));
}

template <int KIND>
static void run(const char *what, const uint32_t *src, uint32_t words, uint64_t *out, uint32_t active, int cus)
{
    auto kern = k_stage<KIND>;
    CK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    const int reps = 50;
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipLaunchKernelGGL(kern, dim3(cus), dim3(1024), 80 * 1024, 0, src, words, out, 5, active);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    hipLaunchKernelGGL(kern, dim3(cus), dim3(1024), 80 * 1024, 0, src, words, out, reps, active);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    std::vector<uint64_t> h((size_t)cus * 32);
    CK(hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost));
    std::vector<uint64_t> iss, don;
    for (int bk = 0; bk < cus; ++bk) for (uint32_t w = 0; w < active; ++w) { iss.push_back(h[(bk * 16 + w) * 2]); don.push_back(h[(bk * 16 + w) * 2 + 1]); }
    std::sort(iss.begin(), iss.end()); std::sort(don.begin(), don.end());
    printf("%-58s %2u waves: issue median %5llu max %5llu | landed median %5llu max %5llu cycles | %.2f us per staging (wall, %d CUs)\n", what, active,
           (unsigned long long)iss[iss.size() / 2], (unsigned long long)iss.back(), (unsigned long long)don[don.size() / 2], (unsigned long long)don.back(),
           ms * 1000.f / reps, cus);
}

int main()
{
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const uint32_t words = 19100;                                 // a 1080p filter: 76.4 KB
    uint32_t *src; uint64_t *out;
    CK(hipMalloc(&src, words * 4 + 4096)); CK(hipMemset(src, 0x5A, words * 4 + 4096));
    CK(hipMalloc(&out, (size_t)cus * 32 * 8));
    for (uint32_t active : {16u, 8u, 4u, 1u}) {
        run<0>("LDS-DMA x4, M0 saved/restored per load (dma_filter)", src, words, out, active, cus);
        run<1>("LDS-DMA x4, M0 set per load, no restore", src, words, out, active, cus);
        run<3>("LDS-DMA x1 (dword per lane)", src, words, out, active, cus);
        if (active * 5 * 1024 >= words * 4) run<2>("global_load_dwordx4 -> VGPR -> ds_write_b128", src, words, out, active, cus);
    }
    printf("# probes vs staging: 12 waves probe the lower half of LDS while 4 waves stage into the upper half\n");
    contend<0>("nothing staged (4 waves idle)", src, words, out, 4, cus);
    contend<1>("LDS-DMA x4 staging", src, words, out, 4, cus);
    contend<2>("global_load_dwordx4 + ds_write_b128 staging", src, words, out, 4, cus);
    printf("# one CU only (no L2 contention)\n");
    run<0>("LDS-DMA x4, M0 saved/restored per load (dma_filter)", src, words, out, 16, 1);
    run<2>("global_load_dwordx4 -> VGPR -> ds_write_b128", src, words, out, 16, 1);
    return 0;
}
