// microbench3.hip -- how fast can a workgroup stage a 76 KB filter into LDS, 29 times, on all 256 CUs?
//   (a) LDS-DMA (global_load_lds_dwordx4, what k_query_lds uses)   (b) global_load_dwordx4 -> VGPRs -> ds_write_b128
// Both with and without concurrent LDS probe traffic from the same waves.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include "../new_bloom_filter_repo_amd/csrc/rbf_kernels_lds.h"
using namespace rbf;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int MODE /*0 dma, 1 regs, 2 dma issued piecemeal between probe groups*/, bool PROBE>
__global__ __launch_bounds__(1024) void k_stage(const uint32_t *__restrict__ filters, uint64_t stride_words, uint32_t words, uint32_t nframes, uint32_t *sink)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
    const uint32_t bufwords = (words + 3u) & ~3u;
    uint32_t acc = 0, x = threadIdx.x * 2654435761u + blockIdx.x;
    uint32_t cur = 0;
    for (uint32_t f = 0; f < nframes; ++f) {
        const uint32_t *src = filters + (uint64_t)f * stride_words;
        uint32_t *dst = lds + cur * bufwords;
        const uint32_t npieces_d = words >> 2, nchunks_d = (npieces_d + 63u) >> 6;
        const uint32_t base_d = __builtin_amdgcn_readfirstlane(lds_addr_of(dst));
        if (MODE == 0) {
            dma_filter(dst, src, words, wave, lane, nwaves);
        } else if (MODE == 2) {
        } else if (MODE == 3) {
        } else {
            typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
            const uint32_t npieces = words >> 2;
            u32x4 v[5];
#pragma unroll
            for (int c = 0; c < 5; ++c) { const uint32_t piece = ((wave + c * nwaves) << 6) + lane; if (piece < npieces) v[c] = *reinterpret_cast<const u32x4 *>(src + (piece << 2)); }
#pragma unroll
            for (int c = 0; c < 5; ++c) { const uint32_t piece = ((wave + c * nwaves) << 6) + lane; if (piece < npieces) *reinterpret_cast<u32x4 *>(dst + (piece << 2)) = v[c]; }
        }
        typedef uint32_t u32x4b __attribute__((ext_vector_type(4)));
        u32x4b r3[5];
        if (MODE == 3) {
#pragma unroll
            for (int c = 0; c < 5; ++c) { const uint32_t piece = ((wave + c * nwaves) << 6) + lane; if (piece < npieces_d) r3[c] = __builtin_nontemporal_load(reinterpret_cast<const u32x4b *>(src + (piece << 2))); }
        }
        if (PROBE) {                                             // 24 random LDS reads per lane per frame from the OTHER buffer
            const uint32_t *probe = lds + (cur ^ 1u) * bufwords;
#pragma unroll
            for (int i = 0; i < 24; ++i) {
                if (MODE == 2 && (i % 4) == 0 && i / 4 < 5) {            // one DMA chunk per wave before every 4th probe
                    const uint32_t c = wave + (i / 4) * nwaves;
                    const uint32_t piece = (c << 6) + lane;
                    if (c < nchunks_d && piece < npieces_d) dma16(src + (piece << 2), __builtin_amdgcn_readfirstlane(base_d + (c << 10)));
                }
                x = x * 1664525u + 1013904223u; acc += probe[(x >> 8) % words];
            }
        }
        if (MODE == 3) {
#pragma unroll
            for (int c = 0; c < 5; ++c) { const uint32_t piece = ((wave + c * nwaves) << 6) + lane; if (piece < npieces_d) *reinterpret_cast<u32x4b *>(dst + (piece << 2)) = r3[c]; }
        }
        if (MODE == 0 || MODE == 2) dma_wait_all();
        __syncthreads();
        cur ^= 1u;
    }
    if (acc == 0x12345678u) sink[0] = acc + lds[threadIdx.x];
}

template <int MODE, bool PROBE>
static float run(const uint32_t *filters, uint64_t stride, uint32_t words, uint32_t F, uint32_t *sink)
{
    auto kern = k_stage<MODE, PROBE>;
    const size_t lds = (size_t)2 * ((words + 3) & ~3u) * 4;
    CK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(kern, dim3(254), dim3(1024), lds, 0, filters, stride, words, F, sink);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int r = 0; r < 10; ++r) hipLaunchKernelGGL(kern, dim3(254), dim3(1024), lds, 0, filters, stride, words, F, sink);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    return ms / 10 * 1000.f;
}

int main()
{
    const uint32_t F = 29, m = 611158, words = (m + 31) / 32;
    const uint64_t stride = (words + 3) & ~3ull;
    uint32_t *df, *sink;
    CK(hipMalloc(&df, stride * F * 4)); CK(hipMemset(df, 0x5A, stride * F * 4)); CK(hipMalloc(&sink, 64));
    printf("staging 29 x %.1f KB per workgroup, 254 workgroups of 1024 threads\n", words * 4 / 1024.0);
    printf("LDS-DMA, no probes                 %7.1f us\n", run<0, false>(df, stride, words, F, sink));
    printf("global_load + ds_write, no probes  %7.1f us\n", run<1, false>(df, stride, words, F, sink));
    printf("LDS-DMA + 24 LDS probes/lane/frame %7.1f us\n", run<0, true>(df, stride, words, F, sink));
    printf("load+ds_write + 24 probes          %7.1f us\n", run<1, true>(df, stride, words, F, sink));
    printf("loads before / ds_write after probes %5.1f us\n", run<3, true>(df, stride, words, F, sink));
    printf("DMA piecemeal between probe groups %7.1f us\n", run<2, true>(df, stride, words, F, sink));
    return 0;
}
