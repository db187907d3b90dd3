// bench_insert.hip -- ablation timing of k_insert_lds on a synthetic 1080p x 29-frame batch (profiling aid).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../new_bloom_filter_repo_amd/csrc/rbf_kernels_lds.h"
using namespace rbf;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int IAB>
static float run(const uint8_t *masks, uint64_t mstride, uint64_t n, uint32_t F, uint32_t S, const FrameTable &tab, Seeds sd, uint32_t *partials,
                 uint64_t pstride, uint32_t tile_words, size_t lds, uint32_t s_extra = 0)
{
    auto kern = k_insert_lds<true, IAB>;
    CK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    SliceTable sl{};                         // frames [0, s_extra) get S slices, the others S - 1 (s_extra == 0: all get S)
    uint32_t per_tile = 0;
    for (uint32_t f = 0; f < F; ++f) { sl.n[f] = (uint8_t)((s_extra == 0 || f < s_extra) ? S : S - 1); per_tile += sl.n[f]; }
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(kern, dim3(per_tile), dim3(IL_THREADS), lds, 0, masks, mstride, n, tab, sd, partials, pstride, tile_words, sl, per_tile, S);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    const int R = 10;
    for (int r = 0; r < R; ++r) hipLaunchKernelGGL(kern, dim3(per_tile), dim3(IL_THREADS), lds, 0, masks, mstride, n, tab, sd, partials, pstride, tile_words, sl, per_tile, S);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    return ms / R * 1000.f;
}

int main()
{
    const uint64_t n = 1920 * 1080; const uint32_t F = getenv("F") ? (uint32_t)atoi(getenv("F")) : 29, S = 8; const uint32_t m = 611158;
    const uint64_t mstride = ((n + 63) / 64) * 8, fwords = (m + 31) / 32, pstride = (fwords + 3) & ~3ull;
    std::vector<uint8_t> hm(mstride * F);
    srand(1);
    for (auto &x : hm) { uint8_t v = 0; for (int b = 0; b < 8; ++b) if (rand() % 1000 < 89) v |= 1u << b; x = v; }
    uint8_t *dm; uint32_t *dp;
    CK(hipMalloc(&dm, hm.size())); CK(hipMalloc(&dp, (size_t)F * S * pstride * 4));
    CK(hipMemcpy(dm, hm.data(), hm.size(), hipMemcpyHostToDevice));
    FrameTable tab{};
    for (uint32_t f = 0; f < F; ++f) { tab.f[f].m = m - 37 * f; tab.f[f].floor_k = 2; tab.f[f].T = 0x4D00000000000000ull; tab.f[f].M = (uint64_t)((((unsigned __int128)1) << 64) / tab.f[f].m); }
    Seeds sd{0x12345678, 0x87654321, 999};
    const uint32_t tile_words = (uint32_t)pstride;
    const size_t lds = (size_t)tile_words * 4 + (size_t)IL_WAVES * IL_QUEUE * 4;
    if (getenv("SWEEP")) {          // workgroup-count experiment: slices per frame (S, frames with the extra slice)
        CK(hipFree(dp)); CK(hipMalloc(&dp, (size_t)F * 12 * pstride * 4));
        const uint32_t cfg[][2] = {{8, 0}, {8, 0}, {9, 24}, {9, 0}, {9, 10}, {7, 0}, {6, 0}, {4, 0}, {10, 0}, {12, 0}, {8, 0}};
        for (auto &c : cfg)
            printf("S=%u s_extra=%u (workgroups %u): %8.1f us\n", c[0], c[1], c[1] ? c[1] * c[0] + (F - c[1]) * (c[0] - 1) : F * c[0],
                   run<0>(dm, mstride, n, F, c[0], tab, sd, dp, pstride, tile_words, lds, c[1]));
        return 0;
    }
#define RUN(AB, what) printf("%-44s %8.1f us\n", what, run<AB>(dm, mstride, n, F, S, tab, sd, dp, pstride, tile_words, lds));
    RUN(0, "warm-up (ignore)");
    RUN(0, "full kernel");
    RUN(1, "no hashing");
    RUN(2, "no LDS atomics");
    RUN(4, "no LDS zeroing / partial store");
    RUN(8, "no queueing (bytes read only)");
    RUN(1 | 2, "no hashing, no atomics");
    RUN(1 | 2 | 4, "no hashing/atomics/zero/store");
    RUN(0, "full kernel (again)");
    return 0;
}
