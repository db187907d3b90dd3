// bench_insert.hip -- ablation timing of k_insert_lds on a synthetic 1080p x 29-frame batch (profiling aid).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cstring>
#include "../new_bloom_filter_repo_amd/csrc/rbf_kernels_i64.h"
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

template <int IAB>
static float run_tab(const uint8_t *masks, uint64_t mstride, uint64_t n, uint32_t F, uint32_t S, const FrameTable &tab, const uint4 *table, uint32_t *partials,
                     uint64_t pstride, uint32_t tile_words, size_t lds)
{
    auto kern = k_insert_tab<IAB>;
    CK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    FrameTable itab = tab;
    for (uint32_t f = 0; f < F; ++f) { const double ninv = -1.0 / (double)itab.f[f].m; memcpy(&itab.f[f].M, &ninv, 8); }
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    SliceTable sl{};
    uint32_t per_tile = 0;
    for (uint32_t f = 0; f < F; ++f) { sl.n[f] = (uint8_t)S; per_tile += S; }
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(kern, dim3(per_tile), dim3(IL_THREADS), lds, 0, masks, mstride, n, itab, table, partials, pstride, tile_words, sl, per_tile, S);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    const int R = 10;
    for (int r = 0; r < R; ++r) hipLaunchKernelGGL(kern, dim3(per_tile), dim3(IL_THREADS), lds, 0, masks, mstride, n, itab, table, partials, pstride, tile_words, sl, per_tile, S);
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
    const size_t lds = (size_t)tile_words * 4 + (size_t)IL_WAVES * IT_QUEUE * 4;
    if (getenv("SWEEP")) {          // workgroup-count experiment: slices per frame (S, frames with the extra slice)
        CK(hipFree(dp)); CK(hipMalloc(&dp, (size_t)F * 12 * pstride * 4));
        const uint32_t cfg[][2] = {{8, 0}, {8, 0}, {9, 24}, {9, 0}, {9, 10}, {7, 0}, {6, 0}, {4, 0}, {10, 0}, {12, 0}, {8, 0}};
        for (auto &c : cfg)
            printf("S=%u s_extra=%u (workgroups %u): %8.1f us\n", c[0], c[1], c[1] ? c[1] * c[0] + (F - c[1]) * (c[0] - 1) : F * c[0],
                   run<0>(dm, mstride, n, F, c[0], tab, sd, dp, pstride, tile_words, lds, c[1]));
        return 0;
    }
    {   // table-driven insert: hash table build + gather insert; partial filters must equal k_insert_lds's
        uint4 *dt; CK(hipMalloc(&dt, (n + 512) * 32));
        const uint32_t segs = (uint32_t)((n + QL_SEG_PIXELS - 1) / QL_SEG_PIXELS);
        hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
        hipLaunchKernelGGL(k_hash_table, dim3((segs + 3) / 4), dim3(HT_THREADS), 0, 0, n, sd, dt);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(a));
        for (int r = 0; r < 10; ++r) hipLaunchKernelGGL(k_hash_table, dim3((segs + 3) / 4), dim3(HT_THREADS), 0, 0, n, sd, dt);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        printf("%-44s %8.1f us\n", "k_hash_table (66 MB)", ms * 100.f);
        const size_t pw = (size_t)F * S * pstride;
        std::vector<uint32_t> ref(pw), got(pw);
        CK(hipMemset(dp, 0, pw * 4));
        run<0>(dm, mstride, n, F, S, tab, sd, dp, pstride, tile_words, lds);
        CK(hipMemcpy(ref.data(), dp, pw * 4, hipMemcpyDeviceToHost));
        CK(hipMemset(dp, 0, pw * 4));
        printf("%-44s %8.1f us\n", "[table] k_insert_tab full", run_tab<0>(dm, mstride, n, F, S, tab, dt, dp, pstride, tile_words, lds));
        CK(hipMemcpy(got.data(), dp, pw * 4, hipMemcpyDeviceToHost));
        size_t diff = 0; uint64_t bitsset = 0;                  // the kernels cut a frame into slices differently: compare the OR over the slices
        for (uint32_t f = 0; f < F; ++f) for (uint32_t w = 0; w < (tab.f[f].m + 31) / 32; ++w) {
            uint32_t a = 0, b = 0;
            for (uint32_t sl = 0; sl < S; ++sl) { const size_t i = ((size_t)f * S + sl) * pstride + w; a |= ref[i]; b |= got[i]; }
            diff += a != b; bitsset += __builtin_popcount(a); }
        printf("k_insert_tab vs k_insert_lds filters (OR of the slices): %zu differing words (%llu bits set)\n", diff, (unsigned long long)bitsset);
        printf("%-44s %8.1f us\n", "[table] no gather (fake entries)", run_tab<1>(dm, mstride, n, F, S, tab, dt, dp, pstride, tile_words, lds));
        printf("%-44s %8.1f us\n", "[table] no LDS atomics", run_tab<2>(dm, mstride, n, F, S, tab, dt, dp, pstride, tile_words, lds));
        printf("%-44s %8.1f us\n", "[table] no zeroing / partial store", run_tab<4>(dm, mstride, n, F, S, tab, dt, dp, pstride, tile_words, lds));
        printf("%-44s %8.1f us\n", "[table] no queueing (bytes read only)", run_tab<8>(dm, mstride, n, F, S, tab, dt, dp, pstride, tile_words, lds));
        printf("%-44s %8.1f us\n", "[table] no gather, no atomics", run_tab<1 | 2>(dm, mstride, n, F, S, tab, dt, dp, pstride, tile_words, lds));
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
