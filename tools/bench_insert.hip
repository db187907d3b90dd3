// bench_insert.hip -- ablation timing of k_insert_lds on a synthetic 1080p x 29-frame batch (profiling aid).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cstring>
#include "legacy/rbf_kernels_i64.h"      // round-3 snapshot (namespace rbf::legacy)
using namespace rbf;
using namespace rbf::legacy;
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
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(kern, dim3(per_tile), dim3(IL_THREADS), lds, 0, masks, mstride, n, itab, table, Seeds{}, partials, pstride, tile_words, sl, per_tile, S);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    const int R = 10;
    for (int r = 0; r < R; ++r) hipLaunchKernelGGL(kern, dim3(per_tile), dim3(IL_THREADS), lds, 0, masks, mstride, n, itab, table, Seeds{}, partials, pstride, tile_words, sl, per_tile, S);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    return ms / R * 1000.f;
}

// BIG=1: the 2160p geometry (8 coded frames, 306 KB filters): k_insert_tab over 3 LDS tiles against k_insert_tab_g (L2 atomics)
static int big()
{
    const uint64_t n = 3840ull * 2160; const uint32_t F = 8, m = 2444632;
    const uint64_t mstride = ((n + 63) / 64) * 8, fwords = (m + 31) / 32, pstride = (fwords + 3) & ~3ull;
    std::vector<uint8_t> hm(mstride * F);
    srand(1);
    for (auto &x : hm) { uint8_t v = 0; for (int b = 0; b < 8; ++b) if (rand() % 1000 < 89) v |= 1u << b; x = v; }
    uint8_t *dm; uint32_t *dp, *dg; uint4 *dt;
    const uint32_t S = 10, tiles = 4;
    CK(hipMalloc(&dm, hm.size())); CK(hipMalloc(&dp, (size_t)F * S * pstride * 4)); CK(hipMalloc(&dg, (size_t)F * pstride * 4)); CK(hipMalloc(&dt, (n + 8192) * 32));
    CK(hipMemcpy(dm, hm.data(), hm.size(), hipMemcpyHostToDevice));
    FrameTable tab{};
    for (uint32_t f = 0; f < F; ++f) { tab.f[f].m = m - 37 * f; tab.f[f].floor_k = 2; tab.f[f].T = 0x4D00000000000000ull; const double ninv = -1.0 / (double)tab.f[f].m; memcpy(&tab.f[f].M, &ninv, 8); }
    Seeds sd{0x12345678, 0x87654321, 999};
    const uint32_t segs = (uint32_t)((n + QL_SEG_PIXELS - 1) / QL_SEG_PIXELS);
    hipLaunchKernelGGL(k_hash_table, dim3((segs + 3) / 4), dim3(HT_THREADS), 0, 0, n, sd, dt);
    CK(hipDeviceSynchronize());
    printf("masks %p..%p partials %p..%p filters %p..%p table %p..%p\n", dm, dm + hm.size(), dp, dp + (size_t)F * S * pstride, dg, dg + (size_t)F * pstride, dt, dt + 2 * (n + 8192)); fflush(stdout);
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    float ms;
    const size_t pw = (size_t)F * S * pstride;
    std::vector<uint32_t> ref(pw), got((size_t)F * pstride);
    {
        const uint32_t tile_words = (uint32_t)(((fwords + tiles - 1) / tiles + 3) & ~3ull);
        const size_t lds = (size_t)tile_words * 4 + (size_t)IL_WAVES * IT_QUEUE * 4;
        auto kern = k_insert_tab<0>;
        CK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        SliceTable sl{}; uint32_t per_tile = 0;
        for (uint32_t f = 0; f < F; ++f) { sl.n[f] = (uint8_t)S; per_tile += S; }
        CK(hipMemset(dp, 0, pw * 4));
        for (int r = 0; r < 12; ++r) {
            if (r == 2) CK(hipEventRecord(a));
            hipLaunchKernelGGL(kern, dim3(per_tile * tiles), dim3(IL_THREADS), lds, 0, dm, mstride, n, tab, dt, Seeds{}, dp, pstride, tile_words, sl, per_tile, S);
        }
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&ms, a, b));
        CK(hipGetLastError());
        printf("%-52s %8.1f us\n", "2160p x 8: k_insert_tab, 4 LDS tiles, 10 slices", ms * 100.f); fflush(stdout);
        CK(hipMemcpy(ref.data(), dp, pw * 4, hipMemcpyDeviceToHost));
    }
    // two kernels: one walk into position records, then tiles of all of LDS filled from the records
    uint2 *dr; uint32_t *dc;
    std::vector<uint64_t> ones(F, 0);
    for (uint32_t f = 0; f < F; ++f) for (uint64_t i = 0; i < mstride; ++i) ones[f] += __builtin_popcount(hm[f * mstride + i]);     // n is a multiple of 64: no pad bits
    uint64_t total = 0;
    FrameTable itab = tab, rtab = tab;
    for (uint32_t f = 0; f < F; ++f) { itab.f[f].floor_k = (uint32_t)total; rtab.f[f].T = total; total += ones[f]; }
    CK(hipMalloc(&dr, total * 8)); CK(hipMalloc(&dc, F * 4));
    const uint32_t tiles2 = 2, S2 = 16;
    const uint32_t tw2 = (uint32_t)(((fwords + tiles2 - 1) / tiles2 + 3) & ~3ull);
    CK(hipFuncSetAttribute((const void *)k_insert_records, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    SliceTable sl2{}; uint32_t per_tile2 = 0;
    for (uint32_t f = 0; f < F; ++f) { sl2.n[f] = (uint8_t)S2; per_tile2 += S2; }
    uint32_t *dp2; CK(hipMalloc(&dp2, (size_t)F * S2 * pstride * 4)); CK(hipMemset(dp2, 0, (size_t)F * S2 * pstride * 4));
    const uint32_t S1s[] = {64, 128, 248, 504};
    for (uint32_t S1 : S1s) {
        float t1 = 0, t2 = 0;
        hipEvent_t c; CK(hipEventCreate(&c));
        for (int r = 0; r < 12; ++r) {
            CK(hipMemsetAsync(dc, 0, F * 4, 0));
            CK(hipEventRecord(a));
            hipLaunchKernelGGL(k_insert_positions<0>, dim3(S1, F), dim3(IP_THREADS), 0, 0, dm, mstride, n, itab, dt, sd, dr, dc);
            CK(hipEventRecord(c));
            hipLaunchKernelGGL(k_insert_records, dim3(per_tile2 * tiles2), dim3(IL_THREADS), (size_t)tw2 * 4, 0, dr, dc, rtab, dp2, pstride, tw2, sl2, per_tile2, S2);
            CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
            float x, y; CK(hipEventElapsedTime(&x, a, c)); CK(hipEventElapsedTime(&y, c, b));
            if (r >= 2) { t1 += x; t2 += y; }
        }
        CK(hipGetLastError());
        std::vector<uint32_t> got2((size_t)F * S2 * pstride);
        CK(hipMemcpy(got2.data(), dp2, got2.size() * 4, hipMemcpyDeviceToHost));
        size_t diff = 0;
        for (uint32_t f = 0; f < F; ++f) for (uint32_t w = 0; w < (tab.f[f].m + 31) / 32; ++w) {
            uint32_t x = 0, y = 0;
            for (uint32_t sl = 0; sl < S; ++sl) x |= ref[((size_t)f * S + sl) * pstride + w];
            for (uint32_t sl = 0; sl < S2; ++sl) y |= got2[((size_t)f * S2 + sl) * pstride + w];
            diff += x != y; }
        printf("2160p x 8: k_insert_positions (%3u slices) %6.1f us + k_insert_records (2 tiles x 16 slices) %6.1f us   (%zu words differ from the tiled kernel's)\n",
               S1, t1 * 100.f, t2 * 100.f, diff);
    }
    for (uint32_t S1 : {64u, 128u, 256u, 504u, 1008u}) {   // hashing the set positions in the kernel instead of gathering their table entries
        float t = 0;
        for (int r = 0; r < 12; ++r) {
            CK(hipMemsetAsync(dc, 0, F * 4, 0));
            CK(hipEventRecord(a));
            hipLaunchKernelGGL((k_insert_positions<0, true>), dim3(S1, F), dim3(IP_THREADS), 0, 0, dm, mstride, n, itab, dt, sd, dr, dc);
            CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
            float x; CK(hipEventElapsedTime(&x, a, b));
            if (r >= 2) t += x;
        }
        hipLaunchKernelGGL(k_insert_records, dim3(per_tile2 * tiles2), dim3(IL_THREADS), (size_t)tw2 * 4, 0, dr, dc, rtab, dp2, pstride, tw2, sl2, per_tile2, S2);
        std::vector<uint32_t> got2((size_t)F * S2 * pstride);
        CK(hipMemcpy(got2.data(), dp2, got2.size() * 4, hipMemcpyDeviceToHost));
        size_t diff = 0;
        for (uint32_t f = 0; f < F; ++f) for (uint32_t w = 0; w < (tab.f[f].m + 31) / 32; ++w) {
            uint32_t x = 0, y = 0;
            for (uint32_t sl = 0; sl < S; ++sl) x |= ref[((size_t)f * S + sl) * pstride + w];
            for (uint32_t sl = 0; sl < S2; ++sl) y |= got2[((size_t)f * S2 + sl) * pstride + w];
            diff += x != y; }
        printf("2160p x 8: k_insert_positions<HASHED> (%4u slices) %6.1f us   (%zu words differ from the tiled kernel's)\n", S1, t * 100.f, diff);
    }
    for (int ab : {1, 8}) {                                       // what the walk costs without the table gather / without the queueing
        float t = 0;
        for (int r = 0; r < 12; ++r) {
            CK(hipMemsetAsync(dc, 0, F * 4, 0));
            CK(hipEventRecord(a));
            if (ab == 1) hipLaunchKernelGGL(k_insert_positions<1>, dim3(128, F), dim3(IP_THREADS), 0, 0, dm, mstride, n, itab, dt, sd, dr, dc);
            else hipLaunchKernelGGL(k_insert_positions<8>, dim3(128, F), dim3(IP_THREADS), 0, 0, dm, mstride, n, itab, dt, sd, dr, dc);
            CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
            float x; CK(hipEventElapsedTime(&x, a, b));
            if (r >= 2) t += x;
        }
        printf("2160p x 8: k_insert_positions (128 slices), ablation %d (1 = no table gather, 8 = mask bytes read only) %6.1f us\n", ab, t * 100.f);
    }
    return 0;
}

int main()
{
    if (getenv("BIG")) return big();
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
