// bench_insert2.hip -- round 4: k_insert_w32 (rbf_kernels_w32.h) against k_insert_tab on a synthetic 1080p x 29-frame batch: the OR over
// the slices of every frame's partial filters must be identical (random masks at p = 0.089, clustered masks, a sparse frame, a ragged
// frame size, the HASHED variants), then both are timed, with ablations.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o build/bench_insert2 tools/bench_insert2.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cstring>
#include "legacy/rbf_kernels_w32.h"                               // the experiment (namespace rbf), on top of the library's rbf_kernels_i64.h
#include "legacy/rbf_kernels_i64.h"                               // round 3's k_insert_tab with its ablation bits (namespace rbf::legacy): the reference
using namespace rbf;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

struct Setup { const uint8_t *masks; uint64_t mstride, n; uint32_t F, S; FrameTable itab; const uint4 *table; Seeds sd; uint32_t *partials; uint64_t pstride; uint32_t tile_words; };

template <int KIND, int IAB, bool HASHED, bool SINGLE = true>
static float run(const Setup &u, int R = 10)
{
    SliceTable sl{}; uint32_t per_tile = 0;
    for (uint32_t f = 0; f < u.F; ++f) { sl.n[f] = (uint8_t)u.S; per_tile += u.S; }
    const size_t lds = KIND == 0 ? (size_t)u.tile_words * 4 + (size_t)IL_WAVES * legacy::IT_QUEUE * 4 : w32_lds_bytes(u.tile_words);
    auto launch = [&]() {
        if constexpr (KIND == 0) hipLaunchKernelGGL((legacy::k_insert_tab<IAB, HASHED>), dim3(per_tile), dim3(IL_THREADS), lds, 0, u.masks, u.mstride, u.n, u.itab, u.table, u.sd, u.partials, u.pstride, u.tile_words, sl, per_tile, u.S);
        else hipLaunchKernelGGL((k_insert_w32<IAB, HASHED, SINGLE>), dim3(per_tile), dim3(IL_THREADS), lds, 0, u.masks, u.mstride, u.n, u.itab, u.table, u.sd, u.partials, u.pstride, u.tile_words, sl, per_tile, u.S);
    };
    if constexpr (KIND == 0) CK(hipFuncSetAttribute((const void *)legacy::k_insert_tab<IAB, HASHED>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    else CK(hipFuncSetAttribute((const void *)k_insert_w32<IAB, HASHED, SINGLE>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int w = 0; w < 2; ++w) launch();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int r = 0; r < R; ++r) launch();
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    CK(hipGetLastError());
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    return ms / R * 1000.f;
}

static void fill_masks(std::vector<uint8_t> &hm, uint64_t mstride, uint64_t n, uint32_t F, int kind)
{
    srand(7 + kind);
    std::fill(hm.begin(), hm.end(), 0);
    for (uint32_t f = 0; f < F; ++f) {
        uint8_t *row = hm.data() + f * mstride;
        if (kind == 0) { for (uint64_t i = 0; i < n; ++i) if (rand() % 1000 < 89) row[i >> 3] |= 0x80u >> (i & 7); }
        else if (kind == 1) {                       // clustered: runs of changed pixels (moving objects), same density
            uint64_t i = 0;
            while (i < n) { const uint64_t gap = rand() % 9000, run = rand() % 900; i += gap; for (uint64_t j = 0; j < run && i < n; ++j, ++i) row[i >> 3] |= 0x80u >> (i & 7); }
        } else { for (uint64_t i = 0; i < n; ++i) if (rand() % 1000 < (f == 3 ? 2 : 89)) row[i >> 3] |= 0x80u >> (i & 7); }      // frame 3 nearly static
    }
}

int main()
{
    const uint32_t F = 29, S = 8, m = 611158;
    const uint64_t nmax = 1920 * 1080;
    const uint64_t mstride = ((nmax + 63) / 64) * 8, fwords = (m + 31) / 32, pstride = (fwords + 3) & ~3ull;
    std::vector<uint8_t> hm(mstride * F);
    uint8_t *dm; uint32_t *dp; uint4 *dt;
    CK(hipMalloc(&dm, hm.size())); CK(hipMalloc(&dp, (size_t)F * S * pstride * 4)); CK(hipMalloc(&dt, (nmax + 512) * 32));
    Seeds sd{0x12345678, 0x87654321, 999};
    const uint32_t segs = (uint32_t)((nmax + QL_SEG_PIXELS - 1) / QL_SEG_PIXELS);
    hipLaunchKernelGGL(k_hash_table, dim3((segs + 3) / 4), dim3(HT_THREADS), 0, 0, nmax, sd, dt);
    CK(hipDeviceSynchronize());
    Setup u{dm, mstride, nmax, F, S, {}, dt, sd, dp, pstride, (uint32_t)pstride};
    for (uint32_t f = 0; f < F; ++f) { u.itab.f[f].m = m - 37 * f; u.itab.f[f].floor_k = f % 5 == 4 ? 3 : 2; u.itab.f[f].T = 0x4D00000000000000ull; const double ninv = -1.0 / (double)u.itab.f[f].m; memcpy(&u.itab.f[f].M, &ninv, 8); }
    const size_t pw = (size_t)F * S * pstride;
    std::vector<uint32_t> ref(pw), got(pw);
    auto compare = [&](const char *what, auto a, auto b) {
        CK(hipMemset(dp, 0, pw * 4)); a(); CK(hipMemcpy(ref.data(), dp, pw * 4, hipMemcpyDeviceToHost));
        CK(hipMemset(dp, 0, pw * 4)); b(); CK(hipMemcpy(got.data(), dp, pw * 4, hipMemcpyDeviceToHost));
        size_t diff = 0; uint64_t bits = 0;
        for (uint32_t f = 0; f < F; ++f) for (uint32_t w = 0; w < (u.itab.f[f].m + 31) / 32; ++w) {
            uint32_t x = 0, y = 0;
            for (uint32_t sl = 0; sl < S; ++sl) { const size_t i = ((size_t)f * S + sl) * pstride + w; x |= ref[i]; y |= got[i]; }
            diff += x != y; bits += __builtin_popcount(x); }
        printf("%-64s differing filter words %zu (%llu bits set)\n", what, diff, (unsigned long long)bits);
    };
    for (int kind = 0; kind < 3; ++kind) {
        fill_masks(hm, mstride, nmax, F, kind);
        CK(hipMemcpy(dm, hm.data(), hm.size(), hipMemcpyHostToDevice));
        const char *kn[3] = {"random p = 0.089", "clustered runs", "one nearly static frame"};
        char name[128];
        snprintf(name, sizeof name, "k_insert_w32 vs k_insert_tab, %s", kn[kind]);
        compare(name, [&]() { run<0, 0, false>(u, 1); }, [&]() { run<1, 0, false>(u, 1); });
        if (kind == 0) compare("k_insert_w32 (general tile test) vs k_insert_tab", [&]() { run<0, 0, false>(u, 1); }, [&]() { run<1, 0, false, false>(u, 1); });
        if (kind == 0) compare("k_insert_w32<HASHED> vs k_insert_tab", [&]() { run<0, 0, false>(u, 1); }, [&]() { run<1, 0, true>(u, 1); });
        if (kind == 0) { Setup r = u; r.n = nmax - 3001; compare("k_insert_w32 vs k_insert_tab, ragged frame size", [&]() { run<0, 0, false>(r, 1); }, [&]() { run<1, 0, false>(r, 1); }); }
        for (int rep = 0; rep < 2; ++rep)
            printf("  %-24s k_insert_tab %6.1f us | k_insert_w32 %6.1f us | w32 no gather %6.1f | w32 no atomics %6.1f | w32 no zero/store %6.1f | w32 hashed %6.1f | tab hashed %6.1f us\n", kn[kind],
                   run<0, 0, false>(u), run<1, 0, false>(u), run<1, 1, false>(u), run<1, 2, false>(u), run<1, 4, false>(u), run<1, 0, true>(u), run<0, 0, true>(u));
    }
    return 0;
}
