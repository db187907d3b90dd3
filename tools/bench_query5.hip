// bench_query5.hip -- round 4: k_query_u64 (rbf_kernels_u64.h) against k_query_s64 on the synthetic 1080p x 29-frame batch of
// bench_query3.hip: outputs compared (pass bytes and segment counts; uniform floor(k*) = 2 and a mixed batch with floor(k*) 0..5,
// repeated thresholds), then timed: full kernel | no staging | pure passes (no staging, no barrier) | pure without outputs.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o build/bench_query5 tools/bench_query5.hip     Run: build/bench_query5 [reps]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <cstring>
#include "../new_bloom_filter_repo_amd/csrc/rbf_kernels_u64.h"
#include "legacy/rbf_kernels_u64_ab.h"                           // k_query_u64 / u64w with their measurement variants (namespace rbf::ab); AB = 0 runs the LIBRARY's kernels
#include "legacy/rbf_kernels_s64.h"                              // round 3's k_query_s64 / s64w (namespace rbf::legacy): the reference of the comparison
using namespace rbf;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

static const uint32_t *g_image = nullptr;

static FrameTable s64_tab(const FrameTable &tab, uint32_t F, uint32_t *nactive, uint64_t (&empty)[2])       // rbf_api.hip: query_table_s64
{
    FrameTable q; memset(&q, 0, sizeof q); empty[0] = empty[1] = 0;
    uint64_t sorted[MAX_BATCH]; uint32_t coded = 0;
    for (uint32_t f = 0; f < F; ++f) { if (tab.f[f].m) sorted[coded++] = tab.f[f].T; else empty[f >> 6] |= 1ull << (f & 63); }
    std::sort(sorted, sorted + coded);
    uint32_t j = 0;
    for (uint32_t f = 0; f < F; ++f) {
        if (!tab.f[f].m) continue;
        const double ninv = -1.0 / (double)tab.f[f].m;
        q.f[j].m = tab.f[f].m; memcpy(&q.f[j].M, &ninv, 8);
        q.f[j].floor_k = tab.f[f].floor_k | ((uint32_t)(std::lower_bound(sorted, sorted + coded, tab.f[f].T) - sorted) << 8) | (f << 16);
        q.f[j].T = sorted[j]; ++j;
    }
    *nactive = coded;
    return q;
}

enum Kern { S64, U64, U64W, S64W };
template <int AB, Kern K>
static float run(uint64_t n, uint32_t F, const FrameTable &tab, Seeds sd, uint64_t fstride, uint32_t fwmax, uint32_t *seg_cnt, uint64_t nseg, uint64_t *pwords, int R = 20)
{
    uint32_t nactive; uint64_t empty[2]; U64Classes cls{};
    const FrameTable qt = (K == S64 || K == S64W) ? s64_tab(tab, F, &nactive, empty) : query_table_u64(tab, F, &nactive, &cls, empty);
    const size_t lds = 2 * ((size_t)((fwmax + 3) & ~3u) + 4) * 4 + ((K == S64 || K == S64W) ? legacy::S64_GEO_BYTES : u64_geo_bytes(nactive));
    const uint32_t bx = (uint32_t)((nseg + QL_WAVES - 1) / QL_WAVES);
    auto launch = [&]() {
        if constexpr (K == S64) legacy::k_query_s64<AB><<<bx, QL_THREADS, lds, 0>>>(n, nactive, qt, sd, g_image, fstride, fwmax, seg_cnt, nseg, pwords, nullptr, empty[0], empty[1]);
        else if constexpr (K == S64W) legacy::k_query_s64w<AB><<<bx, QL_THREADS, lds, 0>>>(n, nactive, qt, sd, g_image, fstride, fwmax, seg_cnt, nseg, pwords, nullptr, empty[0], empty[1]);
        else if constexpr (K == U64 && AB == 0) k_query_u64<<<bx, QL_THREADS, lds, 0>>>(n, nactive, qt, cls, sd, g_image, fstride, fwmax, seg_cnt, nseg, pwords, nullptr, empty[0], empty[1]);
        else if constexpr (K == U64) ab::k_query_u64<AB><<<bx, QL_THREADS, lds, 0>>>(n, nactive, qt, cls, sd, g_image, fstride, fwmax, seg_cnt, nseg, pwords, nullptr, empty[0], empty[1]);
        else if constexpr (AB == 0) k_query_u64w<<<bx, QL_THREADS, lds, 0>>>(n, nactive, qt, cls, sd, g_image, fstride, fwmax, seg_cnt, nseg, pwords, nullptr, empty[0], empty[1]);
        else ab::k_query_u64w<AB><<<bx, QL_THREADS, lds, 0>>>(n, nactive, qt, cls, sd, g_image, fstride, fwmax, seg_cnt, nseg, pwords, nullptr, empty[0], empty[1]);
    };
    if constexpr (K == S64) CK(hipFuncSetAttribute((const void *)legacy::k_query_s64<AB>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    else if constexpr (K == S64W) CK(hipFuncSetAttribute((const void *)legacy::k_query_s64w<AB>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    else if constexpr (K == U64 && AB == 0) CK(hipFuncSetAttribute((const void *)k_query_u64, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    else if constexpr (K == U64) CK(hipFuncSetAttribute((const void *)ab::k_query_u64<AB>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    else if constexpr (AB == 0) CK(hipFuncSetAttribute((const void *)k_query_u64w, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    else CK(hipFuncSetAttribute((const void *)ab::k_query_u64w<AB>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int w = 0; w < 3; ++w) launch();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int r = 0; r < R; ++r) launch();
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    CK(hipGetLastError());
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    CK(hipEventDestroy(a)); CK(hipEventDestroy(b));
    return ms / R * 1000.f;
}

int main(int argc, char **argv)
{
    const int reps = argc > 1 ? atoi(argv[1]) : 2;
    const uint64_t n = 1920 * 1080; const uint32_t F = 29; const uint32_t m = 611158;
    const uint64_t fwords = (m + 31) / 32, fstride = ((fwords + 3) & ~3ull);
    const uint64_t nseg = (n + QL_SEG_PIXELS - 1) / QL_SEG_PIXELS;
    std::vector<uint32_t> hf(fstride * F);
    srand(1);
    for (auto &x : hf) x = ((uint32_t)rand() << 16) ^ (uint32_t)rand();
    uint32_t *sb, *sc;
    CK(hipMalloc(&sb, (size_t)F * nseg * QL_P * 8)); CK(hipMalloc(&sc, (size_t)F * nseg * 4));
    { std::vector<uint32_t> img(hf.size()); for (size_t i = 0; i < hf.size(); ++i) img[i] = ~__builtin_bswap32(hf[i]);
      uint32_t *di; CK(hipMalloc(&di, img.size() * 4 + 64)); CK(hipMemcpy(di, img.data(), img.size() * 4, hipMemcpyHostToDevice)); g_image = di; }
    FrameTable tab{};
    for (uint32_t f = 0; f < F; ++f) { tab.f[f].m = m - 37 * f; tab.f[f].floor_k = 2; tab.f[f].T = 0x4D00000000000000ull; tab.f[f].M = 0; }
    FrameTable vt = tab;                                          // varied thresholds (distinct and repeated values), floor(k*) 0..5, two frames not coded
    for (uint32_t f = 0; f < F; ++f) { vt.f[f].T = 0x1000000000000000ull * ((f * 7) % 13 + 1) + f % 3; vt.f[f].floor_k = f < 14 ? 2 : (f - 14) % 6; }
    vt.f[5].m = 0; vt.f[23].m = 0;
    FrameTable odd = tab;                                         // 28 coded frames: an even count (the packed pass counts end differently)
    odd.f[28].m = 0;
    Seeds sd{0x12345678, 0x87654321, 999};
    const uint32_t fwmax = (uint32_t)fwords;
    const size_t pwb = (size_t)F * nseg * QL_P * 8, scb = (size_t)F * nseg * 4;
    std::vector<uint8_t> a(pwb), b(pwb); std::vector<uint32_t> ca(F * nseg), cb(F * nseg);
    for (int w = 0; w < 30; ++w) run<0, S64>(n, F, tab, sd, fstride, fwmax, sc, nseg, (uint64_t *)sb, 10);      // warm the clocks up
    uint64_t cmp_nseg = nseg;
    auto compare = [&](const char *what, const FrameTable &t, auto ref, auto got) {
        CK(hipMemset(sb, 0xEE, pwb)); CK(hipMemset(sc, 0xEE, scb));
        ref(t);
        CK(hipMemcpy(a.data(), sb, pwb, hipMemcpyDeviceToHost)); CK(hipMemcpy(ca.data(), sc, scb, hipMemcpyDeviceToHost));
        CK(hipMemset(sb, 0xDD, pwb)); CK(hipMemset(sc, 0xDD, scb));
        got(t);
        CK(hipMemcpy(b.data(), sb, pwb, hipMemcpyDeviceToHost)); CK(hipMemcpy(cb.data(), sc, scb, hipMemcpyDeviceToHost));
        size_t diff = 0; for (size_t i = 0; i < (size_t)F * cmp_nseg * QL_P * 8; ++i) diff += a[i] != b[i];      // (what both kernels own: frames x segments of THIS geometry)
        size_t dc = 0; for (size_t i = 0; i < (size_t)F * cmp_nseg; ++i) dc += ca[i] != cb[i];
        size_t passes = 0; for (size_t i = 0; i < ca.size(); ++i) passes += ca[i] < 0xEEEEEEEEu ? ca[i] : 0;
        printf("%-58s differing pass bytes %zu, differing segment counts %zu (%zu passes)\n", what, diff, dc, passes);
        if (dc) { int shown = 0; for (size_t i = 0; i < ca.size() && shown < 12; ++i) if (ca[i] != cb[i]) { printf("    count index %zu: want %u got %u\n", i, ca[i], cb[i]); ++shown; } }
        if (diff) { int shown = 0; for (size_t i = 0; i < pwb && shown < 12; i += 64) if (memcmp(&a[i], &b[i], 64)) { printf("    pass bytes of count index %zu differ: want %02x %02x.. got %02x %02x..\n", i / 64, a[i], a[i + 1], b[i], b[i + 1]); ++shown; } }
    };
    compare("k_query_u64 vs k_query_s64, floor(k*) = 2", tab, [&](const FrameTable &t) { run<0, S64>(n, F, t, sd, fstride, fwmax, sc, nseg, (uint64_t *)sb, 1); },
            [&](const FrameTable &t) { run<0, U64>(n, F, t, sd, fstride, fwmax, sc, nseg, (uint64_t *)sb, 1); });
    compare("k_query_u64 vs k_query_s64, 28 coded frames", odd, [&](const FrameTable &t) { run<0, S64>(n, F, t, sd, fstride, fwmax, sc, nseg, (uint64_t *)sb, 1); },
            [&](const FrameTable &t) { run<0, U64>(n, F, t, sd, fstride, fwmax, sc, nseg, (uint64_t *)sb, 1); });
    compare("k_query_u64 vs k_query_s64, mixed floor(k*), thresholds", vt, [&](const FrameTable &t) { run<0, S64>(n, F, t, sd, fstride, fwmax, sc, nseg, (uint64_t *)sb, 1); },
            [&](const FrameTable &t) { run<0, U64>(n, F, t, sd, fstride, fwmax, sc, nseg, (uint64_t *)sb, 1); });
    compare("k_query_u64w vs k_query_s64w, mixed", vt, [&](const FrameTable &t) { run<0, S64W>(n, F, t, sd, fstride, fwmax, sc, nseg, (uint64_t *)sb, 1); },
            [&](const FrameTable &t) { run<0, U64W>(n, F, t, sd, fstride, fwmax, sc, nseg, (uint64_t *)sb, 1); });
    {   // a frame size whose last workgroup has partial and dead waves
        const uint64_t n2 = 1920 * 1080 - 3000, nseg2 = (n2 + QL_SEG_PIXELS - 1) / QL_SEG_PIXELS;
        cmp_nseg = nseg2;
        compare("k_query_u64 vs k_query_s64, ragged last segment", vt, [&](const FrameTable &t) { run<0, S64>(n2, F, t, sd, fstride, fwmax, sc, nseg2, (uint64_t *)sb, 1); },
                [&](const FrameTable &t) { run<0, U64>(n2, F, t, sd, fstride, fwmax, sc, nseg2, (uint64_t *)sb, 1); });
        cmp_nseg = nseg;
    }
    {   // smaller filters (the DMA's lane masks: rows shorter than a piece row, and of 2.5 piece rows)
        for (uint32_t msmall : {40000u, 330000u}) {
            FrameTable st = vt;
            for (uint32_t f = 0; f < F; ++f) if (st.f[f].m) st.f[f].m = msmall - 11 * f;
            const uint32_t fw2 = (msmall + 31) / 32; const uint64_t fs2 = (fw2 + 3) & ~3ull;
            char name[96]; snprintf(name, sizeof name, "k_query_u64 vs k_query_s64, m = %u", msmall);
            compare(name, st, [&](const FrameTable &t) { run<0, S64>(n, F, t, sd, fs2, fw2, sc, nseg, (uint64_t *)sb, 1); },
                    [&](const FrameTable &t) { run<0, U64>(n, F, t, sd, fs2, fw2, sc, nseg, (uint64_t *)sb, 1); });
        }
    }
#define ROW(K, name) printf("%-28s full %6.1f (mixed %6.1f) | no staging %6.1f | pure %6.1f | pure, no counts %6.1f | pure, no outputs at all %6.1f us\n", name, \
        run<0, K>(n, F, tab, sd, fstride, fwmax, sc, nseg, (uint64_t *)sb), run<0, K>(n, F, vt, sd, fstride, fwmax, sc, nseg, (uint64_t *)sb), \
        run<8, K>(n, F, tab, sd, fstride, fwmax, sc, nseg, (uint64_t *)sb), run<8 | 32, K>(n, F, tab, sd, fstride, fwmax, sc, nseg, (uint64_t *)sb), \
        run<8 | 32 | 4, K>(n, F, tab, sd, fstride, fwmax, sc, nseg, (uint64_t *)sb), run<8 | 32 | 4 | 64, K>(n, F, tab, sd, fstride, fwmax, sc, nseg, (uint64_t *)sb))
    for (int rep = 0; rep < reps; ++rep) {
        ROW(S64, "k_query_s64 (round 3)");
        ROW(U64, "k_query_u64");
        printf("k_query_u64 without the barrier (wrong results) %6.1f | k_query_u64w full %6.1f (mixed %6.1f) us\n",
               run<32, U64>(n, F, tab, sd, fstride, fwmax, sc, nseg, (uint64_t *)sb), run<0, U64W>(n, F, tab, sd, fstride, fwmax, sc, nseg, (uint64_t *)sb), run<0, U64W>(n, F, vt, sd, fstride, fwmax, sc, nseg, (uint64_t *)sb));
        printf("k_query_u64 without wave priorities: full %6.1f us\n", run<2048, U64>(n, F, tab, sd, fstride, fwmax, sc, nseg, (uint64_t *)sb));
    }
    for (uint32_t ff : {1u, 2u, 8u, 15u, 29u})
        printf("frames %2u: k_query_u64 full %6.1f | pure %6.1f | k_query_s64 full %6.1f | pure %6.1f us\n", ff, run<0, U64>(n, ff, tab, sd, fstride, fwmax, sc, nseg, (uint64_t *)sb),
               run<8 | 32, U64>(n, ff, tab, sd, fstride, fwmax, sc, nseg, (uint64_t *)sb), run<0, S64>(n, ff, tab, sd, fstride, fwmax, sc, nseg, (uint64_t *)sb),
               run<8 | 32, S64>(n, ff, tab, sd, fstride, fwmax, sc, nseg, (uint64_t *)sb));
    return 0;
}
