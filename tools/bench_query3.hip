// bench_query3.hip -- round 3: k_query_s64 (rbf_kernels_s64.h) against k_query_r64 and k_query_f64 on the synthetic 1080p x 29-frame
// batch of bench_query.hip: outputs compared (pass bytes and segment counts), then timed in four modes -- full kernel | no staging
// (barrier kept) | pure passes (no staging, no barrier) | pure passes without pass counting -- with the frame geometry read from LDS
// (default) and by scalar loads from the kernel-argument segment in every frame (AB & 256: what k_query_r64 does), plus ablations.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o build/bench_query3 tools/bench_query3.hip     Run: build/bench_query3 [reps]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <cstring>
#include "legacy/rbf_kernels_s64.h"      // round-3 snapshot (namespace rbf::legacy): these kernels left the library in round 4
using namespace rbf;
using namespace rbf::legacy;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

static const uint32_t *g_image = nullptr;

// FrameTable as k_query_r64 reads it: M = bits of -1/m, T = sorted thresholds, floor_k |= c << 8 (s64: | frame << 16)
static FrameTable rank_tab(const FrameTable &tab, uint32_t F, bool s64)
{
    FrameTable q = tab;
    std::vector<uint64_t> ts;
    for (uint32_t f = 0; f < F; ++f) ts.push_back(tab.f[f].T);
    std::sort(ts.begin(), ts.end());
    for (uint32_t f = 0; f < F; ++f) {
        const double ninv = -1.0 / (double)q.f[f].m; memcpy(&q.f[f].M, &ninv, 8);
        uint32_t c = 0; for (uint64_t t : ts) c += t < tab.f[f].T;
        q.f[f].floor_k = tab.f[f].floor_k | (c << 8) | (s64 ? f << 16 : 0u);
        q.f[f].T = ts[f];
    }
    return q;
}

enum Kern { F64, R64, S64 };
template <int AB, Kern K = S64, int THREADS = QL_THREADS>
static float run(uint64_t n, uint32_t F, const FrameTable &tab, Seeds sd, uint64_t fstride, uint32_t fwmax, uint32_t *seg_cnt, uint64_t nseg, uint64_t *pwords, int R = 20)
{
    const size_t lds = 2 * ((size_t)((fwmax + 3) & ~3u) + 4) * 4 + (K == S64 ? S64_GEO_BYTES : 0);
    FrameTable qtab = tab;
    if (K == F64) for (uint32_t f = 0; f < F; ++f) { const double ninv = -1.0 / (double)qtab.f[f].m; memcpy(&qtab.f[f].M, &ninv, 8); }
    else qtab = rank_tab(tab, F, K == S64);
    const uint32_t bx = (uint32_t)((nseg + THREADS / 64 - 1) / (THREADS / 64));
    auto launch = [&]() {
        if constexpr (K == F64) k_query_f64<AB><<<bx, THREADS, lds, 0>>>(n, F, qtab, sd, g_image, fstride, fwmax, seg_cnt, nseg, pwords, nullptr);
        else if constexpr (K == R64) k_query_r64<AB><<<bx, THREADS, lds, 0>>>(n, F, qtab, sd, g_image, fstride, fwmax, seg_cnt, nseg, pwords, nullptr, 0u);
        else k_query_s64<AB><<<bx, THREADS, lds, 0>>>(n, F, qtab, sd, g_image, fstride, fwmax, seg_cnt, nseg, pwords, nullptr, 0ull, 0ull);
    };
    if constexpr (K == F64) CK(hipFuncSetAttribute((const void *)k_query_f64<AB>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    else if constexpr (K == R64) CK(hipFuncSetAttribute((const void *)k_query_r64<AB>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    else CK(hipFuncSetAttribute((const void *)k_query_s64<AB>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int w = 0; w < 3; ++w) launch();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int r = 0; r < R; ++r) launch();
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
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
    FrameTable vt = tab;                                          // varied thresholds (distinct and repeated values) and floor(k*) 1..4 and 0, 5
    for (uint32_t f = 0; f < F; ++f) { vt.f[f].T = 0x1000000000000000ull * ((f * 7) % 13 + 1) + f % 3; vt.f[f].floor_k = f < 20 ? 2 : (f - 20) % 6; }
    Seeds sd{0x12345678, 0x87654321, 999};
    const uint32_t fwmax = (uint32_t)fwords;
    const size_t pwb = (size_t)F * nseg * QL_P * 8, scb = (size_t)F * nseg * 4;
    std::vector<uint8_t> a(pwb), b(pwb); std::vector<uint32_t> ca(F * nseg), cb(F * nseg);
    // warm the clocks up: the first timings of a fresh box read ~10 % high (r03a)
    for (int w = 0; w < 30; ++w) run<0, F64>(n, F, vt, sd, fstride, fwmax, sc, nseg, (uint64_t *)sb, 10);
    const float tref = run<0, F64>(n, F, vt, sd, fstride, fwmax, sc, nseg, (uint64_t *)sb);
    CK(hipMemcpy(a.data(), sb, pwb, hipMemcpyDeviceToHost)); CK(hipMemcpy(ca.data(), sc, scb, hipMemcpyDeviceToHost));
    printf("k_query_f64 (reference of the comparison, varied thresholds and floor(k*)) %6.1f us\n", tref);
#define V3(K, V, name) do { \
        CK(hipMemset(sb, 0xEE, pwb)); CK(hipMemset(sc, 0xEE, scb)); \
        const float tv = run<(V), K>(n, F, vt, sd, fstride, fwmax, sc, nseg, (uint64_t *)sb); \
        CK(hipMemcpy(b.data(), sb, pwb, hipMemcpyDeviceToHost)); CK(hipMemcpy(cb.data(), sc, scb, hipMemcpyDeviceToHost)); \
        size_t diff = 0; for (size_t i = 0; i < pwb; ++i) diff += a[i] != b[i]; \
        size_t dc = 0; for (size_t i = 0; i < ca.size(); ++i) dc += ca[i] != cb[i]; \
        const float tf = run<(V), K>(n, F, tab, sd, fstride, fwmax, sc, nseg, (uint64_t *)sb); \
        const float ts = run<(V) | 8, K>(n, F, tab, sd, fstride, fwmax, sc, nseg, (uint64_t *)sb); \
        const float tp = run<(V) | 8 | 32, K>(n, F, tab, sd, fstride, fwmax, sc, nseg, (uint64_t *)sb); \
        const float tn = run<(V) | 8 | 32 | 4, K>(n, F, tab, sd, fstride, fwmax, sc, nseg, (uint64_t *)sb); \
        printf("%-52s full %6.1f (varied %6.1f) | no staging %6.1f | pure %6.1f | pure, no counts %6.1f us | diff vs f64: %zu bytes, %zu counts\n", name, tf, tv, ts, tp, tn, diff, dc); \
        fflush(stdout); } while (0)
    for (int rep = 0; rep < reps; ++rep) {
        V3(R64, 0, "k_query_r64 (round 2)");
        V3(S64, 256, "k_query_s64, geometry by scalar loads per frame");
        V3(S64, 2048, "k_query_s64 without wave priorities");
        V3(S64, 0, "k_query_s64 (geometry from LDS)");
        V3(S64, 8192, "k_query_s64, staggered staging (2 pairs ahead)");
    }
#define A3(K, V, name) printf("%-40s pure %6.1f | no reductions %6.1f | no LDS reads %6.1f | no counts %6.1f | none of the three %6.1f | no hashing %6.1f us\n", name, \
        run<(V) | 8 | 32, K>(n, F, tab, sd, fstride, fwmax, sc, nseg, (uint64_t *)sb), run<(V) | 8 | 32 | 1, K>(n, F, tab, sd, fstride, fwmax, sc, nseg, (uint64_t *)sb), \
        run<(V) | 8 | 32 | 2, K>(n, F, tab, sd, fstride, fwmax, sc, nseg, (uint64_t *)sb), run<(V) | 8 | 32 | 4, K>(n, F, tab, sd, fstride, fwmax, sc, nseg, (uint64_t *)sb), \
        run<(V) | 8 | 32 | 1 | 2 | 4, K>(n, F, tab, sd, fstride, fwmax, sc, nseg, (uint64_t *)sb), run<(V) | 8 | 32 | 16, K>(n, F, tab, sd, fstride, fwmax, sc, nseg, (uint64_t *)sb))
    A3(R64, 0, "k_query_r64");
    A3(S64, 256, "k_query_s64, scalar-load geometry");
    A3(S64, 0, "k_query_s64");
    printf("k_query_s64 with ONE store per launch instead of one per frame (wrong results): full %6.1f | pure %6.1f | pure, none of the three %6.1f us\n",
           run<128>(n, F, tab, sd, fstride, fwmax, sc, nseg, (uint64_t *)sb), run<128 | 8 | 32>(n, F, tab, sd, fstride, fwmax, sc, nseg, (uint64_t *)sb),
           run<128 | 8 | 32 | 1 | 2 | 4>(n, F, tab, sd, fstride, fwmax, sc, nseg, (uint64_t *)sb));
    for (uint32_t ff : {1u, 2u, 8u, 15u, 29u})
        printf("frames %2u: k_query_s64 full %6.1f | pure %6.1f | k_query_r64 full %6.1f us\n", ff, run<0>(n, ff, tab, sd, fstride, fwmax, sc, nseg, (uint64_t *)sb),
               run<8 | 32>(n, ff, tab, sd, fstride, fwmax, sc, nseg, (uint64_t *)sb), run<0, R64>(n, ff, tab, sd, fstride, fwmax, sc, nseg, (uint64_t *)sb));
    {   // timeline of the frame loop: shader-clock stamps of wave 0 and wave 15 of the first workgroups
        uint64_t *dtl; const size_t tlw = (size_t)TL_WGS * 2 * MAX_BATCH * TL_PHASES;
        CK(hipMalloc(&dtl, tlw * 8));
        CK(hipMemcpyToSymbol(HIP_SYMBOL(g_query_timeline), &dtl, sizeof dtl));
        const char *names[6] = {"barrier", "geometry + aim", "pass", "count", "store", "loop"};
        for (int mode = 0; mode < 2; ++mode) {
            CK(hipMemset(dtl, 0, tlw * 8));
            const float t = mode ? run<1024 | 8 | 32>(n, F, tab, sd, fstride, fwmax, sc, nseg, (uint64_t *)sb, 1) : run<1024>(n, F, tab, sd, fstride, fwmax, sc, nseg, (uint64_t *)sb, 1);
            std::vector<uint64_t> tl(tlw); CK(hipMemcpy(tl.data(), dtl, tlw * 8, hipMemcpyDeviceToHost));
            printf("timeline, %s (%.1f us with stamps):\n", mode ? "pure passes" : "full kernel", t);
            for (uint32_t wg = 0; wg < 2; ++wg) for (int w = 0; w < 2; ++w) {
                const uint64_t *tt = tl.data() + ((size_t)wg * 2 + w) * MAX_BATCH * TL_PHASES;
                double sum[6] = {0, 0, 0, 0, 0, 0};
                for (uint32_t f = 1; f + 1 < F; ++f) {
                    for (int ph = 0; ph < 5; ++ph) sum[ph] += (double)(tt[f * TL_PHASES + ph + 1] - tt[f * TL_PHASES + ph]);
                    sum[5] += (double)(tt[(f + 1) * TL_PHASES] - tt[f * TL_PHASES + 5]);
                }
                double tot = 0; for (int ph = 0; ph < 6; ++ph) tot += sum[ph];
                printf("  wg %u wave %2d: ", wg, w ? 15 : 0);
                for (int ph = 0; ph < 6; ++ph) printf("%s %.0f | ", names[ph], sum[ph] / (F - 2));
                printf("per frame %.0f ticks; frame 0 starts %.0f ticks after ... first stamp to last %.0f\n", tot / (F - 2), 0.0, (double)(tt[(F - 1) * TL_PHASES + 5] - tt[0]));
            }
        }
    }
    printf("pure passes, k_query_s64: 4 waves/SIMD %6.1f | 2 waves/SIMD %6.1f | 1 wave/SIMD %6.1f us\n",
           run<8 | 32>(n, F, tab, sd, fstride, fwmax, sc, nseg, (uint64_t *)sb), run<8 | 32, S64, 512>(n, F, tab, sd, fstride, fwmax, sc, nseg, (uint64_t *)sb),
           run<8 | 32, S64, 256>(n, F, tab, sd, fstride, fwmax, sc, nseg, (uint64_t *)sb));
    {   // shader-clock stamps at 13 points of the frame pass (AB & 4096): first and last wave of workgroup 0, averaged over frames 2 .. F-2
        uint64_t *drs; const size_t rsw = (size_t)2 * MAX_BATCH * 16;
        CK(hipMalloc(&drs, rsw * 8));
        CK(hipMemcpyToSymbol(HIP_SYMBOL(g_query_rowstamps), &drs, sizeof drs));
        const char *pts[12] = {"steps+reads 0", "outputs of the previous frame", "reduce 1", "combine 0", "steps+reads 1", "reduce 2", "combine 1", "steps+reads 2", "reduce 3", "combine 2", "steps+reads 3", "combine 3"};
        auto report = [&](const char *name, float us) {
            std::vector<uint64_t> rs(rsw); CK(hipMemcpy(rs.data(), drs, rsw * 8, hipMemcpyDeviceToHost));
            printf("pass stamps, %s (%.1f us with stamps), ticks per interval, first wave | last wave of workgroup 0:\n", name, us);
            for (int w = 0; w < 2; ++w) {
                double sum[12] = {0}; double frame = 0;
                for (uint32_t f = 2; f + 2 < F; ++f) {
                    const uint64_t *t = rs.data() + ((size_t)w * MAX_BATCH + f) * 16;
                    for (int i = 0; i < 12; ++i) sum[i] += (double)(t[i + 1] - t[i]);
                    frame += (double)(rs[((size_t)w * MAX_BATCH + f + 1) * 16] - t[0]);
                }
                printf("  %s wave: ", w ? "last " : "first");
                double tot = 0;
                for (int i = 0; i < 12; ++i) { printf("%s %.0f | ", pts[i], sum[i] / (F - 4)); tot += sum[i] / (F - 4); }
                printf("pass %.0f, frame to frame %.0f ticks\n", tot, frame / (F - 4));
            }
        };
#define RS(V, T, name) do { CK(hipMemset(drs, 0, rsw * 8)); const float t_ = run<(V) | 4096, S64, T>(n, F, tab, sd, fstride, fwmax, sc, nseg, (uint64_t *)sb, 1); report(name, t_); } while (0)
        RS(0, 1024, "full kernel, 4 waves/SIMD");
        RS(8 | 32, 1024, "pure passes, 4 waves/SIMD");
        RS(8 | 32, 256, "pure passes, 1 wave/SIMD");
        RS(8 | 32 | 1, 256, "pure passes without reductions, 1 wave/SIMD");
        RS(8 | 32 | 1 | 2 | 4, 256, "pure passes, none of the three, 1 wave/SIMD");
        RS(8 | 32 | 1 | 2 | 4, 1024, "pure passes, none of the three, 4 waves/SIMD");
    }
    // the same ablations per occupancy (512- and 256-thread workgroups run 2 and 4 rounds on the 256 CUs): does the pass follow its VALU count anywhere?
#define OCC(T, name) printf("pure passes at %s: all %6.1f | no reductions %6.1f | no LDS reads %6.1f | none of the three %6.1f | no hashing %6.1f | 1 frame %6.1f us\n", name, \
        run<8 | 32, S64, T>(n, F, tab, sd, fstride, fwmax, sc, nseg, (uint64_t *)sb), run<8 | 32 | 1, S64, T>(n, F, tab, sd, fstride, fwmax, sc, nseg, (uint64_t *)sb), \
        run<8 | 32 | 2, S64, T>(n, F, tab, sd, fstride, fwmax, sc, nseg, (uint64_t *)sb), run<8 | 32 | 1 | 2 | 4, S64, T>(n, F, tab, sd, fstride, fwmax, sc, nseg, (uint64_t *)sb), \
        run<8 | 32 | 16, S64, T>(n, F, tab, sd, fstride, fwmax, sc, nseg, (uint64_t *)sb), run<8 | 32, S64, T>(n, 1, tab, sd, fstride, fwmax, sc, nseg, (uint64_t *)sb))
    OCC(1024, "4 waves/SIMD");
    OCC(512, "2 waves/SIMD");
    OCC(256, "1 wave/SIMD ");
    printf("full kernel, k_query_s64: 1024-thread workgroups %6.1f | 512 %6.1f us\n",
           run<0>(n, F, tab, sd, fstride, fwmax, sc, nseg, (uint64_t *)sb), run<0, S64, 512>(n, F, tab, sd, fstride, fwmax, sc, nseg, (uint64_t *)sb));
    return 0;
}
