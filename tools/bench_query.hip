// bench_query.hip -- ablation timing of the query kernels (k_query_lds, k_query_f64, k_query_p4) on a synthetic 1080p x 29-frame batch:
// kernel variants against each other (outputs compared), ablations, occupancy scaling and the frame-loop timeline (profiling aid).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <cstring>
#include "legacy/rbf_kernels_r64.h"      // round-3 snapshot (namespace rbf::legacy): these kernels left the library in round 4
using namespace rbf;
using namespace rbf::legacy;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

static uint4 *g_table_out = nullptr;            // k_query_f64: also write the hash table (nullptr = do not)
static const uint32_t *g_image = nullptr;       // probe image (~bswap of every filter dword), same row pitch as the filters
template <int AB, bool DB = true, int THREADS = QL_THREADS, int MODK = 0>
static float run(const uint64_t *masks, uint64_t mstride, uint64_t n, uint32_t F, const FrameTable &tab, Seeds sd, const uint32_t *filters,
                 uint64_t fstride, uint32_t fwmax, uint32_t *seg_bits, uint32_t *seg_cnt, uint64_t nseg, size_t lds)
{
    FrameTable qtab = tab;
    if (MODK == 1) {
        filters = g_image; lds += 32;
        for (uint32_t f = 0; f < F; ++f) { const double ninv = -1.0 / (double)qtab.f[f].m; memcpy(&qtab.f[f].M, &ninv, 8); }
    }
    if (MODK == 2) {                                             // k_query_r64: sorted thresholds in T, c << 8 in floor_k (rbf_kernels_r64.h)
        filters = g_image; lds += 32;
        std::vector<uint64_t> ts;
        for (uint32_t f = 0; f < F; ++f) ts.push_back(tab.f[f].T);
        std::sort(ts.begin(), ts.end());
        for (uint32_t f = 0; f < F; ++f) {
            const double ninv = -1.0 / (double)qtab.f[f].m; memcpy(&qtab.f[f].M, &ninv, 8);
            uint32_t c = 0; for (uint64_t t : ts) c += t < tab.f[f].T;
            qtab.f[f].floor_k = tab.f[f].floor_k | (c << 8);
            qtab.f[f].T = ts[f];
        }
    }
    auto launch = [&](dim3 g, dim3 b, size_t sh, hipStream_t st, uint64_t n_, uint32_t F_, const FrameTable &t_, Seeds s_, const uint32_t *f_, uint64_t fs_, uint32_t fw_, uint32_t *sc_, uint64_t ns_, uint64_t *pw_) {
        if constexpr (MODK == 2) k_query_r64<AB><<<g, b, sh, st>>>(n_, F_, t_, s_, f_, fs_, fw_, sc_, ns_, pw_, g_table_out, (AB & 128) ? 1u : 0u);
        else if constexpr (MODK == 1) k_query_f64<AB><<<g, b, sh, st>>>(n_, F_, t_, s_, f_, fs_, fw_, sc_, ns_, pw_, g_table_out);
        else k_query_lds<DB, true, AB><<<g, b, sh, st>>>(n_, F_, t_, s_, f_, fs_, fw_, sc_, ns_, pw_);
    };
    const void *kern = MODK == 2 ? (const void *)k_query_r64<AB> : MODK == 1 ? (const void *)k_query_f64<AB> : (const void *)k_query_lds<DB, true, AB>;
    uint64_t *pwords = (uint64_t *)seg_bits;
    CK(hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    const uint32_t bx = (uint32_t)((nseg + THREADS / 64 - 1) / (THREADS / 64));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int w = 0; w < 2; ++w)
        launch(dim3(bx), dim3(THREADS), lds, 0, n, F, qtab, sd, filters, fstride, fwmax, seg_cnt, nseg, pwords);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    const int R = 10;
    for (int r = 0; r < R; ++r)
        launch(dim3(bx), dim3(THREADS), lds, 0, n, F, qtab, sd, filters, fstride, fwmax, seg_cnt, nseg, pwords);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    return ms / R * 1000.f;
}

template <int AB, int THREADS = 1024>
static float run_p4(uint64_t n, uint32_t F, const FrameTable &tab, Seeds sd, uint64_t fstride, uint32_t fwmax, uint32_t *seg_cnt, uint64_t *pwords)
{
    auto kern = k_query_p4<AB>;
    CK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    FrameTable qtab = tab;
    for (uint32_t f = 0; f < F; ++f) { const double ninv = -1.0 / (double)qtab.f[f].m; memcpy(&qtab.f[f].M, &ninv, 8); }
    const uint64_t nseg = (n + P4_SEG_PIXELS - 1) / P4_SEG_PIXELS;
    const uint32_t bx = (uint32_t)((nseg + THREADS / 64 - 1) / (THREADS / 64));
    const size_t lds = (size_t)(((fwmax + 3) & ~3u) + 4) * 4;
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int w = 0; w < 2; ++w) kern<<<bx, THREADS, lds, 0>>>(n, F, qtab, sd, g_image, fstride, fwmax, seg_cnt, nseg, pwords, g_table_out);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    const int R = 10;
    for (int r = 0; r < R; ++r) kern<<<bx, THREADS, lds, 0>>>(n, F, qtab, sd, g_image, fstride, fwmax, seg_cnt, nseg, pwords, g_table_out);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    return ms / R * 1000.f;
}

int main()
{
    const uint64_t n = 1920 * 1080; const uint32_t F = 29; const uint32_t m = 611158;
    const uint64_t mstride = ((n + 63) / 64), fwords = (m + 31) / 32, fstride = ((fwords + 3) & ~3ull);
    const uint64_t nseg = (n + QL_SEG_PIXELS - 1) / QL_SEG_PIXELS;
    std::vector<uint64_t> hm(mstride * F); std::vector<uint32_t> hf(fstride * F);
    srand(1);
    for (auto &x : hm) { uint64_t v = 0; for (int b = 0; b < 64; ++b) if (rand() % 100 < 9) v |= 1ull << b; x = v; }
    for (auto &x : hf) x = ((uint32_t)rand() << 16) ^ (uint32_t)rand();
    uint64_t *dm; uint32_t *df, *sb, *sc;
    CK(hipMalloc(&dm, hm.size() * 8)); CK(hipMalloc(&df, hf.size() * 4 + 64));
    CK(hipMalloc(&sb, (size_t)F * nseg * QL_P * 8)); CK(hipMalloc(&sc, (size_t)F * nseg * 4));
    CK(hipMemcpy(dm, hm.data(), hm.size() * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(df, hf.data(), hf.size() * 4, hipMemcpyHostToDevice));
    { std::vector<uint32_t> img(hf.size()); for (size_t i = 0; i < hf.size(); ++i) img[i] = ~__builtin_bswap32(hf[i]);
      uint32_t *di; CK(hipMalloc(&di, img.size() * 4 + 64)); CK(hipMemcpy(di, img.data(), img.size() * 4, hipMemcpyHostToDevice)); g_image = di; }
    FrameTable tab{};
    for (uint32_t f = 0; f < F; ++f) { tab.f[f].m = m - 37 * f; tab.f[f].floor_k = 2; tab.f[f].T = 0x4D00000000000000ull; tab.f[f].M = (uint64_t)((((unsigned __int128)1) << 64) / tab.f[f].m); }
    Seeds sd{0x12345678, 0x87654321, 999};
    const uint32_t fwmax = (uint32_t)fwords;
    const size_t lds = 2 * (size_t)((fwmax + 3) & ~3u) * 4;
#define RUN(AB, what) printf("%-44s %8.1f us\n", what, run<AB>(dm, mstride, n, F, tab, sd, df, fstride, fwmax, sb, sc, nseg, lds));
#define RUN1(AB, what) printf("%-44s %8.1f us\n", "[fp64 mod] " what, run<AB, true, QL_THREADS, 1>(dm, mstride, n, F, tab, sd, df, fstride, fwmax, sb, sc, nseg, lds));
    RUN(0, "full kernel");
    {   // the FP64-reduction kernel must write the same pass bytes and segment counts
        const size_t pwb = (size_t)F * nseg * QL_P * 8, scb = (size_t)F * nseg * 4;
        std::vector<uint8_t> a(pwb), b(pwb); std::vector<uint32_t> ca(F * nseg), cb(F * nseg);
        CK(hipMemcpy(a.data(), sb, pwb, hipMemcpyDeviceToHost)); CK(hipMemcpy(ca.data(), sc, scb, hipMemcpyDeviceToHost));
        CK(hipMemset(sb, 0xEE, pwb)); CK(hipMemset(sc, 0xEE, scb));
        RUN1(0, "full kernel");
        CK(hipMemcpy(b.data(), sb, pwb, hipMemcpyDeviceToHost)); CK(hipMemcpy(cb.data(), sc, scb, hipMemcpyDeviceToHost));
        size_t diff = 0; for (size_t i = 0; i < pwb; ++i) diff += a[i] != b[i];
        size_t dc = 0; for (size_t i = 0; i < ca.size(); ++i) dc += ca[i] != cb[i];
        uint64_t passes = 0; for (auto c : ca) passes += c;
        printf("fp64-mod kernel vs Barrett kernel: %zu differing pass bytes, %zu differing segment counts (%llu passes)\n", diff, dc, (unsigned long long)passes);
    }
    if (getenv("R64")) {   // k_query_r64 (register staging + activation ranks) against k_query_f64: same pass bytes and counts; thresholds varied per frame
        FrameTable vt = tab;
        for (uint32_t f = 0; f < F; ++f) vt.f[f].T = 0x1000000000000000ull * ((f * 7) % 13 + 1) + f % 3;      // distinct and repeated values
        const size_t pwb = (size_t)F * nseg * QL_P * 8, scb = (size_t)F * nseg * 4;
        std::vector<uint8_t> a(pwb), b(pwb); std::vector<uint32_t> ca(F * nseg), cb(F * nseg);
        const float t1 = run<0, true, QL_THREADS, 1>(dm, mstride, n, F, vt, sd, df, fstride, fwmax, sb, sc, nseg, lds);
        CK(hipMemcpy(a.data(), sb, pwb, hipMemcpyDeviceToHost)); CK(hipMemcpy(ca.data(), sc, scb, hipMemcpyDeviceToHost));
        CK(hipMemset(sb, 0xEE, pwb)); CK(hipMemset(sc, 0xEE, scb));
        const float t2 = run<0, true, QL_THREADS, 2>(dm, mstride, n, F, vt, sd, df, fstride, fwmax, sb, sc, nseg, lds);
        CK(hipMemcpy(b.data(), sb, pwb, hipMemcpyDeviceToHost)); CK(hipMemcpy(cb.data(), sc, scb, hipMemcpyDeviceToHost));
        size_t diff = 0; for (size_t i = 0; i < pwb; ++i) diff += a[i] != b[i];
        size_t dc = 0; for (size_t i = 0; i < ca.size(); ++i) dc += ca[i] != cb[i];
        printf("k_query_f64 %.1f us, k_query_r64 %.1f us: %zu differing pass bytes, %zu differing segment counts\n", t1, t2, diff, dc);
        printf("%-60s %8.1f us\n", "[r64] full kernel (bench thresholds)", run<0, true, QL_THREADS, 2>(dm, mstride, n, F, tab, sd, df, fstride, fwmax, sb, sc, nseg, lds));
        printf("%-60s %8.1f us\n", "[r64] no staging (barrier kept)", run<8, true, QL_THREADS, 2>(dm, mstride, n, F, tab, sd, df, fstride, fwmax, sb, sc, nseg, lds));
        printf("%-60s %8.1f us\n", "[r64] no staging, no barrier (pure passes)", run<8 | 32, true, QL_THREADS, 2>(dm, mstride, n, F, tab, sd, df, fstride, fwmax, sb, sc, nseg, lds));
        printf("%-60s %8.1f us\n", "[r64] WITH the passthrough-frame loop (a frame with m = 0 in the batch)", run<128, true, QL_THREADS, 2>(dm, mstride, n, F, tab, sd, df, fstride, fwmax, sb, sc, nseg, lds));
        printf("%-60s %8.1f us\n", "[r64] full kernel again", run<0, true, QL_THREADS, 2>(dm, mstride, n, F, tab, sd, df, fstride, fwmax, sb, sc, nseg, lds));
        printf("%-60s %8.1f us\n", "[r64] no hashing", run<16, true, QL_THREADS, 2>(dm, mstride, n, F, tab, sd, df, fstride, fwmax, sb, sc, nseg, lds));
        printf("%-60s %8.1f us\n", "[f64] full kernel", run<0, true, QL_THREADS, 1>(dm, mstride, n, F, tab, sd, df, fstride, fwmax, sb, sc, nseg, lds));
        return 0;
    }
    if (getenv("R3")) {   // round 3: schedule variants of k_query_r64's frame pass (rbf_kernels_r64.h), each checked against k_query_f64
        FrameTable vt = tab;
        for (uint32_t f = 0; f < F; ++f) vt.f[f].T = 0x1000000000000000ull * ((f * 7) % 13 + 1) + f % 3;
        const size_t pwb = (size_t)F * nseg * QL_P * 8, scb = (size_t)F * nseg * 4;
        std::vector<uint8_t> a(pwb), b(pwb); std::vector<uint32_t> ca(F * nseg), cb(F * nseg);
        run<0, true, QL_THREADS, 1>(dm, mstride, n, F, vt, sd, df, fstride, fwmax, sb, sc, nseg, lds);
        CK(hipMemcpy(a.data(), sb, pwb, hipMemcpyDeviceToHost)); CK(hipMemcpy(ca.data(), sc, scb, hipMemcpyDeviceToHost));
        auto check = [&](const char *name, float t) {
            CK(hipMemcpy(b.data(), sb, pwb, hipMemcpyDeviceToHost)); CK(hipMemcpy(cb.data(), sc, scb, hipMemcpyDeviceToHost));
            size_t diff = 0; for (size_t i = 0; i < pwb; ++i) diff += a[i] != b[i];
            size_t dc = 0; for (size_t i = 0; i < ca.size(); ++i) dc += ca[i] != cb[i];
            printf("%-58s %8.1f us (varied thresholds)   vs k_query_f64: %zu differing pass bytes, %zu differing segment counts\n", name, t, diff, dc);
            CK(hipMemset(sb, 0xEE, pwb)); CK(hipMemset(sc, 0xEE, scb));
        };
#define R3V(V, name) do { \
        CK(hipMemset(sb, 0xEE, pwb)); CK(hipMemset(sc, 0xEE, scb)); \
        check(name, run<(V), true, QL_THREADS, 2>(dm, mstride, n, F, vt, sd, df, fstride, fwmax, sb, sc, nseg, lds)); \
        const float tf = run<(V), true, QL_THREADS, 2>(dm, mstride, n, F, tab, sd, df, fstride, fwmax, sb, sc, nseg, lds); \
        const float ts = run<(V) | 8, true, QL_THREADS, 2>(dm, mstride, n, F, tab, sd, df, fstride, fwmax, sb, sc, nseg, lds); \
        const float tp = run<(V) | 8 | 32, true, QL_THREADS, 2>(dm, mstride, n, F, tab, sd, df, fstride, fwmax, sb, sc, nseg, lds); \
        const float tn = run<(V) | 8 | 32 | 4, true, QL_THREADS, 2>(dm, mstride, n, F, tab, sd, df, fstride, fwmax, sb, sc, nseg, lds); \
        printf("    %-54s full %6.1f | no staging %6.1f | pure passes %6.1f | pure, no ballots %6.1f us\n", name, tf, ts, tp, tn); } while (0)
        for (int rep = 0; rep < 2; ++rep) {
        R3V(0, "r64 base (group schedule, 3-slot stager)");
        // (The schedule variants of section 1 of profiles/r03_query_ablation.txt -- two staging slots, pixel pipelines of depth 2
        // and 3, reads before the previous combine, plane ballots -- were template bits of k_query_r64 in the working tree of that
        // experiment only; what they taught went into k_query_s64 (rbf_kernels_s64.h, tools/bench_query3.hip).)
        }
        return 0;
    }
#define RUNP(AB, TH, PARTS, what) printf("%-60s %8.1f us\n", "[fp64 mod] " what, run<AB, true, TH, 1>(dm, mstride, n, F, tab, sd, df, fstride, fwmax, sb, sc, nseg, lds));
    {   // k_query_p4: 4 pixels per lane, two workgroups per CU; same pass bytes (its segments are 256 pixels, so the counts are compared as sums)
        const size_t pwb = (size_t)F * nseg * QL_P * 8;
        std::vector<uint8_t> a(pwb), b(pwb);
        run<0, true, QL_THREADS, 1>(dm, mstride, n, F, tab, sd, df, fstride, fwmax, sb, sc, nseg, lds);
        CK(hipMemcpy(a.data(), sb, pwb, hipMemcpyDeviceToHost));
        CK(hipMemset(sb, 0xEE, pwb));
        uint32_t *sc4; CK(hipMalloc(&sc4, (size_t)F * nseg * 2 * 4));
        printf("%-60s %8.1f us\n", "[p4] k_query_p4 full kernel", run_p4<0>(n, F, tab, sd, fstride, fwmax, sc4, (uint64_t *)sb));
        CK(hipMemcpy(b.data(), sb, pwb, hipMemcpyDeviceToHost));
        size_t diff = 0; for (size_t i = 0; i < pwb; ++i) diff += a[i] != b[i];
        printf("k_query_p4 vs k_query_f64: %zu differing pass bytes\n", diff);
        printf("%-60s %8.1f us\n", "[p4] no DMA, no barrier (pure passes)", run_p4<8 | 32>(n, F, tab, sd, fstride, fwmax, sc4, (uint64_t *)sb));
        printf("%-60s %8.1f us\n", "[p4] no DMA (barriers kept)", run_p4<8>(n, F, tab, sd, fstride, fwmax, sc4, (uint64_t *)sb));
        printf("%-60s %8.1f us\n", "[p4] no hashing", run_p4<16>(n, F, tab, sd, fstride, fwmax, sc4, (uint64_t *)sb));
        printf("%-60s %8.1f us\n", "[p4] no LDS probes", run_p4<2>(n, F, tab, sd, fstride, fwmax, sc4, (uint64_t *)sb));
        printf("%-60s %8.1f us\n", "[p4] no reductions", run_p4<1>(n, F, tab, sd, fstride, fwmax, sc4, (uint64_t *)sb));
        printf("%-60s %8.1f us\n", "[p4] no ballots", run_p4<4>(n, F, tab, sd, fstride, fwmax, sc4, (uint64_t *)sb));
        printf("%-60s %8.1f us\n", "[p4] 512-thread workgroups (4 per CU)", run_p4<0, 512>(n, F, tab, sd, fstride, fwmax, sc4, (uint64_t *)sb));
    }
    { uint4 *t; CK(hipMalloc(&t, (n + 512) * 32)); g_table_out = t; RUNP(0, 1024, 1, "full kernel + hash table written for the next batch"); g_table_out = nullptr; CK(hipFree(t)); }
    RUNP(0, 1024, 1, "full kernel (again, no table)");
    RUNP(8 | 32 | 8192, 1024, 1, "pure passes, three-instruction probe address (shift, and, add)");
    RUNP(8 | 32, 1024, 1, "pure passes, two-instruction probe address (lshr + lshl_add)");
    RUNP(8192, 1024, 1, "full, three-instruction probe address");
    
    RUNP(8 | 32, 1024, 1, "pure passes, 4 waves/SIMD (1024-thread workgroups)");
    RUNP(8 | 32, 512, 1, "pure passes, 2 waves/SIMD (512-thread workgroups)");
    RUNP(8 | 32, 256, 1, "pure passes, 1 wave/SIMD (256-thread workgroups)");
    RUNP(8 | 32 | 4, 1024, 1, "pure passes, no ballots");
    RUNP(8, 1024, 1, "no DMA (barrier kept)");
    RUN1(8 | 32, "no DMA, no barrier: pure frame passes + stores");
    printf("%-44s %8.1f us\n", "[fp64 mod] same, 512-thread WGs (2 waves/SIMD)", run<8 | 32, true, 512, 1>(dm, mstride, n, F, tab, sd, df, fstride, fwmax, sb, sc, nseg, lds));
    printf("%-44s %8.1f us\n", "[fp64 mod] same, 256-thread WGs (1 wave/SIMD)", run<8 | 32, true, 256, 1>(dm, mstride, n, F, tab, sd, df, fstride, fwmax, sb, sc, nseg, lds));
    printf("%-44s %8.1f us\n", "[fp64 mod] same, 128-thread WGs (2 SIMDs busy)", run<8 | 32, true, 128, 1>(dm, mstride, n, F, tab, sd, df, fstride, fwmax, sb, sc, nseg, lds));
    printf("%-44s %8.1f us\n", "[fp64 mod] full kernel, 512-thread WGs", run<0, true, 512, 1>(dm, mstride, n, F, tab, sd, df, fstride, fwmax, sb, sc, nseg, lds));
    RUN1(8 | 32 | 2, "no DMA, no barrier, no LDS probes");
    RUN1(8 | 32 | 1, "no DMA, no barrier, no reductions");
    RUN1(8 | 32 | 4, "no DMA, no barrier, no pass counting (ballots)");
    RUN1(8 | 32 | 16, "no DMA, no barrier, no hashing");
    RUN1(8 | 32 | 1 | 2 | 4, "no DMA, no barrier, no reductions/probes/ballots");
    RUN1(8, "no DMA (barrier kept)");
    RUN1(32, "no barrier (DMA kept; wrong results)");
    RUN1(2 | 8, "no LDS probes, no filter DMA");
    {   // timeline of the frame loop: shader-clock stamps of wave 0 and wave 15 of the first workgroups
        uint64_t *dtl; const size_t tlw = (size_t)TL_WGS * 2 * MAX_BATCH * TL_PHASES;
        CK(hipMalloc(&dtl, tlw * 8)); CK(hipMemset(dtl, 0, tlw * 8));
        CK(hipMemcpyToSymbol(HIP_SYMBOL(g_query_timeline), &dtl, sizeof dtl));
        RUN1(1024, "with timeline stamps");
        std::vector<uint64_t> tl(tlw); CK(hipMemcpy(tl.data(), dtl, tlw * 8, hipMemcpyDeviceToHost));
        const char *names[6] = {"vmcnt(0) wait", "barrier", "next-frame geometry", "frame pass + DMA issue + store", "hold", "loop overhead to next frame"};
        for (uint32_t wg = 0; wg < TL_WGS; ++wg) for (int w = 0; w < 2; ++w) {
            const uint64_t *t = tl.data() + ((size_t)wg * 2 + w) * MAX_BATCH * TL_PHASES;
            double sum[6] = {0, 0, 0, 0, 0, 0};
            for (uint32_t f = 1; f + 1 < F; ++f) {
                for (int ph = 0; ph < 5; ++ph) sum[ph] += (double)(t[f * TL_PHASES + ph + 1] - t[f * TL_PHASES + ph]);
                sum[5] += (double)(t[(f + 1) * TL_PHASES] - t[f * TL_PHASES + 5]);
            }
            printf("timeline wg %u wave %s: ", wg, w ? "15" : " 0");
            double tot = 0; for (int ph = 0; ph < 6; ++ph) tot += sum[ph];
            for (int ph = 0; ph < 6; ++ph) printf("%s %.0f (%.0f%%) | ", names[ph], sum[ph] / (F - 2), 100.0 * sum[ph] / tot);
            printf("cycles per frame %.0f\n", tot / (F - 2));
        }
    }
    RUN1(16, "no hashing");
    RUN1(1, "no reductions (mod m)");
    RUN1(2, "no LDS probes");
    RUN1(4, "no ballot/compaction");
    RUN1(8, "no filter DMA");
    RUN1(2 | 4 | 8 | 16, "only reductions");
    RUN1(1 | 4 | 8 | 16, "only probes");
    { const size_t lds1 = (size_t)((fwmax + 3) & ~3u) * 4;
      printf("%-44s %8.1f us\n", "single buffer, 1024 thr (1 WG/CU)", run<0, false, 1024>(dm, mstride, n, F, tab, sd, df, fstride, fwmax, sb, sc, nseg, lds1));
      printf("%-44s %8.1f us\n", "single buffer, 512 thr (2 WG/CU)", run<0, false, 512>(dm, mstride, n, F, tab, sd, df, fstride, fwmax, sb, sc, nseg, lds1));
      printf("%-44s %8.1f us\n", "single buffer, 256 thr (2 WG/CU)", run<0, false, 256>(dm, mstride, n, F, tab, sd, df, fstride, fwmax, sb, sc, nseg, lds1));
      printf("%-44s %8.1f us\n", "double buffer, 512 thr (1 WG/CU)", run<0, true, 512>(dm, mstride, n, F, tab, sd, df, fstride, fwmax, sb, sc, nseg, lds)); }
    RUN(32, "no barrier / dma wait (WRONG RESULTS)");
    RUN(64, "no staging flush");
    RUN(32 | 64, "no barrier, no flush");
    RUN(32 | 64 | 8, "no barrier, no flush, no DMA");
    RUN(512, "frame order rotated per workgroup");
    RUN(128, "DMA of half the filter (WRONG RESULTS)");
    RUN(256, "DMA issued by 4 waves only");
    RUN(16, "no hashing");
    RUN(1, "no reductions (mod m)");
    RUN(2, "no LDS probes");
    RUN(4, "no ballot/compaction");
    RUN(8, "no filter DMA");
    RUN(1 | 2, "no reductions, no probes");
    RUN(1 | 2 | 4, "no reductions/probes/compaction");
    RUN(1 | 2 | 4 | 8, "... and no DMA");
    RUN(1 | 2 | 4 | 8 | 16, "... and no hashing (frame loop skeleton)");
    RUN(2 | 4 | 8 | 16, "only reductions");
    RUN(1 | 4 | 8 | 16, "only probes");
    RUN(1 | 2 | 8 | 16, "only compaction");
    return 0;
}
