// bench_query.hip -- ablation timing of k_query_lds on a synthetic 1080p x 29-frame batch (profiling aid).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../new_bloom_filter_repo_amd/csrc/rbf_kernels_lds.h"
using namespace rbf;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int AB, bool DB = true, int THREADS = QL_THREADS>
static float run(const uint64_t *masks, uint64_t mstride, uint64_t n, uint32_t F, const FrameTable &tab, Seeds sd, const uint32_t *filters,
                 uint64_t fstride, uint32_t fwmax, uint32_t *seg_bits, uint32_t *seg_cnt, uint64_t nseg, size_t lds)
{
    auto kern = k_query_lds<DB, true, AB>;
    uint64_t *pwords = (uint64_t *)seg_bits;
    CK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    const uint32_t bx = (uint32_t)((nseg + THREADS / 64 - 1) / (THREADS / 64));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int w = 0; w < 2; ++w)
        hipLaunchKernelGGL(kern, dim3(bx), dim3(THREADS), lds, 0, n, F, tab, sd, filters, fstride, fwmax, seg_cnt, nseg, pwords);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    const int R = 10;
    for (int r = 0; r < R; ++r)
        hipLaunchKernelGGL(kern, dim3(bx), dim3(THREADS), lds, 0, n, F, tab, sd, filters, fstride, fwmax, seg_cnt, nseg, pwords);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    return ms / R * 1000.f;
}

int main()
{
    const uint64_t n = 1920 * 1080; const uint32_t F = 29; const uint32_t m = 611158;
    const uint64_t mstride = ((n + 63) / 64), fwords = (m + 31) / 32, fstride = ((fwords + 1) & ~1ull);
    const uint64_t nseg = (n + QL_SEG_PIXELS - 1) / QL_SEG_PIXELS;
    std::vector<uint64_t> hm(mstride * F); std::vector<uint32_t> hf(fstride * F);
    srand(1);
    for (auto &x : hm) { uint64_t v = 0; for (int b = 0; b < 64; ++b) if (rand() % 100 < 9) v |= 1ull << b; x = v; }
    for (auto &x : hf) x = ((uint32_t)rand() << 16) ^ (uint32_t)rand();
    uint64_t *dm; uint32_t *df, *sb, *sc;
    CK(hipMalloc(&dm, hm.size() * 8)); CK(hipMalloc(&df, hf.size() * 4 + 64));
    CK(hipMalloc(&sb, (size_t)F * nseg * QL_P * 8)); CK(hipMalloc(&sc, (size_t)F * nseg * 4));
    CK(hipMemcpy(dm, hm.data(), hm.size() * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(df, hf.data(), hf.size() * 4, hipMemcpyHostToDevice));
    FrameTable tab{};
    for (uint32_t f = 0; f < F; ++f) { tab.f[f].m = m - 37 * f; tab.f[f].floor_k = 2; tab.f[f].T = 0x4D00000000000000ull; tab.f[f].M = (uint64_t)((((unsigned __int128)1) << 64) / tab.f[f].m); }
    Seeds sd{0x12345678, 0x87654321, 999};
    const uint32_t fwmax = (uint32_t)fwords;
    const size_t lds = 2 * (size_t)((fwmax + 3) & ~3u) * 4;
#define RUN(AB, what) printf("%-44s %8.1f us\n", what, run<AB>(dm, mstride, n, F, tab, sd, df, fstride, fwmax, sb, sc, nseg, lds));
    RUN(0, "full kernel");
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
