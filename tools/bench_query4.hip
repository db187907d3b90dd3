// bench_query4.hip -- round 3: the TILED query kernels at BASELINE config 4's shape (3840x2160, 8 coded frames, m ~ 2.45 Mbit: two
// LDS tiles per frame): k_query_s64t (rbf_kernels_s64.h) against k_query_r64t (rbf_kernels_r64.h).  Outputs compared, then timed:
// full | no staging (barriers kept) | pure (no staging, no barriers) | pure without LDS reads | pure without reductions.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-inline-asm -o build/bench_query4 tools/bench_query4.hip     Run: build/bench_query4 [frames]
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

enum Kern { R64T, S64T };
template <int AB, Kern K>
static float run(uint64_t n, uint32_t F, const FrameTable &tab, Seeds sd, uint64_t fstride, uint32_t tile_words, uint32_t *seg_cnt, uint64_t nseg, uint64_t *pwords, int R = 20)
{
    const FrameTable qtab = rank_tab(tab, F, K == S64T);
    const size_t lds = K == S64T ? s64t_lds_bytes(tile_words) : (size_t)(tile_words + 4) * 4;
    const uint32_t bx = (uint32_t)((nseg + QL_WAVES - 1) / QL_WAVES);
    auto launch = [&]() {
        if constexpr (K == R64T) k_query_r64t<AB><<<bx, QL_THREADS, lds, 0>>>(n, F, qtab, sd, g_image, fstride, tile_words, seg_cnt, nseg, pwords, 0u);
        else k_query_s64t<AB><<<bx, QL_THREADS, lds, 0>>>(n, F, qtab, sd, g_image, fstride, tile_words, seg_cnt, nseg, pwords, 0ull, 0ull);
    };
    if constexpr (K == R64T) CK(hipFuncSetAttribute((const void *)k_query_r64t<AB>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    else CK(hipFuncSetAttribute((const void *)k_query_s64t<AB>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
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
    const uint32_t F = argc > 1 ? (uint32_t)atoi(argv[1]) : 8u;
    const uint64_t n = 3840ull * 2160; const uint32_t m = 2451000;
    const uint64_t fwords = (m + 31) / 32, fstride = ((fwords + 3) & ~3ull);
    const uint64_t nseg = (n + QL_SEG_PIXELS - 1) / QL_SEG_PIXELS;
    const uint32_t cap = (uint32_t)((160 * 1024 - S64_GEO_BYTES) / 4 - 4) & ~3u;
    const uint32_t nt = (uint32_t)((fwords + cap - 1) / cap), tile_words = (uint32_t)(((fwords + nt - 1) / nt + 3) & ~3ull);
    std::vector<uint32_t> hf(fstride * F);
    srand(1);
    for (auto &x : hf) x = ((uint32_t)rand() << 16) ^ (uint32_t)rand();
    uint32_t *sb, *sc;
    CK(hipMalloc(&sb, (size_t)F * nseg * QL_P * 8)); CK(hipMalloc(&sc, (size_t)F * nseg * 4));
    { std::vector<uint32_t> img(hf.size()); for (size_t i = 0; i < hf.size(); ++i) img[i] = ~__builtin_bswap32(hf[i]);
      uint32_t *di; CK(hipMalloc(&di, img.size() * 4 + 64)); CK(hipMemcpy(di, img.data(), img.size() * 4, hipMemcpyHostToDevice)); g_image = di; }
    FrameTable tab{};
    for (uint32_t f = 0; f < F; ++f) { tab.f[f].m = m - 37 * f; tab.f[f].floor_k = 2; tab.f[f].T = 0x4D00000000000000ull + f; tab.f[f].M = 0; }
    Seeds sd{0x12345678, 0x87654321, 999};
    const size_t pwb = (size_t)F * nseg * QL_P * 8, scb = (size_t)F * nseg * 4;
    std::vector<uint8_t> a(pwb), b(pwb); std::vector<uint32_t> ca(F * nseg), cb(F * nseg);
    printf("3840x2160, %u frames, m = %u (%llu words), %u tiles of %u words\n", F, m, (unsigned long long)fwords, nt, tile_words);
    for (int w = 0; w < 20; ++w) run<0, R64T>(n, F, tab, sd, fstride, tile_words, sc, nseg, (uint64_t *)sb, 10);    // warm the clocks up
    const float tref = run<0, R64T>(n, F, tab, sd, fstride, tile_words, sc, nseg, (uint64_t *)sb);
    CK(hipMemcpy(a.data(), sb, pwb, hipMemcpyDeviceToHost)); CK(hipMemcpy(ca.data(), sc, scb, hipMemcpyDeviceToHost));
    CK(hipMemset(sb, 0xEE, pwb)); CK(hipMemset(sc, 0xEE, scb));
    const float tnew = run<0, S64T>(n, F, tab, sd, fstride, tile_words, sc, nseg, (uint64_t *)sb);
    CK(hipMemcpy(b.data(), sb, pwb, hipMemcpyDeviceToHost)); CK(hipMemcpy(cb.data(), sc, scb, hipMemcpyDeviceToHost));
    size_t diff = 0; for (size_t i = 0; i < pwb; ++i) diff += a[i] != b[i];
    size_t dc = 0; for (size_t i = 0; i < ca.size(); ++i) dc += ca[i] != cb[i];
    printf("k_query_r64t %6.1f us | k_query_s64t %6.1f us | outputs differ in %zu bytes, %zu counts\n", tref, tnew, diff, dc);
#define ROW(K, V, name) printf("%-44s full %6.1f | no staging %6.1f | pure %6.1f | pure, no LDS reads %6.1f | pure, no reductions %6.1f | pure, neither %6.1f us\n", name, \
        run<(V), K>(n, F, tab, sd, fstride, tile_words, sc, nseg, (uint64_t *)sb), run<(V) | 8, K>(n, F, tab, sd, fstride, tile_words, sc, nseg, (uint64_t *)sb), \
        run<(V) | 8 | 32, K>(n, F, tab, sd, fstride, tile_words, sc, nseg, (uint64_t *)sb), run<(V) | 8 | 32 | 2, K>(n, F, tab, sd, fstride, tile_words, sc, nseg, (uint64_t *)sb), \
        run<(V) | 8 | 32 | 1, K>(n, F, tab, sd, fstride, tile_words, sc, nseg, (uint64_t *)sb), run<(V) | 8 | 32 | 1 | 2, K>(n, F, tab, sd, fstride, tile_words, sc, nseg, (uint64_t *)sb)); fflush(stdout)
    for (int rep = 0; rep < 2; ++rep) {
        ROW(R64T, 0, "k_query_r64t");
        ROW(S64T, 0, "k_query_s64t");
        ROW(S64T, 4096, "k_query_s64t, first tile through registers");
        ROW(S64T, 2048, "k_query_s64t with k_query_s64's wave priorities");
    }
    for (uint32_t ff : {1u, 2u, 4u, 8u})
        if (ff <= F) printf("frames %u: k_query_s64t full %6.1f | pure %6.1f | k_query_r64t full %6.1f | pure %6.1f us\n", ff, run<0, S64T>(n, ff, tab, sd, fstride, tile_words, sc, nseg, (uint64_t *)sb),
               run<8 | 32, S64T>(n, ff, tab, sd, fstride, tile_words, sc, nseg, (uint64_t *)sb), run<0, R64T>(n, ff, tab, sd, fstride, tile_words, sc, nseg, (uint64_t *)sb),
               run<8 | 32, R64T>(n, ff, tab, sd, fstride, tile_words, sc, nseg, (uint64_t *)sb));
    return 0;
}
