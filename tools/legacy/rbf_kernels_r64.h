// LEGACY (round 2 / round 3): a frozen copy of csrc/rbf_kernels_r64.h as it stood before the round-4 prune, kept for the old harnesses of tools/
// (bench_query*.hip, bench_insert.hip) and for A/B comparisons against the kernels that replaced these.  NOT part of the library: nothing
// under new_bloom_filter_repo_amd/ includes it.  Everything lives in namespace rbf::legacy.
// rbf_kernels_r64.h -- k_query_r64: k_query_f64 (rbf_kernels_q64.h) with the two changes its own measurements asked for.
//
// 1. The probe image of the NEXT frame is staged through REGISTERS -- plain 16-byte global loads issued between the pixel
//    groups of this frame's pass, ds_write_b128 two groups later -- instead of LDS-DMA.  A global_load_lds instruction
//    holds its wave for 160-240 cycles and the CU serialises them: in k_query_f64 the last wave spends ~3 500 cycles per
//    frame issuing its five pieces while the first waves idle ~2 700 cycles at the next barrier
//    (profiles/r02_query_timeline.txt); a plain load and an LDS write issue in a few cycles each.  k_query_f64 could not
//    afford the 12 registers (three pieces in flight) -- it sits at 124 -- hence:
// 2. ACTIVATION RANKS instead of the 64-bit activation hash.  "h_act < T_f" is all a frame ever asks of a pixel's
//    activation hash, and a batch has at most MAX_BATCH thresholds.  With the coded frames' thresholds sorted,
//        rank(pixel) = #{ j : t_j <= h_act },          c_f = #{ j : t_j < T_f }   (host),
//        h_act < T_f  <=>  rank <= c_f
//    (=>: every t_j <= h_act is < T_f; <=: if h_act >= T_f then T_f and everything below it are counted, rank > c_f).
//    The rank is computed once per pixel next to the hashes and kept in ONE BYTE: 2 registers per lane instead of 16.
//
// FrameTable as this kernel reads it (host: launch_query): f[k].M = bits of -1.0 / m_k; f[k].floor_k = floor(k*) | c_k << 8;
// f[j].T = j-th smallest threshold of the coded frames, ~0 past the last one (NOT frame j's own threshold).
// Everything else -- pixel ownership, hashing, hash-table hand-over, pass bytes, segment counts, the double-buffered
// LDS layout with the SAFE dwords -- is k_query_f64's.
#pragma once
#include "rbf_kernels_q64.h"
#include <type_traits>

namespace rbf { namespace legacy {

typedef uint32_t r64_u32x4 __attribute__((ext_vector_type(4)));

// Up to five 1 KiB pieces per wave and frame (a 76 KB image is 77 pieces over 16 waves); three register slots (a fourth
// spills: the kernel sits at 127 VGPRs).  Schedule over the four pixel groups of a pass:  g0: load P0 P1 | g1: load P2 |
// g2: write P0 P1, load P3 P4 | g3: write P2 | end: write P3 P4.
// BRANCH-FREE on purpose: with `if (piece inside the row) load / store` the compiler loses track of the vector-memory counter
// across the exec-masked regions and waits for vmcnt(0) -- a full L2 round trip -- in front of every step (measured: 17 us per
// launch).  Instead every lane's offset is clamped to the row's last 16 bytes: lanes past the end load and rewrite that piece
// (same bytes, same address), and when there is no next frame the stager is aimed at a row and a buffer nobody reads any more.
struct RowStager {
    const uint8_t *row;             // next frame's image row (uniform)
    uint32_t lds_base;              // byte address of the destination buffer (uniform)
    uint32_t last;                  // row bytes - 16 (uniform): the clamp
    uint32_t off0;                  // wave * 1024 + lane * 16
    uint4 a, b, c;

    __device__ __forceinline__ void aim(const uint32_t *row_words, uint32_t lds_dst, uint32_t bytes /* multiple of 16, >= 16 */)
    {
        row = reinterpret_cast<const uint8_t *>(row_words);
        lds_base = lds_dst;
        last = bytes - 16u;
    }
    __device__ __forceinline__ uint32_t off(int i) const { return min(off0 + (uint32_t)i * (QL_WAVES * 1024u), last); }
    __device__ __forceinline__ uint4 load(int i) const { return *reinterpret_cast<const uint4 *>(row + off(i)); }
    __device__ __forceinline__ void store(int i, const uint4 &v) const
    {
        *reinterpret_cast<__attribute__((address_space(3))) r64_u32x4 *>((uintptr_t)(lds_base + off(i))) = r64_u32x4{v.x, v.y, v.z, v.w};   // ds_write_b128
    }
    template <int AB>
    __device__ __forceinline__ void at(int g)
    {
        if (AB & 8) return;
        if (g == 0) { a = load(0); b = load(1); }
        else if (g == 1) { c = load(2); }
        else if (g == 2) { store(0, a); store(1, b); a = load(3); b = load(4); }
        else if (g == 3) { store(2, c); }
        else { store(3, a); store(4, b); }                        // g == 4: after the pass
    }
};

// rank byte B of `ranks` <= c ? p : safe -- the extra probe's position -- in two instructions: an SDWA compare that reads the
// byte in place, and a select.  (All operands in VGPRs: VCC is the select's one allowed scalar source.)
template <int B>
__device__ __forceinline__ uint32_t rank_select(uint32_t ranks, uint32_t c, uint32_t safe, uint32_t p)
{
    uint32_t r;
    if constexpr (B == 0) asm("v_cmp_le_u32_sdwa vcc, %1, %2 src0_sel:BYTE_0 src1_sel:DWORD\n\tv_cndmask_b32 %0, %3, %4, vcc" : "=v"(r) : "v"(ranks), "v"(c), "v"(safe), "v"(p) : "vcc");
    else if constexpr (B == 1) asm("v_cmp_le_u32_sdwa vcc, %1, %2 src0_sel:BYTE_1 src1_sel:DWORD\n\tv_cndmask_b32 %0, %3, %4, vcc" : "=v"(r) : "v"(ranks), "v"(c), "v"(safe), "v"(p) : "vcc");
    else if constexpr (B == 2) asm("v_cmp_le_u32_sdwa vcc, %1, %2 src0_sel:BYTE_2 src1_sel:DWORD\n\tv_cndmask_b32 %0, %3, %4, vcc" : "=v"(r) : "v"(ranks), "v"(c), "v"(safe), "v"(p) : "vcc");
    else asm("v_cmp_le_u32_sdwa vcc, %1, %2 src0_sel:BYTE_3 src1_sel:DWORD\n\tv_cndmask_b32 %0, %3, %4, vcc" : "=v"(r) : "v"(ranks), "v"(c), "v"(safe), "v"(p) : "vcc");
    return r;
}

// frame_part_f64 with (a) the activation taken from the rank bytes (pixel it has its extra probe iff rank <= c) and
// (b) the stager's hooks between the pixel groups.
template <int FK, int AB, bool CHECK_VALID>
__device__ __forceinline__ void frame_pass_r64(
    const double (&hd1)[QL_P], const uint32_t (&hl1)[QL_P], const double (&hd2)[QL_P], const uint32_t (&hl2)[QL_P],
    uint32_t rank_lo, uint32_t rank_hi, uint32_t c /* VGPR */, uint32_t validmask, uint32_t lds_base_bytes, uint32_t safe_pos /* VGPR */, uint32_t m, double ninv,
    uint32_t fk_rt, uint32_t &pbf, uint32_t &npass, RowStager &st)
{
    constexpr int G = 2, NG = QL_P / G;
    auto extra = [&](int it, uint32_t p) -> uint32_t {            // the activated extra probe, or SAFE (`it` is a constant after unrolling)
        const uint32_t rk = it < 4 ? rank_lo : rank_hi;
        switch (it & 3) {
        case 0: return rank_select<0>(rk, c, safe_pos, p);
        case 1: return rank_select<1>(rk, c, safe_pos, p);
        case 2: return rank_select<2>(rk, c, safe_pos, p);
        default: return rank_select<3>(rk, c, safe_pos, p);
        }
    };
    if constexpr (FK < 0) {                                       // runtime floor(k*): rare geometries and partial waves, plain loop
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            st.template at<AB>(g);
#pragma unroll
            for (int e = 0; e < G; ++e) {
                const int it = g * G + e;
                uint32_t pos = mod_m_f64(hd1[it], hl1[it], ninv, m);
                const uint32_t step = mod_m_f64(hd2[it], hl2[it], ninv, m);
                uint32_t fail = CHECK_VALID ? ~(validmask << (31 - it)) & 0x80000000u : 0u;
                for (uint32_t j = 0; j < fk_rt; ++j) {
                    fail = (probe_image_word<AB>(lds_base_bytes, pos) << (pos & 31u)) | fail;
                    const uint32_t s2 = pos + step;
                    pos = min(s2, s2 - m);
                }
                const uint32_t pc = extra(it, pos);
                fail = (probe_image_word<AB>(lds_base_bytes, pc) << (pc & 31u)) | fail;
                pbf = __builtin_amdgcn_alignbit(pbf, fail, 31);
                if (!(AB & 4)) npass += __popcll(__ballot((int32_t)fail >= 0));
            }
        }
        st.template at<AB>(NG);
    } else {
        constexpr int NP = FK + 1;
        uint32_t pos[NG][G][NP], wrd[NG][G][NP];
        auto positions = [&](int g) {
#pragma unroll
            for (int e = 0; e < G; ++e) {
                const int it = g * G + e;
                uint32_t p, step;
                if (AB & 1) { p = hl1[it] & 0x7FFFFu; step = hl2[it] & 0x3FFFFu; }
                else { p = mod_m_f64(hd1[it], hl1[it], ninv, m); step = mod_m_f64(hd2[it], hl2[it], ninv, m); }
#pragma unroll
                for (int j = 0; j < FK; ++j) {
                    pos[g][e][j] = p;
                    const uint32_t s2 = p + step;
                    p = min(s2, s2 - m);
                }
                pos[g][e][FK] = extra(it, p);
            }
        };
        auto loads = [&](int g) {
#pragma unroll
            for (int e = 0; e < G; ++e)
#pragma unroll
                for (int j = 0; j < NP; ++j)
                    wrd[g][e][j] = (AB & 2) ? (pos[g][e][j] * 0x9E3779B1u) : probe_image_word<AB>(lds_base_bytes, pos[g][e][j]);
        };
        auto combine = [&](int g) {
#pragma unroll
            for (int e = 0; e < G; ++e) {
                const int it = g * G + e;
                uint32_t fail = CHECK_VALID ? ~(validmask << (31 - it)) & 0x80000000u : 0u;
#pragma unroll
                for (int j = 0; j < NP; ++j) fail = (wrd[g][e][j] << (pos[g][e][j] & 31u)) | fail;
                pbf = __builtin_amdgcn_alignbit(pbf, fail, 31);   // (pbf << 1) | (fail >> 31)
                if (!(AB & 4)) npass += __popcll(__ballot((int32_t)fail >= 0));
            }
        };
        // P0 L0 S0 | P1 C0 L1 S1 | ... | C(last) S(end): as frame_part_f64, with the stager's step after every group's reads
        positions(0);
        __builtin_amdgcn_sched_barrier(0);
        loads(0);
        st.template at<AB>(0);
#pragma unroll
        for (int g = 1; g < NG; ++g) {
            __builtin_amdgcn_sched_barrier(0);
            positions(g);
            __builtin_amdgcn_sched_barrier(0);
            combine(g - 1);
            __builtin_amdgcn_sched_barrier(0);
            loads(g);
            st.template at<AB>(g);
        }
        __builtin_amdgcn_sched_barrier(0);
        combine(NG - 1);
        st.template at<AB>(NG);
    }
}

// AB bits as in k_query_f64: 8 = no staging, 16 = no hashing, 32 = no barrier (wrong results), 64 = no output.
template <int AB = 0>
__global__ __launch_bounds__(QL_THREADS) void k_query_r64(
    uint64_t n, uint32_t nframes, const FrameTable tab /* see the header comment */, Seeds seeds,
    const uint32_t *__restrict__ image, uint64_t image_stride_words32, uint32_t fwords_max,
    uint32_t *__restrict__ seg_cnt, uint64_t nseg, uint64_t *__restrict__ pass_words,
    uint4 *__restrict__ table_out /* nullable: write the hash table of the frame geometry for the NEXT batch's insert kernel */,
    uint32_t any_passthrough /* some frame has m == 0 */)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    // two buffers; each ends with 4 dwords that the staging never touches, the first of which stays 0 (SAFE)
    const uint32_t bufwords = ((fwords_max + 3u) & ~3u) + 4u;
    const uint32_t safe_pos = ((fwords_max + 3u) & ~3u) << 5;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t nwaves = blockDim.x >> 6;
    const uint64_t seg = (uint64_t)blockIdx.x * nwaves + wave;
    const bool live = seg < nseg;
    const uint64_t base = seg * QL_SEG_PIXELS;
    if (threadIdx.x < 8u) lds[(threadIdx.x >> 2) * bufwords + (bufwords - 4u) + (threadIdx.x & 3u)] = 0u;   // visible after the first barrier

    // ---- frame-independent part: the hashes of my 8 consecutive pixel indices as (double, low dword), and the activation ranks
    static_assert(QL_P == 8, "a lane's verdicts fill one byte; hash3_run8 hashes runs of 8; two rank registers");
    double hd1[QL_P], hd2[QL_P];
    uint32_t hl1[QL_P], hl2[QL_P];
    uint32_t rank_lo = 0, rank_hi = 0;                             // byte it & 3 of (it < 4 ? lo : hi)
    uint32_t validmask = 0;
    const uint64_t i0 = base + (uint64_t)lane * QL_P;
    {
        uint64_t h1[QL_P], h2[QL_P], ha[QL_P];
#pragma unroll
        for (int it = 0; it < QL_P; ++it) {
            h1[it] = 0; h2[it] = 0; ha[it] = ~0ull;
            if (live && i0 + it < n) validmask |= 1u << it;
        }
        if (AB & 16) {
#pragma unroll
            for (int it = 0; it < QL_P; ++it) { h1[it] = (i0 + it) * P1; h2[it] = (i0 + it) * P2 + seeds.h2; ha[it] = (i0 + it) * P3; }
        } else if (!hash3_run8((uint32_t)i0, validmask, seeds, h1, h2, ha)) {
#pragma unroll
            for (int it = 0; it < QL_P; ++it) {                  // mixed key lengths in this wave: index by index
                const bool act = (validmask >> it) & 1u;
                const Hash3 h = hash3_index((uint32_t)(i0 + it), act, seeds);
                h1[it] = h.h1; h2[it] = h.h2; ha[it] = h.ha;
            }
        }
#pragma unroll
        for (int it = 0; it < QL_P; ++it) {
            hd1[it] = (double)h1[it]; hl1[it] = (uint32_t)h1[it];
            hd2[it] = (double)h2[it]; hl2[it] = (uint32_t)h2[it];
        }
        if (table_out && live) {
#pragma unroll
            for (int it = 0; it < QL_P; ++it) hash_table_store(table_out, seg, lane, it, h1[it], h2[it], ha[it]);
        }
        // ranks = upper_bound of h_act in the sorted thresholds: a branch-free binary search over a copy in LDS (buffer 1 is
        // free until the first pass stages into it, which is behind the loop's first barrier).  log2 steps per pixel instead of
        // one compare per (pixel, threshold): 3 us -> 1 us per launch at 29 frames.
        uint64_t *tl = reinterpret_cast<uint64_t *>(lds + bufwords);
        if (threadIdx.x < 2u * MAX_BATCH) tl[threadIdx.x] = threadIdx.x < nframes ? tab.f[threadIdx.x < nframes ? threadIdx.x : 0u].T : ~0ull;
        __syncthreads();
        uint32_t top = 1;                                         // largest power of two <= nframes: 2 * top - 1 >= nframes entries are searched
        while (2u * top <= nframes) top *= 2u;
        top = __builtin_amdgcn_readfirstlane(top);
        uint32_t r[QL_P];
#pragma unroll
        for (int it = 0; it < QL_P; ++it) r[it] = 0;
        for (uint32_t step = top; step; step >>= 1) {
#pragma unroll
            for (int it = 0; it < QL_P; ++it) {
                const uint64_t t = tl[r[it] + step - 1u];
                r[it] |= t <= ha[it] ? step : 0u;
            }
        }
        rank_lo = r[0] | (r[1] << 8) | (r[2] << 16) | (r[3] << 24);
        rank_hi = r[4] | (r[5] << 8) | (r[6] << 16) | (r[7] << 24);
    }
    const bool whole_wave = __builtin_amdgcn_readfirstlane((uint32_t)__all(validmask == 0xFFu)) != 0u;   // every lane owns 8 positions inside the frame
    uint8_t *pass_bytes = reinterpret_cast<uint8_t *>(pass_words);

    // passthrough frames (m == 0): nothing passes.  Only walked when the host saw one: 29 dependent scalar loads otherwise (~2 us).
    if (any_passthrough)
    for (uint32_t g = 0; g < nframes; ++g) {
        if (tab.f[g].m == 0) {
            if (live && lane == 0) seg_cnt[(uint64_t)g * nseg + seg] = 0;
            if (live) pass_bytes[((uint64_t)g * nseg + seg) * (QL_SEG_PIXELS / 8) + lane] = 0;
        }
    }
    auto next_active = [&](uint32_t k) -> uint32_t { while (k < nframes && tab.f[k].m == 0) ++k; return k; };
    auto prepare = [&](uint32_t k) -> Q64Frame {                  // geometry of frame k, everything wave-uniform -> SGPRs
        Q64Frame q;
        const FrameDev fd = tab.f[k];
        q.f = k;
        q.m = __builtin_amdgcn_readfirstlane(fd.m);
        q.fk = __builtin_amdgcn_readfirstlane(fd.floor_k);        // floor(k*) | c << 8
        q.fwords = filter_words(q.m);
        q.Thi = 0; q.Tlo = 0;
        q.ninv_lo = __builtin_amdgcn_readfirstlane((uint32_t)fd.M);
        q.ninv_hi = __builtin_amdgcn_readfirstlane((uint32_t)(fd.M >> 32));
        return q;
    };

    uint32_t k = next_active(0);
    if (k >= nframes) return;
    Q64Frame cf = prepare(k);
    const uint32_t lds0 = __builtin_amdgcn_readfirstlane(lds_addr_of(lds));
    RowStager st;
    st.a = st.b = st.c = make_uint4(0, 0, 0, 0);
    st.off0 = wave * 1024u + lane * 16u;
    auto aim = [&](const Q64Frame &q, uint32_t buf) {            // point the stager at frame q's row -> buffer buf
        st.aim(image + (uint64_t)q.f * image_stride_words32, lds0 + buf * bufwords * 4u, ((q.fwords + 3u) & ~3u) * 4u);
    };
    // the first frame's image: staged in one go (once per launch)
    aim(cf, 0u);
    st.template at<AB>(0); st.template at<AB>(1); st.template at<AB>(2); st.template at<AB>(3); st.template at<AB>(4);
    uint32_t cur = 0;
    const uint32_t safe_v = vgpr_copy(safe_pos);

    while (true) {
        if (!(AB & 32)) __syncthreads();          // everyone's writes of buffer cur have landed; nobody probes buffer cur^1 any more
        const uint32_t kn = __builtin_amdgcn_readfirstlane(next_active(cf.f + 1));
        const bool more = kn < nframes;
        Q64Frame nf = cf;
        if (more) nf = prepare(kn);                               // scalar loads, off the critical path
        aim(nf, cur ^ 1u);                                        // no next frame: nf == cf, restaged into the buffer nobody reads any more
        const uint32_t fbase = vgpr_copy(lds0 + cur * bufwords * 4u);
        const uint32_t m_v = vgpr_copy(cf.m);
        const double ninv = __builtin_bit_cast(double, ((uint64_t)cf.ninv_hi << 32) | cf.ninv_lo);
        const uint32_t fk = cf.fk & 0xFFu;
        const uint32_t c_v = vgpr_copy((cf.fk >> 8) & 0xFFu);
        uint32_t pbf = 0, npass = 0;
#define RBF_R64_PASS(FKV, CV) frame_pass_r64<FKV, AB, CV>(hd1, hl1, hd2, hl2, rank_lo, rank_hi, c_v, validmask, fbase, safe_v, m_v, ninv, fk, pbf, npass, st)
        if (whole_wave) {
            switch (fk) {
            case 1: RBF_R64_PASS(1, false); break;
            case 2: RBF_R64_PASS(2, false); break;
            case 3: RBF_R64_PASS(3, false); break;
            case 4: RBF_R64_PASS(4, false); break;
            default: RBF_R64_PASS(-1, false); break;
            }
        } else {                                                  // the frame's last segments: some positions lie past the end
            RBF_R64_PASS(-1, true);
        }
#undef RBF_R64_PASS
        if (!(AB & 64) && live) {
            pass_bytes[((uint64_t)cf.f * nseg + seg) * (QL_SEG_PIXELS / 8) + lane] = (uint8_t)~pbf;
            if (lane == 0) seg_cnt[(uint64_t)cf.f * nseg + seg] = npass;
        }
        if (!more) break;
        cf = nf;
        cur ^= 1u;
    }
}

// ------------------------------------------------------------------------------------------------------------------
// k_query_r64t -- k_query_f64t (rbf_kernels_q64.h: filters that do not fit LDS twice, walked in tiles of one maximal LDS
// buffer) with the same two changes: activation ranks, and the next frame's first tile staged through registers.
// Needs at least 2 KiB + 16 bytes of dynamic LDS (the thresholds' copy), whatever the tile size.
// ------------------------------------------------------------------------------------------------------------------
template <int AB = 0>
__global__ __launch_bounds__(QL_THREADS) void k_query_r64t(
    uint64_t n, uint32_t nframes, const FrameTable tab /* as for k_query_r64: M = bits of -1/m, T = sorted thresholds, floor_k = floor(k*) | c << 8 */, Seeds seeds,
    const uint32_t *__restrict__ image, uint64_t image_stride_words32, uint32_t tile_words /* multiple of 4; tile_words + 4 dwords of LDS */,
    uint32_t *__restrict__ seg_cnt, uint64_t nseg, uint64_t *__restrict__ pass_words, uint32_t any_passthrough /* some frame has m == 0 */)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    // ONE buffer of tile_words dwords + the SAFE dword (kept 0): measured, the LDS-DMA of a tile does not hide under the
    // probes of another one (it adds, see k_query_f64), while every (frame, tile) stage costs ~3 500 cycles of barriers and
    // DMA issue on top of its probes -- so the tiles are as large as LDS allows and there are as few stages as possible
    // (2160p: 2 per frame; double-buffered 76 KB tiles, 4 per frame, were 2.4x slower).
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t nwaves = blockDim.x >> 6;
    const uint64_t seg = (uint64_t)blockIdx.x * nwaves + wave;
    const bool live = seg < nseg;

    // the hashes stay 64-bit integers here (6 registers per pixel instead of 8: this kernel also keeps pos0, step and fail
    // per pixel) and are converted to the FP64 reduction's (double, low dword) form once per frame, not per tile
    uint64_t h1[QL_P], h2[QL_P];
    uint32_t rank_lo = 0, rank_hi = 0;                            // activation ranks, one byte per pixel (rbf_kernels_r64.h)
    uint32_t validmask = 0;
    const uint64_t i0 = seg * QL_SEG_PIXELS + (uint64_t)lane * QL_P;
    {
        uint64_t ha[QL_P];
#pragma unroll
        for (int it = 0; it < QL_P; ++it) {
            h1[it] = 0; h2[it] = 0; ha[it] = ~0ull;
            if (live && i0 + it < n) validmask |= 1u << it;
        }
        if (!hash3_run8((uint32_t)i0, validmask, seeds, h1, h2, ha)) {
#pragma unroll
            for (int it = 0; it < QL_P; ++it) {
                const bool act = (validmask >> it) & 1u;
                const Hash3 h = hash3_index((uint32_t)(i0 + it), act, seeds);
                h1[it] = h.h1; h2[it] = h.h2; ha[it] = h.ha;
            }
        }
        // ranks by binary search over an LDS copy of the sorted thresholds (the tile buffer is still free)
        uint64_t *tl = reinterpret_cast<uint64_t *>(lds);
        if (threadIdx.x < 2u * MAX_BATCH) tl[threadIdx.x] = threadIdx.x < nframes ? tab.f[threadIdx.x < nframes ? threadIdx.x : 0u].T : ~0ull;
        __syncthreads();
        uint32_t top = 1;
        while (2u * top <= nframes) top *= 2u;
        top = __builtin_amdgcn_readfirstlane(top);
        uint32_t r[QL_P];
#pragma unroll
        for (int it = 0; it < QL_P; ++it) r[it] = 0;
        for (uint32_t step_ = top; step_; step_ >>= 1) {
#pragma unroll
            for (int it = 0; it < QL_P; ++it) {
                const uint64_t t = tl[r[it] + step_ - 1u];
                r[it] |= t <= ha[it] ? step_ : 0u;
            }
        }
        rank_lo = r[0] | (r[1] << 8) | (r[2] << 16) | (r[3] << 24);
        rank_hi = r[4] | (r[5] << 8) | (r[6] << 16) | (r[7] << 24);
        __syncthreads();                          // everyone has read the thresholds before the first tile lands on them
    }
    if (threadIdx.x < 4u) lds[tile_words + threadIdx.x] = 0u;    // SAFE (after the search: with tiny tiles the thresholds lay over it)
    // invalid positions (past the end of the frame, or a dead wave) must fail: their verdict bits are forced afterwards
    uint32_t invalid_byte = 0;                                    // bit 7-j: pixel j is not a position of the frame
#pragma unroll
    for (int it = 0; it < QL_P; ++it) invalid_byte |= ((validmask >> it) & 1u) ? 0u : (0x80u >> it);
    uint8_t *pass_bytes = reinterpret_cast<uint8_t *>(pass_words);

    if (any_passthrough)
    for (uint32_t g = 0; g < nframes; ++g) {
        if (tab.f[g].m == 0) {
            if (live && lane == 0) seg_cnt[(uint64_t)g * nseg + seg] = 0;
            if (live) pass_bytes[((uint64_t)g * nseg + seg) * (QL_SEG_PIXELS / 8) + lane] = 0;
        }
    }
    auto next_active = [&](uint32_t k) -> uint32_t { while (k < nframes && tab.f[k].m == 0) ++k; return k; };
    const uint32_t lds_base = __builtin_amdgcn_readfirstlane(lds_addr_of(lds));
    auto stage_dma = [&](uint32_t f, uint32_t fwords, uint32_t t) {                     // tile t of frame f -> LDS
        const uint32_t w0 = t * tile_words;
        const uint32_t words = fwords - w0 < tile_words ? fwords - w0 : tile_words;
        if (!(AB & 8)) dma_row(lds_base, image + (uint64_t)f * image_stride_words32 + w0, words, wave, lane, nwaves);
    };
    uint32_t f = __builtin_amdgcn_readfirstlane(next_active(0));
    if (f >= nframes) return;
    uint32_t fwords = __builtin_amdgcn_readfirstlane(filter_words(tab.f[f].m));
    uint32_t ntiles = (fwords + tile_words - 1) / tile_words;
    const uint32_t fbase = vgpr_copy(__builtin_amdgcn_readfirstlane(lds_addr_of(lds)));

    // Software-pipelined over the frames as k_query_f64t, but the NEXT frame's first tile is staged THROUGH REGISTERS inside the
    // reductions of that frame (three 16-byte pieces in flight per lane, issued before pixel 0 / 2 / 4 / 6 and written two pixels
    // later): a global_load_lds holds its wave ~200 cycles and a wave has up to ten of them per tile -- time it could not spend on
    // the reductions the DMA was supposed to hide under.  The other tiles of a frame have nothing to overlap with and stay DMA.
    uint32_t pos0[QL_P], step[QL_P], fail[QL_P];
    uint32_t notact = 0, m_v = 0, fk = 0;
    struct {
        const uint8_t *row; uint32_t lds_base, last, off0; uint4 a, b, c;
        __device__ __forceinline__ uint32_t off(int i) const { return min(off0 + (uint32_t)i * (QL_WAVES * 1024u), last); }   // clamped: see RowStager
        __device__ __forceinline__ uint4 load(int i) const { return *reinterpret_cast<const uint4 *>(row + off(i)); }
        __device__ __forceinline__ void store(int i, const uint4 &v) const
        { *reinterpret_cast<__attribute__((address_space(3))) r64_u32x4 *>((uintptr_t)(lds_base + off(i))) = r64_u32x4{v.x, v.y, v.z, v.w}; }
        __device__ __forceinline__ void at(int it)          // up to ten pieces per wave (160 KB of LDS / 16 waves)
        {
            if (it == 0) { a = load(0); b = load(1); c = load(2); }
            else if (it == 2) { store(0, a); store(1, b); store(2, c); a = load(3); b = load(4); c = load(5); }
            else if (it == 4) { store(3, a); store(4, b); store(5, c); a = load(6); b = load(7); c = load(8); }
            else if (it == 6) { store(6, a); store(7, b); store(8, c); a = load(9); }
            else if (it == 8) { store(9, a); }
        }
    } st;
    st.off0 = wave * 1024u + lane * 16u;
    st.lds_base = lds_base;
    auto frame_setup = [&](uint32_t ff, uint32_t ff_words, auto staged) {   // geometry scalars + the two reductions of every pixel, once per frame
        const FrameDev fd = tab.f[ff];
        const uint32_t m_s = __builtin_amdgcn_readfirstlane(fd.m);
        const uint32_t fkc = __builtin_amdgcn_readfirstlane(fd.floor_k);
        fk = fkc & 0xFFu;
        const uint32_t c_v = vgpr_copy((fkc >> 8) & 0xFFu);
        m_v = vgpr_copy(m_s);
        // (__builtin_amdgcn_readfirstlane returns int: every half goes through uint32_t, or the low one sign-extends into the high one)
        const uint32_t nhi = __builtin_amdgcn_readfirstlane((uint32_t)(fd.M >> 32)), nlo = __builtin_amdgcn_readfirstlane((uint32_t)fd.M);
        const double ninv = __builtin_bit_cast(double, ((uint64_t)nhi << 32) | nlo);
        if (decltype(staged)::value) {
            const uint32_t words = ff_words < tile_words ? ff_words : tile_words;             // tile 0
            st.row = reinterpret_cast<const uint8_t *>(image + (uint64_t)ff * image_stride_words32);
            st.last = ((words + 3u) & ~3u) * 4u - 16u;
        }
        notact = 0;
#pragma unroll
        for (int it = 0; it < QL_P; ++it) {
            if (decltype(staged)::value && !(AB & 8)) st.at(it);
            // opaque copies: with the setup inlined before the loop and at its end, the compiler would otherwise keep the
            // 16 converted doubles (frame-invariant) live across the whole loop -- 32 registers, 174 dwords of spills
            uint64_t a1 = h1[it], a2 = h2[it];
            asm volatile("" : "+v"(a1), "+v"(a2));
            pos0[it] = mod_m_f64((double)a1, (uint32_t)a1, ninv, m_v);
            step[it] = mod_m_f64((double)a2, (uint32_t)a2, ninv, m_v);
            const uint32_t rk = ((it < 4 ? rank_lo : rank_hi) >> (8 * (it & 3))) & 0xFFu;
            notact |= (rk <= c_v) ? 0u : (1u << it);
            fail[it] = 0;
        }
        if (decltype(staged)::value && !(AB & 8)) st.at(8);
    };
    frame_setup(f, fwords, std::false_type{});
    if (!(AB & 32)) __syncthreads();              // the SAFE dword is in place
    stage_dma(f, fwords, 0);
    while (true) {
        for (uint32_t t = 0; t < ntiles; ++t) {
            if (t) {
                if (!(AB & 32)) __syncthreads();  // the previous tile's probes are done
                stage_dma(f, fwords, t);
            }
            if (!(AB & 32)) {
                dma_wait_all();                   // my share has landed ...
                __syncthreads();                  // ... and everyone's
            }
            const uint32_t w0 = t * tile_words;
            switch (fk) {
            case 1: tile_part_f64<1, AB>(pos0, step, notact, fbase, w0, tile_words, m_v, fk, fail); break;
            case 2: tile_part_f64<2, AB>(pos0, step, notact, fbase, w0, tile_words, m_v, fk, fail); break;
            case 3: tile_part_f64<3, AB>(pos0, step, notact, fbase, w0, tile_words, m_v, fk, fail); break;
            default: tile_part_f64<-1, AB>(pos0, step, notact, fbase, w0, tile_words, m_v, fk, fail); break;
            }
        }
        uint32_t pbf = 0, npass = 0;
#pragma unroll
        for (int it = 0; it < QL_P; ++it) pbf = __builtin_amdgcn_alignbit(pbf, fail[it], 31);
        const uint32_t fnext = __builtin_amdgcn_readfirstlane(next_active(f + 1));
        const uint32_t fwords_next = fnext < nframes ? __builtin_amdgcn_readfirstlane(filter_words(tab.f[fnext].m)) : 0u;
        if (fnext < nframes && !(AB & 32)) __syncthreads();       // this frame's last probes are done: the buffer is free for the next frame's first tile
        // ---- verdicts of the frame
        const uint32_t pb = ~(pbf | invalid_byte) & 0xFFu;
#pragma unroll
        for (int it = 0; it < QL_P; ++it) npass += __popcll(__ballot(((pb >> (7 - it)) & 1u) != 0));
        if (!(AB & 64) && live) {
            pass_bytes[((uint64_t)f * nseg + seg) * (QL_SEG_PIXELS / 8) + lane] = (uint8_t)pb;
            if (lane == 0) seg_cnt[(uint64_t)f * nseg + seg] = npass;
        }
        if (fnext >= nframes) break;
        f = fnext;
        fwords = fwords_next;
        ntiles = (fwords + tile_words - 1) / tile_words;
        frame_setup(f, fwords, std::true_type{});                 // ... which is staged in here
    }
}

} }  // namespace rbf::legacy
