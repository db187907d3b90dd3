// rbf_kernels_s64.h -- the frame pass of the frames-inner FP64 query kernels, and k_query_s64t, the tiled one.
//
//   frame_pass_rows / frame_pass_plain   one frame's probes of a lane's 8 pixels against a probe image in LDS (used by k_query_u64,
//                                        rbf_kernels_u64.h; round 3's k_query_s64 / k_query_s64w, the first kernels built on them,
//                                        are in the git history, tools/legacy/ up to round 4)
//   k_query_s64t                         filters that do not fit LDS twice (1440p ... 5K, m < 2^23: BASELINE config 4), walked in
//                                        tiles of one maximal LDS buffer
//
// Same outputs everywhere: pass bytes in numpy.packbits order + per-segment pass counts (reference semantics
// improved_video_compressor.py:116-138, :245-253); same arithmetic: hashes once per launch as (RN(h), low dword), h mod m through one
// v_fma_f64 (rbf_kernels_q64.h), probe image, activation ranks.
#pragma once
#include "rbf_kernels_q64.h"

namespace rbf {

typedef uint32_t r64_u32x4 __attribute__((ext_vector_type(4)));

#define RBF_ROW() __builtin_amdgcn_sched_barrier(0)

// The two reductions of the two pixels of pair g, as rows of four: x = {pos0, step} of pixel 2g, {pos0, step} of pixel 2g + 1.
// Needs no filter image, so the kernels run pair 0's IN FRONT of the frame's barrier.
__device__ __forceinline__ void rows_reduce4(int g, const double (&hd1)[QL_P], const uint32_t (&hl1)[QL_P], const double (&hd2)[QL_P], const uint32_t (&hl2)[QL_P],
                                             uint32_t m /* VGPR */, double ninv, uint32_t (&x)[4])
{
    const int i0 = 2 * g, i1 = 2 * g + 1;
    const double t0 = __builtin_fma(hd1[i0], ninv, 0x1.8p52), t1 = __builtin_fma(hd2[i0], ninv, 0x1.8p52);
    const double t2 = __builtin_fma(hd1[i1], ninv, 0x1.8p52), t3 = __builtin_fma(hd2[i1], ninv, 0x1.8p52);
    RBF_ROW();
    // r_est = h - q_est * m as an exact SIGNED 32-bit number (mod_m_f64, rbf_kernels_q64.h, derives the same value modulo 2^24 and
    // sign-extends it): the low dword of t is -q_est mod 2^32 (1.5 * 2^52 has no low bits), hl is h mod 2^32, and
    // |r_est| <= 0.75 m < 2^23.  One multiply-add; then the same fold of a negative r_est back into [0, m).
    const uint32_t s0 = (uint32_t)__builtin_bit_cast(uint64_t, t0) * m + hl1[i0], s1 = (uint32_t)__builtin_bit_cast(uint64_t, t1) * m + hl2[i0];
    const uint32_t s2 = (uint32_t)__builtin_bit_cast(uint64_t, t2) * m + hl1[i1], s3 = (uint32_t)__builtin_bit_cast(uint64_t, t3) * m + hl2[i1];
    RBF_ROW();
    const uint32_t q0 = s0 + m, q1 = s1 + m, q2 = s2 + m, q3 = s3 + m;
    RBF_ROW();
    x[0] = min(s0, q0); x[1] = min(s1, q1); x[2] = min(s2, q2); x[3] = min(s3, q3);
    RBF_ROW();
}

// One frame's pass over a lane's 8 pixels, in pixel PAIRS, written in ROWS: a row holds the same instruction of up to four independent
// chains (the two reductions of the two pixels; for the steps: two chains + the address arithmetic of the probes they feed), rows
// are pinned with sched_barrier.  Per pair g:  reductions(g) | combine(g - 1) | steps + addresses + reads(g) | stager(g): the reads
// of pair g - 1 fly under the reductions of pair g.  `x` arrives holding pair 0's reductions (computed in front of the barrier);
// `after_first_reads()` runs once pair 0's reads are in flight (the kernel puts the previous frame's outputs there); `st.at(g)` is
// the kernel's staging hook (k_query_u64 issues the next image's LDS-DMA at g = 0).
// PRIO: wave priority falls as the wave advances (3, 2, 1, 0 over the four pairs; 0 until the next barrier).  The SIMD's arbiter
// serves the highest priority first and, among equals, the OLDEST wave: left alone the four waves of a SIMD run their passes almost
// one after the other and the youngest finishes alone while fifteen waves stand at the barrier (profiles/r03_query_ablation.txt, 6).
template <int FK, bool OVERLAP = true, bool PRIO = true, typename STAGER, typename HOOK>
__device__ __forceinline__ void frame_pass_rows(
    const double (&hd1)[QL_P], const uint32_t (&hl1)[QL_P], const double (&hd2)[QL_P], const uint32_t (&hl2)[QL_P],
    uint32_t rank_lo, uint32_t rank_hi, uint32_t c /* VGPR */, uint32_t lds_base_bytes /* VGPR */, uint32_t safe_pos /* VGPR */, uint32_t m /* VGPR */, double ninv,
    uint32_t (&x)[4], uint32_t &pbf, STAGER &st, HOOK &&after_first_reads)
{
    static_assert(FK >= 1, "at least one deterministic probe");
    constexpr int NP = FK + 1, NG = QL_P / 2;
    // OVERLAP = false (floor(k*) = 5: nearly static frames; 6 would spill): one pair's positions and words at a time -- they are 2 x 2 x 7 registers
    // each otherwise -- and the pair's reads are waited for right behind their issue; the other waves of the SIMD cover them.
    uint32_t pos[OVERLAP ? 2 : 1][2][NP], wrd[OVERLAP ? 2 : 1][2][NP];       // [pair parity][pixel of the pair][probe]
    uint32_t five = 5u;                                            // opaque: written with a literal 5 the compiler folds shift, shift, add into shift, and, add
    asm volatile("" : "+s"(five));
    auto lds_word = [&](uint32_t addr) -> uint32_t { return *reinterpret_cast<const __attribute__((address_space(3))) uint32_t *>((uintptr_t)addr); };
    auto steps_and_reads = [&](int g, const uint32_t (&x)[4]) {    // positions of the pair's probes; every read is issued as soon as its address exists
        const int par = OVERLAP ? g & 1 : 0;
        uint32_t pa = x[0], pb_ = x[2];
        const uint32_t sa = x[1], sb = x[3];
#pragma unroll
        for (int j = 0; j < FK; ++j) {
            pos[par][0][j] = pa; pos[par][1][j] = pb_;
            const uint32_t wa = pa >> five, wb = pb_ >> five;
            const uint32_t ua = pa + sa, ub = pb_ + sb;
            RBF_ROW();
            const uint32_t aa = (wa << 2) + lds_base_bytes, ab = (wb << 2) + lds_base_bytes;
            const uint32_t va = ua - m, vb = ub - m;
            RBF_ROW();
            wrd[par][0][j] = lds_word(aa); wrd[par][1][j] = lds_word(ab);
            pa = min(ua, va); pb_ = min(ub, vb);
            RBF_ROW();
        }
        const int i0 = 2 * g, i1 = 2 * g + 1;
        const uint32_t rk0 = i0 < 4 ? rank_lo : rank_hi, rk1 = i1 < 4 ? rank_lo : rank_hi;
        uint64_t k0, k1;                                           // the pair's activation masks: rank byte <= c
        if ((i0 & 3) == 0) { k0 = rank_le<0>(rk0, c); k1 = rank_le<1>(rk1, c); }
        else { k0 = rank_le<2>(rk0, c); k1 = rank_le<3>(rk1, c); }
        RBF_ROW();
        pos[par][0][FK] = select_by(k0, safe_pos, pa); pos[par][1][FK] = select_by(k1, safe_pos, pb_);   // the activated extra probe, or SAFE
        RBF_ROW();
        const uint32_t wa = pos[par][0][FK] >> five, wb = pos[par][1][FK] >> five;
        RBF_ROW();
        const uint32_t aa = (wa << 2) + lds_base_bytes, ab = (wb << 2) + lds_base_bytes;
        RBF_ROW();
        wrd[par][0][FK] = lds_word(aa); wrd[par][1][FK] = lds_word(ab);
        RBF_ROW();
    };
    auto combine2 = [&](int g) {                                   // verdicts of pair g: the sign bit of `fail` says "some probed filter bit is 0"
        const int par = OVERLAP ? g & 1 : 0;
        uint32_t f0 = 0u, f1 = 0u;
        __builtin_amdgcn_s_waitcnt(0xC07F);                       // lgkmcnt(0), once: left alone the compiler waits in front of each of the six words
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            f0 = (wrd[par][0][j] << (pos[par][0][j] & 31u)) | f0;
            f1 = (wrd[par][1][j] << (pos[par][1][j] & 31u)) | f1;
            RBF_ROW();
        }
        pbf = __builtin_amdgcn_alignbit(pbf, f0, 31);             // (pbf << 1) | (fail >> 31)
        RBF_ROW();
        pbf = __builtin_amdgcn_alignbit(pbf, f1, 31);
        RBF_ROW();
    };
    if (PRIO) __builtin_amdgcn_s_setprio(3);
    steps_and_reads(0, x);
    st.at(0);
    RBF_ROW();
    after_first_reads();
    RBF_ROW();
#pragma unroll
    for (int g = 1; g < NG; ++g) {
        if (PRIO) { if (g == 1) __builtin_amdgcn_s_setprio(2); else if (g == 2) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0); }
        rows_reduce4(g, hd1, hl1, hd2, hl2, m, ninv, x);          // the reads of pair g - 1 fly under these rows
        combine2(g - 1);
        steps_and_reads(g, x);
        st.at(g);
        RBF_ROW();
    }
    combine2(NG - 1);
    st.at(4);
}

// Any floor(k*) and partial waves (positions past the end of the frame must fail): pixel by pixel, probes in a loop.
template <typename STAGER>
__device__ __forceinline__ void frame_pass_plain(
    const double (&hd1)[QL_P], const uint32_t (&hl1)[QL_P], const double (&hd2)[QL_P], const uint32_t (&hl2)[QL_P],
    uint32_t rank_lo, uint32_t rank_hi, uint32_t c, uint32_t validmask, uint32_t lds_base_bytes, uint32_t safe_pos, uint32_t m, double ninv,
    uint32_t fk, uint32_t &pbf, STAGER &st)
{
#pragma unroll
    for (int it = 0; it < QL_P; ++it) {
        if ((it & 1) == 0) st.at(it >> 1);
        uint32_t pos = mod_m_f64(hd1[it], hl1[it], ninv, m);
        const uint32_t step = mod_m_f64(hd2[it], hl2[it], ninv, m);
        uint32_t fail = ~(validmask << (31 - it)) & 0x80000000u;
        for (uint32_t j = 0; j < fk; ++j) {
            fail = (probe_image_word(lds_base_bytes, pos) << (pos & 31u)) | fail;
            const uint32_t s2 = pos + step;
            pos = min(s2, s2 - m);
        }
        const uint32_t rk = ((it < 4 ? rank_lo : rank_hi) >> (8 * (it & 3))) & 0xFFu;
        const uint32_t pc = rk <= c ? pos : safe_pos;
        fail = (probe_image_word(lds_base_bytes, pc) << (pc & 31u)) | fail;
        pbf = __builtin_amdgcn_alignbit(pbf, fail, 31);
    }
    st.at(4);
}

// ------------------------------------------------------------------------------------------------------------------
// k_query_s64t -- filters that do not fit LDS twice, walked in TILES of one maximal LDS buffer.  The kernel is VALU-bound (four cycles
// per wave-instruction for most opcodes, profiles/r04_opbench2.txt) next to a restaging that runs at the L2s' aggregate rate, so it
// spends as few instructions per (pixel, frame, tile) as it can: the hashes stay in the (double, low dword) form for the whole
// launch, a frame's probe positions are computed ONCE (exact 32-bit remainders) and kept -- floor(k*) + 1 registers per pixel --
// and a tile costs 5 instructions per probe: word index, distance to the tile's first word, unsigned min against the tile length (a
// probe outside the tile reads the SAFE dword behind it, "bit set"), address, combine.  A pixel whose extra probe is not activated
// gets position 2^32 - 1 for it, which is in no tile.  Verdicts accumulate as one FAIL bit per pixel across the tiles.
// Kept positions for floor(k*) = 0 ... 4 (round 3: 0 ... 2, everything else fell back to round 2's k_query_r64t, +15 %); any other
// floor(k*) -- nearly static frames -- keeps (first position, step) and walks its probes again in every tile.
// Staging: LDS-DMA, the first tile issued in steps between the frame's reductions (TileDma), the others between two barriers.  It does
// not hide: 2160p x 8 frames measures 150 us without staging and 217 with (profiles/r03_query_tiled_ablation.txt) -- 64 (workgroup,
// tile) stages per CU x 153 KB = 2.5 GB per launch from L2 at 37 TB/s, which is both the L2s' aggregate peak (8 XCDs x 16 channels x
// 128 B/clk) and the CUs' L1 rate (64 B/clk each).  Two half-size buffers would hide it but double the tile passes (+40 us).
// FrameTable as k_query_u64 reads it, without the class order (host: query_table_s64); outputs as k_query_u64.
// LDS: tile_words + 4 dwords (or the prologue's copy of the sorted thresholds where a test caps the tile below it), then S64_GEO_BYTES.
// ------------------------------------------------------------------------------------------------------------------
constexpr uint32_t S64_GEO_BYTES = MAX_BATCH * 16;                 // 16 bytes of geometry per coded frame
constexpr int S64T_MAX_FK = 4;
__host__ __device__ constexpr uint32_t s64t_geo_word(uint32_t tile_words) { return tile_words + 4u > 4u * MAX_BATCH ? tile_words + 4u : 4u * MAX_BATCH; }
__host__ constexpr size_t s64t_lds_bytes(uint32_t tile_words) { return (size_t)s64t_geo_word(tile_words) * 4 + S64_GEO_BYTES; }

// A frame's FIRST tile by LDS-DMA, issued in steps underneath the frame's reductions: no registers, no wait between a piece's load and
// its LDS write -- each step only ISSUES its pieces (four right behind the barrier, two in front of pixels 2, 4 and 6), and the frame
// waits once, after its reductions.
struct TileDma {
    const uint32_t *row;            // the tile's first dword in the image row (uniform)
    uint32_t lds_base;              // byte address of the buffer (uniform)
    uint32_t npieces;               // 16-byte pieces of the tile (uniform; the image rows are padded to whole pieces)
    uint32_t wave, lane;

    __device__ __forceinline__ void piece(uint32_t i) const      // piece row wave + 16 i of the tile: 1 KiB, one 16-byte piece per lane
    {
        const uint32_t c = wave + i * QL_WAVES;
        if ((c << 6) + 64u <= npieces || (c << 6) + lane < npieces) {
            const uint32_t dst = __builtin_amdgcn_readfirstlane(lds_base + (c << 10));
            const uint32_t off = (lane << 4) + (c << 10);
            uint32_t keep;                                        // M0 is saved and restored inside the block (dma_row, rbf_kernels_q64.h)
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %3\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "s"(dst), "v"(off), "s"(row) : "memory");
        }
    }
    __device__ __forceinline__ void at(int step)
    {
        if (step == 1) { piece(0); piece(1); piece(2); piece(3); }
        else if (step >= 2 && step <= 4) { piece(2 * step); piece(2 * step + 1); }
    }
};

// The frame's reductions, pixel by pixel, two chains (position, step) side by side -- not the rows of four of frame_pass_rows: four
// reductions in flight are 16 more live registers, which this kernel does not have.  The barrier that frees the tile buffer sits
// behind the first pixel; the first tile's DMA is issued in steps from there on.
// FK >= 0: pos[it][0 .. FK] = the frame's probe positions of pixel it (the extra one: 2^32 - 1 when not activated).
// FK < 0:  pos[it][0] = first position, pos[it][1] = step; bit it of `act` = extra probe activated.
template <int FK>
__device__ __forceinline__ void tiled_positions(const double (&hd1)[QL_P], const uint32_t (&hl1)[QL_P], const double (&hd2)[QL_P], const uint32_t (&hl2)[QL_P],
                                                uint32_t rank_lo, uint32_t rank_hi, uint32_t c, uint32_t m, double ninv, uint32_t (&pos)[QL_P][FK < 0 ? 2 : FK + 1], uint32_t &act, TileDma &st)
{
    act = 0;
#pragma unroll
    for (int it = 0; it < QL_P; ++it) {
        if ((it & 1) == 0 && it) st.at(it / 2 + 1);
        const double t0 = __builtin_fma(hd1[it], ninv, 0x1.8p52), t1 = __builtin_fma(hd2[it], ninv, 0x1.8p52);
        const uint32_t s0 = (uint32_t)__builtin_bit_cast(uint64_t, t0) * m + hl1[it], s1 = (uint32_t)__builtin_bit_cast(uint64_t, t1) * m + hl2[it];   // rows_reduce4
        uint32_t p = min(s0, s0 + m);
        const uint32_t stp = min(s1, s1 + m);
        if (it == 0) {
            __syncthreads();                                      // every wave has finished the previous frame's last tile: the buffer is free
            st.at(1);
        }
        const uint32_t rk = it < 4 ? rank_lo : rank_hi;
        const uint64_t k = (it & 3) == 0 ? rank_le<0>(rk, c) : (it & 3) == 1 ? rank_le<1>(rk, c) : (it & 3) == 2 ? rank_le<2>(rk, c) : rank_le<3>(rk, c);
        if constexpr (FK < 0) {
            pos[it][0] = p; pos[it][1] = stp;
            act |= select_by(k, 0u, 1u << it);
        } else {
#pragma unroll
            for (int j = 0; j < FK; ++j) {
                pos[it][j] = p;
                const uint32_t u = p + stp;
                p = min(u, u - m);
            }
            pos[it][FK] = select_or_ones(k, p);                   // activated, or "in no tile"
        }
    }
}

// One tile's probes of a lane's 8 pixels (pairs, rows of 2 x (FK + 1) probes): the FAIL bits of the tile, MSB-first (bit 7 - it).
// (No wave priorities here: with frame_pass_rows' progress-tied s_setprio this kernel measured 282 us instead of 214 at 2160p x 8 --
// the waves still issuing their share of the next tile's LDS-DMA wait behind the ones already probing.)
template <int FK>
__device__ __forceinline__ uint32_t tiled_pass(const uint32_t (&pos)[QL_P][FK + 1], uint32_t lds_base_bytes, uint32_t tile_word0, uint32_t tile_words)
{
    constexpr int NP = FK + 1;
    uint32_t five = 5u;
    asm volatile("" : "+s"(five));
    uint32_t pbf = 0;
    // One pair's words at a time: the other three waves of the SIMD cover the LDS latency, and the registers are needed for the kept positions.
    uint32_t wrd[2][NP];
#pragma unroll
    for (int g = 0; g < QL_P / 2; ++g) {
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            const uint32_t wa = pos[2 * g][j] >> five, wb = pos[2 * g + 1][j] >> five;
            RBF_ROW();
            const uint32_t ra = wa - tile_word0, rb = wb - tile_word0;
            RBF_ROW();
            const uint32_t ia = min(ra, tile_words), ib = min(rb, tile_words);        // in this tile, or the SAFE dword behind it
            RBF_ROW();
            const uint32_t aa = (ia << 2) + lds_base_bytes, ab = (ib << 2) + lds_base_bytes;
            RBF_ROW();
            wrd[0][j] = *reinterpret_cast<const __attribute__((address_space(3))) uint32_t *>((uintptr_t)aa);
            wrd[1][j] = *reinterpret_cast<const __attribute__((address_space(3))) uint32_t *>((uintptr_t)ab);
            RBF_ROW();
        }
        uint32_t f0 = 0u, f1 = 0u;
        __builtin_amdgcn_s_waitcnt(0xC07F);                       // lgkmcnt(0), once per pair
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            f0 = (wrd[0][j] << (pos[2 * g][j] & 31u)) | f0;
            f1 = (wrd[1][j] << (pos[2 * g + 1][j] & 31u)) | f1;
            RBF_ROW();
        }
        pbf = __builtin_amdgcn_alignbit(pbf, f0, 31);
        RBF_ROW();
        pbf = __builtin_amdgcn_alignbit(pbf, f1, 31);
        RBF_ROW();
    }
    return pbf;
}

// The same for any floor(k*): (first position, step) kept, the probes walked again in every tile, pixel by pixel.
__device__ __forceinline__ uint32_t tiled_pass_walk(const uint32_t (&pos)[QL_P][2], uint32_t act, uint32_t fk, uint32_t m, uint32_t lds_base_bytes, uint32_t tile_word0, uint32_t tile_words)
{
    uint32_t pbf = 0;
    auto probe = [&](uint32_t p) -> uint32_t {
        const uint32_t i = min((p >> 5) - tile_word0, tile_words);
        return *reinterpret_cast<const __attribute__((address_space(3))) uint32_t *>((uintptr_t)((i << 2) + lds_base_bytes)) << (p & 31u);
    };
#pragma unroll
    for (int it = 0; it < QL_P; ++it) {
        uint32_t p = pos[it][0], fail = 0;
        const uint32_t stp = pos[it][1];
        for (uint32_t j = 0; j < fk; ++j) {
            fail |= probe(p);
            const uint32_t u = p + stp;
            p = min(u, u - m);
        }
        fail |= probe(((act >> it) & 1u) ? p : 0xFFFFFFFFu);
        pbf = __builtin_amdgcn_alignbit(pbf, fail, 31);
    }
    return pbf;
}

// One coded frame: the probe positions (no filter needed) with the frame's first tile riding into LDS underneath them (the barrier
// that frees the buffer is inside, behind the first pixel), the previous frame's outputs (`flush`), then the tiles.  Returns the FAIL
// bits of the lane's 8 pixels.  FK < 0: runtime floor(k*) = fk.
template <int FK, typename FLUSH>
__device__ __forceinline__ uint32_t tiled_frame(const double (&hd1)[QL_P], const uint32_t (&hl1)[QL_P], const double (&hd2)[QL_P], const uint32_t (&hl2)[QL_P],
                                                uint32_t rank_lo, uint32_t rank_hi, uint32_t c_v, uint32_t m_v, double ninv, uint32_t fk,
                                                const uint32_t *row, uint32_t fwords, uint32_t tile_words, uint32_t lds_base, uint32_t fbase, uint32_t wave, uint32_t lane, FLUSH &&flush)
{
    uint32_t pos[QL_P][FK < 0 ? 2 : FK + 1];
    uint32_t act;
    const uint32_t ntiles = (fwords + tile_words - 1) / tile_words;
    const uint32_t words0 = ((fwords < tile_words ? fwords : tile_words) + 3u) & ~3u;
    TileDma st{row, lds_base, words0 >> 2, wave, lane};
    tiled_positions<FK>(hd1, hl1, hd2, hl2, rank_lo, rank_hi, c_v, m_v, ninv, pos, act, st);
    flush();
    uint32_t pbf = 0;
    for (uint32_t t = 0; t < ntiles; ++t) {
        const uint32_t w0 = t * tile_words;
        if (t) {                                  // the other tiles have nothing to ride under: LDS-DMA between two barriers
            const uint32_t words = fwords - w0 < tile_words ? fwords - w0 : tile_words;
            __syncthreads();                      // the previous tile's probes are done
            dma_row(lds_base, row + w0, words, wave, lane, QL_WAVES);
        }
        dma_wait_all();                           // my pieces have landed ...
        __syncthreads();                          // ... and everyone's
        if constexpr (FK < 0) pbf |= tiled_pass_walk(pos, act, fk, m_v, fbase, w0, tile_words);
        else pbf |= tiled_pass<FK>(pos, fbase, w0, tile_words);
    }
    return pbf;
}

// (120 registers: one wave of the mask / compaction kernels per SIMD runs underneath it, as under k_query_u64)
// THREE kernels, because the register allocator sizes a kernel for its hungriest path and the paths do not fit one budget of 128:
//   MODE 0  every coded frame of the launch has floor(k*) 1 or 2 (the host checks): BASELINE config 4 and every other batch at k* ~ 2.3;
//   MODE 1  floor(k*) 0, 1 or 2, each with its own instantiation of the frame (kept positions);
//   MODE 2  any floor(k*): (first position, step) kept, the probes walked again in every tile -- EVERY frame of such a launch takes the
//           walk (7 instead of 5 VALU per probe and tile for the frames that could have kept their positions).
// Round 4 had MODE 1 and 2 (and kept-position instantiations for floor(k*) 3 and 4) in one kernel: 146 dwords spilled, 111 MB of
// scratch writes per 2160p launch.  Measured with -Rpass-analysis=kernel-resource-usage (profiles/r05_kernel_resources.txt): the
// instantiation for 3 kept positions spills by itself next to any other path (42 dwords next to floor(k*) = 0 alone), the walk spills
// next to any kept-position path (70 dwords) and not alone; {0, 1, 2} and {walk} are both spill-free.
template <int MODE>
__attribute__((amdgpu_num_vgpr(60))) __global__ __launch_bounds__(QL_THREADS) void k_query_s64t(
    uint64_t n, uint32_t nactive, const FrameTable tab /* host: query_table_s64 */, Seeds seeds,
    const uint32_t *__restrict__ image, uint64_t image_stride_words32, uint32_t tile_words /* multiple of 4 */,
    uint32_t *__restrict__ seg_cnt, uint64_t nseg, uint64_t *__restrict__ pass_words, uint64_t empty_lo, uint64_t empty_hi)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    const uint32_t lane = threadIdx.x & 63u, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);      // (scalar: so are seg and live)
    const uint64_t seg = (uint64_t)blockIdx.x * QL_WAVES + wave;
    const bool live = seg < nseg;
    uint4 *geo = reinterpret_cast<uint4 *>(lds + s64t_geo_word(tile_words));
    uint64_t *tl = reinterpret_cast<uint64_t *>(lds);              // sorted thresholds: the tile buffer is free until the first tile lands
    FrameDev fd_mine{};
    if (threadIdx.x < 2u * MAX_BATCH) fd_mine = tab.f[threadIdx.x < nactive ? threadIdx.x : 0u];

    double hd1[QL_P], hd2[QL_P];
    uint32_t hl1[QL_P], hl2[QL_P];
    uint32_t rank_lo = 0, rank_hi = 0;
    uint32_t validmask = 0;
    const uint64_t i0 = seg * QL_SEG_PIXELS + (uint64_t)lane * QL_P;
    {
        uint64_t h1[QL_P], h2[QL_P], ha[QL_P];
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
#pragma unroll
        for (int it = 0; it < QL_P; ++it) {
            hd1[it] = (double)h1[it]; hl1[it] = (uint32_t)h1[it];
            hd2[it] = (double)h2[it]; hl2[it] = (uint32_t)h2[it];
            asm volatile("" : "+v"(hd1[it]), "+v"(hd2[it]));      // converted HERE: the 64-bit forms die before the search (else they spill)
        }
        if (threadIdx.x < 2u * MAX_BATCH) {
            tl[threadIdx.x] = threadIdx.x < nactive ? fd_mine.T : ~0ull;
            if (threadIdx.x < nactive) geo[threadIdx.x] = make_uint4(fd_mine.m, fd_mine.floor_k, (uint32_t)fd_mine.M, (uint32_t)(fd_mine.M >> 32));
        }
        __syncthreads();
        uint32_t top = 1;
        while (2u * top <= nactive) top *= 2u;
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
        __syncthreads();                          // everyone has read the thresholds before the first tile lands on them
    }
    if (threadIdx.x < 4u) lds[tile_words + threadIdx.x] = 0u;    // SAFE (after the search: the thresholds lay over the buffer)
    uint32_t invalid_byte = 0;                                    // bit 7-j: pixel j is not a position of the frame -> must fail
#pragma unroll
    for (int it = 0; it < QL_P; ++it) invalid_byte |= ((validmask >> it) & 1u) ? 0u : (0x80u >> it);
    uint8_t *pass_bytes = reinterpret_cast<uint8_t *>(pass_words);

    for (uint32_t half = 0; half < 2; ++half) {                   // frames that are not coded: nothing passes
        uint64_t bits = half ? empty_hi : empty_lo;
        while (bits) {
            const uint32_t g = half * 64u + (uint32_t)__builtin_ctzll(bits);
            bits &= bits - 1;
            if (live && lane == 0) seg_cnt[(uint64_t)g * nseg + seg] = 0;
            if (live) pass_bytes[((uint64_t)g * nseg + seg) * (QL_SEG_PIXELS / 8) + lane] = 0;
        }
    }
    if (nactive == 0) return;

    const uint32_t lds_base = __builtin_amdgcn_readfirstlane(lds_addr_of(lds));
    const uint32_t fbase = vgpr_copy(lds_base);
    const uint64_t pb_stride = nseg * (QL_SEG_PIXELS / 8);
    uint32_t out_pb = 0, out_f = 0;
    bool out_pending = false;
    auto flush = [&]() {                                          // the previous frame's verdict byte and pass count: popc of the byte per lane, then one ballot per bit of that count
        if (!out_pending) return;
        const uint32_t cnt = __popc(out_pb);
        const uint32_t npass = __popcll(__ballot((cnt & 1u) != 0)) + 2u * __popcll(__ballot((cnt & 2u) != 0)) + 4u * __popcll(__ballot((cnt & 4u) != 0)) + 8u * __popcll(__ballot((cnt & 8u) != 0));
        if (live) {                                               // (addresses rebuilt here: two 64-bit pointers per lane would not fit beside the positions)
            pass_bytes[(uint64_t)out_f * pb_stride + seg * (QL_SEG_PIXELS / 8) + lane] = (uint8_t)out_pb;
            if (lane == 0) seg_cnt[(uint64_t)out_f * nseg + seg] = npass;
        }
    };

    for (uint32_t j = 0; j < nactive; ++j) {
        const uint4 gv = geo[j];
        const uint32_t m_s = __builtin_amdgcn_readfirstlane(gv.x), fkc = __builtin_amdgcn_readfirstlane(gv.y);
        const uint32_t m_v = vgpr_copy(m_s);
        const double ninv = __builtin_bit_cast(double, ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane(gv.w) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane(gv.z));
        const uint32_t fk = fkc & 0xFFu, f = fkc >> 16;
        const uint32_t c_v = vgpr_copy((fkc >> 8) & 0xFFu);
        const uint32_t fwords = filter_words(m_s);
        const uint32_t *row = image + (uint64_t)f * image_stride_words32;
        // One instantiation of the whole frame per floor(k*): with the switch around the two halves instead, the kept positions
        // meet in phi nodes between them and the register allocator spills the hashes.
        uint32_t pbf;
#define RBF_S64T_FRAME(FKV) tiled_frame<FKV>(hd1, hl1, hd2, hl2, rank_lo, rank_hi, c_v, m_v, ninv, fk, row, fwords, tile_words, lds_base, fbase, wave, lane, flush)
        if constexpr (MODE == 2) {
            pbf = RBF_S64T_FRAME(-1);
        } else if constexpr (MODE == 1) {
            if (fk == 0) pbf = RBF_S64T_FRAME(0); else if (fk == 1) pbf = RBF_S64T_FRAME(1); else pbf = RBF_S64T_FRAME(2);
        } else {
            if (fk == 1) pbf = RBF_S64T_FRAME(1); else pbf = RBF_S64T_FRAME(2);
        }
#undef RBF_S64T_FRAME
        out_pb = ~(pbf | invalid_byte) & 0xFFu; out_f = f; out_pending = true;
    }
    flush();
}

#undef RBF_ROW

}  // namespace rbf
