// LEGACY (round 2 / round 3): a frozen copy of csrc/rbf_kernels_s64.h as it stood before the round-4 prune, kept for the old harnesses of tools/
// (bench_query*.hip, bench_insert.hip) and for A/B comparisons against the kernels that replaced these.  NOT part of the library: nothing
// under new_bloom_filter_repo_amd/ includes it.  Everything lives in namespace rbf::legacy.
// rbf_kernels_s64.h -- k_query_s64: the frames-inner FP64 query kernel of round 3 (default for filters of 2^15 <= m < 2^23 bits that
// fit LDS twice: BASELINE config 2).  Same outputs as k_query_r64 / k_query_f64 (pass bytes in numpy.packbits order + per-segment
// pass counts; reference semantics improved_video_compressor.py:116-138, :245-253) and the same arithmetic (hashes once per launch
// as (RN(h), low dword), h mod m through one v_fma_f64, probe image, activation ranks -- rbf_kernels_q64.h / rbf_kernels_r64.h).
//
// What changed, and the measurement behind each change (profiles/r03_query_ablation.txt):
//
//  1. THE FRAME GEOMETRY IS READ FROM LDS, NOT FROM THE KERNEL-ARGUMENT SEGMENT.  k_query_r64's frame loop fetched `tab.f[k]` (m, floor_k,
//     -1/m) with scalar loads once per frame.  The kernarg segment is host-coherent memory: a scalar load that misses the scalar cache
//     is a round trip over the fabric, every 64-byte line of the table (2.7 frames) missed once per scalar cache, the loads of
//     next_active() and prepare() are dependent, and all 16 waves of a workgroup wait for them at the same point.  That -- not VALU
//     issue, not LDS bank conflicts, not the schedule of the probes -- is what the ~2.1 us per frame of the round-2 kernel were made
//     of: with the staging and the barrier removed its time did not move when the reductions (86 of 231 VALU instructions per
//     frame), the LDS probes or the pass counting were taken out (70.9 / 69.4 / 71.3 / 68.8 us), nor when the pass was re-scheduled
//     for ILP or with 15 % fewer instructions.  Here the host hands over a COMPACTED table (entry j = j-th coded frame), the first
//     `nactive` threads copy it into LDS with one vector load each -- one round trip per launch -- and the loop reads its frame's
//     16 bytes with a broadcast ds_read_b128.  No next_active(), no scalar load in the loop.
//  2. The pass is written in ROWS of independent instructions (frame_pass_rows) and the remainder of the FP64 reduction is taken as
//     an exact signed 32-bit number (one v_mad_u64_u32 instead of v_mad_u32_u24 + v_bfe_i32): 214 instead of 231 VALU and 270 instead
//     of 314 instructions per wave and frame, no hazard s_nops, one s_waitcnt per pixel pair instead of six.
//  3. Two staging slots instead of three (8 VGPRs), pass counts from the finished verdict byte (4 ballots per frame instead of 8),
//     verdict / count addresses advanced by one add per frame.
#pragma once
#include "rbf_kernels_r64.h"

namespace rbf { namespace legacy {

constexpr uint32_t S64_GEO_BYTES = MAX_BATCH * 16;                 // LDS behind the two image buffers: 16 bytes of geometry per coded frame

// Two register slots (8 VGPRs) for the next frame's image: a 16-byte piece is loaded in one pixel pair and written to LDS in the
// next.  Pieces per wave: at most five (a buffer of <= 80 KB over 16 waves x 1 KiB).  Branch-free (clamped offsets) for the
// reason given at RowStager (rbf_kernels_r64.h).
struct RowStager2 {
    const uint8_t *row;             // next frame's image row (uniform)
    uint32_t lds_base;              // byte address of the destination buffer (uniform)
    uint32_t last;                  // row bytes - 16 (uniform): the clamp
    uint32_t off0;                  // wave * 1024 + lane * 16
    uint4 a, b;

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
        if (AB & 8192) {                         // staggered: a piece is written two pairs after its load was issued (the first one: one)
            if (g == 0) { a = load(0); b = load(1); }
            else if (g == 1) { store(0, a); a = load(2); }
            else if (g == 2) { store(1, b); b = load(3); }
            else if (g == 3) { store(2, a); a = load(4); }
            else { store(3, b); store(4, a); }
            return;
        }
        if (g == 0) { a = load(0); b = load(1); }
        else if (g == 1) { store(0, a); store(1, b); a = load(2); b = load(3); }
        else if (g == 2) { store(2, a); store(3, b); a = load(4); }
        else if (g == 3) { store(4, a); }
    }
};

// The compare half and the select half of rank_select (rbf_kernels_r64.h) as separate instructions with the lane mask in an SGPR
// pair, so that the two pixels of a pair do not serialise on VCC.
template <int B>
__device__ __forceinline__ uint64_t rank_le(uint32_t ranks, uint32_t c)
{
    uint64_t mask;
    if constexpr (B == 0) asm("v_cmp_le_u32_sdwa %0, %1, %2 src0_sel:BYTE_0 src1_sel:DWORD" : "=s"(mask) : "v"(ranks), "v"(c));
    else if constexpr (B == 1) asm("v_cmp_le_u32_sdwa %0, %1, %2 src0_sel:BYTE_1 src1_sel:DWORD" : "=s"(mask) : "v"(ranks), "v"(c));
    else if constexpr (B == 2) asm("v_cmp_le_u32_sdwa %0, %1, %2 src0_sel:BYTE_2 src1_sel:DWORD" : "=s"(mask) : "v"(ranks), "v"(c));
    else asm("v_cmp_le_u32_sdwa %0, %1, %2 src0_sel:BYTE_3 src1_sel:DWORD" : "=s"(mask) : "v"(ranks), "v"(c));
    return mask;
}
__device__ __forceinline__ uint32_t select_by(uint64_t mask, uint32_t if_clear, uint32_t if_set)
{
    uint32_t r;
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(if_clear), "v"(if_set), "s"(mask));
    return r;
}

__device__ __forceinline__ uint32_t select_or_ones(uint64_t mask, uint32_t if_set)       // mask ? if_set : 0xFFFFFFFF (an inline constant: no register)
{
    uint32_t r;
    asm("v_cndmask_b32_e64 %0, -1, %1, %2" : "=v"(r) : "v"(if_set), "s"(mask));
    return r;
}

#define RBF_ROW() __builtin_amdgcn_sched_barrier(0)

__device__ uint64_t *g_query_rowstamps = nullptr;      // tools/bench_query3.hip, AB & 4096: [first / last wave of workgroup 0][frame][16]

// The two reductions of the two pixels of pair g, as rows of four: x = {pos0, step} of pixel 2g, {pos0, step} of pixel 2g + 1.
// Needs no filter image, so the kernel runs pair 0's IN FRONT of the frame's barrier.
template <int AB>
__device__ __forceinline__ void rows_reduce4(int g, const double (&hd1)[QL_P], const uint32_t (&hl1)[QL_P], const double (&hd2)[QL_P], const uint32_t (&hl2)[QL_P],
                                             uint32_t m /* VGPR */, double ninv, uint32_t (&x)[4])
{
    const int i0 = 2 * g, i1 = 2 * g + 1;
    if (AB & 1) { x[0] = hl1[i0] & 0x7FFFFu; x[1] = hl2[i0] & 0x3FFFFu; x[2] = hl1[i1] & 0x7FFFFu; x[3] = hl2[i1] & 0x3FFFFu; RBF_ROW(); return; }
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
// `after_first_reads()` runs once pair 0's reads are in flight (the kernel puts the previous frame's outputs there).
// AB (ablation mask, tools/bench_query3.hip only; 0 in the library): 1 = no reductions, 2 = no LDS probes, 4 = no pass counting.
template <int FK, int AB, bool OVERLAP = true, typename STAGER, typename HOOK>
__device__ __forceinline__ void frame_pass_rows(
    const double (&hd1)[QL_P], const uint32_t (&hl1)[QL_P], const double (&hd2)[QL_P], const uint32_t (&hl2)[QL_P],
    uint32_t rank_lo, uint32_t rank_hi, uint32_t c /* VGPR */, uint32_t lds_base_bytes /* VGPR */, uint32_t safe_pos /* VGPR */, uint32_t m /* VGPR */, double ninv,
    uint32_t (&x)[4], uint32_t &pbf, STAGER &st, HOOK &&after_first_reads, uint64_t (&ts)[16] /* AB & 4096: s_memtime at 13 points of the pass */)
{
    static_assert(FK >= 1, "at least one deterministic probe");
    constexpr int NP = FK + 1, NG = QL_P / 2;
    // OVERLAP = false (floor(k*) = 5: nearly static frames; 6 would spill): one pair's positions and words at a time -- they are 2 x 2 x 7 registers
    // each otherwise -- and the pair's reads are waited for right behind their issue; the other waves of the SIMD cover them.
    uint32_t pos[OVERLAP ? 2 : 1][2][NP], wrd[OVERLAP ? 2 : 1][2][NP];       // [pair parity][pixel of the pair][probe]
    uint32_t five = 5u;                                            // opaque: written with a literal 5 the compiler folds shift, shift, add into shift, and, add
    asm volatile("" : "+s"(five));
    auto lds_word = [&](uint32_t addr) -> uint32_t {
        return (AB & 2) ? addr * 0x9E3779B1u : *reinterpret_cast<const __attribute__((address_space(3))) uint32_t *>((uintptr_t)addr);
    };
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
        if (!(AB & 2)) __builtin_amdgcn_s_waitcnt(0xC07F);        // lgkmcnt(0), once: left alone the compiler waits in front of each of the six words
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
    // Wave priority falls as the wave advances (3, 2, 1, 0 over the four pairs; 0 until the next barrier).  The SIMD's arbiter serves the
    // highest priority first and, among equals, the OLDEST wave: left alone the four waves of a SIMD run their passes almost one after
    // the other (timeline: the oldest wave's pass takes 2 400 cycles, the youngest's 4 200) and the youngest finishes alone, at the
    // one instruction per ~5 cycles a single wave can issue, while fifteen waves stand at the barrier.  With the priority tied to
    // progress the waves behind are served first and all four finish together.
    // (AB & 4096, tools/bench_query3.hip: the shader clock at 13 points of the pass, read without waiting -- s_memtime returns through
    // lgkmcnt, i.e. with the pass's own waits -- and stored by the kernel after the pass)
#define RBF_STAMP(i) do { if (AB & 4096) { asm volatile("s_memtime %0" : "=s"(ts[i])); RBF_ROW(); } } while (0)
    RBF_STAMP(0);
    if (!(AB & 2048)) __builtin_amdgcn_s_setprio(3);
    steps_and_reads(0, x);
    st.template at<AB>(0);
    RBF_ROW();
    RBF_STAMP(1);
    after_first_reads();
    RBF_ROW();
    RBF_STAMP(2);
#pragma unroll
    for (int g = 1; g < NG; ++g) {
        if (!(AB & 2048)) { if (g == 1) __builtin_amdgcn_s_setprio(2); else if (g == 2) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0); }
        rows_reduce4<AB>(g, hd1, hl1, hd2, hl2, m, ninv, x);      // the reads of pair g - 1 fly under these rows
        RBF_STAMP(3 * g);
        combine2(g - 1);
        RBF_STAMP(3 * g + 1);
        steps_and_reads(g, x);
        st.template at<AB>(g);
        RBF_ROW();
        RBF_STAMP(3 * g + 2);
    }
    combine2(NG - 1);
    st.template at<AB>(4);
    RBF_STAMP(12);
#undef RBF_STAMP
}

// Any floor(k*) and partial waves (positions past the end of the frame must fail): pixel by pixel, probes in a loop.
template <int AB, typename STAGER>
__device__ __forceinline__ void frame_pass_plain(
    const double (&hd1)[QL_P], const uint32_t (&hl1)[QL_P], const double (&hd2)[QL_P], const uint32_t (&hl2)[QL_P],
    uint32_t rank_lo, uint32_t rank_hi, uint32_t c, uint32_t validmask, uint32_t lds_base_bytes, uint32_t safe_pos, uint32_t m, double ninv,
    uint32_t fk, uint32_t &pbf, STAGER &st)
{
#pragma unroll
    for (int it = 0; it < QL_P; ++it) {
        if ((it & 1) == 0) st.template at<AB>(it >> 1);
        uint32_t pos = mod_m_f64(hd1[it], hl1[it], ninv, m);
        const uint32_t step = mod_m_f64(hd2[it], hl2[it], ninv, m);
        uint32_t fail = ~(validmask << (31 - it)) & 0x80000000u;
        for (uint32_t j = 0; j < fk; ++j) {
            fail = (probe_image_word<0>(lds_base_bytes, pos) << (pos & 31u)) | fail;
            const uint32_t s2 = pos + step;
            pos = min(s2, s2 - m);
        }
        const uint32_t rk = ((it < 4 ? rank_lo : rank_hi) >> (8 * (it & 3))) & 0xFFu;
        const uint32_t pc = rk <= c ? pos : safe_pos;
        fail = (probe_image_word<0>(lds_base_bytes, pc) << (pc & 31u)) | fail;
        pbf = __builtin_amdgcn_alignbit(pbf, fail, 31);
    }
    st.template at<AB>(4);
}

// FrameTable as this kernel reads it (host: query_table_s64, rbf_api.hip) -- COMPACTED over the coded frames of the batch:
//   f[j].m, f[j].M = bits of -1.0 / m        of the j-th coded frame,
//   f[j].floor_k = floor(k*) | c << 8 | frame index << 16        (c = coded thresholds below the frame's own, rbf_kernels_r64.h),
//   f[j].T = j-th smallest threshold of the coded frames.
// `empty_lo / empty_hi`: bit f set = frame f of the batch is not coded (m == 0) and this launch writes its (empty) outputs.
// Dynamic LDS: two image buffers of ((fwords_max + 3) & ~3) + 4 dwords, then S64_GEO_BYTES.
// AB bits: 2048 = no wave priorities, 8 = no staging, 16 = no hashing, 32 = no barrier (wrong results), 64 = no output, 256 = frame geometry by scalar loads from
// the kernel-argument segment in every frame (what k_query_r64 does), 1 / 2 / 4 as in frame_pass_rows, 128 = one output store per launch (wrong
// results), 1024 = phase stamps of the frame loop, 4096 = stamps at 13 points of the pass, 8192 = staggered staging (RowStager2).  The library
// instantiates AB = 0 only; every other value exists for tools/bench_query3.hip (profiles/r03_query_ablation.txt).
//
// Two kernels share the body: k_query_s64 (floor(k*) <= 3 in the rows pass; capped at 120 VGPRs, so that with its 16 waves a CU
// keeps 32 registers per SIMD lane free -- exactly one wave of the planar mask kernel (32 VGPRs) or of k_compact_witness (25) per
// SIMD: a neighbour pipeline's mask and compaction kernels then run UNDERNEATH the query instead of queueing behind it, 130 ->
// 126 us per step with four pipelines) and k_query_s64w (rows up to floor(k*) = 5, 127 VGPRs, nothing co-resides), which the host
// picks for batches that contain floor(k*) = 4 or 5.  WIDE = false sends 4 and 5 to the plain pass (correct, slower).
template <int AB, bool WIDE>
__device__ __forceinline__ void query_s64_body(
    uint64_t n, uint32_t nactive, const FrameTable &tab, Seeds seeds,
    const uint32_t *__restrict__ image, uint64_t image_stride_words32, uint32_t fwords_max,
    uint32_t *__restrict__ seg_cnt, uint64_t nseg, uint64_t *__restrict__ pass_words,
    uint4 *__restrict__ table_out /* nullable: write the pixel-index hash table for the NEXT batch's insert kernel */,
    uint64_t empty_lo, uint64_t empty_hi)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    // two buffers; each ends with 4 dwords that the staging never touches, the first of which stays 0 (SAFE); then the geometry
    const uint32_t bufwords = ((fwords_max + 3u) & ~3u) + 4u;
    const uint32_t safe_pos = ((fwords_max + 3u) & ~3u) << 5;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint64_t seg = (uint64_t)blockIdx.x * QL_WAVES + wave;
    const bool live = seg < nseg;
    if (threadIdx.x < 8u) lds[(threadIdx.x >> 2) * bufwords + (bufwords - 4u) + (threadIdx.x & 3u)] = 0u;   // visible after the first barrier
    uint4 *geo = reinterpret_cast<uint4 *>(lds + 2u * bufwords);
    uint64_t *tl = reinterpret_cast<uint64_t *>(lds + bufwords);  // sorted thresholds: buffer 1 is free until the first pass stages into it
    // ONE vector load per thread from the kernel-argument segment (host-coherent memory: a round trip over the fabric) -- the only
    // time the table is read.  Issued here, consumed behind the hashing.
    FrameDev fd_mine{};
    if (threadIdx.x < 2u * MAX_BATCH) fd_mine = tab.f[threadIdx.x < nactive ? threadIdx.x : 0u];

    // ---- frame-independent part: the hashes of my 8 consecutive pixel indices as (double, low dword), and the activation ranks
    static_assert(QL_P == 8, "a lane's verdicts fill one byte; hash3_run8 hashes runs of 8; two rank registers");
    double hd1[QL_P], hd2[QL_P];
    uint32_t hl1[QL_P], hl2[QL_P];
    uint32_t rank_lo = 0, rank_hi = 0;                             // byte it & 3 of (it < 4 ? lo : hi)
    uint32_t validmask = 0;
    const uint64_t i0 = seg * QL_SEG_PIXELS + (uint64_t)lane * QL_P;
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
        if (threadIdx.x < 2u * MAX_BATCH) {
            tl[threadIdx.x] = threadIdx.x < nactive ? fd_mine.T : ~0ull;
            if (threadIdx.x < nactive) geo[threadIdx.x] = make_uint4(fd_mine.m, fd_mine.floor_k, (uint32_t)fd_mine.M, (uint32_t)(fd_mine.M >> 32));
        }
        // ranks = upper_bound of h_act in the sorted thresholds: branch-free binary search over the LDS copy (rbf_kernels_r64.h)
        __syncthreads();
        uint32_t top = 1;                                         // largest power of two <= nactive
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
    }
    const bool whole_wave = __builtin_amdgcn_readfirstlane((uint32_t)__all(validmask == 0xFFu)) != 0u;   // every lane owns 8 positions inside the frame
    uint8_t *pass_bytes = reinterpret_cast<uint8_t *>(pass_words);

    // frames that are not coded: nothing passes (only the frames the host names; none in the common case)
    for (uint32_t half = 0; half < 2; ++half) {
        uint64_t bits = half ? empty_hi : empty_lo;
        while (bits) {
            const uint32_t g = half * 64u + (uint32_t)__builtin_ctzll(bits);
            bits &= bits - 1;
            if (live && lane == 0) seg_cnt[(uint64_t)g * nseg + seg] = 0;
            if (live) pass_bytes[((uint64_t)g * nseg + seg) * (QL_SEG_PIXELS / 8) + lane] = 0;
        }
    }
    if (nactive == 0) return;

    // Geometry of the j-th coded frame: one broadcast LDS read, moved to SGPRs (AB & 256: the round-2 way, scalar loads from the
    // kernarg segment).  Two frames are held: the current one and the next one (whose image the stager fetches).
    struct Geo { uint32_t m, fkc, ninv_lo, ninv_hi; };
    auto geometry_issue = [&](uint32_t j) -> uint4 { return (AB & 256) ? make_uint4(0, 0, 0, 0) : geo[j]; };
    auto geometry_take = [&](uint32_t j, const uint4 &v) -> Geo {
        if (AB & 256) {
            const FrameDev fd = tab.f[j];
            return Geo{(uint32_t)__builtin_amdgcn_readfirstlane(fd.m), (uint32_t)__builtin_amdgcn_readfirstlane(fd.floor_k),
                       (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)fd.M), (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(fd.M >> 32))};
        }
        return Geo{(uint32_t)__builtin_amdgcn_readfirstlane(v.x), (uint32_t)__builtin_amdgcn_readfirstlane(v.y),
                   (uint32_t)__builtin_amdgcn_readfirstlane(v.z), (uint32_t)__builtin_amdgcn_readfirstlane(v.w)};
    };
    const uint32_t lds0 = __builtin_amdgcn_readfirstlane(lds_addr_of(lds));
    RowStager2 st;
    st.a = st.b = make_uint4(0, 0, 0, 0);
    st.off0 = wave * 1024u + lane * 16u;
    auto aim = [&](const Geo &g, uint32_t buf) {                  // point the stager at the image row of the frame with geometry g -> buffer buf (all scalar)
        const uint32_t fw = filter_words(g.m);
        st.row = reinterpret_cast<const uint8_t *>(image + (uint64_t)(g.fkc >> 16) * image_stride_words32);
        st.lds_base = lds0 + buf * bufwords * 4u;
        st.last = ((fw + 3u) & ~3u) * 4u - 16u;
    };
    Geo cg = geometry_take(0, geometry_issue(0));
    // the first frame's image: staged in one go (once per launch)
    aim(cg, 0u);
    st.template at<AB>(0); st.template at<AB>(1); st.template at<AB>(2); st.template at<AB>(3); st.template at<AB>(4);
    uint32_t cur = 0;
    const uint32_t safe_v = vgpr_copy(safe_pos);
    // where my verdict byte / my wave's count of frame f go: base + f * stride
    uint8_t *const pb_lane = pass_bytes + seg * (QL_SEG_PIXELS / 8) + lane;
    const uint64_t pb_stride = nseg * (QL_SEG_PIXELS / 8);
    uint32_t *const cnt_wave = seg_cnt + seg;

    // timeline probe (tools/bench_query3.hip only, AB & 1024): wave 0 and the last wave of the first workgroups stamp the shader clock
    const bool tl_on = (AB & 1024) && blockIdx.x < TL_WGS && (wave == 0 || wave == QL_WAVES - 1) && g_query_timeline;
    uint64_t *tlog = (AB & 1024) && g_query_timeline ? g_query_timeline + ((uint64_t)(blockIdx.x % TL_WGS) * 2 + (wave ? 1 : 0)) * MAX_BATCH * TL_PHASES : nullptr;
    auto stamp = [&](uint32_t slot, uint32_t phase) {
        if ((AB & 1024) && tl_on && lane == 0) tlog[slot * TL_PHASES + phase] = __builtin_readcyclecounter();
    };

    // The outputs of a frame -- its verdict byte and the wave's pass count (popc of the byte per lane, 0..8, then one ballot per bit
    // of that count: 4 compares per frame instead of one per pixel) -- leave DURING THE NEXT FRAME'S PASS, once its first reads are
    // in flight: behind the pass they were ~500 cycles of latency (ballots -> scalar adds -> address -> store) that the last wave
    // of a SIMD ran alone while the other fifteen already stood at the barrier (timeline: profiles/r03_query_timeline.txt).
    uint32_t out_pb = 0, out_f = 0;                                // verdict byte and frame index waiting to be written
    bool out_pending = false;
    auto flush = [&]() {
        if (!out_pending) return;
        uint32_t npass = 0;
        if (!(AB & 4)) {
            const uint32_t cnt = __popc(out_pb);
            npass = __popcll(__ballot((cnt & 1u) != 0)) + 2u * __popcll(__ballot((cnt & 2u) != 0)) + 4u * __popcll(__ballot((cnt & 4u) != 0)) + 8u * __popcll(__ballot((cnt & 8u) != 0));
        }
        if (!(AB & 64) && live) {
            pb_lane[(uint64_t)out_f * pb_stride] = (uint8_t)out_pb;
            if (lane == 0) cnt_wave[(uint64_t)out_f * nseg] = npass;
        }
    };

    for (uint32_t j = 0; j < nactive; ++j) {
        // ---- in front of the barrier: whatever of frame j needs no filter image -- the next frame's geometry, the first pair's reductions
        stamp(j, 0);
        const uint32_t jn = j + 1 < nactive ? j + 1 : j;          // no next frame: this one is restaged into the buffer nobody reads any more
        const uint4 ngv = geometry_issue(jn);                      // (the LDS read flies under the reductions)
        const uint32_t m_v = vgpr_copy(cg.m);
        const double ninv = __builtin_bit_cast(double, ((uint64_t)cg.ninv_hi << 32) | cg.ninv_lo);
        const uint32_t fk = cg.fkc & 0xFFu, f = cg.fkc >> 16;
        const uint32_t c_v = vgpr_copy((cg.fkc >> 8) & 0xFFu);
        const bool rows = whole_wave && fk >= 1u && fk <= (WIDE ? 5u : 3u);       // else: other floor(k*), or the frame's last segments (positions past the end)
        uint32_t x[4] = {0, 0, 0, 0};
        if (rows) rows_reduce4<AB>(0, hd1, hl1, hd2, hl2, m_v, ninv, x);
        const Geo ng = geometry_take(jn, ngv);
        stamp(j, 1);
        if (!(AB & 32)) __syncthreads();          // everyone's writes of buffer cur have landed; nobody probes buffer cur^1 any more
        stamp(j, 2);
        aim(ng, cur ^ 1u);
        const uint32_t fbase = vgpr_copy(lds0 + cur * bufwords * 4u);
        uint32_t pbf = 0;
        uint64_t ts[16] = {};
        if (rows) {
            switch (fk) {
            case 1: frame_pass_rows<1, AB>(hd1, hl1, hd2, hl2, rank_lo, rank_hi, c_v, fbase, safe_v, m_v, ninv, x, pbf, st, flush, ts); break;
            case 2: frame_pass_rows<2, AB>(hd1, hl1, hd2, hl2, rank_lo, rank_hi, c_v, fbase, safe_v, m_v, ninv, x, pbf, st, flush, ts); break;
            case 3: frame_pass_rows<3, AB, WIDE>(hd1, hl1, hd2, hl2, rank_lo, rank_hi, c_v, fbase, safe_v, m_v, ninv, x, pbf, st, flush, ts); break;   // (narrow: single-buffered)
            case 4: if constexpr (WIDE) frame_pass_rows<4, AB>(hd1, hl1, hd2, hl2, rank_lo, rank_hi, c_v, fbase, safe_v, m_v, ninv, x, pbf, st, flush, ts); break;
            default: if constexpr (WIDE) frame_pass_rows<5, AB, false>(hd1, hl1, hd2, hl2, rank_lo, rank_hi, c_v, fbase, safe_v, m_v, ninv, x, pbf, st, flush, ts); break;
            }
        } else {
            flush();
            frame_pass_plain<AB>(hd1, hl1, hd2, hl2, rank_lo, rank_hi, c_v, validmask, fbase, safe_v, m_v, ninv, fk, pbf, st);
        }
        if ((AB & 4096) && g_query_rowstamps && blockIdx.x == 0 && (threadIdx.x == 0 || threadIdx.x == blockDim.x - 64)) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            for (int i = 0; i < 13; ++i) g_query_rowstamps[((threadIdx.x ? 1u : 0u) * MAX_BATCH + j) * 16 + i] = ts[i];
        }
        stamp(j, 3);
        out_pb = ~pbf & 0xFFu; out_f = f; out_pending = true;
        cg = ng;
        cur ^= 1u;
        stamp(j, 4);
        stamp(j, 5);
    }
    flush();
}


#define RBF_S64_PARAMS uint64_t n, uint32_t nactive, const FrameTable tab, Seeds seeds, const uint32_t *__restrict__ image, uint64_t image_stride_words32, \
    uint32_t fwords_max, uint32_t *__restrict__ seg_cnt, uint64_t nseg, uint64_t *__restrict__ pass_words, uint4 *__restrict__ table_out, uint64_t empty_lo, uint64_t empty_hi
#define RBF_S64_ARGS n, nactive, tab, seeds, image, image_stride_words32, fwords_max, seg_cnt, nseg, pass_words, table_out, empty_lo, empty_hi
// (amdgpu_num_vgpr counts HALF of gfx950's unified register file: 60 = 120 VGPRs of the 512 / 4 a wave may have at 4 waves per SIMD)
template <int AB = 0>
__attribute__((amdgpu_num_vgpr(60))) __global__ __launch_bounds__(QL_THREADS) void k_query_s64(RBF_S64_PARAMS) { query_s64_body<AB, false>(RBF_S64_ARGS); }
template <int AB = 0>
__global__ __launch_bounds__(QL_THREADS) void k_query_s64w(RBF_S64_PARAMS) { query_s64_body<AB, true>(RBF_S64_ARGS); }
#undef RBF_S64_PARAMS
#undef RBF_S64_ARGS

// ------------------------------------------------------------------------------------------------------------------
// k_query_s64t -- the same kernel for filters that do not fit LDS twice (1440p ... 5K, m < 2^23: BASELINE config 4), walked in TILES
// of one maximal LDS buffer as k_query_r64t does (rbf_kernels_r64.h), rebuilt on the round-3 cost model: the kernel is VALU-bound at
// four cycles per wave-instruction, and k_query_r64t spent 116 VALU instructions per (pixel, frame) at 2160p x 8 (rocprofv3
// SQ_INSTS_VALU, profiles/r03_rocprofv3_summary_2160p.txt): ~20 on converting its 64-bit hashes to doubles again in every frame,
// ~22 per tile on re-stepping the probe positions.  Here the hashes stay in the (double, low dword) form for the whole launch, a
// frame's probe positions are computed ONCE (rows of four reductions, exact 32-bit remainders) and kept -- floor(k*) + 1 registers
// per pixel -- and a tile costs 5 instructions per probe: word index, distance to the tile's first word, unsigned min against the
// tile length (a probe outside the tile reads the SAFE dword behind it, "bit set"), address, combine.  A pixel whose extra probe
// is not activated gets position 2^32 - 1 for it, which is in no tile.  Verdicts accumulate as one FAIL bit per pixel across the
// tiles.  Staging: LDS-DMA, the first tile issued in steps between the frame's reductions (TileDma), the others between two barriers.
// It does not hide: 2160p x 8 frames measures 150 us without staging and 217 with (tools/bench_query4.hip, profiles/
// r03_query_tiled_ablation.txt) -- 64 (workgroup, tile) stages per CU x 153 KB = 2.5 GB per launch from L2 at 37 TB/s, which is both
// the L2s' aggregate peak (8 XCDs x 16 channels x 128 B/clk) and the CUs' L1 rate (64 B/clk each); staging the first tile through
// registers (TileStager, AB & 4096) measures the same.  Two half-size buffers would hide it but double the tile passes (+40 us).
// Only for batches whose coded frames all have floor(k*) <= S64T_MAX_FK (the kept positions are registers); the host sends
// anything else to k_query_r64t.  Table, outputs and LDS geometry as k_query_s64; LDS: tile_words + 4 dwords, then S64_GEO_BYTES.
// ------------------------------------------------------------------------------------------------------------------
constexpr uint32_t S64T_MAX_FK = 2;
// dwords before the geometry: the tile and its SAFE dwords, or the prologue's copy of the sorted thresholds where a test caps the tile below it
__host__ __device__ constexpr uint32_t s64t_geo_word(uint32_t tile_words) { return tile_words + 4u > 4u * MAX_BATCH ? tile_words + 4u : 4u * MAX_BATCH; }
__host__ constexpr size_t s64t_lds_bytes(uint32_t tile_words) { return (size_t)s64t_geo_word(tile_words) * 4 + S64_GEO_BYTES; }

// Two register slots (8 VGPRs) that carry a frame's FIRST tile into the LDS buffer underneath the frame's reductions: up to ten
// 16-byte pieces per lane (a 160 KB buffer over 16 waves x 1 KiB), two loaded at one step and written at the next.  Clamped offsets instead of branches, as RowStager (rbf_kernels_r64.h).
struct TileStager {
    const uint8_t *row;             // the tile's first byte in the image row (uniform)
    uint32_t lds_base;              // byte address of the buffer (uniform)
    uint32_t last;                  // tile bytes - 16 (uniform): the clamp
    uint32_t off0;                  // wave * 1024 + lane * 16
    uint4 a, b;

    __device__ __forceinline__ uint32_t off(int i) const { return min(off0 + (uint32_t)i * (QL_WAVES * 1024u), last); }
    __device__ __forceinline__ uint4 load(int i) const { return *reinterpret_cast<const uint4 *>(row + off(i)); }
    __device__ __forceinline__ void store(int i, const uint4 &v) const
    {
        *reinterpret_cast<__attribute__((address_space(3))) r64_u32x4 *>((uintptr_t)(lds_base + off(i))) = r64_u32x4{v.x, v.y, v.z, v.w};
    }
    template <int AB>
    __device__ __forceinline__ void at(int g)
    {
        if (AB & 8) return;
        if (g == 0) { a = load(0); b = load(1); }                // step 0: in front of the barrier that frees the buffer (loads only)
        else if (g < 5) { store(2 * g - 2, a); store(2 * g - 1, b); a = load(2 * g); b = load(2 * g + 1); }      // step 1: right behind it; 2..4: pairs 1..3
        else { store(8, a); store(9, b); }
    }
};

// The same through LDS-DMA: no registers, no wait between a piece's load and its LDS write -- each step only ISSUES its pieces
// (four right behind the barrier, two in front of pixels 2, 4 and 6), and the frame waits once, after its reductions.
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
    template <int AB>
    __device__ __forceinline__ void at(int step)
    {
        if (AB & 8) return;
        if (step == 1) { piece(0); piece(1); piece(2); piece(3); }
        else if (step >= 2 && step <= 4) { piece(2 * step); piece(2 * step + 1); }
    }
};

template <int FK, int AB, typename STAGER>
__device__ __forceinline__ void tiled_positions(const double (&hd1)[QL_P], const uint32_t (&hl1)[QL_P], const double (&hd2)[QL_P], const uint32_t (&hl2)[QL_P],
                                                uint32_t rank_lo, uint32_t rank_hi, uint32_t c, uint32_t m, double ninv, uint32_t (&pos)[QL_P][S64T_MAX_FK + 1], STAGER &st)
{
    // Pixel by pixel, two chains (position, step) side by side -- not the rows of four of k_query_s64: four reductions in flight are
    // 16 more live registers, which this kernel does not have (the hashes, the kept positions and the stager's slots: ~105 of 128).
#pragma unroll
    for (int it = 0; it < QL_P; ++it) {
        if (it == 0) st.template at<AB>(0); else if ((it & 1) == 0) st.template at<AB>(it / 2 + 1);
        uint32_t p, stp;
        if (AB & 1) { p = hl1[it] & 0x7FFFFu; stp = hl2[it] & 0x3FFFFu; }
        else {
            const double t0 = __builtin_fma(hd1[it], ninv, 0x1.8p52), t1 = __builtin_fma(hd2[it], ninv, 0x1.8p52);
            const uint32_t s0 = (uint32_t)__builtin_bit_cast(uint64_t, t0) * m + hl1[it], s1 = (uint32_t)__builtin_bit_cast(uint64_t, t1) * m + hl2[it];   // rows_reduce4
            p = min(s0, s0 + m); stp = min(s1, s1 + m);
        }
        if (it == 0) {
            if (!(AB & 32)) __syncthreads();                      // every wave has finished the previous frame's last tile: the buffer is free
            st.template at<AB>(1);
        }
#pragma unroll
        for (int j = 0; j < FK; ++j) {
            pos[it][j] = p;
            const uint32_t u = p + stp;
            p = min(u, u - m);
        }
        const uint32_t rk = it < 4 ? rank_lo : rank_hi;
        const uint64_t k = (it & 3) == 0 ? rank_le<0>(rk, c) : (it & 3) == 1 ? rank_le<1>(rk, c) : (it & 3) == 2 ? rank_le<2>(rk, c) : rank_le<3>(rk, c);
        pos[it][FK] = select_or_ones(k, p);                       // activated, or "in no tile"
    }
}

// One tile's probes of a lane's 8 pixels (pairs, rows of 2 x (FK + 1) probes): the FAIL bits of the tile, MSB-first (bit 7 - it).
template <int FK, int AB>
__device__ __forceinline__ uint32_t tiled_pass(const uint32_t (&pos)[QL_P][S64T_MAX_FK + 1], uint32_t lds_base_bytes, uint32_t tile_word0, uint32_t tile_words)
{
    constexpr int NP = FK + 1;
    uint32_t five = 5u;
    asm volatile("" : "+s"(five));
    uint32_t pbf = 0;
    // One pair's words at a time: the other three waves of the SIMD cover the LDS latency (profiles/r03_query_ablation.txt: the
    // depth of the read pipeline of k_query_s64 does not show in its time), and the registers are needed for the kept positions.
    uint32_t wrd[2][NP];
#pragma unroll
    for (int g = 0; g < QL_P / 2; ++g) {
        // (no wave priorities here: with k_query_s64's progress-tied s_setprio this kernel measured 282 us instead of 214 at 2160p x 8 --
        // the waves still issuing their share of the next tile's LDS-DMA wait behind the ones already probing; tools/bench_query4.hip)
        if (AB & 2048) { if (g == 0) __builtin_amdgcn_s_setprio(3); else if (g == 1) __builtin_amdgcn_s_setprio(2); else if (g == 2) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0); }
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
            wrd[0][j] = (AB & 2) ? aa * 0x9E3779B1u : *reinterpret_cast<const __attribute__((address_space(3))) uint32_t *>((uintptr_t)aa);
            wrd[1][j] = (AB & 2) ? ab * 0x9E3779B1u : *reinterpret_cast<const __attribute__((address_space(3))) uint32_t *>((uintptr_t)ab);
            RBF_ROW();
        }
        uint32_t f0 = 0u, f1 = 0u;
        if (!(AB & 2)) __builtin_amdgcn_s_waitcnt(0xC07F);        // lgkmcnt(0), once per pair
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

// One coded frame of k_query_s64t: the probe positions (no filter needed) with the frame's first tile riding into LDS underneath them
// (the barrier that frees the buffer is inside, after the first pixel), the previous frame's outputs
// (`flush`), then the tiles.  Returns the FAIL bits of the lane's 8 pixels.
template <int FK, int AB, typename FLUSH>
__device__ __forceinline__ uint32_t tiled_frame(const double (&hd1)[QL_P], const uint32_t (&hl1)[QL_P], const double (&hd2)[QL_P], const uint32_t (&hl2)[QL_P],
                                                uint32_t rank_lo, uint32_t rank_hi, uint32_t c_v, uint32_t m_v, double ninv,
                                                const uint32_t *row, uint32_t fwords, uint32_t tile_words, uint32_t lds_base, uint32_t fbase, uint32_t wave, uint32_t lane, FLUSH &&flush)
{
    uint32_t pos[QL_P][S64T_MAX_FK + 1];
    const uint32_t ntiles = (fwords + tile_words - 1) / tile_words;
    const uint32_t words0 = ((fwords < tile_words ? fwords : tile_words) + 3u) & ~3u;
    if (AB & 4096) {                              // tools/bench_query4.hip: the first tile through registers
        TileStager st;
        st.a = st.b = make_uint4(0, 0, 0, 0);
        st.off0 = wave * 1024u + lane * 16u;
        st.lds_base = lds_base;
        st.row = reinterpret_cast<const uint8_t *>(row);
        st.last = words0 * 4u - 16u;
        tiled_positions<FK, AB>(hd1, hl1, hd2, hl2, rank_lo, rank_hi, c_v, m_v, ninv, pos, st);
        st.template at<AB>(5);
    } else {
        TileDma st{row, lds_base, words0 >> 2, wave, lane};
        tiled_positions<FK, AB>(hd1, hl1, hd2, hl2, rank_lo, rank_hi, c_v, m_v, ninv, pos, st);
    }
    flush();
    uint32_t pbf = 0;
    for (uint32_t t = 0; t < ntiles; ++t) {
        const uint32_t w0 = t * tile_words;
        if (t) {                                  // the other tiles have nothing to ride under: LDS-DMA between two barriers
            const uint32_t words = fwords - w0 < tile_words ? fwords - w0 : tile_words;
            if (!(AB & 32)) __syncthreads();      // the previous tile's probes are done
            if (!(AB & 8)) dma_row(lds_base, row + w0, words, wave, lane, QL_WAVES);
            if (!(AB & 32)) dma_wait_all();       // my share has landed ...
        } else if (!(AB & 32)) { if (AB & 4096) __builtin_amdgcn_s_waitcnt(0xC07F); else dma_wait_all(); }      // my pieces of the first tile have landed ...
        if (!(AB & 32)) __syncthreads();          // ... and everyone's
        pbf |= tiled_pass<FK, AB>(pos, fbase, w0, tile_words);
    }
    return pbf;
}

// AB bits (tools/bench_query4.hip only; the library instantiates 0): 1 = no reductions, 2 = no LDS reads, 8 = no staging, 32 = no barriers / waits
// (wrong results), 2048 = k_query_s64's wave priorities in the tile passes, 4096 = the first tile through registers (TileStager) instead of LDS-DMA.
// (120 registers as k_query_s64, for the same reason: one wave of the mask / compaction kernels per SIMD runs underneath it)
template <int AB = 0>
__attribute__((amdgpu_num_vgpr(60))) __global__ __launch_bounds__(QL_THREADS) void k_query_s64t(
    uint64_t n, uint32_t nactive, const FrameTable tab /* as for k_query_s64 */, Seeds seeds,
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
    auto flush = [&]() {                                          // the previous frame's verdict byte and pass count (as k_query_s64)
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
        // meet in 24 phi nodes between them and the register allocator spills the hashes (198 dwords).
        uint32_t pbf;
        switch (fk) {
        case 0: pbf = tiled_frame<0, AB>(hd1, hl1, hd2, hl2, rank_lo, rank_hi, c_v, m_v, ninv, row, fwords, tile_words, lds_base, fbase, wave, lane, flush); break;
        case 1: pbf = tiled_frame<1, AB>(hd1, hl1, hd2, hl2, rank_lo, rank_hi, c_v, m_v, ninv, row, fwords, tile_words, lds_base, fbase, wave, lane, flush); break;
        default: pbf = tiled_frame<2, AB>(hd1, hl1, hd2, hl2, rank_lo, rank_hi, c_v, m_v, ninv, row, fwords, tile_words, lds_base, fbase, wave, lane, flush); break;
        }
        out_pb = ~(pbf | invalid_byte) & 0xFFu; out_f = f; out_pending = true;
    }
    flush();
}

#undef RBF_ROW

} }  // namespace rbf::legacy
