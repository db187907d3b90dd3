// LEGACY (round 2 / round 3): a frozen copy of csrc/rbf_kernels_q64.h as it stood before the round-4 prune, kept for the old harnesses of tools/
// (bench_query*.hip, bench_insert.hip) and for A/B comparisons against the kernels that replaced these.  NOT part of the library: nothing
// under new_bloom_filter_repo_amd/ includes it.  Everything lives in namespace rbf::legacy.
// rbf_kernels_q64.h -- k_query_f64: the frames-inner query kernel for filters of 2^15 <= m < 2^23 bits that fit
// LDS twice (1080p-class frames: BASELINE config 2).  Same outputs as k_query_lds (pass bytes in numpy.packbits
// order + per-segment pass counts; reference semantics improved_video_compressor.py:116-138, :245-253), rebuilt
// around what the round-2 measurements showed (profiles/r02_opbench.txt, profiles/r02_query_timeline.txt):
//
//  * VALU cost on gfx950 is per opcode CLASS, not per instruction: v_add / v_sub / v_and / v_or / v_xor with VGPR or
//    inline-constant operands issue in ~1.2 cycles per wave; shifts, min, bfe, mad, cndmask, alignbit, fma (f32 and
//    f64 alike) and ANY opcode with an SGPR source in ~3.2; compares in ~4.  32-bit multiplies are not special.
//    => h mod m goes through ONE v_fma_f64 + ONE v_mad_u32_u24 (mod_m_f64) instead of four multiplies + fix-ups, the
//       wave-uniform m / LDS base live in VGPRs, and probes test a PROBE IMAGE (~bswap of the packed filter) so a
//       probe is shift / and / add + ds_read + one v_lshl_or.
//  * Timeline stamps inside the frame loop (tools/bench_query.hip, profiles/r02_query_timeline.txt): staging a 76 KB
//    filter costs the CU ~1 200 cycles whichever way it is issued (tools/dmabench.hip: LDS-DMA and global_load +
//    ds_write both land 76 KB in 1 150 - 1 550 cycles, ~50 B/clk/CU; one wave alone needs 160 cycles per 1 KiB
//    instruction), and that time ADDS to the frame passes instead of hiding under them: 86 us without the DMA,
//    101 us with it.  Variants measured and NOT kept because they changed nothing or lost (profiles/r02_query_ablation.txt
//    and the round-2 git history have them): issuing the DMA share of wave group g in front of pixel part g of the frame
//    pass (2 / 4 parts: 100 - 108 us vs 101), storing the verdicts one barrier late (+7 us), counting passes with popc +
//    a wave reduction instead of ballots (+4 us), staging through registers -- one plain 16-byte load per pipeline slot,
//    the ds_write_b128 a slot later -- instead of LDS-DMA (125 us: it needs four more VGPRs than there are; k_query_r64,
//    rbf_kernels_r64.h, frees them with activation ranks and IS the default FP64 query kernel now: 92 us), and a
//    4-pixels-per-lane re-cut at 8 waves per SIMD (k_query_p4 below, 112 us).
//  * The next frame's geometry (scalar loads; -1/m comes from the host in FrameDev::M) is fetched one frame ahead.
#pragma once
#include "rbf_kernels_lds.h"

namespace rbf { namespace legacy {

// ---- h mod m through the FP64 pipe (2^15 <= m < 2^23) --------------------------------------------------------
// With hd = RN(h) as a double (frame-independent, computed once per pixel next to the hash) and ninv = -1/m:
//     t = fma(hd, ninv, 1.5 * 2^52)  ->  t = 1.5 * 2^52 - q_est,  q_est = RN(h/m + d),  |d| < 2^-3:
//         |RN(h) - h| / m          <= 2^10 / 2^15 = 2^-5      (h < 2^64 is rounded to 53 bits: half an ulp of 2^11), plus
//         (h/m) * |rel. error of RN(-1/m)|  <  2^49 * 2^-53 = 2^-4
// (h/m < 2^49 because m >= 2^15; t lies in [2^52, 2^53), where doubles are integers), so q_est is floor(h/m) or
// floor(h/m) + 1 and r_est = h - q_est * m lies in [-0.75 m, 0.75 m].  The low dword of t's mantissa is -q_est mod 2^32;
// only r_est mod 2^24 is needed (|r_est| < 2^23 as m < 2^23), and that depends only on the low 24 bits of q_est, m
// and h: ONE v_mad_u32_u24 computes (-q_est * m + h_lo) mod 2^24, v_bfe_i32 sign-extends it, and one add +
// unsigned min folds a negative r_est back into [0, m).  Exactness is checked against integer arithmetic on the host
// (tests/c/mod_f64_check.c restates these five steps in C) and by the GPU parity tests.
// `m` arrives in a VGPR on purpose (vgpr_copy): see the opcode classes above.
__device__ __forceinline__ uint32_t mod_m_f64(double hd, uint32_t hl, double ninv, uint32_t m)
{
    const double t = __builtin_fma(hd, ninv, 0x1.8p52);
    const uint32_t nq = (uint32_t)__builtin_bit_cast(uint64_t, t);
    uint32_t r;
    asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(r) : "v"(nq), "v"(m), "v"(hl));     // the compiler would pick v_mad_u64_u32 (it sees only 24 demanded bits)
    const uint32_t rs = (uint32_t)(((int32_t)(r << 8)) >> 8);                     // v_bfe_i32 r, 0, 24
    return min(rs, rs + m);
}
constexpr uint32_t F64MOD_M_MIN = 1u << 15, F64MOD_M_MAX = (1u << 23) - 1u;      // eligible filter sizes (host: make_plan)

__device__ __forceinline__ uint32_t vgpr_copy(uint32_t uniform)
{
    uint32_t v;
    asm volatile("v_mov_b32 %0, %1" : "=v"(v) : "s"(uniform));                    // volatile: must not be folded back into an SGPR operand
    return v;
}

// The kernel probes a PROBE IMAGE of the filter: dword w of the image is ~bswap(packed dword w), i.e. stream bit i of
// the dword sits at bit 31 - i, inverted.  A probe is then
//     fail = (image[pos >> 5] << (pos & 31)) | fail          (v_lshl_or_b32: the shifter takes pos's low 5 bits itself)
// and the sign bit of `fail` says "some probed filter bit is 0" -- no xor for the MSB-first bit order, no and-tree.
// The image is written by k_filter_reduce next to the packed filter (encode) or by k_probe_image (decode).
// The activated extra probe is made unconditional by steering the non-activated pixels to SAFE, a dword past the
// filter that the kernel keeps 0 in both LDS buffers ("bit set"): one v_cndmask instead of a masked merge.
template <int AB = 0>
__device__ __forceinline__ uint32_t probe_image_word(uint32_t lds_base_bytes /* in a VGPR */, uint32_t pos)
{
    uint32_t addr;
    if (!(AB & 8192)) {                                          // two instructions: v_lshrrev_b32 + v_lshl_add_u32
        uint32_t w;
        asm("v_lshrrev_b32 %0, 5, %1" : "=v"(w) : "v"(pos));     // opaque, or the compiler rewrites it as shift / and / add
        addr = (w << 2) + lds_base_bytes;
    } else
        addr = ((pos >> 3) & ~3u) + lds_base_bytes;
    return *reinterpret_cast<const __attribute__((address_space(3))) uint32_t *>((uintptr_t)addr);
}

// One frame's pass over pixels [IT0, IT1) of a lane's QL_P pixels, software-pipelined by hand in groups of G pixels:
//     positions(g) -> issue the group's LDS reads -> positions(g+1) -> combine(g) -> reads(g+1) ...
// Left to itself hipcc emitted the pass pixel by pixel -- 13 VALU, three ds_reads, s_waitcnt, combine -- i.e. eight
// exposed LDS round trips per frame and wave (profiles/r02_query_schedule.txt), which is what kept the VALU at ~25 % of
// its issue rate with four waves per SIMD.  Here the reads of a group are in flight while the positions of the next
// group are computed; __builtin_amdgcn_sched_barrier pins the phase order.
// pbf accumulates the lane's FAIL bits MSB-first (after all parts: bit 7-j = pixel j failed), npass the wave's number
// of passing positions.  CHECK_VALID: lanes may own positions past the end of the frame (validmask), which must fail;
// the common whole-wave case skips that.
// AB (ablation mask, tools/bench_query.hip only; 0 in the library): 1 = no reductions, 2 = no LDS probes, 4 = no ballot.
template <int FK, int AB, bool CHECK_VALID, int IT0, int IT1>
__device__ __forceinline__ void frame_part_f64(
    const double (&hd1)[QL_P], const uint32_t (&hl1)[QL_P], const double (&hd2)[QL_P], const uint32_t (&hl2)[QL_P],
    const uint64_t (&ha)[QL_P], uint32_t validmask, uint32_t lds_base_bytes, uint32_t safe_pos, uint32_t m, double ninv, uint64_t T,
    uint32_t fk_rt, uint32_t &pbf, uint32_t &npass)
{
    if constexpr (FK < 0) {                                       // runtime floor(k*): rare geometries, plain loop
#pragma unroll
        for (int it = IT0; it < IT1; ++it) {
            uint32_t pos = mod_m_f64(hd1[it], hl1[it], ninv, m);
            const uint32_t step = mod_m_f64(hd2[it], hl2[it], ninv, m);
            uint32_t fail = CHECK_VALID ? ~(validmask << (31 - it)) & 0x80000000u : 0u;
            for (uint32_t j = 0; j < fk_rt; ++j) {
                fail = (probe_image_word<AB>(lds_base_bytes, pos) << (pos & 31u)) | fail;
                const uint32_t s2 = pos + step;
                pos = min(s2, s2 - m);
            }
            const uint32_t pc = (ha[it] < T) ? pos : safe_pos;
            fail = (probe_image_word<AB>(lds_base_bytes, pc) << (pc & 31u)) | fail;
            pbf = __builtin_amdgcn_alignbit(pbf, fail, 31);
            if (!(AB & 4)) npass += __popcll(__ballot((int32_t)fail >= 0));
        }
    } else {
        constexpr int G = 2, NG = (IT1 - IT0) / G, NP = FK + 1;
        static_assert((IT1 - IT0) % G == 0, "whole groups");
        uint32_t pos[NG][G][NP], wrd[NG][G][NP];
        auto positions = [&](int g) {
#pragma unroll
            for (int e = 0; e < G; ++e) {
                const int it = IT0 + g * G + e;
                uint32_t p, step;
                if (AB & 1) { p = hl1[it] & 0x7FFFFu; step = hl2[it] & 0x3FFFFu; }
                else { p = mod_m_f64(hd1[it], hl1[it], ninv, m); step = mod_m_f64(hd2[it], hl2[it], ninv, m); }
#pragma unroll
                for (int j = 0; j < FK; ++j) {
                    pos[g][e][j] = p;
                    const uint32_t s2 = p + step;
                    p = min(s2, s2 - m);
                }
                pos[g][e][FK] = (ha[it] < T) ? p : safe_pos;      // the activated extra probe, or SAFE
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
                const int it = IT0 + g * G + e;
                uint32_t fail = CHECK_VALID ? ~(validmask << (31 - it)) & 0x80000000u : 0u;
#pragma unroll
                for (int j = 0; j < NP; ++j) fail = (wrd[g][e][j] << (pos[g][e][j] & 31u)) | fail;
                pbf = __builtin_amdgcn_alignbit(pbf, fail, 31);   // (pbf << 1) | (fail >> 31)
                if (!(AB & 4)) npass += __popcll(__ballot((int32_t)fail >= 0));
            }
        };
        // P0 L0 | P1 C0 L1 | P2 C1 L2 | ... | C(last): the reads of group g fly while the positions of group g+1 are computed
        positions(0);
        __builtin_amdgcn_sched_barrier(0);
        loads(0);
#pragma unroll
        for (int g = 1; g < NG; ++g) {
            __builtin_amdgcn_sched_barrier(0);
            positions(g);
            __builtin_amdgcn_sched_barrier(0);
            combine(g - 1);
            __builtin_amdgcn_sched_barrier(0);
            loads(g);
        }
        __builtin_amdgcn_sched_barrier(0);
        combine(NG - 1);
    }
}

// The pixel-index hash table (32 bytes per index: RN(h1), RN(h2) as doubles | low dwords of h1, h2 | h_act) that
// k_insert_tab gathers from (rbf_kernels_i64.h).  Inside a 512-index segment the entries are stored SLOT-MAJOR -- the
// entry of index seg * 512 + lane * 8 + it sits at seg * 512 + it * 64 + lane -- because that is the order in which the
// kernels that produce it (a lane owns 8 consecutive indices) can store it with fully coalesced 2 KiB wave stores.
__device__ __forceinline__ uint32_t hash_table_slot(uint32_t index)
{
    return (index & ~511u) | ((index & 7u) << 6) | ((index >> 3) & 63u);
}
__device__ __forceinline__ void hash_table_store(uint4 *__restrict__ table, uint64_t seg, uint32_t lane, int it, uint64_t h1, uint64_t h2, uint64_t ha)
{
    const uint64_t d1 = __builtin_bit_cast(uint64_t, (double)h1), d2 = __builtin_bit_cast(uint64_t, (double)h2);
    uint4 *e = table + 2 * (seg * QL_SEG_PIXELS + (uint32_t)it * 64u + lane);
    e[0] = make_uint4((uint32_t)d1, (uint32_t)(d1 >> 32), (uint32_t)d2, (uint32_t)(d2 >> 32));
    e[1] = make_uint4((uint32_t)h1, (uint32_t)h2, (uint32_t)ha, (uint32_t)(ha >> 32));
}

// dma_filter (rbf_kernels_lds.h) costs ~30 instructions per 1 KiB piece -- M0 saved and restored, a 64-bit address per lane,
// the bounds test -- and a wave issues five pieces per frame: ~150 of the ~520 instructions it executes per frame went into
// ISSUING the staging (ISA count, profiles/r02_query_isa.txt).  This form keeps the row pointer in an SGPR pair (saddr
// addressing: the VGPR holds a 32-bit byte offset) and tests bounds only on the row's last piece: 6 instructions per piece
// (M0 is saved and restored inside the asm block; round 2 listed it as a clobber, which the compiler rejects as reserved).
__device__ __forceinline__ void dma_row(uint32_t lds_byte_addr /* uniform */, const uint32_t *row /* uniform */, uint32_t words, uint32_t wave, uint32_t lane, uint32_t nwaves)
{
    const uint32_t npieces = words >> 2;                          // whole 16-byte pieces
    const uint32_t lane_off = lane << 4;
    for (uint32_t c = wave; (c << 6) < npieces; c += nwaves) {
        const uint32_t dst = __builtin_amdgcn_readfirstlane(lds_byte_addr + (c << 10));
        const uint32_t off = lane_off + (c << 10);
        if ((c << 6) + 64u <= npieces || (c << 6) + lane < npieces) {
            uint32_t keep;                                        // M0 is saved and restored inside the block: it is a reserved register, not a clobber
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %3\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "s"(dst), "v"(off), "s"(row) : "memory");
        }
    }
    const uint32_t tail = words & 3u;                             // 0..3 dwords left: 4-byte DMA by wave 0
    if (wave == 0 && lane < tail) {
        const uint32_t dst = __builtin_amdgcn_readfirstlane(lds_byte_addr + (npieces << 4));
        const uint32_t off = (npieces << 4) + (lane << 2);
        uint32_t keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %2, %3\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "s"(dst), "v"(off), "s"(row) : "memory");
    }
}

// Timeline probe (tools/bench_query.hip only, AB & 1024): wave 0 and the last wave of the first workgroups record the
// shader clock at the phases of every frame iteration into this buffer ([wg][wave 0 / last][frame][phase]).
__device__ uint64_t *g_query_timeline = nullptr;
constexpr uint32_t TL_WGS = 4, TL_PHASES = 6;


// Per-frame scalars, prepared one frame ahead.
struct Q64Frame {
    uint32_t m, fk, fwords, f;
    uint32_t Thi, Tlo, ninv_lo, ninv_hi;
};

// AB bits also understood here: 8 = no filter DMA, 16 = no hashing, 32 = no barrier / DMA wait (wrong results), 64 = no output,
// 1024 = timeline stamps, 8192 = three-instruction probe address (shift, and, add).
template <int AB = 0>
__global__ __launch_bounds__(QL_THREADS) void k_query_f64(
    uint64_t n, uint32_t nframes, const FrameTable tab, Seeds seeds,
    const uint32_t *__restrict__ image, uint64_t image_stride_words32, uint32_t fwords_max,
    uint32_t *__restrict__ seg_cnt, uint64_t nseg, uint64_t *__restrict__ pass_words,
    uint4 *__restrict__ table_out /* nullable: write the hash table of the frame geometry for the NEXT batch's insert kernel */)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    // two buffers; each ends with 4 dwords that no DMA touches, the first of which stays 0 (SAFE)
    const uint32_t bufwords = ((fwords_max + 3u) & ~3u) + 4u;
    const uint32_t safe_pos = ((fwords_max + 3u) & ~3u) << 5;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t nwaves = blockDim.x >> 6;
    const uint64_t seg = (uint64_t)blockIdx.x * nwaves + wave;
    const bool live = seg < nseg;
    const uint64_t base = seg * QL_SEG_PIXELS;
    if (threadIdx.x < 8u) lds[(threadIdx.x >> 2) * bufwords + (bufwords - 4u) + (threadIdx.x & 3u)] = 0u;   // visible after the first barrier

    // ---- frame-independent part: the three hashes of my 8 consecutive pixel indices, as (double, low dword) ------
    static_assert(QL_P == 8, "a lane's verdicts fill one byte; hash3_run8 hashes runs of 8");
    double hd1[QL_P], hd2[QL_P];
    uint32_t hl1[QL_P], hl2[QL_P];
    uint64_t ha[QL_P];
    uint32_t validmask = 0;
    const uint64_t i0 = base + (uint64_t)lane * QL_P;
    {
        uint64_t h1[QL_P], h2[QL_P];
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
        // The hashes depend on the index and the seeds only, and this kernel has just computed them for every index of
        // the frame: they are handed to the next batch's insert kernel (same geometry, same values) instead of being
        // computed a second time by k_hash_table.  64 lanes x 32 bytes per store pair: whole 2 KiB runs.
        if (table_out && live) {
#pragma unroll
            for (int it = 0; it < QL_P; ++it) hash_table_store(table_out, seg, lane, it, h1[it], h2[it], ha[it]);
        }
    }
    const bool whole_wave = __builtin_amdgcn_readfirstlane((uint32_t)__all(validmask == 0xFFu)) != 0u;   // every lane owns 8 positions inside the frame
    uint8_t *pass_bytes = reinterpret_cast<uint8_t *>(pass_words);

    // passthrough frames (m == 0): nothing passes
    for (uint32_t g = 0; g < nframes; ++g) {
        if (tab.f[g].m == 0) {
            if (live && lane == 0) seg_cnt[(uint64_t)g * nseg + seg] = 0;
            if (live) pass_bytes[((uint64_t)g * nseg + seg) * (QL_SEG_PIXELS / 8) + lane] = 0;
        }
    }
    auto next_active = [&](uint32_t k) -> uint32_t { while (k < nframes && tab.f[k].m == 0) ++k; return k; };
    // geometry of frame k (k < nframes), everything wave-uniform -> SGPRs
    auto prepare = [&](uint32_t k) -> Q64Frame {
        Q64Frame q;
        const FrameDev fd = tab.f[k];
        q.f = k;
        q.m = __builtin_amdgcn_readfirstlane(fd.m);
        q.fk = __builtin_amdgcn_readfirstlane(fd.floor_k);
        q.fwords = filter_words(q.m);
        q.Thi = __builtin_amdgcn_readfirstlane((uint32_t)(fd.T >> 32));
        q.Tlo = __builtin_amdgcn_readfirstlane((uint32_t)fd.T);
        q.ninv_lo = __builtin_amdgcn_readfirstlane((uint32_t)fd.M);        // the host put the bits of -1.0 / m (IEEE double) into M
        q.ninv_hi = __builtin_amdgcn_readfirstlane((uint32_t)(fd.M >> 32));
        return q;
    };

    uint32_t k = next_active(0);
    if (k >= nframes) return;
    Q64Frame cf = prepare(k);
    const uint32_t lds0 = __builtin_amdgcn_readfirstlane(lds_addr_of(lds));
    if (!(AB & 8)) dma_row(lds0, image + (uint64_t)cf.f * image_stride_words32, cf.fwords, wave, lane, nwaves);
    uint32_t cur = 0;
    const bool tl_on = (AB & 1024) && blockIdx.x < TL_WGS && (wave == 0 || wave == nwaves - 1) && g_query_timeline;
    uint64_t *tl = (AB & 1024) && g_query_timeline ? g_query_timeline + ((uint64_t)(blockIdx.x % TL_WGS) * 2 + (wave ? 1 : 0)) * MAX_BATCH * TL_PHASES : nullptr;
    auto stamp = [&](uint32_t frame_slot, uint32_t phase) {
        if ((AB & 1024) && tl_on && lane == 0) tl[frame_slot * TL_PHASES + phase] = __builtin_readcyclecounter();
    };

    while (true) {
        stamp(cf.f, 0);
        if (!(AB & 32)) {
            dma_wait_all();           // my share of DMA(cf.f) has landed (it was issued a frame ago) ...
            stamp(cf.f, 1);
            __syncthreads();          // ... and everyone's; nobody probes buffer cur^1 any more
        }
        stamp(cf.f, 2);
        const uint32_t kn = __builtin_amdgcn_readfirstlane(next_active(cf.f + 1));
        const bool more = kn < nframes;
        Q64Frame nf = cf;
        if (more) nf = prepare(kn);                               // scalar loads, off the critical path
        if (more && !(AB & 8)) dma_row(lds0 + (cur ^ 1u) * bufwords * 4u, image + (uint64_t)nf.f * image_stride_words32, nf.fwords, wave, lane, nwaves);
        const uint32_t fbase = vgpr_copy(lds0 + cur * bufwords * 4u);
        const uint32_t m_v = vgpr_copy(cf.m);
        const double ninv = __builtin_bit_cast(double, ((uint64_t)cf.ninv_hi << 32) | cf.ninv_lo);
        const uint64_t T = ((uint64_t)cf.Thi << 32) | cf.Tlo;
        const uint32_t fk = cf.fk;
        uint32_t pbf = 0, npass = 0;
        stamp(cf.f, 3);
        // floor(k*) is a small integer: straight-line, hand-pipelined code for the common values; a plain loop otherwise
#define RBF_Q64_PASS(FKV, CV) frame_part_f64<FKV, AB, CV, 0, QL_P>(hd1, hl1, hd2, hl2, ha, validmask, fbase, safe_pos, m_v, ninv, T, fk, pbf, npass)
        if (whole_wave) {
            switch (fk) {
            case 1: RBF_Q64_PASS(1, false); break;
            case 2: RBF_Q64_PASS(2, false); break;
            case 3: RBF_Q64_PASS(3, false); break;
            case 4: RBF_Q64_PASS(4, false); break;
            default: RBF_Q64_PASS(-1, false); break;
            }
        } else {                                                  // the frame's last segments: some positions lie past the end
            RBF_Q64_PASS(-1, true);
        }
#undef RBF_Q64_PASS
        stamp(cf.f, 4);
        if (!(AB & 64) && live) {
            pass_bytes[((uint64_t)cf.f * nseg + seg) * (QL_SEG_PIXELS / 8) + lane] = (uint8_t)~pbf;
            if (lane == 0) seg_cnt[(uint64_t)cf.f * nseg + seg] = npass;
        }
        stamp(cf.f, 5);
        if (!more) break;
        cf = nf;
        cur ^= 1u;
    }
}

// ------------------------------------------------------------------------------------------------------------------
// k_query_f64t -- the same kernel for filters that do not fit LDS twice (2160p: 306 KB; up to m < 2^23 bits = 1 MiB):
// the probe image is walked in TILES of `tile_words` dwords.  Per frame the two reductions of a pixel are done once (pos0, step stay in
// registers); per tile every probe position is rebuilt by stepping (3 cheap instructions) and redirected to the tile's
// SAFE dword unless it falls into the tile: word index relative to the tile, unsigned min against tile_words -- one
// instruction for "in this tile?" and the redirect.  A pixel whose extra probe is not activated gets position 2^32 - 1
// for it, which is in no tile.  Replaces k_query_tiled's scheme (single buffer: stage, wait, probe; per-pixel hashing;
// Barrett reductions; two compares + two selects per probe and tile) for every geometry the FP64 reduction covers.
// ------------------------------------------------------------------------------------------------------------------
template <int FK, int AB>
__device__ __forceinline__ void tile_part_f64(const uint32_t (&pos0)[QL_P], const uint32_t (&step)[QL_P], uint32_t notact /* bit it: no extra probe */,
                                              uint32_t lds_base_bytes, uint32_t tile_word0, uint32_t tile_words, uint32_t m, uint32_t fk_rt,
                                              uint32_t (&fail)[QL_P])
{
    const uint32_t fk = FK >= 0 ? (uint32_t)FK : fk_rt;
    auto probe = [&](uint32_t p) -> uint32_t {                    // image word of position p if it lies in this tile, else the SAFE dword (0)
        const uint32_t idx = min((p >> 5) - tile_word0, tile_words);
        const uint32_t w = *reinterpret_cast<const __attribute__((address_space(3))) uint32_t *>((uintptr_t)((idx << 2) + lds_base_bytes));
        return w << (p & 31u);
    };
    if constexpr (FK < 0) {
#pragma unroll
        for (int it = 0; it < QL_P; ++it) {
            uint32_t p = pos0[it];
            for (uint32_t j = 0; j < fk; ++j) {
                fail[it] |= probe(p);
                const uint32_t s2 = p + step[it];
                p = min(s2, s2 - m);
            }
            fail[it] |= probe(p | (uint32_t)(((int32_t)(notact << (31 - it))) >> 31));
        }
    } else {
        constexpr int G = 2, NG = QL_P / G, NP = FK + 1;
        uint32_t pp[NG][G][NP], wrd[NG][G][NP];
        auto positions = [&](int g) {
#pragma unroll
            for (int e = 0; e < G; ++e) {
                const int it = g * G + e;
                uint32_t p = pos0[it];
                // opaque per tile: otherwise the compiler hoists all (FK + 1) * 8 positions and shifts out of the tile loop
                // (48 registers that are live across it, and the kernel spills); stepping again per tile costs 3 instructions
                asm volatile("" : "+v"(p));
#pragma unroll
                for (int j = 0; j < FK; ++j) {
                    pp[g][e][j] = p;
                    const uint32_t s2 = p + step[it];
                    p = min(s2, s2 - m);
                }
                pp[g][e][FK] = p | (uint32_t)(((int32_t)(notact << (31 - it))) >> 31);
            }
        };
        auto loads = [&](int g) {
#pragma unroll
            for (int e = 0; e < G; ++e)
#pragma unroll
                for (int j = 0; j < NP; ++j) {
                    const uint32_t idx = min((pp[g][e][j] >> 5) - tile_word0, tile_words);
                    wrd[g][e][j] = (AB & 2) ? idx : *reinterpret_cast<const __attribute__((address_space(3))) uint32_t *>((uintptr_t)((idx << 2) + lds_base_bytes));
                }
        };
        auto combine = [&](int g) {
#pragma unroll
            for (int e = 0; e < G; ++e)
#pragma unroll
                for (int j = 0; j < NP; ++j) fail[g * G + e] = (wrd[g][e][j] << (pp[g][e][j] & 31u)) | fail[g * G + e];
        };
        positions(0);
        __builtin_amdgcn_sched_barrier(0);
        loads(0);
#pragma unroll
        for (int g = 1; g < NG; ++g) {
            __builtin_amdgcn_sched_barrier(0);
            positions(g);
            __builtin_amdgcn_sched_barrier(0);
            combine(g - 1);
            __builtin_amdgcn_sched_barrier(0);
            loads(g);
        }
        __builtin_amdgcn_sched_barrier(0);
        combine(NG - 1);
    }
}

template <int AB = 0>
__global__ __launch_bounds__(QL_THREADS) void k_query_f64t(
    uint64_t n, uint32_t nframes, const FrameTable tab /* M = bits of -1/m */, Seeds seeds,
    const uint32_t *__restrict__ image, uint64_t image_stride_words32, uint32_t tile_words /* multiple of 4; tile_words + 4 dwords of LDS */,
    uint32_t *__restrict__ seg_cnt, uint64_t nseg, uint64_t *__restrict__ pass_words)
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
    if (threadIdx.x < 4u) lds[tile_words + threadIdx.x] = 0u;

    // the hashes stay 64-bit integers here (6 registers per pixel instead of 8: this kernel also keeps pos0, step and fail
    // per pixel) and are converted to the FP64 reduction's (double, low dword) form once per frame, not per tile
    uint64_t h1[QL_P], h2[QL_P], ha[QL_P];
    uint32_t validmask = 0;
    const uint64_t i0 = seg * QL_SEG_PIXELS + (uint64_t)lane * QL_P;
    {
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
    }
    // invalid positions (past the end of the frame, or a dead wave) must fail: their verdict bits are forced afterwards
    uint32_t invalid_byte = 0;                                    // bit 7-j: pixel j is not a position of the frame
#pragma unroll
    for (int it = 0; it < QL_P; ++it) invalid_byte |= ((validmask >> it) & 1u) ? 0u : (0x80u >> it);
    uint8_t *pass_bytes = reinterpret_cast<uint8_t *>(pass_words);

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

    // Software-pipelined over the frames: the DMA of the NEXT frame's first tile is issued as soon as the last probes of this
    // frame are done (one barrier), and flies while this frame's verdicts go out and the next frame's 16 reductions per lane
    // -- which need no filter -- are computed (~1 600 of the ~2 400 cycles a 153 KB tile takes to arrive).
    uint32_t pos0[QL_P], step[QL_P], fail[QL_P];
    uint32_t notact = 0, m_v = 0, fk = 0;
    auto frame_setup = [&](uint32_t ff) {                            // geometry scalars + the two reductions of every pixel, once per frame
        const FrameDev fd = tab.f[ff];
        const uint32_t m_s = __builtin_amdgcn_readfirstlane(fd.m);
        fk = __builtin_amdgcn_readfirstlane(fd.floor_k);
        m_v = vgpr_copy(m_s);
        // (__builtin_amdgcn_readfirstlane returns int: every half goes through uint32_t, or the low one sign-extends into the high one)
        const uint32_t nhi = __builtin_amdgcn_readfirstlane((uint32_t)(fd.M >> 32)), nlo = __builtin_amdgcn_readfirstlane((uint32_t)fd.M);
        const uint32_t thi = __builtin_amdgcn_readfirstlane((uint32_t)(fd.T >> 32)), tlo = __builtin_amdgcn_readfirstlane((uint32_t)fd.T);
        const double ninv = __builtin_bit_cast(double, ((uint64_t)nhi << 32) | nlo);
        const uint64_t T = ((uint64_t)thi << 32) | tlo;
        notact = 0;
#pragma unroll
        for (int it = 0; it < QL_P; ++it) {
            // opaque copies: with the setup inlined before the loop and at its end, the compiler would otherwise keep the
            // 16 converted doubles (frame-invariant) live across the whole loop -- 32 registers, 174 dwords of spills
            uint64_t a1 = h1[it], a2 = h2[it];
            asm volatile("" : "+v"(a1), "+v"(a2));
            pos0[it] = mod_m_f64((double)a1, (uint32_t)a1, ninv, m_v);
            step[it] = mod_m_f64((double)a2, (uint32_t)a2, ninv, m_v);
            notact |= (ha[it] < T) ? 0u : (1u << it);
            fail[it] = 0;
        }
    };
    frame_setup(f);
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
        if (fnext < nframes) {
            if (!(AB & 32)) __syncthreads();      // this frame's last probes are done: the buffer is free
            stage_dma(fnext, fwords_next, 0);
        }
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
        frame_setup(f);
    }
}

} }  // namespace rbf::legacy

namespace rbf { namespace legacy {

// ------------------------------------------------------------------------------------------------------------------
// k_query_p4 -- k_query_f64 re-cut for OCCUPANCY: a lane owns 4 consecutive pixels instead of 8, which halves the
// per-lane hash state (32 registers), lets the kernel live in 64 VGPRs = 8 waves per SIMD, and lets TWO 1024-thread
// workgroups share a CU, each with its own single LDS buffer (76 KB + SAFE for a 1080p filter).
// Why: the frame pass is latency-bound -- measured on k_query_f64, pure frame passes take 144 / 95 / 74 us at 1 / 2 / 4
// waves per SIMD (T ~ 52 + 114 / waves us), and 4 is all that 8 pixels per lane (117 VGPRs) allow -- and the staging of a
// filter (~1 200 cycles of the CU's vector-memory front end) does not hide under the same workgroup's probes.  With two
// independent workgroups per CU one stages while the other probes, and within a workgroup the reductions of the NEXT
// frame (which need no filter) run between the DMA issue and its completion.
// Outputs: pass bytes in the same packed order; the segment (the unit of seg_cnt) is a wave's 256 pixels.
// MEASURED (tools/bench_query.hip, 1080p x 29): 112 us against k_query_f64's 104 us -- pure frame passes 79 us in both.
// Eight waves per SIMD buy nothing: per pixel the kernel issues ~30 % more instructions (loop, barrier, DMA issue, stores
// and ballots are per wave and frame, and a wave now carries half the pixels), and that cancels what the occupancy
// gains.  Kept selectable (rbf_ctx_force_generic bit 6) and parity-tested; k_query_f64 stays the default.
// ------------------------------------------------------------------------------------------------------------------
constexpr int P4_P = 4;                            // pixels per lane
constexpr int P4_SEG_PIXELS = P4_P * WAVE;         // 256

template <int FK, int AB>
__device__ __forceinline__ void p4_positions(const double (&hd1)[P4_P], const uint32_t (&hl1)[P4_P], const double (&hd2)[P4_P], const uint32_t (&hl2)[P4_P],
                                             const uint64_t (&ha)[P4_P], uint32_t safe_pos, uint32_t m, double ninv, uint64_t T, uint32_t (&pos)[P4_P][FK + 1])
{
#pragma unroll
    for (int it = 0; it < P4_P; ++it) {
        uint32_t p, step;
        if (AB & 1) { p = hl1[it] & 0x7FFFFu; step = hl2[it] & 0x3FFFFu; }
        else { p = mod_m_f64(hd1[it], hl1[it], ninv, m); step = mod_m_f64(hd2[it], hl2[it], ninv, m); }
#pragma unroll
        for (int j = 0; j < FK; ++j) {
            pos[it][j] = p;
            const uint32_t s2 = p + step;
            p = min(s2, s2 - m);
        }
        pos[it][FK] = (ha[it] < T) ? p : safe_pos;
    }
}

template <int FK, int AB>
__device__ __forceinline__ uint32_t p4_probe(const uint32_t (&pos)[P4_P][FK + 1], uint32_t lds_base_bytes, uint32_t invalid_nibble, uint32_t &npass)
{
    uint32_t w[P4_P][FK + 1];
#pragma unroll
    for (int it = 0; it < P4_P; ++it)
#pragma unroll
        for (int j = 0; j <= FK; ++j) w[it][j] = (AB & 2) ? pos[it][j] * 0x9E3779B1u : probe_image_word<AB>(lds_base_bytes, pos[it][j]);
    uint32_t nib = 0;                                             // bit 3-it: pixel `it` FAILED
#pragma unroll
    for (int it = 0; it < P4_P; ++it) {
        uint32_t fail = 0;
#pragma unroll
        for (int j = 0; j <= FK; ++j) fail = (w[it][j] << (pos[it][j] & 31u)) | fail;
        nib = __builtin_amdgcn_alignbit(nib, fail, 31);
    }
    nib |= invalid_nibble;
    const uint32_t pass = ~nib & 0xFu;
    if (!(AB & 4)) {
#pragma unroll
        for (int it = 0; it < P4_P; ++it) npass += __popcll(__ballot(((pass >> (3 - it)) & 1u) != 0));
    }
    return pass;
}

template <int AB = 0>
__global__ __launch_bounds__(QL_THREADS, 8) void k_query_p4(
    uint64_t n, uint32_t nframes, const FrameTable tab /* M = bits of -1/m */, Seeds seeds,
    const uint32_t *__restrict__ image, uint64_t image_stride_words32, uint32_t fwords_max,
    uint32_t *__restrict__ seg_cnt, uint64_t nseg /* of 256 pixels */, uint64_t *__restrict__ pass_words,
    uint4 *__restrict__ table_out /* nullable, see k_query_f64 */)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    const uint32_t safe_word = (fwords_max + 3u) & ~3u;           // one buffer; the dword after it stays 0 (SAFE)
    const uint32_t safe_pos = safe_word << 5;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t nwaves = blockDim.x >> 6;
    const uint64_t seg = (uint64_t)blockIdx.x * nwaves + wave;
    const bool live = seg < nseg;
    if (threadIdx.x < 4u) lds[safe_word + threadIdx.x] = 0u;

    double hd1[P4_P], hd2[P4_P];
    uint32_t hl1[P4_P], hl2[P4_P];
    uint64_t ha[P4_P];
    uint32_t validmask = 0;
    const uint64_t i0 = seg * P4_SEG_PIXELS + (uint64_t)lane * P4_P;
    {
        uint64_t h1[P4_P], h2[P4_P];
#pragma unroll
        for (int it = 0; it < P4_P; ++it) {
            h1[it] = 0; h2[it] = 0; ha[it] = ~0ull;
            if (live && i0 + it < n) validmask |= 1u << it;
        }
        if (AB & 16) {
#pragma unroll
            for (int it = 0; it < P4_P; ++it) { h1[it] = (i0 + it) * P1; h2[it] = (i0 + it) * P2 + seeds.h2; ha[it] = (i0 + it) * P3; }
        } else if (!hash3_run4((uint32_t)i0, validmask, seeds, h1, h2, ha)) {
#pragma unroll
            for (int it = 0; it < P4_P; ++it) {
                const bool act = (validmask >> it) & 1u;
                const Hash3 h = hash3_index((uint32_t)(i0 + it), act, seeds);
                h1[it] = h.h1; h2[it] = h.h2; ha[it] = h.ha;
            }
        }
#pragma unroll
        for (int it = 0; it < P4_P; ++it) {
            hd1[it] = (double)h1[it]; hl1[it] = (uint32_t)h1[it];
            hd2[it] = (double)h2[it]; hl2[it] = (uint32_t)h2[it];
        }
        if (table_out && live) {                                  // slot-major inside the 512-index segment (hash_table_slot)
#pragma unroll
            for (int it = 0; it < P4_P; ++it) {
                const uint32_t idx = (uint32_t)(i0 + it);
                const uint64_t d1 = __builtin_bit_cast(uint64_t, hd1[it]), d2 = __builtin_bit_cast(uint64_t, hd2[it]);
                uint4 *e = table_out + 2 * (uint64_t)hash_table_slot(idx);
                e[0] = make_uint4((uint32_t)d1, (uint32_t)(d1 >> 32), (uint32_t)d2, (uint32_t)(d2 >> 32));
                e[1] = make_uint4(hl1[it], hl2[it], (uint32_t)ha[it], (uint32_t)(ha[it] >> 32));
            }
        }
    }
    uint32_t invalid_nibble = 0;
#pragma unroll
    for (int it = 0; it < P4_P; ++it) invalid_nibble |= ((validmask >> it) & 1u) ? 0u : (8u >> it);
    uint8_t *pass_bytes = reinterpret_cast<uint8_t *>(pass_words);
    const uint64_t seg_bytes = P4_SEG_PIXELS / 8;                 // 32 bytes of verdicts per wave and frame

    for (uint32_t g = 0; g < nframes; ++g) {
        if (tab.f[g].m == 0) {
            if (live && lane == 0) seg_cnt[(uint64_t)g * nseg + seg] = 0;
            if (live && lane < seg_bytes) pass_bytes[((uint64_t)g * nseg + seg) * seg_bytes + lane] = 0;
        }
    }
    auto next_active = [&](uint32_t k) -> uint32_t { while (k < nframes && tab.f[k].m == 0) ++k; return k; };
    const uint32_t fbase = vgpr_copy(__builtin_amdgcn_readfirstlane(lds_addr_of(lds)));
    uint32_t f = __builtin_amdgcn_readfirstlane(next_active(0));
    while (f < nframes) {
        const FrameDev fd = tab.f[f];
        const uint32_t m_s = __builtin_amdgcn_readfirstlane(fd.m);
        const uint32_t fk = __builtin_amdgcn_readfirstlane(fd.floor_k);
        const uint32_t nhi = __builtin_amdgcn_readfirstlane((uint32_t)(fd.M >> 32)), nlo = __builtin_amdgcn_readfirstlane((uint32_t)fd.M);
        const uint32_t thi = __builtin_amdgcn_readfirstlane((uint32_t)(fd.T >> 32)), tlo = __builtin_amdgcn_readfirstlane((uint32_t)fd.T);
        const double ninv = __builtin_bit_cast(double, ((uint64_t)nhi << 32) | nlo);
        const uint64_t T = ((uint64_t)thi << 32) | tlo;
        const uint32_t m_v = vgpr_copy(m_s);
        if (!(AB & 32)) __syncthreads();                          // everyone is done probing the previous filter
        if (!(AB & 8)) dma_row(__builtin_amdgcn_readfirstlane(lds_addr_of(lds)), image + (uint64_t)f * image_stride_words32, filter_words(m_s), wave, lane, nwaves);
        // the reductions need no filter: they run while the DMA flies (and while the CU's other workgroup probes)
        uint32_t npass = 0, pass;
#define RBF_P4_FRAME(FKV)                                                                                          \
        do {                                                                                                       \
            uint32_t pos[P4_P][FKV + 1];                                                                           \
            p4_positions<FKV, AB>(hd1, hl1, hd2, hl2, ha, safe_pos, m_v, ninv, T, pos);                            \
            if (!(AB & 32)) { dma_wait_all(); __syncthreads(); }                                                   \
            pass = p4_probe<FKV, AB>(pos, fbase, invalid_nibble, npass);                                           \
        } while (0)
        switch (fk) {
        case 0: RBF_P4_FRAME(0); break;
        case 1: RBF_P4_FRAME(1); break;
        case 2: RBF_P4_FRAME(2); break;
        case 3: RBF_P4_FRAME(3); break;
        case 4: RBF_P4_FRAME(4); break;
        case 5: RBF_P4_FRAME(5); break;
        default: {                                                // floor(k*) > 5 (p < 1.2 %): stepping inside the probe loop
            if (!(AB & 32)) { dma_wait_all(); __syncthreads(); }
            uint32_t nib = 0;
#pragma unroll
            for (int it = 0; it < P4_P; ++it) {
                uint32_t p = mod_m_f64(hd1[it], hl1[it], ninv, m_v);
                const uint32_t step = mod_m_f64(hd2[it], hl2[it], ninv, m_v);
                uint32_t fail = 0;
                for (uint32_t j = 0; j < fk; ++j) {
                    fail = (probe_image_word<AB>(fbase, p) << (p & 31u)) | fail;
                    const uint32_t s2 = p + step;
                    p = min(s2, s2 - m_v);
                }
                const uint32_t pc = (ha[it] < T) ? p : safe_pos;
                fail = (probe_image_word<AB>(fbase, pc) << (pc & 31u)) | fail;
                nib = __builtin_amdgcn_alignbit(nib, fail, 31);
            }
            pass = ~(nib | invalid_nibble) & 0xFu;
#pragma unroll
            for (int it = 0; it < P4_P; ++it) npass += __popcll(__ballot(((pass >> (3 - it)) & 1u) != 0));
        } break;
        }
#undef RBF_P4_FRAME
        // a byte of the packed pass vector = the nibbles of an even lane (pixels 8j .. 8j+3) and its odd neighbour
        const uint32_t other = (uint32_t)__shfl_xor((int)pass, 1);
        if (!(AB & 64) && live) {
            if (!(lane & 1u)) pass_bytes[((uint64_t)f * nseg + seg) * seg_bytes + (lane >> 1)] = (uint8_t)((pass << 4) | other);
            if (lane == 0) seg_cnt[(uint64_t)f * nseg + seg] = npass;
        }
        f = __builtin_amdgcn_readfirstlane(next_active(f + 1));
    }
}

} }  // namespace rbf::legacy
