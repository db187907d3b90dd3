// rbf_kernels_u64.h -- k_query_u64: round 4's frames-inner FP64 query kernel (filters of 2^15 <= m < 2^23 bits that fit LDS twice:
// BASELINE config 2).  Same outputs and the same arithmetic as k_query_s64 (rbf_kernels_s64.h; reference semantics
// improved_video_compressor.py:116-138, :245-253) -- the frame pass itself IS k_query_s64's (frame_pass_rows / frame_pass_plain).
// What is new is everything AROUND the pass.  Round 3's stamps showed a wave spending 45 % of a frame outside the pass, and the
// disassembly showed why: ~75 scalar and ~50 vector instructions per wave and frame of loop head (next geometry -> SGPRs, the
// stager's address arithmetic, a switch over floor(k*)) and of deferred outputs (four ballots, scalar adds, two 64-bit address
// multiplies, two exec-masked stores), all of them issued by all sixteen waves at the same moment, right behind the frame's barrier,
// through the CU's ONE scalar unit (tools/opbench2.hip, profiles/r04_opbench2.txt: a scalar instruction costs a SIMD ~3 issue cycles
// next to its VALU stream, and sixteen waves issuing scalar code together get one instruction per 16 cycles each).  Here:
//
//  1. FRAME RECORDS IN LDS, 32 bytes, written once by the prologue: {m, c, -1/m} for the pass of frame j, and -- already shifted to
//     where the loop needs them -- the image row of frame j + 1 (what the stager fetches during pass j) and the output row of frame
//     j - 1 (what leaves during pass j).  The loop head is two broadcast LDS reads and three v_readfirstlane (-1/m, the output row); m,
//     c and the image row's offset are USED FROM THE VGPRS THE READ RETURNS -- no scalar arithmetic at all.
//  2. THE HOST ORDERS THE CODED FRAMES BY floor(k*) (the compacted table may be in any order: every record names its output row), and
//     the kernel runs one loop per class: no switch inside the frame loop.
//  3. THE NEXT IMAGE RIDES IN BY LDS-DMA (global_load_lds_dwordx4), all five pieces of a wave issued at the top of the pass and
//     waited for once, in front of the next barrier: no register slots (the kernel drops from 119 to 111 VGPRs), no ds_write
//     instructions, no load a pass waits for.  Round 2's LDS-DMA kernel (k_query_f64) lost against register staging; in THIS loop it
//     wins 6 us per launch (profiles/r04_query_u64.txt).  Every row is staged at the batch's row pitch (the rows are padded to it
//     anyway), so offsets and the lane mask of the last piece are launch constants.
//  4. Pass counts: a lane adds popc(verdict byte) of two consecutive frames into one packed register; every second frame ONE
//     six-step DPP reduction yields both wave totals in lane 63, which stores them.  No ballots, no scalar adds.
//  5. Outputs are addressed as base + 32-bit offset (one v_lshl_add per store) from the output row in the record.
#pragma once
#include "rbf_kernels_s64.h"
#include <algorithm>
#include <cstring>
#include <type_traits>

namespace rbf {

constexpr uint32_t U64_REC_BYTES = 32;
__host__ __device__ constexpr uint32_t u64_geo_bytes(uint32_t nactive) { return (nactive + 1u) * U64_REC_BYTES; }   // LDS behind the two image buffers: one record per coded frame + 1
constexpr int U64_CLASSES = 6;                                    // floor(k*) = 1, 2, 3, 4, 5 in rows, then everything else (plain pass)

struct U64Classes { uint32_t n[U64_CLASSES]; };                  // coded frames per class, in the order of the compacted table

// Host: the FrameTable k_query_u64 reads (see the kernel) from the batch's plain table (m, floor_k, T per frame; m == 0: not coded).
__host__ inline FrameTable query_table_u64(const FrameTable &tab, uint32_t nframes, uint32_t *nactive, U64Classes *cls, uint64_t (&empty)[2])
{
    FrameTable q;
    memset(&q, 0, sizeof q);
    memset(cls, 0, sizeof *cls);
    empty[0] = empty[1] = 0;
    uint64_t sorted[MAX_BATCH];
    uint32_t coded = 0;
    for (uint32_t f = 0; f < nframes; ++f) {
        if (tab.f[f].m) sorted[coded++] = tab.f[f].T;
        else empty[f >> 6] |= 1ull << (f & 63);
    }
    std::sort(sorted, sorted + coded);
    uint32_t j = 0;
    for (int k = 0; k < U64_CLASSES; ++k)
        for (uint32_t f = 0; f < nframes; ++f) {
            if (!tab.f[f].m) continue;
            const uint32_t fk = tab.f[f].floor_k;
            const int kf = fk >= 1u && fk <= 5u ? (int)fk - 1 : 5;
            if (kf != k) continue;
            const double ninv = -1.0 / (double)tab.f[f].m;
            q.f[j].m = tab.f[f].m;
            memcpy(&q.f[j].M, &ninv, 8);
            q.f[j].floor_k = (fk & 0xFFu) | ((uint32_t)(std::lower_bound(sorted, sorted + coded, tab.f[f].T) - sorted) << 8) | (f << 16);
            q.f[j].T = sorted[j];
            ++cls->n[k];
            ++j;
        }
    *nactive = coded;
    return q;
}

// Next frame's image -> the other LDS buffer by LDS-DMA (global_load_lds_dwordx4: a lane's 16 bytes land at M0 + lane * 16): no
// register slots, no ds_write, nothing of the pass waits for a load -- all five 16 KiB piece rows of a workgroup (a wave: 1 KiB of
// each) are issued at the top of the pass (the buffer is free behind the frame's barrier) and the wave waits for them once, in
// front of the next barrier.  M0 = LDS address of the wave's piece; saved and restored around the block (a reserved register, not a
// clobber).  Every piece runs under a launch-constant lane mask (rows are staged at the batch's pitch: lanes past its end must not
// write behind the buffer; a piece wholly past it runs with no lanes).
template <bool ON>
struct RowDmaC {
    const uint8_t *image;           // the batch's probe images (the kernel argument: saddr addressing)
    uint32_t src;                   // byte offset of the row to stage + wave * 5120 + lane * 16 + 2048 (the image block is < 4 GB)
    uint32_t dst_m0;                // LDS byte address of the destination buffer + wave * 5120 + 2048 (uniform: SGPR)
    uint64_t mask[5];               // lanes of piece i inside the row (uniform)
    bool full;                      // all five pieces of this wave lie inside the row (uniform): no exec masks

    // A wave stages FIVE CONSECUTIVE KiB of the row.  The instruction's immediate offset moves the global AND the LDS address, so the
    // five pieces are one M0 value and the offsets -2048 ... +2048 (13 signed bits): 3 scalar instructions per frame and wave.  (Until
    // round 4 a wave's pieces lay 16 KiB apart -- piece = wave + 16 i -- which cost an s_add of M0, an exec mask and a v_add of the
    // source per piece: 14 SALU + 4 VALU per frame and wave in a loop that is bound by instruction issue.)
    __device__ __forceinline__ void issue() const
    {
        uint32_t keep; uint64_t keepx;
        if (full) {
            asm volatile("s_mov_b32 %0, m0\n\t"
                         "s_mov_b32 m0, %1\n\t"
                         "s_nop 0\n\t"
                         "global_load_lds_dwordx4 %2, %3 offset:-2048\n\t"
                         "global_load_lds_dwordx4 %2, %3 offset:-1024\n\t"
                         "global_load_lds_dwordx4 %2, %3\n\t"
                         "global_load_lds_dwordx4 %2, %3 offset:1024\n\t"
                         "global_load_lds_dwordx4 %2, %3 offset:2048\n\t"
                         "s_mov_b32 m0, %0"
                         : "=&s"(keep) : "s"(dst_m0), "v"(src), "s"(image) : "memory");
            return;
        }
        asm volatile("s_mov_b32 %0, m0\n\t"
                     "s_mov_b64 %1, exec\n\t"
                     "s_mov_b32 m0, %2\n\t"
                     "s_mov_b64 exec, %5\n\t"
                     "global_load_lds_dwordx4 %3, %4 offset:-2048\n\t"
                     "s_mov_b64 exec, %6\n\t"
                     "global_load_lds_dwordx4 %3, %4 offset:-1024\n\t"
                     "s_mov_b64 exec, %7\n\t"
                     "global_load_lds_dwordx4 %3, %4\n\t"
                     "s_mov_b64 exec, %8\n\t"
                     "global_load_lds_dwordx4 %3, %4 offset:1024\n\t"
                     "s_mov_b64 exec, %9\n\t"
                     "global_load_lds_dwordx4 %3, %4 offset:2048\n\t"
                     "s_mov_b64 exec, %1\n\t"
                     "s_mov_b32 m0, %0"
                     : "=&s"(keep), "=&s"(keepx)
                     : "s"(dst_m0), "v"(src), "s"(image), "s"(mask[0]), "s"(mask[1]), "s"(mask[2]), "s"(mask[3]), "s"(mask[4])
                     : "memory");
    }
    __device__ __forceinline__ void at(int g)
    {
        if (ON && g == 0) issue();
    }
};

// (the pass counts of two frames are summed over the wave as a packed pair of 16-bit counts, each total <= 512: wave_sum_to_lane63)

// FrameTable as this kernel reads it (host: query_table_u64, rbf_api.hip) -- COMPACTED over the coded frames and ORDERED BY CLASS
// (floor(k*) = 1, 2, 3, 4, 5, then the rest):
//   f[j].m, f[j].M = bits of -1.0 / m        of the j-th coded frame of that order,
//   f[j].floor_k = floor(k*) | c << 8 | frame index << 16        (c = coded thresholds below the frame's own),
//   f[j].T = j-th smallest threshold of the coded frames (any order of frames: the thresholds are only searched).
// `cls.n[k]`: frames of class k.  `empty_lo / empty_hi`: bit f set = frame f of the batch is not coded and this launch writes its
// (empty) outputs.  Dynamic LDS: two image buffers of ((fwords_max + 3) & ~3) + 4 dwords, then u64_geo_bytes(nactive).
// `image_stride_words32` is also what is staged per frame: rows must be readable over their whole pitch (the library's are).
// (The measurement variants of this body -- no staging, no barrier, no outputs, no priorities -- were tools/legacy/rbf_kernels_u64_ab.h up to round 4: git history.)
template <bool WIDE>
__device__ __forceinline__ void query_u64_body(
    uint64_t n, uint32_t nactive, const FrameTable &tab, const U64Classes &cls, Seeds seeds,
    const uint32_t *__restrict__ image, uint64_t image_stride_words32, uint32_t fwords_max,
    uint32_t *__restrict__ seg_cnt, uint64_t nseg, uint64_t *__restrict__ pass_words,
    uint4 *__restrict__ table_out, uint64_t empty_lo, uint64_t empty_hi)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    const uint32_t bufwords = ((fwords_max + 3u) & ~3u) + 4u;
    const uint32_t safe_pos = ((fwords_max + 3u) & ~3u) << 5;
    const uint32_t lane = threadIdx.x & 63u, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);      // (scalar: so are seg and live)
    const uint32_t seg = blockIdx.x * QL_WAVES + wave;            // < 2^23: n < 2^32 pixels in segments of 512
    const bool live = seg < nseg;
    if (threadIdx.x < 8u) lds[(threadIdx.x >> 2) * bufwords + (bufwords - 4u) + (threadIdx.x & 3u)] = 0u;   // SAFE; visible after the first barrier
    uint4 *geo = reinterpret_cast<uint4 *>(lds + 2u * bufwords);
    uint64_t *tl = reinterpret_cast<uint64_t *>(lds + bufwords);  // sorted thresholds: buffer 1 is free until the first pass stages into it
    FrameDev fd_mine{};
    if (threadIdx.x < 2u * MAX_BATCH) fd_mine = tab.f[threadIdx.x < nactive ? threadIdx.x : 0u];

    // ---- frame-independent part (as k_query_s64): hashes of my 8 consecutive pixel indices as (double, low dword), activation ranks
    static_assert(QL_P == 8, "a lane's verdicts fill one byte");
    double hd1[QL_P], hd2[QL_P];
    uint32_t hl1[QL_P], hl2[QL_P];
    uint32_t rank_lo = 0, rank_hi = 0;
    uint32_t validmask = 0;
    const uint64_t i0 = (uint64_t)seg * QL_SEG_PIXELS + (uint64_t)lane * QL_P;
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
        }
        if (table_out && live) hash_table_store8(table_out, n, seg, lane, h1, h2, ha);
        if (threadIdx.x < 2u * MAX_BATCH) {
            const uint32_t t = threadIdx.x;
            tl[t] = t < nactive ? fd_mine.T : ~0ull;
            if (t < nactive) {
                // record t: my own {m, c, -1/m}; my image row goes into record t - 1 (staged during pass t - 1), my output row into
                // record t + 1 (written during pass t + 1).  Record nactive - 1 stages its own row again (into the buffer nobody reads).
                const uint32_t f = fd_mine.floor_k >> 16;
                const uint32_t row = (uint32_t)((uint64_t)f * image_stride_words32 * 4u);  // byte offset of my image row (the image block is < 4 GB: the host's condition)
                const uint32_t out_row = (uint32_t)((uint64_t)f * nseg);                   // index of (frame f, segment 0) in seg_cnt; x 64 = its pass byte
                geo[2u * t] = make_uint4(fd_mine.m, (fd_mine.floor_k >> 8) & 0xFFu, (uint32_t)fd_mine.M, (uint32_t)(fd_mine.M >> 32));
                uint32_t *w = reinterpret_cast<uint32_t *>(geo);
                if (t) w[8u * (t - 1u) + 4u] = row;
                if (t + 1u == nactive) w[8u * t + 4u] = row;
                if (t == 0u) w[8u * nactive + 4u] = row;                                   // frame 0's own row: the prologue's staging
                w[8u * t + 6u] = fd_mine.floor_k & 0xFFu;
                w[8u * (t + 1u) + 7u] = out_row;                                           // (record nactive exists: u64_geo_bytes)
            }
        }
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
    const bool whole_wave = __builtin_amdgcn_readfirstlane((uint32_t)__all(validmask == 0xFFu)) != 0u;
    uint8_t *pass_bytes = reinterpret_cast<uint8_t *>(pass_words);

    for (uint32_t half = 0; half < 2; ++half) {                   // frames that are not coded: nothing passes (none in the common case)
        uint64_t bits = half ? empty_hi : empty_lo;
        while (bits) {
            const uint32_t g = half * 64u + (uint32_t)__builtin_ctzll(bits);
            bits &= bits - 1;
            if (live && lane == 0) seg_cnt[(uint64_t)g * nseg + seg] = 0;
            if (live) pass_bytes[((uint64_t)g * nseg + seg) * (QL_SEG_PIXELS / 8) + lane] = 0;
        }
    }
    if (nactive == 0) return;

    const uint32_t lds0 = __builtin_amdgcn_readfirstlane(lds_addr_of(lds));
    const uint32_t pitch_bytes = (uint32_t)image_stride_words32 * 4u;
    const uint32_t off0 = wave * 5120u + lane * 16u;             // a wave stages five consecutive KiB (RowDmaC)
    RowDmaC<true> dm;
    dm.image = reinterpret_cast<const uint8_t *>(image);
#pragma unroll
    for (int i = 0; i < 5; ++i) dm.mask[i] = __ballot(off0 + (uint32_t)i * 1024u + 16u <= pitch_bytes);
    dm.full = dm.mask[4] == ~0ull;
    const uint32_t off0b = off0 + 2048u, wave_dst = wave * 5120u + 2048u;
    const uint32_t buf_sum = lds0 + lds0 + bufwords * 4u;        // buffer 0 + buffer 1: the other buffer of b is buf_sum - b
    uint32_t fbase = vgpr_copy(lds0);                             // the buffer the pass probes (the stager fills the other one)
    {   // the first frame's image into buffer 0 (its row is in the record behind the last one); waited for at the head of the first frame
        const uint32_t *w = reinterpret_cast<const uint32_t *>(geo);
        dm.src = w[8u * nactive + 4u] + off0 + 2048u;
        dm.dst_m0 = lds0 + wave * 5120u + 2048u;
        dm.at(0);
    }
    const uint32_t safe_v = vgpr_copy(safe_pos);
    const uint32_t seg_lane = seg * (QL_SEG_PIXELS / 8) + lane;      // my verdict byte inside a frame's row of pass bytes
    uint32_t cnt2 = 0;                                            // popc(verdict byte) of the last two frames, 16 bits each (newest in the high half)
    uint32_t row_prev = 0, row_prev2 = 0;                         // ... and their output rows (uniform: SGPRs)
    uint32_t out_row = 0;                                         // output row (record field 7) of the frame whose outputs are pending (uniform: SGPR)
    uint32_t out_pb = 0;                                          // verdict byte waiting to be written
    bool out_pending = false, have_two = false;

    // outputs of the previous frame, run from inside the current frame's pass once its first reads are in flight
    auto flush = [&]() {
        if (!out_pending) return;
        if (live) pass_bytes[(out_row << 6) + seg_lane] = (uint8_t)out_pb;        // one v_lshl_add_u32, one store
        {
            cnt2 = __builtin_amdgcn_alignbit(__popc(out_pb), cnt2, 16);   // (cnt << 16) | (cnt2 >> 16)
            row_prev2 = row_prev; row_prev = out_row;
            if (have_two) {
                const uint32_t tot = wave_sum_to_lane63(cnt2);
                if (live && lane == 63u) {
                    seg_cnt[row_prev2 + seg] = tot & 0xFFFFu;
                    seg_cnt[row_prev + seg] = tot >> 16;
                }
            }
            have_two = !have_two;
        }
    };

    uint32_t j = 0;
    auto frames = [&](auto fk_tag, uint32_t count) {
        constexpr int FK = decltype(fk_tag)::value;               // 1..5: rows pass; 0: plain pass with the record's floor(k*)
        const uint32_t end = j + count;
        for (; j < end; ++j) {
            // ---- head: this frame's record.  m, c, -1/m stay in the VGPRs the broadcast read returns; the row pointer goes to SGPRs.
            const uint4 ga = geo[2u * j], gb = geo[2u * j + 1u];
            const uint32_t m_v = ga.x, c_v = ga.y;
            const double ninv = __builtin_bit_cast(double, ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane(ga.w) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane(ga.z));
            uint32_t x[4] = {0, 0, 0, 0};
            const bool rows = FK > 0 && whole_wave;
            if (rows) rows_reduce4(0, hd1, hl1, hd2, hl2, m_v, ninv, x);          // needs no image: in front of the barrier
            dm.src = gb.x + off0b;                                                     // (a uniform value used from the VGPR the read returned)
            dm.dst_m0 = (uint32_t)__builtin_amdgcn_readfirstlane(buf_sum - fbase) + wave_dst;
            dma_wait_all();                   // my pieces of THIS frame's image (issued during the previous pass, or by the prologue) have landed
            out_row = (uint32_t)__builtin_amdgcn_readfirstlane(gb.w);                 // output row of the PREVIOUS frame (shifted by the prologue)
            __syncthreads();                  // everyone's writes of the buffer I probe have landed; nobody probes the other one any more
            uint32_t pbf = 0;
            if (rows) {
                if constexpr (FK > 0) frame_pass_rows<FK, (FK < 5), true>(hd1, hl1, hd2, hl2, rank_lo, rank_hi, c_v, fbase, safe_v, m_v, ninv, x, pbf, dm, flush);
            } else {
                flush();
                const uint32_t fk = FK > 0 ? (uint32_t)FK : (uint32_t)__builtin_amdgcn_readfirstlane(gb.z);
                frame_pass_plain(hd1, hl1, hd2, hl2, rank_lo, rank_hi, c_v, validmask, fbase, safe_v, m_v, ninv, fk, pbf, dm);
            }
            out_pb = ~pbf & 0xFFu; out_pending = true;
            fbase = buf_sum - fbase;
        }
    };
    frames(std::integral_constant<int, 1>{}, cls.n[0]);
    frames(std::integral_constant<int, 2>{}, cls.n[1]);
    frames(std::integral_constant<int, 3>{}, cls.n[2]);
    if constexpr (WIDE) {
        frames(std::integral_constant<int, 4>{}, cls.n[3]);
        frames(std::integral_constant<int, 5>{}, cls.n[4]);
    }
    frames(std::integral_constant<int, 0>{}, cls.n[5] + (WIDE ? 0u : cls.n[3] + cls.n[4]));
    // the last frame's outputs: its row is in the record behind the last one
    out_row = (uint32_t)__builtin_amdgcn_readfirstlane(reinterpret_cast<const uint32_t *>(geo)[8u * nactive + 7u]);
    flush();
    if (have_two) {                                  // an odd number of frames: one count is still packed
        const uint32_t tot = wave_sum_to_lane63(cnt2);
        if (live && lane == 63u) seg_cnt[row_prev + seg] = tot >> 16;
    }
}

#define RBF_U64_PARAMS uint64_t n, uint32_t nactive, const FrameTable tab, const U64Classes cls, Seeds seeds, const uint32_t *__restrict__ image, uint64_t image_stride_words32, \
    uint32_t fwords_max, uint32_t *__restrict__ seg_cnt, uint64_t nseg, uint64_t *__restrict__ pass_words, uint4 *__restrict__ table_out, uint64_t empty_lo, uint64_t empty_hi
#define RBF_U64_ARGS n, nactive, tab, cls, seeds, image, image_stride_words32, fwords_max, seg_cnt, nseg, pass_words, table_out, empty_lo, empty_hi
// k_query_u64 is capped at 120 registers (it needs 110): registers are handed out in eights, so two waves of a neighbour pipeline's mask
// or compaction kernel fit next to its four on every SIMD.  (Padding it to 119 / 127 so that fewer fit costs the four-pipeline step 1-3 %:
// profiles/r04_coresidency.txt.)
__attribute__((amdgpu_num_vgpr(60))) __global__ __launch_bounds__(QL_THREADS) void k_query_u64(RBF_U64_PARAMS) { query_u64_body<false>(RBF_U64_ARGS); }
__global__ __launch_bounds__(QL_THREADS) void k_query_u64w(RBF_U64_PARAMS) { query_u64_body<true>(RBF_U64_ARGS); }
#undef RBF_U64_PARAMS
#undef RBF_U64_ARGS

}  // namespace rbf
