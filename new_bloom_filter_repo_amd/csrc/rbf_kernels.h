// rbf_kernels.h -- gfx950 kernels of the rational-Bloom residual coder (generic path).
//
// Work decomposition: a SEGMENT is 1024 consecutive pixels (16 wave-iterations of 64); a wave
// owns one segment, so the per-segment pass count is produced without inter-wave communication.
// blockIdx.y is the frame of the batch.
//
// Bit vectors at rest are MSB-first per byte (see rbf_device.h); the 64-bit pass words that only
// kernels see are natural order (bit l = lane l).
#pragma once
#include "rbf_device.h"

namespace rbf {

constexpr int SEG_PIXELS = 1024;                 // pixels per segment (one wave)
constexpr int SEG_ITERS = SEG_PIXELS / WAVE;     // 16
constexpr int WG_THREADS = 256;
constexpr int WG_WAVES = WG_THREADS / WAVE;      // 4

// ---- wave-wide sums without LDS: DPP adds (a __shfl_* is a ds_bpermute round trip through LDS on gfx950) ------------------------
// Sum over the wave's 64 lanes, valid in lane 63 (rows 1-3 hold partial totals): quad swaps, row rotates -- every lane of a row holds
// the row's sum -- then the rows' totals broadcast forward.  Also used on a packed pair of 16-bit counts.
__device__ __forceinline__ uint32_t wave_sum_to_lane63(uint32_t v)
{
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, true);      // quad_perm [1,0,3,2]
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, true);      // quad_perm [2,3,0,1]
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x124, 0xF, 0xF, true);     // row_ror:4
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x128, 0xF, 0xF, true);     // row_ror:8
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, true);     // row_bcast:15 -> rows 1 and 3
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, true);     // row_bcast:31 -> rows 2 and 3
    return v;
}
// Inclusive prefix sum over the wave's 64 lanes in six DPP adds: a Hillis-Steele scan inside every row of 16 lanes
// (row_shr 1, 2, 4, 8; lanes without a source add 0), then row 0's / row 2's total onto rows 1 / 3 (row_bcast:15) and the
// total of rows 0-1 onto rows 2-3 (row_bcast:31).
__device__ __forceinline__ uint32_t wave_inclusive_scan(uint32_t x)
{
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, false);      // row_shr:1
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, false);      // row_shr:2
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xf, false);      // row_shr:4
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xf, 0xf, false);      // row_shr:8
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xa, 0xf, false);      // row_bcast:15 -> rows 1 and 3
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xc, 0xf, false);      // row_bcast:31 -> rows 2 and 3
    return x;
}

// ------------------------------------------------------------------------------------------
// A1  residual mask: bit = abs_int16(prev - curr) > thr   (improved_video_compressor.py:801,808)
// ------------------------------------------------------------------------------------------
template <typename SAMPLE>
__device__ __forceinline__ bool residual_bit(SAMPLE a, SAMPLE b, int32_t thr)
{
    // numpy: astype(int16) wraps uint16; int16 - int16 wraps; np.abs(int16 -32768) stays -32768.
    const int16_t d = (int16_t)(uint16_t)((uint16_t)a - (uint16_t)b);
    const int16_t ad = d < 0 ? (int16_t)(uint16_t)(0u - (uint16_t)d) : d;
    return (int32_t)ad > thr;
}

template <typename SAMPLE>
__global__ __launch_bounds__(WG_THREADS) void k_residual_mask(
    const uint8_t *__restrict__ frames, uint64_t frame_stride, uint32_t width, uint64_t n,
    uint64_t row_pitch, uint32_t pixel_stride, int32_t thr_all, const int32_t *__restrict__ thr_tab /* nullable: per pair */,
    uint64_t *__restrict__ masks, uint64_t mask_stride_words, uint64_t *__restrict__ ones, uint64_t first_word)
{
    __shared__ uint32_t wave_ones[WG_WAVES];
    const uint32_t f = blockIdx.y;
    const int32_t thr = thr_tab ? thr_tab[f] : thr_all;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint8_t *prev = frames + (uint64_t)f * frame_stride;
    const uint8_t *curr = prev + frame_stride;
    uint64_t *mask = masks + (uint64_t)f * mask_stride_words;
    const uint64_t nwords = (n + 63) >> 6;
    uint32_t cnt = 0;
    // one 64-pixel word per wave per step, block-contiguous
    for (uint64_t w = first_word + (uint64_t)blockIdx.x * WG_WAVES + wave; w < nwords; w += (uint64_t)gridDim.x * WG_WAVES) {
        const uint64_t i = w * 64 + lane;
        bool bit = false;
        if (i < n) {
            const uint64_t y = i / width, x = i - y * width;
            const uint64_t off = y * row_pitch + x * pixel_stride;
            const SAMPLE a = *(const SAMPLE *)(prev + off);
            const SAMPLE b = *(const SAMPLE *)(curr + off);
            bit = residual_bit<SAMPLE>(a, b, thr);
        }
        const uint64_t word = __ballot(bit);
        cnt += __popcll(word);
        if (lane == 0) mask[w] = flip_bytes64(word);
    }
    if (lane == 0) wave_ones[wave] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t s = 0;
        for (int k = 0; k < WG_WAVES; ++k) s += wave_ones[k];
        if (s) atomicAdd((unsigned long long *)&ones[f], (unsigned long long)s);
    }
}

// ------------------------------------------------------------------------------------------
// A4  insert: every '1' position of the mask sets its probe bits (add_index, :99-114)
// ------------------------------------------------------------------------------------------
// Each lane takes one 32-bit mask word; the wave expands the set positions into an LDS list so
// that the (expensive) hashing runs with all lanes busy although only ~9 % of pixels are set.
__global__ __launch_bounds__(WG_THREADS) void k_insert(
    const uint32_t *__restrict__ masks, uint64_t mask_stride_words32, uint64_t n,
    const FrameTable tab, Seeds seeds,
    uint32_t *__restrict__ filters, uint64_t filter_stride_words32)
{
    __shared__ uint32_t list[WG_WAVES][WAVE * 32];
    const uint32_t f = blockIdx.y;
    const FrameDev fd = tab.f[f];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    if (fd.m == 0) return;                                  // frame not Bloom-coded (passthrough)
    const uint32_t *mask = masks + (uint64_t)f * mask_stride_words32;
    uint32_t *filt = filters + (uint64_t)f * filter_stride_words32;
    const uint64_t nwords = (n + 31) >> 5;
    uint32_t *mylist = list[wave];

    for (uint64_t w0 = ((uint64_t)blockIdx.x * WG_WAVES + wave) * WAVE; w0 < nwords;
         w0 += (uint64_t)gridDim.x * WG_WAVES * WAVE) {
        const uint64_t w = w0 + lane;
        uint32_t bits = 0;
        if (w < nwords) {
            bits = flip_bytes32(mask[w]);                       // natural order
            const uint64_t last = n - (w << 5);                 // valid bits in this word
            if (last < 32) bits &= (1u << last) - 1u;           // ignore pad bits
        }
        // exclusive prefix of popcounts across the wave
        const uint32_t c = __popc(bits);
        uint32_t incl = c;
#pragma unroll
        for (int d = 1; d < WAVE; d <<= 1) {
            const uint32_t t = __shfl_up(incl, d);
            if (lane >= (uint32_t)d) incl += t;
        }
        const uint32_t total = __shfl(incl, WAVE - 1);
        uint32_t off = incl - c;
        const uint32_t base = (uint32_t)(w << 5);
        while (bits) {
            const uint32_t b = __builtin_ctz(bits);
            mylist[off++] = base + b;
            bits &= bits - 1u;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        for (uint32_t t = lane; t < total; t += WAVE) {
            Probe p = make_probe(mylist[t], fd, seeds);
            for (uint32_t j = 0; j < fd.floor_k; ++j) {
                atomicOr(&filt[p.pos >> 5], msb_bit(p.pos));
                advance(p, fd.m);
            }
            if (p.extra) atomicOr(&filt[p.pos >> 5], msb_bit(p.pos));
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
}

// ------------------------------------------------------------------------------------------
// A5 / A6  query: test every position in order (check_index, :116-138) -- generic path, filter
// probed in global memory.  One wave per segment of 1024 pixels:
//   pass_words[(f*nseg + seg)*16 + it]  pass word of wave-iteration it, PACKED like masks and witnesses
//                                       (numpy.packbits order: position 64*it + 8b + r at byte b, bit 7 - r)
//   seg_cnt[f*nseg + seg]               passing positions of the segment
// k_compact_witness (encode) / k_expand_mask_p (decode) consume them.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(WG_THREADS) void k_query(
    uint64_t n, const FrameTable tab, Seeds seeds,
    const uint32_t *__restrict__ filters, uint64_t filter_stride_words32,
    uint32_t *__restrict__ seg_cnt, uint64_t nseg, uint64_t *__restrict__ pass_words)
{
    const uint32_t f = blockIdx.y;
    const FrameDev fd = tab.f[f];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint64_t seg = (uint64_t)blockIdx.x * WG_WAVES + wave;
    if (seg >= nseg) return;
    uint64_t *pw_out = pass_words + ((uint64_t)f * nseg + seg) * SEG_ITERS;
    if (fd.m == 0) {                                        // passthrough frame: nothing passes
        if (lane == 0) seg_cnt[(uint64_t)f * nseg + seg] = 0;
        if (lane < SEG_ITERS) pw_out[lane] = 0;
        return;
    }
    const uint32_t *filt = filters + (uint64_t)f * filter_stride_words32;
    uint32_t woff = 0;
    const uint64_t base = seg * SEG_PIXELS;
    for (int it = 0; it < SEG_ITERS; ++it) {
        const uint64_t i64 = base + (uint64_t)it * WAVE + lane;
        bool pass = i64 < n;
        if (pass) {
            Probe p = make_probe((uint32_t)i64, fd, seeds);
            for (uint32_t j = 0; j < fd.floor_k; ++j) {
                pass = pass && ((filt[p.pos >> 5] >> msb_pos(p.pos)) & 1u);
                advance(p, fd.m);
            }
            if (p.extra) pass = pass && ((filt[p.pos >> 5] >> msb_pos(p.pos)) & 1u);
        }
        const uint64_t pw = __ballot(pass);
        if (lane == 0) pw_out[it] = flip_bytes64(pw);
        woff += __popcll(pw);
    }
    if (lane == 0) seg_cnt[(uint64_t)f * nseg + seg] = woff;
}

// Block-wide exclusive scan helper (blockDim.x == 1024): returns exclusive prefix, *total = sum.
__device__ __forceinline__ uint32_t block_excl_scan_1024(uint32_t v, uint32_t *smem /*[16]*/, uint32_t *total)
{
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t incl = v;
#pragma unroll
    for (int d = 1; d < WAVE; d <<= 1) {
        const uint32_t t = __shfl_up(incl, d);
        if (lane >= (uint32_t)d) incl += t;
    }
    __syncthreads();                     // protect smem reuse across calls
    if (lane == WAVE - 1) smem[wave] = incl;
    __syncthreads();
    uint32_t wave_off = 0, tot = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const uint32_t s = smem[k];
        if ((uint32_t)k < wave) wave_off += s;
        tot += s;
    }
    *total = tot;
    return wave_off + incl - v;
}

// Exclusive scan of the per-segment pass counts of every frame (one 1024-thread workgroup per
// frame): seg_off = bit offset of each segment in the frame's witness stream; totals (nullable)
// gets the stream length at totals[f * totals_stride].
__global__ __launch_bounds__(1024) void k_scan_segments(
    const uint32_t *__restrict__ seg_cnt, uint64_t *__restrict__ seg_off, uint64_t nseg,
    uint64_t *__restrict__ totals, uint32_t totals_stride)
{
    __shared__ uint32_t smem[16];
    const uint32_t f = blockIdx.x;
    const uint32_t *cnt = seg_cnt + (uint64_t)f * nseg;
    uint64_t *off = seg_off + (uint64_t)f * nseg;
    uint64_t carry = 0;
    for (uint64_t s0 = 0; s0 < nseg; s0 += 1024) {
        const uint64_t s = s0 + threadIdx.x;
        const uint32_t v = s < nseg ? cnt[s] : 0u;
        uint32_t tot;
        const uint32_t ex = block_excl_scan_1024(v, smem, &tot);
        if (s < nseg) off[s] = carry + ex;
        carry += tot;
    }
    if (totals && threadIdx.x == 0) totals[(uint64_t)f * totals_stride] = carry;
}

// ------------------------------------------------------------------------------------------
// per-index surface (RationalBloomFilter.add_index / check_index on arbitrary index lists)
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(WG_THREADS) void k_index_insert(
    uint32_t *__restrict__ filt, FrameDev fd, Seeds seeds, const uint32_t *__restrict__ idx, uint64_t count)
{
    for (uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; t < count; t += (uint64_t)gridDim.x * blockDim.x) {
        Probe p = make_probe(idx[t], fd, seeds);
        for (uint32_t j = 0; j < fd.floor_k; ++j) {
            atomicOr(&filt[p.pos >> 5], msb_bit(p.pos));
            advance(p, fd.m);
        }
        if (p.extra) atomicOr(&filt[p.pos >> 5], msb_bit(p.pos));
    }
}

__global__ __launch_bounds__(WG_THREADS) void k_index_query(
    const uint32_t *__restrict__ filt, FrameDev fd, Seeds seeds, const uint32_t *__restrict__ idx, uint64_t count,
    uint8_t *__restrict__ out)
{
    for (uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; t < count; t += (uint64_t)gridDim.x * blockDim.x) {
        Probe p = make_probe(idx[t], fd, seeds);
        bool pass = true;
        for (uint32_t j = 0; j < fd.floor_k; ++j) {
            pass = pass && ((filt[p.pos >> 5] >> msb_pos(p.pos)) & 1u);
            advance(p, fd.m);
        }
        if (p.extra) pass = pass && ((filt[p.pos >> 5] >> msb_pos(p.pos)) & 1u);
        out[t] = pass ? 1 : 0;
    }
}

// ------------------------------------------------------------------------------------------
// string-keyed twins (rational_bloom_filter.py): keys are arbitrary byte strings, so the whole XXH64
// algorithm is needed (stripe loop for >= 32 bytes).  One thread per key; byte loads (keys are short).
//   standard_k == 0 : RationalBloomFilter.add / contains  (:139-182), double hashing + activation
//   standard_k  > 0 : StandardBloomFilter.add / contains  (:29-41), k hashes with seed = j, `% m`
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t load_le(const uint8_t *p, int nbytes)
{
    uint64_t v = 0;
    for (int b = 0; b < nbytes; ++b) v |= (uint64_t)p[b] << (8 * b);
    return v;
}

__device__ __forceinline__ uint64_t xxh_round(uint64_t acc, uint64_t input)
{
    acc += input * P2;
    return rotl64(acc, 31) * P1;
}

__device__ inline uint64_t xxh64_bytes(const uint8_t *p, uint32_t len, uint64_t seed)
{
    const uint8_t *end = p + len;
    uint64_t h;
    if (len >= 32) {
        uint64_t v1 = seed + P1 + P2, v2 = seed + P2, v3 = seed, v4 = seed - P1;
        do {
            v1 = xxh_round(v1, load_le(p, 8)); p += 8;
            v2 = xxh_round(v2, load_le(p, 8)); p += 8;
            v3 = xxh_round(v3, load_le(p, 8)); p += 8;
            v4 = xxh_round(v4, load_le(p, 8)); p += 8;
        } while (p + 32 <= end);
        h = rotl64(v1, 1) + rotl64(v2, 7) + rotl64(v3, 12) + rotl64(v4, 18);
        h = (h ^ xxh_round(0, v1)) * P1 + P4;
        h = (h ^ xxh_round(0, v2)) * P1 + P4;
        h = (h ^ xxh_round(0, v3)) * P1 + P4;
        h = (h ^ xxh_round(0, v4)) * P1 + P4;
    } else {
        h = seed + P5;
    }
    h += (uint64_t)len;
    while (p + 8 <= end) { h ^= xxh_round(0, load_le(p, 8)); h = rotl64(h, 27) * P1 + P4; p += 8; }
    if (p + 4 <= end) { h ^= load_le(p, 4) * P1; h = rotl64(h, 23) * P2 + P3; p += 4; }
    while (p < end) { h ^= (uint64_t)(*p) * P5; h = rotl64(h, 11) * P1; ++p; }
    h ^= h >> 33; h *= P2;
    h ^= h >> 29; h *= P3;
    h ^= h >> 32;
    return h;
}

template <bool INSERT>
__global__ __launch_bounds__(WG_THREADS) void k_keys(
    uint32_t *__restrict__ filt, FrameDev fd, Seeds seeds, uint32_t standard_k,
    const uint8_t *__restrict__ bytes, const uint32_t *__restrict__ offsets, uint64_t count, uint8_t *__restrict__ out)
{
    for (uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; t < count; t += (uint64_t)gridDim.x * blockDim.x) {
        const uint8_t *key = bytes + offsets[t];
        const uint32_t len = offsets[t + 1] - offsets[t];
        bool pass = true;
        if (standard_k) {
            for (uint32_t j = 0; j < standard_k; ++j) {
                const uint32_t pos = mod_m(xxh64_bytes(key, len, (uint64_t)j), fd.m, fd.M);
                if (INSERT) atomicOr(&filt[pos >> 5], msb_bit(pos));
                else pass = pass && ((filt[pos >> 5] >> msb_pos(pos)) & 1u);
            }
        } else {
            Probe p;
            p.pos = mod_m(xxh64_bytes(key, len, seeds.h1), fd.m, fd.M);
            p.step = mod_m(xxh64_bytes(key, len, seeds.h2), fd.m, fd.M);
            p.extra = xxh64_bytes(key, len, seeds.act) < fd.T;
            for (uint32_t j = 0; j < fd.floor_k; ++j) {
                if (INSERT) atomicOr(&filt[p.pos >> 5], msb_bit(p.pos));
                else pass = pass && ((filt[p.pos >> 5] >> msb_pos(p.pos)) & 1u);
                advance(p, fd.m);
            }
            if (p.extra) {
                if (INSERT) atomicOr(&filt[p.pos >> 5], msb_bit(p.pos));
                else pass = pass && ((filt[p.pos >> 5] >> msb_pos(p.pos)) & 1u);
            }
        }
        if (!INSERT) out[t] = pass ? 1 : 0;
    }
}

// ------------------------------------------------------------------------------------------
// A2 / A8  changed-value gather / scatter in raster order (:811-842, :886-903)
// ------------------------------------------------------------------------------------------
// segment popcounts of packed masks (one wave per segment; blockIdx.y = mask of the batch)
__global__ __launch_bounds__(WG_THREADS) void k_mask_segment_counts(
    const uint64_t *__restrict__ masks, uint64_t mask_stride_words64, uint64_t n, uint32_t *__restrict__ seg_cnt, uint64_t nseg)
{
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint64_t seg = (uint64_t)blockIdx.x * WG_WAVES + wave;
    if (seg >= nseg) return;
    const uint64_t *mask = masks + (uint64_t)blockIdx.y * mask_stride_words64;
    const uint64_t nwords = (n + 63) >> 6;
    const uint64_t w = seg * SEG_ITERS + lane;
    uint32_t c = (lane < SEG_ITERS && w < nwords) ? __popcll(mask[w]) : 0u;
#pragma unroll
    for (int d = 8; d >= 1; d >>= 1) c += __shfl_down(c, d);
    if (lane == 0) seg_cnt[(uint64_t)blockIdx.y * nseg + seg] = c;
}

template <typename SAMPLE, bool SCATTER>
__global__ __launch_bounds__(WG_THREADS) void k_values(
    uint8_t *__restrict__ frame, uint32_t width, uint64_t n, uint64_t row_pitch, uint32_t pixel_stride,
    uint32_t channels, const uint64_t *__restrict__ mask, const uint64_t *__restrict__ seg_off, uint64_t nseg,
    SAMPLE *__restrict__ values)
{
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint64_t seg = (uint64_t)blockIdx.x * WG_WAVES + wave;
    if (seg >= nseg) return;
    const uint64_t nwords = (n + 63) >> 6;
    uint64_t o = seg_off[seg];
    for (int it = 0; it < SEG_ITERS; ++it) {
        const uint64_t w = seg * SEG_ITERS + it;
        if (w >= nwords) break;
        const uint64_t p = flip_bytes64(mask[w]);          // natural order
        if ((p >> lane) & 1ull) {
            const uint64_t i = w * 64 + lane;
            const uint64_t y = i / width, x = i - y * width;
            SAMPLE *px = (SAMPLE *)(frame + y * row_pitch + x * pixel_stride);
            SAMPLE *v = values + (o + rank_below(p)) * channels;
            for (uint32_t c = 0; c < channels; ++c) {
                if (SCATTER) px[c] = v[c]; else v[c] = px[c];
            }
        }
        o += __popcll(p);
    }
}

// ------------------------------------------------------------------------------------------
// A2 for a whole GOP: the changed values of every pair, concatenated in frame order
// ------------------------------------------------------------------------------------------
// exclusive scan of a short array by one thread (one entry per frame pair)
__global__ void k_frame_offsets(const uint64_t *__restrict__ totals, uint64_t *__restrict__ offsets, uint32_t count)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        uint64_t acc = 0;
        for (uint32_t i = 0; i < count; ++i) { offsets[i] = acc; acc += totals[i]; }
        offsets[count] = acc;
    }
}

// One lane per 64-pixel mask word of pair f = blockIdx.y; the pair's values come from frame f+1 (`curr`, :811-842)
// and land at (frame_off[f] + rank of the pixel within the mask) * channels.  Nothing is written past `capacity`
// pixels (the caller sizes the buffer from the ones counts it already has).
template <typename SAMPLE>
__global__ __launch_bounds__(WG_THREADS) void k_gather_words(
    const uint8_t *__restrict__ frames, uint64_t frame_stride, uint32_t width, uint64_t n, uint64_t row_pitch, uint32_t pixel_stride,
    uint32_t channels, const uint64_t *__restrict__ masks, uint64_t mask_stride_words64,
    const uint64_t *__restrict__ seg_off, uint64_t nseg, const uint64_t *__restrict__ frame_off,
    SAMPLE *__restrict__ values, uint64_t capacity)
{
    const uint32_t f = blockIdx.y;
    const uint64_t nwords = (n + 63) >> 6;
    const uint64_t w = (uint64_t)blockIdx.x * WG_THREADS + threadIdx.x;
    if (w >= nwords) return;
    const uint64_t *mask = masks + (uint64_t)f * mask_stride_words64;
    uint64_t p = flip_bytes64(mask[w]);                           // natural order: bit b = pixel 64w + b
    if (!p) return;
    const uint64_t seg = w / SEG_ITERS;
    uint64_t o = frame_off[f] + seg_off[(uint64_t)f * nseg + seg];
    for (uint64_t j = seg * SEG_ITERS; j < w; ++j) o += __popcll(mask[j]);
    const uint8_t *cur = frames + (uint64_t)(f + 1) * frame_stride;
    const bool flat = row_pitch == (uint64_t)width * pixel_stride;
    while (p) {
        const uint32_t b = __builtin_ctzll(p);
        p &= p - 1;
        const uint64_t i = w * 64 + b;
        uint64_t off;
        if (flat) off = i * pixel_stride;
        else { const uint64_t y = i / width; off = y * row_pitch + (i - y * width) * pixel_stride; }
        if (o < capacity) {
            const SAMPLE *px = (const SAMPLE *)(cur + off);
            SAMPLE *v = values + o * channels;
            for (uint32_t c = 0; c < channels; ++c) v[c] = px[c];
        }
        ++o;
    }
}

// Pixels whose mask bit is 0 although some channel differs between frame f and f+1: the residual the
// luma-only mask cannot carry (SURVEY 8a row A8 "lossy by construction").  uncovered[f] gets their number.
template <typename SAMPLE>
__global__ __launch_bounds__(WG_THREADS) void k_uncovered_changes(
    const uint8_t *__restrict__ frames, uint64_t frame_stride, uint32_t width, uint64_t n, uint64_t row_pitch, uint32_t pixel_stride,
    uint32_t channels, const uint64_t *__restrict__ masks, uint64_t mask_stride_words64, unsigned long long *__restrict__ uncovered)
{
    const uint32_t f = blockIdx.y, lane = threadIdx.x & 63u;
    const uint8_t *prev = frames + (uint64_t)f * frame_stride, *cur = prev + frame_stride;
    const uint64_t *mask = masks + (uint64_t)f * mask_stride_words64;
    const bool flat = row_pitch == (uint64_t)width * pixel_stride;
    uint32_t cnt = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * WG_THREADS + threadIdx.x; i < ((n + 63) & ~63ull); i += (uint64_t)gridDim.x * WG_THREADS) {
        bool bad = false;
        if (i < n) {
            uint64_t off;
            if (flat) off = i * pixel_stride;
            else { const uint64_t y = i / width; off = y * row_pitch + (i - y * width) * pixel_stride; }
            const SAMPLE *a = (const SAMPLE *)(prev + off), *b = (const SAMPLE *)(cur + off);
            bool diff = false;
            for (uint32_t c = 0; c < channels; ++c) diff |= a[c] != b[c];
            const uint64_t word = flip_bytes64(mask[i >> 6]);
            bad = diff && !((word >> (i & 63)) & 1ull);
        }
        cnt += __popcll(__ballot(bad));
    }
    if (lane == 0 && cnt) atomicAdd(&uncovered[f], (unsigned long long)cnt);
}

}  // namespace rbf
