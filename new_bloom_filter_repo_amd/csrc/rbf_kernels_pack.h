// Exact-size output record of a batch: the padded per-frame rows (filters | witnesses) an encode
// leaves in HBM are compacted ON THE DEVICE into one contiguous, self-describing block, so that the
// multi-GPU gather to rank 0 moves the ~150 KB a 1080p frame really needs instead of
// its 340 KB worst-case slot, without the host ever learning the witness lengths.
//
// Layout (little-endian uint64 words):
//   [0] RECORD_MAGIC  [1] nframes  [2] used bytes (header + payload)  [3] 1 if the block was too small
//   then one 8-word row per frame:
//       m  floor_k  threshold  k (float64 bits)  witness_bits  filter_ones  filter_offset  witness_offset
//   then the payload rows, each 8-byte aligned at its byte offset from the block start:
//       filter  ceil(m/64)*8 bytes            (m == 0, a frame the reference does not Bloom-code
//                                              (:215-225): the packed mask itself, ceil(n/64)*8 bytes;
//                                              m == 0 and floor_k == 0xFFFFFFFF, a pair across a keyframe
//                                              of a multi-run block (rbf_encode_runs): nothing)
//       witness ceil(witness_bits/64)*8 bytes
#pragma once
#include "rbf_device.h"

namespace rbf {

constexpr uint64_t RECORD_MAGIC = 0x3130434552464252ull;        // "RBFREC01"
constexpr int RECORD_HEADER_WORDS = 4, RECORD_ROW_WORDS = 8;
constexpr int PACK_BATCH = 128;
constexpr uint32_t PACK_PAIR_SKIPPED = 0xFFFFFFFFu;             // = RBF_PAIR_SKIPPED (include/rbf.h), in the floor_k field of a row with m == 0

struct PackRow {
    uint32_t m, floor_k;
    uint64_t threshold;
    uint64_t k_bits;
};
struct PackTable { PackRow r[PACK_BATCH]; };                    // 3 KiB of kernel arguments

// grid (x, 2*count): blockIdx.y = 2*frame + {0: filter or passthrough mask, 1: witness}.  Every workgroup
// recomputes the (cheap) offset scan of its chunk of <= 128 frames, so one launch does header + copy;
// workgroup (0, 0) writes the header rows.  Chunks after the first continue from chunk_base[chunk],
// which the previous launch's workgroup (0, 0) left there (a different word from the one it reads).
__global__ __launch_bounds__(256) void k_pack_records(
    const PackTable tab, uint32_t first, uint32_t count, uint32_t nframes, uint64_t n,
    const uint64_t *__restrict__ stats,
    const uint8_t *__restrict__ masks, uint64_t mask_stride,
    const uint8_t *__restrict__ filters, uint64_t filter_stride,
    const uint8_t *__restrict__ witnesses, uint64_t witness_stride,
    uint64_t *__restrict__ record, uint64_t capacity, uint64_t *__restrict__ chunk_base)
{
    __shared__ uint64_t wave_sum[PACK_BATCH / WAVE];
    __shared__ uint64_t row_off[PACK_BATCH], row_fbytes[PACK_BATCH], row_wbits[PACK_BATCH];
    const uint32_t t = threadIdx.x, lane = t & 63u, wave = t >> 6;
    const uint32_t chunk = first / PACK_BATCH;
    const uint64_t header_bytes = (uint64_t)(RECORD_HEADER_WORDS + RECORD_ROW_WORDS * (uint64_t)nframes) * 8;
    const uint64_t base = first == 0 ? header_bytes : chunk_base[chunk];
    uint64_t fbytes = 0, wbytes = 0, wbits = 0;
    if (t < count) {
        wbits = stats[(uint64_t)(first + t) * 4 + 0];
        const uint32_t m = tab.r[t].m;
        const bool skipped = m == 0 && tab.r[t].floor_k == PACK_PAIR_SKIPPED;    // rbf_encode_runs: a pair across a keyframe has no payload at all
        fbytes = skipped ? 0 : ((m ? (uint64_t)m : n) + 63) / 64 * 8;
        wbytes = (wbits + 63) / 64 * 8;
    }
    uint64_t incl = fbytes + wbytes;
    if (t < PACK_BATCH) {
#pragma unroll
        for (int d = 1; d < WAVE; d <<= 1) {
            const uint64_t v = __shfl_up(incl, d);
            if (lane >= (uint32_t)d) incl += v;
        }
        if (lane == WAVE - 1) wave_sum[wave] = incl;
    }
    __syncthreads();
    if (t < PACK_BATCH) {
        uint64_t off = base + incl - (fbytes + wbytes);
        for (uint32_t w = 0; w < wave; ++w) off += wave_sum[w];
        row_off[t] = off; row_fbytes[t] = fbytes; row_wbits[t] = wbits;
    }
    __syncthreads();
    if (blockIdx.x == 0 && blockIdx.y == 0) {
        if (t < count) {
            const PackRow row = tab.r[t];
            uint64_t *r = record + RECORD_HEADER_WORDS + (uint64_t)(first + t) * RECORD_ROW_WORDS;
            r[0] = row.m; r[1] = row.floor_k; r[2] = row.threshold; r[3] = row.k_bits;
            r[4] = wbits; r[5] = stats[(uint64_t)(first + t) * 4 + 1]; r[6] = row_off[t]; r[7] = row_off[t] + fbytes;
        }
        if (t == 0) {
            uint64_t total = base;
            for (uint32_t w = 0; w < PACK_BATCH / WAVE; ++w) total += wave_sum[w];
            if (first == 0) { record[0] = RECORD_MAGIC; record[1] = nframes; record[3] = 0; }
            record[2] = total;
            chunk_base[chunk + 1] = total;
            if (total > capacity) record[3] = 1;
        }
    }
    const uint32_t fl = blockIdx.y >> 1, which = blockIdx.y & 1u, f = first + fl;
    const uint32_t m = tab.r[fl].m;
    const uint8_t *src;
    uint64_t bytes, off;
    if (which == 0) {
        src = m ? filters + (uint64_t)f * filter_stride : masks + (uint64_t)f * mask_stride;
        bytes = row_fbytes[fl];
        off = row_off[fl];
    } else {
        src = witnesses + (uint64_t)f * witness_stride;
        bytes = (row_wbits[fl] + 63) / 64 * 8;
        off = row_off[fl] + row_fbytes[fl];
    }
    if (off + bytes > capacity) return;                         // flagged in the header
    const uint64_t *s = reinterpret_cast<const uint64_t *>(src);
    uint64_t *d = record + off / 8;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + t; i < bytes / 8; i += (uint64_t)gridDim.x * blockDim.x)
        d[i] = s[i];
}

}  // namespace rbf
