// rbf_kernels_i64.h -- the insert path for filters of 2^15 <= m < 2^23 bits (1080p / 2160p frames):
//
//   k_hash_table   the three XXH64 of EVERY pixel index of the frame geometry (they depend on the index and the seeds
//                  only, not on the frame: improved_video_compressor.py:77-78,94).  Runs for the FIRST batch of a
//                  geometry; afterwards a context that holds the table alone has k_query_u64, which computes the same hashes for
//                  its own probes in every batch, write it again for the following batch's insert (to keep it cached).  Layout:
//                  rbf_kernels_q64.h (16-byte (h1, h2) entries, 16-bit activation tags, the full h_act for ties).  A lane
//                  owns 8 consecutive indices, so the decade-prefix sharing of hash3_run8 applies (~120 instead of ~600
//                  instructions per index), and over a 29-frame batch 93 % of all indices are set in at least one frame, so
//                  nothing is hashed in vain.
//   k_insert_tab   k_insert_lds with the hashing replaced by ONE 16-byte gather (+ a 2-byte tag) from that table: workgroup
//                  (slice, frame[, tile]) builds a partial filter in LDS from its slice of the mask; set positions are compacted
//                  through a per-wave LDS queue so that the gather, the two reductions (mod_m_f64) and the LDS atomics
//                  always run on full waves; the gather of one batch of 64 keys flies while the next mask bytes are
//                  compacted.  All workgroups of a slice run on the same XCD (slice = blockIdx % 8 when a frame has 8
//                  slices), so the table lines of a slice are fetched from HBM once and then hit that XCD's L2 for the
//                  other frames.
//
// In k_insert_lds the three hashes of the p*n set positions cost ~33 of its ~70 us per 1080p x 29 batch (64-bit
// multiplies: tools/bench_insert.hip ablation); the table costs one ~54 MB write, once.
#pragma once
#include "rbf_kernels_q64.h"

namespace rbf {

constexpr int HT_THREADS = 256;

__global__ __launch_bounds__(HT_THREADS) void k_hash_table(uint64_t n, Seeds seeds, uint4 *__restrict__ table)
{
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint64_t seg = (uint64_t)blockIdx.x * (HT_THREADS / WAVE) + wave;
    if (seg * QL_SEG_PIXELS >= n) return;                        // the grid is rounded up to 4 segments; the table is not
    const uint64_t i0 = seg * QL_SEG_PIXELS + (uint64_t)lane * QL_P;
    uint64_t h1[QL_P], h2[QL_P], ha[QL_P];
    uint32_t validmask = 0;
#pragma unroll
    for (int it = 0; it < QL_P; ++it) {
        h1[it] = 0; h2[it] = 0; ha[it] = 0;
        if (i0 + it < n) validmask |= 1u << it;
    }
    if (!hash3_run8((uint32_t)i0, validmask, seeds, h1, h2, ha)) {
#pragma unroll
        for (int it = 0; it < QL_P; ++it) {                      // mixed key lengths in this wave: index by index
            const bool act = (validmask >> it) & 1u;
            const Hash3 h = hash3_index((uint32_t)(i0 + it), act, seeds);
            h1[it] = h.h1; h2[it] = h.h2; ha[it] = h.ha;
        }
    }
    // entries past the end of the frame are written too (the table is padded to whole segments): they are never read
    hash_table_store8(table, n, seg, lane, h1, h2, ha);
}

constexpr int IT_STEP_BYTES = 128;                 // mask bytes per wave step: a lane owns 16 pixels (two bytes)
constexpr int IT_CHUNK_STEPS = 16;                 // wave steps whose mask bytes are staged in LDS at a time
constexpr int IT_QUEUE = 64 + IT_STEP_BYTES * 8;   // queue entries (16 bits each): carry (< 64) + one wave step
constexpr int IT_STAGE_BYTES = IT_CHUNK_STEPS * IT_STEP_BYTES;
constexpr int IT_WAVE_LDS_BYTES = IT_QUEUE * 2 + IT_STAGE_BYTES;      // per wave: the queue, then the staged mask bytes (4224)
static_assert(IT_WAVE_LDS_BYTES % 16 == 0 && (IT_QUEUE * 2) % 16 == 0, "the stage is written by LDS-DMA");

// `fd`: M carries the bits of -1.0 / m (IEEE double, computed on the host) instead of the Barrett constant.
//
// The body shared by the two table-driven kernels that walk a mask: this wave takes wave steps g0 + wave, + NWAVES, ... < g1,
// compacts the set positions through its LDS queue and, for every batch of <= 64 keys, gathers the table entries and
// reduces them.  RECORDS = false (k_insert_tab): the probe positions are OR-ed into the LDS tile `filt` covering bits
// [tile_bit0, tile_bit0 + tile_bits); WHOLE: the tile is the whole filter (every frame whose filter fits LDS next to the queues),
// no in-tile test per probe.  RECORDS = true (k_insert_positions): the batch is appended to the frame's list of
// InsertRecords, 64 contiguous records per batch, starting at records[rpos] (this wave's own range of the list).
// HASHED: no table -- the batch's three hashes are computed on the spot (hash3_index, as k_insert_lds does).  Cheaper than the
// gather once the table no longer fits the 256 MB Infinity Cache: at 2160p the gather of a GOP's 5.9 M
// entries costs 58 us of random HBM reads, hashing them ~35.
//
// THE MASK BYTES COME THROUGH LDS (round 4).  A wave's steps are staged IT_CHUNK_STEPS at a time by LDS-DMA (eight
// global_load_lds_dword per chunk, no registers) and then read with ds_read_u16.  Rounds 2 and 3 loaded every step's two bytes per
// lane with a global load "one step ahead" -- but vmcnt counts the mask loads and the table gathers in ONE in-order queue, and with
// a data-dependent number of gathers between two mask loads the compiler can only wait for a mask load with vmcnt(0)/(1): every step
// drained the gather it had just issued, and (the load sitting inside `if (in range)`) waited for its own prefetch as well.  Neither
// the "gather flies under the next step's compaction" nor the prefetch ever happened: ~28 exposed L2 round trips per wave.  With the
// mask bytes on the LDS counter the only vector-memory loads left in the loop are the gathers, and the wait in front of `finish` is
// the only one.  The queue holds 16-bit entries (step of the chunk << 10 | bit of the step) to make room for the stage; the carry is
// flushed as a partial batch at the end of a chunk (1080p: one chunk per wave).
template <bool RECORDS, int NWAVES, bool HASHED = false, bool WHOLE = false>
__device__ __forceinline__ void insert_tab_steps(const uint8_t *__restrict__ mask /* wave-uniform */, uint64_t row_bytes /* the row pitch: bytes that may be read */, uint64_t n,
                                                 const FrameDev &fd, const uint4 *__restrict__ table, const Seeds &seeds,
                                                 uint32_t *filt, uint32_t tile_bit0, uint32_t tile_bits, uint2 *__restrict__ records, uint32_t rpos,
                                                 uint32_t *wave_lds /* IT_WAVE_LDS_BYTES of this wave */, uint64_t g0, uint64_t g1, uint32_t lane, uint32_t wave)
{
    static_assert((NWAVES & (NWAVES - 1)) == 0, "a power of two");
    uint16_t *q = reinterpret_cast<uint16_t *>(wave_lds);
    const uint32_t stage = __builtin_amdgcn_readfirstlane(lds_addr_of(wave_lds) + IT_QUEUE * 2);      // LDS byte address of the staged steps
    uint32_t qn = 0;                                               // wave-uniform queue length
    const uint32_t m = vgpr_copy(__builtin_amdgcn_readfirstlane(fd.m));
    const uint32_t fk = __builtin_amdgcn_readfirstlane(fd.floor_k);
    const double ninv = __builtin_bit_cast(double, fd.M);
    const uint64_t T = fd.T;
    auto set_bit = [&](uint32_t pos) {
        if (WHOLE) { atomicOr(&filt[pos >> 5], msb_bit(pos)); return; }
        const uint32_t rel = pos - tile_bit0;                      // unsigned: out-of-tile positions wrap high
        if (rel < tile_bits) atomicOr(&filt[rel >> 5], msb_bit(pos));
    };

    // one batch of <= 64 keys in flight: its table entries are requested (`fetch`) when the batch leaves the queue and
    // consumed (`finish`) when the next batch is ready -- or at the end -- so the gather latency hides under compaction
    const HashTable tv(const_cast<uint4 *>(table), n);
    const uint32_t Ttag = (uint32_t)(T >> 48);
    uint4 e0 = make_uint4(0, 0, 0, 0);                             // h1, h2 of my key of the batch in flight
    uint32_t tag = 0, slot_kept = 0;                               // h_act >> 48 | where the full h_act is
    uint64_t ha_kept = 0;                                          // (HASHED: the full h_act itself)
    uint32_t pending = 0;                                          // keys of the batch in flight (wave-uniform)
    uint32_t cb = 0;                                               // first step of the chunk being walked
    auto fetch = [&](uint32_t first, uint32_t count) {
        const uint32_t e = lane < count ? q[first + lane] : 0u;    // idle lanes read the chunk's first pixel's entry (always there)
        const uint32_t idx = ((cb + (e >> 10) * NWAVES) << 10) + (e & 1023u);
        if (HASHED) {
            const Hash3 h = hash3_index(idx, lane < count, seeds);      // wave-uniform call (it votes on the key length)
            e0 = make_uint4((uint32_t)h.h1, (uint32_t)(h.h1 >> 32), (uint32_t)h.h2, (uint32_t)(h.h2 >> 32));    // the table's entry format
            tag = (uint32_t)(h.ha >> 48);
            ha_kept = h.ha;
        } else {
            slot_kept = hash_table_slot(idx);
            e0 = tv.pos[slot_kept];                                // (round 5 measured a non-temporal gather: insert 33 -> 55 us alone, profiles/r05_sweep1.txt; and batches of 128 / 256 keys,
                                                                   //  two / four gathers per lane in flight: 35.3 / 36.1 us, the step unchanged, profiles/r05_insert_batch_keys.txt)
            tag = tv.tag[idx];
        }
        pending = count;
    };
    auto finish = [&]() {
        if (!pending) return;
        if (lane < pending) {
            uint32_t pos = mod_m_f64(rn_double(e0.x, e0.y), e0.x, ninv, m);
            const uint32_t step = mod_m_f64(rn_double(e0.z, e0.w), e0.z, ninv, m);
            bool act = tag < Ttag;
            if (tag == Ttag) act = (HASHED ? ha_kept : tv.act[slot_kept]) < T;      // 2^-16 of the keys
            if (RECORDS) {
                records[rpos + lane] = make_uint2(pos, step | (act ? 0x80000000u : 0u));
            } else {
                for (uint32_t j = 0; j < fk; ++j) {
                    set_bit(pos);
                    const uint32_t s2 = pos + step;
                    pos = min(s2, s2 - m);
                }
                if (act) set_bit(pos);
            }
        }
        if (RECORDS) rpos += pending;
        pending = 0;
    };

    const uint32_t nbytes32 = (uint32_t)((n + 7) >> 3), n32 = (uint32_t)n, row32 = (uint32_t)row_bytes;      // n < 2^32 (rbf_plan_batch)
    const uint32_t gend = (uint32_t)g1;
    // my two bytes of step `s` of the chunk as 16 bits in natural order (bit j = pixel 16 * lane + j of the step)
    auto staged = [&](uint32_t s) -> uint32_t {
        return *reinterpret_cast<const __attribute__((address_space(3))) uint16_t *>((uintptr_t)(stage + s * IT_STEP_BYTES + lane * 2u));
    };
    auto decode = [&](uint32_t v, uint32_t g) -> uint32_t {        // byte 0 = pixels 0..7 MSB-first, byte 1 = pixels 8..15
        const uint32_t byte = g * IT_STEP_BYTES + lane * 2u;
        const uint32_t x = __builtin_bitreverse32(v) >> 16;        // bits 8..15 = byte 0 reversed, bits 0..7 = byte 1 reversed
        uint32_t b = ((x & 0xFFu) << 8) | (x >> 8);
        const uint32_t rem = n32 - byte * 8u;
        if (rem < 16u) b &= (1u << rem) - 1u;                      // ignore pad bits
        return (g < gend && byte < nbytes32) ? b : 0u;             // (steps past the slice and bytes past the row were not staged)
    };
    for (uint32_t chunk = __builtin_amdgcn_readfirstlane((uint32_t)g0 + wave); chunk < gend; chunk += NWAVES * IT_CHUNK_STEPS) {
        cb = chunk;                                                // (the entries still queued after the loop belong to the last chunk)
        // stage the chunk: instruction i brings steps 2i and 2i + 1 (lanes 0-31 / 32-63, a dword each); rows are padded to 8 bytes and
        // `row_bytes` is a multiple of 8, so a dword that starts inside the row ends inside it
        wave_lds_fence();                                          // the previous chunk's reads are done
#pragma unroll
        for (int i = 0; i < IT_CHUNK_STEPS / 2; ++i) {
            const uint32_t g = cb + (2u * i + (lane >> 5)) * NWAVES;
            const uint32_t off = g * IT_STEP_BYTES + (lane & 31u) * 4u;
            if (g < gend && off < row32) {
                const uint32_t dst = stage + (uint32_t)i * 256u;
                uint32_t keep;
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %2, %3\n\ts_mov_b32 m0, %0"
                             : "=&s"(keep) : "s"(dst), "v"(off), "s"(mask) : "memory");
            }
        }
        dma_wait_all();                                            // (also the batch in flight: once per chunk)
        wave_lds_fence();
        uint32_t nxt = staged(0);
        for (uint32_t s = 0; s < IT_CHUNK_STEPS && cb + s * NWAVES < gend; ++s) {
            const uint32_t g = cb + s * NWAVES;
            uint32_t bits = decode(nxt, g);
            nxt = staged(s + 1 < IT_CHUNK_STEPS ? s + 1 : s);
            // exclusive prefix of the per-lane counts (0..16): six DPP adds (the five ballots + ten mbcnt of k_insert_lds were a
            // third of this loop's skeleton)
            const uint32_t c = __popc(bits);
            const uint32_t incl = wave_inclusive_scan(c);
            const uint32_t excl = incl - c;
            const uint32_t total = __builtin_amdgcn_readlane(incl, 63);
            uint32_t off = qn + excl;
            const uint32_t ebase = (s << 10) | (lane << 4);
            while (bits) {
                q[off++] = (uint16_t)(ebase + __builtin_ctz(bits));
                bits &= bits - 1u;
            }
            qn += total;
            wave_lds_fence();
            while (qn >= WAVE) {                                   // full waves only; order is irrelevant (OR)
                qn -= WAVE;
                finish();
                fetch(qn, WAVE);
            }
            wave_lds_fence();                                      // queue reads done before it is refilled
        }
        if (cb + NWAVES * IT_CHUNK_STEPS < gend && qn) {           // another chunk follows: its entries count from a new `cb`
            finish();
            fetch(0, qn);
            qn = 0;
            wave_lds_fence();
        }
    }
    finish();
    if (qn) { fetch(0, qn); finish(); }
}

template <bool HASHED = false, bool WHOLE = false>
__global__ __launch_bounds__(IL_THREADS) void k_insert_tab(
    const uint8_t *__restrict__ masks, uint64_t mask_stride_bytes, uint64_t n,
    const FrameTable tab, const uint4 *__restrict__ table /* unused when HASHED: the set positions are hashed on the spot */, Seeds seeds,
    uint32_t *__restrict__ partials, uint64_t part_stride_words32, uint32_t tile_words /* even */,
    const SliceTable slices, uint32_t per_tile /* sum of slices.n */, uint32_t Smax /* max of slices.n: row pitch of the partials */)
{
    // workgroup -> (tile, frame, slice) exactly as in k_insert_lds (one-dimensional grid, slice fastest)
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    uint32_t *filt = lds;                                         // [tile_words]
    uint32_t *queues = lds + tile_words;                          // [IL_WAVES][IT_WAVE_LDS_BYTES]: queue + staged mask bytes per wave (tile_words is a multiple of 4)
    const uint32_t tile = blockIdx.x / per_tile;
    uint32_t s = blockIdx.x - tile * per_tile, f = 0;
    while (s >= slices.n[f]) { s -= slices.n[f]; ++f; }
    const uint32_t S = slices.n[f];
    const FrameDev fd = tab.f[f];
    if (fd.m == 0) return;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t fwords = filter_words(fd.m);
    const uint32_t tile0 = tile * tile_words;
    if (tile0 >= fwords) return;
    for (uint32_t i = threadIdx.x; i < tile_words; i += IL_THREADS) filt[i] = 0;
    __syncthreads();

    const uint64_t nbytes = (n + 7) >> 3;
    const uint64_t groups = (nbytes + IT_STEP_BYTES - 1) / IT_STEP_BYTES;   // wave steps of 128 mask bytes = 1024 pixels
    const uint64_t gper = (groups + S - 1) / S;
    const uint64_t g0 = (uint64_t)s * gper;
    const uint64_t g1 = g0 + gper < groups ? g0 + gper : groups;
    insert_tab_steps<false, IL_WAVES, HASHED, WHOLE>(masks + (uint64_t)f * mask_stride_bytes, mask_stride_bytes, n, fd, table, seeds, filt, tile0 << 5, tile_words << 5, nullptr, 0u,
                                                     queues + wave * (IT_WAVE_LDS_BYTES / 4), g0, g1, lane, wave);
    __syncthreads();
    uint32_t *part = partials + ((uint64_t)f * Smax + s) * part_stride_words32 + tile0;
    const uint32_t mine = fwords - tile0 < tile_words ? fwords - tile0 : tile_words;
    const uint32_t pairs = (mine + 1) >> 1;                       // tile0 is even: 8-byte aligned
    for (uint32_t i = threadIdx.x; i < pairs; i += IL_THREADS)
        reinterpret_cast<uint2 *>(part)[i] = reinterpret_cast<const uint2 *>(filt)[i];
}

// ------------------------------------------------------------------------------------------
// Filters that do not fit one LDS tile next to the queues (1440p and up).  k_insert_tab walks the mask, compacts and
// gathers the table once PER TILE (2160p: 3 tiles, 166 us for 8 frames; the 265 MB table no longer sits in the 256 MB L3).
// Two kernels instead:
//   k_insert_positions   one walk: (slice, frame) workgroups of 256 threads compact + gather + reduce as above and append
//                        one 8-byte InsertRecord per set position -- x = first probe position, y = probe step | activated
//                        extra probe << 31 -- to the frame's list, in no particular order (the filter is an OR).  A
//                        workgroup first counts the set bits of its slice (the mask row is L2-resident) and reserves its
//                        range of the list with ONE atomicAdd; every wave then owns a contiguous sub-range.  (One atomicAdd
//                        per 64-record batch was measured first: same-address atomics with return complete every ~90 ns,
//                        1.07 ms for the 92 k batches of a 2160p GOP.)
//   k_insert_records     (tile, frame, slice) workgroups stream their share of the list with coalesced loads -- full waves
//                        by construction, no queue, so the tile may use all of LDS (2160p: 2 tiles) -- and OR the in-tile
//                        probes into it; partial filters out, k_filter_reduce as before.
// (OR-ing straight into the filter rows with no-return L2 atomics was measured first: 27 G atomics/s, 520 us at 2160p.)
// ------------------------------------------------------------------------------------------
constexpr int IP_THREADS = 256, IP_WAVES = IP_THREADS / WAVE;

template <bool HASHED = false>
__global__ __launch_bounds__(IP_THREADS) void k_insert_positions(
    const uint8_t *__restrict__ masks, uint64_t mask_stride_bytes, uint64_t n,
    const FrameTable tab /* M = bits of -1.0 / m, floor_k = index of the frame's first record */, const uint4 *__restrict__ table /* unused when HASHED */, Seeds seeds,
    uint2 *__restrict__ records, uint32_t *__restrict__ counters /* zeroed; records appended per frame */)
{
    __shared__ __attribute__((aligned(16))) uint32_t queues[IP_WAVES * (IT_WAVE_LDS_BYTES / 4)];
    const uint32_t f = blockIdx.y, s = blockIdx.x, S = gridDim.x;
    const FrameDev fd = tab.f[f];
    if (fd.m == 0) return;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint64_t nbytes = (n + 7) >> 3;
    const uint64_t groups = (nbytes + IT_STEP_BYTES - 1) / IT_STEP_BYTES;
    const uint64_t gper = (groups + S - 1) / S;
    const uint64_t g0 = (uint64_t)s * gper;
    const uint64_t g1 = g0 + gper < groups ? g0 + gper : groups;
    const uint8_t *mask = masks + (uint64_t)f * mask_stride_bytes;
    // set bits of this wave's steps -> the wave's range inside the workgroup's reservation
    __shared__ uint32_t wcount[IP_WAVES];
    __shared__ uint32_t wg_base;
    uint32_t mine = 0;
    for (uint64_t g = g0 + wave; g < g1; g += IP_WAVES) {
        const uint64_t byte = g * IT_STEP_BYTES + lane * 2;
        if (byte < nbytes) {
            uint32_t v = *reinterpret_cast<const uint16_t *>(mask + byte);
            const uint64_t rem = n - byte * 8;
            if (rem < 16) {                                        // pad bits do not count (same bit order as load_bits)
                const uint32_t x = __builtin_bitreverse32(v) >> 16;
                v = (((x & 0xFFu) << 8) | (x >> 8)) & ((1u << rem) - 1u);
            }
            mine += __popc(v);
        }
    }
    const uint32_t wave_total = __builtin_amdgcn_readlane(wave_inclusive_scan(mine), 63);
    if (lane == 0) wcount[wave] = wave_total;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t t = 0;
        for (int k = 0; k < IP_WAVES; ++k) t += wcount[k];
        wg_base = t ? atomicAdd(counters + f, t) : 0u;
    }
    __syncthreads();
    uint32_t rpos = wg_base;
    for (uint32_t k = 0; k < wave; ++k) rpos += wcount[k];
    insert_tab_steps<true, IP_WAVES, HASHED>(mask, mask_stride_bytes, n, fd, table, seeds, nullptr, 0u, 0u, records + fd.floor_k, (uint32_t)__builtin_amdgcn_readfirstlane((int)rpos),
                                             queues + wave * (IT_WAVE_LDS_BYTES / 4), g0, g1, lane, wave);
}

constexpr int IR_UNROLL = 4;                        // records in flight per lane

__global__ __launch_bounds__(IL_THREADS) void k_insert_records(
    const uint2 *__restrict__ records, const uint32_t *__restrict__ counters, const FrameTable tab /* T = index of the frame's first record */,
    uint32_t *__restrict__ partials, uint64_t part_stride_words32, uint32_t tile_words /* even */,
    const SliceTable slices, uint32_t per_tile /* sum of slices.n */, uint32_t Smax)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];   // [tile_words]
    const uint32_t tile = blockIdx.x / per_tile;
    uint32_t s = blockIdx.x - tile * per_tile, f = 0;
    while (s >= slices.n[f]) { s -= slices.n[f]; ++f; }
    const uint32_t S = slices.n[f];
    const uint32_t m = tab.f[f].m;
    if (m == 0) return;
    const uint32_t fk = tab.f[f].floor_k;
    const uint32_t fwords = filter_words(m);
    const uint32_t tile0 = tile * tile_words;
    if (tile0 >= fwords) return;
    const uint32_t tile_bit0 = tile0 << 5, tile_bits = tile_words << 5;
    for (uint32_t i = threadIdx.x; i < tile_words; i += IL_THREADS) lds[i] = 0;
    __syncthreads();
    const uint32_t count = counters[f];
    const uint32_t per = (((count + S - 1) / S) + 63u) & ~63u;
    const uint32_t r0 = s * per < count ? s * per : count, r1 = r0 + per < count ? r0 + per : count;
    const uint2 *rec = records + tab.f[f].T;
    auto insert = [&](uint2 r) {
        uint32_t pos = r.x;
        const uint32_t step = r.y & 0x7FFFFFFFu;
        for (uint32_t j = 0; j < fk; ++j) {
            const uint32_t rel = pos - tile_bit0;
            if (rel < tile_bits) atomicOr(&lds[rel >> 5], msb_bit(pos));
            const uint32_t s2 = pos + step;
            pos = min(s2, s2 - m);
        }
        const uint32_t rel = pos - tile_bit0;
        if ((r.y >> 31) && rel < tile_bits) atomicOr(&lds[rel >> 5], msb_bit(pos));
    };
    uint32_t i = r0 + threadIdx.x;
    for (; i + (IR_UNROLL - 1) * IL_THREADS < r1; i += IR_UNROLL * IL_THREADS) {
        uint2 r[IR_UNROLL];
#pragma unroll
        for (int u = 0; u < IR_UNROLL; ++u) r[u] = rec[i + u * IL_THREADS];
#pragma unroll
        for (int u = 0; u < IR_UNROLL; ++u) insert(r[u]);
    }
    for (; i < r1; i += IL_THREADS) insert(rec[i]);
    __syncthreads();
    uint32_t *part = partials + ((uint64_t)f * Smax + s) * part_stride_words32 + tile0;
    const uint32_t mine = fwords - tile0 < tile_words ? fwords - tile0 : tile_words;
    const uint32_t pairs = (mine + 1) >> 1;                       // tile0 is even: 8-byte aligned
    for (uint32_t k = threadIdx.x; k < pairs; k += IL_THREADS)
        reinterpret_cast<uint2 *>(part)[k] = reinterpret_cast<const uint2 *>(lds)[k];
}

}  // namespace rbf
