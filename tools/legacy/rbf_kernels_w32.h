// EXPERIMENT (round 4), NOT part of the library: a queue-less insert kernel ("every lane walks its own mask dword, dwords dealt out on
// demand").  Bit-identical to k_insert_tab on every mask tried (tools/bench_insert2.hip), but SLOWER: 52 us against 36 at 1080p x 29
// (profiles/r04_insert_w32.txt) -- ~110 instructions per key-iteration where ~50 were planned (claiming costs 36 of them), same-address
// LDS atomics of idle lanes, and a gather that still waits.  Kept as the record of that negative result.
// rbf_kernels_w32.h -- k_insert_w32: round 4's insert kernel for filters of 2^15 <= m < 2^23 bits (reference semantics:
// RationalBloomFilter.add_index over the set positions of the mask, improved_video_compressor.py:99-114, :235-237).  Same job, same
// interface and the same arithmetic as k_insert_tab (rbf_kernels_i64.h): workgroup (tile, frame, slice) ORs the probe positions of
// its slice's set pixels into a partial filter in LDS; the three hashes of a pixel come from the pixel-index hash table (one 32-byte
// gather) or, HASHED, are computed on the spot; the positions are the two FP64 reductions of mod_m_f64.
//
// What is different is how a wave gets from mask bits to full batches of keys.  k_insert_tab compacts the set positions of 1024
// pixels at a time through a per-wave LDS queue (per-lane popcount, a six-step DPP scan, a readlane, scattered queue writes, two LDS
// fences, batches of 64 read back): ~140 instructions of skeleton per 64 keys where the gather, the reductions and the atomics need
// ~45 -- and with ~3 800 instructions per wave the kernel sat on the SIMDs' issue rate, not on the gather.
// Here a wave stages 512 mask dwords (16 384 pixels) in LDS and every LANE WALKS ITS OWN DWORD: pop the lowest set bit (v_ffbl, two
// more), index = dword * 32 + (bit ^ 7) (the mask is MSB-first per byte), gather, reduce, OR.  A lane whose dword is used up CLAIMS
// the wave's next unclaimed dword (one ballot, mbcnt, one scalar add): the dwords are dealt out on demand, so the lanes stay busy
// until the wave's pool is empty whatever the mask looks like (a static split costs max / mean = 1.5 at p = 0.09; clustered masks
// are worse).  No queue, no scan, no fence, no barrier inside the walk.  The gather is software-pipelined one iteration deep: the
// entry requested in iteration t is consumed in iteration t + 1, behind the next pop and claim.
#pragma once
#include "../../new_bloom_filter_repo_amd/csrc/rbf_kernels_i64.h"

namespace rbf {

constexpr uint32_t W32_PIECE_WORDS = 512;                        // mask dwords a wave stages and deals out at a time (16 384 pixels)
constexpr int W32_DEPTH = 4;                                      // table entries in flight per lane
__host__ __device__ constexpr size_t w32_lds_bytes(uint32_t tile_words) { return (size_t)tile_words * 4 + (size_t)IL_WAVES * W32_PIECE_WORDS * 8; }

// IAB (tools/bench_insert2.hip only; 0 in the library): 1 = no table gather (fake entries), 2 = no LDS atomics, 4 = no zeroing / partial store.
// `tab`: M carries the bits of -1.0 / m (as for k_insert_tab).  Grid, slices and partial-filter layout exactly as k_insert_tab's.
// SINGLE: the tile covers the whole filter (tile_words >= every frame's filter words: no in-tile test per probe).
template <int IAB = 0, bool HASHED = false, bool SINGLE = false>
__global__ __launch_bounds__(IL_THREADS) void k_insert_w32(
    const uint8_t *__restrict__ masks, uint64_t mask_stride_bytes, uint64_t n,
    const FrameTable tab, const uint4 *__restrict__ table /* unused when HASHED */, Seeds seeds,
    uint32_t *__restrict__ partials, uint64_t part_stride_words32, uint32_t tile_words /* multiple of 4 */,
    const SliceTable slices, uint32_t per_tile /* sum of slices.n */, uint32_t Smax /* max of slices.n: row pitch of the partials */)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    uint32_t *filt = lds;                                         // [tile_words]
    const uint32_t tile = blockIdx.x / per_tile;
    uint32_t s = blockIdx.x - tile * per_tile, f = 0;
    while (s >= slices.n[f]) { s -= slices.n[f]; ++f; }
    const uint32_t S = slices.n[f];
    const FrameDev fd = tab.f[f];
    if (fd.m == 0) return;
    const uint32_t lane = threadIdx.x & 63u, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t fwords = filter_words(fd.m);
    const uint32_t tile0 = tile * tile_words;
    if (tile0 >= fwords) return;
    if (!(IAB & 4)) {
        const uint4 z = make_uint4(0, 0, 0, 0);
        for (uint32_t i = threadIdx.x; i < tile_words / 4; i += IL_THREADS) reinterpret_cast<uint4 *>(filt)[i] = z;
    }
    // this wave's pool: the NONZERO mask dwords of the piece it is walking and the pixel index of each one's bit 0
    uint32_t *pool_val = lds + tile_words + wave * (2u * W32_PIECE_WORDS), *pool_base = pool_val + W32_PIECE_WORDS;
    __syncthreads();

    // the slice: mask dwords [w0, w1) of the frame's row (rows are padded to 8 bytes with zero bits, so whole dwords can be read)
    const uint32_t nwords = (uint32_t)((n + 31) >> 5);
    const uint32_t wper = (nwords + S - 1) / S;
    const uint32_t w0 = s * wper, w1 = w0 + wper < nwords ? w0 + wper : nwords;
    const uint32_t *mask = reinterpret_cast<const uint32_t *>(masks + (uint64_t)f * mask_stride_bytes);

    const uint32_t m = vgpr_copy(__builtin_amdgcn_readfirstlane(fd.m));
    const uint32_t fk = __builtin_amdgcn_readfirstlane(fd.floor_k);
    const double ninv = __builtin_bit_cast(double, fd.M);
    const uint64_t T = fd.T;
    const uint32_t tile_bit0 = tile0 << 5, tile_bits = tile_words << 5;
    // OR bit `pos` of the filter into the tile when `on` (branch-free: an idle lane ORs in nothing, at a position that exists)
    auto set_bit = [&](uint32_t pos, bool on) {
        uint32_t rel = pos;
        if (!SINGLE) { rel = pos - tile_bit0; on = on && rel < tile_bits; rel = rel < tile_bits ? rel : 0u; }     // unsigned: out-of-tile positions wrap high
        const uint32_t bit = on ? msb_bit(pos) : 0u;
        if (IAB & 2) filt[(rel >> 5) & 1023u] = bit; else atomicOr(&filt[rel >> 5], bit);
    };
    struct Key { uint4 e0, e1; uint32_t idx; bool have; };
    auto request = [&](Key &k, uint32_t idx, bool valid) {
        k.idx = idx; k.have = valid;
        if (HASHED) return;
        const uint32_t slot = hash_table_slot(valid ? idx : 0u);  // idle lanes read entry 0 (always there)
        if (IAB & 1) { k.e0 = make_uint4(idx * 0x9E3779B1u, 0x41D00000u + (idx & 0xFFFFFu), idx * 0x85EBCA77u, 0x41E00000u + (idx & 0xFFFFu)); k.e1 = make_uint4(idx * 3u, idx * 7u, idx * 11u, idx * 13u); }
        else { k.e0 = table[2u * (uint64_t)slot]; k.e1 = table[2u * (uint64_t)slot + 1u]; }
    };
    auto consume = [&](Key &k) {
        if (HASHED) {
            const Hash3 h = hash3_index(k.idx, k.have, seeds);    // wave-uniform call (it votes on the key length)
            const uint64_t d1 = __builtin_bit_cast(uint64_t, (double)h.h1), d2 = __builtin_bit_cast(uint64_t, (double)h.h2);
            k.e0 = make_uint4((uint32_t)d1, (uint32_t)(d1 >> 32), (uint32_t)d2, (uint32_t)(d2 >> 32));
            k.e1 = make_uint4((uint32_t)h.h1, (uint32_t)h.h2, (uint32_t)h.ha, (uint32_t)(h.ha >> 32));
        }
        const double hd1 = __builtin_bit_cast(double, ((uint64_t)k.e0.y << 32) | k.e0.x), hd2 = __builtin_bit_cast(double, ((uint64_t)k.e0.w << 32) | k.e0.z);
        const uint64_t ha = ((uint64_t)k.e1.w << 32) | k.e1.z;
        uint32_t pos = mod_m_f64(hd1, k.e1.x, ninv, m);
        const uint32_t step = mod_m_f64(hd2, k.e1.y, ninv, m);
        for (uint32_t j = 0; j < fk; ++j) {
            set_bit(pos, k.have);
            const uint32_t s2 = pos + step;
            pos = min(s2, s2 - m);
        }
        set_bit(pos, k.have && ha < T);
        k.have = false;
    };

    Key keys[W32_DEPTH];
#pragma unroll
    for (int d = 0; d < W32_DEPTH; ++d) { keys[d].e0 = keys[d].e1 = make_uint4(0, 0, 0, 0); keys[d].idx = 0; keys[d].have = false; }
    uint32_t cur = 0, base = 0;                                   // my current mask dword (bits still to pop) and the pixel index of its bit 0
    for (uint32_t p0 = w0 + wave * W32_PIECE_WORDS; p0 < w1; p0 += IL_WAVES * W32_PIECE_WORDS) {
        const uint32_t pw = w1 - p0 < W32_PIECE_WORDS ? w1 - p0 : W32_PIECE_WORDS;     // dwords of this piece (uniform)
        // stage the piece: 8 coalesced dwords per lane, all in flight together; only the nonzero ones enter the pool (a nearly static
        // frame, or one whose changes are clustered, is mostly zero dwords: nobody claims those)
        uint32_t v[W32_PIECE_WORDS / WAVE];
#pragma unroll
        for (uint32_t i = 0; i < W32_PIECE_WORDS / WAVE; ++i) {
            const uint32_t w = i * WAVE + lane;
            v[i] = mask[p0 + (w < pw ? w : pw - 1u)];
        }
        uint32_t count = 0;                                       // (uniform)
#pragma unroll
        for (uint32_t i = 0; i < W32_PIECE_WORDS / WAVE; ++i) {
            const uint32_t w = i * WAVE + lane;
            const bool nz = w < pw && v[i] != 0u;
            const uint64_t b = __ballot(nz);
            const uint32_t at = count + __builtin_amdgcn_mbcnt_hi((uint32_t)(b >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b, 0u));
            if (nz) { pool_val[at] = v[i]; pool_base[at] = (p0 + w) << 5; }
            count += (uint32_t)__popcll(b);
        }
        wave_lds_fence();
        uint32_t next = 0;                                        // dwords of the pool dealt out so far (uniform)
        bool more = true;
        while (more) {
#pragma unroll
            for (int d = 0; d < W32_DEPTH; ++d) {
                // lanes without bits claim the next dwords of the pool, in lane order
                const uint64_t need = __ballot(cur == 0u);
                if (need && next < count) {
                    const uint32_t mine = next + __builtin_amdgcn_mbcnt_hi((uint32_t)(need >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)need, 0u));
                    if (cur == 0u && mine < count) { cur = pool_val[mine]; base = pool_base[mine]; }
                    next += (uint32_t)__popcll(need);
                }
                const bool has = cur != 0u;
                const uint32_t b = (uint32_t)__builtin_ctz(has ? cur : 1u);
                cur &= cur - 1u;
                const uint32_t idx = base + (b ^ 7u);             // byte k of the dword = pixels 8k .. 8k + 7, MSB first
                consume(keys[d]);                                 // the key requested W32_DEPTH pops ago
                request(keys[d], idx, has && idx < n);
            }
            more = next < count || __any(cur != 0u);              // (keys in flight are consumed by the following pops, or behind the loop)
        }
        wave_lds_fence();                                         // every lane has read its last pool dword before the next piece overwrites it
    }
#pragma unroll
    for (int d = 0; d < W32_DEPTH; ++d) consume(keys[d]);
    __syncthreads();
    uint32_t *part = partials + ((uint64_t)f * Smax + s) * part_stride_words32 + tile0;
    const uint32_t mine = fwords - tile0 < tile_words ? fwords - tile0 : tile_words;
    const uint32_t pairs = (IAB & 4) ? 1u : (mine + 1) >> 1;      // tile0 is even: 8-byte aligned
    for (uint32_t i = threadIdx.x; i < pairs; i += IL_THREADS)
        reinterpret_cast<uint2 *>(part)[i] = reinterpret_cast<const uint2 *>(filt)[i];
}

}  // namespace rbf
