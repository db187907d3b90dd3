// rbf_kernels_lds.h -- the LDS-resident fast path (filters up to ~1.3 Mbit, i.e. 1080p-class frames).
//
// Measured on MI355X (tools/microbench.hip): random dword probes run at ~430 G/s from a 76 KB
// global region and ~260 G/s from an L2-resident 2 MB one, random global atomicOr at ~25 G/s --
// while LDS probes / LDS atomics run at >1600 G/s.  So both the filter build (insert) and the
// filter test (query) keep the whole filter of one frame in LDS:
//
//   k_insert_lds   grid (S, F): workgroup (s, f) builds a PARTIAL filter of frame f in LDS from slice
//                  s of the mask (set positions are compacted through a per-wave LDS queue so the
//                  hashing always runs with full waves) and stores it; k_filter_reduce ORs the S
//                  partials -- no global atomics anywhere.
//   k_query_lds    "frames-inner": a wave owns a segment of P*64 consecutive pixels for the whole
//                  batch.  The three XXH64 of each pixel index depend only on the index, so they are
//                  computed ONCE and kept in registers; then for every frame of the batch the
//                  workgroup stages that frame's filter into LDS and each lane does the per-frame
//                  part only: two Barrett reductions mod m_f, the LDS probes, ballot + compaction.
#pragma once
#include "rbf_kernels.h"

namespace rbf {

// Cache policy for data that is read or written ONCE.  A step of several GOPs moves ~750 MB through 32 MB of L2 and the 256 MB Infinity
// Cache, next to two things that must stay cached: the pixel-index hash table the insert gathers from (54 MB) and the probe images every
// query workgroup restages.  Non-temporal loads / stores keep the one-shot streams from evicting them.  Measured (profiles/r05_cache_policy.txt,
// four pipelines): the mask kernel's frame loads alone +2.6 % at one GOP per call and +1.6 % at four; with the witness-row clears (since
// removed altogether), the reduce kernel's partial loads and filter stores and the compaction's pass-word and mask loads +5.4 % at four GOPs
// per call, +3.8 % at three, nothing at two, -1 % at one (so those follow the batch size: STREAM).  The query kernel's pass-byte stores and the compaction's
// witness stores must NOT stream (-2 ... -3 %: their consumers follow at once), nor the table gathers (insert 33 -> 55 us).
typedef uint32_t nt_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store_stream(uint4 *p, uint4 v) { const nt_u32x4 x = {v.x, v.y, v.z, v.w}; __builtin_nontemporal_store(x, reinterpret_cast<nt_u32x4 *>(p)); }
__device__ __forceinline__ uint4 load_stream(const uint4 *p) { const nt_u32x4 x = __builtin_nontemporal_load(reinterpret_cast<const nt_u32x4 *>(p)); return make_uint4(x.x, x.y, x.z, x.w); }

constexpr int QL_THREADS = 1024;                   // 16 waves, one workgroup per CU, filter double-buffered in LDS
constexpr int QL_WAVES = QL_THREADS / WAVE;
constexpr int QL_P = 8;                            // pixels per lane
constexpr int QL_SEG_PIXELS = QL_P * WAVE;         // 512

constexpr int IL_THREADS = 1024;                   // insert: one workgroup per CU
constexpr int IL_WAVES = IL_THREADS / WAVE;
constexpr int IL_QUEUE = 64 + 512;                 // carry (<64) + one wave-step of 64 mask bytes

__device__ __forceinline__ void wave_lds_fence()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// h mod m for 2 <= m <= 2^30 (m2 = 2m), three 32x32 multiplies for the quotient estimate:
// q' = hh*Mh + hi32(hh*Ml) + hi32(hl*Mh) >= floor(h*M/2^64) - 2 >= floor(h/m) - 3, so
// r' = h - q'*m < 4m <= 2^32 and everything is carried modulo 2^32; two conditional subtracts
// (2m, then m) finish the reduction.
__device__ __forceinline__ uint32_t mod_m_small(uint64_t h, uint32_t m, uint32_t m2, uint32_t Mh, uint32_t Ml)
{
    const uint32_t hh = (uint32_t)(h >> 32), hl = (uint32_t)h;
    const uint32_t q = hh * Mh + __umulhi(hh, Ml) + __umulhi(hl, Mh);
    uint32_t r = hl - q * m;
    r = min(r, r - m2);
    r = min(r, r - m);
    return r;
}

// ------------------------------------------------------------------------------------------
// insert
// ------------------------------------------------------------------------------------------
template <bool SMALL_M>
__global__ __launch_bounds__(IL_THREADS) void k_insert_lds(
    const uint8_t *__restrict__ masks, uint64_t mask_stride_bytes, uint64_t n,
    const FrameTable tab, Seeds seeds,
    uint32_t *__restrict__ partials, uint64_t part_stride_words32, uint32_t tile_words /* even */,
    const SliceTable slices, uint32_t per_tile /* sum of slices.n */, uint32_t Smax /* max of slices.n: row pitch of the partials */)
{
    // 1-D grid of tiles * per_tile workgroups: frame f is cut into slices.n[f] mask slices (0 for a frame that is
    // not Bloom-coded), chosen on the host so that the workgroups fill the 256 CUs whatever the frame count is
    // (29 frames: 24 x 9 + 5 x 8).  The grid is one-dimensional on purpose: consecutive workgroup ids go to
    // consecutive XCDs, and a 2-D grid whose x extent is not a multiple of 8 left some XCDs with more workgroups
    // than CUs (measured: grid (9, 29) -> 114 us instead of 70 us for the same 242 workgroups).
    // The tile index is the slowest coordinate: this workgroup keeps words [tile0, tile0 + tile_words) of the
    // partial filter in LDS and sets only the positions that fall into them.  Filters that fit LDS whole
    // (1080p: 76 KB) have one tile; a 4K filter (306 KB) is built in 3 tiles (keys re-hashed per tile).
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    uint32_t *filt = lds;                                         // [tile_words]
    uint32_t *queues = lds + tile_words;                          // [IL_WAVES][IL_QUEUE]
    const uint32_t tile = blockIdx.x / per_tile;
    uint32_t s = blockIdx.x - tile * per_tile, f = 0;
    while (s >= slices.n[f]) { s -= slices.n[f]; ++f; }          // workgroup-uniform walk over <= 128 bytes
    const uint32_t S = slices.n[f];
    const FrameDev fd = tab.f[f];
    if (fd.m == 0) return;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t fwords = filter_words(fd.m);
    const uint32_t tile0 = tile * tile_words;                     // first word of my tile
    if (tile0 >= fwords) return;
    const uint32_t tile_bit0 = tile0 << 5, tile_bits = tile_words << 5;
    for (uint32_t i = threadIdx.x; i < tile_words; i += IL_THREADS) filt[i] = 0;
    __syncthreads();

    const uint8_t *mask = masks + (uint64_t)f * mask_stride_bytes;
    const uint64_t nbytes = (n + 7) >> 3;
    const uint64_t groups = (nbytes + 63) >> 6;                    // 64-byte wave steps
    const uint64_t gper = (groups + S - 1) / S;
    const uint64_t g0 = (uint64_t)s * gper;
    const uint64_t g1 = g0 + gper < groups ? g0 + gper : groups;
    uint32_t *q = queues + wave * IL_QUEUE;
    uint32_t qn = 0;                                               // wave-uniform queue length

    const uint32_t m = fd.m, m2 = fd.m << 1, Mh = (uint32_t)(fd.M >> 32), Ml = (uint32_t)fd.M;
    auto drain_at = [&](uint32_t first, uint32_t count) {          // hash `count` (<= 64) queued positions
        const bool act = lane < count;                             // (wave-uniform call: hash3_index votes)
        const uint32_t idx = act ? q[first + lane] : 0u;
        const Hash3 h = hash3_index(idx, act, seeds);
        if (act) {
            uint32_t pos, step;
            if (SMALL_M) { pos = mod_m_small(h.h1, m, m2, Mh, Ml); step = mod_m_small(h.h2, m, m2, Mh, Ml); }
            else         { pos = mod_m(h.h1, m, fd.M);             step = mod_m(h.h2, m, fd.M); }
            for (uint32_t j = 0; j < fd.floor_k; ++j) {
                const uint32_t rel = pos - tile_bit0;              // unsigned: out-of-tile positions wrap high
                if (rel < tile_bits) atomicOr(&filt[rel >> 5], msb_bit(pos));
                const uint64_t s2 = (uint64_t)pos + step;
                pos = (uint32_t)(s2 >= m ? s2 - m : s2);
            }
            const uint32_t rel = pos - tile_bit0;
            if (h.ha < fd.T && rel < tile_bits) atomicOr(&filt[rel >> 5], msb_bit(pos));
        }
    };

    auto load_bits = [&](uint64_t g) -> uint32_t {                 // my byte of wave step g in natural bit order
        const uint64_t byte = g * 64 + lane;
        if (g >= g1 || byte >= nbytes) return 0u;
        uint32_t b = __builtin_bitreverse32((uint32_t)mask[byte]) >> 24;
        const uint64_t rem = n - byte * 8;
        if (rem < 8) b &= (1u << rem) - 1u;                        // ignore pad bits
        return b;
    };
    uint32_t nxt = load_bits(g0 + wave);
    for (uint64_t g = g0 + wave; g < g1; g += IL_WAVES) {
        uint32_t bits = nxt;
        nxt = load_bits(g + IL_WAVES);                             // prefetch: the load flies while we hash
        // exclusive prefix of the per-lane counts (0..8) without a cross-lane scan: one ballot per bit
        // of the count, rank of the ballot below my lane (mbcnt), weighted sum -- no LDS round trips
        const uint32_t c = __popc(bits);
        const uint64_t b0 = __ballot((c & 1u) != 0), b1 = __ballot((c & 2u) != 0);
        const uint64_t b2 = __ballot((c & 4u) != 0), b3 = __ballot((c & 8u) != 0);
        const uint32_t excl = rank_below(b0) + 2u * rank_below(b1) + 4u * rank_below(b2) + 8u * rank_below(b3);
        const uint32_t total = __popcll(b0) + 2u * __popcll(b1) + 4u * __popcll(b2) + 8u * __popcll(b3);
        uint32_t off = qn + excl;
        const uint32_t base = (uint32_t)((g * 64 + lane) << 3);
        while (bits) {
            q[off++] = base + __builtin_ctz(bits);
            bits &= bits - 1u;
        }
        qn += total;
        wave_lds_fence();
        while (qn >= WAVE) {                                       // full waves only; order is irrelevant (OR)
            qn -= WAVE;
            drain_at(qn, WAVE);
        }
        wave_lds_fence();                                          // queue reads done before it is refilled
    }
    drain_at(0, qn);
    __syncthreads();
    uint32_t *part = partials + ((uint64_t)f * Smax + s) * part_stride_words32 + tile0;
    const uint32_t mine = fwords - tile0 < tile_words ? fwords - tile0 : tile_words;
    const uint32_t pairs = (mine + 1) >> 1;                       // tile0 is even: 8-byte aligned
    for (uint32_t i = threadIdx.x; i < pairs; i += IL_THREADS)
        reinterpret_cast<uint2 *>(part)[i] = reinterpret_cast<const uint2 *>(filt)[i];
}

// OR the S partial filters of every frame into the final packed filter; count its set bits.
// 16-byte accesses (rows are 8-byte padded and 16-byte aligned bases are not guaranteed, so the
// vector path is taken only when both strides are multiples of 4 words and the bases are aligned).
template <bool STREAM>
__global__ __launch_bounds__(WG_THREADS) void k_filter_reduce(
    const uint32_t *partials, uint64_t part_stride_words32, uint32_t Smax /* row pitch of the partials, in slices */,
    const SliceTable slices /* partial filters per frame */,
    const FrameTable tab,
    uint32_t *filters /* may alias partials when Smax == 1 */, uint64_t filter_stride_words32,
    uint64_t *__restrict__ stats, uint32_t vec_ok,
    uint32_t *__restrict__ image /* nullable: probe image rows (~bswap of every dword) for the FP64 query kernel */, uint64_t image_stride_words32)
{
    __shared__ uint32_t red[WG_WAVES];
    const uint32_t f = blockIdx.y;
    const uint32_t S = slices.n[f];
    const uint32_t m = tab.f[f].m;
    const uint32_t fwords = m ? filter_words(m) : 0u;
    uint32_t *filt = filters + (uint64_t)f * filter_stride_words32;
    const uint32_t *part = partials + (uint64_t)f * Smax * part_stride_words32;
    uint32_t pc = 0;
    if (vec_ok) {
        const uint64_t quads = filter_stride_words32 >> 2;
        for (uint64_t q = (uint64_t)blockIdx.x * WG_THREADS + threadIdx.x; q < quads; q += (uint64_t)gridDim.x * WG_THREADS) {
            const uint64_t w = q << 2;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (w < fwords) {
                // eight partials in flight (clamped slice index: a repeated partial ORs in nothing new); one load per loop
                // iteration waits for each L2 round trip in turn
                for (uint32_t s0 = 0; s0 < S; s0 += 8) {
                    uint4 x[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const uint32_t sj = s0 + j < S ? s0 + j : S - 1;
                        const uint4 *src = reinterpret_cast<const uint4 *>(part + (uint64_t)sj * part_stride_words32 + w);
                        x[j] = STREAM ? load_stream(src) : *src;
                    }
#pragma unroll
                    for (int j = 0; j < 8; ++j) { v.x |= x[j].x; v.y |= x[j].y; v.z |= x[j].z; v.w |= x[j].w; }
                }
                if (w + 1 >= fwords) v.y = 0;                    // words past the filter end hold LDS padding
                if (w + 2 >= fwords) v.z = 0;
                if (w + 3 >= fwords) v.w = 0;
            }
            if (m) {                                             // passthrough frames: filter untouched
                if (STREAM) store_stream(reinterpret_cast<uint4 *>(filt + w), v); else *reinterpret_cast<uint4 *>(filt + w) = v;
            }
            if (image && w < image_stride_words32)
                *reinterpret_cast<uint4 *>(image + (uint64_t)f * image_stride_words32 + w) =
                    make_uint4(~__builtin_bswap32(v.x), ~__builtin_bswap32(v.y), ~__builtin_bswap32(v.z), ~__builtin_bswap32(v.w));
            pc += __popc(v.x) + __popc(v.y) + __popc(v.z) + __popc(v.w);
        }
    } else {
        for (uint64_t w = (uint64_t)blockIdx.x * WG_THREADS + threadIdx.x; w < filter_stride_words32; w += (uint64_t)gridDim.x * WG_THREADS) {
            uint32_t v = 0;
            if (w < fwords)
                for (uint32_t s = 0; s < S; ++s) v |= part[(uint64_t)s * part_stride_words32 + w];
            if (m) filt[w] = v;
            if (image && w < image_stride_words32) image[(uint64_t)f * image_stride_words32 + w] = ~__builtin_bswap32(v);
            pc += __popc(v);
        }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) pc += __shfl_down(pc, d);
    if ((threadIdx.x & 63u) == 0) red[threadIdx.x >> 6] = pc;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t t = 0;
        for (int k = 0; k < WG_WAVES; ++k) t += red[k];
        if (t) atomicAdd((unsigned long long *)&stats[(uint64_t)f * 4 + 1], (unsigned long long)t);
    }
}

// Probe image of caller-supplied packed filters (decode): image[f][w] = ~bswap(filters[f][w]).
__global__ __launch_bounds__(WG_THREADS) void k_probe_image(const uint32_t *__restrict__ filters, uint64_t filter_stride_words32,
                                                            uint32_t *__restrict__ image, uint64_t image_stride_words32)
{
    const uint32_t f = blockIdx.y;
    for (uint64_t w = (uint64_t)blockIdx.x * WG_THREADS + threadIdx.x; w < image_stride_words32; w += (uint64_t)gridDim.x * WG_THREADS)
        image[(uint64_t)f * image_stride_words32 + w] = w < filter_stride_words32 ? ~__builtin_bswap32(filters[(uint64_t)f * filter_stride_words32 + w]) : ~0u;
}

// ------------------------------------------------------------------------------------------
// query, frames-inner
// ------------------------------------------------------------------------------------------
// Filter staging by LDS-DMA (global_load_lds_dwordx4: 1 KiB per wave-instruction, no VGPRs, no
// ds_write pass).  LDS destination = M0 (wave-uniform base) + lane*16; the global source is per lane.
//
// The DMA is issued from inline asm ON PURPOSE: when hipcc sees the builtin it cannot tell the DMA's
// LDS destination (the *other* buffer) from the probes' source, so it drains vmcnt(0) in front of
// every LDS access of the compute phase and the double buffering buys nothing (measured: 35 us of
// a 183 us launch).  Asm DMAs are invisible to its scoreboard; completion is waited for explicitly
// with dma_wait_all() right before the workgroup barrier that hands the buffer over.
__device__ __forceinline__ void dma16(const uint32_t *gsrc, uint32_t lds_byte_addr)
{
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\t"
                 "s_mov_b32 m0, %2\n\t"
                 "s_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, off\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_byte_addr) : "memory");
}
__device__ __forceinline__ void dma4(const uint32_t *gsrc, uint32_t lds_byte_addr)
{
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\t"
                 "s_mov_b32 m0, %2\n\t"
                 "s_nop 0\n\t"
                 "global_load_lds_dword %1, off\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_byte_addr) : "memory");
}
__device__ __forceinline__ void dma_wait_all() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

__device__ __forceinline__ uint32_t lds_addr_of(const uint32_t *p)
{
    return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) uint32_t *)p;
}

__device__ __forceinline__ void dma_filter(uint32_t *lds_dst, const uint32_t *src, uint32_t words, uint32_t wave,
                                           uint32_t lane, uint32_t nwaves)
{
    const uint32_t base = __builtin_amdgcn_readfirstlane(lds_addr_of(lds_dst));
    const uint32_t npieces = words >> 2;                      // whole 16-byte pieces
    const uint32_t nchunks = (npieces + 63u) >> 6;
    for (uint32_t c = wave; c < nchunks; c += nwaves) {
        const uint32_t piece = (c << 6) + lane;
        if (piece < npieces) dma16(src + (piece << 2), __builtin_amdgcn_readfirstlane(base + (c << 10)));
    }
    const uint32_t tail = words & 3u;                         // 0..3 dwords left: 4-byte DMA
    if (wave == 0 && lane < tail) dma4(src + (npieces << 2) + lane, __builtin_amdgcn_readfirstlane(base + (npieces << 4)));
}

// v_writelane_b32: put a wave-uniform value into ONE lane of a VGPR (no builtin in this hipcc).
// The value must not be an SGPR a VALU instruction wrote in the last 5 wait states (the ballot!):
// hipcc pads no hazards inside asm, and without the wait v_writelane reads the PREVIOUS value --
// caught by the parity tests.  Callers batch their writelanes behind one valu_sgpr_hazard_gap().
// The gap takes the SGPR values as in/out operands: "memory" alone would not stop hipcc from sinking
// a register-only ballot below the s_nop (guide rule 18), so the values are threaded THROUGH the asm.
__device__ __forceinline__ void valu_sgpr_hazard_gap(uint32_t (&a)[8], uint32_t (&b)[8])
{
    asm volatile("" : "+s"(a[0]), "+s"(a[1]), "+s"(a[2]), "+s"(a[3]), "+s"(a[4]), "+s"(a[5]), "+s"(a[6]), "+s"(a[7]));
    asm volatile("s_nop 4" : "+s"(b[0]), "+s"(b[1]), "+s"(b[2]), "+s"(b[3]), "+s"(b[4]), "+s"(b[5]), "+s"(b[6]), "+s"(b[7]));
}
__device__ __forceinline__ void write_lane(uint32_t &dst, uint32_t uniform_value, int lane_index)
{
    asm volatile("v_writelane_b32 %0, %1, %2" : "+v"(dst) : "s"(uniform_value), "n"(lane_index));
}

// LDS dword holding filter bit `pos`: (pos >> 5) * 4 + base in two instructions (v_bfe_u32 +
// v_lshl_add_u32); written plainly the compiler folds it to shift / and / add.
__device__ __forceinline__ uint32_t probe_word(const uint32_t *filt, uint32_t pos)
{
    const uint32_t w = __builtin_amdgcn_ubfe(pos, 5, 27);
    return filt[w];
}

// One frame's pass over a lane's QL_P consecutive pixels: reductions mod m, LDS probes, verdict.
// FK >= 0: floor(k*) known at compile time (fully unrolled probes); FK < 0: runtime fk.
// The verdict of pixel j is the sign bit of `acc`; it is shifted into `pb` (one v_alignbit), so after
// the loop pb holds the lane's QL_P verdicts MSB-first -- exactly one byte of the packed pass vector.
// Returns the wave's number of passing positions.  Branch-free for FK >= 0 so the QL_P dependency
// chains interleave.
template <bool SMALL_M, int FK>
__device__ __forceinline__ uint32_t frame_pass(
    const uint64_t (&h1)[QL_P], const uint64_t (&h2)[QL_P], const uint64_t (&ha)[QL_P], uint32_t validmask,
    const uint32_t *filt, uint32_t m, uint64_t M, uint64_t T, uint32_t fk_rt, uint32_t &pb)
{
    const uint32_t Mh = (uint32_t)(M >> 32), Ml = (uint32_t)M;
    const uint32_t fk = FK >= 0 ? (uint32_t)FK : fk_rt;
    const uint32_t m2 = m << 1;
    uint32_t npass = 0;
    pb = 0;
#pragma unroll
    for (int it = 0; it < QL_P; ++it) {
        uint32_t pos, step;
        if (SMALL_M) { pos = mod_m_small(h1[it], m, m2, Mh, Ml); step = mod_m_small(h2[it], m, m2, Mh, Ml); }
        else         { pos = mod_m(h1[it], m, M);            step = mod_m(h2[it], m, M); }
        // Each probe shifts its word LEFT so that the probed bit lands in bit 31: the verdict is the
        // sign bit of the AND of all probes.  MSB-first bit (pos & 31) ^ 7 -> shift (pos ^ 24) & 31.
        uint32_t acc = validmask << (31 - it);                // only the sign bit is ever looked at: bit `it` -> bit 31
#pragma unroll
        for (uint32_t j = 0; j < fk; ++j) {
            acc &= probe_word(filt, pos) << ((pos ^ 24u) & 31u);
            if (SMALL_M) { const uint32_t s2 = pos + step; pos = min(s2, s2 - m); }
            else { const uint64_t s2 = (uint64_t)pos + step; pos = (uint32_t)(s2 >= m ? s2 - m : s2); }
        }
        const uint32_t x = probe_word(filt, pos) << ((pos ^ 24u) & 31u);
        acc &= (ha[it] < T) ? x : 0x80000000u;
        pb = __builtin_amdgcn_alignbit(pb, acc, 31);          // (pb << 1) | (acc >> 31)
        npass += __popcll(__ballot((int32_t)acc < 0));
    }
    return npass;
}

// Query, frames-inner (A5 for encode, A6 for decode: both need the pass word of every 64 pixels).
//   pass_words, as bytes: [(f*nseg + seg)*QL_SEG_PIXELS/8 + b]  bit 7-r = position seg*QL_SEG_PIXELS + 8b + r passes
//                                          filter f, i.e. the packed (numpy.packbits) order of masks and witnesses
// A lane owns QL_P = 8 CONSECUTIVE pixels (its verdicts are one byte of that vector, and their keys share all
// but the last character, see hash3_run8); a wave owns 512 consecutive pixels.
//   seg_cnt[f*nseg + seg]                  passing positions of the segment (= witness bits it owns)
// SMALL_M: every filter of the batch has 2 <= m <= 2^30 (host-checked) -> cheap reductions.
template <bool DOUBLE_BUFFER, bool SMALL_M>
__global__ __launch_bounds__(QL_THREADS) void k_query_lds(
    uint64_t n, uint32_t nframes, const FrameTable tab, Seeds seeds,
    const uint32_t *__restrict__ filters, uint64_t filter_stride_words32, uint32_t fwords_max,
    uint32_t *__restrict__ seg_cnt, uint64_t nseg, uint64_t *__restrict__ pass_words)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    const uint32_t bufwords = (fwords_max + 3u) & ~3u;            // 16-byte multiple
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t nwaves = blockDim.x >> 6;
    const uint64_t seg = (uint64_t)blockIdx.x * nwaves + wave;
    const bool live = seg < nseg;
    const uint64_t base = seg * QL_SEG_PIXELS;

    // ---- frame-independent part: the three hashes of my P consecutive pixel indices ------
    static_assert(QL_P == 8, "a lane's verdicts fill one byte; hash3_run8 hashes runs of 8");
    uint64_t h1[QL_P], h2[QL_P], ha[QL_P];
    uint32_t validmask = 0;
    const uint64_t i0 = base + (uint64_t)lane * QL_P;
#pragma unroll
    for (int it = 0; it < QL_P; ++it) {
        h1[it] = 0; h2[it] = 0; ha[it] = ~0ull;
        if (live && i0 + it < n) validmask |= 1u << it;
    }
    if (!hash3_run8((uint32_t)i0, validmask, seeds, h1, h2, ha)) {
#pragma unroll
        for (int it = 0; it < QL_P; ++it) {                      // mixed key lengths in this wave: index by index
            const bool act = (validmask >> it) & 1u;
            const Hash3 h = hash3_index((uint32_t)(i0 + it), act, seeds);
            h1[it] = h.h1; h2[it] = h.h2; ha[it] = h.ha;
        }
    }
    uint8_t *pass_bytes = reinterpret_cast<uint8_t *>(pass_words);

    // passthrough frames (m == 0): nothing passes
    for (uint32_t g = 0; g < nframes; ++g) {
        if (tab.f[g].m == 0) {
            if (live && lane == 0) seg_cnt[(uint64_t)g * nseg + seg] = 0;
            if (live) pass_bytes[((uint64_t)g * nseg + seg) * (QL_SEG_PIXELS / 8) + lane] = 0;
        }
    }
    // Every workgroup walks the frames in the same order: all CUs then pull the same 76 KB filter at
    // about the same time, which the L2 serves best (measured in round 1: rotating the start frame per workgroup, so that
    // ~29 different filters are in flight, costs +11 us per launch).
    auto frame_at = [&](uint32_t k) -> uint32_t { return k; };
    auto next_active = [&](uint32_t k) -> uint32_t { while (k < nframes && tab.f[frame_at(k)].m == 0) ++k; return k; };
    uint32_t k = next_active(0);
    uint32_t cur = 0;
    if (DOUBLE_BUFFER && k < nframes) {
        const uint32_t f0 = frame_at(k);
        dma_filter(lds, filters + (uint64_t)f0 * filter_stride_words32, filter_words(tab.f[f0].m), wave, lane, nwaves);
    }
    while (k < nframes) {
        k = __builtin_amdgcn_readfirstlane(k);                    // frame indices are wave-uniform: scalar table loads
        const uint32_t kn = __builtin_amdgcn_readfirstlane(next_active(k + 1));
        const uint32_t f = __builtin_amdgcn_readfirstlane(frame_at(k));
        const uint32_t fn = __builtin_amdgcn_readfirstlane(kn < nframes ? frame_at(kn) : 0u);
        const FrameDev fd = tab.f[f];
        const uint32_t *filt;
        if (DOUBLE_BUFFER) {
            dma_wait_all();           // my share of DMA(f) has landed ...
            __syncthreads();          // ... and everyone's; buffer cur^1 is free again
            filt = lds + cur * bufwords;
            if (kn < nframes)
                dma_filter(lds + (cur ^ 1u) * bufwords, filters + (uint64_t)fn * filter_stride_words32, filter_words(tab.f[fn].m), wave, lane, nwaves);
            cur ^= 1u;
        } else {
            __syncthreads();          // previous frame's probes are done
            dma_filter(lds, filters + (uint64_t)f * filter_stride_words32, filter_words(fd.m), wave, lane, nwaves);
            dma_wait_all();
            __syncthreads();
            filt = lds;
        }
        // frame geometry is wave-uniform: keep it in SGPRs so every branch below is scalar
        // (the builtin returns int: go through uint32_t or the low half sign-extends)
        const uint32_t m = __builtin_amdgcn_readfirstlane(fd.m);
        const uint32_t fk = __builtin_amdgcn_readfirstlane(fd.floor_k);
        const uint32_t Mh = __builtin_amdgcn_readfirstlane((uint32_t)(fd.M >> 32));
        const uint32_t Ml = __builtin_amdgcn_readfirstlane((uint32_t)fd.M);
        const uint32_t Thi = __builtin_amdgcn_readfirstlane((uint32_t)(fd.T >> 32));
        const uint32_t Tlo = __builtin_amdgcn_readfirstlane((uint32_t)fd.T);
        const uint64_t T = ((uint64_t)Thi << 32) | Tlo;
        const uint64_t M = ((uint64_t)Mh << 32) | Ml;

        uint32_t pb = 0, npass;
        // floor(k*) is a small integer: straight-line code for the common values lets the compiler
        // issue every LDS probe of all QL_P pixels back to back instead of one round trip at a time.
        switch (fk) {
        case 1: npass = frame_pass<SMALL_M, 1>(h1, h2, ha, validmask, filt, m, M, T, fk, pb); break;
        case 2: npass = frame_pass<SMALL_M, 2>(h1, h2, ha, validmask, filt, m, M, T, fk, pb); break;
        case 3: npass = frame_pass<SMALL_M, 3>(h1, h2, ha, validmask, filt, m, M, T, fk, pb); break;
        case 4: npass = frame_pass<SMALL_M, 4>(h1, h2, ha, validmask, filt, m, M, T, fk, pb); break;
        default: npass = frame_pass<SMALL_M, -1>(h1, h2, ha, validmask, filt, m, M, T, fk, pb); break;
        }
        if (live) {
            pass_bytes[((uint64_t)f * nseg + seg) * (QL_SEG_PIXELS / 8) + lane] = (uint8_t)pb;
            if (lane == 0) seg_cnt[(uint64_t)f * nseg + seg] = npass;
        }
        k = kn;
    }
}

// ------------------------------------------------------------------------------------------
// query for filters larger than LDS (4K frames: 306 KB): the filter is staged tile by tile; every
// (pixel, frame) computes its probe positions once and tests, per tile, the probes that land in it.
// Same outputs as k_query_lds with TQ_P pass words per segment.
// ------------------------------------------------------------------------------------------
constexpr int TQ_P = 8;                            // pixels per lane (more pixels per filter staging)
constexpr int TQ_SEG_PIXELS = TQ_P * WAVE;         // 256

template <bool SMALL_M>
__global__ __launch_bounds__(QL_THREADS) void k_query_tiled(
    uint64_t n, uint32_t nframes, const FrameTable tab, Seeds seeds,
    const uint32_t *__restrict__ filters, uint64_t filter_stride_words32, uint32_t tile_words /* multiple of 4 */,
    uint32_t *__restrict__ seg_cnt, uint64_t nseg, uint64_t *__restrict__ pass_words)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t nwaves = blockDim.x >> 6;
    const uint64_t seg = (uint64_t)blockIdx.x * nwaves + wave;
    const bool live = seg < nseg;
    const uint64_t base = seg * TQ_SEG_PIXELS;
    uint64_t h1[TQ_P], h2[TQ_P], ha[TQ_P];
    uint32_t validmask = 0;
#pragma unroll
    for (int it = 0; it < TQ_P; ++it) {
        const uint64_t i = base + (uint64_t)it * WAVE + lane;
        const bool act = live && i < n;
        const Hash3 h = hash3_index((uint32_t)i, act, seeds);
        h1[it] = act ? h.h1 : 0; h2[it] = act ? h.h2 : 0; ha[it] = act ? h.ha : ~0ull;
        validmask |= act ? 1u << it : 0u;
    }
    for (uint32_t f = 0; f < nframes; ++f) {
        const FrameDev fd = tab.f[f];
        if (fd.m == 0) {                                          // passthrough frame (block-uniform)
            if (live && lane == 0) seg_cnt[(uint64_t)f * nseg + seg] = 0;
            if (live && lane < TQ_P) pass_words[((uint64_t)f * nseg + seg) * TQ_P + lane] = 0;
            continue;
        }
        const uint32_t m = __builtin_amdgcn_readfirstlane(fd.m);
        const uint32_t fk = __builtin_amdgcn_readfirstlane(fd.floor_k);
        const uint32_t Mh = __builtin_amdgcn_readfirstlane((uint32_t)(fd.M >> 32));
        const uint32_t Ml = __builtin_amdgcn_readfirstlane((uint32_t)fd.M);
        const uint32_t Thi = __builtin_amdgcn_readfirstlane((uint32_t)(fd.T >> 32));
        const uint32_t Tlo = __builtin_amdgcn_readfirstlane((uint32_t)fd.T);
        const uint64_t T = ((uint64_t)Thi << 32) | Tlo, M = ((uint64_t)Mh << 32) | Ml;
        const uint32_t fwords = filter_words(m);
        uint32_t pos0[TQ_P], step[TQ_P], acc[TQ_P];
#pragma unroll
        for (int it = 0; it < TQ_P; ++it) {
            if (SMALL_M) { pos0[it] = mod_m_small(h1[it], m, m << 1, Mh, Ml); step[it] = mod_m_small(h2[it], m, m << 1, Mh, Ml); }
            else         { pos0[it] = mod_m(h1[it], m, M);                     step[it] = mod_m(h2[it], m, M); }
            acc[it] = (validmask >> it) << 31;
        }
        for (uint32_t tile0 = 0; tile0 < fwords; tile0 += tile_words) {
            const uint32_t words = fwords - tile0 < tile_words ? fwords - tile0 : tile_words;
            __syncthreads();                                      // previous tile's probes are done
            dma_filter(lds, filters + (uint64_t)f * filter_stride_words32 + tile0, words, wave, lane, nwaves);
            dma_wait_all();
            __syncthreads();
            const uint32_t bit0 = tile0 << 5, nbits = words << 5;
#pragma unroll
            for (int it = 0; it < TQ_P; ++it) {
                uint32_t pos = pos0[it];
                for (uint32_t j = 0; j <= fk; ++j) {               // j == fk: the activated extra probe
                    const uint32_t rel = pos - bit0;
                    const bool in = rel < nbits && (j < fk || ha[it] < T);
                    const uint32_t w = lds[in ? rel >> 5 : 0u];
                    acc[it] &= in ? w << ((pos ^ 24u) & 31u) : 0x80000000u;
                    const uint64_t s2 = (uint64_t)pos + step[it];
                    pos = (uint32_t)(s2 >= m ? s2 - m : s2);
                }
            }
        }
        uint32_t npass = 0, pw_lo = 0, pw_hi = 0;
#pragma unroll
        for (int it = 0; it < TQ_P; ++it) {
            const uint64_t pw = __ballot((int32_t)acc[it] < 0);
            if (lane == (uint32_t)it) { pw_lo = (uint32_t)pw; pw_hi = (uint32_t)(pw >> 32); }
            npass += __popcll(pw);
        }
        if (live) {
            if (lane < TQ_P) pass_words[((uint64_t)f * nseg + seg) * TQ_P + lane] = flip_bytes64(((uint64_t)pw_hi << 32) | pw_lo);
            if (lane == 0) seg_cnt[(uint64_t)f * nseg + seg] = npass;
        }
    }
}

// ------------------------------------------------------------------------------------------
// witness compaction (A5): witness = mask bits at the passing positions, in position order.
// One lane per 64-pixel word: its bits are pext(mask word, pass word) placed at
// seg_off[segment] + (passes of the segment's earlier words) in the pre-zeroed packed witness.
// ------------------------------------------------------------------------------------------
// Where the witness bits of every chunk of WG_THREADS words (= chunk_segs whole segments: the compaction's and the expansion's
// workgroups) start: off[f][c] = passes of all earlier segments of frame f.  One workgroup per frame scans the frame's segment counts once
// -- until round 5 every compaction workgroup summed all the counts in front of it by itself (a third of its instructions, and 1 MB of L2
// reads per 1080p frame).  total < 2^32 (n < 2^32).
constexpr int CO_THREADS = 1024;
__global__ __launch_bounds__(CO_THREADS) void k_chunk_offsets(const uint32_t *__restrict__ seg_cnt, uint64_t nseg, uint32_t chunk_segs, uint32_t nchunks,
                                                              uint32_t *__restrict__ off, uint32_t *__restrict__ witnesses /* nullable (decode) */, uint64_t witness_stride_words32)
{
    // Encode: this kernel also ZEROES the few witness dwords the compaction's workgroups share -- the 64-bit word around every chunk's
    // first bit and around the witness's end (whose pad bits must read 0) -- so that nobody has to clear whole witness rows (until round 5
    // the mask kernel cleared 259 KB per 1080p frame for them: a tenth of its HBM traffic).  Everything else the compaction overwrites.
    // a thread sums `per` consecutive counts (4 when a chunk has that many segments), the workgroup scans the sums, and the thread that
    // holds a chunk's first segments writes the chunk's offset; frames of more than 1024 * per segments take several rounds with a carry
    __shared__ uint32_t wtot[CO_THREADS / WAVE];
    __shared__ uint32_t carry_s;
    const uint32_t f = blockIdx.x;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t *cnt = seg_cnt + (uint64_t)f * nseg;
    uint32_t *out = off + (uint64_t)f * nchunks;
    uint32_t *wit = witnesses ? witnesses + (uint64_t)f * witness_stride_words32 : nullptr;
    auto zero_word_at = [&](uint32_t bit) {
        const uint64_t d = (bit >> 5) & ~1u;
        if (d < witness_stride_words32) wit[d] = 0;
        if (d + 1 < witness_stride_words32) wit[d + 1] = 0;
    };
    const uint32_t per = (chunk_segs & 3u) == 0 ? 4u : (chunk_segs & 1u) == 0 ? 2u : 1u;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (uint64_t base = 0; base < nseg; base += (uint64_t)CO_THREADS * per) {
        const uint64_t i0 = base + (uint64_t)threadIdx.x * per;
        uint32_t sum = 0;
        for (uint32_t k = 0; k < per; ++k) sum += i0 + k < nseg ? cnt[i0 + k] : 0u;
        const uint32_t incl = wave_inclusive_scan(sum);
        if (lane == WAVE - 1) wtot[wave] = incl;
        const uint32_t carry = carry_s;
        __syncthreads();
        uint32_t before = carry;
        for (uint32_t k = 0; k < wave; ++k) before += wtot[k];
        if (i0 < nseg && i0 % chunk_segs == 0) {
            out[i0 / chunk_segs] = before + incl - sum;
            if (wit) zero_word_at(before + incl - sum);
        }
        __syncthreads();
        if (threadIdx.x == CO_THREADS - 1) carry_s = before + incl;
        __syncthreads();
    }
    if (wit && threadIdx.x == 0) zero_word_at(carry_s);          // the end of the witness
}

// Software pext / pdep through a 256-byte LDS table of their 4-bit forms, entry [p4 << 4 | x4] (thread t of a 256-thread workgroup
// writes entry t): pext4 = the bits of x4 at the set positions of p4, packed low; pdep4 = the low popc(p4) bits of x4 dealt out to the
// set positions of p4.  A 64-bit word is sixteen independent look-ups -- a lane reads one byte, lanes reading the same dword are served
// by one broadcast and the table covers each of the 64 banks once, so there are no bank conflicts -- against a loop that ran as long as
// the busiest lane of the wave (~9 rounds of 19 instructions for the compaction, ~22 of 14 for the expansion).
static_assert(WG_THREADS == 256, "one table entry per thread");
__host__ __device__ constexpr uint32_t pext4_entry(uint32_t t)
{
    const uint32_t p4 = t >> 4, x4 = t & 15u;
    uint32_t r = 0, k = 0;
#pragma unroll
    for (uint32_t b = 0; b < 4; ++b)
        if ((p4 >> b) & 1u) { r |= ((x4 >> b) & 1u) << k; ++k; }
    return r;
}
__host__ __device__ constexpr uint32_t pdep4_entry(uint32_t t)
{
    const uint32_t p4 = t >> 4, x4 = t & 15u;
    uint32_t r = 0, k = 0;
#pragma unroll
    for (uint32_t b = 0; b < 4; ++b)
        if ((p4 >> b) & 1u) { r |= ((x4 >> k) & 1u) << b; ++k; }
    return r;
}
// The two tables, built at compile time (until round 5 every workgroup computed its own: ~25 instructions per thread), 64 dwords each in
// constant memory; wave 0 of a workgroup copies one into LDS.
struct Lut256 { uint32_t w[64]; };
template <bool PDEP>
constexpr Lut256 make_lut4()
{
    Lut256 t{};
    for (uint32_t i = 0; i < 256; ++i) t.w[i >> 2] |= (PDEP ? pdep4_entry(i) : pext4_entry(i)) << (8u * (i & 3u));
    return t;
}
static __constant__ Lut256 LUT_PEXT4 = make_lut4<false>();
static __constant__ Lut256 LUT_PDEP4 = make_lut4<true>();
__device__ __forceinline__ void lut_to_lds(uint8_t *lut, const Lut256 &src)
{
    if (threadIdx.x < 64u) reinterpret_cast<uint32_t *>(lut)[threadIdx.x] = src.w[threadIdx.x];
}

// pext(x, p) of a 32-bit half: <= popc(p) <= 32 bits, LSB = the first set position of p
__device__ __forceinline__ uint32_t pext32_lut(const uint8_t *lut, uint32_t x, uint32_t p)
{
    uint32_t out = 0, off = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const uint32_t p4 = __builtin_amdgcn_ubfe(p, 4u * j, 4u), x4 = __builtin_amdgcn_ubfe(x, 4u * j, 4u);
        out |= (uint32_t)lut[(p4 << 4) | x4] << (off & 31u);       // (off + popc(p4) <= 32, the entry has popc(p4) bits; off = 32 only with nothing left)
        off += __popc(p4);
    }
    return out;
}
// pdep(w, p) of a 32-bit half: the low popc(p) bits of w (stream order, LSB first) dealt out to the set positions of p
__device__ __forceinline__ uint32_t pdep32_lut(const uint8_t *lut, uint32_t w, uint32_t p)
{
    uint32_t out = 0, off = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const uint32_t p4 = __builtin_amdgcn_ubfe(p, 4u * j, 4u), w4 = __builtin_amdgcn_ubfe(w, off, 4u);     // (off = 32 only when nothing is left to deal out)
        out |= (uint32_t)lut[(p4 << 4) | w4] << (4u * j);
        off += __popc(p4);
    }
    return out;
}

template <bool STREAM>
__global__ __launch_bounds__(WG_THREADS) void k_compact_witness(
    const uint64_t *__restrict__ pass_words, const uint32_t *__restrict__ seg_cnt, uint64_t nseg, uint32_t words_per_seg,
    const uint64_t *__restrict__ masks, uint64_t mask_stride_words64, uint64_t n,
    uint32_t *__restrict__ witnesses, uint64_t witness_stride_words32, uint64_t *__restrict__ stats,
    const uint32_t *__restrict__ chunk_off /* k_chunk_offsets: [frame][workgroup] */)
{
    // A workgroup owns WG_THREADS consecutive words (= whole segments).  Their witness bits form one contiguous bit range starting at
    // (passes of all earlier segments, from k_chunk_offsets): the range is assembled in LDS with LDS atomics and written with plain
    // coalesced stores; only its first and last dword may be shared with the neighbouring workgroups (atomicOr onto zeroed dwords).  The offsets inside the
    // chunk are a block scan.
    //
    // The step is bound by instruction issue, so this kernel is written for a short instruction stream (round 3: ~415
    // VALU wave-instructions per 64 words, now ~230): the earlier counts are read four to a load, all loads of a thread are in flight
    // before the first wait, both block-wide sums cross ONE barrier, the wave scans are DPP adds, and the pext is sixteen look-ups.
    __shared__ uint32_t buf[WG_THREADS * 2 + 2];
    __shared__ uint32_t wsum[WG_WAVES];
    __shared__ __attribute__((aligned(4))) uint8_t lut[256];
    const uint32_t f = blockIdx.y;
    const uint32_t lane = threadIdx.x & 63u, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t nwords = (uint32_t)((n + 63) >> 6);                        // n < 2^32 (rbf_plan_batch)
    const uint32_t total = (uint32_t)nseg * words_per_seg;
    const uint64_t *pwf = pass_words + (uint64_t)f * total;
    const uint32_t start32 = chunk_off[(uint64_t)f * gridDim.x + blockIdx.x];      // (uniform: a scalar load)
    uint32_t *wit = witnesses + (uint64_t)f * witness_stride_words32;
    const uint32_t wbeg = blockIdx.x * WG_THREADS;
    const uint32_t w = wbeg + threadIdx.x;
    // my word (packed -> bit b = position 64w + b) and its mask word: requested before anything waits
    const bool have = w < total && w < nwords;
    const uint64_t *mkp = masks + (uint64_t)f * mask_stride_words64 + w;
    const uint64_t pw_raw = have ? (STREAM ? __builtin_nontemporal_load(pwf + w) : pwf[w]) : 0ull;        // read once: see the cache-policy note at the top
    const uint64_t mk_raw = have ? (STREAM ? __builtin_nontemporal_load(mkp) : *mkp) : 0ull;
    lut_to_lds(lut, LUT_PEXT4);
    buf[threadIdx.x] = 0;
    buf[threadIdx.x + WG_THREADS] = 0;
    if (threadIdx.x < 2) buf[threadIdx.x + 2 * WG_THREADS] = 0;
    const uint64_t pw = flip_bytes64(pw_raw);
    const uint32_t pw_lo = (uint32_t)pw, pw_hi = (uint32_t)(pw >> 32);
    const uint32_t c_lo = __popc(pw_lo), c = c_lo + __popc(pw_hi);
    const uint32_t incl = wave_inclusive_scan(c);
    if (lane == WAVE - 1) wsum[wave] = incl;
    __syncthreads();
    uint32_t before = 0, chunk_total = 0;
#pragma unroll
    for (int k = 0; k < WG_WAVES; ++k) {
        if ((uint32_t)k < wave) before += wsum[k];
        chunk_total += wsum[k];
    }
    const uint32_t obase = start32 & ~31u;                                    // dword-aligned start of my LDS image
    const uint32_t o = start32 + before + incl - c;
    // pext(mask, pw), LSB = first passing position, as two 32-bit halves; the high half's result is shifted up by the low half's pass count
    const uint64_t mk = flip_bytes64(mk_raw);
    const uint32_t o_lo = pext32_lut(lut, (uint32_t)mk, pw_lo), o_hi = pext32_lut(lut, (uint32_t)(mk >> 32), pw_hi);
    const uint64_t out = (uint64_t)o_lo | ((uint64_t)o_hi << c_lo);
    if (out) {
        const uint32_t rel = o - obase;
        const uint32_t sh = rel & 31u, word = rel >> 5;
        const uint32_t lo = (uint32_t)out, hi = (uint32_t)(out >> 32);
        const uint32_t d0 = lo << sh;
        const uint32_t d1 = sh ? ((lo >> (32u - sh)) | (hi << sh)) : hi;
        const uint32_t d2 = sh ? (hi >> (32u - sh)) : 0u;
        if (d0) atomicOr(&buf[word], d0);
        if (d1) atomicOr(&buf[word + 1], d1);
        if (d2) atomicOr(&buf[word + 2], d2);
    }
    __syncthreads();
    const uint32_t oend = start32 + chunk_total;
    const uint32_t ndw = ((oend - obase) + 31u) >> 5;
    // a dword this workgroup shares with a neighbour (its first one when it does not start on a dword, its last one when it does not end on
    // one) was zeroed by k_chunk_offsets and is OR-ed into; every other dword of the range is written whole, zero or not
    for (uint32_t i = threadIdx.x; i < ndw; i += WG_THREADS) {
        const uint32_t v = buf[i];
        const bool shared = (i == 0 && (start32 & 31u)) || (i + 1 == ndw && (oend & 31u));
        if (shared) { if (v) atomicOr(&wit[(obase >> 5) + i], flip_bytes32(v)); }
        else wit[(obase >> 5) + i] = flip_bytes32(v);
    }
    if (threadIdx.x == 0 && wbeg + WG_THREADS >= total) stats[(uint64_t)f * 4 + 0] = oend;   // len(witness)
}

// A6 expand: out[i] = witness[rank(i)] where position i passes, else 0 (:299-304).  One lane per 64-position word, a workgroup per
// WG_THREADS consecutive words (= whole segments), offsets as in k_compact_witness: the start comes from k_chunk_offsets, the offsets
// inside the chunk are a block scan.  The lane's popc(pass) stream bits are fetched
// as one 64-bit window and dealt out to the set bits of the pass word through the pdep table.  Reads never leave the row.
__global__ __launch_bounds__(WG_THREADS) void k_expand_mask(
    const uint64_t *__restrict__ pass_words, const uint32_t *__restrict__ seg_cnt, uint64_t nseg, uint32_t words_per_seg,
    const uint32_t *__restrict__ witnesses, uint64_t witness_stride_words32,
    uint64_t *__restrict__ masks, uint64_t mask_stride_words64, uint64_t n, const uint32_t *__restrict__ chunk_off /* k_chunk_offsets */)
{
    __shared__ uint32_t wsum[WG_WAVES];
    __shared__ __attribute__((aligned(4))) uint8_t lut[256];
    const uint32_t f = blockIdx.y;
    const uint32_t lane = threadIdx.x & 63u, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t nwords = (uint32_t)((n + 63) >> 6);
    const uint32_t total = (uint32_t)nseg * words_per_seg;
    const uint32_t start32 = chunk_off[(uint64_t)f * gridDim.x + blockIdx.x];
    const uint32_t *wit = witnesses + (uint64_t)f * witness_stride_words32;
    const uint32_t wbeg = blockIdx.x * WG_THREADS;
    const uint32_t w = wbeg + threadIdx.x;
    const bool have = w < total && w < nwords;
    const uint64_t pw_raw = have ? pass_words[(uint64_t)f * total + w] : 0ull;
    lut_to_lds(lut, LUT_PDEP4);
    const uint64_t p = flip_bytes64(pw_raw);                      // packed -> bit b = position 64w + b
    const uint32_t p_lo = (uint32_t)p, p_hi = (uint32_t)(p >> 32);
    const uint32_t c_lo = __popc(p_lo), c = c_lo + __popc(p_hi);
    const uint32_t incl = wave_inclusive_scan(c);
    if (lane == WAVE - 1) wsum[wave] = incl;
    __syncthreads();
    uint32_t o = start32 + incl - c;
#pragma unroll
    for (int k = 0; k < WG_WAVES; ++k)
        if ((uint32_t)k < wave) o += wsum[k];
    if (w >= nwords) return;
    uint64_t out = 0;
    if (c) {
        // stream bits o ... o + c - 1 as a window with bit t = stream bit o + t (flip_bytes32: packed dword -> stream bit b at bit b)
        const uint32_t d0 = o >> 5, dl = (o + c - 1u) >> 5, r = o & 31u;
        const uint64_t x0 = d0 < witness_stride_words32 ? flip_bytes32(wit[d0]) : 0u;
        const uint64_t x1 = (d0 + 1 <= dl && d0 + 1 < witness_stride_words32) ? flip_bytes32(wit[d0 + 1]) : 0u;
        const uint64_t x2 = (d0 + 2 <= dl && d0 + 2 < witness_stride_words32) ? flip_bytes32(wit[d0 + 2]) : 0u;
        uint64_t win = (x0 | (x1 << 32)) >> r;
        win |= r ? x2 << (64u - r) : 0ull;
        out = (uint64_t)pdep32_lut(lut, (uint32_t)win, p_lo) | ((uint64_t)pdep32_lut(lut, (uint32_t)(win >> c_lo), p_hi) << 32);
    }
    masks[(uint64_t)f * mask_stride_words64 + w] = flip_bytes64(out);
}

}  // namespace rbf

namespace rbf {

// ------------------------------------------------------------------------------------------
// A1 fast path: residual masks of a whole GOP, every frame read from HBM exactly once.
// ------------------------------------------------------------------------------------------
// Frames must be flat (row pitch == width * pixel stride), so a frame is an array of n pixels of
// PIXEL_BYTES each whose first sample is the luma.  A lane owns 16 consecutive pixels for the
// whole GOP: it keeps the previous frame's 16*PIXEL_BYTES bytes in registers, streams the next
// frame's with 16-byte loads (two frames of prefetch in flight), and emits its 16 mask bits as one
// MSB-first uint16, so a wave writes 128 contiguous bytes per frame and needs no ballot.
template <typename SAMPLE, int PIXEL_BYTES>
struct LanePixels {
    static constexpr int DW = 16 * PIXEL_BYTES / 4;            // dwords per lane per frame
    uint32_t d[DW];
    template <bool NT>
    __device__ __forceinline__ void load(const uint8_t *p)
    {
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        const u32x4 *q = reinterpret_cast<const u32x4 *>(p);
#pragma unroll
        for (int i = 0; i < DW / 4; ++i) {
            const u32x4 v = NT ? __builtin_nontemporal_load(q + i) : q[i];
            d[4 * i] = v.x; d[4 * i + 1] = v.y; d[4 * i + 2] = v.z; d[4 * i + 3] = v.w;
        }
    }
    __device__ __forceinline__ uint32_t luma(int k) const     // sample 0 of pixel k
    {
        const int byte = k * PIXEL_BYTES;
        const uint32_t w = d[byte >> 2];
        if (sizeof(SAMPLE) == 1) return (w >> (8 * (byte & 3))) & 0xFFu;
        return (w >> (8 * (byte & 3))) & 0xFFFFu;              // 2-byte samples are 2-byte aligned
    }
};

// Threshold 0 (the lossless setting, verify_true_lossless.py:244-246): the bit is "luma changed", which needs no
// per-pixel extraction.  8-bit: XOR whole dwords, pull the luma bytes of four pixels into one dword (v_perm), turn
// "byte != 0" into the byte's top bit with the carry trick and squeeze the four flags into a nibble with one
// multiply: ~3 instructions per pixel instead of ~10.  16-bit: numpy's int16 arithmetic makes abs(a - b) > 0 false for
// a - b == 0 AND for a - b == 0x8000 (abs(-32768) stays negative, :801), i.e. exactly when a and b agree in their LOW 15 BITS -- so
// the bit is ((a ^ b) & 0x7FFF) != 0 and no subtraction is needed.  Planar 16-bit luma (BASELINE config 5): per dword (two pixels) one
// v_bitop3 ((a ^ b) & 0x7FFF7FFF), one v_pk_min_u16 against 0x00010001 (the two flags at bits 0 and 16) and one v_lshl_or into an
// accumulator whose halves interleave at the end: 26 vector instructions per 16 pixels (round 5's v_pk_sub / carry / shift chain: 62).
// `count_src`: a value with the same population count as the returned bits (the 16-bit path's accumulator: no masking needed).
// `one2`: 1 in both halves, made by the caller ONCE with an asm v_mov -- the compiler turns a visible min(t, 1) into v_cmp + v_cndmask
// through VCC (21 cycles a pair, profiles/r04_opbench2.txt), hence also the asm v_pk_min_u16.
template <typename SAMPLE, int PIXEL_BYTES>
__device__ __forceinline__ uint32_t lane_bits_thr0(const LanePixels<SAMPLE, PIXEL_BYTES> &a, const LanePixels<SAMPLE, PIXEL_BYTES> &b, uint32_t one2, uint32_t &count_src)
{
    uint32_t bits = 0;
    if (sizeof(SAMPLE) == 1) {
        constexpr int shift_of_group[4] = {4, 0, 12, 8};       // pixels 4g..4g+3 -> bits (k ^ 7), MSB-first per byte
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            uint32_t z;
            if (PIXEL_BYTES == 1) z = a.d[g] ^ b.d[g];
            else {                                              // 3-byte pixels: lumas at bytes 0, 3, 6, 9 of three dwords
                const uint32_t x0 = a.d[3 * g] ^ b.d[3 * g], x1 = a.d[3 * g + 1] ^ b.d[3 * g + 1], x2 = a.d[3 * g + 2] ^ b.d[3 * g + 2];
                const uint32_t y = __builtin_amdgcn_perm(x1, x0, 0x00060300u);   // {x0.b0, x0.b3, x1.b2, -}
                z = __builtin_amdgcn_perm(x2, y, 0x05020100u);                   // {.., .., .., x2.b1}
            }
            const uint32_t f = (((z & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | z) & 0x80808080u;   // top bit of every non-zero byte
            bits |= (((f >> 7) * 0x08040201u) >> 24) << shift_of_group[g];            // flags at 0,8,16,24 -> 27,26,25,24
        }
        count_src = bits;
    } else if (PIXEL_BYTES == 2) {
        const uint32_t k7 = 0x7FFF7FFFu;
        uint32_t acc = 0;                                       // (one2: 0x00010001 in a VGPR the compiler cannot see through, see the kernel)                                       // dword j's flags: pixel 2j at bit 2 (j ^ 3), pixel 2j+1 sixteen bits above
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const int j = s ^ 4;                                // order 4,5,6,7,0,1,2,3: the first dword ends up highest
            const uint32_t t = (a.d[j] ^ b.d[j]) & k7;
            uint32_t g;
            asm("v_pk_min_u16 %0, %1, %2" : "=v"(g) : "v"(t), "v"(one2));
            acc = (acc << 2) | g;
        }
        bits = (acc >> 16) | (acc << 1);                        // pixel 2j -> odd bit (2j) ^ 7, pixel 2j+1 -> the bit below; bits 16.. are never stored
        count_src = acc;
    } else {
        typedef unsigned short h2 __attribute__((ext_vector_type(2)));
#pragma unroll
        for (int j = 0; j < 8; ++j) {                           // pixels 2j, 2j+1; 6-byte pixels: luma = low half of dword 3j, high half of dword 3j+1
            const h2 t0 = __builtin_bit_cast(h2, a.d[3 * j]) - __builtin_bit_cast(h2, b.d[3 * j]);
            const h2 t1 = __builtin_bit_cast(h2, a.d[3 * j + 1]) - __builtin_bit_cast(h2, b.d[3 * j + 1]);
            const uint32_t d = (__builtin_bit_cast(uint32_t, t0) & 0x0000FFFFu) | (__builtin_bit_cast(uint32_t, t1) & 0xFFFF0000u);
            const uint32_t f = ((d & 0x7FFF7FFFu) + 0x7FFF7FFFu) & 0x80008000u;          // top bit of every half with (d & 0x7FFF) != 0
            const uint32_t two = ((f >> 14) & 2u) | (f >> 31);                          // pixel 2j -> bit 1, pixel 2j+1 -> bit 0
            bits |= two << (((2 * j + 1) ^ 7));
        }
        count_src = bits;
    }
    return bits;
}

// What used to be k_finish_ones' job, folded into the mask kernel when it covers the whole frame (rbf_encode_gop on frames of whole
// 1024-pixel segments): every workgroup clears its share of up to two output regions (round 5: only the stats of the batch -- the
// witness rows are no longer cleared, k_chunk_offsets zeroes the few words compaction ORs into; region a is null), and
// the LAST workgroup to finish (a ticket) hands the counts out -- to the caller's array and into the device-visible pinned block
// whose token the host spins on -- and re-zeroes the accumulator and the ticket.  One launch less per step, and the publish no
// longer queues behind whatever else occupies the GPU (profiles/r02_overlap_4pipelines.txt: the 5 us k_finish_ones took 61 us
// under overlap).
constexpr uint32_t MASK_TICKETS = 64;           // first-level ticket counters; one more dword for the second level
struct MaskFinish {
    uint32_t enabled;              // 0: the caller runs k_finish_ones
    uint32_t count;                // pairs
    uint32_t *ticket;              // context-owned: MASK_TICKETS + 1 counters, zero between launches
    uint64_t *ones_out;            // caller's device array
    uint64_t *host_block;          // nullable: [token | ones...] in pinned host memory
    uint64_t token;
    uint4 *clear_a; uint64_t quads_a;
    uint4 *clear_b; uint64_t quads_b;
};

// The temporal chunks of one launch (blockIdx.y).  count == 0: uniform chunks of `ppc` pairs over the whole block (one run).  Otherwise
// chunk y reads frames first[y] .. first[y] + pairs[y] and writes the mask rows first[y] .. first[y] + pairs[y] - 1 -- so a block that
// holds SEVERAL keyframe-delimited runs (rbf_encode_runs) is cut at the keyframes: no chunk reads across one, the pair in front of a
// keyframe is never diffed.  pairs[y] & MASK_CHUNK_SKIP: those pairs are NOT coded; their rows are written as zeros, nothing is counted.
constexpr uint32_t MASK_MAX_CHUNKS = 256, MASK_CHUNK_SKIP = 0x8000u;
struct MaskChunks {
    uint32_t count, ppc;
    uint16_t first[MASK_MAX_CHUNKS], pairs[MASK_MAX_CHUNKS];
};

template <typename SAMPLE, int PIXEL_BYTES, bool NT = false, bool THR0 = false>
__global__ __launch_bounds__(WG_THREADS) void k_residual_mask_gop(
    const uint8_t *__restrict__ frames, uint64_t frame_stride, uint32_t nframes, uint64_t nsegs /* of 1024 px */,
    int32_t thr_all, const int32_t *__restrict__ thr_tab /* nullable: per pair */,
    uint16_t *__restrict__ masks, uint64_t mask_stride_u16, uint64_t *__restrict__ ones,
    const MaskChunks chunks, const MaskFinish fin)
{
    // blockIdx.y = temporal chunk: frames [f0, f1] (f1 - f0 pairs); chunks of one run overlap by one frame, which
    // buys gridDim.y times more waves in flight for ~gridDim.y/nframes extra reads
    extern __shared__ uint32_t cnt[];                          // [nframes-1] per-workgroup ones
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint64_t seg = (uint64_t)blockIdx.x * WG_WAVES + wave;
    uint32_t f0, f1;
    bool skipped = false;
    if (chunks.count) {
        const uint32_t c = chunks.pairs[blockIdx.y];
        f0 = chunks.first[blockIdx.y];
        f1 = f0 + (c & (MASK_CHUNK_SKIP - 1u));
        skipped = (c & MASK_CHUNK_SKIP) != 0u;
    } else {
        f0 = blockIdx.y * chunks.ppc;
        f1 = f0 + chunks.ppc < nframes - 1 ? f0 + chunks.ppc : nframes - 1;
    }
    for (uint32_t i = threadIdx.x; i + 1 < nframes; i += WG_THREADS) cnt[i] = 0;
    __syncthreads();
    if (skipped) {
        if (seg < nsegs)
            for (uint32_t f = f0; f < f1; ++f) masks[seg * 64 + lane + (uint64_t)f * mask_stride_u16] = 0;
    } else if (seg < nsegs && f0 < f1) {
        using LP = LanePixels<SAMPLE, PIXEL_BYTES>;
        const uint64_t lane_off = (seg * 1024 + (uint64_t)lane * 16) * PIXEL_BYTES;
        const uint8_t *p = frames + lane_off;
        uint16_t *out = masks + seg * 64 + lane;
        uint32_t one2;
        asm volatile("v_mov_b32 %0, 0x10001" : "=v"(one2));
        LP fa, fb, fc, fd;                                        // four frames in registers, roles rotate: TWO loads are in flight while a pair is compared
        fa.template load<NT>(p + (uint64_t)f0 * frame_stride);
        fb.template load<NT>(p + (uint64_t)(f0 + 1) * frame_stride);
        if (f0 + 2 <= f1) fc.template load<NT>(p + (uint64_t)(f0 + 2) * frame_stride);
        // pair (prev, cur) = mask f-1; `nxt2` receives frame f+2 meanwhile (frame f+1 is already on its way).  Returns a value whose
        // population count is this lane's number of set bits.
        auto step = [&](const LP &prev, const LP &cur, LP &nxt2, uint32_t f) -> uint32_t {
            if (f + 2 <= f1) nxt2.template load<NT>(p + (uint64_t)(f + 2) * frame_stride);
            const int32_t thr = thr_tab ? thr_tab[f - 1] : thr_all;
            uint32_t bits = 0, csrc = 0;
            if (THR0) bits = lane_bits_thr0<SAMPLE, PIXEL_BYTES>(prev, cur, one2, csrc);      // host: no per-pair table and thr == 0
            else {
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    const bool b = residual_bit<SAMPLE>((SAMPLE)prev.luma(k), (SAMPLE)cur.luma(k), thr);
                    bits |= (b ? 1u : 0u) << (k ^ 7);         // MSB-first within each byte
                }
                csrc = bits;
            }
            out[(uint64_t)(f - 1) * mask_stride_u16] = (uint16_t)bits;
            return csrc;
        };
        // The wave's counts stay in ONE register until the chunk ends: lane s of `tally` holds the ones of pairs base + 2s (low half) and
        // base + 2s + 1 (high half).  Two pairs share a DPP tree (their lane counts ride as packed 16-bit halves, each total <= 1024),
        // v_readlane hands the packed totals to the scalar unit and ONE v_writelane files them.  Round 5 did a six-step tree, a compare, two exec
        // masks and the compiler's uniform-address atomic loop (~8 vector + ~15 scalar instructions and an LDS atomic) PER PAIR -- in a
        // kernel that runs underneath the issue-bound insert / query kernels of the neighbouring pipelines, where every instruction it
        // issues is one of theirs that waits.
        uint32_t tally = 0, base = f0;
        auto flush = [&]() {
            const uint32_t i = base + 2u * lane;
            if (i < f1 && (tally & 0xFFFFu)) atomicAdd(&cnt[i], tally & 0xFFFFu);     // (per-lane addresses: one ds_add_u32 for the wave)
            if (i + 1u < f1 && (tally >> 16)) atomicAdd(&cnt[i + 1u], tally >> 16);
            tally = 0;
        };
        auto file2 = [&](uint32_t c_even, uint32_t c_odd, uint32_t f) {               // counts of pairs f-1 and f
            const uint32_t tot = wave_sum_to_lane63(__popc(c_even) | (__popc(c_odd) << 16));
            const uint32_t t = (uint32_t)__builtin_amdgcn_readlane((int)tot, 63);
            const uint32_t slot = (f - 1u - base) >> 1;                               // < 64 (scalar: f and base are uniform)
            uint32_t keep;
            // v_writelane takes ONE SGPR over the constant bus, so the lane select rides in M0 (reserved: saved and restored, as RowDmaC does)
            asm volatile("s_mov_b32 %1, m0\n\t"
                         "s_mov_b32 m0, %3\n\t"
                         "s_nop 0\n\t"
                         "v_writelane_b32 %0, %2, m0\n\t"
                         "s_mov_b32 m0, %1"
                         : "+v"(tally), "=&s"(keep) : "s"(t), "s"(slot));
        };
        // unrolled by four so that the rotation prev <- cur <- nxt <- nxt2 costs no register moves
        for (uint32_t f = f0 + 1; f <= f1; f += 4) {
            if (f - 1u - base == 2u * WAVE) { flush(); base += 2u * WAVE; }
            const uint32_t c0 = step(fa, fb, fd, f);
            const uint32_t c1 = f + 1 <= f1 ? step(fb, fc, fa, f + 1) : 0u;
            file2(c0, c1, f);
            if (f + 2 <= f1) {
                const uint32_t c2 = step(fc, fd, fb, f + 2);
                const uint32_t c3 = f + 3 <= f1 ? step(fd, fa, fc, f + 3) : 0u;
                file2(c2, c3, f + 2);
            }
        }
        flush();
    }
    __syncthreads();
    for (uint32_t i = f0 + threadIdx.x; i < f1; i += WG_THREADS)
        if (cnt[i]) atomicAdd((unsigned long long *)&ones[i], (unsigned long long)cnt[i]);
    if (!fin.enabled) return;
    // ---- the tail of the pass (see MaskFinish).  Only wave 0 -- whose lanes issued the workgroup's count atomics -- takes a ticket.
    const uint64_t wg = (uint64_t)blockIdx.y * gridDim.x + blockIdx.x, nwg = (uint64_t)gridDim.x * gridDim.y;
    const uint4 z = make_uint4(0, 0, 0, 0);
    const bool wide = f1 > f0 + WAVE;              // (workgroup-uniform) more than 64 pairs in this chunk: waves 1..3 issued count atomics too
    if (wide) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    if (wave != 0) {                               // waves 1..3: their share of the clears, and out
        for (uint64_t i = wg * WG_THREADS + threadIdx.x; i < fin.quads_a; i += nwg * WG_THREADS) fin.clear_a[i] = z;
        for (uint64_t i = wg * WG_THREADS + threadIdx.x; i < fin.quads_b; i += nwg * WG_THREADS) fin.clear_b[i] = z;
        return;
    }
    // My counts must have been performed before my ticket is.  They are agent-scope atomics (carried out at the device's coherence
    // point, not in this XCD's L2), so waiting for their acknowledgements is enough -- a __threadfence() here writes the L2 back
    // from every workgroup and made the kernel 8x slower (25 -> 190 us).
    if (!wide) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // Two-level ticket: all ~2 000 workgroups of a 1080p GOP are resident at once and finish together, and returning atomics on ONE
    // address complete one every ~6 ns -- a single counter cost the kernel 13 us.  64 first-level counters (workgroup id mod 64),
    // whose last arrivals meet on a second-level one.
    uint32_t is_last = 0;
    if (lane == 0) {
        const uint32_t idx = (uint32_t)(wg % MASK_TICKETS);
        const uint32_t mine = (uint32_t)((nwg + MASK_TICKETS - 1 - idx) / MASK_TICKETS);      // workgroups on this counter
        if (atomicAdd(fin.ticket + idx, 1u) == mine - 1u) {
            __hip_atomic_store(fin.ticket + idx, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const uint32_t groups = nwg < MASK_TICKETS ? (uint32_t)nwg : (uint32_t)MASK_TICKETS;
            is_last = atomicAdd(fin.ticket + MASK_TICKETS, 1u) == groups - 1u ? 1u : 0u;
        }
    }
    is_last = __builtin_amdgcn_readfirstlane(is_last);
    for (uint64_t i = wg * WG_THREADS + threadIdx.x; i < fin.quads_a; i += nwg * WG_THREADS) fin.clear_a[i] = z;
    for (uint64_t i = wg * WG_THREADS + threadIdx.x; i < fin.quads_b; i += nwg * WG_THREADS) fin.clear_b[i] = z;
    if (!is_last) return;
    // the last workgroup: counts out (to the caller's array and the host), accumulator and ticket back to zero
    for (uint32_t i = lane; i < fin.count; i += WAVE) {
        const uint64_t v = __hip_atomic_load(&ones[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // (the other workgroups added at the coherence point)
        fin.ones_out[i] = v;
        __hip_atomic_store(&ones[i], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (fin.host_block) __hip_atomic_store(&fin.host_block[1 + i], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    if (fin.host_block) {
        __threadfence_system();
        if (lane == 0) __hip_atomic_store(&fin.host_block[0], fin.token, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    if (lane == 0) __hip_atomic_store(fin.ticket + MASK_TICKETS, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

}  // namespace rbf
