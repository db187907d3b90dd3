// rbf_device.h -- gfx950 device-side building blocks (wave64, HIP).
//
// XXH64 of the decimal ASCII key str(i), modular double hashing without 2^64 wrap, and the
// MSB-first bit addressing used by every packed bit vector at the ABI.
// Reference semantics: improved_video_compressor.py:65-97 (see include/rbf.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rbf {

constexpr uint64_t P1 = 0x9E3779B185EBCA87ULL;
constexpr uint64_t P2 = 0xC2B2AE3D27D4EB4FULL;
constexpr uint64_t P3 = 0x165667B19E3779F9ULL;
constexpr uint64_t P4 = 0x85EBCA77C2B2AE63ULL;
constexpr uint64_t P5 = 0x27D4EB2F165667C5ULL;

constexpr int WAVE = 64;

// Per-frame filter geometry as the kernels see it.
struct FrameDev {
    uint32_t m;        // filter bits
    uint32_t floor_k;  // deterministic probes
    uint64_t T;        // activation threshold: extra probe iff h_act < T
    uint64_t M;        // floor(2^64 / m) for m >= 2 (Barrett reciprocal); unused when m == 1
};

struct Seeds { uint64_t h1, h2, act; };

// A batch's geometry travels BY VALUE in the kernel-argument segment (3 KiB of the 4 KiB limit):
// no upload, no device buffer, and the per-frame fields arrive through scalar loads.
constexpr int MAX_BATCH = 128;
struct FrameTable { FrameDev f[MAX_BATCH]; };
struct SliceTable { uint8_t n[MAX_BATCH]; };     // insert: partial filters (mask slices) per frame, 0 = frame not coded

// 32-bit words of an m-bit filter; m may be 2^32 - 1, so the rounding is done in 64 bits
__device__ __forceinline__ uint32_t filter_words(uint32_t m) { return (uint32_t)(((uint64_t)m + 31u) >> 5); }

__device__ __forceinline__ uint64_t rotl64(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }

// Packed vectors are MSB-first per byte (numpy.packbits).  In a little-endian 32-bit word the
// stream bit i sits at position (i & 31) ^ 7.
__device__ __forceinline__ uint32_t msb_pos(uint32_t i) { return (i & 31u) ^ 7u; }
__device__ __forceinline__ uint32_t msb_bit(uint32_t i) { return 1u << msb_pos(i); }

// natural (bit i at position i) <-> MSB-first-per-byte, 32- and 64-bit words (an involution).
__device__ __forceinline__ uint32_t flip_bytes32(uint32_t x) { return __builtin_bswap32(__builtin_bitreverse32(x)); }
__device__ __forceinline__ uint64_t flip_bytes64(uint64_t x) { return __builtin_bswap64(__builtin_bitreverse64(x)); }
// packed dword -> stream bit b at bit 31 - b (MSB-first across the whole dword)
__device__ __forceinline__ uint32_t flip_order32(uint32_t x) { return __builtin_bswap32(x); }

__device__ __forceinline__ uint32_t lane_id() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }
// number of set bits of `mask` strictly below this lane
__device__ __forceinline__ uint32_t rank_below(uint64_t mask)
{
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

// ---- decimal key -----------------------------------------------------------------------------
// str(v) as little-endian bytes: byte 0 = most significant digit.  len <= 10 for uint32.
struct DecKey { uint64_t lo; uint32_t hi; uint32_t len; };

__device__ __forceinline__ DecKey make_key(uint32_t v)
{
    DecKey k; k.lo = 0; k.hi = 0; k.len = 0;
    do {
        const uint32_t q = v / 10u;
        const uint32_t d = v - q * 10u;
        k.hi = (k.hi << 8) | (uint32_t)(k.lo >> 56);
        k.lo = (k.lo << 8) | (uint64_t)(0x30u + d);
        v = q;
        ++k.len;
    } while (v);
    return k;
}

// XXH64(key, seed) for keys shorter than 32 bytes (public xxHash spec, "small input" path).
__device__ __forceinline__ uint64_t xxh64_key(const DecKey &k, uint64_t seed)
{
    uint64_t h = seed + P5 + (uint64_t)k.len;
    uint64_t rest;
    uint32_t cnt;
    if (k.len >= 8) {
        uint64_t k1 = k.lo * P2;
        k1 = rotl64(k1, 31) * P1;
        h ^= k1;
        h = rotl64(h, 27) * P1 + P4;
        rest = k.hi; cnt = k.len - 8;              // <= 2 bytes left: no 4-byte round
    } else if (k.len >= 4) {
        h ^= (uint64_t)(uint32_t)k.lo * P1;
        h = rotl64(h, 23) * P2 + P3;
        rest = k.lo >> 32; cnt = k.len - 4;
    } else {
        rest = k.lo; cnt = k.len;
    }
    for (uint32_t t = 0; t < cnt; ++t) {
        h ^= (rest & 0xFFu) * P5;
        h = rotl64(h, 11) * P1;
        rest >>= 8;
    }
    h ^= h >> 33; h *= P2;
    h ^= h >> 29; h *= P3;
    h ^= h >> 32;
    return h;
}

// ---- the three hashes of one index, tuned -------------------------------------------------------
// Neighbouring indices have the same number of decimal digits, so a wave almost always agrees on
// the key length: for the common lengths (5, 6, 7 digits: indices 10^4 .. 10^7-1, i.e. 99.5 % of a
// 1080p frame and 99.9 % of a 2160p one) the XXH64 short-input path is expanded at compile time --
// independent digit extractions (one magic multiply each) instead of a serial divide-by-ten loop,
// no per-byte loops, and the three seeds' dependency chains side by side.
struct Hash3 { uint64_t h1, h2, ha; };

template <int LEN>
__device__ __forceinline__ uint64_t xxh64_digits(const uint32_t (&d)[LEN], uint64_t seed)   // d[0] = most significant
{
    static_assert(LEN >= 5 && LEN <= 7, "4-byte round + 1..3 byte rounds");
    uint64_t h = seed + P5 + (uint64_t)LEN;
    const uint32_t w = 0x30303030u + d[0] + (d[1] << 8) + (d[2] << 16) + (d[3] << 24);
    h ^= (uint64_t)w * P1;
    h = rotl64(h, 23) * P2 + P3;
#pragma unroll
    for (int t = 4; t < LEN; ++t) {
        h ^= (uint64_t)(0x30u + d[t]) * P5;
        h = rotl64(h, 11) * P1;
    }
    h ^= h >> 33; h *= P2;
    h ^= h >> 29; h *= P3;
    h ^= h >> 32;
    return h;
}

template <int LEN>
__device__ __forceinline__ Hash3 hash3_fixed(uint32_t v, const Seeds &s)
{
    constexpr uint32_t P10[8] = {1u, 10u, 100u, 1000u, 10000u, 100000u, 1000000u, 10000000u};
    uint32_t q[LEN + 1];                                     // q[k] = v / 10^k
#pragma unroll
    for (int k = 0; k <= LEN; ++k) q[k] = k < LEN ? v / P10[k] : 0u;
    uint32_t d[LEN];
#pragma unroll
    for (int k = 0; k < LEN; ++k) d[LEN - 1 - k] = q[k] - 10u * q[k + 1];
    Hash3 r;
    r.h1 = xxh64_digits<LEN>(d, s.h1);
    r.h2 = xxh64_digits<LEN>(d, s.h2);
    r.ha = xxh64_digits<LEN>(d, s.act);
    return r;
}

__device__ __forceinline__ Hash3 hash3_generic(uint32_t v, const Seeds &s)
{
    const DecKey key = make_key(v);
    Hash3 r;
    r.h1 = xxh64_key(key, s.h1);
    r.h2 = xxh64_key(key, s.h2);
    r.ha = xxh64_key(key, s.act);
    return r;
}

// `active`: lanes whose v is meaningful (the others may hold anything; they get a well-defined but
// unused result).  The fast paths are taken only when every active lane has the same length.
__device__ __forceinline__ Hash3 hash3_index(uint32_t v, bool active, const Seeds &s)
{
    const bool in7 = v >= 1000000u && v < 10000000u;
    const bool in6 = v >= 100000u && v < 1000000u;
    const bool in5 = v >= 10000u && v < 100000u;
    if (__all(!active || in7)) return hash3_fixed<7>(active ? v : 1000000u, s);
    if (__all(!active || in6)) return hash3_fixed<6>(active ? v : 100000u, s);
    if (__all(!active || in5)) return hash3_fixed<5>(active ? v : 10000u, s);
    return hash3_generic(active ? v : 0u, s);
}

// ---- the hashes of 8 CONSECUTIVE indices (the query kernel's lane) ------------------------------
// str(i) and str(i+1) differ in the last character only (unless a decade ends), and XXH64's short-input
// path consumes the key front to back: the state after all characters but the last is the same for the
// up-to-ten indices of a decade.  A lane owning indices v0 .. v0+7 touches at most two decades, so it runs the
// 4-byte round and the inner byte rounds twice (decade A = v0/10, decade B = A+1) instead of eight times
// and only the last byte round + avalanche per index.  Taken when every lane of the wave agrees on a key
// length of 5, 6 or 7 for all of its indices; anything else goes through hash3_index.
struct Prefix3 { uint64_t h1, h2, ha; };

template <int LEN>
__device__ __forceinline__ Prefix3 hash3_decade_prefix(uint32_t decade /* LEN-1 digits */, const Seeds &s)
{
    static_assert(LEN >= 5 && LEN <= 7, "key = decade digits + one more character");
    constexpr uint32_t P10[7] = {1u, 10u, 100u, 1000u, 10000u, 100000u, 1000000u};
    uint32_t q[LEN];                                          // q[k] = decade / 10^k
#pragma unroll
    for (int k = 0; k < LEN; ++k) q[k] = k < LEN - 1 ? decade / P10[k] : 0u;
    uint32_t d[LEN - 1];                                      // d[0] = most significant digit
#pragma unroll
    for (int k = 0; k < LEN - 1; ++k) d[LEN - 2 - k] = q[k] - 10u * q[k + 1];
    const uint64_t lane4 = (uint64_t)(0x30303030u + d[0] + (d[1] << 8) + (d[2] << 16) + (d[3] << 24)) * P1;
    uint64_t h[3] = {s.h1 + P5 + (uint64_t)LEN, s.h2 + P5 + (uint64_t)LEN, s.act + P5 + (uint64_t)LEN};
#pragma unroll
    for (int c = 0; c < 3; ++c) h[c] = rotl64(h[c] ^ lane4, 23) * P2 + P3;
#pragma unroll
    for (int t = 4; t < LEN - 1; ++t) {
        const uint64_t term = (uint64_t)(0x30u + d[t]) * P5;
#pragma unroll
        for (int c = 0; c < 3; ++c) h[c] = rotl64(h[c] ^ term, 11) * P1;
    }
    return Prefix3{h[0], h[1], h[2]};
}

__device__ __forceinline__ uint64_t xxh64_last_byte(uint64_t h, uint64_t term /* byte * P5 */)
{
    h = rotl64(h ^ term, 11) * P1;
    h ^= h >> 33; h *= P2;
    h ^= h >> 29; h *= P3;
    h ^= h >> 32;
    return h;
}

// Hashes of v0 + j, j = 0..7.  `valid` bit j: index v0 + j is meaningful.  Returns false (nothing written)
// when the wave cannot take the shared-prefix path; the caller then hashes index by index.
template <int LEN>
__device__ __forceinline__ void hash3_run8_fixed(uint32_t v0, const Seeds &s, uint64_t (&h1)[8], uint64_t (&h2)[8], uint64_t (&ha)[8])
{
    const uint32_t decade = v0 / 10u, r0 = v0 - decade * 10u;
    const Prefix3 a = hash3_decade_prefix<LEN>(decade, s);
    const Prefix3 b = hash3_decade_prefix<LEN>(decade + 1u, s);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const uint32_t t = r0 + (uint32_t)j;
        const bool in_b = t >= 10u;
        const uint64_t term = (uint64_t)(0x30u + (in_b ? t - 10u : t)) * P5;
        h1[j] = xxh64_last_byte(in_b ? b.h1 : a.h1, term);
        h2[j] = xxh64_last_byte(in_b ? b.h2 : a.h2, term);
        ha[j] = xxh64_last_byte(in_b ? b.ha : a.ha, term);
    }
}

__device__ __forceinline__ bool hash3_run8(uint32_t v0, uint32_t valid, const Seeds &s, uint64_t (&h1)[8], uint64_t (&h2)[8], uint64_t (&ha)[8])
{
    // every index of an active lane must have the same length; a lane whose run ends inside the frame's
    // last segment may own fewer than 8 valid indices -- their (unused) neighbours still hash fine as long
    // as the length agrees, so the test is on the whole run v0 .. v0+7
    const bool active = valid != 0u;
    const uint32_t v7 = v0 + 7u;
    const bool in7 = v0 >= 1000000u && v7 < 10000000u;
    const bool in6 = v0 >= 100000u && v7 < 1000000u;
    const bool in5 = v0 >= 10000u && v7 < 100000u;
    if (__all(!active || in7)) { hash3_run8_fixed<7>(active ? v0 : 1000000u, s, h1, h2, ha); return true; }
    if (__all(!active || in6)) { hash3_run8_fixed<6>(active ? v0 : 100000u, s, h1, h2, ha); return true; }
    if (__all(!active || in5)) { hash3_run8_fixed<5>(active ? v0 : 10000u, s, h1, h2, ha); return true; }
    return false;
}

// Hashes of v0 + j, j = 0..3 (a lane of k_query_p4 owns 4 consecutive indices): the same decade-prefix sharing as
// hash3_run8 -- a run of 4 touches at most two decades.  Returns false (nothing written) when the wave cannot take the
// shared-prefix path; the caller then hashes index by index.
template <int LEN>
__device__ __forceinline__ void hash3_run4_fixed(uint32_t v0, const Seeds &s, uint64_t (&h1)[4], uint64_t (&h2)[4], uint64_t (&ha)[4])
{
    const uint32_t decade = v0 / 10u, r0 = v0 - decade * 10u;
    const Prefix3 a = hash3_decade_prefix<LEN>(decade, s);
    const Prefix3 b = hash3_decade_prefix<LEN>(decade + 1u, s);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const uint32_t t = r0 + (uint32_t)j;
        const bool in_b = t >= 10u;
        const uint64_t term = (uint64_t)(0x30u + (in_b ? t - 10u : t)) * P5;
        h1[j] = xxh64_last_byte(in_b ? b.h1 : a.h1, term);
        h2[j] = xxh64_last_byte(in_b ? b.h2 : a.h2, term);
        ha[j] = xxh64_last_byte(in_b ? b.ha : a.ha, term);
    }
}

__device__ __forceinline__ bool hash3_run4(uint32_t v0, uint32_t valid, const Seeds &s, uint64_t (&h1)[4], uint64_t (&h2)[4], uint64_t (&ha)[4])
{
    const bool active = valid != 0u;
    const uint32_t v3 = v0 + 3u;
    const bool in7 = v0 >= 1000000u && v3 < 10000000u;
    const bool in6 = v0 >= 100000u && v3 < 1000000u;
    const bool in5 = v0 >= 10000u && v3 < 100000u;
    if (__all(!active || in7)) { hash3_run4_fixed<7>(active ? v0 : 1000000u, s, h1, h2, ha); return true; }
    if (__all(!active || in6)) { hash3_run4_fixed<6>(active ? v0 : 100000u, s, h1, h2, ha); return true; }
    if (__all(!active || in5)) { hash3_run4_fixed<5>(active ? v0 : 10000u, s, h1, h2, ha); return true; }
    return false;
}

// h mod m, exact, via Barrett with M = floor(2^64/m): q in {floor(h/m)-1, floor(h/m)}.
__device__ __forceinline__ uint32_t mod_m(uint64_t h, uint32_t m, uint64_t M)
{
    if (m == 1u) return 0u;
    const uint64_t q = __umul64hi(h, M);
    uint64_t r = h - q * (uint64_t)m;
    if (r >= m) r -= m;
    return (uint32_t)r;
}

// The probe sequence of one key: positions (h1 + j*h2) mod m for j = 0..floor_k-1, plus
// j = floor_k when activated -- evaluated as ((h1 mod m) + j*(h2 mod m)) mod m, which equals the
// reference's unbounded-integer expression (improved_video_compressor.py:81).
struct Probe {
    uint32_t pos;    // current position
    uint32_t step;   // h2 mod m
    bool extra;      // activation decision
};

__device__ __forceinline__ Probe make_probe(uint32_t index, const FrameDev &f, const Seeds &s)
{
    const DecKey key = make_key(index);
    Probe p;
    p.pos = mod_m(xxh64_key(key, s.h1), f.m, f.M);
    p.step = mod_m(xxh64_key(key, s.h2), f.m, f.M);
    p.extra = xxh64_key(key, s.act) < f.T;
    return p;
}

__device__ __forceinline__ void advance(Probe &p, uint32_t m)
{
    const uint64_t s = (uint64_t)p.pos + (uint64_t)p.step;
    p.pos = (uint32_t)(s >= m ? s - m : s);
}

}  // namespace rbf
