// rbf_kernels_q64.h -- what the FP64 kernels share (k_query_u64, k_query_s64t, k_insert_tab, k_insert_positions; filters of
// 2^15 <= m < 2^23 bits): the exact h mod m through one v_fma_f64, the probe image, activation ranks, the pixel-index hash table's
// layout and the LDS-DMA of an image row.  (Rounds 2 and 3 kept their query kernels k_query_f64 / f64t / p4 / r64 here and in
// rbf_kernels_r64.h; they are history now: git keeps them, tools/legacy/ up to round 4.)
#pragma once
#include "rbf_kernels_lds.h"

namespace rbf {

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
// (tests/c/mod_f64_check.c restates these five steps in C) and by the GPU parity tests.  (rows_reduce4, rbf_kernels_s64.h,
// takes the remainder as a full signed 32-bit number with one v_mad_u64_u32 instead: no sign extension.)
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

// The FP64 query kernels probe a PROBE IMAGE of the filter: dword w of the image is ~bswap(packed dword w), i.e. stream bit i of
// the dword sits at bit 31 - i, inverted.  A probe is then
//     fail = (image[pos >> 5] << (pos & 31)) | fail          (v_lshl_or_b32: the shifter takes pos's low 5 bits itself)
// and the sign bit of `fail` says "some probed filter bit is 0" -- no xor for the MSB-first bit order, no and-tree.
// The image is written by k_filter_reduce next to the packed filter (encode) or by k_probe_image (decode).
// The activated extra probe is made unconditional by steering the non-activated pixels to SAFE, a dword past the
// filter that the kernel keeps 0 in its LDS buffers ("bit set"): one v_cndmask instead of a masked merge.
__device__ __forceinline__ uint32_t probe_image_word(uint32_t lds_base_bytes /* in a VGPR */, uint32_t pos)
{
    uint32_t w;
    asm("v_lshrrev_b32 %0, 5, %1" : "=v"(w) : "v"(pos));         // opaque, or the compiler rewrites shift / shift-add as shift / and / add
    const uint32_t addr = (w << 2) + lds_base_bytes;
    return *reinterpret_cast<const __attribute__((address_space(3))) uint32_t *>((uintptr_t)addr);
}

// ---- activation ranks instead of the 64-bit activation hash ----------------------------------------------------
// "h_act < T_f" is all a frame ever asks of a pixel's activation hash, and a batch has at most MAX_BATCH thresholds.  With the
// coded frames' thresholds sorted,
//     rank(pixel) = #{ j : t_j <= h_act },          c_f = #{ j : t_j < T_f }   (host),          h_act < T_f  <=>  rank <= c_f
// (=>: every t_j <= h_act is < T_f; <=: if h_act >= T_f then T_f and everything below it are counted, rank > c_f).  The rank is
// computed once per pixel next to the hashes and kept in ONE BYTE: 2 registers per lane instead of 16.
// The compare (rank byte B of `ranks` <= c) and the select, with the lane mask in an SGPR pair so that two pixels do not serialise
// on VCC (v_cndmask with VCC costs ~21 cycles per wave-instruction on gfx950, with an SGPR pair ~4: profiles/r04_opbench2.txt):
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

// ---- the pixel-index hash table ----------------------------------------------------------------------------------
// What k_insert_tab gathers instead of hashing (rbf_kernels_i64.h): three arrays over the frame's indices, padded to whole 512-index
// segments (`hash_table_entries`), 26 bytes per index inside an allocation of 32:
//     pos[slot]   16 bytes: h1, h2                 -- ONE 16-byte load per key.  Rounds 2 and 3 kept 32-byte entries (RN(h1), RN(h2) as
//                                                    doubles | low dwords | h_act) and read them with two loads: the texture addresser
//                                                    walks the 64 scattered lanes of every load instruction, and the second one cost 6 of
//                                                    the kernel's 36 us (profiles/r04_insert_gather_ablation.txt).  RN(h) is rebuilt from
//                                                    h with two v_cvt_f64_u32 and one fma (one rounding: the same double).
//     tag[index]  2 bytes: h_act >> 48             -- "h_act < T" is decided by the tags unless they are equal (2^-16 of the keys) ...
//     act[slot]   8 bytes: h_act                   -- ... and only then read in full.
// pos and act are SLOT-MAJOR inside a 512-index segment -- the entry of index seg * 512 + lane * 8 + it sits at seg * 512 + it * 64 +
// lane -- because that is the order in which the kernels that produce the table (a lane owns 8 consecutive indices) can store it with
// fully coalesced wave stores; the tags are index-major (a lane stores its eight as one 16-byte piece).  (pos / act index-major as well, so
// that neighbouring set pixels share cache lines: +0.7 % on the four-pipeline step, -4 % with one pipeline, whose query kernel rewrites
// the table with uncoalesced stores then -- profiles/r04_insert_gather_ablation.txt, last block; not kept.)
__device__ __forceinline__ uint32_t hash_table_slot(uint32_t index)
{
    return (index & ~511u) | ((index & 7u) << 6) | ((index >> 3) & 63u);
}
__host__ __device__ __forceinline__ uint64_t hash_table_entries(uint64_t n) { return (n + (QL_SEG_PIXELS - 1)) & ~(uint64_t)(QL_SEG_PIXELS - 1); }
struct HashTable {
    uint4 *pos; uint64_t *act; uint16_t *tag;
    __device__ __forceinline__ HashTable(uint4 *table, uint64_t n)
        : pos(table), act(reinterpret_cast<uint64_t *>(table + hash_table_entries(n))), tag(reinterpret_cast<uint16_t *>(act + hash_table_entries(n))) {}
};
// the eight entries of lane `lane` of segment `seg` (indices seg * 512 + lane * 8 + 0..7)
__device__ __forceinline__ void hash_table_store8(uint4 *__restrict__ table, uint64_t n, uint64_t seg, uint32_t lane, const uint64_t (&h1)[QL_P], const uint64_t (&h2)[QL_P], const uint64_t (&ha)[QL_P])
{
    static_assert(QL_P == 8, "eight 16-bit tags are one 16-byte store");
    const HashTable t(table, n);
#pragma unroll
    for (int it = 0; it < QL_P; ++it) {
        const uint64_t slot = seg * QL_SEG_PIXELS + (uint32_t)it * 64u + lane;
        t.pos[slot] = make_uint4((uint32_t)h1[it], (uint32_t)(h1[it] >> 32), (uint32_t)h2[it], (uint32_t)(h2[it] >> 32));
        t.act[slot] = ha[it];
    }
    uint32_t w[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) w[k] = (uint32_t)(ha[2 * k] >> 48) | ((uint32_t)(ha[2 * k + 1] >> 48) << 16);
    *reinterpret_cast<uint4 *>(t.tag + seg * QL_SEG_PIXELS + lane * QL_P) = make_uint4(w[0], w[1], w[2], w[3]);
}
// RN(h) as a double from the two halves of h: both conversions and the product are exact, the sum is rounded once
__device__ __forceinline__ double rn_double(uint32_t lo, uint32_t hi) { return __builtin_fma((double)hi, 0x1p32, (double)lo); }

// ---- LDS-DMA of an image row -------------------------------------------------------------------------------------
// `words` dwords of `row` -> LDS at lds_byte_addr, 1 KiB (one 16-byte piece per lane) per wave and step, the row pointer in an SGPR
// pair (saddr addressing: the VGPR holds a 32-bit byte offset), bounds tested only on the row's last piece.  M0 is saved and
// restored inside the asm block (a reserved register: the compiler rejects it as a clobber).  Completion: dma_wait_all().
__device__ __forceinline__ void dma_row(uint32_t lds_byte_addr /* uniform */, const uint32_t *row /* uniform */, uint32_t words, uint32_t wave, uint32_t lane, uint32_t nwaves)
{
    const uint32_t npieces = words >> 2;                          // whole 16-byte pieces
    const uint32_t lane_off = lane << 4;
    for (uint32_t c = wave; (c << 6) < npieces; c += nwaves) {
        const uint32_t dst = __builtin_amdgcn_readfirstlane(lds_byte_addr + (c << 10));
        const uint32_t off = lane_off + (c << 10);
        if ((c << 6) + 64u <= npieces || (c << 6) + lane < npieces) {
            uint32_t keep;
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

}  // namespace rbf
