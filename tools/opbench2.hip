// opbench2.hip -- VALU issue cost per opcode on gfx950, measured as THROUGHPUT of a whole-chip launch (HIP events around the
// launch, instruction count known), at 1 / 2 / 4 / 8 waves per SIMD BY CONSTRUCTION: one workgroup of 256 / 512 / 1024 threads per
// CU (a workgroup's waves are dealt round-robin over the CU's four SIMDs: checked below from HW_ID), 8 = two 1024-thread
// workgroups per CU (checked from HW_ID + wall clock: both resident at the same time).  Round 2's opbench.hip relied on dynamic LDS
// to cap the workgroups per CU and divided a per-wave median by the wanted wave count; VERDICT r03 (weak 2) found its rows
// inconsistent with the "four cycles per VALU wave-instruction" model of DESIGN.md.  This one cannot mis-count residency:
//
//   cycles per wave-instruction per SIMD = launch time x shader clock x (CUs x 4) / (wave-instructions of the launch)
//
// with the shader clock taken from s_memtime against the 100 MHz wall clock inside the same launch.  Every opcode goes through inline
// asm (nothing folds), 8 independent chains per wave.  Mixed rows replay the instruction mix of k_query_s64's reductions and steps.
//
// Build: hipcc --offload-arch=gfx950 -O3 -o build/opbench2 tools/opbench2.hip ; run: ./build/opbench2 > profiles/r04_opbench2.txt
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <map>
#include <tuple>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

enum Op {
    ADD_VV, ADD_VS, ADD_VK, SUB_VV, XOR_VV, AND_VV, MIN_VV, MIN_VS, LSHR_VS, LSHR_VK, LSHL_VV, LSHL_ADD, LSHL_OR, ADD3, ALIGNBIT, BFE, AND_OR, PERM, BITOP3,
    MUL_LO, MUL_HI, MAD24, MAD_U64, FMA_F32, PK_FMA_F32, FMA_F64, ADD_F64, CVT_F64_U32, CMP_VCC, CMP_SGPR, CMP_SDWA_SGPR, CNDMASK_SGPR, CNDMASK_VCC,
    MOV, MOV_DPP, READLANE, READFIRSTLANE, MBCNT, SALU_ADD,
    MIX_REDUCE,        // v_fma_f64, v_mad_u64_u32, v_add_u32, v_min_u32: one FP64 reduction of k_query_s64 (rows_reduce4)
    MIX_STEP,          // v_lshrrev, v_add, v_lshl_add, v_sub, v_min: one probe step of frame_pass_rows
    MIX_COMBINE,       // v_and, v_lshlrev, v_or (what the compiler makes of  f = (w << (p & 31)) | f ), v_alignbit every third
    MIX_VALU_SALU,     // v_add_u32 + s_add_u32 alternating: does scalar work take VALU issue slots?
    MIX_F64_INT,       // v_fma_f64 + v_add_u32 alternating: do the FP64 and the integer pipe overlap between waves?
    NOPS
};
static const char *op_name[] = {
    "v_add_u32 v,v,v", "v_add_u32 v,s,v", "v_add_u32 v,7,v", "v_sub_u32 v,v,v", "v_xor_b32 v,v,v", "v_and_b32 v,v,v", "v_min_u32 v,v,v", "v_min_u32 v,s,v",
    "v_lshrrev_b32 v,s,v", "v_lshrrev_b32 v,5,v", "v_lshlrev_b32 v,v,v", "v_lshl_add_u32 v,v,2,v", "v_lshl_or_b32 v,v,v,v", "v_add3_u32 v,v,v,v", "v_alignbit_b32 v,v,v,31",
    "v_bfe_u32 v,v,5,27", "v_and_or_b32 v,v,s,v", "v_perm_b32 v,v,v,s", "v_bitop3_b32 v,v,v,v",
    "v_mul_lo_u32 v,v,s", "v_mul_hi_u32 v,v,s", "v_mad_u32_u24 v,v,v,v", "v_mad_u64_u32 v[2],vcc,v,s,v[2]", "v_fma_f32 v,v,s,v", "v_pk_fma_f32 v[2],v[2],v[2],v[2]",
    "v_fma_f64 v,v,s,v", "v_add_f64 v,v,s", "v_cvt_f64_u32", "v_cmp_lt_u32 vcc,v,v", "v_cmp_lt_u32 s[2],v,v", "v_cmp_le_u32_sdwa s[2],v.b0,v", "v_cndmask_b32 v,v,v,s[2]", "v_cndmask_b32 v,v,v,vcc",
    "v_mov_b32 v,v", "v_mov_b32_dpp row_shr:1", "v_readlane_b32 s,v,3", "v_readfirstlane_b32 s,v", "v_mbcnt_lo_u32_b32 v,s,v", "s_add_u32 (SALU only)",
    "MIX reduce: fma_f64, mad_u64_u32, add, min", "MIX step: lshr, add, lshl_add, sub, min", "MIX combine: and, lshl, or (+alignbit)",
    "MIX v_add_u32 + s_add_u32 (count: both)", "MIX v_fma_f64 + v_add_u32",
};
// wave-instructions emitted by one call of one<OP>()
static int op_instr(int op)
{
    switch (op) { case MIX_REDUCE: return 4; case MIX_STEP: return 5; case MIX_COMBINE: return 3; case MIX_VALU_SALU: return 2; case MIX_F64_INT: return 2; default: return 1; }
}

template <int OP>
__device__ __forceinline__ void one(uint32_t &a, uint32_t &b, uint32_t &c, uint64_t &d, uint32_t s0, uint32_t s1, uint64_t s64, uint32_t &st)
{
    if (OP == ADD_VV) asm volatile("v_add_u32 %0, %1, %0" : "+v"(a) : "v"(b));
    else if (OP == ADD_VS) asm volatile("v_add_u32 %0, %1, %0" : "+v"(a) : "s"(s0));
    else if (OP == ADD_VK) asm volatile("v_add_u32 %0, 7, %0" : "+v"(a));
    else if (OP == SUB_VV) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(a) : "v"(b));
    else if (OP == XOR_VV) asm volatile("v_xor_b32 %0, %1, %0" : "+v"(a) : "v"(b));
    else if (OP == AND_VV) asm volatile("v_and_b32 %0, %1, %0" : "+v"(a) : "v"(b));
    else if (OP == MIN_VV) asm volatile("v_min_u32 %0, %1, %0" : "+v"(a) : "v"(b));
    else if (OP == MIN_VS) asm volatile("v_min_u32 %0, %1, %0" : "+v"(a) : "s"(s0));
    else if (OP == LSHR_VS) asm volatile("v_lshrrev_b32 %0, %1, %0" : "+v"(a) : "s"(s0));
    else if (OP == LSHR_VK) asm volatile("v_lshrrev_b32 %0, 5, %0" : "+v"(a));
    else if (OP == LSHL_VV) asm volatile("v_lshlrev_b32 %0, %1, %0" : "+v"(a) : "v"(b));
    else if (OP == LSHL_ADD) asm volatile("v_lshl_add_u32 %0, %0, 2, %1" : "+v"(a) : "v"(b));
    else if (OP == LSHL_OR) asm volatile("v_lshl_or_b32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));
    else if (OP == ADD3) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));
    else if (OP == ALIGNBIT) asm volatile("v_alignbit_b32 %0, %0, %1, 31" : "+v"(a) : "v"(b));
    else if (OP == BFE) asm volatile("v_bfe_u32 %0, %0, 5, 27" : "+v"(a));
    else if (OP == AND_OR) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(a) : "s"(s0), "v"(b));
    else if (OP == PERM) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "s"(s0));
    else if (OP == BITOP3) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(a) : "v"(b), "v"(c));
    else if (OP == MUL_LO) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a) : "s"(s0));
    else if (OP == MUL_HI) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a) : "s"(s0));
    else if (OP == MAD24) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));
    else if (OP == MAD_U64) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(d) : "v"(a), "s"(s0) : "vcc");
    else if (OP == FMA_F32) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a) : "s"(s0), "v"(b));
    else if (OP == PK_FMA_F32) asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(d));
    else if (OP == FMA_F64) asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(d) : "s"(s64));
    else if (OP == ADD_F64) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d) : "s"(s64));
    else if (OP == CVT_F64_U32) asm volatile("v_cvt_f64_u32 %0, %1" : "=v"(d) : "v"(a));
    else if (OP == CMP_VCC) asm volatile("v_cmp_lt_u32 vcc, %0, %1" : : "v"(a), "v"(b) : "vcc");
    else if (OP == CMP_SGPR) { uint64_t t; asm volatile("v_cmp_lt_u32_e64 %0, %1, %2" : "=s"(t) : "v"(a), "v"(b)); }
    else if (OP == CMP_SDWA_SGPR) { uint64_t t; asm volatile("v_cmp_le_u32_sdwa %0, %1, %2 src0_sel:BYTE_0 src1_sel:DWORD" : "=s"(t) : "v"(a), "v"(b)); }
    else if (OP == CNDMASK_SGPR) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a) : "v"(b), "s"(s64));
    else if (OP == CNDMASK_VCC) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a) : "v"(b) : "vcc");
    else if (OP == MOV) asm volatile("v_mov_b32 %0, %1" : "=v"(a) : "v"(b));
    else if (OP == MOV_DPP) asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a) : "v"(b));
    else if (OP == READLANE) { uint32_t t; asm volatile("v_readlane_b32 %0, %1, 3" : "=s"(t) : "v"(a)); }
    else if (OP == READFIRSTLANE) { uint32_t t; asm volatile("v_readfirstlane_b32 %0, %1" : "=s"(t) : "v"(a)); }
    else if (OP == MBCNT) asm volatile("v_mbcnt_lo_u32_b32 %0, %1, %0" : "+v"(a) : "s"(s0));
    else if (OP == SALU_ADD) asm volatile("s_add_u32 %0, %0, %1" : "+s"(st) : "s"(s1) : "scc");
    else if (OP == MIX_REDUCE) {
        // t = fma(hd, ninv, magic); s = lo(t) * m + hl (v_mad_u64_u32); q = s + m; x = min(s, q)      [hd = d, kept; result chains through a]
        uint64_t t, u;
        asm volatile("v_fma_f64 %0, %1, %2, %1" : "=v"(t) : "v"(d), "s"(s64));
        asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %3" : "=v"(u) : "v"((uint32_t)t), "v"(b), "v"(d) : "vcc");
        asm volatile("v_add_u32 %0, %1, %2" : "=v"(c) : "v"((uint32_t)u), "v"(b));
        asm volatile("v_min_u32 %0, %1, %2" : "=v"(a) : "v"((uint32_t)u), "v"(c));
    } else if (OP == MIX_STEP) {
        uint32_t w, u, ad, v;
        asm volatile("v_lshrrev_b32 %0, %1, %2" : "=v"(w) : "s"(s0), "v"(a));
        asm volatile("v_add_u32 %0, %1, %2" : "=v"(u) : "v"(a), "v"(b));
        asm volatile("v_lshl_add_u32 %0, %1, 2, %2" : "=v"(ad) : "v"(w), "v"(c));
        asm volatile("v_sub_u32 %0, %1, %2" : "=v"(v) : "v"(u), "v"(b));
        asm volatile("v_min_u32 %0, %1, %2" : "=v"(a) : "v"(u), "v"(v));
        asm volatile("" :: "v"(ad));
    } else if (OP == MIX_COMBINE) {
        uint32_t sh, t;
        asm volatile("v_and_b32 %0, 31, %1" : "=v"(sh) : "v"(b));
        asm volatile("v_lshlrev_b32 %0, %1, %2" : "=v"(t) : "v"(sh), "v"(c));
        asm volatile("v_or_b32 %0, %1, %0" : "+v"(a) : "v"(t));
    } else if (OP == MIX_VALU_SALU) {
        asm volatile("v_add_u32 %0, %1, %0" : "+v"(a) : "v"(b));
        asm volatile("s_add_u32 %0, %0, %1" : "+s"(st) : "s"(s1) : "scc");
    } else if (OP == MIX_F64_INT) {
        asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(d) : "s"(s64));
        asm volatile("v_add_u32 %0, %1, %0" : "+v"(a) : "v"(b));
    }
}

constexpr int CH = 8, UNROLL = 8;
struct Rec { uint64_t mem0, mem1, wall0, wall1; uint32_t hwid, xcc; };

template <int OP>
__global__ __launch_bounds__(1024) void k_op(Rec *rec, uint32_t s0, uint32_t s1, uint64_t s64, int reps)
{
    uint32_t a[CH], b[CH], c[CH], st = s0;
    uint64_t d[CH];
#pragma unroll
    for (int i = 0; i < CH; ++i) { a[i] = threadIdx.x * 2654435761u + i; b[i] = threadIdx.x + 17 * i + 1; c[i] = threadIdx.x * 3 + i; d[i] = 0x3FF0000000000000ull + threadIdx.x + i; }
    const uint64_t w0 = wall_clock64();
    const uint64_t t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int r = 0; r < reps; ++r) {
#pragma unroll
        for (int u = 0; u < UNROLL; ++u)
#pragma unroll
            for (int i = 0; i < CH; ++i) one<OP>(a[i], b[i], c[i], d[i], s0, s1, s64, st);
    }
    const uint64_t t1 = __builtin_readcyclecounter();
    const uint64_t w1 = wall_clock64();
    uint32_t x = st;
#pragma unroll
    for (int i = 0; i < CH; ++i) x ^= a[i] ^ b[i] ^ c[i] ^ (uint32_t)d[i] ^ (uint32_t)(d[i] >> 32);
    if ((threadIdx.x & 63u) == 0 || (x == 0x12345u && reps < 0)) {
        uint32_t hwid, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        rec[(size_t)blockIdx.x * 16 + (threadIdx.x >> 6)] = Rec{t0, t1, w0, w1, hwid, xcc};
    }
}

static Rec *g_rec;
static int g_cus = 256;
static double g_clock_mhz = 0;

struct Result { double cyc_tp, cyc_wave, resident; };

template <int OP>
static Result run(int waves_per_simd)
{
    const int threads = waves_per_simd >= 4 ? 1024 : waves_per_simd * 256;
    const int wgs_per_cu = waves_per_simd == 8 ? 2 : 1;
    const int blocks = g_cus * wgs_per_cu;
    const int reps = OP == SALU_ADD ? 400 : 400;
    auto kern = k_op<OP>;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const uint64_t s64 = 0x3ff0000000000001ull;
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, g_rec, 5u, 0x3ff00001u, s64, reps);     // warm-up
    CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int it = 0; it < 3; ++it) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, g_rec, 5u, 0x3ff00001u, s64, reps);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        best = std::min(best, ms);
    }
    const int waves_per_wg = threads / 64;
    std::vector<Rec> h((size_t)blocks * 16);
    CK(hipMemcpy(h.data(), g_rec, h.size() * sizeof(Rec), hipMemcpyDeviceToHost));
    // shader clock from the records; residency: the largest number of waves that ran on one (xcc, se, sh, cu, simd) at overlapping times
    double ticks = 0, walls = 0;
    std::vector<double> per_wave;
    std::map<std::tuple<uint32_t, uint32_t>, std::vector<std::pair<uint64_t, uint64_t>>> by_simd;
    for (int bk = 0; bk < blocks; ++bk)
        for (int w = 0; w < waves_per_wg; ++w) {
            const Rec &r = h[(size_t)bk * 16 + w];
            ticks += (double)(r.mem1 - r.mem0); walls += (double)(r.wall1 - r.wall0);
            per_wave.push_back((double)(r.mem1 - r.mem0));
            const uint32_t simd = (r.hwid >> 4) & 3u, cu = (r.hwid >> 8) & 15u, sh = (r.hwid >> 12) & 1u, se = (r.hwid >> 13) & 7u;
            by_simd[{r.xcc & 15u, (se << 8) | (sh << 7) | (cu << 2) | simd}].push_back({r.wall0, r.wall1});
        }
    if (g_clock_mhz == 0 && walls > 0) g_clock_mhz = ticks / walls * 100.0;
    double res_sum = 0; size_t res_n = 0;
    for (auto &kv : by_simd) {
        auto &v = kv.second;
        int mx = 0;
        for (size_t i = 0; i < v.size(); ++i) {
            const uint64_t mid = (v[i].first + v[i].second) / 2;
            int n = 0;
            for (size_t j = 0; j < v.size(); ++j) if (v[j].first <= mid && mid <= v[j].second) ++n;
            mx = std::max(mx, n);
        }
        res_sum += mx; ++res_n;
    }
    std::sort(per_wave.begin(), per_wave.end());
    const double instr_per_wave = (double)reps * UNROLL * CH * op_instr(OP);
    const double total = instr_per_wave * waves_per_wg * blocks;
    Result r;
    r.cyc_tp = (double)best * 1e-3 * g_clock_mhz * 1e6 * (g_cus * 4.0) / total;
    r.cyc_wave = per_wave[per_wave.size() / 2] / instr_per_wave;            // what ONE wave sees per instruction of its own
    r.resident = res_n ? res_sum / res_n : 0;
    CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
    return r;
}

template <int OP>
static void bench()
{
    printf("%-44s |", op_name[OP]);
    for (int w : {1, 2, 4, 8}) {
        const Result r = run<OP>(w);
        printf(" w=%d %5.2f (wave %6.2f, res %.1f) |", w, r.cyc_tp, r.cyc_wave, r.resident);
    }
    printf("\n");
    fflush(stdout);
}

int main()
{
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    g_cus = prop.multiProcessorCount;
    CK(hipMalloc(&g_rec, (size_t)g_cus * 2 * 16 * sizeof(Rec)));
    (void)run<ADD_VV>(1);
    printf("# %s, %d CUs, clockRate %d kHz; shader clock measured in-kernel (s_memtime vs 100 MHz wall clock): %.1f MHz\n", prop.gcnArchName, g_cus, prop.clockRate, g_clock_mhz);
    printf("# per row and waves/SIMD: cycles per wave-instruction per SIMD from the LAUNCH time (HIP events), (the same instruction as one wave sees it: its own\n"
           "# s_memtime span / its instructions; res = waves found resident together on one SIMD, from HW_ID + wall clock)\n");
    bench<ADD_VV>(); bench<ADD_VS>(); bench<ADD_VK>(); bench<SUB_VV>(); bench<XOR_VV>(); bench<AND_VV>(); bench<MIN_VV>(); bench<MIN_VS>();
    bench<LSHR_VS>(); bench<LSHR_VK>(); bench<LSHL_VV>(); bench<LSHL_ADD>(); bench<LSHL_OR>(); bench<ADD3>(); bench<ALIGNBIT>(); bench<BFE>(); bench<AND_OR>(); bench<PERM>(); bench<BITOP3>();
    bench<MUL_LO>(); bench<MUL_HI>(); bench<MAD24>(); bench<MAD_U64>(); bench<FMA_F32>(); bench<PK_FMA_F32>(); bench<FMA_F64>(); bench<ADD_F64>(); bench<CVT_F64_U32>();
    bench<CMP_VCC>(); bench<CMP_SGPR>(); bench<CMP_SDWA_SGPR>(); bench<CNDMASK_SGPR>(); bench<CNDMASK_VCC>();
    bench<MOV>(); bench<MOV_DPP>(); bench<READLANE>(); bench<READFIRSTLANE>(); bench<MBCNT>(); bench<SALU_ADD>();
    bench<MIX_REDUCE>(); bench<MIX_STEP>(); bench<MIX_COMBINE>(); bench<MIX_VALU_SALU>(); bench<MIX_F64_INT>();
    return 0;
}
