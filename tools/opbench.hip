// opbench.hip -- per-opcode VALU / LDS issue cost on gfx950, measured with the shader clock (s_memtime) inside the
// kernel.  Every opcode is emitted through inline asm (the compiler cannot fold, fuse or reorder the stream), with
// wave-uniform operands in SGPRs (not literals), over CH independent dependency chains, at 1 / 2 / 4 waves per SIMD
// (a 256-thread workgroup puts one wave on each SIMD; dynamic LDS caps the workgroups per CU at the wanted count).
//
//   cycles per wave-instruction per SIMD = (t_end - t_start) / (instructions per wave * waves per SIMD)
//
// Build: hipcc --offload-arch=gfx950 -O3 -o opbench tools/opbench.hip ; run: ./opbench > profiles/r02_opbench.txt
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

enum Op {
    ADD_U32, XOR_B32, MIN_U32, SUB_U32, LSHLREV_B32, LSHRREV_B32, AND_B32,
    LSHL_ADD_U32, ADD3_U32, ALIGNBIT, BFE_U32, AND_OR_B32, PERM_B32, BITOP3,
    MUL_LO_U32, MUL_HI_U32, MUL_U32_U24, MAD_U32_U24, MUL_HI_U32_U24, MAD_U64_U32,
    FMA_F32, FMA_F64, MUL_F64, ADD_F64, CVT_F64_U32, CVT_U32_F64, LSHLREV_B64, LSHRREV_B64,
    CMP_LT_U32, CMP_LT_U64, CNDMASK, ADD_CO_ADDC, MOV_DPP, READLANE, PK_ADD_U16, PK_MIN_U16,
    MAD_U32_U16, MOV_B32, NOP_SALU,
    ADD_VV, ADD_VK, MIN_VV, AND_VV, XOR_VV, LSHLREV_VV, SUBREV_VS, ADD3_VVV, BITOP3_VVV, MAD24_VVV, FMA64_VVV, LSHL_ADD_VVV, ALIGNBIT_VVV,
    CNDMASK_SGPR, CMP_SGPR_DST, ADD_LIT, BFE_I32, MIN_VS_E64,
    NOPS
};

static const char *op_name[] = {
    "v_add_u32", "v_xor_b32", "v_min_u32", "v_sub_u32", "v_lshlrev_b32", "v_lshrrev_b32", "v_and_b32",
    "v_lshl_add_u32", "v_add3_u32", "v_alignbit_b32", "v_bfe_u32", "v_and_or_b32", "v_perm_b32", "v_bitop3_b32",
    "v_mul_lo_u32", "v_mul_hi_u32", "v_mul_u32_u24", "v_mad_u32_u24", "v_mul_hi_u32_u24", "v_mad_u64_u32",
    "v_fma_f32", "v_fma_f64", "v_mul_f64", "v_add_f64", "v_cvt_f64_u32", "v_cvt_u32_f64", "v_lshlrev_b64", "v_lshrrev_b64",
    "v_cmp_lt_u32 (vcc)", "v_cmp_lt_u64 (vcc)", "v_cndmask_b32", "v_add_co+v_addc_co (pair)", "v_mov_b32 dpp row_shr:1", "v_readlane_b32", "v_pk_add_u16", "v_pk_min_u16",
    "v_mad_u32_u16", "v_mov_b32", "s_add_u32 (SALU)",
    "v_add_u32 v,v,v", "v_add_u32 v,7,v", "v_min_u32 v,v,v", "v_and_b32 v,v,v", "v_xor_b32 v,v,v", "v_lshlrev_b32 v,v,v", "v_subrev_u32 v,s,v", "v_add3_u32 v,v,v,v",
    "v_bitop3_b32 v,v,v,v", "v_mad_u32_u24 v,v,v,v", "v_fma_f64 v,v,v,v", "v_lshl_add_u32 v,v,2,v", "v_alignbit_b32 v,v,v,v",
    "v_cndmask_b32 v,v,v,s[]", "v_cmp_lt_u32 s[],v,v", "v_add_u32 v,literal,v", "v_bfe_i32 v,v,0,24", "v_min_u32_e64 v,v,s",
};

template <int OP>
__device__ __forceinline__ void one(uint32_t &a, uint32_t &b, uint64_t &d, uint32_t s0, uint32_t s1, uint32_t &c, uint64_t &e, uint64_t scond)
{
    // a, b: 32-bit chain registers; d: 64-bit chain register; s0, s1: SGPR operands
    if (OP == ADD_U32) asm volatile("v_add_u32 %0, %1, %0" : "+v"(a) : "s"(s0));
    else if (OP == XOR_B32) asm volatile("v_xor_b32 %0, %1, %0" : "+v"(a) : "s"(s0));
    else if (OP == MIN_U32) asm volatile("v_min_u32 %0, %1, %0" : "+v"(a) : "s"(s0));
    else if (OP == SUB_U32) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(a) : "v"(b));
    else if (OP == LSHLREV_B32) asm volatile("v_lshlrev_b32 %0, %1, %0" : "+v"(a) : "s"(s0));
    else if (OP == LSHRREV_B32) asm volatile("v_lshrrev_b32 %0, %1, %0" : "+v"(a) : "s"(s0));
    else if (OP == AND_B32) asm volatile("v_and_b32 %0, %1, %0" : "+v"(a) : "s"(s0));
    else if (OP == LSHL_ADD_U32) asm volatile("v_lshl_add_u32 %0, %0, 3, %1" : "+v"(a) : "v"(b));
    else if (OP == ADD3_U32) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "s"(s0));
    else if (OP == ALIGNBIT) asm volatile("v_alignbit_b32 %0, %0, %1, 31" : "+v"(a) : "v"(b));
    else if (OP == BFE_U32) asm volatile("v_bfe_u32 %0, %0, 5, 27" : "+v"(a));
    else if (OP == AND_OR_B32) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(a) : "s"(s0), "v"(b));
    else if (OP == PERM_B32) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "s"(s0));
    else if (OP == BITOP3) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(a) : "v"(b), "s"(s0));
    else if (OP == MUL_LO_U32) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a) : "s"(s0));
    else if (OP == MUL_HI_U32) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a) : "s"(s0));
    else if (OP == MUL_U32_U24) asm volatile("v_mul_u32_u24 %0, %1, %0" : "+v"(a) : "s"(s0));
    else if (OP == MAD_U32_U24) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(a) : "s"(s0), "v"(b));
    else if (OP == MUL_HI_U32_U24) asm volatile("v_mul_hi_u32_u24 %0, %1, %0" : "+v"(a) : "s"(s0));
    else if (OP == MAD_U64_U32) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(d) : "v"(a), "s"(s0) : "vcc");
    else if (OP == FMA_F32) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a) : "s"(s0), "v"(b));
    else if (OP == FMA_F64) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d) : "s"((uint64_t)s0 << 32 | s1), "v"(d));
    else if (OP == MUL_F64) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d) : "s"((uint64_t)s0 << 32 | s1));
    else if (OP == ADD_F64) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d) : "s"((uint64_t)s0 << 32 | s1));
    else if (OP == CVT_F64_U32) asm volatile("v_cvt_f64_u32 %0, %1" : "=v"(d) : "v"(a));
    else if (OP == CVT_U32_F64) asm volatile("v_cvt_u32_f64 %0, %1" : "=v"(a) : "v"(d));
    else if (OP == LSHLREV_B64) asm volatile("v_lshlrev_b64 %0, 1, %0" : "+v"(d));
    else if (OP == LSHRREV_B64) asm volatile("v_lshrrev_b64 %0, 1, %0" : "+v"(d));
    else if (OP == CMP_LT_U32) asm volatile("v_cmp_lt_u32 vcc, %0, %1" : : "v"(a), "v"(b) : "vcc");
    else if (OP == CMP_LT_U64) asm volatile("v_cmp_lt_u64 vcc, %0, %1" : : "v"(d), "s"((uint64_t)s0 << 32 | s1) : "vcc");
    else if (OP == CNDMASK) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a) : "v"(b) : "vcc");
    else if (OP == ADD_CO_ADDC) asm volatile("v_add_co_u32 %0, vcc, %0, %2\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc" : "+v"(a), "+v"(b) : "s"(s0) : "vcc");
    else if (OP == MOV_DPP) asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a) : "v"(b));
    else if (OP == READLANE) { uint32_t t; asm volatile("v_readlane_b32 %0, %1, 3" : "=s"(t) : "v"(a)); }
    else if (OP == PK_ADD_U16) asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(a) : "v"(b));
    else if (OP == PK_MIN_U16) asm volatile("v_pk_min_u16 %0, %0, %1" : "+v"(a) : "v"(b));
    else if (OP == MAD_U32_U16) asm volatile("v_mad_u32_u16 %0, %0, %1, %2" : "+v"(a) : "s"(s0), "v"(b));
    else if (OP == MOV_B32) asm volatile("v_mov_b32 %0, %1" : "=v"(a) : "v"(b));
    else if (OP == NOP_SALU) { uint32_t t = s0; asm volatile("s_add_u32 %0, %0, %1" : "+s"(t) : "s"(s1) : "scc"); }
    else if (OP == ADD_VV) asm volatile("v_add_u32 %0, %1, %0" : "+v"(a) : "v"(b));
    else if (OP == ADD_VK) asm volatile("v_add_u32 %0, 7, %0" : "+v"(a));
    else if (OP == MIN_VV) asm volatile("v_min_u32 %0, %1, %0" : "+v"(a) : "v"(b));
    else if (OP == AND_VV) asm volatile("v_and_b32 %0, %1, %0" : "+v"(a) : "v"(b));
    else if (OP == XOR_VV) asm volatile("v_xor_b32 %0, %1, %0" : "+v"(a) : "v"(b));
    else if (OP == LSHLREV_VV) asm volatile("v_lshlrev_b32 %0, %1, %0" : "+v"(a) : "v"(b));
    else if (OP == SUBREV_VS) asm volatile("v_subrev_u32 %0, %1, %0" : "+v"(a) : "s"(s0));
    else if (OP == ADD3_VVV) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));
    else if (OP == BITOP3_VVV) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(a) : "v"(b), "v"(c));
    else if (OP == MAD24_VVV) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));
    else if (OP == FMA64_VVV) asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(d) : "v"(e));
    else if (OP == LSHL_ADD_VVV) asm volatile("v_lshl_add_u32 %0, %0, 2, %1" : "+v"(a) : "v"(b));
    else if (OP == ALIGNBIT_VVV) asm volatile("v_alignbit_b32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));
    else if (OP == CNDMASK_SGPR) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a) : "v"(b), "s"(scond));
    else if (OP == CMP_SGPR_DST) { uint64_t t; asm volatile("v_cmp_lt_u32_e64 %0, %1, %2" : "=s"(t) : "v"(a), "v"(b)); }
    else if (OP == ADD_LIT) asm volatile("v_add_u32 %0, 0x12345, %0" : "+v"(a));
    else if (OP == BFE_I32) asm volatile("v_bfe_i32 %0, %0, 0, 24" : "+v"(a));
    else if (OP == MIN_VS_E64) asm volatile("v_min_u32_e64 %0, %0, %1" : "+v"(a) : "s"(s0));
}

constexpr int UNROLL = 8;

template <int OP, int CH>
__global__ __launch_bounds__(256) void k_op(uint64_t *out, uint32_t s0, uint32_t s1, int reps)
{
    extern __shared__ uint32_t lds_pad[];
    uint32_t a[CH], b[CH], cc[CH];
    uint64_t d[CH], e[CH];
    const uint64_t scond = __ballot((threadIdx.x & 1u) != 0);
#pragma unroll
    for (int c = 0; c < CH; ++c) { a[c] = threadIdx.x * 2654435761u + c; b[c] = threadIdx.x + 17 * c + 1; cc[c] = threadIdx.x * 3 + c; d[c] = 0x3FF0000000000000ull + threadIdx.x + c; e[c] = 0x3FF0000000000001ull + c; }
    __syncthreads();
    const uint64_t w0 = wall_clock64();
    const uint64_t t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int r = 0; r < reps; ++r) {
#pragma unroll
        for (int u = 0; u < UNROLL; ++u)
#pragma unroll
            for (int c = 0; c < CH; ++c) one<OP>(a[c], b[c], d[c], s0, s1, cc[c], e[c], scond);
    }
    const uint64_t t1 = __builtin_readcyclecounter();
    uint32_t x = 0;
#pragma unroll
    for (int c = 0; c < CH; ++c) x ^= a[c] ^ b[c] ^ cc[c] ^ (uint32_t)e[c] ^ (uint32_t)d[c] ^ (uint32_t)(d[c] >> 32);
    if (x == 0x12345u && reps < 0) lds_pad[threadIdx.x] = x;
    const uint64_t w1 = wall_clock64();
    if ((threadIdx.x & 63u) == 0) out[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
    if (blockIdx.x == 0 && threadIdx.x == 0) { out[gridDim.x * 4] = t1 - t0; out[gridDim.x * 4 + 1] = w1 - w0; }
}

static uint64_t *g_out;
static int g_cus = 256;

template <int OP, int CH>
static double run(int waves_per_simd)
{
    const int reps = 2000;
    const int blocks = g_cus * waves_per_simd;
    const size_t lds = (160 * 1024 / waves_per_simd) & ~1023;   // at most `waves_per_simd` workgroups fit a CU
    auto kern = k_op<OP, CH>;
    CK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), lds, 0, g_out, 5u, 0x3ff00001u, reps);
    CK(hipDeviceSynchronize());
    std::vector<uint64_t> h(blocks * 4);
    CK(hipMemcpy(h.data(), g_out, h.size() * 8, hipMemcpyDeviceToHost));
    std::sort(h.begin(), h.end());
    const double med = (double)h[h.size() / 2];
    if (OP == ADD_U32 && CH == 8 && waves_per_simd == 1) {
        uint64_t cal[2]; CK(hipMemcpy(cal, g_out + (size_t)blocks * 4, 16, hipMemcpyDeviceToHost));
        printf("# clock calibration: %llu s_memtime ticks in %llu wall_clock64 ticks (100 MHz) -> s_memtime runs at %.1f MHz\n",
               (unsigned long long)cal[0], (unsigned long long)cal[1], (double)cal[0] / (double)cal[1] * 100.0);
    }
    const double instr = (double)reps * UNROLL * CH * (OP == ADD_CO_ADDC ? 2 : 1);
    return med / (instr * waves_per_simd);
}

template <int OP>
static void bench()
{
    // independent chains hide the dependent-issue latency; waves per SIMD hide the rest
    printf("%-28s | ch=1 w=1 %6.2f | ch=4 w=1 %6.2f | ch=8 w=1 %6.2f | ch=4 w=2 %6.2f | ch=8 w=2 %6.2f | ch=4 w=4 %6.2f | ch=8 w=4 %6.2f\n", op_name[OP],
           run<OP, 1>(1), run<OP, 4>(1), run<OP, 8>(1), run<OP, 4>(2), run<OP, 8>(2), run<OP, 4>(4), run<OP, 8>(4));
    fflush(stdout);
}

// ---- LDS: random-address reads, the query kernel's probe pattern ------------------------------------------------
template <int KIND>
__global__ __launch_bounds__(1024) void k_lds(uint64_t *out, uint32_t words, int reps)
{
    extern __shared__ uint32_t lds[];
    for (uint32_t i = threadIdx.x; i < words; i += blockDim.x) lds[i] = i * 2654435761u;
    __syncthreads();
    uint32_t x = threadIdx.x * 2654435761u + blockIdx.x, acc = 0;
    const uint64_t t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int r = 0; r < reps; ++r) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            x = x * 1664525u + 1013904223u;                       // 2 VALU of address generation per probe
            const uint32_t w = (x >> 8) % words;                  // (compiler: mul_hi based; all variants pay the same)
            if (KIND == 0) acc += lds[w];                                             // ds_read_b32, random dword
            else if (KIND == 1) acc += reinterpret_cast<const uint8_t *>(lds)[w * 4 + (x & 3u)];   // ds_read_u8, random byte
            else if (KIND == 2) acc += (uint32_t)reinterpret_cast<const uint64_t *>(lds)[w >> 1];   // ds_read_b64, random qword
            else if (KIND == 3) acc += lds[(w & ~63u) + (threadIdx.x & 63u)];           // ds_read_b32, conflict-free
            else acc += w;                                                             // address generation only
        }
    }
    const uint64_t t1 = __builtin_readcyclecounter();
    if (acc == 0x12345u && reps < 0) lds[0] = acc;
    if ((threadIdx.x & 63u) == 0) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int KIND>
static double run_lds(const char *what)
{
    const int reps = 500;
    const uint32_t words = 19100;                                 // a 1080p filter
    auto kern = k_lds<KIND>;
    CK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(kern, dim3(g_cus), dim3(1024), 150 * 1024, 0, g_out, words, reps);
    CK(hipDeviceSynchronize());
    std::vector<uint64_t> h(g_cus * 16);
    CK(hipMemcpy(h.data(), g_out, h.size() * 8, hipMemcpyDeviceToHost));
    std::sort(h.begin(), h.end());
    const double cyc = (double)h[h.size() / 2] / (reps * 8.0);    // cycles per probe-step of one wave, 4 waves per SIMD in flight
    printf("%-44s %7.2f cycles per wave-probe (16 waves/CU) -> %6.2f cycles per probe per CU\n", what, cyc, cyc / 16.0);
    return cyc;
}

int main()
{
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    g_cus = prop.multiProcessorCount;
    printf("# %s, %d CUs, clockRate %d kHz; cycles = shader clock (s_memtime) per wave-instruction per SIMD\n", prop.gcnArchName, g_cus, prop.clockRate);
    CK(hipMalloc(&g_out, (size_t)g_cus * 8 * 16 * 8 + 64));
    bench<ADD_U32>(); bench<XOR_B32>(); bench<MIN_U32>(); bench<SUB_U32>(); bench<LSHLREV_B32>(); bench<LSHRREV_B32>(); bench<AND_B32>();
    bench<LSHL_ADD_U32>(); bench<ADD3_U32>(); bench<ALIGNBIT>(); bench<BFE_U32>(); bench<AND_OR_B32>(); bench<PERM_B32>(); bench<BITOP3>();
    bench<MUL_LO_U32>(); bench<MUL_HI_U32>(); bench<MUL_U32_U24>(); bench<MAD_U32_U24>(); bench<MUL_HI_U32_U24>(); bench<MAD_U64_U32>();
    bench<FMA_F32>(); bench<FMA_F64>(); bench<MUL_F64>(); bench<ADD_F64>(); bench<CVT_F64_U32>(); bench<CVT_U32_F64>(); bench<LSHLREV_B64>(); bench<LSHRREV_B64>();
    bench<CMP_LT_U32>(); bench<CMP_LT_U64>(); bench<CNDMASK>(); bench<ADD_CO_ADDC>(); bench<MOV_DPP>(); bench<READLANE>(); bench<PK_ADD_U16>(); bench<PK_MIN_U16>();
    bench<MAD_U32_U16>(); bench<MOV_B32>(); bench<NOP_SALU>();
    printf("# operand kinds: the same opcodes with VGPR-only / inline-constant / literal / SGPR operands\n");
    bench<ADD_VV>(); bench<ADD_VK>(); bench<ADD_LIT>(); bench<MIN_VV>(); bench<MIN_VS_E64>(); bench<AND_VV>(); bench<XOR_VV>(); bench<LSHLREV_VV>(); bench<SUBREV_VS>();
    bench<ADD3_VVV>(); bench<BITOP3_VVV>(); bench<MAD24_VVV>(); bench<FMA64_VVV>(); bench<LSHL_ADD_VVV>(); bench<ALIGNBIT_VVV>(); bench<BFE_I32>();
    bench<CNDMASK_SGPR>(); bench<CMP_SGPR_DST>();
    printf("# LDS probes, 1024-thread workgroup per CU, 76 KB table\n");
    const double base = run_lds<4>("address generation only");
    const double r32 = run_lds<0>("ds_read_b32 random dword");
    const double r8 = run_lds<1>("ds_read_u8 random byte");
    const double r64 = run_lds<2>("ds_read_b64 random qword");
    const double rcf = run_lds<3>("ds_read_b32 conflict-free");
    printf("# net of address generation: b32 %.2f, u8 %.2f, b64 %.2f, conflict-free b32 %.2f cycles per wave-probe\n", r32 - base, r8 - base, r64 - base, rcf - base);
    return 0;
}
