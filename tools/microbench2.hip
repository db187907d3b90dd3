// microbench2.hip -- VALU dependent-issue latency vs throughput on gfx950 (1 wave per SIMD).
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int CH, int OP>
__global__ __launch_bounds__(256) void k_chain(uint32_t *sink, uint32_t seed, int reps)
{
    uint32_t a[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) a[c] = seed + threadIdx.x * (c + 3);
#pragma unroll 1
    for (int r = 0; r < reps; ++r) {
#pragma unroll
        for (int u = 0; u < 16; ++u)
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                if (OP == 0) a[c] = a[c] * 0x9E3779B1u;                       // v_mul_lo_u32
                else if (OP == 1) a[c] = __umulhi(a[c], 0x9E3779B1u) + 0x1234567u;  // v_mul_hi + add
                else if (OP == 2) a[c] = (a[c] ^ 0x85EBCA77u) + 0x9E3779B1u;  // xor + add (2 simple ops)
                else if (OP == 3) a[c] = min(a[c], a[c] - 611158u);           // sub + min
            }
    }
    uint32_t x = 0;
#pragma unroll
    for (int c = 0; c < CH; ++c) x ^= a[c];
    if (x == 0x12345) sink[0] = x;
}

template <int CH, int OP>
static void run(const char *what, int ops_per, int blocks, uint32_t *sink)
{
    const int reps = 2000;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL((k_chain<CH, OP>), dim3(blocks), dim3(256), 0, 0, sink, 7u, reps);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL((k_chain<CH, OP>), dim3(blocks), dim3(256), 0, 0, sink, 7u, reps);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    // per wave: reps*16*CH*ops_per instructions
    const double instr = (double)reps * 16 * CH * ops_per;
    const double cyc = ms * 1e-3 * 2.4e9;
    const double waves_per_simd = blocks * 4.0 / 1024.0;
    printf("%-28s chains=%d waves/SIMD=%.0f: %.2f cycles per instr per wave, %.2f cycles per instr per SIMD\n", what, CH, waves_per_simd,
           cyc / instr, cyc / (instr * waves_per_simd));
}

int main()
{
    uint32_t *sink; CK(hipMalloc(&sink, 64));
    run<1, 0>("v_mul_lo_u32", 1, 256, sink); run<2, 0>("v_mul_lo_u32", 1, 256, sink); run<4, 0>("v_mul_lo_u32", 1, 256, sink); run<8, 0>("v_mul_lo_u32", 1, 256, sink);
    run<1, 0>("v_mul_lo_u32", 1, 1024, sink); run<4, 0>("v_mul_lo_u32", 1, 1024, sink); run<4, 0>("v_mul_lo_u32", 1, 2048, sink);
    run<1, 1>("v_mul_hi_u32+add", 2, 256, sink); run<4, 1>("v_mul_hi_u32+add", 2, 256, sink); run<4, 1>("v_mul_hi_u32+add", 2, 1024, sink);
    run<1, 2>("xor+add", 2, 256, sink); run<4, 2>("xor+add", 2, 256, sink); run<8, 2>("xor+add", 2, 256, sink); run<4, 2>("xor+add", 2, 1024, sink); run<8, 2>("xor+add", 2, 2048, sink);
    run<1, 3>("sub+min", 2, 256, sink); run<4, 3>("sub+min", 2, 1024, sink);
    return 0;
}
