// Noise statistics for the adaptive residual threshold
// (VideoFrameCompressor._estimate_noise_level, improved_video_compressor.py:727-744):
//     smoothed = cv2.medianBlur(luma, 5)        5x5 median, border pixels replicated
//     noise    = luma.astype(f32) - smoothed.astype(f32)
// The kernel produces the noise plane (optional, exact: every value is an integer below 2^17)
// and its EXACT integer moments  sum(noise), sum(noise^2); the float32 standard deviation the
// reference takes of that plane is host work (see engine.py), because it is a rounding-order
// property of numpy's pairwise summation, not of the data.
#pragma once
#include "rbf_device.h"

namespace rbf {

constexpr int NZ_TILE_W = 64, NZ_TILE_H = 8, NZ_THREADS = 256;
constexpr int NZ_LDS_W = NZ_TILE_W + 4, NZ_LDS_H = NZ_TILE_H + 4;

typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));

// Batcher's odd-even merge sort on 32 wires, restricted to the 25 wires that carry data (the other
// seven would hold +inf and never move).  Only wire 12 -- the median -- is consumed, so the compiler
// drops every min/max that cannot reach it (202 packed min/max remain out of 280).
struct MedianNet {
    int n;
    unsigned char a[192], b[192];
};
constexpr MedianNet make_median_net()
{
    MedianNet net{};
    for (int p = 1; p < 32; p <<= 1)
        for (int k = p; k >= 1; k >>= 1)
            for (int j = k % p; j + k < 32; j += 2 * k)
                for (int i = 0; i < (k < 32 - j - k ? k : 32 - j - k); ++i)
                    if ((i + j) / (2 * p) == (i + j + k) / (2 * p) && i + j + k < 25) {
                        net.a[net.n] = (unsigned char)(i + j);
                        net.b[net.n] = (unsigned char)(i + j + k);
                        ++net.n;
                    }
    return net;
}

__device__ __forceinline__ u16x2 median25(u16x2 (&v)[25])
{
    constexpr MedianNet net = make_median_net();
#pragma unroll
    for (int c = 0; c < net.n; ++c) {
        const u16x2 lo = __builtin_elementwise_min(v[net.a[c]], v[net.b[c]]);
        const u16x2 hi = __builtin_elementwise_max(v[net.a[c]], v[net.b[c]]);
        v[net.a[c]] = lo;
        v[net.b[c]] = hi;
    }
    return v[12];
}

// grid (ceil(W/64), ceil(H/8), nframes); a thread produces two vertically adjacent outputs, packed
// into the halves of one register so each network step is one v_pk_min_u16 / v_pk_max_u16.
template <typename SAMPLE>
__global__ __launch_bounds__(NZ_THREADS) void k_noise_moments(
    const uint8_t *__restrict__ frames, uint64_t frame_stride, uint32_t width, uint32_t height,
    uint64_t row_pitch, uint32_t pixel_stride,
    unsigned long long *__restrict__ moments /* [nframes][2]: sum d (two's complement), sum d^2 */,
    float *__restrict__ noise /* nullable, [nframes][height*width] */)
{
    __shared__ uint16_t tile[NZ_LDS_H][NZ_LDS_W];
    const uint32_t f = blockIdx.z;
    const uint8_t *src = frames + (uint64_t)f * frame_stride;
    const int x0 = (int)blockIdx.x * NZ_TILE_W, y0 = (int)blockIdx.y * NZ_TILE_H;
    for (int i = threadIdx.x; i < NZ_LDS_W * NZ_LDS_H; i += NZ_THREADS) {
        const int ly = i / NZ_LDS_W, lx = i - ly * NZ_LDS_W;
        int gx = x0 + lx - 2, gy = y0 + ly - 2;
        gx = gx < 0 ? 0 : (gx >= (int)width ? (int)width - 1 : gx);        // BORDER_REPLICATE
        gy = gy < 0 ? 0 : (gy >= (int)height ? (int)height - 1 : gy);
        tile[ly][lx] = *(const SAMPLE *)(src + (uint64_t)gy * row_pitch + (uint64_t)gx * pixel_stride);
    }
    __syncthreads();
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    u16x2 v[25];
#pragma unroll
    for (int r = 0; r < 5; ++r)
#pragma unroll
        for (int c = 0; c < 5; ++c) {
            v[r * 5 + c].x = tile[2 * ty + r][tx + c];
            v[r * 5 + c].y = tile[2 * ty + r + 1][tx + c];
        }
    const int ca = tile[2 * ty + 2][tx + 2], cb = tile[2 * ty + 3][tx + 2];
    const u16x2 med = median25(v);
    const int gx = x0 + tx, gy = y0 + 2 * ty;
    const bool in_a = gx < (int)width && gy < (int)height, in_b = gx < (int)width && gy + 1 < (int)height;
    const int da = in_a ? ca - (int)med.x : 0, db = in_b ? cb - (int)med.y : 0;
    if (noise) {
        float *out = noise + (uint64_t)f * width * height + (uint64_t)gy * width + gx;
        if (in_a) out[0] = (float)da;
        if (in_b) out[width] = (float)db;
    }
    long long s1 = (long long)da + db;
    unsigned long long s2 = (unsigned long long)((long long)da * da) + (unsigned long long)((long long)db * db);
#pragma unroll
    for (int dlt = 32; dlt >= 1; dlt >>= 1) {
        s1 += __shfl_down(s1, dlt);
        s2 += __shfl_down(s2, dlt);
    }
    if ((threadIdx.x & 63) == 0) {
        if (s1) atomicAdd(&moments[2 * f], (unsigned long long)s1);
        if (s2) atomicAdd(&moments[2 * f + 1], s2);
    }
}

// ------------------------------------------------------------------------------------------
// A1, BGR input: luma = cv2.cvtColor(frame, cv2.COLOR_BGR2GRAY) (improved_video_compressor.py:794-795)
// ------------------------------------------------------------------------------------------
// OpenCV 4.x integer path for 8- and 16-bit samples (imgproc color_rgb: RGB2Gray<uchar> / <ushort>):
//     gray = (B*3735 + G*19235 + R*9798 + (1 << 14)) >> 15        coefficients sum to 1 << 15
// One lane per pixel, any channel count >= 3 (a 4th channel is ignored, as OpenCV does for BGRA).
template <typename SAMPLE>
__global__ __launch_bounds__(256) void k_bgr_to_gray(
    const uint8_t *__restrict__ frames, uint64_t frame_stride, uint32_t width, uint64_t n, uint64_t row_pitch, uint32_t pixel_stride,
    SAMPLE *__restrict__ gray /* [nframes][n] */)
{
    const uint8_t *src = frames + (uint64_t)blockIdx.y * frame_stride;
    SAMPLE *dst = gray + (uint64_t)blockIdx.y * n;
    const bool flat = row_pitch == (uint64_t)width * pixel_stride;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        uint64_t off;
        if (flat) off = i * pixel_stride;
        else { const uint64_t y = i / width; off = y * row_pitch + (i - y * width) * pixel_stride; }
        const SAMPLE *px = (const SAMPLE *)(src + off);
        const uint32_t v = (uint32_t)px[0] * 3735u + (uint32_t)px[1] * 19235u + (uint32_t)px[2] * 9798u + (1u << 14);
        dst[i] = (SAMPLE)(v >> 15);
    }
}

// Sample 0 of every pixel (the luma of a YUV frame, improved_video_compressor.py:788-791 / the `y_plane` of
// fixed_video_compressor.py:292-296) into a dense plane: what the planar mask kernel reads.  One lane per pixel.
template <typename SAMPLE>
__global__ __launch_bounds__(256) void k_extract_luma(
    const uint8_t *__restrict__ frames, uint64_t frame_stride, uint32_t width, uint64_t n, uint64_t row_pitch, uint32_t pixel_stride,
    SAMPLE *__restrict__ luma /* [nframes][n] */)
{
    const uint8_t *src = frames + (uint64_t)blockIdx.y * frame_stride;
    SAMPLE *dst = luma + (uint64_t)blockIdx.y * n;
    const bool flat = row_pitch == (uint64_t)width * pixel_stride;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        uint64_t off;
        if (flat) off = i * pixel_stride;
        else { const uint64_t y = i / width; off = y * row_pitch + (i - y * width) * pixel_stride; }
        dst[i] = *(const SAMPLE *)(src + off);
    }
}

}  // namespace rbf
