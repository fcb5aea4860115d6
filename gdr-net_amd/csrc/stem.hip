// ResNet stem for gfx950: 7x7 stride-2 pad-3 convolution 3 -> 64 channels on the zero-padded NHWC4 image canvas
// (resnet_backbone.py:23,69; canvas from gdrn_pack_image), bf16 operands, fp32 accumulate, with the per-tile partial sums
// for the following BatchNorm.
//
// As a 7-tap gather of 16 px x 4 ch on the generic kernel the layer multiplies 448 K-elements per output where 147 are
// real and stages everything through LDS (135 us, 110-130 TFLOP/s).  Here a kernel row is ONE k-step: for output
// (oy, ox) and tap row ky the 8 pixels x 4 channels at canvas[2oy+ky][2ox .. 2ox+7] are 64 contiguous bytes = the 32
// K-elements of a 16x16x32 MFMA (kx = 7 and channel 3 carry zero weights): K = 7 x 32 = 224.  A lane's pixel fragment is
// one 16-byte global load (neighbouring lanes overlap: all L1/L2 hits), the 28 weight fragments (64 co x 224) stay in
// registers for the wave's whole run of 16-pixel tiles, and nothing goes through LDS.
#include "common.h"
#include "../../include/gdrn_hip.h"

namespace {

constexpr int HP = 262, WP = 272;  // canvas rows / columns (gdrn_pack_image layout for a 256x256 image)

// fp32 OIHW [64][3][7][7] -> bf16 [64 co][7 ky][32 = 8 kx x 4 c], zeros for kx = 7 and c = 3
__global__ void pack_stem_w32_kernel(const float* __restrict__ w, bf16_t* __restrict__ dst) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 64 * 7 * 32) return;
    const int c = i & 3, kx = (i >> 2) & 7, ky = (i >> 5) % 7, co = i / 224;
    const float v = (c < 3 && kx < 7) ? w[((co * 3 + c) * 7 + ky) * 7 + kx] : 0.f;
    dst[i] = f2bf(v);
}

__global__ __launch_bounds__(256) void stem_conv_kernel(const bf16_t* __restrict__ canvas, const bf16_t* __restrict__ w32,
                                                        bf16_t* __restrict__ y, float* __restrict__ stats, int ntiles, int tiles_per_wave) {
    const int lane = threadIdx.x & 63, g = lane >> 4, r16 = lane & 15;
    const int gw = blockIdx.x * 4 + (threadIdx.x >> 6);
    // weight fragments: A operand rows = output channels; lane (r16, g) holds w32[co = t*16 + r16][ky][g*8 .. g*8+7]
    uint4 wq[4][7];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int ky = 0; ky < 7; ++ky) wq[t][ky] = *reinterpret_cast<const uint4*>(w32 + ((t * 16 + r16) * 7 + ky) * 32 + g * 8);
    float s1[4][4], s2[4][4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j) { s1[t][j] = 0.f; s2[t][j] = 0.f; }

    const int t_begin = gw * tiles_per_wave, t_end = min(ntiles, t_begin + tiles_per_wave);
    // tile -> (image n, output row oy, 16-pixel segment): 8 segments per row, 128 rows per image
    auto tile_ptr = [&](int tile) {
        const int seg = tile & 7, oy = (tile >> 3) & 127, n = tile >> 10;
        return canvas + ((size_t)(n * HP + 2 * oy) * WP + 2 * (seg * 16 + r16) + 2 * g) * 4;
    };
    uint4 xq[7], xn[7];
    if (t_begin < t_end) {
        const bf16_t* p = tile_ptr(t_begin);
#pragma unroll
        for (int ky = 0; ky < 7; ++ky) xq[ky] = *reinterpret_cast<const uint4*>(p + (size_t)ky * WP * 4);
    }
    for (int tile = t_begin; tile < t_end; ++tile) {
        if (tile + 1 < t_end) {  // next tile's fragments in flight under this tile's MFMAs
            const bf16_t* p = tile_ptr(tile + 1);
#pragma unroll
            for (int ky = 0; ky < 7; ++ky) xn[ky] = *reinterpret_cast<const uint4*>(p + (size_t)ky * WP * 4);
        }
        f32x4_t acc[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ky = 0; ky < 7; ++ky)
#pragma unroll
            for (int t = 0; t < 4; ++t)
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, wq[t][ky]), __builtin_bit_cast(bf16x8_t, xq[ky]),
                                                                 acc[t], 0, 0, 0);
        // D[i = g*4 + j (channel t*16 + i)][col = r16 (pixel)]: 4 consecutive channels of one pixel per lane
        const int seg = tile & 7, oy = (tile >> 3) & 127, n = tile >> 10;
        bf16_t* yp = y + ((size_t)(n * 128 + oy) * 128 + seg * 16 + r16) * 64 + g * 4;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            *reinterpret_cast<uint2*>(yp + t * 16) = make_uint2(pack_bf2(acc[t][0], acc[t][1]), pack_bf2(acc[t][2], acc[t][3]));
#pragma unroll
            for (int j = 0; j < 4; ++j) { s1[t][j] += acc[t][j]; s2[t][j] += acc[t][j] * acc[t][j]; }
        }
#pragma unroll
        for (int ky = 0; ky < 7; ++ky) xq[ky] = xn[ky];
    }
    if (stats != nullptr && t_begin < t_end) {  // one partial row per wave that had work: [gw][2][64]
        float* row = stats + (size_t)gw * 128 + g * 4;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            float u1[4], u2[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) { u1[j] = row16_sum(s1[t][j]); u2[j] = row16_sum(s2[t][j]); }
            if (r16 == 0) {
                *reinterpret_cast<float4*>(row + t * 16) = make_float4(u1[0], u1[1], u1[2], u1[3]);
                *reinterpret_cast<float4*>(row + 64 + t * 16) = make_float4(u2[0], u2[1], u2[2], u2[3]);
            }
        }
    }
}

constexpr int STEM_WAVES = 4096;  // 1024 workgroups

}  // namespace

extern "C" int gdrn_pack_stem_w32(const float* w, void* dst, int dtype, void* stream) {
    if (!w || !dst) return GDRN_ERR_ARG;
    if (dtype != GDRN_DT_BF16) return GDRN_ERR_SHAPE;
    hipLaunchKernelGGL(pack_stem_w32_kernel, dim3(cdiv(64 * 7 * 32, 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), w,
                       reinterpret_cast<bf16_t*>(dst));
    GDRN_CHECK_LAUNCH();
    return GDRN_OK;
}

// number of partial-statistics rows gdrn_stem_conv writes for N images (one per wave that has work)
extern "C" int gdrn_stem_stats_rows(int N) {
    const int ntiles = N * 128 * 8;
    const int tpw = cdiv(ntiles, STEM_WAVES);
    return cdiv(ntiles, tpw);
}

// canvas: [N][262][272][4] bf16 (gdrn_pack_image of 256x256 images); w32: gdrn_pack_stem_w32; y: [N][128][128][64] bf16;
// stats (nullable): [gdrn_stem_stats_rows(N)][2][64] fp32 partial sums / sums of squares for gdrn_bn_finalize.
extern "C" int gdrn_stem_conv(const void* canvas, const void* w32, void* y, float* stats, int N, int dtype, void* stream) {
    if (!canvas || !w32 || !y || N <= 0) return GDRN_ERR_ARG;
    if (dtype != GDRN_DT_BF16) return GDRN_ERR_SHAPE;
    if ((long long)N * HP * WP * 8 >= (1ll << 40)) return GDRN_ERR_SHAPE;
    const int ntiles = N * 128 * 8;
    const int tpw = cdiv(ntiles, STEM_WAVES);
    const int waves = cdiv(ntiles, tpw);
    hipLaunchKernelGGL(stem_conv_kernel, dim3(cdiv(waves, 4)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       reinterpret_cast<const bf16_t*>(canvas), reinterpret_cast<const bf16_t*>(w32), reinterpret_cast<bf16_t*>(y), stats,
                       ntiles, tpw);
    GDRN_CHECK_LAUNCH();
    return GDRN_OK;
}
