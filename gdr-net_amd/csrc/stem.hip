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
        for (int ky = 0; ky < 7; ++ky)  // fragment t, row r16 <-> channel (r16 >> 2) * 16 + t * 4 + (r16 & 3): a result lane owns 16 contiguous channels
            wq[t][ky] = *reinterpret_cast<const uint4*>(w32 + ((((r16 >> 2) * 16 + t * 4 + (r16 & 3)) * 7 + ky) * 32) + g * 8);
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
                acc[t] = GDRN_MFMA16(__builtin_bit_cast(bf16x8_t, wq[t][ky]), __builtin_bit_cast(bf16x8_t, xq[ky]),
                                                                 acc[t]);
        // D[i = g*4 + j][col = r16 (pixel)] of fragment t = channel g*16 + t*4 + j: 16 consecutive channels of one pixel per lane, two
        // 16-byte stores (four 8-byte ones with the plain row order)
        const int seg = tile & 7, oy = (tile >> 3) & 127, n = tile >> 10;
        bf16_t* yp = y + ((size_t)(n * 128 + oy) * 128 + seg * 16 + r16) * 64 + g * 16;
        *reinterpret_cast<uint4*>(yp) = make_uint4(pack_bf2(acc[0][0], acc[0][1]), pack_bf2(acc[0][2], acc[0][3]), pack_bf2(acc[1][0], acc[1][1]), pack_bf2(acc[1][2], acc[1][3]));
        *reinterpret_cast<uint4*>(yp + 8) = make_uint4(pack_bf2(acc[2][0], acc[2][1]), pack_bf2(acc[2][2], acc[2][3]), pack_bf2(acc[3][0], acc[3][1]), pack_bf2(acc[3][2], acc[3][3]));
#pragma unroll
        for (int t = 0; t < 4; ++t) {
#pragma unroll
            for (int j = 0; j < 4; ++j) { s1[t][j] += acc[t][j]; s2[t][j] += acc[t][j] * acc[t][j]; }
        }
#pragma unroll
        for (int ky = 0; ky < 7; ++ky) xq[ky] = xn[ky];
    }
    if (stats != nullptr && t_begin < t_end) {  // one partial row per wave that had work: [gw][2][64]
        float* row = stats + (size_t)gw * 128 + g * 16;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            float u1[4], u2[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) { u1[j] = row16_sum(s1[t][j]); u2[j] = row16_sum(s2[t][j]); }
            if (r16 == 0) {
                *reinterpret_cast<float4*>(row + t * 4) = make_float4(u1[0], u1[1], u1[2], u1[3]);
                *reinterpret_cast<float4*>(row + 64 + t * 4) = make_float4(u2[0], u2[1], u2[2], u2[3]);
            }
        }
    }
}

constexpr int STEM_WAVES = 4096;  // 1024 workgroups

// ---------------------------------------------------------------------------------------------------------------------
// Eval-mode stem in ONE kernel (r5): conv 7x7 s2 -> BatchNorm with running statistics (per-channel scale / shift) -> ReLU -> max-pool 3x3 s2 pad 1
// (resnet_backbone.py:69-72 with module.eval()).  As two kernels the 134 MB conv output is written and fetched 1.5 times again by the pool
// (53 + 43 us at bs = 64); here it never leaves the registers.  A wave owns a vertical strip -- 16 conv columns x 33 conv rows of one image =
// 7 x 16 pooled pixels -- and walks it downwards:
//   * columns: the strip starts at the ODD conv column 14k - 1, so that the seven pooling windows centred on conv columns 14k, 14k + 2, ..
//     lie inside it (result-lane column r16 = 1, 3, .. 13 with its neighbours r16 +- 1: two DPP row shifts, no LDS); strips advance by 14;
//   * rows: consecutive conv rows share five of their seven canvas rows -- a sliding window of seven 16-byte fragments per lane, two new
//     loads per conv row instead of seven; the vertical max runs on the way: h(row) = horizontal 3-max, even row 2p: m = max(h(2p-1), h(2p)),
//     odd row 2p+1: pooled row p = max(m, h(2p+1)) is stored and h(2p+1) kept for p + 1;
//   * arithmetic: the accumulator is rounded to the 16-bit format before the affine, exactly where the two-kernel path stores it, so the
//     pooled tensor is bit-identical to gdrn_stem_conv -> gdrn_bn_relu_maxpool_fwd (post-ReLU values are >= 0: padding contributes 0).
constexpr int SP_BAND = 16;     // pooled rows per wave
constexpr int SP_STRIPS = 10;   // strips per row: pooled columns 7k .. 7k + 6 (the last strip holds column 63 alone)

__device__ __forceinline__ float dpp_row_shr1(float v) {   // lane i <- lane i - 1 of its 16-lane row, 0 at the row's first lane
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x111, 0xF, 0xF, true));
}
__device__ __forceinline__ float dpp_row_shl1(float v) {   // lane i <- lane i + 1, 0 at the row's last lane
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x101, 0xF, 0xF, true));
}

__global__ __launch_bounds__(256) void stem_conv_pool_kernel(const bf16_t* __restrict__ canvas, const bf16_t* __restrict__ w32,
                                                             const float* __restrict__ scale, const float* __restrict__ shift,
                                                             bf16_t* __restrict__ y, int nwaves) {
    const int lane = threadIdx.x & 63, g = lane >> 4, r16 = lane & 15;
    const int gw = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (gw >= nwaves) return;
    const int strip = gw % SP_STRIPS, band = (gw / SP_STRIPS) % (64 / SP_BAND), n = gw / (SP_STRIPS * (64 / SP_BAND));
    uint4 wq[4][7];   // as stem_conv_kernel: fragment t, row r16 <-> channel (r16 >> 2) * 16 + t * 4 + (r16 & 3)
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int ky = 0; ky < 7; ++ky)
            wq[t][ky] = *reinterpret_cast<const uint4*>(w32 + ((((r16 >> 2) * 16 + t * 4 + (r16 & 3)) * 7 + ky) * 32) + g * 8);
    float sc[4][4], sh[4][4];   // the lane's 16 result channels g*16 + t*4 + j
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const float4 a = *reinterpret_cast<const float4*>(scale + g * 16 + t * 4), b = *reinterpret_cast<const float4*>(shift + g * 16 + t * 4);
        sc[t][0] = a.x; sc[t][1] = a.y; sc[t][2] = a.z; sc[t][3] = a.w;
        sh[t][0] = b.x; sh[t][1] = b.y; sh[t][2] = b.z; sh[t][3] = b.w;
    }
    const int ox = 14 * strip - 1 + r16;                 // this lane's conv column
    const bool col_ok = (unsigned)ox < 128u;
    const int oxc = min(max(ox, 0), 127);
    const int pb = band * SP_BAND;                       // first pooled row
    const int oy0 = 2 * pb - 1, oy1 = 2 * pb + 2 * SP_BAND - 1;   // conv rows oy0 (halo: feeds pooled row pb only) .. oy1
    const int oys = max(oy0, 0);
    // canvas row r, this lane's 8 pixels x 4 channels starting at canvas column 2*ox (+ 2*g pixels for k-group g)
    const bf16_t* cbase = canvas + ((size_t)n * HP * WP + 2 * oxc + 2 * g) * 4;
    auto crow = [&](int r) { return *reinterpret_cast<const uint4*>(cbase + (size_t)r * WP * 4); };
    uint4 xr[7];
#pragma unroll
    for (int ky = 0; ky < 7; ++ky) xr[ky] = crow(2 * oys + ky);
    float hprev[4][4], m[4][4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j) { hprev[t][j] = 0.f; m[t][j] = 0.f; }
    const int px = 7 * strip + (r16 >> 1);               // pooled column of an odd result lane
    const bool store_lane = (r16 & 1) && r16 <= 13 && px < 64;
    for (int oy = oys; oy <= oy1; ++oy) {
        uint4 n0 = xr[0], n1 = xr[1];
        if (oy < oy1) { n0 = crow(2 * oy + 7); n1 = crow(2 * oy + 8); }   // the next conv row's two new canvas rows, under this row's MFMAs
        f32x4_t acc[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ky = 0; ky < 7; ++ky)
#pragma unroll
            for (int t = 0; t < 4; ++t)
                acc[t] = GDRN_MFMA16(__builtin_bit_cast(bf16x8_t, wq[t][ky]), __builtin_bit_cast(bf16x8_t, xr[ky]), acc[t]);
        float h[4][4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const uint32_t p01 = pack_bf2(acc[t][0], acc[t][1]), p23 = pack_bf2(acc[t][2], acc[t][3]);   // the stored conv output
            const float x[4] = {h16lo(p01), h16hi(p01), h16lo(p23), h16hi(p23)};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float v = fmaxf(x[j] * sc[t][j] + sh[t][j], 0.f);
                v = col_ok ? v : 0.f;
                h[t][j] = fmaxf(v, fmaxf(dpp_row_shr1(v), dpp_row_shl1(v)));
            }
        }
        if (oy & 1) {
            if (oy > oy0 && store_lane) {   // pooled row (oy - 1) / 2 is complete
                uint32_t o[8];
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    o[2 * t] = pack_bf2(fmaxf(m[t][0], h[t][0]), fmaxf(m[t][1], h[t][1]));
                    o[2 * t + 1] = pack_bf2(fmaxf(m[t][2], h[t][2]), fmaxf(m[t][3], h[t][3]));
                }
                bf16_t* yp = y + ((size_t)(n * 64 + (oy >> 1)) * 64 + px) * 64 + g * 16;
                *reinterpret_cast<uint4*>(yp) = make_uint4(o[0], o[1], o[2], o[3]);
                *reinterpret_cast<uint4*>(yp + 8) = make_uint4(o[4], o[5], o[6], o[7]);
            }
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int j = 0; j < 4; ++j) hprev[t][j] = h[t][j];
        } else {
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int j = 0; j < 4; ++j) m[t][j] = fmaxf(hprev[t][j], h[t][j]);
        }
#pragma unroll
        for (int ky = 0; ky < 5; ++ky) xr[ky] = xr[ky + 2];
        xr[5] = n0;
        xr[6] = n1;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Stem weight gradient fused with the BatchNorm-backward "apply" that precedes it.  The stem has no data gradient, so the
// gradient w.r.t. the conv output, dy = a*g + (b*x + c) per channel (bn_bwd_apply_kernel's formula, rounded to bf16 as that
// kernel stores it), is consumed by this weight gradient only: it is evaluated here while the tile is staged, instead of
// being written (134 MB at bs = 64) and read back seven times by the generic per-tap weight-gradient tiles.
//   dW[co][ky][n = kx*4 + c] = sum_{pixels} dy[pixel][co] * canvas[2oy + ky][2ox*4 + n]        (n < 32, kx = 7 / c = 3 unused)
// A stage = 64 consecutive output pixels of one row: dy^T fragments through the LDS transpose read from a [64 px][64 co]
// tile; the seven canvas rows are kept as contiguous 1072-byte spans and read as a [pixel][32] operand with a 16-byte row
// pitch (row p starts at pixel 2*p = 16 B further: the overlap IS the stride-2 window).  A wave owns 16 output channels x
// all 224 columns (14 accumulator tiles); workgroups own contiguous stage ranges and leave fp32 partials [wg][64][224].
typedef __attribute__((address_space(3))) bf16x4_t lds_bf16x4_t;
constexpr int SW_PA = 160;              // dy tile pitch (128 B + 32 B pad, as conv_wgrad.hip)
constexpr int SW_SPAN = 1088;           // 67 segments of 16 B used, 68 allocated
constexpr int SW_TILE = 64 * SW_PA + 7 * SW_SPAN;  // one stage buffer
constexpr int SW_PARTS = 512;

__global__ __launch_bounds__(256, 2) void stem_wgrad_kernel(const bf16_t* __restrict__ canvas, const bf16_t* __restrict__ gg,
                                                            const bf16_t* __restrict__ raw, const float* __restrict__ ca,
                                                            const float* __restrict__ cb, const float* __restrict__ cc,
                                                            int nstages, int per, float* __restrict__ ws) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * SW_TILE];
    __shared__ float kst[3][64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, r16 = lane & 15;
    if (tid < 64) {
        float a = 1.f, b = 0.f, c = 0.f;
        if (ca != nullptr) { a = ca[tid]; b = cb[tid]; c = cc[tid]; }  // gdrn_bn_bwd_coef's vectors
        kst[0][tid] = a; kst[1][tid] = b; kst[2][tid] = c;
    }
    __syncthreads();
    const int segA = tid & 7, rowA = tid >> 3;  // this thread's 16-byte channel segment / first pixel row of the dy tile
    float ka[8], kb[8], kc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { ka[j] = kst[0][segA * 8 + j]; kb[j] = kst[1][segA * 8 + j]; kc[j] = kst[2][segA * 8 + j]; }
    const bool fused = ca != nullptr;

    const int s_begin = blockIdx.x * per, s_end = min(nstages, s_begin + per);
    // two register sets: the loads of stage s+2 are issued before the MFMAs of stage s.  Named registers and macros, not arrays
    // behind lambda pointers: hipcc put those in scratch (80 B / lane, a scratch round trip per stage).
    uint4 g0a, g0b, x0a, x0b, c0a, c0b, g1a, g1b, x1a, x1b, c1a, c1b;
    const int cid0 = tid, cid1 = min(tid + 256, 7 * 67 - 1);
    const int cky0 = cid0 / 67, cky1 = cid1 / 67;
    const unsigned coff0 = (unsigned)(cky0 * WP * 4 + (cid0 - cky0 * 67) * 8), coff1 = (unsigned)(cky1 * WP * 4 + (cid1 - cky1 * 67) * 8);
    const int cdst0 = cky0 * SW_SPAN + (cid0 - cky0 * 67) * 16, cdst1 = (tid + 256 < 7 * 67) ? cky1 * SW_SPAN + (cid1 - cky1 * 67) * 16 : -1;
    const unsigned aoff0 = (unsigned)(rowA * 64 + segA * 8), aoff1 = (unsigned)((rowA + 32) * 64 + segA * 8);
    const uint4 zero4 = make_uint4(0, 0, 0, 0);
#define SW_LOAD(S_, GA, GB, XA, XB, CA, CB)                                                                      \
    {                                                                                                            \
        const int h_ = (S_) & 1, oy_ = ((S_) >> 1) & 127, n_ = (S_) >> 8;                                        \
        const size_t p0_ = ((size_t)(n_ * 128 + oy_) * 128 + h_ * 64) * 64;                                      \
        GA = *reinterpret_cast<const uint4*>(gg + p0_ + aoff0);                                                  \
        GB = *reinterpret_cast<const uint4*>(gg + p0_ + aoff1);                                                  \
        XA = fused ? *reinterpret_cast<const uint4*>(raw + p0_ + aoff0) : zero4;                                 \
        XB = fused ? *reinterpret_cast<const uint4*>(raw + p0_ + aoff1) : zero4;                                 \
        const bf16_t* cb_ = canvas + ((size_t)(n_ * HP + 2 * oy_) * WP + 2 * (h_ * 64)) * 4;                     \
        CA = *reinterpret_cast<const uint4*>(cb_ + coff0);                                                       \
        CB = *reinterpret_cast<const uint4*>(cb_ + coff1);                                                       \
    }
    auto bn_apply = [&](uint4 gq, uint4 xq) -> uint4 {
        if (!fused) return gq;
        float gv[8], xv[8], o[8];
        Vec16<bf16_t>::unpack(gq, gv);
        Vec16<bf16_t>::unpack(xq, xv);
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = ka[j] * gv[j] + (kb[j] * xv[j] + kc[j]);
        return Vec16<bf16_t>::pack(o);
    };
#define SW_WRITE(BUF_, GA, GB, XA, XB, CA, CB)                                                                   \
    {                                                                                                            \
        unsigned char* tA_ = smem + (BUF_) * SW_TILE;                                                            \
        unsigned char* tX_ = tA_ + 64 * SW_PA;                                                                   \
        *reinterpret_cast<uint4*>(tA_ + rowA * SW_PA + segA * 16) = bn_apply(GA, XA);                            \
        *reinterpret_cast<uint4*>(tA_ + (rowA + 32) * SW_PA + segA * 16) = bn_apply(GB, XB);                     \
        *reinterpret_cast<uint4*>(tX_ + cdst0) = CA;                                                             \
        if (cdst1 >= 0) *reinterpret_cast<uint4*>(tX_ + cdst1) = CB;                                             \
    }

    f32x4_t acc[7][2];
#pragma unroll
    for (int ky = 0; ky < 7; ++ky) { acc[ky][0] = f32x4_t{0.f, 0.f, 0.f, 0.f}; acc[ky][1] = f32x4_t{0.f, 0.f, 0.f, 0.f}; }

    auto compute = [&](int buf) {
        const unsigned char* tA = smem + buf * SW_TILE;
        const unsigned char* tX = tA + 64 * SW_PA;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            // transpose-read map of conv_wgrad.hip: lane group g takes reduction rows {4g..4g+3} and {16+4g..} of the k-step
            const int trow = ks * 32 + g * 4 + (r16 >> 2), tcol = (r16 & 3) * 4;
            const unsigned char* qa = tA + trow * SW_PA + (wave * 16 + tcol) * 2;
            const bf16x4_t alo = GDRN_TR16((lds_bf16x4_t*)(qa));
            const bf16x4_t ahi = GDRN_TR16((lds_bf16x4_t*)(qa + 16 * SW_PA));
            const bf16x8_t fa = __builtin_shufflevector(alo, ahi, 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
            for (int ky = 0; ky < 7; ++ky) {
#pragma unroll
                for (int nf = 0; nf < 2; ++nf) {
                    const unsigned char* qb = tX + ky * SW_SPAN + trow * 16 + (nf * 16 + tcol) * 2;
                    const bf16x4_t blo = GDRN_TR16((lds_bf16x4_t*)(qb));
                    const bf16x4_t bhi = GDRN_TR16((lds_bf16x4_t*)(qb + 16 * 16));
                    const bf16x8_t fb = __builtin_shufflevector(blo, bhi, 0, 1, 2, 3, 4, 5, 6, 7);
                    acc[ky][nf] = GDRN_MFMA16(fa, fb, acc[ky][nf]);
                }
            }
        }
    };
    if (s_begin < s_end) {
        SW_LOAD(s_begin, g0a, g0b, x0a, x0b, c0a, c0b)
        if (s_begin + 1 < s_end) SW_LOAD(s_begin + 1, g1a, g1b, x1a, x1b, c1a, c1b)
        SW_WRITE(0, g0a, g0b, x0a, x0b, c0a, c0b)
    }
    __syncthreads();
    for (int s = s_begin; s < s_end; s += 2) {
        // even step: buffer 0 holds stage s, set 1 holds stage s+1 (in flight), set 0 is free for stage s+2
        if (s + 2 < s_end) SW_LOAD(s + 2, g0a, g0b, x0a, x0b, c0a, c0b)
        compute(0);
        if (s + 1 < s_end) SW_WRITE(1, g1a, g1b, x1a, x1b, c1a, c1b)
        __syncthreads();
        if (s + 1 >= s_end) break;
        // odd step: buffer 1 holds stage s+1, set 0 holds stage s+2, set 1 is free for stage s+3
        if (s + 3 < s_end) SW_LOAD(s + 3, g1a, g1b, x1a, x1b, c1a, c1b)
        compute(1);
        if (s + 2 < s_end) SW_WRITE(0, g0a, g0b, x0a, x0b, c0a, c0b)
        __syncthreads();
    }
#undef SW_LOAD
#undef SW_WRITE
    // D[i = g*4 + j -> co = wave*16 + i][col = r16 -> n = nf*16 + r16]; every workgroup writes its slab (zeros if it had no stage)
    float* slab = ws + (size_t)blockIdx.x * 64 * 224;
#pragma unroll
    for (int ky = 0; ky < 7; ++ky)
#pragma unroll
        for (int nf = 0; nf < 2; ++nf)
#pragma unroll
            for (int j = 0; j < 4; ++j) slab[(wave * 16 + g * 4 + j) * 224 + ky * 32 + nf * 16 + r16] = acc[ky][nf][j];
}

// grad OIHW [64][3][7][7] (zeroed by the caller) += sum over a slice of the partial slabs; grid (64 output channels, SW_SLICES)
constexpr int SW_SLICES = 8;
__global__ __launch_bounds__(256) void stem_wgrad_reduce_kernel(const float* __restrict__ ws, int parts, float* __restrict__ grad) {
    const int co = blockIdx.x, e = threadIdx.x;
    if (e >= 224) return;
    const int per = (parts + SW_SLICES - 1) / SW_SLICES, q0 = blockIdx.y * per, q1 = min(parts, q0 + per);
    const float* p = ws + (size_t)co * 224 + e;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int q = q0;
    for (; q + 4 <= q1; q += 4) {
        s0 += p[(size_t)(q + 0) * 64 * 224];
        s1 += p[(size_t)(q + 1) * 64 * 224];
        s2 += p[(size_t)(q + 2) * 64 * 224];
        s3 += p[(size_t)(q + 3) * 64 * 224];
    }
    for (; q < q1; ++q) s0 += p[(size_t)q * 64 * 224];
    const int ky = e >> 5, kx = (e >> 2) & 7, c = e & 3;
    if (kx < 7 && c < 3) atomicAdd(grad + ((co * 3 + c) * 7 + ky) * 7 + kx, (s0 + s1) + (s2 + s3));
}

}  // namespace

extern "C" int gdrn_stem_wgrad_parts(int N) {
    const int nstages = N * 256;
    const int per = cdiv(nstages, SW_PARTS);
    return cdiv(nstages, per);
}

// Weight gradient of the stem conv, fused with the BatchNorm-backward apply in front of it (a != NULL): see above.
//   canvas [N][262][272][4] bf16; g, raw [N][128][128][64] bf16 (masked upstream gradient / the conv output BatchNorm saw);
//   a, b, c [64]: the coefficients of dy = a*g + (b*raw + c) from gdrn_bn_bwd_coef; a == NULL: plain weight gradient with dy = g.
//   ws: gdrn_stem_wgrad_parts(N) x 64 x 224 floats of scratch; grad: fp32 OIHW [64][3][7][7], overwritten.
extern "C" int gdrn_stem_wgrad(const void* canvas, const void* g, const void* raw, const float* a, const float* b, const float* c, int N,
                               float* ws, float* grad, int dtype, void* stream) {
    if (!canvas || !g || !ws || !grad || N <= 0) return GDRN_ERR_ARG;
    if (a != nullptr && (!raw || !b || !c)) return GDRN_ERR_ARG;
    const bool prezeroed = (dtype & GDRN_PREZEROED) != 0;
    dtype &= ~GDRN_PREZEROED;
    if (dtype != GDRN_DT_H16) return GDRN_ERR_SHAPE;
    if ((long long)N * 128 * 128 * 64 >= (1ll << 40)) return GDRN_ERR_SHAPE;
    const int nstages = N * 256;
    const int per = cdiv(nstages, SW_PARTS);
    const int parts = cdiv(nstages, per);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    GDRN_LAUNCH(stem_wgrad_kernel, dim3(parts), dim3(256), 0, st, reinterpret_cast<const bf16_t*>(canvas), reinterpret_cast<const bf16_t*>(g),
                       reinterpret_cast<const bf16_t*>(raw), a, b, c, nstages, per, ws);
    GDRN_CHECK_LAUNCH();
    if (!prezeroed && hipMemsetAsync(grad, 0, 64 * 147 * sizeof(float), st) != hipSuccess) return GDRN_ERR_LAUNCH;
    GDRN_LAUNCH(stem_wgrad_reduce_kernel, dim3(64, SW_SLICES), dim3(256), 0, st, ws, parts, grad);
    GDRN_CHECK_LAUNCH();
    return GDRN_OK;
}

extern "C" int gdrn_pack_stem_w32(const float* w, void* dst, int dtype, void* stream) {
    if (!w || !dst) return GDRN_ERR_ARG;
    if (dtype != GDRN_DT_H16) return GDRN_ERR_SHAPE;
    GDRN_LAUNCH(pack_stem_w32_kernel, dim3(cdiv(64 * 7 * 32, 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), w,
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
// eval mode: conv + BatchNorm (scale / shift of the running statistics) + ReLU + 3x3 s2 max-pool in one pass; y: [N][64][64][64], bit-identical
// to gdrn_stem_conv -> gdrn_bn_relu_maxpool_fwd
extern "C" int gdrn_stem_conv_pool(const void* canvas, const void* w32, const float* scale, const float* shift, void* y, int N, int dtype, void* stream) {
    if (!canvas || !w32 || !scale || !shift || !y || N <= 0) return GDRN_ERR_ARG;
    if (dtype != GDRN_DT_H16) return GDRN_ERR_SHAPE;
    if ((long long)N * HP * WP * 8 >= (1ll << 40)) return GDRN_ERR_SHAPE;
    const int nwaves = N * (64 / SP_BAND) * SP_STRIPS;
    GDRN_LAUNCH(stem_conv_pool_kernel, dim3(cdiv(nwaves, 4)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                reinterpret_cast<const bf16_t*>(canvas), reinterpret_cast<const bf16_t*>(w32), scale, shift, reinterpret_cast<bf16_t*>(y), nwaves);
    GDRN_CHECK_LAUNCH();
    return GDRN_OK;
}

extern "C" int gdrn_stem_conv(const void* canvas, const void* w32, void* y, float* stats, int N, int dtype, void* stream) {
    if (!canvas || !w32 || !y || N <= 0) return GDRN_ERR_ARG;
    if (dtype != GDRN_DT_H16) return GDRN_ERR_SHAPE;
    if ((long long)N * HP * WP * 8 >= (1ll << 40)) return GDRN_ERR_SHAPE;
    const int ntiles = N * 128 * 8;
    const int tpw = cdiv(ntiles, STEM_WAVES);
    const int waves = cdiv(ntiles, tpw);
    GDRN_LAUNCH(stem_conv_kernel, dim3(cdiv(waves, 4)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       reinterpret_cast<const bf16_t*>(canvas), reinterpret_cast<const bf16_t*>(w32), reinterpret_cast<bf16_t*>(y), stats,
                       ntiles, tpw);
    GDRN_CHECK_LAUNCH();
    return GDRN_OK;
}
