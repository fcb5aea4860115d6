// One ResNet BasicBlock of 64 channels in EVAL mode as ONE launch (round 6):
//     y = relu(conv2(relu(conv1(x) + b1)) + b2 + x)
// with the BatchNorms folded into the weights / biases (Engine.fold): the three blocks of ResNet-34's layer1 at inference
// (resnet_backbone.py:69-80 + the torchvision BasicBlock; gdrn_evaluator.py:556-601 runs it under model.eval()).
//
// In eval mode nothing between the two convs needs the whole batch (train-mode BatchNorm does: its statistics), so a workgroup that owns an
// 8 x 16 tile of output pixels computes conv1 on the 10 x 18 pixels conv2 needs (input patch 12 x 20), keeps that intermediate in LDS in the
// operand layout, runs conv2 on it and adds the identity from the input patch it still holds: the 64-channel intermediate (33 MB per block at
// bs = 64) never goes to HBM, the input is read once instead of twice (second conv's identity), and three launches replace six.  The price is
// conv1 on 180 instead of 128 pixels (+40 % of its MFMAs; the layer runs at ~0.2 of the MFMA peak and 0.5 of the HBM rate as two launches).
//
// Everything else is conv3x3_halo.hip's 64-channel tile: 4 waves, wave w owns output channels [16w, 16w + 16) of both convs for all pixels,
// v_mfma_f32_16x16x32 with the fragment-major weight blocks (gdrn_pack_wfrag, 1 KiB per 16 channels x tap x k-step) streamed L2 -> VGPR through
// a register ring, pixel operands by ds_read_b128 from the even / odd granule patch layout (PITCH 80: conflict-free for every tap shift).
// conv1's pixel fragments are 16 consecutive pixels of the 10 x 18 region in row-major order (12 fragments, the last one 4 pixels), so every
// fragment has its own lane base (12 registers) and the taps are immediates on top of it, as in the halo kernel.
// The accumulation order (tap 0..8, k-step 0 then 1) and the epilogue arithmetic (acc + bias (+ identity), ReLU, one rounding to the storage
// format) are the two-launch path's: the result is BIT-IDENTICAL to conv1 -> conv2 on the halo kernel (tests/test_kernels_gpu.py).
#include <algorithm>
#include <mutex>
#include <type_traits>
#include <utility>

#include "common.h"
#include "../../include/gdrn_hip.h"

namespace {

// compile-time loop: f(std::integral_constant<int, 0>) ... f(std::integral_constant<int, N - 1>)
template <class F, int... Is>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, Is...>) { (f(std::integral_constant<int, Is>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }

constexpr int PITCH = 80;   // LDS pixel pitch inside one half array (conv3x3_halo.hip)
__host__ __device__ constexpr int half_bytes(int ppix) { return (ppix * PITCH + 255) / 256 * 256; }

constexpr int TH = 8, TW = 16;
constexpr int XW = TW + 4, XH = TH + 4, XPIX = XW * XH;      // input patch 12 x 20
constexpr int AW = TW + 2, AH = TH + 2, APIX = AW * AH;      // intermediate 10 x 18
constexpr int HBX = half_bytes(XPIX), HBA = half_bytes(APIX);
constexpr int XBYTES = 2 * HBX, ABYTES = 2 * HBA;            // 38400 + 29184 = 67584 B: two workgroups per CU
constexpr int F1 = (APIX + 15) / 16;                         // 12 pixel fragments of conv1
constexpr int F2 = TH * TW / 16;                             // 8 pixel fragments of conv2

__device__ __forceinline__ f32x4_t mma(uint4 a, uint4 b, f32x4_t c) {
    return GDRN_MFMA16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c);
}

__global__ __launch_bounds__(256, 2) void block64_eval_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ w1, const float* __restrict__ b1,
                                                              const bf16_t* __restrict__ w2, const float* __restrict__ b2, bf16_t* __restrict__ y,
                                                              int N, int H, int W) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* xp = smem;             // input patch, operand layout
    unsigned char* ap = smem + XBYTES;    // conv1's output on the 10 x 18 region, operand layout
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, r16 = lane & 15;
    __builtin_amdgcn_s_setprio(2);

    int bid = blockIdx.x;
    {   // XCD-aware order (neighbouring tiles share an L2)
        const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = bid & 7, j = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
    }
    const int tiles_x = W / TW, tiles_y = H / TH;
    const int tx = bid % tiles_x, ty = (bid / tiles_x) % tiles_y, n = bid / (tiles_x * tiles_y);
    const int y0 = ty * TH, x0 = tx * TW;

    // ---- weight streams of this wave: conv1 then conv2, 18 blocks of 1 KiB each (tap-major, k-step minor), register ring of 3 stages (= taps)
    const char* wl1 = reinterpret_cast<const char*>(w1) + (size_t)lane * 16 + ((size_t)(wave * 9) << 11);
    const char* wl2 = reinterpret_cast<const char*>(w2) + (size_t)lane * 16 + ((size_t)(wave * 9) << 11);
    auto wptr = [&](const char* wl, int tap, int ks) -> const uint4* { return reinterpret_cast<const uint4*>(wl + ((size_t)(tap * 2 + ks) << 10)); };
    uint4 wq0[2], wq1[2], wq2[2];
#define LOADW(dst, wl_, tap_) { dst[0] = *wptr(wl_, tap_, 0); dst[1] = *wptr(wl_, tap_, 1); }
    LOADW(wq0, wl1, 0) LOADW(wq1, wl1, 1) LOADW(wq2, wl1, 2)

    // ---- input patch: 240 pixels x 8 granules, zero outside the image
    {
        uint4 v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int id = i * 256 + tid;
            const int pp = id >> 3, sg = id & 7;
            const int py = pp / XW, px = pp - py * XW;
            const int iy = y0 - 2 + py, ix = x0 - 2 + px;
            const bool ok = id < XPIX * 8 && iy >= 0 && iy < H && ix >= 0 && ix < W;
            const int iyc = min(max(iy, 0), H - 1), ixc = min(max(ix, 0), W - 1);
            const uint4 t = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(x) + ((size_t)((n * H + iyc) * W + ixc) * 128 + sg * 16));
            v[i] = ok ? t : make_uint4(0, 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int id = i * 256 + tid;
            const int pp = id >> 3, sg = id & 7;
            if (id < XPIX * 8) *reinterpret_cast<uint4*>(xp + pp * PITCH + (sg & 1) * HBX + (sg >> 1) * 16) = v[i];
        }
    }
    // lane bases of conv1's pixel fragments: fragment f, lane column r16 -> linear pixel 16 f + r16 of the 10 x 18 region (clamped: the last
    // fragment's 12 surplus lanes compute a copy of pixel 179 that is never stored)
    int lb1[F1];
#pragma unroll
    for (int f = 0; f < F1; ++f) {
        const int L = min(f * 16 + r16, APIX - 1);
        const int py = L / AW, px = L - py * AW;
        lb1[f] = (py * XW + px) * PITCH + (g & 1) * HBX + (g >> 1) * 16;
    }
    __syncthreads();

    // ================= conv1 on the 10 x 18 region
    f32x4_t acc1[F1];
#pragma unroll
    for (int f = 0; f < F1; ++f) acc1[f] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    // 36 stages (tap, k-step, half of the 12 fragments) of 6 MFMAs: the 6 fragment reads of stage s + 1 go out before the MFMAs of stage s
    // (two 6-fragment buffers), so a wave's LDS latency hides under its own MFMAs -- the halo kernel's schedule; written as one stage per
    // scheduling region (hipcc otherwise hoists every read of a tap in front of its first MFMA and the wave waits ~150 cycles per tap)
    uint4 fA[6], fB[6];
    auto rd1 = [&](uint4 (&dst)[6], auto S_) {
        constexpr int s_ = decltype(S_)::value, tap = s_ / 4, ks = (s_ / 2) % 2, half = s_ % 2;
        constexpr int tsh = ((tap / 3) * XW + (tap % 3)) * PITCH + ks * 32;
#pragma unroll
        for (int i = 0; i < 6; ++i) dst[i] = *reinterpret_cast<const uint4*>(xp + lb1[half * 6 + i] + tsh);
    };
    rd1(fA, std::integral_constant<int, 0>{});
    static_for<36>([&](auto S_) {
        constexpr int s_ = decltype(S_)::value, tap = s_ / 4, ks = (s_ / 2) % 2, half = s_ % 2;
        if constexpr (s_ + 1 < 36) {
            if constexpr (s_ % 2 == 0) rd1(fB, std::integral_constant<int, s_ + 1>{});
            else rd1(fA, std::integral_constant<int, s_ + 1>{});
        }
        uint4 (&src)[6] = (s_ % 2 == 0) ? fA : fB;
        uint4 (&wq)[2] = (tap % 3 == 0) ? wq0 : ((tap % 3 == 1) ? wq1 : wq2);
#pragma unroll
        for (int i = 0; i < 6; ++i) acc1[half * 6 + i] = mma(wq[ks], src[i], acc1[half * 6 + i]);
        if constexpr (s_ % 4 == 3) {   // the tap is done: its ring slot takes the weights three taps ahead (running on into conv2's)
            if constexpr (tap + 3 < 9) LOADW(wq, wl1, tap + 3)
            else LOADW(wq, wl2, tap + 3 - 9)
        }
        __builtin_amdgcn_sched_barrier(0);
    });

    // conv1's epilogue: + bias, ReLU, one rounding, zero outside the image (conv2's zero padding is a property of ITS input), into the
    // intermediate patch: lane = channels 16 wave + 4 g .. + 3 of pixel 16 f + r16 -> 8 bytes inside granule 2 wave + (g >> 1)
    {
        const float4 bv = *reinterpret_cast<const float4*>(b1 + wave * 16 + g * 4);
        const int gi = 2 * wave + (g >> 1);
        const int goff = (gi & 1) * HBA + (gi >> 1) * 16 + (g & 1) * 8;
#pragma unroll
        for (int f = 0; f < F1; ++f) {
            const int L = f * 16 + r16;
            const int py = L / AW, px = L - py * AW;
            const int iy = y0 - 1 + py, ix = x0 - 1 + px;
            const bool in = iy >= 0 && iy < H && ix >= 0 && ix < W;
            const float v0 = fmaxf(acc1[f][0] + bv.x, 0.f), v1 = fmaxf(acc1[f][1] + bv.y, 0.f);
            const float v2 = fmaxf(acc1[f][2] + bv.z, 0.f), v3 = fmaxf(acc1[f][3] + bv.w, 0.f);
            const uint2 o = in ? make_uint2(pack_bf2(v0, v1), pack_bf2(v2, v3)) : make_uint2(0u, 0u);
            if (L < APIX) *reinterpret_cast<uint2*>(ap + L * PITCH + goff) = o;
        }
    }
    __syncthreads();

    // ================= conv2 on the tile
    const int lbase2 = r16 * PITCH + (g & 1) * HBA + (g >> 1) * 16;
    constexpr int FROW = AW * PITCH;
    f32x4_t acc2[F2];
#pragma unroll
    for (int b = 0; b < F2; ++b) acc2[b] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    // 18 stages (tap, k-step) of 8 MFMAs, the 8 fragment reads of the next stage in front of them (two 8-fragment buffers)
    uint4 gA[F2], gB[F2];
    auto rd2 = [&](uint4 (&dst)[F2], auto S_) {
        constexpr int s_ = decltype(S_)::value, tap = s_ / 2, ks = s_ % 2;
        constexpr int tsh = ((tap / 3) * AW + (tap % 3)) * PITCH + ks * 32;
#pragma unroll
        for (int b = 0; b < F2; ++b) dst[b] = *reinterpret_cast<const uint4*>(ap + lbase2 + b * FROW + tsh);
    };
    rd2(gA, std::integral_constant<int, 0>{});
    static_for<18>([&](auto S_) {
        constexpr int s_ = decltype(S_)::value, tap = s_ / 2, ks = s_ % 2;
        if constexpr (s_ + 1 < 18) {
            if constexpr (s_ % 2 == 0) rd2(gB, std::integral_constant<int, s_ + 1>{});
            else rd2(gA, std::integral_constant<int, s_ + 1>{});
        }
        uint4 (&src)[F2] = (s_ % 2 == 0) ? gA : gB;
        uint4 (&wq)[2] = (tap % 3 == 0) ? wq0 : ((tap % 3 == 1) ? wq1 : wq2);
#pragma unroll
        for (int b = 0; b < F2; ++b) acc2[b] = mma(wq[ks], src[b], acc2[b]);
        if constexpr (ks == 1 && tap + 3 < 9) LOADW(wq, wl2, tap + 3)
        __builtin_amdgcn_sched_barrier(0);
    });
#undef LOADW

    // conv2's epilogue: + bias + identity (the input patch still in LDS: tile pixel (b, r16) = patch pixel (b + 2, r16 + 2)), ReLU, store
    {
        const float4 bv = *reinterpret_cast<const float4*>(b2 + wave * 16 + g * 4);
        const int gi = 2 * wave + (g >> 1);
        const int goff = (gi & 1) * HBX + (gi >> 1) * 16 + (g & 1) * 8;
        char* yb = reinterpret_cast<char*>(y) + ((size_t)((n * H + y0) * W + x0 + r16) * 64 + wave * 16 + g * 4) * 2;
#pragma unroll
        for (int b = 0; b < F2; ++b) {
            const uint2 idn = *reinterpret_cast<const uint2*>(xp + ((b + 2) * XW + r16 + 2) * PITCH + goff);
            float v0 = acc2[b][0] + bv.x + h16lo(idn.x), v1 = acc2[b][1] + bv.y + h16hi(idn.x);
            float v2 = acc2[b][2] + bv.z + h16lo(idn.y), v3 = acc2[b][3] + bv.w + h16hi(idn.y);
            v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); v2 = fmaxf(v2, 0.f); v3 = fmaxf(v3, 0.f);
            *reinterpret_cast<uint2*>(yb + (size_t)b * W * 128) = make_uint2(pack_bf2(v0, v1), pack_bf2(v2, v3));
        }
    }
}

}  // namespace

// 1 if gdrn_block64_eval covers the shape: H a multiple of 8, W of 16 (ResNet-34's layer1 at 256 x 256 RoIs: 64 x 64 maps)
extern "C" int gdrn_block64_eval_ok(int N, int H, int W, int dtype) {
    return (dtype == GDRN_DT_H16 && N > 0 && H > 0 && W > 0 && (H % TH) == 0 && (W % TW) == 0 && (long long)N * H * W * 128 < (1ll << 32)) ? 1 : 0;
}

// x, y: [N][H][W][64] NHWC 16-bit (y != x); w1, w2: the FRAGMENT-MAJOR operands gdrn_pack_wfrag makes of the row-major [64][9][64] weights
// (eval mode: with the BatchNorm scale folded in); b1, b2: fp32 [64] (the folded BatchNorm shifts).
extern "C" int gdrn_block64_eval(const void* x, const void* w1, const float* b1, const void* w2, const float* b2, void* y, int N, int H, int W,
                                 int dtype, void* stream) {
    if (!x || !w1 || !b1 || !w2 || !b2 || !y || x == y) return GDRN_ERR_ARG;
    if (dtype != GDRN_DT_H16) return GDRN_ERR_ARG;
    if (!gdrn_block64_eval_ok(N, H, W, dtype)) return GDRN_ERR_SHAPE;
    constexpr size_t smem = XBYTES + ABYTES;
    static std::once_flag once;
    static bool attr_ok = false;
    std::call_once(once, [] {
        attr_ok = hipFuncSetAttribute(reinterpret_cast<const void*>(&block64_eval_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) == hipSuccess;
    });
    if (!attr_ok) return GDRN_ERR_LAUNCH;
    const int grid = N * (H / TH) * (W / TW);
    GDRN_LAUNCH(block64_eval_kernel, dim3(grid), dim3(256), smem, reinterpret_cast<hipStream_t>(stream), (const bf16_t*)x, (const bf16_t*)w1, b1,
                (const bf16_t*)w2, b2, (bf16_t*)y, N, H, W);
    GDRN_CHECK_LAUNCH();
    return GDRN_OK;
}
