// Data gradient of the 3x3 STRIDE-2 pad-1 convolutions on MFMA for gfx950 (round 6), optionally with
//   * the data gradient of the block's 1x1 stride-2 shortcut conv added in the same launch (it only reaches the even / even input pixels), and
//   * the ReLU mask + the two BatchNorm-backward sums of the BatchNorm(+ReLU) whose output this gradient belongs to, in the epilogue
// (autograd backward of `layerN.0.conv1` + `layerN.0.downsample.0`, resnet_backbone.py:69-80 / torchvision BasicBlock, and of Patch-PnP's stride-2
// convs, conv_pnp_net.py:76-92; reference: engine.py:279).  Until now: the generic gather kernel in its transposed mode (conv_gemm.hip, 150-280
// TFLOP/s), the shortcut's gradient as a launch of its own that writes a tensor three quarters of which are zeros, and an `addend` pass over it.
//
// dX[iy][ix] = sum over the taps (ky, kx) with iy + 1 - ky and ix + 1 - kx even of W[:, :, ky, kx]^T dY[(iy + 1 - ky) / 2][(ix + 1 - kx) / 2]:
// the input pixels fall into four parity classes (iy & 1, ix & 1) with 1 / 2 / 2 / 4 taps, and every one of the nine (tap, class) products reads
// dY at (a + da, b + db), da, db in {0, 1}, where (a, b) = (iy >> 1, ix >> 1).  A workgroup (4 waves, wave w = 16 of the workgroup's 64 dX
// channels) owns a 4 x 16 block of (a, b) = an 8 x 32 tile of dX, stages the 5 x 17 dY pixels of one 128-byte channel chunk in LDS (the
// stride-1 halo kernel's layout: fragments of 16 consecutive pixels, PITCH 80, even / odd granule arrays) and keeps FOUR accumulator sets,
// one per class.  Per k-step: 4 shifted fragment sets (16 ds_read_b128) feed 36 MFMAs -- the four sets are the pipeline's stages (the next
// set's reads and weight blocks go out in front of the current set's MFMAs).  Weights: the fragment-major operand of gdrn_pack_wfrag of the
// data-gradient operand [Cin rows][9 taps, not flipped][Cout], streamed L2 -> VGPR.
// Maps 8 pixels wide (layer4.0, Patch-PnP's third conv, the head's ConvTranspose): the TW_ = 8 form -- a 4 x 8 block of (a, b) of TWO images per
// workgroup, lanes 0-7 of a fragment one image, lanes 8-15 the next, whose patch starts 8 mod 16 pixels further on (conv3x3s2.hip, Geo).
// FWD: the same sum IS the forward pass of nn.ConvTranspose2d(k = 3, s = 2, p = 1, output_padding = 1) (cdpn_rot_head_region.py:96-101: dy = its
// input, dx = its output, w = [out channels][9][in channels]); the FWD epilogue is a forward conv's -- BatchNorm statistics rows or bias + ReLU.
#include <algorithm>
#include <mutex>
#include <type_traits>
#include <utility>

#include "common.h"
#include "../../include/gdrn_hip.h"

namespace {

template <class F, int... Is>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, Is...>) { (f(std::integral_constant<int, Is>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }

constexpr int PITCH = 80;
__host__ __device__ constexpr int half_bytes(int ppix) { return (ppix * PITCH + 255) / 256 * 256; }
constexpr int TH = 4, BNC = 64;                             // (a, b) block rows; dX channels per workgroup
constexpr int FM = TH;
template <int TW_>
struct Geo {
    static constexpr int TW = TW_, NI = 16 / TW_;               // images per tile
    static constexpr int PW = TW + 1, IPIX = (TH + 1) * PW;     // dY patch of one image: 5 x 17 / 5 x 9
    static constexpr int IMG_OFF = NI == 1 ? 0 : (IPIX + 7) / 16 * 16 + 8, PPIX = IMG_OFF * (NI - 1) + IPIX;   // 8: 56, 101
    static constexpr int HB = half_bytes(PPIX), PBYTES = 2 * HB;
    static constexpr int I2PIX = TH * TW, IMG_OFF2 = NI == 1 ? 0 : (I2PIX + 7) / 16 * 16 + 8, P2PIX = IMG_OFF2 * (NI - 1) + I2PIX;   // the shortcut's output gradient: 4 x 16 / 2 x (4 x 8)
    static constexpr int HB2 = half_bytes(P2PIX), P2BYTES = 2 * HB2;
    static constexpr int NS1 = (NI * IPIX * 8 + 255) / 256, NS2 = NI * I2PIX * 8 / 256;   // patch granules per thread: 3 + 2
};
static_assert(Geo<8>::IMG_OFF % 16 == 8 && Geo<8>::IMG_OFF2 % 16 == 8, "second image: 8 mod 16 pixels behind the first");

__device__ __forceinline__ f32x4_t mma(uint4 a, uint4 b, f32x4_t c) {
    return GDRN_MFMA16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c);
}

// the four fragment sets (da, db) of a k-step and the (tap, class) products that read them; class = (iy & 1) * 2 + (ix & 1)
// set 0 = (da, db) = (0, 0): taps (ky, kx) = (1,1) (1,2) (2,1) (2,2) -> classes 0 1 2 3;  set 1 = (0, 1): (1,0) (2,0) -> 1 3;
// set 2 = (1, 0): (0,1) (0,2) -> 2 3;  set 3 = (1, 1): (0,0) -> 3
template <int SET> struct SetOf { static constexpr int da = SET >> 1, db = SET & 1, n = SET == 0 ? 4 : (SET == 3 ? 1 : 2); };
__host__ __device__ constexpr int set_tap(int set, int i) {
    return set == 0 ? (i == 0 ? 4 : (i == 1 ? 5 : (i == 2 ? 7 : 8))) : (set == 1 ? (i == 0 ? 3 : 6) : (set == 2 ? (i == 0 ? 1 : 2) : 0));
}
__host__ __device__ constexpr int set_cls(int set, int i) {
    return set == 0 ? i : (set == 1 ? (i == 0 ? 1 : 3) : (set == 2 ? (i == 0 ? 2 : 3) : 3));
}

// WD (r6b): the weight blocks of a whole chunk (18 KiB per wave = 72 VGPRs) stay in registers, each refilled right behind its last MFMA with the
// block the NEXT chunk needs there: a weight load has a chunk (72 MFMAs, ~1150 cycles) to arrive.  For the grids that give a CU one workgroup
// (<= 256 workgroups: the 8-wide maps), where the one-stage-ahead ring of the other form (4-16 MFMAs ahead, 30-110 ns against ~400 ns of L2
// latency) left every stage waiting.
template <bool DS, bool BNB, int TW_, bool FWD = false, bool WD = false>
__global__ __launch_bounds__(256, 2) void conv3x3s2_dgrad_kernel(const gdrn_s2d_params p) {
    static_assert(!FWD || (!DS && !BNB), "forward (ConvTranspose) epilogue: no shortcut, no BatchNorm-backward sums");
    using G = Geo<TW_>;
    constexpr int TW = G::TW, NI = G::NI, PW = G::PW, IPIX = G::IPIX, HB = G::HB, PBYTES = G::PBYTES, HB2 = G::HB2, NS1 = G::NS1, NS2 = G::NS2;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // two stages of (dY patch | the shortcut's patch)
    constexpr int STAGE = PBYTES + (DS ? G::P2BYTES : 0);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, r16 = lane & 15;
    __builtin_amdgcn_s_setprio(2);

    const int NTn = p.Cin / BNC;
    int bid = blockIdx.x;
    {
        const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = bid & 7, j = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
    }
    const int nt = bid % NTn, mt = bid / NTn;
    const int tiles_x = p.Wo / TW, tiles_y = p.Ho / TH;
    const int tx = mt % tiles_x, ty = (mt / tiles_x) % tiles_y, n = (mt / (tiles_x * tiles_y)) * NI;   // first image of the tile
    const int y0 = ty * TH, x0 = tx * TW;
    const int kch = p.Cout / 64;

    // this wave's 16 dX channels: weight block cb; operands with more than 64 rows are packed in two-fragment groups (gdrn_pack_wfrag: the blocks
    // of a 32-row group interleave in units of 4 rows), so a result lane (g, j) holds channel ch0 + j
    const int cb = nt * 4 + wave;
    const bool il = p.w_rows > 64;
    const int ch0 = il ? ((cb >> 1) * 32 + g * 8 + (cb & 1) * 4) : (cb * 16 + g * 4);
    const char* wl = reinterpret_cast<const char*>(p.w) + (size_t)lane * 16;
    auto wptr = [&](int kc, int tap, int ks) -> const uint4* {
        return reinterpret_cast<const uint4*>(wl + ((size_t)(((cb * 9 + tap) * kch + kc) * 2 + ks) << 10));
    };

    // ---- patch slot geometry: dY pixel (y0 + pa, x0 + pb), pa < 5, pb < 17 (zero outside the map); the shortcut's 4 x 16 pixels
    unsigned poff[NS1], poff2[NS2];
    int pdst[NS1], pdst2[NS2];
    unsigned pokm = 0;
#pragma unroll
    for (int i = 0; i < NS1; ++i) {
        const int id = i * 256 + tid, qq = id >> 3, sg = id & 7;
        const int img = NI == 1 ? 0 : qq / IPIX, q = qq - img * IPIX;
        const int pa = q / PW, pb = q - pa * PW;
        const int oy = y0 + pa, ox = x0 + pb;
        const bool in = qq < NI * IPIX, ok = in && oy < p.Ho && ox < p.Wo;
        poff[i] = (unsigned)(((n + min(img, NI - 1)) * p.Ho + min(oy, p.Ho - 1)) * p.Wo + min(ox, p.Wo - 1)) * (unsigned)p.dy_cs * 2u + sg * 16;
        pdst[i] = in ? ((img * G::IMG_OFF + q) * PITCH + (sg & 1) * HB + (sg >> 1) * 16) : -1;
        pokm |= ok ? (1u << i) : 0u;
    }
    if constexpr (DS) {
#pragma unroll
        for (int i = 0; i < NS2; ++i) {
            const int id = i * 256 + tid, qq = id >> 3, sg = id & 7;
            const int img = NI == 1 ? 0 : qq / G::I2PIX, q = qq - img * G::I2PIX;
            poff2[i] = (unsigned)(((n + img) * p.Ho + y0 + q / TW) * p.Wo + x0 + (q % TW)) * (unsigned)p.dyd_cs * 2u + sg * 16;
            pdst2[i] = (img * G::IMG_OFF2 + q) * PITCH + (sg & 1) * HB2 + (sg >> 1) * 16;
        }
    }
    const char* yg = reinterpret_cast<const char*>(p.dy);
    const char* yg2 = reinterpret_cast<const char*>(p.dyd);
    static_assert(NS2 == 2, "the shortcut's patch: two granules per thread");
    uint4 pv[NS1], pv2a = make_uint4(0, 0, 0, 0), pv2b = pv2a;   // (named registers: as an array hipcc kept the shortcut's pair in scratch -- a store right behind each load)
#define LOADP(kc_)                                                                                             \
    {                                                                                                          \
        _Pragma("unroll") for (int i = 0; i < NS1; ++i) {                                                      \
            const uint4 t_ = *reinterpret_cast<const uint4*>(yg + (poff[i] + (unsigned)((kc_) * 128)));        \
            pv[i] = ((pokm >> i) & 1u) ? t_ : make_uint4(0, 0, 0, 0);                                          \
        }                                                                                                      \
        if constexpr (DS) {                                                                                    \
            pv2a = *reinterpret_cast<const uint4*>(yg2 + (poff2[0] + (unsigned)((kc_) * 128)));                  \
            pv2b = *reinterpret_cast<const uint4*>(yg2 + (poff2[1] + (unsigned)((kc_) * 128)));                  \
        }                                                                                                      \
    }
#define WRITEP(buf_)                                                                                           \
    {                                                                                                          \
        unsigned char* sb_ = smem + (buf_) * STAGE;                                                            \
        _Pragma("unroll") for (int i = 0; i < NS1; ++i)                                                        \
            if (pdst[i] >= 0) *reinterpret_cast<uint4*>(sb_ + pdst[i]) = pv[i];                                \
        if constexpr (DS) {                                                                                    \
            *reinterpret_cast<uint4*>(sb_ + PBYTES + pdst2[0]) = pv2a;                                           \
            *reinterpret_cast<uint4*>(sb_ + PBYTES + pdst2[1]) = pv2b;                                           \
        }                                                                                                      \
    }
    // Patch pipeline (r6b): two LDS stages, ONE register set.  At the top of chunk kc the registers hold chunk kc + 1 (requested a whole chunk
    // ago); they go to the idle stage, then take chunk kc + 2's loads -- a chunk's global loads have a full chunk of MFMAs to land under, and
    // there is one barrier per chunk.  (First version: one stage, loads at the top of the chunk, barrier - write - barrier behind it: with one
    // workgroup per CU -- the 256-workgroup grids of the 8-wide maps -- every chunk waited for its loads: 220-290 TFLOP/s.)
    LOADP(0)
    WRITEP(0)
    if (kch > 1) { LOADP(1) }

    // shortcut operand (row-major [rows][Cout]): this lane's row of the wave's fragment = the channel the packed operand has there
    const char* wdl = nullptr;
    if constexpr (DS) {
        const int row = il ? ((cb >> 1) * 32 + (r16 >> 2) * 8 + (cb & 1) * 4 + (r16 & 3)) : (cb * 16 + r16);
        wdl = reinterpret_cast<const char*>(p.wdd) + ((size_t)row * p.Cout + g * 8) * 2;
    }

    const int limg = r16 / TW, lcol = r16 % TW;   // image of the tile / column of the (a, b) block this lane's fragment pixel belongs to
    const int lb = (limg * G::IMG_OFF + lcol) * PITCH + (g & 1) * HB + (g >> 1) * 16;
    const int lb2 = (limg * G::IMG_OFF2 + lcol) * PITCH + (g & 1) * HB2 + (g >> 1) * 16;
    f32x4_t acc[4][FM];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int b = 0; b < FM; ++b) acc[c][b] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    __syncthreads();

    // stage s of a chunk = (k-step s >> 2, fragment set s & 3); two fragment buffers, two weight buffers (4 blocks each) -- or (WD) the chunk's 18 blocks
    uint4 fA[FM], fB[FM], wA[4], wB[4], wc[WD ? 18 : 1];
    constexpr int SETBASE[4] = {0, 4, 6, 8};   // first block of fragment set 0..3 inside a k-step's nine
    const unsigned char* scur = smem;   // the stage the current chunk reads
    auto rd = [&](uint4 (&dst)[FM], auto S_) {
        constexpr int s_ = decltype(S_)::value, ks = s_ >> 2;
        using Q = SetOf<(s_ & 3)>;
#pragma unroll
        for (int b = 0; b < FM; ++b) dst[b] = *reinterpret_cast<const uint4*>(scur + lb + ((b + Q::da) * PW + Q::db) * PITCH + ks * 32);
    };
    auto ldw = [&](uint4 (&dst)[4], int kc, auto S_) {
        constexpr int s_ = decltype(S_)::value, ks = s_ >> 2;
        using Q = SetOf<(s_ & 3)>;
#pragma unroll
        for (int i = 0; i < Q::n; ++i) dst[i] = *wptr(kc, set_tap(s_ & 3, i), ks);
    };
    if constexpr (WD) {
        static_for<8>([&](auto S_) {
            constexpr int s_ = decltype(S_)::value, ks = s_ >> 2, set = s_ & 3;
#pragma unroll
            for (int i = 0; i < SetOf<set>::n; ++i) wc[ks * 9 + SETBASE[set] + i] = *wptr(0, set_tap(set, i), ks);
        });
    } else {
        ldw(wA, 0, std::integral_constant<int, 0>{});
    }
    for (int kc = 0; kc < kch; ++kc) {
        const bool more = kc + 1 < kch;
        scur = smem + (kc & 1) * STAGE;
        if (more) {
            WRITEP((kc & 1) ^ 1)
            if (kc + 2 < kch) { LOADP(kc + 2) }
        }
        uint4 wdq[DS ? 2 : 1];
        if constexpr (DS) {
            wdq[0] = *reinterpret_cast<const uint4*>(wdl + (size_t)(kc * 64) * 2);
            wdq[1] = *reinterpret_cast<const uint4*>(wdl + (size_t)(kc * 64 + 32) * 2);
        }
        rd(fA, std::integral_constant<int, 0>{});
        static_for<8>([&](auto S_) {
            constexpr int s_ = decltype(S_)::value, ks = s_ >> 2;
            using Q = SetOf<(s_ & 3)>;
            uint4 (&src)[FM] = (s_ % 2 == 0) ? fA : fB;
            uint4 (&wq)[4] = (s_ % 2 == 0) ? wA : wB;
            if constexpr (s_ + 1 < 8) {
                if constexpr (s_ % 2 == 0) { rd(fB, std::integral_constant<int, s_ + 1>{}); if constexpr (!WD) ldw(wB, kc, std::integral_constant<int, s_ + 1>{}); }
                else { rd(fA, std::integral_constant<int, s_ + 1>{}); if constexpr (!WD) ldw(wA, kc, std::integral_constant<int, s_ + 1>{}); }
            } else if constexpr (!WD) {
                if (more) ldw(wA, kc + 1, std::integral_constant<int, 0>{});   // (stage 7 reads wB: wA is free for the next chunk's first stage)
            }
#pragma unroll
            for (int i = 0; i < Q::n; ++i)
#pragma unroll
                for (int b = 0; b < FM; ++b) {
                    if constexpr (WD) acc[set_cls(s_ & 3, i)][b] = mma(wc[ks * 9 + SETBASE[s_ & 3] + i], src[b], acc[set_cls(s_ & 3, i)][b]);
                    else acc[set_cls(s_ & 3, i)][b] = mma(wq[i], src[b], acc[set_cls(s_ & 3, i)][b]);
                }
            if constexpr (WD) {   // the stage's blocks are consumed: the next chunk's go into the same registers
                if (more) {
#pragma unroll
                    for (int i = 0; i < Q::n; ++i) wc[ks * 9 + SETBASE[s_ & 3] + i] = *wptr(kc + 1, set_tap(s_ & 3, i), ks);
                }
            }
            if constexpr (DS && (s_ & 3) == 3) {   // the 1x1 shortcut's gradient: even / even pixels only, its own (unshifted) fragments
#pragma unroll
                for (int b = 0; b < FM; ++b) {
                    const uint4 f2 = *reinterpret_cast<const uint4*>(scur + PBYTES + lb2 + b * TW * PITCH + ks * 32);
                    acc[0][b] = mma(wdq[ks], f2, acc[0][b]);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        });
        if (more) __syncthreads();   // this stage's reads are done, the other stage's writes visible
    }
#undef LOADP
#undef WRITEP

    // ---- epilogue: lane = channels ch0 .. ch0 + 3 of input pixel (2 (y0 + b) + py, 2 (x0 + lcol) + px) of image n + limg for class (py, px), fragment row b
    char* xo = reinterpret_cast<char*>(p.dx);
    float t1[4] = {0.f, 0.f, 0.f, 0.f}, t2[4] = {0.f, 0.f, 0.f, 0.f};
    float kmu[4] = {0.f, 0.f, 0.f, 0.f}, kis[4] = {0.f, 0.f, 0.f, 0.f};
    if constexpr (BNB) {
        const float4 mu = *reinterpret_cast<const float4*>(p.bnb_mean + ch0), is = *reinterpret_cast<const float4*>(p.bnb_invstd + ch0);
        kmu[0] = mu.x; kmu[1] = mu.y; kmu[2] = mu.z; kmu[3] = mu.w;
        kis[0] = is.x; kis[1] = is.y; kis[2] = is.z; kis[3] = is.w;
    }
    if constexpr (FWD) {
        // forward conv epilogue (ConvTranspose2d): per-tile BatchNorm statistics rows of the raw sums (train mode) / bias + ReLU (eval mode, folded BatchNorm)
        if (p.stats != nullptr) {
            float s1[4], s2[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float u = 0.f, q = 0.f;
#pragma unroll
                for (int c = 0; c < 4; ++c)
#pragma unroll
                    for (int b = 0; b < FM; ++b) { const float v = acc[c][b][j]; u += v; q += v * v; }
                s1[j] = row16_sum(u);
                s2[j] = row16_sum(q);
            }
            if (r16 == 0) {
                float* srow = p.stats + (size_t)mt * 2 * p.Cin + ch0;
                *reinterpret_cast<float4*>(srow) = make_float4(s1[0], s1[1], s1[2], s1[3]);
                *reinterpret_cast<float4*>(srow + p.Cin) = make_float4(s2[0], s2[1], s2[2], s2[3]);
            }
        }
        float bq[4] = {0.f, 0.f, 0.f, 0.f};
        if (p.bias != nullptr) { const float4 bv = *reinterpret_cast<const float4*>(p.bias + ch0); bq[0] = bv.x; bq[1] = bv.y; bq[2] = bv.z; bq[3] = bv.w; }
        const bool relu = p.act == 1;
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int b = 0; b < FM; ++b) {
                const unsigned pix = (unsigned)(((n + limg) * p.Hi + 2 * (y0 + b) + (c >> 1)) * p.Wi + 2 * (x0 + lcol) + (c & 1));
                float v[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) { v[j] = acc[c][b][j] + bq[j]; if (relu) v[j] = fmaxf(v[j], 0.f); }
                *reinterpret_cast<uint2*>(xo + ((size_t)pix * p.dx_cs + ch0) * 2) = make_uint2(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]));
            }
        return;
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int py = c >> 1, px = c & 1;
        unsigned pix[FM];
#pragma unroll
        for (int b = 0; b < FM; ++b) pix[b] = (unsigned)(((n + limg) * p.Hi + 2 * (y0 + b) + py) * p.Wi + 2 * (x0 + lcol) + px);
        if constexpr (BNB) {
            // data gradient w.r.t. a BatchNorm(+ReLU)'s output: ReLU mask (stored activation > 0) and that BatchNorm's backward sums here, on
            // the accumulators (conv3x3_halo.hip's bnb epilogue): all loads of the class first, then arithmetic and stores
            const char* xb = reinterpret_cast<const char*>(p.bnb_x);
            const char* mb = reinterpret_cast<const char*>(p.bnb_mask);
            uint2 xq[FM], mq[FM];
#pragma unroll
            for (int b = 0; b < FM; ++b) {
                xq[b] = *reinterpret_cast<const uint2*>(xb + ((size_t)pix[b] * p.bnb_cs + ch0) * 2);
                mq[b] = *reinterpret_cast<const uint2*>(mb + ((size_t)pix[b] * p.bnb_cs + ch0) * 2);
            }
#pragma unroll
            for (int b = 0; b < FM; ++b) {
                const float xv[4] = {h16lo(xq[b].x), h16hi(xq[b].x), h16lo(xq[b].y), h16hi(xq[b].y)};
                const float mv[4] = {h16lo(mq[b].x), h16hi(mq[b].x), h16lo(mq[b].y), h16hi(mq[b].y)};
                float v[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float gv = (mv[j] > 0.f) ? acc[c][b][j] : 0.f;
                    v[j] = gv;
                    t1[j] += gv;
                    t2[j] += gv * (xv[j] - kmu[j]) * kis[j];
                }
                *reinterpret_cast<uint2*>(xo + ((size_t)pix[b] * p.dx_cs + ch0) * 2) = make_uint2(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]));
            }
        } else {
#pragma unroll
            for (int b = 0; b < FM; ++b)
                *reinterpret_cast<uint2*>(xo + ((size_t)pix[b] * p.dx_cs + ch0) * 2) =
                    make_uint2(pack_bf2(acc[c][b][0], acc[c][b][1]), pack_bf2(acc[c][b][2], acc[c][b][3]));
        }
    }
    if constexpr (BNB) {
        float u1[4], u2[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) { u1[j] = row16_sum(t1[j]); u2[j] = row16_sum(t2[j]); }
        if (r16 == 0) {
            float* srow = p.bnb_rows + (size_t)mt * 2 * p.Cin + ch0;
            *reinterpret_cast<float4*>(srow) = make_float4(u1[0], u1[1], u1[2], u1[3]);
            *reinterpret_cast<float4*>(srow + p.Cin) = make_float4(u2[0], u2[1], u2[2], u2[3]);
        }
    }
}

}  // namespace

// (a, b)-block width for a shape: 16, 8 (8-wide maps, two images per tile: an even image count), 0: not covered
static int s2d_tw(const gdrn_s2d_params* p) {
    if (!p || p->Wo <= 0) return 0;
    if (p->Wo % 16 == 0) return 16;
    if (p->Wo % 8 == 0 && (p->N % 2) == 0) return 8;
    return 0;
}

extern "C" int gdrn_conv3x3s2_dgrad_ok(const gdrn_s2d_params* p) {
    if (!p) return 0;
    if (p->dtype != GDRN_DT_H16 || p->N <= 0 || p->Hi != 2 * p->Ho || p->Wi != 2 * p->Wo || (p->Ho % TH) || s2d_tw(p) == 0) return 0;
    if (p->Cin <= 0 || (p->Cin % BNC) || p->Cout <= 0 || (p->Cout % 64) || p->w_rows < p->Cin || (p->w_rows % 16)) return 0;
    if (p->w_rows > 64 && (p->w_rows % 32)) return 0;
    if ((p->dy_cs & 7) || p->dy_cs < p->Cout || (p->dx_cs & 3) || p->dx_cs < p->Cin) return 0;
    if (p->dyd && ((p->dyd_cs & 7) || p->dyd_cs < p->Cout || p->wdd_rows < p->Cin)) return 0;
    if (p->bnb_x && ((p->bnb_cs & 3) || p->bnb_cs < p->Cin)) return 0;
    if (p->act < 0 || p->act > 1) return 0;
    if ((p->stats || p->bias || p->act) && (p->dyd || p->bnb_x)) return 0;   // forward (ConvTranspose) epilogue: a plain launch
    if ((unsigned long long)p->N * p->Ho * p->Wo * std::max(p->dy_cs, p->dyd_cs) * 2ull >= (1ull << 32)) return 0;
    if ((unsigned long long)p->N * p->Hi * p->Wi >= (1ull << 31)) return 0;
    return 1;
}

extern "C" int gdrn_conv3x3s2_dgrad_rows(const gdrn_s2d_params* p) {
    if (!gdrn_conv3x3s2_dgrad_ok(p)) return GDRN_ERR_SHAPE;
    const int tw = s2d_tw(p);
    return (p->N / (16 / tw)) * (p->Ho / TH) * (p->Wo / tw);
}

namespace {
template <int TW_>
int launch_s2d(const gdrn_s2d_params& p, hipStream_t st) {
    using G = Geo<TW_>;
    const size_t smem = 2 * (size_t)(G::PBYTES + (p.dyd ? G::P2BYTES : 0));
    const int grid = (p.N / G::NI) * (p.Ho / TH) * (p.Wo / TW_) * (p.Cin / BNC);
    const bool wd = p.dyd != nullptr || grid <= 1024;   // (measured: the plain form with > 1024 workgroups -- three per CU -- is the one launch it loses on: 44 vs 42 us)
    const int v = (p.stats || p.bias || p.act) ? 4 : ((p.dyd ? 2 : 0) | (p.bnb_x ? 1 : 0));
#define S2D_LAUNCH(DS_, BNB_, FWD_)                                                                                            \
    {                                                                                                                          \
        if (wd) GDRN_LAUNCH((conv3x3s2_dgrad_kernel<DS_, BNB_, TW_, FWD_, true>), dim3(grid), dim3(256), smem, st, p);         \
        else GDRN_LAUNCH((conv3x3s2_dgrad_kernel<DS_, BNB_, TW_, FWD_, false>), dim3(grid), dim3(256), smem, st, p);           \
    }
    switch (v) {
        case 0: S2D_LAUNCH(false, false, false) break;
        case 1: S2D_LAUNCH(false, true, false) break;
        case 2: S2D_LAUNCH(true, false, false) break;
        case 3: S2D_LAUNCH(true, true, false) break;
        default: S2D_LAUNCH(false, false, true) break;
    }
#undef S2D_LAUNCH
    GDRN_CHECK_LAUNCH();
    return GDRN_OK;
}
}  // namespace

extern "C" int gdrn_conv3x3s2_dgrad(const gdrn_s2d_params* pp, void* stream) {
    if (!pp || !pp->dy || !pp->w || !pp->dx) return GDRN_ERR_ARG;
    if (pp->dtype != GDRN_DT_H16) return GDRN_ERR_ARG;
    if ((pp->dyd != nullptr) != (pp->wdd != nullptr)) return GDRN_ERR_ARG;
    if (pp->bnb_x && (!pp->bnb_mask || !pp->bnb_mean || !pp->bnb_invstd || !pp->bnb_rows)) return GDRN_ERR_ARG;
    if (!gdrn_conv3x3s2_dgrad_ok(pp)) return GDRN_ERR_SHAPE;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    return s2d_tw(pp) == 16 ? launch_s2d<16>(*pp, st) : launch_s2d<8>(*pp, st);
}
