// Halo-tiled 3x3 STRIDE-2 pad-1 convolution (forward) on MFMA for gfx950, optionally with the block's 1x1 stride-2 shortcut conv in the same
// launch (round 6): the three stage-entry convs of ResNet-34 (`layerN.0.conv1` + `layerN.0.downsample.0`, resnet_backbone.py:69-80 / torchvision
// BasicBlock with stride 2) and Patch-PnP's three stride-2 convs (conv_pnp_net.py:76-92); 8-wide maps two images to a pixel tile (Geo).  Until now these ran on
// the generic gather kernel (conv_gemm.hip) at 240-430 TFLOP/s, the shortcut as a launch of its own (7-9 us for 1 GFLOP).
//
// conv3x3_halo.hip's scheme with one change of geometry.  A workgroup (4 waves) owns 4 x 16 output pixels x 128 output channels (wave w:
// channels [32w, 32w + 32) for all pixels, two 16-channel weight fragments -- the fragment-major operand of gdrn_pack_wfrag streamed L2 -> VGPR
// through a three-tap register ring).  Its 9 x 33 input pixels of one 128-byte channel chunk live in LDS DE-INTERLEAVED BY PARITY: four planes
// (even / odd input row) x (even / odd input column) of 5 x 17, 5 x 16, 4 x 17, 4 x 16 pixels.  Tap (ky, kx) of output pixel (oy, ox) reads
// input (2 oy + ky - 1, 2 ox + kx - 1) = plane (ky & 1, kx & 1), row oy + (ky >> 1), column ox + (kx >> 1): a fragment of 16 output pixels of
// one output row is 16 CONSECUTIVE pixels of one plane row -- the stride-1 kernel's conflict-free ds_read_b128 pattern (PITCH 80, even / odd
// granule arrays), every tap an immediate on one lane base.  The 1x1 shortcut reads exactly the centre tap's fragments (input (2 oy, 2 ox)):
// eight more MFMAs per k-step into a second accumulator set, its row-major operand gathered from global memory (2 KiB per wave and chunk).
// Epilogue = the halo kernel's fast path: per-tile BatchNorm statistics rows (train mode), or bias + ReLU (eval mode: folded BatchNorm).
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
constexpr int TH = 4, BN = 128;
constexpr int FM = TH, FN = 2;
// Geometry of a pixel tile.  TW_ = 16: 4 x 16 output pixels of one image.  TW_ = 8 (maps 8 pixels wide: layer4.0, Patch-PnP's third conv): 4 x 8
// output pixels of TWO images side by side -- a fragment's lanes 0-7 are a row of image 2 m, lanes 8-15 the same row of image 2 m + 1, whose
// planes start IMG_OFF pixels further on.  IMG_OFF = 8 mod 16 pixels keeps the ds_read_b128 pattern conflict-free: the second half-row then
// lands on the bank slots pixels 8-15 of a 16-pixel row would have (PITCH 80 B: 16 consecutive pixels cover the 64 banks once).
template <int TW_>
struct Geo {
    static constexpr int TW = TW_, NI = 16 / TW_;                  // images per tile
    static constexpr int PE = TW + 1, PO = TW;                     // plane widths: even / odd input columns
    static constexpr int B_EE = 0, B_EO = (TH + 1) * PE, B_OE = B_EO + (TH + 1) * PO, B_OO = B_OE + TH * PE, IPIX = B_OO + TH * PO;   // 16: 0, 85, 165, 233, 297
    static constexpr int IMG_OFF = NI == 1 ? 0 : (IPIX + 7) / 16 * 16 + 8;                                                            //  8: 0, 45, 85, 121, 153 -> 168
    static constexpr int PPIX = IMG_OFF * (NI - 1) + IPIX;
    static constexpr int HB = half_bytes(PPIX), PBYTES = 2 * HB;   // 47616 B / 51456 B
    static constexpr int NSLOT = (NI * IPIX * 8 + 255) / 256;      // patch granules per thread and chunk (10)
    // LDS pixel index of tap TAP for output row b, lane column 0: plane base + (b + (ky >> 1)) * plane width + (kx >> 1)
    template <int TAP>
    static __host__ __device__ constexpr int tap_pix(int b) {
        constexpr int ky = TAP / 3, kx = TAP % 3;
        constexpr int base = (ky & 1) ? ((kx & 1) ? B_OO : B_OE) : ((kx & 1) ? B_EO : B_EE);
        constexpr int pw = (kx & 1) ? PO : PE;
        return base + (b + (ky >> 1)) * pw + (kx >> 1);
    }
};
static_assert(Geo<8>::IMG_OFF % 16 == 8 && Geo<8>::IMG_OFF >= Geo<8>::IPIX, "second image's planes: 8 mod 16 pixels behind the first's");

__device__ __forceinline__ f32x4_t mma(uint4 a, uint4 b, f32x4_t c) {
    return GDRN_MFMA16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c);
}

// BNB (r6b, with !DS): the launch is a DATA GRADIENT -- the head's ConvTranspose2d backward is a stride-2 conv of the output gradient -- w.r.t. the
// output of a BatchNorm(+ReLU): ReLU mask (stored activation > 0) and that BatchNorm's two backward sums on the accumulators, one partial row per
// pixel tile (conv3x3_halo.hip's bnb epilogue).
template <bool DS, int TW_, bool BNB = false>
__global__ __launch_bounds__(256, 2) void conv3x3s2_kernel(const gdrn_s2_params p) {
    static_assert(!(DS && BNB), "the BatchNorm-backward epilogue: no shortcut conv");
    using G = Geo<TW_>;
    constexpr int TW = G::TW, NI = G::NI, PE = G::PE, PO = G::PO, B_EO = G::B_EO, B_OE = G::B_OE, B_OO = G::B_OO, IPIX = G::IPIX, HB = G::HB, NSLOT = G::NSLOT;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, r16 = lane & 15;
    __builtin_amdgcn_s_setprio(2);

    const int NTn = p.Cout / BN;
    int bid = blockIdx.x;
    {   // XCD-aware order: neighbouring tiles / the channel tiles of a pixel tile share one L2
        const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = bid & 7, j = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
    }
    const int nt = bid % NTn, mt = bid / NTn;
    const int tiles_x = p.Wo / TW, tiles_y = p.Ho / TH;
    const int tx = mt % tiles_x, ty = (mt / tiles_x) % tiles_y, n = (mt / (tiles_x * tiles_y)) * NI;   // first image of the tile
    const int co0 = nt * BN, y0 = ty * TH, x0 = tx * TW;
    const int kch = p.Cin / 64;

    // ---- weight stream (fragment-major 3x3 operand): block ((cb * 9 + tap) * kch + kc) * 2 + ks, 1 KiB each; ring of three taps
    const int cb0 = (co0 + wave * 32) / 16;
    const char* wl = reinterpret_cast<const char*>(p.w) + (size_t)lane * 16;
    auto wptr = [&](int kc, int tap, int a, int ks) -> const uint4* {
        return reinterpret_cast<const uint4*>(wl + ((size_t)((((cb0 + a) * 9 + tap) * kch + kc) * 2 + ks) << 10));
    };
    uint4 wq0[4], wq1[4], wq2[4];
#define LOADW(dst, kc_, tap_)                                                                                  \
    { dst[0] = *wptr(kc_, tap_, 0, 0); dst[1] = *wptr(kc_, tap_, 0, 1); dst[2] = *wptr(kc_, tap_, 1, 0); dst[3] = *wptr(kc_, tap_, 1, 1); }
    LOADW(wq0, 0, 0) LOADW(wq1, 0, 1) LOADW(wq2, 0, 2)

    // ---- patch slot geometry of this thread: LDS pixel q of the parity planes <- input pixel (2 y0 - 1 + r, 2 x0 - 1 + c)
    unsigned poff[NSLOT];
    int pdst[NSLOT];
    unsigned pokm = 0;
#pragma unroll
    for (int i = 0; i < NSLOT; ++i) {
        const int id = i * 256 + tid, qq = id >> 3, sg = id & 7;
        const int img = NI == 1 ? 0 : qq / IPIX, q = qq - img * IPIX;   // image of the tile, pixel of its planes
        int r, c;
        if (q < B_EO) { r = 2 * (q / PE); c = 2 * (q % PE); }
        else if (q < B_OE) { const int t = q - B_EO; r = 2 * (t / PO); c = 2 * (t % PO) + 1; }
        else if (q < B_OO) { const int t = q - B_OE; r = 2 * (t / PE) + 1; c = 2 * (t % PE); }
        else { const int t = q - B_OO; r = 2 * (t / PO) + 1; c = 2 * (t % PO) + 1; }
        const int iy = 2 * y0 - 1 + r, ix = 2 * x0 - 1 + c;
        const bool in = qq < NI * IPIX;
        const bool ok = in && iy >= 0 && iy < p.Hi && ix >= 0 && ix < p.Wi;
        const int iyc = min(max(iy, 0), p.Hi - 1), ixc = min(max(ix, 0), p.Wi - 1);
        poff[i] = (unsigned)(((n + min(img, NI - 1)) * p.Hi + iyc) * p.Wi + ixc) * (unsigned)p.x_cs * 2u + sg * 16;
        pdst[i] = in ? ((img * G::IMG_OFF + q) * PITCH + (sg & 1) * HB + (sg >> 1) * 16) : -1;
        pokm |= ok ? (1u << i) : 0u;
    }
    const char* xg = reinterpret_cast<const char*>(p.x);
    uint4 pv[NSLOT];
#define LOADP(kc_)                                                                                             \
    _Pragma("unroll") for (int i = 0; i < NSLOT; ++i) {                                                        \
        const uint4 t_ = *reinterpret_cast<const uint4*>(xg + (poff[i] + (unsigned)((kc_) * 128)));            \
        pv[i] = ((pokm >> i) & 1u) ? t_ : make_uint4(0, 0, 0, 0);                                              \
    }
#define WRITEP()                                                                                               \
    _Pragma("unroll") for (int i = 0; i < NSLOT; ++i)                                                          \
        if (pdst[i] >= 0) *reinterpret_cast<uint4*>(smem + pdst[i]) = pv[i];
    LOADP(0)
    WRITEP()

    // shortcut operand: row-major [wd_rows][Cin]; fragment a, lane (rr = lane & 15, g): row co0 + 32 wave + (rr >> 2) * 8 + a * 4 + (rr & 3)
    // (the interleave of gdrn_pack_wfrag's two-fragment groups: a result lane then holds the same 8 contiguous channels for both convs)
    const char* wdl = nullptr;
    if constexpr (DS) wdl = reinterpret_cast<const char*>(p.wd) + ((size_t)(co0 + wave * 32 + (r16 >> 2) * 8 + (r16 & 3)) * p.Cin + g * 8) * 2;

    const int lb = ((r16 / TW) * G::IMG_OFF + (r16 % TW)) * PITCH + (g & 1) * HB + (g >> 1) * 16;
    f32x4_t acc[FN][FM], accd[DS ? FN : 1][DS ? FM : 1];
#pragma unroll
    for (int a = 0; a < FN; ++a)
#pragma unroll
        for (int b = 0; b < FM; ++b) {
            acc[a][b] = f32x4_t{0.f, 0.f, 0.f, 0.f};
            if constexpr (DS) accd[a][b] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        }
    __syncthreads();

    uint4 fA[FM], fB[FM];
    auto rd = [&](uint4 (&dst)[FM], auto S_) {
        constexpr int s_ = decltype(S_)::value, tap = s_ / 2, ks = s_ % 2;
#pragma unroll
        for (int b = 0; b < FM; ++b) dst[b] = *reinterpret_cast<const uint4*>(smem + lb + G::template tap_pix<tap>(b) * PITCH + ks * 32);
    };
    for (int kc = 0; kc < kch; ++kc) {
        const bool more = kc + 1 < kch;
        if (more) { LOADP(kc + 1) }    // the next chunk's patch travels under this chunk's MFMAs
        uint4 wdq[DS ? 4 : 1];
        if constexpr (DS) {
#pragma unroll
            for (int a = 0; a < FN; ++a)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
                    wdq[a * 2 + ks] = *reinterpret_cast<const uint4*>(wdl + ((size_t)(a * 4) * p.Cin + kc * 64 + ks * 32) * 2);
        }
        rd(fA, std::integral_constant<int, 0>{});
        // 18 stages (tap, k-step) of 8 MFMAs (+ 8 of the shortcut on the centre tap); the fragment reads of stage s + 1 in front of them
        static_for<18>([&](auto S_) {
            constexpr int s_ = decltype(S_)::value, tap = s_ / 2, ks = s_ % 2;
            if constexpr (s_ + 1 < 18) {
                if constexpr (s_ % 2 == 0) rd(fB, std::integral_constant<int, s_ + 1>{});
                else rd(fA, std::integral_constant<int, s_ + 1>{});
            }
            uint4 (&src)[FM] = (s_ % 2 == 0) ? fA : fB;
            uint4 (&wq)[4] = (tap % 3 == 0) ? wq0 : ((tap % 3 == 1) ? wq1 : wq2);
#pragma unroll
            for (int a = 0; a < FN; ++a)
#pragma unroll
                for (int b = 0; b < FM; ++b) acc[a][b] = mma(wq[a * 2 + ks], src[b], acc[a][b]);
            if constexpr (DS && tap == 4) {
#pragma unroll
                for (int a = 0; a < FN; ++a)
#pragma unroll
                    for (int b = 0; b < FM; ++b) accd[a][b] = mma(wdq[a * 2 + ks], src[b], accd[a][b]);
            }
            if constexpr (ks == 1) {   // the tap is done: its ring slot takes the weights three taps ahead (the next chunk's behind tap 5)
                constexpr int ntap = (tap + 3) % 9;
                const int nk = kc + ((tap + 3) >= 9 ? 1 : 0);
                if (nk < kch) LOADW(wq, nk, ntap)
            }
            __builtin_amdgcn_sched_barrier(0);
        });
        if (more) {
            __syncthreads();
            WRITEP()
            __syncthreads();
        }
    }
#undef LOADW
#undef LOADP
#undef WRITEP

    // ---- epilogue: lane holds channels cl .. cl + 7 (cl = co0 + 32 wave + 8 g; fragment a = channels + 4 a) of pixel (y0 + b, x0 + r16 % TW) of image n + r16 / TW
    const int cl = co0 + wave * 32 + g * 8;
    const int prow0 = ((n + r16 / TW) * p.Ho + y0) * p.Wo + x0 + (r16 % TW);
    auto finish = [&](f32x4_t (&A)[FN][FM], float* stats, const float* bias, char* yb, int y_cs, bool relu) {
        if (stats != nullptr) {
            float* srow = stats + (size_t)mt * 2 * p.Cout + cl;
#pragma unroll
            for (int a = 0; a < FN; ++a) {
                float s1[4], s2[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float u = 0.f, q = 0.f;
#pragma unroll
                    for (int b = 0; b < FM; ++b) { const float v = A[a][b][j]; u += v; q += v * v; }
                    s1[j] = row16_sum(u);
                    s2[j] = row16_sum(q);
                }
                if (r16 == 0) {
                    *reinterpret_cast<float4*>(srow + a * 4) = make_float4(s1[0], s1[1], s1[2], s1[3]);
                    *reinterpret_cast<float4*>(srow + p.Cout + a * 4) = make_float4(s2[0], s2[1], s2[2], s2[3]);
                }
            }
        }
        float bq[FN][4];
#pragma unroll
        for (int a = 0; a < FN; ++a) {
            float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
            if (bias != nullptr) bv = *reinterpret_cast<const float4*>(bias + cl + a * 4);
            bq[a][0] = bv.x; bq[a][1] = bv.y; bq[a][2] = bv.z; bq[a][3] = bv.w;
        }
#pragma unroll
        for (int b = 0; b < FM; ++b) {
            uint32_t o[4];
#pragma unroll
            for (int a = 0; a < FN; ++a) {
                float v0 = A[a][b][0] + bq[a][0], v1 = A[a][b][1] + bq[a][1], v2 = A[a][b][2] + bq[a][2], v3 = A[a][b][3] + bq[a][3];
                if (relu) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); v2 = fmaxf(v2, 0.f); v3 = fmaxf(v3, 0.f); }
                o[a * 2] = pack_bf2(v0, v1);
                o[a * 2 + 1] = pack_bf2(v2, v3);
            }
            *reinterpret_cast<uint4*>(yb + ((size_t)(prow0 + b * p.Wo) * y_cs + cl) * 2) = make_uint4(o[0], o[1], o[2], o[3]);
        }
    };
    if constexpr (BNB) {
        const char* xb = reinterpret_cast<const char*>(p.bnb_x);
        const char* mb = reinterpret_cast<const char*>(p.bnb_mask);
        char* yb = reinterpret_cast<char*>(p.y);
        float kmu[8], kis[8], t1[8], t2[8];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const float4 mu = *reinterpret_cast<const float4*>(p.bnb_mean + cl + 4 * h), is = *reinterpret_cast<const float4*>(p.bnb_invstd + cl + 4 * h);
            kmu[4 * h] = mu.x; kmu[4 * h + 1] = mu.y; kmu[4 * h + 2] = mu.z; kmu[4 * h + 3] = mu.w;
            kis[4 * h] = is.x; kis[4 * h + 1] = is.y; kis[4 * h + 2] = is.z; kis[4 * h + 3] = is.w;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) { t1[j] = 0.f; t2[j] = 0.f; }
        uint4 xq[FM], mq[FM];   // all loads of the tile first, then arithmetic and stores
#pragma unroll
        for (int b = 0; b < FM; ++b) {
            const size_t o = ((size_t)(prow0 + b * p.Wo) * p.bnb_cs + cl) * 2;
            xq[b] = *reinterpret_cast<const uint4*>(xb + o);
            mq[b] = *reinterpret_cast<const uint4*>(mb + o);
        }
#pragma unroll
        for (int b = 0; b < FM; ++b) {
            float xv[8], mv[8], v[8];
            Vec16<bf16_t>::unpack(xq[b], xv);
            Vec16<bf16_t>::unpack(mq[b], mv);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float gv = (mv[j] > 0.f) ? acc[j >> 2][b][j & 3] : 0.f;
                v[j] = gv;
                t1[j] += gv;
                t2[j] += gv * (xv[j] - kmu[j]) * kis[j];
            }
            *reinterpret_cast<uint4*>(yb + ((size_t)(prow0 + b * p.Wo) * p.y_cs + cl) * 2) = Vec16<bf16_t>::pack(v);
        }
        float u1[8], u2[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { u1[j] = row16_sum(t1[j]); u2[j] = row16_sum(t2[j]); }
        if (r16 == 0) {
            float* srow = p.bnb_rows + (size_t)mt * 2 * p.Cout + cl;
            *reinterpret_cast<float4*>(srow) = make_float4(u1[0], u1[1], u1[2], u1[3]);
            *reinterpret_cast<float4*>(srow + 4) = make_float4(u1[4], u1[5], u1[6], u1[7]);
            *reinterpret_cast<float4*>(srow + p.Cout) = make_float4(u2[0], u2[1], u2[2], u2[3]);
            *reinterpret_cast<float4*>(srow + p.Cout + 4) = make_float4(u2[4], u2[5], u2[6], u2[7]);
        }
        return;
    }
    finish(acc, p.stats, p.bias, reinterpret_cast<char*>(p.y), p.y_cs, p.act == 1);
    if constexpr (DS) finish(accd, p.stats_d, p.bias_d, reinterpret_cast<char*>(p.yd), p.yd_cs, false);
}

}  // namespace

// pixel-tile width for a shape: 16 (maps a multiple of 16 wide), 8 (8-wide maps, two images per tile: an even image count), 0: not covered
static int s2_tw(const gdrn_s2_params* p) {
    if (!p || p->Wo <= 0) return 0;
    if (p->Wo % 16 == 0) return 16;
    if (p->Wo % 8 == 0 && (p->N % 2) == 0) return 8;
    return 0;
}

// 1 if gdrn_conv3x3s2 covers the shape
extern "C" int gdrn_conv3x3s2_ok(const gdrn_s2_params* p) {
    if (!p) return 0;
    const int tw = s2_tw(p);
    if (p->dtype != GDRN_DT_H16 || p->N <= 0 || p->Hi != 2 * p->Ho || p->Wi != 2 * p->Wo || (p->Ho % TH) || tw == 0) return 0;
    if (p->Cin <= 0 || (p->Cin % 64) || p->Cout <= 0 || (p->Cout % BN) || p->w_rows < p->Cout || (p->x_cs & 7) || p->x_cs < p->Cin) return 0;
    if ((p->y_cs & 7) || p->y_cs < p->Cout || p->act < 0 || p->act > 1) return 0;
    if (p->wd && ((p->yd_cs & 7) || p->yd_cs < p->Cout || p->wd_rows < p->Cout)) return 0;
    if ((unsigned long long)p->N * p->Hi * p->Wi * p->x_cs * 2ull >= (1ull << 32)) return 0;   // 32-bit byte offsets of the patch loads
    if (p->bnb_x && (p->wd || p->stats || p->bias || p->act || (p->bnb_cs & 7) || p->bnb_cs < p->Cout)) return 0;   // BatchNorm-backward epilogue: a plain launch
    return 1;
}

extern "C" int gdrn_conv3x3s2_stats_rows(const gdrn_s2_params* p) {
    if (!gdrn_conv3x3s2_ok(p)) return GDRN_ERR_SHAPE;
    const int tw = s2_tw(p);
    return (p->N / (16 / tw)) * (p->Ho / TH) * (p->Wo / tw);
}

namespace {
template <bool DS, int TW_, bool BNB = false>
int launch_s2(const gdrn_s2_params& p, hipStream_t st) {
    constexpr size_t smem = Geo<TW_>::PBYTES;
    static std::once_flag once;
    static bool attr_ok = false;
    std::call_once(once, [] {
        attr_ok = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3s2_kernel<DS, TW_, BNB>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) == hipSuccess;
    });
    if (!attr_ok) return GDRN_ERR_LAUNCH;
    const int grid = (p.N / Geo<TW_>::NI) * (p.Ho / TH) * (p.Wo / TW_) * (p.Cout / BN);
    GDRN_LAUNCH((conv3x3s2_kernel<DS, TW_, BNB>), dim3(grid), dim3(256), smem, st, p);
    GDRN_CHECK_LAUNCH();
    return GDRN_OK;
}
}  // namespace

// w: the FRAGMENT-MAJOR operand gdrn_pack_wfrag makes of the row-major [w_rows][9][Cin] forward weights; wd (optional): the ROW-MAJOR
// [wd_rows][Cin] operand of the block's 1x1 stride-2 shortcut, evaluated in the same launch into yd (+ stats_d / bias_d).
extern "C" int gdrn_conv3x3s2(const gdrn_s2_params* pp, void* stream) {
    if (!pp || !pp->x || !pp->w || !pp->y) return GDRN_ERR_ARG;
    if (pp->dtype != GDRN_DT_H16) return GDRN_ERR_ARG;
    if ((pp->wd != nullptr) != (pp->yd != nullptr)) return GDRN_ERR_ARG;
    if (!pp->wd && (pp->stats_d || pp->bias_d)) return GDRN_ERR_ARG;
    if (!gdrn_conv3x3s2_ok(pp)) return GDRN_ERR_SHAPE;
    const gdrn_s2_params& p = *pp;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (p.bnb_x) {
        if (!p.bnb_mask || !p.bnb_mean || !p.bnb_invstd || !p.bnb_rows) return GDRN_ERR_ARG;
        return s2_tw(pp) == 16 ? launch_s2<false, 16, true>(p, st) : launch_s2<false, 8, true>(p, st);
    }
    if (s2_tw(pp) == 16) return p.wd ? launch_s2<true, 16>(p, st) : launch_s2<false, 16>(p, st);
    return p.wd ? launch_s2<true, 8>(p, st) : launch_s2<false, 8>(p, st);
}
