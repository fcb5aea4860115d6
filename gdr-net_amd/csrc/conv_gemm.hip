// Implicit-GEMM gather convolution on MFMA for gfx950 -- forward conv, data-gradient and transposed
// (stride-2) convolution of the GDR-Net RoI path, NHWC, fp32 accumulate.
//
//   y[m][co] = act( sum_{tap} sum_{c<Cin} X[pix(m,tap)][c] * Wp[co][tap][c]  (+ bias[co]) (+ addend[m][co]) )
//
// Replaces the ATen/cuDNN convolutions the reference dispatches at resnet_backbone.py:23,69-80 (via
// torchvision BasicBlock), cdpn_rot_head_region.py:81-135 (ConvTranspose2d + Conv2d) and
// conv_pnp_net.py:76-92 (Conv2d + Linear) -- forward, and the data-gradient of each in backward.
//
// Tile: BM output pixels x BN output channels per 256-thread workgroup (4 waves, 2x2), K staged in
// 128-byte rows (64 bf16 / 32 fp32 per stage) through double-buffered, XOR-swizzled LDS.  Register
// staging (global -> VGPR -> LDS) is used on purpose: it lets out-of-image taps be zero-filled and
// keeps the door open for fusing the producer's BN-apply+ReLU into the operand load.
// MFMA: v_mfma_f32_16x16x32_bf16 (T = bf16) / v_mfma_f32_16x16x4_f32 (T = float; exact fp32).
// The weight rows feed the MFMA "A" operand and the pixel rows the "B" operand, so each lane ends up
// with 4 consecutive output channels of one pixel (8/16-byte stores, per-channel BN statistics by a
// 16-lane butterfly).
#include "common.h"
#include "../../include/gdrn_hip.h"

namespace {

constexpr int ROWB = 128;  // bytes of K per LDS row per stage

template <typename T>
__device__ __forceinline__ f32x4_t mma_step(uint4 a, uint4 b, f32x4_t c);

template <>
__device__ __forceinline__ f32x4_t mma_step<bf16_t>(uint4 a, uint4 b, f32x4_t c) {
    return GDRN_MFMA16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c);
}

template <>
__device__ __forceinline__ f32x4_t mma_step<float>(uint4 a, uint4 b, f32x4_t c) {
    // lane group g holds k = 4g..4g+3 of this 16-wide k-step for BOTH operands; the four MFMAs below
    // therefore cover k = {j, 4+j, 8+j, 12+j} for j = 0..3 -- every k exactly once.
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.x), __uint_as_float(b.x), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.y), __uint_as_float(b.y), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.z), __uint_as_float(b.z), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.w), __uint_as_float(b.w), c, 0, 0, 0);
    return c;
}

template <typename T, int BM, int BN>
__global__ __launch_bounds__(256, (sizeof(T) == 2 && BM == 64 && BN == 128) ? 3 : 2) void conv_gemm_kernel(const gdrn_conv_params p) {
    constexpr int EPS = ROWB / (int)sizeof(T);  // elements of K per stage row
    constexpr int A_LD = BN / 32;               // weight-tile 16B loads per thread
    constexpr int B_LD = BM / 32;               // pixel-tile 16B loads per thread
    constexpr int WM = BM / 2, WN = BN / 2;     // per-wave tile
    constexpr int FM = WM / 16, FN = WN / 16;   // 16x16 fragments per wave
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int* s_taps = reinterpret_cast<int*>(smem);                 // [64] valid taps + [64]=count
    unsigned char* sW = smem + 512;                              // 2 stages x BN rows x 128 B
    unsigned char* sX = sW + 2 * BN * ROWB;                      // 2 stages x BM rows x 128 B

    __builtin_amdgcn_s_setprio(2);  // (the chain shares its CUs with a side-stream weight-gradient launch: its waves go first)
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int g = lane >> 4, r16 = lane & 15;

    const int NT = (p.Cout + BN - 1) / BN;
    // XCD-aware tile order: hardware places workgroup b on XCD b % 8 (each XCD has a private 4 MiB L2).
    // Give every XCD a contiguous run of tiles, so the N-tiles of one pixel tile and the neighbouring
    // pixel tiles (which share 3x3 halo rows) hit the same L2.  Bijective for any grid size.
    int bid = blockIdx.x;
    {
        const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = bid & 7, j = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
    }
    const int mt = bid / NT, nt = bid % NT;
    const int co0 = nt * BN;

    // ---- tile -> pixel mapping -------------------------------------------------------------
    int cls = 0, row0;
    int Hc = p.Ho, Wc = p.Wo, Mrows = p.M;
    if (p.mode == 1) {
        Hc = p.Ho >> 1; Wc = p.Wo >> 1;
        const int tpc = (p.M + BM - 1) / BM;
        cls = mt / tpc;
        row0 = (mt % tpc) * BM;
    } else {
        row0 = mt * BM;
    }
    const int py = cls >> 1, px = cls & 1;

    if (tid == 0) {
        int n = 0;
        for (int t = 0; t < p.KH * p.KW; ++t) {
            const int ky = t / p.KW, kx = t % p.KW;
            bool ok = true;
            if (p.mode == 1) ok = (((py + p.pad - ky) & 1) == 0) && (((px + p.pad - kx) & 1) == 0);
            if (ok) s_taps[n++] = t;
        }
        s_taps[64] = n;
    }

    const int seg = tid & 7, lrow = tid >> 3;
    const int pseg = seg ^ (lrow & 7);  // swizzled 16B slot inside the 128B LDS row (row&7 == lrow&7)

    // pixel rows this thread stages
    int pbase[B_LD], pya[B_LD], pxa[B_LD];
    bool pval[B_LD];
#pragma unroll
    for (int i = 0; i < B_LD; ++i) {
        const int r = row0 + lrow + 32 * i;
        pval[i] = r < Mrows;
        const int rr = pval[i] ? r : 0;
        const int n = rr / (Hc * Wc);
        const int rem = rr - n * (Hc * Wc);
        int oy = rem / Wc, ox = rem - (rem / Wc) * Wc;
        if (p.mode == 1) {
            pya[i] = 2 * oy + py + p.pad;  // iy = (pya - ky) >> 1
            pxa[i] = 2 * ox + px + p.pad;
        } else {
            pya[i] = oy * p.stride - p.pad;  // iy = pya + ky
            pxa[i] = ox * p.stride - p.pad;
        }
        pbase[i] = n * p.Hi * p.Wi;
    }

    __syncthreads();
    const int ntap = s_taps[64];
    const int kch = p.Cin / EPS;
    const int nstage = ntap * kch;
    const int KK = p.KH * p.KW;

    f32x4_t acc[FN][FM];
#pragma unroll
    for (int a = 0; a < FN; ++a)
#pragma unroll
        for (int b = 0; b < FM; ++b) acc[a][b] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    // Staging registers.  Loads are branch-free (out-of-image taps read a clamped in-bounds address and
    // are then zeroed) with 32-bit byte offsets from a uniform base, so that nothing spills and the
    // loads of stage s+1 stay in flight under the MFMAs of stage s.
    // (named scalars instead of arrays: hipcc left a uint4 staging array in scratch memory, which put a
    //  vmcnt(0) + scratch round trip between the global loads and the MFMAs of every stage.)
    uint4 rw0, rw1, rw2, rw3, rx0, rx1, rx2, rx3;
    rw0 = rw1 = rw2 = rw3 = rx0 = rx1 = rx2 = rx3 = make_uint4(0, 0, 0, 0);
    const char* xg = reinterpret_cast<const char*>(p.x);
    const char* wg = reinterpret_cast<const char*>(p.w);
    // weight buffer is zero-padded to a multiple of BN rows
    // LDS row l of the weight tile holds the operand's row co0 + perm(l): inside a wave's WN rows the FN 16-row fragments interleave in
    // units of 4 rows, so that an MFMA result lane (rows 4g..4g+3 of every fragment) owns 4*FN CONTIGUOUS output channels and the
    // epilogue moves 16 bytes per access instead of 8 (the halo kernel gets the same order from gdrn_pack_wfrag)
    unsigned wrow[A_LD];
#pragma unroll
    for (int i = 0; i < A_LD; ++i) {
        const int l = lrow + 32 * i, lw = l % WN, a_ = lw / 16, r_ = lw % 16;
        const int pr = (l / WN) * WN + (r_ >> 2) * (4 * FN) + a_ * 4 + (r_ & 3);
        wrow[i] = (unsigned)(co0 + pr) * (unsigned)(KK * p.Cin) * (unsigned)sizeof(T) + seg * 16;
    }
    const int HiM = p.Hi - 1, WiM = p.Wi - 1;

    // stage s <-> (channel chunk kc, valid-tap index ti): chunk outer, tap inner -- the taps of one
    // 128-byte channel chunk re-read the same few input rows back to back (L1/L2 hits).
    int n_kc = 0, n_ti = 0;  // coordinates of the NEXT stage to load
#define GDRN_LDW(i, dst) dst = *reinterpret_cast<const uint4*>(wg + (wrow[i] + wo_));
#define GDRN_LDX(i, dst)                                                                                       \
    {                                                                                                         \
        int iy_, ix_;                                                                                         \
        if (p.mode == 1) { iy_ = (pya[i] - ky_) >> 1; ix_ = (pxa[i] - kx_) >> 1; }                            \
        else { iy_ = pya[i] + ky_; ix_ = pxa[i] + kx_; }                                                      \
        const bool ok_ = pval[i] && iy_ >= 0 && iy_ <= HiM && ix_ >= 0 && ix_ <= WiM;                         \
        const int iyc_ = min(max(iy_, 0), HiM), ixc_ = min(max(ix_, 0), WiM);                                 \
        const unsigned off_ = ((unsigned)(pbase[i] + iyc_ * p.Wi + ixc_) * (unsigned)p.x_cs + (unsigned)(n_kc * EPS)) * \
                                  (unsigned)sizeof(T) + seg * 16;                                             \
        const uint4 v_ = *reinterpret_cast<const uint4*>(xg + off_);                                          \
        dst = ok_ ? v_ : make_uint4(0, 0, 0, 0);                                                              \
    }
#define GDRN_LOAD_STAGE()                                                                                     \
    do {                                                                                                      \
        const int tap_ = s_taps[n_ti];                                                                        \
        const int ky_ = tap_ / p.KW, kx_ = tap_ - ky_ * p.KW;                                                 \
        const unsigned wo_ = (unsigned)(tap_ * p.Cin + n_kc * EPS) * (unsigned)sizeof(T);                     \
        GDRN_LDW(0, rw0) GDRN_LDW(1, rw1)                                                                     \
        if constexpr (A_LD > 2) { GDRN_LDW(2, rw2) GDRN_LDW(3, rw3) }                                         \
        GDRN_LDX(0, rx0) GDRN_LDX(1, rx1)                                                                     \
        if constexpr (B_LD > 2) { GDRN_LDX(2, rx2) GDRN_LDX(3, rx3) }                                         \
        if (++n_ti == ntap) { n_ti = 0; ++n_kc; }                                                             \
    } while (0)
#define GDRN_STW(i, src) *reinterpret_cast<uint4*>(sW + (buf_) * (BN * ROWB) + (lrow + 32 * (i)) * ROWB + pseg * 16) = src;
#define GDRN_STX(i, src) *reinterpret_cast<uint4*>(sX + (buf_) * (BM * ROWB) + (lrow + 32 * (i)) * ROWB + pseg * 16) = src;
#define GDRN_WRITE_STAGE(b__)                                                                                 \
    do {                                                                                                      \
        const int buf_ = (b__);                                                                               \
        GDRN_STW(0, rw0) GDRN_STW(1, rw1)                                                                     \
        if constexpr (A_LD > 2) { GDRN_STW(2, rw2) GDRN_STW(3, rw3) }                                         \
        GDRN_STX(0, rx0) GDRN_STX(1, rx1)                                                                     \
        if constexpr (B_LD > 2) { GDRN_STX(2, rx2) GDRN_STX(3, rx3) }                                         \
    } while (0)

    if (nstage > 0) {
        GDRN_LOAD_STAGE();
        GDRN_WRITE_STAGE(0);
    }
    __syncthreads();

    for (int s = 0; s < nstage; ++s) {
        const int buf = s & 1;
        if (s + 1 < nstage) GDRN_LOAD_STAGE();
        const unsigned char* aW = sW + (size_t)buf * BN * ROWB;
        const unsigned char* aX = sX + (size_t)buf * BM * ROWB;
        // fp32 (parity) mode: accumulate each 32-deep stage in a fresh register tile and add it to the
        // running sum, so no fp32 addition chain is longer than 32 + #stages (instead of K = taps*Cin):
        // keeps the rounding error of K = 2304..4608 reductions at the level of a blocked CPU sum.
        f32x4_t part[FN][FM];
        if constexpr (sizeof(T) == 4) {
#pragma unroll
            for (int a = 0; a < FN; ++a)
#pragma unroll
                for (int b = 0; b < FM; ++b) part[a][b] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            uint4 fa[FN], fb[FM];
            const int lseg = ks * 4 + g;
#pragma unroll
            for (int a = 0; a < FN; ++a) {
                const int row = wn * WN + a * 16 + r16;
                fa[a] = *reinterpret_cast<const uint4*>(aW + row * ROWB + ((lseg ^ (row & 7)) << 4));
            }
#pragma unroll
            for (int b = 0; b < FM; ++b) {
                const int row = wm * WM + b * 16 + r16;
                fb[b] = *reinterpret_cast<const uint4*>(aX + row * ROWB + ((lseg ^ (row & 7)) << 4));
            }
#pragma unroll
            for (int a = 0; a < FN; ++a)
#pragma unroll
                for (int b = 0; b < FM; ++b) {
                    if constexpr (sizeof(T) == 4) part[a][b] = mma_step<T>(fa[a], fb[b], part[a][b]);
                    else acc[a][b] = mma_step<T>(fa[a], fb[b], acc[a][b]);
                }
        }
        if constexpr (sizeof(T) == 4) {
#pragma unroll
            for (int a = 0; a < FN; ++a)
#pragma unroll
                for (int b = 0; b < FM; ++b) acc[a][b] += part[a][b];
        }
        if (s + 1 < nstage) GDRN_WRITE_STAGE(buf ^ 1);
        __syncthreads();
    }
#undef GDRN_LOAD_STAGE
#undef GDRN_WRITE_STAGE
#undef GDRN_LDW
#undef GDRN_LDX
#undef GDRN_STW
#undef GDRN_STX

    // ---- epilogue ----------------------------------------------------------------------------
    // lane holds D[i = g*4+j][col = r16] of fragment (a,b): channel co0 + wn*WN + a*16 + g*4 + j,
    // pixel row wm*WM + b*16 + r16.
    if (p.stats != nullptr) {
        float* red = reinterpret_cast<float*>(smem + 512);  // [2 (wm)][BN][2]
#pragma unroll
        for (int a = 0; a < FN; ++a) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float s1, s2;
                if constexpr (sizeof(T) == 4) {
                    // parity (fp32) mode: the per-tile sums in fp64 -- the pose error against the reference sits at the
                    // fp32 summation-order noise floor (~1e-4 after the ~1600x amplification through BN at B=4), so the
                    // statistics should not add their own rounding to it
                    double d1 = 0.0, d2 = 0.0;
#pragma unroll
                    for (int b = 0; b < FM; ++b) { const double v = (double)acc[a][b][j]; d1 += v; d2 += v * v; }
#pragma unroll
                    for (int o = 8; o > 0; o >>= 1) { d1 += __shfl_xor(d1, o, 64); d2 += __shfl_xor(d2, o, 64); }
                    s1 = (float)d1;
                    s2 = (float)d2;
                } else {
                    s1 = 0.f;
                    s2 = 0.f;
#pragma unroll
                    for (int b = 0; b < FM; ++b) { const float v = acc[a][b][j]; s1 += v; s2 += v * v; }
                    s1 = row16_sum(s1);
                    s2 = row16_sum(s2);
                }
                if (r16 == 0) {
                    const int c = wn * WN + g * (4 * FN) + a * 4 + j;
                    red[(wm * BN + c) * 2 + 0] = s1;
                    red[(wm * BN + c) * 2 + 1] = s2;
                }
            }
        }
        __syncthreads();
        if (tid < BN && co0 + tid < p.Cout) {
            const float s1 = red[tid * 2 + 0] + red[(BN + tid) * 2 + 0];
            const float s2 = red[tid * 2 + 1] + red[(BN + tid) * 2 + 1];
            p.stats[((size_t)mt * 2 + 0) * p.Cout + co0 + tid] = s1;
            p.stats[((size_t)mt * 2 + 1) * p.Cout + co0 + tid] = s2;
        }
    }

    // Fast path (full channel tile, storage-dtype output, no bias / activation; optional addend): straight-line stores
    // with 32-bit offsets.  The generic code below is ~2000-3500 instructions with 200-400 branches per wave, which
    // exceeds the MFMA loop of the short-K layers (1x1 convs, stride-2 convs of the small feature maps).
    if (p.act <= 1 && (!p.out_f32 || sizeof(T) == 4) && co0 + BN <= p.Cout &&
        (sizeof(T) == 4 || (((p.y_cs | p.add_cs | p.bnb_cs) & 7) == 0))) {   // bf16: rows of whole 16-byte groups
        const char* ab = reinterpret_cast<const char*>(p.addend);
        char* yb = reinterpret_cast<char*>(p.y);
        const unsigned cl = (unsigned)(co0 + wn * WN + g * (4 * FN));  // + a*4: the lane's 4*FN contiguous channels
        // bf16: the lane's channels in 16-byte accesses (two fragments a = 2k, 2k + 1 per access)
        auto ld_ch = [&](const char* base, unsigned elem_off, uint2 (&dst)[FN]) {
            if constexpr ((FN & 1) == 0) {
#pragma unroll
                for (int k = 0; k < FN / 2; ++k) {
                    const uint4 q = *reinterpret_cast<const uint4*>(base + (elem_off + (unsigned)(k * 8)) * 2u);
                    dst[2 * k] = make_uint2(q.x, q.y);
                    dst[2 * k + 1] = make_uint2(q.z, q.w);
                }
            } else {
#pragma unroll
                for (int a = 0; a < FN; ++a) dst[a] = *reinterpret_cast<const uint2*>(base + (elem_off + (unsigned)(a * 4)) * 2u);
            }
        };
        auto st_ch = [&](char* base, unsigned elem_off, const uint2 (&src)[FN]) {
            if constexpr ((FN & 1) == 0) {
#pragma unroll
                for (int k = 0; k < FN / 2; ++k)
                    *reinterpret_cast<uint4*>(base + (elem_off + (unsigned)(k * 8)) * 2u) = make_uint4(src[2 * k].x, src[2 * k].y, src[2 * k + 1].x, src[2 * k + 1].y);
            } else {
#pragma unroll
                for (int a = 0; a < FN; ++a) *reinterpret_cast<uint2*>(base + (elem_off + (unsigned)(a * 4)) * 2u) = src[a];
            }
        };
        if constexpr (sizeof(T) == 2) {
            if (p.bnb_x != nullptr) {
                // data gradient feeding a BatchNorm(+ReLU) backward (the 1x1 output conv, the stride-2 / transposed data gradients):
                // ReLU mask + the BatchNorm backward's two per-channel sums on the accumulators, as the halo kernel's epilogue does
                // (conv3x3_halo.hip) -- the separate gdrn_bn_bwd_reduce pass over (dy, x, mask) disappears
                const char* xb = reinterpret_cast<const char*>(p.bnb_x);
                const char* mb = reinterpret_cast<const char*>(p.bnb_mask);
                const bool affine = mb == nullptr && p.bnb_scale != nullptr;
                float kmu[FN][4], kis[FN][4], ksc[FN][4], ksh[FN][4], t1[FN][4], t2[FN][4];
#pragma unroll
                for (int a = 0; a < FN; ++a) {
                    const float4 mu = *reinterpret_cast<const float4*>(p.bnb_mean + cl + a * 4);
                    const float4 is = *reinterpret_cast<const float4*>(p.bnb_invstd + cl + a * 4);
                    kmu[a][0] = mu.x; kmu[a][1] = mu.y; kmu[a][2] = mu.z; kmu[a][3] = mu.w;
                    kis[a][0] = is.x; kis[a][1] = is.y; kis[a][2] = is.z; kis[a][3] = is.w;
                    float4 sc = make_float4(0.f, 0.f, 0.f, 0.f), sh = make_float4(1.f, 1.f, 1.f, 1.f);  // mask always true
                    if (affine) {
                        sc = *reinterpret_cast<const float4*>(p.bnb_scale + cl + a * 4);
                        sh = *reinterpret_cast<const float4*>(p.bnb_shift + cl + a * 4);
                    }
                    ksc[a][0] = sc.x; ksc[a][1] = sc.y; ksc[a][2] = sc.z; ksc[a][3] = sc.w;
                    ksh[a][0] = sh.x; ksh[a][1] = sh.y; ksh[a][2] = sh.z; ksh[a][3] = sh.w;
#pragma unroll
                    for (int j = 0; j < 4; ++j) { t1[a][j] = 0.f; t2[a][j] = 0.f; }
                }
#pragma unroll
                for (int b = 0; b < FM; ++b) {
                    const int r = row0 + wm * WM + b * 16 + r16;
                    if (r >= Mrows) continue;
                    unsigned orow;
                    if (p.mode == 1) {
                        const int n = r / (Hc * Wc);
                        const int rem = r - n * (Hc * Wc);
                        const int yc = rem / Wc, xc = rem - yc * Wc;
                        orow = (unsigned)((n * p.Ho + 2 * yc + py) * p.Wo + 2 * xc + px);
                    } else {
                        orow = (unsigned)r;
                    }
                    uint2 xq[FN], aq[FN], mq[FN], ov[FN];
                    ld_ch(xb, orow * (unsigned)p.bnb_cs + cl, xq);
#pragma unroll
                    for (int a = 0; a < FN; ++a) { aq[a] = make_uint2(0u, 0u); mq[a] = make_uint2(GDRN_H16_ONE2, GDRN_H16_ONE2); }
                    if (ab != nullptr) ld_ch(ab, orow * (unsigned)p.add_cs + cl, aq);
                    if (mb != nullptr) ld_ch(mb, orow * (unsigned)p.bnb_cs + cl, mq);
#pragma unroll
                    for (int a = 0; a < FN; ++a) {
                        const float xv[4] = {h16lo(xq[a].x), h16hi(xq[a].x),
                                             h16lo(xq[a].y), h16hi(xq[a].y)};
                        const float mv[4] = {h16lo(mq[a].x), h16hi(mq[a].x),
                                             h16lo(mq[a].y), h16hi(mq[a].y)};
                        const float ad[4] = {h16lo(aq[a].x), h16hi(aq[a].x),
                                             h16lo(aq[a].y), h16hi(aq[a].y)};
                        float v[4];
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const bool keep = (mv[j] > 0.f) && (xv[j] * ksc[a][j] + ksh[a][j] > 0.f);
                            const float gv = keep ? acc[a][b][j] + ad[j] : 0.f;
                            v[j] = gv;
                            t1[a][j] += gv;
                            t2[a][j] += gv * (xv[j] - kmu[a][j]) * kis[a][j];
                        }
                        ov[a] = make_uint2(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]));
                    }
                    st_ch(yb, orow * (unsigned)p.y_cs + cl, ov);
                }
                // one row of the two sums per pixel tile (the two pixel halves of the tile are added through LDS in a fixed order)
                float* red = reinterpret_cast<float*>(smem + 512);  // [2 (wm)][BN][2]
#pragma unroll
                for (int a = 0; a < FN; ++a)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float u1 = row16_sum(t1[a][j]), u2 = row16_sum(t2[a][j]);
                        if (r16 == 0) {
                            const int c = wn * WN + g * (4 * FN) + a * 4 + j;
                            red[(wm * BN + c) * 2 + 0] = u1;
                            red[(wm * BN + c) * 2 + 1] = u2;
                        }
                    }
                __syncthreads();
                if (tid < BN) {
                    p.bnb_rows[((size_t)mt * 2 + 0) * p.Cout + co0 + tid] = red[tid * 2 + 0] + red[(BN + tid) * 2 + 0];
                    p.bnb_rows[((size_t)mt * 2 + 1) * p.Cout + co0 + tid] = red[tid * 2 + 1] + red[(BN + tid) * 2 + 1];
                }
                return;
            }
        }
#pragma unroll
        for (int b = 0; b < FM; ++b) {
            const int r = row0 + wm * WM + b * 16 + r16;
            if (r >= Mrows) continue;
            unsigned orow;
            if (p.mode == 1) {
                const int n = r / (Hc * Wc);
                const int rem = r - n * (Hc * Wc);
                const int yc = rem / Wc, xc = rem - yc * Wc;
                orow = (unsigned)((n * p.Ho + 2 * yc + py) * p.Wo + 2 * xc + px);
            } else {
                orow = (unsigned)r;
            }
            uint2 aqb[FN], ovb[FN];
            if constexpr (sizeof(T) == 2) {
                if (ab != nullptr) ld_ch(ab, orow * (unsigned)p.add_cs + cl, aqb);
            }
#pragma unroll
            for (int a = 0; a < FN; ++a) {
                float v0 = acc[a][b][0], v1 = acc[a][b][1], v2 = acc[a][b][2], v3 = acc[a][b][3];
                const unsigned c = cl + a * 4;
                if (p.bias != nullptr) {  // eval mode: the folded BatchNorm shift
                    const float4 bq = *reinterpret_cast<const float4*>(p.bias + c);
                    v0 += bq.x; v1 += bq.y; v2 += bq.z; v3 += bq.w;
                }
                if (ab != nullptr) {
                    if constexpr (sizeof(T) == 2) {
                        const uint2 q = aqb[a];
                        v0 += h16lo(q.x); v1 += h16hi(q.x);
                        v2 += h16lo(q.y); v3 += h16hi(q.y);
                    } else {
                        const float4 q = *reinterpret_cast<const float4*>(ab + ((size_t)orow * p.add_cs + c) * 4);
                        v0 += q.x; v1 += q.y; v2 += q.z; v3 += q.w;
                    }
                }
                if (p.act == 1) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); v2 = fmaxf(v2, 0.f); v3 = fmaxf(v3, 0.f); }
                if constexpr (sizeof(T) == 2) ovb[a] = make_uint2(pack_bf2(v0, v1), pack_bf2(v2, v3));
                else *reinterpret_cast<float4*>(yb + ((size_t)orow * p.y_cs + c) * 4) = make_float4(v0, v1, v2, v3);
            }
            if constexpr (sizeof(T) == 2) st_ch(yb, orow * (unsigned)p.y_cs + cl, ovb);
        }
        return;
    }
    // fp32 rows out of a bf16 GEMM with a bias and a partial channel tile (the head's 1x1 output conv: 69 channels in 72-float rows):
    // whole 16-byte groups, no per-element bounds checks -- the group that straddles Cout also writes the row's pad floats (zero
    // weights: the accumulator, no bias).  The generic code below ran this layer at 75 us.
    if constexpr (sizeof(T) == 2) {
        if (p.out_f32 && p.act == 0 && p.addend == nullptr && p.mode == 0 && (p.y_cs & 3) == 0) {
            float* yf = reinterpret_cast<float*>(p.y);
#pragma unroll
            for (int a = 0; a < FN; ++a) {
                const int c0 = co0 + wn * WN + g * (4 * FN) + a * 4;
                if (c0 >= p.y_cs) continue;
                float bq[4] = {0.f, 0.f, 0.f, 0.f};
                if (p.bias != nullptr) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) if (c0 + j < p.Cout) bq[j] = p.bias[c0 + j];
                }
#pragma unroll
                for (int b = 0; b < FM; ++b) {
                    const int r = row0 + wm * WM + b * 16 + r16;
                    if (r >= Mrows) continue;
                    *reinterpret_cast<float4*>(yf + (size_t)r * p.y_cs + c0) =
                        make_float4(acc[a][b][0] + bq[0], acc[a][b][1] + bq[1], acc[a][b][2] + bq[2], acc[a][b][3] + bq[3]);
                }
            }
            return;
        }
    }
#pragma unroll
    for (int b = 0; b < FM; ++b) {
        const int r = row0 + wm * WM + b * 16 + r16;
        if (r >= Mrows) continue;
        size_t orow;
        if (p.mode == 1) {
            const int n = r / (Hc * Wc);
            const int rem = r - n * (Hc * Wc);
            const int yc = rem / Wc, xc = rem - yc * Wc;
            orow = (size_t)(n * p.Ho + 2 * yc + py) * p.Wo + 2 * xc + px;
        } else {
            orow = (size_t)r;
        }
#pragma unroll
        for (int a = 0; a < FN; ++a) {
            const int c0 = co0 + wn * WN + g * (4 * FN) + a * 4;
            if (c0 >= p.Cout) continue;
            float v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = acc[a][b][j];
            const bool full = (c0 + 3 < p.Cout);
            if (p.bias != nullptr) {
#pragma unroll
                for (int j = 0; j < 4; ++j) if (c0 + j < p.Cout) v[j] += p.bias[c0 + j];
            }
            if (p.addend != nullptr) {
                const T* ap = reinterpret_cast<const T*>(p.addend) + orow * p.add_cs + c0;
#pragma unroll
                for (int j = 0; j < 4; ++j) if (c0 + j < p.Cout) v[j] += ld1<T>(ap + j);
            }
            if (p.act == 1) {
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], 0.f);
            } else if (p.act == 2) {
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = v[j] > 0.f ? v[j] : 0.1f * v[j];
            }
            if (p.out_f32 || sizeof(T) == 4) {
                float* yp = reinterpret_cast<float*>(p.y) + orow * p.y_cs + c0;
                if (full) *reinterpret_cast<float4*>(yp) = make_float4(v[0], v[1], v[2], v[3]);
                else
#pragma unroll
                    for (int j = 0; j < 4; ++j) if (c0 + j < p.Cout) yp[j] = v[j];
            } else {
                bf16_t* yp = reinterpret_cast<bf16_t*>(p.y) + orow * p.y_cs + c0;
                if (full) *reinterpret_cast<uint2*>(yp) = make_uint2(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]));
                else
#pragma unroll
                    for (int j = 0; j < 4; ++j) if (c0 + j < p.Cout) yp[j] = f2bf(v[j]);
            }
        }
    }
}

template <typename T, int BM, int BN>
int launch(const gdrn_conv_params& p, hipStream_t st) {
    constexpr size_t smem = 512 + 2 * (size_t)(BM + BN) * ROWB;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_gemm_kernel<T, BM, BN>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess)
            return GDRN_ERR_LAUNCH;
        attr_set = true;
    }
    const int NT = cdiv(p.Cout, BN);
    const int MT = (p.mode == 1) ? 4 * cdiv(p.M, BM) : cdiv(p.M, BM);
    GDRN_LAUNCH((conv_gemm_kernel<T, BM, BN>), dim3(MT * NT), dim3(256), smem, st, p);
    GDRN_CHECK_LAUNCH();
    return GDRN_OK;
}

}  // namespace

extern "C" int gdrn_conv_tile(const gdrn_conv_params* p, int* bm, int* bn) {
    if (!p || !bm || !bn) return GDRN_ERR_ARG;
    // BN: 64 for <= 64 output channels, else 128.  BM: 128 unless that leaves the chip mostly idle.
    *bn = (p->Cout <= 64) ? 64 : 128;
    const int nt = cdiv(p->Cout, *bn);
    const int rows = (p->mode == 1) ? 4 * p->M : p->M;
    *bm = (cdiv(rows, 128) * nt >= 512) ? 128 : 64;
    return GDRN_OK;
}

extern "C" int gdrn_conv_stats_rows(const gdrn_conv_params* p) {
    int bm, bn;
    if (gdrn_conv_tile(p, &bm, &bn) != GDRN_OK) return GDRN_ERR_ARG;
    return (p->mode == 1) ? 4 * cdiv(p->M, bm) : cdiv(p->M, bm);
}

extern "C" int gdrn_conv_gemm(const gdrn_conv_params* pp, void* stream) {
    if (!pp || !pp->x || !pp->w || !pp->y) return GDRN_ERR_ARG;
    const gdrn_conv_params& p = *pp;
    const int esz = (p.dtype == GDRN_DT_H16) ? 2 : 4;
    if (p.dtype != GDRN_DT_F32 && p.dtype != GDRN_DT_H16) return GDRN_ERR_ARG;
    if (p.Cin <= 0 || (p.Cin * esz) % ROWB != 0) return GDRN_ERR_SHAPE;
    if ((p.x_cs * esz) % 8 != 0 || p.KH * p.KW > 64 || p.KH * p.KW < 1) return GDRN_ERR_SHAPE;
    if (p.mode == 1 && (p.stride != 2 || (p.Ho & 1) || (p.Wo & 1))) return GDRN_ERR_SHAPE;
    if (p.mode != 0 && p.mode != 1) return GDRN_ERR_ARG;
    if (p.M <= 0 || p.Cout <= 0 || (p.y_cs & 3)) return GDRN_ERR_SHAPE;
    if (p.addend && (p.add_cs & 3)) return GDRN_ERR_SHAPE;
    int bm, bn;
    gdrn_conv_tile(pp, &bm, &bn);
    if (p.w_rows < cdiv(p.Cout, bn) * bn) return GDRN_ERR_SHAPE;
    if (p.bnb_x) {  // fused BatchNorm-backward statistics: the straight-line bf16 epilogue only
        if (p.dtype != GDRN_DT_H16 || !p.bnb_mean || !p.bnb_invstd || !p.bnb_rows || (p.bnb_scale != nullptr) != (p.bnb_shift != nullptr)) return GDRN_ERR_ARG;
        if (p.bias || p.act || p.out_f32 || p.stats || (p.Cout % bn) || (p.bnb_cs & 7) || (p.y_cs & 7) || (p.addend && (p.add_cs & 7)) || p.bnb_cs < p.Cout)
            return GDRN_ERR_SHAPE;  // (rows of whole 16-byte groups: the epilogue's accesses)
        const unsigned long long rows_out = (p.mode == 1) ? 4ull * (unsigned long long)p.M : (unsigned long long)p.M;
        if (rows_out * (unsigned long long)std::max(std::max(p.y_cs, p.add_cs), p.bnb_cs) * 2ull >= (1ull << 32)) return GDRN_ERR_SHAPE;  // 32-bit offsets
    }
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (p.dtype == GDRN_DT_H16) {
        if (bn == 64) return bm == 128 ? launch<bf16_t, 128, 64>(p, st) : launch<bf16_t, 64, 64>(p, st);
        return bm == 128 ? launch<bf16_t, 128, 128>(p, st) : launch<bf16_t, 64, 128>(p, st);
    } else {
        if (bn == 64) return bm == 128 ? launch<float, 128, 64>(p, st) : launch<float, 64, 64>(p, st);
        return bm == 128 ? launch<float, 128, 128>(p, st) : launch<float, 64, 128>(p, st);
    }
}
