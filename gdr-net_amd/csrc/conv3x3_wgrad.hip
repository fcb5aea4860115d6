// Halo-tiled weight gradient of the 3x3 pad-1 convolutions (stride 1 and stride 2) on MFMA for gfx950 (bf16 operands, fp32
// accumulate):   dWp[co][tap][ci] += sum_{pixels} dY[pix][co] * X[stride * pix + tap][ci]
// (autograd backward-weight of the BasicBlock / head 3x3 convs, reference: engine.py:279).
//
// The generic kernel (conv_wgrad.hip) handles one tap per workgroup and re-reads dY and X once per tap.  Here a
// workgroup owns a 64(co) x 64(ci) tile of ALL NINE taps: per stage it stages an 8x8 pixel patch of dY and the
// 10x10 halo patch of X in LDS once and runs the nine taps against them (227 FLOP per byte moved instead of 64,
// 9x fewer atomics per MAC).  Both MFMA operands need the pixel (reduction) axis transposed: ds_read_b64_tr_b16
// with `lane base + immediate` addresses (row pitch 160 B: the 8 pixel rows of a 32-lane half hit disjoint banks).
// Accumulators: 9 taps x (2x2 fragments) x 4 = 144 VGPRs per lane; the dY fragments of a k-step are loaded once
// and reused by the nine taps.  Pixel patches are split over workgroups.  Partial tiles either go to p.dw with fp32
// atomics (p.ws == NULL) or -- the engine's path -- are stored as plain 16-byte vectors in fragment order into the
// workspace p.ws[split][tile][36 fragments][4 waves][64 lanes][4] and summed by gdrn_wgrad_reduce_multi, which also
// scatters into the parameter's gradient layout.  (Measured: every layer pays 256 workgroups x 36864 atomics ~ 31 us,
// 50+ us when all workgroups hit the same 64x64 tile as in layer1 -- more than the MFMA work of a ResNet layer.)
#include <hip/amd_detail/amd_hip_unsafe_atomics.h>

#include <algorithm>
#include <mutex>
#include <type_traits>
#include <utility>

#include "common.h"
#include "../../include/gdrn_hip.h"

namespace {

typedef __attribute__((address_space(3))) bf16x4_t lds_bf16x4_t;
// bytes per LDS pixel row of the dY patch: NA fragments of 16 output channels * 2 B + 32 pad.  160 B = 40 banks and 288 B = 72 banks = 8 mod 64
// both spread the 8 pixels x 32 B a 32-lane transposed read touches over the 64 banks without a conflict
template <int NA> struct DyGeo { static constexpr int PITCH = NA * 32 + 32, DYB = 64 * PITCH; };
// per-stage geometry.  Stride 1: an 8x8 patch of output pixels (two 32-pixel k-steps) against the 10x10 input halo patch.
// Stride 2 (the three ResNet stage-entry convs, the three Patch-PnP convs, and the head's ConvTranspose with the roles of
// input and output gradient swapped): a 4x8 patch of output pixels (one k-step) against the (2*4+1) x (2*8+1) = 9x17 input
// patch -- 153 pixels, so the staging registers (5 + 1 instead of 4 + 2 uint4) and the LDS stage stay the size of the stride-1 case.
template <int S> struct Geo;
// XP = LDS pitch of an X-patch pixel: the 32 lanes a transposed read services together touch 8 pixels (4 per 16-lane group) x 8 bytes x
// 4 channel quads; they must fall into 64 distinct banks.  Consecutive pixels (stride 1): 160 B = 40 banks -> 0, 40, 16, 56 (+32 for the
// second group).  Every second pixel (stride 2): 2 * 144 B = 72 banks = 8 mod 64 -> 0, 8, ..., 56; with 160 B the two groups collide
// (8 * 160 B = 0 mod 256 B): a 2-way conflict on every X read.
template <> struct Geo<1> { static constexpr int TH = 8, KS = 2, PW = 10, PH = 10, NX = 4, ND = 2, XP = 160; };
template <> struct Geo<2> { static constexpr int TH = 4, KS = 1, PW = 17, PH = 9, NX = 5, ND = 1, XP = 144; };
constexpr int XB = 153 * 144;            // X patch: 10x10 pixels x 160 B or 9x17 pixels x 144 B
template <int NA> constexpr int stage_bytes() { return DyGeo<NA>::DYB + XB; }   // dY patch (up to 8x8 pixels) | X patch

// compile-time loop: f(std::integral_constant<int, 0>) ... f(std::integral_constant<int, N - 1>)
template <class F, int... Is>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, Is...>) { (f(std::integral_constant<int, Is>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }

__device__ __forceinline__ bf16x8_t tr_pair(const unsigned char* p0, const unsigned char* p1) {
    bf16x4_t lo = GDRN_TR16((lds_bf16x4_t*)p0);
    bf16x4_t hi = GDRN_TR16((lds_bf16x4_t*)p1);
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

// XCD-aware order: consecutive logical ids (all tiles of one pixel range) run on the same XCD and share its L2
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int qq = nwg >> 3, r = nwg & 7, xcd = bid & 7, j = bid >> 3;
    return (xcd < r ? xcd * (qq + 1) : r * (qq + 1) + (xcd - r) * qq) + j;
}

// one workgroup: tile `bid % ntile`, pixel-range split `bid / ntile` of the layer described by p.  NA = 16-channel output fragments per
// wave = the workgroup tile's output channels / 16 = 4 (64 x 64 tile, 144 accumulator registers, two workgroups per CU)
template <int S, int NA>
__device__ __forceinline__ void wgrad_tile(const gdrn_wgrad_params& p, int bid, int npatch, int nsplit, unsigned char* smem) {
    using G_ = Geo<S>;
    constexpr int PW = G_::PW, NPIX = G_::PW * G_::PH, NSEG = NPIX * 8, XP = G_::XP;
    constexpr int PITCH = DyGeo<NA>::PITCH, DYB = DyGeo<NA>::DYB, STAGEB = stage_bytes<NA>();
    constexpr int DSEGS = 2 * NA;                  // 16-byte granules per dY pixel
    constexpr int NDL = G_::ND * NA / 4;           // dY granules per thread and stage
    constexpr int DPSTEP = 256 / DSEGS;            // pixels between a thread's dY granules (a multiple of the patch width 8)
    static_assert(NPIX * XP <= XB, "X patch fits its LDS slot");
    static_assert(NA == 4, "64 x 64 workgroup tile (the 128 x 64 form of rounds 4-5 -- NA = 8, accumulators named in the AGPRs -- was removed in round 6)");
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wa = wave >> 1, wb = wave & 1;
    const int g = lane >> 4, q = lane & 15;

    const int ncot = p.Cout / (16 * NA), ncit = p.Cin / 64, ntile = ncot * ncit;
    const int split = bid / ntile, tile = bid - split * ntile;
    const int co0 = (tile % ncot) * (16 * NA), ci0 = (tile / ncot) * 64;
    const int per = (npatch + nsplit - 1) / nsplit;
    const int p_begin = split * per, p_end = min(npatch, p_begin + per);
    if (p_begin >= p_end) return;
    const int tiles_x = p.Wo >> 3, tiles_y = p.Ho / G_::TH;

    // ---- staging geometry (constant per thread)
    // dY: TH*8 px x 8 segs -> ids tid (, tid+256) ; X: NPIX px x 8 segs -> ids tid + 256*i, i < NX
    const int xseg = tid & 7;                                              // X granule of this thread (8 per pixel)
    const int dseg = tid & (DSEGS - 1), dpix0 = tid / DSEGS;              // dY granule; pixels dpix0 + DPSTEP * i
    // global address space stated explicitly: in the grouped kernel the task's pointers come out of a table in memory, and hipcc then
    // emits FLAT loads -- which count on lgkmcnt as well, so every LDS wait in the MFMA loop became lgkmcnt(0) and also waited for the
    // next patch's global loads (the prefetch overlapped nothing)
    typedef const __attribute__((address_space(1))) char* gptr_t;
    typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
    typedef const __attribute__((address_space(1))) u32x4_t* gvec_t;
#define GLD16(ptr_) __builtin_bit_cast(uint4, *(gvec_t)(ptr_))
    const gptr_t dyg = (gptr_t)(reinterpret_cast<const char*>(p.dy) + (size_t)co0 * 2 + dseg * 16);
    const gptr_t xg = (gptr_t)(reinterpret_cast<const char*>(p.x) + (size_t)ci0 * 2);

    uint4 d0, d1, d2, d3, x0, x1, x2, x3, x4;
    d0 = d1 = d2 = d3 = x0 = x1 = x2 = x3 = x4 = make_uint4(0, 0, 0, 0);

    // The next patch's addresses are computed branch-free in pieces (the dY pair, one X slot each) that sit INSIDE the MFMA groups of
    // the current stage, so their ~100 VALU instructions issue in the MFMAs' shadow; as a block in front of the MFMA loop they cost a
    // wave ~1000 cycles per stage with the matrix pipe idle.  Out-of-image granules are loaded from a clamped address and zeroed at
    // the LDS write (bit i of okm); the last stage re-loads its own patch (nothing is written from it).
    unsigned cpp[G_::NX];  // X slot i of this thread: patch pixel row | column << 8
#pragma unroll
    for (int i = 0; i < G_::NX; ++i) {
        const int pp_ = (tid + 256 * i) >> 3;
        const int py_ = pp_ / PW;
        cpp[i] = (unsigned)py_ | ((unsigned)(pp_ - py_ * PW) << 8);
    }
    const unsigned xrow = (unsigned)p.x_cs * 2u, drow = (unsigned)p.dy_cs * 2u;
    unsigned okm = 0;
    int pn_n, pn_y, pn_x;  // image / pixel-tile coordinates of the patch being fetched
    {
        int t_ = p_begin;
        pn_x = t_ % tiles_x; t_ /= tiles_x;
        pn_y = t_ % tiles_y;
        pn_n = t_ / tiles_y;
    }
#define LDD()                                                                                         \
    {                                                                                                 \
        const int y0_ = pn_y * G_::TH, x0_ = pn_x << 3;                                               \
        const unsigned ra_ = (unsigned)((pn_n * p.Ho + y0_ + (dpix0 >> 3)) * p.Wo + x0_ + (dpix0 & 7)); \
        constexpr unsigned RS_ = DPSTEP / 8;   /* image rows between a thread's granules */              \
        d0 = GLD16(dyg + ra_ * drow);                                                                 \
        if constexpr (NDL >= 2) d1 = GLD16(dyg + (ra_ + RS_ * (unsigned)p.Wo) * drow);                \
        if constexpr (NDL == 4) {                                                                     \
            d2 = GLD16(dyg + (ra_ + 2u * RS_ * (unsigned)p.Wo) * drow);                               \
            d3 = GLD16(dyg + (ra_ + 3u * RS_ * (unsigned)p.Wo) * drow);                               \
        }                                                                                             \
    }
#define LDX(i, dst)                                                                                   \
    {                                                                                                 \
        const int iy_ = S * pn_y * G_::TH + (int)(cpp[i] & 255u) - 1, ix_ = S * (pn_x << 3) + (int)(cpp[i] >> 8) - 1; \
        const bool ok_ = (unsigned)iy_ < (unsigned)p.Hi && (unsigned)ix_ < (unsigned)p.Wi;            \
        const int iyc_ = min(max(iy_, 0), p.Hi - 1), ixc_ = min(max(ix_, 0), p.Wi - 1);               \
        dst = GLD16(xg + ((unsigned)((pn_n * p.Hi + iyc_) * p.Wi + ixc_) * xrow + xseg * 16));        \
        okm = (okm & ~(1u << (i))) | (ok_ ? (1u << (i)) : 0u);                                        \
    }
#define STX(i, src)                                                                                   \
    {                                                                                                 \
        const int id_ = tid + 256 * (i);                                                              \
        if (id_ < NSEG) *reinterpret_cast<uint4*>(sb_ + DYB + (id_ >> 3) * XP + (id_ & 7) * 16) = ((okm >> (i)) & 1u) ? src : make_uint4(0, 0, 0, 0); \
    }
#define WRITE_PATCH(buf)                                                                              \
    {                                                                                                 \
        unsigned char* sb_ = smem + (buf) * STAGEB;                                                   \
        *reinterpret_cast<uint4*>(sb_ + dpix0 * PITCH + dseg * 16) = d0;                              \
        if constexpr (NDL >= 2) *reinterpret_cast<uint4*>(sb_ + (dpix0 + DPSTEP) * PITCH + dseg * 16) = d1; \
        if constexpr (NDL == 4) {                                                                     \
            *reinterpret_cast<uint4*>(sb_ + (dpix0 + 2 * DPSTEP) * PITCH + dseg * 16) = d2;           \
            *reinterpret_cast<uint4*>(sb_ + (dpix0 + 3 * DPSTEP) * PITCH + dseg * 16) = d3;           \
        }                                                                                             \
        STX(0, x0) STX(1, x1) STX(2, x2) STX(3, x3)                                                   \
        if constexpr (G_::NX == 5) STX(4, x4)                                                         \
    }

    // wave tile: ALL 64 output channels (4 fragments) x 16 input channels (wave w: ci block w).  The X fragments are re-read for
    // every tap (9 x 2 k-steps), the dY fragments once per k-step: a 64 x 16 wave tile needs 16 + 36 = 52 transposed LDS reads per
    // stage where the 32 x 32 one needed 8 + 72 = 80, for the same 72 MFMAs
    f32x4_t acc[9][NA];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int a = 0; a < NA; ++a) acc[t][a] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    (void)wa; (void)wb;

    // transpose-read lane bases.  k (pixel) map of one 32-pixel k-step (4 image rows x 8): read r of lane group g
    // covers pixels (row = 2r + (g>>1), col = (g&1)*4 + 0..3); lane q supplies pixel col + (q>>2), channels (q&3)*4..
    const int lpy = g >> 1, lpx = (g & 1) * 4 + (q >> 2);
    const int baseA = (lpy * 8 + lpx) * PITCH + ((q & 3) * 4) * 2;                         // dY patch: 8 px per row
    const int baseB = DYB + (S * lpy * PW + S * lpx) * XP + (wave * 16 + (q & 3) * 4) * 2;  // X patch: PW px per row, output pixel (y, x) at (S*y, S*x)

    // Patch pipeline (r6b): ONE register set, two LDS stages, and a patch's registers are held for a whole stage -- the loads of patch pi + 2 go out
    // in the second half of stage pi (right behind the LDS write of patch pi + 1, which frees the registers) and are written in the middle of stage
    // pi + 1: ~17 MFMA groups (~1200 cycles) between a load and its use.  (Until r6: loads of patch pi + 1 in the first half of stage pi, written
    // behind its last group -- 9 to 18 groups; the probe build that re-fetches one cached patch showed a fifth of the kernel waiting there.)
#ifdef WGRAD_PROBE_SAMEPATCH   // probe build (tools/wgrad_lat_probe.py): every stage re-fetches the first patch -- cache hits, the arithmetic of a stage unchanged
#define ADVANCE_PATCH() {}
#else
#define ADVANCE_PATCH()                                                                               \
    {                                                                                                 \
        if (++pn_x == tiles_x) {                                                                      \
            pn_x = 0;                                                                                 \
            if (++pn_y == tiles_y) { pn_y = 0; ++pn_n; }                                              \
        }                                                                                             \
    }
#endif
    LDD() LDX(0, x0) LDX(1, x1) LDX(2, x2) LDX(3, x3)
    if constexpr (G_::NX == 5) LDX(4, x4)
    WRITE_PATCH(0)
    if (p_begin + 1 < p_end) {   // patch p_begin + 1 into the registers
        ADVANCE_PATCH()
        LDD() LDX(0, x0) LDX(1, x1) LDX(2, x2) LDX(3, x3)
        if constexpr (G_::NX == 5) LDX(4, x4)
    }
    __syncthreads();

    for (int pi = p_begin; pi < p_end; ++pi) {
        const int buf = (pi - p_begin) & 1;
        // (uniform) advance to patch pi + 2, the target of this stage's loads; without one the loads re-fetch the last patch and the write
        // below stores stale registers into the LDS stage nobody reads any more -- no branch inside the MFMA groups
        if (pi + 2 < p_end) ADVANCE_PATCH()
        const unsigned char* sa = smem + buf * STAGEB + baseA;
        const unsigned char* sx = smem + buf * STAGEB + baseB;
        // One stage = J = 9 * KS tap groups of 4 MFMAs (k-step ks = j / 9, tap t = j % 9).  Software pipeline over the groups: the X
        // fragment of group j + RD is read before the MFMAs of group j, the dY fragments of the second k-step one per group from
        // group 9 - 4 - RD on, so every transposed read has RD groups (~70 cycles each) of MFMAs to land under; hipcc's own order issues
        // the reads of a tap right before its MFMAs and waits out the LDS latency seven times per stage.
        constexpr int J = 9 * G_::KS, RD = 3;
        constexpr int PSTEP = (G_::KS == 2) ? 2 : 1;  // address pieces go into groups 1, 1 + PSTEP, ...
        constexpr int APG = NA / 4;                   // dY fragments of the second k-step read per group
        bf16x8_t fa[G_::KS][NA], fb[RD + 1];
#define FA_(ks_, a_) tr_pair(sa + ((ks_) * 4) * 8 * PITCH + (a_) * 32, sa + ((ks_) * 4 + 2) * 8 * PITCH + (a_) * 32)
#define FB_(j_) tr_pair(sx + ((S * ((j_) / 9) * 4 + ((j_) % 9) / 3) * PW + (((j_) % 9) % 3)) * XP,                               \
                        sx + ((S * ((j_) / 9) * 4 + ((j_) % 9) / 3) * PW + (((j_) % 9) % 3)) * XP + 2 * S * PW * XP)
#pragma unroll
        for (int a = 0; a < NA; ++a) fa[0][a] = FA_(0, a);
#pragma unroll
        for (int j = 0; j < RD; ++j) fb[j] = FB_(j);
#pragma unroll
        for (int j = 0; j < J; ++j) {
            if (j + RD < J) fb[(j + RD) % (RD + 1)] = FB_(j + RD);
            if constexpr (G_::KS == 2) {
                if (j >= 9 - 4 - RD && j < 9 - RD) {
#pragma unroll
                    for (int u = 0; u < APG; ++u) fa[1][(j - (9 - 4 - RD)) * APG + u] = FA_(1, (j - (9 - 4 - RD)) * APG + u);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int a = 0; a < NA; ++a) acc[j % 9][a] = GDRN_MFMA16(fa[j / 9][a], fb[j % (RD + 1)], acc[j % 9][a]);
            // group JW: the registers (patch pi + 1, requested a stage ago) go to the idle LDS stage; behind it, one address piece + its loads per
            // group for patch pi + 2 (same scheduling region as the MFMAs above)
            constexpr int JW = (G_::KS == 2) ? 8 : 2;
            if (j == JW) WRITE_PATCH(buf ^ 1)
            if (j == JW + 1) LDD()
            if (j == JW + 2) LDX(0, x0)
            if (j == JW + 2 + PSTEP) LDX(1, x1)
            if (j == JW + 2 + 2 * PSTEP) LDX(2, x2)
            if (j == JW + 2 + 3 * PSTEP) LDX(3, x3)
            if constexpr (G_::NX == 5) {
                if (j == JW + 2 + 4 * PSTEP) LDX(4, x4)
            }
            __builtin_amdgcn_sched_barrier(0);
        }
#undef FA_
#undef FB_
        __syncthreads();
    }
#undef ADVANCE_PATCH
#undef LDD
#undef GLD16
#undef LDX
#undef STX
#undef WRITE_PATCH

    auto tuple = [&](auto T_, auto A_) -> f32x4_t { return acc[decltype(T_)::value][decltype(A_)::value]; };
    if (p.ws != nullptr) {
        // workspace order (read by wgrad_reduce_multi): per 64 x 64 tile [36 fragments (t, a', b')][4 tile quadrants (wa', wb')][64 lanes][4]
        // with the 16 x 16 block (co16 = wa'*2 + a', ci16 = wb'*2 + b'): this wave holds co16 = a (0..3), ci16 = wave.  A 128-channel
        // workgroup tile is stored as its two 64 x 64 halves (a >> 2), which sit next to each other in the slab order
        // [split][ci tile][co tile of 64]
        constexpr int HT = NA / 4;
        const int ntile64 = ntile * HT, tile64 = (tile / ncot) * (ncot * HT) + (tile % ncot) * HT;
        float* wsb = p.ws + ((size_t)split * ntile64 + tile64) * 36864 + (size_t)lane * 4;
        static_for<9>([&](auto T_) {
            static_for<NA>([&](auto A_) {
                constexpr int t = decltype(T_)::value, a = decltype(A_)::value;
                *reinterpret_cast<f32x4_t*>(wsb + (size_t)(a >> 2) * 36864 + ((t * 2 + (a & 1)) * 2 + (wave & 1)) * 1024 +
                                            (((a & 3) >> 1) * 2 + (wave >> 1)) * 256) = tuple(T_, A_);
            });
        });
        return;
    }
    // D[i = g*4 + j (co)][col = q (ci)]
    static_for<9>([&](auto T_) {
        static_for<NA>([&](auto A_) {
            constexpr int t = decltype(T_)::value, a = decltype(A_)::value;
            const f32x4_t v = tuple(T_, A_);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int co = co0 + a * 16 + g * 4 + j;
                const int ci = ci0 + wave * 16 + q;
                unsafeAtomicAdd(p.dw + ((size_t)co * 9 + t) * p.Cin + ci, v[j]);
            }
        });
    });
}

// pixel patches of a layer: 8x8 output pixels per stage for stride 1, 4x8 for stride 2
__host__ __device__ __forceinline__ int wgrad_npatch(const gdrn_wgrad_params& p) {
    return (p.M / (p.Ho * p.Wo)) * (p.Ho / (p.stride == 2 ? 4 : 8)) * (p.Wo >> 3);
}

__global__ __launch_bounds__(256, 2) void conv3x3_wgrad_kernel(const gdrn_wgrad_params p, int npatch, int nsplit) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];  // 2 x (dY patch | X patch)
    if (p.stride == 2) wgrad_tile<2, 4>(p, xcd_remap(blockIdx.x, gridDim.x), npatch, nsplit, smem);
    else wgrad_tile<1, 4>(p, xcd_remap(blockIdx.x, gridDim.x), npatch, nsplit, smem);
}

// Grouped launch: the weight gradients of several layers (one gradient bucket) in one grid.  Weight gradients are off
// the critical path of the backward pass, so the engine defers them to the end of their bucket and gives every
// workgroup the same number of pixel patches: the grid fills the chip with far fewer pixel-range splits per layer than a
// per-layer launch needs (8x less partial-tile traffic), and there is one tail instead of one per layer.
__global__ __launch_bounds__(256, 2) void conv3x3_wgrad_multi_kernel(const gdrn_wgrad_params* __restrict__ tasks,
                                                                    const int* __restrict__ blk_start, int ntasks) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    int lo = 0, hi = ntasks;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (blk_start[mid] <= bid) lo = mid; else hi = mid;
    }
    lo = __builtin_amdgcn_readfirstlane(lo);
    const gdrn_wgrad_params p = tasks[lo];
    const int npatch = wgrad_npatch(p);
    if (p.stride == 2) wgrad_tile<2, 4>(p, bid - blk_start[lo], npatch, p.splits, smem);
    else wgrad_tile<1, 4>(p, bid - blk_start[lo], npatch, p.splits, smem);
}

// Sum the workspace partials of one 16(co) x 16(ci) x 9(tap) unit per workgroup and write it in the parameter's layout
// (for OIHW gradients 16 runs of 144 contiguous floats); LDS turns fragment order into that layout.
__global__ __launch_bounds__(256) void wgrad_reduce_multi_kernel(const gdrn_wreduce_task* __restrict__ tasks,
                                                                 const int* __restrict__ blk_start, int ntasks) {
    __shared__ float tile_s[2304];
    int lo = 0, hi = ntasks;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (blk_start[mid] <= (int)blockIdx.x) lo = mid; else hi = mid;
    }
    const gdrn_wreduce_task k = tasks[lo];
    const int unit = blockIdx.x - blk_start[lo];       // (tile, wv, a, b)
    const int b = unit & 1, a = (unit >> 1) & 1, wv = (unit >> 2) & 3, tile = unit >> 4;
    const int ncot = k.Cout / 64, ntile = ncot * (k.Cin / 64);
    // 9 taps x 64 fragment lanes = 576 float4 columns; a thread sums one column over all splits (4 loads in flight),
    // so the memory parallelism does not depend on the split count
    const float* base = k.ws + (size_t)tile * 36864 + (size_t)((a * 2 + b) * 4 + wv) * 256;
    const size_t sstride = (size_t)ntile * 36864;
    for (int idx = threadIdx.x; idx < 576; idx += 256) {
        const int t = idx >> 6, ln = idx & 63;
        const float* src = base + t * 4096 + ln * 4;
        f32x4_t a0 = f32x4_t{0.f, 0.f, 0.f, 0.f}, a1 = a0, a2 = a0, a3 = a0;
        int sp = 0;
        for (; sp + 4 <= k.nsplit; sp += 4) {
            a0 += *reinterpret_cast<const f32x4_t*>(src + (size_t)sp * sstride);
            a1 += *reinterpret_cast<const f32x4_t*>(src + (size_t)(sp + 1) * sstride);
            a2 += *reinterpret_cast<const f32x4_t*>(src + (size_t)(sp + 2) * sstride);
            a3 += *reinterpret_cast<const f32x4_t*>(src + (size_t)(sp + 3) * sstride);
        }
        for (; sp < k.nsplit; ++sp) a0 += *reinterpret_cast<const f32x4_t*>(src + (size_t)sp * sstride);
        a0 += a1;
        a2 += a3;
        a0 += a2;
        const int g = ln >> 4, q = ln & 15;
#pragma unroll
        for (int j = 0; j < 4; ++j) tile_s[((g * 4 + j) * 16 + q) * 9 + t] = a0[j];
    }
    __syncthreads();
    const int co0 = (tile % ncot) * 64 + (wv >> 1) * 32 + a * 16, ci0 = (tile / ncot) * 64 + (wv & 1) * 32 + b * 16;
    const int civ = k.cin_valid > 0 ? k.cin_valid : k.Cin;  // the operand's channels beyond the parameter's (zero padding) have no gradient slot
    for (int i = threadIdx.x; i < 2304; i += 256) {
        const int col = i / 144, r = i - col * 144, cil = r / 9, t = r - cil * 9;
        if (ci0 + cil < civ)
            k.dst[(long long)(co0 + col) * k.s_co + (long long)(ci0 + cil) * k.s_ci + (long long)t * k.s_t] = tile_s[i];
    }
}

}  // namespace

// 1 if the shape is covered by the halo weight-gradient kernel
extern "C" int gdrn_conv3x3_wgrad_ok(const gdrn_wgrad_params* p) {
    if (!p) return 0;
    if (!(p->dtype == GDRN_DT_H16 && p->KH == 3 && p->KW == 3 && p->pad == 1 && (p->Cin % 64) == 0 && (p->Cout % 64) == 0 &&
          (p->x_cs % 8) == 0 && (p->dy_cs % 8) == 0 && (p->Wo % 8) == 0))
        return 0;
    if (p->variant != 0) return 0;   // (GDRN_WGRAD_W128, the 128 x 64 tile of rounds 4-5, is gone: r6)
    if (p->stride == 1) return p->Hi == p->Ho && p->Wi == p->Wo && (p->Ho % 8) == 0;
    if (p->stride == 2) return p->Hi == 2 * p->Ho && p->Wi == 2 * p->Wo && (p->Ho % 4) == 0;
    return 0;
}

// workgroup tiles of a layer (64 x 64)
static inline int wgrad_tiles(const gdrn_wgrad_params& p) { return (p.Cout / 64) * (p.Cin / 64); }

// number of pixel-range splits the launcher uses for these params (p->splits <= 0: automatic); the workspace of the
// p->ws path holds splits * Cout * Cin * 9 floats
extern "C" int gdrn_conv3x3_wgrad_splits(const gdrn_wgrad_params* pp) {
    if (!pp || !gdrn_conv3x3_wgrad_ok(pp)) return 0;
    const gdrn_wgrad_params& p = *pp;
    const int hw = p.Ho * p.Wo;
    if (p.M <= 0 || p.M % hw) return 0;
    const int npatch = wgrad_npatch(p);
    const int tiles = wgrad_tiles(p);
    int splits = p.splits;
    // one partial tile per workgroup either way: target one workgroup per CU (two when the partials are plain stores and two fit a CU)
    if (splits <= 0) splits = std::max(1, std::min(npatch / 4 > 0 ? npatch / 4 : 1, cdiv(p.ws ? 512 : 256, tiles)));
    splits = std::min(splits, npatch);
    const int per = cdiv(npatch, splits);
    return cdiv(npatch, per);  // no empty split: every workspace slab gets written
}

extern "C" int gdrn_conv3x3_wgrad(const gdrn_wgrad_params* pp, void* stream) {
    if (!pp || !pp->x || !pp->dy || (!pp->dw && !pp->ws)) return GDRN_ERR_ARG;
    if (!gdrn_conv3x3_wgrad_ok(pp)) return GDRN_ERR_SHAPE;
    const gdrn_wgrad_params& p = *pp;
    const int hw = p.Ho * p.Wo;
    if (p.M <= 0 || p.M % hw) return GDRN_ERR_SHAPE;
    const int npatch = wgrad_npatch(p);
    const int tiles = wgrad_tiles(p);
    const int splits = gdrn_conv3x3_wgrad_splits(pp);
    if (splits <= 0) return GDRN_ERR_SHAPE;
    constexpr size_t smem = 2 * (size_t)stage_bytes<4>();
    static std::once_flag once;
    static hipError_t attr_err = hipSuccess;
    std::call_once(once, [] {
        attr_err = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_wgrad_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    });
    if (attr_err != hipSuccess) return GDRN_ERR_LAUNCH;
    GDRN_LAUNCH(conv3x3_wgrad_kernel, dim3(tiles * splits), dim3(256), smem, reinterpret_cast<hipStream_t>(stream), p, npatch,
                       splits);
    GDRN_CHECK_LAUNCH();
    return GDRN_OK;
}

// lds_bytes > the kernel's own 64 KiB: the launch requests that much dynamic LDS, i.e. limits itself to ONE workgroup per CU and leaves the
// rest of the CU (LDS, one wave slot per SIMD, half the register file) to another stream's kernels -- the engine runs a bucket's grouped
// weight gradient on a side stream under the next bucket's chain of small-map data-gradient kernels (one workgroup per CU, MFMA pipe 20 % busy).
extern "C" int gdrn_conv3x3_wgrad_multi_lds(const gdrn_wgrad_params* tasks_dev, const int* blk_start_dev, int ntasks, int nblocks, int lds_bytes,
                                            void* stream) {
    if (!tasks_dev || !blk_start_dev || ntasks <= 0 || nblocks <= 0) return GDRN_ERR_ARG;
    constexpr size_t smem0 = 2 * (size_t)stage_bytes<4>();
    const size_t smem = std::max(smem0, (size_t)std::max(lds_bytes, 0));
    if (smem > 160 * 1024) return GDRN_ERR_ARG;
    {   // the attribute only ever grows, under a lock (the unguarded static was a race between host threads)
        static std::mutex mu;
        static size_t attr_set = 0;
        std::lock_guard<std::mutex> lk(mu);
        if (attr_set < smem) {
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_wgrad_multi_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)smem) != hipSuccess)
                return GDRN_ERR_LAUNCH;
            attr_set = smem;
        }
    }
    GDRN_LAUNCH(conv3x3_wgrad_multi_kernel, dim3(nblocks), dim3(256), smem, reinterpret_cast<hipStream_t>(stream), tasks_dev,
                       blk_start_dev, ntasks);
    GDRN_CHECK_LAUNCH();
    return GDRN_OK;
}

extern "C" int gdrn_conv3x3_wgrad_multi(const gdrn_wgrad_params* tasks_dev, const int* blk_start_dev, int ntasks, int nblocks, void* stream) {
    return gdrn_conv3x3_wgrad_multi_lds(tasks_dev, blk_start_dev, ntasks, nblocks, 0, stream);
}

extern "C" int gdrn_wgrad_reduce_multi(const gdrn_wreduce_task* tasks_dev, const int* blk_start_dev, int ntasks, int nblocks, void* stream) {
    if (!tasks_dev || !blk_start_dev || ntasks <= 0 || nblocks <= 0) return GDRN_ERR_ARG;
    GDRN_LAUNCH(wgrad_reduce_multi_kernel, dim3(nblocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), tasks_dev,
                       blk_start_dev, ntasks);
    GDRN_CHECK_LAUNCH();
    return GDRN_OK;
}
