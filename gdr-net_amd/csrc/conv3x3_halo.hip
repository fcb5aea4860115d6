// Halo-tiled 3x3 stride-1 pad-1 convolution on MFMA for gfx950 (NHWC, fp32 accumulate) -- the layers that
// carry ~90 % of the path's MACs: every BasicBlock conv of ResNet-34 except the three stride-2 ones
// (resnet_backbone.py:69-80) and the six 256->256 head convs (cdpn_rot_head_region.py:103-123), forward
// and data-gradient (the latter with tap-flipped, transposed weights).
//
// Structure (what the PMC counters of the generic gather kernel asked for: it spent 41 % of its wave
// cycles parked at s_waitcnt / s_barrier and issued 4 VALU per MFMA for swizzled LDS addressing):
//   * a workgroup (4 waves) owns a TH x TW patch of output pixels of one image and BN output channels;
//     wave w owns ALL pixels x channels [w*BN/4, (w+1)*BN/4): no weight is loaded twice in a workgroup;
//   * the (TH+2) x (TW+2) input patch of one 128-byte channel chunk lives in LDS (zero-filled halo, rows of
//     144 bytes = 128 + 16 pad: 16 consecutive pixels hit 16 distinct 16-byte bank slots).  The nine taps
//     read it at shifted positions -- every ds_read_b128 is `lane base + immediate`, no address VALU;
//   * the weights never touch LDS: they are pre-packed FRAGMENT-MAJOR (gdrn_pack_wfrag: one contiguous
//     1 KiB block per (16 channels, tap, chunk, k-step) holding exactly what the 64 lanes of an MFMA A
//     operand need), so a wave fetches its operand with one fully coalesced global_load_dwordx4 from L2,
//     three stages ahead of use in a register ring;
//   * hence NO barrier inside a channel chunk: one __syncthreads per 9 taps (patch double buffer swap).
// MFMA: v_mfma_f32_16x16x32_bf16 / v_mfma_f32_16x16x4_f32, weights = A operand, pixels = B operand, so a
// lane ends with 4 consecutive output channels of one pixel (8/16-byte stores).
#include <hip/amd_detail/amd_hip_unsafe_atomics.h>

#include <algorithm>
#include <cstdlib>
#include <mutex>

#include "common.h"
#include "halo_xf.h"
#include "../../include/gdrn_hip.h"

namespace {

constexpr int ROWB = 128;   // bytes of K per stage row
// LDS patch layout: the eight 16-byte channel granules of a pixel's 128-byte chunk are split into an EVEN-granule
// and an ODD-granule array ([pixel][4 granules + 16 B pad], pitch 80 B), the odd array a multiple of 256 B after the
// even one.  ds_read_b128 services lanes {0-3,12-15} of one k-group together with lanes {4-11} of the NEXT k-group
// (hardware lane groups); with one [pixel][8 granules] array those two sets collide on 7 of 8 banks whatever the pitch
// (measured: SQ_LDS_BANK_CONFLICT = 45 % of the LDS cycles).  Even/odd k-groups in bank-aligned arrays make the two
// sets land on the slots of DIFFERENT pixels -> conflict-free for every tap shift (16-pixel-wide tiles).
constexpr int PITCH = 80;   // LDS pixel pitch inside one half array
__host__ __device__ constexpr int half_bytes(int ppix) { return (ppix * PITCH + 255) / 256 * 256; }

template <typename T>
__device__ __forceinline__ f32x4_t mma_step(uint4 a, uint4 b, f32x4_t c);
template <>
__device__ __forceinline__ f32x4_t mma_step<bf16_t>(uint4 a, uint4 b, f32x4_t c) {
    return GDRN_MFMA16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c);
}

// fp32 (parity mode): a 16-byte operand granule = four k-steps of v_mfma_f32_16x16x4_f32 (lane group g supplies k = g of each); ks_ selects the
// granule's float.  The four MFMAs of one granule pair are issued ACROSS the accumulator tuples (see MM below), not back to back on one.
__device__ __forceinline__ f32x4_t mma_f32(uint4 a, uint4 b, f32x4_t c, int ks_) {
    const unsigned av = ks_ == 0 ? a.x : (ks_ == 1 ? a.y : (ks_ == 2 ? a.z : a.w));
    const unsigned bv = ks_ == 0 ? b.x : (ks_ == 1 ? b.y : (ks_ == 2 ? b.z : b.w));
    return __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(av), __uint_as_float(bv), c, 0, 0, 0);
}

// fragment-major weight packing: granule (16 B) permutation of the row-major [rows][9][Cin] operand
template <typename T>
__global__ __launch_bounds__(256) void pack_wfrag_kernel(const T* __restrict__ src, T* __restrict__ dst, int rows, int Cin) {
    constexpr int EPS = ROWB / (int)sizeof(T);
    constexpr int GE = 16 / (int)sizeof(T);  // elements per 16-byte granule
    const int kch = Cin / EPS;
    const long long total = (long long)rows * 9 * Cin / GE;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        // destination granule index: ((((cb*9 + tap)*kch + kc)*2 + ks)*64 + lane)
        const int lane = (int)(i & 63);
        long long r = i >> 6;
        const int ks = (int)(r & 1); r >>= 1;
        const int kc = (int)(r % kch); r /= kch;
        const int tap = (int)(r % 9);
        const int cb = (int)(r / 9);
        // row of block cb, fragment row r: the FNp blocks of a 16*FNp-row group interleave in units of 4 rows, so that the lane of an MFMA
        // result that holds rows 4g..4g+3 of block a = 0 holds the NEXT four rows in block a = 1: 8 contiguous output channels per lane
        // (one 16-byte access in the conv's epilogue instead of two 8-byte ones).  FNp = 1 for the 64-row operands (64-channel tile).
        const int FNp = (rows <= 64 || sizeof(T) == 4) ? 1 : 2, rr = lane & 15;   // (fp32: always the 64-channel tile, rows in natural order)
        const int co = (cb / FNp) * 16 * FNp + (rr >> 2) * (4 * FNp) + (cb % FNp) * 4 + (rr & 3), g = lane >> 4;
        const size_t so = ((size_t)(co * 9 + tap) * Cin + (size_t)kc * EPS) + (size_t)(ks * 4 + g) * GE;
        *reinterpret_cast<uint4*>(dst + i * GE) = *reinterpret_cast<const uint4*>(src + so);
    }
}

#ifdef HALO_DBG
__device__ unsigned long long* g_halo_dbg;
__device__ __forceinline__ unsigned long long halo_clock() {
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return t;
}
struct HaloDbg {
    unsigned long long t[6] = {0, 0, 0, 0, 0, 0};
    int slot, lane;
    __device__ ~HaloDbg() {
        t[4] = halo_clock();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        t[5] = halo_clock();
        if (lane == 0 && g_halo_dbg)
            for (int i = 0; i < 6; ++i) g_halo_dbg[(size_t)slot * 8 + i] = t[i];
    }
};
extern "C" int gdrn_halo_set_dbg(unsigned long long* buf) { return hipMemcpyToSymbol(HIP_SYMBOL(g_halo_dbg), &buf, sizeof(buf)) == hipSuccess ? 0 : -3; }
#define HALO_STAMP(i_) hdbg_.t[i_] = halo_clock();
#else
#define HALO_STAMP(i_)
#endif

// KS = 2 (round 5): the EIGHT-wave form of the same tile for grids that give a CU one workgroup (8x8 - 16x16 maps at bs = 64: 256 workgroups =
// one wave per SIMD, every LDS / L2 / MFMA latency exposed).  Waves 4-7 are a second copy of waves 0-3 -- same pixels, same channels, same
// LDS patch -- that takes k-step 1 of every tap stage while the first copy takes k-step 0: each wave walks HALF the reduction (one 1 KiB
// weight block and FM fragment reads per tap instead of two), the weight stream per CU is the same, the patch is staged once by all 512
// threads, and the two partial accumulator tiles meet once, through LDS, in front of the epilogue.
template <typename T, int TH, int TW, int BN, int XF, int KS = 1>
__global__ __launch_bounds__(256 * KS, KS == 2 ? 1 : ((BN == 64 && TW == 16 && sizeof(T) == 2) ? 3 : 2)) void conv3x3_halo_kernel(const gdrn_conv_params p) {
    static_assert(KS == 1 || (KS == 2 && sizeof(T) == 2 && BN == 128), "the 8-wave form: 16-bit, 128-channel tile");
    constexpr int EPS = ROWB / (int)sizeof(T);
    constexpr int BM = TH * TW;
    constexpr int PW = TW + 2, PH = TH + 2, PPIX = PH * PW;
    constexpr int HB = half_bytes(PPIX);      // odd-granule array offset
    constexpr int PBYTES = 2 * HB;
    constexpr int PSEG = PPIX * 8;            // 16-byte segments of one patch chunk
    constexpr int PSLICE = XF ? ((PSEG + 8) / 9 + 7) / 8 * 8 : (PSEG + 8) / 9;    // segments fetched per tap stage
    static_assert(PSLICE <= 256, "one patch segment per thread per stage");
    static_assert(XF == 0 || sizeof(T) == 2, "operand transforms are bf16 only");
    constexpr int FM = BM / 16;               // pixel fragments per wave (all pixels)
    constexpr int FN = BN / 64;               // 16-channel fragments per wave
    constexpr int WQ = FN * 2 / KS;           // weight 1-KiB loads per wave per stage (FN frags x 2 k-steps; one k-step per wave group in the 8-wave form)
    constexpr int NSL = KS == 2 ? 5 : 9;      // patch slices a thread moves per chunk (8-wave form: slices of the group's parity)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];  // 2 x patch

    const int lane = threadIdx.x & 63;
    const int grp = KS == 2 ? __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 8) : 0;   // wave group = k-step of every tap stage
    const int tid = threadIdx.x & 255, wave = tid >> 6;    // tid: thread index inside the wave group
    const int g = lane >> 4, r16 = lane & 15;
#ifdef HALO_DBG
    HaloDbg hdbg_;
    hdbg_.slot = (int)blockIdx.x * 4 + wave;
    hdbg_.lane = lane;
#endif
    HALO_STAMP(0)
    // the data-gradient chain shares its CUs with a side-stream weight-gradient launch (engine: wgrad_stream): its waves go first
    __builtin_amdgcn_s_setprio(2);

    const int NTn = (p.Cout + BN - 1) / BN;
    int bid = blockIdx.x;
    {   // XCD-aware order: neighbouring patches / the N tiles of a patch share one L2
        const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = bid & 7, j = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
    }
    const int nt = bid % NTn;
    const int mt = bid / NTn;  // pixel-tile index (stats row)
    int t = mt;
    const int tiles_x = p.Wo / TW, tiles_y = p.Ho / TH;
    const int tx = t % tiles_x; t /= tiles_x;
    const int ty = t % tiles_y;
    const int n = t / tiles_y;
    const int co0 = nt * BN;
    const int y0 = ty * TH, x0 = tx * TW;

    const int kch = p.Cin / EPS;
    const char* xg = reinterpret_cast<const char*>(p.x);
    // this wave's fragment-major weight stream: block index ((cb*9 + tap)*kch + kc)*2 + ks, 1 KiB each
    const int cb0 = (co0 + wave * (BN / 4)) / 16;
    const char* wl = reinterpret_cast<const char*>(p.w) + (size_t)lane * 16 + (size_t)(grp << 10);

    // weight stream: stage s = kc*9 + tap; per stage WQ loads of 1 KiB
    auto wptr = [&](int kc, int tap, int a, int ks) -> const uint4* {
        return reinterpret_cast<const uint4*>(wl + ((size_t)((((cb0 + a) * 9 + tap) * kch + kc) * 2 + ks) << 10));
    };
    uint4 wq0[WQ], wq1[WQ], wq2[WQ];  // 3-stage register ring (slot = tap % 3)

#define LOADW(dst, kc_, tap_)                                                   \
    {                                                                           \
        _Pragma("unroll") for (int a_ = 0; a_ < FN; ++a_) {                     \
            if constexpr (KS == 1) {                                            \
                dst[a_ * 2 + 0] = *wptr(kc_, tap_, a_, 0);                      \
                dst[a_ * 2 + 1] = *wptr(kc_, tap_, a_, 1);                      \
            } else dst[a_] = *wptr(kc_, tap_, a_, 0);                           \
        }                                                                       \
    }
    // the first three stages' weights go out before the patch geometry below is computed: their L2 latency hides under ~300
    // address instructions instead of following them
    LOADW(wq0, 0, 0) LOADW(wq1, 0, 1) LOADW(wq2, 0, 2)

    // ---- patch slice geometry of this thread (slice st = linear segment ids [st*PSLICE, (st+1)*PSLICE))
    // (8-wave form: wave group grp moves the slices st = 2j + grp; arrays and masks are indexed by j)
    unsigned poff[NSL];
    int pdst[NSL];
    unsigned pokm = 0, pinm = 0, ppm = 0;  // slice st: input pixel inside the image / one of this tile's own (interior) pixels / slot exists
#pragma unroll
    for (int sj = 0; sj < NSL; ++sj) {
        const int st = KS == 2 ? 2 * sj + grp : sj;
        const int id = st * PSLICE + tid;
        const int pp = id >> 3, sg = id & 7;
        const int py = pp / PW, px = pp - py * PW;
        const int iy = y0 + py - 1, ix = x0 + px - 1;
        const bool inpatch = tid < PSLICE && id < PSEG && st < 9;
        const bool ok = inpatch && iy >= 0 && iy < p.Hi && ix >= 0 && ix < p.Wi;
        const int iyc = min(max(iy, 0), p.Hi - 1), ixc = min(max(ix, 0), p.Wi - 1);
        poff[sj] = (unsigned)((n * p.Hi + iyc) * p.Wi + ixc) * (unsigned)p.x_cs * (unsigned)sizeof(T) + sg * 16;
        pdst[sj] = inpatch ? (pp * PITCH + (sg & 1) * HB + (sg >> 1) * 16) : -1;
        pokm |= ok ? (1u << sj) : 0u;
        if constexpr (XF != 0) {
            pinm |= (inpatch && py >= 1 && py <= TH && px >= 1 && px <= TW) ? (1u << sj) : 0u;
            ppm |= inpatch ? (1u << sj) : 0u;
        }
    }
    // PSLICE % 8 == 0 in the transform instantiations: the LDS slot of slice st is pdst0 + st * PDSTEP (no per-slice register)
    constexpr int PDSTEP = (PSLICE / 8) * PITCH;
    const int pdst0 = (tid >> 3) * PITCH + (tid & 1) * HB + ((tid & 7) >> 1) * 16 + grp * PDSTEP;
    // slice st_ belongs to this thread's wave group / its index in the per-thread arrays and masks
#define OWN(st_) (KS == 1 || (((st_) & 1) == grp))
#define SJ(st_) (KS == 1 ? (st_) : ((st_) >> 1))
    // operand transform state: LDS table of the per-channel vectors (built below, behind the patch buffers), second input,
    // optional copy-out of the transformed tile (first channel tile of a pixel tile only)
    const float* xtab = nullptr;
    const char* xg2 = nullptr;
    char* xo = nullptr;
    float xlo = 0.f;
    if constexpr (XF != 0) {
        float* tabw = reinterpret_cast<float*>(smem + (kch == 1 ? 1 : 2) * PBYTES);
        for (int c = (int)threadIdx.x; c < p.Cin; c += 256 * KS) {
            tabw[c] = p.xf_a ? p.xf_a[c] : 1.f;
            tabw[p.Cin + c] = p.xf_c[c];
            if constexpr (XF >= 2) tabw[2 * p.Cin + c] = p.xf_b ? p.xf_b[c] : 1.f;
            if constexpr (XF == 2) tabw[3 * p.Cin + c] = p.xf_c2 ? p.xf_c2[c] : 0.f;
            if constexpr (XF == 4) { tabw[3 * p.Cin + c] = p.xf_msc[c]; tabw[4 * p.Cin + c] = p.xf_msh[c]; }
        }
        xtab = tabw + (tid & 7) * 8;   // + kc * 64: this thread's 8 channels of chunk kc
        xg2 = reinterpret_cast<const char*>(p.xf_x2);
        xo = (nt == 0) ? reinterpret_cast<char*>(p.xf_out) : nullptr;
        xlo = p.xf_relu ? 0.f : -__builtin_inff();
    }

    // ---- pixel-fragment lane base inside a patch: fragment b, lane column r16 -> pixel (oy, ox)
    int lbase;
    if constexpr (TW == 16) lbase = r16 * PITCH + (g & 1) * HB + (g >> 1) * 16;                        // oy = b, ox = r16
    else lbase = ((r16 >> 3) * PW + (r16 & 7)) * PITCH + (g & 1) * HB + (g >> 1) * 16;             // oy = 2b + (r16>>3)
    lbase += grp * 32;   // 8-wave form: the group's k-step of every stage
    constexpr int FROW = (TW == 16) ? PW * PITCH : 2 * PW * PITCH;                    // byte step per fragment b

    f32x4_t acc[FN][FM];
#pragma unroll
    for (int a = 0; a < FN; ++a)
#pragma unroll
        for (int b = 0; b < FM; ++b) acc[a][b] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    // fp32 (parity) mode: every 32-deep tap stage accumulates in a fresh register tile that is then added to the running sum, so that no
    // fp32 addition chain is longer than 32 + #stages (the generic kernel's rule, conv_gemm.hip: the pose parity sits at the fp32
    // summation-order noise floor); FN = 1 in that mode (64-channel tile), so the second tile costs 32 registers
    f32x4_t part[sizeof(T) == 4 ? FN : 1][sizeof(T) == 4 ? FM : 1];

    // next-chunk patch slices in flight: slice s is loaded at stage s-1 (0 and 1 at stage 0) and written to LDS at stage s+2, i.e.
    // three tap stages (~1500 cycles) after its load: with a single register written one stage later every stage waited on HBM
    uint4 rp0 = make_uint4(0, 0, 0, 0), rp1 = rp0, rp2 = rp0;
    uint4 rq0 = rp0, rq1 = rp0, rq2 = rp0;  // second input of the operand transform (XF >= 2)
#define RP(i_) (((i_) % 3) == 0 ? rp0 : (((i_) % 3) == 1 ? rp1 : rp2))
#define RQ(i_) (((i_) % 3) == 0 ? rq0 : (((i_) % 3) == 1 ? rq1 : rq2))

    // XF == 0: out-of-image granules are zeroed at the load; XF != 0: at the write, AFTER the transform (the padding is a
    // property of the conv's input v, and v(0) != 0)
#define LOADP(dst, dst2, kc_, st_)                                              \
    if (OWN(st_)) {                                                             \
        const uint4 v_ = *reinterpret_cast<const uint4*>(xg + (poff[SJ(st_)] + (unsigned)((kc_) * ROWB))); \
        if constexpr (XF == 0) dst = ((pokm >> SJ(st_)) & 1u) ? v_ : make_uint4(0, 0, 0, 0);          \
        else dst = v_;                                                          \
        if constexpr (XF >= 2) dst2 = *reinterpret_cast<const uint4*>(xg2 + (poff[SJ(st_)] + (unsigned)((kc_) * ROWB))); \
    }
#define WRITEP(src, src2, pb_, st_, kc_)                                        \
    if (OWN(st_)) {                                                             \
        if constexpr (XF == 0) {                                                \
            if (pdst[SJ(st_)] >= 0) *reinterpret_cast<uint4*>(smem + (pb_) * PBYTES + pdst[SJ(st_)]) = src; \
        } else {                                                                \
            if ((ppm >> SJ(st_)) & 1u) {                                        \
                uint4 t_ = xf_apply<XF>(src, src2, xtab + (kc_) * EPS, p.Cin, xlo); \
                t_ = ((pokm >> SJ(st_)) & 1u) ? t_ : make_uint4(0, 0, 0, 0);   \
                *reinterpret_cast<uint4*>(smem + (pb_) * PBYTES + pdst0 + SJ(st_) * (PDSTEP * KS)) = t_; \
                if (xo != nullptr && ((pinm >> SJ(st_)) & 1u))                  \
                    *reinterpret_cast<uint4*>(xo + (poff[SJ(st_)] + (unsigned)((kc_) * ROWB))) = t_; \
            }                                                                   \
        }                                                                       \
    }

    // ---- prologue: patch of chunk 0 (nine slices in flight together); the weights of stages 0..2 are already in flight
    {
        uint4 q0, q1, q2, q3, q4, q5, q6, q7, q8;
        uint4 u0 = rp0, u1 = rp0, u2 = rp0, u3 = rp0, u4 = rp0, u5 = rp0, u6 = rp0, u7 = rp0, u8 = rp0;
        LOADP(q0, u0, 0, 0) LOADP(q1, u1, 0, 1) LOADP(q2, u2, 0, 2) LOADP(q3, u3, 0, 3) LOADP(q4, u4, 0, 4)
        LOADP(q5, u5, 0, 5) LOADP(q6, u6, 0, 6) LOADP(q7, u7, 0, 7) LOADP(q8, u8, 0, 8)
        if constexpr (XF != 0) __syncthreads();  // the transform table is complete
        WRITEP(q0, u0, 0, 0, 0) WRITEP(q1, u1, 0, 1, 0) WRITEP(q2, u2, 0, 2, 0) WRITEP(q3, u3, 0, 3, 0) WRITEP(q4, u4, 0, 4, 0)
        WRITEP(q5, u5, 0, 5, 0) WRITEP(q6, u6, 0, 6, 0) WRITEP(q7, u7, 0, 7, 0) WRITEP(q8, u8, 0, 8, 0)
    }
    HALO_STAMP(1)
    __syncthreads();
    HALO_STAMP(2)

    // One tap stage = 2 k-steps x (FN weight frags in registers) x (FM pixel frags from LDS at immediate offsets), software
    // pipelined over the k-steps: the LDS reads of k-step 1 are issued before the MFMAs of k-step 0, and the reads of the
    // NEXT tap's k-step 0 before the MFMAs of k-step 1, so a wave's LDS latency hides under its own MFMAs (hipcc
    // otherwise issues all 16 reads of a stage up front and the first MFMA waits ~150 cycles per stage).
    uint4 fbA[FM], fbB[FM];
#define RD(dst_, TAP_, KS_)                                                                                     \
    {                                                                                                           \
        constexpr int tsh_ = ((TAP_) / 3) * PW * PITCH + ((TAP_) % 3) * PITCH;                                  \
        _Pragma("unroll") for (int b_ = 0; b_ < FM; ++b_)                                                       \
            dst_[b_] = *reinterpret_cast<const uint4*>(pcur + (b_ * FROW + tsh_ + (KS_) * 32));                 \
    }
#define MM(WQ_, KS_, src_)                                                                                      \
    {                                                                                                           \
        if constexpr (sizeof(T) == 4) {                                                                         \
            _Pragma("unroll") for (int k4_ = 0; k4_ < 4; ++k4_)                                                 \
                _Pragma("unroll") for (int a_ = 0; a_ < FN; ++a_)                                               \
                    _Pragma("unroll") for (int b_ = 0; b_ < FM; ++b_)                                           \
                        part[a_][b_] = mma_f32(WQ_[a_ * 2 + (KS_)], src_[b_], part[a_][b_], k4_);               \
        } else {                                                                                                \
            _Pragma("unroll") for (int a_ = 0; a_ < FN; ++a_)                                                   \
                _Pragma("unroll") for (int b_ = 0; b_ < FM; ++b_)                                               \
                    acc[a_][b_] = mma_step<T>(WQ_[a_ * (2 / KS) + (KS_)], src_[b_], acc[a_][b_]);               \
        }                                                                                                       \
    }
    // stage TAP_ of chunk kc: compute, refill the ring slot for stage +3, move one slice of the next patch
#define STEP(WQ_, TAP_)                                                                                         \
    {                                                                                                           \
        if (more_p) {                                                                                           \
            if constexpr ((TAP_) >= 2) WRITEP(RP((TAP_) + 1), RQ((TAP_) + 1), pb ^ 1, (TAP_) - 2, kc + 1)       \
            if constexpr ((TAP_) == 0) { LOADP(rp0, rq0, kc + 1, 0) LOADP(rp1, rq1, kc + 1, 1) }                \
            else if constexpr ((TAP_) <= 7) LOADP(RP((TAP_) + 1), RQ((TAP_) + 1), kc + 1, (TAP_) + 1)           \
        }                                                                                                       \
        if constexpr (KS == 2) {   /* one k-step per tap and wave: fragments of even taps in fbA, of odd taps in fbB */      \
            if constexpr ((TAP_) < 8) {                                                                         \
                if constexpr ((TAP_) & 1) RD(fbA, (TAP_) + 1, 0) else RD(fbB, (TAP_) + 1, 0)                    \
            }                                                                                                   \
            if constexpr ((TAP_) & 1) MM(WQ_, 0, fbB) else MM(WQ_, 0, fbA)                                      \
        } else {                                                                                                \
        RD(fbB, TAP_, 1)                                                                                        \
        if constexpr (sizeof(T) == 4) {                                                                         \
            _Pragma("unroll") for (int a_ = 0; a_ < FN; ++a_)                                                   \
                _Pragma("unroll") for (int b_ = 0; b_ < FM; ++b_) part[a_][b_] = f32x4_t{0.f, 0.f, 0.f, 0.f};   \
        }                                                                                                       \
        MM(WQ_, 0, fbA)                                                                                         \
        \
        if constexpr ((TAP_) < 8) RD(fbA, (TAP_) + 1, 0)                                                        \
        \
        MM(WQ_, 1, fbB)                                                                                         \
        if constexpr (sizeof(T) == 4) {                                                                         \
            _Pragma("unroll") for (int a_ = 0; a_ < FN; ++a_)                                                   \
                _Pragma("unroll") for (int b_ = 0; b_ < FM; ++b_) acc[a_][b_] += part[a_][b_];                  \
        }                                                                                                       \
        }                                                                                                       \
        {                                                                                                       \
            constexpr int nt_ = ((TAP_) + 3) % 9;                                                               \
            const int nk_ = kc + (((TAP_) + 3) >= 9 ? 1 : 0);                                                   \
            if (nk_ < kch) LOADW(WQ_, nk_, nt_)                                                                 \
        }                                                                                                       \
    }

    for (int kc = 0; kc < kch; ++kc) {
        const int pb = kc & 1;
        const bool more_p = kc + 1 < kch;
        const unsigned char* pcur = smem + pb * PBYTES + lbase;
        RD(fbA, 0, 0)
        STEP(wq0, 0) STEP(wq1, 1) STEP(wq2, 2)
        STEP(wq0, 3) STEP(wq1, 4) STEP(wq2, 5)
        STEP(wq0, 6) STEP(wq1, 7) STEP(wq2, 8)
        if (more_p) { WRITEP(rp1, rq1, pb ^ 1, 7, kc + 1) WRITEP(rp2, rq2, pb ^ 1, 8, kc + 1) }
        __syncthreads();
    }
    HALO_STAMP(3)
#undef LOADW
#undef LOADP
#undef WRITEP
#undef RD
#undef MM
#undef STEP
#undef RP
#undef RQ
#undef OWN
#undef SJ

    if constexpr (KS == 2) {
        // the two halves of the reduction meet: group 1 parks its accumulator tile in LDS (the patch buffers are dead behind the loop's
        // last barrier; [wave][fragment][lane] x 16 bytes = consecutive lanes on consecutive banks) and is done; group 0 adds it and runs
        // the epilogue on the complete sums
        float4* ex = reinterpret_cast<float4*>(smem) + wave * (FN * FM * 64) + lane;
        if (grp == 1) {
#pragma unroll
            for (int a = 0; a < FN; ++a)
#pragma unroll
                for (int b = 0; b < FM; ++b) ex[(a * FM + b) * 64] = make_float4(acc[a][b][0], acc[a][b][1], acc[a][b][2], acc[a][b][3]);
        }
        __syncthreads();
        if (grp == 1) return;
#pragma unroll
        for (int a = 0; a < FN; ++a)
#pragma unroll
            for (int b = 0; b < FM; ++b) {
                const float4 o = ex[(a * FM + b) * 64];
                acc[a][b][0] += o.x; acc[a][b][1] += o.y; acc[a][b][2] += o.z; acc[a][b][3] += o.w;
            }
    }

    // ---- epilogue: lane holds channels c0..c0+3 (c0 = co0 + wave*BN/4 + g*4*FN + a*4) of pixel b*16 + r16
    // Fast path for what the engine actually launches (full channel tiles, bf16 out, bias / ReLU / addend / statistics
    // optional): straight-line code, 32-bit offsets from uniform bases, DPP row sums.  The generic
    // epilogue below costs ~4600 instructions per wave (264 branches) -- more than the MFMA loop of a 256-channel
    // layer (3700) and 5x the loop of a 64-channel layer.
    if constexpr (sizeof(T) == 2) {
        if (p.act <= 1 && !p.out_f32 && (p.Cout % BN) == 0) {
            const int cl = co0 + wave * (BN / 4) + g * (4 * FN);  // + a*4: the lane's 4*FN contiguous channels (row order of gdrn_pack_wfrag)
            // one access per pixel for the lane's 4*FN contiguous channels: 16 bytes on the 128-channel tile (FN = 2), 8 on the 64-channel one
            auto ld_ch = [&](const char* base, unsigned elem_off, uint2 (&dst)[FN]) {
                if constexpr (FN == 2) {
                    const uint4 q = *reinterpret_cast<const uint4*>(base + elem_off * 2u);
                    dst[0] = make_uint2(q.x, q.y);
                    dst[1] = make_uint2(q.z, q.w);
                } else {
#pragma unroll
                    for (int a = 0; a < FN; ++a) dst[a] = *reinterpret_cast<const uint2*>(base + (elem_off + (unsigned)(a * 4)) * 2u);
                }
            };
            auto st_ch = [&](char* base, unsigned elem_off, const uint2 (&src)[FN]) {
                if constexpr (FN == 2) {
                    *reinterpret_cast<uint4*>(base + elem_off * 2u) = make_uint4(src[0].x, src[0].y, src[1].x, src[1].y);
                } else {
#pragma unroll
                    for (int a = 0; a < FN; ++a) *reinterpret_cast<uint2*>(base + (elem_off + (unsigned)(a * 4)) * 2u) = src[a];
                }
            };
            if (p.stats != nullptr) {
                float* srow = p.stats + (size_t)mt * 2 * p.Cout + cl;
#pragma unroll
                for (int a = 0; a < FN; ++a) {
                    float s1[4], s2[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        float u = 0.f, q = 0.f;
#pragma unroll
                        for (int b = 0; b < FM; ++b) { const float v = acc[a][b][j]; u += v; q += v * v; }
                        s1[j] = row16_sum(u);
                        s2[j] = row16_sum(q);
                    }
                    if (r16 == 0) {
                        *reinterpret_cast<float4*>(srow + a * 4) = make_float4(s1[0], s1[1], s1[2], s1[3]);
                        *reinterpret_cast<float4*>(srow + p.Cout + a * 4) = make_float4(s2[0], s2[1], s2[2], s2[3]);
                    }
                }
            }
            // pixel of fragment b, lane column r16: (y0 + b, x0 + r16) for TW = 16, (y0 + 2b + (r16>>3), x0 + (r16&7)) for TW = 8
            const int prow0 = (n * p.Ho + y0 + (TW == 16 ? 0 : (r16 >> 3))) * p.Wo + x0 + (TW == 16 ? r16 : (r16 & 7));
            const int pstep = (TW == 16 ? 1 : 2) * p.Wo;  // pixel-row step per fragment b
            char* yb = reinterpret_cast<char*>(p.y);
            const char* ab = reinterpret_cast<const char*>(p.addend);
            if (p.bnb_x != nullptr) {
                // data gradient feeding a BatchNorm(+ReLU) backward: ReLU mask + the two per-channel sums of that
                // backward here, on the accumulators, instead of a separate pass over (dy, x, mask)
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
                // All tile loads of the epilogue go out first (the ring / fragment registers of the loop are dead: 96 VGPRs of
                // uint2), then the arithmetic and the stores.  Written as load-use-store per fragment row, hipcc kept that order
                // (the stores may alias the loads) and every one of the FM rows waited a full memory round trip: 8 serialized
                // HBM latencies per workgroup.
                constexpr int HB_ = FM / 2;  // in two halves: all FM rows at once (96 VGPRs for three tensors) spilled on the 128-wide tile
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    uint2 xq[HB_][FN], av[HB_][FN], mq[HB_][FN];
#pragma unroll
                    for (int bb = 0; bb < HB_; ++bb) {
                        const unsigned pr = (unsigned)(prow0 + (h * HB_ + bb) * pstep);
                        ld_ch(xb, pr * (unsigned)p.bnb_cs + (unsigned)cl, xq[bb]);
#pragma unroll
                        for (int a = 0; a < FN; ++a) {
                            av[bb][a] = make_uint2(0u, 0u);
                            mq[bb][a] = make_uint2(GDRN_H16_ONE2, GDRN_H16_ONE2);
                        }
                    }
                    if (ab != nullptr) {
#pragma unroll
                        for (int bb = 0; bb < HB_; ++bb) {
                            const unsigned pr = (unsigned)(prow0 + (h * HB_ + bb) * pstep);
                            ld_ch(ab, pr * (unsigned)p.add_cs + (unsigned)cl, av[bb]);
                        }
                    }
                    if (mb != nullptr) {
#pragma unroll
                        for (int bb = 0; bb < HB_; ++bb) {
                            const unsigned pr = (unsigned)(prow0 + (h * HB_ + bb) * pstep);
                            ld_ch(mb, pr * (unsigned)p.bnb_cs + (unsigned)cl, mq[bb]);
                        }
                    }
#pragma unroll
                    for (int bb = 0; bb < HB_; ++bb) {
                        const int b = h * HB_ + bb;
                        const unsigned pr = (unsigned)(prow0 + b * pstep);
                        uint2 ov[FN];
#pragma unroll
                        for (int a = 0; a < FN; ++a) {
                            const uint2 xw = xq[bb][a], aw = av[bb][a], mw = mq[bb][a];
                            const float xv[4] = {h16lo(xw.x), h16hi(xw.x),
                                                 h16lo(xw.y), h16hi(xw.y)};
                            const float mv[4] = {h16lo(mw.x), h16hi(mw.x),
                                                 h16lo(mw.y), h16hi(mw.y)};
                            const float ad[4] = {h16lo(aw.x), h16hi(aw.x),
                                                 h16lo(aw.y), h16hi(aw.y)};
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
                        st_ch(yb, pr * (unsigned)p.y_cs + (unsigned)cl, ov);
                    }
                }
                // per-tile partial rows with plain stores (as the forward statistics); gdrn_bn_bwd_coef adds them up in a fixed order
                // and turns them into the backward's coefficients.  (Atomics from every workgroup onto 16 x 2C addresses
                // cost 25-55 us per launch on the large maps -- same-address atomics run at a few G/s.)
                float* srow = p.bnb_rows + (size_t)mt * 2 * p.Cout + cl;
#pragma unroll
                for (int a = 0; a < FN; ++a) {
                    float u1[4], u2[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) { u1[j] = row16_sum(t1[a][j]); u2[j] = row16_sum(t2[a][j]); }
                    if (r16 == 0) {
                        *reinterpret_cast<float4*>(srow + a * 4) = make_float4(u1[0], u1[1], u1[2], u1[3]);
                        *reinterpret_cast<float4*>(srow + p.Cout + a * 4) = make_float4(u2[0], u2[1], u2[2], u2[3]);
                    }
                }
                return;
            }
            // optional bias (eval mode: the folded BatchNorm shift) and ReLU
            float bq[FN][4];
#pragma unroll
            for (int a = 0; a < FN; ++a) {
                float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
                if (p.bias != nullptr) bv = *reinterpret_cast<const float4*>(p.bias + cl + a * 4);
                bq[a][0] = bv.x; bq[a][1] = bv.y; bq[a][2] = bv.z; bq[a][3] = bv.w;
            }
            const bool relu = p.act == 1;
            if (ab != nullptr) {
                uint2 avq[FM][FN];  // all addend loads first (see the fused BatchNorm path above)
#pragma unroll
                for (int b = 0; b < FM; ++b) {
                    const unsigned pr = (unsigned)(prow0 + b * pstep);
                    ld_ch(ab, pr * (unsigned)p.add_cs + (unsigned)cl, avq[b]);
                }
#pragma unroll
                for (int b = 0; b < FM; ++b) {
                    const unsigned pr = (unsigned)(prow0 + b * pstep);
                    uint2 ov[FN];
#pragma unroll
                    for (int a = 0; a < FN; ++a) {
                        const uint2 av_ = avq[b][a];
                        float v0 = acc[a][b][0] + bq[a][0] + h16lo(av_.x), v1 = acc[a][b][1] + bq[a][1] + h16hi(av_.x);
                        float v2 = acc[a][b][2] + bq[a][2] + h16lo(av_.y), v3 = acc[a][b][3] + bq[a][3] + h16hi(av_.y);
                        if (relu) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); v2 = fmaxf(v2, 0.f); v3 = fmaxf(v3, 0.f); }
                        ov[a] = make_uint2(pack_bf2(v0, v1), pack_bf2(v2, v3));
                    }
                    st_ch(yb, pr * (unsigned)p.y_cs + (unsigned)cl, ov);
                }
            } else {
#pragma unroll
                for (int b = 0; b < FM; ++b) {
                    const unsigned pr = (unsigned)(prow0 + b * pstep);
                    uint2 ov[FN];
#pragma unroll
                    for (int a = 0; a < FN; ++a) {
                        float v0 = acc[a][b][0] + bq[a][0], v1 = acc[a][b][1] + bq[a][1], v2 = acc[a][b][2] + bq[a][2], v3 = acc[a][b][3] + bq[a][3];
                        if (relu) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); v2 = fmaxf(v2, 0.f); v3 = fmaxf(v3, 0.f); }
                        ov[a] = make_uint2(pack_bf2(v0, v1), pack_bf2(v2, v3));
                    }
                    st_ch(yb, pr * (unsigned)p.y_cs + (unsigned)cl, ov);
                }
            }
            return;
        }
    }
    if (p.stats != nullptr) {
#pragma unroll
        for (int a = 0; a < FN; ++a) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float s1 = 0.f, s2 = 0.f;
#pragma unroll
                for (int b = 0; b < FM; ++b) { const float v = acc[a][b][j]; s1 += v; s2 += v * v; }
#pragma unroll
                for (int o = 8; o > 0; o >>= 1) { s1 += __shfl_xor(s1, o, 64); s2 += __shfl_xor(s2, o, 64); }
                const int c = co0 + wave * (BN / 4) + g * (4 * FN) + a * 4 + j;
                if (r16 == 0 && c < p.Cout) {
                    p.stats[((size_t)mt * 2 + 0) * p.Cout + c] = s1;
                    p.stats[((size_t)mt * 2 + 1) * p.Cout + c] = s2;
                }
            }
        }
    }

#pragma unroll
    for (int b = 0; b < FM; ++b) {
        const int m = b * 16 + r16;
        const size_t orow = (size_t)(n * p.Ho + y0 + m / TW) * p.Wo + x0 + (m % TW);
#pragma unroll
        for (int a = 0; a < FN; ++a) {
            const int c0 = co0 + wave * (BN / 4) + g * (4 * FN) + a * 4;
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

constexpr int XF_MAX_CIN = 512;  // channels of the operand-transform table

template <typename T, int TH, int TW, int BN, int XF, int KS = 1>
int launch(const gdrn_conv_params& p, int N, hipStream_t st) {
    constexpr size_t smem = 2 * 2 * (size_t)half_bytes((TH + 2) * (TW + 2));
    constexpr size_t xbytes = KS == 2 ? (size_t)4 * (BN / 64) * (TH * TW / 16) * 64 * 16 : 0;   // accumulator exchange of the 8-wave form
    constexpr size_t smem_max = std::max(smem + xf_nk(XF) * XF_MAX_CIN * sizeof(float), xbytes);
    static std::once_flag attr_once;
    static bool attr_ok = false;
    std::call_once(attr_once, [] {
        attr_ok = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_halo_kernel<T, TH, TW, BN, XF, KS>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_max) == hipSuccess;
    });
    if (!attr_ok) return GDRN_ERR_LAUNCH;
    const int grid = N * (p.Ho / TH) * (p.Wo / TW) * cdiv(p.Cout, BN);
    // a single 128-byte channel chunk (Cin = 64) never touches the second patch buffer: half the LDS -> a third workgroup per
    // CU on the 64-channel variants (142 VGPRs), whose runs are all prologue / one chunk / epilogue
    const size_t smem_used = std::max(((p.Cin * (int)sizeof(T) == ROWB) ? smem / 2 : smem) + (size_t)xf_nk(XF) * p.Cin * sizeof(float), xbytes);
    GDRN_LAUNCH((conv3x3_halo_kernel<T, TH, TW, BN, XF, KS>), dim3(grid), dim3(256 * KS), smem_used, st, p);
    GDRN_CHECK_LAUNCH();
    return GDRN_OK;
}

// 8-wave form (kernel comment): p.halo_waves = 8 asks for it, 4 forbids it, 0 leaves the choice to the grid -- a launch that gives no CU a
// second workgroup (<= 256 workgroups of the 128-channel tile) runs it
__host__ inline bool halo_split(const gdrn_conv_params& p, int th, int tw, int bn, int N) {
    if (bn != 128 || p.dtype != GDRN_DT_H16 || p.halo_waves == 4) return false;
    if (p.halo_waves == 8) return true;
    return (long long)N * (p.Ho / th) * (p.Wo / tw) * cdiv(p.Cout, bn) <= 256;
}

template <int XF>
int launch_tile(const gdrn_conv_params& p, int tw, int bn, int N, hipStream_t st) {
    if (halo_split(p, 8, tw, bn, N)) return tw == 16 ? launch<bf16_t, 8, 16, 128, XF, 2>(p, N, st) : launch<bf16_t, 8, 8, 128, XF, 2>(p, N, st);
    if (tw == 16) return bn == 64 ? launch<bf16_t, 8, 16, 64, XF>(p, N, st) : launch<bf16_t, 8, 16, 128, XF>(p, N, st);
    return bn == 64 ? launch<bf16_t, 8, 8, 64, XF>(p, N, st) : launch<bf16_t, 8, 8, 128, XF>(p, N, st);
}

// fp32 (parity mode): the 64-channel tile, no operand transform
int launch_tile_f32(const gdrn_conv_params& p, int tw, int N, hipStream_t st) {
    return tw == 8 ? launch<float, 8, 8, 64, 0>(p, N, st) : GDRN_ERR_SHAPE;
}

}  // namespace

// second-generation kernel (conv3x3_v3.hip), selected by p->w_frag == 2
int gdrn_v3_config(const gdrn_conv_params* p);
int gdrn_v3_preferred(const gdrn_conv_params* p);
int gdrn_v3_tile(const gdrn_conv_params* p, int* th, int* tw, int* bn);
int gdrn_v3_launch(const gdrn_conv_params* p, void* stream);

// operand layout the library prefers for a shape: 2 = gdrn_pack_wfrag32 (v3 kernel), 1 = gdrn_pack_wfrag, 0 = no halo tiling
// (p->w_frag carries the caller's policy INTO this query: 0 = the library's choice -- the second-generation kernel where it measured faster --,
//  1 = never the second-generation kernel, 2 = wherever it covers the shape; A/B runs and the plan variants of the tests)
extern "C" int gdrn_conv3x3_wfrag(const gdrn_conv_params* p) {
    if (!p) return GDRN_ERR_ARG;
    if (p->w_frag < 0 || p->w_frag > 2) return GDRN_ERR_ARG;
    if (p->w_frag == 2 && gdrn_v3_config(p) > 0) return 2;
    if (p->w_frag == 0 && gdrn_v3_preferred(p)) return 2;
    gdrn_conv_params q = *p;
    q.w_frag = 0;
    int th, tw, bn;
    gdrn_conv3x3_tile(&q, &th, &tw, &bn);
    return th > 0 ? 1 : 0;
}

// tile of the halo kernel for a shape: th x tw output pixels, bn channels; 0 if the shape is not covered
extern "C" int gdrn_conv3x3_tile(const gdrn_conv_params* p, int* th, int* tw, int* bn) {
    if (!p || !th || !tw || !bn) return GDRN_ERR_ARG;
    *th = *tw = *bn = 0;
    if (p->w_frag == 2) { gdrn_v3_tile(p, th, tw, bn); return GDRN_OK; }
    if (p->mode != 0 || p->KH != 3 || p->KW != 3 || p->stride != 1 || p->pad != 1 || p->Hi != p->Ho || p->Wi != p->Wo) return GDRN_OK;
    if ((p->Ho % 8) || (p->Wo % 8)) return GDRN_OK;
    if (p->dtype != GDRN_DT_H16 && p->dtype != GDRN_DT_F32) return GDRN_OK;
    *th = 8;
    *tw = (p->Wo % 16 == 0 && p->dtype != GDRN_DT_F32) ? 16 : 8;
    // parity (fp32) mode: always the 8x8-pixel x 64-channel tile -- its per-stage partial accumulators (see the kernel) double the accumulator
    // registers (the 8x16 tile spills 160+ registers with them, the 128-channel tile cannot hold them at all); v_mfma_f32_16x16x4_f32 is
    // 16x slower than the 16-bit forms, so the small tile's higher LDS / weight traffic per MFMA does not matter
    *bn = (p->Cout <= 64 || p->dtype == GDRN_DT_F32) ? 64 : 128;
    // (measured and rejected: 4x16-pixel tiles -- 164 VGPRs, three workgroups per CU -- are 4 % slower over the step; 64-channel tiles
    //  for the small grids of the 8x8 / 16x16 maps: +0.1 ms.  The channel tile is a function of Cout alone: gdrn_pack_wfrag's row
    //  order depends on it.)
    return GDRN_OK;
}

// waves per workgroup of the launch gdrn_conv3x3_halo makes for p (first halo kernel): 4, or 8 = the K-split form; 0: not a launch of that kernel
extern "C" int gdrn_conv3x3_halo_waves(const gdrn_conv_params* p) {
    if (!p || p->w_frag == 2) return 0;
    int th, tw, bn;
    if (gdrn_conv3x3_tile(p, &th, &tw, &bn) != GDRN_OK || th == 0) return 0;
    return halo_split(*p, th, tw, bn, p->M / (p->Ho * p->Wo)) ? 8 : 4;
}

extern "C" int gdrn_conv3x3_stats_rows(const gdrn_conv_params* p) {
    int th, tw, bn;
    if (gdrn_conv3x3_tile(p, &th, &tw, &bn) != GDRN_OK || th == 0) return GDRN_ERR_SHAPE;
    const int hw = p->Ho * p->Wo;
    return (p->M / hw) * (p->Ho / th) * (p->Wo / tw);
}

extern "C" int gdrn_pack_wfrag(const void* src, void* dst, int rows, int Cin, int dtype, void* stream) {
    if (!src || !dst || rows <= 0 || (rows % 16) || Cin <= 0) return GDRN_ERR_ARG;
    const int esz = dtype == GDRN_DT_H16 ? 2 : 4;
    if ((Cin * esz) % ROWB) return GDRN_ERR_SHAPE;
    const long long n = (long long)rows * 9 * Cin * esz / 16;
    const int grid = (int)std::min<long long>((n + 255) / 256, 4096);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (dtype == GDRN_DT_F32) GDRN_LAUNCH(pack_wfrag_kernel<float>, dim3(grid), dim3(256), 0, st, (const float*)src, (float*)dst, rows, Cin);
    else if (dtype == GDRN_DT_H16) GDRN_LAUNCH(pack_wfrag_kernel<bf16_t>, dim3(grid), dim3(256), 0, st, (const bf16_t*)src, (bf16_t*)dst, rows, Cin);
    else return GDRN_ERR_ARG;
    GDRN_CHECK_LAUNCH();
    return GDRN_OK;
}

// p->w must be the FRAGMENT-MAJOR operand produced by gdrn_pack_wfrag from the row-major [w_rows][9][Cin] one.
extern "C" int gdrn_conv3x3_halo(const gdrn_conv_params* pp, void* stream) {
    if (!pp || !pp->x || !pp->w || !pp->y) return GDRN_ERR_ARG;
    const gdrn_conv_params& p = *pp;
    if (p.dtype != GDRN_DT_F32 && p.dtype != GDRN_DT_H16) return GDRN_ERR_ARG;
    if (p.w_frag == 2) return gdrn_v3_launch(pp, stream);
    const int esz = p.dtype == GDRN_DT_H16 ? 2 : 4;
    int th, tw, bn;
    gdrn_conv3x3_tile(pp, &th, &tw, &bn);
    if (th == 0) return GDRN_ERR_SHAPE;
    if (p.Cin <= 0 || (p.Cin * esz) % ROWB != 0 || (p.x_cs * esz) % 16 != 0 || p.Cout <= 0 || (p.y_cs & 3)) return GDRN_ERR_SHAPE;
    if (bn == 128 && ((p.y_cs & 7) || (p.addend && (p.add_cs & 7)) || (p.bnb_x && (p.bnb_cs & 7)))) return GDRN_ERR_SHAPE;  // 16-byte epilogue accesses
    const int hw = p.Ho * p.Wo;
    if (p.M <= 0 || p.M % hw != 0) return GDRN_ERR_SHAPE;
    if (p.w_rows < cdiv(p.Cout, bn) * bn) return GDRN_ERR_SHAPE;
    if (p.addend && (p.add_cs & 3)) return GDRN_ERR_SHAPE;
    if (p.w_frag < 0 || p.w_frag > 2) return GDRN_ERR_ARG;   // (ABI 1's padding field: an uninitialised value must not pick a kernel)
    if (p.halo_waves != 0 && p.halo_waves != 4 && p.halo_waves != 8) return GDRN_ERR_ARG;
    if (p.bnb_x) {  // fused BatchNorm-backward statistics: fast epilogue only
        if (!p.bnb_mean || !p.bnb_invstd || !p.bnb_rows || (p.bnb_scale != nullptr) != (p.bnb_shift != nullptr)) return GDRN_ERR_ARG;
        if (p.bias || p.act || p.out_f32 || (p.Cout % bn) || (p.bnb_cs & 3) || p.bnb_cs < p.Cout) return GDRN_ERR_SHAPE;
        if ((unsigned long long)p.M * (unsigned long long)p.bnb_cs * 4ull >= (1ull << 32)) return GDRN_ERR_SHAPE;
    }
    if ((unsigned long long)p.M * (unsigned long long)std::max(p.y_cs, p.add_cs) * 4ull >= (1ull << 32)) return GDRN_ERR_SHAPE;  // 32-bit offsets
    if (p.dtype == GDRN_DT_F32 && (p.xf_mode || p.bnb_x)) return GDRN_ERR_ARG;   // transforms / fused BatchNorm-backward epilogue: 16-bit only
    if (p.xf_mode) {  // operand transform while staging
        if (p.xf_mode < 0 || p.xf_mode > 4 || !p.xf_c || p.Cin > XF_MAX_CIN || (p.Cin & 7)) return GDRN_ERR_ARG;
        if (p.xf_mode >= 2 && !p.xf_x2) return GDRN_ERR_ARG;
        if (p.xf_mode == 4 && (!p.xf_msc || !p.xf_msh)) return GDRN_ERR_ARG;
        if (p.xf_mode != 2 && p.xf_c2) return GDRN_ERR_ARG;
    }
    const int N = p.M / hw;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (p.dtype == GDRN_DT_F32) return launch_tile_f32(p, tw, N, st);
    switch (p.xf_mode) {
        case 0: return launch_tile<0>(p, tw, bn, N, st);
        case 1: return launch_tile<1>(p, tw, bn, N, st);
        case 2: return launch_tile<2>(p, tw, bn, N, st);
        case 3: return launch_tile<3>(p, tw, bn, N, st);
        default: return launch_tile<4>(p, tw, bn, N, st);
    }
}
